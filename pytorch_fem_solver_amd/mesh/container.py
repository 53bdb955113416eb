"""Nested mesh container.

The reference stores a mesh in a third-party ``tensordict.TensorDict``
(abstract_mesh.py:60-74) and only uses it as a nested dictionary addressed by
``mesh["cells"]`` / ``mesh["cells", "vertices"]``.  ``MeshData`` provides that
behaviour without the dependency; a real TensorDict (or any mapping) is accepted
as *input* by the mesh classes.
"""

from __future__ import annotations

from collections.abc import Mapping

import torch


class MeshData(dict):
    """dict with tuple-path access and a TensorDict-like ``batch_size``."""

    def __init__(self, source=None, batch_size=None, **kwargs):
        super().__init__()
        self.batch_size = torch.Size(batch_size or [])
        items = dict(source.items()) if source is not None else {}
        items.update(kwargs)
        for key, value in items.items():
            is_nested = isinstance(value, Mapping) and not isinstance(value, MeshData)
            dict.__setitem__(self, key, MeshData(value) if is_nested else value)

    def __getitem__(self, key):
        if isinstance(key, tuple):
            node = self
            for part in key:
                node = dict.__getitem__(node, part)
            return node
        return dict.__getitem__(self, key)

    def __setitem__(self, key, value):
        if isinstance(key, tuple):
            node = self
            for part in key[:-1]:
                if part not in node:
                    dict.__setitem__(node, part, MeshData())
                node = dict.__getitem__(node, part)
            dict.__setitem__(node, key[-1], value)
        else:
            dict.__setitem__(self, key, value)

    def auto_batch_size_(self):
        """Common leading dimensions of all tensor leaves (children first)."""
        shapes = []
        for value in self.values():
            if isinstance(value, MeshData):
                value.auto_batch_size_()
                shapes.append(tuple(value.batch_size))
            elif isinstance(value, torch.Tensor):
                shapes.append(tuple(value.shape))
        common = []
        for dims in zip(*shapes) if shapes else ():
            if len(set(dims)) != 1:
                break
            common.append(dims[0])
        self.batch_size = torch.Size(common)
        return self

    def to(self, device):
        out = MeshData(batch_size=self.batch_size)
        for key, value in self.items():
            moved = value.to(device) if isinstance(value, (torch.Tensor, MeshData)) else value
            dict.__setitem__(out, key, moved)
        return out
