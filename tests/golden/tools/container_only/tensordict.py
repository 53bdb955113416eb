"""Container-only stand-in for the third-party ``tensordict`` package.

TEST TOOLING, used by ``tests/golden/tools/make_golden.py`` in the build
container only.  The reference imports ``tensordict`` purely as a nested
dictionary container (reference torch_fem/mesh/abstract_mesh.py:6,60-74); the
package is absent from this image and cannot be fetched.  This module provides
that container behaviour and nothing else: it performs NO arithmetic, so every
number in the golden fixtures is produced by the reference's own torch code.
It is never imported by the product package, the tests or the bench.
"""

import numpy as np
import torch


def _leading_shape(value):
    if isinstance(value, TensorDict):
        return tuple(value.batch_size)
    if isinstance(value, (torch.Tensor, np.ndarray)):
        return tuple(value.shape)
    return None


class TensorDict:
    def __init__(self, source=None, batch_size=None, **kwargs):
        self._data = {}
        self.batch_size = torch.Size(batch_size if batch_size is not None else [])
        if isinstance(source, TensorDict):
            source = source._data
        items = dict(source or {})
        items.update(kwargs)
        for key, value in items.items():
            self._data[key] = TensorDict(value) if isinstance(value, dict) else value

    # --- mapping protocol -------------------------------------------------
    def __getitem__(self, key):
        if isinstance(key, tuple) and all(isinstance(k, str) for k in key):
            node = self
            for part in key:
                node = node._data[part]
            return node
        if isinstance(key, str):
            return self._data[key]
        return TensorDict(
            {
                k: (v[key] if _leading_shape(v) is not None else v)
                for k, v in self._data.items()
            }
        ).auto_batch_size_()

    def __setitem__(self, key, value):
        if isinstance(key, tuple):
            node = self
            for part in key[:-1]:
                if part not in node._data:
                    node._data[part] = TensorDict({})
                node = node._data[part]
            node._data[key[-1]] = value
        else:
            self._data[key] = value

    def __contains__(self, key):
        return key in self._data

    def __iter__(self):
        if len(self.batch_size) == 0:
            raise TypeError("iteration over a TensorDict without batch dimension")
        for i in range(self.batch_size[0]):
            yield self[i]

    def keys(self):
        return self._data.keys()

    def values(self):
        return self._data.values()

    def items(self):
        return self._data.items()

    # --- batch size -------------------------------------------------------
    def auto_batch_size_(self):
        shapes = []
        for value in self._data.values():
            if isinstance(value, TensorDict):
                value.auto_batch_size_()
            shape = _leading_shape(value)
            if shape is not None:
                shapes.append(shape)
        common = []
        if shapes:
            for dims in zip(*shapes):
                if all(d == dims[0] for d in dims):
                    common.append(dims[0])
                else:
                    break
        self.batch_size = torch.Size(common)
        return self


def stack(tensordicts, dim=0):
    first = tensordicts[0]
    out = {}
    for key, value in first.items():
        column = [td[key] for td in tensordicts]
        if isinstance(value, TensorDict):
            out[key] = stack(column, dim)
        elif isinstance(value, np.ndarray):
            out[key] = np.stack(column, axis=dim)
        elif isinstance(value, torch.Tensor):
            out[key] = torch.stack(column, dim=dim)
        else:
            out[key] = value
    return TensorDict(out).auto_batch_size_()
