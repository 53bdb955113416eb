"""``MeshTri``: one 2-D triangulation (mirror of reference torch_fem/mesh/abstract_mesh.py
+ mesh_tri.py).  Input: a ``triangle``-format dictionary (see meshgen.py)."""

from __future__ import annotations

import abc
from collections.abc import Mapping
from typing import Any

import numpy as np
import torch

from .container import MeshData
from . import topology

#: triangle key -> (group, name), abstract_mesh.py:33-40
_KEY_MAP = {
    "vertices": ("vertices", "coordinates"),
    "vertex_markers": ("vertices", "markers"),
    "triangles": ("cells", "vertices"),
    "neighbors": ("cells", "neighbors"),
    "edges": ("edges", "vertices"),
    "edge_markers": ("edges", "markers"),
}


def _as_tensor(value):
    """int arrays -> torch.int32, float arrays -> default dtype (abstract_mesh.py:51-58).

    The reference accepts only numpy int32/float64 and silently drops anything else
    (SURVEY.md appendix C-2); torch tensors and other integer/float widths are accepted
    here as well.  Returns None for unsupported values.
    """
    if isinstance(value, torch.Tensor):
        if value.dtype.is_floating_point:
            return value.to(dtype=torch.get_default_dtype(), device=torch.get_default_device())
        if value.dtype in (torch.int32, torch.int64, torch.int16, torch.uint8):
            return value.to(dtype=torch.int32, device=torch.get_default_device())
        return None
    array = np.asarray(value)
    if np.issubdtype(array.dtype, np.integer):
        return torch.tensor(array, dtype=torch.int)
    if np.issubdtype(array.dtype, np.floating):
        return torch.tensor(array, dtype=torch.get_default_dtype())
    return None


def triangulation_to_meshdata(mesh_dict: Mapping) -> MeshData:
    groups = {"vertices": {}, "cells": {}, "edges": {}}
    for key, value in mesh_dict.items():
        if key not in _KEY_MAP:
            continue
        tensor = _as_tensor(value)
        if tensor is not None:
            group, name = _KEY_MAP[key]
            groups[group][name] = tensor
    return MeshData({name: MeshData(content) for name, content in groups.items()}).auto_batch_size_()


class AbstractMesh(abc.ABC):
    """Dictionary-like mesh (abstract_mesh.py:10-29)."""

    #: single meshes derive edge topology on first access: it is one-off preprocessing
    #: (10.9 s on CPU at 1e6 elements) that assembly never reads
    _lazy_topology = True

    def __init__(self, triangulation: Mapping[str, Any]):
        data = self._triangle_to_tensordict(triangulation)
        self._topology_pending = False
        if self._lazy_topology and isinstance(data, MeshData):
            data["cells", "coordinates"] = self.compute_coordinates_4_cells(
                data["vertices", "coordinates"], data["cells", "vertices"]
            )
            self._triangulation = data
            self._topology_pending = True
        else:
            self._triangulation = self._build_optional_parameters(data)

    def _needs_topology(self, key):
        head = key[0] if isinstance(key, tuple) else key
        if head in ("interior_edges", "boundary_edges", "edges"):
            return True
        return head == "cells" and (not isinstance(key, tuple) or key[1:] == ("length",))

    def _finish_topology(self):
        if self._topology_pending:
            self._topology_pending = False
            self._triangulation = self._build_optional_parameters(self._triangulation)

    def __getitem__(self, key):
        if self._topology_pending and self._needs_topology(key):
            self._finish_topology()
        return self._triangulation[key]

    def __setitem__(self, key, value):
        self._triangulation[key] = value

    def __contains__(self, key):
        self._finish_topology()
        return key in self._triangulation

    def batch_size(self):
        return self._triangulation.batch_size

    def _triangle_to_tensordict(self, mesh_dict):
        return triangulation_to_meshdata(mesh_dict)

    def _build_optional_parameters(self, triangulation: MeshData) -> MeshData:
        return topology.complete_single_mesh(triangulation)

    @staticmethod
    def compute_coordinates_4_cells(coordinates_4_vertices, vertices_4_cells):
        """``X[conn]`` (abstract_mesh.py:257-262)."""
        return topology.gather_rows(coordinates_4_vertices, vertices_4_cells)

    @property
    @abc.abstractmethod
    def _edges_permutations(self) -> torch.Tensor:
        ...


class MeshTri(AbstractMesh):
    """Triangular mesh (mesh_tri.py:7-12)."""

    def __init__(self, triangulation: Mapping[str, Any]):
        super().__init__(triangulation)

    @property
    def _edges_permutations(self):
        return torch.tensor(topology.TRI_EDGES)
