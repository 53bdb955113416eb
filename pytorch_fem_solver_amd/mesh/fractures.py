"""Stacks of equally-sized triangulations and planar fractures embedded in 3-D.

Mirror of reference torch_fem/mesh/meshes_tri.py and fractures_tri.py.  F is 1-2 in
every use; the per-mesh topology is computed mesh by mesh
with the single-mesh routines and stacked, which is what the reference's Python loops
over F amount to.
"""

from __future__ import annotations

from collections.abc import Mapping

import numpy as np
import torch

from . import topology
from .container import MeshData
from .tri import MeshTri, _KEY_MAP, _as_tensor


def _stack_meshdata(parts):
    first = parts[0]
    out = MeshData()
    for key, value in first.items():
        column = [p[key] for p in parts]
        if isinstance(value, MeshData):
            out[key] = _stack_meshdata(column)
        elif isinstance(value, torch.Tensor):
            out[key] = torch.stack(column, dim=0)
        else:
            out[key] = value
    return out.auto_batch_size_()


class MeshesTri(MeshTri):
    """F triangulations with identical array shapes, stacked on a leading dimension
    (meshes_tri.py:8-31)."""

    def __init__(self, triangulations):
        super().__init__(self._stack_triangulations(triangulations))

    def _stack_triangulations(self, fracture_triangulations):
        if isinstance(fracture_triangulations, _StackedInput):
            return fracture_triangulations
        return _StackedInput(list(fracture_triangulations))

    def _triangle_to_tensordict(self, mesh_dict):
        singles = []
        for tri in mesh_dict.parts:
            groups = {"vertices": {}, "cells": {}, "edges": {}}
            for key, value in tri.items():
                if key in _KEY_MAP:
                    tensor = _as_tensor(value)
                    if tensor is not None:
                        group, name = _KEY_MAP[key]
                        groups[group][name] = tensor
            singles.append(MeshData({k: MeshData(v) for k, v in groups.items()}))
        return singles

    def _build_optional_parameters(self, triangulation):
        completed = []
        for single in triangulation:
            data = topology.complete_single_mesh(single)
            # the batched reference keeps boundary cells as a column (meshes_tri.py:92)
            cells = data["boundary_edges", "cells"]
            if cells.dim() == 1:
                data["boundary_edges", "cells"] = cells.unsqueeze(1)
            completed.append(data)
        return _stack_meshdata(completed)

    @staticmethod
    def compute_coordinates_4_cells(coordinates_4_vertices, vertices_4_cells):
        """Per-mesh gather (meshes_tri.py:33-41)."""
        batch = torch.arange(coordinates_4_vertices.size(0))[:, None, None]
        return coordinates_4_vertices[batch, vertices_4_cells]

    @staticmethod
    def apply_mask(tensor, mask):
        """Per-mesh boolean/integer indexing (meshes_tri.py:43-52)."""
        return torch.cat([t[m].unsqueeze(0) for t, m in zip(tensor, mask)], dim=0)


class _StackedInput:
    """The list of triangulation dictionaries, tagged so a second stacking is a no-op
    (the reference stacks twice, fractures_tri.py:12 + meshes_tri.py:13)."""

    def __init__(self, parts):
        self.parts = [dict(p.items()) if isinstance(p, Mapping) else p for p in parts]


class FracturesTri(MeshesTri):
    """Planar fractures: each 2-D mesh is mapped affinely into 3-D (fractures_tri.py:7-33)."""

    def __init__(self, triangulations, fractures_3d_data: torch.Tensor):
        super().__init__(triangulations)
        self._compute_fracture_map(fractures_3d_data)
        jac = self["jacobian_fracture_map"]
        shift = self["translation_vector"]
        self["vertices", "coordinates_3d"] = (jac @ self["vertices", "coordinates"].mT + shift).mT
        self["cells", "coordinates_3d"] = self.compute_coordinates_4_cells(
            self["vertices", "coordinates_3d"], self["cells", "vertices"]
        )
        self["interior_edges", "normals_3d"] = (
            jac.unsqueeze(-3) @ self["interior_edges", "normals"].mT + shift.unsqueeze(-3)
        ).mT

    def _compute_fracture_map(self, fractures_3d_data: torch.Tensor):
        """x3 = J x2 + t from the first three vertices of each mesh and of each fracture
        (fractures_tri.py:35-67)."""
        fractures_3d_data = torch.as_tensor(fractures_3d_data)
        corners_2d = self["vertices", "coordinates"][:, :3, :]
        corners_3d = fractures_3d_data[:, :3, :]
        homogeneous = torch.cat([corners_2d, torch.ones_like(corners_3d[..., [-1]])], dim=-1)
        affine = corners_3d.mT @ torch.inverse(homogeneous).mT
        jac = affine[..., :2]
        shift = affine[..., [-1]]
        col_a, col_b = torch.split(jac, 1, dim=-1)
        area_factor = torch.norm(
            torch.cross(col_a, col_b, dim=-2), p=2, dim=-2, keepdim=True, dtype=jac.dtype
        )
        pseudo_inverse = torch.inverse(jac.mT @ jac) @ jac.mT
        self["jacobian_fracture_map"] = jac
        self["inv_jacobian_fracture_map"] = pseudo_inverse
        self["det_jacobian_fracture_map"] = area_factor
        self["translation_vector"] = shift
