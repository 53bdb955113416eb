"""Mesh topology derived from a triangulation (single mesh; torch expressions).

One-off preprocessing outside the assembly kernel's scope (SURVEY.md section 2
row 7).  Behaviour follows reference torch_fem/mesh/abstract_mesh.py:104-309,
including the row order of every output -- see notes on each function -- so that
downstream jump computations see the same tensors.
"""

from __future__ import annotations

import torch

from .container import MeshData

#: local vertex pairs of the three edges of a triangle (mesh_tri.py:10-12)
TRI_EDGES = ((0, 1), (1, 2), (0, 2))


def gather_rows(table: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    """``table[index]`` (abstract_mesh.py:257-262)."""
    return table[index]


def edges_from_cells(cells: torch.Tensor):
    """Unique edges and their cell counts when the input has no ``edges``.

    The reference's version of this step cannot run (shape bug, SURVEY.md appendix
    C-1); this is the evidently intended result: endpoints sorted, one row per
    undirected edge, marker = number of cells sharing it (1 = boundary).
    """
    local = torch.tensor(TRI_EDGES, device=cells.device)
    pairs = torch.sort(cells[..., local].reshape(-1, 2), dim=-1)[0]
    unique, counts = torch.unique(pairs, dim=0, return_counts=True)
    return unique.to(cells.dtype), counts.to(torch.int32).unsqueeze(-1)


def split_edges_by_marker(edge_vertices, edge_markers):
    """Boundary edges carry marker 1, everything else is interior (abstract_mesh.py:183-196)."""
    flag = edge_markers.squeeze(-1)
    return edge_vertices[flag == 1], edge_vertices[flag != 1]


def cells_of_edges_from_neighbors(neighbors: torch.Tensor):
    """abstract_mesh.py:207-230.  Interior rows are the SORTED unique cell pairs, i.e.
    not in edge-list order -- reproduced on purpose."""
    n_cells, n_local = neighbors.shape[-2], neighbors.shape[-1]
    owner = torch.arange(n_cells, device=neighbors.device).repeat_interleave(n_local)
    other = neighbors.reshape(-1)
    has_neighbor = other != -1
    lo = torch.minimum(owner[has_neighbor], other[has_neighbor])
    hi = torch.maximum(owner[has_neighbor], other[has_neighbor])
    interior = torch.unique(torch.stack([lo, hi], dim=1), dim=0)
    boundary = owner[other == -1]
    return boundary, interior


def cells_of_edges_by_matching(cells, boundary_edges, interior_edges):
    """abstract_mesh.py:232-253 without the O(N_e N_T) comparison tensor.

    boundary: first cell containing both endpoints, shape (N_b, 1);
    interior: the two such cells in ascending order, shape (N_i, 2).
    """
    n_cells = cells.shape[0]
    local = torch.tensor(TRI_EDGES, device=cells.device)
    n_verts = int(cells.max()) + 1 if cells.numel() else 1
    pairs = torch.sort(cells.long()[:, local], dim=-1)[0].reshape(-1, 2)
    keys = pairs[:, 0] * n_verts + pairs[:, 1]
    owner = torch.arange(n_cells, device=cells.device).repeat_interleave(3)
    order = torch.argsort(keys * n_cells + owner)  # by edge, then by cell id
    keys, owner = keys[order], owner[order]

    def lookup(edges):
        e = torch.sort(edges.long(), dim=-1)[0]
        return torch.searchsorted(keys, e[:, 0] * n_verts + e[:, 1])

    first_b = lookup(boundary_edges)
    first_i = lookup(interior_edges)
    boundary = owner[first_b].unsqueeze(-1)
    interior = torch.stack([owner[first_i], owner[first_i + 1]], dim=-1)
    return boundary, interior


def interior_edge_geometry(vertex_xy, cell_xy, interior_vertices, interior_cells):
    """Coordinates, lengths and oriented unit normals (abstract_mesh.py:117-162)."""
    xy = gather_rows(vertex_xy, interior_vertices)
    start, end = torch.split(xy, 1, dim=-2)
    tangent = end - start
    length = torch.norm(tangent, dim=-1, keepdim=True)
    normal = tangent[..., [1, 0]] * torch.tensor([-1.0, 1.0]) / length
    # orient from the first listed cell towards the second
    centroids = gather_rows(cell_xy, interior_cells).mean(dim=-2)
    c_first, c_second = torch.split(centroids, 1, dim=-2)
    pointing = (normal * (c_second - c_first)).sum(dim=-1)
    normal[pointing < 0] *= -1
    return xy, length, normal


def cell_edge_lengths(vertex_xy, cells):
    """Per-cell edge lengths, shape (N_T, 3, 1, 1) (abstract_mesh.py:283-309; the min in
    the reference reduces a singleton dimension)."""
    local = torch.tensor(TRI_EDGES, device=cells.device)
    ends = torch.sort(cells[..., local], dim=-1)[0]
    xy = gather_rows(vertex_xy, ends)
    start, end = torch.split(xy, 1, dim=-2)
    lengths = torch.norm(end - start, dim=-1, keepdim=True)
    return torch.min(lengths, dim=-2, keepdim=True)[0]


def complete_single_mesh(data: MeshData) -> MeshData:
    """Fill in everything ``AbstractMesh._build_optional_parameters`` adds
    (abstract_mesh.py:76-102) for ONE triangulation (no batch dimension)."""
    vertex_xy = data["vertices", "coordinates"]
    cells = data["cells", "vertices"]
    if "coordinates" not in data["cells"]:
        data["cells", "coordinates"] = gather_rows(vertex_xy, cells)

    if "vertices" not in data["edges"]:
        edge_vertices, edge_markers = edges_from_cells(cells)
        data["edges", "vertices"] = edge_vertices
        data["edges", "markers"] = edge_markers

    boundary_vertices, interior_vertices = split_edges_by_marker(
        data["edges", "vertices"], data["edges", "markers"]
    )
    if "neighbors" in data["cells"]:
        boundary_cells, interior_cells = cells_of_edges_from_neighbors(data["cells", "neighbors"])
    else:
        boundary_cells, interior_cells = cells_of_edges_by_matching(
            cells, boundary_vertices, interior_vertices
        )
    interior_xy, interior_length, interior_normal = interior_edge_geometry(
        vertex_xy, data["cells", "coordinates"], interior_vertices, interior_cells
    )
    data["interior_edges"] = MeshData(
        {
            "cells": interior_cells,
            "vertices": interior_vertices,
            "coordinates": interior_xy,
            "length": interior_length,
            "normals": interior_normal,
        }
    ).auto_batch_size_()
    data["boundary_edges"] = MeshData(
        {
            "cells": boundary_cells,
            "vertices": boundary_vertices,
            "coordinates": gather_rows(vertex_xy, boundary_vertices),
        }
    ).auto_batch_size_()
    data["cells", "length"] = cell_edge_lengths(vertex_xy, cells)
    return data
