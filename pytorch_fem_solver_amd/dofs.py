"""Degree-of-freedom numbering helpers.

P1: DoFs = vertices (reference torch_fem/basis/basis.py:22-24).
P2: the reference has shape functions for order 2 (element_tri.py:43-70) but no
global numbering -- ``Basis._compute_dofs`` raises (basis.py:50-51); the intent
left in its commented-out block (basis.py:26-49) is "vertex DoFs, then
``edge_id + N_v``".  This module defines exactly that, with the local edge order
of the P2 shape functions: (v0,v1), (v1,v2), (v2,v0) (element_tri.py:50-52).
"""

from __future__ import annotations

import numpy as np

__all__ = ["p2_dofs_numpy", "p2_dofs_torch", "edge_ids_for_cells"]


def edge_ids_for_cells(triangles: np.ndarray, edges: np.ndarray) -> np.ndarray:
    """(N_T,3) index into ``edges`` of the local edges (v0,v1), (v1,v2), (v2,v0)."""
    tri = np.asarray(triangles, dtype=np.int64)
    e = np.sort(np.asarray(edges, dtype=np.int64), axis=1)
    nv = int(max(tri.max(), e.max())) + 1
    keys = e[:, 0] * nv + e[:, 1]
    order = np.argsort(keys, kind="stable")
    sorted_keys = keys[order]
    local = np.stack([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]], axis=1)
    local.sort(axis=2)
    want = local[..., 0] * nv + local[..., 1]
    pos = np.searchsorted(sorted_keys, want)
    pos = np.clip(pos, 0, sorted_keys.shape[0] - 1)
    if not np.array_equal(sorted_keys[pos], want):
        raise ValueError("mesh 'edges' does not contain every edge of every triangle")
    return order[pos]


def p2_dofs_numpy(vertices, triangles, edges, edge_markers, vertex_markers):
    """Return (conn6 int32 (N_T,6), dof coordinates (N_v+N_e,2), markers (N_v+N_e,1))."""
    vertices = np.asarray(vertices, dtype=np.float64)
    nv = vertices.shape[0]
    eid = edge_ids_for_cells(triangles, edges)
    conn6 = np.concatenate(
        [np.asarray(triangles, dtype=np.int64), eid + nv], axis=1
    ).astype(np.int32)
    mid = vertices[np.asarray(edges, dtype=np.int64)].mean(axis=1)
    coords = np.concatenate([vertices, mid], axis=0)
    markers = np.concatenate(
        [np.asarray(vertex_markers).reshape(-1, 1), np.asarray(edge_markers).reshape(-1, 1)]
    ).astype(np.int32)
    return conn6, coords, markers


def p2_dofs_torch(vertices, triangles, edges, edge_markers, vertex_markers):
    """The same numbering with torch operations on the tensors' own device (sort + binary
    search on the GPU for a device-resident mesh: the P2 numbering of a 1e6-element mesh takes
    milliseconds instead of the seconds of the numpy path).  Returns (conn6 int32 (N_T,6), dof
    coordinates (N_v+N_e,2), markers int32 (N_v+N_e,1)) on that device."""
    import torch

    tri = triangles.long()
    e = torch.sort(edges.long(), dim=1)[0]
    nv = vertices.shape[0]
    keys = e[:, 0] * nv + e[:, 1]
    sorted_keys, order = torch.sort(keys, stable=True)
    local = torch.stack([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]], dim=1)
    local = torch.sort(local, dim=2)[0]
    want = (local[..., 0] * nv + local[..., 1]).contiguous()
    pos = torch.searchsorted(sorted_keys, want).clamp_(0, max(sorted_keys.shape[0] - 1, 0))
    if not torch.equal(sorted_keys[pos], want):
        raise ValueError("mesh 'edges' does not contain every edge of every triangle")
    eid = order[pos]
    conn6 = torch.cat([tri, eid + nv], dim=1).to(torch.int32)
    mid = vertices[edges.long()].mean(dim=1)
    coords = torch.cat([vertices, mid], dim=0)
    markers = torch.cat([vertex_markers.reshape(-1, 1), edge_markers.reshape(-1, 1)]).to(torch.int32)
    return conn6, coords, markers
