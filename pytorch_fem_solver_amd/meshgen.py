"""Synthetic triangle-format meshes (the input format of ``MeshTri``).

The reference meshes everything with the third-party ``triangle`` package
(reference tests/test_assembly.py:23-25, examples/example_fractures_fem.py:44-46),
which is not part of this build.  These generators emit the same dictionary
layout ``triangle.triangulate`` returns (keys consumed at reference
torch_fem/mesh/abstract_mesh.py:33-40):

    vertices        float64 (N_v, 2)
    vertex_markers  int32   (N_v, 1)   1 = boundary
    triangles       int32   (N_T, 3)   counter-clockwise
    edges           int32   (N_e, 2)
    edge_markers    int32   (N_e, 1)   1 = boundary
    neighbors       int32   (N_T, 3)   neighbour k is opposite corner k, -1 = none

Families:
    S(n, jitter, seed)  jittered structured unit square, N_T = 2 n^2
    D(N_v, seed)        Delaunay triangulation of random points (scipy)
    L(m)                the two-fracture geometry of example_fractures_fem.py on
                        [-1,1]x[0,1] with 2m x m cells
"""

from __future__ import annotations

import numpy as np

__all__ = [
    "structured_rectangle",
    "unit_square",
    "delaunay_square",
    "fracture_rectangle",
    "morton_order",
    "permute_mesh",
]


def structured_rectangle(
    nx: int,
    ny: int,
    x0: float = 0.0,
    x1: float = 1.0,
    y0: float = 0.0,
    y1: float = 1.0,
    jitter: float = 0.0,
    seed: int = 0,
) -> dict:
    """Grid of ``nx`` x ``ny`` cells, every cell cut along its (0,0)-(1,1) diagonal.

    Vertex id = iy*(nx+1)+ix, triangle ids 2*(iy*nx+ix)+{0,1}.  Interior vertices
    are displaced by U(-jitter*h, +jitter*h) per coordinate (h = cell size).
    """
    nvx, nvy = nx + 1, ny + 1
    gx, gy = np.meshgrid(np.arange(nvx), np.arange(nvy), indexing="xy")
    hx, hy = (x1 - x0) / nx, (y1 - y0) / ny
    xs = x0 + gx.astype(np.float64) * hx
    ys = y0 + gy.astype(np.float64) * hy
    on_boundary = (gx == 0) | (gx == nx) | (gy == 0) | (gy == ny)
    if jitter > 0.0:
        rng = np.random.default_rng(seed)
        dx = rng.uniform(-jitter * hx, jitter * hx, size=xs.shape)
        dy = rng.uniform(-jitter * hy, jitter * hy, size=ys.shape)
        xs = np.where(on_boundary, xs, xs + dx)
        ys = np.where(on_boundary, ys, ys + dy)
    vertices = np.stack([xs.reshape(-1), ys.reshape(-1)], axis=1)
    vertex_markers = on_boundary.reshape(-1, 1).astype(np.int32)

    cx, cy = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
    cx = cx.reshape(-1)
    cy = cy.reshape(-1)
    v00 = cy * nvx + cx
    v10 = v00 + 1
    v01 = v00 + nvx
    v11 = v01 + 1
    lower = np.stack([v00, v10, v11], axis=1)
    upper = np.stack([v00, v11, v01], axis=1)
    triangles = np.empty((2 * nx * ny, 3), dtype=np.int32)
    triangles[0::2] = lower
    triangles[1::2] = upper

    cell = cy * nx + cx
    none = np.full_like(cell, -1)
    lower_nb = np.stack(
        [
            np.where(cx < nx - 1, 2 * (cell + 1) + 1, none),  # across right edge
            2 * cell + 1,  # across the diagonal
            np.where(cy > 0, 2 * (cell - nx) + 1, none),  # across bottom edge
        ],
        axis=1,
    )
    upper_nb = np.stack(
        [
            np.where(cy < ny - 1, 2 * (cell + nx), none),  # across top edge
            np.where(cx > 0, 2 * (cell - 1), none),  # across left edge
            2 * cell,  # across the diagonal
        ],
        axis=1,
    )
    neighbors = np.empty((2 * nx * ny, 3), dtype=np.int32)
    neighbors[0::2] = lower_nb
    neighbors[1::2] = upper_nb

    # horizontal, vertical, diagonal edges
    hx_i, hy_i = np.meshgrid(np.arange(nx), np.arange(nvy), indexing="xy")
    h_a = (hy_i * nvx + hx_i).reshape(-1)
    h_edges = np.stack([h_a, h_a + 1], axis=1)
    h_mark = ((hy_i == 0) | (hy_i == ny)).reshape(-1)
    vx_i, vy_i = np.meshgrid(np.arange(nvx), np.arange(ny), indexing="xy")
    v_a = (vy_i * nvx + vx_i).reshape(-1)
    v_edges = np.stack([v_a, v_a + nvx], axis=1)
    v_mark = ((vx_i == 0) | (vx_i == nx)).reshape(-1)
    d_edges = np.stack([v00, v11], axis=1)
    d_mark = np.zeros(d_edges.shape[0], dtype=bool)
    edges = np.concatenate([h_edges, v_edges, d_edges], axis=0).astype(np.int32)
    edge_markers = (
        np.concatenate([h_mark, v_mark, d_mark]).reshape(-1, 1).astype(np.int32)
    )

    return {
        "vertices": np.ascontiguousarray(vertices),
        "vertex_markers": vertex_markers,
        "triangles": triangles,
        "edges": edges,
        "edge_markers": edge_markers,
        "neighbors": neighbors,
    }


def unit_square(n: int, jitter: float = 0.25, seed: int = 0) -> dict:
    """Mesh family S(n, jitter, seed): N_T = 2 n^2, N_v = (n+1)^2."""
    return structured_rectangle(n, n, jitter=jitter, seed=seed)


def _edges_from_triangles(triangles: np.ndarray):
    """Unique undirected edges + boundary flag (edge seen by one triangle only)."""
    t = triangles.astype(np.int64)
    pairs = np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]], axis=0)
    pairs.sort(axis=1)
    nv = int(t.max()) + 1
    keys = pairs[:, 0] * nv + pairs[:, 1]
    uniq, counts = np.unique(keys, return_counts=True)
    edges = np.stack([uniq // nv, uniq % nv], axis=1).astype(np.int32)
    return edges, (counts == 1)


def delaunay_square(n_vertices: int, seed: int = 0) -> dict:
    """Mesh family D(N_v, seed): boundary grid points + uniform random interior points."""
    from scipy.spatial import Delaunay

    rng = np.random.default_rng(seed)
    nb = max(2, int(round(np.sqrt(n_vertices))))
    t = np.linspace(0.0, 1.0, nb + 1)
    boundary = np.concatenate(
        [
            np.stack([t[:-1], np.zeros(nb)], 1),
            np.stack([np.ones(nb), t[:-1]], 1),
            np.stack([t[:0:-1], np.ones(nb)], 1),
            np.stack([np.zeros(nb), t[:0:-1]], 1),
        ]
    )
    n_int = max(0, n_vertices - boundary.shape[0])
    h = 1.0 / nb
    interior = rng.uniform(0.5 * h, 1.0 - 0.5 * h, size=(n_int, 2))
    pts = np.concatenate([boundary, interior])
    tri = Delaunay(pts)
    simplices = tri.simplices.astype(np.int32)
    neighbors = tri.neighbors.astype(np.int32)
    p = pts[simplices]
    area2 = (p[:, 1, 0] - p[:, 0, 0]) * (p[:, 2, 1] - p[:, 0, 1]) - (
        p[:, 2, 0] - p[:, 0, 0]
    ) * (p[:, 1, 1] - p[:, 0, 1])
    flip = area2 < 0
    simplices[flip] = simplices[flip][:, [0, 2, 1]]
    neighbors[flip] = neighbors[flip][:, [0, 2, 1]]
    keep = np.abs(area2) > 1e-14  # drop degenerate slivers on the boundary grid
    if not keep.all():
        remap = np.cumsum(keep) - 1
        simplices = simplices[keep]
        neighbors = neighbors[keep]
        neighbors = np.where(
            (neighbors >= 0) & keep[np.clip(neighbors, 0, None)],
            remap[np.clip(neighbors, 0, None)],
            -1,
        ).astype(np.int32)
    edges, is_boundary = _edges_from_triangles(simplices)
    vertex_markers = np.zeros((pts.shape[0], 1), dtype=np.int32)
    vertex_markers[edges[is_boundary].reshape(-1)] = 1
    return {
        "vertices": np.ascontiguousarray(pts),
        "vertex_markers": vertex_markers,
        "triangles": np.ascontiguousarray(simplices),
        "edges": edges,
        "edge_markers": is_boundary.reshape(-1, 1).astype(np.int32),
        "neighbors": np.ascontiguousarray(neighbors),
    }


def fracture_rectangle(m: int, jitter: float = 0.0, seed: int = 0) -> dict:
    """Mesh family L(m): [-1,1]x[0,1], 2m x m cells.

    The first six vertices are the planar-straight-line-graph corners in the
    order the fracture example lists them (reference
    examples/example_fractures_fem.py:32-40), because the fracture map is
    built from the first three vertices (reference
    torch_fem/mesh/fractures_tri.py:37-39).  The line x = 0 (the trace of the
    second fracture) is a mesh line and is never displaced by ``jitter``.
    """
    mesh = structured_rectangle(2 * m, m, -1.0, 1.0, 0.0, 1.0, jitter=0.0)
    nvx = 2 * m + 1
    if jitter > 0.0:
        rng = np.random.default_rng(seed)
        v = mesh["vertices"]
        ix = np.arange(v.shape[0]) % nvx
        free = (mesh["vertex_markers"][:, 0] == 0) & (ix != m)
        h = 1.0 / m
        v[free] += rng.uniform(-jitter * h, jitter * h, size=(int(free.sum()), 2))

    def vid(ix, iy):
        return iy * nvx + ix

    first = [vid(0, 0), vid(2 * m, 0), vid(0, m), vid(2 * m, m), vid(m, 0), vid(m, m)]
    nv = mesh["vertices"].shape[0]
    rest = np.setdiff1d(np.arange(nv), np.array(first), assume_unique=False)
    new_to_old = np.concatenate([np.array(first), rest])
    return permute_mesh(mesh, vertex_order=new_to_old, sort_edges=False)  # edge order of the committed fixtures


def morton_order(points: np.ndarray, bits: int = 16) -> np.ndarray:
    """Permutation sorting 2-D points along a Z-order curve."""
    p = np.asarray(points, dtype=np.float64)
    lo = p.min(axis=0)
    span = np.maximum(p.max(axis=0) - lo, 1e-300)
    q = np.minimum(((p - lo) / span * (2**bits)).astype(np.uint64), 2**bits - 1)

    def spread(x):
        x = x & np.uint64(0xFFFFFFFF)
        x = (x | (x << np.uint64(16))) & np.uint64(0x0000FFFF0000FFFF)
        x = (x | (x << np.uint64(8))) & np.uint64(0x00FF00FF00FF00FF)
        x = (x | (x << np.uint64(4))) & np.uint64(0x0F0F0F0F0F0F0F0F)
        x = (x | (x << np.uint64(2))) & np.uint64(0x3333333333333333)
        x = (x | (x << np.uint64(1))) & np.uint64(0x5555555555555555)
        return x

    key = spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1))
    return np.argsort(key, kind="stable")


def permute_mesh(mesh: dict, vertex_order=None, triangle_order=None, sort_edges=True) -> dict:
    """Renumber vertices (``vertex_order[new] = old``) and/or reorder triangles.  With a vertex
    renumbering the edge list is re-sorted by its (smaller, larger) new vertex ids (``sort_edges``):
    the numbering of the edges -- the edge DoFs of a P2 basis -- then has the locality of the vertex
    numbering instead of the generator's order."""
    out = {k: np.array(v, copy=True) for k, v in mesh.items()}
    if vertex_order is not None:
        vertex_order = np.asarray(vertex_order)
        old_to_new = np.empty_like(vertex_order)
        old_to_new[vertex_order] = np.arange(vertex_order.shape[0])
        out["vertices"] = np.ascontiguousarray(mesh["vertices"][vertex_order])
        out["vertex_markers"] = np.ascontiguousarray(mesh["vertex_markers"][vertex_order])
        out["triangles"] = old_to_new[mesh["triangles"]].astype(np.int32)
        if "edges" in mesh:
            out["edges"] = old_to_new[mesh["edges"]].astype(np.int32)
            if sort_edges:
                lo, hi = out["edges"].min(axis=1).astype(np.int64), out["edges"].max(axis=1).astype(np.int64)
                order = np.argsort(lo * vertex_order.shape[0] + hi, kind="stable")
                out["edges"] = np.ascontiguousarray(out["edges"][order])
                if "edge_markers" in mesh:
                    out["edge_markers"] = np.ascontiguousarray(out["edge_markers"][order])
    if triangle_order is not None:
        triangle_order = np.asarray(triangle_order)
        out["triangles"] = np.ascontiguousarray(out["triangles"][triangle_order])
        if "neighbors" in mesh:
            old_to_new_t = np.empty_like(triangle_order)
            old_to_new_t[triangle_order] = np.arange(triangle_order.shape[0])
            nb = out["neighbors"][triangle_order]
            out["neighbors"] = np.where(
                nb >= 0, old_to_new_t[np.clip(nb, 0, None)], -1
            ).astype(np.int32)
    return out
