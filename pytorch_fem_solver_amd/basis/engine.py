"""Launches the HIP assembly kernels for one (mesh, element) pair.

Owns the device-resident inputs (coordinates, connectivity), the symbolic CSR pattern
and the slot map, and hands torch tensors to libtfem_hip through the C ABI
(include/tfem_assembly.h).  There is no torch/CPU implementation of these operations in
this package: without the library or without a GPU every method raises.
"""

from __future__ import annotations

import ctypes
import os
from ctypes import c_void_p

import numpy as np
import torch

from .. import _native
from ..sparse import CSRMatrix


class NoDeviceError(RuntimeError):
    pass


def _compute_device(home: torch.device) -> torch.device:
    if home.type == "cuda":
        return home
    if torch.cuda.is_available():
        return torch.device("cuda", torch.cuda.current_device())
    raise NoDeviceError(
        "torch_fem assembly runs on an MI355X through libtfem_hip and has no CPU "
        "fallback, but no GPU is visible to this process"
    )


class PatternHandle:
    """A live tfem_csr_pattern handle (with the connectivity it points into), for the plan
    builder that starts from its incidence; released explicitly or with the object."""

    def __init__(self, handle, conn):
        self.handle, self.conn = handle, conn

    def release(self):
        if self.handle is not None:
            _native.load().tfem_csr_pattern_destroy(self.handle)
            self.handle = self.conn = None

    def __del__(self):
        try:
            self.release()
        except Exception:  # interpreter shutdown: the library may be gone already
            pass


def pattern_host(conn_dof, n_dofs, keep=False):
    """CSR pattern of the operator (replaces the index tensors of basis.py:64-85): rowptr int64
    (N+1), colind int32 (nnz), numpy in, numpy out; one pass of the multi-threaded host builder.
    keep=True: also the live PatternHandle (third value) for ring_plan_host(pattern=...)."""
    lib = _native.load()
    conn = np.ascontiguousarray(np.asarray(conn_dof).astype(np.int32, copy=False))
    conn = conn.reshape(-1, conn.shape[-1])
    e, n = conn.shape
    handle, nnz = c_void_p(), ctypes.c_int64(0)
    _native.check(lib.tfem_csr_pattern_create(c_void_p(conn.ctypes.data), 4, e, n, int(n_dofs),
                                              ctypes.byref(handle), ctypes.byref(nnz)))
    keeper = PatternHandle(handle, conn)
    try:
        rowptr = np.empty(int(n_dofs) + 1, dtype=np.int64)
        colind = np.empty(max(nnz.value, 1), dtype=np.int32)
        _native.check(lib.tfem_csr_pattern_export(handle, c_void_p(rowptr.ctypes.data), c_void_p(colind.ctypes.data)))
    except BaseException:
        keeper.release()
        raise
    if keep:
        return rowptr, colind[: nnz.value], keeper
    keeper.release()
    return rowptr, colind[: nnz.value]


def slots_host(conn_dof, n_dofs, rowptr, colind):
    """slots int32 (E*n*n): per element entry the CSR position it adds to (the scatter / gather
    paths; the row-form plans do not need it)."""
    lib = _native.load()
    conn = np.ascontiguousarray(np.asarray(conn_dof).astype(np.int32, copy=False))
    conn = conn.reshape(-1, conn.shape[-1])
    e, n = conn.shape
    slots = np.empty(max(e * n * n, 1), dtype=np.int32)
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    colind = np.ascontiguousarray(colind, dtype=np.int32)
    _native.check(lib.tfem_csr_symbolic_slots(c_void_p(conn.ctypes.data), 4, e, n, int(n_dofs),
                                              c_void_p(rowptr.ctypes.data), c_void_p(colind.ctypes.data),
                                              c_void_p(slots.ctypes.data)))
    return slots[: e * n * n]


def symbolic_host(conn_dof, n_dofs):
    """Host symbolic phase of the C ABI (replaces basis.py:64-85): CSR pattern of the
    operator and, per element entry, the CSR position it adds to.  numpy in, numpy out:
    rowptr int64 (N+1), colind int32 (nnz), slots int32 (E*n*n)."""
    rowptr, colind = pattern_host(conn_dof, n_dofs)
    return rowptr, colind, slots_host(conn_dof, n_dofs, rowptr, colind)


#: default tile capacities: 45 KB of LDS per workgroup -> 3 workgroups per CU
TILE_DEFAULTS = {"own": 512, "acc": 4096, "vert": 704}


def tile_plan_host(conn, n_verts, coords, rowptr, colind, own_cap=None, acc_cap=None,
                   vert_cap=None, elem_cap=None):
    """Build the tile plan on the host (tfem_tile_plan_*).  Returns a dict of numpy arrays
    and the size vector, or raises NotImplementedError when the mesh does not fit the
    plan's format (rows longer than 16 entries)."""
    lib = _native.load()
    env = lambda key, default: int(os.environ.get(key, default))  # noqa: E731
    own_cap = own_cap or env("TFEM_TILE_OWN", TILE_DEFAULTS["own"])
    acc_cap = acc_cap or env("TFEM_TILE_ACC", TILE_DEFAULTS["acc"])
    vert_cap = vert_cap or env("TFEM_TILE_VERT", TILE_DEFAULTS["vert"])
    elem_cap = min(elem_cap or 1 << 30, lib.tfem_tile_capacity(0))
    vert_cap = min(vert_cap, lib.tfem_tile_capacity(1))
    own_cap = min(own_cap, lib.tfem_tile_capacity(2))
    conn = np.ascontiguousarray(np.asarray(conn).astype(np.int32, copy=False)).reshape(-1, 3)
    coords = np.ascontiguousarray(np.asarray(coords, dtype=np.float64)).reshape(-1, 2)
    rowptr = np.ascontiguousarray(np.asarray(rowptr, dtype=np.int64))
    colind = np.ascontiguousarray(np.asarray(colind, dtype=np.int32))
    handle = c_void_p()
    _native.check(
        lib.tfem_tile_plan_create(
            c_void_p(conn.ctypes.data), 4, conn.shape[0], int(n_verts),
            c_void_p(coords.ctypes.data), c_void_p(rowptr.ctypes.data),
            c_void_p(colind.ctypes.data), elem_cap, vert_cap, acc_cap, own_cap,
            ctypes.byref(handle),
        )
    )
    try:
        layout = np.zeros(24, dtype=np.int64)
        _native.check(lib.tfem_tile_plan_sizes(handle, c_void_p(layout.ctypes.data)))
        blob = np.zeros(int(layout[19]), dtype=np.uint8)
        _native.check(lib.tfem_tile_plan_pack(handle, c_void_p(blob.ctypes.data)))
    finally:
        lib.tfem_tile_plan_destroy(handle)
    return unpack_plan(blob, layout)


def unpack_plan(blob, layout):
    """Views of the packed plan's arrays (host) + the blob and layout themselves."""
    z = [int(x) for x in layout]

    def view(i, dtype, count):
        return np.frombuffer(blob, dtype=dtype, count=count, offset=z[12 + i])

    return {
        "blob": blob,
        "layout": np.ascontiguousarray(layout, dtype=np.int64),
        "sizes": np.asarray(layout[:12]),
        "desc": view(0, np.int32, 12 * z[0]),
        "records": view(1, np.uint32, z[20] * z[1]),
        "rec_words": z[20],
        "vert_gid": view(2, np.int32, z[2]),
        "row_loff": view(3, np.uint16, z[3]),
        "run_delta": view(4, np.int32, z[4]),
        "run_lstart": view(5, np.uint16, z[11]),
        "elem_id": view(6, np.int32, z[1]),
    }


#: ring-plan tile capacities: one owned row per lane of a 256-lane workgroup
RING_DEFAULTS = {"own": 256, "vert": 448}


def ring_plan_host(conn, n_verts, coords, rowptr, colind, own_cap=None, vert_cap=None, priority=None,
                   pattern=None):
    """Build the ring plan on the host (tfem_ring_plan_*): the row-form plan of the P1
    stiffness/mass kernel.  Raises NotImplementedError when the triangles around a vertex
    do not form fans (non-manifold edge, duplicated element) or a row has > 16 entries.
    priority: per-vertex flags; the tiles owning a flagged vertex come first ("n_priority").
    pattern: the live PatternHandle of this connectivity (pattern_host(keep=True)): the builder
    takes the vertex -> elements incidence from it instead of building it again."""
    lib = _native.load()
    env = lambda key, default: int(os.environ.get(key, default))  # noqa: E731
    own_cap = min(own_cap or env("TFEM_RING_OWN", RING_DEFAULTS["own"]), lib.tfem_ring_capacity(0))
    vert_cap = min(vert_cap or env("TFEM_RING_VERT", RING_DEFAULTS["vert"]), lib.tfem_ring_capacity(1))
    conn = np.ascontiguousarray(np.asarray(conn).astype(np.int32, copy=False)).reshape(-1, 3)
    coords = np.ascontiguousarray(np.asarray(coords, dtype=np.float64)).reshape(-1, 2)
    rowptr = np.ascontiguousarray(np.asarray(rowptr, dtype=np.int64))
    colind = np.ascontiguousarray(np.asarray(colind, dtype=np.int32))
    handle = c_void_p()
    n_priority = ctypes.c_int64(0)
    if priority is not None:
        priority = np.ascontiguousarray(np.asarray(priority).reshape(-1) != 0).view(np.uint8)
        if priority.shape[0] != int(n_verts):
            raise ValueError(f"priority: {priority.shape[0]} flags for {n_verts} vertices")
    flags = c_void_p(priority.ctypes.data) if priority is not None else None
    if pattern is not None and pattern.handle is not None:
        _native.check(
            lib.tfem_ring_plan_create_from_pattern(
                pattern.handle, c_void_p(coords.ctypes.data), c_void_p(colind.ctypes.data), own_cap,
                vert_cap, flags, ctypes.byref(handle), ctypes.byref(n_priority),
            )
        )
    else:
        _native.check(
            lib.tfem_ring_plan_create_priority(
                c_void_p(conn.ctypes.data), 4, conn.shape[0], int(n_verts),
                c_void_p(coords.ctypes.data), c_void_p(rowptr.ctypes.data),
                c_void_p(colind.ctypes.data), own_cap, vert_cap, flags, ctypes.byref(handle),
                ctypes.byref(n_priority),
            )
        )
    try:
        layout = np.zeros(32, dtype=np.int64)
        _native.check(lib.tfem_ring_plan_sizes(handle, c_void_p(layout.ctypes.data)))
        blob = np.empty(int(layout[12]), dtype=np.uint8)  # pack writes every byte
        _native.check(lib.tfem_ring_plan_pack(handle, c_void_p(blob.ctypes.data)))
    finally:
        lib.tfem_ring_plan_destroy(handle)
    plan = unpack_ring_plan(blob, layout)
    plan["n_priority"] = int(n_priority.value)
    return plan


def unpack_ring_plan(blob, layout):
    """Views of the packed ring plan's arrays (host) + the blob and layout themselves."""
    z = [int(x) for x in layout]

    def view(i, dtype, count):
        return np.frombuffer(blob, dtype=dtype, count=count, offset=z[8 + i])

    return {
        "blob": blob,
        "layout": np.ascontiguousarray(layout, dtype=np.int64),
        "n_tiles": z[0],
        "slots": z[6],
        "words": z[7],
        "chunked": bool(z[13]),
        "desc": view(0, np.int32, 20 * z[0]),
        "rows": view(1, np.uint32, z[7] * z[1]),
        "rowstart": view(2, np.int32, z[1]),
        "vert_gid": view(3, np.int32, z[2]),
        "row_ecodes": np.frombuffer(blob, dtype=np.uint32, count=((12 * z[6] + 31) // 32) * z[1], offset=z[15]),
        "tile_elems": np.frombuffer(blob, dtype=np.int32, count=z[19], offset=z[16]),
        "tile_tverts": np.frombuffer(blob, dtype=np.uint32, count=z[21], offset=z[20]),
        "long_rows": np.frombuffer(blob, dtype=np.uint32, count=24 * z[23], offset=z[22]),
        "elems_staged": bool(z[18]),
        # source-program launches: walk order of the tiles, positions per workgroup block, and per
        # owned row the local id its vertex has in the previous tile of its block (0xFFFF: none)
        "chain_order": np.frombuffer(blob, dtype=np.int32, count=z[0], offset=z[24]),
        "chain_len": z[25],
        "hand_in": np.frombuffer(blob, dtype=np.uint16, count=z[1], offset=z[26]),
        "max_n_tv": z[27],
        # z[28] > 0: that many runs of the chain order, one per resident workgroup (first positions + n_tiles)
        "n_runs": z[28],
        "runs": np.frombuffer(blob, dtype=np.int32, count=(z[28] + 1 if z[28] > 0 else 0), offset=z[30]),
    }


def p2_plan_host(conn_dof, n_verts, n_dofs, coords, rowptr, colind):
    """Build the P2 row plan on the host (tfem_p2_plan_*).  Raises NotImplementedError when
    the mesh does not fit it (DoF layout, vertices with more than 15 neighbours, numbering
    without locality).  Vertices with 8 .. 15 neighbours become long rows (plan["long_rows"])."""
    lib = _native.load()
    conn = np.ascontiguousarray(np.asarray(conn_dof).astype(np.int32)).reshape(-1, 6)
    coords = np.ascontiguousarray(np.asarray(coords, dtype=np.float64)).reshape(-1, 2)
    rowptr = np.ascontiguousarray(np.asarray(rowptr, dtype=np.int64))
    colind = np.ascontiguousarray(np.asarray(colind, dtype=np.int32))
    handle = c_void_p()
    _native.check(
        lib.tfem_p2_plan_create(
            c_void_p(conn.ctypes.data), conn.shape[0], int(n_verts), int(n_dofs),
            c_void_p(coords.ctypes.data), c_void_p(rowptr.ctypes.data),
            c_void_p(colind.ctypes.data), ctypes.byref(handle),
        )
    )
    try:
        layout = np.zeros(24, dtype=np.int64)
        _native.check(lib.tfem_p2_plan_sizes(handle, c_void_p(layout.ctypes.data)))
        blob = np.zeros(int(layout[16]), dtype=np.uint8)
        _native.check(lib.tfem_p2_plan_pack(handle, c_void_p(blob.ctypes.data)))
    finally:
        lib.tfem_p2_plan_destroy(handle)
    z = [int(x) for x in layout]
    view = lambda i, dtype, count: np.frombuffer(blob, dtype=dtype, count=count, offset=z[10 + i])  # noqa: E731
    return {
        "blob": blob,
        "layout": np.ascontiguousarray(layout, dtype=np.int64),
        "vertex": {"desc": view(0, np.int32, 16 * z[0]), "rows": view(1, np.uint32, 8 * z[2]),
                   "vert_gid": view(2, np.int32, z[7])},
        "edge": {"desc": view(3, np.int32, 16 * z[1]), "rows": view(4, np.uint32, 4 * z[3]),
                 "vert_gid": view(5, np.int32, z[8])},
        "long_rows": np.frombuffer(blob, dtype=np.uint32, count=32 * z[18], offset=z[17]),
        # load vector in row form: element * 4 + local index per fan slot / per triangle of an edge
        "vertex_codes": np.frombuffer(blob, dtype=np.uint32, count=8 * z[2], offset=z[19]),
        "edge_codes": np.frombuffer(blob, dtype=np.uint32, count=2 * z[3], offset=z[20]),
        "long_codes": np.frombuffer(blob, dtype=np.uint32, count=16 * z[18], offset=z[21]),
    }


class _LazyTriple:
    """(rowptr, colind, slots) whose third entry is built when somebody takes it."""

    def __init__(self, rowptr, colind, make_slots):
        self._head = (rowptr, colind)
        self._make, self._slots = make_slots, None

    def __getitem__(self, i):
        if i in (0, 1):
            return self._head[i]
        if i in (2, -1):
            if self._slots is None:
                self._slots = self._make()
            return self._slots
        raise IndexError(i)

    def __iter__(self):
        return iter((self._head[0], self._head[1], self[2]))

    def __len__(self):
        return 3


def _morton_permutation(coords):
    """Vertex ids along the Morton (Z-order) curve of their coordinates, on the tensor's own
    device: 21 bits per axis, interleaved; ties keep the caller's order (stable sort)."""
    xy = coords.detach().double()
    lo = xy.min(dim=0).values
    span = (xy.max(dim=0).values - lo).max().clamp_min(1e-300)
    q = ((xy - lo) / span * float((1 << 21) - 1)).to(torch.int64).clamp_(0, (1 << 21) - 1)

    def spread(v):  # abc -> 0a0b0c
        v = (v | (v << 16)) & 0x0000FFFF0000FFFF
        v = (v | (v << 8)) & 0x00FF00FF00FF00FF
        v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0F
        v = (v | (v << 2)) & 0x3333333333333333
        return (v | (v << 1)) & 0x5555555555555555

    code = spread(q[:, 0]) | (spread(q[:, 1]) << 1)
    return torch.sort(code, stable=True).indices


class AssemblyEngine:
    def __init__(self, coords, conn_geo, conn_dof, n_dofs, poly_order, quad_order, fracture=None):
        """coords (N_v,2) or (F,N_v,2); conn_geo (N_T,3) or (F,N_T,3) vertex ids (per mesh);
        conn_dof (E,n) global DoF ids; fracture = (pinv (F,2,3), det (F,1,1)) or None."""
        self.lib = _native.load()
        self.home = coords.device
        self.dtype = coords.dtype
        if self.dtype not in (torch.float64, torch.float32):
            raise TypeError(f"unsupported real dtype {self.dtype}")
        self.real_bytes = 8 if self.dtype == torch.float64 else 4
        self.poly_order = int(poly_order)
        self.quad_order = int(quad_order)
        self.n_quad = self.lib.tfem_quadrature_size(self.quad_order)
        if self.n_quad == 0:
            raise NotImplementedError("Integration order not implemented")
        self.n_dofs = int(n_dofs)
        self.lead_shape = tuple(conn_geo.shape[:-1])
        self.n_elems = int(np.prod(self.lead_shape)) if len(self.lead_shape) else 0
        self.n_local = int(conn_dof.shape[-1])
        self.n_fractures = int(coords.shape[0]) if coords.dim() == 3 else 0
        self.coords_per_mesh = int(coords.shape[-2])
        #: engine numbering -> caller's DoF (and back), or None: a P1 mesh whose vertex numbering has
        #: no locality (a mesh generator's output order) is renumbered along the Morton curve HERE,
        #: once -- every plan, kernel and array of the engine then lives in that numbering (the row
        #: kernels stream coordinates and CSR values contiguously), and the engine translates at its
        #: boundary: vectors go out / come in in the caller's numbering, operators carry `perm`
        self._perm = self._inv = None
        self._boundary_depth = 0  # > 0 inside a public method: nested calls stay in engine numbering
        self.kernel = os.environ.get("TFEM_KERNEL", "auto")
        if self._wants_renumbering(coords, conn_geo, conn_dof):
            perm = _morton_permutation(coords)
            inv = torch.empty_like(perm)
            inv[perm] = torch.arange(perm.numel(), device=perm.device)
            coords = coords[perm]
            conn_geo = inv[conn_geo.long()].to(conn_geo.dtype)
            conn_dof = conn_geo if conn_dof.shape == conn_geo.shape else inv[conn_dof.long()].to(conn_dof.dtype)
            self._perm, self._inv = perm, inv
        elif self._wants_edge_renumbering(coords, conn_geo, conn_dof):
            # P2 ("vertices, then edges"): the edge DoFs in the order of the mesh's edge list --
            # a generator's strips -- are renumbered along the Morton curve of the edge midpoints:
            # 256 consecutive edge rows then touch ~150 vertices instead of ~720, and the edge
            # tiles of the row plan gather a fifth of the coordinates
            n_v, n_e = self.coords_per_mesh, self.n_dofs - self.coords_per_mesh
            tri, edge = conn_geo.long(), conn_dof.long()[:, 3:] - n_v
            mid = torch.zeros((n_e, 2), dtype=coords.dtype, device=coords.device)
            for j in range(3):  # local edge j = (v_j, v_{j+1}) (element_tri.py:50-52)
                mid[edge[:, j]] = 0.5 * (coords[tri[:, j]] + coords[tri[:, (j + 1) % 3]])
            edge_perm = _morton_permutation(mid)
            edge_inv = torch.empty_like(edge_perm)
            edge_inv[edge_perm] = torch.arange(n_e, device=edge_perm.device)
            conn_dof = torch.cat([conn_dof[:, :3].long(), n_v + edge_inv[edge]], dim=1).to(conn_dof.dtype)
            ids = torch.arange(n_v, device=edge_perm.device)
            self._perm = torch.cat([ids, n_v + edge_perm])
            self._inv = torch.cat([ids, n_v + edge_inv])
        self._host_coords = coords
        self._host_conn_geo = conn_geo
        self._host_conn_dof = conn_dof.reshape(-1, self.n_local)
        self._host_fracture = fracture
        self._dev = None
        self._csr = None
        self._csr_host = None
        self._tiles = None
        self._rings = None
        self._priority_vertices = None
        self._gather = None
        self._slots_host = None
        self._p2rows = None
        self._pattern = None    # live CSR pattern handle between csr_structure() and ring_plan()
        self._conn_np = None    # DoF connectivity, int32 on the host (one copy for every builder)
        self._coords_np = None  # coordinates, float64 on the host
        self._same_conn = None
        self._edge_cells_checked = None  # (data_ptr, rows) of the last validated edge -> cells table
        #: "auto" (the best plan the mesh allows), "rings", "tiles", "rows" (P2 row kernels),
        #: "gather" (element blocks / vectors + gather, no plan) or "atomic" (one-pass scatter)
        #: (read at the top of the constructor)

    # ------------------------------------------------------------------ renumbering
    #: meshes below this many DoFs keep the caller's numbering (nothing to gain, and their CSR
    #: arrays stay directly comparable); TFEM_RENUMBER=0 / 1 switches the renumbering off / forces it
    RENUMBER_MIN_DOFS = int(os.environ.get("TFEM_RENUMBER_MIN", "50000"))

    def _wants_renumbering(self, coords, conn_geo, conn_dof):
        switch = os.environ.get("TFEM_RENUMBER", "")
        if switch == "0" or self.kernel != "auto" or self.poly_order != 1 or coords.dim() != 2 or conn_geo.dim() != 2:
            return False
        if conn_dof.shape != conn_geo.shape or self.n_elems == 0:
            return False
        a, b = conn_geo.reshape(-1), conn_dof.reshape(-1)
        if not (a.data_ptr() == b.data_ptr() or torch.equal(a.long(), b.to(a.device).long())):
            return False
        if switch == "1":
            return True
        if self.n_dofs < self.RENUMBER_MIN_DOFS:
            return False
        # locality of the numbering: the ids of two vertices of an element differ by about sqrt(N)
        # in a structured or curve-ordered numbering, by about N / 3 in a random one
        spread = (conn_geo[:, 0].long() - conn_geo[:, 1].long()).abs().double().median().item()
        return spread > 32.0 * (self.n_dofs ** 0.5)

    def _wants_edge_renumbering(self, coords, conn_geo, conn_dof):
        switch = os.environ.get("TFEM_RENUMBER", "")
        if switch == "0" or self.kernel != "auto" or self.poly_order != 2 or coords.dim() != 2 or conn_geo.dim() != 2:
            return False
        if conn_dof.dim() != 2 or conn_dof.shape[-1] != 6 or self.n_elems == 0 or self.n_dofs <= self.coords_per_mesh:
            return False
        if switch != "1" and self.n_dofs < self.RENUMBER_MIN_DOFS:
            return False
        # the layout the row plan needs: vertex DoF id = vertex id, then the edges
        head = conn_dof[:, :3].to(conn_geo.device)
        return bool(torch.equal(head.long(), conn_geo.long()) and int(conn_dof[:, 3:].min()) >= self.coords_per_mesh)

    @property
    def renumbered(self):
        return self._perm is not None

    def _dofs_out(self, vec):
        """A per-DoF vector of the engine -> the caller's numbering."""
        if self._perm is None or vec is None:
            return vec
        flat = vec.reshape(-1)
        return flat.index_select(0, self._inv.to(flat.device)).reshape(vec.shape)

    def _dofs_in(self, vec):
        """A per-DoF vector of the caller -> the engine's numbering."""
        if self._perm is None or vec is None:
            return vec
        flat = vec.reshape(-1)
        return flat.index_select(0, self._perm.to(flat.device)).reshape(vec.shape)

    # ------------------------------------------------------------------ device state
    @property
    def device(self):
        return _compute_device(self.home)

    def _inputs(self):
        if self._dev is None:
            dev = self.device
            d = {
                "coords": self._host_coords.to(dev).contiguous(),
                "conn_geo": self._host_conn_geo.to(dev, torch.int32).contiguous(),
                "conn_dof": self._host_conn_dof.to(dev, torch.int32).contiguous(),
                "pinv": None,
                "fdet": None,
            }
            if self._host_fracture is not None:
                pinv, det = self._host_fracture
                d["pinv"] = pinv.to(dev, self.dtype).contiguous()
                d["fdet"] = det.to(dev, self.dtype).reshape(-1).contiguous()
            self._dev = d
        return self._dev

    def _stream(self):
        return _native.current_stream(self.device)

    def _pairs(self, tensor):
        """Contiguous and aligned for the kernels' (x, y) pair accesses (a contiguous view can
        start anywhere inside its storage)."""
        tensor = tensor.contiguous()
        return tensor if tensor.data_ptr() % (2 * self.real_bytes) == 0 else tensor.clone()

    def _edge_inputs(self, edge_cells, points):
        if self.poly_order != 1 or self.n_fractures:
            raise NotImplementedError("edge interpolation kernel: P1 on one 2-D mesh")
        dev = self.device
        cells = edge_cells.to(dev, torch.int64).contiguous()
        points = self._pairs(points.detach().to(dev, self.dtype))
        n_edges, n_points = int(points.shape[0]), int(points.shape[1])
        if tuple(cells.shape) != (n_edges, 2) or points.dim() != 3 or points.shape[2] != 2:
            raise ValueError("edge interpolation: edge_cells (N_e, 2) and points (N_e, Q, 2) expected")
        # ids are checked once per table (two device syncs): the verdict is remembered for the
        # CALLER's tensor (identity and version), never for an address -- a freed device copy's
        # address can come back for another table
        key = (id(edge_cells), edge_cells._version, n_edges)
        if n_edges and key != self._edge_cells_checked:
            if int(cells.min()) < 0 or int(cells.max()) >= self.n_elems:
                raise IndexError("edge interpolation: cell id outside the mesh")
            self._edge_cells_checked = key
            self._edge_cells_keepalive = edge_cells  # the id stays this tensor's while we hold it
        return cells, points, n_edges, n_points

    def edge_interpolate(self, edge_cells, points, u, prepared=False):
        """P1 DoF vector u on both sides of the interior edges: one tfem_edge_interpolate_p1
        launch.  edge_cells (N_e, 2) cell ids, points (N_e, Q, 2); returns value (N_e, 2, Q)
        and gradient (N_e, 2, 2) on the compute device."""
        d = self._inputs()
        if prepared:  # the outputs of _edge_inputs, kept by the caller
            cells, n_edges, n_points = edge_cells, int(points.shape[0]), int(points.shape[1])
        else:
            cells, points, n_edges, n_points = self._edge_inputs(edge_cells, points)
        dev = self.device
        u = u.detach().to(dev, self.dtype).reshape(-1).contiguous()
        if u.numel() != self.coords_per_mesh:
            raise ValueError("edge interpolation: u must hold one value per vertex")
        value = torch.empty((n_edges, 2, n_points), dtype=self.dtype, device=dev)
        grad = torch.empty((n_edges, 2, 2), dtype=self.dtype, device=dev)
        with torch.cuda.device(dev):
            _native.check(
                self.lib.tfem_edge_interpolate_p1(
                    _native.ptr(d["coords"]), self.real_bytes, _native.ptr(d["conn_geo"]),
                    _native.ptr(cells), _native.ptr(points), n_edges, n_points, _native.ptr(u),
                    _native.ptr(value), _native.ptr(grad), self._stream(),
                )
            )
        return value, grad

    def edge_incidence(self, cells):
        """(inc_ptr, inc_side) of tfem_edge_interpolate_p1_backward_rows for the device table
        `cells` (N_e, 2): per vertex the entries 4 * side + local index, ascending.  Built with
        torch on the device, once per table (kept by the caller)."""
        conn = self._inputs()["conn_geo"].long()
        sides = cells.reshape(-1)
        verts = conn[sides]  # (2 N_e, 3)
        codes = (4 * torch.arange(sides.numel(), device=cells.device)[:, None] + torch.arange(3, device=cells.device)).reshape(-1)
        order = torch.argsort(verts.reshape(-1) * (4 * sides.numel() + 4) + codes)
        counts = torch.bincount(verts.reshape(-1), minlength=self.coords_per_mesh)
        ptr = torch.zeros(self.coords_per_mesh + 1, dtype=torch.int64, device=cells.device)
        ptr[1:] = torch.cumsum(counts, 0)
        return ptr.contiguous(), codes[order].contiguous()

    def edge_interpolate_backward(self, edge_cells, points, g_value, g_grad, prepared=False, incidence=None):
        """Adjoint of edge_interpolate in u: (N_v,) on the compute device from the cotangents
        g_value (N_e, 2, Q) and g_grad (N_e, 2, 2).  With the incidence table of edge_incidence:
        tfem_edge_interpolate_p1_backward_rows (no atomics, reproducible); without it
        tfem_edge_interpolate_p1_backward (hardware atomics)."""
        d = self._inputs()
        if prepared:
            cells, n_edges, n_points = edge_cells, int(points.shape[0]), int(points.shape[1])
        else:
            cells, points, n_edges, n_points = self._edge_inputs(edge_cells, points)
        dev = self.device
        g_value = g_value.detach().to(dev, self.dtype).reshape(n_edges, 2, n_points).contiguous()
        g_grad = self._pairs(g_grad.detach().to(dev, self.dtype).reshape(n_edges, 2, 2))
        grad_u = torch.empty(self.coords_per_mesh, dtype=self.dtype, device=dev)
        with torch.cuda.device(dev):
            if incidence is not None:
                _native.check(
                    self.lib.tfem_edge_interpolate_p1_backward_rows(
                        _native.ptr(d["coords"]), self.real_bytes, _native.ptr(d["conn_geo"]),
                        _native.ptr(cells), _native.ptr(points), n_edges, n_points, _native.ptr(g_value),
                        _native.ptr(g_grad), _native.ptr(incidence[0]), _native.ptr(incidence[1]),
                        _native.ptr(grad_u), self.coords_per_mesh, self._stream(),
                    )
                )
            else:
                _native.check(
                    self.lib.tfem_edge_interpolate_p1_backward(
                        _native.ptr(d["coords"]), self.real_bytes, _native.ptr(d["conn_geo"]),
                        _native.ptr(cells), _native.ptr(points), n_edges, n_points, _native.ptr(g_value),
                        _native.ptr(g_grad), _native.ptr(grad_u), self.coords_per_mesh, self._stream(),
                    )
                )
        return grad_u

    def edge_interpolate_fracture(self, coords3d, edge_cells, points, u):
        """FractureBasis.interpolate on the interior edges of every fracture: one
        tfem_edge_interpolate_p1_fracture launch.  coords3d (F, N_v, 3), edge_cells (F, N_e, 2),
        points (F, N_e, Q, 3); returns value (F, N_e, 2, Q) and grad (F, N_e, 2, 3)."""
        if self.poly_order != 1 or not self.n_fractures:
            raise NotImplementedError("fracture edge interpolation: P1 on a fracture mesh")
        d = self._inputs()
        dev = self.device
        f, n_v = self.n_fractures, self.coords_per_mesh
        n_cells = self.n_elems // f
        cells = edge_cells.to(dev, torch.int64).contiguous()
        n_edges = int(cells.shape[1])
        points = points.detach().to(dev, self.dtype).reshape(f, n_edges, -1, 3).contiguous()
        n_points = int(points.shape[2])
        x3 = coords3d.detach().to(dev, self.dtype).reshape(f, n_v, 3).contiguous()
        u = u.detach().to(dev, self.dtype).reshape(-1).contiguous()
        if tuple(cells.shape) != (f, n_edges, 2):
            raise ValueError("fracture edge interpolation: edge_cells (F, N_e, 2) expected")
        if u.numel() < n_v:
            raise IndexError("fracture edge interpolation: the vector is shorter than a fracture's vertex list")
        key = (id(edge_cells), edge_cells._version, n_edges)
        if n_edges and key != getattr(self, "_frac_cells_checked", None):
            if int(cells.min()) < 0 or int(cells.max()) >= n_cells:
                raise IndexError("fracture edge interpolation: cell id outside the fracture")
            self._frac_cells_checked, self._frac_cells_keepalive = key, edge_cells
        value = torch.empty((f, n_edges, 2, n_points), dtype=self.dtype, device=dev)
        grad = torch.empty((f, n_edges, 2, 3), dtype=self.dtype, device=dev)
        with torch.cuda.device(dev):
            _native.check(
                self.lib.tfem_edge_interpolate_p1_fracture(
                    _native.ptr(d["coords"]), _native.ptr(x3), self.real_bytes, _native.ptr(d["conn_geo"]),
                    _native.ptr(cells), _native.ptr(points), _native.ptr(d["pinv"]), f, n_v, n_cells,
                    n_edges, n_points, _native.ptr(u), u.numel(), _native.ptr(value), _native.ptr(grad),
                    self._stream(),
                )
            )
        return value, grad

    def _home(self, tensor):
        """Result on the caller's device.  A host-resident caller gets large results through a
        pinned staging copy (pageable device-to-host copies run at ~4 GB/s on this platform)."""
        if tensor.device == self.home:
            return tensor
        if self.home.type == "cpu" and tensor.is_cuda and tensor.numel() * tensor.element_size() >= (1 << 20):
            out = torch.empty(tensor.shape, dtype=tensor.dtype, device="cpu", pin_memory=True)
            out.copy_(tensor, non_blocking=True)
            torch.cuda.current_stream(tensor.device).synchronize()
            return out
        return tensor.to(self.home)

    # ------------------------------------------------------------------ symbolic phase
    def _conn_host_np(self):
        """The DoF connectivity as one contiguous int32 host array (narrowed where it lives:
        a third of the bytes of an int64 device tensor cross the bus)."""
        if self._conn_np is None:
            conn = self._host_conn_dof
            if conn.numel() and int(conn.max()) >= 2**31:
                raise ValueError("DoF ids beyond the int32 range of the kernels")
            self._conn_np = np.ascontiguousarray(conn.to(torch.int32).cpu().numpy())
        return self._conn_np

    def _coords_host_np(self):
        if self._coords_np is None:
            self._coords_np = np.ascontiguousarray(self._host_coords.detach().cpu().double().numpy())
        return self._coords_np

    def _geometry_is_dof_connectivity(self):
        """P1 on one mesh: the vertex connectivity IS the DoF connectivity (compared where the
        tensors live, once)."""
        if self._same_conn is None:
            a, b = self._host_conn_geo.reshape(-1), self._host_conn_dof.reshape(-1)
            if a.shape != b.shape:
                self._same_conn = False
            elif a.data_ptr() == b.data_ptr() and a.dtype == b.dtype:
                self._same_conn = True
            else:
                b = b.to(a.device)
                self._same_conn = bool(torch.equal(a, b) if a.dtype == b.dtype else torch.equal(a.long(), b.long()))
        return self._same_conn

    def csr_structure(self):
        """(rowptr int64, colind int32, slots) on the compute device.  `slots` (int32 (E,n,n), the
        CSR position of every element entry) is built on first use: only the scatter / gather
        paths read it."""
        if self._csr is None:
            keep = self.kernel in ("auto", "rings") and self._p1_plan_eligible()  # the ring plan starts from it
            rowptr, colind, *kept = pattern_host(self._conn_host_np(), self.n_dofs, keep=keep)
            self._pattern = kept[0] if kept else None
            self._csr_host = (rowptr, colind)
            dev = self.device
            self._csr = _LazyTriple(torch.from_numpy(rowptr).to(dev), torch.from_numpy(colind).to(dev), self._device_slots)
        return self._csr

    def _host_slots(self):
        if self._slots_host is None:
            self.csr_structure()
            rowptr, colind = self._csr_host
            self._slots_host = slots_host(self._conn_host_np(), self.n_dofs, rowptr, colind)
        return self._slots_host

    def _device_slots(self):
        return torch.from_numpy(self._host_slots()).to(self.device)

    def tile_plan(self):
        """Device copy of the tile plan, or None when this basis cannot use it (P2,
        fractures, float coordinates of another layout, rows longer than 16 entries)."""
        if self._tiles is None:
            self._tiles = False
            eligible = (
                self.kernel in ("auto", "tiles", "rings") and self.poly_order == 1
                and self.n_fractures == 0 and self._host_conn_geo.dim() == 2
                and self._geometry_is_dof_connectivity()
            )
            if eligible:
                self.csr_structure()
                rowptr, colind = self._csr_host
                try:
                    plan = tile_plan_host(
                        self._conn_host_np(), self.n_dofs, self._coords_host_np(), rowptr, colind,
                    )
                except NotImplementedError:
                    plan = None
                if plan is not None:
                    self._tiles = {
                        "blob": torch.from_numpy(plan["blob"]).to(self.device),
                        "layout": plan["layout"],
                        "sizes": [int(x) for x in plan["sizes"]],
                    }
            if self._tiles is False and self.kernel == "tiles":
                raise NotImplementedError("the tile-plan kernel does not apply to this basis")
        return self._tiles or None

    def _p1_plan_eligible(self):
        return (
            self.kernel != "atomic" and self.poly_order == 1 and self.n_fractures == 0
            and self._host_conn_geo.dim() == 2
            and self._geometry_is_dof_connectivity()
        )

    def set_priority_vertices(self, flags):
        if self._perm is not None:
            raise NotImplementedError("priority vertices (sharded runs) on an internally renumbered mesh: "
                                      "the shards of a partition keep the numbering's locality")
        return self._set_priority_vertices(flags)

    def _set_priority_vertices(self, flags):
        """Multi-GPU: flag the vertices shared with other ranks BEFORE the first
        assembly; the ring plan then lists the tiles owning them first, and
        assemble_system(..., tiles="priority" / "rest") launches the two tile ranges, so the
        exchange of the shared rows runs beside the launch over the rest."""
        if self._rings is not None:
            raise RuntimeError("set_priority_vertices: the ring plan is built already")
        flags = np.asarray(flags.cpu() if torch.is_tensor(flags) else flags).reshape(-1) != 0
        if flags.shape[0] != self.n_dofs:
            raise ValueError(f"set_priority_vertices: {flags.shape[0]} flags for {self.n_dofs} DoFs")
        self._priority_vertices = flags

    def tile_range(self, which):
        """(first, count) of the ring plan's tile list: "priority", "rest" or "all"."""
        rings = self.ring_plan()
        if rings is None:
            raise NotImplementedError("tile ranges need the ring plan")
        n, p = rings["n_tiles"], rings["n_priority"]
        if which != "all" and int(rings["layout"][23]) > 0:
            # the rows of vertices with 8 .. 15 neighbours (TFEM_RING_LONG=1 plans) are written by a
            # launch of their own over ALL of them: a part of the tiles would leave some unwritten
            raise NotImplementedError("tile ranges are not available for a ring plan with long rows")
        try:
            return {"priority": (0, p), "rest": (p, n - p), "all": (0, n)}[which]
        except KeyError:
            raise ValueError(f"tiles: {which!r} is not 'priority', 'rest' or 'all'") from None

    def _resident_source_workgroups(self):
        """Workgroups of a source-program ring launch that are resident at once on this device: four
        per CU (128 VGPRs, < 40 KB of LDS: csrc/tfem_rings.hip), on the CUs the launches may use."""
        cus = 256
        if self.device.type == "cuda":
            cus = int(torch.cuda.get_device_properties(self.device).multi_processor_count)
        try:
            reserve = max(0, int(os.environ.get("TFEM_RINGS_RESERVE_CUS", "0") or 0))
        except ValueError:
            reserve = 0
        return (max(8, cus - 8 * reserve) * 4 // 8) * 8

    def ring_plan(self):
        """Device copy of the ring plan (row form of the P1 stiffness/mass kernel), or None
        when this basis cannot use it (P2, fractures, fans that have no ring form)."""
        if self._rings is None:
            self._rings = False
            if self.kernel in ("auto", "rings") and self._p1_plan_eligible():
                self.csr_structure()
                rowptr, colind = self._csr_host
                # one run of the chain order per RESIDENT workgroup of a source-program launch: four per
                # CU the launches may use (TFEM_RINGS_RESERVE_CUS: CUs per XCD a sharded step leaves to
                # the interface exchange).  A launch with fewer workgroups than the plan has runs still
                # writes the same values, but some workgroups then walk two runs one after the other.
                wgs_given = os.environ.get("TFEM_RING_WGS")
                if wgs_given is None:
                    os.environ["TFEM_RING_WGS"] = str(self._resident_source_workgroups())
                try:
                    plan = ring_plan_host(
                        self._conn_host_np(), self.n_dofs, self._coords_host_np(), rowptr, colind,
                        priority=self._priority_vertices, pattern=self._pattern,
                    )
                except NotImplementedError:
                    plan = None
                finally:
                    if wgs_given is None:
                        os.environ.pop("TFEM_RING_WGS", None)
                    if self._pattern is not None:
                        self._pattern.release()
                        self._pattern = None
                if plan is not None:
                    # rows per output run (group of rows contiguous in the CSR array): the ring
                    # kernel streams a wave's rows out run by run, so a numbering without
                    # locality (about one run per row) is served better by the tile kernel
                    step = np.diff(plan["rowstart"].astype(np.int64))
                    n_runs = 1 + int(np.count_nonzero((step <= 0) | (step > 16)))
                    self._rings = {
                        "blob": torch.from_numpy(plan["blob"]).to(self.device),
                        "layout": plan["layout"],
                        "chunked": plan["chunked"],
                        "elems_staged": plan["elems_staged"],
                        "has_tverts": int(plan["layout"][21]) > 0 or int(plan["layout"][19]) == 0,
                        # long rows: their load-vector entries come from the element-form
                        # accumulation of the source-program launch only
                        "fq_ok": bool(plan["elems_staged"]) and int(plan["layout"][23]) == 0,
                        "rows_per_run": plan["rowstart"].size / n_runs,
                        "n_tiles": plan["n_tiles"],
                        "n_priority": plan["n_priority"],
                    }
            if self._rings is False and self.kernel == "rings":
                raise NotImplementedError("the ring-plan kernel does not apply to this basis")
        return self._rings or None

    def p2_plan(self):
        """Device copy of the P2 row plan, or None when this basis cannot use it."""
        if self._p2rows is None:
            self._p2rows = False
            if (self.kernel in ("auto", "rows") and self.poly_order == 2 and self.n_fractures == 0
                    and self._host_conn_geo.dim() == 2):
                self.csr_structure()
                rowptr, colind = self._csr_host
                try:
                    plan = p2_plan_host(
                        self._conn_host_np(), self.coords_per_mesh, self.n_dofs, self._coords_host_np(),
                        rowptr, colind,
                    )
                except NotImplementedError:
                    plan = None
                if plan is not None:
                    self._p2rows = {
                        "blob": torch.from_numpy(plan["blob"]).to(self.device),
                        "layout": plan["layout"],
                    }
            if self._p2rows is False and self.kernel == "rows":
                raise NotImplementedError("the P2 row kernels do not apply to this basis")
        return self._p2rows or None

    def _use_rings(self):
        """Ring kernel when its plan exists and the output runs are long enough (always with
        consecutive-vertex tiles), or when nothing else applies / it was asked for."""
        rings = self.ring_plan()
        if rings is None:
            return False
        if self.kernel == "rings" or rings["chunked"] or rings["rows_per_run"] >= 8.0:
            return True
        return self.tile_plan() is None

    def gather_map(self):
        """(gptr int64, gsrc int32) on the compute device: for every CSR entry the local-block
        entries that add to it, in the reference's accumulation order (tfem_csr_gather_map)."""
        if self._gather is None:
            self.csr_structure()
            nn = self.n_local * self.n_local
            nnz = int(self._csr_host[1].shape[0])
            slots = np.ascontiguousarray(self._host_slots(), dtype=np.int32)
            gptr = np.zeros(nnz + 1, dtype=np.int64)
            gsrc = np.zeros(max(slots.size, 1), dtype=np.int32)
            _native.check(self.lib.tfem_csr_gather_map(
                c_void_p(slots.ctypes.data), self.n_elems, nn, nnz,
                c_void_p(gptr.ctypes.data), c_void_p(gsrc.ctypes.data)))
            dev = self.device
            self._gather = (torch.from_numpy(gptr).to(dev), torch.from_numpy(gsrc[: slots.size]).to(dev))
        return self._gather

    def gather_map_linear(self):
        """The same for vectors: for every DoF the element-vector entries that add to it
        (tfem_csr_gather_map applied to the DoF connectivity)."""
        if getattr(self, "_gather_lin", None) is None:
            conn = self._conn_host_np()
            gptr = np.zeros(self.n_dofs + 1, dtype=np.int64)
            gsrc = np.zeros(max(conn.size, 1), dtype=np.int32)
            _native.check(self.lib.tfem_csr_gather_map(
                c_void_p(conn.ctypes.data), self.n_elems, self.n_local, self.n_dofs,
                c_void_p(gptr.ctypes.data), c_void_p(gsrc.ctypes.data)))
            dev = self.device
            self._gather_lin = (torch.from_numpy(gptr).to(dev), torch.from_numpy(gsrc[: conn.size]).to(dev))
        return self._gather_lin

    def _gather_local_vector(self, local):
        gptr, gsrc = self.gather_map_linear()
        out = torch.empty(self.n_dofs, dtype=self.dtype, device=self.device)
        with torch.cuda.device(self.device):
            _native.check(self.lib.tfem_csr_gather(
                _native.ptr(local), self.real_bytes, _native.ptr(gptr), _native.ptr(gsrc), self.n_dofs,
                _native.ptr(out), self._stream()))
        return out

    def _gather_local(self, local):
        """CSR values from entry-major local blocks (n*n, E): one tfem_csr_gather launch."""
        gptr, gsrc = self.gather_map()
        nnz = int(gptr.shape[0]) - 1
        vals = torch.empty(nnz, dtype=self.dtype, device=self.device)
        with torch.cuda.device(self.device):
            _native.check(self.lib.tfem_csr_gather(
                _native.ptr(local), self.real_bytes, _native.ptr(gptr), _native.ptr(gsrc), nnz,
                _native.ptr(vals), self._stream()))
        return vals

    def kernel_name(self):
        """Name of the dominant numeric kernel (the one that writes K) as rocprofv3 reports it."""
        if self.poly_order != 1:
            if self.p2_plan() is not None:
                return "k_p2_rows"
            return "k_p2_bilinear_atomic"  # element blocks (+ k_csr_gather unless TFEM_KERNEL=atomic)
        if self._use_rings():
            return "k_p1_rings"
        return "k_p1_tiles_pipe" if self.tile_plan() is not None else "k_p1_bilinear_atomic"

    def wrap_csr(self, vals):
        csr = self.csr_structure()
        perm = None if self._perm is None else self._perm.to(vals.device)
        return CSRMatrix(csr[0], csr[1], vals, (self.n_dofs, self.n_dofs), perm)

    def wrap_csr_home(self, vals):
        """The operator on the caller's device: the pattern is copied there once, the values
        per call (through the pinned staging copy of _home)."""
        if self.home == self.device:
            return self.wrap_csr(vals)
        if getattr(self, "_csr_home", None) is None:
            csr = self.csr_structure()
            self._csr_home = (csr[0].to(self.home), csr[1].to(self.home),
                              None if self._perm is None else self._perm.to(self.home))
        return CSRMatrix(self._csr_home[0], self._csr_home[1], self._home(vals), (self.n_dofs, self.n_dofs),
                         self._csr_home[2])

    # ------------------------------------------------------------------ kernels
    def geometry(self):
        """v_grad, dx, points, inv_jac in flat layouts (see tfem_tri_geometry)."""
        d = self._inputs()
        dev, e, q = self.device, self.n_elems, self.n_quad
        n = 3 if self.poly_order == 1 else 6
        vg_shape = (e, 3, 2) if self.poly_order == 1 else (e, q, 6, 2)
        v_grad = torch.empty(vg_shape, dtype=self.dtype, device=dev)
        dx = torch.empty((e, q), dtype=self.dtype, device=dev)
        points = torch.empty((e, q, 2), dtype=self.dtype, device=dev)
        inv = torch.empty((e, 2, 2), dtype=self.dtype, device=dev)
        with torch.cuda.device(dev):
            # per-mesh vertex ids: offset the coordinates per fracture on the host side of
            # the ABI by flattening (F, N_v, 2) and shifting ids once
            conn = d["conn_geo"]
            coords = d["coords"]
            if self.n_fractures:
                shift = (torch.arange(self.n_fractures, device=dev, dtype=torch.int32)
                         * self.coords_per_mesh)[:, None, None]
                conn = (conn + shift).contiguous()
            _native.check(
                self.lib.tfem_tri_geometry(
                    _native.ptr(coords), self.real_bytes, _native.ptr(conn), 4, e,
                    coords.numel() // 2, self.poly_order, self.quad_order,
                    _native.ptr(v_grad), _native.ptr(dx), _native.ptr(points), _native.ptr(inv),
                    self._stream(),
                )
            )
        return v_grad, dx, points, inv

    def bilinear(self, alpha: float, beta: float, out=None):
        """CSR values of alpha*stiffness + beta*mass (fused kernel).  ``out``: write into this
        preallocated device buffer of nnz entries (one device copy on the paths whose kernels
        allocate their own output)."""
        if out is not None and not self._use_rings():
            vals = self.bilinear(alpha, beta)
            self._output(out, vals.numel(), "CSR values").copy_(vals.view(-1))
            return out
        d = self._inputs()
        if self._use_rings():
            return self._assemble_rings(alpha, beta, out=(out, None))
        if self.tile_plan() is not None:
            return self._assemble_tiles(alpha, beta, want_matrix=True, fq=None)[0]
        csr = self.csr_structure()
        colind = csr[1]
        nnz = int(colind.shape[0])
        if self.p2_plan() is not None:
            rows = self.p2_plan()
            vals = torch.empty(nnz, dtype=self.dtype, device=self.device)
            with torch.cuda.device(self.device):
                _native.check(
                    self.lib.tfem_p2_assemble_rows(
                        _native.ptr(d["coords"]), self.real_bytes, self.quad_order, float(alpha),
                        float(beta), _native.ptr(rows["blob"]), c_void_p(rows["layout"].ctypes.data),
                        _native.ptr(vals), nnz, self._stream(),
                    )
                )
            return vals
        # P2, fractures, meshes without a plan: element blocks -> gather (no atomics, the
        # reference's accumulation order); TFEM_KERNEL=atomic keeps the one-pass atomic scatter
        two_pass = self.kernel != "atomic"
        nn = self.n_local * self.n_local
        out_len = nn * self.n_elems if two_pass else nnz
        out = torch.empty(out_len, dtype=self.dtype, device=self.device)
        with torch.cuda.device(self.device):
            _native.check(
                self.lib.tfem_tri_bilinear_csr(
                    _native.ptr(d["coords"]), self.real_bytes, _native.ptr(d["conn_geo"]), 4,
                    self.n_elems, self.coords_per_mesh, self.poly_order, self.quad_order,
                    float(alpha), float(beta), None if two_pass else _native.ptr(csr[2]),
                    _native.ptr(out), out_len,
                    _native.ptr(d["pinv"]), _native.ptr(d["fdet"]), self.n_fractures,
                    self.coords_per_mesh, self._stream(),
                )
            )
        return self._gather_local(out) if two_pass else out

    def _output(self, given, numel, what):
        """A caller-provided result buffer (pipelines that rotate preallocated buffers) or a
        fresh one; every kernel writes each entry exactly once, so nothing is cleared."""
        if given is None:
            return torch.empty(numel, dtype=self.dtype, device=self.device)
        if (given.dtype != self.dtype or given.device != self.device or given.numel() != numel
                or not given.is_contiguous()):
            raise ValueError(f"out: {what} must be a contiguous {self.dtype} tensor of {numel} "
                             f"entries on {self.device}")
        return given.view(-1)

    def _assemble_rings(self, alpha, beta, fq=None, want_matrix=True, out=(None, None), source=None,
                        tiles=None):
        """One tfem_p1_assemble_rings launch: CSR values of alpha*stiffness + beta*mass and,
        with source values fq (E, Q) or a source program (evaluated in the launch), the load
        vector (want_matrix=False: the vector alone).  tiles: (first, count) of the plan's tile
        list -- the launch writes the rows those tiles own and leaves the others untouched."""
        d = self._inputs()
        rings = self.ring_plan()
        nnz = int(self.csr_structure()[1].shape[0])
        # rows of vertices without elements are empty, every other entry is written once
        vals = self._output(out[0], nnz, "CSR values") if want_matrix else None
        fout = None
        with_load = fq is not None or source is not None
        if fq is not None:
            fq = fq.to(self.device, self.dtype).reshape(self.n_elems, self.n_quad).contiguous()
        if with_load:
            fout = self._output(out[1], self.n_dofs, "load vector")
        with torch.cuda.device(self.device):
            if tiles is not None:
                _native.check(
                    self.lib.tfem_p1_assemble_rings_range(
                        _native.ptr(d["coords"]), self.real_bytes, self.n_dofs, self.quad_order,
                        float(alpha), float(beta), _native.ptr(rings["blob"]),
                        c_void_p(rings["layout"].ctypes.data), _native.ptr(vals), nnz,
                        _native.ptr(fq), ctypes.byref(source) if source is not None else None,
                        self.n_elems, _native.ptr(fout), int(tiles[0]), int(tiles[1]), self._stream(),
                    )
                )
            elif source is not None:
                _native.check(
                    self.lib.tfem_p1_assemble_rings_source(
                        _native.ptr(d["coords"]), self.real_bytes, self.n_dofs, self.quad_order,
                        float(alpha), float(beta), _native.ptr(rings["blob"]),
                        c_void_p(rings["layout"].ctypes.data), _native.ptr(vals), nnz,
                        ctypes.byref(source), self.n_elems, _native.ptr(fout), self._stream(),
                    )
                )
            else:
                _native.check(
                    self.lib.tfem_p1_assemble_rings(
                        _native.ptr(d["coords"]), self.real_bytes, self.n_dofs, self.quad_order,
                        float(alpha), float(beta), _native.ptr(rings["blob"]),
                        c_void_p(rings["layout"].ctypes.data), _native.ptr(vals), nnz,
                        _native.ptr(fq), self.n_elems, _native.ptr(fout), self._stream(),
                    )
                )
        if not want_matrix:
            return fout
        return (vals, fout) if with_load else vals

    def prepared_system(self, alpha, beta, out, fq=None, source=None, tiles=None):
        """The launch of assemble_system(alpha, beta, fq | source, out=out, tiles=tiles) with every
        argument converted ONCE: the returned callable only enqueues on the current stream
        (launch-bound callers: the steps of a sharded run, small meshes).  It holds references to
        ``out``, ``fq`` and the plan; the source program is copied."""
        if (fq is None) == (source is None):
            raise ValueError("prepared_system: source values fq OR a source program")
        rings = self.ring_plan() if self._use_rings() else None
        if rings is None or (source is not None and not self._rings_take_source()) or (source is None and not rings["fq_ok"]):
            raise NotImplementedError("prepared launches need the ring plan of this basis")
        d = self._inputs()
        nnz = int(self.csr_structure()[1].shape[0])
        vals = self._output(out[0], nnz, "CSR values")
        fout = self._output(out[1], self.n_dofs, "load vector")
        first, count = self.tile_range(tiles) if tiles is not None else (0, -1)
        if fq is not None:
            fq = fq.to(self.device, self.dtype).reshape(self.n_elems, self.n_quad).contiguous()
        program = None
        if source is not None:
            program = _native.SourceProgram()
            ctypes.memmove(ctypes.byref(program), ctypes.byref(source), ctypes.sizeof(program))
        fixed = (_native.ptr(d["coords"]), self.real_bytes, self.n_dofs, self.quad_order, float(alpha), float(beta),
                 _native.ptr(rings["blob"]), c_void_p(rings["layout"].ctypes.data), _native.ptr(vals), nnz,
                 _native.ptr(fq), ctypes.byref(program) if program is not None else None, self.n_elems,
                 _native.ptr(fout), int(first), int(count))
        keep = (d, rings, vals, fout, fq, program)  # what the raw pointers above point into
        launch_fn, device, check = self.lib.tfem_p1_assemble_rings_range, self.device, _native.check
        current_stream = torch.cuda.current_stream

        def launch(stream=None):
            """Enqueue on `stream` (a torch.cuda.Stream; default: the current one)."""
            handle = (stream if stream is not None else current_stream(device)).cuda_stream
            status = launch_fn(*fixed, c_void_p(handle))
            if status:
                check(status)
            return keep[2], keep[3]

        return launch

    # ------------------------------------------------------------------ source programs
    def supports_source(self):
        """Source programs f(x, y) apply to one 2-D mesh (fracture points are 3-D)."""
        return self.n_fractures == 0 and self._host_coords.dim() == 2

    def _rings_take_source(self):
        if self.kernel == "tiles" or not self._use_rings():
            return False
        # TFEM_DETERMINISTIC=1: load vectors bit for bit the same from launch to launch.  The launch
        # that evaluates the source itself sums a vertex's element shares in LDS in the order the waves
        # reach them (reproducible to rounding, <= 1e-15 relative); with this switch the source goes
        # to memory first (tfem_source_eval) and the row-form launch that reads source values sums
        # every row's shares in fan order -- two launches, ~0.30 instead of 0.16 ms at 1e7 elements.
        if os.environ.get("TFEM_DETERMINISTIC", "0") not in ("", "0"):
            return False
        rings = self.ring_plan()
        return bool(rings["elems_staged"] and rings["has_tverts"])

    def source_values(self, program):
        """fq (E, Q) = the program at the integration points: one tfem_source_eval launch (for
        the kernels that take pre-evaluated source values)."""
        d = self._inputs()
        fq = torch.empty((self.n_elems, self.n_quad), dtype=self.dtype, device=self.device)
        with torch.cuda.device(self.device):
            _native.check(
                self.lib.tfem_source_eval(
                    _native.ptr(d["coords"]), self.real_bytes, _native.ptr(d["conn_geo"]), 4,
                    self.n_elems, self.coords_per_mesh, self.quad_order, ctypes.byref(program),
                    _native.ptr(fq), self._stream(),
                )
            )
        return fq

    # ------------------------------------------------------------------ VPINN residual form
    def supports_residual(self):
        """The fused residual form f v + s grad v . g: P1 on one 2-D mesh, DoFs = vertices."""
        return self.poly_order == 1 and self.supports_source() and self._host_conn_geo.dim() == 2

    def _flat_flux(self, flux):
        """(E, Q, 2) contiguous from a tensor broadcastable to (..., Q, 1, 2), or None."""
        want = tuple(self.lead_shape) + (self.n_quad, 1, 2)
        try:
            if tuple(torch.broadcast_shapes(tuple(flux.shape), want)) != want:
                return None
        except RuntimeError:
            return None
        return flux.detach().to(self.device, self.dtype).expand(want).reshape(self.n_elems, self.n_quad, 2).contiguous()

    def residual(self, fq, program, flux, flux_sign):
        """(N_dof,) vector of sum_q dx (f v + flux_sign grad v . g): one tfem_p1_residual_local
        launch + the gather.  fq (E, Q) or program or neither; flux (E, Q, 2) or None."""
        d = self._inputs()
        local = torch.empty(3 * self.n_elems, dtype=self.dtype, device=self.device)
        with torch.cuda.device(self.device):
            _native.check(
                self.lib.tfem_p1_residual_local(
                    _native.ptr(d["coords"]), self.real_bytes, _native.ptr(d["conn_geo"]), 4,
                    self.n_elems, self.coords_per_mesh, self.quad_order, _native.ptr(fq),
                    ctypes.byref(program) if program is not None else None, _native.ptr(flux),
                    float(flux_sign), _native.ptr(local), self._stream(),
                )
            )
        return self._gather_local_vector(local)

    def residual_backward(self, cotangent, flux_sign, want_fq, want_flux):
        """Cotangents of the source values (E, Q) and of the flux (E, Q, 2) from the cotangent of
        the residual vector: one tfem_p1_residual_backward launch."""
        d = self._inputs()
        cot = cotangent.detach().to(self.device, self.dtype).reshape(-1).contiguous()
        grad_fq = torch.empty((self.n_elems, self.n_quad), dtype=self.dtype, device=self.device) if want_fq else None
        grad_flux = torch.empty((self.n_elems, self.n_quad, 2), dtype=self.dtype, device=self.device) if want_flux else None
        with torch.cuda.device(self.device):
            _native.check(
                self.lib.tfem_p1_residual_backward(
                    _native.ptr(d["coords"]), self.real_bytes, _native.ptr(d["conn_geo"]), 4,
                    self.n_elems, self.coords_per_mesh, self.quad_order, _native.ptr(cot),
                    float(flux_sign), _native.ptr(grad_fq), _native.ptr(grad_flux), self._stream(),
                )
            )
        return grad_fq, grad_flux

    def load_source(self, program, out=None):
        """(N_dof,) load vector of the source program: evaluated inside the ring launch where
        the ring plan applies, else tfem_source_eval + the kernels that read source values."""
        if self._rings_take_source():
            return self._assemble_rings(0.0, 0.0, want_matrix=False, out=(None, out), source=program)
        f = self.load(self.source_values(program))
        if out is not None:
            self._output(out, f.numel(), "load vector").copy_(f.view(-1))
            return out
        return f

    def _assemble_tiles(self, alpha, beta, want_matrix, fq):
        """One tfem_p1_assemble_tiles launch: CSR values and/or the load vector."""
        d = self._inputs()
        tiles = self.tile_plan()
        vals = fout = None
        if want_matrix:
            vals = torch.empty(int(self.csr_structure()[1].shape[0]), dtype=self.dtype,
                               device=self.device)
        if fq is not None:
            fq = fq.to(self.device, self.dtype).reshape(self.n_elems, self.n_quad).contiguous()
            fout = torch.empty(self.n_dofs, dtype=self.dtype, device=self.device)
        nnz = int(vals.shape[0]) if vals is not None else 0
        with torch.cuda.device(self.device):
            _native.check(
                self.lib.tfem_p1_assemble_tiles(
                    _native.ptr(d["coords"]), self.real_bytes, self.n_dofs, self.quad_order,
                    float(alpha), float(beta), _native.ptr(tiles["blob"]),
                    c_void_p(tiles["layout"].ctypes.data), _native.ptr(vals), nnz,
                    _native.ptr(fq), self.n_elems, _native.ptr(fout), self._stream(),
                )
            )
        return vals, fout

    def assemble_system(self, alpha, beta, fq=None, out=None, source=None, tiles=None):
        """CSR values of alpha*stiffness + beta*mass AND the load vector of the source
        values fq (E, Q) or of the source program `source`: one fused launch on the ring and
        tile paths, two launches otherwise.  ``out=(vals, f)``: write into these preallocated
        device buffers.  ``tiles`` ("priority" / "rest" / "all", see set_priority_vertices):
        only the rows owned by that range of the ring plan's tiles are written (into ``out``,
        which is then required), the other entries are left as they are."""
        if (fq is None) == (source is None):
            raise ValueError("assemble_system: source values fq OR a source program")
        if tiles is not None:
            if out is None:
                raise ValueError("assemble_system(tiles=...): the launch writes part of the rows, pass out=(vals, f)")
            span = self.tile_range(tiles)
            if source is not None and not self._rings_take_source():
                raise NotImplementedError("tile ranges with a source program need the ring plan's element table")
            if source is None and not self.ring_plan()["fq_ok"]:
                raise NotImplementedError("tile ranges with source values need the staged element lists")
            return self._assemble_rings(alpha, beta, fq, out=out, source=source, tiles=span)
        if source is not None:
            if self._rings_take_source():
                return self._assemble_rings(alpha, beta, out=out or (None, None), source=source)
            fq = self.source_values(source)
        if self._use_rings() and self.ring_plan()["fq_ok"]:
            return self._assemble_rings(alpha, beta, fq, out=out or (None, None))
        if self.tile_plan() is not None:
            vals, f = self._assemble_tiles(alpha, beta, want_matrix=True, fq=fq)
        else:
            vals, f = self.bilinear(alpha, beta), self.load(fq)
        if out is not None:  # kernels without an output argument: one device copy each
            self._output(out[0], vals.numel(), "CSR values").copy_(vals.view(-1))
            self._output(out[1], f.numel(), "load vector").copy_(f.view(-1))
            return out[0], out[1]
        return vals, f

    def load(self, fq):
        """(N_dof,) vector of sum_q f_q phi_i dx_q; fq is (E, Q) on any device."""
        # the vector alone: row form with the source values staged per tile (142 us at 1e7
        # elements) when the ring plan applies, the element-form tile kernel (149 us) otherwise
        if self.kernel != "tiles" and self._use_rings() and self.ring_plan()["fq_ok"]:
            return self._assemble_rings(0.0, 0.0, fq, want_matrix=False)
        if self.tile_plan() is not None:
            return self._assemble_tiles(0.0, 0.0, want_matrix=False, fq=fq)[1]
        d = self._inputs()
        fq = fq.to(self.device, self.dtype).reshape(self.n_elems, self.n_quad).contiguous()
        if self.p2_plan() is not None and os.environ.get("TFEM_P2_LOAD", "rows") == "rows":
            # P2 with a row plan: one lane per DoF over the tiles of the stiffness launch
            rows = self.p2_plan()
            out = torch.empty(self.n_dofs, dtype=self.dtype, device=self.device)
            with torch.cuda.device(self.device):
                _native.check(
                    self.lib.tfem_p2_load_rows(
                        _native.ptr(d["coords"]), self.real_bytes, self.quad_order, _native.ptr(rows["blob"]),
                        c_void_p(rows["layout"].ctypes.data), _native.ptr(fq), self.n_elems, _native.ptr(out),
                        self.n_dofs, self._stream(),
                    )
                )
            return out
        # P2, fractures, meshes without a plan: element vectors -> gather (no atomics, the
        # reference's accumulation order); TFEM_KERNEL=atomic keeps the one-pass atomic scatter
        two_pass = self.kernel != "atomic"
        out_len = self.n_local * self.n_elems if two_pass else self.n_dofs
        out = torch.empty(out_len, dtype=self.dtype, device=self.device)
        with torch.cuda.device(self.device):
            _native.check(
                self.lib.tfem_tri_load_vector(
                    _native.ptr(d["coords"]), self.real_bytes, _native.ptr(d["conn_geo"]),
                    None if two_pass else _native.ptr(d["conn_dof"]), 4, self.n_elems,
                    self.coords_per_mesh, self.poly_order, self.quad_order, _native.ptr(fq),
                    _native.ptr(out), out_len, _native.ptr(d["fdet"]), self.n_fractures,
                    self.coords_per_mesh, self._stream(),
                )
            )
        return self._gather_local_vector(out) if two_pass else out

    def _flatten_integrand(self, integrand, dx_shape, inner):
        """Broadcast against dx (..., Q, 1, 1) and view as (E, Q, *inner) with the inner
        block contiguous; returns (tensor, element stride, quadrature stride)."""
        integrand = integrand.to(self.device, self.dtype)
        full = torch.broadcast_shapes(tuple(integrand.shape), tuple(dx_shape))
        if tuple(full[-2:]) != tuple(inner) or full[-3] != self.n_quad:
            raise ValueError(
                f"integrand broadcasts to {tuple(full)}, expected (..., {self.n_quad}, {inner[0]}, {inner[1]})"
            )
        expanded = integrand.expand(full)
        target = (self.n_elems, self.n_quad) + tuple(inner)
        try:
            flat = expanded.view(target)
        except RuntimeError:
            flat = expanded.contiguous().view(target)
        if flat.stride(-1) != 1 or (inner[0] > 1 and flat.stride(-2) != inner[1]):
            flat = flat.contiguous()
        return flat, flat.stride(0), flat.stride(1)

    def reduce_bilinear(self, integrand, dx):
        n = self.n_local
        flat, es, qs = self._flatten_integrand(integrand, dx.shape, (n, n))
        dxf = dx.to(self.device, self.dtype).reshape(self.n_elems, self.n_quad).contiguous()
        csr = self.csr_structure()
        nnz = int(csr[1].shape[0])
        two_pass = self.kernel != "atomic"
        out_len = n * n * self.n_elems if two_pass else nnz
        out = torch.empty(out_len, dtype=self.dtype, device=self.device)
        with torch.cuda.device(self.device):
            _native.check(
                self.lib.tfem_reduce_scatter_bilinear(
                    _native.ptr(flat), self.real_bytes, es, qs, _native.ptr(dxf), self.n_elems,
                    self.n_quad, n, None if two_pass else _native.ptr(csr[2]), _native.ptr(out),
                    out_len, self._stream(),
                )
            )
        return self._gather_local(out) if two_pass else out

    def reduce_linear(self, integrand, dx):
        n = self.n_local
        d = self._inputs()
        flat, es, qs = self._flatten_integrand(integrand, dx.shape, (n, 1))
        dxf = dx.to(self.device, self.dtype).reshape(self.n_elems, self.n_quad).contiguous()
        two_pass = self.kernel != "atomic"
        out_len = n * self.n_elems if two_pass else self.n_dofs
        out = torch.empty(out_len, dtype=self.dtype, device=self.device)
        with torch.cuda.device(self.device):
            _native.check(
                self.lib.tfem_reduce_scatter_linear(
                    _native.ptr(flat), self.real_bytes, es, qs, _native.ptr(dxf), self.n_elems,
                    self.n_quad, n, None if two_pass else _native.ptr(d["conn_dof"]), 4,
                    _native.ptr(out), out_len, self._stream(),
                )
            )
        return self._gather_local_vector(out) if two_pass else out

    def reduce_functional(self, integrand, dx):
        integrand = integrand.to(self.device, self.dtype)
        full = torch.broadcast_shapes(tuple(integrand.shape), tuple(dx.shape))
        if full[-1] != 1:
            raise NotImplementedError("functional integrands with a trailing dimension > 1")
        n_inner = int(full[-2])
        flat, es, qs = self._flatten_integrand(integrand, dx.shape, (n_inner, 1))
        dxf = dx.to(self.device, self.dtype).reshape(self.n_elems, self.n_quad).contiguous()
        out = torch.empty(self.n_elems, dtype=self.dtype, device=self.device)
        with torch.cuda.device(self.device):
            _native.check(
                self.lib.tfem_reduce_functional(
                    _native.ptr(flat), self.real_bytes, es, qs, _native.ptr(dxf), self.n_elems,
                    self.n_quad, n_inner, _native.ptr(out), self._stream(),
                )
            )
        return out.reshape(tuple(full[:-3]) + (1,))


# ---------------------------------------------------------------------------------------------
# An internally renumbered engine (AssemblyEngine._perm) translates per-DoF vectors at its
# boundary; everything inside -- plans, kernels, gather maps -- works in the engine's numbering.
# ---------------------------------------------------------------------------------------------
def _translate_vector_result(name):
    inner = getattr(AssemblyEngine, name)

    def method(self, *args, **kwargs):
        if self._perm is None or self._boundary_depth:
            return inner(self, *args, **kwargs)
        given = kwargs.pop("out", None)
        self._boundary_depth += 1
        try:
            result = inner(self, *args, **kwargs)
        finally:
            self._boundary_depth -= 1
        result = self._dofs_out(result)
        if given is not None:
            self._output(given, result.numel(), "load vector").copy_(result.view(-1))
            return given
        return result

    method.__name__, method.__doc__ = name, inner.__doc__
    setattr(AssemblyEngine, name, method)


def _translate_vector_argument(name, position, keyword):
    inner = getattr(AssemblyEngine, name)

    def method(self, *args, **kwargs):
        if self._perm is not None and not self._boundary_depth:
            if keyword in kwargs:
                kwargs[keyword] = self._dofs_in(kwargs[keyword])
            elif len(args) > position:
                args = args[:position] + (self._dofs_in(args[position]),) + args[position + 1:]
        return inner(self, *args, **kwargs)

    method.__name__, method.__doc__ = name, inner.__doc__
    setattr(AssemblyEngine, name, method)


def _translate_system():
    inner = AssemblyEngine.assemble_system

    def assemble_system(self, alpha, beta, fq=None, out=None, source=None, tiles=None):
        if self._perm is None or self._boundary_depth:
            return inner(self, alpha, beta, fq, out, source, tiles)
        if tiles is not None:
            raise NotImplementedError("tile ranges (sharded runs) on an internally renumbered mesh")
        self._boundary_depth += 1
        try:
            vals, f = inner(self, alpha, beta, fq, None if out is None else (out[0], None), source, None)
        finally:
            self._boundary_depth -= 1
        f = self._dofs_out(f)
        if out is not None and out[1] is not None:
            self._output(out[1], f.numel(), "load vector").copy_(f.view(-1))
            f = out[1]
        return vals, f

    assemble_system.__doc__ = inner.__doc__
    AssemblyEngine.assemble_system = assemble_system

    prepared = AssemblyEngine.prepared_system

    def prepared_system(self, *args, **kwargs):
        if self._perm is not None:
            raise NotImplementedError("prepared launches write the engine's own numbering: not on an internally "
                                      "renumbered mesh (TFEM_RENUMBER=0 keeps the caller's numbering)")
        return prepared(self, *args, **kwargs)

    prepared_system.__doc__ = prepared.__doc__
    AssemblyEngine.prepared_system = prepared_system


for _name in ("load", "load_source", "reduce_linear", "residual", "edge_interpolate_backward"):
    _translate_vector_result(_name)
_translate_vector_argument("edge_interpolate", 2, "u")
_translate_vector_argument("residual_backward", 0, "cotangent")
_translate_system()
