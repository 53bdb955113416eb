"""CPU-only tests of the host side: C ABI surface, symbolic phase, tile plan, form tracer,
mesh topology against the reference-generated fixtures.  No GPU compute is called."""

import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import REPO, load_golden, mesh_from_golden, scaled_error
from oracle import assembly_oracle as orc
from plan_emulator import run_plan
import ring_emulator
from ring_emulator import run_ring_plan
from p2rows_emulator import run_p2_load_plan, run_p2_plan


#: 2 pi^2 sin(pi x) sin(pi y) as the tracer compiles it (tests/test_assembly.py:75-77)
_SIN_SIN_PROGRAM = (
    [orc.SRC_PUSH_X, orc.SRC_SIN, orc.SRC_PUSH_Y, orc.SRC_SIN, orc.SRC_MUL],
    [np.pi, 2.0 * np.pi**2, np.pi, 1.0, 0.0],
)


@pytest.fixture(autouse=True)
def _cpu_defaults():
    torch.set_default_device("cpu")
    torch.set_default_dtype(torch.float64)
    yield
    torch.set_default_dtype(torch.float32)


def test_library_loads_and_exports_every_declared_symbol():
    from pytorch_fem_solver_amd import _native

    lib = _native.load()
    header = open(os.path.join(REPO, "include", "tfem_assembly.h")).read()
    declared = set(re.findall(r"^(?:int|void|const char \*)\s*(tfem_[a-z0-9_]+)\(", header, re.M))
    assert len(declared) >= 20
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in tfem_assembly.h but not exported"
        assert name in _native.SIGNATURES, f"{name} has no ctypes signature"
    assert lib.tfem_abi_version() == _native.ABI_VERSION
    version = int(re.search(r"#define TFEM_ABI_VERSION (\d+)", header).group(1))
    assert version == _native.ABI_VERSION
    assert lib.tfem_device_count() >= 0
    assert [lib.tfem_quadrature_size(q) for q in (1, 2, 3, 4, 5)] == [1, 3, 4, 6, 0]


def test_error_reporting_maps_to_reference_exceptions():
    from pytorch_fem_solver_amd import _native

    lib = _native.load()
    nodes = (ctypes.c_double * 12)()
    weights = (ctypes.c_double * 6)()
    status = lib.tfem_quadrature_rule(7, nodes, weights)
    assert status == 2
    with pytest.raises(NotImplementedError, match="Integration order not implemented"):
        _native.check(status)
    rowptr = np.zeros(4, dtype=np.int64)
    nnz = ctypes.c_int64()
    bad = np.array([[0, 1, 5]], dtype=np.int32)  # vertex 5 out of range
    status = lib.tfem_csr_symbolic_count(bad.ctypes.data, 4, 1, 3, 3, rowptr.ctypes.data, ctypes.byref(nnz))
    with pytest.raises(ValueError, match="outside"):
        _native.check(status)


@pytest.mark.parametrize("kind", ["structured", "delaunay", "p2"])
def test_symbolic_phase_against_oracle(kind):
    from pytorch_fem_solver_amd import dofs, meshgen
    from pytorch_fem_solver_amd.basis.engine import symbolic_host

    mesh = meshgen.unit_square(12, 0.25, 1) if kind != "delaunay" else meshgen.delaunay_square(400, 3)
    conn = mesh["triangles"]
    n = mesh["vertices"].shape[0]
    if kind == "p2":
        conn, xy, _ = dofs.p2_dofs_numpy(mesh["vertices"], mesh["triangles"], mesh["edges"],
                                         mesh["edge_markers"], mesh["vertex_markers"])
        n = xy.shape[0]
    rowptr, colind, slots = symbolic_host(conn, n)
    r2, c2, s2 = orc.csr_pattern(conn, n)
    assert np.array_equal(rowptr, r2) and np.array_equal(colind, c2)
    assert np.array_equal(slots, s2.reshape(-1))


@pytest.mark.parametrize("kind", ["p1", "p2"])
def test_gather_map_lists_every_contribution_in_reference_order(kind):
    """tfem_csr_gather_map: every local-block entry appears exactly once, under the CSR entry
    its slot names, and the contributions of an entry are in ascending (element, local entry)
    order -- the order of a sequential index_put_(accumulate=True)."""
    import ctypes

    from pytorch_fem_solver_amd import _native, dofs, meshgen
    from pytorch_fem_solver_amd.basis.engine import symbolic_host

    mesh = meshgen.delaunay_square(300, 7)
    conn, n = mesh["triangles"], mesh["vertices"].shape[0]
    if kind == "p2":
        conn, xy, _ = dofs.p2_dofs_numpy(mesh["vertices"], mesh["triangles"], mesh["edges"],
                                         mesh["edge_markers"], mesh["vertex_markers"])
        n = xy.shape[0]
    rowptr, colind, slots = symbolic_host(conn, n)
    ne, nl = conn.shape
    nn, nnz = nl * nl, colind.shape[0]
    gptr = np.zeros(nnz + 1, dtype=np.int64)
    gsrc = np.zeros(ne * nn, dtype=np.int32)
    lib = _native.load()
    _native.check(lib.tfem_csr_gather_map(ctypes.c_void_p(slots.ctypes.data), ne, nn, nnz,
                                          ctypes.c_void_p(gptr.ctypes.data), ctypes.c_void_p(gsrc.ctypes.data)))
    assert gptr[0] == 0 and gptr[-1] == ne * nn and np.all(np.diff(gptr) >= 1)
    assert np.array_equal(np.sort(gsrc), np.arange(ne * nn))
    k, e = np.divmod(gsrc.astype(np.int64), ne)  # entry-major local index = k * E + e
    owner = np.repeat(np.arange(nnz), np.diff(gptr))
    assert np.array_equal(slots.reshape(ne, nn)[e, k], owner)
    flat = e * nn + k
    same = owner[1:] == owner[:-1]
    assert np.all(flat[1:][same] > flat[:-1][same])
    # the sums through the map equal the oracle's scatter
    rng = np.random.default_rng(0)
    local = rng.standard_normal((ne, nl, nl))
    want = orc.assemble_csr_values(local, slots.reshape(ne, nl, nl), nnz)
    got = np.add.reduceat(local.reshape(ne, nn).T.reshape(-1)[gsrc], gptr[:-1])
    assert scaled_error(got, want) <= 1e-14


def test_pattern_handle_exports_the_same_columns_twice():
    """tfem_csr_pattern_export hands over the columns the handle kept from measuring the rows; a
    second export forms them again from the incidence: same arrays."""
    import ctypes
    from ctypes import c_void_p

    from pytorch_fem_solver_amd import _native, meshgen

    lib = _native.load()
    mesh = meshgen.delaunay_square(3000, 4)
    conn = np.ascontiguousarray(mesh["triangles"].astype(np.int32))
    nv = mesh["vertices"].shape[0]
    handle, nnz = c_void_p(), ctypes.c_int64(0)
    _native.check(lib.tfem_csr_pattern_create(c_void_p(conn.ctypes.data), 4, conn.shape[0], 3, nv,
                                              ctypes.byref(handle), ctypes.byref(nnz)))
    try:
        out = []
        for _ in range(2):
            rowptr = np.empty(nv + 1, dtype=np.int64)
            colind = np.full(nnz.value, -7, dtype=np.int32)
            _native.check(lib.tfem_csr_pattern_export(handle, c_void_p(rowptr.ctypes.data), c_void_p(colind.ctypes.data)))
            out.append((rowptr, colind))
    finally:
        lib.tfem_csr_pattern_destroy(handle)
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    want_rowptr, want_colind, _ = orc.csr_pattern(mesh["triangles"], nv)
    assert np.array_equal(out[0][0], want_rowptr) and np.array_equal(out[0][1], want_colind)


def test_symbolic_phase_empty_mesh():
    from pytorch_fem_solver_amd.basis.engine import symbolic_host

    rowptr, colind, slots = symbolic_host(np.zeros((0, 3), dtype=np.int32), 5)
    assert rowptr.tolist() == [0] * 6 and colind.size == 0 and slots.size == 0


def _stiffness_blocks(xy):
    geo = orc.geometry(xy, 1, 3)
    return orc.integrate_local(orc.integrand_stiffness(geo), geo["dx"])


@pytest.mark.parametrize("kind", ["structured", "delaunay", "delaunay_shuffled", "tiny", "isolated_vertex"])
def test_tile_plan_is_a_valid_exact_cover(kind):
    """Walk the plan like the kernel does (tests/plan_emulator.py): every CSR entry is
    written exactly once and the sums equal the oracle's."""
    from pytorch_fem_solver_amd import meshgen
    from pytorch_fem_solver_amd.basis.engine import symbolic_host, tile_plan_host

    if kind == "structured":
        mesh = meshgen.unit_square(40, 0.25, 0)
    elif kind == "tiny":
        mesh = meshgen.unit_square(1, 0.0, 0)
    else:
        mesh = meshgen.delaunay_square(3000, 2)
        if kind == "delaunay_shuffled":
            perm = np.random.default_rng(0).permutation(mesh["vertices"].shape[0])
            mesh = meshgen.permute_mesh(mesh, vertex_order=perm)
        if kind == "isolated_vertex":
            mesh["vertices"] = np.concatenate([mesh["vertices"], [[0.5, 0.5]]])
            mesh["vertex_markers"] = np.concatenate([mesh["vertex_markers"], [[0]]]).astype(np.int32)
    nv = mesh["vertices"].shape[0]
    rowptr, colind, slots = symbolic_host(mesh["triangles"], nv)
    plan = tile_plan_host(mesh["triangles"], nv, mesh["vertices"], rowptr, colind)
    sizes = plan["sizes"]
    assert sizes[5] <= 1024 and sizes[6] <= 1024 and sizes[7] <= 512 and sizes[8] <= 4096
    vals, writes = run_plan(plan, _stiffness_blocks, nv, colind.shape[0], mesh["vertices"])
    assert (writes == 1).all()
    local, _ = orc.p1_assemble(mesh["vertices"], mesh["triangles"], 3, "stiffness")
    want = orc.assemble_csr_values(local, slots.reshape(-1, 3, 3), colind.shape[0])
    assert scaled_error(vals, want) <= 1e-13
    # every element record carries its original element id; each element appears once per
    # tile that owns one of its vertices
    eid = plan["elem_id"][: sizes[1]]
    assert eid.min(initial=0) >= 0 and eid.max(initial=0) < max(mesh["triangles"].shape[0], 1)
    assert np.unique(eid).size == mesh["triangles"].shape[0]


def _ring_case(kind):
    from pytorch_fem_solver_amd import meshgen

    if kind == "structured":
        return meshgen.unit_square(40, 0.25, 0)
    if kind == "tiny":
        return meshgen.unit_square(1, 0.0, 0)
    if kind == "clockwise_mixed":
        mesh = meshgen.unit_square(17, 0.25, 4)
        tri = mesh["triangles"].copy()
        flip = np.random.default_rng(5).random(tri.shape[0]) < 0.4
        tri[flip] = tri[flip][:, [0, 2, 1]]
        mesh["triangles"] = tri
        return mesh
    mesh = meshgen.delaunay_square(3000, 2)
    if kind == "delaunay_shuffled":
        rng = np.random.default_rng(0)
        mesh = meshgen.permute_mesh(mesh, rng.permutation(mesh["vertices"].shape[0]),
                                    rng.permutation(mesh["triangles"].shape[0]))
    return mesh


@pytest.mark.parametrize("kind", ["structured", "tiny", "clockwise_mixed", "delaunay", "delaunay_shuffled",
                                  "delaunay+long_rows", "delaunay_shuffled+long_rows"])
@pytest.mark.parametrize("form", ["stiffness", "stiffness_mass"])
def test_ring_plan_is_a_valid_exact_cover(kind, form, monkeypatch):
    """Walk the ring plan like k_p1_rings does (tests/ring_emulator.py): every CSR entry is
    written exactly once and the values equal the oracle's.  +long_rows: the plan variant
    TFEM_RING_LONG=1 (4-dword records, vertices with 8 .. 15 neighbours in the long-row list)."""
    from pytorch_fem_solver_amd.basis.engine import ring_plan_host, symbolic_host

    kind, _, variant = kind.partition("+")
    long_rows = variant == "long_rows"
    if long_rows:
        monkeypatch.setenv("TFEM_RING_LONG", "1")
    else:
        monkeypatch.delenv("TFEM_RING_LONG", raising=False)
    mesh = _ring_case(kind)
    nv = mesh["vertices"].shape[0]
    rowptr, colind, slots = symbolic_host(mesh["triangles"], nv)
    plan = ring_plan_host(mesh["triangles"], nv, mesh["vertices"], rowptr, colind,
                          own_cap=64 if kind != "delaunay" else None,
                          vert_cap=160 if kind != "delaunay" else None)
    longest = int(np.diff(rowptr).max())
    assert plan["slots"] == (7 if longest <= 8 or long_rows else 15)
    assert (plan["long_rows"].size > 0) == (longest > 8 and long_rows)
    weights = np.asarray(orc.gauss_rule(3)[1]).reshape(-1)
    bary = np.asarray(orc.barycentric_coordinates(orc.gauss_rule(3)[0])).reshape(-1, 3)
    w = 0.5 * weights.sum()
    md = mo = 0.0
    if form == "stiffness_mass":
        md = float((0.5 * weights * bary[:, 0] * bary[:, 0]).sum())
        mo = float((0.5 * weights * bary[:, 0] * bary[:, 1]).sum())
    lamw = (bary * (0.5 * weights)[:, None]).T  # (3, Q): l_i(q) w_q / 2
    fl, geo = orc.p1_assemble(mesh["vertices"], mesh["triangles"], 3, "load")
    fq = orc.source_sin_sin(geo["integration_points"])[..., 0, 0]
    vals, writes, covered, fvec = run_ring_plan(plan, mesh["vertices"], colind.shape[0], w, md, mo,
                                                fq=fq, lamw=lamw)
    assert covered == nv
    assert (writes == 1).all()
    local, _ = orc.p1_assemble(mesh["vertices"], mesh["triangles"], 3, form)
    want = orc.assemble_csr_values(local, slots.reshape(-1, 3, 3), colind.shape[0])
    assert scaled_error(vals, want) <= 1e-13
    # the load vector through the plan's row_elems (element and local index per slot); a plan with
    # long rows (Delaunay meshes) serves source programs only (below)
    want_f = orc.assemble_linear(fl, mesh["triangles"], nv).reshape(-1)
    has_long = plan["long_rows"].size > 0
    if not has_long:
        assert scaled_error(fvec, want_f) <= 1e-13
    # the same vector with the source evaluated per tile from the tile's own coordinates through
    # the element vertex table (what the SRC kernels do), program of tests/test_assembly.py:75-77
    _, _, _, fsrc = run_ring_plan(plan, mesh["vertices"], colind.shape[0], w, md, mo, lamw=lamw,
                                  conn=mesh["triangles"], source=_SIN_SIN_PROGRAM, lam=bary.T)
    assert scaled_error(fsrc, want_f) <= 1e-13


@pytest.mark.parametrize("kind", ["structured", "delaunay", "delaunay_shuffled", "clockwise_mixed"])
@pytest.mark.parametrize("chain", [2, 3, 5])
def test_chain_blocks_evaluate_every_element_share_once(kind, chain, monkeypatch):
    """Source-program launches walk the tiles in chain order (tests/ring_emulator.source_load_vector):
    inside a block of `chain` positions an element in the fans of two consecutive tiles is in the
    table of the earlier tile only, which hands the later tile's shares over; the load vector is
    the oracle's, over the whole order and over the two ranges a sharded step launches (tiles
    owning flagged vertices / the rest), and fewer elements are evaluated than without blocks."""
    from pytorch_fem_solver_amd.basis.engine import ring_plan_host, symbolic_host

    mesh = _ring_case(kind)
    nv = mesh["vertices"].shape[0]
    rowptr, colind, _ = symbolic_host(mesh["triangles"], nv)
    weights = np.asarray(orc.gauss_rule(3)[1]).reshape(-1)
    bary = np.asarray(orc.barycentric_coordinates(orc.gauss_rule(3)[0])).reshape(-1, 3)
    lamw = (bary * (0.5 * weights)[:, None]).T
    fl, _ = orc.p1_assemble(mesh["vertices"], mesh["triangles"], 3, "load")
    want_f = orc.assemble_linear(fl, mesh["triangles"], nv).reshape(-1)
    flags = np.zeros(nv, dtype=bool)
    flags[np.random.default_rng(11).choice(nv, 9, replace=False)] = True
    caps = dict(own_cap=64, vert_cap=160)
    monkeypatch.setenv("TFEM_RING_WGS", "8")  # as if eight workgroups were resident: the small meshes get runs of several tiles
    monkeypatch.setenv("TFEM_RING_RUNS", "0")
    monkeypatch.setenv("TFEM_RING_CHAIN", "1")
    alone = ring_plan_host(mesh["triangles"], nv, mesh["vertices"], rowptr, colind, **caps)
    assert alone["chain_len"] == 1 and alone["n_runs"] <= 0 and np.all(alone["hand_in"] == 0xFFFF)
    d1 = alone["desc"].reshape(-1, 20)
    assert np.array_equal(d1[:, 18] >> 8, d1[:, 17])  # no blocks: every tile evaluates all of its elements
    monkeypatch.setenv("TFEM_RING_CHAIN", str(chain))
    for runs, priority in ((True, None), (False, None), (False, flags)):
        monkeypatch.setenv("TFEM_RING_RUNS", "1" if runs else "0")
        plan = ring_plan_host(mesh["triangles"], nv, mesh["vertices"], rowptr, colind, priority=priority, **caps)
        assert plan["chain_len"] == chain
        # runs (the default without flags): one run of the chain order per resident workgroup, hand-over
        # from every tile to the next; blocks (with flags: the two launches of a sharded step): `chain`
        # positions each, breaking between the ranges
        starts = ring_emulator.chain_block_starts(plan)
        assert (plan["n_runs"] > 0) == runs and starts[0]
        if runs:
            first = plan["runs"]
            assert plan["n_runs"] == 8 and first.size == 9 and first[0] == 0 and first[-1] == plan["n_tiles"]
            # the first quarter of the launch order (the oldest waves of their SIMDs: served first by the
            # vector pipe) takes more tiles than the last: 1.2 : 1.05 : 0.92 : 0.83
            lengths = np.diff(first)
            if plan["n_tiles"] >= 64:  # rounding aside
                assert lengths[:2].sum() >= lengths[2:4].sum() >= lengths[4:6].sum() >= lengths[6:].sum()
            assert lengths.max() - lengths.min() <= max(2, int(0.45 * lengths.mean()) + 1)
        d = plan["desc"].reshape(-1, 20)
        assert plan["tile_tverts"].size == (d[:, 18] >> 8).sum() <= d1[:, 17].sum()
        if plan["n_tiles"] >= 2 * 8:  # enough tiles for runs / blocks of two and more
            assert (d[:, 18] >> 8).sum() < d1[:, 17].sum()
        n_pri, n_tiles = plan["n_priority"], plan["n_tiles"]
        order = plan["chain_order"]
        assert sorted(order[:n_pri].tolist()) == list(range(n_pri))  # the flagged tiles first in both orders
        args = (plan, mesh["vertices"], _SIN_SIN_PROGRAM, bary.T, lamw)
        f_all = ring_emulator.source_load_vector(*args, conn=mesh["triangles"])
        assert scaled_error(f_all, want_f) <= 1e-13
        if priority is not None:
            first = ring_emulator.source_load_vector(*args, tiles=(0, n_pri))
            rest = ring_emulator.source_load_vector(*args, tiles=(n_pri, n_tiles - n_pri))
            assert not np.any(np.isnan(first[flags])) and np.all(np.isnan(first) != np.isnan(rest))
            assert np.array_equal(np.where(np.isnan(first), rest, first), f_all)


@pytest.mark.parametrize("kind", ["structured", "delaunay"])
def test_ring_plan_lists_the_tiles_of_flagged_vertices_first(kind):
    """tfem_ring_plan_create_priority (multi-GPU): same tiles as without flags, the
    ones owning a flagged vertex first; the range launches partition the rows."""
    from pytorch_fem_solver_amd.basis.engine import ring_plan_host, symbolic_host

    mesh = _ring_case(kind)
    nv = mesh["vertices"].shape[0]
    rowptr, colind, slots = symbolic_host(mesh["triangles"], nv)
    flags = np.zeros(nv, dtype=bool)
    flags[np.random.default_rng(5).choice(nv, 7, replace=False)] = True  # a few vertices anywhere
    flags[nv - 40:nv - 20] = True  # and a run of consecutive ids (a strip's shared row)
    caps = dict(own_cap=64, vert_cap=160)
    plain = ring_plan_host(mesh["triangles"], nv, mesh["vertices"], rowptr, colind, **caps)
    plan = ring_plan_host(mesh["triangles"], nv, mesh["vertices"], rowptr, colind, priority=flags, **caps)
    n_pri, n_tiles = plan["n_priority"], plan["n_tiles"]
    assert plain["n_priority"] == 0 and n_tiles == plain["n_tiles"] and 0 < n_pri < n_tiles

    def owned(p):
        d = p["desc"].reshape(-1, 20)
        return [tuple(p["vert_gid"][int(t[0]):int(t[0]) + int(t[7])]) for t in d]

    with_flags, without = owned(plan), owned(plain)
    assert sorted(with_flags) == sorted(without)
    has = [bool(flags[list(t)].any()) for t in with_flags]
    assert all(has[:n_pri]) and not any(has[n_pri:])
    # stable: both groups keep the order of the plan without flags
    assert [t for t in without if flags[list(t)].any()] == with_flags[:n_pri]
    assert [t for t in without if not flags[list(t)].any()] == with_flags[n_pri:]
    nnz = colind.shape[0]
    full, writes, covered = run_ring_plan(plan, mesh["vertices"], nnz)[:3]
    assert covered == nv and (writes == 1).all()
    first, w_first, c_first = run_ring_plan(plan, mesh["vertices"], nnz, tiles=(0, n_pri))[:3]
    rest, w_rest, c_rest = run_ring_plan(plan, mesh["vertices"], nnz, tiles=(n_pri, n_tiles - n_pri))[:3]
    assert c_first + c_rest == nv and ((w_first + w_rest) == 1).all()
    row_of = np.repeat(np.arange(nv), np.diff(rowptr))
    assert (w_first[flags[row_of]] == 1).all()  # every flagged row is complete after the first range
    assert np.array_equal(np.where(w_first == 1, first, rest), full)
    with pytest.raises(ValueError):
        ring_plan_host(mesh["triangles"], nv, mesh["vertices"], rowptr, colind, priority=flags[:-1])


@pytest.mark.parametrize("kind", ["structured", "delaunay", "delaunay_shuffled"])
def test_ring_plan_from_the_pattern_handle_is_the_same_plan(kind):
    """tfem_ring_plan_create_from_pattern (the engine's path: the incidence of the CSR pattern
    handle is reused) against tfem_ring_plan_create_priority: byte-identical plans."""
    from pytorch_fem_solver_amd.basis.engine import pattern_host, ring_plan_host

    mesh = _ring_case(kind)
    nv = mesh["vertices"].shape[0]
    conn = np.ascontiguousarray(mesh["triangles"].astype(np.int32))
    rowptr, colind, keeper = pattern_host(conn, nv, keep=True)
    flags = np.zeros(nv, dtype=bool)
    flags[::37] = True
    for priority in (None, flags):
        alone = ring_plan_host(conn, nv, mesh["vertices"], rowptr, colind, priority=priority)
        shared = ring_plan_host(conn, nv, mesh["vertices"], rowptr, colind, priority=priority, pattern=keeper)
        assert np.array_equal(alone["layout"], shared["layout"]) and alone["n_priority"] == shared["n_priority"]
        assert alone["blob"].tobytes() == shared["blob"].tobytes()
    keeper.release()
    assert keeper.handle is None
    keeper.release()  # idempotent
    # P2 pattern handles are refused
    p2 = np.ascontiguousarray(np.tile(np.arange(6, dtype=np.int32), (2, 1)))
    _, _, k6 = pattern_host(p2, 6, keep=True)
    with pytest.raises(ValueError):
        ring_plan_host(conn, nv, mesh["vertices"], rowptr, colind, pattern=k6)
    k6.release()


def test_ring_plan_open_fans_and_isolated_vertices():
    """Two fans meeting in one vertex (a bow tie) chain as two open fans; a vertex without
    elements owns an empty row."""
    from pytorch_fem_solver_amd.basis.engine import ring_plan_host, symbolic_host

    verts = np.array([[0, 0], [1, 0], [1, 1], [-1, 0], [-1, -1], [5, 5], [0.3, 1.2]], dtype=np.float64)
    tris = np.array([[0, 1, 2], [0, 3, 4], [2, 6, 0]], dtype=np.int32)
    rowptr, colind, slots = symbolic_host(tris, 7)
    plan = ring_plan_host(tris, 7, verts, rowptr, colind)
    vals, writes, covered = run_ring_plan(plan, verts, colind.shape[0])
    assert covered == 7 and (writes == 1).all()
    local, _ = orc.p1_assemble(verts, tris, 1, "stiffness")
    want = orc.assemble_csr_values(local, slots.reshape(-1, 3, 3), colind.shape[0])
    assert scaled_error(vals, want) <= 1e-13


def test_ring_plan_rejects_fans_without_a_ring_form():
    from pytorch_fem_solver_amd.basis.engine import ring_plan_host, symbolic_host

    # three triangles on the edge (0, 1)
    verts = np.array([[0, 0], [1, 0], [0.5, 1], [0.5, -1], [0.5, 2]], dtype=np.float64)
    tris = np.array([[0, 1, 2], [1, 0, 3], [0, 1, 4]], dtype=np.int32)
    rowptr, colind, _ = symbolic_host(tris, 5)
    with pytest.raises(NotImplementedError, match="fans"):
        ring_plan_host(tris, 5, verts, rowptr, colind)
    # the same triangle twice
    tris = np.array([[0, 1, 2], [1, 2, 0]], dtype=np.int32)
    rowptr, colind, _ = symbolic_host(tris, 5)
    with pytest.raises(NotImplementedError, match="fans"):
        ring_plan_host(tris, 5, verts, rowptr, colind)


@pytest.mark.parametrize("kind", ["structured", "clockwise_mixed", "tiny"])
@pytest.mark.parametrize("form", ["stiffness", "stiffness_mass"])
def test_p2_row_plan_is_a_valid_exact_cover(kind, form):
    """Walk the P2 row plan like k_p2_rows does (tests/p2rows_emulator.py): every CSR entry is
    written exactly once and the values equal the oracle's P2 assembly."""
    from pytorch_fem_solver_amd import dofs
    from pytorch_fem_solver_amd.basis.engine import p2_plan_host, symbolic_host

    mesh = _ring_case(kind)
    conn6, xy, _ = dofs.p2_dofs_numpy(mesh["vertices"], mesh["triangles"], mesh["edges"],
                                      mesh["edge_markers"], mesh["vertex_markers"])
    nv, nd = mesh["vertices"].shape[0], xy.shape[0]
    rowptr, colind, slots = symbolic_host(conn6, nd)
    plan = p2_plan_host(conn6, nv, nd, mesh["vertices"], rowptr, colind)
    beta = 1.0 if form == "stiffness_mass" else 0.0
    vals, writes = run_p2_plan(plan, mesh["vertices"], rowptr, nv, 2, 1.0, beta)
    assert (writes == 1).all()
    geo = orc.geometry(mesh["vertices"][mesh["triangles"]], 2, 2)
    integrand = orc.integrand_stiffness_mass(geo) if beta else orc.integrand_stiffness(geo)
    local = orc.integrate_local(integrand, geo["dx"])
    want = orc.assemble_csr_values(local, slots.reshape(-1, 6, 6), colind.shape[0])
    assert scaled_error(vals, want) <= 1e-13
    _check_p2_load_codes(plan, mesh, conn6, nv, nd)


def _check_p2_load_codes(plan, mesh, conn6, nv, nd):
    """The load vector walked from the plan's element codes (k_p2_load_rows, tests/p2rows_emulator.py):
    every DoF written once, equal to the oracle's assembled P2 load vector -- with source values that
    differ from point to point and from element to element, so a wrong element or local index shows."""
    for order in (2, 4):
        geo = orc.geometry(mesh["vertices"][mesh["triangles"]], 2, order)
        nq = geo["dx"].shape[1]
        fq = np.random.default_rng(5).uniform(-1.0, 2.0, size=(conn6.shape[0], nq))
        f, writes = run_p2_load_plan(plan, mesh["vertices"], nv, nd, fq, order)
        assert (writes == 1).all()
        local = orc.integrate_local(fq.reshape(-1, nq, 1, 1) * geo["v"], geo["dx"])
        want = orc.assemble_linear(local, conn6, nd).reshape(-1)
        assert scaled_error(f, want) <= 1e-13


def test_p2_row_plan_long_rows_on_delaunay_meshes_and_what_it_rejects():
    """Vertices with 8 .. 15 neighbours (every Delaunay mesh) become long rows; the walk of the
    plan -- tile rows, long rows, edge rows -- writes every CSR entry once with the oracle's
    values.  A numbering without locality and a vertex with 16 neighbours are refused."""
    from pytorch_fem_solver_amd import dofs, meshgen
    from pytorch_fem_solver_amd.basis.engine import p2_plan_host, symbolic_host

    native = meshgen.delaunay_square(3000, 2)
    mesh = meshgen.permute_mesh(native, vertex_order=meshgen.morton_order(native["vertices"]))
    conn6, xy, _ = dofs.p2_dofs_numpy(mesh["vertices"], mesh["triangles"], mesh["edges"],
                                      mesh["edge_markers"], mesh["vertex_markers"])
    nv, nd = mesh["vertices"].shape[0], xy.shape[0]
    rowptr, colind, slots = symbolic_host(conn6, nd)
    assert int(np.diff(rowptr[: nv + 1]).max()) > 22  # rows of vertices with more than 7 neighbours
    try:
        plan = p2_plan_host(conn6, nv, nd, mesh["vertices"], rowptr, colind)
    except NotImplementedError as exc:  # the edge numbering of this generator may lack locality
        assert "locality" in str(exc)
        plan = None
    if plan is not None:
        assert plan["long_rows"].size >= 32
        for beta in (0.0, 1.0):
            vals, writes = run_p2_plan(plan, mesh["vertices"], rowptr, nv, 2, 1.0, beta)
            assert (writes == 1).all()
            geo = orc.geometry(mesh["vertices"][mesh["triangles"]], 2, 2)
            integrand = orc.integrand_stiffness_mass(geo) if beta else orc.integrand_stiffness(geo)
            local = orc.integrate_local(integrand, geo["dx"])
            want = orc.assemble_csr_values(local, slots.reshape(-1, 6, 6), colind.shape[0])
            assert scaled_error(vals, want) <= 1e-13
        assert plan["long_codes"].size >= 16
        _check_p2_load_codes(plan, mesh, conn6, nv, nd)
    # a fan of 16 triangles around one vertex
    k = 16
    ang = np.linspace(0, 2 * np.pi, k, endpoint=False)
    verts = np.concatenate([[[0.0, 0.0]], np.stack([np.cos(ang), np.sin(ang)], 1)])
    tris = np.array([[0, 1 + i, 1 + (i + 1) % k] for i in range(k)], dtype=np.int32)
    edges = np.unique(np.sort(np.concatenate([tris[:, [0, 1]], tris[:, [1, 2]], tris[:, [2, 0]]]), axis=1), axis=0)
    conn6, xy, _ = dofs.p2_dofs_numpy(verts, tris, edges.astype(np.int32), np.zeros((edges.shape[0], 1), np.int32),
                                      np.zeros((k + 1, 1), np.int32))
    rowptr, colind, _ = symbolic_host(conn6, xy.shape[0])
    with pytest.raises(NotImplementedError, match="15 neighbours"):
        p2_plan_host(conn6, k + 1, xy.shape[0], verts, rowptr, colind)


@pytest.mark.parametrize("seed", range(10))
def test_random_meshes_through_the_plan_emulators(seed):
    """Seeded random meshes (holes, flips, rotated local numbering, renumbering: the generator
    of tests/test_hip_fuzz.py) through the host plan builders and the numpy emulators of the
    ring and tile kernels: exact covers, values equal to the oracle's."""
    from test_hip_fuzz import _random_mesh

    from pytorch_fem_solver_amd.basis.engine import ring_plan_host, symbolic_host, tile_plan_host

    rng = np.random.default_rng(500 + seed)
    verts, tris = _random_mesh(rng)
    if tris.shape[0] > 6000:
        tris = tris[:6000]
    nv = verts.shape[0]
    rowptr, colind, slots = symbolic_host(tris, nv)
    geo = orc.geometry(verts[tris], 1, 2)
    local = orc.integrate_local(orc.integrand_stiffness(geo), geo["dx"])
    want = orc.assemble_csr_values(local, slots.reshape(-1, 3, 3), colind.shape[0])
    try:
        plan = ring_plan_host(tris, nv, verts, rowptr, colind)
    except NotImplementedError:
        plan = None
    if plan is not None:
        weights = np.asarray(orc.gauss_rule(2)[1]).reshape(-1)
        bary = np.asarray(orc.barycentric_coordinates(orc.gauss_rule(2)[0])).reshape(-1, 3)
        lamw = (bary * (0.5 * weights)[:, None]).T
        fq = orc.source_sin_sin(geo["integration_points"])[..., 0, 0]
        vals, writes, covered, fvec = run_ring_plan(plan, verts, colind.shape[0], 0.5 * weights.sum(), fq=fq, lamw=lamw)
        assert covered == nv and (writes == 1).all()
        assert scaled_error(vals, want) <= 1e-12
        fl = orc.integrate_local(orc.integrand_load(geo), geo["dx"])
        if plan["long_rows"].size == 0:
            assert scaled_error(fvec, orc.assemble_linear(fl, tris, nv).reshape(-1)) <= 1e-12
        _, _, _, fsrc = run_ring_plan(plan, verts, colind.shape[0], 0.5 * weights.sum(), lamw=lamw,
                                      conn=tris, source=_SIN_SIN_PROGRAM, lam=bary.T)
        assert scaled_error(fsrc, orc.assemble_linear(fl, tris, nv).reshape(-1)) <= 1e-12
    try:
        tplan = tile_plan_host(tris, nv, verts, rowptr, colind)
    except NotImplementedError:
        tplan = None
    if tplan is not None:
        vals, writes = run_plan(tplan, _stiffness_blocks, nv, colind.shape[0], verts)
        assert (writes == 1).all() and scaled_error(vals, want) <= 1e-12
    assert plan is not None or tplan is not None or int(np.diff(rowptr).max()) > 16


def test_tile_plan_rejects_rows_longer_than_16_entries():
    from pytorch_fem_solver_amd.basis.engine import symbolic_host, tile_plan_host

    # a fan of 20 triangles around vertex 0: its row has 21 entries
    k = 20
    ang = np.linspace(0, 2 * np.pi, k, endpoint=False)
    verts = np.concatenate([[[0.0, 0.0]], np.stack([np.cos(ang), np.sin(ang)], 1)])
    tris = np.array([[0, 1 + i, 1 + (i + 1) % k] for i in range(k)], dtype=np.int32)
    rowptr, colind, _ = symbolic_host(tris, k + 1)
    with pytest.raises(NotImplementedError, match="> 16"):
        tile_plan_host(tris, k + 1, verts, rowptr, colind)


# ---------------------------------------------------------------------------------------
# form tracer (basis/forms.py)
# ---------------------------------------------------------------------------------------
class _FakeBasis:
    def __init__(self):
        self.integration_points = torch.rand(5, 4, 1, 2)
        self.v = torch.rand(4, 3, 1)
        self.v_grad = torch.rand(5, 1, 3, 2)
        self.mesh = "mesh"


def _rhs(x, y):
    import math

    return 2.0 * math.pi**2 * torch.sin(math.pi * x) * torch.sin(math.pi * y)


def test_tracer_recognises_the_reference_vocabulary():
    from pytorch_fem_solver_amd.basis import forms

    b = _FakeBasis()
    e = forms.trace(lambda basis: basis.v_grad @ basis.v_grad.mT, b, (), {})
    assert isinstance(e, forms.BilinearExpr) and (e.alpha, e.beta) == (1.0, 0.0)
    e = forms.trace(lambda basis: basis.v_grad @ basis.v_grad.mT + basis.v @ basis.v.mT, b, (), {})
    assert (e.alpha, e.beta) == (1.0, 1.0)
    e = forms.trace(lambda basis: 2.5 * (basis.v @ basis.v.mT) + (basis.v_grad @ basis.v_grad.mT) * 3, b, (), {})
    assert (e.alpha, e.beta) == (3.0, 2.5)
    e = forms.trace(lambda basis: basis.v_grad @ basis.v_grad.mT - basis.v @ basis.v.mT, b, (), {})
    assert (e.alpha, e.beta) == (1.0, -1.0)

    def load(basis, scale):  # tests/test_assembly.py:75-84
        x, y = torch.split(basis.integration_points, 1, dim=-1)
        return scale * _rhs(x, y) * basis.v

    e = forms.trace(load, b, (1.0,), {})
    assert isinstance(e, forms.LinearExpr) and isinstance(e.coefficient, forms.SourceExpr) and e.flux is None
    names = [name for name, _ in forms.compile_ops(e.coefficient.node)]
    assert names == ["PUSH_X", "SIN", "PUSH_Y", "SIN", "MUL", "MUL_C"]
    assert torch.equal(e.materialize(), load(b, 1.0))  # the same torch operations in the same order
    program = e.coefficient.program()
    assert program.n_ops == 6 and list(program.ops[:3]) == [forms.OPS["PUSH_X"], forms.OPS["SIN"], forms.OPS["PUSH_Y"]]
    assert list(program.consts[:2]) == [np.pi, 2.0 * np.pi**2]
    # coordinate columns by indexing; a tensor coefficient (evaluated by the caller)
    e = forms.trace(lambda basis: basis.v * basis.integration_points[..., [0]], b, (), {})
    assert isinstance(e, forms.LinearExpr) and e.coefficient.node == ("x",)
    e = forms.trace(lambda basis: basis.integration_points[..., 1:2] ** 2 * basis.v, b, (), {})
    assert e.coefficient.node == ("powi", ("y",), 2)
    coefficient = torch.rand(5, 4, 1, 1)
    e = forms.trace(lambda basis: coefficient * basis.v, b, (), {})
    assert isinstance(e, forms.LinearExpr) and e.coefficient is coefficient
    # the fracture example's source (example_fractures_fem.py:69-99) on a 2-D mesh
    def frac(basis):
        x, y = torch.split(basis.integration_points, 1, dim=-1)
        return (6.0 * (y - y**2) * torch.abs(x) - 2.0 * (torch.abs(x) ** 3 - torch.abs(x))) * basis.v

    e = forms.trace(frac, b, (), {})
    assert isinstance(e.coefficient, forms.SourceExpr) and e.coefficient.program() is not None
    assert torch.equal(e.materialize(), frac(b))
    # a functional of the coordinates (tests/test_assembly.py:86-90)
    e = forms.trace(lambda basis: _rhs(*torch.split(basis.integration_points, 1, dim=-1)) ** 2, b, (), {})
    assert isinstance(e, forms.SourceExpr) and forms.compile_ops(e.node)[-1] == ("POW_I", 2.0)
    assert forms.trace(lambda basis: basis.mesh, b, (), {}) == "mesh"  # real attributes pass through


def test_tracer_residual_form_with_an_opaque_gradient_field():
    """examples/example_weak.py:64-75: f v - v_grad @ g.mT with g from code the tracer cannot see
    into (requires_grad_ + autograd.grad on the points, model/neural_network.py:85-100)."""
    from pytorch_fem_solver_amd.basis import forms

    b = _FakeBasis()
    lin = torch.nn.Linear(2, 1).double()

    def gradient(points):
        points.requires_grad_(True)
        out = torch.tanh(lin(points))
        return torch.autograd.grad([out], [points], [torch.ones_like(out)], create_graph=True)[0]

    def residual(basis, grad):
        points = basis.integration_points
        x, y = torch.split(points, 1, dim=-1)
        return _rhs(x, y) * basis.v - (basis.v_grad @ grad(points).mT)

    e = forms.trace(residual, b, (gradient,), {})
    assert isinstance(e, forms.LinearExpr) and isinstance(e.coefficient, forms.SourceExpr)
    assert e.flux.shape == (5, 4, 1, 2) and e.flux_sign == -1.0 and e.flux.requires_grad
    assert not b.integration_points.requires_grad  # the basis's cached tensor is left alone
    want = residual(_FakeBasis.__new__(_FakeBasis).__class__() if False else b, gradient)
    assert torch.allclose(e.materialize(), want, rtol=0, atol=1e-15)
    # the loss differentiates through the flux into the network
    e.materialize().sum().backward()
    assert lin.weight.grad is not None and float(lin.weight.grad.abs().sum()) > 0


def test_tracer_hands_back_tensors_for_anything_else():
    """Outside the vocabulary the symbols turn into the tensors they stand for: the callable runs
    once and the result equals the callable on the real basis."""
    from pytorch_fem_solver_amd.basis import forms

    b = _FakeBasis()
    unknown = [
        lambda basis: basis.v @ basis.v_grad[..., [0]].mT,           # convection: indexing
        lambda basis: basis.v_grad[..., [1]] @ basis.v.mT,           # mixed
        lambda basis: torch.sin(basis.v),                            # torch function on a symbol
        lambda basis: basis.v * basis.v,
        lambda basis: (basis.v_grad @ basis.v_grad.mT) * torch.ones(5, 4, 1, 1),  # tensor coefficient
        lambda basis: basis.integration_points * 2.0,               # no symbol at all
        lambda basis: torch.cat(torch.split(basis.integration_points, 1, dim=-1), dim=-1).sum(-1, keepdim=True) * basis.v,
        lambda basis: torch.split(basis.integration_points, 1, dim=-1)[0] ** 2.5,  # pow outside the format
    ]
    for fn in unknown:
        got = forms.trace(fn, b, (), {})
        got = forms.materialize(got)
        assert isinstance(got, torch.Tensor) and not isinstance(got, forms.PointsSymbol)
        assert torch.equal(got, fn(b))
    calls = []

    def assigns(basis):  # the proxy refuses; the real basis is used, and directly from then on
        calls.append(type(basis).__name__)
        basis.scratch = 1
        return basis.v * 2.0

    assert torch.equal(forms.trace(assigns, b, (), {}), b.v * 2.0)
    assert torch.equal(forms.trace(assigns, b, (), {}), b.v * 2.0)
    assert calls == ["TracingBasis", "_FakeBasis", "_FakeBasis"]


def test_source_program_compiler_and_validator():
    """Sethi-Ullman ordering keeps the stack within the format; the library's validator accepts
    what the compiler emits and rejects broken programs."""
    import ctypes as ct

    from pytorch_fem_solver_amd import _native
    from pytorch_fem_solver_amd.basis import forms
    from oracle import assembly_oracle as orc

    lib = _native.load()
    b = _FakeBasis()
    x, y = forms.SourceExpr(b, ("x",)), forms.SourceExpr(b, ("y",))
    deep = x + (y * (x - (y / ((x + 2.0) * (y - x)))))  # right-leaning: needs the reversed operations
    ops = forms.compile_ops(deep.node)
    assert ops is not None and {"SUB_R", "DIV_R"} <= {name for name, _ in ops}
    program = deep.program()
    assert lib.tfem_source_validate(ct.byref(program)) == 0
    pts = b.integration_points
    got = orc.source_program_eval(list(program.ops[: program.n_ops]), list(program.consts[: program.n_ops]),
                                  pts[..., 0:1].numpy(), pts[..., 1:2].numpy())
    assert np.allclose(got, deep.materialize().numpy(), rtol=1e-15, atol=0)
    balanced = ((x + y) * (x - y)) / ((x * y) + (x / y))  # needs 3 entries
    assert forms.compile_ops(balanced.node) is not None
    too_deep = balanced
    for _ in range(3):
        too_deep = (too_deep * (x + 1.0)) / ((too_deep - y) * (too_deep + y))
    assert too_deep.program() is None  # more than 32 operations / 4 entries: torch evaluates it
    bad = _native.SourceProgram()
    bad.n_ops = 1
    bad.ops[0] = forms.OPS["ADD"]
    assert lib.tfem_source_validate(ct.byref(bad)) == 1 and b"empty stack" in lib.tfem_last_error()
    bad.ops[0] = forms.OPS["PUSH_X"]
    bad.n_ops = 2
    bad.ops[1] = forms.OPS["PUSH_Y"]
    assert lib.tfem_source_validate(ct.byref(bad)) == 1 and b"leaves 2" in lib.tfem_last_error()
    bad.ops[1] = forms.OPS["POW_I"]
    bad.consts[1] = 2.5
    assert lib.tfem_source_validate(ct.byref(bad)) == 1
    bad.ops[1] = 99
    assert lib.tfem_source_validate(ct.byref(bad)) == 1


# ---------------------------------------------------------------------------------------
# meshes (torch host code) against the reference's own outputs
# ---------------------------------------------------------------------------------------
def test_mesh_topology_matches_reference():
    import pytorch_fem_solver_amd as tf

    d = load_golden("mesh_topology_n4.npz")
    mesh = tf.MeshTri(mesh_from_golden(d))
    assert mesh._topology_pending  # edge topology is derived on first access only
    assert np.array_equal(mesh["cells", "coordinates"].numpy(), d["out_cells_coordinates"])
    for group in ("interior_edges", "boundary_edges"):
        for key, value in mesh[group].items():
            assert np.array_equal(value.numpy(), d[f"out_{group}_{key}"]), (group, key)
    assert np.array_equal(mesh["cells", "length"].numpy(), d["out_cells_length"])
    # without `neighbors` the edge -> cells map is found by matching (same cells, per edge)
    no_nb = {k: v for k, v in mesh_from_golden(d).items() if k != "neighbors"}
    m2 = tf.MeshTri(no_nb)
    cells = d["in_triangles"]
    for e, (a, b) in enumerate(m2["interior_edges", "vertices"].tolist()):
        want = [t for t in range(cells.shape[0]) if a in cells[t] and b in cells[t]]
        assert m2["interior_edges", "cells"][e].tolist() == want


@pytest.mark.parametrize("fixture", ["fracture_L4.npz", "fracture_L3_jitter.npz"])
def test_fracture_mesh_and_global_numbering_match_reference(fixture):
    import pytorch_fem_solver_amd as tf

    d = load_golden(fixture)
    tri = mesh_from_golden(d)
    mesh = tf.FracturesTri([tri, tri], torch.tensor(d["in_fractures_3d"]))
    for key in ("jacobian_fracture_map", "inv_jacobian_fracture_map", "det_jacobian_fracture_map",
                "translation_vector"):
        assert np.array_equal(mesh[key].numpy(), d["out_mesh_" + key])
    assert np.array_equal(mesh["vertices", "coordinates_3d"].numpy(), d["out_mesh_vertices_coordinates_3d"])
    for key, value in mesh["interior_edges"].items():
        assert np.array_equal(value.numpy(), d["out_mesh_interior_edges_" + key]), key
    basis = tf.FractureBasis(mesh, tf.ElementTri(1, 4))  # no kernel launch until a form is integrated
    for key, value in basis.global_triangulation.items():
        assert np.array_equal(value.numpy(), d["out_gt_" + key]), key
    assert np.array_equal(basis._basis_parameters["inner_dofs"].numpy(), d["out_inner_dofs"])
    rows, cols = basis._basis_parameters["bilinear_form_idx"]  # lazily built dense indices
    conn = d["out_gt_triangles"]
    assert np.array_equal(rows.numpy(), np.tile(conn, (1, 3)).reshape(-1))
    assert np.array_equal(cols.numpy(), np.repeat(conn.reshape(-1), 3))


def test_elements_and_quadrature_tables_match_reference():
    import pytorch_fem_solver_amd as tf

    d = load_golden("p1_square_n8.npz")
    for order in (1, 2, 3, 4):
        el = tf.ElementTri(1, order)
        assert np.array_equal(el.gaussian_nodes.numpy(), d[f"out_q{order}_gaussian_nodes"])
        assert np.array_equal(el.gaussian_weights.numpy(), d[f"out_q{order}_gaussian_weights"])
    p2 = load_golden("p2_element.npz")
    el = tf.ElementTri(2, 3)
    bar = el.compute_barycentric_coordinates(el.gaussian_nodes)
    v, v_grad = el.compute_shape_functions(bar, torch.tensor(p2["out_q3_inv"]))
    assert scaled_error(v, p2["out_q3_v"]) <= 1e-15
    assert scaled_error(v_grad, p2["out_q3_v_grad"]) <= 1e-14
    with pytest.raises(NotImplementedError):
        tf.ElementTri(1, 9)
    with pytest.raises(NotImplementedError):
        tf.ElementLine(1, 5)


def test_assembly_without_a_gpu_fails_loudly():
    """There is no CPU fallback: the first hot-path call raises."""
    import pytorch_fem_solver_amd as tf
    from pytorch_fem_solver_amd.basis.engine import NoDeviceError

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    d = load_golden("p1_square_n8.npz")
    basis = tf.Basis(tf.MeshTri(mesh_from_golden(d)), tf.ElementTri(1, 3))
    with pytest.raises(NoDeviceError):
        basis.integrate_bilinear_form(lambda b: b.v_grad @ b.v_grad.mT)
    with pytest.raises(NoDeviceError):
        basis.v_grad


def test_torch_fem_alias_package():
    import torch_fem
    import pytorch_fem_solver_amd as tf

    for name in ("Basis", "FractureBasis", "InteriorEdgesBasis", "InteriorEdgesFractureBasis",
                 "ElementLine", "ElementTri", "FracturesTri", "MeshTri"):
        assert getattr(torch_fem, name) is getattr(tf, name)
    from torch_fem.basis import Basis as B2
    from torch_fem.element import ElementTri as E2
    from torch_fem.mesh import MeshTri as M2

    assert B2 is tf.Basis and E2 is tf.ElementTri and M2 is tf.MeshTri


def test_bench_reads_the_committed_profiles():
    """bench.py fills roofline.traffic / kernel_ms_rocprofv3 from the committed rocprofv3
    summary under profiles/ -- when it was taken from the kernel sources of this tree (digest in
    the summary).  Both launches of the bench kernel must be found there."""
    import json

    import bench

    with open(os.path.join(REPO, "profiles", f"{bench.PROFILE_TAG}_bench_pmc_summary.json")) as fh:
        summary = json.load(fh)
    names = list(summary["kernels"])
    assert sum(bench.is_launch(n, "k_p1_rings", True) for n in names) == 1
    assert sum(bench.is_launch(n, "k_p1_rings", False) for n in names) == 1
    profile = bench.committed_profile(2236, 3, "k_p1_rings")
    if summary["source_sha"] != bench.source_sha():
        assert profile == {"fused": {}, "k_only": {}}  # stale profile: nothing is quoted
        return
    for key, with_load in (("fused", True), ("k_only", False)):
        traffic, duration = profile[key]["traffic"], profile[key]["kernel_ms_rocprofv3"]
        # the counters saw the launch's compulsory traffic (their resolution leaves a percent or so
        # below it on the matrix-only launch) and at most 1.5x of it
        algo = bench.algorithmic_bytes(9999392, 5004169, 35011289, with_load)
        assert 0.98 * algo <= traffic <= 1.5 * algo
        assert 0.05 <= duration <= 0.40  # ms
    assert bench.committed_profile(100, 3, "k_p1_rings") == {"fused": {}, "k_only": {}}  # another workload


def test_host_builders_do_not_depend_on_the_thread_count_and_keep_their_digest():
    """The once-per-mesh builders (CSR pattern, slot map, ring plan) are multi-threaded
    (csrc/tfem_threads.hpp): 1, 3 and the default number of threads give identical bytes, and the
    bytes are the ones the sequential builder produced for these meshes (digests taken from that
    build before the builders were restructured; retaken in round 3 when the plan gained the chain
    order, the carry maps and per-tile element tables, and again when the run lengths became the
    integers that minimise the longest run -- the arrays of round 2 are unchanged)."""
    import hashlib

    from pytorch_fem_solver_amd import meshgen
    from pytorch_fem_solver_amd.basis.engine import ring_plan_host, symbolic_host

    delaunay = meshgen.delaunay_square(30000, 3)
    cases = {
        "S300": (meshgen.unit_square(300, 0.25, 0), "3adf95e8ef4c2a99", "9b34763e1a02eab9"),
        "Dmorton": (meshgen.permute_mesh(delaunay, vertex_order=meshgen.morton_order(delaunay["vertices"])),
                    "d4e6d43a65fa5d31", None),
        "Dnative": (delaunay, "84c9b046a5316196", None),
    }
    saved = os.environ.get("TFEM_HOST_THREADS")
    saved_long = os.environ.get("TFEM_RING_LONG")
    os.environ.pop("TFEM_RING_LONG", None)  # the digests are those of the default plans (8-dword records)
    try:
        for name, (mesh, plan_digest, pattern_digest) in cases.items():
            nv = mesh["vertices"].shape[0]
            seen = set()
            for threads in ("1", "3", None):
                if threads is None:
                    os.environ.pop("TFEM_HOST_THREADS", None)
                else:
                    os.environ["TFEM_HOST_THREADS"] = threads
                rowptr, colind, slots = symbolic_host(mesh["triangles"], nv)
                plan = ring_plan_host(mesh["triangles"], nv, mesh["vertices"], rowptr, colind)
                sym = hashlib.sha256(rowptr.tobytes() + colind.tobytes() + slots.tobytes()).hexdigest()[:16]
                seen.add((hashlib.sha256(plan["blob"].tobytes()).hexdigest()[:16], sym))
            assert len(seen) == 1, name
            got_plan, got_sym = next(iter(seen))
            assert got_plan == plan_digest, name
            assert pattern_digest is None or got_sym == pattern_digest
        # the plans with long rows (TFEM_RING_LONG=1): thread-count invariance
        os.environ["TFEM_RING_LONG"] = "1"
        for name in ("Dmorton", "Dnative"):
            mesh = cases[name][0]
            nv = mesh["vertices"].shape[0]
            rowptr, colind, _ = symbolic_host(mesh["triangles"], nv)
            seen = set()
            for threads in ("1", "5"):
                os.environ["TFEM_HOST_THREADS"] = threads
                plan = ring_plan_host(mesh["triangles"], nv, mesh["vertices"], rowptr, colind)
                assert plan["slots"] == 7 and plan["long_rows"].size > 0
                seen.add(hashlib.sha256(plan["blob"].tobytes()).hexdigest())
            assert len(seen) == 1, name
    finally:
        for key, value in (("TFEM_HOST_THREADS", saved), ("TFEM_RING_LONG", saved_long)):
            if value is None:
                os.environ.pop(key, None)
            else:
                os.environ[key] = value
