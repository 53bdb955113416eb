"""Seeded random sweep of the P1 / P2 kernels against the oracle: Delaunay and structured
meshes of random size with random element removal (open fans, several fans per vertex,
isolated vertices), random orientation flips, random vertex / element renumbering, every
kernel mode the engine offers.  Runs on a real MI355X only (-m gpu)."""

import numpy as np
import pytest
import torch

from conftest import scaled_error
from oracle import assembly_oracle as orc

import os

pytestmark = pytest.mark.gpu

#: TFEM_FUZZ_SEEDS=n widens the sweep (developer runs); the default keeps the suite short
N_P1 = int(os.environ.get("TFEM_FUZZ_SEEDS", "100"))
N_P2 = max(N_P1 // 3, 8)


@pytest.fixture(autouse=True)
def _gpu_defaults():
    assert torch.cuda.is_available()
    torch.set_default_dtype(torch.float64)
    torch.set_default_device("cuda")
    yield
    torch.set_default_device("cpu")
    torch.set_default_dtype(torch.float32)


def _random_mesh(rng):
    from pytorch_fem_solver_amd import meshgen

    big = rng.random() < 0.1  # now and then a mesh of many tiles
    if rng.random() < 0.5:
        mesh = meshgen.unit_square(int(rng.integers(2, 200 if big else 60)), float(rng.uniform(0.0, 0.3)),
                                   int(rng.integers(1 << 30)))
    else:
        mesh = meshgen.delaunay_square(int(rng.integers(30, 40000 if big else 4000)), int(rng.integers(1 << 30)))
    verts, tris = mesh["vertices"].copy(), mesh["triangles"].copy()
    if rng.random() < 0.5:  # holes: open fans, several fans per vertex, isolated vertices
        keep = rng.random(tris.shape[0]) >= rng.uniform(0.02, 0.3)
        if keep.sum() >= 1:
            tris = tris[keep]
    if rng.random() < 0.5:  # stored orientation
        flip = rng.random(tris.shape[0]) < rng.uniform(0.05, 0.6)
        tris[flip] = tris[flip][:, [0, 2, 1]]
    if rng.random() < 0.5:  # rotate the local numbering of elements
        shift = rng.integers(0, 3, size=tris.shape[0])
        tris = np.stack([tris[np.arange(tris.shape[0]), (shift + j) % 3] for j in range(3)], axis=1)
    renumber = rng.random()
    if renumber < 0.7:  # numbering without locality (< 0.35) or along a Morton curve
        perm = rng.permutation(verts.shape[0]) if renumber < 0.35 else meshgen.morton_order(verts)
        inv = np.empty_like(perm)
        inv[perm] = np.arange(perm.size)
        verts, tris = verts[perm], inv[tris].astype(np.int32)
    if rng.random() < 0.4:
        tris = tris[rng.permutation(tris.shape[0])]
    return verts, np.ascontiguousarray(tris.astype(np.int32))


def _caller_values(eng, vals):
    """CSR values in the caller's numbering (an engine that renumbered its mesh stores them in its own)."""
    return eng.wrap_csr(vals).caller_numbering().values if eng.renumbered else vals


@pytest.mark.parametrize("seed", range(N_P1))
def test_random_p1_meshes_every_kernel_mode(seed, monkeypatch):
    from pytorch_fem_solver_amd.basis.engine import AssemblyEngine

    rng = np.random.default_rng(1000 + seed)
    if seed % 4 == 3:  # every fourth mesh through the engine's own Morton renumbering, whatever its size
        monkeypatch.setenv("TFEM_RENUMBER", "1")
    verts, tris = _random_mesh(rng)
    nv = verts.shape[0]
    order = int(rng.integers(1, 5))
    single = rng.random() < 0.15  # float32: the tables and the geometry are rounded first
    if single:
        verts = verts.astype(np.float32)
    tol = 5e-5 if single else 1e-12
    dtype = torch.float32 if single else torch.float64
    alpha, beta = (1.0, 0.0) if rng.random() < 0.4 else (float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.0, 3.0)))
    geo = orc.geometry(verts[tris], 1, order)
    integrand = alpha * orc.integrand_stiffness(geo) + beta * orc.integrand_mass(geo)
    local = orc.integrate_local(integrand, geo["dx"])
    _, colind, slots = orc.csr_pattern(tris, nv)
    want = orc.assemble_csr_values(local, slots, colind.shape[0])
    fq_np = orc.source_sin_sin(geo["integration_points"])[..., 0, 0]
    fl = orc.integrate_local(orc.integrand_load(geo), geo["dx"])
    want_f = orc.assemble_linear(fl, tris, nv).reshape(-1)
    scale = max(float(np.abs(want).max()), 1e-300)
    tried = []
    for kernel in ("auto", "rings", "tiles", "gather", "atomic"):
        idx = torch.tensor(tris) if rng.random() < 0.5 else torch.tensor(tris.astype(np.int64))
        eng = AssemblyEngine(torch.tensor(verts, dtype=dtype), idx, idx, nv, 1, order)
        eng.kernel = kernel
        try:
            vals = eng.bilinear(alpha, beta)
        except NotImplementedError:
            assert kernel in ("rings", "tiles")  # a plan the mesh does not fit
            continue
        tried.append(kernel)
        fq_t = torch.tensor(fq_np, dtype=dtype)
        vals = _caller_values(eng, vals)
        assert np.abs(vals.cpu().double().numpy() - want).max() / scale <= tol, (seed, kernel, eng.kernel_name())
        vals2, f = eng.assemble_system(alpha, beta, fq_t)
        vals2 = _caller_values(eng, vals2)
        assert np.abs(vals2.cpu().double().numpy() - want).max() / scale <= tol, (seed, kernel, "system")
        assert scaled_error(f.cpu().double().numpy().reshape(-1), want_f) <= tol, (seed, kernel, "load")
        assert scaled_error(eng.load(fq_t).cpu().double().numpy().reshape(-1), want_f) <= tol
    assert "auto" in tried and "gather" in tried and "atomic" in tried


@pytest.mark.parametrize("seed", range(N_P2))
def test_random_p2_meshes_every_kernel_mode(seed, monkeypatch):
    from pytorch_fem_solver_amd import dofs, meshgen
    from pytorch_fem_solver_amd.basis.engine import AssemblyEngine

    rng = np.random.default_rng(2000 + seed)
    if seed % 3 == 2:  # the engine's own order of the edge DoFs (Morton order of the midpoints)
        monkeypatch.setenv("TFEM_RENUMBER", "1")
    if seed % 2 == 0:
        mesh = meshgen.unit_square(int(rng.integers(2, 50)), float(rng.uniform(0.0, 0.3)), seed)
    else:
        mesh = meshgen.delaunay_square(int(rng.integers(50, 3000)), seed)
    tris = mesh["triangles"].copy()
    if rng.random() < 0.4:  # holes: boundary edges inside, isolated vertices
        keep = rng.random(tris.shape[0]) >= rng.uniform(0.02, 0.2)
        if keep.sum() >= 1:
            tris = tris[keep]
    flip = rng.random(tris.shape[0]) < 0.3
    tris[flip] = tris[flip][:, [0, 2, 1]]
    edges, on_boundary = meshgen._edges_from_triangles(tris)
    conn6, xy, _ = dofs.p2_dofs_numpy(mesh["vertices"], tris, edges, on_boundary.astype(np.int32).reshape(-1, 1),
                                      mesh["vertex_markers"])
    order = int(rng.integers(2, 5))
    alpha, beta = float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.0, 3.0))
    geo = orc.geometry(mesh["vertices"][tris], 2, order)
    integrand = alpha * orc.integrand_stiffness(geo) + beta * orc.integrand_mass(geo)
    local = orc.integrate_local(integrand, geo["dx"])
    _, colind, slots = orc.csr_pattern(conn6, xy.shape[0])
    want = orc.assemble_csr_values(local, slots, colind.shape[0])
    loads = {}
    for kernel in ("auto", "gather", "atomic"):
        eng = AssemblyEngine(torch.tensor(mesh["vertices"]), torch.tensor(tris), torch.tensor(conn6),
                             xy.shape[0], 2, order)
        eng.kernel = kernel
        assert eng.renumbered == (seed % 3 == 2)
        vals = _caller_values(eng, eng.bilinear(alpha, beta))
        assert scaled_error(vals.cpu().numpy(), want) <= 1e-12, (seed, kernel, eng.kernel_name())
        fq = torch.tensor(np.cos(np.arange(tris.shape[0] * eng.n_quad)).reshape(tris.shape[0], -1))
        # the load vector of every mode (auto: row form over the P2 plan where it exists) against the oracle's
        local_f = orc.integrate_local(fq.cpu().numpy().reshape(-1, eng.n_quad, 1, 1) * geo["v"], geo["dx"])
        want_f = orc.assemble_linear(local_f, conn6, xy.shape[0]).reshape(-1)
        got_f = eng.load(fq).cpu().numpy().reshape(-1)
        assert scaled_error(got_f, want_f) <= 1e-12, (seed, kernel, "load")
        if kernel == "auto":  # per-DoF vectors leave the engine in the caller's numbering
            loads[eng.renumbered] = got_f
    if True in loads:  # the same vector as the engine gives without its renumbering
        monkeypatch.setenv("TFEM_RENUMBER", "0")
        plain = AssemblyEngine(torch.tensor(mesh["vertices"]), torch.tensor(tris), torch.tensor(conn6), xy.shape[0], 2, order)
        fq = torch.tensor(np.cos(np.arange(tris.shape[0] * plain.n_quad)).reshape(tris.shape[0], -1))
        assert scaled_error(loads[True], plain.load(fq).cpu().numpy()) <= 1e-12


@pytest.mark.parametrize("seed", range(16))
def test_random_meshes_through_the_public_api(seed, monkeypatch):
    """The reference's own forms through Basis.integrate_* on random meshes (CPU- or
    GPU-resident, dense or CSR result, recognised and generic callables) against the oracle;
    every third mesh through the engine's internal renumbering."""
    import math

    if seed % 3 == 2:
        monkeypatch.setenv("TFEM_RENUMBER", "1")

    import pytorch_fem_solver_amd as tfm
    from pytorch_fem_solver_amd import meshgen

    rng = np.random.default_rng(3000 + seed)
    mesh_np = (meshgen.unit_square(int(rng.integers(2, 40)), float(rng.uniform(0, 0.3)), seed) if seed % 2
               else meshgen.delaunay_square(int(rng.integers(40, 1500)), seed))
    if rng.random() < 0.5:
        order_v = rng.permutation(mesh_np["vertices"].shape[0]) if rng.random() < 0.5 else meshgen.morton_order(mesh_np["vertices"])
        mesh_np = meshgen.permute_mesh(mesh_np, vertex_order=order_v,
                                       triangle_order=rng.permutation(mesh_np["triangles"].shape[0]))
    on_cpu = rng.random() < 0.4
    torch.set_default_device("cpu" if on_cpu else "cuda")
    q = int(rng.integers(1, 5))
    verts, tris = mesh_np["vertices"], mesh_np["triangles"]
    nv = verts.shape[0]
    basis = tfm.Basis(tfm.MeshTri(triangulation=mesh_np), tfm.ElementTri(1, q))

    def rhs(x, y):
        return 2.0 * math.pi**2 * torch.sin(math.pi * x) * torch.sin(math.pi * y)

    def stiffness_mass(b):
        return b.v_grad @ b.v_grad.mT + b.v @ b.v.mT

    def opaque(b):  # the same form, unrecognisable to the tracer: generic route
        g = b.v_grad
        return torch.matmul(g, g.transpose(-1, -2)) + b.v @ b.v.mT

    def load(b):
        x, y = torch.split(b.integration_points, 1, dim=-1)
        return rhs(x, y) * b.v

    geo = orc.geometry(verts[tris], 1, q)
    local = orc.integrate_local(orc.integrand_stiffness_mass(geo), geo["dx"])
    want = orc.assemble_dense_bilinear(local, tris, nv)
    for form in (stiffness_mass, opaque):
        K = basis.integrate_bilinear_form(form)
        assert K.is_cuda != on_cpu
        assert scaled_error(K.cpu(), want) <= 1e-12, (seed, form.__name__)
    Kc = basis.integrate_bilinear_form(stiffness_mass, layout="csr")
    assert scaled_error(Kc.to_dense().cpu(), want) <= 1e-12
    fl = orc.integrate_local(orc.integrand_load(geo), geo["dx"])
    f = basis.integrate_linear_form(load)
    assert scaled_error(f.cpu(), orc.assemble_linear(fl, tris, nv)) <= 1e-12
    total = basis.integrate_functional(lambda b: rhs(*torch.split(b.integration_points, 1, dim=-1)))
    want_total = orc.integrate_functional(orc.source_sin_sin(geo["integration_points"]), geo["dx"])
    assert abs(float(total.sum()) - float(np.sum(want_total))) <= 1e-11 * abs(float(np.sum(want_total)))
