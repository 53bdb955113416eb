"""Parity of the HIP assembly path (through the C ABI) with the CPU oracle and with the
reference-generated golden fixtures.  Runs on a real MI355X only (-m gpu).

Tolerance: fp64, scaled max-abs and relative-Frobenius error <= 1e-12 (BASELINE.json
north_star); float32 <= 2e-6.
"""

import math

import numpy as np
import pytest
import torch

from conftest import load_golden, mesh_from_golden, rowwise_error, scaled_error
from oracle import assembly_oracle as orc

pytestmark = pytest.mark.gpu

TOL = 1e-12


@pytest.fixture(autouse=True)
def _gpu_defaults():
    assert torch.cuda.is_available()
    torch.set_default_dtype(torch.float64)
    torch.set_default_device("cuda")
    yield
    torch.set_default_device("cpu")
    torch.set_default_dtype(torch.float32)


def tf():
    import pytorch_fem_solver_amd

    return pytorch_fem_solver_amd


def stiffness(basis):
    return basis.v_grad @ basis.v_grad.mT


def stiffness_mass(basis):
    return basis.v_grad @ basis.v_grad.mT + basis.v @ basis.v.mT


def mass(basis):
    return basis.v @ basis.v.mT


def convection_x(basis):
    return basis.v @ basis.v_grad[..., [0]].mT


def rhs(x, y):
    return 2.0 * math.pi**2 * torch.sin(math.pi * x) * torch.sin(math.pi * y)


def load(basis):
    x, y = torch.split(basis.integration_points, 1, dim=-1)
    return rhs(x, y) * basis.v


def rhs_squared(basis):
    x, y = torch.split(basis.integration_points, 1, dim=-1)
    return rhs(x, y) ** 2


def grad_field(points):
    x, y = torch.split(points, 1, dim=-1)
    return torch.cat([torch.cos(3.0 * x) * y, x * x - torch.sin(2.0 * y)], dim=-1)


def weak_residual(basis, field):
    x, y = torch.split(basis.integration_points, 1, dim=-1)
    return rhs(x, y) * basis.v - (basis.v_grad @ field(basis.integration_points).mT)


def test_native_library_is_the_one_loaded():
    from pytorch_fem_solver_amd import _native

    lib = _native.load()
    assert lib.tfem_device_count() >= 1
    with open("/proc/self/maps") as maps:
        assert "libtfem_hip.so" in maps.read()


@pytest.mark.parametrize(
    "fixture,orders",
    [
        ("p1_square_n8.npz", (1, 2, 3, 4)),
        ("p1_square_n5_clockwise.npz", (3,)),
        ("p1_delaunay_170.npz", (3,)),
    ],
)
def test_p1_against_golden(fixture, orders):
    d = load_golden(fixture)
    mesh = tf().MeshTri(triangulation=mesh_from_golden(d))
    for order in orders:
        basis = tf().Basis(mesh, tf().ElementTri(polynomial_order=1, integration_order=order))
        tag = f"out_q{order}_"
        # geometry cache in the reference's shapes
        assert scaled_error(basis.v.cpu(), d[tag + "v"]) <= TOL
        assert scaled_error(basis.v_grad.cpu(), d[tag + "v_grad"]) <= TOL
        assert scaled_error(basis.integration_points.cpu(), d[tag + "integration_points"]) <= TOL
        assert scaled_error(basis._dx.cpu(), d[tag + "dx"]) <= TOL
        assert scaled_error(basis._inv_map_jacobian.cpu(), d[tag + "inv_map_jacobian"]) <= TOL
        # fused forms
        for name, form in (
            ("K_stiffness", stiffness),
            ("K_stiffness_mass", stiffness_mass),
            ("K_mass", mass),
            ("K_convection_x", convection_x),  # generic reduce+scatter, non-symmetric
        ):
            dense = basis.integrate_bilinear_form(form)
            assert dense.is_cuda and dense.shape == d[tag + name].shape
            assert scaled_error(dense.cpu(), d[tag + name]) <= TOL, (name, order)
            csr = basis.integrate_bilinear_form(form, layout="csr")
            assert scaled_error(csr.to_dense().cpu(), d[tag + name]) <= TOL
            # entry by entry against the scale of the entry's own row (rtol 1e-12 of north_star);
            # the pattern is the oracle's, from the connectivity alone
            rowptr, colind, _ = orc.csr_pattern(d["in_triangles"], d["in_vertices"].shape[0])
            assert np.array_equal(csr.crow_indices.cpu().numpy(), rowptr) and np.array_equal(csr.col_indices.cpu().numpy(), colind)
            rows = np.repeat(np.arange(rowptr.size - 1), np.diff(rowptr))
            assert rowwise_error(csr.values.cpu(), d[tag + name][rows, colind], rowptr) <= TOL, (name, order)
        f = basis.integrate_linear_form(load)
        assert f.shape == d[tag + "f_load"].shape
        assert scaled_error(f.cpu(), d[tag + "f_load"]) <= TOL
        fw = basis.integrate_linear_form(weak_residual, grad_field)  # generic linear path
        assert scaled_error(fw.cpu(), d[tag + "f_weak_residual"]) <= TOL
        fun = basis.integrate_functional(rhs_squared)
        assert fun.shape == d[tag + "functional_rhs2"].shape
        assert scaled_error(fun.cpu(), d[tag + "functional_rhs2"]) <= TOL
        # downstream: solve + interpolate (torch host code fed by the assembled operator)
        A = basis.integrate_bilinear_form(stiffness)
        u = basis.solve(A, basis.solution_tensor(), f)
        assert scaled_error(u.cpu(), d[tag + "u_h"]) <= 1e-10
        val, grad = basis.interpolate(basis, u)
        assert scaled_error(val.cpu(), d[tag + "interp_self_val"]) <= 1e-10
        assert scaled_error(grad.cpu(), d[tag + "interp_self_grad"]) <= 1e-10


def test_generic_path_equals_fused_path():
    d = load_golden("p1_square_n8.npz")
    mesh = tf().MeshTri(triangulation=mesh_from_golden(d))
    basis = tf().Basis(mesh, tf().ElementTri(1, 3))

    def opaque(b):  # same form, written so the tracer cannot recognise it
        g = b.v_grad
        return torch.matmul(g, g.transpose(-1, -2)) + b.v @ b.v.mT

    fused = basis.integrate_bilinear_form(stiffness_mass)
    generic = basis.integrate_bilinear_form(opaque)
    assert scaled_error(generic.cpu(), fused.cpu()) <= 1e-14


def test_cpu_resident_mesh_is_staged_through_the_gpu():
    """tests/test_assembly.py of the reference runs with CPU default tensors."""
    torch.set_default_device("cpu")
    d = load_golden("p1_delaunay_170.npz")
    mesh = tf().MeshTri(triangulation=mesh_from_golden(d))
    basis = tf().Basis(mesh, tf().ElementTri(1, 3))
    K = basis.integrate_bilinear_form(stiffness_mass)
    f = basis.integrate_linear_form(load)
    fun = basis.integrate_functional(rhs_squared)
    assert not K.is_cuda and not f.is_cuda and not fun.is_cuda
    assert scaled_error(K, d["out_q3_K_stiffness_mass"]) <= TOL
    assert scaled_error(f, d["out_q3_f_load"]) <= TOL
    assert scaled_error(fun, d["out_q3_functional_rhs2"]) <= TOL


def test_float32():
    torch.set_default_dtype(torch.float32)
    d = load_golden("p1_square_n6_float32.npz")
    mesh = tf().MeshTri(triangulation=mesh_from_golden(d))
    basis = tf().Basis(mesh, tf().ElementTri(1, 4))
    assert basis.v_grad.dtype == torch.float32
    assert scaled_error(basis.v_grad.cpu(), d["out_q4_v_grad"]) <= 2e-6
    K = basis.integrate_bilinear_form(stiffness)
    assert K.dtype == torch.float32
    assert scaled_error(K.cpu(), d["out_q4_K_stiffness"]) <= 2e-6
    f = basis.integrate_linear_form(load)
    assert scaled_error(f.cpu(), d["out_q4_f_load"]) <= 2e-6


@pytest.mark.parametrize("order", [2, 4])
def test_p2_global_against_golden(order):
    d = load_golden("p2_global_n4.npz")
    mesh = tf().MeshTri(triangulation=mesh_from_golden(d))
    basis = tf().Basis(mesh, tf().ElementTri(polynomial_order=2, integration_order=order))
    assert np.array_equal(basis._global_dofs4elements.cpu().numpy(), d["in_p2_connectivity"])
    tag = f"out_q{order}_"
    K = basis.integrate_bilinear_form(stiffness)
    assert scaled_error(K.cpu(), d[tag + "K_stiffness"]) <= TOL
    KM = basis.integrate_bilinear_form(stiffness_mass)
    assert scaled_error(KM.cpu(), d[tag + "K_stiffness_mass"]) <= TOL
    f = basis.integrate_linear_form(load)
    assert scaled_error(f.cpu(), d[tag + "f_load"]) <= TOL


@pytest.mark.parametrize("order", [2, 3, 4])
def test_p2_element_level_geometry(order):
    """P2 v_grad (N_T, Q, 6, 2) from the geometry kernel vs the reference element code."""
    d = load_golden("p2_element.npz")
    cells = d["in_cell_coordinates"]
    n = cells.shape[0]
    mesh_np = {
        "vertices": cells.reshape(-1, 2),
        "vertex_markers": np.zeros((3 * n, 1), dtype=np.int32),
        "triangles": np.arange(3 * n, dtype=np.int32).reshape(n, 3),
    }
    mesh = tf().MeshTri(triangulation=mesh_np)
    basis = tf().Basis(mesh, tf().ElementTri(polynomial_order=2, integration_order=order))
    tag = f"out_q{order}_"
    assert scaled_error(basis.v.cpu(), d[tag + "v"]) <= TOL
    assert scaled_error(basis.v_grad.cpu(), d[tag + "v_grad"]) <= TOL
    assert scaled_error(basis._dx.cpu(), d[tag + "dx"]) <= TOL


@pytest.mark.parametrize("fixture", ["fracture_L4.npz", "fracture_L3_jitter.npz"])
def test_fracture_example_pipeline(fixture):
    """Config 5: examples/example_fractures_fem.py:58-64,235-297 on the committed mesh."""
    d = load_golden(fixture)
    tri = mesh_from_golden(d)
    mesh = tf().FracturesTri(
        triangulations=[tri, tri], fractures_3d_data=torch.tensor(d["in_fractures_3d"])
    )
    V = tf().FractureBasis(mesh, tf().ElementTri(polynomial_order=1, integration_order=4))
    assert scaled_error(V.v_grad.cpu(), d["out_frac_v_grad"]) <= TOL
    assert scaled_error(V._dx.cpu(), d["out_frac_dx"]) <= TOL
    assert scaled_error(V.integration_points.cpu(), d["out_frac_integration_points"]) <= TOL
    assert scaled_error(V._inv_map_jacobian.cpu(), d["out_frac_inv_map_jacobian"]) <= TOL

    def frac_rhs(c):
        x, y, z = torch.split(c, 1, dim=-1)
        x1, _ = torch.split(x, 1, dim=0)
        y1, y2 = torch.split(y, 1, dim=0)
        _, z2 = torch.split(z, 1, dim=0)
        r1 = 6.0 * (y1 - y1**2) * torch.abs(x1) - 2.0 * (torch.abs(x1) ** 3 - torch.abs(x1))
        r2 = -6.0 * (y2 - y2**2) * torch.abs(z2) + 2.0 * (torch.abs(z2) ** 3 - torch.abs(z2))
        return torch.cat([r1, r2], dim=0)

    def exact(c):
        x, y, z = torch.split(c, 1, dim=-1)
        x1, _ = torch.split(x, 1, dim=0)
        y1, y2 = torch.split(y, 1, dim=0)
        _, z2 = torch.split(z, 1, dim=0)
        e1 = -y1 * (1 - y1) * torch.abs(x1) * (x1**2 - 1)
        e2 = y2 * (1 - y2) * torch.abs(z2) * (z2**2 - 1)
        return torch.cat([e1, e2], dim=0)

    A = V.integrate_bilinear_form(stiffness)
    b = V.integrate_linear_form(lambda basis: frac_rhs(basis.integration_points) * basis.v)
    fun = V.integrate_functional(lambda basis: exact(basis.integration_points) ** 2)
    assert scaled_error(A.cpu(), d["out_A"]) <= TOL
    assert scaled_error(b.cpu(), d["out_b"]) <= TOL
    assert scaled_error(fun.cpu(), d["out_functional_exact_sq"]) <= TOL
    u_h = V.solve(A, V.solution_tensor(), b)
    assert scaled_error(u_h.cpu(), d["out_u_h"]) <= 1e-10
    val, grad = V.interpolate(V, u_h)
    assert scaled_error(val.cpu(), d["out_interp_self_val"]) <= 1e-10
    assert scaled_error(grad.cpu(), d["out_interp_self_grad"]) <= 1e-10
    VE = tf().InteriorEdgesFractureBasis(mesh, tf().ElementLine(polynomial_order=1, integration_order=2))
    assert scaled_error(VE.integration_points.cpu(), d["out_edge_integration_points"]) <= TOL
    assert scaled_error(VE._dx.cpu(), d["out_edge_dx"]) <= TOL
    _, eg = V.interpolate(VE, u_h)
    n_E = mesh["interior_edges", "normals_3d"].unsqueeze(-2)
    plus, minus = torch.unbind(eg, dim=-4)
    jump = (plus * n_E).sum(-1) + (minus * -n_E).sum(-1)
    assert scaled_error(jump.cpu(), d["out_jump"]) <= 1e-9


def test_fracture_example_pipeline_m64():
    """Config 5 at SURVEY 8(d)'s larger size (m = 64: 2 x 16,384 cells, 16,705 DoFs): operator
    (compared on the reference's nonzero entries and by its total), load vector, solution and
    jump against the reference-generated fixture; the mesh is rebuilt from the generator's
    arguments and checked against the fixture's digest."""
    import hashlib

    from pytorch_fem_solver_amd import meshgen

    d = load_golden("fracture_L64.npz")
    tri = meshgen.fracture_rectangle(int(d["in_m"]), jitter=float(d["in_jitter"]), seed=int(d["in_seed"]))
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(tri["vertices"]).tobytes())
    h.update(np.ascontiguousarray(tri["triangles"]).tobytes())
    assert h.digest() == d["in_vertices_sha"].tobytes(), "the generator no longer builds the fixture's mesh"
    mesh = tf().FracturesTri(triangulations=[tri, tri], fractures_3d_data=torch.tensor(d["in_fractures_3d"]))
    V = tf().FractureBasis(mesh, tf().ElementTri(polynomial_order=1, integration_order=4))

    def frac_rhs(c):
        x, y, z = torch.split(c, 1, dim=-1)
        x1, _ = torch.split(x, 1, dim=0)
        y1, y2 = torch.split(y, 1, dim=0)
        _, z2 = torch.split(z, 1, dim=0)
        r1 = 6.0 * (y1 - y1**2) * torch.abs(x1) - 2.0 * (torch.abs(x1) ** 3 - torch.abs(x1))
        r2 = -6.0 * (y2 - y2**2) * torch.abs(z2) + 2.0 * (torch.abs(z2) ** 3 - torch.abs(z2))
        return torch.cat([r1, r2], dim=0)

    def exact(c):
        x, y, z = torch.split(c, 1, dim=-1)
        x1, _ = torch.split(x, 1, dim=0)
        y1, y2 = torch.split(y, 1, dim=0)
        _, z2 = torch.split(z, 1, dim=0)
        e1 = -y1 * (1 - y1) * torch.abs(x1) * (x1**2 - 1)
        e2 = y2 * (1 - y2) * torch.abs(z2) * (z2**2 - 1)
        return torch.cat([e1, e2], dim=0)

    A = V.integrate_bilinear_form(stiffness)
    assert tuple(A.shape) == tuple(int(x) for x in d["out_A_shape"])
    rows, cols = torch.tensor(d["out_A_rows"]).long(), torch.tensor(d["out_A_cols"]).long()
    want_vals = d["out_A_vals"]
    assert scaled_error(A[rows, cols].cpu(), want_vals) <= TOL
    # nothing outside the reference's nonzero entries beyond rounding (cancelled entries)
    assert abs(float(A.abs().sum()) - np.abs(want_vals).sum()) <= 1e-9 * np.abs(want_vals).sum()
    b = V.integrate_linear_form(lambda basis: frac_rhs(basis.integration_points) * basis.v)
    assert scaled_error(b.cpu(), d["out_b"]) <= TOL
    assert np.array_equal(V._basis_parameters["inner_dofs"].cpu().numpy().astype(np.int32).ravel(),
                          d["out_inner_dofs"].ravel())
    fun = V.integrate_functional(lambda basis: exact(basis.integration_points) ** 2)
    assert abs(float(fun.sum()) - float(d["out_functional_exact_sq_sum"])) <= 1e-12 * abs(float(d["out_functional_exact_sq_sum"]))
    u_h = V.solve(A, V.solution_tensor(), b)
    assert scaled_error(u_h.cpu(), d["out_u_h"]) <= 1e-9
    VE = tf().InteriorEdgesFractureBasis(mesh, tf().ElementLine(polynomial_order=1, integration_order=2))
    _, eg = V.interpolate(VE, u_h)
    n_E = mesh["interior_edges", "normals_3d"].unsqueeze(-2)
    plus, minus = torch.unbind(eg, dim=-4)
    jump = (plus * n_E).sum(-1) + (minus * -n_E).sum(-1)
    assert scaled_error(jump.cpu(), d["out_jump"]) <= 1e-8


def test_interior_edges_basis_and_interpolation():
    d = load_golden("mesh_topology_n4.npz")
    mesh = tf().MeshTri(triangulation=mesh_from_golden(d))
    basis = tf().Basis(mesh, tf().ElementTri(1, 3))
    edge_basis = tf().InteriorEdgesBasis(mesh, tf().ElementLine(1, 2))
    assert scaled_error(edge_basis.integration_points.cpu(), d["out_edge_integration_points"]) <= TOL
    assert scaled_error(edge_basis._dx.cpu(), d["out_edge_dx"]) <= TOL
    u = torch.tensor(d["in_vertex_field"])
    val, grad = basis.interpolate(edge_basis, u)
    assert scaled_error(val.cpu(), d["out_interp_edges_val"]) <= 1e-11
    assert scaled_error(grad.cpu(), d["out_interp_edges_grad"]) <= 1e-11


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_edge_interpolation_kernel_against_the_torch_expressions(dtype):
    """tfem_edge_interpolate_p1 (the tensor branch of Basis.interpolate on interior edges,
    basis.py:98-177) against the reference's expression sequence evaluated by torch -- the
    path a vector with autograd history takes -- on a jittered and a Delaunay mesh."""
    from pytorch_fem_solver_amd import meshgen

    tol = 1e-12 if dtype == torch.float64 else 2e-5
    for mesh_np in (meshgen.unit_square(40, 0.25, 3), meshgen.delaunay_square(3000, seed=2)):
        mesh_np = dict(mesh_np, vertices=mesh_np["vertices"].astype(np.float64 if dtype == torch.float64 else np.float32))
        # without "neighbors" the edge -> cells table is matched edge by edge; with it the
        # reference pairs the SORTED cell pairs with the edge list (abstract_mesh.py:207-230,
        # SURVEY.md appendix C-3; the golden fixture of the test above pins that behaviour)
        mesh_np.pop("neighbors", None)
        mesh = tf().MeshTri(triangulation=mesh_np)
        basis = tf().Basis(mesh, tf().ElementTri(1, 3))
        edge_basis = tf().InteriorEdgesBasis(mesh, tf().ElementLine(1, 2))
        n_edges = edge_basis.integration_points.shape[0]
        n_points = edge_basis.integration_points.shape[-2]
        xy = torch.as_tensor(mesh_np["vertices"], dtype=dtype)
        u = (torch.sin(3 * xy[:, :1]) * torch.cos(2 * xy[:, 1:]) + xy[:, :1] ** 2).to(dtype)
        val, grad = basis.interpolate(edge_basis, u)
        assert val.shape == (n_edges, 2, n_points, 1, 1) and grad.shape == (n_edges, 2, 1, 1, 2)
        reference_path = tf().Basis(mesh, tf().ElementTri(1, 3))
        reference_path.edge_kernel = False  # the reference's expression sequence, run by torch
        want_val, want_grad = reference_path.interpolate(edge_basis, u)
        assert want_val.shape == val.shape and want_grad.shape == grad.shape
        assert scaled_error(val.cpu(), want_val.cpu()) <= tol
        assert scaled_error(grad.cpu(), want_grad.cpu()) <= tol
        # derivative in the nodal values: tfem_edge_interpolate_p1_backward against autograd
        # through the expressions (a jump-like loss: weighted values + squared gradients)
        weights = torch.linspace(0.5, 1.5, val.numel()).reshape(val.shape).to(dtype)
        grads = []
        for b in (basis, reference_path):
            u_var = u.clone().requires_grad_(True)
            v_, g_ = b.interpolate(edge_basis, u_var)
            ((weights * v_).sum() + (g_ ** 2).sum()).backward()
            grads.append(u_var.grad.detach().cpu())
        assert grads[0].shape == u.shape
        assert scaled_error(grads[0], grads[1]) <= (1e-12 if dtype == torch.float64 else 1e-4)
        if dtype == torch.float64:  # and against the numpy oracle (pinned by the reference's fixture)
            o_val, o_grad = orc.edge_interpolate_p1(
                mesh_np["vertices"], mesh_np["triangles"], mesh["interior_edges", "cells"].cpu().numpy(),
                edge_basis.integration_points.cpu().numpy(), u.cpu().numpy())
            assert scaled_error(val.cpu(), o_val) <= tol and scaled_error(grad.cpu(), o_grad) <= tol
        # a P1 field is continuous: both sides agree on the edge; the tangential derivative too
        assert (val[:, 0] - val[:, 1]).abs().max().item() <= 50 * tol * val.abs().max().item()
        # the closures of the function branch reach the same launch
        interp, interp_grad = basis.interpolate(edge_basis)
        field = lambda nodes: torch.sin(3 * nodes[:, :1]) * torch.cos(2 * nodes[:, 1:]) + nodes[:, :1] ** 2  # noqa: E731
        assert scaled_error(interp(field).cpu(), val.cpu()) <= 10 * tol
        assert scaled_error(interp_grad(field).cpu(), grad.cpu()) <= 10 * tol


def test_edge_interpolation_adjoint_in_row_form_is_reproducible_and_equals_the_atomic_one():
    """tfem_edge_interpolate_p1_backward_rows (one lane per vertex, fixed summation order) against
    tfem_edge_interpolate_p1_backward (hardware atomics) and bit for bit against itself."""
    from pytorch_fem_solver_amd import meshgen

    mesh_np = meshgen.delaunay_square(20000, seed=4)
    mesh_np.pop("neighbors", None)
    mesh = tf().MeshTri(triangulation=mesh_np)
    basis = tf().Basis(mesh, tf().ElementTri(1, 3))
    edge_basis = tf().InteriorEdgesBasis(mesh, tf().ElementLine(1, 2))
    eng = basis._engine
    pts = edge_basis.integration_points
    n_edges, n_points = pts.shape[0], pts.shape[-2]
    cells, points, _, _ = eng._edge_inputs(mesh["interior_edges", "cells"], pts.reshape(n_edges, n_points, 2))
    incidence = eng.edge_incidence(cells)
    # every (side, local index) appears exactly once, under its vertex
    conn = torch.tensor(mesh_np["triangles"]).long()
    side, loc = incidence[1] >> 2, incidence[1] & 3
    owner = torch.repeat_interleave(torch.arange(eng.coords_per_mesh), incidence[0][1:] - incidence[0][:-1])
    assert incidence[1].numel() == 6 * n_edges and torch.equal(conn[cells.reshape(-1)[side], loc], owner)
    assert torch.equal(torch.sort(incidence[1]).values, torch.sort(4 * torch.arange(2 * n_edges)[:, None] + torch.arange(3)).values.reshape(-1).sort().values)
    gen = torch.Generator(device="cuda").manual_seed(1)
    g_value = torch.randn(n_edges, 2, n_points, generator=gen)
    g_grad = torch.randn(n_edges, 2, 2, generator=gen)
    rows = eng.edge_interpolate_backward(cells, points, g_value, g_grad, prepared=True, incidence=incidence)
    again = eng.edge_interpolate_backward(cells, points, g_value, g_grad, prepared=True, incidence=incidence)
    atomic = eng.edge_interpolate_backward(cells, points, g_value, g_grad, prepared=True)
    assert torch.equal(rows, again)
    assert scaled_error(rows.cpu(), atomic.cpu()) <= 1e-13
    # the public path uses the row form: two backward passes give identical bits
    u = torch.randn(eng.coords_per_mesh, 1, generator=gen)
    out = []
    for _ in range(2):
        u_var = u.clone().requires_grad_(True)
        v_, g_ = basis.interpolate(edge_basis, u_var)
        ((v_ ** 2).sum() + (g_ ** 2).sum()).backward()
        out.append(u_var.grad.clone())
    assert torch.equal(out[0], out[1])


@pytest.mark.parametrize("fixture", ["fracture_L4.npz", "fracture_L3_jitter.npz"])
def test_fracture_edge_interpolation_kernel_against_the_reference(fixture):
    """tfem_edge_interpolate_p1_fracture (FractureBasis.interpolate on the interior edges,
    fracture_basis.py:225-272, with the reference's per-fracture ids indexing the global vector)
    against the reference-generated fixture and against the torch expression sequence."""
    d = load_golden(fixture)
    tri = mesh_from_golden(d)
    mesh = tf().FracturesTri(triangulations=[tri, tri], fractures_3d_data=torch.tensor(d["in_fractures_3d"]))
    V = tf().FractureBasis(mesh, tf().ElementTri(1, 4))
    VE = tf().InteriorEdgesFractureBasis(mesh, tf().ElementLine(1, 2))
    u_h = torch.tensor(d["out_u_h"])
    calls = []
    original = V._engine.edge_interpolate_fracture
    V._engine.edge_interpolate_fracture = lambda *a: calls.append(1) or original(*a)
    val, grad = V.interpolate(VE, u_h)
    assert calls == [1]
    assert val.shape == d["out_interp_edges_val"].shape and grad.shape == d["out_interp_edges_grad"].shape
    assert scaled_error(val.cpu(), d["out_interp_edges_val"]) <= TOL
    assert scaled_error(grad.cpu(), d["out_interp_edges_grad"]) <= TOL
    V.edge_kernel = False  # the expression sequence, run by torch
    want_val, want_grad = V.interpolate(VE, u_h)
    assert calls == [1]
    assert scaled_error(val.cpu(), want_val.cpu()) <= TOL and scaled_error(grad.cpu(), want_grad.cpu()) <= TOL
    n_E = mesh["interior_edges", "normals_3d"].unsqueeze(-2)
    plus, minus = torch.unbind(grad, dim=-4)  # example_fractures_fem.py:295-297
    jump = (plus * n_E).sum(-1) + (minus * -n_E).sum(-1)
    assert scaled_error(jump.cpu(), d["out_jump"]) <= TOL


# ---------------------------------------------------------------------------------------
# oracle comparisons on seeded meshes + size-independent properties at full size
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,order", [(71, 3), (200, 4)])
def test_p1_csr_against_oracle(n, order):
    from pytorch_fem_solver_amd import meshgen

    mesh_np = meshgen.unit_square(n, 0.25, 0)
    mesh = tf().MeshTri(triangulation=mesh_np)
    basis = tf().Basis(mesh, tf().ElementTri(1, order))
    K = basis.integrate_bilinear_form(stiffness_mass, layout="csr")
    nv = mesh_np["vertices"].shape[0]
    local, _ = orc.p1_assemble(mesh_np["vertices"], mesh_np["triangles"], order, "stiffness_mass")
    rowptr, colind, slots = orc.csr_pattern(mesh_np["triangles"], nv)
    want = orc.assemble_csr_values(local, slots, colind.shape[0])
    assert np.array_equal(K.crow_indices.cpu().numpy(), rowptr)
    assert np.array_equal(K.col_indices.cpu().numpy(), colind)
    assert scaled_error(K.values.cpu(), want) <= TOL
    f = basis.integrate_linear_form(load)
    fl, _ = orc.p1_assemble(mesh_np["vertices"], mesh_np["triangles"], order, "load")
    assert scaled_error(f.cpu(), orc.assemble_linear(fl, mesh_np["triangles"], nv)) <= TOL


@pytest.mark.parametrize("mesh_kind", ["structured", "delaunay", "delaunay_morton"])
def test_tile_kernel_and_atomic_kernel_agree_with_oracle(mesh_kind):
    from pytorch_fem_solver_amd import meshgen

    if mesh_kind == "structured":
        mesh_np = meshgen.unit_square(150, 0.25, 3)
    else:
        mesh_np = meshgen.delaunay_square(20000, 5)
        if mesh_kind == "delaunay_morton":
            mesh_np = meshgen.permute_mesh(mesh_np, vertex_order=meshgen.morton_order(mesh_np["vertices"]))
    nv = mesh_np["vertices"].shape[0]
    local, _ = orc.p1_assemble(mesh_np["vertices"], mesh_np["triangles"], 3, "stiffness_mass")
    _, colind, slots = orc.csr_pattern(mesh_np["triangles"], nv)
    want = orc.assemble_csr_values(local, slots, colind.shape[0])
    got = {}
    names = {"rings": "k_p1_rings", "tiles": "k_p1_tiles_pipe", "atomic": "k_p1_bilinear_atomic"}
    for kernel in ("rings", "tiles", "atomic"):
        basis = tf().Basis(tf().MeshTri(mesh_np), tf().ElementTri(1, 3))
        basis._engine.kernel = kernel
        K = basis.integrate_bilinear_form(stiffness_mass, layout="csr")
        assert basis._engine.kernel_name() == names[kernel]
        assert scaled_error(K.values.cpu(), want) <= TOL, kernel
        got[kernel] = K.values
    assert scaled_error(got["tiles"].cpu(), got["atomic"].cpu()) <= 1e-14
    assert scaled_error(got["rings"].cpu(), got["atomic"].cpu()) <= 1e-13


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("form", ["stiffness", "stiffness_mass", "mass"])
def test_ring_kernel_mixed_orientation_and_open_fans(dtype, form):
    """The row-form kernel on a mesh whose elements are stored with mixed orientation (the
    reference integrates with the signed determinant) and on a bow-tie mesh (two fans
    meeting in one vertex, an isolated vertex): compared with the oracle entry by entry."""
    from pytorch_fem_solver_amd import meshgen

    torch.set_default_dtype(dtype)
    tol = TOL if dtype == torch.float64 else 5e-6
    mesh_np = meshgen.unit_square(90, 0.25, 4)
    tri = mesh_np["triangles"].copy()
    flip = np.random.default_rng(5).random(tri.shape[0]) < 0.4
    tri[flip] = tri[flip][:, [0, 2, 1]]
    mesh_np["triangles"] = tri
    callable_ = {"stiffness": stiffness, "stiffness_mass": stiffness_mass, "mass": mass}[form]
    bow = {
        "vertices": np.array([[0, 0], [1, 0], [1, 1], [-1, 0], [-1, -1], [5, 5], [0.3, 1.2]], dtype=np.float64),
        "triangles": np.array([[0, 1, 2], [0, 3, 4], [2, 6, 0]], dtype=np.int32),
    }
    for name, m in (("mixed", mesh_np), ("bow", bow)):
        verts, tris = m["vertices"], m["triangles"]
        nv = verts.shape[0]
        if dtype == torch.float32:
            verts = verts.astype(np.float32)
        local, _ = orc.p1_assemble(verts, tris, 3, form)
        _, colind, slots = orc.csr_pattern(tris, nv)
        want = orc.assemble_csr_values(local, slots, colind.shape[0])
        if name == "mixed":
            basis = tf().Basis(tf().MeshTri(m), tf().ElementTri(1, 3))
            basis._engine.kernel = "rings"
            K = basis.integrate_bilinear_form(callable_, layout="csr")
            assert basis._engine.kernel_name() == "k_p1_rings"
            got = K.values
        else:
            from pytorch_fem_solver_amd.basis.engine import AssemblyEngine

            eng = AssemblyEngine(torch.tensor(verts, dtype=dtype), torch.tensor(tris), torch.tensor(tris), nv, 1, 3)
            eng.kernel = "rings"
            ab = {"stiffness": (1.0, 0.0), "stiffness_mass": (1.0, 1.0), "mass": (0.0, 1.0)}[form]
            got = eng.bilinear(*ab)
        assert scaled_error(got.cpu().double(), want) <= tol, (name, form)


@pytest.mark.parametrize("kernel", ["rings", "tiles", "auto_morton"])
@pytest.mark.parametrize("order", [1, 2, 3, 4])
def test_fused_system_launch_matches_separate_forms(order, kernel):
    """engine.assemble_system = ONE launch for K and f (bench.py's step): the ring kernel (Delaunay
    mesh: 15-slot records; native numbering = Z-order tiles with one output run per row, Morton
    numbering = consecutive-vertex tiles, the plan the engine picks by itself) and the tile
    kernel."""
    from pytorch_fem_solver_amd import meshgen

    mesh_np = meshgen.delaunay_square(6000, 11)
    if kernel == "auto_morton":
        mesh_np = meshgen.permute_mesh(mesh_np, vertex_order=meshgen.morton_order(mesh_np["vertices"]))
    nv = mesh_np["vertices"].shape[0]
    basis = tf().Basis(tf().MeshTri(mesh_np), tf().ElementTri(1, order))
    if kernel != "auto_morton":
        basis._engine.kernel = kernel
    x, y = torch.split(basis.integration_points, 1, dim=-1)
    fq = rhs(x, y).reshape(-1, basis._engine.n_quad)
    vals, f = basis._engine.assemble_system(1.0, 1.0, fq)
    assert basis._engine.kernel_name() == ("k_p1_tiles_pipe" if kernel == "tiles" else "k_p1_rings")
    if kernel == "auto_morton":
        assert basis._engine.ring_plan()["chunked"]
    local, _ = orc.p1_assemble(mesh_np["vertices"], mesh_np["triangles"], order, "stiffness_mass")
    _, colind, slots = orc.csr_pattern(mesh_np["triangles"], nv)
    assert scaled_error(vals.cpu(), orc.assemble_csr_values(local, slots, colind.shape[0])) <= TOL
    fl, _ = orc.p1_assemble(mesh_np["vertices"], mesh_np["triangles"], order, "load")
    assert scaled_error(f.cpu().reshape(-1, 1), orc.assemble_linear(fl, mesh_np["triangles"], nv)) <= TOL
    # preallocated result buffers (pipelines that rotate buffers, bench.py at N > 1)
    out = (torch.full_like(vals, float("nan")), torch.full_like(f.view(-1), float("nan")))
    vals2, f2 = basis._engine.assemble_system(1.0, 1.0, fq, out=out)
    assert vals2.data_ptr() == out[0].data_ptr() and f2.data_ptr() == out[1].data_ptr()
    # bit for bit the same from the row form; the tile kernel's LDS atomics add in varying order, and
    # so does this launch on a ring plan with long rows (TFEM_RING_LONG=1; source VALUES then take the
    # tile kernel: only a source PROGRAM runs inside the ring launch there)
    fq_rings = kernel != "tiles" and basis._engine.ring_plan()["fq_ok"]
    assert fq_rings == (kernel != "tiles" and int(basis._engine.ring_plan()["layout"][23]) == 0)
    assert torch.equal(vals2.view(-1), vals.view(-1)) or not fq_rings
    assert scaled_error(vals2.cpu(), vals.cpu()) <= 1e-14 and scaled_error(f2.cpu().view(-1), f.cpu().view(-1)) <= 1e-14
    with pytest.raises(ValueError):
        basis._engine.assemble_system(1.0, 1.0, fq, out=(out[0][:-1], out[1]))
    # the load-only launch and the strict-order atomic kernels agree too
    f_only = basis._engine.load(fq)
    assert scaled_error(f_only.cpu(), f.cpu()) <= 1e-14
    basis2 = tf().Basis(tf().MeshTri(mesh_np), tf().ElementTri(1, order))
    basis2._engine.kernel = "atomic"
    assert scaled_error(basis2._engine.load(fq).cpu(), f.cpu()) <= 1e-13


@pytest.mark.parametrize("numbering", ["morton", "native"])
def test_ring_plan_variant_with_long_rows(numbering, monkeypatch):
    """TFEM_RING_LONG=1: 4-dword records in the tiles, the vertices with 8 .. 15 neighbours through
    k_p1_long_rows, holes in the tile kernel's output runs (consecutive-vertex tiles) / its run
    detection (Z-order tiles).  Not the default (it is slower, profiles/r02_delaunay_long_rows.log);
    kept correct."""
    from pytorch_fem_solver_amd import meshgen

    monkeypatch.setenv("TFEM_RING_LONG", "1")
    mesh_np = meshgen.delaunay_square(7000, 21)
    if numbering == "morton":
        mesh_np = meshgen.permute_mesh(mesh_np, vertex_order=meshgen.morton_order(mesh_np["vertices"]))
    nv = mesh_np["vertices"].shape[0]
    basis = tf().Basis(tf().MeshTri(mesh_np), tf().ElementTri(1, 3))
    eng = basis._engine
    eng.kernel = "rings"
    plan = eng.ring_plan()
    assert int(plan["layout"][6]) == 7 and int(plan["layout"][23]) > 100 and plan["chunked"] == (numbering == "morton")
    _, colind, slots = orc.csr_pattern(mesh_np["triangles"], nv)
    for form, ab in (("stiffness", (1.0, 0.0)), ("stiffness_mass", (1.0, 1.0)), ("mass", (0.0, 1.0))):
        local, _ = orc.p1_assemble(mesh_np["vertices"], mesh_np["triangles"], 3, form)
        want = orc.assemble_csr_values(local, slots, colind.shape[0])
        got = torch.full((colind.shape[0],), float("nan"))
        eng._assemble_rings(*ab, out=(got, None))
        assert scaled_error(got.cpu(), want) <= TOL, form
    f = basis.integrate_linear_form(load)  # the source program inside the launch
    fl, _ = orc.p1_assemble(mesh_np["vertices"], mesh_np["triangles"], 3, "load")
    assert scaled_error(f.cpu(), orc.assemble_linear(fl, mesh_np["triangles"], nv)) <= TOL
    assert not plan["fq_ok"]  # source VALUES do not take this plan
    # a launch over a PART of the tiles would leave long rows unwritten (they have one launch over
    # all of them): refused, by the engine and by the library
    with pytest.raises(NotImplementedError):
        eng.tile_range("rest")
    got = torch.zeros(colind.shape[0])
    status = eng.lib.tfem_p1_assemble_rings_range(
        eng._inputs()["coords"].data_ptr(), 8, nv, 3, 1.0, 0.0, plan["blob"].data_ptr(),
        plan["layout"].ctypes.data, got.data_ptr(), colind.shape[0], None, None, eng.n_elems, None,
        0, max(1, plan["n_tiles"] // 2), torch.cuda.current_stream().cuda_stream)
    assert status != 0 and "long rows" in eng.lib.tfem_last_error().decode()


def test_engine_picks_the_tile_kernel_for_a_numbering_without_locality():
    from pytorch_fem_solver_amd import meshgen

    mesh_np = meshgen.delaunay_square(6000, 11)  # scipy's point order: about one output run per row
    basis = tf().Basis(tf().MeshTri(mesh_np), tf().ElementTri(1, 3))
    assert basis._engine.kernel_name() == "k_p1_tiles_pipe"
    assert basis._engine.ring_plan()["rows_per_run"] < 8.0
    basis = tf().Basis(tf().MeshTri(meshgen.unit_square(50, 0.25, 0)), tf().ElementTri(1, 3))
    assert basis._engine.kernel_name() == "k_p1_rings" and basis._engine.ring_plan()["chunked"]


def test_edge_cases_empty_and_single_element():
    mesh_np = {
        "vertices": np.array([[0.0, 0.0], [2.0, 0.0], [0.0, 1.0]]),
        "vertex_markers": np.ones((3, 1), dtype=np.int32),
        "triangles": np.array([[0, 1, 2]], dtype=np.int32),
        "edges": np.array([[0, 1], [1, 2], [0, 2]], dtype=np.int32),
        "edge_markers": np.ones((3, 1), dtype=np.int32),
    }
    basis = tf().Basis(tf().MeshTri(mesh_np), tf().ElementTri(1, 2))
    K = basis.integrate_bilinear_form(stiffness)
    want = orc.p1_stiffness_closed_form(mesh_np["vertices"], mesh_np["triangles"])[0]
    # local[i, j] -> A[conn[j], conn[i]]; symmetric here
    assert scaled_error(K.cpu(), want) <= TOL


def test_unsupported_orders_raise_like_the_reference():
    with pytest.raises(NotImplementedError):
        tf().ElementTri(1, 5)
    d = load_golden("p1_square_n8.npz")
    mesh = tf().MeshTri(triangulation=mesh_from_golden(d))
    with pytest.raises(NotImplementedError):
        tf().Basis(mesh, tf().ElementTri(3, 2))


def test_full_size_properties_1e6():
    """C2 (999,698 elements): properties that need no oracle -- row sums of the stiffness
    operator vanish, sum(M) = |Omega| = 1, sum(f) = quadrature of the source, symmetry of
    the CSR values, run-to-run agreement."""
    from pytorch_fem_solver_amd import meshgen

    mesh_np = meshgen.unit_square(707, 0.25, 0)
    mesh = tf().MeshTri(triangulation=mesh_np)
    basis = tf().Basis(mesh, tf().ElementTri(1, 3))
    K = basis.integrate_bilinear_form(stiffness, layout="csr")
    assert K.nnz == 3503186
    ones = torch.ones(K.shape[0], 1)
    scale = K.values.abs().max().item()
    assert (K.matvec(ones).abs().max().item()) <= 1e-12 * scale * 8
    M = basis.integrate_bilinear_form(mass, layout="csr")
    assert abs(M.values.sum().item() - 1.0) <= 1e-12
    Kt = K.to_sparse_csr().to_sparse_coo().t().coalesce()
    Kc = K.to_sparse_csr().to_sparse_coo().coalesce()
    assert torch.equal(Kt.indices(), Kc.indices())
    assert (Kt.values() - Kc.values()).abs().max().item() <= 1e-12 * scale
    f = basis.integrate_linear_form(load)
    total = basis.integrate_functional(lambda b: rhs(*torch.split(b.integration_points, 1, dim=-1)))
    assert abs(f.sum().item() - total.sum().item()) <= 1e-11 * abs(total.sum().item())
    K2 = basis.integrate_bilinear_form(stiffness, layout="csr")
    assert (K2.values - K.values).abs().max().item() <= 1e-13 * scale
    # against the oracle on a strided sample of elements (local blocks, closed form)
    sample = np.arange(0, mesh_np["triangles"].shape[0], 997)
    local, _ = orc.p1_assemble(mesh_np["vertices"], mesh_np["triangles"][sample], 3, "stiffness")
    closed = orc.p1_stiffness_closed_form(mesh_np["vertices"], mesh_np["triangles"][sample])
    assert scaled_error(local, closed) <= 1e-12


@pytest.mark.parametrize("n", [1000, 2236])
def test_bench_workload_against_c_oracle_at_full_size(n):
    """The bench's own launch (fused K + f, order 3) compared entry by entry with the
    C/OpenMP oracle at 2e6 and at the full 9,999,392 elements."""
    import __graft_entry__ as ge

    ge.build_oracle()
    from oracle import c_oracle
    from pytorch_fem_solver_amd import meshgen

    mesh_np = meshgen.unit_square(n, 0.25, 0)
    verts, tris = mesh_np["vertices"], mesh_np["triangles"]
    nv = verts.shape[0]
    basis = tf().Basis(tf().MeshTri(mesh_np), tf().ElementTri(1, 3))
    eng = basis._engine
    pts = c_oracle.points(verts, tris, 3)
    fq_np = orc.source_sin_sin(pts)[..., 0]
    vals, f = eng.assemble_system(1.0, 0.0, torch.tensor(fq_np))
    assert eng.kernel_name() == "k_p1_rings"
    # the CSR pattern and the element -> entry map the expected values go through are the ORACLE's
    # (numpy, from the connectivity alone), and the product's pattern has to be that pattern
    rowptr, colind, slots = orc.csr_pattern(tris, nv)
    got_rowptr, got_colind, _ = (t.cpu().numpy() for t in eng.csr_structure())
    assert np.array_equal(got_rowptr, rowptr) and np.array_equal(got_colind, colind)
    k_local, f_local = c_oracle.p1_local(verts, tris, 3, 1.0, 0.0, fq_np)
    want_vals = c_oracle.scatter_csr(k_local, slots, colind.shape[0])
    want_f = c_oracle.scatter_vector(f_local, tris, nv)
    f_scale = c_oracle.scatter_vector(np.abs(f_local), tris, nv)  # what every entry is a sum of
    assert scaled_error(vals.cpu(), want_vals) <= TOL
    assert scaled_error(f.cpu(), want_f) <= TOL
    # entry by entry, every entry against the scale of its own row (north_star: rtol 1e-12)
    assert rowwise_error(vals.cpu(), want_vals, rowptr) <= TOL
    assert rowwise_error(f.cpu(), want_f, scale=f_scale) <= TOL
    # K alone (row-form kernel, the launch the roofline target is quoted on)
    k_alone = eng.bilinear(1.0, 0.0).cpu()
    assert scaled_error(k_alone, want_vals) <= TOL and rowwise_error(k_alone, want_vals, rowptr) <= TOL
    # the integration points the HIP geometry kernel hands to user callables
    assert scaled_error(basis.integration_points.cpu().reshape(-1, 4, 2), pts) <= TOL


def test_p2_config3_properties_1e6():
    """Config 3 (P2, 6x6 blocks, ~1e6 elements): the reference has no global P2 assembly
    to compare with (basis.py:50-51), so check what needs no oracle -- constants lie in the
    kernel of the stiffness operator, sum(M) = |Omega|, symmetry -- plus a strided sample of
    local blocks against the oracle's element-level P2 (which IS pinned to the reference)."""
    from pytorch_fem_solver_amd import meshgen

    mesh_np = meshgen.unit_square(707, 0.25, 0)
    basis = tf().Basis(tf().MeshTri(mesh_np), tf().ElementTri(polynomial_order=2, integration_order=2))
    K = basis.integrate_bilinear_form(stiffness, layout="csr")
    n = K.shape[0]
    assert n == 708 * 708 + mesh_np["edges"].shape[0]
    scale = K.values.abs().max().item()
    assert K.matvec(torch.ones(n, 1)).abs().max().item() <= 1e-11 * scale
    basis4 = tf().Basis(tf().MeshTri(mesh_np), tf().ElementTri(polynomial_order=2, integration_order=4))
    M = basis4.integrate_bilinear_form(mass, layout="csr")
    assert abs(M.values.sum().item() - 1.0) <= 1e-11
    Kc = K.to_sparse_csr().to_sparse_coo().coalesce()
    Kt = K.to_sparse_csr().to_sparse_coo().t().coalesce()
    assert (Kc.values() - Kt.values()).abs().max().item() <= 1e-12 * scale
    # entries of a few elements that do not share DoFs: the assembled diagonal block of an
    # interior edge DoF equals the sum of its two elements' local entries (oracle)
    sample = np.arange(0, mesh_np["triangles"].shape[0], 50021)
    cells = mesh_np["vertices"][mesh_np["triangles"][sample].astype(np.int64)]
    geo = orc.geometry(cells, 2, 2)
    local = orc.integrate_local(orc.integrand_stiffness(geo), geo["dx"])
    conn6 = basis._global_dofs4elements.cpu().numpy()[sample]
    # the engine keeps P2 edge DoFs in an order of its own (Morton order of the edge midpoints): the
    # arrays in the CALLER's numbering
    assert basis._engine.renumbered and K.perm is not None
    plain = K.caller_numbering()
    crow, col, val = (plain.crow_indices.cpu().numpy(), plain.col_indices.cpu().numpy(), plain.values.cpu().numpy())
    o_rowptr, o_colind, _ = orc.csr_pattern(basis._global_dofs4elements.cpu().numpy(), n)
    assert np.array_equal(crow, o_rowptr) and np.array_equal(col, o_colind)  # the oracle's pattern
    for e in range(sample.shape[0]):
        # vertex-vertex entries get contributions from other elements too; the entry between
        # the element's two "own" edge DoFs of edges 0 and 1 belongs to this element alone
        a, b = conn6[e, 3], conn6[e, 4]
        row = slice(crow[a], crow[a + 1])
        got = val[row][col[row] == b][0]
        assert abs(got - local[e, 4, 3]) <= 1e-12 * scale


def test_linear_form_is_differentiable_like_the_reference():
    """the VPINN training step differentiates through integrate_linear_form
    (examples/example_weak.py:132-152).  Gradient of r^T r with r = integrate_linear_form(
    f v - v_grad @ g_theta^T) w.r.t. theta, against the same expression in plain torch."""
    d = load_golden("p1_square_n8.npz")
    mesh = tf().MeshTri(triangulation=mesh_from_golden(d))
    basis = tf().Basis(mesh, tf().ElementTri(1, 4))
    theta = torch.tensor([0.7, -1.3], requires_grad=True)

    def field(points, th):
        x, y = torch.split(points, 1, dim=-1)
        return torch.cat([th[0] * torch.cos(3.0 * x) * y, th[1] * x * x - torch.sin(2.0 * y)], dim=-1)

    def residual(b, th):
        x, y = torch.split(b.integration_points, 1, dim=-1)
        return rhs(x, y) * b.v - (b.v_grad @ field(b.integration_points, th).mT)

    r = basis.integrate_linear_form(residual, theta)
    loss = (r * r).sum()
    (g_hip,) = torch.autograd.grad(loss, theta)
    # plain-torch evaluation of the same reference expressions (abstract_basis.py:95-112)
    theta2 = theta.detach().clone().requires_grad_(True)
    integrand = (residual(basis, theta2) * basis._dx).sum(-3)
    ref = torch.zeros(basis._basis_parameters["linear_form_shape"]).index_put(
        (basis._global_dofs4elements.reshape(-1).long(),), integrand.reshape(-1, 1), accumulate=True
    )
    loss2 = (ref * ref).sum()
    (g_ref,) = torch.autograd.grad(loss2, theta2)
    assert scaled_error(r.detach().cpu(), ref.detach().cpu()) <= TOL
    assert scaled_error(g_hip.cpu(), g_ref.cpu()) <= 1e-11


def test_interface_pack_unpack_kernels():
    """tfem_interface_pack / _unpack (the copies around the multi-GPU all-reduce) against the
    same copies done with torch indexing; a one-rank exchange (all-reduce = identity) through
    InterfaceExchange leaves vals and f unchanged."""
    import ctypes

    from pytorch_fem_solver_amd import _native, parallel

    lib = _native.load()
    g = torch.Generator(device="cuda").manual_seed(3)
    vals = torch.rand(5000, generator=g)
    f = torch.rand(700, generator=g)
    nk, nf, nbuf = 311, 97, 600
    k_idx = torch.randperm(5000, generator=g)[:nk]
    f_idx = torch.randperm(700, generator=g)[:nf]
    pos = torch.randperm(nbuf, generator=g)
    k_pos, f_pos = pos[:nk].contiguous(), pos[nk:nk + nf].contiguous()
    buf = torch.full((nbuf,), 7.0)
    stream = _native.current_stream(buf.device)
    _native.check(lib.tfem_interface_pack(_native.ptr(vals), _native.ptr(f), 8, _native.ptr(k_idx),
                                          _native.ptr(k_pos), nk, _native.ptr(f_idx), _native.ptr(f_pos),
                                          nf, _native.ptr(buf), nbuf, stream))
    want = torch.zeros(nbuf)
    want[k_pos] = vals[k_idx]
    want[f_pos] = f[f_idx]
    assert torch.equal(buf, want)
    buf2 = torch.rand(nbuf, generator=g)
    v2, f2 = vals.clone(), f.clone()
    _native.check(lib.tfem_interface_unpack(_native.ptr(v2), _native.ptr(f2), 8, _native.ptr(k_idx),
                                            _native.ptr(k_pos), nk, _native.ptr(f_idx), _native.ptr(f_pos),
                                            nf, _native.ptr(buf2), stream))
    vw, fw = vals.clone(), f.clone()
    vw[k_idx] = buf2[k_pos]
    fw[f_idx] = buf2[f_pos]
    assert torch.equal(v2, vw) and torch.equal(f2, fw)
    # NULL parts are skipped
    _native.check(lib.tfem_interface_pack(None, _native.ptr(f), 8, _native.ptr(k_idx), _native.ptr(k_pos), nk,
                                          _native.ptr(f_idx), _native.ptr(f_pos), nf, _native.ptr(buf), nbuf,
                                          stream))
    want = torch.zeros(nbuf)
    want[f_pos] = f[f_idx]
    assert torch.equal(buf, want)
    assert ctypes.c_int(lib.tfem_interface_pack(None, None, 3, None, None, 0, None, None, 0, None, 0, None)).value == 1


@pytest.mark.parametrize("order", [2, 4])
def test_two_pass_gather_equals_atomic_scatter_and_is_reproducible(order):
    """P2 (and every path without a plan) forms element blocks and gathers them per CSR entry
    in the reference's accumulation order: equal to the atomic scatter up to summation order,
    bitwise identical from run to run, equal to the oracle."""
    from pytorch_fem_solver_amd import meshgen

    mesh_np = meshgen.delaunay_square(5000, 13)
    got = {}
    for kernel in ("auto", "atomic"):
        basis = tf().Basis(tf().MeshTri(mesh_np), tf().ElementTri(2, order))
        basis._engine.kernel = kernel
        got[kernel] = basis._engine.bilinear(1.0, 1.0)
        if kernel == "auto":
            again = basis._engine.bilinear(1.0, 1.0)
            assert torch.equal(again, got[kernel])
    assert scaled_error(got["auto"].cpu(), got["atomic"].cpu()) <= 1e-14
    # P1 through the generic integrand route (user torch ops + reduce + gather)
    basis = tf().Basis(tf().MeshTri(mesh_np), tf().ElementTri(1, 3))
    K = basis.integrate_bilinear_form(convection_x, layout="csr")
    basis2 = tf().Basis(tf().MeshTri(mesh_np), tf().ElementTri(1, 3))
    basis2._engine.kernel = "atomic"
    K2 = basis2.integrate_bilinear_form(convection_x, layout="csr")
    assert scaled_error(K.values.cpu(), K2.values.cpu()) <= 1e-14
    # linear forms: the P2 load vector (element vectors + gather) and the generic route
    for p_order, q_order in ((2, 4), (1, 3)):
        b_auto = tf().Basis(tf().MeshTri(mesh_np), tf().ElementTri(p_order, q_order))
        b_atomic = tf().Basis(tf().MeshTri(mesh_np), tf().ElementTri(p_order, q_order))
        b_atomic._engine.kernel = "atomic"
        if p_order == 1:
            b_auto._engine.kernel = "gather"  # no plan: element vectors + gather
        f1, f1b, f2 = (b.integrate_linear_form(load) for b in (b_auto, b_auto, b_atomic))
        assert torch.equal(f1, f1b)
        assert scaled_error(f1.cpu(), f2.cpu()) <= 1e-14
        r1 = b_auto.integrate_linear_form(weak_residual, grad_field)
        r2 = b_atomic.integrate_linear_form(weak_residual, grad_field)
        assert scaled_error(r1.cpu(), r2.cpu()) <= 1e-14


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("form", ["stiffness", "stiffness_mass", "mass"])
@pytest.mark.parametrize("order", [2, 4])
@pytest.mark.parametrize("mesh_kind", ["structured", "delaunay_morton"])
def test_p2_row_kernels_against_oracle_and_gather(dtype, form, order, mesh_kind):
    """k_p2_rows (owner-computes vertex rows and edge rows) on a mesh stored with mixed
    orientation: equal to the oracle's P2 assembly entry by entry and to the element-block +
    gather path.  delaunay_morton: vertices with 8 .. 15 neighbours -> k_p2_long_rows, holes in
    the tile kernel's output runs."""
    from pytorch_fem_solver_amd import dofs, meshgen

    torch.set_default_dtype(dtype)
    tol = TOL if dtype == torch.float64 else 2e-5
    if mesh_kind == "structured":
        mesh_np = meshgen.unit_square(70, 0.25, 4)
    else:
        native = meshgen.delaunay_square(6000, 7)
        mesh_np = meshgen.permute_mesh(native, vertex_order=meshgen.morton_order(native["vertices"]))
    tri = mesh_np["triangles"].copy()
    flip = np.random.default_rng(5).random(tri.shape[0]) < 0.4
    tri[flip] = tri[flip][:, [0, 2, 1]]
    mesh_np["triangles"] = tri
    ab = {"stiffness": (1.0, 0.0), "stiffness_mass": (1.0, 1.0), "mass": (0.0, 1.0)}[form]
    got = {}
    for kernel in ("rows", "gather"):
        basis = tf().Basis(tf().MeshTri(mesh_np), tf().ElementTri(2, order))
        basis._engine.kernel = kernel
        got[kernel] = basis._engine.bilinear(*ab)
        assert basis._engine.kernel_name() == ("k_p2_rows" if kernel == "rows" else "k_p2_bilinear_atomic")
        if kernel == "rows":
            n_long = int(basis._engine.p2_plan()["layout"][18])
            assert (n_long > 100) == (mesh_kind == "delaunay_morton")
    verts = mesh_np["vertices"] if dtype == torch.float64 else mesh_np["vertices"].astype(np.float32)
    conn6, xy, _ = dofs.p2_dofs_numpy(mesh_np["vertices"], tri, mesh_np["edges"], mesh_np["edge_markers"],
                                      mesh_np["vertex_markers"])
    geo = orc.geometry(verts[tri], 2, order)
    integrand = {"stiffness": orc.integrand_stiffness, "stiffness_mass": orc.integrand_stiffness_mass,
                 "mass": orc.integrand_mass}[form](geo)
    local = orc.integrate_local(integrand, geo["dx"])
    _, colind, slots = orc.csr_pattern(conn6, xy.shape[0])
    want = orc.assemble_csr_values(local, slots, colind.shape[0])
    assert scaled_error(got["rows"].cpu().double(), want) <= tol
    assert scaled_error(got["rows"].cpu().double(), got["gather"].cpu().double()) <= (1e-13 if dtype == torch.float64 else 2e-5)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("order", [1, 2, 3, 4])
@pytest.mark.parametrize("mesh_kind", ["structured", "delaunay_morton", "holes"])
def test_p2_load_rows_against_oracle_and_gather(dtype, order, mesh_kind, monkeypatch):
    """k_p2_load_rows / k_p2_load_long_rows (the P2 load vector in row form: one lane per DoF over the
    tiles of the stiffness launch, element codes from the plan) on meshes stored with mixed
    orientation, with source values that differ from point to point: equal to the oracle's
    `(f v dx).sum(-3)` + `index_put_(accumulate=True)` (abstract_basis.py:95-112) entry by entry
    and to the element-vector + gather path.  holes: removed elements -- open fans, edges with one
    triangle, isolated vertices (rows without a triangle are written as zeros)."""
    from pytorch_fem_solver_amd import dofs, meshgen

    torch.set_default_dtype(dtype)
    tol = TOL if dtype == torch.float64 else 2e-5
    if mesh_kind == "delaunay_morton":
        native = meshgen.delaunay_square(6000, 7)
        mesh_np = meshgen.permute_mesh(native, vertex_order=meshgen.morton_order(native["vertices"]))
    else:
        mesh_np = meshgen.unit_square(70, 0.25, 4)
    tri = mesh_np["triangles"].copy()
    rng = np.random.default_rng(11)
    if mesh_kind == "holes":
        tri = tri[rng.random(tri.shape[0]) >= 0.1]
    flip = rng.random(tri.shape[0]) < 0.4
    tri[flip] = tri[flip][:, [0, 2, 1]]
    edges, on_boundary = meshgen._edges_from_triangles(tri)
    conn6, xy, _ = dofs.p2_dofs_numpy(mesh_np["vertices"], tri, edges, on_boundary.astype(np.int32).reshape(-1, 1),
                                      mesh_np["vertex_markers"])
    from pytorch_fem_solver_amd.basis.engine import AssemblyEngine

    verts = mesh_np["vertices"] if dtype == torch.float64 else mesh_np["vertices"].astype(np.float32)
    geo = orc.geometry(verts[tri], 2, order)
    nq = geo["dx"].shape[1]
    fq_np = rng.uniform(-1.0, 2.0, size=(tri.shape[0], nq)).astype(verts.dtype)
    got = {}
    for path in ("rows", "gather"):
        monkeypatch.setenv("TFEM_P2_LOAD", path)
        eng = AssemblyEngine(torch.tensor(mesh_np["vertices"]), torch.tensor(tri), torch.tensor(conn6), xy.shape[0], 2, order)
        assert eng.p2_plan() is not None
        calls = []
        inner = eng.lib.tfem_p2_load_rows
        monkeypatch.setattr(eng.lib, "tfem_p2_load_rows", lambda *a: calls.append(1) or inner(*a))
        got[path] = eng.load(torch.tensor(fq_np)).cpu().double().numpy().reshape(-1)
        assert len(calls) == (1 if path == "rows" else 0)
        if path == "rows":
            assert (int(eng.p2_plan()["layout"][18]) > 100) == (mesh_kind == "delaunay_morton")
        monkeypatch.setattr(eng.lib, "tfem_p2_load_rows", inner)
    local = orc.integrate_local(fq_np.reshape(-1, nq, 1, 1) * geo["v"], geo["dx"])
    want = orc.assemble_linear(local, conn6, xy.shape[0]).reshape(-1).astype(np.float64)
    assert scaled_error(got["rows"], want) <= tol
    assert scaled_error(got["rows"], got["gather"]) <= (1e-13 if dtype == torch.float64 else 2e-5)
    if dtype == torch.float64:  # every entry against the scale of its own row: the magnitudes of its terms
        terms = orc.integrate_local(np.abs(fq_np.reshape(-1, nq, 1, 1) * geo["v"]), np.abs(geo["dx"]))
        shares = np.zeros(xy.shape[0])
        np.add.at(shares, conn6.reshape(-1), terms.reshape(-1))
        assert (np.abs(got["rows"] - want) <= 1e-12 * np.maximum(shares, 1e-300)).all()


def test_p2_row_plan_on_unstructured_meshes_and_its_fallback():
    """A Delaunay mesh with a numbering that has locality (Morton) runs the row kernels (its
    vertices with more than seven neighbours as long rows); the native scipy numbering of a
    larger one has no locality and takes element blocks + gather, as before."""
    from pytorch_fem_solver_amd import meshgen

    native = meshgen.delaunay_square(8000, 3)
    basis = tf().Basis(tf().MeshTri(native), tf().ElementTri(2, 2))
    assert basis._engine.p2_plan() is None and basis._engine.kernel_name() == "k_p2_bilinear_atomic"
    want = basis._engine.bilinear(1.0, 1.0)
    morton = meshgen.permute_mesh(native, vertex_order=meshgen.morton_order(native["vertices"]))
    basis = tf().Basis(tf().MeshTri(morton), tf().ElementTri(2, 2))
    assert basis._engine.kernel_name() == "k_p2_rows" and int(basis._engine.p2_plan()["layout"][18]) > 0
    got = basis.integrate_bilinear_form(stiffness_mass, layout="csr")
    # the same operator up to the renumbering: compare invariants (the entry-by-entry check against
    # the oracle is test_p2_row_kernels_against_oracle_and_gather)
    assert abs(float(got.values.sum()) - float(want.sum())) <= 1e-9 * float(want.abs().sum())
    assert abs(float((got.values ** 2).sum()) - float((want ** 2).sum())) <= 1e-9 * float((want ** 2).sum())
    basis = tf().Basis(tf().MeshTri(meshgen.unit_square(30, 0.25, 1)), tf().ElementTri(2, 2))
    assert basis._engine.kernel_name() == "k_p2_rows" and int(basis._engine.p2_plan()["layout"][18]) == 0


def test_csr_spmv_and_cg_solve_against_the_dense_reference_solve():
    """The consumer of large assembled operators: tfem_csr_spmv against torch's
    dense product, and Basis.solve(method="cg") on the CSR operator against the reference's
    dense reduce + torch.linalg.solve on the same Poisson problem."""
    from pytorch_fem_solver_amd import meshgen

    mesh_np = meshgen.unit_square(60, 0.25, 2)
    basis = tf().Basis(tf().MeshTri(mesh_np), tf().ElementTri(1, 3))
    K = basis.integrate_bilinear_form(stiffness, layout="csr")
    x = torch.rand(K.shape[0], 1)
    assert scaled_error(K.matvec(x).cpu(), (K.to_dense() @ x).cpu()) <= 1e-14
    f = basis.integrate_linear_form(load)
    u_dense = basis.solve(K.to_dense(), basis.solution_tensor(), f)
    u_cg = basis.solve(K, basis.solution_tensor(), f, method="cg")
    assert scaled_error(u_cg.cpu(), u_dense.cpu()) <= 1e-9
    # P2 operator (rows of 6-22 entries)
    basis2 = tf().Basis(tf().MeshTri(mesh_np), tf().ElementTri(2, 2))
    K2 = basis2.integrate_bilinear_form(stiffness_mass, layout="csr")
    y = torch.rand(K2.shape[0])
    assert scaled_error(K2.matvec(y).cpu(), (K2.to_dense() @ y).cpu()) <= 1e-14


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_interface_pack_in_one_launch_equals_zero_fill_plus_pack(dtype, monkeypatch):
    """tfem_interface_pack_dense (what the prepared steps of a sharded run enqueue: one launch) writes
    the buffer tfem_interface_pack (memset + scatter) writes: shared entries of vals and f at their
    positions, zeros at the positions other ranks own -- also over what a previous all-reduce left."""
    from pytorch_fem_solver_amd import parallel

    rng = np.random.default_rng(3)
    nnz, nv, n_matrix, n_vector = 5000, 700, 700, 200
    nbuf = n_matrix + n_vector  # matrix entries first, then vector entries
    k_pos, f_pos = rng.permutation(n_matrix)[:450], rng.permutation(n_vector)[:150]
    k_idx, f_idx = rng.permutation(nnz)[:450], rng.permutation(nv)[:150]
    ex = parallel.InterfaceExchange(k_idx, k_pos, f_idx, f_pos, n_matrix, n_vector, torch.device("cuda"), dtype)
    vals = torch.rand(nnz, dtype=dtype, device="cuda")
    f = torch.rand(nv, dtype=dtype, device="cuda")
    got = {}
    for mode in ("dense", "scatter"):
        monkeypatch.setenv("TFEM_INTERFACE_PACK", mode)
        ex._src = None
        pack, unpack = ex.prepared(vals, f)
        ex.buffer.fill_(float("nan"))  # whatever was there before
        pack()
        torch.cuda.synchronize()
        got[mode] = ex.buffer.clone()
    assert torch.equal(got["dense"], got["scatter"])
    want = torch.zeros(nbuf, dtype=dtype, device="cuda")
    want[torch.as_tensor(k_pos, device="cuda")] = vals[torch.as_tensor(k_idx, device="cuda")]
    want[n_matrix + torch.as_tensor(f_pos, device="cuda")] = f[torch.as_tensor(f_idx, device="cuda")]
    assert torch.equal(got["dense"], want)


@pytest.mark.parametrize("order_kind", ["morton", "native"])
def test_partitioned_assembly_with_interface_sum_equals_the_global_operator(order_kind):
    """BASELINE config 4 (general element-range partition) with the device path on every
    shard: four ranks emulated in ONE process -- each shard assembled by the HIP kernels,
    packed with tfem_interface_pack, the all-reduce replaced by the sum of the four packed
    buffers, unpacked with tfem_interface_unpack -- against the operator of the whole mesh.
    After the exchange every entry a rank holds must equal the global one (fp64, 1e-12)."""
    from pytorch_fem_solver_amd import meshgen, parallel

    world = 4
    mesh_np = meshgen.delaunay_square(6000, seed=5)
    n_global = mesh_np["vertices"].shape[0]
    mesh = tf().MeshTri(triangulation=mesh_np)
    basis = tf().Basis(mesh, tf().ElementTri(1, 3))
    K = basis.integrate_bilinear_form(stiffness_mass, layout="csr")
    f = basis.integrate_linear_form(load)
    dense = K.to_dense().cpu().numpy()
    f_global = f.view(-1).cpu().numpy()

    element_order, bounds = parallel.partition_elements(
        mesh_np["vertices"], mesh_np["triangles"], world, order=order_kind)
    shards = []
    for rank in range(world):
        local, l2g = parallel.extract_shard(mesh_np, element_order[bounds[rank]:bounds[rank + 1]])
        lbasis = tf().Basis(tf().MeshTri(triangulation=local), tf().ElementTri(1, 3))
        Kl = lbasis.integrate_bilinear_form(stiffness_mass, layout="csr")
        fl = lbasis.integrate_linear_form(load)
        rowptr, colind = Kl.crow_indices.cpu().numpy(), Kl.col_indices.cpu().numpy()
        ex = parallel.InterfaceExchange.from_partition(
            mesh_np, element_order, bounds, rank, rowptr, colind, l2g, torch.device("cuda"), torch.float64)
        vals, fv = Kl.values.clone().contiguous(), fl.clone().contiguous()
        shards.append((ex, vals, fv, rowptr, colind, l2g, vals.clone()))
    assert len({(s[0].n_matrix, s[0].n_vector) for s in shards}) == 1  # one global interface numbering
    total = torch.zeros_like(shards[0][0].buffer)
    for ex, vals, fv, *_ in shards:
        total += ex.pack(vals, fv)
    for ex, vals, fv, rowptr, colind, l2g, before in shards:
        ex.buffer.copy_(total)
        ex.unpack(vals, fv)
        rows = np.repeat(np.arange(rowptr.shape[0] - 1), np.diff(rowptr))
        want = dense[l2g[rows], l2g[colind]]
        assert scaled_error(vals.cpu(), want) <= TOL
        assert np.abs(fv.view(-1).cpu().numpy() - f_global[l2g]).max() <= TOL * np.abs(f_global).max()
        assert (vals != before).any()  # the exchange added the neighbours' share
        assert 0 < ex.n_vector < n_global


@pytest.mark.parametrize("mesh_kind", ["strip", "partition"])
@pytest.mark.parametrize("load_kind", ["program", "values", "matrix_only"])
def test_interface_tiles_first_two_range_launches_equal_one_launch(mesh_kind, load_kind):
    """with the shared vertices flagged the ring plan lists their tiles first and
    tfem_p1_assemble_rings_range launches the two tile ranges.  The first launch completes every
    shared row (what the exchange packs), the two together reproduce the single launch bit for
    bit, and the exchange may pack / unpack on a side stream while the second range runs."""
    from pytorch_fem_solver_amd import meshgen, parallel
    from pytorch_fem_solver_amd.basis import forms

    if mesh_kind == "strip":  # bench.py's weak-scaling strip, middle rank of three
        n = 150
        mesh_np = meshgen.structured_rectangle(n, n, 0.0, 1.0, 1.0, 2.0, jitter=0.25, seed=0)
        l2g = None
    else:  # bench.py --scaling strong: one of four Morton ranges of a Delaunay mesh
        whole = meshgen.delaunay_square(200000, seed=2)
        whole = meshgen.permute_mesh(whole, vertex_order=meshgen.morton_order(whole["vertices"]))
        element_order, bounds = parallel.partition_elements(whole["vertices"], whole["triangles"], 4, "morton")
        mesh_np, l2g = parallel.extract_shard(whole, element_order[bounds[1]:bounds[2]])
    nv = mesh_np["vertices"].shape[0]

    def engine_of():
        basis = tf().Basis(tf().MeshTri(triangulation=mesh_np), tf().ElementTri(1, 3))
        return basis, basis._engine

    basis, eng = engine_of()
    csr = eng.csr_structure()
    rowptr, colind = csr[0].cpu().numpy(), csr[1].cpu().numpy()
    if mesh_kind == "strip":
        ex = parallel.InterfaceExchange.for_strips(mesh_np, 1, 3, eng)
    else:
        ex = parallel.InterfaceExchange.from_partition(
            whole, element_order, bounds, 1, rowptr, colind, l2g, torch.device("cuda"), torch.float64)
    flags = ex.shared_vertices(nv)
    assert 0 < flags.sum() < nv
    program = forms.trace(load, basis, (), {}).coefficient.program()
    x, y = torch.split(basis.integration_points, 1, dim=-1)
    fq = rhs(x, y).reshape(-1, eng.n_quad).contiguous()
    kw = {"program": dict(source=program), "values": dict(fq=fq), "matrix_only": {}}[load_kind]

    def one_launch(e, out=None, tiles=None):
        if load_kind == "matrix_only":
            span = e.tile_range(tiles) if tiles else None
            vals = e._assemble_rings(1.0, 0.5, out=(out[0] if out else None, None), tiles=span)
            return vals, None
        return e.assemble_system(1.0, 0.5, out=out, tiles=tiles, **kw)

    if load_kind == "values" and not eng.ring_plan()["fq_ok"]:
        pytest.skip("this plan takes source programs only")
    want_v, want_f = one_launch(eng)
    # the same mesh with the interface flagged (flags go in before the plan is built)
    basis2, eng2 = engine_of()
    eng2.set_priority_vertices(flags)
    first, n_pri = eng2.tile_range("priority")
    rest_first, n_rest = eng2.tile_range("rest")
    n_all = eng2.tile_range("all")[1]
    assert first == 0 and rest_first == n_pri and n_pri + n_rest == n_all
    assert 0 < n_pri < (n_all / 4 if mesh_kind == "strip" else n_all)
    with pytest.raises(RuntimeError):
        eng2.set_priority_vertices(flags)  # the plan exists now
    nnz = colind.shape[0]
    out = (torch.full((nnz,), float("nan")), torch.full((nv,), float("nan")))
    one_launch(eng2, out=out, tiles="priority")
    torch.cuda.synchronize()
    row_of = np.repeat(np.arange(nv), np.diff(rowptr))
    shared_entries = torch.from_numpy(flags[row_of]).cuda()
    assert torch.equal(out[0][shared_entries], want_v.view(-1)[shared_entries])  # shared rows complete
    assert torch.isnan(out[0]).any()  # and the other rows still untouched
    # the load vector of a source program is summed with LDS atomics inside a tile (order of
    # arrival): equal to rounding, not bit for bit; the vector of source values is in row form
    same_f = (lambda a, b: scaled_error(a.cpu(), b.cpu()) <= 1e-15) if load_kind == "program" else torch.equal
    if want_f is not None:
        on = torch.from_numpy(flags).cuda()
        assert same_f(out[1][on], want_f.view(-1)[on])
    # exchange of the shared rows on a side stream, beside the launch over the rest (one rank:
    # the all-reduce is the identity; pack and unpack still read and write the shared entries)
    side = torch.cuda.Stream()
    ready = torch.cuda.Event()
    ready.record()
    with torch.cuda.stream(side):
        side.wait_event(ready)
        ex.pack(out[0], out[1] if want_f is not None else None)
        ex.unpack(out[0], out[1] if want_f is not None else None)
    one_launch(eng2, out=out, tiles="rest")
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert torch.equal(out[0], want_v.view(-1))
    if want_f is not None:
        assert same_f(out[1], want_f.view(-1))
    # bench.py's step: the interface rows and their exchange on a high-priority side stream, the
    # other rows on the assembly stream at the same time (the launches write disjoint rows)
    for t in out:
        t.fill_(float("nan"))
    torch.cuda.synchronize()
    fast = torch.cuda.Stream(priority=-1)
    with torch.cuda.stream(fast):
        one_launch(eng2, out=out, tiles="priority")
        ex.pack(out[0], out[1] if want_f is not None else None)
        ex.unpack(out[0], out[1] if want_f is not None else None)
    one_launch(eng2, out=out, tiles="rest")
    torch.cuda.synchronize()
    assert torch.equal(out[0], want_v.view(-1))
    if want_f is not None:
        assert same_f(out[1], want_f.view(-1))
    # argument checks of the range entry
    with pytest.raises(ValueError):
        eng2.assemble_system(1.0, 0.5, source=program, tiles="priority")  # needs out=
    with pytest.raises(ValueError):
        eng2.assemble_system(1.0, 0.5, source=program, out=out, tiles="some")
    with pytest.raises(ValueError):
        eng2._assemble_rings(1.0, 0.5, out=(out[0], None), tiles=(n_all - 1, 2))


def test_prepared_launches_enqueue_the_same_work():
    """engine.prepared_system / InterfaceExchange.prepared (arguments converted once, for the
    host-bound steps of a sharded run) against the general entry points, on the current stream and
    on a stream given explicitly."""
    from pytorch_fem_solver_amd import meshgen, parallel
    from pytorch_fem_solver_amd.basis import forms

    n = 60
    mesh_np = meshgen.structured_rectangle(n, n, 0.0, 1.0, 1.0, 2.0, jitter=0.25, seed=0)
    basis = tf().Basis(tf().MeshTri(triangulation=mesh_np), tf().ElementTri(1, 3))
    eng = basis._engine
    ex = parallel.InterfaceExchange.for_strips(mesh_np, 1, 3, eng)
    eng.set_priority_vertices(ex.shared_vertices(mesh_np["vertices"].shape[0]))
    program = forms.trace(load, basis, (), {}).coefficient.program()
    x, y = torch.split(basis.integration_points, 1, dim=-1)
    fq = rhs(x, y).reshape(-1, eng.n_quad).contiguous()
    want_v, want_f = eng.assemble_system(1.0, 0.5, source=program)
    nnz = want_v.numel()
    side = torch.cuda.Stream()
    for kw in (dict(source=program), dict(fq=fq)):
        out = (torch.full((nnz,), float("nan")), torch.full((eng.n_dofs,), float("nan")))
        whole = eng.prepared_system(1.0, 0.5, out, **kw)
        got = whole()
        assert got[0].data_ptr() == out[0].data_ptr() and torch.equal(out[0], want_v.view(-1))
        assert scaled_error(out[1].cpu(), want_f.view(-1).cpu()) <= 1e-14
        for t in out:
            t.fill_(float("nan"))
        first = eng.prepared_system(1.0, 0.5, out, tiles="priority", **kw)
        rest = eng.prepared_system(1.0, 0.5, out, tiles="rest", **kw)
        pack, unpack = ex.prepared(*out)
        torch.cuda.synchronize()
        first(side)
        pack(side)
        unpack(side)  # one rank: what was packed comes back
        rest()
        torch.cuda.synchronize()
        assert torch.equal(out[0], want_v.view(-1))
        assert scaled_error(out[1].cpu(), want_f.view(-1).cpu()) <= 1e-14
        reference = ex.pack(out[0], out[1]).clone()
        ex.buffer.zero_()
        pack()
        torch.cuda.synchronize()
        assert torch.equal(ex.buffer, reference)
    with pytest.raises(ValueError):
        eng.prepared_system(1.0, 0.5, out)
    with pytest.raises(ValueError):
        eng.prepared_system(1.0, 0.5, (out[0][:-1], out[1]), source=program)


def test_assembly_launches_can_be_captured_in_a_hip_graph():
    """Launch-bound callers (small meshes, thousands of steps) capture the launch once and
    replay it: the C ABI enqueues on the caller's stream only, so torch.cuda.graph records it.
    Replays write the preallocated buffers and reproduce the eager result bit for bit."""
    from pytorch_fem_solver_amd import meshgen

    mesh_np = meshgen.unit_square(71, 0.25, 0)  # C1: 10,082 elements
    basis = tf().Basis(tf().MeshTri(mesh_np), tf().ElementTri(1, 3))
    eng = basis._engine
    x, y = torch.split(basis.integration_points, 1, dim=-1)
    fq = rhs(x, y).reshape(-1, eng.n_quad).contiguous()
    vals, f = eng.assemble_system(1.0, 0.5, fq)
    out = (torch.empty_like(vals), torch.empty_like(f.view(-1)))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):  # plans and attributes are set up outside the capture
        eng.assemble_system(1.0, 0.5, fq, out=out)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        eng.assemble_system(1.0, 0.5, fq, out=out)
    for _ in range(3):
        out[0].fill_(float("nan"))
        out[1].fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out[0], vals.view(-1)) and torch.equal(out[1], f.view(-1))
    # new source values in place: the replay picks them up
    fq.mul_(2.0)
    graph.replay()
    torch.cuda.synchronize()
    assert scaled_error(out[1].cpu(), (2.0 * f.view(-1)).cpu()) <= 1e-15
    assert torch.equal(out[0], vals.view(-1))


@pytest.mark.parametrize("mesh_kind", ["structured", "delaunay_morton"])
@pytest.mark.parametrize("beta", [0.0, 0.75])
def test_p2_persistent_launch_equals_the_two_launches(mesh_kind, beta, monkeypatch):
    """TFEM_P2_PERSIST=1 (k_p2_rows_all: vertex and edge rows in one persistent, software-pipelined
    launch; not the default, it is slower) writes the same operator as the two launches of
    k_p2_rows, long rows included."""
    from pytorch_fem_solver_amd import meshgen

    if mesh_kind == "structured":
        mesh_np = meshgen.unit_square(90, 0.25, 2)
    else:
        mesh_np = meshgen.delaunay_square(9000, 4)
        mesh_np = meshgen.permute_mesh(mesh_np, vertex_order=meshgen.morton_order(mesh_np["vertices"]))
    out = {}
    for persist in ("0", "1"):
        monkeypatch.setenv("TFEM_P2_PERSIST", persist)
        basis = tf().Basis(tf().MeshTri(mesh_np), tf().ElementTri(2, 4))
        eng = basis._engine
        assert eng.kernel_name() == "k_p2_rows"
        vals = eng.bilinear(1.0, beta)
        torch.cuda.synchronize()
        out[persist] = vals.cpu().numpy().copy()
    assert np.isfinite(out["1"]).all()
    assert scaled_error(out["1"], out["0"]) <= 1e-14


def test_engine_renumbers_a_mesh_without_locality_and_translates_at_its_boundary(monkeypatch):
    """A generator's vertex order (scipy Delaunay: no locality) at >= 50,000 DoFs: the engine
    renumbers the mesh along the Morton curve once, takes the ring kernel with consecutive-vertex
    tiles, and everything a caller sees is in the caller's numbering -- the operator (CSRMatrix with
    the renumbering attached: caller_numbering(), matvec, solve), load vectors, the public fused call,
    autograd through the generic linear form, the interior-edge interpolation and its adjoint --
    against the oracle and against the same mesh assembled without the renumbering."""
    from pytorch_fem_solver_amd import meshgen

    mesh_np = meshgen.delaunay_square(56000, 17)
    mesh_np.pop("neighbors", None)
    nv = mesh_np["vertices"].shape[0]
    mesh = tf().MeshTri(triangulation=mesh_np)
    basis = tf().Basis(mesh, tf().ElementTri(1, 3))
    eng = basis._engine
    assert eng.renumbered and eng.kernel_name() == "k_p1_rings" and eng.ring_plan()["chunked"]
    monkeypatch.setenv("TFEM_RENUMBER", "0")
    plain = tf().Basis(mesh, tf().ElementTri(1, 3))
    assert not plain._engine.renumbered and plain._engine.kernel_name() == "k_p1_tiles_pipe"
    monkeypatch.delenv("TFEM_RENUMBER")

    rowptr, colind, slots = orc.csr_pattern(mesh_np["triangles"], nv)
    local, _ = orc.p1_assemble(mesh_np["vertices"], mesh_np["triangles"], 3, "stiffness_mass")
    want = orc.assemble_csr_values(local, slots, colind.shape[0])
    K = basis.integrate_bilinear_form(stiffness_mass, layout="csr")
    assert K.perm is not None and sorted(K.perm.cpu().tolist()) == list(range(nv))
    C = K.caller_numbering()
    assert C.perm is None
    assert np.array_equal(C.crow_indices.cpu().numpy(), rowptr) and np.array_equal(C.col_indices.cpu().numpy(), colind)
    assert rowwise_error(C.values.cpu(), want, rowptr) <= TOL
    fl, _ = orc.p1_assemble(mesh_np["vertices"], mesh_np["triangles"], 3, "load")
    want_f = orc.assemble_linear(fl, mesh_np["triangles"], nv)
    f = basis.integrate_linear_form(load)
    assert scaled_error(f.cpu(), want_f) <= TOL
    K2, f2 = basis.assemble_system(stiffness_mass, load, layout="csr")
    assert torch.equal(K2.values, K.values) and scaled_error(f2.cpu(), want_f) <= TOL
    # source VALUES (a tensor coefficient) and the generic reduce + scatter path
    fq = rhs(*torch.split(basis.integration_points, 1, dim=-1))
    assert scaled_error(basis.integrate_linear_form(lambda b: fq * b.v).cpu(), want_f) <= TOL
    fw, fw_plain = basis.integrate_linear_form(weak_residual, grad_field), plain.integrate_linear_form(weak_residual, grad_field)
    assert scaled_error(fw.cpu(), fw_plain.cpu()) <= TOL
    Kc = basis.integrate_bilinear_form(lambda b: b.v @ b.v_grad[..., [0]].mT, layout="csr").caller_numbering()
    Kc_plain = plain.integrate_bilinear_form(lambda b: b.v @ b.v_grad[..., [0]].mT, layout="csr")
    assert scaled_error(Kc.values.cpu(), Kc_plain.values.cpu()) <= TOL  # non-symmetric: rows and columns both translated
    # the operator applied and solved in the caller's numbering
    x = torch.sin(torch.arange(nv, dtype=torch.float64) * 0.37).reshape(-1, 1)
    K_plain = plain.integrate_bilinear_form(stiffness_mass, layout="csr")
    assert scaled_error(K.matvec(x).cpu(), K_plain.matvec(x).cpu()) <= TOL
    assert scaled_error(K.diagonal().cpu(), K_plain.diagonal().cpu()) <= TOL
    u = basis.solve(K, basis.solution_tensor(), f, method="cg")
    u_plain = plain.solve(K_plain, plain.solution_tensor(), f, method="cg")
    assert scaled_error(u.cpu(), u_plain.cpu()) <= 1e-8
    # autograd through the generic linear form (cotangent gathered by the CALLER's DoF ids)
    for b in (basis, plain):
        coefficient = torch.ones_like(fq, requires_grad=True)
        out = b.integrate_linear_form(lambda bb, c=coefficient: (c * fq) * torch.cos(bb.v))
        (out.reshape(-1) * x.reshape(-1)).sum().backward()
        b._test_grad = coefficient.grad.clone()
    assert scaled_error(basis._test_grad.cpu(), plain._test_grad.cpu()) <= TOL
    # interior edges: a vertex field in, its adjoint out
    edge_basis = tf().InteriorEdgesBasis(mesh, tf().ElementLine(1, 2))
    got = {}
    for name, b in (("renumbered", basis), ("plain", plain)):
        nodal = x.clone().requires_grad_(True)
        val, grad = b.interpolate(edge_basis, nodal)
        ((val**2).sum() + grad.sum()).backward()
        got[name] = (val.detach(), grad.detach(), nodal.grad.clone())
    for a, b in zip(got["renumbered"], got["plain"]):
        assert scaled_error(a.cpu(), b.cpu()) <= 1e-11
    # sharded runs keep their shards' numbering: refused on a renumbered engine, loudly
    with pytest.raises(NotImplementedError):
        eng.set_priority_vertices(np.zeros(nv, dtype=bool))
    with pytest.raises(NotImplementedError):
        eng.prepared_system(1.0, 0.0, (torch.empty(colind.shape[0]), torch.empty(nv)), source=None, fq=fq.reshape(-1, 4))
