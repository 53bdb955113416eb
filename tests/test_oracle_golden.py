"""Pin the CPU oracle (oracle/assembly_oracle.py) to the reference's own outputs.

The fixtures were produced by running the reference (tests/golden/tools/
make_golden.py); these tests need neither the reference nor a GPU.
"""

import numpy as np
import pytest

from conftest import load_golden, scaled_error
from oracle import assembly_oracle as orc

TOL = 1e-14  # CPU restatement vs reference CPU run, fp64, scaled norms


def _cells(d):
    return d["in_vertices"][d["in_triangles"].astype(np.int64)]


@pytest.mark.parametrize("order", [1, 2, 3, 4])
def test_quadrature_tables(order):
    d = load_golden("p1_square_n8.npz")
    nodes, weights = orc.gauss_rule(order)
    assert np.array_equal(nodes, d[f"out_q{order}_gaussian_nodes"])
    assert np.array_equal(weights, d[f"out_q{order}_gaussian_weights"])


@pytest.mark.parametrize(
    "fixture,orders",
    [
        ("p1_square_n8.npz", (1, 2, 3, 4)),
        ("p1_square_n5_clockwise.npz", (3,)),
        ("p1_delaunay_170.npz", (3,)),
    ],
)
def test_p1_geometry_and_forms(fixture, orders):
    d = load_golden(fixture)
    conn = d["in_triangles"]
    n = d["in_vertices"].shape[0]
    for order in orders:
        geo = orc.geometry(_cells(d), 1, order)
        tag = f"out_q{order}_"
        assert scaled_error(geo["v"], d[tag + "v"]) <= TOL
        assert scaled_error(geo["v_grad"], d[tag + "v_grad"]) <= TOL
        assert scaled_error(geo["integration_points"], d[tag + "integration_points"]) <= TOL
        assert scaled_error(geo["dx"], d[tag + "dx"]) <= TOL
        assert scaled_error(geo["inv_map_jacobian"], d[tag + "inv_map_jacobian"]) <= TOL
        for name, integrand in (
            ("K_stiffness", orc.integrand_stiffness(geo)),
            ("K_stiffness_mass", orc.integrand_stiffness_mass(geo)),
            ("K_mass", orc.integrand_mass(geo)),
            ("K_convection_x", geo["v"] @ np.swapaxes(geo["v_grad"][..., [0]], -1, -2)),
        ):
            local = orc.integrate_local(integrand, geo["dx"])
            dense = orc.assemble_dense_bilinear(local, conn, n)
            assert scaled_error(dense, d[tag + name]) <= TOL, name
            rowptr, colind, slots = orc.csr_pattern(conn, n)
            vals = orc.assemble_csr_values(local, slots, colind.shape[0])
            assert scaled_error(orc.csr_to_dense(rowptr, colind, vals, n), d[tag + name]) <= TOL
        f = orc.assemble_linear(orc.integrate_local(orc.integrand_load(geo), geo["dx"]), conn, n)
        assert scaled_error(f, d[tag + "f_load"]) <= TOL
        fun = orc.integrate_functional(orc.source_sin_sin(geo["integration_points"]) ** 2, geo["dx"])
        assert scaled_error(fun, d[tag + "functional_rhs2"]) <= TOL


def test_signed_determinant_is_kept():
    d = load_golden("p1_square_n5_clockwise.npz")
    geo = orc.geometry(_cells(d), 1, 3)
    assert (geo["det"] < 0).any() and (geo["det"] > 0).any()


def test_float32_path():
    d = load_golden("p1_square_n6_float32.npz")
    cells = d["in_vertices"].astype(np.float32)[d["in_triangles"].astype(np.int64)]
    geo = orc.geometry(cells, 1, 4)
    assert geo["v_grad"].dtype == np.float32
    assert scaled_error(geo["v_grad"], d["out_q4_v_grad"]) <= 1e-6
    local = orc.integrate_local(orc.integrand_stiffness(geo), geo["dx"])
    dense = orc.assemble_dense_bilinear(local, d["in_triangles"], d["in_vertices"].shape[0])
    assert scaled_error(dense, d["out_q4_K_stiffness"]) <= 1e-5


def test_closed_form_stiffness_matches_reference_output():
    d = load_golden("p1_square_n8.npz")
    k = orc.p1_stiffness_closed_form(d["in_vertices"], d["in_triangles"])
    dense = orc.assemble_dense_bilinear(k, d["in_triangles"], d["in_vertices"].shape[0])
    assert scaled_error(dense, d["out_q3_K_stiffness"]) <= 1e-13


@pytest.mark.parametrize("order", [2, 3, 4])
def test_p2_element_level(order):
    d = load_golden("p2_element.npz")
    geo = orc.geometry(d["in_cell_coordinates"], 2, order)
    tag = f"out_q{order}_"
    assert scaled_error(geo["v"], d[tag + "v"]) <= TOL
    assert scaled_error(geo["v_grad"], d[tag + "v_grad"]) <= TOL
    assert scaled_error(geo["dx"], d[tag + "dx"]) <= TOL
    k = orc.integrate_local(orc.integrand_stiffness(geo), geo["dx"])
    m = orc.integrate_local(orc.integrand_mass(geo), geo["dx"])
    assert scaled_error(k, d[tag + "local_stiffness"]) <= TOL
    assert scaled_error(m, d[tag + "local_mass"]) <= TOL


@pytest.mark.parametrize("order", [2, 4])
def test_p2_global(order):
    d = load_golden("p2_global_n4.npz")
    conn6 = d["in_p2_connectivity"]
    n = int(conn6.max()) + 1
    geo = orc.geometry(_cells(d), 2, order)
    tag = f"out_q{order}_"
    k = orc.integrate_local(orc.integrand_stiffness(geo), geo["dx"])
    assert scaled_error(orc.assemble_dense_bilinear(k, conn6, n), d[tag + "K_stiffness"]) <= TOL
    km = orc.integrate_local(orc.integrand_stiffness_mass(geo), geo["dx"])
    assert scaled_error(orc.assemble_dense_bilinear(km, conn6, n), d[tag + "K_stiffness_mass"]) <= TOL
    f = orc.integrate_local(orc.integrand_load(geo), geo["dx"])
    assert scaled_error(orc.assemble_linear(f, conn6, n), d[tag + "f_load"]) <= TOL


@pytest.mark.parametrize("fixture", ["fracture_L4.npz", "fracture_L3_jitter.npz"])
def test_fracture_geometry_and_assembly(fixture):
    d = load_golden(fixture)
    verts = np.stack([d["in_vertices"]] * 2)
    tris = d["in_triangles"].astype(np.int64)
    fmap = orc.fracture_map(verts, d["in_fractures_3d"])
    assert scaled_error(fmap["jacobian"], d["out_mesh_jacobian_fracture_map"]) <= TOL
    assert scaled_error(fmap["pinv"], d["out_mesh_inv_jacobian_fracture_map"]) <= TOL
    assert scaled_error(fmap["det"], d["out_mesh_det_jacobian_fracture_map"]) <= TOL
    cells = verts[:, tris]
    geo = orc.fracture_geometry(cells, fmap, 4)
    assert scaled_error(geo["v_grad"], d["out_frac_v_grad"]) <= TOL
    assert scaled_error(geo["dx"], d["out_frac_dx"]) <= TOL
    assert scaled_error(geo["integration_points"], d["out_frac_integration_points"]) <= TOL
    conn = d["out_gt_triangles"]
    n = d["out_gt_vertices_2D"].shape[0]
    k = orc.integrate_local(orc.integrand_stiffness(geo), geo["dx"]).reshape(-1, 3, 3)
    assert scaled_error(orc.assemble_dense_bilinear(k, conn, n), d["out_A"]) <= TOL


def test_fracture_assembly_m64():
    """The oracle against the reference's operator at SURVEY 8(d)'s C5 size m = 64 (fixture
    keeps the nonzero entries of the 16,705^2 dense operator and the load vector)."""
    import scipy.sparse as sp

    from pytorch_fem_solver_amd import meshgen

    d = load_golden("fracture_L64.npz")
    tri = meshgen.fracture_rectangle(int(d["in_m"]), jitter=float(d["in_jitter"]), seed=int(d["in_seed"]))
    verts = np.stack([tri["vertices"]] * 2)
    tris = tri["triangles"].astype(np.int64)
    fmap = orc.fracture_map(verts, d["in_fractures_3d"])
    geo = orc.fracture_geometry(verts[:, tris], fmap, 4)
    conn = d["out_gt_triangles"].reshape(-1, 3).astype(np.int64)
    n = int(d["out_A_shape"][0])
    k = orc.integrate_local(orc.integrand_stiffness(geo), geo["dx"]).reshape(-1, 3, 3)
    rows, cols = orc.scatter_indices(conn)
    A = sp.coo_matrix((k.reshape(-1), (rows, cols)), shape=(n, n)).tocsr()
    got = np.asarray(A[d["out_A_rows"], d["out_A_cols"]]).ravel()
    assert scaled_error(got, d["out_A_vals"]) <= TOL
    assert abs(np.abs(A.data).sum() - np.abs(d["out_A_vals"]).sum()) <= 1e-9 * np.abs(d["out_A_vals"]).sum()


def _grad_field_np(points):  # tests/golden/tools/make_golden.py grad_field
    x, y = points[..., [0]], points[..., [1]]
    return np.concatenate([np.cos(3.0 * x) * y, x * x - np.sin(2.0 * y)], axis=-1)


@pytest.mark.parametrize("fixture,orders", [("p1_square_n8.npz", (1, 2, 3, 4)), ("p1_delaunay_170.npz", (3,)),
                                            ("p1_square_n5_clockwise.npz", (3,))])
def test_weak_residual_form_and_its_adjoint(fixture, orders):
    """The VPINN residual f v - v_grad @ g.mT (examples/example_weak.py:64-75) against the
    reference's integrate_linear_form output, and the restated adjoint against finite differences
    of the restated form (linear in g and f: exact up to rounding)."""
    d = load_golden(fixture)
    tris = d["in_triangles"]
    n = d["in_vertices"].shape[0]
    for order in orders:
        geo = orc.geometry(_cells(d), 1, order)
        flux = _grad_field_np(geo["integration_points"])
        local = orc.integrate_local(orc.integrand_weak_residual(geo, flux), geo["dx"])
        assert scaled_error(orc.assemble_linear(local, tris, n), d[f"out_q{order}_f_weak_residual"]) <= TOL
        rng = np.random.default_rng(order)
        cot = rng.standard_normal(n)
        g_flux, g_f = orc.weak_residual_adjoint(geo, tris, cot)
        # <cot, r(flux + h dflux)> - <cot, r(flux)> = h <g_flux, dflux>
        dflux = rng.standard_normal(flux.shape)
        r0 = orc.assemble_linear(local, tris, n).reshape(-1)
        r1 = orc.assemble_linear(orc.integrate_local(orc.integrand_weak_residual(geo, flux + dflux), geo["dx"]), tris, n).reshape(-1)
        assert abs(cot @ (r1 - r0) - (g_flux * dflux).sum()) <= 1e-11 * np.abs(g_flux * dflux).sum()
        df = rng.standard_normal(geo["dx"].shape)
        r2 = orc.assemble_linear(orc.integrate_local(
            orc.integrand_weak_residual(geo, flux, source=lambda p: orc.source_sin_sin(p) + df), geo["dx"]), tris, n).reshape(-1)
        assert abs(cot @ (r2 - r0) - (g_f * df).sum()) <= 1e-11 * np.abs(g_f * df).sum()


def test_edge_interpolation_restatement():
    """oracle.edge_interpolate_p1 against the reference's Basis.interpolate(InteriorEdgesBasis, u)
    output (make_golden.py), with the reference's own edge -> cells table and edge points."""
    d = load_golden("mesh_topology_n4.npz")
    val, grad = orc.edge_interpolate_p1(
        d["in_vertices"], d["in_triangles"], d["out_interior_edges_cells"],
        d["out_edge_integration_points"], d["in_vertex_field"])
    assert val.shape == d["out_interp_edges_val"].shape and grad.shape == d["out_interp_edges_grad"].shape
    assert scaled_error(val, d["out_interp_edges_val"]) <= TOL
    assert scaled_error(grad, d["out_interp_edges_grad"]) <= TOL


def test_c_oracle_matches_numpy_oracle_and_golden():
    """oracle/assembly_oracle.c (OpenMP; CPU baseline of bench.py) against the numpy oracle
    and the reference's own outputs."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    import __graft_entry__ as ge

    ge.build_oracle()
    from oracle import c_oracle

    for fixture, order in (("p1_square_n8.npz", 1), ("p1_square_n8.npz", 4),
                           ("p1_square_n5_clockwise.npz", 3), ("p1_delaunay_170.npz", 3)):
        d = load_golden(fixture)
        verts, tris = d["in_vertices"], d["in_triangles"]
        n = verts.shape[0]
        pts = c_oracle.points(verts, tris, order)
        assert scaled_error(pts.reshape(-1, pts.shape[1], 1, 2), d[f"out_q{order}_integration_points"]) <= TOL
        fq = orc.source_sin_sin(pts)[..., 0]
        k, f = c_oracle.p1_local(verts, tris, order, 1.0, 1.0, fq)
        k_np, _ = orc.p1_assemble(verts, tris, order, "stiffness_mass")
        f_np, _ = orc.p1_assemble(verts, tris, order, "load")
        assert scaled_error(k, k_np) <= TOL and scaled_error(f, f_np[..., 0]) <= TOL
        rowptr, colind, slots = orc.csr_pattern(tris, n)
        vals = c_oracle.scatter_csr(k, slots, colind.shape[0])
        assert scaled_error(orc.csr_to_dense(rowptr, colind, vals, n), d[f"out_q{order}_K_stiffness_mass"]) <= TOL
        fv = c_oracle.scatter_vector(f, tris, n)
        assert scaled_error(fv.reshape(-1, 1), d[f"out_q{order}_f_load"]) <= TOL
    assert c_oracle.threads() >= 1


@pytest.mark.parametrize(
    "fixture,orders",
    [
        ("p1_square_n8.npz", (1, 2, 3, 4)),
        ("p1_square_n5_clockwise.npz", (3,)),
        ("p1_delaunay_170.npz", (3,)),
    ],
)
def test_torch_restatement_against_the_reference_outputs(fixture, orders):
    """oracle/torch_restatement.py (the torch op sequence bench.py's cpu_baseline times, SURVEY.md
    8(d)) against what the reference itself produced: geometry cache, every form of the fixtures,
    the dense scatter and the CSR scatter (the non-symmetric form pins the transposed convention)."""
    import torch

    from oracle import torch_restatement as tr

    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        d = load_golden(fixture)
        verts, tris = torch.from_numpy(d["in_vertices"]), torch.from_numpy(d["in_triangles"])
        n = verts.shape[0]
        rowptr, colind, slots = orc.csr_pattern(d["in_triangles"], n)
        for order in orders:
            tag = f"out_q{order}_"
            geo = tr.geometry_cache(verts, tris, order)
            for key, name in (("v", "v"), ("v_grad", "v_grad"), ("integration_points", "integration_points"),
                              ("dx", "dx"), ("inv_map_jacobian", "inv_map_jacobian")):
                assert geo[key].shape == d[tag + name].shape
                assert scaled_error(geo[key].numpy(), d[tag + name]) <= TOL, key
            forms = {
                "K_stiffness": tr.stiffness_integrand,
                "K_stiffness_mass": tr.stiffness_mass_integrand,
                "K_convection_x": lambda g: g["v"] @ g["v_grad"][..., [0]].mT,
            }
            for name, integrand in forms.items():
                local = tr.local_bilinear(geo, integrand)
                dense = tr.scatter_bilinear_dense(local, tris, n)
                assert scaled_error(dense.numpy(), d[tag + name]) <= TOL, name
                vals = tr.scatter_bilinear_csr(local, torch.from_numpy(slots), colind.shape[0])
                assert scaled_error(orc.csr_to_dense(rowptr, colind, vals.numpy(), n), d[tag + name]) <= TOL, name
            f = tr.scatter_linear(tr.local_linear(geo), tris, n)
            assert f.shape == d[tag + "f_load"].shape
            assert scaled_error(f.numpy(), d[tag + "f_load"]) <= TOL
            x, y = torch.split(geo["integration_points"], 1, dim=-1)
            fun = tr.functional(geo, lambda g: tr.rhs(x, y) ** 2)
            assert scaled_error(fun.numpy(), d[tag + "functional_rhs2"]) <= TOL
    finally:
        torch.set_default_dtype(prev)
