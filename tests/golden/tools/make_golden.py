"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Build-container tool (needs /root/reference; never runs on the GPU box, never
imported by tests).  Every array under ``out_*`` keys is produced by the
reference's own torch code (torch_fem/{element,mesh,basis}) on the committed
``in_*`` inputs.  The single absent third-party import of that code path,
``tensordict`` (a nested-dict container, no arithmetic), is satisfied by
tests/golden/tools/container_only/tensordict.py.

    python tests/golden/tools/make_golden.py
"""

import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
GOLDEN = os.path.abspath(os.path.join(HERE, ".."))
REFERENCE = os.environ.get("TFEM_REFERENCE_ROOT", "/root/reference")
# the repository also holds a package named torch_fem (the drop-in alias of this build): the
# reference's must win, so its root goes in front of the repository's
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(HERE, "container_only"))
sys.path.insert(0, REFERENCE)

import torch_fem as ref  # noqa: E402  (the reference, /root/reference/torch_fem)

assert os.path.abspath(ref.__file__).startswith(os.path.abspath(REFERENCE) + os.sep), ref.__file__

from pytorch_fem_solver_amd import meshgen  # noqa: E402  (inputs only)


def npy(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def mesh_inputs(mesh, prefix="in_"):
    return {prefix + k: v for k, v in mesh.items()}


# ---- the closed vocabulary of forms the reference uses ----
def stiffness(basis):
    return basis.v_grad @ basis.v_grad.mT


def stiffness_mass(basis):  # reference tests/test_assembly.py:68-73
    return basis.v_grad @ basis.v_grad.mT + basis.v @ basis.v.mT


def mass(basis):
    return basis.v @ basis.v.mT


def convection_x(basis):  # non-symmetric: pins the transposed scatter convention
    return basis.v @ basis.v_grad[..., [0]].mT


def rhs(x, y):  # reference tests/test_assembly.py:75-77
    return 2.0 * math.pi**2 * torch.sin(math.pi * x) * torch.sin(math.pi * y)


def load(basis):  # reference tests/test_assembly.py:79-84
    x, y = torch.split(basis.integration_points, 1, dim=-1)
    return rhs(x, y) * basis.v


def rhs_squared(basis):  # reference tests/test_assembly.py:86-90
    x, y = torch.split(basis.integration_points, 1, dim=-1)
    return rhs(x, y) ** 2


def weak_residual(basis, grad_field):  # reference examples/example_weak.py:64-75
    x, y = torch.split(basis.integration_points, 1, dim=-1)
    grad = grad_field(basis.integration_points)
    return rhs(x, y) * basis.v - (basis.v_grad @ grad.mT)


def grad_field(points):
    x, y = torch.split(points, 1, dim=-1)
    return torch.cat([torch.cos(3.0 * x) * y, x * x - torch.sin(2.0 * y)], dim=-1)


def basis_outputs(basis, tag):
    out = {
        f"out_{tag}_v": npy(basis.v),
        f"out_{tag}_v_grad": npy(basis.v_grad),
        f"out_{tag}_integration_points": npy(basis.integration_points),
        f"out_{tag}_dx": npy(basis._dx),
        f"out_{tag}_inv_map_jacobian": npy(basis._inv_map_jacobian),
    }
    return out


def p1_case(mesh, orders, dtype=torch.float64):
    torch.set_default_dtype(dtype)
    data = mesh_inputs(mesh)
    m = ref.MeshTri(triangulation=mesh)
    for order in orders:
        element = ref.ElementTri(polynomial_order=1, integration_order=order)
        basis = ref.Basis(m, element)
        tag = f"q{order}"
        data.update(basis_outputs(basis, tag))
        data[f"out_{tag}_gaussian_nodes"] = npy(element.gaussian_nodes)
        data[f"out_{tag}_gaussian_weights"] = npy(element.gaussian_weights)
        data[f"out_{tag}_K_stiffness"] = npy(basis.integrate_bilinear_form(stiffness))
        data[f"out_{tag}_K_stiffness_mass"] = npy(
            basis.integrate_bilinear_form(stiffness_mass)
        )
        data[f"out_{tag}_K_mass"] = npy(basis.integrate_bilinear_form(mass))
        data[f"out_{tag}_K_convection_x"] = npy(
            basis.integrate_bilinear_form(convection_x)
        )
        data[f"out_{tag}_f_load"] = npy(basis.integrate_linear_form(load))
        data[f"out_{tag}_f_weak_residual"] = npy(
            basis.integrate_linear_form(weak_residual, grad_field)
        )
        data[f"out_{tag}_functional_rhs2"] = npy(
            basis.integrate_functional(rhs_squared)
        )
        data[f"out_{tag}_inner_dofs"] = npy(basis._basis_parameters["inner_dofs"])
        A = basis.integrate_bilinear_form(stiffness)
        b = basis.integrate_linear_form(load)
        u = basis.solve(A, basis.solution_tensor(), b)
        data[f"out_{tag}_u_h"] = npy(u)
        val, grad = basis.interpolate(basis, u)
        data[f"out_{tag}_interp_self_val"] = npy(val)
        data[f"out_{tag}_interp_self_grad"] = npy(grad)
    torch.set_default_dtype(torch.float64)
    return data


def mesh_topology_case(mesh):
    torch.set_default_dtype(torch.float64)
    data = mesh_inputs(mesh)
    m = ref.MeshTri(triangulation=mesh)
    for group in ("interior_edges", "boundary_edges"):
        for key, value in m[group].items():
            data[f"out_{group}_{key}"] = npy(value)
    data["out_cells_length"] = npy(m["cells", "length"])
    data["out_cells_coordinates"] = npy(m["cells", "coordinates"])
    # interior-edge basis + interpolation of a vertex field
    element = ref.ElementTri(polynomial_order=1, integration_order=3)
    basis = ref.Basis(m, element)
    edge_basis = ref.InteriorEdgesBasis(
        m, ref.ElementLine(polynomial_order=1, integration_order=2)
    )
    data["out_edge_integration_points"] = npy(edge_basis.integration_points)
    data["out_edge_dx"] = npy(edge_basis._dx)
    data["out_edge_v"] = npy(edge_basis.v)
    xy = m["vertices", "coordinates"]
    u = (torch.sin(2.0 * xy[:, [0]]) * torch.cos(xy[:, [1]])).reshape(-1, 1)
    data["in_vertex_field"] = npy(u)
    val, grad = basis.interpolate(edge_basis, u)
    data["out_interp_edges_val"] = npy(val)
    data["out_interp_edges_grad"] = npy(grad)
    data["out_edge_functional"] = npy(
        edge_basis.integrate_functional(lambda b: (b.integration_points**2).sum(-1, keepdim=True))
    )
    return data


def p2_element_case(seed=5, n_tri=24):
    """P2 exists in the reference at element level only."""
    torch.set_default_dtype(torch.float64)
    rng = np.random.default_rng(seed)
    base = np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]])
    coords = base[None] + rng.uniform(-0.3, 0.3, size=(n_tri, 3, 2))
    coords[::5] = coords[::5][:, [0, 2, 1]]  # some clockwise triangles
    data = {"in_cell_coordinates": coords}
    X = torch.tensor(coords)
    for order in (2, 3, 4):
        element = ref.ElementTri(polynomial_order=2, integration_order=order)
        jac = X.mT @ element.barycentric_grad  # reference basis.py:87-88
        det, inv = element.compute_det_and_inv_map(jac)
        bar = element.compute_barycentric_coordinates(element.gaussian_nodes)
        v, v_grad = element.compute_shape_functions(bar, inv)
        dx = element.reference_element_area * element.gaussian_weights * det  # basis.py:93-96
        local_K = (v_grad @ v_grad.mT * dx).sum(-3)  # abstract_basis.py:83
        local_M = (v @ v.mT * dx).sum(-3)
        tag = f"q{order}"
        data[f"out_{tag}_bar_coords"] = npy(bar)
        data[f"out_{tag}_det"] = npy(det)
        data[f"out_{tag}_inv"] = npy(inv)
        data[f"out_{tag}_v"] = npy(v)
        data[f"out_{tag}_v_grad"] = npy(v_grad)
        data[f"out_{tag}_dx"] = npy(dx)
        data[f"out_{tag}_local_stiffness"] = npy(local_K)
        data[f"out_{tag}_local_mass"] = npy(local_M)
    return data


class _P2Probe(ref.Basis):
    """Reference Basis with ONLY the DoF numbering supplied from outside.

    The reference raises NotImplementedError for P2 in ``_compute_dofs``
    (basis.py:20-51); everything else (geometry cache, integrate_*, scatter
    index construction) is the reference's own code and runs unchanged.
    """

    p2_connectivity = None
    p2_coordinates = None
    p2_markers = None

    def _compute_dofs(self, mesh, element):
        conn = type(self).p2_connectivity
        coords = type(self).p2_coordinates
        return coords, conn, type(self).p2_markers, coords[conn]


def p2_global_case(mesh):
    torch.set_default_dtype(torch.float64)
    from pytorch_fem_solver_amd.dofs import p2_dofs_numpy

    conn6, dof_xy, dof_markers = p2_dofs_numpy(
        mesh["vertices"], mesh["triangles"], mesh["edges"], mesh["edge_markers"],
        mesh["vertex_markers"],
    )
    data = mesh_inputs(mesh)
    data["in_p2_connectivity"] = conn6
    m = ref.MeshTri(triangulation=mesh)
    _P2Probe.p2_connectivity = torch.tensor(conn6, dtype=torch.int32)
    _P2Probe.p2_coordinates = torch.tensor(dof_xy)
    _P2Probe.p2_markers = torch.tensor(dof_markers, dtype=torch.int32)
    for order in (2, 4):
        basis = _P2Probe(m, ref.ElementTri(polynomial_order=2, integration_order=order))
        tag = f"q{order}"
        data[f"out_{tag}_K_stiffness"] = npy(basis.integrate_bilinear_form(stiffness))
        data[f"out_{tag}_K_stiffness_mass"] = npy(
            basis.integrate_bilinear_form(stiffness_mass)
        )
        data[f"out_{tag}_f_load"] = npy(basis.integrate_linear_form(load))
    return data


# ---- config 5: the two-fracture example (examples/example_fractures_fem.py) ----
FRACTURES_3D = [
    [[-1.0, 0.0, 0.0], [1.0, 0.0, 0.0], [-1.0, 1.0, 0.0], [1.0, 1.0, 0.0]],
    [[0.0, 0.0, -1.0], [0.0, 0.0, 1.0], [0.0, 1.0, -1.0], [0.0, 1.0, 1.0]],
]


def frac_rhs(coordinates):  # example_fractures_fem.py:69-99
    x, y, z = torch.split(coordinates, 1, dim=-1)
    x1, _ = torch.split(x, 1, dim=0)
    y1, y2 = torch.split(y, 1, dim=0)
    _, z2 = torch.split(z, 1, dim=0)
    r1 = 6.0 * (y1 - y1**2) * torch.abs(x1) - 2.0 * (torch.abs(x1) ** 3 - torch.abs(x1))
    r2 = -6.0 * (y2 - y2**2) * torch.abs(z2) + 2.0 * (torch.abs(z2) ** 3 - torch.abs(z2))
    return torch.cat([r1, r2], dim=0)


def frac_load(basis):  # example_fractures_fem.py:102-109
    return frac_rhs(basis.integration_points) * basis.v


def frac_exact(coordinates):  # example_fractures_fem.py:127-151
    x, y, z = torch.split(coordinates, 1, dim=-1)
    x1, _ = torch.split(x, 1, dim=0)
    y1, y2 = torch.split(y, 1, dim=0)
    _, z2 = torch.split(z, 1, dim=0)
    e1 = -y1 * (1 - y1) * torch.abs(x1) * (x1**2 - 1)
    e2 = y2 * (1 - y2) * torch.abs(z2) * (z2**2 - 1)
    return torch.cat([e1, e2], dim=0)


def frac_exact_sq(basis):
    return frac_exact(basis.integration_points) ** 2


def fracture_case(m, jitter, compact=False):
    """compact: the inputs are the generator's arguments (the mesh is rebuilt by the test),
    the dense operator is stored as its nonzero entries and only O(N) outputs are kept."""
    torch.set_default_dtype(torch.float64)
    tri = meshgen.fracture_rectangle(m, jitter=jitter, seed=2)
    if compact:
        return fracture_case_compact(m, jitter, tri)
    data = mesh_inputs(tri)
    data["in_fractures_3d"] = np.array(FRACTURES_3D)
    mesh = ref.FracturesTri(
        triangulations=[tri, tri], fractures_3d_data=torch.tensor(FRACTURES_3D)
    )
    V = ref.FractureBasis(mesh, ref.ElementTri(polynomial_order=1, integration_order=4))
    data.update(basis_outputs(V, "frac"))
    for key in (
        "jacobian_fracture_map",
        "inv_jacobian_fracture_map",
        "det_jacobian_fracture_map",
        "translation_vector",
    ):
        data[f"out_mesh_{key}"] = npy(mesh[key])
    data["out_mesh_vertices_coordinates_3d"] = npy(mesh["vertices", "coordinates_3d"])
    data["out_mesh_interior_normals_3d"] = npy(mesh["interior_edges", "normals_3d"])
    for key, value in mesh["interior_edges"].items():
        data[f"out_mesh_interior_edges_{key}"] = npy(value)
    for key, value in V.global_triangulation.items():
        data[f"out_gt_{key}"] = npy(value)
    data["out_inner_dofs"] = npy(V._basis_parameters["inner_dofs"])
    A = V.integrate_bilinear_form(stiffness)
    b = V.integrate_linear_form(frac_load)
    data["out_A"] = npy(A)
    data["out_b"] = npy(b)
    data["out_functional_exact_sq"] = npy(V.integrate_functional(frac_exact_sq))
    u_h = V.solve(A, V.solution_tensor(), b)
    data["out_u_h"] = npy(u_h)
    val, grad = V.interpolate(V, u_h)
    data["out_interp_self_val"] = npy(val)
    data["out_interp_self_grad"] = npy(grad)
    VE = ref.InteriorEdgesFractureBasis(
        mesh, ref.ElementLine(polynomial_order=1, integration_order=2)
    )
    data["out_edge_integration_points"] = npy(VE.integration_points)
    data["out_edge_dx"] = npy(VE._dx)
    ev, eg = V.interpolate(VE, u_h)
    data["out_interp_edges_val"] = npy(ev)
    data["out_interp_edges_grad"] = npy(eg)
    n_E = mesh["interior_edges", "normals_3d"].unsqueeze(-2)
    plus, minus = torch.unbind(eg, dim=-4)  # example_fractures_fem.py:295-297
    data["out_jump"] = npy((plus * n_E).sum(-1) + (minus * -n_E).sum(-1))
    return data


def fracture_case_compact(m, jitter, tri):
    data = {"in_m": np.int64(m), "in_jitter": np.float64(jitter), "in_seed": np.int64(2),
            "in_vertices_sha": np.frombuffer(_sha(tri["vertices"], tri["triangles"]), dtype=np.uint8),
            "in_fractures_3d": np.array(FRACTURES_3D)}
    mesh = ref.FracturesTri(
        triangulations=[tri, tri], fractures_3d_data=torch.tensor(FRACTURES_3D)
    )
    V = ref.FractureBasis(mesh, ref.ElementTri(polynomial_order=1, integration_order=4))
    A = V.integrate_bilinear_form(stiffness)
    b = V.integrate_linear_form(frac_load)
    rows, cols = torch.nonzero(A, as_tuple=True)
    data["out_A_shape"] = np.array(A.shape, dtype=np.int64)
    data["out_A_rows"] = npy(rows).astype(np.int32)
    data["out_A_cols"] = npy(cols).astype(np.int32)
    data["out_A_vals"] = npy(A[rows, cols])
    data["out_b"] = npy(b)
    data["out_inner_dofs"] = npy(V._basis_parameters["inner_dofs"]).astype(np.int32)
    data["out_gt_triangles"] = npy(V.global_triangulation["triangles"]).astype(np.int32)
    data["out_functional_exact_sq_sum"] = npy(V.integrate_functional(frac_exact_sq).sum())
    u_h = V.solve(A, V.solution_tensor(), b)
    del A
    data["out_u_h"] = npy(u_h)
    VE = ref.InteriorEdgesFractureBasis(
        mesh, ref.ElementLine(polynomial_order=1, integration_order=2)
    )
    _, eg = V.interpolate(VE, u_h)
    n_E = mesh["interior_edges", "normals_3d"].unsqueeze(-2)
    plus, minus = torch.unbind(eg, dim=-4)  # example_fractures_fem.py:295-297
    data["out_jump"] = npy((plus * n_E).sum(-1) + (minus * -n_E).sum(-1))
    return data


def _sha(*arrays):
    import hashlib

    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.digest()


def clockwise_mixed():
    cw = meshgen.unit_square(5, 0.2, 1)
    cw["triangles"][::3] = cw["triangles"][::3][:, [0, 2, 1]]
    return cw


#: fixture -> thunk producing its arrays
CASES = {
    # C1: jittered structured square (128 elements), all four quadrature rules
    "p1_square_n8.npz": lambda: p1_case(meshgen.unit_square(8, 0.25, 0), (1, 2, 3, 4)),
    # C1: clockwise triangles mixed in -> signed determinants
    "p1_square_n5_clockwise.npz": lambda: p1_case(clockwise_mixed(), (3,)),
    # C1 stand-in for the "qea0.005" mesh of tests/test_assembly.py (~300 elements)
    "p1_delaunay_170.npz": lambda: p1_case(meshgen.delaunay_square(170, 1), (3,)),
    # float32 default dtype (examples/example_weak.py:20)
    "p1_square_n6_float32.npz": lambda: p1_case(meshgen.unit_square(6, 0.25, 4), (4,), dtype=torch.float32),
    "mesh_topology_n4.npz": lambda: mesh_topology_case(meshgen.unit_square(4, 0.2, 7)),
    "p2_element.npz": p2_element_case,
    "p2_global_n4.npz": lambda: p2_global_case(meshgen.unit_square(4, 0.25, 3)),
    "fracture_L4.npz": lambda: fracture_case(4, 0.0),
    "fracture_L3_jitter.npz": lambda: fracture_case(3, 0.2),
    # SURVEY 8(d) C5 at m = 64 (2 x 16,384 cells): the dense operator (2.2 GB) is kept as its
    # nonzero entries, the O(N) outputs whole
    "fracture_L64.npz": lambda: fracture_case(64, 0.0, compact=True),
}


def main():
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=GOLDEN, help="directory the fixtures are written to")
    ap.add_argument("--only", nargs="*", default=None, help="fixture names (default: all)")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    for name, thunk in CASES.items():
        if args.only is not None and name not in args.only:
            continue
        torch.manual_seed(0)
        data = thunk()
        path = os.path.join(args.out, name)
        np.savez_compressed(path, **data)
        print(f"{name}: {len(data)} arrays, {os.path.getsize(path) / 1024:.1f} KiB", flush=True)


if __name__ == "__main__":
    main()
