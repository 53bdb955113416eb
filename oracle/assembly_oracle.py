"""CPU ORACLE -- test infrastructure, NOT product code.

numpy restatement of the reference's element-wise assembly path
(Nicolas-Zamorano/pytorch_fem_solver, ``torch_fem``).  Every function cites the
reference file:line whose arithmetic (operation order included) it follows.
Pinned against the golden fixtures in tests/golden/*.npz, which were produced
by running the reference itself (tests/golden/tools/make_golden.py) -- see
tests/test_oracle_golden.py.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product package (pytorch_fem_solver_amd) never
does; its hot path is the HIP library and fails loudly without it.

Shapes follow the reference:
    cell coordinates  (N_T, 3, 2)        v        (Q, n, 1)
    v_grad P1         (N_T, 1, 3, 2)     v_grad P2 (N_T, Q, 6, 2)
    integration pts   (N_T, Q, 1, 2)     dx       (N_T, Q, 1, 1)
"""

from __future__ import annotations

import math

import numpy as np

# --------------------------------------------------------------------------- #
# reference element (torch_fem/element/element_tri.py)
# --------------------------------------------------------------------------- #

#: element_tri.py:10-12
BARYCENTRIC_GRAD = np.array([[-1.0, -1.0], [1.0, 0.0], [0.0, 1.0]])
#: element_tri.py:14-16
REFERENCE_AREA = 0.5


def gauss_rule(integration_order: int, dtype=np.float64):
    """Nodes (Q,2) and weights (Q,1,1) -- literals of element_tri.py:77-130."""
    if integration_order == 1:
        nodes = [[1 / 3, 1 / 3]]
        weights = [1.0]
    elif integration_order == 2:
        nodes = [[1 / 6, 1 / 6], [2 / 3, 1 / 6], [1 / 6, 2 / 3]]
        weights = [1 / 3, 1 / 3, 1 / 3]
    elif integration_order == 3:
        nodes = [[1 / 3, 1 / 3], [0.6, 0.2], [0.2, 0.6], [0.2, 0.2]]
        weights = [-9 / 16, 25 / 48, 25 / 48, 25 / 48]
    elif integration_order == 4:
        nodes = [
            [0.816847572980459, 0.091576213509771],
            [0.091576213509771, 0.816847572980459],
            [0.091576213509771, 0.091576213509771],
            [0.108103018168070, 0.445948490915965],
            [0.445948490915965, 0.108103018168070],
            [0.445948490915965, 0.445948490915965],
        ]
        weights = [0.109951743655322] * 3 + [0.223381589678011] * 3
    else:
        raise NotImplementedError("Integration order not implemented")
    return (
        np.array(nodes, dtype=dtype),
        np.array(weights, dtype=dtype).reshape(-1, 1, 1),
    )


def barycentric_coordinates(points):
    """(..., 2) -> (..., 3, 1): (1-x-y, x, y), element_tri.py:23-26."""
    x = points[..., [0]]
    y = points[..., [1]]
    return np.stack([1.0 - x - y, x, y], axis=-2)


def jacobian_map(cell_coordinates):
    """X^T G, basis.py:87-88 -> (N_T, 2, 2)."""
    g = BARYCENTRIC_GRAD.astype(cell_coordinates.dtype)
    return np.swapaxes(cell_coordinates, -1, -2) @ g


def det_and_inverse(map_jacobian):
    """Signed det (..,1,1,1) and (1/det)*adj (..,1,2,2), element_tri.py:132-145."""
    a = map_jacobian[..., [0], :][..., [0]]
    b = map_jacobian[..., [0], :][..., [1]]
    c = map_jacobian[..., [1], :][..., [0]]
    d = map_jacobian[..., [1], :][..., [1]]
    det = np.expand_dims(a * d - b * c, -3)
    adj = np.stack(
        [np.concatenate([d, -b], axis=-1), np.concatenate([-c, a], axis=-1)], axis=-2
    )
    return det, (1 / det) * adj


def shape_functions(polynomial_order, bar_coords, inv_map_jacobian):
    """v and v_grad, element_tri.py:28-75."""
    dtype = inv_map_jacobian.dtype
    g = BARYCENTRIC_GRAD.astype(dtype)
    l1, l2, l3 = (bar_coords[..., [i], :] for i in range(3))
    g1, g2, g3 = (g[[i], :] for i in range(3))
    if polynomial_order == 1:
        return bar_coords, g @ inv_map_jacobian
    if polynomial_order == 2:
        v = np.concatenate(
            [
                l1 * (2 * l1 - 1),
                l2 * (2 * l2 - 1),
                l3 * (2 * l3 - 1),
                4 * l1 * l2,
                4 * l2 * l3,
                4 * l3 * l1,
            ],
            axis=-2,
        )
        ref_grad = np.concatenate(
            [
                (4 * l1 - 1) * g1,
                (4 * l2 - 1) * g2,
                (4 * l3 - 1) * g3,
                4 * (l2 * g1 + l1 * g2),
                4 * (l3 * g2 + l2 * g3),
                4 * (l1 * g3 + l3 * g1),
            ],
            axis=-2,
        )
        return v, ref_grad @ inv_map_jacobian
    raise NotImplementedError("Polynomial order not implemented")


# --------------------------------------------------------------------------- #
# P1 field on the two sides of the interior edges (torch_fem/basis/basis.py:98-177)
# --------------------------------------------------------------------------- #


def edge_interpolate_p1(vertices, triangles, edge_cells, edge_points, nodal):
    """Tensor branch of ``Basis.interpolate(InteriorEdgesBasis, u)``, basis.py:110-160.

    edge_cells (N_e, 2) cell ids, edge_points (N_e, 1, Q, 2) = the edge basis's integration
    points, nodal (N_v, 1).  Returns value (N_e, 2, Q, 1, 1) and gradient (N_e, 2, 1, 1, 2):
    origin = first vertex of the cell (:122-124), pull-back (x - x0) J^-T
    (abstract_element.py:18-26), barycentric shape functions and G J^-1 (element_tri.py:28-41),
    contraction with the three nodal values (:150-158).
    """
    vertices = np.asarray(vertices)
    tri = np.asarray(triangles, dtype=np.int64)
    cells = np.asarray(edge_cells, dtype=np.int64)
    cell_xy = vertices[tri]  # (N_T, 3, 2)
    _, inv = det_and_inverse(jacobian_map(cell_xy))  # (N_T, 1, 2, 2)
    inv_side = inv[cells]  # (N_e, 2, 1, 2, 2)
    origin = cell_xy[:, [0], :][cells][:, :, None]  # (N_e, 2, 1, 1, 2)
    points = np.asarray(edge_points)[:, None]  # (N_e, 1, 1, Q, 2)
    local = (points - origin) @ np.swapaxes(inv_side, -1, -2)  # (N_e, 2, 1, Q, 2)
    bar = barycentric_coordinates(local[:, :, 0])  # (N_e, 2, Q, 3, 1)
    v, v_grad = shape_functions(1, bar, inv_side)  # (N_e, 2, Q, 3, 1), (N_e, 2, 1, 3, 2)
    u = np.asarray(nodal).reshape(-1, 1)[tri[cells]][:, :, None]  # (N_e, 2, 1, 3, 1)
    return (u * v).sum(-2, keepdims=True), (u * v_grad).sum(-2, keepdims=True)


# --------------------------------------------------------------------------- #
# geometry cache (torch_fem/basis/abstract_basis.py:42-63, basis.py:87-96)
# --------------------------------------------------------------------------- #


def geometry(cell_coordinates, polynomial_order, integration_order):
    """The five cached tensors of ``AbstractBasis._compute_integral_values``."""
    dtype = cell_coordinates.dtype
    nodes, weights = gauss_rule(integration_order, dtype)
    jac = jacobian_map(cell_coordinates)
    det, inv = det_and_inverse(jac)
    bar = barycentric_coordinates(nodes)
    v, v_grad = shape_functions(polynomial_order, bar, inv)
    # basis.py:90-91  bar^T @ X.unsqueeze(-3)
    points = np.swapaxes(bar, -1, -2) @ np.expand_dims(cell_coordinates, -3)
    # basis.py:93-96  area * w * det
    dx = REFERENCE_AREA * weights * det
    return {
        "v": v,
        "v_grad": v_grad,
        "integration_points": points,
        "dx": dx,
        "inv_map_jacobian": inv,
        "det": det,
    }


def fracture_map(vertices_2d, fractures_3d):
    """Affine 2D->3D map per fracture, fractures_tri.py:35-67.

    vertices_2d (F, N_v, 2), fractures_3d (F, 4, 3).
    """
    v2 = vertices_2d[:, :3, :]
    v3 = fractures_3d[:, :3, :]
    ext = np.concatenate([v2, np.ones_like(v3[..., [-1]])], axis=-1)
    lin = np.swapaxes(v3, -1, -2) @ np.swapaxes(np.linalg.inv(ext), -1, -2)
    jac = lin[..., :2]
    trans = lin[..., [-1]]
    cross = np.cross(jac[..., 0], jac[..., 1])
    det = np.sqrt((cross**2).sum(-1)).reshape(-1, 1, 1)
    jt = np.swapaxes(jac, -1, -2)
    pinv = np.linalg.inv(jt @ jac) @ jt
    return {"jacobian": jac, "translation": trans, "det": det, "pinv": pinv}


def fracture_geometry(cell_coordinates, fmap, integration_order):
    """Geometry cache of ``FractureBasis`` (fracture_basis.py:15-26,189-210).

    cell_coordinates (F, N_T, 3, 2).
    """
    geo = geometry(cell_coordinates, 1, integration_order)
    pinv = fmap["pinv"][:, None, None]  # unsqueeze(-3).unsqueeze(-3)
    geo["v_grad"] = geo["v_grad"] @ pinv
    geo["inv_map_jacobian"] = geo["inv_map_jacobian"] @ pinv
    geo["dx"] = geo["dx"] * fmap["det"][..., None, None]
    p2 = geo["integration_points"]
    geo["integration_points"] = np.swapaxes(
        fmap["jacobian"][:, None, None] @ np.swapaxes(p2, -1, -2)
        + fmap["translation"][:, None, None],
        -1,
        -2,
    )
    return geo


# --------------------------------------------------------------------------- #
# integrands: the closed vocabulary the reference uses
# --------------------------------------------------------------------------- #


def integrand_stiffness(geo):
    """v_grad @ v_grad^T, examples/example_fractures_fem.py:112-116."""
    return geo["v_grad"] @ np.swapaxes(geo["v_grad"], -1, -2)


def integrand_mass(geo):
    return geo["v"] @ np.swapaxes(geo["v"], -1, -2)


def integrand_stiffness_mass(geo):
    """tests/test_assembly.py:68-73."""
    return integrand_stiffness(geo) + integrand_mass(geo)


def source_sin_sin(points):
    """2 pi^2 sin(pi x) sin(pi y), tests/test_assembly.py:75-77."""
    x = points[..., [0]]
    y = points[..., [1]]
    return 2.0 * math.pi**2 * np.sin(math.pi * x) * np.sin(math.pi * y)


# op codes of include/tfem_assembly.h (enum tfem_source_op)
SRC_PUSH_X, SRC_PUSH_Y, SRC_PUSH_C = 1, 2, 3
SRC_ADD, SRC_SUB, SRC_SUB_R, SRC_MUL, SRC_DIV, SRC_DIV_R = 4, 5, 6, 7, 8, 9
SRC_ADD_C, SRC_MUL_C, SRC_RSUB_C, SRC_RDIV_C = 10, 11, 12, 13
SRC_NEG, SRC_ABS, SRC_POW_I = 14, 15, 16
SRC_SIN, SRC_COS, SRC_EXP, SRC_SQRT, SRC_LOG, SRC_TANH = 17, 18, 19, 20, 21, 22


def source_program_eval(ops, consts, x, y):
    """CPU restatement of a source program (include/tfem_assembly.h, tfem_source_program): the
    postfix form of the torch expressions a caller applies to the coordinate columns of
    basis.integration_points before `* v` (tests/test_assembly.py:75-84); numpy arrays x, y of
    one shape in, f of that shape out.  One numpy operation per program operation, in program
    order -- the order the tracer recorded the caller's torch operations in."""
    x = np.asarray(x)
    y = np.asarray(y)
    stack = []
    unary = {SRC_NEG: np.negative, SRC_ABS: np.abs, SRC_SIN: np.sin, SRC_COS: np.cos, SRC_EXP: np.exp,
             SRC_SQRT: np.sqrt, SRC_LOG: np.log, SRC_TANH: np.tanh}
    for op, c in zip(ops, consts):
        c = x.dtype.type(c)
        if op == SRC_PUSH_X:
            stack.append(c * x)
        elif op == SRC_PUSH_Y:
            stack.append(c * y)
        elif op == SRC_PUSH_C:
            stack.append(np.full_like(x, c))
        elif op in (SRC_ADD, SRC_SUB, SRC_SUB_R, SRC_MUL, SRC_DIV, SRC_DIV_R):
            hi = stack.pop()
            lo = stack.pop()
            stack.append({SRC_ADD: lambda: lo + hi, SRC_SUB: lambda: lo - hi, SRC_SUB_R: lambda: hi - lo,
                          SRC_MUL: lambda: lo * hi, SRC_DIV: lambda: lo / hi, SRC_DIV_R: lambda: hi / lo}[op]())
        elif op == SRC_ADD_C:
            stack.append(stack.pop() + c)
        elif op == SRC_MUL_C:
            stack.append(stack.pop() * c)
        elif op == SRC_RSUB_C:
            stack.append(c - stack.pop())
        elif op == SRC_RDIV_C:
            stack.append(c / stack.pop())
        elif op == SRC_POW_I:
            t = stack.pop()
            r = t * t
            for _ in range(2, int(c)):
                r = r * t
            stack.append(r)
        elif op in (SRC_NEG, SRC_ABS):
            stack.append(unary[op](stack.pop()))
        elif op in unary:  # the functions carry a factor
            stack.append(c * unary[op](stack.pop()))
        else:
            raise ValueError(f"unknown source operation {op}")
    if len(stack) != 1:
        raise ValueError("the program leaves %d values" % len(stack))
    return stack[0]


def integrand_load(geo, source=source_sin_sin):
    """f(x_q) * v, tests/test_assembly.py:79-84."""
    return source(geo["integration_points"]) * geo["v"]


# --------------------------------------------------------------------------- #
# local integration + scatter (torch_fem/basis/abstract_basis.py:65-112)
# --------------------------------------------------------------------------- #


def integrand_weak_residual(geo, flux, source=source_sin_sin, flux_sign=-1.0):
    """f(x_q) * v - v_grad @ g.mT, examples/example_weak.py:64-75 (flux = g, (..., Q, 1, 2))."""
    f = source(geo["integration_points"])  # (..., Q, 1, 1)
    return f * geo["v"] + flux_sign * (geo["v_grad"] @ np.swapaxes(flux, -1, -2))


def weak_residual_adjoint(geo, connectivity, cotangent, flux_sign=-1.0):
    """What autograd derives from abstract_basis.py:95-112 for the residual form: the cotangents
    of g (..., Q, 1, 2) and of f (..., Q, 1, 1) from the cotangent of the assembled vector."""
    conn = np.asarray(connectivity).reshape(-1, connectivity.shape[-1]).astype(np.int64)
    ct = np.asarray(cotangent).reshape(-1)[conn]  # (E, 3)
    dx = geo["dx"]  # (E, Q, 1, 1)
    grad_flux = flux_sign * dx * np.einsum("ei,eqik->eqk", ct, np.broadcast_to(
        geo["v_grad"], (conn.shape[0], dx.shape[1]) + geo["v_grad"].shape[-2:]))[:, :, None, :]
    grad_f = dx * np.einsum("ei,qi->eq", ct, geo["v"][..., 0])[:, :, None, None]
    return grad_flux, grad_f


def integrate_local(integrand, dx):
    """(integrand * dx).sum(-3), abstract_basis.py:83,104."""
    return (integrand * dx).sum(-3)


def integrate_functional(integrand, dx):
    """(...).sum(-3).sum(-2), abstract_basis.py:65-72."""
    return (integrand * dx).sum(-3).sum(-2)


def scatter_indices(connectivity):
    """rows = conn tiled, cols = conn repeated: basis.py:73-76.

    local[i, j] lands in A[conn[j], conn[i]] (the transposed convention).
    """
    conn = np.asarray(connectivity).reshape(-1, connectivity.shape[-1])
    n = conn.shape[-1]
    rows = np.tile(conn, (1, n)).reshape(-1)
    cols = np.repeat(conn.reshape(-1), n)
    return rows.astype(np.int64), cols.astype(np.int64)


def assemble_dense_bilinear(local, connectivity, n_dofs):
    """index_put_(accumulate=True) into zeros((N,N)), abstract_basis.py:81-91."""
    rows, cols = scatter_indices(connectivity)
    out = np.zeros((n_dofs, n_dofs), dtype=local.dtype)
    np.add.at(out, (rows, cols), local.reshape(-1))
    return out


def assemble_linear(local, connectivity, n_dofs):
    """abstract_basis.py:95-112 -> (N, 1)."""
    out = np.zeros((n_dofs, 1), dtype=local.dtype)
    np.add.at(out, (np.asarray(connectivity).reshape(-1).astype(np.int64),), local.reshape(-1, 1))
    return out


# --------------------------------------------------------------------------- #
# sparse global operator (the reference has none: its matrix is dense,
# abstract_basis.py:81; CSR here is the same sum written to the same (row, col))
# --------------------------------------------------------------------------- #


def csr_pattern(connectivity, n_dofs):
    """Sorted-column CSR pattern of the assembled operator + per-entry slot map.

    Returns rowptr (N+1) int64, colind (nnz) int32, slots (N_T, n, n) int64 with
    ``slots[e, i, j]`` = CSR position of (row conn[e, j], col conn[e, i]).
    """
    rows, cols = scatter_indices(connectivity)
    key = rows * np.int64(n_dofs) + cols
    uniq, inverse = np.unique(key, return_inverse=True)
    urow = uniq // n_dofs
    colind = (uniq % n_dofs).astype(np.int32)
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(urow, minlength=n_dofs))]).astype(np.int64)
    n = connectivity.shape[-1]
    return rowptr, colind, inverse.reshape(-1, n, n)


def assemble_csr_values(local, slots, nnz):
    vals = np.zeros(nnz, dtype=local.dtype)
    np.add.at(vals, slots.reshape(-1), local.reshape(-1))
    return vals


def csr_to_dense(rowptr, colind, vals, n_dofs):
    out = np.zeros((n_dofs, n_dofs), dtype=vals.dtype)
    rows = np.repeat(np.arange(n_dofs), np.diff(rowptr))
    out[rows, colind] = vals
    return out


# --------------------------------------------------------------------------- #
# convenience drivers used by tests / smoke / bench
# --------------------------------------------------------------------------- #


def p1_assemble(vertices, triangles, integration_order, form="stiffness", source=source_sin_sin):
    """Local blocks of one of the named forms on a 2-D P1 mesh.

    Returns (local (N_T,3,3) or (N_T,3,1) or (N_T,1), geo).
    """
    cells = vertices[np.asarray(triangles, dtype=np.int64)]  # abstract_mesh.py:257-262
    geo = geometry(cells, 1, integration_order)
    if form == "stiffness":
        return integrate_local(integrand_stiffness(geo), geo["dx"]), geo
    if form == "mass":
        return integrate_local(integrand_mass(geo), geo["dx"]), geo
    if form == "stiffness_mass":
        return integrate_local(integrand_stiffness_mass(geo), geo["dx"]), geo
    if form == "load":
        return integrate_local(integrand_load(geo, source), geo["dx"]), geo
    raise ValueError(form)


def p1_stiffness_closed_form(vertices, triangles):
    """K_e = (1/(4|T|)) e_i . e_j with e_i the edge opposite vertex i (textbook
    identity, independent of the reference; used as a second anchor)."""
    p = vertices[np.asarray(triangles, dtype=np.int64)]
    e = np.stack([p[:, 2] - p[:, 1], p[:, 0] - p[:, 2], p[:, 1] - p[:, 0]], axis=1)
    area2 = (p[:, 1, 0] - p[:, 0, 0]) * (p[:, 2, 1] - p[:, 0, 1]) - (
        p[:, 2, 0] - p[:, 0, 0]
    ) * (p[:, 1, 1] - p[:, 0, 1])
    return (e @ np.swapaxes(e, -1, -2)) / (2.0 * area2)[:, None, None]
