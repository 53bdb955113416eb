"""``Basis``: H1-conforming Lagrange basis on one 2-D triangle mesh.

Mirror of reference torch_fem/basis/basis.py.  P1 as in the reference (DoFs = vertices,
basis.py:22-24).  P2 is an extension: the reference has the shape functions
(element_tri.py:43-70) but raises in ``_compute_dofs`` (basis.py:50-51); here the edge
DoFs are numbered ``N_v + edge_id`` as its commented-out block intends (dofs.py).
"""

from __future__ import annotations

import torch

from .. import dofs
from .base import AbstractBasis, LazyIndexDict
from .engine import AssemblyEngine


class _EdgeInterpolateFunction(torch.autograd.Function):
    """tfem_edge_interpolate_p1 with its adjoint in the nodal values (jump terms inside a
    training loss; the reference gets the same derivative from autograd on basis.py:150-158)."""

    @staticmethod
    def forward(ctx, u, engine, edge_cells, points, incidence=None):
        ctx.engine, ctx.edge_cells, ctx.points, ctx.incidence = engine, edge_cells, points, incidence
        ctx.u_shape, ctx.u_device = u.shape, u.device
        value, grad = engine.edge_interpolate(edge_cells, points, u, prepared=True)
        return value.to(u.device), grad.to(u.device)

    @staticmethod
    def backward(ctx, g_value, g_grad):
        # with the vertex -> (edge, side) incidence table: the row-form adjoint (no atomics)
        grad_u = ctx.engine.edge_interpolate_backward(ctx.edge_cells, ctx.points, g_value, g_grad,
                                                      prepared=True, incidence=ctx.incidence)
        return grad_u.reshape(ctx.u_shape).to(ctx.u_device), None, None, None, None


class Basis(AbstractBasis):
    #: False: Basis.interpolate on interior edges evaluates the reference's expression sequence
    #: with torch instead of launching tfem_edge_interpolate_p1 (the tests compare the two)
    edge_kernel = True
    #: True: the adjoint of the edge interpolation runs in row form over the vertices (no atomics,
    #: reproducible); False: hardware floating-point atomics (tfem_edge_interpolate_p1_backward)
    edge_backward_rows = True

    def _compute_dofs(self, mesh, element):
        if element.polynomial_order == 1:
            coords = mesh["vertices", "coordinates"]
            conn = mesh["cells", "vertices"]
            markers = mesh["vertices", "markers"]
            return coords, conn, markers, mesh["cells", "coordinates"]
        if element.polynomial_order == 2:
            # vertex DoFs, then one DoF per edge (dofs.py): sort + binary search on the mesh's device
            conn, coords, markers = dofs.p2_dofs_torch(
                mesh["vertices", "coordinates"], mesh["cells", "vertices"], mesh["edges", "vertices"],
                mesh["edges", "markers"], mesh["vertices", "markers"],
            )
            return coords, conn, markers, coords[conn.long()]
        raise NotImplementedError("Polynomial order not implemented")

    def _compute_basis_parameters(self, coords4global_dofs, global_dofs4elements, nodes4boundary_dofs):
        nb_global_dofs = coords4global_dofs.size(-2)
        inner_dofs = torch.nonzero(nodes4boundary_dofs != 1, as_tuple=True)[-2]
        return LazyIndexDict(
            {
                "bilinear_form_shape": (nb_global_dofs, nb_global_dofs),
                "linear_form_shape": (nb_global_dofs, 1),
                "linear_form_idx": (global_dofs4elements.reshape(-1),),
                "inner_dofs": inner_dofs,
                "nb_dofs": nb_global_dofs,
            },
            connectivity=global_dofs4elements,
        )

    def _make_engine(self, mesh, element):
        return AssemblyEngine(
            mesh["vertices", "coordinates"],
            mesh["cells", "vertices"],
            self._global_dofs4elements,
            self._basis_parameters["nb_dofs"],
            element.polynomial_order,
            element.integration_order,
        )

    def _compute_integral_values(self, mesh, element):
        """One tfem_tri_geometry launch; reshaped to the reference's layouts
        (abstract_basis.py:42-63; shapes in SURVEY.md appendix A)."""
        eng = self._engine
        v_grad, dx, points, inv = eng.geometry()
        e, q = eng.n_elems, eng.n_quad
        v_grad = v_grad.reshape(e, 1, 3, 2) if eng.poly_order == 1 else v_grad
        return {
            "v_grad": eng._home(v_grad),
            "integration_points": eng._home(points.reshape(e, q, 1, 2)),
            "_dx": eng._home(dx.reshape(e, q, 1, 1)),
            "_inv_map_jacobian": eng._home(inv.reshape(e, 1, 2, 2)),
        }

    # kept for API parity; the fused kernels do not call them
    def _compute_jacobian_map(self, mesh, element):  # basis.py:87-88
        return mesh["cells", "coordinates"].mT @ element.barycentric_grad

    def interpolate(self, basis, tensor=None):
        """Evaluate a DoF vector (or a function of the nodes) at this basis's own
        quadrature points or on the interior edges (basis.py:98-177).  On the interior edges
        a P1 DoF vector goes through ONE tfem_edge_interpolate_p1 launch,
        differentiable in the vector (tfem_edge_interpolate_p1_backward); the own-points case
        is a torch expression on the device."""
        from .edges import InteriorEdgesBasis

        on_edges = basis.__class__ == InteriorEdgesBasis
        if basis is not self and not on_edges:
            raise NotImplementedError("Interpolation for this basis not implemented")

        def is_dof_vector(t):
            n = self._basis_parameters["nb_dofs"]
            return torch.is_tensor(t) and t.dtype == self._engine.dtype and tuple(t.shape) == (n, 1)

        def edge_kernel(values):
            staged = getattr(basis, "_edge_kernel_inputs", None)
            pts = basis.integration_points
            if staged is None or staged[0] != (id(self._engine), pts.data_ptr()):
                # device copies of the edge -> cells table and the edge points, validated once
                n_edges, n_points = pts.shape[0], pts.shape[-2]
                cells, points, _, _ = self._engine._edge_inputs(
                    basis.mesh["interior_edges", "cells"], pts.detach().reshape(n_edges, n_points, 2))
                incidence = self._engine.edge_incidence(cells) if self.edge_backward_rows else None
                staged = basis._edge_kernel_inputs = ((id(self._engine), pts.data_ptr()), cells, points, incidence)
            _, cells, points, incidence = staged
            n_edges, n_points = points.shape[0], points.shape[1]
            val, grad = _EdgeInterpolateFunction.apply(values, self._engine, cells, points, incidence)
            return val.reshape(n_edges, 2, n_points, 1, 1), grad.reshape(n_edges, 2, 1, 1, 2)

        kernel_ok = self.edge_kernel and on_edges and self._element.polynomial_order == 1
        if kernel_ok and tensor is not None and is_dof_vector(tensor):
            return edge_kernel(tensor)

        lazy = {}

        def shape_values():
            """(dof ids, v, v_grad) of the torch-expression path, built on first use."""
            if not lazy:
                if basis is self:
                    lazy["ids"] = self._global_dofs4elements.unsqueeze(-2)
                    lazy["v"], lazy["v_grad"] = self.v, self.v_grad
                else:
                    edge_mesh = basis.mesh
                    cell_pairs = edge_mesh["interior_edges", "cells"]
                    gather = edge_mesh.compute_coordinates_4_cells
                    lazy["ids"] = gather(edge_mesh["cells", "vertices"], cell_pairs).unsqueeze(-2)
                    origin = gather(self.mesh["cells", "coordinates"][..., [0], :], cell_pairs).unsqueeze(-3)
                    inv_jac = gather(self._inv_map_jacobian, cell_pairs)
                    edge_points = basis.integration_points.unsqueeze(-3)
                    local_points = self._element.compute_inverse_map(origin, edge_points, inv_jac)
                    bar = self._element.compute_barycentric_coordinates(local_points.squeeze(-3))
                    lazy["v"], lazy["v_grad"] = self._element.compute_shape_functions(bar, inv_jac)
            return lazy["ids"], lazy["v"], lazy["v_grad"]

        if tensor is not None:
            dof_ids, v, v_grad = shape_values()
            nodal = tensor[dof_ids]
            return (nodal * v).sum(-2, keepdim=True), (nodal * v_grad).sum(-2, keepdim=True)

        nodes = self._coords4global_dofs

        def interpolator(function):
            values = function(nodes)
            if kernel_ok and is_dof_vector(values):
                return edge_kernel(values)[0]
            dof_ids, v, _ = shape_values()
            return (values[dof_ids] * v).sum(-2, keepdim=True)

        def interpolator_grad(function):
            values = function(nodes)
            if kernel_ok and is_dof_vector(values):
                return edge_kernel(values)[1]
            dof_ids, _, v_grad = shape_values()
            return (values[dof_ids] * v_grad).sum(-2, keepdim=True)

        return interpolator, interpolator_grad
