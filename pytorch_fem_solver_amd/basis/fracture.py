"""``FractureBasis``: P1 basis on F planar fractures glued along their traces.

Mirror of reference torch_fem/basis/fracture_basis.py.  The global numbering
(coincident 3-D vertices merged) is one-off index building in torch
(fracture_basis.py:28-129); assembly is the same fused HIP kernel as ``Basis`` with the
per-fracture 2x3 pseudo-inverse and area factor applied inside the kernel.
"""

from __future__ import annotations

import torch

from ..mesh.container import MeshData
from .base import AbstractBasis, LazyIndexDict
from .engine import AssemblyEngine


def _first_occurrence(inverse, n_unique, n_total):
    """Smallest flat index mapping to each unique row (fracture_basis.py:48-58)."""
    first = torch.full((n_unique,), n_total + 1, dtype=torch.int64)
    first.scatter_reduce_(0, inverse, torch.arange(n_total), reduce="amin", include_self=True)
    return first


class FractureBasis(AbstractBasis):
    #: False: interpolate on the interior edges evaluates the reference's expression sequence with
    #: torch instead of launching tfem_edge_interpolate_p1_fracture (the tests compare the two)
    edge_kernel = True

    def __init__(self, mesh, element):
        self.global_triangulation = self._build_global_triangulation(mesh)
        super().__init__(mesh, element)

    def _build_global_triangulation(self, mesh):
        n_frac, n_vert, _ = mesh["vertices", "coordinates"].shape
        n_edge = mesh["edges", "vertices"].shape[-2]

        xyz = mesh["vertices", "coordinates_3d"].reshape(-1, 3)
        vertices_3d, to_global, multiplicity = torch.unique(
            xyz, dim=0, return_inverse=True, return_counts=True
        )
        trace_vertices = torch.nonzero(multiplicity > 1, as_tuple=True)[0]
        representative = _first_occurrence(to_global, vertices_3d.size(-2), n_frac * n_vert)
        vertices_2d = mesh["vertices", "coordinates"].reshape(-1, 2)[representative]

        vertex_shift = torch.arange(n_frac)[:, None, None] * n_vert
        triangles = to_global[mesh["cells", "vertices"] + vertex_shift].reshape(-1, 3)
        edges_as_global = to_global[mesh["edges", "vertices"] + vertex_shift].reshape(-1, 2)
        edges, edge_to_global, edge_multiplicity = torch.unique(
            edges_as_global, dim=0, return_inverse=True, return_counts=True
        )
        trace_edges = torch.nonzero(edge_multiplicity > 1, as_tuple=True)[0]
        edge_shift = torch.arange(n_frac)[:, None] * n_edge
        trace_edges_local = (
            torch.nonzero(torch.isin(edge_to_global, trace_edges), as_tuple=True)[0].reshape(n_frac, -1)
            - edge_shift
        )
        edge_representative = _first_occurrence(edge_to_global, edges.size(-2), n_frac * n_edge)

        return MeshData(
            vertices_3D=vertices_3d,
            vertices_2D=vertices_2d,
            vertex_markers=mesh["vertices", "markers"].reshape(-1)[representative],
            triangles=triangles,
            edges=edges,
            edge_markers=mesh["edges", "markers"].reshape(-1)[edge_representative],
            global2local_idx=to_global,
            local2global_idx=representative,
            traces__global_vertices_idx=trace_vertices,
            traces_global_edges_idx=trace_edges,
            traces_local_edges_idx=trace_edges_local,
        )

    def _compute_dofs(self, mesh, element):
        if element.polynomial_order != 1:
            raise NotImplementedError("Polynomial order not implemented")
        coords = self.global_triangulation["vertices_2D"]
        conn = self.global_triangulation["triangles"]
        boundary = torch.nonzero(self.global_triangulation["vertex_markers"] == 1)[:, 0]
        return coords, conn, boundary, coords[conn]

    def _compute_basis_parameters(self, coords4global_dofs, global_dofs4elements, nodes4boundary_dofs):
        n = self.global_triangulation["vertices_2D"].shape[-2]
        all_dofs = torch.arange(n)
        inner_dofs = all_dofs[~torch.isin(all_dofs, nodes4boundary_dofs)]
        conn = self.global_triangulation["triangles"]
        return LazyIndexDict(
            {
                "bilinear_form_shape": (n, n),
                "linear_form_shape": (n, 1),
                "linear_form_idx": (conn.reshape(-1),),
                "inner_dofs": inner_dofs,
                "nb_dofs": n,
            },
            connectivity=conn,
        )

    def _make_engine(self, mesh, element):
        return AssemblyEngine(
            mesh["vertices", "coordinates"],
            mesh["cells", "vertices"],
            self._global_dofs4elements,
            self._basis_parameters["nb_dofs"],
            element.polynomial_order,
            element.integration_order,
            fracture=(mesh["inv_jacobian_fracture_map"], mesh["det_jacobian_fracture_map"]),
        )

    def _compute_integral_values(self, mesh, element):
        """2-D geometry from the kernel, then the fracture map: gradients @ J_F^+
        (fracture_basis.py:20-26), weights * |J_F| (:189-197), points J_F x + t (:199-207)."""
        eng = self._engine
        v_grad, dx, points, inv = eng.geometry()
        f, q = eng.n_fractures, eng.n_quad
        n_t = eng.n_elems // f
        pinv = mesh["inv_jacobian_fracture_map"].unsqueeze(-3).unsqueeze(-3)
        jac = mesh["jacobian_fracture_map"].unsqueeze(-3).unsqueeze(-3)
        shift = mesh["translation_vector"].unsqueeze(-3).unsqueeze(-3)
        area = mesh["det_jacobian_fracture_map"].unsqueeze(-1).unsqueeze(-1)
        v_grad = eng._home(v_grad.reshape(f, n_t, 1, 3, 2))
        inv = eng._home(inv.reshape(f, n_t, 1, 2, 2))
        points = eng._home(points.reshape(f, n_t, q, 1, 2))
        dx = eng._home(dx.reshape(f, n_t, q, 1, 1))
        return {
            "v_grad": v_grad @ pinv,
            "integration_points": (jac @ points.mT + shift).mT,
            "_dx": dx * area,
            "_inv_map_jacobian": inv @ pinv,
        }

    def interpolate(self, basis, tensor=None):
        """fracture_basis.py:212-293 (post-processing; torch expressions)."""
        from .edges import InteriorEdgesFractureBasis

        if basis is self:
            n_frac = self.mesh.batch_size()[0]
            n_local = self.mesh["cells", "vertices"].shape[-1]
            dof_ids = self._global_dofs4elements.reshape(n_frac, -1, 1, n_local)
            v, v_grad = self.v, self.v_grad
        elif (basis.__class__ == InteriorEdgesFractureBasis and self.edge_kernel and tensor is not None
              and torch.is_tensor(tensor) and not tensor.requires_grad and tensor.dtype == self._engine.dtype
              and tensor.dim() == 2 and tensor.shape[-1] == 1 and self._element.polynomial_order == 1):
            # one tfem_edge_interpolate_p1_fracture launch; shapes of the torch
            # expressions below: value (F, N_e, 2, Q, 1, 1), gradient (F, N_e, 2, 1, 1, 3)
            edge_mesh = basis.mesh
            pts = basis.integration_points
            value, grad = self._engine.edge_interpolate_fracture(
                self.mesh["vertices", "coordinates_3d"], edge_mesh["interior_edges", "cells"], pts, tensor)
            f, n_e, _, n_q = value.shape
            return (self._engine._home(value).reshape(f, n_e, 2, n_q, 1, 1),
                    self._engine._home(grad).reshape(f, n_e, 2, 1, 1, 3))
        elif basis.__class__ == InteriorEdgesFractureBasis:
            edge_mesh = basis.mesh
            cell_pairs = edge_mesh["interior_edges", "cells"]
            gather = edge_mesh.compute_coordinates_4_cells
            # NB: per-fracture LOCAL vertex ids index the GLOBAL vector, as in the reference
            # 
            dof_ids = gather(edge_mesh["cells", "vertices"], cell_pairs).unsqueeze(-2)
            origin = gather(self.mesh["cells", "coordinates_3d"][..., [0], :], cell_pairs).unsqueeze(-3)
            inv_jac = gather(self._inv_map_jacobian, cell_pairs)
            edge_points = basis.integration_points.unsqueeze(-3)
            local_points = self._element.compute_inverse_map(origin, edge_points, inv_jac)
            bar = self._element.compute_barycentric_coordinates(local_points.squeeze(-3))
            v, v_grad = self._element.compute_shape_functions(bar, inv_jac)
        else:
            raise NotImplementedError("Interpolation to {basis.__class__} not implemented")

        if tensor is not None:
            nodal = tensor[dof_ids]
            return (nodal * v).sum(-2, keepdim=True), (nodal * v_grad).sum(-2, keepdim=True)

        nodes = self.mesh["vertices", "coordinates_3d"]

        def interpolator(function):
            return (function(nodes)[dof_ids] * v).sum(-2, keepdim=True)

        def interpolator_grad(function):
            return (self.mesh.apply_mask(function(nodes), [dof_ids]) * v_grad).sum(-2, keepdim=True)

        return interpolator, interpolator_grad
