"""Bases on interior edges, used to integrate gradient jumps.

Mirror of reference torch_fem/basis/interior_edges_basis.py and
interior_edges_fracture_basis.py.  Only ``integration_points``, ``_dx``, ``v`` and
``integrate_functional`` are meaningful (the reference marks the DoF maps of these
classes as incorrect, interior_edges_basis.py:20).  Edge integrals are O(N_edges * 2)
work outside the assembly kernel's scope: the bases' own data are
torch expressions, as in the reference; evaluating a P1 DoF vector on the edges
(``Basis.interpolate``) is one launch of libtfem_hip (``tfem_edge_interpolate_p1``).
"""

from __future__ import annotations

import torch

from .base import AbstractBasis, LazyIndexDict


class InteriorEdgesBasis(AbstractBasis):
    def _make_engine(self, mesh, element):
        return None

    def _compute_dofs(self, mesh, element):
        if element.polynomial_order != 1:
            raise NotImplementedError("Polynomial order not implemented")
        coords = mesh["vertices", "coordinates"]
        conn = mesh["cells", "vertices"]
        return coords, conn, mesh["vertices", "markers"], mesh["cells", "coordinates"]

    def _compute_basis_parameters(self, coords4global_dofs, global_dofs4elements, nodes4boundary_dofs):
        n = coords4global_dofs.size(-2)
        return LazyIndexDict(
            {
                "bilinear_form_shape": (n, n),
                "linear_form_shape": (n, 1),
                "linear_form_idx": (global_dofs4elements.reshape(-1),),
                "inner_dofs": torch.nonzero(nodes4boundary_dofs != 1, as_tuple=True)[-2],
                "nb_dofs": n,
            },
            connectivity=global_dofs4elements,
        )

    def _edge_coordinates(self, mesh):
        return mesh["interior_edges", "coordinates"]

    def _compute_shape_values(self, element):
        return element.compute_barycentric_coordinates(element.gaussian_nodes)

    def _map_points(self, mesh, points_2d):
        return points_2d

    def _weight_factor(self, mesh):
        return 1.0

    def _compute_integral_values(self, mesh, element):
        """abstract_basis.py:42-63 specialised to segments (interior_edges_basis.py:63-72)."""
        xy = self._edge_coordinates(mesh)
        jac = xy.mT @ element.barycentric_grad
        det, inv = element.compute_det_and_inv_map(jac)
        bar = element.compute_barycentric_coordinates(element.gaussian_nodes)
        _, v_grad = element.compute_shape_functions(bar, inv)
        points = self._map_points(mesh, bar.mT @ xy.unsqueeze(-3))
        dx = element.reference_element_area * element.gaussian_weights * det * self._weight_factor(mesh)
        return {"v_grad": v_grad, "integration_points": points, "_dx": dx, "_inv_map_jacobian": inv}

    def integrate_functional(self, function, *args, **kwargs):
        return (function(self, *args, **kwargs) * self._dx).sum(-3).sum(-2)

    def integrate_bilinear_form(self, function, *args, **kwargs):
        raise NotImplementedError("edge bases carry no DoF map (reference: 'NOT CORRECT')")

    integrate_linear_form = integrate_bilinear_form


class InteriorEdgesFractureBasis(InteriorEdgesBasis):
    """Edges of fractures embedded in 3-D (interior_edges_fracture_basis.py:63-86)."""

    def _map_points(self, mesh, points_2d):
        jac = mesh["jacobian_fracture_map"].unsqueeze(-3).unsqueeze(-3)
        shift = mesh["translation_vector"].unsqueeze(-3).unsqueeze(-3)
        return (jac @ points_2d.mT + shift).mT

    def _weight_factor(self, mesh):
        return mesh["det_jacobian_fracture_map"]
