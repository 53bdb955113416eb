"""Triangle element: Gauss rules 1-4, P1/P2 shape functions.

Host-side mirror of reference torch_fem/element/element_tri.py.  The batched
per-element work (Jacobian, det/inverse, physical gradients at every quadrature
point) is done by the HIP kernels in csrc/; the methods here act on the small
reference-element tables and on whatever tensors a caller hands them
(``Basis.interpolate`` uses them on edge quadrature points).
"""

from __future__ import annotations

import torch

from .. import _native
from .base import AbstractElement

_G = ((-1.0, -1.0), (1.0, 0.0), (0.0, 1.0))


class ElementTri(AbstractElement):
    """2-D simplex element (element_tri.py:7)."""

    @property
    def barycentric_grad(self):  # element_tri.py:10-12
        return torch.tensor(_G)

    @property
    def reference_element_area(self):  # element_tri.py:14-16
        return 0.5

    @property
    def outward_normal(self):  # element_tri.py:18-21
        return torch.tensor([[1.0, 1.0], [-1.0, 0.0], [0.0, -1.0]])

    def _compute_gauss_values(self):
        """The rule comes from the native table (csrc/tfem_host.cpp, literals of
        element_tri.py:77-130) so host and device can never disagree."""
        import ctypes

        lib = _native.load()
        nq = lib.tfem_quadrature_size(int(self.integration_order))
        if nq == 0:
            raise NotImplementedError("Integration order not implemented")
        nodes = (ctypes.c_double * (2 * nq))()
        weights = (ctypes.c_double * nq)()
        _native.check(lib.tfem_quadrature_rule(int(self.integration_order), nodes, weights))
        gaussian_nodes = torch.tensor([[nodes[2 * q], nodes[2 * q + 1]] for q in range(nq)])
        gaussian_weights = torch.tensor([[[weights[q]]] for q in range(nq)])
        return gaussian_nodes, gaussian_weights

    def compute_barycentric_coordinates(self, x: torch.Tensor):  # element_tri.py:23-26
        xi, eta = x[..., [0]], x[..., [1]]
        return torch.stack([1.0 - xi - eta, xi, eta], dim=-2)

    def compute_shape_functions(self, bar_coords, inv_map_jacobian):  # element_tri.py:28-75
        order = self.polynomial_order
        if order not in (1, 2):
            raise NotImplementedError("Polynomial order not implemented")
        grads = self.barycentric_grad
        if order == 1:
            return bar_coords, grads @ inv_map_jacobian
        lam = torch.split(bar_coords, 1, dim=-2)
        dlam = torch.split(grads, 1, dim=-2)
        corner = [lam[i] * (2 * lam[i] - 1) for i in range(3)]
        corner_grad = [(4 * lam[i] - 1) * dlam[i] for i in range(3)]
        pairs = ((0, 1), (1, 2), (2, 0))  # edge order of element_tri.py:50-52
        edge = [4 * lam[a] * lam[b] for a, b in pairs]
        edge_grad = [4 * (lam[b] * dlam[a] + lam[a] * dlam[b]) for a, b in pairs]
        v = torch.concat(corner + edge, dim=-2)
        v_grad = torch.concat(corner_grad + edge_grad, dim=-2) @ inv_map_jacobian
        return v, v_grad

    def compute_det_and_inv_map(self, map_jacobian):  # element_tri.py:132-145
        a = map_jacobian[..., 0:1, 0:1]
        b = map_jacobian[..., 0:1, 1:2]
        c = map_jacobian[..., 1:2, 0:1]
        d = map_jacobian[..., 1:2, 1:2]
        det = (a * d - b * c).unsqueeze(-3)
        adjugate = torch.stack(
            [torch.concat([d, -b], dim=-1), torch.concat([-c, a], dim=-1)], dim=-2
        )
        return det, (1 / det) * adjugate
