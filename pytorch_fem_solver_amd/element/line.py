"""Segment element for interior-edge (jump) integrals.

Mirror of reference torch_fem/element/element_line.py.  Edge integrals are
O(N_edges) trivial work outside the assembly kernel's scope (SURVEY.md section 2
row 1); they stay torch expressions.
"""

from __future__ import annotations

import torch

from .base import AbstractElement


class ElementLine(AbstractElement):
    """Reference segment [-1, 1] (length 2, element_line.py:14-16)."""

    @property
    def barycentric_grad(self):  # element_line.py:10-12
        return torch.tensor([[-0.5], [0.5]])

    @property
    def reference_element_area(self):
        return 2.0

    def compute_barycentric_coordinates(self, x):  # element_line.py:18-19
        return torch.concat([0.5 * (1.0 - x), 0.5 * (1.0 + x)], dim=-1)

    def _compute_gauss_values(self):  # element_line.py:21-43
        if self.integration_order == 2:
            # computed with torch at the default dtype, like the reference (:25)
            node = 1.0 / torch.sqrt(torch.tensor(3.0))
            gaussian_nodes = torch.tensor([[-node], [node]])
            gaussian_weights = torch.tensor([[[0.5]], [[0.5]]])
        elif self.integration_order == 3:
            node = torch.sqrt(torch.tensor(3 / 5))
            gaussian_nodes = torch.tensor([[0], [-node], [node]])
            gaussian_weights = torch.tensor([[[8 / 18]], [[5 / 18]], [[5 / 18]]])
        else:
            raise NotImplementedError("Integration order not implemented")
        return gaussian_nodes, gaussian_weights.unsqueeze(0)

    def compute_shape_functions(self, bar_coords, inv_map_jacobian):  # element_line.py:45-59
        if self.polynomial_order != 1:
            raise NotImplementedError("Polynomial order not implemented")
        return bar_coords, self.barycentric_grad @ inv_map_jacobian

    def compute_det_and_inv_map(self, map_jacobian):  # element_line.py:61-73
        det = torch.linalg.vector_norm(map_jacobian, dim=-2, keepdim=True)
        return det.unsqueeze(-1), 1.0 / det
