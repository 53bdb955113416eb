"""Reference-element interface (mirrors reference torch_fem/element/abstract_element.py:8-62)."""

from __future__ import annotations

import abc
from typing import Tuple

import torch


class AbstractElement(abc.ABC):
    """Quadrature rule + shape functions of one reference cell.

    Tables are created with ``torch.tensor`` at construction, so they follow the
    process-wide default dtype/device at that moment, like the reference
    (abstract_element.py:11-16).
    """

    def __init__(self, polynomial_order: int, integration_order: int):
        self.polynomial_order = polynomial_order
        self.integration_order = integration_order
        self.gaussian_nodes, self.gaussian_weights = self._compute_gauss_values()

    def compute_inverse_map(self, first_node, integration_points, inv_map_jacobian):
        """Physical -> reference coordinates, (x - x0) J^-T (abstract_element.py:18-26)."""
        return (integration_points - first_node) @ inv_map_jacobian.mT

    @abc.abstractmethod
    def compute_shape_functions(
        self, bar_coords: torch.Tensor, inv_map_jacobian: torch.Tensor
    ) -> Tuple[torch.Tensor, torch.Tensor]:
        ...

    @abc.abstractmethod
    def _compute_gauss_values(self) -> Tuple[torch.Tensor, torch.Tensor]:
        ...

    @abc.abstractmethod
    def compute_barycentric_coordinates(self, x: torch.Tensor) -> torch.Tensor:
        ...

    @abc.abstractmethod
    def compute_det_and_inv_map(self, map_jacobian: torch.Tensor):
        ...

    @property
    @abc.abstractmethod
    def reference_element_area(self) -> float:
        ...

    @property
    @abc.abstractmethod
    def barycentric_grad(self) -> torch.Tensor:
        ...
