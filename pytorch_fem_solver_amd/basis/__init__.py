"""Bases (mirror of reference torch_fem/basis/__init__.py:3-8; ``PatchesBasis`` is out
of scope)."""

from .base import AbstractBasis
from .edges import InteriorEdgesBasis, InteriorEdgesFractureBasis
from .fracture import FractureBasis
from .standard import Basis

__all__ = [
    "AbstractBasis",
    "Basis",
    "FractureBasis",
    "InteriorEdgesBasis",
    "InteriorEdgesFractureBasis",
]
