"""Drop-in import name: ``import torch_fem`` resolves to the MI355X-native
implementation in ``pytorch_fem_solver_amd`` (same public names as the reference
package for the assembly path, reference torch_fem/__init__.py:3-28)."""

from pytorch_fem_solver_amd import (  # noqa: F401
    Basis,
    CSRMatrix,
    ElementLine,
    ElementTri,
    FractureBasis,
    FracturesTri,
    InteriorEdgesBasis,
    InteriorEdgesFractureBasis,
    MeshesTri,
    MeshTri,
)

__all__ = [
    "Basis",
    "FractureBasis",
    "InteriorEdgesBasis",
    "InteriorEdgesFractureBasis",
    "ElementLine",
    "ElementTri",
    "FracturesTri",
    "MeshTri",
]
