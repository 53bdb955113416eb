"""MI355X-native element-wise FEM assembly behind the ``torch_fem`` API.

Same public names as the reference package (torch_fem/__init__.py:3-28) for the
assembly path: elements, meshes and bases.  The batched element loop, quadrature
reduction and scatter into the global operator run in hand-written HIP kernels for
gfx950 (csrc/, C ABI in include/tfem_assembly.h).
"""

from . import meshgen
from .basis import (
    AbstractBasis,
    Basis,
    FractureBasis,
    InteriorEdgesBasis,
    InteriorEdgesFractureBasis,
)
from .element import AbstractElement, ElementLine, ElementTri
from .mesh import AbstractMesh, FracturesTri, MeshData, MeshesTri, MeshTri
from .sparse import CSRMatrix

__all__ = [
    "Basis",
    "FractureBasis",
    "InteriorEdgesBasis",
    "InteriorEdgesFractureBasis",
    "ElementLine",
    "ElementTri",
    "FracturesTri",
    "MeshTri",
    "MeshesTri",
    "CSRMatrix",
    "meshgen",
]
