"""Meshes (mirror of reference torch_fem/mesh/__init__.py:3-6; ``Patches`` is out of
scope: it cannot be constructed in the reference at HEAD)."""

from .container import MeshData
from .fractures import FracturesTri, MeshesTri
from .tri import AbstractMesh, MeshTri

__all__ = ["AbstractMesh", "MeshTri", "MeshesTri", "FracturesTri", "MeshData"]
