"""Reference elements (mirror of reference torch_fem/element/__init__.py:3-5)."""

from .base import AbstractElement
from .line import ElementLine
from .tri import ElementTri

__all__ = ["AbstractElement", "ElementLine", "ElementTri"]
