"""Recognise the closed vocabulary of integrands the reference's callers use
 so they can run as one fused HIP kernel.

The user's callable is called ONCE with a proxy basis whose ``v`` / ``v_grad`` are
symbols.  If the expression it builds is one of

    v_grad @ v_grad.mT            (stiffness)       examples/example_fractures_fem.py:112-116
    v @ v.mT                      (mass)
    c1 * stiffness + c2 * mass    (python scalars)  tests/test_assembly.py:68-73
    f * v   /   v * f             (f a tensor broadcastable to (..., Q, 1, 1))
                                                    tests/test_assembly.py:79-84

the fused kernel is used.  Anything else raises ``Untraceable`` inside the proxy and
the caller evaluates the callable on the real tensors and hands the resulting integrand
to the generic quadrature-reduce + scatter kernel -- never changing semantics.
"""

from __future__ import annotations

import numbers

import torch


class Untraceable(Exception):
    """The expression left the recognised vocabulary."""


class _Symbol:
    __array_priority__ = 1000

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        name = getattr(func, "__name__", "")
        if name in ("mul", "__mul__", "__rmul__", "multiply") and len(args) == 2 and not kwargs:
            left, right = args
            if isinstance(left, _Symbol):
                return left.__mul__(right)
            return right.__rmul__(left)
        if name in ("matmul", "__matmul__") and len(args) == 2 and isinstance(args[0], _Symbol):
            return args[0].__matmul__(args[1])
        raise Untraceable(name)

    def _unsupported(self, *_, **__):
        raise Untraceable(type(self).__name__)

    __add__ = __radd__ = __sub__ = __rsub__ = __mul__ = __rmul__ = _unsupported
    __matmul__ = __rmatmul__ = __truediv__ = __rtruediv__ = __neg__ = __pow__ = _unsupported
    __getitem__ = _unsupported

    def __getattr__(self, name):
        raise Untraceable(f"{type(self).__name__}.{name}")


class _Transposed(_Symbol):
    def __init__(self, of):
        object.__setattr__(self, "of", of)


class ShapeFunctions(_Symbol):
    """``basis.v``"""

    @property
    def mT(self):
        return _Transposed(self)

    def __matmul__(self, other):
        if isinstance(other, _Transposed) and isinstance(other.of, ShapeFunctions):
            return BilinearExpr(0.0, 1.0)
        raise Untraceable("v @ ?")

    def __mul__(self, other):
        if isinstance(other, torch.Tensor):
            return LinearExpr(other)
        raise Untraceable("v * ?")

    __rmul__ = __mul__


class ShapeGradients(_Symbol):
    """``basis.v_grad``"""

    @property
    def mT(self):
        return _Transposed(self)

    def __matmul__(self, other):
        if isinstance(other, _Transposed) and isinstance(other.of, ShapeGradients):
            return BilinearExpr(1.0, 0.0)
        raise Untraceable("v_grad @ ?")


class BilinearExpr(_Symbol):
    """alpha * (v_grad @ v_grad.mT) + beta * (v @ v.mT)"""

    def __init__(self, alpha, beta):
        object.__setattr__(self, "alpha", float(alpha))
        object.__setattr__(self, "beta", float(beta))

    def __add__(self, other):
        if isinstance(other, BilinearExpr):
            return BilinearExpr(self.alpha + other.alpha, self.beta + other.beta)
        raise Untraceable("form + ?")

    __radd__ = __add__

    def __mul__(self, other):
        if isinstance(other, numbers.Real) and not isinstance(other, bool):
            return BilinearExpr(self.alpha * other, self.beta * other)
        raise Untraceable("form * ?")

    __rmul__ = __mul__


class LinearExpr(_Symbol):
    """coefficient(x_q) * v"""

    def __init__(self, coefficient):
        object.__setattr__(self, "coefficient", coefficient)


class TracingBasis:
    """Stands in for the basis while the callable is traced: ``v`` and ``v_grad`` are
    symbols, every other attribute is the real basis's."""

    def __init__(self, basis):
        object.__setattr__(self, "_basis", basis)
        object.__setattr__(self, "v", ShapeFunctions())
        object.__setattr__(self, "v_grad", ShapeGradients())

    def __getattr__(self, name):
        return getattr(object.__getattribute__(self, "_basis"), name)

    def __setattr__(self, name, value):
        raise Untraceable("assignment on the basis")


def trace(function, basis, args, kwargs):
    """Return a BilinearExpr / LinearExpr, or None when the callable is not recognised."""
    try:
        result = function(TracingBasis(basis), *args, **kwargs)
    except Untraceable:
        return None
    if isinstance(result, (BilinearExpr, LinearExpr)):
        return result
    return None
