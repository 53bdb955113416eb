"""Recognise the closed vocabulary of integrands the reference's callers use
 so they can run as one fused HIP kernel.

The user's callable is called ONCE with a proxy basis whose ``v``, ``v_grad`` and
``integration_points`` are symbols.  If the expression it builds is one of

    v_grad @ v_grad.mT            (stiffness)       examples/example_fractures_fem.py:112-116
    v @ v.mT                      (mass)
    c1 * stiffness + c2 * mass    (python scalars)  tests/test_assembly.py:68-73
    f * v   /   v * f             f a tensor broadcastable to (..., Q, 1, 1), or an
                                  expression of the coordinate columns of
                                  ``integration_points`` (``torch.split(points, 1, dim=-1)``,
                                  + - * / **, sin cos exp sqrt abs log tanh, python scalars)
                                                    tests/test_assembly.py:75-84
    f * v - v_grad @ g.mT         (g a tensor (..., Q, 1, 2): the VPINN residual)
                                                    examples/example_weak.py:64-75

the fused kernels are used; a coordinate expression is compiled into a source program
(include/tfem_assembly.h, ``tfem_source_program``) that the assembly launch evaluates at the
integration points itself.  The symbols are *total*: any operation outside this vocabulary
turns the symbols involved into the real tensors they stand for (``materialize``) and carries
on with torch, so the callable still runs once and the caller receives the integrand tensor
for the generic quadrature-reduce + scatter kernels -- semantics never change.
"""

from __future__ import annotations

import math
import numbers

import torch

from .. import _native


class Untraceable(Exception):
    """The callable did something the proxy basis cannot stand in for."""


def _is_scalar(value):
    if isinstance(value, bool):
        return False
    if isinstance(value, numbers.Real):
        return True
    # 0-dim tensors on the CPU only: float() of a device scalar is a host synchronisation per
    # integrate_* call (training loops), with the value baked into the program -- those take the
    # torch path like any other tensor
    return isinstance(value, torch.Tensor) and value.dim() == 0 and not value.requires_grad \
        and value.dtype.is_floating_point and value.device.type == "cpu"


def _scalar(value):
    return float(value)


def materialize(value):
    """The real tensor a symbol stands for (tensors and everything else pass through)."""
    if isinstance(value, _Symbol):
        return value.materialize()
    if isinstance(value, (tuple, list)):
        return type(value)(materialize(v) for v in value)
    return value


class _Symbol:
    __array_priority__ = 1000

    def __init__(self, basis):
        object.__setattr__(self, "_basis", basis)

    def materialize(self):
        raise NotImplementedError

    # any torch function outside the vocabulary: carry on with the real tensors
    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        handled = _dispatch_torch_function(func, args, kwargs)
        if handled is not NotImplemented:
            return handled
        return func(*materialize(tuple(args)), **{k: materialize(v) for k, v in kwargs.items()})

    def _binary(self, other, op, reflected=False):
        a, b = self.materialize(), materialize(other)
        return op(b, a) if reflected else op(a, b)

    def __add__(self, other): return self._binary(other, lambda a, b: a + b)
    def __radd__(self, other): return self._binary(other, lambda a, b: a + b, True)
    def __sub__(self, other): return self._binary(other, lambda a, b: a - b)
    def __rsub__(self, other): return self._binary(other, lambda a, b: a - b, True)
    def __mul__(self, other): return self._binary(other, lambda a, b: a * b)
    def __rmul__(self, other): return self._binary(other, lambda a, b: a * b, True)
    def __truediv__(self, other): return self._binary(other, lambda a, b: a / b)
    def __rtruediv__(self, other): return self._binary(other, lambda a, b: a / b, True)
    def __matmul__(self, other): return self._binary(other, lambda a, b: a @ b)
    def __rmatmul__(self, other): return self._binary(other, lambda a, b: a @ b, True)
    def __pow__(self, other): return self._binary(other, lambda a, b: a ** b)
    def __rpow__(self, other): return self._binary(other, lambda a, b: a ** b, True)
    def __neg__(self): return -self.materialize()
    def __abs__(self): return abs(self.materialize())
    def __getitem__(self, index): return self.materialize()[index]
    def __iter__(self): return iter(self.materialize())
    def __len__(self): return len(self.materialize())

    def __getattr__(self, name):
        # methods and attributes of the tensor the symbol stands for (shape, mT, sum ...)
        if name.startswith("__"):
            raise AttributeError(name)
        return getattr(self.materialize(), name)

    def __setattr__(self, name, value):
        raise Untraceable("assignment on a symbol")


# ------------------------------------------------------------------------------------------
# basis.v / basis.v_grad and the forms built from them
# ------------------------------------------------------------------------------------------
class _Transposed(_Symbol):
    def __init__(self, of):
        super().__init__(of._basis)
        object.__setattr__(self, "of", of)

    def materialize(self):
        return self.of.materialize().mT


class ShapeFunctions(_Symbol):
    """``basis.v``"""

    def materialize(self):
        return self._basis.v

    @property
    def mT(self):
        return _Transposed(self)

    def __matmul__(self, other):
        if isinstance(other, _Transposed) and isinstance(other.of, ShapeFunctions):
            return BilinearExpr(self._basis, 0.0, 1.0)
        return super().__matmul__(other)

    def __mul__(self, other):
        if isinstance(other, SourceExpr) or (isinstance(other, torch.Tensor) and not isinstance(other, _Symbol)):
            return LinearExpr(self._basis, other)
        if _is_scalar(other):
            return LinearExpr(self._basis, SourceExpr(self._basis, ("c", _scalar(other))))
        return super().__mul__(other)

    __rmul__ = __mul__


class ShapeGradients(_Symbol):
    """``basis.v_grad``"""

    def materialize(self):
        return self._basis.v_grad

    @property
    def mT(self):
        return _Transposed(self)

    def __matmul__(self, other):
        if isinstance(other, _Transposed) and isinstance(other.of, ShapeGradients):
            return BilinearExpr(self._basis, 1.0, 0.0)
        if isinstance(other, torch.Tensor) and not isinstance(other, _Symbol) and other.dim() >= 2 \
                and other.shape[-1] == 1:
            return LinearExpr(self._basis, None, flux=other.mT, flux_sign=1.0)  # v_grad @ g.mT
        return super().__matmul__(other)


class BilinearExpr(_Symbol):
    """alpha * (v_grad @ v_grad.mT) + beta * (v @ v.mT)"""

    def __init__(self, basis, alpha, beta):
        super().__init__(basis)
        object.__setattr__(self, "alpha", float(alpha))
        object.__setattr__(self, "beta", float(beta))

    def materialize(self):
        b = self._basis
        out = None
        if self.alpha != 0.0:
            out = self.alpha * (b.v_grad @ b.v_grad.mT) if self.alpha != 1.0 else b.v_grad @ b.v_grad.mT
        if self.beta != 0.0:
            m = self.beta * (b.v @ b.v.mT) if self.beta != 1.0 else b.v @ b.v.mT
            out = m if out is None else out + m
        return out if out is not None else 0.0 * (b.v @ b.v.mT)

    def __add__(self, other):
        if isinstance(other, BilinearExpr):
            return BilinearExpr(self._basis, self.alpha + other.alpha, self.beta + other.beta)
        return super().__add__(other)

    __radd__ = __add__

    def __sub__(self, other):
        if isinstance(other, BilinearExpr):
            return BilinearExpr(self._basis, self.alpha - other.alpha, self.beta - other.beta)
        return super().__sub__(other)

    def __mul__(self, other):
        if _is_scalar(other):
            return BilinearExpr(self._basis, self.alpha * _scalar(other), self.beta * _scalar(other))
        return super().__mul__(other)

    __rmul__ = __mul__

    def __neg__(self):
        return BilinearExpr(self._basis, -self.alpha, -self.beta)


class LinearExpr(_Symbol):
    """``coefficient(x_q) * v  +  flux_sign * v_grad @ flux.mT``: the load vector of a source
    (coefficient: a tensor broadcastable to (..., Q, 1, 1) or a SourceExpr, or None) and the
    VPINN residual's flux term (flux: a tensor broadcastable to (..., Q, 1, D), or None)."""

    def __init__(self, basis, coefficient, flux=None, flux_sign=0.0):
        super().__init__(basis)
        object.__setattr__(self, "coefficient", coefficient)
        object.__setattr__(self, "flux", flux)
        object.__setattr__(self, "flux_sign", float(flux_sign) if flux is not None else 0.0)

    def materialize(self):
        b = self._basis
        out = None
        if self.coefficient is not None:
            out = materialize(self.coefficient) * b.v
        if self.flux is not None:
            term = b.v_grad @ self.flux.mT
            term = term if self.flux_sign == 1.0 else self.flux_sign * term
            out = term if out is None else out + term
        return out

    def _combine(self, other, sign):
        if isinstance(other, LinearExpr):
            if self.coefficient is not None and other.coefficient is not None:
                return None
            if self.flux is not None and other.flux is not None:
                return None
            if other.coefficient is not None and sign != 1.0:
                return None  # c * v - f * v: leave to torch
            coefficient = self.coefficient if self.coefficient is not None else other.coefficient
            flux, flux_sign = (self.flux, self.flux_sign) if self.flux is not None else (other.flux, sign * other.flux_sign)
            return LinearExpr(self._basis, coefficient, flux, flux_sign)
        return None

    def __add__(self, other):
        merged = self._combine(other, 1.0)
        return merged if merged is not None else super().__add__(other)

    __radd__ = __add__

    def __sub__(self, other):
        merged = self._combine(other, -1.0)
        return merged if merged is not None else super().__sub__(other)


# ------------------------------------------------------------------------------------------
# basis.integration_points and scalar fields of its coordinate columns
# ------------------------------------------------------------------------------------------
def _columns(basis):
    """(x, y) coordinate symbols -- only for one 2-D mesh: the kernels evaluate f(x, y)."""
    points = basis.integration_points
    if points.shape[-1] != 2 or points.dim() != 4:
        return None
    return (SourceExpr(basis, ("x",)), SourceExpr(basis, ("y",)))


class PointsSymbol(torch.Tensor):
    """``basis.integration_points`` (..., Q, 1, D): an alias OF the real tensor (same storage), so
    that code the tracer cannot see into -- a scripted network, ``requires_grad_`` followed by
    ``torch.autograd.grad(inputs=[points])`` as in model/neural_network.py:85-100 -- works on
    it unchanged, while ``torch.split(points, 1, dim=-1)`` and ``points[..., i:i+1]`` hand out
    coordinate symbols.  Every other operation runs on it as on a plain tensor."""

    @staticmethod
    def __new__(cls, basis):
        alias = torch.Tensor._make_subclass(cls, basis.integration_points.detach(), False)
        alias._tfem_basis = basis
        return alias

    def materialize(self):
        return self

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, "__name__", "")
        first = args[0] if args else None
        if isinstance(first, PointsSymbol) and getattr(first, "_tfem_basis", None) is not None:
            basis = first._tfem_basis
            if name == "split" and len(args) <= 3:
                dim = kwargs.get("dim", args[2] if len(args) > 2 else 0)
                size = kwargs.get("split_size_or_sections", kwargs.get("split_size", args[1] if len(args) > 1 else None))
                if size == 1 and dim in (-1, first.dim() - 1):
                    cols = _columns(basis)
                    if cols is not None:
                        return cols
            if name == "__getitem__" and len(args) == 2:
                index = args[1]
                if isinstance(index, tuple) and len(index) == 2 and index[0] is Ellipsis:
                    last, cols = index[1], _columns(basis)
                    if cols is not None:
                        if isinstance(last, list) and len(last) == 1 and last[0] in (0, 1, -1, -2):
                            return cols[last[0]]
                        if isinstance(last, slice) and last.step in (None, 1):
                            start, stop, _ = last.indices(2)
                            if stop - start == 1:
                                return cols[start]
        with torch._C.DisableTorchFunctionSubclass():
            return func(*materialize(tuple(args)), **{k: materialize(v) for k, v in kwargs.items()})


_UNARY = ("neg", "abs", "sin", "cos", "exp", "sqrt", "log", "tanh")
_TORCH_UNARY = {
    "neg": torch.neg, "abs": torch.abs, "sin": torch.sin, "cos": torch.cos, "exp": torch.exp,
    "sqrt": torch.sqrt, "log": torch.log, "tanh": torch.tanh,
}


class SourceExpr(_Symbol):
    """A scalar field f(x, y) of the coordinate columns, shape (..., Q, 1, 1)."""

    def __init__(self, basis, node):
        super().__init__(basis)
        object.__setattr__(self, "node", node)

    # ---- evaluation with torch (fallback and tests) -----------------------------------
    def materialize(self):
        points = self._basis.integration_points
        return _evaluate(self.node, points[..., 0:1], points[..., 1:2])

    # ---- arithmetic ---------------------------------------------------------------------
    def _lift(self, other):
        if isinstance(other, SourceExpr):
            return other.node
        if _is_scalar(other):
            return ("c", _scalar(other))
        return None

    def _arith(self, other, op, reflected=False):
        node = self._lift(other)
        if node is None:
            return None
        a, b = (node, self.node) if reflected else (self.node, node)
        if a[0] == "c" and b[0] == "c":
            return SourceExpr(self._basis, ("c", _fold(op, a[1], b[1])))
        return SourceExpr(self._basis, (op, a, b))

    def __add__(self, other):
        out = self._arith(other, "add")
        return out if out is not None else super().__add__(other)

    def __radd__(self, other):
        out = self._arith(other, "add", True)
        return out if out is not None else super().__radd__(other)

    def __sub__(self, other):
        out = self._arith(other, "sub")
        return out if out is not None else super().__sub__(other)

    def __rsub__(self, other):
        out = self._arith(other, "sub", True)
        return out if out is not None else super().__rsub__(other)

    def __mul__(self, other):
        if isinstance(other, ShapeFunctions):
            return other.__mul__(self)
        out = self._arith(other, "mul")
        return out if out is not None else super().__mul__(other)

    def __rmul__(self, other):
        if isinstance(other, ShapeFunctions):
            return other.__mul__(self)
        out = self._arith(other, "mul", True)
        return out if out is not None else super().__rmul__(other)

    def __truediv__(self, other):
        out = self._arith(other, "div")
        return out if out is not None else super().__truediv__(other)

    def __rtruediv__(self, other):
        out = self._arith(other, "div", True)
        return out if out is not None else super().__rtruediv__(other)

    def __neg__(self):
        return self._unary("neg")

    def __abs__(self):
        return self._unary("abs")

    def __pow__(self, exponent):
        if _is_scalar(exponent):
            e = _scalar(exponent)
            if e == 1.0:
                return self
            if e == 0.5:
                return self._unary("sqrt")
            if e == -1.0:
                return SourceExpr(self._basis, ("div", ("c", 1.0), self.node))
            if e == float(int(e)) and 2 <= int(e) <= 8:
                if self.node[0] == "c":
                    return SourceExpr(self._basis, ("c", self.node[1] ** int(e)))
                return SourceExpr(self._basis, ("powi", self.node, int(e)))
        return super().__pow__(exponent)

    def _unary(self, op):
        if self.node[0] == "c":
            value = _TORCH_UNARY[op](torch.tensor(self.node[1], dtype=torch.float64)).item()
            return SourceExpr(self._basis, ("c", value))
        return SourceExpr(self._basis, (op, self.node))

    # tensor-style methods the callers use
    def sin(self): return self._unary("sin")
    def cos(self): return self._unary("cos")
    def exp(self): return self._unary("exp")
    def sqrt(self): return self._unary("sqrt")
    def abs(self): return self._unary("abs")
    def log(self): return self._unary("log")
    def tanh(self): return self._unary("tanh")
    def neg(self): return self._unary("neg")
    def square(self): return self.__pow__(2)
    def pow(self, exponent): return self.__pow__(exponent)

    # ---- compilation --------------------------------------------------------------------
    def program(self):
        """The source program of this field (``_native.SourceProgram``), or None when it does
        not fit (more than SOURCE_MAX_OPS operations or SOURCE_STACK stack entries)."""
        return compile_program(self.node)


def _fold(op, a, b):
    if op == "add":
        return a + b
    if op == "sub":
        return a - b
    if op == "mul":
        return a * b
    return a / b


def _evaluate(node, x, y):
    kind = node[0]
    if kind == "x":
        return x
    if kind == "y":
        return y
    if kind == "c":
        return torch.full_like(x, node[1])
    if kind in _UNARY:
        return _TORCH_UNARY[kind](_evaluate(node[1], x, y))
    if kind == "powi":
        return _evaluate(node[1], x, y) ** node[2]
    a, b = node[1], node[2]
    # python scalars stay python scalars, as in the user's expression
    av = a[1] if a[0] == "c" else _evaluate(a, x, y)
    bv = b[1] if b[0] == "c" else _evaluate(b, x, y)
    if kind == "add":
        return av + bv
    if kind == "sub":
        return av - bv
    if kind == "mul":
        return av * bv
    return av / bv


# op codes of include/tfem_assembly.h (enum tfem_source_op)
OPS = {
    "PUSH_X": 1, "PUSH_Y": 2, "PUSH_C": 3, "ADD": 4, "SUB": 5, "SUB_R": 6, "MUL": 7, "DIV": 8,
    "DIV_R": 9, "ADD_C": 10, "MUL_C": 11, "RSUB_C": 12, "RDIV_C": 13, "NEG": 14, "ABS": 15,
    "POW_I": 16, "SIN": 17, "COS": 18, "EXP": 19, "SQRT": 20, "LOG": 21, "TANH": 22,
}
_UNARY_OPS = {"neg": "NEG", "abs": "ABS", "sin": "SIN", "cos": "COS", "exp": "EXP",
              "sqrt": "SQRT", "log": "LOG", "tanh": "TANH"}


def _need(node):
    """Stack entries the evaluation of `node` needs (Sethi-Ullman numbering)."""
    kind = node[0]
    if kind in ("x", "y", "c"):
        return 1
    if kind in _UNARY or kind == "powi":
        return _need(node[1])
    a, b = node[1], node[2]
    if a[0] == "c" or b[0] == "c":
        return _need(b if a[0] == "c" else a)
    na, nb = _need(a), _need(b)
    return max(na, nb) if na != nb else na + 1


_SCALED = ("PUSH_X", "PUSH_Y", "SIN", "COS", "EXP", "SQRT", "LOG", "TANH")  # ops with a factor


def _emit_scale(ops, factor):
    """top *= factor: folded into the operation that produced the top when that operation has
    a factor of its own that is still 1 (c * x, c * sin(.): the same product), else MUL_C."""
    if ops and ops[-1][0] in _SCALED and ops[-1][1] == 1.0:
        ops[-1] = (ops[-1][0], factor)
    else:
        ops.append(("MUL_C", factor))


def _emit(node, ops):
    kind = node[0]
    if kind == "x":
        ops.append(("PUSH_X", 1.0))
    elif kind == "y":
        ops.append(("PUSH_Y", 1.0))
    elif kind == "c":
        ops.append(("PUSH_C", node[1]))
    elif kind in _UNARY:
        _emit(node[1], ops)
        name = _UNARY_OPS[kind]
        ops.append((name, 1.0 if name in _SCALED else 0.0))
    elif kind == "powi":
        _emit(node[1], ops)
        ops.append(("POW_I", float(node[2])))
    else:
        a, b = node[1], node[2]
        if b[0] == "c":  # expr (op) constant
            _emit(a, ops)
            if kind == "add":
                ops.append(("ADD_C", b[1]))
            elif kind == "sub":
                ops.append(("ADD_C", -b[1]))
            elif kind == "mul":
                _emit_scale(ops, b[1])
            else:  # expr / c: a true division, as torch evaluates it
                ops.append(("PUSH_C", b[1]))
                ops.append(("DIV", 0.0))
        elif a[0] == "c":  # constant (op) expr
            _emit(b, ops)
            if kind == "mul":
                _emit_scale(ops, a[1])
            else:
                ops.append(({"add": "ADD_C", "sub": "RSUB_C", "div": "RDIV_C"}[kind], a[1]))
        elif _need(a) >= _need(b):
            _emit(a, ops)
            _emit(b, ops)
            ops.append(({"add": "ADD", "sub": "SUB", "mul": "MUL", "div": "DIV"}[kind], 0.0))
        else:  # the deeper operand first: the reversed operations keep the operand order
            _emit(b, ops)
            _emit(a, ops)
            ops.append(({"add": "ADD", "sub": "SUB_R", "mul": "MUL", "div": "DIV_R"}[kind], 0.0))


def compile_ops(node):
    """[(op name, constant)] of the field, or None when it does not fit the format."""
    if _need(node) > _native.SOURCE_STACK:
        return None
    ops = []
    _emit(node, ops)
    if len(ops) > _native.SOURCE_MAX_OPS:
        return None
    # `expr / c` pushes a constant on top of the operand: re-check the real depth
    depth = peak = 0
    for name, _ in ops:
        if name.startswith("PUSH"):
            depth += 1
        elif name in ("ADD", "SUB", "SUB_R", "MUL", "DIV", "DIV_R"):
            depth -= 1
        peak = max(peak, depth)
    return ops if peak <= _native.SOURCE_STACK else None


def compile_program(node):
    ops = compile_ops(node)
    if ops is None:
        return None
    program = _native.SourceProgram()
    program.n_ops = len(ops)
    for i, (name, constant) in enumerate(ops):
        program.ops[i] = OPS[name]
        program.consts[i] = float(constant)
    return program


def _dispatch_torch_function(func, args, kwargs):
    """torch.sin(x), torch.split(points, 1, dim=-1), torch.mul(f, v) ... on symbols."""
    name = getattr(func, "__name__", "")
    first = args[0] if args else None
    if isinstance(first, SourceExpr) and not kwargs:
        if name in _UNARY and len(args) == 1:
            return first._unary(name)
        if name in ("absolute",) and len(args) == 1:
            return first._unary("abs")
        if name in ("negative",) and len(args) == 1:
            return first._unary("neg")
        if name == "square" and len(args) == 1:
            return first.__pow__(2)
        if name in ("pow", "__pow__") and len(args) == 2:
            return first.__pow__(args[1])
        if name == "ones_like" and len(args) == 1:
            return SourceExpr(first._basis, ("c", 1.0))
        if name == "zeros_like" and len(args) == 1:
            return SourceExpr(first._basis, ("c", 0.0))
    if len(args) == 2 and not kwargs:
        left, right = args
        table = {
            "add": ("__add__", "__radd__"), "sub": ("__sub__", "__rsub__"), "subtract": ("__sub__", "__rsub__"),
            "mul": ("__mul__", "__rmul__"), "multiply": ("__mul__", "__rmul__"),
            "div": ("__truediv__", "__rtruediv__"), "true_divide": ("__truediv__", "__rtruediv__"),
            "divide": ("__truediv__", "__rtruediv__"), "matmul": ("__matmul__", "__rmatmul__"),
        }
        if name in table:
            forward, backward = table[name]
            if isinstance(left, _Symbol):
                return getattr(left, forward)(right)
            return getattr(right, backward)(left)
    return NotImplemented


class TracingBasis:
    """Stands in for the basis while the callable is traced: ``v``, ``v_grad`` and
    ``integration_points`` are symbols, every other attribute is the real basis's."""

    def __init__(self, basis):
        object.__setattr__(self, "_basis", basis)
        object.__setattr__(self, "v", ShapeFunctions(basis))
        object.__setattr__(self, "v_grad", ShapeGradients(basis))

    def __getattr__(self, name):
        basis = object.__getattribute__(self, "_basis")
        if name == "integration_points":  # built on first use: the geometry cache is lazy
            points = PointsSymbol(basis)
            object.__setattr__(self, "integration_points", points)
            return points
        return getattr(basis, name)

    def __setattr__(self, name, value):
        raise Untraceable("assignment on the basis")


def trace(function, basis, args, kwargs):
    """Call ``function`` once with the proxy basis.  Returns a BilinearExpr / LinearExpr /
    SourceExpr when the result is inside the vocabulary, else the integrand as a real tensor
    (the symbols turned into tensors along the way).  A callable the proxy cannot serve at all
    (it assigns attributes on the basis, tests ``isinstance`` ...) is called on the real basis,
    and remembered: the next call goes to the real basis directly.  That ONE time the callable
    has run twice (up to the point where the proxy failed, then on the real basis): side effects
    in front of that point -- random numbers drawn, counters -- happen twice on the first call.
    ``x ** n`` for integer 2 <= n <= 8 becomes repeated multiplication in a source program
    (torch calls pow): equal to rounding, not bit for bit."""
    refused = basis.__dict__.setdefault("_untraceable_callables", set())
    key = getattr(function, "__code__", None) or id(function)
    if key not in refused:
        try:
            result = function(TracingBasis(basis), *args, **kwargs)
        except Exception:  # noqa: BLE001 -- whatever went wrong with the proxy, the real basis decides
            refused.add(key)
        else:
            if isinstance(result, (BilinearExpr, LinearExpr, SourceExpr)):
                return result
            return materialize(result)
    return function(basis, *args, **kwargs)


__all__ = [
    "BilinearExpr", "LinearExpr", "SourceExpr", "PointsSymbol", "TracingBasis", "Untraceable",
    "compile_ops", "compile_program", "materialize", "trace", "OPS", "math",
]
