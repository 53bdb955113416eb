"""``AbstractBasis``: the reference's integration API on top of the HIP assembly path.

Mirror of reference torch_fem/basis/abstract_basis.py.  What the reference does with a
chain of torch expressions per call -- geometry at quadrature points (:42-63),
``(integrand * dx).sum(-3)`` (:83,:104,:72) and ``index_put_(accumulate=True)`` into a
dense target (:81-91,:102-110) -- happens here inside libtfem_hip (csrc/), for tensors
on the GPU directly and for CPU tensors by staging through the GPU.  No torch
implementation of those steps exists in this class.
"""

from __future__ import annotations

import abc
import os
import warnings

import torch

from .. import _native
from ..sparse import CSRMatrix
from . import forms

#: above this many bytes the dense (N, N) layout of the reference is not materialised
DENSE_LIMIT_BYTES = int(os.environ.get("TORCH_FEM_DENSE_LIMIT_BYTES", str(4 << 30)))


class _LinearFormFunction(torch.autograd.Function):
    """Differentiable quadrature-reduce + scatter (the VPINN training step differentiates
    through integrate_linear_form, reference examples/example_weak.py:132-152)."""

    @staticmethod
    def forward(ctx, integrand, basis):
        ctx.basis = basis
        ctx.in_shape = integrand.shape
        ctx.in_device = integrand.device
        return basis._engine.reduce_linear(integrand.detach(), basis._dx)

    @staticmethod
    def backward(ctx, grad_out):
        basis = ctx.basis
        # the caller's DoF ids (the engine may work in a renumbering of its own)
        conn = basis._global_dofs4elements.to(grad_out.device).long()
        conn = conn.reshape(-1, conn.shape[-1])
        dx = basis._dx.to(grad_out.device)
        lead = tuple(dx.shape[:-3])
        g = grad_out.reshape(-1)[conn].reshape(lead + (1, conn.shape[-1], 1))
        grad_integrand = g * dx
        return grad_integrand.sum_to_size(ctx.in_shape).to(ctx.in_device), None


class _FunctionalFunction(torch.autograd.Function):
    """Differentiable per-element integral (abstract_basis.py:65-72): the reference's
    ``(f * dx).sum(-3).sum(-2)`` stays in the autograd graph and is the training loss of
    examples/example_loss_is_error.py:101-106, example_jump.py:147-151.  Forward: one
    tfem_reduce_functional launch; backward: the cotangent of every element times dx."""

    @staticmethod
    def forward(ctx, integrand, basis):
        ctx.basis = basis
        ctx.in_shape = integrand.shape
        ctx.in_device = integrand.device
        return basis._engine.reduce_functional(integrand.detach(), basis._dx)

    @staticmethod
    def backward(ctx, grad_out):
        dx = ctx.basis._dx.to(grad_out.device)  # (..., Q, 1, 1)
        full = torch.broadcast_shapes(tuple(ctx.in_shape), tuple(dx.shape))
        grad_integrand = (grad_out[..., None, None] * dx).expand(full)
        return grad_integrand.sum_to_size(ctx.in_shape).to(ctx.in_device), None


class _ResidualFormFunction(torch.autograd.Function):
    """The VPINN residual form ``f * v + s * v_grad @ g.mT`` (examples/example_weak.py:64-75), fused:
    forward = tfem_p1_residual_local + gather, backward = tfem_p1_residual_backward; neither
    direction materialises the (N_T, Q, 3, 1) integrand.  `coefficient` is a tensor (..., Q, 1, 1) or
    None (then `program`, a source program, or no source at all); `flux` a tensor (..., Q, 1, 2)."""

    @staticmethod
    def forward(ctx, coefficient, flux, basis, program, flux_sign):
        engine = basis._engine
        fq = None
        if coefficient is not None:
            fq = basis._source_values(coefficient.detach().to(engine.device, engine.dtype))
        ctx.basis, ctx.flux_sign = basis, flux_sign
        ctx.coefficient_meta = None if coefficient is None else (coefficient.shape, coefficient.device)
        ctx.flux_meta = (flux.shape, flux.device)
        return engine.residual(fq, program, engine._flat_flux(flux), flux_sign)

    @staticmethod
    def backward(ctx, grad_out):
        engine = ctx.basis._engine
        want_fq = ctx.coefficient_meta is not None and ctx.needs_input_grad[0]
        want_flux = ctx.needs_input_grad[1]
        grad_fq, grad_flux = engine.residual_backward(grad_out, ctx.flux_sign, want_fq, want_flux)
        lead = tuple(engine.lead_shape)
        g_coefficient = g_flux = None
        if want_fq:
            shape, device = ctx.coefficient_meta
            g_coefficient = grad_fq.reshape(lead + (engine.n_quad, 1, 1)).sum_to_size(shape).to(device)
        if want_flux:
            shape, device = ctx.flux_meta
            g_flux = grad_flux.reshape(lead + (engine.n_quad, 1, 2)).sum_to_size(shape).to(device)
        return g_coefficient, g_flux, None, None, None


class AbstractBasis(abc.ABC):
    """Finite-element basis on a mesh (abstract_basis.py:10-40)."""

    #: geometry attributes computed on first access by one tfem_tri_geometry launch
    _LAZY = ("v_grad", "integration_points", "_dx", "_inv_map_jacobian")

    def __init__(self, mesh, element):
        self._element = element
        self.mesh = mesh
        self._geometry_cache = {}
        (
            self._coords4global_dofs,
            self._global_dofs4elements,
            self._nodes4boundary_dofs,
            self._coords4elements,
        ) = self._compute_dofs(mesh, element)
        self._basis_parameters = self._compute_basis_parameters(
            self._coords4global_dofs, self._global_dofs4elements, self._nodes4boundary_dofs
        )
        self._engine = self._make_engine(mesh, element)
        self.v = self._compute_shape_values(element)

    # ---- lazily materialised geometry cache (abstract_basis.py:42-63) -------------
    def __getattr__(self, name):
        if name in AbstractBasis._LAZY:
            cache = self.__dict__.get("_geometry_cache")
            if cache is None:
                raise AttributeError(name)
            if name not in cache:
                cache.update(self._compute_integral_values(self.mesh, self._element))
            return cache[name]
        raise AttributeError(f"{type(self).__name__!s} has no attribute {name!r}")

    def __setattr__(self, name, value):
        if name in AbstractBasis._LAZY:
            self._geometry_cache[name] = value
        else:
            object.__setattr__(self, name, value)

    def _compute_shape_values(self, element):
        bar = element.compute_barycentric_coordinates(element.gaussian_nodes)
        v, _ = element.compute_shape_functions(
            bar, torch.zeros((1, 1, 2, 2), dtype=bar.dtype, device=bar.device)
        )
        return v

    @abc.abstractmethod
    def _make_engine(self, mesh, element):
        ...

    @abc.abstractmethod
    def _compute_integral_values(self, mesh, element) -> dict:
        """Return {'v_grad','integration_points','_dx','_inv_map_jacobian'} in the
        reference's shapes."""

    @abc.abstractmethod
    def _compute_dofs(self, mesh, element):
        ...

    @abc.abstractmethod
    def _compute_basis_parameters(self, coords4global_dofs, global_dofs4elements, nodes4boundary_dofs):
        ...

    # ---- integration API ---------------------------------------------------------------
    def integrate_functional(self, function, *args, **kwargs):
        """Per-element integral of ``function(basis, ...)`` (abstract_basis.py:65-72);
        differentiable in the integrand."""
        integrand = forms.materialize(forms.trace(function, self, args, kwargs))
        if integrand.requires_grad and torch.is_grad_enabled():
            out = _FunctionalFunction.apply(integrand, self)
        else:
            out = self._engine.reduce_functional(integrand.detach(), self._dx)
        return self._engine._home(out)

    def integrate_bilinear_form(self, function, *args, layout=None, **kwargs):
        """Global operator of a bilinear form (abstract_basis.py:74-93).

        ``layout``: "dense" (the reference's (N, N) tensor), "csr" (``CSRMatrix``) or None
        = dense while it fits ``DENSE_LIMIT_BYTES``, CSR beyond.
        """
        expr = forms.trace(function, self, args, kwargs)
        if isinstance(expr, forms.BilinearExpr):
            vals = self._engine.bilinear(expr.alpha, expr.beta)
        else:
            integrand = forms.materialize(expr)
            if integrand.requires_grad and torch.is_grad_enabled():
                raise NotImplementedError(
                    "integrate_bilinear_form: the integrand carries autograd history, but the "
                    "assembled operator is written by the HIP kernels and is not differentiable; "
                    "detach() the integrand, or differentiate a linear form / functional instead"
                )
            vals = self._engine.reduce_bilinear(integrand, self._dx)
        return self._finish_matrix(vals, layout)

    def assemble_system(self, bilinear, linear, *args, layout=None, **kwargs):
        """``(integrate_bilinear_form(bilinear, ...), integrate_linear_form(linear, ...))`` -- the
        pair every solve of the reference asks for (examples/example_fractures_fem.py:239-241,
        tests/test_assembly.py:86-93) -- in ONE launch when both callables are in the fused
        vocabulary: ``alpha * v_grad @ v_grad.mT + beta * v @ v.mT`` and ``f * v`` with ``f`` an
        expression of the integration points (evaluated inside the launch) or a tensor of source
        values.  Anything else: the two calls one after the other, same results."""
        a_expr = forms.trace(bilinear, self, args, kwargs)
        l_expr = forms.trace(linear, self, args, kwargs)
        fused = None
        if isinstance(a_expr, forms.BilinearExpr) and isinstance(l_expr, forms.LinearExpr) and l_expr.flux is None:
            coefficient = l_expr.coefficient
            if isinstance(coefficient, forms.SourceExpr):
                program = self._source_program(coefficient)
                if program is not None:
                    fused = self._engine.assemble_system(a_expr.alpha, a_expr.beta, source=program)
                else:
                    coefficient = coefficient.materialize()
            if fused is None and torch.is_tensor(coefficient) and not coefficient.requires_grad:
                values = self._source_values(coefficient)
                if values is not None:
                    fused = self._engine.assemble_system(a_expr.alpha, a_expr.beta, values)
        if fused is None:
            return (self.integrate_bilinear_form(bilinear, *args, layout=layout, **kwargs),
                    self.integrate_linear_form(linear, *args, **kwargs))
        vals, f = fused
        return self._finish_matrix(vals, layout), self._engine._home(f).reshape(-1, 1)

    def _source_program(self, coefficient):
        """The source program of a traced coefficient, or None -- said ONCE per basis when it is the
        program's limits that send the expression to torch (include/tfem_assembly.h: 32 operations,
        a stack of 4): the results are the same, the step then reads 8 Q bytes per element more."""
        if not self._engine.supports_source():
            return None
        program = coefficient.program()
        if program is None and not self.__dict__.get("_warned_source_limits"):
            self.__dict__["_warned_source_limits"] = True
            warnings.warn(
                "the source expression does not fit a source program (more than "
                f"{_native.SOURCE_MAX_OPS} operations or {_native.SOURCE_STACK} values at a time): torch evaluates it at the "
                "cached integration points and the assembly launch reads the values from memory",
                RuntimeWarning, stacklevel=3)
        return program

    def _finish_matrix(self, vals, layout):
        matrix = self._engine.wrap_csr(vals)
        n = matrix.shape[0]
        if layout is None:
            layout = "dense" if n * n * vals.element_size() <= DENSE_LIMIT_BYTES else "csr"
        if layout == "csr":
            return self._engine.wrap_csr_home(vals)
        if layout != "dense":
            raise ValueError(f"unknown layout {layout!r}")
        return self._engine._home(matrix.to_dense())

    def integrate_linear_form(self, function, *args, **kwargs):
        """Global vector of a linear form, shape (N, 1) (abstract_basis.py:95-112)."""
        expr = forms.trace(function, self, args, kwargs)
        if isinstance(expr, forms.LinearExpr) and expr.flux is None:
            coefficient = expr.coefficient
            if isinstance(coefficient, forms.SourceExpr):
                # f(x, y) recorded by the tracer: evaluated inside the assembly launch
                program = self._source_program(coefficient)
                if program is not None:
                    return self._engine._home(self._engine.load_source(program)).reshape(-1, 1)
                coefficient = coefficient.materialize()
            if not coefficient.requires_grad:
                values = self._source_values(coefficient)
                if values is not None:
                    return self._engine._home(self._engine.load(values)).reshape(-1, 1)
        if isinstance(expr, forms.LinearExpr) and expr.flux is not None and self._engine.supports_residual():
            out = self._residual_form(expr)
            if out is not None:
                return self._engine._home(out).reshape(-1, 1)
        integrand = forms.materialize(expr)
        if integrand.requires_grad and torch.is_grad_enabled():
            out = _LinearFormFunction.apply(integrand, self)
        else:
            out = self._engine.reduce_linear(integrand.detach(), self._dx)
        return self._engine._home(out).reshape(-1, 1)

    def _residual_form(self, expr):
        """f * v + s * v_grad @ g.mT through the fused residual kernels, or None when the
        operands do not have the reference's shapes (then torch evaluates the integrand)."""
        engine = self._engine
        if engine._flat_flux(expr.flux) is None:
            return None
        coefficient, program = expr.coefficient, None
        if isinstance(coefficient, forms.SourceExpr):
            program = coefficient.program()
            coefficient = None if program is not None else coefficient.materialize()
        if coefficient is not None and self._source_values(coefficient) is None:
            return None
        return _ResidualFormFunction.apply(coefficient, expr.flux, self, program, expr.flux_sign)

    def _source_values(self, coefficient):
        """(E, Q) source values if ``coefficient`` broadcasts to (..., Q, 1, 1), else None."""
        lead = tuple(self._engine.lead_shape)
        want = lead + (self._engine.n_quad, 1, 1)
        try:
            full = torch.broadcast_shapes(tuple(coefficient.shape), want)
        except RuntimeError:
            return None
        if tuple(full) != want:
            return None
        return coefficient.expand(want).reshape(-1, self._engine.n_quad)

    def reduce(self, tensor):
        """Restrict to interior DoFs (abstract_basis.py:114-117)."""
        idx = self._basis_parameters["inner_dofs"]
        if isinstance(tensor, CSRMatrix):
            tensor = tensor.to_dense()
        return tensor[idx, :][:, idx] if tensor.size(-1) != 1 else tensor[idx]

    def reshape_for_assembly(self, local_matrices, form):
        """abstract_basis.py:162-171"""
        if form == "bilinear":
            return local_matrices.reshape(-1)
        if form == "linear":
            return local_matrices.reshape(-1, 1)
        raise NotImplementedError(f"Unknown form type: {format(form)}")

    def solution_tensor(self):
        """Zero vector (N, 1) (abstract_basis.py:173-175)."""
        return torch.zeros(self._basis_parameters["linear_form_shape"])

    #: a CSR operator with more rows than this is solved by conjugate gradients on the CSR
    #: values instead of the reference's dense solve (its dense copy would not fit)
    DENSE_SOLVE_LIMIT = 20000

    def solve(self, matrix, solution, vector, only_inner_dofs=True, method=None):
        """Dense solve on the interior DoFs (abstract_basis.py:177-195).  A CSRMatrix beyond
        DENSE_SOLVE_LIMIT rows (or method="cg") is solved by Jacobi-preconditioned conjugate
        gradients on the CSR values (CSRMatrix.solve_cg; symmetric positive definite forms):
        the step after the assembly for operators the reference cannot hold."""
        if isinstance(matrix, CSRMatrix) and (method == "cg" or (method is None and matrix.shape[0] > self.DENSE_SOLVE_LIMIT)):
            free = self._basis_parameters["inner_dofs"] if only_inner_dofs is True else None
            x, _, _ = matrix.solve_cg(vector, free=free)
            if free is None:
                solution += x.reshape(solution.shape).to(solution.device)
            else:
                solution[free] += x.reshape(solution.shape)[free].to(solution.device)
            return solution
        if only_inner_dofs is True:
            matrix = self.reduce(matrix)
            vector = self.reduce(vector)
        elif isinstance(matrix, CSRMatrix):
            matrix = matrix.to_dense()
        solution[self._basis_parameters["inner_dofs"]] += torch.linalg.solve(matrix, vector)
        return solution


class LazyIndexDict(dict):
    """``_basis_parameters`` with the 2 x (n^2 N_T) dense scatter indices of the reference
    (basis.py:73-76) built only if somebody asks for them."""

    def __init__(self, *args, connectivity=None, **kwargs):
        super().__init__(*args, **kwargs)
        self._connectivity = connectivity

    def __missing__(self, key):
        if key != "bilinear_form_idx":
            raise KeyError(key)
        conn = self._connectivity.reshape(-1, self._connectivity.shape[-1])
        n = conn.shape[-1]
        value = (conn.repeat(1, n).reshape(-1), conn.repeat_interleave(n).reshape(-1))
        self[key] = value
        return value
