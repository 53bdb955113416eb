"""CSR global operator.

The reference assembles into a dense ``torch.zeros((N, N))`` (abstract_basis.py:81),
which is 2 TB at 5e5 DoFs.  The HIP path always produces CSR values; ``to_dense``
materialises the reference's layout when it fits.
"""

from __future__ import annotations

import torch

from . import _native


class CSRMatrix:
    """``crow_indices`` int64 (N+1), ``col_indices`` int32 (nnz, ascending per row),
    ``values`` (nnz); all on one device.

    ``perm`` (int64 (N,) or None): the operator is stored in a RENUMBERING of the caller's DoFs --
    row / column k of the stored pattern is DoF ``perm[k]`` of the caller.  The assembly engine
    renumbers a mesh whose vertex numbering has no locality along a space-filling curve once, at
    set-up, so that the row kernels stream coordinates and values contiguously; every method below
    takes and returns vectors in the CALLER's numbering, ``to_dense`` gives the caller's matrix, and
    ``caller_numbering()`` the plain CSR arrays in the caller's numbering."""

    def __init__(self, crow_indices, col_indices, values, shape, perm=None):
        self.crow_indices = crow_indices
        self.col_indices = col_indices
        self.values = values
        self.shape = tuple(shape)
        self.perm = perm
        self._inv = None

    def _inverse(self):
        if self._inv is None:
            self._inv = torch.empty_like(self.perm)
            self._inv[self.perm] = torch.arange(self.perm.numel(), device=self.perm.device)
        return self._inv

    def _stored(self):
        """The same stored arrays without the renumbering attached (vectors in stored numbering)."""
        return CSRMatrix(self.crow_indices, self.col_indices, self.values, self.shape)

    def caller_numbering(self):
        """Plain CSRMatrix (perm None) with rows, columns and values in the caller's numbering."""
        if self.perm is None:
            return self
        n = self.shape[0]
        counts = self.crow_indices[1:] - self.crow_indices[:-1]
        rows = self.perm[torch.repeat_interleave(torch.arange(n, device=self.device), counts)]
        cols = self.perm[self.col_indices.long()]
        order = torch.argsort(rows * n + cols)
        crow = torch.zeros(n + 1, dtype=torch.int64, device=self.device)
        crow[1:] = torch.cumsum(torch.bincount(rows, minlength=n), 0)
        return CSRMatrix(crow, cols[order].to(torch.int32), self.values[order], self.shape)

    @property
    def nnz(self):
        return int(self.values.shape[0])

    @property
    def device(self):
        return self.values.device

    @property
    def dtype(self):
        return self.values.dtype

    def to(self, device):
        return CSRMatrix(
            self.crow_indices.to(device), self.col_indices.to(device), self.values.to(device), self.shape,
            None if self.perm is None else self.perm.to(device),
        )

    def to_sparse_csr(self):
        if self.perm is not None:
            return self.caller_numbering().to_sparse_csr()
        return torch.sparse_csr_tensor(
            self.crow_indices, self.col_indices.to(torch.int64), self.values, size=self.shape
        )

    def to_dense(self):
        n = self.shape[0]
        if self.perm is not None:
            inv = self._inverse()
            return self._stored().to_dense()[inv][:, inv]
        if self.values.is_cuda:
            lib = _native.load()
            dense = torch.empty(self.shape, dtype=self.dtype, device=self.device)
            with torch.cuda.device(self.device):
                _native.check(
                    lib.tfem_csr_to_dense(
                        _native.ptr(self.crow_indices),
                        _native.ptr(self.col_indices),
                        _native.ptr(self.values),
                        self.values.element_size(),
                        n,
                        _native.ptr(dense),
                        _native.current_stream(self.device),
                    )
                )
            return dense
        # host copy of an already assembled operator: pure data movement, no arithmetic
        dense = torch.zeros(self.shape, dtype=self.dtype)
        rows = torch.repeat_interleave(torch.arange(n), self.crow_indices[1:] - self.crow_indices[:-1])
        dense[rows, self.col_indices.long()] = self.values
        return dense

    def matvec(self, x):
        """A @ x for x of shape (N,) or (N, 1): libtfem_hip's CSR kernel on the GPU."""
        if self.perm is not None:
            flat = x.to(self.device, self.dtype).reshape(-1)
            return self._stored().matvec(flat[self.perm])[self._inverse()].reshape(x.shape)
        if not self.values.is_cuda:
            return self.to_sparse_csr() @ x
        lib = _native.load()
        flat = x.to(self.device, self.dtype).reshape(-1).contiguous()
        if flat.shape[0] != self.shape[1]:
            raise ValueError(f"matvec: x has {flat.shape[0]} entries, the operator {self.shape[1]} columns")
        y = torch.empty(self.shape[0], dtype=self.dtype, device=self.device)
        with torch.cuda.device(self.device):
            _native.check(
                lib.tfem_csr_spmv(
                    _native.ptr(self.crow_indices), _native.ptr(self.col_indices), _native.ptr(self.values),
                    self.values.element_size(), self.shape[0], _native.ptr(flat), _native.ptr(y),
                    _native.current_stream(self.device),
                )
            )
        return y.reshape(x.shape)

    def diagonal(self):
        """Diagonal entries (0 where a row stores none)."""
        if self.perm is not None:
            return self._stored().diagonal()[self._inverse()]
        n = self.shape[0]
        counts = self.crow_indices[1:] - self.crow_indices[:-1]
        rows = torch.repeat_interleave(torch.arange(n, device=self.device), counts)
        hit = self.col_indices.long() == rows
        diag = torch.zeros(n, dtype=self.dtype, device=self.device)
        diag[rows[hit]] = self.values[hit]
        return diag

    def solve_cg(self, b, free=None, x0=None, rtol=1e-12, maxiter=None):
        """Jacobi-preconditioned conjugate gradients for the symmetric positive definite
        operator restricted to the DoFs `free` (index tensor; the others keep x0's values, 0 by
        default: homogeneous Dirichlet rows and columns are simply masked, no submatrix is
        formed).  Returns (x, iterations, relative residual).  Stands where the reference's
        dense `reduce` + `torch.linalg.solve` (abstract_basis.py:114-117,177-195) stops being
        possible."""
        if self.perm is not None:  # solve in the stored numbering, vectors translated at the boundary
            inv = self._inverse()
            to_stored = lambda v: None if v is None else v.to(self.device, self.dtype).reshape(-1)[self.perm]  # noqa: E731
            free_s = None if free is None else inv[free.to(self.device).reshape(-1)]
            x, it, res = self._stored().solve_cg(to_stored(b), free_s, to_stored(x0), rtol, maxiter)
            return x[inv].reshape(b.shape), it, res
        n = self.shape[0]
        shape = b.shape
        b = b.to(self.device, self.dtype).reshape(-1)
        mask = torch.ones(n, dtype=self.dtype, device=self.device)
        if free is not None:
            mask.zero_()
            mask[free.to(self.device).reshape(-1)] = 1
        x = torch.zeros(n, dtype=self.dtype, device=self.device) if x0 is None else x0.to(self.device, self.dtype).reshape(-1).clone()
        inv_diag = mask / torch.where(self.diagonal() != 0, self.diagonal(), torch.ones_like(mask))
        r = mask * (b - self.matvec(x))
        x = x.clone()
        z = inv_diag * r
        p = z.clone()
        rz = torch.dot(r, z)
        b_norm = torch.linalg.vector_norm(mask * b).clamp_min(torch.finfo(self.dtype).tiny)
        maxiter = maxiter or 10 * n
        it, res = 0, float(torch.linalg.vector_norm(r) / b_norm)
        while it < maxiter and res > rtol:
            ap = mask * self.matvec(p)
            alpha = rz / torch.dot(p, ap)
            x += alpha * p
            r -= alpha * ap
            z = inv_diag * r
            rz_new = torch.dot(r, z)
            p = z + (rz_new / rz) * p
            rz = rz_new
            it += 1
            if it % 25 == 0 or it == maxiter:  # one host synchronisation every 25 iterations
                res = float(torch.linalg.vector_norm(r) / b_norm)
        res = float(torch.linalg.vector_norm(r) / b_norm)
        return x.reshape(shape), it, res

    def __repr__(self):
        extra = "" if self.perm is None else ", stored in a renumbering of the DoFs"
        return f"CSRMatrix(shape={self.shape}, nnz={self.nnz}, dtype={self.dtype}, device={self.device}{extra})"
