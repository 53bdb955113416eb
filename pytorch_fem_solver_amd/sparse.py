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
    ``values`` (nnz); all on one device."""

    def __init__(self, crow_indices, col_indices, values, shape):
        self.crow_indices = crow_indices
        self.col_indices = col_indices
        self.values = values
        self.shape = tuple(shape)

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
            self.crow_indices.to(device), self.col_indices.to(device), self.values.to(device), self.shape
        )

    def to_sparse_csr(self):
        return torch.sparse_csr_tensor(
            self.crow_indices, self.col_indices.to(torch.int64), self.values, size=self.shape
        )

    def to_dense(self):
        n = self.shape[0]
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
        return self.to_sparse_csr() @ x

    def __repr__(self):
        return f"CSRMatrix(shape={self.shape}, nnz={self.nnz}, dtype={self.dtype}, device={self.device})"
