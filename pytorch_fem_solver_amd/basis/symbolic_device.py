"""Symbolic phase on the device: the CSR pattern of the operator and the element -> entry map,
built from a device-resident connectivity with sorts and scans -- no host round trip.

The reference has no sparse structure at all: it scatters through the index tensors of
basis.py:64-85 (rows = conn repeated, cols = conn interleaved) into a dense (N, N) target.  The
pattern below is exactly the set of (row, col) pairs those index tensors name, stored CSR with
ascending columns -- the same bytes the multi-threaded host builder produces
(csrc/tfem_host.cpp, tfem_csr_pattern_*; tests/test_hip_device_builders.py compares them).

Device code here means torch's device sort / unique / searchsorted (rocPRIM radix sorts and
scans underneath): set-up work, once per mesh, outside the timed path.
"""

from __future__ import annotations

import torch


def pattern_device(conn_dof, n_dofs):
    """rowptr int64 (N+1), colind int32 (nnz) on conn_dof's device.

    conn_dof (E, n) int32 / int64: the entries of the operator are the pairs (conn[e, j], conn[e, i])
    over all elements (basis.py:73-76); a pair named by several elements is one entry."""
    conn = conn_dof.reshape(-1, conn_dof.shape[-1]).to(torch.int64)
    n = int(n_dofs)
    e, nl = conn.shape
    if e == 0:
        return torch.zeros(n + 1, dtype=torch.int64, device=conn.device), torch.zeros(0, dtype=torch.int32, device=conn.device)
    # one key per ordered pair: row * N + col; the set is symmetric, so which of the two is the row
    # does not matter for the pattern (it does for the slots below)
    keys = (conn.unsqueeze(2) * n + conn.unsqueeze(1)).reshape(-1)
    keys = torch.unique(keys)  # sorted: by row, then by column
    rows = torch.div(keys, n, rounding_mode="floor")
    colind = (keys - rows * n).to(torch.int32)
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=conn.device)
    rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=n), 0)
    return rowptr, colind


def slots_device(conn_dof, n_dofs, rowptr, colind):
    """slots int32 (E, n, n): slots[e, i, j] = CSR position of (row conn[e, j], col conn[e, i]) --
    the reference's transposed scatter convention (local[i, j] -> A[conn[j], conn[i]], basis.py:73-76
    with abstract_basis.py:166-167)."""
    conn = conn_dof.reshape(-1, conn_dof.shape[-1]).to(torch.int64)
    n = int(n_dofs)
    counts = rowptr[1:] - rowptr[:-1]
    row_of = torch.repeat_interleave(torch.arange(n, device=conn.device), counts)
    keys = row_of * n + colind.to(torch.int64)  # ascending by construction
    want = conn.unsqueeze(1) * n + conn.unsqueeze(2)  # [e, i, j] -> conn[e, j] * N + conn[e, i]
    pos = torch.searchsorted(keys, want.reshape(-1))
    return pos.to(torch.int32).reshape(conn.shape[0], conn.shape[1], conn.shape[1])
