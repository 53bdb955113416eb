"""CPU ORACLE (torch) -- test infrastructure, NOT product code.

The reference's element-wise assembly path restated as the SAME SEQUENCE OF TORCH OPERATIONS
the reference runs (SURVEY.md section 8, rows a-1 ... a-9), on CPU tensors.  This is the CPU
baseline SURVEY.md 8(d) / BASELINE.md section 4 specify: the reference's own op sequence timed
with ``torch.set_num_threads(os.cpu_count())`` on the GPU box's host cores, in three stages
(geometry cache / local integration / global scatter) -- `/root/reference` itself cannot
travel to that box.  Pinned to the fixtures the reference generated
(tests/test_oracle_golden.py::test_torch_restatement_*).

Only tests/ and bench.py's ``cpu_baseline`` leg import this module (same rule as
oracle/assembly_oracle.py).  One deviation from the reference, flagged where it is made: the
reference scatters into a DENSE (N, N) tensor (abstract_basis.py:81), which is 2 TB at 5e5
DoFs; `scatter_bilinear_csr` puts the same values through the same `index_put_(accumulate=True)`
into the CSR value array instead ("not a reference capability").
"""

from __future__ import annotations

import math

import torch

# element_tri.py:10-12 -- re-created per use, at the default dtype, as the reference's property is
def barycentric_grad():
    return torch.tensor([[-1.0, -1.0], [1.0, 0.0], [0.0, 1.0]])


def gauss_values(integration_order):
    """element_tri.py:77-130: nodes (Q, 2), weights (Q, 1, 1); the reference's literals."""
    table = {
        1: ([[1 / 3, 1 / 3]], [1.0]),
        2: ([[1 / 6, 1 / 6], [2 / 3, 1 / 6], [1 / 6, 2 / 3]], [1 / 3, 1 / 3, 1 / 3]),
        3: ([[1 / 3, 1 / 3], [0.6, 0.2], [0.2, 0.6], [0.2, 0.2]], [-9 / 16, 25 / 48, 25 / 48, 25 / 48]),
        4: (
            [
                [0.816847572980459, 0.091576213509771],
                [0.091576213509771, 0.816847572980459],
                [0.091576213509771, 0.091576213509771],
                [0.108103018168070, 0.445948490915965],
                [0.445948490915965, 0.108103018168070],
                [0.445948490915965, 0.445948490915965],
            ],
            [0.109951743655322] * 3 + [0.223381589678011] * 3,
        ),
    }
    if integration_order not in table:
        raise NotImplementedError("Integration order not implemented")
    nodes, weights = table[integration_order]
    return torch.tensor(nodes), torch.tensor([[[w]] for w in weights])


def geometry_cache(vertices, triangles, integration_order):
    """Stage 1, what `Basis.__init__` caches for P1 (abstract_basis.py:42-63).

    vertices (N_v, 2), triangles (N_T, 3) -> dict of v (Q,3,1), v_grad (N_T,1,3,2),
    integration_points (N_T,Q,1,2), dx (N_T,Q,1,1), inv_map_jacobian (N_T,1,2,2),
    cells (N_T,3,2)."""
    cells = vertices[triangles.long()]                          # a-1  abstract_mesh.py:257-262
    grad = barycentric_grad()
    map_jacobian = cells.mT @ grad                              # a-2  basis.py:87-88
    ab, cd = torch.split(map_jacobian, 1, dim=-2)               # a-3  element_tri.py:132-145
    a, b = torch.split(ab, 1, dim=-1)
    c, d = torch.split(cd, 1, dim=-1)
    det = (a * d - b * c).unsqueeze(-3)
    inv = (1 / det) * torch.stack([torch.concat([d, -b], dim=-1), torch.concat([-c, a], dim=-1)], dim=-2)
    nodes, weights = gauss_values(integration_order)            # a-4
    bar = torch.stack([1.0 - nodes[..., [0]] - nodes[..., [1]], nodes[..., [0]], nodes[..., [1]]], dim=-2)  # a-5 :23-26
    v = bar                                                     # element_tri.py:39
    v_grad = grad @ inv                                         # element_tri.py:41
    points = bar.mT @ cells.unsqueeze(-3)                       # a-6  basis.py:90-91
    dx = 0.5 * weights * det                                    # basis.py:93-96
    return {"v": v, "v_grad": v_grad, "integration_points": points, "dx": dx, "inv_map_jacobian": inv,
            "cells": cells}


def scatter_indices(triangles):
    """a-9, basis.py:64-85: rows_idx, cols_idx (9 N_T,), form_idx (3 N_T,) -- the transposed
    convention local[i, j] -> A[conn[j], conn[i]] follows from these two lines."""
    conn = triangles.unsqueeze(0) if triangles.dim() == 2 else triangles
    n_local = conn.size(-1)
    rows_idx = conn.repeat(1, 1, n_local).reshape(-1)
    cols_idx = conn.repeat_interleave(n_local).reshape(-1)
    return rows_idx, cols_idx, conn.reshape(-1)


def stiffness_integrand(geo):       # examples/example_fractures_fem.py:112-116
    return geo["v_grad"] @ geo["v_grad"].mT


def stiffness_mass_integrand(geo):  # tests/test_assembly.py:68-73
    return geo["v_grad"] @ geo["v_grad"].mT + geo["v"] @ geo["v"].mT


def rhs(x, y):                      # tests/test_assembly.py:75-77
    return 2.0 * math.pi**2 * torch.sin(math.pi * x) * torch.sin(math.pi * y)


def load_integrand(geo):            # tests/test_assembly.py:79-84
    x, y = torch.split(geo["integration_points"], 1, dim=-1)
    return rhs(x, y) * geo["v"]


def local_bilinear(geo, integrand=stiffness_integrand):
    """Stage 2 (a-7, a-8): `(function(self) * self._dx).sum(-3)`, abstract_basis.py:83 -> (N_T, 3, 3)."""
    return (integrand(geo) * geo["dx"]).sum(-3)


def local_linear(geo, integrand=load_integrand):
    """abstract_basis.py:104 -> (N_T, 3, 1); the user's f is evaluated inside, on every call."""
    return (integrand(geo) * geo["dx"]).sum(-3)


def functional(geo, integrand):
    """abstract_basis.py:65-72 -> (N_T, 1)."""
    return (integrand(geo) * geo["dx"]).sum(-3).sum(-2)


def scatter_bilinear_dense(local, triangles, n_dofs):
    """Stage 3 as the reference runs it (abstract_basis.py:81-91, reshape :162-167): dense target."""
    rows_idx, cols_idx, _ = scatter_indices(triangles)
    out = torch.zeros((n_dofs, n_dofs), dtype=local.dtype)
    out.index_put_((rows_idx.long(), cols_idx.long()), local.reshape(-1), accumulate=True)
    return out


def scatter_bilinear_csr(local, slots, nnz):
    """Stage 3 where the dense target cannot exist: the same flattened values, the same
    `index_put_(accumulate=True)`, into the CSR value array through the element -> slot map
    (slots[e, i, j] = CSR position of A[conn[j], conn[i]]).  Not a reference capability."""
    out = torch.zeros(nnz, dtype=local.dtype)
    out.index_put_((slots.reshape(-1).long(),), local.reshape(-1), accumulate=True)
    return out


def scatter_linear(local, triangles, n_dofs):
    """abstract_basis.py:102-110 -> (N, 1)."""
    _, _, form_idx = scatter_indices(triangles)
    out = torch.zeros((n_dofs, 1), dtype=local.dtype)
    out.index_put_((form_idx.long(),), local.reshape(-1, 1), accumulate=True)
    return out
