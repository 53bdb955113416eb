"""Poisson assembly with the reference's forms (stiffness + mass, 2 pi^2 sin sin load,
integral of the squared source) on a synthetic triangle mesh -- the flow of the reference's
tests/test_assembly.py:62-98 with scikit-fem replaced by closed-form checks.

    python examples/poisson_assembly.py [n]
"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_fem import Basis, ElementTri, MeshTri  # noqa: E402  (the MI355X-native package)
from pytorch_fem_solver_amd import meshgen  # noqa: E402

torch.set_default_device("cuda" if torch.cuda.is_available() else "cpu")
torch.set_default_dtype(torch.float64)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mesh = MeshTri(triangulation=meshgen.unit_square(n, 0.25, 0))
basis_h = Basis(mesh, ElementTri(polynomial_order=1, integration_order=3))


def bilinear(basis):
    return basis.v_grad @ basis.v_grad.mT + basis.v @ basis.v.mT


def rhs(x, y):
    return 2.0 * math.pi**2 * torch.sin(math.pi * x) * torch.sin(math.pi * y)


def residual(basis):
    x, y = torch.split(basis.integration_points, 1, dim=-1)
    return rhs(x, y) * basis.v


def rhs_functional_form(basis):
    x, y = torch.split(basis.integration_points, 1, dim=-1)
    return rhs(x, y) ** 2


A = basis_h.integrate_bilinear_form(bilinear, layout="csr")
b = basis_h.integrate_linear_form(residual)
energy = basis_h.integrate_functional(rhs_functional_form)

ones = torch.ones(A.shape[0], 1)
print(f"{mesh['cells', 'vertices'].shape[0]} elements, {A.shape[0]} DoFs, nnz {A.nnz}")
print("1^T (K + M) 1 = |Omega| :", float((ones.T @ A.matvec(ones)).item()))  # stiffness rows sum to 0
print("sum f                   :", float(b.sum()), "(exact integral 8)")
print("integral of f^2         :", float(energy.sum()), "(exact pi^4)")
K = basis_h.integrate_bilinear_form(lambda basis: basis.v_grad @ basis.v_grad.mT)
u = basis_h.solve(K, basis_h.solution_tensor(), b)
xy = mesh["vertices", "coordinates"]
exact = torch.sin(math.pi * xy[:, [0]]) * torch.sin(math.pi * xy[:, [1]])
print("max nodal error of u_h  :", float((u - exact).abs().max()))
