"""Poisson problem far beyond what the reference can hold: assemble K (CSR) and f on a mesh
of ~2e6 elements with the reference's forms and solve the 1e6-DoF system by conjugate
gradients on the assembled operator -- the reference's dense (N, N) matrix would take 8 TB.
Then the step after the solve in the reference's fracture example (example_fractures_fem.py):
the jump of the normal derivative over the interior edges, the edge part of a residual error
estimator, with the reference's InteriorEdgesBasis / Basis.interpolate calls.

    python examples/poisson_large_cg.py [n]
"""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_fem import Basis, ElementLine, ElementTri, InteriorEdgesBasis, MeshTri  # noqa: E402  (the MI355X-native package)
from pytorch_fem_solver_amd import meshgen  # noqa: E402

torch.set_default_device("cuda")
torch.set_default_dtype(torch.float64)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
mesh_np = meshgen.unit_square(n, 0.25, 0)
mesh_np.pop("neighbors")  # edge -> cells table matched edge by edge (mesh/topology.py)
mesh = MeshTri(triangulation=mesh_np)
basis = Basis(mesh, ElementTri(polynomial_order=1, integration_order=3))


def stiffness(b):
    return b.v_grad @ b.v_grad.mT


def load(b):
    x, y = torch.split(b.integration_points, 1, dim=-1)
    return 2.0 * math.pi**2 * torch.sin(math.pi * x) * torch.sin(math.pi * y) * b.v


K = basis.integrate_bilinear_form(stiffness, layout="csr")  # plans are built here, once
f = basis.integrate_linear_form(load)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = basis.integrate_bilinear_form(stiffness, layout="csr")
f = basis.integrate_linear_form(load)
torch.cuda.synchronize()
t1 = time.perf_counter()
u = basis.solve(K, basis.solution_tensor(), f, method="cg")
torch.cuda.synchronize()
t2 = time.perf_counter()
pts = torch.as_tensor(mesh_np["vertices"])
exact = (torch.sin(math.pi * pts[:, 0]) * torch.sin(math.pi * pts[:, 1])).reshape(-1, 1)
print(f"{mesh_np['triangles'].shape[0]} elements, {K.shape[0]} DoFs, nnz {K.nnz}")
print(f"assembly of K and f: {(t1 - t0) * 1e3:.2f} ms (Python call overhead included)")
print(f"CG solve           : {t2 - t1:.2f} s")
print(f"max nodal error    : {float((u - exact).abs().max()):.2e}")

# edge part of the residual estimator: eta^2 = sum_e |e| int_e [grad u_h . n]^2 ds
edges = InteriorEdgesBasis(mesh, ElementLine(polynomial_order=1, integration_order=2))
normals = mesh["interior_edges", "normals"].unsqueeze(-2).unsqueeze(-2)  # (N_e, 1, 1, 1, 2)
length = mesh["interior_edges", "length"].reshape(-1, 1)
edges._dx, edges.integration_points  # the edge basis's own quadrature data, once per mesh
torch.cuda.synchronize()
t3 = time.perf_counter()
_, grad_on_edges = basis.interpolate(edges, u)  # (N_e, 2, 1, 1, 2): one HIP launch
jump = ((grad_on_edges[:, [0]] - grad_on_edges[:, [1]]) * normals).sum(-1).reshape(-1, 1, 1, 1)
eta2 = (length * edges.integrate_functional(lambda b: jump**2 * torch.ones_like(b._dx))).sum()
torch.cuda.synchronize()
t4 = time.perf_counter()
print(f"jump estimator     : eta = {float(eta2.sqrt()):.4e} over {length.shape[0]} interior edges, {(t4 - t3) * 1e3:.2f} ms")
