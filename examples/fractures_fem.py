"""Two perpendicular fractures glued along their trace (config 5): the pipeline of the
reference's examples/example_fractures_fem.py:58-64,235-297 on a synthetic mesh of the
same geometry ([-1,1]x[0,1], trace at x = 0), without the plotting.

    python examples/fractures_fem.py [m]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_fem import (  # noqa: E402
    ElementLine,
    ElementTri,
    FractureBasis,
    FracturesTri,
    InteriorEdgesFractureBasis,
)
from pytorch_fem_solver_amd import meshgen  # noqa: E402

torch.set_default_device("cuda" if torch.cuda.is_available() else "cpu")
torch.set_default_dtype(torch.float64)

m = int(sys.argv[1]) if len(sys.argv) > 1 else 16
tri = meshgen.fracture_rectangle(m)
fractures_data = torch.tensor(
    [
        [[-1.0, 0.0, 0.0], [1.0, 0.0, 0.0], [-1.0, 1.0, 0.0], [1.0, 1.0, 0.0]],
        [[0.0, 0.0, -1.0], [0.0, 0.0, 1.0], [0.0, 1.0, -1.0], [0.0, 1.0, 1.0]],
    ]
)
mesh = FracturesTri(triangulations=[tri, tri], fractures_3d_data=fractures_data)
V = FractureBasis(mesh, ElementTri(polynomial_order=1, integration_order=4))


def split(coordinates):
    x, y, z = torch.split(coordinates, 1, dim=-1)
    x1, _ = torch.split(x, 1, dim=0)
    y1, y2 = torch.split(y, 1, dim=0)
    _, z2 = torch.split(z, 1, dim=0)
    return x1, y1, y2, z2


def rhs(coordinates):
    x1, y1, y2, z2 = split(coordinates)
    r1 = 6.0 * (y1 - y1**2) * torch.abs(x1) - 2.0 * (torch.abs(x1) ** 3 - torch.abs(x1))
    r2 = -6.0 * (y2 - y2**2) * torch.abs(z2) + 2.0 * (torch.abs(z2) ** 3 - torch.abs(z2))
    return torch.cat([r1, r2], dim=0)


def exact(coordinates):
    x1, y1, y2, z2 = split(coordinates)
    e1 = -y1 * (1 - y1) * torch.abs(x1) * (x1**2 - 1)
    e2 = y2 * (1 - y2) * torch.abs(z2) * (z2**2 - 1)
    return torch.cat([e1, e2], dim=0)


A = V.integrate_bilinear_form(lambda basis: basis.v_grad @ basis.v_grad.mT)
b = V.integrate_linear_form(lambda basis: rhs(basis.integration_points) * basis.v)
u_h = V.solve(A, V.solution_tensor(), b)
I_u_h, I_u_h_grad = V.interpolate(V, u_h)
l2 = V.integrate_functional(lambda basis: (exact(basis.integration_points) - I_u_h) ** 2)
norm = V.integrate_functional(lambda basis: exact(basis.integration_points) ** 2)
print(f"{2 * tri['triangles'].shape[0]} elements, {A.shape[0]} global DoFs")
print("relative L2 error per fracture:", torch.sqrt(l2.sum(-2) / norm.sum(-2)).flatten().tolist())

V_edges = InteriorEdgesFractureBasis(mesh, ElementLine(polynomial_order=1, integration_order=2))
_, grad_on_edges = V.interpolate(V_edges, u_h)
n_E = mesh["interior_edges", "normals_3d"].unsqueeze(-2)
plus, minus = torch.unbind(grad_on_edges, dim=-4)
jump = (plus * n_E).sum(-1) + (minus * -n_E).sum(-1)
print("max |gradient jump| over interior edges:", float(jump.abs().max()))
