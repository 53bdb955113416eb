"""torch.profiler breakdown of tools/time_vpinn_step.py's fused step at S(n)."""
import math
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pytorch_fem_solver_amd as tf  # noqa: E402
from pytorch_fem_solver_amd import meshgen  # noqa: E402

torch.set_default_dtype(torch.float64)
torch.set_default_device("cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 224
basis = tf.Basis(tf.MeshTri(meshgen.unit_square(n, 0.25, 0)), tf.ElementTri(1, 3))
net = torch.nn.Sequential(torch.nn.Linear(2, 25), torch.nn.Tanh(), torch.nn.Linear(25, 25), torch.nn.Tanh(),
                          torch.nn.Linear(25, 1, bias=False))
params = list(net.parameters())
inner = basis._basis_parameters["inner_dofs"]


def gradient(points):
    points.requires_grad_(True)
    out = net(points)
    return torch.autograd.grad([out], [points], [torch.ones_like(out)], create_graph=True)[0]


def residual(b, grad):
    x, y = torch.split(b.integration_points, 1, dim=-1)
    return 2.0 * math.pi**2 * torch.sin(math.pi * x) * torch.sin(math.pi * y) * b.v - (b.v_grad @ grad(b.integration_points).mT)


def step():
    r = basis.integrate_linear_form(residual, gradient)
    return torch.autograd.grad((r[inner] ** 2).sum(), params)


for _ in range(5):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=60))
