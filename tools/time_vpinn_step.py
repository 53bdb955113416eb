"""One training step of the reference's weak-form VPINN (examples/example_weak.py:64-75,132-152
restated): residual r = integrate_linear_form(f v - grad v . grad u_theta), loss = sum(r_inner^2),
backward, for a small network.  Through the fused residual kernels (this package) and as the
reference evaluates it (torch expressions on the cached tensors + index_put_), both on the GPU.

    python tools/time_vpinn_step.py [n ...]      (mesh S(n): 2 n^2 elements)
"""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pytorch_fem_solver_amd as tf  # noqa: E402
from pytorch_fem_solver_amd import meshgen  # noqa: E402

torch.set_default_dtype(torch.float64)
torch.set_default_device("cuda")


def rhs(x, y):
    return 2.0 * math.pi**2 * torch.sin(math.pi * x) * torch.sin(math.pi * y)


def run(n, steps=50):
    basis = tf.Basis(tf.MeshTri(meshgen.unit_square(n, 0.25, 0)), tf.ElementTri(1, 3))
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(2, 25), torch.nn.Tanh(), torch.nn.Linear(25, 25), torch.nn.Tanh(),
                              torch.nn.Linear(25, 1, bias=False))
    params = list(net.parameters())
    inner = basis._basis_parameters["inner_dofs"]

    def gradient(points):  # model/neural_network.py:85-100
        points.requires_grad_(True)
        out = net(points)
        return torch.autograd.grad([out], [points], [torch.ones_like(out)], create_graph=True)[0]

    def residual(b, grad):
        x, y = torch.split(b.integration_points, 1, dim=-1)
        return rhs(x, y) * b.v - (b.v_grad @ grad(b.integration_points).mT)

    def fused_step():
        r = basis.integrate_linear_form(residual, gradient)
        loss = (r[inner] ** 2).sum()
        return torch.autograd.grad(loss, params)

    conn = basis._global_dofs4elements.reshape(-1).long()
    shape = basis._basis_parameters["linear_form_shape"]

    def torch_step():  # abstract_basis.py:95-112 as written
        integrand = (residual(basis, gradient) * basis._dx).sum(-3)
        r = torch.zeros(shape).index_put((conn,), integrand.reshape(-1, 1), accumulate=True)
        loss = (r[inner] ** 2).sum()
        return torch.autograd.grad(loss, params)

    out = {}
    for name, fn in (("fused kernels", fused_step), ("torch expressions", torch_step)):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / steps * 1e3
    g1, g2 = fused_step(), torch_step()
    err = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-300)) for a, b in zip(g1, g2))
    print(f"S({n}) = {2 * n * n:9d} elements   fused {out['fused kernels']:8.3f} ms/step   "
          f"torch {out['torch expressions']:8.3f} ms/step   gradients agree to {err:.1e}", flush=True)


for n in [int(a) for a in sys.argv[1:]] or [16, 71, 224, 707]:
    run(n)
