"""Per-call cost of the public API on small meshes (C1 and below): integrate_bilinear_form
(dense, as the reference returns it, and layout="csr") and integrate_linear_form, wall clock per
call with the device kept busy (no synchronise between calls) and with a synchronise per call."""
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


def stiffness(b):
    return b.v_grad @ b.v_grad.mT


def load(b):
    x, y = torch.split(b.integration_points, 1, dim=-1)
    return 2.0 * math.pi**2 * torch.sin(math.pi * x) * torch.sin(math.pi * y) * b.v


def per_call(fn, n=300, sync_each=False):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
        if sync_each:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for n in (16, 71, 224):
    basis = tf.Basis(tf.MeshTri(meshgen.unit_square(n, 0.25, 0)), tf.ElementTri(1, 3))
    rows = [("bilinear csr", lambda: basis.integrate_bilinear_form(stiffness, layout="csr")),
            ("linear", lambda: basis.integrate_linear_form(load)),
            ("engine.assemble_system", None)]
    from pytorch_fem_solver_amd.basis import forms
    program = forms.trace(load, basis, (), {}).coefficient.program()
    rows[2] = ("engine.assemble_system", lambda: basis._engine.assemble_system(1.0, 0.0, source=program))
    if n <= 71:
        rows.insert(1, ("bilinear dense", lambda: basis.integrate_bilinear_form(stiffness)))
    print(f"S({n}) = {2 * n * n} elements: " + "   ".join(
        f"{name} {per_call(fn):.0f} us ({per_call(fn, sync_each=True):.0f} with sync)" for name, fn in rows), flush=True)
