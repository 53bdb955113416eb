"""Developer tool (GPU box): the PCIe-inclusive rate -- a CPU-resident mesh (the reference's
tests run on CPU tensors): inputs are staged to the GPU once per Basis, every result comes back
over PCIe."""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_fem_solver_amd as tf  # noqa: E402
from pytorch_fem_solver_amd import meshgen  # noqa: E402

torch.set_default_dtype(torch.float64)
torch.set_default_device("cpu")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2236
mesh_np = meshgen.unit_square(n, 0.25, 0)
basis = tf.Basis(tf.MeshTri(mesh_np), tf.ElementTri(1, 3))


def stiffness(b):
    return b.v_grad @ b.v_grad.mT


K = basis.integrate_bilinear_form(stiffness, layout="csr")
assert not K.values.is_cuda
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    K = basis.integrate_bilinear_form(stiffness, layout="csr")
    best = min(best, time.perf_counter() - t0)
ne = mesh_np["triangles"].shape[0]
print(f"CPU-resident mesh, {ne} elements: K (CSR values back on the host) {best * 1e3:.2f} ms per call = "
      f"{ne / best / 1e6:.0f} Melements/s, {K.values.numel() * 8 / best / 1e9:.1f} GB/s of results over PCIe")
