"""Developer tool (GPU box): time the P2 stiffness kernel (config 3: 6x6 blocks, ~1e6 elements)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_fem_solver_amd as tf  # noqa: E402
from pytorch_fem_solver_amd import meshgen  # noqa: E402

torch.set_default_dtype(torch.float64)
torch.set_default_device("cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 707
mesh_np = meshgen.unit_square(n, 0.25, 0)
for kernel in ("auto", "gather", "atomic"):
    basis = tf.Basis(tf.MeshTri(mesh_np), tf.ElementTri(2, 2))
    eng = basis._engine
    eng.kernel = kernel
    vals = eng.bilinear(1.0, 0.0)
    torch.cuda.synchronize()
    ne, nnz, ndof = mesh_np["triangles"].shape[0], vals.shape[0], eng.n_dofs
    times = []
    for _ in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            eng.bilinear(1.0, 0.0)
        b.record()
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b) / 5 * 1e3)
    t = float(np.median(times))
    alg = 24 * ne + 16 * mesh_np["vertices"].shape[0] + 8 * nnz
    how = {"auto": "row kernels (k_p2_rows: vertex rows, edge rows)",
           "gather": "element blocks + gather (two launches)", "atomic": "atomic scatter"}[kernel]
    ref = vals if kernel == "auto" else ref
    if kernel != "auto":
        print("   max scaled difference to the row kernels: %.2e" % ((vals - ref).abs().max().item() / ref.abs().max().item()))
    print(f"P2 stiffness order 2: {ne} elements, {ndof} DoFs, nnz {nnz}: {how}: median {t:.1f} us "
          f"{ne / t:.0f} Melem/s, algorithmic {alg / ne:.0f} B/elem -> {alg / t / 1e3:.0f} GB/s = {alg / t / 8e6 * 100:.1f} % of 8 TB/s")
