"""Developer tool (GPU box): the K-only launch with the memory-side cache (256 MB MALL) warm
(back-to-back launches re-read the same mesh and plan) and cold (1 GiB written in between)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_fem_solver_amd as tf  # noqa: E402
from pytorch_fem_solver_amd import meshgen  # noqa: E402

torch.set_default_dtype(torch.float64)
torch.set_default_device("cuda")
for n in [int(v) for v in (sys.argv[1:] or ["2236", "3162"])]:
    mesh_np = meshgen.unit_square(n, 0.25, 0)
    basis = tf.Basis(tf.MeshTri(mesh_np), tf.ElementTri(1, 3))
    eng = basis._engine
    ne, nv = mesh_np["triangles"].shape[0], mesh_np["vertices"].shape[0]
    nnz = int(eng.csr_structure()[1].shape[0])
    out = (torch.empty(nnz), None)
    eng.bilinear(1.0, 0.0)
    flush = torch.empty(1 << 27)  # 1 GiB of doubles

    def one(cold):
        if cold:
            flush.fill_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.bilinear(1.0, 0.0)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e3

    alg = 12 * ne + 16 * nv + 8 * nnz
    for cold in (False, True, False, True):
        one(cold)
        t = float(np.median([one(cold) for _ in range(20)]))
        print(f"S({n}) {ne} elements, K only, single launches, cache {'cold' if cold else 'warm'}: "
              f"{t:7.1f} us = {alg / t / 8e6 * 100:5.1f} % of 8 TB/s (algorithmic {alg / 1e6:.0f} MB)")
    del flush, basis, eng
    torch.cuda.empty_cache()
