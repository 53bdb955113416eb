"""Developer tool (GPU box): the K-only launch at S(2236) in the three cache states bench.py quotes
(back to back / behind 512 MB of unrelated reads / behind 512 MB of unrelated writes), with the
kernel's non-temporal value stores and with plain stores (ablation build, flag 1024), and with a
pause between the writes and the launch (is it the write-back of the dirty lines that collides?).

    python tools/time_k_states.py [n]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_fem_solver_amd as tf  # noqa: E402
from pytorch_fem_solver_amd import meshgen  # noqa: E402

torch.set_default_dtype(torch.float64)
torch.set_default_device("cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2236
mesh_np = meshgen.unit_square(n, 0.25, 0)
ne, nv = mesh_np["triangles"].shape[0], mesh_np["vertices"].shape[0]
scrub = torch.empty(64 * 1024 * 1024)  # 512 MB
for label, policy in (("non-temporal value stores", "nt"), ("plain value stores", "plain")):
    os.environ["TFEM_RINGS_STORES"] = policy
    basis = tf.Basis(tf.MeshTri(mesh_np), tf.ElementTri(1, 3))
    eng = basis._engine
    nnz = int(eng.csr_structure()[1].shape[0])
    vals = torch.empty(nnz)
    alg = 12 * ne + 16 * nv + 8 * nnz
    for _ in range(50):
        eng.bilinear(1.0, 0.0, out=vals)
    torch.cuda.synchronize()

    def one(state, pause_us=0):
        if state == "reads":
            scrub.sum()
        elif state == "writes":
            scrub.fill_(1.0)
        if pause_us:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            while (time.perf_counter() - t0) * 1e6 < pause_us:
                pass
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.bilinear(1.0, 0.0, out=vals)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e3

    print(label)
    # launches in a row, no host synchronisation in between (what bench.py calls back to back)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        eng.bilinear(1.0, 0.0, out=vals)
    b.record()
    torch.cuda.synchronize()
    t = a.elapsed_time(b) * 1e3 / 50
    print(f"    50 launches in a row                   : {t:7.1f} us = {alg / t / 8e6 * 100:5.1f} % of 8 TB/s", flush=True)
    for state, pause in (("back to back", 0), ("reads", 0), ("writes", 0), ("writes", 2000)):
        one(state, pause)
        t = float(np.median([one(state, pause) for _ in range(15)]))
        print(f"    single launch behind {state:13s} pause {pause:5d} us: {t:7.1f} us = {alg / t / 8e6 * 100:5.1f} % of 8 TB/s", flush=True)
    del basis, eng, vals
