"""Developer tool (GPU box): the fused K + f launch (sin*sin evaluated in the launch) and the K-only
launch over mesh sizes S(n) -- what a rank of a strong split (BASELINE config 4) holds at 2, 4, 8 GPUs.

    python tools/time_fused_sizes.py [n ...]
"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pytorch_fem_solver_amd as tf  # noqa: E402
from pytorch_fem_solver_amd import meshgen  # noqa: E402
from pytorch_fem_solver_amd.basis import forms  # noqa: E402

torch.set_default_dtype(torch.float64)
torch.set_default_device("cuda")


def load(b):
    x, y = torch.split(b.integration_points, 1, dim=-1)
    return 2.0 * math.pi**2 * torch.sin(math.pi * x) * torch.sin(math.pi * y) * b.v


def timed(fn, reps=300, warm=200):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        best.append(a.elapsed_time(b) / reps * 1e3)
    return sorted(best)[len(best) // 2]


sizes = [int(v) for v in sys.argv[1:]] or [500, 790, 1118, 1581, 2236]
for n in sizes:
    basis = tf.Basis(tf.MeshTri(meshgen.unit_square(n, 0.25, 0)), tf.ElementTri(1, 3))
    eng = basis._engine
    program = forms.trace(load, basis, (), {}).coefficient.program()
    vals, f = eng.assemble_system(1.0, 0.0, source=program)
    out = (vals, f)
    step = eng.prepared_system(1.0, 0.0, out, source=program)
    t_sys = timed(step)
    kout = torch.empty_like(vals)
    t_k = timed(lambda: eng.bilinear(1.0, 0.0, out=kout))
    plan = eng.ring_plan()
    ne = eng.n_elems
    print(f"S({n}) {ne:9d} elements, {int(plan['layout'][0]):6d} tiles, {int(plan['layout'][28]):5d} runs:  K+f {t_sys:7.1f} us "
          f"({ne / t_sys / 1e3:6.1f} G elements/s)   K {t_k:7.1f} us ({ne / t_k / 1e3:6.1f} G elements/s)", flush=True)
    del basis, eng, vals, f, out, step, kout
    torch.cuda.empty_cache()
