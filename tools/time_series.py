"""Developer tool (GPU box): how the duration of the fused launch evolves over a run of
back-to-back steps that starts on an idle device (HIP events around every 5th launch)."""
import math
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
mesh_np = meshgen.unit_square(2236, 0.25, 0)
basis = tf.Basis(tf.MeshTri(mesh_np), tf.ElementTri(1, 3))
eng = basis._engine
pts = eng.geometry()[2]
fq = (2.0 * math.pi**2 * torch.sin(math.pi * pts[..., 0]) * torch.sin(math.pi * pts[..., 1])).contiguous()
del pts
eng.assemble_system(1.0, 0.0, fq)
torch.cuda.synchronize()
for idle in (0.5, 0.0):
    time.sleep(idle)
    steps = 600
    ev = {i: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for i in range(0, steps, 5)}
    t0 = time.perf_counter()
    marks = {}
    for i in range(steps):
        if i in ev:
            ev[i][0].record()
        eng.assemble_system(1.0, 0.0, fq)
        if i in ev:
            ev[i][1].record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e6
    t = np.array([ev[i][0].elapsed_time(ev[i][1]) * 1e3 for i in sorted(ev)])
    since = np.array([ev[0][0].elapsed_time(ev[i][0]) for i in sorted(ev)])
    print(f"after {idle:.1f} s idle: {wall:.1f} us per step over {steps} steps; launch duration by step:")
    for a in range(0, len(t), 10):
        print(f"   steps {5 * a:3d}-{5 * (a + 10) - 1:3d} (t = {since[a]:6.1f} ms): mean {t[a:a + 10].mean():6.1f} us, min {t[a:a + 10].min():6.1f}")
