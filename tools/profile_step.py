"""Developer tool (GPU box): run the bench workload's kernels a few times so that
rocprofv3 can attribute time / counters to them.

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -- python3 tools/profile_step.py
"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_fem_solver_amd as tf  # noqa: E402
from pytorch_fem_solver_amd import meshgen  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--n", type=int, default=2236)
p.add_argument("--reps", type=int, default=5)
p.add_argument("--no-load", action="store_true")
args = p.parse_args()

torch.set_default_dtype(torch.float64)
torch.set_default_device("cuda")
mesh_np = meshgen.unit_square(args.n, 0.25, 0)
basis = tf.Basis(tf.MeshTri(mesh_np), tf.ElementTri(1, 3))
eng = basis._engine
pts = eng.geometry()[2]
fq = (2.0 * math.pi**2 * torch.sin(math.pi * pts[..., 0]) * torch.sin(math.pi * pts[..., 1])).contiguous()
del pts
for _ in range(args.reps):
    vals = eng.bilinear(1.0, 0.0)
    if not args.no_load:
        f = eng.load(fq)
torch.cuda.synchronize()
print("done", eng.kernel_name(), float(vals.sum()))
