"""Developer tool (GPU box): per-step time of the bench loop (fused K + f launch, 1e7 elements)
with HIP events around every launch, without them, and as a HIP graph of all steps."""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_fem_solver_amd as tf  # noqa: E402
from pytorch_fem_solver_amd import meshgen  # noqa: E402

torch.set_default_dtype(torch.float64)
torch.set_default_device("cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2236
steps = 50
mesh_np = meshgen.unit_square(n, 0.25, 0)
basis = tf.Basis(tf.MeshTri(mesh_np), tf.ElementTri(1, 3))
eng = basis._engine
pts = eng.geometry()[2]
fq = (2.0 * math.pi**2 * torch.sin(math.pi * pts[..., 0]) * torch.sin(math.pi * pts[..., 1])).contiguous()
del pts
nnz = int(eng.csr_structure()[1].shape[0])
out = (torch.empty(nnz), torch.empty(eng.n_dofs))
for _ in range(5):
    eng.assemble_system(1.0, 0.0, fq, out=out)
torch.cuda.synchronize()


def wall(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


def with_events():
    for _ in range(steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.assemble_system(1.0, 0.0, fq)
        b.record()


def fresh_outputs():
    for _ in range(steps):
        eng.assemble_system(1.0, 0.0, fq)


def fixed_outputs():
    for _ in range(steps):
        eng.assemble_system(1.0, 0.0, fq, out=out)


def k_only():
    for _ in range(steps):
        eng.bilinear(1.0, 0.0)


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    eng.assemble_system(1.0, 0.0, fq, out=out)
torch.cuda.current_stream().wait_stream(side)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    for _ in range(steps):
        eng.assemble_system(1.0, 0.0, fq, out=out)

for name, fn in (("events around every launch", with_events), ("no events, fresh outputs", fresh_outputs),
                 ("no events, fixed outputs", fixed_outputs), ("HIP graph of all steps", graph.replay),
                 ("K only, no events", k_only)) * 2:
    print(f"{name:32s} {min(wall(fn) for _ in range(3)):7.1f} us per step")
