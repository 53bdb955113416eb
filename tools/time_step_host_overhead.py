"""Host time per step of a sharded run (bench.py --gpus N: parallel.ShardedSteps), on ONE GPU with
ONE rank over RCCL and a small mesh -- the launches are short, so what the wall clock shows per
step is the host: Python + ctypes + stream bookkeeping + the collective's enqueue for the eager
steps, a fraction of one graph launch for the steps recorded into HIP graphs.

    python tools/time_step_host_overhead.py [n] [steps per graph] [pairs] [tiny]
"""
import math
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pytorch_fem_solver_amd as tf  # noqa: E402
from pytorch_fem_solver_amd import meshgen, parallel  # noqa: E402
from pytorch_fem_solver_amd.basis import forms  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
torch.set_default_dtype(torch.float64)
torch.set_default_device("cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
graph_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 3  # rotating (vals, f) pairs
mesh_np = meshgen.structured_rectangle(n, n, 0.0, 1.0, 1.0, 2.0, jitter=0.25, seed=0)
basis = tf.Basis(tf.MeshTri(triangulation=mesh_np), tf.ElementTri(1, 3))
eng = basis._engine
nv = mesh_np["vertices"].shape[0]
nnz = int(eng.csr_structure()[1].shape[0])
# what a strip between two other strips shares: its first and last grid row (entries of the row's
# vertices and of the horizontal edges), carried through the all-reduce of this one rank
rows = np.concatenate([np.arange(n + 1), nv - 1 - np.arange(n + 1)])
rowptr = eng.csr_structure()[0].cpu().numpy()
k_idx = np.concatenate([np.arange(rowptr[r], rowptr[r + 1]) for r in rows])
if len(sys.argv) > 4 and sys.argv[4] == "tiny":  # an exchange of one entry: what the launch chain alone costs in a graph
    k_idx, rows = k_idx[:1], rows[:1]
ex = parallel.InterfaceExchange(k_idx, np.arange(k_idx.size), rows, np.arange(rows.size), k_idx.size, rows.size,
                                eng.device, torch.float64)


def load(b):
    x, y = torch.split(b.integration_points, 1, dim=-1)
    return 2.0 * math.pi**2 * torch.sin(math.pi * x) * torch.sin(math.pi * y) * b.v


program = forms.trace(load, basis, (), {}).coefficient.program()


def timed(steps, label, count=3000):
    steps.run(300)
    steps.sync()
    t0 = time.perf_counter()
    steps.run(count)
    host = time.perf_counter() - t0
    steps.sync()
    wall = time.perf_counter() - t0
    print(f"{label:58s} host {host / count * 1e6:6.1f} us per step   wall {wall / count * 1e6:6.1f} us per step", flush=True)


print(f"S({n}) strip, {eng.n_elems} elements, interface buffer {ex.nbytes} B, one rank over RCCL")
for first in (False, True):
    if first:
        basis = tf.Basis(tf.MeshTri(triangulation=mesh_np), tf.ElementTri(1, 3))
        eng = basis._engine
        eng.set_priority_vertices(ex.shared_vertices(nv))
    steps = parallel.ShardedSteps(eng, ex, 1.0, 0.0, source=program, depth=depth, interface_first=first)
    what = "interface tiles first (two launches)" if first else "one launch per step"
    timed(steps, f"eager, {what}")
    ok = steps.capture(graph_steps)
    if not ok:
        print("capture failed:", steps.capture_error)
        continue
    timed(steps, f"HIP graphs of {graph_steps} steps, {what}")
dist.destroy_process_group()
