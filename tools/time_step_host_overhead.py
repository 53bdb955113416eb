"""Host time of the multi-GPU step of bench.py without the collective itself, on ONE GPU with a
small mesh (the launches are short: what remains is Python + ctypes + stream bookkeeping per step).
The all-reduce adds its own host time (RCCL enqueue) on top."""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pytorch_fem_solver_amd as tf  # noqa: E402
from pytorch_fem_solver_amd import meshgen, parallel  # noqa: E402
from pytorch_fem_solver_amd.basis import forms  # noqa: E402

torch.set_default_dtype(torch.float64)
torch.set_default_device("cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
mesh_np = meshgen.structured_rectangle(n, n, 0.0, 1.0, 1.0, 2.0, jitter=0.25, seed=0)
basis = tf.Basis(tf.MeshTri(triangulation=mesh_np), tf.ElementTri(1, 3))
eng = basis._engine
ex = parallel.InterfaceExchange.for_strips(mesh_np, 1, 3, eng)
eng.set_priority_vertices(ex.shared_vertices(mesh_np["vertices"].shape[0]))


def load(b):
    x, y = torch.split(b.integration_points, 1, dim=-1)
    return 2.0 * math.pi**2 * torch.sin(math.pi * x) * torch.sin(math.pi * y) * b.v


program = forms.trace(load, basis, (), {}).coefficient.program()
nnz = int(eng.csr_structure()[1].shape[0])
out = (torch.empty(nnz), torch.empty(eng.n_dofs))
comm = torch.cuda.Stream(priority=-1)


def step():
    with torch.cuda.stream(comm):
        eng.assemble_system(1.0, 0.0, source=program, out=out, tiles="priority")
        ex.pack(*out)
        ex.unpack(*out)
        done = torch.cuda.Event()
        done.record(comm)
    eng.assemble_system(1.0, 0.0, source=program, out=out, tiles="rest")
    return done


launch_priority = eng.prepared_system(1.0, 0.0, out, source=program, tiles="priority")
launch_rest = eng.prepared_system(1.0, 0.0, out, source=program, tiles="rest")
pack, unpack = ex.prepared(*out)


def prepared_step():
    with torch.cuda.stream(comm):
        launch_priority()
        pack()
        unpack()
        done = torch.cuda.Event()
        done.record(comm)
    launch_rest()
    return done


def timed(fn):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        torch.cuda.current_stream().wait_event(fn())
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 2000 * 1e6


print(f"prepared launches: {timed(prepared_step):.1f} us per step")
for _ in range(200):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000):
    torch.cuda.current_stream().wait_event(step())
torch.cuda.synchronize()
print(f"S({n}) strip, {eng.n_elems} elements: {(time.perf_counter() - t0) / 2000 * 1e6:.1f} us per step "
      f"(two range launches, pack, unpack, events; no collective)")
