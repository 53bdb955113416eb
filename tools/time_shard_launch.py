"""Developer tool (GPU box): the fused K + f launch on ONE shard of a strong split of S(n) into `world`
Morton ranges (what rank `rank` of bench.py --gpus `world` launches per step), with the launch
configuration of a sharded step (TFEM_RINGS_RESERVE_CUS=1), next to S(m) of the same size.

    python tools/time_shard_launch.py [world ...]
"""
import math
import os
import sys

os.environ.setdefault("TFEM_RINGS_RESERVE_CUS", "1")
import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pytorch_fem_solver_amd as tf  # noqa: E402
from pytorch_fem_solver_amd import meshgen, parallel  # noqa: E402
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


def launch_us(mesh_np):
    basis = tf.Basis(tf.MeshTri(mesh_np), tf.ElementTri(1, 3))
    eng = basis._engine
    program = forms.trace(load, basis, (), {}).coefficient.program()
    out = eng.assemble_system(1.0, 0.0, source=program)
    step = eng.prepared_system(1.0, 0.0, out, source=program)
    plan = eng.ring_plan()
    z = plan["layout"]
    return timed(step), eng.n_elems, int(z[0]), int(z[28]), bool(plan["chunked"]), int(z[2]) / max(1, int(z[1]))


n = 2236
global_mesh = meshgen.unit_square(n, 0.25, 0)
total = global_mesh["triangles"].shape[0]
for world in [int(v) for v in sys.argv[1:]] or [2, 4, 8]:
    order, bounds = parallel.partition_elements(global_mesh["vertices"], global_mesh["triangles"], world, "morton")
    worst = 0.0
    for rank in sorted({0, world // 2, world - 1}):
        shard, _ = parallel.extract_shard(global_mesh, order[bounds[rank]:bounds[rank + 1]])
        us, ne, tiles, runs, chunked, local = launch_us(shard)
        worst = max(worst, us)
        print(f"world {world} rank {rank}: {ne:8d} elements, {tiles:5d} tiles, {runs:4d} runs, "
              f"{'consecutive-vertex' if chunked else 'z-order'} tiles, {local:.2f} local vertices per vertex:  "
              f"K+f {us:6.1f} us", flush=True)
    m = int(round(math.sqrt(total / world / 2)))
    us, ne, tiles, runs, chunked, local = launch_us(meshgen.unit_square(m, 0.25, 0))
    print(f"    S({m}) {ne:8d} elements, {tiles:5d} tiles: K+f {us:6.1f} us   (slowest of the shards timed: {worst:.1f} us)", flush=True)
