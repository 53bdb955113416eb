"""Developer tool (GPU box): time the tile kernel with phases switched off.

    python tools/ablate_tiles.py [--n 2236]

Uses the diagnostic build tfem_p1_bilinear_tiles_debug; results of ablated runs are wrong
by design.  Reports ms per launch for: full kernel, no LDS atomics, no element phase, no
value stores, no coordinate gather, and combinations.
"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_fem_solver_amd as tf  # noqa: E402
from pytorch_fem_solver_amd import _native, meshgen  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--n", type=int, default=2236)
p.add_argument("--reps", type=int, default=20)
p.add_argument("--variants", default="1")
p.add_argument("--load", action="store_true", help="fused K + load vector")
args = p.parse_args()

torch.set_default_dtype(torch.float64)
torch.set_default_device("cuda")
mesh_np = meshgen.unit_square(args.n, 0.25, 0)
basis = tf.Basis(tf.MeshTri(mesh_np), tf.ElementTri(1, 3))
eng = basis._engine
tiles = eng.tile_plan()
sz = tiles["sizes"]
d = eng._inputs()
nnz = int(eng.csr_structure()[1].shape[0])
vals = torch.empty(nnz)
stamps = torch.zeros(8 * 4 * 4096, dtype=torch.int64)
lib = _native.load()
fn = lib.tfem_p1_tiles_debug
fn.restype = ctypes.c_int
ne = mesh_np["triangles"].shape[0]
print(f"elements {ne}, tiles {sz[0]}, records/elem {sz[1]/ne:.3f}, plan sizes {sz}")


p_load = args.load
fq = None
fout = None
if p_load:
    import math
    pts = eng.geometry()[2]
    fq = (2.0 * math.pi**2 * torch.sin(math.pi * pts[..., 0]) * torch.sin(math.pi * pts[..., 1])).contiguous()
    fout = torch.empty(eng.n_dofs)
    del pts


layout = tiles["layout"]


def run(flags, variant):
    _native.check(fn(_native.ptr(d["coords"]), ctypes.c_int64(eng.n_dofs), 3,
                     _native.ptr(tiles["blob"]), ctypes.c_void_p(layout.ctypes.data),
                     _native.ptr(vals), ctypes.c_int64(nnz), _native.ptr(fq),
                     ctypes.c_int64(eng.n_elems), _native.ptr(fout),
                     _native.current_stream(eng.device), flags, _native.ptr(stamps)))


CASES = (("full", 0), ("full + stagger", 32), ("stores wrapped into 1 MB (L2 hits)", 64), ("no atomics", 1), ("no element phase", 2), ("no stores", 4),
         ("no gather", 8), ("no atomics+no stores", 5), ("no elem+no stores", 6),
         ("no elem+no gather", 10), ("only records+zero+barriers", 14),
         ("no gather+no stores", 12))

for variant in [int(v) for v in args.variants.split(",")]:
    print("k_p1_tiles_pipe (persistent, pipelined)", "K + f" if p_load else "K only")
    for name, flags in CASES:
        for _ in range(3):
            run(flags, variant)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(args.reps):
            run(flags, variant)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / args.reps
        print(f"  {name:32s} flags={flags:2d}  {ms*1e3:8.1f} us   {ne/ms/1e3:9.0f} Melem/s")

# in-kernel stamps (flag 16): where a wave's cycles go, per half-iteration
names = ["S2 elem", "bar(S2)", "vmcnt0", "S3+S4", "S5 store", "bar(S5)"]
print("cycles per tile per wave:  " + "  ".join(f"{n:>9s}" for n in names) + "      total")
for label, extra in (("full", 0), ("stores->1MB", 64), ("no atomics", 1), ("no stores", 4), ("no gather", 8),
                     ("no atomics+stores", 5), ("no elem", 2)):
    stamps.zero_()
    run(16 | extra, 1)
    torch.cuda.synchronize()
    t = stamps.cpu().numpy().reshape(-1, 8)
    t = t[t[:, 6] > 0]
    per = t[:, :6].sum(0) / t[:, 6].sum()
    print(f"  {label:24s} " + "  ".join(f"{x:9.0f}" for x in per) + f"  {per.sum():9.0f}"
          + f"   [in store stmts: {t[:, 7].sum() / t[:, 6].sum():7.0f}]")
