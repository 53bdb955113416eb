"""Developer tool (GPU box): time the K-only kernels (rings / tiles / atomic) on the bench
mesh, with optional ablation, and print the in-kernel stamps of the ring kernel.

    python tools/time_rings.py [--n 2236] [--kernels rings,tiles]
    kernel spec: name[:label[:workgroups per CU[:ablation flags[:zorder]]]], e.g.
    rings:x:3 (3 workgroups per CU), rings:x::1 (no value stores), rings:x:::zorder
    (Z-order vertex tiles).  Flags (TFEM_RINGS_DEBUG, results are wrong by design): 1 no value
    stores, 2 no row arithmetic, 4 no coordinate loads, 8 no staging and stores, 16 no record
    loads, 1024 plain instead of non-temporal stores.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_fem_solver_amd as tf  # noqa: E402
from pytorch_fem_solver_amd import meshgen  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--n", type=int, default=2236)
p.add_argument("--reps", type=int, default=30)
p.add_argument("--kernels", default="rings,tiles")
p.add_argument("--mesh", default="structured")
args = p.parse_args()

torch.set_default_dtype(torch.float64)
torch.set_default_device("cuda")
if args.mesh == "structured":
    mesh_np = meshgen.unit_square(args.n, 0.25, 0)
else:
    mesh_np = meshgen.delaunay_square(args.n * args.n, 0)
    if args.mesh == "delaunay_morton":
        mesh_np = meshgen.permute_mesh(mesh_np, vertex_order=meshgen.morton_order(mesh_np["vertices"]))
ne = mesh_np["triangles"].shape[0]
nv = mesh_np["vertices"].shape[0]
ref = None
for kernel in args.kernels.split(","):
    os.environ.pop("TFEM_RINGS", None)
    os.environ.pop("TFEM_RINGS_PER_CU", None)
    os.environ.pop("TFEM_RINGS_DEBUG", None)
    os.environ.pop("TFEM_RING_TILES", None)
    if ":" in kernel:  # rings:simple, rings:pipe:3 (variant, workgroups per CU)
        parts = kernel.split(":")
        kernel = parts[0]
        os.environ["TFEM_RINGS"] = parts[1]
        if len(parts) > 2 and parts[2]:
            os.environ["TFEM_RINGS_PER_CU"] = parts[2]
        if len(parts) > 3 and parts[3]:  # ablation flags (results are wrong by design)
            os.environ["TFEM_RINGS_DEBUG"] = parts[3]
        if len(parts) > 4:  # zorder: vertex tiles along the Z-order curve only
            os.environ["TFEM_RING_TILES"] = parts[4]
    basis = tf.Basis(tf.MeshTri(mesh_np), tf.ElementTri(1, 3))
    eng = basis._engine
    eng.kernel = kernel
    vals = eng.bilinear(1.0, 0.0)
    torch.cuda.synchronize()
    nnz = vals.shape[0]
    if kernel == "rings":
        z = eng.ring_plan()["layout"]
        print(f"ring plan: tiles {z[0]}, local verts/vertex {z[2] / nv:.3f}, max verts/tile {z[3]}, "
              f"slots {z[6]}, plan bytes/elem {z[12] / ne:.2f}, consecutive-vertex tiles {bool(z[13])}")
    for _ in range(3):
        eng.bilinear(1.0, 0.0)
    times = []
    batch = 10  # back-to-back launches between two events: the host's launch cost is hidden
    for _ in range(args.reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(batch):
            eng.bilinear(1.0, 0.0)
        b.record()
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b) * 1e3 / batch)
    t = float(np.median(times))
    alg = 12 * ne + 16 * nv + 8 * nnz
    print(f"{kernel:8s} {os.environ.get('TFEM_RINGS', ''):6s} {os.environ.get('TFEM_RINGS_PER_CU', ''):2s} dbg={os.environ.get('TFEM_RINGS_DEBUG', '0'):3s} {eng.kernel_name():18s} median {t:8.1f} us  min {min(times):8.1f} us  "
          f"{ne / t:9.0f} Melem/s  algorithmic {alg / t / 1e3:7.1f} GB/s = {alg / t / 8e6 * 100:5.1f} % of 8 TB/s")
    if ref is None:
        ref = vals
    else:
        err = (vals - ref).abs().max().item() / ref.abs().max().item()
        print(f"         max scaled difference to {args.kernels.split(',')[0]}: {err:.2e}")

# in-kernel stamps (ablation build, flag 256): where a wave's cycles go per tile
import ctypes  # noqa: E402
from pytorch_fem_solver_amd import _native  # noqa: E402

if "rings" in args.kernels:
    for key in ("TFEM_RINGS", "TFEM_RINGS_PER_CU", "TFEM_RINGS_DEBUG", "TFEM_RING_TILES"):
        os.environ.pop(key, None)
    basis = tf.Basis(tf.MeshTri(mesh_np), tf.ElementTri(1, 3))
    eng = basis._engine
    eng.kernel = "rings"
    rings = eng.ring_plan()
    d = eng._inputs()
    nnz = int(eng.csr_structure()[1].shape[0])
    vals = torch.empty(nnz)
    stamps = torch.zeros(12 * 4 * 4096, dtype=torch.int64)
    fn = _native.load().tfem_p1_rings_debug
    fn.restype = ctypes.c_int
    names = ["A loads", "B rows", "stage", "vmcnt0", "park", "stores", "barrier"]
    import math
    pts = eng.geometry()[2]
    fq = (2.0 * math.pi**2 * torch.sin(math.pi * pts[..., 0]) * torch.sin(math.pi * pts[..., 1])).contiguous()
    del pts
    fout = torch.empty(eng.n_dofs)
    # the fused K + f launch with parts switched off (ablation build; wrong results by design)
    for label, extra in (("K + f full", 0), ("no value / f stores", 1), ("no source-value loads", 32),
                         ("no element ids + source values", 96), ("no g staging", 128), ("no row arithmetic", 2),
                         ("no coordinate loads", 4), ("no record / code loads", 16), ("loads only (2+8+128)", 138),
                         ("loads only, 2 workgroups/CU", 138 | 2 << 16), ("loads only, 1 workgroup/CU", 138 | 1 << 16)):
        def run_fused():
            _native.check(fn(_native.ptr(d["coords"]), ctypes.c_int64(eng.n_dofs), 3,
                             _native.ptr(rings["blob"]), ctypes.c_void_p(rings["layout"].ctypes.data),
                             _native.ptr(vals), ctypes.c_int64(nnz), _native.current_stream(eng.device),
                             512 | (extra & 0xFFFF), extra >> 16, None, _native.ptr(fq), ctypes.c_int64(eng.n_elems),
                             _native.ptr(fout)))
        for _ in range(3):
            run_fused()
        torch.cuda.synchronize()
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ea.record()
        for _ in range(20):
            run_fused()
        eb.record()
        torch.cuda.synchronize()
        print(f"  fused, {label:34s} {ea.elapsed_time(eb) / 20 * 1e3:8.1f} us")
    print("cycles per tile per wave:   " + " ".join(f"{n:>8s}" for n in names) + "    total")
    for label, extra, load in (("full", 0, False), ("no stores", 1, False), ("no arithmetic", 2, False),
                               ("no gather", 4, False), ("no stores+arith", 3, False),
                               ("K + f full", 0, True), ("K + f no stores", 1, True), ("K + f no arith", 2, True)):
        for per_cu in (0, 2) if not load else (0, 1):
            stamps.zero_()
            _native.check(fn(_native.ptr(d["coords"]), ctypes.c_int64(eng.n_dofs), 3,
                             _native.ptr(rings["blob"]), ctypes.c_void_p(rings["layout"].ctypes.data),
                             _native.ptr(vals), ctypes.c_int64(nnz), _native.current_stream(eng.device),
                             256 | extra, per_cu, _native.ptr(stamps),
                             _native.ptr(fq) if load else None, ctypes.c_int64(eng.n_elems if load else 0),
                             _native.ptr(fout) if load else None))
            torch.cuda.synchronize()
            t = stamps.cpu().numpy().reshape(-1, 12)
            t = t[t[:, 7] > 0]
            per = t[:, :7].sum(0) / t[:, 7].sum()
            print(f"  {label:18s} wg/cu={per_cu or 'max'!s:3s} " + " ".join(f"{x:8.0f}" for x in per)
                  + f" {per.sum():8.0f}")
