"""Developer tool (GPU box): time the K-only kernels (rings / tiles) on the bench mesh.

    python tools/time_rings.py [--n 2236] [--kernels rings,tiles]
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
    basis = tf.Basis(tf.MeshTri(mesh_np), tf.ElementTri(1, 3))
    eng = basis._engine
    eng.kernel = kernel
    vals = eng.bilinear(1.0, 0.0)
    torch.cuda.synchronize()
    nnz = vals.shape[0]
    if kernel == "rings":
        z = eng.ring_plan()["layout"]
        print(f"ring plan: tiles {z[0]}, local verts/vertex {z[2] / nv:.3f}, max verts/tile {z[3]}, "
              f"slots {z[6]}, plan bytes/elem {z[12] / ne:.2f}")
    for _ in range(3):
        eng.bilinear(1.0, 0.0)
    times = []
    for _ in range(args.reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.bilinear(1.0, 0.0)
        b.record()
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b) * 1e3)
    t = float(np.median(times))
    alg = 12 * ne + 16 * nv + 8 * nnz
    print(f"{kernel:8s} {eng.kernel_name():22s} median {t:8.1f} us  min {min(times):8.1f} us  "
          f"{ne / t:9.0f} Melem/s  algorithmic {alg / t / 1e3:7.1f} GB/s = {alg / t / 8e6 * 100:5.1f} % of 8 TB/s")
    if ref is None:
        ref = vals
    else:
        err = (vals - ref).abs().max().item() / ref.abs().max().item()
        print(f"         max scaled difference to {args.kernels.split(',')[0]}: {err:.2e}")
