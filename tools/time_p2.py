"""P2 stiffness (tfem_p2_assemble_rows) on S(707) = BASELINE config 3 and on a Morton-numbered
Delaunay mesh: ms per call by HIP events (run under rocprofv3 --kernel-trace --stats for the split
into vertex rows / edge rows / long rows)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pytorch_fem_solver_amd as tf  # noqa: E402
from pytorch_fem_solver_amd import meshgen  # noqa: E402

torch.set_default_dtype(torch.float64)
torch.set_default_device("cuda")


def measure(name, mesh_np):
    basis = tf.Basis(tf.MeshTri(mesh_np), tf.ElementTri(2, 2))
    eng = basis._engine
    vals = eng.bilinear(1.0, 0.0)
    for _ in range(200):
        eng.bilinear(1.0, 0.0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200):
        eng.bilinear(1.0, 0.0)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 200
    ne, nv, nnz = mesh_np["triangles"].shape[0], mesh_np["vertices"].shape[0], int(vals.shape[0])
    algo = 24 * ne + 16 * nv + 8 * nnz
    z = eng.p2_plan()["layout"]
    print(f"{name:28s} {ne:9d} elements  {ms * 1e3:7.1f} us  {algo / ms / 1e9:6.2f} TB/s algorithmic = "
          f"{algo / ms / 1e9 / 8 * 100:5.1f} %   tiles {int(z[0])} + {int(z[1])}, long rows {int(z[18])}", flush=True)


measure("S(707) P2 (config 3)", meshgen.unit_square(707, 0.25, 0))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
native = meshgen.delaunay_square(n, 1)
measure("Delaunay, Morton-numbered P2", meshgen.permute_mesh(native, vertex_order=meshgen.morton_order(native["vertices"])))
