"""P2 stiffness on a Morton-renumbered Delaunay mesh (for rocprofv3 --kernel-trace --stats)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pytorch_fem_solver_amd as tf  # noqa: E402
from pytorch_fem_solver_amd import meshgen  # noqa: E402

torch.set_default_dtype(torch.float64)
torch.set_default_device("cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
native = meshgen.delaunay_square(n, 1)
mesh = meshgen.permute_mesh(native, vertex_order=meshgen.morton_order(native["vertices"]))
basis = tf.Basis(tf.MeshTri(mesh), tf.ElementTri(2, 2))
eng = basis._engine
for _ in range(30):
    eng.bilinear(1.0, 0.0)
torch.cuda.synchronize()
z = eng.p2_plan()["layout"]
print("elements", eng.n_elems, "vertex tiles", int(z[0]), "edge tiles", int(z[1]), "long rows", int(z[18]),
      "vertex-tile local verts", int(z[7]), "edge-tile local verts", int(z[8]))
