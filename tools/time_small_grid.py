"""K-only and fused launch at ~1e6 elements for different numbers of resident workgroups per CU
(TFEM_RINGS_PER_CU): how long a persistent workgroup's tile list must be to pay for its ramp."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tools.time_source import timed  # noqa: E402
import pytorch_fem_solver_amd as tf  # noqa: E402
from pytorch_fem_solver_amd import meshgen  # noqa: E402
from pytorch_fem_solver_amd.basis import forms  # noqa: E402

torch.set_default_dtype(torch.float64)
torch.set_default_device("cuda")
for n in (707, 1000, 1414):
    basis = tf.Basis(tf.MeshTri(meshgen.unit_square(n, 0.25, 0)), tf.ElementTri(1, 3))
    eng = basis._engine

    def load(b):
        x, y = torch.split(b.integration_points, 1, dim=-1)
        return 2.0 * math.pi**2 * torch.sin(math.pi * x) * torch.sin(math.pi * y) * b.v

    program = forms.trace(load, basis, (), {}).coefficient.program()
    eng.bilinear(1.0, 0.0)
    n_tiles = int(eng.ring_plan()["layout"][0])
    out = []
    for per in ("1", "2", "3", "4", "5", "8"):
        os.environ["TFEM_RINGS_PER_CU"] = per
        out.append("%s: K %.1f  K+f %.1f" % (per, timed(lambda: eng.bilinear(1.0, 0.0), 200, 100),
                                            timed(lambda: eng.assemble_system(1.0, 0.0, source=program), 100, 50)))
    print(f"S({n}) {eng.n_elems} elements, {n_tiles} tiles | per CU " + " | ".join(out), flush=True)
