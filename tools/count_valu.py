"""A few launches of every variant of the P1 launch, for `rocprofv3 --pmc SQ_INSTS_VALU ...`:
which variant issues how many vector instructions (tools/count_valu.sh summarises)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pytorch_fem_solver_amd as tf  # noqa: E402
from pytorch_fem_solver_amd import meshgen  # noqa: E402
from pytorch_fem_solver_amd.basis import forms  # noqa: E402

torch.set_default_dtype(torch.float64)
torch.set_default_device("cuda")
basis = tf.Basis(tf.MeshTri(meshgen.unit_square(2236, 0.25, 0)), tf.ElementTri(1, 3))
eng = basis._engine


def load(b):
    x, y = torch.split(b.integration_points, 1, dim=-1)
    return 2.0 * math.pi**2 * torch.sin(math.pi * x) * torch.sin(math.pi * y) * b.v


program = forms.trace(load, basis, (), {}).coefficient.program()
one = forms.compile_program(("c", 1.0))
poly = forms.compile_program(("add", ("mul", ("x",), ("y",)), ("c", 1.0)))
pts = eng.geometry()[2]
fq = (torch.sin(pts[..., 0]) * pts[..., 1]).contiguous()
for _ in range(5):
    eng.bilinear(1.0, 0.0)
for _ in range(6):
    eng.assemble_system(1.0, 0.0, fq)
for _ in range(7):
    eng.assemble_system(1.0, 0.0, source=one)
for _ in range(8):
    eng.assemble_system(1.0, 0.0, source=poly)
for _ in range(9):
    eng.assemble_system(1.0, 0.0, source=program)
torch.cuda.synchronize()
