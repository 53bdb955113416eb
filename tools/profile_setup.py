"""cProfile of the once-per-mesh set-up at S(n): where the host time of MeshTri / Basis / the first
assembly / the edge topology goes.    python tools/profile_setup.py [n]"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pytorch_fem_solver_amd as tf  # noqa: E402
from pytorch_fem_solver_amd import meshgen  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2236
torch.set_default_dtype(torch.float64)
torch.set_default_device("cuda")
torch.zeros(1).sum().item()
mesh_np = meshgen.unit_square(n, 0.25, 0)
state = {}


def stage(name, fn):
    prof = cProfile.Profile()
    prof.enable()
    fn()
    torch.cuda.synchronize()
    prof.disable()
    print(f"\n===== {name}", flush=True)
    pstats.Stats(prof).sort_stats("cumulative").print_stats(18)


stage("MeshTri", lambda: state.update(mesh=tf.MeshTri(triangulation=mesh_np)))
stage("Basis", lambda: state.update(basis=tf.Basis(state["mesh"], tf.ElementTri(1, 3))))
stage("first assembly", lambda: state["basis"]._engine.bilinear(1.0, 0.0))
stage("edge topology", lambda: state["mesh"]["interior_edges", "cells"])
