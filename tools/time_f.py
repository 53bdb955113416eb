import math, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_fem_solver_amd as tf
from pytorch_fem_solver_amd import meshgen
torch.set_default_dtype(torch.float64); torch.set_default_device("cuda")
mesh_np = meshgen.unit_square(2236, 0.25, 0)
basis = tf.Basis(tf.MeshTri(mesh_np), tf.ElementTri(1, 3))
eng = basis._engine
pts = eng.geometry()[2]
fq = (2.0 * math.pi**2 * torch.sin(math.pi * pts[..., 0]) * torch.sin(math.pi * pts[..., 1])).contiguous()
del pts
def t(fn, reps=100):
    for _ in range(150): fn()  # steady state: the first ~20 ms after an idle phase run slower
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
print("K rings           %.1f us" % t(lambda: eng.bilinear(1.0, 0.0)))
print("f rings only      %.1f us" % t(lambda: eng._assemble_rings(0.0, 0.0, fq, want_matrix=False)))
print("f tiles only      %.1f us" % t(lambda: eng._assemble_tiles(0.0, 0.0, False, fq)[1]))
fr, ft = eng._assemble_rings(0.0, 0.0, fq, want_matrix=False), eng._assemble_tiles(0.0, 0.0, False, fq)[1]
print("f rings vs tiles  %.2e" % ((fr - ft).abs().max().item() / ft.abs().max().item()))
print("K+f fused tiles   %.1f us" % t(lambda: eng._assemble_tiles(1.0, 0.0, True, fq)))
print("K+f fused rings   %.1f us" % t(lambda: eng._assemble_rings(1.0, 0.0, fq)))
for per_cu in (2, 3, 4):
    os.environ["TFEM_RINGS_PER_CU"] = str(per_cu)
    print("K+f fused rings, %d workgroups/CU  %.1f us" % (per_cu, t(lambda: eng._assemble_rings(1.0, 0.0, fq))))
os.environ.pop("TFEM_RINGS_PER_CU")
v1, f1 = eng._assemble_rings(1.0, 0.0, fq)
v2, f2 = eng._assemble_tiles(1.0, 0.0, True, fq)
print("rings vs tiles: K %.2e  f %.2e" % ((v1 - v2).abs().max().item() / v2.abs().max().item(),
                                          (f1 - f2).abs().max().item() / f2.abs().max().item()))
