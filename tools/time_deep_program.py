import math, os, sys, torch
sys.path.insert(0, "/root/repo")
import pytorch_fem_solver_amd as tf
from pytorch_fem_solver_amd import meshgen
from pytorch_fem_solver_amd.basis import forms
torch.set_default_dtype(torch.float64); torch.set_default_device("cuda")
basis = tf.Basis(tf.MeshTri(meshgen.unit_square(2236, 0.25, 0)), tf.ElementTri(1, 3))
eng = basis._engine
deep = forms.compile_program(("mul", ("mul", ("sin", ("x",)), ("sin", ("y",))), ("mul", ("add", ("x",), ("c", 1.0)), ("add", ("y",), ("c", 2.0)))))
def timed(fn, reps=100, warm=100):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
out = eng.assemble_system(1.0, 0.0, source=deep)
print(f"stack-3 program, TFEM_SRC_WIDE={os.environ.get('TFEM_SRC_WIDE','-')}: {timed(lambda: eng.assemble_system(1.0, 0.0, source=deep, out=out)):.1f} us")
