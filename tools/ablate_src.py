"""Developer tool (GPU box): the fused K + f launch of bench.py's step under every library in
tools/variants/ (tools/build_variants.py), one subprocess per library.

    python tools/ablate_src.py [n]
"""
import glob
import math
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))


def child(n):
    import torch

    import pytorch_fem_solver_amd as tf
    from pytorch_fem_solver_amd import meshgen
    from pytorch_fem_solver_amd.basis import forms

    torch.set_default_dtype(torch.float64)
    torch.set_default_device("cuda")
    basis = tf.Basis(tf.MeshTri(meshgen.unit_square(n, 0.25, 0)), tf.ElementTri(1, 3))
    eng = basis._engine

    def load(b):
        x, y = torch.split(b.integration_points, 1, dim=-1)
        return 2.0 * math.pi**2 * torch.sin(math.pi * x) * torch.sin(math.pi * y) * b.v

    program = forms.trace(load, basis, (), {}).coefficient.program()
    vals, f = eng.assemble_system(1.0, 0.0, source=program)
    out = (vals, f)

    def timed(fn, reps=200, warm=150):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        best = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                fn()
            b.record()
            torch.cuda.synchronize()
            best.append(a.elapsed_time(b) / reps * 1e3)
        return sorted(best)[len(best) // 2]

    t_sys = timed(lambda: eng.assemble_system(1.0, 0.0, source=program, out=out))
    t_f = timed(lambda: eng.load_source(program))
    one = forms.compile_program(("c", 1.0))
    t_one = timed(lambda: eng.assemble_system(1.0, 0.0, source=one, out=out))
    print(f"{os.environ.get('TFEM_VARIANT', 'product'):20s} per_cu={os.environ.get('TFEM_RINGS_PER_CU', '-'):2s} K+f {t_sys:7.1f} us   "
          f"f only {t_f:7.1f} us   K+f(const) {t_one:7.1f} us", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(int(sys.argv[2]))
    else:
        n = sys.argv[1] if len(sys.argv) > 1 else "2236"
        only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
        libs = [("product", None)] + [(os.path.basename(p)[8:-3], p) for p in sorted(glob.glob(os.path.join(HERE, "variants", "libtfem_*.so")))]
        per_cus = sys.argv[3].split(",") if len(sys.argv) > 3 else [""]
        for name, path in [(nm, pth) for nm, pth in libs for _ in per_cus]:
            if only and name not in only:
                continue
            env = dict(os.environ, TFEM_VARIANT=name)
            pc = per_cus[0]
            per_cus = per_cus[1:] + per_cus[:1]
            if pc:
                env["TFEM_RINGS_PER_CU"] = pc
            if path:
                env["TFEM_HIP_LIB"] = path
            r = subprocess.run([sys.executable, __file__, "--child", n], env=env, capture_output=True, text=True, timeout=600)
            sys.stdout.write(r.stdout[-400:] if r.returncode == 0 else f"{name}: FAILED\n{r.stderr[-1500:]}\n")
            sys.stdout.write("".join(ln + "\n" for ln in r.stderr.splitlines() if ln.startswith("[stamps]")))
            sys.stdout.flush()
