"""Fused K + f launch at S(n): source values from memory against the source program in the
launch, and the load-only launches; HIP events, steady state.

    python tools/time_source.py [n] [order]
"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def rhs(x, y):
    return 2.0 * math.pi**2 * torch.sin(math.pi * x) * torch.sin(math.pi * y)


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


def main():
    import pytorch_fem_solver_amd as tf
    from pytorch_fem_solver_amd import meshgen
    from pytorch_fem_solver_amd.basis import forms

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2236
    order = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    torch.set_default_dtype(torch.float64)
    torch.set_default_device("cuda")
    mesh_np = meshgen.unit_square(n, 0.25, 0)
    basis = tf.Basis(tf.MeshTri(mesh_np), tf.ElementTri(1, order))
    eng = basis._engine

    def load(b):
        x, y = torch.split(b.integration_points, 1, dim=-1)
        return rhs(x, y) * b.v

    expr = forms.trace(load, basis, (), {})
    program = expr.coefficient.program()
    pts = eng.geometry()[2]
    fq = rhs(pts[..., 0], pts[..., 1]).contiguous()
    del pts
    ne = eng.n_elems
    nv = eng.n_dofs
    nnz = int(eng.csr_structure()[1].shape[0])
    algo = 12 * ne + 16 * nv + 8 * nnz + 8 * nv
    poly = forms.compile_program(("add", ("mul", ("x",), ("y",)), ("c", 1.0)))
    chain = ("x",)
    for i in range(16):
        chain = ("add", chain, ("c", 1.0 + i))
    chain = forms.compile_program(chain)
    one = forms.compile_program(("c", 1.0))
    # three values at a time: (x y) ((x + 1) (y + 2)) is evaluated as two products of two operands
    # each, the second while the first waits on the stack -> the one-element-per-pass interpreter
    deep = forms.compile_program(("mul", ("mul", ("sin", ("x",)), ("sin", ("y",))),
                                  ("mul", ("add", ("x",), ("c", 1.0)), ("add", ("y",), ("c", 2.0)))))
    rows = [
        ("K only", lambda: eng.bilinear(1.0, 0.0), algo - 8 * nv),
        ("K + f, fq from memory", lambda: eng.assemble_system(1.0, 0.0, fq), algo),
        ("K + f, program sin*sin", lambda: eng.assemble_system(1.0, 0.0, source=program), algo),
        ("K + f, program x*y+1", lambda: eng.assemble_system(1.0, 0.0, source=poly), algo),
        ("K + f, program constant (1 op)", lambda: eng.assemble_system(1.0, 0.0, source=one), algo),
        ("K + f, program x + 16 constants (17 ops)", lambda: eng.assemble_system(1.0, 0.0, source=chain), algo),
        ("K + f, program sin x sin y (x+1)(y+2) (stack of 3: one element per pass)",
         lambda: eng.assemble_system(1.0, 0.0, source=deep), algo),
        ("f only, fq from memory", lambda: eng.load(fq), None),
        ("f only, program sin*sin", lambda: eng.load_source(program), None),
        ("tfem_source_eval sin*sin", lambda: eng.source_values(program), None),
        ("tfem_source_eval constant (1 op)", lambda: eng.source_values(one), None),
        ("tfem_source_eval 17 ops", lambda: eng.source_values(chain), None),
        ("torch: f(x_q) from cached points", lambda: rhs(*torch.split(basis.integration_points, 1, dim=-1)), None),
        ("API: integrate_bilinear_form(csr) + integrate_linear_form",
         lambda: (basis.integrate_bilinear_form(lambda b: b.v_grad @ b.v_grad.mT, layout="csr"), basis.integrate_linear_form(load)), algo),
    ]
    print(f"S({n}) = {ne} elements, order {order}, xcd={os.environ.get('TFEM_RINGS_XCD', 'default')}, "
          f"per_cu={os.environ.get('TFEM_RINGS_PER_CU', 'default')}")
    for name, fn, nbytes in rows:
        us = timed(fn)
        extra = f"  {nbytes / us / 1e6:7.2f} TB/s algorithmic = {nbytes / us / 1e6 / 8 * 100:5.1f} %" if nbytes else ""
        print(f"{name:62s} {us:8.1f} us{extra}", flush=True)


if __name__ == "__main__":
    main()
