"""Delaunay mesh D(N_v, seed), Morton-renumbered and native: K alone and the fused K + f step,
with long rows (4-dword records + k_p1_long_rows) and with 8-dword records (TFEM_RING_LONG=0).

    python tools/time_delaunay.py [n_points]
"""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tools.time_source import timed  # noqa: E402


def main():
    import pytorch_fem_solver_amd as tf
    from pytorch_fem_solver_amd import meshgen
    from pytorch_fem_solver_amd.basis import forms

    n_points = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    torch.set_default_dtype(torch.float64)
    torch.set_default_device("cuda")
    t = time.time()
    native = meshgen.delaunay_square(n_points, 1)
    print(f"Delaunay mesh of {n_points} points: {native['triangles'].shape[0]} elements ({time.time() - t:.1f} s on the host)", flush=True)
    meshes = {"morton": meshgen.permute_mesh(native, vertex_order=meshgen.morton_order(native["vertices"])), "native": native}

    def load(b):
        x, y = torch.split(b.integration_points, 1, dim=-1)
        return 2.0 * math.pi**2 * torch.sin(math.pi * x) * torch.sin(math.pi * y) * b.v

    for name, mesh_np in meshes.items():
        for long_rows in ("1", "0"):
            os.environ["TFEM_RING_LONG"] = long_rows
            basis = tf.Basis(tf.MeshTri(mesh_np), tf.ElementTri(1, 3))
            eng = basis._engine
            program = forms.trace(load, basis, (), {}).coefficient.program()
            vals = eng.bilinear(1.0, 0.0)
            ne, nv, nnz = eng.n_elems, eng.n_dofs, int(vals.shape[0])
            algo_k = 12 * ne + 16 * nv + 8 * nnz
            k_us = timed(lambda: eng.bilinear(1.0, 0.0), 100, 60)
            kf_us = timed(lambda: eng.assemble_system(1.0, 0.0, source=program), 100, 60)
            plan = eng.ring_plan() if eng.kernel_name() == "k_p1_rings" else None
            desc = "-" if plan is None else (
                f"{'chunked' if plan['chunked'] else 'z-order'} tiles, {int(plan['layout'][6])}-slot records, "
                f"{int(plan['layout'][23])} long rows, plan {int(plan['layout'][12]) / ne:.1f} B/element")
            print(f"{name:7s} TFEM_RING_LONG={long_rows}  {eng.kernel_name():16s} K {k_us:7.1f} us ({algo_k / k_us / 8e6 * 100:5.1f} %)   "
                  f"K+f(sin*sin in the launch) {kf_us:7.1f} us ({(algo_k + 8 * nv) / kf_us / 8e6 * 100:5.1f} %)   {desc}", flush=True)
            del basis, eng, vals
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
