"""Once-per-mesh set-up on the GPU box, phase by phase: mesh container,
edge topology, Basis (DoFs), CSR pattern, ring plan, device copies, first launch.  The phases
run twice, on two meshes of the same size: the first pass also pays what a process pays once
(torch loading its device kernels on first use, the library, the allocator's first blocks), the
second is what a caller who re-meshes pays per mesh.

    python tools/time_setup.py [n] [p]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import pytorch_fem_solver_amd as tf
    from pytorch_fem_solver_amd import meshgen
    from pytorch_fem_solver_amd.basis.engine import pattern_host, ring_plan_host

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2236
    p = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    torch.set_default_dtype(torch.float64)
    torch.set_default_device("cuda")
    torch.zeros(1).sum().item()
    for seed in (0, 1):
        print(f"---- pass {seed + 1} ({'first in the process' if seed == 0 else 'steady state: what re-meshing costs'})")
        one_pass(tf, meshgen, pattern_host, ring_plan_host, n, p, seed)


def one_pass(tf, meshgen, pattern_host, ring_plan_host, n, p, seed):
    t = time.perf_counter()
    mesh_np = meshgen.unit_square(n, 0.25, seed)
    print(f"[host] mesh generator S({n}): {mesh_np['triangles'].shape[0]} elements  {time.perf_counter() - t:7.3f} s "
          f"(not part of the set-up)")

    def lap(what, t0):
        torch.cuda.synchronize()
        print(f"{what:58s} {(time.perf_counter() - t0) * 1e3:9.1f} ms", flush=True)

    print(f"host threads: {os.environ.get('TFEM_HOST_THREADS', 'default')} of {os.cpu_count()} cores")
    t0 = time.perf_counter()
    mesh = tf.MeshTri(triangulation=mesh_np)
    lap("MeshTri (host arrays -> device tensors, X[conn])", t0)
    t0 = time.perf_counter()
    basis = tf.Basis(mesh, tf.ElementTri(p, 3 if p == 1 else 2))
    lap(f"Basis (P{p} DoFs)", t0)
    eng = basis._engine
    t0 = time.perf_counter()
    conn = eng._conn_host_np()
    lap("connectivity back on the host (int32)", t0)
    t0 = time.perf_counter()
    rowptr, colind = pattern_host(conn, eng.n_dofs)
    lap("CSR pattern (tfem_csr_pattern_*)", t0)
    if p == 1:
        coords = eng._coords_host_np()
        t0 = time.perf_counter()
        plan = ring_plan_host(conn, eng.n_dofs, coords, rowptr, colind)
        lap("ring plan (tfem_ring_plan_*)", t0)
        t0 = time.perf_counter()
        blob = torch.from_numpy(plan["blob"]).to("cuda")
        lap(f"ring plan to the device ({plan['blob'].nbytes / 1e6:.0f} MB, pageable)", t0)
        del blob
    t0 = time.perf_counter()
    eng.bilinear(1.0, 0.0)
    lap("engine: pattern + plan + copies + first K launch (pattern and plan built again)", t0)
    t0 = time.perf_counter()
    eng.bilinear(1.0, 0.0)
    lap("second K launch", t0)
    t0 = time.perf_counter()
    _ = mesh["interior_edges", "cells"]
    lap("edge topology (interior / boundary edges, normals, lengths; torch on the device)", t0)


if __name__ == "__main__":
    main()
