"""Benchmark of the assembly hot path: Melements/s assembled (global K + f).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path over one mesh: P1 stiffness operator K (CSR values)
and load vector f, integration order 3, fp64, from vertex coordinates + connectivity
resident in HBM.  Workload at N = 1: mesh S(2236, 0.25, 0) = 9,999,392 elements (the
10 M-element mesh BASELINE.json's target is quoted on).  For N > 1 every rank holds one
such mesh strip of a [0,N]x[0,1] domain (weak scaling) and the shared-DoF rows are
exchanged with an RCCL all-reduce (pytorch_fem_solver_amd/parallel.py).

Prints ONE JSON line on rank 0.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    # defaults: 0.1 s of timed work behind 0.01 s of warm-up -- the first milliseconds after an
    # idle device run at lower clocks (tools/time_steps.py: 176 us per step in steady state)
    p.add_argument("--steps", type=int, default=500)
    p.add_argument("--warmup", type=int, default=50)
    p.add_argument("--probe-every", type=int, default=10,
                   help="HIP events around every n-th step's launch (an event pair costs 7-9 us "
                   "of stream time per step when recorded around every launch)")
    p.add_argument("--grid", dest="n", type=int, default=2236, help="grid cells per side (N_T = 2 n^2)")
    p.add_argument("--order", type=int, default=3, help="integration order")
    p.add_argument("--cpu-sample", type=int, default=2236, help="n of the CPU-baseline sample mesh")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--kernel", default="auto", help="auto | rings | tiles | atomic")
    p.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    p.add_argument("--no-other-configs", action="store_true",
                   help="skip the short measurement of BASELINE.json's config 3 (P2, 1e6 elements)")
    return p.parse_args()


def algorithmic_bytes(n_elems, n_verts, nnz, with_load=True):
    """conn 12 B/elem + each vertex once 16 B + each CSR value once 8 B for
    K (48 B/element); + 8 B per vertex for f (its inputs are already counted): 52 B/element
    for the fused K + f launch."""
    return 12 * n_elems + 16 * n_verts + 8 * nnz + (8 * n_verts if with_load else 0)


def measured_traffic(n, order, kernel, with_load):
    """HBM bytes per launch from the committed rocprofv3 PMC summary of this same command
    (profiles/r01_bench_pmc_summary.json, written by tools/summarize_pmc.py: 2 x FETCH_SIZE +
    WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md), or None when no summary matches
    the workload.  The instantiation is recognised by its name: the fifth template argument of
    `k_p1_rings<double, SLOTS, MASS, CHUNK, Q, ...>` is Q, 0 for the matrix-only launch."""
    path = os.path.join(REPO, "profiles", "r01_bench_pmc_summary.json")
    try:
        with open(path) as fh:
            summary = json.load(fh)
        w = summary["workload"]
        if (w["n"], w["order"]) != (n, order):
            return None
        for name, entry in summary["kernels"].items():
            if kernel + "<double" not in name:
                continue
            q = int(name.split("<", 1)[1].split(",")[4])  # k_p1_rings<T, SLOTS, MASS, CHUNK, Q, ...>
            if (q > 0) == with_load:
                return float(entry["hbm_traffic_bytes_per_launch"]["total"])
    except (OSError, KeyError, ValueError, IndexError):
        pass
    return None


def profiled_kernel_ms(kernel, with_load):
    """Average duration of the kernel in the committed rocprofv3 --kernel-trace --stats summary
    of this command (profiles/r01_bench_kernel_stats.csv), or None.  The HIP-event interval
    around ONE launch (roofline.kernel_ms) also holds the launch and completion latency of the
    dispatch (~10-15 us for this grid); the profiler's figure is the kernel alone."""
    import csv

    path = os.path.join(REPO, "profiles", "r01_bench_kernel_stats.csv")
    try:
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row["Name"]
                if kernel + "<double" not in name:
                    continue
                q = int(name.split("<", 1)[1].split(",")[4])
                if (q > 0) == with_load:
                    return float(row["AverageNs"]) * 1e-6
    except (OSError, KeyError, ValueError, IndexError):
        pass
    return None


def cpu_baseline(n, order):
    """The C/OpenMP oracle (oracle/assembly_oracle.c: a port of the reference's op sequence,
    one element per iteration) timed on this host's cores: local K + local f + scatter into
    CSR values / vector -- the same work as one GPU step."""
    import __graft_entry__ as ge

    ge.build_oracle()
    from oracle import assembly_oracle as orc
    from oracle import c_oracle
    from pytorch_fem_solver_amd import meshgen
    from pytorch_fem_solver_amd.basis.engine import symbolic_host

    mesh = meshgen.unit_square(n, 0.25, 0)
    verts, tris = mesh["vertices"], mesh["triangles"]
    nv = verts.shape[0]
    _, colind, slots = symbolic_host(tris, nv)  # symbolic phase, not timed (as on the GPU)
    fq = orc.source_sin_sin(c_oracle.points(verts, tris, order))[..., 0]
    best = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        k_local, f_local = c_oracle.p1_local(verts, tris, order, 1.0, 0.0, fq)
        vals = c_oracle.scatter_csr(k_local, slots, colind.shape[0])
        f = c_oracle.scatter_vector(f_local, tris, nv)
        best = min(best, time.perf_counter() - t0)
    del vals, f
    n_elems = tris.shape[0]
    return {
        "value": n_elems / best / 1e6,
        "unit": "Melements/s",
        "cores": c_oracle.threads(),
        "kind": "port",
        "sample": f"S({n},0.25,0) = {n_elems} elements, P1 K+f order {order}, C/OpenMP oracle "
        f"(oracle/assembly_oracle.c), best of 3",
    }


def p2_config3(device):
    """BASELINE.json config 3 (P2, 6x6 blocks, S(707) = 999,698 elements) on this GPU: the
    two launches of k_p2_rows (vertex rows, edge rows), timed with HIP events after the main
    measurement.  Algorithmic bytes per SURVEY.md 8(d): 24 B conn + 16 B per vertex + 8 B per
    CSR value."""
    import pytorch_fem_solver_amd as tf
    from pytorch_fem_solver_amd import meshgen

    mesh_np = meshgen.unit_square(707, 0.25, 0)
    basis = tf.Basis(tf.MeshTri(triangulation=mesh_np), tf.ElementTri(polynomial_order=2, integration_order=2))
    eng = basis._engine
    vals = eng.bilinear(1.0, 0.0)
    for _ in range(10):
        eng.bilinear(1.0, 0.0)
    torch.cuda.synchronize()
    batches = []  # median of five batches of ten launches (one host hiccup does not decide it)
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            eng.bilinear(1.0, 0.0)
        b.record()
        torch.cuda.synchronize()
        batches.append(a.elapsed_time(b) / 10)
    ms = float(np.median(batches))
    ne, nv, nnz = mesh_np["triangles"].shape[0], mesh_np["vertices"].shape[0], int(vals.shape[0])
    algo = 24 * ne + 16 * nv + 8 * nnz
    return {
        "workload": f"P2 stiffness K (CSR), order 2, mesh S(707,0.25,0) = {ne} elements, {eng.n_dofs} DoFs, nnz {nnz}",
        "kernel": eng.kernel_name(),
        "kernel_ms": ms,
        "value": ne / ms / 1e3,
        "unit": "Melements/s",
        "algorithmic_bytes_per_launch": algo,
        "achieved": algo / (ms * 1e-3) / 1e9,
        "frac": algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    local_rank = local_rank % max(torch.cuda.device_count(), 1)  # rehearsals share one card
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    distributed = world > 1
    if distributed:
        import torch.distributed as dist

        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    import pytorch_fem_solver_amd as tf
    from pytorch_fem_solver_amd import meshgen, parallel

    if args.kernel != "auto":
        os.environ["TFEM_KERNEL"] = args.kernel
    torch.set_default_dtype(torch.float64)
    n = args.n
    # rank r owns the strip [r, r+1] x [0, 1]; identical jitter pattern per strip so that
    # the shared boundary column coincides (boundary vertices are never displaced)
    mesh_np = meshgen.structured_rectangle(n, n, float(rank), float(rank + 1), 0.0, 1.0, jitter=0.25, seed=0)
    n_elems = mesh_np["triangles"].shape[0]
    n_verts = mesh_np["vertices"].shape[0]

    torch.set_default_device(device)
    mesh = tf.MeshTri(triangulation=mesh_np)
    basis = tf.Basis(mesh, tf.ElementTri(polynomial_order=1, integration_order=args.order))
    engine = basis._engine
    _, colind, _ = engine.csr_structure()
    nnz = int(colind.shape[0])
    nq = engine.n_quad

    # source values at the integration points: the user's f(x_q), evaluated by torch once
    # (tests/test_assembly.py:75-84); the per-step hot path consumes them from HBM
    import math

    pts = engine.geometry()[2]
    fq = (2.0 * math.pi**2 * torch.sin(math.pi * pts[..., 0]) * torch.sin(math.pi * pts[..., 1])).contiguous()
    del pts

    exchange = parallel.InterfaceExchange.for_strips(mesh_np, rank, world, engine) if distributed else None
    # the interface all-reduce of step i runs on a side stream and overlaps the assembly
    # launch of step i+1 (steps are independent; every step's exchange completes inside the
    # timed region, which ends with a device-wide synchronise)
    comm_stream = torch.cuda.Stream(device=device) if distributed else None

    # N > 1: the results of step i are still being exchanged while step i+1 assembles, so the
    # steps rotate over three preallocated (vals, f) pairs; before a pair is written again the
    # assembly stream waits (on the device, not the host) for the exchange that last used it
    depth = 3
    pairs = [(torch.empty(nnz), torch.empty(n_verts)) for _ in range(depth)] if distributed else None
    exchanged = [None] * depth
    counter = [0]

    def claim_pair():
        if exchange is not None and exchanged[counter[0] % depth] is not None:
            torch.cuda.current_stream().wait_event(exchanged[counter[0] % depth])

    def step():
        if exchange is None:
            return engine.assemble_system(1.0, 0.0, fq)  # one fused launch: K and f
        slot = counter[0] % depth
        counter[0] += 1
        vals, f = engine.assemble_system(1.0, 0.0, fq, out=pairs[slot])
        return vals, f, slot

    def exchange_step(vals, f, slot):
        exchanged[slot] = exchange.reduce_on(comm_stream, vals, f, record=False)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # an idle device runs its first ~20 ms of this launch slower (profiles/r01_launch_series.log:
    # 221 us per launch over the first 50, 186 over the next 50, 179 from then on), so the device
    # is brought to its steady state with 170 of the same steps (30 ms) before the W warm-up steps
    # the caller asked for; reported in config.device_warmup_steps
    # (a fixed count, the same on every rank: the steps of an N > 1 run hold collectives)
    device_warmup_steps = 170
    for k in range(device_warmup_steps):
        claim_pair()
        out = step()
        if exchange is not None:
            exchange_step(*out)
        if k % 10 == 9:
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        claim_pair()
        out = step()
        if exchange is not None:
            exchange_step(*out)
    barrier()

    # live launch duration: HIP events on the launch stream around every probe-th step
    probe = max(1, args.probe_every)
    starts = {i: torch.cuda.Event(enable_timing=True) for i in range(0, args.steps, probe)}
    ends = {i: torch.cuda.Event(enable_timing=True) for i in starts}
    t0 = time.perf_counter()
    for i in range(args.steps):
        claim_pair()
        if i in starts:
            starts[i].record()
        out = step()
        if i in ends:
            ends[i].record()
        if exchange is not None:
            exchange_step(*out)
    barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    k_ms = float(np.mean([starts[i].elapsed_time(ends[i]) for i in starts]))
    # the stiffness-only launch (the kernel BASELINE.json's 60 % target is quoted on), timed
    # after the measured region
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    engine.bilinear(1.0, 0.0)
    s0.record()
    for _ in range(30):
        engine.bilinear(1.0, 0.0)
    s1.record()
    torch.cuda.synchronize()
    k_only_ms = s0.elapsed_time(s1) / 30

    if rank == 0:
        total_elems = n_elems * world
        ms_per_step = elapsed * 1e3 / args.steps
        algo = algorithmic_bytes(n_elems, n_verts, nnz)
        achieved = algo / (k_ms * 1e-3) / 1e9
        line = {
            "metric": "Melements/s assembled (global K + f)",
            "value": total_elems / (elapsed / args.steps) / 1e6,
            "unit": "Melements/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"P1 Poisson stiffness K (CSR) + load f, order {args.order}, "
                f"mesh S({n},0.25,0) per GPU = {n_elems} elements, {n_verts} DoFs, nnz {nnz}",
                "elements_per_gpu": n_elems,
                "partition": "one unit-square strip per rank" if world > 1 else "single mesh",
                "kernel": engine.kernel_name(),
                "device_warmup_steps": device_warmup_steps,
                "probe_every": probe,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": engine.kernel_name(),
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic(n, args.order, engine.kernel_name(), True),
                "algorithmic_bytes_per_launch": algo,
                "kernel_ms": k_ms,
                "launch": "fused K + f (52 B/element algorithmic, SURVEY.md 8(d); the Q source values "
                "fq the launch must read, 8 Q B/element, are not part of that figure)",
                "algorithmic_bytes_incl_fq": algo + 8 * nq * n_elems,
                "frac_incl_fq": (algo + 8 * nq * n_elems) / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "stiffness_only": {
                    "kernel_ms": k_only_ms,
                    "traffic": measured_traffic(n, args.order, engine.kernel_name(), False),
                    "algorithmic_bytes_per_launch": algorithmic_bytes(n_elems, n_verts, nnz, False),
                    "achieved": algorithmic_bytes(n_elems, n_verts, nnz, False) / (k_only_ms * 1e-3) / 1e9,
                    "frac": algorithmic_bytes(n_elems, n_verts, nnz, False) / (k_only_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                },
            },
        }
        # what the HBM interface moved (PMC bytes of profiles/) over the live launch duration
        if n == 2236 and args.order == 3:  # the profiled workload
            line["roofline"]["kernel_ms_rocprofv3"] = profiled_kernel_ms(engine.kernel_name(), True)
            line["roofline"]["stiffness_only"]["kernel_ms_rocprofv3"] = profiled_kernel_ms(engine.kernel_name(), False)
        for obj, ms in ((line["roofline"], k_ms), (line["roofline"]["stiffness_only"], k_only_ms)):
            if obj["traffic"]:
                obj["traffic_GBps"] = obj["traffic"] / (ms * 1e-3) / 1e9
                obj["traffic_frac_of_peak"] = obj["traffic_GBps"] / HBM_PEAK_GBS
        if world == 1 and not args.no_other_configs:
            line["other_configs"] = {"C3_p2_stiffness_1e6": p2_config3(device)}
        if world == 1 and not args.no_cpu_baseline:  # rank 0 at N = 1 only
            torch.set_default_device("cpu")
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.order)
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
