"""Benchmark of the assembly hot path: Melements/s assembled (global K + f).

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path over one mesh, from vertex coordinates + connectivity
resident in HBM to the P1 stiffness operator K (CSR values) AND the load vector f of the
reference's own source f(x, y) = 2 pi^2 sin(pi x) sin(pi y) (tests/test_assembly.py:75-84),
integration order 3, fp64.  The source is not pre-evaluated: the tracer records the caller's
expression once (pytorch_fem_solver_amd/basis/forms.py) and every step's launch evaluates it
at its integration points (k_p1_rings, SRC instantiation) -- the per-call work of the
reference's integrate_linear_form is inside the timed step.

Workload at N = 1: mesh S(2236, 0.25, 0) = 9,999,392 elements (the 10 M-element mesh
BASELINE.json's target is quoted on).  N > 1, --scaling weak (default): every rank holds one
such strip of a [0,N]x[0,1] domain; --scaling strong: ONE S(2236) mesh cut into N element
ranges along the Morton curve of the centroids (BASELINE config 4).  Either way the DoFs
shared between ranks are summed with one RCCL all-reduce of a packed interface buffer
(pytorch_fem_solver_amd/parallel.py) on a side stream.

Prints ONE JSON line on rank 0.
"""

from __future__ import annotations

import argparse
import hashlib
import json
import math
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
PROFILE_TAG = "r03"


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    # defaults: 0.13 s of timed work behind 0.01 s of warm-up -- the first milliseconds after an
    # idle device run at lower clocks
    p.add_argument("--steps", type=int, default=500)
    p.add_argument("--warmup", type=int, default=50)
    p.add_argument("--probe-every", type=int, default=10,
                   help="HIP events around every n-th step's launch (an event pair costs 7-9 us "
                   "of stream time per step when recorded around every launch)")
    p.add_argument("--grid", dest="n", type=int, default=2236, help="grid cells per side (N_T = 2 n^2)")
    p.add_argument("--order", type=int, default=3, help="integration order")
    p.add_argument("--scaling", choices=("weak", "strong"), default="strong",
                   help="N > 1. strong (default) = BASELINE config 4 as stated: ONE S(2236) mesh cut into N "
                   "element ranges; weak: one S(2236) strip per rank")
    p.add_argument("--cpu-sample", type=int, default=2236, help="n of the CPU-baseline sample mesh")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--kernel", default="auto", help="auto | rings | tiles | atomic")
    p.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    p.add_argument("--interface-first", action="store_true",
                   help="N > 1: two launches per step, the tiles owning shared rows first (on the exchange "
                   "stream, in front of their exchange); default: one launch per step, its exchange beside "
                   "the next step's launch")
    p.add_argument("--step-mode", choices=("auto", "graph", "eager"), default="auto",
                   help="N > 1: auto = the steps recorded into HIP graphs over RCCL, eager over gloo")
    p.add_argument("--graph-steps", type=int, default=24,
                   help="steps per HIP graph (a multiple of 3; capped by --steps).  A replay ends with an exchange nothing "
                        "overlaps and a gap to the next replay, ~24 us in all: 41.8 us per step in graphs of 6, 38.3 in "
                        "graphs of 24 at 1.25e6 elements per rank (profiles/r03_step_host_overhead.log)")
    p.add_argument("--no-other-configs", action="store_true",
                   help="skip the short measurements of the other configurations (P2; Delaunay meshes)")
    p.add_argument("--delaunay-points", type=int, default=1_000_000,
                   help="vertices of the Delaunay meshes of other_configs (5000000 = the ~1e7-element "
                   "size of profiles/; generating it takes over a minute of host time)")
    return p.parse_args()


def algorithmic_bytes(n_elems, n_verts, nnz, with_load=True):
    """conn 12 B/elem + each vertex once 16 B + each CSR value once 8 B for
    K (48 B/element); + 8 B per vertex for f (its inputs are already counted): 52 B/element
    for the fused K + f launch."""
    return 12 * n_elems + 16 * n_verts + 8 * nnz + (8 * n_verts if with_load else 0)


def rhs(x, y):  # tests/test_assembly.py:75-77
    return 2.0 * math.pi**2 * torch.sin(math.pi * x) * torch.sin(math.pi * y)


def load_form(basis):  # tests/test_assembly.py:79-84
    x, y = torch.split(basis.integration_points, 1, dim=-1)
    return rhs(x, y) * basis.v


def stiffness_form(basis):  # examples/example_fractures_fem.py:112-116
    return basis.v_grad @ basis.v_grad.mT


def source_sha():
    """Digest of the kernel sources and of the flags they are compiled with: profiles taken from
    another build are not quoted."""
    import __graft_entry__ as build

    h = hashlib.sha256()
    h.update(repr((build.HIPCC_FLAGS, sorted(build.PER_FILE_FLAGS.items()))).encode())
    csrc = os.path.join(REPO, "pytorch_fem_solver_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".hpp", ".cpp")):
            with open(os.path.join(csrc, name), "rb") as fh:
                h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def template_args(kernel_name):
    inside = kernel_name.split("<", 1)[1].split(">", 1)[0]  # the kernel's own (flat) argument list
    return [t.strip() for t in inside.split(",")]


def is_launch(kernel_name, kernel, with_load):
    """`k_p1_rings<T, SLOTS, MASS, CHUNK, Q, DBG[, KMAT[, SRC]]>`: the fused launch of the step is
    the fp64 instantiation with Q > 0, KMAT = true and SRC > 0, the matrix-only one has Q = 0."""
    if kernel + "<double" not in kernel_name:
        return False
    args = template_args(kernel_name)
    q = int(args[4])
    if not with_load:
        return q == 0
    return q > 0 and len(args) >= 8 and args[7] in ("true", "1", "2") and args[6] == "true"


def committed_profile(n, order, kernel):
    """HBM traffic per launch (2 x FETCH_SIZE + WRITE_SIZE, the gfx950 rule of
    MI355X_MICROARCH.md) and steady-state kernel durations from the committed rocprofv3 runs of
    this command (profiles/r02_*; tools/profile_bench.sh) -- only when they were taken from the
    kernel sources of this tree (digest stored with the summary) and on this workload."""
    out = {"fused": {}, "k_only": {}}
    try:
        with open(os.path.join(REPO, "profiles", f"{PROFILE_TAG}_bench_pmc_summary.json")) as fh:
            summary = json.load(fh)
        w = summary["workload"]
        if (w["n"], w["order"]) != (n, order) or summary.get("source_sha") != source_sha():
            return out
        for name, entry in summary["kernels"].items():
            for key, with_load in (("fused", True), ("k_only", False)):
                if is_launch(name, kernel, with_load):
                    out[key]["traffic"] = float(entry["hbm_traffic_bytes_per_launch"]["total"])
                    steady = entry.get("kernel_us_steady")
                    if steady:
                        out[key]["kernel_ms_rocprofv3"] = steady["mean"] * 1e-3
                    util = entry.get("valu_utilisation")
                    if util is not None:
                        out[key]["valu_utilisation"] = util
    except (OSError, KeyError, ValueError, IndexError):
        pass
    return out


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_cpu_share():
    """Hardware threads this process may actually use: the affinity mask, cut down to the cgroup's
    CPU quota when there is one (on a shared box the mask shows every core of the host)."""
    count = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:  # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = fh.read().split()[:2]
        if quota != "max":
            count = min(count, max(1, int(math.ceil(int(quota) / int(period)))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                quota, period = int(fq.read()), int(fp.read())
            if quota > 0:
                count = min(count, max(1, int(math.ceil(quota / period))))
        except (OSError, ValueError):
            pass
    return count


def cpu_baseline(n, order):
    """SURVEY.md 8(d) / BASELINE.md section 4: the reference's own torch op sequence (rows a-1 ...
    a-9 restated in oracle/torch_restatement.py, pinned to the reference's outputs by
    tests/test_oracle_golden.py) on this host's cores with torch.set_num_threads(all of them), in
    three stages -- (1) geometry cache = Basis.__init__, (2) local stage (a(V) * dx).sum(-3) and the
    local load vector INCLUDING f(x_q) (torch, every call, as the reference evaluates it),
    (3) global scatter: index_put_(accumulate=True) into the CSR value array through the slot map
    (the reference's dense target is 2 TB at this size: "not a reference capability") and into the
    vector.  `value` = elements / (stage 2 + stage 3) = one K + f step on a cached Basis, the work of
    one GPU step; best of 3 after one warm-up pass.  Second entry: the C/OpenMP port of the same
    path (oracle/assembly_oracle.c), f(x_q) by torch on all threads as well."""
    import __graft_entry__ as ge

    ge.build_oracle()
    from oracle import c_oracle
    from oracle import torch_restatement as tr
    from pytorch_fem_solver_amd import meshgen
    from pytorch_fem_solver_amd.basis.engine import symbolic_host

    # every hardware thread this process may use (os.cpu_count() of the host, cut to the container's
    # CPU quota: 256 threads on a 16-core share ran the same passes 3-4 x slower)
    threads = host_cpu_share()
    torch.set_num_threads(threads)
    c_oracle.set_threads(threads)
    mesh = meshgen.unit_square(n, 0.25, 0)
    verts_np, tris_np = mesh["vertices"], mesh["triangles"]
    nv, n_elems = verts_np.shape[0], tris_np.shape[0]
    _, colind, slots_np = symbolic_host(tris_np, nv)  # symbolic phase, not timed (as on the GPU)
    nnz = int(colind.shape[0])
    verts, tris, slots = torch.from_numpy(verts_np), torch.from_numpy(tris_np), torch.from_numpy(slots_np)
    stage = {"geometry": [], "local_K": [], "local_f": [], "scatter_K": [], "scatter_f": []}
    for rep in range(4):
        t0 = time.perf_counter()
        geo = tr.geometry_cache(verts, tris, order)
        t1 = time.perf_counter()
        k_local = tr.local_bilinear(geo)
        t2 = time.perf_counter()
        f_local = tr.local_linear(geo)
        t3 = time.perf_counter()
        vals = tr.scatter_bilinear_csr(k_local, slots, nnz)
        t4 = time.perf_counter()
        f = tr.scatter_linear(f_local, tris, nv)
        t5 = time.perf_counter()
        if rep:  # pass 0 is the warm-up
            for key, dt in zip(stage, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                stage[key].append(dt)
        del k_local, f_local, vals, f
    best = {k: min(v) for k, v in stage.items()}
    step = best["local_K"] + best["local_f"] + best["scatter_K"] + best["scatter_f"]
    # the C/OpenMP port: f(x_q) by torch (all threads) on the cached points + one pass per element
    pts = geo["integration_points"]
    c_best = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        x, y = torch.split(pts, 1, dim=-1)
        fq = tr.rhs(x, y).reshape(n_elems, -1).numpy()
        k_l, f_l = c_oracle.p1_local(verts_np, tris_np, order, 1.0, 0.0, fq)
        c_oracle.scatter_csr(k_l, slots_np, nnz)
        c_oracle.scatter_vector(f_l, tris_np, nv)
        c_best = min(c_best, time.perf_counter() - t0)
    return {
        "value": n_elems / step / 1e6,
        "unit": "Melements/s",
        "cores": threads,
        "host_cpu_count": os.cpu_count(),
        "cpu": cpu_model(),
        "kind": "port",
        "sample": f"S({n},0.25,0) = {n_elems} elements, P1 K + f order {order}, fp64: the reference's torch op "
        f"sequence (oracle/torch_restatement.py), torch.set_num_threads({threads}), best of 3 after a warm-up "
        "pass; value = one K + f step on a cached Basis (local stage incl. f(x_q) + scatter)",
        "stages_s": {
            "geometry_cache (Basis.__init__)": best["geometry"],
            "local_stage ((a(V) * dx).sum(-3): K, and f incl. f(x_q))": best["local_K"] + best["local_f"],
            "local_K": best["local_K"],
            "local_f": best["local_f"],
            "global_scatter (index_put_ into CSR values through the slot map + vector; "
            "the dense target is not a reference capability at this size)": best["scatter_K"] + best["scatter_f"],
            "scatter_K": best["scatter_K"],
            "scatter_f": best["scatter_f"],
        },
        "with_geometry_cache": {"value": n_elems / (step + best["geometry"]) / 1e6, "unit": "Melements/s"},
        "c_port": {
            "value": n_elems / c_best / 1e6,
            "unit": "Melements/s",
            "cores": c_oracle.threads(),
            "sample": "oracle/assembly_oracle.c (OpenMP, one element per iteration: local K + f, scatter) + "
            "f(x_q) by torch on the cached points, same mesh, best of 3",
        },
    }


def event_ms(fn, reps, batches=5):
    """Median over `batches` of the mean launch-to-launch time of `reps` calls (HIP events)."""
    out = []
    for _ in range(batches):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) / reps)
    return float(np.median(out))


def p2_config3():
    """BASELINE.json config 3 (P2, 6x6 blocks, S(707) = 999,698 elements) on this GPU: the
    two launches of k_p2_rows (vertex rows, edge rows).  Algorithmic bytes per SURVEY.md 8(d):
    24 B conn + 16 B per vertex + 8 B per CSR value."""
    import pytorch_fem_solver_amd as tf
    from pytorch_fem_solver_amd import meshgen

    mesh_np = meshgen.unit_square(707, 0.25, 0)
    basis = tf.Basis(tf.MeshTri(triangulation=mesh_np), tf.ElementTri(polynomial_order=2, integration_order=2))
    eng = basis._engine
    vals = eng.bilinear(1.0, 0.0)
    for _ in range(10):
        eng.bilinear(1.0, 0.0)
    torch.cuda.synchronize()
    ms = event_ms(lambda: eng.bilinear(1.0, 0.0), 10)
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


def delaunay_configs(n_points, order):
    """SURVEY 8(d) family D(N_v, seed): scipy Delaunay triangulation of a random point set, in
    its native numbering (worst case for locality) and after Morton renumbering of the vertices
    (what a caller who cares about speed does once per mesh): K alone and the fused K + f step."""
    import pytorch_fem_solver_amd as tf
    from pytorch_fem_solver_amd import meshgen
    from pytorch_fem_solver_amd.basis import forms

    native = meshgen.delaunay_square(n_points, 1)
    meshes = {
        "D_morton": meshgen.permute_mesh(native, vertex_order=meshgen.morton_order(native["vertices"])),
        "D_native": native,
    }
    out = {}
    for name, mesh_np in meshes.items():
        t0 = time.perf_counter()
        basis = tf.Basis(tf.MeshTri(triangulation=mesh_np), tf.ElementTri(1, order))
        eng = basis._engine
        program = forms.trace(load_form, basis, (), {}).coefficient.program()
        vals = eng.bilinear(1.0, 0.0)
        torch.cuda.synchronize()
        setup_ms = (time.perf_counter() - t0) * 1e3
        ne, nv, nnz = eng.n_elems, eng.n_dofs, int(vals.shape[0])
        for _ in range(30):
            eng.bilinear(1.0, 0.0)
            eng.assemble_system(1.0, 0.0, source=program)
        torch.cuda.synchronize()
        k_ms = event_ms(lambda: eng.bilinear(1.0, 0.0), 20)
        kf_ms = event_ms(lambda: eng.assemble_system(1.0, 0.0, source=program), 20)
        algo_k, algo_kf = algorithmic_bytes(ne, nv, nnz, False), algorithmic_bytes(ne, nv, nnz, True)
        rings = eng.ring_plan() if eng.kernel_name() == "k_p1_rings" else None
        out[f"{name}_{ne / 1e6:.1f}M"] = {
            "workload": f"Delaunay mesh of {nv} random points, {ne} elements, "
            f"{'Morton-renumbered vertices' if name == 'D_morton' else 'native scipy numbering'}, order {order}",
            "kernel": eng.kernel_name(),
            "renumbered_by_the_engine": bool(eng.renumbered),
            "plan": None if rings is None else ("consecutive-vertex tiles" if rings["chunked"] else "Z-order tiles"),
            "setup_ms": setup_ms,
            "stiffness_only": {"kernel_ms": k_ms, "frac": algo_k / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "fused_K_f": {"kernel_ms": kf_ms, "value": ne / kf_ms / 1e3, "unit": "Melements/s",
                          "frac": algo_kf / (kf_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
        del basis, eng, vals
        torch.cuda.empty_cache()
    # config 3 on the unstructured mesh: P2 on the Morton-renumbered Delaunay triangulation
    # (vertices with 8 .. 15 neighbours go through k_p2_long_rows)
    mesh_np = meshes["D_morton"]
    basis = tf.Basis(tf.MeshTri(triangulation=mesh_np), tf.ElementTri(2, 2))
    eng = basis._engine
    vals = eng.bilinear(1.0, 0.0)
    for _ in range(10):
        eng.bilinear(1.0, 0.0)
    torch.cuda.synchronize()
    ms = event_ms(lambda: eng.bilinear(1.0, 0.0), 10)
    ne, nv, nnz = eng.n_elems, eng.coords_per_mesh, int(vals.shape[0])
    algo = 24 * ne + 16 * nv + 8 * nnz
    plan = eng.p2_plan()
    out[f"D_morton_P2_{ne / 1e6:.1f}M"] = {
        "workload": f"P2 stiffness K (CSR), order 2, Delaunay mesh of {nv} points, {ne} elements, {eng.n_dofs} DoFs, nnz {nnz}",
        "kernel": eng.kernel_name(),
        "long_vertex_rows": None if plan is None else int(plan["layout"][18]),
        "kernel_ms": ms,
        "value": ne / ms / 1e3,
        "unit": "Melements/s",
        "frac": algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
    }
    return out


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
    from pytorch_fem_solver_amd.basis import forms

    if args.kernel != "auto":
        os.environ["TFEM_KERNEL"] = args.kernel
    torch.set_default_dtype(torch.float64)
    n = args.n
    strong = distributed and args.scaling == "strong"
    if strong:
        # ONE mesh for the whole job; rank r assembles the r-th range of the elements sorted
        # along the Morton curve of their centroids (every rank derives the same partition)
        global_mesh = meshgen.unit_square(n, 0.25, 0)
        element_order, bounds = parallel.partition_elements(
            global_mesh["vertices"], global_mesh["triangles"], world, "morton")
        mesh_np, local_to_global = parallel.extract_shard(
            global_mesh, element_order[bounds[rank]:bounds[rank + 1]])
        total_elems = int(global_mesh["triangles"].shape[0])
    else:
        # rank r owns the strip [0, 1] x [r, r+1]; identical jitter pattern per strip so that
        # the shared boundary row coincides (boundary vertices are never displaced)
        mesh_np = meshgen.structured_rectangle(n, n, 0.0, 1.0, float(rank), float(rank + 1), jitter=0.25, seed=0)
        total_elems = int(mesh_np["triangles"].shape[0]) * world
    n_elems = mesh_np["triangles"].shape[0]
    n_verts = mesh_np["vertices"].shape[0]
    t_mesh = time.perf_counter()

    torch.set_default_device(device)
    mesh = tf.MeshTri(triangulation=mesh_np)
    basis = tf.Basis(mesh, tf.ElementTri(polynomial_order=1, integration_order=args.order))
    engine = basis._engine
    csr = engine.csr_structure()
    rowptr, colind = csr[0], csr[1]
    nnz = int(colind.shape[0])
    # the caller's source, recorded once: every step evaluates it inside its launch
    traced = forms.trace(load_form, basis, (), {})
    program = traced.coefficient.program()
    assert program is not None
    exchange = None
    if distributed and strong:
        exchange = parallel.InterfaceExchange.from_partition(
            global_mesh, element_order, bounds, rank, rowptr.cpu().numpy(), colind.cpu().numpy(),
            local_to_global, device, torch.float64)
        del global_mesh
    elif distributed:
        exchange = parallel.InterfaceExchange.for_strips(mesh_np, rank, world, engine)
    # N > 1 (pytorch_fem_solver_amd/parallel.py, ShardedSteps): the launches follow each other on the
    # assembly stream into three rotating (vals, f) pairs; the exchange of step i -- pack, ONE RCCL
    # all-reduce of the packed interface entries, unpack -- runs on the exchange stream beside the
    # launch of step i + 1 (the launch leaves a few CUs free for it); over RCCL the steps are
    # recorded into HIP graphs (--graph-steps steps each, both streams and the collectives inside),
    # so a step costs the host a fraction of one graph launch.  --interface-first: the tiles owning
    # shared rows are launched first, on the exchange stream (two launches per step).
    interface_first = exchange is not None and args.interface_first
    if interface_first:
        engine.set_priority_vertices(exchange.shared_vertices(n_verts))
    if distributed:
        os.environ.setdefault("TFEM_RINGS_RESERVE_CUS", "1")  # one CU per XCD stays free for the exchange
    engine.assemble_system(1.0, 0.0, source=program)  # builds the plans, first launch
    torch.cuda.synchronize()
    setup_ms = (time.perf_counter() - t_mesh) * 1e3  # symbolic phase + plans + device copies
    if interface_first and engine.tile_range("priority")[1] in (0, engine.tile_range("all")[1]):
        interface_first = False  # nothing (or everything) is shared: one launch
    sharded = None
    if exchange is not None:
        sharded = parallel.ShardedSteps(engine, exchange, 1.0, 0.0, source=program, depth=3,
                                        interface_first=interface_first)

    def run(n):
        if sharded is not None:
            return sharded.run(n)
        out = None
        for _ in range(n):
            out = engine.assemble_system(1.0, 0.0, source=program)  # one fused launch: K and f
        return out

    def barrier():
        if sharded is not None:
            sharded.sync()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # an idle device runs its first ~20 ms slower (profiles/r01_launch_series.log), so it is
    # brought to its steady state with a fixed number of the same steps (about 40 ms) before the W
    # warm-up steps the caller asked for; reported in config.device_warmup_steps
    # (a fixed count, the same on every rank: the steps of an N > 1 run hold collectives)
    device_warmup_steps = 170
    for k in range(0, device_warmup_steps, 10):
        run(10)
        barrier() if sharded is not None else torch.cuda.synchronize()
    step_mode = "single launch per step"
    if sharded is not None:
        want_graph = args.step_mode == "graph" or (args.step_mode == "auto" and args.backend == "nccl")
        graph_steps = (min(args.graph_steps, args.steps) // 3) * 3  # whole replays inside the timed steps
        if want_graph and graph_steps >= 3 and sharded.capture(graph_steps):
            run(2 * graph_steps)  # the replays themselves, once, before anything is timed
            barrier()
        step_mode = sharded.mode
    run(args.warmup)
    if sharded is not None:
        sharded.align()  # (untimed) the timed steps start where a replay can: replays first, eager remainder last
    barrier()

    # live launch duration: HIP events on the launch stream around the WHOLE timed region (the
    # average launch-to-launch time of the K steps: what every other launch of this file is
    # measured by, event_ms) and around every probe-th step (a single launch between two event
    # packets reads 2-3 % longer).  N > 1: two streams and a collective per step -- the roofline
    # figures come from the wall clock of the region instead.
    probe = max(1, args.probe_every)
    starts = {i: torch.cuda.Event(enable_timing=True) for i in range(0, args.steps, probe)} if world == 1 else {}
    ends = {i: torch.cuda.Event(enable_timing=True) for i in starts}
    region = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    t0 = time.perf_counter()
    if world == 1:
        region[0].record()
        for i in range(args.steps):
            if i in starts:
                starts[i].record()
            out = run(1)
            if i in ends:
                ends[i].record()
        region[1].record()
        host_s = time.perf_counter() - t0
    else:
        out = run(args.steps)
        host_s = time.perf_counter() - t0  # the host's share: enqueueing (replaying) the steps
    # this rank's K steps are complete -- every exchange of theirs included, which no rank finishes
    # before all ranks have contributed -- when its streams have drained; the slowest rank's time is
    # the job's (MAX below).  The closing barrier follows OUTSIDE the clock: its own latency (a
    # collective of tens of microseconds) is not part of K steps of a few hundred.
    if sharded is not None:
        sharded.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    if distributed:
        t = torch.tensor([elapsed], device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        names = [None] * world
        dist.all_gather_object(names, f"rank {rank}: {torch.cuda.get_device_name(device)} (cuda:{device.index})")
    if world == 1:
        k_ms = region[0].elapsed_time(region[1]) / args.steps
        k_ms_probed = float(np.mean([starts[i].elapsed_time(ends[i]) for i in starts]))
    else:
        k_ms = k_ms_probed = elapsed * 1e3 / args.steps

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        algo = algorithmic_bytes(n_elems, n_verts, nnz)
        algo_k = algorithmic_bytes(n_elems, n_verts, nnz, False)
        achieved = algo / (k_ms * 1e-3) / 1e9
        profile = committed_profile(n, args.order, engine.kernel_name()) if world == 1 else {"fused": {}, "k_only": {}}
        # the stiffness-only launch (the kernel BASELINE.json's 60 % target is quoted on): back to
        # back (what a caller that re-assembles on a fixed mesh sees: its ~200 MB read set survives
        # in the 256 MB memory-side cache) and cold (every launch behind 512 MB of unrelated writes)
        k_vals = torch.empty(nnz)  # one output buffer, as in the step
        for _ in range(30):
            engine.bilinear(1.0, 0.0, out=k_vals)
        torch.cuda.synchronize()
        k_only_ms = event_ms(lambda: engine.bilinear(1.0, 0.0, out=k_vals), 50, batches=5)
        scrub = torch.empty(64 * 1024 * 1024)  # 512 MB
        cold = []
        for _ in range(10):
            scrub.fill_(1.0)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            engine.bilinear(1.0, 0.0, out=k_vals)
            b.record()
            torch.cuda.synchronize()
            cold.append(a.elapsed_time(b))
        # the same behind 512 MB of unrelated READS: the caches hold nothing of the launch's data
        # either, but no dirty lines whose write-back (up to the 256 MB of the memory-side cache)
        # shares the HBM with the launch
        cold_read = []
        for _ in range(10):
            scrub.sum()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            engine.bilinear(1.0, 0.0, out=k_vals)
            b.record()
            torch.cuda.synchronize()
            cold_read.append(a.elapsed_time(b))
        del scrub
        line = {
            "metric": "Melements/s assembled (global K + f)",
            "value": total_elems / (elapsed / args.steps) / 1e6,
            "unit": "Melements/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"P1 Poisson stiffness K (CSR) + load f of 2 pi^2 sin(pi x) sin(pi y) evaluated "
                f"in the launch, order {args.order}, "
                + (f"ONE mesh S({n},0.25,0) = {total_elems} elements cut into {world} Morton ranges"
                   if strong else f"mesh S({n},0.25,0) per GPU = {n_elems} elements, {n_verts} DoFs, nnz {nnz}"),
                "elements_per_gpu": n_elems,
                "partition": ("element ranges of one mesh along the Morton curve" if strong else
                              "one unit-square strip per rank") if world > 1 else "single mesh",
                "kernel": engine.kernel_name(),
                "source": "program recorded by the tracer (%d operations), evaluated per step inside the launch" % program.n_ops,
                "device_warmup_steps": device_warmup_steps,
                "probe_every": probe,
                "setup_ms": setup_ms,
                "interface_buffer_bytes": exchange.nbytes if exchange is not None else 0,
                "interface_tiles_first": (list(engine.tile_range("priority")) + [engine.tile_range("all")[1]]
                                          if interface_first else None),
                "baseline_config": ("C4: ONE 1e7-element P1 mesh sharded by element range, all-reduce of the shared rows"
                                    if strong else "C2/C4 kernel on one GPU" if world == 1 else
                                    "weak-scaling variant of C4 (one 1e7-element strip per GPU)"),
                "backend": (args.backend + (" (RCCL)" if args.backend == "nccl" else "")) if distributed else None,
                "world_size": dist.get_world_size() if distributed else 1,
                "devices": names if distributed else [f"rank 0: {torch.cuda.get_device_name(device)} (cuda:{device.index})"],
                "step_mode": step_mode + (f", {sharded.graph_steps} steps per HIP graph" if sharded is not None
                                          and sharded.mode == "graph" else ""),
                "graph_capture_error": getattr(sharded, "capture_error", None),
                "host_us_per_step": host_s * 1e6 / args.steps,
                "reserved_cus_per_xcd": int(os.environ.get("TFEM_RINGS_RESERVE_CUS", "0")) if distributed else 0,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": engine.kernel_name(),
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": profile["fused"].get("traffic"),
                "algorithmic_bytes_per_launch": algo,
                "kernel_ms": k_ms,
                "kernel_ms_probed": k_ms_probed,
                "launch": "fused K + f with the source evaluated in the launch (52 B/element algorithmic, "
                "SURVEY.md 8(d)); the launch is bound by fp64 vector issue (76.9 M wave-instructions at 2.3-2.4 GHz), not by HBM: DESIGN.md sections 0 and 3"
                + ("" if world == 1 else "; N > 1: kernel_ms = wall clock of the timed region / steps (two streams and a "
                   "collective per step), achieved = this rank's algorithmic bytes over it"),
                "stiffness_only": {
                    # the launch BASELINE.json's 60 % target is quoted on.  `frac` is the CACHE-FREE figure:
                    # every launch behind 512 MB of unrelated reads, so that neither the L2s nor the 256 MB
                    # memory-side cache (Infinity Cache) hold any of its ~200 MB read set.  Back to back into
                    # one buffer (what a caller re-assembling on a fixed mesh sees) that read set survives
                    # in the memory-side cache from launch to launch: `frac_back_to_back`; FETCH_SIZE
                    # counts those hits as fetched bytes, so `traffic` is fabric traffic, not HBM traffic.
                    # `frac_behind_writes`: every launch behind 512 MB of unrelated WRITES, whose
                    # write-back (up to 256 MB of dirty lines) shares the HBM with the launch.
                    "kernel_ms": float(np.median(cold_read)),
                    "frac": algo_k / (float(np.median(cold_read)) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "achieved": algo_k / (float(np.median(cold_read)) * 1e-3) / 1e9,
                    "state": "cold caches: behind 512 MB of unrelated reads",
                    "back_to_back_ms": k_only_ms,
                    "frac_back_to_back": algo_k / (k_only_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "behind_writes_ms": float(np.median(cold)),
                    "frac_behind_writes": algo_k / (float(np.median(cold)) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "traffic": profile["k_only"].get("traffic"),
                    "algorithmic_bytes_per_launch": algo_k,
                },
            },
        }
        for key, obj, ms in (("fused", line["roofline"], k_ms), ("k_only", line["roofline"]["stiffness_only"], k_only_ms)):  # traffic was counted back to back
            for extra in ("kernel_ms_rocprofv3", "valu_utilisation"):
                if extra in profile[key]:
                    obj[extra] = profile[key][extra]
            if obj["traffic"]:
                obj["traffic_GBps"] = obj["traffic"] / (ms * 1e-3) / 1e9
                obj["traffic_frac_of_peak"] = obj["traffic_GBps"] / HBM_PEAK_GBS
        if world == 1:
            line["roofline"]["profile_source"] = (
                f"profiles/{PROFILE_TAG}_bench_pmc_summary.json (kernel sources {source_sha()})"
                if profile["fused"] else "no committed profile of these kernel sources: traffic not quoted")
            # the same work through the reference's public API, per call: tracer + two launches
            def api_step():
                basis.assemble_system(stiffness_form, load_form, layout="csr")

            for _ in range(20):
                api_step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(100):
                api_step()
            torch.cuda.synchronize()
            line["api_ms_per_step"] = (time.perf_counter() - t1) * 1e3 / 100
            line["api_note"] = ("Basis.assemble_system(v_grad @ v_grad.mT, f(x_q) * v, layout='csr') per call at this mesh: "
                                "both callables traced, ONE fused launch with the source inside")

            def api_two_calls():
                basis.integrate_bilinear_form(stiffness_form, layout="csr")
                basis.integrate_linear_form(load_form)

            for _ in range(10):
                api_two_calls()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(50):
                api_two_calls()
            torch.cuda.synchronize()
            line["api_two_calls_ms_per_step"] = (time.perf_counter() - t1) * 1e3 / 50
        if world == 1 and not args.no_other_configs:
            # the same launch with other sources: what the source itself costs (the launch is
            # bound by fp64 vector issue), and with pre-evaluated source values
            # streamed from HBM (round 1's step: the evaluation of f was outside the timed region)
            x_sym, y_sym = forms.SourceExpr(basis, ("x",)), forms.SourceExpr(basis, ("y",))
            variants = {
                "constant f = 1": forms.SourceExpr(basis, ("c", 1.0)).program(),
                "polynomial f = x y + 1": (x_sym * y_sym + 1.0).program(),
            }
            sources = {"2 pi^2 sin(pi x) sin(pi y) (the step)": {"kernel_ms": k_ms, "frac": achieved / HBM_PEAK_GBS}}
            for label, prog in variants.items():
                for _ in range(20):
                    engine.assemble_system(1.0, 0.0, source=prog)
                torch.cuda.synchronize()
                ms = event_ms(lambda: engine.assemble_system(1.0, 0.0, source=prog), 30, batches=3)
                sources[label] = {"kernel_ms": ms, "frac": algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            fq = engine.source_values(program)
            for _ in range(20):
                engine.assemble_system(1.0, 0.0, fq)
            torch.cuda.synchronize()
            ms = event_ms(lambda: engine.assemble_system(1.0, 0.0, fq), 30, batches=3)
            sources["source values from HBM (f evaluated beforehand, 8 Q B/element more)"] = {
                "kernel_ms": ms, "frac": algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            ms = event_ms(lambda: engine.source_values(program), 30, batches=3)
            sources["tfem_source_eval alone (f at the integration points -> HBM)"] = {"kernel_ms": ms}
            del fq
            line["roofline"]["sources"] = sources
        if world == 1 and not args.no_other_configs:
            # what a caller who re-meshes pays per mesh: the same set-up on a second mesh of the
            # family (other jitter seed) in this process -- setup_ms above also holds what a
            # process pays once (torch loading its device kernels on first use, the library)
            del engine, basis, mesh
            other = meshgen.unit_square(n, 0.25, 1)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            mesh2 = tf.MeshTri(triangulation=other)
            basis2 = tf.Basis(mesh2, tf.ElementTri(polynomial_order=1, integration_order=args.order))
            basis2._engine.assemble_system(1.0, 0.0, source=program)
            torch.cuda.synchronize()
            line["config"]["setup_remesh_ms"] = (time.perf_counter() - t2) * 1e3
            del basis2, mesh2, other
            line["other_configs"] = {"C3_p2_stiffness_1e6": p2_config3()}
            line["other_configs"].update(delaunay_configs(args.delaunay_points, args.order))
        if world == 1 and not args.no_cpu_baseline:  # rank 0 at N = 1 only
            torch.set_default_device("cpu")
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.order)
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
