"""One rank of the sharded assembly (BASELINE config 4) on the GPU, end to end: shard of ONE mesh
(Morton element ranges) or a strip, ring plan with the interface tiles first, the bench's
two-stream step (interface rows + exchange on the exchange stream, the rest on the assembly stream),
then every entry this rank holds against the operator of the WHOLE mesh assembled on the same GPU.

Started by tests/test_hip_distributed.py under torch.distributed.run (gloo: several ranks share the
one card of the test box; over RCCL the same code runs with --backend nccl).  Every rank prints one JSON
line and, with --out-dir, writes it to rank<r>.json."""
import argparse
import json
import math
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def rhs(x, y):
    return 2.0 * math.pi**2 * torch.sin(math.pi * x) * torch.sin(math.pi * y)


def load_form(basis):
    x, y = torch.split(basis.integration_points, 1, dim=-1)
    return rhs(x, y) * basis.v


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--layout", choices=("partition", "strips"), default="partition")
    p.add_argument("--backend", default="gloo")
    p.add_argument("--points", type=int, default=20000)
    p.add_argument("--grid", type=int, default=96)
    p.add_argument("--out-dir", default=None, help="rank r writes rank<r>.json there (stdout lines of ranks interleave)")
    p.add_argument("--stepper", choices=("inline", "sharded", "sharded-first", "graph", "graph-first"), default="inline",
                   help="inline: the two-stream step spelled out here; sharded / graph: parallel.ShardedSteps, eager "
                   "or recorded into HIP graphs (graph needs --backend nccl), -first: interface tiles first")
    args = p.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist.init_process_group(args.backend, **({"device_id": device} if args.backend == "nccl" else {}))

    import pytorch_fem_solver_amd as tf
    from pytorch_fem_solver_amd import meshgen, parallel
    from pytorch_fem_solver_amd.basis import forms

    torch.set_default_dtype(torch.float64)
    torch.set_default_device(device)
    n = args.grid
    if args.layout == "partition":
        whole = meshgen.delaunay_square(args.points, seed=11)
        whole = meshgen.permute_mesh(whole, vertex_order=meshgen.morton_order(whole["vertices"]))
        order, bounds = parallel.partition_elements(whole["vertices"], whole["triangles"], world, "morton")
        mesh_np, l2g = parallel.extract_shard(whole, order[bounds[rank]:bounds[rank + 1]])
    else:  # the strips of bench.py's weak scaling, stacked: the whole domain is [0,1] x [0,world]
        whole = meshgen.structured_rectangle(n, n * world, 0.0, 1.0, 0.0, float(world), jitter=0.0)
        mesh_np = meshgen.structured_rectangle(n, n, 0.0, 1.0, float(rank), float(rank + 1), jitter=0.0)
        l2g = rank * n * (n + 1) + np.arange((n + 1) * (n + 1))
    nv = mesh_np["vertices"].shape[0]
    basis = tf.Basis(tf.MeshTri(triangulation=mesh_np), tf.ElementTri(1, 3))
    eng = basis._engine
    csr = eng.csr_structure()
    rowptr, colind = csr[0].cpu().numpy(), csr[1].cpu().numpy()
    if world == 1:
        # one rank (the RCCL rehearsal on a one-GPU box): nothing is shared, so the exchange gets an
        # arbitrary set of entries to carry through pack -> all-reduce over one rank -> unpack
        k_idx, f_idx = np.arange(0, colind.shape[0], 7)[:5000], np.arange(0, nv, 3)[:2000]
        ex = parallel.InterfaceExchange(k_idx, np.arange(k_idx.size), f_idx, np.arange(f_idx.size), k_idx.size,
                                        f_idx.size, device, torch.float64)
    elif args.layout == "partition":
        ex = parallel.InterfaceExchange.from_partition(whole, order, bounds, rank, rowptr, colind, l2g, device, torch.float64)
    else:
        ex = parallel.InterfaceExchange.for_strips(mesh_np, rank, world, eng)
    eng.set_priority_vertices(ex.shared_vertices(nv))
    program = forms.trace(load_form, basis, (), {}).coefficient.program()
    n_pri, n_all = eng.tile_range("priority")[1], eng.tile_range("all")[1]
    nnz = int(colind.shape[0])
    mode = "inline"
    if args.stepper == "inline":
        out = (torch.full((nnz,), float("nan")), torch.full((nv,), float("nan")))
        comm = torch.cuda.Stream(device=device, priority=-1)
        for _ in range(3):  # the two-stream step, three times into the same buffers
            torch.cuda.current_stream().wait_stream(comm)
            with torch.cuda.stream(comm):
                comm.wait_stream(torch.cuda.current_stream())
                eng.assemble_system(1.0, 0.5, source=program, out=out, tiles="priority")
                ex.reduce(*out)
            eng.assemble_system(1.0, 0.5, source=program, out=out, tiles="rest")
        torch.cuda.current_stream().wait_stream(comm)
        torch.cuda.synchronize()
    else:  # the steps of bench.py: rotating pairs, exchange beside the next launch, optionally as HIP graphs
        steps = parallel.ShardedSteps(eng, ex, 1.0, 0.5, source=program, depth=3,
                                      interface_first=args.stepper.endswith("-first") and 0 < n_pri < n_all)
        for pair in steps.pairs:
            pair[0].fill_(float("nan"))
            pair[1].fill_(float("nan"))
        steps.run(4)
        if args.stepper.startswith("graph"):
            assert steps.capture(6), steps.capture_error
        out = steps.run(6 + 6 + 2)  # replays and eager steps, in this order and mixed
        assert steps.align() == (-(4 + 14)) % 3 and steps.counter % 3 == 0  # bench.py: timed steps start at a replay
        out = steps.run(6 + 1)
        steps.sync()
        mode = steps.mode
        for pair in steps.pairs:  # every pair holds the same complete result (K bit for bit)
            assert torch.equal(pair[0], out[0])
            assert float((pair[1] - out[1]).abs().max()) <= 1e-14 * float(out[1].abs().max())

    # the whole mesh on this GPU (every rank for itself)
    gbasis = tf.Basis(tf.MeshTri(triangulation=whole), tf.ElementTri(1, 3))
    geng = gbasis._engine
    gprogram = forms.trace(load_form, gbasis, (), {}).coefficient.program()
    gvals, gf = geng.assemble_system(1.0, 0.5, source=gprogram)
    gcsr = geng.csr_structure()
    g_rowptr, g_colind = gcsr[0].cpu().numpy(), gcsr[1].cpu().numpy()
    # position of (l2g[row], l2g[col]) in the global pattern
    rows = np.repeat(np.arange(nv), np.diff(rowptr))
    pos = parallel._csr_positions(g_rowptr, g_colind, l2g[rows], l2g[colind])
    want_v = gvals.view(-1)[torch.from_numpy(pos).to(device)]
    want_f = gf.view(-1)[torch.from_numpy(np.asarray(l2g)).to(device)]
    # an entry is complete here when every element that contributes lies in this shard or the
    # entry is shared; entries of interior rows of OTHER ranks do not exist in this pattern
    err_v = float((out[0] - want_v).abs().max() / want_v.abs().max())
    err_f = float((out[1] - want_f).abs().max() / want_f.abs().max())
    result = json.dumps({"rank": rank, "world": world, "layout": args.layout, "n_local_vertices": int(nv), "mode": mode,
                         "priority_tiles": n_pri, "tiles": n_all, "interface_entries": int(ex.n_matrix + ex.n_vector),
                         "err_values": err_v, "err_vector": err_f})
    if args.out_dir:
        with open(os.path.join(args.out_dir, f"rank{rank}.json"), "w") as fh:
            fh.write(result)
    print(result, flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
