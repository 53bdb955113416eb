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
    if args.layout == "partition":
        ex = parallel.InterfaceExchange.from_partition(whole, order, bounds, rank, rowptr, colind, l2g, device, torch.float64)
    else:
        ex = parallel.InterfaceExchange.for_strips(mesh_np, rank, world, eng)
    eng.set_priority_vertices(ex.shared_vertices(nv))
    program = forms.trace(load_form, basis, (), {}).coefficient.program()
    n_pri, n_all = eng.tile_range("priority")[1], eng.tile_range("all")[1]
    nnz = int(colind.shape[0])
    out = (torch.full((nnz,), float("nan")), torch.full((nv,), float("nan")))
    comm = torch.cuda.Stream(device=device, priority=-1)
    for _ in range(3):  # the step of bench.py, three times into the same buffers
        torch.cuda.current_stream().wait_stream(comm)
        with torch.cuda.stream(comm):
            comm.wait_stream(torch.cuda.current_stream())
            eng.assemble_system(1.0, 0.5, source=program, out=out, tiles="priority")
            ex.reduce(*out)
        eng.assemble_system(1.0, 0.5, source=program, out=out, tiles="rest")
    torch.cuda.current_stream().wait_stream(comm)
    torch.cuda.synchronize()

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
    result = json.dumps({"rank": rank, "world": world, "layout": args.layout, "n_local_vertices": int(nv),
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
