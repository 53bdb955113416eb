"""N > 1 path on CPU: world_size-2 gloo runs of the element-range sharding and the packed
interface all-reduce (pytorch_fem_solver_amd/parallel.py).  Local assembly is done by the
CPU oracle here (tests may use it); on GPUs the same exchange runs over RCCL on the values
the HIP kernels produce."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import assembly_oracle as orc


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _local_system(mesh, order=3):
    nv = mesh["vertices"].shape[0]
    rowptr, colind, slots = orc.csr_pattern(mesh["triangles"], nv)
    k_local, _ = orc.p1_assemble(mesh["vertices"], mesh["triangles"], order, "stiffness_mass")
    f_local, _ = orc.p1_assemble(mesh["vertices"], mesh["triangles"], order, "load")
    vals = orc.assemble_csr_values(k_local, slots, colind.shape[0])
    f = orc.assemble_linear(f_local, mesh["triangles"], nv).reshape(-1)
    return rowptr, colind, vals, f


class _FakeEngine:
    """What InterfaceExchange.for_strips needs from an AssemblyEngine."""

    def __init__(self, rowptr, colind):
        self._csr = (torch.from_numpy(rowptr), torch.from_numpy(colind), None)
        self.device = torch.device("cpu")
        self.dtype = torch.float64

    def csr_structure(self):
        return self._csr


def _worker_partition(rank, world, port, order_kind, results, mesh_kind="delaunay"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pytorch_fem_solver_amd import meshgen, parallel

        # "structured": the S(n) family bench.py --scaling strong cuts into Morton ranges
        mesh = meshgen.delaunay_square(900, 4) if mesh_kind == "delaunay" else meshgen.unit_square(24, 0.25, 0)
        n_global = mesh["vertices"].shape[0]
        element_order, bounds = parallel.partition_elements(mesh["vertices"], mesh["triangles"], world, order_kind)
        mine = element_order[bounds[rank]:bounds[rank + 1]]
        local, l2g = parallel.extract_shard(mesh, mine)
        rowptr, colind, vals, f = _local_system(local)
        ex = parallel.InterfaceExchange.from_partition(
            mesh, element_order, bounds, rank, rowptr, colind, l2g, torch.device("cpu"), torch.float64
        )
        tv, tf_ = torch.from_numpy(vals.copy()), torch.from_numpy(f.copy())
        ex.reduce(tv, tf_)
        # the global operator assembled in one piece
        g_rowptr, g_colind, g_vals, g_f = _local_system(mesh)
        dense = orc.csr_to_dense(g_rowptr, g_colind, g_vals, n_global)
        rows = np.repeat(np.arange(rowptr.shape[0] - 1), np.diff(rowptr))
        want = dense[l2g[rows], l2g[colind]]
        # rows of vertices all of whose elements are local or shared are complete after the
        # exchange; an entry is incomplete only if a third rank-free element were missing
        err_k = np.abs(tv.numpy() - want).max() / np.abs(want).max()
        owned_or_shared = np.ones(l2g.shape[0], dtype=bool)
        err_f = np.abs(tf_.numpy() - g_f[l2g])[owned_or_shared].max() / np.abs(g_f).max()
        results[rank] = (float(err_k), float(err_f), int(ex.n_matrix), int(ex.n_vector),
                         int(np.sum(np.abs(vals - tv.numpy()) > 0)))
    finally:
        dist.destroy_process_group()


def _worker_strips(rank, world, port, results):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pytorch_fem_solver_amd import meshgen, parallel

        n = 6
        strips = [meshgen.structured_rectangle(n, n, 0.0, 1.0, float(r), float(r + 1), jitter=0.25, seed=0)
                  for r in range(world)]
        systems = [_local_system(s) for s in strips]
        rowptr, colind, vals, f = systems[rank]
        ex = parallel.InterfaceExchange.for_strips(strips[rank], rank, world, _FakeEngine(rowptr, colind))
        tv, tf_ = torch.from_numpy(vals.copy()), torch.from_numpy(f.copy())
        ex.reduce(tv, tf_)
        # expected: own values + the neighbour's values on the shared row of vertices
        nvx = n + 1
        want_v, want_f = vals.copy(), f.copy()
        dense = [orc.csr_to_dense(s[0], s[1], s[2], nvx * nvx) for s in systems]
        for other, my_iy, their_iy in ((rank - 1, 0, n), (rank + 1, n, 0)):
            if other < 0 or other >= world:
                continue
            mine_col = my_iy * nvx + np.arange(nvx)
            theirs_col = their_iy * nvx + np.arange(nvx)
            want_f[mine_col] += systems[other][3][theirs_col]
            rows = np.repeat(np.arange(nvx * nvx), np.diff(rowptr))
            pos_of = {v: i for i, v in enumerate(mine_col)}
            for k in range(colind.shape[0]):
                if rows[k] in pos_of and colind[k] in pos_of:
                    want_v[k] += dense[other][theirs_col[pos_of[rows[k]]], theirs_col[pos_of[colind[k]]]]
        results[rank] = (float(np.abs(tv.numpy() - want_v).max()), float(np.abs(tf_.numpy() - want_f).max()),
                         int(ex.n_matrix), int(ex.n_vector))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("order_kind", ["morton", "native"])
def test_element_range_partition_and_interface_all_reduce(order_kind):
    world = 2
    with mp.Manager() as manager:
        results = manager.dict()
        mp.spawn(_worker_partition, args=(world, _free_port(), order_kind, results), nprocs=world, join=True)
        assert len(results) == world
        for rank in range(world):
            err_k, err_f, n_matrix, n_vector, changed = results[rank]
            assert err_k <= 1e-13 and err_f <= 1e-13, (rank, results[rank])
            assert 0 < n_vector < 900 and n_matrix >= n_vector
            assert changed > 0  # the exchange really added the neighbour's share
        assert results[0][2:4] == results[1][2:4]  # same global interface numbering


def test_strong_scaling_partition_of_one_mesh_over_four_ranks():
    """bench.py --scaling strong (BASELINE config 4): ONE S(n) mesh cut into four element ranges
    along the Morton curve; after the exchange every rank holds complete values on all its rows."""
    world = 4
    with mp.Manager() as manager:
        results = manager.dict()
        mp.spawn(_worker_partition, args=(world, _free_port(), "morton", results, "structured"), nprocs=world, join=True)
        assert len(results) == world
        for rank in range(world):
            err_k, err_f, n_matrix, n_vector, changed = results[rank]
            assert err_k <= 1e-13 and err_f <= 1e-13, (rank, results[rank])
            assert changed > 0
        assert len({results[r][2:4] for r in range(world)}) == 1  # one global interface numbering
        assert 0 < results[0][3] < 200  # interface vertices: a few grid lines of the 625


def test_weak_scaling_strips_exchange():
    world = 2
    with mp.Manager() as manager:
        results = manager.dict()
        mp.spawn(_worker_strips, args=(world, _free_port(), results), nprocs=world, join=True)
        for rank in range(world):
            err_v, err_f, n_matrix, n_vector = results[rank]
            assert err_v <= 1e-13 and err_f <= 1e-13, results[rank]
            assert n_matrix == 3 * 6 + 1 and n_vector == 7


def test_partition_helpers():
    from pytorch_fem_solver_amd import meshgen, parallel

    mesh = meshgen.unit_square(10, 0.25, 0)
    order, bounds = parallel.partition_elements(mesh["vertices"], mesh["triangles"], 3)
    assert sorted(order.tolist()) == list(range(200)) and bounds.tolist() == [0, 66, 133, 200]
    local, l2g = parallel.extract_shard(mesh, order[bounds[1]:bounds[2]])
    assert local["triangles"].shape == (67, 3)
    assert np.array_equal(local["vertices"], mesh["vertices"][l2g])
    assert np.array_equal(l2g[local["triangles"]], mesh["triangles"][order[bounds[1]:bounds[2]]])
