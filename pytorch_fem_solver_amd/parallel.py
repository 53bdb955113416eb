"""Multi-GPU assembly: shard the mesh by element range, exchange only the shared DoFs.

The reference is single-process.  Here every rank (one process
per GPU, ``torch.distributed`` over RCCL/xGMI) assembles its own element range into its
own local CSR operator / load vector with the same HIP kernels, and the entries that
receive contributions from more than one rank -- CSR entries (i, j) and vector entries i
whose DoFs lie on an inter-rank interface -- are summed with ONE all-reduce of a packed
interface buffer.  The result stays row-distributed: each rank ends up with complete
values for every row it holds; nothing is replicated beyond the interface.

The exchange is backend-agnostic (``nccl`` = RCCL on GPUs, ``gloo`` in the CPU tests).
"""

from __future__ import annotations

import os

import numpy as np
import torch

from .meshgen import morton_order

__all__ = ["partition_elements", "extract_shard", "InterfaceExchange", "ShardedSteps"]


def partition_elements(vertices, triangles, world, order="morton"):
    """Contiguous element ranges after a space-filling-curve sort of the centroids.

    Returns ``element_order`` (permutation of element ids) and ``bounds`` (world+1):
    rank r owns ``element_order[bounds[r]:bounds[r+1]]``.
    """
    tri = np.asarray(triangles, dtype=np.int64)
    if order == "morton":
        centroids = np.asarray(vertices)[tri].mean(axis=1)
        element_order = morton_order(centroids)
    elif order == "native":
        element_order = np.arange(tri.shape[0])
    else:
        raise ValueError(order)
    bounds = np.linspace(0, tri.shape[0], world + 1).astype(np.int64)
    return element_order, bounds


def extract_shard(mesh, element_ids):
    """Local mesh of one rank: its elements with vertices renumbered 0..n_local-1 in
    ascending global id.  Returns (local mesh dict, local_to_global vertex ids)."""
    tri = np.asarray(mesh["triangles"], dtype=np.int64)[element_ids]
    local_to_global = np.unique(tri)
    local_tri = np.searchsorted(local_to_global, tri).astype(np.int32)
    local = {
        "vertices": np.ascontiguousarray(np.asarray(mesh["vertices"])[local_to_global]),
        "vertex_markers": np.ascontiguousarray(np.asarray(mesh["vertex_markers"])[local_to_global]),
        "triangles": np.ascontiguousarray(local_tri),
    }
    return local, local_to_global


def _csr_positions(rowptr, colind, rows, cols):
    """Position of (rows[k], cols[k]) in a CSR pattern with ascending columns per row."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    colind = np.asarray(colind, dtype=np.int64)
    n = rowptr.shape[0] - 1
    row_of_entry = np.repeat(np.arange(n, dtype=np.int64), np.diff(rowptr))
    width = int(colind.max()) + 1 if colind.size else 1
    keys = row_of_entry * width + colind  # ascending by construction
    want = np.asarray(rows, dtype=np.int64) * width + np.asarray(cols, dtype=np.int64)
    pos = np.searchsorted(keys, want)
    if not np.array_equal(keys[np.clip(pos, 0, keys.size - 1)], want):
        raise ValueError("interface entry missing from the local CSR pattern")
    return pos


class InterfaceExchange:
    """Packed interface buffer + one all-reduce.

    ``k_idx`` / ``f_idx``: positions in the local CSR values / local vector;
    ``k_pos`` / ``f_pos``: positions in the global interface buffer (same on every rank
    that shares the entry).  Buffer layout: [matrix entries | vector entries].
    """

    def __init__(self, k_idx, k_pos, f_idx, f_pos, n_matrix, n_vector, device, dtype, group=None):
        as_long = lambda a: torch.as_tensor(np.asarray(a, dtype=np.int64), device=device)  # noqa: E731
        self.k_idx, self.k_pos = as_long(k_idx), as_long(k_pos)
        self.f_idx, self.f_pos = as_long(f_idx), as_long(f_pos) + int(n_matrix)
        self.n_matrix, self.n_vector = int(n_matrix), int(n_vector)
        self.buffer = torch.zeros(self.n_matrix + self.n_vector, dtype=dtype, device=device)
        self.group = group

    def shared_vertices(self, n_local):
        """Flags (n_local,) of the local vertices whose rows take part in the exchange: what
        AssemblyEngine.set_priority_vertices wants (every shared matrix entry (i, j) lies in the
        row of a shared vertex i)."""
        flags = np.zeros(int(n_local), dtype=bool)
        flags[self.f_idx.cpu().numpy()] = True
        return flags

    @property
    def nbytes(self):
        return self.buffer.numel() * self.buffer.element_size()

    def _exchange_copy(self, entry, vals, f):
        """Device side of the exchange: one launch of libtfem_hip that packs the shared
        entries of vals / f into the buffer (``tfem_interface_pack``) or writes the summed
        buffer back (``tfem_interface_unpack``)."""
        from . import _native

        buf = self.buffer
        lib = _native.load()
        flat_f = f.view(-1) if f is not None else None
        ok = lambda t: t is None or (t.is_cuda and t.is_contiguous() and t.dtype == buf.dtype)  # noqa: E731
        if not (ok(vals) and ok(flat_f)):
            raise ValueError("interface exchange: vals / f must be contiguous device tensors of the buffer's dtype")
        args = [_native.ptr(vals), _native.ptr(flat_f), buf.element_size(), _native.ptr(self.k_idx),
                _native.ptr(self.k_pos), self.k_idx.numel(), _native.ptr(self.f_idx),
                _native.ptr(self.f_pos), self.f_idx.numel(), _native.ptr(buf)]
        if entry == "pack":
            args.append(buf.numel())
        with torch.cuda.device(buf.device):
            args.append(_native.current_stream(buf.device))
            _native.check(getattr(lib, "tfem_interface_" + entry)(*args))

    def prepared(self, vals, f):
        """(pack, unpack) callables for these device tensors with every argument converted once:
        each only enqueues its launch on the current stream (the steps of a sharded run are bound
        by the host otherwise).  The tensors must stay alive and in place."""
        from . import _native

        buf = self.buffer
        if not buf.is_cuda:
            return (lambda stream=None: self.pack(vals, f)), (lambda stream=None: self.unpack(vals, f))
        lib = _native.load()
        flat_f = f.view(-1) if f is not None else None
        ok = lambda t: t is None or (t.is_cuda and t.is_contiguous() and t.dtype == buf.dtype)  # noqa: E731
        if not (ok(vals) and ok(flat_f)):
            raise ValueError("interface exchange: vals / f must be contiguous device tensors of the buffer's dtype")
        head = (_native.ptr(vals), _native.ptr(flat_f), buf.element_size(), _native.ptr(self.k_idx),
                _native.ptr(self.k_pos), self.k_idx.numel(), _native.ptr(self.f_idx),
                _native.ptr(self.f_pos), self.f_idx.numel(), _native.ptr(buf))
        keep = (self, vals, flat_f, self.k_idx, self.k_pos, self.f_idx, self.f_pos, buf)  # what `head` points into
        device, current_stream = buf.device, torch.cuda.current_stream
        from ctypes import c_void_p

        def run(fn, args):
            def call(stream=None, _keep=keep):  # the callables keep the tensors alive
                handle = (stream if stream is not None else current_stream(device)).cuda_stream
                status = fn(*args, c_void_p(handle))
                if status:
                    _native.check(status)
            return call

        unpack = run(lib.tfem_interface_unpack, head)
        if vals is not None and flat_f is not None and os.environ.get("TFEM_INTERFACE_PACK", "dense") == "dense":
            # pack in ONE launch (no memset in front): the exchange chain of a step -- pack, all-reduce,
            # unpack, each behind the other -- bounds the step at ~1e6 elements per rank
            if getattr(self, "_src", None) is None:
                src = torch.full((buf.numel(),), -1, dtype=torch.int64, device=device)
                src[self.k_pos] = self.k_idx
                src[self.f_pos] = -self.f_idx - 2
                self._src = src
            dense = (_native.ptr(vals), _native.ptr(flat_f), buf.element_size(), _native.ptr(self._src),
                     buf.numel(), _native.ptr(buf))
            keep = keep + (self._src,)
            return run(lib.tfem_interface_pack_dense, dense), unpack
        return run(lib.tfem_interface_pack, head + (buf.numel(),)), unpack

    def pack(self, vals=None, f=None):
        """Zero the interface buffer and copy this rank's shared entries into it."""
        buf = self.buffer
        if buf.is_cuda:
            self._exchange_copy("pack", vals, f)
            return buf
        # host tensors (the gloo tests of the exchange logic): the same copies with torch
        buf.zero_()
        if vals is not None:
            buf[self.k_pos] = vals[self.k_idx]
        if f is not None:
            buf[self.f_pos] = f.view(-1)[self.f_idx]
        return buf

    def unpack(self, vals=None, f=None):
        """Write the (summed) interface buffer back into vals / f, in place."""
        buf = self.buffer
        if buf.is_cuda:
            self._exchange_copy("unpack", vals, f)
            return vals, f
        if vals is not None:
            vals[self.k_idx] = buf[self.k_pos]
        if f is not None:
            f.view(-1)[self.f_idx] = buf[self.f_pos]
        return vals, f

    def reduce(self, vals=None, f=None):
        """Sum the shared entries of ``vals`` (local CSR values) and ``f`` (local vector,
        any shape with N_local entries) across ranks, in place: pack, ONE all-reduce of the
        packed buffer, unpack."""
        import torch.distributed as dist

        self.pack(vals, f)
        dist.all_reduce(self.buffer, op=dist.ReduceOp.SUM, group=self.group)
        return self.unpack(vals, f)

    def reduce_on(self, stream, vals=None, f=None, record=True):
        """``reduce`` enqueued on a side stream behind the work already queued on the current
        stream, so that the next (independent) assembly launch overlaps the exchange.
        Returns the event that marks the end of the exchange on ``stream``: the caller waits
        on it (``torch.cuda.current_stream().wait_event``) before it reuses or reads vals / f.
        ``record=True`` also tells the caching allocator that ``stream`` uses the tensors;
        callers that rotate their own preallocated buffers and wait on the event pass False."""
        ready = torch.cuda.Event()
        ready.record()  # everything queued so far on the current stream (the assembly kernel)
        with torch.cuda.stream(stream):
            stream.wait_event(ready)
            if record:
                for t in (vals, f):
                    if t is not None:
                        t.record_stream(stream)
            self.reduce(vals, f)
            done = torch.cuda.Event()
            done.record(stream)
        return done

    # ------------------------------------------------------------------ constructors
    @classmethod
    def from_partition(cls, mesh, element_order, bounds, rank, rowptr, colind, local_to_global,
                       device, dtype, group=None):
        """General element-range partition of ONE global mesh (BASELINE config 4).

        Every rank derives the same global interface numbering from the global
        connectivity, so no communication is needed to set up the exchange.
        """
        tri = np.asarray(mesh["triangles"], dtype=np.int64)
        n_global = int(np.asarray(mesh["vertices"]).shape[0])
        world = len(bounds) - 1
        touches = np.zeros(n_global, dtype=np.int32)
        for r in range(world):
            ids = element_order[bounds[r]:bounds[r + 1]]
            touches[np.unique(tri[ids])] += 1
        shared_vertex = touches > 1
        shared_ids = np.nonzero(shared_vertex)[0]
        # interface matrix entries: ordered pairs of shared vertices inside one element
        in_elem = shared_vertex[tri]
        cand = tri[in_elem.sum(axis=1) >= 1]
        pairs = np.stack(
            [np.repeat(cand, 3, axis=1).reshape(-1), np.tile(cand, (1, 3)).reshape(-1)], axis=1
        )
        pairs = pairs[shared_vertex[pairs[:, 0]] & shared_vertex[pairs[:, 1]]]
        global_keys = np.unique(pairs[:, 0] * n_global + pairs[:, 1])
        # what this rank holds of it
        l2g = np.asarray(local_to_global, dtype=np.int64)
        rowptr = np.asarray(rowptr, dtype=np.int64)
        row_of_entry = np.repeat(np.arange(rowptr.shape[0] - 1, dtype=np.int64), np.diff(rowptr))
        gi, gj = l2g[row_of_entry], l2g[np.asarray(colind, dtype=np.int64)]
        mine = np.nonzero(shared_vertex[gi] & shared_vertex[gj])[0]
        k_pos = np.searchsorted(global_keys, gi[mine] * n_global + gj[mine])
        f_idx = np.nonzero(shared_vertex[l2g])[0]
        f_pos = np.searchsorted(shared_ids, l2g[f_idx])
        return cls(mine, k_pos, f_idx, f_pos, global_keys.size, shared_ids.size, device, dtype, group)

    @classmethod
    def for_strips(cls, mesh, rank, world, engine, group=None):
        """Weak-scaling layout of bench.py: rank r holds the structured strip
        [0, 1] x [r, r+1] (n x n cells, vertex id iy*(n+1)+ix); its top row is the bottom row
        of rank r+1, so the shared vertices are two runs of consecutive ids (a handful of
        the ring plan's tiles, contiguous pack / unpack).  Interface k (between ranks k and
        k+1) carries, for the n+1 row vertices, the diagonal entries, the 2n entries of the
        horizontal edges and the n+1 vector entries."""
        n = int(round(np.sqrt(mesh["triangles"].shape[0] // 2)))
        nvx = n + 1
        csr = engine.csr_structure()
        rowptr, colind = csr[0].cpu().numpy(), csr[1].cpu().numpy()
        ix = np.arange(nvx, dtype=np.int64)
        per_k = nvx + 2 * n
        k_idx, k_pos, f_idx, f_pos = [], [], [], []
        for interface, iy in ((rank - 1, 0), (rank, n)):
            if interface < 0 or interface >= world - 1:
                continue
            line = iy * nvx + ix
            rows = np.concatenate([line, line[:-1], line[1:]])
            cols = np.concatenate([line, line[1:], line[:-1]])
            k_idx.append(_csr_positions(rowptr, colind, rows, cols))
            k_pos.append(interface * per_k + np.arange(per_k))
            f_idx.append(line)
            f_pos.append(interface * nvx + ix)
        cat = lambda parts: np.concatenate(parts) if parts else np.zeros(0, dtype=np.int64)  # noqa: E731
        return cls(cat(k_idx), cat(k_pos), cat(f_idx), cat(f_pos), max(world - 1, 0) * per_k,
                   max(world - 1, 0) * nvx, engine.device, engine.dtype, group)


class ShardedSteps:
    """The repeated step of a sharded run (BASELINE config 4): every rank assembles K and f of its
    element range into one of `depth` preallocated (vals, f) pairs and sums the entries it shares
    with other ranks -- pack, ONE all-reduce of the packed interface buffer (RCCL), unpack.

    The assembly launches follow each other on the assembly stream; the exchange of step i runs on
    the exchange stream beside the launch of step i + 1 (steps are independent: the pair of step i
    is written again by step i + depth, which waits for its exchange on the device).  The host side
    of a step is four enqueue calls and a collective; over RCCL `capture()` records `graph_steps`
    consecutive steps -- both streams, the collectives included -- into ONE HIP graph, so that a
    step costs the host a fraction of a graph launch (tools/time_step_host_overhead.py).

    interface_first: the ring plan lists the tiles owning shared vertices first
    (engine.set_priority_vertices before the plan is built); a step then launches those on the
    exchange stream, in front of its exchange, and the rest on the assembly stream (the shared rows
    reach the other ranks a launch earlier; two launches per step)."""

    def __init__(self, engine, exchange, alpha, beta, source=None, fq=None, depth=3, interface_first=False):
        import torch.distributed as dist

        self.dist = dist
        self.engine, self.exchange, self.depth = engine, exchange, int(depth)
        device = engine.device
        nnz = int(engine.csr_structure()[1].shape[0])
        self.pairs = [(torch.empty(nnz, dtype=engine.dtype, device=device),
                       torch.empty(engine.n_dofs, dtype=engine.dtype, device=device)) for _ in range(self.depth)]
        self.main = torch.cuda.Stream(device=device)
        self.comm = torch.cuda.Stream(device=device, priority=-1)
        self.main.wait_stream(torch.cuda.current_stream(device))  # behind the set-up (plan copies ...)
        self.interface_first = bool(interface_first)
        self.launches = []
        for pair in self.pairs:
            pack, unpack = exchange.prepared(*pair)
            if self.interface_first:
                first = engine.prepared_system(alpha, beta, pair, fq=fq, source=source, tiles="priority")
                rest = engine.prepared_system(alpha, beta, pair, fq=fq, source=source, tiles="rest")
            else:
                first, rest = None, engine.prepared_system(alpha, beta, pair, fq=fq, source=source)
            self.launches.append((first, rest, pack, unpack))
        self.counter = 0
        self.done = [None] * self.depth      # end of the exchange that last used the pair (eager steps)
        self.graph, self.graph_steps = None, 0
        self.mode, self.capture_error = "eager", None

    # one step's work on the two streams; `done`: per pair the event of its last exchange
    def _enqueue(self, slot, done):
        first, rest, pack, unpack = self.launches[slot]
        main, comm = self.main, self.comm
        if done[slot] is not None:
            main.wait_event(done[slot])      # the pair is free: its exchange of `depth` steps ago is over
        fork = torch.cuda.Event()
        fork.record(main)
        comm.wait_event(fork)                # (also what makes the exchange stream part of a capture)
        if first is not None:
            first(comm)                      # rows of the shared vertices, in front of their exchange
            rest(main)
        else:
            rest(main)
            assembled = torch.cuda.Event()
            assembled.record(main)
            comm.wait_event(assembled)
        pack(comm)
        with torch.cuda.stream(comm):
            self.dist.all_reduce(self.exchange.buffer, op=self.dist.ReduceOp.SUM, group=self.exchange.group)
        unpack(comm)
        if first is not None:                # the pair is complete when BOTH streams are through
            rest_done = torch.cuda.Event()
            rest_done.record(main)
            comm.wait_event(rest_done)
        done[slot] = torch.cuda.Event()
        done[slot].record(comm)

    def capture(self, graph_steps=None):
        """Record `graph_steps` steps (a multiple of depth; default 2 * depth) into one HIP graph.
        Returns True when replays are in use from now on; on any failure the eager steps stay."""
        steps = int(graph_steps or 2 * self.depth)
        if steps % self.depth:
            raise ValueError("graph_steps must be a multiple of depth")
        self.sync()
        try:
            graph = torch.cuda.CUDAGraph()
            # thread_local: the process group's watchdog thread may query its events meanwhile
            with torch.cuda.graph(graph, stream=self.main, capture_error_mode="thread_local"):
                done = [None] * self.depth
                for i in range(steps):
                    self._enqueue(i % self.depth, done)
                self.main.wait_stream(self.comm)  # join: the capture ends on one stream
        except Exception as exc:  # noqa: BLE001 -- whatever the capture refuses: eager steps
            self.capture_error = f"{type(exc).__name__}: {exc}"
            torch.cuda.synchronize(self.engine.device)
            return False
        self.graph, self.graph_steps, self.mode = graph, steps, "graph"
        return True

    def run(self, n):
        """Enqueue n steps; returns the pair of the last one (complete after sync())."""
        n = int(n)
        while n > 0:
            if self.graph is not None and self.counter % self.depth == 0 and n >= self.graph_steps:
                if any(e is not None for e in self.done):
                    self.main.wait_stream(self.comm)
                    self.done = [None] * self.depth
                with torch.cuda.stream(self.main):
                    self.graph.replay()
                self.counter += self.graph_steps
                n -= self.graph_steps
                continue
            if self.graph is not None and all(e is None for e in self.done):
                # behind a replay everything is complete at the END of the assembly stream's queue
                self.comm.wait_stream(self.main)
            self._enqueue(self.counter % self.depth, self.done)
            self.counter += 1
            n -= 1
        return self.pairs[(self.counter - 1) % self.depth]

    def align(self):
        """Eager steps until the next step writes pair 0 again -- where a replay can start.  For
        callers that time a run of steps: with the replays in FRONT, the host's time for the eager
        steps of the remainder hides behind the device's queue instead of starving it."""
        extra = (-self.counter) % self.depth
        if extra:
            self.run(extra)
        return extra

    def sync(self):
        """Host wait for everything enqueued so far (both streams)."""
        self.main.synchronize()
        self.comm.synchronize()
