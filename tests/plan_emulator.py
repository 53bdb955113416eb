"""Host emulation of the tile kernel's data flow (tests only): walks a tile plan exactly
as k_p1_bilinear_tiles does -- records -> local ids/positions -> LDS accumulators ->
row write-out -- but takes the per-element 3x3 blocks from the CPU oracle.  Validates
the plan (bounds, ownership, positions) without a GPU."""

import numpy as np


def run_plan(plan, local_blocks_by_coords, n_verts, nnz, coords):
    """local_blocks_by_coords(xy (k,3,2)) -> (k,3,3) blocks.  Returns vals (nnz,) and a
    write count per CSR entry."""
    sizes = plan["sizes"]
    n_tiles = int(sizes[0])
    desc = plan["desc"].reshape(-1, 12)
    words = int(plan.get("rec_words", 3))
    rec = plan["records"].reshape(-1, words)
    vals = np.full(nnz, np.nan)
    writes = np.zeros(nnz, dtype=np.int64)
    for t in range(n_tiles):
        (elem_off, n_elem, vert_off, n_vert, n_own, acc_size, loff_off, run_off, n_runs,
         lrun_off) = desc[t][:10]
        assert n_runs <= sizes[10]
        assert n_elem <= sizes[5] and n_vert <= sizes[6] and n_own <= sizes[7] and acc_size <= sizes[8]
        gid = plan["vert_gid"][vert_off:vert_off + n_vert]
        assert gid.min() >= 0 and gid.max() < n_verts
        assert np.unique(gid).size == gid.size
        assert np.all(np.diff(gid[:n_own]) > 0)  # owned rows ascending
        xy = coords[gid]
        loff = np.append(plan["row_loff"][loff_off:loff_off + n_own].astype(np.int64), acc_size)
        assert loff[0] == 0 and np.all(np.diff(loff) >= 0)
        acc = np.zeros(acc_size)
        r = rec[elem_off:elem_off + n_elem]
        if words == 3:
            assert np.all((r & 0xF) == 0)
            lid = ((r & 0xFFFF) >> 4).astype(np.int64)
        else:
            lid = np.stack([(r[:, 0] >> (10 * j)) & 0x3FF for j in range(3)], axis=1).astype(np.int64)
        assert lid.max(initial=0) < n_vert
        blocks = local_blocks_by_coords(xy[lid])
        for j in range(3):
            owned = lid[:, j] < n_own
            base = loff[np.where(owned, lid[:, j], 0)]
            for i in range(3):
                if words == 3:
                    pos = ((r[:, j] >> (16 + 4 * i)) & 0xF).astype(np.int64)
                else:
                    pos = ((r[:, 1] >> (3 * (3 * j + i))) & 0x7).astype(np.int64)
                row_len = loff[np.where(owned, lid[:, j], 0) + 1] - base
                assert np.all(pos[owned] < row_len[owned])
                np.add.at(acc, (base + pos)[owned], blocks[owned, i, j])
        runl = plan["run_lstart"][lrun_off:lrun_off + n_runs + 1].astype(np.int64)
        rund = plan["run_delta"][run_off:run_off + n_runs].astype(np.int64)
        assert runl[-1] == acc_size and (n_runs == 0 or runl[0] == 0) and np.all(np.diff(runl) > 0)
        for r in range(n_runs):
            seg = slice(runl[r] + rund[r], runl[r + 1] + rund[r])
            vals[seg] = acc[runl[r]:runl[r + 1]]
            writes[seg] += 1
    return vals, writes
