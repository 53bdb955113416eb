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
    desc = plan["desc"].reshape(-1, 8)
    rec = plan["records"].reshape(-1, 3)
    vals = np.full(nnz, np.nan)
    writes = np.zeros(nnz, dtype=np.int64)
    for t in range(n_tiles):
        elem_off, n_elem, vert_off, n_vert, n_own, row_off, acc_size, loff_off = desc[t]
        assert n_elem <= sizes[5] and n_vert <= sizes[6] and n_own <= sizes[7] and acc_size <= sizes[8]
        gid = plan["vert_gid"][vert_off:vert_off + n_vert]
        assert gid.min() >= 0 and gid.max() < n_verts
        assert np.unique(gid).size == gid.size
        assert np.all(np.diff(gid[:n_own]) > 0)  # owned rows ascending
        xy = coords[gid]
        loff = plan["row_loff"][loff_off:loff_off + n_own + 1].astype(np.int64)
        assert loff[0] == 0 and loff[-1] == acc_size and np.all(np.diff(loff) >= 0)
        acc = np.zeros(acc_size)
        r = rec[elem_off:elem_off + n_elem]
        lid = (r & 0xFFF).astype(np.int64)
        assert lid.max(initial=0) < n_vert
        blocks = local_blocks_by_coords(xy[lid])
        for j in range(3):
            owned = lid[:, j] < n_own
            base = loff[np.where(owned, lid[:, j], 0)]
            for i in range(3):
                pos = ((r[:, j] >> (12 + 4 * i)) & 0xF).astype(np.int64)
                row_len = loff[np.where(owned, lid[:, j], 0) + 1] - base
                assert np.all(pos[owned] < row_len[owned])
                np.add.at(acc, (base + pos)[owned], blocks[owned, i, j])
        gstart = plan["row_gstart"][row_off:row_off + n_own].astype(np.int64)
        for row in range(n_own):
            seg = slice(gstart[row], gstart[row] + loff[row + 1] - loff[row])
            vals[seg] = acc[loff[row]:loff[row + 1]]
            writes[seg] += 1
    return vals, writes
