"""numpy walk of a ring plan exactly as k_p1_rings does it (csrc/tfem_rings.hip): decode the
row records, evaluate the fan of every owned row from the tile-local coordinates, permute
into CSR order, count the writes.  Test infrastructure for the host-side plan builder."""

import numpy as np


def decode_rows(rows, slots):
    """rows (n, words) uint32 -> dict of (n, slots) arrays id / flag / pos and (n,) k / dpos."""
    w = rows.astype(np.uint64)
    n = w.shape[0]
    ids = np.zeros((n, slots), dtype=np.int64)
    flag = np.zeros((n, slots), dtype=np.int64)
    pos = np.zeros((n, slots), dtype=np.int64)
    for i in range(slots):
        ids[:, i] = (w[:, i // 3] >> np.uint64(10 * (i % 3))) & np.uint64(0x3FF)
        if slots == 7:
            flag[:, i] = (w[:, 2] >> np.uint64(10 + 2 * i)) & np.uint64(3)
            pos[:, i] = (w[:, 3] >> np.uint64(3 * i)) & np.uint64(7)
        else:
            flag[:, i] = (w[:, 5] >> np.uint64(2 * i)) & np.uint64(3)
            pos[:, i] = (w[:, 6 + (i >> 3)] >> np.uint64(4 * (i & 7))) & np.uint64(15)
    if slots == 7:
        k = (w[:, 0] >> np.uint64(30)) | (((w[:, 1] >> np.uint64(30)) & np.uint64(1)) << np.uint64(2))
        dpos = (w[:, 2] >> np.uint64(24)) & np.uint64(7)
    else:
        k = (w[:, 0] >> np.uint64(30)) | ((w[:, 1] >> np.uint64(30)) << np.uint64(2))
        dpos = (w[:, 2] >> np.uint64(30)) | ((w[:, 3] >> np.uint64(30)) << np.uint64(2))
    return {"id": ids, "flag": flag, "pos": pos, "k": k.astype(np.int64), "dpos": dpos.astype(np.int64)}


def chain_block_starts(plan):
    """First position of every block of hand-overs: the runs of the plan (one per resident
    workgroup), or -- plans with flagged vertices -- blocks of chain_len positions that break between
    the two launch ranges."""
    n, clen = int(plan["n_tiles"]), int(plan["chain_len"])
    starts = np.zeros(n, dtype=bool)
    if int(plan["n_runs"]) <= 0:
        starts[::clen] = True
        if 0 < plan.get("n_priority", 0) < n:
            starts[plan["n_priority"]] = True
        return starts
    runs = plan["runs"]
    assert runs[0] == 0 and runs[-1] == n and np.all(np.diff(runs) >= 0)
    starts[runs[:-1][np.diff(runs) > 0]] = True
    return starts


def source_load_vector(plan, coords, source, lam, lamw, tiles=None, conn=None):
    """The load vector of a source program as the SRC launches form it: the tiles in CHAIN ORDER,
    blocks of chain_len positions per workgroup; a tile evaluates the elements of its own table
    (tile_tverts, desc[18] >> 8 of them) from its tile-local coordinates and adds the three shares
    det_T g[T][i] to the sums of the elements' tile-LOCAL vertices, halo included; a row takes its
    own sum and what the tile before it in the block summed for its vertex (hand_in: the vertex's
    local id there).  tiles = (first, count): positions of the chain order."""
    from oracle.assembly_oracle import source_program_eval

    desc = plan["desc"].reshape(-1, 20)
    order, clen, hand_in = plan["chain_order"], int(plan["chain_len"]), plan["hand_in"]
    n = desc.shape[0]
    assert sorted(order.tolist()) == list(range(n)) and clen >= 1
    u0, u1 = (0, n) if tiles is None else (tiles[0], tiles[0] + tiles[1])
    starts = chain_block_starts(plan)
    assert tiles is None or int(plan["n_runs"]) <= 0, "tile ranges: a plan with flagged vertices"
    fvec = np.full(coords.shape[0], np.nan)
    for u in range(u0, u1):
        d = desc[order[u]]
        vert_off, n_vert, row_off, n_own, n_elem, n_tv, tv_off = (
            int(d[0]), int(d[1]), int(d[2]), int(d[7]), int(d[17]), int(d[18]) >> 8, int(d[19]))
        assert n_tv <= n_elem and n_tv <= 768
        gid = plan["vert_gid"][vert_off:vert_off + n_vert]
        xy = coords[gid]
        tv = plan["tile_tverts"][tv_off:tv_off + n_tv].astype(np.int64)
        local = np.stack([tv & 0x3FF, (tv >> 10) & 0x3FF, (tv >> 20) & 0x3FF], axis=1).reshape(-1, 3)
        assert n_tv == 0 or local.max() < n_vert
        if conn is not None and n_tv:  # every table entry is an element, in the element's own local order
            have = {tuple(r) for r in conn[np.nonzero(np.isin(conn, gid[:n_own]).any(axis=1))[0]].tolist()}
            assert all(tuple(r) in have for r in gid[local].tolist())
        cx, cy = xy[local][..., 0], xy[local][..., 1]
        xq = (np.outer(cx[:, 0], lam[0]) + np.outer(cx[:, 1], lam[1])) + np.outer(cx[:, 2], lam[2])
        yq = (np.outer(cy[:, 0], lam[0]) + np.outer(cy[:, 1], lam[1])) + np.outer(cy[:, 2], lam[2])
        fq_tile = source_program_eval(source[0], source[1], xq, yq) if n_tv else np.zeros((0, lam.shape[1]))
        det = (cx[:, 1] - cx[:, 0]) * (cy[:, 2] - cy[:, 0]) - (cx[:, 2] - cx[:, 0]) * (cy[:, 1] - cy[:, 0])
        acc = np.zeros(n_vert)
        for i in range(3):
            np.add.at(acc, local[:, i], det * (fq_tile @ lamw[i]))
        hin = hand_in[row_off:row_off + n_own].astype(np.int64)
        first_of_block = bool(starts[u]) or u == u0
        if first_of_block:  # nobody hands anything to the first tile of a block or of a launch's range
            assert np.all(hin == 0xFFFF)
        else:
            # the previous tile's sums are indexed by ITS local ids: the ids name this tile's vertices
            assert np.array_equal(prev_gid[hin[hin != 0xFFFF]], gid[:n_own][hin != 0xFFFF])
        handed = np.where(hin != 0xFFFF, prev_acc[np.minimum(hin, prev_acc.size - 1)], 0.0) if not first_of_block else 0.0
        fvec[gid[:n_own]] = acc[:n_own] + handed
        prev_acc, prev_gid = acc, gid
    return fvec


def run_ring_plan(plan, coords, nnz, stiff_w=0.5, mass_d=0.0, mass_o=0.0, fq=None, lamw=None,
                  conn=None, source=None, lam=None, tiles=None):
    """Returns (vals, writes, covered[, f]): CSR values of stiff_w-weighted stiffness + mass,
    how often every CSR entry was written, the number of rows covered and -- with source
    values fq (E, Q) and the table lamw (3, Q) = l_i(q) w_q / 2 -- the load vector.
    conn: the connectivity, to check the tiles' element vertex tables against.  source = (ops,
    consts) with lam (3, Q) = l_i(q): the source values are not taken from fq but computed per
    tile from the tile-local coordinates through tile_tverts, as the SRC kernels do.
    tiles = (first, count): that range of the tile list only (tfem_p1_assemble_rings_range)."""
    from oracle.assembly_oracle import source_program_eval

    slots, words = plan["slots"], plan["words"]
    desc = plan["desc"].reshape(-1, 20)
    rows = plan["rows"].reshape(-1, words)
    vals = np.full(nnz, np.nan)
    writes = np.zeros(nnz, dtype=np.int64)
    covered = 0
    long_seen = {}
    fvec = np.full(coords.shape[0], np.nan) if (fq is not None or source is not None) else None
    fsource = source_load_vector(plan, coords, source, lam, lamw, tiles, conn) if source is not None else None
    ewords = (12 * slots + 31) // 32  # packed 12-bit slot codes
    row_ecodes = plan["row_ecodes"].reshape(-1, ewords)
    if tiles is not None:
        desc = desc[tiles[0]:tiles[0] + tiles[1]]
    for d in desc:
        vert_off, n_vert, row_off, ws0, ws1, ws2, ws3, n_own = (int(x) for x in d[:8])
        wave_start = [ws0, ws1, ws2, ws3, n_own]
        assert ws0 == 0 and n_vert - n_own <= 256
        assert all(0 <= b - a <= 64 for a, b in zip(wave_start[:-1], wave_start[1:]))
        gid = plan["vert_gid"][vert_off:vert_off + n_vert]
        assert n_own <= n_vert <= 1024
        assert np.all(np.diff(gid[:n_own]) > 0), "owned rows ascending"
        if plan.get("chunked"):
            for a, b in zip(wave_start[:-1], wave_start[1:]):  # a wave's rows are consecutive vertices
                assert np.all(np.diff(gid[a:b]) == 1)
        assert np.unique(gid).size == n_vert
        xy = coords[gid]
        rec = decode_rows(rows[row_off:row_off + n_own], slots)
        elem_off, n_elem = int(d[16]), int(d[17])
        if int(d[18]) & 0xFF:  # <= 8 runs of consecutive ids: first ids, then list positions they end at
            first, upto = plan["tile_elems"][elem_off:elem_off + 8], plan["tile_elems"][elem_off + 8:elem_off + 16]
            pos = np.arange(n_elem)
            tile_elems = first[0] + pos
            for r in range(1, 8):
                tile_elems = np.where(pos >= upto[r - 1], first[r] + (pos - upto[r - 1]), tile_elems)
            assert upto[7] == n_elem
        else:
            tile_elems = plan["tile_elems"][elem_off:elem_off + n_elem]
        assert np.all(np.diff(tile_elems) > 0) and (n_elem <= 768 or not plan["elems_staged"])
        facc_tile = None
        if source is not None:  # element form, in the order of the SRC launches: source_load_vector
            facc_tile = fsource[gid]
        rowstart = plan["rowstart"][row_off:row_off + n_own]
        for w, (a, b) in enumerate(zip(wave_start[:-1], wave_start[1:])):
            if b > a:  # what the consecutive-vertex kernel takes from the descriptor
                assert d[8 + w] == gid[a] and d[12 + w] == rowstart[a]
        covered += n_own
        raw = rows[row_off:row_off + n_own]
        for r in range(n_own):
            k = int(rec["k"][r])
            if k == 0 and slots == 7 and int(raw[r][3]) >> 31:
                # a long row (8 .. 15 neighbours): written from the long-row list below; its sum of
                # the load vector comes from the element-form accumulators (source programs only)
                long_seen[int(gid[r])] = int(raw[r][3]) & 0x7FFFFFFF
                if fvec is not None:
                    fvec[gid[r]] = facc_tile[r] if facc_tile is not None else np.nan
                continue
            if k == 0:
                if fvec is not None:
                    fvec[gid[r]] = 0.0
                continue
            ids, flag, pos = rec["id"][r], rec["flag"][r], rec["pos"][r]
            assert ids[:k].max() < n_vert and np.all(ids[:k] != r)
            e = xy[ids[:k]] - xy[r]
            off = np.zeros(k)
            diag = 0.0
            facc = 0.0
            ew = row_ecodes[row_off + r]
            bits = sum(int(ew[j]) << (32 * j) for j in range(ewords))
            se = np.array([(bits >> (12 * i)) & 0xFFF for i in range(slots)])
            assert np.all(se[k:] == 0xFFF)
            for i in range(k):
                nxt = 0 if i + 1 == k else i + 1
                if flag[i] == 0:
                    assert se[i] == 0xFFF
                    continue
                dvec = e[nxt] - e[i]
                cross = e[i, 0] * e[nxt, 1] - e[i, 1] * e[nxt, 0]
                sdet = cross if flag[i] == 1 else -cross
                cs = stiff_w / sdet
                diag += cs * dvec.dot(dvec) + mass_d * sdet
                off[i] += -cs * dvec.dot(e[nxt]) + mass_o * sdet
                off[nxt] += cs * dvec.dot(e[i]) + mass_o * sdet
                if fvec is not None and facc_tile is None:
                    elem, loc = int(tile_elems[se[i] & 0x3FF]), int(se[i] >> 10)
                    assert loc < 3
                    facc += sdet * float(np.dot(fq[elem], lamw[loc]))
            targets = np.concatenate([rowstart[r] + pos[:k], [rowstart[r] + rec["dpos"][r]]])
            assert np.unique(targets).size == k + 1
            assert targets.max() < rowstart[r] + k + 1
            vals[targets] = np.concatenate([off, [diag]])
            writes[targets] += 1
            if fvec is not None:
                fvec[gid[r]] = facc if facc_tile is None else facc_tile[r]
    # ---- long rows: global ids, as k_p1_long_rows walks them
    long_rows = plan.get("long_rows", np.zeros(0, dtype=np.uint32)).reshape(-1, 24)
    assert {int(r[0]): None for r in long_rows}.keys() == long_seen.keys() and len(long_rows) == len(long_seen)
    for r in long_rows:
        v, start = int(r[0]), int(r[1])
        k, dpos, flagword = int(r[2]) & 0xFF, int(r[2]) >> 8, int(r[3])
        assert 8 <= k <= 15 and long_seen[v] == k + 1
        ids = [int(r[4 + i]) for i in range(k)]
        pos = [(int(r[19 + i // 8]) >> (4 * (i % 8))) & 15 for i in range(k)]
        e = coords[ids] - coords[v]
        off, diag = np.zeros(k), 0.0
        for i in range(k):
            nxt = 0 if i + 1 == k else i + 1
            flag = (flagword >> (2 * i)) & 3
            if flag == 0:
                continue
            dvec = e[nxt] - e[i]
            cross = e[i, 0] * e[nxt, 1] - e[i, 1] * e[nxt, 0]
            sdet = cross if flag == 1 else -cross
            cs = stiff_w / sdet
            diag += cs * dvec.dot(dvec) + mass_d * sdet
            off[i] += -cs * dvec.dot(e[nxt]) + mass_o * sdet
            off[nxt] += cs * dvec.dot(e[i]) + mass_o * sdet
        targets = np.concatenate([start + np.asarray(pos), [start + dpos]])
        assert np.unique(targets).size == k + 1 and targets.max() < start + k + 1
        vals[targets] = np.concatenate([off, [diag]])
        writes[targets] += 1
    if fvec is not None:
        return vals, writes, covered, fvec
    return vals, writes, covered
