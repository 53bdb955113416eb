"""numpy walk of a P2 row plan exactly as k_p2_rows does it (csrc/tfem_p2rows.hip): decode the
vertex-row and edge-row records, evaluate one row of the element block per incident triangle
from the tile-local coordinates through the constant maps A, B, D, M, place the entries by
the recorded CSR positions, count the writes.  Test infrastructure for the host-side plan
builder."""

import numpy as np

from oracle import assembly_oracle as orc


def block_tables(integration_order):
    """(A, B, D, M), each (6, 6): K_ab = A G11 + B G12 + D G22 (stiffness), M det (mass)."""
    nodes, weights = orc.gauss_rule(integration_order)
    bary = orc.barycentric_coordinates(nodes)
    phi, rg = orc.shape_functions(2, bary, np.eye(2))  # identity inverse: reference gradients
    hw = 0.5 * np.asarray(weights).reshape(-1)
    phi = np.asarray(phi).reshape(-1, 6)
    rg = np.asarray(rg).reshape(-1, 6, 2)
    a = np.einsum("q,qa,qb->ab", hw, rg[:, :, 0], rg[:, :, 0])
    b = np.einsum("q,qa,qb->ab", hw, rg[:, :, 0], rg[:, :, 1]) + np.einsum("q,qa,qb->ab", hw, rg[:, :, 1], rg[:, :, 0])
    d = np.einsum("q,qa,qb->ab", hw, rg[:, :, 1], rg[:, :, 1])
    m = np.einsum("q,qa,qb->ab", hw, phi, phi)
    return a, b, d, m


def _block_row(tab, row, p0, p1, p2, alpha, beta):
    a, b, d, m = tab
    e1, e2 = p1 - p0, p2 - p0
    det = e1[0] * e2[1] - e1[1] * e2[0]
    g11, g12, g22 = e2.dot(e2) / det, -e1.dot(e2) / det, e1.dot(e1) / det
    return alpha * (a[row] * g11 + b[row] * g12 + d[row] * g22) + beta * m[row] * det


def run_p2_plan(plan, coords, rowptr, n_verts, integration_order=2, alpha=1.0, beta=0.0):
    """Returns (vals, writes): CSR values and how often every CSR entry was written."""
    tab = block_tables(integration_order)
    nnz = int(rowptr[-1])
    vals = np.full(nnz, np.nan)
    writes = np.zeros(nnz, dtype=np.int64)
    long_seen = set()
    # ---- vertex rows
    part = plan["vertex"]
    rows = part["rows"].reshape(-1, 8).astype(np.uint64)
    for d in part["desc"].reshape(-1, 16):
        vert_off, n_vert, row_off = int(d[0]), int(d[1]), int(d[2])
        ws = [int(d[3]), int(d[4]), int(d[5]), int(d[6]), int(d[7])]
        assert ws[0] == 0 and all(0 <= y - x <= 64 for x, y in zip(ws[:-1], ws[1:]))
        n_own = ws[4]
        gid = part["vert_gid"][vert_off:vert_off + n_vert]
        assert n_vert <= 1000 and n_vert - n_own <= 256 and np.unique(gid).size == n_vert
        xy = coords[gid]
        for wv in range(4):
            if ws[wv] < ws[wv + 1]:
                assert np.all(np.diff(gid[ws[wv]:ws[wv + 1]]) == 1) and d[8 + wv] == gid[ws[wv]]
                assert d[12 + wv] == rowptr[gid[ws[wv]]]
        for r in range(n_own):
            w = rows[row_off + r]
            k = int((w[2] >> np.uint64(24)) & np.uint64(7))
            v = int(gid[r])
            start = int(rowptr[v])
            if k == 0 and int(w[3] >> np.uint64(31)):
                # a long row (8 .. 15 neighbours): its length moves the rows behind it, the
                # entries come from the long-row record below
                assert rowptr[v + 1] - start == int(w[3] & np.uint64(0x7FFFFFFF))
                long_seen.add(v)
                continue
            if k == 0:
                assert rowptr[v + 1] == start
                continue
            ids = [int((w[i // 3] >> np.uint64(10 * (i % 3))) & np.uint64(0x3FF)) for i in range(k)]
            flags = [int((w[2] >> np.uint64(10 + 2 * i)) & np.uint64(3)) for i in range(k)]
            field = lambda f: int((w[3 + f // 6] >> np.uint64(5 * (f % 6))) & np.uint64(31))  # noqa: E731
            diag, vcol, ecol = 0.0, np.zeros(k), np.zeros(k)
            out = {}
            for i in range(k):
                nxt = 0 if i + 1 == k else i + 1
                if flags[i] == 0:
                    continue
                p1, p2 = (xy[ids[i]], xy[ids[nxt]]) if flags[i] == 1 else (xy[ids[nxt]], xy[ids[i]])
                rr = _block_row(tab, 0, xy[r], p1, p2, alpha, beta)
                i1, i2 = (i, nxt) if flags[i] == 1 else (nxt, i)
                diag += rr[0]
                vcol[i1] += rr[1]
                vcol[i2] += rr[2]
                ecol[i1] += rr[3]
                ecol[i2] += rr[5]
                out[field(14 + i)] = rr[4]
            for i in range(k):
                out[field(i)] = vcol[i]
                out[field(7 + i)] = ecol[i]
            out[int(w[2] >> np.uint64(27))] = diag
            length = 1 + 2 * k + sum(1 for f in flags if f)
            assert sorted(out) == list(range(length)) and rowptr[v + 1] - start == length
            for pos, value in out.items():
                vals[start + pos] = value
                writes[start + pos] += 1
    # ---- long vertex rows: global ids, as k_p2_long_rows walks them
    long_rows = plan["long_rows"].reshape(-1, 32).astype(np.uint64)
    assert {int(r[0]) for r in long_rows} == long_seen and len(long_rows) == len(long_seen)
    for r in long_rows:
        v, start = int(r[0]), int(r[1])
        k, dpos, flagword = int(r[2] & np.uint64(0xFF)), int(r[2] >> np.uint64(8)), int(r[3])
        assert 8 <= k <= 15 and start == rowptr[v]
        ids = [int(r[4 + i]) for i in range(k)]
        flags = [(flagword >> (2 * i)) & 3 for i in range(k)]
        field = lambda f: int((r[19 + f // 5] >> np.uint64(6 * (f % 5))) & np.uint64(63))  # noqa: E731
        diag, vcol, ecol = 0.0, np.zeros(k), np.zeros(k)
        out = {}
        for i in range(k):
            nxt = 0 if i + 1 == k else i + 1
            if flags[i] == 0:
                continue
            p1, p2 = (coords[ids[i]], coords[ids[nxt]]) if flags[i] == 1 else (coords[ids[nxt]], coords[ids[i]])
            rr = _block_row(tab, 0, coords[v], p1, p2, alpha, beta)
            i1, i2 = (i, nxt) if flags[i] == 1 else (nxt, i)
            diag += rr[0]
            vcol[i1] += rr[1]
            vcol[i2] += rr[2]
            ecol[i1] += rr[3]
            ecol[i2] += rr[5]
            out[field(30 + i)] = rr[4]
        for i in range(k):
            out[field(i)] = vcol[i]
            out[field(15 + i)] = ecol[i]
        out[dpos] = diag
        length = 1 + 2 * k + sum(1 for f in flags if f)
        assert sorted(out) == list(range(length)) and rowptr[v + 1] - start == length
        for pos, value in out.items():
            vals[start + pos] = value
            writes[start + pos] += 1
    # ---- edge rows
    part = plan["edge"]
    rows = part["rows"].reshape(-1, 4).astype(np.uint64)
    for d in part["desc"].reshape(-1, 16):
        vert_off, n_vert, row_off = int(d[0]), int(d[1]), int(d[2])
        ws = [int(d[3]), int(d[4]), int(d[5]), int(d[6]), int(d[7])]
        n_own = ws[4]
        gid = part["vert_gid"][vert_off:vert_off + n_vert]
        assert n_vert <= 1000 and np.unique(gid).size == n_vert
        xy = coords[gid]
        first = int(d[8])
        for r in range(n_own):
            w = rows[row_off + r]
            row = first + r  # a tile's edge rows are consecutive DoFs
            start = int(rowptr[row])
            a, b, c = (int((w[0] >> np.uint64(s)) & np.uint64(0x3FF)) for s in (0, 10, 20))
            dd = int(w[1] & np.uint64(0x3FF))
            has2, rev = bool((w[1] >> np.uint64(10)) & np.uint64(1)), bool((w[1] >> np.uint64(11)) & np.uint64(1))
            pos = [int((w[2] >> np.uint64(4 * f)) & np.uint64(15)) for f in range(8)] + [int(w[3] & np.uint64(15))]
            rr = _block_row(tab, 3, xy[a], xy[b], xy[c], alpha, beta)
            ss = np.zeros(6)
            if has2:
                o, t = (xy[b], xy[a]) if rev else (xy[a], xy[b])
                ss = _block_row(tab, 3, o, t, xy[dd], alpha, beta)
            out = {pos[0]: rr[0] + (ss[1] if rev else ss[0]), pos[1]: rr[1] + (ss[0] if rev else ss[1]),
                   pos[2]: rr[2], pos[3]: rr[3] + ss[3], pos[4]: rr[4], pos[5]: rr[5]}
            if has2:
                out.update({pos[6]: ss[2], pos[7]: ss[4], pos[8]: ss[5]})
            length = 9 if has2 else 6
            assert sorted(out) == list(range(length)) and rowptr[row + 1] - start == length
            for p, value in out.items():
                vals[start + p] = value
                writes[start + p] += 1
        for wv in range(4):
            if ws[wv] < ws[wv + 1]:
                assert d[8 + wv] == first + ws[wv] and d[12 + wv] == rowptr[first + ws[wv]]
    return vals, writes


def load_tables(integration_order):
    """(6, Q): phi_a(q) * w_q / 2 -- what a row multiplies the source values of a triangle with."""
    nodes, weights = orc.gauss_rule(integration_order)
    bary = orc.barycentric_coordinates(nodes)
    phi, _ = orc.shape_functions(2, bary, np.eye(2))
    return (np.asarray(phi).reshape(-1, 6) * (0.5 * np.asarray(weights).reshape(-1, 1))).T


def run_p2_load_plan(plan, coords, n_verts, n_dofs, fq, integration_order=2):
    """The load vector as k_p2_load_rows / k_p2_load_long_rows form it (csrc/tfem_p2load.hip): per row
    the triangles of the stiffness record, the element and the DoF's local index from the codes
    (element * 4 + index among the three DoFs of the row's kind).  Returns (f, writes)."""
    tab = load_tables(integration_order)
    f = np.full(n_dofs, np.nan)
    writes = np.zeros(n_dofs, dtype=np.int64)
    none = 0xFFFFFFFF

    def share(code, base, p0, p1, p2):
        e1, e2 = p1 - p0, p2 - p0
        return (e1[0] * e2[1] - e1[1] * e2[0]) * float(fq[code >> 2] @ tab[base + (code & 3)])

    part = plan["vertex"]
    rows = part["rows"].reshape(-1, 8).astype(np.uint64)
    codes = plan["vertex_codes"].reshape(-1, 8)
    assert codes.shape[0] == rows.shape[0]
    for d in part["desc"].reshape(-1, 16):
        vert_off, n_vert, row_off, n_own = int(d[0]), int(d[1]), int(d[2]), int(d[7])
        gid = part["vert_gid"][vert_off:vert_off + n_vert]
        xy = coords[gid]
        for r in range(n_own):
            w, c = rows[row_off + r], codes[row_off + r]
            k = int((w[2] >> np.uint64(24)) & np.uint64(7))
            v = int(gid[r])
            if k == 0 and int(w[3] >> np.uint64(31)):
                assert np.all(c == none)
                continue
            ids = [int((w[i // 3] >> np.uint64(10 * (i % 3))) & np.uint64(0x3FF)) for i in range(k)]
            flags = [int((w[2] >> np.uint64(10 + 2 * i)) & np.uint64(3)) for i in range(k)]
            acc = 0.0
            for i in range(8):
                if i >= k or flags[i] == 0:
                    assert c[i] == none
                    continue
                nxt = 0 if i + 1 == k else i + 1
                p1, p2 = (xy[ids[i]], xy[ids[nxt]]) if flags[i] == 1 else (xy[ids[nxt]], xy[ids[i]])
                acc += share(int(c[i]), 0, xy[r], p1, p2)
            f[v] = acc
            writes[v] += 1
    long_rows = plan["long_rows"].reshape(-1, 32).astype(np.uint64)
    long_codes = plan["long_codes"].reshape(-1, 16)
    assert long_codes.shape[0] == long_rows.shape[0]
    for r, c in zip(long_rows, long_codes):
        v, k, flagword = int(r[0]), int(r[2] & np.uint64(0xFF)), int(r[3])
        ids = [int(r[4 + i]) for i in range(k)]
        acc = 0.0
        for i in range(16):
            flag = (flagword >> (2 * i)) & 3 if i < k else 0
            if flag == 0:
                assert c[i] == none
                continue
            nxt = 0 if i + 1 == k else i + 1
            p1, p2 = (coords[ids[i]], coords[ids[nxt]]) if flag == 1 else (coords[ids[nxt]], coords[ids[i]])
            acc += share(int(c[i]), 0, coords[v], p1, p2)
        f[v] = acc
        writes[v] += 1
    part = plan["edge"]
    rows = part["rows"].reshape(-1, 4).astype(np.uint64)
    codes = plan["edge_codes"].reshape(-1, 2)
    assert codes.shape[0] == rows.shape[0]
    for d in part["desc"].reshape(-1, 16):
        vert_off, n_vert, row_off, n_own, first = int(d[0]), int(d[1]), int(d[2]), int(d[7]), int(d[8])
        xy = coords[part["vert_gid"][vert_off:vert_off + n_vert]]
        for r in range(n_own):
            w, c = rows[row_off + r], codes[row_off + r]
            a, b, cc = (int((w[0] >> np.uint64(s)) & np.uint64(0x3FF)) for s in (0, 10, 20))
            dd = int(w[1] & np.uint64(0x3FF))
            has2, rev = bool((w[1] >> np.uint64(10)) & np.uint64(1)), bool((w[1] >> np.uint64(11)) & np.uint64(1))
            acc = share(int(c[0]), 3, xy[a], xy[b], xy[cc])
            assert (c[1] != none) == has2
            if has2:
                o, t = (xy[b], xy[a]) if rev else (xy[a], xy[b])
                acc += share(int(c[1]), 3, o, t, xy[dd])
            f[first + r] = acc
            writes[first + r] += 1
    return f, writes
