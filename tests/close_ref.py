"""Host restatement (numpy / plain Python) of dn_close_units_build_i32 (include/dn_hip.h): the per-tile entry lists with membership
masks and the workgroup-major unit records of the H = 256 closing launch.  Test infrastructure only."""
import numpy as np

DEDUP_CAP = 256          # kCbCap in csrc/dn_close.hip: tiles with more raw list entries are listed without merging


AGG_GAP = 8              # kAggGap: NOP units between a workgroup's tiles and its AGG units


def graph_tiles_ref(seg_ptr, seg_nodes, N, add_idx=None):
    """dn_fold_graph_tiles_build_i32: -> (ok, tile_ptr [S+1], fold_info [S,12] int32).  add_idx [S]: the row each segment's
    product is added to; it must lie inside the segment's own block."""
    sp, sn = np.asarray(seg_ptr, dtype=np.int64), np.asarray(seg_nodes, dtype=np.int64)
    S = len(sp) - 1
    firsts, lasts = [], []
    for j in range(S):
        nodes = sn[sp[j]:sp[j + 1]]
        if len(nodes) == 0 or np.any(np.diff(nodes) != 1) or nodes[0] < 0 or nodes[-1] >= N:
            return False, None, None
        firsts.append(int(nodes[0])); lasts.append(int(nodes[-1]))
    tile_ptr = np.zeros(S + 1, dtype=np.int64)
    info = np.zeros((S, 12), dtype=np.int32)
    for j in range(S):
        b0 = 0 if j == 0 else firsts[j]
        nxt = firsts[j + 1] if j + 1 < S else N
        if j + 1 < S and nxt <= lasts[j]:
            return False, None, None
        if not (1 <= nxt - b0 <= 32):
            return False, None, None
        if add_idx is not None and not (b0 <= int(add_idx[j]) < nxt):
            return False, None, None
        tile_ptr[j] = b0
        ids = np.array([0 if firsts[j] <= b0 + i <= lasts[j] else 255 for i in range(32)], dtype=np.uint8)
        info[j, :8] = ids.view(np.int32)
        info[j, 8], info[j, 9], info[j, 10] = j, 1, 2
    tile_ptr[S] = N
    return True, tile_ptr, info


def graph_tiles_multi_ref(seg_ptr, seg_nodes, N, C, add_idx=None):
    """dn_fold_graph_tiles_multi_build_i32: -> (ok, chunk_tile [C+1], chunk_graph [C+1], tile_ptr [T+1], fold_info [T,12]): C chunks cut at
    graph boundaries (graph j in chunk b0_j C // N), every chunk cut into consecutive 32-node tiles that run across its graphs."""
    sp, sn = np.asarray(seg_ptr, dtype=np.int64), np.asarray(seg_nodes, dtype=np.int64)
    S = len(sp) - 1
    firsts, lasts = [], []
    for j in range(S):
        nodes = sn[sp[j]:sp[j + 1]]
        if len(nodes) == 0 or np.any(np.diff(nodes) != 1) or nodes[0] < 0 or nodes[-1] >= N:
            return False, None, None, None, None
        firsts.append(int(nodes[0])); lasts.append(int(nodes[-1]))
    gstart = [0 if j == 0 else firsts[j] for j in range(S)] + [N]
    for j in range(S):
        if (j + 1 < S and gstart[j + 1] <= lasts[j]) or gstart[j + 1] - gstart[j] < 1:
            return False, None, None, None, None
        if add_idx is not None and not (gstart[j] <= int(add_idx[j]) < gstart[j + 1]):
            return False, None, None, None, None
    chunk_graph = [next((j for j in range(S) if gstart[j] >= -(-c * N // C)), S) for c in range(C)] + [S]
    chunk_tile, tile_ptr, info = [0], [], []
    for c in range(C):
        n_lo, n_hi = gstart[chunk_graph[c]], gstart[chunk_graph[c + 1]]
        for p0 in range(n_lo, n_hi, 32):
            pend = min(p0 + 32, n_hi)
            tile_ptr.append(p0)
            ids, present = [], []
            for v in range(p0, p0 + 32):
                j = max(k for k in range(chunk_graph[c], chunk_graph[c + 1]) if gstart[k] <= v) if v < pend else -1
                if v < pend and firsts[j] <= v <= lasts[j]:
                    if j not in present:
                        present.append(j)
                    ids.append(j - present[0])
                else:
                    ids.append(255)
            rec = np.zeros(12, dtype=np.int32)
            rec[:8] = np.array(ids, dtype=np.uint8).view(np.int32)
            if present:
                rec[8], rec[9] = present[0], present[-1] - present[0] + 1
                rec[10] = (1 if firsts[present[0]] < p0 else 0) | (2 if lasts[present[-1]] < pend else 0)
            info.append(rec)
        chunk_tile.append(len(tile_ptr))
    tile_ptr.append(N)
    return True, np.array(chunk_tile), np.array(chunk_graph), np.array(tile_ptr), np.array(info, dtype=np.int32).reshape(-1, 12)


def tiles_of_workgroup(w, T, G, order):
    """the tiles of workgroup w, in the order it takes them (include/dn_hip.h: xcd_order)"""
    if not order:
        return list(range(w, T, G))
    x, j, W8 = w % 8, w // 8, G // 8
    lo, hi = x * T // 8, (x + 1) * T // 8
    return list(range(hi - 1 - j, lo - 1, -W8))


def close_units_ref(list_ptr, list_rows, N, P, G, drop=(0, 0), tile_ptr=None, agg=False, order=0, multi=None):
    """-> (unit_ptr [G+1], units [U,4], tiles: {t: (first entry offset, [rows], [masks])}).  orders 2 / 3 with multi = (chunk_tile,
    chunk_graph), K G + 1 entries each: the CHUNKS are dealt to the workgroups as orders 0 / 1 deal tiles; a workgroup's stream = the
    tiles of its chunks, the gap, one AGG record per 32 graphs of each chunk."""
    xcd_order = int(order)                             # (`order` is reused below for a sort permutation)
    lp, lr = np.asarray(list_ptr, dtype=np.int64), np.asarray(list_rows, dtype=np.int64)
    T = (N + 31) // 32 if tile_ptr is None else len(tile_ptr) - 1
    tp = np.minimum(np.arange(T + 1) * 32, N) if tile_ptr is None else np.asarray(tile_ptr, dtype=np.int64)
    tiles = {}
    for t in range(T):
        p0, pend = int(tp[t]), int(min(tp[t + 1], tp[t] + 32))
        kept = []
        for v in range(p0, pend):
            kept.append([int(r) for r in lr[lp[v]:lp[v + 1]] if r < P and not (drop[0] <= r < drop[1])])
        plain = int(lp[pend] - lp[p0]) > DEDUP_CAP
        if not plain:                                  # a row that one node lists twice NOT next to each other: list the tile as is
            for rows in kept:
                seen = set()
                for k, r in enumerate(rows):
                    if r in seen and rows[k - 1] != r:
                        plain = True
                    seen.add(r)
        ent_r, ent_m, where = [], [], {}
        for i, rows in enumerate(kept):
            prev = None
            for r in rows:
                if plain or r == prev:                 # a node's own repeat stays a separate entry
                    ent_r.append(r); ent_m.append(1 << i)
                elif r in where:
                    ent_m[where[r]] |= 1 << i
                else:
                    where[r] = len(ent_r)
                    ent_r.append(r); ent_m.append(1 << i)
                prev = r
        if not plain:                                  # common path: a tile's entries leave sorted by row (stable)
            order = sorted(range(len(ent_r)), key=lambda k: (ent_r[k], k))
            ent_r, ent_m = [ent_r[k] for k in order], [ent_m[k] for k in order]
        tiles[t] = (int(lp[p0]), ent_r, ent_m)
    unit_ptr, units = [0], []
    if xcd_order >= 2:
        ct, cg = (np.asarray(a, dtype=np.int64) for a in multi)
        C = len(ct) - 1
        for w in range(G):
            mine = tiles_of_workgroup(w, C, G, xcd_order - 2)
            for c in mine:
                for t in range(int(ct[c]), int(ct[c + 1])):
                    e0, ent_r, _ = tiles[t]
                    cn, p0, pend = len(ent_r), int(tp[t]), int(min(tp[t + 1], tp[t] + 32))
                    ne, rows = (cn + 31) // 32, (pend - p0) << 8
                    units.append([(2 if ne == 0 else 0) | rows, p0, pend, t])
                    for i in range(ne):
                        units.append([1 | (2 if i == ne - 1 else 0) | rows, e0 + 32 * i, e0 + min(32 * (i + 1), cn), p0])
            if agg and any(ct[c + 1] > ct[c] for c in mine):
                units += [[8, 0, 1, 0]] * AGG_GAP
                for c in mine:
                    units += [[4 | 2 | 32, g0, min(g0 + 32, int(cg[c + 1])), 0] for g0 in range(int(cg[c]), int(cg[c + 1]), 32)]
            unit_ptr.append(len(units))
        return np.array(unit_ptr, dtype=np.int64), np.array(units, dtype=np.int64).reshape(-1, 4), tiles
    for w in range(G):
        nw = 0
        for t in tiles_of_workgroup(w, T, G, xcd_order):
            nw += 1
            e0, ent_r, _ = tiles[t]
            c, p0, pend = len(ent_r), int(tp[t]), int(min(tp[t + 1], tp[t] + 32))
            ne, rows = (c + 31) // 32, (pend - p0) << 8
            units.append([(2 if ne == 0 else 0) | rows, p0, pend, t])
            for i in range(ne):
                units.append([1 | (2 if i == ne - 1 else 0) | rows, e0 + 32 * i, e0 + min(32 * (i + 1), c), p0])
        if agg and nw:
            units += [[8, 0, 1, 0]] * AGG_GAP
            units += [[4 | 2 | (16 if xcd_order else 0), 32 * i, min(32 * (i + 1), nw), T if xcd_order else 0] for i in range((nw + 31) // 32)]
        unit_ptr.append(len(units))
    return np.array(unit_ptr, dtype=np.int64), np.array(units, dtype=np.int64).reshape(-1, 4), tiles
