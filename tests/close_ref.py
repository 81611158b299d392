"""Host restatement (numpy / plain Python) of dn_close_units_build_i32 (include/dn_hip.h): the per-tile entry lists with membership
masks and the workgroup-major unit records of the H = 256 closing launch.  Test infrastructure only."""
import numpy as np

DEDUP_CAP = 256          # kCbCap in csrc/dn_close.hip: tiles with more raw list entries are listed without merging


def close_units_ref(list_ptr, list_rows, N, P, G, drop=(0, 0)):
    """-> (unit_ptr [G+1], units [U,4], tiles: {t: (first entry offset, [rows], [masks])})."""
    lp, lr = np.asarray(list_ptr, dtype=np.int64), np.asarray(list_rows, dtype=np.int64)
    T = (N + 31) // 32
    tiles = {}
    for t in range(T):
        p0, pend = 32 * t, min(32 * t + 32, N)
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
    Tper = (T + G - 1) // G if T else 0
    unit_ptr, units = [0], []
    for w in range(G):
        for n in range(Tper):
            t = n * G + w
            if t >= T:
                continue
            e0, ent_r, _ = tiles[t]
            c, p0, pend = len(ent_r), 32 * t, min(32 * t + 32, N)
            ne = (c + 31) // 32
            units.append([2 if ne == 0 else 0, p0, pend, t])
            for i in range(ne):
                units.append([1 | (2 if i == ne - 1 else 0), e0 + 32 * i, e0 + min(32 * (i + 1), c), p0])
        unit_ptr.append(len(units))
    return np.array(unit_ptr, dtype=np.int64), np.array(units, dtype=np.int64).reshape(-1, 4), tiles
