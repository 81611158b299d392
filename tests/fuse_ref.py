"""Host (numpy) builder of the unified unit stream of dn_rows_fused_bf16 (include/dn_hip.h) from a RowIndex's device tables: the
closing launch's per-tile units regrouped by chunks of the batch, with the transform units of every chunk dealt to the
workgroups of their relation, the publish / gate flags and the spacing the hand-off needs.  Test and experiment infrastructure
(tests/test_gpu_fused.py, tools/fused_exp.py): the fused launch is an EXPERIMENTAL entry point, not on the product path
(docs/LAB_NOTES.md, round 5)."""
import numpy as np
import torch

U_ENTRY, U_LAST, U_AGG, U_NOP, U_T, U_PUB, U_GATE = 1, 2, 4, 8, 16, 32, 64
AGG_GAP = 8                                          # kAggGap (csrc/dn_close.hip, dn_fuse.hip)


def relation_workgroups(tiles_per_rel, G):
    """k_r workgroups for relation r, proportional to its tile count (>= 1 where it has rows), summing to G."""
    t = np.asarray(tiles_per_rel, dtype=np.float64)
    live = t > 0
    assert live.sum() <= G
    k = np.where(live, np.maximum(1, np.floor(G * t / max(t.sum(), 1.0))), 0).astype(np.int64)
    while k.sum() > G:                                   # (the floors of tiny relations were raised to 1)
        k[np.argmax(np.where(k > 1, k / np.maximum(t, 1e-9), -1))] -= 1
    while k.sum() < G:                                   # hand the left-over workgroups to the relations with most tiles per workgroup
        k[np.argmax(np.where(live, t / np.maximum(k, 1), -1))] += 1
    return k


def build(ix, direction, ops, chunk_tiles=1024, lead=2, only=None, gates=True):
    """-> dict(unit_ptr, units (device int32), num_wg, num_chunks, stats) for one direction of a RowIndex whose fold is absorbed."""
    fold = ops._row_index_fold(ix, direction, "units")
    cu = ix.close_units(direction)
    assert fold is not None and cu.agg, "the fused launch needs the absorbed fold (every graph inside one tile)"
    dev = cu.units.device
    G, T = cu.num_wg, cu.num_tiles
    units = cu.units.cpu().numpy().astype(np.int64)
    uptr = cu.unit_ptr.cpu().numpy()
    tile_ptr = fold.graph_tiles[0].cpu().numpy().astype(np.int64)
    src_rows = (ix.row_in if direction == "f" else ix.row_out).cpu().numpy().astype(np.int64)
    R = ix.num_rels
    rel_ptr = np.asarray(ix.rel_ptr_host[:R + 1], dtype=np.int64)
    # ---- the closing units of every tile and every workgroup's tail (gap + AGG units), out of the device-built table
    tile_units, tails = [None] * T, []
    for w in range(G):
        recs = units[uptr[w]:uptr[w + 1]]
        i = 0
        while i < len(recs) and not (recs[i, 0] & (U_NOP | U_AGG)):
            assert not (recs[i, 0] & U_ENTRY)
            t = int(recs[i, 3])
            j = i + 1
            while j < len(recs) and (recs[j, 0] & U_ENTRY):
                j += 1
            tile_units[t] = recs[i:j]
            i = j
        tails.append(recs[i:])
    assert all(u is not None for u in tile_units)
    nc = -(-T // chunk_tiles)
    assert nc <= 250
    # ---- transform tiles per (chunk, relation), dealt to the relation's workgroups
    live = [r for r in range(R) if r != fold.rel and rel_ptr[r + 1] > rel_ptr[r]]
    tiles_per_rel = np.zeros(R)
    bounds = {}
    for r in live:
        a, b = int(rel_ptr[r]), int(rel_ptr[r + 1])
        tile_of = np.searchsorted(tile_ptr, src_rows[a:b], side="right") - 1
        ch = tile_of // chunk_tiles
        assert np.all(np.diff(ch) >= 0), "rows of a relation must be in batch order"
        bounds[r] = a + np.searchsorted(ch, np.arange(nc + 1), side="left")
        tiles_per_rel[r] = sum(-(-(int(bounds[r][c + 1]) - int(bounds[r][c])) // 32) for c in range(nc))
    k = relation_workgroups(tiles_per_rel, G)
    first_wg = np.concatenate([[0], np.cumsum(k)])
    rel_of_wg = np.repeat(np.arange(R), k)
    tparts = [[[] for _ in range(nc)] for _ in range(G)]
    for r in live:
        kr, o = int(k[r]), int(first_wg[r])
        for c in range(nc):
            a, b = int(bounds[r][c]), int(bounds[r][c + 1])
            for i, beg in enumerate(range(a, b, 32)):
                w = o + (i + c) % kr
                tparts[w][c].append((U_T | (r << 16), beg, min(beg + 32, b), beg))
    # ---- streams.  A gate is examined 7 positions ahead of the compute position, and the workgroup must have COMPUTED its own
    # last transform unit of that chunk by then: at least 8 units between them (zero-row transform units as padding).
    out, ptr = [], [0]
    n_t = n_c = n_pad = 0
    for w in range(G):
        recs = []
        pub_pos = {}
        for c in range(nc + lead):
            if c < nc and only != "C":
                tl = tparts[w][c] or [(U_T | (int(rel_of_wg[w]) << 16), 0, 0, 0)]       # nothing to transform here: count in all the same
                tl = list(tl)
                f, a, b, y = tl[-1]
                tl[-1] = (f | U_PUB | (c << 24), a, b, y)
                recs += tl
                n_t += len(tl)
                pub_pos[c] = len(recs) - 1
            cc = c - lead
            if cc >= 0 and only != "T":
                t0, t1 = cc * chunk_tiles, min((cc + 1) * chunk_tiles, T)
                first = True
                for t in range(t0 + ((w - t0) % G), t1, G):
                    tu = tile_units[t].copy()
                    if first and only is None and gates:
                        while len(recs) - pub_pos[cc] < 9:
                            recs.append((U_T | (int(rel_of_wg[w]) << 16), 0, 0, 0))
                            n_pad += 1
                        tu[0, 0] |= U_GATE | (cc << 16)
                        first = False
                    recs += [tuple(int(v) for v in row) for row in tu]
                    n_c += len(tu)
        if only != "T":                                  # the workgroup's gap + AGG units for ITS tiles (w, w + G, ...: dn_rows_fused_bf16's own rule)
            nw = len(range(w, T, G))
            if nw:
                recs += [(U_NOP, 0, 1, 0)] * AGG_GAP + [(U_AGG | U_LAST, 32 * i, min(32 * (i + 1), nw), 0) for i in range((nw + 31) // 32)]
        out += recs
        ptr.append(len(out))
    arr = np.asarray(out, dtype=np.int64)
    arr = np.where(arr >= 2 ** 31, arr - 2 ** 32, arr).astype(np.int32)          # (chunk << 24 may set the sign bit)
    return dict(unit_ptr=torch.from_numpy(np.asarray(ptr, dtype=np.int32)).to(dev), units=torch.from_numpy(arr).to(dev),
                num_wg=G, num_chunks=nc, rel_wgs=k, stats=dict(transform_units=n_t, closing_units=n_c, chunks=nc, pads=n_pad))
