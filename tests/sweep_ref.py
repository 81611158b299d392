"""Host restatement (numpy, closed forms only) of the L2-blocked "sweep" tile order of the relation-wise transform launches.
Test infrastructure: the device builder (dn_sweep_tables_build_i32) is checked against this, tools/sweep_exp.py uses it for
experiments.  Nothing in the product path imports it.

Rows are relation-major, inside a relation in batch order (key = the node a row belongs to, non-decreasing).  The batch is cut
into `groups` (= 8 XCDs) contiguous key ranges; group x is walked by the W workgroups b = j * groups + x (blocks b, b + 8, ...
share an XCD).  Inside a group the relations' 32-row tiles are laid on a line (relation-major) and the line is cut into W
shares (wg_shares): most workgroups serve ONE relation for the whole launch, a few helpers take the left-overs of several, one
relation after the other (one weight reload per relation, as with contiguous tile ranges).  But instead of a CONTIGUOUS run of
a relation's tiles a workgroup takes every k-th one: relation r's tiles are dealt to its participants (quota a_s) in the order
of the events (s, n) -> time (n + 1/2) / a_s, ties by s.  The workgroups that serve one relation therefore move through the
group's graphs at the same pace, and a source row fetched for one relation is still in the XCD's L2 when the other relations'
workgroups ask for it."""
import numpy as np

TILE = 32


def _rank(quotas, s, n):
    """position of event (s, n) among all events (s', n'), n' < quotas[s'], ordered by ((2n'+1) / (2 quotas[s']), s')."""
    a_s = int(quotas[s])
    r = 0
    for s2, a2 in enumerate(quotas):
        num = (2 * n + 1) * int(a2)
        if s2 < s:
            q = num // a_s                # #{n': (2n'+1) a_s <= num}
        else:
            q = (num - 1) // a_s          # #{n': (2n'+1) a_s <  num}   (s2 == s: the events before n)
        r += (q + 1) // 2
    return r


def group_bounds(rel_ptr, key, N, groups, skip_mask=0):
    """lo[x, r] = first row of relation r that belongs to group x (key >= x N / groups); lo[groups, r] = the relation's end."""
    R = len(rel_ptr) - 1
    lo = np.zeros((groups + 1, R), dtype=np.int64)
    for r in range(R):
        a, b = int(rel_ptr[r]), int(rel_ptr[r + 1])
        for x in range(groups + 1):
            if x == 0:
                lo[x, r] = a
            elif x == groups:
                lo[x, r] = b
            else:
                kx = (x * int(N)) // groups
                lo[x, r] = max(a + int(np.searchsorted(key[a:b], kx, side="left")), lo[x - 1, r])
        if (skip_mask >> r) & 1:
            lo[:, r] = a
    return lo


PURE_MIN_S = 64      # below this many tiles per workgroup no relation gets workgroups of its own (sw_pure_min in dn_index.hip)
HELP_PCT, HELP_SW = 5, 4    # a helper's tile costs 5 % more than a pure workgroup's, a change of relation 4 tiles (sw_help_* there)


def wg_shares(T_row, W, pure_min=PURE_MIN_S):
    """Shares of one group's W workgroups: -> (S, shares) with shares[j] = [(relation, quota), ...] in processing order.
    S = ceil(T / W) tiles per workgroup.  Relation r first gets floor(T_r / S) workgroups of its own ("pure": quota S each, all of
    them in lock step); what is left of every relation (< S tiles each) is laid on a line, relation-major, and cut into balanced
    segments for the remaining workgroups ("helpers": a few relations each, one after the other).
    S < pure_min (small launches: an eighth of BASELINE config 5 has S = 32): NO pure workgroups -- the whole line is cut into W
    segments, a workgroup serves at most two relations when T_r >= S.  With pure workgroups the two or three helpers of a group
    take the left-overs of ALL its relations, a weight reload (~2.7 us) each: 21 us on a launch whose median workgroup needs 39."""
    T_row = [int(v) for v in T_row]
    Tx = sum(T_row)
    S = (Tx + W - 1) // W
    shares = [[] for _ in range(W)]
    if Tx == 0:
        return 0, shares
    pure = [(t // S if S >= pure_min else 0) for t in T_row]
    # Helpers are slower per tile (their share of a relation is spread over the group's whole node range while the pure workgroups
    # are still at its beginning: L2 misses where those hit) and reload weights per relation -- 387-395 us against a median of 351 on
    # BASELINE config 5.  The pure quota therefore grows by delta tiles, so that the helpers' line shrinks by (pure workgroups) x
    # delta: balanced when S + delta = (1 + pct) x (helper tiles) + sw x (relations per helper).
    n_pure, n_rem = sum(pure), sum(1 for t, k in zip(T_row, pure) if t - k * S > 0)
    Wh0 = W - n_pure
    if n_pure > 0 and Wh0 > 0 and n_rem > 0:
        k = -(-n_rem // Wh0)
        S += ((S * HELP_PCT) // 100 + HELP_SW * k) * Wh0 // (Wh0 + n_pure + (n_pure * HELP_PCT) // 100)
        pure = [t // S for t in T_row]
    j = 0
    for r, k in enumerate(pure):
        for _ in range(k):
            shares[j].append((r, S))
            j += 1
    rem = [t - k * S for t, k in zip(T_row, pure)]
    Wh, RT = W - j, sum(rem)
    assert RT <= Wh * S
    if RT:
        c = np.concatenate([[0], np.cumsum(rem)])
        for k in range(Wh):
            p0, p1 = (k * RT) // Wh, ((k + 1) * RT) // Wh
            for r in range(len(rem)):
                a = min(int(c[r + 1]), p1) - max(int(c[r]), p0)
                if a > 0:
                    shares[j + k].append((r, a))
    return S, shares


def sweep_tables(rel_ptr, key, N, W, skip_mask=0, groups=8, xcd_major=False, s_cap=None):
    """-> (table [groups * W, S, 4] int32 {rel, beg, end, 0}, S).  Unused slots are empty tiles {0, 0, 0, 0}."""
    rel_ptr = np.asarray(rel_ptr, dtype=np.int64)
    R = len(rel_ptr) - 1
    lo = group_bounds(rel_ptr, key, N, groups, skip_mask)
    T = np.zeros((groups, R), dtype=np.int64)
    for x in range(groups):
        T[x] = (lo[x + 1] - lo[x] + TILE - 1) // TILE
    all_shares = [wg_shares(T[x], W) for x in range(groups)]
    Sx = np.asarray([sh[0] for sh in all_shares], dtype=np.int64)       # slots the fullest workgroup of a group needs
    S = int(Sx.max()) if s_cap is None else int(s_cap)
    assert S >= int(Sx.max())
    table = np.zeros((groups * W, max(S, 1), 4), dtype=np.int32)
    for x in range(groups):
        _, shares = all_shares[x]
        parts = [[] for _ in range(R)]                       # participants of every relation, in workgroup order
        for j, sh in enumerate(shares):
            for r, a in sh:
                parts[r].append((j, a))
        for r in range(R):
            assert sum(a for _, a in parts[r]) == int(T[x, r])
        for j, sh in enumerate(shares):
            b = (x * W + j) if xcd_major else (j * groups + x)
            slot0 = 0
            for r, a in sh:
                qs = [q for _, q in parts[r]]
                who = [jj for jj, _ in parts[r]].index(j)
                for n in range(a):
                    i = _rank(qs, who, n)
                    beg = int(lo[x, r]) + TILE * i
                    end = min(beg + TILE, int(lo[x + 1, r]))
                    assert table[b, slot0 + n, 2] == 0 and end > beg
                    table[b, slot0 + n] = (r, beg, end, 0)
                slot0 += a
    return table, max(S, 1)


def check_partition(table, rel_ptr, skip_mask=0):
    """every row of every kept relation is covered exactly once; tiles lie inside their relation."""
    t = table.reshape(-1, 4)
    t = t[t[:, 2] > t[:, 1]]
    R = len(rel_ptr) - 1
    cover = np.zeros(int(rel_ptr[-1]) + 1, dtype=np.int64)
    for r, beg, end, _ in t:
        assert rel_ptr[r] <= beg < end <= rel_ptr[r + 1] and end - beg <= TILE
    np.add.at(cover, t[:, 1], 1)
    np.add.at(cover, t[:, 2], -1)
    cov = np.cumsum(cover)[:-1]
    want = np.zeros_like(cov)
    for r in range(R):
        if not (skip_mask >> r) & 1:
            want[int(rel_ptr[r]):int(rel_ptr[r + 1])] = 1
    assert np.array_equal(cov, want)
