// Closing launch of the row-factorised message pass, H = 256 bf16, as a stream of 32-row UNITS through an LDS-DMA ring
// (the structure of dn_rel_ring.hip; replaces rows_selfsum_kernel + overflow_rows_add_kernel at this width).
//
//   out[v, :] = x[v, :] @ W_loop (+ bias)  +  sum_{rows p of v's list} S[p, :]          (rgin.py:137-146 after the fn.sum reduce)
//
// A tile = 32 consecutive nodes.  Its work is cut into units of 32 rows x 512 bytes that all travel the same way, global -> LDS by
// LDS-DMA into a ring of eight 16-KiB stages, issued by four loader waves that do nothing else:
//   * ONE X unit: the tile's 32 rows of x.  The eight compute waves (32 output columns each, their slice of W_loop in 64 VGPRs for
//     the whole launch) multiply it exactly like the ring transform: 16 ds_read_b128 fragment reads, 32 v_mfma_f32_16x16x32_bf16 on
//     the transposed tile (A = weights, B = rows), weight rows interleaved so that a lane ends with 8 consecutive columns of a row.
//   * ceil(c / 32) ENTRY units: the c DISTINCT rows of S that the tile's nodes sum (dn_close_units_build_i32 lists them per tile,
//     each with a 32-bit membership mask: bit i = node i of the tile adds this row; a row shared by many nodes of the tile -- the
//     product of a collapsed "dummy -> u" relation -- is fetched once).  The sum is one more MFMA k-step per unit INTO THE SAME
//     ACCUMULATORS:  acc[column][node] += sum_e S[e][column] * mask[e][node]  with A = the unit's rows read TRANSPOSED from the
//     row-major LDS image (ds_read_b64_tr_b16; the chunk addresses a lane supplies reproduce the interleaved column order) and
//     B = the 0/1 selection matrix built from the masks.  So there is no per-slot vector add, no fixed slot count and no overflow
//     launch (a node with 500 rows is 16 more units), no staging of the products through LDS, one rounding at the end, and the
//     finished rows leave as 16-byte streaming stores straight from the accumulators.
//   * FOLD (the pre-aggregation of a collapsed relation absorbed here, as in rows_selfsum_kernel): the per-(graph, tile) column sums
//     of x are the same transposed read of the X unit against a 0/1 segment indicator.
// LDS image: unpadded rows, 16-byte pieces XOR-swizzled by (row & 15) on the per-lane SOURCE address of the DMA.  The
// ds_read_b128 fragment reads (lane = row & 15 + 16 k-group) are conflict-free as in the ring transform; a transposed read takes
// its four rows 4 apart (rows g + 4 q + 16 jh for lane group g), so the 16 lanes of a group hit 16 distinct 16-byte slots, and the
// two groups that share a 32-lane bank cycle read the two different 8-byte halves (the registers are swapped back with selects).
//
// Numerics: everything that reaches an output element is summed in the fp32 accumulators in a fixed order and rounded once.
// A non-finite element of S makes its column of the whole TILE NaN (0 x Inf inside the selection product), where the slot
// kernel handed it to the nodes that sum that row only: DN_CLOSE_RING=0 restores the slot path for tracing overflows.
#include "dn_common.h"
#include "dn_internal.h"
#include "../../include/dn_hip.h"

#include <algorithm>
#include <cstddef>
#include <cstring>
#include <cstdlib>

#include <rocprim/rocprim.hpp>

namespace {

typedef __bf16 bf16_t;
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef short short8v __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Unit {
    int32_t flags, beg, end, aux;      // X unit: nodes [beg, end), aux = tile;  ENTRY unit: entries [beg, end), aux = the tile's first node
};                                     // AGG unit: the workgroup's tiles number [beg, end) (their segments' aux rows); NOP: nothing
// flags: bit 0 entry unit, bit 1 last unit of its tile (epilogue), bit 2 AGG unit, bit 3 NOP unit, bits 8-15 rows of the tile
constexpr int kUnitEntry = 1, kUnitLast = 2, kUnitAgg = 4, kUnitNop = 8;
constexpr int kUnitXcd = 16;           // on an AGG unit: the workgroup's tiles follow the XCD order below (aux = the number of tiles)
constexpr int kUnitAggAbs = 32;        // on an AGG unit: [beg, end) are the aux rows (= graphs) themselves (order 2)

// Which tile is the n-th of workgroup w (of G)?
//   order 0: w + G n -- the launch walks the batch upwards as ONE front;
//   order 1 (G a multiple of 8): workgroup w = 8 j + x runs on XCD x (round-robin dispatch) and takes the tiles of the x-th EIGHTH of
//   the batch, DOWNWARDS from its end: hi_x - 1 - (j + G/8 n).  The transform launch in front of this one walks the eighth of XCD x
//   upwards (sweep order, dn_sweep_tables_build_i32), so the closing launch starts on the rows the transform wrote and gathered LAST
//   -- still in that XCD's L2 and in the Infinity Cache (measured: transform + closing launch 743-746 -> 728-729 us at config 5).
__host__ __device__ __forceinline__ int32_t cb_eighth(int x, int32_t T) { return (int32_t)(((int64_t)x * T) / 8); }
__device__ __forceinline__ void cb_position(int32_t t, int32_t T, int32_t G, int32_t order, int32_t& w, int32_t& n, bool& last) {
    if (!order) { w = t % G; n = t / G; last = (int64_t)t + G >= T; return; }
    int x = (int)(((int64_t)t * 8) / T);
    while (x < 7 && cb_eighth(x + 1, T) <= t) ++x;
    const int32_t W8 = G >> 3, lo = cb_eighth(x, T), hi = cb_eighth(x + 1, T), r = hi - 1 - t;
    w = 8 * (r % W8) + x; n = r / W8;
    last = r + W8 >= hi - lo;
}
constexpr int kAggGap = 8;             // NOP units between a workgroup's last tile and its first AGG unit (>= the loaders' run-ahead)

// Orders 2 / 3 (round 6: graphs of any size, dn_fold_graph_tiles_multi_build_i32).  The batch is cut into C = K G CHUNKS at graph
// boundaries and a chunk into consecutive 32-node tiles (chunk_tile [C + 1]: first tile of a chunk, chunk_graph [C + 1]: its first
// graph).  The chunks -- not the tiles -- are dealt to the workgroups, K each, by cb_position (order 2: as order 0, order 3: as order
// 1), so a graph's tiles follow each other in ONE stream (the per-graph column sum continues from tile to tile inside the
// workgroup, the AGG unit's read-modify-write of the dummy node's row stays inside the workgroup that stored it) while the launch
// still sweeps the batch as a front of short runs.  A workgroup's AGG units (one per 32 graphs of each of its chunks: the aux rows
// themselves) close its stream behind one gap.
__device__ __forceinline__ int32_t cb2_chunk_of_tile(int32_t t, int32_t G, const int32_t* __restrict__ chunk_tile) {
    int32_t lo = 0, hi = G;                                               // the last c with chunk_tile[c] <= t
    while (lo < hi) {
        const int32_t mid = (lo + hi + 1) >> 1;
        if (chunk_tile[mid] <= t) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}
// the n-th tile (chunk) of workgroup w: the inverse of cb_position
__device__ __forceinline__ int32_t cb_nth(int32_t w, int32_t n, int32_t T, int32_t G, int32_t order) {
    if (!order) return w + G * n;
    const int x = w & 7;
    return cb_eighth(x + 1, T) - 1 - ((w >> 3) + (G >> 3) * n);
}
// gap + AGG units that close workgroup w's stream (0: it has no tiles)
__device__ __forceinline__ int32_t cb2_tail_units(int32_t w, int32_t G, int32_t K, int32_t sub, const int32_t* __restrict__ chunk_tile,
                                                  const int32_t* __restrict__ chunk_graph) {
    int32_t ex = 0;
    bool any = false;
    for (int32_t n = 0; n < K; ++n) {
        const int32_t c = cb_nth(w, n, G * K, G, sub);
        ex += (chunk_graph[c + 1] - chunk_graph[c] + 31) / 32;
        any = any || chunk_tile[c + 1] > chunk_tile[c];
    }
    return any ? kAggGap + ex : 0;
}

// ---------------------------------------------------------------------------------------------------------------- tables
constexpr int kCbWaves = 4;            // tiles per workgroup of the builder (one wavefront each)
constexpr int kCbCap = 256;            // list entries of a tile that are de-duplicated (LDS hash table); larger tiles are listed as is
constexpr int kCbSlots = 512;          // (9 KB of LDS per wavefront: four workgroups of four tiles per CU)
constexpr int kCbHashShift = 23;       // 32 - log2(kCbSlots)
constexpr uint32_t kEmpty = 0xffffffffu;
constexpr int kFoldInfoWords = 12;       // int32 words per tile of a fold table

__device__ __forceinline__ int wave_excl_sum(int v, int lane, int& total) {
    int s = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(s, d, 64);
        if (lane >= d) s += o;
    }
    total = __shfl(s, 63, 64);
    return s - v;
}

// One wavefront per tile.  kept = rows < P (the self-loop rows are not list material here) outside the dropped range.  Entries of
// DIFFERENT nodes with the same row merge into one (mask = OR of the node bits); a row that one node lists twice (parallel edges
// into a fanned-out row; adjacent, the lists are row-ordered) stays a second entry of its own.  A tile's entries are emitted
// SORTED BY ROW: the rows of one (relation, graph) are neighbours in S, so consecutive entries -- the two rows of one LDS-DMA, the
// rows of consecutive DMAs -- are consecutive in memory (runs of ~2 KB instead of single 512-byte rows).
// Common path (<= kCbCap raw list entries): the tile's slice of the lists goes to LDS in one coalesced sweep, every later step is
// lane-per-entry out of LDS: hash-table merge, ballot ranks into a staging list, rank sort.  Larger tiles (or a node that repeats
// a row NOT next to itself) are listed as they are, lane-per-node, in list order.
struct CbLds {
    uint32_t key[kCbSlots], first[kCbSlots], mask[kCbSlots];
    int32_t raw[kCbCap];
    uint32_t srow[kCbCap + 4], smask[kCbCap];
    uint8_t node[kCbCap];              // the node (0 .. 31) whose list holds raw entry i
    int32_t ptr[36];
    int32_t plain;
};

// Per-direction arguments of the two table kernels: blockIdx.y picks the direction (dn_conv_index_build_i32 builds the forward and the
// backward stream of a batch in ONE set of launches; dn_close_units_build_i32 passes one).
struct CbDir {
    const int32_t *tile_ptr, *lptr, *lrows, *drop_enable, *dyn;
    const int32_t *chunk_tile, *chunk_graph;                               // orders 2 / 3: first tile / first graph of every chunk [K G + 1]
    const int32_t* tile_ptr_alt;                                           // dyn[3] == 2: the chunked tables' tile_ptr (see CbAlt)
    int32_t *ent_row, *tile_cnt, *ucnt, *unit_ptr;
    uint32_t* ent_mask;
    const int32_t* uoff;
    Unit* units;
    int32_t P, drop_beg, drop_end;
};
struct CbPair {
    CbDir d[2];
};
// A second table form the SAME launches can build instead, picked per direction by the device word dyn[3] (ril_plan's go: 1 = the
// launch arguments' order over tile_ptr, 2 = this chunked order over tile_ptr_alt): dn_conv_index_build_i32 queues ONE set of
// launches before it knows whether every graph of the batch fits a tile.  order 0: no alternative.
struct CbAlt {
    int32_t order, chunks_per_wg, tile_bound;
};

__global__ __launch_bounds__(kCbWaves * 64) void close_entries_kernel(int32_t N, int32_t T, int32_t G, int32_t Tper, int32_t agg, int32_t order,
                                                                      CbPair pr, CbAlt alt) {
    __shared__ __attribute__((aligned(16))) CbLds Ls[kCbWaves];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = (int)blockIdx.x * kCbWaves + wave;
    CbDir a = pr.d[blockIdx.y];
    if (a.dyn != nullptr) {
        if (a.dyn[3] == 0) return;                                         // go = 0: nothing to do
        if (a.dyn[3] == 2) { order = alt.order; Tper = alt.chunks_per_wg; T = alt.tile_bound; a.tile_ptr = a.tile_ptr_alt; }
    }
    if (t >= T) return;                                                    // (no workgroup barrier below)
    if (order >= 2) {                                                      // chunked: Tper = K chunks per workgroup; T was a bound
        if (agg && t < G && lane == 0) {                                   // (T >= K G >= G) the tail of workgroup t's stream
            const int32_t ex = cb2_tail_units(t, G, Tper, order - 2, a.chunk_tile, a.chunk_graph);
            if (ex) atomicAdd(&a.ucnt[(int64_t)t * Tper + Tper - 1], ex);
        }
        T = a.chunk_tile[G * Tper];
        if (t >= T) return;
    }
    const int32_t* __restrict__ tile_ptr = a.tile_ptr;
    const int32_t* __restrict__ lptr = a.lptr;
    const int32_t* __restrict__ lrows = a.lrows;
    int32_t* __restrict__ ent_row = a.ent_row;
    uint32_t* __restrict__ ent_mask = a.ent_mask;
    int32_t* __restrict__ tile_cnt = a.tile_cnt;
    int32_t* __restrict__ ucnt = a.ucnt;
    int32_t P = a.P, drop_beg = a.drop_beg, drop_end = a.drop_end;
    if (a.dyn != nullptr) {                                                // queued behind the row index (dn_conv_index_build_i32): the
        P = a.dyn[0]; drop_beg = a.dyn[1]; drop_end = a.dyn[2];            // counts the host does not know yet
    }
    if (a.drop_enable != nullptr && *a.drop_enable == 0) drop_beg = drop_end = 0;
    CbLds& L = Ls[wave];
    const int p0 = tile_ptr ? tile_ptr[t] : t * 32;
    const int pend = tile_ptr ? min(tile_ptr[t + 1], p0 + 32) : min(p0 + 32, N), nn = pend - p0;   // (a tile never has more than 32 nodes)
    const int lb = lptr[p0], raw = lptr[pend] - lb;
    auto kept = [&](int r) { return r < P && !(r >= drop_beg && r < drop_end); };
    // units of this tile in the workgroup-major table: one X unit + one per 32 entries; a workgroup's LAST tile also carries the
    // workgroup's NOP gap and AGG units (one per 32 of its pn + 1 tiles)
    int32_t pw = 0, pn = 0;
    bool plast;
    int64_t upos;                                                          // where this tile's unit count goes (the scan's order)
    if (order >= 2) {                                                      // the tile's CHUNK has a position; unit counts add up per chunk
        cb_position(cb2_chunk_of_tile(t, G * Tper, a.chunk_tile), G * Tper, G, order - 2, pw, pn, plast);
        plast = false;                                                     // (the tails were added above)
    } else {
        cb_position(t, T, G, order, pw, pn, plast);
    }
    upos = (int64_t)pw * Tper + pn;
    auto units_of = [&](int c) { return 1 + (c + 31) / 32 + ((agg && plast) ? kAggGap + (pn + 1 + 31) / 32 : 0); };
    auto put_units = [&](int c) {
        if (order >= 2) atomicAdd(&ucnt[upos], units_of(c));
        else ucnt[upos] = units_of(c);
    };
    bool plain = raw > kCbCap;
    if (!plain) {
        const int my0 = lane <= nn ? lptr[p0 + lane] - lb : raw;
        if (lane <= nn) L.ptr[lane] = my0;
        for (int i = lane; i < raw; i += 64) L.raw[i] = lrows[lb + i];
        {
            const int my1 = __shfl_down(my0, 1, 64);                       // lane v < nn: its list is [my0, my1)
            if (lane < nn)
                for (int i = my0; i < my1; ++i) L.node[i] = (uint8_t)lane;
        }
        {                                                                  // key / first = kEmpty, mask = 0: the three arrays lie back to back
            static_assert(offsetof(CbLds, first) == 4 * kCbSlots && offsetof(CbLds, mask) == 8 * kCbSlots && kCbSlots % 4 == 0, "CbLds layout");
            uint4* t4 = reinterpret_cast<uint4*>(L.key);
            for (int i = lane; i < 3 * kCbSlots / 4; i += 64)
                t4[i] = i < 2 * kCbSlots / 4 ? make_uint4(kEmpty, kEmpty, kEmpty, kEmpty) : make_uint4(0u, 0u, 0u, 0u);
        }
        if (lane == 0) L.plain = 0;
        __builtin_amdgcn_wave_barrier();
        auto node_of = [&](int i) { return (int)L.node[i]; };
        auto slot_of = [&](int r) {
            uint32_t h = ((uint32_t)r * 2654435761u) >> kCbHashShift;
            while (L.key[h] != (uint32_t)r) h = (h + 1) & (kCbSlots - 1);
            return h;
        };
        for (int base = 0; base < raw; base += 64) {
            const int i = base + lane;
            if (i >= raw) continue;
            const int r = L.raw[i];
            if (!kept(r)) continue;
            const int v = node_of(i);
            if (i > L.ptr[v] && L.raw[i - 1] == r) continue;               // my node's own repeat: a separate entry, not merged
            uint32_t h = ((uint32_t)r * 2654435761u) >> kCbHashShift;
            for (;;) {
                const uint32_t old = atomicCAS(&L.key[h], kEmpty, (uint32_t)r);
                if (old == kEmpty || old == (uint32_t)r) break;
                h = (h + 1) & (kCbSlots - 1);
            }
            const uint32_t was = atomicOr(&L.mask[h], 1u << v);
            if (was & (1u << v)) L.plain = 1;                              // a repeat that is not adjacent: list the tile as it is
            else atomicMin(&L.first[h], (uint32_t)i);
        }
        __builtin_amdgcn_wave_barrier();
        plain = L.plain != 0;
        if (!plain) {
            int n = 0;                                                     // emitted entries so far (wave-uniform)
            bool repeats = false;                                          // some row has more than one entry (wave-uniform)
            for (int base = 0; base < raw; base += 64) {
                const int i = base + lane;
                bool emit = false, own = false;
                int r = 0;
                uint32_t m = 0;
                if (i < raw) {
                    r = L.raw[i];
                    if (kept(r)) {
                        const int v = node_of(i);
                        if (i > L.ptr[v] && L.raw[i - 1] == r) { emit = true; own = true; m = 1u << v; }
                        else {
                            const uint32_t h = slot_of(r);
                            if (L.first[h] == (uint32_t)i) { emit = true; m = L.mask[h]; }
                        }
                    }
                }
                repeats = repeats || __ballot(own) != 0ull;
                const unsigned long long b = __ballot(emit);
                if (emit) {
                    const int at = n + __popcll(b & ((1ull << lane) - 1ull));
                    L.srow[at] = (uint32_t)r;
                    L.smask[at] = m;
                }
                n += __popcll(b);
            }
            if (lane < 4) L.srow[n + lane] = 0xffffffffu;                  // sentinels: the rank loop reads four rows at a time
            __builtin_amdgcn_wave_barrier();
            // rank = entries with a lower row (+ the equal ones before me: only a row listed twice by one node has any).  The
            // sentinels (0xffffffff) are never lower.
            for (int i = lane; i < n; i += 64) {
                const uint32_t ri = L.srow[i];
                int rank = 0;
                for (int j = 0; j < n; j += 4) {
                    const uint4 q = *reinterpret_cast<const uint4*>(&L.srow[j]);
                    rank += (int)(q.x < ri) + (int)(q.y < ri) + (int)(q.z < ri) + (int)(q.w < ri);
                }
                if (repeats) {
                    for (int j = 0; j < n; j += 4) {
                        const uint4 q = *reinterpret_cast<const uint4*>(&L.srow[j]);
                        rank += (int)(q.x == ri && j < i) + (int)(q.y == ri && j + 1 < i) + (int)(q.z == ri && j + 2 < i) +
                                (int)(q.w == ri && j + 3 < i);
                    }
                }
                ent_row[lb + rank] = (int32_t)ri;
                ent_mask[lb + rank] = L.smask[i];
            }
            if (lane == 0) { tile_cnt[t] = n; put_units(n); }
            return;
        }
    }
    // listed as it is: lane v = node p0 + v walks its own list, every kept row an entry of its own
    const bool node = lane < nn;
    const int my_b = node ? lptr[p0 + lane] : 0, my_e = node ? lptr[p0 + lane + 1] : 0;
    int mine = 0;
    for (int i = my_b; i < my_e; ++i) mine += kept(lrows[i]) ? 1 : 0;
    int total;
    int at = lb + wave_excl_sum(mine, lane, total);
    for (int i = my_b; i < my_e; ++i) {
        const int r = lrows[i];
        if (!kept(r)) continue;
        ent_row[at] = r; ent_mask[at] = 1u << lane; ++at;
    }
    if (lane == 0) { tile_cnt[t] = total; put_units(total); }
}

// Unit offsets in WORKGROUP-MAJOR order: workgroup w of G takes the tiles w, w + G, ... (round robin: the launch sweeps the nodes
// as one stream; order 1: see cb_position); position k' = w * Tper + n holds the n-th tile of workgroup w.  The entries kernel leaves each tile's unit count there, one
// exclusive scan (rocPRIM) gives the offsets.
__global__ void close_fill_kernel(int32_t N, int32_t T, int32_t G, int32_t Tper, int32_t agg, int32_t order, CbPair pr, CbAlt alt) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    CbDir a = pr.d[blockIdx.y];
    if (a.dyn != nullptr) {
        if (a.dyn[3] == 0) return;
        if (a.dyn[3] == 2) { order = alt.order; Tper = alt.chunks_per_wg; T = alt.tile_bound; a.tile_ptr = a.tile_ptr_alt; }
    }
    const int32_t* __restrict__ tile_ptr = a.tile_ptr;
    const int32_t* __restrict__ lptr = a.lptr;
    const int32_t* __restrict__ tile_cnt = a.tile_cnt;
    const int32_t* __restrict__ uoff = a.uoff;
    Unit* __restrict__ units = a.units;
    const int32_t base = uoff[0];                                          // (both directions share one scan: the second starts at the first's total)
    int32_t w = 0, n = 0;
    bool plast;
    int64_t k;
    int32_t inner = 0;                                                     // units of the earlier tiles of my chunk (chunked orders)
    if (order >= 2) {
        const int32_t K = Tper, C = G * K, sub = order - 2;
        T = a.chunk_tile[C];                                               // (the argument was a bound)
        if (t <= G) a.unit_ptr[t] = uoff[t * K] - base;
        if (agg && t < G) {                                                // the tail of workgroup t: the gap, then its chunks' AGG units
            const int32_t ex = cb2_tail_units((int32_t)t, G, K, sub, a.chunk_tile, a.chunk_graph);
            if (ex) {
                Unit* q = units + (uoff[(t + 1) * K] - base - ex);
                for (int i = 0; i < kAggGap; ++i) q[i] = Unit{kUnitNop, 0, 1, 0};
                q += kAggGap;
                for (int32_t nn = 0; nn < K; ++nn) {
                    const int32_t c = cb_nth((int32_t)t, nn, C, G, sub);
                    for (int32_t g0 = a.chunk_graph[c]; g0 < a.chunk_graph[c + 1]; g0 += 32)
                        *q++ = Unit{kUnitAgg | kUnitLast | kUnitAggAbs, g0, min(g0 + 32, a.chunk_graph[c + 1]), 0};
                }
            }
        }
        if (t >= T) return;
        const int32_t c = cb2_chunk_of_tile((int32_t)t, C, a.chunk_tile);
        cb_position(c, C, G, sub, w, n, plast);
        plast = false;
        for (int32_t tt = a.chunk_tile[c]; tt < (int32_t)t; ++tt) inner += 1 + (tile_cnt[tt] + 31) / 32;
        k = (int64_t)w * K + n;
    } else {
        if (t <= G) a.unit_ptr[t] = uoff[t * Tper] - base;                 // (position G * Tper holds the total)
        if (t >= T) return;
        cb_position((int32_t)t, T, G, order, w, n, plast);
        k = (int64_t)w * Tper + n;
    }
    const int32_t p0 = tile_ptr ? tile_ptr[t] : (int32_t)t * 32;
    const int32_t pend = tile_ptr ? min(tile_ptr[t + 1], p0 + 32) : min(p0 + 32, N);
    const int32_t c = tile_cnt[t], e0 = lptr[p0], rows = (pend - p0) << 8;
    Unit* u = units + (uoff[k] - base + inner);
    const int ne = (c + 31) / 32;
    u[0] = Unit{(ne == 0 ? kUnitLast : 0) | rows, p0, pend, (int32_t)t};
    for (int i = 0; i < ne; ++i)
        u[1 + i] = Unit{kUnitEntry | (i == ne - 1 ? kUnitLast : 0) | rows, e0 + 32 * i, e0 + min(32 * (i + 1), c), p0};
    if (agg && plast) {                                                    // the workgroup's last tile: its NOP gap and AGG units
        const int32_t nw = (int32_t)n + 1;
        Unit* q = u + 1 + ne;
        for (int i = 0; i < kAggGap; ++i) q[i] = Unit{kUnitNop, 0, 1, 0};
        q += kAggGap;
        for (int i = 0; 32 * i < nw; ++i) q[i] = Unit{kUnitAgg | kUnitLast | (order ? kUnitXcd : 0), 32 * i, min(32 * (i + 1), nw), order ? T : 0};
    }
}

// Tiles = the graphs of a batch (segment-complete tiles for the absorbed fold): segment j = the nodes seg_nodes[seg_ptr[j] ..
// seg_ptr[j+1]) (a graph's nodes that feed its dummy node).  Block j = [first node of segment j (0 for j = 0), first node of
// segment j + 1 (N for the last)).  Valid when every segment is a non-empty contiguous ascending run, the segments ascend,
// every block has at most 32 nodes AND the row the segment's product is added to (add_idx[j]: the dummy node) lies inside
// block j -- the AGG unit of tile j is a read-modify-write of that row by the workgroup that STORED tile j, so a target in
// another block (dummy node first, all dummy nodes at the end of the batch) would race with that block's owner.  Then
// tile j = block j, fold record j = {local ids: 0 inside the segment, 255 outside; first aux row = j; count = 1}.
__global__ void fold_graph_tiles_kernel(int32_t N, int32_t S, const int32_t* __restrict__ sptr, const int32_t* __restrict__ snodes,
                                        const int32_t* __restrict__ add_idx, int32_t* __restrict__ tile_ptr,
                                        int32_t* __restrict__ info, int32_t* __restrict__ ok) {
    dn_fold_graph_tile_one((int64_t)blockIdx.x * blockDim.x + threadIdx.x, N, S, sptr, snodes, add_idx, tile_ptr, info, ok);
}

struct FmPair {                        // the directions of one batch share the launches (blockIdx.y)
    dn_internal::FoldMultiDir d[2];
};
// gate (may be NULL): a device word that must be 2 (ril_plan's "chunked tiles") for the tables to be wanted at all.
__global__ void fold_multi_valid_kernel(int32_t N, int32_t S, FmPair pr) {
    const dn_internal::FoldMultiDir& a = pr.d[blockIdx.y];
    if (a.gate != nullptr && *a.gate != 2) return;
    dn_fold_multi_valid_one((int64_t)blockIdx.x * blockDim.x + threadIdx.x, N, S, a.seg_ptr, a.seg_nodes, a.add_idx, a.dev_ok);
}
// ONE workgroup per direction: chunk c = the graphs whose blocks start in [ceil(c N / C), ceil((c + 1) N / C)); tiles per chunk, their
// prefix sums.
constexpr int kChunkThreads = 1024, kChunkMax = 16384;
__global__ __launch_bounds__(kChunkThreads) void fold_multi_chunks_kernel(int32_t N, int32_t S, int32_t C, FmPair pr) {
    __shared__ int32_t cg[kChunkMax + 1];
    __shared__ int32_t wsum[kChunkThreads / 64];
    const dn_internal::FoldMultiDir& a = pr.d[blockIdx.x];
    if (*a.dev_ok == 0 || (a.gate != nullptr && *a.gate != 2)) return;    // (an invalid batch's tables are never read)
    const int32_t* __restrict__ sptr = a.seg_ptr;
    const int32_t* __restrict__ snodes = a.seg_nodes;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c <= C; c += kChunkThreads)
        cg[c] = c == C ? S : dn_fold_first_graph_from((int32_t)(((int64_t)c * N + C - 1) / C), N, S, sptr, snodes);
    __syncthreads();
    constexpr int K = kChunkMax / kChunkThreads;                          // consecutive chunks per thread
    int32_t nt[K], mine = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int c = tid * K + k;
        nt[k] = 0;
        if (c < C) {
            const int32_t n_lo = dn_fold_gstart(cg[c], N, S, sptr, snodes), n_hi = dn_fold_gstart(cg[c + 1], N, S, sptr, snodes);
            nt[k] = (n_hi - n_lo + 31) / 32;
        }
        mine += nt[k];
    }
    int32_t run = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int32_t o = __shfl_up(run, d, 64);
        if (lane >= d) run += o;
    }
    if (lane == 63) wsum[wave] = run;
    __syncthreads();
    int32_t carry = 0;
    for (int w = 0; w < wave; ++w) carry += wsum[w];
    run += carry - mine;                                                   // tiles of the chunks before mine
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int c = tid * K + k;
        if (c <= C) { a.chunk_tile[c] = run; a.chunk_graph[c] = cg[c]; }
        run += nt[k];
    }
}
__global__ void fold_multi_tiles_kernel(int32_t N, int32_t S, int32_t C, FmPair pr) {
    const dn_internal::FoldMultiDir& a = pr.d[blockIdx.y];
    if (*a.dev_ok == 0 || (a.gate != nullptr && *a.gate != 2)) return;
    dn_fold_multi_tile_one((int64_t)blockIdx.x * blockDim.x + threadIdx.x, N, S, C, a.seg_ptr, a.seg_nodes, a.chunk_tile, a.chunk_graph,
                           a.tile_ptr, a.fold_info);
}

// ---------------------------------------------------------------------------------------------------------------- kernel
constexpr int kH = 256;
constexpr int kRowB = 2 * kH;          // bytes per row
constexpr int kTR = 32;                // rows per unit
constexpr int kStageB = kTR * kRowB;   // 16 KiB
constexpr int kNS = 8;                 // ring stages (128 KiB)
constexpr int kCompute = 8, kLoaders = 4;
constexpr int kThreads = 64 * (kCompute + kLoaders);
constexpr int kRowsPerLoader = kTR / kLoaders;      // 8
constexpr int kDmaPerTile = kRowsPerLoader / 2;     // 4 DMA wave-instructions per loader and unit (2 rows each)
constexpr int kBatch = 8;                           // units per batch of records / indices / masks
constexpr int kRecRing = 32, kIdxRing = 16, kDescRing = 32, kMaskRing = 32, kFoldRing = 32;
constexpr int kFoldInfo = 12;                       // int32 words per tile of the fold table (dn_fold_tables_build_i32)

__device__ int32_t g_close_zero[64];                // zeros (device globals are zero-initialised): the mask of a row past a unit's end
#ifdef DN_CLOSE_TIMES
// diagnostic build (-DDN_CLOSE_TIMES): wall ticks (100 MHz) of compute wave 0 of every workgroup of the last closing launch
// {first tick, last tick, units} -- tools/close_exp.py --spread
__device__ unsigned long long g_close_times[256][3];
#endif

__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
    typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v;
    v[0] = (bf16_t)a;
    v[1] = (bf16_t)b;
    return __builtin_bit_cast(uint32_t, v);
}

#define DN_DS_READ128(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))

// FOLD: 0 none; 1 the per-(segment, tile) column sums of x leave as fp32 partial rows (seg_part; dn_fold_tail_bf16 finishes);
// 2 every segment lies inside one tile: its column sum leaves as the bf16 aux row, and the workgroup's AGG units at the end of
// its stream multiply those rows by W_agg and add each product to its output row (agg_idx) -- no partial rows, no tail launch.
template <int FOLD>
__global__ __launch_bounds__(kThreads) void rows_close_ring_kernel(
    const bf16_t* __restrict__ X, const bf16_t* __restrict__ W, int32_t w_kn, const bf16_t* __restrict__ bias,
    const bf16_t* __restrict__ S, const Unit* __restrict__ units, const int32_t* __restrict__ unit_ptr,
    const int32_t* __restrict__ ent_row, const uint32_t* __restrict__ ent_mask, int32_t N, int32_t flags, bf16_t* __restrict__ out,
    const int32_t* __restrict__ fold_info, float* __restrict__ seg_part, const bf16_t* __restrict__ W_agg,
    bf16_t* __restrict__ aux, const int32_t* __restrict__ agg_idx) {
    __shared__ __attribute__((aligned(1024))) char lds[kNS * kStageB];
    __shared__ __attribute__((aligned(1024))) int32_t foldR[kFoldRing][16];                   // fold records of the X units (12 words used)
    __shared__ __attribute__((aligned(256))) uint32_t maskR[kMaskRing][32];                   // membership masks of the entry units, k order
    __shared__ __attribute__((aligned(16))) int32_t descL[kDescRing][4];                      // unit records for the compute waves
    __shared__ __attribute__((aligned(128))) int32_t recR[kLoaders][kRecRing][4];             // loader-private rings: unit records
    __shared__ __attribute__((aligned(256))) int32_t idxR[kLoaders][kIdxRing][kRowsPerLoader]; // ... and source rows
    __shared__ __attribute__((aligned(16))) char wscr[kCompute][2048];                         // transposition scratch of a [k][n] W
    __shared__ __attribute__((aligned(16))) float segL[FOLD == 2 ? kCompute : 1][4][8];       // running column sums of a multi-tile graph
    typedef __attribute__((address_space(3))) char* lds_wp;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_wp)lds;
    const unsigned desc_base = (unsigned)(uintptr_t)(lds_wp)&descL[0][0];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = (int)blockIdx.x, nwg = (int)gridDim.x;
    const int u_beg = unit_ptr[wg];
    const int nt = unit_ptr[wg + 1] - u_beg;
    if (nt <= 0) return;
    units += u_beg;

    if (wave >= kCompute) {
        // ------------------------------------------------------------------------------------------------ loaders
        // (the protocol of dn_rel_ring.hip: records two batches ahead, source rows and masks one batch ahead, rows kNS - 2 units
        // ahead; one rendezvous per pair of units; the counted wait only ever gets stricter by the extra DMAs of a batch)
        static_assert(kBatch == 8 && kNS - 1 <= kBatch && 3 * kBatch <= kRecRing && 2 * kBatch <= kIdxRing &&
                      3 * kBatch <= kDescRing && 3 * kBatch <= kMaskRing && 3 * kBatch <= kFoldRing &&
                      kDmaPerTile * (kNS - 2) < 64 && kNS >= 6 && (kBatch & 1) == 0, "ring sizes");
        const int q = wave - kCompute;
        const int rin = lane >> 5, pos = lane & 31;
        int swoff[kDmaPerTile];
#pragma unroll
        for (int j = 0; j < kDmaPerTile; ++j) {
            const int rl = kRowsPerLoader * q + 2 * j + rin;               // row of the stage this lane fills
            swoff[j] = (pos ^ (rl & 15)) * 16;                             // source byte offset inside the row
        }
        const unsigned rec_base = (unsigned)(uintptr_t)(lds_wp)&recR[q][0][0];
        const unsigned idx_base = (unsigned)(uintptr_t)(lds_wp)&idxR[q][0][0];
        const unsigned mask_base = (unsigned)(uintptr_t)(lds_wp)&maskR[0][0];
        const unsigned fold_base = (unsigned)(uintptr_t)(lds_wp)&foldR[0][0];
        const int myrow = 2 * (lane & 3) + ((lane >> 2) & 1);              // the index ring holds a unit's 8 rows as {0,2,4,6,1,3,5,7}
        // source bases as one base + differences: a three-way select of captured variables makes hipcc keep the closure on the stack
        const uint64_t baseX = (uint64_t)(uintptr_t)X, dS = (uint64_t)(uintptr_t)S - baseX, dA = (uint64_t)(uintptr_t)aux - baseX;
        auto dma_recs = [&](int T0) __attribute__((always_inline)) {                                      // records of units T0 .. T0 + 7 (clamped: valid memory)
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(rec_base + (unsigned)(T0 % kRecRing) * 16u));
            if (lane < kBatch) glds16(units + min(T0 + lane, nt - 1), dst);
        };
        auto stage_idx = [&](int T0) __attribute__((always_inline)) {                                     // source rows of my 8 rows of units T0 .. T0 + 7; records -> descL
            const int T = T0 + (lane >> 3);
            const int32_t* rp = &recR[q][T % kRecRing][0];
            const int fl = rp[0], beg = rp[1], end = rp[2];
            const int pc = end > beg ? min(beg + kRowsPerLoader * q + myrow, end - 1) : 0;   // past the end: the last row again
            if (q == 0 && lane < 4 * kBatch)
                descL[(T0 + (lane >> 2)) % kDescRing][lane & 3] = recR[0][(T0 + (lane >> 2)) % kRecRing][lane & 3];
            const unsigned dst =
                (unsigned)__builtin_amdgcn_readfirstlane((int)(idx_base + (unsigned)(T0 % kIdxRing) * (4u * kRowsPerLoader)));
            if (fl & kUnitEntry) glds4(ent_row + pc, dst);                 // lane l lands at + 4 l: [unit][8 rows]
            else if (FOLD == 2 && (fl & kUnitAgg))                         // the aux row of my pc-th tile (= its segment)
                idxR[q][T % kIdxRing][lane & 7] =
                    (fl & kUnitAggAbs) ? pc
                    : (fl & kUnitXcd) ? (int)(((int64_t)((wg & 7) + 1) * rp[3]) >> 3) - 1 - (wg >> 3) - (nwg >> 3) * pc : wg + nwg * pc;
            else idxR[q][T % kIdxRing][lane & 7] = (fl & kUnitNop) ? 0 : pc;   // an X unit's rows are its nodes
            // membership masks of units T0 + 2 q, T0 + 2 q + 1, in the k order of the transposed reads:
            // position k = 8 g + 4 jh + qq  <->  row g + 16 jh + 4 qq of the unit
            {
                const int Tm = T0 + 2 * q + (lane >> 5), kk = lane & 31;
                const int r = (kk >> 3) + 16 * ((kk >> 2) & 1) + 4 * (kk & 3);
                const int32_t* mp = &recR[q][Tm % kRecRing][0];
                const int e = mp[1] + r;
                const bool ok = (mp[0] & kUnitEntry) && e < mp[2];
                const void* msrc = ok ? (const void*)(ent_mask + e) : (const void*)(g_close_zero + kk);
                if constexpr (FOLD == 2) {                                 // an AGG unit's "masks" are the output rows of its 32 products
                    const int ord = mp[1] + kk;
                    if ((mp[0] & kUnitAgg) && ord < mp[2])
                        msrc = agg_idx + ((mp[0] & kUnitAggAbs) ? ord
                                          : (mp[0] & kUnitXcd) ? (int)(((int64_t)((wg & 7) + 1) * mp[3]) >> 3) - 1 - (wg >> 3) - (nwg >> 3) * ord
                                                               : wg + nwg * ord);
                }
                const unsigned mdst =
                    (unsigned)__builtin_amdgcn_readfirstlane((int)(mask_base + (unsigned)((T0 + 2 * q) % kMaskRing) * 128u));
                glds4(msrc, mdst);
            }
            if constexpr (FOLD != 0) {
                if (q == 1) {                                              // fold records of the batch's X units: lane 4 k + c = 16-byte chunk c
                    const int Tf = T0 + (lane >> 2), c = lane & 3;
                    const int32_t* fp = &recR[1][Tf % kRecRing][0];
                    const unsigned fdst =
                        (unsigned)__builtin_amdgcn_readfirstlane((int)(fold_base + (unsigned)(T0 % kFoldRing) * 64u));
                    if (lane < 4 * kBatch && c < 3 && !(fp[0] & (kUnitEntry | kUnitAgg | kUnitNop)))
                        glds16(fold_info + (size_t)fp[3] * kFoldInfo + 4 * c, fdst);
                }
            }
        };
        const char* srcA[kDmaPerTile];
        const char* srcB[kDmaPerTile];
        auto prep = [&](int u, const char* (&src)[kDmaPerTile]) __attribute__((always_inline)) {
            typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
            const i32x4 iv = *reinterpret_cast<const i32x4*>(&idxR[q][u % kIdxRing][4 * rin]);   // rows rin, 2 + rin, 4 + rin, 6 + rin
            const int kind = recR[q][u % kRecRing][0];
            const bool ent = (kind & kUnitEntry) != 0, agg = FOLD == 2 && (kind & kUnitAgg) != 0;
            // (selects, not branches: a branch here keeps hipcc from holding the source addresses in registers)
            const uint64_t base0 = baseX + (ent ? dS : 0ull) + (agg ? dA : 0ull);      // (X, S or aux)
#ifdef DN_TUNING_ENV
            const uint32_t rmask = (flags & (ent ? 4 : 8)) ? 1023u : ((ent && (flags & 128)) ? 0x1ffffu : 0xffffffffu);   // (ablation: the rows come from L2 / a 64 MB window)
#else
            constexpr uint32_t rmask = 0xffffffffu;
#endif
#pragma unroll
            for (int j = 0; j < kDmaPerTile; ++j)
                src[j] = reinterpret_cast<const char*>(base0 + (uint64_t)((uint32_t)iv[j] & rmask) * kRowB + (uint64_t)swoff[j]);
        };
        auto rows = [&](int u, const char* (&src)[kDmaPerTile]) __attribute__((always_inline)) {
            const unsigned st = (unsigned)__builtin_amdgcn_readfirstlane(
                (int)(lds_base + (unsigned)(u % kNS) * kStageB + (unsigned)(kRowsPerLoader * q) * kRowB));
#pragma unroll
            for (int j = 0; j < kDmaPerTile; ++j) glds16(src[j], st + (unsigned)(2 * j) * kRowB);   // lane l lands at + 16 l
        };
        auto batch = [&](int u) __attribute__((always_inline)) {                                          // u = 8 b: rows / masks of batch b + 1, records of batch b + 2
            if ((u & (kBatch - 1)) == 0) {                                 // wave-uniform
                stage_idx(u + kBatch);
                dma_recs(u + 2 * kBatch);
            }
        };
        dma_recs(0);
        dma_recs(kBatch);
        wait_vmcnt<0>();
        stage_idx(0);
        wait_vmcnt<0>();
#pragma unroll 1
        for (int u = 0; u < kNS - 2; ++u) {
            batch(u);
            prep(u, srcA);
            rows(u, srcA);
        }
        batch(kNS - 2);
        prep(kNS - 2, srcA);
        batch(kNS - 1);
        prep(kNS - 1, srcB);
        wait_vmcnt<kDmaPerTile*(kNS - 3)>();                               // unit 0 has landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll 1
        for (int t = 0; t < nt; t += 2) {
#ifdef DN_TUNING_ENV
            if (flags & 64) wait_vmcnt<kDmaPerTile*(kNS - 4)>();           // (experiment: landed up to unit t + 1 only)
            else
#endif
            wait_vmcnt<kDmaPerTile*(kNS - 5)>();                           // issued: up to unit t + 5; landed: up to unit t + 2
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // ... and the records I copied for the compute waves
            __builtin_amdgcn_s_barrier();                                  // everyone's have; stages of units t-2, t-1 are free
            rows(t + kNS - 2, srcA);
            rows(t + kNS - 1, srcB);
            batch(t + kNS);
            prep(t + kNS, srcA);
            prep(t + kNS + 1, srcB);
        }
        wait_vmcnt<0>();                                                   // nothing may land after the LDS is given back
        return;
    }

    // ---------------------------------------------------------------------------------------------------- compute
#ifdef DN_CLOSE_TIMES
    struct WgTimes {
        unsigned long long t0;
        int wave, lane, nt;
        __device__ ~WgTimes() {
            if (wave == 0 && lane == 0 && blockIdx.x < 256) {
                g_close_times[blockIdx.x][0] = t0; g_close_times[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
                g_close_times[blockIdx.x][2] = (unsigned long long)nt;
            }
        }
    } wg_times{__builtin_amdgcn_s_memrealtime(), wave, lane, nt};
#endif
    const bool nt_store = (flags & 2) != 0;
    const int n0 = 32 * wave;
    const int j = lane & 15, g = lane >> 4;
    // fragment of k-step ks inside stage 0: row j (+ 16 m), piece (4 ks + g) ^ j  (dn_rel_ring.hip)
    const unsigned off0 = lds_base + (unsigned)(j * kRowB + ((g ^ j) << 4));
    const int colA0 = 8 * (j >> 2) + (j & 3);                              // output column (minus n0) of A row j, MFMA tile 0
    const size_t ocol = (size_t)(n0 + 8 * g);
    // transposed reads: lane (group g, q4 = j >> 2, p4 = j & 3) supplies row g + 4 q4 (+ 16 jh), logical piece 4 wave + p4, 8-byte half
    // nn; read t = 2 jh + s takes half s ^ (g & 1)
    const int q4 = j >> 2, p4 = j & 3, rt = g + 4 * q4;
    const unsigned tr0 = (unsigned)(rt * kRowB + (((4 * wave + p4) ^ rt) << 4) + 8 * (g & 1));     // (relative to the stage)
    const bool odd = (g & 1) != 0;
    bf16x8 wf[8][2];
    if (w_kn == 0) {                                                       // W given as [out][in] (the caller's transposed copy)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int n = 0; n < 2; ++n)
                wf[ks][n] = *reinterpret_cast<const bf16x8*>(W + (size_t)(n0 + colA0 + 4 * n) * kH + ks * 32 + 8 * g);
    } else {                                                               // W as the parameter stores it, [in][out]: once per launch
        dn_load_w_kn32<8>(W, kH, n0, lane, wscr[wave], wf);
    }
    u32x4 bv = {0u, 0u, 0u, 0u};                                           // bias of my 8 columns (bf16 x 8)
    if (bias) bv = *reinterpret_cast<const u32x4*>(bias + n0 + 8 * g);
    f32x4 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    typedef short4v __attribute__((address_space(3))) * lds_tr;
    // A operand of a transposed product over the 32 rows of a stage: element jj of a[n] = row g + 16 (jj >> 2) + 4 (jj & 3),
    // column n0 + 8 (j >> 2) + 4 n + (j & 3)
    auto tr_frags = [&](unsigned sb, bf16x8 (&a)[2]) __attribute__((always_inline)) {
        const unsigned b0 = sb + tr0;
        const short4v r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(uintptr_t)b0);
        const short4v r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(uintptr_t)(b0 ^ 8u));
        const short4v r2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(uintptr_t)(b0 + 16u * kRowB));
        const short4v r3 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(uintptr_t)((b0 ^ 8u) + 16u * kRowB));
        const short4v lo0 = odd ? r1 : r0, hi0 = odd ? r3 : r2;           // half 0 of the pieces: columns + 0..3
        const short4v lo1 = odd ? r0 : r1, hi1 = odd ? r2 : r3;           // half 1: columns + 4..7
        const short8v f0 = {lo0[0], lo0[1], lo0[2], lo0[3], hi0[0], hi0[1], hi0[2], hi0[3]};
        const short8v f1 = {lo1[0], lo1[1], lo1[2], lo1[3], hi1[0], hi1[1], hi1[2], hi1[3]};
        a[0] = __builtin_bit_cast(bf16x8, f0);
        a[1] = __builtin_bit_cast(bf16x8, f1);
    };

    // rows p0 + j and p0 + 16 + j of the tile, my 8 columns: bias, bf16, one 16-byte store each
    auto epilogue = [&](int32_t p0, int32_t pend) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int p = p0 + j + 16 * m;
            float v[8];
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) v[4 * n + i] = acc[m][n][i];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[2 * i] += __uint_as_float(bv[i] << 16);
                v[2 * i + 1] += __uint_as_float(bv[i] & 0xffff0000u);
            }
            u32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
#ifdef DN_TUNING_ENV
            if (p < pend && !((flags & 16) && p != 0)) {                   // (flags & 16: ablation, no stores)
#else
            if (p < pend) {
#endif
                u32x4* dst = reinterpret_cast<u32x4*>(out + (size_t)p * kH + ocol);
                if (nt_store) __builtin_nontemporal_store(o, dst);
                else *dst = o;
            }
        }
    };

    // AGG unit: row r of the unit is the product of one segment's aux row with W_agg; it is ADDED to output row tgt[r] (this
    // workgroup stored that row itself, units ago, and drained its stores at the NOP gap in between).  The rows' current values
    // are requested before the unit's MFMAs (agg_fetch) and added behind them (epilogue_agg).
    u32x4 agg_old[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    u32x4* agg_dst[2] = {nullptr, nullptr};
    auto agg_fetch = [&](const uint32_t* tgt, int32_t cnt) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int r = j + 16 * m;
            agg_dst[m] = reinterpret_cast<u32x4*>(out + (size_t)tgt[r < cnt ? r : 0] * kH + ocol);
            if (r < cnt) agg_old[m] = *agg_dst[m];
        }
    };
    auto epilogue_agg = [&](int32_t cnt) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int r = j + 16 * m;
            if (r < cnt) {
                const u32x4 old = agg_old[m];
                u32x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int n = i >> 1, e = 2 * (i & 1);
                    o[i] = pack_bf16x2(acc[m][n][e] + __uint_as_float(old[i] << 16), acc[m][n][e + 1] + __uint_as_float(old[i] & 0xffff0000u));
                }
                *agg_dst[m] = o;
            }
        }
    };

#define DN_FETCH(KS, SB)                                                                                              \
    {                                                                                                                 \
        const unsigned a_ = ((SB) + off0) ^ (unsigned)(((KS) & 3) << 6);                                              \
        if ((KS) < 4) {                                                                                               \
            DN_DS_READ128(xf[KS][0], a_, 0);                                                                          \
            DN_DS_READ128(xf[KS][1], a_, 8192); /* rows 16..31 of the stage */                                        \
        } else {                                                                                                      \
            DN_DS_READ128(xf[KS][0], a_, 256);                                                                        \
            DN_DS_READ128(xf[KS][1], a_, 8448);                                                                       \
        }                                                                                                             \
    }
#define DN_KSTEP(KS, CNT) /* behind this k-step's two reads in the queue: 2 (7 - KS) reads + the next record */       \
    asm volatile("s_waitcnt lgkmcnt(" #CNT ")" : "+v"(xf[KS][0]), "+v"(xf[KS][1]));                                    \
    _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                                      \
    _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                                      \
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[KS][n], xf[KS][m], acc[m][n], 0, 0, 0);              \
    __builtin_amdgcn_sched_barrier(0);

    u32x4 dn;                                                              // record of the next unit
    __builtin_amdgcn_s_barrier();                                          // unit 0 has landed
    {
        const unsigned a0 = desc_base;
        asm volatile("ds_read_b128 %0, %1" : "=v"(dn) : "v"(a0));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dn) : : "memory");
    int32_t u_fl = __builtin_amdgcn_readfirstlane((int)dn[0]);
    int32_t u_beg2 = __builtin_amdgcn_readfirstlane((int)dn[1]);
    int32_t u_end = __builtin_amdgcn_readfirstlane((int)dn[2]);
    int32_t u_aux = __builtin_amdgcn_readfirstlane((int)dn[3]);

    auto unit = [&](int u) __attribute__((always_inline)) {
        if ((u & 1) == 0) __builtin_amdgcn_s_barrier();                    // one per pair of units: units up to u + 2 have landed
        const unsigned an = desc_base + (unsigned)((u + 1) % kDescRing) * 16u;
        const unsigned sb = (unsigned)(u % kNS) * kStageB;
        int32_t p0 = 0;
        if (u_fl & kUnitNop) {
            // ---- gap between the workgroup's tiles and its AGG units: my stores of aux rows and output rows have left
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("ds_read_b128 %0, %1" : "=v"(dn) : "v"(an));
        } else if (!(u_fl & kUnitEntry)) {
            // ---- X unit: acc = W_loop^T-slice x rows^T      (AGG unit: the same product on 32 aux rows with W_agg)
            p0 = u_beg2;
            bf16x8 xf[8][2];
            DN_FETCH(0, sb) DN_FETCH(1, sb) DN_FETCH(2, sb) DN_FETCH(3, sb) DN_FETCH(4, sb) DN_FETCH(5, sb) DN_FETCH(6, sb) DN_FETCH(7, sb)
            asm volatile("ds_read_b128 %0, %1" : "=v"(dn) : "v"(an));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
            DN_KSTEP(0, 15) DN_KSTEP(1, 13) DN_KSTEP(2, 11) DN_KSTEP(3, 9) DN_KSTEP(4, 7) DN_KSTEP(5, 5) DN_KSTEP(6, 3) DN_KSTEP(7, 1)
            if constexpr (FOLD != 0) {
                // column sums of the tile's x rows per segment (graph): D[column][s] = sum_r x[r][column] [row r in segment s]
                const int32_t* fr = &foldR[u % kFoldRing][0];
                const int cnt = __builtin_amdgcn_readfirstlane(fr[9]);
                if (cnt > 0) {
                    const int first = fr[8];
                    const u32x4 w0 = *reinterpret_cast<const u32x4*>(fr), w1 = *reinterpret_cast<const u32x4*>(fr + 4);
                    uint32_t id[8];                                        // local segment id of my row jj: byte g of word jj
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        id[i] = (w0[i] >> (8 * g)) & 0xffu;
                        id[4 + i] = (w1[i] >> (8 * g)) & 0xffu;
                    }
                    // (FOLD 2: the sum a tile's first segment may continue -- requested in front of the transposed reads, so that
                    //  one wait covers both; used or not is decided below)
                    f32x4 sg0 = {0.f, 0.f, 0.f, 0.f}, sg1 = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (FOLD == 2) {
                        sg0 = *reinterpret_cast<const f32x4*>(&segL[wave][g][0]);
                        sg1 = *reinterpret_cast<const f32x4*>(&segL[wave][g][4]);
                    }
                    bf16x8 a[2];
                    tr_frags(lds_base + sb, a);
                    for (int m0 = 0; m0 < cnt; m0 += 16) {                 // (more than 16 segments in 32 rows: graphs of 1-2 nodes)
                        const uint32_t me = (uint32_t)(m0 + j);
                        u32x4 iw;
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            iw[i] = (id[2 * i] == me ? 0x3f80u : 0u) | (id[2 * i + 1] == me ? 0x3f800000u : 0u);
                        const bf16x8 ind = __builtin_bit_cast(bf16x8, iw);
                        if constexpr (FOLD == 1) {
#pragma unroll
                            for (int n = 0; n < 2; ++n) {
                                const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[n], ind, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                                if ((int)me < cnt)
                                    *reinterpret_cast<f32x4*>(seg_part + (size_t)(first + (int)me) * kH + ocol + 4 * n) = d;
                            }
                        } else {                                           // a segment that is complete: its sum IS the aux row
                            // (graphs of any size -- fold record word 10: bit 0 = the tile's FIRST segment continues the previous X
                            //  unit's sum, bit 1 = its LAST segment ends in this tile.  A sum that goes on waits in this wave's 128
                            //  bytes of segL, not in registers; graphs inside one tile touch neither)
                            const int fb = __builtin_amdgcn_readfirstlane(fr[10]);
                            const bool carry = (fb & 1) && me == 0u;
                            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                            const f32x4 c0 = carry ? sg0 : z, c1 = carry ? sg1 : z;
                            const f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], ind, c0, 0, 0, 0);
                            const f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], ind, c1, 0, 0, 0);
                            if (!(fb & 2) && (int)me == cnt - 1) {         // goes on in the next tile (a later X unit of this wave)
                                *reinterpret_cast<f32x4*>(&segL[wave][g][0]) = d0;
                                *reinterpret_cast<f32x4*>(&segL[wave][g][4]) = d1;
                            } else if ((int)me < cnt) {
                                const u32x4 o = {pack_bf16x2(d0[0], d0[1]), pack_bf16x2(d0[2], d0[3]), pack_bf16x2(d1[0], d1[1]),
                                                 pack_bf16x2(d1[2], d1[3])};
                                *reinterpret_cast<u32x4*>(aux + (size_t)(first + (int)me) * kH + ocol) = o;
                            }
                        }
                    }
                }
            }
        } else {
            // ---- entry unit: acc[column][node] += sum_e S[e][column] * mask[e][node]
            p0 = u_aux;
#ifdef DN_TUNING_ENV
            if (flags & 32) {                                              // (ablation: entry units are fetched but not summed)
                asm volatile("ds_read_b128 %0, %1" : "=v"(dn) : "v"(an));
            } else
#endif
            {
            const u32x4 mA = *reinterpret_cast<const u32x4*>(&maskR[u % kMaskRing][8 * g]);
            const u32x4 mB = *reinterpret_cast<const u32x4*>(&maskR[u % kMaskRing][8 * g + 4]);
            bf16x8 a[2];
            tr_frags(lds_base + sb, a);
            asm volatile("ds_read_b128 %0, %1" : "=v"(dn) : "v"(an));
            bf16x8 sel[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const uint32_t rb = (uint32_t)(j + 16 * m);
                u32x4 sw;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    sw[i] = (((mA[2 * i] >> rb) & 1u) | (((mA[2 * i + 1] >> rb) & 1u) << 16)) * 0x3f80u;
                    sw[2 + i] = (((mB[2 * i] >> rb) & 1u) | (((mB[2 * i + 1] >> rb) & 1u) << 16)) * 0x3f80u;
                }
                sel[m] = __builtin_bit_cast(bf16x8, sw);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[n], sel[m], acc[m][n], 0, 0, 0);
            }
        }
        if (u_fl & kUnitLast) epilogue(p0, p0 + ((u_fl >> 8) & 0xff));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dn) : : "memory");     // my LDS reads are done before I arrive at a barrier
        u_fl = __builtin_amdgcn_readfirstlane((int)dn[0]);
        u_beg2 = __builtin_amdgcn_readfirstlane((int)dn[1]);
        u_end = __builtin_amdgcn_readfirstlane((int)dn[2]);
        u_aux = __builtin_amdgcn_readfirstlane((int)dn[3]);
    };
    int u = 0;
#pragma unroll 1
    for (; u < nt; ++u) {
        if (FOLD == 2 && (u_fl & kUnitAgg)) break;
        unit(u);
    }
    if constexpr (FOLD == 2) {
        if (u < nt) {                                                      // the workgroup's AGG units: W_agg replaces W_loop in wf
            if (w_kn == 0) {
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        wf[ks][n] = *reinterpret_cast<const bf16x8*>(W_agg + (size_t)(n0 + colA0 + 4 * n) * kH + ks * 32 + 8 * g);
            } else {
                dn_load_w_kn32<8>(W_agg, kH, n0, lane, wscr[wave], wf);
            }
            // phase B: nothing but AGG units from here on -- the same product on 32 aux rows, added to their output rows
            auto agg_unit = [&](int u) __attribute__((always_inline)) {
                if ((u & 1) == 0) __builtin_amdgcn_s_barrier();
                const unsigned an = desc_base + (unsigned)((u + 1) % kDescRing) * 16u;
                const unsigned sb = (unsigned)(u % kNS) * kStageB;
                const int32_t cnt = (u_fl & kUnitAgg) ? u_end - u_beg2 : 0;
                agg_fetch(&maskR[u % kMaskRing][0], cnt);
                bf16x8 xf[8][2];
                DN_FETCH(0, sb) DN_FETCH(1, sb) DN_FETCH(2, sb) DN_FETCH(3, sb) DN_FETCH(4, sb) DN_FETCH(5, sb) DN_FETCH(6, sb) DN_FETCH(7, sb)
                asm volatile("ds_read_b128 %0, %1" : "=v"(dn) : "v"(an));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
                DN_KSTEP(0, 15) DN_KSTEP(1, 13) DN_KSTEP(2, 11) DN_KSTEP(3, 9) DN_KSTEP(4, 7) DN_KSTEP(5, 5) DN_KSTEP(6, 3) DN_KSTEP(7, 1)
                epilogue_agg(cnt);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dn) : : "memory");
                u_fl = __builtin_amdgcn_readfirstlane((int)dn[0]);
                u_beg2 = __builtin_amdgcn_readfirstlane((int)dn[1]);
                u_end = __builtin_amdgcn_readfirstlane((int)dn[2]);
            };
#pragma unroll 1
            for (; u < nt; ++u) agg_unit(u);
        }
    }
#undef DN_FETCH
#undef DN_KSTEP
}
#undef DN_DS_READ128

}  // namespace

#ifdef DN_CLOSE_TIMES
extern "C" int dn_debug_close_times(unsigned long long* out) {             // diagnostic build only: 256 x 3 words of the last launch
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_close_times), sizeof(g_close_times)) == hipSuccess ? 0 : -2;
}
#endif

namespace dn_internal {

// dn_close_units_build_i32 for nd = 1 or 2 directions of one batch in ONE set of launches (same tiles count, lists and outputs per
// direction).  dirs[k].dyn != NULL: {edge rows, dropped range, go} are read on the device (ril_plan_kernel's words) instead of
// num_edge_rows / drop_beg / drop_end -- the launches are queued before the host knows them.  workspace: nd times
// dn_close_units_workspace_bytes.
// alt_order != 0 (dn_conv_index_build_i32): the same launches can build the CHUNKED tables instead (order alt_order over dirs[k].tile_ptr_alt
// / chunk_tile / chunk_graph / chunks_per_wg, alt_num_tiles = the bound those tables were sized by), picked per direction on the device
// by dyn[3] (1: the primary form, 2: the alternative); capacities and the workspace must cover max(num_tiles, alt_num_tiles).
int close_units_queue(int64_t N, int32_t num_wg, int64_t num_tiles, int32_t agg_units, int32_t xcd_order, int64_t num_list_entries,
                      int64_t unit_capacity, int nd, const CloseUnitsDir* dirs, void* workspace, size_t workspace_bytes, hipStream_t st,
                      int32_t alt_order, int64_t alt_num_tiles) {
    DN_REQUIRE(N >= 0 && N < 0x7fffffffLL && num_wg > 0 && num_wg <= 4096 && num_list_entries >= 0 && num_tiles >= 0 &&
               num_tiles < 0x7fffffffLL && (nd == 1 || nd == 2), "dn_close_units_build: bad sizes");
    DN_REQUIRE(xcd_order == 0 || xcd_order == 2 || ((xcd_order == 1 || xcd_order == 3) && num_wg % 8 == 0 && num_tiles < 0x0fffffffLL),
               "dn_close_units_build: the XCD order needs a multiple of 8 workgroups");
    for (int k = 0; k < nd; ++k) {
        DN_REQUIRE(dirs[k].num_edge_rows >= 0, "dn_close_units_build: bad sizes");
        DN_REQUIRE(xcd_order < 2 || (dirs[k].tile_ptr && dirs[k].chunk_tile && dirs[k].chunk_graph && dirs[k].chunks_per_wg >= 1 &&
                                     (int64_t)dirs[k].chunks_per_wg * num_wg <= num_tiles && dirs[k].chunks_per_wg == dirs[0].chunks_per_wg),
                   "dn_close_units_build: orders 2 / 3 take the tables of dn_fold_graph_tiles_multi_build_i32 (tile_ptr, chunk_tile, chunk_graph, chunks per workgroup)");
        DN_REQUIRE(dirs[k].tile_ptr != nullptr || num_tiles == dn_cdiv(N, 32), "dn_close_units_build: without tile_ptr the tiles are the "
                   "%lld 32-node windows", (long long)dn_cdiv(N, 32));
        DN_REQUIRE(dirs[k].unit_ptr, "dn_close_units_build: NULL pointer");
    }
    if (N == 0 || num_tiles == 0) {
        for (int k = 0; k < nd; ++k) DN_CHECK_HIP(hipMemsetAsync(dirs[k].unit_ptr, 0, sizeof(int32_t) * ((size_t)num_wg + 1), st));
        return DN_OK;
    }
    DN_REQUIRE(workspace, "dn_close_units_build: NULL pointer");
    for (int k = 0; k < nd; ++k) {
        DN_REQUIRE(dirs[k].list_ptr && dirs[k].list_rows && dirs[k].units && dirs[k].ent_row && dirs[k].ent_mask, "dn_close_units_build: NULL pointer");
        DN_REQUIRE(reinterpret_cast<uintptr_t>(dirs[k].units) % 16 == 0, "dn_close_units_build: unaligned pointer");
    }
    CbAlt alt{0, 0, 0};
    if (alt_order != 0) {
        DN_REQUIRE((alt_order == 2 || (alt_order == 3 && num_wg % 8 == 0)) && xcd_order < 2 && alt_num_tiles > 0 && alt_num_tiles < 0x0fffffffLL,
                   "dn_close_units_build: the alternative form is a chunked order behind a plain one");
        for (int k = 0; k < nd; ++k)
            DN_REQUIRE(dirs[k].dyn && dirs[k].tile_ptr_alt && dirs[k].chunk_tile && dirs[k].chunk_graph && dirs[k].chunks_per_wg >= 1 &&
                       (int64_t)dirs[k].chunks_per_wg * num_wg <= alt_num_tiles && dirs[k].chunks_per_wg == dirs[0].chunks_per_wg,
                       "dn_close_units_build: the alternative form needs dyn and the chunked tables");
        alt = CbAlt{alt_order, dirs[0].chunks_per_wg, (int32_t)alt_num_tiles};
    }
    const int64_t Tmax = alt_order != 0 && alt_num_tiles > num_tiles ? alt_num_tiles : num_tiles;
    DN_REQUIRE(unit_capacity >= dn_close_units_capacity(Tmax, num_list_entries, num_wg), "dn_close_units_build: unit table too small");
    DN_REQUIRE(workspace_bytes >= (size_t)nd * dn_close_units_workspace_bytes(Tmax, num_wg), "dn_close_units_build: workspace too small");
    DN_REQUIRE(reinterpret_cast<uintptr_t>(workspace) % 16 == 0, "dn_close_units_build: unaligned pointer");
    // (chunked orders: the scan runs over the workgroups' K chunk slots, Tper = K)
    const int32_t T = (int32_t)num_tiles, Tper = xcd_order >= 2 ? dirs[0].chunks_per_wg : (int32_t)dn_cdiv(T, num_wg);
    const int64_t M = std::max((int64_t)num_wg * Tper, (int64_t)num_wg * alt.chunks_per_wg);
    DN_REQUIRE((int64_t)nd * (unit_capacity + 1) < 0x7fffffffLL, "dn_close_units_build: unit tables too large");   // (one scan over both)
    char* wsp = reinterpret_cast<char*>(workspace);
    int32_t* tile_cnt[2] = {nullptr, nullptr};
    for (int k = 0; k < nd; ++k) { tile_cnt[k] = reinterpret_cast<int32_t*>(wsp); wsp += dn_align_up((size_t)(Tmax + 1) * 4, 256); }
    const size_t ne = (size_t)nd * (size_t)(M + 1);                          // ucnt / uoff of the directions back to back: ONE scan
    int32_t* ucnt = reinterpret_cast<int32_t*>(wsp);
    wsp += dn_align_up(ne * 4, 256);
    int32_t* uoff = reinterpret_cast<int32_t*>(wsp);
    wsp += dn_align_up(ne * 4, 256);
    size_t tb = 0;
    DN_CHECK_HIP(rocprim::exclusive_scan(nullptr, tb, ucnt, uoff, (int32_t)0, ne, rocprim::plus<int32_t>(), st));
    DN_REQUIRE(tb <= (size_t)nd * (65536 + 4 * (size_t)(M + 1)), "dn_close_units_build: scan storage %zu exceeds the reserved bound", tb);
    DN_CHECK_HIP(hipMemsetAsync(ucnt, 0, ne * 4, st));                       // positions without a tile (T not a multiple of num_wg) and the totals' slots
    CbPair pr;
    for (int k = 0; k < 2; ++k) {
        const CloseUnitsDir& d = dirs[k < nd ? k : 0];
        pr.d[k] = CbDir{d.tile_ptr, d.list_ptr, d.list_rows, d.drop_enable, d.dyn, d.chunk_tile, d.chunk_graph, d.tile_ptr_alt, d.ent_row,
                        tile_cnt[k < nd ? k : 0],
                        ucnt + (size_t)(k < nd ? k : 0) * (size_t)(M + 1), d.unit_ptr, d.ent_mask,
                        uoff + (size_t)(k < nd ? k : 0) * (size_t)(M + 1), reinterpret_cast<Unit*>(d.units), d.num_edge_rows, d.drop_beg,
                        d.drop_end};
    }
    hipLaunchKernelGGL(close_entries_kernel, dim3((unsigned)dn_cdiv(Tmax, kCbWaves), (unsigned)nd), dim3(kCbWaves * 64), 0, st, (int32_t)N, T,
                       num_wg, Tper, agg_units ? 1 : 0, xcd_order, pr, alt);
    DN_CHECK_LAUNCH();
    DN_CHECK_HIP(rocprim::exclusive_scan(wsp, tb, ucnt, uoff, (int32_t)0, ne, rocprim::plus<int32_t>(), st));
    const int64_t nthreads = Tmax > num_wg + 1 ? Tmax : num_wg + 1;
    hipLaunchKernelGGL(close_fill_kernel, dim3((unsigned)dn_cdiv(nthreads, 256), (unsigned)nd), dim3(256), 0, st, (int32_t)N, T, num_wg, Tper,
                       agg_units ? 1 : 0, xcd_order, pr, alt);
    DN_CHECK_LAUNCH();
    return DN_OK;
}

int fold_multi_queue(int64_t N, int64_t S, int nd, const FoldMultiDir* dirs, int32_t num_chunks, int64_t tile_capacity, bool with_valid,
                     hipStream_t st) {
    DN_REQUIRE(num_chunks >= 1 && num_chunks < kChunkMax, "dn_fold_graph_tiles_multi_build: 1 .. %d chunks", kChunkMax - 1);
    DN_REQUIRE(tile_capacity >= N / 32 + num_chunks + 1 && (nd == 1 || nd == 2), "dn_fold_graph_tiles_multi_build: tile tables too small");
    FmPair pr;
    for (int k = 0; k < 2; ++k) pr.d[k] = dirs[k < nd ? k : 0];
    if (with_valid)
        hipLaunchKernelGGL(fold_multi_valid_kernel, dim3((unsigned)dn_cdiv(S, 256), (unsigned)nd), dim3(256), 0, st, (int32_t)N, (int32_t)S, pr);
    hipLaunchKernelGGL(fold_multi_chunks_kernel, dim3((unsigned)nd), dim3(kChunkThreads), 0, st, (int32_t)N, (int32_t)S, num_chunks, pr);
    hipLaunchKernelGGL(fold_multi_tiles_kernel, dim3((unsigned)dn_cdiv(tile_capacity + 1, 256), (unsigned)nd), dim3(256), 0, st, (int32_t)N,
                       (int32_t)S, num_chunks, pr);
    DN_CHECK_LAUNCH();
    return DN_OK;
}

}  // namespace dn_internal

extern "C" {

int64_t dn_fold_graph_tiles_multi_capacity(int64_t N, int32_t num_chunks) { return N / 32 + num_chunks + 1; }

int dn_fold_graph_tiles_multi_build_i32(int64_t N, int64_t num_segments, const int32_t* seg_ptr, const int32_t* seg_nodes,
                                        const int32_t* add_idx, int32_t num_chunks, int32_t* chunk_tile, int32_t* chunk_graph,
                                        int32_t* tile_ptr, int32_t* fold_info, int64_t tile_capacity, int32_t* dev_ok,
                                        dn_stream_t stream) {
    DN_REQUIRE(N >= 0 && num_segments >= 0 && N < INT32_MAX && num_segments < INT32_MAX, "dn_fold_graph_tiles_multi_build: bad sizes");
    DN_REQUIRE(dev_ok, "dn_fold_graph_tiles_multi_build: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    if (N == 0 || num_segments == 0) { DN_CHECK_HIP(hipMemsetAsync(dev_ok, 0, sizeof(int32_t), st)); return DN_OK; }
    DN_REQUIRE(seg_ptr && seg_nodes && chunk_tile && chunk_graph && tile_ptr && fold_info, "dn_fold_graph_tiles_multi_build: NULL pointer");
    DN_REQUIRE(reinterpret_cast<uintptr_t>(fold_info) % 16 == 0, "dn_fold_graph_tiles_multi_build: unaligned pointer");
    DN_CHECK_HIP(hipMemsetAsync(dev_ok, 0x01, sizeof(int32_t), st));             // any non-zero value: "still valid"
    const dn_internal::FoldMultiDir d{seg_ptr, seg_nodes, add_idx, chunk_tile, chunk_graph, tile_ptr, fold_info, dev_ok, nullptr};
    return dn_internal::fold_multi_queue(N, num_segments, 1, &d, num_chunks, tile_capacity, true, st);
}

int64_t dn_close_units_capacity(int64_t num_tiles, int64_t num_list_entries, int32_t num_wg) {
    // X units + the rounding of the entry units, entry units, per workgroup the gap + the rounding of the AGG units, AGG units
    // (order 2: one per 32 GRAPHS, and a tile may hold up to 32 of them)
    return 2 * num_tiles + num_list_entries / 32 + 1 + (int64_t)num_wg * (kAggGap + 1) + num_tiles;
}

size_t dn_close_units_workspace_bytes(int64_t num_tiles, int32_t num_wg) {
    if (num_tiles < 0 || num_wg <= 0) { dn_set_error("dn_close_units_workspace_bytes: bad sizes"); return 0; }
    const int64_t M = (int64_t)num_wg * dn_cdiv(num_tiles, num_wg);
    // (the scan's temporary storage -- a look-back state per few thousand elements -- is bounded here without asking rocPRIM, so
    //  that the size query needs no device; dn_close_units_build_i32 checks the real requirement against it)
    const size_t scan_tmp = 65536 + 4 * (size_t)(M + 1);                  // (rocPRIM 4.x asks for ~(M + 1) / 8 bytes + 2 KB here)
    return dn_align_up((size_t)(num_tiles + 1) * 4, 256) + 2 * dn_align_up((size_t)(M + 1) * 4, 256) + dn_align_up(scan_tmp, 256) + 512;
}

int dn_close_units_build_i32(int64_t N, int32_t num_edge_rows, int32_t num_wg, const int32_t* tile_ptr, int64_t num_tiles,
                             int32_t agg_units, int32_t xcd_order, const int32_t* list_ptr, const int32_t* list_rows, int64_t num_list_entries,
                             int32_t drop_beg, int32_t drop_end, const int32_t* drop_enable, int32_t* unit_ptr, int32_t* units,
                             int64_t unit_capacity, int32_t* ent_row, uint32_t* ent_mask, const int32_t* chunk_tile,
                             const int32_t* chunk_graph, int32_t chunks_per_wg, void* workspace, size_t workspace_bytes,
                             dn_stream_t stream) {
    const dn_internal::CloseUnitsDir d{tile_ptr, list_ptr, list_rows, num_edge_rows, drop_beg, drop_end, drop_enable, nullptr, unit_ptr,
                                       units, ent_row, ent_mask, chunk_tile, chunk_graph, chunks_per_wg};
    return dn_internal::close_units_queue(N, num_wg, num_tiles, agg_units, xcd_order, num_list_entries, unit_capacity, 1, &d, workspace,
                                          workspace_bytes, (hipStream_t)stream, 0, 0);
}

int dn_fold_graph_tiles_build_i32(int64_t N, int64_t num_segments, const int32_t* seg_ptr, const int32_t* seg_nodes,
                                  const int32_t* add_idx, int32_t* tile_ptr, int32_t* fold_info, int32_t* dev_ok,
                                  dn_stream_t stream) {
    DN_REQUIRE(N >= 0 && num_segments >= 0 && N < INT32_MAX && num_segments < INT32_MAX, "dn_fold_graph_tiles_build: bad sizes");
    DN_REQUIRE(dev_ok, "dn_fold_graph_tiles_build: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    if (N == 0 || num_segments == 0) { DN_CHECK_HIP(hipMemsetAsync(dev_ok, 0, sizeof(int32_t), st)); return DN_OK; }
    DN_REQUIRE(seg_ptr && seg_nodes && tile_ptr && fold_info, "dn_fold_graph_tiles_build: NULL pointer");
    DN_REQUIRE(reinterpret_cast<uintptr_t>(fold_info) % 16 == 0, "dn_fold_graph_tiles_build: unaligned pointer");
    DN_CHECK_HIP(hipMemsetAsync(dev_ok, 0x01, sizeof(int32_t), st));             // any non-zero value: "still valid"
    hipLaunchKernelGGL(fold_graph_tiles_kernel, dim3((unsigned)dn_cdiv(num_segments + 1, 256)), dim3(256), 0, st, (int32_t)N,
                       (int32_t)num_segments, seg_ptr, seg_nodes, add_idx, tile_ptr, fold_info, dev_ok);
    DN_CHECK_LAUNCH();
    return DN_OK;
}

int dn_rows_close_bf16(const void* X, int32_t H, const void* W, int32_t w_kn, const void* bias, const void* S,
                       const int32_t* unit_ptr, const int32_t* units, int32_t num_wg, const int32_t* ent_row,
                       const uint32_t* ent_mask, int64_t N, void* out, const int32_t* fold_info, float* seg_part,
                       const void* W_agg, void* aux, const int32_t* agg_idx, dn_stream_t stream) {
    DN_REQUIRE(H == 256, "dn_rows_close: unsupported width %d (256 only; dn_rows_selfsum_bf16 serves 64 / 128)", H);
    DN_REQUIRE(N >= 0 && N < 0x7fffffffLL && num_wg > 0 && num_wg <= 4096, "dn_rows_close: bad sizes");
    const bool agg = W_agg != nullptr || aux != nullptr || agg_idx != nullptr;
    DN_REQUIRE(fold_info == nullptr || seg_part != nullptr || agg, "dn_rows_close: fold_info needs seg_part (or W_agg / aux / agg_idx)");
    DN_REQUIRE(!agg || (fold_info && W_agg && aux && agg_idx && !seg_part),
               "dn_rows_close: the absorbed fold takes fold_info, W_agg, aux and agg_idx together, without seg_part");
    if (N == 0) return DN_OK;
    DN_REQUIRE(X && W && unit_ptr && units && ent_row && ent_mask && out, "dn_rows_close: NULL pointer");
    DN_REQUIRE((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(S) |
                reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(units) | reinterpret_cast<uintptr_t>(bias) |
                reinterpret_cast<uintptr_t>(fold_info) | reinterpret_cast<uintptr_t>(seg_part) | reinterpret_cast<uintptr_t>(W_agg) |
                reinterpret_cast<uintptr_t>(aux)) % 16 == 0, "dn_rows_close: unaligned pointer");
    hipStream_t st = (hipStream_t)stream;
    static const int nt = dn_knob("DN_NT", 3);
    const int abl = dn_knob("DN_CLOSE_ABL", 0);   // tuning build only (read per call): 1 entry rows from L2, 2 x rows from L2, 4 no stores, 8 entry units not summed
    const int32_t flags = ((nt & 2) ? 2 : 0) | ((abl & 63) << 2);
    const bf16_t* s = S ? (const bf16_t*)S : (const bf16_t*)X;             // (no entry unit can exist without S; never dereferenced)
#define DN_CLOSE_LAUNCH(F)                                                                                                         \
    hipLaunchKernelGGL((rows_close_ring_kernel<F>), dim3((unsigned)num_wg), dim3(kThreads), 0, st, (const bf16_t*)X, (const bf16_t*)W, \
                       w_kn, (const bf16_t*)bias, s, reinterpret_cast<const Unit*>(units), unit_ptr, ent_row, ent_mask, (int32_t)N, \
                       flags, (bf16_t*)out, fold_info, seg_part, (const bf16_t*)W_agg, (bf16_t*)aux, agg_idx)
    if (agg) DN_CLOSE_LAUNCH(2);
    else if (fold_info) DN_CLOSE_LAUNCH(1);
    else DN_CLOSE_LAUNCH(0);
#undef DN_CLOSE_LAUNCH
    DN_CHECK_LAUNCH();
    return DN_OK;
}

}  // extern "C"
