// Graph-local build of the row factorisation (dn_row_index_build_local_i32): the same tables as dn_row_index_build_i32,
// bit for bit, for a batch given as a disjoint union of graphs (node_ptr / edge_ptr: every graph owns a contiguous node range
// and a contiguous edge range -- what dgl.batch / the PyG collate produce, subgraph_isomorphism/dataset.py:1605-1611).
//
// The general builder orders the whole batch with three device-wide radix sorts, eleven scans and ~60 launches (1.4 ms for
// the 4 M edges of BASELINE config 5, a step's worth for small batches where every launch is latency).  In a batch of graphs
// every ordering it establishes is LOCAL: the stable sort by (relation, node) is graph-major inside a relation, because node
// ids ascend with the graph.  So here one wavefront takes one graph, and the only batch-wide step is ONE exclusive scan over the
// packed per-(relation, graph) and per-node counts.  Three kernels + one scan + one read-back.
//
//   ril_stats_kernel   per relation: #edges, #distinct destinations, #distinct sources      (-> EDGE / AGG / TF, as before);
//                      per edge: is it the first of its (relation, destination) / (relation, source) pair (one byte, reused below).
//                      First occurrences come from an open-addressing hash table in LDS (compare-and-swap claims a slot per key,
//                      min keeps the lowest edge number); graphs over 256 edges compare all pairs instead.
//   ril_count_kernel   per (relation, graph): rows, AGG rows, TF rows, AGG edges, TF edges;  per node: list lengths -- sums over
//                      the edges, taken with LDS atomics (no comparisons)
//   (exclusive scan of the packed counts)
//   ril_fill_kernel    every table, each entry at  scan offset + rank inside the graph.  The ranks are counts of edges that
//                      compare lower in the (relation, key node, edge) order.  Round 5: for graphs of up to 256 edges whose
//                      (relation, node) buckets fit an 11.5 KB table they come from bit sets in LDS (atomic OR, popcounts, one
//                      prefix sum: fill_fast, no pair work -- the O(m^2) pass was 0.28 of the 0.94 ms a config-5 batch's index
//                      cost); larger graphs: all lanes at once against ONE edge whose packed words are wave-uniform (broadcast
//                      LDS read -> SGPRs), O(m^2) compares, no branch on the common path
//
// Round 6: graphs that do not fit the bit-set pass no longer compare all pairs of edges (O(m^2) per graph: 2.7 ms for the fill pass
// of 16,384 PROTEINS-shaped graphs).  Their ranks come from SORTS in LDS -- one bitonic sort of the packed (relation, key node,
// edge) words gives every edge's position, two more of (node, position) words the ranks inside the nodes' lists, prefix counts of
// the bucket starts the collapsed rows' numbers -- O(m log^2 m), identical tables.  And a graph over kLocM edges no longer sends
// the WHOLE batch to the general builder: the one-wavefront kernels skip it and list it, and a second launch of each pass
// (ril_stats_big_kernel / ril_fill_big_kernel: one 1024-thread workgroup per listed graph, the same sort code on 16 wavefronts)
// takes graphs of up to kBigM - 1 edges and kBigNodes nodes.  An empty list costs two empty launches.
//
// Graphs beyond those limits, an endpoint outside the graph's node range or a relation id outside [0, R) raise a flag
// instead (host_status = 1): the caller then runs the general builder.
#include <cstring>
#include <cstdlib>

#include "dn_common.h"
#include "dn_internal.h"
#include "../../include/dn_hip.h"

#include <rocprim/rocprim.hpp>

namespace {

constexpr int kLocM = 1024;          // edges of one graph held in one wavefront's LDS slice
constexpr int kBigM = 8192;          // ... in a whole workgroup's (the listed graphs: kLocM < edges < kBigM)
constexpr int kBigNodes = 8191;      // nodes of a listed graph (13-bit local ids next to 13-bit edge numbers)
constexpr int kBigWaves = 16;        // wavefronts that share one listed graph
constexpr int kBigGrid = 128;        // workgroups of the two launches over the list
constexpr int kLocNodes = 1 << 14;   // nodes of one graph (local ids are packed into 14 bits)
constexpr int kLocWaves = 4;         // graphs in flight per workgroup (one wavefront each)
constexpr int kLocShare = 16;        // consecutive graphs a workgroup owns per round (its wavefronts take them one at a time)
constexpr int kLocR = 64;            // relations (per-workgroup counters, one lane per relation)
constexpr int kEdge = 0, kAgg = 1, kTf = 2;
constexpr int kSeg = 5;              // packed count arrays over (relation, graph): rows, AGG rows, TF rows, AGG edges, TF edges
typedef uint32_t u32;

// One edge of the graph in LDS, two words:
//   w0 = relation << 26 | key node << 10 | edge          the edge's place in the (relation, key node, edge) order the general
//                                                         builder's stable sort establishes; key node = source in a TF relation,
//                                                         destination otherwise (local ids: node - first node of the graph)
//   w1 = head << 30 | mode << 28 | source << 14 | destination
// In the statistics pass (modes not known yet) w0 = relation << 14 | destination, w1 = relation << 14 | source.
struct LocLds {
    uint2 e[kLocWaves][kLocM + 4];   // + 4 sentinels behind the last edge (the pair loops read two edges at a time, one pair ahead)
};
constexpr u32 kSentinel0 = 0xffffffffu, kSentinel1 = 3u << 28;       // never "before" an edge, never equal to a key, mode 3

__device__ __forceinline__ int mode_of(int32_t Er, int32_t Dr, int32_t Sr, float edge_frac) {
    const int mn = Dr < Sr ? Dr : Sr;
    if (Er == 0 || (float)mn > edge_frac * (float)Er) return kEdge;       // (dn_index.hip: ri_mode_kernel)
    return Dr <= Sr ? kAgg : kTf;
}
__device__ __forceinline__ u32 sgpr(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }

// ---- the sort path: NW wavefronts (64 NW threads, `tid` among them) work on one graph's arrays in LDS.  NW = 1: the wavefront's own
// slice, ordered by the in-order LDS queue (a scheduling barrier is enough); NW > 1: the whole workgroup, real barriers.
template <int NW>
__device__ __forceinline__ void coop_sync() {
    if constexpr (NW == 1) __builtin_amdgcn_wave_barrier();
    else __syncthreads();
}
template <int NW>
__device__ __forceinline__ bool coop_any(bool v) {
    if constexpr (NW == 1) return __any(v) != 0;
    else return __syncthreads_or(v ? 1 : 0) != 0;
}
// One wavefront, NE keys per lane (key index = 64 r + lane): the bitonic network without LDS traffic -- partners 64 or more apart are
// two registers of the same lane, closer ones are exchanged lane to lane (ds_bpermute).  The LDS form below took ~300 cycles per stage
// (read -> compare -> write, one dependent round trip each); this one is bound by the exchange's latency across NE independent keys.
template <int NE>
__device__ __forceinline__ void wave_sort_regs(u32 (&v)[NE], int lane) {
#pragma unroll 1
    for (int k = 2; k <= NE * 64; k <<= 1) {
#pragma unroll
        for (int rj = NE / 2; rj >= 1; rj >>= 1) {
            if (rj * 64 < k) {                                             // stage j = 64 rj of this k (k >= 128: the direction is r's)
#pragma unroll
                for (int r = 0; r < NE; ++r)
                    if ((r & rj) == 0) {
                        const bool up = ((r * 64) & k) == 0;
                        const u32 a = v[r], b = v[r | rj];
                        const u32 lo = a < b ? a : b, hi = a < b ? b : a;
                        v[r] = up ? lo : hi;
                        v[r | rj] = up ? hi : lo;
                    }
            }
        }
#pragma unroll 1
        for (int j = (k >> 1) < 32 ? (k >> 1) : 32; j >= 1; j >>= 1) {
            const bool lower = (lane & j) == 0;
#pragma unroll
            for (int r = 0; r < NE; ++r) {
                const u32 o = (u32)__shfl_xor((int)v[r], j, 64);
                const bool up = (((r * 64) | lane) & k) == 0;
                const u32 mn = v[r] < o ? v[r] : o, mx = v[r] < o ? o : v[r];
                v[r] = (lower == up) ? mn : mx;
            }
        }
    }
}
template <int NE>
__device__ __forceinline__ void wave_sort_lds(u32* A, int lane) {
    u32 v[NE];
#pragma unroll
    for (int r = 0; r < NE; ++r) v[r] = A[r * 64 + lane];
    wave_sort_regs<NE>(v, lane);
#pragma unroll
    for (int r = 0; r < NE; ++r) A[r * 64 + lane] = v[r];
    __builtin_amdgcn_wave_barrier();
}

// ascending bitonic sort of A[0 .. n), n a power of two >= 64 (the caller pads with 0xffffffff); one wavefront: n <= 1024
template <int NW>
__device__ __forceinline__ void bitonic_sort(u32* A, int n, int tid) {
    if constexpr (NW == 1) {
        if (n <= 64) wave_sort_lds<1>(A, tid);
        else if (n == 128) wave_sort_lds<2>(A, tid);
        else if (n == 256) wave_sort_lds<4>(A, tid);
        else if (n == 512) wave_sort_lds<8>(A, tid);
        else wave_sort_lds<16>(A, tid);
        return;
    }
    constexpr int NT = 64 * NW;
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (n >> 1); t += NT) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));       // bit j clear
                const int l = i | j;
                const u32 a = A[i], b = A[l];
                const bool up = (i & k) == 0;
                const u32 lo = a < b ? a : b, hi = a < b ? b : a;
                A[i] = up ? lo : hi;
                A[l] = up ? hi : lo;
            }
            coop_sync<NW>();
        }
    }
}
__device__ __forceinline__ int lds_lower_bound(const u32* A, int n, u32 key) {   // first p in [0, n) with A[p] >= key (n: none)
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (A[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}
// exclusive scan of one value per thread, in thread order: sum (identity 0) or max (identity -1).  scratch: NW + 1 ints (NW > 1).
template <int NW, bool MAX>
__device__ __forceinline__ int coop_excl_scan(int v, int tid, int* scratch) {
    const int lane = tid & 63;
    int s = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(s, d, 64);
        if (lane >= d) s = MAX ? (o > s ? o : s) : s + o;
    }
    int excl = __shfl_up(s, 1, 64);
    if (lane == 0) excl = MAX ? -1 : 0;
    if constexpr (NW > 1) {
        const int wave = tid >> 6;
        if (lane == 63) scratch[wave] = s;
        __syncthreads();
        int carry = MAX ? -1 : 0;
        for (int w = 0; w < wave; ++w) carry = MAX ? (scratch[w] > carry ? scratch[w] : carry) : carry + scratch[w];
        __syncthreads();
        excl = MAX ? (carry > excl ? carry : excl) : excl + carry;
    }
    return excl;
}
__device__ __forceinline__ int pow2_at_least(int m) {
    int n = 64;
    while (n < m) n <<= 1;
    return n;
}

// body(j, w0, w1) for the edges j = 0 .. jend-1 (and possibly one more: a later edge or a sentinel -- every body is written so
// that such an edge contributes nothing), their two words wave-uniform in SGPRs; two edges per LDS read, the next pair in flight
template <typename F>
__device__ __forceinline__ void for_each_edge(const uint2* X, int jend, F&& body) {
    uint4 q = *reinterpret_cast<const uint4*>(X);
    for (int j = 0; j < jend; j += 2) {
        const uint4 qn = *reinterpret_cast<const uint4*>(X + j + 2);
        body(j, sgpr(q.x), sgpr(q.y));
        body(j + 1, sgpr(q.z), sgpr(q.w));
        q = qn;
    }
}

// Edges of graph g -> LDS (this wavefront's slice), packed for the statistics pass (s_mode == nullptr) or with the modes.
// Returns the edge count, or -1 when the graph is not taken (flagged in *bad by the statistics pass).
__device__ __forceinline__ int load_graph(uint2* X, int lane, int64_t g, int32_t R, const int32_t* node_ptr,
                                          const int32_t* edge_ptr, const int32_t* src, const int32_t* dst, const int32_t* etype,
                                          const int32_t* s_mode, const uint8_t* hbits, int64_t N, int64_t E, int& n0, int& n1,
                                          int& e0, int32_t* bad) {
    n0 = node_ptr[g]; n1 = node_ptr[g + 1];
    e0 = edge_ptr[g];
    const int m = edge_ptr[g + 1] - e0;
    // (ranges that leave [0, N) / [0, E) would index past the caller's arrays: never touch such a graph)
    if (m < 0 || m > kLocM || n1 < n0 || n1 - n0 > kLocNodes || n0 < 0 || n1 > N || e0 < 0 || (int64_t)e0 + m > E) {
        if (bad != nullptr && lane == 0) atomicOr(bad, 1);
        return -1;
    }
    bool oob = false;
    for (int i = lane; i < m; i += 64) {
        const int r = etype[e0 + i], s = src[e0 + i], d = dst[e0 + i];
        const bool o = (s < n0) | (s >= n1) | (d < n0) | (d >= n1) | (r < 0) | (r >= R);
        oob |= o;
        if (!o) {
            const u32 sl = (u32)(s - n0), dl = (u32)(d - n0);
            if (s_mode == nullptr) X[i] = make_uint2(((u32)r << 14) | dl, ((u32)r << 14) | sl);
            else {
                // head: first edge, in edge order, of its (relation, key node) pair -- the statistics pass left one byte per edge:
                // bit 0 = an earlier edge has my (relation, destination), bit 1 = ... my (relation, source)
                const u32 md = (u32)s_mode[r], hb = hbits[e0 + i];
                const u32 head = ((md == kTf ? hb >> 1 : hb) & 1u) ^ 1u;
                X[i] = make_uint2(((u32)r << 26) | ((md == kTf ? sl : dl) << 10) | (u32)i, (head << 30) | (md << 28) | (sl << 14) | dl);
            }
        }
    }
    if (lane < 4) X[m + lane] = make_uint2(kSentinel0, kSentinel1);
    if (__any(oob)) {
        if (bad != nullptr && lane == 0) atomicOr(bad, 1);
        return -1;
    }
    __builtin_amdgcn_wave_barrier();
    return m;
}

// The statistics of a graph from two SORTS (graphs the hash table below does not take): the words (relation, destination, edge)
// sorted, an edge is the first of its (relation, destination) pair exactly when its predecessor has another pair; the same with
// the sources.  A: n words, F: m words (a flag byte per edge would do; words keep the accesses plain).  NBITS: bits of a local
// node id, EB: bits of an edge number (6 + NBITS + EB <= 32).
template <int NW, int EB, int NBITS>
__device__ __forceinline__ void stats_sorted(u32* A, u32* F, int tid, int n0, int n1, int e0, int m, int32_t R,
                                             const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                             const int32_t* __restrict__ etype, int32_t (*cnt)[kLocR], uint8_t* __restrict__ hbits,
                                             int32_t* bad) {
    constexpr int NT = 64 * NW;
    constexpr u32 EM = (1u << EB) - 1u;
    static_assert(6 + NBITS + EB <= 32, "packed word");
    const int n = pow2_at_least(m);
    bool oob = false;
    for (int i = tid; i < n; i += NT) {
        u32 key = 0xffffffffu;
        if (i < m) {
            const int r = etype[e0 + i], sv = src[e0 + i], d = dst[e0 + i];
            const bool o = (sv < n0) | (sv >= n1) | (d < n0) | (d >= n1) | (r < 0) | (r >= R);
            oob |= o;
            if (!o) key = ((((u32)r << NBITS) | (u32)(d - n0)) << EB) | (u32)i;
            F[i] = 0;
        }
        A[i] = key;
    }
    if (coop_any<NW>(oob)) {                                               // (the graph is not taken: nothing has been counted)
        if (tid == 0) atomicOr(bad, 1);
        coop_sync<NW>();
        return;
    }
    coop_sync<NW>();
    bitonic_sort<NW>(A, n, tid);
    for (int p = tid; p < m; p += NT)
        if (p > 0 && (A[p] >> EB) == (A[p - 1] >> EB)) F[A[p] & EM] = 1u;  // an earlier edge has my (relation, destination)
    coop_sync<NW>();
    for (int i = tid; i < n; i += NT)
        A[i] = i < m ? ((((u32)etype[e0 + i] << NBITS) | (u32)(src[e0 + i] - n0)) << EB) | (u32)i : 0xffffffffu;
    coop_sync<NW>();
    bitonic_sort<NW>(A, n, tid);
    for (int p = tid; p < m; p += NT)
        if (p > 0 && (A[p] >> EB) == (A[p - 1] >> EB)) F[A[p] & EM] |= 2u;
    coop_sync<NW>();
    for (int i = tid; i < m; i += NT) {
        const u32 hb = F[i];
        const int r = etype[e0 + i];
        hbits[e0 + i] = (uint8_t)hb;                                       // (read back by the two later passes: no second search)
        atomicAdd(&cnt[0][r], 1);
        if ((hb & 1u) == 0) atomicAdd(&cnt[1][r], 1);
        if ((hb & 2u) == 0) atomicAdd(&cnt[2][r], 1);
    }
    coop_sync<NW>();                                                       // the next graph overwrites the arrays
}

// The statistics of a graph with at most kHashM edges WITHOUT pair work: both keys of every edge -- (relation, destination) and
// (relation, source) -- go into one open-addressing hash table in this wavefront's LDS slice (LDS compare-and-swap claims a slot
// for a key, LDS min keeps the lowest edge number per key); an edge is the first of its pair exactly when its own number came out.
constexpr int kHashM = 256;          // 2 keys x 256 edges in kHashSlots slots: load factor <= 0.5
constexpr int kHashSlots = 1024;     // key words [0, 1024) and minimum words [1024, 2048): 8 KB of the 8.2 KB slice
__device__ __forceinline__ void stats_hashed(u32* tab, int lane, int64_t g, int32_t R, const int32_t* node_ptr, const int32_t* edge_ptr,
                                             const int32_t* src, const int32_t* dst, const int32_t* etype, int64_t N, int64_t E,
                                             int32_t (*cnt)[kLocR], uint8_t* hbits, int32_t* bad) {
    const int n0 = node_ptr[g], n1 = node_ptr[g + 1], e0 = edge_ptr[g], m = edge_ptr[g + 1] - e0;
    if (m < 0 || n1 < n0 || n1 - n0 > kLocNodes || n0 < 0 || n1 > N || e0 < 0 || (int64_t)e0 + m > E) {
        if (lane == 0) atomicOr(bad, 1);
        return;
    }
    if (m == 0) return;
    u32* keys = tab;
    u32* mins = tab + kHashSlots;
    for (int i = lane; i < kHashSlots; i += 64) { keys[i] = 0xffffffffu; mins[i] = 0xffffffffu; }
    __builtin_amdgcn_wave_barrier();
    constexpr int KE = kHashM / 64;
    int slot_d[KE], slot_s[KE], rel[KE];
    bool oob = false;
#pragma unroll
    for (int k = 0; k < KE; ++k) {
        const int i = lane + 64 * k;
        slot_d[k] = slot_s[k] = -1; rel[k] = 0;
        if (i >= m) continue;
        const int r = etype[e0 + i], s = src[e0 + i], d = dst[e0 + i];
        if ((s < n0) | (s >= n1) | (d < n0) | (d >= n1) | (r < 0) | (r >= R)) { oob = true; continue; }
        rel[k] = r;
        const u32 key2[2] = {((u32)r << 14) | (u32)(d - n0), (1u << 31) | ((u32)r << 14) | (u32)(s - n0)};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            u32 h = (key2[q] * 2654435761u) >> 22;                         // 10 bits
            for (;;) {
                const u32 old = atomicCAS(&keys[h], 0xffffffffu, key2[q]);
                if (old == 0xffffffffu || old == key2[q]) break;
                h = (h + 1) & (kHashSlots - 1);
            }
            atomicMin(&mins[h], (u32)i);
            if (q == 0) slot_d[k] = (int)h; else slot_s[k] = (int)h;
        }
    }
    if (__any(oob)) {                                                      // (the graph is not taken: the tables stay unused)
        if (lane == 0) atomicOr(bad, 1);
        __builtin_amdgcn_wave_barrier();
        return;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < KE; ++k) {
        const int i = lane + 64 * k;
        if (i >= m) continue;
        const u32 dd = mins[slot_d[k]] != (u32)i ? 1u : 0u, ds = mins[slot_s[k]] != (u32)i ? 1u : 0u;
        hbits[e0 + i] = (uint8_t)(dd | (ds << 1));
        atomicAdd(&cnt[0][rel[k]], 1);
        if (dd == 0) atomicAdd(&cnt[1][rel[k]], 1);
        if (ds == 0) atomicAdd(&cnt[2][rel[k]], 1);
    }
    __builtin_amdgcn_wave_barrier();                                       // the next graph re-initialises the table
}

// Which pass takes graph g?  0: nobody (flagged), 1: the one-wavefront kernels, 2: the listed-graph kernels.
__device__ __forceinline__ int graph_class(int64_t g, const int32_t* node_ptr, const int32_t* edge_ptr, int64_t N, int64_t E) {
    const int n0 = node_ptr[g], n1 = node_ptr[g + 1], e0 = edge_ptr[g], m = edge_ptr[g + 1] - e0;
    // (ranges that leave [0, N) / [0, E) would index past the caller's arrays: never touch such a graph)
    if (m < 0 || n1 < n0 || n0 < 0 || n1 > N || e0 < 0 || (int64_t)e0 + m > E) return 0;
    if (m <= kLocM) return n1 - n0 <= kLocNodes ? 1 : 0;
    return (m < kBigM && n1 - n0 <= kBigNodes) ? 2 : 0;
}

__global__ __launch_bounds__(kLocWaves * 64) void ril_stats_kernel(int64_t G, int32_t R, const int32_t* __restrict__ node_ptr,
                                                                   const int32_t* __restrict__ edge_ptr,
                                                                   const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                                                   const int32_t* __restrict__ etype, int64_t N, int64_t E,
                                                                   int32_t* Er, int32_t* Dr, int32_t* Sr, int32_t* bad,
                                                                   uint8_t* __restrict__ hbits, int32_t* __restrict__ big_list) {
    __shared__ __attribute__((aligned(16))) LocLds L;
    __shared__ int32_t cnt[3][kLocR];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint2* X = L.e[wave];
    if (threadIdx.x < 3 * kLocR) cnt[threadIdx.x / kLocR][threadIdx.x % kLocR] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0 &&                           // the graphs must tile the node and edge ranges
        (node_ptr[0] != 0 || edge_ptr[0] != 0 || node_ptr[G] != N || edge_ptr[G] != E))
        atomicOr(bad, 1);
    __syncthreads();
    int max_n = 0, max_m = 0;                                            // the largest graph this wavefront has seen
    // (a workgroup owns kLocShare consecutive graphs per round and its wavefronts take them one at a time from an LDS counter:
    //  TU-shaped batches mix 10-edge and 1,000-edge graphs whose sorts differ a hundredfold, and a fixed pair per wavefront left
    //  three wavefronts waiting for the unlucky one.  A device-wide counter instead serialises: 32,768 atomics on one word took 0.5 ms)
    __shared__ int take;
    for (int64_t base = (int64_t)blockIdx.x * kLocShare; base < G; base += (int64_t)gridDim.x * kLocShare) {
      if (threadIdx.x == 0) take = 0;
      __syncthreads();
      for (;;) {
        int k = 0;
        if (lane == 0) k = atomicAdd(&take, 1);
        k = __shfl(k, 0, 64);
        const int64_t g = base + k;
        if (k >= kLocShare || g >= G) break;
        const int cls = graph_class(g, node_ptr, edge_ptr, N, E);
        if (cls == 0) {
            if (lane == 0) atomicOr(bad, 1);
            continue;
        }
        max_n = max(max_n, node_ptr[g + 1] - node_ptr[g]);
        max_m = max(max_m, edge_ptr[g + 1] - edge_ptr[g]);
        if (cls == 2) {                                                    // over this wavefront's slice: listed for ril_stats_big_kernel
            if (lane == 0) big_list[1 + atomicAdd(big_list, 1)] = (int32_t)g;
            continue;
        }
        const int m = edge_ptr[g + 1] - edge_ptr[g];
        if (m <= kHashM) {                                                 // up to 256 edges: first occurrences through a hash table
            stats_hashed(reinterpret_cast<u32*>(X), lane, g, R, node_ptr, edge_ptr, src, dst, etype, N, E, cnt, hbits, bad);
            continue;
        }
        u32* A = reinterpret_cast<u32*>(X);
        stats_sorted<1, 10, 14>(A, A + kLocM, lane, node_ptr[g], node_ptr[g + 1], edge_ptr[g], m, R, src, dst, etype, cnt, hbits, bad);
      }
      __syncthreads();                                                     // (the counter is reset for the next round)
    }
    // (bad[1], bad[2], zeroed with bad: the batch's largest graph.  Thousands of wavefronts raising the same two words serialise --
    //  0.2 ms at config 5 -- so a wavefront only asks when the word it reads is still below its own value)
    if (lane == 0 && max_n > 0) {
        if (max_n > __hip_atomic_load(bad + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(bad + 1, max_n);
        if (max_m > __hip_atomic_load(bad + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(bad + 2, max_m);
    }
    __syncthreads();
    if (threadIdx.x < 3 * kLocR) {
        const int k = threadIdx.x / kLocR, r = threadIdx.x % kLocR;
        if (r < R && cnt[k][r] != 0) atomicAdd((k == 0 ? Er : k == 1 ? Dr : Sr) + r, cnt[k][r]);
    }
}

// The same statistics for the listed graphs (kLocM < edges < kBigM): one workgroup of kBigWaves wavefronts per graph.
// big_list[0] = their number (left by ril_stats_kernel), big_list[1 ...] = the graphs, in no particular order.
__global__ __launch_bounds__(kBigWaves * 64) void ril_stats_big_kernel(int32_t R, const int32_t* __restrict__ node_ptr,
                                                                       const int32_t* __restrict__ edge_ptr,
                                                                       const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                                                       const int32_t* __restrict__ etype, int32_t* Er, int32_t* Dr,
                                                                       int32_t* Sr, int32_t* bad, uint8_t* __restrict__ hbits,
                                                                       const int32_t* __restrict__ big_list) {
    __shared__ __attribute__((aligned(16))) u32 A[kBigM];
    __shared__ __attribute__((aligned(16))) u32 F[kBigM];
    __shared__ int32_t cnt[3][kLocR];
    const int nbig = big_list[0];
    if ((int)blockIdx.x >= nbig) return;
    if (threadIdx.x < 3 * kLocR) cnt[threadIdx.x / kLocR][threadIdx.x % kLocR] = 0;
    __syncthreads();
    for (int k = blockIdx.x; k < nbig; k += gridDim.x) {
        const int64_t g = big_list[1 + k];
        stats_sorted<kBigWaves, 13, 13>(A, F, (int)threadIdx.x, node_ptr[g], node_ptr[g + 1], edge_ptr[g], edge_ptr[g + 1] - edge_ptr[g], R,
                                        src, dst, etype, cnt, hbits, bad);
    }
    __syncthreads();
    if (threadIdx.x < 3 * kLocR) {
        const int k = threadIdx.x / kLocR, r = threadIdx.x % kLocR;
        if (r < R && cnt[k][r] != 0) atomicAdd((k == 0 ? Er : k == 1 ? Dr : Sr) + r, cnt[k][r]);
    }
}

// C layout (int32): [kSeg][R][G] counts over (relation, graph), then N + 1 forward list lengths (the last one 0), then N + 1
// backward list lengths, then one closing 0 -- one exclusive scan gives every offset (a segment's own offsets = scan - scan at
// the segment's first element).
__global__ __launch_bounds__(kLocWaves * 64) void ril_count_kernel(int64_t G, int64_t N, int64_t E, int32_t R, float edge_frac, int32_t self_loop,
                                                                   const int32_t* __restrict__ node_ptr,
                                                                   const int32_t* __restrict__ edge_ptr,
                                                                   const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                                                   const int32_t* __restrict__ etype, const int32_t* __restrict__ Er,
                                                                   const int32_t* __restrict__ Dr, const int32_t* __restrict__ Sr,
                                                                   const uint8_t* __restrict__ hbits, int32_t* mode_out,
                                                                   int32_t* __restrict__ C) {
    // No pair work here: with the head bytes of the statistics pass every count is a sum over the graph's edges, taken with LDS
    // atomics -- per node one word {forward list entries | backward list entries << 16}, per relation {edges, heads}.  Node
    // counters cover kLocCnt nodes at a time (a larger graph walks its edges once per window).
    constexpr int kLocCnt = 1024;
    __shared__ uint32_t s_cnt[kLocWaves][kLocCnt];
    __shared__ uint32_t s_rel[kLocWaves][2][kLocR];
    __shared__ int32_t s_mode[kLocR];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* cnt = s_cnt[wave];
    if (threadIdx.x < kLocR) {
        const int md = threadIdx.x < R ? mode_of(Er[threadIdx.x], Dr[threadIdx.x], Sr[threadIdx.x], edge_frac) : kEdge;
        s_mode[threadIdx.x] = md;
        if (blockIdx.x == 0 && threadIdx.x < R) mode_out[threadIdx.x] = md;
    }
    __syncthreads();
    const int64_t RG = (int64_t)R * G;
    int32_t* Cf = C + kSeg * RG;
    int32_t* Cb = Cf + (N + 1);
    const int my_mode = s_mode[lane];
    for (int64_t g = (int64_t)blockIdx.x * kLocWaves + wave; g < G; g += (int64_t)gridDim.x * kLocWaves) {
        const int n0 = node_ptr[g], n1 = node_ptr[g + 1], e0 = edge_ptr[g], m = edge_ptr[g + 1] - e0, n = n1 - n0;
        if (graph_class(g, node_ptr, edge_ptr, N, E) == 0) continue;        // (flagged)
        s_rel[wave][0][lane] = 0; s_rel[wave][1][lane] = 0;
        for (int v0 = 0; v0 < max(n, 1); v0 += kLocCnt) {
            for (int i = lane; i < kLocCnt; i += 64) cnt[i] = 0;
            __builtin_amdgcn_wave_barrier();
            for (int i = lane; i < m; i += 64) {
                const int r = etype[e0 + i], sl = src[e0 + i] - n0, dl = dst[e0 + i] - n0;
                if (r < 0 || r >= R || sl < 0 || sl >= n || dl < 0 || dl >= n) continue;     // (flagged by the statistics pass)
                const u32 md = (u32)s_mode[r], hb = hbits[e0 + i];
                const u32 head = ((md == kTf ? hb >> 1 : hb) & 1u) ^ 1u;
                if (v0 == 0) {
                    atomicAdd(&s_rel[wave][0][r], 1u);
                    if (head) atomicAdd(&s_rel[wave][1][r], 1u);
                }
                const u32 wf = md == kAgg ? head : 1u, wb = md == kTf ? head : 1u;            // a per-edge entry, or a collapsed row's one
                if (wf && dl >= v0 && dl < v0 + kLocCnt) atomicAdd(&cnt[dl - v0], 1u);
                if (wb && sl >= v0 && sl < v0 + kLocCnt) atomicAdd(&cnt[sl - v0], 1u << 16);
            }
            __builtin_amdgcn_wave_barrier();
            const int selfs = self_loop ? 1 : 0;
            for (int v = v0 + lane; v < min(n, v0 + kLocCnt); v += 64) {
                const u32 c = cnt[v - v0];
                Cf[n0 + v] = selfs + (int)(c & 0xffffu);
                Cb[n0 + v] = selfs + (int)(c >> 16);
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (lane < R) {
            const int edges = (int)s_rel[wave][0][lane], heads = (int)s_rel[wave][1][lane];
            const int64_t at = (int64_t)lane * G + g;
            C[at] = my_mode == kEdge ? edges : heads;
            C[RG + at] = my_mode == kAgg ? heads : 0;
            C[2 * RG + at] = my_mode == kTf ? heads : 0;
            C[3 * RG + at] = my_mode == kAgg ? edges : 0;
            C[4 * RG + at] = my_mode == kTf ? edges : 0;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

struct FillArgs {
    int64_t G, N;
    const int32_t *S0, *S1, *S2, *S3, *S4, *Sf, *Sb;
    int32_t b1, b2, b3, b4, bf, bb, self_loop;
    int32_t *row_in, *row_out, *aux_f_ptr, *aux_f_idx, *aux_b_ptr, *aux_b_idx, *dst_rows, *src_rows;
};

// What a pass over the graph's edges leaves per edge of this lane (one per 64-edge chunk) -- the five counts the tables' offsets
// are made of -- and the writes they lead to.
template <int NC>
struct Ranks {
    u32 w0[NC], w1[NC];
    int rank_rel[NC], heads_before[NC], rank_d[NC], rank_s[NC], later_rows[NC];
};

template <int NC>
__device__ __forceinline__ void fill_emit(const Ranks<NC>& K, int lane, int m, int c0, int64_t g, int n0, const FillArgs& A) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int i = (c0 + c) * 64 + lane;
        if (i >= m) continue;
        const u32 w0 = K.w0[c], w1 = K.w1[c];
        const int r = (int)(w0 >> 26), md = (int)((w1 >> 28) & 3u), hd = (int)((w1 >> 30) & 1u);
        const int s = n0 + (int)((w1 >> 14) & 0x3fffu), d = n0 + (int)(w1 & 0x3fffu), kn = md == kTf ? s : d;
        const int64_t at = (int64_t)r * A.G + g;
        const int32_t row = A.S0[at] + (md == kEdge ? K.rank_rel[c] : K.heads_before[c]);
        if (md == kEdge) {
            A.row_in[row] = s; A.row_out[row] = d;
        } else if (md == kAgg) {
            const int32_t pos = A.S3[at] - A.b3 + K.rank_rel[c];            // among the AGG edges, (relation, dst, edge) order
            A.aux_f_idx[pos] = s;
            if (hd) {
                const int32_t a = A.S1[at] - A.b1 + K.heads_before[c];
                A.row_in[row] = (int32_t)A.N + a; A.row_out[row] = kn;
                A.aux_f_ptr[a] = pos;
                // node kn's list: per-edge entries, then its collapsed rows in row order, then the self loop (if any): counted
                // from the END of the list, which the scan knows
                A.dst_rows[A.Sf[kn + 1] - A.bf - A.self_loop - 1 - K.later_rows[c]] = row;
            }
        } else {
            const int32_t pos = A.S4[at] - A.b4 + K.rank_rel[c];
            A.aux_b_idx[pos] = d;
            if (hd) {
                const int32_t a = A.S2[at] - A.b2 + K.heads_before[c];
                A.row_in[row] = kn; A.row_out[row] = (int32_t)A.N + a;
                A.aux_b_ptr[a] = pos;
                A.src_rows[A.Sb[kn + 1] - A.bb - A.self_loop - 1 - K.later_rows[c]] = row;
            }
        }
        if (md != kAgg) A.dst_rows[A.Sf[d] - A.bf + K.rank_d[c]] = row;
        if (md != kTf) A.src_rows[A.Sb[s] - A.bb + K.rank_s[c]] = row;
    }
}

// One edge's table entries from its five ranks (fill_emit for a single edge).
__device__ __forceinline__ void fill_emit_one(u32 r, u32 md, u32 hd, int s, int d, int rank_rel, int heads_before, int rank_d, int rank_s,
                                              int later_rows, int64_t g, const FillArgs& A) {
    const int kn = md == kTf ? s : d;
    const int64_t at = (int64_t)r * A.G + g;
    const int32_t row = A.S0[at] + (md == kEdge ? rank_rel : heads_before);
    if (md == kEdge) {
        A.row_in[row] = s; A.row_out[row] = d;
    } else if (md == kAgg) {
        const int32_t pos = A.S3[at] - A.b3 + rank_rel;
        A.aux_f_idx[pos] = s;
        if (hd) {
            const int32_t a = A.S1[at] - A.b1 + heads_before;
            A.row_in[row] = (int32_t)A.N + a; A.row_out[row] = kn;
            A.aux_f_ptr[a] = pos;
            A.dst_rows[A.Sf[kn + 1] - A.bf - A.self_loop - 1 - later_rows] = row;
        }
    } else {
        const int32_t pos = A.S4[at] - A.b4 + rank_rel;
        A.aux_b_idx[pos] = d;
        if (hd) {
            const int32_t a = A.S2[at] - A.b2 + heads_before;
            A.row_in[row] = kn; A.row_out[row] = (int32_t)A.N + a;
            A.aux_b_ptr[a] = pos;
            A.src_rows[A.Sb[kn + 1] - A.bb - A.self_loop - 1 - later_rows] = row;
        }
    }
    if (md != kAgg) A.dst_rows[A.Sf[d] - A.bf + rank_d] = row;
    if (md != kTf) A.src_rows[A.Sb[s] - A.bb + rank_s] = row;
}

// B sorted ascending: words (node << PB | position) of the participating edges, 0xffffffff behind them.  An edge's rank inside
// its node's run = its index in B minus the index of the run's first word.  Every thread takes n / (64 NW) consecutive words; the
// run start it inherits comes from a max-scan over the threads.  store(position, rank).
template <int NW, int PB, typename F>
__device__ __forceinline__ void run_ranks(const u32* B, int n, int tid, int* scratch, F&& store) {
    const int K = n / (64 * NW), q0 = tid * K;
    int last = -1;
    for (int k = 0; k < K; ++k) {
        const int q = q0 + k;
        if (q == 0 || (B[q - 1] >> PB) != (B[q] >> PB)) last = q;
    }
    int cur = coop_excl_scan<NW, true>(last, tid, scratch);
    for (int k = 0; k < K; ++k) {
        const int q = q0 + k;
        const u32 key = B[q];
        if (q == 0 || (B[q - 1] >> PB) != (key >> PB)) cur = q;
        if (key != 0xffffffffu) store((int)(key & ((1u << PB) - 1u)), q - cur);
    }
}

// The five counts from SORTS instead of pair work, for any graph the bit sets below do not take (and, on 16 wavefronts, for the
// listed graphs).  X: the graph's edges as load_graph packs them with EB bits of edge number -- w0 = relation << 26 | key node <<
// EB | edge, w1 = head << 30 | mode << 28 | source << 14 | destination.  A, B: n words each (n = the power of two >= m), rsv: R + 1.
//   1. A = the w0 words sorted: an edge's index p in A is its POSITION in the (relation, key node, edge) order; rank inside the
//      relation = p - rsv[relation] (rsv by binary search).  X[edge].x is free from here on (A holds the word): it takes the ranks.
//   2. B = (destination << EB | p) of the edges with a per-edge forward entry, sorted: rank inside the destination's list = index
//      in B - start of the destination's run (run_ranks).  The same with the sources for the backward lists.
//   3. B = inclusive count of BUCKET STARTS (a new (relation, key node) in A): the number of a collapsed row inside its relation
//      = buckets before it.
//   4. the rows of OTHER collapsed relations of the same mode at the same key node that come after this one in the node's list
//      (later_rows): one binary search per such relation, heads of collapsed rows only.
template <int NW, int EB>
__device__ __forceinline__ void fill_sorted(uint2* X, u32* A, u32* B, int* rsv, int* scratch, int tid, int m, int32_t R, int64_t g,
                                            int n0, const FillArgs& Ar, const int32_t* s_mode) {
    constexpr int NT = 64 * NW;
    constexpr u32 EM = (1u << EB) - 1u, KM = (1u << (26 - EB)) - 1u;
    const int n = pow2_at_least(m);
    for (int i = tid; i < n; i += NT) {
        A[i] = i < m ? X[i].x : 0xffffffffu;
        if (i < m) X[i].x = 0u;
    }
    coop_sync<NW>();
    bitonic_sort<NW>(A, n, tid);
    for (int r = tid; r < R; r += NT) rsv[r] = lds_lower_bound(A, m, (u32)r << 26);
    if (tid == 0) rsv[R] = m;
    for (int pass = 0; pass < 2; ++pass) {                                 // 0: forward lists (by destination), 1: backward (by source)
        for (int p = tid; p < n; p += NT) {
            u32 key = 0xffffffffu;
            if (p < m) {
                const u32 w1 = X[A[p] & EM].y, md = (w1 >> 28) & 3u;
                const bool part = pass == 0 ? (md == kEdge || md == kTf) : (md == kEdge || md == kAgg);
                const u32 node = pass == 0 ? (w1 & 0x3fffu) : ((w1 >> 14) & 0x3fffu);
                if (part) key = (node << EB) | (u32)p;
            }
            B[p] = key;
        }
        coop_sync<NW>();
        bitonic_sort<NW>(B, n, tid);
        if (pass == 0) run_ranks<NW, EB>(B, n, tid, scratch, [&](int p, int rank) { X[A[p] & EM].x = (u32)rank; });
        else run_ranks<NW, EB>(B, n, tid, scratch, [&](int p, int rank) { X[A[p] & EM].x |= (u32)rank << 16; });
        coop_sync<NW>();
    }
    {   // B[p] = bucket starts in A[0 .. p]
        const int K = n / NT, q0 = tid * K;
        int c = 0;
        for (int k = 0; k < K; ++k) {
            const int q = q0 + k;
            if (q < m && (q == 0 || (A[q - 1] >> EB) != (A[q] >> EB))) ++c;
        }
        int run = coop_excl_scan<NW, false>(c, tid, scratch);
        for (int k = 0; k < K; ++k) {
            const int q = q0 + k;
            if (q < m && (q == 0 || (A[q - 1] >> EB) != (A[q] >> EB))) ++run;
            B[q] = (u32)run;
        }
    }
    coop_sync<NW>();
    for (int p = tid; p < m; p += NT) {
        const u32 w0 = A[p];
        const uint2 x = X[w0 & EM];
        const u32 r = w0 >> 26, md = (x.y >> 28) & 3u, hd = (x.y >> 30) & 1u, kl = (w0 >> EB) & KM;
        const int rs0 = rsv[r];
        int heads_before = 0, later = 0;
        if (md != kEdge) {
            heads_before = (int)B[p] - 1 - (rs0 > 0 ? (int)B[rs0 - 1] : 0);
            if (hd)
                for (int r2 = (int)r + 1; r2 < R; ++r2) {
                    if (s_mode[r2] != (int)md) continue;
                    const u32 k2 = ((u32)r2 << 26) | (kl << EB);
                    const int q = lds_lower_bound(A, m, k2);
                    later += (q < m && (A[q] >> EB) == (k2 >> EB)) ? 1 : 0;
                }
        }
        fill_emit_one(r, md, hd, n0 + (int)((x.y >> 14) & 0x3fffu), n0 + (int)(x.y & 0x3fffu), p - rs0, heads_before, (int)(x.x & 0xffffu),
                      (int)(x.x >> 16), later, g, Ar);
    }
    coop_sync<NW>();                                                       // the next graph rewrites the arrays
}

// The same five counts WITHOUT pair work, for a graph small enough for bit sets in this wavefront's LDS table T (kFastWords
// words): m <= 256 edges, n R (MW + 1) <= kFastWords with MW = ceil(m / 32).  All of it is order-free (LDS atomic OR, popcounts,
// one prefix sum), so the result is the O(m^2) pass's, bit for bit:
//   1. E[bucket] = the set of edge numbers of bucket (relation, key node)              (n R buckets x MW words)
//   2. start[bucket] = edges in the buckets before it (prefix sum of the popcounts)    -> an edge's POSITION in the (relation,
//      key node, edge) order = start[my bucket] + (members of my bucket with a smaller edge number); rank inside the relation
//      = position - start[first bucket of the relation]
//   3. D[node] / S[node] = the set of POSITIONS of the per-edge forward / backward entries at that destination / source
//      (over E's space, which is dead by then)  -> rank inside the node's list = members below my position
//   4. the few heads of collapsed rows are compacted into a list; every lane walks it (the slow pass's rare branch).
constexpr int kFastWords = 2944;     // 11.5 KB per wavefront (two workgroups of four per CU next to the edge slices)
constexpr int kFastM = 256;
// sets are MQ 16-byte quads wide (MQ = 1: up to 128 members, 2: up to 256): one or two ds_read_b128 per set, no loops over words
__device__ __forceinline__ bool fast_fits(int m, int n, int R) {
    const int MW = m <= 128 ? 4 : 8;
    // (the bucket sets + start[] first, then -- over the same words -- the two position sets per node + 64 listed heads)
    return m <= kFastM && n >= 1 && (int64_t)n * R * (MW + 1) <= kFastWords && (int64_t)2 * n * MW + 2 * 64 + 4 <= kFastWords;
}
template <int MQ>
__device__ __forceinline__ int set_count(const u32* B) {
    int c = 0;
#pragma unroll
    for (int q = 0; q < MQ; ++q) {
        const uint4 v = *reinterpret_cast<const uint4*>(B + 4 * q);
        c += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
    }
    return c;
}
template <int MQ>
__device__ __forceinline__ int set_below(const u32* B, int p) {             // members of the set below position p
    int c = 0;
    const int wq = p >> 5;
    const u32 part = (1u << (p & 31)) - 1u;
#pragma unroll
    for (int q = 0; q < MQ; ++q) {
        const uint4 v = *reinterpret_cast<const uint4*>(B + 4 * q);
        const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int wi = 4 * q + k;
            c += __popc(w[k] & (wi < wq ? 0xffffffffu : (wi == wq ? part : 0u)));
        }
    }
    return c;
}
template <int NC, int MQ>
__device__ __forceinline__ void fill_fast(const uint2* X, u32* T, int lane, int m, int n, int R, int64_t g, int n0, const FillArgs& A) {
    constexpr int MW = 4 * MQ;
    const int NB = n * R;
    u32* const St = T + NB * MW;                                            // start[bucket]
    Ranks<NC> K;
    int bucket[NC], pos[NC];
    for (int w = lane; w < NB * MQ; w += 64) reinterpret_cast<uint4*>(T)[w] = make_uint4(0u, 0u, 0u, 0u);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int i = c * 64 + lane;
        const uint2 x = i < m ? X[i] : make_uint2(kSentinel0, kSentinel1);
        K.w0[c] = x.x; K.w1[c] = x.y;
        K.rank_rel[c] = K.heads_before[c] = K.rank_d[c] = K.rank_s[c] = K.later_rows[c] = 0;
        bucket[c] = i < m ? (int)(x.x >> 26) * n + (int)((x.x >> 10) & 0xffffu) : 0;
        if (i < m) atomicOr(&T[bucket[c] * MW + (i >> 5)], 1u << (i & 31));
    }
    __builtin_amdgcn_wave_barrier();
    {   // start[]: every lane takes BPL consecutive buckets (their sizes requested together), one wave-wide exclusive sum joins them
        constexpr int kBplMax = kFastWords / 5 / 64 + 1;                    // NB <= kFastWords / (MW + 1)
        const int BPL = (NB + 63) >> 6, b0 = lane * BPL;
        int sz[kBplMax];
        int mine = 0;
#pragma unroll
        for (int k = 0; k < kBplMax; ++k) {
            sz[k] = (k < BPL && b0 + k < NB) ? set_count<MQ>(T + (b0 + k) * MW) : 0;
            mine += sz[k];
        }
        int run = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(run, d, 64);
            if (lane >= d) run += o;
        }
        run -= mine;                                                       // edges in the buckets of the lanes before me
#pragma unroll
        for (int k = 0; k < kBplMax; ++k) {
            if (k < BPL && b0 + k < NB) St[b0 + k] = (u32)run;
            run += sz[k];
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int i = c * 64 + lane;
        pos[c] = 0;
        if (i < m) {
            pos[c] = (int)St[bucket[c]] + set_below<MQ>(T + bucket[c] * MW, i);
            K.rank_rel[c] = pos[c] - (int)St[(int)(K.w0[c] >> 26) * n];
        }
    }
    __builtin_amdgcn_wave_barrier();                                       // E is dead: its space holds D (n x MW) and S (n x MW) now
    u32* const Dd = T;
    u32* const Ss = T + n * MW;
    uint2* const Hl = reinterpret_cast<uint2*>(T + 2 * n * MW);            // heads of collapsed rows (<= 64 kept here)
    for (int w = lane; w < 2 * n * MQ; w += 64) reinterpret_cast<uint4*>(T)[w] = make_uint4(0u, 0u, 0u, 0u);
    __builtin_amdgcn_wave_barrier();
    int hn = 0;                                                            // heads listed so far (wave-uniform)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int i = c * 64 + lane;
        const u32 md = (K.w1[c] >> 28) & 3u;
        const bool live = i < m;
        if (live && (md == kEdge || md == kTf)) atomicOr(&Dd[(int)(K.w1[c] & 0x3fffu) * MW + (pos[c] >> 5)], 1u << (pos[c] & 31));
        if (live && (md == kEdge || md == kAgg)) atomicOr(&Ss[(int)((K.w1[c] >> 14) & 0x3fffu) * MW + (pos[c] >> 5)], 1u << (pos[c] & 31));
        const bool head = live && ((K.w1[c] >> 30) & 1u) != 0 && (md == kAgg || md == kTf);
        const unsigned long long hb = __ballot(head);
        const int at = hn + __popcll(hb & ((1ull << lane) - 1ull));
        if (head && at < 64) Hl[at] = make_uint2(K.w0[c], K.w1[c]);
        hn += __popcll(hb);
    }
    const bool many = hn > 64;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int i = c * 64 + lane;
        if (i < m) {
            K.rank_d[c] = set_below<MQ>(Dd + (int)(K.w1[c] & 0x3fffu) * MW, pos[c]);
            K.rank_s[c] = set_below<MQ>(Ss + (int)((K.w1[c] >> 14) & 0x3fffu) * MW, pos[c]);
        }
    }
    // the collapsed rows' two counts, against the heads only (or, a graph with more than 64 of them, against every edge as the
    // O(m^2) pass does)
    auto head_counts = [&](u32 j0, u32 j1) {
        const u32 rj = j0 >> 26, mdj = (j1 >> 28) & 3u;
        const u32 kkj = j0 >> 10, ckj = (mdj << 16) | (kkj & 0xffffu);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const u32 rcc = K.w0[c] >> 26, ckc = (((K.w1[c] >> 28) & 3u) << 16) | ((K.w0[c] >> 10) & 0xffffu);
            K.heads_before[c] += ((rcc == rj) & (kkj < (K.w0[c] >> 10))) ? 1 : 0;
            K.later_rows[c] += ((ckc == ckj) & (rj > rcc)) ? 1 : 0;
        }
    };
    if (!many) {
        for (int k = 0; k < hn; ++k) {
            const uint2 h = Hl[k];
            head_counts(sgpr(h.x), sgpr(h.y));
        }
    } else {
        for_each_edge(X, m, [&](int, u32 j0, u32 j1) {
            const u32 mdj = (j1 >> 28) & 3u;
            if (((j1 >> 30) & 1u) != 0 && (mdj == kAgg || mdj == kTf)) head_counts(j0, j1);
        });
    }
    fill_emit<NC>(K, lane, m, 0, g, n0, A);
    __builtin_amdgcn_wave_barrier();                                       // the next graph rewrites the table
}

__global__ __launch_bounds__(kLocWaves * 64) void ril_fill_kernel(
    int64_t G, int64_t N, int64_t E, int32_t R, int32_t self_loop, const int32_t* __restrict__ node_ptr, const int32_t* __restrict__ edge_ptr,
    const int32_t* __restrict__ src, const int32_t* __restrict__ dst, const int32_t* __restrict__ etype,
    const int32_t* __restrict__ mode, const uint8_t* __restrict__ hbits, const int32_t* __restrict__ S, int32_t* __restrict__ row_in,
    int32_t* __restrict__ row_out,
    int32_t* __restrict__ aux_f_ptr, int32_t* __restrict__ aux_f_idx, int32_t* __restrict__ aux_b_ptr,
    int32_t* __restrict__ aux_b_idx, int32_t* __restrict__ dst_ptr, int32_t* __restrict__ dst_rows, int32_t* __restrict__ src_ptr,
    int32_t* __restrict__ src_rows, int32_t* __restrict__ meta /* [5] totals, [R + 1] rel_ptr, [R] modes, status */,
    int32_t* __restrict__ bad, int32_t* __restrict__ rel_ptr_out /* may be NULL: [R + 2] = rel_ptr, then the end of the self-loop rows */) {
    __shared__ __attribute__((aligned(16))) LocLds L;
    __shared__ __attribute__((aligned(16))) u32 ftab[kLocWaves][kFastWords];
    __shared__ int32_t s_mode[kLocR];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint2* X = L.e[wave];
    if (threadIdx.x < kLocR) s_mode[threadIdx.x] = threadIdx.x < R ? mode[threadIdx.x] : kEdge;
    __syncthreads();
    const int64_t RG = (int64_t)R * G;
    FillArgs A;
    A.G = G; A.N = N; A.self_loop = self_loop ? 1 : 0;
    A.S0 = S; A.S1 = S + RG; A.S2 = S + 2 * RG; A.S3 = S + 3 * RG; A.S4 = S + 4 * RG; A.Sf = S + kSeg * RG; A.Sb = A.Sf + (N + 1);
    A.b1 = A.S1[0]; A.b2 = A.S2[0]; A.b3 = A.S3[0]; A.b4 = A.S4[0]; A.bf = A.Sf[0]; A.bb = A.Sb[0];           // (S0[0] == 0)
    A.row_in = row_in; A.row_out = row_out; A.aux_f_ptr = aux_f_ptr; A.aux_f_idx = aux_f_idx; A.aux_b_ptr = aux_b_ptr;
    A.aux_b_idx = aux_b_idx; A.dst_rows = dst_rows; A.src_rows = src_rows;
    const int32_t P = A.b1, n_agg = A.b2 - A.b1, n_tf = A.b3 - A.b2, n_agg_e = A.b4 - A.b3, n_tf_e = A.bf - A.b4;
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) {
            meta[0] = P; meta[1] = n_agg; meta[2] = n_tf; meta[3] = n_agg_e; meta[4] = n_tf_e;
            aux_f_ptr[n_agg] = n_agg_e;
            aux_b_ptr[n_tf] = n_tf_e;
            const int32_t tf_ = A.Sb[0] - A.bf, tb_ = A.Sb[N + 1] - A.bb;
            dst_ptr[N] = tf_; dst_ptr[N + 1] = tf_;
            src_ptr[N] = tb_; src_ptr[N + 1] = tb_;
        }
        if (threadIdx.x <= R) {
            const int32_t rp = threadIdx.x < R ? (G > 0 ? A.S0[(int64_t)threadIdx.x * G] : 0) : P;
            meta[5 + threadIdx.x] = rp;
            if (rel_ptr_out != nullptr) {                                 // the same offsets for the device-side table builders: no upload
                rel_ptr_out[threadIdx.x] = rp;
                if (threadIdx.x == R) rel_ptr_out[R + 1] = P + (int32_t)N;
            }
        }
        // modes and the verdict ride in the same block: ONE device -> host copy per build (both are final before this launch)
        if (threadIdx.x < R) meta[5 + R + 1 + threadIdx.x] = mode[threadIdx.x];
        if (threadIdx.x == 0) {                                           // (+ the two fold verdicts: "still valid"; the verdict launch's ticket)
            meta[5 + 2 * R + 1] = *bad; meta[5 + 2 * R + 2] = 3; meta[5 + 2 * R + 3] = 3;    // (bit 0: graphs as tiles, bit 1: chunked tiles)
            meta[5 + 2 * R + 4 + dn_internal::kRilPlanWords] = 0;
            meta[5 + 2 * R + 4 + 12] = bad[1]; meta[5 + 2 * R + 4 + 13] = bad[2];   // the largest graph: nodes, edges
        }
    }
    __shared__ int take;                                                 // (graphs of a workgroup's share one at a time: see ril_stats_kernel)
    for (int64_t base = (int64_t)blockIdx.x * kLocShare; base < G; base += (int64_t)gridDim.x * kLocShare) {
      if (threadIdx.x == 0) take = 0;
      __syncthreads();
      for (;;) {
        int k = 0;
        if (lane == 0) k = atomicAdd(&take, 1);
        k = __shfl(k, 0, 64);
        const int64_t g = base + k;
        if (k >= kLocShare || g >= G) break;
        int n0, n1, e0;
        const int m = load_graph(X, lane, g, R, node_ptr, edge_ptr, src, dst, etype, s_mode, hbits, N, E, n0, n1, e0, nullptr);
        if (m < 0) continue;
        for (int v = n0 + lane; v < n1; v += 64) {
            const int32_t pf = A.Sf[v] - A.bf, pb = A.Sb[v] - A.bb;
            dst_ptr[v] = pf;
            src_ptr[v] = pb;
            if (self_loop) {
                row_in[P + v] = v; row_out[P + v] = v;
                dst_rows[A.Sf[v + 1] - A.bf - 1] = P + v;                  // the self loop closes every list
                src_rows[A.Sb[v + 1] - A.bb - 1] = P + v;
            }
        }
        if (m > 0 && fast_fits(m, n1 - n0, R)) {                           // (wave-uniform) the bit-set pass: no pair work
            const int nc = (m + 63) >> 6;
            if (nc == 1) fill_fast<1, 1>(X, ftab[wave], lane, m, n1 - n0, R, g, n0, A);
            else if (nc == 2) fill_fast<2, 1>(X, ftab[wave], lane, m, n1 - n0, R, g, n0, A);
            else fill_fast<4, 2>(X, ftab[wave], lane, m, n1 - n0, R, g, n0, A);
            continue;
        }
        if (m > 0) {                                                       // the sort pass (A, B, rsv over the bit-set table's words)
            static_assert(2 * kLocM + kLocR + 1 <= kFastWords, "sort arrays");
            u32* T = ftab[wave];
            fill_sorted<1, 10>(X, T, T + kLocM, reinterpret_cast<int*>(T + 2 * kLocM), nullptr, lane, m, R, g, n0, A, s_mode);
        }
        __builtin_amdgcn_wave_barrier();
      }
      __syncthreads();
    }
}

// The fill pass for the listed graphs (kLocM < edges < kBigM): one workgroup of kBigWaves wavefronts per graph, the sort pass with
// 13-bit edge numbers.
__global__ __launch_bounds__(kBigWaves * 64) void ril_fill_big_kernel(
    int64_t G, int64_t N, int32_t R, int32_t self_loop, const int32_t* __restrict__ node_ptr, const int32_t* __restrict__ edge_ptr,
    const int32_t* __restrict__ src, const int32_t* __restrict__ dst, const int32_t* __restrict__ etype,
    const int32_t* __restrict__ mode, const uint8_t* __restrict__ hbits, const int32_t* __restrict__ S, int32_t* __restrict__ row_in,
    int32_t* __restrict__ row_out, int32_t* __restrict__ aux_f_ptr, int32_t* __restrict__ aux_f_idx, int32_t* __restrict__ aux_b_ptr,
    int32_t* __restrict__ aux_b_idx, int32_t* __restrict__ dst_ptr, int32_t* __restrict__ dst_rows, int32_t* __restrict__ src_ptr,
    int32_t* __restrict__ src_rows, const int32_t* __restrict__ big_list) {
    __shared__ __attribute__((aligned(16))) uint2 X[kBigM];
    __shared__ __attribute__((aligned(16))) u32 Aw[kBigM];
    __shared__ __attribute__((aligned(16))) u32 Bw[kBigM];
    __shared__ int rsv[kLocR + 1];
    __shared__ int scratch[kBigWaves + 1];
    __shared__ int32_t s_mode[kLocR];
    const int nbig = big_list[0];
    if ((int)blockIdx.x >= nbig) return;
    const int tid = threadIdx.x;
    if (tid < kLocR) s_mode[tid] = tid < R ? mode[tid] : kEdge;
    __syncthreads();
    const int64_t RG = (int64_t)R * G;
    FillArgs A;
    A.G = G; A.N = N; A.self_loop = self_loop ? 1 : 0;
    A.S0 = S; A.S1 = S + RG; A.S2 = S + 2 * RG; A.S3 = S + 3 * RG; A.S4 = S + 4 * RG; A.Sf = S + kSeg * RG; A.Sb = A.Sf + (N + 1);
    A.b1 = A.S1[0]; A.b2 = A.S2[0]; A.b3 = A.S3[0]; A.b4 = A.S4[0]; A.bf = A.Sf[0]; A.bb = A.Sb[0];
    A.row_in = row_in; A.row_out = row_out; A.aux_f_ptr = aux_f_ptr; A.aux_f_idx = aux_f_idx; A.aux_b_ptr = aux_b_ptr;
    A.aux_b_idx = aux_b_idx; A.dst_rows = dst_rows; A.src_rows = src_rows;
    const int32_t P = A.b1;
    for (int k = blockIdx.x; k < nbig; k += gridDim.x) {
        const int64_t g = big_list[1 + k];
        const int n0 = node_ptr[g], n1 = node_ptr[g + 1], e0 = edge_ptr[g], m = edge_ptr[g + 1] - e0;
        bool oob = false;
        for (int i = tid; i < m; i += kBigWaves * 64) {
            const int r = etype[e0 + i], sv = src[e0 + i], d = dst[e0 + i];
            const bool o = (sv < n0) | (sv >= n1) | (d < n0) | (d >= n1) | (r < 0) | (r >= R);
            oob |= o;
            if (!o) {
                const u32 sl = (u32)(sv - n0), dl = (u32)(d - n0), md = (u32)s_mode[r], hb = hbits[e0 + i];
                const u32 head = ((md == kTf ? hb >> 1 : hb) & 1u) ^ 1u;
                X[i] = make_uint2(((u32)r << 26) | ((md == kTf ? sl : dl) << 13) | (u32)i, (head << 30) | (md << 28) | (sl << 14) | dl);
            }
        }
        if (__syncthreads_or(oob ? 1 : 0)) continue;                      // (flagged by the statistics pass: the tables stay unused)
        for (int v = n0 + tid; v < n1; v += kBigWaves * 64) {
            dst_ptr[v] = A.Sf[v] - A.bf;
            src_ptr[v] = A.Sb[v] - A.bb;
            if (self_loop) {
                row_in[P + v] = v; row_out[P + v] = v;
                dst_rows[A.Sf[v + 1] - A.bf - 1] = P + v;
                src_rows[A.Sb[v + 1] - A.bb - 1] = P + v;
            }
        }
        fill_sorted<kBigWaves, 13>(X, Aw, Bw, rsv, scratch, tid, m, R, g, n0, A, s_mode);
    }
}

// Can the closing launch ABSORB the fold of this batch (dn_rows_close_bf16 with AGG units)?  The question ops._closing_tables
// used to ask with two more launches and a second read-back, answered here behind ril_fill_kernel with what the device already
// knows (meta): the candidate is the ONE collapsed relation of the direction -- mode AGG forward (rows into few destinations: u ->
// dummy), TF backward -- with rows, all of the direction's aux lists its own; its segments (the aux lists) must tile the batch as
// dn_fold_graph_tile_one checks, the target of segment j being the relation's j-th row's output (forward) / input (backward) node.
// tile_ptr [G + 1], info [G][12] as dn_fold_graph_tiles_build_i32 writes them; verdict -> meta[5 + 2 R + 2 + direction].
struct VdDir {
    const int32_t *aux_ptr, *aux_idx, *row_target;
    int32_t *tile_ptr, *info;
};
struct VdPair {
    VdDir d[2];
};

// What the table builders behind the row index need from its counts, left ON THE DEVICE so that dn_conv_index_build_i32 can queue
// them without waiting for the read-back: plan = meta + 5 + 2 R + 4, kRilPlanWords words
//   [0..3] forward closing stream:  {edge rows P, first / end row of the folded relation (the rows the stream leaves out), go:
//          1 = the graphs are the tiles, 2 = chunked tiles over graphs of any size}
//   [4..7] backward closing stream: the same
//   [8..9] forward sweep order: {relation to skip (-1: none), go};  [10..11] backward sweep order
// go = the build is valid (no graph raised the flag), the direction's fold can be absorbed (the verdicts below) and its segments
// are the batch's G graphs -- the one case the queued builders are sized for; otherwise they do nothing.
__device__ void ril_plan(int64_t G, int32_t R, int32_t* __restrict__ meta) {
    int32_t* plan = meta + 5 + 2 * R + 4;
    const int32_t P = meta[0], status = meta[5 + 2 * R + 1];
    for (int d = 0; d < 2; ++d) {
        const int want = d == 0 ? kAgg : kTf;
        int rel = -1;
        for (int r = 0; r < R; ++r)
            if (meta[5 + R + 1 + r] == want && meta[5 + r + 1] > meta[5 + r]) rel = r;   // (exactly one when the verdict stands)
        const int32_t verdict = __hip_atomic_load(meta + 5 + 2 * R + 2 + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // go: 0 nothing, 1 the graphs are the tiles (every block within 32 nodes), 2 chunked tiles (graphs of any size)
        const int32_t go = (status == 0 && verdict != 0 && rel >= 0 && (int64_t)meta[1 + d] == G) ? ((verdict & 1) ? 1 : 2) : 0;
        plan[4 * d] = P;
        plan[4 * d + 1] = go ? meta[5 + rel] : 0;
        plan[4 * d + 2] = go ? meta[5 + rel + 1] : 0;
        plan[4 * d + 3] = go;
        plan[8 + 2 * d] = go ? rel : -1;
        plan[8 + 2 * d + 1] = go ? 1 : 0;
    }
}

// Can the closing launch ABSORB the fold of this batch (dn_rows_close_bf16 with AGG units)?  The question ops._closing_tables
// used to ask with two more launches and a second read-back, answered here behind ril_fill_kernel with what the device already
// knows (meta), for both directions in one launch (blockIdx.y): the candidate is the ONE collapsed relation of the direction --
// mode AGG forward (rows into few destinations: u -> dummy), TF backward -- with rows, all of the direction's aux lists its own;
// its segments (the aux lists) must tile the batch as dn_fold_graph_tile_one checks, the target of segment j being the relation's
// j-th row's output (forward) / input (backward) node.  tile_ptr [G + 1], info [G][12] as dn_fold_graph_tiles_build_i32 writes
// them; verdict -> meta[5 + 2 R + 2 + direction].  want_plan: the workgroup that finishes LAST (a counter behind the plan words)
// writes ril_plan's words -- every verdict has landed by then.
__global__ void ril_fold_verdict_kernel(int64_t G, int64_t N, int32_t R, int32_t self_loop, int32_t* __restrict__ meta, VdPair pr,
                                        int32_t want_plan) {
    const int direction = blockIdx.y;
    const VdDir& a = pr.d[direction];
    const int want = direction == 0 ? kAgg : kTf;
    const int32_t n_aux = meta[1 + direction];
    int32_t* ok = meta + 5 + 2 * R + 2 + direction;
    int found = 0, rel = 0;
    for (int r = 0; r < R; ++r)
        if (meta[5 + R + 1 + r] == want && meta[5 + r + 1] > meta[5 + r]) { ++found; rel = r; }
    const int32_t beg = meta[5 + rel], end = meta[5 + rel + 1];
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (!(self_loop && found == 1 && n_aux > 0 && end - beg == n_aux && (int64_t)n_aux <= G)) {
        if (j == 0) {
            __hip_atomic_store(ok, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    } else {                                                               // (the same branch for the whole launch: full wavefronts inside)
        dn_fold_verdicts_one(j <= n_aux, j, (int32_t)N, n_aux, a.aux_ptr, a.aux_idx, a.row_target + beg, a.tile_ptr, a.info, ok);
    }
    if (!want_plan) return;
    // (the verdict words are written and read with device-scope atomics -- no cache write-back per workgroup; everything else the
    //  plan reads was written by the previous launch)
    __syncthreads();                                                       // (every lane's verdict store has completed: it waited)
    if (threadIdx.x == 0) {
        int32_t* ticket = meta + 5 + 2 * R + 4 + dn_internal::kRilPlanWords;
        // Release side: every verdict store of this workgroup is a device-scope atomic that has COMPLETED (s_waitcnt vmcnt(0) in
        // the storing lane, then the barrier above) before the ticket moves -- the hardware form of a release without the per-
        // workgroup L2 write-back an agent-scope release fence costs here (20 -> 52 us for the launch).  Acquire side:
        // only the LAST workgroup pays for a real fence before it reads the other workgroups' words.
        if (__hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int32_t)(gridDim.x * gridDim.y) - 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            ril_plan(G, R, meta);
        }
    }
}

struct LocWs {
    int32_t *Er, *Dr, *Sr, *mode, *bad, *meta, *C, *S, *big;
    uint8_t* hbits;
    void* scan_tmp;
    size_t scan_tmp_bytes, zero_bytes;
    int64_t L;
};

int loc_layout(char* base, size_t cap, size_t& need, LocWs& w, int64_t G, int64_t N, int64_t R, int64_t E) {
    size_t off = 0;
    auto take = [&](size_t bytes) -> char* {
        off = dn_align_up(off, 256);
        char* p = base ? base + off : nullptr;
        off += bytes > 0 ? bytes : 1;
        return p;
    };
    w.L = kSeg * R * G + 2 * (N + 1) + 1;
    // zeroed in one memset: Er, Dr, Sr, bad, then the packed counts
    char* z0 = take(sizeof(int32_t) * (size_t)(3 * kLocR + 64));
    w.Er = (int32_t*)z0; w.Dr = w.Er + kLocR; w.Sr = w.Dr + kLocR; w.bad = w.Sr + kLocR;
    w.C = (int32_t*)take(sizeof(int32_t) * (size_t)w.L);
    w.zero_bytes = base ? (size_t)((char*)(w.C + w.L) - z0) : 0;
    w.S = (int32_t*)take(sizeof(int32_t) * (size_t)w.L);
    w.mode = (int32_t*)take(sizeof(int32_t) * kLocR);
    w.hbits = (uint8_t*)take((size_t)E);
    w.big = (int32_t*)take(sizeof(int32_t) * (size_t)(G + 2));            // [0] the number of listed graphs, then the graphs
    w.meta = (int32_t*)take(sizeof(int32_t) * (size_t)(5 + 2 * kLocR + 4 + dn_internal::kRilPlanWords + 1 + 4));   // (+ the verdict launch's ticket, 2 x 2 words of the sweep builder)
    w.scan_tmp_bytes = 0;
    if (rocprim::exclusive_scan(nullptr, w.scan_tmp_bytes, (const int32_t*)nullptr, (int32_t*)nullptr, (int32_t)0, (size_t)w.L,
                                rocprim::plus<int32_t>(), (hipStream_t)0) != hipSuccess) {
        dn_set_error("dn_row_index_build_local: scan size query failed");
        return DN_ERR_HIP;
    }
    w.scan_tmp = take(w.scan_tmp_bytes);
    need = off + 256;
    (void)cap;
    return DN_OK;
}

}  // namespace

namespace dn_internal {

// The launches of dn_row_index_build_local_i32 without its read-back: *meta_dev = the device words the host unpacks afterwards
// (ril_unpack); plan: ril_plan's words behind them (needs verdicts).
int ril_queue(int64_t G, int64_t N, int64_t R, int64_t E, const int32_t* node_ptr, const int32_t* edge_ptr, const int32_t* src,
              const int32_t* dst, const int32_t* etype, int32_t self_loop, float edge_frac, int32_t* row_in, int32_t* row_out,
              int32_t* aux_f_ptr, int32_t* aux_f_idx, int32_t* aux_b_ptr, int32_t* aux_b_idx, int32_t* dst_ptr, int32_t* dst_rows,
              int32_t* src_ptr, int32_t* src_rows, int32_t* rel_ptr_dev, int32_t* tile_ptr_f, int32_t* fold_info_f,
              int32_t* tile_ptr_b, int32_t* fold_info_b, bool verdicts, bool plan, void* workspace, size_t workspace_bytes,
              int32_t** meta_dev, hipStream_t st) {
    DN_REQUIRE(G >= 0 && N >= 0 && R >= 1 && E >= 0, "dn_row_index_build_local: bad sizes");
    DN_REQUIRE(R <= kLocR, "dn_row_index_build_local: more than 64 relations (use dn_row_index_build_i32)");
    DN_REQUIRE(2 * E + N < 0x7fffffffLL && kSeg * R * G + 2 * (N + 1) + 1 < 0x7fffffffLL,
               "dn_row_index_build_local: sizes must fit int32");
    DN_REQUIRE(node_ptr && edge_ptr && row_in && row_out && aux_f_ptr && aux_b_ptr && dst_ptr && dst_rows && src_ptr && src_rows &&
               workspace && meta_dev, "dn_row_index_build_local: NULL pointer");
    DN_REQUIRE(E == 0 || (src && dst && etype && aux_f_idx && aux_b_idx), "dn_row_index_build_local: NULL pointer");
    LocWs w;
    size_t need = 0;
    int rc = loc_layout((char*)workspace, workspace_bytes, need, w, G, N, R, E);
    if (rc != DN_OK) return rc;
    if (need > workspace_bytes) { dn_set_error("dn_row_index_build_local: workspace too small (%zu < %zu)", workspace_bytes, need); return DN_ERR_WORKSPACE; }
    DN_CHECK_HIP(hipMemsetAsync(w.Er, 0, w.zero_bytes, st));
    DN_CHECK_HIP(hipMemsetAsync(w.big, 0, sizeof(int32_t), st));
    const unsigned grid = (unsigned)(G > 0 ? (dn_cdiv(G, kLocWaves) < 2048 ? dn_cdiv(G, kLocWaves) : 2048) : 1);
    const unsigned sgrid = (unsigned)(G > 0 ? (dn_cdiv(G, kLocShare) < 2048 ? dn_cdiv(G, kLocShare) : 2048) : 1);   // (stats / fill: shares)
    hipLaunchKernelGGL(ril_stats_kernel, dim3(sgrid), dim3(kLocWaves * 64), 0, st, G, (int32_t)R, node_ptr, edge_ptr, src, dst, etype,
                       N, E, w.Er, w.Dr, w.Sr, w.bad, w.hbits, w.big);
    // (the graphs over one wavefront's LDS slice, if any: the list's length is only known on the device -- an empty list costs two
    //  empty launches)
    const bool may_big = E > kLocM;
    if (may_big)
        hipLaunchKernelGGL(ril_stats_big_kernel, dim3(kBigGrid), dim3(kBigWaves * 64), 0, st, (int32_t)R, node_ptr, edge_ptr, src, dst, etype,
                           w.Er, w.Dr, w.Sr, w.bad, w.hbits, w.big);
    hipLaunchKernelGGL(ril_count_kernel, dim3(grid), dim3(kLocWaves * 64), 0, st, G, N, E, (int32_t)R, edge_frac, self_loop, node_ptr,
                       edge_ptr, src, dst, etype, w.Er, w.Dr, w.Sr, w.hbits, w.mode, w.C);
    DN_CHECK_HIP(rocprim::exclusive_scan(w.scan_tmp, w.scan_tmp_bytes, (const int32_t*)w.C, w.S, (int32_t)0, (size_t)w.L,
                                         rocprim::plus<int32_t>(), st));
    hipLaunchKernelGGL(ril_fill_kernel, dim3(sgrid), dim3(kLocWaves * 64), 0, st, G, N, E, (int32_t)R, self_loop, node_ptr, edge_ptr, src,
                       dst, etype, w.mode, w.hbits, w.S, row_in, row_out, aux_f_ptr, aux_f_idx, aux_b_ptr, aux_b_idx, dst_ptr, dst_rows,
                       src_ptr, src_rows, w.meta, w.bad, rel_ptr_dev);
    if (may_big)
        hipLaunchKernelGGL(ril_fill_big_kernel, dim3(kBigGrid), dim3(kBigWaves * 64), 0, st, G, N, (int32_t)R, self_loop, node_ptr, edge_ptr,
                           src, dst, etype, w.mode, w.hbits, w.S, row_in, row_out, aux_f_ptr, aux_f_idx, aux_b_ptr, aux_b_idx, dst_ptr,
                           dst_rows, src_ptr, src_rows, w.big);
    DN_CHECK_LAUNCH();
    if (verdicts) {
        DN_REQUIRE(tile_ptr_f && fold_info_f && tile_ptr_b && fold_info_b, "dn_row_index_build_local: the fold verdicts need the four tile buffers");
        DN_REQUIRE((reinterpret_cast<uintptr_t>(fold_info_f) | reinterpret_cast<uintptr_t>(fold_info_b)) % 16 == 0,
                   "dn_row_index_build_local: unaligned fold_info");
        const unsigned vg = (unsigned)dn_cdiv(G + 1, 256);
        VdPair pr;
        pr.d[0] = VdDir{aux_f_ptr, aux_f_idx, row_out, tile_ptr_f, fold_info_f};
        pr.d[1] = VdDir{aux_b_ptr, aux_b_idx, row_in, tile_ptr_b, fold_info_b};
        hipLaunchKernelGGL(ril_fold_verdict_kernel, dim3(vg, 2), dim3(256), 0, st, G, N, (int32_t)R, self_loop, w.meta, pr, plan ? 1 : 0);
        DN_CHECK_LAUNCH();
    }
    DN_REQUIRE(!plan || verdicts, "dn_row_index_build_local: the plan needs the fold verdicts");
    *meta_dev = w.meta;
    return DN_OK;
}

void ril_unpack(const int32_t* h_meta, int64_t R, int64_t* host_counts, int32_t* host_rel_ptr, int32_t* host_modes,
                int32_t* host_status, int32_t* host_absorb) {
    if (host_absorb) {
        host_absorb[0] = h_meta[5 + 2 * R + 2]; host_absorb[1] = h_meta[5 + 2 * R + 3];
        host_absorb[2] = h_meta[5 + 2 * R + 4 + 12]; host_absorb[3] = h_meta[5 + 2 * R + 4 + 13];
    }
    *host_status = h_meta[5 + 2 * R + 1];
    for (int k = 0; k < 5; ++k) host_counts[k] = h_meta[k];
    for (int64_t r = 0; r <= R; ++r) host_rel_ptr[r] = h_meta[5 + r];
    for (int64_t r = 0; r < R; ++r) host_modes[r] = h_meta[5 + R + 1 + r];
}

}  // namespace dn_internal

extern "C" {

size_t dn_row_index_local_workspace_bytes(int64_t G, int64_t N, int64_t R, int64_t E) {
    if (G < 0 || N < 0 || R < 1 || E < 0 || R > kLocR) { dn_set_error("dn_row_index_local_workspace_bytes: bad sizes"); return 0; }
    if (kSeg * R * G + 2 * (N + 1) + 1 >= 0x7fffffffLL) { dn_set_error("dn_row_index_local_workspace_bytes: too large"); return 0; }
    LocWs w;
    size_t need = 0;
    if (loc_layout(nullptr, 0, need, w, G, N, R, E) != DN_OK) return 0;
    return need;
}

int dn_row_index_build_local_i32(int64_t G, int64_t N, int64_t R, int64_t E, const int32_t* node_ptr, const int32_t* edge_ptr,
                                 const int32_t* src, const int32_t* dst, const int32_t* etype, int32_t self_loop, float edge_frac,
                                 int32_t* row_in, int32_t* row_out, int32_t* aux_f_ptr, int32_t* aux_f_idx, int32_t* aux_b_ptr,
                                 int32_t* aux_b_idx, int32_t* dst_ptr, int32_t* dst_rows, int32_t* src_ptr, int32_t* src_rows,
                                 int64_t* host_counts, int32_t* host_rel_ptr, int32_t* host_modes, int32_t* host_status,
                                 int32_t* rel_ptr_dev, int32_t* tile_ptr_f, int32_t* fold_info_f, int32_t* tile_ptr_b,
                                 int32_t* fold_info_b, int32_t* host_absorb, void* workspace, size_t workspace_bytes,
                                 dn_stream_t stream) {
    DN_REQUIRE(host_counts && host_rel_ptr && host_modes && host_status, "dn_row_index_build_local: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    int32_t* meta = nullptr;
    const int rc = dn_internal::ril_queue(G, N, R, E, node_ptr, edge_ptr, src, dst, etype, self_loop, edge_frac, row_in, row_out, aux_f_ptr,
                                          aux_f_idx, aux_b_ptr, aux_b_idx, dst_ptr, dst_rows, src_ptr, src_rows, rel_ptr_dev, tile_ptr_f,
                                          fold_info_f, tile_ptr_b, fold_info_b, host_absorb != nullptr, false, workspace, workspace_bytes,
                                          &meta, st);
    if (rc != DN_OK) return rc;
    int32_t h_meta[5 + 2 * kLocR + 4 + dn_internal::kRilPlanWords];
    DN_CHECK_HIP(hipMemcpyAsync(h_meta, meta, sizeof(int32_t) * (size_t)(5 + 2 * R + 4 + dn_internal::kRilPlanWords), hipMemcpyDeviceToHost, st));
    DN_CHECK_HIP(hipStreamSynchronize(st));
    dn_internal::ril_unpack(h_meta, R, host_counts, host_rel_ptr, host_modes, host_status, host_absorb);
    return DN_OK;
}

}  // extern "C"
