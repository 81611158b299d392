// Graph-local build of the row factorisation (dn_row_index_build_local_i32): the same tables as dn_row_index_build_i32,
// bit for bit, for a batch given as a disjoint union of graphs (node_ptr / edge_ptr: every graph owns a contiguous node range
// and a contiguous edge range -- what dgl.batch / the PyG collate produce, subgraph_isomorphism/dataset.py:1605-1611).
//
// The general builder orders the whole batch with three device-wide radix sorts, eleven scans and ~60 launches (1.4 ms for
// the 4 M edges of BASELINE config 5, a step's worth for small batches where every launch is latency).  In a batch of graphs
// every ordering it establishes is LOCAL: the stable sort by (relation, node) is graph-major inside a relation, because node
// ids ascend with the graph.  So here one wavefront takes one graph into LDS and RANK-sorts it (a few hundred edges: the
// O(m^2) comparisons are broadcast LDS reads), and the only batch-wide step is ONE exclusive scan over the packed per-(relation,
// graph) and per-node counts.  Three kernels + one scan + one read-back.
//
//   ril_stats_kernel   per relation: #edges, #distinct destinations, #distinct sources      (-> EDGE / AGG / TF, as before)
//   ril_count_kernel   per (relation, graph): rows, AGG rows, TF rows, AGG edges, TF edges;  per node: list lengths
//   (exclusive scan of the packed counts)
//   ril_fill_kernel    every table, each entry at  scan offset + rank inside the graph
//
// Graphs with more than kLocM edges, an endpoint outside the graph's node range or a relation id outside [0, R) raise a flag
// instead (host_status = 1): the caller then runs the general builder.
#include <cstring>
#include <cstdlib>

#include "dn_common.h"
#include "../../include/dn_hip.h"

#include <rocprim/rocprim.hpp>

namespace {

constexpr int kLocM = 1024;          // edges of one graph held in LDS
constexpr int kLocWaves = 4;         // graphs in flight per workgroup (one wavefront each)
constexpr int kLocR = 64;            // relations (per-workgroup counters, one lane per relation)
constexpr int kEdge = 0, kAgg = 1, kTf = 2;
constexpr int kSeg = 5;              // packed count arrays over (relation, graph): rows, AGG rows, TF rows, AGG edges, TF edges

struct LocLds {
    int32_t rel[kLocWaves][kLocM];   // relation | mode << 8 | head << 10
    int32_t src[kLocWaves][kLocM];
    int32_t dst[kLocWaves][kLocM];
};

__device__ __forceinline__ int mode_of(int32_t Er, int32_t Dr, int32_t Sr, float edge_frac) {
    const int mn = Dr < Sr ? Dr : Sr;
    if (Er == 0 || (float)mn > edge_frac * (float)Er) return kEdge;       // (dn_index.hip: ri_mode_kernel)
    return Dr <= Sr ? kAgg : kTf;
}

// edges of graph g -> LDS (this wavefront's slice).  Returns the edge count, or -1 when the graph is not taken.
__device__ __forceinline__ int load_graph(LocLds& L, int wave, int lane, int64_t g, int32_t R, const int32_t* node_ptr,
                                          const int32_t* edge_ptr, const int32_t* src, const int32_t* dst, const int32_t* etype,
                                          int& n0, int& n1, int& e0, int32_t* bad) {
    n0 = node_ptr[g]; n1 = node_ptr[g + 1]; e0 = edge_ptr[g];
    const int m = edge_ptr[g + 1] - e0;
    if (m < 0 || m > kLocM || n1 < n0) {
        if (bad != nullptr && lane == 0) atomicOr(bad, 1);
        return -1;
    }
    bool oob = false;
    for (int i = lane; i < m; i += 64) {
        const int r = etype[e0 + i], s = src[e0 + i], d = dst[e0 + i];
        L.rel[wave][i] = r; L.src[wave][i] = s; L.dst[wave][i] = d;
        oob |= (s < n0) | (s >= n1) | (d < n0) | (d >= n1) | (r < 0) | (r >= R);
    }
    if (__any(oob)) {
        if (bad != nullptr && lane == 0) atomicOr(bad, 1);
        return -1;
    }
    return m;
}

// relation word of every edge of the graph: relation | mode << 8 | head << 10   (head: first edge, in edge order, of its
// (relation, key node) pair; key node = source in a TF relation, destination otherwise)
__device__ __forceinline__ void flag_graph(LocLds& L, int wave, int lane, int m, const int32_t* s_mode) {
    int32_t w[kLocM / 64];
#pragma unroll
    for (int k = 0; k < kLocM / 64; ++k) {
        const int i = lane + 64 * k;
        if (i >= m) break;
        const int r = L.rel[wave][i], md = s_mode[r];
        const int kn = md == kTf ? L.src[wave][i] : L.dst[wave][i];
        bool head = true;
        for (int j = 0; j < i; ++j) {
            const int rj = L.rel[wave][j];                                   // (still the plain relation id: written below)
            if (rj == r) head &= (md == kTf ? L.src[wave][j] : L.dst[wave][j]) != kn;
        }
        w[k] = r | (md << 8) | ((head ? 1 : 0) << 10);
    }
    // every lane has finished READING the plain ids (the loops above run in lockstep inside the wavefront) before any writes
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < kLocM / 64; ++k) {
        const int i = lane + 64 * k;
        if (i >= m) break;
        L.rel[wave][i] = w[k];
    }
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(kLocWaves * 64) void ril_stats_kernel(int64_t G, int32_t R, const int32_t* __restrict__ node_ptr,
                                                                   const int32_t* __restrict__ edge_ptr,
                                                                   const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                                                   const int32_t* __restrict__ etype, int64_t N, int64_t E,
                                                                   int32_t* Er, int32_t* Dr, int32_t* Sr, int32_t* bad) {
    __shared__ LocLds L;
    __shared__ int32_t cnt[3][kLocR];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 3 * kLocR) cnt[threadIdx.x / kLocR][threadIdx.x % kLocR] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0 &&                           // the graphs must tile the node and edge ranges
        (node_ptr[0] != 0 || edge_ptr[0] != 0 || node_ptr[G] != N || edge_ptr[G] != E))
        atomicOr(bad, 1);
    __syncthreads();
    for (int64_t g = (int64_t)blockIdx.x * kLocWaves + wave; g < G; g += (int64_t)gridDim.x * kLocWaves) {
        int n0, n1, e0;
        const int m = load_graph(L, wave, lane, g, R, node_ptr, edge_ptr, src, dst, etype, n0, n1, e0, bad);
        if (m <= 0) continue;
        for (int i = lane; i < m; i += 64) {
            const int r = L.rel[wave][i], s = L.src[wave][i], d = L.dst[wave][i];
            bool fd = true, fs = true;
            for (int j = 0; j < i; ++j) {
                if (L.rel[wave][j] == r) { fd &= L.dst[wave][j] != d; fs &= L.src[wave][j] != s; }
            }
            atomicAdd(&cnt[0][r], 1);
            if (fd) atomicAdd(&cnt[1][r], 1);
            if (fs) atomicAdd(&cnt[2][r], 1);
        }
        __builtin_amdgcn_wave_barrier();                                   // the next graph overwrites this wavefront's slice
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
__global__ __launch_bounds__(kLocWaves * 64) void ril_count_kernel(int64_t G, int64_t N, int32_t R, float edge_frac, int32_t self_loop,
                                                                   const int32_t* __restrict__ node_ptr,
                                                                   const int32_t* __restrict__ edge_ptr,
                                                                   const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                                                   const int32_t* __restrict__ etype, const int32_t* __restrict__ Er,
                                                                   const int32_t* __restrict__ Dr, const int32_t* __restrict__ Sr,
                                                                   int32_t* mode_out, int32_t* __restrict__ C) {
    __shared__ LocLds L;
    __shared__ int32_t s_mode[kLocR];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < kLocR) {
        const int md = threadIdx.x < R ? mode_of(Er[threadIdx.x], Dr[threadIdx.x], Sr[threadIdx.x], edge_frac) : kEdge;
        s_mode[threadIdx.x] = md;
        if (blockIdx.x == 0 && threadIdx.x < R) mode_out[threadIdx.x] = md;
    }
    __syncthreads();
    const int64_t RG = (int64_t)R * G;
    int32_t* Cf = C + kSeg * RG;
    int32_t* Cb = Cf + (N + 1);
    for (int64_t g = (int64_t)blockIdx.x * kLocWaves + wave; g < G; g += (int64_t)gridDim.x * kLocWaves) {
        int n0, n1, e0;
        const int m = load_graph(L, wave, lane, g, R, node_ptr, edge_ptr, src, dst, etype, n0, n1, e0, nullptr);
        if (m < 0) continue;                                               // (flagged by the statistics pass: tables are unused)
        flag_graph(L, wave, lane, m, s_mode);
        if (lane < R) {                                                    // one lane per relation
            int rows = 0, heads = 0, edges = 0;
            for (int j = 0; j < m; ++j) {
                const int w = L.rel[wave][j];
                if ((w & 0xff) == lane) { ++edges; heads += (w >> 10) & 1; }
            }
            const int md = s_mode[lane];
            rows = md == kEdge ? edges : heads;
            const int64_t at = (int64_t)lane * G + g;
            C[at] = rows;
            C[RG + at] = md == kAgg ? heads : 0;
            C[2 * RG + at] = md == kTf ? heads : 0;
            C[3 * RG + at] = md == kAgg ? edges : 0;
            C[4 * RG + at] = md == kTf ? edges : 0;
        }
        for (int v = n0 + lane; v < n1; v += 64) {                         // list lengths of the graph's nodes
            int f = self_loop ? 1 : 0, b = f;
            for (int j = 0; j < m; ++j) {
                const int w = L.rel[wave][j], md = (w >> 8) & 3, hd = (w >> 10) & 1;
                if (L.dst[wave][j] == v) f += (md != kAgg) ? 1 : hd;       // per-edge entry, or the one entry of an AGG row
                if (L.src[wave][j] == v) b += (md != kTf) ? 1 : hd;
            }
            Cf[v] = f;
            Cb[v] = b;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ __launch_bounds__(kLocWaves * 64) void ril_fill_kernel(
    int64_t G, int64_t N, int32_t R, int32_t self_loop, const int32_t* __restrict__ node_ptr, const int32_t* __restrict__ edge_ptr,
    const int32_t* __restrict__ src, const int32_t* __restrict__ dst, const int32_t* __restrict__ etype,
    const int32_t* __restrict__ mode, const int32_t* __restrict__ S, int32_t* __restrict__ row_in, int32_t* __restrict__ row_out,
    int32_t* __restrict__ aux_f_ptr, int32_t* __restrict__ aux_f_idx, int32_t* __restrict__ aux_b_ptr,
    int32_t* __restrict__ aux_b_idx, int32_t* __restrict__ dst_ptr, int32_t* __restrict__ dst_rows, int32_t* __restrict__ src_ptr,
    int32_t* __restrict__ src_rows, int32_t* __restrict__ meta /* [5] totals, [R + 1] rel_ptr */) {
    __shared__ LocLds L;
    __shared__ int32_t s_mode[kLocR];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < kLocR) s_mode[threadIdx.x] = threadIdx.x < R ? mode[threadIdx.x] : kEdge;
    __syncthreads();
    const int64_t RG = (int64_t)R * G;
    const int32_t* S0 = S;
    const int32_t* S1 = S + RG;
    const int32_t* S2 = S + 2 * RG;
    const int32_t* S3 = S + 3 * RG;
    const int32_t* S4 = S + 4 * RG;
    const int32_t* Sf = S + kSeg * RG;
    const int32_t* Sb = Sf + (N + 1);
    const int32_t b1 = S1[0], b2 = S2[0], b3 = S3[0], b4 = S4[0], bf = Sf[0], bb = Sb[0];      // (S0[0] == 0)
    const int32_t P = b1, n_agg = b2 - b1, n_tf = b3 - b2, n_agg_e = b4 - b3, n_tf_e = bf - b4;
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) {
            meta[0] = P; meta[1] = n_agg; meta[2] = n_tf; meta[3] = n_agg_e; meta[4] = n_tf_e;
            aux_f_ptr[n_agg] = n_agg_e;
            aux_b_ptr[n_tf] = n_tf_e;
            const int32_t tf_ = Sb[0] - bf, tb_ = Sb[N + 1] - bb;
            dst_ptr[N] = tf_; dst_ptr[N + 1] = tf_;
            src_ptr[N] = tb_; src_ptr[N + 1] = tb_;
        }
        if (threadIdx.x <= R) meta[5 + threadIdx.x] = threadIdx.x < R ? (G > 0 ? S0[(int64_t)threadIdx.x * G] : 0) : P;
    }
    for (int64_t g = (int64_t)blockIdx.x * kLocWaves + wave; g < G; g += (int64_t)gridDim.x * kLocWaves) {
        int n0, n1, e0;
        const int m = load_graph(L, wave, lane, g, R, node_ptr, edge_ptr, src, dst, etype, n0, n1, e0, nullptr);
        if (m < 0) continue;
        flag_graph(L, wave, lane, m, s_mode);
        for (int v = n0 + lane; v < n1; v += 64) {
            const int32_t pf = Sf[v] - bf, pb = Sb[v] - bb;
            dst_ptr[v] = pf;
            src_ptr[v] = pb;
            if (self_loop) {
                row_in[P + v] = v; row_out[P + v] = v;
                dst_rows[Sf[v + 1] - bf - 1] = P + v;                      // the self loop closes every list
                src_rows[Sb[v + 1] - bb - 1] = P + v;
            }
        }
        for (int i = lane; i < m; i += 64) {
            const int w = L.rel[wave][i], r = w & 0xff, md = (w >> 8) & 3, hd = (w >> 10) & 1;
            const int s = L.src[wave][i], d = L.dst[wave][i], kn = md == kTf ? s : d;
            const bool collapsed_head = hd && md != kEdge;
            int rank_rel = 0, heads_before = 0, rank_d = 0, rank_s = 0, others = 0, same_before = 0;
            for (int j = 0; j < m; ++j) {
                const int wj = L.rel[wave][j], rj = wj & 0xff, mdj = (wj >> 8) & 3, hdj = (wj >> 10) & 1;
                const int sj = L.src[wave][j], dj = L.dst[wave][j], knj = mdj == kTf ? sj : dj;
                const bool lt_in_rel = knj < kn || (knj == kn && j < i);
                const bool lt = rj < r || (rj == r && lt_in_rel);           // position in the (relation, key node, edge) order
                if (rj == r) { rank_rel += lt_in_rel; heads_before += hdj & (knj < kn ? 1 : 0); }
                if (dj == d && mdj != kAgg) rank_d += lt;                    // per-edge forward entries of node d before mine
                if (sj == s && mdj != kTf) rank_s += lt;
                if (collapsed_head) {
                    // entries of node kn's list that precede this collapsed row's entry: every per-edge entry, and the
                    // collapsed rows of lower relations (row order)
                    if (md == kAgg) { others += (dj == kn && mdj != kAgg); same_before += (hdj && mdj == kAgg && knj == kn && rj < r); }
                    else            { others += (sj == kn && mdj != kTf);  same_before += (hdj && mdj == kTf && knj == kn && rj < r); }
                }
            }
            const int64_t at = (int64_t)r * G + g;
            const int32_t row = S0[at] + (md == kEdge ? rank_rel : heads_before);
            if (md == kEdge) {
                row_in[row] = s; row_out[row] = d;
            } else if (md == kAgg) {
                const int32_t pos = S3[at] - b3 + rank_rel;                 // among the AGG edges, (relation, dst, edge) order
                aux_f_idx[pos] = s;
                if (hd) {
                    const int32_t a = S1[at] - b1 + heads_before;
                    row_in[row] = (int32_t)N + a; row_out[row] = kn;
                    aux_f_ptr[a] = pos;
                    dst_rows[Sf[kn] - bf + others + same_before] = row;
                }
            } else {
                const int32_t pos = S4[at] - b4 + rank_rel;
                aux_b_idx[pos] = d;
                if (hd) {
                    const int32_t a = S2[at] - b2 + heads_before;
                    row_in[row] = kn; row_out[row] = (int32_t)N + a;
                    aux_b_ptr[a] = pos;
                    src_rows[Sb[kn] - bb + others + same_before] = row;
                }
            }
            if (md != kAgg) dst_rows[Sf[d] - bf + rank_d] = row;
            if (md != kTf) src_rows[Sb[s] - bb + rank_s] = row;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

struct LocWs {
    int32_t *Er, *Dr, *Sr, *mode, *bad, *meta, *C, *S;
    void* scan_tmp;
    size_t scan_tmp_bytes, zero_bytes;
    int64_t L;
};

int loc_layout(char* base, size_t cap, size_t& need, LocWs& w, int64_t G, int64_t N, int64_t R) {
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
    w.meta = (int32_t*)take(sizeof(int32_t) * (size_t)(5 + kLocR + 1));
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

extern "C" {

size_t dn_row_index_local_workspace_bytes(int64_t G, int64_t N, int64_t R, int64_t E) {
    if (G < 0 || N < 0 || R < 1 || E < 0 || R > kLocR) { dn_set_error("dn_row_index_local_workspace_bytes: bad sizes"); return 0; }
    if (kSeg * R * G + 2 * (N + 1) + 1 >= 0x7fffffffLL) { dn_set_error("dn_row_index_local_workspace_bytes: too large"); return 0; }
    LocWs w;
    size_t need = 0;
    if (loc_layout(nullptr, 0, need, w, G, N, R) != DN_OK) return 0;
    return need;
}

int dn_row_index_build_local_i32(int64_t G, int64_t N, int64_t R, int64_t E, const int32_t* node_ptr, const int32_t* edge_ptr,
                                 const int32_t* src, const int32_t* dst, const int32_t* etype, int32_t self_loop, float edge_frac,
                                 int32_t* row_in, int32_t* row_out, int32_t* aux_f_ptr, int32_t* aux_f_idx, int32_t* aux_b_ptr,
                                 int32_t* aux_b_idx, int32_t* dst_ptr, int32_t* dst_rows, int32_t* src_ptr, int32_t* src_rows,
                                 int64_t* host_counts, int32_t* host_rel_ptr, int32_t* host_modes, int32_t* host_status,
                                 void* workspace, size_t workspace_bytes, dn_stream_t stream) {
    DN_REQUIRE(G >= 0 && N >= 0 && R >= 1 && E >= 0, "dn_row_index_build_local: bad sizes");
    DN_REQUIRE(R <= kLocR, "dn_row_index_build_local: more than 64 relations (use dn_row_index_build_i32)");
    DN_REQUIRE(2 * E + N < 0x7fffffffLL && kSeg * R * G + 2 * (N + 1) + 1 < 0x7fffffffLL,
               "dn_row_index_build_local: sizes must fit int32");
    DN_REQUIRE(node_ptr && edge_ptr && row_in && row_out && aux_f_ptr && aux_b_ptr && dst_ptr && dst_rows && src_ptr && src_rows &&
               host_counts && host_rel_ptr && host_modes && host_status && workspace, "dn_row_index_build_local: NULL pointer");
    DN_REQUIRE(E == 0 || (src && dst && etype && aux_f_idx && aux_b_idx), "dn_row_index_build_local: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    LocWs w;
    size_t need = 0;
    int rc = loc_layout((char*)workspace, workspace_bytes, need, w, G, N, R);
    if (rc != DN_OK) return rc;
    if (need > workspace_bytes) { dn_set_error("dn_row_index_build_local: workspace too small (%zu < %zu)", workspace_bytes, need); return DN_ERR_WORKSPACE; }
    DN_CHECK_HIP(hipMemsetAsync(w.Er, 0, w.zero_bytes, st));
    const unsigned grid = (unsigned)(G > 0 ? (dn_cdiv(G, kLocWaves) < 2048 ? dn_cdiv(G, kLocWaves) : 2048) : 1);
    hipLaunchKernelGGL(ril_stats_kernel, dim3(grid), dim3(kLocWaves * 64), 0, st, G, (int32_t)R, node_ptr, edge_ptr, src, dst, etype,
                       N, E, w.Er, w.Dr, w.Sr, w.bad);
    hipLaunchKernelGGL(ril_count_kernel, dim3(grid), dim3(kLocWaves * 64), 0, st, G, N, (int32_t)R, edge_frac, self_loop, node_ptr,
                       edge_ptr, src, dst, etype, w.Er, w.Dr, w.Sr, w.mode, w.C);
    DN_CHECK_HIP(rocprim::exclusive_scan(w.scan_tmp, w.scan_tmp_bytes, (const int32_t*)w.C, w.S, (int32_t)0, (size_t)w.L,
                                         rocprim::plus<int32_t>(), st));
    hipLaunchKernelGGL(ril_fill_kernel, dim3(grid), dim3(kLocWaves * 64), 0, st, G, N, (int32_t)R, self_loop, node_ptr, edge_ptr, src,
                       dst, etype, w.mode, w.S, row_in, row_out, aux_f_ptr, aux_f_idx, aux_b_ptr, aux_b_idx, dst_ptr, dst_rows,
                       src_ptr, src_rows, w.meta);
    DN_CHECK_LAUNCH();
    int32_t h_meta[5 + kLocR + 1], h_mode[kLocR], h_bad = 0;
    DN_CHECK_HIP(hipMemcpyAsync(h_meta, w.meta, sizeof(int32_t) * (size_t)(5 + R + 1), hipMemcpyDeviceToHost, st));
    DN_CHECK_HIP(hipMemcpyAsync(h_mode, w.mode, sizeof(int32_t) * (size_t)R, hipMemcpyDeviceToHost, st));
    DN_CHECK_HIP(hipMemcpyAsync(&h_bad, w.bad, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    DN_CHECK_HIP(hipStreamSynchronize(st));
    *host_status = h_bad;
    for (int k = 0; k < 5; ++k) host_counts[k] = h_meta[k];
    for (int64_t r = 0; r <= R; ++r) host_rel_ptr[r] = h_meta[5 + r];
    for (int64_t r = 0; r < R; ++r) host_modes[r] = h_mode[r];
    return DN_OK;
}

}  // extern "C"
