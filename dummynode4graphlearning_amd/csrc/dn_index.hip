// One-shot device-side index builds (gfx950): CSR grouping, dummy-node augmentation, and the
// edge-to-vertex ("conjugate") transform L_Phi.  Integer work only; results are bit-exact with the
// reference's nested Python loops, including OUTPUT ORDER, because every "first occurrence" rule of the
// reference is restated as a STABLE radix sort + head flag + order-preserving compaction.
// Device-wide sort / scan primitives come from rocPRIM (AMD's own, compiled here for gfx950).
#include <cstring>
#include <cstdlib>

#include "dn_common.h"
#include "../../include/dn_hip.h"

#include <rocprim/rocprim.hpp>

namespace {

constexpr int kBlock = 256;
typedef unsigned long long u64;

struct Arena {
    char* base;
    size_t off;
    size_t cap;
    explicit Arena(void* p, size_t c) : base((char*)p), off(0), cap(c) {}
    template <typename T> T* take(int64_t n) {
        off = dn_align_up(off, 256);
        T* r = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += (size_t)(n > 0 ? n : 1) * sizeof(T);
        return r;
    }
    void* take_bytes(size_t n) {
        off = dn_align_up(off, 256);
        void* r = base ? base + off : nullptr;
        off += n > 0 ? n : 1;
        return r;
    }
    bool ok() const { return base == nullptr || off <= cap; }
};

static inline unsigned grid_for(int64_t n) { return (unsigned)(n > 0 ? dn_cdiv(n, kBlock) : 1); }
static inline int bits_for(u64 max_value) {
    int b = 1;
    while (b < 64 && (max_value >> b) != 0) ++b;
    return b;
}

// ----- rocPRIM wrappers: size query (temp == nullptr) or run --------------------------------------
template <typename K>
hipError_t sort_pairs(void* temp, size_t& bytes, const K* kin, K* kout, const int32_t* vin, int32_t* vout, int64_t n,
                      int end_bit, hipStream_t st) {
    return rocprim::radix_sort_pairs(temp, bytes, kin, kout, vin, vout, (size_t)n, 0u, (unsigned)end_bit, st);
}
hipError_t excl_scan(void* temp, size_t& bytes, const int32_t* in, int32_t* out, int64_t n, hipStream_t st) {
    return rocprim::exclusive_scan(temp, bytes, in, out, (int32_t)0, (size_t)n, rocprim::plus<int32_t>(), st);
}

// ----- small kernels ------------------------------------------------------------------------------
__global__ void iota_kernel(int32_t* v, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) v[i] = (int32_t)i;
}

// ptr[k] = first sorted position whose key >= k  (fills runs of empty keys too)
__global__ void ptr_from_sorted_kernel(const int32_t* __restrict__ skey, int64_t M, int64_t num_keys,
                                       int32_t* __restrict__ ptr) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i > M) return;
    const int64_t kprev = (i == 0) ? -1 : skey[i - 1];
    const int64_t kcur = (i == M) ? num_keys : skey[i];
    for (int64_t k = kprev + 1; k <= kcur; ++k) ptr[k] = (int32_t)i;
}

// largest g with p[g] <= x, p non-decreasing, p[0] <= x < p[G]
template <typename F> __device__ __forceinline__ int64_t find_graph(F p, int64_t G, int64_t x) {
    int64_t lo = 0, hi = G;  // invariant p(lo) <= x < p(hi)
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (p(mid) <= x) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ void degrees_kernel(int64_t E, const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                               int32_t* in_deg, int32_t* out_deg) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= E) return;
    if (in_deg) atomicAdd(&in_deg[dst[e]], 1);
    if (out_deg) atomicAdd(&out_deg[src[e]], 1);
}

__global__ void node_norm_kernel(int32_t self_loop, int64_t N, const int32_t* __restrict__ in_deg,
                                 const int32_t* __restrict__ out_deg, float* in_norm, float* out_norm) {
    const int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (v >= N) return;
    if (in_norm) {
        const int d = in_deg[v];
        in_norm[v] = self_loop ? 1.0f / ((float)d + 1.0f) : (d == 0 ? 0.0f : 1.0f / (float)d);
    }
    if (out_norm) {
        const int d = out_deg[v];
        out_norm[v] = self_loop ? 1.0f / ((float)d + 1.0f) : (d == 0 ? 0.0f : 1.0f / (float)d);
    }
}

__global__ void edge_norm_kernel(int32_t mode, int64_t E, const int32_t* __restrict__ src,
                                 const int32_t* __restrict__ dst, const float* __restrict__ in_norm,
                                 const float* __restrict__ out_norm, float* __restrict__ edge_norm) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= E) return;
    if (mode == 1) edge_norm[e] = in_norm[dst[e]];
    else edge_norm[e] = __fsqrt_rn(out_norm[src[e]] * in_norm[dst[e]]);
}

// ----- dummy augmentation -------------------------------------------------------------------------
__global__ void dummy_ptr_kernel(int64_t G, const int32_t* __restrict__ node_ptr, const int32_t* __restrict__ edge_ptr,
                                 int32_t* out_node_ptr, int32_t* out_edge_ptr) {
    const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (g > G) return;
    out_node_ptr[g] = node_ptr[g] + (int32_t)g;
    out_edge_ptr[g] = edge_ptr[g] + 2 * node_ptr[g];
}

// si == 0: GC layout (labels 0, ids = local index); si == 1: SI layout (ids/labels from the vocabulary)
__global__ void dummy_nodes_kernel(int32_t si, int64_t G, int64_t Nout, const int32_t* __restrict__ node_ptr,
                                   const int32_t* __restrict__ node_id, const int32_t* __restrict__ node_label,
                                   int32_t max_nv, int32_t max_nvl, int32_t* out_node_id, int32_t* out_node_label,
                                   uint8_t* out_is_dummy) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= Nout) return;
    const int64_t g = find_graph([&](int64_t k) { return (int64_t)node_ptr[k] + k; }, G, i);
    const int64_t local = i - (node_ptr[g] + g);
    const int64_t n = node_ptr[g + 1] - node_ptr[g];
    const bool dummy = local >= n;
    const int64_t o = node_ptr[g] + local;
    out_is_dummy[i] = dummy ? 1 : 0;
    if (si) {
        out_node_id[i] = dummy ? max_nv : node_id[o];
        out_node_label[i] = dummy ? max_nvl : node_label[o];
    } else {
        out_node_id[i] = (int32_t)local;
        out_node_label[i] = dummy ? 0 : node_label[o];
    }
}

__global__ void dummy_edges_kernel(int32_t si, int64_t G, int64_t Eout, const int32_t* __restrict__ node_ptr,
                                   const int32_t* __restrict__ edge_ptr, const int32_t* __restrict__ src,
                                   const int32_t* __restrict__ dst, const int32_t* __restrict__ edge_id,
                                   const int32_t* __restrict__ edge_label, const uint8_t* __restrict__ in_rev,
                                   int32_t max_ne, int32_t max_nel, int32_t* out_src, int32_t* out_dst,
                                   int32_t* out_edge_id, int32_t* out_edge_label, uint8_t* out_is_dummy,
                                   uint8_t* out_rev) {
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= Eout) return;
    const int64_t g = find_graph([&](int64_t k) { return (int64_t)edge_ptr[k] + 2 * (int64_t)node_ptr[k]; }, G, j);
    const int64_t local = j - ((int64_t)edge_ptr[g] + 2 * (int64_t)node_ptr[g]);
    const int64_t m = edge_ptr[g + 1] - edge_ptr[g];
    const int64_t n = node_ptr[g + 1] - node_ptr[g];
    const int64_t base = node_ptr[g] + g;  // first output node of graph g
    if (local < m) {
        const int64_t e = edge_ptr[g] + local;
        out_src[j] = src[e] + (int32_t)g;
        out_dst[j] = dst[e] + (int32_t)g;
        out_edge_label[j] = edge_label[e];
        out_edge_id[j] = si ? edge_id[e] : (int32_t)local;
        out_is_dummy[j] = 0;
        if (out_rev) out_rev[j] = in_rev ? in_rev[e] : 0;
        return;
    }
    const int64_t t = local - m;
    out_is_dummy[j] = 1;
    if (si) {
        // blocked: all (u -> dummy), then all (dummy -> u)
        const bool second = t >= n;
        const int64_t u = second ? t - n : t;
        out_src[j] = (int32_t)(second ? base + n : base + u);
        out_dst[j] = (int32_t)(second ? base + u : base + n);
        out_edge_id[j] = max_ne + (second ? 1 : 0);
        out_edge_label[j] = max_nel + (second ? 1 : 0);
        if (out_rev) out_rev[j] = second ? 1 : 0;
    } else {
        // interleaved: (n, v), (v, n)
        const int64_t v = t >> 1;
        const bool second = (t & 1) != 0;
        out_src[j] = (int32_t)(second ? base + v : base + n);
        out_dst[j] = (int32_t)(second ? base + n : base + v);
        out_edge_id[j] = (int32_t)local;
        out_edge_label[j] = 0;
    }
}

// ----- conjugate ----------------------------------------------------------------------------------
__global__ void edge_graph_kernel(int64_t G, int64_t E, const int32_t* __restrict__ edge_ptr, int32_t* egraph) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= E) return;
    egraph[e] = (int32_t)find_graph([&](int64_t k) { return (int64_t)edge_ptr[k]; }, G, e);
}

__global__ void first_dummy_kernel(int64_t E, const int32_t* __restrict__ egraph, const int32_t* __restrict__ edge_ptr,
                                   const uint8_t* __restrict__ is_dummy, int32_t* first_dummy) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= E || !is_dummy[e]) return;
    const int g = egraph[e];
    atomicMin(&first_dummy[g], (int32_t)(e - edge_ptr[g]));
}

__global__ void fill_i32_kernel(int32_t* p, int64_t n, int32_t v) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) p[i] = v;
}

// key = (graph << 32) | conj id of the edge inside its graph
__global__ void vertex_key_kernel(int32_t mode, int64_t E, const int32_t* __restrict__ egraph,
                                  const int32_t* __restrict__ edge_ptr, const int32_t* __restrict__ edge_id,
                                  const uint8_t* __restrict__ is_dummy, const int32_t* __restrict__ first_dummy,
                                  u64* key) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= E) return;
    const int g = egraph[e];
    int32_t cid;
    if (mode == DN_CONJ_SI) cid = edge_id[e];
    else if (mode == DN_CONJ_GC && is_dummy[e]) cid = first_dummy[g];
    else cid = edge_id ? edge_id[e] : (int32_t)(e - edge_ptr[g]);
    key[e] = ((u64)(uint32_t)g << 32) | (u64)(uint32_t)cid;
}

__global__ void head_flag_u64_kernel(const u64* __restrict__ skey, int64_t n, int32_t* flag) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    flag[i] = (i == 0 || skey[i] != skey[i - 1]) ? 1 : 0;
}

// after exclusive scan of head flags: vertex index of sorted position p = scan[p] + flag[p] - 1
__global__ void vertex_assign_kernel(int64_t E, const u64* __restrict__ skey, const int32_t* __restrict__ sval,
                                     const int32_t* __restrict__ flag, const int32_t* __restrict__ scan,
                                     int32_t* vmap, int32_t* rep_edge, int32_t* vgraph_count) {
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= E) return;
    const int32_t k = scan[p] + flag[p] - 1;
    vmap[sval[p]] = k;
    if (flag[p]) {
        rep_edge[k] = sval[p];  // stable sort => first (lowest) edge carrying that id
        atomicAdd(&vgraph_count[(int32_t)(skey[p] >> 32)], 1);
    }
}

__global__ void raw_count_kernel(int64_t E, const int32_t* __restrict__ src, const int32_t* __restrict__ in_ptr,
                                 int32_t* cnt) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= E) return;
    const int s = src[e];
    cnt[e] = in_ptr[s + 1] - in_ptr[s];
}

// raw conj edge t = (i -> e): e ascending, i ascending over the in-edges of src(e)
__global__ void raw_fill_kernel(int64_t E, int64_t T, const int32_t* __restrict__ raw_off,
                                const int32_t* __restrict__ src, const int32_t* __restrict__ in_ptr,
                                const int32_t* __restrict__ in_perm, const int32_t* __restrict__ vmap,
                                const int32_t* __restrict__ node_label, u64* key, int32_t* label, int32_t* shared) {
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= T) return;
    const int64_t e = find_graph([&](int64_t k) { return (int64_t)raw_off[k]; }, E, t);
    const int s = src[e];
    const int i = in_perm[in_ptr[s] + (int)(t - raw_off[e])];
    key[t] = ((u64)(uint32_t)vmap[i] << 32) | (u64)(uint32_t)vmap[e];
    shared[t] = s;
    if (label) label[t] = node_label[s];
}

__global__ void gather_u64_kernel(const u64* __restrict__ in, const int32_t* __restrict__ perm, int64_t n, u64* out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[i] = in[perm[i]];
}

// keep[t] for the element at sorted position p (original raw index sval[p])
__global__ void keep_flag_kernel(int32_t mode, int64_t T, const u64* __restrict__ skey, const int32_t* __restrict__ sval,
                                 const int32_t* __restrict__ label /* original order, may be NULL */,
                                 const int32_t* __restrict__ rep_edge, const uint8_t* __restrict__ is_dummy,
                                 int32_t* keep) {
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= T) return;
    const int t = sval[p];
    bool head = true;
    if (p > 0 && skey[p] == skey[p - 1]) {
        head = false;
        if (mode == DN_CONJ_SI && label[t] != label[sval[p - 1]]) head = true;
    }
    if (mode == DN_CONJ_GC && head) {
        const uint32_t cu = (uint32_t)(skey[p] >> 32), cv = (uint32_t)skey[p];
        if (cu == cv && is_dummy[rep_edge[cu]]) head = false;  // the (Phi, Phi) self loop
    }
    keep[t] = head ? 1 : 0;
}

__global__ void compact_kernel(int64_t T, const u64* __restrict__ key, const int32_t* __restrict__ shared,
                               const int32_t* __restrict__ keep, const int32_t* __restrict__ kscan, int32_t* csrc,
                               int32_t* cdst, int32_t* out_shared) {
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= T || !keep[t]) return;
    const int o = kscan[t];
    csrc[o] = (int32_t)(key[t] >> 32);
    cdst[o] = (int32_t)(uint32_t)key[t];
    out_shared[o] = shared[t];
}

__global__ void cedge_ptr_kernel(int64_t G, int64_t E, int64_t T, const int32_t* __restrict__ edge_ptr,
                                 const int32_t* __restrict__ raw_off, const int32_t* __restrict__ kscan,
                                 const int32_t* __restrict__ keep, int32_t* cedge_ptr) {
    const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (g > G) return;
    const int64_t t = raw_off[edge_ptr[g]];  // raw_off has E+1 entries, raw_off[E] == T
    cedge_ptr[g] = (t >= T) ? (T > 0 ? kscan[T - 1] + keep[T - 1] : 0) : kscan[t];
}

int csr_build(const int32_t* key, int64_t M, int64_t num_keys, int32_t* ptr, int32_t* perm, void* ws, size_t ws_bytes,
              hipStream_t st, size_t* need_out) {
    Arena a(ws, ws_bytes);
    int32_t* skey = a.take<int32_t>(M);
    int32_t* vin = a.take<int32_t>(M);
    size_t tb = 0;
    const int end_bit = bits_for((u64)(num_keys > 0 ? num_keys - 1 : 0));
    if (M > 0) {
        hipError_t e = sort_pairs<int32_t>(nullptr, tb, key, skey, vin, perm, M, end_bit, st);
        if (e != hipSuccess) { dn_set_error("rocprim size query failed: %s", hipGetErrorString(e)); return DN_ERR_HIP; }
    }
    void* temp = a.take_bytes(tb);
    if (need_out) { *need_out = a.off; return DN_OK; }
    if (!a.ok()) { dn_set_error("dn_csr_build: workspace too small (%zu < %zu)", ws_bytes, a.off); return DN_ERR_WORKSPACE; }
    if (M > 0) {
        hipLaunchKernelGGL(iota_kernel, dim3(grid_for(M)), dim3(kBlock), 0, st, vin, M);
        DN_CHECK_HIP(sort_pairs<int32_t>(temp, tb, key, skey, vin, perm, M, end_bit, st));
    }
    hipLaunchKernelGGL(ptr_from_sorted_kernel, dim3(grid_for(M + 1)), dim3(kBlock), 0, st, skey, M, num_keys, ptr);
    DN_CHECK_LAUNCH();
    return DN_OK;
}

struct ConjWs {
    // shared by count and build
    int32_t *in_ptr, *in_perm, *cnt, *raw_off;
    void* csr_ws; size_t csr_ws_bytes;
    void* scan_tmp; size_t scan_tmp_bytes;
    // build only
    int32_t *egraph, *first_dummy, *vflag, *vscan, *vmap, *vcount, *sval_e, *iota_e;
    u64 *vkey, *vkey_s;
    void* sortE_tmp; size_t sortE_tmp_bytes;
    u64 *rkey, *rkey_s, *rkey_g;
    int32_t *rlabel, *rlabel_s, *rshared, *rval, *rval_s, *rval_s2, *keep, *kscan;
    void* sortT_tmp; size_t sortT_tmp_bytes;
    void* sortL_tmp; size_t sortL_tmp_bytes;
    void* scanT_tmp; size_t scanT_tmp_bytes;
    void* scanG_tmp; size_t scanG_tmp_bytes;
};

// Lays out the workspace; with base == nullptr only sizes are computed.  T < 0 => count-phase only.
int conj_layout(Arena& a, ConjWs& w, int64_t G, int64_t N, int64_t E, int64_t T, hipStream_t st) {
    memset(&w, 0, sizeof(w));
    hipError_t e;
    w.in_ptr = a.take<int32_t>(N + 1);
    w.in_perm = a.take<int32_t>(E);
    w.cnt = a.take<int32_t>(E + 1);
    w.raw_off = a.take<int32_t>(E + 1);
    size_t need = 0;
    int rc = csr_build(nullptr, E, N, nullptr, nullptr, nullptr, 0, st, &need);
    if (rc != DN_OK) return rc;
    w.csr_ws_bytes = need;
    w.csr_ws = a.take_bytes(need);
    e = excl_scan(nullptr, w.scan_tmp_bytes, w.cnt, w.raw_off, E + 1, st);
    if (e != hipSuccess) { dn_set_error("rocprim scan size query failed: %s", hipGetErrorString(e)); return DN_ERR_HIP; }
    w.scan_tmp = a.take_bytes(w.scan_tmp_bytes);
    if (T < 0) return DN_OK;
    w.egraph = a.take<int32_t>(E);
    w.first_dummy = a.take<int32_t>(G + 1);
    w.vflag = a.take<int32_t>(E);
    w.vscan = a.take<int32_t>(E);
    w.vmap = a.take<int32_t>(E);
    w.vcount = a.take<int32_t>(G + 1);
    w.sval_e = a.take<int32_t>(E);
    w.iota_e = a.take<int32_t>(E);
    w.vkey = a.take<u64>(E);
    w.vkey_s = a.take<u64>(E);
    if (E > 0) {
        e = sort_pairs<u64>(nullptr, w.sortE_tmp_bytes, w.vkey, w.vkey_s, w.iota_e, w.sval_e, E, 64, st);
        if (e != hipSuccess) { dn_set_error("rocprim size query failed: %s", hipGetErrorString(e)); return DN_ERR_HIP; }
    }
    w.sortE_tmp = a.take_bytes(w.sortE_tmp_bytes);
    e = excl_scan(nullptr, w.scanG_tmp_bytes, w.vcount, w.first_dummy, G + 1, st);
    if (e != hipSuccess) { dn_set_error("rocprim scan size query failed: %s", hipGetErrorString(e)); return DN_ERR_HIP; }
    w.scanG_tmp = a.take_bytes(w.scanG_tmp_bytes);
    w.rkey = a.take<u64>(T);
    w.rkey_s = a.take<u64>(T);
    w.rkey_g = a.take<u64>(T);
    w.rlabel = a.take<int32_t>(T);
    w.rlabel_s = a.take<int32_t>(T);
    w.rshared = a.take<int32_t>(T);
    w.rval = a.take<int32_t>(T);
    w.rval_s = a.take<int32_t>(T);
    w.rval_s2 = a.take<int32_t>(T);
    w.keep = a.take<int32_t>(T);
    w.kscan = a.take<int32_t>(T);
    if (T > 0) {
        e = sort_pairs<u64>(nullptr, w.sortT_tmp_bytes, w.rkey, w.rkey_s, w.rval, w.rval_s, T, 64, st);
        if (e == hipSuccess) e = sort_pairs<int32_t>(nullptr, w.sortL_tmp_bytes, w.rlabel, w.rlabel_s, w.rval, w.rval_s, T, 32, st);
        if (e == hipSuccess) e = excl_scan(nullptr, w.scanT_tmp_bytes, w.keep, w.kscan, T, st);
        if (e != hipSuccess) { dn_set_error("rocprim size query failed: %s", hipGetErrorString(e)); return DN_ERR_HIP; }
    }
    w.sortT_tmp = a.take_bytes(w.sortT_tmp_bytes);
    w.sortL_tmp = a.take_bytes(w.sortL_tmp_bytes);
    w.scanT_tmp = a.take_bytes(w.scanT_tmp_bytes);
    return DN_OK;
}

// in-CSR by dst + raw offsets; leaves raw_off[E] = T on the device
int conj_count_phase(ConjWs& w, int64_t N, int64_t E, const int32_t* src, const int32_t* dst, hipStream_t st) {
    int rc = csr_build(dst, E, N, w.in_ptr, w.in_perm, w.csr_ws, w.csr_ws_bytes, st, nullptr);
    if (rc != DN_OK) return rc;
    DN_CHECK_HIP(hipMemsetAsync(w.cnt, 0, sizeof(int32_t) * (size_t)(E + 1), st));
    if (E > 0) hipLaunchKernelGGL(raw_count_kernel, dim3(grid_for(E)), dim3(kBlock), 0, st, E, src, w.in_ptr, w.cnt);
    DN_CHECK_HIP(excl_scan(w.scan_tmp, w.scan_tmp_bytes, w.cnt, w.raw_off, E + 1, st));
    DN_CHECK_LAUNCH();
    return DN_OK;
}

// ----- relation-aware segment index ((rel, dst) segments) -------------------------------------------
__global__ void rel_key_kernel(int64_t E, int64_t N, const int32_t* __restrict__ dst, const int32_t* __restrict__ etype,
                               int32_t* key) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e < E) key[e] = (int32_t)((int64_t)etype[e] * N + dst[e]);
}
__global__ void head_flag_i32_kernel(const int32_t* __restrict__ skey, int64_t n, int32_t* flag) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    flag[i] = (i == 0 || skey[i] != skey[i - 1]) ? 1 : 0;
}
__global__ void rel_segments_kernel(int64_t E, int64_t N, const int32_t* __restrict__ skey, const int32_t* __restrict__ perm1,
                                    const int32_t* __restrict__ flag, const int32_t* __restrict__ scan,
                                    const int32_t* __restrict__ src, int32_t* src1, int32_t* seg_ptr, int32_t* seg_dst,
                                    int32_t* seg_rel, int32_t* seg_of_edge) {
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p > E) return;
    if (p == E) {  // sentinel: seg_ptr[P] = E
        const int32_t P = E > 0 ? scan[E - 1] + flag[E - 1] : 0;
        seg_ptr[P] = (int32_t)E;
        return;
    }
    const int32_t k = scan[p] + flag[p] - 1;
    const int32_t e = perm1[p];
    src1[p] = src[e];
    seg_of_edge[e] = k;
    if (flag[p]) {
        seg_ptr[k] = (int32_t)p;
        seg_dst[k] = (int32_t)(skey[p] % N);
        seg_rel[k] = (int32_t)(skey[p] / N);
    }
}
__global__ void gather_i32_kernel(const int32_t* __restrict__ in, const int32_t* __restrict__ perm, int64_t n, int32_t* out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[i] = in[perm[i]];
}

struct RelWs {
    int32_t *key, *skey, *iota, *flag, *scan, *seg_rel, *seg_of_edge;
    void* sort_tmp; size_t sort_tmp_bytes;
    void* scan_tmp; size_t scan_tmp_bytes;
    void* csr_ws; size_t csr_ws_bytes;
};
int rel_layout(Arena& a, RelWs& w, int64_t N, int64_t R, int64_t E, hipStream_t st) {
    memset(&w, 0, sizeof(w));
    w.key = a.take<int32_t>(E);
    w.skey = a.take<int32_t>(E);
    w.iota = a.take<int32_t>(E);
    w.flag = a.take<int32_t>(E);
    w.scan = a.take<int32_t>(E);
    w.seg_rel = a.take<int32_t>(E);
    w.seg_of_edge = a.take<int32_t>(E);
    hipError_t e = hipSuccess;
    if (E > 0) {
        e = sort_pairs<int32_t>(nullptr, w.sort_tmp_bytes, w.key, w.skey, w.iota, w.flag, E, 32, st);
        if (e == hipSuccess) e = excl_scan(nullptr, w.scan_tmp_bytes, w.flag, w.scan, E, st);
        if (e != hipSuccess) { dn_set_error("rocprim size query failed: %s", hipGetErrorString(e)); return DN_ERR_HIP; }
    }
    w.sort_tmp = a.take_bytes(w.sort_tmp_bytes);
    w.scan_tmp = a.take_bytes(w.scan_tmp_bytes);
    size_t need = 0;
    int rc = csr_build(nullptr, E, N, nullptr, nullptr, nullptr, 0, st, &need);
    if (rc != DN_OK) return rc;
    w.csr_ws_bytes = need;
    w.csr_ws = a.take_bytes(need);
    (void)R;
    return DN_OK;
}

// ----- row factorisation index (bf16 fused RGIN/RGCN path; see dn_hip.h dn_row_index_build_i32) -----------------
constexpr int kModeEdge = 0, kModeAgg = 1, kModeTf = 2;

// per relation: #edges, #distinct destinations, #distinct sources -- what decides EDGE / AGG / TF.
// No sorting: one bit per (relation, node) pair, set with atomicOr -- the lane that flips a bit counts it (counts do not depend
// on the order, so the result is deterministic).  Batches list their edges graph by graph, so neighbouring lanes mostly hold
// the same relation and, for the dummy relations, the same (relation, node) pair: a lane equal to its left neighbour skips
// the atomic, and every run of equal relations adds its three counts once (LDS histogram per block, then R global atomics).
constexpr int kStatsMaxR = 1024;
__global__ __launch_bounds__(kBlock) void ri_stats_bitmap_kernel(int64_t E, int64_t N, int32_t R, const int32_t* __restrict__ src,
                                                                 const int32_t* __restrict__ dst,
                                                                 const int32_t* __restrict__ etype,
                                                                 unsigned int* __restrict__ bitsD,
                                                                 unsigned int* __restrict__ bitsS, int32_t* __restrict__ Er,
                                                                 int32_t* __restrict__ Dr, int32_t* __restrict__ Sr) {
    __shared__ int32_t h[3][kStatsMaxR];
    const bool use_lds = R <= kStatsMaxR;
    if (use_lds) {
        for (int i = threadIdx.x; i < 3 * kStatsMaxR; i += blockDim.x) (&h[0][0])[i] = 0;
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const int64_t chunks = (E + blockDim.x - 1) / blockDim.x;
    for (int64_t c = blockIdx.x; c < chunks; c += gridDim.x) {                    // whole waves stay in the loop together
        const int64_t e = c * blockDim.x + threadIdx.x;
        const bool ok = e < E;
        const int r = ok ? etype[e] : -1;
        const int64_t kd = ok ? (int64_t)r * N + dst[e] : -1, ks = ok ? (int64_t)r * N + src[e] : -2;
        const int rp = __shfl_up(r, 1);
        const long long kdp = __shfl_up((long long)kd, 1), ksp = __shfl_up((long long)ks, 1);
        const bool head = ok && (lane == 0 || r != rp);
        bool newd = false, news = false;
        if (ok && (lane == 0 || kd != kdp)) {
            const unsigned int m = 1u << (kd & 31);
            newd = !(atomicOr(&bitsD[kd >> 5], m) & m);
        }
        if (ok && (lane == 0 || ks != ksp)) {
            const unsigned int m = 1u << (ks & 31);
            news = !(atomicOr(&bitsS[ks >> 5], m) & m);
        }
        const unsigned long long Vm = __ballot(ok), Hm = __ballot(head), Dm = __ballot(newd), Sm = __ballot(news);
        if (head) {
            const unsigned long long later = lane == 63 ? 0ull : (Hm >> (lane + 1)) << (lane + 1);   // heads after me
            const int next = later ? __ffsll((long long)later) - 1 : 64;
            const unsigned long long upto = next == 64 ? ~0ull : ((1ull << next) - 1ull);
            const unsigned long long run = upto & ~((1ull << lane) - 1ull) & Vm;                     // lanes [me, next head)
            const int ne = __popcll(run), nd = __popcll(run & Dm), ns = __popcll(run & Sm);
            if (use_lds) {
                atomicAdd(&h[0][r], ne);
                if (nd) atomicAdd(&h[1][r], nd);
                if (ns) atomicAdd(&h[2][r], ns);
            } else {
                atomicAdd(&Er[r], ne);
                if (nd) atomicAdd(&Dr[r], nd);
                if (ns) atomicAdd(&Sr[r], ns);
            }
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int r = threadIdx.x; r < R; r += blockDim.x) {
            if (h[0][r]) atomicAdd(&Er[r], h[0][r]);
            if (h[1][r]) atomicAdd(&Dr[r], h[1][r]);
            if (h[2][r]) atomicAdd(&Sr[r], h[2][r]);
        }
    }
}
// The same counts with plain stores when N * R bytes are affordable (<= kByteMapMax): one BYTE per (relation, node) pair is set
// to 1 (idempotent: no atomic, no return trip -- the scattered atomicOr above manages ~18 G/s), then the bytes are summed per
// relation.  #edges per relation comes from the same run aggregation.
constexpr int64_t kByteMapMax = 1ll << 28;
__global__ __launch_bounds__(kBlock) void ri_stats_mark_kernel(int64_t E, int64_t Np, int32_t R, const int32_t* __restrict__ src,
                                                               const int32_t* __restrict__ dst,
                                                               const int32_t* __restrict__ etype, uint8_t* __restrict__ mapD,
                                                               uint8_t* __restrict__ mapS, int32_t* __restrict__ Er) {
    __shared__ int32_t h[kStatsMaxR];
    const bool use_lds = R <= kStatsMaxR;
    if (use_lds) {
        for (int i = threadIdx.x; i < kStatsMaxR; i += blockDim.x) h[i] = 0;
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const int64_t chunks = (E + blockDim.x - 1) / blockDim.x;
    for (int64_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        const int64_t e = c * blockDim.x + threadIdx.x;
        const bool ok = e < E;
        const int r = ok ? etype[e] : -1;
        if (ok) {
            mapD[(int64_t)r * Np + dst[e]] = 1;                   // Np: node count rounded up to 16 (aligned 16-byte reads below)
            mapS[(int64_t)r * Np + src[e]] = 1;
        }
        const int rp = __shfl_up(r, 1);
        const bool head = ok && (lane == 0 || r != rp);
        const unsigned long long Vm = __ballot(ok), Hm = __ballot(head);
        if (head) {
            const unsigned long long later = lane == 63 ? 0ull : (Hm >> (lane + 1)) << (lane + 1);
            const int next = later ? __ffsll((long long)later) - 1 : 64;
            const unsigned long long upto = next == 64 ? ~0ull : ((1ull << next) - 1ull);
            const int ne = __popcll(upto & ~((1ull << lane) - 1ull) & Vm);
            if (use_lds) atomicAdd(&h[r], ne);
            else atomicAdd(&Er[r], ne);
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int r = threadIdx.x; r < R; r += blockDim.x)
            if (h[r]) atomicAdd(&Er[r], h[r]);
    }
}
// Dr[r] / Sr[r] = number of set bytes of relation r (grid: x = slices of the node range, y = relation, z = map)
__global__ __launch_bounds__(kBlock) void ri_stats_count_kernel(int64_t Np, const uint8_t* __restrict__ mapD,
                                                                const uint8_t* __restrict__ mapS, int32_t* __restrict__ Dr,
                                                                int32_t* __restrict__ Sr) {
    const uint4* m = reinterpret_cast<const uint4*>((blockIdx.z == 0 ? mapD : mapS) + (int64_t)blockIdx.y * Np);
    int32_t cnt = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < Np / 16; i += (int64_t)gridDim.x * blockDim.x) {
        const uint4 v = m[i];                                    // bytes are 0 or 1
        cnt += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&(blockIdx.z == 0 ? Dr : Sr)[blockIdx.y], cnt);
}
// the ten scan totals the host needs, packed so that ONE copy fetches them: out[k] = scan[k][E-1] + flag[k][E-1]
__global__ void ri_totals_kernel(int64_t E, const int32_t* a0, const int32_t* a1, const int32_t* b0, const int32_t* b1,
                                 const int32_t* c0, const int32_t* c1, const int32_t* d0, const int32_t* d1, const int32_t* e0,
                                 const int32_t* e1, int32_t* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = a0[E - 1] + a1[E - 1]; out[1] = b0[E - 1] + b1[E - 1]; out[2] = c0[E - 1] + c1[E - 1];
        out[3] = d0[E - 1] + d1[E - 1]; out[4] = e0[E - 1] + e1[E - 1];
    }
}
__global__ void ri_mode_kernel(int64_t R, float edge_frac, const int32_t* __restrict__ Er, const int32_t* __restrict__ Dr,
                               const int32_t* __restrict__ Sr, int32_t* mode) {
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= R) return;
    const int mn = Dr[r] < Sr[r] ? Dr[r] : Sr[r];
    if (Er[r] == 0 || (float)mn > edge_frac * (float)Er[r]) mode[r] = kModeEdge;
    else mode[r] = Dr[r] <= Sr[r] ? kModeAgg : kModeTf;
}
__global__ void ri_rowkey_kernel(int64_t E, int64_t N, const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                 const int32_t* __restrict__ etype, const int32_t* __restrict__ mode, int32_t* key) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= E) return;
    const int r = etype[e];
    key[e] = (int32_t)((int64_t)r * N + (mode[r] == kModeTf ? src[e] : dst[e]));
}
// flags over the sorted edges: row head, AGG-row head, TF-row head, AGG edge, TF edge
__global__ void ri_flags_kernel(int64_t E, int64_t N, const int32_t* __restrict__ skey, const int32_t* __restrict__ mode,
                                int32_t* head, int32_t* agg_head, int32_t* tf_head, int32_t* agg_edge, int32_t* tf_edge) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= E) return;
    const int m = mode[skey[i] / N];
    const int h = (m == kModeEdge || i == 0 || skey[i] != skey[i - 1]) ? 1 : 0;
    head[i] = h;
    agg_head[i] = (h && m == kModeAgg) ? 1 : 0;
    tf_head[i] = (h && m == kModeTf) ? 1 : 0;
    agg_edge[i] = m == kModeAgg ? 1 : 0;
    tf_edge[i] = m == kModeTf ? 1 : 0;
}
// per sorted edge i (row = rows[i] = inclusive head count - 1): fill row tables at heads, aux lists, list entries.
// List entries are written COMPACT (only the ones that exist, so the two per-node sorts handle ~E + N entries instead of
// 2E + N), in the order the stable sort must preserve: the per-edge entries in sorted-edge order, then one entry per collapsed
// row in row order, then (ri_tail_kernel) the self loops.
//   forward : edge i unless its relation is AGG -> slot i - agge_scan[i];   AGG row a -> nf_e + a      (nf_e = E - #AGG edges)
//   backward: edge i unless its relation is TF  -> slot i - tfe_scan[i];    TF row a  -> nb_e + a      (nb_e = E - #TF edges)
__global__ void ri_fill_kernel(int64_t E, int64_t N, int64_t nf_e, int64_t nb_e, const int32_t* __restrict__ skey,
                               const int32_t* __restrict__ order, const int32_t* __restrict__ mode,
                               const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                               const int32_t* __restrict__ head, const int32_t* __restrict__ head_scan,
                               const int32_t* __restrict__ aggh_scan, const int32_t* __restrict__ tfh_scan,
                               const int32_t* __restrict__ agge_scan, const int32_t* __restrict__ tfe_scan,
                               int32_t* row_in, int32_t* row_out, int32_t* row_rel, int32_t* aux_f_idx, int32_t* aux_f_ptr,
                               int32_t* aux_b_idx, int32_t* aux_b_ptr, int32_t* f_key, int32_t* f_row, int32_t* b_key,
                               int32_t* b_row) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= E) return;
    const int rel = skey[i] / (int32_t)N, node = skey[i] % (int32_t)N;
    const int m = mode[rel], e = order[i], h = head[i];
    const int row = head_scan[i] + h - 1;                       // exclusive scan + own flag - 1
    if (h) {
        row_rel[row] = rel;
        if (m == kModeAgg) {
            const int a = aggh_scan[i];
            row_in[row] = (int32_t)N + a; row_out[row] = node;
            aux_f_ptr[a] = agge_scan[i];
            f_key[nf_e + a] = node; f_row[nf_e + a] = row;      // one forward entry per AGG row
        } else if (m == kModeTf) {
            const int a = tfh_scan[i];
            row_in[row] = node; row_out[row] = (int32_t)N + a;
            aux_b_ptr[a] = tfe_scan[i];
            b_key[nb_e + a] = node; b_row[nb_e + a] = row;
        } else {
            row_in[row] = src[e]; row_out[row] = dst[e];
        }
    }
    // per-edge entries
    const int64_t pf = i - agge_scan[i], pb = i - tfe_scan[i];
    if (m == kModeAgg) {
        aux_f_idx[agge_scan[i]] = src[e];                       // (forward: covered by the row entry)
        b_key[pb] = src[e]; b_row[pb] = row;
    } else if (m == kModeTf) {
        aux_b_idx[tfe_scan[i]] = dst[e];
        f_key[pf] = dst[e]; f_row[pf] = row;
    } else {
        f_key[pf] = dst[e]; f_row[pf] = row;
        b_key[pb] = src[e]; b_row[pb] = row;
    }
}
__global__ void ri_tail_kernel(int64_t N, int64_t P, int64_t n_agg, int64_t n_tf, int64_t n_agg_e, int64_t n_tf_e,
                               int64_t f_self, int64_t b_self, int32_t self_loop, int32_t* row_in, int32_t* row_out,
                               int32_t* aux_f_ptr, int32_t* aux_b_ptr, int32_t* f_key, int32_t* f_row, int32_t* b_key,
                               int32_t* b_row) {
    const int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (v == 0) { aux_f_ptr[n_agg] = (int32_t)n_agg_e; aux_b_ptr[n_tf] = (int32_t)n_tf_e; }
    if (v >= N || !self_loop) return;
    row_in[P + v] = (int32_t)v; row_out[P + v] = (int32_t)v;
    f_key[f_self + v] = (int32_t)v; f_row[f_self + v] = (int32_t)(P + v);        // self-loop entries close every list
    b_key[b_self + v] = (int32_t)v; b_row[b_self + v] = (int32_t)(P + v);
}

// ----- tile / chunk tables of relation-major rows, built on the device (no host loop, no device -> host sync) ----------------
// entry i of a table = {rel, beg, end, 0}: piece k of relation r covers rows [rel_ptr[r] + k*step, min(.. + step, rel_ptr[r+1])).
// Tables are sized by an upper bound (rows / step + relations); the unused tail is filled with empty pieces of the last
// relation (beg == end: consumers skip them without reloading weights).
__global__ void row_tables_kernel(int32_t Rt, const int32_t* __restrict__ rel_ptr, int32_t step, int64_t max_entries,
                                  int32_t* __restrict__ table, int32_t* __restrict__ piece_ptr, unsigned long long skip_mask) {
    __shared__ int32_t pp[1025];                                  // pieces before relation r (Rt <= 1024)
    if (threadIdx.x == 0) {
        int32_t acc = 0;
        for (int r = 0; r < Rt; ++r) {
            pp[r] = acc;
            const int32_t cnt = (r < 64 && ((skip_mask >> r) & 1ull)) ? 0 : rel_ptr[r + 1] - rel_ptr[r];   // skipped relation: no pieces
            acc += (cnt + step - 1) / step;
        }
        pp[Rt] = acc;
    }
    __syncthreads();
    if (piece_ptr && blockIdx.x == 0)
        for (int r = threadIdx.x; r <= Rt; r += blockDim.x) piece_ptr[r] = pp[r];
    const int32_t total = pp[Rt];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < max_entries; i += (int64_t)gridDim.x * blockDim.x) {
        int32_t rel, beg, end;
        if (i < total) {
            int lo = 0, hi = Rt;                                  // last r with pp[r] <= i
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (pp[mid] <= (int32_t)i) lo = mid; else hi = mid; }
            rel = lo;
            beg = rel_ptr[rel] + ((int32_t)i - pp[rel]) * step;
            end = min(beg + step, rel_ptr[rel + 1]);
        } else {
            rel = Rt - 1; beg = rel_ptr[Rt]; end = beg;
        }
        table[4 * i] = rel; table[4 * i + 1] = beg; table[4 * i + 2] = end; table[4 * i + 3] = 0;
    }
}

// ----- L2-blocked ("sweep") tile order of relation-major rows for the persistent transform launch (dn_hip.h) --------------
// One launch, closed forms only: every block recomputes the small per-group tables in LDS (9 x R lower bounds, the shares of
// the W workgroups of each of the 8 groups), then fills its entries (workgroup b = j * 8 + x, slot m).  Mirrors
// tests/sweep_ref.py (wg_shares + the event ranks) line by line.
constexpr int kSwGroups = 8, kSwMaxRel = 64, kSwMaxW = 64, kSwTile = 32;
// Below this many tiles per workgroup no relation gets workgroups of its own: the group's whole tile line is cut into W segments
// (sweep_ref.PURE_MIN_S).  With "pure" workgroups the two or three helpers of a group take the left-overs of ALL its relations,
// one weight reload (~2.7 us) each -- 21 us on a launch whose median workgroup needs 39 (an eighth of config 5: 32 tiles each).
int sw_pure_min() {
    static const int v = dn_knob("DN_SW_PURE_MIN", 64);
    return v;
}
// What a helper's share is weighed by (sweep_ref.HELP_PCT / HELP_SW): per cent a helper's tile costs more than a pure workgroup's,
// tiles a change of relation costs.  The pure quota grows by delta so that both kinds finish together (sweep_ref.wg_shares).
int sw_help_pct() {
    static const int v = dn_knob("DN_SW_HELP_PCT", 5);
    return v;
}
int sw_help_sw() {
    static const int v = dn_knob("DN_SW_HELP_SW", 4);
    return v;
}

struct SwDir {
    int32_t *table, *info;
    const int32_t* dyn;
    unsigned long long skip_mask;
};
struct SwPair {
    SwDir d[2];
};

__global__ __launch_bounds__(256) void sweep_tables_kernel(int32_t R, const int32_t* __restrict__ rel_ptr,
                                                            const int32_t* __restrict__ row_in, const int32_t* __restrict__ row_out,
                                                            int32_t N, int32_t W, int32_t S_cap, SwPair pr, int32_t pure_min,
                                                            int32_t help_pct, int32_t help_sw) {
    // blockIdx.y picks the table (dn_conv_index_build_i32 builds the orders of both transform launches in one launch)
    unsigned long long skip_mask = pr.d[blockIdx.y].skip_mask;
    int32_t* __restrict__ table = pr.d[blockIdx.y].table;
    int32_t* __restrict__ info = pr.d[blockIdx.y].info;
    const int32_t* __restrict__ dyn = pr.d[blockIdx.y].dyn;
    if (dyn != nullptr) {                                         // queued behind the row index: {relation to skip or -1, go} read on
        if (dyn[1] == 0) return;                                  // the device; go = 0: nothing to do
        skip_mask = dyn[0] >= 0 ? 1ull << dyn[0] : 0ull;
    }
    __shared__ int32_t lo[kSwGroups + 1][kSwMaxRel];              // first row of relation r in group x
    __shared__ int32_t T[kSwGroups][kSwMaxRel];                   // tiles of (group, relation)
    __shared__ int32_t pure0[kSwGroups][kSwMaxRel + 1];           // first pure workgroup of relation r (prefix of the pure counts)
    __shared__ int32_t crem[kSwGroups][kSwMaxRel + 1];            // prefix of the left-over tiles
    __shared__ int32_t Sx[kSwGroups], Wh[kSwGroups];
    __shared__ int32_t pp[kSwMaxRel + 1];                         // plain order: tiles before relation r
    __shared__ int32_t s_plain, s_stride;
    const int tid = threadIdx.x;
    auto key_of = [&](int32_t p) -> int32_t { const int32_t o = row_out[p]; return o < N ? o : row_in[p]; };
    for (int i = tid; i < (kSwGroups + 1) * R; i += blockDim.x) {
        const int x = i / R, r = i % R;
        const int32_t a = rel_ptr[r], b = rel_ptr[r + 1];
        int32_t v;
        if ((skip_mask >> r) & 1ull) v = a;
        else if (x == 0) v = a;
        else if (x == kSwGroups) v = b;
        else {
            const int32_t kx = (int32_t)(((int64_t)x * N) / kSwGroups);
            int32_t l = a, h = b;                                 // first row with key >= kx
            while (l < h) { const int32_t mid = l + ((h - l) >> 1); if (key_of(mid) < kx) l = mid + 1; else h = mid; }
            v = l;
        }
        lo[x][r] = v;
    }
    __syncthreads();
    if (tid < R) {                                                // keys need not be monotone: make the bounds so (any cut is valid)
        for (int x = 1; x <= kSwGroups; ++x) lo[x][tid] = max(lo[x][tid], lo[x - 1][tid]);
        if (!((skip_mask >> tid) & 1ull)) lo[kSwGroups][tid] = rel_ptr[tid + 1];
    }
    __syncthreads();
    for (int i = tid; i < kSwGroups * R; i += blockDim.x) {
        const int x = i / R, r = i % R;
        T[x][r] = (lo[x + 1][r] - lo[x][r] + kSwTile - 1) / kSwTile;
    }
    __syncthreads();
    if (tid < kSwGroups) {                                        // shares of group x (sweep_ref.wg_shares)
        const int x = tid;
        int64_t Tx = 0;
        for (int r = 0; r < R; ++r) Tx += T[x][r];
        int32_t S = (int32_t)((Tx + W - 1) / W);
        if (S > 0 && S >= pure_min) {                             // the pure quota grows by what the helpers' slower tiles weigh
            int32_t n_pure = 0, n_rem = 0;
            for (int r = 0; r < R; ++r) {
                const int32_t k = T[x][r] / S;
                n_pure += k;
                n_rem += (T[x][r] - k * S) > 0 ? 1 : 0;
            }
            const int32_t wh0 = W - n_pure;
            if (n_pure > 0 && wh0 > 0 && n_rem > 0) {
                const int32_t k = (n_rem + wh0 - 1) / wh0;
                S += (int32_t)((((int64_t)S * help_pct) / 100 + (int64_t)help_sw * k) * wh0 / (wh0 + n_pure + ((int64_t)n_pure * help_pct) / 100));
            }
        }
        Sx[x] = S;
        int32_t j = 0, acc = 0;
        for (int r = 0; r < R; ++r) {
            pure0[x][r] = j;
            crem[x][r] = acc;
            const int32_t k = (S > 0 && S >= pure_min) ? T[x][r] / S : 0;
            j += k;
            acc += T[x][r] - k * S;
        }
        pure0[x][R] = j;
        crem[x][R] = acc;
        Wh[x] = W - j;
    }
    if (tid == 0) {
        int32_t acc = 0;
        for (int r = 0; r < R; ++r) {
            pp[r] = acc;
            const int32_t cnt = ((skip_mask >> r) & 1ull) ? 0 : rel_ptr[r + 1] - rel_ptr[r];
            acc += (cnt + kSwTile - 1) / kSwTile;
        }
        pp[R] = acc;
    }
    __syncthreads();
    if (tid == 0) {
        int32_t smax = 0;
        for (int x = 0; x < kSwGroups; ++x) smax = max(smax, Sx[x]);
        // queued behind the row index the table was sized by a bound of the rows (2x what they come to): its stride shrinks to the
        // slots the fullest group needs -- empty slots cost the transform launch a ring stage each -- and info[1] tells the host
        s_stride = (dyn != nullptr && smax <= S_cap) ? max(smax, 1) : S_cap;
        s_plain = smax > S_cap ? 1 : 0;                           // a group does not fit the table: plain order (valid as long as
        // the table holds all pp[R] tiles; info[0] = 2 reports a table that cannot even hold those -- rows would go untransformed)
        if (blockIdx.x == 0 && info) {
            info[0] = s_plain ? ((int64_t)pp[R] > (int64_t)kSwGroups * W * S_cap ? 2 : 1) : 0;
            info[1] = smax;
        }
    }
    __syncthreads();
    const int32_t stride = s_stride;
    const int64_t total = (int64_t)kSwGroups * W * stride;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + tid; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int32_t rel = 0, beg = 0, end = 0;
        if (s_plain) {                                            // relation-major tiles, contiguous ranges per workgroup
            if (e < pp[R]) {
                int l = 0, h = R;
                while (h - l > 1) { const int mid = (l + h) >> 1; if (pp[mid] <= (int32_t)e) l = mid; else h = mid; }
                rel = l;
                beg = rel_ptr[rel] + ((int32_t)e - pp[rel]) * kSwTile;
                end = min(beg + kSwTile, rel_ptr[rel + 1]);
            }
        } else {
            const int32_t b = (int32_t)(e / stride), m = (int32_t)(e % stride);
            const int x = b % kSwGroups, j = b / kSwGroups;
            const int32_t S = Sx[x], np = pure0[x][R], RT = crem[x][R], wh = Wh[x];
            int r = -1;
            int32_t n = 0, a_me = 0, s_me = 0;
            if (S > 0 && j < np) {                                // a workgroup of its own relation
                if (m < S) {
                    int l = 0, h = R;                             // last r with pure0[r] <= j (relations without pure workgroups share
                    while (h - l > 1) { const int mid = (l + h) >> 1; if (pure0[x][mid] <= j) l = mid; else h = mid; }   // a value: take the last)
                    r = l; n = m; a_me = S; s_me = j - pure0[x][r];
                }
            } else if (S > 0 && RT > 0 && wh > 0) {               // a helper: a segment of the left-over line
                const int32_t k = j - np;
                const int32_t p0 = (int32_t)(((int64_t)k * RT) / wh), p1 = (int32_t)(((int64_t)(k + 1) * RT) / wh);
                const int32_t pos = p0 + m;
                if (pos < p1) {
                    int l = 0, h = R;                             // last r with crem[r] <= pos (empty left-overs share a value)
                    while (h - l > 1) { const int mid = (l + h) >> 1; if (crem[x][mid] <= pos) l = mid; else h = mid; }
                    r = l;
                    const int32_t c0 = crem[x][r], c1 = crem[x][r + 1];
                    n = pos - max(c0, p0);
                    a_me = min(c1, p1) - max(c0, p0);
                    int32_t kf = 0;                               // first helper that overlaps relation r's left-over
                    while ((int32_t)(((int64_t)(kf + 1) * RT) / wh) <= c0) ++kf;
                    s_me = (pure0[x][r + 1] - pure0[x][r]) + (k - kf);
                }
            }
            if (r >= 0) {
                // rank of event (s_me, n) among all events of relation r: time (2 n' + 1) / (2 a_s'), ties by s'
                const int32_t npr = pure0[x][r + 1] - pure0[x][r];
                const int64_t two_n1 = 2 * (int64_t)n + 1;
                int64_t rank = 0;
                for (int s2 = 0; s2 < npr; ++s2) {
                    const int64_t num = two_n1 * S;
                    const int64_t q = s2 < s_me ? num / a_me : (num - 1) / a_me;
                    rank += (q + 1) / 2;
                }
                if (RT > 0 && wh > 0) {
                    const int32_t c0 = crem[x][r], c1 = crem[x][r + 1];
                    int32_t kf = 0;
                    while ((int32_t)(((int64_t)(kf + 1) * RT) / wh) <= c0 && kf < wh) ++kf;
                    for (int32_t k2 = kf; k2 < wh; ++k2) {
                        const int32_t q0 = (int32_t)(((int64_t)k2 * RT) / wh), q1 = (int32_t)(((int64_t)(k2 + 1) * RT) / wh);
                        if (q0 >= c1) break;
                        const int32_t a2 = min(c1, q1) - max(c0, q0);
                        if (a2 <= 0) continue;
                        const int s2 = npr + (k2 - kf);
                        const int64_t num = two_n1 * a2;
                        const int64_t q = s2 < s_me ? num / a_me : (num - 1) / a_me;
                        rank += (q + 1) / 2;
                    }
                }
                rel = r;
                beg = lo[x][r] + (int32_t)rank * kSwTile;
                end = min(beg + kSwTile, lo[x + 1][r]);
            }
        }
        table[4 * e] = rel; table[4 * e + 1] = beg; table[4 * e + 2] = end; table[4 * e + 3] = 0;
    }
}

struct RowWs {
    int32_t *key, *skey, *iota, *order, *Er, *Dr, *Sr, *mode;
    int32_t *head, *aggh, *tfh, *agge, *tfe, *head_s, *aggh_s, *tfh_s, *agge_s, *tfe_s, *row_rel;
    int32_t *f_key, *f_row, *b_key, *b_row, *perm, *rel_ptr, *totals;
    unsigned int *bitsD, *bitsS; size_t bits_words;
    void* sort_tmp; size_t sort_tmp_bytes;
    void* scan_tmp; size_t scan_tmp_bytes;
    void* csr_ws; size_t csr_ws_bytes;
};
int row_layout(Arena& a, RowWs& w, int64_t N, int64_t R, int64_t E, hipStream_t st) {
    memset(&w, 0, sizeof(w));
    const int64_t L = 2 * E + N;                        // list-entry slots: E edges + E row slots + N self loops
    w.key = a.take<int32_t>(E); w.skey = a.take<int32_t>(E); w.iota = a.take<int32_t>(E); w.order = a.take<int32_t>(E);
    w.Er = a.take<int32_t>(R); w.Dr = a.take<int32_t>(R); w.Sr = a.take<int32_t>(R); w.mode = a.take<int32_t>(R);
    w.head = a.take<int32_t>(E); w.aggh = a.take<int32_t>(E); w.tfh = a.take<int32_t>(E); w.agge = a.take<int32_t>(E);
    w.tfe = a.take<int32_t>(E); w.head_s = a.take<int32_t>(E); w.aggh_s = a.take<int32_t>(E); w.tfh_s = a.take<int32_t>(E);
    w.agge_s = a.take<int32_t>(E); w.tfe_s = a.take<int32_t>(E); w.row_rel = a.take<int32_t>(E);
    w.f_key = a.take<int32_t>(L); w.f_row = a.take<int32_t>(L); w.b_key = a.take<int32_t>(L); w.b_row = a.take<int32_t>(L);
    w.perm = a.take<int32_t>(L);
    w.rel_ptr = a.take<int32_t>(R + 2);
    w.totals = a.take<int32_t>(8);
    w.bits_words = (N * R <= kByteMapMax ? (size_t)(((N + 15) / 16 * 16) * R / 4) : (size_t)((N * R + 31) / 32)) + 4;   // byte map / bitmap
    w.bitsD = a.take<unsigned int>((int64_t)w.bits_words); w.bitsS = a.take<unsigned int>((int64_t)w.bits_words);
    hipError_t e = hipSuccess;
    if (E > 0) {
        e = sort_pairs<int32_t>(nullptr, w.sort_tmp_bytes, w.key, w.skey, w.iota, w.order, E, 32, st);
        if (e == hipSuccess) e = excl_scan(nullptr, w.scan_tmp_bytes, w.head, w.head_s, E, st);
        if (e != hipSuccess) { dn_set_error("rocprim size query failed: %s", hipGetErrorString(e)); return DN_ERR_HIP; }
    }
    w.sort_tmp = a.take_bytes(w.sort_tmp_bytes);
    w.scan_tmp = a.take_bytes(w.scan_tmp_bytes);
    size_t need = 0;
    int rc = csr_build(nullptr, L, N + 1, nullptr, nullptr, nullptr, 0, st, &need);
    if (rc != DN_OK) return rc;
    w.csr_ws_bytes = need;
    w.csr_ws = a.take_bytes(need);
    return DN_OK;
}

}  // namespace

// ----- fixed-width slot tables for dn_rows_selfsum_bf16 ----------------------------------------------------------------------
// Per node v the kept rows of its list are those < num_edge_rows (self-loop rows dropped) outside the dropped range.  cnt <= K:
// slots = the rows, -1 padded.  cnt > K: the first K-1 rows, slot K-1 = -2 and over[v] = 1: dn_overflow_rows_add_bf16 walks
// such a node's list itself after the closing launch (rows K-1 .. cnt-1), screening the byte per node.
__global__ void slot_fill_kernel(int64_t N, int32_t P, int32_t K, const int32_t* __restrict__ lptr,
                                 const int32_t* __restrict__ lrows, int32_t* __restrict__ slots, int32_t drop_beg, int32_t drop_end,
                                 const int32_t* __restrict__ drop_enable, uint8_t* __restrict__ over) {
    const int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (v >= N) return;
    if (drop_enable != nullptr && *drop_enable == 0) drop_beg = drop_end = 0;     // (decided on the device: no host round trip)
    int k = 0;
    for (int i = lptr[v]; i < lptr[v + 1]; ++i) {
        const int r = lrows[i];
        if (r >= P || (r >= drop_beg && r < drop_end)) continue;
        if (k < K) slots[v * (int64_t)K + k] = r;
        ++k;
    }
    for (int j = min(k, K); j < K; ++j) slots[v * (int64_t)K + j] = -1;
    if (k > K) slots[v * (int64_t)K + K - 1] = -2;               // more rows than slots: dn_overflow_rows_add_bf16 adds the rest
    if (over) over[v] = k > K ? 1 : 0;                           // ... and screens this byte per node instead of the 24-byte slot line
}

// ---- tables of a folded pre-aggregation (dn_rows_selfsum_bf16 with fold_info) ---------------------------------------------
// Segment j = nodes seg_nodes[seg_ptr[j] .. seg_ptr[j+1]) (a graph's nodes: the sources of its dummy node).  Valid only if every
// segment is a non-empty contiguous ascending run and the segments ascend; then the (segment, 32-row tile) pairs that share a
// row, numbered segment-major, are consecutive within each tile.
constexpr int kFoldTile = 32;                                    // rows per tile of the closing launch (kSsRows in dn_rel.hip)
constexpr int kFoldInfo = 12;                                    // int32 words per tile record (kFoldInfo in dn_rel.hip)

__global__ void fold_init_kernel(int64_t tiles, int32_t* __restrict__ info) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < tiles * kFoldInfo) info[i] = (i % kFoldInfo) < 8 ? -1 : 0;       // ids 255 ("no segment"), {first, count, 0, 0} = 0
}

__device__ __forceinline__ void fold_seg_span(const int32_t* ptr, const int32_t* nodes, int64_t j, int32_t& first, int32_t& last) {
    first = nodes[ptr[j]];
    last = nodes[ptr[j + 1] - 1];
}

__global__ void fold_count_kernel(int64_t n, int64_t N, const int32_t* __restrict__ ptr, const int32_t* __restrict__ nodes,
                                  int32_t* __restrict__ ntile, int32_t* __restrict__ ok) {
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j > n) return;
    if (j == n) { ntile[n] = 0; return; }
    const int32_t cnt = ptr[j + 1] - ptr[j];
    bool good = cnt > 0;
    int32_t nt = 1;
    if (good) {
        int32_t first, last;
        fold_seg_span(ptr, nodes, j, first, last);
        good = first >= 0 && last < N && last - first == cnt - 1;
        if (good && j > 0 && ptr[j] > ptr[j - 1]) good = nodes[ptr[j] - 1] < first;          // segments ascend
        if (good) nt = last / kFoldTile - first / kFoldTile + 1;
    }
    ntile[j] = nt;
    if (!good) *ok = 0;
}

__global__ void fold_check_kernel(int64_t n, const int32_t* __restrict__ ptr, const int32_t* __restrict__ nodes,
                                  int32_t* __restrict__ ok) {      // inside a segment: strictly +1 (one thread per segment)
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= n) return;
    bool good = true;
    for (int32_t e = ptr[j]; e + 1 < ptr[j + 1]; ++e) good &= nodes[e + 1] == nodes[e] + 1;
    if (!good) *ok = 0;
}

__global__ void fold_fill_kernel(int64_t n, const int32_t* __restrict__ ptr, const int32_t* __restrict__ nodes,
                                 const int32_t* __restrict__ pptr, const int32_t* __restrict__ ok,
                                 int32_t* __restrict__ info) {
    uint8_t* ids = reinterpret_cast<uint8_t*>(info);
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= n || *ok == 0) return;
    int32_t first, last;
    fold_seg_span(ptr, nodes, j, first, last);
    const int32_t t0 = first / kFoldTile, t1 = last / kFoldTile, p = pptr[j];
    // first partial row of tile t0: the earliest segment with a row in it (at most 32 steps back)
    int32_t lo0 = p;
    for (int64_t k = j - 1; k >= 0; --k) {
        int32_t f, l;
        fold_seg_span(ptr, nodes, k, f, l);
        if (l / kFoldTile != t0) break;
        lo0 = pptr[k] + (t0 - f / kFoldTile);
        if (f / kFoldTile != t0) break;                          // started in an earlier tile: nothing before it is in t0
    }
    for (int32_t v = first; v <= last; ++v)
        ids[(size_t)(v / kFoldTile) * (kFoldInfo * 4) + v % kFoldTile] = (uint8_t)(v / kFoldTile == t0 ? p - lo0 : 0);
    // {first partial row, count} of a tile is written by the LAST segment with a row in it
    for (int32_t t = t0; t <= t1; ++t) {
        bool is_last = t < t1;
        if (!is_last) {
            is_last = j + 1 >= n;
            if (!is_last) is_last = nodes[ptr[j + 1]] / kFoldTile > t1;
        }
        if (is_last) {
            const int32_t mine = p + (t - t0), lo = t == t0 ? lo0 : mine;
            info[(size_t)t * kFoldInfo + 8] = lo;
            info[(size_t)t * kFoldInfo + 9] = mine - lo + 1;
        }
    }
}

namespace dn_internal {

// dn_sweep_tables_build_i32 for nd = 1 or 2 tables over the same rows in one launch; dyn[k] != NULL: {relation to skip or -1, go}
// are read on the device instead of skip_mask[k].
int sweep_tables_queue(int32_t num_rels, const int32_t* rel_ptr, const int32_t* row_in, const int32_t* row_out, int64_t num_nodes,
                       int32_t workgroups_per_group, int32_t tiles_per_workgroup, int nd, const uint64_t* skip_mask,
                       const int32_t* const* dyn, int32_t* const* table, int32_t* const* info, hipStream_t st) {
    DN_REQUIRE(num_rels >= 1 && num_rels <= kSwMaxRel, "dn_sweep_tables_build: 1 <= num_rels <= 64");
    DN_REQUIRE(workgroups_per_group >= 1 && workgroups_per_group <= kSwMaxW, "dn_sweep_tables_build: 1 <= workgroups_per_group <= 64");
    DN_REQUIRE(tiles_per_workgroup >= 1 && num_nodes >= 0 && num_nodes < 0x7fffffffLL && (nd == 1 || nd == 2), "dn_sweep_tables_build: bad sizes");
    DN_REQUIRE(rel_ptr && row_in && row_out && table[0] && table[nd - 1], "dn_sweep_tables_build: NULL pointer");
    const int64_t total = (int64_t)kSwGroups * workgroups_per_group * tiles_per_workgroup;
    DN_REQUIRE(total < 0x7fffffffLL, "dn_sweep_tables_build: table too large");
    const int64_t blocks = dn_cdiv(total, 256);
    SwPair pr;
    for (int k = 0; k < 2; ++k) {
        const int q = k < nd ? k : 0;
        pr.d[k] = SwDir{table[q], info ? info[q] : nullptr, dyn ? dyn[q] : nullptr, (unsigned long long)skip_mask[q]};
    }
    hipLaunchKernelGGL(sweep_tables_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024), (unsigned)nd), dim3(256), 0, st, num_rels, rel_ptr,
                       row_in, row_out, (int32_t)num_nodes, workgroups_per_group, tiles_per_workgroup, pr, (int32_t)sw_pure_min(),
                       (int32_t)sw_help_pct(), (int32_t)sw_help_sw());
    DN_CHECK_LAUNCH();
    return DN_OK;
}

}  // namespace dn_internal

extern "C" {

size_t dn_csr_build_workspace_bytes(int64_t M, int64_t num_keys) {
    if (M < 0 || num_keys < 0) { dn_set_error("dn_csr_build_workspace_bytes: negative size"); return 0; }
    size_t need = 0;
    if (csr_build(nullptr, M, num_keys, nullptr, nullptr, nullptr, 0, nullptr, &need) != DN_OK) return 0;
    return need;
}

int dn_csr_build_i32(const int32_t* key, int64_t M, int64_t num_keys, int32_t* ptr, int32_t* perm, void* workspace,
                     size_t workspace_bytes, dn_stream_t stream) {
    DN_REQUIRE(M >= 0 && num_keys >= 0, "dn_csr_build: negative size");
    DN_REQUIRE(M < 0x7fffffffLL && num_keys < 0x7fffffffLL, "dn_csr_build: sizes must fit int32");
    DN_REQUIRE(ptr != nullptr, "dn_csr_build: ptr is NULL");
    DN_REQUIRE(M == 0 || (key && perm && workspace), "dn_csr_build: NULL pointer");
    return csr_build(key, M, num_keys, ptr, perm, workspace, workspace_bytes, (hipStream_t)stream, nullptr);
}

int dn_degrees_i32(int64_t N, int64_t E, const int32_t* src, const int32_t* dst, int32_t* in_deg, int32_t* out_deg,
                   dn_stream_t stream) {
    DN_REQUIRE(N >= 0 && E >= 0, "dn_degrees: negative size");
    hipStream_t st = (hipStream_t)stream;
    if (in_deg) DN_CHECK_HIP(hipMemsetAsync(in_deg, 0, sizeof(int32_t) * (size_t)N, st));
    if (out_deg) DN_CHECK_HIP(hipMemsetAsync(out_deg, 0, sizeof(int32_t) * (size_t)N, st));
    if (E > 0) {
        DN_REQUIRE(src && dst, "dn_degrees: NULL pointer");
        hipLaunchKernelGGL(degrees_kernel, dim3(grid_for(E)), dim3(kBlock), 0, st, E, src, dst, in_deg, out_deg);
        DN_CHECK_LAUNCH();
    }
    return DN_OK;
}

int dn_edge_norm_f32(int32_t mode, int32_t self_loop, int64_t N, int64_t E, const int32_t* src, const int32_t* dst,
                     const int32_t* in_deg, const int32_t* out_deg, float* in_norm, float* out_norm, float* edge_norm,
                     dn_stream_t stream) {
    DN_REQUIRE(mode == 1 || mode == 2, "dn_edge_norm: mode must be 1 (in) or 2 (both)");
    DN_REQUIRE(N >= 0 && E >= 0, "dn_edge_norm: negative size");
    DN_REQUIRE(in_norm == nullptr || in_deg != nullptr, "dn_edge_norm: in_norm needs in_deg");
    DN_REQUIRE(out_norm == nullptr || out_deg != nullptr, "dn_edge_norm: out_norm needs out_deg");
    DN_REQUIRE(edge_norm == nullptr || (in_norm && (mode == 1 || out_norm)), "dn_edge_norm: edge_norm needs node norms");
    hipStream_t st = (hipStream_t)stream;
    if (N > 0 && (in_norm || out_norm))
        hipLaunchKernelGGL(node_norm_kernel, dim3(grid_for(N)), dim3(kBlock), 0, st, self_loop, N, in_deg, out_deg, in_norm,
                           out_norm);
    if (E > 0 && edge_norm) {
        DN_REQUIRE(src && dst, "dn_edge_norm: NULL pointer");
        hipLaunchKernelGGL(edge_norm_kernel, dim3(grid_for(E)), dim3(kBlock), 0, st, mode, E, src, dst, in_norm, out_norm,
                           edge_norm);
    }
    DN_CHECK_LAUNCH();
    return DN_OK;
}

static int dummy_augment(int32_t si, int64_t G, int64_t N, int64_t E, const int32_t* node_ptr, const int32_t* edge_ptr,
                         const int32_t* src, const int32_t* dst, const int32_t* node_id, const int32_t* node_label,
                         const int32_t* edge_id, const int32_t* edge_label, const uint8_t* in_rev, int32_t max_nv,
                         int32_t max_nvl, int32_t max_ne, int32_t max_nel, int32_t* out_node_ptr, int32_t* out_edge_ptr,
                         int32_t* out_src, int32_t* out_dst, int32_t* out_node_id, int32_t* out_node_label,
                         int32_t* out_edge_id, int32_t* out_edge_label, uint8_t* out_dn, uint8_t* out_de, uint8_t* out_rev,
                         hipStream_t st) {
    DN_REQUIRE(G >= 0 && N >= 0 && E >= 0, "dn_dummy_augment: negative size");
    DN_REQUIRE(E + 2 * N < 0x7fffffffLL && N + G < 0x7fffffffLL, "dn_dummy_augment: output sizes must fit int32");
    DN_REQUIRE(node_ptr && edge_ptr && out_node_ptr && out_edge_ptr, "dn_dummy_augment: NULL ptr array");
    hipLaunchKernelGGL(dummy_ptr_kernel, dim3(grid_for(G + 1)), dim3(kBlock), 0, st, G, node_ptr, edge_ptr, out_node_ptr,
                       out_edge_ptr);
    if (G > 0 && N + G > 0) {
        DN_REQUIRE(node_label && out_node_id && out_node_label && out_dn, "dn_dummy_augment: NULL node array");
        hipLaunchKernelGGL(dummy_nodes_kernel, dim3(grid_for(N + G)), dim3(kBlock), 0, st, si, G, N + G, node_ptr, node_id,
                           node_label, max_nv, max_nvl, out_node_id, out_node_label, out_dn);
    }
    if (G > 0 && E + 2 * N > 0) {
        DN_REQUIRE(out_src && out_dst && out_edge_id && out_edge_label && out_de, "dn_dummy_augment: NULL edge array");
        DN_REQUIRE(E == 0 || (src && dst && edge_label), "dn_dummy_augment: NULL input edge array");
        hipLaunchKernelGGL(dummy_edges_kernel, dim3(grid_for(E + 2 * N)), dim3(kBlock), 0, st, si, G, E + 2 * N, node_ptr,
                           edge_ptr, src, dst, edge_id, edge_label, in_rev, max_ne, max_nel, out_src, out_dst, out_edge_id,
                           out_edge_label, out_de, out_rev);
    }
    DN_CHECK_LAUNCH();
    return DN_OK;
}

int dn_dummy_augment_gc_i32(int64_t G, int64_t N, int64_t E, const int32_t* node_ptr, const int32_t* edge_ptr,
                            const int32_t* src, const int32_t* dst, const int32_t* node_label, const int32_t* edge_label,
                            int32_t* out_node_ptr, int32_t* out_edge_ptr, int32_t* out_src, int32_t* out_dst,
                            int32_t* out_node_label, int32_t* out_edge_label, uint8_t* out_is_dummy_node,
                            uint8_t* out_is_dummy_edge, int32_t* out_node_id, int32_t* out_edge_id, dn_stream_t stream) {
    return dummy_augment(0, G, N, E, node_ptr, edge_ptr, src, dst, nullptr, node_label, nullptr, edge_label, nullptr, 0, 0,
                         0, 0, out_node_ptr, out_edge_ptr, out_src, out_dst, out_node_id, out_node_label, out_edge_id,
                         out_edge_label, out_is_dummy_node, out_is_dummy_edge, nullptr, (hipStream_t)stream);
}

int dn_dummy_augment_si_i32(int64_t G, int64_t N, int64_t E, const int32_t* node_ptr, const int32_t* edge_ptr,
                            const int32_t* src, const int32_t* dst, const int32_t* node_id, const int32_t* node_label,
                            const int32_t* edge_id, const int32_t* edge_label, const uint8_t* in_is_reversed,
                            int32_t max_nv, int32_t max_nvl, int32_t max_ne, int32_t max_nel, int32_t* out_node_ptr,
                            int32_t* out_edge_ptr, int32_t* out_src, int32_t* out_dst, int32_t* out_node_id,
                            int32_t* out_node_label, int32_t* out_edge_id, int32_t* out_edge_label,
                            uint8_t* out_is_dummy_node, uint8_t* out_is_dummy_edge, uint8_t* out_is_reversed,
                            dn_stream_t stream) {
    DN_REQUIRE(N == 0 || node_id != nullptr, "dn_dummy_augment_si: node_id is NULL");
    DN_REQUIRE(E == 0 || edge_id != nullptr, "dn_dummy_augment_si: edge_id is NULL");
    DN_REQUIRE(out_is_reversed != nullptr || (E + 2 * N) == 0, "dn_dummy_augment_si: out_is_reversed is NULL");
    return dummy_augment(1, G, N, E, node_ptr, edge_ptr, src, dst, node_id, node_label, edge_id, edge_label, in_is_reversed,
                         max_nv, max_nvl, max_ne, max_nel, out_node_ptr, out_edge_ptr, out_src, out_dst, out_node_id,
                         out_node_label, out_edge_id, out_edge_label, out_is_dummy_node, out_is_dummy_edge, out_is_reversed,
                         (hipStream_t)stream);
}

size_t dn_conjugate_workspace_bytes(int64_t G, int64_t N, int64_t E, int64_t num_raw) {
    if (G < 0 || N < 0 || E < 0) { dn_set_error("dn_conjugate_workspace_bytes: negative size"); return 0; }
    Arena a(nullptr, 0);
    ConjWs w;
    if (conj_layout(a, w, G, N, E, num_raw, nullptr) != DN_OK) return 0;
    return a.off + 256;
}

int dn_conjugate_count_i32(int64_t N, int64_t E, const int32_t* src, const int32_t* dst, int64_t* host_num_raw,
                           void* workspace, size_t workspace_bytes, dn_stream_t stream) {
    DN_REQUIRE(N >= 0 && E >= 0 && host_num_raw, "dn_conjugate_count: bad arguments");
    DN_REQUIRE(E < 0x7fffffffLL && N < 0x7fffffffLL, "dn_conjugate_count: sizes must fit int32");
    hipStream_t st = (hipStream_t)stream;
    if (E == 0) { *host_num_raw = 0; return DN_OK; }
    DN_REQUIRE(src && dst && workspace, "dn_conjugate_count: NULL pointer");
    Arena a(workspace, workspace_bytes);
    ConjWs w;
    int rc = conj_layout(a, w, 0, N, E, -1, st);
    if (rc != DN_OK) return rc;
    if (!a.ok()) { dn_set_error("dn_conjugate_count: workspace too small (%zu < %zu)", workspace_bytes, a.off); return DN_ERR_WORKSPACE; }
    rc = conj_count_phase(w, N, E, src, dst, st);
    if (rc != DN_OK) return rc;
    int32_t total = 0;
    DN_CHECK_HIP(hipMemcpyAsync(&total, w.raw_off + E, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    DN_CHECK_HIP(hipStreamSynchronize(st));
    // NB: a 32-bit scan; overflow shows up as a negative total
    DN_REQUIRE(total >= 0, "dn_conjugate_count: raw 2-path count overflows int32");
    *host_num_raw = total;
    return DN_OK;
}

int dn_conjugate_build_i32(int32_t mode, int64_t G, int64_t N, int64_t E, int64_t T, const int32_t* node_ptr,
                           const int32_t* edge_ptr, const int32_t* src, const int32_t* dst, const int32_t* node_label,
                           const int32_t* edge_id, const uint8_t* is_dummy_edge, int32_t* out_cnode_ptr,
                           int32_t* out_cedge_ptr, int32_t* out_csrc, int32_t* out_cdst, int32_t* out_rep_edge,
                           int32_t* out_shared_node, int64_t* host_counts, void* workspace, size_t workspace_bytes,
                           dn_stream_t stream) {
    DN_REQUIRE(mode == DN_CONJ_GC || mode == DN_CONJ_SI || mode == DN_CONJ_LINE, "dn_conjugate_build: bad mode %d", mode);
    DN_REQUIRE(G >= 0 && N >= 0 && E >= 0 && T >= 0, "dn_conjugate_build: negative size");
    DN_REQUIRE(E < 0x7fffffffLL && N < 0x7fffffffLL && T < 0x7fffffffLL, "dn_conjugate_build: sizes must fit int32");
    DN_REQUIRE(node_ptr && edge_ptr && out_cnode_ptr && out_cedge_ptr && host_counts, "dn_conjugate_build: NULL pointer");
    DN_REQUIRE(mode != DN_CONJ_SI || E == 0 || (edge_id && node_label), "dn_conjugate_build: SI mode needs edge_id and node_label");
    DN_REQUIRE(mode != DN_CONJ_GC || E == 0 || is_dummy_edge, "dn_conjugate_build: GC mode needs is_dummy_edge");
    hipStream_t st = (hipStream_t)stream;
    host_counts[0] = host_counts[1] = 0;
    if (E == 0) {
        DN_CHECK_HIP(hipMemsetAsync(out_cnode_ptr, 0, sizeof(int32_t) * (size_t)(G + 1), st));
        DN_CHECK_HIP(hipMemsetAsync(out_cedge_ptr, 0, sizeof(int32_t) * (size_t)(G + 1), st));
        return DN_OK;
    }
    DN_REQUIRE(src && dst && workspace && out_rep_edge, "dn_conjugate_build: NULL pointer");
    DN_REQUIRE(T == 0 || (out_csrc && out_cdst && out_shared_node), "dn_conjugate_build: NULL output");
    Arena a(workspace, workspace_bytes);
    ConjWs w;
    int rc = conj_layout(a, w, G, N, E, T, st);
    if (rc != DN_OK) return rc;
    if (!a.ok()) { dn_set_error("dn_conjugate_build: workspace too small (%zu < %zu)", workspace_bytes, a.off); return DN_ERR_WORKSPACE; }

    // (1) in-incidence CSR + raw offsets
    rc = conj_count_phase(w, N, E, src, dst, st);
    if (rc != DN_OK) return rc;

    // (2) conj vertices: distinct (graph, conj id), ascending id; representative = first edge with that id
    hipLaunchKernelGGL(edge_graph_kernel, dim3(grid_for(E)), dim3(kBlock), 0, st, G, E, edge_ptr, w.egraph);
    if (mode == DN_CONJ_GC) {
        hipLaunchKernelGGL(fill_i32_kernel, dim3(grid_for(G + 1)), dim3(kBlock), 0, st, w.first_dummy, G + 1, 0x7fffffff);
        hipLaunchKernelGGL(first_dummy_kernel, dim3(grid_for(E)), dim3(kBlock), 0, st, E, w.egraph, edge_ptr, is_dummy_edge,
                           w.first_dummy);
    }
    hipLaunchKernelGGL(vertex_key_kernel, dim3(grid_for(E)), dim3(kBlock), 0, st, mode, E, w.egraph, edge_ptr, edge_id,
                       is_dummy_edge, w.first_dummy, w.vkey);
    hipLaunchKernelGGL(iota_kernel, dim3(grid_for(E)), dim3(kBlock), 0, st, w.iota_e, E);
    DN_CHECK_HIP(sort_pairs<u64>(w.sortE_tmp, w.sortE_tmp_bytes, w.vkey, w.vkey_s, w.iota_e, w.sval_e, E, 64, st));
    hipLaunchKernelGGL(head_flag_u64_kernel, dim3(grid_for(E)), dim3(kBlock), 0, st, w.vkey_s, E, w.vflag);
    {
        size_t tb = w.scan_tmp_bytes;  // sized for E+1 >= E elements
        DN_CHECK_HIP(excl_scan(w.scan_tmp, tb, w.vflag, w.vscan, E, st));
    }
    DN_CHECK_HIP(hipMemsetAsync(w.vcount, 0, sizeof(int32_t) * (size_t)(G + 1), st));
    hipLaunchKernelGGL(vertex_assign_kernel, dim3(grid_for(E)), dim3(kBlock), 0, st, E, w.vkey_s, w.sval_e, w.vflag, w.vscan,
                       w.vmap, out_rep_edge, w.vcount);
    DN_CHECK_HIP(excl_scan(w.scanG_tmp, w.scanG_tmp_bytes, w.vcount, out_cnode_ptr, G + 1, st));

    // (3) raw conj edges, (4) first-occurrence dedupe, (5) order-preserving compaction
    if (T > 0) {
        hipLaunchKernelGGL(raw_fill_kernel, dim3(grid_for(T)), dim3(kBlock), 0, st, E, T, w.raw_off, src, w.in_ptr, w.in_perm,
                           w.vmap, node_label, w.rkey, mode == DN_CONJ_SI ? w.rlabel : nullptr, w.rshared);
        if (mode == DN_CONJ_LINE) {
            hipLaunchKernelGGL(fill_i32_kernel, dim3(grid_for(T)), dim3(kBlock), 0, st, w.keep, T, 1);
        } else {
            hipLaunchKernelGGL(iota_kernel, dim3(grid_for(T)), dim3(kBlock), 0, st, w.rval, T);
            const int32_t* order = w.rval;
            const u64* keys_in = w.rkey;
            if (mode == DN_CONJ_SI) {
                // LSD: stable sort by label first, then by (cu, cv)
                DN_CHECK_HIP(sort_pairs<int32_t>(w.sortL_tmp, w.sortL_tmp_bytes, w.rlabel, w.rlabel_s, w.rval, w.rval_s2, T, 32, st));
                hipLaunchKernelGGL(gather_u64_kernel, dim3(grid_for(T)), dim3(kBlock), 0, st, w.rkey, w.rval_s2, T, w.rkey_g);
                order = w.rval_s2;
                keys_in = w.rkey_g;
            }
            DN_CHECK_HIP(sort_pairs<u64>(w.sortT_tmp, w.sortT_tmp_bytes, keys_in, w.rkey_s, order, w.rval_s, T, 64, st));
            hipLaunchKernelGGL(keep_flag_kernel, dim3(grid_for(T)), dim3(kBlock), 0, st, mode, T, w.rkey_s, w.rval_s,
                               mode == DN_CONJ_SI ? w.rlabel : nullptr, out_rep_edge, is_dummy_edge, w.keep);
        }
        DN_CHECK_HIP(excl_scan(w.scanT_tmp, w.scanT_tmp_bytes, w.keep, w.kscan, T, st));
        hipLaunchKernelGGL(compact_kernel, dim3(grid_for(T)), dim3(kBlock), 0, st, T, w.rkey, w.rshared, w.keep, w.kscan,
                           out_csrc, out_cdst, out_shared_node);
    }
    hipLaunchKernelGGL(cedge_ptr_kernel, dim3(grid_for(G + 1)), dim3(kBlock), 0, st, G, E, T, edge_ptr, w.raw_off, w.kscan,
                       w.keep, out_cedge_ptr);
    DN_CHECK_LAUNCH();
    int32_t counts[2] = {0, 0};
    DN_CHECK_HIP(hipMemcpyAsync(&counts[0], out_cnode_ptr + G, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    DN_CHECK_HIP(hipMemcpyAsync(&counts[1], out_cedge_ptr + G, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    DN_CHECK_HIP(hipStreamSynchronize(st));
    host_counts[0] = counts[0];
    host_counts[1] = counts[1];
    return DN_OK;
}

size_t dn_rel_index_workspace_bytes(int64_t N, int64_t R, int64_t E) {
    if (N < 0 || R < 0 || E < 0) { dn_set_error("dn_rel_index_workspace_bytes: negative size"); return 0; }
    Arena a(nullptr, 0);
    RelWs w;
    if (rel_layout(a, w, N, R, E, nullptr) != DN_OK) return 0;
    return a.off + 256;
}

int dn_rel_index_build_i32(int64_t N, int64_t R, int64_t E, const int32_t* src, const int32_t* dst, const int32_t* etype,
                           int32_t* perm1, int32_t* src1, int32_t* seg_ptr, int32_t* seg_dst, int32_t* rel_ptr,
                           int32_t* dptr, int32_t* sperm, int32_t* optr, int32_t* operm, int32_t* seg_by_src,
                           int64_t* host_P, int32_t* host_rel_ptr, void* workspace, size_t workspace_bytes,
                           dn_stream_t stream) {
    DN_REQUIRE(N >= 0 && R >= 1 && E >= 0, "dn_rel_index_build: bad sizes");
    DN_REQUIRE(N * R < 0x7fffffffLL && E < 0x7fffffffLL, "dn_rel_index_build: N*R and E must fit int32");
    DN_REQUIRE(seg_ptr && rel_ptr && dptr && optr && host_P && host_rel_ptr, "dn_rel_index_build: NULL pointer");
    DN_REQUIRE(E == 0 || (src && dst && etype && perm1 && src1 && seg_dst && sperm && operm && seg_by_src && workspace),
               "dn_rel_index_build: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    Arena a(workspace, workspace_bytes);
    RelWs w;
    int rc = rel_layout(a, w, N, R, E, st);
    if (rc != DN_OK) return rc;
    if (workspace && !a.ok()) { dn_set_error("dn_rel_index_build: workspace too small (%zu < %zu)", workspace_bytes, a.off); return DN_ERR_WORKSPACE; }
    *host_P = 0;
    if (E == 0) {
        DN_CHECK_HIP(hipMemsetAsync(seg_ptr, 0, sizeof(int32_t), st));
        DN_CHECK_HIP(hipMemsetAsync(rel_ptr, 0, sizeof(int32_t) * (size_t)(R + 1), st));
        DN_CHECK_HIP(hipMemsetAsync(dptr, 0, sizeof(int32_t) * (size_t)(N + 1), st));
        DN_CHECK_HIP(hipMemsetAsync(optr, 0, sizeof(int32_t) * (size_t)(N + 1), st));
        memset(host_rel_ptr, 0, sizeof(int32_t) * (size_t)(R + 1));
        return DN_OK;
    }
    // (1) stable sort of edges by (rel, dst)
    hipLaunchKernelGGL(rel_key_kernel, dim3(grid_for(E)), dim3(kBlock), 0, st, E, N, dst, etype, w.key);
    hipLaunchKernelGGL(iota_kernel, dim3(grid_for(E)), dim3(kBlock), 0, st, w.iota, E);
    DN_CHECK_HIP(sort_pairs<int32_t>(w.sort_tmp, w.sort_tmp_bytes, w.key, w.skey, w.iota, perm1, E,
                                     bits_for((u64)(N * R > 0 ? N * R - 1 : 0)), st));
    // (2) segments = runs of equal key
    hipLaunchKernelGGL(head_flag_i32_kernel, dim3(grid_for(E)), dim3(kBlock), 0, st, w.skey, E, w.flag);
    DN_CHECK_HIP(excl_scan(w.scan_tmp, w.scan_tmp_bytes, w.flag, w.scan, E, st));
    hipLaunchKernelGGL(rel_segments_kernel, dim3(grid_for(E + 1)), dim3(kBlock), 0, st, E, N, w.skey, perm1, w.flag, w.scan,
                       src, src1, seg_ptr, seg_dst, w.seg_rel, w.seg_of_edge);
    int32_t last[2] = {0, 0};
    DN_CHECK_HIP(hipMemcpyAsync(&last[0], w.scan + (E - 1), sizeof(int32_t), hipMemcpyDeviceToHost, st));
    DN_CHECK_HIP(hipMemcpyAsync(&last[1], w.flag + (E - 1), sizeof(int32_t), hipMemcpyDeviceToHost, st));
    DN_CHECK_HIP(hipStreamSynchronize(st));
    const int64_t P = (int64_t)last[0] + last[1];
    *host_P = P;
    // (3) relation ranges over segments, (4) segments grouped by dst, (5) edges grouped by src
    hipLaunchKernelGGL(ptr_from_sorted_kernel, dim3(grid_for(P + 1)), dim3(kBlock), 0, st, w.seg_rel, P, R, rel_ptr);
    rc = csr_build(seg_dst, P, N, dptr, sperm, w.csr_ws, w.csr_ws_bytes, st, nullptr);
    if (rc != DN_OK) return rc;
    rc = csr_build(src, E, N, optr, operm, w.csr_ws, w.csr_ws_bytes, st, nullptr);
    if (rc != DN_OK) return rc;
    hipLaunchKernelGGL(gather_i32_kernel, dim3(grid_for(E)), dim3(kBlock), 0, st, w.seg_of_edge, operm, E, seg_by_src);
    DN_CHECK_LAUNCH();
    DN_CHECK_HIP(hipMemcpyAsync(host_rel_ptr, rel_ptr, sizeof(int32_t) * (size_t)(R + 1), hipMemcpyDeviceToHost, st));
    DN_CHECK_HIP(hipStreamSynchronize(st));
    return DN_OK;
}

size_t dn_row_index_workspace_bytes(int64_t N, int64_t R, int64_t E) {
    if (N < 0 || R < 0 || E < 0) { dn_set_error("dn_row_index_workspace_bytes: negative size"); return 0; }
    Arena a(nullptr, 0);
    RowWs w;
    if (row_layout(a, w, N, R, E, nullptr) != DN_OK) return 0;
    return a.off + 256;
}

int dn_row_index_build_i32(int64_t N, int64_t R, int64_t E, const int32_t* src, const int32_t* dst, const int32_t* etype,
                           int32_t self_loop, float edge_frac, int32_t* row_in, int32_t* row_out, int32_t* aux_f_ptr,
                           int32_t* aux_f_idx, int32_t* aux_b_ptr, int32_t* aux_b_idx, int32_t* dst_ptr, int32_t* dst_rows,
                           int32_t* src_ptr, int32_t* src_rows, int64_t* host_counts, int32_t* host_rel_ptr,
                           int32_t* host_modes, void* workspace, size_t workspace_bytes, dn_stream_t stream) {
    DN_REQUIRE(N >= 0 && R >= 1 && E >= 0, "dn_row_index_build: bad sizes");
    DN_REQUIRE(N * R < 0x7fffffffLL && 2 * E + N < 0x7fffffffLL, "dn_row_index_build: N*R and 2E+N must fit int32");
    DN_REQUIRE(row_in && row_out && aux_f_ptr && aux_b_ptr && dst_ptr && dst_rows && src_ptr && src_rows && host_counts &&
               host_rel_ptr && host_modes && workspace, "dn_row_index_build: NULL pointer");
    DN_REQUIRE(E == 0 || (src && dst && etype && aux_f_idx && aux_b_idx), "dn_row_index_build: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    Arena a(workspace, workspace_bytes);
    RowWs w;
    int rc = row_layout(a, w, N, R, E, st);
    if (rc != DN_OK) return rc;
    if (!a.ok()) { dn_set_error("dn_row_index_build: workspace too small (%zu < %zu)", workspace_bytes, a.off); return DN_ERR_WORKSPACE; }
    const int kb = bits_for((u64)(N * R > 0 ? N * R - 1 : 0));
    DN_CHECK_HIP(hipMemsetAsync(w.Er, 0, sizeof(int32_t) * (size_t)R, st));
    DN_CHECK_HIP(hipMemsetAsync(w.Dr, 0, sizeof(int32_t) * (size_t)R, st));
    DN_CHECK_HIP(hipMemsetAsync(w.Sr, 0, sizeof(int32_t) * (size_t)R, st));
    int64_t P = 0, n_agg = 0, n_tf = 0, n_agg_e = 0, n_tf_e = 0;
    if (E > 0) {
        // (1) per relation: #edges, #distinct destinations, #distinct sources -> EDGE / AGG / TF   (bitmaps, no sort)
        DN_CHECK_HIP(hipMemsetAsync(w.bitsD, 0, sizeof(unsigned int) * w.bits_words, st));
        DN_CHECK_HIP(hipMemsetAsync(w.bitsS, 0, sizeof(unsigned int) * w.bits_words, st));
        const int64_t sb = dn_cdiv(E, kBlock);
        if (N * R <= kByteMapMax && R <= 65535) {
            const int64_t Np = (N + 15) / 16 * 16;
            hipLaunchKernelGGL(ri_stats_mark_kernel, dim3((unsigned)(sb < 4096 ? sb : 4096)), dim3(kBlock), 0, st, E, Np, (int32_t)R, src,
                               dst, etype, (uint8_t*)w.bitsD, (uint8_t*)w.bitsS, w.Er);
            const int64_t slices = dn_cdiv(Np / 16, 4 * kBlock);
            hipLaunchKernelGGL(ri_stats_count_kernel, dim3((unsigned)(slices < 64 ? slices : 64), (unsigned)R, 2), dim3(kBlock), 0, st, Np,
                               (const uint8_t*)w.bitsD, (const uint8_t*)w.bitsS, w.Dr, w.Sr);
        } else {
            hipLaunchKernelGGL(ri_stats_bitmap_kernel, dim3((unsigned)(sb < 2048 ? sb : 2048)), dim3(kBlock), 0, st, E, N, (int32_t)R,
                               src, dst, etype, w.bitsD, w.bitsS, w.Er, w.Dr, w.Sr);
        }
        hipLaunchKernelGGL(iota_kernel, dim3(grid_for(E)), dim3(kBlock), 0, st, w.iota, E);
    }
    hipLaunchKernelGGL(ri_mode_kernel, dim3(grid_for(R)), dim3(kBlock), 0, st, R, edge_frac, w.Er, w.Dr, w.Sr, w.mode);
    if (E > 0) {
        // (2) rows: stable sort by (rel, dst | src), heads, scans
        hipLaunchKernelGGL(ri_rowkey_kernel, dim3(grid_for(E)), dim3(kBlock), 0, st, E, N, src, dst, etype, w.mode, w.key);
        DN_CHECK_HIP(sort_pairs<int32_t>(w.sort_tmp, w.sort_tmp_bytes, w.key, w.skey, w.iota, w.order, E, kb, st));
        hipLaunchKernelGGL(ri_flags_kernel, dim3(grid_for(E)), dim3(kBlock), 0, st, E, N, w.skey, w.mode, w.head, w.aggh, w.tfh,
                           w.agge, w.tfe);
        DN_CHECK_HIP(excl_scan(w.scan_tmp, w.scan_tmp_bytes, w.head, w.head_s, E, st));
        DN_CHECK_HIP(excl_scan(w.scan_tmp, w.scan_tmp_bytes, w.aggh, w.aggh_s, E, st));
        DN_CHECK_HIP(excl_scan(w.scan_tmp, w.scan_tmp_bytes, w.tfh, w.tfh_s, E, st));
        DN_CHECK_HIP(excl_scan(w.scan_tmp, w.scan_tmp_bytes, w.agge, w.agge_s, E, st));
        DN_CHECK_HIP(excl_scan(w.scan_tmp, w.scan_tmp_bytes, w.tfe, w.tfe_s, E, st));
        int32_t tot[5];
        hipLaunchKernelGGL(ri_totals_kernel, dim3(1), dim3(64), 0, st, E, w.head_s, w.head, w.aggh_s, w.aggh, w.tfh_s, w.tfh, w.agge_s,
                           w.agge, w.tfe_s, w.tfe, w.totals);
        DN_CHECK_HIP(hipMemcpyAsync(tot, w.totals, sizeof(tot), hipMemcpyDeviceToHost, st));
        DN_CHECK_HIP(hipStreamSynchronize(st));
        P = tot[0]; n_agg = tot[1]; n_tf = tot[2]; n_agg_e = tot[3]; n_tf_e = tot[4];
        hipLaunchKernelGGL(ri_fill_kernel, dim3(grid_for(E)), dim3(kBlock), 0, st, E, N, E - n_agg_e, E - n_tf_e, w.skey, w.order,
                           w.mode, src, dst, w.head, w.head_s, w.aggh_s, w.tfh_s, w.agge_s, w.tfe_s, row_in, row_out, w.row_rel,
                           aux_f_idx, aux_f_ptr, aux_b_idx, aux_b_ptr, w.f_key, w.f_row, w.b_key, w.b_row);
    }
    const int64_t f_self = E - n_agg_e + n_agg, b_self = E - n_tf_e + n_tf;       // list entries before the self loops
    hipLaunchKernelGGL(ri_tail_kernel, dim3(grid_for(N + 1)), dim3(kBlock), 0, st, N, P, n_agg, n_tf, n_agg_e, n_tf_e, f_self, b_self,
                       self_loop, row_in, row_out, aux_f_ptr, aux_b_ptr, w.f_key, w.f_row, w.b_key, w.b_row);
    // (3) per-node lists of contributing rows.  N + 1 keys (the last segment stays empty), so dst_ptr / src_ptr receive N + 2
    // entries as before
    const int64_t n_f = f_self + (self_loop ? N : 0), n_b = b_self + (self_loop ? N : 0);
    if (n_f > 0) {
        rc = csr_build(w.f_key, n_f, N + 1, dst_ptr, w.perm, w.csr_ws, w.csr_ws_bytes, st, nullptr);
        if (rc != DN_OK) return rc;
        hipLaunchKernelGGL(gather_i32_kernel, dim3(grid_for(n_f)), dim3(kBlock), 0, st, w.f_row, w.perm, n_f, dst_rows);
    } else {
        DN_CHECK_HIP(hipMemsetAsync(dst_ptr, 0, sizeof(int32_t) * (size_t)(N + 2), st));
    }
    if (n_b > 0) {
        rc = csr_build(w.b_key, n_b, N + 1, src_ptr, w.perm, w.csr_ws, w.csr_ws_bytes, st, nullptr);
        if (rc != DN_OK) return rc;
        hipLaunchKernelGGL(gather_i32_kernel, dim3(grid_for(n_b)), dim3(kBlock), 0, st, w.b_row, w.perm, n_b, src_rows);
    } else {
        DN_CHECK_HIP(hipMemsetAsync(src_ptr, 0, sizeof(int32_t) * (size_t)(N + 2), st));
    }
    // (4) relation ranges over rows (edge rows are relation-major; the caller appends the self-loop rows as relation R)
    int32_t* rel_ptr_dev = w.rel_ptr;
    hipLaunchKernelGGL(ptr_from_sorted_kernel, dim3(grid_for(P + 1)), dim3(kBlock), 0, st, w.row_rel, P, R, rel_ptr_dev);
    DN_CHECK_LAUNCH();
    DN_CHECK_HIP(hipMemcpyAsync(host_rel_ptr, rel_ptr_dev, sizeof(int32_t) * (size_t)(R + 1), hipMemcpyDeviceToHost, st));
    DN_CHECK_HIP(hipMemcpyAsync(host_modes, w.mode, sizeof(int32_t) * (size_t)R, hipMemcpyDeviceToHost, st));
    DN_CHECK_HIP(hipStreamSynchronize(st));
    host_counts[0] = P; host_counts[1] = n_agg; host_counts[2] = n_tf; host_counts[3] = n_agg_e; host_counts[4] = n_tf_e;
    return DN_OK;
}

int dn_sweep_tables_build_i32(int32_t num_rels, const int32_t* rel_ptr, const int32_t* row_in, const int32_t* row_out,
                              int64_t num_nodes, int32_t workgroups_per_group, int32_t tiles_per_workgroup, uint64_t skip_mask,
                              int32_t* table, int32_t* info, dn_stream_t stream) {
    DN_REQUIRE(table, "dn_sweep_tables_build: NULL pointer");
    return dn_internal::sweep_tables_queue(num_rels, rel_ptr, row_in, row_out, num_nodes, workgroups_per_group, tiles_per_workgroup, 1,
                                           &skip_mask, nullptr, &table, &info, (hipStream_t)stream);
}

int dn_row_tables_build_i32(int32_t num_rels, const int32_t* rel_ptr, int32_t step, int64_t max_entries, int32_t* table,
                            int32_t* piece_ptr, uint64_t skip_mask, dn_stream_t stream) {
    DN_REQUIRE(num_rels >= 1 && num_rels <= 1024, "dn_row_tables_build: 1 <= num_rels <= 1024");
    DN_REQUIRE(step >= 1 && max_entries >= 0, "dn_row_tables_build: bad sizes");
    DN_REQUIRE(rel_ptr && (max_entries == 0 || table), "dn_row_tables_build: NULL pointer");
    const int64_t blocks = max_entries > 0 ? dn_cdiv(max_entries, 256) : 1;
    hipLaunchKernelGGL(row_tables_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, (hipStream_t)stream,
                       num_rels, rel_ptr, step, max_entries, table, piece_ptr, (unsigned long long)skip_mask);
    DN_CHECK_LAUNCH();
    return DN_OK;
}

int dn_slot_table_build_i32(int64_t N, int32_t num_edge_rows, int32_t K, const int32_t* list_ptr, const int32_t* list_rows,
                            int32_t drop_beg, int32_t drop_end, const int32_t* drop_enable, int32_t* slots, uint8_t* overflow,
                            dn_stream_t stream) {
    DN_REQUIRE(N >= 0 && K >= 2 && num_edge_rows >= 0, "dn_slot_table_build: bad sizes");
    if (N == 0) return DN_OK;
    DN_REQUIRE(list_ptr && list_rows && slots, "dn_slot_table_build: NULL pointer");
    hipLaunchKernelGGL(slot_fill_kernel, dim3(grid_for(N)), dim3(kBlock), 0, (hipStream_t)stream, N, num_edge_rows, K, list_ptr,
                       list_rows, slots, drop_beg, drop_end, drop_enable, overflow);
    DN_CHECK_LAUNCH();
    return DN_OK;
}

size_t dn_fold_tables_workspace_bytes(int64_t num_segments) {
    if (num_segments < 0) { dn_set_error("dn_fold_tables_workspace_bytes: negative size"); return 0; }
    Arena a(nullptr, 0);
    a.take<int32_t>(num_segments + 1); a.take<int32_t>(1);
    size_t tb = 0;
    if (excl_scan(nullptr, tb, nullptr, nullptr, num_segments + 1, nullptr) != hipSuccess) { dn_set_error("rocprim scan size query failed"); return 0; }
    a.take_bytes(tb);
    return a.off + 256;
}

int dn_fold_tables_build_async_i32(int64_t N, int64_t num_segments, const int32_t* seg_ptr, const int32_t* seg_nodes,
                                   int32_t* fold_info, int32_t* part_ptr, int32_t* dev_ok, void* workspace, size_t workspace_bytes,
                                   dn_stream_t stream) {
    DN_REQUIRE(N >= 0 && num_segments >= 0 && N < INT32_MAX && num_segments < INT32_MAX, "dn_fold_tables_build: bad sizes");
    DN_REQUIRE(dev_ok, "dn_fold_tables_build: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    if (N == 0 || num_segments == 0) { DN_CHECK_HIP(hipMemsetAsync(dev_ok, 0, sizeof(int32_t), st)); return DN_OK; }
    DN_REQUIRE(seg_ptr && seg_nodes && fold_info && part_ptr && workspace, "dn_fold_tables_build: NULL pointer");
    Arena a(workspace, workspace_bytes);
    int32_t* ntile = a.take<int32_t>(num_segments + 1);
    a.take<int32_t>(1);
    size_t tb = 0;
    DN_CHECK_HIP(excl_scan(nullptr, tb, ntile, part_ptr, num_segments + 1, st));
    void* temp = a.take_bytes(tb);
    if (!a.ok()) { dn_set_error("dn_fold_tables_build: workspace too small"); return DN_ERR_WORKSPACE; }
    const int64_t tiles = dn_cdiv(N, kFoldTile);
    DN_CHECK_HIP(hipMemsetAsync(dev_ok, 0x01, sizeof(int32_t), st));             // any non-zero value: "still valid"
    hipLaunchKernelGGL(fold_init_kernel, dim3(grid_for(tiles * kFoldInfo)), dim3(kBlock), 0, st, tiles, fold_info);
    DN_CHECK_LAUNCH();
    hipLaunchKernelGGL(fold_count_kernel, dim3(grid_for(num_segments + 1)), dim3(kBlock), 0, st, num_segments, N, seg_ptr, seg_nodes,
                       ntile, dev_ok);
    DN_CHECK_LAUNCH();
    hipLaunchKernelGGL(fold_check_kernel, dim3(grid_for(num_segments)), dim3(kBlock), 0, st, num_segments, seg_ptr, seg_nodes, dev_ok);
    DN_CHECK_LAUNCH();
    DN_CHECK_HIP(excl_scan(temp, tb, ntile, part_ptr, num_segments + 1, st));
    hipLaunchKernelGGL(fold_fill_kernel, dim3(grid_for(num_segments)), dim3(kBlock), 0, st, num_segments, seg_ptr, seg_nodes, part_ptr,
                       dev_ok, fold_info);
    DN_CHECK_LAUNCH();
    return DN_OK;
}

int dn_fold_tables_build_i32(int64_t N, int64_t num_segments, const int32_t* seg_ptr, const int32_t* seg_nodes,
                             int32_t* fold_info, int32_t* part_ptr, int32_t* host_ok, void* workspace, size_t workspace_bytes,
                             dn_stream_t stream) {
    DN_REQUIRE(N >= 0 && num_segments >= 0 && N < INT32_MAX && num_segments < INT32_MAX, "dn_fold_tables_build: bad sizes");
    DN_REQUIRE(host_ok, "dn_fold_tables_build: NULL pointer");
    *host_ok = 0;
    if (N == 0 || num_segments == 0) return DN_OK;
    DN_REQUIRE(workspace, "dn_fold_tables_build: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    Arena a(workspace, workspace_bytes);
    a.take<int32_t>(num_segments + 1);
    int32_t* ok = a.take<int32_t>(1);                                              // (same layout as the async entry point)
    int rc = dn_fold_tables_build_async_i32(N, num_segments, seg_ptr, seg_nodes, fold_info, part_ptr, ok, workspace, workspace_bytes,
                                            stream);
    if (rc != DN_OK) return rc;
    int32_t h = 0;
    DN_CHECK_HIP(hipMemcpyAsync(&h, ok, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    DN_CHECK_HIP(hipStreamSynchronize(st));
    *host_ok = h != 0 ? 1 : 0;
    return DN_OK;
}

}  // extern "C"
