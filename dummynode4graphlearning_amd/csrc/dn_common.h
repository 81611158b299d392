// Shared helpers for the dn_hip C-ABI library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define DN_OK 0
#define DN_ERR_ARG (-1)
#define DN_ERR_HIP (-2)
#define DN_ERR_WORKSPACE (-3)
#define DN_ERR_UNSUPPORTED (-4)

// thread-local last-error string (dn_error.cpp)
void dn_set_error(const char* fmt, ...);

#define DN_CHECK_HIP(expr)                                                              \
    do {                                                                                \
        hipError_t dn_e_ = (expr);                                                      \
        if (dn_e_ != hipSuccess) {                                                      \
            dn_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,                  \
                         hipGetErrorString(dn_e_));                                     \
            return DN_ERR_HIP;                                                          \
        }                                                                               \
    } while (0)

#define DN_REQUIRE(cond, ...)                                                           \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            dn_set_error(__VA_ARGS__);                                                  \
            return DN_ERR_ARG;                                                          \
        }                                                                               \
    } while (0)

#define DN_CHECK_LAUNCH() DN_CHECK_HIP(hipGetLastError())

static inline int64_t dn_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t dn_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Experiment knobs.  The shipped library has NO environment access of its own: every knob is its compile-time default (the
// measured best).  A tuning build (-DDN_TUNING_ENV: `python -m dummynode4graphlearning_amd.csrc.build --tuning`, used by
// tools/ab.sh) reads the DN_* variables instead.
#ifdef DN_TUNING_ENV
#include <stdlib.h>
static inline int dn_knob(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
#else
static constexpr int dn_knob(const char*, int dflt) { return dflt; }
#endif

// MI355X: 8 XCDs, workgroups are dealt round-robin over them (blocks b and b+8 share an XCD's L2).
// Map the hardware block id to a logical chunk id so that the blocks of one XCD walk a CONTIGUOUS
// range of chunks (neighbouring graphs -> same L2).  Speed only, never correctness.
#define DN_NUM_XCD 8
__device__ __forceinline__ int64_t dn_xcd_chunk(int64_t b, int64_t nblocks) {
    const int64_t per = (nblocks + DN_NUM_XCD - 1) / DN_NUM_XCD;
    return (b % DN_NUM_XCD) * per + b / DN_NUM_XCD;
}

// LDS-DMA issued from inline asm (cdna_hip_programming.md, inline-asm section: M0 written in the statement that reads
// it): hipcc's s_waitcnt pass treats a builtin LDS-DMA as a pending LDS write and drains vmcnt(0) before EVERY later
// ds_read, which would empty the ring each tile; an asm DMA is outside its bookkeeping and is counted by hand below.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

__device__ __forceinline__ void glds4(const void* gsrc, unsigned lds_dst) {      // 4 bytes per lane: lane l lands at lds_dst + 4 l
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

__device__ __forceinline__ void glds4_sc1(const void* gsrc, unsigned lds_dst) {  // ... past this CU's L1 (a word another workgroup updates)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off sc1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// A wave's slice of a weight matrix stored [k][n] (the layout of the reference's parameters) as MFMA A-operand fragments, which
// want 8 consecutive k per lane: one 32-k-row slab at a time goes global -> registers (coalesced 16-byte pieces, all slabs
// requested up front) -> a wave-private LDS scratch -> ds_read_b64_tr_b16 (the hardware transpose).  2-byte gathers instead
// cost ~30 us per call: hipcc serialises the 128 partial-register loads of a wave behind vmcnt(0) waits.
//   NC = 32: the wave owns columns n0 .. n0+31, fragment [ks][n] row i <-> column n0 + 8 (i >> 2) + 4 n + (i & 3)  (ring kernels)
//   NC = 16: columns n0 .. n0+15, fragment [ks] row i <-> column n0 + i
// scratch: 32 x 2 NC bytes, 16-byte aligned, private to the wave.  KS = K / 32.
typedef short dn_short4v __attribute__((ext_vector_type(4)));
typedef short dn_short8v __attribute__((ext_vector_type(8)));
typedef __bf16 dn_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int dn_u32x4 __attribute__((ext_vector_type(4)));

template <int KS>
__device__ __forceinline__ void dn_load_w_kn32(const __bf16* __restrict__ w, int ldw, int n0, int lane, char* scratch,
                                               dn_bf16x8 (&wf)[KS][2]) {
    typedef dn_short4v __attribute__((address_space(3))) * lds_tr;
    const int r16 = lane >> 2, pc4 = lane & 3;
    dn_u32x4 raw[KS][2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h)
            raw[ks][h] = *reinterpret_cast<const dn_u32x4*>(w + (size_t)(32 * ks + 16 * h + r16) * ldw + n0 + 8 * pc4);
    const int g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        *reinterpret_cast<dn_u32x4*>(scratch + r16 * 64 + 16 * pc4) = raw[ks][0];
        *reinterpret_cast<dn_u32x4*>(scratch + (16 + r16) * 64 + 16 * pc4) = raw[ks][1];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const char* a0 = scratch + (8 * g + q4) * 64 + 16 * p4 + 8 * n;
            const dn_short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(a0));
            const dn_short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(a0 + 4 * 64));
            const dn_short8v f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            wf[ks][n] = __builtin_bit_cast(dn_bf16x8, f);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// (the same, KC k-slabs per round trip instead of all KS at once: 8 KC staging registers instead of 8 KS -- for a reload in
//  the middle of a kernel that has no registers to spare; KS / KC dependent round trips)
template <int KS, int KC>
__device__ __forceinline__ void dn_load_w_kn32_lean(const __bf16* __restrict__ w, int ldw, int n0, int lane, char* scratch,
                                                    dn_bf16x8 (&wf)[KS][2]) {
    typedef dn_short4v __attribute__((address_space(3))) * lds_tr;
    const int r16 = lane >> 2, pc4 = lane & 3;
    const int g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
#pragma unroll
    for (int k0 = 0; k0 < KS; k0 += KC) {
        dn_u32x4 raw[KC][2];
#pragma unroll
        for (int kk = 0; kk < KC; ++kk)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                raw[kk][h] = *reinterpret_cast<const dn_u32x4*>(w + (size_t)(32 * (k0 + kk) + 16 * h + r16) * ldw + n0 + 8 * pc4);
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
            *reinterpret_cast<dn_u32x4*>(scratch + r16 * 64 + 16 * pc4) = raw[kk][0];
            *reinterpret_cast<dn_u32x4*>(scratch + (16 + r16) * 64 + 16 * pc4) = raw[kk][1];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const char* a0 = scratch + (8 * g + q4) * 64 + 16 * p4 + 8 * n;
                const dn_short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(a0));
                const dn_short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(a0 + 4 * 64));
                const dn_short8v f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                wf[k0 + kk][n] = __builtin_bit_cast(dn_bf16x8, f);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// (the same with the PLAIN column order of the dense-row kernels: fragment [ks][n] row i <-> column n0 + 16 n + i)
template <int KS>
__device__ __forceinline__ void dn_load_w_kn32p(const __bf16* __restrict__ w, int ldw, int n0, int lane, char* scratch,
                                                dn_bf16x8 (&wf)[KS][2]) {
    typedef dn_short4v __attribute__((address_space(3))) * lds_tr;
    const int r16 = lane >> 2, pc4 = lane & 3;
    dn_u32x4 raw[KS][2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h)
            raw[ks][h] = *reinterpret_cast<const dn_u32x4*>(w + (size_t)(32 * ks + 16 * h + r16) * ldw + n0 + 8 * pc4);
    const int g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        *reinterpret_cast<dn_u32x4*>(scratch + r16 * 64 + 16 * pc4) = raw[ks][0];
        *reinterpret_cast<dn_u32x4*>(scratch + (16 + r16) * 64 + 16 * pc4) = raw[ks][1];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const char* a0 = scratch + (8 * g + q4) * 64 + 32 * n + 8 * p4;
            const dn_short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(a0));
            const dn_short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(a0 + 4 * 64));
            const dn_short8v f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            wf[ks][n] = __builtin_bit_cast(dn_bf16x8, f);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int KS>
__device__ __forceinline__ void dn_load_w_kn16(const __bf16* __restrict__ w, int ldw, int n0, int lane, char* scratch,
                                               dn_bf16x8 (&wf)[KS]) {
    typedef dn_short4v __attribute__((address_space(3))) * lds_tr;
    const int r32 = lane >> 1, pc2 = lane & 1;
    dn_u32x4 raw[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) raw[ks] = *reinterpret_cast<const dn_u32x4*>(w + (size_t)(32 * ks + r32) * ldw + n0 + 8 * pc2);
    const int g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        *reinterpret_cast<dn_u32x4*>(scratch + r32 * 32 + 16 * pc2) = raw[ks];
        __builtin_amdgcn_wave_barrier();
        const char* a0 = scratch + (8 * g + q4) * 32 + 8 * p4;
        const dn_short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(a0));
        const dn_short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(a0 + 4 * 32));
        const dn_short8v f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        wf[ks] = __builtin_bit_cast(dn_bf16x8, f);
        __builtin_amdgcn_wave_barrier();
    }
}

// Tiles = the graphs of a batch (segment-complete tiles for the absorbed fold of dn_rows_close_bf16), one segment per call: segment j
// = the nodes seg_nodes[seg_ptr[j] .. seg_ptr[j+1]); block j = [first node of segment j (0 for j = 0), first node of segment j + 1
// (N for the last)).  Clears *ok unless the segment is a non-empty contiguous ascending run, the next segment starts behind it,
// the block has at most 32 nodes and the row the segment's product is added to (add_idx[j], when given) lies inside the block;
// else writes tile_ptr[j] and fold record j.  Shared by dn_fold_graph_tiles_build_i32 (dn_close.hip) and the graph-local index
// builder (dn_index_local.hip), which runs it on the candidate relation it finds itself -- no second read-back.
__device__ __forceinline__ void dn_fold_graph_tile_one(int64_t j, int32_t N, int32_t S, const int32_t* __restrict__ sptr,
                                                       const int32_t* __restrict__ snodes, const int32_t* __restrict__ add_idx,
                                                       int32_t* __restrict__ tile_ptr, int32_t* __restrict__ info, int32_t* __restrict__ ok) {
    if (j > S) return;
    if (j == S) { tile_ptr[S] = N; return; }
    const int32_t cnt = sptr[j + 1] - sptr[j];
    bool good = cnt > 0;
    int32_t first = 0, last = 0, nxt = N;
    if (good) {
        first = snodes[sptr[j]];
        last = snodes[sptr[j + 1] - 1];
        good = first >= 0 && last < N && last - first == cnt - 1;
        for (int32_t e = sptr[j]; good && e + 1 < sptr[j + 1]; ++e) good = snodes[e + 1] == snodes[e] + 1;
        if (good && j + 1 < S) {
            good = sptr[j + 2] > sptr[j + 1];
            if (good) { nxt = snodes[sptr[j + 1]]; good = nxt > last; }
        }
    }
    const int32_t b0 = j == 0 ? 0 : first;
    if (good) good = nxt - b0 <= 32 && nxt - b0 >= 1;
    if (good && add_idx != nullptr) good = add_idx[j] >= b0 && add_idx[j] < nxt;
    if (!good) {                                   // device scope + completed before this lane goes on: a later workgroup of the SAME
        __hip_atomic_store(ok, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // launch may read it (ril_fold_verdict_kernel's ticket)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    tile_ptr[j] = b0;
    uint8_t ids[32];
    for (int i = 0; i < 32; ++i) ids[i] = (b0 + i >= first && b0 + i <= last) ? 0 : 255;
    int32_t* rec = info + (size_t)j * 12;
    for (int i = 0; i < 8; ++i)
        rec[i] = (int32_t)((uint32_t)ids[4 * i] | ((uint32_t)ids[4 * i + 1] << 8) | ((uint32_t)ids[4 * i + 2] << 16) | ((uint32_t)ids[4 * i + 3] << 24));
    rec[8] = (int32_t)j; rec[9] = 1; rec[10] = 2; rec[11] = 0;                // [10]: bit 0 = continues the previous tile's sum, bit 1 = completes it
}

// Are the runs snodes[beg .. beg + cnt) consecutive integers?  One run per lane of a CONVERGED wavefront (lanes without one pass
// want = false); the lanes walk each run together (coalesced), instead of every lane walking its own -- a 600-node graph next to
// 63 small ones would keep its wavefront 600 dependent loads long.
__device__ __forceinline__ bool dn_wave_runs_contiguous(bool want, int32_t beg, int32_t cnt, const int32_t* __restrict__ snodes) {
    const int lane = (int)(threadIdx.x & 63);
    bool good = true;
    unsigned long long todo = __ballot(want && cnt > 1);
    while (todo) {
        const int sl = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int32_t b = __shfl(beg, sl, 64), c = __shfl(cnt, sl, 64);
        bool ok = true;
        for (int32_t e = lane; e + 1 < c; e += 64) ok = ok && snodes[b + e + 1] == snodes[b + e] + 1;
        const bool all = __all(ok) != 0;
        if (lane == sl) good = all;
    }
    return good;
}

// BOTH verdicts of a batch's segments in one pass (the graph-local index builder's verdict launch): *ok starts as 3; an invalid
// segment clears it, a valid one whose block has more than 32 nodes clears bit 0 (no graphs-as-tiles tables: dn_fold_graph_tile_one)
// and leaves bit 1 (chunked tiles: dn_fold_multi_*); a block within 32 nodes gets its single-tile record.  Called by EVERY lane of a
// converged wavefront (active: the lane has an index j <= S).
__device__ __forceinline__ void dn_fold_verdicts_one(bool active, int64_t j, int32_t N, int32_t S, const int32_t* __restrict__ sptr,
                                                     const int32_t* __restrict__ snodes, const int32_t* __restrict__ add_idx,
                                                     int32_t* __restrict__ tile_ptr, int32_t* __restrict__ info, int32_t* __restrict__ ok) {
    const bool seg = active && j < S;
    int32_t beg = 0, cnt = 0, first = 0, last = 0, nxt = N;
    if (seg) { beg = sptr[j]; cnt = sptr[j + 1] - beg; }
    bool good = seg && cnt > 0;
    if (good) {
        first = snodes[beg];
        last = snodes[beg + cnt - 1];
        good = first >= 0 && last < N && last - first == cnt - 1;
    }
    const bool runs = dn_wave_runs_contiguous(good, beg, cnt, snodes);
    good = good && runs;
    if (good && j + 1 < S) {
        good = sptr[j + 2] > sptr[j + 1];
        if (good) { nxt = snodes[sptr[j + 1]]; good = nxt > last; }
    }
    const int32_t b0 = j == 0 ? 0 : first;
    if (good) good = nxt - b0 >= 1;
    if (good && add_idx != nullptr) good = add_idx[j] >= b0 && add_idx[j] < nxt;
    if (seg) {
        if (!good) {
            __hip_atomic_store(ok, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (nxt - b0 > 32) {
            __hip_atomic_fetch_and(ok, ~1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            tile_ptr[j] = b0;
            int32_t* rec = info + (size_t)j * 12;
            for (int i = 0; i < 8; ++i) {
                uint32_t w = 0;
                for (int b = 0; b < 4; ++b) w |= ((b0 + 4 * i + b >= first && b0 + 4 * i + b <= last) ? 0u : 255u) << (8 * b);
                rec[i] = (int32_t)w;
            }
            rec[8] = (int32_t)j; rec[9] = 1; rec[10] = 2; rec[11] = 0;
        }
    }
    if (active && j == S) tile_ptr[S] = N;
}

// The same for graphs of ANY size (round 6).  The batch is cut into C CHUNKS at graph boundaries -- graph j (block start b0_j)
// belongs to chunk floor(b0_j C / N) -- and every chunk into consecutive 32-node tiles that run ACROSS the graphs inside it (the
// last tile of a chunk is the only partial one: T <= N / 32 + C).  The unit stream keeps a chunk in ONE workgroup, so a segment's
// column sum may continue from tile to tile in that workgroup (fold record word 10: bit 0 = the tile's FIRST segment continues the
// previous tile's sum, bit 1 = its LAST segment is complete inside this tile), and the AGG unit's read-modify-write of the dummy
// node's row stays inside the workgroup that stored it.
//   valid:  the test of dn_fold_graph_tile_one without the 32-node limit, one segment per call
//   chunks: chunk_graph [C + 1] (first graph of chunk c), chunk_tile [C + 1] (first tile; chunk_tile[C] = T) -- one workgroup
//   tiles:  tile_ptr [T + 1], fold record per tile {local segment ids in order of appearance, 255 outside every segment; aux row of
//           the first segment present = its graph; segments present; flags}
__device__ __forceinline__ int32_t dn_fold_gstart(int64_t j, int32_t N, int32_t S, const int32_t* __restrict__ sptr,
                                                  const int32_t* __restrict__ snodes) {
    return j <= 0 ? 0 : (j >= S ? N : snodes[sptr[j]]);                   // block j = [gstart(j), gstart(j + 1))
}
// (called by every lane of a converged wavefront: the contiguity walk is shared, dn_wave_runs_contiguous)
__device__ __forceinline__ void dn_fold_multi_valid_one(int64_t j, int32_t N, int32_t S, const int32_t* __restrict__ sptr,
                                                        const int32_t* __restrict__ snodes, const int32_t* __restrict__ add_idx,
                                                        int32_t* __restrict__ ok) {
    const bool seg = j < S;
    int32_t beg = 0, cnt = 0, first = 0, last = 0, nxt = N;
    if (seg) { beg = sptr[j]; cnt = sptr[j + 1] - beg; }
    bool good = seg && cnt > 0;
    if (good) {
        first = snodes[beg];
        last = snodes[beg + cnt - 1];
        good = first >= 0 && last < N && last - first == cnt - 1;
    }
    const bool runs = dn_wave_runs_contiguous(good, beg, cnt, snodes);
    good = good && runs;
    if (good && j + 1 < S) {
        good = sptr[j + 2] > sptr[j + 1];
        if (good) { nxt = snodes[sptr[j + 1]]; good = nxt > last; }
    }
    const int32_t b0 = j == 0 ? 0 : first;
    if (good) good = nxt - b0 >= 1;
    if (good && add_idx != nullptr) good = add_idx[j] >= b0 && add_idx[j] < nxt;
    if (seg && !good) {
        __hip_atomic_store(ok, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}
// first graph j in [0, S] whose block starts at or behind node v (S: none)
__device__ __forceinline__ int32_t dn_fold_first_graph_from(int32_t v, int32_t N, int32_t S, const int32_t* __restrict__ sptr,
                                                            const int32_t* __restrict__ snodes) {
    int32_t lo = 0, hi = S;
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (dn_fold_gstart(mid, N, S, sptr, snodes) < v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ void dn_fold_multi_tile_one(int64_t t, int32_t N, int32_t S, int32_t C, const int32_t* __restrict__ sptr,
                                                       const int32_t* __restrict__ snodes, const int32_t* __restrict__ chunk_tile,
                                                       const int32_t* __restrict__ chunk_graph, int32_t* __restrict__ tile_ptr,
                                                       int32_t* __restrict__ info) {
    const int32_t T = chunk_tile[C];
    if (t > T) return;
    if (t == T) { tile_ptr[T] = N; return; }
    int32_t lo = 0, hi = C;                                                // the chunk of tile t: the last c with chunk_tile[c] <= t
    while (lo < hi) {
        const int32_t mid = (lo + hi + 1) >> 1;
        if (chunk_tile[mid] <= (int32_t)t) lo = mid;
        else hi = mid - 1;
    }
    const int32_t c = lo, g_lo = chunk_graph[c], g_hi = chunk_graph[c + 1];
    const int32_t n_lo = dn_fold_gstart(g_lo, N, S, sptr, snodes), n_hi = dn_fold_gstart(g_hi, N, S, sptr, snodes);
    const int32_t p0 = n_lo + 32 * ((int32_t)t - chunk_tile[c]), pend = p0 + 32 < n_hi ? p0 + 32 : n_hi;
    tile_ptr[t] = p0;
    // the graph of node p0: the last j in [g_lo, g_hi) with gstart(j) <= p0
    int32_t a = g_lo, b = g_hi - 1;
    while (a < b) {
        const int32_t mid = (a + b + 1) >> 1;
        if (dn_fold_gstart(mid, N, S, sptr, snodes) <= p0) a = mid;
        else b = mid - 1;
    }
    int32_t jg = a, nxt = dn_fold_gstart(jg + 1, N, S, sptr, snodes), sfirst = snodes[sptr[jg]], slast = snodes[sptr[jg + 1] - 1];
    int32_t first_seg = -1, last_seg = -1;
    uint32_t w[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    for (int i = 0; i < 32; ++i) {
        const int32_t v = p0 + i;
        uint32_t id = 255u;
        if (v < pend) {
            while (v >= nxt) {
                ++jg;
                nxt = dn_fold_gstart(jg + 1, N, S, sptr, snodes);
                sfirst = snodes[sptr[jg]]; slast = snodes[sptr[jg + 1] - 1];
            }
            if (v >= sfirst && v <= slast) {
                if (first_seg < 0) first_seg = jg;
                last_seg = jg;
                id = (uint32_t)(jg - first_seg);                           // (< 32: every segment present holds a node of the tile)
            }
        }
        w[i >> 2] |= id << (8 * (i & 3));
    }
    int32_t* rec = info + (size_t)t * 12;
    for (int i = 0; i < 8; ++i) rec[i] = (int32_t)w[i];
    int32_t flags = 0;
    if (first_seg >= 0) {
        if (snodes[sptr[first_seg]] < p0) flags |= 1;                      // its first nodes lie in the previous tile
        if (snodes[sptr[last_seg + 1] - 1] < pend) flags |= 2;             // its last node lies in this tile
    }
    rec[8] = first_seg < 0 ? 0 : first_seg;
    rec[9] = first_seg < 0 ? 0 : last_seg - first_seg + 1;
    rec[10] = flags; rec[11] = 0;
}

// ---- ReLU / leaky ReLU.  slope = 0: ReLU (max(v, 0): the negative side is an exact 0 whatever v is); slope > 0: leaky ReLU, the
// reference's default activation (`leaky_relu`, slope 1 / 5.5: subgraph_isomorphism/utils/act.py:466, constants.py:10).  The backward
// multiplies a gradient by 1 where the saved activation (or its sign bit) is > 0 and by `slope` elsewhere.
__device__ __forceinline__ float dn_neg(float v, float slope) { return slope != 0.f ? v * slope : 0.f; }
__device__ __forceinline__ float dn_act(float v, float slope) { return v > 0.f ? v : dn_neg(v, slope); }
__device__ __forceinline__ uint32_t dn_bf16_bits(float v) {
    return (uint32_t)__builtin_bit_cast(unsigned short, (__bf16)v);
}
// one packed pair of bf16 values: element kept where its flag is set, scaled by slope (0: zeroed) otherwise
__device__ __forceinline__ uint32_t dn_pair_keep_or_scale(uint32_t w, bool keep_lo, bool keep_hi, float slope) {
    if (slope == 0.f) return w & ((keep_lo ? 0x0000ffffu : 0u) | (keep_hi ? 0xffff0000u : 0u));
    const uint32_t lo = keep_lo ? (w & 0xffffu) : dn_bf16_bits(__uint_as_float(w << 16) * slope);
    const uint32_t hi = keep_hi ? (w >> 16) : dn_bf16_bits(__uint_as_float(w & 0xffff0000u) * slope);
    return lo | (hi << 16);
}
__device__ __forceinline__ bool dn_bf16_pos(uint32_t h) { return h != 0u && h < 0x8000u; }       // bf16 bits > 0
// 8 bf16 values (16 bytes), bit i of `bits` <-> element i
__device__ __forceinline__ uint4 dn_keep_or_scale_bits(const uint4& v, uint32_t bits, float slope) {
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = dn_pair_keep_or_scale(w[i], (bits >> (2 * i)) & 1u, (bits >> (2 * i + 1)) & 1u, slope);
    return make_uint4(w[0], w[1], w[2], w[3]);
}
// ... kept where the bf16 mask element (a saved activation) is > 0
__device__ __forceinline__ uint4 dn_keep_or_scale_mask(const uint4& v, const uint4& mk, float slope) {
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
    const uint32_t m[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = dn_pair_keep_or_scale(w[i], dn_bf16_pos(m[i] & 0xffffu), dn_bf16_pos(m[i] >> 16), slope);
    return make_uint4(w[0], w[1], w[2], w[3]);
}
