// fp32 twins of the relation-wise matrix-core kernels (dn_rel.hip) on the exact-f32 MFMA of gfx950
// (v_mfma_f32_16x16x4_f32: f32 in / f32 accumulate, bit-for-bit an fmaf chain, 1/16 of the bf16 rate).
// The reference is fp32 only, so this is the path that keeps its numerics (1e-4 parity) while still replacing the
// per-edge [E,H,H] weight gather (subgraph_isomorphism/models/rgin.py:109-110) by the row factorisation.
//
// Two arithmetic modes behind the same entry points (`precision` argument):
//   0 (default)  3-term bf16 split on the fast matrix path: every f32 operand is cut into hi = bf16(x) and lo = bf16(x - hi)
//                when it is staged (rows: global -> LDS; weights: global -> registers), and a product is evaluated as
//                hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16 with f32 accumulation.  The dropped lo*lo term and the
//                two roundings are O(2^-16) relative per product: 1e-5-level agreement with the reference, inside its 1e-4
//                bar, at 3/16 of the exact path's matrix time (MI355X: exact f32 MFMA = 1/16 of the bf16 rate).
//   1            exact f32 (v_mfma_f32_16x16x4_f32): the checker, and what the parity suite pins the split against.
//
// Same structure as the bf16 kernels; what changes in exact mode is the fragment shape: an MFMA step reduces over 4 values, lane
// (r = lane & 15, g = lane >> 4) supplies one f32 of row r.  Four consecutive steps are fed from ONE 16-byte load per
// lane by letting step j use k = 16*blk + 4*g + j on BOTH operands (the sum over k does not care about the order), so
// LDS and weight reads stay 128-bit.
#include "dn_common.h"
#include "../../include/dn_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16_t;
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef short short8v __attribute__((ext_vector_type(8)));

// x = hi + lo (+ O(2^-17 |x|)): hi = bf16(x) (round to nearest even), lo = bf16(x - hi)
__device__ __forceinline__ void split4(const float4& v, bf16x4& hi, bf16x4& lo) {
    const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        hi[i] = (bf16_t)f[i];
        lo[i] = (bf16_t)(f[i] - (float)hi[i]);
    }
}
__device__ __forceinline__ void split8(const float4& a, const float4& b, bf16x8& hi, bf16x8& lo) {
    const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        hi[i] = (bf16_t)f[i];
        lo[i] = (bf16_t)(f[i] - (float)hi[i]);
    }
}
// fragment of a K-strided bf16 operand held row-major in LDS (hardware transpose read, as dn_rel.hip:tr_frag)
__device__ __forceinline__ bf16x8 tr_frag16(const bf16_t* tile, int stride, int col0, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const bf16_t* a0 = tile + (8 * g + q) * stride + col0 + 4 * p;
    typedef short4v __attribute__((address_space(3))) * lds_p;
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a0));
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a0 + 4 * stride));
    const short8v f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, f);
}

constexpr int kThreads = 512;
constexpr int kRows = 32;

// Where a tile's weight matrix and bias row live (dn_rows_transform_f32): relation `loop_rel` reads W_loop instead of Wn[rel]
// (the layer's self-loop parameter, rgin.py:66-67, no concatenated copy); kn = the matrices are [k][n] (parameter layout);
// bias_rel >= 0: `bias` is ONE row [H] that only tiles of that relation add (rgin.py:146), < 0: one row per relation.
struct WeightForm {
    const float* W_loop;
    int32_t loop_rel, bias_rel, kn;
    __device__ __forceinline__ const float* matrix(const float* Wn, int rel, int H) const {
        return rel == loop_rel ? W_loop : Wn + (size_t)rel * H * H;
    }
    __device__ __forceinline__ const float* bias_row(const float* bias, int rel, int H) const {
        if (bias == nullptr) return nullptr;
        if (bias_rel < 0) return bias + (size_t)rel * H;
        return rel == bias_rel ? bias : nullptr;
    }
};

struct Chunk {
    int32_t rel, beg, end, pad;
};

// ---------------------------------------------------------------------------------------------------------------
// weight gradient:  partial[chunk][k][n] = sum_{p in chunk} A[ia[p]][k] * G[ig[p]][n]
// ---------------------------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(kThreads) void rows_wgrad_f32_kernel(const float* __restrict__ A, const float* __restrict__ A2,
                                                                  int32_t na1, const int32_t* __restrict__ ia,
                                                                  const float* __restrict__ G, const float* __restrict__ G2,
                                                                  int32_t ng1, const int32_t* __restrict__ ig,
                                                                  const Chunk* __restrict__ chunks, float* __restrict__ partial,
                                                                  int32_t colsum_of, float* __restrict__ colsum_partial,
                                                                  const float* __restrict__ maskA, float* __restrict__ A_out, float slope) {
    constexpr int S = H + 16;                                   // LDS row stride in floats: rows p, p+1 land 16 banks apart
    constexpr int MT = H / 2 / 16, NT = H / 4 / 16;
    constexpr int NP = kRows * H / 4;                           // 16-byte pieces per operand tile
    constexpr int P = (NP + kThreads - 1) / kThreads;
    static_assert(MT >= 1 && NT >= 1, "unsupported width");
    __shared__ __attribute__((aligned(16))) float lds[4 * kRows * S];
    auto bufA = [&](int b) -> float* { return lds + b * (kRows * S); };
    auto bufG = [&](int b) -> float* { return lds + 2 * kRows * S + b * (kRows * S); };

    const Chunk ch = chunks[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k0 = (wave >> 2) * (H / 2), n0 = (wave & 3) * (H / 4);
    const int ntiles = (ch.end - ch.beg + kRows - 1) / kRows;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    float4 ra[P], rg[P], rm[P];
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    int32_t sa[P], sg[P], sa_cur[P];
    auto load_idx = [&](int t) {
        const int row0 = ch.beg + t * kRows;
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int piece = tid + j * kThreads, p = row0 + piece / (H / 4);
            const bool ok = piece < NP && p < ch.end;
            sa[j] = ok ? (ia ? ia[p] : p) : -1;
            sg[j] = ok ? (ig ? ig[p] : p) : -1;
        }
    };
    auto load_tile = [&]() {
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int c = (tid + j * kThreads) % (H / 4);
            sa_cur[j] = sa[j];
            ra[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            rg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (sa[j] >= 0) {
                const float* base = sa[j] < na1 ? A + (size_t)sa[j] * H : A2 + (size_t)(sa[j] - na1) * H;
                ra[j] = *reinterpret_cast<const float4*>(base + c * 4);
                if (maskA) rm[j] = *reinterpret_cast<const float4*>(maskA + (size_t)sa[j] * H + c * 4);
            }
            if (sg[j] >= 0) {
                const float* base = sg[j] < ng1 ? G + (size_t)sg[j] * H : G2 + (size_t)(sg[j] - ng1) * H;
                rg[j] = *reinterpret_cast<const float4*>(base + c * 4);
            }
        }
    };
    auto store_tile = [&](int b) {
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int piece = tid + j * kThreads, r = piece / (H / 4), c = piece % (H / 4);
            if (maskA && sa_cur[j] >= 0) {
                ra[j].x = rm[j].x > 0.f ? ra[j].x : dn_neg(ra[j].x, slope); ra[j].y = rm[j].y > 0.f ? ra[j].y : dn_neg(ra[j].y, slope);
                ra[j].z = rm[j].z > 0.f ? ra[j].z : dn_neg(ra[j].z, slope); ra[j].w = rm[j].w > 0.f ? ra[j].w : dn_neg(ra[j].w, slope);
                if (A_out) *reinterpret_cast<float4*>(A_out + (size_t)sa_cur[j] * H + c * 4) = ra[j];
            }
            if (piece < NP) {
                *reinterpret_cast<float4*>(bufA(b) + r * S + c * 4) = ra[j];
                *reinterpret_cast<float4*>(bufG(b) + r * S + c * 4) = rg[j];
                const float4 v = colsum_of == 1 ? ra[j] : rg[j];
                if (colsum_of != 0) { cs[0] += v.x; cs[1] += v.y; cs[2] += v.z; cs[3] += v.w; }
            }
        }
    };

    if (ntiles > 0) {
        load_idx(0);
        load_tile();
        store_tile(0);
        load_idx(1);
    }
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int b = t & 1;
        if (t + 1 < ntiles) {
            load_tile();
            load_idx(t + 2);
        }
        const float* ta = bufA(b) + (lane >> 4) * S + (lane & 15);
        const float* tg = bufG(b) + (lane >> 4) * S + (lane & 15);
#pragma unroll
        for (int pb = 0; pb < kRows / 4; ++pb) {                 // 4 rows (reduction index) per MFMA step
            float fb[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) fb[n] = tg[pb * 4 * S + n0 + n * 16];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float fa = ta[pb * 4 * S + k0 + m * 16];
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb[n], acc[m][n], 0, 0, 0);
            }
        }
        if (t + 1 < ntiles) store_tile(b ^ 1);
        __syncthreads();
    }
    float* out = partial + (size_t)blockIdx.x * H * H;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = k0 + m * 16 + (lane >> 4) * 4 + i, c = n0 + n * 16 + (lane & 15);
                out[(size_t)k * H + c] = acc[m][n][i];
            }
    if (colsum_of != 0) {
        constexpr int TPC = kThreads / (H / 4);                  // threads sharing a 4-column chunk
        float* red = lds;
        const int cchunk = tid % (H / 4), slot = tid / (H / 4);
        const bool has = (NP >= kThreads) || tid < NP;
#pragma unroll
        for (int i = 0; i < 4; ++i) red[slot * H + cchunk * 4 + i] = has ? cs[i] : 0.f;
        __syncthreads();
        if (tid < H) {
            float sum = 0.f;
            for (int sl = 0; sl < TPC; ++sl) sum += red[sl * H + tid];
            colsum_partial[(size_t)blockIdx.x * H + tid] = sum;
        }
    }
}

// 3-term split twin of the weight gradient: the four bf16 tiles Ah, Al, Gh, Gl of a 32-row stage sit row-major in LDS
// ([32][H + 8], as the bf16 kernel's), fragments come through the transposing LDS read, three MFMAs per (m, n) pair.
template <int H>
__device__ __forceinline__ void rows_wgrad_f32s_body(const float* __restrict__ A, const float* __restrict__ A2,
                                                     int32_t na1, const int32_t* __restrict__ ia,
                                                     const float* __restrict__ G, const float* __restrict__ G2,
                                                     int32_t ng1, const int32_t* __restrict__ ig,
                                                     const Chunk ch, float* __restrict__ partial,
                                                     int32_t colsum_of, float* __restrict__ colsum_partial,
                                                     const float* __restrict__ maskA, float* __restrict__ A_out, float slope,
                                                     bool zero_colsum) {
    constexpr int S = H + 8;                                    // bf16 elements per LDS row
    constexpr int MT = H / 2 / 16, NT = H / 4 / 16;
    constexpr int NP = kRows * H / 4;                           // 4-float pieces per operand tile
    constexpr int P = (NP + kThreads - 1) / kThreads;
    __shared__ __attribute__((aligned(16))) bf16_t lds[2 * 4 * kRows * S];
    auto tile = [&](int b, int which) -> bf16_t* { return lds + (b * 4 + which) * (kRows * S); };   // 0 Ah, 1 Al, 2 Gh, 3 Gl

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k0 = (wave >> 2) * (H / 2), n0 = (wave & 3) * (H / 4);
    const int ntiles = (ch.end - ch.beg + kRows - 1) / kRows;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 ra[P], rg[P], rm[P];
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    int32_t sa[P], sg[P], sa_cur[P];
    auto load_idx = [&](int t) {
        const int row0 = ch.beg + t * kRows;
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int piece = tid + j * kThreads, p = row0 + piece / (H / 4);
            const bool ok = piece < NP && p < ch.end;
            sa[j] = ok ? (ia ? ia[p] : p) : -1;
            sg[j] = ok ? (ig ? ig[p] : p) : -1;
        }
    };
    auto load_tile = [&]() {
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int c = (tid + j * kThreads) % (H / 4);
            sa_cur[j] = sa[j];
            ra[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            rg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (sa[j] >= 0) {
                const float* base = sa[j] < na1 ? A + (size_t)sa[j] * H : A2 + (size_t)(sa[j] - na1) * H;
                ra[j] = *reinterpret_cast<const float4*>(base + c * 4);
                if (maskA) rm[j] = *reinterpret_cast<const float4*>(maskA + (size_t)sa[j] * H + c * 4);
            }
            if (sg[j] >= 0) {
                const float* base = sg[j] < ng1 ? G + (size_t)sg[j] * H : G2 + (size_t)(sg[j] - ng1) * H;
                rg[j] = *reinterpret_cast<const float4*>(base + c * 4);
            }
        }
    };
    auto store_tile = [&](int b) {
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int piece = tid + j * kThreads, r = piece / (H / 4), c = piece % (H / 4);
            if (maskA && sa_cur[j] >= 0) {
                ra[j].x = rm[j].x > 0.f ? ra[j].x : dn_neg(ra[j].x, slope); ra[j].y = rm[j].y > 0.f ? ra[j].y : dn_neg(ra[j].y, slope);
                ra[j].z = rm[j].z > 0.f ? ra[j].z : dn_neg(ra[j].z, slope); ra[j].w = rm[j].w > 0.f ? ra[j].w : dn_neg(ra[j].w, slope);
                if (A_out) *reinterpret_cast<float4*>(A_out + (size_t)sa_cur[j] * H + c * 4) = ra[j];
            }
            if (piece < NP) {
                bf16x4 h, l;
                split4(ra[j], h, l);
                *reinterpret_cast<bf16x4*>(tile(b, 0) + r * S + c * 4) = h;
                *reinterpret_cast<bf16x4*>(tile(b, 1) + r * S + c * 4) = l;
                split4(rg[j], h, l);
                *reinterpret_cast<bf16x4*>(tile(b, 2) + r * S + c * 4) = h;
                *reinterpret_cast<bf16x4*>(tile(b, 3) + r * S + c * 4) = l;
                const float4 v = colsum_of == 1 ? ra[j] : rg[j];
                if (colsum_of != 0) { cs[0] += v.x; cs[1] += v.y; cs[2] += v.z; cs[3] += v.w; }
            }
        }
    };
    if (ntiles > 0) {
        load_idx(0);
        load_tile();
        store_tile(0);
        load_idx(1);
    }
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int b = t & 1;
        if (t + 1 < ntiles) {
            load_tile();
            load_idx(t + 2);
        }
        bf16x8 gh[NT], gl[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            gh[n] = tr_frag16(tile(b, 2), S, n0 + n * 16, lane);
            gl[n] = tr_frag16(tile(b, 3), S, n0 + n * 16, lane);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const bf16x8 ah = tr_frag16(tile(b, 0), S, k0 + m * 16, lane);
            const bf16x8 al = tr_frag16(tile(b, 1), S, k0 + m * 16, lane);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, gh[n], acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, gl[n], acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, gh[n], acc[m][n], 0, 0, 0);
            }
        }
        if (t + 1 < ntiles) store_tile(b ^ 1);
        __syncthreads();
    }
    float* out = partial + (size_t)blockIdx.x * H * H;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = k0 + m * 16 + (lane >> 4) * 4 + i, c = n0 + n * 16 + (lane & 15);
                out[(size_t)k * H + c] = acc[m][n][i];
            }
    if (colsum_of != 0 || zero_colsum) {                         // (a job without column sums in a multi-job launch: zeros)
        constexpr int TPC = kThreads / (H / 4);
        float* red = reinterpret_cast<float*>(lds);
        const int cchunk = tid % (H / 4), slot = tid / (H / 4);
        const bool has = (NP >= kThreads) || tid < NP;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) red[slot * H + cchunk * 4 + i] = has ? cs[i] : 0.f;
        __syncthreads();
        if (tid < H) {
            float sum = 0.f;
            for (int sl = 0; sl < TPC; ++sl) sum += red[sl * H + tid];
            colsum_partial[(size_t)blockIdx.x * H + tid] = sum;
        }
    }
}

template <int H>
__global__ __launch_bounds__(kThreads) void rows_wgrad_f32s_kernel(const float* __restrict__ A, const float* __restrict__ A2,
                                                                   int32_t na1, const int32_t* __restrict__ ia,
                                                                   const float* __restrict__ G, const float* __restrict__ G2,
                                                                   int32_t ng1, const int32_t* __restrict__ ig,
                                                                   const Chunk* __restrict__ chunks, float* __restrict__ partial,
                                                                   int32_t colsum_of, float* __restrict__ colsum_partial,
                                                                   const float* __restrict__ maskA, float* __restrict__ A_out, float slope) {
    rows_wgrad_f32s_body<H>(A, A2, na1, ia, G, G2, ng1, ig, chunks[blockIdx.x], partial, colsum_of, colsum_partial, maskA, A_out, slope, false);
}

// Several weight gradients in one launch (dn_rows_wgrad_multi_f32; as rows_wgrad_multi_kernel of dn_rel.hip): the jobs' relations are
// numbered through, their rows lie end to end in one virtual row space that the chunk table covers; a chunk finds its job by its
// relation (selects between the argument sets, not an indexed read: the struct lives in kernel-argument SGPRs).
struct WgJobF {
    const float *A, *A2;
    const int32_t* ia;
    const float *G, *G2;
    const int32_t* ig;
    const float* maskA;
    int32_t na1, ng1, colsum_of, first_rel, row0;
    float slope;
};
struct WgJobsF {
    WgJobF j[3];
    int32_t n;
};
template <int H>
__global__ __launch_bounds__(kThreads) void rows_wgrad_f32s_multi_kernel(WgJobsF jobs, const Chunk* __restrict__ chunks,
                                                                         float* __restrict__ partial, float* __restrict__ colsum_partial) {
    Chunk ch = chunks[blockIdx.x];
    int k = 0;
    if (jobs.n > 1 && ch.rel >= jobs.j[1].first_rel) k = 1;
    if (jobs.n > 2 && ch.rel >= jobs.j[2].first_rel) k = 2;
    const WgJobF J = k == 0 ? jobs.j[0] : (k == 1 ? jobs.j[1] : jobs.j[2]);
    ch.beg -= J.row0; ch.end -= J.row0;
    rows_wgrad_f32s_body<H>(J.A, J.A2, J.na1, J.ia, J.G, J.G2, J.ng1, J.ig, ch, partial, J.colsum_of, colsum_partial, J.maskA, nullptr, J.slope, true);
}

// out[r] = sum of chunk partials (same slice-parallel fixed-order fold as the bf16 library's reducer)
__global__ __launch_bounds__(256) void wgrad_reduce_f32_kernel(const float* __restrict__ partial,
                                                               const int32_t* __restrict__ chunk_ptr, int64_t tile_elems,
                                                               float* __restrict__ out, const float* __restrict__ cs_partial,
                                                               int32_t H, float* __restrict__ out_colsum) {
    constexpr int SL = 8, EL = 256 / SL;
    __shared__ float red[SL][EL];
    const int r = blockIdx.y;
    const int elem = threadIdx.x % EL, slice = threadIdx.x / EL;
    const int cb = chunk_ptr[r], ce = chunk_ptr[r + 1];
    const int64_t tile_blocks = (tile_elems + EL - 1) / EL;
    const bool is_cs = (int64_t)blockIdx.x >= tile_blocks;
    const float* src = is_cs ? cs_partial : partial;
    const int64_t stride = is_cs ? (int64_t)H : tile_elems;
    const int64_t i = (is_cs ? (int64_t)blockIdx.x - tile_blocks : (int64_t)blockIdx.x) * EL + elem;
    float s = 0.f;
    if (i < stride) {
        int c = cb + slice;
        for (; c + 3 * SL < ce; c += 4 * SL) {
            const float v0 = src[(size_t)c * stride + i], v1 = src[(size_t)(c + SL) * stride + i];
            const float v2 = src[(size_t)(c + 2 * SL) * stride + i], v3 = src[(size_t)(c + 3 * SL) * stride + i];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; c < ce; c += SL) s += src[(size_t)c * stride + i];
    }
    red[slice][elem] = s;
    __syncthreads();
    if (slice == 0 && i < stride) {
        float t = red[0][elem];
#pragma unroll
        for (int k = 1; k < SL; ++k) t += red[k][elem];
        if (is_cs) out_colsum[(size_t)r * H + i] = t;
        else out[(size_t)r * tile_elems + i] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// transform:  Y[p, n] = epi( sum_k Xcat[idx[p], k] * Wn[rel(p)][n][k] )
// ---------------------------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(kThreads, 2) void rows_transform_f32_kernel(
    const float* __restrict__ X, const float* __restrict__ X2, int32_t n1, const int32_t* __restrict__ idx,
    const float* __restrict__ Wn, const float* __restrict__ bias, int32_t relu, float slope, const float* __restrict__ mask_pos,
    const Chunk* __restrict__ tiles, int32_t num_tiles, int32_t tiles_per_wg, float* __restrict__ Y, WeightForm wform) {
    constexpr int S = H + 4;                                    // 16-byte row pad: conflict-free ds_read_b128 fragments
    constexpr int KB = H / 16;                                  // 16-wide k blocks (4 MFMA steps each)
    constexpr int NT = (H / 8 + 15) / 16;
    constexpr int MT = kRows / 16;
    constexpr int NP = kRows * H / 4;
    constexpr int P = (NP + kThreads - 1) / kThreads;
    __shared__ __attribute__((aligned(16))) float lds[3 * kRows * S];
    auto bufX = [&](int b) -> float* { return lds + b * (kRows * S); };
    float* bufY = lds + 2 * kRows * S;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = wave * (NT * 16);
    const bool wave_active = n0 < H;
    const int t_beg = blockIdx.x * tiles_per_wg;
    const int t_end = min(t_beg + tiles_per_wg, num_tiles);
    if (t_beg >= t_end) return;

    float4 wf[KB][NT];
    int cur_rel = -1;
    int32_t nidx[P];
    float4 rx[P];
    auto load_idx = [&](int t) {
        if (t >= t_end) return;
        const Chunk tl = tiles[t];
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int piece = tid + j * kThreads, p = tl.beg + piece / (H / 4);
            nidx[j] = (piece < NP && p < tl.end) ? (idx ? idx[p] : p) : -1;
        }
    };
    auto load_rows = [&]() {
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int c = (tid + j * kThreads) % (H / 4);
            rx[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (nidx[j] >= 0) {
                const float* base = nidx[j] < n1 ? X + (size_t)nidx[j] * H : X2 + (size_t)(nidx[j] - n1) * H;
                rx[j] = *reinterpret_cast<const float4*>(base + c * 4);
            }
        }
    };
    auto store_rows = [&](int b) {
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int piece = tid + j * kThreads, r = piece / (H / 4), c = piece % (H / 4);
            if (piece < NP) *reinterpret_cast<float4*>(bufX(b) + r * S + c * 4) = rx[j];
        }
    };

    load_idx(t_beg);
    load_rows();
    store_rows(0);
    load_idx(t_beg + 1);
    __syncthreads();

    for (int t = t_beg; t < t_end; ++t) {
        const int b = (t - t_beg) & 1;
        const Chunk tl = tiles[t];
        if (t + 1 < t_end) load_rows();
        if (tl.rel != cur_rel && wave_active) {
            cur_rel = tl.rel;
            const float* w = wform.matrix(Wn, cur_rel, H);
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int n = n0 + nt * 16 + (lane & 15), k = kb * 16 + 4 * (lane >> 4);
                    if (wform.kn)                               // [k][n] as the parameter is stored: 16 lanes share a 64-byte run
                        wf[kb][nt] = make_float4(w[(size_t)k * H + n], w[(size_t)(k + 1) * H + n], w[(size_t)(k + 2) * H + n],
                                                 w[(size_t)(k + 3) * H + n]);
                    else
                        wf[kb][nt] = *reinterpret_cast<const float4*>(w + (size_t)n * H + k);
                }
        }
        if (wave_active) {
            f32x4 acc[MT][NT];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* xt = bufX(b) + (lane & 15) * S + 4 * (lane >> 4);
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                float4 xf[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) xf[m] = *reinterpret_cast<const float4*>(xt + m * 16 * S + kb * 16);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[kb][n].x, xf[m].x, acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[kb][n].y, xf[m].y, acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[kb][n].z, xf[m].z, acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[kb][n].w, xf[m].w, acc[m][n], 0, 0, 0);
                    }
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int col = n0 + n * 16 + 4 * (lane >> 4);
                    float4 v = make_float4(acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]);
                    if (const float* brow = wform.bias_row(bias, cur_rel, H)) {
                        const float4 bv = *reinterpret_cast<const float4*>(brow + col);
                        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                    }
                    if (relu) { v.x = dn_act(v.x, slope); v.y = dn_act(v.y, slope); v.z = dn_act(v.z, slope); v.w = dn_act(v.w, slope); }
                    *reinterpret_cast<float4*>(bufY + (m * 16 + (lane & 15)) * S + col) = v;
                }
        }
        if (t + 1 < t_end) store_rows(b ^ 1);
        load_idx(t + 2);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int piece = tid + j * kThreads, r = piece / (H / 4), c = piece % (H / 4);
            const int p = tl.beg + r;
            if (piece < NP && p < tl.end) {
                float4 v = *reinterpret_cast<const float4*>(bufY + r * S + c * 4);
                if (mask_pos) {
                    const float4 mk = *reinterpret_cast<const float4*>(mask_pos + (size_t)p * H + c * 4);
                    v.x = mk.x > 0.f ? v.x : dn_neg(v.x, slope); v.y = mk.y > 0.f ? v.y : dn_neg(v.y, slope);
                    v.z = mk.z > 0.f ? v.z : dn_neg(v.z, slope); v.w = mk.w > 0.f ? v.w : dn_neg(v.w, slope);
                }
                *reinterpret_cast<float4*>(Y + (size_t)p * H + c * 4) = v;
            }
        }
        __syncthreads();
    }
}

// 3-term split twin of the transform: rows are cut into hi / lo bf16 tiles when they are staged into LDS (same LDS bytes as
// the f32 tile), the wave's weight slice is cut into hi / lo register fragments when the relation changes.
template <int H>
__global__ __launch_bounds__(kThreads, 2) void rows_transform_f32s_kernel(
    const float* __restrict__ X, const float* __restrict__ X2, int32_t n1, const int32_t* __restrict__ idx,
    const float* __restrict__ Wn, const float* __restrict__ bias, int32_t relu, float slope, const float* __restrict__ mask_pos,
    const Chunk* __restrict__ tiles, int32_t num_tiles, int32_t tiles_per_wg, float* __restrict__ Y, WeightForm wform) {
    constexpr int SX = H + 8;                                   // bf16 elements per LDS row of an input tile
    constexpr int SY = H + 4;                                   // floats per LDS row of the output tile
    constexpr int KS = H / 32;
    constexpr int NT = (H / 8 + 15) / 16;
    constexpr int MT = kRows / 16;
    constexpr int NP = kRows * H / 4;
    constexpr int P = (NP + kThreads - 1) / kThreads;
    __shared__ __attribute__((aligned(16))) bf16_t ldx[2 * 2 * kRows * SX];
    __shared__ __attribute__((aligned(16))) float ldy[kRows * SY];
    auto bufX = [&](int b, int lo) -> bf16_t* { return ldx + (b * 2 + lo) * (kRows * SX); };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = wave * (NT * 16);
    const bool wave_active = n0 < H;
    const int t_beg = blockIdx.x * tiles_per_wg;
    const int t_end = min(t_beg + tiles_per_wg, num_tiles);
    if (t_beg >= t_end) return;

    bf16x8 wh[KS][NT], wl[KS][NT];
    int cur_rel = -1;
    int32_t nidx[P];
    float4 rx[P];
    auto load_idx = [&](int t) {
        if (t >= t_end) return;
        const Chunk tl = tiles[t];
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int piece = tid + j * kThreads, p = tl.beg + piece / (H / 4);
            nidx[j] = (piece < NP && p < tl.end) ? (idx ? idx[p] : p) : -1;
        }
    };
    auto load_rows = [&]() {
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int c = (tid + j * kThreads) % (H / 4);
            rx[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (nidx[j] >= 0) {
                const float* base = nidx[j] < n1 ? X + (size_t)nidx[j] * H : X2 + (size_t)(nidx[j] - n1) * H;
                rx[j] = *reinterpret_cast<const float4*>(base + c * 4);
            }
        }
    };
    auto store_rows = [&](int b) {
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int piece = tid + j * kThreads, r = piece / (H / 4), c = piece % (H / 4);
            if (piece < NP) {
                bf16x4 h, l;
                split4(rx[j], h, l);
                *reinterpret_cast<bf16x4*>(bufX(b, 0) + r * SX + c * 4) = h;
                *reinterpret_cast<bf16x4*>(bufX(b, 1) + r * SX + c * 4) = l;
            }
        }
    };
    load_idx(t_beg);
    load_rows();
    store_rows(0);
    load_idx(t_beg + 1);
    __syncthreads();
    for (int t = t_beg; t < t_end; ++t) {
        const int b = (t - t_beg) & 1;
        const Chunk tl = tiles[t];
        if (t + 1 < t_end) load_rows();
        if (tl.rel != cur_rel && wave_active) {
            cur_rel = tl.rel;
            const float* w = wform.matrix(Wn, cur_rel, H);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int n = n0 + nt * 16 + (lane & 15), k = ks * 32 + 8 * (lane >> 4);
                    if (wform.kn) {                             // [k][n] as the parameter is stored: 16 lanes share a 64-byte run
                        const float* wp = w + (size_t)k * H + n;
                        split8(make_float4(wp[0], wp[H], wp[2 * H], wp[3 * H]),
                               make_float4(wp[4 * H], wp[5 * H], wp[6 * H], wp[7 * H]), wh[ks][nt], wl[ks][nt]);
                    } else {
                        const float* wp = w + (size_t)n * H + k;
                        split8(*reinterpret_cast<const float4*>(wp), *reinterpret_cast<const float4*>(wp + 4), wh[ks][nt], wl[ks][nt]);
                    }
                }
        }
        if (wave_active) {
            f32x4 acc[MT][NT];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
            const bf16_t* xh = bufX(b, 0);
            const bf16_t* xl = bufX(b, 1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 fh[MT], fl[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    fh[m] = *reinterpret_cast<const bf16x8*>(xh + (m * 16 + (lane & 15)) * SX + ks * 32 + 8 * (lane >> 4));
                    fl[m] = *reinterpret_cast<const bf16x8*>(xl + (m * 16 + (lane & 15)) * SX + ks * 32 + 8 * (lane >> 4));
                }
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ks][n], fh[m], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ks][n], fl[m], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ks][n], fh[m], acc[m][n], 0, 0, 0);
                    }
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int col = n0 + n * 16 + 4 * (lane >> 4);
                    float4 v = make_float4(acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]);
                    if (const float* brow = wform.bias_row(bias, cur_rel, H)) {
                        const float4 bv = *reinterpret_cast<const float4*>(brow + col);
                        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                    }
                    if (relu) { v.x = dn_act(v.x, slope); v.y = dn_act(v.y, slope); v.z = dn_act(v.z, slope); v.w = dn_act(v.w, slope); }
                    *reinterpret_cast<float4*>(ldy + (m * 16 + (lane & 15)) * SY + col) = v;
                }
        }
        if (t + 1 < t_end) store_rows(b ^ 1);
        load_idx(t + 2);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int piece = tid + j * kThreads, r = piece / (H / 4), c = piece % (H / 4);
            const int p = tl.beg + r;
            if (piece < NP && p < tl.end) {
                float4 v = *reinterpret_cast<const float4*>(ldy + r * SY + c * 4);
                if (mask_pos) {
                    const float4 mk = *reinterpret_cast<const float4*>(mask_pos + (size_t)p * H + c * 4);
                    v.x = mk.x > 0.f ? v.x : dn_neg(v.x, slope); v.y = mk.y > 0.f ? v.y : dn_neg(v.y, slope);
                    v.z = mk.z > 0.f ? v.z : dn_neg(v.z, slope); v.w = mk.w > 0.f ? v.w : dn_neg(v.w, slope);
                }
                *reinterpret_cast<float4*>(Y + (size_t)p * H + c * 4) = v;
            }
        }
        __syncthreads();
    }
}

// Two dense layers in one pass over fp32 rows on the 3-term split (dn_rows_chain2_f32; H = 64 / 128: the widths whose launches are
// chains of latency):  X0 = mask0 ? keep-or-scale(X, mask0) : X;  Y1 = mask1 ? keep-or-scale(epi1(X0 W1), mask1) : epi1(X0 W1);
// Y2 = epi2(Y1 W2), epi = (+ bias) then the optional activation.  Serves the forward of the reference MLP (Linear-act-Linear-act,
// rgin.py:50-57: no masks, both biases) and the backward's input-gradient chain (mask0 = the saved output: the outer activation's
// mask, W1 = Linear 2's weight as stored ([k][n]), mask1 = the saved hidden rows, W2 = Linear 1's weight) -- one launch where the
// separate ones were two (forward) and three (backward: dn_relu_bwd_f32 + two dn_rows_transform_f32).  Per 32-row tile: rows -> hi / lo
// bf16 tiles in LDS (the next tile's rows are in flight under this one), stage-1 MFMAs, the fp32 tile through LDS to Y1 and -- split
// again -- to the stage-2 operand tiles, stage-2 MFMAs, the tile through LDS to Y2.  Both weight slices stay in registers.
template <int H>
__global__ __launch_bounds__(kThreads, 2) void rows_chain2_f32s_kernel(const float* __restrict__ X, const float* __restrict__ W1,
                                                                       const float* __restrict__ b1, int32_t relu1,
                                                                       const float* __restrict__ mask0, const float* __restrict__ mask1,
                                                                       const float* __restrict__ W2, const float* __restrict__ b2,
                                                                       int32_t relu2, int32_t N, int32_t tiles_per_wg,
                                                                       float* __restrict__ Y1, float* __restrict__ Y2, int32_t w_kn,
                                                                       float slope, const float* __restrict__ residual,
                                                                       float* __restrict__ Y2_plus) {
    static_assert(H == 64 || H == 128, "unsupported width");
    constexpr int SX = H + 8;                                   // bf16 elements per LDS row of an operand tile
    constexpr int SY = H + 4;                                   // floats per LDS row of the output tile
    constexpr int KS = H / 32;
    constexpr int NT = 1;                                       // a wave owns 16 output columns (H / 16 of the 8 waves are active)
    constexpr int MT = kRows / 16;
    constexpr int NP = kRows * H / 4;
    constexpr int P = (NP + kThreads - 1) / kThreads;
    __shared__ __attribute__((aligned(16))) bf16_t ldx[2 * 2 * kRows * SX];   // stage-1 operand tiles: [buffer][hi / lo]
    __shared__ __attribute__((aligned(16))) bf16_t ldz[2 * kRows * SX];       // stage-2 operand tiles: [hi / lo]
    __shared__ __attribute__((aligned(16))) float ldy[kRows * SY];
    auto bufX = [&](int b, int lo) -> bf16_t* { return ldx + (b * 2 + lo) * (kRows * SX); };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = wave * 16;
    const bool wave_active = n0 < H;
    const int num_tiles = (N + kRows - 1) / kRows;
    const int t_beg = blockIdx.x * tiles_per_wg;
    const int t_end = min(t_beg + tiles_per_wg, num_tiles);
    if (t_beg >= t_end) return;

    bf16x8 wh1[KS], wl1[KS], wh2[KS], wl2[KS];
    auto load_w = [&](const float* w, bool kn, bf16x8* wh, bf16x8* wl) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int n = n0 + (lane & 15), k = ks * 32 + 8 * (lane >> 4);
            if (kn) {                                           // [k][n] as the parameter is stored
                const float* wp = w + (size_t)k * H + n;
                split8(make_float4(wp[0], wp[H], wp[2 * H], wp[3 * H]), make_float4(wp[4 * H], wp[5 * H], wp[6 * H], wp[7 * H]), wh[ks], wl[ks]);
            } else {
                const float* wp = w + (size_t)n * H + k;
                split8(*reinterpret_cast<const float4*>(wp), *reinterpret_cast<const float4*>(wp + 4), wh[ks], wl[ks]);
            }
        }
    };
    if (wave_active) {
        load_w(W1, (w_kn & 1) != 0, wh1, wl1);
        load_w(W2, (w_kn & 2) != 0, wh2, wl2);
    }
    float4 rx[P], rm[P];
    auto load_rows = [&](int t) {                               // the tile's rows (and their mask rows) into registers
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int piece = tid + j * kThreads, p = t * kRows + piece / (H / 4), c = piece % (H / 4);
            rx[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            rm[j] = make_float4(1.f, 1.f, 1.f, 1.f);
            if (piece < NP && p < N) {
                rx[j] = *reinterpret_cast<const float4*>(X + (size_t)p * H + c * 4);
                if (mask0) rm[j] = *reinterpret_cast<const float4*>(mask0 + (size_t)p * H + c * 4);
            }
        }
    };
    auto keep = [&](float4& v, const float4& mk) {
        v.x = mk.x > 0.f ? v.x : dn_neg(v.x, slope); v.y = mk.y > 0.f ? v.y : dn_neg(v.y, slope);
        v.z = mk.z > 0.f ? v.z : dn_neg(v.z, slope); v.w = mk.w > 0.f ? v.w : dn_neg(v.w, slope);
    };
    auto store_rows = [&](int b) {
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int piece = tid + j * kThreads, r = piece / (H / 4), c = piece % (H / 4);
            if (piece < NP) {
                if (mask0) keep(rx[j], rm[j]);
                bf16x4 h, l;
                split4(rx[j], h, l);
                *reinterpret_cast<bf16x4*>(bufX(b, 0) + r * SX + c * 4) = h;
                *reinterpret_cast<bf16x4*>(bufX(b, 1) + r * SX + c * 4) = l;
            }
        }
    };
    // one stage: acc = W (registers) x the hi / lo tiles, epilogue (bias, activation) into the fp32 tile
    auto stage = [&](const bf16_t* xh, const bf16_t* xl, const bf16x8* wh, const bf16x8* wl, const float* bias, int32_t relu) {
        if (!wave_active) return;
        f32x4 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 fh[MT], fl[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                fh[m] = *reinterpret_cast<const bf16x8*>(xh + (m * 16 + (lane & 15)) * SX + ks * 32 + 8 * (lane >> 4));
                fl[m] = *reinterpret_cast<const bf16x8*>(xl + (m * 16 + (lane & 15)) * SX + ks * 32 + 8 * (lane >> 4));
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ks], fh[m], acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ks], fl[m], acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ks], fh[m], acc[m], 0, 0, 0);
            }
        }
        const int col = n0 + 4 * (lane >> 4);
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) bv = *reinterpret_cast<const float4*>(bias + col);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float4 v = make_float4(acc[m][0] + bv.x, acc[m][1] + bv.y, acc[m][2] + bv.z, acc[m][3] + bv.w);
            if (relu) { v.x = dn_act(v.x, slope); v.y = dn_act(v.y, slope); v.z = dn_act(v.z, slope); v.w = dn_act(v.w, slope); }
            *reinterpret_cast<float4*>(ldy + (m * 16 + (lane & 15)) * SY + col) = v;
        }
    };
    load_rows(t_beg);
    store_rows(0);
    __syncthreads();
    for (int t = t_beg; t < t_end; ++t) {
        const int b = (t - t_beg) & 1;
        if (t + 1 < t_end) load_rows(t + 1);                    // in flight under both stages of this tile
        float4 rs[P];
        if (residual) {                                         // ... as are the rows the output is added to (Y2_plus = Y2 + residual)
#pragma unroll
            for (int j = 0; j < P; ++j) {
                const int piece = tid + j * kThreads, p = t * kRows + piece / (H / 4), c = piece % (H / 4);
                rs[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (piece < NP && p < N) rs[j] = *reinterpret_cast<const float4*>(residual + (size_t)p * H + c * 4);
            }
        }
        float4 m1[P];
        if (mask1) {                                            // ... as are this tile's stage-1 mask rows
#pragma unroll
            for (int j = 0; j < P; ++j) {
                const int piece = tid + j * kThreads, p = t * kRows + piece / (H / 4), c = piece % (H / 4);
                m1[j] = make_float4(1.f, 1.f, 1.f, 1.f);
                if (piece < NP && p < N) m1[j] = *reinterpret_cast<const float4*>(mask1 + (size_t)p * H + c * 4);
            }
        }
        stage(bufX(b, 0), bufX(b, 1), wh1, wl1, b1, relu1);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < P; ++j) {                           // Y1 out, and split into the stage-2 operand tiles
            const int piece = tid + j * kThreads, r = piece / (H / 4), c = piece % (H / 4);
            const int p = t * kRows + r;
            if (piece < NP) {
                float4 v = *reinterpret_cast<const float4*>(ldy + r * SY + c * 4);
                if (mask1) keep(v, m1[j]);
                if (p < N) *reinterpret_cast<float4*>(Y1 + (size_t)p * H + c * 4) = v;
                else v = make_float4(0.f, 0.f, 0.f, 0.f);
                bf16x4 h, l;
                split4(v, h, l);
                *reinterpret_cast<bf16x4*>(ldz + r * SX + c * 4) = h;
                *reinterpret_cast<bf16x4*>(ldz + kRows * SX + r * SX + c * 4) = l;
            }
        }
        __syncthreads();
        stage(ldz, ldz + kRows * SX, wh2, wl2, b2, relu2);
        if (t + 1 < t_end) store_rows(b ^ 1);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int piece = tid + j * kThreads, r = piece / (H / 4), c = piece % (H / 4);
            const int p = t * kRows + r;
            if (piece < NP && p < N) {
                const float4 v = *reinterpret_cast<const float4*>(ldy + r * SY + c * 4);
                *reinterpret_cast<float4*>(Y2 + (size_t)p * H + c * 4) = v;
                if (residual)
                    *reinterpret_cast<float4*>(Y2_plus + (size_t)p * H + c * 4) = make_float4(v.x + rs[j].x, v.y + rs[j].y, v.z + rs[j].z, v.w + rs[j].w);
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void relu_bwd_f32_kernel(const float4* __restrict__ g, const float4* __restrict__ y,
                                                           float4* __restrict__ out, int64_t n4, float slope) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 gv = g[i], yv = y[i];
        out[i] = make_float4(yv.x > 0.f ? gv.x : dn_neg(gv.x, slope), yv.y > 0.f ? gv.y : dn_neg(gv.y, slope),
                             yv.z > 0.f ? gv.z : dn_neg(gv.z, slope), yv.w > 0.f ? gv.w : dn_neg(gv.w, slope));
    }
}

template <int H>
int launch_wgrad(const float* A, const float* A2, int32_t na1, const int32_t* ia, const float* G, const float* G2, int32_t ng1,
                 const int32_t* ig, const Chunk* chunks, int64_t num_chunks, float* partial, int32_t colsum_of, float* csp,
                 const float* maskA, float* A_out, float slope, int32_t exact, hipStream_t st) {
    if (exact)
        hipLaunchKernelGGL((rows_wgrad_f32_kernel<H>), dim3((unsigned)num_chunks), dim3(kThreads), 0, st, A, A2, na1, ia, G, G2,
                           ng1, ig, chunks, partial, colsum_of, csp, maskA, A_out, slope);
    else
        hipLaunchKernelGGL((rows_wgrad_f32s_kernel<H>), dim3((unsigned)num_chunks), dim3(kThreads), 0, st, A, A2, na1, ia, G, G2,
                           ng1, ig, chunks, partial, colsum_of, csp, maskA, A_out, slope);
    DN_CHECK_LAUNCH();
    return DN_OK;
}

template <int H>
int launch_transform(const float* X, const float* X2, int32_t n1, const int32_t* idx, const float* Wn, const float* bias,
                     int32_t relu, float slope, const float* mask_pos, const Chunk* tiles, int64_t num_tiles, float* Y, int32_t exact,
                     WeightForm wform, hipStream_t st) {
    const int64_t max_wg = 256 * (H == 256 ? 1 : 2);             // LDS: one 100 KB workgroup per CU at H = 256
    const int64_t tiles_per_wg = dn_cdiv(num_tiles, max_wg);
    const int64_t grid = dn_cdiv(num_tiles, tiles_per_wg);
    if (exact)
        hipLaunchKernelGGL((rows_transform_f32_kernel<H>), dim3((unsigned)grid), dim3(kThreads), 0, st, X, X2, n1, idx, Wn, bias,
                           relu, slope, mask_pos, tiles, (int32_t)num_tiles, (int32_t)tiles_per_wg, Y, wform);
    else
        hipLaunchKernelGGL((rows_transform_f32s_kernel<H>), dim3((unsigned)grid), dim3(kThreads), 0, st, X, X2, n1, idx, Wn, bias,
                           relu, slope, mask_pos, tiles, (int32_t)num_tiles, (int32_t)tiles_per_wg, Y, wform);
    DN_CHECK_LAUNCH();
    return DN_OK;
}

}  // namespace

extern "C" {

int dn_relu_bwd_f32(const float* g, const float* y, float* out, int64_t numel, float act_slope, dn_stream_t stream) {
    DN_REQUIRE(numel >= 0 && numel % 4 == 0, "dn_relu_bwd_f32: numel must be a non-negative multiple of 4");
    if (numel == 0) return DN_OK;
    DN_REQUIRE(g && y && out, "dn_relu_bwd_f32: NULL pointer");
    DN_REQUIRE((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(out)) % 16 == 0,
               "dn_relu_bwd_f32: unaligned pointer");
    const int64_t n4 = numel / 4;
    const int64_t grid = dn_cdiv(n4, 256) < 256 * 16 ? dn_cdiv(n4, 256) : 256 * 16;
    hipLaunchKernelGGL(relu_bwd_f32_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const float4*)g,
                       (const float4*)y, (float4*)out, n4, act_slope);
    DN_CHECK_LAUNCH();
    return DN_OK;
}

int dn_rows_wgrad_f32(const float* A, const float* A2, int32_t na1, const int32_t* idx_a, const float* G, const float* G2,
                      int32_t ng1, const int32_t* idx_g, int32_t Hi, int32_t Ho, int64_t R, const int32_t* chunks,
                      int64_t num_chunks, const int32_t* chunk_ptr, float* out, int32_t colsum_of, float* out_colsum,
                      const float* mask_a, float* a_out, int32_t precision, float act_slope, void* workspace, size_t workspace_bytes,
                      dn_stream_t stream) {
    DN_REQUIRE(R >= 0 && num_chunks >= 0, "dn_rows_wgrad_f32: negative size");
    DN_REQUIRE(precision == 0 || precision == 1, "dn_rows_wgrad_f32: precision must be 0 (bf16 split) or 1 (exact f32)");
    DN_REQUIRE(Hi == Ho && (Hi == 64 || Hi == 128 || Hi == 256), "dn_rows_wgrad_f32: unsupported widths %d x %d "
               "(square 64/128/256 only)", Hi, Ho);
    DN_REQUIRE(A2 != nullptr || na1 == 0x7fffffff, "dn_rows_wgrad_f32: A2 == NULL requires na1 == INT32_MAX");
    DN_REQUIRE(G2 != nullptr || ng1 == 0x7fffffff, "dn_rows_wgrad_f32: G2 == NULL requires ng1 == INT32_MAX");
    DN_REQUIRE(colsum_of >= 0 && colsum_of <= 2 && (colsum_of == 0 || out_colsum != nullptr), "dn_rows_wgrad_f32: bad colsum arguments");
    DN_REQUIRE(mask_a == nullptr || A2 == nullptr, "dn_rows_wgrad_f32: mask_a needs a single A source");
    DN_REQUIRE(a_out == nullptr || (mask_a != nullptr && idx_a == nullptr), "dn_rows_wgrad_f32: a_out needs mask_a and idx_a == NULL");
    if (R == 0) return DN_OK;
    DN_REQUIRE(out && chunk_ptr, "dn_rows_wgrad_f32: NULL pointer");
    DN_REQUIRE(num_chunks == 0 || (A && G && chunks && workspace), "dn_rows_wgrad_f32: NULL pointer");
    DN_REQUIRE(workspace_bytes >= (size_t)num_chunks * ((size_t)Hi * Ho + Hi) * sizeof(float), "dn_rows_wgrad_f32: workspace too small");
    DN_REQUIRE((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(A2) | reinterpret_cast<uintptr_t>(G) |
                reinterpret_cast<uintptr_t>(G2) | reinterpret_cast<uintptr_t>(mask_a) | reinterpret_cast<uintptr_t>(a_out)) % 16 == 0,
               "dn_rows_wgrad_f32: unaligned pointer");
    hipStream_t st = (hipStream_t)stream;
    const Chunk* ch = reinterpret_cast<const Chunk*>(chunks);
    float* ws = (float*)workspace;
    float* csp = ws + (size_t)num_chunks * Hi * Ho;
    int rc = DN_OK;
    if (num_chunks > 0) {
        if (Hi == 256) rc = launch_wgrad<256>(A, A2, na1, idx_a, G, G2, ng1, idx_g, ch, num_chunks, ws, colsum_of, csp, mask_a, a_out, act_slope, precision, st);
        else if (Hi == 128) rc = launch_wgrad<128>(A, A2, na1, idx_a, G, G2, ng1, idx_g, ch, num_chunks, ws, colsum_of, csp, mask_a, a_out, act_slope, precision, st);
        else rc = launch_wgrad<64>(A, A2, na1, idx_a, G, G2, ng1, idx_g, ch, num_chunks, ws, colsum_of, csp, mask_a, a_out, act_slope, precision, st);
        if (rc != DN_OK) return rc;
    }
    const int64_t tile = (int64_t)Hi * Ho;
    const float* cspc = colsum_of ? csp : nullptr;
    dim3 grid((unsigned)(dn_cdiv(tile, 32) + (cspc ? dn_cdiv(Hi, 32) : 0)), (unsigned)R);
    hipLaunchKernelGGL(wgrad_reduce_f32_kernel, grid, dim3(256), 0, st, (const float*)ws, chunk_ptr, tile, out, cspc, Hi, out_colsum);
    DN_CHECK_LAUNCH();
    return DN_OK;
}

int dn_rows_wgrad_multi_f32(const dn_wgrad_job* jobs, int32_t num_jobs, int32_t H, int64_t R, const int32_t* chunks, int64_t num_chunks,
                            const int32_t* chunk_ptr, float* out, float* out_colsum, void* workspace, size_t workspace_bytes,
                            dn_stream_t stream) {
    DN_REQUIRE(jobs && num_jobs >= 1 && num_jobs <= 3, "dn_rows_wgrad_multi_f32: 1 .. 3 jobs");
    DN_REQUIRE(H == 64 || H == 128, "dn_rows_wgrad_multi_f32: unsupported width %d (64 / 128: the widths whose launches are latency)", H);
    DN_REQUIRE(R >= 1 && num_chunks >= 0, "dn_rows_wgrad_multi_f32: bad sizes");
    DN_REQUIRE(out && chunk_ptr && out_colsum, "dn_rows_wgrad_multi_f32: NULL pointer");
    DN_REQUIRE(num_chunks == 0 || (chunks && workspace), "dn_rows_wgrad_multi_f32: NULL pointer");
    DN_REQUIRE(workspace_bytes >= (size_t)num_chunks * ((size_t)H * H + H) * sizeof(float), "dn_rows_wgrad_multi_f32: workspace too small");
    DN_REQUIRE((reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(out_colsum)) % 16 == 0,
               "dn_rows_wgrad_multi_f32: workspace / out_colsum must be 16-byte aligned");
    WgJobsF wj;
    wj.n = num_jobs;
    for (int k = 0; k < 3; ++k) {
        const dn_wgrad_job& q = jobs[k < num_jobs ? k : 0];
        DN_REQUIRE(q.A && q.G, "dn_rows_wgrad_multi_f32: NULL operand");
        DN_REQUIRE((reinterpret_cast<uintptr_t>(q.A) | reinterpret_cast<uintptr_t>(q.G) | reinterpret_cast<uintptr_t>(q.A2) |
                    reinterpret_cast<uintptr_t>(q.G2)) % 16 == 0, "dn_rows_wgrad_multi_f32: unaligned input");
        DN_REQUIRE(q.A2 != nullptr || q.na1 == 0x7fffffff, "dn_rows_wgrad_multi_f32: A2 == NULL requires na1 == INT32_MAX");
        DN_REQUIRE(q.G2 != nullptr || q.ng1 == 0x7fffffff, "dn_rows_wgrad_multi_f32: G2 == NULL requires ng1 == INT32_MAX");
        DN_REQUIRE(q.colsum_of >= 0 && q.colsum_of <= 2 && q.first_rel >= 0 && q.row0 >= 0, "dn_rows_wgrad_multi_f32: bad job");
        DN_REQUIRE(q.mask_a_bits == nullptr || (q.A2 == nullptr && q.idx_a == nullptr && reinterpret_cast<uintptr_t>(q.mask_a_bits) % 16 == 0),
                   "dn_rows_wgrad_multi_f32: a mask excludes A2 / idx_a");
        DN_REQUIRE(k == 0 || k >= num_jobs || q.first_rel > jobs[k - 1].first_rel, "dn_rows_wgrad_multi_f32: jobs must ascend in first_rel");
        wj.j[k] = WgJobF{(const float*)q.A, (const float*)q.A2, q.idx_a, (const float*)q.G, (const float*)q.G2, q.idx_g,
                         (const float*)q.mask_a_bits, q.na1, q.ng1, q.colsum_of, q.first_rel, q.row0, q.act_slope};
    }
    DN_REQUIRE(jobs[0].first_rel == 0, "dn_rows_wgrad_multi_f32: the first job starts at relation 0");
    hipStream_t st = (hipStream_t)stream;
    const Chunk* ch = reinterpret_cast<const Chunk*>(chunks);
    float* ws = (float*)workspace;
    const int64_t tile = (int64_t)H * H;
    float* csp = ws + (size_t)num_chunks * tile;
    if (num_chunks > 0) {
        if (H == 128) hipLaunchKernelGGL((rows_wgrad_f32s_multi_kernel<128>), dim3((unsigned)num_chunks), dim3(kThreads), 0, st, wj, ch, ws, csp);
        else hipLaunchKernelGGL((rows_wgrad_f32s_multi_kernel<64>), dim3((unsigned)num_chunks), dim3(kThreads), 0, st, wj, ch, ws, csp);
        DN_CHECK_LAUNCH();
    }
    dim3 grid((unsigned)(dn_cdiv(tile, 32) + dn_cdiv(H, 32)), (unsigned)R);
    hipLaunchKernelGGL(wgrad_reduce_f32_kernel, grid, dim3(256), 0, st, (const float*)ws, chunk_ptr, tile, out, (const float*)csp, H, out_colsum);
    DN_CHECK_LAUNCH();
    return DN_OK;
}

int dn_rows_chain2_f32(const float* X, int32_t H, const float* W1, const float* b1, int32_t relu1, const float* mask0, const float* mask1,
                       const float* W2, const float* b2, int32_t relu2, int64_t N, float* Y1, float* Y2, int32_t w_kn, float act_slope,
                       const float* residual, float* Y2_plus, dn_stream_t stream) {
    DN_REQUIRE((residual == nullptr) == (Y2_plus == nullptr), "dn_rows_chain2_f32: residual and Y2_plus come together");
    DN_REQUIRE((reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(Y2_plus)) % 16 == 0, "dn_rows_chain2_f32: unaligned pointer");
    DN_REQUIRE(H == 64 || H == 128, "dn_rows_chain2_f32: unsupported width %d (64 / 128 only)", H);
    DN_REQUIRE(N >= 0 && N < 0x7fffffffLL && w_kn >= 0 && w_kn <= 3, "dn_rows_chain2_f32: bad arguments");
    if (N == 0) return DN_OK;
    DN_REQUIRE(X && W1 && W2 && Y1 && Y2, "dn_rows_chain2_f32: NULL pointer");
    DN_REQUIRE((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W1) | reinterpret_cast<uintptr_t>(W2) |
                reinterpret_cast<uintptr_t>(b1) | reinterpret_cast<uintptr_t>(b2) | reinterpret_cast<uintptr_t>(mask0) |
                reinterpret_cast<uintptr_t>(mask1) | reinterpret_cast<uintptr_t>(Y1) | reinterpret_cast<uintptr_t>(Y2)) % 16 == 0,
               "dn_rows_chain2_f32: unaligned pointer");
    const int64_t num_tiles = dn_cdiv(N, (int64_t)kRows);
    const int64_t tiles_per_wg = dn_cdiv(num_tiles, (int64_t)512);            // two workgroups per CU
    const int64_t grid = dn_cdiv(num_tiles, tiles_per_wg);
    hipStream_t st = (hipStream_t)stream;
    if (H == 64)
        hipLaunchKernelGGL((rows_chain2_f32s_kernel<64>), dim3((unsigned)grid), dim3(kThreads), 0, st, X, W1, b1, relu1, mask0, mask1, W2, b2,
                           relu2, (int32_t)N, (int32_t)tiles_per_wg, Y1, Y2, w_kn, act_slope, residual, Y2_plus);
    else
        hipLaunchKernelGGL((rows_chain2_f32s_kernel<128>), dim3((unsigned)grid), dim3(kThreads), 0, st, X, W1, b1, relu1, mask0, mask1, W2, b2,
                           relu2, (int32_t)N, (int32_t)tiles_per_wg, Y1, Y2, w_kn, act_slope, residual, Y2_plus);
    DN_CHECK_LAUNCH();
    return DN_OK;
}

int dn_rows_transform_f32(const float* X, const float* X2, int32_t n1, const int32_t* idx, int32_t Hi, int32_t Ho,
                          const float* Wn, const float* bias, int32_t relu, const float* mask_pos, const int32_t* tiles,
                          int64_t num_tiles, float* Y, int32_t precision, float act_slope, const float* W_loop, int32_t loop_rel,
                          int32_t bias_rel, int32_t w_kn, dn_stream_t stream) {
    DN_REQUIRE(num_tiles >= 0 && num_tiles < 0x7fffffffLL, "dn_rows_transform_f32: bad tile count");
    DN_REQUIRE(precision == 0 || precision == 1, "dn_rows_transform_f32: precision must be 0 (bf16 split) or 1 (exact f32)");
    DN_REQUIRE(Hi == Ho && (Hi == 64 || Hi == 128 || Hi == 256), "dn_rows_transform_f32: unsupported widths %d x %d "
               "(square 64/128/256 only)", Hi, Ho);
    if (num_tiles == 0) return DN_OK;
    DN_REQUIRE(X && Wn && tiles && Y, "dn_rows_transform_f32: NULL pointer");
    DN_REQUIRE(W_loop != nullptr || loop_rel < 0, "dn_rows_transform_f32: loop_rel >= 0 needs W_loop");
    DN_REQUIRE(reinterpret_cast<uintptr_t>(W_loop) % 16 == 0, "dn_rows_transform_f32: unaligned pointer");
    const WeightForm wform{W_loop, W_loop ? loop_rel : -1, bias_rel, w_kn ? 1 : 0};
    DN_REQUIRE(X2 != nullptr || n1 == 0x7fffffff, "dn_rows_transform_f32: X2 == NULL requires n1 == INT32_MAX");
    DN_REQUIRE((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(X2) | reinterpret_cast<uintptr_t>(Wn) |
                reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(mask_pos)) % 16 == 0,
               "dn_rows_transform_f32: unaligned pointer");
    hipStream_t st = (hipStream_t)stream;
    const Chunk* tl = reinterpret_cast<const Chunk*>(tiles);
    if (Hi == 256) return launch_transform<256>(X, X2, n1, idx, Wn, bias, relu, act_slope, mask_pos, tl, num_tiles, Y, precision, wform, st);
    if (Hi == 128) return launch_transform<128>(X, X2, n1, idx, Wn, bias, relu, act_slope, mask_pos, tl, num_tiles, Y, precision, wform, st);
    return launch_transform<64>(X, X2, n1, idx, Wn, bias, relu, act_slope, mask_pos, tl, num_tiles, Y, precision, wform, st);
}

}  // extern "C"
