// Gather + segment-reduce kernels (gfx950 / CDNA4): the message-passing core of the GIN / RGCN / RGIN
// layers and the per-graph readouts.  HBM/L2-bound row traffic, no matrix work, no atomics.
//
// Layout: feature matrices row-major [rows, H].  A "row group" of LPR lanes (power of two, 16 bytes per
// lane) owns one destination segment at a time: the group walks the segment's element list, each lane
// loading one 16-byte piece of every gathered row (a 64-lane wave covers 64/LPR segments), accumulates
// in fp32 registers and writes the finished row once.  Gather indices are fetched coalesced, LPR at a
// time, and broadcast inside the group with ds_bpermute (__shfl), so the dependent index->row load
// chain is paid once per LPR elements, and four row loads are kept in flight per group.
// Workgroups take CONTIGUOUS chunks of segments, and the block->chunk map is XCD-aware
// (dn_common.h): batched graphs are block-diagonal, so a chunk's source rows sit in the same few KB
// and are re-read from that XCD's L2 rather than from HBM.
#include "dn_common.h"
#include "../../include/dn_hip.h"

namespace {

constexpr int kBlock = 256;
constexpr int kSegsPerGroup = 2;  // segments a row group walks per block (chunk = groups * this)

typedef __bf16 bf16_t;

template <typename T> struct Vec;
template <> struct Vec<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <> struct Vec<bf16_t> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
        const uint4 t = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
        typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (bf16_t)v[i];  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
        *reinterpret_cast<bf16x8*>(p) = o;
    }
};

template <typename T> __device__ __forceinline__ float to_f32(T x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x) { return (T)x; }

// ---------------------------------------------------------------------------------------------
// Vector path: H % Vec<T>::N == 0, rows 16-byte aligned.  LPR lanes x 16 B cover one pass of a row;
// rows wider than LPR*16 B are covered in ceil(H / (LPR*VN)) column passes.
// ---------------------------------------------------------------------------------------------
template <typename T, int LPR, bool HAS_SCALE>
__global__ __launch_bounds__(kBlock) void gather_segsum_vec_kernel(
    const T* __restrict__ in, const int32_t* __restrict__ idx, const float* __restrict__ scale,
    const int32_t* __restrict__ ptr, int64_t S, int32_t H, T* __restrict__ out, const T* __restrict__ self_in,
    float self_coef, int32_t mean, int64_t nchunks) {
    // A lane group walks its segments one after the other; what it needs for a segment is a chain of three dependent loads
    // (ptr -> idx -> rows), each an L2 / HBM round trip.  The chain is software-pipelined ACROSS segments: while the rows of
    // segment i are in flight, the first LPR indices (and scales) of segment i+1 and the bounds of segment i+2 are too, and
    // the rows themselves go 8 at a time.  (Graph-local gathers are L2-latency-bound, not HBM-bound: DESIGN.md 4.)
    constexpr int VN = Vec<T>::N;
    constexpr int GPB = kBlock / LPR;
    constexpr int KU = 8;                                     // row loads in flight per lane
    const int lane = threadIdx.x % LPR;
    const int group = threadIdx.x / LPR;
    const int64_t chunk = dn_xcd_chunk(blockIdx.x, gridDim.x);
    if (chunk >= nchunks) return;
    const int64_t seg0 = chunk * (GPB * kSegsPerGroup);
    // interleave groups over the chunk so that a wave's groups touch neighbouring segments
    auto seg_of = [&](int it) -> int64_t { return seg0 + (int64_t)it * GPB + group; };
    auto bounds = [&](int it, int& beg, int& end) {
        const int64_t s = seg_of(it);
        beg = end = 0;
        if (it < kSegsPerGroup && s < S) {
            if (ptr != nullptr) { beg = ptr[s]; end = ptr[s + 1]; } else { beg = (int)s; end = (int)s + 1; }
        }
    };
    auto first_idx = [&](int beg, int end, int& my, float& mysc) {
        my = 0;
        mysc = 1.f;
        if (lane < end - beg) {
            my = idx != nullptr ? idx[beg + lane] : beg + lane;
            if (HAS_SCALE) mysc = scale[beg + lane];
        }
    };

    for (int col0 = 0; col0 < H; col0 += LPR * VN) {
        const int col = col0 + lane * VN;
        const bool colok = col < H;
        int beg, end, beg1, end1, my, my1;
        float mysc, mysc1;
        bounds(0, beg, end);
        bounds(1, beg1, end1);
        first_idx(beg, end, my, mysc);
#pragma unroll 1
        for (int it = 0; it < kSegsPerGroup; ++it) {
            const int64_t s = seg_of(it);
            if (s >= S) break;
            int beg2, end2;
            first_idx(beg1, end1, my1, mysc1);                // segment it+1's indices: in flight under this segment's rows
            bounds(it + 2, beg2, end2);
            float acc[VN];
#pragma unroll
            for (int i = 0; i < VN; ++i) acc[i] = 0.f;
            for (int base = beg; base < end; base += LPR) {
                const int n = min(LPR, end - base);
                if (base != beg) {                            // a segment longer than LPR entries (a hub): fetch as we go
                    my = 0;
                    mysc = 1.f;
                    if (lane < n) {
                        my = idx != nullptr ? idx[base + lane] : base + lane;
                        if (HAS_SCALE) mysc = scale[base + lane];
                    }
                }
                for (int j = 0; j < n; j += KU) {
                    float v[KU][VN];
                    float w[KU];
#pragma unroll
                    for (int k = 0; k < KU; ++k) {
                        const int jj = (j + k) & (LPR - 1);
                        const int r = __shfl(my, jj, LPR);
                        w[k] = HAS_SCALE ? __shfl(mysc, jj, LPR) : 1.f;
                        if (j + k < n && colok) {
                            Vec<T>::load(in + (size_t)r * H + col, v[k]);
                        } else {
#pragma unroll
                            for (int i = 0; i < VN; ++i) v[k][i] = 0.f;
                            w[k] = 0.f;
                        }
                    }
#pragma unroll
                    for (int k = 0; k < KU; ++k) {
#pragma unroll
                        for (int i = 0; i < VN; ++i) {
                            if (HAS_SCALE) acc[i] = fmaf(w[k], v[k][i], acc[i]);
                            else acc[i] += v[k][i];
                        }
                    }
                }
            }
            if (colok) {
                if (mean) {
                    const float inv = end > beg ? 1.f / (float)(end - beg) : 0.f;
#pragma unroll
                    for (int i = 0; i < VN; ++i) acc[i] *= inv;
                }
                if (self_in != nullptr) {
                    float sv[VN];
                    Vec<T>::load(self_in + (size_t)s * H + col, sv);
#pragma unroll
                    for (int i = 0; i < VN; ++i) acc[i] = fmaf(self_coef, sv[i], acc[i]);
                }
                Vec<T>::store(out + (size_t)s * H + col, acc);
            }
            beg = beg1; end = end1; my = my1; mysc = mysc1;
            beg1 = beg2; end1 = end2;
        }
    }
}

// The round-1 form of the vector path (one segment at a time, 4 row loads in flight): kept for long CONTIGUOUS lists (the
// pre-aggregation of the collapsed dummy relations), where it measures faster than the pipelined form above.
template <typename T, int LPR, bool HAS_SCALE>
__global__ __launch_bounds__(kBlock) void gather_segsum_vec1_kernel(
    const T* __restrict__ in, const int32_t* __restrict__ idx, const float* __restrict__ scale,
    const int32_t* __restrict__ ptr, int64_t S, int32_t H, T* __restrict__ out, const T* __restrict__ self_in,
    float self_coef, int32_t mean, int64_t nchunks) {
    constexpr int VN = Vec<T>::N;
    constexpr int GPB = kBlock / LPR;
    const int lane = threadIdx.x % LPR;
    const int group = threadIdx.x / LPR;
    const int64_t chunk = dn_xcd_chunk(blockIdx.x, gridDim.x);
    if (chunk >= nchunks) return;
    const int64_t seg0 = chunk * (GPB * kSegsPerGroup);

    for (int col0 = 0; col0 < H; col0 += LPR * VN) {
        const int col = col0 + lane * VN;
        const bool colok = col < H;
#pragma unroll 1
        for (int it = 0; it < kSegsPerGroup; ++it) {
            // interleave groups over the chunk so that a wave's groups touch neighbouring segments
            const int64_t s = seg0 + (int64_t)it * GPB + group;
            if (s >= S) break;
            int beg, end;
            if (ptr != nullptr) { beg = ptr[s]; end = ptr[s + 1]; } else { beg = (int)s; end = (int)s + 1; }
            float acc[VN];
#pragma unroll
            for (int i = 0; i < VN; ++i) acc[i] = 0.f;
            for (int base = beg; base < end; base += LPR) {
                const int n = min(LPR, end - base);
                int my = 0;
                float mysc = 1.f;
                if (lane < n) {
                    my = idx != nullptr ? idx[base + lane] : base + lane;
                    if (HAS_SCALE) mysc = scale[base + lane];
                }
                for (int j = 0; j < n; j += 4) {
                    float v[4][VN];
                    float w[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int jj = (j + k) & (LPR - 1);
                        const int r = __shfl(my, jj, LPR);
                        w[k] = HAS_SCALE ? __shfl(mysc, jj, LPR) : 1.f;
                        if (j + k < n && colok) {
                            Vec<T>::load(in + (size_t)r * H + col, v[k]);
                        } else {
#pragma unroll
                            for (int i = 0; i < VN; ++i) v[k][i] = 0.f;
                            w[k] = 0.f;
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
#pragma unroll
                        for (int i = 0; i < VN; ++i) {
                            if (HAS_SCALE) acc[i] = fmaf(w[k], v[k][i], acc[i]);
                            else acc[i] += v[k][i];
                        }
                    }
                }
            }
            if (colok) {
                if (mean) {
                    const float inv = end > beg ? 1.f / (float)(end - beg) : 0.f;
#pragma unroll
                    for (int i = 0; i < VN; ++i) acc[i] *= inv;
                }
                if (self_in != nullptr) {
                    float sv[VN];
                    Vec<T>::load(self_in + (size_t)s * H + col, sv);
#pragma unroll
                    for (int i = 0; i < VN; ++i) acc[i] = fmaf(self_coef, sv[i], acc[i]);
                }
                Vec<T>::store(out + (size_t)s * H + col, acc);
            }
        }
    }
}
// Few, long segments (the pre-aggregation of a collapsed dummy relation on a small batch: ~500 lists of ~50 rows): one
// WORKGROUP per segment.  Its GPB lane groups take the entries beg + group, beg + group + GPB, ... -- all rows of a ~50-row
// list are in flight after two dependent round trips (bounds -> indices -> rows) instead of a dozen -- and group 0 adds the GPB
// partial sums in a fixed order (bit-reproducible).
template <typename T, int LPR, bool HAS_SCALE>
__global__ __launch_bounds__(kBlock) void gather_segsum_block_kernel(
    const T* __restrict__ in, const int32_t* __restrict__ idx, const float* __restrict__ scale,
    const int32_t* __restrict__ ptr, int64_t S, int32_t H, T* __restrict__ out, const T* __restrict__ self_in,
    float self_coef, int32_t mean) {
    constexpr int VN = Vec<T>::N;
    constexpr int GPB = kBlock / LPR;
    constexpr int KU = 4;
    __shared__ float part[GPB][LPR * VN + 1];
    const int lane = threadIdx.x % LPR;
    const int group = threadIdx.x / LPR;
    const int64_t s = blockIdx.x;
    if (s >= S) return;
    const int beg = ptr[s], end = ptr[s + 1];
    for (int col0 = 0; col0 < H; col0 += LPR * VN) {
        const int col = col0 + lane * VN;
        const bool colok = col < H;
        float acc[VN];
#pragma unroll
        for (int i = 0; i < VN; ++i) acc[i] = 0.f;
        for (int e0 = beg + group; e0 < end; e0 += GPB * KU) {
            int r[KU];
            float w[KU];
#pragma unroll
            for (int k = 0; k < KU; ++k) {
                const int e = e0 + k * GPB;
                r[k] = e < end ? (idx != nullptr ? idx[e] : e) : -1;
                w[k] = (HAS_SCALE && e < end) ? scale[e] : 1.f;
            }
            float v[KU][VN];
#pragma unroll
            for (int k = 0; k < KU; ++k) {
                if (r[k] >= 0 && colok) {
                    Vec<T>::load(in + (size_t)r[k] * H + col, v[k]);
                } else {
#pragma unroll
                    for (int i = 0; i < VN; ++i) v[k][i] = 0.f;
                }
            }
#pragma unroll
            for (int k = 0; k < KU; ++k) {
#pragma unroll
                for (int i = 0; i < VN; ++i) {
                    if (HAS_SCALE) acc[i] = fmaf(w[k], v[k][i], acc[i]);
                    else acc[i] += v[k][i];
                }
            }
        }
        if (col0 > 0) __syncthreads();                        // the previous pass's partials have been read
#pragma unroll
        for (int i = 0; i < VN; ++i) part[group][lane * VN + i] = acc[i];
        __syncthreads();
        if (group == 0 && colok) {
#pragma unroll 4
            for (int g = 1; g < GPB; ++g) {
#pragma unroll
                for (int i = 0; i < VN; ++i) acc[i] += part[g][lane * VN + i];
            }
            if (mean) {
                const float inv = end > beg ? 1.f / (float)(end - beg) : 0.f;
#pragma unroll
                for (int i = 0; i < VN; ++i) acc[i] *= inv;
            }
            if (self_in != nullptr) {
                float sv[VN];
                Vec<T>::load(self_in + (size_t)s * H + col, sv);
#pragma unroll
                for (int i = 0; i < VN; ++i) acc[i] = fmaf(self_coef, sv[i], acc[i]);
            }
            Vec<T>::store(out + (size_t)s * H + col, acc);
        }
    }
}

// Scalar path: any H (one element per lane per pass); used for narrow / unaligned rows
// (e.g. the [N, num_classes] readout of gconv.py:210).
template <typename T, int LPR, bool HAS_SCALE>
__global__ __launch_bounds__(kBlock) void gather_segsum_scalar_kernel(
    const T* __restrict__ in, const int32_t* __restrict__ idx, const float* __restrict__ scale,
    const int32_t* __restrict__ ptr, int64_t S, int32_t H, T* __restrict__ out, const T* __restrict__ self_in,
    float self_coef, int32_t mean, int64_t nchunks) {
    constexpr int GPB = kBlock / LPR;
    const int lane = threadIdx.x % LPR;
    const int group = threadIdx.x / LPR;
    const int64_t chunk = dn_xcd_chunk(blockIdx.x, gridDim.x);
    if (chunk >= nchunks) return;
    const int64_t seg0 = chunk * (GPB * kSegsPerGroup);
    for (int it = 0; it < kSegsPerGroup; ++it) {
        const int64_t s = seg0 + (int64_t)it * GPB + group;
        if (s >= S) break;
        int beg, end;
        if (ptr != nullptr) { beg = ptr[s]; end = ptr[s + 1]; } else { beg = (int)s; end = (int)s + 1; }
        for (int col = lane; col < H; col += LPR) {
            float acc = 0.f;
            for (int i = beg; i < end; ++i) {
                const int r = idx != nullptr ? idx[i] : i;
                const float x = to_f32<T>(in[(size_t)r * H + col]);
                if (HAS_SCALE) acc = fmaf(scale[i], x, acc);
                else acc += x;
            }
            if (mean) acc *= end > beg ? 1.f / (float)(end - beg) : 0.f;
            if (self_in != nullptr) acc = fmaf(self_coef, to_f32<T>(self_in[(size_t)s * H + col]), acc);
            out[(size_t)s * H + col] = from_f32<T>(acc);
        }
    }
}

template <typename T, int LPR, bool VECP>
int launch_lpr(const T* in, const int32_t* idx, const float* scale, const int32_t* ptr, int64_t S, int64_t M, int32_t H, T* out,
               const T* self_in, float self_coef, int32_t mean, hipStream_t st) {
    constexpr int GPB = kBlock / LPR;
    // long lists (>= 16 entries per segment on average: the pre-aggregation of a collapsed dummy relation walks ~n rows per
    // graph) keep the simple kernel; short graph-local lists take the form that is pipelined across segments
    static const int force = dn_knob("DN_GATHER_V1", -1);
    const bool use_v1 = force >= 0 ? force != 0 : (ptr != nullptr && S > 0 && M / S >= 16);
    // few long lists: a workgroup per segment (the chip is empty otherwise: S / (GPB * kSegsPerGroup) workgroups)
    static const int block_max = dn_knob("DN_GATHER_BLOCK", 8192);
    if constexpr (VECP) if (ptr != nullptr && S > 0 && S <= block_max && M / S >= 24) {
        if (scale != nullptr)
            hipLaunchKernelGGL((gather_segsum_block_kernel<T, LPR, true>), dim3((unsigned)S), dim3(kBlock), 0, st, in, idx, scale, ptr,
                               S, H, out, self_in, self_coef, mean);
        else
            hipLaunchKernelGGL((gather_segsum_block_kernel<T, LPR, false>), dim3((unsigned)S), dim3(kBlock), 0, st, in, idx, scale, ptr,
                               S, H, out, self_in, self_coef, mean);
        DN_CHECK_LAUNCH();
        return DN_OK;
    }
    const int64_t nchunks = dn_cdiv(S, (int64_t)GPB * kSegsPerGroup);
    const int64_t grid = dn_cdiv(nchunks, DN_NUM_XCD) * DN_NUM_XCD;
    if (grid > 0x7fffffffLL) { dn_set_error("dn_gather_segsum: grid too large"); return DN_ERR_ARG; }
#define DN_LAUNCH(SC)                                                                                         \
    do {                                                                                                      \
        if (VECP)                                                                                             \
            if (use_v1)                                                                                       \
                hipLaunchKernelGGL((gather_segsum_vec1_kernel<T, LPR, SC>), dim3((unsigned)grid), dim3(kBlock), 0, st, in, \
                                   idx, scale, ptr, S, H, out, self_in, self_coef, mean, nchunks);            \
            else                                                                                              \
                hipLaunchKernelGGL((gather_segsum_vec_kernel<T, LPR, SC>), dim3((unsigned)grid), dim3(kBlock), 0, st, in, \
                                   idx, scale, ptr, S, H, out, self_in, self_coef, mean, nchunks);            \
        else                                                                                                  \
            hipLaunchKernelGGL((gather_segsum_scalar_kernel<T, LPR, SC>), dim3((unsigned)grid), dim3(kBlock), 0, st, \
                               in, idx, scale, ptr, S, H, out, self_in, self_coef, mean, nchunks);            \
    } while (0)
    if (scale != nullptr) DN_LAUNCH(true);
    else DN_LAUNCH(false);
#undef DN_LAUNCH
    DN_CHECK_LAUNCH();
    return DN_OK;
}

template <typename T>
int gather_segsum(const T* in, int64_t in_rows, int32_t H, const int32_t* idx, const float* scale, const int32_t* ptr,
                  int64_t S, int64_t M, T* out, const T* self_in, float self_coef, int32_t mean, hipStream_t st) {
    DN_REQUIRE(H > 0, "dn_gather_segsum: H must be > 0 (got %d)", H);
    DN_REQUIRE(S >= 0 && M >= 0 && in_rows >= 0, "dn_gather_segsum: negative size");
    DN_REQUIRE(S < 0x7fffffffLL && M < 0x7fffffffLL && in_rows < 0x7fffffffLL, "dn_gather_segsum: sizes must fit int32");
    DN_REQUIRE(ptr != nullptr || S == M, "dn_gather_segsum: ptr == NULL requires S == M");
    DN_REQUIRE(idx != nullptr || M <= in_rows, "dn_gather_segsum: idx == NULL requires M <= in_rows");
    if (S == 0) return DN_OK;
    DN_REQUIRE(in != nullptr || M == 0, "dn_gather_segsum: in is NULL");
    DN_REQUIRE(out != nullptr, "dn_gather_segsum: out is NULL");
    constexpr int VN = Vec<T>::N;
    const bool vec = (H % VN == 0) && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out) |
                                        reinterpret_cast<uintptr_t>(self_in)) % 16 == 0);
    const int pieces = vec ? H / VN : H;
    if (vec) {
        if (pieces <= 4) return launch_lpr<T, 4, true>(in, idx, scale, ptr, S, M, H, out, self_in, self_coef, mean, st);
        if (pieces <= 8) return launch_lpr<T, 8, true>(in, idx, scale, ptr, S, M, H, out, self_in, self_coef, mean, st);
        if (pieces <= 16) return launch_lpr<T, 16, true>(in, idx, scale, ptr, S, M, H, out, self_in, self_coef, mean, st);
        if (pieces <= 32) return launch_lpr<T, 32, true>(in, idx, scale, ptr, S, M, H, out, self_in, self_coef, mean, st);
        return launch_lpr<T, 64, true>(in, idx, scale, ptr, S, M, H, out, self_in, self_coef, mean, st);
    }
    if (pieces <= 8) return launch_lpr<T, 8, false>(in, idx, scale, ptr, S, M, H, out, self_in, self_coef, mean, st);
    return launch_lpr<T, 64, false>(in, idx, scale, ptr, S, M, H, out, self_in, self_coef, mean, st);
}

// ---------------------------------------------------------------------------------------------
// segment max (+argmax) over contiguous rows and its backward
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void segment_max_kernel(const T* __restrict__ in, int32_t H,
                                                             const int32_t* __restrict__ ptr, int64_t S,
                                                             T* __restrict__ out, int32_t* __restrict__ argmax) {
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= S * H) return;
    const int64_t s = t / H;
    const int h = (int)(t % H);
    const int beg = ptr[s], end = ptr[s + 1];
    float best = 0.f;
    int arg = -1;
    for (int i = beg; i < end; ++i) {
        const float x = to_f32<T>(in[(size_t)i * H + h]);
        if (arg < 0 || x > best) { best = x; arg = i; }
    }
    out[t] = from_f32<T>(best);
    argmax[t] = arg;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void segment_max_bwd_kernel(const T* __restrict__ gout,
                                                                 const int32_t* __restrict__ argmax, int32_t H,
                                                                 const int32_t* __restrict__ ptr, int64_t S,
                                                                 T* __restrict__ gin) {
    // one thread per (segment, column): walk the segment's rows, writing g or 0
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= S * H) return;
    const int64_t s = t / H;
    const int h = (int)(t % H);
    const int beg = ptr[s], end = ptr[s + 1];
    const int arg = argmax[t];
    const T g = gout[t];
    for (int i = beg; i < end; ++i) gin[(size_t)i * H + h] = (i == arg) ? g : from_f32<T>(0.f);
}

template <typename T>
int segment_max(const T* in, int32_t H, const int32_t* ptr, int64_t S, T* out, int32_t* argmax, hipStream_t st) {
    DN_REQUIRE(H > 0 && S >= 0, "dn_segment_max: bad sizes");
    if (S == 0) return DN_OK;
    DN_REQUIRE(ptr && out && argmax, "dn_segment_max: NULL pointer");
    const int64_t grid = dn_cdiv(S * H, kBlock);
    DN_REQUIRE(grid <= 0x7fffffffLL, "dn_segment_max: grid too large");
    hipLaunchKernelGGL((segment_max_kernel<T>), dim3((unsigned)grid), dim3(kBlock), 0, st, in, H, ptr, S, out, argmax);
    DN_CHECK_LAUNCH();
    return DN_OK;
}

template <typename T>
int segment_max_bwd(const T* gout, const int32_t* argmax, int32_t H, const int32_t* ptr, int64_t S, T* gin,
                    hipStream_t st) {
    DN_REQUIRE(H > 0 && S >= 0, "dn_segment_max_bwd: bad sizes");
    if (S == 0) return DN_OK;
    DN_REQUIRE(gout && argmax && ptr && gin, "dn_segment_max_bwd: NULL pointer");
    const int64_t grid = dn_cdiv(S * H, kBlock);
    DN_REQUIRE(grid <= 0x7fffffffLL, "dn_segment_max_bwd: grid too large");
    hipLaunchKernelGGL((segment_max_bwd_kernel<T>), dim3((unsigned)grid), dim3(kBlock), 0, st, gout, argmax, H, ptr, S,
                       gin);
    DN_CHECK_LAUNCH();
    return DN_OK;
}

// ---------------------------------------------------------------------------------------------
// Edge dot product (SDDMM):  out[e] = < a[ia[e], :], b[ib[e], :] >    (gradient of a per-edge scalar weight)
// Gather + segment MAX over gathered rows and its backward (SAGEConv aggr='max').
// One lane group per edge / segment, lanes stride the feature dimension; fp32 math; no atomics.
// ---------------------------------------------------------------------------------------------
template <typename T, int G>
__global__ __launch_bounds__(kBlock) void edge_dot_kernel(const T* __restrict__ a, const int32_t* __restrict__ ia,
                                                          const T* __restrict__ b, const int32_t* __restrict__ ib, int32_t H,
                                                          int64_t E, float* __restrict__ out) {
    const int lane = threadIdx.x % G;
    const int64_t e = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / G;
    if (e >= E) return;                                   // whole groups leave together (E is checked per group)
    const T* ra = a + (size_t)(ia ? ia[e] : e) * H;
    const T* rb = b + (size_t)(ib ? ib[e] : e) * H;
    float acc = 0.f;
    for (int h = lane; h < H; h += G) acc = fmaf(to_f32<T>(ra[h]), to_f32<T>(rb[h]), acc);
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, G);
    if (lane == 0) out[e] = acc;
}

template <typename T, int G>
__global__ __launch_bounds__(kBlock) void gather_segmax_kernel(const T* __restrict__ in, const int32_t* __restrict__ idx,
                                                               const int32_t* __restrict__ ptr, int64_t S, int32_t H,
                                                               T* __restrict__ out, int32_t* __restrict__ argmax) {
    const int lane = threadIdx.x % G;
    const int64_t s = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / G;
    if (s >= S) return;
    const int beg = ptr[s], end = ptr[s + 1];
    for (int h = lane; h < H; h += G) {
        float best = 0.f;
        int arg = -1;
        for (int i = beg; i < end; ++i) {
            const float x = to_f32<T>(in[(size_t)idx[i] * H + h]);
            if (arg < 0 || x > best) { best = x; arg = i; }   // ties keep the first (lowest slot)
        }
        out[(size_t)s * H + h] = from_f32<T>(best);
        argmax[(size_t)s * H + h] = arg;
    }
}

// grad_in[u, h] = sum over the slots i that gathered row u of (argmax[seg(i), h] == i ? gout[seg(i), h] : 0)
template <typename T, int G>
__global__ __launch_bounds__(kBlock) void gather_segmax_bwd_kernel(const T* __restrict__ gout, const int32_t* __restrict__ argmax,
                                                                   const int32_t* __restrict__ tptr,
                                                                   const int32_t* __restrict__ tslot,
                                                                   const int32_t* __restrict__ seg_of_slot, int64_t rows,
                                                                   int32_t H, T* __restrict__ gin) {
    const int lane = threadIdx.x % G;
    const int64_t u = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / G;
    if (u >= rows) return;
    const int beg = tptr[u], end = tptr[u + 1];
    for (int h = lane; h < H; h += G) {
        float acc = 0.f;
        for (int k = beg; k < end; ++k) {
            const int i = tslot[k], s = seg_of_slot[i];
            if (argmax[(size_t)s * H + h] == i) acc += to_f32<T>(gout[(size_t)s * H + h]);
        }
        gin[(size_t)u * H + h] = from_f32<T>(acc);
    }
}

// fp32 rows of H = 64 / 128 / 256 (16-byte aligned): a lane group of H / 4 lanes per edge / segment / row, 16 bytes per lane and
// entry, several entries' loads in flight (round 6: the lane-strided forms above take 4 bytes per lane behind one dependent index
// load per entry -- GraphSAGE's max aggregation on 20 k nodes spent 0.27 ms in ONE backward launch, 0.4 TB/s).  Same results: the
// first maximum in slot order wins; the backward adds in slot order.
template <int LPR>
__global__ __launch_bounds__(kBlock) void edge_dot_v4_kernel(const float* __restrict__ a, const int32_t* __restrict__ ia,
                                                             const float* __restrict__ b, const int32_t* __restrict__ ib, int64_t E,
                                                             float* __restrict__ out) {
    constexpr int H = 4 * LPR;
    const int lane = threadIdx.x % LPR;
    const int64_t e = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / LPR;
    if (e >= E) return;
    const float4 va = *reinterpret_cast<const float4*>(a + (size_t)(ia ? ia[e] : e) * H + lane * 4);
    const float4 vb = *reinterpret_cast<const float4*>(b + (size_t)(ib ? ib[e] : e) * H + lane * 4);
    float acc = fmaf(va.x, vb.x, 0.f);
    acc = fmaf(va.y, vb.y, acc); acc = fmaf(va.z, vb.z, acc); acc = fmaf(va.w, vb.w, acc);
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, LPR);
    if (lane == 0) out[e] = acc;
}

template <int LPR>
__global__ __launch_bounds__(kBlock) void gather_segmax_v4_kernel(const float* __restrict__ in, const int32_t* __restrict__ idx,
                                                                  const int32_t* __restrict__ ptr, int64_t S, float* __restrict__ out,
                                                                  int32_t* __restrict__ argmax) {
    constexpr int H = 4 * LPR, KU = 4;
    const int lane = threadIdx.x % LPR;
    const int64_t s = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / LPR;
    if (s >= S) return;
    const int beg = ptr[s], end = ptr[s + 1];
    float best[4] = {0.f, 0.f, 0.f, 0.f};
    int arg[4] = {-1, -1, -1, -1};
    for (int i0 = beg; i0 < end; i0 += KU) {
        float4 v[KU];
#pragma unroll
        for (int k = 0; k < KU; ++k) {
            const int i = min(i0 + k, end - 1);                               // (clamped: unconditional loads, masked below)
            v[k] = *reinterpret_cast<const float4*>(in + (size_t)(idx ? idx[i] : i) * H + lane * 4);
        }
#pragma unroll
        for (int k = 0; k < KU; ++k) {
            if (i0 + k >= end) break;
            const float x[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (arg[c] < 0 || x[c] > best[c]) { best[c] = x[c]; arg[c] = i0 + k; }   // ties keep the first (lowest slot)
        }
    }
    *reinterpret_cast<float4*>(out + (size_t)s * H + lane * 4) = make_float4(best[0], best[1], best[2], best[3]);
    *reinterpret_cast<int4*>(argmax + (size_t)s * H + lane * 4) = make_int4(arg[0], arg[1], arg[2], arg[3]);
}

template <int LPR>
__global__ __launch_bounds__(kBlock) void gather_segmax_bwd_v4_kernel(const float* __restrict__ gout, const int32_t* __restrict__ argmax,
                                                                      const int32_t* __restrict__ tptr, const int32_t* __restrict__ tslot,
                                                                      const int32_t* __restrict__ seg_of_slot, int64_t rows,
                                                                      float* __restrict__ gin) {
    constexpr int H = 4 * LPR, KU = 4;
    const int lane = threadIdx.x % LPR;
    const int64_t u = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / LPR;
    if (u >= rows) return;
    const int beg = tptr[u], end = tptr[u + 1];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = beg; k0 < end; k0 += KU) {
        int slot[KU];
        int4 am[KU];
        float4 g[KU];
#pragma unroll
        for (int k = 0; k < KU; ++k) {
            slot[k] = tslot[min(k0 + k, end - 1)];
            const size_t off = (size_t)seg_of_slot[slot[k]] * H + lane * 4;
            am[k] = *reinterpret_cast<const int4*>(argmax + off);
            g[k] = *reinterpret_cast<const float4*>(gout + off);
        }
#pragma unroll
        for (int k = 0; k < KU; ++k) {
            if (k0 + k >= end) break;
            acc[0] += am[k].x == slot[k] ? g[k].x : 0.f;
            acc[1] += am[k].y == slot[k] ? g[k].y : 0.f;
            acc[2] += am[k].z == slot[k] ? g[k].z : 0.f;
            acc[3] += am[k].w == slot[k] ? g[k].w : 0.f;
        }
    }
    *reinterpret_cast<float4*>(gin + (size_t)u * H + lane * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

template <typename T>
constexpr bool dn_is_f32() { return sizeof(T) == 4; }
__host__ inline bool dn_v4_ok(int32_t H, const void* p0, const void* p1, const void* p2) {
    return (H == 64 || H == 128 || H == 256) &&
           ((reinterpret_cast<uintptr_t>(p0) | reinterpret_cast<uintptr_t>(p1) | reinterpret_cast<uintptr_t>(p2)) % 16 == 0);
}
#define DN_V4_DISPATCH(KERNEL, COUNT, ...)                                                                                     \
    do {                                                                                                                       \
        if (H == 64) hipLaunchKernelGGL((KERNEL<16>), dim3((unsigned)dn_cdiv((COUNT) * 16, kBlock)), dim3(kBlock), 0, st, __VA_ARGS__);       \
        else if (H == 128) hipLaunchKernelGGL((KERNEL<32>), dim3((unsigned)dn_cdiv((COUNT) * 32, kBlock)), dim3(kBlock), 0, st, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<64>), dim3((unsigned)dn_cdiv((COUNT) * 64, kBlock)), dim3(kBlock), 0, st, __VA_ARGS__);               \
    } while (0)

template <typename T>
int edge_dot(const T* a, const int32_t* ia, const T* b, const int32_t* ib, int32_t H, int64_t E, float* out, hipStream_t st) {
    DN_REQUIRE(H > 0 && E >= 0, "dn_edge_dot: bad sizes");
    if (E == 0) return DN_OK;
    DN_REQUIRE(a && b && out, "dn_edge_dot: NULL pointer");
    if constexpr (dn_is_f32<T>()) {
        if (dn_v4_ok(H, a, b, nullptr)) {
            DN_V4_DISPATCH(edge_dot_v4_kernel, E, (const float*)a, ia, (const float*)b, ib, E, out);
            DN_CHECK_LAUNCH();
            return DN_OK;
        }
    }
    if (H <= 16) {
        hipLaunchKernelGGL((edge_dot_kernel<T, 16>), dim3((unsigned)dn_cdiv(E * 16, kBlock)), dim3(kBlock), 0, st, a, ia, b, ib, H, E, out);
    } else {
        hipLaunchKernelGGL((edge_dot_kernel<T, 64>), dim3((unsigned)dn_cdiv(E * 64, kBlock)), dim3(kBlock), 0, st, a, ia, b, ib, H, E, out);
    }
    DN_CHECK_LAUNCH();
    return DN_OK;
}

template <typename T>
int gather_segmax(const T* in, const int32_t* idx, const int32_t* ptr, int64_t S, int32_t H, T* out, int32_t* argmax,
                  hipStream_t st) {
    DN_REQUIRE(H > 0 && S >= 0, "dn_gather_segmax: bad sizes");
    if (S == 0) return DN_OK;
    DN_REQUIRE(ptr && out && argmax, "dn_gather_segmax: NULL pointer");
    if constexpr (dn_is_f32<T>()) {
        if (dn_v4_ok(H, in, out, argmax)) {
            DN_V4_DISPATCH(gather_segmax_v4_kernel, S, (const float*)in, idx, ptr, S, (float*)out, argmax);
            DN_CHECK_LAUNCH();
            return DN_OK;
        }
    }
    if (H <= 16) {
        hipLaunchKernelGGL((gather_segmax_kernel<T, 16>), dim3((unsigned)dn_cdiv(S * 16, kBlock)), dim3(kBlock), 0, st, in, idx, ptr, S, H, out, argmax);
    } else {
        hipLaunchKernelGGL((gather_segmax_kernel<T, 64>), dim3((unsigned)dn_cdiv(S * 64, kBlock)), dim3(kBlock), 0, st, in, idx, ptr, S, H, out, argmax);
    }
    DN_CHECK_LAUNCH();
    return DN_OK;
}

template <typename T>
int gather_segmax_bwd(const T* gout, const int32_t* argmax, const int32_t* tptr, const int32_t* tslot,
                      const int32_t* seg_of_slot, int64_t rows, int32_t H, T* gin, hipStream_t st) {
    DN_REQUIRE(H > 0 && rows >= 0, "dn_gather_segmax_bwd: bad sizes");
    if (rows == 0) return DN_OK;
    DN_REQUIRE(tptr && gin, "dn_gather_segmax_bwd: NULL pointer");
    if constexpr (dn_is_f32<T>()) {
        if (dn_v4_ok(H, gout, argmax, gin)) {
            DN_V4_DISPATCH(gather_segmax_bwd_v4_kernel, rows, (const float*)gout, argmax, tptr, tslot, seg_of_slot, rows, (float*)gin);
            DN_CHECK_LAUNCH();
            return DN_OK;
        }
    }
    if (H <= 16) {
        hipLaunchKernelGGL((gather_segmax_bwd_kernel<T, 16>), dim3((unsigned)dn_cdiv(rows * 16, kBlock)), dim3(kBlock), 0, st, gout, argmax, tptr, tslot, seg_of_slot, rows, H, gin);
    } else {
        hipLaunchKernelGGL((gather_segmax_bwd_kernel<T, 64>), dim3((unsigned)dn_cdiv(rows * 64, kBlock)), dim3(kBlock), 0, st, gout, argmax, tptr, tslot, seg_of_slot, rows, H, gin);
    }
    DN_CHECK_LAUNCH();
    return DN_OK;
}


// ---------------------------------------------------------------------------------------------
// dn_overflow_rows_add_bf16: out[v] += sum of the rows of node v's list that did not fit its slots (dn_slot_table_build_i32
// leaves -2 in the last slot of such a node and 1 in its byte of `overflow`).  Runs after dn_rows_selfsum_bf16; a block screens 256 bytes,
// collects the few flagged ones in LDS and finishes each with a lane group (16 bytes per lane), fp32 sum, one rounding.
// ---------------------------------------------------------------------------------------------
template <int LPR>
__global__ __launch_bounds__(kBlock) void overflow_rows_add_kernel(const bf16_t* __restrict__ S, const uint8_t* __restrict__ over,
                                                                   int32_t K, int64_t N, const int32_t* __restrict__ lptr,
                                                                   const int32_t* __restrict__ lrows, int32_t num_edge_rows,
                                                                   int32_t drop_beg, int32_t drop_end, bf16_t* __restrict__ out) {
    constexpr int H = LPR * 8;
    constexpr int kMaxList = 64;                                  // list entries a lane group takes in one sweep (longer lists: loop)
    __shared__ int32_t found[kBlock];
    __shared__ int32_t ent[kBlock / LPR][kMaxList];               // per lane group: the EXTRA rows of the node it is finishing
    __shared__ int32_t nfound;
    if (threadIdx.x == 0) nfound = 0;
    __syncthreads();
    const int64_t v0 = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (v0 < N && over[v0] != 0) found[atomicAdd(&nfound, 1)] = (int32_t)v0;
    __syncthreads();
    const int cnt = nfound;
    if (cnt == 0) return;
    // one WAVE per flagged node: its 64 lanes read 64 list entries at once (one round trip), a ballot ranks the kept ones, the
    // extras go to LDS; then the wave's lane groups (LPR lanes = one row) take the extras round-robin, all row loads in flight
    // together, and the groups' fp32 sums meet in LDS -- four dependent round trips per node instead of two per list entry
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int GPW = 64 / LPR;                                 // lane groups per wave
    const int grp = lane / LPR, pc = lane % LPR;
    __shared__ float part[kBlock / 64][GPW][H];
    for (int q = wave; q < cnt; q += kBlock / 64) {
        const int32_t v = found[q];
        const int beg = lptr[v], end = lptr[v + 1];
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int kept_before = 0;
        for (int base = beg; base < end; base += 64) {
            const int i = base + lane;
            const int r = i < end ? lrows[i] : -1;
            const bool keep = i < end && r < num_edge_rows && !(r >= drop_beg && r < drop_end);     // the table builder's filter
            const unsigned long long m = __ballot(keep);
            const int rank = kept_before + __popcll(m & ((1ull << lane) - 1ull));
            const int first_extra = K - 1;                         // ranks below sit in the slots
            const int n_extra_before = max(kept_before - first_extra, 0);
            if (keep && rank >= first_extra) ent[wave][rank - first_extra - n_extra_before] = r;
            const int kept_here = __popcll(m);
            const int n_extra = max(kept_before + kept_here - first_extra, 0) - n_extra_before;
            __builtin_amdgcn_wave_barrier();
            for (int e = grp; e < n_extra; e += GPW) {
                const int rr = ent[wave][e];
                const uint4 x4 = *reinterpret_cast<const uint4*>(S + (size_t)rr * H + pc * 8);
                const uint32_t w[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a[2 * j] += __uint_as_float(w[j] << 16);
                    a[2 * j + 1] += __uint_as_float(w[j] & 0xffff0000u);
                }
            }
            kept_before += kept_here;
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) part[wave][grp][pc * 8 + j] = a[j];
        __builtin_amdgcn_wave_barrier();
        if (grp == 0) {                                           // fixed order over the groups: deterministic
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                t[j] = 0.f;
                for (int g2 = 0; g2 < GPW; ++g2) t[j] += part[wave][g2][pc * 8 + j];
            }
            uint4* po = reinterpret_cast<uint4*>(out + (size_t)v * H + pc * 8);
            const uint4 o = *po;
            const uint32_t w[4] = {o.x, o.y, o.z, o.w};
            uint32_t res[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16_t lo = (bf16_t)(t[2 * j] + __uint_as_float(w[j] << 16));
                const bf16_t hi = (bf16_t)(t[2 * j + 1] + __uint_as_float(w[j] & 0xffff0000u));
                res[j] = (uint32_t)__builtin_bit_cast(uint16_t, lo) | ((uint32_t)__builtin_bit_cast(uint16_t, hi) << 16);
            }
            *po = make_uint4(res[0], res[1], res[2], res[3]);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

extern "C" {

int dn_gather_segsum_f32(const float* in, int64_t in_rows, int32_t H, const int32_t* idx, const float* scale,
                         const int32_t* ptr, int64_t S, int64_t M, float* out, const float* self_in, float self_coef,
                         int32_t mean, dn_stream_t stream) {
    return gather_segsum<float>(in, in_rows, H, idx, scale, ptr, S, M, out, self_in, self_coef, mean, (hipStream_t)stream);
}
int dn_gather_segsum_bf16(const void* in, int64_t in_rows, int32_t H, const int32_t* idx, const float* scale,
                          const int32_t* ptr, int64_t S, int64_t M, void* out, const void* self_in, float self_coef,
                          int32_t mean, dn_stream_t stream) {
    return gather_segsum<bf16_t>((const bf16_t*)in, in_rows, H, idx, scale, ptr, S, M, (bf16_t*)out,
                                 (const bf16_t*)self_in, self_coef, mean, (hipStream_t)stream);
}

int dn_overflow_rows_add_bf16(const void* S, int32_t H, const uint8_t* overflow, int32_t num_slots, int64_t N, const int32_t* list_ptr,
                              const int32_t* list_rows, int32_t num_edge_rows, int32_t drop_beg, int32_t drop_end, void* out,
                              dn_stream_t stream) {
    DN_REQUIRE(H == 64 || H == 128 || H == 256, "dn_overflow_rows_add: unsupported width %d (64/128/256 only)", H);
    DN_REQUIRE(N >= 0 && N < 0x7fffffffLL && num_slots >= 2 && num_edge_rows >= 0, "dn_overflow_rows_add: bad sizes");
    if (N == 0) return DN_OK;
    DN_REQUIRE(overflow && list_ptr && list_rows && out && (S || num_edge_rows == 0), "dn_overflow_rows_add: NULL pointer");
    DN_REQUIRE((reinterpret_cast<uintptr_t>(S) | reinterpret_cast<uintptr_t>(out)) % 16 == 0, "dn_overflow_rows_add: unaligned pointer");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)dn_cdiv(N, kBlock)), block(kBlock);
    const bf16_t* s = (const bf16_t*)S;
    bf16_t* o = (bf16_t*)out;
    if (H == 256) hipLaunchKernelGGL((overflow_rows_add_kernel<32>), grid, block, 0, st, s, overflow, num_slots, N, list_ptr, list_rows, num_edge_rows, drop_beg, drop_end, o);
    else if (H == 128) hipLaunchKernelGGL((overflow_rows_add_kernel<16>), grid, block, 0, st, s, overflow, num_slots, N, list_ptr, list_rows, num_edge_rows, drop_beg, drop_end, o);
    else hipLaunchKernelGGL((overflow_rows_add_kernel<8>), grid, block, 0, st, s, overflow, num_slots, N, list_ptr, list_rows, num_edge_rows, drop_beg, drop_end, o);
    DN_CHECK_LAUNCH();
    return DN_OK;
}

static int64_t dn_rows_unknown() { return 0x7ffffffeLL; }

int dn_segment_sum_f32(const float* in, int32_t H, const int32_t* ptr, int64_t S, float* out, dn_stream_t stream) {
    DN_REQUIRE(ptr != nullptr || S == 0, "dn_segment_sum: ptr is NULL");
    return gather_segsum<float>(in, dn_rows_unknown(), H, nullptr, nullptr, ptr, S, 0, out, nullptr, 0.f, 0,
                                (hipStream_t)stream);
}
int dn_segment_sum_bf16(const void* in, int32_t H, const int32_t* ptr, int64_t S, void* out, dn_stream_t stream) {
    DN_REQUIRE(ptr != nullptr || S == 0, "dn_segment_sum: ptr is NULL");
    return gather_segsum<bf16_t>((const bf16_t*)in, dn_rows_unknown(), H, nullptr, nullptr, ptr, S, 0, (bf16_t*)out,
                                 nullptr, 0.f, 0, (hipStream_t)stream);
}
int dn_segment_mean_f32(const float* in, int32_t H, const int32_t* ptr, int64_t S, float* out, dn_stream_t stream) {
    DN_REQUIRE(ptr != nullptr || S == 0, "dn_segment_mean: ptr is NULL");
    return gather_segsum<float>(in, dn_rows_unknown(), H, nullptr, nullptr, ptr, S, 0, out, nullptr, 0.f, 1,
                                (hipStream_t)stream);
}
int dn_segment_mean_bf16(const void* in, int32_t H, const int32_t* ptr, int64_t S, void* out, dn_stream_t stream) {
    DN_REQUIRE(ptr != nullptr || S == 0, "dn_segment_mean: ptr is NULL");
    return gather_segsum<bf16_t>((const bf16_t*)in, dn_rows_unknown(), H, nullptr, nullptr, ptr, S, 0, (bf16_t*)out,
                                 nullptr, 0.f, 1, (hipStream_t)stream);
}
int dn_segment_max_f32(const float* in, int32_t H, const int32_t* ptr, int64_t S, float* out, int32_t* argmax,
                       dn_stream_t stream) {
    return segment_max<float>(in, H, ptr, S, out, argmax, (hipStream_t)stream);
}
int dn_segment_max_bf16(const void* in, int32_t H, const int32_t* ptr, int64_t S, void* out, int32_t* argmax,
                        dn_stream_t stream) {
    return segment_max<bf16_t>((const bf16_t*)in, H, ptr, S, (bf16_t*)out, argmax, (hipStream_t)stream);
}
int dn_segment_max_bwd_f32(const float* grad_out, const int32_t* argmax, int32_t H, const int32_t* ptr, int64_t S,
                           float* grad_in, dn_stream_t stream) {
    return segment_max_bwd<float>(grad_out, argmax, H, ptr, S, grad_in, (hipStream_t)stream);
}
int dn_segment_max_bwd_bf16(const void* grad_out, const int32_t* argmax, int32_t H, const int32_t* ptr, int64_t S,
                            void* grad_in, dn_stream_t stream) {
    return segment_max_bwd<bf16_t>((const bf16_t*)grad_out, argmax, H, ptr, S, (bf16_t*)grad_in, (hipStream_t)stream);
}

int dn_edge_dot_f32(const float* a, const int32_t* ia, const float* b, const int32_t* ib, int32_t H, int64_t E, float* out,
                    dn_stream_t stream) {
    return edge_dot<float>(a, ia, b, ib, H, E, out, (hipStream_t)stream);
}
int dn_edge_dot_bf16(const void* a, const int32_t* ia, const void* b, const int32_t* ib, int32_t H, int64_t E, float* out,
                     dn_stream_t stream) {
    return edge_dot<bf16_t>((const bf16_t*)a, ia, (const bf16_t*)b, ib, H, E, out, (hipStream_t)stream);
}
int dn_gather_segmax_f32(const float* in, const int32_t* idx, const int32_t* ptr, int64_t S, int32_t H, float* out,
                         int32_t* argmax, dn_stream_t stream) {
    DN_REQUIRE(idx != nullptr || S == 0, "dn_gather_segmax: idx is NULL");
    return gather_segmax<float>(in, idx, ptr, S, H, out, argmax, (hipStream_t)stream);
}
int dn_gather_segmax_bf16(const void* in, const int32_t* idx, const int32_t* ptr, int64_t S, int32_t H, void* out,
                          int32_t* argmax, dn_stream_t stream) {
    DN_REQUIRE(idx != nullptr || S == 0, "dn_gather_segmax: idx is NULL");
    return gather_segmax<bf16_t>((const bf16_t*)in, idx, ptr, S, H, (bf16_t*)out, argmax, (hipStream_t)stream);
}
int dn_gather_segmax_bwd_f32(const float* grad_out, const int32_t* argmax, const int32_t* tptr, const int32_t* tslot,
                             const int32_t* seg_of_slot, int64_t rows, int32_t H, float* grad_in, dn_stream_t stream) {
    return gather_segmax_bwd<float>(grad_out, argmax, tptr, tslot, seg_of_slot, rows, H, grad_in, (hipStream_t)stream);
}
int dn_gather_segmax_bwd_bf16(const void* grad_out, const int32_t* argmax, const int32_t* tptr, const int32_t* tslot,
                              const int32_t* seg_of_slot, int64_t rows, int32_t H, void* grad_in, dn_stream_t stream) {
    return gather_segmax_bwd<bf16_t>((const bf16_t*)grad_out, argmax, tptr, tslot, seg_of_slot, rows, H, (bf16_t*)grad_in,
                                     (hipStream_t)stream);
}

}  // extern "C"
