// BatchNorm over the ROWS of a [N, C] node-feature matrix, training mode (batch statistics): the `BatchNorm1d` inside the MLPs of
// the GC models (graph_classification/graph_neural_networks/models/gconv.py:187-194, rgconv.py:85-93), applied to every node of
// the batch right after the aggregation.  torch's generic kernels spend 35 us on the statistics of a 20 k x 128 matrix (and as
// much again in the backward reduction): about as long as the whole gather they follow.  Here both reductions are one streaming
// pass each over row chunks (coalesced 16-byte pieces, fp32 partial sums per chunk) + a tiny fixed-order combine; the
// element-wise halves are plain streaming kernels.  Deterministic (no atomics).
//
//   forward   mean[c], var[c] (biased) over rows;  y = (x - mean) * rstd * weight + bias,  rstd = 1 / sqrt(var + eps)
//   backward  s1[c] = sum dy,  s2[c] = sum dy * xhat;   dx = weight * rstd * (dy - s1 / N - xhat * s2 / N);  dweight = s2, dbias = s1
// Sums of squares are taken about a per-column shift (row 0) so that a large mean does not cancel the variance away.
#include "dn_common.h"
#include "../../include/dn_hip.h"

namespace {

typedef __bf16 bf16_t;
constexpr int kBlock = 256;
constexpr int kChunkRows = 64;

template <typename T> struct V4;
template <> struct V4<float> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct V4<bf16_t> {
    static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[4]) {
        const uint2 t = *reinterpret_cast<const uint2*>(p);
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
        v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[4]) {
        typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));
        bf16x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (bf16_t)v[i];
        *reinterpret_cast<bf16x4*>(p) = o;
    }
};

// One workgroup per chunk of kChunkRows rows: partial[chunk][0][c] = sum_r f(r, c), partial[chunk][1][c] = sum_r g(r, c).
//   MODE 0 (forward statistics):  f = x - shift,  g = (x - shift)^2        (shift[c] = x[0][c])
//   MODE 1 (backward reduction):  f = dy,         g = dy * (x - mean) * rstd      (dy masked by y > 0 when a ReLU is fused)
template <typename T, int MODE>
__global__ __launch_bounds__(kBlock) void colreduce_kernel(const T* __restrict__ X, const T* __restrict__ DY,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ w, const float* __restrict__ bias, int32_t relu,
                                                           int64_t N, int32_t C, float* __restrict__ partial) {
    extern __shared__ float red[];                     // [groups][2][C]
    const int LPR = C / 4;                             // lanes per row (C % 4 == 0, C <= 1024)
    const int GPB = kBlock / LPR;                      // rows in flight per block
    const int lane = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    const int64_t r0 = (int64_t)blockIdx.x * kChunkRows;
    const int c = lane * 4;
    float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f}, s[4], m[4], rs[4], gw[4], gb[4];
    if (grp < GPB) {
        if (MODE == 0) V4<T>::load(X + c, s);
        else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { m[i] = mean[c + i]; rs[i] = rstd[c + i]; gw[i] = w ? w[c + i] : 1.f; gb[i] = bias ? bias[c + i] : 0.f; }
        }
        for (int64_t r = r0 + grp; r < r0 + kChunkRows && r < N; r += GPB) {
            float x[4];
            V4<T>::load(X + (size_t)r * C + c, x);
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float d = x[i] - s[i]; a[i] += d; b[i] = fmaf(d, d, b[i]); }
            } else {
                float dy[4];
                V4<T>::load(DY + (size_t)r * C + c, dy);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float xh = (x[i] - m[i]) * rs[i];
                    const float keep = (relu && !(fmaf(xh, gw[i], gb[i]) > 0.f)) ? 0.f : 1.f;       // fused ReLU: dy where y > 0
                    const float d = dy[i] * keep;
                    a[i] += d;
                    b[i] = fmaf(d, xh, b[i]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { red[(grp * 2 + 0) * C + c + i] = a[i]; red[(grp * 2 + 1) * C + c + i] = b[i]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += kBlock) {
        float t = 0.f;
        for (int g = 0; g < GPB; ++g) t += red[g * 2 * C + i];           // fixed order
        partial[(size_t)blockIdx.x * 2 * C + i] = t;
    }
}

// combine the chunk partials: 8 threads share a column (thread `slice` adds chunks slice, slice + 8, ... with 4 loads in flight),
// the 8 slice sums are folded in slice order -- a fixed association, so the result is reproducible; a single thread walking a few
// hundred partials is a 40 us latency chain.  MODE 0: mean / var / rstd;  MODE 1: s1, s2 as they are
template <int MODE, typename T>
__global__ __launch_bounds__(kBlock) void colfinal_kernel(const float* __restrict__ partial, int64_t nchunks, int64_t N, int32_t C,
                                                          const T* __restrict__ X, float eps, float* __restrict__ out0,
                                                          float* __restrict__ out1, float* __restrict__ out2,
                                                          float* __restrict__ run_mean, float* __restrict__ run_var,
                                                          float momentum, long long* __restrict__ batches_tracked) {
    constexpr int SL = 8, CPB = kBlock / SL;                            // 32 columns per block
    __shared__ float r1[SL][CPB], r2[SL][CPB];
    const int cl = threadIdx.x % CPB, slice = threadIdx.x / CPB;
    const int c = blockIdx.x * CPB + cl;
    float s1 = 0.f, s2 = 0.f;
    if (c < C) {
        int64_t k = slice;
        for (; k + 3 * SL < nchunks; k += 4 * SL) {
            const float a0 = partial[(size_t)k * 2 * C + c], a1 = partial[(size_t)(k + SL) * 2 * C + c];
            const float a2 = partial[(size_t)(k + 2 * SL) * 2 * C + c], a3 = partial[(size_t)(k + 3 * SL) * 2 * C + c];
            const float b0 = partial[(size_t)k * 2 * C + C + c], b1 = partial[(size_t)(k + SL) * 2 * C + C + c];
            const float b2 = partial[(size_t)(k + 2 * SL) * 2 * C + C + c], b3 = partial[(size_t)(k + 3 * SL) * 2 * C + C + c];
            s1 += a0; s1 += a1; s1 += a2; s1 += a3;
            s2 += b0; s2 += b1; s2 += b2; s2 += b3;
        }
        for (; k < nchunks; k += SL) {
            s1 += partial[(size_t)k * 2 * C + c];
            s2 += partial[(size_t)k * 2 * C + C + c];
        }
    }
    r1[slice][cl] = s1;
    r2[slice][cl] = s2;
    __syncthreads();
    if (slice != 0 || c >= C) return;
    s1 = r1[0][cl];
    s2 = r2[0][cl];
#pragma unroll
    for (int j = 1; j < SL; ++j) { s1 += r1[j][cl]; s2 += r2[j][cl]; }
    if (MODE == 0) {
        const float inv = 1.f / (float)N;
        const float d = s1 * inv;                                       // mean - shift
        const float var = fmaxf(s2 * inv - d * d, 0.f);
        const float mean = (float)X[c] + d;
        out0[c] = mean;
        out1[c] = var;
        out2[c] = rsqrtf(var + eps);
        if (batches_tracked && c == 0) *batches_tracked += 1;          // (BatchNorm's num_batches_tracked buffer: no launch of its own)
        if (run_mean) {                                                 // torch's update: r = (1 - m) r + m * new, unbiased variance
            const float unb = var * ((float)N / (float)(N > 1 ? N - 1 : 1));
            run_mean[c] = run_mean[c] * (1.f - momentum) + momentum * mean;
            run_var[c] = run_var[c] * (1.f - momentum) + momentum * unb;
        }
    } else {
        out0[c] = s1;
        out1[c] = s2;
    }
}

// MODE 0: y = relu?((x - mean) * rstd * w + b);   MODE 1: dx = w * rstd * (dy' - s1 / N - xhat * s2 / N), dy' = dy masked by y > 0
template <typename T, int MODE>
__global__ __launch_bounds__(kBlock) void colapply_kernel(const T* __restrict__ X, const T* __restrict__ DY,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ w, const float* __restrict__ b,
                                                          const float* __restrict__ s1, const float* __restrict__ s2, int32_t relu,
                                                          int64_t N, int32_t C, T* __restrict__ out, int32_t cvec) {
    const int64_t total = N * (C / 4);
    const float invN = 1.f / (float)N;
    // a thread's column group is the same in every trip of the loop when the stride is a multiple of C / 4 (C / 4 divides 256 for
    // C = 64 .. 1024 in powers of two): its per-column constants are loaded once, not per element (ten loads per 16-byte piece)
    const bool fixed = (kBlock % (C / 4)) == 0;
    float cm[4], cr[4], cg[4], cb[4], c1[4], c2[4];
    // (cvec: every per-column array is 16-byte aligned -- one load per array instead of four; at a 16 k-row batch the kernel was
    //  bound by its count of vector-memory instructions, 27 per 16-byte piece: 26 us for 49 MB)
    auto consts = [&](int c) {
        if (cvec) {
            const float4 m4 = *reinterpret_cast<const float4*>(mean + c), r4 = *reinterpret_cast<const float4*>(rstd + c);
            const float4 g4 = w ? *reinterpret_cast<const float4*>(w + c) : float4{1.f, 1.f, 1.f, 1.f};
            const float4 b4 = b ? *reinterpret_cast<const float4*>(b + c) : float4{0.f, 0.f, 0.f, 0.f};
            cm[0] = m4.x; cm[1] = m4.y; cm[2] = m4.z; cm[3] = m4.w;
            cr[0] = r4.x; cr[1] = r4.y; cr[2] = r4.z; cr[3] = r4.w;
            cg[0] = g4.x; cg[1] = g4.y; cg[2] = g4.z; cg[3] = g4.w;
            cb[0] = b4.x; cb[1] = b4.y; cb[2] = b4.z; cb[3] = b4.w;
            if (MODE == 1) {
                const float4 p4 = *reinterpret_cast<const float4*>(s1 + c), q4 = *reinterpret_cast<const float4*>(s2 + c);
                c1[0] = p4.x * invN; c1[1] = p4.y * invN; c1[2] = p4.z * invN; c1[3] = p4.w * invN;
                c2[0] = q4.x * invN; c2[1] = q4.y * invN; c2[2] = q4.z * invN; c2[3] = q4.w * invN;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) c1[k] = c2[k] = 0.f;
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            cm[k] = mean[c + k]; cr[k] = rstd[c + k]; cg[k] = w ? w[c + k] : 1.f; cb[k] = b ? b[c + k] : 0.f;
            c1[k] = MODE == 1 ? s1[c + k] * invN : 0.f; c2[k] = MODE == 1 ? s2[c + k] * invN : 0.f;
        }
    };
    if (fixed) consts((int)(((int64_t)blockIdx.x * kBlock + threadIdx.x) % (C / 4)) * 4);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        if (!fixed) consts((int)(i % (C / 4)) * 4);
        float x[4], o[4];
        V4<T>::load(X + i * 4, x);
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                o[k] = fmaf((x[k] - cm[k]) * cr[k], cg[k], cb[k]);
                if (relu) o[k] = fmaxf(o[k], 0.f);
            }
        } else {
            float dy[4];
            V4<T>::load(DY + i * 4, dy);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xh = (x[k] - cm[k]) * cr[k];
                // (as a product: hipcc 7.0 turns `cond ? 0.f : dy[k]` here into "dy[k] = 0; if (!cond) {}" -- the gradient vanished)
                const float keep = (relu && !(fmaf(xh, cg[k], cb[k]) > 0.f)) ? 0.f : 1.f;
                const float d = dy[k] * keep;
                o[k] = cg[k] * cr[k] * (d - c1[k] - xh * c2[k]);
            }
        }
        V4<T>::store(out + i * 4, o);
    }
}

template <typename T>
int bn_forward(const T* X, int64_t N, int32_t C, const float* w, const float* b, float eps, T* Y, float* mean, float* var, float* rstd,
               float* run_mean, float* run_var, float momentum, int32_t relu, long long* batches_tracked, float* ws, size_t ws_bytes,
               hipStream_t st) {
    DN_REQUIRE((run_mean == nullptr) == (run_var == nullptr), "dn_batchnorm_rows: running_mean and running_var come together");
    DN_REQUIRE(N >= 1 && C >= 4 && C % 4 == 0 && C <= 1024, "dn_batchnorm_rows: need N >= 1 and C a multiple of 4 in [4, 1024] (got %lld x %d)",
               (long long)N, C);
    DN_REQUIRE(X && Y && mean && var && rstd && ws, "dn_batchnorm_rows: NULL pointer");
    DN_REQUIRE((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y)) % 16 == 0, "dn_batchnorm_rows: unaligned pointer");
    const int64_t nchunks = dn_cdiv(N, kChunkRows);
    DN_REQUIRE(ws_bytes >= (size_t)nchunks * 2 * C * sizeof(float), "dn_batchnorm_rows: workspace too small");
    const int GPB = kBlock / (C / 4);
    DN_REQUIRE(GPB >= 1, "dn_batchnorm_rows: C too large");
    hipLaunchKernelGGL((colreduce_kernel<T, 0>), dim3((unsigned)nchunks), dim3(kBlock), (size_t)GPB * 2 * C * sizeof(float), st, X,
                       (const T*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, N, C, ws);
    hipLaunchKernelGGL((colfinal_kernel<0, T>), dim3((unsigned)dn_cdiv(C, kBlock / 8)), dim3(kBlock), 0, st, (const float*)ws, nchunks, N, C, X,
                       eps, mean, var, rstd, run_mean, run_var, momentum, batches_tracked);
    const int64_t blocks = dn_cdiv(N * (C / 4), kBlock);
    const int32_t cvec = ((reinterpret_cast<uintptr_t>(mean) | reinterpret_cast<uintptr_t>(rstd) | reinterpret_cast<uintptr_t>(w) |
                           reinterpret_cast<uintptr_t>(b)) % 16) == 0;
    // (at most 2,048 workgroups: a thread then takes several pieces of the same columns and loads their constants once)
    hipLaunchKernelGGL((colapply_kernel<T, 0>), dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(kBlock), 0, st, X, (const T*)nullptr,
                       (const float*)mean, (const float*)rstd, w, b, (const float*)nullptr, (const float*)nullptr, relu, N, C, Y, cvec);
    DN_CHECK_LAUNCH();
    return DN_OK;
}

template <typename T>
int bn_backward(const T* DY, const T* X, int64_t N, int32_t C, const float* mean, const float* rstd, const float* w, const float* b,
                int32_t relu, T* DX, float* s1, float* s2, float* ws, size_t ws_bytes, hipStream_t st) {
    DN_REQUIRE(N >= 1 && C >= 4 && C % 4 == 0 && C <= 1024, "dn_batchnorm_rows_bwd: bad sizes");
    DN_REQUIRE(DY && X && mean && rstd && DX && s1 && s2 && ws, "dn_batchnorm_rows_bwd: NULL pointer");
    DN_REQUIRE((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(DY) | reinterpret_cast<uintptr_t>(DX)) % 16 == 0,
               "dn_batchnorm_rows_bwd: unaligned pointer");
    const int64_t nchunks = dn_cdiv(N, kChunkRows);
    DN_REQUIRE(ws_bytes >= (size_t)nchunks * 2 * C * sizeof(float), "dn_batchnorm_rows_bwd: workspace too small");
    const int GPB = kBlock / (C / 4);
    hipLaunchKernelGGL((colreduce_kernel<T, 1>), dim3((unsigned)nchunks), dim3(kBlock), (size_t)GPB * 2 * C * sizeof(float), st, X, DY, mean,
                       rstd, w, b, relu, N, C, ws);
    hipLaunchKernelGGL((colfinal_kernel<1, T>), dim3((unsigned)dn_cdiv(C, kBlock / 8)), dim3(kBlock), 0, st, (const float*)ws, nchunks, N, C, X,
                       0.f, s1, s2, (float*)nullptr, (float*)nullptr, (float*)nullptr, 0.f, (long long*)nullptr);
    const int64_t blocks = dn_cdiv(N * (C / 4), kBlock);
    const int32_t cvec = ((reinterpret_cast<uintptr_t>(mean) | reinterpret_cast<uintptr_t>(rstd) | reinterpret_cast<uintptr_t>(w) |
                           reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(s1) | reinterpret_cast<uintptr_t>(s2)) % 16) == 0;
    hipLaunchKernelGGL((colapply_kernel<T, 1>), dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(kBlock), 0, st, X, DY, mean, rstd, w,
                       b, (const float*)s1, (const float*)s2, relu, N, C, DX, cvec);
    DN_CHECK_LAUNCH();
    return DN_OK;
}

}  // namespace

extern "C" {

size_t dn_batchnorm_rows_workspace_bytes(int64_t N, int32_t C) {
    if (N < 0 || C <= 0) { dn_set_error("dn_batchnorm_rows_workspace_bytes: bad sizes"); return 0; }
    return (size_t)(N > 0 ? dn_cdiv(N, kChunkRows) : 1) * 2 * (size_t)C * sizeof(float);
}
int dn_batchnorm_rows_f32(const float* X, int64_t N, int32_t C, const float* weight, const float* bias, float eps, float* Y, float* mean,
                          float* var, float* rstd, float* running_mean, float* running_var, float momentum, int32_t relu,
                          int64_t* num_batches_tracked, void* workspace, size_t workspace_bytes, dn_stream_t stream) {
    return bn_forward<float>(X, N, C, weight, bias, eps, Y, mean, var, rstd, running_mean, running_var, momentum, relu,
                             (long long*)num_batches_tracked, (float*)workspace, workspace_bytes, (hipStream_t)stream);
}
int dn_batchnorm_rows_bf16(const void* X, int64_t N, int32_t C, const float* weight, const float* bias, float eps, void* Y, float* mean,
                           float* var, float* rstd, float* running_mean, float* running_var, float momentum, int32_t relu,
                           int64_t* num_batches_tracked, void* workspace, size_t workspace_bytes, dn_stream_t stream) {
    return bn_forward<bf16_t>((const bf16_t*)X, N, C, weight, bias, eps, (bf16_t*)Y, mean, var, rstd, running_mean, running_var, momentum,
                              relu, (long long*)num_batches_tracked, (float*)workspace, workspace_bytes, (hipStream_t)stream);
}
int dn_batchnorm_rows_bwd_f32(const float* DY, const float* X, int64_t N, int32_t C, const float* mean, const float* rstd,
                              const float* weight, const float* bias, int32_t relu, float* DX, float* sum_dy, float* sum_dy_xhat,
                              void* workspace, size_t workspace_bytes, dn_stream_t stream) {
    return bn_backward<float>(DY, X, N, C, mean, rstd, weight, bias, relu, DX, sum_dy, sum_dy_xhat, (float*)workspace, workspace_bytes,
                              (hipStream_t)stream);
}
int dn_batchnorm_rows_bwd_bf16(const void* DY, const void* X, int64_t N, int32_t C, const float* mean, const float* rstd,
                               const float* weight, const float* bias, int32_t relu, void* DX, float* sum_dy, float* sum_dy_xhat,
                               void* workspace, size_t workspace_bytes, dn_stream_t stream) {
    return bn_backward<bf16_t>((const bf16_t*)DY, (const bf16_t*)X, N, C, mean, rstd, weight, bias, relu, (bf16_t*)DX, sum_dy,
                               sum_dy_xhat, (float*)workspace, workspace_bytes, (hipStream_t)stream);
}

}  // extern "C"
