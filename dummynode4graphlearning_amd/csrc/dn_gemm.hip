// Relation-grouped dense products for ANY layer widths (the widths the matrix-core kernels of dn_rel.hip do not cover: the
// first conv of a GC model maps F node features to H hidden units, F = 5 ... 38; hidden sizes such as 32 or 100):
//
//   dn_rows_gemm_*        Y[p, :]  = A[p, :] @ W[rel(p)]          (W [R, K, N], or [R, N, K] read transposed)
//   dn_rows_wgrad_any_*   gW[r]    = sum_{p in relation r} A[p, :]^T G[p, :]     ([K, N] per relation)
//
// rows p are relation-major (tile / chunk tables of dn_row_tables_build_i32; pieces never cross relations).  These replace the
// per-relation loop PyG's RGCNConv runs (`for i in range(num_relations): ... h @ weight[i]`, call sites rgconv.py:17-18,96)
// -- R small library GEMMs + R launches per conv -- by one launch per product, whatever R is.
// Plain fp32 FMA tiles through LDS (64 x 64 outputs per 256-thread workgroup, 4 x 4 per thread, K in steps of 16): at these
// widths the products are a few MFLOP per thousand rows and the launch count, not the arithmetic, is what the loop cost.
// Deterministic: split-K partials are added in chunk order.
#include "dn_common.h"
#include "../../include/dn_hip.h"

namespace {

typedef __bf16 bf16_t;
struct Piece { int32_t rel, beg, end, pad; };

constexpr int TM = 64, TN = 64, TK = 16, kThreads = 256;

template <typename T> __device__ __forceinline__ float ld(const T* p) { return (float)*p; }
template <typename T> __device__ __forceinline__ T cvt(float v) { return (T)v; }

// Y[tile rows, n0 : n0 + 64] = A[tile rows, :] @ Wr   with Wr[k][n] = transposed ? W[rel][n][k] : W[rel][k][n]
template <typename T>
__global__ __launch_bounds__(kThreads) void rows_gemm_kernel(const T* __restrict__ A, const T* __restrict__ W,
                                                             const T* __restrict__ bias, int32_t K, int32_t N,
                                                             int32_t transposed, const Piece* __restrict__ tiles,
                                                             T* __restrict__ Y) {
    __shared__ float As[TK][TM + 1];
    __shared__ float Ws[TK][TN + 1];
    const Piece tl = tiles[blockIdx.x];
    const int rows = tl.end - tl.beg;
    if (rows <= 0) return;
    const int n0 = blockIdx.y * TN;
    const int tid = threadIdx.x, tx = tid % 16, ty = tid / 16;          // thread owns rows ty*4.., cols tx*4..
    const T* Wr = W + (size_t)tl.rel * K * N;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += TK) {
        for (int i = tid; i < TM * TK; i += kThreads) {                  // A tile: TM rows x TK
            const int r = i / TK, k = i % TK;
            As[k][r] = (r < rows && k0 + k < K) ? ld(A + (size_t)(tl.beg + r) * K + k0 + k) : 0.f;
        }
        for (int i = tid; i < TK * TN; i += kThreads) {                  // W tile: TK x TN
            const int k = transposed ? i % TK : i / TN, n = transposed ? i / TK : i % TN;
            float v = 0.f;
            if (k0 + k < K && n0 + n < N) v = transposed ? ld(Wr + (size_t)(n0 + n) * K + k0 + k) : ld(Wr + (size_t)(k0 + k) * N + n0 + n);
            Ws[k][n] = v;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < TK; ++k) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = As[k][ty * 4 + i]; b[i] = Ws[k][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = ty * 4 + i;
        if (r >= rows) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n < N) Y[(size_t)(tl.beg + r) * N + n] = cvt<T>(acc[i][j] + (bias ? ld(bias + (size_t)tl.rel * N + n) : 0.f));
        }
    }
}

// The fp32 product on the matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate -- bit for bit an fmaf
// chain, so the GC models' 1e-4 parity through BatchNorm is untouched), any K and N: the first conv of a GC model (F = 5 .. 38
// node features -> H hidden units), hidden sizes outside {64, 128, 256}, the [H, 2] classifier heads.  One 256-thread workgroup
// per 64 x 64 output tile: wave w owns rows 16 w .. 16 w + 15 of the tile and all 64 columns (4 accumulators); the TRANSPOSED
// product is formed (A operand = 16 columns of the weight, B operand = the wave's 16 rows), so a lane ends up with 4 consecutive
// columns of one row = one 16-byte store.  Operands are staged through LDS in K-steps of 32 (zero-padded at the edges).
typedef float f32x4_t __attribute__((ext_vector_type(4)));
// KC: columns of k per step.  A step is a dependent chain (loads -> LDS -> barrier -> a few MFMAs -> barrier), so its cost is the
// loads' latency whatever KC is: K >= 128 takes steps of 128 (a 256-wide readout Linear on 16 k rows: eight round trips -> two).
template <int KC>
__global__ __launch_bounds__(kThreads) void rows_gemm_mfma_f32_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                                      const float* __restrict__ bias, int32_t K, int32_t N,
                                                                      int32_t transposed, const Piece* __restrict__ tiles,
                                                                      float* __restrict__ Y) {
    __shared__ float As[TM][KC + 1];                                     // [row][k]
    __shared__ float Ws[KC][TN + 1];                                     // [k][n]
    const Piece tl = tiles[blockIdx.x];
    const int rows = tl.end - tl.beg;
    if (rows <= 0) return;
    const int n0 = blockIdx.y * TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const float* Wr = W + (size_t)tl.rel * K * N;
    f32x4_t acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += KC) {
        for (int i = tid; i < TM * KC; i += kThreads) {                  // A tile: consecutive threads along k (contiguous in memory)
            const int r = i / KC, k = i % KC;
            As[r][k] = (r < rows && k0 + k < K) ? A[(size_t)(tl.beg + r) * K + k0 + k] : 0.f;
        }
        if (transposed) {                                                // W[rel][n][k]: consecutive threads along k
            for (int i = tid; i < KC * TN; i += kThreads) {
                const int n = i / KC, k = i % KC;
                Ws[k][n] = (k0 + k < K && n0 + n < N) ? Wr[(size_t)(n0 + n) * K + k0 + k] : 0.f;
            }
        } else {                                                         // W[rel][k][n]: consecutive threads along n
            for (int i = tid; i < KC * TN; i += kThreads) {
                const int k = i / TN, n = i % TN;
                Ws[k][n] = (k0 + k < K && n0 + n < N) ? Wr[(size_t)(k0 + k) * N + n0 + n] : 0.f;
            }
        }
        __syncthreads();
        const int kend = (K - k0 < KC ? K - k0 : KC);
        for (int kk = 0; kk < kend; kk += 4) {                           // (a step past K multiplies the zero padding)
            const float b = As[16 * wave + j][kk + g];                   // B operand: row 16 w + j of the tile, k = kk + g
#pragma unroll
            for (int n = 0; n < 4; ++n)
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ws[kk + g][16 * n + j], b, acc[n], 0, 0, 0);
        }
        __syncthreads();
    }
    // lane (j, g): row 16 w + j of the tile, columns n0 + 16 n + 4 g + (0 .. 3) for n = 0 .. 3
    const int r = 16 * wave + j;
    if (r >= rows) return;
    float* yrow = Y + (size_t)(tl.beg + r) * N;
    const float* brow = bias ? bias + (size_t)tl.rel * N : nullptr;
    const bool vec = (N % 4 == 0) && (reinterpret_cast<uintptr_t>(Y) % 16 == 0);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int c = n0 + 16 * n + 4 * g;
        if (c >= N) continue;
        float v[4] = {acc[n][0], acc[n][1], acc[n][2], acc[n][3]};
        if (brow) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (c + i < N) v[i] += brow[c + i];
        }
        if (vec) *reinterpret_cast<float4*>(yrow + c) = make_float4(v[0], v[1], v[2], v[3]);   // (N % 4 == 0: c + 3 < N)
        else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (c + i < N) yrow[c + i] = v[i];
        }
    }
}

// partial[chunk][k0 : k0 + 64, n0 : n0 + 64] = sum_{p in chunk} A[p, k]^T G[p, n]
template <typename T>
__global__ __launch_bounds__(kThreads) void rows_wgrad_any_kernel(const T* __restrict__ A, const T* __restrict__ G, int32_t K,
                                                                  int32_t N, const Piece* __restrict__ chunks,
                                                                  float* __restrict__ partial, float* __restrict__ cs_partial) {
    __shared__ float As[TK][TM + 1];                                     // [row in step][k]
    __shared__ float Gs[TK][TN + 1];                                     // [row in step][n]
    const Piece ch = chunks[blockIdx.x];
    const int k0 = blockIdx.y * TM, n0 = blockIdx.z * TN;
    const int tid = threadIdx.x, tx = tid % 16, ty = tid / 16;
    float acc[4][4] = {};
    float cs[4] = {};                                                    // column sums of A (n-tile 0, threads tx == 0 only)
    const bool do_cs = cs_partial != nullptr && blockIdx.z == 0 && tx == 0;
    for (int p0 = ch.beg; p0 < ch.end; p0 += TK) {
        for (int i = tid; i < TK * TM; i += kThreads) {
            const int r = i / TM, k = i % TM;
            As[r][k] = (p0 + r < ch.end && k0 + k < K) ? ld(A + (size_t)(p0 + r) * K + k0 + k) : 0.f;
        }
        for (int i = tid; i < TK * TN; i += kThreads) {
            const int r = i / TN, n = i % TN;
            Gs[r][n] = (p0 + r < ch.end && n0 + n < N) ? ld(G + (size_t)(p0 + r) * N + n0 + n) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < TK; ++r) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = As[r][ty * 4 + i]; b[i] = Gs[r][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
            if (do_cs) {
#pragma unroll
                for (int i = 0; i < 4; ++i) cs[i] += a[i];
            }
        }
        __syncthreads();
    }
    if (do_cs) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (k0 + ty * 4 + i < K) cs_partial[(size_t)blockIdx.x * K + k0 + ty * 4 + i] = cs[i];
    }
    float* out = partial + (size_t)blockIdx.x * K * N;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + ty * 4 + i;
        if (k >= K) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n < N) out[(size_t)k * N + n] = acc[i][j];
        }
    }
}

// out[r] = sum over relation r's chunks of partial[chunk], chunk order (fixed association)
template <typename T>
__global__ __launch_bounds__(256) void wgrad_any_reduce_kernel(const float* __restrict__ partial, const int32_t* __restrict__ chunk_ptr,
                                                               int64_t elems, T* __restrict__ out, const float* __restrict__ cs_partial,
                                                               int32_t K, float* __restrict__ cs_out) {
    const int r = blockIdx.y;
    const int64_t tile_blocks = (elems + 255) / 256;
    if ((int64_t)blockIdx.x >= tile_blocks) {                            // blocks past the tile: the column sums, same order
        const int64_t k = ((int64_t)blockIdx.x - tile_blocks) * 256 + threadIdx.x;
        if (k >= K) return;
        float s = 0.f;
        int c = chunk_ptr[r];
        const int ce = chunk_ptr[r + 1];
        for (; c + 8 <= ce; c += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = cs_partial[(size_t)(c + j) * K + k];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        for (; c < ce; ++c) s += cs_partial[(size_t)c * K + k];
        cs_out[(size_t)r * K + k] = s;
        return;
    }
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= elems) return;
    // eight loads in flight, added in chunk order (the same association as one at a time: a thread's chain of dependent
    // round trips was 21 us per launch at config 4, for a 38 x 256 gradient)
    float s = 0.f;
    int c = chunk_ptr[r];
    const int ce = chunk_ptr[r + 1];
    for (; c + 8 <= ce; c += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = partial[(size_t)(c + k) * elems + i];
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[k];
    }
    for (; c < ce; ++c) s += partial[(size_t)c * elems + i];
    out[(size_t)r * elems + i] = cvt<T>(s);
}

template <typename T>
int rows_gemm(const T* A, const T* W, const T* bias, int32_t K, int32_t N, int32_t transposed, const int32_t* tiles, int64_t num_tiles,
              T* Y, hipStream_t st) {
    DN_REQUIRE(K >= 1 && N >= 1 && num_tiles >= 0 && num_tiles < 0x7fffffffLL, "dn_rows_gemm: bad sizes");
    if (num_tiles == 0) return DN_OK;
    DN_REQUIRE(A && W && tiles && Y, "dn_rows_gemm: NULL pointer");
    if constexpr (sizeof(T) == 4) {                                       // fp32: exact-f32 MFMA tiles
        static const int mfma = dn_knob("DN_GEMM_MFMA", 1);               // tuning build: 0 keeps the FMA tiles
        if (mfma) {
            if (K >= 128)
                hipLaunchKernelGGL(rows_gemm_mfma_f32_kernel<128>, dim3((unsigned)num_tiles, (unsigned)dn_cdiv(N, TN)), dim3(kThreads), 0, st,
                                   (const float*)A, (const float*)W, (const float*)bias, K, N, transposed,
                                   reinterpret_cast<const Piece*>(tiles), (float*)Y);
            else
                hipLaunchKernelGGL(rows_gemm_mfma_f32_kernel<32>, dim3((unsigned)num_tiles, (unsigned)dn_cdiv(N, TN)), dim3(kThreads), 0, st,
                                   (const float*)A, (const float*)W, (const float*)bias, K, N, transposed,
                                   reinterpret_cast<const Piece*>(tiles), (float*)Y);
            DN_CHECK_LAUNCH();
            return DN_OK;
        }
    }
    hipLaunchKernelGGL((rows_gemm_kernel<T>), dim3((unsigned)num_tiles, (unsigned)dn_cdiv(N, TN)), dim3(kThreads), 0, st, A, W, bias, K,
                       N, transposed, reinterpret_cast<const Piece*>(tiles), Y);
    DN_CHECK_LAUNCH();
    return DN_OK;
}

template <typename T>
int rows_wgrad_any(const T* A, const T* G, int32_t K, int32_t N, int64_t R, const int32_t* chunks, int64_t num_chunks,
                   const int32_t* chunk_ptr, T* out, float* colsum_out, void* workspace, size_t workspace_bytes, hipStream_t st) {
    DN_REQUIRE(K >= 1 && N >= 1 && R >= 0 && num_chunks >= 0 && num_chunks < 65536LL * 32768, "dn_rows_wgrad_any: bad sizes");
    if (R == 0) return DN_OK;
    DN_REQUIRE(out && chunk_ptr, "dn_rows_wgrad_any: NULL pointer");
    DN_REQUIRE(num_chunks == 0 || (A && G && chunks && workspace), "dn_rows_wgrad_any: NULL pointer");
    DN_REQUIRE(workspace_bytes >= (size_t)num_chunks * K * (N + 1) * sizeof(float), "dn_rows_wgrad_any: workspace too small");
    float* cs_partial = colsum_out ? (float*)workspace + (size_t)num_chunks * K * N : nullptr;
    DN_REQUIRE(dn_cdiv(K, TM) <= 65535 && dn_cdiv(N, TN) <= 65535, "dn_rows_wgrad_any: layer too wide");
    if (num_chunks > 0) {
        hipLaunchKernelGGL((rows_wgrad_any_kernel<T>), dim3((unsigned)num_chunks, (unsigned)dn_cdiv(K, TM), (unsigned)dn_cdiv(N, TN)),
                           dim3(kThreads), 0, st, A, G, K, N, reinterpret_cast<const Piece*>(chunks), (float*)workspace, cs_partial);
        DN_CHECK_LAUNCH();
    }
    const int64_t elems = (int64_t)K * N;
    const int64_t cs_blocks = colsum_out ? dn_cdiv(K, 256) : 0;
    hipLaunchKernelGGL((wgrad_any_reduce_kernel<T>), dim3((unsigned)(dn_cdiv(elems, 256) + cs_blocks), (unsigned)R), dim3(256), 0, st,
                       (const float*)workspace, chunk_ptr, elems, out, (const float*)cs_partial, K, colsum_out);
    DN_CHECK_LAUNCH();
    return DN_OK;
}

}  // namespace

extern "C" {

int dn_rows_gemm_f32(const float* A, const float* W, const float* bias, int32_t K, int32_t N, int32_t transposed, const int32_t* tiles,
                     int64_t num_tiles, float* Y, dn_stream_t stream) {
    return rows_gemm<float>(A, W, bias, K, N, transposed, tiles, num_tiles, Y, (hipStream_t)stream);
}
int dn_rows_gemm_bf16(const void* A, const void* W, const void* bias, int32_t K, int32_t N, int32_t transposed, const int32_t* tiles,
                      int64_t num_tiles, void* Y, dn_stream_t stream) {
    return rows_gemm<bf16_t>((const bf16_t*)A, (const bf16_t*)W, (const bf16_t*)bias, K, N, transposed, tiles, num_tiles, (bf16_t*)Y,
                             (hipStream_t)stream);
}
size_t dn_rows_wgrad_any_workspace_bytes(int64_t num_chunks, int32_t K, int32_t N) {
    if (num_chunks < 0 || K <= 0 || N <= 0) { dn_set_error("dn_rows_wgrad_any_workspace_bytes: bad sizes"); return 0; }
    return (size_t)(num_chunks > 0 ? num_chunks : 1) * (size_t)K * (N + 1) * sizeof(float);        // products + column sums
}
int dn_rows_wgrad_any_f32(const float* A, const float* G, int32_t K, int32_t N, int64_t R, const int32_t* chunks, int64_t num_chunks,
                          const int32_t* chunk_ptr, float* out, float* colsum_out, void* workspace, size_t workspace_bytes,
                          dn_stream_t stream) {
    return rows_wgrad_any<float>(A, G, K, N, R, chunks, num_chunks, chunk_ptr, out, colsum_out, workspace, workspace_bytes,
                                 (hipStream_t)stream);
}
int dn_rows_wgrad_any_bf16(const void* A, const void* G, int32_t K, int32_t N, int64_t R, const int32_t* chunks, int64_t num_chunks,
                           const int32_t* chunk_ptr, void* out, float* colsum_out, void* workspace, size_t workspace_bytes,
                           dn_stream_t stream) {
    return rows_wgrad_any<bf16_t>((const bf16_t*)A, (const bf16_t*)G, K, N, R, chunks, num_chunks, chunk_ptr, (bf16_t*)out, colsum_out,
                                  workspace, workspace_bytes, (hipStream_t)stream);
}

}  // extern "C"
