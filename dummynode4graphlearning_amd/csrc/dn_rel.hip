// Relation-wise dense kernels of the RGCN/RGIN layers on the matrix cores (gfx950 MFMA, bf16 in / fp32 acc).
//
// dn_rows_wgrad_bf16:  gW[r] = sum_{p in relation r} A[ia[p], :]^T  G[ig[p], :]      ([Hi x Ho] per relation)
//   the weight gradient of  Y[p] = A[p] W[rel(p)]  (rows p are relation-major: rel_ptr).  Reduction runs over the
//   ROW index of both operands ("TN" GEMM with a huge K and a 256-wide output), which library GEMMs serve with a
//   16-workgroup launch; here K is split into row chunks, one workgroup per chunk keeps the whole Hi x Ho tile in
//   its accumulators, and a second kernel adds the chunk partials in a FIXED order (deterministic, no atomics).
//
// LDS image: row-major [32 rows][H + 8] bf16 tiles (16-byte row pad); both MFMA operands are K-strided in that
// image, so fragments are fetched with ds_read_b64_tr_b16 (hardware transpose read, cdna_hip_programming.md T10).
#include "dn_common.h"
#include "../../include/dn_hip.h"

namespace {

typedef __bf16 bf16_t;
typedef short short4v __attribute__((ext_vector_type(4)));
typedef short short8v __attribute__((ext_vector_type(8)));
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWgThreads = 512;  // 8 waves: 2 (rows of the output tile) x 4 (columns)
constexpr int kTileRows = 32;    // K-step of one MFMA 16x16x32
constexpr int kPad = 8;          // bf16 elements of row padding in LDS

struct Chunk {
    int32_t rel, beg, end, pad;
};

__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* tile, int stride, int col0, int lane) {
    // fragment of a K-strided operand: element j of lane l = tile[8*(l>>4) + j][col0 + (l&15)]
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const bf16_t* a0 = tile + (8 * g + q) * stride + col0 + 4 * p;
    typedef short4v __attribute__((address_space(3))) * lds_p;
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a0));
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a0 + 4 * stride));
    const short8v f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, f);
}

// partial[chunk][k][n] = sum_{p in chunk} A[ia[p]][k] * G[ig[p]][n]
template <int HI, int HO>
__global__ __launch_bounds__(kWgThreads) void rows_wgrad_kernel(const bf16_t* __restrict__ A,
                                                                const int32_t* __restrict__ ia,
                                                                const bf16_t* __restrict__ G,
                                                                const int32_t* __restrict__ ig,
                                                                const Chunk* __restrict__ chunks,
                                                                float* __restrict__ partial) {
    constexpr int SA = HI + kPad, SG = HO + kPad;
    constexpr int MT = HI / 2 / 16, NT = HO / 4 / 16;           // 16x16 tiles per wave
    constexpr int NPA = kTileRows * HI / 8, NPG = kTileRows * HO / 8;   // 16-byte pieces per tile
    constexpr int PA = (NPA + kWgThreads - 1) / kWgThreads;             // pieces per thread per tile (A)
    constexpr int PG = (NPG + kWgThreads - 1) / kWgThreads;
    static_assert(MT >= 1 && NT >= 1, "unsupported width");
    __shared__ __attribute__((aligned(16))) bf16_t lds[2 * kTileRows * (SA + SG)];
    auto bufA = [&](int b) -> bf16_t* { return lds + b * (kTileRows * SA); };
    auto bufG = [&](int b) -> bf16_t* { return lds + 2 * kTileRows * SA + b * (kTileRows * SG); };

    const Chunk ch = chunks[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;                     // wave position in the output tile
    const int k0 = wm * (HI / 2), n0 = wn * (HO / 4);
    const int ntiles = (ch.end - ch.beg + kTileRows - 1) / kTileRows;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint4 ra[PA], rg[PG];
    auto load_tile = [&](int t) {
        const int row0 = ch.beg + t * kTileRows;
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            const int piece = tid + j * kWgThreads, r = piece / (HI / 8), c = piece % (HI / 8);
            const int p = row0 + r;
            ra[j] = make_uint4(0, 0, 0, 0);
            if (piece < NPA && p < ch.end) {
                const size_t src = ia ? (size_t)ia[p] : (size_t)p;
                ra[j] = *reinterpret_cast<const uint4*>(A + src * HI + c * 8);
            }
        }
#pragma unroll
        for (int j = 0; j < PG; ++j) {
            const int piece = tid + j * kWgThreads, r = piece / (HO / 8), c = piece % (HO / 8);
            const int p = row0 + r;
            rg[j] = make_uint4(0, 0, 0, 0);
            if (piece < NPG && p < ch.end) {
                const size_t src = ig ? (size_t)ig[p] : (size_t)p;
                rg[j] = *reinterpret_cast<const uint4*>(G + src * HO + c * 8);
            }
        }
    };
    auto store_tile = [&](int b) {
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            const int piece = tid + j * kWgThreads, r = piece / (HI / 8), c = piece % (HI / 8);
            if (piece < NPA) *reinterpret_cast<uint4*>(bufA(b) + r * SA + c * 8) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < PG; ++j) {
            const int piece = tid + j * kWgThreads, r = piece / (HO / 8), c = piece % (HO / 8);
            if (piece < NPG) *reinterpret_cast<uint4*>(bufG(b) + r * SG + c * 8) = rg[j];
        }
    };

    if (ntiles > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int b = t & 1;
        if (t + 1 < ntiles) load_tile(t + 1);                    // global loads in flight under the MFMAs
        bf16x8 fb[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) fb[n] = tr_frag(bufG(b), SG, n0 + n * 16, lane);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const bf16x8 fa = tr_frag(bufA(b), SA, k0 + m * 16, lane);
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[n], acc[m][n], 0, 0, 0);
        }
        if (t + 1 < ntiles) store_tile(b ^ 1);
        __syncthreads();
    }
    // C layout of mfma 16x16: col = lane & 15, row = (lane >> 4) * 4 + i
    float* out = partial + (size_t)blockIdx.x * HI * HO;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = k0 + m * 16 + (lane >> 4) * 4 + i, c = n0 + n * 16 + (lane & 15);
                out[(size_t)k * HO + c] = acc[m][n][i];
            }
}

// out[r] = sum of the partials of relation r's chunks, in chunk order
template <typename TO>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial,
                                                           const int32_t* __restrict__ chunk_ptr, int64_t tile_elems,
                                                           TO* __restrict__ out) {
    const int r = blockIdx.y;
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= tile_elems) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = chunk_ptr[r]; c < chunk_ptr[r + 1]; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(partial + (size_t)c * tile_elems + i);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    TO* o = out + (size_t)r * tile_elems + i;
    o[0] = (TO)s.x; o[1] = (TO)s.y; o[2] = (TO)s.z; o[3] = (TO)s.w;
}

template <int HI, int HO>
int launch_wgrad(const bf16_t* A, const int32_t* ia, const bf16_t* G, const int32_t* ig, const Chunk* chunks,
                 int64_t num_chunks, float* partial, hipStream_t st) {
    hipLaunchKernelGGL((rows_wgrad_kernel<HI, HO>), dim3((unsigned)num_chunks), dim3(kWgThreads), 0, st, A, ia, G, ig, chunks,
                       partial);
    DN_CHECK_LAUNCH();
    return DN_OK;
}

}  // namespace

extern "C" {

size_t dn_rows_wgrad_workspace_bytes(int64_t num_chunks, int32_t Hi, int32_t Ho) {
    if (num_chunks < 0 || Hi <= 0 || Ho <= 0) { dn_set_error("dn_rows_wgrad_workspace_bytes: bad sizes"); return 0; }
    return (size_t)(num_chunks > 0 ? num_chunks : 1) * Hi * Ho * sizeof(float);
}

int dn_rows_wgrad_bf16(const void* A, const int32_t* idx_a, const void* G, const int32_t* idx_g, int32_t Hi, int32_t Ho,
                       int64_t R, const int32_t* chunks, int64_t num_chunks, const int32_t* chunk_ptr, void* out,
                       int32_t out_is_f32, void* workspace, size_t workspace_bytes, dn_stream_t stream) {
    DN_REQUIRE(R >= 0 && num_chunks >= 0, "dn_rows_wgrad: negative size");
    DN_REQUIRE(Hi == Ho && (Hi == 64 || Hi == 128 || Hi == 256), "dn_rows_wgrad: unsupported widths %d x %d "
               "(square 64/128/256 only)", Hi, Ho);
    if (R == 0) return DN_OK;
    DN_REQUIRE(out && chunk_ptr, "dn_rows_wgrad: NULL pointer");
    DN_REQUIRE(num_chunks == 0 || (A && G && chunks && workspace), "dn_rows_wgrad: NULL pointer");
    DN_REQUIRE(workspace_bytes >= (size_t)num_chunks * Hi * Ho * sizeof(float), "dn_rows_wgrad: workspace too small");
    DN_REQUIRE((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(G)) % 16 == 0, "dn_rows_wgrad: unaligned input");
    hipStream_t st = (hipStream_t)stream;
    const Chunk* ch = reinterpret_cast<const Chunk*>(chunks);
    int rc = DN_OK;
    if (num_chunks > 0) {
        if (Hi == 256) rc = launch_wgrad<256, 256>((const bf16_t*)A, idx_a, (const bf16_t*)G, idx_g, ch, num_chunks, (float*)workspace, st);
        else if (Hi == 128) rc = launch_wgrad<128, 128>((const bf16_t*)A, idx_a, (const bf16_t*)G, idx_g, ch, num_chunks, (float*)workspace, st);
        else rc = launch_wgrad<64, 64>((const bf16_t*)A, idx_a, (const bf16_t*)G, idx_g, ch, num_chunks, (float*)workspace, st);
        if (rc != DN_OK) return rc;
    }
    const int64_t tile = (int64_t)Hi * Ho;
    dim3 grid((unsigned)dn_cdiv(tile, 1024), (unsigned)R);
    if (out_is_f32)
        hipLaunchKernelGGL((wgrad_reduce_kernel<float>), grid, dim3(256), 0, st, (const float*)workspace, chunk_ptr, tile, (float*)out);
    else
        hipLaunchKernelGGL((wgrad_reduce_kernel<bf16_t>), grid, dim3(256), 0, st, (const float*)workspace, chunk_ptr, tile, (bf16_t*)out);
    DN_CHECK_LAUNCH();
    return DN_OK;
}

}  // extern "C"
