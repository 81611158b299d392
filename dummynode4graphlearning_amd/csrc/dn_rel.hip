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
#include <cstdlib>

#include "dn_common.h"
#include "dn_internal.h"
#include "../../include/dn_hip.h"

namespace {

typedef __bf16 bf16_t;
typedef short short4v __attribute__((ext_vector_type(4)));
typedef short short8v __attribute__((ext_vector_type(8)));
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWgThreads = 512;  // 8 waves: 2 (rows of the output tile) x 4 (columns)
constexpr int kTileRows = 64;    // rows staged per iteration (two MFMA 16x16x32 K-steps)
constexpr int kPad = 8;          // bf16 elements of row padding in LDS

struct Chunk {
    int32_t rel, beg, end, pad;
};

// colsum_of as the C entry takes it: operand (0 none, 1 A, 2 G) | (relation + 1) << 8 -- column sums of ONE relation's rows only
// (0 in bits 8..: of every relation's).  What a chunk of relation `rel` has to sum: the operand, or nothing (its sums are zeros).
__device__ __forceinline__ int32_t colsum_for(int32_t colsum_arg, int32_t rel) {
    const int32_t only = colsum_arg >> 8;
    return (only == 0 || rel + 1 == only) ? (colsum_arg & 0xff) : 0;
}

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

__device__ __forceinline__ uint4 keep_bits(const uint4& v, uint32_t bits) {     // bit i of `bits` <-> bf16 element i of v
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i)
        w[i] &= ((bits >> (2 * i)) & 1u ? 0x0000ffffu : 0u) | ((bits >> (2 * i + 1)) & 1u ? 0xffff0000u : 0u);
    return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ uint32_t positive_bits(const uint4& v) {             // bf16 > 0: sign clear, magnitude non-zero
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t bits = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t lo = w[i] & 0xffffu, hi = w[i] >> 16;
        bits |= ((lo != 0u && lo < 0x8000u) ? 1u : 0u) << (2 * i);
        bits |= ((hi != 0u && hi < 0x8000u) ? 1u : 0u) << (2 * i + 1);
    }
    return bits;
}

// partial[chunk][k][n] = sum_{p in chunk} A[ia[p]][k] * G[ig[p]][n]   (the chunk = this workgroup's: ch)
template <int HI, int HO>
__device__ __forceinline__ void rows_wgrad_body(const bf16_t* __restrict__ A, const bf16_t* __restrict__ A2, int32_t na1,
                                                const int32_t* __restrict__ ia, const bf16_t* __restrict__ G,
                                                const bf16_t* __restrict__ G2, int32_t ng1, const int32_t* __restrict__ ig,
                                                const Chunk ch, float* __restrict__ partial, int32_t colsum_of,
                                                float* __restrict__ colsum_partial, const bf16_t* __restrict__ maskA,
                                                bf16_t* __restrict__ A_out, const uint8_t* __restrict__ maskBits, float slope) {
    constexpr int SA = HI + kPad, SG = HO + kPad;
    constexpr int MT = HI / 2 / 16, NT = HO / 4 / 16;           // 16x16 tiles per wave
    constexpr int NPA = kTileRows * HI / 8, NPG = kTileRows * HO / 8;   // 16-byte pieces per tile
    constexpr int PA = (NPA + kWgThreads - 1) / kWgThreads;             // pieces per thread per tile (A)
    constexpr int PG = (NPG + kWgThreads - 1) / kWgThreads;
    static_assert(MT >= 1 && NT >= 1, "unsupported width");
    __shared__ __attribute__((aligned(16))) bf16_t lds[2 * kTileRows * (SA + SG)];
    auto bufA = [&](int b) -> bf16_t* { return lds + b * (kTileRows * SA); };
    auto bufG = [&](int b) -> bf16_t* { return lds + 2 * kTileRows * SA + b * (kTileRows * SG); };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;                     // wave position in the output tile
    const int k0 = wm * (HI / 2), n0 = wn * (HO / 4);
    const int ntiles = (ch.end - ch.beg + kTileRows - 1) / kTileRows;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint4 ra[PA], rg[PG], rm[PA];                             // rm: ReLU mask pieces of A (maskA != NULL)
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // column sums of my 8 columns (colsum_of != 0)
    auto add_cs = [&](const uint4& v) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            cs[2 * i] += __uint_as_float(w[i] << 16);
            cs[2 * i + 1] += __uint_as_float(w[i] & 0xffff0000u);
        }
    };
    int32_t sa[PA], sg[PG];                                   // source rows of my pieces (fetched one tile ahead)
    int32_t sa_cur[PA];                                       // source rows of the pieces currently held in ra
    auto load_idx = [&](int t) {
        const int row0 = ch.beg + t * kTileRows;
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            const int piece = tid + j * kWgThreads, p = row0 + piece / (HI / 8);
            sa[j] = (piece < NPA && p < ch.end) ? (ia ? ia[p] : p) : -1;
        }
#pragma unroll
        for (int j = 0; j < PG; ++j) {
            const int piece = tid + j * kWgThreads, p = row0 + piece / (HO / 8);
            sg[j] = (piece < NPG && p < ch.end) ? (ig ? ig[p] : p) : -1;
        }
    };
    auto load_tile = [&](int) {
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            const int c = (tid + j * kWgThreads) % (HI / 8);
            sa_cur[j] = sa[j];
            ra[j] = make_uint4(0, 0, 0, 0);
            if (sa[j] >= 0) {
                const bf16_t* base = sa[j] < na1 ? A + (size_t)sa[j] * HI : A2 + (size_t)(sa[j] - na1) * HI;
                ra[j] = *reinterpret_cast<const uint4*>(base + c * 8);
                if (maskA) rm[j] = *reinterpret_cast<const uint4*>(maskA + (size_t)sa[j] * HI + c * 8);
                else if (maskBits) rm[j].x = maskBits[(size_t)sa[j] * (HI / 8) + c];      // 8 ReLU-mask bits of this piece
            }
        }
#pragma unroll
        for (int j = 0; j < PG; ++j) {
            const int c = (tid + j * kWgThreads) % (HO / 8);
            rg[j] = make_uint4(0, 0, 0, 0);
            if (sg[j] >= 0) {
                const bf16_t* base = sg[j] < ng1 ? G + (size_t)sg[j] * HO : G2 + (size_t)(sg[j] - ng1) * HO;
                rg[j] = *reinterpret_cast<const uint4*>(base + c * 8);
            }
        }
    };
    auto store_tile = [&](int b) {
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            const int piece = tid + j * kWgThreads, r = piece / (HI / 8), c = piece % (HI / 8);
            if (maskA && sa_cur[j] >= 0) {                       // A <- A where mask > 0, slope * A elsewhere (activation backward)
                ra[j] = dn_keep_or_scale_mask(ra[j], rm[j], slope);
                if (A_out) *reinterpret_cast<uint4*>(A_out + (size_t)sa_cur[j] * HI + c * 8) = ra[j];
            }
            if (maskBits && sa_cur[j] >= 0) ra[j] = dn_keep_or_scale_bits(ra[j], rm[j].x, slope);
            if (piece < NPA) *reinterpret_cast<uint4*>(bufA(b) + r * SA + c * 8) = ra[j];
            if (colsum_of == 1 && piece < NPA) add_cs(ra[j]);
        }
#pragma unroll
        for (int j = 0; j < PG; ++j) {
            const int piece = tid + j * kWgThreads, r = piece / (HO / 8), c = piece % (HO / 8);
            if (piece < NPG) *reinterpret_cast<uint4*>(bufG(b) + r * SG + c * 8) = rg[j];
            if (colsum_of == 2 && piece < NPG) add_cs(rg[j]);
        }
    };

    if (ntiles > 0) {
        load_idx(0);
        load_tile(0);
        store_tile(0);
        load_idx(1);
    }
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int b = t & 1;
        if (t + 1 < ntiles) {
            load_tile(t + 1);                                    // global loads in flight under the MFMAs
            load_idx(t + 2);                                     // (rows past the chunk end give -1)
        }
#pragma unroll
        for (int kk = 0; kk < kTileRows / 32; ++kk) {
            bf16x8 fb[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) fb[n] = tr_frag(bufG(b) + kk * 32 * SG, SG, n0 + n * 16, lane);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const bf16x8 fa = tr_frag(bufA(b) + kk * 32 * SA, SA, k0 + m * 16, lane);
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[n], acc[m][n], 0, 0, 0);
            }
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
    if (colsum_of != 0 || (colsum_partial != nullptr && tid < HI)) {   // (a job without column sums in a multi-job launch: zeros)
        // every thread owns column chunk (tid % (H/8)) in all of its pieces: fold the threads of a chunk through LDS
        constexpr int HC = HI;                                  // square: HI == HO
        constexpr int TPC = kWgThreads / (HC / 8);              // threads per column chunk
        float* red = reinterpret_cast<float*>(lds);             // tile buffers are dead after the last barrier
        const int cchunk = tid % (HC / 8), slot = tid / (HC / 8);
        if (kTileRows * HC / 8 >= kWgThreads || tid < kTileRows * HC / 8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) red[(slot * HC) + cchunk * 8 + i] = cs[i];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) red[(slot * HC) + cchunk * 8 + i] = 0.f;
        }
        __syncthreads();
        if (tid < HC) {
            float sum = 0.f;
            for (int sl = 0; sl < TPC; ++sl) sum += red[sl * HC + tid];
            colsum_partial[(size_t)blockIdx.x * HC + tid] = sum;
        }
    }
}

template <int HI, int HO>
__global__ __launch_bounds__(kWgThreads) void rows_wgrad_kernel(const bf16_t* __restrict__ A,
                                                                const bf16_t* __restrict__ A2, int32_t na1,
                                                                const int32_t* __restrict__ ia,
                                                                const bf16_t* __restrict__ G,
                                                                const bf16_t* __restrict__ G2, int32_t ng1,
                                                                const int32_t* __restrict__ ig,
                                                                const Chunk* __restrict__ chunks,
                                                                float* __restrict__ partial, int32_t colsum_of,
                                                                float* __restrict__ colsum_partial,
                                                                const bf16_t* __restrict__ maskA,
                                                                bf16_t* __restrict__ A_out,
                                                                const uint8_t* __restrict__ maskBits, float slope) {
    const Chunk ch = chunks[blockIdx.x];
    rows_wgrad_body<HI, HO>(A, A2, na1, ia, G, G2, ng1, ig, ch, partial, colsum_for(colsum_of, ch.rel), colsum_partial, maskA, A_out, maskBits, slope);
}

// Several weight gradients in ONE launch (round 6: a whole RGIN layer's -- the conv's R + 1 matrices over gathered rows, the two MLP
// layers' over dense rows -- where every launch is latency: BASELINE config 3).  The jobs' relations are numbered through
// (first_rel) and their rows laid end to end in one virtual row space (row0), so that ONE chunk table covers them; a workgroup
// finds its job by its chunk's relation.
struct WgJob {
    const bf16_t *A, *A2;
    const int32_t* ia;
    const bf16_t *G, *G2;
    const int32_t* ig;
    const uint8_t* maskBits;
    int32_t na1, ng1, colsum_of, first_rel, row0;
    float slope;
};
struct WgJobs {
    WgJob j[3];
    int32_t n;
};
template <int HI, int HO>
__global__ __launch_bounds__(kWgThreads) void rows_wgrad_multi_kernel(WgJobs jobs, const Chunk* __restrict__ chunks,
                                                                      float* __restrict__ partial, float* __restrict__ colsum_partial) {
    Chunk ch = chunks[blockIdx.x];
    int k = 0;
    if (jobs.n > 1 && ch.rel >= jobs.j[1].first_rel) k = 1;
    if (jobs.n > 2 && ch.rel >= jobs.j[2].first_rel) k = 2;
    // (selects between the three argument sets, not an indexed read: the struct lives in kernel-argument SGPRs)
    const WgJob J = k == 0 ? jobs.j[0] : (k == 1 ? jobs.j[1] : jobs.j[2]);
    ch.beg -= J.row0; ch.end -= J.row0;
    rows_wgrad_body<HI, HO>(J.A, J.A2, J.na1, J.ia, J.G, J.G2, J.ng1, J.ig, ch, partial, J.colsum_of, colsum_partial, nullptr, nullptr,
                            J.maskBits, J.slope);
}

// -------------------------------------------------------------------------------------------------
// rows_wgrad_dma_kernel: the same product with the operand rows sent global -> LDS by LDS-DMA
// (global_load_lds_dwordx4, no staging VGPRs) into a ring of 32-row stages, NST-1 stages in flight while one is
// multiplied.  The register-staged kernel above keeps 128 accumulator VGPRs + one 64-row tile in registers and is
// latency-bound (<= 64 KB in flight per CU, rocprof: 3.9 TB/s of reads); the ring keeps 96 KB (H = 256) in flight.
//   * an LDS-DMA wave-instruction writes 1 KiB lane-linearly (2 rows of 512 B): the image is unpadded, and the
//     bank spread the transposed fragment reads need comes from an XOR swizzle of the 32-byte column slots applied
//     to the per-lane SOURCE address (cdna_hip_programming.md 5.4 rule 21) and again on the read side:
//         LDS row r, slot s  holds  global slot  s ^ f(r),   f(r) = (r & 3) | ((r >> 3) & 1) << 2
//     (ds_read_b64_tr_b16 serves lanes 0-31 / 32-63 per cycle = rows {8g+q : g in {0,1} or {2,3}, q < 4}: the 8 values
//     of f put their 32-byte reads on 8 distinct slots of the 256-byte bank row -> conflict-free)
//   * row indices come through scalar loads (wave-uniform: a wave stages 4 consecutive rows of each operand per
//     tile), so no VGPR-destination global load shares the vmcnt queue with the DMAs; the wait is a counted
//     s_waitcnt vmcnt(GL * (NST - 2)) followed by ONE raw s_barrier per tile (never __syncthreads(): its fence would
//     drain the ring).  Rows past the end of the chunk read a zero row, tiles past the end are issued as zero tiles so
//     the count stays static; the ring is drained (vmcnt(0)) before the LDS is reused or the workgroup ends.
//   * used when no ReLU mask is folded in (maskA == NULL); H in {128, 256} (64-row stages at H = 128: same bytes per stage).
// -------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) uint4 g_zero_row[64];        // 1 KiB of zeros (device globals are zero-initialised)

#ifndef DN_WGM_ABL
#define DN_WGM_ABL 0        // diagnostic builds (DN_BUILD_EXTRA=-DDN_WGM_ABL=n): 1 no mask pass, 2 mask bits from the zero row
#endif
template <int H, bool MASKED, bool DENSE>
__global__ __launch_bounds__(kWgThreads) void rows_wgrad_dma_kernel(const bf16_t* __restrict__ A,
                                                                    const bf16_t* __restrict__ A2, int32_t na1,
                                                                    const int32_t* __restrict__ ia,
                                                                    const bf16_t* __restrict__ G,
                                                                    const bf16_t* __restrict__ G2, int32_t ng1,
                                                                    const int32_t* __restrict__ ig,
                                                                    const Chunk* __restrict__ chunks,
                                                                    float* __restrict__ partial, int32_t colsum_arg,
                                                                    float* __restrict__ colsum_partial,
                                                                    const uint8_t* __restrict__ maskBits, float slope) {
    static_assert(H == 256 || H == 128, "unsupported width");
    constexpr int TR = (H == 256) ? 32 : 64;       // rows per stage: 32 KiB per stage at either width (1 or 2 MFMA K-steps)
    constexpr int NST = 4;                         // ring stages (128 KiB)
    constexpr int ROWB = 2 * H;                    // bytes per row
    constexpr int MATB = TR * ROWB;                // bytes per operand per stage
    constexpr int BITB = MASKED ? 1024 : 0;        // per stage: the tile's ReLU-mask bits (TR rows x H/8 bytes = 1 KiB)
    constexpr int STB = 2 * MATB + BITB;
    constexpr int LPRW = H / 8;                    // lanes (16-byte pieces) per row
    constexpr int RPI = 64 / LPRW;                 // rows per DMA wave-instruction (2 or 4)
    constexpr int RW = TR / 8;                     // rows of each operand a wave stages per tile (4 or 8)
    constexpr int PPW = RW / RPI;                  // DMA instructions per wave, operand and tile
    constexpr int GL = 2 * PPW;                    // DMA instructions per wave and tile (+ 1 on the wave that fetches a tile's mask bits)
    constexpr int MT = H / 2 / 16, NT = H / 4 / 16;
    __shared__ __attribute__((aligned(1024))) char lds[NST * STB];
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;

    const Chunk ch = chunks[blockIdx.x];
    const int32_t colsum_all = colsum_arg & 0xff, colsum_of = colsum_for(colsum_arg, ch.rel);   // (a chunk of another relation: zeros)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int ntiles = (ch.end - ch.beg + TR - 1) / TR;
    const char* zero = reinterpret_cast<const char*>(g_zero_row);

    // ---- DMA side: this lane's place in a piece ---------------------------------------------------------------
    const int rin = lane / LPRW, cpos = lane % LPRW;
    int32_t nxa[RW], nxg[RW];                      // wave-uniform source rows of the next tile to issue
    auto load_idx = [&](int T) {                   // scalar loads (wave-uniform addresses), branch-free
        if constexpr (DENSE) return;                // (both operands in row order: the addresses follow from the tile number)
        const int p0 = ch.beg + T * TR + RW * wave, pe = ch.end - 1;
        int32_t pc[RW];
#pragma unroll
        for (int k = 0; k < RW; ++k) pc[k] = min(p0 + k, pe);
        if (ia) {
#pragma unroll
            for (int k = 0; k < RW; ++k) nxa[k] = ia[pc[k]];
        } else {
#pragma unroll
            for (int k = 0; k < RW; ++k) nxa[k] = pc[k];
        }
        if (ig) {
#pragma unroll
            for (int k = 0; k < RW; ++k) nxg[k] = ig[pc[k]];
        } else {
#pragma unroll
            for (int k = 0; k < RW; ++k) nxg[k] = pc[k];
        }
#pragma unroll
        for (int k = 0; k < RW; ++k) {
            const bool ok = p0 + k <= pe;          // rows past the chunk end (and whole tiles past it) read the zero row
            nxa[k] = ok ? nxa[k] : -1;
            nxg[k] = ok ? nxg[k] : -1;
        }
    };
    auto issue = [&](int T) {
        const unsigned st = lds_base + (unsigned)(T % NST) * STB;
        uint64_t pgs[PPW];
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int rl = RW * wave + RPI * j + rin;                      // row of the stage this lane fills
            const int f = (rl & 3) | (((rl >> 3) & 1) << 2);
            const int gch = ((((cpos >> 1) ^ f) << 1) | (cpos & 1)) * 16;     // source byte offset inside the row
            if constexpr (DENSE) {
                // no index, no second source (the MLP's weight gradients): row p of both operands, or the zero row past the
                // chunk's end -- one 64-bit multiply-add and one select per operand instead of the general path's ~25
                // instructions per DMA pair (what the issue phase of these launches costs is this arithmetic: see rows_wgrad_ix_kernel)
                const int p = ch.beg + T * TR + rl;
                const bool okp = p < ch.end;
                const uint64_t off = (uint64_t)(uint32_t)p * ROWB + (uint64_t)gch;
                const uint64_t za = (uint64_t)(uintptr_t)zero + (uint64_t)gch;
                const uint64_t pa = okp ? (uint64_t)(uintptr_t)A + off : za;
                const uint64_t pg = okp ? (uint64_t)(uintptr_t)G + off : za;
                const unsigned da = st + (unsigned)(RW * wave + RPI * j) * ROWB;
                glds16(reinterpret_cast<const char*>(pa), da);
                if constexpr (MASKED) pgs[j] = pg;
                else glds16(reinterpret_cast<const char*>(pg), da + MATB);
                continue;
            }
            int32_t ra = nxa[RPI * j], rg = nxg[RPI * j];
#pragma unroll
            for (int k = 1; k < RPI; ++k) {
                ra = (rin == k) ? nxa[RPI * j + k] : ra;
                rg = (rin == k) ? nxg[RPI * j + k] : rg;
            }
            // (selects, not branches: written as nested ?: on pointers hipcc emits two exec-masked regions per DMA)
            const bool a2 = ra >= na1, g2 = rg >= ng1;
            const uint64_t ba = a2 ? (uint64_t)(uintptr_t)A2 : (uint64_t)(uintptr_t)A;
            const uint64_t bg = g2 ? (uint64_t)(uintptr_t)G2 : (uint64_t)(uintptr_t)G;
            const uint64_t oa = (uint64_t)(uint32_t)(a2 ? ra - na1 : ra) * ROWB, og = (uint64_t)(uint32_t)(g2 ? rg - ng1 : rg) * ROWB;
            const uint64_t pa = (ra < 0 ? (uint64_t)(uintptr_t)zero : ba + oa) + (uint64_t)gch;
            const uint64_t pg = (rg < 0 ? (uint64_t)(uintptr_t)zero : bg + og) + (uint64_t)gch;
            const unsigned da = st + (unsigned)(RW * wave + RPI * j) * ROWB; // wave-uniform; lane l lands at + 16 l
            glds16(reinterpret_cast<const char*>(pa), da);
            if constexpr (MASKED) pgs[j] = pg;                              // (masked: every A piece and the bits go first, see the loop)
            else glds16(reinterpret_cast<const char*>(pg), da + MATB);
        }
        if constexpr (MASKED) {
            // the tile's mask bits (TR rows x H/8 bytes = 1 KiB, contiguous: the masked operand is never gathered) go as ONE
            // full-width DMA by wave T % 8 -- eight 128-byte requests per tile (every wave its own rows) measured 16 us per
            // launch at 127 k rows, as much as the mask pass itself
            constexpr int LPB = H / 128;                                   // lanes per row of bits
            if (wave == (T & 7)) {
                const int p = ch.beg + T * TR + lane / LPB;
                const char* pb = (p >= ch.end || (DN_WGM_ABL & 2)) ? zero + lane * 16
                                                                   : reinterpret_cast<const char*>(maskBits) + (size_t)p * (H / 8) + (lane % LPB) * 16;
                glds16(pb, st + 2 * MATB);
            }
#pragma unroll
            for (int j = 0; j < PPW; ++j)
                glds16(reinterpret_cast<const char*>(pgs[j]), st + (unsigned)(RW * wave + RPI * j) * ROWB + MATB);
        }
    };

    // ---- MFMA side ----------------------------------------------------------------------------------------
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fg = lane >> 4, fq = (lane & 15) >> 2, fp = lane & 3;
    const int fsw = fq | ((fg & 1) << 2);                                  // f(r) of the rows this lane reads
    const int frow = (8 * fg + fq) * ROWB + 8 * fp;
    typedef short4v __attribute__((address_space(3))) * lds_p;
    auto frag = [&](const char* mat, int slot) -> bf16x8 {
        const char* a0 = mat + frow + ((slot ^ fsw) << 5);
        const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a0));
        const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a0 + 4 * ROWB));
        const short8v f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, f);
    };
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    constexpr int CPT = TR * LPRW / kWgThreads;                            // column-sum pieces per thread and tile (2 or 1)
    const int cchunk = tid % LPRW, crow = tid / LPRW;

    // ---- prologue: NST-1 tiles in flight (chunk tables never hold empty chunks; guard anyway: pe must be a valid row) ----
    if (ntiles <= 0) {
        for (int i = tid; i < H * H; i += kWgThreads) partial[(size_t)blockIdx.x * H * H + i] = 0.f;
        if (colsum_all != 0 && tid < H) colsum_partial[(size_t)blockIdx.x * H + tid] = 0.f;
        return;
    }
    load_idx(0);
#pragma unroll 1
    for (int T = 0; T < NST - 1; ++T) {
        issue(T);
        load_idx(T + 1);
    }
    // Masked variant: the activation mask is applied to a landed A tile IN PLACE, one tile AHEAD of the products: tile t+1 is
    // masked (and its column sums taken, from the registers the masked piece passes through) in the same barrier interval in
    // which tile t is multiplied, so a tile costs ONE barrier and the read-modify-write is off the path between a tile's arrival
    // and its MFMAs.  A wave issues per tile [A pieces][bits][G pieces]; masking tile t+1 needs its A pieces and bits only, so
    // the counted wait leaves the G pieces of t+1 and all of t+2 in flight (half a stage less than the unmasked ring).
    constexpr int MPT = TR * LPRW / kWgThreads;                            // pieces a thread masks per tile (2), rows mrow + j * MRS
    constexpr int MRS = kWgThreads / LPRW;                                 // (a multiple of 16: the swizzle f(r) is the same for both)
    const int mrow = tid / LPRW, mq = tid % LPRW;
    const int mf = (mrow & 3) | (((mrow >> 3) & 1) << 2);
    const int mchunk = (((mq >> 1) ^ mf) << 1) | (mq & 1);                  // the 8 columns this thread's pieces hold
    static_assert(MRS % 16 == 0, "both pieces of a thread must share their column chunk");
    auto mask_tile = [&](int T) {
        char* sT = lds + (T % NST) * STB;
        const uint8_t* sB = reinterpret_cast<const uint8_t*>(sT + 2 * MATB);
#pragma unroll
        for (int j = 0; j < MPT; ++j) {
            const int r = mrow + j * MRS;
            uint4* pp = reinterpret_cast<uint4*>(sT + r * ROWB + mq * 16);
            const uint4 v = dn_keep_or_scale_bits(*pp, sB[r * (H / 8) + mchunk], slope);
            *pp = v;
            if (colsum_of == 1) {
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    cs[2 * i] += __uint_as_float(w[i] << 16);
                    cs[2 * i + 1] += __uint_as_float(w[i] & 0xffff0000u);
                }
            }
        }
    };
    if constexpr (MASKED) {
        wait_vmcnt<PPW + GL*(NST - 2)>();                                  // my A pieces and bits of tile 0 have landed
        __builtin_amdgcn_s_barrier();
        mask_tile(0);
    }
#pragma unroll 1
    for (int t = 0; t < ntiles; ++t) {
        if constexpr (MASKED) wait_vmcnt<PPW + GL*(NST - 3)>();            // all of tile t, A pieces + bits of tile t+1
        else wait_vmcnt<GL*(NST - 2)>();                                   // my pieces of tile t have landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // my fragment reads of tile t-1 (and mask writes of tile t) are done
        __builtin_amdgcn_s_barrier();                                      // everyone's have / are
        issue(t + NST - 1);                                                // refill the stage tile t-1 used
        load_idx(t + NST);
        char* sA = lds + (t % NST) * STB;
        const char* sG = sA + MATB;
#pragma unroll
        for (int kk = 0; kk < TR / 32; ++kk) {
            bf16x8 fb[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) fb[n] = frag(sG + kk * 32 * ROWB, wn * NT + n);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const bf16x8 fa = frag(sA + kk * 32 * ROWB, wm * MT + m);
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[n], acc[m][n], 0, 0, 0);
            }
        }
        if constexpr (MASKED) { if (!(DN_WGM_ABL & 1)) mask_tile(t + 1); } // behind the MFMAs in program order (a zero tile past the end stays zero)
        if (colsum_of != 0 && !(MASKED && colsum_of == 1)) {
            const char* M = colsum_of == 1 ? sA : sG;
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                const int r = crow + j * (kWgThreads / LPRW);
                const int f = (r & 3) | (((r >> 3) & 1) << 2);
                const int pos = (((cchunk >> 1) ^ f) << 1) | (cchunk & 1);
                const uint4 v = *reinterpret_cast<const uint4*>(M + r * ROWB + pos * 16);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    cs[2 * i] += __uint_as_float(w[i] << 16);
                    cs[2 * i + 1] += __uint_as_float(w[i] & 0xffff0000u);
                }
            }
        }
    }
    wait_vmcnt<0>();                                                       // drain the zero tiles still in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    float* out = partial + (size_t)blockIdx.x * H * H;
    const int k0 = wm * (H / 2), n0 = wn * (H / 4);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = k0 + m * 16 + (lane >> 4) * 4 + i, c = n0 + n * 16 + (lane & 15);
                out[(size_t)k * H + c] = acc[m][n][i];
            }
    if (colsum_all != 0) {
        constexpr int TPC = kWgThreads / LPRW;                             // threads per column chunk
        float* red = reinterpret_cast<float*>(lds);
        const int ochunk = (MASKED && colsum_of == 1) ? mchunk : cchunk;    // (masked: summed where the piece was masked)
#pragma unroll
        for (int i = 0; i < 8; ++i) red[crow * H + ochunk * 8 + i] = cs[i];
        __syncthreads();
        if (tid < H) {
            float sum = 0.f;
            for (int sl = 0; sl < TPC; ++sl) sum += red[sl * H + tid];
            colsum_partial[(size_t)blockIdx.x * H + tid] = sum;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// rows_wgrad_ix_kernel: the LDS-DMA ring above for GATHERED operands (both index arrays given; H = 256; the conv's weight
// gradient), with the per-tile bookkeeping moved off the scalar unit.  In the kernel above every wave fetches its 8 row indices
// with scalar loads right before the fragment reads, and the `s_waitcnt lgkmcnt(0)` in front of the first MFMA (SMEM shares the
// counter with LDS) waits for them: ~2,400-2,900 cycles per 32-row tile against 1,024 of MFMA, from L2 as from HBM.  Here a
// wave's 8 indices of a tile arrive as ONE 8-lane LDS-DMA two tiles ahead of their use (into a small per-wave LDS ring), the
// row addresses are picked with selects instead of exec-masked branches, and nothing in the loop touches SMEM.
// vmcnt arithmetic: a wave issues per tile [index DMA] [4 row DMAs], in this order.  (Walking the rows in the L2-blocked sweep order
// with one partial product per run of tiles was built on top of this kernel and measured: 1.1 GB fewer HBM reads per launch, the same
// time -- docs/LAB_NOTES.md, round 3.)
// -------------------------------------------------------------------------------------------------
#ifdef DN_WG_STATS
// diagnostic build (-DDN_WG_STATS): shader-clock cycles of the last launch per workgroup, waves 0 (issues its DMAs first) and 4
// (multiplies first): {loop, wait + barrier, DMA issue, fragments + MFMAs + column sums}, and the loop's wall ticks (100 MHz)
__device__ unsigned long long g_wg_stats[256][2][5];
#define DN_WG_STAMP() __builtin_amdgcn_s_memtime()
#define DN_WG_STAT(var, expr) var += (expr)
#else
#define DN_WG_STAMP() 0ull
#define DN_WG_STAT(var, expr)
#endif
__global__ __launch_bounds__(kWgThreads) void rows_wgrad_ix_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ A2,
                                                                   int32_t na1, const int32_t* __restrict__ ia,
                                                                   const bf16_t* __restrict__ G, const bf16_t* __restrict__ G2,
                                                                   int32_t ng1, const int32_t* __restrict__ ig,
                                                                   const Chunk* __restrict__ chunks, float* __restrict__ partial,
                                                                   int32_t colsum_arg, float* __restrict__ colsum_partial) {
    constexpr int H = 256, TR = 32, NST = 4, ROWB = 2 * H, MATB = TR * ROWB, STB = 2 * MATB;
    constexpr int LPRW = H / 8, RW = TR / 8;       // 32 lanes per row; 4 rows of each operand per wave and tile
    constexpr int MT = H / 2 / 16, NT = H / 4 / 16;
    constexpr int kSlots = 8;                      // per-wave ring of index octets
    constexpr int kOps = 5;                        // vector-memory operations a wave issues per tile
    __shared__ __attribute__((aligned(1024))) char lds[NST * STB];
    __shared__ __attribute__((aligned(32))) int32_t idxL[8][kSlots][8];       // {ia[p0..p0+3], ig[p0..p0+3]} of a wave's rows
    typedef __attribute__((address_space(3))) char* lds_wp;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_wp)lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const Chunk ch = chunks[blockIdx.x];
    const int32_t colsum_all = colsum_arg & 0xff, colsum_of = colsum_for(colsum_arg, ch.rel);
    const int ntiles = (ch.end - ch.beg + TR - 1) / TR;
    const int32_t ch_beg = ch.beg, ch_end = ch.end;
    const char* zero = reinterpret_cast<const char*>(g_zero_row);
    const unsigned idx_base = (unsigned)(uintptr_t)(lds_wp)&idxL[wave][0][0];

    // {first row of my 4, last row of the tile, live} of tile T (wave-uniform values; tiles past the end repeat the last one)
    auto tile_rows = [&](int T, int& p0, int& pe, bool& live) {
        live = T < ntiles;
        const int Tc = min(T, ntiles - 1);
        p0 = ch_beg + Tc * TR + RW * wave;
        pe = min(ch_beg + Tc * TR + TR, ch_end) - 1;
    };
    // The index octet of a wave and tile sits in the ring as {ia0, ia2, ig0, ig2, ia1, ia3, ig1, ig3} (rows p0 .. p0 + 3 of the wave):
    // the four values a lane needs for its two DMA pairs -- rows rin and 2 + rin of both operands -- are ONE 16-byte LDS read.
    auto dma_idx = [&](int T) {                    // lanes 0..7 fetch one index each (lane l lands at + 4 l)
        int p0, pe;
        bool live;
        tile_rows(T, p0, pe, live);
        const int row = (lane >> 2) + 2 * (lane & 1);                      // lane l: row (l >> 2) + 2 (l & 1) of ia (l & 2 clear) / ig
        const int pc = max(min(p0 + row, pe), 0);
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(idx_base + (unsigned)(T % kSlots) * 32u));
        if (lane < 8) glds4(((lane & 2) ? ig : ia) + pc, dst);
    };
    const int rin = lane / LPRW, cpos = lane % LPRW;
    // Address of a row = base + index * 512 (+ the second source's displacement for indices >= n1): one 64-bit multiply-add, one
    // compare and one 64-bit select.  (Round 3's form -- compare, base select, subtract, shift, add, zero-row select per operand
    // behind two dependent LDS reads and two exec-masked regions -- was ~230 of a tile's ~790 timer ticks per wave: the address
    // arithmetic, not the DMA instructions, is what the issue phase cost; ablations in docs/LAB_NOTES.md, round 4.)
    const uint64_t dA = (uint64_t)(uintptr_t)A2 - (uint64_t)(uintptr_t)A - (uint64_t)(uint32_t)na1 * ROWB;   // (A2 == NULL: never selected)
    const uint64_t dG = (uint64_t)(uintptr_t)G2 - (uint64_t)(uintptr_t)G - (uint64_t)(uint32_t)ng1 * ROWB;
    int gch2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rl = RW * wave + 2 * j + rin;                            // row of the stage this lane fills
        const int f = (rl & 3) | (((rl >> 3) & 1) << 2);
        gch2[j] = ((((cpos >> 1) ^ f) << 1) | (cpos & 1)) * 16;             // source byte offset inside the row
    }
    auto issue = [&](int T) {                      // the 4 row DMAs of tile T (indices of tile T have landed)
        int p0, pe;
        bool live;
        tile_rows(T, p0, pe, live);
        typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
        const i32x4 iv = *reinterpret_cast<const i32x4*>(&idxL[wave][T % kSlots][4 * rin]);   // {ia[k], ia[2 + k], ig[k], ig[2 + k]}, k = rin
        const unsigned st = lds_base + (unsigned)(T % NST) * STB;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = 2 * j + rin;                                     // my row of the wave's 4
            // rows past the chunk's end (and whole tiles past it): G reads the zero row, so the pair adds nothing; A repeats the
            // chunk's last row (the index was clamped), unless its column sums are wanted -- then it reads the zero row too
            const bool ok = live && p0 + k <= pe;
            const int32_t ra = iv[j], rg = iv[2 + j];
            uint64_t pa = (uint64_t)(uintptr_t)A + (uint64_t)(uint32_t)ra * ROWB + (ra >= na1 ? dA : 0ull);
            uint64_t pg = (uint64_t)(uintptr_t)G + (uint64_t)(uint32_t)rg * ROWB + (rg >= ng1 ? dG : 0ull);
            pg = ok ? pg : (uint64_t)(uintptr_t)zero;
            if (colsum_of == 1) pa = ok ? pa : (uint64_t)(uintptr_t)zero;  // (wave-uniform condition)
            const unsigned da = st + (unsigned)(RW * wave + 2 * j) * ROWB;  // wave-uniform; lane l lands at + 16 l
#if defined(DN_WG_STATS) && defined(DN_WG_ABL)
            if (DN_WG_ABL & 1) { asm volatile("" :: "v"(pa), "v"(pg)); continue; }                       // (ablation: address math, no DMAs)
            if (DN_WG_ABL & 2) { glds16(zero + lane * 16, da); glds16(zero + lane * 16, da + MATB); continue; }   // (DMAs from one row, no math)
#endif
            glds16(reinterpret_cast<const char*>(pa + (uint64_t)gch2[j]), da);
            glds16(reinterpret_cast<const char*>(pg + (uint64_t)gch2[j]), da + MATB);
        }
    };

    // ---- MFMA side (as in rows_wgrad_dma_kernel) ----------------------------------------------------------------
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fg = lane >> 4, fq = (lane & 15) >> 2, fp = lane & 3;
    const int fsw = fq | ((fg & 1) << 2);
    const int frow = (8 * fg + fq) * ROWB + 8 * fp;
    typedef short4v __attribute__((address_space(3))) * lds_p;
    auto frag = [&](const char* mat, int slot) -> bf16x8 {
        const char* a0 = mat + frow + ((slot ^ fsw) << 5);
        const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a0));
        const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a0 + 4 * ROWB));
        const short8v f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, f);
    };
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    constexpr int CPT = TR * LPRW / kWgThreads;                            // column-sum pieces per thread and tile (2)
    const int cchunk = tid % LPRW, crow = tid / LPRW;
    auto flush = [&](int slab) {                   // one partial product (+ its column sums) out to workspace slab `slab`; sums restart
        float* out = partial + (size_t)slab * H * H;
        const int k0 = wm * (H / 2), n0 = wn * (H / 4);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = k0 + m * 16 + (lane >> 4) * 4 + i, c = n0 + n * 16 + (lane & 15);
                    out[(size_t)k * H + c] = acc[m][n][i];
                }
                acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
    };

    // ---- prologue --------------------------------------------------------------------------------------------------
    if (ntiles <= 0) {
        for (int i = tid; i < H * H; i += kWgThreads) partial[(size_t)blockIdx.x * H * H + i] = 0.f;
        if (colsum_all != 0 && tid < H) colsum_partial[(size_t)blockIdx.x * H + tid] = 0.f;
        return;
    }
#pragma unroll 1
    for (int T = 0; T < 5; ++T) dma_idx(T);
    wait_vmcnt<0>();
#pragma unroll 1
    for (int T = 0; T < NST - 1; ++T) issue(T);
    wait_vmcnt<8>();                                                       // tile 0 has landed (the loop's count starts at tile 1)
    const bool dma_first = wave < 4;
    unsigned long long st_bar = 0, st_dma = 0, st_mm = 0;
    const unsigned long long l0 = DN_WG_STAMP();
#ifdef DN_WG_STATS
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll 1
    for (int t = 0; t < ntiles; ++t) {
        const unsigned long long a0 = DN_WG_STAMP();
        // issued after the index octet of tile t+3 (two tiles ago): its tile's 4 row DMAs and last tile's kOps operations
        wait_vmcnt<kOps + 4>();                                            // rows of tile t, indices of tile t+3
        __builtin_amdgcn_s_barrier();                                      // everyone's have; stage of tile t-1 is free
        const unsigned long long a1 = DN_WG_STAMP();
        DN_WG_STAT(st_bar, a1 - a0);
        // the two waves of a SIMD (w and w + 4) take turns: one issues its DMAs while the other multiplies (issuing a DMA costs a
        // wave 100-185 cycles; in lock step the SIMD's matrix pipe would idle through both waves' issue phases)
        if (dma_first) {
            dma_idx(t + 5);
            issue(t + NST - 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long a2 = DN_WG_STAMP();
        if (dma_first) DN_WG_STAT(st_dma, a2 - a1);
        char* sA = lds + (t % NST) * STB;
        const char* sG = sA + MATB;
        {
            bf16x8 fb[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) fb[n] = frag(sG, wn * NT + n);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const bf16x8 fa = frag(sA, wm * MT + m);
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[n], acc[m][n], 0, 0, 0);
            }
        }
        if (colsum_of != 0) {
            const char* M = colsum_of == 1 ? sA : sG;
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                const int r = crow + j * (kWgThreads / LPRW);
                const int f = (r & 3) | (((r >> 3) & 1) << 2);
                const int pos = (((cchunk >> 1) ^ f) << 1) | (cchunk & 1);
                const uint4 v = *reinterpret_cast<const uint4*>(M + r * ROWB + pos * 16);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    cs[2 * i] += __uint_as_float(w[i] << 16);
                    cs[2 * i + 1] += __uint_as_float(w[i] & 0xffff0000u);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long a3 = DN_WG_STAMP();
        DN_WG_STAT(st_mm, a3 - a2);
        if (!dma_first) {
            dma_idx(t + 5);
            issue(t + NST - 1);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // my LDS reads of tile t are done before the next barrier
        if (!dma_first) DN_WG_STAT(st_dma, DN_WG_STAMP() - a3);
    }
#ifdef DN_WG_STATS
    if ((wave == 0 || wave == 4) && lane == 0 && blockIdx.x < 256) {
        unsigned long long* o = g_wg_stats[blockIdx.x][wave >> 2];
        o[0] = DN_WG_STAMP() - l0; o[1] = st_bar; o[2] = st_dma; o[3] = st_mm; o[4] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
#endif
    wait_vmcnt<0>();                                                       // drain what is still in flight
    __builtin_amdgcn_s_barrier();
    flush((int)blockIdx.x);
    if (colsum_all != 0) {
        constexpr int TPC = kWgThreads / LPRW;
        float* red = reinterpret_cast<float*>(lds);
#pragma unroll
        for (int i = 0; i < 8; ++i) red[crow * H + cchunk * 8 + i] = cs[i];
        __syncthreads();
        if (tid < H) {
            float sum = 0.f;
            for (int sl = 0; sl < TPC; ++sl) sum += red[sl * H + tid];
            colsum_partial[(size_t)blockIdx.x * H + tid] = sum;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// rows_wgrad_ls_kernel ("loader-specialised", H = 256): the ring of the two kernels above with the ROLES of a tile's work on
// separate waves, as in the ring transform (dn_rel_ring.hip): 12 waves per workgroup, 3 per SIMD --
//   * waves 8..11 (loaders) issue every LDS-DMA of a tile (8 rows of each operand per loader, the index octets two groups ahead
//     through a per-wave LDS ring), apply the activation mask to a landed tile in place (MODE 2) and take the column sums;
//   * waves 0..7 (compute) do nothing but fragment reads and MFMAs (128 x 64 of the 256 x 256 product each, as above).
// Why: with the rows cache-resident (indices folded into 1 MB, tools/wgrad_local_exp.py) the kernels above still need 1.09 us per
// tile and workgroup against 0.49 of MFMA work -- every wave's tile was [wait, barrier, 4-5 DMA issues at 60-185 cycles each +
// their address arithmetic, 24 fragment reads, 32 MFMAs, column sums], two such streams per SIMD; the launch was bound by its
// own instruction stream (docs/LAB_NOTES.md, round 6), which is why neither fewer HBM bytes nor cache residency made it faster.
// Same tiles in the same order, the same MFMA per tile and accumulator, the same partition of the column sums: results are
// bit-identical to the kernels above.
//   MODE 0: gathered operands (both index arrays; second sources A2 / G2), 1: both operands in row order, 2: row order + the A
//   operand masked by bits (dn_rows_wgrad_bf16's maskBits).
// vmcnt: a loader issues per tile ONE group, [index octet of tile s + 7] [8 row pieces of tile s + 3] behind barrier s (MODE 0;
// 8 pieces in the other modes, + the tile's mask bits on loader s & 3 between the A and G pieces in MODE 2).
// -------------------------------------------------------------------------------------------------
constexpr int kLsCompute = 8, kLsLoaders = 4, kLsThreads = 64 * (kLsCompute + kLsLoaders);

// The body takes its LDS from the kernel that calls it (a kernel serving several MODEs -- rows_wgrad_ls_multi_kernel -- declares it
// once, for the largest): lds = NST stages of [A rows | G rows | mask bits (MODE 2)], idxL / adrL = the loaders' index and address
// rings.  ch: the workgroup's chunk with rows and relation LOCAL to the job (a multi-job launch lays the jobs' rows end to end:
// row0 = where this job's begin in the chunk table, vrel = the chunk's relation as the table and chunk_ptr number it).
constexpr int kLsStageBytes = 2 * 32 * 512, kLsStages = 4, kLsSlots = 8;
template <int MODE>
__device__ __forceinline__ void wgrad_ls_body(const bf16_t* __restrict__ A, const bf16_t* __restrict__ A2,
                                              int32_t na1, const int32_t* __restrict__ ia,
                                              const bf16_t* __restrict__ G, const bf16_t* __restrict__ G2,
                                              int32_t ng1, const int32_t* __restrict__ ig,
                                              const Chunk* __restrict__ chunks, const Chunk ch, int32_t vrel, int32_t row0,
                                              float* __restrict__ partial,
                                              int32_t colsum_arg, float* __restrict__ colsum_partial,
                                              const uint8_t* __restrict__ maskBits, float slope,
                                              const int32_t* __restrict__ chunk_ptr, char* lds, int32_t (*idxL)[kLsSlots][16],
                                              uint64_t (*adrL)[kLsSlots][16]) {
    constexpr bool GATHER = MODE == 0, MASKED = MODE == 2;
    constexpr int H = 256, TR = 32, NST = kLsStages, ROWB = 2 * H, MATB = TR * ROWB, BITB = MASKED ? 1024 : 0, STB = 2 * MATB + BITB;
    static_assert(2 * MATB == kLsStageBytes, "stage size");
    constexpr int LPRW = H / 8;                    // 32 lanes per row
    constexpr int MT = H / 2 / 16, NT = H / 4 / 16;
    constexpr int kSlots = kLsSlots;               // per-loader ring of index octets (16 words a slot)
    constexpr int kGroup = GATHER ? 9 : 8;         // vector-memory operations of a loader's group (without the mask bits)
    typedef __attribute__((address_space(3))) char* lds_wp;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_wp)lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int32_t colsum_all = colsum_arg & 0xff, colsum_of = colsum_for(colsum_arg, ch.rel);   // (a chunk of another relation: zeros)
    // Tile T of this workgroup = rows [ch_beg + T * t_step, + 32) below ch_end.  chunk_ptr == NULL: the chunk's rows, t_step = 32.
    // chunk_ptr given (MODE 0): the K chunks of a relation are taken as K INTERLEAVED pieces of the relation's rows -- piece k =
    // tiles k, k + K, k + 2 K, ... -- so that all workgroups walk the node range at the same pace and the relations that share an
    // x / g row ask for it within a few tiles of each other (the partial of chunk i is then the sum over piece i's tiles; the
    // reduce adds a relation's partials as before).
    int32_t ch_beg = ch.beg, ch_end = ch.end, t_step = TR;
    int ntiles = (ch.end - ch.beg + TR - 1) / TR;
    if (GATHER && chunk_ptr != nullptr) {
        const int32_t c0 = chunk_ptr[vrel], c1 = chunk_ptr[vrel + 1];
        ntiles = 0;
        if ((int32_t)blockIdx.x >= c0 && (int32_t)blockIdx.x < c1) {         // (entries past the last chunk: empty pieces)
            ch_beg = chunks[c0].beg - row0 + TR * ((int32_t)blockIdx.x - c0);
            ch_end = chunks[c1 - 1].end - row0;
            t_step = TR * (c1 - c0);
            ntiles = ch_beg < ch_end ? (ch_end - ch_beg + t_step - 1) / t_step : 0;
        }
    }
    if (ntiles <= 0) {                             // (chunk tables never hold empty chunks; guard anyway)
        for (int i = tid; i < H * H; i += kLsThreads) partial[(size_t)blockIdx.x * H * H + i] = 0.f;
        if (colsum_all != 0 && tid < H) colsum_partial[(size_t)blockIdx.x * H + tid] = 0.f;
        return;
    }
    // column sums: loader thread lt stands for the two threads (crow = lt / 32 and 8 + lt / 32, chunk lt % 32) of the 512-thread
    // kernels above -- rows crow, crow + 16 of every tile, in this order
    float cs[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 8; ++i) cs[s][i] = 0.f;
    int ochunk[2] = {0, 0};

    if (wave >= kLsCompute) {
        // ------------------------------------------------------------------------------------------------ loaders
        const int q = wave - kLsCompute;
        const int rin = lane / LPRW, cpos = lane % LPRW;
        const char* zero = reinterpret_cast<const char*>(g_zero_row);
        const unsigned idx_base = (unsigned)(uintptr_t)(lds_wp)&idxL[GATHER ? q : 0][0][0];
        int gch[4];                                // source byte offset inside the row, per piece (the XOR swizzle of the image)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rl = 8 * q + 2 * j + rin;
            const int f = (rl & 3) | (((rl >> 3) & 1) << 2);
            gch[j] = ((((cpos >> 1) ^ f) << 1) | (cpos & 1)) * 16;
        }
        auto tile_rows = [&](int T, int& p0, int& pe, bool& live) {
            live = T < ntiles;
            const int Tc = min(T, ntiles - 1);
            p0 = ch_beg + Tc * t_step + 8 * q;
            pe = min(ch_beg + Tc * t_step + TR, ch_end) - 1;
        };
        // a loader's index octet of a tile in its ring: {ia0 ia2 ia4 ia6 | ia1 ia3 ia5 ia7 | ig0 ig2 ig4 ig6 | ig1 ig3 ig5 ig7} (rows
        // p0 .. p0 + 7): the four indices of an operand a lane needs -- rows 2 j + rin -- are ONE 16-byte LDS read
        auto dma_idx = [&](int T) {
            if constexpr (!GATHER) return;
            int p0, pe;
            bool live;
            tile_rows(T, p0, pe, live);
            const int k = 2 * (lane & 3) + ((lane >> 2) & 1);
            const int pc = max(min(p0 + k, pe), 0);
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(idx_base + (unsigned)(T % kSlots) * 64u));
            if (lane < 16) glds4(((lane & 8) ? ig : ia) + pc, dst);
        };
        // Row addresses are worked out ONCE per row, not once per lane: lanes 0..15 of a loader hold its 16 rows of a tile (lane l:
        // operand l >> 3, row 2 (l & 3) + ((l >> 2) & 1): the order of the index octet), compute base + index * 512 (+ the second
        // source's displacement; the zero row past the chunk's end) and put the 64-bit address into the loader's address ring a
        // tile ahead of its use; a DMA's lanes then read their row's address back (the four of an operand are two 16-byte reads)
        // and add the in-row offset.  (Per lane and piece -- a 64-bit multiply-add, compares and four selects, x 8 -- the loaders'
        // VALU work per tile was longer than the MFMA waves' tile: those instructions issue in the gaps the MFMAs leave.)
        const uint64_t dA = (uint64_t)(uintptr_t)A2 - (uint64_t)(uintptr_t)A - (uint64_t)(uint32_t)na1 * ROWB;   // (A2 == NULL: never selected)
        const uint64_t dG = (uint64_t)(uintptr_t)G2 - (uint64_t)(uintptr_t)G - (uint64_t)(uint32_t)ng1 * ROWB;
        const int l16 = lane & 15;
        const bool isg = (l16 & 8) != 0;
        const int krow = 2 * (l16 & 3) + ((l16 >> 2) & 1);
        const uint64_t obase = isg ? (uint64_t)(uintptr_t)G : (uint64_t)(uintptr_t)A;
        const uint64_t odisp = isg ? dG : dA;
        const int32_t on1 = isg ? ng1 : na1;
        auto addr_pass = [&](int T) {
            uint64_t ad;
            if constexpr (GATHER) {
                int p0, pe;
                bool live;
                tile_rows(T, p0, pe, live);
                // rows past the chunk's end (and whole tiles past it): G reads the zero row, so the pair adds nothing; A repeats
                // the chunk's last row (the index was clamped), unless its column sums are wanted -- then it reads the zero row too
                const bool ok = live && p0 + krow <= pe;
                const int32_t r = idxL[q][T % kSlots][l16];
                ad = obase + (uint64_t)(uint32_t)r * ROWB + (r >= on1 ? odisp : 0ull);
                ad = (!ok && (isg || colsum_of == 1)) ? (uint64_t)(uintptr_t)zero : ad;
            } else {
                const int p = ch_beg + T * TR + 8 * q + krow;
                ad = p < ch_end ? obase + (uint64_t)(uint32_t)p * ROWB : (uint64_t)(uintptr_t)zero;
            }
            if (lane < 16) adrL[q][T % kSlots][l16] = ad;
        };
        auto issue = [&](int T) {                  // the row pieces of tile T (its addresses are in the ring)
            const unsigned st = lds_base + (unsigned)(T % NST) * STB;
            uint64_t pas[4], pgs[4];
            {
                typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                const u64x2* ap = reinterpret_cast<const u64x2*>(&adrL[q][T % kSlots][4 * rin]);
                const u64x2* gp = reinterpret_cast<const u64x2*>(&adrL[q][T % kSlots][8 + 4 * rin]);
                const u64x2 a01 = ap[0], a23 = ap[1], g01 = gp[0], g23 = gp[1];
                pas[0] = a01[0] + (uint64_t)gch[0]; pas[1] = a01[1] + (uint64_t)gch[1];
                pas[2] = a23[0] + (uint64_t)gch[2]; pas[3] = a23[1] + (uint64_t)gch[3];
                pgs[0] = g01[0] + (uint64_t)gch[0]; pgs[1] = g01[1] + (uint64_t)gch[1];
                pgs[2] = g23[0] + (uint64_t)gch[2]; pgs[3] = g23[1] + (uint64_t)gch[3];
            }
            if constexpr (MASKED) {
                // [A pieces] [the tile's mask bits: 32 rows x 32 bytes = ONE full-width piece, by loader T & 3] [G pieces]: masking a
                // tile needs its A pieces and bits only
#pragma unroll
                for (int j = 0; j < 4; ++j) glds16(reinterpret_cast<const char*>(pas[j]), st + (unsigned)(8 * q + 2 * j) * ROWB);
                if (q == (T & 3)) {
                    const int p = ch_beg + T * TR + lane / 2;
                    const char* pb = p >= ch_end ? zero + lane * 16
                                                 : reinterpret_cast<const char*>(maskBits) + (size_t)p * (H / 8) + (lane % 2) * 16;
                    glds16(pb, st + 2 * MATB);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) glds16(reinterpret_cast<const char*>(pgs[j]), st + (unsigned)(8 * q + 2 * j) * ROWB + MATB);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned da = st + (unsigned)(8 * q + 2 * j) * ROWB;  // wave-uniform; lane l lands at + 16 l
                    glds16(reinterpret_cast<const char*>(pas[j]), da);
                    glds16(reinterpret_cast<const char*>(pgs[j]), da + MATB);
                }
            }
        };
        // my four pieces of a landed tile (rows crow, crow + 16 for crow = lt / 32 and 8 + lt / 32): LDS byte offsets and, for the
        // masked operand, the 8 columns a piece holds (the image is swizzled: position q of row r holds chunk q ^ f(r))
        const int crowL = 2 * q + rin;             // = lt / 32
        int coff[2], mchunk[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int r = crowL + 8 * s;
            const int f = (r & 3) | (((r >> 3) & 1) << 2);
            const int pos = (((cpos >> 1) ^ f) << 1) | (cpos & 1);
            coff[s] = r * ROWB + pos * 16;                                  // column sums: chunk cpos of rows r, r + 16
            mchunk[s] = pos;                                               // mask pass: position cpos of rows r, r + 16 holds chunk `pos`
            ochunk[s] = (MASKED && colsum_of == 1) ? pos : cpos;
        }
        auto add8 = [&](float* c, const uint4& v) {   // (pairs: one v_pk_add_f32 per dword)
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x2 a = {c[2 * i], c[2 * i + 1]}, b = {__uint_as_float(w[i] << 16), __uint_as_float(w[i] & 0xffff0000u)};
                const f32x2 r = a + b;
                c[2 * i] = r[0];
                c[2 * i + 1] = r[1];
            }
        };
        auto mask_tile = [&](int T) {              // the A rows of tile T, in place (+ their column sums, from the registers they pass through)
            char* sT = lds + (T % NST) * STB;
            const uint8_t* sB = reinterpret_cast<const uint8_t*>(sT + 2 * MATB);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int r = crowL + 8 * s + 16 * j;
                    uint4* pp = reinterpret_cast<uint4*>(sT + r * ROWB + cpos * 16);
                    const uint4 v = dn_keep_or_scale_bits(*pp, sB[r * (H / 8) + mchunk[s]], slope);
                    *pp = v;
                    if (colsum_of == 1) add8(cs[s], v);
                }
        };
        auto colsum_tile = [&](int T) {
            if (colsum_of == 0 || (MASKED && colsum_of == 1)) return;
            const char* M = lds + (T % NST) * STB + (colsum_of == 1 ? 0 : MATB);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 2; ++j) add8(cs[s], *reinterpret_cast<const uint4*>(M + coff[s] + 16 * j * ROWB));
        };

        // ---- prologue: three tiles in flight ----
        if constexpr (GATHER) {
#pragma unroll 1
            for (int T = 0; T < 4; ++T) dma_idx(T);
            wait_vmcnt<0>();
        }
#pragma unroll 1
        for (int T = 0; T < 4; ++T) addr_pass(T);
#pragma unroll 1
        for (int s = -3; s < 0; ++s) {
            dma_idx(s + 7);
            issue(s + 3);
        }
        if constexpr (MASKED) {
            wait_vmcnt<4 + 2 * kGroup>();                                  // my A pieces (and the bits) of tile 0 have landed
            __builtin_amdgcn_s_barrier();
            mask_tile(0);
        }
        unsigned long long st_vm = 0, st_bar = 0, st_is = 0, st_cs = 0;
        const unsigned long long l0 = DN_WG_STAMP();
#pragma unroll 1
        for (int t = 0; t < ntiles; ++t) {
            const unsigned long long a0 = DN_WG_STAMP();
            // MODE 2: all of tile t and the A pieces + bits of tile t + 1 (its G pieces and tile t + 2 stay in flight)
            if constexpr (MASKED) wait_vmcnt<4 + kGroup>();
            else wait_vmcnt<2 * kGroup>();                                 // my pieces of tile t (and the octet of tile t + 3) have landed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // my LDS reads / mask writes of the last interval are done
            const unsigned long long a1 = DN_WG_STAMP();
            __builtin_amdgcn_s_barrier();                                  // everyone's have; the stage of tile t - 1 is free
            const unsigned long long a2 = DN_WG_STAMP();
            dma_idx(t + 7);
            issue(t + 3);
            addr_pass(t + 4);                                              // (its index octet came with the group of barrier t - 3)
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long a3 = DN_WG_STAMP();
            if constexpr (MASKED) mask_tile(t + 1);                        // (a zero tile past the end stays zero)
            colsum_tile(t);
            __builtin_amdgcn_sched_barrier(0);
            DN_WG_STAT(st_vm, a1 - a0); DN_WG_STAT(st_bar, a2 - a1); DN_WG_STAT(st_is, a3 - a2); DN_WG_STAT(st_cs, DN_WG_STAMP() - a3);
        }
#ifdef DN_WG_STATS
        if (q == 0 && lane == 0 && blockIdx.x < 256) {
            unsigned long long* o = g_wg_stats[blockIdx.x][1];
            o[0] = DN_WG_STAMP() - l0; o[1] = st_vm; o[2] = st_bar; o[3] = st_is; o[4] = st_cs;
        }
#endif
        wait_vmcnt<0>();                                                   // drain the zero tiles still in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (colsum_all != 0) {
            float* red = reinterpret_cast<float*>(lds);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 8; ++i) red[(crowL + 8 * s) * H + ochunk[s] * 8 + i] = cs[s][i];
        }
    } else {
        // ------------------------------------------------------------------------------------------------ compute
        const int wm = wave >> 2, wn = wave & 3;
        f32x4 acc[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int fg = lane >> 4, fq = (lane & 15) >> 2, fp = lane & 3;
        const int fsw = fq | ((fg & 1) << 2);
        const int frow = (8 * fg + fq) * ROWB + 8 * fp;
        // LDS byte address of this lane's part of fragment `slot` = stage | row (bits 9+) | (slot ^ fsw) << 5 | 8 fp: bit fields, so
        // the address of slot s0 + m is the address of slot s0 XOR m << 5 -- one register per operand instead of one per fragment
        const unsigned a_lane = lds_base + (unsigned)(frow + (((wm * MT) ^ fsw) << 5));
        const unsigned b_lane = lds_base + (unsigned)(MATB + frow + (((wn * NT) ^ fsw) << 5));
        // Fragment reads by inline asm with hand-counted waits (hipcc serialises them: one fragment, lgkmcnt(0), four MFMAs): a
        // tile's four G fragments and three A fragments go out behind the barrier, row m of the MFMAs waits for ITS fragment only
        // and the A fragment three rows on is requested right behind it.
        struct Frag { short4v lo, hi; };
#define DN_TR_READ(fr, addr) do {                                                                                   \
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"((fr).lo) : "v"(addr));                                  \
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"((fr).hi) : "v"(addr));                      \
        } while (0)
        static_assert(4 * ROWB == 2048, "offset of the fragment's second half");
        auto val = [](const Frag& f) -> bf16x8 {
            const short8v v = {f.lo[0], f.lo[1], f.lo[2], f.lo[3], f.hi[0], f.hi[1], f.hi[2], f.hi[3]};
            return __builtin_bit_cast(bf16x8, v);
        };
        if constexpr (MASKED) __builtin_amdgcn_s_barrier();                // (the loaders' barrier in front of mask_tile(0))
        unsigned long long sc_bar = 0;
        const unsigned long long c_l0 = DN_WG_STAMP();
#ifdef DN_WG_STATS
        const unsigned long long c_rt0 = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll 1
        for (int t = 0; t < ntiles; ++t) {
            const unsigned va = a_lane + (unsigned)(t % NST) * STB, vb = b_lane + (unsigned)(t % NST) * STB;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long c0 = DN_WG_STAMP();
            __builtin_amdgcn_s_barrier();                                  // tile t is in LDS (and masked)
            DN_WG_STAT(sc_bar, DN_WG_STAMP() - c0);
            Frag fb[NT], fa[3];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const unsigned ad = vb ^ (unsigned)(n << 5);
                DN_TR_READ(fb[n], ad);
            }
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const unsigned ad = va ^ (unsigned)(m << 5);
                DN_TR_READ(fa[m], ad);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                Frag& f = fa[m % 3];
                // reads requested behind fragment m: the next two A fragments (while there are any)
                if (m == 0)
                    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(f.lo), "+v"(f.hi), "+v"(fb[0].lo), "+v"(fb[0].hi), "+v"(fb[1].lo), "+v"(fb[1].hi),
                                 "+v"(fb[2].lo), "+v"(fb[2].hi), "+v"(fb[3].lo), "+v"(fb[3].hi));
                else if (m < MT - 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(f.lo), "+v"(f.hi));
                else if (m == MT - 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(f.lo), "+v"(f.hi));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.lo), "+v"(f.hi));
                const bf16x8 av = val(f);
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, val(fb[n]), acc[m][n], 0, 0, 0);
                if (m + 3 < MT) {
                    const unsigned ad = va ^ (unsigned)((m + 3) << 5);
                    DN_TR_READ(f, ad);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#undef DN_TR_READ
#ifdef DN_WG_STATS
        if (wave == 0 && lane == 0 && blockIdx.x < 256) {
            unsigned long long* o = g_wg_stats[blockIdx.x][0];
            o[0] = DN_WG_STAMP() - c_l0; o[1] = sc_bar; o[2] = 0; o[3] = 0; o[4] = __builtin_amdgcn_s_memrealtime() - c_rt0;
        }
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float* out = partial + (size_t)blockIdx.x * H * H;
        const int k0 = wm * (H / 2), n0 = wn * (H / 4);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = k0 + m * 16 + (lane >> 4) * 4 + i, c = n0 + n * 16 + (lane & 15);
                    out[(size_t)k * H + c] = acc[m][n][i];
                }
    }
    if (colsum_all != 0) {
        __syncthreads();
        const float* red = reinterpret_cast<const float*>(lds);
        if (tid < H) {
            float sum = 0.f;
            for (int sl = 0; sl < 16; ++sl) sum += red[sl * H + tid];
            colsum_partial[(size_t)blockIdx.x * H + tid] = sum;
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(kLsThreads) void rows_wgrad_ls_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ A2,
                                                                   int32_t na1, const int32_t* __restrict__ ia,
                                                                   const bf16_t* __restrict__ G, const bf16_t* __restrict__ G2,
                                                                   int32_t ng1, const int32_t* __restrict__ ig,
                                                                   const Chunk* __restrict__ chunks, float* __restrict__ partial,
                                                                   int32_t colsum_arg, float* __restrict__ colsum_partial,
                                                                   const uint8_t* __restrict__ maskBits, float slope,
                                                                   const int32_t* __restrict__ chunk_ptr) {
    __shared__ __attribute__((aligned(1024))) char lds[kLsStages * (kLsStageBytes + (MODE == 2 ? 1024 : 0))];
    __shared__ __attribute__((aligned(64))) int32_t idxL[MODE == 0 ? kLsLoaders : 1][kLsSlots][16];
    __shared__ __attribute__((aligned(128))) uint64_t adrL[kLsLoaders][kLsSlots][16];   // per loader and tile: the 16 row addresses
    const Chunk ch = chunks[blockIdx.x];
    wgrad_ls_body<MODE>(A, A2, na1, ia, G, G2, ng1, ig, chunks, ch, ch.rel, 0, partial, colsum_arg, colsum_partial, maskBits, slope,
                        chunk_ptr, lds, idxL, adrL);
}

// Several weight gradients of H = 256 in ONE launch (rows_wgrad_multi_kernel's contract, above): the conv's gathered rows (MODE 0),
// Linear rows in row order (MODE 1) and with the activation mask as bits (MODE 2) -- a workgroup takes the body of its chunk's
// job.  Why at this width, where no launch is latency: the Linears' gradients alone are HBM-bound (1 GB of rows that no one reads
// twice, 229-249 us at config 5) and the conv's MFMA-bound on rows the L2s serve (431 us); side by side in one launch the dense
// rows stream under the conv's matrix work, and the launch writes ONE round of partial tiles instead of three.
__global__ __launch_bounds__(kLsThreads) void rows_wgrad_ls_multi_kernel(WgJobs jobs, const Chunk* __restrict__ chunks,
                                                                         float* __restrict__ partial, float* __restrict__ colsum_partial,
                                                                         const int32_t* __restrict__ chunk_ptr) {
    __shared__ __attribute__((aligned(1024))) char lds[kLsStages * (kLsStageBytes + 1024)];
    __shared__ __attribute__((aligned(64))) int32_t idxL[kLsLoaders][kLsSlots][16];
    __shared__ __attribute__((aligned(128))) uint64_t adrL[kLsLoaders][kLsSlots][16];
    Chunk ch = chunks[blockIdx.x];
    int k = 0;
    if (jobs.n > 1 && ch.rel >= jobs.j[1].first_rel) k = 1;
    if (jobs.n > 2 && ch.rel >= jobs.j[2].first_rel) k = 2;
    const WgJob J = k == 0 ? jobs.j[0] : (k == 1 ? jobs.j[1] : jobs.j[2]);
    const int32_t vrel = ch.rel;
    ch.beg -= J.row0; ch.end -= J.row0; ch.rel -= J.first_rel;
    if (J.ia != nullptr)
        wgrad_ls_body<0>(J.A, J.A2, J.na1, J.ia, J.G, J.G2, J.ng1, J.ig, chunks, ch, vrel, J.row0, partial, J.colsum_of, colsum_partial,
                         nullptr, 0.f, chunk_ptr, lds, idxL, adrL);
    else if (J.maskBits != nullptr)
        wgrad_ls_body<2>(J.A, nullptr, 0x7fffffff, nullptr, J.G, nullptr, 0x7fffffff, nullptr, chunks, ch, vrel, J.row0, partial,
                         J.colsum_of, colsum_partial, J.maskBits, J.slope, nullptr, lds, idxL, adrL);
    else
        wgrad_ls_body<1>(J.A, nullptr, 0x7fffffff, nullptr, J.G, nullptr, 0x7fffffff, nullptr, chunks, ch, vrel, J.row0, partial,
                         J.colsum_of, colsum_partial, nullptr, 0.f, nullptr, lds, idxL, adrL);
}

// -------------------------------------------------------------------------------------------------
// dn_rows_transform_bf16:   Y[p, :] = epi( X[idx[p], :] @ Wn[rel(p)]^T )          (rows p relation-major)
//   Wn[r] is [HO][HI] (k contiguous), i.e. Y[p][n] = sum_k X[idx[p]][k] * Wn[r][n][k].
//   One workgroup (4 waves) walks a contiguous range of 32-row tiles (tile table: {rel, beg, end}); each wave
//   owns a 64-column slice of the output and keeps ITS slice of Wn[rel] in registers (128 VGPRs), reloading it
//   only when the relation changes.  Per tile: the 32 gathered rows (16 B per lane, 32 lanes per row) are staged
//   registers -> LDS one tile ahead (the gather of tile t+1 is in flight under tile t's MFMAs, its row indices
//   two tiles ahead), fragments come from LDS with ds_read_b128 (rows padded by 16 B: conflict-free), the MFMA
//   computes the TRANSPOSED tile (A operand = weights, B operand = rows) so that each lane ends up with 4
//   consecutive output columns of one row, and the finished tile goes through LDS to be written as whole
//   512-byte rows.
// -------------------------------------------------------------------------------------------------
// (An LDS-DMA ring version of this kernel -- 3 or 4 16 KB stages in flight per workgroup instead of one register-staged tile,
//  96 KB per CU -- ran at the same 390-397 us per config-5 launch as this one: the launch moves 1.03 GB of gathered rows in and
//  1.07 GB of Y rows out at 5.4 TB/s combined, which is what a mixed read/write stream reaches on this part (a device-to-device
//  copy: 5.2 TB/s, torch add: 6.2), not a bytes-in-flight limit.  Not kept.)
constexpr int kTfThreads = 512;   // 8 waves, each owning HO/8 output columns
constexpr int kTfRows = 32;

template <int HI, int HO, int D>
__global__ __launch_bounds__(kTfThreads, (D == 1 ? 4 : 2)) void rows_transform_kernel(
    const bf16_t* __restrict__ X, const bf16_t* __restrict__ X2, int32_t n1, const int32_t* __restrict__ idx,
    const bf16_t* __restrict__ Wn, const bf16_t* __restrict__ bias, int32_t relu, float slope, const bf16_t* __restrict__ mask_pos,
    const Chunk* __restrict__ tiles, int32_t num_tiles, int32_t tiles_per_wg, bf16_t* __restrict__ Y) {
    const bool nt_store = (relu & 2) != 0;          // bit 1: streaming (non-temporal) stores of Y
    const bool sc1_store = (relu & 4) != 0;         // bit 2: sc1 stores (the written line does not stay in the XCD's L2)
#ifdef DN_TUNING_ENV
    const bool abl_nostore = (relu & 8) != 0, abl_hit = (relu & 16) != 0;   // tuning build only: ablations
#else
    constexpr bool abl_nostore = false, abl_hit = false;
#endif
    const bool w_kn = (relu & 32) != 0;             // bit 5: Wn[r] stored [k][n] (the parameter's own layout; NT == 1 widths only)
    relu &= 1;
    constexpr int SX = HI + kPad;                   // LDS row stride (elements) of the input tile
    constexpr int SY = HO + kPad;                   // ... of the output tile
    constexpr int KS = HI / 32;                     // k-steps
    constexpr int NT = (HO / 8 + 15) / 16;          // 16-col tiles per wave (waves beyond HO/16 idle in the MFMA part)
    constexpr int MT = kTfRows / 16;
    constexpr int NPX = kTfRows * HI / 8;           // 16-byte pieces of an input tile
    constexpr int PX = (NPX + kTfThreads - 1) / kTfThreads;
    constexpr int NPY = kTfRows * HO / 8;
    constexpr int PY = (NPY + kTfThreads - 1) / kTfThreads;
    static_assert(NT >= 1 && KS >= 1, "unsupported width");
    __shared__ __attribute__((aligned(16))) bf16_t lds[2 * kTfRows * SX + kTfRows * SY + HO];
    __shared__ __attribute__((aligned(16))) char wscr[NT == 1 ? (HO / 16) * 1024 : 16];   // transposition scratch of a [k][n] W, 1 KB per active wave
    auto bufX = [&](int b) -> bf16_t* { return lds + b * (kTfRows * SX); };
    bf16_t* bufY = lds + 2 * kTfRows * SX;
    bf16_t* biasL = lds + 2 * kTfRows * SX + kTfRows * SY;   // bias[rel]: each wave keeps ITS column slice here (no barrier)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = wave * (NT * 16);
    const bool wave_active = n0 < HO;
    const int t_beg = blockIdx.x * tiles_per_wg;
    const int t_end = min(t_beg + tiles_per_wg, num_tiles);
    if (t_beg >= t_end) return;

    bf16x8 wf[KS][NT];                              // this wave's slice of Wn[rel]: A operand fragments
    int cur_rel = -1;
    int32_t nidx[PX];                               // source row of each of my pieces, for the tile being loaded
    uint4 rx[D][PX];                                // D gathered tiles in flight (tile k lives in slot k % D)

    auto load_idx = [&](int t) {
        if (t >= t_end) return;
        const Chunk tl = tiles[t];
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const int piece = tid + j * kTfThreads, r = piece / (HI / 8);
            const int p = tl.beg + r;
            nidx[j] = -1;
            if (piece < NPX && p < tl.end) nidx[j] = idx ? idx[p] : p;
            if (abl_hit && nidx[j] >= 0) nidx[j] &= 1023;
        }
    };
    auto load_rows = [&](uint4 (&r)[PX]) {
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const int piece = tid + j * kTfThreads, c = piece % (HI / 8);
            r[j] = make_uint4(0, 0, 0, 0);
            if (nidx[j] >= 0) {
                const bf16_t* base = nidx[j] < n1 ? X + (size_t)nidx[j] * HI : X2 + (size_t)(nidx[j] - n1) * HI;
                r[j] = *reinterpret_cast<const uint4*>(base + c * 8);
            }
        }
    };
    auto store_rows = [&](int b, const uint4 (&r)[PX]) {
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const int piece = tid + j * kTfThreads, rr = piece / (HI / 8), c = piece % (HI / 8);
            if (piece < NPX) *reinterpret_cast<uint4*>(bufX(b) + rr * SX + c * 8) = r[j];
        }
    };

    // prologue: tiles 0 .. D-1 in flight, tile 0 in LDS, indices of tile D fetched
#pragma unroll
    for (int u = 0; u < D; ++u) {
        load_idx(t_beg + u);
        if (t_beg + u < t_end) load_rows(rx[u]);
    }
    store_rows(0, rx[0]);
    load_idx(t_beg + D);
    __syncthreads();

    for (int t0 = t_beg; t0 < t_end; t0 += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
        const int t = t0 + u;
        if (t >= t_end) break;
        const int b = (t - t_beg) & 1;
        const Chunk tl = tiles[t];
        if (t + D < t_end) load_rows(rx[u]);        // gather of tile t+D into the slot tile t just left
        if (tl.rel != cur_rel && wave_active) {     // (re)load this wave's weight slice: wave-uniform branch
            cur_rel = tl.rel;
            const bf16_t* w = Wn + (size_t)cur_rel * HO * HI;
            bool done = false;
            if constexpr (NT == 1) {
                if (w_kn) {                             // [k = HI][n = HO]: 16-byte pieces along n, transposed through the wave's scratch
                    bf16x8 wl[KS];
                    dn_load_w_kn16<KS>(w, HO, n0, lane, wscr + (n0 / 16) * 1024, wl);
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) wf[ks][0] = wl[ks];
                    done = true;
                }
            }
            if (!done) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    wf[ks][nt] = *reinterpret_cast<const bf16x8*>(w + (size_t)(n0 + nt * 16 + (lane & 15)) * HI + ks * 32 +
                                                                  8 * (lane >> 4));
            }
            if (bias && lane < NT * 16) biasL[n0 + lane] = bias[(size_t)cur_rel * HO + n0 + lane];
            // pin the wait for the new weights INSIDE this (rare) branch: left to the compiler it sits in front of the MFMAs on
            // the common path as vmcnt(15) ... vmcnt(0), i.e. every tile waits there for the gather issued just above
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) asm volatile("" : "+v"(wf[ks][nt]));
        }
        f32x4 acc[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bf16_t* xt = bufX(b);
        if (wave_active) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 xf[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m)
                xf[m] = *reinterpret_cast<const bf16x8*>(xt + (m * 16 + (lane & 15)) * SX + ks * 32 + 8 * (lane >> 4));
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][n], xf[m], acc[m][n], 0, 0, 0);
        }
        // D = Wn_slice x rows^T : lane holds row m = mt*16 + (lane&15), columns n0 + nt*16 + 4*(lane>>4) + i
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int col = n0 + n * 16 + 4 * (lane >> 4);
                float v[4] = {acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]};
                if (bias) {
                    typedef bf16_t bf16x4b __attribute__((ext_vector_type(4)));
                    const bf16x4b bv = *reinterpret_cast<const bf16x4b*>(biasL + col);   // written by this wave: in order
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += (float)bv[i];
                }
                if (relu) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = dn_act(v[i], slope);
                }
                typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));
                bf16x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (bf16_t)v[i];
                *reinterpret_cast<bf16x4*>(bufY + (m * 16 + (lane & 15)) * SY + col) = o;
            }
        }
        if (t + 1 < t_end) store_rows(b ^ 1, rx[(u + 1) % D]);
        load_idx(t + D + 1);
        __syncthreads();
        // whole rows out: 16 B per lane, HO/8 lanes per row
#pragma unroll
        for (int j = 0; j < PY; ++j) {
            const int piece = tid + j * kTfThreads, r = piece / (HO / 8), c = piece % (HO / 8);
            const int p = tl.beg + r;
            if (piece < NPY && p < tl.end && !(abl_nostore && p != 0)) {
                uint4 v = *reinterpret_cast<const uint4*>(bufY + r * SY + c * 8);
                if (mask_pos)                                    // activation backward: keep where the saved activation is > 0
                    v = dn_keep_or_scale_mask(v, *reinterpret_cast<const uint4*>(mask_pos + (size_t)p * HO + c * 8), slope);
                if (sc1_store) {
                    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 vv = {v.x, v.y, v.z, v.w};
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(Y + (size_t)p * HO + c * 8), "v"(vv) : "memory");
                }
                else if (nt_store) {
                    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 vv = {v.x, v.y, v.z, v.w};
                    __builtin_nontemporal_store(vv, reinterpret_cast<u32x4*>(Y + (size_t)p * HO + c * 8));
                }
                else *reinterpret_cast<uint4*>(Y + (size_t)p * HO + c * 8) = v;
            }
        }
        // bufY is rewritten only after the next tile's MFMAs and its barrier-separated store: add a barrier here so no
        // wave overwrites bufY (next iteration's epilogue) while a slower wave still reads it
        __syncthreads();
    }
    }
}

// -------------------------------------------------------------------------------------------------
// dn_rows_selfsum_bf16:  out[v, :] = X[v, :] @ Wn^T (+ bias)  +  sum_{k < K} Scat[slots[v][k], :]     (slots < 0: empty)
//   The closing launch of the row-factorised message pass: the self-loop transform of node v (rgin.py:140-142) and the
//   per-node sum of the transformed edge rows that point at v (the reference's fn.sum reduce) in ONE pass, so the
//   self-loop products never make the round trip through HBM and no separate per-node gather launch is needed.
//   Every node has a fixed number of slots (K = 6); nodes with more incoming rows get their excess pre-summed into an
//   overflow row (Scat = S followed by S2 at row n1) by the caller, so the kernel has no data-dependent loop.
//   One workgroup = H/16 waves (1024 threads at H = 256), wave w owns 16 output columns with its slice of Wn in
//   registers for the whole launch; a tile = 32 consecutive nodes, one 16-byte piece per thread.  Per tile: the K slot
//   rows of my piece are requested first (their ids were read one tile ahead), then the next tile's X piece and slot
//   ids, then the MFMAs run on the LDS image of this tile, and the epilogue adds bias tile + slot rows in fp32.
// -------------------------------------------------------------------------------------------------
constexpr int kSsRows = 32;
constexpr int kSsSlots = 6;
constexpr int kFoldInfo = 12;   // int32 words per tile of the fold table (dn_fold_tables_build_i32)

// DN_STRIDE (default 1): dense-row launches deal tiles round-robin over the workgroups instead of contiguous ranges
static bool stride_tiles() {
    static const bool on = dn_knob("DN_STRIDE", 1) != 0;
    return on;
}

template <int H, bool FOLD>
__global__ __launch_bounds__(H * 4) void rows_selfsum_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ Wn,
                                                             const bf16_t* __restrict__ bias, const bf16_t* __restrict__ S,
                                                             const bf16_t* __restrict__ S2, int32_t n1,
                                                             const int32_t* __restrict__ slots, int32_t N,
                                                             int32_t num_tiles, int32_t tiles_per_wg,
                                                             bf16_t* __restrict__ out, int32_t nt,
                                                             const int32_t* __restrict__ fold_info,
                                                             float* __restrict__ seg_part, const int32_t* __restrict__ lptr,
                                                             const int32_t* __restrict__ lrows, int32_t P_edge, int32_t drop_beg,
                                                             int32_t drop_end) {
    constexpr int T = H * 4, K = kSsSlots;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    constexpr int SX = H + kPad, SY = H + kPad;
    constexpr int KS = H / 32, MT = kSsRows / 16;
    constexpr int LPR = H / 8;                                   // 16-byte pieces per row
    static_assert(kSsRows * LPR == T, "one piece per thread");
    __shared__ __attribute__((aligned(16))) bf16_t lds[2 * kSsRows * SX + kSsRows * SY];
    __shared__ __attribute__((aligned(16))) float biasL[H];
    __shared__ int2 slotL[3][T];                                             // the next tile's slot ids, parked between their
                                                                             // arrival and the top of the next iteration
    __shared__ __attribute__((aligned(16))) int32_t segI[2][kFoldInfo];      // FOLD: the tile's record (see dn_hip.h): 32 local
                                                                             // partial-row ids (bytes), first partial row, count
    auto bufX = [&](int b) -> bf16_t* { return lds + b * (kSsRows * SX); };
    bf16_t* bufY = lds + 2 * kSsRows * SX;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = wave * 16;
    const int pr = tid / LPR, pc = tid % LPR;                    // my piece: row of the tile, 16-byte column chunk
    // tiles dealt round-robin when tiles_per_wg <= 0 (see rows_chain2_kernel), contiguous ranges otherwise
    const bool rr = tiles_per_wg <= 0;
    const int first = rr ? (int)blockIdx.x : (int)blockIdx.x * tiles_per_wg, step = rr ? (int)gridDim.x : 1;
    const int t_beg = 0;
    const int t_end = rr ? ((first < num_tiles) ? (num_tiles - first + step - 1) / step : 0)
                         : max(min(first + tiles_per_wg, num_tiles) - first, 0);
    if (t_beg >= t_end) return;
    auto rowbase = [&](int t) -> int { return (first + t * step) * kSsRows; };

    bf16x8 wf[KS];                                               // A operand: Wn rows n0 + (lane & 15), k = ks*32 + 8*(lane>>4)
    if (nt & 2) {                                                // Wn stored [k][n] (`loop_weight` as it is): transposed through LDS
        dn_load_w_kn16<KS>(Wn, H, n0, lane, reinterpret_cast<char*>(lds) + wave * 1024, wf);
        __syncthreads();                                         // (lds holds the tiles from here on)
    } else {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            wf[ks] = *reinterpret_cast<const bf16x8*>(Wn + (size_t)(n0 + (lane & 15)) * H + ks * 32 + 8 * (lane >> 4));
    }
    nt &= 1;
    if (tid < H) biasL[tid] = bias ? (float)bias[tid] : 0.f;     // read back per tile: 4 fewer registers across the loop

    auto load_x = [&](int t) -> uint4 {
        const int p = rowbase(t) + pr;
        return (t < t_end && p < N) ? *reinterpret_cast<const uint4*>(X + (size_t)p * H + pc * 8) : make_uint4(0, 0, 0, 0);
    };
    struct Slots { int2 a, b, c; };                              // K = 6 ids, 24 bytes per node (8-byte aligned)
    auto load_slots = [&](int t) -> Slots {
        const int p = rowbase(t) + pr;
        Slots v = {make_int2(-1, -1), make_int2(-1, -1), make_int2(-1, -1)};
        if (t < t_end && p < N) {
            const int2* q = reinterpret_cast<const int2*>(slots + (size_t)p * K);
            v.a = q[0]; v.b = q[1]; v.c = q[2];
        }
        return v;
    };

    auto load_seg = [&](int t) -> int32_t {                       // thread i < kFoldInfo: word i of tile t's fold record
        return (FOLD && tid < kFoldInfo && t < t_end) ? fold_info[(size_t)(first + t * step) * kFoldInfo + tid] : 0;
    };
    auto store_seg = [&](int b, int32_t v) {
        if (tid < kFoldInfo) segI[b][tid] = v;
    };
    uint4 rx = load_x(t_beg);
    {
        const Slots s0 = load_slots(t_beg);
        slotL[0][tid] = s0.a; slotL[1][tid] = s0.b; slotL[2][tid] = s0.c;
    }
    *reinterpret_cast<uint4*>(bufX(0) + pr * SX + pc * 8) = rx;
    if (FOLD) store_seg(0, load_seg(t_beg));
    rx = load_x(t_beg + 1);
    int32_t sg_next = load_seg(t_beg + 1);
    __syncthreads();

    for (int t = t_beg; t < t_end; ++t) {
        const int b = (t - t_beg) & 1;
        // (1) the slot rows of my piece
        uint4 g[K];
        const int2 sa = slotL[0][tid], sb_ = slotL[1][tid], sc = slotL[2][tid];     // (my own entries: no barrier needed)
        const int sid[K] = {sa.x, sa.y, sb_.x, sb_.y, sc.x, sc.y};
#pragma unroll
        for (int k = 0; k < K; ++k) {
            g[k] = make_uint4(0, 0, 0, 0);
            if (sid[k] >= 0) {
                const bf16_t* base = sid[k] < n1 ? S + (size_t)sid[k] * H : S2 + (size_t)(sid[k] - n1) * H;
                g[k] = *reinterpret_cast<const uint4*>(base + pc * 8);
            }
        }
        // (2) next tile's slot ids (its X piece is already in flight / in rx)
        const Slots sl_next = load_slots(t + 1);
        // (3) MFMAs: D = Wn_slice x rows^T, lane holds row m*16 + (lane&15), columns n0 + 4*(lane>>4) + i
        f32x4 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = *reinterpret_cast<const f32x4*>(biasL + n0 + 4 * (lane >> 4));
        const bf16_t* xt = bufX(b);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xt + (m * 16 + (lane & 15)) * SX + ks * 32 + 8 * (lane >> 4));
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], xf, acc[m], 0, 0, 0);
            }
        }
        // the tile is staged in fp32 halves?  no: bf16, like the rows it is added to (both are rounded once)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));
            bf16x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (bf16_t)acc[m][i];
            *reinterpret_cast<bf16x4*>(bufY + (m * 16 + (lane & 15)) * SY + n0 + 4 * (lane >> 4)) = o;
        }
        *reinterpret_cast<uint4*>(bufX(b ^ 1) + pr * SX + pc * 8) = rx;          // (zeros past the last tile)
        slotL[0][tid] = sl_next.a; slotL[1][tid] = sl_next.b; slotL[2][tid] = sl_next.c;
        if (FOLD) store_seg(b ^ 1, sg_next);
        // (3b) column sums of this tile's X rows per SEGMENT (the rows a collapsed relation pre-aggregates: all nodes of a graph
        //      feeding its dummy node) -- the rows are in LDS anyway, so the separate pre-aggregation pass over X (0.5 GB per
        //      direction) disappears.  One more MFMA per wave: D[s][c] = sum_r [row r belongs to the tile's s-th segment] X[r][c]
        //      (0/1 indicator as the A operand, exact products, the matrix unit's fixed accumulation order), written as one fp32
        //      partial row per (segment, tile); the partial rows of a tile's segments are consecutive (segments are contiguous
        //      ascending node ranges).  A tail launch adds a segment's partials in tile order.
        if (FOLD) {
            const int p0 = segI[b][8], cnt = __builtin_amdgcn_readfirstlane(segI[b][9]);
            if (cnt > 0) {
                const uint2 sb = *reinterpret_cast<const uint2*>(&segI[b][2 * (lane >> 4)]);    // my 8 rows' local ids
                const bf16x8 xf = tr_frag(bufX(b), SX, n0, lane);  // element j: X[8*(lane>>4) + j][n0 + (lane & 15)]
                for (int m0 = 0; m0 < cnt; m0 += 16) {             // (more than 16 segments in 32 rows: graphs of 1-2 nodes)
                    const uint32_t me = (uint32_t)(m0 + (lane & 15));
                    bf16x8 ind;                                   // element j: [row 8*(lane>>4) + j is in segment m0 + (lane & 15)]
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        ind[j] = (bf16_t)(((((j < 4 ? sb.x : sb.y) >> (8 * (j & 3))) & 0xffu) == me) ? 1.0f : 0.0f);
                    // D[c][s] = sum_r X[r][n0 + c] ind[r][s]: lane holds columns n0 + 4*(lane>>4) + i of segment m0 + (lane&15)
                    const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, ind, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    if ((int)me < cnt)
                        *reinterpret_cast<f32x4*>(seg_part + (size_t)(p0 + (int)me) * H + n0 + 4 * (lane >> 4)) = d;
                }
            }
        }
        rx = load_x(t + 2);
        sg_next = load_seg(t + 2);
        __syncthreads();
        // (4) epilogue: my piece of the tile + my slot rows, fp32, one rounding
        const int p = rowbase(t) + pr;
        if (p < N) {
            const uint4 y = *reinterpret_cast<const uint4*>(bufY + pr * SY + pc * 8);
            float a[8];
            const uint32_t yw[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[2 * i] = __uint_as_float(yw[i] << 16);
                a[2 * i + 1] = __uint_as_float(yw[i] & 0xffff0000u);
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint32_t w[4] = {g[k].x, g[k].y, g[k].z, g[k].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a[2 * i] += __uint_as_float(w[i] << 16);
                    a[2 * i + 1] += __uint_as_float(w[i] & 0xffff0000u);
                }
            }
            if (lptr != nullptr && sid[K - 1] == -2) {
                // a node with more rows than slots, finished HERE (small batches: the separate dn_overflow_rows_add_bf16 launch costs
                // more than this walk -- dependent loads of a few nodes -- stalls; large ones keep the launch): the rows of its list
                // behind the first K - 1 kept ones, in list order, same filter as the table builder
                int kept = 0;
                for (int i = lptr[p]; i < lptr[p + 1]; ++i) {
                    const int r = lrows[i];
                    if (r < P_edge && !(r >= drop_beg && r < drop_end)) {
                        if (kept >= K - 1) {
                            const bf16_t* base = r < n1 ? S + (size_t)r * H : S2 + (size_t)(r - n1) * H;
                            const uint4 e = *reinterpret_cast<const uint4*>(base + pc * 8);
                            const uint32_t w[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                a[2 * q] += __uint_as_float(w[q] << 16);
                                a[2 * q + 1] += __uint_as_float(w[q] & 0xffff0000u);
                            }
                        }
                        ++kept;
                    }
                }
            }
            bf16x8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (bf16_t)a[i];
            if (nt & 1) __builtin_nontemporal_store(__builtin_bit_cast(u32x4, o), reinterpret_cast<u32x4*>(out + (size_t)p * H + pc * 8));
            else *reinterpret_cast<bf16x8*>(out + (size_t)p * H + pc * 8) = o;
        }
        __syncthreads();                                         // bufY is rewritten by the next tile's MFMA phase
    }
}

template <int H>
int launch_selfsum(const bf16_t* X, const bf16_t* Wn, const bf16_t* bias, const bf16_t* S, const bf16_t* S2, int32_t n1,
                   const int32_t* slots, int64_t N, bf16_t* out, const int32_t* fold_info, float* seg_part,
                   hipStream_t st, int32_t w_kn, const int32_t* lptr, const int32_t* lrows, int32_t P_edge, int32_t drop_beg,
                   int32_t drop_end) {
    const int64_t num_tiles = dn_cdiv(N, kSsRows);
    const int64_t tiles_per_wg = dn_cdiv(num_tiles, 256 * (1024 / (H * 4)));   // 16 waves per CU
    const int64_t grid = dn_cdiv(num_tiles, tiles_per_wg);
    static const int nt = dn_knob("DN_NT", 3);
    const int32_t tpw = (int32_t)(stride_tiles() ? 0 : tiles_per_wg);
    if (fold_info)
        hipLaunchKernelGGL((rows_selfsum_kernel<H, true>), dim3((unsigned)grid), dim3(H * 4), 0, st, X, Wn, bias, S, S2, n1, slots,
                           (int32_t)N, (int32_t)num_tiles, tpw, out, ((nt >> 1) & 1) | (w_kn ? 2 : 0), fold_info, seg_part, lptr, lrows, P_edge, drop_beg, drop_end);
    else
        hipLaunchKernelGGL((rows_selfsum_kernel<H, false>), dim3((unsigned)grid), dim3(H * 4), 0, st, X, Wn, bias, S, S2, n1, slots,
                           (int32_t)N, (int32_t)num_tiles, tpw, out, ((nt >> 1) & 1) | (w_kn ? 2 : 0), fold_info, seg_part, lptr, lrows, P_edge, drop_beg, drop_end);
    DN_CHECK_LAUNCH();
    return DN_OK;
}

// -------------------------------------------------------------------------------------------------
// dn_rows_chain2_bf16:  Y1 = epi1(m0(X) @ W1n^T),  Y2 = epi2(Y1 @ W2n^T)   -- two dense layers in one pass over the rows
//   m0   = zero the elements of X whose bit in mask0_bits is clear (optional)
//   epi1 = (+ b1) -> ReLU? -> zero where the bit in mask1_bits is clear (optional);   epi2 = (+ b2) -> ReLU?
//   bits1 / bits2 (optional outputs): 1 bit per element of Y1 / Y2, set where the element is > 0
//   Forward of the reference's two-layer post-aggregate MLP (rgin.py:50-57: Linear-ReLU-Linear, then the layer's ReLU),
//   emitting the two ReLU masks as BIT tensors (1/16 of the bf16 activation they replace in the backward), and -- with
//   mask0 = bits of the output activation, mask1 = bits of the hidden one, no bias/ReLU -- the whole input-gradient chain
//   of its backward (outer ReLU mask, dgrad 2, inner ReLU mask, dgrad 1).  Y1 is written (the weight gradient of layer 1 /
//   the backward need it) but never re-read.  H/16 waves, tile = 32 rows, one 16-byte piece per thread per tile.
// -------------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(H * 4) void rows_chain2_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W1n,
                                                            const bf16_t* __restrict__ b1, const bf16_t* __restrict__ W2n,
                                                            const bf16_t* __restrict__ b2, int32_t flags,
                                                            const uint8_t* __restrict__ mask0, const uint8_t* __restrict__ mask1,
                                                            int32_t N, int32_t num_tiles, int32_t tiles_per_wg,
                                                            bf16_t* __restrict__ Y1, bf16_t* __restrict__ Y2,
                                                            uint8_t* __restrict__ bits1, uint8_t* __restrict__ bits2, float slope) {
    // Wave-specialised two-stage pipeline: the first half of the waves owns layer 1 (32 output columns each, their slice of
    // W1n in 64 VGPRs), the second half layer 2; in one iteration layer 1 works on tile t while layer 2 works on tile t-1, so
    // the two products overlap and every wave re-reads the 32-row LDS tile for 32 columns instead of 16 (LDS fragment reads
    // were the limiter of the one-role-per-wave version: 16 waves x 16 KB per stage).
    constexpr int T = H * 4;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));
    constexpr int SX = H + kPad;
    constexpr int KS = H / 32, MT = kSsRows / 16, NT = 2;
    constexpr int LPR = H / 8;                                   // 16-byte pieces = mask bytes per row
    constexpr int HALF = H / 32;                                 // waves per role
    static_assert(kSsRows * LPR == T, "one piece per thread");
    __shared__ __attribute__((aligned(16))) bf16_t lds[6 * kSsRows * SX + 4 * H + kSsRows * LPR];
    auto bufX = [&](int b) -> bf16_t* { return lds + b * (kSsRows * SX); };
    auto buf1 = [&](int b) -> bf16_t* { return lds + (2 + b) * (kSsRows * SX); };   // stage-1 result = stage-2 input
    auto buf2 = [&](int b) -> bf16_t* { return lds + (4 + b) * (kSsRows * SX); };   // stage-2 result
    float* biasL = reinterpret_cast<float*>(lds + 6 * kSsRows * SX);                 // [2][H] fp32: b1, b2 (0 when absent)
    uint8_t* bitsL = reinterpret_cast<uint8_t*>(lds + 6 * kSsRows * SX + 4 * H);     // [2][32][LPR]: mask1 bits of a tile
    const bool nt = flags & 4;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int role = wave / HALF;                                // 0: layer 1, 1: layer 2 (wave-uniform)
    const int n0 = (wave % HALF) * 32;
    const int pr = tid / LPR, pc = tid % LPR;
    // tiles are dealt round-robin (tile = blockIdx.x + i * gridDim.x): at any moment the workgroups stream ADJACENT tiles, so
    // the launch sweeps HBM like one sequential stream instead of gridDim.x streams a fixed 2 MB apart (same channels)
    // (tiles_per_wg > 0 selects contiguous ranges instead: DN_STRIDE=0)
    const bool rr = tiles_per_wg <= 0;
    const int first = rr ? (int)blockIdx.x : (int)blockIdx.x * tiles_per_wg, step = rr ? (int)gridDim.x : 1;
    const int t_beg = 0;
    const int t_end = rr ? ((first < num_tiles) ? (num_tiles - first + step - 1) / step : 0)
                         : max(min(first + tiles_per_wg, num_tiles) - first, 0);
    if (t_beg >= t_end) return;
    auto rowbase = [&](int t) -> int { return (first + t * step) * kSsRows; };

    const bf16_t* Wn = role ? W2n : W1n;
    const bool relu = role ? (flags & 2) : (flags & 1);
    bf16x8 wf[KS][NT];
    if (flags & (role ? 16 : 8)) {                               // this layer's weights are stored [k][n] (a Linear's own weight seen
        // from the input-gradient side): transposed through a wave-private scratch in buf1 / buf2 (idle until the first barrier)
        dn_load_w_kn32p<KS>(Wn, H, n0, lane, reinterpret_cast<char*>(buf1(0)) + wave * 2048, wf);
    } else {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int n = 0; n < NT; ++n)
                wf[ks][n] = *reinterpret_cast<const bf16x8*>(Wn + (size_t)(n0 + n * 16 + (lane & 15)) * H + ks * 32 + 8 * (lane >> 4));
    }
    for (int i = threadIdx.x; i < 2 * H; i += T) {
        const bf16_t* bb = i < H ? b1 : b2;
        biasL[i] = bb ? (float)bb[i % H] : 0.f;
    }
    const float* myBias = biasL + role * H + n0 + 4 * (lane >> 4);   // visible after the first barrier below

    auto load_x = [&](int t) -> uint4 {
        const int p = rowbase(t) + pr;
        return (t < t_end && p < N) ? *reinterpret_cast<const uint4*>(X + (size_t)p * H + pc * 8) : make_uint4(0, 0, 0, 0);
    };
    auto load_bits = [&](const uint8_t* m, int t) -> uint32_t {  // my piece's 8 mask bits (0xff when there is no mask)
        const int p = rowbase(t) + pr;
        return (m && t < t_end && p < N) ? (uint32_t)m[(size_t)p * LPR + pc] : 0xffu;
    };
    // one dense stage on the LDS tile `src`: D = W_slice x rows^T -> bf16 tile `dst` (my 32 columns of all 32 rows)
    const int kbase = (lane & 15) * LPR + (n0 >> 3) + (lane >> 5), kshift = ((lane >> 4) & 1) * 4;
    auto stage = [&](const bf16_t* src, bf16_t* dst, const uint8_t* keep) {   // keep: the tile's mask bits in LDS, or NULL
        f32x4 acc[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = *reinterpret_cast<const f32x4*>(myBias + n * 16);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const bf16x8 xf = *reinterpret_cast<const bf16x8*>(src + (m * 16 + (lane & 15)) * SX + ks * 32 + 8 * (lane >> 4));
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][n], xf, acc[m][n], 0, 0, 0);
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int row = m * 16 + (lane & 15), col = n0 + n * 16 + 4 * (lane >> 4);
                uint32_t nib = 0xfu;                             // ReLU-backward mask of my 4 columns of this row
                if (keep) nib = (uint32_t)keep[kbase + m * 16 * LPR + n * 2] >> kshift;
                bf16x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = relu ? dn_act(acc[m][n][i], slope) : acc[m][n][i];
                    o[i] = (bf16_t)(((nib >> i) & 1u) ? v : dn_neg(v, slope));
                }
                *reinterpret_cast<bf16x4*>(dst + row * SX + col) = o;
            }
    };
    auto put = [&](bf16_t* Yout, uint8_t* bout, int p, const uint4& v) {
        const u32x4 vv = {v.x, v.y, v.z, v.w};
        if (nt) __builtin_nontemporal_store(vv, reinterpret_cast<u32x4*>(Yout + (size_t)p * H + pc * 8));
        else *reinterpret_cast<u32x4*>(Yout + (size_t)p * H + pc * 8) = vv;
        if (bout) bout[(size_t)p * LPR + pc] = (uint8_t)positive_bits(v);
    };

    // X pieces (and their mask bits) and the stage-1 mask bits one tile ahead (in flight across the whole iteration)
    uint4 rx = dn_keep_or_scale_bits(load_x(t_beg), load_bits(mask0, t_beg), slope);
    *reinterpret_cast<uint4*>(bufX(0) + pr * SX + pc * 8) = rx;
    bitsL[pr * LPR + pc] = (uint8_t)load_bits(mask1, t_beg);
    rx = load_x(t_beg + 1);
    uint32_t m0 = load_bits(mask0, t_beg + 1);
    uint32_t mk1 = load_bits(mask1, t_beg + 1);
    __syncthreads();

    // Iteration t: (1) the finished rows of the previous iteration leave for HBM FIRST (Y1 of tile t-1 from buf1, Y2 of tile
    // t-2 from buf2), so those stores drain under this iteration's MFMAs instead of in front of the next wait for a loaded
    // value (hipcc waits vmcnt(0) there, which includes every earlier store); (2) layer 1 on tile t -- its epilogue applies
    // the ReLU-backward mask from the tile's bits in LDS --, layer 2 on tile t-1; (3) the prefetched X piece and mask bits of
    // tile t+1 go to LDS, the loads of tile t+2 are issued.  One barrier per tile.
    for (int t = t_beg; t <= t_end + 1; ++t) {
        const int b = (t - t_beg) & 1;
        if (t > t_beg && t - 1 < t_end) {
            const int p = rowbase(t - 1) + pr;
            if (p < N) put(Y1, bits1, p, *reinterpret_cast<const uint4*>(buf1(b ^ 1) + pr * SX + pc * 8));
        }
        if (t > t_beg + 1) {
            const int p = rowbase(t - 2) + pr;
            if (p < N) put(Y2, bits2, p, *reinterpret_cast<const uint4*>(buf2(b) + pr * SX + pc * 8));
        }
        if (role == 0) {
            if (t < t_end) stage(bufX(b), buf1(b), mask1 ? bitsL + b * (kSsRows * LPR) : nullptr);
        } else {
            if (t > t_beg && t - 1 < t_end) stage(buf1(b ^ 1), buf2(b ^ 1), nullptr);
        }
        if (t + 1 < t_end) {
            *reinterpret_cast<uint4*>(bufX(b ^ 1) + pr * SX + pc * 8) = mask0 ? dn_keep_or_scale_bits(rx, m0, slope) : rx;
            bitsL[(b ^ 1) * (kSsRows * LPR) + pr * LPR + pc] = (uint8_t)mk1;
        }
        rx = load_x(t + 2);
        m0 = load_bits(mask0, t + 2);
        mk1 = load_bits(mask1, t + 2);
        __syncthreads();                                         // buf1(b), buf2(b^1), bufX(b^1), bitsL(b^1) complete
    }
}

template <int H>
int launch_chain2(const bf16_t* X, const bf16_t* W1n, const bf16_t* b1, const bf16_t* W2n, const bf16_t* b2, int32_t flags,
                  const uint8_t* mask0, const uint8_t* mask1, int64_t N, bf16_t* Y1, bf16_t* Y2, uint8_t* bits1,
                  uint8_t* bits2, float slope, hipStream_t st) {
    const int64_t num_tiles = dn_cdiv(N, kSsRows);
    const int64_t tiles_per_wg = dn_cdiv(num_tiles, 256 * (1024 / (H * 4)));
    const int64_t grid = dn_cdiv(num_tiles, tiles_per_wg);
    hipLaunchKernelGGL((rows_chain2_kernel<H>), dim3((unsigned)grid), dim3(H * 4), 0, st, X, W1n, b1, W2n, b2, flags, mask0,
                       mask1, (int32_t)N, (int32_t)num_tiles, (int32_t)(stride_tiles() ? 0 : tiles_per_wg), Y1, Y2, bits1, bits2, slope);
    DN_CHECK_LAUNCH();
    return DN_OK;
}

static int tf_depth() {
    static const int d = [] { const int v = dn_knob("DN_TF_DEPTH", 1); return (v >= 1 && v <= 4) ? v : 1; }();
    return d;
}
static int tf_wg_per_cu() {
    static const int d = dn_knob("DN_TF_WGS", 0);
    return d;
}

template <int HI, int HO>
int launch_transform(const bf16_t* X, const bf16_t* X2, int32_t n1, const int32_t* idx, const bf16_t* Wn, const bf16_t* bias,
                     int32_t relu, float slope, const bf16_t* mask_pos, const Chunk* tiles, int64_t num_tiles, bf16_t* Y, hipStream_t st,
                     int32_t w_kn = 0) {
    // contiguous tile ranges keep a workgroup inside one relation most of the time
    const int depth = tf_depth();
    // streaming (non-temporal) stores of the output rows: they are re-read only after ~1 GB of other traffic, so keeping
    // them out of L2 / Infinity Cache is worth 1.5-2 % of the step (DN_NT=0 turns it off: bit 0 transform, bit 1 selfsum)
    static const int nt = dn_knob("DN_NT", 3);
    relu = (relu ? 1 : 0) | ((nt & 1) ? 2 : 0) | ((nt & 4) ? 4 : 0);
    static const int abl = dn_knob("DN_TF_ABL", 0);
    relu |= (abl & 3) << 3;
    relu |= w_kn ? 32 : 0;
    // workgroups per CU: a tile is 32 rows x 2*HI bytes, so narrower rows need more workgroups in flight to keep the same
    // bytes per CU outstanding (H = 128: 64 VGPRs, 26 KB LDS -> 4 fit; measured 2.24 -> 2.07 ms per step at H = 128)
    const int64_t per_cu = tf_wg_per_cu() > 0 ? tf_wg_per_cu() : (depth == 1 ? (HI <= 128 ? 4 : 2) : 1);
    const int64_t max_wg = 256 * per_cu;
    const int64_t tiles_per_wg = dn_cdiv(num_tiles, max_wg);
    const int64_t grid = dn_cdiv(num_tiles, tiles_per_wg);
#define DN_TF_LAUNCH(DEPTH)                                                                                                  \
    hipLaunchKernelGGL((rows_transform_kernel<HI, HO, DEPTH>), dim3((unsigned)grid), dim3(kTfThreads), 0, st, X, X2, n1, idx, \
                       Wn, bias, relu, slope, mask_pos, tiles, (int32_t)num_tiles, (int32_t)tiles_per_wg, Y)
    if (depth == 1) DN_TF_LAUNCH(1);
    else if (depth == 2) DN_TF_LAUNCH(2);
    else if (depth == 3) DN_TF_LAUNCH(3);
    else DN_TF_LAUNCH(4);
#undef DN_TF_LAUNCH
    DN_CHECK_LAUNCH();
    return DN_OK;
}

// out[r] = sum of the partials of relation r's chunks.  The sum is latency-bound (a few hundred 256 KB slabs, each element
// read once, most of them still in the memory-side cache the weight-gradient launch wrote them through), so what counts is bytes
// in flight: a thread owns FOUR consecutive elements (one 16-byte load per slab), 8 threads share them -- thread `slice` adds
// chunks slice, slice+8, ... with 8 loads in flight -- and the 8 slice sums are folded through LDS in slice order.
// Fixed association -> bitwise reproducible.  (Round 3's form -- one float per thread, 4 in flight -- ran at 3.9 TB/s.)
template <typename TO>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial,
                                                           const int32_t* __restrict__ chunk_ptr, int64_t tile_elems,
                                                           TO* __restrict__ out, const float* __restrict__ cs_partial,
                                                           int32_t H, float* __restrict__ out_colsum, TO* __restrict__ out_colsum_lp) {
    constexpr int SL = 8, TL = 256 / SL, EL = 4 * TL;                       // 32 threads x 4 elements per slice
    __shared__ float4 red[SL][TL];
    const int r = blockIdx.y;
    const int quad = threadIdx.x % TL, slice = threadIdx.x / TL;
    const int cb = chunk_ptr[r], ce = chunk_ptr[r + 1];
    // blocks past the tile handle the column sums (bias gradient) the same way
    const int64_t tile_blocks = (tile_elems + EL - 1) / EL;
    const bool is_cs = (int64_t)blockIdx.x >= tile_blocks;
    const float* src = is_cs ? cs_partial : partial;
    const int64_t stride = is_cs ? (int64_t)H : tile_elems;                 // (both multiples of 4: H in {64, 128, 256})
    const int64_t i = (is_cs ? (int64_t)blockIdx.x - tile_blocks : (int64_t)blockIdx.x) * EL + 4 * quad;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    auto add = [&](const float4& v) { s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; };
    if (i < stride) {
        const float* base = src + i;
        int c = cb + slice;
        for (; c + 7 * SL < ce; c += 8 * SL) {
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float4*>(base + (size_t)(c + k * SL) * stride);
#pragma unroll
            for (int k = 0; k < 8; ++k) add(v[k]);
        }
        for (; c < ce; c += SL) add(*reinterpret_cast<const float4*>(base + (size_t)c * stride));
    }
    red[slice][quad] = s;
    __syncthreads();
    if (slice == 0 && i < stride) {
        float4 t = red[0][quad];
#pragma unroll
        for (int k = 1; k < SL; ++k) { const float4 u = red[k][quad]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        const float tv[4] = {t.x, t.y, t.z, t.w};
        if (is_cs) {
            *reinterpret_cast<float4*>(out_colsum + (size_t)r * H + i) = t;
            if (out_colsum_lp) {                                               // (the bias gradient in the parameter's dtype: no cast launch)
#pragma unroll
                for (int k = 0; k < 4; ++k) out_colsum_lp[(size_t)r * H + i + k] = (TO)tv[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) out[(size_t)r * tile_elems + i + k] = (TO)tv[k];
        }
    }
}

// DN_WGRAD_DMA: 0 = never, 1 (default) = at H = 256 (where the register-staged kernel is pinned at one workgroup per CU by
// its 128 accumulator VGPRs), 2 = at H = 128 too (there the register-staged kernel fits two workgroups per CU and is
// ~3 % faster than the ring: 1.95 vs 2.01 ms per step)
int wgrad_dma_mode() {
    static const int mode = dn_knob("DN_WGRAD_DMA", 1);
    return mode;
}

int wgrad_ix_mode() {                              // tuning build: DN_WGRAD_IX=0 keeps the scalar-index ring for gathered operands
    static const int mode = dn_knob("DN_WGRAD_IX", 1);
    return mode;
}

int wgrad_il_mode() {                              // tuning build: DN_WGRAD_IL=0 keeps a relation's chunks contiguous
    static const int mode = dn_knob("DN_WGRAD_IL", 1);
    return mode;
}

// DN_WGRAD_LS (tuning build): bit 0 gathered operands, bit 1 row order, bit 2 row order + mask bits on rows_wgrad_ls_kernel
int wgrad_ls_mode() {
    static const int mode = dn_knob("DN_WGRAD_LS", 3);
    return mode;
}

template <int HI, int HO>
int launch_wgrad(const bf16_t* A, const bf16_t* A2, int32_t na1, const int32_t* ia, const bf16_t* G, const bf16_t* G2,
                 int32_t ng1, const int32_t* ig, const Chunk* chunks, int64_t num_chunks, float* partial, int32_t colsum_of,
                 float* cs_partial, const bf16_t* maskA, bf16_t* A_out, const uint8_t* maskBits, float slope, hipStream_t st,
                 const int32_t* chunk_ptr = nullptr) {
    if constexpr (HI == HO && (HI == 256 || HI == 128)) {
        if (maskA == nullptr && A_out == nullptr && wgrad_dma_mode() >= (HI == 256 ? 1 : 2)) {
            if constexpr (HI == 256) {
                const bool dense_ls = ia == nullptr && ig == nullptr && A2 == nullptr && G2 == nullptr;
                const int ls = wgrad_ls_mode();
                const int ls_mode = (maskBits == nullptr && ia != nullptr && ig != nullptr) ? ((ls & 1) ? 0 : -1)
                                    : (dense_ls ? (maskBits ? ((ls & 4) ? 2 : -1) : ((ls & 2) ? 1 : -1)) : -1);
                if (ls_mode >= 0) {                                        // loaders and MFMA waves apart
#define DN_WGRAD_LS(M)                                                                                                  \
                    hipLaunchKernelGGL((rows_wgrad_ls_kernel<M>), dim3((unsigned)num_chunks), dim3(kLsThreads), 0, st, A, A2, na1, ia, \
                                       G, G2, ng1, ig, chunks, partial, colsum_of, cs_partial, maskBits, slope,                     \
                                       (M == 0 && wgrad_il_mode()) ? chunk_ptr : nullptr)
                    if (ls_mode == 0) DN_WGRAD_LS(0);
                    else if (ls_mode == 1) DN_WGRAD_LS(1);
                    else DN_WGRAD_LS(2);
#undef DN_WGRAD_LS
                    DN_CHECK_LAUNCH();
                    return DN_OK;
                }
                if (maskBits == nullptr && ia != nullptr && ig != nullptr && wgrad_ix_mode()) {   // gathered operands: index ring
                    hipLaunchKernelGGL(rows_wgrad_ix_kernel, dim3((unsigned)num_chunks), dim3(kWgThreads), 0, st, A, A2,
                                       na1, ia, G, G2, ng1, ig, chunks, partial, colsum_of, cs_partial);
                    DN_CHECK_LAUNCH();
                    return DN_OK;
                }
            }
            const bool dense = ia == nullptr && ig == nullptr && A2 == nullptr && G2 == nullptr;
#define DN_WGRAD_DMA(M, D)                                                                                            \
            hipLaunchKernelGGL((rows_wgrad_dma_kernel<HI, M, D>), dim3((unsigned)num_chunks), dim3(kWgThreads), 0, st, A, A2, \
                               na1, ia, G, G2, ng1, ig, chunks, partial, colsum_of, cs_partial, maskBits, slope)
            if (maskBits) {                                                // (mask bits come with ia == NULL: dn_rows_wgrad_bf16)
                if (dense) DN_WGRAD_DMA(true, true);
                else DN_WGRAD_DMA(true, false);
            } else {
                if (dense) DN_WGRAD_DMA(false, true);
                else DN_WGRAD_DMA(false, false);
            }
#undef DN_WGRAD_DMA
            DN_CHECK_LAUNCH();
            return DN_OK;
        }
    }
    hipLaunchKernelGGL((rows_wgrad_kernel<HI, HO>), dim3((unsigned)num_chunks), dim3(kWgThreads), 0, st, A, A2, na1, ia, G, G2,
                       ng1, ig, chunks, partial, colsum_of, cs_partial, maskA, A_out, maskBits, slope);
    DN_CHECK_LAUNCH();
    return DN_OK;
}

// out = (y > 0) ? g : 0, 8 bf16 per lane
__global__ __launch_bounds__(256) void relu_bwd_kernel(const uint4* __restrict__ g, const uint4* __restrict__ y,
                                                       uint4* __restrict__ out, int64_t n16, float slope) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256)
        out[i] = dn_keep_or_scale_mask(g[i], y[i], slope);
}


// ---- tail of a folded pre-aggregation (dn_rows_selfsum_bf16 with fold_info): per tile of 32 segments
//        aux[j]       = bf16( sum of segment j's partial rows, tile order )          (kept: the weight gradient's operand)
//        out[idx[j]] += aux[j] @ Wn^T                                               (fp32 product added to the bf16 row, one rounding)
//      One workgroup per tile, H*4 threads: thread (pr, pc) sums its 8 columns of segment pr, the waves multiply like
//      rows_selfsum_kernel (wave w: output columns 16w..16w+15), the products go through LDS in fp32 back to (pr, pc).
template <int H>
__global__ __launch_bounds__(H * 4) void fold_tail_kernel(const float* __restrict__ part, const int32_t* __restrict__ pptr,
                                                          const bf16_t* __restrict__ Wn, const int32_t* __restrict__ idx,
                                                          int32_t n, int32_t num_tiles, bf16_t* __restrict__ aux,
                                                          bf16_t* __restrict__ out, int32_t w_kn) {
    constexpr int SX = H + kPad, SYF = H + 4;
    constexpr int KS = H / 32, MT = kSsRows / 16, LPR = H / 8;
    static_assert(kSsRows * LPR == H * 4, "one piece per thread");
    __shared__ __attribute__((aligned(16))) bf16_t bufX[kSsRows * SX];
    __shared__ __attribute__((aligned(16))) float bufY[kSsRows * SYF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = wave * 16, pr = tid / LPR, pc = tid % LPR;
    bf16x8 wf[KS];                                               // the relation's weights: loaded once per workgroup
    if (w_kn) {                                                  // stored [k][n] (the parameter's own layout): transposed through LDS
        dn_load_w_kn16<KS>(Wn, H, n0, lane, reinterpret_cast<char*>(bufY) + wave * 1024, wf);
        __syncthreads();                                         // (bufY is the tiles' staging buffer from here on)
    } else {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            wf[ks] = *reinterpret_cast<const bf16x8*>(Wn + (size_t)(n0 + (lane & 15)) * H + ks * 32 + 8 * (lane >> 4));
    }
#pragma unroll 1
    for (int tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
        const int j = tile * kSsRows + pr;
        // my output row's current value: in flight under the partial sums and the MFMAs
        int32_t orow = -1;
        uint4 cur = make_uint4(0, 0, 0, 0);
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (j < n) {
            orow = idx[j];
            cur = *reinterpret_cast<const uint4*>(out + (size_t)orow * H + pc * 8);
            for (int k = pptr[j]; k < pptr[j + 1]; ++k) {
                const float4 u = *reinterpret_cast<const float4*>(part + (size_t)k * H + pc * 8);
                const float4 v = *reinterpret_cast<const float4*>(part + (size_t)k * H + pc * 8 + 4);
                a[0] += u.x; a[1] += u.y; a[2] += u.z; a[3] += u.w; a[4] += v.x; a[5] += v.y; a[6] += v.z; a[7] += v.w;
            }
        }
        bf16x8 xr;
#pragma unroll
        for (int i = 0; i < 8; ++i) xr[i] = (bf16_t)a[i];
        *reinterpret_cast<bf16x8*>(bufX + pr * SX + pc * 8) = xr;
        if (j < n) *reinterpret_cast<bf16x8*>(aux + (size_t)j * H + pc * 8) = xr;
        __syncthreads();
        f32x4 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const bf16x8 xf = *reinterpret_cast<const bf16x8*>(bufX + (m * 16 + (lane & 15)) * SX + ks * 32 + 8 * (lane >> 4));
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], xf, acc[m], 0, 0, 0);
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)                               // lane holds row m*16 + (lane&15), columns n0 + 4*(lane>>4) + i
            *reinterpret_cast<f32x4*>(bufY + (m * 16 + (lane & 15)) * SYF + n0 + 4 * (lane >> 4)) = acc[m];
        __syncthreads();
        if (j < n) {
            const float4 y0 = *reinterpret_cast<const float4*>(bufY + pr * SYF + pc * 8);
            const float4 y1 = *reinterpret_cast<const float4*>(bufY + pr * SYF + pc * 8 + 4);
            const float y[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
            const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
            bf16x8 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o[2 * i] = (bf16_t)(__uint_as_float(w[i] << 16) + y[2 * i]);
                o[2 * i + 1] = (bf16_t)(__uint_as_float(w[i] & 0xffff0000u) + y[2 * i + 1]);
            }
            *reinterpret_cast<bf16x8*>(out + (size_t)orow * H + pc * 8) = o;
        }
        __syncthreads();                                         // bufY has been read: the next tile may overwrite bufX / bufY
    }
}

}  // namespace

extern "C" {

int dn_fold_tail_bf16(const float* part, const int32_t* part_ptr, int64_t num_segments, int32_t H, const void* Wn,
                      const int32_t* idx, void* aux, void* out, int32_t w_kn, dn_stream_t stream) {
    DN_REQUIRE(num_segments >= 0 && num_segments < INT32_MAX, "dn_fold_tail: bad sizes");
    DN_REQUIRE(H == 64 || H == 128 || H == 256, "dn_fold_tail: H must be 64, 128 or 256");
    if (num_segments == 0) return DN_OK;
    DN_REQUIRE(part && part_ptr && Wn && idx && aux && out, "dn_fold_tail: NULL pointer");
    DN_REQUIRE((reinterpret_cast<uintptr_t>(part) | reinterpret_cast<uintptr_t>(Wn) | reinterpret_cast<uintptr_t>(aux) |
                reinterpret_cast<uintptr_t>(out)) % 16 == 0, "dn_fold_tail: unaligned pointer");
    const int32_t num_tiles = (int32_t)dn_cdiv(num_segments, kSsRows);
    const unsigned grid = (unsigned)(num_tiles < 512 ? num_tiles : 512);      // two workgroups per CU loop over the tiles
    hipStream_t st = (hipStream_t)stream;
    const float* p = part;
    const bf16_t *w = (const bf16_t*)Wn;
    bf16_t *a = (bf16_t*)aux, *o = (bf16_t*)out;
    const int32_t n = (int32_t)num_segments;
    if (H == 256) hipLaunchKernelGGL((fold_tail_kernel<256>), dim3(grid), dim3(1024), 0, st, p, part_ptr, w, idx, n, num_tiles, a, o, w_kn);
    else if (H == 128) hipLaunchKernelGGL((fold_tail_kernel<128>), dim3(grid), dim3(512), 0, st, p, part_ptr, w, idx, n, num_tiles, a, o, w_kn);
    else hipLaunchKernelGGL((fold_tail_kernel<64>), dim3(grid), dim3(256), 0, st, p, part_ptr, w, idx, n, num_tiles, a, o, w_kn);
    DN_CHECK_LAUNCH();
    return DN_OK;
}

int dn_relu_bwd_bf16(const void* g, const void* y, void* out, int64_t numel, float act_slope, dn_stream_t stream) {
    DN_REQUIRE(numel >= 0 && numel % 8 == 0, "dn_relu_bwd: numel must be a non-negative multiple of 8");
    if (numel == 0) return DN_OK;
    DN_REQUIRE(g && y && out, "dn_relu_bwd: NULL pointer");
    DN_REQUIRE((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(out)) % 16 == 0,
               "dn_relu_bwd: unaligned pointer");
    const int64_t n16 = numel / 8;
    const int64_t grid = dn_cdiv(n16, 256) < 256 * 16 ? dn_cdiv(n16, 256) : 256 * 16;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const uint4*)g, (const uint4*)y,
                       (uint4*)out, n16, act_slope);
    DN_CHECK_LAUNCH();
    return DN_OK;
}

size_t dn_rows_wgrad_workspace_bytes(int64_t num_chunks, int32_t Hi, int32_t Ho) {
    if (num_chunks < 0 || Hi <= 0 || Ho <= 0) { dn_set_error("dn_rows_wgrad_workspace_bytes: bad sizes"); return 0; }
    return (size_t)(num_chunks > 0 ? num_chunks : 1) * ((size_t)Hi * Ho + Hi) * sizeof(float);
}

int dn_rows_wgrad_bf16(const void* A, const void* A2, int32_t na1, const int32_t* idx_a, const void* G, const void* G2,
                       int32_t ng1, const int32_t* idx_g, int32_t Hi, int32_t Ho, int64_t R, const int32_t* chunks,
                       int64_t num_chunks, const int32_t* chunk_ptr, void* out, int32_t out_is_f32, int32_t colsum_of,
                       float* out_colsum, const void* mask_a, void* a_out, const void* mask_a_bits, void* out_colsum_lp,
                       float act_slope, void* workspace, size_t workspace_bytes, dn_stream_t stream) {
    DN_REQUIRE(mask_a == nullptr || A2 == nullptr, "dn_rows_wgrad: mask_a needs a single A source");
    DN_REQUIRE(out_colsum_lp == nullptr || colsum_of != 0, "dn_rows_wgrad: out_colsum_lp needs colsum_of");
    DN_REQUIRE(mask_a_bits == nullptr || (mask_a == nullptr && A2 == nullptr && idx_a == nullptr),
               "dn_rows_wgrad: mask_a_bits excludes mask_a / A2 / idx_a (the bit-masked operand is read in row order)");
    DN_REQUIRE(a_out == nullptr || (mask_a != nullptr && idx_a == nullptr), "dn_rows_wgrad: a_out needs mask_a and idx_a == NULL");
    DN_REQUIRE((reinterpret_cast<uintptr_t>(mask_a) | reinterpret_cast<uintptr_t>(a_out)) % 16 == 0, "dn_rows_wgrad: unaligned mask");
    DN_REQUIRE(colsum_of >= 0 && (colsum_of & 0xff) <= 2 && (colsum_of >> 8) <= R && ((colsum_of & 0xff) != 0 || (colsum_of >> 8) == 0) &&
               (colsum_of == 0 || out_colsum != nullptr), "dn_rows_wgrad: bad colsum arguments");
    DN_REQUIRE(A2 != nullptr || na1 == 0x7fffffff, "dn_rows_wgrad: A2 == NULL requires na1 == INT32_MAX");
    DN_REQUIRE(G2 != nullptr || ng1 == 0x7fffffff, "dn_rows_wgrad: G2 == NULL requires ng1 == INT32_MAX");
    DN_REQUIRE(R >= 0 && num_chunks >= 0, "dn_rows_wgrad: negative size");
    DN_REQUIRE(Hi == Ho && (Hi == 64 || Hi == 128 || Hi == 256), "dn_rows_wgrad: unsupported widths %d x %d "
               "(square 64/128/256 only)", Hi, Ho);
    if (R == 0) return DN_OK;
    DN_REQUIRE(out && chunk_ptr, "dn_rows_wgrad: NULL pointer");
    DN_REQUIRE(num_chunks == 0 || (A && G && chunks && workspace), "dn_rows_wgrad: NULL pointer");
    DN_REQUIRE(workspace_bytes >= (size_t)num_chunks * ((size_t)Hi * Ho + Hi) * sizeof(float), "dn_rows_wgrad: workspace too small");
    DN_REQUIRE((reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(out_colsum)) % 16 == 0,
               "dn_rows_wgrad: workspace / out_colsum must be 16-byte aligned");
    DN_REQUIRE((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(G)) % 16 == 0, "dn_rows_wgrad: unaligned input");
    hipStream_t st = (hipStream_t)stream;
    const Chunk* ch = reinterpret_cast<const Chunk*>(chunks);
    int rc = DN_OK;
    if (num_chunks > 0) {
        const bf16_t *a = (const bf16_t*)A, *a2 = (const bf16_t*)A2, *g = (const bf16_t*)G, *g2 = (const bf16_t*)G2;
        float* ws = (float*)workspace;
        float* csp = ws + (size_t)num_chunks * Hi * Ho;
        const bf16_t* mk = (const bf16_t*)mask_a;
        bf16_t* ao = (bf16_t*)a_out;
        const uint8_t* mb = (const uint8_t*)mask_a_bits;
        if (Hi == 256) rc = launch_wgrad<256, 256>(a, a2, na1, idx_a, g, g2, ng1, idx_g, ch, num_chunks, ws, colsum_of, csp, mk, ao, mb, act_slope, st, chunk_ptr);
        else if (Hi == 128) rc = launch_wgrad<128, 128>(a, a2, na1, idx_a, g, g2, ng1, idx_g, ch, num_chunks, ws, colsum_of, csp, mk, ao, mb, act_slope, st);
        else rc = launch_wgrad<64, 64>(a, a2, na1, idx_a, g, g2, ng1, idx_g, ch, num_chunks, ws, colsum_of, csp, mk, ao, mb, act_slope, st);
        if (rc != DN_OK) return rc;
    }
    const int64_t tile = (int64_t)Hi * Ho;
    const float* csp = colsum_of ? (const float*)workspace + (size_t)num_chunks * tile : nullptr;
    dim3 grid((unsigned)(dn_cdiv(tile, 128) + (csp ? dn_cdiv(Hi, 128) : 0)), (unsigned)R);
    if (out_is_f32)
        hipLaunchKernelGGL((wgrad_reduce_kernel<float>), grid, dim3(256), 0, st, (const float*)workspace, chunk_ptr, tile,
                           (float*)out, csp, Hi, out_colsum, (float*)out_colsum_lp);
    else
        hipLaunchKernelGGL((wgrad_reduce_kernel<bf16_t>), grid, dim3(256), 0, st, (const float*)workspace, chunk_ptr, tile,
                           (bf16_t*)out, csp, Hi, out_colsum, (bf16_t*)out_colsum_lp);
    DN_CHECK_LAUNCH();
    return DN_OK;
}

int dn_rows_wgrad_multi_bf16(const dn_wgrad_job* jobs, int32_t num_jobs, int32_t H, int64_t R, const int32_t* chunks, int64_t num_chunks,
                             const int32_t* chunk_ptr, void* out, int32_t out_is_f32, float* out_colsum, void* out_colsum_lp,
                             void* workspace, size_t workspace_bytes, dn_stream_t stream) {
    DN_REQUIRE(jobs && num_jobs >= 1 && num_jobs <= 3, "dn_rows_wgrad_multi: 1 .. 3 jobs");
    DN_REQUIRE(H == 64 || H == 128 || H == 256, "dn_rows_wgrad_multi: unsupported width %d (64 / 128 / 256)", H);
    DN_REQUIRE(R >= 1 && num_chunks >= 0, "dn_rows_wgrad_multi: bad sizes");
    DN_REQUIRE(out && chunk_ptr && out_colsum, "dn_rows_wgrad_multi: NULL pointer");
    DN_REQUIRE(num_chunks == 0 || (chunks && workspace), "dn_rows_wgrad_multi: NULL pointer");
    DN_REQUIRE(workspace_bytes >= (size_t)num_chunks * ((size_t)H * H + H) * sizeof(float), "dn_rows_wgrad_multi: workspace too small");
    DN_REQUIRE((reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(out_colsum)) % 16 == 0,
               "dn_rows_wgrad_multi: workspace / out_colsum must be 16-byte aligned");
    WgJobs wj;
    wj.n = num_jobs;
    for (int k = 0; k < 3; ++k) {
        const dn_wgrad_job& q = jobs[k < num_jobs ? k : 0];
        DN_REQUIRE(q.A && q.G, "dn_rows_wgrad_multi: NULL operand");
        DN_REQUIRE((reinterpret_cast<uintptr_t>(q.A) | reinterpret_cast<uintptr_t>(q.G) | reinterpret_cast<uintptr_t>(q.A2) |
                    reinterpret_cast<uintptr_t>(q.G2)) % 16 == 0, "dn_rows_wgrad_multi: unaligned input");
        DN_REQUIRE(q.A2 != nullptr || q.na1 == 0x7fffffff, "dn_rows_wgrad_multi: A2 == NULL requires na1 == INT32_MAX");
        DN_REQUIRE(q.G2 != nullptr || q.ng1 == 0x7fffffff, "dn_rows_wgrad_multi: G2 == NULL requires ng1 == INT32_MAX");
        // (H = 256: colsum_of may name ONE relation of the job whose rows are summed -- (relation + 1) << 8, as dn_rows_wgrad_bf16 takes it)
        DN_REQUIRE(q.colsum_of >= 0 && (q.colsum_of & 0xff) <= 2 && (H == 256 || q.colsum_of <= 2) && q.first_rel >= 0 && q.row0 >= 0,
                   "dn_rows_wgrad_multi: bad job");
        DN_REQUIRE(q.mask_a_bits == nullptr || (q.A2 == nullptr && q.idx_a == nullptr), "dn_rows_wgrad_multi: mask_a_bits excludes A2 / idx_a");
        DN_REQUIRE(k == 0 || k >= num_jobs || q.first_rel > jobs[k - 1].first_rel, "dn_rows_wgrad_multi: jobs must ascend in first_rel");
        // H = 256 (rows_wgrad_ls_multi_kernel): a job is gathered on both sides, or in row order on both (with or without mask bits)
        DN_REQUIRE(H != 256 || ((q.idx_a != nullptr) == (q.idx_g != nullptr) && (q.idx_a != nullptr || (q.A2 == nullptr && q.G2 == nullptr))),
                   "dn_rows_wgrad_multi: at H = 256 a job has both index arrays or neither (and second sources only with them)");
        wj.j[k] = WgJob{(const bf16_t*)q.A, (const bf16_t*)q.A2, q.idx_a, (const bf16_t*)q.G, (const bf16_t*)q.G2, q.idx_g,
                        (const uint8_t*)q.mask_a_bits, q.na1, q.ng1, q.colsum_of, q.first_rel, q.row0, q.act_slope};
    }
    DN_REQUIRE(jobs[0].first_rel == 0, "dn_rows_wgrad_multi: the first job starts at relation 0");
    hipStream_t st = (hipStream_t)stream;
    const Chunk* ch = reinterpret_cast<const Chunk*>(chunks);
    float* ws = (float*)workspace;
    const int64_t tile = (int64_t)H * H;
    float* csp = ws + (size_t)num_chunks * tile;
    if (num_chunks > 0) {
        if (H == 256)
            hipLaunchKernelGGL(rows_wgrad_ls_multi_kernel, dim3((unsigned)num_chunks), dim3(kLsThreads), 0, st, wj, ch, ws, csp,
                               wgrad_il_mode() ? chunk_ptr : (const int32_t*)nullptr);
        else if (H == 128) hipLaunchKernelGGL((rows_wgrad_multi_kernel<128, 128>), dim3((unsigned)num_chunks), dim3(kWgThreads), 0, st, wj, ch, ws, csp);
        else hipLaunchKernelGGL((rows_wgrad_multi_kernel<64, 64>), dim3((unsigned)num_chunks), dim3(kWgThreads), 0, st, wj, ch, ws, csp);
        DN_CHECK_LAUNCH();
    }
    dim3 grid((unsigned)(dn_cdiv(tile, 128) + dn_cdiv(H, 128)), (unsigned)R);
    if (out_is_f32)
        hipLaunchKernelGGL((wgrad_reduce_kernel<float>), grid, dim3(256), 0, st, (const float*)workspace, chunk_ptr, tile, (float*)out, csp, H,
                           out_colsum, (float*)out_colsum_lp);
    else
        hipLaunchKernelGGL((wgrad_reduce_kernel<bf16_t>), grid, dim3(256), 0, st, (const float*)workspace, chunk_ptr, tile, (bf16_t*)out, csp,
                           H, out_colsum, (bf16_t*)out_colsum_lp);
    DN_CHECK_LAUNCH();
    return DN_OK;
}

int dn_rows_transform_bf16(const void* X, const void* X2, int32_t n1, const int32_t* idx, int32_t Hi, int32_t Ho,
                           const void* Wn, const void* bias, int32_t relu, const void* mask_pos, const int32_t* tiles,
                           int64_t num_tiles, void* Y, int32_t w_kn, float act_slope, dn_stream_t stream) {
    DN_REQUIRE(num_tiles >= 0 && num_tiles < 0x7fffffffLL, "dn_rows_transform: bad tile count");
    DN_REQUIRE(Hi == Ho && (Hi == 64 || Hi == 128 || Hi == 256), "dn_rows_transform: unsupported widths %d x %d "
               "(square 64/128/256 only)", Hi, Ho);
    if (num_tiles == 0) return DN_OK;
    DN_REQUIRE(X && Wn && tiles && Y, "dn_rows_transform: NULL pointer");
    DN_REQUIRE(X2 != nullptr || n1 == 0x7fffffff, "dn_rows_transform: X2 == NULL requires n1 == INT32_MAX");
    DN_REQUIRE((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(X2) | reinterpret_cast<uintptr_t>(Wn) |
                reinterpret_cast<uintptr_t>(Y)) % 16 == 0, "dn_rows_transform: unaligned pointer");
    hipStream_t st = (hipStream_t)stream;
    const Chunk* tl = reinterpret_cast<const Chunk*>(tiles);
    const bf16_t *x = (const bf16_t*)X, *x2 = (const bf16_t*)X2, *w = (const bf16_t*)Wn, *b = (const bf16_t*)bias;
    const bf16_t* mk = (const bf16_t*)mask_pos;
    DN_REQUIRE(reinterpret_cast<uintptr_t>(mk) % 16 == 0, "dn_rows_transform: unaligned mask");
    static const int ring = dn_knob("DN_TF_RING", 1);
    static const int nt_knob = dn_knob("DN_NT", 3);
    // (identity rows over the concatenation [X; X2] -- idx == NULL with a second source -- stay on the register-staged kernel:
    //  the ring kernel's loaders only tell the two sources apart through the row index)
    if (Hi == 256 && ring && (idx != nullptr || X2 == nullptr))
        return dn_internal::launch_transform_ring256(X, X2, n1, idx, Wn, bias, relu, nt_knob & 1, mask_pos, tiles, num_tiles, 0, Y, w_kn, act_slope, st);
    if (w_kn && Hi == 256) { dn_set_error("dn_rows_transform: at H = 256 w_kn = 1 is served by the ring kernel only"); return DN_ERR_UNSUPPORTED; }
    if (Hi == 256) return launch_transform<256, 256>(x, x2, n1, idx, w, b, relu, act_slope, mk, tl, num_tiles, (bf16_t*)Y, st);
    if (Hi == 128) return launch_transform<128, 128>(x, x2, n1, idx, w, b, relu, act_slope, mk, tl, num_tiles, (bf16_t*)Y, st, w_kn);
    return launch_transform<64, 64>(x, x2, n1, idx, w, b, relu, act_slope, mk, tl, num_tiles, (bf16_t*)Y, st, w_kn);
}

int dn_rows_selfsum_bf16(const void* X, int32_t H, const void* Wn, const void* bias, const void* S, const void* S2,
                         int32_t n1, const int32_t* slots, int32_t num_slots, int64_t N, void* out, const int32_t* fold_info,
                         float* seg_part, int32_t w_kn, const int32_t* list_ptr, const int32_t* list_rows, int32_t num_edge_rows,
                         int32_t drop_beg, int32_t drop_end, dn_stream_t stream) {
    DN_REQUIRE(fold_info == nullptr || seg_part != nullptr, "dn_rows_selfsum: fold_info needs seg_part");
    DN_REQUIRE((list_ptr == nullptr) == (list_rows == nullptr), "dn_rows_selfsum: list_ptr and list_rows go together");
    DN_REQUIRE(N >= 0 && N < 0x7fffffffLL, "dn_rows_selfsum: bad row count");
    DN_REQUIRE(H == 64 || H == 128 || H == 256, "dn_rows_selfsum: unsupported width %d (64/128/256 only)", H);
    DN_REQUIRE(num_slots == kSsSlots, "dn_rows_selfsum: the slot table must have %d columns", kSsSlots);
    if (N == 0) return DN_OK;
    DN_REQUIRE(X && Wn && slots && out, "dn_rows_selfsum: NULL pointer");
    DN_REQUIRE(S2 != nullptr || n1 == 0x7fffffff, "dn_rows_selfsum: S2 == NULL requires n1 == INT32_MAX");
    DN_REQUIRE((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Wn) | reinterpret_cast<uintptr_t>(S) |
                reinterpret_cast<uintptr_t>(S2) | reinterpret_cast<uintptr_t>(out)) % 16 == 0 && reinterpret_cast<uintptr_t>(slots) % 8 == 0,
               "dn_rows_selfsum: unaligned pointer");
    hipStream_t st = (hipStream_t)stream;
    const bf16_t *x = (const bf16_t*)X, *w = (const bf16_t*)Wn, *b = (const bf16_t*)bias, *s1 = (const bf16_t*)S,
                 *s2 = (const bf16_t*)S2;
    if (H == 256) return launch_selfsum<256>(x, w, b, s1, s2, n1, slots, N, (bf16_t*)out, fold_info, seg_part, st, w_kn, list_ptr, list_rows, num_edge_rows, drop_beg, drop_end);
    if (H == 128) return launch_selfsum<128>(x, w, b, s1, s2, n1, slots, N, (bf16_t*)out, fold_info, seg_part, st, w_kn, list_ptr, list_rows, num_edge_rows, drop_beg, drop_end);
    return launch_selfsum<64>(x, w, b, s1, s2, n1, slots, N, (bf16_t*)out, fold_info, seg_part, st, w_kn, list_ptr, list_rows, num_edge_rows, drop_beg, drop_end);
}

int dn_rows_chain2_bf16(const void* X, int32_t H, const void* W1n, const void* b1, int32_t relu1, const void* mask0_bits,
                        const void* mask1_bits, const void* W2n, const void* b2, int32_t relu2, int64_t N, void* Y1, void* Y2,
                        void* bits1, void* bits2, int32_t w_kn, float act_slope, dn_stream_t stream) {
    DN_REQUIRE(N >= 0 && N < 0x7fffffffLL, "dn_rows_chain2: bad row count");
    DN_REQUIRE(H == 64 || H == 128 || H == 256, "dn_rows_chain2: unsupported width %d (64/128/256 only)", H);
    if (N == 0) return DN_OK;
    DN_REQUIRE(X && W1n && W2n && Y1 && Y2, "dn_rows_chain2: NULL pointer");
    DN_REQUIRE((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W1n) | reinterpret_cast<uintptr_t>(W2n) |
                reinterpret_cast<uintptr_t>(Y1) | reinterpret_cast<uintptr_t>(Y2)) % 16 == 0,
               "dn_rows_chain2: unaligned pointer");
    static const int nt = dn_knob("DN_NT", 3);
    const int32_t flags = (relu1 ? 1 : 0) | (relu2 ? 2 : 0) | ((nt & 1) ? 4 : 0) | ((w_kn & 1) ? 8 : 0) | ((w_kn & 2) ? 16 : 0);
    hipStream_t st = (hipStream_t)stream;
    const bf16_t *x = (const bf16_t*)X, *w1 = (const bf16_t*)W1n, *w2 = (const bf16_t*)W2n, *bb1 = (const bf16_t*)b1,
                 *bb2 = (const bf16_t*)b2;
    const uint8_t *m0 = (const uint8_t*)mask0_bits, *m1 = (const uint8_t*)mask1_bits;
    uint8_t *o1 = (uint8_t*)bits1, *o2 = (uint8_t*)bits2;
    static const int c2ring = dn_knob("DN_C2_RING", 1);                // tuning build: 0 keeps the register-staged kernel at H = 256
    if (H == 256 && c2ring && dn_internal::chain2_ring_supported(m0 != nullptr, m1 != nullptr, o1 != nullptr, o2 != nullptr) &&
        ((reinterpret_cast<uintptr_t>(m0) | reinterpret_cast<uintptr_t>(m1) | reinterpret_cast<uintptr_t>(o1) |
          reinterpret_cast<uintptr_t>(o2) | reinterpret_cast<uintptr_t>(b1) | reinterpret_cast<uintptr_t>(b2)) % 16 == 0))
        return dn_internal::launch_chain2_ring256(X, W1n, b1, W2n, b2, flags, m0, m1, N, Y1, Y2, o1, o2, act_slope, st);
    if (H == 256) return launch_chain2<256>(x, w1, bb1, w2, bb2, flags, m0, m1, N, (bf16_t*)Y1, (bf16_t*)Y2, o1, o2, act_slope, st);
    if (H == 128) return launch_chain2<128>(x, w1, bb1, w2, bb2, flags, m0, m1, N, (bf16_t*)Y1, (bf16_t*)Y2, o1, o2, act_slope, st);
    return launch_chain2<64>(x, w1, bb1, w2, bb2, flags, m0, m1, N, (bf16_t*)Y1, (bf16_t*)Y2, o1, o2, act_slope, st);
}

}  // extern "C"

#ifdef DN_WG_STATS
extern "C" int dn_debug_wgrad_stats(unsigned long long* out) {             // diagnostic build only: 256 x 2 x 5 counters of the last launch
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_stats), sizeof(g_wg_stats)) == hipSuccess ? 0 : -2;
}
#endif
