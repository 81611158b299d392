// One persistent launch per conv direction at H = 256 bf16: the unit stream of dn_close.hip on EIGHT waves (2 per SIMD, 256 VGPRs a
// wave) that issue their own LDS-DMAs, so that a wave has the registers for TWO weight slices -- its relation's (the transform
// units: Y[p] = X[idx p] W[rel p]) and the self loop's (the X units of the closing tiles) -- and a workgroup can alternate between
// the transform of chunk c + 2 and the closing tiles of chunk c without a kernel boundary: the product rows Y of a chunk are
// read back while they are still in the Infinity Cache (rgin.py:102-120,137-160 is ONE update_all).
//
// Skeleton (what differs from dn_close.hip / dn_rel_ring.hip):
//   * no loader waves: wave w of 8 owns output columns [32 w, 32 w + 32) AND rows [4 w, 4 w + 4) of every unit's LDS stage.  Per
//     PAIR of units a wave runs the loaders' half of the old iteration (counted wait, barrier, the row DMAs of units t + 6 and
//     t + 7, every fourth pair the records / source rows / masks of the next batches, the source addresses of units t + 8, t + 9)
//     and then the compute half (units t, t + 1) -- the two waves of a SIMD drift apart by themselves.
//   * vmcnt counts a wave's DMAs AND stores in issue order, so "the rows of unit t + 2 have landed" = at most
//     2 (kNS - 5) + (stores of the last four units) operations outstanding: every unit kind issues a FIXED number of stores (rows
//     past a tile's end go to a dump slot instead of being predicated off) and the wait is picked by that count.
#include "dn_common.h"
#include "dn_internal.h"
#include "../../include/dn_hip.h"

namespace {

typedef __bf16 bf16_t;
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef short short8v __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Unit {
    int32_t flags, beg, end, aux;      // as in dn_close.hip
};
constexpr int kUnitEntry = 1, kUnitLast = 2, kUnitAgg = 4, kUnitNop = 8;
// the fused launch adds TRANSFORM units {kUnitT | flags, first row p, end row, first row of Y they are written to}: the rows
// idx[p .. end) of X times the relation's weights (relation = bits 16-23).  kUnitPub on a workgroup's LAST transform unit of a
// chunk (chunk = bits 24-31): once its rows have left, the workgroup counts itself in done[chunk].  kUnitGate on the first
// closing unit of a chunk in a workgroup's stream (chunk = bits 16-31): its rows may only be requested once done[chunk] == the
// number of workgroups (every product row of the chunk is in memory).
constexpr int kUnitT = 16, kUnitPub = 32, kUnitGate = 64;
constexpr int kPollRing = 32;
constexpr long long kSpinTicks = 200000000ll;                              // give-up budget of a gate (s_memrealtime: 100 MHz -> 2 s)

constexpr int kH = 256;
constexpr int kRowB = 2 * kH;
constexpr int kTR = 32;
constexpr int kStageB = kTR * kRowB;   // 16 KiB
constexpr int kNS = 8;                 // ring stages (128 KiB)
constexpr int kWaves = 8;
constexpr int kThreads = 64 * kWaves;
constexpr int kRowsPerWave = kTR / kWaves;          // 4
constexpr int kDma = kRowsPerWave / 2;              // 2 DMA wave-instructions per wave and unit (2 rows each)
constexpr int kBatch = 8;
constexpr int kRecRing = 32, kIdxRing = 16, kDescRing = 32, kMaskRing = 32, kFoldRing = 32;
constexpr int kFoldInfo = 12;

__device__ int32_t g_fuse_zero[64];
__device__ __attribute__((aligned(16))) uint4 g_fuse_dump[kWaves * 64];     // where the stores of rows past a tile's end go (1 KiB per wave)

__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
    typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v;
    v[0] = (bf16_t)a;
    v[1] = (bf16_t)b;
    return __builtin_bit_cast(uint32_t, v);
}

// "at most 2 (kNS - 5) + s operations outstanding", s = the stores this wave issued during the last four units (0 .. 14)
__device__ __forceinline__ void wait_landed(int s) {
    constexpr int B = kDma * (kNS - 5);
    switch (s) {
    case 0: wait_vmcnt<B + 0>(); break;
    case 1: wait_vmcnt<B + 1>(); break;
    case 2: wait_vmcnt<B + 2>(); break;
    case 3: wait_vmcnt<B + 3>(); break;
    case 4: wait_vmcnt<B + 4>(); break;
    case 5: wait_vmcnt<B + 5>(); break;
    case 6: wait_vmcnt<B + 6>(); break;
    case 7: wait_vmcnt<B + 7>(); break;
    case 8: wait_vmcnt<B + 8>(); break;
    case 9: wait_vmcnt<B + 9>(); break;
    case 10: wait_vmcnt<B + 10>(); break;
    case 11: wait_vmcnt<B + 11>(); break;
    case 12: wait_vmcnt<B + 12>(); break;
    case 13: wait_vmcnt<B + 13>(); break;
    default: wait_vmcnt<B + 14>(); break;
    }
}

#define DN_DS_READ128(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))

// FOLD: 0 none; 2 the absorbed fold (graph tiles, AGG units) -- the partial-row form (1) stays with dn_rows_close_bf16.
// FUSED: the stream also holds transform units (Wrel [R][H][H] in W's layout, tidx = the source row of every product row, S is
// written by them and read by the entry units), gates and publishes (done [chunks] zeroed by the launcher, err [1]).
template <int FOLD, bool FUSED>
__global__ __launch_bounds__(kThreads) void rows_close8_kernel(
    const bf16_t* __restrict__ X, const bf16_t* __restrict__ W, int32_t w_kn, const bf16_t* __restrict__ bias,
    bf16_t* __restrict__ S, const Unit* __restrict__ units, const int32_t* __restrict__ unit_ptr,
    const int32_t* __restrict__ ent_row, const uint32_t* __restrict__ ent_mask, int32_t N, int32_t flags, bf16_t* __restrict__ out,
    const int32_t* __restrict__ fold_info, const bf16_t* __restrict__ W_agg, bf16_t* __restrict__ aux,
    const int32_t* __restrict__ agg_idx, const bf16_t* __restrict__ Wrel, const int32_t* __restrict__ tidx,
    int32_t* __restrict__ done, int32_t* __restrict__ err) {
    __shared__ __attribute__((aligned(1024))) char lds[kNS * kStageB];
    __shared__ __attribute__((aligned(1024))) int32_t foldR[kFoldRing][16];
    __shared__ __attribute__((aligned(256))) uint32_t maskR[kMaskRing][32];
    __shared__ __attribute__((aligned(16))) int32_t descL[kDescRing][4];
    __shared__ __attribute__((aligned(128))) int32_t recR[kWaves][kRecRing][4];              // wave-private rings: unit records
    __shared__ __attribute__((aligned(256))) int32_t idxR[kWaves][kIdxRing][kRowsPerWave];   // ... and source rows
    __shared__ __attribute__((aligned(16))) char wscr[kWaves][2048];
    __shared__ __attribute__((aligned(16))) u32x4 biasL[kH / 8];
    __shared__ __attribute__((aligned(128))) int32_t pollR[kPollRing];                        // done[chunk] of the gated units, fetched a batch ahead
    typedef __attribute__((address_space(3))) char* lds_wp;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_wp)lds;
    const unsigned desc_base = (unsigned)(uintptr_t)(lds_wp)&descL[0][0];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = (int)blockIdx.x, nwg = (int)gridDim.x;
    const int u_beg = unit_ptr[wg];
    const int nt = unit_ptr[wg + 1] - u_beg;
    if (nt <= 0) return;
    units += u_beg;

    // ------------------------------------------------------------------------------------------------ DMA side of a wave
    static_assert(kBatch == 8 && kNS - 1 <= kBatch && 3 * kBatch <= kRecRing && 2 * kBatch <= kIdxRing && 3 * kBatch <= kDescRing &&
                  3 * kBatch <= kMaskRing && 3 * kBatch <= kFoldRing && kNS >= 6 && (kBatch & 1) == 0, "ring sizes");
    const int q = wave;
    const int rin = lane >> 5, pos = lane & 31;
    const unsigned rec_base = (unsigned)(uintptr_t)(lds_wp)&recR[q][0][0];
    const unsigned idx_base = (unsigned)(uintptr_t)(lds_wp)&idxR[q][0][0];
    const unsigned mask_base = (unsigned)(uintptr_t)(lds_wp)&maskR[0][0];
    const unsigned fold_base = (unsigned)(uintptr_t)(lds_wp)&foldR[0][0];
    const int myrow = 2 * (lane & 1) + ((lane >> 1) & 1);                  // the index ring holds a unit's 4 rows as {0, 2, 1, 3}
    const uint64_t baseX = (uint64_t)(uintptr_t)X, dS = (uint64_t)(uintptr_t)S - baseX, dA = (uint64_t)(uintptr_t)aux - baseX;
    auto dma_recs = [&](int T0) __attribute__((always_inline)) {
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(rec_base + (unsigned)(T0 % kRecRing) * 16u));
        if (lane < kBatch) glds16(units + min(T0 + lane, nt - 1), dst);
    };
    auto stage_idx = [&](int T0) __attribute__((always_inline)) {          // source rows of my 4 rows of units T0 .. T0 + 7 (lanes 0 .. 31)
        const int T = T0 + ((lane >> 2) & 7);
        const int32_t* rp = &recR[q][T % kRecRing][0];
        const int fl = rp[0], beg = rp[1], end = rp[2];
        const int pc = end > beg ? min(beg + kRowsPerWave * q + myrow, end - 1) : 0;
        if (q == 0 && lane < 4 * kBatch)
            descL[(T0 + (lane >> 2)) % kDescRing][lane & 3] = recR[0][(T0 + (lane >> 2)) % kRecRing][lane & 3];
        const unsigned dst =
            (unsigned)__builtin_amdgcn_readfirstlane((int)(idx_base + (unsigned)(T0 % kIdxRing) * (4u * kRowsPerWave)));
        if (lane < 4 * kBatch) {
            if (fl & kUnitEntry) glds4(ent_row + pc, dst);                 // lane l lands at + 4 l: [unit][4 rows]
            else if (FUSED && (fl & kUnitT)) glds4(end > beg ? (const void*)(tidx + pc) : (const void*)g_fuse_zero, dst);
            else if (FOLD == 2 && (fl & kUnitAgg)) idxR[q][T % kIdxRing][lane & 3] = wg + nwg * pc;
            else idxR[q][T % kIdxRing][lane & 3] = (fl & kUnitNop) ? 0 : pc;
        }
        {   // membership masks of unit T0 + q, in the k order of the transposed reads (dn_close.hip); lanes 0 .. 31
            const int Tm = T0 + q, kk = lane & 31;
            const int r = (kk >> 3) + 16 * ((kk >> 2) & 1) + 4 * (kk & 3);
            const int32_t* mp = &recR[q][Tm % kRecRing][0];
            const int e = mp[1] + r;
            const bool ok = (mp[0] & kUnitEntry) && e < mp[2];
            const void* msrc = ok ? (const void*)(ent_mask + e) : (const void*)(g_fuse_zero + kk);
            if constexpr (FOLD == 2) {
                const int ord = mp[1] + kk;
                if ((mp[0] & kUnitAgg) && ord < mp[2]) msrc = agg_idx + (wg + nwg * ord);
            }
            const unsigned mdst = (unsigned)__builtin_amdgcn_readfirstlane((int)(mask_base + (unsigned)(Tm % kMaskRing) * 128u));
            if (lane < 32) glds4(msrc, mdst);
        }
        if constexpr (FOLD != 0) {
            if (q == 1) {
                const int Tf = T0 + (lane >> 2), c = lane & 3;
                const int32_t* fp = &recR[1][Tf % kRecRing][0];
                const unsigned fdst = (unsigned)__builtin_amdgcn_readfirstlane((int)(fold_base + (unsigned)(T0 % kFoldRing) * 64u));
                if (lane < 4 * kBatch && c < 3 && !(fp[0] & (kUnitEntry | kUnitAgg | kUnitNop | kUnitT)))
                    glds16(fold_info + (size_t)fp[3] * kFoldInfo + 4 * c, fdst);
            }
        }
        if constexpr (FUSED) {
            if (q == 2) {                                                  // done[chunk] of the batch's gated units (lanes 0 .. 7), as of now
                const int32_t* gp = &recR[2][(T0 + (lane & 7)) % kRecRing][0];
                const bool gate = (gp[0] & kUnitGate) != 0;
                const void* psrc = gate ? (const void*)(done + ((uint32_t)gp[0] >> 16)) : (const void*)g_fuse_zero;
                const unsigned pdst = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(uintptr_t)(lds_wp)&pollR[T0 % kPollRing]));
                if (lane < kBatch) glds4_sc1(psrc, pdst);
            }
        }
    };
    // (a pair's source ROWS are read from the index ring ahead of the barrier that frees their stages; the 64-bit addresses are
    //  formed at issue: two registers per DMA less to carry through the compute half)
    uint32_t srcA[kDma], srcB[kDma];
    int kindA = 0, kindB = 0, pollA = 0, pollB = 0;                        // the pair's unit flags / fetched done[] values (wave-uniform)
    auto prep = [&](int u, uint32_t (&src)[kDma], int& kind, int& poll) __attribute__((always_inline)) {
        typedef int32_t i32x2 __attribute__((ext_vector_type(2)));
        const i32x2 iv = *reinterpret_cast<const i32x2*>(&idxR[q][u % kIdxRing][2 * rin]);     // rows rin, 2 + rin
        const int k_ = recR[q][u % kRecRing][0];
        const int p_ = FUSED ? pollR[u % kPollRing] : 0;
#pragma unroll
        for (int j = 0; j < kDma; ++j) src[j] = (uint32_t)iv[j];
        kind = __builtin_amdgcn_readfirstlane(k_);
        poll = __builtin_amdgcn_readfirstlane(p_);
    };
    auto rows = [&](int u, const uint32_t (&src)[kDma], int kind) __attribute__((always_inline)) {
        const bool ent = (kind & kUnitEntry) != 0, agg = FOLD == 2 && (kind & kUnitAgg) != 0;
        const uint64_t base0 = baseX + (ent ? dS : 0ull) + (agg ? dA : 0ull);      // (X, S or aux)
        const unsigned st = (unsigned)__builtin_amdgcn_readfirstlane(
            (int)(lds_base + (unsigned)(u % kNS) * kStageB + (unsigned)(kRowsPerWave * q) * kRowB));
#pragma unroll
        for (int j = 0; j < kDma; ++j) {
            const int rl = kRowsPerWave * q + 2 * j + rin;                 // row of the stage this lane fills: piece pos ^ (rl & 15)
            glds16(reinterpret_cast<const char*>(base0 + (uint64_t)src[j] * kRowB + (uint64_t)((pos ^ (rl & 15)) * 16)),
                   st + (unsigned)(2 * j) * kRowB);
        }
    };
    auto batch = [&](int u) __attribute__((always_inline)) {
        if ((u & (kBatch - 1)) == 0) {
            stage_idx(u + kBatch);
            dma_recs(u + 2 * kBatch);
        }
    };

    // ------------------------------------------------------------------------------------------------ compute side of a wave
    const bool nt_store = (flags & 2) != 0;
    const int n0 = 32 * wave;
    const int j = lane & 15, g = lane >> 4;
    const unsigned off0 = lds_base + (unsigned)(j * kRowB + ((g ^ j) << 4));
    const int colA0 = 8 * (j >> 2) + (j & 3);
    const uint32_t ocol32 = (uint32_t)(n0 + 8 * g);
    // (my first output column as a value hipcc cannot fold into loop-invariant 64-bit bases: it hoisted `aux + ocol`, `out + ocol`,
    //  `S + ocol` out of the loop, ran out of registers and reloaded one of them from scratch in every X unit -- behind an
    //  s_waitcnt vmcnt(0) that drained the wave's DMA queue once per tile)
    auto ocol_now = [&]() __attribute__((always_inline)) -> size_t {
        uint32_t v = ocol32;
        asm volatile("" : "+v"(v));
        return (size_t)v;
    };
    const int q4 = j >> 2, p4 = j & 3, rt = g + 4 * q4;
    const unsigned tr0 = (unsigned)(rt * kRowB + (((4 * wave + p4) ^ rt) << 4) + 8 * (g & 1));
    const bool odd = (g & 1) != 0;
#define DN_DUMP (reinterpret_cast<char*>(g_fuse_dump) + wave * 1024 + lane * 16)
    bf16x8 wf[8][2];
    if (w_kn == 0) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int n = 0; n < 2; ++n)
                wf[ks][n] = *reinterpret_cast<const bf16x8*>(W + (size_t)(n0 + colA0 + 4 * n) * kH + ks * 32 + 8 * g);
    } else {
        dn_load_w_kn32_lean<8, 4>(W, kH, n0, lane, wscr[wave], wf);
    }
    bf16x8 wfT[FUSED ? 8 : 1][2];                                          // FUSED: my slice of the current relation's weights
    int cur_rel = -1;
    if (tid < kH / 8) biasL[tid] = bias ? *reinterpret_cast<const u32x4*>(bias + 8 * tid) : u32x4{0u, 0u, 0u, 0u};   // (read per epilogue)
    wait_vmcnt<0>();                                                       // no ordinary load may be pending once the DMAs start
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int n = 0; n < 2; ++n) asm volatile("" : "+v"(wf[ks][n]));
    f32x4 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    typedef short4v __attribute__((address_space(3))) * lds_tr;
    auto tr_frags = [&](unsigned sb, bf16x8 (&a)[2]) __attribute__((always_inline)) {
        const unsigned b0 = sb + tr0;
        const short4v r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(uintptr_t)b0);
        const short4v r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(uintptr_t)(b0 ^ 8u));
        const short4v r2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(uintptr_t)(b0 + 16u * kRowB));
        const short4v r3 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(uintptr_t)((b0 ^ 8u) + 16u * kRowB));
        const short4v lo0 = odd ? r1 : r0, hi0 = odd ? r3 : r2;
        const short4v lo1 = odd ? r0 : r1, hi1 = odd ? r2 : r3;
        const short8v f0 = {lo0[0], lo0[1], lo0[2], lo0[3], hi0[0], hi0[1], hi0[2], hi0[3]};
        const short8v f1 = {lo1[0], lo1[1], lo1[2], lo1[3], hi1[0], hi1[1], hi1[2], hi1[3]};
        a[0] = __builtin_bit_cast(bf16x8, f0);
        a[1] = __builtin_bit_cast(bf16x8, f1);
    };

    // rows p0 + j and p0 + 16 + j of the tile, my 8 columns: bias, bf16, TWO 16-byte stores (a row past the tile's end: dump slot)
    auto epilogue = [&](int32_t p0, int32_t pend) __attribute__((always_inline)) {
        const u32x4 bv = biasL[4 * wave + g];                              // bias of my 8 columns (bf16 x 8)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int p = p0 + j + 16 * m;
            float v[8];
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) v[4 * n + i] = acc[m][n][i];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[2 * i] += __uint_as_float(bv[i] << 16);
                v[2 * i + 1] += __uint_as_float(bv[i] & 0xffff0000u);
            }
            u32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
            u32x4* dst = p < pend ? reinterpret_cast<u32x4*>(out + (size_t)p * kH + ocol_now()) : reinterpret_cast<u32x4*>(DN_DUMP);
            if (nt_store) __builtin_nontemporal_store(o, dst);
            else *dst = o;
        }
    };

    // transform unit: product rows pbeg + j, pbeg + 16 + j -> rows ydst + j, ydst + 16 + j of S, my 8 columns, WRITE-THROUGH (sc1):
    // another workgroup, on any XCD, reads them once this one has counted itself in done[chunk]
    auto epilogue_T = [&](int32_t pbeg, int32_t pend, int32_t ydst) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int r = j + 16 * m;
            u32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = pack_bf16x2(acc[m][i >> 1][2 * (i & 1)], acc[m][i >> 1][2 * (i & 1) + 1]);
            const char* dst = pbeg + r < pend ? reinterpret_cast<const char*>(S + (size_t)(ydst + r) * kH + ocol_now()) : DN_DUMP;
#ifdef DN_TUNING_ENV
            if (flags & 4) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(dst), "v"(o) : "memory");           // (experiment: plain)
            else if (flags & 8) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(dst), "v"(o) : "memory");   // (experiment: streaming)
            else
#endif
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(dst), "v"(o) : "memory");
        }
    };

    u32x4 agg_old[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    u32x4* agg_dst[2] = {nullptr, nullptr};
    auto agg_fetch = [&](const uint32_t* tgt, int32_t cnt) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int r = j + 16 * m;
            agg_dst[m] = reinterpret_cast<u32x4*>(out + (size_t)tgt[r < cnt ? r : 0] * kH + ocol_now());
            if (r < cnt) agg_old[m] = *agg_dst[m];
        }
    };
    auto epilogue_agg = [&](int32_t cnt) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int r = j + 16 * m;
            if (r < cnt) {
                const u32x4 old = agg_old[m];
                u32x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int n = i >> 1, e = 2 * (i & 1);
                    o[i] = pack_bf16x2(acc[m][n][e] + __uint_as_float(old[i] << 16), acc[m][n][e + 1] + __uint_as_float(old[i] & 0xffff0000u));
                }
                *agg_dst[m] = o;
            }
        }
    };

// One 32 x 256 by 256 x 32 product of the unit in stage SB with the weight slice WF into acc (zeroed first): the fragments of
// k-steps 0 .. 3 are requested at once, the record of the next unit behind them, k-steps 4 .. 7 into the registers the first four
// just left (an MFMA reads its operands when it issues); counted waits: 7 7 7 7 6 4 2 0 reads may still be in the queue.
#define DN_FETCHS(SLOT, KS, SB)                                                                                       \
    {                                                                                                                 \
        const unsigned a_ = ((SB) + off0) ^ (unsigned)(((KS) & 3) << 6);                                              \
        if ((KS) < 4) {                                                                                               \
            DN_DS_READ128(xf[SLOT][0], a_, 0);                                                                        \
            DN_DS_READ128(xf[SLOT][1], a_, 8192);                                                                     \
        } else {                                                                                                      \
            DN_DS_READ128(xf[SLOT][0], a_, 256);                                                                      \
            DN_DS_READ128(xf[SLOT][1], a_, 8448);                                                                     \
        }                                                                                                             \
    }
#define DN_MF(WF, SLOT, KS, CNT)                                                                                      \
    asm volatile("s_waitcnt lgkmcnt(" #CNT ")" : "+v"(xf[SLOT][0]), "+v"(xf[SLOT][1]));                                \
    _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                                      \
    _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                                      \
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[KS][n], xf[SLOT][m], acc[m][n], 0, 0, 0);            \
    __builtin_amdgcn_sched_barrier(0);
#define DN_BLOCK(WF, SB, AN)                                                                                          \
    {                                                                                                                 \
        bf16x8 xf[4][2];                                                                                              \
        DN_FETCHS(0, 0, SB) DN_FETCHS(1, 1, SB) DN_FETCHS(2, 2, SB) DN_FETCHS(3, 3, SB)                               \
        asm volatile("ds_read_b128 %0, %1" : "=v"(dn) : "v"(AN));                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                                  \
        _Pragma("unroll") for (int n = 0; n < 2; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};                          \
        DN_MF(WF, 0, 0, 7) DN_FETCHS(0, 4, SB) __builtin_amdgcn_sched_barrier(0);                                     \
        DN_MF(WF, 1, 1, 7) DN_FETCHS(1, 5, SB) __builtin_amdgcn_sched_barrier(0);                                     \
        DN_MF(WF, 2, 2, 7) DN_FETCHS(2, 6, SB) __builtin_amdgcn_sched_barrier(0);                                     \
        DN_MF(WF, 3, 3, 7) DN_FETCHS(3, 7, SB) __builtin_amdgcn_sched_barrier(0);                                     \
        DN_MF(WF, 0, 4, 6) DN_MF(WF, 1, 5, 4) DN_MF(WF, 2, 6, 2) DN_MF(WF, 3, 7, 0)                                   \
    }

    // chunks this workgroup has finished transforming, waiting for their stores to have left: set during turn t (pub0), counted in
    // done[] at the top of turn t + 6 -- the counted wait there covers every store of the units up to t + 1 in EVERY wave
    int pub0_c = 0, pub0_n = 0, pub1_c = 0, pub1_n = 0, pub2_c = 0, pub2_n = 0;
    auto fire = [&](int c, int n) __attribute__((always_inline)) {
        if (FUSED && n > 0 && wave == 0 && lane < n)
            __hip_atomic_fetch_add(done + c + lane, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    // Gate in front of the rows of unit v (the first closing unit of a chunk in this stream): every product row of the chunk must
    // be in memory, i.e. done[chunk] == the number of workgroups.  Usual case: the value fetched a batch ago (pollR) already says
    // so.  Otherwise wave 0 polls (bounded: *err is set after 2 s and the launch goes on with whatever is there) and the others wait
    // for it at an extra rendezvous -- `have` is read behind a barrier that the fetching wave passed after its fetch had landed,
    // so every wave takes the same branch.  Then an acquire: this wave's next loads go past the L1.
    auto gate = [&](int fl, int have) __attribute__((always_inline)) {
        if constexpr (FUSED) {
            if (fl & kUnitGate) {
                if (have != nwg) {
                    // Before waiting: count this workgroup in every chunk it has finished transforming (their units lie behind
                    // the compute position; drain the stores, then publish) -- two workgroups that each hold back the other's
                    // missing count would wait for ever.  (The builder keeps a chunk's last transform unit at least 8 positions
                    // in front of the chunk's gate, so that unit HAS been computed when the gate is examined 7 positions early.)
                    wait_vmcnt<0>();
                    __builtin_amdgcn_s_barrier();
                    fire(pub2_c, pub2_n); fire(pub1_c, pub1_n); fire(pub0_c, pub0_n);
                    pub2_n = pub1_n = pub0_n = 0;
                    if (wave == 0) {
                        const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
                        int32_t* cnt = done + ((uint32_t)fl >> 16);
#ifdef DN_TUNING_ENV
                        if (lane == 0) {                                   // (diagnostics: slow gates, gates whose fetched value was 0, first-poll hits)
                            atomicAdd(err + 1, 1);
                            if (have == 0) atomicAdd(err + 2, 1);
                            if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nwg) atomicAdd(err + 3, 1);
                        }
#endif
                        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != nwg) {
                            if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > kSpinTicks) {
                                if (lane == 0) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                break;
                            }
                            __builtin_amdgcn_s_sleep(8);
                        }
                    }
                    __builtin_amdgcn_s_barrier();
                }
                asm volatile("buffer_inv sc1" ::: "memory");
            }
        }
    };

    // ------------------------------------------------------------------------------------------------ prologue
    dma_recs(0);
    dma_recs(kBatch);
    wait_vmcnt<0>();
    stage_idx(0);
    wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                          // descL of batch 0 (wave 0's copy) is visible
#pragma unroll 1
    for (int u = 0; u < kNS - 2; ++u) {
        batch(u);
        prep(u, srcA, kindA, pollA);
        gate(kindA, -1);                                                   // (a stream that starts with closing units: no fetched value yet)
        rows(u, srcA, kindA);
    }
    batch(kNS - 2);
    prep(kNS - 2, srcA, kindA, pollA);
    batch(kNS - 1);
    prep(kNS - 1, srcB, kindB, pollB);

    u32x4 dn;                                                              // record of the next unit
    {
        const unsigned a0 = desc_base;
        asm volatile("ds_read_b128 %0, %1" : "=v"(dn) : "v"(a0));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dn) : : "memory");
    int32_t u_fl = __builtin_amdgcn_readfirstlane((int)dn[0]);
    int32_t u_beg2 = __builtin_amdgcn_readfirstlane((int)dn[1]);
    int32_t u_end = __builtin_amdgcn_readfirstlane((int)dn[2]);
    int32_t u_aux = __builtin_amdgcn_readfirstlane((int)dn[3]);
    int st_prev = 0, st_cur = 0;                                           // stores issued during the previous pair / this pair (wave-uniform)
    auto unit = [&](int u) __attribute__((always_inline)) {
        const unsigned an = desc_base + (unsigned)((u + 1) % kDescRing) * 16u;
        const unsigned sb = (unsigned)(u % kNS) * kStageB;
        int32_t p0 = 0;
        if (u_fl & kUnitNop) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // (the gap in front of the AGG units: everything has left / landed)
            asm volatile("ds_read_b128 %0, %1" : "=v"(dn) : "v"(an));
        } else if (FUSED && (u_fl & kUnitT)) {
            // ---- transform unit: acc = W[rel]^T-slice x rows^T -> S
            if constexpr (FUSED) {
                DN_BLOCK(wfT, sb, an)
                epilogue_T(u_beg2, u_end, u_aux);
                st_cur += 2;
                if (u_fl & kUnitPub) {                                     // (a pair may close two chunks: consecutive ones)
                    if (pub0_n == 0) pub0_c = (int)((uint32_t)u_fl >> 24);
                    pub0_n += 1;
                }
            }
        } else if (!(u_fl & kUnitEntry)) {
            p0 = u_beg2;
            DN_BLOCK(wf, sb, an)
            if constexpr (FOLD == 2) {
                // the tile's segment is complete inside it: its column sum IS the aux row (ONE store per X unit, dump slot if none)
                const int32_t* fr = &foldR[u % kFoldRing][0];
                const int cnt = __builtin_amdgcn_readfirstlane(fr[9]);
                const int first = fr[8];
                const u32x4 w0 = *reinterpret_cast<const u32x4*>(fr), w1 = *reinterpret_cast<const u32x4*>(fr + 4);
                uint32_t id[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    id[i] = (w0[i] >> (8 * g)) & 0xffu;
                    id[4 + i] = (w1[i] >> (8 * g)) & 0xffu;
                }
                bf16x8 a[2];
                tr_frags(lds_base + sb, a);
                const uint32_t me = (uint32_t)j;
                u32x4 iw;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    iw[i] = (id[2 * i] == me ? 0x3f80u : 0u) | (id[2 * i + 1] == me ? 0x3f800000u : 0u);
                const bf16x8 ind = __builtin_bit_cast(bf16x8, iw);
                const f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], ind, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                const f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], ind, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                const u32x4 o = {pack_bf16x2(d0[0], d0[1]), pack_bf16x2(d0[2], d0[3]), pack_bf16x2(d1[0], d1[1]), pack_bf16x2(d1[2], d1[3])};
                u32x4* dst = (int)me < cnt ? reinterpret_cast<u32x4*>(aux + (size_t)(first + (int)me) * kH + ocol_now()) : reinterpret_cast<u32x4*>(DN_DUMP);
                *dst = o;
                st_cur += 1;
            }
        } else {
            p0 = u_aux;
            const u32x4 mA = *reinterpret_cast<const u32x4*>(&maskR[u % kMaskRing][8 * g]);
            const u32x4 mB = *reinterpret_cast<const u32x4*>(&maskR[u % kMaskRing][8 * g + 4]);
            bf16x8 a[2];
            tr_frags(lds_base + sb, a);
            asm volatile("ds_read_b128 %0, %1" : "=v"(dn) : "v"(an));
            bf16x8 sel[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const uint32_t rb = (uint32_t)(j + 16 * m);
                u32x4 sw;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    sw[i] = (((mA[2 * i] >> rb) & 1u) | (((mA[2 * i + 1] >> rb) & 1u) << 16)) * 0x3f80u;
                    sw[2 + i] = (((mB[2 * i] >> rb) & 1u) | (((mB[2 * i + 1] >> rb) & 1u) << 16)) * 0x3f80u;
                }
                sel[m] = __builtin_bit_cast(bf16x8, sw);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[n], sel[m], acc[m][n], 0, 0, 0);
        }
        if (u_fl & kUnitLast) {
            epilogue(p0, p0 + ((u_fl >> 8) & 0xff));
            st_cur += 2;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dn) : : "memory");
        u_fl = __builtin_amdgcn_readfirstlane((int)dn[0]);
        u_beg2 = __builtin_amdgcn_readfirstlane((int)dn[1]);
        u_end = __builtin_amdgcn_readfirstlane((int)dn[2]);
        u_aux = __builtin_amdgcn_readfirstlane((int)dn[3]);
    };

    // ------------------------------------------------------------------------------------------------ main loop, a pair of units per turn
    // DMA half of a turn at even position t: the rows up to unit t + 2 have landed (counted: `stores` = the stores this wave issued
    // during the last four units; DRAIN: everything), one rendezvous, then the rows of units t + 6, t + 7 into the stages that
    // units t - 2, t - 1 used, every fourth turn the next batches' records / source rows / masks, and the addresses of t + 8, t + 9.
    auto dma_half = [&](int tt, int stores, bool drain) __attribute__((always_inline)) {
        if (drain) wait_vmcnt<0>();
        else wait_landed(stores);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if constexpr (FUSED) {
            fire(pub2_c, pub2_n);
            pub2_c = pub1_c; pub2_n = pub1_n; pub1_c = pub0_c; pub1_n = pub0_n; pub0_n = 0;
            gate(kindA, pollA);
            gate(kindB, pollB);
        }
        rows(tt + kNS - 2, srcA, kindA);
        rows(tt + kNS - 1, srcB, kindB);
        batch(tt + kNS);
        prep(tt + kNS, srcA, kindA, pollA);
        prep(tt + kNS + 1, srcB, kindB, pollB);
    };
    wait_vmcnt<kDma*(kNS - 3)>();                                          // unit 0 has landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int u = 0;
#pragma unroll 1
    for (; u < nt; u += 2) {
        if (FOLD == 2 && (u_fl & kUnitAgg)) break;
        dma_half(u, st_prev + st_cur, false);
        st_prev = st_cur;
        st_cur = 0;
        if constexpr (FUSED) {
            // the ONE place where a workgroup changes relation: a pair's first unit (the builder puts a switch at an even position);
            // a second reload site inside unit() made hipcc carry both weight sets through copies (200 spilled registers)
            if (u_fl & kUnitT) {
                const int rel = (u_fl >> 16) & 0xff;
                if (rel != cur_rel) {                                      // wave-uniform, rare (a workgroup serves one relation, a helper a few)
                    cur_rel = rel;
                    const bf16_t* w = Wrel + (size_t)rel * kH * kH;
                    if (w_kn) {
                        dn_load_w_kn32_lean<8, 2>(w, kH, n0, lane, wscr[wave], wfT);
                    } else {
#pragma unroll
                        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                            for (int n = 0; n < 2; ++n)
                                wfT[ks][n] = *reinterpret_cast<const bf16x8*>(w + (size_t)(n0 + colA0 + 4 * n) * kH + ks * 32 + 8 * g);
                    }
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                        for (int n = 0; n < 2; ++n) asm volatile("" : "+v"(wfT[ks][n]));   // the wait for them stays in this branch
                }
            }
        }
        unit(u);
        if (u + 1 >= nt || (FOLD == 2 && (u_fl & kUnitAgg))) { u += 1; break; }
        unit(u + 1);
    }
    if constexpr (FOLD == 2) {
        if (u < nt) {                                                      // the workgroup's AGG units: W_agg replaces W_loop in wf
            if (w_kn == 0) {
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        wf[ks][n] = *reinterpret_cast<const bf16x8*>(W_agg + (size_t)(n0 + colA0 + 4 * n) * kH + ks * 32 + 8 * g);
            } else {
                dn_load_w_kn32<8>(W_agg, kH, n0, lane, wscr[wave], wf);
            }
            // phase B: nothing but AGG units -- the X-unit product on 32 aux rows with W_agg, added to their output rows; a short
            // tail (one unit per 32 tiles of the workgroup), so every turn simply drains the wave's queue
#pragma unroll 1
            for (; u < nt; ++u) {
                if ((u & 1) == 0) dma_half(u, 0, true);
                const unsigned an = desc_base + (unsigned)((u + 1) % kDescRing) * 16u;
                const unsigned sb = (unsigned)(u % kNS) * kStageB;
                const int32_t cnt = (u_fl & kUnitAgg) ? u_end - u_beg2 : 0;
                agg_fetch(&maskR[u % kMaskRing][0], cnt);
                DN_BLOCK(wf, sb, an)
                epilogue_agg(cnt);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dn) : : "memory");
                u_fl = __builtin_amdgcn_readfirstlane((int)dn[0]);
                u_beg2 = __builtin_amdgcn_readfirstlane((int)dn[1]);
                u_end = __builtin_amdgcn_readfirstlane((int)dn[2]);
            }
        }
    }
#undef DN_DUMP
#undef DN_FETCHS
#undef DN_MF
#undef DN_BLOCK
    wait_vmcnt<0>();                                                       // nothing may land after the LDS is given back
    if constexpr (FUSED) {
        if (pub0_n + pub1_n + pub2_n > 0) {                                // (a stream that ends right behind transform units)
            __builtin_amdgcn_s_barrier();
            fire(pub2_c, pub2_n); fire(pub1_c, pub1_n); fire(pub0_c, pub0_n);
        }
    }
}
#undef DN_DS_READ128

}  // namespace

namespace dn_internal {

int launch_close8(const void* X, const void* W, int32_t w_kn, const void* bias, const void* S, const int32_t* unit_ptr,
                  const int32_t* units, int32_t num_wg, const int32_t* ent_row, const uint32_t* ent_mask, int64_t N, void* out,
                  const int32_t* fold_info, const void* W_agg, void* aux, const int32_t* agg_idx, int32_t nt_store,
                  const void* Wrel, const int32_t* tidx, int32_t* done, int32_t num_chunks, int32_t* err, hipStream_t st) {
    const int32_t flags = (nt_store ? 2 : 0) | ((dn_knob("DN_FUSE_YSTORE", 0) & 3) << 2);   // tuning build: 1 plain, 2 streaming stores of the product rows
    bf16_t* s = S ? (bf16_t*)const_cast<void*>(S) : (bf16_t*)const_cast<void*>(X);
    const bool fused = Wrel != nullptr;
    if (fused) DN_CHECK_HIP(hipMemsetAsync(done, 0, sizeof(int32_t) * (size_t)num_chunks, st));   // every launch counts from zero
#define DN_C8_LAUNCH(F, U)                                                                                                         \
    hipLaunchKernelGGL((rows_close8_kernel<F, U>), dim3((unsigned)num_wg), dim3(kThreads), 0, st, (const bf16_t*)X, (const bf16_t*)W, w_kn, \
                       (const bf16_t*)bias, s, reinterpret_cast<const Unit*>(units), unit_ptr, ent_row, ent_mask, (int32_t)N, flags,  \
                       (bf16_t*)out, fold_info, (const bf16_t*)W_agg, (bf16_t*)aux, agg_idx, (const bf16_t*)Wrel, tidx, done, err)
    if (W_agg) { if (fused) DN_C8_LAUNCH(2, true); else DN_C8_LAUNCH(2, false); }
    else { if (fused) DN_C8_LAUNCH(0, true); else DN_C8_LAUNCH(0, false); }
#undef DN_C8_LAUNCH
    DN_CHECK_LAUNCH();
    return DN_OK;
}

}  // namespace dn_internal

extern "C" {

int dn_rows_fused_bf16(const void* X, int32_t H, const void* Wrel, const void* W_loop, int32_t w_kn, const void* bias, void* S,
                       const int32_t* row_src, const int32_t* unit_ptr, const int32_t* units, int32_t num_wg,
                       const int32_t* ent_row, const uint32_t* ent_mask, int64_t N, void* out, const int32_t* fold_info,
                       const void* W_agg, void* aux, const int32_t* agg_idx, int32_t* done, int32_t num_chunks, int32_t* err,
                       dn_stream_t stream) {
    DN_REQUIRE(H == 256, "dn_rows_fused: unsupported width %d (256 only)", H);
    DN_REQUIRE(N >= 0 && N < 0x7fffffffLL && num_wg > 0 && num_wg <= 4096 && num_chunks > 0 && num_chunks <= 256,
               "dn_rows_fused: bad sizes");
    const bool agg = W_agg != nullptr || aux != nullptr || agg_idx != nullptr;
    DN_REQUIRE(agg ? (fold_info && W_agg && aux && agg_idx) : fold_info == nullptr,
               "dn_rows_fused: fold_info, W_agg, aux and agg_idx go together (the absorbed fold) or not at all");
    if (N == 0) return DN_OK;
    DN_REQUIRE(X && Wrel && W_loop && S && row_src && unit_ptr && units && ent_row && ent_mask && out && done && err,
               "dn_rows_fused: NULL pointer");
    DN_REQUIRE((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Wrel) | reinterpret_cast<uintptr_t>(W_loop) |
                reinterpret_cast<uintptr_t>(S) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(units) |
                reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(fold_info) | reinterpret_cast<uintptr_t>(W_agg) |
                reinterpret_cast<uintptr_t>(aux)) % 16 == 0, "dn_rows_fused: unaligned pointer");
    // every workgroup must be resident at once (a closing unit waits for the transform units of ALL workgroups): one per CU
    int dev = 0, cus = 0;
    DN_CHECK_HIP(hipGetDevice(&dev));
    DN_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    DN_REQUIRE(num_wg <= cus, "dn_rows_fused: %d workgroups on a device with %d compute units (they wait for each other)", num_wg, cus);
    return dn_internal::launch_close8(X, W_loop, w_kn, bias, S, unit_ptr, units, num_wg, ent_row, ent_mask, N, out, fold_info, W_agg, aux,
                                      agg_idx, 1, Wrel, row_src, done, num_chunks, err, (hipStream_t)stream);
}

}  // extern "C"
