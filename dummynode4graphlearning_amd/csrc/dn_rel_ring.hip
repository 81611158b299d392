// Relation-wise transform of gathered rows, H = 256 bf16: persistent workgroups, wave-specialised.
//
//   Y[p, :] = epi( X[idx[p], :] @ Wn[rel(p)]^T )     (rows p relation-major; tile table {rel, beg, end, 0}, <= 32 rows a tile)
//
// One workgroup per CU, 12 waves (3 per SIMD: two compute waves and a loader):
//   * waves 8..11 ("loaders") issue nothing but LDS-DMAs (global_load_lds): the 32 gathered rows of a tile go global -> LDS
//     without staging VGPRs into a ring of NS 16-KiB stages.  Tile records and row indices travel AHEAD of the rows through two
//     small per-wave LDS rings, also by LDS-DMA and in batches of 8 tiles (one wave-instruction fetches the 64 indices a loader
//     needs for 8 tiles): a scalar or vector load per tile instead (record -> index -> row address) is a dependent chain
//     through HBM that one tile of look-ahead cannot hide -- the index stream is read once, every tile misses.  The source
//     addresses of a pair of tiles are worked out BEFORE the barrier that frees their stages, so the eight row DMAs leave
//     right behind it.  vmcnt counts every vector-memory operation of a wave in issue order, so "the rows up to tile t + 2 have
//     landed" is a counted s_waitcnt vmcnt(4 (NS-5)) (a batch's two extra DMAs only make it stricter).
//   * waves 0..7 ("compute") own 32 output columns each, their slice of Wn[rel] in 64 VGPRs (reloaded when the relation
//     changes).  Per tile: 16 ds_read_b128 fragment reads (inline asm with hand-counted s_waitcnt lgkmcnt(n) per k-step: hipcc
//     either serialises them -- one read, lgkmcnt(0), two MFMAs -- or, behind a sched_barrier, waits for all sixteen; a pending
//     scalar load would force lgkmcnt(0) too, so the loop has no SMEM: the next tile's record arrives as a 17th LDS read), 32
//     v_mfma_f32_16x16x32_bf16 on the TRANSPOSED tile (A = weights, B = rows), and the finished rows leave straight from the
//     accumulators: the weight rows of the two 16-column MFMA tiles are interleaved (A row i of tile n <-> output column
//     8 (i >> 2) + 4 n + (i & 3)), so a lane ends up with 8 CONSECUTIVE columns of one row = one 16-byte streaming store, no
//     second trip through LDS and no second barrier.
//   * the compute loop is software-pipelined over tiles: the fragments of the first two k-steps of tile t + 1 are fetched during
//     tile t (into the registers those k-steps just consumed), so a tile's MFMAs start right behind the previous tile's and the
//     other six k-steps' reads land behind them.  (All eight a tile ahead, or the two waves of a SIMD staggered by half a tile
//     with the sums kept across the barrier, do not fit the 168 registers a wave has at 3 waves per SIMD: hipcc spills.)
//   * ONE raw s_barrier per PAIR of tiles joins all twelve waves (every rendezvous makes eleven waves wait for the slowest):
//     behind the barrier of tiles 2 p, 2 p + 1 the stages up to tile 2 p + 2 are visible to the compute waves (each loader
//     waited for its own DMAs) and the stages of tiles 2 p - 2, 2 p - 1 are free (every compute wave drained its LDS reads
//     before arriving), so the loaders refill them with tiles 2 p + 6, 2 p + 7.
//   * an LDS-DMA wave-instruction writes 1 KiB lane-linearly (2 rows): the image is unpadded and the bank spread of the fragment
//     reads comes from an XOR swizzle of the 16-byte pieces applied to the per-lane SOURCE address: LDS (row r, position q)
//     holds global piece q ^ (r & 15).  ds_read_b128 serves a wave in 4 groups of 16 lanes ({0-3,12-15,20-27}, ...); with
//     lane = (row & 15) + 16 (k-group) those 16 reads fall on 16 distinct 16-byte bank slots.
//
// Why: the register-staged kernel (dn_rel.hip) needs ~2,300 shader cycles per tile and CU whatever the memory system does
// (hipcc serialises its fragment reads, one tile in flight per workgroup, two barriers and a trip through LDS for the
// output), and under this traffic the chip holds ~1.45 GHz: the launch was bound by its own instruction stream, which is why
// neither fewer HBM bytes (L2-blocked tile order) nor cache residency made it faster.  This
// kernel needs ~1,450 (MFMA floor: 1,024); measurements: DESIGN.md section 4.
#include "dn_common.h"
#include "dn_internal.h"

namespace {

typedef __bf16 bf16_t;
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Tile {
    int32_t rel, beg, end, pad;
};

constexpr int kH = 256;
constexpr int kRowB = 2 * kH;          // bytes per row
constexpr int kTR = 32;                // rows per tile
constexpr int kStageB = kTR * kRowB;   // 16 KiB
constexpr int kNS = 8;                 // ring stages (128 KiB)
constexpr int kCompute = 8, kLoaders = 4;
constexpr int kThreads = 64 * (kCompute + kLoaders);
constexpr int kRowsPerLoader = kTR / kLoaders;      // 8
constexpr int kDmaPerTile = kRowsPerLoader / 2;     // 4 DMA wave-instructions per loader and tile (2 rows each)
constexpr int kBatch = 8;                           // tiles per batch of records / indices
constexpr int kRecRing = 32, kIdxRing = 16, kDescRing = 32;

__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
    typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v;
    v[0] = (bf16_t)a;
    v[1] = (bf16_t)b;
    return __builtin_bit_cast(uint32_t, v);
}

#ifdef DN_RING_STATS
// diagnostic build (-DDN_RING_STATS): cycle counters of the last launch, per workgroup: compute wave 0 {loop, barrier wait, reads + MFMAs, epilogue},
// loader 0 {loop, vm wait, barrier wait, body}, wall ticks (100 MHz) of the compute loop, its start tick
__device__ unsigned long long g_ring_stats[256][10];
#define DN_STAMP() __builtin_amdgcn_s_memtime()
#define DN_STAT(var, expr) var += (expr)
#else
#define DN_STAMP() 0ull
#define DN_STAT(var, expr)
#endif

__device__ int32_t g_zero_idx[64];     // row 0 (device globals are zero-initialised): what an EMPTY tile's rows gather

#define DN_DS_READ128(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))

template <bool MASK, bool IDX, bool EPI, bool HASX2>
__global__ __launch_bounds__(kThreads) void rows_transform_ring_kernel(
    const bf16_t* __restrict__ X, const bf16_t* __restrict__ X2, int32_t n1, const int32_t* __restrict__ idx,
    const bf16_t* __restrict__ Wn, const bf16_t* __restrict__ bias, int32_t flags, const bf16_t* __restrict__ mask_pos,
    const Tile* __restrict__ tiles, int32_t num_tiles, int32_t tiles_per_wg, bf16_t* __restrict__ Y, float slope) {
    __shared__ __attribute__((aligned(1024))) char lds[kNS * kStageB];
    __shared__ __attribute__((aligned(16))) int32_t descL[kDescRing][4]; // tile records for the compute waves (copied by loader 0)
    __shared__ __attribute__((aligned(128))) int32_t recR[kLoaders][kRecRing][4];            // loader-private rings: tile records
    __shared__ __attribute__((aligned(256))) int32_t idxR[kLoaders][kIdxRing][kRowsPerLoader];  // ... and row indices, by LDS-DMA
    __shared__ __attribute__((aligned(16))) char wscr[kCompute][2048];     // per compute wave: transposition scratch of [k][n] weights
    typedef __attribute__((address_space(3))) char* lds_wp;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_wp)lds;
    const unsigned desc_base = (unsigned)(uintptr_t)(lds_wp)&descL[0][0];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t_beg = (int)blockIdx.x * tiles_per_wg;
    int nt = min(t_beg + tiles_per_wg, num_tiles) - t_beg;
    if (nt <= 0) return;
    tiles += t_beg;
    // Trailing EMPTY slots of this workgroup's range are not walked (an empty tile costs a ring stage and a tile's MFMAs; the sweep
    // order cuts its helper workgroups' shares short on purpose -- dn_index.hip): nt = last live slot + 1, found by every wave for
    // itself from the same records (uniform), the loads of up to 512 slots in flight together.
#ifdef DN_TUNING_ENV
    if (nt <= 512 && !(flags & (128 << 3))) {                              // (DN_TF_ABL bit 7: every slot walked, as before round 6)
#else
    if (nt <= 512) {
#endif
        bool live[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int k = 64 * c + lane;
            live[c] = false;
            if (k < nt) { const Tile tl = tiles[k]; live[c] = tl.end > tl.beg; }
        }
        int last = -1;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const unsigned long long m = __ballot(live[c]);
            if (m) last = 64 * c + 63 - __builtin_clzll(m);
        }
        nt = last + 1;
        if (nt <= 0) return;
    }

    if (wave >= kCompute) {
        // ------------------------------------------------------------------------------------------------ loaders
        // Batch b = tiles [8 b, 8 b + 8).  At tile u = 8 b (batch(u)) the indices of batch b + 1 and
        // the records of batch b + 2 are requested; both are consumed 8 tiles later, by which time the counted wait at the top
        // of an iteration (at most 3 tiles' DMAs outstanding) has covered them.
        static_assert(kBatch == 8 && kNS - 1 <= kBatch && 3 * kBatch <= kRecRing && 2 * kBatch <= kIdxRing &&
                      3 * kBatch <= kDescRing && kDmaPerTile * (kNS - 2) < 64 && kNS >= 6 && (kBatch & 1) == 0, "ring sizes");
        const int q = wave - kCompute;
        const int rin = lane >> 5, pos = lane & 31;
#ifdef DN_TUNING_ENV
        if (flags & 256) __builtin_amdgcn_s_setprio(3);                    // (experiment: the loaders win the issue arbitration)
#endif
        int swoff[kDmaPerTile];
#pragma unroll
        for (int j = 0; j < kDmaPerTile; ++j) {
            const int rl = kRowsPerLoader * q + 2 * j + rin;               // row of the stage this lane fills
            swoff[j] = (pos ^ (rl & 15)) * 16;                             // source byte offset inside the row
        }
        const unsigned rec_base = (unsigned)(uintptr_t)(lds_wp)&recR[q][0][0];
        const unsigned idx_base = (unsigned)(uintptr_t)(lds_wp)&idxR[q][0][0];
        // the index ring holds a tile's 8 rows as {0, 2, 4, 6, 1, 3, 5, 7}: the four rows a lane needs (2 j + rin) are one 16-byte read
        const int myrow = 2 * (lane & 3) + ((lane >> 2) & 1);              // row (of my 8) whose index lane 8 k + l fetches
        auto dma_recs = [&](int T0) {                                      // records of tiles T0 .. T0 + 7 (clamped: valid memory)
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(rec_base + (unsigned)(T0 % kRecRing) * 16u));
            if (lane < kBatch) glds16(tiles + min(T0 + lane, nt - 1), dst);
        };
        auto stage_idx = [&](int T0) {                                     // indices of my 8 rows of tiles T0 .. T0 + 7; records -> descL
            // lane 8 k + l: tile T0 + k, row myrow.  Everything in the vector domain (no scalar round trip).  A tile past the end
            // repeats the last record: its rows are fetched and never used.  Rows past a tile's end repeat its last row; an EMPTY
            // tile gathers row 0 (the row in front of its range may belong to a relation this launch leaves out, whose index
            // points into an X2 the caller did not pass).
            const int T = T0 + (lane >> 3);
            const int32_t* rp = &recR[q][T % kRecRing][0];
            const int beg = rp[1], end = rp[2];
            const int pc = end > beg ? min(beg + kRowsPerLoader * q + myrow, end - 1) : 0;
            if (q == 0 && lane < 4 * kBatch)
                descL[(T0 + (lane >> 2)) % kDescRing][lane & 3] = recR[0][(T0 + (lane >> 2)) % kRecRing][lane & 3];
            if constexpr (IDX) {
                const unsigned dst =
                    (unsigned)__builtin_amdgcn_readfirstlane((int)(idx_base + (unsigned)(T0 % kIdxRing) * (4u * kRowsPerLoader)));
                glds4(end > beg ? idx + pc : g_zero_idx, dst);             // lane l lands at + 4 l: [tile][8 rows]
            } else {
                idxR[q][T % kIdxRing][lane & 7] = pc;
            }
        };
        // The rows of tile u are requested right behind the barrier that frees their stage, from addresses worked out BEFORE it
        // (prep(u): the index read and the address arithmetic of the next tile run while the loader would otherwise wait).
        const char* srcA[kDmaPerTile];
        const char* srcB[kDmaPerTile];
        auto prep = [&](int u, const char* (&src)[kDmaPerTile]) {
            typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
            const i32x4 iv = *reinterpret_cast<const i32x4*>(&idxR[q][u % kIdxRing][4 * rin]);   // rows rin, 2 + rin, 4 + rin, 6 + rin
#pragma unroll
            for (int j = 0; j < kDmaPerTile; ++j) {
                int32_t r = iv[j];
#ifdef DN_TUNING_ENV
                if (flags & 16) r &= 1023;                                 // (ablation: every gather hits L2)
#endif
                const char* base = reinterpret_cast<const char*>(X) + (size_t)r * kRowB;
                if constexpr (HASX2)
                    if (r >= n1) base = reinterpret_cast<const char*>(X2) + (size_t)(r - n1) * kRowB;
                src[j] = base + swoff[j];
            }
        };
        auto rows = [&](int u, const char* (&src)[kDmaPerTile]) {
            const unsigned st = lds_base + (unsigned)(u % kNS) * kStageB + (unsigned)(kRowsPerLoader * q) * kRowB;
#ifdef DN_TUNING_ENV
            if (flags & 128) return;                                       // (ablation: no row DMAs)
#endif
#pragma unroll
            for (int j = 0; j < kDmaPerTile; ++j) glds16(src[j], st + (unsigned)(2 * j) * kRowB);   // lane l lands at + 16 l
        };
        auto batch = [&](int u) {                                          // u = 8 b: indices of batch b + 1, records of batch b + 2
            if ((u & (kBatch - 1)) == 0) {                                 // wave-uniform
                stage_idx(u + kBatch);
                dma_recs(u + 2 * kBatch);
            }
        };
        dma_recs(0);
        dma_recs(kBatch);
        wait_vmcnt<0>();
        stage_idx(0);
        wait_vmcnt<0>();
        // One rendezvous per PAIR of tiles (every barrier makes twelve waves wait for the slowest of them): at the barrier
        // of tiles 2 p, 2 p + 1 the tiles up to 2 p + 2 have landed (the compute waves fetch one tile's first fragments during
        // the tile before it) and the stages of tiles 2 p - 2, 2 p - 1 are free; the rows of tiles 2 p + 6, 2 p + 7 go there.
#pragma unroll 1
        for (int u = 0; u < kNS - 2; ++u) {
            batch(u);
            prep(u, srcA);
            rows(u, srcA);
        }
        batch(kNS - 2);
        prep(kNS - 2, srcA);
        batch(kNS - 1);
        prep(kNS - 1, srcB);
        wait_vmcnt<kDmaPerTile*(kNS - 3)>();                               // tile 0 has landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                      // (the compute waves fetch tile 0's fragments)
        unsigned long long st_vm = 0, st_bar = 0, st_body = 0;
        const unsigned long long l0 = DN_STAMP();
#pragma unroll 1
        for (int t = 0; t < nt; t += 2) {
            const unsigned long long a0 = DN_STAMP();
#ifdef DN_TUNING_ENV
            if (flags & 128) wait_vmcnt<0>();                              // (ablation without row DMAs: the count no longer bounds)
#endif
            wait_vmcnt<kDmaPerTile*(kNS - 5)>();                           // issued: up to tile t + 5; landed: up to tile t + 2
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // ... and the records I copied for the compute waves
            const unsigned long long a1 = DN_STAMP();
            __builtin_amdgcn_s_barrier();                                  // everyone's have; stages of tiles t-2, t-1 are free
            const unsigned long long a2 = DN_STAMP();
            rows(t + kNS - 2, srcA);
            rows(t + kNS - 1, srcB);
            batch(t + kNS);
            prep(t + kNS, srcA);
            prep(t + kNS + 1, srcB);
            DN_STAT(st_vm, a1 - a0); DN_STAT(st_bar, a2 - a1); DN_STAT(st_body, DN_STAMP() - a2);
        }
        wait_vmcnt<0>();                                                   // nothing may land after the LDS is given back
#ifdef DN_RING_STATS
        if (q == 0 && lane == 0 && blockIdx.x < 256) {
            g_ring_stats[blockIdx.x][4] = DN_STAMP() - l0; g_ring_stats[blockIdx.x][5] = st_vm;
            g_ring_stats[blockIdx.x][6] = st_bar; g_ring_stats[blockIdx.x][7] = st_body;
        }
#endif
        return;
    }

    // ---------------------------------------------------------------------------------------------------- compute
    const bool relu = (flags & 1) != 0, nt_store = (flags & 2) != 0;
#ifdef DN_TUNING_ENV
    if ((flags & 1024)) __builtin_amdgcn_s_setprio(2);                          // (experiment: compute over loaders)
#endif
    const int n0 = 32 * wave;
    const int j = lane & 15, g = lane >> 4;
    // byte address of my fragment of k-step ks inside stage 0: row j (+ 16 m), piece (4 ks + g) ^ j
    // piece (4 ks + g) ^ j = (g ^ j) ^ (4 ks): with the stage bases multiples of 1 KiB, k-step ks is at (stage + off0) ^ (64 (ks & 3))
    // + 256 (ks >> 2) -- one register, an XOR with a constant and an immediate offset
    const unsigned off0 = lds_base + (unsigned)(j * kRowB + ((g ^ j) << 4));
    const int colA0 = 8 * (j >> 2) + (j & 3);                              // output column (minus n0) of A row j, MFMA tile 0
    const size_t ocol = (size_t)(n0 + 8 * g);
    bf16x8 wf[8][2];
    u32x4 bv = {0u, 0u, 0u, 0u};                                           // bias of my 8 columns (bf16 x 8)
    int cur_rel = -1;
    int32_t t_rel = 0, t_pbeg = 0, t_pend = 0;                             // record of the current tile
    f32x4 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // rows pbeg + j and pbeg + 16 + j of a tile, my 8 columns: bias, ReLU, bf16, mask, one 16-byte store each
    auto epilogue = [&](int32_t pbeg, int32_t pend) {
        const int p0 = pbeg + j, p1 = p0 + 16;
        u32x4 mk0 = {0u, 0u, 0u, 0u}, mk1 = {0u, 0u, 0u, 0u};
        if constexpr (MASK) {                                              // ReLU backward: the saved activation of my 8 columns
            if (p0 < pend) mk0 = *reinterpret_cast<const u32x4*>(mask_pos + (size_t)p0 * kH + ocol);   // (dense rows only:
            if (p1 < pend) mk1 = *reinterpret_cast<const u32x4*>(mask_pos + (size_t)p1 * kH + ocol);   //  not the hot path)
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int p = m ? p1 : p0;
            float v[8];
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) v[4 * n + i] = acc[m][n][i];
            if constexpr (EPI) {                                           // (the conv's launches carry neither bias nor ReLU)
                if (bias) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        v[2 * i] += __uint_as_float(bv[i] << 16);
                        v[2 * i + 1] += __uint_as_float(bv[i] & 0xffff0000u);
                    }
                }
                if (relu) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = dn_act(v[i], slope);
                }
            }
            u32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
            if constexpr (MASK) {                                          // keep where the saved activation is > 0, x slope elsewhere
                const u32x4 mk = m ? mk1 : mk0;
                const uint4 r4 = dn_keep_or_scale_mask(make_uint4(o[0], o[1], o[2], o[3]), make_uint4(mk[0], mk[1], mk[2], mk[3]), slope);
                o = u32x4{r4.x, r4.y, r4.z, r4.w};
            }
#ifdef DN_TUNING_ENV
            if (p < pend && !((flags & 8) && p != 0)) {                    // (flags & 8: ablation, no stores)
#else
            if (p < pend) {
#endif
#ifdef DN_TUNING_ENV
                const size_t yrow = (flags & 512) ? (size_t)(p & 0x1ffff) : (size_t)p;   // (experiment: Y inside a 64 MB window)
#else
                const size_t yrow = (size_t)p;
#endif
                u32x4* dst = reinterpret_cast<u32x4*>(Y + yrow * kH + ocol);
                if (nt_store) __builtin_nontemporal_store(o, dst);
                else *dst = o;
            }
        }
    };

    // Software pipeline over tiles: the first two k-steps' fragments of tile t + 1 are fetched during tile t, each into the
    // registers its k-step just consumed (an MFMA reads its A/B operands when it issues; the LDS answer comes >= 64 cycles
    // later), and have arrived before the loop's back edge (lgkmcnt(0) behind the epilogue: no value the compiler may copy there
    // is still in flight).  A wave therefore starts a tile's MFMAs at once; the other six k-steps' reads are issued at the top of
    // the tile and land behind those MFMAs (counted waits).  Four or eight k-steps ahead made hipcc spill weights to scratch
    // (168 registers a wave: weights 64 + fragments 64 + sums 16 + the values carried over the back edge).
#define DN_FETCH(KS, SB)                                                                                              \
    {                                                                                                                 \
        const unsigned a_ = ((SB) + off0) ^ (unsigned)(((KS) & 3) << 6);                                              \
        if ((KS) < 4) {                                                                                               \
            DN_DS_READ128(xf[KS][0], a_, 0);                                                                          \
            DN_DS_READ128(xf[KS][1], a_, 8192); /* rows 16..31 of the stage */                                        \
        } else {                                                                                                      \
            DN_DS_READ128(xf[KS][0], a_, 256);                                                                        \
            DN_DS_READ128(xf[KS][1], a_, 8448);                                                                       \
        }                                                                                                             \
    }
#ifdef DN_TUNING_ENV
#define DN_SB(SBV) ((flags & 64) ? ((desc_base - off0) & ~1023u) : (SBV))             // (flags & 64: ablation, every lane reads one word)
#else
#define DN_SB(SBV) (SBV)
#endif
    bf16x8 xf[8][2];
    u32x4 dn;                                                              // record of the next tile
    __builtin_amdgcn_s_barrier();                                          // tile 0 has landed
    {
        const unsigned a0 = desc_base;
        asm volatile("ds_read_b128 %0, %1" : "=v"(dn) : "v"(a0));
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) DN_FETCH(ks, 0u)
    }
#define DN_ARRIVED()                                                                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                               \
                 : "+v"(xf[0][0]), "+v"(xf[0][1]), "+v"(xf[1][0]), "+v"(xf[1][1]), "+v"(dn))
    DN_ARRIVED();
    t_rel = __builtin_amdgcn_readfirstlane((int)dn[0]);
    t_pbeg = __builtin_amdgcn_readfirstlane((int)dn[1]);
    t_pend = __builtin_amdgcn_readfirstlane((int)dn[2]);

    unsigned long long sc_bar = 0, sc_mfma = 0, sc_epi = 0;
    const unsigned long long c0 = DN_STAMP();
#ifdef DN_RING_STATS
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll 1
    for (int t = 0; t < nt; ++t) {
        const unsigned long long b0 = DN_STAMP();
        if ((t & 1) == 0) __builtin_amdgcn_s_barrier();                    // one per pair of tiles: tiles up to t + 2 have landed
        const unsigned long long b1 = DN_STAMP();
        DN_STAT(sc_bar, b1 - b0);
        const unsigned long long b2 = DN_STAMP();
        const unsigned an = desc_base + (unsigned)((t + 1) % kDescRing) * 16u;
        const unsigned sb = DN_SB((unsigned)(t % kNS) * kStageB);
        const unsigned nsb = DN_SB((unsigned)((t + 1) % kNS) * kStageB);
        if (t_pend > t_pbeg) {
            DN_FETCH(2, sb) DN_FETCH(3, sb) DN_FETCH(4, sb) DN_FETCH(5, sb) DN_FETCH(6, sb) DN_FETCH(7, sb)
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t_pend > t_pbeg) {
            if (t_rel != cur_rel) {                                        // wave-uniform, rare
                cur_rel = t_rel;
                const bf16_t* w = Wn + (size_t)cur_rel * kH * kH;
                if (flags & 4) {                                           // the relation's weights as the parameter stores them,
                    dn_load_w_kn32<8>(w, kH, n0, lane, wscr[wave], wf);    // [k][n]: transposed through a wave-private LDS scratch
                } else {
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
                            wf[ks][n] = *reinterpret_cast<const bf16x8*>(w + (size_t)(n0 + colA0 + 4 * n) * kH + ks * 32 + 8 * g);
                }
                if constexpr (EPI)
                    if (bias) bv = *reinterpret_cast<const u32x4*>(bias + (size_t)cur_rel * kH + n0 + 8 * g);
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                    for (int n = 0; n < 2; ++n) asm volatile("" : "+v"(wf[ks][n]));   // the wait for them stays in this branch
                if constexpr (EPI) asm volatile("" : "+v"(bv));
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#define DN_MFMA4(KS)                                                                                                  \
            _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                              \
            _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                              \
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[KS][n], xf[KS][m], acc[m][n], 0, 0, 0);      \
            __builtin_amdgcn_sched_barrier(0);
#define DN_KSTEP_A(KS)  /* fragments fetched a tile ahead; refill them for tile t + 1 */                              \
            DN_MFMA4(KS)                                                                                               \
            DN_FETCH(KS, nsb)                                                                                          \
            __builtin_amdgcn_sched_barrier(0);
#define DN_KSTEP_B(KS, CNT)  /* fragments fetched behind the barrier: behind them in the queue 2 (7 - KS) + 4 reads */  \
            asm volatile("s_waitcnt lgkmcnt(" #CNT ")" : "+v"(xf[KS][0]), "+v"(xf[KS][1]));                            \
            DN_MFMA4(KS)
#ifdef DN_TUNING_ENV
            if (flags & 32) {                                              // (ablation: no MFMAs)
                DN_FETCH(0, nsb) DN_FETCH(1, nsb)
                asm volatile("ds_read_b128 %0, %1" : "=v"(dn) : "v"(an));
            } else
#endif
            {
            DN_KSTEP_A(0) DN_KSTEP_A(1)
            DN_KSTEP_B(2, 14) DN_KSTEP_B(3, 12) DN_KSTEP_B(4, 10) DN_KSTEP_B(5, 8) DN_KSTEP_B(6, 6) DN_KSTEP_B(7, 4)
            asm volatile("ds_read_b128 %0, %1" : "=v"(dn) : "v"(an));    // (behind the last k-step: its registers are free)
            }
#undef DN_KSTEP_A
#undef DN_KSTEP_B
#undef DN_MFMA4
            const unsigned long long b3 = DN_STAMP();
            DN_STAT(sc_mfma, b3 - b2);
            epilogue(t_pbeg, t_pend);
            DN_STAT(sc_epi, DN_STAMP() - b3);
        } else {
            DN_FETCH(0, nsb) DN_FETCH(1, nsb)
            asm volatile("ds_read_b128 %0, %1" : "=v"(dn) : "v"(an));
        }
        DN_ARRIVED();
        t_rel = __builtin_amdgcn_readfirstlane((int)dn[0]);
        t_pbeg = __builtin_amdgcn_readfirstlane((int)dn[1]);
        t_pend = __builtin_amdgcn_readfirstlane((int)dn[2]);
    }
#undef DN_FETCH
#undef DN_SB
#undef DN_ARRIVED
#ifdef DN_RING_STATS
    if (wave == 0 && lane == 0 && blockIdx.x < 256) {
        g_ring_stats[blockIdx.x][0] = DN_STAMP() - c0; g_ring_stats[blockIdx.x][1] = sc_bar;
        g_ring_stats[blockIdx.x][2] = sc_mfma; g_ring_stats[blockIdx.x][3] = sc_epi;
        g_ring_stats[blockIdx.x][8] = __builtin_amdgcn_s_memrealtime() - rt0; g_ring_stats[blockIdx.x][9] = rt0;
    }
#endif
}
#undef DN_DS_READ128

}  // namespace

namespace dn_internal {

int launch_transform_ring256(const void* X_, const void* X2, int32_t n1, const int32_t* idx, const void* Wn, const void* bias,
                             int32_t relu, int32_t nt_store, const void* mask_pos, const int32_t* tiles, int64_t num_tiles,
                             int64_t tiles_per_wg, void* Y, int32_t w_kn, float slope, hipStream_t st) {
#ifdef DN_TUNING_ENV
    if (tiles_per_wg <= 0 && dn_knob("DN_TF_RING_WGS", 0) > 0)             // (experiment: a table laid out [workgroups][tiles])
        tiles_per_wg = dn_cdiv(num_tiles, dn_knob("DN_TF_RING_WGS", 0));
#endif
    if (tiles_per_wg <= 0) tiles_per_wg = dn_cdiv(num_tiles, 256);       // one persistent workgroup per CU
    const int64_t grid = dn_cdiv(num_tiles, tiles_per_wg);
    const int abl = dn_knob("DN_TF_ABL", 0);   // tuning build only (read per call): 1 no stores, 2 gathers hit L2, 4 no MFMAs, 8 trivial LDS reads, 16 no row DMAs, 64 Y in a 64 MB window
    if (dn_knob("DN_TF_NT", 1) == 0) nt_store = 0;
    const int32_t flags = (relu ? 1 : 0) | (nt_store ? 2 : 0) | (w_kn ? 4 : 0) | ((abl & 255) << 3);
#define DN_RING_LAUNCH(M, I, E, X)                                                                                      \
    hipLaunchKernelGGL((rows_transform_ring_kernel<M, I, E, X>), dim3((unsigned)grid), dim3(kThreads), 0, st, (const bf16_t*)X_, \
                       (const bf16_t*)X2, n1, idx, (const bf16_t*)Wn, (const bf16_t*)bias, flags, (const bf16_t*)mask_pos,      \
                       reinterpret_cast<const Tile*>(tiles), (int32_t)num_tiles, (int32_t)tiles_per_wg, (bf16_t*)Y, slope)
    // the conv's launches (gathered rows, no epilogue, one source) get the leanest instruction stream: the loop is bound by
    // vector-instruction issue, not by the matrix pipe (~180 VALU instructions per SIMD and tile before this split)
    const bool epi = bias != nullptr || relu != 0;
    const bool x2 = X2 != nullptr;
    if (mask_pos) {
        if (idx) DN_RING_LAUNCH(true, true, true, true);
        else DN_RING_LAUNCH(true, false, true, false);
    } else if (idx) {
        if (!epi && !x2) DN_RING_LAUNCH(false, true, false, false);
        else if (!epi) DN_RING_LAUNCH(false, true, false, true);
        else DN_RING_LAUNCH(false, true, true, true);
    } else {
        if (!epi) DN_RING_LAUNCH(false, false, false, false);
        else DN_RING_LAUNCH(false, false, true, false);
    }
#undef DN_RING_LAUNCH
    DN_CHECK_LAUNCH();
    return DN_OK;
}

}  // namespace dn_internal

#ifdef DN_RING_STATS
extern "C" int dn_debug_ring_stats(unsigned long long* out) {              // tuning build only: 256 x 10 counters of the last launch
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ring_stats), sizeof(g_ring_stats)) == hipSuccess ? 0 : -2;
}
#endif
