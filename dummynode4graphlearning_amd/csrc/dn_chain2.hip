// dn_rows_chain2_bf16 at H = 256: two dense layers in one pass over the rows, rebuilt on the skeleton of the ring kernels.
//
//   Y1 = epi1(m0(X) @ W1n^T),  Y2 = epi2(Y1 @ W2n^T)        (semantics: dn_rel.hip, rows_chain2_kernel -- still the H = 64 / 128 path)
//
// The register-staged kernel it replaces spent a tile's time in LDS: 16 waves x 16 KB of fragment reads per 32-row tile and layer,
// the X tile staged through registers into a padded image, both results written to LDS as 8-byte pieces at a 528-byte row
// stride (2-way bank conflicts: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.44) and read back for the 16-byte global stores,
// and one __syncthreads() per tile draining every global access: 348-370 us for 1.56 GB at config 5 (4.3-4.5 TB/s).
//
// Here one workgroup per CU, 8 waves at 2 per SIMD (256 VGPRs a wave), so a wave owns 64 output columns of ONE layer -- its
// weight slice in 128 VGPRs -- and a tile's rows are read by 4 waves per layer instead of 8 or 16 (LDS fragment traffic per tile:
// 128 KB instead of 256-512 KB):
//   * waves 0-3 run layer 1 on tile t while waves 4-7 run layer 2 on tile t - 1 (its input: layer 1's result of the previous
//     iteration in a double-buffered LDS tile), IN OPPOSITE PHASE: layer 1 multiplies, then runs its epilogue; layer 2 first
//     finishes the sums of the previous iteration (kept in registers across the barrier), then multiplies -- on every SIMD one
//     wave is in its MFMAs while the other is in its epilogue's VALU instructions; one raw s_barrier per tile joins them;
//   * the X tiles come by LDS-DMA (two 1 KiB wave-instructions per wave and tile, no staging VGPRs) into a ring of kNS stages,
//     kD tiles requested ahead; the image is the ring kernels' (unpadded, 16-byte pieces XOR-swizzled on the SOURCE address:
//     LDS (row r, position q) holds piece q ^ (r & 15)), fragments are 16 ds_read_b128 per tile and wave with hand-counted
//     lgkmcnt waits, the second half of a tile's k-steps fetched into the registers the first half just consumed;
//   * results leave STRAIGHT from the accumulators: the A rows of the MFMA tiles are the weight rows permuted so that a lane ends
//     up with 8 consecutive columns of a row = one 16-byte store (bias, activation, masks and sign bits applied in registers);
//     layer 1 writes the same 16-byte piece into the hand-off tile in the swizzled image, so layer 2 reads it like an X tile;
//   * the sign-bit outputs are collected in 1 KiB LDS tiles and leave a tile later, a quarter of the tile per wave of the layer;
//   * the backward form (KEEP): the activation-mask bit tensors (mask0 on the input, mask1 on layer 1's result) travel as one
//     1 KiB DMA per tile and mask next to the rows; mask0 is applied to a landed tile in place one tile AHEAD of its use (no
//     second barrier; the counted wait then asks for tile t + 1), mask1 in layer 1's epilogue from the tile's bits in LDS.
//     (Built before the epilogue diet below it measured 419 us against the register-staged kernel's 397 and was shelved; with the
//     diet the forward form runs 305 us in the step against ~370, and this form followed: docs/LAB_NOTES.md, round 4.)
// vmcnt counts loads, stores and LDS-DMAs of a wave together, in issue order.  So that "tile t has landed" stays a COUNTED wait
// with kD tiles in flight, every wave issues the same vector-memory operations per tile -- 2 row DMAs, then 4 result stores
// (rows past the end and the idle role of the first / last iterations store into a dump area instead of being predicated
// off; with sign bits a fifth store, every wave its quarter of the tile) -- and the extra operations of the backward form (the
// mask-bit DMAs, one wave per tile and mask) only make that wave's wait stricter.
#include "dn_common.h"
#include "dn_internal.h"

namespace {

typedef __bf16 bf16_t;
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kH = 256;
constexpr int kRowB = 2 * kH;                 // bytes per row
constexpr int kTR = 32;                       // rows per tile
constexpr int kTileB = kTR * kRowB;           // 16 KiB
constexpr int kBitsB = kTR * (kH / 8);        // 1 KiB: one tile of mask / sign bits
constexpr int kStageB = kTileB + 2 * kBitsB;   // an X stage: the rows, the tile's mask0 bits, its mask1 bits (backward form)
constexpr int kD = 4;                         // tiles requested ahead of the one being multiplied
constexpr int kNS = kD + 1;                   // X ring stages (90 KiB)
constexpr int kThreads = 512;
constexpr int kDma = 2, kStores = 4;          // vector-memory operations every wave issues per tile, in this order (+ 1 store with sign bits)

__device__ __attribute__((aligned(16))) uint4 g_c2_zero[64];       // 1 KiB of zeros: what rows past the end read
__device__ __attribute__((aligned(16))) uint4 g_c2_dump[64 * 8];   // where their results go (1 KiB per wave, never read)

__device__ __forceinline__ uint32_t pack2(float a, float b) {
    typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v;
    v[0] = (bf16_t)a;
    v[1] = (bf16_t)b;
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ uint32_t pos_bits(const u32x4& v) {     // bit i: bf16 element i > 0 (sign clear, magnitude non-zero)
    uint32_t bits = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t lo = v[i] & 0xffffu, hi = v[i] >> 16;
        bits |= ((lo != 0u && lo < 0x8000u) ? 1u : 0u) << (2 * i);
        bits |= ((hi != 0u && hi < 0x8000u) ? 1u : 0u) << (2 * i + 1);
    }
    return bits;
}

// "the rows of tile t have landed" at the top of iteration t: all but the operations issued after them may be outstanding.  Issue
// order of a wave: rows of tiles 0 .. kD-1 (prologue), then per iteration [rows of tile t + kD][ST stores].  A tile still from
// the prologue has (kD - 1 - t) later prologue tiles and t full iterations behind it.  (The mask-bit DMAs of the backward form --
// one wave per tile and mask -- are extra operations behind the rows they belong to: they only make that wave's wait stricter.)
template <int ST, bool AHEAD>                                             // ST: result stores a wave issues per tile (4, + 1 with the sign bits);
__device__ __forceinline__ void c2_wait(int t) {                          // AHEAD: tile t + 1 must have landed (the mask pass works a tile ahead)
    constexpr int A = AHEAD ? 1 : 0, IT = kDma + ST;
    if constexpr (0 + A < kD) if (t == 0) { wait_vmcnt<kDma * (kD - 1 - A)>(); return; }
    if constexpr (1 + A < kD) if (t == 1) { wait_vmcnt<kDma * (kD - 2 - A) + IT>(); return; }
    if constexpr (2 + A < kD) if (t == 2) { wait_vmcnt<kDma * (kD - 3 - A) + 2 * IT>(); return; }
    if constexpr (3 + A < kD) if (t == 3) { wait_vmcnt<kDma * (kD - 4 - A) + 3 * IT>(); return; }
    wait_vmcnt<ST + IT * (kD - 1 - A)>();
}
static_assert(kD == 4, "c2_wait spells out the first kD iterations");

#define DN_C2_READ128(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))

template <bool KEEP, bool SBITS>
__global__ __launch_bounds__(kThreads) void rows_chain2_ring_kernel(
    const bf16_t* __restrict__ X, const bf16_t* __restrict__ W1n, const bf16_t* __restrict__ b1, const bf16_t* __restrict__ W2n,
    const bf16_t* __restrict__ b2, int32_t flags, const uint8_t* __restrict__ mask0, const uint8_t* __restrict__ mask1, int32_t N,
    int32_t num_tiles, bf16_t* __restrict__ Y1, bf16_t* __restrict__ Y2, uint8_t* __restrict__ bits1, uint8_t* __restrict__ bits2,
    float slope) {
    __shared__ __attribute__((aligned(1024))) char lds[kNS * kStageB + 2 * kTileB + 4 * kBitsB];
    typedef __attribute__((address_space(3))) char* lds_wp;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_wp)lds;
    constexpr unsigned kH1Off = kNS * kStageB;                            // the two hand-off tiles (layer 1 -> layer 2)
    constexpr unsigned kObOff = kH1Off + 2 * kTileB;                      // sign-bit tiles [layer][2]
    static_assert(kStageB % 1024 == 0 && kH1Off % 1024 == 0, "the fragment addressing XORs bits 6-7 of a stage-relative offset");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 2, wq = wave & 3;                            // role 0: layer 1, 1: layer 2; 64 columns each
    const int c0 = 64 * wq;
    const int j = lane & 15, g = lane >> 4;
    const int first = (int)blockIdx.x, step = (int)gridDim.x;             // tiles dealt round-robin: the launch sweeps HBM like one stream
    const int nt = first < num_tiles ? (num_tiles - first + step - 1) / step : 0;
    if (nt <= 0) return;
    auto rowbase = [&](int T) -> int64_t { return ((int64_t)first + (int64_t)T * step) * kTR; };   // (>= N for T >= nt)
    const bool nt_store = (flags & 4) != 0;
    const bool relu = role ? (flags & 2) != 0 : (flags & 1) != 0;

    // ---- my weight slice: 2 halves of 32 columns, each in the ring kernels' column order (A row i of MFMA tile n <-> column
    //      8 (i >> 2) + 4 n + (i & 3) of the half), so that a lane ends up with 8 consecutive columns per half
    const bf16_t* Wn = role ? W2n : W1n;
    bf16x8 wf[8][4];
    if (flags & (role ? 16 : 8)) {                                       // stored [k][n]: transposed through a wave-private scratch
        char* scr = lds + kH1Off + wave * 2048;                           // (the hand-off tiles are idle until the first barrier)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            dn_bf16x8 tmp[8][2];
            dn_load_w_kn32<8>(Wn, kH, c0 + 32 * h, lane, scr, tmp);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                for (int n = 0; n < 2; ++n) wf[ks][2 * h + n] = tmp[ks][n];
        }
    } else {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int col = c0 + 32 * (n >> 1) + 8 * (j >> 2) + (j & 3) + 4 * (n & 1);
                wf[ks][n] = *reinterpret_cast<const bf16x8*>(Wn + (size_t)col * kH + ks * 32 + 8 * g);
            }
    }
    const bf16_t* bb = role ? b2 : b1;
    f32x4 biasf[4];                                                       // bias of my 2 x 8 columns as the C operand of the first k-step:
#pragma unroll                                                            // biasf[n][i] = column 8 g + 4 (n & 1) + i of half n >> 1
    for (int n = 0; n < 4; ++n) biasf[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (bb) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const u32x4 bvh = *reinterpret_cast<const u32x4*>(bb + c0 + 32 * h + 8 * g);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                biasf[2 * h + (e >> 2)][e & 3] = __uint_as_float((e & 1) ? (bvh[e >> 1] & 0xffff0000u) : (bvh[e >> 1] << 16));
        }
    }
    wait_vmcnt<0>();                                                      // no ordinary load may be pending once the DMAs start
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int n = 0; n < 4; ++n) asm volatile("" : "+v"(wf[ks][n]));
    asm volatile("" : "+v"(biasf[0]), "+v"(biasf[1]), "+v"(biasf[2]), "+v"(biasf[3]));

    // ---- DMA side: rows 4 w .. 4 w + 3 of a tile are this wave's (two wave-instructions of 2 rows)
    const int rin = lane >> 5, pos = lane & 31;
    const char* zero = reinterpret_cast<const char*>(g_c2_zero);
    auto issue = [&](int T) __attribute__((always_inline)) {
        const unsigned st = lds_base + (unsigned)(T % kNS) * kStageB;
        const int64_t rb = rowbase(T);
#pragma unroll
        for (int jj = 0; jj < kDma; ++jj) {
            const int rl = 4 * wave + 2 * jj + rin;                       // row of the tile this lane fills
            const int64_t p = rb + rl;
            const char* src = (p < N ? reinterpret_cast<const char*>(X) + p * kRowB : zero) + ((pos ^ (rl & 15)) << 4);
            glds16(src, st + (unsigned)(4 * wave + 2 * jj) * kRowB);      // lane l lands at + 16 l
        }
        if constexpr (KEEP) {                                             // the tile's 1 KiB of mask0 / mask1 bits: one full-width DMA each
            const int64_t p = rb + (lane >> 1);
            const size_t boff = (size_t)p * (kH / 8) + (lane & 1) * 16;
            if (wave == (T & 7)) glds16(p < N ? reinterpret_cast<const char*>(mask0) + boff : zero + lane * 16, st + kTileB);
            if (wave == ((T + 4) & 7)) glds16(p < N ? reinterpret_cast<const char*>(mask1) + boff : zero + lane * 16, st + kTileB + kBitsB);
        }
    };
    // mask0 on a landed tile, in place: a thread's two pieces (row r, position q) hold source piece q ^ (r & 15).  Keep where the
    // bit is set, x slope (0: zero) elsewhere; the word masks come from signed 1-bit field extracts (0 / ~0) and one bit-select.
    auto mask_tile = [&](int T) __attribute__((always_inline)) {
        char* sT = lds + (T % kNS) * kStageB;
        const uint8_t* sB = reinterpret_cast<const uint8_t*>(sT + kTileB);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int piece = tid + kThreads * jj, r = piece >> 5, q = piece & 31;
            u32x4* pp = reinterpret_cast<u32x4*>(sT + r * kRowB + q * 16);
            const int32_t kb = (int32_t)sB[r * (kH / 8) + (q ^ (r & 15))];
            u32x4 v = *pp;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe(kb, 2 * i, 1), m1 = (uint32_t)__builtin_amdgcn_sbfe(kb, 2 * i + 1, 1);
                const uint32_t mk = (m0 & 0xffffu) | (m1 & 0xffff0000u);
                uint32_t alt = 0u;
                if (slope != 0.f) alt = pack2(__uint_as_float(v[i] << 16) * slope, __uint_as_float(v[i] & 0xffff0000u) * slope);
                v[i] = (v[i] & mk) | (alt & ~mk);
            }
            *pp = v;
        }
    };
    // ---- compute side
    const unsigned off0 = (unsigned)(j * kRowB + ((g ^ j) << 4));         // my fragment of k-step 0, rows j and 16 + j (+ 8192)
    bf16_t* Yout = role ? Y2 : Y1;
    char* dump = reinterpret_cast<char*>(g_c2_dump) + wave * 1024 + lane * 16;
    f32x4 acc[2][4];
    bf16x8 xf[4][2];

#define DN_C2_FETCH(SLOT, KS)                                                                                             \
    {                                                                                                                 \
        const unsigned a_ = (sb + off0) ^ (unsigned)(((KS) & 3) << 6);                                                \
        if ((KS) < 4) {                                                                                               \
            DN_C2_READ128(xf[SLOT][0], a_, 0);                                                                        \
            DN_C2_READ128(xf[SLOT][1], a_, 8192);                                                                     \
        } else {                                                                                                      \
            DN_C2_READ128(xf[SLOT][0], a_, 256);                                                                      \
            DN_C2_READ128(xf[SLOT][1], a_, 8448);                                                                     \
        }                                                                                                             \
    }
#define DN_C2_MFMA8(SLOT, KS)                                                                                             \
    _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                                      \
    _Pragma("unroll") for (int n = 0; n < 4; ++n)                                                                      \
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[KS][n], xf[SLOT][m], acc[m][n], 0, 0, 0);             \
    __builtin_amdgcn_sched_barrier(0);
#define DN_C2_KSTEP(SLOT, KS, CNT)                                                                                        \
    asm volatile("s_waitcnt lgkmcnt(" #CNT ")" : "+v"(xf[SLOT][0]), "+v"(xf[SLOT][1]));                               \
    DN_C2_MFMA8(SLOT, KS)
    // one layer on one 32-row tile in the swizzled image at LDS byte address sb: sums start from the bias
    auto mfma_block = [&](unsigned sb) __attribute__((always_inline)) {
        // (older LDS operations of this wave -- the sign-bit read, the previous epilogue's byte writes -- may still be pending: they
        //  complete first, so the counted waits below only get stricter)
        __builtin_amdgcn_sched_barrier(0);
        DN_C2_FETCH(0, 0) DN_C2_FETCH(1, 1) DN_C2_FETCH(2, 2) DN_C2_FETCH(3, 3)
        __builtin_amdgcn_sched_barrier(0);
        // reads in the queue behind the pair a k-step waits for: 6 while the second half is being fetched, then 6, 4, 2, 0
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(xf[0][0]), "+v"(xf[0][1]));
#pragma unroll
        for (int m = 0; m < 2; ++m)                                        // (the sums start from the bias: no initialisation pass)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][n], xf[0][m], biasf[n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        DN_C2_FETCH(0, 4) __builtin_amdgcn_sched_barrier(0);
        DN_C2_KSTEP(1, 1, 6) DN_C2_FETCH(1, 5) __builtin_amdgcn_sched_barrier(0);
        DN_C2_KSTEP(2, 2, 6) DN_C2_FETCH(2, 6) __builtin_amdgcn_sched_barrier(0);
        DN_C2_KSTEP(3, 3, 6) DN_C2_FETCH(3, 7) __builtin_amdgcn_sched_barrier(0);
        DN_C2_KSTEP(0, 4, 6)
        DN_C2_KSTEP(1, 5, 4)
        DN_C2_KSTEP(2, 6, 2)
        DN_C2_KSTEP(3, 7, 0)
    };
#undef DN_C2_KSTEP
#undef DN_C2_MFMA8
#undef DN_C2_FETCH
    // rows j and 16 + j of tile Te, my 2 x 8 columns, straight from the accumulators: activation, mask, sign bits, one 16-byte
    // store each (+ layer 1: the same piece into hand-off tile `par` in the swizzled image)
    // The launch is bound by the VALU instructions of this epilogue (two waves a SIMD: ~400 of them per wave and tile next to 64
    // MFMAs of 16 cycles measured 420 us; without the sign bits 312), so everything is in its cheapest form: one max per element
    // (two for a leaky slope in (0, 1]), the sign bit as a clamp of the fp32 pattern to [0, 1] + shift-or, lane-constant offsets
    // with immediates for the four pieces, the bias as the C operand of the first k-step.
    const unsigned yoff0 = (unsigned)(j * kRowB + (8 * wq + g) * 16);     // my piece of row j of a tile in Y: + 64 h, + 8192 m
    const unsigned hoff0 = (unsigned)(j * kRowB + (((8 * wq + g) ^ j) << 4));   // ... in the hand-off image: ^ 64 h, + 8192 m
    const unsigned boff0 = (unsigned)(j * (kH / 8) + 8 * wq + g);         // ... in a sign-bit tile: + 4 h, + 512 m
    const int act_mode = !relu ? 0 : (slope == 0.f ? 1 : ((slope > 0.f && slope <= 1.f) ? 2 : 3));
    auto epi_block = [&](int Te, int par) __attribute__((always_inline)) {
        const bool live = Te >= 0 && Te < nt;
        const int64_t rb = live ? rowbase(Te) : (int64_t)N;               // (idle role / past the end: every row is "past the end")
        char* ybase = reinterpret_cast<char*>(Yout) + rb * kRowB + yoff0;
        char* h1w = lds + kH1Off + par * kTileB + hoff0;
        uint8_t* obL = reinterpret_cast<uint8_t*>(lds + kObOff + (2 * role + par) * kBitsB + boff0);
        const uint8_t* keepL = reinterpret_cast<const uint8_t*>(lds + ((Te + kNS) % kNS) * kStageB + kTileB + kBitsB + boff0);   // mask1 bits of tile Te (layer 1)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const bool ok = rb + 16 * m + j < N;
            char* yrow = ok ? ybase + 8192 * m : dump;                     // (rows past the end: both pieces into the dump slot)
            const int ystep = ok ? 64 : 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float v[8];
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[4 * n + i] = acc[m][2 * h + n][i];
                // (inline asm: fmaxf() costs a second v_max to quiet a signalling NaN first)
                if (act_mode == 1) {                                       // (wave-uniform)
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm("v_max_f32 %0, %1, 0" : "=v"(v[i]) : "v"(v[i]));          // (NaN -> 0, as dn_act)
                } else if (act_mode == 2) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float vs = v[i] * slope;
                        asm("v_max_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(vs));
                    }
                } else if (act_mode == 3) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = dn_act(v[i], slope);
                }
                if constexpr (KEEP) {
                    if (role == 0) {                                       // mask1: keep where the bit is set, x slope (0: zero) elsewhere
                        const int32_t kb = (int32_t)keepL[512 * m + 4 * h];
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const uint32_t mk = (uint32_t)__builtin_amdgcn_sbfe(kb, i, 1);
                            const uint32_t alt = slope != 0.f ? __float_as_uint(v[i] * slope) : 0u;
                            v[i] = __uint_as_float((__float_as_uint(v[i]) & mk) | (alt & ~mk));
                        }
                    }
                }
                u32x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = pack2(v[2 * i], v[2 * i + 1]);
                if constexpr (SBITS) {
                    // element > 0 <=> its fp32 pattern is a positive integer (a NaN with a clear sign included, as for the stored
                    // bf16; a positive fp32 rounds to a positive bf16 or to +0 only below 2^-134, which no activation reaches)
                    uint32_t bits = 0;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        uint32_t c;
                        asm("v_med3_i32 %0, %1, 0, 1" : "=v"(c) : "v"(v[i]));                  // clamp the pattern to [0, 1]
                        asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(bits) : "v"(c), "n"(i), "v"(bits));
                    }
                    obL[512 * m + 4 * h] = (uint8_t)bits;                  // (one dword write after trading the bytes between the four
                    //  lanes of a row -- three cross-lane reads -- measured the same: the byte stores are not what the bits cost)
                }
                if (role == 0) *reinterpret_cast<u32x4*>(reinterpret_cast<uintptr_t>(h1w + 8192 * m) ^ (uintptr_t)(64 * h)) = o;
                u32x4* dst = reinterpret_cast<u32x4*>(yrow + h * ystep);
                if (nt_store) __builtin_nontemporal_store(o, dst);
                else *dst = o;
            }
        }
    };
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int T = 0; T < kD; ++T) issue(T);
    if constexpr (KEEP) {
        wait_vmcnt<kDma*(kD - 1)>();                                      // tile 0 (and its bits) have landed
        __builtin_amdgcn_s_barrier();
        mask_tile(0);
    }

#pragma unroll 1
    for (int t = 0; t <= nt + 1; ++t) {
        c2_wait<kStores + (SBITS ? 1 : 0), KEEP>(t);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // my LDS reads / writes of the previous iteration are done
        __builtin_amdgcn_s_barrier();
        issue(t + kD);                                                     // into the stage tile t - 1 used
        if constexpr (KEEP) mask_tile(t + 1);                              // (a tile past the end is zeros and stays zeros)
        // Sign bits collected during the previous iteration (layer 1's of tile t - 1, layer 2's of tile t - 3): every wave of the
        // layer takes a quarter of the 1 KiB tile (16 lanes x 16 bytes), behind its MFMA block (the other wave of the SIMD is busy
        // meanwhile), so that no fifth store on ONE wave's vmcnt queue (which would make that wave wait for a tile
        // more than the others, and the others for it at the barrier) is on the iteration's critical path.
        auto put_bits = [&]() __attribute__((always_inline)) {
            if constexpr (SBITS) {
                if (lane < 16) {                                           // (always issued: the count above relies on it)
                    const int T = t - 1 - 2 * role;
                    const int q = 16 * wq + lane;                          // 16-byte piece of the tile: row q >> 1, half q & 1
                    const int64_t p = (T >= 0 && T < nt ? rowbase(T) : (int64_t)N) + (q >> 1);
                    uint8_t* bo = role ? bits2 : bits1;
                    char* dst = p < N ? reinterpret_cast<char*>(bo) + p * (kH / 8) + (q & 1) * 16 : dump;
                    *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(lds + kObOff + (2 * role + ((t - 1) & 1)) * kBitsB + q * 16);
                }
            }
        };
        // The two roles run their halves of an iteration in OPPOSITE order -- layer 1: products of tile t, then its epilogue
        // (the hand-off tile must be complete at the next barrier); layer 2: the epilogue of the sums it formed in the PREVIOUS
        // iteration (tile t - 2, kept in registers across the barrier), then the products of tile t - 1 -- so that on every SIMD
        // one wave is in its MFMAs while the other is in its ~250 epilogue VALU instructions.  (Both in the same order: the
        // two waves of a SIMD fight for the matrix pipe, then for the VALU: +12 us per launch over the old kernel.)
        if (role == 0) {
            mfma_block(lds_base + (unsigned)(t % kNS) * kStageB);
            put_bits();
            epi_block(t, t & 1);
        } else {
            epi_block(t - 2, t & 1);
            mfma_block(lds_base + kH1Off + (unsigned)((t - 1) & 1) * kTileB);
            put_bits();
        }
    }
    if constexpr (SBITS) {                                                  // layer 2's sign bits of the last tile
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (role == 1 && lane < 16) {
            const int q = 16 * wq + lane;
            const int64_t p = rowbase(nt - 1) + (q >> 1);
            const uint4 v = *reinterpret_cast<const uint4*>(lds + kObOff + (2 + ((nt + 1) & 1)) * kBitsB + q * 16);
            if (p < N) *reinterpret_cast<uint4*>(bits2 + p * (kH / 8) + (q & 1) * 16) = v;
        }
    }
    wait_vmcnt<0>();                                                       // the DMAs of the tiles past the end: nothing may land later
}

}  // namespace

namespace dn_internal {

bool chain2_ring_supported(bool has_mask0, bool has_mask1, bool has_bits1, bool has_bits2) {
    if (has_mask0 != has_mask1 || has_bits1 != has_bits2) return false;
    return !(has_mask0 && has_bits1);                                       // forward: no masks; backward: both masks, no sign bits
}

int launch_chain2_ring256(const void* X, const void* W1n, const void* b1, const void* W2n, const void* b2, int32_t flags,
                          const void* mask0, const void* mask1, int64_t N, void* Y1, void* Y2, void* bits1, void* bits2,
                          float slope, hipStream_t st) {
    const int64_t num_tiles = dn_cdiv(N, (int64_t)kTR);
    const unsigned grid = (unsigned)(num_tiles < 256 ? num_tiles : 256);  // one workgroup per CU (126 KiB of LDS)
    const bf16_t *x = (const bf16_t*)X, *w1 = (const bf16_t*)W1n, *w2 = (const bf16_t*)W2n, *bb1 = (const bf16_t*)b1,
                 *bb2 = (const bf16_t*)b2;
    const uint8_t *m0 = (const uint8_t*)mask0, *m1 = (const uint8_t*)mask1;
    bf16_t *y1 = (bf16_t*)Y1, *y2 = (bf16_t*)Y2;
    uint8_t *o1 = (uint8_t*)bits1, *o2 = (uint8_t*)bits2;
    if (m0 != nullptr)
        hipLaunchKernelGGL((rows_chain2_ring_kernel<true, false>), dim3(grid), dim3(kThreads), 0, st, x, w1, bb1, w2, bb2, flags,
                           m0, m1, (int32_t)N, (int32_t)num_tiles, y1, y2, o1, o2, slope);
    else if (o1 != nullptr)
        hipLaunchKernelGGL((rows_chain2_ring_kernel<false, true>), dim3(grid), dim3(kThreads), 0, st, x, w1, bb1, w2, bb2, flags,
                           m0, m1, (int32_t)N, (int32_t)num_tiles, y1, y2, o1, o2, slope);
    else
        hipLaunchKernelGGL((rows_chain2_ring_kernel<false, false>), dim3(grid), dim3(kThreads), 0, st, x, w1, bb1, w2, bb2, flags,
                           m0, m1, (int32_t)N, (int32_t)num_tiles, y1, y2, o1, o2, slope);
    DN_CHECK_LAUNCH();
    return DN_OK;
}

}  // namespace dn_internal
