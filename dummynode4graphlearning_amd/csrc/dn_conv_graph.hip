// One launch per direction of a relational conv on a batch of SMALL graphs at the reference's default width (H = 64, bf16):
//
//   out[v, :] = sum_{e: key_out(e) = v} x[key_in(e), :] @ W[etype(e)]  +  x[v, :] @ W_loop  (+ bias)
//
// i.e. rgin.py:102-160 / rgcn.py:160-196 (message UDF + fn.sum + the self loop of the apply UDF) forward with (key_in, key_out) =
// (src, dst), and the input gradient with (dst, src) and the weights read transposed.  At config 3 of BASELINE.json (512 graphs of
// 50 nodes, R = 8, H = 64: the reference CLI's default `--hid_dim`, subgraph_isomorphism/config.py:456-461) the row-factorised
// path needs three launches per direction -- gathered-row transform, closing launch, fold tail -- for 67 MB of algorithmic
// bytes: launch-bound.  Here ONE workgroup takes ONE graph from the batch's raw arrays (node_ptr / edge_ptr / src / dst / etype:
// no row index, no Y round trip):
//   * 4 wavefronts, wavefront w owns output columns 16 w .. 16 w + 15 and keeps its slice of EVERY relation's weights (and of
//     W_loop) in registers for the whole launch: 8 VGPRs per matrix;
//   * the graph's x rows go to LDS once (<= 64 rows x 128 bytes, 16-byte pieces XOR-swizzled by the row); the edges are bucketed
//     by relation in LDS, STABLY (ballot ranks in edge order: the summation order is fixed, results are bitwise repeatable), every
//     bucket padded to 32 edges with null edges;
//   * per 32 edges of one relation:  P[e][n] = x[in(e)] . W_r  as two 16-edge MFMAs whose A fragments are indexed row reads of the
//     LDS image (no gather buffer), then  out[node][n] += sum_e [out(e) = node] P[e][n]  as one more MFMA per 16 nodes whose B
//     operand is P STRAIGHT FROM THE ACCUMULATORS (rounded to bf16: the storage point of the row-factorised path's Y): the K index
//     of that product is an edge slot, and slot (g, t) of lane group g is defined as the edge the first product left in that
//     lane's registers (t < 4: edge 4 g + t, else 16 + 4 g + t - 4) -- the selection matrix is built to match, so no transpose
//     through LDS is needed;
//   * the self loop is the same MFMA on the node rows; finished rows leave through LDS as 16-byte stores.
// Every relation is taken edge by edge (the dummy relations' 2 n edges per graph too: ~4 more 32-edge blocks per graph, noise),
// so the launch needs none of the row factorisation.  The weight gradient still runs on the row index (dn_rows_wgrad_bf16), whose
// collapsed relations take per-graph column sums of x / of the gradient as operands: the launch writes them as a by-product (aux).
// Numerics: fp32 accumulation in a fixed order, one bf16 rounding of every per-edge product (as Y), one of the output row.
#include "dn_common.h"
#include "../../include/dn_hip.h"

namespace {

typedef __bf16 bf16_t;
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32;

constexpr int kH = 64, kWaves = kH / 16, kThreads = 64 * kWaves;
constexpr int kMaxNodes = 64;          // nodes of a graph (4 node blocks of 16)
constexpr int kMaxEdges = 1024;        // edges of a graph
constexpr int kRowB = 2 * kH;          // 128 bytes per row
constexpr int kZeroRow = kMaxNodes;    // row 64 of the image: zeros (the input row of a null edge)

__device__ __forceinline__ u32 pack2(float a, float b) {
    typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v;
    v[0] = (bf16_t)a;
    v[1] = (bf16_t)b;
    return __builtin_bit_cast(u32, v);
}

struct GraphLds {
    __attribute__((aligned(16))) char x[(kMaxNodes + 1) * kRowB];          // the conv's input rows + the zero row, swizzled
    __attribute__((aligned(16))) char o[kMaxNodes * kRowB];                // a tile of finished rows (swizzled like x)
    __attribute__((aligned(16))) char h[kMaxNodes * kRowB];                // the MLP's middle rows (layer modes)
    u32 raw[kMaxEdges];                                                    // rel << 16 | out << 8 | in  (local node numbers)
    uint8_t e_in[kMaxEdges + 32 * 16 + 64], e_out[kMaxEdges + 32 * 16 + 64];   // bucketed by relation, each bucket a multiple of 32
    int32_t cnt[17];
    u32 mbits[kMaxNodes * 2];                                              // mode 2: the graph's rows of the inner activation's sign bits
    float colsum[4][kH];
    __attribute__((aligned(16))) char wscr[kWaves][1024];
};

// an image row r lies at r * 128, its 16-byte piece p at position p ^ (r & 7)
__device__ __forceinline__ int sw_off(int row, int piece) { return row * kRowB + ((piece ^ (row & 7)) << 4); }
__device__ __forceinline__ const bf16x8& x_frag(const char* img, int row, int piece) {
    return *reinterpret_cast<const bf16x8*>(img + sw_off(row, piece));
}
__device__ __forceinline__ float act(float v, float slope) { return v > 0.f ? v : v * slope; }

// Mode 1 / 2: the layer's MLP around the conv (rgin.py:50-57,147-151: Linear - act - Linear - act, act = ReLU / leaky ReLU).
struct LayerArgs {
    const bf16_t *W1, *b1, *W2, *b2;   // Linear weights [out][in] as the parameters store them; biases may be NULL
    bf16_t *R0, *R1;                   // mode 1: conv rows h, layer-1 rows (outputs);  mode 2: g0 = grad of h, g1 = grad of the layer-1 rows (outputs)
    uint8_t *bits1, *bits2;            // sign bits of the layer-1 / layer-2 rows [N][H / 8]: outputs (mode 1) / inputs (mode 2)
    float slope;
};

// RB = relation slots held in registers (8 or 16).  MODE 0: the conv; 1: conv -> Linear-act-Linear-act (a layer's forward);
// 2: the layer's input gradient: mask, dgrad 2, mask, dgrad 1, then the conv on the transposed weights.
// (two workgroups per CU -- two waves per SIMD -- where the relation slots allow it: a batch of 512 graphs is then ONE round)
template <int RB, int MODE>
__global__ __launch_bounds__(kThreads, RB <= 8 ? 2 : 1) void conv_graphs_kernel(
    const bf16_t* __restrict__ X, const bf16_t* __restrict__ W, int32_t w_kn, const bf16_t* __restrict__ W_loop,
    const bf16_t* __restrict__ bias, int32_t R, const int32_t* __restrict__ node_ptr, const int32_t* __restrict__ edge_ptr,
    const int32_t* __restrict__ key_in, const int32_t* __restrict__ key_out, const int32_t* __restrict__ etype, int32_t G,
    bf16_t* __restrict__ out, const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ seg_nodes, bf16_t* __restrict__ aux,
    int32_t* __restrict__ err, LayerArgs la) {
    __shared__ GraphLds L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int n0 = 16 * wave;                                              // my output columns
    // Everything a workgroup's FIRST graph needs from memory is requested before anything is used -- its rows, its edges, the weight
    // slices -- so that the launch pays one round trip, not one per stage.
    constexpr int kXp = ((kMaxNodes + 1) * 8 + kThreads - 1) / kThreads;   // 16-byte pieces of the image per thread (3)
    constexpr int kEp = kMaxEdges / kThreads;                              // edges per thread (4)
    u32x4 xv[kXp];
    u32 xb[kXp];                                                           // mode 2: the 8 outer-mask bits of each piece
    u32 mb1 = 0;                                                           // mode 2: my word of the graph's inner-mask rows
    int ea[kEp], eb[kEp], er[kEp];
    int gi = blockIdx.x;
    int v0 = 0, n = 0, e0 = 0, m = 0;
    auto request = [&](int graph) __attribute__((always_inline)) {
        v0 = node_ptr[graph]; n = node_ptr[graph + 1] - v0; e0 = edge_ptr[graph]; m = edge_ptr[graph + 1] - e0;
        const bool fits = n >= 0 && n <= kMaxNodes && m >= 0 && m <= kMaxEdges;
#pragma unroll
        for (int k = 0; k < kXp; ++k) {
            const int i = tid + k * kThreads, row = i >> 3, piece = i & 7;
            const bool on = fits && row < n;
            xv[k] = on ? *reinterpret_cast<const u32x4*>(X + (size_t)(v0 + row) * kH + 8 * piece) : u32x4{0u, 0u, 0u, 0u};
            if constexpr (MODE == 2) xb[k] = on ? (u32)la.bits2[(size_t)(v0 + row) * (kH / 8) + piece] : 0u;
        }
        if constexpr (MODE == 2) mb1 = (fits && tid < 2 * n) ? reinterpret_cast<const u32*>(la.bits1 + (size_t)v0 * (kH / 8))[tid] : 0u;
#pragma unroll
        for (int k = 0; k < kEp; ++k) {
            const int i = tid + k * kThreads;
            const bool on = fits && i < m;
            ea[k] = on ? key_in[e0 + i] : 0; eb[k] = on ? key_out[e0 + i] : 0; er[k] = on ? etype[e0 + i] : 0;
        }
    };
#ifdef DN_CG_STATS
    long long stamp[8];
    stamp[0] = wall_clock64();
#endif
    if (gi < G) request(gi);
    // ---- my slice of every weight matrix as B fragments: lane (column n0 + j, group g) holds k = 32 ks + 8 g .. + 7.  Matrix m:
    // relation m (< RB), the self loop (RB), then the two Linears (layer modes).  A matrix whose memory is [k][n] (n contiguous) is
    // transposed through the wave's scratch; one that is [n][k] is read as it lies.  All loads are requested before the first use.
    constexpr int NM = RB + 1 + (MODE != 0 ? 2 : 0);
    bf16x8 wf[NM][2];
    {
        typedef dn_short4v __attribute__((address_space(3))) * lds_tr;
        const int r32 = lane >> 1, pc2 = lane & 1;
        const int q4 = j >> 2, p4 = j & 3;
        char* scratch = L.wscr[wave];
        constexpr int HALF = NM <= 9 ? NM : (NM <= 12 ? 6 : 10);           // matrices in flight at once (register budget: two workgroups per CU)
#pragma unroll
        for (int m0 = 0; m0 < NM; m0 += HALF) {
            u32x4 rawv[HALF][2];
#pragma unroll
            for (int q = 0; q < HALF; ++q) {
                const int mi = m0 + q;
                const bf16_t* w = mi < RB ? (mi < R ? W + (size_t)mi * kH * kH : nullptr) : mi == RB ? W_loop : mi == RB + 1 ? la.W1 : la.W2;
                // (mode 1: the conv reads [k][n] parameters, the Linears' [out][in] are [n][k]; mode 2: the other way round)
                const bool kn = mi <= RB ? w_kn != 0 : MODE == 2;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    rawv[q][ks] = (mi >= NM || w == nullptr) ? u32x4{0u, 0u, 0u, 0u}
                                  : kn ? *reinterpret_cast<const u32x4*>(w + (size_t)(32 * ks + r32) * kH + n0 + 8 * pc2)
                                       : *reinterpret_cast<const u32x4*>(w + (size_t)(n0 + j) * kH + 32 * ks + 8 * g);
            }
#pragma unroll
            for (int q = 0; q < HALF; ++q) {
                const int mi = m0 + q;
                if (mi >= NM) continue;
                const bool kn = mi <= RB ? w_kn != 0 : MODE == 2;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    if (!kn) { wf[mi][ks] = __builtin_bit_cast(bf16x8, rawv[q][ks]); continue; }
                    *reinterpret_cast<u32x4*>(scratch + r32 * 32 + 16 * pc2) = rawv[q][ks];   // (dn_load_w_kn16's transpose: 32 k x 16 n)
                    __builtin_amdgcn_wave_barrier();
                    const char* a0 = scratch + (8 * g + q4) * 32 + 8 * p4;
                    const dn_short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(a0));
                    const dn_short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(a0 + 4 * 32));
                    const dn_short8v f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    wf[mi][ks] = __builtin_bit_cast(bf16x8, f);
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    }
    const float bcol = bias != nullptr ? (float)bias[n0 + j] : 0.f;
    float b1col = 0.f, b2col = 0.f;
    if constexpr (MODE == 1) {
        if (la.b1 != nullptr) b1col = (float)la.b1[n0 + j];
        if (la.b2 != nullptr) b2col = (float)la.b2[n0 + j];
    }
#ifdef DN_CG_STATS
    stamp[1] = wall_clock64();
#endif
    // a [node][16 columns] tile in the accumulators (lane = column j, nodes 16 nb + 4 g + i) -> bf16 rows in an LDS image
    auto put_tile = [&](char* img, const f32x4 (&t)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int node = 16 * nb + 4 * g + i, col = n0 + j;
                if (node < n) *reinterpret_cast<bf16_t*>(img + sw_off(node, col >> 3) + 2 * (col & 7)) = (bf16_t)t[nb][i];
            }
    };
    // an image's rows 0 .. n -> global rows v0 .. (16-byte stores)
    auto store_rows = [&](const char* img, bf16_t* dst) __attribute__((always_inline)) {
        for (int i = tid; i < n * 8; i += kThreads) {
            const int row = i >> 3, piece = i & 7;
            *reinterpret_cast<u32x4*>(dst + (size_t)(v0 + row) * kH + 8 * piece) = *reinterpret_cast<const u32x4*>(img + sw_off(row, piece));
        }
    };
    // t[node][column] = sum_k img[node][k] * w[k][column] for my 16 columns
    auto dense = [&](const char* img, const bf16x8 (&w)[2], f32x4 (&t)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            t[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (16 * nb < n) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    t[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_frag(img, 16 * nb + j, 4 * ks + g), w[ks], t[nb], 0, 0, 0);
            }
        }
    };
    // sign bits of a tile's bf16 values -> bits[node][column / 8] (a lane group's 16 ballot bits = one node's 16 columns)
    auto put_bits = [&](uint8_t* bits, const f32x4 (&t)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned long long bal = __ballot((float)(bf16_t)t[nb][i] > 0.f);
                const int node = 16 * nb + 4 * g + i;
                if (j == 0 && node < n)
                    *reinterpret_cast<uint16_t*>(bits + (size_t)(v0 + node) * (kH / 8) + 2 * wave) = (uint16_t)(bal >> (16 * g));
            }
    };

#pragma unroll 1
    for (; gi < G; gi += gridDim.x, (gi < G ? request(gi) : (void)0)) {
        if (n > kMaxNodes || m > kMaxEdges || n < 0 || m < 0) {            // (the caller asked the index builder first: never expected)
            if (tid == 0) atomicOr(err, 1);
            continue;
        }
        // ---- the graph's rows -> LDS (rows n .. 64 zero), edges -> LDS, relation counts
#pragma unroll
        for (int k = 0; k < kXp; ++k) {
            const int i = tid + k * kThreads, row = i >> 3, piece = i & 7;
            if (i >= (kMaxNodes + 1) * 8) continue;
            if constexpr (MODE == 2) {                                     // the incoming gradient, masked by the outer activation, is the
                *reinterpret_cast<u32x4*>(L.x + sw_off(row, piece)) = u32x4{0u, 0u, 0u, 0u};        // MLP chain's input; the conv's image comes later
                if (row < kMaxNodes) {
                    const uint4 v = dn_keep_or_scale_bits(make_uint4(xv[k][0], xv[k][1], xv[k][2], xv[k][3]), xb[k], la.slope);
                    *reinterpret_cast<u32x4*>(L.o + sw_off(row, piece)) = u32x4{v.x, v.y, v.z, v.w};
                }
            } else {
                *reinterpret_cast<u32x4*>(L.x + sw_off(row, piece)) = xv[k];
            }
        }
        if constexpr (MODE == 2) { if (tid < 2 * kMaxNodes) L.mbits[tid] = mb1; }
        if (tid < 17) L.cnt[tid] = 0;
        __syncthreads();
#ifdef DN_CG_STATS
        stamp[2] = wall_clock64();
#endif
        bool bad = false;
#pragma unroll
        for (int k = 0; k < kEp; ++k) {
            const int i = tid + k * kThreads;
            if (i < m) {
                const int a = ea[k] - v0, b = eb[k] - v0, r = er[k];
                const bool o = a < 0 || a >= n || b < 0 || b >= n || r < 0 || r >= R || r >= RB;
                bad |= o;
                L.raw[i] = o ? 0xffffffffu : ((u32)r << 16) | ((u32)b << 8) | (u32)a;
                if (!o) atomicAdd(&L.cnt[r], 1);
            }
        }
        if (__syncthreads_or(bad ? 1 : 0)) {
            if (tid == 0) atomicOr(err, 1);
            continue;
        }
        int beg[RB + 1];                                                   // bucket starts: every bucket a multiple of 32 edges
        {
            int at = 0;
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                beg[r] = at;
                at += (L.cnt[r] + 31) & ~31;
            }
            beg[RB] = at;
        }
        // ---- stable bucketing: wave w places the relations r = w, w + 4, ...; null edges fill every bucket's tail
#pragma unroll
        for (int rr = 0; rr < RB / kWaves; ++rr) {
            const int r = wave + kWaves * rr;
            int at = 0, end = 0;
#pragma unroll
            for (int q = 0; q < RB; ++q)
                if (q == r) { at = beg[q]; end = beg[q + 1]; }
            for (int c = 0; c < m; c += 64) {
                const int i = c + lane;
                const u32 w = i < m ? L.raw[i] : 0xffffffffu;
                const bool mine = (w >> 16) == (u32)r;
                const unsigned long long bal = __ballot(mine);
                if (mine) {
                    const int p = at + __popcll(bal & ((1ull << lane) - 1ull));
                    L.e_in[p] = (uint8_t)(w & 0xffu);
                    L.e_out[p] = (uint8_t)((w >> 8) & 0xffu);
                }
                at += __popcll(bal);
            }
            for (int p = at + lane; p < end; p += 64) { L.e_in[p] = (uint8_t)kZeroRow; L.e_out[p] = 255; }
        }
        f32x4 acc[4];
        if constexpr (MODE == 2) {
            // ---- the MLP's input-gradient chain on the graph's rows: g1 = mask1(mask2(g) W2), g0 = g1 W1 -> the conv's image
            dense(L.o, wf[RB + 2], acc);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int node = 16 * nb + 4 * g + i, col = n0 + j;
                    const u32 word = L.mbits[2 * node + (col >> 5)];
                    if (!((word >> (col & 31)) & 1u)) acc[nb][i] *= la.slope;
                }
            put_tile(L.h, acc);
            __syncthreads();
            store_rows(L.h, la.R1);
            dense(L.h, wf[RB + 1], acc);
            put_tile(L.x, acc);
        }
        __syncthreads();
        if constexpr (MODE == 2) store_rows(L.x, la.R0);
#ifdef DN_CG_STATS
        stamp[3] = wall_clock64();
#endif

        // ---- out[node][n] for my 16 columns: lane (column j, group g) holds nodes 16 nb + 4 g .. + 3; the self loop first
        dense(L.x, wf[RB], acc);
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int b_end = beg[r + 1];
#pragma unroll 1
            for (int blk = beg[r]; blk < b_end; blk += 32) {
                // P[e][n] = x[in(e)] . W_r for the block's edges e = blk + j (first product) and blk + 16 + j (second)
                const int ra = L.e_in[blk + j], rb = L.e_in[blk + 16 + j];
                f32x4 p1 = {0.f, 0.f, 0.f, 0.f}, p2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    p1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_frag(L.x, ra, 4 * ks + g), wf[r][ks], p1, 0, 0, 0);
                    p2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_frag(L.x, rb, 4 * ks + g), wf[r][ks], p2, 0, 0, 0);
                }
                // lane (column j, group g) now holds P[blk + 4 g + i][j] (p1) and P[blk + 16 + 4 g + i][j] (p2): exactly the B
                // fragment of a product over 32 edge SLOTS, slot 8 g + t <-> edge 4 g + t (t < 4) / 16 + 4 g + t - 4
                const u32x4 pw = {pack2(p1[0], p1[1]), pack2(p1[2], p1[3]), pack2(p2[0], p2[1]), pack2(p2[2], p2[3])};
                const bf16x8 pb = __builtin_bit_cast(bf16x8, pw);
                const u32 d1 = *reinterpret_cast<const u32*>(&L.e_out[blk + 4 * g]);         // out nodes of my 8 slots
                const u32 d2 = *reinterpret_cast<const u32*>(&L.e_out[blk + 16 + 4 * g]);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    if (16 * nb >= n) continue;                            // (uniform)
                    const u32 me = (u32)(16 * nb + j);                     // A = selection: row = node me, slot t = [out(edge) == me]
                    u32x4 sw;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        sw[q] = (((d1 >> (16 * q)) & 0xffu) == me ? 0x3f80u : 0u) | (((d1 >> (16 * q + 8)) & 0xffu) == me ? 0x3f800000u : 0u);
                        sw[2 + q] = (((d2 >> (16 * q)) & 0xffu) == me ? 0x3f80u : 0u) | (((d2 >> (16 * q + 8)) & 0xffu) == me ? 0x3f800000u : 0u);
                    }
                    acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, sw), pb, acc[nb], 0, 0, 0);
                }
            }
        }
#ifdef DN_CG_STATS
        stamp[4] = wall_clock64();
#endif
        // ---- finished rows -> LDS -> 16-byte stores; the per-graph column sum of the segment's INPUT rows -> aux
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[nb][i] += bcol;
        put_tile(L.o, acc);
        if (aux != nullptr) {                                              // segment gi = nodes seg_nodes[seg_ptr[gi] ..): a contiguous run
            const int s0 = seg_ptr[gi], sc = seg_ptr[gi + 1] - s0;
            const int first = sc > 0 ? seg_nodes[s0] - v0 : 0;
            const int col = tid & 63, part = tid >> 6;
            float sum = 0.f;
            for (int k = part; k < sc; k += 4) {
                const int row = first + k;
                if (row >= 0 && row < n) sum += (float)*reinterpret_cast<const bf16_t*>(L.x + sw_off(row, col >> 3) + 2 * (col & 7));
            }
            L.colsum[part][col] = sum;
        }
        __syncthreads();
        if constexpr (MODE == 1) {
            // ---- the layer's MLP on the rows just finished: Linear - act - Linear - act, sign bits of both for the backward
            store_rows(L.o, la.R0);
            f32x4 t[4];
            dense(L.o, wf[RB + 1], t);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int i = 0; i < 4; ++i) t[nb][i] = act(t[nb][i] + b1col, la.slope);
            put_tile(L.h, t);
            put_bits(la.bits1, t);
            __syncthreads();
            store_rows(L.h, la.R1);
            dense(L.h, wf[RB + 2], t);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int i = 0; i < 4; ++i) t[nb][i] = act(t[nb][i] + b2col, la.slope);
            put_bits(la.bits2, t);
            put_tile(L.o, t);                                              // (every wave has read L.o for the first Linear: the barrier above)
            __syncthreads();
        }
        store_rows(L.o, out);
        if (aux != nullptr && tid < kH)
            aux[(size_t)gi * kH + tid] = (bf16_t)(((L.colsum[0][tid] + L.colsum[1][tid]) + L.colsum[2][tid]) + L.colsum[3][tid]);
        __syncthreads();                                                   // the next graph rewrites the images
#ifdef DN_CG_STATS
        stamp[5] = wall_clock64();
        if (blockIdx.x == 0 && tid == 0)
            for (int k = 0; k < 6; ++k) reinterpret_cast<long long*>(err + 2)[k] = stamp[k];
#endif
    }
}

}  // namespace

extern "C" {

int32_t dn_conv_graphs_max_nodes(void) { return kMaxNodes; }
int32_t dn_conv_graphs_max_edges(void) { return kMaxEdges; }

static int conv_graphs_launch(int mode, const void* X, int32_t H, const void* W, int32_t w_kn, const void* W_loop, const void* bias,
                              int32_t num_rels, const int32_t* node_ptr, const int32_t* edge_ptr, const int32_t* key_in,
                              const int32_t* key_out, const int32_t* etype, int64_t num_graphs, int64_t N, void* out, const int32_t* seg_ptr,
                              const int32_t* seg_nodes, void* aux, int32_t* dev_err, const LayerArgs& la, dn_stream_t stream) {
    DN_REQUIRE(H == kH, "dn_conv_graphs: unsupported width %d (64 only; the row-factorised launches serve the others)", H);
    DN_REQUIRE(num_rels >= 1 && num_rels <= 16, "dn_conv_graphs: 1 .. 16 relations");
    DN_REQUIRE(num_graphs >= 0 && num_graphs < INT32_MAX && N >= 0 && N < INT32_MAX, "dn_conv_graphs: bad sizes");
    if (num_graphs == 0 || N == 0) return DN_OK;
    DN_REQUIRE(X && W && node_ptr && edge_ptr && key_in && key_out && etype && out && dev_err, "dn_conv_graphs: NULL pointer");
    DN_REQUIRE((aux == nullptr) || (seg_ptr && seg_nodes), "dn_conv_graphs: aux needs the segments");
    DN_REQUIRE((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(W_loop) |
                reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(la.W1) | reinterpret_cast<uintptr_t>(la.W2) |
                reinterpret_cast<uintptr_t>(la.R0) | reinterpret_cast<uintptr_t>(la.R1)) % 16 == 0, "dn_conv_graphs: unaligned pointer");
    DN_REQUIRE((reinterpret_cast<uintptr_t>(la.bits1) | reinterpret_cast<uintptr_t>(la.bits2)) % 4 == 0, "dn_conv_graphs: unaligned bit tensors");
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)(num_graphs < 1024 ? num_graphs : 1024);
#define DN_CG_LAUNCH(RB, M)                                                                                                         \
    hipLaunchKernelGGL((conv_graphs_kernel<RB, M>), dim3(grid), dim3(kThreads), 0, st, (const bf16_t*)X, (const bf16_t*)W, w_kn,   \
                       (const bf16_t*)W_loop, (const bf16_t*)bias, num_rels, node_ptr, edge_ptr, key_in, key_out, etype,            \
                       (int32_t)num_graphs, (bf16_t*)out, seg_ptr, seg_nodes, (bf16_t*)aux, dev_err, la)
    if (num_rels <= 8) {
        if (mode == 0) DN_CG_LAUNCH(8, 0);
        else if (mode == 1) DN_CG_LAUNCH(8, 1);
        else DN_CG_LAUNCH(8, 2);
    } else {
        if (mode == 0) DN_CG_LAUNCH(16, 0);
        else if (mode == 1) DN_CG_LAUNCH(16, 1);
        else DN_CG_LAUNCH(16, 2);
    }
#undef DN_CG_LAUNCH
    DN_CHECK_LAUNCH();
    return DN_OK;
}

int dn_conv_graphs_bf16(const void* X, int32_t H, const void* W, int32_t w_kn, const void* W_loop, const void* bias, int32_t num_rels,
                        const int32_t* node_ptr, const int32_t* edge_ptr, const int32_t* key_in, const int32_t* key_out,
                        const int32_t* etype, int64_t num_graphs, int64_t N, void* out, const int32_t* seg_ptr, const int32_t* seg_nodes,
                        void* aux, int32_t* dev_err, dn_stream_t stream) {
    return conv_graphs_launch(0, X, H, W, w_kn, W_loop, bias, num_rels, node_ptr, edge_ptr, key_in, key_out, etype, num_graphs, N, out,
                              seg_ptr, seg_nodes, aux, dev_err, LayerArgs{}, stream);
}

int dn_layer_graphs_fwd_bf16(const void* X, int32_t H, const void* W, const void* W_loop, const void* bias, int32_t num_rels,
                             const void* W1, const void* b1, const void* W2, const void* b2, float act_slope, const int32_t* node_ptr,
                             const int32_t* edge_ptr, const int32_t* src, const int32_t* dst, const int32_t* etype, int64_t num_graphs,
                             int64_t N, void* conv_out, void* mid, void* out, void* bits1, void* bits2, const int32_t* seg_ptr,
                             const int32_t* seg_nodes, void* aux, int32_t* dev_err, dn_stream_t stream) {
    DN_REQUIRE(num_graphs == 0 || N == 0 || (W1 && W2 && conv_out && mid && bits1 && bits2), "dn_layer_graphs_fwd: NULL pointer");
    const LayerArgs la{(const bf16_t*)W1, (const bf16_t*)b1, (const bf16_t*)W2, (const bf16_t*)b2, (bf16_t*)conv_out, (bf16_t*)mid,
                       (uint8_t*)bits1, (uint8_t*)bits2, act_slope};
    return conv_graphs_launch(1, X, H, W, 1, W_loop, bias, num_rels, node_ptr, edge_ptr, src, dst, etype, num_graphs, N, out, seg_ptr,
                              seg_nodes, aux, dev_err, la, stream);
}

int dn_layer_graphs_bwd_bf16(const void* G, int32_t H, const void* W, const void* W_loop, int32_t num_rels, const void* W1,
                             const void* W2, float act_slope, const void* bits1, const void* bits2, const int32_t* node_ptr,
                             const int32_t* edge_ptr, const int32_t* src, const int32_t* dst, const int32_t* etype, int64_t num_graphs,
                             int64_t N, void* g_mid, void* g_conv, void* g_in, const int32_t* seg_ptr, const int32_t* seg_nodes,
                             void* aux, int32_t* dev_err, dn_stream_t stream) {
    DN_REQUIRE(num_graphs == 0 || N == 0 || (W1 && W2 && g_mid && g_conv && bits1 && bits2), "dn_layer_graphs_bwd: NULL pointer");
    const LayerArgs la{(const bf16_t*)W1, nullptr, (const bf16_t*)W2, nullptr, (bf16_t*)g_conv, (bf16_t*)g_mid, (uint8_t*)bits1,
                       (uint8_t*)bits2, act_slope};
    // (the input gradient of the conv: edges taken from destination to source, the relation weights read as they lie = transposed)
    return conv_graphs_launch(2, G, H, W, 0, W_loop, nullptr, num_rels, node_ptr, edge_ptr, dst, src, etype, num_graphs, N, g_in, seg_ptr,
                              seg_nodes, aux, dev_err, la, stream);
}

}  // extern "C"
