// Graph-local neighbour sum on the matrix cores (gfx950):
//   out[v, :] = self_coef * x[v, :] + sum_{i in [ptr[v], ptr[v+1])} x[idx[i], :]        for the rows of the listed tiles
// -- the aggregation of GINConv (graph_classification/graph_neural_networks/models/gconv.py:197, PyG `propagate(aggr="add")` +
// `(1 + eps) * x_i`) on a batch of small graphs, where a tile is a run of WHOLE graphs (<= 64 rows), so every neighbour row of
// a tile's rows lies inside the tile.
//
// The plain kernel (dn_gather_segsum_*) walks ptr -> idx -> rows per destination: three dependent round trips per segment, the
// gathered rows re-read through L2 (34 % of the HBM peak on compulsory bytes).  Here a tile's rows arrive
// ONCE, coalesced; they are cut into three bf16 planes hi / mid / lo in LDS (hi + mid + lo reproduces the fp32 value to its last
// bit or so), the tile's adjacency COUNTS (how often row s feeds row d; small integers, exact in bf16) are scattered into a
// 64 x 64 bf16 matrix in LDS, and the sum is the dense product  Adj x (hi + mid + lo)  on v_mfma_f32_16x16x32_bf16 with fp32
// accumulation: exact products, no dependent loads, HBM sees rows in, rows out and the index once.
#include "dn_common.h"
#include "../../include/dn_hip.h"

namespace {

typedef __bf16 bf16_t;
typedef short short4v __attribute__((ext_vector_type(4)));
typedef short short8v __attribute__((ext_vector_type(8)));
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kGtRows = 64;        // rows of a tile
constexpr int gt_threads(int H) { return H >= 128 ? 512 : 256; }   // 8 waves measured best at H = 128 (4: -10 %, 16: -12 %)
constexpr int kGtPad = 8;          // bf16 elements of row padding (planes and adjacency)

// element j of lane l = tile[8 * (l >> 4) + j][col0 + (l & 15)]   (hardware transpose read, as in dn_rel.hip)
__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* tile, int stride, int col0, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const bf16_t* a0 = tile + (8 * g + q) * stride + col0 + 4 * p;
    typedef short4v __attribute__((address_space(3))) * lds_p;
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a0));
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a0 + 4 * stride));
    const short8v f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, f);
}

// One persistent workgroup walks tiles blockIdx.x, + gridDim.x, ...; what the NEXT tile needs from memory (its rows, its list
// bounds, its index entries) is requested right after this tile's rows have been handed to LDS and lands under this tile's
// adjacency build, MFMAs and stores: the memory latency of a tile is off its critical path, which is what a two-workgroup-per-CU
// kernel with five barriers per tile needs.  Measured on the PROTEINS-shaped GIN batch of bench.py (H = 128): 0.44 GB of
// compulsory traffic in 100-110 us per direction for the 64 % of the rows that sit in graphs of <= 64 rows = 4.0-4.3 TB/s,
// against 2.9-3.0 TB/s for the plain kernel; phase ablation: rows in / planes / barriers / rows staged alone 70 us, + stores
// 27 us, + adjacency 18 us, + MFMAs 20 us (the phases of a workgroup do not overlap; a second workgroup per CU is what does).
template <int H>
__global__ __launch_bounds__(gt_threads(H)) void graph_tile_sum_kernel(const float* __restrict__ x, const int32_t* __restrict__ ptr,
                                                                    const int32_t* __restrict__ idx,
                                                                    const int32_t* __restrict__ seg,
                                                                    const int32_t* __restrict__ tiles, int32_t num_tiles,
                                                                    int32_t num_rows, int32_t num_entries, float self_coef,
                                                                    float* __restrict__ out, int32_t* __restrict__ bad) {
    constexpr int kGtThreads = gt_threads(H);
    constexpr int SP = H + kGtPad;                  // plane row stride (bf16)
    constexpr int SA = kGtRows + kGtPad;            // adjacency row stride (bf16)
    constexpr int SO = H + 4;                       // output staging row stride (fp32)
    constexpr int LPR = H / 4;                      // float4 pieces per row
    constexpr int PP = kGtRows * LPR / kGtThreads;  // pieces per thread
    constexpr int NW = kGtThreads / 64, NCT = H / 16;
    constexpr int MS = NW > NCT ? NW / NCT : 1;     // waves that share a column tile split the row tiles
    constexpr int NT = NW >= NCT ? 1 : NCT / NW;    // 16-column tiles per wave
    constexpr int MTW = (kGtRows / 16) / MS;        // 16-row tiles per wave
    constexpr int EP = 1024 / kGtThreads;           // index entries per thread kept in registers (64 rows x 16 entries)
    static_assert(kGtRows * LPR % kGtThreads == 0 && H % 64 == 0 && NW % (NCT / NT) == 0 && MTW >= 1, "unsupported width");
    static_assert(kGtRows * SO * 4 <= 3 * kGtRows * SP * 2, "the output staging tile lives in the planes");
    __shared__ __attribute__((aligned(16))) bf16_t planes[3 * kGtRows * SP];
    __shared__ __attribute__((aligned(16))) uint32_t adjW[kGtRows * SA / 2];
    __shared__ int32_t ptrL[kGtRows + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    struct Tile {                                   // one tile's memory image in registers
        int row_beg, rows, e_beg, e_end;
        float4 xr[PP];
        int32_t ent[EP], sg[EP];                    // index entries and (when the caller has them) their destination rows
        int32_t pv;                                 // ptr[row_beg + tid] (tid <= rows)
    };
    // Every load below is UNCONDITIONAL (row / entry numbers clamped into the tile, values masked afterwards): written as
    // `cond ? load : 0` hipcc put each load in its own branch and a vmcnt(0) behind it -- eight serial round trips per tile.
    // tile record: {first row, end row, first index entry, end entry} -- read TWO tiles ahead (uniform address), so that the
    // loads of the next tile start without a dependent chain tiles -> ptr -> idx in front of them
    auto record = [&](int t) -> int4 {
        int4 m = make_int4(0, 0, 0, 0);
        if (t < num_tiles) m = *reinterpret_cast<const int4*>(tiles + 4 * (size_t)t);
        return m;
    };
    auto request = [&](const int4& m, Tile& T) {
        T.row_beg = m.x; T.rows = m.y - m.x; T.e_beg = m.z; T.e_end = m.w;
        if (T.rows < 0 || T.rows > kGtRows || T.row_beg < 0 || (int64_t)T.row_beg + T.rows > num_rows || T.e_beg < 0 ||
            T.e_end < T.e_beg || T.e_end > num_entries) {                  // (never dereferenced)
            if (tid == 0) atomicOr(bad, 1);
            T.row_beg = 0; T.rows = 0; T.e_beg = T.e_end = 0;
        }
        const int last = T.rows > 0 ? T.rows - 1 : 0;                      // (num_rows >= 1: row 0 exists)
#pragma unroll
        for (int k = 0; k < PP; ++k) {
            const int p = tid + k * kGtThreads, r = min(p / LPR, last), c4 = p % LPR;
            T.xr[k] = *reinterpret_cast<const float4*>(x + (size_t)(T.row_beg + r) * H + c4 * 4);
        }
        const int e_last = T.e_end > T.e_beg ? T.e_end - 1 : 0;
#pragma unroll
        for (int k = 0; k < EP; ++k) {
            const int e = min(T.e_beg + tid + k * kGtThreads, e_last);
            T.ent[k] = num_entries > 0 ? idx[e] : 0;
            T.sg[k] = (seg != nullptr && num_entries > 0) ? seg[e] : 0;
        }
        T.pv = ptr[T.row_beg + min(tid, T.rows)];
    };

    Tile cur, nxt;
    request(record((int)blockIdx.x), cur);
    int4 rec_next = record((int)blockIdx.x + (int)gridDim.x);
    for (int t = (int)blockIdx.x; t < num_tiles; t += (int)gridDim.x) {
        const int row_beg = cur.row_beg, rows = cur.rows, e_beg = cur.e_beg, e_end = cur.e_end;
        // (1) this tile's rows -> three bf16 planes; list bounds -> LDS; adjacency cleared
        if (tid <= rows) ptrL[tid] = cur.pv;
        for (int i = tid; i < kGtRows * SA / 2; i += kGtThreads) adjW[i] = 0;
#pragma unroll
        for (int k = 0; k < PP; ++k) {
            const int p = tid + k * kGtThreads, r = p / LPR, c4 = p % LPR;
            // hi = the top 16 bits of x (truncated, exactly a bf16), mid = the top 16 bits of x - hi (exact in fp32), lo = x - hi - mid
            // rounded to bf16: hi + mid + lo == x up to the last rounding.  Truncation keeps the split to an AND and a subtract per
            // plane (the kernel is issue-bound: SQ_ACTIVE_INST fills 83 % of the SIMD cycles).
            const uint32_t keep = r < rows ? 0xffff0000u : 0u;             // (clamped loads: rows past the tile are zero planes)
            const float v[4] = {cur.xr[k].x, cur.xr[k].y, cur.xr[k].z, cur.xr[k].w};
            uint32_t hb[4], mb[4];
            bf16x4 lo;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                hb[i] = __float_as_uint(v[i]) & keep;
                const float r1 = (keep ? v[i] : 0.f) - __uint_as_float(hb[i]);
                mb[i] = __float_as_uint(r1) & 0xffff0000u;
                lo[i] = (bf16_t)(r1 - __uint_as_float(mb[i]));
            }
            const uint2 hi = make_uint2((hb[0] >> 16) | hb[1], (hb[2] >> 16) | hb[3]);
            const uint2 mid = make_uint2((mb[0] >> 16) | mb[1], (mb[2] >> 16) | mb[3]);
            *reinterpret_cast<uint2*>(planes + (0 * kGtRows + r) * SP + c4 * 4) = hi;
            *reinterpret_cast<uint2*>(planes + (1 * kGtRows + r) * SP + c4 * 4) = mid;
            *reinterpret_cast<bf16x4*>(planes + (2 * kGtRows + r) * SP + c4 * 4) = lo;
        }
        request(rec_next, nxt);                                            // in flight until the top of the next iteration
        rec_next = record(t + 2 * (int)gridDim.x);
        __syncthreads();
        // (2) adjacency counts: entry e of the tile's index range feeds row dst(e) (found in the tile's bounds) from row idx[e]
        auto add_entry = [&](int e, int src_row, int dst_row) {
            const int s = src_row - row_beg;
            int d = dst_row - row_beg;
            if (seg == nullptr) {                                          // destination of entry e: found in the tile's bounds
                int lo = 0, hi = rows;
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (ptrL[mid] <= e) lo = mid; else hi = mid;
                }
                d = lo;
            }
            if (s < 0 || s >= rows || d < 0 || d >= rows) { atomicOr(bad, 1); return; }   // not a run of whole graphs
            atomicAdd(&adjW[(d * SA + s) >> 1], (s & 1) ? 0x10000u : 1u);
        };
#pragma unroll
        for (int k = 0; k < EP; ++k) {
            const int e = e_beg + tid + k * kGtThreads;
            if (e < e_end) add_entry(e, cur.ent[k], cur.sg[k]);
        }
        for (int e = e_beg + tid + EP * kGtThreads; e < e_end; e += kGtThreads)                 // (denser tiles)
            add_entry(e, idx[e], seg != nullptr ? seg[e] : 0);
        __syncthreads();
        for (int i = tid; i < kGtRows * SA / 2; i += kGtThreads) {          // counts -> bf16 (exact up to 256)
            const uint32_t w = adjW[i];
            // more than 256 parallel edges between one pair of nodes would be rounded by the conversion: raise the flag, the
            // caller's plan then falls back to the plain gather (ops._tile_neighbor_sum reads it once per batch)
            if ((w & 0xffffu) > 256u || (w >> 16) > 256u) atomicOr(bad, 1);
            adjW[i] = (__float_as_uint((float)(w & 0xffffu)) >> 16) | (__float_as_uint((float)(w >> 16)) & 0xffff0000u);
        }
        __syncthreads();
        // (3) out tile (transposed) = planes^T x Adj^T:  D[col][dst] = sum_src X[src][col] * Adj[dst][src].  A operand = the plane,
        //     K-strided (transpose read); B operand = Adj row-major (element j of lane l = Adj[dst = l & 15][src = 8 (l >> 4) + j]).
        //     A lane then holds 4 CONSECUTIVE columns of one destination row: the tile is staged with 16-byte LDS writes.  Per
        //     32-row K-step every fragment is requested before the first MFMA (Adj fragments serve all three planes).
        const bf16_t* adj = reinterpret_cast<const bf16_t*>(adjW);
        const int MT = (rows + 15) >> 4, KS = (rows + 31) >> 5;
        const int wc = wave % (NCT / NT), m0 = (wave / (NCT / NT)) * MTW;  // my column tiles wc * NT + n, my row tiles m0 + m
        f32x4 acc[MTW][NT];
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 af[MTW], xf[3][NT];
#pragma unroll
            for (int m = 0; m < MTW; ++m)
                af[m] = *reinterpret_cast<const bf16x8*>(adj + ((m0 + m) * 16 + (lane & 15)) * SA + ks * 32 + 8 * (lane >> 4));
#pragma unroll
            for (int tt = 0; tt < 3; ++tt)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    xf[tt][n] = tr_frag(planes + (tt * kGtRows + ks * 32) * SP, SP, (wc * NT + n) * 16, lane);
#pragma unroll
            for (int tt = 0; tt < 3; ++tt)
#pragma unroll
                for (int m = 0; m < MTW; ++m) {
                    if (m0 + m >= MT) break;
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[tt][n], af[m], acc[m][n], 0, 0, 0);
                }
        }
        __syncthreads();                                                   // the planes are free: stage the tile for whole-row stores
        float* outL = reinterpret_cast<float*>(planes);
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            if (m0 + m >= MT) break;
#pragma unroll
            for (int n = 0; n < NT; ++n)                                   // lane holds D[col = 16(..) + 4(l>>4) + i][dst = 16(m0+m) + (l&15)]
                *reinterpret_cast<f32x4*>(outL + ((m0 + m) * 16 + (lane & 15)) * SO + (wc * NT + n) * 16 + 4 * (lane >> 4)) = acc[m][n];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PP; ++k) {
            const int p = tid + k * kGtThreads, r = p / LPR, c4 = p % LPR;
            if (r < rows) {
                const float4 d = *reinterpret_cast<const float4*>(outL + r * SO + c4 * 4);
                float4 o;
                o.x = fmaf(self_coef, cur.xr[k].x, d.x); o.y = fmaf(self_coef, cur.xr[k].y, d.y);
                o.z = fmaf(self_coef, cur.xr[k].z, d.z); o.w = fmaf(self_coef, cur.xr[k].w, d.w);
                typedef float f32x4v __attribute__((ext_vector_type(4)));   // streaming store (the tile's rows are done: -6 % launch time)
                __builtin_nontemporal_store(f32x4v{o.x, o.y, o.z, o.w}, reinterpret_cast<f32x4v*>(out + (size_t)(row_beg + r) * H + c4 * 4));
            }
        }
        __syncthreads();                                                   // outL (the planes) is rewritten by the next tile
        cur = nxt;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same sum for a LIST of rows (the rows of graphs too large for a tile), straight from the full CSR -- no compacted copy of
// the lists or of x: out[s] = self_coef * x[s] + sum_{i in [ptr[s], ptr[s+1])} x[idx[i]]  for s in rows[0 .. n).
//   COOP = false: one lane group (H / 4 lanes, 16 bytes per lane) per listed row, 8 row loads in flight.
//   COOP = true : one workgroup per listed row (the hubs: a dummy node of a 600-node graph has 600 entries): its lane groups
//                 take the entries round-robin, the partial rows are added in group order through LDS (fixed order: deterministic).
// ---------------------------------------------------------------------------------------------------------------------
template <int H, bool COOP, bool RECORDS>
__global__ __launch_bounds__(256) void gather_rows_sum_kernel(const float* __restrict__ x, const int32_t* __restrict__ ptr,
                                                              const int32_t* __restrict__ idx, const int32_t* __restrict__ rows,
                                                              int32_t n, float self_coef, float* __restrict__ out) {
    constexpr int LPR = H / 4, GPB = 256 / LPR, KU = 8;
    const int lane = threadIdx.x % LPR, group = threadIdx.x / LPR;
    __shared__ __attribute__((aligned(16))) float part[COOP ? GPB * H : 4];
    // The listed rows are graph-major: a workgroup takes a CONTIGUOUS chunk of the list and the workgroups of one XCD take
    // neighbouring chunks (dn_xcd_chunk), so the rows of one large graph -- each other's neighbours -- are gathered through ONE
    // L2 (dealt round-robin over the chip every XCD fetched them for itself: 3.8x the compulsory reads).
    const int64_t chunk = dn_xcd_chunk(blockIdx.x, gridDim.x);
    const int per = (int)(((int64_t)n + gridDim.x - 1) / gridDim.x);
    const int c_beg = (int)min((int64_t)n, chunk * per), c_end = (int)min((int64_t)n, (chunk + 1) * per);
    const int first = COOP ? c_beg : c_beg + group, step = COOP ? 1 : GPB;
    for (int i = first; i < c_end; i += step) {
        int s, beg, end;
        if (RECORDS) {                                                     // {row, first entry, end entry, -}: one load, no ptr round trip
            const int4 rc = *reinterpret_cast<const int4*>(rows + 4 * (size_t)i);
            s = rc.x; beg = rc.y; end = rc.z;
        } else {
            s = rows[i]; beg = ptr[s]; end = ptr[s + 1];
        }
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const int e0 = COOP ? beg + group : beg, de = COOP ? GPB : 1;
        for (int e = e0; e < end; e += KU * de) {
            float4 v[KU];
#pragma unroll
            for (int k = 0; k < KU; ++k) {
                const int ee = e + k * de;
                const int r = idx[min(ee, end - 1)];                       // (clamped: the load is unconditional, masked below)
                v[k] = *reinterpret_cast<const float4*>(x + (size_t)r * H + lane * 4);
                if (ee >= end) v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int k = 0; k < KU; ++k) { acc[0] += v[k].x; acc[1] += v[k].y; acc[2] += v[k].z; acc[3] += v[k].w; }
        }
        if (COOP) {
            *reinterpret_cast<float4*>(part + group * H + lane * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            __syncthreads();
            if (group == 0) {
#pragma unroll
                for (int g = 1; g < GPB; ++g) {
                    const float4 p = *reinterpret_cast<const float4*>(part + g * H + lane * 4);
                    acc[0] += p.x; acc[1] += p.y; acc[2] += p.z; acc[3] += p.w;
                }
            }
            __syncthreads();
            if (group != 0) continue;
        }
        const float4 xs = *reinterpret_cast<const float4*>(x + (size_t)s * H + lane * 4);
        typedef float f32x4v __attribute__((ext_vector_type(4)));           // streaming store: the row is re-read by a later launch only
        __builtin_nontemporal_store(f32x4v{fmaf(self_coef, xs.x, acc[0]), fmaf(self_coef, xs.y, acc[1]), fmaf(self_coef, xs.z, acc[2]),
                                           fmaf(self_coef, xs.w, acc[3])},
                                    reinterpret_cast<f32x4v*>(out + (size_t)s * H + lane * 4));
    }
}

template <int H>
int launch_rows_sum(const float* x, const int32_t* ptr, const int32_t* idx, const int32_t* rows, int64_t n, int32_t coop, int32_t records,
                    float self_coef, float* out, hipStream_t st) {
    constexpr int GPB = 256 / (H / 4);
    // chunks of ~4 rows per lane group (one row per workgroup for the hubs), a multiple of the XCD count of workgroups
    int64_t blocks = coop ? n : dn_cdiv(n, (int64_t)GPB * 4);
    blocks = dn_cdiv(blocks < 16384 ? blocks : 16384, DN_NUM_XCD) * DN_NUM_XCD;
    const unsigned grid = (unsigned)blocks;
#define DN_GO(C, R) hipLaunchKernelGGL((gather_rows_sum_kernel<H, C, R>), dim3(grid), dim3(256), 0, st, x, ptr, idx, rows, (int32_t)n, self_coef, out)
    if (coop) { if (records) DN_GO(true, true); else DN_GO(true, false); }
    else { if (records) DN_GO(false, true); else DN_GO(false, false); }
#undef DN_GO
    DN_CHECK_LAUNCH();
    return DN_OK;
}

}  // namespace

extern "C" {

int dn_graph_tile_sum_f32(const float* x, int64_t num_rows, int32_t H, const int32_t* ptr, const int32_t* idx, const int32_t* seg,
                          int64_t num_entries, const int32_t* tiles, int64_t num_tiles, float self_coef, float* out, int32_t* bad,
                          dn_stream_t stream) {
    DN_REQUIRE(H == 64 || H == 128 || H == 256, "dn_graph_tile_sum: H must be 64, 128 or 256");
    DN_REQUIRE(num_tiles >= 0 && num_tiles < 0x7fffffffLL && num_rows >= 0 && num_rows < 0x7fffffffLL && num_entries >= 0 &&
               num_entries < 0x7fffffffLL, "dn_graph_tile_sum: bad sizes");
    if (num_tiles == 0 || num_rows == 0) return DN_OK;
    DN_REQUIRE(x && ptr && (idx || num_entries == 0) && tiles && out && bad, "dn_graph_tile_sum: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    // persistent workgroups: two per CU at H <= 128 (61 KB of LDS each), one at H = 256
    const int64_t slots = 256 * (H == 256 ? 1 : 2);
    const unsigned grid = (unsigned)(num_tiles < slots ? num_tiles : slots);
    if (H == 64) hipLaunchKernelGGL((graph_tile_sum_kernel<64>), dim3(grid), dim3(gt_threads(64)), 0, st, x, ptr, idx, seg, tiles, (int32_t)num_tiles, (int32_t)num_rows, (int32_t)num_entries, self_coef, out, bad);
    else if (H == 128) hipLaunchKernelGGL((graph_tile_sum_kernel<128>), dim3(grid), dim3(gt_threads(128)), 0, st, x, ptr, idx, seg, tiles, (int32_t)num_tiles, (int32_t)num_rows, (int32_t)num_entries, self_coef, out, bad);
    else hipLaunchKernelGGL((graph_tile_sum_kernel<256>), dim3(grid), dim3(gt_threads(256)), 0, st, x, ptr, idx, seg, tiles, (int32_t)num_tiles, (int32_t)num_rows, (int32_t)num_entries, self_coef, out, bad);
    DN_CHECK_LAUNCH();
    return DN_OK;
}

int dn_gather_rows_sum_f32(const float* x, int32_t H, const int32_t* ptr, const int32_t* idx, const int32_t* rows, int32_t rows_are_records,
                           int64_t num_listed, int32_t workgroup_per_row, float self_coef, float* out, dn_stream_t stream) {
    DN_REQUIRE(H == 64 || H == 128 || H == 256, "dn_gather_rows_sum: H must be 64, 128 or 256");
    DN_REQUIRE(num_listed >= 0 && num_listed < 0x7fffffffLL, "dn_gather_rows_sum: bad row count");
    if (num_listed == 0) return DN_OK;
    DN_REQUIRE(x && (ptr || rows_are_records) && idx && rows && out, "dn_gather_rows_sum: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    if (H == 64) return launch_rows_sum<64>(x, ptr, idx, rows, num_listed, workgroup_per_row, rows_are_records, self_coef, out, st);
    if (H == 128) return launch_rows_sum<128>(x, ptr, idx, rows, num_listed, workgroup_per_row, rows_are_records, self_coef, out, st);
    return launch_rows_sum<256>(x, ptr, idx, rows, num_listed, workgroup_per_row, rows_are_records, self_coef, out, st);
}

/* host-side packing (no GPU): greedy runs of WHOLE graphs with at most max_rows rows each; larger graphs are left out */
int dn_graph_tiles_host(const int32_t* node_ptr, int64_t G, int32_t max_rows, int32_t* tiles, int64_t cap, int64_t* num_tiles) {
    DN_REQUIRE(G >= 0 && max_rows >= 1 && num_tiles, "dn_graph_tiles_host: bad arguments");
    DN_REQUIRE(G == 0 || (node_ptr && tiles), "dn_graph_tiles_host: NULL pointer");
    int64_t T = 0;
    int32_t beg = -1, end = -1;
    auto close = [&]() -> bool {
        if (beg >= 0 && end > beg) {
            if (T >= cap) return false;
            tiles[2 * T] = beg; tiles[2 * T + 1] = end; ++T;
        }
        beg = -1;
        return true;
    };
    for (int64_t g = 0; g < G; ++g) {
        const int32_t a = node_ptr[g], b = node_ptr[g + 1];
        DN_REQUIRE(b >= a, "dn_graph_tiles_host: node_ptr must be non-decreasing");
        if (b == a) continue;
        if (b - a > max_rows) { if (!close()) break; continue; }
        if (beg >= 0 && b - beg > max_rows && !close()) break;
        if (beg < 0) beg = a;
        end = b;
    }
    if (!close()) { dn_set_error("dn_graph_tiles_host: tile table too small"); return DN_ERR_WORKSPACE; }
    *num_tiles = T;
    return DN_OK;
}

}  // extern "C"
