// dn_rows_pipe_bf16: one direction of the row-factorised relation-wise message pass
//     out[v] = bias + sum_{rows p into v} ( in_row(p) W[rel p]^T )          (the self loop is relation R: one row per node)
// (rgin.py:102-120,137-145: per-edge transform, fn.sum reduce, self loop, bias) as ONE persistent launch in which the
// transformed rows never travel through HBM.
//
// Why.  The two-launch form (dn_rows_transform_bf16 writes the transformed rows Y, dn_rows_selfsum_bf16 gathers them per
// node) moves every Y row to HBM and back: 2.1 of the 4.8 GB per direction at the benchmark size, and both launches sit at
// the rate one CU can pull from HBM (~21 GB/s per CU); data served by the XCD's L2 comes 3x faster
// (MI355X_MICROARCH.md, "Indexed rows").  The products cannot stay in one CU either: MFMA tiles need rows grouped by
// relation (weight-stationary registers), the sum needs them grouped by destination.
//
// How.  The batch is cut into 8 groups of graphs (one per XCD) and every group into small BATCHES of consecutive graphs
// whose transformed rows (a few hundred KB) fit the XCD's 4 MiB L2.  The workgroups of an XCD take fixed roles:
//   T (transform) roles own one relation's weights in registers (the self loop is one more relation) and walk the batches:
//     gather the rows of (batch, relation), MFMA, write the products into a small RING of batch slots (plain stores: the
//     lines stay in this XCD's L2);
//   S (sum) roles hold no weights -- all their registers go to loads in flight: per 64-node tile the per-node sum of the
//     node's product rows, read straight back from the ring (L1-bypassing loads, L2 hits), + bias -> out (streaming stores).
// Hand-off per batch through two counters: done[b] (T units of batch b finished) gates the S tiles of b; cdone[b] (S tiles
// of b finished) gates the T roles that want to overwrite b's ring slot D batches later.  S of batch b depends only on T of
// batch b, T of batch b only on S of batch b-D: no cycle.  All workgroups of the grid must be co-resident (grid = 2 per CU).
//
// Same-XCD visibility: a plain store whose vmcnt has retired is in the XCD's L2, and an `nt` load bypasses the reader's L1
// and is served by that same L2 -- so no agent-scope release (L2 write-back) is needed as long as producer and consumer
// share the XCD.  HIP promises nothing about placement, so the kernel CHECKS it: the workgroups of a group record their
// physical XCC_ID; any mismatch -- or a wait that exceeds its wall-clock budget -- raises the abort word, every workgroup
// leaves, and the caller falls back to the two-launch path.  The kernel cannot hang.
#include "dn_common.h"
#include "../../include/dn_hip.h"

namespace {

typedef __bf16 bf16_t;
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 512;     // 8 waves, each owning H/8 output columns (2 x 16 at H = 256)
constexpr int kRows = 32;         // rows per T tile
constexpr int kRowsS = 32;        // nodes per S tile
constexpr int kPad = 8;           // bf16 elements of LDS row padding
constexpr int kLocCache = 640;    // list entries of an S tile staged through LDS (the rest is read from global)

struct PTile { int32_t beg, end, batch, relflags; };     // relflags: relation | first-of-unit << 16 | last-of-unit << 17
struct PRole { int32_t kind, tile_beg, tile_end, pad; }; // kind 0: T, 1: S, anything else: idle
struct PBatch { int32_t rowbase, ringoff, need_c, wait_batch, need_t, pad0, pad1, pad2; };

__device__ __forceinline__ uint64_t now_ticks() { return wall_clock64(); }   // constant 100 MHz

// thread 0 only.  true: flag reached `need`; false: abort raised (by us on timeout, or by anyone else)
__device__ __forceinline__ bool spin_until(int32_t* flag, int32_t need, int32_t* abort_word, uint64_t budget_ticks) {
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) return true;
    const uint64_t t0 = now_ticks();
    for (int it = 0;; ++it) {
        __builtin_amdgcn_s_sleep(2);
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) return true;
        if ((it & 7) == 7) {
            if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
            if (now_ticks() - t0 > budget_ticks) {
                __hip_atomic_store(abort_word, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
}

template <int H, bool STATS>
__global__ __launch_bounds__(kThreads, 4) void rows_pipe_kernel(
    const bf16_t* __restrict__ X, const bf16_t* __restrict__ X2, int32_t n1, const int32_t* __restrict__ row_idx,
    const bf16_t* __restrict__ Wn, const bf16_t* __restrict__ bias, const PRole* __restrict__ roles,
    int32_t num_groups, int32_t roles_per_group, const PTile* __restrict__ tiles, const PBatch* __restrict__ batches,
    const int32_t* __restrict__ list_ptr, const int32_t* __restrict__ list_local,
    bf16_t* __restrict__ ring, int32_t* __restrict__ done, int32_t* __restrict__ cdone, int32_t* __restrict__ grp_xcc,
    int32_t* __restrict__ abort_word, bf16_t* __restrict__ out, uint64_t budget_ticks, int64_t* __restrict__ stats_arg) {
    int64_t* const stats = STATS ? stats_arg : nullptr;      // the instrumented build is its own instantiation (it costs ~14 VGPRs)
    constexpr int SX = H + kPad;
    constexpr int KS = H / 32;
    constexpr int NT = (H / 8 + 15) / 16;
    constexpr int MT = kRows / 16;
    constexpr int LPR = H / 8;                          // 16-byte pieces per row
    constexpr int NP = kRows * LPR;                     // pieces per T tile
    constexpr int PX = (NP + kThreads - 1) / kThreads;  // pieces per thread (T)
    constexpr int NPS = kRowsS * LPR;                   // pieces per S tile
    constexpr int PXS = (NPS + kThreads - 1) / kThreads;
    __shared__ __attribute__((aligned(16))) bf16_t lds[3 * kRows * SX];
    __shared__ int s_state;                             // 0: go on, 1: leave
    __shared__ int32_t lptrL[3][kRowsS + 1];            // S role: list_ptr of the tiles t, t+1, t+2 (stage = tile % 3)
    __shared__ int32_t llocL[2][kLocCache];             // S role: the first kLocCache list entries of the tiles t, t+1
    auto bufX = [&](int b) -> bf16_t* { return lds + b * (kRows * SX); };
    bf16_t* bufY = lds + 2 * kRows * SX;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint64_t st_t0 = 0, st_wait = 0;                    // (thread 0) launch statistics: total / waiting ticks
    uint64_t sec[4] = {0, 0, 0, 0}, mark = 0;           // ... and ticks per section of the tile loop
    if (stats && tid == 0) st_t0 = now_ticks();
    auto lap = [&](int i) { if (stats && tid == 0) { const uint64_t n = now_ticks(); sec[i] += n - mark; mark = n; } };
    auto put_stats = [&](int kind, int ntiles) {
        if (stats && tid == 0) {
            int64_t* s8 = stats + 8 * (size_t)blockIdx.x;
            s8[0] = (int64_t)(now_ticks() - st_t0); s8[1] = (int64_t)st_wait; s8[2] = ntiles; s8[3] = kind;
            s8[4] = (int64_t)sec[0]; s8[5] = (int64_t)sec[1]; s8[6] = (int64_t)sec[2]; s8[7] = (int64_t)sec[3];
        }
    };
    // blocks b and b + 8 share an XCD (observed round-robin dealing; verified below): group = b mod 8, whatever num_groups is
    const int group = (int)(blockIdx.x % DN_NUM_XCD), slot = (int)(blockIdx.x / DN_NUM_XCD);
    if (group >= num_groups || slot >= roles_per_group) return;
    const PRole role = roles[group * roles_per_group + slot];

    // ---- placement check: every workgroup of a group must sit on the same physical XCD -------------------------------------
    if (tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const int mine = (int)(xcc & 15u) + 1;
        const int prev = atomicCAS(grp_xcc + group, 0, mine);
        int st = 0;
        if (prev != 0 && prev != mine) {
            __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            st = 1;
        }
        s_state = st;
    }
    __syncthreads();
    if (s_state != 0) return;
    if (role.kind != 0 && role.kind != 1) return;
    const int t_beg = role.tile_beg, t_end = role.tile_end;
    if (t_beg >= t_end) return;

    if (role.kind == 0) {
        // ============================== T role: gather -> MFMA -> ring ========================================================
        const int n0 = wave * (NT * 16);
        const bool wave_active = n0 < H;
        bf16x8 wf[KS][NT];
        int cur_rel = -1;
        uint4 rx[PX];
        int32_t nidx[PX];
        auto load_idx = [&](const PTile& tl, bool valid) {
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                const int piece = tid + j * kThreads, p = tl.beg + piece / LPR;
                nidx[j] = (valid && piece < NP && p < tl.end) ? row_idx[p] : -1;
            }
        };
        auto load_rows = [&]() {
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                const int c = (tid + j * kThreads) % LPR;
                rx[j] = make_uint4(0, 0, 0, 0);
                if (nidx[j] >= 0) {
                    const bf16_t* base = nidx[j] < n1 ? X + (size_t)nidx[j] * H : X2 + (size_t)(nidx[j] - n1) * H;
                    rx[j] = *reinterpret_cast<const uint4*>(base + c * 8);
                }
            }
        };
        auto store_rows = [&](int b) {
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                const int piece = tid + j * kThreads, rr = piece / LPR, c = piece % LPR;
                if (piece < NP) *reinterpret_cast<uint4*>(bufX(b) + rr * SX + c * 8) = rx[j];
            }
        };
        const PTile none = {0, 0, 0, 0};
        // tile descriptors travel two tiles ahead (tl2), their row indices one and a half, their rows one
        PTile tl = tiles[t_beg];
        PTile tl1 = t_beg + 1 < t_end ? tiles[t_beg + 1] : none;
        load_idx(tl, true);
        load_rows();
        store_rows(0);
        load_idx(tl1, t_beg + 1 < t_end);
        int pending = -1;                                    // batch whose `done` signal is still owed (deferred by one tile)
        __syncthreads();
        for (int t = t_beg; t < t_end; ++t) {
            const int b = (t - t_beg) & 1;
            const int rel = tl.relflags & 0xffff;
            const bool first = (tl.relflags >> 16) & 1, last = (tl.relflags >> 17) & 1;
            if (stats && tid == 0) mark = now_ticks();
            const PTile tl2 = t + 2 < t_end ? tiles[t + 2] : none;      // scalar loads, consumed at the end of the iteration
            const int4 bt = *reinterpret_cast<const int4*>(batches + tl.batch);   // {rowbase, ringoff, need_c, wait_batch}
            if (t + 1 < t_end) load_rows();                  // gather of tile t+1 in flight under this tile's MFMAs
            if (rel != cur_rel && wave_active) {
                cur_rel = rel;
                const bf16_t* w = Wn + (size_t)rel * H * H;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        wf[ks][nt] = *reinterpret_cast<const bf16x8*>(w + (size_t)(n0 + nt * 16 + (lane & 15)) * H + ks * 32 + 8 * (lane >> 4));
            }
            if (wave_active) {
                // D = W_slice x rows^T: lane ends with row m*16 + (lane & 15), columns n0 + n*16 + 4*(lane>>4) + i
                const bf16_t* xt = bufX(b);
                f32x4 acc[MT][NT];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
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
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        bf16x4 o;
#pragma unroll
                        for (int i = 0; i < 4; ++i) o[i] = (bf16_t)acc[m][n][i];
                        *reinterpret_cast<bf16x4*>(bufY + (m * 16 + (lane & 15)) * SX + n0 + n * 16 + 4 * (lane >> 4)) = o;
                    }
            }
            lap(0);
            // rows of tile t+1 into LDS (waits for their gather); then, if tile t-1 closed a unit, make sure its ring stores
            // have retired before the signal they owe goes out below -- by now they have had a whole tile's time, and nothing
            // else of this wave is in flight at this point, so the wait is normally free
            if (t + 1 < t_end) store_rows(b ^ 1);
            if (pending >= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            load_idx(tl2, t + 2 < t_end);
            lap(1);
            __syncthreads();                                 // bufY complete; every wave's stores of tile t-1 are in L2
            // the signal owed for the previous unit goes out BEFORE this role may block on its own ring slot: the S tiles it
            // waits for can themselves be waiting for exactly that signal
            if (pending >= 0 && tid == 0) __hip_atomic_fetch_add(done + pending, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pending = last ? tl.batch : -1;
            if (first && bt.w >= 0) {
                if (tid == 0) {
                    const uint64_t w0 = stats ? now_ticks() : 0;
                    s_state = spin_until(cdone + bt.w, batches[bt.w].need_c, abort_word, budget_ticks) ? 0 : 1;
                    if (stats) st_wait += now_ticks() - w0;
                }
                __syncthreads();
                if (s_state != 0) return;
            }
            // whole rows into the ring slot of this batch: plain stores (the lines stay in this XCD's L2)
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                const int piece = tid + j * kThreads, r = piece / LPR, c = piece % LPR;
                const int p = tl.beg + r;
                if (piece < NP && p < tl.end)
                    *reinterpret_cast<uint4*>(ring + ((size_t)bt.y + (size_t)(p - bt.x)) * H + c * 8) =
                        *reinterpret_cast<const uint4*>(bufY + r * SX + c * 8);
            }
            lap(2);
            __syncthreads();                                 // bufY free again
            lap(3);
            tl = tl1; tl1 = tl2;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the last tile's stores
        __syncthreads();
        if (pending >= 0 && tid == 0) __hip_atomic_fetch_add(done + pending, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        put_stats(0, t_end - t_beg);
        return;
    }

    // ================================== S role: per-node sum of the ring rows + bias -> out ====================================
    constexpr int KU = 6;                                // ring rows requested at once per piece
    // Index data of a tile (its nodes' list_ptr entries, then the list entries themselves) is staged through LDS two / one
    // tile ahead: read on demand it is a chain of three dependent misses per piece (list_ptr -> list_local -> ring row).
    auto fetch_lptr = [&](int t) -> int32_t {            // thread i <= rows(t): list_ptr[beg(t) + i]
        if (t >= t_end || tid > kRowsS) return 0;
        const PTile q = tiles[t];
        return (q.beg + tid <= q.end) ? list_ptr[q.beg + tid] : 0;
    };
    auto fetch_lloc = [&](int t, int32_t (&v)[2]) {      // entries tid and tid + 512 of tile t's list range
        v[0] = v[1] = 0;
        if (t >= t_end) return;
        const PTile q = tiles[t];
        const int32_t* lp = lptrL[(t - t_beg) % 3];
        const int32_t base = lp[0], cnt = lp[q.end - q.beg] - base;
        if (tid < cnt && tid < kLocCache) v[0] = list_local[base + tid];
        if (tid + kThreads < cnt && tid + kThreads < kLocCache) v[1] = list_local[base + tid + kThreads];
    };
    float bv[8];
    {
        const int c = tid % LPR;
#pragma unroll
        for (int i = 0; i < 8; ++i) bv[i] = bias ? (float)bias[c * 8 + i] : 0.f;   // every piece of a thread has the same columns
    }
    {
        const int32_t p0 = fetch_lptr(t_beg), p1 = fetch_lptr(t_beg + 1);
        if (tid <= kRowsS) { lptrL[0][tid] = p0; lptrL[1][tid] = p1; }
    }
    __syncthreads();
    {
        int32_t l0[2];
        fetch_lloc(t_beg, l0);
        if (tid < kLocCache) llocL[0][tid] = l0[0];
        if (tid + kThreads < kLocCache) llocL[0][tid + kThreads] = l0[1];
    }
    __syncthreads();
    for (int t = t_beg; t < t_end; ++t) {
        const int b = (t - t_beg) & 1;
        const PTile tl = tiles[t];
        const bool first = (tl.relflags >> 16) & 1;
        const PBatch bt = batches[tl.batch];
        if (stats && tid == 0) mark = now_ticks();
        const int32_t lp_next = fetch_lptr(t + 2);       // in flight across the sums
        int32_t ll_next[2];
        fetch_lloc(t + 1, ll_next);                      // (its list_ptr stage was written one iteration ago)
        lap(0);
        if (first) {
            if (tid == 0) {
                const uint64_t w0 = stats ? now_ticks() : 0;
                s_state = spin_until(done + tl.batch, bt.need_t, abort_word, budget_ticks) ? 0 : 1;
                if (stats) st_wait += now_ticks() - w0;
            }
            __syncthreads();
            if (s_state != 0) return;
        }
        lap(1);
        const bf16_t* rbase = ring + (size_t)bt.ringoff * H;
        const int32_t* lp = lptrL[(t - t_beg) % 3];
        const int32_t* ll = llocL[b];
        const int32_t lbase = lp[0];
        // all my pieces together: KU ring rows per piece in flight at once (PXS * KU 16-byte loads), fixed order of addition
        int32_t lbeg[PXS], lend[PXS], lmax = 0;
        float a[PXS][8];
#pragma unroll
        for (int j = 0; j < PXS; ++j) {
            const int piece = tid + j * kThreads, r = piece / LPR;
            lbeg[j] = lend[j] = 0;
            if (piece < NPS && tl.beg + r < tl.end) { lbeg[j] = lp[r]; lend[j] = lp[r + 1]; }
            lmax = max(lmax, lend[j] - lbeg[j]);
#pragma unroll
            for (int i = 0; i < 8; ++i) a[j][i] = bv[i];
        }
        const int c = tid % LPR;
        for (int i0 = 0; i0 < lmax; i0 += KU) {
            u32x4 g[PXS][KU];
#pragma unroll
            for (int j = 0; j < PXS; ++j)
#pragma unroll
                for (int k = 0; k < KU; ++k) {
                    g[j][k] = u32x4{0u, 0u, 0u, 0u};
                    const int i = lbeg[j] + i0 + k;
                    if (i < lend[j]) {
                        const int e = i - lbase;
                        const int32_t loc = e < kLocCache ? ll[e] : list_local[i];
                        g[j][k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(rbase + (size_t)loc * H + c * 8));
                    }
                }
#pragma unroll
            for (int j = 0; j < PXS; ++j)
#pragma unroll
                for (int k = 0; k < KU; ++k)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        a[j][2 * i] += __uint_as_float(g[j][k][i] << 16);
                        a[j][2 * i + 1] += __uint_as_float(g[j][k][i] & 0xffff0000u);
                    }
        }
#pragma unroll
        for (int j = 0; j < PXS; ++j) {
            const int piece = tid + j * kThreads, r = piece / LPR;
            const int v = tl.beg + r;
            if (piece < NPS && v < tl.end) {
                bf16x8 o;
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = (bf16_t)a[j][i];
                __builtin_nontemporal_store(__builtin_bit_cast(u32x4, o), reinterpret_cast<u32x4*>(out + (size_t)v * H + c * 8));
            }
        }
        lap(2);
        if (tid <= kRowsS) lptrL[(t + 2 - t_beg) % 3][tid] = lp_next;
        if (tid < kLocCache) llocL[b ^ 1][tid] = ll_next[0];
        if (tid + kThreads < kLocCache) llocL[b ^ 1][tid + kThreads] = ll_next[1];
        __syncthreads();                                 // every ring read of this tile has returned; the stages are free again
        if (tid == 0) __hip_atomic_fetch_add(cdone + tl.batch, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lap(3);
    }
    put_stats(1, t_end - t_beg);
}

}  // namespace

extern "C" {

size_t dn_rows_pipe_sync_words(int64_t num_batches) {
    if (num_batches < 0) { dn_set_error("dn_rows_pipe_sync_words: negative size"); return 0; }
    return (size_t)(2 * num_batches + 16);
}

int dn_rows_pipe_bf16(const void* X, const void* X2, int32_t n1, const int32_t* row_idx, int32_t H, const void* Wn,
                      const void* bias, const int32_t* roles, int32_t num_groups, int32_t roles_per_group,
                      const int32_t* tiles, int64_t num_tiles, const int32_t* batches, int64_t num_batches,
                      const int32_t* list_ptr, const int32_t* list_local, void* ring, int32_t* sync,
                      int64_t N, void* out, int32_t timeout_ms, int64_t* stats, dn_stream_t stream) {
    DN_REQUIRE(H == 64 || H == 128 || H == 256, "dn_rows_pipe: unsupported width %d (64/128/256 only)", H);
    DN_REQUIRE(N >= 0 && N < 0x7fffffffLL && num_tiles >= 0 && num_batches >= 0, "dn_rows_pipe: bad sizes");
    DN_REQUIRE(num_groups >= 1 && num_groups <= 8 && roles_per_group >= 1 && roles_per_group <= 64,
               "dn_rows_pipe: num_groups must be in [1, 8] and roles_per_group in [1, 64] (two workgroups per CU stay resident)");
    if (N == 0 || num_tiles == 0) return DN_OK;
    DN_REQUIRE(X && Wn && roles && tiles && batches && list_ptr && ring && sync && out, "dn_rows_pipe: NULL pointer");
    DN_REQUIRE(X2 != nullptr || n1 == 0x7fffffff, "dn_rows_pipe: X2 == NULL requires n1 == INT32_MAX");
    DN_REQUIRE((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(X2) | reinterpret_cast<uintptr_t>(Wn) |
                reinterpret_cast<uintptr_t>(ring) | reinterpret_cast<uintptr_t>(out)) % 16 == 0, "dn_rows_pipe: unaligned pointer");
    DN_REQUIRE(timeout_ms >= 1 && timeout_ms <= 10000, "dn_rows_pipe: timeout_ms must be in [1, 10000]");
    hipStream_t st = (hipStream_t)stream;
    const size_t words = 2 * (size_t)num_batches + 16;
    DN_CHECK_HIP(hipMemsetAsync(sync, 0, words * sizeof(int32_t), st));
    int32_t* done = sync;
    int32_t* cdone = sync + num_batches;
    int32_t* grp_xcc = sync + 2 * num_batches;
    int32_t* abort_word = sync + 2 * num_batches + 8;
    const uint64_t budget = (uint64_t)timeout_ms * 100000ull;              // wall_clock64 ticks at 100 MHz
    const dim3 grid((unsigned)(DN_NUM_XCD * roles_per_group)), block(kThreads);
    const bf16_t *x = (const bf16_t*)X, *x2 = (const bf16_t*)X2, *w = (const bf16_t*)Wn, *bb = (const bf16_t*)bias;
    const PRole* rl = reinterpret_cast<const PRole*>(roles);
    const PTile* tl = reinterpret_cast<const PTile*>(tiles);
    const PBatch* bt = reinterpret_cast<const PBatch*>(batches);
#define DN_PIPE_LAUNCH(HH)                                                                                                   \
    if (stats) DN_PIPE_LAUNCH2(HH, true); else DN_PIPE_LAUNCH2(HH, false)
#define DN_PIPE_LAUNCH2(HH, ST)                                                                                              \
    hipLaunchKernelGGL((rows_pipe_kernel<HH, ST>), grid, block, 0, st, x, x2, n1, row_idx, w, bb, rl, num_groups,       \
                       roles_per_group, tl, bt, list_ptr, list_local, (bf16_t*)ring, done, cdone, grp_xcc,            \
                       abort_word, (bf16_t*)out, budget, stats)
    if (H == 256) DN_PIPE_LAUNCH(256);
    else if (H == 128) DN_PIPE_LAUNCH(128);
    else DN_PIPE_LAUNCH(64);
#undef DN_PIPE_LAUNCH
#undef DN_PIPE_LAUNCH2
    DN_CHECK_LAUNCH();
    return DN_OK;
}

}  // extern "C"
