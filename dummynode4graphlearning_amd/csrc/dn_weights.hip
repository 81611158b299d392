// Block-diagonal ("bdd") relation weights of the reference's RGIN / RGCN layers as dense [R, in, out] matrices, and the gradient
// back to the blocks -- one launch each, in place of the broadcast-multiply with an identity mask the Python layer used
// (subgraph_isomorphism/models/rgin.py:114-120: `weight.index_select(0, etype).view(-1, B, si, so)` + per-block bmm; here the
// blocks of relation r are laid on the diagonal of its dense matrix once per step, so the relation transform runs on the same
// MFMA kernels as `basis`).
#include "dn_common.h"
#include "../../include/dn_hip.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void bdd_compose_kernel(const T* __restrict__ blocks, int32_t B, int32_t si, int32_t so,
                                                          int64_t total, T* __restrict__ dense) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;             // dense element (r, i, o)
    if (e >= total) return;
    const int64_t Ho = (int64_t)B * so, Hi = (int64_t)B * si;
    const int64_t o = e % Ho, i = (e / Ho) % Hi, r = e / (Ho * Hi);
    const int64_t bi = i / si, bo = o / so;
    dense[e] = bi == bo ? blocks[((r * B + bi) * si + i % si) * so + o % so] : T(0);
}

template <typename T>
__global__ __launch_bounds__(256) void bdd_extract_kernel(const T* __restrict__ dense, int32_t B, int32_t si, int32_t so,
                                                          int64_t total, T* __restrict__ blocks) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;             // block element (r, b, i, o)
    if (e >= total) return;
    const int64_t o = e % so, i = (e / so) % si, b = (e / ((int64_t)so * si)) % B, r = e / ((int64_t)so * si * B);
    blocks[e] = dense[(r * ((int64_t)B * si) + b * si + i) * ((int64_t)B * so) + b * so + o];
}

template <typename T>
int bdd_launch(const void* src, int64_t R, int32_t B, int32_t si, int32_t so, void* dst, bool compose, hipStream_t st) {
    const int64_t total = compose ? R * B * si * B * so : R * B * si * so;
    if (total == 0) return DN_OK;
    const unsigned grid = (unsigned)dn_cdiv(total, 256);
    if (compose) hipLaunchKernelGGL(bdd_compose_kernel<T>, dim3(grid), dim3(256), 0, st, (const T*)src, B, si, so, total, (T*)dst);
    else hipLaunchKernelGGL(bdd_extract_kernel<T>, dim3(grid), dim3(256), 0, st, (const T*)src, B, si, so, total, (T*)dst);
    DN_CHECK_LAUNCH();
    return DN_OK;
}

int bdd_check(int64_t R, int32_t B, int32_t si, int32_t so, const void* a, const void* b, int32_t elem_bytes) {
    DN_REQUIRE(R >= 0 && B > 0 && si > 0 && so > 0 && elem_bytes > 0, "dn_bdd: bad sizes");
    DN_REQUIRE(R * (int64_t)B * si * B * so < (int64_t)1 << 40, "dn_bdd: too large");
    DN_REQUIRE(R == 0 || (a && b), "dn_bdd: NULL pointer");
    return DN_OK;
}

}  // namespace

extern "C" {

int dn_bdd_compose(const void* blocks, int64_t R, int32_t B, int32_t si, int32_t so, int32_t elem_bytes, void* dense,
                   dn_stream_t stream) {
    int rc = bdd_check(R, B, si, so, blocks, dense, elem_bytes);
    if (rc != DN_OK) return rc;
    DN_REQUIRE(elem_bytes == 2 || elem_bytes == 4, "dn_bdd_compose: elem_bytes must be 2 (bf16) or 4 (f32)");
    if (elem_bytes == 2) return bdd_launch<__bf16>(blocks, R, B, si, so, dense, true, (hipStream_t)stream);
    return bdd_launch<float>(blocks, R, B, si, so, dense, true, (hipStream_t)stream);
}

int dn_bdd_extract(const void* dense, int64_t R, int32_t B, int32_t si, int32_t so, int32_t elem_bytes, void* blocks,
                   dn_stream_t stream) {
    int rc = bdd_check(R, B, si, so, dense, blocks, elem_bytes);
    if (rc != DN_OK) return rc;
    DN_REQUIRE(elem_bytes == 2 || elem_bytes == 4, "dn_bdd_extract: elem_bytes must be 2 (bf16) or 4 (f32)");
    if (elem_bytes == 2) return bdd_launch<__bf16>(dense, R, B, si, so, blocks, false, (hipStream_t)stream);
    return bdd_launch<float>(dense, R, B, si, so, blocks, false, (hipStream_t)stream);
}

}  // extern "C"
