// Tuning aid (not part of libdn_hip.so): what rate does this part give a bare "gather R rows of 512 B, add, write one row"
// stream?  R = 1 is the traffic shape of rows_transform_kernel (1 read : 1 write), R = 3 of rows_selfsum_kernel (3 : 1).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int R, int U>
__global__ __launch_bounds__(256) void gather_sum_kernel(const uint4* __restrict__ X, const int32_t* __restrict__ idx, int64_t P,
                                                         uint4* __restrict__ Y, int nt) {
    // 32 lanes per row (16 B each); a lane group walks rows p = g, g + G, ...; U rows in flight per group
    const int64_t G = (int64_t)gridDim.x * 8, g = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    const int c = threadIdx.x & 31;
    for (int64_t p0 = g; p0 < P; p0 += G * U) {
        uint4 v[U][R];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t p = p0 + (int64_t)u * G;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                v[u][k] = make_uint4(0, 0, 0, 0);
                if (p < P) v[u][k] = X[(int64_t)idx[p * R + k] * 32 + c];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t p = p0 + (int64_t)u * G;
            if (p >= P) continue;
            float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const uint32_t w[4] = {v[u][k].x, v[u][k].y, v[u][k].z, v[u][k].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a[2 * i] += __uint_as_float(w[i] << 16);
                    a[2 * i + 1] += __uint_as_float(w[i] & 0xffff0000u);
                }
            }
            u32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                o[i] = (__float_as_uint(a[2 * i]) >> 16) | (__float_as_uint(a[2 * i + 1]) & 0xffff0000u);
            if (nt) __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(Y + p * 32 + c));
            else *reinterpret_cast<u32x4*>(Y + p * 32 + c) = o;
        }
    }
}

extern "C" int gather_sum(const void* X, const int32_t* idx, int64_t P, int R, int U, int blocks, int nt, void* Y, void* stream) {
    hipStream_t st = (hipStream_t)stream;
#define GO(r, u) hipLaunchKernelGGL((gather_sum_kernel<r, u>), dim3(blocks), dim3(256), 0, st, (const uint4*)X, idx, P, (uint4*)Y, nt)
    if (R == 1 && U == 4) GO(1, 4); else if (R == 1 && U == 8) GO(1, 8); else if (R == 1 && U == 2) GO(1, 2);
    else if (R == 3 && U == 2) GO(3, 2); else if (R == 3 && U == 4) GO(3, 4); else if (R == 3 && U == 1) GO(3, 1);
    else return -1;
    return (int)hipGetLastError();
}
