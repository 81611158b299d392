// Shared helpers for the dn_hip C-ABI library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define DN_OK 0
#define DN_ERR_ARG (-1)
#define DN_ERR_HIP (-2)
#define DN_ERR_WORKSPACE (-3)
#define DN_ERR_UNSUPPORTED (-4)

// thread-local last-error string (dn_error.cpp)
void dn_set_error(const char* fmt, ...);

#define DN_CHECK_HIP(expr)                                                              \
    do {                                                                                \
        hipError_t dn_e_ = (expr);                                                      \
        if (dn_e_ != hipSuccess) {                                                      \
            dn_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,                  \
                         hipGetErrorString(dn_e_));                                     \
            return DN_ERR_HIP;                                                          \
        }                                                                               \
    } while (0)

#define DN_REQUIRE(cond, ...)                                                           \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            dn_set_error(__VA_ARGS__);                                                  \
            return DN_ERR_ARG;                                                          \
        }                                                                               \
    } while (0)

#define DN_CHECK_LAUNCH() DN_CHECK_HIP(hipGetLastError())

static inline int64_t dn_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t dn_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Experiment knobs.  The shipped library has NO environment access of its own: every knob is its compile-time default (the
// measured best).  A tuning build (-DDN_TUNING_ENV: `python -m dummynode4graphlearning_amd.csrc.build --tuning`, used by
// tools/ab.sh) reads the DN_* variables instead.
#ifdef DN_TUNING_ENV
#include <stdlib.h>
static inline int dn_knob(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
#else
static constexpr int dn_knob(const char*, int dflt) { return dflt; }
#endif

// MI355X: 8 XCDs, workgroups are dealt round-robin over them (blocks b and b+8 share an XCD's L2).
// Map the hardware block id to a logical chunk id so that the blocks of one XCD walk a CONTIGUOUS
// range of chunks (neighbouring graphs -> same L2).  Speed only, never correctness.
#define DN_NUM_XCD 8
__device__ __forceinline__ int64_t dn_xcd_chunk(int64_t b, int64_t nblocks) {
    const int64_t per = (nblocks + DN_NUM_XCD - 1) / DN_NUM_XCD;
    return (b % DN_NUM_XCD) * per + b / DN_NUM_XCD;
}

// LDS-DMA issued from inline asm (cdna_hip_programming.md, inline-asm section: M0 written in the statement that reads
// it): hipcc's s_waitcnt pass treats a builtin LDS-DMA as a pending LDS write and drains vmcnt(0) before EVERY later
// ds_read, which would empty the ring each tile; an asm DMA is outside its bookkeeping and is counted by hand below.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

__device__ __forceinline__ void glds4(const void* gsrc, unsigned lds_dst) {      // 4 bytes per lane: lane l lands at lds_dst + 4 l
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
