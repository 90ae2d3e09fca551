// Shared helpers for the gfx950 kernels (wave64, MFMA f32 32x32x2).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "../../include/ssd_gfx950.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SSD_CHECK_LAUNCH() do { if (hipGetLastError() != hipSuccess) return SSD_ERR_LAUNCH; } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a property of (kernel, DEVICE): a call site keeps one bit per device in a flag
// word of its own (devices 0 .. 63; a race only sets the idempotent attribute twice).
static inline bool ssd_attr_needed(const std::atomic<unsigned long long>& seen, int& dev) {
    dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return true;
    return ((seen.load(std::memory_order_relaxed) >> (dev & 63)) & 1ull) == 0;
}
static inline void ssd_attr_done(std::atomic<unsigned long long>& seen, int dev) { seen.fetch_or(1ull << (dev & 63), std::memory_order_relaxed); }

static inline bool ssd_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int ssd_cdiv(int a, int b) { return (a + b - 1) / b; }

// Round-robin dispatch puts block b on XCD b%8 (speed only, never correctness:
// cdna_hip_programming.md T1).  Map so that each XCD walks a contiguous range
// of logical tiles and neighbouring tiles share operand panels in one L2.
__device__ __forceinline__ int xcd_swizzle(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, j = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + j;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
