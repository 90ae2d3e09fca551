// Achievable v_mfma_f32_32x32x2_f32 rate on this chip with no memory traffic: dependent chains of length CH per wave,
// W waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CH>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CH; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    if (s == 12345.f) out[threadIdx.x] = s;
}

// same loop on RANDOM operands (16 a / 16 b values per lane, cycled): the clock the chip holds depends on the data
template <int CH>
__global__ __launch_bounds__(256) void k_rand(float* out, const float* in, int iters, float zero_frac, unsigned long long* clk) {
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a[8], b[8];
    for (int j = 0; j < 8; ++j) {
        a[j] = in[(threadIdx.x + 256 * j) & 4095];
        b[j] = in[(threadIdx.x + 256 * j + 2048) & 4095];
        if (a[j] < 2.f * zero_frac - 1.f) a[j] = 0.f;          // zero_frac of the A operands are exact zeros (post-ReLU data)
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u * CH + c) & 7], b[(u + c * 3) & 7], acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CH; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    if (s == 12345.f) out[threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = __builtin_readcyclecounter() - c0;          // shader clock cycles
        clk[1] = wall_clock64() - w0;                        // constant 100 MHz ticks
    }
}

template <int CH>
void run_rand(int waves_per_simd, float* d, const float* in, float zero_frac) {
    const int blocks = 256 * waves_per_simd;
    const int iters = 20000 / CH;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned long long* clk;
    hipMalloc(&clk, 16);
    hipLaunchKernelGGL(k_rand<CH>, dim3(blocks), dim3(256), 0, 0, d, in, iters, zero_frac, clk);
    hipDeviceSynchronize();
    float best = 1e9f, last = 0.f;
    for (int rep = 0; rep < 8; ++rep) {                 // sustained: eight launches back to back (~20 ms of load each at 8 waves)
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_rand<CH>, dim3(blocks), dim3(256), 0, 0, d, in, iters, zero_frac, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&last, e0, e1);
        if (last < best) best = last;
    }
    const double flop = (double)blocks * 4 * iters * 4 * CH * 4096.0;
    unsigned long long h[2];
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("RANDOM operands (%.0f %% zeros in A)  chains/wave %d  waves/SIMD %d : best %.3f ms %.1f TFLOP/s, last (warm chip) %.3f ms %.1f TFLOP/s, "
           "shader clock of the last launch %.0f MHz\n", zero_frac * 100, CH, waves_per_simd, best, flop / best / 1e9, last, flop / last / 1e9,
           (double)h[0] / ((double)h[1] / 100.0));
}

template <int CH>
void run(int waves_per_simd, float* d) {
    const int blocks = 256 * waves_per_simd;          // 4 waves per block = 1 wave per SIMD per block
    const int iters = 20000 / CH;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * 4 * CH * 4096.0;
    printf("chains/wave %d  waves/SIMD %d : %.3f ms  %.1f TFLOP/s\n", CH, waves_per_simd, ms, flop / ms / 1e9);
}

int main() {
    float* d;
    hipMalloc(&d, 4096);
    float* in;
    hipMalloc(&in, 4096 * 4);
    {
        float h[4096];
        unsigned x = 12345u;
        for (int i = 0; i < 4096; ++i) { x = x * 1664525u + 1013904223u; h[i] = ((int)(x >> 8) - (1 << 23)) / (float)(1 << 22); }
        hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    }
    for (int w : {2, 8}) run_rand<1>(w, d, in, 0.f);
    for (int w : {2, 8}) run_rand<1>(w, d, in, 0.5f);
    for (int w : {2, 8}) run_rand<1>(w, d, in, 1.0f);
    run_rand<2>(4, d, in, 0.f);
    for (int w : {1, 2, 4, 6, 8}) run<1>(w, d);
    for (int w : {1, 2, 4}) run<2>(w, d);
    for (int w : {1, 2}) run<4>(w, d);
    return 0;
}
