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
    for (int w : {1, 2, 4, 6, 8}) run<1>(w, d);
    for (int w : {1, 2, 4}) run<2>(w, d);
    for (int w : {1, 2}) run<4>(w, d);
    return 0;
}
