// Probe for the fused Winograd kernel's inner loop: v_mfma_f32_16x16x4_f32, ONE wave per SIMD (256-thread block per CU, 288 accumulator
// registers in the real kernel), two accumulator chains, operands (a) in registers, (b) re-read from LDS by 12 ds_read_b128 per 32 MFMAs
// with the kernel's XOR swizzle, (c) + one s_barrier per 32 MFMAs, (d) + accumulators rotating over 36 planes.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma16_probe.hip -o tools/mfma16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, const float* in, int stages, unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) float lds[5 * 6144];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 5 * 6144; i += 256) lds[i] = in[i & 4095];
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    const int r15 = lane & 15, kq = lane >> 4;
    const int a_rd = r15 * 64, b_rd = 32 * 64 + (wave * 16 + r15) * 64;
    int sw[4];
    for (int j = 0; j < 4; ++j) sw[j] = ((4 * j + kq) ^ r15) * 4;
    f32x4 acc[36][2];
#pragma unroll
    for (int x = 0; x < 36; ++x) { acc[x][0] = f32x4{0, 0, 0, 0}; acc[x][1] = f32x4{0, 0, 0, 0}; }
    f32x4 fa0[4], fa1[4], fb[4];
    for (int j = 0; j < 4; ++j) { fa0[j] = *(f32x4*)(lds + a_rd + sw[j]); fa1[j] = *(f32x4*)(lds + a_rd + 1024 + sw[j]); fb[j] = *(f32x4*)(lds + b_rd + sw[j]); }
    int slot = 0;
    for (int it = 0; it < stages / 36; ++it) {
#pragma unroll
        for (int x = 0; x < 36; ++x) {
            const int xi = MODE >= 3 ? x : 0;
            f32x4 na0[4], na1[4], nb[4];
            if (MODE >= 2) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
            if (MODE >= 1) {
                slot = slot + 1 == 5 ? 0 : slot + 1;
                const float* st = lds + slot * 6144;
#pragma unroll
                for (int j = 0; j < 4; ++j) { na0[j] = *(const f32x4*)(st + a_rd + sw[j]); na1[j] = *(const f32x4*)(st + a_rd + 1024 + sw[j]); nb[j] = *(const f32x4*)(st + b_rd + sw[j]); }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0[j][e], fb[j][e], acc[xi][0], 0, 0, 0);
                    acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1[j][e], fb[j][e], acc[xi][1], 0, 0, 0);
                }
            if (MODE >= 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { fa0[j] = na0[j]; fa1[j] = na1[j]; fb[j] = nb[j]; }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int x = 0; x < 36; ++x) s += acc[x][0][0] + acc[x][1][3];
    if (s == 12345.f) out[tid] = s;
    if (blockIdx.x == 0 && tid == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = wall_clock64() - w0; }
}

template <int MODE> void run(float* d, const float* in, const char* what) {
    const int stages = 36 * 400;
    unsigned long long* clk; hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, d, in, stages, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flops = 256.0 * 4 * stages * 32 * 2048;
    printf("%-70s %7.3f ms  %6.1f TFLOP/s  %5.0f cycles/stage (32 MFMAs: 1024 at the issue rate)  clock %.2f GHz\n", what, ms, flops / ms / 1e9,
           (double)h[0] / stages, (double)h[0] / ((double)h[1] * 10.0) );
}

int main() {
    float *d, *in; hipMalloc(&d, 4096); hipMalloc(&in, 4096 * 4);
    float h[4096]; unsigned x = 12345u;
    for (int i = 0; i < 4096; ++i) { x = x * 1664525u + 1013904223u; h[i] = ((int)(x >> 8) - (1 << 23)) / (float)(1 << 22); }
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<0>(d, in, "16x16x4 f32, 2 chains, operands in registers, 1 wave/SIMD");
    run<1>(d, in, "+ 12 ds_read_b128 per 32 MFMAs (next stage's fragments, swizzled)");
    run<2>(d, in, "+ s_barrier per stage");
    run<3>(d, in, "+ accumulators rotate over 36 planes (288 registers)");
    return 0;
}
