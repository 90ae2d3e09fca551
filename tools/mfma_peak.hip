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

// Decomposition of the igemm main loop (64x64 tile, 4 waves, K step 32): what each ingredient costs the MFMA rate.
//   LEVEL 0: 16 MFMAs per step on registers           1: + 8 ds_read_b128 per step feeding them (conflict-free rows)
//   LEVEL 2: + the two block barriers per step         3: + 4 ds_write_b128 per step between the barriers
//   LEVEL 4: + 4 global 16-byte loads per step (L2-resident source) whose data the ds_writes store
//   LEVEL 5: levels 0-2 + 4 global -> LDS DMA loads of 16 bytes per lane per step (no staging registers, no ds_write)
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int LEVEL>
__global__ __launch_bounds__(256) void k_loop(float* out, const float* __restrict__ src, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[128 * 36];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5, wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < 128 * 36; i += 256) lds[i] = src[i & 4095];
    __syncthreads();
    const float* a_rd = lds + (wm * 32 + lr) * 36 + lh * 4;
    const float* b_rd = lds + (64 + wn * 32 + lr) * 36 + lh * 4;
    const int chunk = tid & 7, row0 = tid >> 3;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f32x4v st[4] = {{1.f, 2.f, 3.f, 4.f}, {1.f, 2.f, 3.f, 4.f}, {1.f, 2.f, 3.f, 4.f}, {1.f, 2.f, 3.f, 4.f}};
    const f32x4v* g4 = reinterpret_cast<const f32x4v*>(src);
    unsigned goff = (blockIdx.x * 256 + tid) & 1023;
    for (int it = 0; it < iters; ++it) {
        if (LEVEL == 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) st[j] = g4[(goff + 256 * j + it * 64) & 1023];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4v af, bf;
            if (LEVEL >= 1) {
                af = *reinterpret_cast<const f32x4v*>(a_rd + q * 8);
                bf = *reinterpret_cast<const f32x4v*>(b_rd + q * 8);
            } else {
                af = st[q]; bf = st[(q + 1) & 3];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc, 0, 0, 0);
        }
        if (LEVEL >= 2) __syncthreads();
        if (LEVEL == 5) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g4 + ((goff + 256 * j + it * 64) & 1023)),
                                                 (__attribute__((address_space(3))) void*)(lds + (wave * 4 + j) * 256), 16, 0, 0);
            __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0)
        }
        if (LEVEL >= 3 && LEVEL < 5) {
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4v*>(&lds[(row0 + 32 * j) * 36 + chunk * 4]) = st[j];
        }
        if (LEVEL >= 2) __syncthreads();
    }
    float sum = 0.f;
    for (int r = 0; r < 16; ++r) sum += acc[r];
    if (sum == 12345.f) out[tid] = sum;
}

// The full register-staged loop (level 4) for a block tile of (64*TM) x (64*TN): each wave owns TM x TN accumulators
template <int TM, int TN>
__global__ __launch_bounds__(256) void k_loop_tile(float* out, const float* __restrict__ src, int iters) {
    constexpr int ROWS = 64 * TM + 64 * TN, NL = ROWS / 32;
    __shared__ __attribute__((aligned(16))) float lds[ROWS * 36];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5, wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < ROWS * 36; i += 256) lds[i] = src[i & 4095];
    __syncthreads();
    const float* a_rd = lds + (wm * 32 * TM + lr) * 36 + lh * 4;
    const float* b_rd = lds + (64 * TM + wn * 32 * TN + lr) * 36 + lh * 4;
    const int chunk = tid & 7, row0 = tid >> 3;
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i)
        for (int j = 0; j < TN; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4v st[NL];
    const f32x4v* g4 = reinterpret_cast<const f32x4v*>(src);
    unsigned goff = (blockIdx.x * 256 + tid) & 1023;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NL; ++j) st[j] = g4[(goff + 256 * j + it * 64) & 1023];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4v af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4v*>(a_rd + i * 32 * 36 + q * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4v*>(b_rd + j * 32 * 36 + q * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NL; ++j) *reinterpret_cast<f32x4v*>(&lds[(row0 + 32 * j) * 36 + chunk * 4]) = st[j];
        __syncthreads();
    }
    float sum = 0.f;
    for (int i = 0; i < TM; ++i)
        for (int j = 0; j < TN; ++j)
            for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    if (sum == 12345.f) out[tid] = sum;
}

template <int TM, int TN>
void run_tile(int blocks_per_cu, float* d, const float* in) {
    const int blocks = 256 * blocks_per_cu, iters = 3000 / (TM * TN);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_loop_tile<TM, TN>), dim3(blocks), dim3(256), 0, 0, d, in, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_loop_tile<TM, TN>), dim3(blocks), dim3(256), 0, 0, d, in, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * 16 * TM * TN * 4096.0;
    printf("igemm loop, block tile %dx%d, %d blocks/CU : %.3f ms  %.1f TFLOP/s\n", 64 * TM, 64 * TN, blocks_per_cu, ms, flop / ms / 1e9);
}

// LEVEL 6: two LDS stages; the 4 LDS-DMA loads of the next tile are issued BEFORE this tile's MFMAs, retired after them; one barrier
template <int DUMMY>
__global__ __launch_bounds__(256) void k_loop_dma2(float* out, const float* __restrict__ src, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 128 * 36];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5, wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < 2 * 128 * 36; i += 256) lds[i] = src[i & 4095];
    __syncthreads();
    const int a_off = (wm * 32 + lr) * 36 + lh * 4, b_off = (64 + wn * 32 + lr) * 36 + lh * 4;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const f32x4v* g4 = reinterpret_cast<const f32x4v*>(src);
    unsigned goff = (blockIdx.x * 256 + tid) & 1023;
    int cur = 0;
    for (int it = 0; it < iters; ++it) {
        float* nxt = lds + (cur ^ 1) * 128 * 36;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g4 + ((goff + 256 * j + it * 64) & 1023)),
                                             (__attribute__((address_space(3))) void*)(nxt + (wave * 4 + j) * 256), 16, 0, 0);
        const float* st = lds + cur * 128 * 36;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4v af = *reinterpret_cast<const f32x4v*>(st + a_off + q * 8);
            const f32x4v bf = *reinterpret_cast<const f32x4v*>(st + b_off + q * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc, 0, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0): this wave's DMA pieces have landed
        __builtin_amdgcn_s_barrier();
        cur ^= 1;
    }
    float sum = 0.f;
    for (int r = 0; r < 16; ++r) sum += acc[r];
    if (sum == 12345.f) out[tid] = sum;
}

void run_dma2(int blocks_per_cu, float* d, const float* in) {
    const int blocks = 256 * blocks_per_cu, iters = 3000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_loop_dma2<0>, dim3(blocks), dim3(256), 0, 0, d, in, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_loop_dma2<0>, dim3(blocks), dim3(256), 0, 0, d, in, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * 16 * 4096.0;
    printf("igemm loop level 6 (2 LDS stages, DMA under the MFMAs, 1 barrier), %d blocks/CU : %.3f ms  %.1f TFLOP/s\n", blocks_per_cu, ms, flop / ms / 1e9);
}

template <int LEVEL>
void run_loop(int blocks_per_cu, float* d, const float* in) {
    const int blocks = 256 * blocks_per_cu, iters = 3000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_loop<LEVEL>, dim3(blocks), dim3(256), 0, 0, d, in, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_loop<LEVEL>, dim3(blocks), dim3(256), 0, 0, d, in, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * 16 * 4096.0;
    printf("igemm loop level %d, %d blocks/CU : %.3f ms  %.1f TFLOP/s\n", LEVEL, blocks_per_cu, ms, flop / ms / 1e9);
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
    for (int b : {6, 3}) {
        run_loop<0>(b, d, in); run_loop<1>(b, d, in); run_loop<2>(b, d, in); run_loop<3>(b, d, in); run_loop<4>(b, d, in); run_loop<5>(b, d, in);
    }
    for (int b : {4, 3, 2}) run_dma2(b, d, in);
    run_tile<1, 1>(6, d, in); run_tile<2, 1>(5, d, in); run_tile<2, 1>(4, d, in); run_tile<1, 2>(5, d, in);
    run_tile<2, 2>(4, d, in); run_tile<2, 2>(3, d, in); run_tile<2, 2>(2, d, in);
    for (int w : {2, 8}) run_rand<1>(w, d, in, 0.f);
    for (int w : {2, 8}) run_rand<1>(w, d, in, 0.5f);
    for (int w : {2, 8}) run_rand<1>(w, d, in, 1.0f);
    run_rand<2>(4, d, in, 0.f);
    for (int w : {1, 2, 4, 6, 8}) run<1>(w, d);
    for (int w : {1, 2, 4}) run<2>(w, d);
    for (int w : {1, 2}) run<4>(w, d);
    return 0;
}
