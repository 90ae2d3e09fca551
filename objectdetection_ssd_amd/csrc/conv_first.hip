// conv1_1 (Conv2d(3, 64, 3, padding=1) + ReLU on the caller's NCHW batch, Model.py:135 features[0:2]) in one kernel.
//
// The im2col form (elementwise.hip: im2col_first + a 1x1 MFMA convolution) writes 369 MB of [pixel][32] rows at batch 32 and reads
// them back; the convolution itself is bound by its 737 MB of output.  Here a workgroup takes a 4 x 64 pixel tile: the 6 x 66 x 3 input
// halo goes to LDS once (4.7 KB), every wave multiplies one tile row -- two blocks of 32 pixels x 64 channels x K = 27 (+1 zero) on
// v_mfma_f32_32x32x2_f32, the A operand read straight from the halo image (k = (r*3+s)*3 + c selects a constant LDS offset, lanes
// walk the pixels) -- and bias + ReLU + the NHWC store close it.  When the weight gradient wants them (training), the same workgroup
// also writes its [pixel][32] rows from the halo image, so that the separate unfold pass and its read of x disappear as well.
#include <type_traits>
#include "common.h"

namespace {

constexpr int TH = 4, TW = 64, HH = TH + 2, HW_ = TW + 2, PLANE = HH * HW_;       // tile, halo, floats per channel plane of the halo

// LDS offset (floats) of tap k = (r*3+s)*3 + c relative to the pixel's own position in plane 0; k = 27 (the zero column of the weights)
// re-reads tap 26: a finite product with a zero weight unless x itself holds a NaN there, which the true sum would carry anyway
__device__ __forceinline__ constexpr int tap_off(int k) {
    const int kk = k < 27 ? k : 26;
    return (kk % 3) * PLANE + (kk / 9) * HW_ + (kk / 3) % 3;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// BF16 (the bf16-tensor mode, BASELINE.json configs[2]): x and the filter are rounded to bf16 on their way into LDS -- the products
// of two bf16 values are exact in the f32 MFMA, so this is the bf16-operand convolution with f32 accumulation -- and y is stored
// as bf16 NHWC (`y` then points to bf16).
// PERSISTENT (round 4): a workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ...  The filter rows go to registers once; the halo of
// the NEXT tile is requested before this tile's MFMAs and parked in the other LDS buffer after them, so that a tile's loads, its MFMAs
// and its 64 KB of stores overlap those of its neighbours in time instead of queueing behind the ~2 us of first-load latency every
// one-tile workgroup paid (0.224 -> 0.165 ms at batch 32; same MFMA chain per output element: bit-identical results).
template <bool BF16>
__global__ __launch_bounds__(256) void conv_first_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wrows,
                                                             const float* __restrict__ bias, float* __restrict__ y, float* __restrict__ col,
                                                             int N, int H, int W, int tiles_h, int tiles_w, int relu, int ntiles) {
    auto rnd = [](float v) { return BF16 ? (float)(__bf16)v : v; };
    constexpr int XS = 3 * PLANE + 8;
    constexpr int EPT = (3 * PLANE + 255) / 256;                       // halo elements per thread
    __shared__ float xs2[2][XS];
    __shared__ float ws[64 * 33];
    __shared__ __attribute__((aligned(16))) float ys[4 * 32 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t HWs = (size_t)H * W;
    // this thread's halo elements e = tid + 256 u: (channel, halo row, halo column) never change, only the tile origin does
    int e_c[EPT], e_r[EPT], e_cc[EPT];
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        const int e = tid + 256 * u;
        const int c = e / PLANE, rem = e - c * PLANE, r = rem / HW_;
        e_c[u] = e < 3 * PLANE ? c : -1;
        e_r[u] = r;
        e_cc[u] = rem - r * HW_;
    }
    auto origin = [&](int t, int& n, int& h0, int& w0) {
        const int twi = t % tiles_w, thi = (t / tiles_w) % tiles_h;
        n = t / (tiles_w * tiles_h);
        h0 = thi * TH;
        w0 = twi * TW;
    };
    float hv[EPT];
    auto fetch = [&](int t) {
        int n, h0, w0;
        origin(t, n, h0, w0);
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            const int ih = h0 - 1 + e_r[u], iw = w0 - 1 + e_cc[u];
            hv[u] = (e_c[u] >= 0 && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
                        ? rnd(x[((size_t)n * 3 + e_c[u]) * HWs + (size_t)ih * W + iw]) : 0.f;
        }
    };
    auto park = [&](float* xs) {
#pragma unroll
        for (int u = 0; u < EPT; ++u)
            if (e_c[u] >= 0) xs[tid + 256 * u] = hv[u];
    };
    int tile = blockIdx.x;
    if (tile < ntiles) fetch(tile);
    // the 64 x 32 filter rows: two coalesced 16-byte loads per thread into LDS (row stride 33: the fragment reads below walk the rows)
    {
        const f32x4 w0v = *reinterpret_cast<const f32x4*>(wrows + tid * 4), w1v = *reinterpret_cast<const f32x4*>(wrows + 1024 + tid * 4);
        const int r0 = tid >> 3, c0 = (tid & 7) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ws[r0 * 33 + c0 + e] = rnd(w0v[e]);
            ws[(r0 + 32) * 33 + c0 + e] = rnd(w1v[e]);
        }
    }
    if (tile < ntiles) park(xs2[0]);
    const int lr = lane & 31, lh = lane >> 5;
    __syncthreads();
    float bw[2][14];                                                   // B operand: w[co = 32 i + lr][k = 2 q + lh]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 14; ++q) bw[i][q] = ws[(32 * i + lr) * 33 + 2 * q + lh];
    float bv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) bv[i] = bias != nullptr ? bias[32 * i + lr] : 0.f;

    for (int it = 0; tile < ntiles; ++it, tile += gridDim.x) {
        const float* xs = xs2[it & 1];
        int n, h0, w0;
        origin(tile, n, h0, w0);
        const int next = tile + gridDim.x;
        if (next < ntiles) fetch(next);                                // in flight across this tile's MFMAs
        f32x16 acc[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
        const int base = wave * HW_ + lr;                              // the wave's tile row, this lane's pixel of a 32-pixel block
#pragma unroll
        for (int q = 0; q < 14; ++q) {
            const int off = lh ? tap_off(2 * q + 1) : tap_off(2 * q);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float a = xs[base + 32 * j + off];
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bw[i][q], acc[j][i], 0, 0, 0);
            }
        }

        // C/D map: channel = 32 i + (lane & 31), pixel of the block = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5): a lane holds single channels of
        // 16 pixels.  Stored like that (4 bytes per lane, 128-byte runs) the kernel writes at ~3 TB/s; so each 32-pixel block goes through
        // a per-wave LDS image [pixel][64 ch] and leaves as 16 bytes per lane, 1 KB (four whole pixels) per instruction.
        const int oh = h0 + wave;
        float* ysw = ys + wave * (32 * 64);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[j][i][r] + bv[i];
                    if (relu) v = v < 0.f ? 0.f : v;                   // NaN stays NaN, like torch.relu
                    ysw[((r & 3) + 8 * (r >> 2) + 4 * lh) * 64 + 32 * i + lr] = v;
                }
            // same wave writes and reads: the LDS queue is in order, no barrier
            if (oh < H) {
                if constexpr (BF16) {                                  // 8 lanes per pixel (128 bytes of bf16), 8 pixels per instruction
                    __bf16* po = reinterpret_cast<__bf16*>(y) + (((size_t)n * H + oh) * W + w0 + 32 * j) * 64 + (lane & 7) * 8;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int px = 8 * t + (lane >> 3);
                        const f32x4 a = *reinterpret_cast<const f32x4*>(ysw + px * 64 + (lane & 7) * 8);
                        const f32x4 b = *reinterpret_cast<const f32x4*>(ysw + px * 64 + (lane & 7) * 8 + 4);
                        if (w0 + 32 * j + px < W)
                            *reinterpret_cast<bf16x8*>(po + (size_t)px * 64) = bf16x8{(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3],
                                                                                      (__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
                    }
                } else {
                    float* po = y + (((size_t)n * H + oh) * W + w0 + 32 * j) * 64 + (lane & 15) * 4;
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const int px = 4 * t + (lane >> 4);
                        const f32x4 v = *reinterpret_cast<const f32x4*>(ysw + px * 64 + (lane & 15) * 4);
                        if (w0 + 32 * j + px < W) *reinterpret_cast<f32x4*>(po + (size_t)px * 64) = v;
                    }
                }
            }
        }
        if (col != nullptr) {                                          // uniform: the [pixel][32] rows of the weight gradient
            // 8 threads per pixel (128 contiguous bytes), 32 pixels per pass; a thread keeps its 16-byte chunk, i.e. its four taps
            const int chunk = tid & 7;
            int toff[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = chunk * 4 + e;
                toff[e] = k < 27 ? (k % 3) * PLANE + (k / 9) * HW_ + (k / 3) % 3 : -1;
            }
#pragma unroll
            for (int pi = 0; pi < 8; ++pi) {
                const int pix = pi * 32 + (tid >> 3), py = pix >> 6, px = pix & 63;
                if (h0 + py < H && w0 + px < W) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = toff[e] >= 0 ? xs[py * HW_ + px + toff[e]] : 0.f;
                    *reinterpret_cast<f32x4*>(col + ((((size_t)n * H + h0 + py) * W + w0 + px) * 8 + chunk) * 4) = v;
                }
            }
        }
        if (next < ntiles) park(xs2[(it + 1) & 1]);                    // (the buffer the PREVIOUS tile read: every wave passed the barrier below since)
        __syncthreads();
    }
}


// ---- conv1_1 + ReLU written DIRECTLY as the F(4x4,3x3) input planes of conv1_2 (round 4; SSD_EXPERIMENTAL builds only) ---------------------
// MEASURED: correct and bit-identical, but SLOWER than the two kernels it replaces: 0.964 ms against conv1_1 0.248 + conv1_2's input transform
// 0.52 (bench.py --layers, batch 32); in the step 20.16 against 19.97 ms (8 tiles per workgroup), 20.36 against 20.25 (16 tiles).  The 1.47 GB it
// keeps out of HBM are worth 0.3 ms, but a 6 x 34 patch costs 3x conv1_1's MFMA work (6 rows for 4, two 32-column blocks for 34 columns) and the
// workgroup's load / MFMA / LDS / transform phases do not overlap at two workgroups per CU.  Kept as the tested starting point (the two halo
// columns as ONE gathered MFMA block per workgroup would halve the MFMA work); `_Engine.first_wino` is off.
#ifdef SSD_EXPERIMENTAL
// In training the only reader of conv1_1's output is conv1_2 (Model.py:135 features[0:4]), a Winograd layer: its input transform reads the
// 737 MB activation (batch 32) that this kernel has just written, and nothing else ever needs it -- the backward gates conv1_2's data gradient
// through the ReLU BITS the transform leaves, conv1_2's weight gradient multiplies the kept planes, conv1_1's own weight gradient reads x and
// dy.  So a workgroup computes the 6 x 66 patch of a1_1 under one tile row segment of 16 tiles (rows 4 th - 1 .. 4 th + 4, the 64 columns + one on
// each side; 2.25x conv1_1's small MFMA work: K = 27), parks it in LDS, and writes V = B^T d B for its 16 tiles x 64 channels and the bit words.
// Positions outside the map are ZERO in the patch (conv1_2's zero padding), not conv1_1 evaluated out there.  The MFMA chain per output
// element and the transform's sums are those of conv_first_fwd_kernel and wino4_input_kernel (csrc/winograd.hip): planes and bits are
// bit-identical to the two-kernel form (tests/test_gpu_kernels.py::test_conv1_1_written_as_winograd_planes).
#ifndef FW_FT
#define FW_FT 8        // tiles per workgroup.  16: 123 KB of LDS = one workgroup per CU, nothing overlaps its load / MFMA / transform phases (measured: the
#endif                 // fused kernel slower than the two it replaces); 8: 68 KB, two workgroups per CU cover each other's phases
constexpr int FT = FW_FT, FNB = (4 * FT + 2 + 31) / 32;              // column blocks of 32 per patch row
constexpr int FPR = 6, FPC = 4 * FT + 2, FXR = FPR + 2, FXC = FPC + 2, FXPLANE = FXR * FXC;     // tiles / patch rows x cols / input halo
constexpr int FPS = 68;                                               // floats per patch pixel in LDS: 64 channels + 4 (16-byte aligned, spreads the banks)
constexpr int FW_XS = 3 * FXPLANE + 64, FW_WS = 64 * 33, FW_PATCH = FPR * FPC * FPS;
constexpr size_t FW_LDS_BYTES = (size_t)(FW_XS + FW_WS + FW_PATCH) * 4;
__device__ constexpr float FW_BT[6][6] = {{4, 0, -5, 0, 1, 0}, {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0},
                                          {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};
__device__ __forceinline__ constexpr int ftap_off(int k) {
    const int kk = k < 27 ? k : 26;
    return (kk % 3) * FXPLANE + (kk / 9) * FXC + (kk / 3) % 3;
}

__global__ __launch_bounds__(384) void conv_first_wino_kernel(const float* __restrict__ x, const float* __restrict__ wrows, const float* __restrict__ bias,
                                                              float* __restrict__ V, unsigned long long* __restrict__ bits, int N, int H, int W,
                                                              int TH, int TW, int nblk_w) {
    extern __shared__ __attribute__((aligned(16))) float fw_lds[];
    float* xs = fw_lds;
    float* ws = fw_lds + FW_XS;
    float* patch = fw_lds + FW_XS + FW_WS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bx = blockIdx.x;
    const int twb = bx % nblk_w, th = (bx / nblk_w) % TH, n = bx / (nblk_w * TH);
    const int h0 = 4 * th - 1, w0 = 4 * FT * twb - 1;                    // map position of patch element (0, 0)
    const size_t HWs = (size_t)H * W;
    for (int e = tid; e < FW_XS; e += 384) {
        float v = 0.f;
        if (e < 3 * FXPLANE) {
            const int c = e / FXPLANE, rem = e - c * FXPLANE, r = rem / FXC, cc = rem - r * FXC;
            const int ih = h0 - 1 + r, iw = w0 - 1 + cc;
            if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) v = x[((size_t)n * 3 + c) * HWs + (size_t)ih * W + iw];
        }
        xs[e] = v;
    }
    if (tid < 256) {
        const f32x4 w0v = *reinterpret_cast<const f32x4*>(wrows + tid * 4), w1v = *reinterpret_cast<const f32x4*>(wrows + 1024 + tid * 4);
        const int r0 = tid >> 3, c0 = (tid & 7) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ws[r0 * 33 + c0 + e] = w0v[e];
            ws[(r0 + 32) * 33 + c0 + e] = w1v[e];
        }
    }
    const int lr = lane & 31, lh = lane >> 5;
    __syncthreads();
    {
        // wave = patch row; FNB blocks of 32 columns (the last holds the two halo columns and 30 unused ones) x two halves of the 64 channels
        float bw[2][14];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 14; ++q) bw[i][q] = ws[(32 * i + lr) * 33 + 2 * q + lh];
        f32x16 acc[FNB][2];
#pragma unroll
        for (int j = 0; j < FNB; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
        const int base = wave * FXC + lr;
#pragma unroll
        for (int q = 0; q < 14; ++q) {
            const int off = lh ? ftap_off(2 * q + 1) : ftap_off(2 * q);
#pragma unroll
            for (int j = 0; j < FNB; ++j) {
                const float a = xs[base + 32 * j + off];
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bw[i][q], acc[j][i], 0, 0, 0);
            }
        }
        const int oh = h0 + wave;
        float bv[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) bv[i] = bias != nullptr ? bias[32 * i + lr] : 0.f;
#pragma unroll
        for (int j = 0; j < FNB; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pc = 32 * j + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (pc < FPC) {
                        const int ow = w0 + pc;
                        float v = acc[j][i][r] + bv[i];
                        v = v < 0.f ? 0.f : v;                               // NaN stays NaN, like torch.relu
                        if ((unsigned)oh >= (unsigned)H || (unsigned)ow >= (unsigned)W) v = 0.f;     // conv1_2's zero padding
                        patch[(wave * FPC + pc) * FPS + 32 * i + lr] = v;
                    }
                }
    }
    __syncthreads();
    if (tid < 16 * FT) {
        const int t = tid >> 4, c4 = tid & 15;
        const int tw = FT * twb + t;
        if (tw < TW) {
            const size_t tiles = (size_t)N * TH * TW, plane = tiles * 64;
            const size_t tile = ((size_t)n * TH + th) * TW + tw;
            f32x4 tt[6][6];                                          // B^T d, one patch column at a time (wino4_input_kernel's order)
            unsigned long long word = 0ull;
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                f32x4 d[6];
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    d[a] = *reinterpret_cast<const f32x4*>(patch + ((a * FPC) + 4 * t + b) * FPS + c4 * 4);
                    if (a >= 1 && a <= 4 && b >= 1 && b <= 4) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) word |= (unsigned long long)(d[a][e] > 0.f) << (((a - 1) * 4 + (b - 1)) * 4 + e);
                    }
                }
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < 6; ++k)
                        if (FW_BT[a][k] != 0.f) acc2 += FW_BT[a][k] * d[k];
                    tt[a][b] = acc2;
                }
            }
            float* dst = V + tile * 64 + c4 * 4;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = 0; b < 6; ++b) {
                    f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < 6; ++k)
                        if (FW_BT[b][k] != 0.f) acc2 += FW_BT[b][k] * tt[a][k];
                    *reinterpret_cast<f32x4*>(dst + (size_t)(a * 6 + b) * plane) = acc2;
                }
            if (bits != nullptr) bits[tile * 16 + c4] = word;
        }
    }
}
#endif  // SSD_EXPERIMENTAL


// Weight + bias gradient of conv1_1 from the NCHW input itself (no [pixel][32] rows in memory):
//   dw[co][k] = sum_p dy[p][co] * x[p + tap(k)][c(k)],  k = (r*3+s)*3 + c < 27;   column 27 multiplies ones: db[co] = sum_p dy[p][co].
// Same 4 x 64 pixel tiles and halo image as the forward kernel; the reduction runs over the pixels: per 32-pixel block of its tile row a
// wave stages dy [32 px][64 ch] in LDS (16-byte loads) and issues 16 x 2 v_mfma_f32_32x32x2_f32 (A = dy, lanes walk the channels;
// B = the halo image at this lane's tap offset, or 1 for lane 27).  Workgroups are persistent (grid-stride over the tiles); their
// [64][32] partial sums go to a slab that conv_first_wgrad_reduce_kernel adds up in block order (reproducible).
// DY_BF16: dy is the bf16 NHWC gradient of the bf16-tensor mode (x stays the caller's f32 image; f32 products and sums)
template <bool DY_BF16>
__global__ __launch_bounds__(256) void conv_first_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ slab,
                                                               int N, int H, int W, int tiles_h, int tiles_w, int ntiles) {
    __shared__ float xs[3 * PLANE + 8];
    __shared__ __attribute__((aligned(16))) float ds[4 * 32 * 64];      // per wave: dy of one 32-pixel block; at the end: the waves' partial sums
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const size_t HWs = (size_t)H * W;
    const int my_off = lr < 27 ? tap_off(lr) : 0;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float* dsw = ds + wave * (32 * 64);
    // Round 4: the NEXT tile's operands are requested while this tile is multiplied -- the dy registers of a 32-pixel block as soon as that
    // block has been parked in LDS, the halo pixels behind them -- so that a tile's ~2 us of load latency hides under the previous tile's
    // 64 MFMAs per wave instead of in front of its own (same sums in the same order: bit-identical).
    constexpr int EPT = (3 * PLANE + 255) / 256;                       // halo elements per thread
    int e_c[EPT], e_r[EPT], e_cc[EPT];
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        const int e = tid + 256 * u;
        const int c = e / PLANE, rem = e - c * PLANE, r = rem / HW_;
        e_c[u] = e < 3 * PLANE ? c : -1;
        e_r[u] = r;
        e_cc[u] = rem - r * HW_;
    }
    typedef typename std::conditional<DY_BF16, bf16x4, f32x4>::type dy4;
    dy4 dv[2][8];
    float hv[EPT];
    auto origin = [&](int t, int& n, int& h0, int& w0) {
        const int twi = t % tiles_w, thi = (t / tiles_w) % tiles_h;
        n = t / (tiles_w * tiles_h);
        h0 = thi * TH;
        w0 = twi * TW;
    };
    // dy of this wave's tile row, block j of 32 pixels: lane = (pixel 4 t + lane / 16, channel quad lane % 16); zero outside the image
    auto fetch_dy = [&](int t_, int j) {
        int n, h0, w0;
        origin(t_, n, h0, w0);
        const int oh = h0 + wave;
        const size_t so = (((size_t)n * H + oh) * W + w0 + 32 * j) * 64 + (lane & 15) * 4;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int px = 4 * t + (lane >> 4);
            if constexpr (DY_BF16) dv[j][t] = bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
            else dv[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (oh < H && w0 + 32 * j + px < W) {
                if constexpr (DY_BF16) dv[j][t] = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(dy) + so + (size_t)px * 64);
                else dv[j][t] = *reinterpret_cast<const f32x4*>(dy + so + (size_t)px * 64);
            }
        }
    };
    auto fetch_halo = [&](int t_) {
        int n, h0, w0;
        origin(t_, n, h0, w0);
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            const int ih = h0 - 1 + e_r[u], iw = w0 - 1 + e_cc[u];
            hv[u] = (e_c[u] >= 0 && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
                        ? x[((size_t)n * 3 + e_c[u]) * HWs + (size_t)ih * W + iw] : 0.f;
        }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) {
        fetch_dy(tile, 0);
        fetch_dy(tile, 1);
        fetch_halo(tile);
    }
    for (; tile < ntiles; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        __syncthreads();                                               // the previous tile's halo image is no longer read
#pragma unroll
        for (int u = 0; u < EPT; ++u)
            if (e_c[u] >= 0) xs[tid + 256 * u] = hv[u];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                f32x4 v;
                if constexpr (DY_BF16) v = f32x4{(float)dv[j][t][0], (float)dv[j][t][1], (float)dv[j][t][2], (float)dv[j][t][3]};
                else v = dv[j][t];
                *reinterpret_cast<f32x4*>(dsw + (4 * t + (lane >> 4)) * 64 + (lane & 15) * 4) = v;
            }
            if (next < ntiles) {                                       // this block's registers are free again: the next tile's block j
                fetch_dy(next, j);
                if (j == 1) fetch_halo(next);
            }
            // the wave reads what it wrote itself: the LDS queue is in order
            const int bbase = wave * HW_ + 32 * j + my_off;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int p = 2 * q + lh;                              // pixel of the block: the reduction index of this MFMA half
                const float b = lr == 27 ? 1.f : xs[bbase + p];
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(dsw[p * 64 + 32 * i + lr], b, acc[i], 0, 0, 0);
            }
        }
    }
    // D[m = channel][n = tap]: lane holds tap lr of channels 32 i + (reg & 3) + 8 (reg >> 2) + 4 lh; add the four waves in wave order
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) ds[wave * 2048 + (32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + lr] = acc[i][r];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int idx = e * 256 + tid;
        slab[(size_t)blockIdx.x * 2048 + idx] = ((ds[idx] + ds[2048 + idx]) + ds[4096 + idx]) + ds[6144 + idx];
    }
}

// dw_rows[co][k] (k < 27; 27..31 zero) and db[co] = column 27, summed over the workgroups' partial slabs in a fixed order: a block owns 8
// of the 2048 entries, its 256 threads = 8 entries x 32 slices of the slab list (four running sums each), then the slices in order
__global__ __launch_bounds__(256) void conv_first_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw_rows, float* __restrict__ db,
                                                                      int nblk) {
    __shared__ float red[32][9];
    const int e = threadIdx.x & 7, sl = threadIdx.x >> 3;
    const int idx = blockIdx.x * 8 + e;                                // 0 .. 2047
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b = sl; b < nblk; b += 128) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (b + 32 * u < nblk) s[u] += slab[(size_t)(b + 32 * u) * 2048 + idx];
    }
    red[sl][e] = (s[0] + s[1]) + (s[2] + s[3]);
    __syncthreads();
    if (sl == 0) {
        float v = red[0][e];
        for (int k = 1; k < 32; ++k) v += red[k][e];
        const int k = idx & 31, co = idx >> 5;
        dw_rows[idx] = k < 27 ? v : 0.f;
        if (k == 27 && db != nullptr) db[co] = v;
    }
}

}  // namespace

// Workgroups of the persistent forward kernel.  Measured at batch 32 (bench.py --layers; 12 000 tiles): one workgroup per tile 0.224 ms; grids of
// 512 / 1 024 / 1 536 workgroups 0.164-0.168 ms; 256, 640, 768, 1 280 (tile strides that put the concurrently written tiles on fewer
// channels) 0.183-0.208.  bf16 output: 0.247 -> 0.20 ms.
constexpr long long g_first_grid = 1024;
static int conv1_first_fwd_impl(const float* x_nchw, const float* w_rows, const float* bias, void* y_nhwc, float* col_out, int N, int H,
                                int W, int relu, void* stream, bool bf16) {
    if (!x_nchw || !w_rows || !y_nhwc) return SSD_ERR_NULL;
    if (N <= 0 || H <= 0 || W <= 0) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(y_nhwc) || (col_out && !ssd_aligned16(col_out))) return SSD_ERR_ALIGN;
    const int tiles_h = ssd_cdiv(H, TH), tiles_w = ssd_cdiv(W, TW);
    const long long blocks = (long long)N * tiles_h * tiles_w;
    if (blocks >= (1ll << 31)) return SSD_ERR_BAD_SHAPE;
    // persistent: three workgroups fit a CU (LDS: two halo buffers + filter + the four waves' output images = 51 KB)
    const unsigned grid = (unsigned)(blocks < g_first_grid ? blocks : g_first_grid);
    if (bf16)
        hipLaunchKernelGGL(conv_first_fwd_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x_nchw, w_rows, bias,
                           static_cast<float*>(y_nhwc), col_out, N, H, W, tiles_h, tiles_w, relu, (int)blocks);
    else
        hipLaunchKernelGGL(conv_first_fwd_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x_nchw, w_rows, bias,
                           static_cast<float*>(y_nhwc), col_out, N, H, W, tiles_h, tiles_w, relu, (int)blocks);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}
extern "C" int ssd_conv1_first_fwd(const float* x_nchw, const float* w_rows, const float* bias, float* y_nhwc, float* col_out, int N, int H,
                                   int W, int relu, void* stream) {
    return conv1_first_fwd_impl(x_nchw, w_rows, bias, y_nhwc, col_out, N, H, W, relu, stream, false);
}
// bf16-tensor mode: x and the filter rounded to bf16 (f32 accumulate), y stored as bf16 NHWC
// conv1_1 (+ bias + ReLU) as the F(4x4) input planes of the 64 -> 64 convolution behind it: V [36][N * TH * TW][64] f32 (TH = ceil(H / 4),
// TW = ceil(W / 4): ssd_conv3x3_wino_*'s tile grid at dilation 1) and, optionally, the ReLU bit words (N * TH * TW x 16).
extern "C" int ssd_conv1_first_wino_fwd(const float* x_nchw, const float* w_rows, const float* bias, float* planes, uint64_t* relu_bits, int N,
                                        int H, int W, void* stream) {
#ifndef SSD_EXPERIMENTAL
    return SSD_ERR_BAD_SHAPE;                       // not in this build (SSD_EXPERIMENTAL=1 python -m objectdetection_ssd_amd.build): measured slower
#else
    if (!x_nchw || !w_rows || !planes) return SSD_ERR_NULL;
    if (N <= 0 || H <= 0 || W <= 0) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(w_rows) || !ssd_aligned16(planes) || (relu_bits && ((uintptr_t)relu_bits & 7))) return SSD_ERR_ALIGN;
    const int TH_ = (H + 3) / 4, TW_ = (W + 3) / 4, nbw = (TW_ + FT - 1) / FT;
    const size_t nblk = (size_t)N * TH_ * nbw;
    if (nblk >= (1ull << 31) || (size_t)N * TH_ * TW_ >= (1ull << 31)) return SSD_ERR_BAD_SHAPE;
    static std::atomic<unsigned long long> raised{0};
    int dev;
    if (ssd_attr_needed(raised, dev)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_first_wino_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FW_LDS_BYTES) !=
            hipSuccess)
            return SSD_ERR_LAUNCH;
        ssd_attr_done(raised, dev);
    }
    hipLaunchKernelGGL(conv_first_wino_kernel, dim3((unsigned)nblk), dim3(384), FW_LDS_BYTES, (hipStream_t)stream, x_nchw, w_rows, bias, planes,
                       reinterpret_cast<unsigned long long*>(relu_bits), N, H, W, TH_, TW_, nbw);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
#endif
}

extern "C" int ssd_conv1_first_fwd_bf16(const float* x_nchw, const float* w_rows, const float* bias, void* y_nhwc_bf16, int N, int H, int W,
                                        int relu, void* stream) {
    return conv1_first_fwd_impl(x_nchw, w_rows, bias, y_nhwc_bf16, nullptr, N, H, W, relu, stream, true);
}

// Workgroups of the weight-gradient kernel: two rounds of the 768 the chip holds (3 per CU); each leaves 8 KB of partial sums.
constexpr int WGRAD_BLOCKS = 1536;
extern "C" size_t ssd_conv1_first_wgrad_workspace(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)WGRAD_BLOCKS * 2048 * sizeof(float);
}

static int conv1_first_wgrad_impl(const float* x_nchw, const void* dy_nhwc, float* dw_rows, float* dbias, int N, int H, int W,
                                  void* workspace, size_t workspace_bytes, void* stream, bool dy_bf16);
extern "C" int ssd_conv1_first_wgrad(const float* x_nchw, const float* dy_nhwc, float* dw_rows, float* dbias, int N, int H, int W,
                                     void* workspace, size_t workspace_bytes, void* stream) {
    return conv1_first_wgrad_impl(x_nchw, dy_nhwc, dw_rows, dbias, N, H, W, workspace, workspace_bytes, stream, false);
}
extern "C" int ssd_conv1_first_wgrad_bf16(const float* x_nchw, const void* dy_nhwc_bf16, float* dw_rows, float* dbias, int N, int H, int W,
                                          void* workspace, size_t workspace_bytes, void* stream) {
    return conv1_first_wgrad_impl(x_nchw, dy_nhwc_bf16, dw_rows, dbias, N, H, W, workspace, workspace_bytes, stream, true);
}
static int conv1_first_wgrad_impl(const float* x_nchw, const void* dy_nhwc, float* dw_rows, float* dbias, int N, int H, int W,
                                  void* workspace, size_t workspace_bytes, void* stream, bool dy_bf16) {
    if (!x_nchw || !dy_nhwc || !dw_rows || !workspace) return SSD_ERR_NULL;
    if (N <= 0 || H <= 0 || W <= 0) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(dy_nhwc) || !ssd_aligned16(workspace)) return SSD_ERR_ALIGN;
    const int tiles_h = ssd_cdiv(H, TH), tiles_w = ssd_cdiv(W, TW);
    const long long ntiles = (long long)N * tiles_h * tiles_w;
    if (ntiles >= (1ll << 31)) return SSD_ERR_BAD_SHAPE;
    const int blocks = ntiles < WGRAD_BLOCKS ? (int)ntiles : WGRAD_BLOCKS;
    if (workspace_bytes < (size_t)blocks * 2048 * sizeof(float)) return SSD_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* slab = static_cast<float*>(workspace);
    if (dy_bf16)
        hipLaunchKernelGGL(conv_first_wgrad_kernel<true>, dim3(blocks), dim3(256), 0, st, x_nchw, static_cast<const float*>(dy_nhwc), slab, N, H, W,
                           tiles_h, tiles_w, (int)ntiles);
    else
        hipLaunchKernelGGL(conv_first_wgrad_kernel<false>, dim3(blocks), dim3(256), 0, st, x_nchw, static_cast<const float*>(dy_nhwc), slab, N, H, W,
                           tiles_h, tiles_w, (int)ntiles);
    SSD_CHECK_LAUNCH();
    hipLaunchKernelGGL(conv_first_wgrad_reduce_kernel, dim3(256), dim3(256), 0, st, slab, dw_rows, dbias, blocks);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}
