// HBM-bound NHWC kernels of the bf16-tensor mode (BASELINE.json configs[2]; activations and gradients stored in bf16):
// max pooling (Model.py:135-142 pools), conv4_3 L2 normalisation (Model.py:206-209), the heads' gradient gather, casts.
// Same arithmetic as csrc/elementwise.hip in f32 registers; tensors move as 8 channels = 16 bytes per lane, results are
// rounded to bf16 once, at the store.  Max pooling of bf16 values is exact (max commutes with the monotone rounding).
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

inline int grid_for(size_t total, int block = 256, int cap = 4096) {
    size_t b = (total + block - 1) / block;
    return (int)(b > (size_t)cap ? cap : (b == 0 ? 1 : b));
}

__device__ __forceinline__ f32x8 ld8(const __bf16* p) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
    f32x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)v[e];
    return o;
}
__device__ __forceinline__ void st8(__bf16* p, const f32x8 v) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
    *reinterpret_cast<bf16x8*>(p) = o;
}

// thread = (n, oh, ow, 8 channels); torch semantics: the first valid element seeds the max, a later one replaces it if larger or NaN
__global__ __launch_bounds__(256) void maxpool_fwd_bf16_kernel(const __bf16* __restrict__ x, __bf16* __restrict__ y, uint8_t* __restrict__ am, int N,
                                                               int H, int W, int C, int k, int stride, int pad, int Ho, int Wo) {
    const int C8 = C >> 3;
    const size_t total = (size_t)N * Ho * Wo * C8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c8 = (int)(i % C8);
        size_t rest = i / C8;
        const int ow = (int)(rest % Wo);
        rest /= Wo;
        const int oh = (int)(rest % Ho), n = (int)(rest / Ho);
        f32x8 best;
        uint64_t code = 0;
        bool first = true;
        for (int r = 0; r < k; ++r) {
            const int ih = oh * stride - pad + r;
            if (ih < 0 || ih >= H) continue;
            for (int s = 0; s < k; ++s) {
                const int iw = ow * stride - pad + s;
                if (iw < 0 || iw >= W) continue;
                const f32x8 v = ld8(x + (((size_t)n * H + ih) * W + iw) * C + c8 * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (first || v[e] > best[e] || v[e] != v[e]) {
                        best[e] = v[e];
                        code = (code & ~(0xffull << (8 * e))) | ((uint64_t)(r * k + s) << (8 * e));
                    }
                }
                first = false;
            }
        }
        st8(y + i * 8, best);
        if (am) *reinterpret_cast<uint64_t*>(am + i * 8) = code;
    }
}

// gather form (any k / stride / pad): every input element sums dy over the windows whose argmax it is; optional gate by the pooled
// output (y > 0 = the ReLU mask of the pool's input), optional += dx, optional ReLU mask from the activation itself
__global__ __launch_bounds__(256) void maxpool_bwd_bf16_kernel(const __bf16* __restrict__ dy, const uint8_t* __restrict__ am, __bf16* __restrict__ dx,
                                                               const __bf16* __restrict__ mask, const __bf16* __restrict__ y_gate, int accumulate,
                                                               int N, int H, int W, int C, int k, int stride, int pad, int Ho, int Wo) {
    const int C8 = C >> 3;
    const size_t total = (size_t)N * H * W * C8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c8 = (int)(i % C8);
        size_t rest = i / C8;
        const int iw = (int)(rest % W);
        rest /= W;
        const int ih = (int)(rest % H), n = (int)(rest / H);
        f32x8 g = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < k; ++r) {
            const int th = ih + pad - r;
            if (th < 0 || th % stride != 0) continue;
            const int oh = th / stride;
            if (oh >= Ho) continue;
            for (int s = 0; s < k; ++s) {
                const int tw = iw + pad - s;
                if (tw < 0 || tw % stride != 0) continue;
                const int ow = tw / stride;
                if (ow >= Wo) continue;
                const size_t o = (((size_t)n * Ho + oh) * Wo + ow) * C + c8 * 8;
                const uint64_t a = *reinterpret_cast<const uint64_t*>(am + o);
                f32x8 d = ld8(dy + o);
                if (y_gate != nullptr) {
                    const f32x8 yv = ld8(y_gate + o);
#pragma unroll
                    for (int e = 0; e < 8; ++e) d[e] = yv[e] > 0.f ? d[e] : 0.f;
                }
                const uint64_t me = (uint64_t)(r * k + s);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (((a >> (8 * e)) & 0xffull) == me) g[e] += d[e];
            }
        }
        __bf16* dst = dx + i * 8;
        if (accumulate) g += ld8(dst);
        if (mask) {
            const f32x8 m = ld8(mask + i * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] = m[e] > 0.f ? g[e] : 0.f;
        }
        st8(dst, g);
    }
}

// scatter form of the gated 2x2 / stride-2 / no-pad pool (sole consumer): one thread per window reads dy, codes and gate once
__global__ __launch_bounds__(256) void maxpool2_bwd_scatter_bf16_kernel(const __bf16* __restrict__ dy, const uint8_t* __restrict__ am,
                                                                        const __bf16* __restrict__ y_gate, __bf16* __restrict__ dx, int N, int H,
                                                                        int W, int C, int Ho, int Wo) {
    const int C8 = C >> 3;
    const size_t total = (size_t)N * Ho * Wo * C8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c8 = (int)(i % C8);
        size_t rest = i / C8;
        const int ow = (int)(rest % Wo);
        rest /= Wo;
        const int oh = (int)(rest % Ho), n = (int)(rest / Ho);
        f32x8 d = ld8(dy + i * 8);
        const f32x8 yv = ld8(y_gate + i * 8);
        const uint64_t a = *reinterpret_cast<const uint64_t*>(am + i * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] = yv[e] > 0.f ? d[e] : 0.f;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int ih = 2 * oh + r;
            if (ih >= H) continue;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int iw = 2 * ow + q;
                if (iw >= W) continue;
                f32x8 g;
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = ((a >> (8 * e)) & 0xffull) == (uint64_t)(r * 2 + q) ? d[e] : 0.f;
                st8(dx + ((((size_t)n * H + ih) * W + iw) * C8 + c8) * 8, g);
            }
        }
    }
}

// L2 norm over C = 512 channels: one wave per pixel, 8 channels per lane; no epsilon (Model.py:207)
__global__ __launch_bounds__(256) void l2norm_fwd_bf16_kernel(const __bf16* __restrict__ x, const float* __restrict__ gamma, __bf16* __restrict__ y,
                                                              int M) {
    const int lane = threadIdx.x & 63;
    for (int m = blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += gridDim.x * 4) {
        const f32x8 v = ld8(x + (size_t)m * 512 + lane * 8);
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) ss += v[e] * v[e];
        ss = wave_sum(ss);
        const float nrm = sqrtf(ss);
        f32x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = v[e] / nrm * gamma[lane * 8 + e];
        st8(y + (size_t)m * 512 + lane * 8, o);
    }
}

// u = x/|x|, gdy = gamma*dy:  dx = (gdy - u*(u.gdy))/|x| ;  dgamma_c = sum_m dy_c*u_c (per-block partials, fixed-order sum by the caller)
__global__ __launch_bounds__(256) void l2norm_bwd_bf16_kernel(const __bf16* __restrict__ x, const float* __restrict__ gamma,
                                                              const __bf16* __restrict__ dy, __bf16* __restrict__ dx, float* __restrict__ dg_slab,
                                                              int M) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ float red[4][512];
    f32x8 dg = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = gamma[lane * 8 + e];
    for (int m = blockIdx.x * 4 + wave; m < M; m += gridDim.x * 4) {
        const size_t o = (size_t)m * 512 + lane * 8;
        const f32x8 v = ld8(x + o), d = ld8(dy + o);
        float ss = 0.f, dot = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ss += v[e] * v[e];
            dot += v[e] * d[e] * g[e];
        }
        ss = wave_sum(ss);
        dot = wave_sum(dot);
        const float inv = 1.f / sqrtf(ss);
        const float coef = dot * inv * inv * inv;
        f32x8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            r[e] = g[e] * d[e] * inv - v[e] * coef;
            dg[e] += d[e] * v[e] * inv;
        }
        st8(dx + o, r);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wave][lane * 8 + e] = dg[e];
    __syncthreads();
    for (int c = threadIdx.x; c < 512; c += 256) dg_slab[(size_t)blockIdx.x * 512 + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}

// out[c] = sum_k slab[k][c], c < 512: 16 columns per block, 16 threads per column stride over the slabs (16 loads in flight per column
// instead of one dependent chain), then a fixed-order LDS sum -- reproducible
__global__ __launch_bounds__(256) void slab_sum512_kernel(const float* __restrict__ slab, float* __restrict__ out, int nslab) {
    __shared__ float part[16][17];
    const int c = threadIdx.x & 15, l = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + c;
    float s = 0.f;
    for (int k = l; k < nslab; k += 16) s += slab[(size_t)k * 512 + i];
    part[l][c] = s;
    __syncthreads();
    if (l == 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += part[k][c];
        out[i] = t;
    }
}

// packed[m][c] (bf16, ld columns, pad columns zero) from dloc (N,P,4) / dconf (N,P,ncls) f32
__global__ void heads_gather_bf16_kernel(const float* __restrict__ dloc, const float* __restrict__ dconf, __bf16* __restrict__ packed, int ld,
                                         int N, int HW, int A, int prior_off, int P, int ncls) {
    const int cw = A * (4 + ncls);
    const size_t total = (size_t)N * HW * ld;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % ld);
        const size_t m = i / ld;
        const int n = (int)(m / HW), pix = (int)(m % HW);
        const size_t pbase = (size_t)n * P + prior_off + (size_t)pix * A;
        float v = 0.f;
        if (c < 4 * A) v = dloc[pbase * 4 + c];
        else if (c < cw) v = dconf[pbase * ncls + (c - 4 * A)];
        packed[i] = (__bf16)v;
    }
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ x, __bf16* __restrict__ y, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const f32x4 a = reinterpret_cast<const f32x4*>(x)[2 * i], b = reinterpret_cast<const f32x4*>(x)[2 * i + 1];
        st8(y + i * 8, f32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]});
    }
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const __bf16* __restrict__ x, float* __restrict__ y, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const f32x8 v = ld8(x + i * 8);
        reinterpret_cast<f32x4*>(y)[2 * i] = f32x4{v[0], v[1], v[2], v[3]};
        reinterpret_cast<f32x4*>(y)[2 * i + 1] = f32x4{v[4], v[5], v[6], v[7]};
    }
}

}  // namespace

extern "C" int ssd_maxpool_fwd_bf16(const void* x, void* y, uint8_t* argmax, int N, int H, int W, int C, int k, int stride, int pad, int Ho,
                                    int Wo, void* stream) {
    if (!x || !y) return SSD_ERR_NULL;
    if (C % 8 != 0 || k <= 0 || k > 15 || stride <= 0 || pad < 0 || N <= 0 || Ho <= 0 || Wo <= 0) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(x) || !ssd_aligned16(y) || (argmax && ((uintptr_t)argmax & 7))) return SSD_ERR_ALIGN;
    const size_t total = (size_t)N * Ho * Wo * (C / 8);
    hipLaunchKernelGGL(maxpool_fwd_bf16_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, static_cast<const __bf16*>(x),
                       static_cast<__bf16*>(y), argmax, N, H, W, C, k, stride, pad, Ho, Wo);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

extern "C" int ssd_maxpool_bwd_bf16(const void* dy, const uint8_t* argmax, void* dx, const void* relu_mask, const void* y_gate, int accumulate,
                                    int N, int H, int W, int C, int k, int stride, int pad, int Ho, int Wo, void* stream) {
    if (!dy || !argmax || !dx) return SSD_ERR_NULL;
    if (C % 8 != 0 || k <= 0 || k > 15 || stride <= 0 || pad < 0 || N <= 0) return SSD_ERR_BAD_SHAPE;
    if (y_gate && (accumulate || relu_mask)) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(dy) || !ssd_aligned16(dx) || ((uintptr_t)argmax & 7) || (relu_mask && !ssd_aligned16(relu_mask)) ||
        (y_gate && !ssd_aligned16(y_gate)))
        return SSD_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    if (y_gate && k == 2 && stride == 2 && pad == 0 && 2 * Ho >= H && 2 * Wo >= W) {
        hipLaunchKernelGGL(maxpool2_bwd_scatter_bf16_kernel, dim3(grid_for((size_t)N * Ho * Wo * (C / 8))), dim3(256), 0, st,
                           static_cast<const __bf16*>(dy), argmax, static_cast<const __bf16*>(y_gate), static_cast<__bf16*>(dx), N, H, W, C, Ho, Wo);
        SSD_CHECK_LAUNCH();
        return SSD_OK;
    }
    hipLaunchKernelGGL(maxpool_bwd_bf16_kernel, dim3(grid_for((size_t)N * H * W * (C / 8))), dim3(256), 0, st, static_cast<const __bf16*>(dy),
                       argmax, static_cast<__bf16*>(dx), static_cast<const __bf16*>(relu_mask), static_cast<const __bf16*>(y_gate), accumulate, N,
                       H, W, C, k, stride, pad, Ho, Wo);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

extern "C" int ssd_l2norm_fwd_bf16(const void* x, const float* gamma, void* y, int M, int C, void* stream) {
    if (!x || !gamma || !y) return SSD_ERR_NULL;
    if (C != 512 || M <= 0) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(x) || !ssd_aligned16(y)) return SSD_ERR_ALIGN;
    const int blocks = ssd_cdiv(M, 4) > 2048 ? 2048 : ssd_cdiv(M, 4);
    hipLaunchKernelGGL(l2norm_fwd_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, static_cast<const __bf16*>(x), gamma,
                       static_cast<__bf16*>(y), M);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

static int l2norm_bwd_bf16_blocks(int M) { return ssd_cdiv(M, 4) > 512 ? 512 : ssd_cdiv(M, 4); }
extern "C" size_t ssd_l2norm_bwd_bf16_workspace(int M, int C) { return (M <= 0 || C != 512) ? 0 : (size_t)l2norm_bwd_bf16_blocks(M) * 512 * sizeof(float); }
extern "C" int ssd_l2norm_bwd_bf16(const void* x, const float* gamma, const void* dy, void* dx, float* dgamma, int M, int C, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    if (!x || !gamma || !dy || !dx || !dgamma || !workspace) return SSD_ERR_NULL;
    if (C != 512 || M <= 0) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(x) || !ssd_aligned16(dy) || !ssd_aligned16(dx)) return SSD_ERR_ALIGN;
    const int blocks = l2norm_bwd_bf16_blocks(M);
    if (workspace_bytes < (size_t)blocks * 512 * sizeof(float)) return SSD_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* slab = static_cast<float*>(workspace);
    hipLaunchKernelGGL(l2norm_bwd_bf16_kernel, dim3(blocks), dim3(256), 0, st, static_cast<const __bf16*>(x), gamma,
                       static_cast<const __bf16*>(dy), static_cast<__bf16*>(dx), slab, M);
    SSD_CHECK_LAUNCH();
    hipLaunchKernelGGL(slab_sum512_kernel, dim3(32), dim3(256), 0, st, slab, dgamma, blocks);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

extern "C" int ssd_heads_gather_bf16(const float* dloc, const float* dconf, void* packed, int ld, int N, int HW, int A, int prior_off, int P,
                                     int ncls, void* stream) {
    if (!dloc || !dconf || !packed) return SSD_ERR_NULL;
    if (N <= 0 || HW <= 0 || A <= 0 || ld < A * (4 + ncls) || prior_off < 0 || prior_off + HW * A > P) return SSD_ERR_BAD_SHAPE;
    const size_t total = (size_t)N * HW * ld;
    hipLaunchKernelGGL(heads_gather_bf16_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dloc, dconf,
                       static_cast<__bf16*>(packed), ld, N, HW, A, prior_off, P, ncls);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

extern "C" int ssd_cast_f32_bf16(const float* x, void* y, size_t n, void* stream) {
    if (!x || !y) return SSD_ERR_NULL;
    if (n == 0 || n % 8 != 0) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(x) || !ssd_aligned16(y)) return SSD_ERR_ALIGN;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, x, static_cast<__bf16*>(y), n / 8);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}
extern "C" int ssd_cast_bf16_f32(const void* x, float* y, size_t n, void* stream) {
    if (!x || !y) return SSD_ERR_NULL;
    if (n == 0 || n % 8 != 0) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(x) || !ssd_aligned16(y)) return SSD_ERR_ALIGN;
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, static_cast<const __bf16*>(x), y, n / 8);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}
