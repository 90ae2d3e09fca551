// HBM-bound NHWC kernels around the convolutions: weight re-layout, conv1_1 im2col,
// max pooling, conv4_3 L2 normalisation, head scatter/gather, fused SGD.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------
// weight layouts
// ---------------------------------------------------------------------------------------
__global__ void oihw_to_ohwi_kernel(const float* __restrict__ w, float* __restrict__ o, int Co, int Ci, int T, int Co_pad) {
    const size_t total = (size_t)Co_pad * T * Ci;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Ci);
        const size_t rest = i / Ci;
        const int t = (int)(rest % T), co = (int)(rest / T);
        o[i] = co < Co ? w[((size_t)co * Ci + ci) * T + t] : 0.f;
    }
}
__global__ void oihw_to_ihwo_kernel(const float* __restrict__ w, float* __restrict__ o, int Co, int Ci, int T, int Co_pad) {
    const size_t total = (size_t)Ci * T * Co_pad;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % Co_pad);
        const size_t rest = i / Co_pad;
        const int t = (int)(rest % T), ci = (int)(rest / T);
        o[i] = co < Co ? w[((size_t)co * Ci + ci) * T + t] : 0.f;
    }
}

inline int grid_for(size_t total, int block = 256, int cap = 4096) {
    size_t b = (total + block - 1) / block;
    return (int)(b > (size_t)cap ? cap : (b == 0 ? 1 : b));
}

// ---------------------------------------------------------------------------------------
// conv1_1 (Model.py:136 features[0], Ci = 3): im2col so that the MFMA kernels can take it
// ---------------------------------------------------------------------------------------
// im2col of the 3-channel NCHW input for conv1_1: out[pix][k], k = (r*3+s)*3 + c for k < 27, zero for 27..31.
// 8 threads per pixel, each writes one 16-byte chunk (4 k values) -> 128 contiguous bytes per pixel.
__global__ __launch_bounds__(256) void im2col_first_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int H, int W) {
    const size_t HW = (size_t)H * W;
    const size_t total = (size_t)N * HW * 8;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
        const size_t pix = g >> 3;
        const int q = (int)(g & 7);
        const int n = (int)(pix / HW);
        const int rem = (int)(pix - (size_t)n * HW);
        const int oh = rem / W, ow = rem - oh * W;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = q * 4 + e;
            if (k < 27) {
                const int c = k % 3, t = k / 3;
                const int ih = oh + t / 3 - 1, iw = ow + t % 3 - 1;
                if (ih >= 0 && ih < H && iw >= 0 && iw < W) v[e] = x[((size_t)n * 3 + c) * HW + (size_t)ih * W + iw];
            }
        }
        *reinterpret_cast<f32x4*>(out + g * 4) = v;
    }
}

// General im2col of a 3-channel NCHW batch (ResNet-34 stem, Model.py:26 seq1[0]: Conv2d(3,64,7,stride 2,pad 3)):
// out[pix][k], k = (r*S+s)*3 + c for k < R*S*3, zero up to Kpad.  Kpad/4 threads per output pixel, 16 bytes each.
__global__ __launch_bounds__(256) void im2col_nchw3_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int H, int W,
                                                           int R, int S, int stride, int pad, int Ho, int Wo, int Kpad) {
    const int q_per = Kpad >> 2, K = R * S * 3;
    const size_t HW = (size_t)H * W;
    const size_t total = (size_t)N * Ho * Wo * q_per;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
        const size_t pix = g / q_per;
        const int q = (int)(g - pix * q_per);
        const int n = (int)(pix / ((size_t)Ho * Wo));
        const int rem = (int)(pix - (size_t)n * Ho * Wo);
        const int oh = rem / Wo, ow = rem - oh * Wo;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = q * 4 + e;
            if (k < K) {
                const int c = k % 3, t = k / 3;
                const int ih = oh * stride + t / S - pad, iw = ow * stride + t % S - pad;
                if (ih >= 0 && ih < H && iw >= 0 && iw < W) v[e] = x[((size_t)n * 3 + c) * HW + (size_t)ih * W + iw];
            }
        }
        *reinterpret_cast<f32x4*>(out + g * 4) = v;
    }
}

// y = x * scale[c] + shift[c] (optionally ReLU) over [M][C] rows: eval-mode BatchNorm that follows a ReLU
// (Model.py:56-62 Conv -> ReLU -> BN) and so cannot be folded into the convolution before it.
__global__ __launch_bounds__(256) void channel_affine_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, float* __restrict__ y, size_t total4,
                                                             int C4, int relu) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        const f32x4 a = reinterpret_cast<const f32x4*>(scale)[c], b = reinterpret_cast<const f32x4*>(shift)[c];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = v[e] * a[e] + b[e];
            if (relu) o[e] = o[e] < 0.f ? 0.f : o[e];
        }
        reinterpret_cast<f32x4*>(y)[i] = o;
    }
}

// Diagnostic: shader-clock counter and the constant 100 MHz wall clock, one slot per XCC (each XCD has its own counter).
// Two probes around a region give the average shader clock the chip held there (bench.py reports it next to the roofline:
// under sustained f32-MFMA load on non-constant data this chip lowers its clock well below 2.4 GHz).
__global__ void clock_probe_kernel(unsigned long long* __restrict__ out) {
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;       // HW_REG_XCC_ID[3:0]
        out[xcc * 2 + 0] = __builtin_readcyclecounter();
        out[xcc * 2 + 1] = wall_clock64();
    }
}

// out[i] = sum_k slab[k][i]: 16 threads per column group stride over the slabs (16 loads in flight per column instead of one
// dependent chain of nslab loads), then a fixed-order LDS tree -- reproducible.  blockDim = (16 columns, 16 slab lanes).
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ slab, float* __restrict__ out, int n, int nslab) {
    __shared__ float part[16][17];
    const int c = threadIdx.x & 15, l = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + c;
    float s = 0.f;
    if (i < n)
        for (int k = l; k < nslab; k += 16) s += slab[(size_t)k * n + i];
    part[l][c] = s;
    __syncthreads();
    if (l == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += part[k][c];
        out[i] = t;
    }
}

// ---------------------------------------------------------------------------------------
// max pooling, NHWC, 4 channels per thread
// ---------------------------------------------------------------------------------------
__global__ void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ am, int N,
                                   int H, int W, int C, int k, int stride, int pad, int Ho, int Wo) {
    const int C4 = C >> 2;
    const size_t total = (size_t)N * Ho * Wo * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        size_t rest = i / C4;
        const int ow = (int)(rest % Wo);
        rest /= Wo;
        const int oh = (int)(rest % Ho), n = (int)(rest / Ho);
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {0, 0, 0, 0};
        bool first = true;
        for (int r = 0; r < k; ++r) {
            const int ih = oh * stride - pad + r;
            if (ih < 0 || ih >= H) continue;
            for (int s = 0; s < k; ++s) {
                const int iw = ow * stride - pad + s;
                if (iw < 0 || iw >= W) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((size_t)n * H + ih) * W + iw) * C + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // torch: take v if (v > best) or isnan(v); the first valid element seeds the max
                    if (first || v[e] > best[e] || v[e] != v[e]) {
                        best[e] = v[e];
                        bi[e] = r * k + s;
                    }
                }
                first = false;
            }
        }
        *reinterpret_cast<f32x4*>(y + i * 4) = best;
        if (am) {
            uint32_t packed = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
            *reinterpret_cast<uint32_t*>(am + i * 4) = packed;
        }
    }
}

// gather form: every input element sums dy over the windows whose argmax it is
// y_gate (optional): the pooled output.  An input that is a window's argmax equals that window's output, so gating each
// window's dy by y > 0 is the ReLU mask x > 0 of the pool's input without reading the 4x larger x.
__global__ void maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ am, float* __restrict__ dx,
                                   const float* __restrict__ mask, const float* __restrict__ y_gate, int accumulate, int N, int H,
                                   int W, int C, int k, int stride, int pad, int Ho, int Wo) {
    const int C4 = C >> 2;
    const size_t total = (size_t)N * H * W * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        size_t rest = i / C4;
        const int iw = (int)(rest % W);
        rest /= W;
        const int ih = (int)(rest % H), n = (int)(rest / H);
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
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
                const size_t o = (((size_t)n * Ho + oh) * Wo + ow) * C + c4 * 4;
                const uint32_t a = *reinterpret_cast<const uint32_t*>(am + o);
                f32x4 d = *reinterpret_cast<const f32x4*>(dy + o);
                if (y_gate != nullptr) {
                    const f32x4 yv = *reinterpret_cast<const f32x4*>(y_gate + o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[e] = yv[e] > 0.f ? d[e] : 0.f;
                }
                const uint32_t me = (uint32_t)(r * k + s);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (((a >> (8 * e)) & 0xffu) == me) g[e] += d[e];
            }
        }
        f32x4* dst = reinterpret_cast<f32x4*>(dx + i * 4);
        if (accumulate) g += *dst;
        if (mask) {
            const f32x4 m = *reinterpret_cast<const f32x4*>(mask + i * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = m[e] > 0.f ? g[e] : 0.f;
        }
        *dst = g;
    }
}

// scatter form for the 2x2 / stride-2 / no-pad pool whose windows tile the whole input (the gated case: sole consumer, no accumulate):
// one thread per window and channel quad reads dy, the argmax codes and the gate once and writes the window's (up to) four inputs --
// the gather form above reads each of them from all four inputs of the window.
__global__ __launch_bounds__(256) void maxpool2_bwd_scatter_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ am,
                                                                   const float* __restrict__ y_gate, float* __restrict__ dx, int N, int H, int W,
                                                                   int C, int Ho, int Wo) {
    const int C4 = C >> 2;
    const size_t total = (size_t)N * Ho * Wo * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        size_t rest = i / C4;
        const int ow = (int)(rest % Wo);
        rest /= Wo;
        const int oh = (int)(rest % Ho), n = (int)(rest / Ho);
        f32x4 d = *reinterpret_cast<const f32x4*>(dy + i * 4);
        const f32x4 yv = *reinterpret_cast<const f32x4*>(y_gate + i * 4);
        const uint32_t a = *reinterpret_cast<const uint32_t*>(am + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = yv[e] > 0.f ? d[e] : 0.f;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int ih = 2 * oh + r;
            if (ih >= H) continue;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int iw = 2 * ow + q;
                if (iw >= W) continue;
                f32x4 g;
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = ((a >> (8 * e)) & 0xffu) == (uint32_t)(r * 2 + q) ? d[e] : 0.f;
                *reinterpret_cast<f32x4*>(dx + ((((size_t)n * H + ih) * W + iw) * C4 + c4) * 4) = g;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// L2 norm over channels: one wave per pixel, C = 512 -> 8 floats per lane
// ---------------------------------------------------------------------------------------
template <int VPL>   // float4 vectors per lane: C = 64*4*VPL
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         float* __restrict__ y, int M) {
    const int lane = threadIdx.x & 63;
    const int C = 256 * VPL;
    for (int m = blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += gridDim.x * 4) {
        f32x4 v[VPL];
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            v[j] = *reinterpret_cast<const f32x4*>(x + (size_t)m * C + (j * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) ss += v[j][e] * v[j][e];
        }
        ss = wave_sum(ss);
        const float nrm = sqrtf(ss);               // no epsilon: Model.py:207
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + (j * 64 + lane) * 4);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = v[j][e] / nrm * g[e];
            *reinterpret_cast<f32x4*>(y + (size_t)m * C + (j * 64 + lane) * 4) = o;
        }
    }
}

// u = x/|x|, gdy = gamma*dy:  dx = (gdy - u*(u.gdy))/|x| ;  dgamma_c = sum_m dy_c*u_c (per-block partials)
template <int VPL>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ dy, float* __restrict__ dx,
                                                         float* __restrict__ dg_slab, int M) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C = 256 * VPL;
    __shared__ float red[4][256 * VPL];
    f32x4 dg[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) dg[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int m = blockIdx.x * 4 + wave; m < M; m += gridDim.x * 4) {
        f32x4 v[VPL], d[VPL];
        float ss = 0.f, dot = 0.f;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const size_t o = (size_t)m * C + (j * 64 + lane) * 4;
            v[j] = *reinterpret_cast<const f32x4*>(x + o);
            d[j] = *reinterpret_cast<const f32x4*>(dy + o);
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + (j * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ss += v[j][e] * v[j][e];
                dot += v[j][e] * d[j][e] * g[e];
            }
        }
        ss = wave_sum(ss);
        dot = wave_sum(dot);
        const float inv = 1.f / sqrtf(ss);
        const float coef = dot * inv * inv * inv;          // (x.gdy)/|x|^3
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + (j * 64 + lane) * 4);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = g[e] * d[j][e] * inv - v[j][e] * coef;
                dg[j][e] += d[j][e] * v[j][e] * inv;
            }
            *reinterpret_cast<f32x4*>(dx + (size_t)m * C + (j * 64 + lane) * 4) = o;
        }
    }
#pragma unroll
    for (int j = 0; j < VPL; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave][(j * 64 + lane) * 4 + e] = dg[j][e];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256)
        dg_slab[(size_t)blockIdx.x * C + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}

// ---------------------------------------------------------------------------------------
// heads: packed [N*HW][ld] <-> loc (N,P,4) / conf (N,P,ncls)
// ---------------------------------------------------------------------------------------
__global__ void heads_scatter_kernel(const float* __restrict__ packed, int ld, float* __restrict__ loc,
                                     float* __restrict__ conf, int N, int HW, int A, int prior_off, int P, int ncls) {
    const int cw = A * (4 + ncls);
    const size_t total = (size_t)N * HW * cw;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cw);
        const size_t m = i / cw;
        const int n = (int)(m / HW), pix = (int)(m % HW);
        const float v = packed[m * ld + c];
        const size_t pbase = (size_t)n * P + prior_off + (size_t)pix * A;
        if (c < 4 * A) loc[pbase * 4 + c] = v;
        else conf[pbase * ncls + (c - 4 * A)] = v;
    }
}
__global__ void heads_gather_kernel(const float* __restrict__ dloc, const float* __restrict__ dconf,
                                    float* __restrict__ packed, int ld, int N, int HW, int A, int prior_off, int P,
                                    int ncls) {
    const int cw = A * (4 + ncls);
    const size_t total = (size_t)N * HW * ld;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % ld);
        const size_t m = i / ld;
        const int n = (int)(m / HW), pix = (int)(m % HW);
        const size_t pbase = (size_t)n * P + prior_off + (size_t)pix * A;
        float v = 0.f;                                  // pad columns cw..ld-1 are written as zero
        if (c < 4 * A) v = dloc[pbase * 4 + c];
        else if (c < cw) v = dconf[pbase * ncls + (c - 4 * A)];
        packed[i] = v;
    }
}

// ---------------------------------------------------------------------------------------
// SGD with momentum and weight decay (torch.optim.SGD semantics, train.py:53-55):
//   g = grad*scale + wd*p ; buf = first ? g : mom*buf + g ; p -= lr*buf
// The rounding sequence is the one torch's foreach path performs with its separate launches -- add(grad, p, alpha=wd) and
// add_(p, buf, alpha=-lr) are single fused multiply-adds, mul_(buf, mom) and add_(buf, g) round separately -- so a run through
// this kernel and a run through torch.optim.SGD stay bit-identical.
// ---------------------------------------------------------------------------------------
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, size_t n, float lr,
                           float mom, float wd, const float* __restrict__ scale, int first) {
#pragma clang fp contract(off)
    const float sc = scale ? *scale : 1.f;
    const float neg_lr = -lr;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float w = p[i];
        const float gs = scale ? g[i] * sc : g[i];
        const float d = wd != 0.f ? __builtin_fmaf(wd, w, gs) : gs;
        float b = d;
        if (!first) {
            const float mb = mom * buf[i];
            b = mb + d;
        }
        buf[i] = b;
        p[i] = __builtin_fmaf(neg_lr, b, w);
    }
}

}  // namespace

extern "C" int ssd_abi_version(void) { return SSD_ABI_VERSION; }

extern "C" const char* ssd_status_string(int s) {
    switch (s) {
        case SSD_OK: return "ok";
        case SSD_ERR_BAD_SHAPE: return "unsupported or inconsistent shape";
        case SSD_ERR_WORKSPACE: return "workspace too small";
        case SSD_ERR_NULL: return "required pointer is NULL";
        case SSD_ERR_LAUNCH: return "kernel launch failed (hipGetLastError)";
        case SSD_ERR_ALIGN: return "pointer or leading dimension not 16-byte aligned";
        default: return "unknown status";
    }
}

extern "C" int ssd_weight_oihw_to_ohwi(const float* w, float* o, int Co, int Ci, int R, int S, int Co_pad, void* stream) {
    if (!w || !o) return SSD_ERR_NULL;
    if (Co <= 0 || Ci <= 0 || R <= 0 || S <= 0 || Co_pad < Co) return SSD_ERR_BAD_SHAPE;
    const size_t total = (size_t)Co_pad * R * S * Ci;
    hipLaunchKernelGGL(oihw_to_ohwi_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, w, o, Co, Ci, R * S, Co_pad);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}
extern "C" int ssd_weight_oihw_to_ihwo(const float* w, float* o, int Co, int Ci, int R, int S, int Co_pad, void* stream) {
    if (!w || !o) return SSD_ERR_NULL;
    if (Co <= 0 || Ci <= 0 || R <= 0 || S <= 0 || Co_pad < Co) return SSD_ERR_BAD_SHAPE;
    const size_t total = (size_t)Co_pad * R * S * Ci;
    hipLaunchKernelGGL(oihw_to_ihwo_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, w, o, Co, Ci, R * S, Co_pad);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

extern "C" int ssd_im2col_first(const float* x_nchw, float* out, int N, int H, int W, void* stream) {
    if (!x_nchw || !out) return SSD_ERR_NULL;
    if (N <= 0 || H <= 0 || W <= 0) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(out)) return SSD_ERR_ALIGN;
    const size_t total = (size_t)N * H * W * 8;
    hipLaunchKernelGGL(im2col_first_kernel, dim3(grid_for(total, 256, 8192)), dim3(256), 0, (hipStream_t)stream, x_nchw, out, N, H, W);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

extern "C" int ssd_clock_probe(uint64_t* out32, void* stream) {
    if (!out32) return SSD_ERR_NULL;
    hipLaunchKernelGGL(clock_probe_kernel, dim3(256), dim3(64), 0, (hipStream_t)stream, reinterpret_cast<unsigned long long*>(out32));
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

// Host-only diagnostic: how many nodes -- and how many of them kernel launches -- a captured HIP graph holds (what one replay of
// ddp.GraphedTrainStep stands for; bench.py prints it).
extern "C" int ssd_graph_node_counts(void* graph, int* kernel_nodes, int* total_nodes) {
    if (!graph || !kernel_nodes || !total_nodes) return SSD_ERR_NULL;
    size_t n = 0;
    if (hipGraphGetNodes((hipGraph_t)graph, nullptr, &n) != hipSuccess) return SSD_ERR_LAUNCH;
    hipGraphNode_t* nodes = n ? new hipGraphNode_t[n] : nullptr;
    int k = 0;
    if (n && hipGraphGetNodes((hipGraph_t)graph, nodes, &n) != hipSuccess) {
        delete[] nodes;
        return SSD_ERR_LAUNCH;
    }
    for (size_t i = 0; i < n; ++i) {
        hipGraphNodeType t;
        if (hipGraphNodeGetType(nodes[i], &t) == hipSuccess && t == hipGraphNodeTypeKernel) ++k;
    }
    delete[] nodes;
    *kernel_nodes = k;
    *total_nodes = (int)n;
    return SSD_OK;
}

extern "C" int ssd_im2col_nchw3(const float* x_nchw, float* out, int N, int H, int W, int R, int S, int stride, int pad,
                                int Ho, int Wo, int Kpad, void* stream) {
    if (!x_nchw || !out) return SSD_ERR_NULL;
    if (N <= 0 || H <= 0 || W <= 0 || R <= 0 || S <= 0 || stride <= 0 || pad < 0 || Kpad % 4 != 0 || Kpad < R * S * 3)
        return SSD_ERR_BAD_SHAPE;
    if (Ho != (H + 2 * pad - R) / stride + 1 || Wo != (W + 2 * pad - S) / stride + 1 || Ho <= 0 || Wo <= 0) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(out)) return SSD_ERR_ALIGN;
    const size_t total = (size_t)N * Ho * Wo * (Kpad / 4);
    hipLaunchKernelGGL(im2col_nchw3_kernel, dim3(grid_for(total, 256, 8192)), dim3(256), 0, (hipStream_t)stream, x_nchw, out, N, H, W,
                       R, S, stride, pad, Ho, Wo, Kpad);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

extern "C" int ssd_channel_affine(const float* x, const float* scale, const float* shift, float* y, size_t M, int C, int relu,
                                  void* stream) {
    if (!x || !scale || !shift || !y) return SSD_ERR_NULL;
    if (M == 0 || C <= 0 || C % 4 != 0) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(x) || !ssd_aligned16(y) || !ssd_aligned16(scale) || !ssd_aligned16(shift)) return SSD_ERR_ALIGN;
    const size_t total4 = M * (size_t)(C / 4);
    hipLaunchKernelGGL(channel_affine_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, x, scale, shift, y, total4,
                       C / 4, relu);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

extern "C" int ssd_maxpool_fwd(const float* x, float* y, uint8_t* argmax, int N, int H, int W, int C, int k, int stride,
                               int pad, int Ho, int Wo, void* stream) {
    if (!x || !y) return SSD_ERR_NULL;
    if (C % 4 != 0 || k <= 0 || k > 15 || stride <= 0 || pad < 0 || 2 * pad > k || Ho <= 0 || Wo <= 0) return SSD_ERR_BAD_SHAPE;
    // every window must contain at least one valid element (torch guarantees this for its own Ho/Wo)
    if ((Ho - 1) * stride - pad >= H || (Wo - 1) * stride - pad >= W) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(x) || !ssd_aligned16(y)) return SSD_ERR_ALIGN;
    const size_t total = (size_t)N * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, argmax, N, H, W, C, k, stride, pad, Ho, Wo);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}
extern "C" int ssd_maxpool_bwd(const float* dy, const uint8_t* argmax, float* dx, const float* relu_mask, int accumulate,
                               int N, int H, int W, int C, int k, int stride, int pad, int Ho, int Wo, void* stream) {
    if (!dy || !argmax || !dx) return SSD_ERR_NULL;
    if (C % 4 != 0 || k <= 0 || k > 15 || stride <= 0 || pad < 0) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(dy) || !ssd_aligned16(dx)) return SSD_ERR_ALIGN;
    const size_t total = (size_t)N * H * W * (C / 4);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dy, argmax, dx, relu_mask,
                       static_cast<const float*>(nullptr), accumulate, N, H, W, C, k, stride, pad, Ho, Wo);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}
extern "C" int ssd_maxpool_bwd_gated(const float* dy, const uint8_t* argmax, const float* y, float* dx, int N, int H, int W, int C,
                                     int k, int stride, int pad, int Ho, int Wo, void* stream) {
    if (!dy || !argmax || !y || !dx) return SSD_ERR_NULL;
    if (C % 4 != 0 || k <= 0 || k > 15 || stride <= 0 || pad < 0) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(dy) || !ssd_aligned16(dx) || !ssd_aligned16(y)) return SSD_ERR_ALIGN;
    if (k == 2 && stride == 2 && pad == 0 && 2 * Ho >= H && 2 * Wo >= W && ((uintptr_t)argmax & 3) == 0) {
        hipLaunchKernelGGL(maxpool2_bwd_scatter_kernel, dim3(grid_for((size_t)N * Ho * Wo * (C / 4))), dim3(256), 0, (hipStream_t)stream, dy,
                           argmax, y, dx, N, H, W, C, Ho, Wo);
        SSD_CHECK_LAUNCH();
        return SSD_OK;
    }
    const size_t total = (size_t)N * H * W * (C / 4);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dy, argmax, dx,
                       static_cast<const float*>(nullptr), y, 0, N, H, W, C, k, stride, pad, Ho, Wo);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

extern "C" int ssd_l2norm_fwd(const float* x, const float* gamma, float* y, int M, int C, void* stream) {
    if (!x || !gamma || !y) return SSD_ERR_NULL;
    if (C != 512 || M <= 0) return SSD_ERR_BAD_SHAPE;
    const int blocks = ssd_cdiv(M, 4) > 2048 ? 2048 : ssd_cdiv(M, 4);
    hipLaunchKernelGGL(l2norm_fwd_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, gamma, y, M);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}
static int l2norm_bwd_blocks(int M) { const int b = ssd_cdiv(M, 4); return b > 512 ? 512 : b; }
extern "C" size_t ssd_l2norm_bwd_workspace(int M, int C) { return (size_t)l2norm_bwd_blocks(M) * C * sizeof(float) + 256; }
extern "C" int ssd_l2norm_bwd(const float* x, const float* gamma, const float* dy, float* dx, float* dgamma, int M, int C,
                              void* ws, size_t ws_bytes, void* stream) {
    if (!x || !gamma || !dy || !dx || !dgamma || !ws) return SSD_ERR_NULL;
    if (C != 512 || M <= 0) return SSD_ERR_BAD_SHAPE;
    if (ws_bytes < ssd_l2norm_bwd_workspace(M, C)) return SSD_ERR_WORKSPACE;
    const int blocks = l2norm_bwd_blocks(M);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(l2norm_bwd_kernel<2>, dim3(blocks), dim3(256), 0, st, x, gamma, dy, dx, reinterpret_cast<float*>(ws), M);
    SSD_CHECK_LAUNCH();
    hipLaunchKernelGGL(slab_sum_kernel, dim3(ssd_cdiv(C, 16)), dim3(256), 0, st, reinterpret_cast<const float*>(ws), dgamma, C, blocks);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

extern "C" int ssd_heads_scatter(const float* packed, int ld, float* loc, float* conf, int N, int HW, int A, int prior_off,
                                 int P, int ncls, void* stream) {
    if (!packed || !loc || !conf) return SSD_ERR_NULL;
    if (ld < A * (4 + ncls) || prior_off < 0 || prior_off + HW * A > P) return SSD_ERR_BAD_SHAPE;
    const size_t total = (size_t)N * HW * A * (4 + ncls);
    hipLaunchKernelGGL(heads_scatter_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, packed, ld, loc, conf, N, HW, A, prior_off, P, ncls);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}
extern "C" int ssd_heads_gather(const float* dloc, const float* dconf, float* packed, int ld, int N, int HW, int A,
                                int prior_off, int P, int ncls, void* stream) {
    if (!packed || !dloc || !dconf) return SSD_ERR_NULL;
    if (ld < A * (4 + ncls) || prior_off < 0 || prior_off + HW * A > P) return SSD_ERR_BAD_SHAPE;
    const size_t total = (size_t)N * HW * ld;
    hipLaunchKernelGGL(heads_gather_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dloc, dconf, packed, ld, N, HW, A, prior_off, P, ncls);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

extern "C" int ssd_sgd_momentum(float* param, const float* grad, float* buf, size_t n, float lr, float momentum,
                                float weight_decay, const float* grad_scale_dev, int first_step, void* stream) {
    if (!param || !grad || !buf) return SSD_ERR_NULL;
    if (n == 0) return SSD_OK;
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, (hipStream_t)stream, param, grad, buf, n, lr, momentum, weight_decay, grad_scale_dev, first_step);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}
