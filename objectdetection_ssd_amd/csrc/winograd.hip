// Winograd F(2x2, 3x3) for the 3x3 / stride 1 / pad 1 convolutions with many channels (conv3 ... conv5), forward and dgrad:
// 16 multiplies per 2x2 output tile instead of 36 -- 2.25x fewer MFMAs for the same convolution (Lavin & Gray 2016).
//
//   W1 weights   U[xi][n][k] = (G g G^T)[xi]       per parameter update; dgrad uses the transposed, 180-degree-rotated filter
//   W2 input     V[xi][tile][k] = (B^T d B)[xi]    d = the 4x4 input patch of a 2x2 output tile (zero outside the image)
//   W3 multiply  M[xi] = V[xi] * U[xi]^T           sixteen GEMMs in one launch of the f32 MFMA igemm (blockIdx.z = xi)
//   W4 output    y = A^T m A (+ bias, ReLU | + previous dx, ReLU mask), the two rows / columns of each tile
// The transforms are exact in binary arithmetic up to the usual rounding of their additions (factors 1, 1/2, 1/4); the result
// differs from the direct sum at the 1e-6 level.  V and M live in a caller-provided workspace (16 planes each).
#include "common.h"

int ssd_internal_gemm_batched(const float* a, const float* w, float* out, int M, int K, int N, int n_rows, int nbatch,
                              size_t batch_w_elems, hipStream_t st);

namespace {

// U[xi][n][k]: rows n = output channels of the GEMM, k = its reduction channels.
// mode 0 (forward): n = co, k = ci, filter g[r][s] = w[co][ci][r][s]
// mode 1 (dgrad)  : n = ci, k = co (padded to K), filter g[r][s] = w[co][ci][2-r][2-s]
__global__ void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int Co, int Ci, int Nrows, int K, int mode) {
    const size_t total = (size_t)Nrows * K;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % K), n = (int)(i / K);
        const int co = mode == 0 ? n : k, ci = mode == 0 ? k : n;
        float g[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s)
                g[r][s] = (co < Co && ci < Ci) ? w[((size_t)co * Ci + ci) * 9 + (mode == 0 ? r * 3 + s : (2 - r) * 3 + (2 - s))] : 0.f;
        float t[4][3];                                       // G g
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            t[0][s] = g[0][s];
            t[1][s] = 0.5f * (g[0][s] + g[1][s] + g[2][s]);
            t[2][s] = 0.5f * (g[0][s] - g[1][s] + g[2][s]);
            t[3][s] = g[2][s];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {                        // (G g) G^T
            const float u0 = t[r][0], u1 = 0.5f * (t[r][0] + t[r][1] + t[r][2]), u2 = 0.5f * (t[r][0] - t[r][1] + t[r][2]), u3 = t[r][2];
            U[((size_t)(r * 4 + 0) * Nrows + n) * K + k] = u0;
            U[((size_t)(r * 4 + 1) * Nrows + n) * K + k] = u1;
            U[((size_t)(r * 4 + 2) * Nrows + n) * K + k] = u2;
            U[((size_t)(r * 4 + 3) * Nrows + n) * K + k] = u3;
        }
    }
}

// thread per (tile, 4 channels): 16 f32x4 loads, 16 f32x4 stores (one per plane), coalesced along the channels
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, float* __restrict__ V, int N, int H, int W, int C,
                                                         int TH, int TW) {
    const int C4 = C >> 2;
    const size_t tiles = (size_t)N * TH * TW, total = tiles * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        const size_t tile = i / C4;
        const int tw = (int)(tile % TW), th = (int)((tile / TW) % TH), n = (int)(tile / ((size_t)TW * TH));
        f32x4 d[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int ih = 2 * th - 1 + a;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int iw = 2 * tw - 1 + b;
                const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                d[a][b] = ok ? *reinterpret_cast<const f32x4*>(x + (((size_t)n * H + ih) * W + iw) * C + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        f32x4 t[4][4];                                       // B^T d
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            t[0][b] = d[0][b] - d[2][b];
            t[1][b] = d[1][b] + d[2][b];
            t[2][b] = d[2][b] - d[1][b];
            t[3][b] = d[1][b] - d[3][b];
        }
        float* dst = V + tile * C + c4 * 4;
        const size_t plane = tiles * C;
#pragma unroll
        for (int a = 0; a < 4; ++a) {                        // (B^T d) B
            *reinterpret_cast<f32x4*>(dst + (size_t)(a * 4 + 0) * plane) = t[a][0] - t[a][2];
            *reinterpret_cast<f32x4*>(dst + (size_t)(a * 4 + 1) * plane) = t[a][1] + t[a][2];
            *reinterpret_cast<f32x4*>(dst + (size_t)(a * 4 + 2) * plane) = t[a][2] - t[a][1];
            *reinterpret_cast<f32x4*>(dst + (size_t)(a * 4 + 3) * plane) = t[a][1] - t[a][3];
        }
    }
}

// thread per (tile, 4 channels): 16 plane loads -> 2x2 outputs.  out = [prev +] y (+ bias) -> ReLU -> mask
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ Mx, float* __restrict__ out, int N, int H, int W, int C,
                                                          int Cvalid, int ldo, int TH, int TW, const float* __restrict__ bias,
                                                          const float* __restrict__ mask, int relu, int accumulate) {
    const int C4 = C >> 2;
    const size_t tiles = (size_t)N * TH * TW, total = tiles * C4;
    const size_t plane = tiles * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        const size_t tile = i / C4;
        const int tw = (int)(tile % TW), th = (int)((tile / TW) % TH), n = (int)(tile / ((size_t)TW * TH));
        const float* src = Mx + tile * C + c4 * 4;
        f32x4 m[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) m[a][b] = *reinterpret_cast<const f32x4*>(src + (size_t)(a * 4 + b) * plane);
        f32x4 t[2][4];                                       // A^T m
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            t[0][b] = m[0][b] + m[1][b] + m[2][b];
            t[1][b] = m[1][b] - m[2][b] - m[3][b];
        }
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (bias != nullptr) {
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[e] = c4 * 4 + e < Cvalid ? bias[c4 * 4 + e] : 0.f;      // Cvalid may end inside the last vector
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int oh = 2 * th + a;
            if (oh >= H) continue;
            f32x4 y[2];
            y[0] = t[a][0] + t[a][1] + t[a][2];
            y[1] = t[a][1] - t[a][2] - t[a][3];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int ow = 2 * tw + b;
                if (ow >= W) continue;
                const size_t idx = (((size_t)n * H + oh) * W + ow) * ldo + c4 * 4;
                f32x4 v = y[b] + bv;
                if (accumulate) v += *reinterpret_cast<const f32x4*>(out + idx);
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];
                }
                if (mask != nullptr) {
                    const f32x4 mk = *reinterpret_cast<const f32x4*>(mask + idx);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = mk[e] > 0.f ? v[e] : 0.f;
                }
                *reinterpret_cast<f32x4*>(out + idx) = v;
            }
        }
    }
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
inline int grid_for(size_t total) { const size_t b = (total + 255) / 256; return (int)(b > 8192 ? 8192 : (b == 0 ? 1 : b)); }

// one F(2x2,3x3) convolution: in (N,H,W,Cin) -> out (N,H,W,ldo) first Cout channels
int wino_conv(const float* in, int Cin, const float* U, int U_rows, float* out, int ldo, int Cout, const float* bias, const float* mask,
              int relu, int accumulate, int N, int H, int W, void* ws, size_t ws_bytes, hipStream_t st) {
    const int TH = (H + 1) / 2, TW = (W + 1) / 2;
    const size_t tiles = (size_t)N * TH * TW;
    const int Cvalid = Cout;
    Cout = (Cout + 3) / 4 * 4;                 // the GEMM and the output transform work on whole 4-channel vectors: the filter rows
    if (tiles >= (1ull << 31) || Cin % 32 != 0 || Cout > ldo) return SSD_ERR_BAD_SHAPE;     // beyond Cvalid read as zero
    const size_t vb = align256(16 * tiles * Cin * 4), mb = align256(16 * tiles * Cout * 4);
    if (ws_bytes < vb + mb) return SSD_ERR_WORKSPACE;
    float* V = static_cast<float*>(ws);
    float* Mx = reinterpret_cast<float*>(static_cast<char*>(ws) + vb);
    hipLaunchKernelGGL(wino_input_kernel, dim3(grid_for(tiles * (Cin / 4))), dim3(256), 0, st, in, V, N, H, W, Cin, TH, TW);
    SSD_CHECK_LAUNCH();
    if (int e = ssd_internal_gemm_batched(V, U, Mx, (int)tiles, Cin, Cout, U_rows, 16, (size_t)U_rows * Cin, st)) return e;
    hipLaunchKernelGGL(wino_output_kernel, dim3(grid_for(tiles * (Cout / 4))), dim3(256), 0, st, Mx, out, N, H, W, Cout, Cvalid, ldo, TH, TW,
                       bias, mask, relu, accumulate);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

bool wino_geom_ok(const ssd_conv_geom* g) {
    return g && g->R == 3 && g->S == 3 && g->stride == 1 && g->dil == 1 && g->pad == 1 && g->Ho == g->H && g->Wo == g->W && g->N > 0 &&
           g->H > 0 && g->W > 0 && g->Ci > 0 && g->Co > 0;
}

}  // namespace

// U_fwd: [16][Co][Ci]; U_bwd: [16][Ci][Co_pad] (either may be NULL)
extern "C" int ssd_wino_weights(const float* w_oihw, float* U_fwd, float* U_bwd, int Co, int Ci, int Co_pad, void* stream) {
    if (!w_oihw || (!U_fwd && !U_bwd)) return SSD_ERR_NULL;
    if (Co <= 0 || Ci <= 0 || Co_pad < Co) return SSD_ERR_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (U_fwd) {
        hipLaunchKernelGGL(wino_weight_kernel, dim3(grid_for((size_t)Co * Ci)), dim3(256), 0, st, w_oihw, U_fwd, Co, Ci, Co, Ci, 0);
        SSD_CHECK_LAUNCH();
    }
    if (U_bwd) {
        hipLaunchKernelGGL(wino_weight_kernel, dim3(grid_for((size_t)Ci * Co_pad)), dim3(256), 0, st, w_oihw, U_bwd, Co, Ci, Ci, Co_pad, 1);
        SSD_CHECK_LAUNCH();
    }
    return SSD_OK;
}

extern "C" size_t ssd_conv3x3_wino_workspace(const ssd_conv_geom* g, int direction) {
    if (!wino_geom_ok(g)) return 0;
    const size_t tiles = (size_t)g->N * ((g->H + 1) / 2) * ((g->W + 1) / 2);
    const int co_pad = (g->Co + 31) / 32 * 32;
    const size_t cin = direction == 0 ? g->Ci : co_pad, cout = direction == 0 ? (size_t)(g->Co + 3) / 4 * 4 : (size_t)(g->Ci + 3) / 4 * 4;
    return align256(16 * tiles * cin * 4) + align256(16 * tiles * cout * 4);
}

extern "C" int ssd_conv3x3_wino_fwd(const float* x, const float* U_fwd, const float* bias, float* y, int ldy, const ssd_conv_geom* g,
                                    int relu, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !U_fwd || !y || !workspace) return SSD_ERR_NULL;
    if (!wino_geom_ok(g) || g->Ci % 32 != 0 || ldy < (g->Co + 3) / 4 * 4) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(x) || !ssd_aligned16(y) || !ssd_aligned16(workspace) || !ssd_aligned16(U_fwd) || (bias && !ssd_aligned16(bias)) ||
        ldy % 4 != 0)
        return SSD_ERR_ALIGN;
    return wino_conv(x, g->Ci, U_fwd, g->Co, y, ldy, g->Co, bias, nullptr, relu, 0, g->N, g->H, g->W, workspace, workspace_bytes,
                     (hipStream_t)stream);
}

extern "C" int ssd_conv3x3_wino_dgrad(const float* dy, int ldy, const float* U_bwd, int Co_pad, float* dx, const float* relu_mask,
                                      int accumulate, const ssd_conv_geom* g, void* workspace, size_t workspace_bytes, void* stream) {
    if (!dy || !U_bwd || !dx || !workspace) return SSD_ERR_NULL;
    if (!wino_geom_ok(g) || Co_pad % 32 != 0 || Co_pad < g->Co || ldy != Co_pad || g->Ci % 4 != 0) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(dy) || !ssd_aligned16(dx) || !ssd_aligned16(workspace) || !ssd_aligned16(U_bwd) ||
        (relu_mask && !ssd_aligned16(relu_mask)))
        return SSD_ERR_ALIGN;
    return wino_conv(dy, Co_pad, U_bwd, g->Ci, dx, g->Ci, g->Ci, nullptr, relu_mask, 0, accumulate, g->N, g->H, g->W, workspace,
                     workspace_bytes, (hipStream_t)stream);
}
