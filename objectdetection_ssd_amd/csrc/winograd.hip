// Winograd F(2x2, 3x3) for the 3x3 / stride 1 / pad 1 convolutions with many channels (conv3 ... conv5), forward and dgrad:
// 16 multiplies per 2x2 output tile instead of 36 -- 2.25x fewer MFMAs for the same convolution (Lavin & Gray 2016).
//
//   W1 weights   U[xi][n][k] = (G g G^T)[xi]       per parameter update; dgrad uses the transposed, 180-degree-rotated filter
//   W2 input     V[xi][tile][k] = (B^T d B)[xi]    d = the 4x4 input patch of a 2x2 output tile (zero outside the image)
//   W3 multiply  M[xi] = V[xi] * U[xi]^T           sixteen GEMMs in one launch of the f32 MFMA igemm (blockIdx.z = xi)
//   W4 output    y = A^T m A (+ bias, ReLU | + previous dx, ReLU mask), the two rows / columns of each tile
// The transforms are exact in binary arithmetic up to the usual rounding of their additions (factors 1, 1/2, 1/4); the result
// differs from the direct sum at the 1e-6 level.  V and M live in a caller-provided workspace (16 planes each).
#include <cstdlib>
#include "common.h"

int ssd_internal_gemm_batched(const float* a, const float* w, float* out, int M, int K, int N, int n_rows, int nbatch,
                              size_t batch_a_elems, size_t batch_w_elems, int ksplit, hipStream_t st);
int ssd_internal_gemm_batched_x3(const float* a, const void* w3, float* out, int M, int K, int N, int n_rows, int nbatch, size_t batch_a_elems,
                                 hipStream_t st);
int ssd_internal_gemm_tn_x3(const float* a, const float* c, float* out, int M, int N, int K, int lda, int ldc, int nbatch, int ksplit,
                            int ksteps_per_split, size_t batch_a, size_t batch_c, hipStream_t st);
int ssd_internal_wino4_gemm_out(const float* V, const float* U, int tiles, int K, int Nrows, int Nout, float* out, int ldo, int Cvalid,
                                const float* bias, const float* mask, const unsigned long long* mask_bits, int relu, int accumulate, int H,
                                int W, int TH, int TW, float* yp, uint8_t* am, int Ho, int Wo, hipStream_t st);

int ssd_internal_wino4_full(const float* x, const float* U, int tiles, int K, int Nrows, int Nout, float* out, int ldo, int Cvalid,
                            const float* bias, const float* mask, const unsigned long long* mask_bits, int relu, int accumulate, int H, int W,
                            int TH, int TW, float* yp, uint8_t* am, int Ho, int Wo, float* V_keep, unsigned long long* bits_out, hipStream_t st);

namespace {

// Dilated 3x3 convolutions (dilation D = padding, stride 1: fc6, Model.py:149) are D x D independent 3x3 / pad-1 convolutions, one per
// sub-lattice x[D*i + a][D*j + b] of the map: y[D*i + a][D*j + b] = sum_rs w[r][s] x[D*(i + r - 1) + a][D*(j + s - 1) + b].  The F(4x4)
// kernels therefore only need another map from (tile row, row inside the patch) to the image row: tile rows are numbered through the
// sub-lattices of offsets 0 .. D-1 one after the other (offset a has ceil(ceil((H - a) / D) / 4) of them, first[a] = where they start).
// D = 1 is the plain convolution: one lattice, image row = 4 * tile - 1 + patch row.  GEMMs, filters and plane layouts do not change.
struct Lat { int D; int h0[4], w0[4]; };
// tile row / column t of a map whose lattices start at first[]: offset of its sub-lattice, lattice coordinate of its first OUTPUT row
__device__ __forceinline__ void lat_tile(const int (&first)[4], int D, int t, int& off, int& base) {
    off = 0;
#pragma unroll
    for (int a = 1; a < 4; ++a)
        if (a < D && t >= first[a]) off = a;
    base = 4 * (t - first[off]);
}

// U[xi][n][k]: rows n = output channels of the GEMM, k = its reduction channels.
// mode 0 (forward): n = co, k = ci, filter g[r][s] = w[co][ci][r][s]
// mode 1 (dgrad)  : n = ci, k = co (padded to K), filter g[r][s] = w[co][ci][2-r][2-s]
__global__ void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int Co, int Ci, int Nrows, int K, int mode) {
    const size_t total = (size_t)Nrows * K;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % K), n = (int)(i / K);
        const int co = mode == 0 ? n : k, ci = mode == 0 ? k : n;      // mode 2: the adjoint form -- rows ci, K = co like mode 1, taps NOT rotated
        float g[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s)
                g[r][s] = (co < Co && ci < Ci) ? w[((size_t)co * Ci + ci) * 9 + (mode != 1 ? r * 3 + s : (2 - r) * 3 + (2 - s))] : 0.f;
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

// ---- weight gradient:  dg = G^T [ sum over tiles of (A dy A^T) (x) (B^T d B) ] G  (the adjoint of the forward) ---------------------
// Transposed transforms: plane xi of channel c is ONE row [tile] (Tpad columns, zero beyond the last tile), so that the sum over
// tiles is the contiguous K dimension of the f32 igemm.  Thread = (tile, 4 channels), tiles fastest: each of the 64 stores of a
// thread is a 256-byte run across the wave.  MODE 0: B^T d B of the 4x4 input patch; MODE 1: A dy A^T of the 2x2 dy tile.
template <int MODE>
__global__ __launch_bounds__(256) void wino_xform_t_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int H, int W, int C,
                                                           int TH, int TW, int Tpad) {
    // block = 32 tiles x 32 channels: 8 lanes cover the 128-byte channel run of one tile (coalesced loads); the sixteen planes
    // go through LDS four at a time and leave as rows [plane][channel] of 32 consecutive tiles (128-byte stores).
    __shared__ float tr[4][32][33];
    const int C4 = C >> 2;
    const int cgroups = (C4 + 7) / 8;
    const size_t tiles = (size_t)N * TH * TW;
    const int tile_blocks = Tpad / 32;
    const int q = threadIdx.x & 7, tl = threadIdx.x >> 3;            // channel quad within the group, tile within the block
    for (int blk = blockIdx.x; blk < tile_blocks * cgroups; blk += gridDim.x) {
        const int cg = blk % cgroups;
        const size_t tile0 = (size_t)(blk / cgroups) * 32, tile = tile0 + tl;
        const int c4 = cg * 8 + q;
        const bool c_ok = c4 < C4;
        f32x4 v[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b2 = 0; b2 < 4; ++b2) v[a][b2] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (tile < tiles && c_ok) {
            const int tw = (int)(tile % TW), th = (int)((tile / TW) % TH), n = (int)(tile / ((size_t)TW * TH));
            if (MODE == 0) {
                f32x4 d[4][4], t[4][4];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b2 = 0; b2 < 4; ++b2) {
                        const int ih = 2 * th - 1 + a, iw = 2 * tw - 1 + b2;
                        const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                        d[a][b2] = ok ? *reinterpret_cast<const f32x4*>(src + (((size_t)n * H + ih) * W + iw) * C + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                for (int b2 = 0; b2 < 4; ++b2) {
                    t[0][b2] = d[0][b2] - d[2][b2]; t[1][b2] = d[1][b2] + d[2][b2]; t[2][b2] = d[2][b2] - d[1][b2]; t[3][b2] = d[1][b2] - d[3][b2];
                }
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    v[a][0] = t[a][0] - t[a][2]; v[a][1] = t[a][1] + t[a][2]; v[a][2] = t[a][2] - t[a][1]; v[a][3] = t[a][1] - t[a][3];
                }
            } else {
                f32x4 d[2][2], t[4][2];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b2 = 0; b2 < 2; ++b2) {
                        const int oh = 2 * th + a, ow = 2 * tw + b2;
                        d[a][b2] = (oh < H && ow < W) ? *reinterpret_cast<const f32x4*>(src + (((size_t)n * H + oh) * W + ow) * C + c4 * 4)
                                                      : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2) {             // A d: rows (1,0), (1,1), (1,-1), (0,-1)
                    t[0][b2] = d[0][b2]; t[1][b2] = d[0][b2] + d[1][b2]; t[2][b2] = d[0][b2] - d[1][b2]; t[3][b2] = -d[1][b2];
                }
#pragma unroll
                for (int a = 0; a < 4; ++a) {                // (A d) A^T
                    v[a][0] = t[a][0]; v[a][1] = t[a][0] + t[a][1]; v[a][2] = t[a][0] - t[a][1]; v[a][3] = -t[a][1];
                }
            }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {                        // planes 4a .. 4a+3
            __syncthreads();
#pragma unroll
            for (int b2 = 0; b2 < 4; ++b2)
#pragma unroll
                for (int e = 0; e < 4; ++e) tr[b2][q * 4 + e][tl] = v[a][b2][e];
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int row = it * 8 + (threadIdx.x >> 5), col = threadIdx.x & 31;        // row = plane-in-group * 32 + channel
                const int b2 = row >> 5, cl = row & 31;
                const int c = cg * 32 + cl;
                if (c < C) dst[((size_t)(a * 4 + b2) * C + c) * Tpad + tile0 + col] = tr[b2][cl][col];
            }
        }
    }
}

// dw[co][ci][3][3] = G^T Z G with Z[xi] = sum over the K slices of Zs[xi][slice][co][ci] (slices added in index order)
__global__ void wino_wgrad_finish_kernel(const float* __restrict__ Zs, float* __restrict__ dw, int Co, int Ci, int ksplit) {
    const size_t total = (size_t)Co * Ci;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float z[4][4];
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            float s = 0.f;
            for (int k = 0; k < ksplit; ++k) s += Zs[((size_t)xi * ksplit + k) * total + i];
            z[xi >> 2][xi & 3] = s;
        }
        float t[3][4];                                       // G^T Z
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            t[0][b] = z[0][b] + 0.5f * (z[1][b] + z[2][b]);
            t[1][b] = 0.5f * (z[1][b] - z[2][b]);
            t[2][b] = 0.5f * (z[1][b] + z[2][b]) + z[3][b];
        }
        float* o = dw + i * 9;
#pragma unroll
        for (int r = 0; r < 3; ++r) {                        // (G^T Z) G
            o[r * 3 + 0] = t[r][0] + 0.5f * (t[r][1] + t[r][2]);
            o[r * 3 + 1] = 0.5f * (t[r][1] - t[r][2]);
            o[r * 3 + 2] = 0.5f * (t[r][1] + t[r][2]) + t[r][3];
        }
    }
}

// db[c] = sum over pixels of dy[pixel][c]: per-block partial rows, then a fixed-order sum over the blocks (reproducible).
// partial: a block covers 1024 / C4' consecutive pixel rows per sweep with 16-byte loads along the channels (C4' = channel quads
// rounded up to a power of two <= 256); rows of one block are combined through LDS in row order.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ dy, float* __restrict__ part, size_t M, int ld, int C,
                                                             int cq_pow2) {
    __shared__ f32x4 red[256];
    const int C4 = (C + 3) / 4, rpi = 256 / cq_pow2;            // rows per sweep
    const int cq = threadIdx.x % cq_pow2, r = threadIdx.x / cq_pow2;
    for (int q0 = 0; q0 < C4; q0 += cq_pow2) {
        const int q = q0 + cq;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (q < C4 && q * 4 + 3 < ld)
            for (size_t m = (size_t)blockIdx.x * rpi + r; m < M; m += (size_t)gridDim.x * rpi) s += *reinterpret_cast<const f32x4*>(dy + m * ld + q * 4);
        red[threadIdx.x] = s;
        __syncthreads();
        if (r == 0 && q < C4) {
            f32x4 t = red[cq];
            for (int k = 1; k < rpi; ++k) t += red[k * cq_pow2 + cq];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (q * 4 + e < C) part[(size_t)blockIdx.x * C + q * 4 + e] = t[e];
        }
        __syncthreads();
    }
}
// final: 16 channels per block, 16 lanes stride over the partial rows of each channel, combined in lane order
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, float* __restrict__ db, int nblk, int C) {
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, l = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float s = 0.f;
    if (c < C)
        for (int k = l; k < nblk; k += 16) s += part[(size_t)k * C + c];
    red[l][cl] = s;
    __syncthreads();
    if (l == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][cl];
        db[c] = t;
    }
}

// final for the partial rows wino4_dy_kernel<true> leaves (row stride ldp = channels rounded up to 4): 16 channels per block as four
// 16-byte quads, 64 lanes stride over the rows of each quad with two sums in flight, combined in lane order
__device__ __forceinline__ void colsum_final4_body(const float* __restrict__ part, float* __restrict__ db, int nblk, int C, int ldp, int bx) {
    __shared__ f32x4 red[64][5];
    const int q = threadIdx.x & 3, l = threadIdx.x >> 2;
    const int c0 = (bx * 4 + q) * 4;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    if (c0 < ldp)
        for (int k = l; k < nblk; k += 128) {
            s0 += *reinterpret_cast<const f32x4*>(part + (size_t)k * ldp + c0);
            if (k + 64 < nblk) s1 += *reinterpret_cast<const f32x4*>(part + (size_t)(k + 64) * ldp + c0);
        }
    red[l][q] = s0 + s1;
    __syncthreads();
    if (l == 0 && c0 < ldp) {
        f32x4 t = red[0][q];
        for (int k = 1; k < 64; ++k) t += red[k][q];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c0 + e < C) db[c0 + e] = t[e];
    }
}
__global__ __launch_bounds__(256) void colsum_final4_kernel(const float* __restrict__ part, float* __restrict__ db, int nblk, int C, int ldp) {
    colsum_final4_body(part, db, nblk, C, ldp, (int)blockIdx.x);
}
// the same work as extra blocks at the end of wino4_wgrad_finish_kernel's grid (one launch less per layer); part == NULL: none
struct BiasTail { const float* part; float* db; int nblk, C, ldp; };
int g_bias_tail = 1;              // tuning aid (ssd_tune_set_wino_bias_tail): 0 = the final sums as their own launch

// ---- F(4x4, 3x3): 36 multiplies per 4x4 output tile (4x fewer than direct, 1.78x fewer than F(2x2)); 6x6 input patches, planes
// 2.25x the tensor.  Interpolation points 0, +-1, +-2, inf (Lavin & Gray); coefficients up to 8 and 1/24, so the f32 result is
// less exact than F(2x2)'s (measured ~1e-5 of the output scale) -- used where that still clears the 1e-4 bar.
__device__ constexpr float W4_BT[6][6] = {{4, 0, -5, 0, 1, 0}, {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0},
                                          {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};
__device__ constexpr float W4_G[6][3] = {{0.25f, 0, 0}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                                         {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0, 0, 1}};
__device__ constexpr float W4_AT[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};
// one row of G against three values, as an explicit product + two fused multiply-adds: the filter transform is written in three
// kernels (per layer, job table, limb rows) whose results must agree bit for bit, whatever the compiler would contract in each
__device__ __forceinline__ float w4_g_dot(int row, float x0, float x1, float x2) {
    return __builtin_fmaf(W4_G[row][2], x2, __builtin_fmaf(W4_G[row][1], x1, W4_G[row][0] * x0));
}

// One element (plane p, row n, column k) of a transformed filter.  x3 = 0: U [36][Nrows][K] f32.  x3 = 1: the three exact bf16 limbs of
// the value in the layout of csrc/gemm_x3.hip, [36][K/16][3][pad128(Nrows)][16] bf16 (rows beyond Nrows: never written, zero from the
// allocation).
__device__ __forceinline__ void store_u(float* __restrict__ U, int x3, int p, int n, int k, int Nrows, int K, float v) {
    if (!x3) {
        U[((size_t)p * Nrows + n) * K + k] = v;
        return;
    }
    const __bf16 h = (__bf16)v;                       // round-to-nearest limbs; both residuals are exact (gemm_x3.hip)
    const float r1 = v - (float)h;
    const __bf16 m = (__bf16)r1;
    const size_t limb = (size_t)((Nrows + 127) / 128 * 128) * 16;
    __bf16* d = reinterpret_cast<__bf16*>(U) + ((size_t)p * (K >> 4) + (k >> 4)) * 3 * limb + (size_t)n * 16 + (k & 15);
    d[0] = h;
    d[limb] = m;
    d[2 * limb] = (__bf16)(r1 - (float)m);
}

__global__ void wino4_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int Co, int Ci, int Nrows, int K, int mode, int x3) {
    const size_t total = (size_t)Nrows * K;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % K), n = (int)(i / K);
        const int co = mode == 0 ? n : k, ci = mode == 0 ? k : n;      // mode 2: the adjoint form -- rows ci, K = co like mode 1, taps NOT rotated
        float g[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s)
                g[r][s] = (co < Co && ci < Ci) ? w[((size_t)co * Ci + ci) * 9 + (mode != 1 ? r * 3 + s : (2 - r) * 3 + (2 - s))] : 0.f;
        float t[6][3];
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int s = 0; s < 3; ++s) t[a][s] = w4_g_dot(a, g[0][s], g[1][s], g[2][s]);
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b)
                store_u(U, x3, a * 6 + b, n, k, Nrows, K, w4_g_dot(b, t[a][0], t[a][1], t[a][2]));
    }
}

// bits (optional): per (tile, channel quad) one 64-bit word, bit (a*4+b)*4+e = x[4th+a][4tw+b][4c4+e] > 0 -- the ReLU mask of this
// layer's input on the tile grid its data gradient is written on: the dgrad epilogue then reads 1 bit instead of 32 per element.
__global__ __launch_bounds__(256) void wino4_input_kernel(const float* __restrict__ x, float* __restrict__ V, int N, int H, int W, int C,
                                                          int TH, int TW, unsigned long long* __restrict__ bits, const Lat lat) {
    const int C4 = C >> 2;
    const size_t tiles = (size_t)N * TH * TW, total = tiles * C4;
    const size_t plane = tiles * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        const size_t tile = i / C4;
        const int tw = (int)(tile % TW), th = (int)((tile / TW) % TH), n = (int)(tile / ((size_t)TW * TH));
        int offh, bh, offw, bw;
        lat_tile(lat.h0, lat.D, th, offh, bh);
        lat_tile(lat.w0, lat.D, tw, offw, bw);
        f32x4 t[6][6];                                       // B^T d, one input row at a time
        unsigned long long word = 0ull;
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            f32x4 d[6];
            const int lw = bw - 1 + b, iw = lat.D * lw + offw;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                const int lh = bh - 1 + a, ih = lat.D * lh + offh;
                const bool ok = lh >= 0 && lw >= 0 && ih < H && iw < W;
                d[a] = ok ? *reinterpret_cast<const f32x4*>(x + (((size_t)n * H + ih) * W + iw) * C + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                if (a >= 1 && a <= 4 && b >= 1 && b <= 4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) word |= (unsigned long long)(d[a][e] > 0.f) << (((a - 1) * 4 + (b - 1)) * 4 + e);
                }
            }
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    if (W4_BT[a][k] != 0.f) acc += W4_BT[a][k] * d[k];
                t[a][b] = acc;
            }
        }
        float* dst = V + tile * C + c4 * 4;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    if (W4_BT[b][k] != 0.f) acc += W4_BT[b][k] * t[a][k];
                *reinterpret_cast<f32x4*>(dst + (size_t)(a * 6 + b) * plane) = acc;
            }
        if (bits != nullptr) bits[i] = word;
    }
}

__global__ __launch_bounds__(256) void wino4_output_kernel(const float* __restrict__ Mx, float* __restrict__ out, int N, int H, int W, int C,
                                                           int Cvalid, int ldo, int TH, int TW, const float* __restrict__ bias,
                                                           const float* __restrict__ mask, int relu, int accumulate,
                                                           const unsigned long long* __restrict__ mask_bits, const Lat lat) {
    const int C4 = C >> 2;
    const size_t tiles = (size_t)N * TH * TW, total = tiles * C4;
    const size_t plane = tiles * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        const size_t tile = i / C4;
        const int tw = (int)(tile % TW), th = (int)((tile / TW) % TH), n = (int)(tile / ((size_t)TW * TH));
        int offh, bh, offw, bw;
        lat_tile(lat.h0, lat.D, th, offh, bh);
        lat_tile(lat.w0, lat.D, tw, offw, bw);
        const float* src = Mx + tile * C + c4 * 4;
        f32x4 t[4][6];                                       // A^T m, one plane column at a time
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            f32x4 m[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) m[a] = *reinterpret_cast<const f32x4*>(src + (size_t)(a * 6 + b) * plane);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    if (W4_AT[a][k] != 0.f) acc += W4_AT[a][k] * m[k];
                t[a][b] = acc;
            }
        }
        const unsigned long long word = mask_bits != nullptr ? mask_bits[i] : 0ull;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (bias != nullptr) {
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[e] = c4 * 4 + e < Cvalid ? bias[c4 * 4 + e] : 0.f;
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int oh = lat.D * (bh + a) + offh;
            if (oh >= H) continue;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int ow = lat.D * (bw + b) + offw;
                if (ow >= W) continue;
                f32x4 v = bv;
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    if (W4_AT[b][k] != 0.f) v += W4_AT[b][k] * t[a][k];
                const size_t idx = (((size_t)n * H + oh) * W + ow) * ldo + c4 * 4;
                if (accumulate) v += *reinterpret_cast<const f32x4*>(out + idx);
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];
                }
                if (mask_bits != nullptr) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ((word >> ((a * 4 + b) * 4 + e)) & 1ull) ? v[e] : 0.f;
                } else if (mask != nullptr) {
                    const f32x4 mk = *reinterpret_cast<const f32x4*>(mask + idx);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = mk[e] > 0.f ? v[e] : 0.f;
                }
                *reinterpret_cast<f32x4*>(out + idx) = v;
            }
        }
    }
}

// Output transform + bias + ReLU + the 2x2 / stride-2 max pool that follows (conv1_2, conv2_2, conv3_3): a 4x4 output tile holds
// exactly four pool windows, so the full-resolution activation -- which only the pool reads -- never goes to memory.  Window scan
// order, NaN rule and argmax encoding are maxpool_fwd_kernel's (elementwise.hip), windows cut by the map's edge (ceil mode) included.
__global__ __launch_bounds__(256) void wino4_output_pool_kernel(const float* __restrict__ Mx, float* __restrict__ yp, uint8_t* __restrict__ am,
                                                                int N, int H, int W, int C, int TH, int TW, const float* __restrict__ bias,
                                                                int Ho, int Wo) {
    const int C4 = C >> 2;
    const size_t tiles = (size_t)N * TH * TW, total = tiles * C4;
    const size_t plane = tiles * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        const size_t tile = i / C4;
        const int tw = (int)(tile % TW), th = (int)((tile / TW) % TH), n = (int)(tile / ((size_t)TW * TH));
        const float* src = Mx + tile * C + c4 * 4;
        f32x4 t[4][6];
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            f32x4 m[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) m[a] = *reinterpret_cast<const f32x4*>(src + (size_t)(a * 6 + b) * plane);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    if (W4_AT[a][k] != 0.f) acc += W4_AT[a][k] * m[k];
                t[a][b] = acc;
            }
        }
        const f32x4 bv = bias != nullptr ? *reinterpret_cast<const f32x4*>(bias + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pa = 0; pa < 2; ++pa) {
            const int oh = 2 * th + pa;
            f32x4 y[2][4];                                   // the two activation rows this pooled row reads
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    f32x4 v = bv;
#pragma unroll
                    for (int k = 0; k < 6; ++k)
                        if (W4_AT[b][k] != 0.f) v += W4_AT[b][k] * t[2 * pa + r][k];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];
                    y[r][b] = v;
                }
            if (oh >= Ho) continue;
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                const int ow = 2 * tw + pb;
                if (ow >= Wo) continue;
                f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                int bi[4] = {0, 0, 0, 0};
                bool first = true;
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    if (2 * oh + r >= H) continue;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        if (2 * ow + q >= W) continue;
                        const f32x4 v = y[r][2 * pb + q];
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (first || v[e] > best[e] || v[e] != v[e]) {
                                best[e] = v[e];
                                bi[e] = r * 2 + q;
                            }
                        first = false;
                    }
                }
                const size_t o = (((size_t)n * Ho + oh) * Wo + ow) * C4 + c4;
                *reinterpret_cast<f32x4*>(yp + o * 4) = best;
                if (am != nullptr)
                    *reinterpret_cast<uint32_t*>(am + o * 4) =
                        (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
            }
        }
    }
}

// F(4x4,3x3) weight gradient pieces: transposed transforms into rows [plane][channel][tile] (36 planes), inverse transform.
template <int MODE>
__global__ __launch_bounds__(256) void wino4_xform_t_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int H, int W, int C,
                                                            int TH, int TW, int Tpad) {
    __shared__ float tr[6][32][33];
    const int C4 = C >> 2;
    const int cgroups = (C4 + 7) / 8;
    const size_t tiles = (size_t)N * TH * TW;
    const int tile_blocks = Tpad / 32;
    const int q = threadIdx.x & 7, tl = threadIdx.x >> 3;
    for (int blk = blockIdx.x; blk < tile_blocks * cgroups; blk += gridDim.x) {
        const int cg = blk % cgroups;
        const size_t tile0 = (size_t)(blk / cgroups) * 32, tile = tile0 + tl;
        const int c4 = cg * 8 + q;
        const bool live = tile < tiles && c4 < C4;
        const int tw = live ? (int)(tile % TW) : 0, th = live ? (int)((tile / TW) % TH) : 0, n = live ? (int)(tile / ((size_t)TW * TH)) : 0;
        f32x4 t[6][6];                                       // first-dimension transform, one source column at a time
        constexpr int NS = MODE == 0 ? 6 : 4;                // source patch edge: 6x6 input pixels / 4x4 dy pixels
#pragma unroll
        for (int b2 = 0; b2 < NS; ++b2) {
            f32x4 d[NS];
#pragma unroll
            for (int a = 0; a < NS; ++a) {
                const int ih = MODE == 0 ? 4 * th - 1 + a : 4 * th + a, iw = MODE == 0 ? 4 * tw - 1 + b2 : 4 * tw + b2;
                const bool ok = live && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                d[a] = ok ? *reinterpret_cast<const f32x4*>(src + (((size_t)n * H + ih) * W + iw) * C + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    const float cf = MODE == 0 ? W4_BT[a][k] : W4_AT[k][a];      // B^T d   |   A dy  (A = (A^T)^T)
                    if (cf != 0.f) acc += cf * d[k];
                }
                t[a][b2] = acc;
            }
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) {                        // planes (a, 0..5): second-dimension transform, then out through LDS
            __syncthreads();
#pragma unroll
            for (int b2 = 0; b2 < 6; ++b2) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    const float cf = MODE == 0 ? W4_BT[b2][k] : W4_AT[k][b2];
                    if (cf != 0.f) acc += cf * t[a][k];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) tr[b2][q * 4 + e][tl] = acc[e];
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 24; ++it) {
                const int row = it * 8 + (threadIdx.x >> 5), col = threadIdx.x & 31;        // row = plane-in-group * 32 + channel
                const int b2 = row >> 5, cl = row & 31;
                const int c = cg * 32 + cl;
                if (c < C) dst[((size_t)(a * 6 + b2) * C + c) * Tpad + tile0 + col] = tr[b2][cl][col];
            }
        }
    }
}

// ---- weight gradient without transposed copies --------------------------------------------------------------------------------
// dy tiles into the plane layout of the forward transforms: Y[plane][tile][channel] = A dy A^T (4x4 block -> 6x6 planes)
// BIAS: the pass also leaves per-block column sums of dy (the bias gradient) in part[block][Cvalid]; the launch makes the thread
// count a multiple of C/4, so a thread keeps its channel quad over the grid-stride loop, and a block combines its threads in a fixed order.
// Vd (optional): the same pass also writes B^T dy B of the 6x6 patch around each block -- the input planes of this layer's dgrad
// (wino4_input_kernel's output) -- so dy is read from HBM once for both; the 4x4 block is the patch's interior and comes from cache.
// POOLED: dy is not in memory.  It is the gradient of a 2x2 / stride-2 max pool of this layer's ReLU output, given as the pooled
// gradient dpool (Hp x Wp), the window argmax codes (byte = 2 * row + column inside the window, as the pool kernels write them) and the
// pooled forward output (the ReLU gate: > 0).  The 4x4 pooled cells under a 6x6 patch are read once, and dy(ih, iw) = the gated dpool
// of the cell where the code names (ih & 1, iw & 1), else 0 -- exactly what ssd_maxpool_bwd_gated would have scattered to memory.
struct PoolSrc { const uint8_t* am; const float* gate; int Hp, Wp; };
template <bool BIAS, bool POOLED = false>
__global__ __launch_bounds__(256) void wino4_dy_kernel(const float* __restrict__ dy, float* __restrict__ Y, int N, int H, int W, int C,
                                                       int TH, int TW, float* __restrict__ part, int Cvalid, float* __restrict__ Vd,
                                                       const PoolSrc ps, const Lat lat) {
    const int C4 = C >> 2;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    const size_t tiles = (size_t)N * TH * TW, total = tiles * C4;
    const size_t plane = tiles * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        const size_t tile = i / C4;
        const int tw = (int)(tile % TW), th = (int)((tile / TW) % TH), n = (int)(tile / ((size_t)TW * TH));
        int offh, bh, offw, bw;                              // sub-lattice of the tile (POOLED sources are plain: D = 1)
        lat_tile(lat.h0, lat.D, th, offh, bh);
        lat_tile(lat.w0, lat.D, tw, offw, bw);
        f32x4 pg[4][4];                                      // POOLED: gated pooled gradient and codes of cells (2th-1+r, 2tw-1+q)
        uint32_t pc[4][4];
        if (POOLED) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool inner = (r == 1 || r == 2) && (q == 1 || q == 2);
                    const int ph = 2 * th - 1 + r, pw = 2 * tw - 1 + q;
                    const bool ok = (inner || Vd != nullptr) && (unsigned)ph < (unsigned)ps.Hp && (unsigned)pw < (unsigned)ps.Wp;
                    f32x4 d = {0.f, 0.f, 0.f, 0.f};
                    uint32_t a = 0xffffffffu;
                    if (ok) {
                        const size_t idx = (((size_t)n * ps.Hp + ph) * ps.Wp + pw) * C4 + c4;
                        d = *reinterpret_cast<const f32x4*>(dy + idx * 4);
                        const f32x4 yv = *reinterpret_cast<const f32x4*>(ps.gate + idx * 4);
                        a = *reinterpret_cast<const uint32_t*>(ps.am + idx * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) d[e] = yv[e] > 0.f ? d[e] : 0.f;
                    }
                    pg[r][q] = d;
                    pc[r][q] = a;
                }
        }
        // element (a, b) of the 6x6 patch whose corner is (4th-1, 4tw-1); a, b are compile-time after unrolling
        auto patch = [&](int a, int b) -> f32x4 {
            const int lh = bh - 1 + a, lw = bw - 1 + b;
            const int ih = lat.D * lh + offh, iw = lat.D * lw + offw;
            const bool ok = lh >= 0 && lw >= 0 && ih < H && iw < W;
            if (!POOLED)
                return ok ? *reinterpret_cast<const f32x4*>(dy + (((size_t)n * H + ih) * W + iw) * C + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            const int r = (a + 1) >> 1, q = (b + 1) >> 1;
            const uint32_t code = (uint32_t)(((a + 1) & 1) * 2 + ((b + 1) & 1));
            f32x4 g;
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = (ok && ((pc[r][q] >> (8 * e)) & 0xffu) == code) ? pg[r][q][e] : 0.f;
            return g;
        };
        if (Vd != nullptr) {                                 // uniform: the dgrad planes, as wino4_input_kernel forms them
            f32x4 u[6][6];
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                f32x4 d[6];
#pragma unroll
                for (int a = 0; a < 6; ++a) d[a] = patch(a, b);
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < 6; ++k)
                        if (W4_BT[a][k] != 0.f) acc += W4_BT[a][k] * d[k];
                    u[a][b] = acc;
                }
            }
            float* dv = Vd + tile * C + c4 * 4;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = 0; b < 6; ++b) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < 6; ++k)
                        if (W4_BT[b][k] != 0.f) acc += W4_BT[b][k] * u[a][k];
                    *reinterpret_cast<f32x4*>(dv + (size_t)(a * 6 + b) * plane) = acc;
                }
        }
        f32x4 t[6][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            f32x4 d[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) d[a] = patch(a + 1, b + 1);      // the 4x4 block is the patch's interior (from cache / registers)
            if (BIAS) bsum += (d[0] + d[1]) + (d[2] + d[3]);
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (W4_AT[k][a] != 0.f) acc += W4_AT[k][a] * d[k];
                t[a][b] = acc;
            }
        }
        float* dst = Y + tile * C + c4 * 4;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (W4_AT[k][b] != 0.f) acc += W4_AT[k][b] * t[a][k];
                *reinterpret_cast<f32x4*>(dst + (size_t)(a * 6 + b) * plane) = acc;
            }
    }
    if (BIAS) {
        __shared__ f32x4 red[256];
        red[threadIdx.x] = bsum;
        __syncthreads();
        if ((int)threadIdx.x < C4) {
            f32x4 t = red[threadIdx.x];
            for (int k = threadIdx.x + C4; k < 256; k += C4) t += red[k];
            const int c4 = (int)(((size_t)blockIdx.x * 256 + threadIdx.x) % C4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c4 * 4 + e < Cvalid) part[(size_t)blockIdx.x * ((Cvalid + 3) / 4 * 4) + c4 * 4 + e] = t[e];
        }
    }
}

// ---- data gradient in the ADJOINT Winograd form (round 4) -----------------------------------------------------------------------------
// The forward y = A^T [(G g G^T) (.) (B^T d B)] A is linear in d; its transpose maps dy to dx:
//     dx patch (6x6, corner (4th-1, 4tw-1)) = B [ sum_co (G g G^T)[co][ci] (.) (A dy A^T)[co] ] B^T,    patches OVERLAP-ADDED
// so the data gradient multiplies the SAME planes A dy A^T the weight gradient multiplies (no second set B^T dy B: 2.25x the gradient
// tensor neither written nor read), against the FORWARD filter transform laid out [plane][ci][co] (weight job, pad0 bit 2).  Measured
// against f64 it is as exact as the rotated-filter form (tools: 3.2e-6 vs 3.3e-6 at 256 channels).  What the overlap-add needs from the
// neighbours of tile (th, tw) is little: B has B[0][:] = (4,0,0,0,0,0) and B[5][:] = (0,0,0,0,0,1), so row 4th of dx receives only plane
// row 5 of the tile above, row 4th+3 only plane row 0 (x4) of the tile below, likewise the columns, and the four corner pixels one plane
// each of the diagonal tiles: 36 + 4*6 + 4 plane values per tile, the 28 extra ones from L2 (they are another thread's own planes).
// A thread = (tile, channel quad) forms its 4x4 block of dx in registers, applies the ReLU mask, and then EITHER stores it (dx / += dx)
// OR -- MAKE_Y -- takes it as the dy block of the layer below and writes that layer's planes A dy A^T (+ the bias partial sums, as
// wino4_dy_kernel<true> does): between two chained layers the gradient tensor itself is never written or read.
template <bool MAKE_Y, bool BIAS>
__global__ __launch_bounds__(256) void wino4_adj_out_kernel(const float* __restrict__ Md, int N, int H, int W, int C, int TH, int TW,
                                                            float* __restrict__ dx, int ldo, const float* __restrict__ mask,
                                                            const unsigned long long* __restrict__ mask_bits, int accumulate,
                                                            float* __restrict__ Y, float* __restrict__ part, int Cvalid) {
    const int C4 = C >> 2;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    const size_t tiles = (size_t)N * TH * TW, total = tiles * C4;
    const size_t plane = tiles * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        const size_t tile = i / C4;
        const int tw = (int)(tile % TW), th = (int)((tile / TW) % TH), n = (int)(tile / ((size_t)TW * TH));
        const float* src = Md + tile * C + c4 * 4;
        auto ld = [&](const float* base, int a, int b) -> f32x4 { return *reinterpret_cast<const f32x4*>(base + (size_t)(a * 6 + b) * plane); };
        // r[j] = sum_b m[b] B^T[b][j+1]: one plane row against the four interior columns of B^T
        auto rowx = [&](const f32x4 (&m)[6], f32x4 (&r)[4]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int b = 0; b < 6; ++b)
                    if (W4_BT[b][j + 1] != 0.f) acc += W4_BT[b][j + 1] * m[b];
                r[j] = acc;
            }
        };
        f32x4 d[4][4];
#pragma unroll
        for (int a2 = 0; a2 < 4; ++a2)
#pragma unroll
            for (int b2 = 0; b2 < 4; ++b2) d[a2][b2] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 6; ++a) {                        // the tile's own patch, interior 4x4: sum_a B^T[a][i+1] (M[a][:] B^T)[j+1]
            f32x4 m[6], r[4];
#pragma unroll
            for (int b = 0; b < 6; ++b) m[b] = ld(src, a, b);
            rowx(m, r);
#pragma unroll
            for (int i2 = 0; i2 < 4; ++i2)
                if (W4_BT[a][i2 + 1] != 0.f) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) d[i2][j] += W4_BT[a][i2 + 1] * r[j];
                }
        }
        const bool up = th > 0, down = th + 1 < TH, left = tw > 0, right = tw + 1 < TW;
        if (up) {                                            // row 5 of the patch above lands on this tile's row 0
            const float* s2 = src - (size_t)TW * C;
            f32x4 m[6], r[4];
#pragma unroll
            for (int b = 0; b < 6; ++b) m[b] = ld(s2, 5, b);
            rowx(m, r);
#pragma unroll
            for (int j = 0; j < 4; ++j) d[0][j] += r[j];
        }
        if (down) {                                          // row 0 of the patch below (B[0][0] = 4) on row 3
            const float* s2 = src + (size_t)TW * C;
            f32x4 m[6], r[4];
#pragma unroll
            for (int b = 0; b < 6; ++b) m[b] = ld(s2, 0, b);
            rowx(m, r);
#pragma unroll
            for (int j = 0; j < 4; ++j) d[3][j] += 4.f * r[j];
        }
        if (left) {                                          // column 5 of the patch to the left on column 0
            const float* s2 = src - C;
            f32x4 m[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) m[a] = ld(s2, a, 5);
#pragma unroll
            for (int i2 = 0; i2 < 4; ++i2) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int a = 0; a < 6; ++a)
                    if (W4_BT[a][i2 + 1] != 0.f) acc += W4_BT[a][i2 + 1] * m[a];
                d[i2][0] += acc;
            }
        }
        if (right) {                                         // column 0 of the patch to the right (x4) on column 3
            const float* s2 = src + C;
            f32x4 m[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) m[a] = ld(s2, a, 0);
#pragma unroll
            for (int i2 = 0; i2 < 4; ++i2) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int a = 0; a < 6; ++a)
                    if (W4_BT[a][i2 + 1] != 0.f) acc += W4_BT[a][i2 + 1] * m[a];
                d[i2][3] += 4.f * acc;
            }
        }
        if (up && left) d[0][0] += ld(src - (size_t)(TW + 1) * C, 5, 5);
        if (up && right) d[0][3] += 4.f * ld(src - (size_t)(TW - 1) * C, 5, 0);
        if (down && left) d[3][0] += 4.f * ld(src + (size_t)(TW - 1) * C, 0, 5);
        if (down && right) d[3][3] += 16.f * ld(src + (size_t)(TW + 1) * C, 0, 0);

        const unsigned long long word = mask_bits != nullptr ? mask_bits[i] : ~0ull;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int oh = 4 * th + a;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int ow = 4 * tw + b;
                const bool in = oh < H && ow < W;
                f32x4 v = d[a][b];
                const size_t idx = (((size_t)n * H + oh) * W + ow) * ldo + c4 * 4;
                if (!MAKE_Y && accumulate && in) v += *reinterpret_cast<const f32x4*>(dx + idx);
                if (mask_bits != nullptr) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ((word >> ((a * 4 + b) * 4 + e)) & 1ull) ? v[e] : 0.f;
                } else if (mask != nullptr && in) {
                    const f32x4 mk = *reinterpret_cast<const f32x4*>(mask + idx);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = mk[e] > 0.f ? v[e] : 0.f;
                }
                if (!in) v = f32x4{0.f, 0.f, 0.f, 0.f};        // beyond the map: no such pixel (and, as a dy block, zero)
                if (!MAKE_Y) {
                    if (in) *reinterpret_cast<f32x4*>(dx + idx) = v;
                } else {
                    d[a][b] = v;
                }
            }
        }
        if (MAKE_Y) {                                        // the block is the dy of the layer below: its planes A dy A^T (wino4_dy_kernel's sums)
            f32x4 t[6][4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (BIAS) bsum += (d[0][b] + d[1][b]) + (d[2][b] + d[3][b]);
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (W4_AT[k][a] != 0.f) acc += W4_AT[k][a] * d[k][b];
                    t[a][b] = acc;
                }
            }
            float* dst = Y + tile * C + c4 * 4;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = 0; b < 6; ++b) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (W4_AT[k][b] != 0.f) acc += W4_AT[k][b] * t[a][k];
                    *reinterpret_cast<f32x4*>(dst + (size_t)(a * 6 + b) * plane) = acc;
                }
        }
    }
    if (MAKE_Y && BIAS) {
        __shared__ f32x4 red[256];
        red[threadIdx.x] = bsum;
        __syncthreads();
        if ((int)threadIdx.x < C4) {
            f32x4 t = red[threadIdx.x];
            for (int k = threadIdx.x + C4; k < 256; k += C4) t += red[k];
            const int c4 = (int)(((size_t)blockIdx.x * 256 + threadIdx.x) % C4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c4 * 4 + e < Cvalid) part[(size_t)blockIdx.x * ((Cvalid + 3) / 4 * 4) + c4 * 4 + e] = t[e];
        }
    }
}

// Batched C[m][n] = sum_k A[k][m] B[k][n] on v_mfma_f32_32x32x2_f32: both operands arrive with the reduction index as the slow
// dimension (rows = tiles, columns = channels -- the layout the forward transforms write), so a 32-row slab of each goes to LDS
// exactly as it lies in memory and an MFMA lane fetches its operand with one ds_read_b32 (lanes 0-31 walk the columns of row 2q,
// lanes 32-63 those of row 2q+1; row stride 96 floats puts the two half-waves on disjoint banks).  64x64 tile, four waves 2x2,
// split-K over blockIdx.y into out[batch][split][M][N].  Rows beyond K and columns beyond lda / ldb read as zero (buffer range check).
struct TnParams {
    const float* a; const float* b; float* out;
    int M, N, K, lda, ldb, tiles_m, tiles_n, ksplit, ksteps_per_split, groups;      // groups = planes x ksplit
    size_t batch_a, batch_b;
    unsigned a_bytes, b_bytes;
};
constexpr int TN_LD = 96;
__device__ __forceinline__ f32x4 tn_load16(__amdgpu_buffer_rsrc_t srd, unsigned voff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, (int)voff, 0, 0));
}
__global__ __launch_bounds__(256) void wino_gemm_tn_kernel(const TnParams p) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 32 * TN_LD];
    float* As = lds;
    float* Bs = lds + 32 * TN_LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // Workgroups go to the 8 XCDs round-robin by linear id, and each XCD has its own L2.  All output tiles of one (plane, K slice) --
    // the only blocks that share operand panels -- are therefore given ids with the same residue mod 8: they run on ONE XCD, next to
    // each other in time, and every panel leaves HBM / the fabric once instead of once per XCD that holds one of its tiles.
    const int nblk = p.tiles_m * p.tiles_n;
    const int idx = (int)(blockIdx.x >> 3), grp = (idx / nblk) * 8 + (int)(blockIdx.x & 7), lid = idx % nblk;
    if (grp >= p.groups) return;                                     // uniform: the grid is padded to 8 x ceil(groups / 8) groups
    const int m0 = (lid / p.tiles_n) * 64, n0 = (lid % p.tiles_n) * 64;
    const int by = grp % p.ksplit, bz = grp / p.ksplit;
    const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a + (size_t)bz * p.batch_a), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b + (size_t)bz * p.batch_b), 0, (int)p.b_bytes, 0x00020000);
    const int mchunk = tid & 15, krow = tid >> 4;
    const unsigned OOB = 0xFFFFFFF0u;
    const int col_a = m0 + mchunk * 4, col_b = n0 + mchunk * 4;
    const bool ok_a = col_a < p.lda, ok_b = col_b < p.ldb;
    const int ksteps = (p.K + 31) / 32;
    const int kt_begin = by * p.ksteps_per_split;
    const int KT = min(ksteps - kt_begin, p.ksteps_per_split);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f32x4 ra[2], rb[2];
    auto issue = [&](int kt) {
        const int k0 = (kt_begin + kt) * 32 + krow;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            ra[j] = tn_load16(srd_a, ok_a ? (unsigned)(((size_t)(k0 + 16 * j) * p.lda + col_a) * 4) : OOB);
            rb[j] = tn_load16(srd_b, ok_b ? (unsigned)(((size_t)(k0 + 16 * j) * p.ldb + col_b) * 4) : OOB);
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            *reinterpret_cast<f32x4*>(&As[(krow + 16 * j) * TN_LD + mchunk * 4]) = ra[j];
            *reinterpret_cast<f32x4*>(&Bs[(krow + 16 * j) * TN_LD + mchunk * 4]) = rb[j];
        }
    };
    const int lr = lane & 31, lh = lane >> 5;
    const int a_rd = lh * TN_LD + wm * 32 + lr, b_rd = lh * TN_LD + wn * 32 + lr;
    if (KT > 0) {
        issue(0);
        store();
    }
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const bool more = kt + 1 < KT;
        if (more) issue(kt + 1);
#pragma unroll
        for (int q = 0; q < 16; ++q)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[a_rd + 2 * q * TN_LD], Bs[b_rd + 2 * q * TN_LD], acc, 0, 0, 0);
        __syncthreads();
        if (more) {
            store();
            __syncthreads();
        }
    }
    float* out = p.out + ((size_t)bz * p.ksplit + by) * p.M * p.N;
    const int n = n0 + wn * 32 + lr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < p.M && n < p.N) out[(size_t)m * p.N + n] = acc[r];
    }
}

// Split-K partial planes Zs[plane][split][Co*Ci] summed in place into split 0, in split order: one thread per (plane, 4 entries).  The
// finish kernel below has only Co*Ci/4 threads -- 16 blocks for a 128 x 128 filter -- so with many splits the sum is taken here, wide, first.
__global__ __launch_bounds__(256) void wino_splitk_sum_kernel(float* __restrict__ Zs, size_t total, int P, int ksplit) {
    const size_t quads = total >> 2, n = quads * P;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (size_t)gridDim.x * 256) {
        const size_t pl = idx / quads, q = idx - pl * quads;
        float* src = Zs + pl * ksplit * total + q * 4;
        f32x4 sum = *reinterpret_cast<const f32x4*>(src);
        for (int k = 1; k < ksplit; ++k) sum += *reinterpret_cast<const f32x4*>(src + (size_t)k * total);
        *reinterpret_cast<f32x4*>(src) = sum;
    }
}

// four neighbouring (co, ci) entries per thread: 16-byte loads of the 36 x ksplit partial planes (plane stride pstride floats), 144
// contiguous output bytes
__global__ __launch_bounds__(256) void wino4_wgrad_finish_kernel(const float* __restrict__ Zs, float* __restrict__ dw, int Co, int Ci, int ksplit,
                                                                 size_t pstride, int nfin, const BiasTail bt) {
    if ((int)blockIdx.x >= nfin) {                                     // uniform: the bias gradient's final sums (blocks nfin ...)
        colsum_final4_body(bt.part, bt.db, bt.nblk, bt.C, bt.ldp, (int)blockIdx.x - nfin);
        return;
    }
    const size_t total = (size_t)Co * Ci, quads = total >> 2;          // Ci % 4 == 0
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < quads; q += (size_t)nfin * 256) {
        const size_t i = q * 4;
        f32x4 z[6][6];                                       // all 36 planes in flight at once, then the remaining K slices in order
#pragma unroll
        for (int pl = 0; pl < 36; ++pl) z[pl / 6][pl % 6] = *reinterpret_cast<const f32x4*>(Zs + (size_t)pl * pstride + i);
        for (int k = 1; k < ksplit; ++k)
#pragma unroll
            for (int pl = 0; pl < 36; ++pl) z[pl / 6][pl % 6] += *reinterpret_cast<const f32x4*>(Zs + (size_t)pl * pstride + (size_t)k * total + i);
        f32x4 t[3][6];                                       // G^T Z
#pragma unroll
        for (int b = 0; b < 6; ++b)
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int a = 0; a < 6; ++a) acc += W4_G[a][r] * z[a][b];
                t[r][b] = acc;
            }
        float flat[36];                                      // [entry][3x3 tap], as the four OIHW entries lie in memory
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int b = 0; b < 6; ++b) acc += t[r][b] * W4_G[b][s2];
#pragma unroll
                for (int e = 0; e < 4; ++e) flat[e * 9 + r * 3 + s2] = acc[e];
            }
        f32x4* o = reinterpret_cast<f32x4*>(dw + i * 9);
#pragma unroll
        for (int j = 0; j < 9; ++j) o[j] = f32x4{flat[4 * j], flat[4 * j + 1], flat[4 * j + 2], flat[4 * j + 3]};
    }
}

int g_fused = -1;                  // ssd_tune_set_wino_fused: -1 automatic, 0 never, 1 wherever K % 64 == 0
int g_full = -1;                   // ssd_tune_set_wino_full: -1 automatic, 0 never, 1 wherever K is 64 or 128
// input transform + GEMMs + output transform in one kernel (wino_fused.hip: wino4_full_kernel)?
inline bool use_full(int mo, int K, int Nout) {
    if (mo != 4 || (K != 64 && K != 128) || g_full == 0 || g_fused == 0) return false;
    // Measured at batch 32 (tools/wino_bench.py, wf_stamps.py full): conv1_2 forward 1.62 ms against 1.16 for input transform + fused
    // kernel.  Its B fragments come straight from L2 (the V pass fills the LDS): 16 tiles per workgroup need 32 B/cycle/CU of filter
    // stream and run L2-bound at half the MFMA rate; 32 tiles need the K passes, whose transform registers push the 144 accumulators
    // to scratch.  Off unless forced; kept as the correct, tested starting point for an LDS-DMA filter ring beside a 16-channel V pass.
    return g_full == 1;
}
// The 36 plane GEMMs of an F(4x4) layer on the bf16 MFMA from three exact bf16 limbs per operand (csrc/gemm_x3.hip) instead of the f32 MFMA:
// long reductions only (K >= 256: the layers whose GEMMs are MFMA-bound; at K <= 128 the fused GEMM + output transform kernel wins and
// the planes, not the MFMA, bound the layer).  The filters of such a layer are kept as limb planes (store_u), so the rule must not
// change between a layer's filter transform and its convolutions: ops.py checks the filter tensor's dtype against it on every call.
int g_wino_x3 = -1;                // ssd_tune_set_wino_x3: -1 = environment SSD_WINO_X3 (default on), 0 off, 1 on
inline bool use_x3(int mo, int K) {
    if (g_wino_x3 < 0) {
        const char* e = getenv("SSD_WINO_X3");
        g_wino_x3 = (e != nullptr && e[0] == '0') ? 0 : 1;
    }
    return g_wino_x3 == 1 && mo == 4 && K >= 256 && K % 32 == 0;      // (the limb GEMM walks K in pairs of 16-deep steps)
}
inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
// GEMMs + output transform in one kernel (wino_fused.hip)?  It removes the write and the read-back of the M planes; what it costs is
// MFMA efficiency on long reductions (one workgroup per CU, 16x16x4 MFMAs).
inline bool use_fused(int mo, int K, int Nout) {
    if (mo != 4 || (K != 64 && K != 128 && K != 256) || g_fused == 0 || use_x3(mo, K)) return false;
    if (g_fused == 1) return true;
    return K <= 128;                   // measured at batch 32 (tools/wino_bench.py): conv1_2 forward 1.52 -> 1.14 ms, dgrad 1.26 -> 0.94, conv2_1 forward 0.67 -> 0.50
}
int g_xform_cap = 8192;            // ssd_tune_set_wino_xform_blocks: most blocks a transform kernel is launched with (grid-stride loops do the rest)
inline int grid_for(size_t total) {
    const size_t b = (total + 255) / 256, cap = (size_t)g_xform_cap;
    return (int)(b > cap ? cap : (b == 0 ? 1 : b));
}

// tile grid of an H x W map under dilation D (F(4x4) only): TH x TW tile rows / columns through all D x D sub-lattices
struct LatPlan { Lat lat; int TH, TW; };
LatPlan lat_plan(int H, int W, int D) {
    LatPlan p{};
    p.lat.D = D;
    int th = 0, tw = 0;
    for (int a = 0; a < 4; ++a) {
        p.lat.h0[a] = th;
        p.lat.w0[a] = tw;
        if (a < D) {
            const int sh = a < H ? (H - a + D - 1) / D : 0, sw = a < W ? (W - a + D - 1) / D : 0;       // rows / columns of sub-lattice a
            th += (sh + 3) / 4;
            tw += (sw + 3) / 4;
        }
    }
    p.TH = th; p.TW = tw;
    return p;
}

// one launch of the dy pass: bias_blocks > 0 = with the bias partial sums on a grid of exactly that many blocks; ps = pooled source or NULL
void launch_dy_pass(const float* dy, float* Y, const ssd_conv_geom* g, int ldy, const LatPlan& lp, float* part, int bias_blocks, float* Vd,
                    const PoolSrc* ps, hipStream_t st) {
    const int TH = lp.TH, TW = lp.TW;
    const size_t tiles = (size_t)g->N * TH * TW;
    const PoolSrc none{nullptr, nullptr, 0, 0};
    if (bias_blocks > 0) {
        if (ps) hipLaunchKernelGGL((wino4_dy_kernel<true, true>), dim3(bias_blocks), dim3(256), 0, st, dy, Y, g->N, g->H, g->W, ldy, TH, TW, part, g->Co, Vd, *ps, lp.lat);
        else hipLaunchKernelGGL((wino4_dy_kernel<true, false>), dim3(bias_blocks), dim3(256), 0, st, dy, Y, g->N, g->H, g->W, ldy, TH, TW, part, g->Co, Vd, none, lp.lat);
    } else {
        const dim3 grid(grid_for(tiles * (ldy / 4)));
        if (ps) hipLaunchKernelGGL((wino4_dy_kernel<false, true>), grid, dim3(256), 0, st, dy, Y, g->N, g->H, g->W, ldy, TH, TW, static_cast<float*>(nullptr), 0, Vd, *ps, lp.lat);
        else hipLaunchKernelGGL((wino4_dy_kernel<false, false>), grid, dim3(256), 0, st, dy, Y, g->N, g->H, g->W, ldy, TH, TW, static_cast<float*>(nullptr), 0, Vd, none, lp.lat);
    }
}

// one Winograd convolution, F(mo x mo, 3x3) with mo = 2 or 4: in (N,H,W,Cin) -> out (N,H,W,ldo) first Cout channels
struct PooledOut { float* y; uint8_t* argmax; int Ho, Wo; };     // destination of the fused conv -> ReLU -> 2x2/s2 max pool form
int wino_conv(int mo, const float* in, int Cin, const float* U, int U_rows, float* out, int ldo, int Cout, const float* bias,
              const float* mask, int relu, int accumulate, int N, int H, int W, void* ws, size_t ws_bytes, hipStream_t st,
              const PooledOut* pooled = nullptr, float* V_keep = nullptr, const float* V_given = nullptr,
              unsigned long long* bits_out = nullptr, const unsigned long long* mask_bits = nullptr, int D = 1) {
    if (D < 1 || D > 4 || (D > 1 && (mo != 4 || pooled != nullptr))) return SSD_ERR_BAD_SHAPE;      // dilation: F(4x4), not into a pool
    const LatPlan lp = lat_plan(H, W, D);
    const int TH = mo == 4 ? lp.TH : (H + mo - 1) / mo, TW = mo == 4 ? lp.TW : (W + mo - 1) / mo, P = (mo + 2) * (mo + 2);
    const size_t tiles = (size_t)N * TH * TW;
    const int Cvalid = Cout;
    Cout = (Cout + 3) / 4 * 4;                 // the GEMM and the output transform work on whole 4-channel vectors: the filter rows
    if (tiles >= (1ull << 31) || Cin % 32 != 0 || Cout > ldo) return SSD_ERR_BAD_SHAPE;     // beyond Cvalid read as zero
    const size_t vb = align256((size_t)P * tiles * Cin * 4), mb = align256((size_t)P * tiles * Cout * 4);
    if (ws_bytes < vb + mb) return SSD_ERR_WORKSPACE;
    float* V = V_keep != nullptr ? V_keep : static_cast<float*>(ws);       // kept planes: the weight gradient multiplies them again
    float* Mx = reinterpret_cast<float*>(static_cast<char*>(ws) + vb);
    if (D == 1 && V_given == nullptr && use_full(mo, Cin, Cout))            // one kernel from the activation to the output
        return ssd_internal_wino4_full(in, U, (int)tiles, Cin, U_rows, Cout, out, ldo, Cvalid, bias, mask, mask_bits, relu, accumulate, H, W, TH,
                                       TW, pooled ? pooled->y : nullptr, pooled ? pooled->argmax : nullptr, pooled ? pooled->Ho : 0,
                                       pooled ? pooled->Wo : 0, V_keep, bits_out, st);
    if (V_given != nullptr) V = const_cast<float*>(V_given);                // input planes already formed (by the dy pass of the wgrad)
    else if (mo == 2) hipLaunchKernelGGL(wino_input_kernel, dim3(grid_for(tiles * (Cin / 4))), dim3(256), 0, st, in, V, N, H, W, Cin, TH, TW);
    else hipLaunchKernelGGL(wino4_input_kernel, dim3(grid_for(tiles * (Cin / 4))), dim3(256), 0, st, in, V, N, H, W, Cin, TH, TW, bits_out, lp.lat);
    SSD_CHECK_LAUNCH();
    if (mask_bits != nullptr && mo != 4) return SSD_ERR_BAD_SHAPE;
    if (D == 1 && use_fused(mo, Cin, Cout))
        return ssd_internal_wino4_gemm_out(V, U, (int)tiles, Cin, U_rows, Cout, out, ldo, Cvalid, bias, mask, mask_bits, relu, accumulate, H, W, TH, TW,
                                           pooled ? pooled->y : nullptr, pooled ? pooled->argmax : nullptr, pooled ? pooled->Ho : 0,
                                           pooled ? pooled->Wo : 0, st);
    if (use_x3(mo, Cin)) {
        if (int e = ssd_internal_gemm_batched_x3(V, U, Mx, (int)tiles, Cin, Cout, U_rows, P, tiles * Cin, st)) return e;
    } else if (int e = ssd_internal_gemm_batched(V, U, Mx, (int)tiles, Cin, Cout, U_rows, P, tiles * Cin, (size_t)U_rows * Cin, 1, st)) {
        return e;
    }
    if (pooled != nullptr)
        hipLaunchKernelGGL(wino4_output_pool_kernel, dim3(grid_for(tiles * (Cout / 4))), dim3(256), 0, st, Mx, pooled->y, pooled->argmax, N, H, W,
                           Cout, TH, TW, bias, pooled->Ho, pooled->Wo);
    else if (mo == 2)
        hipLaunchKernelGGL(wino_output_kernel, dim3(grid_for(tiles * (Cout / 4))), dim3(256), 0, st, Mx, out, N, H, W, Cout, Cvalid, ldo, TH, TW,
                           bias, mask, relu, accumulate);
    else
        hipLaunchKernelGGL(wino4_output_kernel, dim3(grid_for(tiles * (Cout / 4))), dim3(256), 0, st, Mx, out, N, H, W, Cout, Cvalid, ldo, TH,
                           TW, bias, mask, relu, accumulate, mask_bits, lp.lat);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

bool wino_geom_ok(const ssd_conv_geom* g) {
    return g && g->R == 3 && g->S == 3 && g->stride == 1 && g->dil >= 1 && g->dil <= 4 && g->pad == g->dil && g->Ho == g->H && g->Wo == g->W && g->N > 0 &&
           g->H > 0 && g->W > 0 && g->Ci > 0 && g->Co > 0;
}

}  // namespace

// mo = 2: F(2x2,3x3), P = 16 planes; mo = 4: F(4x4,3x3), P = 36 planes.  U_fwd: [P][Co][Ci]; U_bwd: [P][Ci][Co_pad] (either may be NULL)
extern "C" int ssd_wino_weights(const float* w_oihw, float* U_fwd, float* U_bwd, int Co, int Ci, int Co_pad, int mo, void* stream) {
    if (!w_oihw || (!U_fwd && !U_bwd)) return SSD_ERR_NULL;
    if (Co <= 0 || Ci <= 0 || Co_pad < Co || (mo != 2 && mo != 4)) return SSD_ERR_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (U_fwd) {
        if (mo == 2) hipLaunchKernelGGL(wino_weight_kernel, dim3(grid_for((size_t)Co * Ci)), dim3(256), 0, st, w_oihw, U_fwd, Co, Ci, Co, Ci, 0);
        else hipLaunchKernelGGL(wino4_weight_kernel, dim3(grid_for((size_t)Co * Ci)), dim3(256), 0, st, w_oihw, U_fwd, Co, Ci, Co, Ci, 0, (int)use_x3(4, Ci));
        SSD_CHECK_LAUNCH();
    }
    if (U_bwd) {
        if (mo == 2) hipLaunchKernelGGL(wino_weight_kernel, dim3(grid_for((size_t)Ci * Co_pad)), dim3(256), 0, st, w_oihw, U_bwd, Co, Ci, Ci, Co_pad, 1);
        else hipLaunchKernelGGL(wino4_weight_kernel, dim3(grid_for((size_t)Ci * Co_pad)), dim3(256), 0, st, w_oihw, U_bwd, Co, Ci, Ci, Co_pad, 1, (int)use_x3(4, Co_pad));
        SSD_CHECK_LAUNCH();
    }
    return SSD_OK;
}

extern "C" size_t ssd_conv3x3_wino_workspace(const ssd_conv_geom* g, int direction, int mo) {
    if (!wino_geom_ok(g) || (mo != 2 && mo != 4)) return 0;
    if (mo == 2 && g->dil != 1) return 0;
    const LatPlan lp = lat_plan(g->H, g->W, g->dil);
    const size_t tiles = mo == 4 ? (size_t)g->N * lp.TH * lp.TW : (size_t)g->N * ((g->H + 1) / 2) * ((g->W + 1) / 2), P = (size_t)(mo + 2) * (mo + 2);
    const int co_pad = (g->Co + 31) / 32 * 32;
    const size_t cin = direction == 0 ? g->Ci : co_pad, cout = direction == 0 ? (size_t)(g->Co + 3) / 4 * 4 : (size_t)(g->Ci + 3) / 4 * 4;
    return align256(P * tiles * cin * 4) + align256(P * tiles * cout * 4);
}

extern "C" int ssd_conv3x3_wino_fwd(const float* x, const float* U_fwd, const float* bias, float* y, int ldy, const ssd_conv_geom* g,
                                    int relu, int mo, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !U_fwd || !y || !workspace) return SSD_ERR_NULL;
    if (!wino_geom_ok(g) || g->Ci % 32 != 0 || ldy < (g->Co + 3) / 4 * 4 || (mo != 2 && mo != 4)) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(x) || !ssd_aligned16(y) || !ssd_aligned16(workspace) || !ssd_aligned16(U_fwd) || ldy % 4 != 0) return SSD_ERR_ALIGN;
    return wino_conv(mo, x, g->Ci, U_fwd, g->Co, y, ldy, g->Co, bias, nullptr, relu, 0, g->N, g->H, g->W, workspace, workspace_bytes,
                     (hipStream_t)stream, nullptr, nullptr, nullptr, nullptr, nullptr, g->dil);
}

extern "C" int ssd_conv3x3_wino_fwd_keep(const float* x, const float* U_fwd, const float* bias, float* y, int ldy, const ssd_conv_geom* g,
                                         int relu, float* planes_keep, void* workspace, size_t workspace_bytes, void* stream) {
    return ssd_conv3x3_wino_fwd_keep_bits(x, U_fwd, bias, y, ldy, g, relu, planes_keep, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int ssd_conv3x3_wino_fwd_keep_bits(const float* x, const float* U_fwd, const float* bias, float* y, int ldy, const ssd_conv_geom* g,
                                              int relu, float* planes_keep, uint64_t* relu_bits_out, void* workspace, size_t workspace_bytes,
                                              void* stream) {
    if (!x || !U_fwd || !y || !workspace || !planes_keep) return SSD_ERR_NULL;
    if (relu_bits_out && ((uintptr_t)relu_bits_out & 7)) return SSD_ERR_ALIGN;
    if (!wino_geom_ok(g) || g->Ci % 32 != 0 || ldy < (g->Co + 3) / 4 * 4) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(x) || !ssd_aligned16(y) || !ssd_aligned16(workspace) || !ssd_aligned16(U_fwd) || !ssd_aligned16(planes_keep) || ldy % 4 != 0)
        return SSD_ERR_ALIGN;
    return wino_conv(4, x, g->Ci, U_fwd, g->Co, y, ldy, g->Co, bias, nullptr, relu, 0, g->N, g->H, g->W, workspace, workspace_bytes,
                     (hipStream_t)stream, nullptr, planes_keep, nullptr, reinterpret_cast<unsigned long long*>(relu_bits_out), nullptr, g->dil);
}

extern "C" int ssd_conv3x3_wino_fwd_pool(const float* x, const float* U_fwd, const float* bias, float* y_pooled, uint8_t* argmax,
                                         const ssd_conv_geom* g, int ceil_mode, float* planes_keep, void* workspace, size_t workspace_bytes,
                                         void* stream) {
    return ssd_conv3x3_wino_fwd_pool_bits(x, U_fwd, bias, y_pooled, argmax, g, ceil_mode, planes_keep, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int ssd_conv3x3_wino_fwd_pool_bits(const float* x, const float* U_fwd, const float* bias, float* y_pooled, uint8_t* argmax,
                                              const ssd_conv_geom* g, int ceil_mode, float* planes_keep, uint64_t* relu_bits_out,
                                              void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !U_fwd || !y_pooled || !workspace) return SSD_ERR_NULL;
    if (relu_bits_out && ((uintptr_t)relu_bits_out & 7)) return SSD_ERR_ALIGN;
    if (!wino_geom_ok(g) || g->dil != 1 || g->Ci % 32 != 0 || g->Co % 4 != 0) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(x) || !ssd_aligned16(y_pooled) || !ssd_aligned16(workspace) || !ssd_aligned16(U_fwd) || (bias && !ssd_aligned16(bias)) ||
        (argmax && ((uintptr_t)argmax & 3)) || (planes_keep && !ssd_aligned16(planes_keep)))
        return SSD_ERR_ALIGN;
    const PooledOut po = {y_pooled, argmax, ceil_mode ? (g->H + 1) / 2 : g->H / 2, ceil_mode ? (g->W + 1) / 2 : g->W / 2};
    if (po.Ho <= 0 || po.Wo <= 0) return SSD_ERR_BAD_SHAPE;
    return wino_conv(4, x, g->Ci, U_fwd, g->Co, nullptr, g->Co, g->Co, bias, nullptr, 1, 0, g->N, g->H, g->W, workspace, workspace_bytes,
                     (hipStream_t)stream, &po, planes_keep, nullptr, reinterpret_cast<unsigned long long*>(relu_bits_out));
}

// Forward of an F(4x4) layer whose input planes already exist (written by the producer of its input: ssd_conv1_first_wino_fwd): the plane
// GEMMs + output transform only.  y_pooled != NULL: conv -> ReLU -> 2x2 / stride-2 max pool (ssd_conv3x3_wino_fwd_pool's outputs), else y
// (N,H,W,ldy) with bias (+ ReLU).
extern "C" int ssd_conv3x3_wino_fwd_from_planes(const float* planes, const float* U_fwd, const float* bias, float* y, int ldy, float* y_pooled,
                                                uint8_t* argmax, const ssd_conv_geom* g, int relu, int ceil_mode, void* workspace,
                                                size_t workspace_bytes, void* stream) {
    if (!planes || !U_fwd || !workspace || (!y && !y_pooled)) return SSD_ERR_NULL;
    if (!wino_geom_ok(g) || g->dil != 1 || g->Ci % 32 != 0 || g->Co % 4 != 0) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(planes) || !ssd_aligned16(U_fwd) || !ssd_aligned16(workspace) || (y && !ssd_aligned16(y)) || (y_pooled && !ssd_aligned16(y_pooled)) ||
        (bias && !ssd_aligned16(bias)) || (argmax && ((uintptr_t)argmax & 3)))
        return SSD_ERR_ALIGN;
    if (y_pooled) {
        const PooledOut po = {y_pooled, argmax, ceil_mode ? (g->H + 1) / 2 : g->H / 2, ceil_mode ? (g->W + 1) / 2 : g->W / 2};
        if (po.Ho <= 0 || po.Wo <= 0) return SSD_ERR_BAD_SHAPE;
        return wino_conv(4, nullptr, g->Ci, U_fwd, g->Co, nullptr, g->Co, g->Co, bias, nullptr, 1, 0, g->N, g->H, g->W, workspace, workspace_bytes,
                         (hipStream_t)stream, &po, nullptr, planes);
    }
    if (ldy < g->Co || ldy % 4 != 0) return SSD_ERR_BAD_SHAPE;
    return wino_conv(4, nullptr, g->Ci, U_fwd, g->Co, y, ldy, g->Co, bias, nullptr, relu, 0, g->N, g->H, g->W, workspace, workspace_bytes,
                     (hipStream_t)stream, nullptr, nullptr, planes);
}

extern "C" int ssd_conv3x3_wino_dgrad(const float* dy, int ldy, const float* U_bwd, int Co_pad, float* dx, const float* relu_mask,
                                      int accumulate, const ssd_conv_geom* g, int mo, void* workspace, size_t workspace_bytes,
                                      void* stream) {
    if (!dy || !U_bwd || !dx || !workspace) return SSD_ERR_NULL;
    if (!wino_geom_ok(g) || Co_pad % 32 != 0 || Co_pad < g->Co || ldy != Co_pad || g->Ci % 4 != 0 || (mo != 2 && mo != 4))
        return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(dy) || !ssd_aligned16(dx) || !ssd_aligned16(workspace) || !ssd_aligned16(U_bwd) ||
        (relu_mask && !ssd_aligned16(relu_mask)))
        return SSD_ERR_ALIGN;
    return wino_conv(mo, dy, Co_pad, U_bwd, g->Ci, dx, g->Ci, g->Ci, nullptr, relu_mask, 0, accumulate, g->N, g->H, g->W, workspace,
                     workspace_bytes, (hipStream_t)stream, nullptr, nullptr, nullptr, nullptr, nullptr, g->dil);
}

namespace {
constexpr int COLSUM_BLOCKS = 512;
constexpr int DY_BIAS_BLOCKS = 2048;      // >= COLSUM_BLOCKS: rows of the bias-gradient partial buffer
int g_wgrad_tn = 1;               // F(4x4) weight gradient on untransposed planes + the TN GEMM (0: transposed planes + the NT GEMM)
struct WinoWgradPlan { int TH, TW, Tpad, ks, cdy, P, x3; size_t tiles, yb, vb, zb, pb; LatPlan lp; };
// The 36 TN GEMMs dU = dY^T V of an F(4x4) layer from three bf16 limbs per operand (gemm_x3.hip, 128 x 128 tiles): where both channel
// counts fill the tiles (padding <= 25 %: the trunk from conv2_2 on and fc6; the heads' 100 / 150 rows stay on the 64 x 64 f32 kernel)
inline bool use_x3_tn(const ssd_conv_geom* g, int mo) {
    return use_x3(mo, 256) && g_wgrad_tn && g->Co >= 128 && g->Ci >= 128 && (g->Co + 127) / 128 * 128 * 4 <= g->Co * 5 &&
           (g->Ci + 127) / 128 * 128 * 4 <= g->Ci * 5;
}
WinoWgradPlan wino_wgrad_plan(const ssd_conv_geom* g, int ldy, int mo) {
    WinoWgradPlan w;
    w.x3 = use_x3_tn(g, mo) ? 1 : 0;
    w.P = (mo + 2) * (mo + 2);
    w.lp = lat_plan(g->H, g->W, mo == 4 ? g->dil : 1);
    w.TH = mo == 4 ? w.lp.TH : (g->H + mo - 1) / mo; w.TW = mo == 4 ? w.lp.TW : (g->W + mo - 1) / mo;
    w.tiles = (size_t)g->N * w.TH * w.TW;
    w.Tpad = (int)((w.tiles + 31) / 32 * 32);
    w.cdy = ldy;
    const int bp = ((g->Co + 63) / 64) * ((g->Ci + 63) / 64) * w.P;      // 64x64 output tiles over the planes
    int ks = (1536 + bp - 1) / bp;                                         // fill the ~1500 resident-block slots
    const int ksteps = w.Tpad / 32;
    if (ks > ksteps / 8) ks = ksteps / 8;
    if (ks < 1) ks = 1;
    int per = (ksteps + ks - 1) / ks;
    w.ks = (ksteps + per - 1) / per;
    // Output tiles that share panels run as one group per XCD (wino_gemm_tn_kernel): keep planes x slices a multiple of 8 (36 planes:
    // an even slice count) so that every XCD gets the same number of groups.
    if (bp > w.P && mo == 4 && (w.ks & 1) && ksteps >= 2 * (w.ks + 1)) {
        per = (ksteps + w.ks) / (w.ks + 1);
        const int k2 = (ksteps + per - 1) / per;
        if ((k2 & 1) == 0) w.ks = k2;
    }
    if (w.x3) {
        // one round of resident blocks (3 per CU): as many K slices as keep the grid within 768 blocks, an even count (partial sums
        // are written and read back: no more slices than that), at least 8 steps of 16 tiles each.  (Round 4, tools/grid_audit.py: the
        // 512 x 512-channel layers are 576 blocks = three quarters of a round; 2 / 4 slices -- 1 152 / 2 304 blocks, the latter three exact
        // rounds -- measured: conv4 0.40 -> 0.37 / 0.39 ms, conv5 0.136 -> 0.144 / 0.183, the step 20.07 -> 20.15 / 20.25: not taken.)
        const int bp128 = ((g->Co + 127) / 128) * ((g->Ci + 127) / 128) * w.P, steps16 = (int)((w.tiles + 15) / 16);
        int k3 = 768 / bp128;
        if (k3 > 1) k3 &= ~1;
        if (k3 > steps16 / 8) k3 = steps16 / 8;
        if (k3 < 1) k3 = 1;
        const int per3 = (steps16 + k3 - 1) / k3;
        w.ks = (steps16 + per3 - 1) / per3;
    }
    w.yb = align256((size_t)w.P * ldy * w.Tpad * 4);
    w.vb = align256((size_t)w.P * g->Ci * w.Tpad * 4);
    w.zb = align256((size_t)w.P * w.ks * g->Co * g->Ci * 4);
    w.pb = align256((size_t)DY_BIAS_BLOCKS * ldy * 4);
    return w;
}
// the split-K TN GEMMs of the F(4x4) weight gradient: Zs[plane][slice][Co][Ci] = (slice of) Y[plane]^T V[plane]
int launch_wgrad_tn(const float* Y, const float* V, float* Zs, const ssd_conv_geom* g, int ldy, const WinoWgradPlan& w, hipStream_t st) {
    if (w.x3) {
        const int steps16 = (int)((w.tiles + 15) / 16);
        return ssd_internal_gemm_tn_x3(Y, V, Zs, g->Co, g->Ci, (int)w.tiles, ldy, g->Ci, w.P, w.ks, (steps16 + w.ks - 1) / w.ks, w.tiles * ldy,
                                       w.tiles * g->Ci, st);
    }
    TnParams q;
    q.a = Y; q.b = V; q.out = Zs;
    q.M = g->Co; q.N = g->Ci; q.K = (int)w.tiles; q.lda = ldy; q.ldb = g->Ci;
    q.tiles_m = (g->Co + 63) / 64; q.tiles_n = (g->Ci + 63) / 64;
    const int ksteps = (q.K + 31) / 32;
    q.ksplit = w.ks;
    q.ksteps_per_split = (ksteps + w.ks - 1) / w.ks;
    q.batch_a = w.tiles * ldy; q.batch_b = w.tiles * g->Ci;
    if (q.batch_a * 4 >= 0xFFFFFFF0ull || q.batch_b * 4 >= 0xFFFFFFF0ull) return SSD_ERR_BAD_SHAPE;
    q.a_bytes = (unsigned)(q.batch_a * 4); q.b_bytes = (unsigned)(q.batch_b * 4);
    q.groups = w.P * w.ks;
    hipLaunchKernelGGL(wino_gemm_tn_kernel, dim3((unsigned)(q.tiles_m * q.tiles_n * ((q.groups + 7) / 8) * 8)), dim3(256), 0, st, q);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}
}  // namespace

extern "C" int ssd_tune_set_wino_x3(int on) {
    g_wino_x3 = on < 0 ? -1 : (on != 0);
    return SSD_OK;
}
extern "C" int ssd_wino_uses_x3(int mo, int K) { return use_x3(mo, K) ? 1 : 0; }

extern "C" int ssd_tune_set_wino_bias_tail(int on) {
    g_bias_tail = on != 0;
    return SSD_OK;
}
extern "C" int ssd_tune_set_wino_xform_blocks(int blocks) {
    if (blocks < 64 || blocks > 65535) return SSD_ERR_BAD_SHAPE;
    g_xform_cap = blocks;
    return SSD_OK;
}

extern "C" int ssd_tune_set_wino_full(int mode) {
    if (mode < -1 || mode > 1) return SSD_ERR_BAD_SHAPE;
#ifndef SSD_EXPERIMENTAL
    if (mode == 1) return SSD_ERR_BAD_SHAPE;       // wino4_full_kernel is not in this build
#endif
    g_full = mode;
    return SSD_OK;
}

// 1 if the forward (direction 0) / data gradient from dy (direction 1) of this geometry runs as ONE kernel from the activation (no planes read)
extern "C" int ssd_conv3x3_wino_uses_full(const ssd_conv_geom* g, int direction) {
    if (!wino_geom_ok(g) || g->dil != 1) return 0;
    const int co_pad = (g->Co + 31) / 32 * 32;
    return direction == 0 ? use_full(4, g->Ci, (g->Co + 3) / 4 * 4) : use_full(4, co_pad, g->Ci);
}

extern "C" int ssd_tune_set_wino_fused(int mode) {
    if (mode < -1 || mode > 1) return SSD_ERR_BAD_SHAPE;
    g_fused = mode;
    return SSD_OK;
}

extern "C" int ssd_tune_set_wino_wgrad_tn(int on) {
    g_wgrad_tn = on ? 1 : 0;
    return SSD_OK;
}

extern "C" size_t ssd_conv3x3_wino_wgrad_workspace(const ssd_conv_geom* g, int ldy, int mo) {
    if (!wino_geom_ok(g) || ldy < g->Co || (mo != 2 && mo != 4)) return 0;
    const WinoWgradPlan w = wino_wgrad_plan(g, ldy, mo);
    return w.yb + w.vb + w.zb + w.pb;
}

// dw (Co,Ci,3,3) OIHW and, if asked, dbias (Co) from x (N,H,W,Ci) and dy (N,H,W,ldy; columns >= Co zero); mo = 2 or 4
namespace {
int wino_wgrad(const float* x, const float* planes, const float* dy, int ldy, float* dw_oihw, float* dbias, const ssd_conv_geom* g, int mo,
               float* dgrad_planes, void* workspace, size_t workspace_bytes, void* stream, const PoolSrc* pool = nullptr);
}
extern "C" int ssd_conv3x3_wino_wgrad(const float* x, const float* dy, int ldy, float* dw_oihw, float* dbias, const ssd_conv_geom* g,
                                      int mo, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x) return SSD_ERR_NULL;
    if (!ssd_aligned16(x)) return SSD_ERR_ALIGN;
    return wino_wgrad(x, nullptr, dy, ldy, dw_oihw, dbias, g, mo, nullptr, workspace, workspace_bytes, stream);
}
extern "C" int ssd_conv3x3_wino_wgrad_planes(const float* planes, const float* dy, int ldy, float* dw_oihw, float* dbias,
                                             const ssd_conv_geom* g, float* dgrad_planes_out, void* workspace, size_t workspace_bytes,
                                             void* stream) {
    if (!planes) return SSD_ERR_NULL;
    if (!ssd_aligned16(planes) || (dgrad_planes_out && !ssd_aligned16(dgrad_planes_out))) return SSD_ERR_ALIGN;
    return wino_wgrad(nullptr, planes, dy, ldy, dw_oihw, dbias, g, 4, dgrad_planes_out, workspace, workspace_bytes, stream);
}
extern "C" int ssd_conv3x3_wino_dgrad_bits(const float* dy, int ldy, const float* U_bwd, int Co_pad, float* dx, const uint64_t* relu_bits,
                                           int accumulate, const ssd_conv_geom* g, void* workspace, size_t workspace_bytes, void* stream) {
    if (!dy || !U_bwd || !dx || !workspace || !relu_bits) return SSD_ERR_NULL;
    if (!wino_geom_ok(g) || Co_pad % 32 != 0 || Co_pad < g->Co || ldy != Co_pad || g->Ci % 4 != 0) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(dy) || !ssd_aligned16(dx) || !ssd_aligned16(workspace) || !ssd_aligned16(U_bwd) || ((uintptr_t)relu_bits & 7))
        return SSD_ERR_ALIGN;
    return wino_conv(4, dy, Co_pad, U_bwd, g->Ci, dx, g->Ci, g->Ci, nullptr, nullptr, 0, accumulate, g->N, g->H, g->W, workspace, workspace_bytes,
                     (hipStream_t)stream, nullptr, nullptr, nullptr, nullptr, reinterpret_cast<const unsigned long long*>(relu_bits), g->dil);
}
extern "C" int ssd_conv3x3_wino_dgrad_planes_bits(const float* dy_planes, const float* U_bwd, int Co_pad, float* dx, const uint64_t* relu_bits,
                                                  int accumulate, const ssd_conv_geom* g, void* workspace, size_t workspace_bytes,
                                                  void* stream) {
    if (!dy_planes || !U_bwd || !dx || !workspace || !relu_bits) return SSD_ERR_NULL;
    if (!wino_geom_ok(g) || Co_pad % 32 != 0 || Co_pad < g->Co || g->Ci % 4 != 0) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(dy_planes) || !ssd_aligned16(dx) || !ssd_aligned16(workspace) || !ssd_aligned16(U_bwd) || ((uintptr_t)relu_bits & 7))
        return SSD_ERR_ALIGN;
    return wino_conv(4, nullptr, Co_pad, U_bwd, g->Ci, dx, g->Ci, g->Ci, nullptr, nullptr, 0, accumulate, g->N, g->H, g->W, workspace,
                     workspace_bytes, (hipStream_t)stream, nullptr, nullptr, dy_planes, nullptr,
                     reinterpret_cast<const unsigned long long*>(relu_bits), g->dil);
}
extern "C" int ssd_conv3x3_wino_dgrad_planes(const float* dy_planes, const float* U_bwd, int Co_pad, float* dx, const float* relu_mask,
                                             int accumulate, const ssd_conv_geom* g, void* workspace, size_t workspace_bytes, void* stream) {
    if (!dy_planes || !U_bwd || !dx || !workspace) return SSD_ERR_NULL;
    if (!wino_geom_ok(g) || Co_pad % 32 != 0 || Co_pad < g->Co || g->Ci % 4 != 0) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(dy_planes) || !ssd_aligned16(dx) || !ssd_aligned16(workspace) || !ssd_aligned16(U_bwd) ||
        (relu_mask && !ssd_aligned16(relu_mask)))
        return SSD_ERR_ALIGN;
    return wino_conv(4, nullptr, Co_pad, U_bwd, g->Ci, dx, g->Ci, g->Ci, nullptr, relu_mask, 0, accumulate, g->N, g->H, g->W, workspace,
                     workspace_bytes, (hipStream_t)stream, nullptr, nullptr, dy_planes, nullptr, nullptr, g->dil);
}
namespace {
int wino_wgrad(const float* x, const float* planes, const float* dy, int ldy, float* dw_oihw, float* dbias, const ssd_conv_geom* g, int mo,
               float* dgrad_planes, void* workspace, size_t workspace_bytes, void* stream, const PoolSrc* pool) {
    if (!dy || !dw_oihw || !workspace) return SSD_ERR_NULL;
    if (!wino_geom_ok(g) || g->Ci % 4 != 0 || ldy % 4 != 0 || ldy < g->Co || (mo != 2 && mo != 4)) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(dy) || !ssd_aligned16(workspace) || !ssd_aligned16(dw_oihw)) return SSD_ERR_ALIGN;
    const WinoWgradPlan w = wino_wgrad_plan(g, ldy, mo);
    if (w.tiles >= (1ull << 31)) return SSD_ERR_BAD_SHAPE;
    if (workspace_bytes < w.yb + w.vb + w.zb + w.pb) return SSD_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* base = static_cast<char*>(workspace);
    float* Yt = reinterpret_cast<float*>(base);
    float* Vt = reinterpret_cast<float*>(base + w.yb);
    float* Zs = reinterpret_cast<float*>(base + w.yb + w.vb);
    float* part = reinterpret_cast<float*>(base + w.yb + w.vb + w.zb);
    const int gy = (w.Tpad / 32) * ((ldy / 4 + 7) / 8), gx = (w.Tpad / 32) * ((g->Ci / 4 + 7) / 8);
    const dim3 gyd(gy > 16384 ? 16384 : gy), gxd(gx > 16384 ? 16384 : gx);
    int dy_bias_blocks = 0;
    if (dgrad_planes != nullptr && !(mo == 4 && (g_wgrad_tn || planes != nullptr))) return SSD_ERR_BAD_SHAPE;
    if (g->dil != 1 && (pool != nullptr || !(mo == 4 && (g_wgrad_tn || planes != nullptr)))) return SSD_ERR_BAD_SHAPE;   // dilation: the lattice-aware kernels only
    if (mo == 4 && (g_wgrad_tn || planes != nullptr)) {
        // planes in the forward layout [plane][tile][channel]; the GEMM reduces over the tile rows of both.  The x planes are the
        // forward convolution's own B^T d B when the caller kept them.
        const int c4 = ldy / 4;
        if (dbias != nullptr && c4 <= 256) {                 // bias gradient from the same pass over dy
            int gcd = 256, r = c4;
            while (r) { const int tmp = gcd % r; gcd = r; r = tmp; }
            const int unit = c4 / gcd;                       // grid must be a multiple of this
            int blocks = grid_for(w.tiles * c4);
            if (blocks > DY_BIAS_BLOCKS) blocks = DY_BIAS_BLOCKS;
            blocks = (blocks + unit - 1) / unit * unit;
            if (blocks > DY_BIAS_BLOCKS) blocks -= unit;
            dy_bias_blocks = blocks;
            launch_dy_pass(dy, Yt, g, ldy, w.lp, part, blocks, dgrad_planes, pool, st);
        } else
            launch_dy_pass(dy, Yt, g, ldy, w.lp, nullptr, 0, dgrad_planes, pool, st);
        if (planes == nullptr)
            hipLaunchKernelGGL(wino4_input_kernel, dim3(grid_for(w.tiles * (g->Ci / 4))), dim3(256), 0, st, x, Vt, g->N, g->H, g->W, g->Ci, w.TH,
                               w.TW, static_cast<unsigned long long*>(nullptr), w.lp.lat);
        SSD_CHECK_LAUNCH();
        if (int e = launch_wgrad_tn(Yt, planes != nullptr ? planes : Vt, Zs, g, ldy, w, st)) return e;
    } else if (mo == 2) {
        hipLaunchKernelGGL(wino_xform_t_kernel<1>, gyd, dim3(256), 0, st, dy, Yt, g->N, g->H, g->W, ldy, w.TH, w.TW, w.Tpad);
        hipLaunchKernelGGL(wino_xform_t_kernel<0>, gxd, dim3(256), 0, st, x, Vt, g->N, g->H, g->W, g->Ci, w.TH, w.TW, w.Tpad);
    } else {
        hipLaunchKernelGGL(wino4_xform_t_kernel<1>, gyd, dim3(256), 0, st, dy, Yt, g->N, g->H, g->W, ldy, w.TH, w.TW, w.Tpad);
        hipLaunchKernelGGL(wino4_xform_t_kernel<0>, gxd, dim3(256), 0, st, x, Vt, g->N, g->H, g->W, g->Ci, w.TH, w.TW, w.Tpad);
    }
    SSD_CHECK_LAUNCH();
    if (!(mo == 4 && (g_wgrad_tn || planes != nullptr)))
        if (int e = ssd_internal_gemm_batched(Yt, Vt, Zs, g->Co, w.Tpad, g->Ci, g->Ci, w.P, (size_t)ldy * w.Tpad, (size_t)g->Ci * w.Tpad, w.ks, st))
            return e;
    if (mo == 2)
        hipLaunchKernelGGL(wino_wgrad_finish_kernel, dim3(grid_for((size_t)g->Co * g->Ci)), dim3(256), 0, st, Zs, dw_oihw, g->Co, g->Ci, w.ks);
    else
    {
        const size_t total = (size_t)g->Co * g->Ci;
        int ks_left = w.ks;
        if (w.ks > 1 && total / 4 < 65536) {                 // few entries, many splits: sum the splits with a wide grid first
            hipLaunchKernelGGL(wino_splitk_sum_kernel, dim3(grid_for(total / 4 * w.P)), dim3(256), 0, st, Zs, total, w.P, w.ks);
            SSD_CHECK_LAUNCH();
            ks_left = 1;
        }
        const int nfin = grid_for(total / 4), ldp = (g->Co + 3) / 4 * 4;
        const bool tail = dbias && dy_bias_blocks > 0 && g_bias_tail;    // the bias gradient's final sums ride along as extra blocks
        const BiasTail bt{tail ? part : nullptr, dbias, dy_bias_blocks, g->Co, ldp};
        hipLaunchKernelGGL(wino4_wgrad_finish_kernel, dim3(nfin + (tail ? (ldp + 15) / 16 : 0)), dim3(256), 0, st, Zs, dw_oihw, g->Co, g->Ci,
                           ks_left, (size_t)w.ks * total, nfin, bt);
    }
    SSD_CHECK_LAUNCH();
    if (dbias && dy_bias_blocks > 0 && !g_bias_tail) {
        const int ldp = (g->Co + 3) / 4 * 4;
        hipLaunchKernelGGL(colsum_final4_kernel, dim3((ldp + 15) / 16), dim3(256), 0, st, part, dbias, dy_bias_blocks, g->Co, ldp);
        SSD_CHECK_LAUNCH();
    }
    if (dbias && dy_bias_blocks == 0) {                                  // no partial sums from a dy pass (F(2x2), transposed planes, > 1024 channels)
        const size_t M = (size_t)g->N * g->H * g->W;
        int cq = 1;
        while (cq < (g->Co + 3) / 4 && cq < 256) cq <<= 1;
        hipLaunchKernelGGL(colsum_partial_kernel, dim3(COLSUM_BLOCKS), dim3(256), 0, st, dy, part, M, ldy, g->Co, cq);
        SSD_CHECK_LAUNCH();
        hipLaunchKernelGGL(colsum_final_kernel, dim3((g->Co + 15) / 16), dim3(256), 0, st, part, dbias, COLSUM_BLOCKS, g->Co);
        SSD_CHECK_LAUNCH();
    }
    return SSD_OK;
}
}  // namespace


// ---- the weight gradient in two calls: the pass over dy (the data gradient waits for its planes) and the GEMMs + inverse transform
// (nothing in the backward pass waits for them: the caller may enqueue them on another stream) -----------------------------------------
namespace {
int dy_bias_blocks_for(const ssd_conv_geom* g, int ldy, size_t tiles) {
    const int c4 = ldy / 4;
    if (c4 > 256) return 0;
    int gcd = 256, r = c4;
    while (r) { const int tmp = gcd % r; gcd = r; r = tmp; }
    const int unit = c4 / gcd;
    int blocks = grid_for(tiles * c4);
    if (blocks > DY_BIAS_BLOCKS) blocks = DY_BIAS_BLOCKS;
    blocks = (blocks + unit - 1) / unit * unit;
    if (blocks > DY_BIAS_BLOCKS) blocks -= unit;
    if (blocks < unit) blocks = unit;
    return blocks;
}
}  // namespace

extern "C" size_t ssd_wino4_bias_partial_floats(const ssd_conv_geom* g, int ldy) {
    if (!wino_geom_ok(g) || ldy < g->Co || ldy % 4 != 0) return 0;
    return (size_t)DY_BIAS_BLOCKS * ldy;
}

namespace {
int dy_transform(const float* dy, int ldy, const ssd_conv_geom* g, float* wgrad_planes, float* dgrad_planes_out, float* bias_partial,
                 const PoolSrc* pool, void* stream) {
    if (!dy || !wgrad_planes) return SSD_ERR_NULL;
    if (!wino_geom_ok(g) || ldy % 4 != 0 || ldy < g->Co) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(dy) || !ssd_aligned16(wgrad_planes) || (dgrad_planes_out && !ssd_aligned16(dgrad_planes_out)) ||
        (bias_partial && !ssd_aligned16(bias_partial)))
        return SSD_ERR_ALIGN;
    if (pool != nullptr && g->dil != 1) return SSD_ERR_BAD_SHAPE;
    const LatPlan lp = lat_plan(g->H, g->W, g->dil);
    const size_t tiles = (size_t)g->N * lp.TH * lp.TW;
    if (tiles >= (1ull << 31)) return SSD_ERR_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = bias_partial ? dy_bias_blocks_for(g, ldy, tiles) : 0;
    if (bias_partial && blocks == 0) return SSD_ERR_BAD_SHAPE;             // more than 1024 channels: use ssd_conv3x3_wino_wgrad_planes
    launch_dy_pass(dy, wgrad_planes, g, ldy, lp, bias_partial, blocks, dgrad_planes_out, pool, st);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}
}  // namespace

extern "C" int ssd_wino4_dy_transform(const float* dy, int ldy, const ssd_conv_geom* g, float* wgrad_planes, float* dgrad_planes_out,
                                      float* bias_partial, void* stream) {
    return dy_transform(dy, ldy, g, wgrad_planes, dgrad_planes_out, bias_partial, nullptr, stream);
}

namespace {
// the pooled gradient source of the two ..._pooled entry points, checked: the pool must be the 2x2 / stride-2 / no-pad one over this layer's output
int pool_src(const ssd_conv_geom* g, int ldy, const float* dpool, const unsigned char* argmax, const float* y_pooled, int Hp, int Wp, PoolSrc* ps) {
    if (!dpool || !argmax || !y_pooled) return SSD_ERR_NULL;
    if (!wino_geom_ok(g) || g->dil != 1 || ldy != g->Co || Hp < g->H / 2 || Wp < g->W / 2 || Hp <= 0 || Wp <= 0 || 2 * Hp > g->H + 1 || 2 * Wp > g->W + 1)
        return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(y_pooled) || ((uintptr_t)argmax & 3) != 0) return SSD_ERR_ALIGN;
    ps->am = argmax; ps->gate = y_pooled; ps->Hp = Hp; ps->Wp = Wp;
    return SSD_OK;
}
}  // namespace

extern "C" int ssd_wino4_dy_transform_pooled(const float* dpool, const unsigned char* argmax, const float* y_pooled, int Hp, int Wp, int ldy,
                                             const ssd_conv_geom* g, float* wgrad_planes, float* dgrad_planes_out, float* bias_partial,
                                             void* stream) {
    if (!g) return SSD_ERR_NULL;
    PoolSrc ps;
    const int e = pool_src(g, ldy, dpool, argmax, y_pooled, Hp, Wp, &ps);
    return e != SSD_OK ? e : dy_transform(dpool, ldy, g, wgrad_planes, dgrad_planes_out, bias_partial, &ps, stream);
}

extern "C" int ssd_conv3x3_wino_wgrad_planes_pooled(const float* planes, const float* dpool, const unsigned char* argmax, const float* y_pooled,
                                                    int Hp, int Wp, int ldy, float* dw_oihw, float* dbias, const ssd_conv_geom* g,
                                                    float* dgrad_planes_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!planes || !g) return SSD_ERR_NULL;
    if (!ssd_aligned16(planes) || (dgrad_planes_out && !ssd_aligned16(dgrad_planes_out))) return SSD_ERR_ALIGN;
    PoolSrc ps;
    const int e = pool_src(g, ldy, dpool, argmax, y_pooled, Hp, Wp, &ps);
    return e != SSD_OK ? e : wino_wgrad(nullptr, planes, dpool, ldy, dw_oihw, dbias, g, 4, dgrad_planes_out, workspace, workspace_bytes, stream, &ps);
}

extern "C" size_t ssd_wino4_wgrad_gemm_workspace(const ssd_conv_geom* g, int ldy) {
    if (!wino_geom_ok(g) || ldy < g->Co) return 0;
    return wino_wgrad_plan(g, ldy, 4).zb;
}

extern "C" int ssd_wino4_wgrad_gemm(const float* wgrad_planes, const float* x_planes, int ldy, const float* bias_partial, float* dw_oihw,
                                    float* dbias, const ssd_conv_geom* g, void* workspace, size_t workspace_bytes, void* stream) {
    if (!wgrad_planes || !x_planes || !dw_oihw || !workspace) return SSD_ERR_NULL;
    if (!wino_geom_ok(g) || g->Ci % 4 != 0 || ldy % 4 != 0 || ldy < g->Co) return SSD_ERR_BAD_SHAPE;
    if (dbias && !bias_partial) return SSD_ERR_NULL;
    if (!ssd_aligned16(wgrad_planes) || !ssd_aligned16(x_planes) || !ssd_aligned16(dw_oihw) || !ssd_aligned16(workspace)) return SSD_ERR_ALIGN;
    const WinoWgradPlan w = wino_wgrad_plan(g, ldy, 4);
    if (workspace_bytes < w.zb) return SSD_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* Zs = static_cast<float*>(workspace);
    if (int e = launch_wgrad_tn(wgrad_planes, x_planes, Zs, g, ldy, w, st)) return e;
    const size_t total = (size_t)g->Co * g->Ci;
    int ks_left = w.ks;
    if (w.ks > 1 && total / 4 < 65536) {
        hipLaunchKernelGGL(wino_splitk_sum_kernel, dim3(grid_for(total / 4 * w.P)), dim3(256), 0, st, Zs, total, w.P, w.ks);
        SSD_CHECK_LAUNCH();
        ks_left = 1;
    }
    const int blocks = dbias ? dy_bias_blocks_for(g, ldy, w.tiles) : 0;
    if (dbias && blocks == 0) return SSD_ERR_BAD_SHAPE;
    const int nfin = grid_for(total / 4), ldp = (g->Co + 3) / 4 * 4;
    const BiasTail bt{dbias ? bias_partial : nullptr, dbias, blocks, g->Co, ldp};
    hipLaunchKernelGGL(wino4_wgrad_finish_kernel, dim3(nfin + (dbias ? (ldp + 15) / 16 : 0)), dim3(256), 0, st, Zs, dw_oihw, g->Co, g->Ci, ks_left,
                       (size_t)w.ks * total, nfin, bt);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

// ---- the adjoint-form data gradient (wino4_adj_out_kernel): GEMMs, stand-alone output, output fused with the next dy transform ------
extern "C" int ssd_wino_weights_adj(const float* w_oihw, float* U_adj, int Co, int Ci, int Co_pad, void* stream) {
    if (!w_oihw || !U_adj) return SSD_ERR_NULL;
    if (Co <= 0 || Ci <= 0 || Co_pad < Co) return SSD_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(wino4_weight_kernel, dim3(grid_for((size_t)Ci * Co_pad)), dim3(256), 0, (hipStream_t)stream, w_oihw, U_adj, Co, Ci, Ci, Co_pad, 2,
                       (int)use_x3(4, Co_pad));
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

// md_planes [36][tiles][pad4(Ci)] = y_planes [36][tiles][ldy] . U_adj[p] ([Ci][ldy], f32 or limb planes by ssd_wino_uses_x3(4, ldy))
extern "C" size_t ssd_wino4_adj_planes_floats(const ssd_conv_geom* g) {
    if (!wino_geom_ok(g) || g->dil != 1) return 0;
    const LatPlan lp = lat_plan(g->H, g->W, 1);
    return (size_t)36 * g->N * lp.TH * lp.TW * ((g->Ci + 3) / 4 * 4);
}

extern "C" int ssd_conv3x3_wino_dgrad_adj_gemm(const float* y_planes, int ldy, const float* U_adj, float* md_planes, const ssd_conv_geom* g,
                                               void* stream) {
    if (!y_planes || !U_adj || !md_planes) return SSD_ERR_NULL;
    if (!wino_geom_ok(g) || g->dil != 1 || ldy % 32 != 0 || ldy < g->Co || g->Ci % 4 != 0) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(y_planes) || !ssd_aligned16(U_adj) || !ssd_aligned16(md_planes)) return SSD_ERR_ALIGN;
    const LatPlan lp = lat_plan(g->H, g->W, 1);
    const size_t tiles = (size_t)g->N * lp.TH * lp.TW;
    if (tiles >= (1ull << 31)) return SSD_ERR_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (use_x3(4, ldy)) return ssd_internal_gemm_batched_x3(y_planes, U_adj, md_planes, (int)tiles, ldy, g->Ci, g->Ci, 36, tiles * ldy, st);
    return ssd_internal_gemm_batched(y_planes, U_adj, md_planes, (int)tiles, ldy, g->Ci, g->Ci, 36, tiles * ldy, (size_t)g->Ci * ldy, 1, st);
}

// dx (N,H,W,Ci) [+=] overlap-add of B md B^T, ReLU-masked by the bit words of the forward's input transform or by an f32 tensor
extern "C" int ssd_wino4_adj_output(const float* md_planes, float* dx, const float* relu_mask, const uint64_t* relu_bits, int accumulate,
                                    const ssd_conv_geom* g, void* stream) {
    if (!md_planes || !dx) return SSD_ERR_NULL;
    if (!wino_geom_ok(g) || g->dil != 1 || g->Ci % 4 != 0) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(md_planes) || !ssd_aligned16(dx) || (relu_mask && !ssd_aligned16(relu_mask)) || (relu_bits && ((uintptr_t)relu_bits & 7)))
        return SSD_ERR_ALIGN;
    const LatPlan lp = lat_plan(g->H, g->W, 1);
    const size_t tiles = (size_t)g->N * lp.TH * lp.TW;
    hipLaunchKernelGGL((wino4_adj_out_kernel<false, false>), dim3(grid_for(tiles * (g->Ci / 4))), dim3(256), 0, (hipStream_t)stream, md_planes, g->N, g->H,
                       g->W, g->Ci, lp.TH, lp.TW, dx, g->Ci, relu_mask, reinterpret_cast<const unsigned long long*>(relu_bits), accumulate,
                       static_cast<float*>(nullptr), static_cast<float*>(nullptr), 0);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

// The same block of dx, never stored: masked, it IS the dy of the layer below (g_below: same map, Co = g->Ci, its input of that layer
// being the ReLU output the mask belongs to) -- out come that layer's planes A dy A^T [36][tiles][g->Ci] and its bias partial sums
// (ssd_wino4_bias_partial_floats(g_below, g->Ci) floats), ready for ssd_wino4_wgrad_gemm and ssd_conv3x3_wino_dgrad_adj_gemm.
extern "C" int ssd_wino4_adj_output_to_planes(const float* md_planes, const float* relu_mask, const uint64_t* relu_bits, const ssd_conv_geom* g,
                                              const ssd_conv_geom* g_below, float* y_planes, float* bias_partial, void* stream) {
    if (!md_planes || !y_planes || !g_below) return SSD_ERR_NULL;
    if (!wino_geom_ok(g) || !wino_geom_ok(g_below) || g->dil != 1 || g_below->dil != 1 || g->Ci % 4 != 0 || g_below->Co != g->Ci ||
        g_below->N != g->N || g_below->H != g->H || g_below->W != g->W)
        return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(md_planes) || !ssd_aligned16(y_planes) || (relu_mask && !ssd_aligned16(relu_mask)) || (relu_bits && ((uintptr_t)relu_bits & 7)) ||
        (bias_partial && !ssd_aligned16(bias_partial)))
        return SSD_ERR_ALIGN;
    const LatPlan lp = lat_plan(g->H, g->W, 1);
    const size_t tiles = (size_t)g->N * lp.TH * lp.TW;
    hipStream_t st = (hipStream_t)stream;
    if (bias_partial) {
        const int blocks = dy_bias_blocks_for(g_below, g->Ci, tiles);
        if (blocks == 0) return SSD_ERR_BAD_SHAPE;
        hipLaunchKernelGGL((wino4_adj_out_kernel<true, true>), dim3(blocks), dim3(256), 0, st, md_planes, g->N, g->H, g->W, g->Ci, lp.TH, lp.TW,
                           static_cast<float*>(nullptr), g->Ci, relu_mask, reinterpret_cast<const unsigned long long*>(relu_bits), 0, y_planes, bias_partial,
                           g_below->Co);
    } else {
        hipLaunchKernelGGL((wino4_adj_out_kernel<true, false>), dim3(grid_for(tiles * (g->Ci / 4))), dim3(256), 0, st, md_planes, g->N, g->H, g->W, g->Ci,
                           lp.TH, lp.TW, static_cast<float*>(nullptr), g->Ci, relu_mask, reinterpret_cast<const unsigned long long*>(relu_bits), 0, y_planes,
                           static_cast<float*>(nullptr), 0);
    }
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

// ---- every filter transform / re-layout of a training step in ONE launch ---------------------------------------------------------------
// A step re-lays ~30 weight tensors (Winograd-domain filters for forward and dgrad, or the OHWI / IHWO copies of the direct kernels):
// 56 launches of ~9 microseconds in a row.  Here a job table (device memory, built once by the caller while the pointers stay the same)
// describes them all; a block finds its job by its index and works on a 256-element chunk of it.
namespace {
__global__ __launch_bounds__(256) void weight_jobs_kernel(const ssd_weight_job* __restrict__ jobs, const int* __restrict__ block_start, int njobs) {
    int lo = 0, hi = njobs - 1;                      // the job whose block range holds blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (block_start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const ssd_weight_job j = jobs[lo];
    const size_t i = (size_t)((int)blockIdx.x - block_start[lo]) * 256 + threadIdx.x;
    const int Co = j.co, Ci = j.ci, T = j.taps;
    auto src = [&](int co, int ci, int t) -> float {  // rows co < co0 come from w0, the rest from w1 (a head's bb and cl filters)
        return co < j.co0 ? j.w0[((size_t)co * Ci + ci) * T + t] : j.w1[((size_t)(co - j.co0) * Ci + ci) * T + t];
    };
    if (j.kind == 0) {                               // Winograd F(4x4,3x3): thread = (n, k) of U_fwd [36][Co][Ci] or of U_bwd [36][Ci][co_pad]
        // pad0 bit 0 / 1: out_fwd / out_bwd hold limb planes (ssd_wino_uses_x3 of their K): there a thread transforms 8 consecutive k of
        // one row and writes each plane's limbs as three 16-byte stores
        const int x3f = j.pad0 & 1, x3b = (j.pad0 >> 1) & 1;
        const bool adj = (j.pad0 >> 2) & 1;          // out_bwd in the ADJOINT form: the forward transform G g G^T (taps not rotated), rows ci, K = co
        const size_t nf = (size_t)Co * Ci, nb = (size_t)Ci * j.co_pad;
        const size_t tf = x3f ? nf / 8 : nf, tb = j.out_bwd == nullptr ? 0 : (x3b ? nb / 8 : nb);
        const bool fwd = i < tf;
        if (!fwd && i - tf >= tb) return;
        if (fwd && j.out_fwd == nullptr) return;
        const size_t e = fwd ? i : i - tf;
        const int Nrows = fwd ? Co : Ci, K = fwd ? Ci : j.co_pad;
        float* U = fwd ? j.out_fwd : j.out_bwd;
        if (fwd ? x3f : x3b) {
            // thread order = (half of a 16-k chunk, row, chunk): the 64 lanes of a store instruction write 1 KB of one limb image
            // contiguously (two threads per 32-byte row, consecutive rows), not 32-byte pieces 3 limb images apart
            const int half = (int)(e & 1), n = (int)((e >> 1) % Nrows), k8 = (int)((e >> 1) / Nrows) * 16 + half * 8;
            float g[8][9];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int co = fwd ? n : k8 + q, ci = fwd ? k8 + q : n;
#pragma unroll
                for (int t9 = 0; t9 < 9; ++t9) g[q][t9] = co < Co ? src(co, ci, (fwd || adj) ? t9 : 8 - t9) : 0.f;
            }
            const size_t limb = (size_t)((Nrows + 127) / 128 * 128) * 16;
            __bf16* base = reinterpret_cast<__bf16*>(U) + (size_t)(k8 >> 4) * 3 * limb + (size_t)n * 16 + (k8 & 15);
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                float t[8][3];
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int s2 = 0; s2 < 3; ++s2) t[q][s2] = w4_g_dot(a, g[q][s2], g[q][3 + s2], g[q][6 + s2]);
#pragma unroll
                for (int b = 0; b < 6; ++b) {
                    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
                    bf16x8 h, m, l;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float v = w4_g_dot(b, t[q][0], t[q][1], t[q][2]);
                        h[q] = (__bf16)v;                 // round-to-nearest limbs; both residuals are exact (gemm_x3.hip)
                        const float r1 = v - (float)h[q];
                        m[q] = (__bf16)r1;
                        l[q] = (__bf16)(r1 - (float)m[q]);
                    }
                    __bf16* d = base + (size_t)(a * 6 + b) * (K >> 4) * 3 * limb;
                    *reinterpret_cast<bf16x8*>(d) = h;
                    *reinterpret_cast<bf16x8*>(d + limb) = m;
                    *reinterpret_cast<bf16x8*>(d + 2 * limb) = l;
                }
            }
            return;
        }
        const int k = (int)(e % K), n = (int)(e / K);
        const int co = fwd ? n : k, ci = fwd ? k : n;
        float g[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) g[r][s] = co < Co ? src(co, ci, (fwd || adj) ? r * 3 + s : (2 - r) * 3 + (2 - s)) : 0.f;
        float t[6][3];
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int s = 0; s < 3; ++s) t[a][s] = w4_g_dot(a, g[0][s], g[1][s], g[2][s]);
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b)
                U[((size_t)(a * 6 + b) * Nrows + n) * K + k] = w4_g_dot(b, t[a][0], t[a][1], t[a][2]);
    } else if (j.kind == 1 || j.kind == 3) {
        // kind 1: OHWI [co_pad][T][Ci] and IHWO [Ci][T][co_pad] as f32 (the direct kernels' layouts); kind 3 (bf16-tensor mode): OHWI
        // [co_pad][T][Ci] and IHWO [Ci][T][pad1] as bf16 (pad1 = K of the data gradient).  Both layouts come from ONE read of a brick of
        // 32 output x 32 input channels x T taps: the source rows (32 x T contiguous floats per output channel) are read coalesced into
        // LDS, and both layouts leave as runs of 32 elements.  (One thread per output element, as kind 0 does, made every lane of an
        // IHWO load touch its own cache line: 0.37 ms per bf16-mode step for 210 MB.)
        __shared__ float brick[32][32 * 9 + 1];
        const bool b16 = j.kind == 3;
        const int rows_b = b16 ? j.pad1 : j.co_pad;
        const int nci = (Ci + 31) >> 5;
        const int bid = (int)blockIdx.x - block_start[lo];
        const int co0 = (bid / nci) * 32, ci0 = (bid % nci) * 32;
        const int cw = Ci - ci0 < 32 ? Ci - ci0 : 32;             // input channels of this brick
        const int run = cw * T;                                  // contiguous source floats per output channel
        for (int e = threadIdx.x; e < 32 * run; e += 256) {
            const int r = e / run, c = e - r * run, co = co0 + r;
            float v = 0.f;
            if (co < Co) v = co < j.co0 ? j.w0[((size_t)co * Ci + ci0) * T + c] : j.w1[((size_t)(co - j.co0) * Ci + ci0) * T + c];
            brick[r][c] = v;                                     // c = ci_l * T + t
        }
        __syncthreads();
        if (j.out_fwd != nullptr) {                              // [co][t][ci0 + l]: 32 lanes per (co, t)
            for (int e = threadIdx.x; e < 32 * T * 32; e += 256) {
                const int l = e & 31, rt = e >> 5, t = rt % T, r = rt / T, co = co0 + r;
                if (co < j.co_pad && l < cw) {
                    const size_t o = ((size_t)co * T + t) * Ci + ci0 + l;
                    if (b16) reinterpret_cast<__bf16*>(j.out_fwd)[o] = (__bf16)brick[r][l * T + t];
                    else j.out_fwd[o] = brick[r][l * T + t];
                }
            }
        }
        if (j.out_bwd != nullptr) {                              // [ci][t][co0 + l]: 32 lanes per (ci, t)
            for (int e = threadIdx.x; e < 32 * T * 32; e += 256) {
                const int l = e & 31, ct = e >> 5, t = ct % T, c = ct / T, co = co0 + l;
                if (c < cw && co < rows_b) {
                    const size_t o = ((size_t)(ci0 + c) * T + t) * rows_b + co;
                    if (b16) reinterpret_cast<__bf16*>(j.out_bwd)[o] = (__bf16)brick[l][c * T + t];
                    else j.out_bwd[o] = brick[l][c * T + t];
                }
            }
        }
    } else if (j.kind == 4) {                        // 1x1 filters as limb planes for csrc/gemm_x3.hip: out_fwd rows Co, K = Ci; out_bwd rows Ci, K = co_pad
        const size_t tf = j.out_fwd == nullptr ? 0 : (size_t)Co * Ci / 8, tb = j.out_bwd == nullptr ? 0 : (size_t)Ci * j.co_pad / 8;
        const bool fwd = i < tf;
        if (!fwd && i - tf >= tb) return;
        const size_t e = fwd ? i : i - tf;
        const int Nrows = fwd ? Co : Ci;             // (K = Ci forward, co_pad backward: Nrows * K / 8 threads)
        const int half = (int)(e & 1), n = (int)((e >> 1) % Nrows), k8 = (int)((e >> 1) / Nrows) * 16 + half * 8;      // (store order: as kind 0)
        typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
        bf16x8 h, m, l;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int co = fwd ? n : k8 + q, ci = fwd ? k8 + q : n;
            const float v = co < Co ? src(co, ci, 0) : 0.f;
            h[q] = (__bf16)v;                        // round-to-nearest limbs; both residuals are exact (gemm_x3.hip)
            const float r1 = v - (float)h[q];
            m[q] = (__bf16)r1;
            l[q] = (__bf16)(r1 - (float)m[q]);
        }
        const size_t limb = (size_t)((Nrows + 127) / 128 * 128) * 16;
        __bf16* d = reinterpret_cast<__bf16*>(fwd ? j.out_fwd : j.out_bwd) + (size_t)(k8 >> 4) * 3 * limb + (size_t)n * 16 + (k8 & 15);
        *reinterpret_cast<bf16x8*>(d) = h;
        *reinterpret_cast<bf16x8*>(d + limb) = m;
        *reinterpret_cast<bf16x8*>(d + 2 * limb) = l;
    } else {                                         // conv1_1 rows for the im2col GEMM: [Co][32], k = (r*3+s)*3 + c, zero padded
        const size_t total = (size_t)Co * 32;
        if (i < total) {
            const int k = (int)(i % 32), co = (int)(i / 32);
            j.out_fwd[i] = k < 27 ? j.w0[((size_t)co * 3 + (k % 3)) * 9 + (k / 3)] : 0.f;
        }
    }
}
}  // namespace

extern "C" int ssd_weight_job_blocks(const ssd_weight_job* job) {
    if (!job || job->co <= 0 || job->ci <= 0) return -1;
    size_t elems;
    if (job->kind == 0)          // threads: one per element, or per 8 elements of a limb-plane output (pad0 bit 0: out_fwd, bit 1: out_bwd)
        elems = (size_t)job->co * job->ci / ((job->pad0 & 1) ? 8 : 1) + (job->out_bwd ? (size_t)job->ci * job->co_pad / ((job->pad0 & 2) ? 8 : 1) : 0);
    else if (job->kind == 2) elems = (size_t)job->co * 32;
    else if (job->kind == 4) {
        if (job->taps != 1 || job->ci % 16 != 0 || job->co_pad % 16 != 0 || job->co_pad < job->co) return -1;
        elems = (job->out_fwd ? (size_t)job->co * job->ci / 8 : 0) + (job->out_bwd ? (size_t)job->ci * job->co_pad / 8 : 0);
        if (elems == 0) return -1;
    }
    else if (job->kind == 1 || job->kind == 3) {       // one block per brick of 32 output x 32 input channels (all taps)
        if (job->taps < 1 || job->taps > 9) return -1;
        const int rf = job->out_fwd ? job->co_pad : 0, rb = job->out_bwd ? (job->kind == 3 ? job->pad1 : job->co_pad) : 0, rows_all = rf > rb ? rf : rb;
        if (rows_all <= 0) return -1;
        const size_t b3 = (size_t)((rows_all + 31) / 32) * ((job->ci + 31) / 32);
        return b3 >= (1u << 30) ? -1 : (int)b3;
    }
    else return -1;
    const size_t b = (elems + 255) / 256;
    return b >= (1u << 30) ? -1 : (int)b;
}

extern "C" int ssd_weights_prepare(const ssd_weight_job* jobs_device, const int* block_start_device, int njobs, int total_blocks, void* stream) {
    if (!jobs_device || !block_start_device) return SSD_ERR_NULL;
    if (njobs <= 0 || total_blocks <= 0) return SSD_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(weight_jobs_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_device, block_start_device, njobs);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}
