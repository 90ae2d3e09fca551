// Weight gradient of nn.Conv2d (autograd of Model.py:135-184, train_function.py:94).
//
//   dW[n][tap][c] = sum_m dy[m][n] * x[gather(m,tap)][c],     db[n] = sum_m dy[m][n]
//
// GEMM with the pixel index m as the reduction dimension (up to 2.9 M at bs=32), so
// the grid is (split-K chunk, tap, n-tile, c-tile) and every block writes an f32
// partial tile into a slab; a second kernel adds the slabs in a fixed order and
// writes OIHW, which keeps the result bitwise reproducible (no float atomics).
// LDS tiles are [pixel][channel] exactly as they arrive from NHWC memory; the MFMA
// operands (A = dy^T, B = x) are read one float per lane with consecutive lanes on
// consecutive channels (ds_read_b32, conflict-free).
#include <type_traits>
#include "common.h"

namespace {

struct WgradParams {
    const float* __restrict__ x;
    const float* __restrict__ dy;
    float* __restrict__ slab;        // [nsplit][Co][T][Ci]
    float* __restrict__ bias_slab;   // [nsplit][Co] or null
    unsigned x_bytes, dy_bytes;
    int H, W, Ci, Ho, Wo, Co, ldy;
    int R, S, stride, pad, dil;
    int M, m_per_split, nsplit;
    int tiles_co, tiles_ci;
};

constexpr int WBK = 32;   // pixels per K step
constexpr unsigned OOB = 0xFFFFFF00u;

__device__ __forceinline__ f32x4 buf_load16(__amdgpu_buffer_rsrc_t srd, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, (int)voff, (int)soff, 0));
}

// BT = tile edge (both n and c), 4 waves as 2x2, each wave (BT/2)x(BT/2).
// Both LDS stages hold [pixel][channel] rows as they come from NHWC memory.  Within a wave's
// 64-channel span MFMA tile i takes channels 2*lane+i, so one ds_read_b64 feeds both tiles.
template <int BT, int NBUF>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
    constexpr int TT = BT / 64;                 // 32x32 accumulators per wave per dim
    constexpr int CHUNKS = BT / 4;              // float4 chunks per tile row
    constexpr int ROWS_PER_PASS = 256 / CHUNKS; // pixel rows loaded per pass
    constexpr int PASSES = WBK / ROWS_PER_PASS;
    constexpr int STAGE = 2 * WBK * BT;
    __shared__ __attribute__((aligned(16))) float lds[NBUF * STAGE];
    __shared__ float bias_red[256 * 4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int T = p.R * p.S;
    // block id -> (split, co tile, ci tile, tap); tap fastest so the T blocks that re-read the
    // same dy / x chunk are neighbours in one XCD's L2.
    const int per_split = T * p.tiles_co * p.tiles_ci;
    const int nblk = per_split * p.nsplit;
    int lid = xcd_swizzle(blockIdx.x, nblk);
    const int split = lid / per_split;
    lid -= split * per_split;
    const int t = lid % T;
    lid /= T;
    const int tile_ci = lid % p.tiles_ci, tile_co = lid / p.tiles_ci;
    const int co0 = tile_co * BT, ci0 = tile_ci * BT;
    const int r = t / p.S, s = t - r * p.S;
    const int dh = r * p.dil - p.pad, dw = s * p.dil - p.pad;

    const int chunk = tid % CHUNKS, prow = tid / CHUNKS;
    const int m_begin = split * p.m_per_split;
    const int m_end = min(p.M, m_begin + p.m_per_split);
    const bool y_col_ok = co0 + chunk * 4 < p.ldy;     // columns Co..ldy-1 of dy are zero padding (ldy % 4 == 0)
    const bool x_col_ok = ci0 + chunk * 4 < p.Ci;
    const bool do_bias = p.bias_slab != nullptr && t == 0 && tile_ci == 0;
    const int HoWo = p.Ho * p.Wo;

    const __amdgpu_buffer_rsrc_t srd_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (int)p.dy_bytes, 0x00020000);

    // per-thread pixel rows: coordinates advance by WBK pixels per K step (no division in the loop)
    int pn[PASSES], poh[PASSES], pow_[PASSES], pm[PASSES];
#pragma unroll
    for (int j = 0; j < PASSES; ++j) {
        const int m = m_begin + prow + ROWS_PER_PASS * j;
        pm[j] = m;
        const int mm = m < p.M ? m : 0;
        pn[j] = mm / HoWo;
        const int rem = mm - pn[j] * HoWo;
        poh[j] = rem / p.Wo;
        pow_[j] = rem - poh[j] * p.Wo;
    }
    const unsigned y_col = (unsigned)(co0 + chunk * 4) * 4u, x_col = (unsigned)(ci0 + chunk * 4) * 4u;

    f32x16 acc[TT][TT];
#pragma unroll
    for (int i = 0; i < TT; ++i)
#pragma unroll
        for (int j = 0; j < TT; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

    f32x4 ry[PASSES], rx[PASSES];
    const bool wide = p.Wo >= WBK;               // a 32-pixel step wraps at most one image row
    const unsigned ldy4 = (unsigned)p.ldy * 4u, ci4 = (unsigned)p.Ci * 4u;
    auto issue_loads = [&]() {
#pragma unroll
        for (int j = 0; j < PASSES; ++j) {
            const bool in = pm[j] < m_end;
            const unsigned vy = (in && y_col_ok) ? (unsigned)pm[j] * ldy4 + y_col : OOB;
            const int ih = poh[j] * p.stride + dh, iw = pow_[j] * p.stride + dw;
            const bool xin = in && x_col_ok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            const unsigned vx = xin ? (unsigned)((pn[j] * p.H + ih) * p.W + iw) * ci4 + x_col : OOB;
            ry[j] = buf_load16(srd_y, vy, 0);
            rx[j] = buf_load16(srd_x, vx, 0);
            // advance this row by WBK pixels (branch-free when the map is at least 32 wide)
            pm[j] += WBK;
            if (wide) {
                pow_[j] += WBK;
                const bool w1 = pow_[j] >= p.Wo;
                pow_[j] -= w1 ? p.Wo : 0;
                poh[j] += w1 ? 1 : 0;
                const bool w2 = poh[j] >= p.Ho;
                poh[j] = w2 ? 0 : poh[j];
                pn[j] += w2 ? 1 : 0;
            } else {
                const int mm = pm[j] < p.M ? pm[j] : 0;
                pn[j] = mm / HoWo;
                const int rem = mm - pn[j] * HoWo;
                poh[j] = rem / p.Wo;
                pow_[j] = rem - poh[j] * p.Wo;
            }
        }
    };
    auto store_tile = [&](float* stage) {
        float* Ys = stage;
        float* Xs = stage + WBK * BT;
#pragma unroll
        for (int j = 0; j < PASSES; ++j) {
            const int row = prow + ROWS_PER_PASS * j;
            *reinterpret_cast<f32x4*>(&Ys[row * BT + chunk * 4]) = ry[j];
            *reinterpret_cast<f32x4*>(&Xs[row * BT + chunk * 4]) = rx[j];
            if (do_bias) bsum += ry[j];
        }
    };

    const int lr = lane & 31, lh = lane >> 5;
    // operand read offsets: row (2kk+lh), channels wave_span + TT*lr + i
    const int y_rd = lh * BT + wm * (BT / 2) + TT * lr;
    const int x_rd = WBK * BT + lh * BT + wn * (BT / 2) + TT * lr;
    if (m_begin < m_end) {
        issue_loads();
        store_tile(lds);
        __syncthreads();
        int cur = 0;
        for (int mb = m_begin; mb < m_end; mb += WBK) {
            const bool more = mb + WBK < m_end;
            if (more) issue_loads();
            const float* stage = lds + (NBUF == 2 ? cur * STAGE : 0);
            // operands of k-pair kk+1 are read from LDS before the MFMAs of k-pair kk are issued
            float an[TT], bn[TT];
            auto rd = [&](int kk) {
                if (TT == 2) {
                    const float2 a2 = *reinterpret_cast<const float2*>(stage + y_rd + 2 * kk * BT);
                    const float2 b2 = *reinterpret_cast<const float2*>(stage + x_rd + 2 * kk * BT);
                    an[0] = a2.x; an[TT - 1] = a2.y; bn[0] = b2.x; bn[TT - 1] = b2.y;
                } else {
                    an[0] = stage[y_rd + 2 * kk * BT];
                    bn[0] = stage[x_rd + 2 * kk * BT];
                }
            };
            rd(0);
#pragma unroll
            for (int kk = 0; kk < WBK / 2; ++kk) {
                float af[TT], bf[TT];
#pragma unroll
                for (int i = 0; i < TT; ++i) { af[i] = an[i]; bf[i] = bn[i]; }
                if (kk + 1 < WBK / 2) rd(kk + 1);
#pragma unroll
                for (int i = 0; i < TT; ++i)
#pragma unroll
                    for (int j = 0; j < TT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);        // 2 LDS reads (next k-pair) ...
                __builtin_amdgcn_sched_group_barrier(0x008, TT * TT, 0);  // ... then this k-pair's MFMAs
            }
            if (NBUF == 2) {
                if (more) store_tile(lds + (cur ^ 1) * STAGE);
                __syncthreads();
                cur ^= 1;
            } else {
                __syncthreads();
                if (more) {
                    store_tile(lds);
                    __syncthreads();
                }
            }
        }
    }

    // ---- partial tile -> slab[split][co][t][ci]; accumulator (i,j), row q -> channel TT*row + i ------
    float* slab = p.slab + (size_t)split * p.Co * T * p.Ci;
#pragma unroll
    for (int i = 0; i < TT; ++i)
#pragma unroll
        for (int j = 0; j < TT; ++j) {
            const int ci = ci0 + wn * (BT / 2) + TT * lr + j;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = (q & 3) + 8 * (q >> 2) + 4 * lh;
                const int co = co0 + wm * (BT / 2) + TT * row + i;
                if (co < p.Co && ci < p.Ci) slab[((size_t)co * T + t) * p.Ci + ci] = acc[i][j][q];
            }
        }
    if (do_bias) {   // uniform per block
#pragma unroll
        for (int e = 0; e < 4; ++e) bias_red[tid * 4 + e] = bsum[e];
        __syncthreads();
        if (tid < CHUNKS) {
            f32x4 tot = {0.f, 0.f, 0.f, 0.f};
            for (int q = 0; q < ROWS_PER_PASS; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) tot[e] += bias_red[(q * CHUNKS + tid) * 4 + e];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = co0 + tid * 4 + e;
                if (co < p.Co) p.bias_slab[(size_t)split * p.Co + co] = tot[e];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// All nine taps in one block (3x3, stride 1, dilation 1, padding 1 -- every VGG conv and every head).
// The K dimension is walked in 4x8-pixel output patches: per patch the dy tile [32 px][64 co] and the
// x HALO tile [6x10 px][64 ci] are staged in LDS once, and each of the nine taps multiplies the same dy
// operand with the halo shifted by (r,s) -- an immediate LDS offset.  Against one-tap-per-block this
// loads dy once instead of nine times and x 1.9x instead of 9x, and amortises the two barriers of a
// K step over 144 MFMAs per wave instead of 16.  4 waves as 2x2, each 32x32 per tap: nine accumulators.
// ---------------------------------------------------------------------------------------------
constexpr int PH = 4, PW = 8, HH = PH + 2, HW = PW + 2, HPIX = HH * HW;   // patch 4x8, halo 6x10 = 60 pixels

// Patch shapes: 4x8 (32 pixels) is the default; 1x38 and 2x19 (38 pixels = 19 k-pairs) tile the 38- / 75- / 19-pixel
// maps of conv3, conv4 and conv5 without the 8-33 % of padded MFMAs the 4x8 grid has there (plan_wgrad picks by waste).
template <int PH_, int PW_>
__global__ __launch_bounds__(256, 2) void wgrad3x3_kernel(const WgradParams p) {
    constexpr int BT = 64, CHUNKS = 16, RPP = 16;
    constexpr int NPIX = PH_ * PW_, HH_ = PH_ + 2, HW_ = PW_ + 2, HPIX_ = HH_ * HW_;
    constexpr int YP = (NPIX + RPP - 1) / RPP, XP = (HPIX_ + RPP - 1) / RPP;      // load passes of the block's 16 pixel rows
    static_assert(NPIX % 2 == 0, "pixels are consumed in k-pairs");
    __shared__ __attribute__((aligned(16))) float Ys[NPIX * BT];
    __shared__ __attribute__((aligned(16))) float Xs[HPIX_ * BT];
    __shared__ float bias_red[256 * 4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;
    // block -> (split, co tile, ci tile); ci fastest so neighbours share the dy patch stream in L2
    const int per_split = p.tiles_co * p.tiles_ci;
    const int nblk = per_split * p.nsplit;
    int lid = xcd_swizzle(blockIdx.x, nblk);
    const int split = lid / per_split;
    lid -= split * per_split;
    const int tile_ci = lid % p.tiles_ci, tile_co = lid / p.tiles_ci;
    const int co0 = tile_co * BT, ci0 = tile_ci * BT;
    const int npw = (p.Wo + PW_ - 1) / PW_, nph = (p.Ho + PH_ - 1) / PH_;
    const int per_img = npw * nph;
    const int npatch = per_img * (p.M / (p.Ho * p.Wo));
    const int pb = split * p.m_per_split;                       // here: patches per split
    const int pe = min(npatch, pb + p.m_per_split);

    const int chunk = tid % CHUNKS, prow = tid / CHUNKS;
    const bool y_col_ok = co0 + chunk * 4 < p.ldy;
    const bool x_col_ok = ci0 + chunk * 4 < p.Ci;
    const bool do_bias = p.bias_slab != nullptr && tile_ci == 0;
    const unsigned y_col = (unsigned)(co0 + chunk * 4) * 4u, x_col = (unsigned)(ci0 + chunk * 4) * 4u;
    const unsigned ldy4 = (unsigned)p.ldy * 4u, ci4 = (unsigned)p.Ci * 4u;

    const __amdgpu_buffer_rsrc_t srd_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (int)p.dy_bytes, 0x00020000);

    // this thread's pixels inside a patch / halo (fixed for the whole kernel)
    int ypy[YP], ypx[YP], xhy[XP], xhx[XP];
#pragma unroll
    for (int j = 0; j < YP; ++j) { const int q = prow + RPP * j; ypy[j] = q / PW_; ypx[j] = q % PW_; }      // q >= NPIX: unused
#pragma unroll
    for (int j = 0; j < XP; ++j) { const int q = prow + RPP * j; xhy[j] = q / HW_; xhx[j] = q % HW_; }      // q >= HPIX: unused

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    f32x4 ry[YP], rx[XP];

    auto issue_loads = [&](int patch) {
        const int n = patch / per_img, rem = patch - n * per_img;          // uniform: scalar unit
        const int oh0 = (rem / npw) * PH_, ow0 = (rem % npw) * PW_;
#pragma unroll
        for (int j = 0; j < YP; ++j) {
            const int oh = oh0 + ypy[j], ow = ow0 + ypx[j];
            const bool ok = y_col_ok && (prow + RPP * j) < NPIX && oh < p.Ho && ow < p.Wo;
            const unsigned v = ok ? (unsigned)((n * p.Ho + oh) * p.Wo + ow) * ldy4 + y_col : OOB;
            ry[j] = buf_load16(srd_y, v, 0);
        }
#pragma unroll
        for (int j = 0; j < XP; ++j) {
            const int ih = oh0 + xhy[j] - 1, iw = ow0 + xhx[j] - 1;
            const bool ok = x_col_ok && (prow + RPP * j) < HPIX_ && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            const unsigned v = ok ? (unsigned)((n * p.H + ih) * p.W + iw) * ci4 + x_col : OOB;
            rx[j] = buf_load16(srd_x, v, 0);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int j = 0; j < YP; ++j)
            if (prow + RPP * j < NPIX) {
                *reinterpret_cast<f32x4*>(&Ys[(prow + RPP * j) * BT + chunk * 4]) = ry[j];
                if (do_bias) bsum += ry[j];
            }
#pragma unroll
        for (int j = 0; j < XP; ++j)
            if (prow + RPP * j < HPIX_) *reinterpret_cast<f32x4*>(&Xs[(prow + RPP * j) * BT + chunk * 4]) = rx[j];
    };

    // operand addresses: k-pair kk covers patch pixels q = 2kk + lh (lane half); pixel q sits at halo position
    // (q / PW, q % PW) + (r, s).  With an even PW both pixels of a pair are in one row and lh is a constant offset.
    const float* y_rd = Ys + lh * BT + wm * 32 + lr;
    const float* x_rd = Xs + wn * 32 + lr + (PW_ % 2 == 0 ? lh * BT : 0);
    if (pb < pe) {
        issue_loads(pb);
        store_tile();
        __syncthreads();
        for (int patch = pb; patch < pe; ++patch) {
            const bool more = patch + 1 < pe;
            if (more) issue_loads(patch + 1);
#pragma unroll
            for (int kk = 0; kk < NPIX / 2; ++kk) {
                const float a = y_rd[2 * kk * BT];
                const int q0 = 2 * kk, q1 = 2 * kk + 1;
                const int h0 = ((q0 / PW_) * HW_ + (q0 % PW_)) * BT, h1 = ((q1 / PW_) * HW_ + (q1 % PW_)) * BT;
                const int hbase = (PW_ % 2 == 0) ? h0 : (lh ? h1 : h0);
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int s2 = 0; s2 < 3; ++s2) {
                        const float b = x_rd[hbase + (r * HW_ + s2) * BT];
                        acc[r * 3 + s2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[r * 3 + s2], 0, 0, 0);
                    }
            }
            __syncthreads();
            if (more) {
                store_tile();
                __syncthreads();
            }
        }
    }

    float* slab = p.slab + (size_t)split * p.Co * 9 * p.Ci;
    const int ci = ci0 + wn * 32 + lr;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int co = co0 + wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
            if (co < p.Co && ci < p.Ci) slab[((size_t)co * 9 + t) * p.Ci + ci] = acc[t][q];
        }
    if (do_bias) {   // uniform per block
#pragma unroll
        for (int e = 0; e < 4; ++e) bias_red[tid * 4 + e] = bsum[e];
        __syncthreads();
        if (tid < CHUNKS) {
            f32x4 tot = {0.f, 0.f, 0.f, 0.f};
            for (int q = 0; q < RPP; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) tot[e] += bias_red[(q * CHUNKS + tid) * 4 + e];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = co0 + tid * 4 + e;
                if (co < p.Co) p.bias_slab[(size_t)split * p.Co + co] = tot[e];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// bf16-operand version of the fused nine-tap kernel (BASELINE.json configs[2]).  Same patches, same halo, f32 tensors in
// HBM and f32 accumulators / slabs; the dy and x tiles are rounded to bf16 on their way into LDS ([pixel][channel], 192-byte
// rows: conflict-free for the transposed read) and the MFMA operands -- 8 consecutive PIXELS of one channel per lane --
// are fetched with ds_read_b64_tr_b16, the gfx950 transposing LDS read (4 pixel rows x 16 channels per 16-lane group).
// K step = one 4x8 patch = two 32x32x16 MFMAs per tap (sub-step ks covers patch rows 2ks, 2ks+1; lane half h = the row).
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
constexpr int LDT = 96;        // LDS row stride in bf16 elements (64 channels + 32 pad = 192 bytes)

// fragment of a 32x32x16 operand: rows row0..row0+7 of `tile` (k), columns col0 + (lane & 31) (the M / N index)
__device__ __forceinline__ bf16x8 tr_frag(const __bf16* tile, int row0, int col0, int lane) {
    const int q = (lane & 15) >> 2, p4 = (lane & 3) * 4, g16 = ((lane >> 4) & 1) * 16;
    const __bf16* a0 = tile + (row0 + q) * LDT + col0 + g16 + p4;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * LDT));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// IN_BF16: x and dy are bf16 tensors (the bf16-tensor mode): 8-byte loads of the same four channels per thread, straight into LDS.
// TAPS = 9: 3x3 filter with padding = dilation = DIL (DIL = 1: the VGG layers and heads; DIL = 4: fc6, Model.py:149 -- the halo of a
// 4 x 8 patch is then 12 x 16 pixels); TAPS = 1: 1x1 filter (fc7, the aux blocks' first convolutions: no halo, one accumulator).
template <bool IN_BF16, int TAPS, int DIL>
__global__ __launch_bounds__(256, 2) void wgrad3x3_bf16_kernel(const WgradParams p) {
    constexpr unsigned ES = IN_BF16 ? 2u : 4u;                  // bytes per element of x / dy
    constexpr int BT = 64, CHUNKS = 16, RPP = 16;
    constexpr int PADX = TAPS == 9 ? DIL : 0;
    // TAPS = 1: a "patch" is 128 CONSECUTIVE pixels of the flattened batch (no 2-D structure to respect, no halo): 8 k-steps = 8 MFMAs per
    // wave between two barriers instead of the 2 a 32-pixel patch gives a single tap (fc7: 0.22 ms at 112 TFLOP/s with 32-pixel patches)
    constexpr int NPP = TAPS == 1 ? 128 : PH * PW;             // pixels per patch
    constexpr int YP = NPP / RPP, KSTEPS = NPP / 16;
    constexpr int HHd = PH + 2 * PADX, HWd = PW + 2 * PADX, HPIXd = TAPS == 1 ? NPP : HHd * HWd, XP = (HPIXd + RPP - 1) / RPP;
    // (Round 4: two tile buffers -- the next patch parked in the other buffer right after this patch's MFMAs, ONE barrier per patch instead of
    // barrier - park - barrier -- measured the same: conv4 0.253 against 0.251 ms, conv1_2 0.264 against 0.254.  One buffer stays.)
    __shared__ __attribute__((aligned(16))) __bf16 Ys[NPP * LDT];
    __shared__ __attribute__((aligned(16))) __bf16 Xs[HPIXd * LDT];
    __shared__ float bias_red[256 * 4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;
    const int per_split = p.tiles_co * p.tiles_ci;
    const int nblk = per_split * p.nsplit;
    int lid = xcd_swizzle(blockIdx.x, nblk);
    const int split = lid / per_split;
    lid -= split * per_split;
    const int tile_ci = lid % p.tiles_ci, tile_co = lid / p.tiles_ci;
    const int co0 = tile_co * BT, ci0 = tile_ci * BT;
    const int npw = (p.Wo + PW - 1) / PW, nph = (p.Ho + PH - 1) / PH;
    const int per_img = npw * nph;
    const int npatch = TAPS == 1 ? (p.M + NPP - 1) / NPP : per_img * (p.M / (p.Ho * p.Wo));
    const int pb = split * p.m_per_split;
    const int pe = min(npatch, pb + p.m_per_split);

    const int chunk = tid % CHUNKS, prow = tid / CHUNKS;
    const bool y_col_ok = co0 + chunk * 4 < p.ldy;
    const bool x_col_ok = ci0 + chunk * 4 < p.Ci;
    const bool do_bias = p.bias_slab != nullptr && tile_ci == 0;
    const unsigned y_col = (unsigned)(co0 + chunk * 4) * ES, x_col = (unsigned)(ci0 + chunk * 4) * ES;
    const unsigned ldy4 = (unsigned)p.ldy * ES, ci4 = (unsigned)p.Ci * ES;

    const __amdgpu_buffer_rsrc_t srd_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (int)p.dy_bytes, 0x00020000);

    int ypy[YP], ypx[YP], xhy[XP], xhx[XP];
#pragma unroll
    for (int j = 0; j < YP; ++j) { const int q = prow + RPP * j; ypy[j] = q / PW; ypx[j] = q % PW; }
#pragma unroll
    for (int j = 0; j < XP; ++j) { const int q = prow + RPP * j; xhy[j] = q / HWd; xhx[j] = q % HWd; }

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    typedef typename std::conditional<IN_BF16, u32x2, f32x4>::type raw_t;      // four channels as they come from memory
    raw_t ry[YP], rx[XP];
    auto load4 = [](__amdgpu_buffer_rsrc_t srd, unsigned voff) -> raw_t {
        if constexpr (IN_BF16) return __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(srd, (int)voff, 0, 0));
        else return buf_load16(srd, voff, 0);
    };

    auto issue_loads = [&](int patch) {
        if constexpr (TAPS == 1) {                             // pixel m = patch * 128 + row: the same row of dy and of x (1x1, stride 1, no padding)
#pragma unroll
            for (int j = 0; j < YP; ++j) {
                const int m = patch * NPP + prow + RPP * j;
                const bool ok = m < p.M;
                ry[j] = load4(srd_y, ok && y_col_ok ? (unsigned)m * ldy4 + y_col : OOB);
                rx[j] = load4(srd_x, ok && x_col_ok ? (unsigned)m * ci4 + x_col : OOB);
            }
            return;
        }
        const int n = patch / per_img, rem = patch - n * per_img;
        const int oh0 = (rem / npw) * PH, ow0 = (rem % npw) * PW;
#pragma unroll
        for (int j = 0; j < YP; ++j) {
            const int oh = oh0 + ypy[j], ow = ow0 + ypx[j];
            const bool ok = y_col_ok && oh < p.Ho && ow < p.Wo;
            const unsigned v = ok ? (unsigned)((n * p.Ho + oh) * p.Wo + ow) * ldy4 + y_col : OOB;
            ry[j] = load4(srd_y, v);
        }
#pragma unroll
        for (int j = 0; j < XP; ++j) {
            const int ih = oh0 + xhy[j] - PADX, iw = ow0 + xhx[j] - PADX;
            const bool ok = x_col_ok && (prow + RPP * j) < HPIXd && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            const unsigned v = ok ? (unsigned)((n * p.H + ih) * p.W + iw) * ci4 + x_col : OOB;
            rx[j] = load4(srd_x, v);
        }
    };
    auto to_bf16 = [](const raw_t v) -> bf16x4 {
        if constexpr (IN_BF16) return __builtin_bit_cast(bf16x4, v);
        else return bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int j = 0; j < YP; ++j) {
            const bf16x4 b = to_bf16(ry[j]);
            *reinterpret_cast<bf16x4*>(&Ys[(prow + RPP * j) * LDT + chunk * 4]) = b;
            if (do_bias) {                                    // bias gradient: f32 sums of dy as it is stored
                if constexpr (IN_BF16) bsum += f32x4{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
                else bsum += ry[j];
            }
        }
#pragma unroll
        for (int j = 0; j < XP; ++j)
            if (prow + RPP * j < HPIXd) *reinterpret_cast<bf16x4*>(&Xs[(prow + RPP * j) * LDT + chunk * 4]) = to_bf16(rx[j]);
    };

    if (pb < pe) {
        issue_loads(pb);
        store_tile();
        __syncthreads();
        for (int patch = pb; patch < pe; ++patch) {
            const bool more = patch + 1 < pe;
            if (more) issue_loads(patch + 1);
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                // k = 8*lh + j  <->  patch pixel (py = 2ks + lh, px = j)
                const bf16x8 a = tr_frag(Ys, (2 * ks + lh) * PW, wm * 32, lane);
                if constexpr (TAPS == 9) {
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int s2 = 0; s2 < 3; ++s2) {
                            const bf16x8 b = tr_frag(Xs, (2 * ks + lh + r * DIL) * HWd + s2 * DIL, wn * 32, lane);
                            acc[r * 3 + s2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[r * 3 + s2], 0, 0, 0);
                        }
                } else {
                    const bf16x8 b = tr_frag(Xs, (2 * ks + lh) * HWd, wn * 32, lane);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[0], 0, 0, 0);
                }
            }
            __syncthreads();
            if (more) {
                store_tile();
                __syncthreads();
            }
        }
    }

    float* slab = p.slab + (size_t)split * p.Co * TAPS * p.Ci;
    const int ci = ci0 + wn * 32 + lr;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int co = co0 + wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
            if (co < p.Co && ci < p.Ci) slab[((size_t)co * TAPS + t) * p.Ci + ci] = acc[t][q];
        }
    if (do_bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bias_red[tid * 4 + e] = bsum[e];
        __syncthreads();
        if (tid < CHUNKS) {
            f32x4 tot = {0.f, 0.f, 0.f, 0.f};
            for (int q = 0; q < RPP; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) tot[e] += bias_red[(q * CHUNKS + tid) * 4 + e];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = co0 + tid * 4 + e;
                if (co < p.Co) p.bias_slab[(size_t)split * p.Co + co] = tot[e];
            }
        }
    }
}

// out_oihw[co][ci][t] = sum_split slab[split][co][t][ci], and db[co] = sum_split bias_slab[split][co], in one launch.
// Block = (output channel co, chunk of 64 input channels).  The T x 64 sums are read along ci (coalesced in the slab), parked
// in LDS and written along the OIHW order ci*T + t, which for a fixed co is one contiguous run of 64*T floats -- no strided
// 4-byte stores.  The split loop is unrolled into eight independent partial sums (loads in flight instead of one dependent
// load per ~1 us); every association order is fixed, so the result stays bitwise reproducible.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int Co, int Ci, int T,
                                                           int nsplit, const float* __restrict__ bias_slab, float* __restrict__ db) {
    __shared__ float tile[49 * 64];                              // up to 7x7 taps
    const int chunks = (Ci + 63) / 64;
    const int co = blockIdx.x / chunks, ci0 = (blockIdx.x % chunks) * 64;
    const int nci = min(64, Ci - ci0);
    const size_t total = (size_t)Co * T * Ci;
    for (int e = threadIdx.x; e < T * 64; e += 256) {
        const int t = e >> 6, c = e & 63;
        float s = 0.f;
        if (c < nci) {
            const size_t i = ((size_t)co * T + t) * Ci + ci0 + c;
            float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            int k = 0;
            for (; k + 8 <= nsplit; k += 8)
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u] += slab[(size_t)(k + u) * total + i];
            for (int u = 0; k < nsplit; ++k, ++u) a[u] += slab[(size_t)k * total + i];
            s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        }
        tile[t * 64 + c] = s;
    }
    __syncthreads();
    float* dst = dw + ((size_t)co * Ci + ci0) * T;
    for (int j = threadIdx.x; j < nci * T; j += 256) dst[j] = tile[(j % T) * 64 + j / T];
    if (db != nullptr && ci0 == 0 && threadIdx.x < 64) {         // wave 0: lanes stride over the splits, fixed shuffle tree
        float s = 0.f;
        for (int k = threadIdx.x; k < nsplit; k += 64) s += bias_slab[(size_t)k * Co + co];
        s = wave_sum(s);
        if (threadIdx.x == 0) db[co] = s;
    }
}

// The same sums for layers with FEW output channels and MANY splits (conv1_2: 64 x 1 blocks above, each thread adding 512 slabs one
// dependent round of 8 loads after the other: 67 us, latency-bound on a quarter of the CUs).  Block = (co, 64-channel chunk, TAP); the four
// waves take a quarter of the splits each (same eight-accumulator order within a quarter), the quarters are added in fixed order.
__global__ __launch_bounds__(256) void wgrad_reduce_tap_kernel(const float* __restrict__ slab, float* __restrict__ dw, int Co, int Ci, int T,
                                                               int nsplit, const float* __restrict__ bias_slab, float* __restrict__ db) {
    __shared__ float part[4][64];
    const int chunks = (Ci + 63) / 64;
    const int t = blockIdx.x % T, rest = blockIdx.x / T;
    const int co = rest / chunks, ci0 = (rest % chunks) * 64;
    const int nci = min(64, Ci - ci0);
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
    const size_t total = (size_t)Co * T * Ci;
    const int per = (nsplit + 3) / 4, k0 = q * per, k1 = min(nsplit, k0 + per);
    float s = 0.f;
    if (c < nci) {
        const size_t i = ((size_t)co * T + t) * Ci + ci0 + c;
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int k = k0;
        for (; k + 8 <= k1; k += 8)
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += slab[(size_t)(k + u) * total + i];
        for (int u = 0; k < k1; ++k, ++u) a[u] += slab[(size_t)k * total + i];
        s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    part[q][c] = s;
    __syncthreads();
    if (q == 0 && c < nci) dw[((size_t)co * Ci + ci0 + c) * T + t] = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
    if (db != nullptr && ci0 == 0 && t == 0 && threadIdx.x < 64) {   // wave 0: lanes stride over the splits, fixed shuffle tree
        float b = 0.f;
        for (int k = threadIdx.x; k < nsplit; k += 64) b += bias_slab[(size_t)k * Co + co];
        b = wave_sum(b);
        if (threadIdx.x == 0) db[co] = b;
    }
}

int g_force_bt = -1, g_force_wnbuf = -1, g_force_blocks_per_cu = -1;   // tuning aid (ssd_tune_set_wgrad)

struct WgradPlan {
    int bt, nbuf, fused, shape, tiles_co, tiles_ci, nsplit, m_per_split;      // shape: 0 = 4x8 patches, 1 = 1x38, 2 = 2x19
    size_t slab_floats, bias_floats;
};

int g_force_fused = -1;          // tuning aid: 0 = never use the fused 3x3 kernel, 1 = whenever applicable
int g_force_shape = -1;          // tuning aid: patch shape of the f32 fused kernel (0 = 4x8, 1 = 1x38, 2 = 2x19); -1 = least padding

WgradPlan plan_wgrad(const ssd_conv_geom* g, bool bf16 = false) {
    WgradPlan pl;
    const int T = g->R * g->S;
    const int M = g->N * g->Ho * g->Wo;
    // fused nine-tap kernel: measured 117-130 TFLOP/s on conv1_2..conv4_3 against 92-123 for one tap per block.  Its K loop
    // walks pixel patches, so maps that the patch grid does not tile pay for padded MFMAs: 4x8 patches waste 8 % on 75x75,
    // 11 % on 38x38 and 33 % on 19x19, where the 38-pixel shapes (one row of 38, or 2x19) waste 1 %, 0 % and 5 %.  The
    // shape with the least padding is taken; on a tie 2x19, then 1x38 (measured: 171 instead of 144 MFMAs per barrier pair
    // is worth +5 % on conv1_2, and 2x19 reads 2.2x the patch as halo where 1x38 reads 3.2x).  Above 12 % padding the
    // 128-wide one-tap kernel is as fast and the layer keeps the old path unless forced.
    static const int SHAPES[3][2] = {{4, 8}, {1, 38}, {2, 19}};
    double waste[3];
    for (int k = 0; k < 3; ++k)
        waste[k] = (double)ssd_cdiv(g->Ho, SHAPES[k][0]) * SHAPES[k][0] * ssd_cdiv(g->Wo, SHAPES[k][1]) * SHAPES[k][1] /
                   ((double)g->Ho * g->Wo);
    pl.shape = 0;
    if (!bf16 && g_force_shape < 0) {
        static const int PREF[3] = {2, 1, 0};
        pl.shape = PREF[0];
        for (int i = 1; i < 3; ++i)
            if (waste[PREF[i]] < waste[pl.shape] - 0.002) pl.shape = PREF[i];
    } else if (!bf16) {
        pl.shape = g_force_shape;
    }
    pl.fused = g->R == 3 && g->S == 3 && g->stride == 1 && g->dil == 1 && g->pad == 1 && g_force_fused != 0 &&
               (g_force_fused == 1 || bf16 || waste[pl.shape] <= 1.12);   // bf16: the fused kernel is 6x the f32 rate, patch waste is irrelevant
    // bf16 operands: the patch kernel also takes fc6 (3x3, padding = dilation = 4) and the 1x1 / stride-1 layers (fc7, seq8.0 ... seq11.0)
    if (bf16 && g_force_fused != 0 && g->stride == 1 && g->R == g->S &&
        ((g->R == 3 && g->dil == 4 && g->pad == 4) || (g->R == 1 && g->pad == 0)))
        pl.fused = true;
    if (pl.fused) {
        pl.bt = 64; pl.nbuf = 1;
        pl.tiles_co = ssd_cdiv(g->Co, 64);
        pl.tiles_ci = ssd_cdiv(g->Ci, 64);
        const int npatch = (bf16 && g->R == 1) ? ssd_cdiv(M, 128)                    // the bf16 patch kernel's 1x1 form: 128 consecutive pixels
                                               : g->N * ssd_cdiv(g->Ho, SHAPES[pl.shape][0]) * ssd_cdiv(g->Wo, SHAPES[pl.shape][1]);
        const int per_split = pl.tiles_co * pl.tiles_ci;
        const int bpc = g_force_blocks_per_cu > 0 ? g_force_blocks_per_cu : 2;       // exactly the 2 resident blocks per CU: no tail round (3 for the one-tap bf16 form, whose registers would allow it, measured no faster: fc7 0.103 -> 0.108 ms)
        // (rounded DOWN: 48 tiles per split -- the c_7 head, 150 x 1024 channels -- gave 11 splits = 528 blocks, sixteen of them a second
        // round behind the 512 resident ones; 10 splits = 480 blocks is one round)
        int ns = (256 * bpc) / per_split;
        const int max_by_m = npatch / 4 > 0 ? npatch / 4 : 1;                        // at least 4 patches per block
        if (ns > max_by_m) ns = max_by_m;
        if (ns > 1024) ns = 1024;
        if (ns < 1) ns = 1;
        pl.m_per_split = ssd_cdiv(npatch, ns);                                       // patches per split
        pl.nsplit = ssd_cdiv(npatch, pl.m_per_split);
        pl.slab_floats = (size_t)pl.nsplit * g->Co * T * g->Ci;
        pl.bias_floats = (size_t)pl.nsplit * g->Co;
        return pl;
    }
    // 128x128 tiles when both channel counts fill them; 64x64 otherwise (also for Co = 150, where
    // two 128-row tiles would be 41 % padding against 22 % with three 64-row tiles)
    pl.bt = (g->Co > 64 && g->Ci > 64) ? 128 : 64;
    if (pl.bt == 128 && ssd_cdiv(g->Co, 128) * 128 * 100 > ssd_cdiv(g->Co, 64) * 64 * 115) pl.bt = 64;
    if (g_force_bt == 64 || g_force_bt == 128) pl.bt = g_force_bt;
    pl.nbuf = (g_force_wnbuf == 1 || g_force_wnbuf == 2) ? g_force_wnbuf : 1;
    // measured (tools/conv_bench.py): BT=128 peaks at ~9 blocks per CU of split-K work (3 resident x 3 rounds),
    // BT=64 at ~18; one LDS stage beats two (occupancy).  A variant that loaded the MFMA operands straight
    // from global memory into registers (no LDS, no barrier) and one that split each K step across the
    // block's waves (one ds_read_b64 per two MFMAs at the 64x64 granularity) both measured 10-25 % slower
    // than these and were dropped.
    const int target_per_cu = g_force_blocks_per_cu > 0 ? g_force_blocks_per_cu : (pl.bt == 128 ? 9 : 18);
    pl.tiles_co = ssd_cdiv(g->Co, pl.bt);
    pl.tiles_ci = ssd_cdiv(g->Ci, pl.bt);
    const int per_split = T * pl.tiles_co * pl.tiles_ci;
    // split-K so that the grid is a few rounds of resident blocks; at least 8 K steps per block, at most 1024 slabs
    int ns = ssd_cdiv(256 * target_per_cu, per_split);
    const int max_by_m = M / (WBK * 8) > 0 ? M / (WBK * 8) : 1;
    if (ns > max_by_m) ns = max_by_m;
    if (ns > 1024) ns = 1024;
    if (ns < 1) ns = 1;
    int mps = ssd_cdiv(M, ns);
    mps = ssd_cdiv(mps, WBK) * WBK;
    pl.m_per_split = mps;
    pl.nsplit = ssd_cdiv(M, mps);
    pl.slab_floats = (size_t)pl.nsplit * g->Co * T * g->Ci;
    pl.bias_floats = (size_t)pl.nsplit * g->Co;
    return pl;
}

}  // namespace

extern "C" size_t ssd_conv2d_wgrad_workspace(const ssd_conv_geom* g) {
    if (g == nullptr) return 0;
    const WgradPlan a = plan_wgrad(g, false), b = plan_wgrad(g, true);       // covers both entry points
    const size_t fa = a.slab_floats + a.bias_floats, fb = b.slab_floats + b.bias_floats;
    return (fa > fb ? fa : fb) * sizeof(float) + 256;
}

static int conv2d_wgrad_impl(const float* x, const float* dy, int ldy, float* dw_oihw, float* dbias,
                             const ssd_conv_geom* g, void* workspace, size_t workspace_bytes, void* stream, bool bf16, bool in_bf16 = false) {
    if (!g || !x || !dy || !dw_oihw || !workspace) return SSD_ERR_NULL;
    if (g->Ci % 4 != 0 || ldy % 4 != 0 || ldy < g->Co) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(x) || !ssd_aligned16(dy) || !ssd_aligned16(workspace)) return SSD_ERR_ALIGN;
    const WgradPlan pl = plan_wgrad(g, bf16);
    if (workspace_bytes < (pl.slab_floats + pl.bias_floats) * sizeof(float) + 256) return SSD_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    WgradParams p{};
    p.x = x; p.dy = dy;
    {
        const size_t es = in_bf16 ? 2 : 4;
        const size_t xb = (size_t)g->N * g->H * g->W * g->Ci * es, yb = (size_t)g->N * g->Ho * g->Wo * ldy * es;
        if (xb >= 0xF0000000ull || yb >= 0xF0000000ull) return SSD_ERR_BAD_SHAPE;   // 32-bit buffer offsets
        p.x_bytes = (unsigned)xb; p.dy_bytes = (unsigned)yb;
    }
    p.slab = reinterpret_cast<float*>(workspace);
    p.bias_slab = dbias ? p.slab + pl.slab_floats : nullptr;
    p.H = g->H; p.W = g->W; p.Ci = g->Ci; p.Ho = g->Ho; p.Wo = g->Wo; p.Co = g->Co; p.ldy = ldy;
    p.R = g->R; p.S = g->S; p.stride = g->stride; p.pad = g->pad; p.dil = g->dil;
    p.M = g->N * g->Ho * g->Wo; p.m_per_split = pl.m_per_split; p.nsplit = pl.nsplit;
    p.tiles_co = pl.tiles_co; p.tiles_ci = pl.tiles_ci;
    const int T = g->R * g->S;
    const int nblk = T * pl.tiles_co * pl.tiles_ci * pl.nsplit;
    if (in_bf16 && !(pl.fused && bf16)) return SSD_ERR_BAD_SHAPE;       // bf16 tensors: the patch kernel only
    if (pl.fused && bf16) {
        const dim3 grid(pl.tiles_co * pl.tiles_ci * pl.nsplit);
        const int form = g->R == 1 ? 2 : (g->dil == 4 ? 1 : 0);
        if (in_bf16) {
            if (form == 0) hipLaunchKernelGGL((wgrad3x3_bf16_kernel<true, 9, 1>), grid, dim3(256), 0, st, p);
            else if (form == 1) hipLaunchKernelGGL((wgrad3x3_bf16_kernel<true, 9, 4>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((wgrad3x3_bf16_kernel<true, 1, 1>), grid, dim3(256), 0, st, p);
        } else {
            if (form == 0) hipLaunchKernelGGL((wgrad3x3_bf16_kernel<false, 9, 1>), grid, dim3(256), 0, st, p);
            else if (form == 1) hipLaunchKernelGGL((wgrad3x3_bf16_kernel<false, 9, 4>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((wgrad3x3_bf16_kernel<false, 1, 1>), grid, dim3(256), 0, st, p);
        }
    } else if (pl.fused) {
        const dim3 grid(pl.tiles_co * pl.tiles_ci * pl.nsplit);
        if (pl.shape == 1) hipLaunchKernelGGL((wgrad3x3_kernel<1, 38>), grid, dim3(256), 0, st, p);
        else if (pl.shape == 2) hipLaunchKernelGGL((wgrad3x3_kernel<2, 19>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((wgrad3x3_kernel<4, 8>), grid, dim3(256), 0, st, p);
    } else if (pl.bt == 128) {
        if (pl.nbuf == 2) hipLaunchKernelGGL((wgrad_kernel<128, 2>), dim3(nblk), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((wgrad_kernel<128, 1>), dim3(nblk), dim3(256), 0, st, p);
    } else {
        if (pl.nbuf == 2) hipLaunchKernelGGL((wgrad_kernel<64, 2>), dim3(nblk), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((wgrad_kernel<64, 1>), dim3(nblk), dim3(256), 0, st, p);
    }
    SSD_CHECK_LAUNCH();
    if (T > 49) return SSD_ERR_BAD_SHAPE;                      // the reduction's LDS tile holds up to 7x7 taps
    const int rblocks = g->Co * ssd_cdiv(g->Ci, 64);
    if (rblocks < 512 && pl.nsplit >= 32)                       // few blocks, long sums: one block per tap as well
        hipLaunchKernelGGL(wgrad_reduce_tap_kernel, dim3(rblocks * T), dim3(256), 0, st, p.slab, dw_oihw, g->Co, g->Ci, T, pl.nsplit,
                           dbias ? p.bias_slab : nullptr, dbias);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rblocks), dim3(256), 0, st, p.slab, dw_oihw, g->Co, g->Ci, T,
                           pl.nsplit, dbias ? p.bias_slab : nullptr, dbias);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

extern "C" int ssd_conv2d_wgrad_tile(const ssd_conv_geom* g, int* bt, int* nsplit) {
    if (!g || !bt || !nsplit) return SSD_ERR_NULL;
    const WgradPlan pl = plan_wgrad(g);
    *bt = pl.fused ? 3 : pl.bt;
    *nsplit = pl.nsplit;
    return SSD_OK;
}

// Tuning aid: force tile edge (64/128), LDS stage count (1/2) and split-K target in blocks per CU; -1 = automatic.
extern "C" int ssd_tune_set_wgrad(int bt, int nbuf, int blocks_per_cu) {
    g_force_fused = bt == 3 ? 1 : (bt == 64 || bt == 128 ? 0 : -1);       // bt = 3 selects the fused 3x3 kernel
    g_force_bt = bt;
    g_force_wnbuf = nbuf;
    g_force_blocks_per_cu = blocks_per_cu;
    return SSD_OK;
}

// Tuning aid: patch shape of the f32 fused kernel: -1 least padding, 0 = 4x8, 1 = 1x38, 2 = 2x19.
extern "C" int ssd_tune_set_wgrad_patch(int shape) {
    if (shape < -1 || shape > 2) return SSD_ERR_BAD_SHAPE;
    g_force_shape = shape;
    return SSD_OK;
}

extern "C" int ssd_conv2d_wgrad(const float* x, const float* dy, int ldy, float* dw_oihw, float* dbias,
                                const ssd_conv_geom* g, void* workspace, size_t workspace_bytes, void* stream) {
    return conv2d_wgrad_impl(x, dy, ldy, dw_oihw, dbias, g, workspace, workspace_bytes, stream, false);
}
// bf16-operand weight gradient: the fused nine-tap kernel multiplies bf16-rounded tiles (f32 accumulate);
// layers the fused kernel does not take (small maps, 1x1, dilated, strided) still run the f32 kernels.
extern "C" int ssd_conv2d_wgrad_bf16(const float* x, const float* dy, int ldy, float* dw_oihw, float* dbias,
                                     const ssd_conv_geom* g, void* workspace, size_t workspace_bytes, void* stream) {
    return conv2d_wgrad_impl(x, dy, ldy, dw_oihw, dbias, g, workspace, workspace_bytes, stream, true);
}
// bf16-tensor mode: x (N,H,W,Ci) and dy (N,Ho,Wo,ldy) are bf16; 3x3 / stride 1 / pad 1 / dilation 1 layers (the fused nine-tap kernel);
// SSD_ERR_BAD_SHAPE for any other geometry.  dw / dbias f32.
extern "C" int ssd_conv3x3_wgrad_bf16t(const void* x_bf16, const void* dy_bf16, int ldy, float* dw_oihw, float* dbias,
                                       const ssd_conv_geom* g, void* workspace, size_t workspace_bytes, void* stream) {
    return conv2d_wgrad_impl(static_cast<const float*>(x_bf16), static_cast<const float*>(dy_bf16), ldy, dw_oihw, dbias, g, workspace,
                             workspace_bytes, stream, true, true);
}
