// Implicit-GEMM convolution on the f32 MFMA (v_mfma_f32_32x32x2_f32), NHWC.
//
// Replaces nn.Conv2d(+ReLU) forward (Model.py:135-184) and its autograd
// data-gradient (train_function.py:94).  One kernel serves both directions:
//
//   out[m][n] = epi( sum_tap sum_c  A[gather(m,tap)][c] * Wt[n][tap][c] )
//
// forward : A = x,  Wt = [Co_pad][T][Ci], gather = (oh*stride - pad + r*dil, ...)
// dgrad   : A = dy, Wt = [Ci][T][Co_pad], gather = ((ih + pad - r*dil)/stride, ...) when divisible
//
// Tiling: 256 threads = 4 waves; block tile BM x BN, K step 32 (always inside one
// tap because every channel count on the path is a multiple of 32); each wave owns
// TM x TN accumulators of 32x32.  A and B tiles sit in LDS as [row][k] with a
// 36-float row stride: the global side reads whole 128-B channel runs per 8 lanes,
// the MFMA side reads 16 B per lane (ds_read_b128, conflict-free at stride 36) and
// feeds four 32x32x2 MFMAs per read.  The k order inside a K step is permuted
// (lane half h takes k = 8q+4h+e): A and B use the same permutation, so the sum is
// the same set of products.
#include <atomic>
#include "common.h"

namespace {

struct IgemmParams {
    const float* __restrict__ a;
    const float* __restrict__ w;
    const float* __restrict__ bias;
    float* __restrict__ out;
    const float* __restrict__ mask;
    unsigned a_bytes, w_bytes; // extents for the buffer descriptors (hardware range check -> 0)
    int Ha, Wa, Ca;          // A-side spatial size and channels (= K per tap)
    int Ho, Wo;              // output spatial size
    int Nout;                // valid output channels (store mask)
    int Nrows;               // valid weight rows (load mask)
    int ldo;                 // output row stride
    int R, S;
    int sm, sd;              // output coord multiplier; divisor (dgrad of strided conv)
    int off, dstep;          // tap 0 offset; per-tap step
    int M;                   // N*Ho*Wo
    int M_img;               // PARITY: images in the batch
    int tiles_m, tiles_n;
    int relu, accumulate;
    int ksplit, kt_per_split;   // split-K (small grids, deep K): blockIdx.y = split; partial tiles go to `slab`
    float* slab;                // [ksplit][M][Nout] raw partial sums, reduced by splitk_reduce_kernel
    float rcp_howo, rcp_wo;     // 1 / (Ho*Wo), 1 / Wo for div_small_q (set by launch_igemm)
    int nbatch;                 // > 1: blockIdx.z-th problem of a batch of equal-shaped GEMMs (the 16 Winograd planes)
    size_t batch_a, batch_w, batch_out;      // element strides between the problems of a batch
    // batched, no split-K: a unit = (problem, part of its row tiles) runs on ONE XCD (1-D grid, block ids of one residue mod 8), so a
    // filter plane leaves HBM once per part instead of once per XCD; units = 0: the 3-D grid (x = tile, y = K slice, z = problem)
    int units, msplit, mper, bpu;
    // PARITY (data gradient of a stride-2 convolution, dilation 1): the M rows are the pixels of dx grouped by the parity (a, b) of
    // their coordinates -- class c = 2a + b holds the rows par_base[c] .. of (n, i, j) -> pixel (2i + a, 2j + b), padded to whole
    // 64-row tiles -- so that every block sees ONE class and multiplies only the taps of that parity (1, 2, 2 or 4 of the 9; the
    // plain kernel multiplies zeros for the others).  par_hc[a] / par_wc[b]: rows / columns of dx with that parity.
    int par_base[5], par_hc[2], par_wc[2];
    unsigned long long* stamps; // diagnostic (ssd_tune_set_igemm_stamps): shader-clock stamps of every 64th block, else NULL
};

constexpr int BK = 32;
constexpr int LDS_LD = 36;
constexpr unsigned OOB = 0xFFFFFF00u;    // beyond any descriptor extent: the load returns 0

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 buf_load16(__amdgpu_buffer_rsrc_t srd, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, (int)voff, (int)soff, 0));
}

// The K loop walks taps (r,s) outermost and 32-channel chunks innermost.  Per tap every thread
// recomputes the byte offset of its A rows once (or OOB when the tap falls outside the image:
// the buffer range check then returns zeros, no branch); inside a tap only a scalar offset moves.
// NBUF = 2: the next tile is written into the other LDS stage while this one is multiplied
// (one barrier per K step); NBUF = 1 keeps a single stage (two barriers) where LDS is scarce.
// q = m / d for 0 <= m < 2^31 when the quotient is below 2^21 (image index, output row): float estimate, one correction.
// Vector instructions are scarce while the resident waves' MFMA loops hold the SIMD (in-kernel stamps: ~200 cycles per
// instruction), so the prologue avoids the ~35-instruction integer division sequence.
__device__ __forceinline__ int div_small_q(int m, int d, float rcp) {
    int q = (int)((float)m * rcp);
    const int r = m - q * d;
    q += r < 0 ? -1 : (r >= d ? 1 : 0);
    return q;
}

// Epilogue shared by the 32x32-accumulator kernels (f32, bf16-operand, f32x3): C/D map col = lane&31,
// row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).  out = [accumulate: out +] acc (+ bias) -> ReLU -> ReLU mask.
template <int BM, int TM, int TN>
__device__ __forceinline__ void igemm_epilogue(const IgemmParams& p, const f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn,
                                               int lr, int lh) {
    // Full tiles (all but the last tile row) take straight-line paths: no per-element bounds test, every load of a group of
    // eight rows in flight before the first use, ~10 vector instructions per stored value less than the general loop below.
    const bool full_m = m0 + BM <= p.M;                    // uniform
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + lr;
            const bool n_ok = n < p.Nout;
            const float bv = (p.bias != nullptr && n_ok) ? p.bias[n] : 0.f;
            if (full_m) {
                if (n_ok) {
                    const size_t row = (size_t)p.ldo;
                    float* po = p.out + (size_t)(m0 + (wm * TM + i) * 32 + 4 * lh) * row + n;
                    if (p.mask == nullptr && !p.accumulate) {            // forward: bias (+ ReLU)
                        if (p.relu) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const float v = acc[i][j][r] + bv;
                                po[(size_t)((r & 3) + 8 * (r >> 2)) * row] = v < 0.f ? 0.f : v;      // NaN stays NaN, like torch.relu
                            }
                        } else {
#pragma unroll
                            for (int r = 0; r < 16; ++r) po[(size_t)((r & 3) + 8 * (r >> 2)) * row] = acc[i][j][r] + bv;
                        }
                    } else {                                               // dgrad: (+ previous dx) (* ReLU mask)
                        const float* pm = p.mask != nullptr ? p.mask + (po - p.out) : nullptr;
#pragma unroll
                        for (int half = 0; half < 2; ++half) {
                            float prev[8], mk[8];
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                const size_t o = (size_t)(((half * 8 + q) & 3) + 8 * ((half * 8 + q) >> 2)) * row;
                                prev[q] = p.accumulate ? po[o] : 0.f;
                                mk[q] = pm != nullptr ? pm[o] : 1.f;
                            }
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                const int r = half * 8 + q;
                                float v = acc[i][j][r] + bv + prev[q];
                                if (p.relu) v = v < 0.f ? 0.f : v;
                                po[(size_t)((r & 3) + 8 * (r >> 2)) * row] = mk[q] > 0.f ? v : 0.f;
                            }
                        }
                    }
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (n_ok && m < p.M) {
                    const size_t idx = (size_t)m * p.ldo + n;
                    float v = acc[i][j][r] + bv;
                    if (p.accumulate) v += p.out[idx];
                    if (p.relu) v = v < 0.f ? 0.f : v;            // NaN stays NaN, like torch.relu
                    if (p.mask != nullptr) v = p.mask[idx] > 0.f ? v : 0.f;
                    p.out[idx] = v;
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int NBUF, bool BATCHED = false, bool PARITY = false>      // BATCHED: its own symbol, so profiles tell the Winograd GEMMs apart
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmParams p_in) {
    IgemmParams p = p_in;
    int unit_tile = -1;                                   // BATCHED with units: the logical tile of this block inside its problem
    if (BATCHED) {
        int batch = blockIdx.z;
        if (p.units > 0) {
            const int idx = (int)(blockIdx.x >> 3), unit = (idx / p.bpu) * 8 + (int)(blockIdx.x & 7), inner = idx % p.bpu;
            if (unit >= p.units) return;                  // uniform: the grid is padded to whole rounds of 8 units
            batch = unit / p.msplit;
            const int part = unit - batch * p.msplit, tm = part * p.mper + inner / p.tiles_n;
            if (tm >= min((part + 1) * p.mper, p.tiles_m)) return;     // uniform: the last part may be shorter
            unit_tile = tm * p.tiles_n + inner % p.tiles_n;
        }
        p.a += (size_t)batch * p.batch_a;
        p.w += (size_t)batch * p.batch_w;
        p.out += (size_t)batch * p.batch_out;
    }
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_ROWS = BM / 32, B_ROWS = BN / 32;
    constexpr int STAGE = (BM + BN) * LDS_LD;
    static_assert(WM * WN == 4, "4 waves");
    __shared__ __attribute__((aligned(16))) float lds[NBUF * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool stamp = p.stamps != nullptr && (blockIdx.x & 63) == 0 && tid == 0;
    unsigned long long t_start = 0, t_loop = 0, t_epi = 0;
    if (stamp) t_start = __builtin_readcyclecounter();
    const int wm = wave / WN, wn = wave % WN;
    const int nblk = p.tiles_m * p.tiles_n;
    const int lid = unit_tile >= 0 ? unit_tile : xcd_swizzle(blockIdx.x, nblk);
    const int tile_m = lid / p.tiles_n, tile_n = lid % p.tiles_n;   // n fastest: neighbours share A rows
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int chunk = tid & 7, row0 = tid >> 3;

    const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)p.w_bytes, 0x00020000);

    // ---- per-thread A rows: decompose m once --------------------------------
    int a_h[A_ROWS], a_w[A_ROWS];
    unsigned a_base[A_ROWS], voff_a[A_ROWS];
    const int HoWo = p.Ho * p.Wo;
    // PARITY: the class of this block's rows (uniform), its size and where it starts
    int pc = 0;
    if (PARITY) pc = (m0 >= p.par_base[1]) + (m0 >= p.par_base[2]) + (m0 >= p.par_base[3]);
    const int pa = pc >> 1, pb = pc & 1, pHc = PARITY ? p.par_hc[pa] : 1, pWc = PARITY ? p.par_wc[pb] : 1;
    const int pcnt = PARITY ? (p.par_base[pc + 1] - p.par_base[pc]) : 0;      // padded; rows beyond N * pHc * pWc are idle
    const int pvalid = PARITY ? (p.M_img * pHc * pWc) : 0, pm0 = PARITY ? m0 - p.par_base[pc] : 0;
    const float prcp_hw = PARITY ? 1.0f / (float)(pHc * pWc) : 0.f, prcp_w = PARITY ? 1.0f / (float)pWc : 0.f;
    (void)pcnt;
#pragma unroll
    for (int j = 0; j < A_ROWS; ++j) {
        int n, oh, ow;
        bool ok;
        if (PARITY) {
            const int ml = pm0 + row0 + 32 * j;
            ok = ml < pvalid;
            const int mm = ok ? ml : 0;
            n = div_small_q(mm, pHc * pWc, prcp_hw);
            const int rem = mm - n * pHc * pWc, i = div_small_q(rem, pWc, prcp_w);
            oh = 2 * i + pa;
            ow = 2 * (rem - i * pWc) + pb;
        } else {
            const int m = m0 + row0 + 32 * j;
            ok = m < p.M;
            const int mm = ok ? m : 0;
            n = div_small_q(mm, HoWo, p.rcp_howo);
            const int rem = mm - n * HoWo;
            oh = div_small_q(rem, p.Wo, p.rcp_wo);
            ow = rem - oh * p.Wo;
        }
        a_h[j] = ok ? oh * p.sm + p.off : -(1 << 24);       // never in range for a row past M
        a_w[j] = ow * p.sm + p.off;
        a_base[j] = (unsigned)n * (unsigned)(p.Ha * p.Wa * p.Ca) * 4u + chunk * 16u;
    }
    const int T = p.R * p.S;
    unsigned voff_b[B_ROWS];
#pragma unroll
    for (int j = 0; j < B_ROWS; ++j) {
        const int n = n0 + row0 + 32 * j;
        voff_b[j] = n < p.Nrows ? ((unsigned)n * (unsigned)(T * p.Ca)) * 4u + chunk * 16u : OOB;
    }

    auto tap_offsets = [&](int r, int s) {
        const int dh = r * p.dstep, dw = s * p.dstep;
#pragma unroll
        for (int j = 0; j < A_ROWS; ++j) {
            int th = a_h[j] + dh, tw = a_w[j] + dw;
            bool ok = th >= 0 && tw >= 0;
            if (p.sd == 2) {                                 // dgrad of a stride-2 conv (uniform branch)
                ok = ok && ((th | tw) & 1) == 0;
                th >>= 1;
                tw >>= 1;
            } else if (p.sd > 2) {
                ok = ok && (th % p.sd == 0) && (tw % p.sd == 0);
                th /= p.sd;
                tw /= p.sd;
            }
            ok = ok && th < p.Ha && tw < p.Wa;
            voff_a[j] = ok ? a_base[j] + (unsigned)((th * p.Wa + tw) * p.Ca) * 4u : OOB;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int kc = p.Ca / BK;          // K steps per tap
    // this block's K steps: all of them, or the blockIdx.y-th slice when the launch is split-K
    const int kt_begin = p.ksplit > 1 ? (int)blockIdx.y * p.kt_per_split : 0;
    // PARITY: only the taps r = r_par, r_par + 2, ... / s = s_par, ... reach this class: (oh + off - r) must be even (dstep = -1)
    const int r_par = PARITY ? ((pa + p.off) & 1) : 0, s_par = PARITY ? ((pb + p.off) & 1) : 0;
    const int KT = PARITY ? ((p.R - r_par + 1) >> 1) * ((p.S - s_par + 1) >> 1) * kc
                          : (p.ksplit > 1 ? min(T * kc - kt_begin, p.kt_per_split) : T * kc);
    f32x4 ra[A_ROWS], rb[B_ROWS];
    int c_nxt = PARITY ? 0 : kt_begin % kc, r_nxt = PARITY ? r_par : (kt_begin / kc) / p.S, s_nxt = PARITY ? s_par : (kt_begin / kc) % p.S;
    unsigned soff_a = (unsigned)c_nxt * BK * 4, soff_b = PARITY ? (unsigned)((r_par * p.S + s_par) * p.Ca) * 4u : (unsigned)kt_begin * BK * 4;
    tap_offsets(r_nxt, s_nxt);

    auto issue_into = [&](f32x4* qa, f32x4* qb) {
#pragma unroll
        for (int j = 0; j < A_ROWS; ++j) qa[j] = buf_load16(srd_a, voff_a[j], soff_a);
#pragma unroll
        for (int j = 0; j < B_ROWS; ++j) qb[j] = buf_load16(srd_w, voff_b[j], soff_b);
        soff_a += BK * 4;
        soff_b += BK * 4;
        if (++c_nxt == kc) {                                 // next tile starts a new tap
            c_nxt = 0;
            soff_a = 0;
            if (PARITY) {                                    // the next tap of this parity (rows of the filter are [tap][Ca])
                s_nxt += 2;
                if (s_nxt >= p.S) { s_nxt = s_par; r_nxt += 2; }
                soff_b = (unsigned)((r_nxt * p.S + s_nxt) * p.Ca) * 4u;
            } else if (++s_nxt == p.S) { s_nxt = 0; ++r_nxt; }
            tap_offsets(r_nxt, s_nxt);
        }
    };
    auto store_from = [&](float* stage, const f32x4* qa, const f32x4* qb) {
        float* As = stage;
        float* Bs = stage + BM * LDS_LD;
#pragma unroll
        for (int j = 0; j < A_ROWS; ++j)
            *reinterpret_cast<f32x4*>(&As[(row0 + 32 * j) * LDS_LD + chunk * 4]) = qa[j];
#pragma unroll
        for (int j = 0; j < B_ROWS; ++j)
            *reinterpret_cast<f32x4*>(&Bs[(row0 + 32 * j) * LDS_LD + chunk * 4]) = qb[j];
    };
    auto issue_loads = [&]() { issue_into(ra, rb); };
    auto store_tile = [&](float* stage) { store_from(stage, ra, rb); };

    const int lr = lane & 31, lh = lane >> 5;
    const int a_rd = (wm * TM * 32 + lr) * LDS_LD + lh * 4;
    const int b_rd = BM * LDS_LD + (wn * TN * 32 + lr) * LDS_LD + lh * 4;

    issue_loads();
    store_tile(lds);
    __syncthreads();
    if (stamp) t_loop = __builtin_readcyclecounter();

    auto multiply = [&](const float* stage) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(stage + a_rd + i * 32 * LDS_LD + q * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(stage + b_rd + j * 32 * LDS_LD + q * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
        }
    };
    {
        int cur = 0;
        for (int kt = 0; kt < KT; ++kt) {
            const bool more = kt + 1 < KT;
            if (more) issue_loads();                  // global loads stay in flight under the MFMAs
            multiply(lds + (NBUF == 2 ? cur * STAGE : 0));
            if (NBUF == 2) {
                if (more) store_tile(lds + (cur ^ 1) * STAGE);   // other stage: last read one iteration ago
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

    if (stamp) t_epi = __builtin_readcyclecounter();
    // ---- epilogue: C/D map col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) --------
    if (p.ksplit > 1) {                                   // uniform: raw partial tile, finished by splitk_reduce_kernel
        float* slab = p.slab + ((size_t)blockIdx.z * p.ksplit + blockIdx.y) * p.M * p.Nout;      // [batch][split][M][N]
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + (wn * TN + j) * 32 + lr;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (n < p.Nout && m < p.M) slab[(size_t)m * p.Nout + n] = acc[i][j][r];
                }
            }
        return;
    }
    if (PARITY) {                                         // rows -> pixels of dx: (n, i, j) of the class -> (2i + a, 2j + b); + previous dx, ReLU mask
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = pm0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (ml >= pvalid) continue;
                const int n = div_small_q(ml, pHc * pWc, prcp_hw), rem = ml - n * pHc * pWc, ii = div_small_q(rem, pWc, prcp_w);
                const size_t row = (((size_t)n * p.Ho + 2 * ii + pa) * p.Wo + 2 * (rem - ii * pWc) + pb) * p.ldo;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int nn = n0 + (wn * TN + j) * 32 + lr;
                    if (nn < p.Nout) {
                        float v = acc[i][j][r];
                        if (p.accumulate) v += p.out[row + nn];
                        if (p.mask != nullptr) v = p.mask[row + nn] > 0.f ? v : 0.f;
                        p.out[row + nn] = v;
                    }
                }
            }
        return;
    }
    igemm_epilogue<BM, TM, TN>(p, acc, m0, n0, wm, wn, lr, lh);
    if (stamp) {
        unsigned long long* o = p.stamps + (size_t)(blockIdx.x >> 6) * 4;
        o[0] = t_start; o[1] = t_loop; o[2] = t_epi; o[3] = __builtin_readcyclecounter();
    }
}

int g_lds_pad = 0;            // tuning aid: extra dynamic LDS per block (caps the blocks resident per CU)
int g_batched_units = 1;      // tuning aid (ssd_tune_set_batched_units): 0 = batched GEMMs on the 3-D grid
int g_dgrad_parity = 1;       // tuning aid (ssd_tune_set_dgrad_parity): 0 = stride-2 data gradients on the plain kernel (3/4 of the taps multiply zeros)
unsigned long long* g_stamps = nullptr;   // diagnostic buffer (ssd_tune_set_igemm_stamps)
int g_force_ksplit = -1;      // tuning aid: 1 = never split K, k > 1 = always k slices (when a workspace is given); -1 = automatic

// out = [accumulate: out +] sum_split slab[split] (+ bias) -> ReLU -> ReLU mask; splits added in index order (reproducible)
__global__ void splitk_reduce_kernel(const float* __restrict__ slab, int ksplit, int M, int N, int ldo, const float* __restrict__ bias,
                                     float* __restrict__ out, const float* __restrict__ mask, int relu, int accumulate) {
    const size_t total = (size_t)M * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i % N);
        const size_t m = i / N;
        float v = slab[i];
        for (int k = 1; k < ksplit; ++k) v += slab[(size_t)k * total + i];
        if (bias != nullptr) v += bias[n];
        const size_t idx = m * ldo + n;
        if (accumulate) v += out[idx];
        if (relu) v = v < 0.f ? 0.f : v;
        if (mask != nullptr) v = mask[idx] > 0.f ? v : 0.f;
        out[idx] = v;
    }
}

// K slices for a launch of `blocks` 64x64 tiles with `kt` K steps each: only when the grid leaves most of the 256 CUs'
// ~1800 resident-block slots empty and every slice keeps at least 4 K steps; the slab is capped at 64 MB.
int pick_ksplit(int blocks, int kt, size_t tile_elems_total) {
    if (g_force_ksplit == 1) return 1;
    int k;
    if (g_force_ksplit > 1) k = g_force_ksplit;
    else {
        if (blocks >= 1024 || kt < 16) return 1;
        k = 1792 / blocks;
        if (k > 16) k = 16;
    }
    if (k > kt / 4) k = kt / 4;
    while (k > 1 && (size_t)k * tile_elems_total * 4 > ((size_t)64 << 20)) --k;
    return k < 2 ? 1 : k;
}

template <int BM, int BN, int WM, int WN, int NBUF, bool BATCHED = false, bool PARITY = false>
int launch_igemm(IgemmParams& p, hipStream_t st) {
    p.tiles_m = ssd_cdiv(p.M, BM);
    p.tiles_n = ssd_cdiv(p.Nout, BN);
    p.rcp_howo = 1.0f / (float)(p.Ho * p.Wo);
    p.rcp_wo = 1.0f / (float)p.Wo;
    const int ks = p.ksplit > 1 ? p.ksplit : 1;
    p.units = 0;
    if (BATCHED && ks == 1 && g_batched_units && p.tiles_m * p.tiles_n >= 8) {
        p.msplit = 1;                                     // units a multiple of 8: every XCD gets the same number
        for (int sp = 1; sp <= 8 && sp <= p.tiles_m; ++sp)
            if ((p.nbatch * sp) % 8 == 0) { p.msplit = sp; break; }
        p.mper = ssd_cdiv(p.tiles_m, p.msplit);
        p.msplit = ssd_cdiv(p.tiles_m, p.mper);           // no empty part
        p.units = p.nbatch * p.msplit;
        p.bpu = p.mper * p.tiles_n;
        hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, NBUF, BATCHED, PARITY>), dim3((unsigned)(ssd_cdiv(p.units, 8) * 8 * p.bpu)), dim3(256), g_lds_pad, st, p);
        SSD_CHECK_LAUNCH();
        return SSD_OK;
    }
    hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, NBUF, BATCHED, PARITY>), dim3(p.tiles_m * p.tiles_n, ks, BATCHED ? p.nbatch : 1), dim3(256),
                       g_lds_pad, st, p);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

// Tile choice: wide tiles when there are enough blocks to fill 256 CUs a few times over,
// smaller tiles for the deep / small layers so the grid still covers the chip.
// SSD_IGEMM_TILE=256x64|128x128|128x64|64x64 forces one configuration (tuning aid).
enum Tile { T256x64 = 0, T128x128 = 1, T128x64 = 2, T64x64 = 3, TAUTO = -1 };
int g_force_tile = TAUTO, g_force_nbuf = -1;      // tuning aid (ssd_tune_set); -1 = automatic

struct TileChoice { Tile tile; int nbuf; };
TileChoice pick_tile(int M, int Nout) {
    // Measured on MI355X at bs=32 (tools/conv_bench.py, gpurun_out/convbench1.log): on every layer of the
    // network the 64x64 tile with ONE LDS stage (18 KB, 70 registers -> 7 blocks per CU) beats the wider
    // tiles and the double-buffered variants (125-141 vs 95-130 TFLOP/s): with a 64-cycle MFMA, occupancy
    // hides the barrier and staging better than a second stage does, and small tiles quantise better.
    TileChoice c;
    (void)M; (void)Nout;
    c.tile = T64x64;
    if (g_force_tile != TAUTO) c.tile = (Tile)g_force_tile;
    c.nbuf = 1;
    if (g_force_nbuf == 1 || g_force_nbuf == 2) c.nbuf = g_force_nbuf;
    return c;
}

// K slices this launch would use (1 = none) for M x Nout outputs with Ca-channel taps
int plan_ksplit(int M, int Nout, int Ca, int taps) {
    const int blocks = ssd_cdiv(M, 64) * ssd_cdiv(Nout, 64), kt = taps * (Ca / BK);
    const int k = pick_ksplit(blocks, kt, (size_t)M * Nout);
    return k > 1 ? ssd_cdiv(kt, ssd_cdiv(kt, k)) : 1;            // no empty slice
}

int dispatch_igemm(IgemmParams& p, hipStream_t st, void* ws = nullptr, size_t ws_bytes = 0) {
    const TileChoice c = pick_tile(p.M, p.Nout);
    p.ksplit = 1;
    p.stamps = g_stamps;
    if (ws != nullptr && c.tile == T64x64) {
        const int k = plan_ksplit(p.M, p.Nout, p.Ca, p.R * p.S);
        if (k > 1 && (size_t)k * p.M * p.Nout * sizeof(float) <= ws_bytes) {
            p.ksplit = k;
            p.kt_per_split = ssd_cdiv(p.R * p.S * (p.Ca / BK), k);
            p.slab = static_cast<float*>(ws);
        }
    }
    int e;
    switch (c.tile) {
        case T256x64: e = c.nbuf == 2 ? launch_igemm<256, 64, 4, 1, 2>(p, st) : launch_igemm<256, 64, 4, 1, 1>(p, st); break;
        case T128x128: e = c.nbuf == 2 ? launch_igemm<128, 128, 2, 2, 2>(p, st) : launch_igemm<128, 128, 2, 2, 1>(p, st); break;
        case T128x64: e = c.nbuf == 2 ? launch_igemm<128, 64, 4, 1, 2>(p, st) : launch_igemm<128, 64, 4, 1, 1>(p, st); break;
        default: e = c.nbuf == 2 ? launch_igemm<64, 64, 2, 2, 2>(p, st) : launch_igemm<64, 64, 2, 2, 1>(p, st); break;
    }
    if (e != SSD_OK || p.ksplit <= 1) return e;
    const size_t total = (size_t)p.M * p.Nout;
    const int rb = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rb), dim3(256), 0, st, p.slab, p.ksplit, p.M, p.Nout, p.ldo, p.bias, p.out, p.mask,
                       p.relu, p.accumulate);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

// ---------------------------------------------------------------------------------------------
// bf16-operand variant (BASELINE.json configs[2]: "bf16 convs"): same implicit GEMM, same f32 global
// tensors, f32 accumulation and epilogue; the A and B tiles are rounded to bf16 on their way into LDS
// (v_cvt_pk_bf16_f32) and multiplied on v_mfma_f32_32x32x16_bf16 (16x the f32 MFMA rate).  LDS rows are
// [row][32 bf16] with an 80-byte stride (conflict-free ds_read_b128 of the 8 k values a lane needs).
// At this MFMA rate the kernel is bound by the L2 -> LDS stream, so the tile is 128 x 128 / 256 x 128
// (twice / 2.7x the FLOPs per loaded byte of the 64 x 64 tile the f32 kernel prefers).
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int LDH = 40;       // LDS row stride in bf16 elements (32 + 8 pad)

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void igemm_bf16_kernel(const IgemmParams p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_ROWS = BM / 32, B_ROWS = BN / 32;
    static_assert(WM * WN == 4, "4 waves");
    __shared__ __attribute__((aligned(16))) __bf16 lds[(BM + BN) * LDH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int nblk = p.tiles_m * p.tiles_n;
    const int lid = xcd_swizzle(blockIdx.x, nblk);
    const int tile_m = lid / p.tiles_n, tile_n = lid % p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int chunk = tid & 7, row0 = tid >> 3;

    const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)p.w_bytes, 0x00020000);

    int a_h[A_ROWS], a_w[A_ROWS];
    unsigned a_base[A_ROWS], voff_a[A_ROWS];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int j = 0; j < A_ROWS; ++j) {
        const int m = m0 + row0 + 32 * j;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n = div_small_q(mm, HoWo, p.rcp_howo), rem = mm - n * HoWo;
        const int oh = div_small_q(rem, p.Wo, p.rcp_wo), ow = rem - oh * p.Wo;
        a_h[j] = ok ? oh * p.sm + p.off : -(1 << 24);
        a_w[j] = ow * p.sm + p.off;
        a_base[j] = (unsigned)n * (unsigned)(p.Ha * p.Wa * p.Ca) * 4u + chunk * 16u;
    }
    const int T = p.R * p.S;
    unsigned voff_b[B_ROWS];
#pragma unroll
    for (int j = 0; j < B_ROWS; ++j) {
        const int n = n0 + row0 + 32 * j;
        voff_b[j] = n < p.Nrows ? ((unsigned)n * (unsigned)(T * p.Ca)) * 4u + chunk * 16u : OOB;
    }
    auto tap_offsets = [&](int r, int s) {
        const int dh = r * p.dstep, dw = s * p.dstep;
#pragma unroll
        for (int j = 0; j < A_ROWS; ++j) {
            int th = a_h[j] + dh, tw = a_w[j] + dw;
            bool ok = th >= 0 && tw >= 0;
            if (p.sd == 2) {
                ok = ok && ((th | tw) & 1) == 0;
                th >>= 1;
                tw >>= 1;
            } else if (p.sd > 2) {
                ok = ok && (th % p.sd == 0) && (tw % p.sd == 0);
                th /= p.sd;
                tw /= p.sd;
            }
            ok = ok && th < p.Ha && tw < p.Wa;
            voff_a[j] = ok ? a_base[j] + (unsigned)((th * p.Wa + tw) * p.Ca) * 4u : OOB;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int kc = p.Ca / BK;
    // this block's K steps: all of them, or the blockIdx.y-th slice of a split-K launch (small grids with a deep K loop: the tiny maps of
    // the aux blocks and heads -- one or two blocks walking 72 K steps were 60 us per layer, slower than the f32 path's split launches)
    const int kt_begin = p.ksplit > 1 ? (int)blockIdx.y * p.kt_per_split : 0;
    const int KT = p.ksplit > 1 ? min(T * kc - kt_begin, p.kt_per_split) : T * kc;
    f32x4 ra[A_ROWS], rb[B_ROWS];
    int c_nxt = kt_begin % kc, r_nxt = (kt_begin / kc) / p.S, s_nxt = (kt_begin / kc) % p.S;
    unsigned soff_a = (unsigned)c_nxt * BK * 4, soff_b = (unsigned)kt_begin * BK * 4;
    tap_offsets(r_nxt, s_nxt);
    auto issue_loads = [&]() {
#pragma unroll
        for (int j = 0; j < A_ROWS; ++j) ra[j] = buf_load16(srd_a, voff_a[j], soff_a);
#pragma unroll
        for (int j = 0; j < B_ROWS; ++j) rb[j] = buf_load16(srd_w, voff_b[j], soff_b);
        soff_a += BK * 4;
        soff_b += BK * 4;
        if (++c_nxt == kc) {
            c_nxt = 0;
            soff_a = 0;
            if (++s_nxt == p.S) { s_nxt = 0; ++r_nxt; }
            tap_offsets(r_nxt, s_nxt);
        }
    };
    auto to_bf16 = [](const f32x4 v) { return bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]}; };
    auto store_tile = [&]() {
#pragma unroll
        for (int j = 0; j < A_ROWS; ++j)
            *reinterpret_cast<bf16x4*>(&lds[(row0 + 32 * j) * LDH + chunk * 4]) = to_bf16(ra[j]);
#pragma unroll
        for (int j = 0; j < B_ROWS; ++j)
            *reinterpret_cast<bf16x4*>(&lds[(BM + row0 + 32 * j) * LDH + chunk * 4]) = to_bf16(rb[j]);
    };

    const int lr = lane & 31, lh = lane >> 5;
    // A operand of 32x32x16: lane (row lr, half lh) holds k = 8*lh .. 8*lh+7 of the 16-deep sub-step
    const __bf16* a_rd = lds + (wm * TM * 32 + lr) * LDH + lh * 8;
    const __bf16* b_rd = lds + (BM + wn * TN * 32 + lr) * LDH + lh * 8;

    issue_loads();
    store_tile();
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const bool more = kt + 1 < KT;
        if (more) issue_loads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(a_rd + i * 32 * LDH + ks * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(b_rd + j * 32 * LDH + ks * 16);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            store_tile();
            __syncthreads();
        }
    }

    if (p.ksplit > 1) {                                   // uniform: raw partial tile, finished by splitk_reduce_kernel
        float* slab = p.slab + (size_t)blockIdx.y * p.M * p.Nout;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + (wn * TN + j) * 32 + lr;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (n < p.Nout && m < p.M) slab[(size_t)m * p.Nout + n] = acc[i][j][r];
                }
            }
        return;
    }
    igemm_epilogue<BM, TM, TN>(p, acc, m0, n0, wm, wn, lr, lh);
}

// ---------------------------------------------------------------------------------------------
// "f32 from bf16 limbs" variant (opt-in, conv_dtype = "f32x3"): every f32 operand is split EXACTLY into three bf16
// limbs x = hi + mid + lo (8 + 8 + 8 significant bits) and a product block is formed from the six limb products
// whose weight is >= 2^-16 (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid) on v_mfma_f32_32x32x16_bf16 with f32
// accumulation; the dropped terms (mid*lo, lo*mid, lo*lo) are <= 2^-24 relative, i.e. at f32 rounding level.
// Six bf16 MFMAs cost 6/16 of one f32 MFMA block.  The activation tile is split on its way into LDS (VALU),
// the weights arrive pre-split (three bf16 planes, ssd_weight_split_bf16x3).
// ---------------------------------------------------------------------------------------------
struct X3Params {
    IgemmParams g;
    const __bf16* __restrict__ w3;      // [3][Nrows_alloc][T][Ca] bf16 limb planes of the weights
    unsigned plane_bytes;               // bytes of one plane
};

__device__ __forceinline__ void split3(const f32x4 v, bf16x4& hi, bf16x4& mid, bf16x4& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 h = (__bf16)v[e];
        const float r1 = v[e] - (float)h;          // exact: the difference fits in 16 significant bits
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;            // exact, <= 8 significant bits
        hi[e] = h; mid[e] = m; lo[e] = (__bf16)r2;
    }
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void igemm_x3_kernel(const X3Params q) {
    const IgemmParams& p = q.g;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_ROWS = BM / 32, B_ROWS = BN / 32;
    constexpr int PLANE_A = BM * LDH, PLANE_B = BN * LDH;
    static_assert(WM * WN == 4, "4 waves");
    __shared__ __attribute__((aligned(16))) __bf16 lds[3 * (PLANE_A + PLANE_B)];
    __bf16* const As = lds;                         // [3][BM][LDH]
    __bf16* const Bs = lds + 3 * PLANE_A;           // [3][BN][LDH]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int nblk = p.tiles_m * p.tiles_n;
    const int lid = xcd_swizzle(blockIdx.x, nblk);
    const int tile_m = lid / p.tiles_n, tile_n = lid % p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int chunk = tid & 7, row0 = tid >> 3;

    const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(q.w3), 0, (int)(3u * q.plane_bytes), 0x00020000);

    int a_h[A_ROWS], a_w[A_ROWS];
    unsigned a_base[A_ROWS], voff_a[A_ROWS];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int j = 0; j < A_ROWS; ++j) {
        const int m = m0 + row0 + 32 * j;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n = div_small_q(mm, HoWo, p.rcp_howo), rem = mm - n * HoWo;
        const int oh = div_small_q(rem, p.Wo, p.rcp_wo), ow = rem - oh * p.Wo;
        a_h[j] = ok ? oh * p.sm + p.off : -(1 << 24);
        a_w[j] = ow * p.sm + p.off;
        a_base[j] = (unsigned)n * (unsigned)(p.Ha * p.Wa * p.Ca) * 4u + chunk * 16u;
    }
    const int T = p.R * p.S;
    unsigned voff_b[B_ROWS];
#pragma unroll
    for (int j = 0; j < B_ROWS; ++j) {
        const int n = n0 + row0 + 32 * j;
        voff_b[j] = n < p.Nrows ? ((unsigned)n * (unsigned)(T * p.Ca)) * 2u + chunk * 8u : OOB;      // bf16 planes: 2 bytes / element
    }
    auto tap_offsets = [&](int r, int s) {
        const int dh = r * p.dstep, dw = s * p.dstep;
#pragma unroll
        for (int j = 0; j < A_ROWS; ++j) {
            int th = a_h[j] + dh, tw = a_w[j] + dw;
            bool ok = th >= 0 && tw >= 0;
            if (p.sd == 2) {
                ok = ok && ((th | tw) & 1) == 0;
                th >>= 1;
                tw >>= 1;
            } else if (p.sd > 2) {
                ok = ok && (th % p.sd == 0) && (tw % p.sd == 0);
                th /= p.sd;
                tw /= p.sd;
            }
            ok = ok && th < p.Ha && tw < p.Wa;
            voff_a[j] = ok ? a_base[j] + (unsigned)((th * p.Wa + tw) * p.Ca) * 4u : OOB;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int kc = p.Ca / BK;
    const int KT = T * kc;
    f32x4 ra[A_ROWS];
    bf16x4 rb[3][B_ROWS];
    int c_nxt = 0, r_nxt = 0, s_nxt = 0;
    unsigned soff_a = 0, soff_b = 0;
    tap_offsets(0, 0);
    auto issue_loads = [&]() {
#pragma unroll
        for (int j = 0; j < A_ROWS; ++j) ra[j] = buf_load16(srd_a, voff_a[j], soff_a);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int j = 0; j < B_ROWS; ++j)
                rb[pl][j] = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(srd_w, (int)voff_b[j], (int)(soff_b + pl * q.plane_bytes), 0));
        soff_a += BK * 4;
        soff_b += BK * 2;
        if (++c_nxt == kc) {
            c_nxt = 0;
            soff_a = 0;
            if (++s_nxt == p.S) { s_nxt = 0; ++r_nxt; }
            tap_offsets(r_nxt, s_nxt);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int j = 0; j < A_ROWS; ++j) {
            bf16x4 hi, mid, lo;
            split3(ra[j], hi, mid, lo);
            const int o = (row0 + 32 * j) * LDH + chunk * 4;
            *reinterpret_cast<bf16x4*>(&As[o]) = hi;
            *reinterpret_cast<bf16x4*>(&As[PLANE_A + o]) = mid;
            *reinterpret_cast<bf16x4*>(&As[2 * PLANE_A + o]) = lo;
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int j = 0; j < B_ROWS; ++j)
                *reinterpret_cast<bf16x4*>(&Bs[pl * PLANE_B + (row0 + 32 * j) * LDH + chunk * 4]) = rb[pl][j];
    };

    const int lr = lane & 31, lh = lane >> 5;
    const __bf16* a_rd = As + (wm * TM * 32 + lr) * LDH + lh * 8;
    const __bf16* b_rd = Bs + (wn * TN * 32 + lr) * LDH + lh * 8;

    issue_loads();
    store_tile();
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const bool more = kt + 1 < KT;
        if (more) issue_loads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[3][TM], bf[3][TN];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[pl][i] = *reinterpret_cast<const bf16x8*>(a_rd + pl * PLANE_A + i * 32 * LDH + ks * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[pl][j] = *reinterpret_cast<const bf16x8*>(b_rd + pl * PLANE_B + j * 32 * LDH + ks * 16);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    // smallest terms first; limb index sum <= 2
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][i], bf[0][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[2][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[1][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[0][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[1][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
        if (more) {
            store_tile();
            __syncthreads();
        }
    }

    igemm_epilogue<BM, TM, TN>(p, acc, m0, n0, wm, wn, lr, lh);
}

// ---------------------------------------------------------------------------------------------
// Halo-tile kernel for the 3x3 / stride 1 / dilation 1 / padding 1 layers (forward AND data gradient: the latter is the
// same convolution with the taps mirrored).  M tile = a PH x PW patch of output pixels, N tile = BN channels.  Per
// 32-channel chunk the (PH+2) x (PW+2) input HALO is loaded and split into limb planes ONCE and serves all nine taps
// (tap = an LDS row offset), so the global A traffic and the limb-splitting VALU work drop 9x against the generic
// kernel, where they -- not the MFMA -- bound the "f32x3" and bf16 variants.  Only the weight tile (pre-split bf16
// planes, L2-resident) is re-staged per tap, double-buffered: one barrier per tap, one more per channel chunk.
// PLANES = 3: "f32x3" (six limb products per block); PLANES = 1: plain bf16 operands (plane 0 = RNE bf16 of the weight).
// ---------------------------------------------------------------------------------------------
template <int PH_, int PW_, int BN, int PLANES>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(const X3Params q) {
    const IgemmParams& p = q.g;
    constexpr int HH_ = PH_ + 2, HW_ = PW_ + 2, HPIX_ = HH_ * HW_, MPIX = PH_ * PW_;
    constexpr int WM = MPIX / 32, WN = 4 / WM, TN = BN / WN / 32;
    constexpr int A_LOADS = (HPIX_ + 31) / 32;                  // float4 loads per thread per chunk (8 lanes per halo pixel)
    constexpr int PLANE_A = HPIX_ * LDH, PLANE_B = BN * LDH;
    static_assert(WM * WN == 4 && TN >= 1 && BN * 4 == 256, "4 waves; one 16-byte weight load per thread and plane");
    __shared__ __attribute__((aligned(16))) __bf16 lds[PLANES * PLANE_A + 2 * PLANES * PLANE_B];
    __bf16* const As = lds;                                      // [PLANES][HPIX][LDH]
    __bf16* const Bs = lds + PLANES * PLANE_A;                   // [2][PLANES][BN][LDH]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;
    const int npw = (p.Wo + PW_ - 1) / PW_, nph = (p.Ho + PH_ - 1) / PH_;
    const int per_img = npw * nph;
    const int nblk = p.tiles_m * p.tiles_n;
    const int lid = xcd_swizzle(blockIdx.x, nblk);
    const int patch = lid / p.tiles_n, tile_n = lid % p.tiles_n;            // n fastest: neighbours share the halo in L2
    const int n0 = tile_n * BN;
    const int img = patch / per_img, prem = patch - img * per_img;
    const int oh0 = (prem / npw) * PH_, ow0 = (prem % npw) * PW_;

    const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(q.w3), 0, (int)(3u * q.plane_bytes), 0x00020000);

    // ---- A halo: pixel hp = (tid>>3) + 32 j, 16-byte chunk tid&7; offsets are fixed for the whole kernel -------------
    const int chunk = tid & 7;
    unsigned voff_a[A_LOADS];
#pragma unroll
    for (int j = 0; j < A_LOADS; ++j) {
        const int hp = (tid >> 3) + 32 * j;
        const int ih = oh0 - 1 + hp / HW_, iw = ow0 - 1 + hp % HW_;
        const bool ok = hp < HPIX_ && (unsigned)ih < (unsigned)p.Ha && (unsigned)iw < (unsigned)p.Wa;
        voff_a[j] = ok ? (unsigned)((img * p.Ha + ih) * p.Wa + iw) * (unsigned)p.Ca * 4u + chunk * 16u : OOB;
    }
    // ---- B tile: row n0 + (tid>>2), 8 k values (16 bytes) at chunk tid&3 -------------------------------------------------
    const int brow = tid >> 2, bchunk = tid & 3;
    const unsigned voff_b = (n0 + brow) < p.Nrows ? ((unsigned)(n0 + brow) * (unsigned)(9 * p.Ca)) * 2u + bchunk * 16u : OOB;

    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    f32x4 ra[A_LOADS];
    f32x4 rb[PLANES];                                            // 8 bf16 each
    const int KC = p.Ca / BK, NS = KC * 9;
    auto issue_a = [&](int kc) {
#pragma unroll
        for (int j = 0; j < A_LOADS; ++j) ra[j] = buf_load16(srd_a, voff_a[j], (unsigned)kc * BK * 4);
    };
    auto issue_b = [&](int st) {
        const int kc = st / 9, t = st - kc * 9;
        const unsigned so = (unsigned)(t * p.Ca + kc * BK) * 2u;
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl) rb[pl] = buf_load16(srd_w, voff_b, so + pl * q.plane_bytes);
    };
    auto store_a = [&]() {
#pragma unroll
        for (int j = 0; j < A_LOADS; ++j) {
            const int hp = (tid >> 3) + 32 * j;
            if (hp < HPIX_) {
                const int o = hp * LDH + chunk * 4;
                if constexpr (PLANES == 3) {
                    bf16x4 hi, mid, lo;
                    split3(ra[j], hi, mid, lo);
                    *reinterpret_cast<bf16x4*>(&As[o]) = hi;
                    *reinterpret_cast<bf16x4*>(&As[PLANE_A + o]) = mid;
                    *reinterpret_cast<bf16x4*>(&As[2 * PLANE_A + o]) = lo;
                } else {
                    *reinterpret_cast<bf16x4*>(&As[o]) = bf16x4{(__bf16)ra[j][0], (__bf16)ra[j][1], (__bf16)ra[j][2], (__bf16)ra[j][3]};
                }
            }
        }
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl)
            *reinterpret_cast<f32x4*>(&Bs[(buf * PLANES + pl) * PLANE_B + brow * LDH + bchunk * 8]) = rb[pl];
    };

    // this lane's output pixel inside the patch and its halo row for tap (0,0)
    const int pix = wm * 32 + lr, py = pix / PW_, px = pix % PW_;
    const __bf16* a_rd = As + (py * HW_ + px) * LDH + lh * 8;
    const __bf16* b_rd = Bs + (wn * TN * 32 + lr) * LDH + lh * 8;

    issue_a(0);
    issue_b(0);
    store_a();
    store_b(0);
    __syncthreads();
    for (int st = 0; st < NS; ++st) {
        const int kc = st / 9, t = st - kc * 9;
        const bool more = st + 1 < NS, new_chunk = t == 8 && kc + 1 < KC;
        if (more) issue_b(st + 1);
        if (new_chunk) issue_a(kc + 1);
        // tap (r,s) reads input (oh + off + r*dstep, ow + off + s*dstep); the halo origin is (oh0-1, ow0-1)
        const int r = t / 3, s2 = t - r * 3;
        const int hoff = ((1 + p.off + r * p.dstep) * HW_ + (1 + p.off + s2 * p.dstep)) * LDH;
        const __bf16* bb = b_rd + (st & 1) * PLANES * PLANE_B;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[PLANES], bf[PLANES][TN];
#pragma unroll
            for (int pl = 0; pl < PLANES; ++pl) {
                af[pl] = *reinterpret_cast<const bf16x8*>(a_rd + pl * PLANE_A + hoff + ks * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[pl][j] = *reinterpret_cast<const bf16x8*>(bb + pl * PLANE_B + j * 32 * LDH + ks * 16);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (PLANES == 3) {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bf[0][j], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[2][j], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[1][j], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[0][j], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[1][j], acc[j], 0, 0, 0);
                }
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0][j], acc[j], 0, 0, 0);
            }
        }
        if (more) store_b((st + 1) & 1);               // other buffer: last read one step ago
        __syncthreads();
        if (new_chunk) {
            store_a();
            __syncthreads();
        }
    }

    // ---- epilogue: this lane's accumulator rows are patch pixels (wm*32 + row) -------------------------------------------
    // Patches that lie inside the map take straight-line paths (no per-element bounds test; the loads of eight rows in flight
    // before their first use): vector instructions next to the resident MFMA loops are the expensive part of a block's edges.
    const bool inside = oh0 + PH_ <= p.Ho && ow0 + PW_ <= p.Wo;        // uniform
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + lr;
        const bool n_ok = n < p.Nout;
        const float bv = (p.bias != nullptr && n_ok) ? p.bias[n] : 0.f;
        if (inside) {
            if (n_ok) {
                const size_t row = (size_t)p.ldo;
                float* po = p.out + ((size_t)(img * p.Ho + oh0) * p.Wo + ow0) * row + n;
                size_t off[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pq = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    off[r] = ((size_t)(pq / PW_) * p.Wo + (pq % PW_)) * row;
                }
                if (p.mask == nullptr && !p.accumulate) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[j][r] + bv;
                        if (p.relu) v = v < 0.f ? 0.f : v;
                        po[off[r]] = v;
                    }
                } else {
                    const float* pm = p.mask != nullptr ? p.mask + (po - p.out) : nullptr;
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        float prev[8], mk[8];
#pragma unroll
                        for (int q2 = 0; q2 < 8; ++q2) {
                            prev[q2] = p.accumulate ? po[off[half * 8 + q2]] : 0.f;
                            mk[q2] = pm != nullptr ? pm[off[half * 8 + q2]] : 1.f;
                        }
#pragma unroll
                        for (int q2 = 0; q2 < 8; ++q2) {
                            float v = acc[j][half * 8 + q2] + bv + prev[q2];
                            if (p.relu) v = v < 0.f ? 0.f : v;
                            po[off[half * 8 + q2]] = mk[q2] > 0.f ? v : 0.f;
                        }
                    }
                }
            }
            continue;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pq = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int oh = oh0 + pq / PW_, ow = ow0 + pq % PW_;
            if (n_ok && oh < p.Ho && ow < p.Wo) {
                const size_t idx = ((size_t)(img * p.Ho + oh) * p.Wo + ow) * p.ldo + n;
                float v = acc[j][r] + bv;
                if (p.accumulate) v += p.out[idx];
                if (p.relu) v = v < 0.f ? 0.f : v;
                if (p.mask != nullptr) v = p.mask[idx] > 0.f ? v : 0.f;
                p.out[idx] = v;
            }
        }
    }
}

int g_halo = -1;             // tuning aid: 0 = never use the halo kernel, 1 = 8x8 patches, 2 = 8x16 patches; -1 = automatic

// returns SSD_OK when launched, 1 when the geometry is not a halo case
template <int PLANES>
int try_halo(X3Params& q, const ssd_conv_geom* g, hipStream_t st) {
    if (g_halo == 0 || g->R != 3 || g->S != 3 || g->stride != 1 || g->dil != 1 || g->pad != 1) return 1;
    IgemmParams& p = q.g;
    const int images = p.M / (p.Ho * p.Wo);
    int shape = g_halo;
    if (shape < 0) {
        // measured (tools/conv_bench.py x3 / halobf16): with three limbs the halo tile only wins on the 150^2 and 300^2 maps;
        // with one bf16 plane it wins down to 38^2.  8x16 patches beat 8x8 everywhere.
        if (p.Ho < 30 || p.Wo < (PLANES == 3 ? 64 : 30)) return 1;
        shape = 2;
    }
    p.tiles_n = ssd_cdiv(p.Nout, 64);
    if (shape == 2) {
        p.tiles_m = images * ssd_cdiv(p.Ho, 8) * ssd_cdiv(p.Wo, 16);
        hipLaunchKernelGGL((conv3x3_halo_kernel<8, 16, 64, PLANES>), dim3(p.tiles_m * p.tiles_n), dim3(256), 0, st, q);
    } else {
        p.tiles_m = images * ssd_cdiv(p.Ho, 8) * ssd_cdiv(p.Wo, 8);
        hipLaunchKernelGGL((conv3x3_halo_kernel<8, 8, 64, PLANES>), dim3(p.tiles_m * p.tiles_n), dim3(256), 0, st, q);
    }
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

int g_x3_tile = -1;          // tuning aid: 1 = 128x128, 2 = 128x64, 3 = 64x64; -1 = automatic

template <int BM, int BN, int WM, int WN>
int launch_igemm_x3(X3Params& q, hipStream_t st) {
    q.g.tiles_m = ssd_cdiv(q.g.M, BM);
    q.g.tiles_n = ssd_cdiv(q.g.Nout, BN);
    q.g.rcp_howo = 1.0f / (float)(q.g.Ho * q.g.Wo);
    q.g.rcp_wo = 1.0f / (float)q.g.Wo;
    hipLaunchKernelGGL((igemm_x3_kernel<BM, BN, WM, WN>), dim3(q.g.tiles_m * q.g.tiles_n), dim3(256), 0, st, q);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

int dispatch_igemm_x3(X3Params& q, hipStream_t st) {
    int t = g_x3_tile;
    if (t < 0) {
        // measured (tools/conv_bench.py x3): 128x128 for the 256/512-channel layers on large maps, 128x64 elsewhere
        const long b128 = (long)ssd_cdiv(q.g.M, 128) * ssd_cdiv(q.g.Nout, 128);
        t = (q.g.Nout >= 256 && b128 >= 1000) ? 1 : (q.g.M >= 4096 ? 2 : 3);
    }
    switch (t) {
        case 1: return launch_igemm_x3<128, 128, 2, 2>(q, st);
        case 2: return launch_igemm_x3<128, 64, 4, 1>(q, st);
        default: return launch_igemm_x3<64, 64, 2, 2>(q, st);
    }
}

__global__ void split_bf16x3_kernel(const float* __restrict__ w, __bf16* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = w[i];
        const __bf16 h = (__bf16)v;
        const float r1 = v - (float)h;
        const __bf16 m = (__bf16)r1;
        out[i] = h;
        out[n + i] = m;
        out[2 * n + i] = (__bf16)(r1 - (float)m);
    }
}

int g_bf16_tile = -1;        // tuning aid: 0 = 256x128, 1 = 128x128, 2 = 128x64, 3 = 64x64; -1 = automatic

template <int BM, int BN, int WM, int WN>
int launch_igemm_bf16(IgemmParams& p, hipStream_t st) {
    p.tiles_m = ssd_cdiv(p.M, BM);
    p.tiles_n = ssd_cdiv(p.Nout, BN);
    p.rcp_howo = 1.0f / (float)(p.Ho * p.Wo);
    p.rcp_wo = 1.0f / (float)p.Wo;
    hipLaunchKernelGGL((igemm_bf16_kernel<BM, BN, WM, WN>), dim3(p.tiles_m * p.tiles_n, p.ksplit > 1 ? p.ksplit : 1), dim3(256), 0, st, p);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

int plan_ksplit(int M, int Nout, int Ca, int taps);
// K slices of a bf16-operand launch on 128 x 128 tiles (1 = none): only where those tiles are chosen (>= 300 of them), fill less than two
// thirds of the 768 resident workgroups, and every slice keeps at least 32 K steps; the slab is capped at 128 MB
int plan_ksplit128(int M, int Nout, int Ca, int taps) {
    if (g_force_ksplit == 1) return 1;
    const long b128 = (long)ssd_cdiv(M, 128) * ssd_cdiv(Nout, 128);
    const int kt = taps * (Ca / BK);
    if (Nout <= 64 || b128 < 300 || b128 > 512 || kt < 64) return 1;
    int k = (int)(768 / b128);
    if (k > kt / 32) k = kt / 32;
    while (k > 1 && (size_t)k * M * Nout * 4 > ((size_t)128 << 20)) --k;
    return k < 2 ? 1 : ssd_cdiv(kt, ssd_cdiv(kt, k));
}

int dispatch_igemm_bf16(IgemmParams& p, hipStream_t st, void* ws = nullptr, size_t ws_bytes = 0) {
    int t = g_bf16_tile;
    p.ksplit = 1;
    if (t < 0) {
        // measured (tools/conv_bench.py bf16): 128x128 wins once it yields >= ~300 blocks (300-620 TFLOP/s),
        // 128x64 for the 64-channel outputs, 64x64 for the small maps
        const long b128 = (long)ssd_cdiv(p.M, 128) * ssd_cdiv(p.Nout, 128);
        t = p.Nout <= 64 ? 2 : (b128 >= 300 ? 1 : 3);
    }
    if (t == 1 && ws != nullptr) {
        // 128 x 128 tiles, three workgroups per CU = 768 resident: a grid that fills less than two thirds of them with a deep K loop
        // (fc6's data gradient in the bf16 mode: 364 blocks, 288 K steps) runs in K slices (round 4: 0.254 -> 0.202 ms with the reduction)
        const int k = plan_ksplit128(p.M, p.Nout, p.Ca, p.R * p.S);
        if (k > 1 && (size_t)k * p.M * p.Nout * sizeof(float) <= ws_bytes) {
            p.ksplit = k;
            p.kt_per_split = ssd_cdiv(p.R * p.S * (p.Ca / BK), k);
            p.slab = static_cast<float*>(ws);
        }
    }
    if (t == 3 && ws != nullptr) {                              // 64 x 64 tiles on a small grid with a deep K loop: K slices + the f32 path's reduction
        const int k = plan_ksplit(p.M, p.Nout, p.Ca, p.R * p.S);
        if (k > 1 && (size_t)k * p.M * p.Nout * sizeof(float) <= ws_bytes) {
            p.ksplit = k;
            p.kt_per_split = ssd_cdiv(p.R * p.S * (p.Ca / BK), k);
            p.slab = static_cast<float*>(ws);
        }
    }
    int e;
    switch (t) {
        case 0: e = launch_igemm_bf16<256, 128, 4, 1>(p, st); break;
        case 1: e = launch_igemm_bf16<128, 128, 2, 2>(p, st); break;
        case 2: e = launch_igemm_bf16<128, 64, 4, 1>(p, st); break;
        default: e = launch_igemm_bf16<64, 64, 2, 2>(p, st); break;
    }
    if (e != SSD_OK || p.ksplit <= 1) return e;
    const size_t total = (size_t)p.M * p.Nout;
    const int rb = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rb), dim3(256), 0, st, p.slab, p.ksplit, p.M, p.Nout, p.ldo, p.bias, p.out, p.mask,
                       p.relu, p.accumulate);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

int check_geom(const ssd_conv_geom* g) {
    if (g == nullptr) return SSD_ERR_NULL;
    if (g->N <= 0 || g->H <= 0 || g->W <= 0 || g->Ci <= 0 || g->Co <= 0 || g->R <= 0 || g->S <= 0 ||
        g->stride <= 0 || g->dil <= 0 || g->pad < 0)
        return SSD_ERR_BAD_SHAPE;
    const int ho = (g->H + 2 * g->pad - g->dil * (g->R - 1) - 1) / g->stride + 1;
    const int wo = (g->W + 2 * g->pad - g->dil * (g->S - 1) - 1) / g->stride + 1;
    if (ho != g->Ho || wo != g->Wo || ho <= 0 || wo <= 0) return SSD_ERR_BAD_SHAPE;
    if ((long)g->N * g->Ho * g->Wo >= (1L << 31) || (long)g->N * g->H * g->W >= (1L << 31)) return SSD_ERR_BAD_SHAPE;
    if (g->N >= (1 << 20) || g->H >= (1 << 20) || g->Ho >= (1 << 20)) return SSD_ERR_BAD_SHAPE;      // div_small_q quotients
    return SSD_OK;
}

}  // namespace

static int conv2d_fwd_impl(const float* x, const float* w_ohwi, const float* bias, float* y, int ldy,
                           const ssd_conv_geom* g, int relu, void* stream, bool bf16, int accumulate = 0,
                           void* ws = nullptr, size_t ws_bytes = 0) {
    if (int e = check_geom(g)) return e;
    if (!x || !w_ohwi || !y) return SSD_ERR_NULL;
    if (g->Ci % 32 != 0 || ldy < g->Co) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(x) || !ssd_aligned16(w_ohwi)) return SSD_ERR_ALIGN;
    IgemmParams p{};
    p.a = x; p.w = w_ohwi; p.bias = bias; p.out = y; p.mask = nullptr;
    {
        const size_t ab = (size_t)g->N * g->H * g->W * g->Ci * 4, wb = (size_t)g->Co * g->R * g->S * g->Ci * 4;
        if (ab >= 0xF0000000ull || wb >= 0xF0000000ull) return SSD_ERR_BAD_SHAPE;   // 32-bit buffer offsets
        p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb;
    }
    p.Ha = g->H; p.Wa = g->W; p.Ca = g->Ci; p.Ho = g->Ho; p.Wo = g->Wo;
    p.Nout = g->Co; p.Nrows = g->Co; p.ldo = ldy; p.R = g->R; p.S = g->S;
    p.sm = g->stride; p.sd = 1; p.off = -g->pad; p.dstep = g->dil;
    p.M = g->N * g->Ho * g->Wo; p.relu = relu; p.accumulate = accumulate;
    return bf16 ? dispatch_igemm_bf16(p, (hipStream_t)stream, ws, ws_bytes) : dispatch_igemm(p, (hipStream_t)stream, ws, ws_bytes);
}

// Workspace the split-K path of ssd_conv2d_fwd_ws (direction 0) / ssd_conv2d_dgrad_ws (direction 1) wants for this
// geometry; 0 = the launch is not split (large grids), and the _ws entry points accept workspace == NULL.
extern "C" size_t ssd_conv2d_igemm_workspace(const ssd_conv_geom* g, int direction) {
    if (check_geom(g) != SSD_OK) return 0;
    const int taps = g->R * g->S;
    if (direction == 0) {
        if (g->Ci % 32 != 0) return 0;
        const int M = g->N * g->Ho * g->Wo;
        const int k = plan_ksplit(M, g->Co, g->Ci, taps), k2 = plan_ksplit128(M, g->Co, g->Ci, taps), km = k > k2 ? k : k2;
        return km > 1 ? (size_t)km * M * g->Co * sizeof(float) : 0;     // (the bf16-operand path's 128 x 128 rule or the 64 x 64 rule, whichever wants more)
    }
    const int M = g->N * g->H * g->W, co_pad = (g->Co + 31) / 32 * 32;
    const int k = plan_ksplit(M, g->Ci, co_pad, taps), k2 = plan_ksplit128(M, g->Ci, co_pad, taps), km = k > k2 ? k : k2;
    return km > 1 ? (size_t)km * M * g->Ci * sizeof(float) : 0;
}
extern "C" int ssd_conv2d_fwd_ws(const float* x, const float* w_ohwi, const float* bias, float* y, int ldy, const ssd_conv_geom* g,
                                 int relu, void* workspace, size_t workspace_bytes, void* stream) {
    if (workspace != nullptr && !ssd_aligned16(workspace)) return SSD_ERR_ALIGN;
    return conv2d_fwd_impl(x, w_ohwi, bias, y, ldy, g, relu, stream, false, 0, workspace, workspace_bytes);
}

extern "C" int ssd_conv2d_fwd_accum(const float* x, const float* w_ohwi, const float* bias, float* y_inout, int ldy,
                                    const ssd_conv_geom* g, int relu, void* stream) {
    return conv2d_fwd_impl(x, w_ohwi, bias, y_inout, ldy, g, relu, stream, false, 1);
}
extern "C" int ssd_conv2d_fwd_accum_bf16(const float* x, const float* w_ohwi, const float* bias, float* y_inout, int ldy,
                                         const ssd_conv_geom* g, int relu, void* stream) {
    return conv2d_fwd_impl(x, w_ohwi, bias, y_inout, ldy, g, relu, stream, true, 1);
}

extern "C" int ssd_conv2d_fwd(const float* x, const float* w_ohwi, const float* bias, float* y, int ldy,
                              const ssd_conv_geom* g, int relu, void* stream) {
    return conv2d_fwd_impl(x, w_ohwi, bias, y, ldy, g, relu, stream, false);
}
// bf16-operand kernels with the split-K workspace of ssd_conv2d_igemm_workspace (same plan as the f32 kernels: small grids, deep K)
extern "C" int ssd_conv2d_fwd_bf16_ws(const float* x, const float* w_ohwi, const float* bias, float* y, int ldy, const ssd_conv_geom* g,
                                      int relu, void* workspace, size_t workspace_bytes, void* stream) {
    if (workspace != nullptr && !ssd_aligned16(workspace)) return SSD_ERR_ALIGN;
    return conv2d_fwd_impl(x, w_ohwi, bias, y, ldy, g, relu, stream, true, 0, workspace, workspace_bytes);
}
extern "C" int ssd_conv2d_fwd_bf16(const float* x, const float* w_ohwi, const float* bias, float* y, int ldy,
                                   const ssd_conv_geom* g, int relu, void* stream) {
    return conv2d_fwd_impl(x, w_ohwi, bias, y, ldy, g, relu, stream, true);
}

static int conv2d_dgrad_impl(const float* dy, int ldy, const float* w_ihwo, int Co_pad, float* dx,
                             const float* relu_mask, int accumulate, const ssd_conv_geom* g, void* stream, bool bf16,
                             void* ws = nullptr, size_t ws_bytes = 0) {
    if (int e = check_geom(g)) return e;
    if (!dy || !w_ihwo || !dx) return SSD_ERR_NULL;
    if (Co_pad % 32 != 0 || Co_pad < g->Co || ldy != Co_pad || g->Ci % 4 != 0) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(dy) || !ssd_aligned16(w_ihwo)) return SSD_ERR_ALIGN;
    IgemmParams p{};
    p.a = dy; p.w = w_ihwo; p.bias = nullptr; p.out = dx; p.mask = relu_mask;
    {
        const size_t ab = (size_t)g->N * g->Ho * g->Wo * Co_pad * 4, wb = (size_t)g->Ci * g->R * g->S * Co_pad * 4;
        if (ab >= 0xF0000000ull || wb >= 0xF0000000ull) return SSD_ERR_BAD_SHAPE;
        p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb;
    }
    p.Ha = g->Ho; p.Wa = g->Wo; p.Ca = Co_pad; p.Ho = g->H; p.Wo = g->W;
    p.Nout = g->Ci; p.Nrows = g->Ci; p.ldo = g->Ci; p.R = g->R; p.S = g->S;
    p.sm = 1; p.sd = g->stride; p.off = g->pad; p.dstep = -g->dil;
    p.M = g->N * g->H * g->W; p.relu = 0; p.accumulate = accumulate;
    if (!bf16 && g_dgrad_parity && g->stride == 2 && g->dil == 1 && g_force_tile == TAUTO) {
        // rows grouped by the parity of the dx pixel: every block multiplies only the taps that reach its class
        int base = 0;
        for (int c = 0; c < 4; ++c) {
            const int hc = (g->H + 1 - (c >> 1)) / 2, wc = (g->W + 1 - (c & 1)) / 2;
            p.par_base[c] = base;
            p.par_hc[c >> 1] = hc; p.par_wc[c & 1] = wc;
            base += (g->N * hc * wc + 63) / 64 * 64;
        }
        p.par_base[4] = base;
        p.M_img = g->N;
        p.M = base;
        p.ksplit = 1;
        p.stamps = nullptr;
        return launch_igemm<64, 64, 2, 2, 1, false, true>(p, (hipStream_t)stream);
    }
    return bf16 ? dispatch_igemm_bf16(p, (hipStream_t)stream, ws, ws_bytes) : dispatch_igemm(p, (hipStream_t)stream, ws, ws_bytes);
}

extern "C" int ssd_tune_set_dgrad_parity(int on) {
    g_dgrad_parity = on != 0;
    return SSD_OK;
}

extern "C" int ssd_conv2d_dgrad_ws(const float* dy, int ldy, const float* w_ihwo, int Co_pad, float* dx, const float* relu_mask,
                                   int accumulate, const ssd_conv_geom* g, void* workspace, size_t workspace_bytes, void* stream) {
    if (workspace != nullptr && !ssd_aligned16(workspace)) return SSD_ERR_ALIGN;
    return conv2d_dgrad_impl(dy, ldy, w_ihwo, Co_pad, dx, relu_mask, accumulate, g, stream, false, workspace, workspace_bytes);
}

extern "C" int ssd_conv2d_dgrad(const float* dy, int ldy, const float* w_ihwo, int Co_pad, float* dx,
                                const float* relu_mask, int accumulate, const ssd_conv_geom* g, void* stream) {
    return conv2d_dgrad_impl(dy, ldy, w_ihwo, Co_pad, dx, relu_mask, accumulate, g, stream, false);
}
extern "C" int ssd_conv2d_dgrad_bf16(const float* dy, int ldy, const float* w_ihwo, int Co_pad, float* dx,
                                     const float* relu_mask, int accumulate, const ssd_conv_geom* g, void* stream) {
    return conv2d_dgrad_impl(dy, ldy, w_ihwo, Co_pad, dx, relu_mask, accumulate, g, stream, true);
}
extern "C" int ssd_conv2d_dgrad_bf16_ws(const float* dy, int ldy, const float* w_ihwo, int Co_pad, float* dx, const float* relu_mask,
                                        int accumulate, const ssd_conv_geom* g, void* workspace, size_t workspace_bytes, void* stream) {
    if (workspace != nullptr && !ssd_aligned16(workspace)) return SSD_ERR_ALIGN;
    return conv2d_dgrad_impl(dy, ldy, w_ihwo, Co_pad, dx, relu_mask, accumulate, g, stream, true, workspace, workspace_bytes);
}
extern "C" int ssd_tune_set_igemm_bf16(int tile) {
    if (tile < -1 || tile > 3) return SSD_ERR_BAD_SHAPE;
    g_bf16_tile = tile;
    return SSD_OK;
}

extern "C" int ssd_conv2d_igemm_tile(const ssd_conv_geom* g, int direction, int* bm, int* bn) {
    if (int e = check_geom(g)) return e;
    if (!bm || !bn || (direction != 0 && direction != 1)) return SSD_ERR_NULL;
    const int M = direction == 0 ? g->N * g->Ho * g->Wo : g->N * g->H * g->W;
    const int Nout = direction == 0 ? g->Co : g->Ci;
    switch (pick_tile(M, Nout).tile) {
        case T256x64: *bm = 256; *bn = 64; break;
        case T128x128: *bm = 128; *bn = 128; break;
        case T128x64: *bm = 128; *bn = 64; break;
        default: *bm = 64; *bn = 64; break;
    }
    return SSD_OK;
}

// Tuning aid: force the igemm tile (0 = 256x64, 1 = 128x128, 2 = 128x64, 3 = 64x64) and LDS stage count; -1 = automatic.
extern "C" int ssd_tune_set_igemm(int tile, int nbuf) {
    if (tile < -1 || tile > 3 || nbuf < -1 || nbuf > 2 || nbuf == 0) return SSD_ERR_BAD_SHAPE;
    g_force_tile = tile;
    g_force_nbuf = nbuf;
    return SSD_OK;
}

// Tuning aid: K slices of the _ws entry points: -1 automatic, 1 never, k > 1 always k (bounded by K/4 steps and the workspace).
extern "C" int ssd_tune_set_igemm_splitk(int k) {
    if (k < -1 || k == 0 || k > 64) return SSD_ERR_BAD_SHAPE;
    g_force_ksplit = k;
    return SSD_OK;
}

// Diagnostic: device buffer of 4 x uint64 per 64 blocks (block start, main loop start, epilogue start, end; shader-clock
// counter) filled by every 64th block of the f32 igemm launches that follow; NULL switches it off.
extern "C" int ssd_tune_set_igemm_stamps(uint64_t* device_buffer) {
    g_stamps = reinterpret_cast<unsigned long long*>(device_buffer);
    return SSD_OK;
}

// Tuning aid: extra dynamic LDS bytes per igemm block (occupancy cap experiments); 0 = none.
extern "C" int ssd_tune_set_batched_units(int on) {
    g_batched_units = on != 0;
    return SSD_OK;
}
extern "C" int ssd_tune_set_igemm_lds_pad(int bytes) {
    if (bytes < 0 || bytes > 120 * 1024) return SSD_ERR_BAD_SHAPE;
    g_lds_pad = bytes;
    return SSD_OK;
}

constexpr int PROF_MAX = 1024;
static std::atomic<bool> g_prof_on{false};        // forward (caller thread) and backward (autograd thread) both launch GEMMs:
static std::atomic<int> g_prof_n{0};              // slots of the recorder are claimed atomically
static hipEvent_t g_prof_ev[2 * PROF_MAX] = {};
static double g_prof_flops[PROF_MAX];
static int g_prof_kind[PROF_MAX];                 // 0: batched igemm (Winograd plane GEMMs), 1: fused GEMM + output transform kernel

// Internal: bracket one launch of another file's kernel with the recorder's events (kind as above).  begin returns the slot or -1.
__attribute__((visibility("hidden"))) int ssd_internal_prof_open(double flops, int kind, hipStream_t st) {
    const int i = g_prof_on.load(std::memory_order_acquire) ? g_prof_n.fetch_add(1, std::memory_order_relaxed) : PROF_MAX;
    if (i >= PROF_MAX) return -1;
    g_prof_flops[i] = flops;
    g_prof_kind[i] = kind;
    (void)hipEventRecord(g_prof_ev[2 * i], st);
    return i;
}
__attribute__((visibility("hidden"))) void ssd_internal_prof_close(int slot, hipStream_t st) {
    if (slot >= 0) (void)hipEventRecord(g_prof_ev[2 * slot + 1], st);
}

int ssd_internal_gemm_nt_wants(int M, int K, int N, int n_rows, int nbatch);
int ssd_internal_gemm_nt(const float* a, const float* w, float* out, int M, int K, int N, int n_rows, int nbatch, size_t batch_a_elems,
                         size_t batch_w_elems, hipStream_t st);
// Internal (not part of the C ABI): `nbatch` independent GEMMs out[b][M][N] = a[b][M][K] * w[b][N][K]^T on the 64x64 f32 kernel
// (the sixteen planes of a Winograd F(2x2,3x3) convolution).  K % 32 == 0; rows of w beyond n_rows read as zero.
// ksplit > 1: every GEMM is cut into K slices that write raw partial tiles to out[b][slice][M][N] (the caller adds them up).
__attribute__((visibility("hidden"))) int ssd_internal_gemm_batched(const float* a, const float* w, float* out, int M, int K, int N,
                                                                     int n_rows, int nbatch, size_t batch_a_elems, size_t batch_w_elems,
                                                                     int ksplit, hipStream_t st) {
    if (K % 32 != 0 || M <= 0 || N <= 0 || nbatch <= 0 || nbatch > 65535 || ksplit < 1 || ksplit > 65535) return SSD_ERR_BAD_SHAPE;
    const size_t ab = (size_t)M * K * 4, wb = (size_t)n_rows * K * 4;
    if (ab >= 0xF0000000ull || wb >= 0xF0000000ull) return SSD_ERR_BAD_SHAPE;
    IgemmParams p{};
    p.a = a; p.w = w; p.out = out;
    p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb;
    p.Ha = 1; p.Wa = M; p.Ca = K; p.Ho = 1; p.Wo = M;
    p.Nout = N; p.Nrows = n_rows; p.ldo = N; p.R = 1; p.S = 1;
    p.sm = 1; p.sd = 1; p.off = 0; p.dstep = 1;
    p.M = M; p.relu = 0; p.accumulate = 0;
    p.ksplit = 1; p.stamps = nullptr;
    if (ksplit > 1) {
        p.kt_per_split = ssd_cdiv(K / 32, ksplit);
        p.ksplit = ssd_cdiv(K / 32, p.kt_per_split);
        if (p.ksplit != ksplit) return SSD_ERR_BAD_SHAPE;       // the caller sized `out` for exactly ksplit slices
        p.slab = out;
    }
    p.nbatch = nbatch; p.batch_a = batch_a_elems; p.batch_w = batch_w_elems; p.batch_out = (size_t)M * N;
    const bool nt = ksplit == 1 && ssd_internal_gemm_nt_wants(M, K, N, n_rows, nbatch);      // the 128x128 LDS-DMA kernel (gemm_nt.hip)
    const int i = g_prof_on.load(std::memory_order_acquire) ? g_prof_n.fetch_add(1, std::memory_order_relaxed) : PROF_MAX;
    if (i < PROF_MAX) {                                  // measurement aid: this launch alone between two events of the library
        g_prof_flops[i] = 2.0 * M * K * N * nbatch;
        g_prof_kind[i] = nt ? 3 : 0;
        (void)hipEventRecord(g_prof_ev[2 * i], st);
        const int e = nt ? ssd_internal_gemm_nt(a, w, out, M, K, N, n_rows, nbatch, batch_a_elems, batch_w_elems, st)
                         : launch_igemm<64, 64, 2, 2, 1, true>(p, st);
        (void)hipEventRecord(g_prof_ev[2 * i + 1], st);
        return e;
    }
    if (nt) return ssd_internal_gemm_nt(a, w, out, M, K, N, n_rows, nbatch, batch_a_elems, batch_w_elems, st);
    return launch_igemm<64, 64, 2, 2, 1, true>(p, st);
}

extern "C" int ssd_gemm_planes_f32(const float* a, const float* w, float* out, int M, int K, int N, int n_rows, int nbatch, void* stream) {
    if (!a || !w || !out) return SSD_ERR_NULL;
    if (!ssd_aligned16(a) || !ssd_aligned16(w) || !ssd_aligned16(out)) return SSD_ERR_ALIGN;
    return ssd_internal_gemm_batched(a, w, out, M, K, N, n_rows, nbatch, (size_t)M * K, (size_t)n_rows * K, 1, (hipStream_t)stream);
}

// Measurement aid (bench.py): time every batched Winograd GEMM launch by itself, so that the kernel's own rate can be held
// against its rocprof row.  begin() creates the events on first use and arms the recorder; collect() (after the caller has
// synchronised) returns up to `max` (milliseconds, executed FLOPs) pairs in launch order and disarms it.
extern "C" int ssd_prof_gemm_begin(void) {
    for (int i = 0; i < 2 * PROF_MAX; ++i)
        if (g_prof_ev[i] == nullptr && hipEventCreate(&g_prof_ev[i]) != hipSuccess) return SSD_ERR_LAUNCH;
    g_prof_n.store(0);
    g_prof_on.store(true, std::memory_order_release);
    return SSD_OK;
}
extern "C" int ssd_prof_gemm_collect(float* ms_out, double* flops_out, int max) { return ssd_prof_gemm_collect_kinds(ms_out, flops_out, nullptr, max); }
extern "C" int ssd_prof_gemm_collect_kinds(float* ms_out, double* flops_out, int* kinds_out, int max) {
    g_prof_on.store(false);
    if (!ms_out || !flops_out) return SSD_ERR_NULL;
    int n = 0;
    const int recorded = g_prof_n.load() < PROF_MAX ? g_prof_n.load() : PROF_MAX;
    for (; n < recorded && n < max; ++n) {
        if (hipEventElapsedTime(&ms_out[n], g_prof_ev[2 * n], g_prof_ev[2 * n + 1]) != hipSuccess) return -n - 100;
        flops_out[n] = g_prof_flops[n];
        if (kinds_out) kinds_out[n] = g_prof_kind[n];
    }
    g_prof_n.store(0);
    return n;
}

// ---- "f32 from three bf16 limbs" entry points (opt-in; see igemm_x3_kernel) -----------------------------------
extern "C" int ssd_weight_split_bf16x3(const float* w, void* planes, size_t n, void* stream) {
    if (!w || !planes) return SSD_ERR_NULL;
    if (n == 0) return SSD_OK;
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)stream, w,
                       reinterpret_cast<__bf16*>(planes), n);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

extern "C" int ssd_conv2d_fwd_x3(const float* x, const void* w3_ohwi, int w_rows, const float* bias, float* y, int ldy,
                                 const ssd_conv_geom* g, int relu, void* stream) {
    if (int e = check_geom(g)) return e;
    if (!x || !w3_ohwi || !y) return SSD_ERR_NULL;
    if (g->Ci % 32 != 0 || ldy < g->Co || w_rows < g->Co) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(x) || !ssd_aligned16(w3_ohwi)) return SSD_ERR_ALIGN;
    X3Params q{};
    IgemmParams& p = q.g;
    p.a = x; p.w = nullptr; p.bias = bias; p.out = y; p.mask = nullptr;
    const size_t ab = (size_t)g->N * g->H * g->W * g->Ci * 4, pb = (size_t)w_rows * g->R * g->S * g->Ci * 2;
    if (ab >= 0xF0000000ull || 3 * pb >= 0xF0000000ull) return SSD_ERR_BAD_SHAPE;
    p.a_bytes = (unsigned)ab; q.plane_bytes = (unsigned)pb; q.w3 = reinterpret_cast<const __bf16*>(w3_ohwi);
    p.Ha = g->H; p.Wa = g->W; p.Ca = g->Ci; p.Ho = g->Ho; p.Wo = g->Wo;
    p.Nout = g->Co; p.Nrows = g->Co; p.ldo = ldy; p.R = g->R; p.S = g->S;
    p.sm = g->stride; p.sd = 1; p.off = -g->pad; p.dstep = g->dil;
    p.M = g->N * g->Ho * g->Wo; p.relu = relu; p.accumulate = 0;
    { const int h = try_halo<3>(q, g, (hipStream_t)stream); if (h <= 0) return h; }
    return dispatch_igemm_x3(q, (hipStream_t)stream);
}

extern "C" int ssd_conv2d_dgrad_x3(const float* dy, int ldy, const void* w3_ihwo, int Co_pad, float* dx,
                                   const float* relu_mask, int accumulate, const ssd_conv_geom* g, void* stream) {
    if (int e = check_geom(g)) return e;
    if (!dy || !w3_ihwo || !dx) return SSD_ERR_NULL;
    if (Co_pad % 32 != 0 || Co_pad < g->Co || ldy != Co_pad || g->Ci % 4 != 0) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(dy) || !ssd_aligned16(w3_ihwo)) return SSD_ERR_ALIGN;
    X3Params q{};
    IgemmParams& p = q.g;
    p.a = dy; p.w = nullptr; p.bias = nullptr; p.out = dx; p.mask = relu_mask;
    const size_t ab = (size_t)g->N * g->Ho * g->Wo * Co_pad * 4, pb = (size_t)g->Ci * g->R * g->S * Co_pad * 2;
    if (ab >= 0xF0000000ull || 3 * pb >= 0xF0000000ull) return SSD_ERR_BAD_SHAPE;
    p.a_bytes = (unsigned)ab; q.plane_bytes = (unsigned)pb; q.w3 = reinterpret_cast<const __bf16*>(w3_ihwo);
    p.Ha = g->Ho; p.Wa = g->Wo; p.Ca = Co_pad; p.Ho = g->H; p.Wo = g->W;
    p.Nout = g->Ci; p.Nrows = g->Ci; p.ldo = g->Ci; p.R = g->R; p.S = g->S;
    p.sm = 1; p.sd = g->stride; p.off = g->pad; p.dstep = -g->dil;
    p.M = g->N * g->H * g->W; p.relu = 0; p.accumulate = accumulate;
    { const int h = try_halo<3>(q, g, (hipStream_t)stream); if (h <= 0) return h; }
    return dispatch_igemm_x3(q, (hipStream_t)stream);
}

extern "C" int ssd_tune_set_igemm_x3(int tile) {
    if (tile < -1 || tile > 3 || tile == 0) return SSD_ERR_BAD_SHAPE;
    g_x3_tile = tile;
    return SSD_OK;
}
extern "C" int ssd_tune_set_halo(int mode) {
    if (mode < -1 || mode > 2) return SSD_ERR_BAD_SHAPE;
    g_halo = mode;
    return SSD_OK;
}

// bf16-operand halo entry points (configs[2]): same pre-split weight planes as f32x3 (plane 0 = RNE bf16 of the weight);
// geometries the halo kernel does not take return 1 so that the caller uses ssd_conv2d_fwd_bf16 / _dgrad_bf16 instead.
extern "C" int ssd_conv3x3_halo_fwd_bf16(const float* x, const void* w3_ohwi, int w_rows, const float* bias, float* y, int ldy,
                                         const ssd_conv_geom* g, int relu, void* stream) {
    if (int e = check_geom(g)) return e;
    if (!x || !w3_ohwi || !y) return SSD_ERR_NULL;
    if (g->Ci % 32 != 0 || ldy < g->Co || w_rows < g->Co) return SSD_ERR_BAD_SHAPE;
    X3Params q{};
    IgemmParams& p = q.g;
    p.a = x; p.bias = bias; p.out = y; p.mask = nullptr;
    const size_t ab = (size_t)g->N * g->H * g->W * g->Ci * 4, pb = (size_t)w_rows * g->R * g->S * g->Ci * 2;
    if (ab >= 0xF0000000ull || 3 * pb >= 0xF0000000ull) return SSD_ERR_BAD_SHAPE;
    p.a_bytes = (unsigned)ab; q.plane_bytes = (unsigned)pb; q.w3 = reinterpret_cast<const __bf16*>(w3_ohwi);
    p.Ha = g->H; p.Wa = g->W; p.Ca = g->Ci; p.Ho = g->Ho; p.Wo = g->Wo;
    p.Nout = g->Co; p.Nrows = g->Co; p.ldo = ldy; p.R = 3; p.S = 3;
    p.sm = 1; p.sd = 1; p.off = -g->pad; p.dstep = g->dil;
    p.M = g->N * g->Ho * g->Wo; p.relu = relu; p.accumulate = 0;
    return try_halo<1>(q, g, (hipStream_t)stream);
}
extern "C" int ssd_conv3x3_halo_dgrad_bf16(const float* dy, int ldy, const void* w3_ihwo, int Co_pad, float* dx,
                                           const float* relu_mask, int accumulate, const ssd_conv_geom* g, void* stream) {
    if (int e = check_geom(g)) return e;
    if (!dy || !w3_ihwo || !dx) return SSD_ERR_NULL;
    if (Co_pad % 32 != 0 || Co_pad < g->Co || ldy != Co_pad || g->Ci % 4 != 0) return SSD_ERR_BAD_SHAPE;
    X3Params q{};
    IgemmParams& p = q.g;
    p.a = dy; p.bias = nullptr; p.out = dx; p.mask = relu_mask;
    const size_t ab = (size_t)g->N * g->Ho * g->Wo * Co_pad * 4, pb = (size_t)g->Ci * g->R * g->S * Co_pad * 2;
    if (ab >= 0xF0000000ull || 3 * pb >= 0xF0000000ull) return SSD_ERR_BAD_SHAPE;
    p.a_bytes = (unsigned)ab; q.plane_bytes = (unsigned)pb; q.w3 = reinterpret_cast<const __bf16*>(w3_ihwo);
    p.Ha = g->Ho; p.Wa = g->Wo; p.Ca = Co_pad; p.Ho = g->H; p.Wo = g->W;
    p.Nout = g->Ci; p.Nrows = g->Ci; p.ldo = g->Ci; p.R = 3; p.S = 3;
    p.sm = 1; p.sd = g->stride; p.off = g->pad; p.dstep = -g->dil;
    p.M = g->N * g->H * g->W; p.relu = 0; p.accumulate = accumulate;
    return try_halo<1>(q, g, (hipStream_t)stream);
}
