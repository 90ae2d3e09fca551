// 3x3 / stride 1 / pad 1 convolution on bf16 tensors (BASELINE.json configs[2]: "bf16 convs"), forward and data gradient.
//
// Replaces nn.Conv2d(+ReLU) of the VGG trunk and the heads (Model.py:135-143, 176-184) and its autograd data gradient in
// the bf16 mode: activations and gradients live in HBM as NHWC bf16 (written once, by the producing kernel's epilogue),
// weights arrive as bf16 [rows][9][K] copies of the f32 masters, products are accumulated in f32 on
// v_mfma_f32_32x32x16_bf16.
//
// Structure (one workgroup = 512 threads = 8 waves, one workgroup per CU):
//   * M tile = BM output positions, N tile = BN output channels, K walked as (64-channel chunk) x (9 taps).
//   * Per chunk the input HALO of the tile (every input row any tap of any position reads: (PH+2) x (PW+2) pixels of a 2-D
//     patch, or BM + 2*Wp + 2 consecutive positions of the flattened, zero-padded map) goes to LDS ONCE, by LDS-DMA
//     (global_load_lds_dwordx4: bf16 needs no conversion on the way, so no staging registers and no ds_write); the nine taps
//     are LDS row offsets.  The chunk after it is prefetched into the other halo buffer, one piece per wave and stage.
//   * Per (chunk, tap) stage the BN x 64 weight tile streams through a 3-slot ring, two stages ahead, counted vmcnt + raw
//     s_barrier (the DMA of stage s+2 is in flight across the barrier of stage s).
//   * LDS rows are unpadded 128 bytes (64 channels); the 16-byte chunk c of row h sits at c ^ ((kappa(h) >> 1) & 7), kappa =
//     the row's column in the halo (2-D) or its flat index: lanes of one ds_read_b128 group then hit 16 different 16-byte
//     slots of the 256-byte bank row (conflict-free), and the DMA applies the same XOR on its SOURCE address.
//   * The MFMA computes D^T: A operand = weight rows, B operand = pixels, so a lane ends up with 4 CONSECUTIVE output
//     channels of one pixel per accumulator quad: the epilogue (bias, ReLU, += dx, ReLU mask of the data gradient) stores 8
//     bytes (bf16) or 16 bytes (f32, the heads) per lane straight to NHWC memory -- no LDS round trip.
//
// Position spaces ("mode"):
//   0  2-D patches of 8 x 32 (BM = 256) or 16 x 32 (BM = 512) output pixels    (300^2, 150^2 maps)
//   1  2-D patches of 16 x 16                                                    (75^2)
//   2  flat: q runs over [image][H+1][W+2] (one zero row between images, two zero columns per row), a tile is BM
//      consecutive q; positions on a zero row / column compute garbage that is never stored   (38^2, 19^2, 10^2 ...)
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
// (the LDS-DMA builtin takes its source as `const float*` below: with a `const __bf16*` argument the HOST pass silently drops the
// kernel's stub -- the instantiation is marked invalid there -- and the library fails to load with an undefined symbol)

__device__ __attribute__((aligned(128))) unsigned int g_zero_page[32];     // 128 zero bytes: the DMA source of every padding row

struct HaloParams {
    const __bf16* __restrict__ x;       // [N][H][W][ldx] bf16
    const __bf16* __restrict__ w;       // [Nrows][9][K] bf16
    const float* __restrict__ bias;     // [Nout] or null
    void* __restrict__ out;             // [N][H][W][ldo] bf16 or f32
    const __bf16* __restrict__ mask;    // [N][H][W][ldo] bf16 (x of the forward: the data gradient passes where x > 0) or null
    int N, H, W;
    int K, ldx;                         // reduction channels per tap (multiple of 64); row stride of x
    int Nout, Nrows, ldo;
    int relu, accumulate, out_f32, flip;
    int tiles_m, tiles_n;
    int npw, nph;                       // modes 0, 1: patches per map row / column
    int Wp, img_pitch, total_q;         // mode 2: padded row pitch W+2, positions per image (H+1)*Wp, positions in the batch
};

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_vm(int n) {       // n: a constant after unrolling (0..3); folds to one s_waitcnt
    if (n >= 3) wait_vmcnt<3>();
    else if (n == 2) wait_vmcnt<2>();
    else if (n == 1) wait_vmcnt<1>();
    else wait_vmcnt<0>();
}

// TM x TN 32x32 accumulators per wave, WVM x WVN waves; MODE as above; PH = patch rows (modes 0, 1); APW = halo pieces
// (8 rows = 1 KB each) per wave; ADBL: two halo buffers (the next chunk is prefetched while this one is multiplied)
template <int TM, int TN, int WVM, int WVN, int MODE, int PH, int APW, bool ADBL>
__global__ __launch_bounds__(512, 2) void conv3x3_bf16_kernel(const HaloParams p) {
    static_assert(WVM * WVN == 8, "8 waves");
    constexpr int BM = WVM * TM * 32, BN = WVN * TN * 32;
    constexpr int PW = MODE == 1 ? 16 : 32;
    constexpr int HW = PW + 2, HH = PH + 2;
    constexpr int A_ROWS = APW * 64;                       // halo rows one buffer holds
    constexpr int A_BYTES = A_ROWS * 128;
    constexpr int B_PIECES = BN / 8, BPW = B_PIECES / 8;   // weight-tile pieces per stage / per wave
    constexpr int B_BYTES = BN * 128;
    constexpr int NSLOT = 3;
    static_assert(B_PIECES % 8 == 0 && BPW >= 1, "every wave issues the same number of weight pieces");
    static_assert(MODE == 2 || (PH * PW == BM && HH * HW <= A_ROWS), "patch / halo size");
    static_assert((ADBL ? 2 : 1) * A_BYTES + NSLOT * B_BYTES <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(128))) unsigned char lds[(ADBL ? 2 : 1) * A_BYTES + NSLOT * B_BYTES];    // the only LDS object
    unsigned char* const As = lds;
    unsigned char* const Bs = lds + (ADBL ? 2 : 1) * A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WVN, wn = wave % WVN;
    const int lr = lane & 31, lh = lane >> 5;
    const int nblk = p.tiles_m * p.tiles_n;
    const int lid = xcd_swizzle(blockIdx.x, nblk);
    const int tile_m = lid / p.tiles_n, tile_n = lid - tile_m * p.tiles_n;          // n fastest: neighbours share the halo in one L2
    const int n0 = tile_n * BN;

    // ---- tile origin ---------------------------------------------------------------------------------------------------
    int img = 0, oy0 = 0, ox0 = 0, q0 = 0;
    if constexpr (MODE == 2) {
        q0 = tile_m * BM;
    } else {
        const int per_img = p.npw * p.nph;
        img = tile_m / per_img;
        const int prem = tile_m - img * per_img;
        oy0 = (prem / p.npw) * PH;
        ox0 = (prem % p.npw) * PW;
    }

    // ---- LDS-DMA sources of the halo: piece q = wave + 8 i covers halo rows 8q .. 8q+7, lane -> (row 8q + lane/8, slot lane%8) ----
    const __bf16* a_src[APW];
    int a_step[APW];                       // 64 for a real pixel (advance by one chunk per chunk), 0 for a zero row
    const __bf16* const zero = reinterpret_cast<const __bf16*>(g_zero_page);
#pragma unroll
    for (int i = 0; i < APW; ++i) {
        const int h = (wave + 8 * i) * 8 + (lane >> 3), pc = lane & 7;
        int n = 0, y = 0, x = 0, kappa = h;
        bool ok;
        if constexpr (MODE == 2) {
            const int fq = q0 - p.Wp - 1 + h;                 // flat position of halo row h
            ok = fq >= 0 && fq < p.total_q && h < BM + 2 * p.Wp + 2;
            const int fqq = ok ? fq : 0;
            n = fqq / p.img_pitch;
            const int rem = fqq - n * p.img_pitch;
            y = rem / p.Wp;
            x = rem - y * p.Wp;
            ok = ok && y < p.H && x < p.W;
        } else {
            const int hy = h / HW, hx = h - hy * HW;
            kappa = hx;
            n = img;
            y = oy0 - 1 + hy;
            x = ox0 - 1 + hx;
            ok = h < HH * HW && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        }
        const int c = pc ^ ((kappa >> 1) & 7);
        a_src[i] = ok ? p.x + ((size_t)(n * p.H + y) * p.W + x) * p.ldx + c * 8 : zero + c * 8;
        a_step[i] = ok ? 64 : 0;
    }
    // ---- weight tile: piece q = wave * BPW + i covers rows 8q .. 8q+7 of the N tile ---------------------------------------------
    const __bf16* b_src[BPW];
    int b_mul[BPW];
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
        const int row = (wave * BPW + i) * 8 + (lane >> 3), pc = lane & 7;
        const int c = pc ^ ((row >> 1) & 7);
        const bool ok = n0 + row < p.Nrows;
        b_src[i] = ok ? p.w + (size_t)(n0 + row) * 9 * p.K + c * 8 : zero + c * 8;
        b_mul[i] = ok ? 1 : 0;
    }
    // A DMA that has nothing to fetch (no next chunk, no stage s+2) still runs, from the zero page into the slot it would have
    // filled: every wave then issues the same number of DMAs in every stage and all wait counts are compile-time constants.
    auto issue_a = [&](int i, int kc, int buf, bool real) {     // piece i of this wave, chunk kc -> halo buffer buf
        const __bf16* src = real ? a_src[i] + kc * a_step[i] : zero;
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src), (lds_void*)(As + buf * A_BYTES + (wave + 8 * i) * 1024), 16, 0, 0);
    };
    auto issue_b = [&](int i, int s, int kc, int t, bool real) {   // piece i of this wave, stage s = kc * 9 + t -> ring slot s % 3
        const __bf16* src = real ? b_src[i] + (t * p.K + kc * 64) * b_mul[i] : zero;
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src), (lds_void*)(Bs + (s % NSLOT) * B_BYTES + (wave * BPW + i) * 1024), 16, 0, 0);
    };

    // ---- fragment addresses ------------------------------------------------------------------------------------------------
    // pixel side: tile row m = (wm*TM + i)*32 + lr -> halo row of tap (0,0) and its swizzle key
    int hbase[TM], kbase[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = (wm * TM + i) * 32 + lr;
        if constexpr (MODE == 2) {
            hbase[i] = m;
            kbase[i] = m;
        } else {
            const int py = m / PW, px = m - py * PW;
            hbase[i] = py * HW + px;
            kbase[i] = px;
        }
    }
    // weight side: row (wn*TN + j)*32 + lr; the 16-byte slot of k chunk 2ks + lh is constant over the stages
    int w_off[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = (wn * TN + j) * 32 + lr;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) w_off[j][ks] = row * 128 + (((2 * ks + lh) ^ ((row >> 1) & 7)) << 4);
    }

    f32x16 acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

    const int KC = p.K >> 6, NS = KC * 9;
    const int row_pitch = MODE == 2 ? p.Wp : HW;

    // ---- prologue: halo of chunk 0, weight tiles of stages 0 and 1 ---------------------------------------------------------
#pragma unroll
    for (int i = 0; i < APW; ++i) issue_a(i, 0, 0, true);
#pragma unroll
    for (int i = 0; i < BPW; ++i) issue_b(i, 0, 0, 0, true);
#pragma unroll
    for (int i = 0; i < BPW; ++i) issue_b(i, 1, 0, 1, true);  // NS >= 9
    wait_vmcnt<BPW>();                                       // everything but stage 1's weights has landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // tap (r, s2) reads halo row (py + dr) * pitch + (px + ds): forward dr = r, data gradient dr = 2 - r
    int xrow[TM], xsw[TM];
    auto tap_setup = [&](int t) {
        const int r = t / 3, s2 = t - 3 * r;
        const int dr = p.flip ? 2 - r : r, ds = p.flip ? 2 - s2 : s2;
        const int hoff = dr * row_pitch + ds;
        const int koff = MODE == 2 ? hoff : ds;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            xrow[i] = (hbase[i] + hoff) * 128;
            xsw[i] = ((kbase[i] + koff) >> 1) & 7;
        }
    };
    bf16x8 xf[2][TM], wf[2][TN];
    auto load_x = [&](int set, const unsigned char* Ab, int ks) {
#pragma unroll
        for (int i = 0; i < TM; ++i) xf[set][i] = *reinterpret_cast<const bf16x8*>(Ab + xrow[i] + (((2 * ks + lh) ^ xsw[i]) << 4));
    };
    auto load_w = [&](int set, const unsigned char* Bb, int ks) {
#pragma unroll
        for (int j = 0; j < TN; ++j) wf[set][j] = *reinterpret_cast<const bf16x8*>(Bb + w_off[j][ks]);
    };
    tap_setup(0);
    load_x(0, As, 0);

    // Stage = (chunk, tap): 4 k-steps of TM x TN MFMAs.  The fragments of k-step ks+1 are requested before the MFMAs of k-step ks, the
    // stage's DMA instructions sit BETWEEN the k-steps (an LDS-DMA holds the wave's issue port for ~100 cycles: issued in a block at the
    // top of the stage, every wave of the CU would stall together with the matrix pipe idle), and the pixel fragments of the NEXT
    // stage's first k-step are requested before the barrier (the halo buffer is complete for the whole chunk; only the weight slot needs
    // the barrier).
    for (int kc = 0; kc < KC; ++kc) {
        const unsigned char* Ab = As + (ADBL ? (kc & 1) : 0) * A_BYTES;
        const unsigned char* An = As + (ADBL ? ((kc + 1) & 1) : 0) * A_BYTES;
        const bool next_chunk = kc + 1 < KC;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int s = kc * 9 + t;
            const unsigned char* Bb = Bs + (s % NSLOT) * B_BYTES;
            const int t2 = t + 2 >= 9 ? t + 2 - 9 : t + 2, kc2 = t + 2 >= 9 ? kc + 1 : kc;
            load_w(0, Bb, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int cur = ks & 1, nxt = cur ^ 1;
                if (ks < 3) {
                    load_x(nxt, Ab, ks + 1);
                    load_w(nxt, Bb, ks + 1);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (ks == 0) {
                    if (ADBL && t < APW) issue_a(t, kc + 1, (kc + 1) & 1, next_chunk);      // one halo piece of the next chunk per stage
                } else if (ks - 1 < BPW) {
                    issue_b(ks - 1, s + 2, kc2, t2, s + 2 < NS);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cur][j], xf[cur][i], acc[j][i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!ADBL && t == 8) {
                // single halo buffer: every wave has read its last fragment of this chunk before the next one may land
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int i = 0; i < APW; ++i) issue_a(i, kc + 1, 0, next_chunk);
                wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                tap_setup(0);
                load_x(0, An, 0);
            } else {
                tap_setup(t == 8 ? 0 : t + 1);
                load_x(0, t == 8 ? An : Ab, 0);               // (after the last stage: a read nobody uses)
                // all but what this stage issued has landed: stage s+1's weights, the halo pieces of earlier stages
                wait_vm(((ADBL && t < APW) ? 1 : 0) + BPW);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        }
    }

    // ---- epilogue: acc[j][i][reg] = out[pixel m(i, lr)][channel n0 + (wn*TN+j)*32 + 8*(reg>>2) + 4*lh + (reg&3)] ------------------
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = (wm * TM + i) * 32 + lr;
        bool ok;
        size_t pix;
        if constexpr (MODE == 2) {
            const int fq = q0 + m;
            const int fqq = fq < p.total_q ? fq : 0;
            const int n = fqq / p.img_pitch, rem = fqq - n * p.img_pitch;
            const int y = rem / p.Wp, x = rem - y * p.Wp;
            ok = fq < p.total_q && y < p.H && x < p.W;
            pix = (size_t)(n * p.H + y) * p.W + x;
        } else {
            const int py = m / PW, px = m - py * PW;
            const int y = oy0 + py, x = ox0 + px;
            ok = y < p.H && x < p.W;
            pix = (size_t)(img * p.H + y) * p.W + x;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + (wn * TN + j) * 32 + 8 * g + 4 * lh;
                if (ok && n < p.Nout) {
                    f32x4 v = {acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]};
                    if (p.bias != nullptr) {                       // the bias has Nrows entries (a head: 150 of the 152 stored columns)
                        if (n + 3 < p.Nrows) {
                            v += *reinterpret_cast<const f32x4*>(p.bias + n);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (n + e < p.Nrows) v[e] += p.bias[n + e];
                        }
                    }
                    const size_t o = pix * p.ldo + n;
                    if (p.accumulate) {
                        if (p.out_f32) {
                            v += *reinterpret_cast<const f32x4*>(static_cast<const float*>(p.out) + o);
                        } else {
                            const bf16x4 pv = *reinterpret_cast<const bf16x4*>(static_cast<const __bf16*>(p.out) + o);
                            v += f32x4{(float)pv[0], (float)pv[1], (float)pv[2], (float)pv[3]};
                        }
                    }
                    if (p.relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];          // NaN stays NaN, like torch.relu
                    }
                    if (p.mask != nullptr) {
                        const bf16x4 mk = *reinterpret_cast<const bf16x4*>(p.mask + o);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (float)mk[e] > 0.f ? v[e] : 0.f;
                    }
                    if (p.out_f32) {
                        *reinterpret_cast<f32x4*>(static_cast<float*>(p.out) + o) = v;
                    } else {
                        *reinterpret_cast<bf16x4*>(static_cast<__bf16*>(p.out) + o) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                    }
                }
            }
        }
    }
}

int g_force_mode = -1;      // tuning aid: position space (0 / 1 / 2), -1 = by map size
int g_force_bn = -1;        // tuning aid: 64 / 128, -1 = by channel count

template <int TM, int TN, int WVM, int WVN, int MODE, int PH, int APW, bool ADBL>
int launch(HaloParams& p, hipStream_t st) {
    constexpr int BM = WVM * TM * 32, BN = WVN * TN * 32;
    constexpr int PW = MODE == 1 ? 16 : 32;
    if (MODE == 2) {
        p.Wp = p.W + 2;
        p.img_pitch = (p.H + 1) * p.Wp;
        const long total = (long)p.N * p.img_pitch;
        if (total >= (1L << 30) || BM + 2 * p.Wp + 2 > APW * 64) return SSD_ERR_BAD_SHAPE;
        p.total_q = (int)total;
        p.tiles_m = ssd_cdiv(p.total_q, BM);
    } else {
        p.npw = ssd_cdiv(p.W, PW);
        p.nph = ssd_cdiv(p.H, PH);
        p.tiles_m = p.N * p.npw * p.nph;
    }
    p.tiles_n = ssd_cdiv(p.Nout, BN);
    hipLaunchKernelGGL((conv3x3_bf16_kernel<TM, TN, WVM, WVN, MODE, PH, APW, ADBL>), dim3(p.tiles_m * p.tiles_n), dim3(512), 0, st, p);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

int dispatch(HaloParams& p, hipStream_t st) {
    int mode = g_force_mode;
    if (mode < 0) mode = p.W + 2 <= 63 ? 2 : ((p.W >= 128 && p.H >= 128) ? 0 : 1);
    if (mode == 2 && p.W + 2 > 63) mode = 1;
    int bn = g_force_bn;
    if (bn < 0) bn = p.Nout <= 64 ? 64 : 128;
    if (bn == 64) {
        // 64 output channels: wave tile 64 x 64 over 512 positions (16 x 32 patches; one halo buffer, K has one or two chunks here)
        if (mode == 0) return launch<2, 2, 8, 1, 0, 16, 10, false>(p, st);
        if (mode == 1) return launch<2, 1, 4, 2, 1, 16, 6, true>(p, st);
        return launch<2, 1, 4, 2, 2, 0, 6, true>(p, st);
    }
    if (mode == 0) return launch<2, 2, 4, 2, 0, 8, 6, true>(p, st);
    if (mode == 1) return launch<2, 2, 4, 2, 1, 16, 6, true>(p, st);
    return launch<2, 2, 4, 2, 2, 0, 6, true>(p, st);
}

}  // namespace

// Tuning aid (tools/conv_bf16_bench.py): force the position space / N tile; -1 = automatic.
extern "C" int ssd_tune_set_conv_bf16(int mode, int bn) {
    if (mode < -1 || mode > 2 || !(bn == -1 || bn == 64 || bn == 128)) return SSD_ERR_BAD_SHAPE;
    g_force_mode = mode;
    g_force_bn = bn;
    return SSD_OK;
}

// see include/ssd_gfx950.h
extern "C" int ssd_conv3x3_bf16(const void* x, int ldx, const void* w, int w_rows, int K, const float* bias, void* out, int ldo, int n_out,
                                int out_f32, const void* relu_mask, int accumulate, int relu, int flip, int N, int H, int W, void* stream) {
    if (!x || !w || !out) return SSD_ERR_NULL;
    if (N <= 0 || H <= 0 || W <= 0 || K <= 0 || K % 64 != 0 || ldx < K || ldx % 8 != 0 || n_out <= 0 || n_out % 4 != 0 || ldo < n_out ||
        ldo % 4 != 0 || w_rows <= 0)
        return SSD_ERR_BAD_SHAPE;
    if ((long)N * (H + 1) * (W + 2) >= (1L << 30)) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(x) || !ssd_aligned16(w) || !ssd_aligned16(out) || (bias && !ssd_aligned16(bias)) || (relu_mask && ((uintptr_t)relu_mask & 7)))
        return SSD_ERR_ALIGN;
    HaloParams p = {};
    p.x = static_cast<const __bf16*>(x); p.w = static_cast<const __bf16*>(w); p.bias = bias; p.out = out;
    p.mask = static_cast<const __bf16*>(relu_mask);
    p.N = N; p.H = H; p.W = W; p.K = K; p.ldx = ldx; p.Nout = n_out; p.Nrows = w_rows; p.ldo = ldo;
    p.relu = relu; p.accumulate = accumulate; p.out_f32 = out_f32; p.flip = flip;
    return dispatch(p, (hipStream_t)stream);
}
