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
//   1  2-D patches of 16 x 16                                                    (maps 95 ... 127 wide, and 75^2 with 64 output channels)
//   2  flat: q runs over [image][H+1][W+1] (one zero row between images, one zero column between rows), a tile is BM
//      consecutive q; positions on a zero row / column compute garbage that is never stored   (75^2, 38^2, 19^2, 10^2 ...)
#include <type_traits>
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
    int Wp, img_pitch, total_q;         // mode 2: padded row pitch W+1, positions per image (H+1)*Wp, positions in the batch
};

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// Fragment reads as inline asm + hand-counted lgkmcnt waits: for its own ds_reads the compiler (ROCm 7.2) waits lgkmcnt(0) -- everything,
// the reads it has just issued for later steps included -- every few steps; an asm read is invisible to that pass, so the count is ours:
// LDS operations retire in issue order, `lgkmcnt(N)` with N = reads issued AFTER the ones a step needs leaves exactly those in flight.
__device__ __forceinline__ bf16x8 lds_read16(unsigned addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm_n(int n) {   // n: a constant after unrolling (0..15); folds to one s_waitcnt
    switch (n) {
        case 0: wait_lgkm<0>(); break;   case 1: wait_lgkm<1>(); break;   case 2: wait_lgkm<2>(); break;   case 3: wait_lgkm<3>(); break;
        case 4: wait_lgkm<4>(); break;   case 5: wait_lgkm<5>(); break;   case 6: wait_lgkm<6>(); break;   case 7: wait_lgkm<7>(); break;
        case 8: wait_lgkm<8>(); break;   case 9: wait_lgkm<9>(); break;   case 10: wait_lgkm<10>(); break; case 11: wait_lgkm<11>(); break;
        case 12: wait_lgkm<12>(); break; case 13: wait_lgkm<13>(); break; case 14: wait_lgkm<14>(); break; default: wait_lgkm<15>(); break;
    }
}
__device__ __forceinline__ void wait_vm(int n) {       // n: a constant after unrolling (0..3); folds to one s_waitcnt
    if (n >= 3) wait_vmcnt<3>();
    else if (n == 2) wait_vmcnt<2>();
    else if (n == 1) wait_vmcnt<1>();
    else wait_vmcnt<0>();
}

// Four consecutive output channels n .. n+3 of one pixel (a quad of an accumulator of D^T):
// out = [accumulate: out +] v (+ bias) -> [relu] -> [mask > 0], one 8-byte (bf16) or 16-byte (f32) store straight to NHWC memory.
__device__ __forceinline__ void store_quad(const HaloParams& p, f32x4 v, bool ok, size_t pix, int n) {
    if (ok && n < p.Nout) {
        if (p.bias != nullptr) {                           // the bias has Nrows entries (a head: 150 of the 152 stored columns)
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
            for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];              // NaN stays NaN, like torch.relu
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

// MFMA tile geometry of a wave's result.  M16 = v_mfma_f32_16x16x32_bf16 (on this chip the same FLOPs per cycle as 32x32x16 and a higher
// sustained clock under load: MI355X_MICROARCH.md, DVFS give-back (7)), else v_mfma_f32_32x32x16_bf16.  With D^T = W x X^T a lane holds
// pixel `lrow` of a tile and quads of four consecutive channels: M32 regs 4g..4g+3 = channels 8g + 4*lk + e (lk = lane / 32), M16 regs
// 0..3 = channels 4*lk + e (lk = lane / 16).
template <bool M16> struct Mfma {
    static constexpr int MR = M16 ? 16 : 32;               // rows (pixels) = columns (channels) of a tile
    static constexpr int NQ = M16 ? 1 : 4;                 // quads per lane and tile
    static constexpr int KS = M16 ? 2 : 4;                 // k-steps per 64-channel stage
    static constexpr int CPK = M16 ? 4 : 2;                // 16-byte chunks of a 128-byte row one k-step reads (= lane groups along k)
    typedef typename std::conditional<M16, f32x4, f32x16>::type acc_t;
    static __device__ __forceinline__ acc_t mma(const bf16x8 a, const bf16x8 b, const acc_t c) {
        if constexpr (M16) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 quad(const acc_t& c, int g) {
        if constexpr (M16) return c;
        else return f32x4{c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]};
    }
    static __device__ __forceinline__ int quad_ch(int g, int lk) { return M16 ? 4 * lk : 8 * g + 4 * lk; }
};

// Coalesced form of the epilogue for bf16 outputs without accumulation: the wave parks its (TM*32 pixels) x (TN*32 channels) result --
// bias and ReLU applied, rounded to bf16 -- in a private LDS slot as [pixel][channel] rows (row stride +16 bytes: the 8-byte writes of
// 16 pixels then fall on 16 different bank pairs), reads it back 16 bytes per lane and stores whole channel runs: TN*64 contiguous bytes
// per pixel and instruction instead of the 8-byte pieces of store_acc (measured on conv1_2: the scattered stores cost more than the
// MFMAs).  The ReLU mask of a data gradient is applied on the way out (a select commutes with the rounding), read with the same 16-byte
// pattern.  The same wave writes and reads its slot (LDS operations retire in order): no barrier inside.  pix_of(row, ok) -> pixel index.
template <bool M16, int MT, int NT, typename PixOf>
__device__ __forceinline__ void store_tile_staged(const HaloParams& p, const typename Mfma<M16>::acc_t (&acc)[NT][MT], unsigned char* slot, int n_wave,
                                                  int lane, PixOf pix_of) {
    typedef Mfma<M16> F;
    constexpr int ROWS = MT * F::MR, RB = NT * F::MR * 2, RS = RB + 16, LPR = RB / 16, RPI = 64 / LPR;   // rows, row bytes, row stride, lanes per row, rows per instruction
    const int lrow = lane & (F::MR - 1), lk = lane / F::MR;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int g = 0; g < F::NQ; ++g) {
                const int c = j * F::MR + F::quad_ch(g, lk), n = n_wave + c;
                f32x4 v = F::quad(acc[j][i], g);
                if (p.bias != nullptr) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < p.Nrows) v[e] += p.bias[n + e];
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];
                }
                *reinterpret_cast<bf16x4*>(slot + (i * F::MR + lrow) * RS + c * 2) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
            }
    __bf16* const out = static_cast<__bf16*>(p.out);
#pragma unroll
    for (int pass = 0; pass < ROWS / RPI; ++pass) {
        const int row = pass * RPI + lane / LPR, ch = (lane % LPR) * 8, n = n_wave + ch;
        bool ok;
        const size_t pix = pix_of(row, ok);
        bf16x8 v = *reinterpret_cast<const bf16x8*>(slot + row * RS + ch * 2);
        if (ok && n < p.Nout) {
            const size_t o = pix * p.ldo + n;
            if (n + 8 <= p.Nout) {
                if (p.mask != nullptr) {
                    const bf16x8 mk = *reinterpret_cast<const bf16x8*>(p.mask + o);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (float)mk[e] > 0.f ? v[e] : (__bf16)0.f;
                }
                *reinterpret_cast<bf16x8*>(out + o) = v;
            } else {                                          // the last, partial group of channels (Nout % 8 == 4)
                if (p.mask != nullptr) {
                    const bf16x4 mk = *reinterpret_cast<const bf16x4*>(p.mask + o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (float)mk[e] > 0.f ? v[e] : (__bf16)0.f;
                }
                *reinterpret_cast<bf16x4*>(out + o) = bf16x4{v[0], v[1], v[2], v[3]};
            }
        }
    }
}

// TM x TN 32x32 accumulators per wave, WVM x WVN waves; MODE as above; PH = patch rows (modes 0, 1); APW = halo pieces
// (8 rows = 1 KB each) per wave; ADBL: two halo buffers (the next chunk is prefetched while this one is multiplied)
template <int TM, int TN, int WVM, int WVN, int MODE, int PH, int APW, bool ADBL, bool M16>
__global__ __launch_bounds__(512, 2) void conv3x3_bf16_kernel(const HaloParams p) {
    typedef Mfma<M16> F;
    constexpr int MR = F::MR, MT = TM * 32 / MR, NT = TN * 32 / MR, KS = F::KS;
    static_assert(WVM * WVN == 8, "8 waves");
    constexpr int BM = WVM * TM * 32, BN = WVN * TN * 32;
    constexpr int PW = MODE == 1 ? 16 : 32;
    constexpr int HW = PW + 2, HH = PH + 2;
    constexpr int A_ROWS = APW * 64;                       // halo rows one buffer holds
    constexpr int A_BYTES = A_ROWS * 128;
    constexpr int B_PIECES = BN / 8, BPW = B_PIECES / 8;   // weight-tile pieces per stage / per wave
    constexpr int B_BYTES = BN * 128;
    // Weight ring.  3 slots: stage s + 2 is requested during stage s and is published by the barrier that ENDS stage s + 1, so the first
    // weight fragments of a stage can only be read AFTER that barrier.  Round-4 experiment (-DCB_RING4, tools/cb_variants.sh): 4 slots, stage
    // s + 3 requested during stage s, stage s + 1's weights published one barrier earlier and its first fragments requested BEFORE the barrier
    // that ends stage s, like the pixel fragments -- measured no faster (forward layers 2.15-2.21 ms against 2.06-2.15, data gradients 1.96
    // against 1.92-1.94, step 13.09 against 13.04 ms on a 1 815 MHz device): that latency is not what holds the kernel; default stays 3.
#ifdef CB_RING4
    constexpr int NSLOT = 4, AHEAD = 3;
#else
    constexpr int NSLOT = 3, AHEAD = 2;
#endif
    static_assert(B_PIECES % 8 == 0 && BPW >= 1, "every wave issues the same number of weight pieces");
    static_assert(MODE == 2 || (PH * PW == BM && HH * HW <= A_ROWS), "patch / halo size");
    static_assert((ADBL ? 2 : 1) * A_BYTES + NSLOT * B_BYTES <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(128))) unsigned char lds[(ADBL ? 2 : 1) * A_BYTES + NSLOT * B_BYTES];    // the only LDS object
    unsigned char* const As = lds;
    unsigned char* const Bs = lds + (ADBL ? 2 : 1) * A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WVN, wn = wave % WVN;
    const int lrow = lane & (MR - 1), lk = lane / MR;       // this lane's row of a fragment, its group along k
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
    // (element offsets from p.x / p.w in 32 bits -- the host checks the tensors are below 2^31 elements -- NONE for a zero row: one
    // register per piece instead of a pointer and a step)
    constexpr unsigned NONE = 0xFFFFFFFFu;
    unsigned a_off[APW];
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
        a_off[i] = ok ? (unsigned)((n * p.H + y) * p.W + x) * (unsigned)p.ldx + c * 8 : NONE;
    }
    // ---- weight tile: piece q = wave * BPW + i covers rows 8q .. 8q+7 of the N tile ---------------------------------------------
    unsigned b_off[BPW];
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
        const int row = (wave * BPW + i) * 8 + (lane >> 3), pc = lane & 7;
        const int c = pc ^ ((row >> 1) & 7);
        b_off[i] = n0 + row < p.Nrows ? (unsigned)(n0 + row) * 9u * (unsigned)p.K + c * 8 : NONE;
    }
    // A DMA that has nothing to fetch (no next chunk, no stage s+2) still runs, from the zero page into the slot it would have
    // filled: every wave then issues the same number of DMAs in every stage and all wait counts are compile-time constants.
    auto issue_a = [&](int i, int kc, int buf, bool real) {     // piece i of this wave, chunk kc -> halo buffer buf
        const __bf16* src = (real && a_off[i] != NONE) ? p.x + (size_t)(a_off[i] + (unsigned)kc * 64u) : zero;
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src), (lds_void*)(As + buf * A_BYTES + (wave + 8 * i) * 1024), 16, 0, 0);
    };
    auto issue_b = [&](int i, int s, int kc, int t, bool real) {   // piece i of this wave, stage s = kc * 9 + t -> ring slot s % 3
        const __bf16* src = (real && b_off[i] != NONE) ? p.w + (size_t)(b_off[i] + (unsigned)(t * p.K + kc * 64)) : zero;
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src), (lds_void*)(Bs + (s % NSLOT) * B_BYTES + (wave * BPW + i) * 1024), 16, 0, 0);
    };

    // ---- fragment addresses ------------------------------------------------------------------------------------------------
    // pixel side: tile row m = wm*TM*32 + i*MR + lrow -> halo row of tap (0,0) and its swizzle key
    int hbase[MT], kbase[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = wm * TM * 32 + i * MR + lrow;
        if constexpr (MODE == 2) {
            hbase[i] = m;
            kbase[i] = m;
        } else {
            const int py = m / PW, px = m - py * PW;
            hbase[i] = py * HW + px;
            kbase[i] = px;
        }
    }
    // weight side: row wn*TN*32 + j*MR + lrow; the 16-byte slot of k chunk CPK*ks + lk is constant over the stages
    int w_off[NT][KS];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int row = wn * TN * 32 + j * MR + lrow;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) w_off[j][ks] = row * 128 + (((F::CPK * ks + lk) ^ ((row >> 1) & 7)) << 4);
    }

    typename F::acc_t acc[NT][MT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4 * F::NQ; ++r) acc[j][i][r] = 0.f;

    const int KC = p.K >> 6, NS = KC * 9;
    const int row_pitch = MODE == 2 ? p.Wp : HW;

    // ---- prologue: halo of chunk 0, weight tiles of stages 0 .. AHEAD - 1 ---------------------------------------------------------
#pragma unroll
    for (int i = 0; i < APW; ++i) issue_a(i, 0, 0, true);
#pragma unroll
    for (int i = 0; i < BPW; ++i) issue_b(i, 0, 0, 0, true);
#pragma unroll
    for (int i = 0; i < BPW; ++i) issue_b(i, 1, 0, 1, true);  // NS >= 9
    if (AHEAD == 3) {
#pragma unroll
        for (int i = 0; i < BPW; ++i) issue_b(i, 2, 0, 2, true);
    }
    wait_vmcnt<BPW>();                                       // everything but the last requested stage's weights has landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // tap (r, s2) reads halo row (py + dr) * pitch + (px + ds): forward dr = r, data gradient dr = 2 - r
    int xrow[MT], xsw[MT];
    auto tap_setup = [&](int t) {
        const int r = t / 3, s2 = t - 3 * r;
        const int dr = p.flip ? 2 - r : r, ds = p.flip ? 2 - s2 : s2;
        const int hoff = dr * row_pitch + ds;
        const int koff = MODE == 2 ? hoff : ds;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            xrow[i] = (hbase[i] + hoff) * 128;
            xsw[i] = ((kbase[i] + koff) >> 1) & 7;
        }
    };
    // M32: two fragment sets (the reads of k-step ks+1 are requested before the MFMAs of k-step ks, and the next stage's first pixel
    // fragments before the barrier).  M16 has twice the fragments per k-step (8 x 4 registers): one set -- the partner wave of the SIMD
    // covers the read latency -- or the kernel spills.
    constexpr int NSET = M16 ? 1 : 2;
    bf16x8 xf[NSET][MT], wf[NSET][NT];
    auto load_x = [&](int set, const unsigned char* Ab, int ks) {
#pragma unroll
        for (int i = 0; i < MT; ++i) xf[set][i] = *reinterpret_cast<const bf16x8*>(Ab + xrow[i] + (((F::CPK * ks + lk) ^ xsw[i]) << 4));
    };
    auto load_w = [&](int set, const unsigned char* Bb, int ks) {
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[set][j] = *reinterpret_cast<const bf16x8*>(Bb + w_off[j][ks]);
    };
    tap_setup(0);
    if (NSET == 2) load_x(0, As, 0);
    if (NSET == 2 && AHEAD == 3) load_w(0, Bs, 0);

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
            const int t2 = t + AHEAD >= 9 ? t + AHEAD - 9 : t + AHEAD, kc2 = t + AHEAD >= 9 ? kc + 1 : kc;
            if (NSET == 2 && AHEAD == 2) load_w(0, Bb, 0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int cur = NSET == 2 ? (ks & 1) : 0, nxt = cur ^ 1;
                if (NSET == 2) {
                    if (ks < KS - 1) {
                        load_x(nxt, Ab, ks + 1);
                        load_w(nxt, Bb, ks + 1);
                    }
                } else {
                    load_x(0, Ab, ks);
                    load_w(0, Bb, ks);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (ks == 0) {
                    if (ADBL && t < APW) issue_a(t, kc + 1, (kc + 1) & 1, next_chunk);      // one halo piece of the next chunk per stage
                    if (KS == 1) {
#pragma unroll
                        for (int i = 0; i < BPW; ++i) issue_b(i, s + AHEAD, kc2, t2, s + AHEAD < NS);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < BPW; ++i)
                        if (i % (KS - 1) == ks - 1) issue_b(i, s + AHEAD, kc2, t2, s + AHEAD < NS);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[j][i] = F::mma(wf[cur][j], xf[cur][i], acc[j][i]);
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
                if (NSET == 2) load_x(0, An, 0);
                if (NSET == 2 && AHEAD == 3) load_w(0, Bs + ((s + 1) % NSLOT) * B_BYTES, 0);
            } else {
                tap_setup(t == 8 ? 0 : t + 1);
                if (NSET == 2) load_x(0, t == 8 ? An : Ab, 0);               // (after the last stage: a read nobody uses)
                // 4-slot ring: stage s + 1's weights landed during stage s - 1 and were published by the barrier that ended it
                if (NSET == 2 && AHEAD == 3) load_w(0, Bs + ((s + 1) % NSLOT) * B_BYTES, 0);
                // all but what this stage issued has landed: stage s+1's weights, the halo pieces of earlier stages
                wait_vm(((ADBL && t < APW) ? 1 : 0) + BPW);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        }
    }

    // ---- epilogue: acc[j][i][reg] = out[pixel m(i, lr)][channel n0 + (wn*TN+j)*32 + 8*(reg>>2) + 4*lh + (reg&3)] ------------------
    auto pix_of_m = [&](int m, bool& ok) -> size_t {
        if constexpr (MODE == 2) {
            const int fq = q0 + m;
            const int fqq = fq < p.total_q ? fq : 0;
            const int n = fqq / p.img_pitch, rem = fqq - n * p.img_pitch;
            const int y = rem / p.Wp, x = rem - y * p.Wp;
            ok = fq < p.total_q && y < p.H && x < p.W;
            return (size_t)(n * p.H + y) * p.W + x;
        } else {
            const int py = m / PW, px = m - py * PW;
            const int y = oy0 + py, x = ox0 + px;
            ok = y < p.H && x < p.W;
            return (size_t)(img * p.H + y) * p.W + x;
        }
    };
    constexpr int SLOT = TM * 32 * (TN * 64 + 16);            // bytes of a wave's staging slot
    if (!p.out_f32 && !p.accumulate && (p.ldo & 7) == 0 && 8 * SLOT <= (int)sizeof(lds)) {
        // (the last stage ended with a barrier: every wave is done with the halo and weight buffers, the slots reuse them)
        store_tile_staged<M16, MT, NT>(p, acc, lds + wave * SLOT, n0 + wn * TN * 32, lane,
                                       [&](int row, bool& ok) { return pix_of_m(wm * TM * 32 + row, ok); });
        return;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        bool ok;
        const size_t pix = pix_of_m(wm * TM * 32 + i * MR + lrow, ok);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int g = 0; g < F::NQ; ++g) store_quad(p, F::quad(acc[j][i], g), ok, pix, n0 + wn * TN * 32 + j * MR + F::quad_ch(g, lk));
    }
}

// ---- K = 64, at most 64 output channels (conv1_2 forward and data gradient, Model.py:135 features[2]) -----------------------------
// The general kernel spends a third of such a block outside its nine stages (halo + first weight tiles in, 64 KB of results out, one
// workgroup per CU so nothing covers it).  Here a PERSISTENT workgroup keeps all nine taps of the 64 x 64 filter in LDS (72 KB, loaded
// once) and walks its share of the 8 x 32 patches with two halo buffers: the next patch's halo arrives by LDS-DMA while this one is
// multiplied, nothing inside a patch needs a barrier (halo and filter are complete), one barrier per patch swaps the buffers, and the
// stores of a patch drain under the next patch's MFMAs.  The layer is then bound by its HBM bytes (in + out = 0.74 GB at batch 32).
template <bool M16>
__global__ __launch_bounds__(512, 2) void conv3x3_bf16_k64_kernel(const HaloParams p) {
    typedef Mfma<M16> F;
    constexpr int TM = 2, TN = 1, WVN = 2, PH = 8, PW = 32, HW = PW + 2, HH = PH + 2;
    constexpr int MR = F::MR, MT = TM * 32 / MR, NT = TN * 32 / MR, KS = F::KS, NSTEP = 9 * KS;
    constexpr int NPIECE = (HH * HW + 7) / 8, A_BYTES = NPIECE * 1024, W_BYTES = 9 * 64 * 128, APW = (NPIECE + 7) / 8;
    static_assert(2 * A_BYTES + W_BYTES <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(128))) unsigned char lds[2 * A_BYTES + W_BYTES];
    unsigned char* const Ws = lds + 2 * A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WVN, wn = wave % WVN;
    const int lrow = lane & (MR - 1), lk = lane / MR;
    const __bf16* const zero = reinterpret_cast<const __bf16*>(g_zero_page);
    const int per_img = p.npw * p.nph, ntiles = p.tiles_m;

    // halo piece i of this wave (q = wave + 8 i < NPIECE) of the patch at (img, oy0, ox0) -> buffer buf.  Everything that does not depend
    // on the patch is computed once: the lane's halo pixel (hy, hx) and its offset from the patch origin; per patch only the bounds
    // test and one add remain (the address arithmetic of six pieces per patch was a quarter of the patch's issue slots otherwise).
    int a_hy[APW], a_hx[APW], a_rel[APW];
#pragma unroll
    for (int i = 0; i < APW; ++i) {
        const int h = (wave + 8 * i) * 8 + (lane >> 3), pc = lane & 7;
        const int hy = h / HW, hx = h - hy * HW;
        a_hy[i] = h < HH * HW ? hy - 1 : -(1 << 20);           // rows past the halo: never inside the map
        a_hx[i] = hx - 1;
        a_rel[i] = ((hy - 1) * p.W + (hx - 1)) * p.ldx + (pc ^ ((hx >> 1) & 7)) * 8;
    }
    auto issue_halo = [&](int i, int img, int oy0, int ox0, int buf) {
        const int q = wave + 8 * i;
        if (q >= NPIECE) return;                               // wave-uniform
        const int y = oy0 + a_hy[i], x = ox0 + a_hx[i];
        const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        const __bf16* src = ok ? p.x + ((size_t)(img * p.H + oy0) * p.W + ox0) * p.ldx + a_rel[i] : zero;
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src), (lds_void*)(lds + buf * A_BYTES + q * 1024), 16, 0, 0);
    };
    auto origin = [&](int tile, int& img, int& oy0, int& ox0) {
        img = tile / per_img;
        const int prem = tile - img * per_img, ty = prem / p.npw;
        oy0 = ty * PH;
        ox0 = (prem - ty * p.npw) * PW;
    };
    // the filter: 72 pieces (tap t, rows 8 r .. 8 r + 7), nine per wave
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int q = wave + 8 * i, t = q >> 3, row = (q & 7) * 8 + (lane >> 3), pc = lane & 7;
        const int c = pc ^ ((row >> 1) & 7);
        const __bf16* src = row < p.Nrows ? p.w + ((size_t)row * 9 + t) * p.K + c * 8 : zero;
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src), (lds_void*)(Ws + q * 1024), 16, 0, 0);
    }
    int tile = blockIdx.x;
    int img = 0, oy0 = 0, ox0 = 0;                             // origin of the current patch
    if (tile < ntiles) {
        origin(tile, img, oy0, ox0);
#pragma unroll
        for (int i = 0; i < APW; ++i) issue_halo(i, img, oy0, ox0, 0);
    }
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int hbase[MT], kbase[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = wm * TM * 32 + i * MR + lrow, py = m / PW, px = m - py * PW;
        hbase[i] = py * HW + px;
        kbase[i] = px;
    }
    int w_off[NT][KS];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int wrow = wn * 32 + j * MR + lrow;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) w_off[j][ks] = wrow * 128 + (((F::CPK * ks + lk) ^ ((wrow >> 1) & 7)) << 4);
    }

    const unsigned lds_addr = (unsigned)(size_t)(lds_void*)lds;        // LDS byte address of the buffers (for the asm fragment reads)
    for (int it = 0; tile < ntiles; ++it, tile += gridDim.x) {
        const int next = tile + gridDim.x;
        int nimg = 0, noy0 = 0, nox0 = 0;
        if (next < ntiles) origin(next, nimg, noy0, nox0);
        typename F::acc_t acc[NT][MT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 4 * F::NQ; ++r) acc[j][i][r] = 0.f;
        // 9 taps x KS k-steps, fully unrolled; the fragments of step s + LOOK are requested before the MFMAs of step s (LOOK + 1 register
        // sets, counted waits): with one step of lookahead every step waited ~200 cycles (eight waves' LDS traffic) for 64 cycles of MFMAs
        constexpr int LOOK = M16 ? 1 : 3, RPS = MT + NT;               // reads per step
        static_assert(RPS * LOOK <= 15, "lgkmcnt is a 4-bit counter");
        bf16x8 xf[LOOK + 1][MT], wf[LOOK + 1][NT];
        const unsigned a_lds = lds_addr + (it & 1) * A_BYTES, w_lds = lds_addr + 2 * A_BYTES;
        auto load_step = [&](int set, int st) {
            const int t = st / KS, ks = st - t * KS;
            const int r = t / 3, s2 = t - 3 * r;
            const int dr = p.flip ? 2 - r : r, ds = p.flip ? 2 - s2 : s2;
#pragma unroll
            for (int i = 0; i < MT; ++i)
                xf[set][i] = lds_read16(a_lds + (hbase[i] + dr * HW + ds) * 128 + (((F::CPK * ks + lk) ^ (((kbase[i] + ds) >> 1) & 7)) << 4));
#pragma unroll
            for (int j = 0; j < NT; ++j) wf[set][j] = lds_read16(w_lds + t * 8192 + w_off[j][ks]);
        };
#pragma unroll
        for (int st = 0; st < LOOK; ++st) load_step(st, st);
#pragma unroll
        for (int st = 0; st < NSTEP; ++st) {
            if (st + LOOK < NSTEP) load_step((st + LOOK) % (LOOK + 1), st + LOOK);
            // the next patch's halo: one DMA per k-step at the start of the patch (an LDS-DMA holds the issue port ~100 cycles)
#ifndef K64_NO_DMA
            if (st < APW && next < ntiles) issue_halo(st, nimg, noy0, nox0, (it & 1) ^ 1);
#endif
            if (st == NSTEP * 3 / 4) wait_vmcnt<0>();         // well after the last DMA and a whole patch after the last stores: nothing left to wait for
            wait_lgkm_n(RPS * (NSTEP - 1 - st < LOOK ? NSTEP - 1 - st : LOOK));   // the reads of the steps after this one may stay in flight
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[j][i] = F::mma(wf[st % (LOOK + 1)][j], xf[st % (LOOK + 1)][i], acc[j][i]);
            __builtin_amdgcn_sched_barrier(0);
        }
        {
            auto pix_of_m = [&](int m, bool& ok) -> size_t {
                const int py = m / PW, px = m - py * PW;
                const int y = oy0 + py, x = ox0 + px;
                ok = y < p.H && x < p.W;
                return (size_t)(img * p.H + y) * p.W + x;
            };
            constexpr int SLOT = TM * 32 * (64 + 16);
            static_assert(8 * SLOT <= A_BYTES, "the eight staging slots reuse the halo buffer of the patch just multiplied");
#ifdef K64_NO_STORE
            if (acc[0][0][0] == 123.456f)                      // probe build: keep the accumulators alive, store nothing
#endif
            if (!p.out_f32 && !p.accumulate && (p.ldo & 7) == 0) {
                __builtin_amdgcn_s_barrier();                  // every wave has read its last fragment of this patch's halo
                asm volatile("" ::: "memory");
                store_tile_staged<M16, MT, NT>(p, acc, lds + (it & 1) * A_BYTES + wave * SLOT, wn * 32, lane,
                                               [&](int row, bool& ok) { return pix_of_m(wm * TM * 32 + row, ok); });
            } else {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    bool ok;
                    const size_t pix = pix_of_m(wm * TM * 32 + i * MR + lrow, ok);
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int g = 0; g < F::NQ; ++g) store_quad(p, F::quad(acc[j][i], g), ok, pix, wn * 32 + j * MR + F::quad_ch(g, lk));
                }
            }
        }
        __builtin_amdgcn_s_barrier();                          // every wave is done with this halo buffer and has seen the next one land
        asm volatile("" ::: "memory");
        img = nimg; oy0 = noy0; ox0 = nox0;
    }
}

int g_force_mode = -1;      // tuning aid: position space (0 / 1 / 2), -1 = by map size
int g_force_bn = -1;        // tuning aid: 64 / 128, -1 = by channel count
int g_k64 = 1;              // tuning aid: 0 = never the persistent K = 64 kernel
// MFMA shape: 0 = 32x32x16 (default), 1 = 16x16x32 (ssd_tune_set_conv_bf16_mfma).  Measured interleaved on one device (tools/conv_bf16_bench.py,
// batch 32): 631-637 vs 625-632 TFLOP/s over the forward layers, 686 vs 668-670 over the data gradients -- the kernels are not bound by MFMA
// issue, so the higher clock the chip holds with the 16x16x32 shape (MI355X_MICROARCH.md) buys nothing here, and its twice-as-many fragment
// registers cost the two-set read pipeline.
int g_m16 = 0;
// Zero columns between two rows of the flat position space (mode 2).  ONE column is the right pad of a row and the left pad of the next
// (pitch W + 1); rounds 1-3 kept two (pitch W + 2).  At 38 x 38 and batch 32 the space shrinks from 49 920 to 48 672 positions = 191
// instead of 195 tiles of 256, x 4 channel tiles = 764 blocks: THREE full rounds of the 256 CUs instead of three and a 5 % fourth
// (measured, tools/conv_bf16_bench.py interleaved: conv4_2 forward 0.283 -> 0.237 ms, data gradient 0.267 -> 0.213 ms = 1 022 TFLOP/s).
constexpr int g_flat_gap = 1;

template <int TM, int TN, int WVM, int WVN, int MODE, int PH, int APW, bool ADBL, bool M16>
int launch_shape(HaloParams& p, hipStream_t st);

template <int TM, int TN, int WVM, int WVN, int MODE, int PH, int APW, bool ADBL>
int launch(HaloParams& p, hipStream_t st) {
    if (g_m16) return launch_shape<TM, TN, WVM, WVN, MODE, PH, APW, ADBL, true>(p, st);
    return launch_shape<TM, TN, WVM, WVN, MODE, PH, APW, ADBL, false>(p, st);
}

template <int TM, int TN, int WVM, int WVN, int MODE, int PH, int APW, bool ADBL, bool M16>
int launch_shape(HaloParams& p, hipStream_t st) {
    constexpr int BM = WVM * TM * 32, BN = WVN * TN * 32;
    constexpr int PW = MODE == 1 ? 16 : 32;
    if (MODE == 2) {
        p.Wp = p.W + g_flat_gap;
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
    hipLaunchKernelGGL((conv3x3_bf16_kernel<TM, TN, WVM, WVN, MODE, PH, APW, ADBL, M16>), dim3(p.tiles_m * p.tiles_n), dim3(512), 0, st, p);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

int dispatch(HaloParams& p, hipStream_t st) {
    int mode = g_force_mode;
    // flat space (mode 2) while the halo of a 256-position tile -- 256 + 2 (W + 1) + 2 rows -- fits the buffers: 6 pieces per wave (384 rows)
    // up to W = 62, 7 pieces (448 rows; two such buffers + the weight ring are exactly the CU's 160 KB) up to W = 94.  Round 4: the
    // 75 x 75 maps (conv3_x) moved here from the 16 x 16 patches: 1 444 blocks instead of 1 600 (no 80 x 80 cover of a 75 x 75 map);
    // measured interleaved (tools/conv_bf16_bench.py): conv3_2 forward 0.290 -> 0.274 ms, data gradient 0.265 -> 0.245, conv3_1's 0.140 -> 0.120.
    const int flat_rows = 256 + 2 * (p.W + g_flat_gap) + 2;
    if (mode < 0) mode = flat_rows <= 448 ? 2 : ((p.W >= 128 && p.H >= 128) ? 0 : 1);
    if (mode == 2 && flat_rows > 448) mode = 1;
    int bn = g_force_bn;
    if (bn < 0) bn = p.Nout <= 64 ? 64 : 128;
    // a launch whose 128-channel tiles would occupy at most half of the CUs (the c_7 head's forward: 50 position tiles x 2 = 100 blocks with
    // 144 stages each) takes 64-channel tiles: 150 blocks of half the length (measured 0.107 -> 0.072 ms)
    if (g_force_bn < 0 && bn == 128 && mode == 2 && flat_rows <= 384 &&
        ssd_cdiv(p.N * (p.H + 1) * (p.W + g_flat_gap), 256) * ssd_cdiv(p.Nout, 128) <= 128)
        bn = 64;
    if (bn == 64 && mode == 0 && p.K == 64 && p.Nout <= 64 && g_k64 != 0) {
        p.npw = ssd_cdiv(p.W, 32);
        p.nph = ssd_cdiv(p.H, 8);
        p.tiles_m = p.N * p.npw * p.nph;
        p.tiles_n = 1;
        const int blocks = p.tiles_m < 256 ? p.tiles_m : 256;                 // one persistent workgroup per CU
        if (g_m16) hipLaunchKernelGGL(conv3x3_bf16_k64_kernel<true>, dim3(blocks), dim3(512), 0, st, p);
        else hipLaunchKernelGGL(conv3x3_bf16_k64_kernel<false>, dim3(blocks), dim3(512), 0, st, p);
        SSD_CHECK_LAUNCH();
        return SSD_OK;
    }
    if (bn == 64) {
        if (mode == 2 && flat_rows > 384) mode = 1;               // (the 64-channel flat form has the 6-piece buffers only)
        // 64 output channels: wave tile 64 x 64 over 512 positions (16 x 32 patches; one halo buffer, K has one or two chunks here)
        if (mode == 0) return launch<2, 2, 8, 1, 0, 16, 10, false>(p, st);
        if (mode == 1) return launch<2, 1, 4, 2, 1, 16, 6, true>(p, st);
        return launch<2, 1, 4, 2, 2, 0, 6, true>(p, st);
    }
    if (mode == 0) return launch<2, 2, 4, 2, 0, 8, 6, true>(p, st);
    if (mode == 1) return launch<2, 2, 4, 2, 1, 16, 6, true>(p, st);
    if (flat_rows > 384) return launch<2, 2, 4, 2, 2, 0, 7, true>(p, st);
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
// Tuning aid: MFMA shape of the bf16-tensor convolution kernels, 32 (v_mfma_f32_32x32x16_bf16, default) or 16 (v_mfma_f32_16x16x32_bf16).
extern "C" int ssd_tune_set_conv_bf16_mfma(int rows) {
    if (rows != 16 && rows != 32) return SSD_ERR_BAD_SHAPE;
    g_m16 = rows == 16;
    return SSD_OK;
}
// Tuning aid: 0 = conv1_2-shaped launches (K = 64, <= 64 output channels, 8 x 32 patches) on the general kernel instead of the persistent one.
extern "C" int ssd_tune_set_conv_bf16_k64(int on) {
    g_k64 = on ? 1 : 0;
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
    if ((long)N * H * W * ldx >= (1L << 31) || (long)w_rows * 9 * K >= (1L << 31)) return SSD_ERR_BAD_SHAPE;      // 32-bit element offsets in the kernels
    if (!ssd_aligned16(x) || !ssd_aligned16(w) || !ssd_aligned16(out) || (bias && !ssd_aligned16(bias)) || (relu_mask && !ssd_aligned16(relu_mask)))
        return SSD_ERR_ALIGN;
    HaloParams p = {};
    p.x = static_cast<const __bf16*>(x); p.w = static_cast<const __bf16*>(w); p.bias = bias; p.out = out;
    p.mask = static_cast<const __bf16*>(relu_mask);
    p.N = N; p.H = H; p.W = W; p.K = K; p.ldx = ldx; p.Nout = n_out; p.Nrows = w_rows; p.ldo = ldo;
    p.relu = relu; p.accumulate = accumulate; p.out_f32 = out_f32; p.flip = flip;
    return dispatch(p, (hipStream_t)stream);
}
