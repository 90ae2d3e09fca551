// Winograd F(4x4,3x3): the 36 plane GEMMs AND the output transform in one kernel (forward and dgrad of the 3x3 / stride-1 layers).
//
//   M[xi][tile][n] = sum_k V[xi][tile][k] * U[xi][n][k]      36 GEMMs            (winograd.hip: batched igemm, M planes to HBM)
//   y = A^T m A (+ bias, ReLU | + previous dx, ReLU mask | -> 2x2 max pool)      (winograd.hip: wino4_output*_kernel, M planes from HBM)
//
// In the two-kernel form the M planes (2.25x the output tensor, 1.66 GB for conv1_2 at batch 32) are written and read back; for the
// 64 / 128 / 256-channel layers that traffic, not the MFMA, sets the time.  Here a workgroup owns 32 tiles x 64 output channels and keeps
// ALL 36 planes of that block in accumulator registers (eight waves of 16 tiles x 16 channels: 36 x f32x4 = 144 per lane, two waves per
// SIMD), so the output transform is lane-local and M never exists in memory.
//
// Loop: stages (plane xi, 64-wide K chunk): A = V[xi][32 tiles][64 k] (8 KB), B = U[xi][64 n][64 k] (16 KB), brought in by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no ds_write) into a 5-slot ring, three stages ahead, counted vmcnt + raw s_barrier
// (cdna_hip_programming.md section 5, "Pipelining across barriers").  The ring rows are 256 B and unpadded (LDS-DMA writes 1 KB
// contiguously per wave instruction), so the 16-byte chunks of a row are XOR-swizzled by the row number on the SOURCE address and on
// the ds_read_b128 address alike (rule 21): the 16 rows a lane group reads land on 16 different chunks.
// MFMA: v_mfma_f32_16x16x4_f32, wave tile 16 tiles x 16 channels; its C/D layout puts a (tile, channel) pair's 36 plane values in one
// lane.  K is walked in the permuted order k = 16j + 4(lane>>4) + e for both operands (a sum does not care).
// Rows beyond the last tile / beyond the filter rows are clamped to the last valid row: their results are never stored.
#include "common.h"

namespace {

__device__ constexpr float F4_AT[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};

struct FusedParams {
    const float* V;            // [36][tiles][K]
    const float* U;            // [36][Nrows][K]
    size_t plane_v, plane_u;   // floats per plane
    int tiles, K, Nrows, Nout; // Nout: channels produced (multiple of 4)
    int groups, nblk_n;        // ceil(tiles / 32), ceil(Nout / 64)
    float* out; int ldo; int Cvalid;
    const float* bias; const float* mask; const unsigned long long* mask_bits; int relu, accumulate;
    int H, W, TH, TW;
    float* yp; uint8_t* am; int Ho, Wo;      // pooled form (yp != nullptr): conv -> ReLU -> 2x2 / stride-2 max pool
    unsigned long long* stamps;              // diagnostic (ssd_tune_set_wino_fused_stamps): 8 shader-clock stamps per 16th block
    int stagger;                             // start delay step (shader cycles) between the first workgroups of the CUs
};

constexpr int FG = 32, FN = 64, FK = 64;       // block tile: tiles x channels, K per stage
#ifndef WF_NS
#define WF_NS 5
#define WF_PD 3
#endif
constexpr int NS = WF_NS, PD = WF_PD;          // ring slots; LDS-DMA of stage s+PD is issued while stage s is multiplied
constexpr int STAGE_F = (FG + FN) * FK;        // floats per slot (24 KB)
constexpr int LOADS = 3;                       // LDS-DMA instructions per wave per stage (24 one-KB pieces over 8 waves)
constexpr int YS_F = FG * 16 * FN;             // epilogue image [32 tiles][16 px][64 ch] (128 KB)
constexpr int LDS_F = NS * STAGE_F > YS_F ? NS * STAGE_F : YS_F;
static_assert(NS - PD >= 2, "a slot is re-filled no earlier than two barriers after its last read");
static_assert(LDS_F * 4 <= 160 * 1024, "LDS");

typedef __attribute__((address_space(3))) void lds_void;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// 512 threads = 8 waves, two per SIMD: wave (mb, nb) = (wave >> 2, wave & 3) owns tiles mb*16 .. +15 x channels nb*16 .. +15 for all 36
// planes (144 accumulator registers).  Each wave has ONE dependent accumulator chain per plane (40-cycle latency against a 32-cycle
// issue slot); the partner wave on the same SIMD takes the other slot, and fills the matrix pipe while this one issues its LDS-DMA
// pieces and fragment reads.
template <int KCN>      // K / 64: the stage loop (36 planes x KCN chunks) is fully unrolled, every wait count and ring slot a constant
__global__ __launch_bounds__(512, 2) void wino4_gemm_out_kernel(const FusedParams p) {
    __shared__ __attribute__((aligned(16))) float lds[LDS_F];                 // the only LDS object of the kernel
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mb = wave >> 2, nbw = wave & 3;
    const int nblk = p.groups * p.nblk_n;
    const int lid = xcd_swizzle(blockIdx.x, nblk);
    const int grp = lid / p.nblk_n, nb = lid - grp * p.nblk_n;              // channel blocks of one tile group are neighbours in one L2
    const int tile0 = grp * FG, n0 = nb * FN;
    constexpr int S = 36 * KCN;
    const bool stamp = p.stamps != nullptr && (blockIdx.x & 15) == 0 && tid == 0;
    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (stamp) ts[0] = __builtin_readcyclecounter();
    // The first workgroup of every CU starts together and all of them have the same length: without help the whole chip loads, multiplies
    // and stores in lockstep and the stores of every CU hit HBM in one burst.  Delay the first round by a per-CU phase (speed only).
    if (p.stagger > 0 && blockIdx.x < 256) {
        const long long want = (long long)((blockIdx.x >> 3) & 7) * p.stagger;
        const long long t0 = (long long)__builtin_readcyclecounter();
        while ((long long)__builtin_readcyclecounter() - t0 < want) __builtin_amdgcn_s_sleep(64);
    }

    // ---- LDS-DMA pieces of this wave: piece q = 3 * wave + i of a stage's 24 (0..7: four A rows each, 8..23: four B rows each) --------
    const int lrow = lane >> 4, pc = lane & 15;                              // row within the piece, physical 16-byte chunk
    unsigned src_off[LOADS];
    bool is_a[LOADS];
    int lds_off[LOADS];
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
        const int q = 3 * wave + i;
        is_a[i] = q < 8;
        const int row = is_a[i] ? 4 * q + lrow : 4 * (q - 8) + lrow;
        const int src_row = is_a[i] ? min(tile0 + row, p.tiles - 1) : min(n0 + row, p.Nrows - 1);
        src_off[i] = (unsigned)src_row * (unsigned)p.K + (unsigned)((pc ^ (row & 15)) * 4);
        lds_off[i] = q * 256;                                               // A pieces then B pieces, contiguous: A = rows 0..31, B = 32..95
    }
    auto issue = [&](int stage) {             // `stage` is a compile-time constant at every call
        const int xi = stage / KCN, kc = stage % KCN, slot = stage % NS;
        const float* vsrc = p.V + (size_t)xi * p.plane_v + kc * FK;
        const float* usrc = p.U + (size_t)xi * p.plane_u + kc * FK;
        float* base = lds + slot * STAGE_F;
#pragma unroll
        for (int i = 0; i < LOADS; ++i)
            __builtin_amdgcn_global_load_lds((is_a[i] ? vsrc : usrc) + src_off[i], (lds_void*)(base + lds_off[i]), 16, 0, 0);
    };

    f32x4 acc[36];
#pragma unroll
    for (int xi = 0; xi < 36; ++xi) acc[xi] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int r15 = lane & 15, kq = lane >> 4;
    const int a_rd = (mb * 16 + r15) * FK, b_rd = FG * FK + (nbw * 16 + r15) * FK;
    int sw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) sw[j] = ((4 * j + kq) ^ r15) * 4;

    // The 8 fragment reads of stage s+1 are issued during the MFMAs of stage s (two register sets, roles fixed by the stage's parity).
    struct Frag { f32x4 a[4], b[4]; };
    Frag F[2];
    auto load_frags = [&](Frag& f, int slot) {
        const float* st = lds + slot * STAGE_F;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f.a[j] = *reinterpret_cast<const f32x4*>(st + a_rd + sw[j]);
            f.b[j] = *reinterpret_cast<const f32x4*>(st + b_rd + sw[j]);
        }
    };
    auto mma4 = [&](f32x4& c, const Frag& f, int j) {
#ifdef WF_NO_MFMA
        c += f.a[j] * f.b[j];
#else
#pragma unroll
        for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[j][e], f.b[j][e], c, 0, 0, 0);
#endif
    };

#pragma unroll
    for (int s = 0; s < PD; ++s)
        if (s < S) issue(s);
    wait_vmcnt<(PD < S ? PD - 1 : S - 1) * LOADS>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    load_frags(F[0], 0);
    if (stamp) ts[1] = __builtin_readcyclecounter();

    // Stage s: 16 MFMAs of one dependent chain (the partner wave of the SIMD owns the other issue slots).  The stage's three LDS-DMA
    // instructions, the wait + barrier that publish stage s+1 and the eight fragment reads of stage s+1 are spread between the four
    // groups of MFMAs, so that an issue stall of this wave costs matrix-pipe time only while the partner is stalled too.
#pragma clang loop unroll(full)
    for (int s = 0; s < S; ++s) {
        f32x4& c = acc[s / KCN];
        const Frag& cur = F[s & 1];
        Frag& nxt = F[(s & 1) ^ 1];
        mma4(c, cur, 0);
        __builtin_amdgcn_sched_barrier(0);
#ifndef WF_NO_LOAD
        if (s + PD < S) issue(s + PD);
#endif
        __builtin_amdgcn_sched_barrier(0);
        mma4(c, cur, 1);
        mma4(c, cur, 2);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < S) {
            // stage s+1 of THIS wave has landed once at most (issued - (s+1) - 1) stages' pieces are outstanding
            const int issued = (s + PD + 1 < S ? s + PD + 1 : S);
            const int ahead = issued - s - 2;
            if (ahead >= 3) wait_vmcnt<3 * LOADS>();
            else if (ahead == 2) wait_vmcnt<2 * LOADS>();
            else if (ahead == 1) wait_vmcnt<LOADS>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();                                     // every wave's pieces of stage s+1 have landed; stage s-1 is fully read
            asm volatile("" ::: "memory");
            load_frags(nxt, (s + 1) % NS);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma4(c, cur, 3);
    }

    if (stamp) ts[2] = __builtin_readcyclecounter();
    // ---- epilogue: lane-local A^T m A, then through LDS so that a thread owns (tile, 4 channels) and moves 16 bytes at a time --------
    // C/D layout of the 16x16 MFMA: register r is tile mb*16 + 4*(lane>>4) + r, channel nbw*16 + (lane&15)
    float* ys = lds;                                                          // [32 tiles][16 px][64 ch]
    const int C4 = p.Nout >> 2;
    __builtin_amdgcn_s_barrier();                                             // the ring is no longer read
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float t[4][6];
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const float m0 = acc[b][r], m1 = acc[6 + b][r], m2 = acc[12 + b][r], m3 = acc[18 + b][r], m4 = acc[24 + b][r], m5 = acc[30 + b][r];
            const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
            t[0][b] = m0 + s1 + s2;
            t[1][b] = d1 + 2.f * d2;
            t[2][b] = s1 + 4.f * s2;
            t[3][b] = d1 + 8.f * d2 + m5;
        }
        float* dst = ys + (size_t)((mb * 16 + 4 * kq + r) * 16) * FN + nbw * 16 + r15;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float s1 = t[i][1] + t[i][2], d1 = t[i][1] - t[i][2], s2 = t[i][3] + t[i][4], d2 = t[i][3] - t[i][4];
            dst[(i * 4 + 0) * FN] = t[i][0] + s1 + s2;
            dst[(i * 4 + 1) * FN] = d1 + 2.f * d2;
            dst[(i * 4 + 2) * FN] = s1 + 4.f * s2;
            dst[(i * 4 + 3) * FN] = d1 + 8.f * d2 + t[i][5];
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (stamp) ts[3] = __builtin_readcyclecounter();
    {
        // thread = (tile tl of the group, channel quad c4l)
        const int tl = tid >> 4, c4l = tid & 15;
        const int tile = tile0 + tl, c4 = (n0 >> 2) + c4l;
        if (tile < p.tiles && c4 < C4) {
            const int tw = tile % p.TW, th = (tile / p.TW) % p.TH, n = tile / (p.TW * p.TH);
            const float* src = ys + (size_t)(tl * 16) * FN + c4l * 4;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (p.bias != nullptr) {
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[e] = c4 * 4 + e < p.Cvalid ? p.bias[c4 * 4 + e] : 0.f;
            }
            if (p.yp == nullptr) {
                const unsigned long long word = p.mask_bits != nullptr ? p.mask_bits[(size_t)tile * C4 + c4] : 0ull;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int oh = 4 * th + a;
                    if (oh >= p.H) continue;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int ow = 4 * tw + b;
                        if (ow >= p.W) continue;
                        f32x4 v = *reinterpret_cast<const f32x4*>(src + (a * 4 + b) * FN) + bv;
                        const size_t idx = (((size_t)n * p.H + oh) * p.W + ow) * p.ldo + c4 * 4;
                        if (p.accumulate) v += *reinterpret_cast<const f32x4*>(p.out + idx);
                        if (p.relu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];
                        }
                        if (p.mask_bits != nullptr) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = ((word >> ((a * 4 + b) * 4 + e)) & 1ull) ? v[e] : 0.f;
                        } else if (p.mask != nullptr) {
                            const f32x4 mk = *reinterpret_cast<const f32x4*>(p.mask + idx);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = mk[e] > 0.f ? v[e] : 0.f;
                        }
                        *reinterpret_cast<f32x4*>(p.out + idx) = v;
                    }
                }
            } else {
                // conv -> ReLU -> MaxPool2d(2, 2): window scan order, NaN rule and argmax codes of maxpool_fwd_kernel (elementwise.hip)
#pragma unroll
                for (int pa = 0; pa < 2; ++pa) {
                    const int oh = 2 * th + pa;
                    if (oh >= p.Ho) continue;
#pragma unroll
                    for (int pb = 0; pb < 2; ++pb) {
                        const int ow = 2 * tw + pb;
                        if (ow >= p.Wo) continue;
                        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                        int bi[4] = {0, 0, 0, 0};
                        bool first = true;
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            if (2 * oh + r >= p.H) continue;
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                if (2 * ow + q >= p.W) continue;
                                f32x4 v = *reinterpret_cast<const f32x4*>(src + ((2 * pa + r) * 4 + 2 * pb + q) * FN) + bv;
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    v[e] = v[e] < 0.f ? 0.f : v[e];
                                    if (first || v[e] > best[e] || v[e] != v[e]) {
                                        best[e] = v[e];
                                        bi[e] = r * 2 + q;
                                    }
                                }
                                first = false;
                            }
                        }
                        const size_t o = (((size_t)n * p.Ho + oh) * p.Wo + ow) * C4 + c4;
                        *reinterpret_cast<f32x4*>(p.yp + o * 4) = best;
                        if (p.am != nullptr)
                            *reinterpret_cast<uint32_t*>(p.am + o * 4) =
                                (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
                    }
                }
            }
        }
    }
    if (stamp) {
        ts[4] = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* o = p.stamps + (size_t)(blockIdx.x >> 4) * 8;
        for (int i = 0; i < 5; ++i) o[i] = ts[i];
        o[5] = __builtin_readcyclecounter();
    }
}

}  // namespace
int ssd_internal_prof_open(double flops, int kind, hipStream_t st);
void ssd_internal_prof_close(int slot, hipStream_t st);
namespace {
unsigned long long* g_fused_stamps = nullptr;
int g_fused_stagger = -1;         // -1 / 0: none; > 0: delay step in shader cycles
}  // namespace

// Diagnostic: device buffer of 8 x uint64 per 16 blocks (start, main loop start, main loop end, output transformed, stores issued,
// stores complete; shader clock) filled by every 16th block of the fused launches that follow; NULL switches it off.
extern "C" int ssd_tune_set_wino_fused_stagger(int cycles) {
    g_fused_stagger = cycles;
    return SSD_OK;
}
extern "C" int ssd_tune_set_wino_fused_stamps(uint64_t* device_buffer) {
    g_fused_stamps = reinterpret_cast<unsigned long long*>(device_buffer);
    return SSD_OK;
}

// Internal (not part of the C ABI; called by winograd.hip's wino_conv): the GEMMs + output transform of one F(4x4,3x3) convolution on
// planes V [36][tiles][K] and filters U [36][Nrows][K].  Returns SSD_ERR_BAD_SHAPE when the geometry is not the kernel's (K % 64).
__attribute__((visibility("hidden"))) int ssd_internal_wino4_gemm_out(const float* V, const float* U, int tiles, int K, int Nrows, int Nout,
                                                                       float* out, int ldo, int Cvalid, const float* bias, const float* mask,
                                                                       const unsigned long long* mask_bits, int relu, int accumulate, int H, int W,
                                                                       int TH, int TW, float* yp, uint8_t* am, int Ho, int Wo, hipStream_t st) {
    if ((K != 64 && K != 128 && K != 256) || tiles <= 0 || Nout <= 0 || Nout % 4 != 0 || Nrows <= 0) return SSD_ERR_BAD_SHAPE;
    if ((size_t)tiles * K >= (1ull << 32) || (size_t)Nrows * K >= (1ull << 32)) return SSD_ERR_BAD_SHAPE;
    FusedParams p;
    p.V = V; p.U = U;
    p.plane_v = (size_t)tiles * K; p.plane_u = (size_t)Nrows * K;
    p.tiles = tiles; p.K = K; p.Nrows = Nrows; p.Nout = Nout;
    p.groups = ssd_cdiv(tiles, FG); p.nblk_n = ssd_cdiv(Nout, FN);
    p.out = out; p.ldo = ldo; p.Cvalid = Cvalid; p.bias = bias; p.mask = mask; p.mask_bits = mask_bits; p.relu = relu; p.accumulate = accumulate;
    p.H = H; p.W = W; p.TH = TH; p.TW = TW;
    p.yp = yp; p.am = am; p.Ho = Ho; p.Wo = Wo;
    p.stamps = g_fused_stamps;
    // The first-round start stagger (eight phases over about one workgroup's life: (K / 64) * 36 * 160 cycles per phase) made the kernel
    // 1-3 % faster by itself, but in the train step its idle start costs more than it returns (23.69 vs 23.84 ms, interleaved A/B): off
    // unless asked for.
    p.stagger = g_fused_stagger < 0 ? 0 : g_fused_stagger;
    const long long nblk = (long long)p.groups * p.nblk_n;
    if (nblk >= (1ll << 31)) return SSD_ERR_BAD_SHAPE;
    const int slot = ssd_internal_prof_open(2.0 * 36 * tiles * (double)K * (double)(p.nblk_n * FN), 1, st);   // FLOPs the grid executes
    switch (K / FK) {
        case 1: hipLaunchKernelGGL(wino4_gemm_out_kernel<1>, dim3((unsigned)nblk), dim3(512), 0, st, p); break;
        case 2: hipLaunchKernelGGL(wino4_gemm_out_kernel<2>, dim3((unsigned)nblk), dim3(512), 0, st, p); break;
        case 4: hipLaunchKernelGGL(wino4_gemm_out_kernel<4>, dim3((unsigned)nblk), dim3(512), 0, st, p); break;
        default: return SSD_ERR_BAD_SHAPE;
    }
    ssd_internal_prof_close(slot, st);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

// =====================================================================================================================================
// The whole F(4x4,3x3) convolution of a 64 / 128-channel layer in ONE kernel: input transform B^T d B, the 36 plane GEMMs and the output
// transform.  The two- and three-kernel forms move the V planes (2.25x the input: 1.66 GB for conv1_2 at batch 32) through HBM once or
// twice more than this one, which reads the activation and writes the output -- plus, in the training forward, the kept planes the weight
// gradient multiplies later (written, never read back here) and the ReLU bit mask of the input.
//
// Workgroup = 16 tiles x 64 output channels, 512 threads.  Phases, for each 64-channel half of K:
//   T  threads 0..255 = (tile, channel quad): 6x6 patch -> 36 values -> LDS Vs[plane][tile][64 k] (16-byte chunks XOR-swizzled by the tile
//      row, as the MFMA fragment reads want them) and, if asked, to the kept planes in HBM;
//   M  wave (nbw, ph) = (wave & 3, wave >> 2): 16 tiles x 16 channels x the 18 planes of transform rows 3 ph .. 3 ph + 2 (72 accumulator
//      registers); A fragments from LDS, B fragments (its own 16 filter rows) straight from L2 to registers, two planes ahead;
//   E  each wave applies A^T . A to its 18 planes (the transform is linear: the two halves add), both partial images go to LDS (V's space),
//      a thread per (tile, channel quad) adds them and runs the usual epilogue (bias / ReLU / accumulate / bit mask / 2x2 pool).
namespace {

__device__ constexpr float F4_BT[6][6] = {{4, 0, -5, 0, 1, 0}, {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0},
                                          {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};

struct FullParams {
    const float* x;            // (N,H,W,K) NHWC
    unsigned x_bytes;
    const float* U;            // [36][Nrows][K]
    size_t plane_u;
    int tiles, K, Nrows, Nout, groups, nblk_n;
    float* out; int ldo; int Cvalid;
    const float* bias; const float* mask; const unsigned long long* mask_bits; int relu, accumulate;
    int H, W, TH, TW;
    float* yp; uint8_t* am; int Ho, Wo;
    float* V_keep;             // [36][tiles][K] or null
    unsigned long long* bits_out;   // [tiles][K/4] or null
    unsigned long long* stamps;
};

#ifdef SSD_EXPERIMENTAL
constexpr int GT = 32;                          // tiles per workgroup
constexpr int KH = 32;                          // channels of K per pass (V of one pass: 36 x 32 tiles x 32 k floats = 144 KB of LDS)
constexpr int VS_F = 36 * GT * KH;
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Workgroup = 32 tiles x 64 output channels, 512 threads = 8 waves (mb, nbw) = (wave >> 2, wave & 3), each 16 tiles x 16 channels x ALL
// 36 planes (144 accumulator registers, as in wino4_gemm_out_kernel).  K is walked in passes of 32 channels:
//   T  every thread = (tile, channel pair): its 6x6 patch (36 8-byte loads, all in flight at once) -> 36 values -> LDS Vs[plane][tile][32 k]
//      (128-byte rows, 16-byte chunks XOR-swizzled by the tile-row pair) and, if asked, to the kept planes in HBM;
//   M  8 MFMAs per plane: A fragments from LDS, B fragments (the wave's 16 filter rows, 2 x 16 bytes per lane) straight from L2, requested
//      three planes ahead.  With 32 tiles per workgroup the filter stream is 16 B/cycle/CU at the full MFMA rate (16 tiles: 32 -- L2-bound).
//   then E as in wino4_gemm_out_kernel: lane-local A^T m A -> LDS image (V's space) -> a thread per (tile, channel quad): epilogue, 16-byte stores.
template <int KP>          // K / 32 passes (2 or 4)
__global__ __launch_bounds__(512, 2) void wino4_full_kernel(const FullParams p) {
    __shared__ __attribute__((aligned(16))) float lds[VS_F];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nbw = wave & 3, mb = wave >> 2;
    const int nblk = p.groups * p.nblk_n;
    const int lid = xcd_swizzle(blockIdx.x, nblk);
    const int grp = lid / p.nblk_n, nb = lid - grp * p.nblk_n;
    const int tile0 = grp * GT, n0 = nb * FN;
    const int r15 = lane & 15, kq = lane >> 4;
    const bool stamp = p.stamps != nullptr && (blockIdx.x & 15) == 0 && tid == 0;
    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (stamp) ts[0] = __builtin_readcyclecounter();

    f32x4 acc[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // B fragment source of this lane: filter row n0 + nbw*16 + r15 (clamped), 16-byte chunk 4j + kq of the pass (j = 0, 1)
    const int brow = min(n0 + nbw * 16 + r15, p.Nrows - 1);
    const float* bsrc = p.U + (size_t)brow * p.K + kq * 4;
    // A fragment address of this lane within a plane: row mb*16 + r15, chunk (4j + kq) ^ swizzle(row)
    const int arow = mb * 16 + r15;
    const int a_rd0 = arow * KH + ((kq ^ ((arow >> 1) & 7)) * 4), a_rd1 = arow * KH + (((4 + kq) ^ ((arow >> 1) & 7)) * 4);

    // T roles: tile tl = tid >> 4 (0..31), channel pair c2l = tid & 15 of the pass
    const int tl = tid >> 4, c2l = tid & 15;
    const int ttile = tile0 + tl;
    const bool live = ttile < p.tiles;
    const int tt = live ? ttile : p.tiles - 1;
    const int ttw = tt % p.TW, tth = (tt / p.TW) % p.TH, tn = tt / (p.TW * p.TH);
    const size_t plane_v = (size_t)p.tiles * p.K;
    const __amdgpu_buffer_rsrc_t srd_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);

#pragma unroll
    for (int kp = 0; kp < KP; ++kp) {
        // ---------------- T: input transform of channels kp*32 .. +31 of the group's 32 tiles ------------------------------------------
        if (kp > 0) __builtin_amdgcn_s_barrier();                             // the previous pass's fragments are all read
        {
            const int ch = kp * KH + c2l * 2;
            // buffer loads: one 32-bit offset per load instead of a 64-bit pointer (36 of them are in flight), pixels outside the image get
            // an out-of-range offset and the hardware range check returns zeros
            f32x2 t[6][6];
            const unsigned base = ((unsigned)tn * (unsigned)(p.H * p.W)) * (unsigned)p.K + (unsigned)ch;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                const int ih = 4 * tth - 1 + a;
#pragma unroll
                for (int b = 0; b < 6; ++b) {
                    const int iw = 4 * ttw - 1 + b;
                    const bool ok = live && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                    const unsigned voff = ok ? (base + (unsigned)(ih * p.W + iw) * (unsigned)p.K) * 4u : 0xFFFFFFF0u;
                    t[a][b] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(srd_x, (int)voff, 0, 0));
                }
            }
            if (p.bits_out != nullptr && nb == 0) {          // uniform.  Two lanes (channel pairs 2q, 2q+1) share the word of quad q
                unsigned long long word = 0ull;
#pragma unroll
                for (int a = 1; a <= 4; ++a)
#pragma unroll
                    for (int b = 1; b <= 4; ++b)
#pragma unroll
                        for (int e = 0; e < 2; ++e)
                            word |= (unsigned long long)(t[a][b][e] > 0.f) << (((a - 1) * 4 + (b - 1)) * 4 + (c2l & 1) * 2 + e);
                const unsigned lo = (unsigned)word, hi = (unsigned)(word >> 32);
                const unsigned olo = __shfl_xor((int)lo, 1, 64), ohi = __shfl_xor((int)hi, 1, 64);
                if (live && (c2l & 1) == 0)
                    p.bits_out[(size_t)ttile * (p.K >> 2) + (ch >> 2)] = word | ((unsigned long long)ohi << 32) | (unsigned long long)olo;
            }
#pragma unroll
            for (int b = 0; b < 6; ++b) {                    // B^T d, column by column, in place
                f32x2 d[6];
#pragma unroll
                for (int a = 0; a < 6; ++a) d[a] = t[a][b];
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    f32x2 s2 = {0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < 6; ++k)
                        if (F4_BT[a][k] != 0.f) s2 += F4_BT[a][k] * d[k];
                    t[a][b] = s2;
                }
            }
            float* vs = lds + tl * KH + (((c2l >> 1) ^ ((tl >> 1) & 7)) * 4) + (c2l & 1) * 2;
            float* vk = (p.V_keep != nullptr && live && nb == 0) ? p.V_keep + (size_t)ttile * p.K + ch : nullptr;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = 0; b < 6; ++b) {
                    f32x2 s2 = {0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < 6; ++k)
                        if (F4_BT[b][k] != 0.f) s2 += F4_BT[b][k] * t[a][k];
                    *reinterpret_cast<f32x2*>(vs + (a * 6 + b) * (GT * KH)) = s2;
                    if (vk != nullptr) *reinterpret_cast<f32x2*>(vk + (size_t)(a * 6 + b) * plane_v) = s2;
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (stamp && kp == 0) ts[1] = __builtin_readcyclecounter();

        // ---------------- M: 36 planes x 8 MFMAs; B fragments of plane i+3 and A fragments of plane i+1 requested before plane i ------
        const float* bk = bsrc + kp * KH;
        f32x4 bf[4][2], af[2][2];
        auto load_b = [&](int slot, int xi) {
            const float* src = bk + (size_t)xi * p.plane_u;
            bf[slot][0] = *reinterpret_cast<const f32x4*>(src);
            bf[slot][1] = *reinterpret_cast<const f32x4*>(src + 16);
        };
        auto load_a = [&](int slot, int xi) {
            const float* as = lds + xi * (GT * KH);
            af[slot][0] = *reinterpret_cast<const f32x4*>(as + a_rd0);
            af[slot][1] = *reinterpret_cast<const f32x4*>(as + a_rd1);
        };
        load_b(0, 0);
        load_b(1, 1);
        load_b(2, 2);
        load_a(0, 0);
#pragma unroll
        for (int i = 0; i < 36; ++i) {
#ifndef WF_FULL_NO_B
            if (i + 3 < 36) load_b((i + 3) % 4, i + 3);
#endif
#ifndef WF_FULL_NO_A
            if (i + 1 < 36) load_a((i + 1) % 2, i + 1);
#endif
            __builtin_amdgcn_sched_barrier(0);            // keep the requests AHEAD of their use (the scheduler would sink them)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i % 2][j][e], bf[i % 4][j][e], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---------------- E: lane-local output transform -> LDS image [32 tiles][16 px][64 ch] (V's space) -----------------------------------
    __builtin_amdgcn_s_barrier();                                             // V is no longer read
    if (stamp) ts[2] = __builtin_readcyclecounter();
    float* ys = lds;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float t[4][6];
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const float m0 = acc[b][r], m1 = acc[6 + b][r], m2 = acc[12 + b][r], m3 = acc[18 + b][r], m4 = acc[24 + b][r], m5 = acc[30 + b][r];
            const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
            t[0][b] = m0 + s1 + s2;
            t[1][b] = d1 + 2.f * d2;
            t[2][b] = s1 + 4.f * s2;
            t[3][b] = d1 + 8.f * d2 + m5;
        }
        float* dst = ys + (size_t)((mb * 16 + 4 * kq + r) * 16) * FN + nbw * 16 + r15;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float s1 = t[i][1] + t[i][2], d1 = t[i][1] - t[i][2], s2 = t[i][3] + t[i][4], d2 = t[i][3] - t[i][4];
            dst[(i * 4 + 0) * FN] = t[i][0] + s1 + s2;
            dst[(i * 4 + 1) * FN] = d1 + 2.f * d2;
            dst[(i * 4 + 2) * FN] = s1 + 4.f * s2;
            dst[(i * 4 + 3) * FN] = d1 + 8.f * d2 + t[i][5];
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (stamp) ts[3] = __builtin_readcyclecounter();
    {
        const int el = tid >> 4, c4l = tid & 15;
        const int tile = tile0 + el, c4 = (n0 >> 2) + c4l;
        const int C4o = p.Nout >> 2;
        if (tile < p.tiles && c4 < C4o) {
            const int tw = tile % p.TW, th = (tile / p.TW) % p.TH, n = tile / (p.TW * p.TH);
            const float* src = lds + (size_t)(el * 16) * FN + c4l * 4;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (p.bias != nullptr) {
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[e] = c4 * 4 + e < p.Cvalid ? p.bias[c4 * 4 + e] : 0.f;
            }
            if (p.yp == nullptr) {
                const unsigned long long word = p.mask_bits != nullptr ? p.mask_bits[(size_t)tile * C4o + c4] : 0ull;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int oh = 4 * th + a;
                    if (oh >= p.H) continue;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int ow = 4 * tw + b;
                        if (ow >= p.W) continue;
                        f32x4 v = *reinterpret_cast<const f32x4*>(src + (a * 4 + b) * FN) + bv;
                        const size_t idx = (((size_t)n * p.H + oh) * p.W + ow) * p.ldo + c4 * 4;
                        if (p.accumulate) v += *reinterpret_cast<const f32x4*>(p.out + idx);
                        if (p.relu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];
                        }
                        if (p.mask_bits != nullptr) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = ((word >> ((a * 4 + b) * 4 + e)) & 1ull) ? v[e] : 0.f;
                        } else if (p.mask != nullptr) {
                            const f32x4 mk = *reinterpret_cast<const f32x4*>(p.mask + idx);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = mk[e] > 0.f ? v[e] : 0.f;
                        }
                        *reinterpret_cast<f32x4*>(p.out + idx) = v;
                    }
                }
            } else {
#pragma unroll
                for (int pa = 0; pa < 2; ++pa) {
                    const int oh = 2 * th + pa;
                    if (oh >= p.Ho) continue;
#pragma unroll
                    for (int pb = 0; pb < 2; ++pb) {
                        const int ow = 2 * tw + pb;
                        if (ow >= p.Wo) continue;
                        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                        int bi[4] = {0, 0, 0, 0};
                        bool first = true;
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            if (2 * oh + r >= p.H) continue;
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                if (2 * ow + q >= p.W) continue;
                                f32x4 v = *reinterpret_cast<const f32x4*>(src + ((2 * pa + r) * 4 + 2 * pb + q) * FN) + bv;
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    v[e] = v[e] < 0.f ? 0.f : v[e];
                                    if (first || v[e] > best[e] || v[e] != v[e]) {
                                        best[e] = v[e];
                                        bi[e] = r * 2 + q;
                                    }
                                }
                                first = false;
                            }
                        }
                        const size_t o = (((size_t)n * p.Ho + oh) * p.Wo + ow) * C4o + c4;
                        *reinterpret_cast<f32x4*>(p.yp + o * 4) = best;
                        if (p.am != nullptr)
                            *reinterpret_cast<uint32_t*>(p.am + o * 4) =
                                (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
                    }
                }
            }
        }
    }
    if (stamp) {
        ts[4] = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* o = p.stamps + (size_t)(blockIdx.x >> 4) * 8;
        for (int i = 0; i < 5; ++i) o[i] = ts[i];
        o[5] = __builtin_readcyclecounter();
    }
}

#endif  // SSD_EXPERIMENTAL

}  // namespace

// Internal: the whole convolution from the NHWC activation (K = 64 or 128 channels).  V_keep / bits_out may be NULL.
__attribute__((visibility("hidden"))) int ssd_internal_wino4_full(const float* x, const float* U, int tiles, int K, int Nrows, int Nout, float* out,
                                                                   int ldo, int Cvalid, const float* bias, const float* mask,
                                                                   const unsigned long long* mask_bits, int relu, int accumulate, int H, int W,
                                                                   int TH, int TW, float* yp, uint8_t* am, int Ho, int Wo, float* V_keep,
                                                                   unsigned long long* bits_out, hipStream_t st) {
#ifndef SSD_EXPERIMENTAL
    return SSD_ERR_BAD_SHAPE;                       // not in this build (SSD_EXPERIMENTAL=1 python -m objectdetection_ssd_amd.build)
#else
    if ((K != 64 && K != 128) || tiles <= 0 || Nout <= 0 || Nout % 4 != 0 || Nrows <= 0) return SSD_ERR_BAD_SHAPE;
    if ((size_t)Nrows * K >= (1ull << 32)) return SSD_ERR_BAD_SHAPE;
    FullParams p;
    const size_t xb = (size_t)(tiles / (TH * TW)) * H * W * K * 4;
    if (xb >= 0xF0000000ull) return SSD_ERR_BAD_SHAPE;
    p.x = x; p.x_bytes = (unsigned)xb; p.U = U; p.plane_u = (size_t)Nrows * K;
    p.tiles = tiles; p.K = K; p.Nrows = Nrows; p.Nout = Nout;
    p.groups = ssd_cdiv(tiles, GT); p.nblk_n = ssd_cdiv(Nout, FN);
    p.out = out; p.ldo = ldo; p.Cvalid = Cvalid; p.bias = bias; p.mask = mask; p.mask_bits = mask_bits; p.relu = relu; p.accumulate = accumulate;
    p.H = H; p.W = W; p.TH = TH; p.TW = TW;
    p.yp = yp; p.am = am; p.Ho = Ho; p.Wo = Wo;
    p.V_keep = V_keep; p.bits_out = bits_out;
    p.stamps = g_fused_stamps;
    const long long nblk = (long long)p.groups * p.nblk_n;
    if (nblk >= (1ll << 31)) return SSD_ERR_BAD_SHAPE;
    const int slot = ssd_internal_prof_open(2.0 * 36 * (double)p.groups * GT * (double)K * (double)(p.nblk_n * FN), 2, st);
    if (K == 64) hipLaunchKernelGGL(wino4_full_kernel<2>, dim3((unsigned)nblk), dim3(512), 0, st, p);
    else hipLaunchKernelGGL(wino4_full_kernel<4>, dim3((unsigned)nblk), dim3(512), 0, st, p);
    ssd_internal_prof_close(slot, st);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
#endif
}
