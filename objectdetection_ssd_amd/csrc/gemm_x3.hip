// Batched plane GEMMs of the Winograd convolutions (Model.py:135-156: the 3x3 layers of the VGG trunk, fc6 and the heads) with f32
// operands multiplied on the bf16 MFMA: every f32 value is split EXACTLY into three bf16 limbs x = hi + mid + lo (8 + 8 + 8 significant
// bits, round-to-nearest: both residual subtractions are exact) and a product block is the six limb products of weight >= 2^-18
// (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid) on v_mfma_f32_32x32x16_bf16, accumulated in f32; the dropped terms (mid*lo, lo*mid,
// lo*lo) are <= 2^-26 relative, below the rounding of an f32 product.  Six bf16 MFMAs cost 6/16 of the f32 MFMA block they replace
// (v_mfma_f32_32x32x2_f32: 256 FLOP/clk/CU against 4096).  Non-finite and out-of-range operands -- where this differs from the f32 MFMA:
//   * x = +-inf splits into (inf, NaN, NaN) -- the residual is inf - inf -- so every output element whose reduction contains x comes out
//     NaN where the f32 MFMA returns +-inf (or NaN, if it meets a zero); NaN stays NaN.
//   * a FINITE |x| above the largest bf16 (0x7f7f0000 = 3.3895e38; f32 reaches 3.4028e38) rounds hi to +-inf and gives the same NaN where
//     the f32 MFMA returns a finite value or an overflowed +-inf.
//   * values below 2^-126 * 2^16 lose their low limbs to the bf16 denormal range (absolute error < 2^-133).
// In all three cases the affected output is non-finite (or differs below 1e-39) in BOTH forms' worst case: a step that reaches them has
// diverged already, and what is pinned is that the limb form never turns a non-finite result into a finite one
// (tests/test_gpu_kernels.py::test_limb_gemm_non_finite_operands, tests/test_gpu_path.py::test_overflowing_step_is_non_finite_in_both_gemm_forms).
//
//   NT form  out[b][m][n] = sum_k a[b][m][k] * w[b][n][k]      (forward / data gradient: a = transformed activation planes [tiles][K],
//                                                               w = transformed filter planes, split ONCE per step by the weight job)
//
// Structure: 128 x 128 tile, 4 waves (2 x 2, 64 x 64 each = 2 x 2 MFMA tiles), K step 16 (one MFMA depth), two LDS stages, ONE barrier
// per step (24 MFMAs per wave between barriers).  The weight limbs arrive pre-split and pre-tiled ([k step][limb][row][16]): one
// LDS-DMA instruction per limb and thread, 4 KB contiguous per limb.  The activation tile is loaded as f32 (two 16-byte loads per
// thread and step, two steps ahead into one of two register sets), split in registers (~50 VALU instructions: v_cvt_pk_bf16_f32 / shift /
// and / sub, placed by hand in the MFMA gaps) and written as three 16-byte LDS stores.  LDS rows are 32 bytes (16 k); the two 16-byte
// halves of row r are swapped when bit 3 of r is set, so that the 16 lanes of a ds_read_b128 group hit 16 different 16-byte slots
// (MI355X_MICROARCH.md LDS table).  The same kernel with an epilogue (bias / ReLU / += / ReLU mask, row stride) is the forward and the
// data gradient of the 1x1 layers with long reductions (fc7, seq8.0); the TN form below is every weight gradient.
#include <type_traits>
#include "common.h"

int ssd_internal_prof_open(double flops, int kind, hipStream_t st);      // conv_igemm.hip: the per-launch recorder behind ssd_prof_gemm_begin / _collect
void ssd_internal_prof_close(int slot, hipStream_t st);

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr unsigned OOB = 0xFFFFFF00u;

struct GemmX3Params {
    const float* __restrict__ a;        // [nbatch][M][K] f32
    const __bf16* __restrict__ w3;      // [nbatch][K/16][3][rows_pad][16] bf16 limbs (ssd_gemm_x3_split_weights)
    float* __restrict__ out;            // [nbatch][M][N] f32
    int M, K, N, rows_pad;
    int tiles_m, tiles_n, nbatch;
    size_t batch_a, batch_out;          // element strides between problems
    // epilogue of the 1x1 convolutions (ldo > 0): out[m * ldo + n] = [accumulate: out +] acc (+ bias[n]) -> ReLU -> ReLU mask; planes: ldo = 0
    const float* __restrict__ bias;
    const float* __restrict__ mask;     // [M][ldo]: the data gradient passes where mask > 0
    int ldo, relu, accumulate;
};

__device__ __forceinline__ f32x4 buf_load16(__amdgpu_buffer_rsrc_t srd, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, (int)voff, (int)soff, 0));
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// (used by the X3_PIPE experiment below)
// Workgroup barrier of the K loops: all of this wave's LDS traffic done (its stores visible, its fragment reads returned), then a RAW
// s_barrier.  __syncthreads() is a fence + barrier, and for the fence the compiler waits vmcnt(0) whenever an LDS-DMA may be in flight --
// every load requested for later steps, which is the whole point of requesting them early; the counted wait_vmcnt in front of each
// barrier says exactly which requests the next step needs.
// (the waits are the BUILTIN, not asm: the compiler's own wait-count pass reads it and so knows that this wave's fragment reads of the
// step have returned -- after an asm wait it still waited lgkmcnt(0) in front of the next step's first MFMA, i.e. for the reads just issued)
__device__ __forceinline__ void loop_barrier() {
    __builtin_amdgcn_s_waitcnt(0xC07F);           // lgkmcnt(0); vmcnt / expcnt fields at their maxima = no wait
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// two f32 -> the packed (a, b) bf16 pairs of their three limbs.  Round-to-nearest limbs (v_cvt_pk_bf16_f32): |mid| <= 2^-9 |x|,
// |lo| <= 2^-17 |x|, both residuals exact (x - bf16(x) has at most 16 significant bits, r1 - bf16(r1) at most 8), so hi + mid + lo = x
// and the dropped products are <= 2^-26 relative (truncated limbs: 2^-24, at the same instruction count).
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_rne(float a, float b) {
    const bf16x2 h = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = pack_rne(a, b);
    const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
    mid = pack_rne(ra, rb);
    lo = pack_rne(ra - __uint_as_float(mid << 16), rb - __uint_as_float(mid & 0xffff0000u));
}
__device__ __forceinline__ void split8(const f32x4 v0, const f32x4 v1, u32x4& hi, u32x4& mid, u32x4& lo) {
    unsigned h[4], m[4], l[4];
    split_pair(v0[0], v0[1], h[0], m[0], l[0]);
    split_pair(v0[2], v0[3], h[1], m[1], l[1]);
    split_pair(v1[0], v1[1], h[2], m[2], l[2]);
    split_pair(v1[2], v1[3], h[3], m[3], l[3]);
    hi = u32x4{h[0], h[1], h[2], h[3]};
    mid = u32x4{m[0], m[1], m[2], m[3]};
    lo = u32x4{l[0], l[1], l[2], l[3]};
}

// Row loads of the K loops as inline asm: 32 bytes of one row into two register quads.  The compiler does not know such a load is pending,
// so it neither waits for it at the first use (where, once an LDS-DMA is in flight, it waits vmcnt(0) -- the requests just issued for
// later steps included) nor may anything touch the destination before the counted wait that covers it.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void buf_load32_async(f32x4& d0, f32x4& d1, i32x4 srd, unsigned voff, unsigned soff) {
    asm volatile("buffer_load_dwordx4 %0, %2, %3, %4 offen\n\tbuffer_load_dwordx4 %1, %2, %3, %4 offen offset:16"
                 : "=&v"(d0), "=&v"(d1) : "v"(voff), "s"(srd), "s"(soff) : "memory");
}

constexpr int X3_LIMB = 128 * 32;          // bytes of one limb image: [128 rows][16 k] bf16
constexpr int X3_OPER = 3 * X3_LIMB;       // one operand, one stage
constexpr int X3_STAGE = 2 * X3_OPER;      // A + B

// M16: the same products on v_mfma_f32_16x16x32_bf16 -- TWO limb products per instruction, the limbs concatenated along the instruction's
// K = 32: [hi | mid] x [hi | hi] = hi hi + mid hi, [hi | mid] x [mid | mid] = hi mid + mid mid, [hi | lo] x [lo | hi] = hi lo + lo hi (lanes 0-31 of
// an operand read limb X, lanes 32-63 limb Y of the same 16 k).  Same cycles per FLOP as 32x32x16; the chip can hold a higher clock on this
// shape (MI355X_MICROARCH.md, DVFS give-back item 7).  20 fragment reads per step instead of 12, 48 MFMAs instead of 24.
template <bool EPI, bool M16 = false>       // EPI: the 1x1 convolutions' epilogue (its own instantiation: the plane GEMMs keep their straight store)
__global__ __launch_bounds__(256, M16 ? 2 : 3) void gemm_planes_x3_kernel(const GemmX3Params p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * X3_STAGE];       // 48 KB: three workgroups per CU
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;
    const int nblk = p.tiles_m * p.tiles_n * p.nbatch;
    int lid = xcd_swizzle(blockIdx.x, nblk);
    const int tile_n = lid % p.tiles_n;
    lid /= p.tiles_n;
    const int tile_m = lid % p.tiles_m, b = lid / p.tiles_m;
    const int m0 = tile_m * 128, n0 = tile_n * 128;
    const int NK = p.K >> 4;

    // A: thread -> (row, half): 8 consecutive k of one row per step
    const int arow = tid >> 1, ahalf = tid & 1;
    const __amdgpu_buffer_rsrc_t srd_a =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a + (size_t)b * p.batch_a), 0, (int)((size_t)p.M * p.K * 4), 0x00020000);
    const unsigned voff_a = m0 + arow < p.M ? ((unsigned)(m0 + arow) * (unsigned)p.K + ahalf * 8u) * 4u : OOB;
#ifdef X3_ASM_ALOAD
    const unsigned long long a_base = reinterpret_cast<unsigned long long>(p.a + (size_t)b * p.batch_a);      // the same descriptor as four words
    const i32x4 srd_a4 = {__builtin_amdgcn_readfirstlane((int)(unsigned)a_base), __builtin_amdgcn_readfirstlane((int)(unsigned)((a_base >> 32) & 0xffffu)),
                          __builtin_amdgcn_readfirstlane((int)((size_t)p.M * p.K * 4)), 0x00020000};      // (readfirstlane: provably scalar for the "s" operand)
#endif
    const unsigned a_wr = arow * 32 + ((ahalf ^ ((arow >> 3) & 1)) << 4);            // byte offset inside a limb image
    // B: thread -> 16-byte slot `tid` of each limb image; the XOR of the image is applied on the source address
    const int brow = tid >> 1, bhalf = (tid & 1) ^ ((brow >> 3) & 1);
    const size_t limb_elems = (size_t)p.rows_pad * 16;
    const __bf16* b_src = p.w3 + (size_t)b * NK * 3 * limb_elems + (size_t)(n0 + brow) * 16 + bhalf * 8;

    // two register sets for the raw activation rows: the loads of step s + 2 are issued at the start of step s, the split of
    // step s + 1's rows (issued a step earlier) runs in the shadow of step s's MFMAs
    f32x4 ra[2][2];
    auto load_a = [&](int ks, int set) {          // beyond the last step: out-of-range offsets (zeros, no traffic) keep vmcnt uniform
#ifdef X3_PROBE_NO_ALOAD                     // timing probes (tools/x3_variants.sh): a piece compiled out, results meaningless
        const unsigned v = OOB;
#else
        const unsigned v = ks < NK ? voff_a : OOB;
#endif
#ifdef X3_ASM_ALOAD
        buf_load32_async(ra[set][0], ra[set][1], srd_a4, v, (unsigned)ks * 64u);
#else
        ra[set][0] = buf_load16(srd_a, v, (unsigned)ks * 64u);
        ra[set][1] = buf_load16(srd_a, v + 16u, (unsigned)ks * 64u);
#endif
    };
    // Sign dither.  The bf16 MFMA's internal sum is truncated, not rounded: against f64 the result carries a bias of about -1.7e-10 of
    // its magnitude per MFMA of the chain, always downwards (tools/x3_bias_probe.py: -3.3e-8 at K = 512, -1.3e-7 at K = 2048, the same
    // for positive, negative and mixed data; the f32 MFMA: < 1e-9).  That is 1/10 of the random rounding error of one product, but it
    // is coherent: through the Winograd output transform and ten layers of backward it sums where rounding errors average out.
    // So row m of the activation operand enters with sign s(m) = +-1 (bit 31 of the raw f32 words, before the split: limbs of -x are
    // minus the limbs of x) and the finished row is multiplied by s(m) again: the bias of an output element keeps its size but takes
    // the sign s(m), s(m) = (-1)^(bit 2 ^ bit 5 of m): zero mean over any 8 consecutive rows.
    const unsigned a_sign = (unsigned)(((arow >> 2) ^ (arow >> 5)) & 1) << 31;
    auto dma_b = [&](int ks, int stage) {
#ifdef X3_PROBE_NO_BDMA
        const __bf16* s = b_src;                  // the same 12 KB every step: L1 / L2 hits
#else
        const __bf16* s = b_src + (size_t)ks * 3 * limb_elems;
#endif
        unsigned char* d = lds + stage * X3_STAGE + X3_OPER + wave * 1024;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(s + pl * limb_elems), (lds_void*)(d + pl * X3_LIMB), 16, 0, 0);
    };

    f32x16 acc[2][2];
    f32x4 acc4[M16 ? 4 : 1][M16 ? 4 : 1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int i = 0; i < (M16 ? 4 : 1); ++i)
#pragma unroll
        for (int j = 0; j < (M16 ? 4 : 1); ++j) acc4[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const unsigned frag = lr * 32 + ((lh ^ ((lr >> 3) & 1)) << 4);
    const unsigned a_rd = wm * 64 * 32 + frag, b_rd = X3_OPER + wn * 64 * 32 + frag;
    // M16: lane = (row r16 of a 16-row block, quarter g4): k half g4 & 1 of limb X (g4 < 2) or limb Y (g4 >= 2)
    const int r16 = lane & 15, g4 = lane >> 4;
    const unsigned frag16 = r16 * 32 + (((g4 & 1) ^ ((r16 >> 3) & 1)) << 4);
    const unsigned up = g4 >> 1;                              // 0: this lane reads the first limb of the pair, 1: the second
    // A' kinds: 0 = [hi | mid], 1 = [hi | lo];  B' kinds: 0 = [hi | hi], 1 = [mid | mid], 2 = [lo | hi]
    const unsigned a16[2] = {wm * 64 * 32 + frag16 + up * X3_LIMB, wm * 64 * 32 + frag16 + up * 2 * X3_LIMB};
    const unsigned b16[3] = {X3_OPER + wn * 64 * 32 + frag16, X3_OPER + wn * 64 * 32 + frag16 + X3_LIMB,
                             X3_OPER + wn * 64 * 32 + frag16 + (1 - up) * 2 * X3_LIMB};

    // one step on LDS stage PH: 12 fragment reads, then 24 MFMAs (product-major, smallest limb products first) with the split of the
    // other register set's rows placed by hand between them -- one stage (<= 4 VALU instructions) of one pair per MFMA, pinned by
    // sched_barrier: a 32-cycle v_mfma_f32_32x32x16_bf16 hides up to five 4-cycle instructions -- and the three LDS stores at the end
    auto step = [&](auto ph_tag) {
        constexpr int PH = decltype(ph_tag)::value;
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
        // (compiler-visible reads: as inline asm with hand-counted lgkmcnt the first MFMAs started four reads earlier, +3 % in isolation,
        // but under this kernel's register pressure the allocator may copy a fragment register between the asm read and the wait --
        // it cannot know the data has not arrived -- and one build of the 1x1 instantiation did exactly that)
        const unsigned char* st = lds + PH * X3_STAGE;
        bf16x8 af[3][2], bf[3][2];
        bf16x8 af4[2][4], bf4[3][4];
        if constexpr (!M16) {
#pragma unroll
        for (int g = 0; g < 3; ++g)                // in the order the products need them
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int pa = g == 0 ? 2 : (g == 1 ? 0 : 1), pb = g == 0 ? 0 : (g == 1 ? 2 : 1);
                af[pa][i] = *reinterpret_cast<const bf16x8*>(st + a_rd + pa * X3_LIMB + i * 1024);
                bf[pb][i] = *reinterpret_cast<const bf16x8*>(st + b_rd + pb * X3_LIMB + i * 1024);
            }
        __builtin_amdgcn_sched_barrier(0);
#ifdef X3_SETPRIO
        __builtin_amdgcn_s_setprio(X3_SETPRIO);
#endif
        } else {
#pragma unroll
            for (int g = 0; g < 3; ++g) {          // in the order the products need them: [hi|lo] x [lo|hi], then [hi|mid] x [mid|mid], then x [hi|hi]
                const int ka = g == 0 ? 1 : 0, kb = g == 0 ? 2 : (g == 1 ? 1 : 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (g < 2) af4[ka][i] = *reinterpret_cast<const bf16x8*>(st + a16[ka] + i * 512);
                    bf4[kb][i] = *reinterpret_cast<const bf16x8*>(st + b16[kb] + i * 512);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#ifdef X3_SETPRIO
            __builtin_amdgcn_s_setprio(X3_SETPRIO);
#endif
        }
        const f32x4 v0 = ra[PH ^ 1][0], v1 = ra[PH ^ 1][1];
        float xa[4] = {v0[0], v0[2], v1[0], v1[2]}, xb[4] = {v0[1], v0[3], v1[1], v1[3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            xa[e] = __uint_as_float(__float_as_uint(xa[e]) ^ a_sign);
            xb[e] = __uint_as_float(__float_as_uint(xb[e]) ^ a_sign);
        }
        unsigned hi[4], mid[4], lo[4];
        float r1a[4], r1b[4];
        if constexpr (M16) {
#pragma unroll
            for (int q = 0; q < 48; ++q) {
                const int pr = q >> 4, i = (q >> 2) & 3, j = q & 3;
                const int ka = pr == 0 ? 1 : 0, kb = pr == 0 ? 2 : (pr == 1 ? 1 : 0);
                acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af4[ka][i], bf4[kb][j], acc4[i][j], 0, 0, 0);
                if (q >= 4 && q < 44 && (q & 1) == 0) {      // one split stage per two 16-cycle MFMAs
                    const int sl = (q - 4) >> 1, e = sl / 5, sg = sl % 5;
                    if (sg == 0) { hi[e] = pack_rne(xa[e], xb[e]); asm volatile("" : "+v"(hi[e])); }
                    if (sg == 1) {
                        r1a[e] = xa[e] - __uint_as_float(hi[e] << 16); r1b[e] = xb[e] - __uint_as_float(hi[e] & 0xffff0000u);
                        asm volatile("" : "+v"(r1a[e]), "+v"(r1b[e]));
                    }
                    if (sg == 2) { mid[e] = pack_rne(r1a[e], r1b[e]); asm volatile("" : "+v"(mid[e])); }
                    if (sg == 3) {
                        r1a[e] -= __uint_as_float(mid[e] << 16); r1b[e] -= __uint_as_float(mid[e] & 0xffff0000u);
                        asm volatile("" : "+v"(r1a[e]), "+v"(r1b[e]));
                    }
                    if (sg == 4) { lo[e] = pack_rne(r1a[e], r1b[e]); asm volatile("" : "+v"(lo[e])); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
        for (int q = 0; q < 24; ++q) {
            const int pr = q >> 2, i = (q >> 1) & 1, j = q & 1;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[pr]][i], bf[PB[pr]][j], acc[i][j], 0, 0, 0);
#ifndef X3_PROBE_NO_SPLIT
            if (q >= 2 && q < 22) {                // (the first MFMAs wait for the fragments anyway)
                const int e = (q - 2) / 5, sg = (q - 2) % 5;
                // (the empty asm pins each stage's results where they are written: pure arithmetic would otherwise sink, at IR level,
                // past every MFMA to its only use, the LDS stores)
                if (sg == 0) { hi[e] = pack_rne(xa[e], xb[e]); asm volatile("" : "+v"(hi[e])); }
                if (sg == 1) {
                    r1a[e] = xa[e] - __uint_as_float(hi[e] << 16); r1b[e] = xb[e] - __uint_as_float(hi[e] & 0xffff0000u);
                    asm volatile("" : "+v"(r1a[e]), "+v"(r1b[e]));
                }
                if (sg == 2) { mid[e] = pack_rne(r1a[e], r1b[e]); asm volatile("" : "+v"(mid[e])); }
                if (sg == 3) {
                    r1a[e] -= __uint_as_float(mid[e] << 16); r1b[e] -= __uint_as_float(mid[e] & 0xffff0000u);
                    asm volatile("" : "+v"(r1a[e]), "+v"(r1b[e]));
                }
                if (sg == 4) { lo[e] = pack_rne(r1a[e], r1b[e]); asm volatile("" : "+v"(lo[e])); }
            }
#else
            if (q == 2) { for (int e = 0; e < 4; ++e) { hi[e] = __float_as_uint(xa[e]); mid[e] = __float_as_uint(xb[e]); lo[e] = hi[e] ^ mid[e]; } }
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        }
#ifdef X3_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        unsigned char* d = lds + (PH ^ 1) * X3_STAGE + a_wr;       // the other stage: last read a step ago, a barrier since
        *reinterpret_cast<u32x4*>(d) = u32x4{hi[0], hi[1], hi[2], hi[3]};
        *reinterpret_cast<u32x4*>(d + X3_LIMB) = u32x4{mid[0], mid[1], mid[2], mid[3]};
        *reinterpret_cast<u32x4*>(d + 2 * X3_LIMB) = u32x4{lo[0], lo[1], lo[2], lo[3]};
    };

    // prologue: stage 0 complete, stage 1's weights and the rows of steps 1 and 2 in flight
    dma_b(0, 0);
    load_a(0, 0);
    load_a(1, 1);
#ifdef X3_ASM_ALOAD
    wait_vmcnt<0>();
#endif
    {
        u32x4 hi, mid, lo;
        f32x4 s0 = ra[0][0], s1 = ra[0][1];       // (the compiler waits for set 0 here)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s0[e] = __uint_as_float(__float_as_uint(s0[e]) ^ a_sign);
            s1[e] = __uint_as_float(__float_as_uint(s1[e]) ^ a_sign);
        }
        split8(s0, s1, hi, mid, lo);
        unsigned char* d = lds + a_wr;
        *reinterpret_cast<u32x4*>(d) = hi;
        *reinterpret_cast<u32x4*>(d + X3_LIMB) = mid;
        *reinterpret_cast<u32x4*>(d + 2 * X3_LIMB) = lo;
    }
    dma_b(1, 1);
    load_a(2, 0);
    wait_vmcnt<7>();                              // issue order: B(0), A(0), A(1), B(1), A(2): everything up to A(0) has landed
    __syncthreads();
    for (int ks = 0; ks < NK; ks += 2) {          // NK is even (K % 32 == 0)
        // (__syncthreads, not a raw barrier: its fence makes the compiler drain every load HERE, a full step after the request; with a
        // raw barrier it drains them at the first use of the row registers instead -- right after the next requests: 4.6 -> 4.9 ms per step)
        step(std::integral_constant<int, 0>{});   // step ks: reads stage 0, writes A(ks + 1) into stage 1
#ifdef X3_ASM_ALOAD
        wait_vmcnt<0>();                          // everything requested at the top of this step: the weights of step ks + 1, the rows of ks + 2
#else
        wait_vmcnt<2>();                          // B(ks + 1) has landed
#endif
        __syncthreads();
        if (ks + 2 < NK) dma_b(ks + 2, 0);
        load_a(ks + 3, 1);
        step(std::integral_constant<int, 1>{});   // step ks + 1
#ifdef X3_ASM_ALOAD
        wait_vmcnt<0>();
#else
        wait_vmcnt<2>();
#endif
        __syncthreads();
        if (ks + 3 < NK) dma_b(ks + 3, 1);
        load_a(ks + 4, 0);
    }
    wait_vmcnt<0>();

    float* out = p.out + (size_t)b * p.batch_out;
    auto store = [&](int m, int n, float v) {
        if (m < p.M && n < p.N) {
            if constexpr (!EPI) {
                out[(size_t)m * p.N + n] = v;
            } else {                                                                 // (same order as igemm_epilogue, conv_igemm.hip)
                const size_t idx = (size_t)m * p.ldo + n;
                if (p.bias != nullptr) v += p.bias[n];
                if (p.accumulate) v += out[idx];
                if (p.relu) v = v < 0.f ? 0.f : v;                                   // NaN stays NaN, like torch.relu
                if (p.mask != nullptr) v = p.mask[idx] > 0.f ? v : 0.f;
                out[idx] = v;
            }
        }
    };
    if constexpr (M16) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + j * 16 + r16;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * 64 + i * 16 + 4 * g4 + r;                    // bits 2 and 5 of m: g4 & 1 and i >> 1
                    store(m, n, ((g4 ^ (i >> 1)) & 1) ? -acc4[i][j][r] : acc4[i][j][r]); // the row's sign s(m) again
                }
            }
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + j * 32 + lr;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 64 + i * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);   // bits 2 and 5 of m: lh and i
                    store(m, n, ((lh ^ i) & 1) ? -acc[i][j][r] : acc[i][j][r]);              // the row's sign s(m) again
                }
            }
    }
}

#ifdef X3_PIPE      // EXPERIMENT, not built by default (tools/x3_variants.sh "" "-DX3_PIPE"): measured below
// ---- the same NT product, software-pipelined: THREE LDS stages, TWO fragment register sets, THREE raw-row register sets.
// At the top of step t a wave requests the weights of step t + 3 (LDS-DMA into the stage whose fragments it already holds) and the rows of
// step t + 4 (registers), reads the FRAGMENTS OF STEP t + 1 from LDS, and then issues the 24 MFMAs of step t on the fragments it read a
// step ago -- no LDS latency in front of them -- with the split of the rows of step t + 2 (requested two steps ago) in their gaps.  One
// counted wait per step (vmcnt(5): everything but this step's own five requests) and one raw barrier.  72 KB of LDS and ~200 registers:
// two workgroups per CU.
// The row loads are inline asm: for its own loads the compiler, once an LDS-DMA is in flight, waits vmcnt(0) in front of the first use
// -- this step's fresh requests included -- which is exactly the stall the pipeline exists to remove.  An asm load completes later than
// the compiler believes; nothing may touch its destination registers before the counted wait two steps on.  The parity tests run this
// very binary on every shape class (a copy of such a register made too early gives garbage, not a small error).
// Probes on the two-stage kernel (tools/x3_variants.sh) put its skeleton -- no global loads, no split -- at 1.28 PFLOP/s: its three
// waves per SIMD fall into lock-step (MFMA phase together, LDS / barrier phase together) and every load was drained at every barrier.

template <bool EPI>
__global__ __launch_bounds__(256, 2) void gemm_planes_x3p_kernel(const GemmX3Params p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[3 * X3_STAGE];       // 72 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;
    const int nblk = p.tiles_m * p.tiles_n * p.nbatch;
    int lid = xcd_swizzle(blockIdx.x, nblk);
    const int tile_n = lid % p.tiles_n;
    lid /= p.tiles_n;
    const int tile_m = lid % p.tiles_m, b = lid / p.tiles_m;
    const int m0 = tile_m * 128, n0 = tile_n * 128;
    const int NK = p.K >> 4;

    const int arow = tid >> 1, ahalf = tid & 1;
    const __amdgpu_buffer_rsrc_t srd_a =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a + (size_t)b * p.batch_a), 0, (int)((size_t)p.M * p.K * 4), 0x00020000);
    const unsigned long long a_base = reinterpret_cast<unsigned long long>(p.a + (size_t)b * p.batch_a);      // the same descriptor as four words
    const i32x4 srd_a4 = {__builtin_amdgcn_readfirstlane((int)(unsigned)a_base), __builtin_amdgcn_readfirstlane((int)(unsigned)((a_base >> 32) & 0xffffu)),
                          __builtin_amdgcn_readfirstlane((int)((size_t)p.M * p.K * 4)), 0x00020000};      // (readfirstlane: provably scalar for the "s" operand)
    const unsigned voff_a = m0 + arow < p.M ? ((unsigned)(m0 + arow) * (unsigned)p.K + ahalf * 8u) * 4u : OOB;
    const unsigned a_wr = arow * 32 + ((ahalf ^ ((arow >> 3) & 1)) << 4);
    const int brow = tid >> 1, bhalf = (tid & 1) ^ ((brow >> 3) & 1);
    const size_t limb_elems = (size_t)p.rows_pad * 16;
    const __bf16* b_src = p.w3 + (size_t)b * NK * 3 * limb_elems + (size_t)(n0 + brow) * 16 + bhalf * 8;
    const unsigned a_sign = (unsigned)(((arow >> 2) ^ (arow >> 5)) & 1) << 31;       // sign dither: see gemm_planes_x3_kernel

    f32x4 ra[3][2];                               // raw rows of three steps in flight
    auto dma_b = [&](int ks, int stage) {         // beyond the last step: step 0's tile again (never read), so that every wait count is a constant
        const __bf16* s = b_src + (size_t)(ks < NK ? ks : 0) * 3 * limb_elems;
        unsigned char* d = lds + stage * X3_STAGE + X3_OPER + wave * 1024;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(s + pl * limb_elems), (lds_void*)(d + pl * X3_LIMB), 16, 0, 0);
    };
    auto split_store = [&](f32x4 s0, f32x4 s1, int stage) {  // (prologue only: the loop splits in the MFMA gaps)
        u32x4 hi, mid, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s0[e] = __uint_as_float(__float_as_uint(s0[e]) ^ a_sign);
            s1[e] = __uint_as_float(__float_as_uint(s1[e]) ^ a_sign);
        }
        split8(s0, s1, hi, mid, lo);
        unsigned char* d = lds + stage * X3_STAGE + a_wr;
        *reinterpret_cast<u32x4*>(d) = hi;
        *reinterpret_cast<u32x4*>(d + X3_LIMB) = mid;
        *reinterpret_cast<u32x4*>(d + 2 * X3_LIMB) = lo;
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const unsigned frag = lr * 32 + ((lh ^ ((lr >> 3) & 1)) << 4);
    const unsigned a_rd = wm * 64 * 32 + frag, b_rd = X3_OPER + wn * 64 * 32 + frag;
    bf16x8 af[2][3][2], bf[2][3][2];              // [fragment set][limb][MFMA tile]

    // step (I = position in the unrolled group of six: every stage / set index below is a constant):
    //   fragments of the NEXT step from stage (I + 1) % 3 into set (I + 1) & 1, 24 MFMAs on set I & 1 with the split of raw set
    //   (I + 2) % 3 (the rows of two steps on) in their gaps, limbs stored into stage (I + 2) % 3
    auto step = [&](auto i_tag) {
        constexpr int I = decltype(i_tag)::value, PH = I & 1, S1 = (I + 1) % 3, S2 = (I + 2) % 3, RS = (I + 2) % 3;
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
        {
            const unsigned char* st = lds + S1 * X3_STAGE;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[PH ^ 1][pl][i] = *reinterpret_cast<const bf16x8*>(st + a_rd + pl * X3_LIMB + i * 1024);
                    bf[PH ^ 1][pl][i] = *reinterpret_cast<const bf16x8*>(st + b_rd + pl * X3_LIMB + i * 1024);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 v0 = ra[RS][0], v1 = ra[RS][1];
        float xa[4] = {v0[0], v0[2], v1[0], v1[2]}, xb[4] = {v0[1], v0[3], v1[1], v1[3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            xa[e] = __uint_as_float(__float_as_uint(xa[e]) ^ a_sign);
            xb[e] = __uint_as_float(__float_as_uint(xb[e]) ^ a_sign);
        }
        unsigned hi[4], mid[4], lo[4];
        float r1a[4], r1b[4];
#pragma unroll
        for (int q = 0; q < 24; ++q) {
            const int pr = q >> 2, i = (q >> 1) & 1, j = q & 1;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PH][PA[pr]][i], bf[PH][PB[pr]][j], acc[i][j], 0, 0, 0);
            if (q >= 2 && q < 22) {
                const int e = (q - 2) / 5, sg = (q - 2) % 5;
                if (sg == 0) { hi[e] = pack_rne(xa[e], xb[e]); asm volatile("" : "+v"(hi[e])); }
                if (sg == 1) {
                    r1a[e] = xa[e] - __uint_as_float(hi[e] << 16); r1b[e] = xb[e] - __uint_as_float(hi[e] & 0xffff0000u);
                    asm volatile("" : "+v"(r1a[e]), "+v"(r1b[e]));
                }
                if (sg == 2) { mid[e] = pack_rne(r1a[e], r1b[e]); asm volatile("" : "+v"(mid[e])); }
                if (sg == 3) {
                    r1a[e] -= __uint_as_float(mid[e] << 16); r1b[e] -= __uint_as_float(mid[e] & 0xffff0000u);
                    asm volatile("" : "+v"(r1a[e]), "+v"(r1b[e]));
                }
                if (sg == 4) { lo[e] = pack_rne(r1a[e], r1b[e]); asm volatile("" : "+v"(lo[e])); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        unsigned char* d = lds + S2 * X3_STAGE + a_wr;
        *reinterpret_cast<u32x4*>(d) = u32x4{hi[0], hi[1], hi[2], hi[3]};
        *reinterpret_cast<u32x4*>(d + X3_LIMB) = u32x4{mid[0], mid[1], mid[2], mid[3]};
        *reinterpret_cast<u32x4*>(d + 2 * X3_LIMB) = u32x4{lo[0], lo[1], lo[2], lo[3]};
    };
    auto rows = [&](int ks) { return ks < NK ? voff_a : OOB; };      // beyond the last step: zeros, no traffic, the same request count

    // prologue: stages 0 and 1 complete, stage 2's weights landed, the rows of steps 2 and 3 in raw sets 2 and 0, the fragments of step 0 in set 0
    dma_b(0, 0);
    dma_b(1, 1);
    dma_b(2, 2);
    {
        const f32x4 p00 = buf_load16(srd_a, rows(0), 0u), p01 = buf_load16(srd_a, rows(0) + 16u, 0u);
        const f32x4 p10 = buf_load16(srd_a, rows(1), 64u), p11 = buf_load16(srd_a, rows(1) + 16u, 64u);
        split_store(p00, p01, 0);
        split_store(p10, p11, 1);
    }
    buf_load32_async(ra[2][0], ra[2][1], srd_a4, rows(2), 2u * 64u);
    buf_load32_async(ra[0][0], ra[0][1], srd_a4, rows(3), 3u * 64u);
    wait_vmcnt<0>();
    loop_barrier();
    {
        const unsigned char* st = lds;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[0][pl][i] = *reinterpret_cast<const bf16x8*>(st + a_rd + pl * X3_LIMB + i * 1024);
                bf[0][pl][i] = *reinterpret_cast<const bf16x8*>(st + b_rd + pl * X3_LIMB + i * 1024);
            }
    }
    loop_barrier();                               // stage 0 is free again: its fragments are in registers
#define X3P_STEP(I)                                                                                                       \
    if (t + I < NK) {                                                                                                     \
        buf_load32_async(ra[(I + 1) % 3][0], ra[(I + 1) % 3][1], srd_a4, rows(t + I + 4), (unsigned)(t + I + 4) * 64u);   \
        dma_b(t + I + 3, I % 3);                                                                                          \
        step(std::integral_constant<int, I>{});                                                                           \
        wait_vmcnt<5>();      /* everything but this step's own five requests has landed */                                \
        loop_barrier();                                                                                                   \
    }
    for (int t = 0; t < NK; t += 6) {
        X3P_STEP(0) X3P_STEP(1) X3P_STEP(2) X3P_STEP(3) X3P_STEP(4) X3P_STEP(5)
    }
#undef X3P_STEP
    wait_vmcnt<0>();                              // (LDS-DMA still in flight must not outlive the workgroup)

    float* out = p.out + (size_t)b * p.batch_out;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
                float v = ((lh ^ i) & 1) ? -acc[i][j][r] : acc[i][j][r];
                if (m < p.M && n < p.N) {
                    if constexpr (!EPI) {
                        out[(size_t)m * p.N + n] = v;
                    } else {
                        const size_t idx = (size_t)m * p.ldo + n;
                        if (p.bias != nullptr) v += p.bias[n];
                        if (p.accumulate) v += out[idx];
                        if (p.relu) v = v < 0.f ? 0.f : v;
                        if (p.mask != nullptr) v = p.mask[idx] > 0.f ? v : 0.f;
                        out[idx] = v;
                    }
                }
            }
        }
}
// Measured against the two-stage kernel on one device (tools/x3_variants.sh, Gaussian operands): conv4_2 0.371 vs 0.365 ms, fc6 0.450 vs
// 0.449, conv3_2 0.380 vs 0.379 -- nothing where the time is -- and conv5_2 0.098 vs 0.107, c_4 0.101 vs 0.112 on the short grids.  With
// s_setprio around the MFMAs, a third weight stage and asm fragment reads also at +-0, and the all-zero-operand skeleton 27 % faster than
// real data at the same instruction stream, the long GEMMs sit at the rate the chip sustains on this data (about 1.0-1.06 PFLOP/s of bf16
// MFMAs), not at a limit of the loop structure: the simpler kernel stays.
#endif

// ---- TN form: out[b][split][m][n] = sum_k a[b][k][m] * c[b][k][n]  (the Winograd weight gradient: a = transformed dy planes [tiles][ldy],
// c = transformed input planes [tiles][Ci], k = tiles; split-K over `ksplit` slices of the tile range) ------------------------------------
// Both operands are activations, both are split in the kernel: per 16-tile step a thread loads 8 consecutive channels of one tile row of
// each operand (2 x 2 16-byte loads, 512-byte rows: coalesced), two steps ahead into one of two register sets, splits them in the
// shadow of the step's 24 MFMAs and stores six 16-byte limb rows.  LDS images are [16 tiles][128 channels] bf16 per limb, plain 256-byte
// rows with the 16-byte chunk ch of row r at ch ^ (((r & 3) << 2) | ((r >> 2) & 3)) (cdna_hip_programming.md T10, image (b): conflict-
// free for the transposing read); the MFMA operands -- 8 consecutive TILES of one channel per lane -- come from ds_read_b64_tr_b16.
struct TnX3Params {
    const float* a; const float* c; float* out;
    int M, N, K, lda, ldc, tiles_m, tiles_n, ksplit, ksteps_per_split, groups;      // K steps of 16; groups = problems x ksplit
    size_t batch_a, batch_c;
    unsigned a_bytes, c_bytes;
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// (compiler-visible reads here: as inline asm the two halves of a fragment land in separate register pairs and the copies that join them
// cost 40 registers -- 208, two workgroups per CU instead of three, 2.53 -> 2.8 ms per step)
__device__ __forceinline__ bf16x8 tr_pair(unsigned a0, unsigned a1) {
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds_u8*)(size_t)a0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds_u8*)(size_t)a1);
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

__global__ __launch_bounds__(256, 2) void gemm_tn_x3_kernel(const TnX3Params p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * X3_STAGE];       // stage = A limbs (3 x 4 KB) + C limbs (3 x 4 KB)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;
    // every XCD walks a contiguous range of (group, tile) ids: the output tiles of one (problem, K slice), which share the operand
    // panels, run on one XCD next to each other in time
    const int nblk = p.tiles_m * p.tiles_n;
    const int sid = xcd_swizzle(blockIdx.x, nblk * p.groups);
    const int grp = sid / nblk, lid = sid - grp * nblk;
    const int m0 = (lid / p.tiles_n) * 128, n0 = (lid % p.tiles_n) * 128;
    const int by = grp % p.ksplit, bz = grp / p.ksplit;
    const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a + (size_t)bz * p.batch_a), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.c + (size_t)bz * p.batch_c), 0, (int)p.c_bytes, 0x00020000);
    const int ksteps = (p.K + 15) >> 4;
    const int kt_begin = by * p.ksteps_per_split;
    const int KT = min(ksteps - kt_begin, p.ksteps_per_split);                      // (an odd count runs one more step on zero rows)

    // loads / LDS stores: thread -> (tile row t of the step, 16-byte chunk ch = 8 channels)
    const int trow = tid >> 4, ch = tid & 15;
    const bool ok_a = m0 + ch * 8 < p.lda, ok_c = n0 + ch * 8 < p.ldc;           // lda, ldc are multiples of 4: a chunk may be half valid
    const bool ok_a2 = m0 + ch * 8 + 4 < p.lda, ok_c2 = n0 + ch * 8 + 4 < p.ldc;
    const unsigned wr = 256 * trow + 16 * (ch ^ (((trow & 3) << 2) | ((trow >> 2) & 3)));
    // sign dither (see gemm_planes_x3_kernel): channel m of the a operand enters with s(m) = (-1)^(bit 3 ^ bit 6 of m) -- constant over a
    // thread's 8 channels -- and row m of the result is multiplied by s(m) again: the truncation bias of the bf16 MFMA chain, always
    // downwards, becomes zero-mean over the rows
    const unsigned a_sign = (unsigned)((ch ^ (ch >> 3)) & 1) << 31;
    f32x4 ra[2][4];                                                                 // [register set][a lo, a hi, c lo, c hi]
    auto load = [&](int kt, int set) {
        const size_t k = (size_t)(kt_begin + kt) * 16 + trow;
        const bool in = kt < KT && k < (size_t)p.K;
        const unsigned va = (unsigned)((k * p.lda + m0 + ch * 8) * 4), vc = (unsigned)((k * p.ldc + n0 + ch * 8) * 4);
        ra[set][0] = buf_load16(srd_a, in && ok_a ? va : OOB, 0);
        ra[set][1] = buf_load16(srd_a, in && ok_a2 ? va + 16u : OOB, 0);
        ra[set][2] = buf_load16(srd_c, in && ok_c ? vc : OOB, 0);
        ra[set][3] = buf_load16(srd_c, in && ok_c2 ? vc + 16u : OOB, 0);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposed fragment reads: lane 4q+p of a 16-lane group supplies row r0 + q, chunk c0 + (p >> 1), half p & 1 of its 4 x 16 block;
    // groups 0 / 1 of a half-wave take channels 0..15 / 16..31 of the 32-wide MFMA tile, the upper half-wave the tiles k = 8..15
    unsigned rd[2][2][2];                                                           // [operand][MFMA tile i][rows +0 / +4]
    {
        const int q = (lane & 15) >> 2, pp = lane & 3, g16 = (lane >> 4) & 1;
#pragma unroll
        for (int op = 0; op < 2; ++op)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = 8 * lh + 4 * h + q;
                    const int c = (op == 0 ? wm : wn) * 8 + i * 4 + g16 * 2 + (pp >> 1);
                    rd[op][i][h] = op * X3_OPER + 256 * row + 16 * (c ^ (((row & 3) << 2) | ((row >> 2) & 3))) + 8 * (pp & 1);
                }
    }

    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    auto step = [&](auto ph_tag) {
        constexpr int PH = decltype(ph_tag)::value;
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
        const unsigned st = lds_base + PH * X3_STAGE;
        bf16x8 af[3][2], bf[3][2];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int pa = g == 0 ? 2 : (g == 1 ? 0 : 1), pb = g == 0 ? 0 : (g == 1 ? 2 : 1);
                af[pa][i] = tr_pair(st + rd[0][i][0] + pa * X3_LIMB, st + rd[0][i][1] + pa * X3_LIMB);
                bf[pb][i] = tr_pair(st + rd[1][i][0] + pb * X3_LIMB, st + rd[1][i][1] + pb * X3_LIMB);
            }
        __builtin_amdgcn_sched_barrier(0);
        float xa[8], xb[8];                                                          // eight pairs: a chunk (4), c chunk (4)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const unsigned sg = v < 2 ? a_sign : 0u;                                 // the sign dither of the a operand's channel group
            xa[2 * v] = __uint_as_float(__float_as_uint(ra[PH ^ 1][v][0]) ^ sg); xb[2 * v] = __uint_as_float(__float_as_uint(ra[PH ^ 1][v][1]) ^ sg);
            xa[2 * v + 1] = __uint_as_float(__float_as_uint(ra[PH ^ 1][v][2]) ^ sg); xb[2 * v + 1] = __uint_as_float(__float_as_uint(ra[PH ^ 1][v][3]) ^ sg);
        }
        unsigned hi[8], mid[8], lo[8];
        float r1a[8], r1b[8];
        auto stage = [&](int sidx) {                                                  // 40 stages: pair e = sidx / 5, stage sidx % 5
            const int e = sidx / 5, sg = sidx % 5;
            if (sg == 0) { hi[e] = pack_rne(xa[e], xb[e]); asm volatile("" : "+v"(hi[e])); }
            if (sg == 1) {
                r1a[e] = xa[e] - __uint_as_float(hi[e] << 16); r1b[e] = xb[e] - __uint_as_float(hi[e] & 0xffff0000u);
                asm volatile("" : "+v"(r1a[e]), "+v"(r1b[e]));
            }
            if (sg == 2) { mid[e] = pack_rne(r1a[e], r1b[e]); asm volatile("" : "+v"(mid[e])); }
            if (sg == 3) {
                r1a[e] -= __uint_as_float(mid[e] << 16); r1b[e] -= __uint_as_float(mid[e] & 0xffff0000u);
                asm volatile("" : "+v"(r1a[e]), "+v"(r1b[e]));
            }
            if (sg == 4) { lo[e] = pack_rne(r1a[e], r1b[e]); asm volatile("" : "+v"(lo[e])); }
        };
#pragma unroll
        for (int q = 0; q < 24; ++q) {
            const int pr = q >> 2, i = (q >> 1) & 1, j = q & 1;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[pr]][i], bf[PB[pr]][j], acc[i][j], 0, 0, 0);
            if (q >= 2 && q < 20) { stage(2 * (q - 2)); stage(2 * (q - 2) + 1); }     // 36 stages, two per MFMA ...
            if (q >= 20) stage(36 + (q - 20));                                        // ... and the last four
            __builtin_amdgcn_sched_barrier(0);
        }
        unsigned char* d = lds + (PH ^ 1) * X3_STAGE + wr;
        *reinterpret_cast<u32x4*>(d) = u32x4{hi[0], hi[1], hi[2], hi[3]};
        *reinterpret_cast<u32x4*>(d + X3_LIMB) = u32x4{mid[0], mid[1], mid[2], mid[3]};
        *reinterpret_cast<u32x4*>(d + 2 * X3_LIMB) = u32x4{lo[0], lo[1], lo[2], lo[3]};
        *reinterpret_cast<u32x4*>(d + X3_OPER) = u32x4{hi[4], hi[5], hi[6], hi[7]};
        *reinterpret_cast<u32x4*>(d + X3_OPER + X3_LIMB) = u32x4{mid[4], mid[5], mid[6], mid[7]};
        *reinterpret_cast<u32x4*>(d + X3_OPER + 2 * X3_LIMB) = u32x4{lo[4], lo[5], lo[6], lo[7]};
    };

    load(0, 0);
    load(1, 1);
    {
        u32x4 h, m, l;
        unsigned char* d = lds + wr;
        f32x4 s0 = ra[0][0], s1 = ra[0][1];           // (the compiler waits for set 0 here)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s0[e] = __uint_as_float(__float_as_uint(s0[e]) ^ a_sign);
            s1[e] = __uint_as_float(__float_as_uint(s1[e]) ^ a_sign);
        }
        split8(s0, s1, h, m, l);
        *reinterpret_cast<u32x4*>(d) = h; *reinterpret_cast<u32x4*>(d + X3_LIMB) = m; *reinterpret_cast<u32x4*>(d + 2 * X3_LIMB) = l;
        split8(ra[0][2], ra[0][3], h, m, l);
        *reinterpret_cast<u32x4*>(d + X3_OPER) = h; *reinterpret_cast<u32x4*>(d + X3_OPER + X3_LIMB) = m;
        *reinterpret_cast<u32x4*>(d + X3_OPER + 2 * X3_LIMB) = l;
    }
    load(2, 0);
    __syncthreads();
    for (int kt = 0; kt < KT; kt += 2) {
        step(std::integral_constant<int, 0>{});       // step kt: reads stage 0, splits set 1 (step kt + 1) into stage 1
        __syncthreads();
        load(kt + 3, 1);
        step(std::integral_constant<int, 1>{});       // step kt + 1
        __syncthreads();
        load(kt + 4, 0);
    }

    float* out = p.out + ((size_t)bz * p.ksplit + by) * p.M * p.N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);       // bits 3 and 6 of m: bit 0 of r >> 2, and wm
                const float v = (((r >> 2) ^ wm) & 1) ? -acc[i][j][r] : acc[i][j][r];
                if (m < p.M && n < p.N) out[(size_t)m * p.N + n] = v;
            }
        }
}

// w [nbatch][rows][K] f32 -> w3 [nbatch][K/16][3][rows_pad][16] bf16 limbs (rows beyond `rows`: zero).  One thread per (b, k step, row).
__global__ __launch_bounds__(256) void split_weights_x3_kernel(const float* __restrict__ w, __bf16* __restrict__ w3, int rows, int rows_pad, int K,
                                                                 int nbatch) {
    const int NK = K >> 4;
    const size_t total = (size_t)nbatch * NK * rows_pad;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int row = (int)(e % rows_pad);
        const size_t t = e / rows_pad;
        const int ks = (int)(t % NK), b = (int)(t / NK);
        f32x4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < rows) v[q] = *reinterpret_cast<const f32x4*>(w + ((size_t)b * rows + row) * K + ks * 16 + q * 4);
        }
        __bf16* d = w3 + (((size_t)b * NK + ks) * 3 * rows_pad + row) * 16;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32x4 hi, mid, lo;
            split8(v[2 * h], v[2 * h + 1], hi, mid, lo);
            *reinterpret_cast<u32x4*>(d + h * 8) = hi;
            *reinterpret_cast<u32x4*>(d + (size_t)rows_pad * 16 + h * 8) = mid;
            *reinterpret_cast<u32x4*>(d + (size_t)rows_pad * 32 + h * 8) = lo;
        }
    }
}

}  // namespace

extern "C" size_t ssd_gemm_x3_weights_bytes(int rows, int K, int nbatch) {
    if (rows <= 0 || K <= 0 || K % 32 != 0 || nbatch <= 0) return 0;
    return (size_t)nbatch * K * ssd_cdiv(rows, 128) * 128 * 3 * 2;
}

extern "C" int ssd_gemm_x3_split_weights(const float* w, void* w3, int rows, int K, int nbatch, void* stream) {
    if (!w || !w3) return SSD_ERR_NULL;
    if (rows <= 0 || K <= 0 || K % 32 != 0 || nbatch <= 0) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(w) || !ssd_aligned16(w3)) return SSD_ERR_ALIGN;
    const int rows_pad = ssd_cdiv(rows, 128) * 128;
    const size_t total = (size_t)nbatch * (K / 16) * rows_pad, blocks = (total + 255) / 256;
    hipLaunchKernelGGL(split_weights_x3_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, (hipStream_t)stream, w,
                       static_cast<__bf16*>(w3), rows, rows_pad, K, nbatch);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

int ssd_internal_gemm_batched_x3s(const float* a, const void* w3, float* out, int M, int K, int N, int n_rows, int nbatch, size_t batch_a_elems,
                                  hipStream_t st);       // csrc/gemm_x3v2.hip: 256 x 256 tiles, two wave groups in ping-pong
static int g_x3_big = 0;       // ssd_tune_set_x3_big (SSD_EXPERIMENTAL builds): 0 = every launch on the 128 x 128 kernel (default), 1 = the large launches on the 256 x 256 ping-pong kernel, 2 = all that fit
static int g_x3_m16 = 0;      // ssd_tune_set_x3_mfma: 1 = the plane GEMMs on v_mfma_f32_16x16x32_bf16 (limb pairs concatenated along K), 0 = 32x32x16
// Internal (not part of the C ABI): the x3 form of ssd_internal_gemm_batched (conv_igemm.hip); w3 from ssd_gemm_x3_split_weights
__attribute__((visibility("hidden"))) int ssd_internal_gemm_batched_x3(const float* a, const void* w3, float* out, int M, int K, int N, int n_rows,
                                                                        int nbatch, size_t batch_a_elems, hipStream_t st) {
    if (K % 32 != 0 || M <= 0 || N <= 0 || n_rows <= 0 || ssd_cdiv(n_rows, 128) * 128 < N || nbatch <= 0) return SSD_ERR_BAD_SHAPE;
    if ((size_t)M * K * 4 >= 0xF0000000ull) return SSD_ERR_BAD_SHAPE;
    GemmX3Params p{};
    p.a = a; p.w3 = static_cast<const __bf16*>(w3); p.out = out;
    p.M = M; p.K = K; p.N = N; p.rows_pad = ssd_cdiv(n_rows, 128) * 128;
    p.tiles_m = ssd_cdiv(M, 128); p.tiles_n = ssd_cdiv(N, 128); p.nbatch = nbatch;
    p.batch_a = batch_a_elems; p.batch_out = (size_t)M * N;
    const size_t nblk = (size_t)p.tiles_m * p.tiles_n * nbatch;
    if (nblk >= (1ull << 31)) return SSD_ERR_BAD_SHAPE;
    // The large launches (conv3_2 ... conv4_3, fc6 and their data gradients: N >= 256 columns, >= 8 row tiles) take the 256 x 256 ping-pong kernel
    // (csrc/gemm_x3v2.hip); short grids and narrow N keep the 128 x 128 tiles, which quantise better (measured: tools/gemm_x3_bench.py).
    if (K >= 64 && N % 4 == 0 && ((g_x3_big == 1 && N >= 256 && M >= 2048) || g_x3_big == 2)) {
        const size_t nblk2 = (size_t)ssd_cdiv(M, 256) * ssd_cdiv(N, 256) * nbatch;
        const int slot2 = ssd_internal_prof_open(6.0 * 2.0 * (double)nblk2 * 256.0 * 256.0 * K, 4, st);
        const int e = ssd_internal_gemm_batched_x3s(a, w3, out, M, K, N, n_rows, nbatch, batch_a_elems, st);
        ssd_internal_prof_close(slot2, st);
        return e;
    }
    // recorder kind 4: the bf16 MFMA FLOPs the grid executes (six limb products per f32 product, whole 128 x 128 tiles)
    const int slot = ssd_internal_prof_open(6.0 * 2.0 * (double)nblk * 128.0 * 128.0 * K, 4, st);
#ifdef X3_PIPE
    {
        static std::atomic<unsigned long long> raised{0};
        int dev;
        if (ssd_attr_needed(raised, dev)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_planes_x3p_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 0);
            ssd_attr_done(raised, dev);
        }
    }
    hipLaunchKernelGGL(gemm_planes_x3p_kernel<false>, dim3((unsigned)nblk), dim3(256), 0, st, p);
#else
#ifdef SSD_EXPERIMENTAL
    if (g_x3_m16) hipLaunchKernelGGL((gemm_planes_x3_kernel<false, true>), dim3((unsigned)nblk), dim3(256), 0, st, p);
    else
#endif
    hipLaunchKernelGGL((gemm_planes_x3_kernel<false, false>), dim3((unsigned)nblk), dim3(256), 0, st, p);
#endif
    ssd_internal_prof_close(slot, st);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

// Tuning aid (SSD_EXPERIMENTAL builds; measured no better in the step, csrc/gemm_x3v2.hip): 0 (default) = every limb plane GEMM on the 128 x 128
// kernel, 1 = the large launches on the 256 x 256 ping-pong kernel, 2 = every launch it takes
extern "C" int ssd_tune_set_x3_big(int mode) {
    if (mode < 0 || mode > 2) return SSD_ERR_BAD_SHAPE;
#ifndef SSD_EXPERIMENTAL
    if (mode != 0) return SSD_ERR_BAD_SHAPE;        // csrc/gemm_x3v2.hip is not in this build
#endif
    g_x3_big = mode;
    return SSD_OK;
}

// Tuning aid: MFMA shape of the limb plane GEMMs (forward / data gradient of the Winograd layers): 32 = v_mfma_f32_32x32x16_bf16 (default),
// 16 = v_mfma_f32_16x16x32_bf16 with two limb products per instruction.
extern "C" int ssd_tune_set_x3_mfma(int rows) {
    if (rows != 16 && rows != 32) return SSD_ERR_BAD_SHAPE;
#ifndef SSD_EXPERIMENTAL
    if (rows == 16) return SSD_ERR_BAD_SHAPE;       // measured within noise of the 32x32x16 form: SSD_EXPERIMENTAL builds only
#endif
    g_x3_m16 = rows == 16;
    return SSD_OK;
}

// Internal: TN split-K plane GEMMs from limbs; out[b][split][M][N] raw partial sums, K steps of 16 rows, `ksplit` slices of
// `ksteps_per_split` steps each (ksplit * ksteps_per_split >= ceil(K / 16))
__attribute__((visibility("hidden"))) int ssd_internal_gemm_tn_x3(const float* a, const float* c, float* out, int M, int N, int K, int lda, int ldc,
                                                                   int nbatch, int ksplit, int ksteps_per_split, size_t batch_a, size_t batch_c,
                                                                   hipStream_t st) {
    if (M <= 0 || N <= 0 || K <= 0 || lda % 4 != 0 || ldc % 4 != 0 || nbatch <= 0 || ksplit < 1 || ksteps_per_split < 1) return SSD_ERR_BAD_SHAPE;
    if ((size_t)ksplit * ksteps_per_split * 16 < (size_t)K) return SSD_ERR_BAD_SHAPE;
    if (batch_a * 4 >= 0xF0000000ull || batch_c * 4 >= 0xF0000000ull) return SSD_ERR_BAD_SHAPE;
    TnX3Params q{};
    q.a = a; q.c = c; q.out = out;
    q.M = M; q.N = N; q.K = K; q.lda = lda; q.ldc = ldc;
    q.tiles_m = ssd_cdiv(M, 128); q.tiles_n = ssd_cdiv(N, 128);
    q.ksplit = ksplit; q.ksteps_per_split = ksteps_per_split; q.groups = nbatch * ksplit;
    q.batch_a = batch_a; q.batch_c = batch_c;
    q.a_bytes = (unsigned)((size_t)K * lda * 4); q.c_bytes = (unsigned)((size_t)K * ldc * 4);
    const size_t nblk = (size_t)q.tiles_m * q.tiles_n * q.groups;
    if (nblk >= (1ull << 31)) return SSD_ERR_BAD_SHAPE;
    // recorder kind 5: executed bf16 MFMA FLOPs (whole tiles, whole 16-row steps of every K slice)
    const int slot = ssd_internal_prof_open(6.0 * 2.0 * (double)nblk * 128.0 * 128.0 * 16.0 * ksteps_per_split, 5, st);
    hipLaunchKernelGGL(gemm_tn_x3_kernel, dim3((unsigned)nblk), dim3(256), 0, st, q);
    ssd_internal_prof_close(slot, st);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

// ---- 1x1 / stride-1 convolutions with long reductions (fc7, seq8.0: Model.py:150-156) on the same kernels -----------------------------------
namespace {
// out[i] = sum_k slab[k][i], four entries per thread, slices in index order (reproducible)
__global__ __launch_bounds__(256) void x3_slab_sum_kernel(const float* __restrict__ slab, float* __restrict__ out, size_t n4, int nslab) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 sum = *reinterpret_cast<const f32x4*>(slab + i * 4);
        for (int k = 1; k < nslab; ++k) sum += *reinterpret_cast<const f32x4*>(slab + (size_t)k * n4 * 4 + i * 4);
        *reinterpret_cast<f32x4*>(out + i * 4) = sum;
    }
}
// column sums of dy [M][ld] (the bias gradient): block b sums rows b, b + gridDim.x, ... of a column quad per thread -> part[b][ld]
__global__ __launch_bounds__(256) void x3_colsum_partial_kernel(const float* __restrict__ dy, float* __restrict__ part, size_t M, int ld) {
    const int q4 = ld >> 2;
    for (int q = threadIdx.x; q < q4; q += 256) {
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f};
        for (size_t m = blockIdx.x; m < M; m += gridDim.x) s0 += *reinterpret_cast<const f32x4*>(dy + m * ld + q * 4);
        *reinterpret_cast<f32x4*>(part + (size_t)blockIdx.x * ld + q * 4) = s0;
    }
}
// 64 channels per block; thread (channel, quarter q) adds the partial rows q, q + 4, ... and the four quarters are added in index order:
// a fixed order (reproducible), 64 dependent adds per thread instead of 256 and 4x the blocks (was 60 us for fc7's 1024 channels)
__global__ __launch_bounds__(256) void x3_colsum_final_kernel(const float* __restrict__ part, float* __restrict__ db, int nblk, int C, int ld) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float t = 0.f;
    if (c < C)
        for (int k = q; k < nblk; k += 4) t += part[(size_t)k * ld + c];
    red[q][cl] = t;
    __syncthreads();
    if (q == 0 && c < C) db[c] = ((red[0][cl] + red[1][cl]) + red[2][cl]) + red[3][cl];
}
constexpr int X3_COLSUM_BLOCKS = 256;
struct X3WgradPlan { int ks, per; size_t slab_floats, part_floats; };
X3WgradPlan x3_wgrad_plan(const ssd_conv_geom* g, int ldy) {
    X3WgradPlan w;
    const size_t M = (size_t)g->N * g->Ho * g->Wo;
    const int tiles = ssd_cdiv(g->Co, 128) * ssd_cdiv(g->Ci, 128), steps16 = (int)((M + 15) / 16);
    int ks = 768 / tiles;                          // one round of resident workgroups
    if (ks > 1) ks &= ~1;
    if (ks > steps16 / 8) ks = steps16 / 8;
    if (ks < 1) ks = 1;
    w.per = ssd_cdiv(steps16, ks);
    w.ks = ssd_cdiv(steps16, w.per);
    w.slab_floats = (size_t)w.ks * g->Co * g->Ci;
    w.part_floats = (size_t)X3_COLSUM_BLOCKS * ldy;
    return w;
}
bool x3_1x1_geom_ok(const ssd_conv_geom* g) {
    return g && g->R == 1 && g->S == 1 && g->stride == 1 && g->pad == 0 && g->Ho == g->H && g->Wo == g->W && g->N > 0 && g->H > 0 && g->W > 0 &&
           g->Ci > 0 && g->Co > 0;
}
}  // namespace

extern "C" int ssd_conv1x1_fwd_x3(const float* x, const void* w3, const float* bias, float* y, int ldy, const ssd_conv_geom* g, int relu, void* stream) {
    if (!x || !w3 || !y) return SSD_ERR_NULL;
    if (!x3_1x1_geom_ok(g) || g->Ci % 32 != 0 || ldy < g->Co) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(x) || !ssd_aligned16(w3) || !ssd_aligned16(y)) return SSD_ERR_ALIGN;
    const size_t M = (size_t)g->N * g->H * g->W;
    if (M >= (1ull << 31) || M * g->Ci * 4 >= 0xF0000000ull) return SSD_ERR_BAD_SHAPE;
    GemmX3Params p{};
    p.a = x; p.w3 = static_cast<const __bf16*>(w3); p.out = y;
    p.M = (int)M; p.K = g->Ci; p.N = g->Co; p.rows_pad = ssd_cdiv(g->Co, 128) * 128;
    p.tiles_m = ssd_cdiv((int)M, 128); p.tiles_n = ssd_cdiv(g->Co, 128); p.nbatch = 1;
    p.bias = bias; p.mask = nullptr; p.ldo = ldy; p.relu = relu; p.accumulate = 0;
    hipLaunchKernelGGL(gemm_planes_x3_kernel<true>, dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(256), 0, (hipStream_t)stream, p);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

// dx [M][Ci] = dy [M][ldy] * w (ldy = Co_pad = the reduction length, a multiple of 32; w3t = limbs of w^T: rows Ci, K = Co_pad)
extern "C" int ssd_conv1x1_dgrad_x3(const float* dy, int ldy, const void* w3t, float* dx, const float* relu_mask, int accumulate,
                                    const ssd_conv_geom* g, void* stream) {
    if (!dy || !w3t || !dx) return SSD_ERR_NULL;
    if (!x3_1x1_geom_ok(g) || ldy % 32 != 0 || ldy < g->Co) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(dy) || !ssd_aligned16(w3t) || !ssd_aligned16(dx) || (relu_mask && !ssd_aligned16(relu_mask))) return SSD_ERR_ALIGN;
    const size_t M = (size_t)g->N * g->H * g->W;
    if (M >= (1ull << 31) || M * ldy * 4 >= 0xF0000000ull) return SSD_ERR_BAD_SHAPE;
    GemmX3Params p{};
    p.a = dy; p.w3 = static_cast<const __bf16*>(w3t); p.out = dx;
    p.M = (int)M; p.K = ldy; p.N = g->Ci; p.rows_pad = ssd_cdiv(g->Ci, 128) * 128;
    p.tiles_m = ssd_cdiv((int)M, 128); p.tiles_n = ssd_cdiv(g->Ci, 128); p.nbatch = 1;
    p.bias = nullptr; p.mask = relu_mask; p.ldo = g->Ci; p.relu = 0; p.accumulate = accumulate;
    hipLaunchKernelGGL(gemm_planes_x3_kernel<true>, dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(256), 0, (hipStream_t)stream, p);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

extern "C" size_t ssd_conv1x1_wgrad_x3_workspace(const ssd_conv_geom* g, int ldy) {
    if (!x3_1x1_geom_ok(g) || ldy < g->Co || ldy % 4 != 0) return 0;
    const X3WgradPlan w = x3_wgrad_plan(g, ldy);
    return (w.slab_floats + w.part_floats) * 4 + 256;
}

// dw [Co][Ci] = dy^T x over the pixels (split-K TN GEMM from limbs + slice sum), dbias = column sums of dy (optional)
extern "C" int ssd_conv1x1_wgrad_x3(const float* x, const float* dy, int ldy, float* dw, float* dbias, const ssd_conv_geom* g, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    if (!x || !dy || !dw || !workspace) return SSD_ERR_NULL;
    if (!x3_1x1_geom_ok(g) || g->Ci % 4 != 0 || ldy % 4 != 0 || ldy < g->Co) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(x) || !ssd_aligned16(dy) || !ssd_aligned16(dw) || !ssd_aligned16(workspace)) return SSD_ERR_ALIGN;
    if (workspace_bytes < ssd_conv1x1_wgrad_x3_workspace(g, ldy)) return SSD_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const X3WgradPlan w = x3_wgrad_plan(g, ldy);
    const size_t M = (size_t)g->N * g->H * g->W;
    if (M >= (1ull << 31)) return SSD_ERR_BAD_SHAPE;
    float* slab = static_cast<float*>(workspace);
    float* part = slab + w.slab_floats;
    float* dst = w.ks == 1 && (g->Co * (size_t)g->Ci) % 4 == 0 ? dw : slab;
    if (int e = ssd_internal_gemm_tn_x3(dy, x, dst, g->Co, g->Ci, (int)M, ldy, g->Ci, 1, w.ks, w.per, M * ldy, M * g->Ci, st)) return e;
    if (dst != dw) {
        const size_t n = (size_t)g->Co * g->Ci;
        if (n % 4 != 0) return SSD_ERR_BAD_SHAPE;
        const size_t blocks = (n / 4 + 255) / 256;
        hipLaunchKernelGGL(x3_slab_sum_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, st, slab, dw, n / 4, w.ks);
        SSD_CHECK_LAUNCH();
    }
    if (dbias != nullptr) {
        hipLaunchKernelGGL(x3_colsum_partial_kernel, dim3(X3_COLSUM_BLOCKS), dim3(256), 0, st, dy, part, M, ldy);
        hipLaunchKernelGGL(x3_colsum_final_kernel, dim3(ssd_cdiv(g->Co, 64)), dim3(256), 0, st, part, dbias, X3_COLSUM_BLOCKS, g->Co, ldy);
        SSD_CHECK_LAUNCH();
    }
    return SSD_OK;
}

extern "C" int ssd_gemm_planes_x3(const float* a, const void* w3, float* out, int M, int K, int N, int n_rows, int nbatch, void* stream) {
    if (!a || !w3 || !out) return SSD_ERR_NULL;
    if (!ssd_aligned16(a) || !ssd_aligned16(w3) || !ssd_aligned16(out)) return SSD_ERR_ALIGN;
    return ssd_internal_gemm_batched_x3(a, w3, out, M, K, N, n_rows, nbatch, (size_t)M * K, (hipStream_t)stream);
}
