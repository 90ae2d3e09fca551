// Plane GEMMs from PRE-SPLIT limbs on both sides (round 4 experiment; reached through ssd_gemm_planes_x3v2 only -- not on the step's path yet):
//
//     out[b][m][n] = sum_k a[b][m][k] * w[b][n][k],   a and w both given as three exact bf16 limbs per f32 value in the pre-tiled layout
//     [b][K/16][3][rows_pad128][16] of ssd_gemm_x3_split_weights (csrc/gemm_x3.hip): six limb products per f32 product, f32 accumulate.
//
// What it tests: the 128 x 128 kernel of gemm_x3.hip sits at 0.47 MFMA busy -- its three small workgroups per CU meet at their barriers
// together, and a wave splits its activation rows in registers.  Here
//   * 256 x 256 tile, 512 threads = 8 waves (2 x 4: wave tile 128 x 64 = 4 x 2 MFMA tiles of v_mfma_f32_32x32x16_bf16), ONE workgroup per CU;
//   * both operands arrive by LDS-DMA (global_load_lds_dwordx4), no VALU in the loop: a K step of 16 is 48 KB (two operands x three limbs
//     x 256 rows x 32 bytes), three slots = 144 KB of LDS, six 1-KB DMA instructions per wave and step;
//   * PING-PONG (cdna_hip_programming.md section 5, the 8-phase template's stagger; MI355X_MICROARCH.md "Two waves per SIMD"): waves 0-3 and
//     waves 4-7 -- one of each per SIMD -- run half a step apart.  A step is a READ phase (wait for the DMA issued a step ago, 18
//     ds_read_b128 of this step's fragments, issue the DMA of step + 2) and a MATRIX phase (48 MFMAs, 1536 cycles), a barrier after each;
//     one group's read phase runs beside the other group's matrix phase, so each SIMD's matrix pipe always has one wave issuing.
// Slot safety: the DMA of step t + 2 overwrites the slot of step t - 1, whose last reads (the other group's) finished before the barrier
// that precedes this read phase; data of step t is waited for (vmcnt(0) at the top of read phase t - 1, a full step after its issue) by
// every wave and published by the barriers in between (rule "read a staged buffer one phase after the wait that retires it").
//
// MEASURED (round 4, tools/gemm_x3_bench.py, interleaved rounds on one device; ms per launch of 36 planes):
//                    128 x 128 kernel (gemm_x3.hip)   pre-split ping-pong (v2)   in-kernel split ping-pong (x3s)
//   conv3_2 (11552 x 256 x 256)      0.343                     0.320                       0.363
//   conv4_2 (3200 x 512 x 512)       0.341                     0.307   (1 184 TFLOP/s)     0.361
//   fc6     (2048 x 512 x 1024)      0.417                     0.375                       0.435
//   conv5_2 (800 x 512 x 512)        0.102                     0.125                       0.152
//   dgrad conv3_1 (N = 128)          0.194                     0.284                       0.339
// The structure is worth 10-13 % on the large launches ONLY when the activation limbs are already in memory; splitting in the kernel (in the
// MFMA gaps: 0.372; in the read phase beside a prio-1 matrix phase: 0.361) gives it back, and writing limb planes from the transform kernels
// costs 6 instead of 4 bytes per plane element (+0.4 ms per step) for -0.4 ms of GEMM time.  NOT in the default build (SSD_EXPERIMENTAL=1,
// like gemm_nt.hip); the 128 x 128 kernel stays on the path.  Step A/B with the x3s dispatch: 20.17 vs 20.18 ms.
#include "common.h"

#ifdef SSD_EXPERIMENTAL
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

struct X3V2Params {
    const __bf16* __restrict__ a3;      // [nbatch][K/16][3][rows_a][16]
    const __bf16* __restrict__ w3;      // [nbatch][K/16][3][rows_w][16]
    float* __restrict__ out;            // [nbatch][M][N]
    int M, K, N, rows_a, rows_w;
    int tiles_m, tiles_n, nbatch;
    int dither;                         // 1: the rows of a3 were stored with the sign s(m) = (-1)^(bit 2 ^ bit 5 of m) (gemm_x3.hip): undo it on the way out
};

constexpr int V2_LIMB = 256 * 32;              // bytes of one limb image: [256 rows][16 k] bf16
constexpr int V2_OPER = 3 * V2_LIMB;           // one operand, one step: 24 KB
constexpr int V2_STAGE = 2 * V2_OPER;          // A + B: 48 KB
constexpr int V2_SLOTS = 3;

__global__ __launch_bounds__(512, 1) void gemm_planes_x3v2_kernel(const X3V2Params p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];          // 144 KB (dynamic: above the 64 KB static limit)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                  // ping-pong group: waves 0-3 lead, waves 4-7 follow half a step behind
    const int w4 = wave & 3;
    const int wm = w4 >> 1, wn2 = (w4 & 1) * 2 + grp;      // wave tile: rows wm * 128 .. + 127, columns wn2 * 64 .. + 63 (each group covers all four column blocks over its two wm)
    const int lr = lane & 31, lh = lane >> 5;
    const int nblk = p.tiles_m * p.tiles_n * p.nbatch;
    int lid = xcd_swizzle(blockIdx.x, nblk);
    const int tile_n = lid % p.tiles_n;
    lid /= p.tiles_n;
    const int tile_m = lid % p.tiles_m, b = lid / p.tiles_m;
    const int m0 = tile_m * 256, n0 = tile_n * 256;
    const int NK = p.K >> 4;

    // ---- DMA pieces of this wave: 48 one-KB pieces per step (operand, limb, 32-row block), six per wave -----------------------------------
    // a piece = 32 rows x 32 bytes of one limb image; lane = (row of the piece, 16-byte half); the image's XOR (half ^ bit 3 of the row) is applied
    // on the source address, the LDS destination is linear (rule 21)
    const int prow = lane >> 1, phalf = lane & 1;
    const __bf16* src[6];
    int dst[6];
    size_t kstride[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int q = wave * 6 + i;                 // 0..47
        const int oper = q / 24, r = q % 24, limb = r / 8, blk = r % 8;
        const int row = blk * 32 + prow;
        const int half = phalf ^ ((row >> 3) & 1);
        const int rows = oper == 0 ? p.rows_a : p.rows_w;
        int grow = (oper == 0 ? m0 : n0) + row;
        grow = grow < rows ? grow : rows - 1;       // beyond the operand: any valid row (its products are never stored)
        const __bf16* base = (oper == 0 ? p.a3 : p.w3) + (size_t)b * NK * 3 * rows * 16;
        src[i] = base + ((size_t)limb * rows + grow) * 16 + half * 8;
        kstride[i] = (size_t)3 * rows * 16;
        dst[i] = oper * V2_OPER + limb * V2_LIMB + blk * 1024;
    }
    auto issue = [&](int ks, int slot) {
        unsigned char* base = lds + slot * V2_STAGE;
        const int k = ks < NK ? ks : NK - 1;        // beyond the last step: re-fetch the last one (never read): every wait count stays a constant
#pragma unroll
        for (int i = 0; i < 6; ++i)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src[i] + (size_t)k * kstride[i]), (lds_void*)(base + dst[i]), 16, 0, 0);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const unsigned frag = lr * 32 + ((lh ^ ((lr >> 3) & 1)) << 4);
    const unsigned a_rd = wm * 128 * 32 + frag, b_rd = V2_OPER + wn2 * 64 * 32 + frag;

    // prologue: steps 0 and 1 requested; the follower group enters the loop one barrier late
    issue(0, 0);
    issue(1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");          // step 0 has landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();                             // ... everyone's
    if (grp == 1) __builtin_amdgcn_s_barrier();               // the stagger: from here on the follower is half a step behind

    for (int ks = 0; ks < NK; ++ks) {
        const int slot = ks % V2_SLOTS;
        // ---- read phase -----------------------------------------------------------------------------------------------------------------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the DMA issued a step ago (step ks + 1): published by the two barriers below
        const unsigned char* st = lds + slot * V2_STAGE;
        bf16x8 af[3][4], bf[3][2];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int i = 0; i < 4; ++i) af[pl][i] = *reinterpret_cast<const bf16x8*>(st + a_rd + pl * V2_LIMB + i * 1024);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[pl][j] = *reinterpret_cast<const bf16x8*>(st + b_rd + pl * V2_LIMB + j * 1024);
        }
        issue(ks + 2, (ks + 2) % V2_SLOTS);                   // into the slot of step ks - 1: both groups finished reading it before the barrier above this phase
        __builtin_amdgcn_s_waitcnt(0xC07F);                   // lgkmcnt(0): the fragments are in registers (the DMA stays in flight)
        __builtin_amdgcn_s_barrier();
        // ---- matrix phase ---------------------------------------------------------------------------------------------------------------
        __builtin_amdgcn_s_setprio(1);
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};      // smallest limb products first (gemm_x3.hip)
#pragma unroll
        for (int pr = 0; pr < 6; ++pr)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[pr]][i], bf[PB[pr]][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();               // the leader's matching extra barrier: both groups have executed the same number
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    float* out = p.out + (size_t)b * ((size_t)p.M * p.N);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn2 * 64 + j * 32 + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 128 + i * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);       // bits 2 and 5 of m: lh and i & 1
                float v = acc[i][j][r];
                if (p.dither && ((lh ^ i) & 1)) v = -v;
                if (m < p.M && n < p.N) out[(size_t)m * p.N + n] = v;
            }
        }
}


// ---- the same structure with the activation operand as it lies in memory today: f32 planes [M][K], split in the kernel ----------------------
// (no change to the transform kernels, the kept planes or the weight-gradient GEMMs.)  A thread owns 8 consecutive k of one row per step, as in
// gemm_x3.hip: the rows of step t + 3 are requested in read phase t (two 16-byte buffer loads), and in read phase t + 1 -- while the OTHER
// group's waves hold the matrix pipe -- they are split into limbs and stored to LDS slot (t + 3) % 3, two steps before either group reads them;
// the matrix phase is 48 MFMAs and nothing else.  (First version: split in the MFMA gaps of the wave's own matrix phase, pinned by
// sched_barrier as in gemm_x3.hip: 0.372 ms on conv4_2 against 0.338 for the 128 x 128 kernel; without the pins 0.337; this form: see DESIGN.)  The weights arrive by LDS-DMA as above (three pieces
// per wave and step).  The row's sign dither s(m) of gemm_x3.hip is applied before the split and undone in the epilogue.
// Epilogue: a wave's 128 x 64 result goes through LDS (the ring is free by then) in two halves and leaves as 16-byte stores, 256 contiguous
// bytes per row -- a quarter of the store instructions of the direct form, which at one workgroup per CU are not hidden behind anything.
constexpr unsigned V2_OOB = 0xFFFFFF00u;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct X3SParams {
    const float* __restrict__ a;        // [nbatch][M][K] f32
    const __bf16* __restrict__ w3;      // [nbatch][K/16][3][rows_w][16]
    float* __restrict__ out;            // [nbatch][M][N]
    int M, K, N, rows_w;
    int tiles_m, tiles_n, nbatch;
    size_t batch_a, batch_out;
};

__device__ __forceinline__ unsigned v2_pack(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const bf16x2 h = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, h);
}

__global__ __launch_bounds__(512, 1) void gemm_planes_x3s_kernel(const X3SParams p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];          // 144 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3;
    const int wm = w4 >> 1, wn2 = (w4 & 1) * 2 + grp;
    const int lr = lane & 31, lh = lane >> 5;
    const int nblk = p.tiles_m * p.tiles_n * p.nbatch;
    int lid = xcd_swizzle(blockIdx.x, nblk);
    const int tile_n = lid % p.tiles_n;
    lid /= p.tiles_n;
    const int tile_m = lid % p.tiles_m, b = lid / p.tiles_m;
    const int m0 = tile_m * 256, n0 = tile_n * 256;
    const int NK = p.K >> 4;

    // A: thread -> (row, half): 8 consecutive k of one row per step (buffer loads: rows beyond M read as zero, no traffic)
    const int arow = tid >> 1, ahalf = tid & 1;
    const __amdgpu_buffer_rsrc_t srd_a =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a + (size_t)b * p.batch_a), 0, (int)((size_t)p.M * p.K * 4), 0x00020000);
    const unsigned voff_a = m0 + arow < p.M ? ((unsigned)(m0 + arow) * (unsigned)p.K + ahalf * 8u) * 4u : V2_OOB;
    const unsigned a_wr = arow * 32 + ((ahalf ^ ((arow >> 3) & 1)) << 4);
    const unsigned a_sign = (unsigned)(((arow >> 2) ^ (arow >> 5)) & 1) << 31;
    f32x4 ra[2][2];
    auto load_a = [&](int ks, int set) {
        const unsigned v = ks < NK ? voff_a : V2_OOB;
        ra[set][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_a, (int)v, ks * 64, 0));
        ra[set][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_a, (int)(v + 16u), ks * 64, 0));
    };
    // B: 24 one-KB pieces per step (limb, 32-row block), three per wave
    const int prow = lane >> 1, phalf = lane & 1;
    const __bf16* bsrc[3];
    int bdst[3];
    const size_t bk = (size_t)3 * p.rows_w * 16;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int q = wave * 3 + i, limb = q / 8, blk = q % 8;
        const int row = blk * 32 + prow;
        const int half = phalf ^ ((row >> 3) & 1);
        int grow = n0 + row;
        grow = grow < p.rows_w ? grow : p.rows_w - 1;
        bsrc[i] = p.w3 + (size_t)b * NK * bk + ((size_t)limb * p.rows_w + grow) * 16 + half * 8;
        bdst[i] = V2_OPER + limb * V2_LIMB + blk * 1024;
    }
    auto issue_b = [&](int ks, int slot) {
        unsigned char* base = lds + slot * V2_STAGE;
        const int k = ks < NK ? ks : NK - 1;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(bsrc[i] + (size_t)k * bk), (lds_void*)(base + bdst[i]), 16, 0, 0);
    };
    unsigned hi[4], mid[4], lo[4];
    auto split_now = [&](int set) {               // prologue only: the loop splits in the MFMA gaps
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x4 v = ra[set][e >> 1];
            const float xa = __uint_as_float(__float_as_uint(v[(e & 1) * 2]) ^ a_sign), xb = __uint_as_float(__float_as_uint(v[(e & 1) * 2 + 1]) ^ a_sign);
            hi[e] = v2_pack(xa, xb);
            const float ra1 = xa - __uint_as_float(hi[e] << 16), rb1 = xb - __uint_as_float(hi[e] & 0xffff0000u);
            mid[e] = v2_pack(ra1, rb1);
            lo[e] = v2_pack(ra1 - __uint_as_float(mid[e] << 16), rb1 - __uint_as_float(mid[e] & 0xffff0000u));
        }
    };
    auto store_limbs = [&](int slot) {
        unsigned char* d = lds + slot * V2_STAGE + a_wr;
        *reinterpret_cast<u32x4*>(d) = u32x4{hi[0], hi[1], hi[2], hi[3]};
        *reinterpret_cast<u32x4*>(d + V2_LIMB) = u32x4{mid[0], mid[1], mid[2], mid[3]};
        *reinterpret_cast<u32x4*>(d + 2 * V2_LIMB) = u32x4{lo[0], lo[1], lo[2], lo[3]};
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const unsigned frag = lr * 32 + ((lh ^ ((lr >> 3) & 1)) << 4);
    const unsigned a_rd = wm * 128 * 32 + frag, b_rd = V2_OPER + wn2 * 64 * 32 + frag;

    // prologue: weights of steps 0 and 1 requested; rows of steps 0 and 1 loaded, split and stored (slots 0, 1); rows of step 2 in flight
    issue_b(0, 0);
    issue_b(1, 1);
    load_a(0, 0);
    load_a(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    split_now(0);
    store_limbs(0);
    split_now(1);
    store_limbs(1);
    load_a(2, 0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();

    for (int ks = 0; ks < NK; ++ks) {
        const int slot = ks % V2_SLOTS;
        // ---- read phase (beside the other group's matrix phase): the rows of step ks + 2 -- requested a step ago -- are split and stored, the
        // fragments of step ks read, the weights of step ks + 2 and the rows of step ks + 3 requested.  Slot (ks + 2) % 3 held step ks - 1, which
        // both groups finished reading before the barrier in front of this phase.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        split_now(0);
        store_limbs((ks + 2) % V2_SLOTS);
        const unsigned char* st = lds + slot * V2_STAGE;
        bf16x8 af[3][4], bf[3][2];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int i = 0; i < 4; ++i) af[pl][i] = *reinterpret_cast<const bf16x8*>(st + a_rd + pl * V2_LIMB + i * 1024);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[pl][j] = *reinterpret_cast<const bf16x8*>(st + b_rd + pl * V2_LIMB + j * 1024);
        }
        issue_b(ks + 2, (ks + 2) % V2_SLOTS);
        load_a(ks + 3, 0);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        // ---- matrix phase: 48 MFMAs, nothing else ----------------------------------------------------------------------------------------
        __builtin_amdgcn_s_setprio(1);
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int q = 0; q < 48; ++q) {
            const int pr = q >> 3, i = (q >> 1) & 3, j = q & 1;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[pr]][i], bf[PB[pr]][j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                             // every wave is out of the loop: the ring can hold the result tiles
#ifdef X3S_DIRECT_EPI
    {
        float* outd = p.out + (size_t)b * p.batch_out;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn2 * 64 + j * 32 + lr;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 128 + i * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
                    const float v = ((lh ^ i) & 1) ? -acc[i][j][r] : acc[i][j][r];
                    if (m < p.M && n < p.N) outd[(size_t)m * p.N + n] = v;
                }
            }
        return;
    }
#endif

    // ---- epilogue through LDS: per wave 64 rows x 64 columns at a time (row stride 68 floats: 16-byte aligned, conflict-light) -------------------
    float* out = p.out + (size_t)b * p.batch_out;
    float* mine = reinterpret_cast<float*>(lds) + wave * (64 * 68);
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
            const int i = h2 * 2 + i2;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i2 * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
                    const float v = ((lh ^ i) & 1) ? -acc[i][j][r] : acc[i][j][r];       // the row's sign s(m) again (bits 2 and 5 of m: lh and i & 1)
                    mine[row * 68 + j * 32 + lr] = v;
                }
        }
        // same wave writes and reads: the LDS queue is in order
        const int c4 = lane & 15, rr = lane >> 4;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int row = t * 4 + rr;
            const f32x4 v = *reinterpret_cast<const f32x4*>(mine + row * 68 + c4 * 4);
            const int m = m0 + wm * 128 + h2 * 64 + row, n = n0 + wn2 * 64 + c4 * 4;
            if (m < p.M) {
                float* po = out + (size_t)m * p.N + n;
                if (n + 3 < p.N) *reinterpret_cast<f32x4*>(po) = v;
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < p.N) po[e] = v[e];
                }
            }
        }
    }
}

}  // namespace
#endif  // SSD_EXPERIMENTAL

// Experiment entry (declared in include/ssd_gfx950.h): both operands as limb planes of ssd_gemm_x3_split_weights (rows_pad = ceil128 of M / n_rows).
extern "C" int ssd_gemm_planes_x3v2(const void* a3, const void* w3, float* out, int M, int K, int N, int n_rows, int nbatch, int dither, void* stream) {
#ifndef SSD_EXPERIMENTAL
    return SSD_ERR_BAD_SHAPE;                       // not in this build (SSD_EXPERIMENTAL=1 python -m objectdetection_ssd_amd.build)
#else
    if (!a3 || !w3 || !out) return SSD_ERR_NULL;
    if (M <= 0 || N <= 0 || K <= 0 || K % 32 != 0 || n_rows < N || nbatch <= 0) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(a3) || !ssd_aligned16(w3) || !ssd_aligned16(out)) return SSD_ERR_ALIGN;
    X3V2Params p{};
    p.a3 = static_cast<const __bf16*>(a3); p.w3 = static_cast<const __bf16*>(w3); p.out = out;
    p.M = M; p.K = K; p.N = N; p.rows_a = ssd_cdiv(M, 128) * 128; p.rows_w = ssd_cdiv(n_rows, 128) * 128;
    p.tiles_m = ssd_cdiv(M, 256); p.tiles_n = ssd_cdiv(N, 256); p.nbatch = nbatch; p.dither = dither ? 1 : 0;
    const size_t nblk = (size_t)p.tiles_m * p.tiles_n * nbatch;
    if (nblk >= (1ull << 31)) return SSD_ERR_BAD_SHAPE;
    static std::atomic<unsigned long long> raised{0};
    int dev;
    if (ssd_attr_needed(raised, dev)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_planes_x3v2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, V2_SLOTS * V2_STAGE) !=
            hipSuccess)
            return SSD_ERR_LAUNCH;
        ssd_attr_done(raised, dev);
    }
    hipLaunchKernelGGL(gemm_planes_x3v2_kernel, dim3((unsigned)nblk), dim3(512), V2_SLOTS * V2_STAGE, (hipStream_t)stream, p);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
#endif
}

// Internal (csrc/gemm_x3.hip dispatches here for the large launches): a [nbatch][M][K] f32 split in the kernel, w3 limb planes.
__attribute__((visibility("hidden"))) int ssd_internal_gemm_batched_x3s(const float* a, const void* w3, float* out, int M, int K, int N, int n_rows,
                                                                         int nbatch, size_t batch_a_elems, hipStream_t st) {
#ifndef SSD_EXPERIMENTAL
    return SSD_ERR_BAD_SHAPE;
#else
    if (K % 32 != 0 || K < 48 || M <= 0 || N <= 0 || N % 4 != 0 || n_rows <= 0 || nbatch <= 0) return SSD_ERR_BAD_SHAPE;
    if ((size_t)M * K * 4 >= 0xF0000000ull) return SSD_ERR_BAD_SHAPE;
    X3SParams p{};
    p.a = a; p.w3 = static_cast<const __bf16*>(w3); p.out = out;
    p.M = M; p.K = K; p.N = N; p.rows_w = ssd_cdiv(n_rows, 128) * 128;
    p.tiles_m = ssd_cdiv(M, 256); p.tiles_n = ssd_cdiv(N, 256); p.nbatch = nbatch;
    p.batch_a = batch_a_elems; p.batch_out = (size_t)M * N;
    const size_t nblk = (size_t)p.tiles_m * p.tiles_n * nbatch;
    if (nblk >= (1ull << 31)) return SSD_ERR_BAD_SHAPE;
    static std::atomic<unsigned long long> raised{0};
    int dev;
    if (ssd_attr_needed(raised, dev)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_planes_x3s_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, V2_SLOTS * V2_STAGE) !=
            hipSuccess)
            return SSD_ERR_LAUNCH;
        ssd_attr_done(raised, dev);
    }
    hipLaunchKernelGGL(gemm_planes_x3s_kernel, dim3((unsigned)nblk), dim3(512), V2_SLOTS * V2_STAGE, st, p);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
#endif
}
