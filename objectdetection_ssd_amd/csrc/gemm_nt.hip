// Batched NT GEMM of the Winograd layers with 256 and more input channels:  out[b][m][n] = sum_k a[b][m][k] * w[b][n][k]
// (a = transformed input planes [36][tiles][K], w = transformed filters [36][N][K], out = M planes [36][tiles][N]).
//
// The generic implicit-GEMM kernel (conv_igemm.hip, 64 x 64 tile, operands staged through registers) needs 16 bytes of L2 -> LDS traffic
// per MFMA cycle of its CU at peak, spends two barriers per 32-wide K step, and -- with K = 256 or 512 -- lives for only 8 or 16 K steps
// between a cold start (address arithmetic, first load round trip) and 64 stores per lane.  This kernel is a plain GEMM and nothing else:
//   * 128 x 128 block tile, four waves of 64 x 64 (four 32x32 accumulators, v_mfma_f32_32x32x2_f32): 8 B per MFMA cycle;
//   * operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write), whole 128-byte rows, in 1 KB
//     pieces of 8 rows; two 32 KB slots, ONE barrier per K step; 64 KB of LDS and ~140 registers let two blocks share a CU;
//   * LDS rows are 128 B and unpadded (LDS-DMA writes linearly), so the 16-byte chunks of a row are XOR-swizzled by (row >> 1) & 7 on the
//     source address and on the ds_read_b128 address alike: the 16 lanes of a ds_read_b128 group (MI355X_MICROARCH.md, LDS) hit 16
//     different 16-byte slots of the 256-byte bank row;
//   * PERSISTENT: the grid is two blocks per CU, and a block walks a list of (plane, row tile, column tile) items with ONE pipeline that
//     never drains -- the LDS-DMA of the next item's first two K steps is issued under the last steps of this item, its first fragments
//     are read before this item's last MFMAs, and the stores of this item run while those loads fly;
//   * the items of one XCD (block id mod 8) are a contiguous eighth of the list, handed out round-robin to its blocks: at any time the
//     blocks that share an L2 work on neighbouring tiles of one plane (shared A panels, one 1 MB filter plane).
// Rows beyond M / beyond the filter rows are clamped to the last valid row on load and never stored.
#include "common.h"

#ifdef SSD_EXPERIMENTAL
namespace {

struct NtParams {
    const float* a; const float* w; float* out;
    int M, N, K, n_rows, tiles_m, tiles_n, items;      // items = batches x tiles_m x tiles_n
    size_t batch_a, batch_w, batch_out;
};

typedef __attribute__((address_space(3))) void lds_void;
// 16 bytes per lane, global -> LDS at (wave-uniform) dst + 16 * lane.  A plain function: inside the kernel template the call would be
// type-dependent, and the host pass then drops the instantiation (and with it the launch stub) instead of accepting the device builtin.
__device__ __forceinline__ void dma16(const float* src, float* dst) { __builtin_amdgcn_global_load_lds(src, (lds_void*)dst, 16, 0, 0); }
template <int N> __device__ __forceinline__ void wait_lgkmcnt() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
// R 16-byte reads at byte address `addr` + r * 4096 (the same columns of rows 32 apart), not tracked by the compiler's wait insertion
template <int R> __device__ __forceinline__ void read_rows(f32x4 (&d)[R], unsigned addr) {
    static_assert(R == 1 || R == 2, "one or two 32-row blocks per wave");
    asm volatile("ds_read_b128 %0, %1" : "=v"(d[0]) : "v"(addr) : "memory");
    if (R == 2) asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(d[R - 1]) : "v"(addr) : "memory");
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void wino_gemm_nt_kernel(const NtParams p) {
    constexpr int TM = BM / 64, TN = BN / 64, ROWS = BM + BN, STAGE_F = ROWS * 32, PIECES = ROWS / 32;      // PIECES: per wave and K step
    constexpr unsigned STAGE_B = STAGE_F * 4u;
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE_F];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- this block's items: XCD x owns items [items * x / 8, items * (x + 1) / 8), its blocks take them round-robin --------------------
    const int xcd = blockIdx.x & 7, bpx = (int)(gridDim.x >> 3);
    const int hi = (int)((long long)p.items * (xcd + 1) / 8);
    int it = (int)((long long)p.items * xcd / 8) + (int)(blockIdx.x >> 3);
    if (it >= hi) return;                                                      // uniform
    const int per_batch = p.tiles_m * p.tiles_n;

    // ---- LDS-DMA pieces of this wave: piece q = wave * PIECES + i holds tile rows 8q .. 8q+7 (A rows first, then B rows) -------------
    const int prow = lane >> 3, pc = lane & 7;
    struct Item { int m0, n0; const float* A; const float* B; float* out; unsigned off[PIECES]; };
    auto locate = [&](int item, Item& t) {
        const int batch = item / per_batch, r = item - batch * per_batch;
        const int mt = r / p.tiles_n, nt = r - mt * p.tiles_n;
        t.m0 = mt * BM; t.n0 = nt * BN;
        t.A = p.a + (size_t)batch * p.batch_a;
        t.B = p.w + (size_t)batch * p.batch_w;
        t.out = p.out + (size_t)batch * p.batch_out;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int row = (wave * PIECES + i) * 8 + prow;
            const bool is_a = row < BM;                                        // uniform per piece
            const int local = is_a ? row : row - BM;
            const int grow = is_a ? min(t.m0 + local, p.M - 1) : min(t.n0 + local, p.n_rows - 1);
            t.off[i] = (unsigned)grow * (unsigned)p.K + (unsigned)((pc ^ ((local >> 1) & 7)) << 2);
        }
    };
    Item cur, nxt;
    locate(it, cur);
    bool has_next = it + bpx < hi;
    if (has_next) locate(it + bpx, nxt);
    const int KT = p.K >> 5;                                                   // >= 2 (the launcher checks)
    // K step k of the current item, k = KT / KT + 1 meaning steps 0 / 1 of the next item
    auto issue = [&](int k, int slot) {
        const bool nx = k >= KT;                                               // uniform
        if (nx && !has_next) return;
#ifdef NT_NO_DMA
        if (k > 1) return;                                                     // experiment: multiply stale LDS contents
#endif
        const Item& t = nx ? nxt : cur;
        const int kk = nx ? k - KT : k;
        float* base = lds + slot * STAGE_F + wave * PIECES * 256;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const bool is_a = (wave * PIECES + i) * 8 < BM;
            dma16((is_a ? t.A : t.B) + t.off[i] + kk * 32, base + i * 256);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lr = lane & 31, lh = lane >> 5, key = (lr >> 1) & 7;
    const int a_rd = (wm * (BM / 2) + lr) * 32, b_rd = (BM + wn * (BN / 2) + lr) * 32;
    // Software pipeline inside a K step: the fragments of K group g+1 are read from LDS while the 16 MFMAs of group g run (two register
    // sets), and the last group of a step is multiplied AFTER the step's barrier, under which the first fragments of the next step are
    // already on their way.  The ds_read_b128 are written out with their own counted waits (LDS operations return in order): the
    // compiler's automatic waits drain the queue, newest reads included.  K order inside a step: k = 8g + 4(lane >> 5) + e, both operands.
    struct Frag { f32x4 a[TM], b[TN]; };
    Frag F[2];
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)lds;
    unsigned a_ad[4], b_ad[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int swz = ((2 * g + lh) ^ key) << 2;
        a_ad[g] = lds0 + (unsigned)(a_rd + swz) * 4u;
        b_ad[g] = lds0 + (unsigned)(b_rd + swz) * 4u;
    }
    auto load_frags = [&](Frag& f, unsigned slot_bytes, int g) {
        read_rows<TM>(f.a, a_ad[g] + slot_bytes);
        read_rows<TN>(f.b, b_ad[g] + slot_bytes);
    };
    auto mma16 = [&](const Frag& f) {
#ifdef NT_NO_MFMA
        acc[0][0][0] += f.a[0][0] * f.b[0][0] + f.a[TM - 1][3] * f.b[TN - 1][3];   // experiment: loads and barriers only
        return;
#endif
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][e], f.b[j][e], acc[i][j], 0, 0, 0);
    };
    constexpr int NF = TM + TN;                                                // ds_read_b128 per fragment set

    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    load_frags(F[0], 0u, 0);
    issue(1, 1);
    int kt = 0;
    unsigned sb = 0;                                                           // byte offset of the slot that holds the current stage
    while (true) {
        __builtin_amdgcn_sched_barrier(0);
        load_frags(F[1], sb, 1);
        wait_lgkmcnt<NF>();                                                    // everything older than the reads just issued: F[0]
        __builtin_amdgcn_sched_barrier(0);
        mma16(F[0]);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(F[0], sb, 2);
        wait_lgkmcnt<NF>();
        __builtin_amdgcn_sched_barrier(0);
        mma16(F[1]);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(F[1], sb, 3);
        wait_lgkmcnt<NF>();
        __builtin_amdgcn_sched_barrier(0);
        mma16(F[0]);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");            // my pieces of the next stage have landed, my reads of this one are done
        __builtin_amdgcn_s_barrier();                                          // ... and so have / are everyone's
        asm volatile("" ::: "memory");
        const bool last = kt + 1 == KT;                                        // uniform
        if (!last || has_next) load_frags(F[0], sb ^ STAGE_B, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma16(F[1]);
        __builtin_amdgcn_sched_barrier(0);
        issue(kt + 2, (int)(sb / STAGE_B));                                    // two stages ahead, into the slot this stage has just left
        __builtin_amdgcn_sched_barrier(0);
        sb ^= STAGE_B;
        if (!last) {
            ++kt;
            continue;
        }
        // ---- the item is complete.  C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) ------
        const bool full = cur.m0 + BM <= p.M && cur.n0 + BN <= p.N;            // uniform
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = cur.n0 + wn * (BN / 2) + j * 32 + lr;
                const int mb = cur.m0 + wm * (BM / 2) + i * 32 + 4 * lh;
                float* po = cur.out + (size_t)mb * p.N + n;
                if (full) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) po[(size_t)((r & 3) + 8 * (r >> 2)) * p.N] = acc[i][j][r];
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dm = (r & 3) + 8 * (r >> 2);
                        if (n < p.N && mb + dm < p.M) po[(size_t)dm * p.N] = acc[i][j][r];
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            }
        if (!has_next) break;
        it += bpx;
        cur = nxt;
        has_next = it + bpx < hi;
        if (has_next) locate(it + bpx, nxt);
        kt = 0;
    }
}

}  // namespace
#endif  // SSD_EXPERIMENTAL

namespace {
int g_nt_mode = -1;                // -1 / 0: the generic 64x64 kernel; 1: this kernel whenever the shape allows

}  // namespace

extern "C" int ssd_tune_set_gemm_nt(int mode) {
    if (mode < -1 || mode > 1) return SSD_ERR_BAD_SHAPE;
#ifndef SSD_EXPERIMENTAL
    if (mode == 1) return SSD_ERR_BAD_SHAPE;       // the kernel is not in this build (python -m objectdetection_ssd_amd.build with SSD_EXPERIMENTAL=1)
#endif
    g_nt_mode = mode;
    return SSD_OK;
}

// Internal (not part of the C ABI; called by ssd_internal_gemm_batched): 1 when this kernel takes the launch.
__attribute__((visibility("hidden"))) int ssd_internal_gemm_nt_wants(int M, int K, int N, int n_rows, int nbatch) {
    if (g_nt_mode == 0) return 0;
    if (K % 32 != 0 || K < 64 || N % 4 != 0 || n_rows < 1 || M < 1 || nbatch < 1) return 0;
    // Measured at batch 32 (tools/gemm_bench.py): 106-118 TFLOP/s against the generic kernel's 108-129 on the same launches, and 113-129
    // with its loads compiled out (-DNT_NO_DMA) -- the sustained f32 MFMA rate of the device on random operands, not the loop around it,
    // bounds both kernels, and the 64x64 tiles quantise better.  So: only on request.
    return g_nt_mode == 1;
}

__attribute__((visibility("hidden"))) int ssd_internal_gemm_nt(const float* a, const float* w, float* out, int M, int K, int N, int n_rows,
                                                               int nbatch, size_t batch_a_elems, size_t batch_w_elems, hipStream_t st) {
    if (!a || !w || !out) return SSD_ERR_NULL;
    if (K % 32 != 0 || K < 64 || M < 1 || N < 1 || n_rows < 1 || nbatch < 1) return SSD_ERR_BAD_SHAPE;
#ifndef SSD_EXPERIMENTAL
    return SSD_ERR_BAD_SHAPE;
#else
    NtParams p;
    p.a = a; p.w = w; p.out = out;
    p.M = M; p.N = N; p.K = K; p.n_rows = n_rows;
    p.tiles_m = ssd_cdiv(M, 128); p.tiles_n = ssd_cdiv(N, 128);
    const long long items = (long long)nbatch * p.tiles_m * p.tiles_n;
    if (items >= (1ll << 30)) return SSD_ERR_BAD_SHAPE;
    p.items = (int)items;
    p.batch_a = batch_a_elems; p.batch_w = batch_w_elems; p.batch_out = (size_t)M * N;
    // two blocks per CU (64 KB of LDS each): 64 per XCD, fewer when an XCD's eighth of the items is smaller
    const int per_xcd = (p.items + 7) / 8;
    const int bpx = per_xcd < 64 ? per_xcd : 64;
    hipLaunchKernelGGL((wino_gemm_nt_kernel<128, 128>), dim3((unsigned)(8 * bpx)), dim3(256), 0, st, p);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
#endif
}

// 1 when the library was built with SSD_EXPERIMENTAL (the kernels that measured no better than the shipped ones and are off by default:
// the persistent 128 x 128 NT plane GEMM of this file, the one-kernel Winograd convolution `wino4_full_kernel`)
extern "C" int ssd_has_experimental(void) {
#ifdef SSD_EXPERIMENTAL
    return 1;
#else
    return 0;
#endif
}
