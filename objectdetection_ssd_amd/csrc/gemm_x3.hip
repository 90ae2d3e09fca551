// Batched plane GEMMs of the Winograd convolutions (Model.py:135-156: the 3x3 layers of the VGG trunk, fc6 and the heads) with f32
// operands multiplied on the bf16 MFMA: every f32 value is split EXACTLY into three bf16 limbs x = hi + mid + lo (8 + 8 + 8 significant
// bits, round-to-nearest: both residual subtractions are exact) and a product block is the six limb products of weight >= 2^-18
// (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid) on v_mfma_f32_32x32x16_bf16, accumulated in f32; the dropped terms (mid*lo, lo*mid,
// lo*lo) are <= 2^-26 relative, below the rounding of an f32 product.  Six bf16 MFMAs cost 6/16 of the f32 MFMA block they replace
// (v_mfma_f32_32x32x2_f32: 256 FLOP/clk/CU against 4096).
//
//   NT form  out[b][m][n] = sum_k a[b][m][k] * w[b][n][k]      (forward / data gradient: a = transformed activation planes [tiles][K],
//                                                               w = transformed filter planes, split ONCE per step by the weight job)
//
// Structure: 128 x 128 tile, 4 waves (2 x 2, 64 x 64 each = 2 x 2 MFMA tiles), K step 16 (one MFMA depth), two LDS stages, ONE barrier
// per step (24 MFMAs per wave between barriers).  The weight limbs arrive pre-split and pre-tiled ([k step][limb][row][16]): one
// LDS-DMA instruction per limb and thread, 4 KB contiguous per limb.  The activation tile is loaded as f32 (two 16-byte loads per
// thread and step, a step ahead), split in registers (~44 VALU instructions: v_cvt_pk_bf16_f32 / shift / and / sub) and written as three 16-byte LDS
// stores.  LDS rows are 32 bytes (16 k); the two 16-byte halves of row r are swapped when bit 3 of r is set, so that the 16 lanes
// of a ds_read_b128 group hit 16 different 16-byte slots (MI355X_MICROARCH.md LDS table).
#include "common.h"

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
};

__device__ __forceinline__ f32x4 buf_load16(__amdgpu_buffer_rsrc_t srd, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, (int)voff, (int)soff, 0));
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

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

constexpr int X3_LIMB = 128 * 32;          // bytes of one limb image: [128 rows][16 k] bf16
constexpr int X3_OPER = 3 * X3_LIMB;       // one operand, one stage
constexpr int X3_STAGE = 2 * X3_OPER;      // A + B

__global__ __launch_bounds__(256, 3) void gemm_planes_x3_kernel(const GemmX3Params p) {
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
    const unsigned a_wr = arow * 32 + ((ahalf ^ ((arow >> 3) & 1)) << 4);            // byte offset inside a limb image
    // B: thread -> 16-byte slot `tid` of each limb image; the XOR of the image is applied on the source address
    const int brow = tid >> 1, bhalf = (tid & 1) ^ ((brow >> 3) & 1);
    const size_t limb_elems = (size_t)p.rows_pad * 16;
    const __bf16* b_src = p.w3 + (size_t)b * NK * 3 * limb_elems + (size_t)(n0 + brow) * 16 + bhalf * 8;

    f32x4 ra0, ra1;
    auto load_a = [&](int ks) {
        ra0 = buf_load16(srd_a, voff_a, (unsigned)ks * 64u);
        ra1 = buf_load16(srd_a, voff_a + 16u, (unsigned)ks * 64u);
    };
    auto dma_b = [&](int ks, int stage) {
        const __bf16* s = b_src + (size_t)ks * 3 * limb_elems;
        unsigned char* d = lds + stage * X3_STAGE + X3_OPER + wave * 1024;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(s + pl * limb_elems), (lds_void*)(d + pl * X3_LIMB), 16, 0, 0);
    };
    auto store_a = [&](int stage) {
        u32x4 hi, mid, lo;
        split8(ra0, ra1, hi, mid, lo);
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

    load_a(0);
    dma_b(0, 0);
    store_a(0);                                   // (the compiler waits for ra0 / ra1 here)
    if (NK > 1) {
        load_a(1);
        dma_b(1, 1);
        wait_vmcnt<5>();                          // stage 0's weights have landed; stage 1's loads stay in flight
    } else {
        wait_vmcnt<0>();
    }
    __syncthreads();
    for (int ks = 0; ks < NK; ++ks) {
        const unsigned char* st = lds + (ks & 1) * X3_STAGE;
        bf16x8 af[3][2], bf[3][2];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[pl][i] = *reinterpret_cast<const bf16x8*>(st + a_rd + pl * X3_LIMB + i * 1024);
                bf[pl][i] = *reinterpret_cast<const bf16x8*>(st + b_rd + pl * X3_LIMB + i * 1024);
            }
        // product-major: four independent accumulators between two MFMAs of one chain; smallest terms first
#define X3_MMA(PA, PB)                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                        \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA][i], bf[PB][j], acc[i][j], 0, 0, 0);
        X3_MMA(2, 0)
        X3_MMA(0, 2)
        X3_MMA(1, 1)
        X3_MMA(1, 0)
        X3_MMA(0, 1)
        X3_MMA(0, 0)
#undef X3_MMA
        if (ks + 1 < NK) store_a((ks + 1) & 1);   // the other stage: last read a step ago, a barrier since
        wait_vmcnt<0>();                          // the weights of step ks + 1
        __syncthreads();
        if (ks + 2 < NK) {
            load_a(ks + 2);
            dma_b(ks + 2, ks & 1);
        }
    }

    float* out = p.out + (size_t)b * p.batch_out;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
                if (m < p.M && n < p.N) out[(size_t)m * p.N + n] = acc[i][j][r];
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
    if (rows <= 0 || K <= 0 || K % 16 != 0 || nbatch <= 0) return 0;
    return (size_t)nbatch * K * ssd_cdiv(rows, 128) * 128 * 3 * 2;
}

extern "C" int ssd_gemm_x3_split_weights(const float* w, void* w3, int rows, int K, int nbatch, void* stream) {
    if (!w || !w3) return SSD_ERR_NULL;
    if (rows <= 0 || K <= 0 || K % 16 != 0 || nbatch <= 0) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(w) || !ssd_aligned16(w3)) return SSD_ERR_ALIGN;
    const int rows_pad = ssd_cdiv(rows, 128) * 128;
    const size_t total = (size_t)nbatch * (K / 16) * rows_pad, blocks = (total + 255) / 256;
    hipLaunchKernelGGL(split_weights_x3_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, (hipStream_t)stream, w,
                       static_cast<__bf16*>(w3), rows, rows_pad, K, nbatch);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

// Internal (not part of the C ABI): the x3 form of ssd_internal_gemm_batched (conv_igemm.hip); w3 from ssd_gemm_x3_split_weights
__attribute__((visibility("hidden"))) int ssd_internal_gemm_batched_x3(const float* a, const void* w3, float* out, int M, int K, int N, int n_rows,
                                                                        int nbatch, size_t batch_a_elems, hipStream_t st) {
    if (K % 16 != 0 || M <= 0 || N <= 0 || n_rows <= 0 || ssd_cdiv(n_rows, 128) * 128 < N || nbatch <= 0) return SSD_ERR_BAD_SHAPE;
    if ((size_t)M * K * 4 >= 0xF0000000ull) return SSD_ERR_BAD_SHAPE;
    GemmX3Params p{};
    p.a = a; p.w3 = static_cast<const __bf16*>(w3); p.out = out;
    p.M = M; p.K = K; p.N = N; p.rows_pad = ssd_cdiv(n_rows, 128) * 128;
    p.tiles_m = ssd_cdiv(M, 128); p.tiles_n = ssd_cdiv(N, 128); p.nbatch = nbatch;
    p.batch_a = batch_a_elems; p.batch_out = (size_t)M * N;
    const size_t nblk = (size_t)p.tiles_m * p.tiles_n * nbatch;
    if (nblk >= (1ull << 31)) return SSD_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(gemm_planes_x3_kernel, dim3((unsigned)nblk), dim3(256), 0, st, p);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

extern "C" int ssd_gemm_planes_x3(const float* a, const void* w3, float* out, int M, int K, int N, int n_rows, int nbatch, void* stream) {
    if (!a || !w3 || !out) return SSD_ERR_NULL;
    if (!ssd_aligned16(a) || !ssd_aligned16(w3) || !ssd_aligned16(out)) return SSD_ERR_ALIGN;
    return ssd_internal_gemm_batched_x3(a, w3, out, M, K, N, n_rows, nbatch, (size_t)M * K, (hipStream_t)stream);
}
