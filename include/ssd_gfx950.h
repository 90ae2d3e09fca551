/*
 * ssd_gfx950.h -- C ABI of libssd_gfx950.so: the MI355X (gfx950) SSD300 hot path.
 *
 * The reference (nitishsaDire/objectDetection_ssd) has no native layer: its hot
 * path is Python calling ATen ops.  This header is the boundary a maintainer
 * binds instead (ctypes stub in INTEGRATION.md).  Every entry point names the
 * reference call site it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - plain pointers + sizes, no framework types; all pointers are DEVICE
 *     pointers unless a parameter is documented as host;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - functions only enqueue work: they never allocate, synchronise or throw;
 *   - return 0 on success or a negative ssd_status code;
 *   - activations are NHWC f32; "packed" head buffers are [N*H*W][ld] f32.
 */
#ifndef SSD_GFX950_H
#define SSD_GFX950_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSD_ABI_VERSION 1

enum ssd_status {
    SSD_OK = 0,
    SSD_ERR_BAD_SHAPE = -1,   /* unsupported / inconsistent dimensions            */
    SSD_ERR_WORKSPACE = -2,   /* workspace too small                              */
    SSD_ERR_NULL = -3,        /* required pointer is NULL                         */
    SSD_ERR_LAUNCH = -4,      /* hipGetLastError() != hipSuccess after enqueue    */
    SSD_ERR_ALIGN = -5        /* pointer / leading dimension not 16-byte aligned  */
};

int ssd_abi_version(void);
const char* ssd_status_string(int status);

/* Geometry of one convolution, forward sense.  (torch.nn.Conv2d arguments of
 * Model.py:135-184: kernel R x S, stride, padding, dilation.) */
typedef struct ssd_conv_geom {
    int32_t N, H, W, Ci;      /* input  NHWC                                     */
    int32_t Ho, Wo, Co;       /* output NHWC                                     */
    int32_t R, S;             /* taps                                            */
    int32_t stride, pad, dil;
} ssd_conv_geom;

/* ---- weight layouts --------------------------------------------------------
 * nn.Conv2d keeps OIHW.  The kernels read K-contiguous rows:
 *   forward : [Co_pad][R*S][Ci]   (rows >= Co zero)
 *   dgrad   : [Ci][R*S][Co_pad]   (columns >= Co zero)
 * Replaces nothing in the reference (ATen re-lays weights internally). */
int ssd_weight_oihw_to_ohwi(const float* w_oihw, float* w_ohwi, int Co, int Ci, int R, int S, int Co_pad, void* stream);
int ssd_weight_oihw_to_ihwo(const float* w_oihw, float* w_ihwo, int Co, int Ci, int R, int S, int Co_pad, void* stream);

/* ---- convolution (Model.py:135-184 nn.Conv2d + nn.ReLU; autograd of them,
 * train_function.py:94) ------------------------------------------------------
 * y[m][n] = act( sum_{tap,c} x[pix(m,tap)][c] * w[n][tap][c] + bias[n] ),  m = (n,ho,wo).
 * Ci % 32 == 0 required (conv1_1 has its own entry below).  ldy = row stride of y
 * in floats (>= Co).  relu: 0/1. */
int ssd_conv2d_fwd(const float* x, const float* w_ohwi, const float* bias, float* y, int ldy,
                   const ssd_conv_geom* g, int relu, void* stream);

/* dx[pix][c] (+)= sum_{tap,n} dy[opix(pix,tap)][n] * w[n][c][tap];  then, if
 * relu_mask != NULL, dx = relu_mask > 0 ? dx : 0 (relu_mask = the post-ReLU
 * activation that produced x).  dy rows have stride ldy (>= Co_pad, pad columns
 * zero), Co_pad % 32 == 0. */
int ssd_conv2d_dgrad(const float* dy, int ldy, const float* w_ihwo, int Co_pad, float* dx,
                     const float* relu_mask, int accumulate, const ssd_conv_geom* g, void* stream);

/* bf16-operand variants (BASELINE.json configs[2] "bf16 convs"): identical signatures and f32 tensors; the
 * operand tiles are rounded to bf16 on their way into LDS and multiplied on v_mfma_f32_32x32x16_bf16 with f32
 * accumulation.  Opt-in (Model.SSD_300.conv_dtype = "bf16").  The weight gradient has a bf16 kernel for the 3x3 / stride 1
 * layers on maps >= 30 px (same workspace query as the f32 entry); other layers fall through to the f32 kernels. */
int ssd_conv2d_fwd_bf16(const float* x, const float* w_ohwi, const float* bias, float* y, int ldy,
                        const ssd_conv_geom* g, int relu, void* stream);
int ssd_conv2d_dgrad_bf16(const float* dy, int ldy, const float* w_ihwo, int Co_pad, float* dx,
                          const float* relu_mask, int accumulate, const ssd_conv_geom* g, void* stream);
/* ... with the split-K workspace of ssd_conv2d_igemm_workspace(g, direction) (may be NULL): small grids with a deep K loop -- the 10x10 ... 1x1
 * maps of the aux blocks and heads -- are cut into K slices, reduced in slice order (reproducible) */
int ssd_conv2d_fwd_bf16_ws(const float* x, const float* w_ohwi, const float* bias, float* y, int ldy, const ssd_conv_geom* g, int relu,
                           void* workspace, size_t workspace_bytes, void* stream);
int ssd_conv2d_dgrad_bf16_ws(const float* dy, int ldy, const float* w_ihwo, int Co_pad, float* dx, const float* relu_mask, int accumulate,
                             const ssd_conv_geom* g, void* workspace, size_t workspace_bytes, void* stream);
int ssd_conv2d_wgrad_bf16(const float* x, const float* dy, int ldy, float* dw_oihw, float* dbias,
                          const ssd_conv_geom* g, void* workspace, size_t workspace_bytes, void* stream);
int ssd_tune_set_igemm_bf16(int tile);   /* 0 = 256x128, 1 = 128x128, 2 = 128x64, 3 = 64x64, -1 = automatic */

/* "f32 from three bf16 limbs" variants (opt-in: Model.SSD_300.conv_dtype = "f32x3"): every operand is split exactly
 * into hi + mid + lo bf16 limbs (8+8+8 significant bits); a product block is the six limb products of weight >= 2^-16
 * on the bf16 MFMA with f32 accumulation (dropped terms <= 2^-24 relative).  Weights are passed pre-split:
 * ssd_weight_split_bf16x3 writes three bf16 planes [3][n] from an f32 array of n elements (the OHWI / IHWO layouts
 * above); w_rows = rows of the OHWI layout (>= Co). */
int ssd_weight_split_bf16x3(const float* w, void* planes_bf16, size_t n, void* stream);
int ssd_conv2d_fwd_x3(const float* x, const void* w3_ohwi, int w_rows, const float* bias, float* y, int ldy,
                      const ssd_conv_geom* g, int relu, void* stream);
int ssd_conv2d_dgrad_x3(const float* dy, int ldy, const void* w3_ihwo, int Co_pad, float* dx,
                        const float* relu_mask, int accumulate, const ssd_conv_geom* g, void* stream);
int ssd_tune_set_igemm_x3(int tile);     /* 1 = 128x128, 2 = 128x64, 3 = 64x64, -1 = automatic */
/* bf16-operand halo-tile kernels for 3x3/s1/p1 layers (plane 0 of the split weights); return 1 (not an error) when the
 * geometry is not a halo case and the caller should use the generic bf16 entry. */
int ssd_conv3x3_halo_fwd_bf16(const float* x, const void* w3_ohwi, int w_rows, const float* bias, float* y, int ldy,
                              const ssd_conv_geom* g, int relu, void* stream);
int ssd_conv3x3_halo_dgrad_bf16(const float* dy, int ldy, const void* w3_ihwo, int Co_pad, float* dx,
                                const float* relu_mask, int accumulate, const ssd_conv_geom* g, void* stream);
int ssd_tune_set_halo(int mode);         /* 3x3/s1 halo-tile kernel: 0 = off, 1 = 8x8 patches, 2 = 8x16 patches, -1 = automatic */

/* ---- bf16 TENSORS (round 3; BASELINE.json configs[2] "bf16 convs" with activations and gradients stored in bf16) ------------
 * 3x3 / stride 1 / pad 1 convolution, forward (flip = 0) and data gradient (flip = 1: the same convolution with the taps
 * mirrored), replacing nn.Conv2d(+ReLU) of Model.py:135-143,176-184 and its autograd data gradient.
 *   x        [N][H][W][ldx] bf16 (forward: the input activation; data gradient: dy), K = reduction channels per tap (multiple of 64, <= ldx)
 *   w        [w_rows][9][K] bf16 (forward: OHWI copy of the f32 master; data gradient: IHWO); rows >= w_rows count as zero
 *   out      [N][H][W][ldo] bf16, or f32 when out_f32 (the heads): columns 0 .. n_out-1 are written (n_out % 4 == 0; bias, if given, has w_rows entries)
 *   out = [accumulate: out +] conv (+ bias) -> [relu] -> [relu_mask (bf16, out's layout): kept where mask > 0]
 * f32 accumulation on v_mfma_f32_32x32x16_bf16, one rounding to bf16 at the store. */
int ssd_conv3x3_bf16(const void* x, int ldx, const void* w, int w_rows, int K, const float* bias, void* out, int ldo, int n_out,
                     int out_f32, const void* relu_mask, int accumulate, int relu, int flip, int N, int H, int W, void* stream);
int ssd_tune_set_conv_bf16_mfma(int rows);        /* MFMA shape of the bf16-tensor convolution kernels: 32 (32x32x16, default) or 16 (16x16x32) */
int ssd_tune_set_conv_bf16_k64(int on);          /* 0 = K = 64 / <= 64-channel launches on the general kernel instead of the persistent one */
int ssd_tune_set_conv_bf16(int mode, int bn);    /* position space 0 / 1 / 2 (patches 8x32, 16x16, flat), N tile 64 / 128; -1 = automatic */
/* Weight gradient of a 3x3 / stride 1 / pad 1 / dilation 1 layer from bf16 x (N,H,W,Ci) and bf16 dy (N,H,W,ldy): dw (Co,Ci,3,3) and dbias
 * in f32 (the fused nine-tap bf16 MFMA kernel; workspace ssd_conv2d_wgrad_workspace(g)); SSD_ERR_BAD_SHAPE for other geometries. */
int ssd_conv3x3_wgrad_bf16t(const void* x_bf16, const void* dy_bf16, int ldy, float* dw_oihw, float* dbias, const ssd_conv_geom* g,
                            void* workspace, size_t workspace_bytes, void* stream);
/* conv1_1 (Model.py:135) in the bf16-tensor mode: x (f32 NCHW image) and the filter rows rounded to bf16, f32 accumulate, y stored as
 * bf16 NHWC; its weight gradient from the f32 image and the bf16 dy (f32 arithmetic). */
int ssd_conv1_first_fwd_bf16(const float* x_nchw, const float* w_rows, const float* bias, void* y_nhwc_bf16, int N, int H, int W, int relu,
                             void* stream);
int ssd_conv1_first_wgrad_bf16(const float* x_nchw, const void* dy_nhwc_bf16, float* dw_rows, float* dbias, int N, int H, int W,
                               void* workspace, size_t workspace_bytes, void* stream);
/* Max pooling (Model.py:135-142), conv4_3 L2 normalisation (Model.py:206-209) and the heads' gradient gather on bf16 NHWC tensors
 * (C % 8 == 0; argmax codes as in ssd_maxpool_fwd; f32 arithmetic, one rounding at the store).  ssd_maxpool_bwd_bf16: relu_mask (the bf16
 * activation, dx kept where > 0) and accumulate, or y_gate (the bf16 pooled output: the gated form, no accumulate). */
int ssd_maxpool_fwd_bf16(const void* x, void* y, uint8_t* argmax, int N, int H, int W, int C, int k, int stride, int pad, int Ho, int Wo,
                         void* stream);
int ssd_maxpool_bwd_bf16(const void* dy, const uint8_t* argmax, void* dx, const void* relu_mask, const void* y_gate, int accumulate, int N,
                         int H, int W, int C, int k, int stride, int pad, int Ho, int Wo, void* stream);
int ssd_l2norm_fwd_bf16(const void* x, const float* gamma, void* y, int M, int C, void* stream);
size_t ssd_l2norm_bwd_bf16_workspace(int M, int C);
int ssd_l2norm_bwd_bf16(const void* x, const float* gamma, const void* dy, void* dx, float* dgamma, int M, int C, void* workspace,
                        size_t workspace_bytes, void* stream);
int ssd_heads_gather_bf16(const float* dloc, const float* dconf, void* packed_bf16, int ld, int N, int HW, int A, int prior_off, int P,
                          int ncls, void* stream);
int ssd_cast_f32_bf16(const float* x, void* y_bf16, size_t n, void* stream);      /* n % 8 == 0 */
int ssd_cast_bf16_f32(const void* x_bf16, float* y, size_t n, void* stream);
/* ssd_weights_prepare job kind 3: out_fwd = bf16 OHWI [co_pad][taps][ci], out_bwd = bf16 IHWO [ci][taps][pad1] (pad1 >= co, zero filled). */

/* dw_oihw[n][c][r][s] = sum_m dy[m][n] * x[pix(m,tap)][c];  dbias[n] = sum_m dy[m][n]
 * (dbias may be NULL).  Deterministic: split-K partial slabs in `workspace`, then a
 * fixed-order reduction. */
size_t ssd_conv2d_wgrad_workspace(const ssd_conv_geom* g);
int ssd_conv2d_wgrad(const float* x, const float* dy, int ldy, float* dw_oihw, float* dbias,
                     const ssd_conv_geom* g, void* workspace, size_t workspace_bytes, void* stream);

/* Introspection for profiling: which tile configuration the dispatcher picks (direction 0 = forward,
 * 1 = dgrad), and the wgrad tile edge / split-K factor.  Used by bench.py to attribute HIP-event
 * timings to kernel instantiations; no effect on results. */
int ssd_conv2d_igemm_tile(const ssd_conv_geom* g, int direction, int* bm, int* bn);
int ssd_conv2d_wgrad_tile(const ssd_conv_geom* g, int* bt, int* nsplit);
/* Tuning aids (process-global, not thread-safe, results unchanged): force the igemm tile
 * (0 = 256x64, 1 = 128x128, 2 = 128x64, 3 = 64x64) / LDS stage count (1|2), and the wgrad tile edge
 * (64|128) / stage count / split-K target in blocks per CU.  -1 = automatic. */
/* Same convolutions with an optional scratch buffer: launches whose 64x64-tile grid would leave most of the chip idle
 * while each block walks a long K loop (the 19x19 and smaller maps: c_7 ... c_11, seq8 ... seq11) are split along K
 * into partial tiles in the workspace and finished (bias / accumulate / ReLU / mask) by a fixed-order reduction.
 * workspace may be NULL or smaller than ssd_conv2d_igemm_workspace(g, direction) asks: the launch is then not split. */
size_t ssd_conv2d_igemm_workspace(const ssd_conv_geom* g, int direction /* 0 forward, 1 dgrad */);
int ssd_conv2d_fwd_ws(const float* x, const float* w_ohwi, const float* bias, float* y, int ldy, const ssd_conv_geom* g,
                      int relu, void* workspace, size_t workspace_bytes, void* stream);
int ssd_conv2d_dgrad_ws(const float* dy, int ldy, const float* w_ihwo, int Co_pad, float* dx, const float* relu_mask,
                        int accumulate, const ssd_conv_geom* g, void* workspace, size_t workspace_bytes, void* stream);
int ssd_tune_set_igemm_splitk(int k);       /* -1 automatic, 1 never split, k > 1 force k slices */
/* ---- Winograd F(mo x mo, 3x3), mo = 2 or 4, for the 3x3 / stride 1 / pad 1 layers: (mo+2)^2 multiplies per mo x mo output tile
 * instead of 9 mo^2 (2.25x / 4x fewer).  Results differ from the direct sum at the 1e-6 (mo = 2) / 1e-5 (mo = 4) level of the
 * output scale.  ssd_wino_weights transforms the OIHW filter once per update: U_fwd [P][Co][Ci] for the forward, U_bwd
 * [P][Ci][Co_pad] (transposed, rotated filter) for dgrad, P = (mo+2)^2; either may be NULL.  The convolutions take the
 * transformed filter, a workspace of ssd_conv3x3_wino_workspace(g, direction, mo) bytes, and fuse the direct kernels' epilogues. */
int ssd_wino_weights(const float* w_oihw, float* U_fwd, float* U_bwd, int Co, int Ci, int Co_pad, int mo, void* stream);
size_t ssd_conv3x3_wino_workspace(const ssd_conv_geom* g, int direction /* 0 forward, 1 dgrad */, int mo);
int ssd_conv3x3_wino_fwd(const float* x, const float* U_fwd, const float* bias, float* y, int ldy, const ssd_conv_geom* g, int relu,
                         int mo, void* workspace, size_t workspace_bytes, void* stream);
int ssd_conv3x3_wino_dgrad(const float* dy, int ldy, const float* U_bwd, int Co_pad, float* dx, const float* relu_mask,
                           int accumulate, const ssd_conv_geom* g, int mo, void* workspace, size_t workspace_bytes, void* stream);
/* conv3x3 -> ReLU -> MaxPool2d(2, 2, ceil_mode) in one pass (Model.py:135-137 via the VGG layer list: conv1_2, conv2_2, conv3_3 and
 * their pools): F(4x4,3x3) whose output transform reduces each 4x4 tile to its four pool windows, so the full-resolution
 * activation -- read by nothing but the pool -- is never written.  y_pooled (N,Ho,Wo,Co), Ho = H/2 (ceil_mode: (H+1)/2); argmax
 * (may be NULL) in ssd_maxpool_fwd's encoding, for ssd_maxpool_bwd / ssd_maxpool_bwd_gated.  Co % 4 == 0; workspace as
 * ssd_conv3x3_wino_workspace(g, 0, 4).  Equal, bit for bit, to ssd_conv3x3_wino_fwd(relu = 1, mo = 4) followed by ssd_maxpool_fwd.
 * planes_keep (may be NULL): see ssd_conv3x3_wino_fwd_keep. */
int ssd_conv3x3_wino_fwd_pool(const float* x, const float* U_fwd, const float* bias, float* y_pooled, uint8_t* argmax,
                              const ssd_conv_geom* g, int ceil_mode, float* planes_keep, void* workspace, size_t workspace_bytes,
                              void* stream);
/* F(4x4,3x3) forward that leaves the transformed input B^T d B in planes_keep -- 36 x tiles x Ci floats, tiles = N ceil(H/4) ceil(W/4),
 * layout [plane][tile][channel] -- for ssd_conv3x3_wino_wgrad_planes, which then skips transforming x a second time in the backward
 * pass (the planes are 2.25x the activation; 288 GB of HBM pay for that).  Otherwise ssd_conv3x3_wino_fwd with mo = 4. */
int ssd_conv3x3_wino_fwd_keep(const float* x, const float* U_fwd, const float* bias, float* y, int ldy, const ssd_conv_geom* g, int relu,
                              float* planes_keep, void* workspace, size_t workspace_bytes, void* stream);
/* Winograd weight gradient: dg = G^T [ sum over tiles (A dy A^T) (x) (B^T d B) ] G -- transposed transforms of dy and x, sixteen
 * batched (split-K) f32-MFMA GEMMs over the tile dimension, inverse transform to OIHW; dbias (may be NULL) by column sums. */
size_t ssd_conv3x3_wino_wgrad_workspace(const ssd_conv_geom* g, int ldy, int mo);
int ssd_conv3x3_wino_wgrad(const float* x, const float* dy, int ldy, float* dw_oihw, float* dbias, const ssd_conv_geom* g, int mo,
                           void* workspace, size_t workspace_bytes, void* stream);
/* The same with the x planes a ..._fwd_keep / ..._fwd_pool call kept (F(4x4) only); workspace as ssd_conv3x3_wino_wgrad_workspace(g, ldy, 4).
 * dgrad_planes_out (may be NULL; 36 x tiles x ldy floats): the pass over dy also writes B^T dy B, the input planes of this layer's
 * data gradient, which ssd_conv3x3_wino_dgrad_planes (Co_pad = ldy) then takes instead of transforming dy again. */
/* Every filter transform / re-layout of a training step in one launch.  A job: kind 0 = Winograd F(4x4,3x3) filters (out_fwd [36][co][ci],
 * out_bwd [36][ci][co_pad]: the transposed, rotated filter of the data gradient), kind 1 = the direct kernels' copies (out_fwd OHWI
 * [co_pad][taps][ci], out_bwd IHWO [ci][taps][co_pad]), kind 2 = conv1_1's rows for the im2col GEMM (out_fwd [co][32]).  The OIHW source
 * may come in two pieces (rows 0..co0-1 from w0, the rest from w1: a head's loc and conf filters); out_bwd may be NULL.  The caller
 * uploads the job array and the exclusive prefix sum of ssd_weight_job_blocks(job) (block_start, njobs entries) to the device once. */
typedef struct ssd_weight_job {
    const float* w0; const float* w1;
    float* out_fwd; float* out_bwd;
    int co0, co, ci, taps, co_pad, kind, pad0, pad1;
} ssd_weight_job;
int ssd_weight_job_blocks(const ssd_weight_job* job);
int ssd_weights_prepare(const ssd_weight_job* jobs_device, const int* block_start_device, int njobs, int total_blocks, void* stream);
/* The F(4x4) weight gradient on kept planes as two calls, so that the caller can put the second on another stream: nothing in the
 * backward pass waits for dw, while the data gradient waits for dgrad_planes_out.  (1) ssd_wino4_dy_transform: one pass over dy ->
 * wgrad_planes (36 x tiles x ldy: A dy A^T), optionally dgrad_planes_out (36 x tiles x ldy: B^T dy B, for ssd_conv3x3_wino_dgrad_planes)
 * and bias_partial (ssd_wino4_bias_partial_floats(g, ldy) floats: per-block column sums of dy; needs ldy <= 1024).
 * (2) ssd_wino4_wgrad_gemm: the 36 split-K TN GEMMs wgrad_planes^T x x_planes, inverse transform to OIHW, and dbias from bias_partial;
 * workspace ssd_wino4_wgrad_gemm_workspace(g, ldy) bytes.  Together they equal ssd_conv3x3_wino_wgrad_planes bit for bit. */
size_t ssd_wino4_bias_partial_floats(const ssd_conv_geom* g, int ldy);
int ssd_wino4_dy_transform(const float* dy, int ldy, const ssd_conv_geom* g, float* wgrad_planes, float* dgrad_planes_out,
                           float* bias_partial, void* stream);
size_t ssd_wino4_wgrad_gemm_workspace(const ssd_conv_geom* g, int ldy);
int ssd_wino4_wgrad_gemm(const float* wgrad_planes, const float* x_planes, int ldy, const float* bias_partial, float* dw_oihw, float* dbias,
                         const ssd_conv_geom* g, void* workspace, size_t workspace_bytes, void* stream);
/* The same two passes when dy is the gradient of the 2x2 / stride-2 / no-pad max pool that is the only reader of this layer's ReLU output
 * (ssd_conv3x3_wino_fwd_pool): dy is never written to memory.  The pass takes the pooled gradient dpool (N x Hp x Wp x Co), the argmax
 * codes and the pooled forward output of that pool (its ReLU gate) and forms dy(h, w) on the fly -- what ssd_maxpool_bwd_gated would
 * have scattered (reference: autograd of MaxPool2d(ReLU(conv)), Model.py:135-137).  ldy must equal Co; Hp = H/2 or (H+1)/2. */
int ssd_wino4_dy_transform_pooled(const float* dpool, const unsigned char* argmax, const float* y_pooled, int Hp, int Wp, int ldy,
                                  const ssd_conv_geom* g, float* wgrad_planes, float* dgrad_planes_out, float* bias_partial, void* stream);
int ssd_conv3x3_wino_wgrad_planes_pooled(const float* planes, const float* dpool, const unsigned char* argmax, const float* y_pooled,
                                         int Hp, int Wp, int ldy, float* dw_oihw, float* dbias, const ssd_conv_geom* g,
                                         float* dgrad_planes_out, void* workspace, size_t workspace_bytes, void* stream);
/* ReLU masks as bits.  The forward's input transform can leave, per (tile, channel quad), one 64-bit word -- bit (a*4+b)*4+e set iff
 * x[4 th + a][4 tw + b][4 c4 + e] > 0 -- in relu_bits_out (tiles x Ci/4 words; may be NULL): the ReLU mask of the layer's INPUT on the tile
 * grid its data gradient is written on (autograd's ReLU backward, Model.py:136-141).  ssd_conv3x3_wino_dgrad_planes_bits applies it in
 * place of ssd_conv3x3_wino_dgrad_planes's float relu_mask, reading 1/32 of the bytes; results are identical. */
int ssd_conv3x3_wino_fwd_keep_bits(const float* x, const float* U_fwd, const float* bias, float* y, int ldy, const ssd_conv_geom* g, int relu,
                                   float* planes_keep, uint64_t* relu_bits_out, void* workspace, size_t workspace_bytes, void* stream);
int ssd_conv3x3_wino_fwd_pool_bits(const float* x, const float* U_fwd, const float* bias, float* y_pooled, uint8_t* argmax,
                                   const ssd_conv_geom* g, int ceil_mode, float* planes_keep, uint64_t* relu_bits_out, void* workspace,
                                   size_t workspace_bytes, void* stream);
int ssd_conv3x3_wino_dgrad_bits(const float* dy, int ldy, const float* U_bwd, int Co_pad, float* dx, const uint64_t* relu_bits, int accumulate,
                                const ssd_conv_geom* g, void* workspace, size_t workspace_bytes, void* stream);
int ssd_conv3x3_wino_dgrad_planes_bits(const float* dy_planes, const float* U_bwd, int Co_pad, float* dx, const uint64_t* relu_bits,
                                       int accumulate, const ssd_conv_geom* g, void* workspace, size_t workspace_bytes, void* stream);
int ssd_conv3x3_wino_wgrad_planes(const float* planes, const float* dy, int ldy, float* dw_oihw, float* dbias, const ssd_conv_geom* g,
                                  float* dgrad_planes_out, void* workspace, size_t workspace_bytes, void* stream);
int ssd_conv3x3_wino_dgrad_planes(const float* dy_planes, const float* U_bwd, int Co_pad, float* dx, const float* relu_mask,
                                  int accumulate, const ssd_conv_geom* g, void* workspace, size_t workspace_bytes, void* stream);
/* conv1_1 written as the next layer's Winograd input (round 4 EXPERIMENT: ssd_conv1_first_wino_fwd returns SSD_ERR_BAD_SHAPE unless built
 * with SSD_EXPERIMENTAL=1 -- bit-identical but measured slower than the two kernels it replaces; Model.py:135 features[0:4]:
 * Conv2d(3,64) -> ReLU -> Conv2d(64,64)).  In
 * training nothing but conv1_2 reads conv1_1's activation, so ssd_conv1_first_wino_fwd leaves it as the F(4x4) input planes
 * (36 x N*ceil(H/4)*ceil(W/4) x 64 f32) + the ReLU bit words of ssd_conv3x3_wino_fwd_keep_bits, bit-identical to ssd_conv1_first_fwd followed
 * by that layer's input transform, and the 64-channel activation (737 MB at batch 32) is neither written nor read.
 * ssd_conv3x3_wino_fwd_from_planes is the forward of a layer whose planes are given: plane GEMMs + output transform (+ the fused 2x2 pool
 * when y_pooled is not NULL; then y may be NULL). */
int ssd_conv1_first_wino_fwd(const float* x_nchw, const float* w_rows, const float* bias, float* planes, uint64_t* relu_bits, int N, int H, int W,
                             void* stream);
int ssd_conv3x3_wino_fwd_from_planes(const float* planes, const float* U_fwd, const float* bias, float* y, int ldy, float* y_pooled, uint8_t* argmax,
                                     const ssd_conv_geom* g, int relu, int ceil_mode, void* workspace, size_t workspace_bytes, void* stream);
/* Data gradient in the ADJOINT Winograd form (round 4; autograd of nn.Conv2d, reference Model.py:135-143 / train_function.py:94).
 * The forward y = A^T[(G g G^T) (.) (B^T d B)]A is linear in d; transposed, dx = overlap-add over the tiles of the 6x6 patches
 * B[(G g G^T) (.) (A dy A^T)]B^T: the data gradient multiplies the SAME planes A dy A^T (ssd_wino4_dy_transform's wgrad_planes) the
 * weight gradient multiplies, so the second plane set B^T dy B (dgrad_planes_out) is neither written nor read, against the forward
 * filter transform laid out [36][Ci][Co_pad] (ssd_wino_weights_adj; weight jobs: pad0 bit 2; f32 or limb planes by
 * ssd_wino_uses_x3(4, Co_pad)).  (1) ssd_conv3x3_wino_dgrad_adj_gemm: md_planes (ssd_wino4_adj_planes_floats(g) floats:
 * 36 x tiles x pad4(Ci)) = y_planes x U_adj, ldy % 32 == 0.  (2a) ssd_wino4_adj_output: dx (N,H,W,Ci) [+=] the overlap-added patches,
 * ReLU-masked by relu_bits (as left by ssd_conv3x3_wino_fwd_keep_bits) or by the f32 tensor relu_mask (either may be NULL); or
 * (2b) ssd_wino4_adj_output_to_planes: the same block of dx is not stored but taken, masked, as the dy of the layer BELOW (g_below:
 * same map, Co = g->Ci) and leaves that layer's planes A dy A^T (36 x tiles x g->Ci) and bias partial sums
 * (ssd_wino4_bias_partial_floats(g_below, g->Ci); may be NULL): between two chained 3x3 layers the gradient tensor never reaches memory.
 * 3x3 / stride 1 / pad 1 / dilation 1 only. */
int ssd_wino_weights_adj(const float* w_oihw, float* U_adj, int Co, int Ci, int Co_pad, void* stream);
size_t ssd_wino4_adj_planes_floats(const ssd_conv_geom* g);
int ssd_conv3x3_wino_dgrad_adj_gemm(const float* y_planes, int ldy, const float* U_adj, float* md_planes, const ssd_conv_geom* g, void* stream);
int ssd_wino4_adj_output(const float* md_planes, float* dx, const float* relu_mask, const uint64_t* relu_bits, int accumulate,
                         const ssd_conv_geom* g, void* stream);
int ssd_wino4_adj_output_to_planes(const float* md_planes, const float* relu_mask, const uint64_t* relu_bits, const ssd_conv_geom* g,
                                   const ssd_conv_geom* g_below, float* y_planes, float* bias_partial, void* stream);
/* Forward / dgrad of the F(4x4,3x3) entry points: 36 plane GEMMs + output transform in ONE kernel (accumulators of all planes in
 * registers, the M planes never reach memory) where the reduction length is a multiple of 64.  -1 (default): where it is the faster
 * form; 0: never (batched GEMM + output transform kernels); 1: wherever the geometry allows. */
int ssd_tune_set_wino_bias_tail(int on);   /* 1 (default): the bias gradient's final column sums run as extra blocks of the weight gradient's finish kernel; 0: own launch */
int ssd_tune_set_wino_xform_blocks(int blocks);   /* grid cap of the Winograd transform kernels (default 8192; 64 .. 65535) */
int ssd_tune_set_wino_fused(int mode);
/* Experiment (round 4, csrc/gemm_x3v2.hip; SSD_EXPERIMENTAL builds only -- otherwise SSD_ERR_BAD_SHAPE; not on the step's path): the same plane GEMMs with BOTH operands given as limb planes of
 * ssd_gemm_x3_split_weights (a3: rows = M, w3: rows = n_rows), 256 x 256 tiles, LDS-DMA on both sides, two wave groups in ping-pong.
 * dither = 1: a3's rows carry the sign s(m) of csrc/gemm_x3.hip and the result is multiplied by it again. */
int ssd_gemm_planes_x3v2(const void* a3, const void* w3, float* out, int M, int K, int N, int n_rows, int nbatch, int dither, void* stream);
int ssd_tune_set_x3_big(int mode);         /* limb plane GEMMs: 0 (default) = always the 128 x 128 kernel; SSD_EXPERIMENTAL builds only: 1 = launches with N >= 256 and M >= 2048 on the 256 x 256 ping-pong kernel (csrc/gemm_x3v2.hip), 2 = every launch it takes */
int ssd_tune_set_x3_mfma(int rows);        /* limb plane GEMMs (csrc/gemm_x3.hip): 32 (default) = v_mfma_f32_32x32x16_bf16; SSD_EXPERIMENTAL builds only: 16 = v_mfma_f32_16x16x32_bf16, two limb products per instruction */
/* The plane GEMMs of the layers that do not take the one-kernel form: persistent 128 x 128 LDS-DMA kernel (gemm_nt.hip) instead of the
 * generic 64 x 64 implicit-GEMM kernel.  1: wherever K % 32 == 0 and K >= 64; -1 (default) / 0: never -- measured no faster (both kernels
 * sit at the device's sustained f32 MFMA rate); kept, tested bit-identical, as the evidence for that statement. */
int ssd_tune_set_gemm_nt(int mode);
/* f32 plane GEMMs on the bf16 MFMA from three exact bf16 limbs per operand (csrc/gemm_x3.hip; the Winograd-domain products of the
 * 3x3 layers, Model.py:135-156): out[b][m][n] = sum_k a[b][m][k] * w[b][n][k].  The filter planes are split once
 * (ssd_gemm_x3_split_weights: w [nbatch][rows][K] f32 -> w3, ssd_gemm_x3_weights_bytes(rows, K, nbatch) bytes); K % 16 == 0, n_rows >= N.
 * ssd_gemm_planes_f32: the same product on the f32 MFMA (K % 32 == 0), the kernel the x3 form replaces -- both are public for the
 * parity test and tools/gemm_x3_bench.py. */
/* ssd_wino_uses_x3(mo, K): 1 if the F(4x4) plane GEMMs of reduction length K run the three-limb form -- the transformed filter of such
 * a layer (U_fwd: K = Ci; U_bwd: K = Co_pad) is then NOT [36][rows][K] f32 but [36][K/16][3][pad128(rows)][16] bf16 limbs
 * (ssd_gemm_x3_weights_bytes(rows, K, 36) bytes, zero-initialised by the caller: padding rows are never written), both from
 * ssd_wino_weights and from a kind-0 weight job with bit 0 (out_fwd) / bit 1 (out_bwd) of pad0 set.  ssd_tune_set_wino_x3(0 | 1 | -1):
 * off / on / the default (environment SSD_WINO_X3, on unless "0"); filters transformed under one setting must not be used under another. */
int ssd_wino_uses_x3(int mo, int K);
int ssd_tune_set_wino_x3(int on);
size_t ssd_gemm_x3_weights_bytes(int rows, int K, int nbatch);
int ssd_gemm_x3_split_weights(const float* w, void* w3, int rows, int K, int nbatch, void* stream);
int ssd_gemm_planes_x3(const float* a, const void* w3, float* out, int M, int K, int N, int n_rows, int nbatch, void* stream);
/* 1x1 / stride-1 convolutions with long reductions (fc7, seq8.0: Model.py:150-156) on the same limb kernels.  Filters as limb planes:
 * forward w3 = limbs of w [Co][Ci] (ssd_gemm_x3_split_weights(w, w3, Co, Ci, 1), or a kind-4 job of ssd_weights_prepare: out_fwd rows Co,
 * K = Ci; out_bwd rows Ci, K = co_pad = limbs of w^T); Ci % 32 == 0 forward, ldy = Co_pad % 32 == 0 for the data gradient (the reduction
 * runs over all ldy columns of dy: padding columns must be zero).  Epilogues as ssd_conv2d_fwd / ssd_conv2d_dgrad. */
int ssd_conv1x1_fwd_x3(const float* x, const void* w3, const float* bias, float* y, int ldy, const ssd_conv_geom* g, int relu, void* stream);
int ssd_conv1x1_dgrad_x3(const float* dy, int ldy, const void* w3t, float* dx, const float* relu_mask, int accumulate, const ssd_conv_geom* g,
                         void* stream);
size_t ssd_conv1x1_wgrad_x3_workspace(const ssd_conv_geom* g, int ldy);
int ssd_conv1x1_wgrad_x3(const float* x, const float* dy, int ldy, float* dw, float* dbias, const ssd_conv_geom* g, void* workspace,
                         size_t workspace_bytes, void* stream);
int ssd_gemm_planes_f32(const float* a, const float* w, float* out, int M, int K, int N, int n_rows, int nbatch, void* stream);
int ssd_has_experimental(void);           /* 1 if built with SSD_EXPERIMENTAL: gemm_nt.hip and wino4_full_kernel (both off by default, forced by ssd_tune_set_gemm_nt(1) / ssd_tune_set_wino_full(1)) are present */
/* The whole convolution (input transform too) in one kernel where the reduction length is 64 (128 when forced): -1 automatic, 0 never, 1 force.
 * ssd_conv3x3_wino_uses_full tells the caller whether a geometry's forward (0) / data gradient from dy (1) takes that kernel -- it then
 * needs no dgrad planes from ssd_wino4_dy_transform. */
int ssd_tune_set_wino_full(int mode);
int ssd_conv3x3_wino_uses_full(const ssd_conv_geom* g, int direction);
int ssd_tune_set_wino_fused_stagger(int cycles);   /* first-round start delay step between CUs (shader cycles); -1 / 0 (default): none */
int ssd_tune_set_wino_fused_stamps(uint64_t* device_buffer);   /* diagnostic: in-kernel phase stamps of the fused kernel; NULL = off */
int ssd_tune_set_wino_wgrad_tn(int on);   /* 1 (default): F(4x4) weight gradient on untransposed planes + TN GEMM; 0: transposed planes */
/* Measurement aid: arm / read back per-launch timings (library-owned HIP events) of the batched Winograd GEMM kernel.
 * collect() returns the number of (milliseconds, executed FLOPs) pairs written; the caller synchronises the stream first. */
int ssd_prof_gemm_begin(void);
int ssd_prof_gemm_collect_kinds(float* ms_out, double* flops_out, int* kinds_out, int max);   /* kind 0: batched plane GEMMs, 1: fused GEMM + output transform */
int ssd_prof_gemm_collect(float* ms_out, double* flops_out, int max);
int ssd_tune_set_igemm(int tile, int nbuf);
int ssd_tune_set_igemm_stamps(uint64_t* device_buffer);   /* diagnostic: per-block shader-clock stamps (see conv_igemm.hip) */
int ssd_tune_set_dgrad_parity(int on);   /* 1 (default): data gradients of stride-2 convolutions with their rows grouped by pixel parity (only the taps that reach a class are multiplied); 0: plain kernel */
int ssd_tune_set_batched_units(int on);   /* 1 (default): the blocks of one (plane, part of the row tiles) of a batched plane GEMM share one XCD; 0: 3-D grid */
int ssd_tune_set_igemm_lds_pad(int bytes);   /* extra dynamic LDS per block: caps resident blocks per CU (experiments) */
int ssd_tune_set_wgrad(int bt, int nbuf, int blocks_per_cu);
int ssd_tune_set_wgrad_patch(int shape);     /* f32 fused 3x3 kernel: -1 least padding, 0 = 4x8 pixel patches, 1 = 1x38, 2 = 2x19 */

/* conv1_1 (Model.py:136 features[0]: Conv2d(3,64,3,padding=1)+ReLU, Ci = 3): im2col of the caller's
 * NCHW image batch (Dataset.py:39 layout) into [N*H*W][32] rows (k = (r*3+s)*3 + c, columns 27..31
 * zero); forward / wgrad are then the 1x1 cases of ssd_conv2d_fwd / ssd_conv2d_wgrad with Ci = 32. */
int ssd_im2col_first(const float* x_nchw, float* out, int N, int H, int W, void* stream);
/* conv1_1 + ReLU in one kernel (Model.py:135, features[0:2]): x (N,3,H,W) NCHW -> y (N,H,W,64) NHWC = relu(conv3x3 pad 1 + bias).
 * w_rows: [64][32] filter rows in ssd_im2col_first's column order (k = (r*3+s)*3 + c, columns 27..31 zero).  col_out (may be NULL):
 * the same pass also writes the (N,H,W,32) rows ssd_im2col_first would, for the weight gradient (ssd_conv2d_wgrad on them). */
int ssd_conv1_first_fwd(const float* x_nchw, const float* w_rows, const float* bias, float* y_nhwc, float* col_out, int N, int H, int W,
                        int relu, void* stream);
/* Weight and bias gradient of conv1_1 from the NCHW input itself (autograd of Model.py:135 features[0]): dw_rows [64][32] in the column
 * order above (columns 27..31 zero; ops.first_weight_grad turns them into OIHW), dbias [64] (may be NULL) = sum of dy over the pixels.
 * dy (N,H,W,64) NHWC dense.  Workspace ssd_conv1_first_wgrad_workspace bytes (partial sums of the persistent workgroups, added in order). */
size_t ssd_conv1_first_wgrad_workspace(int N, int H, int W);
int ssd_conv1_first_wgrad(const float* x_nchw, const float* dy_nhwc, float* dw_rows, float* dbias, int N, int H, int W, void* workspace,
                          size_t workspace_bytes, void* stream);

/* ---- SSD_resnet34 (Model.py:12-126, BASELINE configs[4]) eval-mode forward pieces ----
 * Stem Conv2d(3,64,7,stride 2,pad 3) (Model.py:26 seq1[0]): general 3-channel NCHW im2col into [N*Ho*Wo][Kpad]
 * rows (k = (r*S+s)*3 + c, zero from R*S*3 to Kpad; Kpad % 32 == 0 for the 1x1 case of ssd_conv2d_fwd). */
int ssd_im2col_nchw3(const float* x_nchw, float* out, int N, int H, int W, int R, int S, int stride, int pad,
                     int Ho, int Wo, int Kpad, void* stream);
/* BasicBlock tail `relu(bn2(conv2(o)) + identity)` (torchvision layer list sliced at Model.py:27-30): the
 * convolution (BatchNorm folded into w / bias by the host) is ADDED to y_inout, which holds the identity /
 * downsample branch on entry, then ReLU'd in place. */
int ssd_conv2d_fwd_accum(const float* x, const float* w_ohwi, const float* bias, float* y_inout, int ldy,
                         const ssd_conv_geom* g, int relu, void* stream);
int ssd_conv2d_fwd_accum_bf16(const float* x, const float* w_ohwi, const float* bias, float* y_inout, int ldy,
                              const ssd_conv_geom* g, int relu, void* stream);
/* eval-mode BatchNorm2d after a ReLU (Model.py:56-62 Conv -> ReLU -> BN -> Dropout2d): y = x*scale[c] + shift[c]
 * over [M][C] NHWC rows, C % 4 == 0, optional ReLU; x may alias y. */
int ssd_channel_affine(const float* x, const float* scale, const float* shift, float* y, size_t M, int C, int relu,
                       void* stream);

/* ---- max pooling (Model.py:137,142 nn.MaxPool2d incl. ceil_mode; features[4,9,23]) ----
 * argmax: uint8 window-relative index (r*k+s) of the first maximum, for backward. */
int ssd_maxpool_fwd(const float* x, float* y, uint8_t* argmax, int N, int H, int W, int C,
                    int k, int stride, int pad, int Ho, int Wo, void* stream);
/* dx (+)= routed dy; if relu_mask != NULL dx = relu_mask > 0 ? dx : 0. */
int ssd_maxpool_bwd(const float* dy, const uint8_t* argmax, float* dx, const float* relu_mask, int accumulate,
                    int N, int H, int W, int C, int k, int stride, int pad, int Ho, int Wo, void* stream);
/* Same gradient for a pool whose input is a ReLU output and has no other consumer: dx = relu_mask(x) * scatter(dy) computed
 * as scatter(dy gated by y > 0), y = the pooled output (an arg-max input equals its window's output) -- reads y instead of
 * the 4x larger x. */
int ssd_maxpool_bwd_gated(const float* dy, const uint8_t* argmax, const float* y, float* dx, int N, int H, int W, int C,
                          int k, int stride, int pad, int Ho, int Wo, void* stream);

/* ---- conv4_3 L2 normalisation (Model.py:206-209): y = x / sqrt(sum_c x^2) * gamma_c, no epsilon */
int ssd_l2norm_fwd(const float* x, const float* gamma, float* y, int M, int C, void* stream);
size_t ssd_l2norm_bwd_workspace(int M, int C);
int ssd_l2norm_bwd(const float* x, const float* gamma, const float* dy, float* dx, float* dgamma,
                   int M, int C, void* workspace, size_t workspace_bytes, void* stream);

/* ---- head output <-> (bs,P,4)/(bs,P,21) (Model.py:212-235 permute+view+cat) ------------
 * packed[m][0:4A] = loc channels, packed[m][4A:25A] = conf channels of pixel m=(n,h,w);
 * prior index = prior_off + (h*W+w)*A + a. */
int ssd_heads_scatter(const float* packed, int ld, float* loc, float* conf, int N, int HW, int A,
                      int prior_off, int P, int n_classes, void* stream);
int ssd_heads_gather(const float* dloc, const float* dconf, float* packed, int ld, int N, int HW, int A,
                     int prior_off, int P, int n_classes, void* stream);

/* ---- MultiBox loss (Losses.py:119-199 ssd + ssd1_; Util.py:57-63,98-102,252-301) -------
 * gt_boxes (n_gt,4) xyxy f32, gt_classes (n_gt) f32 values 0..19, img_start (bs+1) int32
 * prefix offsets into them (every image needs >= 1 box: checked by the caller on the host,
 * the reference raises there).  priors_cxcywh/priors_xyxy (P,4).
 * Outputs: losses[0]=loc_loss, losses[1]=conf_loss, losses[2]=n_pos (as float);
 *          obj (bs,P) int32 global GT index; cls (bs,P) int32 (bg_class = n_classes-1);
 *          dloc/dconf = d(loc_loss+conf_loss)/d(loc|conf) (may both be NULL: forward only).
 * norm_mode 0: reference normalisation (divide by batch n_pos).
 * norm_mode 1: un-normalised sums (loc: sum|d|/4, conf: sum CE) and gradients of those,
 *              for data-parallel runs that divide by the global n_pos after the all-reduce.
 * Three kernel launches (per-prior matching + cross entropy, wide; per-image forced matches + hard-negative selection; losses +
 * gradients, wide); ssd_tune_set_loss_form(0) selects the four-launch form of rounds 1-3 (cross-check). */
int ssd_tune_set_loss_form(int three_launch);
size_t ssd_multibox_loss_workspace(int bs, int P, int n_gt);
int ssd_multibox_loss(const float* loc, const float* conf, const float* gt_boxes, const float* gt_classes,
                      const int32_t* img_start, int bs, int n_gt, const float* priors_cxcywh,
                      const float* priors_xyxy, int P, int n_classes, float iou_threshold, int neg_pos_ratio,
                      int norm_mode, float* losses, int32_t* obj, int32_t* cls, float* dloc, float* dconf,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ---- decode + per-class NMS + top-k (Losses.py:11-98 inference; Util.py:86-96) ----------
 * l_ (P,4), c_ (P,n_classes) of ONE image.  Outputs (capacity top_k): boxes (top_k,4) xyxy
 * scaled by (img_w,img_h,img_w,img_h), classes int64, probs f32, prior_ids int32, count int32[1]. */
size_t ssd_decode_nms_workspace(int P, int n_classes);
int ssd_decode_nms(const float* l_, const float* c_, const float* priors_cxcywh, int P, int n_classes,
                   float min_score, float iou_threshold, int top_k, float img_w, float img_h,
                   float* boxes, int64_t* classes, float* probs, int32_t* prior_ids, int32_t* count,
                   void* workspace, size_t workspace_bytes, void* stream);

/* Batched form: l_ (B,P,4), c_ (B,P,n_classes), img_wh (B,2) DEVICE floats
 * (img_w, img_h per image); outputs (B,top_k,...) and count (B).  One launch set for the whole batch. */
size_t ssd_decode_nms_batch_workspace(int B, int P, int n_classes);
int ssd_decode_nms_batch(const float* l_, const float* c_, const float* priors_cxcywh, const float* img_wh, int B, int P,
                         int n_classes, float min_score, float iou_threshold, int top_k, float* boxes, int64_t* classes,
                         float* probs, int32_t* prior_ids, int32_t* count, void* workspace, size_t workspace_bytes, void* stream);

/* ---- mAP evaluator (Util.py:783-885 get_map: per-class 11-point interpolated AP, IoU > 0.5, one claim per
 * ground-truth box, no 'difficult' handling) over B images given as concatenated arrays:
 * det_* (D rows; det_start[B+1] = first row of each image), gt_* (G rows; gt_start[B+1]); classes are int32 in
 * [0, n_classes) (others are ignored, as the reference's `range(20)` loop does).  recall_levels is a HOST array
 * (the reference's torch.arange(0, 1.1, 0.1) as doubles), n_levels <= 16.  Outputs: tp (D bytes, must be zeroed by
 * the caller; 1 = true positive), table (n_classes x n_levels doubles: max precision at recall >= level, 0 where none;
 * AP[c] = mean of row c), counts (2*n_classes int32: detections per class, then ground truth per class). */
size_t ssd_map_eval_workspace(int D, int G);
int ssd_map_eval(const float* det_boxes, const int32_t* det_classes, const float* det_scores, const int32_t* det_start,
                 int D, const float* gt_boxes, const int32_t* gt_classes, const int32_t* gt_start, int G, int B,
                 int n_classes, const double* recall_levels_host, int n_levels, uint8_t* tp, double* table,
                 int32_t* counts, void* workspace, size_t workspace_bytes, void* stream);

/* ---- input pipeline (Dataset.py:10-13,24-39 Resize((300,300)) + ToTensor + Normalize; Util.py:610-749 expand /
 * random_crop / flip as geometry) on 8-bit HWC RGB images packed in one device arena.  The resize reproduces
 * Pillow's Image.resize(BILINEAR) on 8-bit images bit for bit.  Geometry per image, in this order: the source is
 * placed at (place_top, place_left) on a canvas_h x canvas_w canvas of `filler` (no expand: canvas = source, place
 * 0,0); the window (crop_top, crop_left, crop_h, crop_w) of the canvas is taken (no crop: the whole canvas); its
 * columns are mirrored when flip != 0; the result is resized to out_h x out_w and normalised into out_nchw
 * (B,3,out_h,out_w) float32.  descs_dev / descs_host are the same B descriptors in device / host memory. */
typedef struct ssd_image_desc {
    int64_t src_offset;            /* byte offset of the image's first pixel in the arena */
    int32_t src_h, src_w;
    int32_t canvas_h, canvas_w, place_top, place_left;
    int32_t crop_top, crop_left, crop_h, crop_w;
    int32_t flip, reserved;
} ssd_image_desc;
/* photometric_distort (Util.py:752-780) on the SOURCE images of the arena, in place, before ssd_preprocess_u8: up to four
 * ops per image in the image's own order; kind 0 brightness, 1 contrast, 2 saturation (alpha = the enhancement factor as
 * float32, Pillow's Image.blend argument), 3 hue (hue_delta = int(hue_factor * 255) & 255).  Bit-identical to the Pillow
 * arithmetic torchvision's PIL back end runs.  Host and device copies of both descriptor arrays, as for the resize. */
typedef struct ssd_photo_desc {
    int32_t n_ops;
    int32_t kind[4];
    float alpha[4];
    int32_t hue_delta[4];
} ssd_photo_desc;
size_t ssd_photometric_workspace(int B);
int ssd_photometric_u8(uint8_t* arena, const ssd_image_desc* descs_dev, const ssd_image_desc* descs_host,
                       const ssd_photo_desc* photo_dev, const ssd_photo_desc* photo_host, int B, void* workspace,
                       size_t workspace_bytes, void* stream);
size_t ssd_preprocess_workspace(const ssd_image_desc* descs_host, int B, int out_h, int out_w);
int ssd_preprocess_u8(const uint8_t* arena, const ssd_image_desc* descs_dev, const ssd_image_desc* descs_host, int B,
                      int out_h, int out_w, const float* mean3_host, const float* std3_host, const uint8_t* filler3_host,
                      float* out_nchw, void* workspace, size_t workspace_bytes, void* stream);

/* ---- diagnostic: out32[2*xcc] = shader-clock counter, out32[2*xcc+1] = 100 MHz wall clock, for each of the (up to 16)
 * XCCs a block landed on; slots must be zeroed by the caller.  Two probes bracket a region: average shader MHz =
 * 100 * d(counter) / d(wall clock) per XCC. */
int ssd_clock_probe(uint64_t* out32, void* stream);

/* ---- diagnostic (host only): node count of a captured hipGraph_t and how many of the nodes are kernel launches -- what one replay of
 * the captured train step (ddp.GraphedTrainStep; the loop of train_function.py:80-95) stands for. */
int ssd_graph_node_counts(void* graph, int* kernel_nodes, int* total_nodes);

/* ---- fused SGD (train.py:53-55: momentum .9, weight decay 5e-4; bias lr 2x) on a flat buffer;
 * grad_scale multiplies the gradient first (1/n_pos_global in data-parallel runs).  Per element, each line rounded once:
 *   g' = fma(weight_decay, p, g * scale);  buf = first_step ? g' : (momentum * buf) + g';  p = fma(-lr, buf, p)
 * which is torch.optim.SGD's own rounding sequence, so the two stay bit-identical.  momentum == 0: call with first_step = 1. */
int ssd_sgd_momentum(float* param, const float* grad, float* momentum_buf, size_t n, float lr, float momentum,
                     float weight_decay, const float* grad_scale_dev, int first_step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SSD_GFX950_H */
