// Input pipeline on the device: 8-bit HWC images of any size ->
// expand canvas / crop window / horizontal flip (Util.py:610-749, as index transforms of the fetch) ->
// Resize((300,300)) (Dataset.py:10: PIL bilinear with antialiasing, reproduced bit for bit: double-precision triangle
// weights normalised by their running sum, 22-bit fixed point, horizontal pass to 8 bits, vertical pass to 8 bits) ->
// ToTensor (/255) -> Normalize ((x - mean) / std) -> NCHW float32 (Dataset.py:11-12,37).
//
//   P1 coefficients  thread per (image, axis, output index): window start, tap count, fixed-point weights
//   P2 horizontal    thread per (image, input row, output column): 3 channels, 8-bit result into the scratch rows
//   P3 vertical      thread per (image, output row, output column): 8-bit result -> normalised float planes
// Byte work, HBM-bound: the 1.08 MB float output per image dominates the traffic.
#include "common.h"
#pragma clang fp contract(off)

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;
constexpr int MAX_KSIZE = 63;

struct PreArgs {
    const uint8_t* arena;
    const ssd_image_desc* desc;      // device copy
    int B, out_h, out_w, out_max, ks, max_in_h;
    int32_t* bounds;                 // [B][2][out_max][2]  (start, count)
    int32_t* coef;                   // [B][2][out_max][ks]
    uint8_t* tmp;                    // [B][max_in_h][out_w][3]
    float* out;                      // [B][3][out_h][out_w]
    float mean[3], stdv[3];
    uint8_t filler[4];
};

__global__ void coef_kernel(const PreArgs a) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.out_w + a.out_h) return;
    const int axis = t < a.out_w ? 0 : 1;                       // 0 = horizontal (columns), 1 = vertical (rows)
    const int o = axis == 0 ? t : t - a.out_w;
    const ssd_image_desc d = a.desc[b];
    const int in_size = axis == 0 ? d.crop_w : d.crop_h, out_size = axis == 0 ? a.out_w : a.out_h;
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale > 1.0 ? scale : 1.0;
    const double support = 1.0 * filterscale;
    const double ss = 1.0 / filterscale;
    const double center = ((double)o + 0.5) * scale;
    int lo = (int)(center - support + 0.5);
    if (lo < 0) lo = 0;
    int hi = (int)(center + support + 0.5);
    if (hi > in_size) hi = in_size;
    const int n = hi - lo;
    int32_t* kk = a.coef + (((size_t)b * 2 + axis) * a.out_max + o) * a.ks;
    double ww = 0.0;
    for (int x = 0; x < n && x < a.ks; ++x) {
        double v = ((double)(x + lo) - center + 0.5) * ss;
        v = v < 0.0 ? -v : v;
        ww += v < 1.0 ? 1.0 - v : 0.0;
    }
    for (int x = 0; x < a.ks; ++x) {
        double w = 0.0;
        if (x < n) {
            double v = ((double)(x + lo) - center + 0.5) * ss;
            v = v < 0.0 ? -v : v;
            w = v < 1.0 ? 1.0 - v : 0.0;
            if (ww != 0.0) w /= ww;
        }
        kk[x] = w < 0.0 ? (int)(w * (double)(1 << PRECISION_BITS) - 0.5) : (int)(w * (double)(1 << PRECISION_BITS) + 0.5);
    }
    int32_t* bd = a.bounds + (((size_t)b * 2 + axis) * a.out_max + o) * 2;
    bd[0] = lo;
    bd[1] = n < a.ks ? n : a.ks;
}

__device__ __forceinline__ int clip8(int v) {
    v >>= PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__global__ __launch_bounds__(256) void horizontal_kernel(const PreArgs a) {
    const int b = blockIdx.y;
    const ssd_image_desc d = a.desc[b];
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int y = (int)(idx / a.out_w), o = (int)(idx % a.out_w);
    if (y >= d.crop_h) return;
    const int32_t* bd = a.bounds + (((size_t)b * 2 + 0) * a.out_max + o) * 2;
    const int32_t* kk = a.coef + (((size_t)b * 2 + 0) * a.out_max + o) * a.ks;
    const int lo = bd[0], n = bd[1];
    const int sy = d.crop_top + y - d.place_top;
    const bool row_in = sy >= 0 && sy < d.src_h;
    const uint8_t* row = a.arena + d.src_offset + (size_t)(row_in ? sy : 0) * d.src_w * 3;
    int acc0 = 1 << (PRECISION_BITS - 1), acc1 = acc0, acc2 = acc0;
    for (int x = 0; x < n; ++x) {
        const int xx = d.flip ? d.crop_w - 1 - (lo + x) : lo + x;
        const int sx = d.crop_left + xx - d.place_left;
        int p0 = a.filler[0], p1 = a.filler[1], p2 = a.filler[2];
        if (row_in && sx >= 0 && sx < d.src_w) {
            const uint8_t* px = row + (size_t)sx * 3;
            p0 = px[0]; p1 = px[1]; p2 = px[2];
        }
        const int k = kk[x];
        acc0 += p0 * k; acc1 += p1 * k; acc2 += p2 * k;
    }
    uint8_t* t = a.tmp + (((size_t)b * a.max_in_h + y) * a.out_w + o) * 3;
    t[0] = (uint8_t)clip8(acc0); t[1] = (uint8_t)clip8(acc1); t[2] = (uint8_t)clip8(acc2);
}

__global__ __launch_bounds__(256) void vertical_kernel(const PreArgs a) {
    const int b = blockIdx.y;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)a.out_h * a.out_w) return;
    const int oy = (int)(idx / a.out_w), ox = (int)(idx % a.out_w);
    const int32_t* bd = a.bounds + (((size_t)b * 2 + 1) * a.out_max + oy) * 2;
    const int32_t* kk = a.coef + (((size_t)b * 2 + 1) * a.out_max + oy) * a.ks;
    const int lo = bd[0], n = bd[1];
    int acc0 = 1 << (PRECISION_BITS - 1), acc1 = acc0, acc2 = acc0;
    const uint8_t* t = a.tmp + (((size_t)b * a.max_in_h + lo) * a.out_w + ox) * 3;
    for (int y = 0; y < n; ++y) {
        const int k = kk[y];
        acc0 += t[0] * k; acc1 += t[1] * k; acc2 += t[2] * k;
        t += (size_t)a.out_w * 3;
    }
    const size_t plane = (size_t)a.out_h * a.out_w;
    float* o = a.out + (size_t)b * 3 * plane + idx;
    o[0] = ((float)clip8(acc0) / 255.0f - a.mean[0]) / a.stdv[0];
    o[plane] = ((float)clip8(acc1) / 255.0f - a.mean[1]) / a.stdv[1];
    o[2 * plane] = ((float)clip8(acc2) / 255.0f - a.mean[2]) / a.stdv[2];
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

int plan(const ssd_image_desc* d, int B, int out_h, int out_w, int* ks, int* max_in_h) {
    *ks = 3; *max_in_h = 1;
    for (int b = 0; b < B; ++b) {
        const ssd_image_desc& e = d[b];
        if (e.src_h <= 0 || e.src_w <= 0 || e.src_offset < 0 || e.canvas_h < e.src_h || e.canvas_w < e.src_w || e.place_top < 0 ||
            e.place_left < 0 || e.place_top + e.src_h > e.canvas_h || e.place_left + e.src_w > e.canvas_w || e.crop_h <= 0 ||
            e.crop_w <= 0 || e.crop_top < 0 || e.crop_left < 0 || e.crop_top + e.crop_h > e.canvas_h ||
            e.crop_left + e.crop_w > e.canvas_w)
            return SSD_ERR_BAD_SHAPE;
        for (int axis = 0; axis < 2; ++axis) {
            const double scale = (double)(axis ? e.crop_h : e.crop_w) / (double)(axis ? out_h : out_w);
            const double support = scale > 1.0 ? scale : 1.0;
            int k = (int)support;
            if ((double)k < support) ++k;                       // ceil
            k = k * 2 + 1;
            if (k > *ks) *ks = k;
        }
        if (e.crop_h > *max_in_h) *max_in_h = e.crop_h;
    }
    return *ks > MAX_KSIZE ? SSD_ERR_BAD_SHAPE : SSD_OK;
}

}  // namespace

extern "C" size_t ssd_preprocess_workspace(const ssd_image_desc* descs_host, int B, int out_h, int out_w) {
    int ks, mh;
    if (!descs_host || B <= 0 || out_h <= 0 || out_w <= 0 || plan(descs_host, B, out_h, out_w, &ks, &mh) != SSD_OK) return 0;
    const int om = out_h > out_w ? out_h : out_w;
    return align256((size_t)B * 2 * om * 2 * 4) + align256((size_t)B * 2 * om * ks * 4) + align256((size_t)B * mh * out_w * 3);
}

extern "C" int ssd_preprocess_u8(const uint8_t* arena, const ssd_image_desc* descs_dev, const ssd_image_desc* descs_host, int B,
                                 int out_h, int out_w, const float* mean3_host, const float* std3_host, const uint8_t* filler3_host,
                                 float* out_nchw, void* workspace, size_t workspace_bytes, void* stream) {
    if (!arena || !descs_dev || !descs_host || !mean3_host || !std3_host || !filler3_host || !out_nchw || !workspace) return SSD_ERR_NULL;
    if (B <= 0 || B > 65535 || out_h <= 0 || out_w <= 0) return SSD_ERR_BAD_SHAPE;
    PreArgs a{};
    if (int e = plan(descs_host, B, out_h, out_w, &a.ks, &a.max_in_h)) return e;
    if (workspace_bytes < ssd_preprocess_workspace(descs_host, B, out_h, out_w)) return SSD_ERR_WORKSPACE;
    if (!ssd_aligned16(workspace)) return SSD_ERR_ALIGN;
    a.arena = arena; a.desc = descs_dev; a.B = B; a.out_h = out_h; a.out_w = out_w; a.out_max = out_h > out_w ? out_h : out_w;
    char* w = static_cast<char*>(workspace);
    a.bounds = reinterpret_cast<int32_t*>(w); w += align256((size_t)B * 2 * a.out_max * 2 * 4);
    a.coef = reinterpret_cast<int32_t*>(w); w += align256((size_t)B * 2 * a.out_max * a.ks * 4);
    a.tmp = reinterpret_cast<uint8_t*>(w);
    a.out = out_nchw;
    for (int c = 0; c < 3; ++c) { a.mean[c] = mean3_host[c]; a.stdv[c] = std3_host[c]; a.filler[c] = filler3_host[c]; }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(coef_kernel, dim3(ssd_cdiv(out_w + out_h, 64), B), dim3(64), 0, st, a);
    SSD_CHECK_LAUNCH();
    const long n1 = (long)a.max_in_h * out_w;
    hipLaunchKernelGGL(horizontal_kernel, dim3((unsigned)((n1 + 255) / 256), B), dim3(256), 0, st, a);
    SSD_CHECK_LAUNCH();
    const long n2 = (long)out_h * out_w;
    hipLaunchKernelGGL(vertical_kernel, dim3((unsigned)((n2 + 255) / 256), B), dim3(256), 0, st, a);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}
