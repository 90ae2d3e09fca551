// photometric_distort on the device (Util.py:752-780): brightness / contrast / saturation / hue on the 8-bit source images
// in the arena, in place, in each image's own drawn order.  The arithmetic is Pillow's, bit for bit (the reference reaches
// it through torchvision's PIL back end): Image.blend in float32 with truncation (clipped outside [0,1]); L =
// (19595 R + 38470 G + 7471 B + 0x8000) >> 16; contrast pivots on int(mean(L) + .5) of the CURRENT image (so each
// stage that holds a contrast op is preceded by an exact integer sum over the image); hue = Pillow's 8-bit RGB -> HSV,
// H += delta (mod 256), HSV -> RGB with its float / double mix and C round().
#include "common.h"
#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ int lum(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

__device__ __forceinline__ int blend1(int in1, int in2, float alpha, bool inside) {
    const float t = (float)in1 + alpha * (float)(in2 - in1);
    if (inside) return (int)t & 255;                       // interpolation: in range by construction, truncation
    return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
}

__device__ __forceinline__ void hue_shift(int& r, int& g, int& b, int delta) {
    // RGB -> HSV
    const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
    int uh = 0, us = 0;
    const int uv = maxc;
    if (minc != maxc) {
        const float cr = (float)(maxc - minc);
        const float s = cr / (float)maxc;
        const float rc = (float)(maxc - r) / cr, gc = (float)(maxc - g) / cr, bc = (float)(maxc - b) / cr;
        float h;
        if (r == maxc) h = bc - gc;
        else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
        else h = (float)(4.0 + (double)gc - (double)rc);
        const float hh = (float)fmod((double)h / 6.0 + 1.0, 1.0);
        uh = min(255, max(0, (int)((double)hh * 255.0)));
        us = min(255, max(0, (int)((double)s * 255.0)));
    }
    uh = (uh + delta) & 255;
    // HSV -> RGB
    if (us == 0) { r = g = b = uv; return; }
    const double hf = (double)(float)uh * 6.0 / 255.0;
    const double fi = floor(hf);
    const double f = (double)(float)(hf - fi);
    const double fs = (double)(float)((double)us / 255.0);
    const double vf = (double)uv;
    const int p = min(255, max(0, (int)floor(vf * (1.0 - fs) + 0.5)));
    const int q = min(255, max(0, (int)floor(vf * (1.0 - fs * f) + 0.5)));
    const int t = min(255, max(0, (int)floor(vf * (1.0 - fs * (1.0 - f)) + 0.5)));
    switch ((int)fi % 6) {
        case 0: r = uv; g = t; b = p; break;
        case 1: r = q; g = uv; b = p; break;
        case 2: r = p; g = uv; b = t; break;
        case 3: r = p; g = q; b = uv; break;
        case 4: r = t; g = p; b = uv; break;
        default: r = uv; g = p; b = q; break;
    }
}

// exact integer sum of L over every image whose op at `stage` is a contrast (kind 1)
__global__ __launch_bounds__(256) void photo_sum_kernel(const uint8_t* __restrict__ arena, const ssd_image_desc* __restrict__ desc,
                                                        const ssd_photo_desc* __restrict__ photo, int stage,
                                                        unsigned long long* __restrict__ sums) {
    __shared__ unsigned long long part[4];
    const int b = blockIdx.y;
    const ssd_photo_desc ph = photo[b];
    if (stage >= ph.n_ops || ph.kind[stage] != 1) return;          // uniform per block
    const ssd_image_desc d = desc[b];
    const uint8_t* px = arena + d.src_offset;
    const long n = (long)d.src_h * d.src_w;
    unsigned long long s = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += (unsigned)lum(px[3 * i], px[3 * i + 1], px[3 * i + 2]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)s, o, 64), hi = __shfl_xor((unsigned)(s >> 32), o, 64);
        s += ((unsigned long long)hi << 32) | lo;
    }
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&sums[b * 4 + stage], part[0] + part[1] + part[2] + part[3]);      // integer: order-independent
}

__global__ __launch_bounds__(256) void photo_apply_kernel(uint8_t* __restrict__ arena, const ssd_image_desc* __restrict__ desc,
                                                          const ssd_photo_desc* __restrict__ photo, int stage,
                                                          const unsigned long long* __restrict__ sums) {
    const int b = blockIdx.y;
    const ssd_photo_desc ph = photo[b];
    if (stage >= ph.n_ops) return;
    const ssd_image_desc d = desc[b];
    uint8_t* px = arena + d.src_offset;
    const long n = (long)d.src_h * d.src_w;
    const int kind = ph.kind[stage];
    const float alpha = ph.alpha[stage];
    const bool inside = alpha >= 0.f && alpha <= 1.f;
    int mean = 0;
    if (kind == 1) mean = (int)((double)sums[b * 4 + stage] / (double)n + 0.5);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        int r = px[3 * i], g = px[3 * i + 1], bl = px[3 * i + 2];
        if (kind == 3) {
            hue_shift(r, g, bl, ph.hue_delta[stage]);
        } else {
            int d0 = 0, d1 = 0, d2 = 0;                              // brightness: black
            if (kind == 1) d0 = d1 = d2 = mean;
            else if (kind == 2) d0 = d1 = d2 = lum(r, g, bl);
            r = blend1(d0, r, alpha, inside); g = blend1(d1, g, alpha, inside); bl = blend1(d2, bl, alpha, inside);
        }
        px[3 * i] = (uint8_t)r; px[3 * i + 1] = (uint8_t)g; px[3 * i + 2] = (uint8_t)bl;
    }
}

__global__ void photo_zero_kernel(unsigned long long* sums, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sums[i] = 0ull;
}

}  // namespace

extern "C" size_t ssd_photometric_workspace(int B) { return B > 0 ? (size_t)B * 4 * sizeof(unsigned long long) + 256 : 0; }

extern "C" int ssd_photometric_u8(uint8_t* arena, const ssd_image_desc* descs_dev, const ssd_image_desc* descs_host,
                                  const ssd_photo_desc* photo_dev, const ssd_photo_desc* photo_host, int B, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    if (!arena || !descs_dev || !descs_host || !photo_dev || !photo_host || !workspace) return SSD_ERR_NULL;
    if (B <= 0 || B > 65535) return SSD_ERR_BAD_SHAPE;
    if (workspace_bytes < ssd_photometric_workspace(B)) return SSD_ERR_WORKSPACE;
    if (!ssd_aligned16(workspace)) return SSD_ERR_ALIGN;
    int max_ops = 0;
    long max_px = 1;
    for (int b = 0; b < B; ++b) {
        const ssd_photo_desc& ph = photo_host[b];
        if (ph.n_ops < 0 || ph.n_ops > 4) return SSD_ERR_BAD_SHAPE;
        for (int k = 0; k < ph.n_ops; ++k)
            if (ph.kind[k] < 0 || ph.kind[k] > 3) return SSD_ERR_BAD_SHAPE;
        if (descs_host[b].src_h <= 0 || descs_host[b].src_w <= 0 || descs_host[b].src_offset < 0) return SSD_ERR_BAD_SHAPE;
        if (ph.n_ops > max_ops) max_ops = ph.n_ops;
        const long n = (long)descs_host[b].src_h * descs_host[b].src_w;
        if (n > max_px) max_px = n;
    }
    if (max_ops == 0) return SSD_OK;
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* sums = static_cast<unsigned long long*>(workspace);
    hipLaunchKernelGGL(photo_zero_kernel, dim3(ssd_cdiv(B * 4, 256)), dim3(256), 0, st, sums, B * 4);
    SSD_CHECK_LAUNCH();
    const int blocks = (int)((max_px + 1023) / 1024 > 256 ? 256 : (max_px + 1023) / 1024);
    for (int stage = 0; stage < max_ops; ++stage) {
        bool any_contrast = false;
        for (int b = 0; b < B; ++b) any_contrast |= stage < photo_host[b].n_ops && photo_host[b].kind[stage] == 1;
        if (any_contrast) {
            hipLaunchKernelGGL(photo_sum_kernel, dim3(blocks, B), dim3(256), 0, st, arena, descs_dev, photo_dev, stage, sums);
            SSD_CHECK_LAUNCH();
        }
        hipLaunchKernelGGL(photo_apply_kernel, dim3(blocks, B), dim3(256), 0, st, arena, descs_dev, photo_dev, stage, sums);
        SSD_CHECK_LAUNCH();
    }
    return SSD_OK;
}
