// MultiBox matching + loss + gradients (Losses.py:119-199; Util.py:57-63,98-102,252-301).
//
// THREE launches (round 4; no host round trip, no IoU matrix in memory):
//   L1 loss_prior_kernel   wide, a wave per 64 consecutive priors of an image: IoU against the image's boxes (kept in LDS) -> the
//                          prior's best box (first index on ties) AND, per box, the wave's best prior (torch.max's order: NaN above
//                          numbers, first index on ties; Util.py:252-301, Losses.py:160-163); log-softmax CE against the background
//                          class from conf rows staged through LDS by coalesced loads (a lane reading its own 84-byte row makes every
//                          load instruction touch 42 lines)
//   L2 loss_image_kernel   one workgroup per IMAGE, what needs the whole image and little arithmetic: best prior of each box over
//                          L1's per-wave results, forced matches in GT order (last wins, Losses.py:164-167), threshold -> class, CE
//                          and L1 term of the (few) positives, then the k-th largest negative CE by radix select on the float bits
//                          (CE >= 0) over the image's values in LDS -- bins scanned by one wave, histogram atomics aggregated per
//                          wave (the top bytes of CE values are nearly all equal) --, sum of the top k, selection mask (lower index
//                          first among equal values)
//   L3 finalize_kernel     losses and d(loss)/d(loc|conf), wide; conf rows are read for the selected priors only and dconf leaves
//                          through LDS as whole lines
// One workgroup per image for ALL of it was built and measured first (two launches): 124 us against 107 for K1-K3 -- 8 732 priors x ~1 000
// instructions of IoU divisions and exponentials are VALU-bound on ONE CU (57 us) whatever the memory side does; so the arithmetic
// stays wide and only the reductions are per image.
// The FOUR-launch form of rounds 1-3 (K1 best_prior_per_gt, K2 match_ce, K3 hard_negative, then L3) stays as the in-library
// cross-check (ssd_tune_set_loss_form(0); tests/test_gpu_kernels.py compares the two forms: obj / cls / selection identical, sums
// to rounding).
// All sums are reduced in a fixed order (no float atomics): results are bitwise reproducible.
//
// The IoU arithmetic must equal the reference's f32 op sequence bit for bit
// (sub, max, min, mul, add, sub, IEEE divide): this file is compiled with
// -ffp-contract=off and the pragma below keeps a*b+c from being fused.
#include "common.h"
#pragma clang fp contract(off)

namespace {

constexpr int LB = 256;       // threads per block in K2/K4
constexpr int HB = 1024;      // threads per block in K3

__device__ __forceinline__ float iou_xyxy(float ax1, float ay1, float ax2, float ay2, float area_a, float bx1, float by1,
                                          float bx2, float by2, float area_b) {
    const float lx = fmaxf(ax1, bx1), ly = fmaxf(ay1, by1);
    const float hx = fminf(ax2, bx2), hy = fminf(ay2, by2);
    const float dx = fmaxf(hx - lx, 0.f), dy = fmaxf(hy - ly, 0.f);
    const float inter = dx * dy;
    const float uni = (area_a + area_b) - inter;          // Util.py:299: a1 + a2 - inter, left to right
    return inter / uni;                                   // no epsilon (Util.py:301)
}

// torch.max(dim) update rule: take v when it is larger, or when it is NaN and best is not.
__device__ __forceinline__ bool better(float v, float best) { return v > best || (v != v && best == best); }

__global__ __launch_bounds__(256) void best_prior_per_gt_kernel(const float* __restrict__ gt, const float* __restrict__ pri_xyxy,
                                                                int P, int32_t* __restrict__ best_prior) {
    const int k = blockIdx.x;
    const float ax1 = gt[k * 4 + 0], ay1 = gt[k * 4 + 1], ax2 = gt[k * 4 + 2], ay2 = gt[k * 4 + 3];
    const float area_a = (ax2 - ax1) * (ay2 - ay1);
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int p = threadIdx.x; p < P; p += 256) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(pri_xyxy + (size_t)p * 4);
        const float area_b = (b[2] - b[0]) * (b[3] - b[1]);
        const float v = iou_xyxy(ax1, ay1, ax2, ay2, area_a, b[0], b[1], b[2], b[3], area_b);
        if (idx == 0x7fffffff || better(v, best)) { best = v; idx = p; }
    }
    // combine: larger value wins, NaN beats numbers, equal values -> smaller index
    __shared__ float sv[256];
    __shared__ int si[256];
    sv[threadIdx.x] = best;
    si[threadIdx.x] = idx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            const float v2 = sv[threadIdx.x + o];
            const int i2 = si[threadIdx.x + o];
            const float v1 = sv[threadIdx.x];
            const int i1 = si[threadIdx.x];
            bool take;
            if (i2 == 0x7fffffff) take = false;
            else if (i1 == 0x7fffffff) take = true;
            else if (v1 != v1 || v2 != v2) take = (v2 != v2) && (v1 == v1 || i2 < i1);
            else take = v2 > v1 || (v2 == v1 && i2 < i1);
            if (take) { sv[threadIdx.x] = v2; si[threadIdx.x] = i2; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) best_prior[k] = si[0];
}

struct MatchArgs {
    const float* loc; const float* conf; const float* gt; const float* gt_cls; const int32_t* img_start;
    const int32_t* best_prior; const float* pri; const float* pri_xyxy;
    int32_t* obj; int32_t* cls; float* neg; float* part;
    int P, C, NB; float thr;
};

// encode of the matched GT against the prior (Util.py:57-63 xyxy_to_xywh then Util.py:98-102)
__device__ __forceinline__ void encode_gt(const float* __restrict__ gt, int k, const f32x4 pr, float g[4]) {
    const float x1 = gt[k * 4 + 0], y1 = gt[k * 4 + 1], x2 = gt[k * 4 + 2], y2 = gt[k * 4 + 3];
    const float cx = (x2 + x1) / 2.f, cy = (y2 + y1) / 2.f, w = x2 - x1, h = y2 - y1;
    g[0] = (cx - pr[0]) / (pr[2] / 10.f);
    g[1] = (cy - pr[1]) / (pr[3] / 10.f);
    g[2] = logf(w / pr[2]) * 5.f;
    g[3] = logf(h / pr[3]) * 5.f;
}

__device__ __forceinline__ float block_sum_256(float v, float* sm) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

__global__ __launch_bounds__(LB) void match_ce_kernel(const MatchArgs a) {
    __shared__ float red[4];
    const int i = blockIdx.y;
    const int p = blockIdx.x * LB + threadIdx.x;
    const int s = a.img_start[i], e = a.img_start[i + 1];
    float my_pos = 0.f, my_ce = 0.f, my_l1 = 0.f;
    if (p < a.P) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(a.pri_xyxy + (size_t)p * 4);
        const float area_b = (b[2] - b[0]) * (b[3] - b[1]);
        float best = 0.f;
        int idx = s;
        for (int k = s; k < e; ++k) {
            const float ax1 = a.gt[k * 4 + 0], ay1 = a.gt[k * 4 + 1], ax2 = a.gt[k * 4 + 2], ay2 = a.gt[k * 4 + 3];
            const float area_a = (ax2 - ax1) * (ay2 - ay1);
            const float v = iou_xyxy(ax1, ay1, ax2, ay2, area_a, b[0], b[1], b[2], b[3], area_b);
            if (k == s || better(v, best)) { best = v; idx = k; }
        }
        for (int k = s; k < e; ++k)            // Losses.py:164-167: sequential writes, last GT wins
            if (a.best_prior[k] == p) { idx = k; best = 1.f; }
        const int bg = a.C - 1;
        const int c = best < a.thr ? bg : (int)a.gt_cls[idx];
        const bool pos = c != bg;
        const size_t ip = (size_t)i * a.P + p;
        a.obj[ip] = idx;
        a.cls[ip] = c;
        // cross entropy, torch log_softmax order: (x - max) - log(sum exp(x - max))
        const float* x = a.conf + ip * a.C;
        float m = x[0];
        for (int q = 1; q < a.C; ++q) m = fmaxf(m, x[q]);
        float se = 0.f;
        for (int q = 0; q < a.C; ++q) se += expf(x[q] - m);
        const float ce = -((x[c] - m) - logf(se));
        a.neg[ip] = pos ? 0.f : ce;
        if (pos) {
            my_pos = 1.f;
            my_ce = ce;
            const f32x4 pr = *reinterpret_cast<const f32x4*>(a.pri + (size_t)p * 4);
            float g[4];
            encode_gt(a.gt, idx, pr, g);
            const f32x4 l = *reinterpret_cast<const f32x4*>(a.loc + ip * 4);
            my_l1 = (fabsf(l[0] - g[0]) + fabsf(l[1] - g[1])) + (fabsf(l[2] - g[2]) + fabsf(l[3] - g[3]));
        }
    }
    const float t_pos = block_sum_256(my_pos, red);
    const float t_ce = block_sum_256(my_ce, red);
    const float t_l1 = block_sum_256(my_l1, red);
    if (threadIdx.x == 0) {
        float* o = a.part + ((size_t)i * a.NB + blockIdx.x) * 3;
        o[0] = t_pos; o[1] = t_ce; o[2] = t_l1;
    }
}

// ---- K3 --------------------------------------------------------------------------------
__global__ __launch_bounds__(HB) void hard_negative_kernel(const float* __restrict__ neg, const float* __restrict__ part,
                                                           uint8_t* __restrict__ sel, float* __restrict__ stats, int P,
                                                           int NB, int ratio) {
    extern __shared__ __attribute__((aligned(16))) uint32_t vals[];      // P values of this image
    __shared__ int hist[256];
    __shared__ float fred[HB / 64];
    __shared__ int ired[HB / 64];
    __shared__ int s_bcast[4];
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // per-image totals from K2's per-block partials (fixed order)
    if (tid == 0) {
        float npos = 0.f, ce = 0.f, l1 = 0.f;
        for (int b = 0; b < NB; ++b) {
            const float* q = part + ((size_t)i * NB + b) * 3;
            npos += q[0]; ce += q[1]; l1 += q[2];
        }
        stats[i * 4 + 0] = npos; stats[i * 4 + 1] = ce; stats[i * 4 + 2] = l1;
        s_bcast[0] = (int)npos;
    }
    for (int p = tid; p < P; p += HB) {
        const float v = neg[(size_t)i * P + p];
        vals[p] = __float_as_uint(v > 0.f ? v : 0.f);       // also maps -0.0 and NaN to +0.0
    }
    __syncthreads();
    long kk = (long)ratio * s_bcast[0];
    const int k = kk > P ? P : (int)kk;
    if (k <= 0) {
        for (int p = tid; p < P; p += HB) sel[(size_t)i * P + p] = 0;
        if (tid == 0) stats[i * 4 + 3] = 0.f;
        return;
    }
    // radix select of the k-th largest bit pattern, 8 bits per pass from the top
    uint32_t prefix = 0, mask = 0;
    int remaining = k;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int b = tid; b < 256; b += HB) hist[b] = 0;
        __syncthreads();
        for (int p = tid; p < P; p += HB) {
            const uint32_t u = vals[p];
            if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1);
        }
        __syncthreads();
        if (tid == 0) {
            int rem = remaining, d = 255;
            for (; d > 0; --d) {
                if (hist[d] >= rem) break;
                rem -= hist[d];
            }
            s_bcast[1] = d;
            s_bcast[2] = rem;
        }
        __syncthreads();
        prefix |= (uint32_t)s_bcast[1] << shift;
        mask |= 255u << shift;
        remaining = s_bcast[2];
        __syncthreads();
    }
    const uint32_t T = prefix;            // k-th largest value; take `remaining` of the elements equal to it
    // ordered pass: thread t owns the contiguous slice [t*CH, (t+1)*CH)
    const int CH = (P + HB - 1) / HB;
    const int p0 = tid * CH, p1 = min(P, p0 + CH);
    float sum_gt = 0.f;
    int n_eq = 0;
    for (int p = p0; p < p1; ++p) {
        const uint32_t u = vals[p];
        if (u > T) sum_gt += __uint_as_float(u);
        n_eq += (u == T);
    }
    // exclusive prefix of n_eq over threads (wave scan + scan of wave totals)
    int incl = n_eq;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) ired[wave] = incl;
    const float wsum = wave_sum(sum_gt);
    if (lane == 0) fred[wave] = wsum;
    __syncthreads();
    int wave_off = 0;
    for (int w = 0; w < wave; ++w) wave_off += ired[w];
    int rank = wave_off + incl - n_eq;     // number of equal elements before this thread's slice
    for (int p = p0; p < p1; ++p) {
        const uint32_t u = vals[p];
        uint8_t s = 0;
        if (u > T) s = 1;
        else if (u == T) { s = rank < remaining ? 1 : 0; ++rank; }
        sel[(size_t)i * P + p] = s;
    }
    if (tid == 0) {
        float tot = 0.f;
        for (int w = 0; w < HB / 64; ++w) tot += fred[w];
        stats[i * 4 + 3] = tot + (float)remaining * __uint_as_float(T);
    }
}


// ---- L1: one workgroup per image ----------------------------------------------------------------------------------------
// (value, index) under torch.max's order: NaN above every number, equal values -> smaller index; 0x7fffffff = nothing yet
__device__ __forceinline__ bool arg_takes(float v2, int i2, float v1, int i1) {
    if (i2 == 0x7fffffff) return false;
    if (i1 == 0x7fffffff) return true;
    if (v1 != v1 || v2 != v2) return (v2 != v2) && (v1 == v1 || i2 < i1);
    return v2 > v1 || (v2 == v1 && i2 < i1);
}

struct ImageArgs {
    const float* loc; const float* conf; const float* gt; const float* gt_cls; const int32_t* img_start;
    const float* pri; const float* pri_xyxy;
    float* partv; int32_t* parti;                     // [n_gt][NPW]: per GT box, the best prior of each 64-prior wave of L1
    float* bestv; float* cebg;                        // [bs][P]: IoU of the prior's best box (unforced); CE against the class that box gives it
    int32_t* best_prior; int32_t* obj; int32_t* cls; uint8_t* sel; float* stats;
    int P, C, ratio, NPW; float thr;
};

constexpr int SG = 128;        // GT boxes of an image kept in LDS (more are read from memory: correct, slower)

// L1: grid (ceil(P / 256), bs), 256 threads; a wave owns 64 consecutive priors of image blockIdx.y
__global__ __launch_bounds__(256) void loss_prior_kernel(const ImageArgs a) {
    extern __shared__ __attribute__((aligned(16))) float stage_all[];       // [4][64 * C] conf rows
    __shared__ f32x4 sg_box[SG];
    __shared__ float sg_area[SG];
    __shared__ int sg_cls[SG];
    const int i = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = a.img_start[i], e = a.img_start[i + 1];
    const int P = a.P, C = a.C;
    const int ns = e - s < SG ? e - s : SG;
    for (int t = tid; t < ns; t += 256) {
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(a.gt + (size_t)(s + t) * 4);
        sg_box[t] = g4;
        sg_area[t] = (g4[2] - g4[0]) * (g4[3] - g4[1]);
        sg_cls[t] = (int)a.gt_cls[s + t];
    }
    const int pb = blockIdx.x * 256 + wave * 64;          // this wave's first prior
    const int rows = P - pb < 64 ? P - pb : 64;           // (<= 0: a wave past the end; it still reports "nothing" for every box)
    float* const st = stage_all + (size_t)wave * 64 * C;
    if (rows > 0) {
        const int nfl = rows * C;
        const float* src = a.conf + ((size_t)i * P + pb) * C;
        for (int t = lane; t < nfl; t += 64) st[t] = src[t];
    }
    __syncthreads();                                      // boxes staged (and this wave's rows written)
    const int p = pb + lane;
    const bool live = lane < rows;
    f32x4 b = {0.f, 0.f, 0.f, 0.f};
    if (live) b = *reinterpret_cast<const f32x4*>(a.pri_xyxy + (size_t)p * 4);
    const float area_b = (b[2] - b[0]) * (b[3] - b[1]);
    float best = 0.f;
    int idx = s;
    const int slot = blockIdx.x * 4 + wave;
    for (int k = s; k < e; ++k) {
        float v;
        if (k - s < SG) {
            const f32x4 g4 = sg_box[k - s];
            v = iou_xyxy(g4[0], g4[1], g4[2], g4[3], sg_area[k - s], b[0], b[1], b[2], b[3], area_b);
        } else {
            const float ax1 = a.gt[k * 4 + 0], ay1 = a.gt[k * 4 + 1], ax2 = a.gt[k * 4 + 2], ay2 = a.gt[k * 4 + 3];
            v = iou_xyxy(ax1, ay1, ax2, ay2, (ax2 - ax1) * (ay2 - ay1), b[0], b[1], b[2], b[3], area_b);
        }
        if (k == s || better(v, best)) { best = v; idx = k; }
        // this wave's best prior for box k (first index on ties, NaN above numbers: torch.max's rule)
        float wvv = v;
        int wii = live ? p : 0x7fffffff;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(wvv, o, 64);
            const int i2 = __shfl_xor(wii, o, 64);
            if (arg_takes(v2, i2, wvv, wii)) { wvv = v2; wii = i2; }
        }
        if (lane == 0) {
            a.partv[(size_t)k * a.NPW + slot] = wvv;
            a.parti[(size_t)k * a.NPW + slot] = wii;
        }
    }
    if (live) {
        const size_t ip = (size_t)i * P + p;
        // cross entropy against the class the prior has unless a box claims it in L2 (Losses.py:164-167: those few are redone there);
        // torch log_softmax order: (x - max) - log(sum exp(x - max))
        const int c = best < a.thr ? C - 1 : (idx - s < SG ? sg_cls[idx - s] : (int)a.gt_cls[idx]);
        const float* x = st + lane * C;
        float m = x[0];
        for (int q = 1; q < C; ++q) m = fmaxf(m, x[q]);
        float se = 0.f;
        for (int q = 0; q < C; ++q) se += expf(x[q] - m);
        a.cebg[ip] = -((x[c] - m) - logf(se));
        a.bestv[ip] = best;
        a.obj[ip] = idx;
    }
}

// L2: one workgroup of 1024 threads per image
__global__ __launch_bounds__(1024) void loss_image_kernel(const ImageArgs a) {
    constexpr int NW = 16, NT = 1024;
    extern __shared__ __attribute__((aligned(16))) uint32_t vals[];            // [P] negative CE bits of this image
    __shared__ float fred[3][NW];
    __shared__ int ired[NW];
    __shared__ int hist[256];
    __shared__ int s_bcast[4];
    __shared__ int sg_cls[SG], sg_bp[SG];
    __shared__ f32x4 sg_box[SG];
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = a.img_start[i], e = a.img_start[i + 1];
    const int P = a.P, C = a.C;
    const int ns = e - s < SG ? e - s : SG;
    for (int t = tid; t < ns; t += NT) {
        sg_cls[t] = (int)a.gt_cls[s + t];
        sg_box[t] = *reinterpret_cast<const f32x4*>(a.gt + (size_t)(s + t) * 4);
    }
    // ---- best prior of each box: a wave per box over the NPW per-wave results of L1 ----
    for (int k = s + wave; k < e; k += NW) {
        float v = 0.f;
        int ix = 0x7fffffff;
        for (int t = lane; t < a.NPW; t += 64) {
            const float v2 = a.partv[(size_t)k * a.NPW + t];
            const int i2 = a.parti[(size_t)k * a.NPW + t];
            if (arg_takes(v2, i2, v, ix)) { v = v2; ix = i2; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(v, o, 64);
            const int i2 = __shfl_xor(ix, o, 64);
            if (arg_takes(v2, i2, v, ix)) { v = v2; ix = i2; }
        }
        if (lane == 0) {
            if (k - s < SG) sg_bp[k - s] = ix;
            a.best_prior[k] = ix;
        }
    }
    __syncthreads();

    // ---- per prior: forced matches, class, L1 term of the positives.  UP priors per thread in flight: everything a prior may need is
    // requested up front (measured with in-kernel clock stamps: a wave's nine priors in a row, each waiting for its own loads and the positives re-reading their conf row, were 50-70 us per image; this form 12-24) ----
    float my_pos = 0.f, my_ce = 0.f, my_l1 = 0.f;
    const int bg = C - 1;
    constexpr int UP = 3;
    for (int p0 = tid; p0 < P; p0 += UP * NT) {
        float bv[UP], cev[UP];
        int ob[UP];
        f32x4 lc[UP], prr[UP];
#pragma unroll
        for (int u = 0; u < UP; ++u) {
            const int p = p0 + u * NT;
            const size_t ip = (size_t)i * P + (p < P ? p : 0);
            bv[u] = a.bestv[ip];
            ob[u] = a.obj[ip];
            cev[u] = a.cebg[ip];
            lc[u] = *reinterpret_cast<const f32x4*>(a.loc + ip * 4);
            prr[u] = *reinterpret_cast<const f32x4*>(a.pri + (size_t)(p < P ? p : 0) * 4);
        }
#pragma unroll
        for (int u = 0; u < UP; ++u) {
            const int p = p0 + u * NT;
            if (p < P) {
                const size_t ip = (size_t)i * P + p;
                float best = bv[u];
                int idx = ob[u];
                bool forced = false;
                for (int k = 0; k < ns; ++k)                   // Losses.py:164-167: sequential writes, last GT wins
                    if (sg_bp[k] == p) { idx = s + k; best = 1.f; forced = true; }
                for (int k = s + ns; k < e; ++k)
                    if (a.best_prior[k] == p) { idx = k; best = 1.f; forced = true; }
                const int c = best < a.thr ? bg : (idx - s < SG ? sg_cls[idx - s] : (int)a.gt_cls[idx]);
                const bool pos = c != bg;
                a.obj[ip] = idx;
                a.cls[ip] = c;
                float ce = cev[u];
                if (forced) {                                   // (at most one prior per box: its class may have changed, the lane reads its row)
                    const float* x = a.conf + ip * C;
                    float m = x[0];
                    for (int q = 1; q < C; ++q) m = fmaxf(m, x[q]);
                    float se = 0.f;
                    for (int q = 0; q < C; ++q) se += expf(x[q] - m);
                    ce = -((x[c] - m) - logf(se));
                }
                float nv = ce;
                if (pos) {
                    nv = 0.f;
                    my_pos += 1.f;
                    my_ce += ce;
                    float g[4];
                    if (idx - s < SG) {
                        const f32x4 g4 = sg_box[idx - s];
                        const float gb[4] = {g4[0], g4[1], g4[2], g4[3]};
                        encode_gt(gb, 0, prr[u], g);
                    } else {
                        encode_gt(a.gt, idx, prr[u], g);
                    }
                    const f32x4 l = lc[u];
                    my_l1 += (fabsf(l[0] - g[0]) + fabsf(l[1] - g[1])) + (fabsf(l[2] - g[2]) + fabsf(l[3] - g[3]));
                }
                vals[p] = __float_as_uint(nv > 0.f ? nv : 0.f);      // also maps -0.0 and NaN to +0.0
            }
        }
    }
    {
        const float t0 = wave_sum(my_pos), t1 = wave_sum(my_ce), t2 = wave_sum(my_l1);
        if (lane == 0) { fred[0][wave] = t0; fred[1][wave] = t1; fred[2][wave] = t2; }
    }
    __syncthreads();                                       // vals complete, wave sums written
    if (tid == 0) {
        float npos = 0.f, ce = 0.f, l1 = 0.f;
        for (int w = 0; w < NW; ++w) { npos += fred[0][w]; ce += fred[1][w]; l1 += fred[2][w]; }
        a.stats[i * 4 + 0] = npos; a.stats[i * 4 + 1] = ce; a.stats[i * 4 + 2] = l1;
        s_bcast[0] = (int)npos;
    }
    __syncthreads();

    // ---- C: hard negatives ----
    const long kk = (long)a.ratio * s_bcast[0];
    const int k = kk > P ? P : (int)kk;
    if (k <= 0) {
        for (int p = tid; p < P; p += NT) a.sel[(size_t)i * P + p] = 0;
        if (tid == 0) a.stats[i * 4 + 3] = 0.f;
        return;
    }
    uint32_t prefix = 0, mask = 0;
    int remaining = k;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int b = tid; b < 256; b += NT) hist[b] = 0;
        __syncthreads();
        for (int p0 = 0; p0 < P; p0 += NT) {               // uniform trip count: the ballots below need whole waves
            const int p = p0 + tid;
            const uint32_t u = p < P ? vals[p] : 0u;
            bool act = p < P && (u & mask) == prefix;
            const int bin = (int)((u >> shift) & 255u);
            // equal bins of a wave add once (the upper bytes of CE values are nearly all equal: plain atomics would serialise)
            for (int it = 0; it < 4; ++it) {
                const unsigned long long am = __ballot(act);
                if (am == 0ull) break;
                const int leader = __ffsll((long long)am) - 1;
                const int lb = __shfl(bin, leader, 64);
                const bool same = act && bin == lb;
                const unsigned long long sm = __ballot(same);
                if (lane == leader) atomicAdd(&hist[lb], __popcll(sm));
                act = act && !same;
            }
            if (act) atomicAdd(&hist[bin], 1);
        }
        __syncthreads();
        if (wave == 0) {
            // lane l owns bins 255 - 4 l ... 252 - 4 l (descending); walk down from bin 255 until the count reaches `remaining`
            int c4[4], tot = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) { c4[j] = hist[255 - (4 * lane + j)]; tot += c4[j]; }
            int incl = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o, 64);
                if (lane >= o) incl += t;
            }
            int run = incl - tot, d = -1, rem = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int bin = 255 - (4 * lane + j);
                if (d < 0 && bin > 0 && run + c4[j] >= remaining) { d = bin; rem = remaining - run; }
                if (d < 0 && bin > 0) run += c4[j];
            }
            const unsigned long long fm = __ballot(d >= 0);
            if (fm != 0ull) {
                const int first = __ffsll((long long)fm) - 1;
                if (lane == first) { s_bcast[1] = d; s_bcast[2] = rem; }
            } else if (lane == 63) {                       // not reached above bin 0: bin 0 takes what is left (lane 63's run = the count of bins 255 .. 1)
                s_bcast[1] = 0;
                s_bcast[2] = remaining - run;
            }
        }
        __syncthreads();
        prefix |= (uint32_t)s_bcast[1] << shift;
        mask |= 255u << shift;
        remaining = s_bcast[2];
        // (no barrier here: wave 0 writes s_bcast again only behind the next pass's second barrier, which every thread reaches after this read)
    }
    const uint32_t T = prefix;            // k-th largest value; take `remaining` of the elements equal to it
    const int CH = (P + NT - 1) / NT;
    const int p0 = tid * CH, p1 = min(P, p0 + CH);
    float sum_gt = 0.f;
    int n_eq = 0;
    for (int p = p0; p < p1; ++p) {
        const uint32_t u = vals[p];
        if (u > T) sum_gt += __uint_as_float(u);
        n_eq += (u == T);
    }
    int incl = n_eq;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) ired[wave] = incl;
    const float wsum = wave_sum(sum_gt);
    if (lane == 0) fred[0][wave] = wsum;
    __syncthreads();
    int wave_off = 0;
    for (int w = 0; w < wave; ++w) wave_off += ired[w];
    int rank = wave_off + incl - n_eq;     // number of equal elements before this thread's slice
    for (int p = p0; p < p1; ++p) {
        const uint32_t u = vals[p];
        uint8_t sl = 0;
        if (u > T) sl = 1;
        else if (u == T) { sl = rank < remaining ? 1 : 0; ++rank; }
        a.sel[(size_t)i * P + p] = sl;
    }
    if (tid == 0) {
        float tot = 0.f;
        for (int w = 0; w < NW; ++w) tot += fred[0][w];
        a.stats[i * 4 + 3] = tot + (float)remaining * __uint_as_float(T);
    }
}

struct FinalArgs {
    const float* loc; const float* conf; const float* gt; const float* pri;
    const int32_t* obj; const int32_t* cls; const uint8_t* sel; const float* stats;
    float* losses; float* dloc; float* dconf;
    int P, C, bs, norm_mode;
};

__global__ __launch_bounds__(LB) void finalize_kernel(const FinalArgs a) {
    __shared__ float tot[4];
    if (threadIdx.x == 0) {
        float npos = 0.f, ce = 0.f, l1 = 0.f, hn = 0.f;
        for (int i = 0; i < a.bs; ++i) {
            npos += a.stats[i * 4 + 0]; ce += a.stats[i * 4 + 1]; l1 += a.stats[i * 4 + 2]; hn += a.stats[i * 4 + 3];
        }
        tot[0] = npos; tot[1] = ce; tot[2] = l1; tot[3] = hn;
    }
    __syncthreads();
    const float npos = tot[0];
    const float inv = a.norm_mode == 0 ? 1.f / npos : 1.f;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        a.losses[0] = a.norm_mode == 0 ? tot[2] / (npos * 4.f) : tot[2] / 4.f;     // nn.L1Loss mean over n_pos*4
        a.losses[1] = a.norm_mode == 0 ? (tot[3] + tot[1]) / npos : (tot[3] + tot[1]);
        a.losses[2] = npos;
    }
    if (a.dloc == nullptr || a.dconf == nullptr) return;
    extern __shared__ __attribute__((aligned(16))) float fstage[];       // [LB][C]: dconf rows of this block's priors
    const int i = blockIdx.y;
    const int pb = blockIdx.x * LB;
    const int p = pb + threadIdx.x;
    const int rows = a.P - pb < LB ? a.P - pb : LB;
    const size_t ip = (size_t)i * a.P + p;
    if (p < a.P) {
        const int c = a.cls[ip];
        const bool pos = c != a.C - 1;
        const bool on = pos || a.sel[ip] != 0;
        float* dc = fstage + threadIdx.x * a.C;
        if (on) {                                   // (few rows: the conf row is read by its own lane)
            const float* x = a.conf + ip * a.C;
            float m = x[0];
            for (int q = 1; q < a.C; ++q) m = fmaxf(m, x[q]);
            float se = 0.f;
            for (int q = 0; q < a.C; ++q) se += expf(x[q] - m);
            const float r = 1.f / se;
            for (int q = 0; q < a.C; ++q) dc[q] = (expf(x[q] - m) * r - (q == c ? 1.f : 0.f)) * inv;
        } else {
            for (int q = 0; q < a.C; ++q) dc[q] = 0.f;
        }
        f32x4 dl = {0.f, 0.f, 0.f, 0.f};
        if (pos) {
            const f32x4 pr = *reinterpret_cast<const f32x4*>(a.pri + (size_t)p * 4);
            float g[4];
            encode_gt(a.gt, a.obj[ip], pr, g);
            const f32x4 l = *reinterpret_cast<const f32x4*>(a.loc + ip * 4);
            const float sc = inv * 0.25f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float d = l[q] - g[q];
                dl[q] = d > 0.f ? sc : (d < 0.f ? -sc : 0.f);
            }
        }
        *reinterpret_cast<f32x4*>(a.dloc + ip * 4) = dl;
    }
    __syncthreads();
    float* const dst = a.dconf + ((size_t)i * a.P + pb) * a.C;
    const int nfl = rows * a.C;
    for (int t = threadIdx.x; t < nfl; t += LB) dst[t] = fstage[t];
}

int g_loss_form = 1;      // 1 = three launches (L1 wide, L2 per image, L3), 0 = the four-launch form (ssd_tune_set_loss_form)

struct LossWs {
    int32_t* best_prior; float* neg; float* part; uint8_t* sel; float* stats; float* partv; int32_t* parti; float* bestv; size_t bytes;
};
LossWs carve(void* ws, int bs, int P, int n_gt) {
    const int NB = ssd_cdiv(P, LB);
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    char* b = reinterpret_cast<char*>(ws);
    size_t o = 0;
    LossWs w;
    w.best_prior = reinterpret_cast<int32_t*>(b + o); o += up((size_t)n_gt * 4);
    w.neg = reinterpret_cast<float*>(b + o); o += up((size_t)bs * P * 4);
    w.part = reinterpret_cast<float*>(b + o); o += up((size_t)bs * NB * 3 * 4);
    w.sel = reinterpret_cast<uint8_t*>(b + o); o += up((size_t)bs * P);
    w.stats = reinterpret_cast<float*>(b + o); o += up((size_t)bs * 4 * 4);
    w.partv = reinterpret_cast<float*>(b + o); o += up((size_t)n_gt * NB * 4 * 4);
    w.parti = reinterpret_cast<int32_t*>(b + o); o += up((size_t)n_gt * NB * 4 * 4);
    w.bestv = reinterpret_cast<float*>(b + o); o += up((size_t)bs * P * 4);
    w.bytes = o;
    return w;
}

}  // namespace

// Cross-check aid: 1 = the three-launch form (default), 0 = the four-launch form of rounds 1-3.
extern "C" int ssd_tune_set_loss_form(int three_launch) {
    g_loss_form = three_launch ? 1 : 0;
    return SSD_OK;
}

extern "C" size_t ssd_multibox_loss_workspace(int bs, int P, int n_gt) {
    if (bs <= 0 || P <= 0 || n_gt <= 0) return 0;
    return carve(nullptr, bs, P, n_gt).bytes;
}

extern "C" int ssd_multibox_loss(const float* loc, const float* conf, const float* gt_boxes, const float* gt_classes,
                                 const int32_t* img_start, int bs, int n_gt, const float* priors_cxcywh,
                                 const float* priors_xyxy, int P, int n_classes, float iou_threshold, int neg_pos_ratio,
                                 int norm_mode, float* losses, int32_t* obj, int32_t* cls, float* dloc, float* dconf,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    if (!loc || !conf || !gt_boxes || !gt_classes || !img_start || !priors_cxcywh || !priors_xyxy || !losses || !obj ||
        !cls || !workspace)
        return SSD_ERR_NULL;
    if ((dloc == nullptr) != (dconf == nullptr)) return SSD_ERR_NULL;
    if (bs <= 0 || n_gt < bs || P <= 0 || P > 36000 || n_classes < 2 || n_classes > 64 || neg_pos_ratio < 0 ||
        (norm_mode != 0 && norm_mode != 1))
        return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(loc) || !ssd_aligned16(priors_cxcywh) || !ssd_aligned16(priors_xyxy) || (dloc && !ssd_aligned16(dloc)) ||
        !ssd_aligned16(workspace))
        return SSD_ERR_ALIGN;
    if (workspace_bytes < ssd_multibox_loss_workspace(bs, P, n_gt)) return SSD_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const LossWs w = carve(workspace, bs, P, n_gt);
    const int NB = ssd_cdiv(P, LB);
    if (g_loss_form != 0) {
        // L1 wide (a wave per 64 priors), L2 one workgroup per image
        const size_t l1_lds = (size_t)4 * 64 * n_classes * 4, l2_lds = (size_t)P * 4;
        static std::atomic<unsigned long long> raised_l12{0};     // one bit per device (common.h)
        int dev;
        if ((l1_lds > 48 * 1024 || l2_lds > 48 * 1024) && ssd_attr_needed(raised_l12, dev)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(loss_prior_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) !=
                    hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(loss_image_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) !=
                    hipSuccess)
                return SSD_ERR_LAUNCH;
            ssd_attr_done(raised_l12, dev);
        }
        ImageArgs ia{loc, conf, gt_boxes, gt_classes, img_start, priors_cxcywh, priors_xyxy, w.partv, w.parti, w.bestv, w.neg,
                     w.best_prior, obj, cls, w.sel, w.stats, P, n_classes, neg_pos_ratio, NB * 4, iou_threshold};
        hipLaunchKernelGGL(loss_prior_kernel, dim3(NB, bs), dim3(256), l1_lds, st, ia);
        SSD_CHECK_LAUNCH();
        hipLaunchKernelGGL(loss_image_kernel, dim3(bs), dim3(1024), l2_lds, st, ia);
        SSD_CHECK_LAUNCH();
    } else {
        hipLaunchKernelGGL(best_prior_per_gt_kernel, dim3(n_gt), dim3(256), 0, st, gt_boxes, priors_xyxy, P, w.best_prior);
        SSD_CHECK_LAUNCH();
        MatchArgs ma{loc, conf, gt_boxes, gt_classes, img_start, w.best_prior, priors_cxcywh, priors_xyxy,
                     obj, cls, w.neg, w.part, P, n_classes, NB, iou_threshold};
        hipLaunchKernelGGL(match_ce_kernel, dim3(NB, bs), dim3(LB), 0, st, ma);
        SSD_CHECK_LAUNCH();
        const size_t hn_lds = (size_t)P * sizeof(uint32_t);
        if (hn_lds > 48 * 1024) {
            static std::atomic<unsigned long long> raised{0};     // one bit per device (common.h)
            int dev;
            if (ssd_attr_needed(raised, dev)) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(hard_negative_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        150 * 1024) != hipSuccess)
                    return SSD_ERR_LAUNCH;
                ssd_attr_done(raised, dev);
            }
        }
        hipLaunchKernelGGL(hard_negative_kernel, dim3(bs), dim3(HB), hn_lds, st, w.neg, w.part, w.sel, w.stats, P, NB, neg_pos_ratio);
        SSD_CHECK_LAUNCH();
    }
    FinalArgs fa{loc, conf, gt_boxes, priors_cxcywh, obj, cls, w.sel, w.stats, losses, dloc, dconf, P, n_classes, bs, norm_mode};
    const bool grads = dloc != nullptr;
    const size_t fin_lds = grads ? (size_t)LB * n_classes * 4 : 0;
    if (fin_lds > 48 * 1024) {
        static std::atomic<unsigned long long> raised_fin{0};
        int dev;
        if (ssd_attr_needed(raised_fin, dev)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(finalize_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) !=
                hipSuccess)
                return SSD_ERR_LAUNCH;
            ssd_attr_done(raised_fin, dev);
        }
    }
    hipLaunchKernelGGL(finalize_kernel, grads ? dim3(NB, bs) : dim3(1, 1), dim3(LB), fin_lds, st, fa);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}
