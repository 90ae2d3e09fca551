// Per-class 11-point average precision over a set of images (Util.py:783-885 get_map), on the device.
//
//   M1 init      availability byte per ground-truth box, counters
//   M2 match     one wave per (image, class): that image's detections of the class in descending (score, lower
//                flat index first) order -- selection by repeated wave-max of a 64-bit key --; IoU against the
//                image's ground truth of the class on the lanes, wave arg-max with first index on ties
//                (Util.py:852-854); true positive iff IoU > 0.5 and the box is unclaimed (:855-859).
//                Matching never crosses images or classes, so the pairs are independent.
//   M3 count     detections / ground truth per class (integer atomics: order-independent)
//   M4 bucket    block per class: stable compaction of the class's detections, their sort keys
//   M5 rank      rank of every detection inside its class = number of larger keys (keys are unique): the global
//                per-class sort (Util.py:829-831) is a scatter
//   M6 ap        block per class: inclusive scan of the sorted TP flags; precision = cumTP / position and
//                recall = float64(float32(1 / n_gt)) * cumTP in double (the reference's `numpy / long tensor` goes
//                through Tensor.__rtruediv__ = reciprocal() * other with a float32 reciprocal); max precision at
//                recall >= each level (0 where none) -> table[class][level]; the mean over levels is host side.
// IoU is the same contraction-free f32 sequence as the matcher's and the NMS's (Util.py:252-301).
#include "common.h"
#pragma clang fp contract(off)

namespace {

constexpr int MAX_LEVELS = 16;

__device__ __forceinline__ float iou_boxes(const f32x4 a, const f32x4 b) {
    const float lx = fmaxf(a[0], b[0]), ly = fmaxf(a[1], b[1]);
    const float hx = fminf(a[2], b[2]), hy = fminf(a[3], b[3]);
    const float dx = fmaxf(hx - lx, 0.f), dy = fmaxf(hy - ly, 0.f);
    const float inter = dx * dy;
    const float a1 = (a[2] - a[0]) * (a[3] - a[1]);
    const float a2 = (b[2] - b[0]) * (b[3] - b[1]);
    return inter / ((a1 + a2) - inter);
}

// monotone map float -> uint32 (larger float = larger integer)
__device__ __forceinline__ uint32_t ordered_bits(float v) {
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ uint64_t det_key(float score, int flat_index) {
    return ((uint64_t)ordered_bits(score) << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)flat_index);
}
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)v, o, 64), hi = __shfl_xor((uint32_t)(v >> 32), o, 64);
        const uint64_t w = ((uint64_t)hi << 32) | lo;
        v = w > v ? w : v;
    }
    return v;
}

__global__ void map_init_kernel(uint8_t* avail, int G, int32_t* counts, int n_counts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < G) avail[i] = 1;
    if (i < n_counts) counts[i] = 0;
}

__global__ __launch_bounds__(256) void map_match_kernel(const float* __restrict__ det_boxes, const int32_t* __restrict__ det_classes,
                                                        const float* __restrict__ det_scores, const int32_t* __restrict__ det_start,
                                                        const float* __restrict__ gt_boxes, const int32_t* __restrict__ gt_classes,
                                                        const int32_t* __restrict__ gt_start, int B, int n_classes,
                                                        uint8_t* __restrict__ avail, uint8_t* __restrict__ tp) {
    const int lane = threadIdx.x & 63;
    const long pair = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= (long)B * n_classes) return;                 // whole wave leaves together
    const int b = (int)(pair / n_classes), c = (int)(pair % n_classes);
    const int ds = det_start[b], de = det_start[b + 1], gs = gt_start[b], ge = gt_start[b + 1];
    uint64_t prev = ~0ull;
    for (;;) {
        uint64_t best = 0;
        for (int i = ds + lane; i < de; i += 64) {
            if (det_classes[i] == c) {
                const uint64_t k = det_key(det_scores[i], i);
                if (k < prev && k > best) best = k;
            }
        }
        best = wave_max_u64(best);
        if (best == 0) break;                                  // uniform: no detection of this class left
        prev = best;
        const int d = (int)(0xFFFFFFFFu - (uint32_t)best);
        const f32x4 box = *reinterpret_cast<const f32x4*>(det_boxes + (size_t)d * 4);
        float v_best = -1.f;
        int g_best = 0x7FFFFFFF;
        bool nan = false;
        for (int g = gs + lane; g < ge; g += 64) {
            if (gt_classes[g] == c) {
                const float v = iou_boxes(box, *reinterpret_cast<const f32x4*>(gt_boxes + (size_t)g * 4));
                nan |= (v != v);
                if (v > v_best) { v_best = v; g_best = g; }     // ascending g per lane: strict > keeps the first
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(v_best, o, 64);
            const int og = __shfl_xor(g_best, o, 64);
            if (ov > v_best || (ov == v_best && og < g_best)) { v_best = ov; g_best = og; }
        }
        const bool any_nan = __ballot(nan) != 0ull;           // torch.max propagates NaN; NaN > 0.5 is false
        if (lane == 0) {
            uint8_t hit = 0;
            if (!any_nan && g_best != 0x7FFFFFFF && v_best > 0.5f && avail[g_best]) {
                hit = 1;
                avail[g_best] = 0;
            }
            tp[d] = hit;
        }
    }
}

__global__ void map_count_kernel(const int32_t* __restrict__ det_classes, int D, const int32_t* __restrict__ gt_classes, int G,
                                 int n_classes, int32_t* __restrict__ counts) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < D + G; i += gridDim.x * blockDim.x) {
        const int c = i < D ? det_classes[i] : gt_classes[i - D];
        if (c >= 0 && c < n_classes) atomicAdd(&counts[(i < D ? 0 : n_classes) + c], 1);
    }
}

__global__ __launch_bounds__(256) void map_bucket_kernel(const int32_t* __restrict__ det_classes, const float* __restrict__ det_scores,
                                                         int D, const int32_t* __restrict__ counts, int32_t* __restrict__ list,
                                                         uint64_t* __restrict__ keys) {
    __shared__ int wave_cnt[4];
    const int c = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int off = 0;
    for (int k = 0; k < c; ++k) off += counts[k];
    int base = 0;
    for (int i0 = 0; i0 < D; i0 += 256) {
        const int i = i0 + threadIdx.x;
        const bool mine = i < D && det_classes[i] == c;
        const uint64_t bal = __ballot(mine);
        if (lane == 0) wave_cnt[wv] = __popcll(bal);
        __syncthreads();
        int before = 0;
        for (int k = 0; k < wv; ++k) before += wave_cnt[k];
        const int total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        if (mine) {
            const int pos = off + base + before + __popcll(bal & ((1ull << lane) - 1ull));
            list[pos] = i;
            keys[pos] = det_key(det_scores[i], i);
        }
        base += total;
        __syncthreads();
    }
}

__global__ void map_rank_kernel(const int32_t* __restrict__ det_classes, const int32_t* __restrict__ counts, int n_classes,
                                const int32_t* __restrict__ list, const uint64_t* __restrict__ keys, const uint8_t* __restrict__ tp,
                                uint8_t* __restrict__ sorted_tp) {
    int total = 0;
    for (int k = 0; k < n_classes; ++k) total += counts[k];
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < total; p += gridDim.x * blockDim.x) {
        const int i = list[p];
        const int c = det_classes[i];
        int off = 0;
        for (int k = 0; k < c; ++k) off += counts[k];
        const int n = counts[c];
        const uint64_t key = keys[p];
        int rank = 0;
        for (int q = off; q < off + n; ++q) rank += keys[q] > key ? 1 : 0;
        sorted_tp[off + rank] = tp[i];
    }
}

struct LevelArgs {
    double level[MAX_LEVELS];
};

__global__ __launch_bounds__(256) void map_ap_kernel(const int32_t* __restrict__ counts, int n_classes, const uint8_t* __restrict__ sorted_tp,
                                                     const LevelArgs lv, int n_levels, double* __restrict__ table) {
    __shared__ int wave_cnt[4];
    __shared__ double red[4][MAX_LEVELS];
    const int c = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int off = 0;
    for (int k = 0; k < c; ++k) off += counts[k];
    const int n = counts[c], n_gt = counts[n_classes + c];
    const double rinv = (double)(1.0f / (float)n_gt);          // reciprocal() of a long tensor is float32; n_gt = 0 -> inf
    double best[MAX_LEVELS];
#pragma unroll
    for (int t = 0; t < MAX_LEVELS; ++t) best[t] = -1.0;       // -1 = no position reached the level yet
    int run_tp = 0;
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + threadIdx.x;
        const bool is_tp = i < n && sorted_tp[off + i] != 0;
        const uint64_t bal = __ballot(is_tp);
        if (lane == 0) wave_cnt[wv] = __popcll(bal);
        __syncthreads();
        int before = 0;
        for (int k = 0; k < wv; ++k) before += wave_cnt[k];
        const int total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        if (i < n) {
            const int cum_tp = run_tp + before + __popcll(bal & ((lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull)));
            const double prec = (double)cum_tp / (double)(i + 1);      // cumTP + cumFP == position, exactly
            const double rec = rinv * (double)cum_tp;                  // inf * 0 = NaN: compares false
#pragma unroll
            for (int t = 0; t < MAX_LEVELS; ++t)
                if (t < n_levels && rec >= lv.level[t] && prec > best[t]) best[t] = prec;
        }
        run_tp += total;
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < MAX_LEVELS; ++t) {
        double v = best[t];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double w = __shfl_xor(v, o, 64);
            v = w > v ? w : v;
        }
        if (lane == 0) red[wv][t] = v;
    }
    __syncthreads();
    if (threadIdx.x < n_levels) {
        double v = red[0][threadIdx.x];
        for (int k = 1; k < 4; ++k) v = red[k][threadIdx.x] > v ? red[k][threadIdx.x] : v;
        table[(size_t)c * n_levels + threadIdx.x] = v < 0.0 ? 0.0 : v;
    }
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct MapWs {
    uint8_t* avail;
    int32_t* list;
    uint64_t* keys;
    uint8_t* sorted_tp;
    size_t bytes;
};
MapWs carve(void* base, int D, int G) {
    MapWs w;
    size_t o = 0;
    char* b = static_cast<char*>(base);
    w.keys = reinterpret_cast<uint64_t*>(b + o); o += align256((size_t)(D > 0 ? D : 1) * 8);
    w.list = reinterpret_cast<int32_t*>(b + o); o += align256((size_t)(D > 0 ? D : 1) * 4);
    w.avail = reinterpret_cast<uint8_t*>(b + o); o += align256((size_t)(G > 0 ? G : 1));
    w.sorted_tp = reinterpret_cast<uint8_t*>(b + o); o += align256((size_t)(D > 0 ? D : 1));
    w.bytes = o;
    return w;
}

}  // namespace

extern "C" size_t ssd_map_eval_workspace(int D, int G) {
    if (D < 0 || G < 0) return 0;
    return carve(nullptr, D, G).bytes;
}

extern "C" int ssd_map_eval(const float* det_boxes, const int32_t* det_classes, const float* det_scores, const int32_t* det_start,
                            int D, const float* gt_boxes, const int32_t* gt_classes, const int32_t* gt_start, int G, int B,
                            int n_classes, const double* recall_levels_host, int n_levels, uint8_t* tp, double* table,
                            int32_t* counts, void* workspace, size_t workspace_bytes, void* stream) {
    if (!det_start || !gt_start || !recall_levels_host || !table || !counts) return SSD_ERR_NULL;
    if ((D > 0 && (!det_boxes || !det_classes || !det_scores || !tp)) || (G > 0 && (!gt_boxes || !gt_classes))) return SSD_ERR_NULL;
    if (D < 0 || G < 0 || B <= 0 || n_classes <= 0 || n_classes > 256 || n_levels <= 0 || n_levels > MAX_LEVELS) return SSD_ERR_BAD_SHAPE;
    if ((long)B * n_classes >= (1L << 31)) return SSD_ERR_BAD_SHAPE;
    if (!workspace || workspace_bytes < ssd_map_eval_workspace(D, G)) return SSD_ERR_WORKSPACE;
    if (!ssd_aligned16(workspace) || (D > 0 && !ssd_aligned16(det_boxes)) || (G > 0 && !ssd_aligned16(gt_boxes))) return SSD_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    const MapWs w = carve(workspace, D, G);
    LevelArgs lv{};
    for (int t = 0; t < n_levels; ++t) lv.level[t] = recall_levels_host[t];
    const int n_init = (G > 2 * n_classes ? G : 2 * n_classes);
    hipLaunchKernelGGL(map_init_kernel, dim3(ssd_cdiv(n_init, 256)), dim3(256), 0, st, w.avail, G, counts, 2 * n_classes);
    SSD_CHECK_LAUNCH();
    if (D > 0) {
        hipLaunchKernelGGL(map_match_kernel, dim3((unsigned)(((long)B * n_classes + 3) / 4)), dim3(256), 0, st, det_boxes, det_classes,
                           det_scores, det_start, gt_boxes, gt_classes, gt_start, B, n_classes, w.avail, tp);
        SSD_CHECK_LAUNCH();
    }
    if (D + G > 0) {
        const int blocks = ssd_cdiv(D + G, 256) > 1024 ? 1024 : ssd_cdiv(D + G, 256);
        hipLaunchKernelGGL(map_count_kernel, dim3(blocks), dim3(256), 0, st, det_classes, D, gt_classes, G, n_classes, counts);
        SSD_CHECK_LAUNCH();
    }
    if (D > 0) {
        hipLaunchKernelGGL(map_bucket_kernel, dim3(n_classes), dim3(256), 0, st, det_classes, det_scores, D, counts, w.list, w.keys);
        SSD_CHECK_LAUNCH();
        const int blocks = ssd_cdiv(D, 256) > 2048 ? 2048 : ssd_cdiv(D, 256);
        hipLaunchKernelGGL(map_rank_kernel, dim3(blocks), dim3(256), 0, st, det_classes, counts, n_classes, w.list, w.keys, tp, w.sorted_tp);
        SSD_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(map_ap_kernel, dim3(n_classes), dim3(256), 0, st, counts, n_classes, w.sorted_tp, lv, n_levels, table);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}
