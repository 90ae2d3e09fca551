// Decode + softmax + per-class NMS + cross-class top-k for one image
// (Losses.py:11-98 inference; Util.py:86-96 gcxgcy_to_cxcy, xywh_to_xyxy; Util.py:252-301 IoU).
//
//   D1 decode_compact  thread per prior: box (cxcywh -> xyxy), class probabilities (the 21 scores of 256 priors staged through LDS:
//                      coalesced reads), and straight away the candidates prob >= min_score of every class as 64-bit keys
//                      (prob bits << 32 | ~prior index): descending key order = descending prob, lower prior index first on ties
//                      (the CPU sort order).  Slots come from one atomic per (wave, class) -- ballot + prefix --
//                      so the order inside keys[] varies from run to run; the ranks D3 computes from the unique keys do not.
//   D3 rank_scatter    rank of every candidate = number of larger keys (keys are unique), so
//                      the sort is a scatter; all classes and candidates in parallel
//   D4 nms             block per class, greedy in sorted order, 64 rows at a time: suppression words by
//                      wave ballot, in-chunk resolve by v_readlane, removed bitset in LDS
//   D5 topk_emit       class-major offsets and compact list of the kept boxes (first phase of the same block), then:
//                      if more than top_k survive: radix select of the top_k-th probability, ties in
//                      class-major order, rank of the selected entries in LDS; scale boxes by (w,h,w,h)
// IoU uses the same contraction-free f32 sequence as the matcher.
#include "common.h"
#pragma clang fp contract(off)

namespace {

constexpr int NB_T = 1024;

__device__ __forceinline__ float iou_boxes(const f32x4 a, const f32x4 b) {
    const float lx = fmaxf(a[0], b[0]), ly = fmaxf(a[1], b[1]);
    const float hx = fminf(a[2], b[2]), hy = fminf(a[3], b[3]);
    const float dx = fmaxf(hx - lx, 0.f), dy = fmaxf(hy - ly, 0.f);
    const float inter = dx * dy;
    const float a1 = (a[2] - a[0]) * (a[3] - a[1]);
    const float a2 = (b[2] - b[0]) * (b[3] - b[1]);
    return inter / ((a1 + a2) - inter);
}

struct NmsWs {
    float* boxes;        // [P][4] xyxy
    float* probs_t;      // [C-1][P]
    uint64_t* keys;      // [C-1][P] unsorted candidate keys
    int32_t* cand_cnt;   // [C-1]
    float* s_boxes;      // [C-1][P][4] sorted
    float* s_prob;       // [C-1][P]
    int32_t* s_idx;      // [C-1][P]
    int32_t* kept_pos;   // [C-1][P] sorted positions of kept boxes, in order
    uint32_t* kept_prob; // [C-1][P] probability bits of the kept boxes, in the same order
    int32_t* kept_cnt;   // [C-1]
    int32_t* offsets;    // [C] class-major exclusive offsets, [C-1] = total
    uint32_t* k_prob;    // [(C-1)*P] prob bits of the kept boxes, class-major
    int32_t* k_src;      // [(C-1)*P] index into the sorted arrays (c*P + pos)
    size_t bytes;
};

NmsWs carve(void* ws, int P, int C, int B) {
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    char* b = reinterpret_cast<char*>(ws);
    size_t o = 0;
    const size_t K = (size_t)(C - 1) * P * B;
    NmsWs w;
    w.boxes = reinterpret_cast<float*>(b + o); o += up((size_t)P * 16 * B);
    w.probs_t = reinterpret_cast<float*>(b + o); o += up(K * 4);
    w.keys = reinterpret_cast<uint64_t*>(b + o); o += up(K * 8);
    w.cand_cnt = reinterpret_cast<int32_t*>(b + o); o += up((size_t)C * 4 * B);
    w.s_boxes = reinterpret_cast<float*>(b + o); o += up(K * 16);
    w.s_prob = reinterpret_cast<float*>(b + o); o += up(K * 4);
    w.s_idx = reinterpret_cast<int32_t*>(b + o); o += up(K * 4);
    w.kept_pos = reinterpret_cast<int32_t*>(b + o); o += up(K * 4);
    w.kept_prob = reinterpret_cast<uint32_t*>(b + o); o += up(K * 4);
    w.kept_cnt = reinterpret_cast<int32_t*>(b + o); o += up((size_t)C * 4 * B);
    w.offsets = reinterpret_cast<int32_t*>(b + o); o += up((size_t)(C + 1) * 4 * B);
    w.k_prob = reinterpret_cast<uint32_t*>(b + o); o += up(K * 4);
    w.k_src = reinterpret_cast<int32_t*>(b + o); o += up(K * 4);
    w.bytes = o;
    return w;
}

// grid = (ceil(P/256), 1, B).  cand_cnt[(C-1)+1 per image] must be zero on entry (hipMemsetAsync in front of the launch).
__global__ __launch_bounds__(256) void decode_compact_kernel(const float* __restrict__ l_, const float* __restrict__ c_, const float* __restrict__ pri,
                                                             int P, int C, float min_score, float* __restrict__ boxes, uint64_t* __restrict__ keys,
                                                             int32_t* __restrict__ cand_cnt) {
    extern __shared__ float sc[];                                   // [256][C] class scores of this block's priors
    const int p0 = blockIdx.x * 256, tid = threadIdx.x, lane = tid & 63;
    const int p = p0 + tid;
    const size_t img = blockIdx.z;
    const int C1 = C - 1;
    l_ += img * P * 4; c_ += img * P * C; boxes += img * P * 4; keys += img * (size_t)C1 * P; cand_cnt += img * (C1 + 1);
    const int rows = min(256, P - p0);
    for (int e = tid; e < rows * C; e += 256) sc[e] = c_[(size_t)p0 * C + e];      // consecutive threads, consecutive floats
    __syncthreads();
    const bool live = p < P;
    float m = 0.f, se = 1.f;
    if (live) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(l_ + (size_t)p * 4);
        const f32x4 pr = *reinterpret_cast<const f32x4*>(pri + (size_t)p * 4);
        // Util.py:89-91: g_c * p_wh / 10 + p_c ; exp(g_wh / 5) * p_wh
        const float cx = g[0] * pr[2] / 10.f + pr[0], cy = g[1] * pr[3] / 10.f + pr[1];
        const float w = expf(g[2] / 5.f) * pr[2], h = expf(g[3] / 5.f) * pr[3];
        // Util.py:93-96: c - wh/2, c + wh/2
        f32x4 b;
        b[0] = cx - w / 2.f; b[1] = cy - h / 2.f; b[2] = cx + w / 2.f; b[3] = cy + h / 2.f;
        *reinterpret_cast<f32x4*>(boxes + (size_t)p * 4) = b;
        const float* x = sc + tid * C;                              // row stride C = 21 floats: odd, conflict-free
        float* xw = sc + tid * C;
        m = x[0];
        for (int q = 1; q < C; ++q) m = fmaxf(m, x[q]);
        se = 0.f;
        for (int q = 0; q < C; ++q) {                               // exp(x - max) is kept in place of x: the second pass divides, it does
            const float e = expf(x[q] - m);                         // not exponentiate again (same value, 21 expf per prior instead of 41)
            xw[q] = e;
            se += e;
        }
    }
    // Candidates, 32 classes at a time: every lane collects its hit bits (the probability replaces exp in LDS), lane k then owns class
    // q0 + k -- counts the wave's hits of that class and takes the wave's slots with ONE atomic -- so a wave waits for one atomic round
    // trip per 32 classes instead of one per class (20 dependent returns were most of this kernel's time).
    for (int q0 = 0; q0 < C1; q0 += 32) {
        const int nq = min(32, C1 - q0);
        unsigned hits = 0;
        for (int k = 0; k < nq; ++k) {
            const float v = live ? sc[tid * C + q0 + k] / se : 0.f;
            if (live) sc[tid * C + q0 + k] = v;
            hits |= (live && v >= min_score ? 1u : 0u) << k;        // Losses.py:32 (NaN fails, as in torch)
        }
        int cnt = 0;
        for (int k = 0; k < nq; ++k) {                              // uniform trip count: the ballots need every lane
            const int pc = __popcll(__ballot((hits >> k) & 1u));
            if (lane == k) cnt = pc;
        }
        int base = 0;
        if (cnt > 0) base = atomicAdd(&cand_cnt[q0 + lane], cnt);
        for (int k = 0; k < nq; ++k) {
            const uint64_t mask = __ballot((hits >> k) & 1u);
            if (mask == 0) continue;                                // wave-uniform
            const int b0 = __shfl(base, k, 64);
            if ((hits >> k) & 1u) {
                const int pos = b0 + __popcll(mask & ((1ull << lane) - 1ull));
                const float v = sc[tid * C + q0 + k];
                keys[(size_t)(q0 + k) * P + pos] = ((uint64_t)__float_as_uint(v) << 32) | (uint64_t)(0xffffffffu - (uint32_t)p);
            }
        }
    }
}

__global__ __launch_bounds__(256) void rank_scatter_kernel(const uint64_t* __restrict__ keys, const int32_t* __restrict__ cand_cnt,
                                                           const float* __restrict__ boxes, int P, float* __restrict__ s_boxes,
                                                           float* __restrict__ s_prob, int32_t* __restrict__ s_idx) {
    __shared__ uint64_t tile[256];
    const int c = blockIdx.y;
    const size_t img = blockIdx.z, C1 = gridDim.y;
    keys += img * C1 * P; cand_cnt += img * (C1 + 1); boxes += img * P * 4;
    s_boxes += img * C1 * P * 4; s_prob += img * C1 * P; s_idx += img * C1 * P;
    const int n = cand_cnt[c];
    if (blockIdx.x * 256 >= n) return;                          // uniform per block
    const int i = blockIdx.x * 256 + threadIdx.x;
    const uint64_t* kc = keys + (size_t)c * P;
    const uint64_t mine = i < n ? kc[i] : 0;
    int rank = 0;
    for (int j0 = 0; j0 < n; j0 += 256) {
        const int j = j0 + threadIdx.x;
        tile[threadIdx.x] = j < n ? kc[j] : 0;
        __syncthreads();
        const int lim = min(256, n - j0);
        for (int t = 0; t < lim; ++t) rank += tile[t] > mine;
        __syncthreads();
    }
    if (i < n) {
        const int p = (int)(0xffffffffu - (uint32_t)(mine & 0xffffffffu));
        const size_t o = (size_t)c * P + rank;
        *reinterpret_cast<f32x4*>(s_boxes + o * 4) = *reinterpret_cast<const f32x4*>(boxes + (size_t)p * 4);
        s_prob[o] = __uint_as_float((uint32_t)(mine >> 32));
        s_idx[o] = p;
    }
}

// Greedy NMS of one class (block per class, 16 waves), boxes already sorted by descending probability.
// Rows are taken CR (64 or 32) at a time:
//   A  suppression words by wave ballot: word (r,w) = lanes j = 64w..64w+63 with IoU(box[base+r], box[j]) >= thr
//      and j > base+r (rows already removed are skipped); column boxes are loaded once per (chunk, word);
//   B  wave 0 resolves the CR rows against each other serially from the diagonal words held one per lane
//      (v_readlane), giving the chunk's keep mask;
//   C  the kept rows' words are OR-ed into the removed bitset of all later columns.
// All state (removed bitset, CR x n/64 words) lives in LDS; no n x n matrix ever reaches memory.
//
// The test IoU >= thr is the reference's float sequence inter / ((a1 + a2) - inter) >= thr (Util.py:252-301, Losses.py:51) bit for
// bit, but the IEEE division (~12 dependent instructions) is only executed where it can matter: with t = thr * union and
// d = inter - t, |d| > 1e-5 t decides the comparison whatever the two roundings (each <= 6e-8 relative) do, so a wave divides only
// when one of its lanes lies within 1e-5 of the threshold (or the union is not a normal positive number).  Four rows are tested per
// pass -- four independent dependency chains per lane -- and the areas are computed once per box, not once per pair.
__device__ __forceinline__ float hw_max(float x, float y) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ float hw_min(float x, float y) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
struct IouTest { float inter, uni, d, t; };
__device__ __forceinline__ IouTest iou_test(const f32x4 a, float area_a, const f32x4 b, float area_b, float thr) {
    IouTest r;
    // v_max / v_min as they are: fmaxf / fminf would first canonicalise each of the eight inputs (signalling-NaN semantics), eight more
    // instructions per test.  A NaN coordinate makes its box's area NaN, so the test ends in the exact path and is false either way.
    const float lx = hw_max(a[0], b[0]), ly = hw_max(a[1], b[1]);
    const float hx = hw_min(a[2], b[2]), hy = hw_min(a[3], b[3]);
    const float dx = hw_max(hx - lx, 0.f), dy = hw_max(hy - ly, 0.f);
    r.inter = dx * dy;
    r.uni = (area_a + area_b) - r.inter;
    r.t = thr * r.uni;
    r.d = r.inter - r.t;           // decided where |d| > 1e-5 t and t > 1e-20: false for NaN / infinity anywhere, for empty boxes and for thr <= 0
    return r;
}

template <int NT>
__global__ __launch_bounds__(NT) void nms_kernel(const float* __restrict__ s_boxes, const float* __restrict__ s_prob,
                                                   const int32_t* __restrict__ cand_cnt, int P, float thr, int chunk_words,
                                                   int32_t* __restrict__ kept_pos, uint32_t* __restrict__ kept_prob, int32_t* __restrict__ kept_cnt) {
    extern __shared__ __attribute__((aligned(16))) uint64_t sm64[];
    __shared__ int wave_tot[NT / 64];
    __shared__ int running;
    __shared__ uint64_t s_keep;
    __shared__ __attribute__((aligned(16))) float rowbox[64 * 8];            // box + area per row of the chunk
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // provably uniform: the item / row bookkeeping below stays on the scalar unit
    const size_t img = blockIdx.z, C1 = gridDim.x;
    s_boxes += img * C1 * P * 4; s_prob += img * C1 * P; cand_cnt += img * (C1 + 1);
    kept_pos += img * C1 * P; kept_prob += img * C1 * P; kept_cnt += img * (C1 + 1);
    const int n = cand_cnt[c];
    const int nw = (n + 63) >> 6, nwcap = (P + 63) >> 6;
    uint64_t* removed = sm64;                 // [nwcap]
    uint64_t* chunk = sm64 + nwcap;           // [CR][nw]: 64 rows at a time where the class's own word count lets them fit, else 32
    const int CR = 64 * nw <= chunk_words ? 64 : 32;
    const float* bx = s_boxes + (size_t)c * P * 4;
    for (int w = tid; w < nw; w += NT) removed[w] = 0;
    __syncthreads();
    for (int base = 0; base < n; base += CR) {
        const int R = base >> 6, bit0 = base & 63;
        const int span = nw - R;
        // ---- A ------------------------------------------------------------------------------------------
        // the chunk's row boxes (+ areas) go to LDS once; a wave item = (column word w, group of CR/4 rows): every lane
        // loads its column box once and tests it against the rows (LDS broadcast), four rows per pass, one ballot per row
        const uint64_t rem_start = removed[R];
        if (tid < CR) {
            const int i = base + tid;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (i < n) v = *reinterpret_cast<const f32x4*>(bx + (size_t)i * 4);
            *reinterpret_cast<f32x4*>(rowbox + tid * 8) = v;
            rowbox[tid * 8 + 4] = (v[2] - v[0]) * (v[3] - v[1]);
        }
        __syncthreads();
        const uint64_t dead = (rem_start >> bit0) | (n - base >= CR ? 0ull : (~0ull << (n - base)));       // bit r: row base + r needs no words
        const int RG = CR >> 2;
#ifndef NMS_PROBE_NO_A                         // timing probes (tools/nms_probe.sh): a phase compiled out, results meaningless
        for (int item = wave; item < 4 * span; item += NT / 64) {
            const int w = R + (item >> 2), r0 = (item & 3) * RG;
            const int j = (w << 6) + lane;
            f32x4 b = {0.f, 0.f, 0.f, 0.f};
            if (j < n) b = *reinterpret_cast<const f32x4*>(bx + (size_t)j * 4);
            const float area_b = (b[2] - b[0]) * (b[3] - b[1]);
            uint64_t col_ok = n - (w << 6) >= 64 ? ~0ull : ~(~0ull << (n - (w << 6)));                         // lanes with j < n
            const bool diag = w == R;                                                                          // (uniform) the word that holds the rows themselves
            // the rows of this item that still need their words, four at a time whatever their positions (scalar bit scan): a row
            // removed before this chunk costs nothing, and the four tests of a pass are independent instruction chains
            uint64_t todo = ~dead & ((RG == 64 ? ~0ull : ((1ull << RG) - 1ull)) << r0);
            while (todo != 0ull) {
                int rq[4];
                const int cnt = min(4, __popcll(todo));
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    rq[q] = todo != 0ull ? __builtin_ctzll(todo) : rq[0];                                      // (short pass: row rq[0] again, harmless)
                    todo &= todo - 1ull;
                }
                IouTest t4[4];
                uint64_t gt[4], sure = thr > 0.f ? ~0ull : 0ull;
#pragma unroll
                for (int q = 0; q < 4; ++q) {                                                                  // every comparison lands in a scalar mask
                    const f32x4 a = *reinterpret_cast<const f32x4*>(rowbox + rq[q] * 8);
                    t4[q] = iou_test(a, rowbox[rq[q] * 8 + 4], b, area_b, thr);
                    gt[q] = __ballot(t4[q].d > 0.f);
                    sure &= __ballot(fabsf(t4[q].d) > 1e-5f * t4[q].t) & __ballot(t4[q].t > 1e-20f);
                }
                if (sure != ~0ull) {                                                                           // rare: the exact reference expression
#pragma unroll
                    for (int q = 0; q < 4; ++q) gt[q] = __ballot(t4[q].inter / t4[q].uni >= thr);              // Losses.py:51 (NaN >= thr is false)
                }
                uint64_t mine = 0;
                int myr = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint64_t m = gt[q] & col_ok;
                    if (diag) m &= bit0 + rq[q] == 63 ? 0ull : (~0ull << (bit0 + rq[q] + 1));                  // j > i
                    if (lane == q) { mine = m; myr = rq[q]; }
                }
                if (lane < cnt) chunk[myr * nw + w] = mine;
            }
        }
#endif
        __syncthreads();
        // ---- B ------------------------------------------------------------------------------------------
        if (wave == 0) {
            const int i = base + lane;
            const bool live = lane < CR && i < n && !((rem_start >> (bit0 + lane)) & 1ull);
            const uint64_t d = live ? chunk[lane * nw + R] : 0ull;
            const unsigned dlo = (unsigned)d, dhi = (unsigned)(d >> 32);
            uint64_t rem = rem_start, keep = 0;
#ifndef NMS_PROBE_NO_B
            // rows still standing (aligned to the chunk: bit r = row base + r); each kept row removes what its diagonal word names
            uint64_t avail = ~(rem_start >> bit0) & (CR == 64 ? ~0ull : ((1ull << CR) - 1ull));
            if (n - base < CR) avail &= ~(~0ull << (n - base));
            while (avail != 0ull) {                                   // one pass per KEPT row (scalar: find-first-set, two v_readlane, and-not, or)
                const int r = __builtin_ctzll(avail);
                const uint64_t row = ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)dhi, r) << 32) |
                                     (uint64_t)(unsigned)__builtin_amdgcn_readlane((int)dlo, r);
                keep |= 1ull << r;
                rem |= row;
                avail &= ~((row >> bit0) | (1ull << r));
            }
#endif
            if (lane == 0) {
                removed[R] = rem;
                s_keep = keep;
            }
        }
        __syncthreads();
        // ---- C ------------------------------------------------------------------------------------------
        const uint64_t keep = s_keep;
#ifndef NMS_PROBE_NO_C
        for (int w = R + 1 + wave; w < nw; w += NT / 64) {          // a wave per later word: lane r brings row r's word if the row was kept
            uint64_t v = (lane < CR && ((keep >> lane) & 1ull)) ? chunk[lane * nw + w] : 0ull;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned lo = __shfl_xor((unsigned)v, o, 64), hi = __shfl_xor((unsigned)(v >> 32), o, 64);
                v |= ((uint64_t)hi << 32) | lo;
            }
            if (lane == 0) removed[w] |= v;
        }
#endif
        __syncthreads();
    }
    // ---- ordered compaction of the survivors (sorted position + probability bits, for the top-k block) -----------------
    if (tid == 0) running = 0;
    __syncthreads();
    for (int j0 = 0; j0 < n; j0 += NT) {
        const int j = j0 + tid;
        const int keep = (j < n && !((removed[j >> 6] >> (j & 63)) & 1ull)) ? 1 : 0;
        int incl = keep;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int off = running;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        if (keep) {
            kept_pos[(size_t)c * P + off + incl - 1] = j;
            kept_prob[(size_t)c * P + off + incl - 1] = __float_as_uint(s_prob[(size_t)c * P + j]);
        }
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < NT / 64; ++w) t += wave_tot[w];
            running += t;
        }
        __syncthreads();
    }
    if (tid == 0) kept_cnt[c] = running;
}

struct TopkArgs {
    const float* s_boxes; const int32_t* s_idx; const uint32_t* kept_prob; const int32_t* kept_pos; const int32_t* kept_cnt;
    uint32_t* k_prob; int32_t* k_src;           // scratch of this block: class-major compact list of the survivors (prob bits, c*P + sorted position)
    int P, C1, top_k, kp_cap; const float* wh;  // kp_cap: survivors whose probability bits fit the block's LDS; wh: (B,2) device array of (img_w, img_h)
    float* boxes; int64_t* classes; float* probs; int32_t* prior_ids; int32_t* count;
};

// One block.  total <= top_k: everything is emitted in class-major order (Losses.py:71-73).  Otherwise
// (Losses.py:77-81) the top_k-th largest probability is found by radix select on the float bits (probabilities
// are positive), ties at the threshold are taken in class-major order, and the <= top_k selected entries are
// ranked among themselves (prob descending, class-major position ascending) in LDS.
__global__ __launch_bounds__(NB_T) void topk_emit_kernel(TopkArgs a) {
    {
        const size_t img = blockIdx.z, K = (size_t)a.C1 * a.P;
        a.s_boxes += img * K * 4; a.s_idx += img * K; a.k_prob += img * K; a.k_src += img * K;
        a.kept_prob += img * K; a.kept_pos += img * K; a.kept_cnt += img * (a.C1 + 1);
        a.wh += img * 2;
        a.boxes += img * a.top_k * 4; a.classes += img * a.top_k; a.probs += img * a.top_k; a.prior_ids += img * a.top_k; a.count += img;
    }
    const float img_w = a.wh[0], img_h = a.wh[1];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* sel_prob = reinterpret_cast<uint32_t*>(smem);                 // [top_k]
    int32_t* sel_pos = reinterpret_cast<int32_t*>(smem) + a.top_k;          // [top_k]
    __shared__ int hist[256];
    __shared__ int wsum_gt[NB_T / 64], wsum_eq[NB_T / 64];
    __shared__ int bc[3];
    __shared__ int offs[257];                                          // class-major exclusive offsets of the survivors (C1 <= 255)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave == 0) {                                                    // exclusive scan of the per-class counts (C1 <= 255: four per lane)
        int cnt[4], s4 = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = lane * 4 + e;
            cnt[e] = c < a.C1 ? a.kept_cnt[c] : 0;
            s4 += cnt[e];
        }
        int incl = s4;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        int o = incl - s4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = lane * 4 + e;
            if (c <= a.C1) offs[c] = o;
            o += cnt[e];
        }
    }
    __syncthreads();
    const int total = offs[a.C1];
    // the survivors' probability bits stay in LDS for the six passes below when they fit (else they are read back from k_prob)
    uint32_t* kp_lds = reinterpret_cast<uint32_t*>(smem) + 2 * a.top_k;
    const bool in_lds = total <= a.kp_cap;
    auto kp = [&](int g) -> uint32_t { return in_lds ? kp_lds[g] : a.k_prob[g]; };
    // the compact class-major list (the two kernels offsets / gather_kept of the earlier version): written and read by this block only;
    // one entry per thread and pass, its class found by bisection of the 21 offsets (20 classes over 16 waves was two rounds of
    // dependent loads for the last four classes)
    for (int g = tid; g < total; g += NB_T) {
        int lo = 0, hi = a.C1 - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (offs[mid] <= g) lo = mid; else hi = mid - 1;
        }
        const int src = lo * a.P + (g - offs[lo]);
        const uint32_t u = a.kept_prob[src];
        if (in_lds) kp_lds[g] = u; else a.k_prob[g] = u;
        a.k_src[g] = lo * a.P + a.kept_pos[src];
    }
    __threadfence_block();
    __syncthreads();
    auto emit = [&](int gpos, int slot) {
        const int src = a.k_src[gpos];
        const f32x4 b = *reinterpret_cast<const f32x4*>(a.s_boxes + (size_t)src * 4);
        f32x4 o;
        o[0] = b[0] * img_w; o[1] = b[1] * img_h; o[2] = b[2] * img_w; o[3] = b[3] * img_h;    // Losses.py:89
        *reinterpret_cast<f32x4*>(a.boxes + (size_t)slot * 4) = o;
        a.classes[slot] = src / a.P;
        a.probs[slot] = __uint_as_float(kp(gpos));
        a.prior_ids[slot] = a.s_idx[src];
    };
    if (tid == 0) *a.count = total > a.top_k ? a.top_k : total;
    if (total <= a.top_k) {
        for (int g = tid; g < total; g += NB_T) emit(g, g);
        return;
    }
    // ---- radix select of the top_k-th largest value --------------------------------------------------
    uint32_t prefix = 0, mask = 0;
    int remaining = a.top_k;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int b = tid; b < 256; b += NB_T) hist[b] = 0;
        __syncthreads();
        for (int g = tid; g < total; g += NB_T) {
            const uint32_t u = kp(g);
            if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1);
        }
        __syncthreads();
        if (wave == 0) {              // the digit d with count(digits > d) < remaining <= count(digits >= d): suffix sums over four bins per lane
            const int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
            const int s4 = h0 + h1 + h2 + h3;
            int suf = s4;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_down(suf, o, 64);
                if (lane + o < 64) suf += t;
            }
            const int above = suf - s4;
            if (above < remaining && suf >= remaining) {              // exactly one lane
                int r = remaining - above, d = 4 * lane + 3;
                if (h3 < r) { r -= h3; d = 4 * lane + 2; if (h2 < r) { r -= h2; d = 4 * lane + 1; if (h1 < r) { r -= h1; d = 4 * lane; } } }
                bc[0] = d;
                bc[1] = r;
            }
            if (lane == 0 && suf < remaining) {                       // fewer matching entries than asked for (cannot happen: total > top_k)
                bc[0] = 0;
                bc[1] = remaining - (suf - h0);
            }
        }
        __syncthreads();
        prefix |= (uint32_t)bc[0] << shift;
        mask |= 255u << shift;
        remaining = bc[1];
        __syncthreads();
    }
    const uint32_t T = prefix;            // take everything > T and the first `remaining` entries == T
    // ---- ordered selection: thread t owns the contiguous slice [t*CH, (t+1)*CH) ----------------------------
    const int CH = (total + NB_T - 1) / NB_T;
    const int g0 = tid * CH, g1 = min(total, g0 + CH);
    int n_gt = 0, n_eq = 0;
    for (int g = g0; g < g1; ++g) {
        const uint32_t u = kp(g);
        n_gt += u > T;
        n_eq += u == T;
    }
    int i_gt = n_gt, i_eq = n_eq;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t1 = __shfl_up(i_gt, o, 64), t2 = __shfl_up(i_eq, o, 64);
        if (lane >= o) { i_gt += t1; i_eq += t2; }
    }
    if (lane == 63) { wsum_gt[wave] = i_gt; wsum_eq[wave] = i_eq; }
    __syncthreads();
    int off_gt = 0, off_eq = 0, all_gt = 0;
    for (int w2 = 0; w2 < NB_T / 64; ++w2) {
        if (w2 < wave) { off_gt += wsum_gt[w2]; off_eq += wsum_eq[w2]; }
        all_gt += wsum_gt[w2];
    }
    int r_gt = off_gt + i_gt - n_gt;      // entries > T before this slice
    int r_eq = off_eq + i_eq - n_eq;      // entries == T before this slice
    for (int g = g0; g < g1; ++g) {
        const uint32_t u = kp(g);
        int idx = -1;
        if (u > T) idx = r_gt++;
        else if (u == T) { if (r_eq < remaining) idx = all_gt + r_eq; ++r_eq; }
        if (idx >= 0) { sel_prob[idx] = u; sel_pos[idx] = g; }      // exactly top_k entries in all
    }
    __syncthreads();
    // ---- rank inside the selection ------------------------------------------------------------------------
    for (int e = tid; e < a.top_k; e += NB_T) {
        const uint32_t u = sel_prob[e];
        const int g = sel_pos[e];
        int rank = 0;
        for (int f = 0; f < a.top_k; ++f) rank += (sel_prob[f] > u) || (sel_prob[f] == u && sel_pos[f] < g);
        emit(g, rank);
    }
}

__global__ void set_wh_kernel(float* wh, float w, float h) { wh[0] = w; wh[1] = h; }

}  // namespace

extern "C" size_t ssd_decode_nms_batch_workspace(int B, int P, int n_classes) {
    if (B <= 0 || P <= 0 || n_classes < 2) return 0;
    return carve(nullptr, P, n_classes, B).bytes + 256;
}
extern "C" size_t ssd_decode_nms_workspace(int P, int n_classes) { return ssd_decode_nms_batch_workspace(1, P, n_classes); }

extern "C" int ssd_decode_nms_batch(const float* l_, const float* c_, const float* priors_cxcywh, const float* img_wh, int B, int P,
                                    int n_classes, float min_score, float iou_threshold, int top_k, float* boxes, int64_t* classes,
                                    float* probs, int32_t* prior_ids, int32_t* count, void* workspace, size_t workspace_bytes, void* stream) {
    if (!l_ || !c_ || !priors_cxcywh || !img_wh || !boxes || !classes || !probs || !prior_ids || !count || !workspace) return SSD_ERR_NULL;
    if (B <= 0 || B > 65535 || P <= 0 || P > 100000 || n_classes < 2 || n_classes > 256 || top_k <= 0 || top_k > 4096) return SSD_ERR_BAD_SHAPE;
    if (!ssd_aligned16(l_) || !ssd_aligned16(priors_cxcywh) || !ssd_aligned16(boxes) || !ssd_aligned16(workspace)) return SSD_ERR_ALIGN;
    if (workspace_bytes < ssd_decode_nms_batch_workspace(B, P, n_classes)) return SSD_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const NmsWs w = carve(workspace, P, n_classes, B);
    const int C1 = n_classes - 1;
    if (hipMemsetAsync(w.cand_cnt, 0, (size_t)n_classes * 4 * B, st) != hipSuccess) return SSD_ERR_LAUNCH;     // the candidate counters of decode_compact
    hipLaunchKernelGGL(decode_compact_kernel, dim3(ssd_cdiv(P, 256), 1, B), dim3(256), (size_t)256 * n_classes * 4, st, l_, c_, priors_cxcywh, P,
                       n_classes, min_score, w.boxes, w.keys, w.cand_cnt);
    SSD_CHECK_LAUNCH();
    hipLaunchKernelGGL(rank_scatter_kernel, dim3(ssd_cdiv(P, 256), C1, B), dim3(256), 0, st, w.keys, w.cand_cnt, w.boxes, P, w.s_boxes, w.s_prob, w.s_idx);
    SSD_CHECK_LAUNCH();
    // Block size and LDS by batch: a single image is a latency problem (20 workgroups on 256 CUs: 16 waves each, 64-row chunks whatever
    // the candidate count); a batch is a throughput problem -- 8-wave workgroups with a 37 KB chunk buffer fit four to a CU, so up to
    // 1024 (image, class) problems are resident at once instead of two rounds of 512 (64-row chunks up to 4608 candidates per class).
    const int nwcap = (P + 63) / 64;
    const bool wide = (size_t)C1 * B <= 128;
    int chunk_words = wide ? 64 * nwcap : (32 * nwcap > 4608 ? 32 * nwcap : 4608);
    const int cap_words = 140 * 1024 / 8 - nwcap;                  // (SSD512's 24564 priors: 64-row chunks up to 4 480 candidates of a class)
    if (chunk_words > cap_words) chunk_words = cap_words;
    if (chunk_words < 32 * nwcap) return SSD_ERR_BAD_SHAPE;
    const size_t lds = (size_t)(nwcap + chunk_words) * 8;
    if (lds > 48 * 1024) {
        static std::atomic<unsigned long long> raised{0};     // one bit per device (common.h): the attribute belongs to (kernel, device)
        int dev;
        if (ssd_attr_needed(raised, dev)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(nms_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(nms_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
                return SSD_ERR_LAUNCH;
            ssd_attr_done(raised, dev);
        }
    }
    if (wide)
        hipLaunchKernelGGL(nms_kernel<1024>, dim3(C1, 1, B), dim3(1024), lds, st, w.s_boxes, w.s_prob, w.cand_cnt, P, iou_threshold, chunk_words, w.kept_pos,
                           w.kept_prob, w.kept_cnt);
    else
        hipLaunchKernelGGL(nms_kernel<512>, dim3(C1, 1, B), dim3(512), lds, st, w.s_boxes, w.s_prob, w.cand_cnt, P, iou_threshold, chunk_words, w.kept_pos,
                           w.kept_prob, w.kept_cnt);
    SSD_CHECK_LAUNCH();
    const int kp_cap = 16384;                                    // 64 KB of probability bits beside the 2 * top_k selection words
    {
        static std::atomic<unsigned long long> raised{0};
        int dev;
        if (ssd_attr_needed(raised, dev)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(topk_emit_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess)
                return SSD_ERR_LAUNCH;
            ssd_attr_done(raised, dev);
        }
    }
    TopkArgs ta{w.s_boxes, w.s_idx, w.kept_prob, w.kept_pos, w.kept_cnt, w.k_prob, w.k_src, P, C1, top_k, kp_cap, img_wh, boxes, classes, probs, prior_ids, count};
    hipLaunchKernelGGL(topk_emit_kernel, dim3(1, 1, B), dim3(NB_T), (size_t)top_k * 8 + (size_t)kp_cap * 4, st, ta);
    SSD_CHECK_LAUNCH();
    return SSD_OK;
}

// one image, image size passed by value (kept in the last 256 bytes of the workspace)
extern "C" int ssd_decode_nms(const float* l_, const float* c_, const float* priors_cxcywh, int P, int n_classes,
                              float min_score, float iou_threshold, int top_k, float img_w, float img_h, float* boxes,
                              int64_t* classes, float* probs, int32_t* prior_ids, int32_t* count, void* workspace,
                              size_t workspace_bytes, void* stream) {
    if (!workspace) return SSD_ERR_NULL;
    const size_t need = ssd_decode_nms_batch_workspace(1, P, n_classes);
    if (need == 0) return SSD_ERR_BAD_SHAPE;
    if (workspace_bytes < need) return SSD_ERR_WORKSPACE;
    float* wh = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + need - 256);
    hipLaunchKernelGGL(set_wh_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, wh, img_w, img_h);
    SSD_CHECK_LAUNCH();
    return ssd_decode_nms_batch(l_, c_, priors_cxcywh, wh, 1, P, n_classes, min_score, iou_threshold, top_k, boxes, classes, probs,
                                prior_ids, count, workspace, workspace_bytes, stream);
}
