// SGDet front end after the backbone (SURVEY a-12): RPN proposal decode / top-k / NMS and the RoI-heads
// post-processing of [3P] torchvision FasterRCNN in eval mode, as called at sgg_models/rel_model_base.py:210-213.
// The dense parts (RPN 3x3 conv, 1x1 heads, box head fc6/fc7, predictor) run on the MFMA GEMM / conv kernels;
// this file holds the HBM/latency-bound rest.  Sorting is a plain library op (rocPRIM segmented radix sort).
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace {

constexpr float BBOX_XFORM_CLIP = 4.135166556742356f;  // log(1000/16)

struct Box {
    float x1, y1, x2, y2;
};

// [3P] BoxCoder.decode_single
__device__ __forceinline__ Box decode(const float* d, const Box& a, float wx, float wy, float ww, float wh) {
    const float w = a.x2 - a.x1, h = a.y2 - a.y1;
    const float cx = a.x1 + 0.5f * w, cy = a.y1 + 0.5f * h;
    const float dx = d[0] / wx, dy = d[1] / wy;
    const float dw = fminf(d[2] / ww, BBOX_XFORM_CLIP), dh = fminf(d[3] / wh, BBOX_XFORM_CLIP);
    const float pcx = dx * w + cx, pcy = dy * h + cy;
    const float pw = expf(dw) * w, ph = expf(dh) * h;
    return Box{pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw, pcy + 0.5f * ph};
}

__device__ __forceinline__ Box clip(Box b, float h, float w) {
    b.x1 = fminf(fmaxf(b.x1, 0.f), w);
    b.x2 = fminf(fmaxf(b.x2, 0.f), w);
    b.y1 = fminf(fmaxf(b.y1, 0.f), h);
    b.y2 = fminf(fmaxf(b.y2, 0.f), h);
    return b;
}

// RPN: head[(b*HW + pos), 0..A) = objectness logits, [A .. 5A) = deltas (a*4 + c).  anchors in (y,x,a) order.
__global__ __launch_bounds__(256) void rpn_decode_kernel(const float* __restrict__ head, int ldh, const float* __restrict__ base,
                                                         int A, int Hf, int Wf, float sy, float sx, int B,
                                                         float* __restrict__ boxes, float* __restrict__ scores) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long per = (long)Hf * Wf * A;
    if (i >= per * B) return;
    const int b = (int)(i / per);
    const long r = i - (long)b * per;
    const int a = (int)(r % A);
    const int pos = (int)(r / A);
    const int y = pos / Wf, x = pos - y * Wf;
    const float* hrow = head + ((long)b * Hf * Wf + pos) * ldh;
    const Box an{base[a * 4] + x * sx, base[a * 4 + 1] + y * sy, base[a * 4 + 2] + x * sx, base[a * 4 + 3] + y * sy};
    const Box p = decode(hrow + A + a * 4, an, 1.f, 1.f, 1.f, 1.f);
    float* o = boxes + i * 4;
    o[0] = p.x1; o[1] = p.y1; o[2] = p.x2; o[3] = p.y2;
    scores[i] = hrow[a];
}

// v[i] = position of element i inside its segment (equal-length segments, or a scan over the <= 256 offsets)
__global__ __launch_bounds__(256) void iota_kernel(int* __restrict__ v, long n, int seg_len, const int* __restrict__ seg_off,
                                                   int nseg) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (seg_len > 0) {
        v[i] = (int)(i % seg_len);
    } else {
        int s = 0;
        while (s + 1 < nseg && seg_off[s + 1] <= i) ++s;
        v[i] = (int)(i - seg_off[s]);
    }
}

// take the first `take` entries of each sorted segment: out box = clip(boxes[seg][idx]); valid = finite score &&
// w,h >= min_size && (score > score_min)
__global__ __launch_bounds__(256) void gather_topk_kernel(const float* __restrict__ keys, const int* __restrict__ vals,
                                                          const int* __restrict__ seg_off, const float* __restrict__ boxes,
                                                          const int* __restrict__ labels_in, const float* __restrict__ img_hw,
                                                          int B, int take, float min_size, float* __restrict__ out_boxes,
                                                          float* __restrict__ out_scores, int* __restrict__ out_labels,
                                                          unsigned char* __restrict__ valid) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * take) return;
    const int b = i / take, r = i - b * take;
    const int s0 = seg_off[b], s1 = seg_off[b + 1];
    bool ok = s0 + r < s1;
    float sc = -INFINITY;
    Box bx{0, 0, 0, 0};
    int lab = 0;
    if (ok) {
        sc = keys[s0 + r];
        const long src = (long)s0 + vals[s0 + r];
        const float* p = boxes + src * 4;
        bx = clip(Box{p[0], p[1], p[2], p[3]}, img_hw[b * 2], img_hw[b * 2 + 1]);
        if (labels_in) lab = labels_in[src];
        ok = (sc > -INFINITY) && (bx.x2 - bx.x1 >= min_size) && (bx.y2 - bx.y1 >= min_size);
    }
    float* o = out_boxes + (long)i * 4;
    o[0] = bx.x1; o[1] = bx.y1; o[2] = bx.x2; o[3] = bx.y2;
    out_scores[i] = sc;
    if (out_labels) out_labels[i] = lab;
    valid[i] = ok ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// Top-k of every image's candidates, best first -- what sgg_segmented_sort_desc + sgg_gather_topk produce, without sorting the
// candidates that are not wanted (RPN: the 1000 best of 21 660 anchors; detections: the 4096 best of up to 150 000 (proposal, class)
// pairs per image -- round 2-3 sorted all 1.2 M of them with rocPRIM: 0.65 ms of a 6.7-ms step).  One 1024-thread workgroup per image:
//   1. float -> order-preserving u32 key; 4-pass MSB radix SELECT (256-bin LDS histogram per pass, only keys that match the prefix
//      found so far are counted) -> the threshold key thr and how many candidates equal to it are still needed;
//   2. candidates with key > thr (any order) and the `need` lowest-indexed ones with key == thr (an ordered block scan, only when
//      more tie than are needed) go into LDS as 64-bit (key, ~index) words -- all distinct;
//   3. bitonic sort of those <= 4096 words, descending: score descending, index ascending = the order of a stable descending sort;
//   4. the gather of gather_topk_kernel (clip to the image, validity).
// Each pass streams the image's scores once from L2 (600 KB for 150 000 candidates).
// ------------------------------------------------------------------------------------------------
constexpr int TOPK_MAX = 4096;
__device__ __forceinline__ unsigned topk_key(float f) {
    if (f != f) return 0u;                                       // NaN: never selected before a real score
    const unsigned u = __float_as_uint(f == 0.f ? 0.f : f);      // (-0 and +0 are ONE key, as for a comparison sort and for rocPRIM's float sort)
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__global__ __launch_bounds__(1024) void topk_gather_kernel(const float* __restrict__ scores, const int* __restrict__ seg_off,
                                                           const float* __restrict__ boxes, const int* __restrict__ labels_in,
                                                           const float* __restrict__ img_hw, int take, float min_size,
                                                           float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                                           int* __restrict__ out_labels, unsigned char* __restrict__ valid) {
    __shared__ unsigned long long sel[TOPK_MAX];
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_eq, s_cnt, s_wave[16];
    __shared__ int s_need;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int s0 = seg_off[b], n = seg_off[b + 1] - s0;
    const int T = min(min(take, n), TOPK_MAX);
    const float* sc = scores + s0;
    unsigned prefix = 0u, mask = 0u;
    int need = T;
    for (int pass = 3; pass >= 0 && T > 0; --pass) {
        const int shift = pass * 8;
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        // runs of one digit are counted in a register and added once: with a score threshold most of an image's 150 K class candidates are
        // -inf -- one key, one bin, i.e. every thread's every element an atomic on the same LDS word (round 6: 0.85 ms per SGDet step)
        unsigned run_d = 0xffffffffu, run_n = 0u;
        for (int i = tid; i < n; i += 1024) {
            const unsigned k = topk_key(sc[i]);
            if ((k & mask) != prefix) continue;
            const unsigned d = (k >> shift) & 255u;
            if (d == run_d) {
                ++run_n;
            } else {
                if (run_n) atomicAdd(&hist[run_d], run_n);
                run_d = d;
                run_n = 1u;
            }
        }
        if (run_n) atomicAdd(&hist[run_d], run_n);
        __syncthreads();
        if (tid == 0) {
            int cum = 0, d = 255;
            for (; d > 0; --d) {
                if (cum + (int)hist[d] >= need) break;
                cum += (int)hist[d];
            }
            s_prefix = prefix | ((unsigned)d << shift);
            s_need = need - cum;
            s_eq = hist[d];
        }
        __syncthreads();
        prefix = s_prefix;
        need = s_need;
        mask |= 0xffu << shift;
    }
    const unsigned thr = prefix;
    const int eq_total = (int)s_eq;                              // candidates with key == thr; `need` (>= 1) of them are taken
    if (tid == 0) s_cnt = 0u;
    __syncthreads();
    // (a) every candidate above the threshold: exactly T - need of them, in arrival order (the sort below orders them)
    for (int i = tid; i < n && T > 0; i += 1024) {
        const unsigned k = topk_key(sc[i]);
        if (k > thr) sel[atomicAdd(&s_cnt, 1u)] = ((unsigned long long)k << 32) | (unsigned long long)(0xffffffffu - (unsigned)i);
    }
    __syncthreads();
    const int base = T - need;
    if (tid == 0) s_cnt = 0u;
    __syncthreads();
    // (b) `need` of the candidates AT the threshold: the lowest indices (a stable sort's order) -- by an ordered block scan when more
    // tie than are needed; when all of them are taken, or the tie is at -inf (entries that are invalid whichever they are), in
    // arrival order
    const bool ordered_ties = T > 0 && eq_total > need && thr != topk_key(-INFINITY);
    if (!ordered_ties) {
        for (int i = tid; i < n && T > 0; i += 1024) {
            if (topk_key(sc[i]) == thr && *(volatile unsigned*)&s_cnt < (unsigned)need) {      // (once `need` are taken nobody queues up for the counter)
                const unsigned c = atomicAdd(&s_cnt, 1u);
                if (c < (unsigned)need) sel[base + c] = ((unsigned long long)thr << 32) | (unsigned long long)(0xffffffffu - (unsigned)i);
            }
        }
    } else {
        int taken = 0;                                           // (uniform: every thread keeps the same count)
        for (int c0 = 0; c0 < n && taken < need; c0 += 1024) {
            const int i = c0 + tid;
            const bool eq = i < n && topk_key(sc[i]) == thr;
            const unsigned long long bal = __ballot(eq);
            if (lane == 0) s_wave[wv] = (unsigned)__popcll(bal);
            __syncthreads();
            int before = 0, total = 0;
            for (int w = 0; w < 16; ++w) {
                if (w < wv) before += (int)s_wave[w];
                total += (int)s_wave[w];
            }
            const int rank = taken + before + (int)__popcll(bal & ((1ull << lane) - 1ull));
            if (eq && rank < need) sel[base + rank] = ((unsigned long long)thr << 32) | (unsigned long long)(0xffffffffu - (unsigned)i);
            taken += total;
            __syncthreads();
        }
    }
    __syncthreads();
    // pad to a power of two with the lowest word, sort descending
    int P2 = 1;
    while (P2 < T) P2 <<= 1;
    for (int i = T + tid; i < P2; i += 1024) sel[i] = 0ull;
    __syncthreads();
    for (int k = 2; k <= P2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P2; i += 1024) {
                const int p = i ^ j;
                if (p > i) {
                    const unsigned long long a = sel[i], c = sel[p];
                    const bool desc = (i & k) == 0;
                    if (desc ? a < c : a > c) {
                        sel[i] = c;
                        sel[p] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (int r = tid; r < take; r += 1024) {
        bool ok = r < T;
        float scv = -INFINITY;
        Box bx{0, 0, 0, 0};
        int lab = 0;
        if (ok) {
            const unsigned idx = 0xffffffffu - (unsigned)(sel[r] & 0xffffffffull);
            scv = sc[idx];
            const long src = (long)s0 + idx;
            const float* p = boxes + src * 4;
            bx = clip(Box{p[0], p[1], p[2], p[3]}, img_hw[b * 2], img_hw[b * 2 + 1]);
            if (labels_in) lab = labels_in[src];
            ok = (scv > -INFINITY) && (bx.x2 - bx.x1 >= min_size) && (bx.y2 - bx.y1 >= min_size);
        }
        const long o = (long)b * take + r;
        float* ob = out_boxes + o * 4;
        ob[0] = bx.x1; ob[1] = bx.y1; ob[2] = bx.x2; ob[3] = bx.y2;
        out_scores[o] = scv;
        if (out_labels) out_labels[o] = lab;
        valid[o] = ok ? 1 : 0;
    }
}

// NMS suppression bit-matrix: mask[b][i][w] bit j = (j > i) && IoU(i,j) > thresh [&& same label]
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, const int* __restrict__ labels,
                                                      const unsigned char* __restrict__ valid, int n, int nw, float thresh,
                                                      unsigned long long* __restrict__ mask) {
    const int b = blockIdx.z, i = blockIdx.y, w = blockIdx.x, lane = threadIdx.x;
    const int j = w * 64 + lane;
    const float* bi = boxes + ((long)b * n + i) * 4;
    bool bit = false;
    if (j < n && j > i && valid[(long)b * n + i] && valid[(long)b * n + j] && (!labels || labels[(long)b * n + i] == labels[(long)b * n + j])) {
        const float* bj = boxes + ((long)b * n + j) * 4;
        const float a1 = (bi[2] - bi[0]) * (bi[3] - bi[1]), a2 = (bj[2] - bj[0]) * (bj[3] - bj[1]);
        const float iw = fmaxf(fminf(bi[2], bj[2]) - fmaxf(bi[0], bj[0]), 0.f);
        const float ih = fmaxf(fminf(bi[3], bj[3]) - fmaxf(bi[1], bj[1]), 0.f);
        const float inter = iw * ih;
        bit = inter / (a1 + a2 - inter) > thresh;
    }
    const unsigned long long m = __ballot(bit);
    if (lane == 0) mask[((long)b * n + i) * nw + w] = m;
}

// greedy scan, one wave per image: lane l owns word l (+64..) of the removed-set.  The validity flags enter the removed-set up front (a
// ballot per 64 candidates): the loop itself touches memory only when a box is KEPT (its suppression row) -- with a flag load per
// candidate the scan of 1000 RPN candidates cost 182 us (one dependent global load per iteration), not the rows.
__global__ __launch_bounds__(64) void nms_scan_kernel(const unsigned long long* __restrict__ mask,
                                                      const unsigned char* __restrict__ valid, int n, int nw, int max_keep,
                                                      int* __restrict__ keep_idx, int* __restrict__ keep_cnt) {
    const int b = blockIdx.x, lane = threadIdx.x;
    unsigned long long removed[2] = {0ull, 0ull};   // nw <= 128
    for (int w = 0; w < nw; ++w) {
        const int j = w * 64 + lane;
        const unsigned long long inv = __ballot(!(j < n && valid[(long)b * n + j]));
        if (lane == (w & 63)) removed[w >> 6] |= inv;
    }
    int cnt = 0;
    for (int i = 0; i < n && cnt < max_keep; ++i) {
        const int w = i >> 6;
        const unsigned long long rw = w < 64 ? __shfl(removed[0], w, 64) : __shfl(removed[1], w - 64, 64);
        if ((rw >> (i & 63)) & 1ull) continue;           // suppressed or invalid (wave-uniform)
        if (lane == 0) keep_idx[(long)b * max_keep + cnt] = i;
        ++cnt;
        const unsigned long long* row = mask + ((long)b * n + i) * nw;
        if (lane < nw) removed[0] |= row[lane];
        if (lane + 64 < nw) removed[1] |= row[lane + 64];
    }
    if (lane == 0) keep_cnt[b] = cnt;
}

// The same scan with the image's whole suppression matrix in LDS first (n x nw words <= 144 KB: the RPN's 1000 candidates = 125 KB): a kept
// box's row is then an LDS read (~100 cycles) instead of a dependent global load (~1.5 us) -- with a detector whose proposals survive the NMS
// (1000 kept of 1000: BASELINE configs[2] at its size) the global-memory form spent ~0.4 ms per step waiting for rows, one after the other.
// 256 threads copy, wave 0 scans; same order, same result.
__global__ __launch_bounds__(256) void nms_scan_lds_kernel(const unsigned long long* __restrict__ mask,
                                                          const unsigned char* __restrict__ valid, int n, int nw, int max_keep,
                                                          int* __restrict__ keep_idx, int* __restrict__ keep_cnt) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long rows[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const unsigned long long* src = mask + (long)b * n * nw;
    for (long i = tid; i < (long)n * nw; i += 256) rows[i] = src[i];
    __syncthreads();
    if (tid >= 64) return;
    unsigned long long removed[2] = {0ull, 0ull};   // nw <= 128
    for (int w = 0; w < nw; ++w) {
        const int j = w * 64 + lane;
        const unsigned long long inv = __ballot(!(j < n && valid[(long)b * n + j]));
        if (lane == (w & 63)) removed[w >> 6] |= inv;
    }
    int cnt = 0;
    for (int i = 0; i < n && cnt < max_keep; ++i) {
        const int w = i >> 6;
        const unsigned long long rw = w < 64 ? __shfl(removed[0], w, 64) : __shfl(removed[1], w - 64, 64);
        if ((rw >> (i & 63)) & 1ull) continue;           // suppressed or invalid (wave-uniform)
        if (lane == 0) keep_idx[(long)b * max_keep + cnt] = i;
        ++cnt;
        const unsigned long long* row = rows + (long)i * nw;
        if (lane < nw) removed[0] |= row[lane];
        if (lane + 64 < nw) removed[1] |= row[lane + 64];
    }
    if (lane == 0) keep_cnt[b] = cnt;
}

// Greedy NMS when only a few boxes are kept (detections: <= 50 of 4096 candidates): the suppression rows of the KEPT boxes only,
// computed when a box is kept -- round 2-3 built the whole n x n bit-matrix first (nms_mask_kernel: 2 M workgroups for n = 4096,
// 0.46 ms) and scanned it with one wave (0.2 ms).  One 1024-thread workgroup per image, boxes / labels in LDS; the alive set is a
// bitmap of 64 words (wave w owns words w, w+16, w+32, w+48: lane = bit); per kept box: wave 0 finds the next alive candidate
// (64 lanes x one word each, a min-reduce), everybody tests its candidates behind it against that box, each wave clears the
// suppressed bits of its own words with a ballot -- no atomics, the order of kept boxes is the score order.
constexpr int NMS_LAZY_N = 4096;
__global__ __launch_bounds__(1024) void nms_lazy_kernel(const float* __restrict__ boxes, const int* __restrict__ labels,
                                                        const unsigned char* __restrict__ valid, int n, float thresh, int max_keep,
                                                        int* __restrict__ keep_idx, int* __restrict__ keep_cnt) {
    __shared__ float4 sbox[NMS_LAZY_N];
    __shared__ int slab[NMS_LAZY_N];
    __shared__ unsigned long long alive[64];
    __shared__ int s_next;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int j = tid; j < NMS_LAZY_N; j += 1024) {
        if (j < n) {
            sbox[j] = *reinterpret_cast<const float4*>(boxes + ((long)b * n + j) * 4);
            slab[j] = labels ? labels[(long)b * n + j] : 0;
        }
    }
    for (int q = 0; q < 4; ++q) {                              // this wave's four words of the alive bitmap
        const int w = wv + 16 * q, j = w * 64 + lane;
        const unsigned long long m = __ballot(j < n && valid[(long)b * n + j]);
        if (lane == 0) alive[w] = m;
    }
    __syncthreads();
    int cnt = 0, cur = 0;
    while (cnt < max_keep) {
        if (wv == 0) {                                           // first alive candidate at or behind `cur`
            unsigned long long m = alive[lane];
            const int w0 = cur >> 6;
            if (lane < w0) m = 0ull;
            else if (lane == w0) m &= ~0ull << (cur & 63);
            int first = m ? lane * 64 + __builtin_ctzll(m) : 0x7fffffff;
            for (int off = 32; off > 0; off >>= 1) first = min(first, __shfl_xor(first, off, 64));
            if (lane == 0) s_next = first;
        }
        __syncthreads();
        const int i = s_next;
        if (i >= n) break;                                       // (uniform)
        if (tid == 0) keep_idx[(long)b * max_keep + cnt] = i;
        ++cnt;
        cur = i + 1;
        const float4 bi = sbox[i];
        const int li = slab[i];
        const float a1 = (bi.z - bi.x) * (bi.w - bi.y);
        for (int q = 0; q < 4; ++q) {
            const int w = wv + 16 * q, j = w * 64 + lane;
            if (w * 64 + 63 <= i) continue;                      // (uniform per wave: the whole word lies in front of box i)
            bool kill = false;
            if (j > i && j < n && ((alive[w] >> lane) & 1ull) && slab[j] == li) {
                const float4 bj = sbox[j];
                const float a2 = (bj.z - bj.x) * (bj.w - bj.y);
                const float iw = fmaxf(fminf(bi.z, bj.z) - fmaxf(bi.x, bj.x), 0.f);
                const float ih = fmaxf(fminf(bi.w, bj.w) - fmaxf(bi.y, bj.y), 0.f);
                const float inter = iw * ih;
                kill = inter / (a1 + a2 - inter) > thresh;
            }
            const unsigned long long km = __ballot(kill);
            if (lane == 0 && km) alive[w] &= ~km;
        }
        __syncthreads();
    }
    if (tid == 0) keep_cnt[b] = cnt;
}

// compact kept boxes of all images into rois [total,5] = (img, box); offsets[b] = first row of image b
__global__ __launch_bounds__(256) void compact_rois_kernel(const float* __restrict__ boxes, const int* __restrict__ keep_idx,
                                                           const int* __restrict__ keep_cnt, int B, int n, int max_keep,
                                                           float* __restrict__ rois, int* __restrict__ offsets) {
    __shared__ int off[257];
    if (threadIdx.x == 0) {
        int s = 0;
        for (int b = 0; b < B; ++b) {
            off[b] = s;
            s += keep_cnt[b];
        }
        off[B] = s;
        for (int b = 0; b <= B; ++b) offsets[b] = off[b];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < B * max_keep; i += 256) {
        const int b = i / max_keep, r = i - b * max_keep;
        if (r < keep_cnt[b]) {
            const float* p = boxes + ((long)b * n + keep_idx[(long)b * max_keep + r]) * 4;
            float* o = rois + (long)(off[b] + r) * 5;
            o[0] = (float)b; o[1] = p[0]; o[2] = p[1]; o[3] = p[2]; o[4] = p[3];
        }
    }
}

// RoI heads: one wave per RoI.  pred[k, 0..C) = class logits, [C .. 5C) = box regression (c*4 + k).
// cand_score[k*(C-1) + c-1] = softmax prob if > thresh and the decoded, clipped box is >= min_size, else -inf.
__global__ __launch_bounds__(256) void det_candidates_kernel(const float* __restrict__ pred, int ldp, const float* __restrict__ rois,
                                                             int K, int C, const float* __restrict__ img_hw, float thresh,
                                                             float min_size, float* __restrict__ cand_score,
                                                             float* __restrict__ cand_box, int* __restrict__ cand_label) {
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k >= K) return;
    const float* row = pred + (long)k * ldp;
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, row[c]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int c = lane; c < C; c += 64) sum += expf(row[c] - mx);
    sum = wave_sum(sum);
    const float* r = rois + (long)k * 5;
    const int b = (int)r[0];
    const Box prop{r[1], r[2], r[3], r[4]};
    const float ih = img_hw[b * 2], iw = img_hw[b * 2 + 1];
    for (int c = 1 + lane; c < C; c += 64) {
        const float sc = expf(row[c] - mx) / sum;
        const Box bx = clip(decode(row + C + c * 4, prop, 10.f, 10.f, 5.f, 5.f), ih, iw);
        const bool ok = sc > thresh && (bx.x2 - bx.x1 >= min_size) && (bx.y2 - bx.y1 >= min_size);
        const long o = (long)k * (C - 1) + c - 1;
        cand_score[o] = ok ? sc : -INFINITY;
        float* ob = cand_box + o * 4;
        ob[0] = bx.x1; ob[1] = bx.y1; ob[2] = bx.x2; ob[3] = bx.y2;
        cand_label[o] = c;
    }
}

// final gather: dets[b][r] = kept candidate r of image b
__global__ __launch_bounds__(256) void det_output_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                         const int* __restrict__ labels, const int* __restrict__ keep_idx,
                                                         const int* __restrict__ keep_cnt, int B, int n, int max_keep,
                                                         float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                                         int64_t* __restrict__ out_labels) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * max_keep) return;
    const int b = i / max_keep, r = i - b * max_keep;
    if (r >= keep_cnt[b]) return;
    const long src = (long)b * n + keep_idx[(long)b * max_keep + r];
    for (int q = 0; q < 4; ++q) out_boxes[(long)i * 4 + q] = boxes[src * 4 + q];
    out_scores[i] = scores[src];
    out_labels[i] = labels[src];
}

}  // namespace

extern "C" int sgg_rpn_decode(const float* head, int ldh, const float* base_anchors, int A, int Hf, int Wf, float stride_y,
                              float stride_x, int B, float* boxes, float* scores, void* stream) {
    if (!head || !base_anchors || !boxes || !scores || A <= 0 || Hf <= 0 || Wf <= 0 || B <= 0 || ldh < 5 * A) return SGG_ERR_ARG;
    const long total = (long)B * Hf * Wf * A;
    hipLaunchKernelGGL(rpn_decode_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, head, ldh,
                       base_anchors, A, Hf, Wf, stride_y, stride_x, B, boxes, scores);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

// Stable descending sort of each segment [seg_off[s], seg_off[s+1]) of keys (f32); vals_out = position of each sorted
// element inside its segment.  temp == NULL: only *temp_bytes is set (rocPRIM workspace query).  seg_off: device i32[nseg+1].
// seg_len_hint > 0: all segments have that length (cheaper index fill).
extern "C" int sgg_segmented_sort_desc(const float* keys_in, float* keys_out, int* vals_tmp, int* vals_out, int n, int nseg,
                                       const int* seg_off, int seg_len_hint, void* temp, size_t* temp_bytes, void* stream) {
    if (!temp_bytes || n < 0 || nseg <= 0) return SGG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    size_t bytes = temp ? *temp_bytes : 0;
    if (temp) {
        if (!keys_in || !keys_out || !vals_tmp || !vals_out || !seg_off) return SGG_ERR_ARG;
        // vals_tmp = position inside the segment: equal segments (seg_len_hint > 0) use i % len, otherwise the caller's iota
        if (n > 0) hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, vals_tmp, (long)n, seg_len_hint, seg_off, nseg);
    }
    const hipError_t e = rocprim::segmented_radix_sort_pairs_desc(temp, bytes, keys_in, keys_out, vals_tmp, vals_out, (unsigned)n,
                                                                  (unsigned)nseg, seg_off, seg_off + 1, 0, 32, s);
    *temp_bytes = bytes;
    if (e != hipSuccess) return SGG_ERR_LAUNCH;
    return SGG_OK;
}

extern "C" int sgg_gather_topk(const float* keys_sorted, const int* vals_sorted, const int* seg_off, const float* boxes,
                               const int* labels_in, const float* img_hw, int B, int take, float min_size, float* out_boxes,
                               float* out_scores, int* out_labels, unsigned char* valid, void* stream) {
    if (!keys_sorted || !vals_sorted || !seg_off || !boxes || !img_hw || !out_boxes || !out_scores || !valid || B <= 0 || take <= 0)
        return SGG_ERR_ARG;
    hipLaunchKernelGGL(gather_topk_kernel, dim3((B * take + 255) / 256), dim3(256), 0, (hipStream_t)stream, keys_sorted, vals_sorted,
                       seg_off, boxes, labels_in, img_hw, B, take, min_size, out_boxes, out_scores, out_labels, valid);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

// The `take` (<= 4096) best-scoring candidates of every segment, best first (ties: lower index first), gathered like sgg_gather_topk:
// = sgg_segmented_sort_desc + sgg_gather_topk without sorting what is not wanted (radix select + LDS bitonic sort, one workgroup per image)
extern "C" int sgg_topk_gather(const float* scores, const int* seg_off, const float* boxes, const int* labels_in, const float* img_hw, int B,
                               int take, float min_size, float* out_boxes, float* out_scores, int* out_labels, unsigned char* valid,
                               void* stream) {
    if (!scores || !seg_off || !boxes || !img_hw || !out_boxes || !out_scores || !valid || B <= 0 || take <= 0 || take > TOPK_MAX) return SGG_ERR_ARG;
    hipLaunchKernelGGL(topk_gather_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, scores, seg_off, boxes, labels_in, img_hw, take, min_size,
                       out_boxes, out_scores, out_labels, valid);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

// greedy NMS on score-ordered boxes [B,n,4] (n <= 8192): keep_idx i32[B,max_keep], keep_cnt i32[B].
// labels != NULL: class-aware (batched_nms).  mask_ws: u64[B*n*ceil(n/64)] scratch.
extern "C" int sgg_nms(const float* boxes, const int* labels, const unsigned char* valid, int B, int n, float thresh, int max_keep,
                       void* mask_ws, int* keep_idx, int* keep_cnt, void* stream) {
    if (!boxes || !valid || !mask_ws || !keep_idx || !keep_cnt || B <= 0 || n <= 0 || n > 8192 || max_keep <= 0) return SGG_ERR_ARG;
    const int nw = (n + 63) / 64;
    hipStream_t s = (hipStream_t)stream;
    static const char* lazy_off = getenv("SGG_NMS_LAZY");       // "0": always the bit-matrix form (cross-checks)
    if (max_keep <= 128 && n <= NMS_LAZY_N && !(lazy_off && lazy_off[0] == '0')) {
        // few boxes kept (detections): only the kept boxes' suppression rows are ever computed
        hipLaunchKernelGGL(nms_lazy_kernel, dim3(B), dim3(1024), 0, s, boxes, labels, valid, n, thresh, max_keep, keep_idx, keep_cnt);
        SGG_CHECK_LAUNCH();
        return SGG_OK;
    }
    hipLaunchKernelGGL(nms_mask_kernel, dim3(nw, n, B), dim3(64), 0, s, boxes, labels, valid, n, nw, thresh, (unsigned long long*)mask_ws);
    const size_t lds = (size_t)n * nw * 8;
    if (lds <= 144 * 1024) {
        static bool attr_done = false;
        if (!attr_done) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(nms_scan_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024) != hipSuccess)
                return SGG_ERR_LAUNCH;
            attr_done = true;
        }
        hipLaunchKernelGGL(nms_scan_lds_kernel, dim3(B), dim3(256), lds, s, (const unsigned long long*)mask_ws, valid, n, nw, max_keep, keep_idx, keep_cnt);
    } else {
        hipLaunchKernelGGL(nms_scan_kernel, dim3(B), dim3(64), 0, s, (const unsigned long long*)mask_ws, valid, n, nw, max_keep, keep_idx, keep_cnt);
    }
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_compact_rois(const float* boxes, const int* keep_idx, const int* keep_cnt, int B, int n, int max_keep,
                                float* rois, int* offsets, void* stream) {
    if (!boxes || !keep_idx || !keep_cnt || !rois || !offsets || B <= 0 || B > 256) return SGG_ERR_ARG;
    hipLaunchKernelGGL(compact_rois_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, boxes, keep_idx, keep_cnt, B, n, max_keep, rois, offsets);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_det_candidates(const float* pred, int ldp, const float* rois, int K, int C, const float* img_hw,
                                  float score_thresh, float min_size, float* cand_score, float* cand_box, int* cand_label,
                                  void* stream) {
    if (K == 0) return SGG_OK;
    if (!pred || !rois || !img_hw || !cand_score || !cand_box || !cand_label || K < 0 || C < 2 || ldp < 5 * C) return SGG_ERR_ARG;
    hipLaunchKernelGGL(det_candidates_kernel, dim3((K + 3) / 4), dim3(256), 0, (hipStream_t)stream, pred, ldp, rois, K, C, img_hw,
                       score_thresh, min_size, cand_score, cand_box, cand_label);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_det_output(const float* boxes, const float* scores, const int* labels, const int* keep_idx, const int* keep_cnt,
                              int B, int n, int max_keep, float* out_boxes, float* out_scores, int64_t* out_labels, void* stream) {
    if (!boxes || !scores || !labels || !keep_idx || !keep_cnt || !out_boxes || !out_scores || !out_labels || B <= 0) return SGG_ERR_ARG;
    hipLaunchKernelGGL(det_output_kernel, dim3((B * max_keep + 255) / 256), dim3(256), 0, (hipStream_t)stream, boxes, scores, labels,
                       keep_idx, keep_cnt, B, n, max_keep, out_boxes, out_scores, out_labels);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}
