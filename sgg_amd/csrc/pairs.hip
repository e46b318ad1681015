// Candidate-pair indexing (int64 index work, bit-exact with the reference) and the CSR edge lists of the IMP scatter.
//   eval : RelModelBase.get_rel_inds, sgg_models/rel_model_base.py:147-163
//   train: proposal_assignments_gtbox, lib/proposal_assignments_gtbox.py:7-80 (no-sampling path)
// Order-preserving stream compaction: one wave per subject row, ballot / prefix inside the wave, a block scan over
// the row counts.  No atomics decide positions, so results are deterministic.
#include "common.h"

namespace {

__device__ __forceinline__ bool same_img_pair(const int64_t* im, int i, int j) { return i != j && im[i] == im[j]; }

// [3P] torchvision box_iou (no +1) as used by bbox_overlaps, lib/pytorch_misc.py:60-67
__device__ __forceinline__ bool iou_positive(const float* b, int i, int j) {
    const float* p = b + 4 * (long)i;
    const float* q = b + 4 * (long)j;
    const float a1 = (p[2] - p[0]) * (p[3] - p[1]), a2 = (q[2] - q[0]) * (q[3] - q[1]);
    const float w = fmaxf(fminf(p[2], q[2]) - fmaxf(p[0], q[0]), 0.f);
    const float h = fmaxf(fminf(p[3], q[3]) - fmaxf(p[1], q[1]), 0.f);
    const float inter = w * h;
    return inter / (a1 + a2 - inter) > 0.f;
}

__device__ __forceinline__ int lanes_below(unsigned long long mask, int lane) {
    return __popcll(mask & ((1ull << lane) - 1ull));
}

// wave-inclusive prefix sum
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// ---- eval: count (WRITE=false) / write (WRITE=true); grid = N rows, 64 threads
template <bool WRITE>
__global__ __launch_bounds__(64) void pair_eval_kernel(const int64_t* __restrict__ im, const float* __restrict__ boxes, int N,
                                                       int overlap, int* __restrict__ rowoff, int64_t* __restrict__ out,
                                                       int cap) {
    const int i = blockIdx.x, lane = threadIdx.x;
    int base = WRITE ? rowoff[i] : 0;
    for (int j0 = 0; j0 < N; j0 += 64) {
        const int j = j0 + lane;
        bool ok = j < N && same_img_pair(im, i, j);
        if (ok && overlap) ok = iou_positive(boxes, i, j);
        const unsigned long long m = __ballot(ok);
        if (WRITE) {
            const int pos = base + lanes_below(m, lane);
            if (ok && pos < cap) {
                out[3 * (long)pos] = im[i];
                out[3 * (long)pos + 1] = i;
                out[3 * (long)pos + 2] = j;
            }
        }
        base += __popcll(m);
    }
    if (!WRITE && lane == 0) rowoff[i] = base;
}

// exclusive scan of a[0..n) in place, total -> a[n] and *count.  One block of 1024 threads.
__global__ __launch_bounds__(1024) void excl_scan_kernel(int* __restrict__ a, int n, int* __restrict__ count) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int idx = base + tid;
        const int v = idx < n ? a[idx] : 0;
        const int inc = wave_incl_scan(v, lane);
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        const int carry = carry_s;
        if (idx < n) a[idx] = carry + woff + inc - v;
        __syncthreads();
        if (tid == 1023) carry_s = carry + woff + inc;
        __syncthreads();
    }
    if (tid == 0) {
        a[n] = carry_s;
        if (count) *count = carry_s;
    }
}

// ---- train
// cnt[s*N+o] += 1 per FG relation; rank[r] = # earlier FG relations on the same pair (deterministic duplicate order)
__global__ __launch_bounds__(256) void fg_mark_kernel(const int64_t* __restrict__ gt_rels, int R, const int* __restrict__ first,
                                                      int N, int* __restrict__ cnt, int* __restrict__ rank) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const int f = first[gt_rels[4 * (long)r]];
    const long s = f + gt_rels[4 * (long)r + 1], o = f + gt_rels[4 * (long)r + 2];
    atomicAdd(&cnt[s * N + o], 1);
    int rk = 0;
    for (int q = 0; q < r; ++q) {
        const int fq = first[gt_rels[4 * (long)q]];
        if (fq + gt_rels[4 * (long)q + 1] == s && fq + gt_rels[4 * (long)q + 2] == o) ++rk;
    }
    rank[r] = rk;
}

template <bool WRITE>
__global__ __launch_bounds__(64) void pair_train_kernel(const int64_t* __restrict__ im, int N, int* __restrict__ cnt,
                                                        int* __restrict__ rowoff, int64_t* __restrict__ out, int cap) {
    const int i = blockIdx.x, lane = threadIdx.x;
    int base = WRITE ? rowoff[i] : 0;
    for (int j0 = 0; j0 < N; j0 += 64) {
        const int j = j0 + lane;
        const bool ok = j < N && same_img_pair(im, i, j);
        const int c = ok ? cnt[(long)i * N + j] : 0;
        const int rows = ok ? max(c, 1) : 0;
        const int inc = wave_incl_scan(rows, lane);
        if (WRITE && ok) {
            const int pos = base + inc - rows;
            if (c == 0) {
                if (pos < cap) {
                    out[4 * (long)pos] = im[i];
                    out[4 * (long)pos + 1] = i;
                    out[4 * (long)pos + 2] = j;
                    out[4 * (long)pos + 3] = 0;
                }
            } else {
                cnt[(long)i * N + j] = pos;  // position of the pair's first FG row, for fg_fill_kernel
            }
        }
        base += __shfl(inc, 63, 64);
    }
    if (!WRITE && lane == 0) rowoff[i] = base;
}

__global__ __launch_bounds__(256) void fg_fill_kernel(const int64_t* __restrict__ gt_rels, int R, const int* __restrict__ first,
                                                      int N, const int* __restrict__ posmap, const int* __restrict__ rank,
                                                      int64_t* __restrict__ out, int cap) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const int f = first[gt_rels[4 * (long)r]];
    const long s = f + gt_rels[4 * (long)r + 1], o = f + gt_rels[4 * (long)r + 2];
    const int pos = posmap[s * N + o] + rank[r];
    if (pos < cap) {
        out[4 * (long)pos] = gt_rels[4 * (long)r];
        out[4 * (long)pos + 1] = s;
        out[4 * (long)pos + 2] = o;
        out[4 * (long)pos + 3] = gt_rels[4 * (long)r + 3];
    }
}

// ---- CSR lists: one wave per (node, side); scans the edge list in order (ballot compaction keeps edge order).
// When the node->image map is given, rel_inds is sorted by image (both pair-index kernels emit it so) and only the
// node's own image segment [lower_bound, upper_bound) is scanned.
__device__ __forceinline__ int bound_img(const int64_t* rel, int E, int64_t img, bool upper) {
    int lo = 0, hi = E;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const int64_t v = rel[3 * (long)mid];
        if (upper ? v <= img : v < img) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

template <bool WRITE>
__global__ __launch_bounds__(64) void csr_kernel(const int64_t* __restrict__ rel, int E, int N, const int64_t* __restrict__ im,
                                                 int* __restrict__ optr, int* __restrict__ iptr, int* __restrict__ oids,
                                                 int* __restrict__ iids, int* __restrict__ so, int* __restrict__ flags) {
    const int n = blockIdx.x >> 1, side = blockIdx.x & 1, lane = threadIdx.x;
    int* ptr = side ? iptr : optr;
    int* ids = side ? iids : oids;
    int base = WRITE ? ptr[n] : 0;
    int e_lo = 0, e_hi = E;
    if (im) {
        e_lo = bound_img(rel, E, im[n], false);
        e_hi = bound_img(rel, E, im[n], true);
    }
    for (int e0 = e_lo; e0 < e_hi; e0 += 64) {
        const int e = e0 + lane;
        const bool ok = e < e_hi && rel[3 * (long)e + 1 + side] == n;
        const unsigned long long m = __ballot(ok);
        if (WRITE && ok) {
            const int pos = base + lanes_below(m, lane);
            ids[pos] = e;
            if (side == 0) {
                if (so) {   // every edge is an out-edge of exactly one node: compact int32 (subject, object)
                    so[2 * (long)e] = n;
                    so[2 * (long)e + 1] = (int)rel[3 * (long)e + 2];
                }
                if (flags && pos != e) flags[0] = 0;   // out-lists are NOT the identity (edges not sorted by subject)
            }
        }
        base += __popcll(m);
    }
    if (!WRITE && lane == 0) ptr[n] = base;
}

__global__ void set_flag_kernel(int* flags) { flags[0] = 1; }

// lib/rel_assignments.py:60-76, the tables the sampler reads: one thread per (detection, column); columns [0,G) are the
// GT boxes (IoU + class/IoU match), columns [G,G+N) the other detections (relation candidates).  fp32 with torchvision
// box_iou's operation order (the sampling probabilities are products of these IoUs, so they have to be the same bits).
__device__ __forceinline__ float iou_f32(const float* a, const float* b) {
    const float area_a = __fmul_rn(__fsub_rn(a[2], a[0]), __fsub_rn(a[3], a[1]));
    const float area_b = __fmul_rn(__fsub_rn(b[2], b[0]), __fsub_rn(b[3], b[1]));
    const float w = fmaxf(__fsub_rn(fminf(a[2], b[2]), fmaxf(a[0], b[0])), 0.f);
    const float h = fmaxf(__fsub_rn(fminf(a[3], b[3]), fmaxf(a[1], b[1])), 0.f);
    const float inter = __fmul_rn(w, h);
    return __fdiv_rn(inter, __fsub_rn(__fadd_rn(area_a, area_b), inter));
}

__global__ __launch_bounds__(256) void rel_tables_kernel(const float* __restrict__ det, const int64_t* __restrict__ det_img,
                                                         const int64_t* __restrict__ det_lab, const float* __restrict__ gt,
                                                         const int64_t* __restrict__ gt_cls, int N, int G, float thr, int nonov,
                                                         float* __restrict__ gt_iou, unsigned char* __restrict__ match,
                                                         unsigned char* __restrict__ poss) {
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    const int cols = G + N;
    if (id >= (long)N * cols) return;
    const int i = (int)(id / cols), c = (int)(id - (long)i * cols);
    const int64_t img = det_img[i];
    if (c < G) {
        const bool same = gt_cls[2 * c] == img;
        const float v = same ? iou_f32(det + 4 * i, gt + 4 * c) : -1.f;
        gt_iou[(long)i * G + c] = v;
        match[(long)i * G + c] = same && det_lab[i] == gt_cls[2 * c + 1] && v >= thr;          // :61
    } else {
        const int j = c - G;
        bool ok = det_img[j] == img && det_lab[i] != 0 && det_lab[j] != 0;                      // :75-76
        if (ok) {
            if (nonov) {
                const float v = iou_f32(det + 4 * i, det + 4 * j);
                ok = v < 1.f && v > 0.f;                                                        // :66
            } else {
                ok = i != j;                                                                    // :69-71
            }
        }
        poss[(long)i * N + j] = ok;
    }
}

}  // namespace

extern "C" int sgg_pair_index_eval(const int64_t* im_inds, const float* boxes, int N, int require_overlap,
                                   int64_t* rel_inds, int cap, int* count, int* work, void* stream) {
    if (!im_inds || !rel_inds || !count || !work || N <= 0 || cap < 0 || (require_overlap && !boxes)) return SGG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(pair_eval_kernel<false>, dim3(N), dim3(64), 0, s, im_inds, boxes, N, require_overlap, work, rel_inds, cap);
    hipLaunchKernelGGL(excl_scan_kernel, dim3(1), dim3(1024), 0, s, work, N, count);
    hipLaunchKernelGGL(pair_eval_kernel<true>, dim3(N), dim3(64), 0, s, im_inds, boxes, N, require_overlap, work, rel_inds, cap);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_pair_index_train(const int64_t* im_inds, int N, const int64_t* gt_rels, int R, const int* img_first,
                                    int64_t* rel_labels, int cap, int* count, int* work, void* stream) {
    if (!im_inds || !rel_labels || !count || !work || N <= 0 || cap < 0 || R < 0 || (R > 0 && (!gt_rels || !img_first)))
        return SGG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    int* rowoff = work;                 // N+1
    int* cnt = work + N + 2;            // N*N
    int* rank = cnt + (long)N * N;      // R
    if (sgg_fill_u32(cnt, 0u, (size_t)N * N, s) != SGG_OK) return SGG_ERR_LAUNCH;
    if (R > 0) hipLaunchKernelGGL(fg_mark_kernel, dim3((R + 255) / 256), dim3(256), 0, s, gt_rels, R, img_first, N, cnt, rank);
    hipLaunchKernelGGL(pair_train_kernel<false>, dim3(N), dim3(64), 0, s, im_inds, N, cnt, rowoff, rel_labels, cap);
    hipLaunchKernelGGL(excl_scan_kernel, dim3(1), dim3(1024), 0, s, rowoff, N, count);
    hipLaunchKernelGGL(pair_train_kernel<true>, dim3(N), dim3(64), 0, s, im_inds, N, cnt, rowoff, rel_labels, cap);
    if (R > 0) hipLaunchKernelGGL(fg_fill_kernel, dim3((R + 255) / 256), dim3(256), 0, s, gt_rels, R, img_first, N, cnt, rank, rel_labels, cap);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_edge_csr(const int64_t* rel_inds, int E, int N, const int64_t* im_inds, int* out_ptr, int* out_ids,
                            int* in_ptr, int* in_ids, int* so, int* flags, void* stream) {
    if (!rel_inds || !out_ptr || !out_ids || !in_ptr || !in_ids || N <= 0 || E < 0) return SGG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (flags) hipLaunchKernelGGL(set_flag_kernel, dim3(1), dim3(1), 0, s, flags);
    hipLaunchKernelGGL(csr_kernel<false>, dim3(2 * N), dim3(64), 0, s, rel_inds, E, N, im_inds, out_ptr, in_ptr, out_ids, in_ids, so, flags);
    hipLaunchKernelGGL(excl_scan_kernel, dim3(1), dim3(1024), 0, s, out_ptr, N, (int*)nullptr);
    hipLaunchKernelGGL(excl_scan_kernel, dim3(1), dim3(1024), 0, s, in_ptr, N, (int*)nullptr);
    hipLaunchKernelGGL(csr_kernel<true>, dim3(2 * N), dim3(64), 0, s, rel_inds, E, N, im_inds, out_ptr, in_ptr, out_ids, in_ids, so, flags);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_rel_assign_tables(const float* det_boxes, const int64_t* det_img, const int64_t* det_labels, int N,
                                     const float* gt_boxes, const int64_t* gt_classes, int G, float fg_thresh,
                                     int filter_non_overlap, float* gt_iou, uint8_t* match, uint8_t* poss, void* stream) {
    if (N == 0) return SGG_OK;
    if (!det_boxes || !det_img || !det_labels || !poss || N < 0 || G < 0 || (G > 0 && (!gt_boxes || !gt_classes || !gt_iou || !match)))
        return SGG_ERR_ARG;
    const long total = (long)N * (G + N);
    hipLaunchKernelGGL(rel_tables_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, det_boxes, det_img,
                       det_labels, gt_boxes, gt_classes, N, G, fg_thresh, filter_non_overlap, gt_iou, match, poss);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}
