// Scene-graph recall matching: lib/sgg_eval.py:280-417 as one launch over a whole batch of images.
//
// The reference builds, per image, a [G, P] boolean of class-triplet equality (intersect_2d), then for every GT triplet
// with a candidate computes two IoU vectors and appends the GT index to each matching prediction's list;
// R@K = |union(pred_to_gt[:K])| / G.  Only the FIRST matching prediction of each GT triplet matters for every K, so the
// kernel returns first_rank[g] = min{p : pred p matches gt g} and R@K is a count of first_rank < K.
#include "common.h"

namespace {

// torchvision box_iou in fp32 with its operation order (areas, max/min corners, clamp, inter / (a + b - inter)); the
// explicit _rn intrinsics keep the compiler from contracting (a + b) - w*h into an fma.
__device__ __forceinline__ float box_iou_f32(const float* a, const float* b) {
    const float area_a = __fmul_rn(__fsub_rn(a[2], a[0]), __fsub_rn(a[3], a[1]));
    const float area_b = __fmul_rn(__fsub_rn(b[2], b[0]), __fsub_rn(b[3], b[1]));
    const float w = fmaxf(__fsub_rn(fminf(a[2], b[2]), fmaxf(a[0], b[0])), 0.f);
    const float h = fmaxf(__fsub_rn(fminf(a[3], b[3]), fmaxf(a[1], b[1])), 0.f);
    const float inter = __fmul_rn(w, h);
    return __fdiv_rn(inter, __fsub_rn(__fadd_rn(area_a, area_b), inter));
}

// one wave per GT triplet; predictions of its image scanned 64 at a time in rank order
__global__ __launch_bounds__(256) void recall_match_kernel(const int* __restrict__ gt_trip, const float* __restrict__ gt_box,
                                                           const int* __restrict__ gt_img, int G,
                                                           const int* __restrict__ pred_trip, const float* __restrict__ pred_box,
                                                           const int* __restrict__ pred_ptr, const int* __restrict__ gt_pair,
                                                           const int* __restrict__ pred_pair, float thr, int phrdet,
                                                           int* __restrict__ first_rank, int* __restrict__ pair_rank) {
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (g >= G) return;
    const int c0 = gt_trip[3 * g], c1 = gt_trip[3 * g + 1], c2 = gt_trip[3 * g + 2];
    float gb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) gb[k] = gt_box[8 * g + k];
    float gu[4] = {fminf(gb[0], gb[4]), fminf(gb[1], gb[5]), fmaxf(gb[2], gb[6]), fmaxf(gb[3], gb[7])};
    const int img = gt_img[g];
    const int p0 = pred_ptr[img], p1 = pred_ptr[img + 1];
    const int ga = gt_pair ? gt_pair[2 * g] : 0, gbi = gt_pair ? gt_pair[2 * g + 1] : 0;
    int first = 0x7fffffff, prank = -1, pcount = 0;
    bool need_first = true, need_pair = gt_pair != nullptr;
    for (int base = p0; base < p1 && (need_first || need_pair); base += 64) {
        const int p = base + lane;
        bool match = false, same_pair = false;
        if (p < p1) {
            const int* t = pred_trip + 3 * (long)p;
            if (t[0] == c0 && t[1] == c1 && t[2] == c2) {
                const float* pb = pred_box + 8 * (long)p;
                if (phrdet) {
                    const float pu[4] = {fminf(pb[0], pb[4]), fminf(pb[1], pb[5]), fmaxf(pb[2], pb[6]), fmaxf(pb[3], pb[7])};
                    match = box_iou_f32(gu, pu) >= thr;
                } else {
                    match = box_iou_f32(gb, pb) >= thr && box_iou_f32(gb + 4, pb + 4) >= thr;
                }
            }
            if (gt_pair) {
                const int a = pred_pair[2 * (long)p], b = pred_pair[2 * (long)p + 1];
                same_pair = (a == ga && b == gbi) || (a == gbi && b == ga);
            }
        }
        const unsigned long long mm = __ballot(match);
        if (need_first && mm) {
            first = base - p0 + __builtin_ctzll(mm);
            need_first = false;
        }
        if (need_pair) {
            const unsigned long long mp = __ballot(same_pair), mpm = __ballot(match && same_pair);
            if (mpm) {
                const int f = __builtin_ctzll(mpm);
                prank = pcount + __builtin_popcountll(mp & ((1ull << f) - 1ull));
                need_pair = false;
            } else {
                pcount += __builtin_popcountll(mp);
            }
        }
    }
    if (lane == 0) {
        first_rank[g] = first;
        if (pair_rank) pair_rank[g] = prank;
    }
}

}  // namespace

extern "C" int sgg_recall_first_match(const int32_t* gt_trip, const float* gt_box, const int32_t* gt_img, int G,
                                      const int32_t* pred_trip, const float* pred_box, const int32_t* pred_ptr, int B,
                                      const int32_t* gt_pair, const int32_t* pred_pair, float iou_thresh, int phrdet,
                                      int32_t* first_rank, int32_t* pair_rank, void* stream) {
    if (G == 0) return SGG_OK;
    if (!gt_trip || !gt_box || !gt_img || !pred_ptr || !first_rank || G < 0 || B <= 0) return SGG_ERR_ARG;
    if ((gt_pair == nullptr) != (pred_pair == nullptr) || (pair_rank && !gt_pair)) return SGG_ERR_ARG;
    hipLaunchKernelGGL(recall_match_kernel, dim3((G + 3) / 4), dim3(256), 0, (hipStream_t)stream, gt_trip, gt_box, gt_img, G,
                       pred_trip, pred_box, pred_ptr, gt_pair, pred_pair, iou_thresh, phrdet, first_rank, pair_rank);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}
