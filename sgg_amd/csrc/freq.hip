// FrequencyBias (the -use_bias / -test_bias flags): log P(predicate | class_subj, class_obj) looked up per candidate edge and
// added to (or put in place of) rel_dists.  Reference: lib/sparse_targets.py:26-31 (index_with_labels) called from
// sgg_models/rel_model_stanford.py:159-177.  HBM-bound index work: E*P*4 B read + written, the [C*C,P] table stays in L2.
#include "common.h"

namespace {

// one wave per box: most probable class among 1..C-1 of softmax(obj_dists) (rel_model_stanford.py:161-164; ties -> the lowest
// class, torch.sort leaves them unspecified), or the given label (predcls, :166-167)
template <typename T>
__global__ __launch_bounds__(256) void freq_obj_pred_kernel(const T* __restrict__ od, int N, int C, const int64_t* __restrict__ gt,
                                                            int64_t* __restrict__ preds) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    if (gt) {
        if (lane == 0) preds[n] = gt[n];
        return;
    }
    const T* row = od + (long)n * C;
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, Elem<T>::ld(row + c));
    mx = wave_max(mx);
    float sum = 0.f;
    for (int c = lane; c < C; c += 64) sum += expf(Elem<T>::ld(row + c) - mx);
    sum = wave_sum(sum);
    float best = -1.f;
    int bi = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
        if (c == 0) continue;
        const float p = expf(Elem<T>::ld(row + c) - mx) / sum;
        if (p > best) {
            best = p;
            bi = c;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) {
            best = ob;
            bi = oi;
        }
    }
    if (lane == 0) preds[n] = bi;
}

// thread per (edge, predicate): out = [rel_in +] table[pred[s]*C + pred[o]]; the table row index is kept for the backward
__global__ __launch_bounds__(256) void freq_gather_add_kernel(const int64_t* __restrict__ preds, const int64_t* __restrict__ rel,
                                                              int E, int C, int P, const float* __restrict__ table,
                                                              const float* __restrict__ rel_in, float* __restrict__ out,
                                                              int32_t* __restrict__ row_idx) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)E * P) return;
    const int e = (int)(i / P), p = (int)(i - (long)e * P);
    const int row = (int)(preds[rel[(long)e * 3 + 1]] * C + preds[rel[(long)e * 3 + 2]]);
    const float t = table[(long)row * P + p];
    out[i] = rel_in ? rel_in[i] + t : t;
    if (p == 0 && row_idx) row_idx[e] = row;
}

// d_table[row_idx[e], :] += d_out[e, :]  (nn.Embedding's dense weight gradient) WITHOUT atomics: a wave owns edge e; it scans the
// edges before e for the same table row (someone earlier owns the row: nothing to do), otherwise it is the row's first edge and adds
// d_out of every edge with that row in ascending edge order.  O(E^2) 4-byte compares from L2 (E = 8k: 63 M, ~20 us), one writer per
// table row, a fixed summation order: bit-reproducible (the float atomicAdd form it replaces was not; main.py -use_bias training).
__global__ __launch_bounds__(256) void freq_scatter_kernel(const float* __restrict__ d_out, const int32_t* __restrict__ row_idx, int E,
                                                           int P, float* __restrict__ d_table) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= E) return;
    const int row = row_idx[e];
    bool earlier = false;
    for (int j = lane; j < e; j += 64) earlier |= row_idx[j] == row;
    if (__ballot(earlier)) return;                     // (wave-uniform: the whole wave leaves)
    for (int p0 = 0; p0 < P; p0 += 64) {
        const int p = p0 + lane;
        float acc = p < P ? d_table[(long)row * P + p] : 0.f;
        for (int j0 = e; j0 < E; j0 += 64) {           // edges j >= e with the same row, 64 at a time, consumed in ascending order
            const int j = j0 + lane;
            unsigned long long m = __ballot(j < E && row_idx[j] == row);
            while (m) {
                const int k = __builtin_ctzll(m);
                m &= m - 1;
                if (p < P) acc += d_out[(long)(j0 + k) * P + p];
            }
        }
        if (p < P) d_table[(long)row * P + p] = acc;
    }
}

}  // namespace

extern "C" int sgg_freq_bias_fwd(const void* obj_dists, int N, int C, const int64_t* gt_classes, const int64_t* rel_inds, int E,
                                 const float* table, int P, const float* rel_in, float* rel_out, int64_t* obj_preds,
                                 int32_t* row_idx, int dtype, void* stream) {
    if (N < 0 || E < 0 || C < 2 || P < 1 || (long)C * C > 0x7fffffffL) return SGG_ERR_ARG;
    if (!obj_preds || !table || (!obj_dists && !gt_classes) || (E > 0 && (!rel_inds || !rel_out))) return SGG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (N > 0) {
        if (gt_classes) {
            freq_obj_pred_kernel<float><<<(N + 3) / 4, 256, 0, s>>>((const float*)obj_dists, N, C, gt_classes, obj_preds);
        } else {
            SGG_FOR_DTYPE(dtype, (freq_obj_pred_kernel<T><<<(N + 3) / 4, 256, 0, s>>>((const T*)obj_dists, N, C, gt_classes, obj_preds)));
        }
        SGG_CHECK_LAUNCH();
    }
    if (E > 0) {
        const long total = (long)E * P;
        freq_gather_add_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(obj_preds, rel_inds, E, C, P, table, rel_in, rel_out,
                                                                               row_idx);
        SGG_CHECK_LAUNCH();
    }
    return SGG_OK;
}

extern "C" int sgg_freq_bias_bwd(const float* d_out, const int32_t* row_idx, int E, int P, float* d_table, void* stream) {
    if (E < 0 || P < 1) return SGG_ERR_ARG;
    if (E == 0) return SGG_OK;
    if (!d_out || !row_idx || !d_table) return SGG_ERR_ARG;
    freq_scatter_kernel<<<(unsigned)((E + 3) / 4), 256, 0, (hipStream_t)stream>>>(d_out, row_idx, E, P, d_table);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}
