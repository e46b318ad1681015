// Eval tail: class / predicate softmax, triple score, descending sort, gather.
// Reference: sgg_models/rel_model_stanford.py:183-207 and filter_dets, lib/surgery.py:17-55.
#include "common.h"

namespace {

// one wave per box: softmax over C classes, best non-background class (rel_model_stanford.py:187-191)
template <typename T>
__global__ __launch_bounds__(256) void obj_tail_kernel(const T* __restrict__ od, int N, int C, const int64_t* __restrict__ gt,
                                                       float* __restrict__ scores, int64_t* __restrict__ preds) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    if (gt) {  // predcls: rel_model_stanford.py:184-185
        if (lane == 0) {
            scores[n] = 1.0f;
            preds[n] = gt[n];
        }
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
    if (lane == 0) {
        scores[n] = best;
        preds[n] = bi;
    }
}

// one wave per edge: predicate softmax + triple score (rel_model_stanford.py:204, lib/surgery.py:40-46)
template <typename T>
__global__ __launch_bounds__(256) void rel_tail_kernel(const T* __restrict__ rd, int E, int P, const int64_t* __restrict__ rel,
                                                       const float* __restrict__ obj_scores, float* __restrict__ probs,
                                                       float* __restrict__ keys, int* __restrict__ idx, int n2) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= n2) return;
    if (e >= E) {  // padding of the bitonic network: sorts last
        if (lane == 0) {
            keys[e] = -INFINITY;
            idx[e] = 0x7fffffff;
        }
        return;
    }
    const T* row = rd + (long)e * P;
    float mx = -INFINITY;
    for (int p = lane; p < P; p += 64) mx = fmaxf(mx, Elem<T>::ld(row + p));
    mx = wave_max(mx);
    float sum = 0.f;
    for (int p = lane; p < P; p += 64) sum += expf(Elem<T>::ld(row + p) - mx);
    sum = wave_sum(sum);
    float pm = -INFINITY;
    for (int p = lane; p < P; p += 64) {
        const float pr = expf(Elem<T>::ld(row + p) - mx) / sum;
        probs[(long)e * P + p] = pr;
        if (p >= 1) pm = fmaxf(pm, pr);
    }
    pm = wave_max(pm);
    if (lane == 0) {
        const float s0 = obj_scores[rel[3 * (long)e + 1]], s1 = obj_scores[rel[3 * (long)e + 2]];
        const float sc = pm * s0 * s1;
        keys[e] = (sc == sc) ? sc : -INFINITY;
        idx[e] = e;
    }
}

// strict total order "a sorts before b": descending score, ties by ascending edge index (indices are unique;
// NaN scores were mapped to -inf when the keys were made)
__device__ __forceinline__ bool before(float ka, int ia, float kb, int ib) {
    return ka > kb || (ka == kb && ia < ib);
}

// one compare-exchange step of a bitonic network in global memory (E > 16384)
__global__ __launch_bounds__(256) void bitonic_step_kernel(float* __restrict__ k, int* __restrict__ id, int n2, int size,
                                                           int stride) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= (n2 >> 1)) return;
    const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
    const bool asc = (lo & size) == 0;
    const float ka = k[lo], kb = k[hi];
    const int ia = id[lo], ib = id[hi];
    const bool in_order = !before(kb, ib, ka, ia);
    if (in_order != asc) {
        k[lo] = kb; k[hi] = ka;
        id[lo] = ib; id[hi] = ia;
    }
}

// Rank sort for moderate E (<= 16384): rank[i] = #{j : j sorts before i}.  O(E^2) compares spread over
// (E/256) x SPLIT blocks of 256 threads, keys streamed through LDS in 1024-key chunks (broadcast reads).
// Deterministic (strict total order) and ~20x faster than a single-workgroup bitonic network at E = 7936.
constexpr int RANK_CHUNK = 1024;
__global__ __launch_bounds__(256) void rank_kernel(const float* __restrict__ keys, int E, int* __restrict__ rank, int split) {
    __shared__ float sk[RANK_CHUNK];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float ki = i < E ? keys[i] : 0.f;
    const int per = (E + split - 1) / split;
    const int j0 = blockIdx.y * per, j1 = min(E, j0 + per);
    int cnt = 0;
    for (int base = j0; base < j1; base += RANK_CHUNK) {
        const int n = min(RANK_CHUNK, j1 - base);
        __syncthreads();
        for (int t = threadIdx.x; t < n; t += 256) sk[t] = keys[base + t];
        __syncthreads();
        for (int t = 0; t < n; ++t) cnt += before(sk[t], base + t, ki, i) ? 1 : 0;
    }
    if (i < E && cnt) atomicAdd(&rank[i], cnt);
}

// scatter form of the gather: source edge e goes to output row rank[e]
__global__ __launch_bounds__(256) void tail_scatter_kernel(const int* __restrict__ rank, const int64_t* __restrict__ rel,
                                                           const float* __restrict__ probs, int E, int P,
                                                           int64_t* __restrict__ rels, float* __restrict__ out) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= E) return;
    const int r = rank[e];
    if (lane < 2) rels[2 * (long)r + lane] = rel[3 * (long)e + 1 + lane];
    for (int p = lane; p < P; p += 64) out[(long)r * P + p] = probs[(long)e * P + p];
}

__global__ __launch_bounds__(256) void tail_gather_kernel(const int* __restrict__ idx, const int64_t* __restrict__ rel,
                                                          const float* __restrict__ probs, int E, int P,
                                                          int64_t* __restrict__ rels, float* __restrict__ out) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= E) return;
    const int e = idx[r];
    if (lane < 2) rels[2 * (long)r + lane] = rel[3 * (long)e + 1 + lane];
    for (int p = lane; p < P; p += 64) out[(long)r * P + p] = probs[(long)e * P + p];
}

int pow2ceil(int x) {
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

}  // namespace

extern "C" int sgg_eval_tail(const void* obj_dists, int N, int C, const void* rel_dists, int E, int P,
                             const int64_t* rel_inds, const int64_t* gt_classes, float* obj_scores, int64_t* obj_preds,
                             int64_t* rels, float* pred_scores, void* work, int dtype, void* stream) {
    if (!sgg_is_dtype(dtype)) return SGG_ERR_DTYPE;
    if (!obj_dists || !obj_scores || !obj_preds || N <= 0 || C < 2 || E < 0 || P < 2) return SGG_ERR_ARG;
    if (E > 0 && (!rel_dists || !rel_inds || !rels || !pred_scores || !work)) return SGG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL(obj_tail_kernel<T>, dim3((N + 3) / 4), dim3(256), 0, s, (const T*)obj_dists, N, C, gt_classes, obj_scores, obj_preds));
    if (E > 0) {
        const int n2 = pow2ceil(E);
        float* keys = (float*)work;
        int* idx = (int*)work + n2;
        float* probs = (float*)work + 2 * (size_t)n2;
        SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL(rel_tail_kernel<T>, dim3((n2 + 3) / 4), dim3(256), 0, s, (const T*)rel_dists, E, P, rel_inds, obj_scores, probs, keys, idx, n2));
        if (E <= 16384) {
            if (sgg_fill_u32(idx, 0u, (size_t)E, s) != SGG_OK) return SGG_ERR_LAUNCH;      // (not hipMemsetAsync: common.h)
            const int split = E >= 2048 ? 8 : 1;
            hipLaunchKernelGGL(rank_kernel, dim3((E + 255) / 256, split), dim3(256), 0, s, keys, E, idx, split);
            hipLaunchKernelGGL(tail_scatter_kernel, dim3((E + 3) / 4), dim3(256), 0, s, idx, rel_inds, probs, E, P, rels, pred_scores);
        } else {
            for (int size = 2; size <= n2; size <<= 1)
                for (int stride = size >> 1; stride > 0; stride >>= 1)
                    hipLaunchKernelGGL(bitonic_step_kernel, dim3((n2 / 2 + 255) / 256), dim3(256), 0, s, keys, idx, n2, size, stride);
            hipLaunchKernelGGL(tail_gather_kernel, dim3((E + 3) / 4), dim3(256), 0, s, idx, rel_inds, probs, E, P, rels, pred_scores);
        }
    }
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}
