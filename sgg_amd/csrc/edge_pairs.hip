// Unordered box pairs of the relation graphs.  The union box of (subject, object) and of (object, subject) is the same box, so its
// RoIAlign row and the K = 25088 part of fc6 on it are the same for both edges (sgg_models/rel_model_base.py:245-260 pools every edge;
// rel_model_stanford.py:104 runs fc6 on every edge): they are computed once per unordered pair {i < j} of an image -- slot
// u = ubase[img] + i (2n - i - 1) / 2 + (j - i - 1) with image-local i, j and n boxes in the image -- and the per-edge parts (the
// rect term, bias, activation; in the backward the gradient rows of a pair's edges) meet them through these tables.
#include "common.h"
#include "../../include/sgg_hip.h"

namespace {
__global__ __launch_bounds__(256) void pair_slots_kernel(const int64_t* __restrict__ rel, const int* __restrict__ first,
                                                         const int* __restrict__ ubase, const int* __restrict__ cnt, int E, int B,
                                                         int* __restrict__ e2u, int* __restrict__ u2e, int* __restrict__ ucount,
                                                         int* __restrict__ flag) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const int img = (int)rel[3L * e], s = (int)rel[3L * e + 1], o = (int)rel[3L * e + 2];
    if (img < 0 || img >= B) {
        atomicOr(flag, 1);
        e2u[e] = 0;
        return;
    }
    const int n = cnt[img], i = min(s, o) - first[img], j = max(s, o) - first[img];
    if (i < 0 || j >= n || i == j) {
        atomicOr(flag, 1);
        e2u[e] = 0;
        return;
    }
    const int u = ubase[img] + (i * (2 * n - i - 1)) / 2 + (j - i - 1);
    e2u[e] = u;
    const int k = atomicAdd(&ucount[u], 1);
    if (k < 2) u2e[2 * u + k] = e;
    else atomicOr(flag, 2);                 // more than two edges on one unordered pair (duplicate relations): the caller falls back
}

// out[c][u] = x[a][c] + x[b][c] (same element type as x) with (a, b) = the pair's edges in ascending order (-1: none); columns u >= U of the padded output are 0
template <typename T>
__global__ __launch_bounds__(256) void transpose_pairsum_kernel(const T* __restrict__ x, long ldx, const int* __restrict__ u2e,
                                                                T* __restrict__ out, long ld_out, int U, int Up, int C) {
    __shared__ float t[64][65];
    const int u0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int u = u0 + r, c = c0 + tx;
        float v = 0.f;
        if (u < U && c < C) {
            const int a = u2e[2 * u], b = u2e[2 * u + 1];
            const int lo = (a >= 0 && b >= 0) ? min(a, b) : max(a, b), hi = (a >= 0 && b >= 0) ? max(a, b) : -1;
            if (lo >= 0) v = Elem<T>::ld(x + (long)lo * ldx + c);
            if (hi >= 0) v += Elem<T>::ld(x + (long)hi * ldx + c);
        }
        t[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int c = c0 + r, u = u0 + tx;
        if (c < C && u < Up) Elem<T>::st(out + (long)c * ld_out + u, t[tx][r]);
    }
}

// out[u][c] = x[a][c] + x[b][c]: the rows form of the sum above (the TN weight-gradient kernel reads it as it lies); same arithmetic
template <typename T>
__global__ __launch_bounds__(256) void pairsum_kernel(const T* __restrict__ x, long ldx, const int* __restrict__ u2e, T* __restrict__ out,
                                                      long ld_out, int U, int C8) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)U * C8) return;
    const int u = (int)(i / C8), c = (int)(i - (long)u * C8) * 8;
    const int a = u2e[2 * u], b = u2e[2 * u + 1];
    const int lo = (a >= 0 && b >= 0) ? min(a, b) : max(a, b), hi = (a >= 0 && b >= 0) ? max(a, b) : -1;
    float v[8], w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = 0.f;
    if (lo >= 0) load8(x + (long)lo * ldx + c, v);
    if (hi >= 0) {
        load8(x + (long)hi * ldx + c, w);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += w[k];
    }
    store8(out + (long)u * ld_out + c, v);
}

// y[m][j] += r[m][(j + col0) / group],  j < ncol
template <typename T>
__global__ __launch_bounds__(256) void group_bcast_add_kernel(T* __restrict__ y, long ldy, const float* __restrict__ r, long ldr, int M,
                                                              int ncol, int group, int col0) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long per_row = (ncol + 7) / 8;
    const int m = (int)(i / per_row), j0 = (int)(i - (long)m * per_row) * 8;
    if (m >= M) return;
    T* p = y + (long)m * ldy + j0;
    const float* rr = r + (long)m * ldr;
    if (j0 + 8 <= ncol && ((reinterpret_cast<uintptr_t>(p) & (8 * sizeof(T) - 1)) == 0)) {
        float v[8];
        load8(p, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += rr[(j0 + k + col0) / group];
        store8(p, v);
    } else {
        for (int k = 0; k < 8 && j0 + k < ncol; ++k) Elem<T>::st(p + k, Elem<T>::ld(p + k) + rr[(j0 + k + col0) / group]);
    }
}
}  // namespace

extern "C" int sgg_pair_slots(const int64_t* rel_inds, const int* first, const int* ubase, const int* cnt, int E, int B, int U, int* e2u,
                              int* u2e, int* ucount, int* flag, void* stream) {
    if (E == 0) return SGG_OK;
    if (!rel_inds || !first || !ubase || !cnt || !e2u || !u2e || !ucount || !flag || E < 0 || B <= 0 || U <= 0) return SGG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (sgg_fill_u32(u2e, 0xffffffffu, 2 * (size_t)U, s) != SGG_OK || sgg_fill_u32(ucount, 0u, (size_t)U, s) != SGG_OK ||
        sgg_fill_u32(flag, 0u, 1, s) != SGG_OK)
        return SGG_ERR_LAUNCH;
    hipLaunchKernelGGL(pair_slots_kernel, dim3((E + 255) / 256), dim3(256), 0, s, rel_inds, first, ubase, cnt, E, B, e2u, u2e, ucount, flag);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_transpose_pairsum(const void* x, int64_t ldx, const int* u2e, void* out, int64_t ld_out, int U, int C, int dtype,
                                     void* stream) {
    if (U == 0 || C == 0) return SGG_OK;
    if (!x || !u2e || !out || U < 0 || C < 0 || ld_out < U || ldx < C) return SGG_ERR_ARG;
    const int Up = (int)ld_out;
    const dim3 grid((C + 63) / 64, (Up + 63) / 64), blk(256);
    hipStream_t s = (hipStream_t)stream;
    SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL(transpose_pairsum_kernel<T>, grid, blk, 0, s, (const T*)x, (long)ldx, u2e, (T*)out, (long)ld_out, U, Up, C));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_pairsum(const void* x, int64_t ldx, const int* u2e, void* out, int64_t ld_out, int U, int C, int dtype, void* stream) {
    if (U == 0 || C == 0) return SGG_OK;
    if (!x || !u2e || !out || U < 0 || C < 0 || (C & 7) || ld_out < C || ldx < C || ((ldx | ld_out) & 7) || (((uintptr_t)x | (uintptr_t)out) & 15))
        return SGG_ERR_ARG;
    const long total = (long)U * (C / 8);
    hipStream_t s = (hipStream_t)stream;
    SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL(pairsum_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const T*)x, (long)ldx, u2e, (T*)out,
                                            (long)ld_out, U, C / 8));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_group_bcast_add(void* y, int64_t ldy, const float* r, int64_t ldr, int M, int ncol, int group, int col0, int dtype,
                                   void* stream) {
    if (M == 0 || ncol == 0) return SGG_OK;
    if (!y || !r || M < 0 || ncol < 0 || group <= 0 || col0 < 0 || ldy < ncol || ldr * group < (int64_t)ncol + col0) return SGG_ERR_ARG;
    const long per_row = ((long)ncol + 7) / 8;
    const dim3 grid((unsigned)((per_row * M + 255) / 256)), blk(256);
    hipStream_t s = (hipStream_t)stream;
    SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL(group_bcast_add_kernel<T>, grid, blk, 0, s, (T*)y, (long)ldy, r, (long)ldr, M, ncol, group, col0));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}
