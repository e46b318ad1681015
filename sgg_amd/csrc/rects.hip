// Union-box mask raster (the reference's only native code) and the pieces of the "rect conv" that are not GEMMs.
#include "common.h"

namespace {

// minmax(x) of lib/draw_rectangles/draw_rectangles.pyx:24-25 with the same ternary semantics
// (max(x,0) = 0 > x ? 0 : x ; min(t,1) = 1 < t ? 1 : t), so NaN handling matches the C the .pyx compiles to.
__device__ __forceinline__ float minmax01(float x) {
    const float t = (0.f > x) ? 0.f : x;
    return (1.f < t) ? 1.f : t;
}

struct PairGeom {
    float x1[2], y1[2], x2[2], y2[2];  // the two boxes mapped to [0,P] inside the union box
};

// draw_rectangles.pyx:46-59, fp32, same operation order, no FMA contraction (bit-exact with the reference).
__device__ __forceinline__ PairGeom pair_geom(const float* __restrict__ rois, const int64_t* __restrict__ pairs, long e,
                                              int P) {
    const float* a = rois + pairs[2 * e] * 5 + 1;
    const float* b = rois + pairs[2 * e + 1] * 5 + 1;
    const float bx[2][4] = {{a[0], a[1], a[2], a[3]}, {b[0], b[1], b[2], b[3]}};
    const float x1u = fminf(bx[0][0], bx[1][0]), y1u = fminf(bx[0][1], bx[1][1]);
    const float x2u = fmaxf(bx[0][2], bx[1][2]), y2u = fmaxf(bx[0][3], bx[1][3]);
    const float w = __fsub_rn(x2u, x1u), h = __fsub_rn(y2u, y1u);
    const float Pf = (float)P;
    PairGeom g;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        g.x1[i] = __fdiv_rn(__fmul_rn(__fsub_rn(bx[i][0], x1u), Pf), w);
        g.y1[i] = __fdiv_rn(__fmul_rn(__fsub_rn(bx[i][1], y1u), Pf), h);
        g.x2[i] = __fdiv_rn(__fmul_rn(__fsub_rn(bx[i][2], x1u), Pf), w);
        g.y2[i] = __fdiv_rn(__fmul_rn(__fsub_rn(bx[i][3], y1u), Pf), h);
    }
    return g;
}

// coverage of pixel (j,k) by box i: draw_rectangles.pyx:62-66
__device__ __forceinline__ float coverage(const PairGeom& g, int i, int j, int k) {
    const float yc = __fmul_rn(minmax01(__fsub_rn((float)(j + 1), g.y1[i])), minmax01(__fsub_rn(g.y2[i], (float)j)));
    const float xc = __fmul_rn(minmax01(__fsub_rn((float)(k + 1), g.x1[i])), minmax01(__fsub_rn(g.x2[i], (float)k)));
    return __fmul_rn(xc, yc);
}

// a-5: out f32[E,2,P,P] (+offset).  One workgroup per pair; consecutive threads write consecutive pixels.
__global__ __launch_bounds__(256) void union_rects_kernel(const float* __restrict__ rois, const int64_t* __restrict__ pairs,
                                                          int P, float offset, float* __restrict__ out) {
    const long e = blockIdx.x;
    const PairGeom g = pair_geom(rois, pairs, e, P);
    const int n = 2 * P * P;
    float* o = out + e * n;
    for (int t = threadIdx.x; t < n; t += 256) {
        const int i = t / (P * P), rem = t - i * P * P;
        const int j = rem / P, k = rem - j * P;
        o[t] = __fadd_rn(coverage(g, i, j, k), offset);
    }
}

// The conv stack of lib/get_union_boxes.py:51-59 uses stride 16 for BOTH convs (typo at :40-43), so the first conv
// (k7,p3) reads only 2x2 windows at rows/cols {-3..3} and {13..19}.  Emit exactly those 4 patches per pair as
// GEMM rows: out[(e*4 + oy*2+ox)][k = c*49 + ky*7 + kx], zero outside the raster, raster-0.5 inside.
template <typename T>
__global__ __launch_bounds__(256) void rect_patches_kernel(const float* __restrict__ rois, const int64_t* __restrict__ pairs,
                                                           int P, T* __restrict__ out, int Kpad) {
    const long e = blockIdx.x;
    const PairGeom g = pair_geom(rois, pairs, e, P);
    T* o = out + e * 4 * Kpad;
    for (int t = threadIdx.x; t < 4 * Kpad; t += 256) {
        const int pos = t / Kpad, k = t - pos * Kpad;
        float v = 0.f;
        if (k < 98) {
            const int c = k / 49, r = k - c * 49;
            const int ky = r / 7, kx = r - ky * 7;
            const int j = (pos >> 1) * 16 - 3 + ky, kk = (pos & 1) * 16 - 3 + kx;
            if (j >= 0 && j < P && kk >= 0 && kk < P) v = __fadd_rn(coverage(g, c, j, kk), -0.5f);
        }
        Elem<T>::st(o + t, v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void max4_kernel(const T* __restrict__ in, T* __restrict__ out, long total, int C) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;  // index over E*C/8
    if (i >= total) return;
    const int c8 = C >> 3;
    const long e = i / c8;
    const int c = (int)(i - e * c8) * 8;
    const T* p = in + e * 4 * C + c;
    float a[8], t[8];
    load8(p, a);
#pragma unroll
    for (int q = 1; q < 4; ++q) {
        load8(p + (long)q * C, t);
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = fmaxf(a[k], t[k]);
    }
    store8(out + e * C + c, a);
}

// x[r][c][p] += add[r][c]  (8 consecutive elements of a row per thread)
template <typename T>
__global__ __launch_bounds__(256) void bcast_add_kernel(T* __restrict__ x, const float* __restrict__ add, long total, int PP,
                                                        int C) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;  // index over R*C*PP/8
    if (i >= total) return;
    const long row8 = (long)C * PP / 8;
    const long r = i / row8;
    const int L0 = (int)(i - r * row8) * 8;
    float a[8];
    load8(x + i * 8, a);
    int c = L0 / PP, p = L0 - c * PP;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        a[k] += add[r * C + c];
        if (++p == PP) {
            p = 0;
            ++c;
        }
    }
    store8(x + i * 8, a);
}

}  // namespace

extern "C" int sgg_union_rects_fwd(const float* rois, const int64_t* pairs, int E, int P, float offset, float* out,
                                   void* stream) {
    if (E == 0) return SGG_OK;
    if (!rois || !pairs || !out || E < 0 || P <= 0) return SGG_ERR_ARG;
    hipLaunchKernelGGL(union_rects_kernel, dim3(E), dim3(256), 0, (hipStream_t)stream, rois, pairs, P, offset, out);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_union_rect_patches(const float* rois, const int64_t* pairs, int E, int P, void* out, int Kpad,
                                      int dtype, void* stream) {
    if (E == 0) return SGG_OK;
    // geometry of the typo'd stack: k7/p3/s16 must give a 2x2 map
    if (!rois || !pairs || !out || E < 0 || Kpad < 98 || (P + 6 - 7) / 16 + 1 != 2) return SGG_ERR_ARG;
    if (dtype == SGG_BF16)
        hipLaunchKernelGGL(rect_patches_kernel<bf16_t>, dim3(E), dim3(256), 0, (hipStream_t)stream, rois, pairs, P, (bf16_t*)out, Kpad);
    else if (dtype == SGG_F32)
        hipLaunchKernelGGL(rect_patches_kernel<float>, dim3(E), dim3(256), 0, (hipStream_t)stream, rois, pairs, P, (float*)out, Kpad);
    else
        return SGG_ERR_DTYPE;
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_max4_rows(const void* in, void* out, int E, int C, int dtype, void* stream) {
    if (E == 0) return SGG_OK;
    if (!in || !out || E < 0 || C <= 0 || (C & 7)) return SGG_ERR_ARG;
    const long total = (long)E * (C / 8);
    const int grid = (int)((total + 255) / 256);
    if (dtype == SGG_BF16)
        hipLaunchKernelGGL(max4_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, (bf16_t*)out, total, C);
    else if (dtype == SGG_F32)
        hipLaunchKernelGGL(max4_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)in, (float*)out, total, C);
    else
        return SGG_ERR_DTYPE;
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_bcast_add(void* x, const float* add_rc, int R, int PP, int C, int dtype, void* stream) {
    if (R == 0) return SGG_OK;
    if (!x || !add_rc || R < 0 || PP <= 0 || C <= 0 || (((long)C * PP) & 7)) return SGG_ERR_ARG;
    const long total = (long)R * ((long)C * PP / 8);
    const int grid = (int)((total + 255) / 256);
    if (dtype == SGG_BF16)
        hipLaunchKernelGGL(bcast_add_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, add_rc, total, PP, C);
    else if (dtype == SGG_F32)
        hipLaunchKernelGGL(bcast_add_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (float*)x, add_rc, total, PP, C);
    else
        return SGG_ERR_DTYPE;
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}
