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

// edge_model 'raw_boxes' (lib/get_union_boxes.py:69-116): channel i is box i drawn in IMAGE coordinates normalised to [0,1]
// (not relative to the union box): F.grid_sample (bilinear, zero padding, align_corners=False) of an all-ones P x P map at
// x_k = (k/(P-1) - x0) / (x1 - x0) etc. (_boxes_to_grid, :119-157).  Sampling a constant map leaves the sum of the in-range
// bilinear weights, separable in x and y: a soft-edged box mask.
struct RawGeom {
    float x0[2], y0[2], ww[2], hh[2];
};

__device__ __forceinline__ RawGeom raw_geom(const float* __restrict__ rois, const int64_t* __restrict__ pairs, long e,
                                            const float* __restrict__ im_wh) {
    RawGeom g;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float* r = rois + pairs[2 * e + i] * 5;
        const int img = (int)r[0];
        const float w = im_wh[2 * img], h = im_wh[2 * img + 1];
        const float x0 = __fdiv_rn(r[1], w), y0 = __fdiv_rn(r[2], h), x1 = __fdiv_rn(r[3], w), y1 = __fdiv_rn(r[4], h);
        g.x0[i] = x0;
        g.y0[i] = y0;
        g.ww[i] = __fsub_rn(x1, x0);
        g.hh[i] = __fsub_rn(y1, y0);
    }
    return g;
}

// sum of the in-range bilinear weights along one axis for output index k of P: unnormalised source coordinate
// ((g + 1) * P - 1) / 2 with g = 2 * ((k/(P-1) - lo) / extent) - 1
__device__ __forceinline__ float axis_weight(int k, int P, float lo, float extent) {
    const float t = __fdiv_rn(__fsub_rn(__fdiv_rn((float)k, (float)(P - 1)), lo), extent);
    const float gcoord = __fsub_rn(__fmul_rn(t, 2.f), 1.f);
    const float src = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(gcoord, 1.f), (float)P), 1.f), 0.5f);
    const float f0 = floorf(src);
    const float frac = __fsub_rn(src, f0);
    float wsum = 0.f;
    if (f0 >= 0.f && f0 <= (float)(P - 1)) wsum += 1.f - frac;
    if (f0 + 1.f >= 0.f && f0 + 1.f <= (float)(P - 1)) wsum += frac;
    return wsum;            // NaN coordinates (degenerate box) compare false everywhere: 0, like grid_sample's padding
}

__device__ __forceinline__ float coverage(const RawGeom& g, int i, int j, int k, int P) {
    return __fmul_rn(axis_weight(k, P, g.x0[i], g.ww[i]), axis_weight(j, P, g.y0[i], g.hh[i]));
}
__device__ __forceinline__ float coverage(const PairGeom& g, int i, int j, int k, int) { return coverage(g, i, j, k); }

template <bool RAW> struct GeomOf { typedef PairGeom type; };
template <> struct GeomOf<true> { typedef RawGeom type; };
template <bool RAW>
__device__ __forceinline__ typename GeomOf<RAW>::type make_geom(const float* rois, const int64_t* pairs, long e, int P,
                                                                 const float* im_wh) {
    if constexpr (RAW) return raw_geom(rois, pairs, e, im_wh);
    else return pair_geom(rois, pairs, e, P);
}

// a-5: out f32[E,2,P,P] (+offset).  One workgroup per pair; consecutive threads write consecutive pixels.
template <bool RAW>
__global__ __launch_bounds__(256) void union_rects_kernel(const float* __restrict__ rois, const int64_t* __restrict__ pairs,
                                                          int P, float offset, float* __restrict__ out,
                                                          const float* __restrict__ im_wh) {
    const long e = blockIdx.x;
    const auto g = make_geom<RAW>(rois, pairs, e, P, im_wh);
    const int n = 2 * P * P;
    float* o = out + e * n;
    for (int t = threadIdx.x; t < n; t += 256) {
        const int i = t / (P * P), rem = t - i * P * P;
        const int j = rem / P, k = rem - j * P;
        o[t] = __fadd_rn(coverage(g, i, j, k, P), offset);
    }
}

// The conv stack of lib/get_union_boxes.py:51-59 uses stride 16 for BOTH convs (typo at :40-43), so the first conv
// (k7,p3) reads only 2x2 windows at rows/cols {-3..3} and {13..19}.  Emit exactly those 4 patches per pair as
// GEMM rows: out[(e*4 + oy*2+ox)][k = c*49 + ky*7 + kx], zero outside the raster, raster-0.5 inside.
template <typename T, bool RAW>
__global__ __launch_bounds__(256) void rect_patches_kernel(const float* __restrict__ rois, const int64_t* __restrict__ pairs,
                                                           int P, T* __restrict__ out, int Kpad, const float* __restrict__ im_wh) {
    const long e = blockIdx.x;
    const auto g = make_geom<RAW>(rois, pairs, e, P, im_wh);
    T* o = out + e * 4 * Kpad;
    for (int t = threadIdx.x; t < 4 * Kpad; t += 256) {
        const int pos = t / Kpad, k = t - pos * Kpad;
        float v = 0.f;
        if (k < 98) {
            const int c = k / 49, r = k - c * 49;
            const int ky = r / 7, kx = r - ky * 7;
            const int j = (pos >> 1) * 16 - 3 + ky, kk = (pos & 1) * 16 - 3 + kx;
            if (j >= 0 && j < P && kk >= 0 && kk < P) v = __fadd_rn(coverage(g, c, j, kk, P), -0.5f);
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

extern "C" int sgg_union_rects_fwd(const float* rois, const int64_t* pairs, int E, int P, float offset, float* out, int raster,
                                   const float* im_wh, void* stream) {
    if (E == 0) return SGG_OK;
    if (!rois || !pairs || !out || E < 0 || P <= 0 || raster < 0 || raster > 1 || (raster == 1 && (!im_wh || P < 2))) return SGG_ERR_ARG;
    if (raster)
        hipLaunchKernelGGL(union_rects_kernel<true>, dim3(E), dim3(256), 0, (hipStream_t)stream, rois, pairs, P, offset, out, im_wh);
    else
        hipLaunchKernelGGL(union_rects_kernel<false>, dim3(E), dim3(256), 0, (hipStream_t)stream, rois, pairs, P, offset, out, im_wh);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_union_rect_patches(const float* rois, const int64_t* pairs, int E, int P, void* out, int Kpad, int raster,
                                      const float* im_wh, int dtype, void* stream) {
    if (E == 0) return SGG_OK;
    // geometry of the typo'd stack: k7/p3/s16 must give a 2x2 map
    if (!rois || !pairs || !out || E < 0 || Kpad < 98 || (P + 6 - 7) / 16 + 1 != 2 || raster < 0 || raster > 1 || (raster == 1 && !im_wh))
        return SGG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
#define SGG_PATCHES(T, RAW) hipLaunchKernelGGL((rect_patches_kernel<T, RAW>), dim3(E), dim3(256), 0, s, rois, pairs, P, (T*)out, Kpad, im_wh)
    if (raster) {
        SGG_FOR_DTYPE(dtype, SGG_PATCHES(T, true));
    } else {
        SGG_FOR_DTYPE(dtype, SGG_PATCHES(T, false));
    }
#undef SGG_PATCHES
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_max4_rows(const void* in, void* out, int E, int C, int dtype, void* stream) {
    if (E == 0) return SGG_OK;
    if (!in || !out || E < 0 || C <= 0 || (C & 7)) return SGG_ERR_ARG;
    const long total = (long)E * (C / 8);
    const int grid = (int)((total + 255) / 256);
    SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL(max4_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)in, (T*)out, total, C));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_bcast_add(void* x, const float* add_rc, int R, int PP, int C, int dtype, void* stream) {
    if (R == 0) return SGG_OK;
    if (!x || !add_rc || R < 0 || PP <= 0 || C <= 0 || (((long)C * PP) & 7)) return SGG_ERR_ARG;
    const long total = (long)R * ((long)C * PP / 8);
    const int grid = (int)((total + 255) / 256);
    SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL(bcast_add_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (T*)x, add_rc, total, PP, C));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}
