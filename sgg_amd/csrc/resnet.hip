// Glue kernels of the ResNet-50-FPN feature extractor (GQA configuration; sgg_models/rel_model_base.py:58-81): everything
// that is not a contraction.  The contractions themselves -- 1x1 convolutions on NHWC rows, the 7x7 / strided 3x3 convolutions as
// patch matrices, the 3x3 stride-1 convolutions on zero-bordered planes -- run on the GEMM / spatial-conv kernels of gemm.hip,
// gemm256.hip and conv_spatial.hip.  All HBM-bound, 16-byte pieces per lane (8 bf16 / 4 f32 channels of one pixel).
#include "common.h"
#include "../../include/sgg_hip.h"

namespace {

// Patch matrix of a k x k / stride s / padding p convolution: row = output pixel (b, yo, xo), columns (ky, kx, c), zero-filled to
// Kp columns.  src: [B, H + 2 sp, W + 2 sp, Ca] (sp = border already in the plane, Ca allocated channels, C <= Ca used).
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void im2col_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int H, int W, int Ca, int C, int sp,
                                                     int k, int s, int p, int Ho, int Wo, int Kp, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;          // one thread per (row, column)
    if (i >= total) return;
    const int col = (int)(i % Kp);
    long r = i / Kp;
    const int xo = (int)(r % Wo);
    r /= Wo;
    const int yo = (int)(r % Ho);
    const int b = (int)(r / Ho);
    float v = 0.f;
    if (col < k * k * C) {
        const int c = col % C, t = col / C, kx = t % k, ky = t / k;
        const int y = yo * s - p + ky, x = xo * s - p + kx;
        if (y >= 0 && y < H && x >= 0 && x < W) v = Elem<TS>::ld(src + (((long)b * (H + 2 * sp) + y + sp) * (W + 2 * sp) + x + sp) * Ca + c);
    }
    Elem<TD>::st(dst + i, v);
}

// the same with 8 channels per thread (C, Ca, Kp multiples of 8; same element type): a 16-byte piece in, a 16-byte piece out
template <typename T>
__global__ __launch_bounds__(256) void im2col8_kernel(const T* __restrict__ src, T* __restrict__ dst, int H, int W, int Ca, int C, int sp, int k,
                                                      int s, int p, int Ho, int Wo, int Kp, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;          // one thread per (row, 8 columns)
    if (i >= total) return;
    const int K8 = Kp >> 3;
    const int col = (int)(i % K8) * 8;
    const long row = i / K8;
    long r = row;
    const int xo = (int)(r % Wo);
    r /= Wo;
    const int yo = (int)(r % Ho);
    const int b = (int)(r / Ho);
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (col < k * k * C) {
        const int c = col % C, t = col / C, kx = t % k, ky = t / k;
        const int y = yo * s - p + ky, x = xo * s - p + kx;
        if (y >= 0 && y < H && x >= 0 && x < W) load8(src + (((long)b * (H + 2 * sp) + y + sp) * (W + 2 * sp) + x + sp) * Ca + c, v);
    }
    store8(dst + row * Kp + col, v);
}

// Adjoint of im2col (the input gradient of a convolution run as patch matrix x GEMM): d_src[b,y,x,c] = sum over the taps (ky,kx) whose
// output pixel (yo,xo) = ((y + p - ky) / s, (x + p - kx) / s) exists of d_cols[(b,yo,xo), (ky,kx,c)].  One thread per input element: a
// gather, every sum in a fixed order (no atomics).  d_cols [B*Ho*Wo, Kp], d_src [B,H,W,C] dense.
template <typename T>
__global__ __launch_bounds__(256) void col2im_kernel(const T* __restrict__ cols, T* __restrict__ dsrc, int H, int W, int C, int k, int s, int p,
                                                     int Ho, int Wo, int Kp, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    long r = i / C;
    const int x = (int)(r % W);
    r /= W;
    const int y = (int)(r % H);
    const int b = (int)(r / H);
    float acc = 0.f;
    for (int ky = 0; ky < k; ++ky) {
        const int ty = y + p - ky;
        if (ty < 0 || ty % s) continue;
        const int yo = ty / s;
        if (yo >= Ho) continue;
        for (int kx = 0; kx < k; ++kx) {
            const int tx = x + p - kx;
            if (tx < 0 || tx % s) continue;
            const int xo = tx / s;
            if (xo >= Wo) continue;
            acc += Elem<T>::ld(cols + (((long)b * Ho + yo) * Wo + xo) * Kp + (ky * k + kx) * C + c);
        }
    }
    Elem<T>::st(dsrc + i, acc);
}

// MaxPool2d(kernel 3, stride 2, padding 1) on NHWC rows [B, H, W, C] -> [B, Ho, Wo, C]
template <typename T>
__global__ __launch_bounds__(256) void maxpool3s2_kernel(const T* __restrict__ in, T* __restrict__ out, int H, int W, int C, int Ho, int Wo,
                                                         long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c8 = C >> 3;
    const int cc = (int)(i % c8) * 8;
    long r = i / c8;
    const int xo = (int)(r % Wo);
    r /= Wo;
    const int yo = (int)(r % Ho);
    const int b = (int)(r / Ho);
    float a[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = -3.0e38f;
    for (int ky = 0; ky < 3; ++ky) {
        const int y = 2 * yo - 1 + ky;
        if (y < 0 || y >= H) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int x = 2 * xo - 1 + kx;
            if (x < 0 || x >= W) continue;
            float t[8];
            load8(in + (((long)b * H + y) * W + x) * C + cc, t);
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = fmaxf(a[q], t[q]);
        }
    }
    store8(out + i * 8, a);
}

// dst[b, y, x, :] = src[b, y s, x s, :] between planes with their own borders (rows -> zero-bordered plane, stride-2 subsampling)
template <typename T>
__global__ __launch_bounds__(256) void plane_copy_kernel(const T* __restrict__ src, T* __restrict__ dst, int Hs, int Ws, int sp, int Hd, int Wd,
                                                         int dp, int C, int s, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c8 = C >> 3;
    const int cc = (int)(i % c8) * 8;
    long r = i / c8;
    const int x = (int)(r % Wd);
    r /= Wd;
    const int y = (int)(r % Hd);
    const int b = (int)(r / Hd);
    float v[8];
    load8(src + (((long)b * (Hs + 2 * sp) + y * s + sp) * (Ws + 2 * sp) + x * s + sp) * C + cc, v);
    store8(dst + (((long)b * (Hd + 2 * dp) + y + dp) * (Wd + 2 * dp) + x + dp) * C + cc, v);
}

// FPN top-down join: y[b,y,x,:] += top[b, sy(y), sx(x), :], nearest-neighbour source index floor(dst * in / out) computed in fp32 as
// F.interpolate(mode='nearest') does (clamped to the last row / column).  8 channels per thread.
template <typename T>
__global__ __launch_bounds__(256) void upsample_add_kernel(T* __restrict__ y, const T* __restrict__ top, int H, int W, int Ht, int Wt, int C,
                                                           long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c8 = C >> 3;
    const int cc = (int)(i % c8) * 8;
    long r = i / c8;
    const int x = (int)(r % W);
    r /= W;
    const int yy = (int)(r % H);
    const int b = (int)(r / H);
    const float fy = (float)Ht / (float)H, fx = (float)Wt / (float)W;
    const int sy = min((int)floorf((float)yy * fy), Ht - 1), sx = min((int)floorf((float)x * fx), Wt - 1);
    float a[8], t[8];
    load8(y + i * 8, a);
    load8(top + (((long)b * Ht + sy) * Wt + sx) * C + cc, t);
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] += t[q];
    store8(y + i * 8, a);
}

// y = max(y + x, 0)  (the residual join of a bottleneck block)
template <typename T>
__global__ __launch_bounds__(256) void add_relu_kernel(T* __restrict__ y, const T* __restrict__ x, long n8) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    float a[8], b[8];
    load8(y + i * 8, a);
    load8(x + i * 8, b);
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = fmaxf(a[q] + b[q], 0.f);
    store8(y + i * 8, a);
}

inline dim3 grid_for(long total) { return dim3((unsigned)((total + 255) / 256)); }

}  // namespace

extern "C" int sgg_im2col(const void* src, int B, int H, int W, int Ca, int C, int src_pad, int k, int stride, int pad, int Ho, int Wo,
                          void* dst, int Kp, int src_dtype, int dst_dtype, void* stream) {
    if (B == 0) return SGG_OK;
    if (!src || !dst || B < 0 || H <= 0 || W <= 0 || C <= 0 || C > Ca || k <= 0 || stride <= 0 || pad < 0 || src_pad < 0 || Kp < k * k * C)
        return SGG_ERR_ARG;
    if (Ho != (H + 2 * pad - k) / stride + 1 || Wo != (W + 2 * pad - k) / stride + 1) return SGG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const bool vec = src_dtype == dst_dtype && C % 8 == 0 && Ca % 8 == 0 && Kp % 8 == 0;
    if (vec) {
        const long total = (long)B * Ho * Wo * (Kp / 8);
        SGG_FOR_DTYPE(src_dtype, hipLaunchKernelGGL(im2col8_kernel<T>, grid_for(total), dim3(256), 0, s, (const T*)src, (T*)dst, H, W, Ca, C, src_pad, k, stride,
                                                    pad, Ho, Wo, Kp, total));
    } else {
        const long total = (long)B * Ho * Wo * Kp;
        if (src_dtype != SGG_F32 && src_dtype != dst_dtype) return SGG_ERR_DTYPE;      // f32 -> any, or like -> like
        SGG_FOR_DTYPE2(src_dtype, dst_dtype, hipLaunchKernelGGL((im2col_kernel<TA, TB>), grid_for(total), dim3(256), 0, s, (const TA*)src, (TB*)dst, H, W, Ca, C,
                                                                src_pad, k, stride, pad, Ho, Wo, Kp, total));
    }
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_col2im(const void* d_cols, int B, int H, int W, int C, int k, int stride, int pad, int Ho, int Wo, int Kp, void* d_src, int dtype,
                          void* stream) {
    if (B == 0) return SGG_OK;
    if (!d_cols || !d_src || B < 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || stride <= 0 || pad < 0 || Kp < k * k * C) return SGG_ERR_ARG;
    if (Ho != (H + 2 * pad - k) / stride + 1 || Wo != (W + 2 * pad - k) / stride + 1) return SGG_ERR_ARG;
    const long total = (long)B * H * W * C;
    SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL(col2im_kernel<T>, grid_for(total), dim3(256), 0, (hipStream_t)stream, (const T*)d_cols, (T*)d_src, H, W, C, k,
                                            stride, pad, Ho, Wo, Kp, total));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_upsample_add(void* y, const void* top, int B, int H, int W, int Ht, int Wt, int C, int dtype, void* stream) {
    if (B == 0) return SGG_OK;
    if (!y || !top || B < 0 || H <= 0 || W <= 0 || Ht <= 0 || Wt <= 0 || C <= 0 || (C & 7)) return SGG_ERR_ARG;
    const long total = (long)B * H * W * (C / 8);
    SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL(upsample_add_kernel<T>, grid_for(total), dim3(256), 0, (hipStream_t)stream, (T*)y, (const T*)top, H, W, Ht, Wt, C,
                                            total));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_maxpool3x3s2(const void* in, void* out, int B, int H, int W, int C, int dtype, void* stream) {
    if (B == 0) return SGG_OK;
    if (!in || !out || B < 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7)) return SGG_ERR_ARG;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long total = (long)B * Ho * Wo * (C / 8);
    hipStream_t s = (hipStream_t)stream;
    SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL(maxpool3s2_kernel<T>, grid_for(total), dim3(256), 0, s, (const T*)in, (T*)out, H, W, C, Ho, Wo, total));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_plane_copy(const void* src, int Hs, int Ws, int src_pad, void* dst, int Hd, int Wd, int dst_pad, int B, int C, int stride,
                              int dtype, void* stream) {
    if (B == 0) return SGG_OK;
    if (!src || !dst || B < 0 || Hs <= 0 || Ws <= 0 || Hd <= 0 || Wd <= 0 || C <= 0 || (C & 7) || stride <= 0 || src_pad < 0 || dst_pad < 0)
        return SGG_ERR_ARG;
    if ((Hd - 1) * stride >= Hs || (Wd - 1) * stride >= Ws) return SGG_ERR_ARG;
    const long total = (long)B * Hd * Wd * (C / 8);
    hipStream_t s = (hipStream_t)stream;
    SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL(plane_copy_kernel<T>, grid_for(total), dim3(256), 0, s, (const T*)src, (T*)dst, Hs, Ws, src_pad, Hd, Wd, dst_pad, C,
                                            stride, total));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_add_relu(void* y, const void* x, int64_t n, int dtype, void* stream) {
    if (n == 0) return SGG_OK;
    if (!y || !x || n < 0 || (n & 7)) return SGG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL(add_relu_kernel<T>, grid_for(n / 8), dim3(256), 0, s, (T*)y, (const T*)x, (long)(n / 8)));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}
