// GAN generator data movement: augment/layout.py:33-71 (boxes_to_layout) and the gather / pooling steps of
// augment/graphconv.py:51-119 (GraphTripleConv).  HBM-bound, channels-last.
//
// boxes_to_layout in the reference: grid_sample of every object's patch onto its own H x W canvas ([O, D, H, W]: 757 MB fp32 for
// 256 objects x 512 channels x 38 x 38) and a scatter_add of the canvases per image.  Here one workgroup owns one canvas pixel of
// one image and walks the image's objects in order: the bilinear weights are separable (a row tap pair x a column tap pair, both
// functions of the box only), objects that do not touch the pixel are skipped by the whole workgroup, and the per-object canvases
// never exist: the patches ([O, S, S, D], L2-resident) are read and the layout ([N, H, W, D]) is written once.  The sum runs in
// object order (the reference's scatter_add on the GPU is atomic, i.e. unordered).
#include "common.h"

namespace {

struct Tap {
    int i0;
    float w0, w1;   // weights of source rows i0 and i0 + 1; 0 where that row is outside [0, S) (padding_mode='zeros')
};

// torch.linspace(0, 1, n)[pos] in fp32 (filled from both ends), then the box-relative grid coordinate of layout.py:125-133
// and grid_sample's align_corners=False un-normalisation
__device__ __forceinline__ Tap axis_tap(int pos, int n, float lo, float extent, int S) {
    float lin = 0.f;
    if (n > 1) {
        const float step = 1.0f / (float)(n - 1);
        lin = (pos < n / 2) ? step * (float)pos : 1.0f - step * (float)(n - 1 - pos);
    }
    const float g = ((lin - lo) / extent) * 2.0f - 1.0f;
    const float f = ((g + 1.0f) * (float)S - 1.0f) / 2.0f;
    Tap t;
    if (!(fabsf(f) < 1.0e6f)) {   // also NaN / inf of a zero-extent box: nothing is sampled
        t.i0 = -4; t.w0 = 0.f; t.w1 = 0.f;
        return t;
    }
    const float fl = floorf(f);
    t.i0 = (int)fl;
    t.w1 = f - fl;
    t.w0 = 1.0f - t.w1;
    if (t.i0 < 0 || t.i0 >= S) t.w0 = 0.f;
    if (t.i0 + 1 < 0 || t.i0 + 1 >= S) t.w1 = 0.f;
    return t;
}


// 4 consecutive channels (16 / 8 bytes) when the row allows it, element by element at a ragged end
__device__ __forceinline__ void fma4(const float* src, float w, float (&acc)[4], int left, bool vec) {
    if (vec && left >= 4) {
        const float4 v = *reinterpret_cast<const float4*>(src);
        acc[0] += w * v.x; acc[1] += w * v.y; acc[2] += w * v.z; acc[3] += w * v.w;
    } else {
        for (int c = 0; c < 4 && c < left; ++c) acc[c] += w * src[c];
    }
}
template <typename T>
__device__ __forceinline__ void fma4(const T* src, float w, float (&acc)[4], int left, bool vec) {
    if (vec && left >= 4) {
        const uint2 v = *reinterpret_cast<const uint2*>(src);
        acc[0] += w * H16<T>::lo(v.x); acc[1] += w * H16<T>::hi(v.x);
        acc[2] += w * H16<T>::lo(v.y); acc[3] += w * H16<T>::hi(v.y);
    } else {
        for (int c = 0; c < 4 && c < left; ++c) acc[c] += w * Elem<T>::ld(src + c);
    }
}

// vecs [O, S, S, D] (S == 0: [O, D] vectors, which the reference expands to a constant 8x8 patch, layout.py:57-58).
// The tap pairs of an object at this pixel are computed once per workgroup (one lane per object, parked in LDS), not per lane.
template <typename T>
__global__ __launch_bounds__(128) void layout_fwd_kernel(const T* __restrict__ vecs, const float* __restrict__ boxes,
                                                         const int* __restrict__ obj_img, int O, int S, int D,
                                                         int H, int W, int avg, T* __restrict__ out) {
    __shared__ int s_obj[128], s_iy[128], s_ix[128];
    __shared__ float s_w[128][4];
    __shared__ int s_cnt[2];
    const int pix = blockIdx.x, n = pix / (H * W), y = (pix / W) % H, x = pix % W;
    const int b = 0, e = O;           // every object is looked at (in ascending order); those of other images never hit
    const int Sv = S ? S : 8;
    const bool vec = (D & 3) == 0;
    for (int dbase = 0; dbase < D; dbase += 512) {
        int members = 0;
        const int d0 = dbase + threadIdx.x * 4;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int base = b; base < e; base += 128) {
            __syncthreads();
            const int o = base + threadIdx.x;
            const bool mine = o < e && obj_img[o] == n;
            Tap ty, tx;
            bool hit = false;
            if (mine) {
                const float x0 = boxes[4 * o], y0 = boxes[4 * o + 1], x1 = boxes[4 * o + 2], y1 = boxes[4 * o + 3];
                ty = axis_tap(y, H, y0, y1 - y0, Sv);
                tx = axis_tap(x, W, x0, x1 - x0, Sv);
                hit = (ty.w0 != 0.f || ty.w1 != 0.f) && (tx.w0 != 0.f || tx.w1 != 0.f);
            }
            // the objects that touch this pixel, compacted in object order (most of an image's objects do not)
            const unsigned long long hm = __ballot(hit);
            const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
            if (lane == 0) s_cnt[wave] = __popcll(hm);
            members += __syncthreads_count(mine);
            if (hit) {
                const int pos = (wave ? s_cnt[0] : 0) + __popcll(hm & ((1ull << lane) - 1ull));
                s_obj[pos] = o;
                s_iy[pos] = ty.i0;
                s_ix[pos] = tx.i0;
                s_w[pos][0] = ty.w0; s_w[pos][1] = ty.w1; s_w[pos][2] = tx.w0; s_w[pos][3] = tx.w1;
            }
            __syncthreads();
            const int m = s_cnt[0] + s_cnt[1];
            if (d0 >= D) continue;
            for (int j = 0; j < m; ++j) {
                const int o = s_obj[j];
                const float wy0 = s_w[j][0], wy1 = s_w[j][1], wx0 = s_w[j][2], wx1 = s_w[j][3];
                if (S == 0) {
                    const float w = (wy0 + wy1) * (wx0 + wx1);
                    const T* src = vecs + (long)o * D + d0;
                    fma4(src, w, acc, D - d0, vec);
                    continue;
                }
                const int iy = s_iy[j], ix = s_ix[j];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const float wy = a ? wy1 : wy0;
                    if (wy == 0.f) continue;
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2) {
                        const float w = wy * (c2 ? wx1 : wx0);
                        if (w == 0.f) continue;
                        const T* src = vecs + (((long)o * S + iy + a) * S + ix + c2) * D + d0;
                        fma4(src, w, acc, D - d0, vec);
                    }
                }
            }
        }
        const float scale = (avg && members > 0) ? 1.0f / (float)members : 1.0f;
        if (d0 < D) {
            T* dst = out + (long)pix * D + d0;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (d0 + c < D) Elem<T>::st(dst + c, acc[c] * scale);
        }
    }
}

// gradient wrt the patches: one workgroup per (object, source row, source column) gathers from the canvas pixels that sample it
// (S == 0: one workgroup per object, the weight of a pixel is the sum of its valid taps).  The row / column weights of the cell
// are computed once into LDS; only rows and columns with a non-zero weight are visited.
#define SGG_LAYOUT_MAX_SIDE 256
template <typename T>
__global__ __launch_bounds__(128) void layout_bwd_kernel(const T* __restrict__ d_out, const float* __restrict__ boxes,
                                                         const int* __restrict__ obj_img, const int* __restrict__ counts, int S,
                                                         int D, int H, int W, int avg, T* __restrict__ d_vecs) {
    __shared__ float s_wy[SGG_LAYOUT_MAX_SIDE], s_wx[SGG_LAYOUT_MAX_SIDE];
    __shared__ int s_range[4];
    const int Sv = S ? S : 8;
    const int cells = S ? S * S : 1;
    const int o = blockIdx.x / cells, cell = blockIdx.x % cells, iy = cell / (S ? S : 1), ix = cell % (S ? S : 1);
    const int n = obj_img[o];
    const float x0 = boxes[4 * o], y0 = boxes[4 * o + 1], x1 = boxes[4 * o + 2], y1 = boxes[4 * o + 3];
    const int cnt = avg ? counts[n] : 1;
    const float scale = (avg && cnt > 0) ? 1.0f / (float)cnt : 1.0f;
    if (threadIdx.x == 0) { s_range[0] = H; s_range[1] = -1; s_range[2] = W; s_range[3] = -1; }
    __syncthreads();
    for (int i = threadIdx.x; i < H; i += 128) {
        const Tap t = axis_tap(i, H, y0, y1 - y0, Sv);
        const float w = S ? ((t.i0 == iy ? t.w0 : 0.f) + (t.i0 + 1 == iy ? t.w1 : 0.f)) : (t.w0 + t.w1);
        s_wy[i] = w;
        if (w != 0.f) { atomicMin(&s_range[0], i); atomicMax(&s_range[1], i); }
    }
    for (int i = threadIdx.x; i < W; i += 128) {
        const Tap t = axis_tap(i, W, x0, x1 - x0, Sv);
        const float w = S ? ((t.i0 == ix ? t.w0 : 0.f) + (t.i0 + 1 == ix ? t.w1 : 0.f)) : (t.w0 + t.w1);
        s_wx[i] = w;
        if (w != 0.f) { atomicMin(&s_range[2], i); atomicMax(&s_range[3], i); }
    }
    __syncthreads();
    const int ya = s_range[0], yb = s_range[1], xa = s_range[2], xb = s_range[3];
    const bool vec = (D & 3) == 0;
    for (int d0 = threadIdx.x * 4; d0 < D; d0 += 512) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int y = ya; y <= yb; ++y) {
            const float wy = s_wy[y];
            if (wy == 0.f) continue;
            for (int x = xa; x <= xb; ++x) {
                const float w = wy * s_wx[x];
                if (w == 0.f) continue;
                const T* src = d_out + (((long)n * H + y) * W + x) * D + d0;
                fma4(src, w, acc, D - d0, vec);
            }
        }
        T* dst = d_vecs + (long)blockIdx.x * D + d0;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (d0 + c < D) Elem<T>::st(dst + c, acc[c] * scale);
    }
}

// graphconv.py:68-78: row t of the net1 input = [obj[s_t] | pred[t] | obj[o_t]]
template <typename T>
__global__ __launch_bounds__(256) void triple_gather_kernel(const T* __restrict__ obj, const T* __restrict__ pred,
                                                            const int64_t* __restrict__ edges, int Tn, int Din, int De,
                                                            T* __restrict__ out) {
    const int t = blockIdx.x;
    const int64_t s = edges[2 * t], o = edges[2 * t + 1];
    const int Wd = 2 * Din + De;
    for (int c = threadIdx.x; c < Wd; c += 256) {
        T v;
        if (c < Din) v = obj[s * Din + c];
        else if (c < Din + De) v = pred[(long)t * De + (c - Din)];
        else v = obj[o * Din + (c - Din - De)];
        out[(long)t * Wd + c] = v;
    }
}

// graphconv.py:93-115: pooled[n] = (sum of the subject parts of n's outgoing triples + object parts of its incoming ones) / count.
// rows = the net1 output [T, ld]; the subject part starts at column 0, the object part at column o_off.  CSR lists (ascending
// triple order inside a node) make the sum deterministic; the backward of this op is triple_pool_bwd below.
template <typename T>
__global__ __launch_bounds__(128) void triple_pool_kernel(const T* __restrict__ rows, int ld, int o_off, const int* __restrict__ out_ptr,
                                                          const int* __restrict__ out_ids, const int* __restrict__ in_ptr,
                                                          const int* __restrict__ in_ids, int Hd, int avg, T* __restrict__ pooled) {
    const int n = blockIdx.x;
    const int ob = out_ptr[n], oe = out_ptr[n + 1], ib = in_ptr[n], ie = in_ptr[n + 1];
    const int cnt = (oe - ob) + (ie - ib);
    const float scale = (avg && cnt > 1) ? 1.0f / (float)cnt : 1.0f;
    for (int c = threadIdx.x; c < Hd; c += 128) {
        float acc = 0.f;
        for (int k = ob; k < oe; ++k) acc += Elem<T>::ld(rows + (long)out_ids[k] * ld + c);
        for (int k = ib; k < ie; ++k) acc += Elem<T>::ld(rows + (long)in_ids[k] * ld + o_off + c);
        Elem<T>::st(pooled + (long)n * Hd + c, acc * scale);
    }
}

// d_rows[t, 0:Hd] = d_pooled[s_t] / count(s_t); d_rows[t, o_off:o_off+Hd] = d_pooled[o_t] / count(o_t)   (other columns untouched)
template <typename T>
__global__ __launch_bounds__(128) void triple_pool_bwd_kernel(const T* __restrict__ d_pooled, const int64_t* __restrict__ edges,
                                                              const int* __restrict__ out_ptr, const int* __restrict__ in_ptr, int Hd,
                                                              int avg, int ld, int o_off, T* __restrict__ d_rows) {
    const int t = blockIdx.x;
    const int64_t s = edges[2 * t], o = edges[2 * t + 1];
    const int cs = (out_ptr[s + 1] - out_ptr[s]) + (in_ptr[s + 1] - in_ptr[s]);
    const int co = (out_ptr[o + 1] - out_ptr[o]) + (in_ptr[o + 1] - in_ptr[o]);
    const float ss = (avg && cs > 1) ? 1.0f / (float)cs : 1.0f, so = (avg && co > 1) ? 1.0f / (float)co : 1.0f;
    for (int c = threadIdx.x; c < Hd; c += 128) {
        Elem<T>::st(d_rows + (long)t * ld + c, Elem<T>::ld(d_pooled + s * Hd + c) * ss);
        Elem<T>::st(d_rows + (long)t * ld + o_off + c, Elem<T>::ld(d_pooled + o * Hd + c) * so);
    }
}

}  // namespace

extern "C" int sgg_boxes_to_layout_fwd(const void* vecs, const float* boxes, const int* obj_img, int N, int O, int S, int D, int H,
                                       int W, int avg, void* out, int dtype, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0 || D <= 0) return SGG_OK;
    if (!out || O < 0 || S < 0 || (O > 0 && (!vecs || !boxes || !obj_img))) return SGG_ERR_ARG;
    const dim3 grid((unsigned)((long)N * H * W)), blk(128);
    hipStream_t s = (hipStream_t)stream;
    SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL(layout_fwd_kernel<T>, grid, blk, 0, s, (const T*)vecs, boxes, obj_img, O, S, D, H, W, avg, (T*)out));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_boxes_to_layout_bwd(const void* d_out, const float* boxes, const int* obj_img, const int* counts, int O, int S,
                                       int D, int H, int W, int avg, void* d_vecs, int dtype, void* stream) {
    if (O <= 0 || D <= 0) return SGG_OK;
    if (!d_out || !boxes || !obj_img || (avg && !counts) || !d_vecs || S < 0 || H <= 0 || W <= 0 || H > SGG_LAYOUT_MAX_SIDE ||
        W > SGG_LAYOUT_MAX_SIDE)
        return SGG_ERR_ARG;
    const dim3 grid((unsigned)((long)O * (S ? S * S : 1))), blk(128);
    hipStream_t s = (hipStream_t)stream;
    SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL(layout_bwd_kernel<T>, grid, blk, 0, s, (const T*)d_out, boxes, obj_img, counts, S, D, H, W, avg, (T*)d_vecs));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_triple_gather(const void* obj, const void* pred, const int64_t* edges, int T, int Din, int De, void* out, int dtype,
                                 void* stream) {
    if (T <= 0) return SGG_OK;
    if (!obj || !pred || !edges || !out || Din <= 0 || De <= 0) return SGG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int nT = T;     // (`T` names the element type inside SGG_FOR_DTYPE)
    SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL(triple_gather_kernel<T>, dim3(nT), dim3(256), 0, s, (const T*)obj, (const T*)pred, edges, nT, Din, De, (T*)out));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_triple_pool_fwd(const void* rows, int ld, int o_off, const int* out_ptr, const int* out_ids, const int* in_ptr,
                                   const int* in_ids, int O, int Hd, int avg, void* pooled, int dtype, void* stream) {
    if (O <= 0 || Hd <= 0) return SGG_OK;
    if (!out_ptr || !in_ptr || !pooled || ld < Hd || o_off < 0 || o_off + Hd > ld) return SGG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL(triple_pool_kernel<T>, dim3(O), dim3(128), 0, s, (const T*)rows, ld, o_off, out_ptr, out_ids, in_ptr, in_ids, Hd, avg, (T*)pooled));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_triple_pool_bwd(const void* d_pooled, const int64_t* edges, const int* out_ptr, const int* in_ptr, int T, int Hd,
                                   int avg, int ld, int o_off, void* d_rows, int dtype, void* stream) {
    if (T <= 0 || Hd <= 0) return SGG_OK;
    if (!d_pooled || !edges || !out_ptr || !in_ptr || !d_rows || ld < Hd || o_off < 0 || o_off + Hd > ld) return SGG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int nT = T;     // (`T` names the element type inside SGG_FOR_DTYPE)
    SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL(triple_pool_bwd_kernel<T>, dim3(nT), dim3(128), 0, s, (const T*)d_pooled, edges, out_ptr, in_ptr, Hd, avg, ld, o_off, (T*)d_rows));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}
