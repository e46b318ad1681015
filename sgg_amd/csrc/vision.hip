// Detector front-end pieces that are HBM-bound: image prep, VGG conv1_1 (Cin=3), 2x2 max-pool, RoIAlign.
// Layout: zero-bordered channels-last planes, so the 3x3 convs never test image borders.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// a-1  [3P] GeneralizedRCNNTransform (called at sgg_models/rel_model_base.py:183): normalise, bilinear
// resize (align_corners=False, scale recomputed from sizes), into the interior of a zero NHWC4 plane.
// ------------------------------------------------------------------------------------------------
// Pixel sources: a planar fp32 CHW tensor in [0,1] (what the reference's ToTensor hands over), or the decoded
// u8 HWC image itself -- then SquarePad (dataloaders/image_transforms.py:8-13: pad right/bottom to a square with
// the fill colour int(mean*256)) and ToTensor (u8 / 255) happen here, and the PCIe copy is 4x smaller.
struct SrcF32 {
    const float* img;
    int h, w;
    __device__ __forceinline__ float at(int c, int y, int x) const { return img[((long)c * h + y) * w + x]; }
};
struct SrcU8 {
    const unsigned char* img;   // [h0, w0, 3]
    int h0, w0;
    __device__ __forceinline__ float at(int c, int y, int x) const {
        const int fill[3] = {124, 116, 103};   // int(0.485*256), int(0.456*256), int(0.406*256)
        const int u = (y < h0 && x < w0) ? (int)img[((long)y * w0 + x) * 3 + c] : fill[c];
        return __fdiv_rn((float)u, 255.0f);
    }
};

template <typename Src>
__global__ __launch_bounds__(256) void image_prep_kernel(const Src src, int h, int w, int rh, int rw,
                                                         float* __restrict__ out, int b, int Hp, int Wp) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= rw || y >= rh) return;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    float v[3];
    if (rh == h && rw == w) {
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = (src.at(c, y, x) - mean[c]) / stdv[c];
    } else {
        const float sy = (float)h / (float)rh, sx = (float)w / (float)rw;
        float fy = sy * ((float)y + 0.5f) - 0.5f, fx = sx * ((float)x + 0.5f) - 0.5f;
        fy = fy < 0.f ? 0.f : fy;
        fx = fx < 0.f ? 0.f : fx;
        const int y0 = min((int)fy, h - 1), x0 = min((int)fx, w - 1);
        const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float a = (src.at(c, y0, x0) - mean[c]) / stdv[c], bq = (src.at(c, y0, x1) - mean[c]) / stdv[c];
            const float cq = (src.at(c, y1, x0) - mean[c]) / stdv[c], d = (src.at(c, y1, x1) - mean[c]) / stdv[c];
            v[c] = hy * (hx * a + lx * bq) + ly * (hx * cq + lx * d);
        }
    }
    f32x4 o = {v[0], v[1], v[2], 0.f};
    *reinterpret_cast<f32x4*>(out + (((long)b * (Hp + 2) + y + 1) * (Wp + 2) + x + 1) * 4) = o;
}

// Whole batch in one launch: blockIdx.z = image; per-image source / size parameters travel by value (<= PREP_MAX images).
constexpr int PREP_MAX = 16;
struct PrepTab {
    const void* img[PREP_MAX];
    int h0[PREP_MAX], w0[PREP_MAX], rh[PREP_MAX], rw[PREP_MAX];
    unsigned char u8[PREP_MAX];
};
__global__ __launch_bounds__(256) void image_prep_batch_kernel(const PrepTab t, float* __restrict__ out, int b0, int Hp, int Wp) {
    const int i = blockIdx.z, x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    const int rh = t.rh[i], rw = t.rw[i];
    if (x >= rw || y >= rh) return;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    const int h0 = t.h0[i], w0 = t.w0[i];
    const bool u8 = t.u8[i] != 0;
    const int h = u8 ? max(h0, w0) : h0, w = u8 ? max(h0, w0) : w0;       // SquarePad: the transform sees a square
    const SrcF32 sf{(const float*)t.img[i], h0, w0};
    const SrcU8 su{(const unsigned char*)t.img[i], h0, w0};
    auto at = [&](int c, int yy, int xx) { return u8 ? su.at(c, yy, xx) : sf.at(c, yy, xx); };
    float v[3];
    if (rh == h && rw == w) {
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = (at(c, y, x) - mean[c]) / stdv[c];
    } else {
        const float sy = (float)h / (float)rh, sx = (float)w / (float)rw;
        float fy = sy * ((float)y + 0.5f) - 0.5f, fx = sx * ((float)x + 0.5f) - 0.5f;
        fy = fy < 0.f ? 0.f : fy;
        fx = fx < 0.f ? 0.f : fx;
        const int y0 = min((int)fy, h - 1), x0 = min((int)fx, w - 1);
        const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float a = (at(c, y0, x0) - mean[c]) / stdv[c], bq = (at(c, y0, x1) - mean[c]) / stdv[c];
            const float cq = (at(c, y1, x0) - mean[c]) / stdv[c], d = (at(c, y1, x1) - mean[c]) / stdv[c];
            v[c] = hy * (hx * a + lx * bq) + ly * (hx * cq + lx * d);
        }
    }
    f32x4 o = {v[0], v[1], v[2], 0.f};
    *reinterpret_cast<f32x4*>(out + (((long)(b0 + i) * (Hp + 2) + y + 1) * (Wp + 2) + x + 1) * 4) = o;
}

// ------------------------------------------------------------------------------------------------
// a-2  conv1_1 (3 -> 64, 3x3, pad 1) + ReLU, fp32 VALU (K = 27 is too thin for MFMA, the layer is
// write-bound: 64 outputs per 27 inputs).  256 threads = 64 pixels x 4 groups of 16 output channels.
// ------------------------------------------------------------------------------------------------
template <typename OutT, bool PAIR = false>
__global__ __launch_bounds__(256) void conv1_1_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                      const float* __restrict__ bias, OutT* __restrict__ out, long npix,
                                                      int H, int W) {
    __shared__ float ws[27 * 64];
    __shared__ float bs[64];
    for (int i = threadIdx.x; i < 27 * 64; i += 256) {
        const int k = i >> 6, co = i & 63;
        ws[i] = w[co * 27 + k];
    }
    if (threadIdx.x < 64) bs[threadIdx.x] = bias[threadIdx.x];
    __syncthreads();
    const long pix = (long)blockIdx.x * 64 + (threadIdx.x >> 2);
    const int cg = (threadIdx.x & 3) * 16;
    if (pix >= npix) return;
    const long hw = (long)H * W;
    const int b = (int)(pix / hw);
    const int rem = (int)(pix - (long)b * hw);
    const int y = rem / W, x = rem - y * W;
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = bs[cg + j];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const f32x4 p = *reinterpret_cast<const f32x4*>(in + (((long)b * (H + 2) + y + ky) * (W + 2) + x + kx) * 4);
            const float pv[3] = {p.x, p.y, p.z};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float* wk = ws + ((ky * 3 + kx) * 3 + c) * 64 + cg;
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[j] = fmaf(pv[c], wk[j], acc[j]);
            }
        }
    float o0[8], o1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        o0[j] = fmaxf(acc[j], 0.f);
        o1[j] = fmaxf(acc[8 + j], 0.f);
    }
    if constexpr (PAIR) {        // x3 mode: the [hi (64) | lo (64)] pixel of a pair plane (SGG_PAIR16)
        float h0[8], h1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            h0[j] = round_as<f16_t>(o0[j]);
            h1[j] = round_as<f16_t>(o1[j]);
            o0[j] -= h0[j];
            o1[j] -= h1[j];
        }
        OutT* op = out + (((long)b * (H + 2) + y + 1) * (W + 2) + x + 1) * 128 + cg;
        store8(op, h0);
        store8(op + 8, h1);
        store8(op + 64, o0);
        store8(op + 72, o1);
    } else {
        OutT* op = out + (((long)b * (H + 2) + y + 1) * (W + 2) + x + 1) * 64 + cg;
        store8(op, o0);
        store8(op + 8, o1);
    }
}

// ------------------------------------------------------------------------------------------------
// conv1_1 on the matrix cores (16-bit storage modes).  K = 27 padded to 32 = two v_mfma_f32_32x32x16 k-steps.
// A wave owns 32 consecutive pixels of one image row x all 64 output channels: each lane gathers its pixel's 3x3x3
// patch with nine 16-byte loads (NHWC4 fp32, coalesced along x), packs the k-slice its MFMA fragment needs to the 16-bit format,
// and the 32x64 result goes through LDS so that the wave writes one contiguous 4-KiB piece of the output row.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void conv1_1_mfma_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                           const float* __restrict__ bias, T* __restrict__ out,
                                                           int nseg, int segs_per_row, int H, int W) {
    constexpr int LROW = 136;  // bytes per staged pixel row (128 + 8: conflict-free ds_write_b64)
    __shared__ __attribute__((aligned(16))) char stage[4][32 * LROW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 31, h = lane >> 5;
    // weight fragments: n = nt*32 + m, k = s*16 + h*8 + j  (k = (ky*3+kx)*3 + c, zero for k >= 27)
    u32x4 wf[2][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = s * 16 + h * 8 + j;
                v[j] = k < 27 ? w[(nt * 32 + m) * 27 + k] : 0.f;
            }
            wf[nt][s] = pack8<T>(v);
        }
    float bv[2][4][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c) bv[nt][q][c] = bias[nt * 32 + 8 * q + 4 * h + c];
    char* st = stage[wave];
    for (int seg = blockIdx.x * 4 + wave; seg < nseg; seg += gridDim.x * 4) {
        const int row = seg / segs_per_row, x0 = (seg - row * segs_per_row) * 32;
        const int b = row / H, y = row - b * H;
        const int x = min(x0 + m, W - 1);
        float V[32];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const f32x4 p = *reinterpret_cast<const f32x4*>(in + (((long)b * (H + 2) + y + ky) * (W + 2) + x + kx) * 4);
                V[(ky * 3 + kx) * 3 + 0] = p.x;
                V[(ky * 3 + kx) * 3 + 1] = p.y;
                V[(ky * 3 + kx) * 3 + 2] = p.z;
            }
#pragma unroll
        for (int k = 27; k < 32; ++k) V[k] = 0.f;
        u32x4 af[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = h ? V[s * 16 + 8 + j] : V[s * 16 + j];
            af[s] = pack8<T>(t);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s)
                acc = mfma_32x32x16<DtypeOf<T>::value>(wf[nt][s], af[s], acc);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float o0 = fmaxf(acc[4 * q] + bv[nt][q][0], 0.f), o1 = fmaxf(acc[4 * q + 1] + bv[nt][q][1], 0.f);
                const float o2 = fmaxf(acc[4 * q + 2] + bv[nt][q][2], 0.f), o3 = fmaxf(acc[4 * q + 3] + bv[nt][q][3], 0.f);
                u32x2 pk = {H16<T>::pack(o0, o1), H16<T>::pack(o2, o3)};
                *reinterpret_cast<u32x2*>(st + m * LROW + (nt * 32 + 8 * q + 4 * h) * 2) = pk;
            }
        }
        // wave-private staging: LDS ops of one wave complete in order, no barrier needed; the fence keeps the
        // compiler from moving the reads above the writes
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        T* orow = out + (((long)b * (H + 2) + y + 1) * (W + 2) + x0 + 1) * 64;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int px = it * 8 + (lane >> 3), ch = (lane & 7) * 16;
            const u32x2 lo = *reinterpret_cast<const u32x2*>(st + px * LROW + ch);
            const u32x2 hi = *reinterpret_cast<const u32x2*>(st + px * LROW + ch + 8);
            if (x0 + px < W) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(orow) + px * 128 + ch) = u32x4{lo.x, lo.y, hi.x, hi.y};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// ------------------------------------------------------------------------------------------------
// 2x2 / stride-2 max pool between the VGG blocks (thread = output pixel x 8 channels)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool_kernel(const T* __restrict__ in, T* __restrict__ out, int op, int H, int W,
                                                      int C, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c8 = C >> 3;
    const int cc = (int)(i % c8) * 8;
    long p = i / c8;
    const int Wo = W >> 1, Ho = H >> 1;
    const int xo = (int)(p % Wo);
    p /= Wo;
    const int yo = (int)(p % Ho);
    const int b = (int)(p / Ho);
    const T* s = in + (((long)b * (H + 2) + 2 * yo + 1) * (W + 2) + 2 * xo + 1) * C + cc;
    float a[8], t[8];
    load8(s, a);
    load8(s + C, t);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = fmaxf(a[k], t[k]);
    load8(s + (long)(W + 2) * C, t);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = fmaxf(a[k], t[k]);
    load8(s + (long)(W + 2) * C + C, t);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = fmaxf(a[k], t[k]);
    store8(out + (((long)b * (Ho + 2 * op) + yo + op) * (Wo + 2 * op) + xo + op) * C + cc, a);
}

// the same on a PAIR plane (SGG_PAIR16: pixel = [hi (C) | lo (C)], x = hi + lo): the maximum of the four fp32 values, split again (exact:
// the maximum IS one of the four inputs, so its split is that input's)
__global__ __launch_bounds__(256) void maxpool_pair_kernel(const f16_t* __restrict__ in, f16_t* __restrict__ out, int op, int H, int W,
                                                           int C, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c8 = C >> 3;
    const int cc = (int)(i % c8) * 8;
    long p = i / c8;
    const int Wo = W >> 1, Ho = H >> 1;
    const int xo = (int)(p % Wo);
    p /= Wo;
    const int yo = (int)(p % Ho);
    const int b = (int)(p / Ho);
    const f16_t* s = in + (((long)b * (H + 2) + 2 * yo + 1) * (W + 2) + 2 * xo + 1) * 2 * C + cc;
    float a[8], h[8], l[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = -3.0e38f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f16_t* sq = s + ((long)(q >> 1) * (W + 2) + (q & 1)) * 2 * C;
        load8(sq, h);
        load8(sq + C, l);
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = fmaxf(a[k], h[k] + l[k]);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        h[k] = round_as<f16_t>(a[k]);
        l[k] = a[k] - h[k];
    }
    f16_t* o = out + (((long)b * (Ho + 2 * op) + yo + op) * (Wo + 2 * op) + xo + op) * 2 * C + cc;
    store8(o, h);
    store8(o + C, l);
}

// ------------------------------------------------------------------------------------------------
// a-4  RoIAlign, torchvision semantics (aligned=False), union box fused (rel_model_base.py:248-250),
// optional fused broadcast add (lib/get_union_boxes.py:101).  One workgroup per RoI.
// Phase 1: one wave per bin, a lane owns 8 consecutive channels, so every feature-map access is a full 16-byte
//   (bf16) / 32-byte (f32) piece of one pixel's channel vector (1 KiB per wave for C=512); the [P*P][C] result tile
//   stays in LDS.
// Phase 2: the tile is written out transposed, as the reference's [C][P][P] block (what fc6's K axis expects without
//   any weight re-ordering): 25088 contiguous elements per RoI, 16-byte pieces per lane.
// RoIs are visited XCD-contiguously: RoIs are sorted by image, so one image's map stays in one XCD's L2.
// Round 4 also measured a windowed form (one workgroup per (RoI, 128-byte channel slice); the cells a band of bin rows touches copied
// once into LDS by LDS-DMA, taps from LDS: 262 cells per union box on average instead of 784 tap requests): correct, 0.34-0.55 ms against
// 0.177 ms for this kernel -- a workgroup's chain (box loads, window plan, DMA, barrier, taps, barrier, write) is latency, and the LDS
// it needs leaves 12-16 waves per CU to hide it.  tools/roi_bench.py times this kernel at the bench configuration.
// ------------------------------------------------------------------------------------------------
// t = w * tap / t += w * tap on a packed 8-channel piece; the 16-bit float form converts inside the FMA (v_fma_mix_f32)
template <typename T>
__device__ __forceinline__ void tap_mul8(const Raw8<T>& tp, float w, float (&t)[8]) {
    float v[8];
    tp.get(v);
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = w * v[k];
}
template <typename T>
__device__ __forceinline__ void tap_fma8(const Raw8<T>& tp, float w, float (&t)[8]) {
    float v[8];
    tp.get(v);
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = fmaf(w, v[k], t[k]);
}
#define SGG_MIX_LO(d, h, w, c) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(w), "v"(c))
#define SGG_MIX_HI(d, h, w, c) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(w), "v"(c))
template <>
__device__ __forceinline__ void tap_mul8<f16_t>(const Raw8<f16_t>& tp, float w, float (&t)[8]) {
    const float z = 0.f;
    SGG_MIX_LO(t[0], tp.r.x, w, z); SGG_MIX_HI(t[1], tp.r.x, w, z);
    SGG_MIX_LO(t[2], tp.r.y, w, z); SGG_MIX_HI(t[3], tp.r.y, w, z);
    SGG_MIX_LO(t[4], tp.r.z, w, z); SGG_MIX_HI(t[5], tp.r.z, w, z);
    SGG_MIX_LO(t[6], tp.r.w, w, z); SGG_MIX_HI(t[7], tp.r.w, w, z);
}
template <>
__device__ __forceinline__ void tap_fma8<f16_t>(const Raw8<f16_t>& tp, float w, float (&t)[8]) {
    SGG_MIX_LO(t[0], tp.r.x, w, t[0]); SGG_MIX_HI(t[1], tp.r.x, w, t[1]);
    SGG_MIX_LO(t[2], tp.r.y, w, t[2]); SGG_MIX_HI(t[3], tp.r.y, w, t[3]);
    SGG_MIX_LO(t[4], tp.r.z, w, t[4]); SGG_MIX_HI(t[5], tp.r.z, w, t[5]);
    SGG_MIX_LO(t[6], tp.r.w, w, t[6]); SGG_MIX_HI(t[7], tp.r.w, w, t[7]);
}

constexpr int MAXS = 32;  // max P*sampling samples per axis

template <typename T>
__global__ __launch_bounds__(512, 4) void roi_align_kernel(const T* __restrict__ fmap, int B, int H, int W, int C,
                                                        const float* __restrict__ rois, const int64_t* __restrict__ pairs,
                                                        int R, float scale, int P, int S, const float* __restrict__ add_ec,
                                                        T* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    T* tile = reinterpret_cast<T*>(dyn);            // [P*P][CS + 8], CS = channels of this workgroup's slice
    const int nsplit = gridDim.y, CS = C / nsplit, cbase = blockIdx.y * CS;
    const int TS = CS + 8;
    __shared__ int s_lo[2][MAXS], s_hi[2][MAXS];
    __shared__ float s_l[2][MAXS], s_h[2][MAXS];  // s_h < 0 marks an out-of-range sample
    __shared__ int s_b;
    const int r = xcd_remap(blockIdx.x, R);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NWV = blockDim.x >> 6;
    const int PS = P * S;
    if (tid < 2 * PS) {
        const int axis = tid / PS, k = tid - axis * PS;  // axis 0 = y, 1 = x
        float lo_c, hi_c, bi;
        if (pairs) {
            const float* a = rois + pairs[2 * (long)r] * 5;
            const float* b = rois + pairs[2 * (long)r + 1] * 5;
            bi = a[0];
            lo_c = fminf(a[2 - axis], b[2 - axis]);
            hi_c = fmaxf(a[4 - axis], b[4 - axis]);
        } else {
            const float* a = rois + (long)r * 5;
            bi = a[0];
            lo_c = a[2 - axis];
            hi_c = a[4 - axis];
        }
        const int L = axis == 0 ? H : W;
        const float start = lo_c * scale, end = hi_c * scale;
        const float len = fmaxf(end - start, 1.0f);
        const float bin = len / (float)P;
        const int p = k / S, i = k - p * S;
        float c = start + (float)p * bin + ((float)i + 0.5f) * bin / (float)S;
        const bool valid = !(c < -1.0f || c > (float)L);
        c = c <= 0.f ? 0.f : c;
        int lo = (int)c, hi;
        if (lo >= L - 1) {
            hi = lo = L - 1;
            c = (float)lo;
        } else {
            hi = lo + 1;
        }
        const float l = c - (float)lo;
        s_lo[axis][k] = valid ? lo : 0;
        s_hi[axis][k] = valid ? hi : 0;
        s_l[axis][k] = l;
        s_h[axis][k] = valid ? 1.f - l : -1.f;
        if (tid == 0) s_b = min(max((int)bi, 0), B - 1);
    }
    __syncthreads();
    const T* fm = fmap + (long)s_b * H * W * C;
    const float inv = 1.0f / (float)(S * S);
    const int PP = P * P;
    for (int cl = lane * 8; cl < CS; cl += 512) {
        const int c0 = cbase + cl;
        float addv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) addv[k] = add_ec ? add_ec[(long)r * C + c0 + k] : 0.f;
        if (S == 2) {
            // The reference's sampling_ratio.  All 16 taps of a bin are requested before the first is used (16 instead
            // of 4 independent 16-byte loads per lane: the kernel is bound by gather latency, not bytes; 0.48 -> 0.38 ms).
            // Samples outside the map keep weight 0 and read pixel 0.
            // Bilinear interpolation is separable and the bin's four samples share their rows and columns pairwise, so the 16 taps
            // are 4 rows x 4 columns:  out = sum_r wy[r] (sum_c wx[c] F[r][c])  -- 16 + 4 FMAs per channel instead of 16 weight
            // products, 16 conversions and 16 FMAs (the 16-bit taps are converted inside v_fma_mix_f32); round 4: the kernel was
            // co-limited by this VALU work and the L2 -> L1 gathers.  Row r = (sample iy, lo / hi row), column c = (sample ix, lo / hi).
            auto issue = [&](int bin, Raw8<T> (&tap)[16], float (&wy)[4], float (&wx)[4]) {
                const int ph = bin / P, pw = bin - ph * P;
#pragma unroll
                for (int ix = 0; ix < 2; ++ix) {
                    const int kx = pw * 2 + ix;
                    const float hx = s_h[1][kx], lx = s_l[1][kx];
                    wx[2 * ix] = hx >= 0.f ? hx : 0.f;
                    wx[2 * ix + 1] = hx >= 0.f ? lx : 0.f;
                }
#pragma unroll
                for (int iy = 0; iy < 2; ++iy) {
                    const int ky = ph * 2 + iy;
                    const float hy = s_h[0][ky], ly = s_l[0][ky];
                    wy[2 * iy] = hy >= 0.f ? hy : 0.f;
                    wy[2 * iy + 1] = hy >= 0.f ? ly : 0.f;
                    const T* row_lo = fm + (long)s_lo[0][ky] * W * C + c0;
                    const T* row_hi = fm + (long)s_hi[0][ky] * W * C + c0;
#pragma unroll
                    for (int ix = 0; ix < 2; ++ix) {
                        const int kx = pw * 2 + ix;
                        const int xl = s_lo[1][kx] * C, xh = s_hi[1][kx] * C;
                        const int q = (iy * 2 + ix) * 4;
                        tap[q].load(row_lo + xl);
                        tap[q + 1].load(row_lo + xh);
                        tap[q + 2].load(row_hi + xl);
                        tap[q + 3].load(row_hi + xh);
                    }
                }
            };
            auto finish = [&](int bin, const Raw8<T> (&tap)[16], const float (&wy)[4], const float (&wx)[4]) {
                float acc[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll
                for (int iy = 0; iy < 2; ++iy) {
#pragma unroll
                    for (int ry = 0; ry < 2; ++ry) {
                        float t[8];
                        tap_mul8(tap[(iy * 2) * 4 + ry * 2], wx[0], t);
                        tap_fma8(tap[(iy * 2) * 4 + ry * 2 + 1], wx[1], t);
                        tap_fma8(tap[(iy * 2 + 1) * 4 + ry * 2], wx[2], t);
                        tap_fma8(tap[(iy * 2 + 1) * 4 + ry * 2 + 1], wx[3], t);
                        const float w = wy[iy * 2 + ry];
#pragma unroll
                        for (int k = 0; k < 8; ++k) acc[k] = fmaf(w, t[k], acc[k]);
                    }
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] = acc[k] * inv + addv[k];
                store8(tile + bin * TS + cl, acc);
            };
            for (int bin = wave; bin < PP; bin += NWV) {   // (two bins = 32 loads in flight measured slower: 0.46 vs 0.38 ms)
                Raw8<T> tap[16];
                float wy[4], wx[4];
                issue(bin, tap, wy, wx);
                finish(bin, tap, wy, wx);
            }
            continue;
        }
        for (int bin = wave; bin < PP; bin += NWV) {
            const int ph = bin / P, pw = bin - ph * P;
            float acc[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = 0.f;
            for (int iy = 0; iy < S; ++iy) {
                const int ky = ph * S + iy;
                const float hy = s_h[0][ky], ly = s_l[0][ky];
                if (hy < 0.f) continue;
                const T* row_lo = fm + (long)s_lo[0][ky] * W * C + c0;
                const T* row_hi = fm + (long)s_hi[0][ky] * W * C + c0;
                for (int ix = 0; ix < S; ++ix) {
                    const int kx = pw * S + ix;
                    const float hx = s_h[1][kx], lx = s_l[1][kx];
                    if (hx < 0.f) continue;
                    const int xl = s_lo[1][kx] * C, xh = s_hi[1][kx] * C;
                    float v1[8], v2[8], v3[8], v4[8];
                    load8(row_lo + xl, v1);
                    load8(row_lo + xh, v2);
                    load8(row_hi + xl, v3);
                    load8(row_hi + xh, v4);
                    const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[k] += w1 * v1[k] + w2 * v2[k] + w3 * v3[k] + w4 * v4[k];
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = acc[k] * inv + addv[k];
            store8(tile + bin * TS + cl, acc);
        }
    }
    __syncthreads();
    // phase 2: out[r][c][p], 8 consecutive linear elements per thread
    T* o = out + ((long)r * C + cbase) * PP;
    const int total = CS * PP;
    for (int L0 = tid * 8; L0 < total; L0 += (int)blockDim.x * 8) {
        int c = L0 / PP, p = L0 - c * PP;
        if constexpr (sizeof(T) == 2) {
            // raw 16-bit moves: no 16-bit -> f32 -> 16-bit round trip
            const unsigned short* traw = reinterpret_cast<const unsigned short*>(tile);
            unsigned int w[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned int h = (L0 + k < total) ? (unsigned int)traw[p * TS + c] : 0u;
                w[k >> 1] |= h << ((k & 1) * 16);
                if (++p == PP) {
                    p = 0;
                    ++c;
                }
            }
            if (L0 + 8 <= total) {
                *reinterpret_cast<u32x4*>(o + L0) = u32x4{w[0], w[1], w[2], w[3]};
            } else {
                for (int k = 0; L0 + k < total; ++k) reinterpret_cast<unsigned short*>(o)[L0 + k] = (unsigned short)((w[k >> 1] >> ((k & 1) * 16)) & 0xffffu);
            }
        } else {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                v[k] = (L0 + k < total) ? Elem<T>::ld(tile + p * TS + c) : 0.f;
                if (++p == PP) {
                    p = 0;
                    ++c;
                }
            }
            if (L0 + 8 <= total) store8(o + L0, v);
            else
                for (int k = 0; L0 + k < total; ++k) Elem<T>::st(o + L0 + k, v[k]);
        }
    }
}


}  // namespace

namespace {
// ------------------------------------------------------------------------------------------------
// RoIAlign backward into the feature map (the GAN / feature-augmentation callers, where fmap requires grad: main.py:141; SURVEY
// 8b lists it as `sgg_roi_align_bwd`).  Exact adjoint of roi_align_kernel, as a GATHER: a wave owns one feature-map cell (b, y, x) and
// all C channels (a lane = 8 channels); it walks the RoIs in ascending order (64 candidates per ballot: same image, cell inside the rows
// and columns the RoI's samples touch), and for every bin (ph, pw) whose samples reach the cell adds
//     wy(ph) * wx(pw) / S^2 * d_out[r, :, ph, pw],     w(p) = sum over the bin's S samples of ([lo == cell] (1 - l) + [hi == cell] l)
// -- the forward's per-axis sample table (same clamping, same validity rule), separable, so the weights are two short sums.  One
// writer per cell and a fixed order of RoIs and bins: no atomics, bit-reproducible (round 3's form scattered with float atomicAdd).
// d_out is the forward's layout [R, C, P, P] (layout 0) or channels-last [R, P, P, C] (layout 1: 16-byte loads instead of eight 2-byte
// ones at a 98-byte stride; sgg_permute_ncp_to_npc makes it); d_fmap f32 [B,H,W,C] +=.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void roi_axis(const float* __restrict__ rois, const int64_t* __restrict__ pairs, long r, int axis, float scale,
                                         float& start, float& bin, int P) {
    float lo_c, hi_c;
    if (pairs) {
        const float* a = rois + pairs[2 * r] * 5;
        const float* b = rois + pairs[2 * r + 1] * 5;
        lo_c = fminf(a[2 - axis], b[2 - axis]);
        hi_c = fmaxf(a[4 - axis], b[4 - axis]);
    } else {
        const float* a = rois + r * 5;
        lo_c = a[2 - axis];
        hi_c = a[4 - axis];
    }
    start = lo_c * scale;
    const float end = hi_c * scale;
    bin = fmaxf(end - start, 1.0f) / (float)P;
}
// the forward's sample k = p * S + i of one axis: rows lo / hi and the interpolation weight l (valid = false: the sample reads nothing)
__device__ __forceinline__ bool roi_sample(float start, float bin, int p, int i, int S, int L, int& lo, int& hi, float& l) {
    float c = start + (float)p * bin + ((float)i + 0.5f) * bin / (float)S;
    if (c < -1.0f || c > (float)L) return false;
    c = c <= 0.f ? 0.f : c;
    lo = (int)c;
    if (lo >= L - 1) {
        hi = lo = L - 1;
        c = (float)lo;
    } else {
        hi = lo + 1;
    }
    l = c - (float)lo;
    return true;
}
__device__ __forceinline__ float roi_bin_weight(float start, float bin, int p, int S, int L, int cell) {
    float w = 0.f;
    for (int i = 0; i < S; ++i) {
        int lo, hi;
        float l;
        if (!roi_sample(start, bin, p, i, S, L, lo, hi, l)) continue;
        if (lo == cell) w += 1.f - l;
        if (hi == cell) w += l;
    }
    return w;
}

template <typename T, int LAYOUT>
__global__ __launch_bounds__(256) void roi_align_bwd_kernel(const T* __restrict__ d_out, int B, int H, int W, int C,
                                                            const float* __restrict__ rois, const int64_t* __restrict__ pairs, int R,
                                                            float scale, int P, int S, float* __restrict__ d_fmap) {
    const int cell = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (cell >= B * H * W) return;
    const int b = cell / (H * W), yx = cell - b * (H * W), y = yx / W, x = yx - y * W;
    const float inv = 1.0f / (float)(S * S);
    const int PP = P * P;
    for (int cb = 0; cb < C; cb += 512) {
        const int c0 = cb + lane * 8;
        const bool on = c0 < C;
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        for (int r0 = 0; r0 < R; r0 += 64) {
            // candidates: this lane's RoI lies in image b and its first / last sample rows and columns enclose the cell
            const long r = r0 + lane;
            bool touch = false;
            if (r < R) {
                const int bi = min(max((int)rois[(pairs ? pairs[2 * r] : r) * 5], 0), B - 1);
                if (bi == b) {
                    float sy, by, sx, bx;
                    roi_axis(rois, pairs, r, 0, scale, sy, by, P);
                    roi_axis(rois, pairs, r, 1, scale, sx, bx, P);
                    const float y_first = sy + 0.5f * by / (float)S, y_last = sy + (float)(P - 1) * by + ((float)(S - 1) + 0.5f) * by / (float)S;
                    const float x_first = sx + 0.5f * bx / (float)S, x_last = sx + (float)(P - 1) * bx + ((float)(S - 1) + 0.5f) * bx / (float)S;
                    touch = (float)y >= floorf(fmaxf(y_first, 0.f)) - 1.f && (float)y <= floorf(fmaxf(y_last, 0.f)) + 1.f &&
                            (float)x >= floorf(fmaxf(x_first, 0.f)) - 1.f && (float)x <= floorf(fmaxf(x_last, 0.f)) + 1.f;
                }
            }
            unsigned long long m = __ballot(touch);
            while (m) {
                const int j = __builtin_ctzll(m);
                m &= m - 1;
                const long rr = r0 + j;
                float sy, by, sx, bx;
                roi_axis(rois, pairs, rr, 0, scale, sy, by, P);
                roi_axis(rois, pairs, rr, 1, scale, sx, bx, P);
                // only the bins whose sample coordinates can reach this cell (a superset by one bin on each side; bins outside it have weight 0
                // and were skipped one by one before round 6 -- 49 weight evaluations per RoI and cell where 4 - 9 matter: 6.9 -> ms of the GAN
                // iteration's two launches); same bins, same ascending order: the same sum
                int ph0 = max(0, (int)floorf(((float)y - 1.f - sy) / by) - 1), ph1 = min(P - 1, (int)floorf(((float)y + 1.f - sy) / by) + 1);
                int pw0 = max(0, (int)floorf(((float)x - 1.f - sx) / bx) - 1), pw1 = min(P - 1, (int)floorf(((float)x + 1.f - sx) / bx) + 1);
                if (y == 0) ph0 = 0;
                if (y == H - 1) ph1 = P - 1;
                if (x == 0) pw0 = 0;
                if (x == W - 1) pw1 = P - 1;
                for (int ph = ph0; ph <= ph1; ++ph) {
                    const float wy = roi_bin_weight(sy, by, ph, S, H, y);
                    if (wy == 0.f) continue;
                    for (int pw = pw0; pw <= pw1; ++pw) {
                        const float wx = roi_bin_weight(sx, bx, pw, S, W, x);
                        if (wx == 0.f) continue;
                        const float w = wy * wx * inv;
                        if (!on) continue;
                        if constexpr (LAYOUT == 1) {
                            float g[8];
                            load8(d_out + ((long)rr * PP + ph * P + pw) * C + c0, g);
#pragma unroll
                            for (int k = 0; k < 8; ++k) acc[k] += w * g[k];
                        } else {
#pragma unroll
                            for (int k = 0; k < 8; ++k) acc[k] += w * Elem<T>::ld(d_out + ((long)rr * C + c0 + k) * PP + ph * P + pw);
                        }
                    }
                }
            }
        }
        if (on) {
            float* dst = d_fmap + (long)cell * C + c0;
            float old[8];
            load8(dst, old);
#pragma unroll
            for (int k = 0; k < 8; ++k) old[k] += acc[k];
            store8(dst, old);
        }
    }
}
}  // namespace

extern "C" int sgg_roi_align_bwd(const void* d_out, int B, int H, int W, int C, const float* rois, int Nroi, const int64_t* pairs,
                                 int R, float spatial_scale, int P, int sampling, float* d_fmap, int dtype, int layout, void* stream) {
    if (R == 0) return SGG_OK;
    if (!d_out || !rois || !d_fmap || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || R < 0 || P <= 0 || sampling <= 0 ||
        P * sampling > MAXS || Nroi <= 0 || (layout != 0 && layout != 1) || ((uintptr_t)d_fmap & 15) || (layout == 1 && ((uintptr_t)d_out & 15)))
        return SGG_ERR_ARG;
    const dim3 grid((unsigned)(((long)B * H * W + 3) / 4)), blk(256);
    hipStream_t s = (hipStream_t)stream;
    if (layout == 1) {
        SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL((roi_align_bwd_kernel<T, 1>), grid, blk, 0, s, (const T*)d_out, B, H, W, C, rois, pairs, R, spatial_scale, P, sampling, d_fmap));
    } else {
        SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL((roi_align_bwd_kernel<T, 0>), grid, blk, 0, s, (const T*)d_out, B, H, W, C, rois, pairs, R, spatial_scale, P, sampling, d_fmap));
    }
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_image_prep(const float* img, int h, int w, int rh, int rw, float* out, int b, int Hp, int Wp,
                              void* stream) {
    if (!img || !out || h <= 0 || w <= 0 || rh <= 0 || rw <= 0 || rh > Hp || rw > Wp || b < 0) return SGG_ERR_ARG;
    dim3 grid((rw + 255) / 256, rh);
    hipLaunchKernelGGL(image_prep_kernel<SrcF32>, grid, dim3(256), 0, (hipStream_t)stream, SrcF32{img, h, w}, h, w, rh, rw, out, b, Hp, Wp);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_image_prep_u8(const uint8_t* img_hwc, int h0, int w0, int rh, int rw, float* out, int b, int Hp, int Wp,
                                 void* stream) {
    if (!img_hwc || !out || h0 <= 0 || w0 <= 0 || rh <= 0 || rw <= 0 || rh > Hp || rw > Wp || b < 0) return SGG_ERR_ARG;
    const int S = h0 > w0 ? h0 : w0;     // SquarePad: the transform sees an S x S image
    dim3 grid((rw + 255) / 256, rh);
    hipLaunchKernelGGL(image_prep_kernel<SrcU8>, grid, dim3(256), 0, (hipStream_t)stream, SrcU8{img_hwc, h0, w0}, S, S, rh, rw, out, b, Hp, Wp);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_image_prep_batch(const void* const* imgs, const int* h0, const int* w0, const int* rh, const int* rw,
                                    const unsigned char* is_u8, int n, float* out, int Hp, int Wp, void* stream) {
    if (!imgs || !h0 || !w0 || !rh || !rw || !is_u8 || !out || n < 0) return SGG_ERR_ARG;
    for (int b0 = 0; b0 < n; b0 += PREP_MAX) {
        const int m = n - b0 < PREP_MAX ? n - b0 : PREP_MAX;
        PrepTab t{};
        int mh = 0, mw = 0;
        for (int i = 0; i < m; ++i) {
            const int k = b0 + i;
            if (!imgs[k] || h0[k] <= 0 || w0[k] <= 0 || rh[k] <= 0 || rw[k] <= 0 || rh[k] > Hp || rw[k] > Wp) return SGG_ERR_ARG;
            t.img[i] = imgs[k]; t.h0[i] = h0[k]; t.w0[i] = w0[k]; t.rh[i] = rh[k]; t.rw[i] = rw[k]; t.u8[i] = is_u8[k];
            mh = rh[k] > mh ? rh[k] : mh;
            mw = rw[k] > mw ? rw[k] : mw;
        }
        hipLaunchKernelGGL(image_prep_batch_kernel, dim3((mw + 255) / 256, mh, m), dim3(256), 0, (hipStream_t)stream, t, out, b0, Hp, Wp);
        SGG_CHECK_LAUNCH();
    }
    return SGG_OK;
}

extern "C" int sgg_conv1_1(const float* in, const float* w, const float* bias, void* out, int B, int H, int W,
                           int out_dtype, void* stream) {
    if (!in || !w || !bias || !out || B <= 0 || H <= 0 || W <= 0) return SGG_ERR_ARG;
    const long npix = (long)B * H * W;
    const int grid = (int)((npix + 63) / 64);
    if (out_dtype == SGG_BF16 || out_dtype == SGG_F16) {
        const int segs_per_row = (W + 31) / 32, nseg = B * H * segs_per_row;
        const int g2 = min((nseg + 3) / 4, 256 * 8);
        SGG_FOR_DTYPE16(out_dtype, hipLaunchKernelGGL(conv1_1_mfma_kernel<T>, dim3(g2), dim3(256), 0, (hipStream_t)stream, in, w, bias, (T*)out, nseg,
                                                      segs_per_row, H, W));
    }
    else if (out_dtype == SGG_F32)
        hipLaunchKernelGGL(conv1_1_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, w, bias, (float*)out, npix, H, W);
    else if (out_dtype == SGG_PAIR16)      // fp32 arithmetic, the 64-channel map as a pair plane [B, H+2, W+2, 128] (x3 mode)
        hipLaunchKernelGGL((conv1_1_kernel<f16_t, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, in, w, bias, (f16_t*)out, npix, H, W);
    else
        return SGG_ERR_DTYPE;
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_maxpool2x2(const void* in, void* out, int out_pad, int B, int H, int W, int C, int dtype, void* stream) {
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || (C & 7) || (out_pad != 0 && out_pad != 1))
        return SGG_ERR_ARG;
    const long total = (long)B * (H / 2) * (W / 2) * (C / 8);
    const int grid = (int)((total + 255) / 256);
    if (dtype == SGG_PAIR16) {          // C = channels per plane
        hipLaunchKernelGGL(maxpool_pair_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const f16_t*)in, (f16_t*)out, out_pad, H, W, C, total);
        SGG_CHECK_LAUNCH();
        return SGG_OK;
    }
    SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL(maxpool_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)in, (T*)out, out_pad, H, W, C, total));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

// 8 waves per RoI (49 bins over 8 waves): more gathers in flight per CU than 4 waves (0.370 -> 0.342 ms)
static constexpr int roi_threads() { return 512; }

extern "C" int sgg_roi_align_fwd(const void* fmap, int B, int H, int W, int C, const float* rois, int Nroi,
                                 const int64_t* pairs, int R, float spatial_scale, int P, int sampling,
                                 const float* add_ec, void* out, int dtype, void* stream) {
    if (R == 0) return SGG_OK;
    if (!fmap || !rois || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || R < 0 || Nroi <= 0 || P <= 0 ||
        sampling <= 0 || P * sampling > MAXS)
        return SGG_ERR_ARG;
    if (!pairs && R != Nroi) return SGG_ERR_ARG;
    if (!sgg_is_dtype(dtype)) return SGG_ERR_DTYPE;
    const size_t esz = sgg_elem_size(dtype);
    // channel slices per RoI only when the LDS tile would not fit (measured: slicing for occupancy is slower, the
    // kernel is bound by the per-RoI sample setup and gathers, not by resident waves); slices stay multiples of 8
    // channels and (CS*P*P) a multiple of 8 elements so that every store is a full 16-byte piece
    int nsplit = 1;
    while ((size_t)P * P * (C / nsplit + 8) * esz > 104 * 1024 && (C / nsplit) % 16 == 0 && nsplit < 16) nsplit *= 2;
    const size_t smem = (size_t)P * P * (C / nsplit + 8) * esz;
    if (smem > 150 * 1024 || C % (8 * nsplit)) return SGG_ERR_ARG;
    static bool done[3] = {false, false, false};
    SGG_FOR_DTYPE(dtype,
        if (!done[dtype]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(roi_align_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
                return SGG_ERR_LAUNCH;
            done[dtype] = true;
        }
        hipLaunchKernelGGL(roi_align_kernel<T>, dim3(R, nsplit), dim3(roi_threads()), smem, (hipStream_t)stream, (const T*)fmap, B, H, W, C,
                           rois, pairs, R, spatial_scale, P, sampling, add_ec, (T*)out));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}
