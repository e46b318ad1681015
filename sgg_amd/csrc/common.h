// Shared device helpers for the sgg_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sgg_hip.h"

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef _Float16 f16_t;         // IEEE half: the second 16-bit storage / MFMA operand format (same matrix-core rate as bf16, 11-bit
                                // significand instead of 8: the mode whose logits meet the parity clause, DESIGN.md "f16")
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// Run BODY once with `T` = the element type `dtype` names (SGG_F32 / SGG_BF16 / SGG_F16); unknown codes return SGG_ERR_DTYPE.
// SGG_FOR_DTYPE16: the two 16-bit formats only.
#define SGG_FOR_DTYPE(dtype, ...)                                         \
    switch (dtype) {                                                      \
        case SGG_BF16: { using T = bf16_t; __VA_ARGS__; } break;          \
        case SGG_F16: { using T = f16_t; __VA_ARGS__; } break;            \
        case SGG_F32: { using T = float; __VA_ARGS__; } break;            \
        default: return SGG_ERR_DTYPE;                                    \
    }
#define SGG_FOR_DTYPE16(dtype, ...)                                       \
    switch (dtype) {                                                      \
        case SGG_BF16: { using T = bf16_t; __VA_ARGS__; } break;          \
        case SGG_F16: { using T = f16_t; __VA_ARGS__; } break;            \
        default: return SGG_ERR_DTYPE;                                    \
    }
// two element types at once (`TA` for dtype_a, `TB` for dtype_b): f32 on either side and at most ONE 16-bit format in a call
#define SGG_PICK2_(H, da, db, ...)                                                      \
    if ((da) == SGG_F32) { using TA = float; using TB = H; __VA_ARGS__; }               \
    else if ((db) == SGG_F32) { using TA = H; using TB = float; __VA_ARGS__; }          \
    else { using TA = H; using TB = H; __VA_ARGS__; }
#define SGG_FOR_DTYPE2(da, db, ...)                                                                         \
    {                                                                                                       \
        const int h__ = (da) != SGG_F32 ? (da) : (db);                                                      \
        if (((da) != SGG_F32 && (da) != h__) || ((db) != SGG_F32 && (db) != h__)) return SGG_ERR_DTYPE;     \
        switch (h__) {                                                                                      \
            case SGG_F32: { using TA = float; using TB = float; __VA_ARGS__; } break;                       \
            case SGG_BF16: { SGG_PICK2_(bf16_t, da, db, __VA_ARGS__) } break;                               \
            case SGG_F16: { SGG_PICK2_(f16_t, da, db, __VA_ARGS__) } break;                                 \
            default: return SGG_ERR_DTYPE;                                                                  \
        }                                                                                                   \
    }
static inline int sgg_elem_size(int dtype) { return dtype == SGG_F32 ? 4 : 2; }
static inline bool sgg_is_dtype(int dtype) { return dtype == SGG_F32 || dtype == SGG_BF16 || dtype == SGG_F16; }

// out[c] (+)= sum_p parts[p][c], p ascending: the second stage of every column reduction (bias gradients, BatchNorm sums, norms).
// First stages write one partial row per workgroup row instead of meeting in float atomics, whose arrival order -- and therefore the
// sum's last bits -- changes from run to run: training is bit-reproducible (util.hip).
int sgg_reduce_parts(const float* parts, int nparts, int ncols, float* out, int accumulate, hipStream_t s);
// p[0 .. n) = v (32-bit words) with a kernel, on `s` (util.hip).  Used instead of hipMemsetAsync wherever a call may be captured into a
// hipGraph: a captured memset node (sgg_eval_tail's, round 5) made the replay fault on ROCm 7.0; a fill kernel replays fine.
int sgg_fill_u32(void* p, unsigned v, size_t n, hipStream_t s);

#define SGG_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return SGG_ERR_LAUNCH;        \
    } while (0)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned int)v) << 16); }

// f32 -> bf16, round-to-nearest-even: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32, two elements per
// instruction); the integer sequence (add 0x7fff + lsb, shift, NaN test) it replaces was ~11 VALU operations per pair and
// showed up in every kernel that stores bf16
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }

// f32 pair -> f16 pair, round-to-nearest-even (v_cvt_pk_f16_f32 on gfx950); one half -> f32 (v_cvt_f32_f16, SDWA for the high half)
__device__ __forceinline__ unsigned int pack_f16x2(float lo, float hi) {
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2_t{lo, hi}, f16x2_t));
}
__device__ __forceinline__ float f16lo_to_f32(unsigned int w) { return (float)__builtin_bit_cast(f16x2_t, w).x; }
__device__ __forceinline__ float f16hi_to_f32(unsigned int w) { return (float)__builtin_bit_cast(f16x2_t, w).y; }

// The two 16-bit formats behind one interface: pack two f32 into a dword / unpack a dword's halves / round an f32 to what a store keeps
template <typename T> struct H16;
template <> struct H16<bf16_t> {
    static __device__ __forceinline__ unsigned int pack(float lo, float hi) { return pack_bf16x2(lo, hi); }
    static __device__ __forceinline__ float lo(unsigned int w) { return __uint_as_float(w << 16); }
    static __device__ __forceinline__ float hi(unsigned int w) { return __uint_as_float(w & 0xffff0000u); }
};
template <> struct H16<f16_t> {
    static __device__ __forceinline__ unsigned int pack(float lo, float hi) { return pack_f16x2(lo, hi); }
    static __device__ __forceinline__ float lo(unsigned int w) { return f16lo_to_f32(w); }
    static __device__ __forceinline__ float hi(unsigned int w) { return f16hi_to_f32(w); }
};
template <typename T> __device__ __forceinline__ void unpack8(const u32x4& a, float (&v)[8]) {
    v[0] = H16<T>::lo(a.x); v[1] = H16<T>::hi(a.x); v[2] = H16<T>::lo(a.y); v[3] = H16<T>::hi(a.y);
    v[4] = H16<T>::lo(a.z); v[5] = H16<T>::hi(a.z); v[6] = H16<T>::lo(a.w); v[7] = H16<T>::hi(a.w);
}
template <typename T> __device__ __forceinline__ u32x4 pack8(const float (&v)[8]) {
    return u32x4{H16<T>::pack(v[0], v[1]), H16<T>::pack(v[2], v[3]), H16<T>::pack(v[4], v[5]), H16<T>::pack(v[6], v[7])};
}
// x as it reads back after a store in T
template <typename T> __device__ __forceinline__ float round_as(float x) {
    if constexpr (sizeof(T) == 2) return H16<T>::lo(H16<T>::pack(x, 0.f));
    else return x;
}

// element type <-> dtype code
template <typename T> struct DtypeOf;
template <> struct DtypeOf<float> { static constexpr int value = SGG_F32; };
template <> struct DtypeOf<bf16_t> { static constexpr int value = SGG_BF16; };
template <> struct DtypeOf<f16_t> { static constexpr int value = SGG_F16; };

// v_mfma_f32_32x32x16 on 8 packed 16-bit elements per lane, in the format DT names (SGG_BF16 / SGG_F16: same rate, same fragment layout)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
template <int DT>
__device__ __forceinline__ f32x16 mfma_32x32x16(const u32x4& a, const u32x4& b, const f32x16& c) {
    static_assert(DT == SGG_BF16 || DT == SGG_F16, "16-bit operand formats");
    if constexpr (DT == SGG_BF16)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};
template <> struct Elem<f16_t> {
    static __device__ __forceinline__ float ld(const f16_t* p) { return (float)*p; }
    static __device__ __forceinline__ void st(f16_t* p, float v) { *p = (f16_t)v; }
};

// 8 consecutive elements <-> 8 floats (16-B loads for bf16, 2x16-B for f32)
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
    const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
    const u32x4 a = *reinterpret_cast<const u32x4*>(p);
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
    v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
    v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}
__device__ __forceinline__ void load8(const f16_t* p, float (&v)[8]) { unpack8<f16_t>(*reinterpret_cast<const u32x4*>(p), v); }
__device__ __forceinline__ void store8(f16_t* p, const float (&v)[8]) { *reinterpret_cast<u32x4*>(p) = pack8<f16_t>(v); }
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
    u32x4 a = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
    *reinterpret_cast<u32x4*>(p) = a;
}

// raw 8-element row pieces kept packed in registers (4 VGPRs for bf16, 8 for f32) until they are used
template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> {
    u32x4 r;
    __device__ __forceinline__ void load(const bf16_t* p) { r = *reinterpret_cast<const u32x4*>(p); }
    __device__ __forceinline__ void zero() { r = u32x4{0, 0, 0, 0}; }
    __device__ __forceinline__ void get(float (&v)[8]) const {
        v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
        v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
        v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u);
        v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
    }
};
template <> struct Raw8<f16_t> {
    u32x4 r;
    __device__ __forceinline__ void load(const f16_t* p) { r = *reinterpret_cast<const u32x4*>(p); }
    __device__ __forceinline__ void zero() { r = u32x4{0, 0, 0, 0}; }
    __device__ __forceinline__ void get(float (&v)[8]) const { unpack8<f16_t>(r, v); }
};
template <> struct Raw8<float> {
    f32x4 a, b;
    __device__ __forceinline__ void load(const float* p) {
        a = *reinterpret_cast<const f32x4*>(p);
        b = *reinterpret_cast<const f32x4*>(p + 4);
    }
    __device__ __forceinline__ void zero() { a = f32x4{0, 0, 0, 0}; b = a; }
    __device__ __forceinline__ void get(float (&v)[8]) const {
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
};

// dot product of two packed 8-element pieces, fp32 accumulate.  bf16 pieces are unpacked and go through FMAs: the
// v_dot2_f32_bf16 form (__builtin_amdgcn_fdot2_f32_bf16, 4 instructions) is correct in isolation (tools/exp/dot2_probe.hip) but
// gave WRONG gate sums inside imp_fused_kernel, where its results feed DPP wave reductions directly (mean |error| 0.27 on
// e_in at unit scale; round 1, ROCm 7.2 -- suspected missing wait states between the dot and the DPP read).  Do not bring it
// back without the unit test `test_imp_sliced_vs_oracle_math`, which compares this kernel in bf16 with the dense formula.
__device__ __forceinline__ float dot8(const Raw8<bf16_t>& w, const Raw8<bf16_t>& x, float acc) {
    float a[8], b[8];
    w.get(a);
    x.get(b);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = fmaf(a[j], b[j], acc);
    return acc;
}
__device__ __forceinline__ float dot8(const Raw8<f16_t>& w, const Raw8<f16_t>& x, float acc) {
    float a[8], b[8];
    w.get(a);
    x.get(b);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = fmaf(a[j], b[j], acc);
    return acc;
}
__device__ __forceinline__ float dot8(const Raw8<float>& w, const Raw8<float>& x, float acc) {
    acc = fmaf(w.a.x, x.a.x, acc); acc = fmaf(w.a.y, x.a.y, acc); acc = fmaf(w.a.z, x.a.z, acc); acc = fmaf(w.a.w, x.a.w, acc);
    acc = fmaf(w.b.x, x.b.x, acc); acc = fmaf(w.b.y, x.b.y, acc); acc = fmaf(w.b.z, x.b.z, acc); acc = fmaf(w.b.w, x.b.w, acc);
    return acc;
}

// 64-lane wave sum on the DPP path (no LDS crossbar): 4 in-row steps, row_bcast15 / row_bcast31 across rows, total in
// lane 63, broadcast through an SGPR (v_readlane) -- the result is wave-uniform.  gfx9-family DPP controls.
__device__ __forceinline__ float wave_sum(float v) {
    int x;
#define SGG_DPP_ADD(ctrl, rmask)                                                                      \
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xF, false);                  \
    v += __int_as_float(x);
    SGG_DPP_ADD(0xB1, 0xF)   // quad_perm [1,0,3,2]
    SGG_DPP_ADD(0x4E, 0xF)   // quad_perm [2,3,0,1]
    SGG_DPP_ADD(0x141, 0xF)  // row_half_mirror
    SGG_DPP_ADD(0x140, 0xF)  // row_mirror : every lane of a 16-lane row holds the row sum
    SGG_DPP_ADD(0x142, 0xA)  // row_bcast15 into rows 1,3
    SGG_DPP_ADD(0x143, 0xC)  // row_bcast31 into rows 2,3 : lane 63 holds the wave sum
#undef SGG_DPP_ADD
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// 1 / (1 + e^-x) on the hardware reciprocal (v_rcp_f32, 1 ulp): `1.0f / y` compiles to the 11-instruction IEEE division sequence,
// which was a tenth of the VALU work of an IMP edge.  Every kernel that makes a gate (forward, recompute in the backward) uses this one.
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// tanh as 1 - 2 / (e^{2x} + 1) on v_exp_f32 / v_rcp_f32: absolute error ~1e-7, exact limits (+-1) for large |x|.  The GRU cells of the
// forward (imp.hip) and of the backward's recomputation (train.hip) use the SAME two functions, so a recomputed gate equals the forward's.
__device__ __forceinline__ float tanh_fast(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// Workgroup barrier in front of an LDS buffer REFILL (LDS-DMA or ds_write by any wave after the barrier): this wave's LDS reads of
// the buffer must have COMPLETED, not merely been issued, when it arrives.  A bare s_barrier does not say that: the compiler sinks
// the `s_waitcnt lgkmcnt` -- and the MFMAs that consume the reads -- below it, and with two workgroups sharing a CU's LDS a queued
// ds_read then loses the race against the refill.  Found in round 2 as run-to-run differences of conv1_2 (bf16, fused pool): a few
// hundred wrong outputs in 24 % of the launches at benchmark size, none in the small-shape unit tests (one workgroup per CU).
__device__ __forceinline__ void lds_reads_done_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// XCD-aware block remap: consecutive logical ids land on the same XCD (dispatch puts block b on XCD b%8).
// Bijective for any grid size (cdna guide T1).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int nx = 8;
    const int q = nwg / nx, r = nwg % nx;
    const int xcd = bid % nx, idx = bid / nx;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
