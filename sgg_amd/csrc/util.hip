// Load-time utilities: dtype casts and the weight re-layouts ((c,p)->(p,c) K-order for fc6, OIHW->O(HW)I for convs).
#include <cstdlib>
#include <cstring>
#include "common.h"

namespace {

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast_kernel(const TI* __restrict__ in, TO* __restrict__ out, long n) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i + 8 <= n) {
        float v[8];
        load8(in + i, v);
        store8(out + i, v);
    } else {
        for (long k = i; k < n; ++k) Elem<TO>::st(out + k, Elem<TI>::ld(in + k));
    }
}

// out[n][p][c] = in[n][c][p]; tile 32(c) x 32(p) through LDS so both sides are coalesced
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void permute_kernel(const TI* __restrict__ in, TO* __restrict__ out, int C, int Pp) {
    __shared__ float t[32][33];
    const long n = blockIdx.z;
    const int c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        t[r][tx] = (c < C && p < Pp) ? Elem<TI>::ld(in + (n * C + c) * Pp + p) : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        if (c < C && p < Pp) Elem<TO>::st(out + (n * Pp + p) * C + c, t[tx][r]);
    }
}

// out[c][r] = in[r][c] (+ add[r][c / group]) for r < R, c < C (row strides ld_in / ld_out).
// 64x64 tiles through LDS: both the read (along c) and the write (along r) are 16-byte (bf16) / 32-byte (f32)
// pieces per lane when the strides allow it; ragged edges fall back to scalars.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void transpose_kernel(const TI* __restrict__ in, long ld_in, TO* __restrict__ out, long ld_out,
                                                        int R, int C, const float* __restrict__ add, long ld_add, int group,
                                                        float* __restrict__ colsum) {
    __shared__ float t[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tid = threadIdx.x;
    const bool vin = ((ld_in & 7) == 0) && ((reinterpret_cast<uintptr_t>(in) & 15) == 0);
    const bool vout = ((ld_out & 7) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    // read: 64 rows x 8 chunks of 8 elements = 512 chunks, 2 per thread
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int id = tid + q * 256, rr = id >> 3, cc = (id & 7) * 8;
        const int r = r0 + rr, c = c0 + cc;
        float v[8];
        if (r < R && c + 8 <= C && vin) {
            load8(in + (long)r * ld_in + c, v);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (r < R && c + k < C) ? Elem<TI>::ld(in + (long)r * ld_in + c + k) : 0.f;
        }
        if (add && r < R) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (c + k < C) v[k] += add[(long)r * ld_add + (c + k) / group];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) t[rr][cc + k] = v[k];
    }
    __syncthreads();
    if (colsum && tid < 64 && c0 + tid < C) {   // bias gradient for free: column sums of the tile (rows >= R are zero) -> this row block's partial
        float s = 0.f;
#pragma unroll 8
        for (int k = 0; k < 64; ++k) s += t[k][tid];
        colsum[(long)blockIdx.y * C + c0 + tid] = s;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int id = tid + q * 256, cc = id >> 3, rr = (id & 7) * 8;
        const int c = c0 + cc, r = r0 + rr;
        if (c >= C) continue;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = t[rr + k][cc];
        if (r + 8 <= R && vout) {
            store8(out + (long)c * ld_out + r, v);
        } else {
            for (int k = 0; k < 8 && r + k < R; ++k) Elem<TO>::st(out + (long)c * ld_out + r + k, v[k]);
        }
    }
}

// 16-bit-output form: 128x128 tiles (eight 16-byte loads in flight per thread, 256-byte row segments), the tile kept
// in LDS as 16-bit PAIRS [row][col/2] with a 65-word row stride: the b32 writes and the column reads (8 row-chunks x 8
// column pairs per wave) are both bank-conflict free, and one ds_read_b32 feeds two output columns.
template <typename TI, typename TO>
__device__ __forceinline__ void transpose16_tile(unsigned int (&t)[128][65], const int bx, const int by, const TI* __restrict__ in, long ld_in,
                                                 TO* __restrict__ out, long ld_out, int R, int C, const float* __restrict__ add,
                                                 long ld_add, int group, float* __restrict__ colsum) {
    const int r0 = by * 128, c0 = bx * 128;
    const int tid = threadIdx.x;
    const bool vin = ((ld_in & 7) == 0) && ((reinterpret_cast<uintptr_t>(in) & 15) == 0);
    const bool vout = ((ld_out & 7) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    float v[8][8], a0v[8], a1v[8];
    // a thread's eight pieces sit in one column block (c depends on tid & 15 only): its group index is computed once -- inside the
    // guarded loop below the compiler may not hoist a division by a run-time value
    const int c_h = c0 + (tid & 15) * 8;
    const int cg_h = (add && group >= 8) ? c_h / group : 0, left_h = group - (c_h - cg_h * group);
    // An interior tile (the common case): its eight 16-byte pieces per thread are requested as they lie -- no conversion, no bounds test, no
    // branch between the loads -- and unpacked afterwards.  (Round 5: in the guarded loop below every load8 converts at once and sits in its
    // own branch region, and the compiler waits for each load before it issues the next: one load in flight per thread instead of eight.)
    const bool whole = vin && r0 + 128 <= R && c0 + 128 <= C && sizeof(TI) == 2;
    if (whole) {
        u32x4 raw[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int id = tid + q * 256, rr = id >> 4, cc = (id & 15) * 8;
            raw[q] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(in) + ((long)(r0 + rr) * ld_in + c0 + cc) * (long)sizeof(TI));
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if constexpr (sizeof(TI) == 2) unpack8<TI>(raw[q], v[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int id = tid + q * 256, rr = id >> 4, cc = (id & 15) * 8;
        const int r = r0 + rr, c = c0 + cc;
        if (whole) {
            // (loaded above)
        } else if (r < R && c + 8 <= C && vin) {
            load8(in + (long)r * ld_in + c, v[q]);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[q][k] = (r < R && c + k < C) ? Elem<TI>::ld(in + (long)r * ld_in + c + k) : 0.f;
        }
        // the addends are requested together with the tile (an 8-element piece spans at most two groups: two loads), not after it
        a0v[q] = a1v[q] = 0.f;
        if (add && group >= 8 && r < R && c < C) {
            a0v[q] = add[(long)r * ld_add + cg_h];
            if (left_h < 8 && c + left_h < C) a1v[q] = add[(long)r * ld_add + cg_h + 1];
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int id = tid + q * 256, rr = id >> 4, cc = (id & 15) * 8;
        const int r = r0 + rr, c = c0 + cc;
        if (add && r < R && c < C) {
            if (group >= 8) {
                const int left = left_h;
                const float a0 = a0v[q], a1 = a1v[q];
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (c + k < C) v[q][k] += (k < left) ? a0 : a1;
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (c + k < C) v[q][k] += add[(long)r * ld_add + (c + k) / group];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) t[rr][(cc >> 1) + k] = H16<TO>::pack(v[q][2 * k], v[q][2 * k + 1]);
    }
    __syncthreads();
    if (colsum && tid < 128 && c0 + tid < C) {   // bias gradient for free (rows >= R are zero) -> this row block's partial
        float s = 0.f;
#pragma unroll 8
        for (int k = 0; k < 128; ++k) s += (tid & 1) ? H16<TO>::hi(t[k][tid >> 1]) : H16<TO>::lo(t[k][tid >> 1]);
        colsum[(long)by * C + c0 + tid] = s;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int id = tid + q * 256, hi = id >> 6;
        const int rc = (id & 7) + 8 * (hi & 1), cp = ((id >> 3) & 7) + 8 * (hi >> 1);
        const int r = r0 + rc * 8, c = c0 + cp * 2;
        if (c >= C || r >= R) continue;
        unsigned int w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) w[k] = t[rc * 8 + k][cp];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (c + h >= C) break;
            const unsigned sel = h ? 0x07060302u : 0x05040100u;   // v_perm_b32: one selector BYTE per output byte
            TO* dst = out + (long)(c + h) * ld_out + r;
            if (r + 8 <= R && vout) {
                u32x4 o;
                o.x = __builtin_amdgcn_perm(w[1], w[0], sel);
                o.y = __builtin_amdgcn_perm(w[3], w[2], sel);
                o.z = __builtin_amdgcn_perm(w[5], w[4], sel);
                o.w = __builtin_amdgcn_perm(w[7], w[6], sel);
                *reinterpret_cast<u32x4*>(dst) = o;
            } else {
                for (int k = 0; k < 8 && r + k < R; ++k) reinterpret_cast<unsigned short*>(dst)[k] = (unsigned short)((w[k] >> (16 * h)) & 0xffffu);
            }
        }
    }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void transpose16_kernel(const TI* __restrict__ in, long ld_in, TO* __restrict__ out,
                                                          long ld_out, int R, int C, const float* __restrict__ add,
                                                          long ld_add, int group, float* __restrict__ colsum) {
    __shared__ unsigned int t[128][65];
    transpose16_tile<TI, TO>(t, blockIdx.x, blockIdx.y, in, ld_in, out, ld_out, R, C, add, ld_add, group, colsum);
}

// Several plain 16-bit transposes in ONE launch (the W^T copies the backward's dX contractions read, rebuilt after every optimiser update:
// twelve launches of 4 - 67 us on the update's stream in rounds 3 - 5, 0.24 ms per step beside the VGG forward; as one launch their
// 140 MB move in ~50 us).  Workgroup -> (tensor, tile) through the table's running tile counts.
constexpr int TMULTI_MAX = 16;
struct TransposeTab {
    const void* in[TMULTI_MAX];
    void* out[TMULTI_MAX];
    long ld_in[TMULTI_MAX], ld_out[TMULTI_MAX];
    int R[TMULTI_MAX], C[TMULTI_MAX], tile0[TMULTI_MAX + 1];
    int n;
};
template <typename T>
__global__ __launch_bounds__(256) void transpose16_multi_kernel(const TransposeTab tab) {
    __shared__ unsigned int t[128][65];
    const int b = blockIdx.x;
    int k = 0;
    while (k + 1 < tab.n && b >= tab.tile0[k + 1]) ++k;               // (uniform)
    const int local = b - tab.tile0[k], tx = (tab.C[k] + 127) / 128;
    transpose16_tile<T, T>(t, local % tx, local / tx, (const T*)tab.in[k], tab.ld_in[k], (T*)tab.out[k], tab.ld_out[k], tab.R[k], tab.C[k], nullptr, 0, 1,
                           nullptr);
}

// out[n][c] = sum_{p < group} in[n][c*group + p]   (W6sum: the K columns that fold `+ conv(rects)` into fc6)
// One workgroup per (row n, 64 channels): the 64*group floats are loaded coalesced into LDS, then 64 threads each sum
// one run (LDS stride = group floats; conflict-free for odd group, e.g. 49).
template <typename TO>
__global__ __launch_bounds__(256) void group_sum_kernel(const float* __restrict__ in, long ld_in, TO* __restrict__ out, long ld_out,
                                                        int Nn, int C, int group) {
    extern __shared__ float gs_lds[];
    const long n = blockIdx.y;
    const int c0 = blockIdx.x * 64;
    const int cnt = min(64, C - c0) * group;
    const float* src = in + n * ld_in + (long)c0 * group;
    if ((cnt & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        // 16-byte non-temporal pieces: the 411 MB of fp32 masters stream through once per step on the second stream, beside the VGG forward
        for (int i = threadIdx.x * 4; i < cnt; i += 1024)
            *reinterpret_cast<f32x4*>(gs_lds + i) = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + i));
    } else {
        for (int i = threadIdx.x; i < cnt; i += 256) gs_lds[i] = src[i];
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < 64 && c0 + t < C) {
        float s = 0.f;
        for (int k = 0; k < group; ++k) s += gs_lds[t * group + k];
        Elem<TO>::st(out + n * ld_out + c0 + t, s);
    }
}

template <typename T, typename TX>
__global__ __launch_bounds__(256) void add_kernel(T* __restrict__ y, const TX* __restrict__ x, long n8) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    float a[8], b[8];
    load8(y + i * 8, a);
    load8(x + i * 8, b);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] += b[k];
    store8(y + i * 8, a);
}

// out[c] (+)= sum_p parts[p][c] in ascending p (fixed order: bit-reproducible)
__global__ __launch_bounds__(256) void reduce_parts_kernel(const float* __restrict__ parts, int nparts, int ncols, float* __restrict__ out,
                                                           int accumulate) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= ncols) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;      // four chains for memory-level parallelism, combined in a fixed order
    int p = 0;
    for (; p + 4 <= nparts; p += 4) {
        s0 += parts[(long)p * ncols + c];
        s1 += parts[(long)(p + 1) * ncols + c];
        s2 += parts[(long)(p + 2) * ncols + c];
        s3 += parts[(long)(p + 3) * ncols + c];
    }
    for (; p < nparts; ++p) s0 += parts[(long)p * ncols + c];
    const float t = (s0 + s1) + (s2 + s3);
    out[c] = accumulate ? out[c] + t : t;
}

}  // namespace

int sgg_reduce_parts(const float* parts, int nparts, int ncols, float* out, int accumulate, hipStream_t s) {
    if (ncols <= 0) return SGG_OK;
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((ncols + 255) / 256), dim3(256), 0, s, parts, nparts, ncols, out, accumulate);
    return hipGetLastError() == hipSuccess ? SGG_OK : SGG_ERR_LAUNCH;
}

namespace {
__global__ __launch_bounds__(256) void fill_u32_kernel(unsigned* __restrict__ p, unsigned v, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}
}  // namespace

int sgg_fill_u32(void* p, unsigned v, size_t n, hipStream_t s) {
    if (n == 0) return SGG_OK;
    hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (unsigned*)p, v, n);
    return hipGetLastError() == hipSuccess ? SGG_OK : SGG_ERR_LAUNCH;
}

extern "C" int sgg_abi_version(void) { return SGG_ABI_VERSION; }
extern "C" const char* sgg_build_info(void) { return "sgg_hip gfx950 (CDNA4) " __DATE__ " " __TIME__; }

namespace {
// Split-operand form of an fp32 matrix for the 16-bit MFMA (the x3 mode): x = hi + lo with hi = f16(x), lo = f16(x - hi)
// (22 significand bits between them).  Row r of x[rows, K] becomes three K_pad-wide segments of f16: activations (mode 0) [hi | hi | lo],
// weights (mode 1) [hi | lo | hi] -- so that ONE f16 contraction over 3 K_pad computes hi.hi + hi.lo + lo.hi in the MFMA's fp32
// accumulator (the lo.lo term, 2^-22 relative, is dropped).  Columns K .. K_pad - 1 of every segment are zero.
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, long ldx, int K, int Kp, f16_t* __restrict__ out, long ldo,
                                                     long rows, int mode) {
    const int c8 = Kp / 8;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * c8) return;
    const long r = i / c8;
    const int c = (int)(i - r * c8) * 8;
    float v[8], hi[8], lo[8];
    if (c + 8 <= K && ((reinterpret_cast<uintptr_t>(x + r * ldx + c) & 15) == 0)) {
        load8(x + r * ldx + c, v);
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = c + k < K ? x[r * ldx + c + k] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        hi[k] = round_as<f16_t>(v[k]);
        lo[k] = v[k] - hi[k];
    }
    f16_t* o = out + r * ldo + c;
    store8(o, hi);
    store8(o + Kp, mode == 0 ? hi : lo);
    store8(o + 2 * Kp, mode == 0 ? lo : hi);
}

// The PAIR form (SGG_PAIR16): row r of x[rows, K] -> [hi (K_pad) | lo (K_pad)], columns K .. K_pad - 1 zero.  What the x3 GEMMs read
// since round 6 (the kernels walk the plane segments themselves: no duplicated hi plane, 4 bytes written per element instead of 6).
__global__ __launch_bounds__(256) void split2_kernel(const float* __restrict__ x, long ldx, int K, int Kp, f16_t* __restrict__ out, long ldo, long rows) {
    const int c8 = Kp / 8;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * c8) return;
    const long r = i / c8;
    const int c = (int)(i - r * c8) * 8;
    float v[8], hi[8], lo[8];
    if (c + 8 <= K && ((reinterpret_cast<uintptr_t>(x + r * ldx + c) & 15) == 0)) {
        load8(x + r * ldx + c, v);
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = c + k < K ? x[r * ldx + c + k] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        hi[k] = round_as<f16_t>(v[k]);
        lo[k] = v[k] - hi[k];
    }
    f16_t* o = out + r * ldo + c;
    store8(o, hi);
    store8(o + Kp, lo);
}
}  // namespace

// out f16 [rows, 2 * K_pad] (row stride ldo >= 2 K_pad, 16-byte aligned rows) = the PAIR form (SGG_PAIR16) of x f32 [rows, K]
extern "C" int sgg_split2(const float* x, int64_t ldx, int64_t rows, int K, int K_pad, void* out, int64_t ldo, void* stream) {
    if (rows == 0) return SGG_OK;
    if (!x || !out || rows < 0 || K <= 0 || K_pad < K || (K_pad & 7) || ldx < K || ldo < 2L * K_pad || (ldo & 7) || ((uintptr_t)out & 15)) return SGG_ERR_ARG;
    const long n = rows * (K_pad / 8);
    hipLaunchKernelGGL(split2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, K, K_pad, (f16_t*)out, (long)ldo, (long)rows);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

// out f16 [rows, 3 * K_pad] (row stride ldo >= 3 K_pad, 16-byte aligned rows) = the split form of x f32 [rows, K] (row stride ldx);
// K_pad % 8 == 0, K_pad >= K.  mode 0: [hi | hi | lo] (the A operand), mode 1: [hi | lo | hi] (the weights).
extern "C" int sgg_split3(const float* x, int64_t ldx, int64_t rows, int K, int K_pad, void* out, int64_t ldo, int mode, void* stream) {
    if (rows == 0) return SGG_OK;
    if (!x || !out || rows < 0 || K <= 0 || K_pad < K || (K_pad & 7) || ldx < K || ldo < 3L * K_pad || (ldo & 7) || (mode != 0 && mode != 1) ||
        ((uintptr_t)out & 15))
        return SGG_ERR_ARG;
    const long n = rows * (K_pad / 8);
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, K, K_pad, (f16_t*)out, (long)ldo,
                       (long)rows, mode);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_cast(const void* in, void* out, int64_t n, int in_dtype, int out_dtype, void* stream) {
    if (n == 0) return SGG_OK;
    if (!in || !out || n < 0) return SGG_ERR_ARG;
    if ((((uintptr_t)in) | ((uintptr_t)out)) & 15) return SGG_ERR_ARG;
    const dim3 grid((unsigned)((n + 2047) / 2048)), blk(256);
    hipStream_t s = (hipStream_t)stream;
    SGG_FOR_DTYPE2(in_dtype, out_dtype, hipLaunchKernelGGL((cast_kernel<TA, TB>), grid, blk, 0, s, (const TA*)in, (TB*)out, (long)n));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_permute_ncp_to_npc(const void* in, void* out, int Nn, int C, int Pp, int in_dtype, int out_dtype,
                                      void* stream) {
    if (Nn == 0) return SGG_OK;
    if (!in || !out || Nn < 0 || C <= 0 || Pp <= 0 || Nn > 65535) return SGG_ERR_ARG;
    const dim3 grid((Pp + 31) / 32, (C + 31) / 32, Nn), blk(256);
    hipStream_t s = (hipStream_t)stream;
    SGG_FOR_DTYPE2(in_dtype, out_dtype, hipLaunchKernelGGL((permute_kernel<TA, TB>), grid, blk, 0, s, (const TA*)in, (TB*)out, C, Pp));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

// colsum (optional): the column sums of `in` (f32[C]); colsum_ws: f32 scratch of >= ceil(R / 64) * C floats for the row blocks' partial sums
extern "C" int sgg_transpose(const void* in, int64_t ld_in, void* out, int64_t ld_out, int R, int C, const float* add,
                             int64_t ld_add, int group, float* colsum, float* colsum_ws, int in_dtype, int out_dtype, void* stream) {
    if (R == 0 || C == 0) return SGG_OK;
    if (!in || !out || R < 0 || C < 0 || ld_in < C || ld_out < R || (add && group <= 0) || (colsum && !colsum_ws)) return SGG_ERR_ARG;
    if (!sgg_is_dtype(in_dtype) || !sgg_is_dtype(out_dtype)) return SGG_ERR_DTYPE;
    hipStream_t s = (hipStream_t)stream;
    if (group <= 0) group = 1;
    float* part = colsum ? colsum_ws : nullptr;
    int nparts;
    if (out_dtype != SGG_F32) {      // 16-bit output: 128 x 128 tiles
        const dim3 grid16((C + 127) / 128, (R + 127) / 128), blk(256);
        nparts = (int)grid16.y;
        if (in_dtype != SGG_F32 && in_dtype != out_dtype) return SGG_ERR_DTYPE;
        SGG_FOR_DTYPE16(out_dtype,
            if (in_dtype == SGG_F32)
                hipLaunchKernelGGL((transpose16_kernel<float, T>), grid16, blk, 0, s, (const float*)in, (long)ld_in, (T*)out, (long)ld_out, R, C, add,
                                   (long)ld_add, group, part);
            else
                hipLaunchKernelGGL((transpose16_kernel<T, T>), grid16, blk, 0, s, (const T*)in, (long)ld_in, (T*)out, (long)ld_out, R, C, add,
                                   (long)ld_add, group, part));
    } else {
        const dim3 grid((C + 63) / 64, (R + 63) / 64), blk(256);
        nparts = (int)grid.y;
        SGG_FOR_DTYPE(in_dtype, hipLaunchKernelGGL((transpose_kernel<T, float>), grid, blk, 0, s, (const T*)in, (long)ld_in, (float*)out, (long)ld_out,
                                                   R, C, add, (long)ld_add, group, part));
    }
    SGG_CHECK_LAUNCH();
    if (colsum) return sgg_reduce_parts(part, nparts, C, colsum, 0, s);
    return SGG_OK;
}

// n <= 16 transposes in one launch: out[i] [C_i, ld_out_i] = in[i] [R_i, C_i]^T (row stride ld_in_i), 16-bit element type `dtype` on both sides;
// columns R_i .. ld_out_i - 1 of the outputs are NOT written (zero padding made once by the caller stays).
extern "C" int sgg_transpose_multi(const void* const* in, const int64_t* ld_in, void* const* out, const int64_t* ld_out, const int* R, const int* C, int n,
                                   int dtype, void* stream) {
    if (n == 0) return SGG_OK;
    if (!in || !ld_in || !out || !ld_out || !R || !C || n < 0 || n > TMULTI_MAX) return SGG_ERR_ARG;
    if (dtype != SGG_BF16 && dtype != SGG_F16) return SGG_ERR_DTYPE;
    TransposeTab tab{};
    tab.n = n;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        if (!in[i] || !out[i] || R[i] <= 0 || C[i] <= 0 || ld_in[i] < C[i] || ld_out[i] < R[i]) return SGG_ERR_ARG;
        tab.in[i] = in[i]; tab.out[i] = out[i]; tab.ld_in[i] = ld_in[i]; tab.ld_out[i] = ld_out[i]; tab.R[i] = R[i]; tab.C[i] = C[i];
        tab.tile0[i] = tiles;
        tiles += ((C[i] + 127) / 128) * ((R[i] + 127) / 128);
    }
    tab.tile0[n] = tiles;
    hipStream_t s = (hipStream_t)stream;
    SGG_FOR_DTYPE16(dtype, hipLaunchKernelGGL(transpose16_multi_kernel<T>, dim3(tiles), dim3(256), 0, s, tab));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_group_sum(const float* in, int64_t ld_in, void* out, int64_t ld_out, int Nn, int C, int group, int out_dtype,
                             void* stream) {
    if (Nn == 0) return SGG_OK;
    if (!in || !out || Nn < 0 || C <= 0 || group <= 0 || ld_in < (int64_t)C * group || ld_out < C) return SGG_ERR_ARG;
    if (Nn > 65535 || group > 256) return SGG_ERR_ARG;
    const dim3 grid((C + 63) / 64, Nn), blk(256);
    const size_t smem = sizeof(float) * 64 * (size_t)group;
    hipStream_t s = (hipStream_t)stream;
    SGG_FOR_DTYPE(out_dtype, hipLaunchKernelGGL(group_sum_kernel<T>, grid, blk, smem, s, in, (long)ld_in, (T*)out, (long)ld_out, Nn, C, group));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_add(void* y, const void* x, int64_t n, int y_dtype, int x_dtype, void* stream) {
    if (n == 0) return SGG_OK;
    if (!y || !x || n < 0 || (n & 7)) return SGG_ERR_ARG;
    const long n8 = n / 8;
    const dim3 grid((unsigned)((n8 + 255) / 256)), blk(256);
    hipStream_t s = (hipStream_t)stream;
    SGG_FOR_DTYPE2(y_dtype, x_dtype, hipLaunchKernelGGL((add_kernel<TA, TB>), grid, blk, 0, s, (TA*)y, (const TB*)x, n8));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}
