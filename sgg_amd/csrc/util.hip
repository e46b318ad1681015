// Load-time utilities: dtype casts and the weight re-layouts ((c,p)->(p,c) K-order for fc6, OIHW->O(HW)I for convs).
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

// out[c][r] = in[r][c] for r < R, c < C (row strides ld_in / ld_out); 32x32 tiles through LDS
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void transpose_kernel(const TI* __restrict__ in, long ld_in, TO* __restrict__ out, long ld_out,
                                                        int R, int C) {
    __shared__ float t[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        t[k][tx] = (r < R && c < C) ? Elem<TI>::ld(in + (long)r * ld_in + c) : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx;
        if (c < C && r < R) Elem<TO>::st(out + (long)c * ld_out + r, t[tx][k]);
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

// out[n][c][p] = in[n][p][c] + add[n][c]  (fp32; un-does the fc6 K re-order on the gradient and adds the folded
// rect-conv term: d W6[n,c,p] = G[n,(p,c)] + Gsum[n,c])
__global__ __launch_bounds__(256) void unpermute_add_kernel(const float* __restrict__ in, long ld_in, const float* __restrict__ add,
                                                            long ld_add, float* __restrict__ out, int C, int Pp) {
    __shared__ float t[32][33];
    const long n = blockIdx.z;
    const int p0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {   // read in[n][p][c]: c fastest
        const int p = p0 + k, c = c0 + tx;
        t[k][tx] = (p < Pp && c < C) ? in[n * ld_in + (long)p * C + c] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {   // write out[n][c][p]: p fastest
        const int c = c0 + k, p = p0 + tx;
        if (c < C && p < Pp) out[(n * C + c) * Pp + p] = t[tx][k] + (add ? add[n * ld_add + c] : 0.f);
    }
}

}  // namespace

extern "C" int sgg_abi_version(void) { return SGG_ABI_VERSION; }
extern "C" const char* sgg_build_info(void) { return "sgg_hip gfx950 (CDNA4) " __DATE__ " " __TIME__; }

extern "C" int sgg_cast(const void* in, void* out, int64_t n, int in_dtype, int out_dtype, void* stream) {
    if (n == 0) return SGG_OK;
    if (!in || !out || n < 0) return SGG_ERR_ARG;
    if ((((uintptr_t)in) | ((uintptr_t)out)) & 15) return SGG_ERR_ARG;
    const dim3 grid((unsigned)((n + 2047) / 2048)), blk(256);
    hipStream_t s = (hipStream_t)stream;
    if (in_dtype == SGG_F32 && out_dtype == SGG_BF16)
        hipLaunchKernelGGL((cast_kernel<float, bf16_t>), grid, blk, 0, s, (const float*)in, (bf16_t*)out, (long)n);
    else if (in_dtype == SGG_BF16 && out_dtype == SGG_F32)
        hipLaunchKernelGGL((cast_kernel<bf16_t, float>), grid, blk, 0, s, (const bf16_t*)in, (float*)out, (long)n);
    else if (in_dtype == SGG_F32 && out_dtype == SGG_F32)
        hipLaunchKernelGGL((cast_kernel<float, float>), grid, blk, 0, s, (const float*)in, (float*)out, (long)n);
    else if (in_dtype == SGG_BF16 && out_dtype == SGG_BF16)
        hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), grid, blk, 0, s, (const bf16_t*)in, (bf16_t*)out, (long)n);
    else
        return SGG_ERR_DTYPE;
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_permute_ncp_to_npc(const void* in, void* out, int Nn, int C, int Pp, int in_dtype, int out_dtype,
                                      void* stream) {
    if (Nn == 0) return SGG_OK;
    if (!in || !out || Nn < 0 || C <= 0 || Pp <= 0 || Nn > 65535) return SGG_ERR_ARG;
    const dim3 grid((Pp + 31) / 32, (C + 31) / 32, Nn), blk(256);
    hipStream_t s = (hipStream_t)stream;
    if (in_dtype == SGG_F32 && out_dtype == SGG_BF16)
        hipLaunchKernelGGL((permute_kernel<float, bf16_t>), grid, blk, 0, s, (const float*)in, (bf16_t*)out, C, Pp);
    else if (in_dtype == SGG_F32 && out_dtype == SGG_F32)
        hipLaunchKernelGGL((permute_kernel<float, float>), grid, blk, 0, s, (const float*)in, (float*)out, C, Pp);
    else if (in_dtype == SGG_BF16 && out_dtype == SGG_BF16)
        hipLaunchKernelGGL((permute_kernel<bf16_t, bf16_t>), grid, blk, 0, s, (const bf16_t*)in, (bf16_t*)out, C, Pp);
    else if (in_dtype == SGG_BF16 && out_dtype == SGG_F32)
        hipLaunchKernelGGL((permute_kernel<bf16_t, float>), grid, blk, 0, s, (const bf16_t*)in, (float*)out, C, Pp);
    else
        return SGG_ERR_DTYPE;
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_transpose(const void* in, int64_t ld_in, void* out, int64_t ld_out, int R, int C, int in_dtype, int out_dtype,
                             void* stream) {
    if (R == 0 || C == 0) return SGG_OK;
    if (!in || !out || R < 0 || C < 0 || ld_in < C || ld_out < R) return SGG_ERR_ARG;
    const dim3 grid((C + 31) / 32, (R + 31) / 32), blk(256);
    hipStream_t s = (hipStream_t)stream;
    if (in_dtype == SGG_BF16 && out_dtype == SGG_BF16)
        hipLaunchKernelGGL((transpose_kernel<bf16_t, bf16_t>), grid, blk, 0, s, (const bf16_t*)in, (long)ld_in, (bf16_t*)out, (long)ld_out, R, C);
    else if (in_dtype == SGG_F32 && out_dtype == SGG_F32)
        hipLaunchKernelGGL((transpose_kernel<float, float>), grid, blk, 0, s, (const float*)in, (long)ld_in, (float*)out, (long)ld_out, R, C);
    else if (in_dtype == SGG_F32 && out_dtype == SGG_BF16)
        hipLaunchKernelGGL((transpose_kernel<float, bf16_t>), grid, blk, 0, s, (const float*)in, (long)ld_in, (bf16_t*)out, (long)ld_out, R, C);
    else if (in_dtype == SGG_BF16 && out_dtype == SGG_F32)
        hipLaunchKernelGGL((transpose_kernel<bf16_t, float>), grid, blk, 0, s, (const bf16_t*)in, (long)ld_in, (float*)out, (long)ld_out, R, C);
    else
        return SGG_ERR_DTYPE;
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_add(void* y, const void* x, int64_t n, int y_dtype, int x_dtype, void* stream) {
    if (n == 0) return SGG_OK;
    if (!y || !x || n < 0 || (n & 7)) return SGG_ERR_ARG;
    const long n8 = n / 8;
    const dim3 grid((unsigned)((n8 + 255) / 256)), blk(256);
    hipStream_t s = (hipStream_t)stream;
    if (y_dtype == SGG_BF16 && x_dtype == SGG_BF16)
        hipLaunchKernelGGL((add_kernel<bf16_t, bf16_t>), grid, blk, 0, s, (bf16_t*)y, (const bf16_t*)x, n8);
    else if (y_dtype == SGG_F32 && x_dtype == SGG_F32)
        hipLaunchKernelGGL((add_kernel<float, float>), grid, blk, 0, s, (float*)y, (const float*)x, n8);
    else if (y_dtype == SGG_BF16 && x_dtype == SGG_F32)
        hipLaunchKernelGGL((add_kernel<bf16_t, float>), grid, blk, 0, s, (bf16_t*)y, (const float*)x, n8);
    else if (y_dtype == SGG_F32 && x_dtype == SGG_BF16)
        hipLaunchKernelGGL((add_kernel<float, bf16_t>), grid, blk, 0, s, (float*)y, (const bf16_t*)x, n8);
    else
        return SGG_ERR_DTYPE;
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_unpermute_add(const float* in, int64_t ld_in, const float* add, int64_t ld_add, float* out, int Nn, int C,
                                 int Pp, void* stream) {
    if (Nn == 0) return SGG_OK;
    if (!in || !out || Nn < 0 || Nn > 65535 || C <= 0 || Pp <= 0 || ld_in < (int64_t)C * Pp || (add && ld_add < C)) return SGG_ERR_ARG;
    const dim3 grid((C + 31) / 32, (Pp + 31) / 32, Nn), blk(256);
    hipLaunchKernelGGL(unpermute_add_kernel, grid, blk, 0, (hipStream_t)stream, in, (long)ld_in, add, (long)ld_add, out, C, Pp);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}
