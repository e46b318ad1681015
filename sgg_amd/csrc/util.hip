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
