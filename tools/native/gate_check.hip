// Diagnostic for tools/gate_race.py (not part of libsgg_hip.so): the f32 gate kernel's dot-product epilogue as it was until round 5
// (weights loaded per lane, 8 FMAs per gate, ds_bpermute butterfly over the 64 lanes of a row), checking itself: the weights are loaded a
// second time (volatile) and compared, and the butterfly's total is compared with a DPP reduction of the same partials.
// Build: tools/native/build.sh
#include <hip/hip_runtime.h>

template <int CTRL, int ROW_MASK = 0xF> __device__ __forceinline__ float dpp_get(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum_last_lane(float x) {
    x += dpp_get<0xB1>(x);
    x += dpp_get<0x4E>(x);
    x += dpp_get<0x141>(x);
    x += dpp_get<0x140>(x);
    x += dpp_get<0x142, 0xA>(x);
    x += dpp_get<0x143, 0xC>(x);
    return x;
}

// counts[0]: lanes whose second weight load differs from the first; [1]: rows whose butterfly total differs from the DPP total by more
// than 1e-3; [2]: rows whose butterfly total differs BETWEEN lanes (lane 0 vs lane 63); [3 + k]: per gate k of [1]
__global__ __launch_bounds__(256) void gate_check_kernel(const float* __restrict__ gi, const float* __restrict__ b_hh, float* __restrict__ h_out,
                                                         long total, int H, const float* __restrict__ dot_w, int dot_ld, float* __restrict__ dots,
                                                         unsigned* __restrict__ counts) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int h8 = H >> 3;
    const long m = i / h8;
    const int c = (int)(i - m * h8) * 8;
    float o[8];
    const float* gim = gi + m * 3 * H + c;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float r = 1.f / (1.f + __expf(-(gim[j] + b_hh[c + j])));
        const float z = 1.f / (1.f + __expf(-(gim[H + j] + b_hh[H + c + j])));
        const float n = tanhf(gim[2 * H + j] + r * b_hh[2 * H + c + j]);
        o[j] = (0.f - n) * z + n;
        h_out[m * H + c + j] = o[j];
    }
    float p[4];
    unsigned wbad = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float4 a = *reinterpret_cast<const float4*>(dot_w + (long)k * dot_ld + c);
        const float4 b = *reinterpret_cast<const float4*>(dot_w + (long)k * dot_ld + c + 4);
        const float w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        p[k] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) p[k] = fmaf(w[j], o[j], p[k]);
        const volatile float* wv = dot_w + (long)k * dot_ld + c;
#pragma unroll
        for (int j = 0; j < 8; ++j) wbad |= (wv[j] != w[j]) ? 1u : 0u;
    }
    if (wbad) atomicAdd(counts, 1u);
    float q[4], t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = wave_sum_last_lane(p[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = p[k];
    for (int off = h8 >> 1; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] += __shfl_xor(t[k], off, 64);
    }
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float t0 = __shfl(t[k], 0, 64);
        if (lane == 63) {
            if (fabsf(t[k] - q[k]) > 1e-3f) {
                atomicAdd(counts + 1, 1u);
                atomicAdd(counts + 3 + k, 1u);
            }
            if (t0 != t[k]) atomicAdd(counts + 2, 1u);
        }
    }
    if (c == 0) *reinterpret_cast<float4*>(dots + m * 4) = make_float4(t[0], t[1], t[2], t[3]);
}

extern "C" int gate_check(const float* gi, const float* b_hh, float* h_out, int M, int H, const float* dot_w, int dot_ld, float* dots,
                          unsigned* counts, void* stream) {
    const long total = (long)M * (H / 8);
    hipLaunchKernelGGL(gate_check_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gi, b_hh, h_out, total, H, dot_w,
                       dot_ld, dots, counts);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
