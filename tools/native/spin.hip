// Diagnostic for tools/pair_probe.py (not part of libsgg_hip.so): workgroups that occupy wave slots (and optionally stream memory) for a
// given time -- what does a co-resident workgroup per CU cost the VGG forward?   Build: tools/native/build.sh
#include <hip/hip_runtime.h>

// every wave sleeps until `ticks` of s_memrealtime (100 MHz) have passed: occupies its wave slots and registers, touches no memory
__global__ __launch_bounds__(256) void spin_kernel(long ticks, float* sink) {
    const unsigned long long t0 = wall_clock64();
    float acc = 0.f;
    for (int it = 0; it < 200000 && (long)(wall_clock64() - t0) < ticks; ++it) {      // (bounded: never more than ~0.1 s)
        __builtin_amdgcn_s_sleep(32);
        acc += 1.f;
    }
    if (acc < 0.f) sink[0] = acc;
}

// the same occupancy, streaming memory: every thread copies float4s from src to dst (grid-stride) `passes` times
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_kernel(const f32x4_t* __restrict__ src, f32x4_t* __restrict__ dst, long n4, int passes) {
    for (int p = 0; p < passes; ++p)
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256 * 4) {
            f32x4_t v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const long j = i + (long)k * gridDim.x * 256;
                if (j < n4) v[k] = __builtin_nontemporal_load(src + j);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const long j = i + (long)k * gridDim.x * 256;
                if (j < n4) __builtin_nontemporal_store(v[k], dst + j);
            }
        }
}

extern "C" int spin(int blocks, long ticks, float* sink, void* stream) {
    hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ticks, sink);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
extern "C" int stream_copy(int blocks, const void* src, void* dst, long n4, int passes, void* stream) {
    hipLaunchKernelGGL(stream_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f32x4_t*)src, (f32x4_t*)dst, n4, passes);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
