// Diagnostic for tools/bperm_race.py (not part of libsgg_hip.so): does a ds_bpermute butterfly (what __shfl_xor compiles to) return the
// right sums while other kernels share the chip?  Every wave sums four lane-dependent values with the xor butterfly `iters` times and
// counts the results that differ from the known totals.  Build: tools/native/build.sh
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void bperm_check_kernel(int iters, int width, const float* __restrict__ seed, unsigned* __restrict__ errors,
                                                          float* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    // (values arrive through memory so that nothing folds at compile time; seed[k] = k + 1)
    float base[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) base[k] = seed[k] * (float)(lane + 1);
    unsigned bad = 0;
    float keep = 0.f;
    for (int it = 0; it < iters; ++it) {
        float p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) p[k] = base[k] + (float)it;
        for (int off = width >> 1; off > 0; off >>= 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) p[k] += __shfl_xor(p[k], off, 64);
        }
        // group of `width` lanes starting at g0: sum (k+1)(l+1) + it = (k+1) (width g0 + width (width+1) / 2) + width it
        const int g0 = lane & ~(width - 1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float want = seed[k] * (float)(width * g0 + width * (width + 1) / 2) + (float)(width * it);
            if (p[k] != want) bad += 1u << (8 * k);
            keep += p[k];
        }
    }
    if (bad) atomicAdd(errors, 1u), atomicOr(errors + 1, bad);
    if (keep == -1.f) sink[0] = keep;
}

extern "C" int bperm_check(int blocks, int iters, int width, const float* seed, unsigned* errors, float* sink, void* stream) {
    hipLaunchKernelGGL(bperm_check_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, width, seed, errors, sink);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
