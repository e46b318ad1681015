// Diagnostic for tools/x3_repro.py (not part of libsgg_hip.so): fills the whole LDS of every CU with a bit pattern, so that a kernel
// which reads LDS it has not written sees a different value from run to run.  Build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void lds_fill_kernel(unsigned pattern, int words, unsigned* sink) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < words; i += 256) lds[i] = pattern;
    __syncthreads();
    unsigned acc = 0;
    for (int i = threadIdx.x; i < words; i += 256 * 61) acc ^= lds[i];
    if (acc != pattern && acc != 0 && sink) sink[0] = acc;      // (keeps the stores alive)
}

extern "C" int lds_poison(unsigned pattern, int blocks, void* stream) {
    const int bytes = 160 * 1024;
    static bool done = false;
    if (!done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(lds_fill_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return -1;
        done = true;
    }
    hipLaunchKernelGGL(lds_fill_kernel, dim3(blocks), dim3(256), bytes, (hipStream_t)stream, pattern, bytes / 4, (unsigned*)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
