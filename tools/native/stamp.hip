// tools/step_stamps.py: one thread writes the constant-rate clock (s_memrealtime, 100 MHz) to out[0] -- a time stamp at a
// stream position, comparable across streams (torch events give durations between two events, not where a queue sat idle).
#include <hip/hip_runtime.h>
__global__ void stamp_kernel(unsigned long long* out) { out[0] = wall_clock64(); }
extern "C" int stamp_launch(unsigned long long* out, void* stream) {
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, out);
    return (int)hipGetLastError();
}
