// Streaming-pattern probe for the IMP step's data movement (no arithmetic): 256 persistent workgroups of 1024 threads, each walking
// units of 4 chunks x 32 KiB through a 3-slot LDS ring filled by LDS-DMA (read stream) and storing 32 KiB per chunk (write stream).
//   pattern S: a unit is a 128-byte column slice of 1024 rows of 1 KiB (what imp_ctx_mfma_kernel reads / writes)
//   pattern C: a unit is 128 KiB of contiguous memory
// modes: r (read only), w (write only), rw.   hipcc --offload-arch=gfx950 -O3 tools/exp/dma_pattern.hip -o gpurun_out/dma_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16(const void* sbase, unsigned voff, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_base), "v"(voff), "s"(sbase) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_vm_dyn(int n) {
    switch (n) {
#define W(k) case k: wait_vm<k>(); break;
        W(0) W(1) W(2) W(3) W(4) W(5) W(6) W(7) W(8) W(9) W(10) W(11) W(12) W(13) W(14) W(15) W(16)
#undef W
        default: wait_vm<16>();
    }
}
// SLICED: unit u -> graph u / 8, slice u % 8; row r of the unit at g * 1 MiB + r * 1024 + slice * 128
template <bool SLICED, bool RD, bool WR, int DEPTH>
__global__ __launch_bounds__(1024) void probe(const char* __restrict__ src, char* __restrict__ dst, int nunits, unsigned* sink) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const unsigned ring = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
    // chunk c of unit u: 256 rows x 128 B.  DMA instruction i (0..31) of the chunk: 8 rows (lane>>3), 16 B each (lane&7)
    auto off_of = [&](int u, int c, int i) {   // byte offset of this lane's 16 bytes
        if (SLICED) return (unsigned)((u >> 3) * (1u << 20) + (unsigned)((c * 256 + i * 8 + (lane >> 3)) * 1024) + (u & 7) * 128 + (lane & 7) * 16);
        return (unsigned)(u * (128u << 10) + c * 32768 + i * 1024 + lane * 16);
    };
    const int total = ((nunits - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x) * 4;   // chunks of this workgroup
    auto issue = [&](int k) {                  // k-th chunk of this workgroup
        const int u = blockIdx.x + (k >> 2) * gridDim.x, c = k & 3, slot = k % (DEPTH + 1);
        dma16(src, off_of(u, c, wv * 2), __builtin_amdgcn_readfirstlane(ring + slot * 32768 + wv * 2048));
        dma16(src, off_of(u, c, wv * 2 + 1), __builtin_amdgcn_readfirstlane(ring + slot * 32768 + wv * 2048 + 1024));
    };
    unsigned acc = 0;
    if (RD)
        for (int k = 0; k < DEPTH && k < total; ++k) issue(k);
    for (int k = 0; k < total; ++k) {
        const int u = blockIdx.x + (k >> 2) * gridDim.x, c = k & 3;
        if (WR) {                              // 2 stores of 16 B per thread: rows wv*2, wv*2+1 (x8) of the chunk
            u32x4 val = {(unsigned)k, (unsigned)tid, acc, 7u};
            *reinterpret_cast<u32x4*>(dst + off_of(u, c, wv * 2)) = val;
            *reinterpret_cast<u32x4*>(dst + off_of(u, c, wv * 2 + 1)) = val;
        }
        if (RD) {
            // outstanding after this chunk's DMA: later chunks' DMA (2 each) and stores issued since (2 per chunk)
            const int later = min(DEPTH - 1, total - 1 - k);
            wait_vm_dyn(later * 2 + (WR ? later * 2 + 2 : 0));
            __syncthreads();
            if (k + DEPTH < total) issue(k + DEPTH);
            acc += *reinterpret_cast<const unsigned*>(smem + (k % (DEPTH + 1)) * 32768 + tid * 4);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
template <bool S, bool R, bool W, int D>
float run(const char* src, char* dst, int nunits, unsigned* sink, int reps) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<S, R, W, D>), hipFuncAttributeMaxDynamicSharedMemorySize, (D + 1) * 32768);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<S, R, W, D>), dim3(256), dim3(1024), (D + 1) * 32768, 0, src, dst, nunits, sink);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<S, R, W, D>), dim3(256), dim3(1024), (D + 1) * 32768, 0, src, dst, nunits, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1000.f;
}
int main(int argc, char** argv) {
    for (int B : {128, 512}) {
        const int nunits = B * 8;
        const size_t bytes = (size_t)B << 20;
        char *src, *dst;
        unsigned* sink;
        hipMalloc(&src, bytes);
        hipMalloc(&dst, bytes);
        hipMalloc(&sink, 64);
        hipMemset(src, 1, bytes);
        hipMemset(dst, 0, bytes);
#define RUN(S, R, W, D, name) { const float us = run<S, R, W, D>(src, dst, nunits, sink, 20); \
        printf("B=%d %-28s %8.2f us  %7.1f GB/s\n", B, name, us, ((R ? 1 : 0) + (W ? 1 : 0)) * (double)bytes / us / 1e3); }
        RUN(true, true, false, 2, "sliced  read       depth 2")
        RUN(false, true, false, 2, "contig  read       depth 2")
        RUN(true, true, false, 3, "sliced  read       depth 3")
        RUN(false, true, false, 3, "contig  read       depth 3")
        RUN(true, false, true, 2, "sliced  write")
        RUN(false, false, true, 2, "contig  write")
        RUN(true, true, true, 2, "sliced  read+write depth 2")
        RUN(false, true, true, 2, "contig  read+write depth 2")
        RUN(true, true, true, 3, "sliced  read+write depth 3")
        RUN(false, true, true, 3, "contig  read+write depth 3")
        hipFree(src); hipFree(dst); hipFree(sink);
    }
    return 0;
}
