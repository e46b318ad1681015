// How fast can the chip move a [rows][1 KiB] array when every workgroup owns a W-byte column slice of 992 consecutive rows?
// (the access pattern of the sliced IMP step: W = 64 today).  Plain 16-byte loads / stores, 8 in flight per lane, no LDS, no math.
// build: hipcc --offload-arch=gfx950 -O3 tools/exp/piece_bw.hip -o tools/exp/piece_bw ; run: tools/exp/piece_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int W>   // bytes per piece (64, 128, 256, 1024)
__global__ __launch_bounds__(1024) void slice_copy(const char* __restrict__ src, char* __restrict__ dst, int graphs, int rows_per_graph) {
    constexpr int LP = W / 16, SL = 1024 / W;           // lanes per piece, slices per row
    const int units = graphs * SL;
    const int grp = threadIdx.x / LP, sub = threadIdx.x % LP, G = 1024 / LP;
    for (int u = blockIdx.x; u < units; u += gridDim.x) {
        // consecutive blocks -> consecutive slices of the same graph (same XCD would need b%8 remap; keep simple: u ordered by (graph, slice))
        const int g = u / SL, sl = u % SL;
        const long base = (long)g * rows_per_graph * 1024 + (long)sl * W + sub * 16;
        for (int r0 = grp; r0 < rows_per_graph; r0 += G * 8) {
            u32x4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int r = min(r0 + k * G, rows_per_graph - 1);
                v[k] = *reinterpret_cast<const u32x4*>(src + base + (long)r * 1024);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int r = r0 + k * G;
                if (r < rows_per_graph) *reinterpret_cast<u32x4*>(dst + base + (long)r * 1024) = v[k];
            }
        }
    }
}

template <int W> float run(const char* s, char* d, int graphs, int rows, int grid, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(slice_copy<W>, dim3(grid), dim3(1024), 0, 0, s, d, graphs, rows);
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(slice_copy<W>, dim3(grid), dim3(1024), 0, 0, s, d, graphs, rows);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    const int graphs = 128, rows = 992;
    const size_t bytes = (size_t)graphs * rows * 1024;
    char *s, *d;
    hipMalloc(&s, bytes); hipMalloc(&d, bytes);
    hipMemset(s, 1, bytes); hipMemset(d, 0, bytes);
    for (int grid : {256, 512}) {
        printf("grid %d x 1024 threads, %d graphs x %d rows x 1 KiB (%.0f MB each way)\n", grid, graphs, rows, bytes / 1e6);
        float t;
        t = run<64>(s, d, graphs, rows, grid, 20);   printf("  W=  64: %7.1f us  %6.0f GB/s\n", t * 1e3, 2.0 * bytes / t / 1e6);
        t = run<128>(s, d, graphs, rows, grid, 20);  printf("  W= 128: %7.1f us  %6.0f GB/s\n", t * 1e3, 2.0 * bytes / t / 1e6);
        t = run<256>(s, d, graphs, rows, grid, 20);  printf("  W= 256: %7.1f us  %6.0f GB/s\n", t * 1e3, 2.0 * bytes / t / 1e6);
        t = run<1024>(s, d, graphs, rows, grid, 20); printf("  W=1024: %7.1f us  %6.0f GB/s\n", t * 1e3, 2.0 * bytes / t / 1e6);
    }
    return 0;
}
