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

// ---- a streaming kernel that keeps to a FEW compute units (prototype for the optimiser's update beside the VGG forward): workgroups read
// their hardware ids; per XCD the first `cus_per_xcd` CUs that show up are claimed, at most `workers` workgroups stay on each as persistent
// workers that fetch chunks from a global counter; every other workgroup exits at once.  state: int[4096] zeroed before the launch --
// [0..2047] arrivals per CU key, [2048..4095-64] status per CU key (0 undecided, 1 claimed, 2 rejected), [4032..4039] CUs claimed per XCD,
// [4040] next chunk, [4041] workers started
__device__ __forceinline__ unsigned hw_cu_key(unsigned& xcc) {
    const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID
    xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 7;                // HW_REG_XCC_ID
    const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    return xcc * 256 + se * 32 + sh * 16 + cu;
}

__global__ __launch_bounds__(256) void claim_stream_kernel(const f32x4_t* __restrict__ src, f32x4_t* __restrict__ dst, long n4, int passes,
                                                           int* __restrict__ state, int cus_per_xcd, int workers) {
    __shared__ int role;
    if (threadIdx.x == 0) {
        unsigned xcc;
        const unsigned key = hw_cu_key(xcc);
        int r = 0;
        const int arrival = atomicAdd(&state[key], 1);
        if (arrival == 0) {
            const int idx = atomicAdd(&state[4032 + xcc], 1);
            const int st = idx < cus_per_xcd ? 1 : 2;
            __atomic_store_n(&state[2048 + key], st, __ATOMIC_RELEASE);
            r = st == 1;
        } else if (arrival < workers) {
            int st;
            while ((st = __atomic_load_n(&state[2048 + key], __ATOMIC_ACQUIRE)) == 0) __builtin_amdgcn_s_sleep(1);
            r = st == 1;
        }
        if (r) atomicAdd(&state[4041], 1);
        role = r;
    }
    __syncthreads();
    if (!role) return;
    constexpr long CH = 256 * 16;                                   // float4s per chunk (64 KiB)
    const long chunks = (n4 + CH - 1) / CH * passes;
    __shared__ long cur;
    for (;;) {
        if (threadIdx.x == 0) cur = atomicAdd(&state[4040], 1);
        __syncthreads();
        const long c = cur;
        __syncthreads();
        if (c >= chunks) return;
        const long base = (c % ((n4 + CH - 1) / CH)) * CH;
        f32x4_t v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const long j = base + k * 256 + threadIdx.x;
            if (j < n4) v[k] = __builtin_nontemporal_load(src + j);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const long j = base + k * 256 + threadIdx.x;
            if (j < n4) __builtin_nontemporal_store(v[k], dst + j);
        }
    }
}

extern "C" int claim_stream_copy(int blocks, const void* src, void* dst, long n4, int passes, int* state, int cus_per_xcd, int workers, void* stream) {
    if (hipMemsetAsync(state, 0, 4096 * sizeof(int), (hipStream_t)stream) != hipSuccess) return -2;
    hipLaunchKernelGGL(claim_stream_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f32x4_t*)src, (f32x4_t*)dst, n4, passes, state,
                       cus_per_xcd, workers);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---- the optimiser's update as a streaming prototype: p (f32), m (f32), g (f16) -> p, m, shadow (f16), n elements, 4096 per workgroup
// iteration; `blocks` workgroups stride statically over the chunks (mode 0) or fetch them from a counter (mode 1)
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void sgd_stream_kernel(float* __restrict__ p, float* __restrict__ m, const _Float16* __restrict__ g,
                                                         _Float16* __restrict__ sh, long n, int mode, int* __restrict__ counter) {
    const long chunks = n / 4096;
    __shared__ long cur;
    long c = blockIdx.x;
    for (;;) {
        if (mode == 1) {
            if (threadIdx.x == 0) cur = atomicAdd(counter, 1);
            __syncthreads();
            c = cur;
            __syncthreads();
        }
        if (c >= chunks) return;
        const long base = c * 4096;
        f32x4_t pv[4], mv[4];
        f16x4_t gv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long i = base + (q * 256 + threadIdx.x) * 4;
            gv[q] = __builtin_nontemporal_load(reinterpret_cast<const f16x4_t*>(g + i));
            pv[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p + i));
            mv[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(m + i));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long i = base + (q * 256 + threadIdx.x) * 4;
            f32x4_t nb, np;
            f16x4_t so;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gg = 0.001f * (float)gv[q][e] + 1e-4f * pv[q][e];
                nb[e] = 0.9f * mv[q][e] + gg;
                np[e] = pv[q][e] - 1e-3f * nb[e];
                so[e] = (_Float16)np[e];
            }
            __builtin_nontemporal_store(nb, reinterpret_cast<f32x4_t*>(m + i));
            __builtin_nontemporal_store(np, reinterpret_cast<f32x4_t*>(p + i));
            *reinterpret_cast<f16x4_t*>(sh + i) = so;
        }
        if (mode == 0) c += gridDim.x;
    }
}

extern "C" int sgd_stream(int blocks, float* p, float* m, const void* g, void* sh, long n, int mode, int* counter, void* stream) {
    if (mode == 1 && hipMemsetAsync(counter, 0, sizeof(int), (hipStream_t)stream) != hipSuccess) return -2;
    hipLaunchKernelGGL(sgd_stream_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, m, (const _Float16*)g, (_Float16*)sh, n, mode, counter);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
