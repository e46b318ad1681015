// Probe of ds_read_b64_tr_b16 semantics on gfx950: LDS holds u16 value = its own element index; every lane passes its own
// byte address; print what each lane receives.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned short* out, int mode) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    // mode 0: lane address = 8 bytes * l (consecutive 4-element pieces)
    // mode 1: lane p of each 16-lane group g: row r = p/4 (stride 128 B = 64 elems), piece q = p%4 -> elem (g*16 + 4q) of row r
    int elem;
    if (mode == 0) elem = 4 * l;
    else {
        const int g = l >> 4, p = l & 15, r = p >> 2, q = p & 3;
        elem = r * 64 + g * 16 + 4 * q;
    }
    // workgroup-relative LDS offset: via an address_space(3) pointer (NOT the low bits of the generic pointer)
    unsigned addr = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned short*)lds + elem * 2;
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)(v >> (16 * j));
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d%s", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3], (l % 4 == 3) ? "\n" : "   ");
    }
    return 0;
}
