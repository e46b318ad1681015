// probe: __builtin_amdgcn_fdot2_f32_bf16 on gfx950 against unpack + fma on random data
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__global__ void k(const unsigned* a, const unsigned* b, float* out, int n) {
    int i = threadIdx.x;
    float d = 0.f, f = 0.f;
    for (int j = 0; j < n; ++j) {
        unsigned x = a[i * n + j], y = b[i * n + j];
        d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, x), __builtin_bit_cast(bf16x2_t, y), d, false);
        f = fmaf(__uint_as_float(x << 16), __uint_as_float(y << 16), f);
        f = fmaf(__uint_as_float(x & 0xffff0000u), __uint_as_float(y & 0xffff0000u), f);
    }
    out[2 * i] = d;
    out[2 * i + 1] = f;
}
static unsigned pk(float lo, float hi) { unsigned l, h; memcpy(&l, &lo, 4); memcpy(&h, &hi, 4); return (l >> 16) | (h & 0xffff0000u); }
int main() {
    const int T = 64, n = 256;
    unsigned *ha = new unsigned[T * n], *hb = new unsigned[T * n], *da, *db; float *dout, ho[2 * T];
    srand(1);
    for (int i = 0; i < T * n; ++i) {
        float s = (i / n) < 32 ? 1.f : 0.02f;      // second half: small magnitudes (products ~1e-4)
        ha[i] = pk(s * (rand() / (float)RAND_MAX - 0.5f), s * (rand() / (float)RAND_MAX - 0.5f));
        hb[i] = pk(s * (rand() / (float)RAND_MAX - 0.5f), s * (rand() / (float)RAND_MAX - 0.5f));
    }
    hipMalloc(&da, T * n * 4); hipMalloc(&db, T * n * 4); hipMalloc(&dout, sizeof(ho));
    hipMemcpy(da, ha, T * n * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb, T * n * 4, hipMemcpyHostToDevice);
    k<<<1, T>>>(da, db, dout, n);
    hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
    for (int i = 0; i < T; i += 8) printf("thread %2d dot2 %.7g  fma %.7g  rel diff %.3g\n", i, ho[2 * i], ho[2 * i + 1], fabs(ho[2 * i] - ho[2 * i + 1]) / fabs(ho[2 * i + 1]));
    return 0;
}
