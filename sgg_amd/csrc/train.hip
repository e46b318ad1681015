// Training-side kernels of the relation head: Dropout, ReLU/Dropout backward, bias-gradient column sums,
// train-mode BatchNorm of the rect conv (lib/get_union_boxes.py:54,58), GRU-cell backward, and the backward of the
// IMP gather / gate / scatter (sgg_models/rel_model_stanford.py:74-91).  Dense gradient contractions reuse sgg_gemm
// on transposed operands (sgg_permute_ncp_to_npc with N=1 is the transpose).
#include "common.h"

namespace {

constexpr int MAXH = 512;

// counter-based RNG: one 32-bit hash per element index (splitmix/murmur-style finaliser)
__device__ __forceinline__ unsigned int hash_u32(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x = x ^ (x >> 31);
    return (unsigned int)(x >> 32);
}

// nn.Dropout(p) in training mode, in place: x = keep ? x/(1-p) : 0   (rel_model_base.py:110-111 classifier Dropouts)
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(T* __restrict__ x, long n8, float p, unsigned long long seed) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    float v[8];
    load8(x + i * 8, v);
    const float scale = 1.f / (1.f - p);
    const unsigned int thr = (unsigned int)(p * 4294967296.0);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = hash_u32(seed ^ (unsigned long long)(i * 8 + k)) >= thr ? v[k] * scale : 0.f;
    store8(x + i * 8, v);
}

// the same with the seed in device memory: seed = seed_dev[0] * 4 + salt (what the host computes for its by-value form) -- a launch captured
// into a hipGraph draws a new mask on every replay once the host has written the step's seed there
template <typename T>
__global__ __launch_bounds__(256) void dropout_dev_kernel(T* __restrict__ x, long n8, float p, const unsigned long long* __restrict__ seed_dev,
                                                          unsigned long long salt) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const unsigned long long seed = seed_dev[0] * 4ull + salt;
    float v[8];
    load8(x + i * 8, v);
    const float scale = 1.f / (1.f - p);
    const unsigned int thr = (unsigned int)(p * 4294967296.0);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = hash_u32(seed ^ (unsigned long long)(i * 8 + k)) >= thr ? v[k] * scale : 0.f;
    store8(x + i * 8, v);
}

// dx = dy * (y > 0) * scale  : backward of ReLU (scale 1) or ReLU->Dropout (y is the saved post-dropout value,
// scale 1/(1-p): y > 0 iff the unit was kept and its pre-activation was positive)
template <typename TG, typename TY>
__global__ __launch_bounds__(256) void act_bwd_kernel(const TG* __restrict__ dy, const TY* __restrict__ y, TG* __restrict__ dx,
                                                      long n8, float scale) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    float g[8], v[8];
    load8(dy + i * 8, g);
    load8(y + i * 8, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) g[k] = v[k] > 0.f ? g[k] * scale : 0.f;
    store8(dx + i * 8, g);
}

// column sums of x[M,N] (bias gradients): grid (N/64, MSPLIT); each block reduces a row range into ITS row of out[MSPLIT][N]
// (sgg_reduce_parts adds the rows in a fixed order: no float atomics, bit-reproducible)
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int M, int N, int ld, float* __restrict__ out,
                                                     int rows_per_block) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float s = 0.f;
    if (c < N)
        for (int r = r0 + w; r < r1; r += 4) s += Elem<T>::ld(x + (long)r * ld + c);
    red[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && c < N) out[(long)blockIdx.y * N + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// column sums with 16-byte loads: 8 lanes x 8 columns per block column, 32 row lanes (N % 8 == 0, ld % 8 == 0, 16-byte base)
template <typename T>
__global__ __launch_bounds__(256) void colsum8_kernel(const T* __restrict__ x, int M, int N, int ld, float* __restrict__ out,
                                                      int rows_per_block) {
    __shared__ float red[32][65];
    const int cl = (threadIdx.x & 7) * 8, rl = threadIdx.x >> 3;
    const int c = blockIdx.x * 64 + cl;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = 0.f;
    if (c < N) {
#pragma unroll 4
        for (int r = r0 + rl; r < r1; r += 32) {
            float v[8];
            load8(x + (long)r * ld + c, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] += v[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[rl][cl + k] = s[k];
    __syncthreads();
    if (threadIdx.x < 64 && blockIdx.x * 64 + threadIdx.x < N) {
        float a = 0.f;
#pragma unroll 8
        for (int k = 0; k < 32; ++k) a += red[k][threadIdx.x];
        out[(long)blockIdx.y * N + blockIdx.x * 64 + threadIdx.x] = a;
    }
}

// per-channel sums for BatchNorm: out[0][c] = sum x, out[1][c] = sum x*y  (y == x for statistics: sum x^2;
// y = xhat-source for the backward reductions).  Same decomposition as colsum.
template <typename T, typename T2>
__global__ __launch_bounds__(256) void colsum2_kernel(const T* __restrict__ x, const T2* __restrict__ y, int M, int N,
                                                      float* __restrict__ out, int rows_per_block) {
    __shared__ float red[2][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float s = 0.f, q = 0.f;
    if (c < N)
        for (int r = r0 + w; r < r1; r += 4) {
            const float a = Elem<T>::ld(x + (long)r * N + c), b = Elem<T2>::ld(y + (long)r * N + c);
            s += a;
            q += a * b;
        }
    red[0][w][threadIdx.x & 63] = s;
    red[1][w][threadIdx.x & 63] = q;
    __syncthreads();
    if (w == 0 && c < N) {
        const int t = threadIdx.x;
        out[(long)blockIdx.y * 2 * N + c] = red[0][0][t] + red[0][1][t] + red[0][2][t] + red[0][3][t];
        out[(long)blockIdx.y * 2 * N + N + c] = red[1][0][t] + red[1][1][t] + red[1][2][t] + red[1][3][t];
    }
}

// BatchNorm statistics (sum x, sum x^2) with 16-byte loads: 8 lanes x 8 channels per block column, 32 row lanes (C % 8 == 0)
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, int M, int C, float* __restrict__ out,
                                                       int rows_per_block) {
    __shared__ float red[2][32][65];
    const int cl = (threadIdx.x & 7) * 8, rl = threadIdx.x >> 3;
    const int c = blockIdx.x * 64 + cl;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float s[8], q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = q[k] = 0.f;
    if (c < C) {
#pragma unroll 4
        for (int r = r0 + rl; r < r1; r += 32) {
            float v[8];
            load8(x + (long)r * C + c, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                s[k] += v[k];
                q[k] = fmaf(v[k], v[k], q[k]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        red[0][rl][cl + k] = s[k];
        red[1][rl][cl + k] = q[k];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, cc = threadIdx.x & 63;
        if (blockIdx.x * 64 + cc < C) {
            float a = 0.f;
#pragma unroll 8
            for (int k = 0; k < 32; ++k) a += red[which][k][cc];
            out[(long)blockIdx.y * 2 * C + which * C + blockIdx.x * 64 + cc] = a;
        }
    }
}

// BatchNorm apply with given per-channel scale/shift, optionally followed by the max over 4 consecutive rows
// (MaxPool2d(3,2,1) on the 2x2 map), recording the arg-max row for the backward.
template <typename T, bool MAX4>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ sc,
                                                       const float* __restrict__ sh, T* __restrict__ out,
                                                       unsigned char* __restrict__ arg, long rows_out, int C) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;  // over rows_out * C/8
    const int c8 = C >> 3;
    if (i >= rows_out * c8) return;
    const long r = i / c8;
    const int c = (int)(i - r * c8) * 8;
    float s[8], t[8], best[8];
    load8(sc + c, s);
    load8(sh + c, t);
    unsigned char bi[8];
    constexpr int R = MAX4 ? 4 : 1;
#pragma unroll
    for (int q = 0; q < R; ++q) {
        float v[8];
        load8(x + (r * R + q) * C + c, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float y = v[k] * s[k] + t[k];
            if (q == 0 || y > best[k]) {
                best[k] = y;
                bi[k] = (unsigned char)q;
            }
        }
    }
    store8(out + r * C + c, best);
    if (MAX4 && arg) {
#pragma unroll
        for (int k = 0; k < 8; ++k) arg[r * C + c + k] = bi[k];
    }
}

// BatchNorm backward elementwise part (x is the pre-BN activation, post-ReLU; the ReLU backward is folded in):
//   dxhat-form: dx = gamma*invstd * (dy - mean_dy - xhat * mean_dyxhat), then * (x > 0)
// With MAX4 the incoming gradient dy[rows/4, C] is routed to the arg-max row (others get 0) first.
template <typename T, bool MAX4>
__global__ __launch_bounds__(256) void bn_bwd_kernel(const T* __restrict__ dy, const unsigned char* __restrict__ arg,
                                                     const T* __restrict__ x, const float* __restrict__ mean,
                                                     const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                     const float* __restrict__ sums /*[2][C]: sum dy, sum dy*xhat*/,
                                                     T* __restrict__ dx, long rows, int C, float inv_count,
                                                     const float* __restrict__ count_dev) {
    if (count_dev) inv_count = 1.0f / *count_dev;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;  // over rows * C/8
    const int c8 = C >> 3;
    if (i >= rows * c8) return;
    const long r = i / c8;
    const int c = (int)(i - r * c8) * 8;
    float g[8], v[8], mu[8], is[8], ga[8], s1[8], s2[8];
    load8(x + r * C + c, v);
    load8(mean + c, mu);
    load8(invstd + c, is);
    load8(gamma + c, ga);
    load8(sums + c, s1);
    load8(sums + C + c, s2);
    if (MAX4) {
        float t[8];
        load8(dy + (r >> 2) * C + c, t);
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] = arg[(r >> 2) * C + c + k] == (unsigned char)(r & 3) ? t[k] : 0.f;
    } else {
        load8(dy + r * C + c, g);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float xh = (v[k] - mu[k]) * is[k];
        const float d = ga[k] * is[k] * (g[k] - s1[k] * inv_count - xh * s2[k] * inv_count);
        g[k] = v[k] > 0.f ? d : 0.f;  // ReLU in front of the BN (v = relu(conv) so v > 0 iff conv > 0)
    }
    store8(dx + r * C + c, g);
}

// reductions for the BN backward: sums[0][c] = sum_r g, sums[1][c] = sum_r g * xhat, g routed through MAX4 if needed
template <typename T, bool MAX4>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dy, const unsigned char* __restrict__ arg,
                                                            const T* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, float* __restrict__ sums,
                                                            int rows, int C, int rows_per_block) {
    // 8 lanes x 8 channels cover the block's 64 channels with 16-byte loads; 32 row-lanes walk the rows (C % 8 == 0)
    __shared__ float red[2][32][65];
    const int cl = (threadIdx.x & 7) * 8, rl = threadIdx.x >> 3;
    const int c = blockIdx.x * 64 + cl;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float s[8], q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = q[k] = 0.f;
    if (c < C) {
        float mu[8], is[8];
        load8(mean + c, mu);
        load8(invstd + c, is);
#pragma unroll 2
        for (int r = r0 + rl; r < r1; r += 32) {
            float g[8], xv[8];
            load8(x + (long)r * C + c, xv);
            if (MAX4) {
                load8(dy + (long)(r >> 2) * C + c, g);
                const uint2 a = *reinterpret_cast<const uint2*>(arg + (long)(r >> 2) * C + c);
                const unsigned want = (unsigned)(r & 3);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const unsigned av = ((k < 4 ? a.x : a.y) >> (8 * (k & 3))) & 0xffu;
                    if (av != want) g[k] = 0.f;
                }
            } else {
                load8(dy + (long)r * C + c, g);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                s[k] += g[k];
                q[k] = fmaf(g[k], (xv[k] - mu[k]) * is[k], q[k]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        red[0][rl][cl + k] = s[k];
        red[1][rl][cl + k] = q[k];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, cc = threadIdx.x & 63;
        if (blockIdx.x * 64 + cc < C) {
            float a = 0.f;
#pragma unroll 8
            for (int k = 0; k < 32; ++k) a += red[which][k][cc];
            sums[(long)blockIdx.y * 2 * C + which * C + blockIdx.x * 64 + cc] = a;
        }
    }
}

// GRU cell backward (pointwise): recomputes r,z,n from the saved pre-activations.
//   d_gi = [d_rpre, d_zpre, d_npre], d_gh = [d_rpre, d_zpre, d_npre * r], dh_prev = dh * z
template <typename T>
__global__ __launch_bounds__(256) void gru_gate_bwd_kernel(const T* __restrict__ dh, const float* __restrict__ gi,
                                                           const float* __restrict__ gh, const float* __restrict__ b_hh,
                                                           const T* __restrict__ h_prev, T* __restrict__ d_gi,
                                                           T* __restrict__ d_gh, T* __restrict__ dh_prev, long total, int H) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;  // over M*H/8
    if (i >= total) return;
    const int h8 = H >> 3;
    const long m = i / h8;
    const int c = (int)(i - m * h8) * 8;
    float ir[8], iz[8], in_[8], hr[8], hz[8], hn[8], hp[8], g[8];
    const float* gim = gi + m * 3 * H + c;
    load8(gim, ir);
    load8(gim + H, iz);
    load8(gim + 2 * H, in_);
    if (gh) {
        const float* ghm = gh + m * 3 * H + c;
        load8(ghm, hr);
        load8(ghm + H, hz);
        load8(ghm + 2 * H, hn);
        load8(h_prev + m * H + c, hp);
    } else {
        load8(b_hh + c, hr);
        load8(b_hh + H + c, hz);
        load8(b_hh + 2 * H + c, hn);
#pragma unroll
        for (int j = 0; j < 8; ++j) hp[j] = 0.f;
    }
    load8(dh + m * H + c, g);
    float o_r[8], o_z[8], o_n[8], o_hn[8], o_h[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float r = sigmoidf_(ir[j] + hr[j]);
        const float z = sigmoidf_(iz[j] + hz[j]);
        const float n = tanh_fast(in_[j] + r * hn[j]);
        const float dn = g[j] * (1.f - z), dz = g[j] * (hp[j] - n);
        const float dnpre = dn * (1.f - n * n);
        o_n[j] = dnpre;
        o_hn[j] = dnpre * r;
        o_r[j] = dnpre * hn[j] * r * (1.f - r);
        o_z[j] = dz * z * (1.f - z);
        o_h[j] = g[j] * z;
    }
    T* a = d_gi + m * 3 * H + c;
    store8(a, o_r);
    store8(a + H, o_z);
    store8(a + 2 * H, o_n);
    T* b = d_gh + m * 3 * H + c;
    store8(b, o_r);
    store8(b + H, o_z);
    store8(b + 2 * H, o_hn);
    if (dh_prev) store8(dh_prev + m * H + c, o_h);
}

// Backward of gru_gate_proj_kernel (imp.hip): the edge GRU of a message-passing iteration whose input pre-activations are
//   gi[e] = g_sub(e) P[s] + g_obj(e) P[o] + b_ih.  Recomputes gi and the cell from the saved gh, P and dot products, then
//   d_gi = [d_rpre, d_zpre, d_npre], d_gh = [d_rpre, d_zpre, d_npre * r], dh_prev = dh * z,
//   dq[e] = (<d_gi, P[s]>, <d_gi, P[o]>)  -- the gradients of the two scalar gates (reduced over the row's H/8 lanes).
template <typename T, typename TG>
__global__ __launch_bounds__(256) void gru_gate_proj_bwd_kernel(const T* __restrict__ dh, const TG* __restrict__ gh,
                                                                const float* __restrict__ P, const float* __restrict__ b_ih,
                                                                const int* __restrict__ so, const float* __restrict__ ndots,
                                                                const float* __restrict__ edots, const float* __restrict__ gb,
                                                                const T* __restrict__ h_prev, T* __restrict__ d_gi, T* __restrict__ d_gh,
                                                                T* __restrict__ dh_prev, float* __restrict__ dq, long total, int H) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;  // over M*H/8
    if (i >= total) return;
    const int h8 = H >> 3;
    const long m = i / h8;
    const int c = (int)(i - m * h8) * 8;
    const int s = so[2 * m], ob = so[2 * m + 1];
    const float g_sub = sigmoidf_(ndots[4L * s] + edots[4 * m] + gb[0]);
    const float g_obj = sigmoidf_(ndots[4L * ob + 1] + edots[4 * m + 1] + gb[1]);
    float hr[8], hz[8], hn[8], hp[8], g[8], ps[3][8], po[3][8], gi3[3][8];
    const TG* ghm = gh + m * 3 * H + c;
    load8(ghm, hr);
    load8(ghm + H, hz);
    load8(ghm + 2 * H, hn);
    load8(h_prev + m * H + c, hp);
    load8(dh + m * H + c, g);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        float bi[8];
        load8(P + (long)s * 3 * H + q * H + c, ps[q]);
        load8(P + (long)ob * 3 * H + q * H + c, po[q]);
        load8(b_ih + q * H + c, bi);
#pragma unroll
        for (int j = 0; j < 8; ++j) gi3[q][j] = fmaf(g_sub, ps[q][j], fmaf(g_obj, po[q][j], bi[j]));
    }
    float o_r[8], o_z[8], o_n[8], o_hn[8], o_h[8];
    float q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float r = sigmoidf_(gi3[0][j] + hr[j]);
        const float z = sigmoidf_(gi3[1][j] + hz[j]);
        const float n = tanh_fast(gi3[2][j] + r * hn[j]);
        const float dn = g[j] * (1.f - z), dz = g[j] * (hp[j] - n);
        const float dnpre = dn * (1.f - n * n);
        o_n[j] = dnpre;
        o_hn[j] = dnpre * r;
        o_r[j] = dnpre * hn[j] * r * (1.f - r);
        o_z[j] = dz * z * (1.f - z);
        o_h[j] = g[j] * z;
        q0 = fmaf(o_r[j], ps[0][j], fmaf(o_z[j], ps[1][j], fmaf(o_n[j], ps[2][j], q0)));
        q1 = fmaf(o_r[j], po[0][j], fmaf(o_z[j], po[1][j], fmaf(o_n[j], po[2][j], q1)));
    }
    T* a = d_gi + m * 3 * H + c;
    store8(a, o_r);
    store8(a + H, o_z);
    store8(a + 2 * H, o_n);
    T* b = d_gh + m * 3 * H + c;
    store8(b, o_r);
    store8(b + H, o_z);
    store8(b + 2 * H, o_hn);
    if (dh_prev) store8(dh_prev + m * H + c, o_h);
    for (int off = h8 >> 1; off > 0; off >>= 1) {      // rows never straddle a wave (H/8 a power of two <= 64): fixed order
        q0 += __shfl_xor(q0, off, 64);
        q1 += __shfl_xor(q1, off, 64);
    }
    if (c == 0) {
        dq[2 * m] = q0;
        dq[2 * m + 1] = q1;
    }
}

// IMP edge-side backward, one wave per edge e=(s,o) (rel_model_stanford.py:76-91 in reverse).  Inputs: e_i (saved), the saved dot
// products, d_ctx rows of s and o, dq (gru_gate_proj_bwd_kernel).  Recomputes the four gates, then
//   d_e[e]    (+)= g_out*d_ctx[s] + g_in*d_ctx[o] + sum_k da_k * w_k[H:]     (accumulated into d_e, which already holds the GRU's part)
//   da[e,0..3] = d g_k * g_k(1-g_k)   with  d g_sub = dq[e,0], d g_obj = dq[e,1], d g_out = <d_ctx[s],e>, d g_in = <d_ctx[o],e>
template <typename T>
__global__ __launch_bounds__(256) void edge_ctx_bwd_kernel(const T* __restrict__ e, const int* __restrict__ so, int E, int H,
                                                           const float* __restrict__ ndots, const float* __restrict__ edots,
                                                           const float* __restrict__ gw, const float* __restrict__ gb,
                                                           const float* __restrict__ dq, const T* __restrict__ d_ctx,
                                                           T* __restrict__ d_e, float* __restrict__ da) {
    const int ed = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (ed >= E) return;
    const long s = so[2L * ed], o = so[2L * ed + 1];
    const int c0 = lane * 8;
    const bool act = c0 < H;
    float ee[8], dcs[8], dco[8];
    float q2 = 0.f, q3 = 0.f;
    if (act) {
        load8(e + (long)ed * H + c0, ee);
        load8(d_ctx + s * H + c0, dcs);
        load8(d_ctx + o * H + c0, dco);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            q2 = fmaf(dcs[j], ee[j], q2);
            q3 = fmaf(dco[j], ee[j], q3);
        }
    }
    q2 = wave_sum(q2);
    q3 = wave_sum(q3);
    const f32x4 de = *reinterpret_cast<const f32x4*>(edots + 4L * ed);
    const f32x4 ns = *reinterpret_cast<const f32x4*>(ndots + 4 * s), no = *reinterpret_cast<const f32x4*>(ndots + 4 * o);
    float gk[4], dak[4];
    gk[0] = sigmoidf_(ns.x + de.x + gb[0]);
    gk[1] = sigmoidf_(no.y + de.y + gb[1]);
    gk[2] = sigmoidf_(ns.z + de.z + gb[2]);
    gk[3] = sigmoidf_(no.w + de.w + gb[3]);
    dak[0] = dq[2L * ed] * gk[0] * (1.f - gk[0]);
    dak[1] = dq[2L * ed + 1] * gk[1] * (1.f - gk[1]);
    dak[2] = q2 * gk[2] * (1.f - gk[2]);
    dak[3] = q3 * gk[3] * (1.f - gk[3]);
    if (act) {
        float r[8];
        load8(d_e + (long)ed * H + c0, r);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float w[8];
            load8(gw + (long)k * 2 * H + H + c0, w);
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = fmaf(dak[k], w[j], r[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += gk[2] * dcs[j] + gk[3] * dco[j];
        store8(d_e + (long)ed * H + c0, r);
    }
    if (lane < 4) da[(long)ed * 4 + lane] = dak[lane];
}

// IMP node-side backward of the GATES, one workgroup per node n (the gate-weighted sums of d_gi are imp_ctx with pair 0):
//   d_v[n] += S_sub*w_sub[:H] + S_out*w_out[:H] + S_obj*w_obj[:H] + S_in*w_in[:H]
//   with S_sub/S_out = sums of da[.,0]/da[.,2] over out(n), S_obj/S_in = sums of da[.,1]/da[.,3] over in(n) (ascending list order);
//   nsum[n,0..3] = (S_sub, S_obj, S_out, S_in) is also written: the gate weights' vertex-half gradient is nsum^T . v.
template <typename T>
__global__ __launch_bounds__(64) void node_gates_bwd_kernel(const float* __restrict__ da, const int* __restrict__ out_ptr,
                                                            const int* __restrict__ out_ids, const int* __restrict__ in_ptr,
                                                            const int* __restrict__ in_ids, const float* __restrict__ gw, int H,
                                                            T* __restrict__ d_v, float* __restrict__ nsum) {
    const int n = blockIdx.x, lane = threadIdx.x;
    float S[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const int* ptr = side ? in_ptr : out_ptr;
        const int* ids = side ? in_ids : out_ids;
        const int beg = ptr[n], end = ptr[n + 1];
        for (int k = beg + lane; k < end; k += 64) {
            const int id = ids[k];
            S[side] += da[(long)id * 4 + side];
            S[2 + side] += da[(long)id * 4 + 2 + side];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) S[k] = wave_sum(S[k]);
    if (lane < 4) nsum[(long)n * 4 + lane] = S[lane];
    for (int c = lane * 8; c < H; c += 512) {
        float t[8];
        load8(d_v + (long)n * H + c, t);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float w[8];
            load8(gw + (long)k * 2 * H + c, w);
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = fmaf(S[k], w[j], t[j]);
        }
        store8(d_v + (long)n * H + c, t);
    }
}

// small dense reduction  out[k][h] += sum_r a[r][k] * x[r][h]  for k < 4 (gate-weight gradients):
// grid (H/64, RSPLIT); a is f32 [R,4], x is T [R,H]; every row block writes its [4][H] partial, rank4_finalize_kernel adds them
template <typename T>
__global__ __launch_bounds__(256) void rank4_reduce_kernel(const float* __restrict__ a, const T* __restrict__ x, int R, int H,
                                                           float* __restrict__ out, int rows_per_block) {
    // 8 lanes x 8 channels (16-byte loads when H % 8 == 0) cover the block's 64 channels, 32 row lanes walk the rows
    __shared__ float red[32][4][65];
    const int cl = (threadIdx.x & 7) * 8, rl = threadIdx.x >> 3;
    const int c = blockIdx.x * 64 + cl;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(R, r0 + rows_per_block);
    float s[4][8];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) s[k][j] = 0.f;
    const bool vec = (H & 7) == 0 && c + 8 <= H;
#pragma unroll 2
    for (int r = r0 + rl; r < r1; r += 32) {
        float xv[8];
        if (vec) {
            load8(x + (long)r * H + c, xv);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[j] = c + j < H ? Elem<T>::ld(x + (long)r * H + c + j) : 0.f;
        }
        const f32x4 av = *reinterpret_cast<const f32x4*>(a + (long)r * 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s[0][j] = fmaf(av.x, xv[j], s[0][j]);
            s[1][j] = fmaf(av.y, xv[j], s[1][j]);
            s[2][j] = fmaf(av.z, xv[j], s[2][j]);
            s[3][j] = fmaf(av.w, xv[j], s[3][j]);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) red[rl][k][cl + j] = s[k][j];
    __syncthreads();
    const int k = threadIdx.x >> 6, cc = threadIdx.x & 63;
    if (blockIdx.x * 64 + cc < H) {
        float t = 0.f;
#pragma unroll 8
        for (int q = 0; q < 32; ++q) t += red[q][k][cc];
        out[((long)blockIdx.y * 4 + k) * H + blockIdx.x * 64 + cc] = t;       // parts [split][4][H]
    }
}

// from (sum x, sum x^2): batch mean / biased var -> invstd, the affine (scale, shift) used by bn_apply, and the
// running-statistics update of nn.BatchNorm2d (momentum m, unbiased variance) -- lib/get_union_boxes.py:54,58
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ sums, int C, float count,
                                                          const float* __restrict__ count_dev,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, float momentum, float* __restrict__ run_mean,
                                                          float* __restrict__ run_var, float* __restrict__ mean,
                                                          float* __restrict__ invstd, float* __restrict__ scale,
                                                          float* __restrict__ shift) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    if (count_dev) count = *count_dev;     // synchronised BN: the all-reduced row count lives next to the all-reduced sums
    const float mu = sums[c] / count;
    const float var = fmaxf(sums[C + c] / count - mu * mu, 0.f);
    const float is = rsqrtf(var + eps);
    mean[c] = mu;
    invstd[c] = is;
    scale[c] = gamma[c] * is;
    shift[c] = beta[c] - mu * gamma[c] * is;
    if (run_mean) {
        run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mu;
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * var * (count > 1.f ? count / (count - 1.f) : 1.f);
    }
}

// ---- optimiser step (main.py:119-120: grad_clip then SGD with momentum, lib/pytorch_misc.py:144,625-656)
// sum of squares of g: one partial per block (part[blockIdx.x]; reduce_scalar_kernel adds them in a fixed order); 4 independent
// 8-element pieces in flight per thread
template <typename T>
__global__ __launch_bounds__(256) void sqnorm_kernel(const T* __restrict__ g, long n, float* __restrict__ part) {
    __shared__ float red[4];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const long stride = (long)gridDim.x * 256 * 8;
    long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
    for (; i + 3 * stride + 8 <= n; i += 4 * stride) {
        float a[8], b[8], c[8], d[8];
        load8(g + i, a);
        load8(g + i + stride, b);
        load8(g + i + 2 * stride, c);
        load8(g + i + 3 * stride, d);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            s0 = fmaf(a[k], a[k], s0);
            s1 = fmaf(b[k], b[k], s1);
            s2 = fmaf(c[k], c[k], s2);
            s3 = fmaf(d[k], d[k], s3);
        }
    }
    for (; i < n; i += stride) {
        if (i + 8 <= n) {
            float v[8];
            load8(g + i, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) s0 = fmaf(v[k], v[k], s0);
        } else {
            for (long k = i; k < n; ++k) {
                const float v = Elem<T>::ld(g + k);
                s0 = fmaf(v, v, s0);
            }
        }
    }
    float s = wave_sum((s0 + s1) + (s2 + s3));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// torch.optim.SGD(momentum, weight_decay, dampening 0) with the global-norm clip folded in:
//   coef = min(1, max_norm / (sqrt(*norm_sq) + 1e-6))  (1 if norm_sq == NULL);  g' = coef*grad_scale*g + wd*p;
//   buf = first ? g' : mom*buf + g';  p -= lr*buf
template <typename TG>
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const TG* __restrict__ g, float* __restrict__ buf,
                                                  long n, float lr, float wd, float mom, int first,
                                                  const float* __restrict__ norm_sq, float max_norm, float grad_scale) {
    float coef = grad_scale;
    if (norm_sq) {
        if (!(*norm_sq < 3.0e38f)) return;      // non-finite gradients (an overflowed scaled gradient of the f16 mode, a NaN): skip the step
        const float c = max_norm / (sqrtf(*norm_sq) * grad_scale + 1e-6f);
        if (c < 1.f) coef *= c;
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float pv = p[i];
        const float gv = coef * Elem<TG>::ld(g + i) + wd * pv;
        const float b = first ? gv : mom * buf[i] + gv;
        buf[i] = b;
        p[i] = pv - lr * b;
    }
}

// ---- multi-tensor forms: ONE launch walks every parameter (36 tensors from 51 floats to 103 M floats), so the
// optimiser costs two kernels per step instead of 72 and the small tensors stop paying a launch each.
constexpr int MT_MAX = 32;          // tensors per launch (table travels by value in the kernarg segment)
constexpr int MT_CHUNK = 4096;      // elements per block-iteration: 256 threads x 16

struct MultiTab {
    const void* g[MT_MAX];
    float* p[MT_MAX];
    float* buf[MT_MAX];
    void* shadow[MT_MAX];           // optional 16-bit copy of the updated parameter (the next forward's operand), type TS of the kernel
    long n[MT_MAX];
    int chunk0[MT_MAX + 1];         // prefix sum of ceil(n / MT_CHUNK)
    float lr[MT_MAX];
    int count;
};

__device__ __forceinline__ int mt_find(const MultiTab& tab, int c) {
    int t = 0;
    while (t + 1 < tab.count && c >= tab.chunk0[t + 1]) ++t;
    return t;
}

typedef unsigned int mt_u32x2 __attribute__((ext_vector_type(2)));
template <typename TG>
__device__ __forceinline__ void mt_load4(const TG* g, float (&v)[4]) {
    if constexpr (sizeof(TG) == 4) {
        const f32x4 q = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g));   // (streams read once: the update runs beside the VGG forward)
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
        const mt_u32x2 q = __builtin_nontemporal_load(reinterpret_cast<const mt_u32x2*>(g));
        v[0] = H16<TG>::lo(q.x); v[1] = H16<TG>::hi(q.x);
        v[2] = H16<TG>::lo(q.y); v[3] = H16<TG>::hi(q.y);
    }
}

// the same 4 elements as they lie (no conversion: nothing waits for the load), and their unpacking
template <typename TG> struct MtRaw4 { mt_u32x2 r; };
template <> struct MtRaw4<float> { f32x4 r; };
template <typename TG>
__device__ __forceinline__ void mt_load4_raw(const TG* g, MtRaw4<TG>& q) {
    q.r = __builtin_nontemporal_load(reinterpret_cast<const decltype(q.r)*>(g));
}
template <typename TG>
__device__ __forceinline__ void mt_unpack4(const MtRaw4<TG>& q, float (&v)[4]) {
    if constexpr (sizeof(TG) == 4) {
        v[0] = q.r.x; v[1] = q.r.y; v[2] = q.r.z; v[3] = q.r.w;
    } else {
        v[0] = H16<TG>::lo(q.r.x); v[1] = H16<TG>::hi(q.r.x);
        v[2] = H16<TG>::lo(q.r.y); v[3] = H16<TG>::hi(q.r.y);
    }
}

template <typename TG>
__global__ __launch_bounds__(256) void sqnorm_multi_kernel(const MultiTab tab, float* __restrict__ part) {
    __shared__ float red[4];
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    const int total = tab.chunk0[tab.count];
    for (int c = blockIdx.x; c < total; c += gridDim.x) {
        const int t = mt_find(tab, c);
        const long n = tab.n[t];
        const TG* g = reinterpret_cast<const TG*>(tab.g[t]);
        const long base = (long)(c - tab.chunk0[t]) * MT_CHUNK;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long i = base + (q * 256 + threadIdx.x) * 4;
            if (i + 4 <= n) {
                float v[4];
                mt_load4(g + i, v);
                s[q] = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], fmaf(v[3], v[3], s[q]))));
            } else {
                for (long k = i; k < n; ++k) {
                    const float v = Elem<TG>::ld(g + k);
                    s[q] = fmaf(v, v, s[q]);
                }
            }
        }
    }
    const float w = wave_sum((s[0] + s[1]) + (s[2] + s[3]));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// (5 waves per SIMD = at most 96 VGPRs: the update's workgroup must fit beside a convolution workgroup of the VGG forward, whose two waves
// per SIMD hold 416 of the 512 registers -- with more it would not share a CU with one, and one of the two kernels would wait for the other)
// QD: quarters of a chunk whose loads are issued together (4: all 12 loads of the chunk in flight per thread; 2: six at a time -- an
// experiment knob, SGG_OPT_DEPTH, for the update beside the VGG forward: fewer requests in flight, less queueing in front of the forward's)
template <typename TG, typename TS, int QD>
__global__ __launch_bounds__(256, 5) void sgd_multi_kernel(const MultiTab tab, float wd, float mom, int first,
                                                        const float* __restrict__ norm_sq, float max_norm, float grad_scale,
                                                        int* __restrict__ skipped) {
    float coef = grad_scale;
    if (norm_sq) {
        if (!(*norm_sq < 3.0e38f)) {            // non-finite gradients (an overflowed scaled gradient of the f16 mode, a NaN): skip the step
            if (skipped && blockIdx.x == 0 && threadIdx.x == 0) skipped[0] += 1;   // ... and say so (one launch per step is handed the counter)
            return;
        }
        const float c = max_norm / (sqrtf(*norm_sq) * grad_scale + 1e-6f);
        if (c < 1.f) coef *= c;
    }
    const int total = tab.chunk0[tab.count];
    for (int c = blockIdx.x; c < total; c += gridDim.x) {
        const int t = mt_find(tab, c);
        const long n = tab.n[t];
        const TG* g = reinterpret_cast<const TG*>(tab.g[t]);
        float* p = tab.p[t];
        float* buf = tab.buf[t];
        TS* sh = reinterpret_cast<TS*>(tab.shadow[t]);
        const float lr = tab.lr[t];
        const long base = (long)(c - tab.chunk0[t]) * MT_CHUNK;
        if (base + MT_CHUNK <= n) {
            // A whole chunk (all but the last one of a tensor): its 12 loads per thread are issued as they lie -- no conversion, no bounds
            // test, no branch between them -- and only then used.  (Round 5: in the per-quarter form below the compiler put a full
            // `s_waitcnt vmcnt(0)` in front of every quarter's loads, 3 loads in flight instead of 12; a bare kernel with this access
            // pattern streams 5.2 GB in 0.92 ms on the same 256 workgroups, this one needed 1.4 - 1.5 -- tools/pair_probe.py.)
            MtRaw4<TG> rg[4];
            f32x4 rp[4], rb[4];
            // (uniform chunk pointers + 32-bit lane offsets: scalar-base addressing, no 64-bit address per load kept in VGPRs)
            const TG* __restrict__ gc = g + base;
            float* __restrict__ pc = p + base;
            float* __restrict__ bc = buf + base;
            TS* __restrict__ sc = sh ? sh + base : nullptr;
#pragma unroll
          for (int q0 = 0; q0 < 4; q0 += QD) {
#pragma unroll
            for (int q = q0; q < q0 + QD; ++q) {
                const int i = (q * 256 + (int)threadIdx.x) * 4;
                mt_load4_raw(gc + i, rg[q]);
                rp[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pc + i));
                if (!first) rb[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(bc + i));
            }
#pragma unroll
            for (int q = q0; q < q0 + QD; ++q) {
                const int i = (q * 256 + (int)threadIdx.x) * 4;
                float gq[4];
                mt_unpack4(rg[q], gq);
                const float pq[4] = {rp[q].x, rp[q].y, rp[q].z, rp[q].w};
                const float bq[4] = {rb[q].x, rb[q].y, rb[q].z, rb[q].w};
                float nb[4], np[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float gg = coef * gq[k] + wd * pq[k];
                    nb[k] = first ? gg : mom * bq[k] + gg;
                    np[k] = pq[k] - lr * nb[k];
                }
                __builtin_nontemporal_store(f32x4{nb[0], nb[1], nb[2], nb[3]}, reinterpret_cast<f32x4*>(bc + i));
                __builtin_nontemporal_store(f32x4{np[0], np[1], np[2], np[3]}, reinterpret_cast<f32x4*>(pc + i));
                if (sc) {
                    uint2 o;
                    o.x = H16<TS>::pack(np[0], np[1]);
                    o.y = H16<TS>::pack(np[2], np[3]);
                    *reinterpret_cast<uint2*>(sc + i) = o;
                }
            }
          }
            continue;
        }
        // the last, partial chunk of a tensor (and tensors shorter than a chunk): element by element -- at most 4095 elements per tensor
        for (long k = base + threadIdx.x; k < n; k += 256) {
            const float pk = p[k];
            const float gg = coef * Elem<TG>::ld(g + k) + wd * pk;
            const float b = first ? gg : mom * buf[k] + gg;
            buf[k] = b;
            p[k] = pk - lr * b;
            if (sh) Elem<TS>::st(sh + k, p[k]);
        }
    }
}

inline int split_rows(int M, int& rows_per_block) {
    int split = (M + 511) / 512;
    if (split > 64) split = 64;
    if (split < 1) split = 1;
    rows_per_block = (M + split - 1) / split;
    return split;
}

}  // namespace

namespace {
// ------------------------------------------------------------------------------------------------
// Cross-entropy of a logit matrix, loss and gradient in ONE launch (lib/losses.py:41-43,74 -- the 'baseline' form: summed CE of the
// rows divided by a batch-level normaliser).  One wave per row: max, sum of exponentials, -log p[label]; the gradient
// (softmax - onehot) * weight / norm is written in the dtype the backward's GEMMs take, zero-padded to `ldg` columns (the heads are
// 151 / 51 wide, the GEMMs want multiples of 64: no separate pad + cast passes).  `norm` lives on the device (a data-parallel step
// all-reduces it); every row block writes its loss partial (reduce_scalar_kernel adds them in a fixed order: no float atomics).
// A label outside [0, C) (e.g. torch's ignore_index) contributes no loss and a zero gradient row, and raises bit 0 of *flag.
// torch's own path for the two heads is ~25 tiny launches per step (log_softmax, nll, their backwards, casts, pads).
// ------------------------------------------------------------------------------------------------
// acc (+)= sum of part[0 .. n): one workgroup, thread t adds part[t], part[t + 256], ... in order, then a fixed LDS tree
__global__ __launch_bounds__(256) void reduce_scalar_kernel(const float* __restrict__ part, int n, float* __restrict__ acc, int accumulate) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) acc[0] = accumulate ? acc[0] + red[0] : red[0];
}

// out[k][c] (+)= sum_p parts[p][k][c]  (k < 4, c < H; out row stride out_ld), p ascending
__global__ __launch_bounds__(256) void rank4_finalize_kernel(const float* __restrict__ parts, int nparts, int H, float* __restrict__ out,
                                                             int out_ld, int accumulate) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 4 * H) return;
    const int k = i / H, c = i - k * H;
    float s = 0.f;
    for (int p = 0; p < nparts; ++p) s += parts[(long)p * 4 * H + i];
    out[(long)k * out_ld + c] = accumulate ? out[(long)k * out_ld + c] + s : s;
}

template <typename TG>
__global__ __launch_bounds__(256) void ce_fwd_bwd_kernel(const float* __restrict__ logits, int ld, const int64_t* __restrict__ labels,
                                                         int label_stride, int M, int C, const float* __restrict__ norm, float weight, float grad_scale,
                                                         float* __restrict__ part, TG* __restrict__ grad, int ldg, int* __restrict__ flag,
                                                         int mode, float alpha, float beta) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    __shared__ float ws4[4];
    float mine = 0.f;
    if (row < M) {
        const float* x = logits + (long)row * ld;
        float mx = -3.0e38f;
        for (int c = lane; c < C; c += 64) mx = fmaxf(mx, x[c]);
        mx = wave_max(mx);
        float se = 0.f;
        for (int c = lane; c < C; c += 64) se += __expf(x[c] - mx);
        se = wave_sum(se);
        const long lab_raw = labels[(long)row * label_stride];
        const bool ok = lab_raw >= 0 && lab_raw < C;
        const int lab = ok ? (int)lab_raw : 0;
        float scale = 0.f;
        if (ok && mode == 0) scale = weight / norm[0];                       // lib/losses.py:41-43: every row 1 / M
        else if (ok) {                                                       // :44-63: density-normalised edge weights, counts on the device
            const float mfg = norm[0], mbg = norm[1];
            float w = 1.f;                                                   // edge_weights = ones(M)
            if (lab_raw > 0) { if (mfg > 0.f) w = alpha / mfg; }             // :50-51
            else if (mode == 1) { if (mbg > 0.f && mfg > 0.f) w = beta / mfg; }   // :56-57  'dnorm'
            else { if (mbg > 0.f) w = beta / mbg; }                          // :59-60  'dnorm-fgbg'
            scale = weight * w;
        }
        const float lse = mx + __logf(se);
        if (lane == 0) mine = ok ? (lse - x[lab]) * scale : 0.f;
        if (!ok && lane == 0 && flag) atomicOr(flag, 1);
        TG* g = grad + (long)row * ldg;
        for (int c = lane; c < ldg; c += 64) {
            float v = 0.f;
            if (c < C) v = (__expf(x[c] - lse) - (c == lab ? 1.f : 0.f)) * scale * grad_scale;
            Elem<TG>::st(g + c, v);
        }
    }
    if (lane == 0) ws4[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (ws4[0] + ws4[1]) + (ws4[2] + ws4[3]);
}

// counts[0] (+)= #{labels > 0}, counts[1] (+)= #{labels == 0}: M_FG / M_BG of lib/losses.py:29-34 as device numbers (integers in f32:
// exact below 2^24 rows).  One workgroup: the rows of a batch are a few thousand.
__global__ __launch_bounds__(1024) void label_counts_kernel(const int64_t* __restrict__ labels, int label_stride, int M,
                                                            float* __restrict__ counts, int accumulate) {
    __shared__ int sfg[16], sbg[16];
    int fg = 0, bg = 0;
    for (int r = threadIdx.x; r < M; r += 1024) {
        const long l = labels[(long)r * label_stride];
        fg += l > 0;
        bg += l == 0;
    }
    for (int o = 32; o; o >>= 1) { fg += __shfl_xor(fg, o); bg += __shfl_xor(bg, o); }
    if ((threadIdx.x & 63) == 0) { sfg[threadIdx.x >> 6] = fg; sbg[threadIdx.x >> 6] = bg; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int a = 0, b = 0;
        for (int i = 0; i < 16; ++i) { a += sfg[i]; b += sbg[i]; }
        counts[0] = (accumulate ? counts[0] : 0.f) + (float)a;
        counts[1] = (accumulate ? counts[1] : 0.f) + (float)b;
    }
}
}  // namespace

// Workspaces: every reduction below is two-stage -- row blocks write partial rows into `ws`, a second launch adds them in a fixed order
// (no float atomics: bit-reproducible).  ws sizes are stated per entry (split_rows caps the row blocks of the column reductions at 64).

// loss[0] += weight / norm[0] * sum_rows CE(logits[row], labels[row * label_stride]);  grad[M, ldg] = grad_scale * d loss / d logits
// (zero-padded).  ws: f32[(M + 3) / 4].  flag (optional): bit 0 raised when a label lies outside [0, C) (that row: no loss, zero gradient).
extern "C" int sgg_label_counts(const int64_t* labels, int label_stride, int M, float* counts, int accumulate, void* stream) {
    if (!counts || M < 0 || (M > 0 && !labels) || label_stride <= 0 || M >= (1 << 24)) return SGG_ERR_ARG;
    hipLaunchKernelGGL(label_counts_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, labels, label_stride, M, counts, accumulate);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_ce_fwd_bwd(const float* logits, int ld, const int64_t* labels, int label_stride, int M, int C, const float* norm,
                              float weight, float grad_scale, float* loss, int accumulate, void* grad, int ldg, float* ws, int* flag, int g_dtype,
                              int mode, float alpha, float beta, void* stream) {
    if (mode < 0 || mode > 2) return SGG_ERR_ARG;
    if (M == 0) {
        if (!accumulate && loss) {          // no rows: the loss is 0
            hipLaunchKernelGGL(reduce_scalar_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, loss, 0, loss, 0);
            SGG_CHECK_LAUNCH();
        }
        return SGG_OK;
    }
    if (!logits || !labels || !norm || !loss || !grad || !ws || M < 0 || C <= 0 || ld < C || ldg < C || label_stride <= 0) return SGG_ERR_ARG;
    const dim3 grid((M + 3) / 4), blk(256);
    hipStream_t s = (hipStream_t)stream;
    SGG_FOR_DTYPE(g_dtype, hipLaunchKernelGGL(ce_fwd_bwd_kernel<T>, grid, blk, 0, s, logits, ld, labels, label_stride, M, C, norm, weight, grad_scale,
                                              ws, (T*)grad, ldg, flag, mode, alpha, beta));
    SGG_CHECK_LAUNCH();
    hipLaunchKernelGGL(reduce_scalar_kernel, dim3(1), dim3(256), 0, s, ws, (int)grid.x, loss, accumulate);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_dropout_fwd(void* x, int64_t n, float p, uint64_t seed, int dtype, void* stream) {
    if (n == 0) return SGG_OK;
    if (!x || n < 0 || (n & 7) || !(p >= 0.f && p < 1.f)) return SGG_ERR_ARG;
    const long n8 = n / 8;
    const dim3 grid((unsigned)((n8 + 255) / 256)), blk(256);
    hipStream_t s = (hipStream_t)stream;
    SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL(dropout_kernel<T>, grid, blk, 0, s, (T*)x, n8, p, (unsigned long long)seed));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_dropout_fwd_dev(void* x, int64_t n, float p, const uint64_t* seed_dev, uint64_t salt, int dtype, void* stream) {
    if (n == 0) return SGG_OK;
    if (!x || !seed_dev || n < 0 || (n & 7) || !(p >= 0.f && p < 1.f)) return SGG_ERR_ARG;
    const long n8 = n / 8;
    const dim3 grid((unsigned)((n8 + 255) / 256)), blk(256);
    hipStream_t s = (hipStream_t)stream;
    SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL(dropout_dev_kernel<T>, grid, blk, 0, s, (T*)x, n8, p, (const unsigned long long*)seed_dev,
                                            (unsigned long long)salt));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_act_bwd(const void* dy, const void* y, void* dx, int64_t n, float scale, int g_dtype, int y_dtype,
                           void* stream) {
    if (n == 0) return SGG_OK;
    if (!dy || !y || !dx || n < 0 || (n & 7)) return SGG_ERR_ARG;
    const long n8 = n / 8;
    const dim3 grid((unsigned)((n8 + 255) / 256)), blk(256);
    hipStream_t s = (hipStream_t)stream;
    SGG_FOR_DTYPE2(g_dtype, y_dtype, hipLaunchKernelGGL((act_bwd_kernel<TA, TB>), grid, blk, 0, s, (const TA*)dy, (const TB*)y, (TA*)dx, n8, scale));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

// out[c] = sum_r x[r][c].  ws: f32[64 * N] (row blocks' partial sums; not needed -- may be NULL -- when M <= 512)
extern "C" int sgg_colsum(const void* x, int M, int N, int ld, float* out, float* ws, int dtype, void* stream) {
    if (!out || N <= 0 || M < 0 || ld < N) return SGG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (M == 0) return sgg_fill_u32(out, 0u, (size_t)N, s);
    if (!x) return SGG_ERR_ARG;
    int rpb;
    const int split = split_rows(M, rpb);
    if (split > 1 && !ws) return SGG_ERR_ARG;
    float* dst = split > 1 ? ws : out;
    const dim3 grid((N + 63) / 64, split), blk(256);
    if ((N & 7) == 0 && (ld & 7) == 0 && ((uintptr_t)x & 15) == 0) {
        SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL(colsum8_kernel<T>, grid, blk, 0, s, (const T*)x, M, N, ld, dst, rpb));
    } else {
        SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL(colsum_kernel<T>, grid, blk, 0, s, (const T*)x, M, N, ld, dst, rpb));
    }
    SGG_CHECK_LAUNCH();
    return split > 1 ? sgg_reduce_parts(ws, split, N, out, 0, s) : SGG_OK;
}

// train-mode BatchNorm statistics of x[M,C]: sums[0][c] = sum x, sums[1][c] = sum x^2.  ws: f32[64 * 2C]
extern "C" int sgg_bn_stats(const void* x, int M, int C, float* sums, float* ws, int dtype, void* stream) {
    if (!x || !sums || !ws || M <= 0 || C <= 0) return SGG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    int rpb;
    const int split = split_rows(M, rpb);
    const dim3 grid((C + 63) / 64, split), blk(256);
    if ((C & 7) == 0) {
        SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL(bn_stats_kernel<T>, grid, blk, 0, s, (const T*)x, M, C, ws, rpb));
    } else {
        SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL((colsum2_kernel<T, T>), grid, blk, 0, s, (const T*)x, (const T*)x, M, C, ws, rpb));
    }
    SGG_CHECK_LAUNCH();
    return sgg_reduce_parts(ws, split, 2 * C, sums, 0, s);
}

extern "C" int sgg_bn_finalize(const float* sums, int C, int count, const float* count_dev, const float* gamma,
                               const float* beta, float eps, float momentum, float* run_mean, float* run_var, float* mean,
                               float* invstd, float* scale, float* shift, void* stream) {
    if (!sums || !gamma || !beta || !mean || !invstd || !scale || !shift || C <= 0 || count <= 0 || (run_mean && !run_var))
        return SGG_ERR_ARG;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, C, (float)count, count_dev,
                       gamma, beta, eps, momentum, run_mean, run_var, mean, invstd, scale, shift);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

// y = x*scale[c] + shift[c]; max4 != 0: additionally max over 4 consecutive rows, arg-max row index to `arg` (u8)
extern "C" int sgg_bn_apply(const void* x, const float* scale, const float* shift, void* out, unsigned char* arg, int rows_out,
                            int C, int max4, int dtype, void* stream) {
    if (rows_out == 0) return SGG_OK;
    if (!x || !scale || !shift || !out || rows_out < 0 || C <= 0 || (C & 7)) return SGG_ERR_ARG;
    const long total = (long)rows_out * (C / 8);
    const dim3 grid((unsigned)((total + 255) / 256)), blk(256);
    hipStream_t s = (hipStream_t)stream;
    if (max4) {
        SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL((bn_apply_kernel<T, true>), grid, blk, 0, s, (const T*)x, scale, shift, (T*)out, arg, (long)rows_out, C));
    } else {
        SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL((bn_apply_kernel<T, false>), grid, blk, 0, s, (const T*)x, scale, shift, (T*)out, arg, (long)rows_out, C));
    }
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

// backward of [ReLU ->] BatchNorm(batch stats) [-> max over 4 rows]: x[rows,C] = post-ReLU input of the BN,
// dy[rows(/4),C]; outputs dx[rows,C] (gradient at the conv output, ReLU folded) and sums[2][C] = (dbeta, dgamma).
// ws: f32[64 * 2C] (phases 0 and 1: the row blocks' partial sums)
extern "C" int sgg_bn_bwd(const void* dy, const unsigned char* arg, const void* x, const float* mean, const float* invstd,
                          const float* gamma, void* dx, float* sums, float* ws, int rows, int C, int max4, int phase,
                          const float* count_dev, int dtype, void* stream) {
    if (!dy || !x || !mean || !invstd || !gamma || !dx || !sums || rows <= 0 || C <= 0 || (C & 7) || (max4 && !arg))
        return SGG_ERR_ARG;
    if (phase < 0 || phase > 2 || (count_dev && phase != 2) || (phase != 2 && !ws)) return SGG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    int rpb;
    const int split = split_rows(rows, rpb);
    const dim3 g1((C + 63) / 64, split);
    const long total = (long)rows * (C / 8);
    const dim3 g2((unsigned)((total + 255) / 256));
    const float inv = 1.0f / (float)rows;
    if (phase != 2) {
        if (max4) {
            SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, true>), g1, dim3(256), 0, s, (const T*)dy, arg, (const T*)x, mean, invstd, ws, rows, C, rpb));
        } else {
            SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, false>), g1, dim3(256), 0, s, (const T*)dy, arg, (const T*)x, mean, invstd, ws, rows, C, rpb));
        }
        SGG_CHECK_LAUNCH();
        const int rc = sgg_reduce_parts(ws, split, 2 * C, sums, 0, s);
        if (rc != SGG_OK) return rc;
    }
    if (phase != 1) {
        if (max4) {
            SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL((bn_bwd_kernel<T, true>), g2, dim3(256), 0, s, (const T*)dy, arg, (const T*)x, mean, invstd, gamma, sums, (T*)dx, (long)rows, C, inv, count_dev));
        } else {
            SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL((bn_bwd_kernel<T, false>), g2, dim3(256), 0, s, (const T*)dy, arg, (const T*)x, mean, invstd, gamma, sums, (T*)dx, (long)rows, C, inv, count_dev));
        }
        SGG_CHECK_LAUNCH();
    }
    return SGG_OK;
}

extern "C" int sgg_gru_gate_bwd(const void* dh, const float* gi, const float* gh, const float* b_hh, const void* h_prev,
                                void* d_gi, void* d_gh, void* dh_prev, int M, int H, int dtype, void* stream) {
    if (M == 0) return SGG_OK;
    if (!dh || !gi || !d_gi || !d_gh || M < 0 || H <= 0 || (H & 7)) return SGG_ERR_ARG;
    if (gh ? !h_prev : !b_hh) return SGG_ERR_ARG;
    const long total = (long)M * (H / 8);
    const dim3 grid((unsigned)((total + 255) / 256)), blk(256);
    hipStream_t s = (hipStream_t)stream;
    SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL(gru_gate_bwd_kernel<T>, grid, blk, 0, s, (const T*)dh, gi, gh, b_hh, (const T*)h_prev, (T*)d_gi, (T*)d_gh,
                                            (T*)dh_prev, total, H));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_gru_gate_proj_bwd(const void* dh, const void* gh, const float* P, const float* b_ih, const int* so,
                                     const float* node_dots, const float* edge_dots, const float* gate_b, const void* h_prev, void* d_gi,
                                     void* d_gh, void* dh_prev, float* dq, int M, int H, int dtype, int gh_dtype, void* stream) {
    if (M == 0) return SGG_OK;
    const int h8 = H / 8;
    if (!dh || !gh || !P || !b_ih || !so || !node_dots || !edge_dots || !gate_b || !h_prev || !d_gi || !d_gh || !dq || M < 0 || H <= 0 ||
        (H & 7) || h8 > 64 || (h8 & (h8 - 1)))
        return SGG_ERR_ARG;
    if (gh_dtype != SGG_F32 && gh_dtype != dtype) return SGG_ERR_DTYPE;
    const long total = (long)M * h8;
    const dim3 grid((unsigned)((total + 255) / 256)), blk(256);
    hipStream_t s = (hipStream_t)stream;
    if (gh_dtype == SGG_F32) {
        SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL((gru_gate_proj_bwd_kernel<T, float>), grid, blk, 0, s, (const T*)dh, (const float*)gh, P, b_ih, so, node_dots,
                                                edge_dots, gate_b, (const T*)h_prev, (T*)d_gi, (T*)d_gh, (T*)dh_prev, dq, total, H));
    } else {
        SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL((gru_gate_proj_bwd_kernel<T, T>), grid, blk, 0, s, (const T*)dh, (const T*)gh, P, b_ih, so, node_dots,
                                                edge_dots, gate_b, (const T*)h_prev, (T*)d_gi, (T*)d_gh, (T*)dh_prev, dq, total, H));
    }
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_imp_edge_ctx_bwd(const void* e, const int* so, int E, int H, const float* node_dots, const float* edge_dots,
                                    const float* gate_w, const float* gate_b, const float* dq, const void* d_ctx, void* d_e, float* da,
                                    int dtype, void* stream) {
    if (E == 0) return SGG_OK;
    if (!e || !so || !node_dots || !edge_dots || !gate_w || !gate_b || !dq || !d_ctx || !d_e || !da || E < 0 || H <= 0 || (H & 7) || H > MAXH)
        return SGG_ERR_ARG;
    const dim3 grid((E + 3) / 4), blk(256);
    hipStream_t s = (hipStream_t)stream;
    SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL(edge_ctx_bwd_kernel<T>, grid, blk, 0, s, (const T*)e, so, E, H, node_dots, edge_dots, gate_w, gate_b,
                                            dq, (const T*)d_ctx, (T*)d_e, da));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_imp_node_gates_bwd(const float* da, const int* out_ptr, const int* out_ids, const int* in_ptr, const int* in_ids,
                                      const float* gate_w, int N, int H, void* d_v, float* nsum, int dtype, void* stream) {
    if (N == 0) return SGG_OK;
    if (!da || !out_ptr || !out_ids || !in_ptr || !in_ids || !gate_w || !d_v || !nsum || N < 0 || H <= 0 || (H & 7)) return SGG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL(node_gates_bwd_kernel<T>, dim3(N), dim3(64), 0, s, da, out_ptr, out_ids, in_ptr, in_ids, gate_w, H,
                                            (T*)d_v, nsum));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

// out[k, :H] += sum_r a[r,k] * x[r,:]  (k < 4; out row stride out_ld; NOT zeroed: accumulates over calls).  ws: f32[64 * 4 * H]
extern "C" int sgg_rank4_reduce(const float* a, const void* x, int R, int H, float* out, int out_ld, float* ws, int accumulate, int dtype,
                                void* stream) {
    if (R == 0 && accumulate) return SGG_OK;
    if (!out || R < 0 || H <= 0 || out_ld < H || (R > 0 && (!a || !x || !ws))) return SGG_ERR_ARG;
    int rpb;
    const int split = split_rows(R, rpb);
    const dim3 grid((H + 63) / 64, split), blk(256);
    hipStream_t s = (hipStream_t)stream;
    SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL(rank4_reduce_kernel<T>, grid, blk, 0, s, a, (const T*)x, R, H, ws, rpb));
    SGG_CHECK_LAUNCH();
    hipLaunchKernelGGL(rank4_finalize_kernel, dim3((4 * H + 255) / 256), dim3(256), 0, s, ws, R > 0 ? split : 0, H, out, out_ld, accumulate);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

// acc (+)= sum(g^2)  (accumulate = 0 for the first tensor of a step, 1 for the others: no separate clearing launch).  ws: f32[2048]
extern "C" int sgg_sqnorm_acc(const void* g, int64_t n, float* acc, float* ws, int accumulate, int dtype, void* stream) {
    if (n == 0 && accumulate) return SGG_OK;
    if ((!g && n > 0) || !acc || !ws || n < 0) return SGG_ERR_ARG;
    long blocks = (n / 8 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipStream_t s = (hipStream_t)stream;
    SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL(sqnorm_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, s, (const T*)g, (long)n, ws));
    SGG_CHECK_LAUNCH();
    hipLaunchKernelGGL(reduce_scalar_kernel, dim3(1), dim3(256), 0, s, ws, (int)blocks, acc, accumulate);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_sgd_step(float* p, const void* g, float* momentum_buf, int64_t n, float lr, float weight_decay,
                            float momentum, int first_step, const float* norm_sq, float max_norm, float grad_scale,
                            int g_dtype, void* stream) {
    if (n == 0) return SGG_OK;
    if (!p || !g || !momentum_buf || n < 0) return SGG_ERR_ARG;
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipStream_t s = (hipStream_t)stream;
    SGG_FOR_DTYPE(g_dtype, hipLaunchKernelGGL(sgd_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, s, p, (const T*)g, momentum_buf, (long)n, lr,
                                              weight_decay, momentum, first_step, norm_sq, max_norm, grad_scale));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

namespace {
// fills `tab` from entries [lo, hi) of the host arrays; returns the chunk count
int mt_fill(MultiTab& tab, int lo, int hi, const void* const* g, float* const* p, float* const* buf, void* const* shadow,
            const int64_t* n, const float* lr) {
    tab.count = 0;
    tab.chunk0[0] = 0;
    for (int i = lo; i < hi; ++i) {
        if (n[i] <= 0) continue;
        const int k = tab.count++;
        tab.g[k] = g[i];
        tab.p[k] = p ? p[i] : nullptr;
        tab.buf[k] = buf ? buf[i] : nullptr;
        tab.shadow[k] = shadow ? shadow[i] : nullptr;
        tab.n[k] = n[i];
        tab.lr[k] = lr ? lr[i] : 0.f;
        tab.chunk0[k + 1] = tab.chunk0[k] + (int)((n[i] + MT_CHUNK - 1) / MT_CHUNK);
    }
    return tab.chunk0[tab.count];
}
}  // namespace

// acc (+)= sum_i sum(g_i^2) over `count` tensors in one launch per 32 tensors (host arrays of device pointers / sizes); accumulate = 0:
// acc is overwritten (the first call of a step).  Every g_i of 4 or more elements must be 16-byte aligned.  ws: f32[2048]
extern "C" int sgg_sqnorm_multi(const void* const* g, const int64_t* n, int count, float* acc, float* ws, int accumulate, int dtype, void* stream) {
    if (!acc || !ws || count < 0 || (count > 0 && (!g || !n))) return SGG_ERR_ARG;
    for (int i = 0; i < count; ++i)        // (tensors of fewer than 4 elements take the scalar path: any alignment)
        if (n[i] < 0 || (n[i] > 0 && !g[i]) || (n[i] >= 4 && ((uintptr_t)g[i] & 15))) return SGG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    bool launched = false;
    for (int lo = 0; lo < count; lo += MT_MAX) {
        MultiTab tab;
        const int chunks = mt_fill(tab, lo, min(count, lo + MT_MAX), g, nullptr, nullptr, nullptr, n, nullptr);
        if (!chunks) continue;
        const dim3 grid((unsigned)min(chunks, 2048));
        SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL(sqnorm_multi_kernel<T>, grid, dim3(256), 0, s, tab, ws));
        SGG_CHECK_LAUNCH();
        hipLaunchKernelGGL(reduce_scalar_kernel, dim3(1), dim3(256), 0, s, ws, (int)grid.x, acc, (accumulate || launched) ? 1 : 0);
        SGG_CHECK_LAUNCH();
        launched = true;
    }
    if (!launched && !accumulate) {         // nothing to add: acc = 0
        hipLaunchKernelGGL(reduce_scalar_kernel, dim3(1), dim3(256), 0, s, ws, 0, acc, 0);
        SGG_CHECK_LAUNCH();
    }
    return SGG_OK;
}

// sgg_sgd_step over `count` tensors in one launch per 32 tensors; lr per tensor (parameter groups).  shadow (optional
// array, entries may be NULL): 16-bit copy (shadow_dtype: SGG_BF16 / SGG_F16) of each updated parameter, written in the same pass.
extern "C" int sgg_sgd_multi(float* const* p, const void* const* g, float* const* momentum_buf, void* const* shadow,
                             const int64_t* n, const float* lr, int count, float weight_decay, float momentum,
                             int first_step, const float* norm_sq, float max_norm, float grad_scale, int g_dtype, int shadow_dtype,
                             int max_blocks, int* skipped, void* stream) {
    if (count == 0) return SGG_OK;
    if (!p || !g || !momentum_buf || !n || !lr || count < 0) return SGG_ERR_ARG;
    if (shadow_dtype != SGG_BF16 && shadow_dtype != SGG_F16) return SGG_ERR_DTYPE;
    for (int i = 0; i < count; ++i) {
        if (n[i] < 0) return SGG_ERR_ARG;
        if (n[i] == 0) continue;
        if (!p[i] || !g[i] || !momentum_buf[i]) return SGG_ERR_ARG;
        // (tensors of fewer than 4 elements -- a gate bias inside a packed block -- take the scalar path: any alignment)
        if (n[i] >= 4 && (((uintptr_t)p[i] | (uintptr_t)g[i] | (uintptr_t)momentum_buf[i] | (shadow ? (uintptr_t)shadow[i] : 0)) & 15))
            return SGG_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    for (int lo = 0; lo < count; lo += MT_MAX) {
        MultiTab tab;
        const int chunks = mt_fill(tab, lo, min(count, lo + MT_MAX), g, p, momentum_buf, shadow, n, lr);
        if (!chunks) continue;
        // few, fat workgroups (12 float4 loads in flight per thread): 512 already stream at full bandwidth, and leave wave
        // slots for the kernels of another stream (the trainer's pipeline mode asks for 256: the VGG forward runs beside it)
        const dim3 grid((unsigned)min(chunks, max_blocks > 0 ? max_blocks : 512));
        static const int depth = [] { const char* e = getenv("SGG_OPT_DEPTH"); return e ? atoi(e) : 4; }();
#define SGG_SGD_LAUNCH(TS_, QD_) SGG_FOR_DTYPE(g_dtype, hipLaunchKernelGGL((sgd_multi_kernel<T, TS_, QD_>), grid, dim3(256), 0, s, tab, weight_decay, \
                                                                          momentum, first_step, norm_sq, max_norm, grad_scale, lo == 0 ? skipped : (int*)nullptr))
        if (shadow_dtype == SGG_BF16) {
            if (depth == 2) { SGG_SGD_LAUNCH(bf16_t, 2); } else if (depth == 1) { SGG_SGD_LAUNCH(bf16_t, 1); } else { SGG_SGD_LAUNCH(bf16_t, 4); }
        } else {
            if (depth == 2) { SGG_SGD_LAUNCH(f16_t, 2); } else if (depth == 1) { SGG_SGD_LAUNCH(f16_t, 1); } else { SGG_SGD_LAUNCH(f16_t, 4); }
        }
#undef SGG_SGD_LAUNCH
        SGG_CHECK_LAUNCH();
    }
    return SGG_OK;
}
