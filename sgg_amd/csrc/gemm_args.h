// Shared argument block of the MFMA tile kernels (gemm.hip: 128x128 / 256x64 tiles; gemm256.hip: 256x256 ping-pong).
#pragma once
#include "common.h"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
struct GemmArgs {
    const char* A;
    const char* A2;
    const char* Wt;
    const char* W2;             // optional second K-segment of the weights (rows = N, k-tiles >= nt1)
    long lda_b, lda2_b, ldw_b, ldw2_b;  // bytes
    int nt1, nt;                // k-tiles in segment 1 / total (in units of the kernel's K-tile)
    const float* bias;
    const float* pscale;
    const float* pshift;
    char* C;
    long ldc;  // elements
    int M, N, act, out_dt;      // out_dt: element type of C (SGG_F32 / SGG_BF16 / SGG_F16)
    // optional gathered-row addend of the pre-activation: += add_rows[(add_idx ? add_idx[m] : m) * ld_add + n]  (f32)
    const float* add_rows;
    const int* add_idx;
    long ld_add;
    // optional group-broadcast addend: += gadd[m * ld_gadd + (n + gcol0) / ggroup]  (f32; one value per output row and column group)
    const float* gadd;
    long ld_gadd;
    int ggroup, gcol0;
    int m_base;                 // first output row of this launch (rows [m_base, M): a launch may cover the tail of a problem only)
    long splitk_stride;         // bytes between the fp32 partial outputs of consecutive K splits (gridDim.y > 1)
    // conv mode: A is a zero-bordered NHWC plane [B,H+2,W+2,Cin]; C is [B,H+2p,W+2p,N]
    int H, W, Cin, out_pad;
    // x3 mode on PAIR operands (SGG_PAIR16: a row / pixel is [hi | lo], two f16 planes): x3t = K-tiles per plane (0 = off).  The K loop then
    // runs over 3 x3t tiles -- segment 0 (A hi, W hi), 1 (A hi, W lo), 2 (A lo, W hi) -- per tap in conv mode.  Conv mode: Cin = channels
    // per PLANE, cin_px = channels per input pixel (2 Cin for a pair plane; 0 = Cin), the weights' taps are [hi (Cin) | lo (Cin)] as well.
    int x3t, cin_px;
    int x3c;                    // plain GEMM: the three segments are walked CHUNK by chunk of x3c K-tiles (x3c divides x3t): (A hi, W hi), (A hi, W lo),
                                // (A lo, W hi) of one K range back to back, so the second use of a hi panel finds it in L2 instead of re-streaming it
    // output: out_dt == SGG_PAIR16 writes hi at the element offset and lo pair_off elements further; conv mode: cout_px = elements per
    // output pixel (2 N for a pair plane; 0 = N)
    long pair_off;
    int cout_px;
};

// K-tile kt of an x3 GEMM on pair operands -> (segment 0..2, K-tile inside the plane): chunks of x3c tiles, the three segments per chunk
__device__ __forceinline__ void x3_tile(const GemmArgs& g, int kt, int& seg, int& kk) {
    const int c3 = 3 * g.x3c;
    const int chunk = kt / c3, r = kt - chunk * c3;
    seg = (r >= g.x3c) + (r >= 2 * g.x3c);
    kk = chunk * g.x3c + (r - seg * g.x3c);
}
static inline int x3_chunk_tiles(int x3t, int max_c) {          // host: the largest divisor of x3t that is <= max_c
    int c = x3t < max_c ? x3t : max_c;
    while (c > 1 && x3t % c) --c;
    return c < 1 ? 1 : c;
}

__device__ __forceinline__ void glds16(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((glb_void_t*)g, (lds_void_t*)l, 16, 0, 0);
}

// same, with an immediate offset that the hardware adds to BOTH the global and the LDS address
template <int OFF>
__device__ __forceinline__ void glds16_off(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((glb_void_t*)g, (lds_void_t*)l, 16, OFF, 0);
}

// a pointer the compiler can prove wave-uniform (SGPR pair): lets global_load_lds take its scalar-base + 32-bit lane offset form
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    const unsigned long v = (unsigned long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long)hi << 32) | lo);
}

// LDS-DMA from (uniform base + per-lane 32-bit offset).  The empty asm pins the offset's zero-extension next to the load, where
// instruction selection folds it into global_load_lds' scalar-base form (hoisted out of a loop it becomes a 64-bit VGPR add)
__device__ __forceinline__ void glds16_su(const char* ubase, unsigned off, char* l) {
    asm volatile("" : "+v"(off));
    __builtin_amdgcn_global_load_lds((glb_void_t*)(ubase + off), (lds_void_t*)l, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// logical tile id -> tile coordinates: groups of GM M-panels walk N (L2 reuse of both operand panels)
__device__ __forceinline__ void tile_coords_logical(int L, int tilesM, int tilesN, int& tm, int& tn) {
    constexpr int GM = 8;
    const int per_group = GM * tilesN;
    const int grp = L / per_group;
    const int gm0 = grp * GM;
    const int gsz = min(GM, tilesM - gm0);
    const int inl = L - grp * per_group;
    tm = gm0 + inl % gsz;
    tn = inl / gsz;
}
// XCD-contiguous logical id of a workgroup (consecutive block ids land on different XCDs), then the walk above
__device__ __forceinline__ void tile_coords(int bid, int tilesM, int tilesN, int& tm, int& tn) {
    tile_coords_logical(xcd_remap(bid, tilesM * tilesN), tilesM, tilesN, tm, tn);
}

// The per-channel epilogue vectors (bias, post-scale, post-shift) of the 8 consecutive n a lane stores: loaded ONCE per
// thread with two 16-byte loads each.  (Loading them element-wise inside every store -- 64 scattered dword loads per
// lane and tile -- cost ~10 us of a 744-tile launch: the vector-memory issue rate, not bytes.)
struct ChanVec8 {
    float bias[8], ps[8], pt[8];
    int gidx[8];                // group-addend column of each of the lane's 8 output columns ((n + k + gcol0) / ggroup: one division per
    bool has_ps, has_pt;        // column and TILE here instead of one per column and stored row in the epilogue)
};
__device__ __forceinline__ void load_chan8(const float* p, int n, int N, float (&out)[8], float fill) {
    if (!p) {
#pragma unroll
        for (int k = 0; k < 8; ++k) out[k] = fill;
        return;
    }
    if (n + 8 <= N && ((reinterpret_cast<uintptr_t>(p + n) & 15) == 0)) {
        load8(p + n, out);
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) out[k] = n + k < N ? p[n + k] : fill;
    }
}
__device__ __forceinline__ ChanVec8 load_chanvec8(const GemmArgs& g, int n) {
    ChanVec8 c;
    load_chan8(g.bias, n, g.N, c.bias, 0.f);
    load_chan8(g.pscale, n, g.N, c.ps, 1.f);
    load_chan8(g.pshift, n, g.N, c.pt, 0.f);
    c.has_ps = g.pscale != nullptr;
    c.has_pt = g.pshift != nullptr;
    if (g.gadd) {
#pragma unroll
        for (int k = 0; k < 8; ++k) c.gidx[k] = (min(n + k, g.N - 1) + g.gcol0) / g.ggroup;
    }
    return c;
}

// bias -> act -> per-channel affine on 8 consecutive n, then store (vector when aligned and full)
__device__ __forceinline__ void epilogue_store8(const GemmArgs& g, const ChanVec8& c, float (&v)[8], int m, int n, long off, bool vec_ok) {
    const int nv = min(8, g.N - n);
    if (g.add_rows) {
        const float* r = g.add_rows + (long)(g.add_idx ? g.add_idx[m] : m) * g.ld_add + n;
        if (nv == 8 && ((reinterpret_cast<uintptr_t>(r) & 15) == 0)) {
            float t[8];
            load8(r, t);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += t[k];
        } else {
            for (int k = 0; k < nv; ++k) v[k] += r[k];
        }
    }
    if (g.gadd) {
        const float* r = g.gadd + (long)m * g.ld_gadd;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < nv) v[k] += r[c.gidx[k]];                                      // (never past column N - 1: the last group may end there)
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float t = v[k] + c.bias[k];
        if (g.act == SGG_ACT_RELU) t = fmaxf(t, 0.f);
        if (c.has_ps) t = t * c.ps[k];
        if (c.has_pt) t = t + c.pt[k];
        v[k] = t;
    }
    if (g.out_dt == SGG_PAIR16) {              // x = hi + lo: two f16 planes, pair_off elements apart
        float hi[8], lo[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            hi[k] = round_as<f16_t>(v[k]);
            lo[k] = v[k] - hi[k];
        }
        f16_t* ph = reinterpret_cast<f16_t*>(g.C) + off;
        if (vec_ok && nv == 8) {
            store8(ph, hi);
            store8(ph + g.pair_off, lo);
        } else {
            for (int k = 0; k < nv; ++k) {
                ph[k] = (f16_t)hi[k];
                ph[g.pair_off + k] = (f16_t)lo[k];
            }
        }
        return;
    }
    if (vec_ok && nv == 8) {
        if (g.out_dt == SGG_BF16) store8(reinterpret_cast<bf16_t*>(g.C) + off, v);
        else if (g.out_dt == SGG_F16) store8(reinterpret_cast<f16_t*>(g.C) + off, v);
        else store8(reinterpret_cast<float*>(g.C) + off, v);
    } else {
        for (int k = 0; k < nv; ++k) {
            if (g.out_dt == SGG_BF16) reinterpret_cast<bf16_t*>(g.C)[off + k] = f32_to_bf16(v[k]);
            else if (g.out_dt == SGG_F16) reinterpret_cast<f16_t*>(g.C)[off + k] = (f16_t)v[k];
            else reinterpret_cast<float*>(g.C)[off + k] = v[k];
        }
    }
}

template <bool CONV>
__device__ __forceinline__ long out_offset(const GemmArgs& g, int m, int n) {
    if constexpr (CONV) {
        const int hw = g.H * g.W;
        const int b = m / hw, rem = m - b * hw;
        const int y = rem / g.W, x = rem - y * g.W;
        const int op = g.out_pad;
        return ((long)(b * (g.H + 2 * op) + y + op) * (g.W + 2 * op) + x + op) * (g.cout_px ? g.cout_px : g.N) + n;
    } else {
        return (long)m * g.ldc + n;
    }
}

template <bool CONV>
__device__ __forceinline__ const char* a_row_ptr(const GemmArgs& g, int m, int esz) {
    if constexpr (CONV) {
        const int hw = g.H * g.W;
        const int b = m / hw, rem = m - b * hw;
        const int y = rem / g.W, x = rem - y * g.W;
        return g.A + ((long)(b * (g.H + 2) + y) * (g.W + 2) + x) * (g.cin_px ? g.cin_px : g.Cin) * esz;
    } else {
        return g.A + (long)m * g.lda_b;
    }
}
