// Iterative message passing (IMP): the obj<->edge gather / gate / scatter loop and the GRU pointwise part.
// Reference: RelModelStanford.message_pass, sgg_models/rel_model_stanford.py:48-94.
//
// HBM view (per iteration, E edges, N nodes, H channels, s bytes/element): read e_i (E*H*s) once for the edge
// kernel, gather v rows (N*H*s, each re-used ~2(n-1) times -> L2), write e_in (E*H*s), write ctx (N*H*s).
// A lane owns 8 consecutive channels, so every row access is 16-byte (bf16) / 32-byte (f32) pieces of one
// contiguous H-row: a wave reads/writes one whole row (1-2 KiB for H=512) per instruction group.
#include <cstdlib>

#include "common.h"
#include "gemm_args.h"

namespace {

constexpr int MAXH = 512;  // one wave covers H <= 512 with 8 channels per lane

// d[n,k] = w_k[:H] . v[n]   (vertex halves of the four Linear(2H,1) gates, rel_model_stanford.py:41-45)
template <typename T>
__global__ __launch_bounds__(256) void node_gate_dots_kernel(const T* __restrict__ v, int N, int H,
                                                             const float* __restrict__ gw, float* __restrict__ dots) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    const int c0 = lane * 8;
    float p[4] = {0.f, 0.f, 0.f, 0.f};
    if (c0 < H) {
        float x[8];
        load8(v + (long)n * H + c0, x);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float w[8];
            load8(gw + (long)k * 2 * H + c0, w);
#pragma unroll
            for (int j = 0; j < 8; ++j) p[k] = fmaf(w[j], x[j], p[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) p[k] = wave_sum(p[k]);
    if (lane < 4) dots[(long)n * 4 + lane] = p[lane];
}

// One wave per edge (rel_model_stanford.py:76-81, 86-89).
template <typename T>
__global__ __launch_bounds__(256) void edge_ctx_kernel(const T* __restrict__ v, const T* __restrict__ e,
                                                       const int64_t* __restrict__ rel_inds, int E, int H,
                                                       const float* __restrict__ dots, const float* __restrict__ gw,
                                                       const float* __restrict__ gb, T* __restrict__ e_in,
                                                       float* __restrict__ gates) {
    const int ed = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (ed >= E) return;
    const long s = rel_inds[(long)ed * 3 + 1], o = rel_inds[(long)ed * 3 + 2];
    const int c0 = lane * 8;
    const bool act = c0 < H;
    float ee[8], sv[8], ov[8];
    float p[4] = {0.f, 0.f, 0.f, 0.f};
    if (act) {
        load8(e + (long)ed * H + c0, ee);
        load8(v + s * H + c0, sv);
        load8(v + o * H + c0, ov);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float w[8];
            load8(gw + (long)k * 2 * H + H + c0, w);
#pragma unroll
            for (int j = 0; j < 8; ++j) p[k] = fmaf(w[j], ee[j], p[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) p[k] = wave_sum(p[k]);
    const float g_sub = sigmoidf_(dots[s * 4 + 0] + p[0] + gb[0]);
    const float g_obj = sigmoidf_(dots[o * 4 + 1] + p[1] + gb[1]);
    const float g_out = sigmoidf_(dots[s * 4 + 2] + p[2] + gb[2]);
    const float g_in = sigmoidf_(dots[o * 4 + 3] + p[3] + gb[3]);
    if (act) {
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = g_sub * sv[j] + g_obj * ov[j];
        store8(e_in + (long)ed * H + c0, r);
    }
    if (lane == 0) {
        gates[(long)ed * 2] = g_out;
        gates[(long)ed * 2 + 1] = g_in;
    }
}

// One workgroup per node: ctx[n] = sum_{out(n)} g_out * e + sum_{in(n)} g_in * e  (rel_model_stanford.py:91).
// The 4 waves stride over the node's edge lists (4 rows in flight per wave), then reduce through LDS.
template <typename T>
__global__ __launch_bounds__(256) void node_scatter_kernel(const T* __restrict__ e, const float* __restrict__ gates,
                                                           const int* __restrict__ out_ptr, const int* __restrict__ out_ids,
                                                           const int* __restrict__ in_ptr, const int* __restrict__ in_ids,
                                                           int H, T* __restrict__ ctx) {
    __shared__ float red[4][MAXH];
    const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = lane * 8;
    const bool act = c0 < H;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const int* ptr = side ? in_ptr : out_ptr;
        const int* ids = side ? in_ids : out_ids;
        const int beg = ptr[n], end = ptr[n + 1];
        for (int k = beg + wave; k < end; k += 16) {
            float x[4][8], gk[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kk = k + 4 * u;
                gk[u] = 0.f;
                if (kk < end) {
                    const int id = ids[kk];
                    gk[u] = gates[(long)id * 2 + side];
                    if (act) load8(e + (long)id * H + c0, x[u]);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[u][j] = 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(gk[u], x[u][j], acc[j]);
        }
    }
    if (act) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[wave][c0 + j] = acc[j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += 256)
        Elem<T>::st(ctx + (long)n * H + c, (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]));
}

// ------------------------------------------------------------------------------------------------
// Fused IMP gather / gate / scatter: ONE launch per iteration (replaces gate_dots + edge_ctx + node_scatter).
// Node-centric units (n, side), persistent waves.  side 0 owns n's out-edges (n -> o): g_sub, g_obj, g_out for each,
// e_in[e] = g_sub*v[n] + g_obj*v[o] written, ctx_out[n] = sum g_out*e reduced in registers.  side 1 owns n's in-edges
// (m -> n): g_in needs only a_in[n] and the edge row it streams anyway, so ctx_in[n] = sum g_in*e has no cross-unit
// dependency: no gate array, no atomics.  ctx = ctx_out + ctx_in is never formed: the node GRU's input GEMM takes the
// two halves as a K-split A operand against [W_ih | W_ih] (linearity).
// W = waves per unit: 1 when there are enough units to fill the chip (no LDS, no barrier), 4 (one workgroup per unit,
// LDS reduction) for small graphs.  Edge indices come as int32 (s,o) pairs; when the edge list is sorted by subject
// (flags[0], set by sgg_edge_csr) out-edge ids are the identity and need no index load.  The six gate vectors a lane
// needs stay packed in registers across units; rows stay packed (4 VGPRs per 8 bf16) until used; the next chunk's
// object indices are fetched one chunk ahead; gate dot products reduce on the DPP path.
// HBM view: each e row is read twice (second read = L2 / Infinity-Cache hit), e_in written once; v rows are L2-resident.
// A lane owns 8 channels: every row access is one 16-byte (bf16) piece per lane, 1 KiB per wave.
// ------------------------------------------------------------------------------------------------
template <typename T, int W>
__global__ __launch_bounds__(256, 3) void imp_fused_kernel(const T* __restrict__ v, const T* __restrict__ e,
                                                           const int* __restrict__ so, const int* __restrict__ flags,
                                                           const int* __restrict__ out_ptr, const int* __restrict__ out_ids,
                                                           const int* __restrict__ in_ptr, const int* __restrict__ in_ids,
                                                           int N, int H, const T* __restrict__ gw, const float* __restrict__ gb,
                                                           T* __restrict__ e_in, T* __restrict__ ctx2) {
    __shared__ float red[W > 1 ? 4 : 1][W > 1 ? MAXH : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = W > 1 ? wave : 0;
    const int u0 = W > 1 ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;
    const int ustride = W > 1 ? (int)gridDim.x : (int)gridDim.x * 4;
    const int c0 = lane * 8;
    const bool act = c0 < H;
    const bool contig = flags[0] != 0;
    Raw8<T> w0, w1, w2, w3, w4, w5;
    w0.zero(); w1.zero(); w2.zero(); w3.zero(); w4.zero(); w5.zero();
    int loaded = -1;
    for (int u = u0; u < 2 * N; u += ustride) {
        const int side = u >= N ? 1 : 0, n = side ? u - N : u;
        if (loaded != side && act) {
            if (side == 0) {
                w0.load(gw + 0 * 2 * H + c0);        // sub_vert, vertex half
                w1.load(gw + 2 * 2 * H + c0);        // out_edge, vertex half
                w2.load(gw + 0 * 2 * H + H + c0);    // sub_vert, edge half
                w3.load(gw + 1 * 2 * H + c0);        // obj_vert, vertex half
                w4.load(gw + 1 * 2 * H + H + c0);    // obj_vert, edge half
                w5.load(gw + 2 * 2 * H + H + c0);    // out_edge, edge half
            } else {
                w0.load(gw + 3 * 2 * H + c0);        // in_edge, vertex half
                w1.load(gw + 3 * 2 * H + H + c0);    // in_edge, edge half
            }
        }
        loaded = side;
        float vn[8], acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) vn[j] = acc[j] = 0.f;
        if (act) load8(v + (long)n * H + c0, vn);
        if (side == 0) {
            float a_sub = 0.f, a_out = 0.f;
            {
                float t0[8], t1[8];
                w0.get(t0);
                w1.get(t1);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    a_sub = fmaf(t0[j], vn[j], a_sub);
                    a_out = fmaf(t1[j], vn[j], a_out);
                }
            }
            a_sub = wave_sum(a_sub) + gb[0];
            a_out = wave_sum(a_out) + gb[2];
            const float b_obj = gb[1];
            const int beg = out_ptr[n], end = out_ptr[n + 1];
            int idn[4], on[4];   // ids / objects of the NEXT chunk (fetched one chunk ahead)
            {
                const int k = beg + sub * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int kk = min(k + q, end - 1);
                    idn[q] = (contig || kk < 0) ? kk : out_ids[kk];
                    on[q] = kk >= 0 ? so[2 * (long)idn[q] + 1] : 0;
                }
            }
            for (int k = beg + sub * 4; k < end; k += 4 * W) {
                int id[4];
                Raw8<T> er[4], vr[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    id[q] = idn[q];
                    if (act) {
                        er[q].load(e + (long)id[q] * H + c0);
                        vr[q].load(v + (long)on[q] * H + c0);
                    } else {
                        er[q].zero();
                        vr[q].zero();
                    }
                }
                {
                    const int kn = k + 4 * W;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int kk = min(kn + q, end - 1);
                        idn[q] = contig ? kk : out_ids[kk];
                        on[q] = so[2 * (long)idn[q] + 1];
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float ee[8], ov[8];
                    er[q].get(ee);
                    vr[q].get(ov);
                    // the three edge-side gate dot products on the packed rows (v_dot2_f32_bf16 for bf16)
                    const float p0 = dot8(w2, er[q], 0.f);
                    const float p1 = dot8(w4, er[q], dot8(w3, vr[q], 0.f));
                    const float p2 = dot8(w5, er[q], 0.f);
                    const float gs = sigmoidf_(a_sub + wave_sum(p0)), go = sigmoidf_(wave_sum(p1) + b_obj);
                    const float gx = sigmoidf_(a_out + wave_sum(p2));
                    if (k + q < end && act) {
                        float r[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            r[j] = gs * vn[j] + go * ov[j];
                            acc[j] = fmaf(gx, ee[j], acc[j]);
                        }
                        store8(e_in + (long)id[q] * H + c0, r);
                    }
                }
            }
        } else {
            float a_in = 0.f;
            {
                float t0[8];
                w0.get(t0);
#pragma unroll
                for (int j = 0; j < 8; ++j) a_in = fmaf(t0[j], vn[j], a_in);
            }
            a_in = wave_sum(a_in) + gb[3];
            const int beg = in_ptr[n], end = in_ptr[n + 1];
            for (int k = beg + sub * 8; k < end; k += 8 * W) {
                Raw8<T> er[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int id = in_ids[min(k + q, end - 1)];
                    if (act) er[q].load(e + (long)id * H + c0);
                    else er[q].zero();
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float ee[8];
                    er[q].get(ee);
                    const float p = dot8(w1, er[q], 0.f);
                    const float g = (k + q < end) ? sigmoidf_(a_in + wave_sum(p)) : 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = fmaf(g, ee[j], acc[j]);
                }
            }
        }
        T* dst = ctx2 + ((long)side * N + n) * H;
        if constexpr (W > 1) {
            __syncthreads();
            if (act) {
#pragma unroll
                for (int j = 0; j < 8; ++j) red[wave][c0 + j] = acc[j];
            }
            __syncthreads();
            for (int c = threadIdx.x; c < H; c += 256) Elem<T>::st(dst + c, (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]));
        } else {
            if (act) store8(dst + c0, acc);
        }
    }
}

// GRU pointwise part, ATen's formulation: r=s(ir+hr) z=s(iz+hz) n=tanh(in + r*hn) h'=(h-n)*z+n.
template <typename TG, typename T>
__global__ __launch_bounds__(256) void gru_gate_kernel(const TG* __restrict__ gi, const TG* __restrict__ gh,
                                                       const float* __restrict__ b_hh, const T* __restrict__ h_prev,
                                                       T* __restrict__ h_out, long total, int H,
                                                       const float* __restrict__ dot_w, int dot_ld, float* __restrict__ dots) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;  // over M*H/8
    if (i >= total) return;
    const int h8 = H >> 3;
    const long m = i / h8;
    const int c = (int)(i - m * h8) * 8;
    float ir[8], iz[8], in_[8], hr[8], hz[8], hn[8], hp[8], o[8];
    const TG* gim = gi + m * 3 * H + c;
    load8(gim, ir);
    load8(gim + H, iz);
    load8(gim + 2 * H, in_);
    if (gh) {
        const TG* ghm = gh + m * 3 * H + c;
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
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float r = 1.f / (1.f + expf(-(ir[j] + hr[j])));
        const float z = 1.f / (1.f + expf(-(iz[j] + hz[j])));
        const float n = tanhf(in_[j] + r * hn[j]);
        o[j] = (hp[j] - n) * z + n;
    }
    store8(h_out + m * H + c, o);
    if (dots) {
        // gate pre-activations of the NEXT message-passing step, produced while the row is in registers:
        // dots[m,k] = dot_w[k,:] . h'[m,:] on the values as stored (rounded to T).  H/8 lanes (a power of two <= 64) hold one
        // row, rows never straddle a wave, and whole rows are active or inactive together (total = M * H/8).
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = bf16_to_f32(f32_to_bf16(o[j]));
        }
        float p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float w[8];
            load8(dot_w + (long)k * dot_ld + c, w);
            p[k] = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) p[k] = fmaf(w[j], o[j], p[k]);
        }
        for (int off = h8 >> 1; off > 0; off >>= 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) p[k] += __shfl_xor(p[k], off, 64);
        }
        if (c == 0) *reinterpret_cast<f32x4*>(dots + m * 4) = f32x4{p[0], p[1], p[2], p[3]};
    }
}

// ------------------------------------------------------------------------------------------------
// Sliced IMP step: every edge row read ONCE.  A workgroup owns (graph g, channel slice): LP lanes x 16 B = one PIECE of a
// row (128 B: 64 bf16 / 32 f32 channels at LP = 8).  The gate pre-activations arrive as dot products made by the kernels
// that wrote v and e (gru_gate_kernel above), so nothing here needs a whole row:
//   phase 0  the graph's vertex pieces, vertex dots, list offsets and in-list ids -> LDS
//   phase 1  a lane group owns (subject n, part of n's out-list): per edge it loads the row piece (U edges in flight),
//            makes the four gates on 4 lanes (shared through DPP quad broadcasts, no LDS), writes
//            e_in piece = g_sub * v[n] + g_obj * v[o] to HBM, adds g_out * row to its ctx_out partial (registers), and parks
//            the row piece and g_in in LDS
//   phase 2  a lane group owns (object n, part of n's in-list): ctx_in partial = sum g_in * parked piece
//   the P lane groups of a node sit in one wave: their partials are added with xor-shuffles, part 0 stores
// HBM traffic = algorithmic (e read once, e_in written once) + 16 B of dots per edge and slice.  LDS instructions per edge
// and lane group: 4 in phase 1 (vertex dot, v[o] piece, park row, park gate) + 3 in phase 2.  Needs the edge list sorted by
// (graph, subject) -- out-lists are ranges -- and the graph to fit: edges <= EMAX, nodes <= SL_NMAX.
// ------------------------------------------------------------------------------------------------
#ifndef SGG_SL_THREADS
#define SGG_SL_THREADS 512   // 8 waves: two workgroups share a CU when the parked pieces leave room (measured best)
#endif
constexpr int SL_THREADS = SGG_SL_THREADS;
constexpr int SL_NMAX = 64;
#ifndef SGG_SLICED_ABL
#define SGG_SLICED_ABL 0   // kernel experiments only: 1 no phase 2, 2 no e_in stores, 5 clock stamps, 6 copy only
#endif

// 16 bytes of a row: 8 bf16 or 4 f32 channels
template <typename T> struct Piece16;
template <> struct Piece16<bf16_t> {
    u32x4 r;
    __device__ __forceinline__ void get(float (&x)[8]) const {
        x[0] = __uint_as_float(r.x << 16); x[1] = __uint_as_float(r.x & 0xffff0000u);
        x[2] = __uint_as_float(r.y << 16); x[3] = __uint_as_float(r.y & 0xffff0000u);
        x[4] = __uint_as_float(r.z << 16); x[5] = __uint_as_float(r.z & 0xffff0000u);
        x[6] = __uint_as_float(r.w << 16); x[7] = __uint_as_float(r.w & 0xffff0000u);
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&x)[8]) { store8(p, x); }
};
template <> struct Piece16<float> {
    f32x4 r;
    __device__ __forceinline__ void get(float (&x)[4]) const { x[0] = r.x; x[1] = r.y; x[2] = r.z; x[3] = r.w; }
    static __device__ __forceinline__ void store(float* p, const float (&x)[4]) {
        *reinterpret_cast<f32x4*>(p) = f32x4{x[0], x[1], x[2], x[3]};
    }
};

template <int LP> struct SliceCfg;
template <> struct SliceCfg<8> { static constexpr int EMAX = 1024; };   // 128 KB of parked pieces
template <> struct SliceCfg<4> { static constexpr int EMAX = 1792; };   // 112 KB
template <> struct SliceCfg<2> { static constexpr int EMAX = 3072; };   //  96 KB

// LDS bytes for graphs of at most `emax` edges (emax a multiple of 8)
template <int LP> constexpr int slice_lds_bytes(int emax) {
    return emax * (LP * 16 + 4 + 2) + SL_NMAX * (LP * 16 + 16) + (SL_NMAX + 4) * 4;
}

// value of lane (quad base + q) for every lane of a quad (DPP quad_perm: VALU only, no LDS crossbar)
template <int CTRL> __device__ __forceinline__ float quad_bcast(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}

template <typename T, int LP>
__global__ __launch_bounds__(SL_THREADS) void imp_sliced_kernel(const T* __restrict__ v, const T* __restrict__ e,
                                                                const int* __restrict__ so, const int* __restrict__ out_ptr,
                                                                const int* __restrict__ in_ptr, const int* __restrict__ in_ids,
                                                                const int* __restrict__ img_ptr, int B, int N, int H,
                                                                const float* __restrict__ ndots, const float* __restrict__ edots,
                                                                const float* __restrict__ gb, T* __restrict__ e_in,
                                                                T* __restrict__ ctx2, int EMAX, int sum_ctx) {
    constexpr int PIECE = LP * 16, CHL = 16 / (int)sizeof(T), GROUPS = SL_THREADS / LP;
    constexpr int U = 8;                         // edges in flight per lane group: their loads are issued before any is used
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* stage = smem;                                                        // [EMAX][PIECE] parked row pieces
    float* gin = reinterpret_cast<float*>(stage + (long)EMAX * PIECE);         // [EMAX] g_in of each edge
    unsigned short* in_loc = reinterpret_cast<unsigned short*>(gin + EMAX);    // [EMAX] in-list entries, graph-local
    char* vs = reinterpret_cast<char*>(in_loc + EMAX);                          // [SL_NMAX][PIECE]
    float* nd = reinterpret_cast<float*>(vs + SL_NMAX * PIECE);                 // [SL_NMAX][4]
    int* iptr = reinterpret_cast<int*>(nd + SL_NMAX * 4);                       // [SL_NMAX + 1], graph-local
    const int S = H * (int)sizeof(T) / PIECE;                                   // slices per graph
    const int L = xcd_remap((int)blockIdx.x, B * S);                            // a graph's slices share an XCD (dots, lists in its L2)
    const int g = L / S, slice = L - g * S;
    const int tid = threadIdx.x, sub = tid % LP, grp = tid / LP;
#if SGG_SLICED_ABL == 5
    long long tck[8];
    tck[0] = clock64();
#define SGG_TICK(i) tck[i] = clock64();
#else
#define SGG_TICK(i)
#endif
    // dependent-load level 1: img_ptr = [node offsets (B+1) | edge offsets (B+1) | per graph: SL_NMAX+2 relative out offsets].
    // Which node a lane group owns depends on P, i.e. on the node count that is being loaded right now: fetch the list range of
    // every candidate (P = 1, 2, 4, ...) from the graph's table -- its address needs blockIdx only -- and pick afterwards.
    constexpr int NC = LP == 8 ? 4 : LP == 4 ? 5 : 6;         // lg P in [0, NC): P LP <= 64
    int cand_b[NC], cand_e[NC];
    {
        const int* tab = img_ptr + 2 * (B + 1) + (long)g * (SL_NMAX + 2);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int nc = min(grp >> c, SL_NMAX);
            cand_b[c] = tab[nc];
            cand_e[c] = tab[nc + 1];
        }
    }
    const int n0 = img_ptr[g], n1 = img_ptr[g + 1], Nn = n1 - n0;
    const int e0 = img_ptr[B + 1 + g], e1 = img_ptr[B + 2 + g], Ee = e1 - e0;
    const int i0 = e0;                           // edges are grouped by graph: the in-lists of earlier graphs hold e0 entries
    const long col = (long)slice * (PIECE / (int)sizeof(T)) + sub * CHL;        // this lane's first channel
    if (Ee > EMAX || Nn > SL_NMAX) {
        // the host's promise about this graph (edge_csr(graphs=...)) does not hold: nothing may be parked.  Poison the graph's
        // outputs instead of corrupting LDS -- NaNs surface in the first loss / score that reads them.
        float nanv[CHL];
#pragma unroll
        for (int j = 0; j < CHL; ++j) nanv[j] = __builtin_nanf("");
        for (int k = grp; k < Ee; k += GROUPS) Piece16<T>::store(e_in + (long)(e0 + k) * H + col, nanv);
        for (int k = grp; k < Nn; k += GROUPS) {
            Piece16<T>::store(ctx2 + (long)(n0 + k) * H + col, nanv);
            if (!sum_ctx) Piece16<T>::store(ctx2 + ((long)N + n0 + k) * H + col, nanv);
        }
        return;
    }
    // the four gates of an edge are spread over the lanes of a quad: lane q makes gate q (LP >= 4), or gates q&1, (q&1)+2 (LP = 2)
    constexpr int GI = LP >= 4 ? 1 : 2;
    int gk[GI];
    float bias[GI];
#pragma unroll
    for (int i = 0; i < GI; ++i) {
        gk[i] = LP >= 4 ? (sub & 3) : (sub & 1) + 2 * i;
        bias[i] = gb[gk[i]];
    }
    // P lane groups share a node (P a power of two, the P groups inside one wave): its lists are cut into P contiguous parts.
    // Nn <= SL_NMAX <= GROUPS: every node has its own lane group(s), one node per group.
    int P = 1, lgP = 0;
    while (2 * P * LP <= 64 && 2 * P * Nn <= GROUPS) {
        P *= 2;
        ++lgP;
    }
    const int part = grp % P, n = grp / P;
    const bool has_node = n < Nn;
    int ob = cand_b[0], oe = cand_e[0];
#pragma unroll
    for (int c = 1; c < NC; ++c) {
        if (lgP == c) {
            ob = cand_b[c];
            oe = cand_e[c];
        }
    }
    // level 2 (vector, all independent): the first edges of this group's list + everything phase 0 parks in LDS
    u32x4 p_v = {0, 0, 0, 0};
    f32x4 p_nd = {0, 0, 0, 0};
    int p_ip = 0;
    if (grp < Nn) p_v = *reinterpret_cast<const u32x4*>(v + (long)(n0 + grp) * H + col);
    if (tid < Nn) p_nd = *reinterpret_cast<const f32x4*>(ndots + (long)(n0 + tid) * 4);
    if (tid <= Nn) p_ip = in_ptr[n0 + tid] - i0;
    constexpr int INL = 3;                       // in-list entries per thread held in registers (more: strided loop below)
    int p_in[INL];
#pragma unroll
    for (int q = 0; q < INL; ++q) p_in[q] = (tid + q * SL_THREADS < Ee) ? in_ids[i0 + tid + q * SL_THREADS] - e0 : 0;
    // level 3: the first U edges of this group's part of the out-list
    const int chunk = (oe - ob + P - 1) / P;
    const int k0 = ob + part * chunk, k1 = min(oe, k0 + chunk);
    Piece16<T> row[U];
    float de[U][GI];
    int on[U];
    auto issue = [&](int kb) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ec = e0 + max(min(kb + u, k1 - 1), 0);
            row[u].r = *reinterpret_cast<const decltype(row[u].r)*>(e + (long)ec * H + col);
#pragma unroll
            for (int i = 0; i < GI; ++i) de[u][i] = edots[(long)ec * 4 + gk[i]];
            on[u] = so[2 * (long)ec + 1] - n0;
        }
    };
    SGG_TICK(1)
    if (k0 < k1) issue(k0);
    // ---- phase 0: park
    if (grp < Nn) *reinterpret_cast<u32x4*>(vs + grp * PIECE + sub * 16) = p_v;
    if (tid < Nn) *reinterpret_cast<f32x4*>(nd + tid * 4) = p_nd;
    if (tid <= Nn) iptr[tid] = p_ip;
#pragma unroll
    for (int q = 0; q < INL; ++q)
        if (tid + q * SL_THREADS < Ee) in_loc[tid + q * SL_THREADS] = (unsigned short)p_in[q];
    for (int k = tid + INL * SL_THREADS; k < Ee; k += SL_THREADS) in_loc[k] = (unsigned short)(in_ids[i0 + k] - e0);
    SGG_TICK(2)
    __syncthreads();
    SGG_TICK(3)
    // ---- phase 1: out-lists
    float out_sum[CHL];                           // sum_ctx: the node's finished ctx_out, kept for the single store after phase 2
#pragma unroll
    for (int j = 0; j < CHL; ++j) out_sum[j] = 0.f;
    if (has_node) {
        Piece16<T> vnp;
        vnp.r = *reinterpret_cast<const decltype(vnp.r)*>(vs + n * PIECE + sub * 16);
        float vn[CHL], acc[CHL];
        vnp.get(vn);
#pragma unroll
        for (int j = 0; j < CHL; ++j) acc[j] = 0.f;
        float ndn[GI];                            // dots of v[n] for this lane's even gates (sub_vert, out_edge)
#pragma unroll
        for (int i = 0; i < GI; ++i) ndn[i] = nd[n * 4 + gk[i]];
        for (int kb = k0; kb < k1; kb += U) {
            if (kb != k0) issue(kb);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int el = kb + u;
                const bool live = el < k1;       // lanes of a quad share `live`: the DPP broadcasts below stay inside a lane group
#if SGG_SLICED_ABL == 6                          // experiment: the access pattern alone (row piece in, row piece out)
                if (live) *reinterpret_cast<decltype(row[u].r)*>(e_in + (long)(e0 + el) * H + col) = row[u].r;
                continue;
#endif
                // gate k: 0 sub_vert(v[s]), 1 obj_vert(v[o]), 2 out_edge(v[s]), 3 in_edge(v[o])  (rel_model_stanford.py:78-89)
                float gate[GI];
#pragma unroll
                for (int i = 0; i < GI; ++i) {
                    const float vd = (gk[i] & 1) ? nd[on[u] * 4 + gk[i]] : ndn[i];
                    gate[i] = sigmoidf_(vd + de[u][i] + bias[i]);
                }
                float gs, go, gx;
                if constexpr (LP >= 4) {
                    gs = quad_bcast<0x00>(gate[0]);
                    go = quad_bcast<0x55>(gate[0]);
                    gx = quad_bcast<0xAA>(gate[0]);
                    if (live && sub == 3) gin[el] = gate[0];
                } else {                          // pairs: lane 0 holds (sub, out), lane 1 holds (obj, in)
                    gs = quad_bcast<0xA0>(gate[0]);
                    go = quad_bcast<0xF5>(gate[0]);
                    gx = quad_bcast<0xA0>(gate[1]);
                    if (live && sub == 1) gin[el] = gate[1];
                }
                if (live) {
                    *reinterpret_cast<decltype(row[u].r)*>(stage + (long)el * PIECE + sub * 16) = row[u].r;
                    Piece16<T> vop;
                    vop.r = *reinterpret_cast<const decltype(vop.r)*>(vs + on[u] * PIECE + sub * 16);
                    float x[CHL], y[CHL], r[CHL];
                    row[u].get(x);
                    vop.get(y);
#pragma unroll
                    for (int j = 0; j < CHL; ++j) {
                        r[j] = gs * vn[j] + go * y[j];
                        acc[j] = fmaf(gx, x[j], acc[j]);
                    }
                    if (SGG_SLICED_ABL != 2) Piece16<T>::store(e_in + (long)(e0 + el) * H + col, r);
                }
            }
        }
        for (int off = LP; off < P * LP; off <<= 1) {
#pragma unroll
            for (int j = 0; j < CHL; ++j) acc[j] += __shfl_xor(acc[j], off, 64);
        }
        if (sum_ctx) {
#pragma unroll
            for (int j = 0; j < CHL; ++j) out_sum[j] = acc[j];
        } else if (part == 0) {
            Piece16<T>::store(ctx2 + (long)(n0 + n) * H + col, acc);
        }
    }
    SGG_TICK(4)
    __syncthreads();
    SGG_TICK(5)
    // ---- phase 2: in-lists, from the parked pieces
    if (has_node && SGG_SLICED_ABL != 1 && SGG_SLICED_ABL != 6) {
        const int beg = iptr[n], end = iptr[n + 1], ch = (end - beg + P - 1) / P;
        const int j0 = beg + part * ch, j1 = min(end, j0 + ch);
        float acc[CHL];
#pragma unroll
        for (int j = 0; j < CHL; ++j) acc[j] = 0.f;
        for (int kb = j0; kb < j1; kb += U) {
            int el[U];
#pragma unroll
            for (int u = 0; u < U; ++u) el[u] = in_loc[min(kb + u, j1 - 1)];
            Piece16<T> rw[U];
            float gv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                rw[u].r = *reinterpret_cast<const decltype(rw[u].r)*>(stage + (long)el[u] * PIECE + sub * 16);
                gv[u] = (kb + u < j1) ? gin[el[u]] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float x[CHL];
                rw[u].get(x);
#pragma unroll
                for (int j = 0; j < CHL; ++j) acc[j] = fmaf(gv[u], x[j], acc[j]);
            }
        }
        for (int off = LP; off < P * LP; off <<= 1) {
#pragma unroll
            for (int j = 0; j < CHL; ++j) acc[j] += __shfl_xor(acc[j], off, 64);
        }
        if (sum_ctx) {                            // training: ctx = ctx_out + ctx_in in one [N,H] tensor (it is an operand of d W_ih)
#pragma unroll
            for (int j = 0; j < CHL; ++j) acc[j] += out_sum[j];
            if (part == 0) Piece16<T>::store(ctx2 + (long)(n0 + n) * H + col, acc);
        } else if (part == 0) {
            Piece16<T>::store(ctx2 + ((long)N + n0 + n) * H + col, acc);
        }
    }
#if SGG_SLICED_ABL == 5
    SGG_TICK(6)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SGG_TICK(7)
    if ((tid & 63) == 0 && blockIdx.x < 4) {
        long long* dbg = reinterpret_cast<long long*>(ctx2) + ((long)blockIdx.x * 16 + (tid >> 6)) * 8;
        for (int i = 0; i < 8; ++i) dbg[i] = tck[i];
    }
#endif
#undef SGG_TICK
}


// ---------------------------------------------------------------------------------------------------------------------
// Persistent, software-pipelined form of the sliced step: the kernel the forward runs.
//
// Same unit of work as imp_sliced_kernel -- (graph, 64-byte slice of the rows) -- and the same arithmetic in the same
// order, so the two are bit-identical.  What changes is WHEN memory is touched.  The short-lived form pays, per unit, three
// dependent load levels (graph tables -> lists / vertex pieces -> edge rows) with nothing of its own in flight, then
// computes, then stores: 70 us at B=128 for 43 us of traffic.  Here a workgroup is resident (<= 2 per CU) and walks its
// XCD's units; while unit u is in its gate / accumulate phase (VALU + LDS), the loads of unit u+1 are already in flight:
//
//     iteration u:
//       1. park B(u); barrier             lists / vertex pieces / vertex dots of u (loaded one iteration ago) into ITS copy of the
//                                         small arrays (two copies, 4.7 KB each: stragglers of phase 2(u-1) still read the other)
//       2. issue A(u+1)                   out-list range of this lane group's node in unit u+1
//       3. phase 1(u)                     rows of u are in registers (issued one iteration ago): gates, e_in stores, park
//       4. issue B(u+1), C(u+1)           small arrays and the first U edge rows of u+1 per lane group (+ scalar header of u+2)
//       5. barrier;  phase 2(u) from LDS; ctx store          -- B(u+1), C(u+1) are in flight under all of this
//
// Two barriers per unit; the header (node / edge offsets) runs two units ahead on the scalar unit.
template <int LP> constexpr int stream_small_bytes(int emax, int nmax) {
    return emax * 2 + nmax * (LP * 16) + nmax * 16 + (nmax + 4) * 4;
}
template <int LP> constexpr int stream_lds_bytes(int emax, int nmax) {
    return emax * (LP * 16 + 4) + 2 * stream_small_bytes<LP>(emax, nmax);
}

template <typename T, int LP>
__global__ __launch_bounds__(SL_THREADS, SL_THREADS / 128) void imp_stream_kernel(
    const T* __restrict__ v, const T* __restrict__ e, const int* __restrict__ so, const int* __restrict__ in_ptr,
    const int* __restrict__ in_ids, const int* __restrict__ img_ptr, int B, int N, int H, const float* __restrict__ ndots,
    const float* __restrict__ edots, const float* __restrict__ gb, T* __restrict__ e_in, T* __restrict__ ctx2, int EMAX, int NMAX,
    int sum_ctx) {
    constexpr int PIECE = LP * 16, CHL = 16 / (int)sizeof(T), GROUPS = SL_THREADS / LP;
    constexpr int U = 8, INL = 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* stage = smem;                                                        // [EMAX][PIECE] parked row pieces
    float* gin = reinterpret_cast<float*>(stage + (long)EMAX * PIECE);         // [EMAX] g_in of each edge
    char* small0 = reinterpret_cast<char*>(gin + EMAX);
    const int small_bytes = EMAX * 2 + NMAX * PIECE + NMAX * 16 + (NMAX + 4) * 4;
    // copy `par` of the small arrays: in_loc u16[EMAX] | vs [NMAX][PIECE] | nd f32[NMAX][4] | iptr int[NMAX+4]
    auto sm_inloc = [&](int par) { return reinterpret_cast<unsigned short*>(small0 + par * small_bytes); };
    auto sm_vs = [&](int par) { return small0 + par * small_bytes + EMAX * 2; };
    auto sm_nd = [&](int par) { return reinterpret_cast<float*>(small0 + par * small_bytes + EMAX * 2 + NMAX * PIECE); };
    auto sm_iptr = [&](int par) { return reinterpret_cast<int*>(small0 + par * small_bytes + EMAX * 2 + NMAX * PIECE + NMAX * 16); };

    const int S = H * (int)sizeof(T) / PIECE, units = B * S;
    // this workgroup's units: XCD x (dispatch puts block b on XCD b % 8) owns a contiguous range of units, so that the
    // slices of a graph -- which share its tables, dots and the other half of every 128-byte line -- meet in one L2
    const int G = (int)gridDim.x, NX = min(G, 8), x = (int)blockIdx.x % NX, w = (int)blockIdx.x / NX;
    const int wx = G / NX + (x < G % NX ? 1 : 0);
    const int uq = units / NX, ur = units % NX;
    const int cnt = uq + (x < ur ? 1 : 0), base = x * uq + min(x, ur);
    if (w >= cnt) return;
    const int tid = threadIdx.x, sub = tid % LP, grp = tid / LP;
    constexpr int GI = LP >= 4 ? 1 : 2;
    int gk[GI];
    float bias[GI];
#pragma unroll
    for (int i = 0; i < GI; ++i) {
        gk[i] = LP >= 4 ? (sub & 3) : (sub & 1) + 2 * i;
        bias[i] = gb[gk[i]];
    }

    // ---- per-unit facts.  Header: offsets of the unit's graph (uniform -> scalar loads).
    struct Hdr { int g, slice, n0, Nn, e0, Ee, bad; };
    auto load_hdr = [&](int idx) {
        Hdr h;
        const int L = base + idx;
        h.g = L / S;
        h.slice = L - h.g * S;
        h.n0 = img_ptr[h.g];
        h.Nn = img_ptr[h.g + 1] - h.n0;
        h.e0 = img_ptr[B + 1 + h.g];
        h.Ee = img_ptr[B + 2 + h.g] - h.e0;
        // the host's promise about this graph (edge_csr(graphs=...)) does not hold: nothing may be parked.  Its outputs are
        // poisoned (below) and the unit runs as an empty graph.
        h.bad = (h.Ee > EMAX || h.Nn > NMAX || h.Nn > SL_NMAX) ? 1 : 0;
        return h;
    };
    // how the lane groups share the nodes of a graph: P groups per node (P a power of two, inside one wave)
    struct Own { int P, part, n, has_node; long col; };
    auto own_of = [&](const Hdr& h) {
        Own o;
        const int Nn = h.bad ? 0 : h.Nn;
        int P = 1;
        while (2 * P * LP <= 64 && 2 * P * Nn <= GROUPS) P *= 2;
        o.P = P;
        o.part = grp % P;
        o.n = grp / P;
        o.has_node = o.n < Nn;
        o.col = (long)h.slice * (PIECE / (int)sizeof(T)) + sub * CHL;
        return o;
    };

    // ---- prefetch registers
    int a_ob = 0, a_oe = 0;                                  // A: this lane group's node's out-list range (graph-local)
    u32x4 p_v = {0, 0, 0, 0};                                // B
    f32x4 p_nd = {0, 0, 0, 0};
    int p_ip = 0, p_in[INL];
    Piece16<T> row[U];                                       // C
    float de[U][GI];
    int on[U];
    int k0 = 0, k1 = 0;

    auto issue_A = [&](const Hdr& h, const Own& o) {
        const int* tab = img_ptr + 2 * (B + 1) + (long)h.g * (SL_NMAX + 2);
        const int nc = min(o.n, SL_NMAX);
        a_ob = tab[nc];
        a_oe = tab[nc + 1];
    };
    auto issue_B = [&](const Hdr& h, const Own& o) {
        const int Nn = h.bad ? 0 : h.Nn, Ee = h.bad ? 0 : h.Ee;
        if (grp < Nn) p_v = *reinterpret_cast<const u32x4*>(v + (long)(h.n0 + grp) * H + o.col);
        if (tid < Nn) p_nd = *reinterpret_cast<const f32x4*>(ndots + (long)(h.n0 + tid) * 4);
        if (tid <= Nn) p_ip = in_ptr[h.n0 + tid] - h.e0;
#pragma unroll
        for (int q = 0; q < INL; ++q) p_in[q] = (tid + q * SL_THREADS < Ee) ? in_ids[h.e0 + tid + q * SL_THREADS] - h.e0 : 0;
    };
    auto park_B = [&](const Hdr& h, int par) {
        const int Nn = h.bad ? 0 : h.Nn, Ee = h.bad ? 0 : h.Ee;
        if (grp < Nn) *reinterpret_cast<u32x4*>(sm_vs(par) + grp * PIECE + sub * 16) = p_v;
        if (tid < Nn) *reinterpret_cast<f32x4*>(sm_nd(par) + tid * 4) = p_nd;
        if (tid <= Nn) sm_iptr(par)[tid] = p_ip;
        unsigned short* in_loc = sm_inloc(par);
#pragma unroll
        for (int q = 0; q < INL; ++q)
            if (tid + q * SL_THREADS < Ee) in_loc[tid + q * SL_THREADS] = (unsigned short)p_in[q];
        for (int k = tid + INL * SL_THREADS; k < Ee; k += SL_THREADS) in_loc[k] = (unsigned short)(in_ids[h.e0 + k] - h.e0);
    };
    auto issue_rows = [&](const Hdr& h, const Own& o, int kb, int kend) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ec = h.e0 + max(min(kb + u, kend - 1), 0);
            row[u].r = *reinterpret_cast<const decltype(row[u].r)*>(e + (long)ec * H + o.col);
#pragma unroll
            for (int i = 0; i < GI; ++i) de[u][i] = edots[(long)ec * 4 + gk[i]];
            on[u] = so[2 * (long)ec + 1] - h.n0;
        }
    };
    auto range_C = [&](const Hdr& h, const Own& o) {         // this lane group's part of its node's out-list
        const int ob = o.has_node ? a_ob : 0, oe = o.has_node ? a_oe : 0;
        const int chunk = (oe - ob + o.P - 1) / o.P;
        k0 = ob + o.part * chunk;
        k1 = min(oe, k0 + chunk);
    };
    auto poison = [&](const Hdr& h, const Own& o) {
        float nanv[CHL];
#pragma unroll
        for (int j = 0; j < CHL; ++j) nanv[j] = __builtin_nanf("");
        for (int k = grp; k < h.Ee; k += GROUPS) Piece16<T>::store(e_in + (long)(h.e0 + k) * H + o.col, nanv);
        for (int k = grp; k < h.Nn; k += GROUPS) {
            Piece16<T>::store(ctx2 + (long)(h.n0 + k) * H + o.col, nanv);
            if (!sum_ctx) Piece16<T>::store(ctx2 + ((long)N + h.n0 + k) * H + o.col, nanv);
        }
    };

    // ---- prologue: the first unit's chain is exposed once per workgroup
    int idx = w, par = 0;
    Hdr hc = load_hdr(idx);
    Own oc = own_of(hc);
    Hdr hn = hc;                                             // header of unit u+1 (valid when idx + wx < cnt)
    if (idx + wx < cnt) hn = load_hdr(idx + wx);
    issue_A(hc, oc);
    issue_B(hc, oc);
    range_C(hc, oc);
    if (k0 < k1) issue_rows(hc, oc, k0, k1);

    for (;;) {
        const bool more = idx + wx < cnt;                    // uniform
        park_B(hc, par);                                     // into this unit's copy: stragglers of the last unit's phase 2 read the other
        if (hc.bad) poison(hc, oc);
        __syncthreads();                                     // (1) small arrays of this unit visible; stage / gin free again
        Own onx = oc;
        if (more) {
            onx = own_of(hn);
            issue_A(hn, onx);
        }
        const char* vs = sm_vs(par);
        const float* nd = sm_nd(par);
        // ---- phase 1: out-lists
        float out_sum[CHL];
#pragma unroll
        for (int j = 0; j < CHL; ++j) out_sum[j] = 0.f;
        if (oc.has_node) {
            const int n = oc.n;
            Piece16<T> vnp;
            vnp.r = *reinterpret_cast<const decltype(vnp.r)*>(vs + n * PIECE + sub * 16);
            float vn[CHL], acc[CHL];
            vnp.get(vn);
#pragma unroll
            for (int j = 0; j < CHL; ++j) acc[j] = 0.f;
            float ndn[GI];
#pragma unroll
            for (int i = 0; i < GI; ++i) ndn[i] = nd[n * 4 + gk[i]];
            for (int kb = k0; kb < k1; kb += U) {
                if (kb != k0) issue_rows(hc, oc, kb, k1);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int el = kb + u;
                    const bool live = el < k1;
                    float gate[GI];
#pragma unroll
                    for (int i = 0; i < GI; ++i) {
                        const float vd = (gk[i] & 1) ? nd[on[u] * 4 + gk[i]] : ndn[i];
                        gate[i] = sigmoidf_(vd + de[u][i] + bias[i]);
                    }
                    float gs, go, gx;
                    if constexpr (LP >= 4) {
                        gs = quad_bcast<0x00>(gate[0]);
                        go = quad_bcast<0x55>(gate[0]);
                        gx = quad_bcast<0xAA>(gate[0]);
                        if (live && sub == 3) gin[el] = gate[0];
                    } else {
                        gs = quad_bcast<0xA0>(gate[0]);
                        go = quad_bcast<0xF5>(gate[0]);
                        gx = quad_bcast<0xA0>(gate[1]);
                        if (live && sub == 1) gin[el] = gate[1];
                    }
                    if (live) {
                        *reinterpret_cast<decltype(row[u].r)*>(stage + (long)el * PIECE + sub * 16) = row[u].r;
                        Piece16<T> vop;
                        vop.r = *reinterpret_cast<const decltype(vop.r)*>(vs + on[u] * PIECE + sub * 16);
                        float xx[CHL], yy[CHL], rr[CHL];
                        row[u].get(xx);
                        vop.get(yy);
#pragma unroll
                        for (int j = 0; j < CHL; ++j) {
                            rr[j] = gs * vn[j] + go * yy[j];
                            acc[j] = fmaf(gx, xx[j], acc[j]);
                        }
                        Piece16<T>::store(e_in + (long)(hc.e0 + el) * H + oc.col, rr);
                    }
                    if (u & 1) __builtin_amdgcn_sched_barrier(0);   // two edges at a time: eight interleaved cost 40 more VGPRs than the budget has
                }
            }
            for (int off = LP; off < oc.P * LP; off <<= 1) {
#pragma unroll
                for (int j = 0; j < CHL; ++j) acc[j] += __shfl_xor(acc[j], off, 64);
            }
            if (sum_ctx) {
#pragma unroll
                for (int j = 0; j < CHL; ++j) out_sum[j] = acc[j];
            } else if (oc.part == 0) {
                Piece16<T>::store(ctx2 + (long)(hc.n0 + n) * H + oc.col, acc);
            }
        }
        // ---- the next unit's rows go out now and fly under phase 2; its small arrays are parked in the other copy
        Hdr hnn = hn;
        if (more) {
            issue_B(hn, onx);
            range_C(hn, onx);
            if (k0 < k1) issue_rows(hn, onx, k0, k1);
            if (idx + 2 * wx < cnt) hnn = load_hdr(idx + 2 * wx);
        }
        __syncthreads();                                     // (2) parked pieces + g_in of this unit visible
        // ---- phase 2: in-lists, from the parked pieces
        if (oc.has_node) {
            const int n = oc.n;
            const int* iptr = sm_iptr(par);
            const unsigned short* in_loc = sm_inloc(par);
            const int beg = iptr[n], end = iptr[n + 1], ch = (end - beg + oc.P - 1) / oc.P;
            const int j0 = beg + oc.part * ch, j1 = min(end, j0 + ch);
            float acc[CHL];
#pragma unroll
            for (int j = 0; j < CHL; ++j) acc[j] = 0.f;
            constexpr int U2 = 4;
            for (int kb = j0; kb < j1; kb += U2) {
                int el[U2];
#pragma unroll
                for (int u = 0; u < U2; ++u) el[u] = in_loc[min(kb + u, j1 - 1)];
                Piece16<T> rw[U2];
                float gv[U2];
#pragma unroll
                for (int u = 0; u < U2; ++u) {
                    rw[u].r = *reinterpret_cast<const decltype(rw[u].r)*>(stage + (long)el[u] * PIECE + sub * 16);
                    gv[u] = (kb + u < j1) ? gin[el[u]] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < U2; ++u) {
                    float xx[CHL];
                    rw[u].get(xx);
#pragma unroll
                    for (int j = 0; j < CHL; ++j) acc[j] = fmaf(gv[u], xx[j], acc[j]);
                }
            }
            for (int off = LP; off < oc.P * LP; off <<= 1) {
#pragma unroll
                for (int j = 0; j < CHL; ++j) acc[j] += __shfl_xor(acc[j], off, 64);
            }
            if (sum_ctx) {
#pragma unroll
                for (int j = 0; j < CHL; ++j) acc[j] += out_sum[j];
                if (oc.part == 0) Piece16<T>::store(ctx2 + (long)(hc.n0 + n) * H + oc.col, acc);
            } else if (oc.part == 0) {
                Piece16<T>::store(ctx2 + ((long)N + hc.n0 + n) * H + oc.col, acc);
            }
        }
        if (!more) break;
        idx += wx;
        par ^= 1;
        hc = hn;
        oc = onx;
        hn = hnn;
    }
}

}  // namespace

#define SGG_DISPATCH_T(dtype, CALL_BF16, CALL_F32) \
    if ((dtype) == SGG_BF16) { CALL_BF16; }        \
    else if ((dtype) == SGG_F32) { CALL_F32; }     \
    else return SGG_ERR_DTYPE;

extern "C" int sgg_imp_node_gate_dots(const void* v, int N, int H, const float* gate_w, float* dots, int dtype, void* stream) {
    if (N == 0) return SGG_OK;
    if (!v || !gate_w || !dots || N < 0 || H <= 0 || (H & 7) || H > MAXH) return SGG_ERR_ARG;
    const dim3 grid((N + 3) / 4), blk(256);
    hipStream_t s = (hipStream_t)stream;
    SGG_DISPATCH_T(dtype,
        hipLaunchKernelGGL(node_gate_dots_kernel<bf16_t>, grid, blk, 0, s, (const bf16_t*)v, N, H, gate_w, dots),
        hipLaunchKernelGGL(node_gate_dots_kernel<float>, grid, blk, 0, s, (const float*)v, N, H, gate_w, dots));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_imp_edge_ctx_fwd(const void* v, const void* e, const int64_t* rel_inds, int E, int H,
                                    const float* node_dots, const float* gate_w, const float* gate_b, void* e_in,
                                    float* gates, int dtype, void* stream) {
    if (E == 0) return SGG_OK;
    if (!v || !e || !rel_inds || !node_dots || !gate_w || !gate_b || !e_in || !gates || E < 0 || H <= 0 || (H & 7) || H > MAXH)
        return SGG_ERR_ARG;
    const dim3 grid((E + 3) / 4), blk(256);
    hipStream_t s = (hipStream_t)stream;
    SGG_DISPATCH_T(dtype,
        hipLaunchKernelGGL(edge_ctx_kernel<bf16_t>, grid, blk, 0, s, (const bf16_t*)v, (const bf16_t*)e, rel_inds, E, H, node_dots, gate_w, gate_b, (bf16_t*)e_in, gates),
        hipLaunchKernelGGL(edge_ctx_kernel<float>, grid, blk, 0, s, (const float*)v, (const float*)e, rel_inds, E, H, node_dots, gate_w, gate_b, (float*)e_in, gates));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_imp_node_scatter_fwd(const void* e, const float* gates, const int* out_ptr, const int* out_ids,
                                        const int* in_ptr, const int* in_ids, int N, int H, void* ctx, int dtype,
                                        void* stream) {
    if (N == 0) return SGG_OK;
    if (!e || !gates || !out_ptr || !out_ids || !in_ptr || !in_ids || !ctx || N < 0 || H <= 0 || (H & 7) || H > MAXH)
        return SGG_ERR_ARG;
    const dim3 grid(N), blk(256);
    hipStream_t s = (hipStream_t)stream;
    SGG_DISPATCH_T(dtype,
        hipLaunchKernelGGL(node_scatter_kernel<bf16_t>, grid, blk, 0, s, (const bf16_t*)e, gates, out_ptr, out_ids, in_ptr, in_ids, H, (bf16_t*)ctx),
        hipLaunchKernelGGL(node_scatter_kernel<float>, grid, blk, 0, s, (const float*)e, gates, out_ptr, out_ids, in_ptr, in_ids, H, (float*)ctx));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_gru_gate_fwd(const void* gi, const void* gh, const float* b_hh, const void* h_prev, void* h_out, int M,
                                int H, const float* dot_w, int dot_ld, float* dots, int g_dtype, int dtype, void* stream) {
    if (M == 0) return SGG_OK;
    if (!gi || !h_out || M < 0 || H <= 0 || (H & 7)) return SGG_ERR_ARG;
    if (gh ? !h_prev : !b_hh) return SGG_ERR_ARG;
    if (dots) {   // the dot epilogue reduces over the H/8 lanes of a row with xor-shuffles
        const int h8 = H / 8;
        if (!dot_w || dot_ld < H || h8 > 64 || (h8 & (h8 - 1))) return SGG_ERR_ARG;
    }
    const long total = (long)M * (H / 8);
    const dim3 grid((unsigned)((total + 255) / 256)), blk(256);
    hipStream_t s = (hipStream_t)stream;
    if (g_dtype == SGG_F32 && dtype == SGG_BF16)
        hipLaunchKernelGGL((gru_gate_kernel<float, bf16_t>), grid, blk, 0, s, (const float*)gi, (const float*)gh, b_hh, (const bf16_t*)h_prev, (bf16_t*)h_out, total, H, dot_w, dot_ld, dots);
    else if (g_dtype == SGG_F32 && dtype == SGG_F32)
        hipLaunchKernelGGL((gru_gate_kernel<float, float>), grid, blk, 0, s, (const float*)gi, (const float*)gh, b_hh, (const float*)h_prev, (float*)h_out, total, H, dot_w, dot_ld, dots);
    else if (g_dtype == SGG_BF16 && dtype == SGG_BF16)
        hipLaunchKernelGGL((gru_gate_kernel<bf16_t, bf16_t>), grid, blk, 0, s, (const bf16_t*)gi, (const bf16_t*)gh, b_hh, (const bf16_t*)h_prev, (bf16_t*)h_out, total, H, dot_w, dot_ld, dots);
    else
        return SGG_ERR_DTYPE;
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

namespace {
__global__ void graph_ptr_kernel(const int64_t* __restrict__ im, int N, int B, const int* __restrict__ out_ptr,
                                 int* __restrict__ ptr) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b > B) return;
    int lo = 0, hi = N;            // first node with im >= b
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (im[mid] < b) lo = mid + 1;
        else hi = mid;
    }
    ptr[b] = lo;
    const int e0 = out_ptr[lo];
    ptr[B + 1 + b] = e0;           // first edge of graph b when the edge list is sorted by (graph, subject)
    if (b == B) return;
    // graph-relative out-list offsets of the graph's nodes at an address that depends on b only: the sliced kernel loads them
    // together with ptr[b] instead of after it.  Entries past the graph's last node repeat its edge count.
    int nxt = lo;
    {
        int l2 = lo, h2 = N;
        while (l2 < h2) {
            const int mid = (l2 + h2) >> 1;
            if (im[mid] < b + 1) l2 = mid + 1;
            else h2 = mid;
        }
        nxt = l2;
    }
    int* tab = ptr + 2 * (B + 1) + (long)b * (SL_NMAX + 2);
    for (int n = 0; n < SL_NMAX + 2; ++n) tab[n] = out_ptr[min(lo + n, nxt)] - e0;
}

template <typename T, int LP>
int launch_sliced(const void* v, const void* e, const int* so, const int* out_ptr, const int* in_ptr, const int* in_ids,
                  const int* img_ptr, int B, int N, int H, const float* ndots, const float* edots, const float* gb, void* e_in,
                  void* ctx2, int max_edges, int sum_ctx, hipStream_t s) {
    auto k = imp_sliced_kernel<T, LP>;
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                slice_lds_bytes<LP>(SliceCfg<LP>::EMAX)) != hipSuccess)
            return SGG_ERR_LAUNCH;
        configured = true;
    }
    // the staging area is sized for THIS batch's largest graph: smaller graphs leave room for a second workgroup on the CU
    const int emax = (max(max_edges, 8) + 7) & ~7;
    const int S = H * (int)sizeof(T) / (LP * 16);
    hipLaunchKernelGGL(k, dim3(B * S), dim3(SL_THREADS), slice_lds_bytes<LP>(emax), s, (const T*)v, (const T*)e, so, out_ptr, in_ptr,
                       in_ids, img_ptr, B, N, H, ndots, edots, gb, (T*)e_in, (T*)ctx2, emax, sum_ctx);
    return hipGetLastError() == hipSuccess ? SGG_OK : SGG_ERR_LAUNCH;
}

// ---------------------------------------------------------------------------------------------------------------------
// Persistent LDS-DMA kernels (the read stream of the split step, imp_ctx_kernel below).
//
// What bounds the step (measured on MI355X, DESIGN.md "IMP step"):
//   * the REQUEST size.  A workgroup that owns a W-byte column slice of a graph's edge rows moves W bytes per memory request;
//     the chip retires about the same number of requests per second whatever their size, so 64-byte pieces (half a cache
//     line) stream at 3.3 TB/s, 128-byte pieces (full lines) at 5.3 TB/s, 256-byte pieces at 5.6 TB/s
//     (tools/exp/piece_bw.hip).  The short-lived kernel above (64-byte pieces) is AT its pattern's ceiling.
//   * VALU instructions: every wave of a 16-wave workgroup walks the same instruction stream, 4 cycles per wave instruction
//     and SIMD, four waves per SIMD.
// A one-kernel persistent form with 128-byte pieces (imp_dma_kernel: two 496-edge batches per unit, gates once per unit,
// small arrays prefetched into registers by asm loads) measured 70-85 us at B=128 and was removed: its asm loads returned into
// registers that the compiler believed defined at issue -- legal only while the register allocator never copies them, which
// it started to do as soon as register pressure rose (DESIGN.md "IMP step").
//
// The DMA is issued from inline asm on purpose: hipcc treats a known LDS-DMA as an LDS write that may alias every later LDS
// read and puts `s_waitcnt vmcnt(0)` in front of each, which would serialise a chunk's compute behind the next chunk's DMA.
constexpr int DM_THREADS = 1024;
constexpr int DM_EMAX = DM_THREADS, DM_NMAX = SL_NMAX;     // one edge per thread for the coalesced fetches
constexpr int DM_LDS_MAX = 160 * 1024;

typedef __attribute__((address_space(3))) char lds_char_t;

// 64 lanes x 16 bytes from per-lane global addresses into 1 KiB of LDS at `lds_base` (wave-uniform)
__device__ __forceinline__ void dma16_to_lds(const void* gptr, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_base), "v"(gptr) : "memory");
}

#ifndef SGG_DMA_ABL
#define SGG_DMA_ABL 0     // kernel experiments only (imp_ctx_kernel): 3 no DMA after the first chunk, 4 DMA and barriers only (no sums), 5 no edge inputs (STEP)
#endif
#ifdef SGG_DMA_TICKS     // kernel experiments only: clock stamps of the first 8 workgroups (one per XCD), wave 0, 16 units x 8 stamps
__device__ long long g_dma_ticks[8 * 16 * 8];
#define SGG_DTICK(i)                                                                                          \
    if (blockIdx.x < 8 && threadIdx.x == 0 && unit_no < 16) g_dma_ticks[(blockIdx.x * 16 + unit_no) * 8 + (i)] = clock64();
#else
#define SGG_DTICK(i)
#endif

// ---------------------------------------------------------------------------------------------------------------------
// The node half of the step on its own (imp_ctx_kernel): ctx_out[n] = sum over n's out-edges of g_out e, ctx_in[n] = sum over its
// in-edges of g_in e -- a READ stream (every edge row once) with two small outputs.  The unit is (graph, PIECE-byte slice of the
// rows), the slice's pieces staged in LDS so that the in-lists can walk them:
//  * the staging is a RING of NBUF buffers of EB consecutive edges and the DMA runs NBUF-1 chunks ahead of the compute, ACROSS unit
//    boundaries (a chunk is an edge range: its extent follows from the graph header alone);
//  * NOTHING a load returns lives in a register across other code: the per-unit small arrays (the two gates of every edge --
//    made by the write stream, imp_edge_in_kernel -- the in-list entries, the list offsets) go global -> LDS by DMA as well, into
//    one of two sets, one unit ahead.  (Asm loads into registers that the compiler believes defined at issue are only safe while
//    the register allocator never copies such a register before the data lands; under pressure it does -- seen here as wrong
//    sums and memory faults that came and went with unrelated code changes.);
//  * out-lists: a chunk's rows belong to few nodes (8 of 32 on a complete 32-node graph), so ALL lane groups share those nodes
//    (Pc parts per node), the parts meet in the wave and one lane group per node and chunk adds to the node's running sum in
//    LDS; in-lists: a lane group owns (node, part) for the whole unit and walks its entries with a cursor (ascending edge ids).
// Waits are counted: a wave tracks how many vector-memory operations it has issued (`ci`, exact for DMA, stores not counted: an
// under-count) and the value of that counter after each thing it will wait for; `s_waitcnt vmcnt(ci - mark)` returns as soon as
// that thing -- and, in issue order, everything older -- has landed, whatever was issued later.
constexpr int CX_NBUF = 4, CX_MAXCHUNKS = 64, CX_PTRS = 128;
constexpr int ctx_epad(int emax) { return (emax + 63) & ~63; }
// LDS beside the ring.  osb: bytes of one node's running out-sum (its piece as f32)
constexpr int ctx_fixed_bytes(int emax, int nmax, int osb) {
    return 2 * (3 * ctx_epad(emax) * 4 + 2 * CX_PTRS * 4)      // two sets of: g_out, g_in, in-list entries (per edge), out- / in-list offsets
           + 2 * CX_MAXCHUNKS * 4 + nmax * osb;                 // node range of every chunk, running out-sums
}
// edges per chunk: a multiple of the edges one DMA instruction moves (64 lanes x 16 bytes)
constexpr int ctx_chunk_edges(int emax, int nmax, int osb, int piece) {
    const int epw = 1024 / piece;
    const int room = (DM_LDS_MAX - ctx_fixed_bytes(emax, nmax, osb)) / (CX_NBUF * piece);
    const int eb = room < emax ? room / epw * epw : (emax + epw - 1) / epw * epw;
    return eb < 16 ? 0 : eb;
}
static_assert(ctx_chunk_edges(992, 32, 256, 128) >= 248, "imp_ctx_kernel: a 992-edge graph goes through in four chunks of 128-byte pieces");
static_assert(ctx_chunk_edges(992, 32, 128, 64) >= 496, "imp_ctx_kernel: ... and in two of 64-byte pieces");

// 64 lanes x 4 bytes from per-lane global addresses into 256 bytes of LDS at `lds_base` (wave-uniform)
__device__ __forceinline__ void dma4_to_lds(const void* gptr, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(lds_base), "v"(gptr) : "memory");
}

template <typename T, int LP>
__global__ __launch_bounds__(DM_THREADS) void imp_ctx_kernel(
    const T* __restrict__ e, const float* __restrict__ gates_oi, const int* __restrict__ in_ptr, const int* __restrict__ in_ids,
    const int* __restrict__ img_ptr, int B, int N, int H, T* __restrict__ ctx2, int EMAX, int NMAX, int EB, int sum_ctx) {
    constexpr int PIECE = LP * 16, CHL = 16 / (int)sizeof(T), CHAN = PIECE / (int)sizeof(T), GROUPS = DM_THREADS / LP, EPW = 64 / LP;
    constexpr int NBUF = CX_NBUF, U = 2, UO = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int EPAD = ctx_epad(EMAX), SETW = 3 * EPAD + 2 * CX_PTRS;                  // words of one set of small arrays
    char* const ring = smem;                                                          // [NBUF][EB][PIECE]
    int* const sets = reinterpret_cast<int*>(ring + (long)NBUF * EB * PIECE);        // [2][SETW]: g_out[EPAD] g_in[EPAD] in[EPAD] optr[] iptr[]
    int* const cna = sets + 2 * SETW;                                                 // [CX_MAXCHUNKS] first / last node whose out-list
    int* const cnb = cna + CX_MAXCHUNKS;                                              //                touches the chunk
    float* const osum = reinterpret_cast<float*>(cnb + CX_MAXCHUNKS);                 // [NMAX][CHAN] running out-sums (f32)
    const unsigned ring_lds = (unsigned)(unsigned long)(lds_char_t*)smem;
    const unsigned sets_lds = ring_lds + (unsigned)(NBUF * EB * PIECE);

    const int S = H * (int)sizeof(T) / PIECE, units = B * S;
    const int G = (int)gridDim.x, NX = min(G, 8), x = (int)blockIdx.x % NX, w = (int)blockIdx.x / NX;
    const int wx = G / NX + (x < G % NX ? 1 : 0);
    const int uq = units / NX, ur = units % NX;
    const int cnt = uq + (x < ur ? 1 : 0), base = x * uq + min(x, ur);
    if (w >= cnt) return;
    const int nunits = (cnt - w + wx - 1) / wx;                 // units of this workgroup: ordinals 0 .. nunits-1, unit index w + k wx
    const int tid = threadIdx.x, sub = tid % LP, grp = tid / LP, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    struct Hdr { int g, slice, n0, Nn, e0, Ee; };              // Ee < 0: the host's promise about this graph is broken (outputs poisoned)
    int lg = base / S, lslice = base - lg * S, lidx = 0;
    auto load_hdr = [&](int k) __attribute__((always_inline)) { // header of ordinal k (loaded in increasing order)
        Hdr h;
        const int idx = w + k * wx;
        lslice += idx - lidx;
        lidx = idx;
        while (lslice >= S) {
            lslice -= S;
            ++lg;
        }
        h.g = lg;
        h.slice = lslice;
        u32x2 nn, ee;
        asm volatile("s_load_dwordx2 %0, %2, 0x0\n\ts_load_dwordx2 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(nn), "=&s"(ee) : "s"(img_ptr + h.g), "s"(img_ptr + (B + 1 + h.g)) : "memory");
        h.n0 = (int)nn.x;
        h.Nn = (int)nn.y - h.n0;
        h.e0 = (int)ee.x;
        h.Ee = (int)ee.y - h.e0;
        if (h.Ee > EMAX || h.Nn > NMAX || h.Nn > SL_NMAX) {
            h.Nn = 0;
            h.Ee = -1;
        }
        return h;
    };
    auto col_of = [&](const Hdr& h) __attribute__((always_inline)) { return (long)h.slice * CHAN + sub * CHL; };
    int ci = 0;                                                  // DMA operations this wave has issued
    // rows [lo, hi) (graph-local) of h's slice -> ring buffer rb
    auto issue_dma = [&](const Hdr& h, int lo, int hi, int rb) __attribute__((always_inline)) {
        const int nch = (hi - lo + EPW - 1) / EPW;
        const char* src = reinterpret_cast<const char*>(e + col_of(h));
        for (int c = wv; c < nch; c += DM_THREADS / 64) {
            const int el = min(lo + c * EPW + lane / LP, hi - 1);
            dma16_to_lds(src + (long)(h.e0 + el) * H * (int)sizeof(T),
                         __builtin_amdgcn_readfirstlane(ring_lds + (unsigned)(rb * EB * PIECE + c * 1024)));
            ++ci;
        }
    };
    auto wait_mark = [&](int mark) __attribute__((always_inline)) {   // everything issued up to `mark` has landed
        switch (min(ci - mark, 15)) {
#define SGG_W(K) case K: asm volatile("s_waitcnt vmcnt(" #K ")" ::: "memory"); break;
            SGG_W(0) SGG_W(1) SGG_W(2) SGG_W(3) SGG_W(4) SGG_W(5) SGG_W(6) SGG_W(7) SGG_W(8) SGG_W(9) SGG_W(10) SGG_W(11) SGG_W(12)
            SGG_W(13) SGG_W(14)
#undef SGG_W
            default: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        }
    };
    // the small arrays of h's graph -> set q: lane t of the workgroup <-> edge t / list offset t
    auto issue_small = [&](const Hdr& h, int q) __attribute__((always_inline)) {
        const unsigned set_lds = sets_lds + (unsigned)(q * SETW * 4);
        if (wv * 64 < h.Ee) {
            const long et = h.e0 + min(tid, h.Ee - 1);
            dma4_to_lds(gates_oi + 2 * et, __builtin_amdgcn_readfirstlane(set_lds + (unsigned)(wv * 256)));
            dma4_to_lds(gates_oi + 2 * et + 1, __builtin_amdgcn_readfirstlane(set_lds + (unsigned)((EPAD + wv * 64) * 4)));
            dma4_to_lds(in_ids + et, __builtin_amdgcn_readfirstlane(set_lds + (unsigned)((2 * EPAD + wv * 64) * 4)));
            ci += 3;
        }
        if (wv * 64 <= h.Nn && h.Ee >= 0) {
            const int t = min(tid, h.Nn);
            dma4_to_lds(img_ptr + (2 * (B + 1) + h.g * (SL_NMAX + 2) + t), __builtin_amdgcn_readfirstlane(set_lds + (unsigned)((3 * EPAD + wv * 64) * 4)));
            dma4_to_lds(in_ptr + (h.n0 + t), __builtin_amdgcn_readfirstlane(set_lds + (unsigned)((3 * EPAD + CX_PTRS + wv * 64) * 4)));
            ci += 2;
        }
    };
    auto poison = [&](const Hdr& h) __attribute__((always_inline)) {
        u32x2 nn;
        asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(nn) : "s"(img_ptr + __builtin_amdgcn_readfirstlane(h.g)) : "memory");
        const int n0 = (int)nn.x, Nn = (int)nn.y - n0;
        const long col = col_of(h);
        float nanv[CHL];
#pragma unroll
        for (int j = 0; j < CHL; ++j) nanv[j] = __builtin_nanf("");
        for (int k = grp; k < Nn; k += GROUPS) {
            Piece16<T>::store(ctx2 + (long)(n0 + k) * H + col, nanv);
            if (!sum_ctx) Piece16<T>::store(ctx2 + ((long)N + n0 + k) * H + col, nanv);
        }
    };
    auto chunks_of = [&](const Hdr& h) __attribute__((always_inline)) { return h.Ee > 0 ? (h.Ee + EB - 1) / EB : 1; };

    // ---- headers of the consumer's unit and the two after it (the producer runs at most NBUF-1 chunks ahead)
    Hdr h0 = load_hdr(0), h1 = h0, h2 = h0;
    if (nunits > 1) h1 = load_hdr(1);
    if (nunits > 2) h2 = load_hdr(2);
    int loaded = min(nunits, 3);
    // producer position: unit ordinal pu, chunk pb; ring slot pr
    int pu = 0, pb = 0, pr = 0, issued_chunks = 0;
    int mk0 = 0, mk1 = 0, mk2 = 0, mk3 = 0;                      // ci after the DMA of the chunk in ring slot r was issued
    auto mark_of = [&](int r) __attribute__((always_inline)) { return r == 0 ? mk0 : r == 1 ? mk1 : r == 2 ? mk2 : mk3; };
    int cu = 0;                                                  // consumer unit ordinal
    auto hdr_rel = [&](int rel) __attribute__((always_inline)) {   // field by field: a selected struct copy would put the headers in scratch
        Hdr h;
        h.g = rel == 0 ? h0.g : rel == 1 ? h1.g : h2.g;
        h.slice = rel == 0 ? h0.slice : rel == 1 ? h1.slice : h2.slice;
        h.n0 = rel == 0 ? h0.n0 : rel == 1 ? h1.n0 : h2.n0;
        h.Nn = rel == 0 ? h0.Nn : rel == 1 ? h1.Nn : h2.Nn;
        h.e0 = rel == 0 ? h0.e0 : rel == 1 ? h1.e0 : h2.e0;
        h.Ee = rel == 0 ? h0.Ee : rel == 1 ? h1.Ee : h2.Ee;
        return h;
    };
    auto produce = [&]() __attribute__((always_inline)) {       // issue the DMA of the next chunk, if there is one within reach
        if (pu >= nunits || pu - cu > 2) return false;
        const Hdr hp = hdr_rel(pu - cu);
        const int lo = pb * EB, hi = min(lo + EB, max(hp.Ee, 0));
        if (SGG_DMA_ABL != 3 || issued_chunks == 0) issue_dma(hp, lo, hi, pr);
        mk0 = pr == 0 ? ci : mk0;
        mk1 = pr == 1 ? ci : mk1;
        mk2 = pr == 2 ? ci : mk2;
        mk3 = pr == 3 ? ci : mk3;
        pr = pr + 1 == NBUF ? 0 : pr + 1;
        ++issued_chunks;
        if (++pb >= chunks_of(hp)) {
            pb = 0;
            ++pu;
        }
        return true;
    };
    int mark_small;
    issue_small(h0, 0);
    mark_small = ci;
    for (int d = 0; d < NBUF - 1; ++d) produce();
    for (int k = tid; k < NMAX * CHAN; k += DM_THREADS) osum[k] = 0.f;   // published by the first unit's barrier (B)
    if (tid < CX_MAXCHUNKS) {
        cna[tid] = 0x7fffffff;
        cnb[tid] = -1;
    }
    int cr = 0, consumed = 0;                                    // consumer ring slot, chunks consumed

    for (cu = 0; cu < nunits; ++cu) {
        const Hdr hc = h0;
        const int q = cu & 1;
        const float* const g_out = reinterpret_cast<const float*>(sets + q * SETW);
        const float* const g_in = g_out + EPAD;
        const int* const in_raw = sets + q * SETW + 2 * EPAD;    // global edge ids
        const int* const optr = in_raw + EPAD;                   // graph-local
        const int* const iptr = optr + CX_PTRS;                  // global positions
        // ---- unit start: its small arrays and first chunk have landed (this wave's parts)
        { [[maybe_unused]] const int unit_no = consumed; SGG_DTICK(0) }
        wait_mark(max(mark_small, mark_of(cr)));
        if (hc.Ee < 0) poison(hc);
        __syncthreads();                                         // (B) ... and every wave's
        if (tid < hc.Nn) {                                       // thread t: node t tells the chunks its out-list touches
            const int a = optr[tid], bnd = optr[tid + 1];
            if (bnd > a)
                for (int c = a / EB; c <= (bnd - 1) / EB; ++c) {
                    atomicMin(&cna[c], tid);
                    atomicMax(&cnb[c], tid);
                }
        }
        __syncthreads();                                         // (C) chunk -> node ranges
        if (cu + 1 < nunits) {
            issue_small(h1, q ^ 1);                              // the next unit's: that set was the last unit's (all waves passed its E)
            mark_small = ci;
        }
        // in-lists: a lane group owns (node n, part) for the whole unit
        int P = 1, lgP = 0;
        while (2 * P * LP <= 64 && 2 * P * hc.Nn <= GROUPS) {
            P *= 2;
            ++lgP;
        }
        const int part = grp & (P - 1), n = grp >> lgP;
        const bool has = n < hc.Nn;
        const long col = col_of(hc);
        float acc_i[CHL];
#pragma unroll
        for (int j = 0; j < CHL; ++j) acc_i[j] = 0.f;
        int ib = 0, mine_i = 0, ki = 0;
        if (has) {
            ib = iptr[n] - hc.e0 + part;
            mine_i = (iptr[n + 1] - hc.e0 - ib + P - 1) >> lgP;
        }
        const int nchunks = chunks_of(hc);
        for (int cb = 0; cb < nchunks; ++cb) {
            const int blo = cb * EB, bhi = min(blo + EB, max(hc.Ee, 0));
            [[maybe_unused]] const int unit_no = consumed;
            if (cb > 0) {
                SGG_DTICK(0)
                wait_mark(mark_of(cr));                          // this chunk's DMA (my part of it)
                __syncthreads();                                 // (D) chunk visible; every wave has left the last chunk
            }
            SGG_DTICK(1)
            while (issued_chunks - consumed < NBUF && produce()) {}   // the ring slot of the last chunk is free: one more chunk ahead
            SGG_DTICK(2)
            const int soff = (cr * EB - blo) * PIECE + sub * 16; // this lane's 16 bytes of graph-local edge el: ring + soff + el * PIECE
            if (SGG_DMA_ABL != 4 && bhi > blo) {
                // ---- out-lists: the chunk's rows belong to the nodes na .. nb
                const int na = __builtin_amdgcn_readfirstlane(cna[cb]), nb = __builtin_amdgcn_readfirstlane(cnb[cb]);
                int Pc = 1, lgPc = 0;
                while (2 * Pc * LP <= 64 && 2 * Pc * (nb - na + 1) <= GROUPS) {
                    Pc *= 2;
                    ++lgPc;
                }
                for (int nc = na + (grp >> lgPc); nc <= nb; nc += GROUPS >> lgPc) {
                    const int pc = grp & (Pc - 1);
                    const int a = max(optr[nc], blo) + pc, bnd = min(optr[nc + 1], bhi);
                    float acc_o[CHL];
#pragma unroll
                    for (int j = 0; j < CHL; ++j) acc_o[j] = 0.f;
                    for (int el0 = a; el0 < bnd; el0 += UO << lgPc) {
                        Piece16<T> rowp[UO];
                        float gz[UO];
#pragma unroll
                        for (int u = 0; u < UO; ++u) {
                            const int el = min(el0 + (u << lgPc), bhi - 1);
                            gz[u] = el0 + (u << lgPc) < bnd ? g_out[el] : 0.f;
                            rowp[u].r = *reinterpret_cast<const decltype(rowp[u].r)*>(ring + (soff + el * PIECE));
                        }
#pragma unroll
                        for (int u = 0; u < UO; ++u) {
                            float xx[CHL];
                            rowp[u].get(xx);
#pragma unroll
                            for (int j = 0; j < CHL; ++j) acc_o[j] = fmaf(gz[u], xx[j], acc_o[j]);
                        }
                    }
                    for (int off = LP; off < Pc * LP; off <<= 1) {
#pragma unroll
                        for (int j = 0; j < CHL; ++j) acc_o[j] += __shfl_xor(acc_o[j], off, 64);
                    }
                    if (pc == 0) {
                        float* dst = osum + nc * CHAN + sub * CHL;
#pragma unroll
                        for (int j = 0; j < CHL; j += 4) {
                            f32x4 t = *reinterpret_cast<const f32x4*>(dst + j);
                            t.x += acc_o[j];
                            t.y += acc_o[j + 1];
                            t.z += acc_o[j + 2];
                            t.w += acc_o[j + 3];
                            *reinterpret_cast<f32x4*>(dst + j) = t;
                        }
                    }
                }
                SGG_DTICK(3)
                // ---- in-list entries inside [blo, bhi) (ascending edge ids: a cursor)
                if (has) {
                    for (;;) {
                        int done = 0;
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const int el = in_raw[ib + (min(ki + u, max(mine_i - 1, 0)) << lgP)] - hc.e0;
                            if (ki + u < mine_i && el < bhi && done == u) {
                                const float gw = g_in[el];
                                Piece16<T> rw;
                                rw.r = *reinterpret_cast<const decltype(rw.r)*>(ring + (soff + el * PIECE));
                                float xx[CHL];
                                rw.get(xx);
#pragma unroll
                                for (int j = 0; j < CHL; ++j) acc_i[j] = fmaf(gw, xx[j], acc_i[j]);
                                ++done;
                            }
                        }
                        ki += done;
                        if (__builtin_amdgcn_ballot_w64(done == U) == 0) break;
                    }
                }
            }
            SGG_DTICK(4)
            cr = cr + 1 == NBUF ? 0 : cr + 1;
            ++consumed;
        }
        { [[maybe_unused]] const int unit_no = consumed - 1; SGG_DTICK(5) }
        __syncthreads();                                         // (E) every wave has left the unit: the running out-sums are final
        if (tid < CX_MAXCHUNKS) {                                // for the next unit (read again only after its barrier C)
            cna[tid] = 0x7fffffff;
            cnb[tid] = -1;
        }
        // ---- unit done: the P in-sum parts of a node meet; the node's owner takes (and clears) its out-sum; one store per sum
        if (has) {
            for (int off = LP; off < P * LP; off <<= 1) {
#pragma unroll
                for (int j = 0; j < CHL; ++j) acc_i[j] += __shfl_xor(acc_i[j], off, 64);
            }
            if (part == 0) {
                float acc_o[CHL];
                float* src = osum + n * CHAN + sub * CHL;
#pragma unroll
                for (int j = 0; j < CHL; j += 4) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(src + j);
                    acc_o[j] = t.x;
                    acc_o[j + 1] = t.y;
                    acc_o[j + 2] = t.z;
                    acc_o[j + 3] = t.w;
                    *reinterpret_cast<f32x4*>(src + j) = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                if (sum_ctx) {
#pragma unroll
                    for (int j = 0; j < CHL; ++j) acc_o[j] += acc_i[j];
                    Piece16<T>::store(ctx2 + (long)(hc.n0 + n) * H + col, acc_o);
                } else {
                    Piece16<T>::store(ctx2 + (long)(hc.n0 + n) * H + col, acc_o);
                    Piece16<T>::store(ctx2 + ((long)N + hc.n0 + n) * H + col, acc_i);
                }
            }
        }
        { [[maybe_unused]] const int unit_no = consumed - 1; SGG_DTICK(6) }
        // headers slide: the consumer moves to the next unit
        h0 = h1;
        h1 = h2;
        if (loaded < nunits) {
            h2 = load_hdr(loaded);
            ++loaded;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The read stream as a block-sparse matrix product on the matrix cores (imp_ctx_mfma_kernel; bf16, graphs of <= 32 nodes).
//   ctx_out[n, :] = sum_e G_out[n, e] R[e, :],  G_out[n, e] = g_out(e) [s(e) == n]        ctx_in likewise with g_in, o(e)
// R = the unit's row pieces [edges x 64 channels] in the LDS ring exactly as imp_ctx_kernel stages them (same DMA ring, same counted
// waits, the small arrays by DMA one unit ahead); the VALU gather / unpack / FMA / cross-lane reduction loops -- ~340 wave
// instructions per 248-row chunk, which is what bounds that kernel -- become v_mfma_f32_16x16x32_bf16: K = 32 edges per step, the
// B fragment (edges x 16 channels: the reduction index is the SLOW axis in LDS) comes out through ds_read_b64_tr_b16, the A fragment
// (16 nodes x 32 edges of the gate matrix) is built in registers from the edge's gate and node id: 8 compares + selects per lane.
// No lists are walked: graph structure enters only through (s, o) of every edge, so any edge order inside a graph works.
// Roles of the 16 waves: (sum: out / in) x (node tile: 0-15 / 16-31) x (K quarter: every fourth 32-edge step); a wave builds each
// of its A fragments ONCE and multiplies it with all four 16-channel tiles (four 16x16 accumulators, kept for the whole unit); at
// the end of the unit the K quarters (and, for the summed ctx of the training step, the two sums) meet through the ring slot that
// was consumed last.
// Measured at B=128 (127 MB of rows): 40 us -- the list-walking kernel above: 51, DMA and barriers alone: 31-32.  On the way:
// K halves x channel halves as wave roles (every A fragment built by two waves), f32 gates and 32-bit node ids read by every lane
// (64 bytes per lane and K step, 16 lanes reading the same bytes), swizzle key (row >> 1) & 7 (two-way conflicts of the transposing
// reads): 46-48 us, bound by the LDS pipe (3000 of its cycles per 256-row chunk).  What is left at 40 us: ~110 wave instructions per
// wave and chunk (VALU issue, 1800 cycles per chunk), 3 barriers + 2 LDS round trips at the end of every unit (3700 cycles), the
// conversion pass at its start.
// Gates enter the product as bf16 (relative error <= 2^-9 per term, below the bf16 rounding of the output): a pass at the start of a
// unit turns the raw per-edge arrays (f32 gates, 32-bit global node ids, DMA'd one unit ahead) into bf16 gates and byte node ids.
// Rows are swizzled on the DMA's source side, slot' = slot ^ mf_key(row), so that the 16 rows one transposing read touches spread
// over the banks (with the plain key (row >> 1) & 7 they fell on 4 keys: two-way conflicts, 64 instead of 32 LDS cycles per K step).
typedef __attribute__((ext_vector_type(8))) __bf16 mf_bf16x8;
typedef __attribute__((ext_vector_type(4))) short mf_s16x4;
typedef __attribute__((ext_vector_type(8))) short mf_s16x8;
typedef __attribute__((address_space(3))) mf_s16x4 mf_lds_s16x4;
constexpr int MF_NBUF = 4, MF_NBUF_STEP = 3, MF_NODES = 32;
constexpr int mfma_set_words(int emax) { return 4 * ctx_epad(emax); }                // raw: g_out, g_in, subject id, object id per edge
constexpr int mfma_fixed_bytes(int emax) { return mfma_set_words(emax) * 4 + 6 * ctx_epad(emax); }   // + packed: 2 x bf16 gates, 2 x byte ids
// whole-step form: raw = 4 dots + 2 ids per edge; packed + g_sub, g_obj (f32); vertex dots (one 1 KiB DMA); two copies of 32 vertex pieces
constexpr int mfma_fixed_bytes_step(int emax) { return 6 * ctx_epad(emax) * 4 + 6 * ctx_epad(emax) + 8 * ctx_epad(emax) + 1024 + 2 * MF_NODES * 128; }
constexpr int mfma_chunk_edges_step(int emax) {
    const int room = (DM_LDS_MAX - mfma_fixed_bytes_step(emax)) / (MF_NBUF_STEP * 128);
    const int need = (emax + 31) / 32 * 32;
    const int eb = (room < need ? room : need) / 32 * 32;
    return eb > 256 ? 256 : eb;                               // 8 K steps per chunk: two for each of the four K-quarter waves
}
static_assert(mfma_chunk_edges_step(992) == 256, "imp_ctx_mfma_kernel<STEP>: a 992-edge graph goes through in four 256-row chunks");
constexpr int mfma_chunk_edges(int emax) {
    const int room = (DM_LDS_MAX - mfma_fixed_bytes(emax)) / (MF_NBUF * 128);
    const int need = (emax + 31) / 32 * 32;
    return (room < need ? room : need) / 32 * 32;
}
static_assert(mfma_chunk_edges(992) == 256, "imp_ctx_mfma_kernel: a 992-edge graph goes through in four 256-row chunks");

// swizzle key of a chunk-local row: the 16 rows one transposing read touches (4 K blocks x 4 rows) get 8 distinct keys per row parity
__device__ __forceinline__ int mf_key(int r) { return ((r >> 1) ^ ((r >> 3) & 3)) & 7; }

__device__ __forceinline__ void dma16_to_lds_s(const void* sbase, unsigned voff, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_base), "v"(voff), "s"(sbase) : "memory");
}

template <bool STEP>
__global__ __launch_bounds__(DM_THREADS) void imp_ctx_mfma_kernel(
    const bf16_t* __restrict__ e, const float* __restrict__ gates_oi, const int* __restrict__ so, const int* __restrict__ img_ptr, int B,
    int N, int H, bf16_t* __restrict__ ctx2, int EMAX, int EB, int sum_ctx,
    // STEP (the whole IMP step in this kernel): vertex rows, the gate dot products and bias, the edge-input rows to write
    const bf16_t* __restrict__ v, const float* __restrict__ ndots, const float* __restrict__ edots, const float* __restrict__ gb,
    bf16_t* __restrict__ e_in) {
    constexpr int PIECE = 128, CHAN = 64, NBUF = STEP ? MF_NBUF_STEP : MF_NBUF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int EPAD = ctx_epad(EMAX), SETW = (STEP ? 6 : 4) * EPAD;
    char* const ring = smem;                                                          // [NBUF][EB][128 B], rows swizzled
    // raw per-edge arrays as the DMA leaves them: g_out, g_in (f32), subject, object (global ids) [4][EPAD]; STEP: the edge's four
    // gate dot products [EPAD][4] f32, then subject, object
    int* const sets = reinterpret_cast<int*>(ring + (long)NBUF * EB * PIECE);
    // what the K steps read, made from the raw arrays at the start of a unit: gates as bf16 (what the A fragment holds anyway) and
    // node ids as bytes (graph-local; 0xff = no node): 24 instead of 64 bytes per lane and K step
    unsigned short* const gbf = reinterpret_cast<unsigned short*>(sets + SETW);       // [2][EPAD] bf16 g_out | g_in
    unsigned char* const nid = reinterpret_cast<unsigned char*>(gbf + 2 * EPAD);      // [2][EPAD] subject | object
    // STEP only: g_sub | g_obj (f32: the edge inputs are exact), the graph's vertex dots, two copies of its vertex pieces (this slice)
    float* const gso = reinterpret_cast<float*>(nid + 2 * EPAD);                      // [2][EPAD]
    float* const ndl = gso + 2 * EPAD;                                                // [64][4] (one DMA instruction: 64 lanes x 16 bytes)
    char* const vsb = reinterpret_cast<char*>(ndl + 256);                             // [2][MF_NODES][128 B]
    const unsigned ring_lds = (unsigned)(unsigned long)(lds_char_t*)smem;
    const unsigned sets_lds = ring_lds + (unsigned)(NBUF * EB * PIECE);

    const int S = H * 2 / PIECE, units = B * S;
    const int G = (int)gridDim.x, NX = min(G, 8), x = (int)blockIdx.x % NX, w = (int)blockIdx.x / NX;
    const int wx = G / NX + (x < G % NX ? 1 : 0);
    const int uq = units / NX, ur = units % NX;
    const int cnt = uq + (x < ur ? 1 : 0), base = x * uq + min(x, ur);
    if (w >= cnt) return;
    const int nunits = (cnt - w + wx - 1) / wx;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = wv >> 3, mt = (wv >> 2) & 1, kq = wv & 3;                          // this wave's role
    const int nks = EB / 32;                                                           // 32-edge K steps per chunk

    struct Hdr { int g, slice, n0, Nn, e0, Ee, nch; };         // Ee < 0: the host's promise about this graph is broken (outputs poisoned); nch chunks
    int lg = base / S, lslice = base - lg * S, lidx = 0;
    auto load_hdr = [&](int k) __attribute__((always_inline)) {
        Hdr h;
        const int idx = w + k * wx;
        lslice += idx - lidx;
        lidx = idx;
        while (lslice >= S) {
            lslice -= S;
            ++lg;
        }
        h.g = lg;
        h.slice = lslice;
        u32x2 nn, ee;
        asm volatile("s_load_dwordx2 %0, %2, 0x0\n\ts_load_dwordx2 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(nn), "=&s"(ee) : "s"(img_ptr + h.g), "s"(img_ptr + (B + 1 + h.g)) : "memory");
        h.n0 = (int)nn.x;
        h.Nn = (int)nn.y - h.n0;
        h.e0 = (int)ee.x;
        h.Ee = (int)ee.y - h.e0;
        if (h.Ee > EMAX || h.Nn > MF_NODES) {
            h.Nn = 0;
            h.Ee = -1;
        }
        h.nch = h.Ee > 0 ? (h.Ee + EB - 1) / EB : 1;             // (the only division: once per header)
        return h;
    };
    int ci = 0;                                                  // DMA operations this wave has issued
    // rows [lo, hi) (graph-local) of h's slice -> ring buffer rb: lane = (row of the instruction's 8, physical 16-byte slot)
    const int l_row = lane >> 3, l_slot = lane & 7;
    auto issue_dma = [&](const Hdr& h, int lo, int hi, int rb) __attribute__((always_inline)) {
        // whole 32-row K steps: the rows past the last edge repeat it (finite data under the gate matrix's zeros -- the slot may
        // hold the f32 partial sums of the last unit's reduction, which read as bf16 can be NaN)
        const int nch = (hi - lo + 31) / 32 * 4;
        const char* sb = reinterpret_cast<const char*>(e) + ((long)h.e0 * H + (long)h.slice * CHAN) * 2;      // uniform
        for (int c = wv; c < nch; c += DM_THREADS / 64) {
            const int r = c * 8 + l_row;                                       // chunk-local row
            const int el = min(lo + r, hi - 1);
            const unsigned voff = (unsigned)el * (unsigned)(H * 2) + (unsigned)((l_slot ^ mf_key(r)) << 4);
            dma16_to_lds_s(sb, voff, __builtin_amdgcn_readfirstlane(ring_lds + (unsigned)(rb * EB * PIECE + c * 1024)));
            ++ci;
        }
    };
    auto wait_mark = [&](int mark) __attribute__((always_inline)) {
        switch (min(ci - mark, 15)) {
#define SGG_W(K) case K: asm volatile("s_waitcnt vmcnt(" #K ")" ::: "memory"); break;
            SGG_W(0) SGG_W(1) SGG_W(2) SGG_W(3) SGG_W(4) SGG_W(5) SGG_W(6) SGG_W(7) SGG_W(8) SGG_W(9) SGG_W(10) SGG_W(11) SGG_W(12)
            SGG_W(13) SGG_W(14)
#undef SGG_W
            default: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        }
    };
    // the four per-edge arrays of h's graph -> set q: lane t of the workgroup <-> edge t
    // the per-edge arrays of h's graph -> raw: lane t of the workgroup <-> edge t (STEP: also the vertex dots, and the vertex pieces
    // of h's slice into copy `vq`)
    auto issue_small = [&](const Hdr& h, int vq) __attribute__((always_inline)) {
        if (wv * 64 < h.Ee) {
            const long et = h.e0 + min(tid, h.Ee - 1);
            if constexpr (STEP) {
                dma16_to_lds(edots + 4 * et, __builtin_amdgcn_readfirstlane(sets_lds + (unsigned)(wv * 1024)));
                dma4_to_lds(so + 2 * et, __builtin_amdgcn_readfirstlane(sets_lds + (unsigned)((4 * EPAD + wv * 64) * 4)));
                dma4_to_lds(so + 2 * et + 1, __builtin_amdgcn_readfirstlane(sets_lds + (unsigned)((5 * EPAD + wv * 64) * 4)));
                ci += 3;
            } else {
                dma4_to_lds(gates_oi + 2 * et, __builtin_amdgcn_readfirstlane(sets_lds + (unsigned)(wv * 256)));
                dma4_to_lds(gates_oi + 2 * et + 1, __builtin_amdgcn_readfirstlane(sets_lds + (unsigned)((EPAD + wv * 64) * 4)));
                dma4_to_lds(so + 2 * et, __builtin_amdgcn_readfirstlane(sets_lds + (unsigned)((2 * EPAD + wv * 64) * 4)));
                dma4_to_lds(so + 2 * et + 1, __builtin_amdgcn_readfirstlane(sets_lds + (unsigned)((3 * EPAD + wv * 64) * 4)));
                ci += 4;
            }
        }
        if constexpr (STEP) {
            if (h.Ee >= 0 && h.Nn > 0) {
                const unsigned ndl_lds = (unsigned)(unsigned long)(lds_char_t*)ndl;
                if (wv == 0) {                                   // vertex dots: lane n <-> node n
                    dma16_to_lds(ndots + 4L * (h.n0 + min(lane, h.Nn - 1)), __builtin_amdgcn_readfirstlane(ndl_lds));
                    ++ci;
                } else if (wv <= MF_NODES / 8) {                 // vertex pieces: 8 nodes x 128 bytes per instruction
                    const int n = min((wv - 1) * 8 + (lane >> 3), h.Nn - 1);
                    dma16_to_lds(reinterpret_cast<const char*>(v) + ((long)(h.n0 + n) * H + (long)h.slice * CHAN) * 2 + (lane & 7) * 16,
                                 __builtin_amdgcn_readfirstlane(ndl_lds + 1024u + (unsigned)(vq * MF_NODES * PIECE + (wv - 1) * 1024)));
                    ++ci;
                }
            }
        }
    };

    Hdr h0 = load_hdr(0), h1 = h0, h2 = h0;
    if (nunits > 1) h1 = load_hdr(1);
    if (nunits > 2) h2 = load_hdr(2);
    int loaded = min(nunits, 3);
    int pu = 0, pb = 0, pr = 0, issued_chunks = 0;
    int mk0 = 0, mk1 = 0, mk2 = 0, mk3 = 0;
    auto mark_of = [&](int r) __attribute__((always_inline)) { return r == 0 ? mk0 : r == 1 ? mk1 : r == 2 ? mk2 : mk3; };
    int cu = 0;
    auto hdr_rel = [&](int rel) __attribute__((always_inline)) {
        Hdr h;
        h.g = rel == 0 ? h0.g : rel == 1 ? h1.g : h2.g;
        h.slice = rel == 0 ? h0.slice : rel == 1 ? h1.slice : h2.slice;
        h.n0 = rel == 0 ? h0.n0 : rel == 1 ? h1.n0 : h2.n0;
        h.Nn = rel == 0 ? h0.Nn : rel == 1 ? h1.Nn : h2.Nn;
        h.e0 = rel == 0 ? h0.e0 : rel == 1 ? h1.e0 : h2.e0;
        h.Ee = rel == 0 ? h0.Ee : rel == 1 ? h1.Ee : h2.Ee;
        h.nch = rel == 0 ? h0.nch : rel == 1 ? h1.nch : h2.nch;
        return h;
    };
    auto produce = [&]() __attribute__((always_inline)) {
        if (pu >= nunits || pu - cu > 2) return false;
        const Hdr hp = hdr_rel(pu - cu);
        const int lo = pb * EB, hi = min(lo + EB, max(hp.Ee, 0));
        if (SGG_DMA_ABL != 3 || issued_chunks == 0) issue_dma(hp, lo, hi, pr);
        mk0 = pr == 0 ? ci : mk0;
        mk1 = pr == 1 ? ci : mk1;
        mk2 = pr == 2 ? ci : mk2;
        mk3 = pr == 3 ? ci : mk3;
        pr = pr + 1 == NBUF ? 0 : pr + 1;
        ++issued_chunks;
        if (++pb >= hp.nch) {
            pb = 0;
            ++pu;
        }
        return true;
    };
    // (stale LDS never meets a zero of the gate matrix: every 32-row K step that is read was filled whole by issue_dma)
    int mark_small;
    issue_small(h0, 0);
    mark_small = ci;
    for (int d = 0; d < NBUF - 1; ++d) produce();
    int cr = 0, consumed = 0;

    // lane constants of the fragments
    const int m16 = lane & 15, kb = lane >> 4;
    // transposing read: lane (kb, q16): row j = q16 >> 2 of its K block, column group cg = q16 & 3 (4 channels)
    const int tj = m16 >> 2, tcg = m16 & 3;
    unsigned boff[4][2];                                         // [16-channel tile][rows 0-3 / 4-7 of the K block], ks = 0
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
            const int r = kb * 8 + hlf * 4 + tj;                 // (ks * 32 does not enter the swizzle key)
            const int slot = (t * 2 + (tcg >> 1)) ^ mf_key(r);
            boff[t][hlf] = (unsigned)(r * PIECE + slot * 16 + (tcg & 1) * 8);
        }

    // thread t: edge t of unit hx -> what the K steps (and, STEP, the edge inputs) read.  Reads the raw arrays this thread's own wave
    // DMA'd (and the vertex dots wave 0 DMA'd): the caller has waited for them and, for the dots, passed a barrier.
    auto convert = [&](const Hdr& hx) __attribute__((always_inline)) {
        if (tid >= EPAD) {
        // (the per-edge arrays hold EPAD entries: small graphs leave most threads without an edge)
    } else if constexpr (STEP) {   // thread t: edge t -> its four gates (rel_model_stanford.py:78-89), byte node ids
        const bool live = tid < hx.Ee;
        const int sl = live ? sets[4 * EPAD + tid] - hx.n0 : 0, ol = live ? sets[5 * EPAD + tid] - hx.n0 : 0;
        const f32x4 de = reinterpret_cast<const f32x4*>(sets)[tid];
        const f32x4 ns = reinterpret_cast<const f32x4*>(ndl)[sl], no = reinterpret_cast<const f32x4*>(ndl)[ol];
        gso[tid] = sigmoidf_(ns.x + de.x + gb[0]);                               // sub_vert (v[s])
        gso[EPAD + tid] = sigmoidf_(no.y + de.y + gb[1]);                        // obj_vert (v[o])
        gbf[tid] = live ? (unsigned short)(pack_bf16x2(sigmoidf_(ns.z + de.z + gb[2]), 0.f) & 0xffffu) : (unsigned short)0;          // out_edge
        gbf[EPAD + tid] = live ? (unsigned short)(pack_bf16x2(sigmoidf_(no.w + de.w + gb[3]), 0.f) & 0xffffu) : (unsigned short)0;   // in_edge
        nid[tid] = live ? (unsigned char)sl : (unsigned char)0xff;
        nid[EPAD + tid] = live ? (unsigned char)ol : (unsigned char)0xff;
    } else {   // thread t: edge t -> bf16 gates, byte node ids (every wave has left the last unit: the packed arrays are free)
        const float* graw = reinterpret_cast<const float*>(sets);
        const bool live = tid < hx.Ee;
        gbf[tid] = live ? (unsigned short)(pack_bf16x2(graw[tid], 0.f) & 0xffffu) : (unsigned short)0;
        gbf[EPAD + tid] = live ? (unsigned short)(pack_bf16x2(graw[EPAD + tid], 0.f) & 0xffffu) : (unsigned short)0;
        nid[tid] = live ? (unsigned char)(sets[2 * EPAD + tid] - hx.n0) : (unsigned char)0xff;
        nid[EPAD + tid] = live ? (unsigned char)(sets[3 * EPAD + tid] - hx.n0) : (unsigned char)0xff;
    }
    };
    // ---- first unit: its small arrays -> packed; then the second unit's small arrays are requested
    wait_mark(mark_small);
    __syncthreads();
    convert(h0);
    __syncthreads();
    if (nunits > 1) {
        issue_small(h1, 1);
        mark_small = ci;
    }

    for (cu = 0; cu < nunits; ++cu) {
        const Hdr hc = h0;
        const unsigned short* const gate = gbf + dir * EPAD;     // g_out | g_in
        const unsigned char* const node = nid + dir * EPAD;      // subject | object
        const unsigned target = (unsigned)(mt * 16 + m16);       // this lane's node (graph-local; ids of real edges are < Nn, padding is 0xff)
        f32x4 acc[2][4];                                         // [step parity][channel tile]
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int nchunks = hc.nch;
        int last_slot = cr;
        for (int cb = 0; cb < nchunks; ++cb) {
            const int blo = cb * EB, bhi = min(blo + EB, max(hc.Ee, 0));
            [[maybe_unused]] const int unit_no = consumed;
            if constexpr (STEP && SGG_DMA_ABL != 5) {
                // the edge inputs of this chunk's edges, e_in = g_sub v[s] + g_obj v[o] (rel_model_stanford.py:78-81): they need no edge
                // row, so they are made while the chunk's DMA is still landing.  Lane group (8 lanes x 16 bytes) <-> row piece.
                const char* vq = vsb + (cu & 1) * MF_NODES * PIECE + (tid & 7) * 16;
                for (int r = blo + (tid >> 3); r < bhi; r += DM_THREADS / 8) {
                    const float gs = gso[r], go = gso[EPAD + r];
                    Piece16<bf16_t> ps, po;
                    ps.r = *reinterpret_cast<const u32x4*>(vq + nid[r] * PIECE);
                    po.r = *reinterpret_cast<const u32x4*>(vq + nid[EPAD + r] * PIECE);
                    float vn[8], yy[8], rr[8];
                    ps.get(vn);
                    po.get(yy);
#pragma unroll
                    for (int j = 0; j < 8; ++j) rr[j] = gs * vn[j] + go * yy[j];
                    store8(e_in + ((long)(hc.e0 + r) * H + hc.slice * CHAN + (tid & 7) * 8), rr);
                }
                for (int r0 = blo + wv * 8; r0 < bhi; r0 += DM_THREADS / 8) ++ci;      // this wave's store instructions (vmcnt counts them)
            }
            SGG_DTICK(0)
            wait_mark(mark_of(cr));
            __syncthreads();                                     // (D) chunk visible; every wave has left the last chunk (and unit)
            SGG_DTICK(1)
            if (issued_chunks - consumed < NBUF) produce();
            SGG_DTICK(2)
            const lds_char_t* const slot_lds = (const lds_char_t*)(ring + cr * EB * PIECE);
            // one 32-edge K step: A fragment = 16 nodes x 32 edges of the gate matrix (lane: node m16, K block kb: 8 edges), B
            // fragments = 32 edges x 16 channels for this wave's two channel tiles through the transposing read (compiler-visible
            // builtins: it hoists the reads of the next step above this step's arithmetic and places the waits itself)
            auto kstep = [&](int ks, f32x4 (&c)[4]) __attribute__((always_inline)) {
                const int el0 = blo + ks * 32 + kb * 8;          // this lane's 8 edges (16-byte / 8-byte aligned in the packed arrays)
                const u32x4 g8 = *reinterpret_cast<const u32x4*>(gate + el0);
                const u32x2 n8 = *reinterpret_cast<const u32x2*>(node + el0);
                const lds_char_t* sl = slot_lds + ks * 32 * PIECE;
                mf_s16x4 b[4][2];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    b[t][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((mf_lds_s16x4*)(sl + boff[t][0]));
                    b[t][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((mf_lds_s16x4*)(sl + boff[t][1]));
                }
                // A fragment element j = gate j where byte j of the ids is this lane's node, else 0: a byte compare and a half-word
                // select per element, straight into the packed bf16 registers (SDWA operand selects: no unpacking, no conversion)
                u32x4 ap = {0u, 0u, 0u, 0u};
                const unsigned zero = 0u;
#define SGG_AEL(IDS, J, AP, GP, HW)                                                                                   \
    asm("v_cmp_eq_u32_sdwa vcc, %2, %3 src0_sel:BYTE_" #J " src1_sel:DWORD\n\t"                                         \
        "v_cndmask_b32_sdwa %0, %4, %1, vcc dst_sel:WORD_" #HW " dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:WORD_" #HW \
        : "+v"(AP) : "v"(GP), "v"(IDS), "v"(target), "v"(zero) : "vcc")
                SGG_AEL(n8.x, 0, ap.x, g8.x, 0); SGG_AEL(n8.x, 1, ap.x, g8.x, 1);
                SGG_AEL(n8.x, 2, ap.y, g8.y, 0); SGG_AEL(n8.x, 3, ap.y, g8.y, 1);
                SGG_AEL(n8.y, 0, ap.z, g8.z, 0); SGG_AEL(n8.y, 1, ap.z, g8.z, 1);
                SGG_AEL(n8.y, 2, ap.w, g8.w, 0); SGG_AEL(n8.y, 3, ap.w, g8.w, 1);
#undef SGG_AEL
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const mf_s16x8 bp = __builtin_shufflevector(b[t][0], b[t][1], 0, 1, 2, 3, 4, 5, 6, 7);
                    c[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mf_bf16x8, ap), __builtin_bit_cast(mf_bf16x8, bp), c[t], 0, 0, 0);
                }
            };
            // this wave's K steps of the chunk: every fourth one; two per turn on separate accumulators
            for (int ks = kq; SGG_DMA_ABL != 4 && ks < nks && blo + ks * 32 < bhi; ks += 8) {
                kstep(ks, acc[0]);
                if (ks + 4 < nks && blo + (ks + 4) * 32 < bhi) kstep(ks + 4, acc[1]);
            }
            SGG_DTICK(3)
            SGG_DTICK(4)
            last_slot = cr;
            cr = cr + 1 == NBUF ? 0 : cr + 1;
            ++consumed;
        }
        // ---- unit done: the four K quarters (and, for the summed ctx, the two sums) meet through LDS
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[0][t] += acc[1][t];
        { [[maybe_unused]] const int unit_no = consumed - 1; SGG_DTICK(5) }
        lds_reads_done_barrier();                                // (E1) every wave has left the unit's last chunk
        char* const fixed_end = STEP ? vsb + 2 * MF_NODES * PIECE : reinterpret_cast<char*>(nid + 2 * EPAD);
        auto put_to = [&](f32x4* d) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < 4; ++t) d[t * 64] = acc[0][t];
        };
        auto add_from = [&](const f32x4* d) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[0][t] += d[t * 64];
        };
        {
            // [8 regions][16 registers][64 lanes] f32 = 32 KiB: the ring slot consumed last when it is that large (256-row chunks),
            // else a region of its own behind the sets.  Quarters 2 and 3 write, 0 and 1 add; 1 writes its sum back over what it
            // read, 0 adds.  The next unit's conversion pass rides in the shadow of the rounds: its raw arrays are waited for before
            // (R1) -- which also publishes the vertex dots wave 0 fetched --, the pass runs between (R1) and (R2), and (R2) publishes
            // its results (the packed arrays have been free since (E1)); then the raw arrays go to the DMA of the unit after next.
            // (Tried: a 16 KiB region of its own + a chain 3 -> 2 -> 1 -> 0, so that the slot is refilled at (E1) already: same time.)
            float* const scratch = EB * PIECE >= 32768 ? reinterpret_cast<float*>(ring + last_slot * EB * PIECE)
                                                       : reinterpret_cast<float*>(smem + (((fixed_end - smem) + 15) & ~15L));
            auto region = [&](int d, int m, int h) __attribute__((always_inline)) {
                return reinterpret_cast<f32x4*>(scratch + ((d * 2 + m) * 2 + h) * 1024) + lane;    // [4 tiles][64 lanes] x 16 bytes
            };
            if (kq >= 2) put_to(region(dir, mt, kq - 2));
            if (cu + 1 < nunits) wait_mark(mark_small);
            __syncthreads();                                     // (R1)
            if (kq < 2) add_from(region(dir, mt, kq));
            if (kq == 1) put_to(region(dir, mt, 1));             // (only this wave read that region)
            if (cu + 1 < nunits) convert(h1);
            __syncthreads();                                     // (R2)
            if (cu + 2 < nunits) {
                issue_small(h2, cu & 1);                         // (the vertex-piece copy this unit used)
                mark_small = ci;
            }
            if (kq == 0) add_from(region(dir, mt, 1));
            if (sum_ctx) {
                if (kq == 0 && dir == 1) put_to(region(1, mt, 0));   // (read by this wave alone in round 1)
                __syncthreads();                                 // (R3)
                if (kq == 0 && dir == 0) add_from(region(1, mt, 0));
            }
        }
        if (kq == 0 && (dir == 0 || !sum_ctx) && hc.Ee >= 0) {
            // lane: channel m16 of each 16-channel tile, nodes kb*4 + i of this wave's node tile
            bf16_t* out = ctx2 + ((long)(dir && !sum_ctx ? N : 0) + hc.n0 + mt * 16 + kb * 4) * H + hc.slice * CHAN + m16;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float v[4] = {acc[0][t].x, acc[0][t].y, acc[0][t].z, acc[0][t].w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (mt * 16 + kb * 4 + i < hc.Nn) out[(long)i * H + t * 16] = f32_to_bf16(v[i]);
            }
        }
        if (hc.Ee < 0) {                                         // broken promise: this graph's slice of the outputs is NaN
            u32x2 nn;
            asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(nn) : "s"(img_ptr + __builtin_amdgcn_readfirstlane(hc.g)) : "memory");
            const int n0 = (int)nn.x, Nn = (int)nn.y - n0;
            for (int k = tid; k < Nn * CHAN; k += DM_THREADS) {
                const long o = (long)(n0 + k / CHAN) * H + hc.slice * CHAN + k % CHAN;
                ctx2[o] = f32_to_bf16(__builtin_nanf(""));
                if (!sum_ctx) ctx2[(long)N * H + o] = f32_to_bf16(__builtin_nanf(""));
            }
            if constexpr (STEP) {
                u32x2 ee;
                asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(ee) : "s"(img_ptr + (B + 1 + __builtin_amdgcn_readfirstlane(hc.g))) : "memory");
                const int e0 = (int)ee.x, Ee = (int)ee.y - e0;
                for (long k = tid; k < (long)Ee * CHAN; k += DM_THREADS)
                    e_in[(e0 + k / CHAN) * H + hc.slice * CHAN + k % CHAN] = f32_to_bf16(__builtin_nanf(""));
            }
        }
        { [[maybe_unused]] const int unit_no = consumed - 1; SGG_DTICK(6) }
        // (the scratch slot is refilled by a DMA only after the next unit's first barrier (D): every reader has consumed its reads by then)
        h0 = h1;
        h1 = h2;
        if (loaded < nunits) {
            h2 = load_hdr(loaded);
            ++loaded;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The edge half of the step on its own:  e_in[e] = g_sub(e) v[s(e)] + g_obj(e) v[o(e)]   (rel_model_stanford.py:78-81).
// The edge ROW is not an input: it enters only through its two gate dot products, which the GRU gate kernel already left in
// edots.  So the step splits into a WRITE stream (this kernel: E rows out, the L2-resident vertex rows in) and a READ stream
// (imp_ctx_kernel: E rows in, two weighted sums per node out) with nothing in common but the gates -- each
// streams whole rows / full cache lines in one direction, and the two can run side by side (the forward puts the read stream on
// the node lane's stream, in front of the node GRU that consumes it).  Any edge list: no graph structure is used.
// A wave owns `rpw` consecutive rows: lane l makes the two gates of row l (coalesced 8-byte / 16-byte fetches), then the rows are
// written one per step, (s, o, g_sub, g_obj) broadcast with v_readlane; the subject's piece stays in registers while s repeats.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void imp_edge_in_kernel(const T* __restrict__ v, const int* __restrict__ so,
                                                          const float* __restrict__ ndots, const float* __restrict__ edots,
                                                          const float* __restrict__ gb, T* __restrict__ e_in,
                                                          float* __restrict__ gates_oi, int E, int H, int rpw) {
    constexpr int CHL = 16 / (int)sizeof(T);
    const int lane = threadIdx.x & 63;
    const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long base = wid * rpw;
    if (base >= E) return;
    const int nrows = (int)min((long)rpw, (long)E - base);
    int s_l = 0, o_l = 0;
    float gs_l = 0.f, go_l = 0.f;
    if (lane < nrows) {
        const long e = base + lane;
        const u32x2 p = *reinterpret_cast<const u32x2*>(so + 2 * e);
        s_l = (int)p.x;
        o_l = (int)p.y;
        const f32x4 d = *reinterpret_cast<const f32x4*>(edots + 4 * e);
        const f32x4 ns = *reinterpret_cast<const f32x4*>(ndots + 4 * (long)s_l), no = *reinterpret_cast<const f32x4*>(ndots + 4 * (long)o_l);
        gs_l = sigmoidf_(ns.x + d.x + gb[0]);
        go_l = sigmoidf_(no.y + d.y + gb[1]);
        if (gates_oi) {                                       // the read stream's two gates of this row (rel_model_stanford.py:86-89)
            f32x2_t gt;
            gt.x = sigmoidf_(ns.z + d.z + gb[2]);             // out_edge (v[s])
            gt.y = sigmoidf_(no.w + d.w + gb[3]);             // in_edge  (v[o])
            *reinterpret_cast<f32x2_t*>(gates_oi + 2 * e) = gt;
        }
    }
    const int CPR = H / CHL;                                  // 16-byte pieces per row
    for (int c = lane; c < CPR; c += 64) {
        const long col = (long)c * CHL;
        int prev_s = -1;
        float vs_[CHL];
#pragma unroll
        for (int j = 0; j < CHL; ++j) vs_[j] = 0.f;
        constexpr int UR = 4;                                 // object pieces in flight
        for (int j0 = 0; j0 < nrows; j0 += UR) {
            Piece16<T> po[UR];
            int sj[UR];
            float gsj[UR], goj[UR];
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const int j = min(j0 + u, nrows - 1);
                sj[u] = __builtin_amdgcn_readlane(s_l, j);
                const int oj = __builtin_amdgcn_readlane(o_l, j);
                gsj[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gs_l), j));
                goj[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(go_l), j));
                po[u].r = *reinterpret_cast<const decltype(po[u].r)*>(v + (long)oj * H + col);
            }
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                if (j0 + u >= nrows) break;                  // wave-uniform
                if (sj[u] != prev_s) {                       // wave-uniform: edges come sorted by subject
                    Piece16<T> ps;
                    ps.r = *reinterpret_cast<const decltype(ps.r)*>(v + (long)sj[u] * H + col);
                    ps.get(vs_);
                    prev_s = sj[u];
                }
                float yy[CHL], rr[CHL];
                po[u].get(yy);
#pragma unroll
                for (int q = 0; q < CHL; ++q) rr[q] = gsj[u] * vs_[q] + goj[u] * yy[q];
                Piece16<T>::store(e_in + (base + j0 + u) * H + col, rr);
            }
        }
    }
}

// persistent form: <= 2 resident workgroups per CU walk the units of their XCD (imp_stream_kernel)
template <typename T, int LP>
int launch_stream(const void* v, const void* e, const int* so, const int* in_ptr, const int* in_ids, const int* img_ptr, int B, int N,
                  int H, const float* ndots, const float* edots, const float* gb, void* e_in, void* ctx2, int max_edges, int max_nodes,
                  int sum_ctx, int max_wgs, hipStream_t s) {
    auto k = imp_stream_kernel<T, LP>;
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                stream_lds_bytes<LP>(SliceCfg<LP>::EMAX, SL_NMAX)) != hipSuccess)
            return SGG_ERR_LAUNCH;
        configured = true;
    }
    const int emax = (max(max_edges, 8) + 7) & ~7, nmax = (max(max_nodes, 8) + 7) & ~7;
    const int lds = stream_lds_bytes<LP>(emax, nmax);
    const int per_cu = lds * 2 <= 160 * 1024 ? 2 : 1;       // register budget (<= 128 VGPRs) allows two 8-wave workgroups
    const int S = H * (int)sizeof(T) / (LP * 16), units = B * S;
    int grid = min(units, 256 * per_cu);
    if (max_wgs > 0) grid = min(grid, max_wgs);
    hipLaunchKernelGGL(k, dim3(grid), dim3(SL_THREADS), lds, s, (const T*)v, (const T*)e, so, in_ptr, in_ids, img_ptr, B, N, H, ndots,
                       edots, gb, (T*)e_in, (T*)ctx2, emax, nmax, sum_ctx);
    return hipGetLastError() == hipSuccess ? SGG_OK : SGG_ERR_LAUNCH;
}

}  // namespace

#ifdef SGG_DMA_TICKS
extern "C" int sgg_dbg_dma_ticks(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dma_ticks), sizeof(long long) * 8 * 16 * 8) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int sgg_graph_ptr(const int64_t* im_inds, int N, int B, const int* out_ptr, int* img_ptr, void* stream) {
    if (!im_inds || !img_ptr || !out_ptr || N < 0 || B < 0) return SGG_ERR_ARG;
    hipLaunchKernelGGL(graph_ptr_kernel, dim3((B + 256) / 256), dim3(256), 0, (hipStream_t)stream, im_inds, N, B, out_ptr, img_ptr);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

namespace {
template <typename T, int LP>
int launch_ctx(const void* e, const float* gates_oi, const int* in_ptr, const int* in_ids, const int* img_ptr, int B, int N, int H,
               void* ctx2, int emax, int nmax, int eb, int sum_ctx, int max_wgs, hipStream_t s) {
    auto k = imp_ctx_kernel<T, LP>;
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, DM_LDS_MAX) != hipSuccess)
            return SGG_ERR_LAUNCH;
        configured = true;
    }
    const int piece = LP * 16;
    const int units = B * (H * (int)sizeof(T) / piece);
    int grid = min(units, 256);
    if (max_wgs > 0) grid = min(grid, max_wgs);
    hipLaunchKernelGGL(k, dim3(grid), dim3(DM_THREADS), CX_NBUF * eb * piece + ctx_fixed_bytes(emax, nmax, 4 * piece / (int)sizeof(T)), s,
                       (const T*)e, gates_oi, in_ptr, in_ids, img_ptr, B, N, H, (T*)ctx2, emax, nmax, eb, sum_ctx);
    return hipGetLastError() == hipSuccess ? SGG_OK : SGG_ERR_LAUNCH;
}

// the matrix-core read stream / whole step (bf16, <= 32 nodes and <= 1024 edges per graph, rows a multiple of 128 bytes); 1 when it
// does not apply.  step: v, node_dots, edge_dots, gate_b, e_in given -> the kernel also makes the gates and the edge inputs.
int ctx_mfma_dispatch(bool step, const void* e, const float* gates_oi, const int* so, const int* img_ptr, int B, int N, int H, void* ctx2,
                      int max_edges, int max_nodes, int sum_ctx, int dtype, int max_wgs, hipStream_t s, const void* v = nullptr,
                      const float* node_dots = nullptr, const float* edge_dots = nullptr, const float* gate_b = nullptr, void* e_in = nullptr) {
    if (dtype != SGG_BF16 || !so || max_nodes > MF_NODES || max_edges > DM_EMAX || (H * 2) % 128) return 1;
    const int emax = (max(max_edges, 32) + 31) & ~31;
    int eb = step ? mfma_chunk_edges_step(emax) : mfma_chunk_edges(emax);
    const char* ebs = getenv("SGG_IMP_EB");     // tests: small chunks (many per unit on small graphs)
    if (ebs && atoi(ebs) >= 32) eb = min(eb, atoi(ebs) & ~31);
    if (eb < 32) return 1;
    auto k = step ? imp_ctx_mfma_kernel<true> : imp_ctx_mfma_kernel<false>;
    static bool configured[2] = {false, false};
    if (!configured[step]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, DM_LDS_MAX) != hipSuccess)
            return SGG_ERR_LAUNCH;
        configured[step] = true;
    }
    const int units = B * (H * 2 / 128);
    int grid = min(units, 256);
    if (max_wgs > 0) grid = min(grid, max_wgs);
    const int smem = (step ? MF_NBUF_STEP : MF_NBUF) * eb * 128 + (step ? mfma_fixed_bytes_step(emax) : mfma_fixed_bytes(emax)) +
                     (eb * 128 >= 32768 ? 0 : 32768 + 16);
    if (smem > DM_LDS_MAX) return 1;
    hipLaunchKernelGGL(k, dim3(grid), dim3(DM_THREADS), smem, s, (const bf16_t*)e, gates_oi, so, img_ptr, B, N, H, (bf16_t*)ctx2, emax, eb,
                       sum_ctx, (const bf16_t*)v, node_dots, edge_dots, gate_b, (bf16_t*)e_in);
    return hipGetLastError() == hipSuccess ? SGG_OK : SGG_ERR_LAUNCH;
}

// the ring-buffered read stream; 1 when the graphs do not fit it (above one edge per thread, rows not a multiple of 64 bytes)
int ctx_dispatch(const void* e, const float* gates_oi, const int* in_ptr, const int* in_ids, const int* img_ptr, int B, int N, int E, int H,
                 void* ctx2, int max_edges, int max_nodes, int sum_ctx, int dtype, int max_wgs, hipStream_t s) {
    const int tsz = dtype == SGG_BF16 ? 2 : 4;
    const int row = H * tsz;
    if (row % 64 || max_nodes > DM_NMAX || max_edges > DM_EMAX) return 1;
    const int emax = (max(max_edges, 16) + 15) & ~15, nmax = (max(max_nodes, 8) + 7) & ~7;
    const char* pc = getenv("SGG_IMP_PIECE");
    int piece = (row % 128 == 0 && (long)B * (row / 128) >= 256) ? 128 : 64;
    if (pc && atoi(pc) == 64) piece = 64;
    if (pc && atoi(pc) == 128 && row % 128 == 0) piece = 128;
    int eb = ctx_chunk_edges(emax, nmax, 4 * piece / tsz, piece);
    if (eb < 16) return 1;
    const char* ebs = getenv("SGG_IMP_EB");     // tests: small chunks (many per unit on small graphs)
    if (ebs && atoi(ebs) >= 16) eb = min(eb, atoi(ebs) & ~15);
#define SGG_CTX(T, LPV) \
    return launch_ctx<T, LPV>(e, gates_oi, in_ptr, in_ids, img_ptr, B, N, H, ctx2, emax, nmax, eb, sum_ctx, max_wgs, s)
    if (dtype == SGG_BF16) {
        if (piece == 128) SGG_CTX(bf16_t, 8);
        SGG_CTX(bf16_t, 4);
    } else {
        if (piece == 128) SGG_CTX(float, 8);
        SGG_CTX(float, 4);
    }
#undef SGG_CTX
}
}  // namespace

// ---- the split step (what the forward runs): a write stream and a read stream that share nothing but the gate dot products
extern "C" int sgg_imp_edge_in_fwd(const void* v, const int* so, const float* node_dots, const float* edge_dots, const float* gate_b,
                                   void* e_in, float* gates_oi, int E, int H, int dtype, void* stream) {
    if (E == 0) return SGG_OK;
    if (!v || !so || !node_dots || !edge_dots || !gate_b || !e_in || E < 0 || H <= 0) return SGG_ERR_ARG;
    if (dtype != SGG_BF16 && dtype != SGG_F32) return SGG_ERR_DTYPE;
    if ((H * (dtype == SGG_BF16 ? 2 : 4)) % 16) return SGG_ERR_ARG;
    // rows per wave: enough waves to fill the chip (256 CUs x 8) on small batches, 64-row batches on large ones
    int rpw = (int)(((long)E + 2047) / 2048);
    rpw = rpw < 4 ? 4 : rpw > 64 ? 64 : rpw;
    const long waves = ((long)E + rpw - 1) / rpw;
    const dim3 grid((unsigned)((waves + 3) / 4)), blk(256);
    hipStream_t s = (hipStream_t)stream;
    SGG_DISPATCH_T(dtype,
        hipLaunchKernelGGL(imp_edge_in_kernel<bf16_t>, grid, blk, 0, s, (const bf16_t*)v, so, node_dots, edge_dots, gate_b, (bf16_t*)e_in, gates_oi, E, H, rpw),
        hipLaunchKernelGGL(imp_edge_in_kernel<float>, grid, blk, 0, s, (const float*)v, so, node_dots, edge_dots, gate_b, (float*)e_in, gates_oi, E, H, rpw));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_imp_ctx_fwd(const void* e, const float* gates_oi, const int* so, const int* in_ptr, const int* in_ids, const int* img_ptr,
                               int B, int N, int E, int H, void* ctx2, int max_edges, int max_nodes, int sum_ctx, int dtype, void* stream) {
    if (N == 0 || B == 0) return SGG_OK;
    if (!e || !gates_oi || !in_ptr || !in_ids || !img_ptr || !ctx2 || N < 0 || E < 0 || B < 0 || H <= 0) return SGG_ERR_ARG;
    if (dtype != SGG_BF16 && dtype != SGG_F32) return SGG_ERR_DTYPE;
    const char* mw = getenv("SGG_IMP_MAX_WGS");
    const char* form = getenv("SGG_IMP_CTX");   // kernel experiments / cross-checks: "valu" = the list-walking kernel for every graph
    int rc = 1;
    if (!(form && form[0] == 'v'))              // bf16 graphs of <= 32 nodes with (s, o) given: the gate-matrix product on the matrix cores
        rc = ctx_mfma_dispatch(false, e, gates_oi, so, img_ptr, B, N, H, ctx2, max_edges, max_nodes, sum_ctx, dtype, mw ? atoi(mw) : 0, (hipStream_t)stream);
    if (rc == 1)
        rc = ctx_dispatch(e, gates_oi, in_ptr, in_ids, img_ptr, B, N, E, H, ctx2, max_edges, max_nodes, sum_ctx, dtype, mw ? atoi(mw) : 0,
                          (hipStream_t)stream);
    return rc == 1 ? SGG_ERR_CAPACITY : rc;
}

// One launch for the whole step on the matrix-core kernel: gates from the dot products, e_in = g_sub v[s] + g_obj v[o], the two context
// sums.  bf16, graphs of <= 32 nodes / <= 1024 edges, rows a multiple of 128 bytes: SGG_ERR_CAPACITY otherwise.
extern "C" int sgg_imp_step_fwd(const void* v, const void* e, const int* so, const int* img_ptr, int B, int N, int E, int H,
                                const float* node_dots, const float* edge_dots, const float* gate_b, void* e_in, void* ctx2, int max_edges,
                                int max_nodes, int sum_ctx, int dtype, void* stream) {
    if (N == 0 || B == 0) return SGG_OK;
    if (!v || !e || !so || !img_ptr || !node_dots || !edge_dots || !gate_b || !e_in || !ctx2 || N < 0 || E < 0 || B < 0 || H <= 0)
        return SGG_ERR_ARG;
    if (dtype != SGG_BF16 && dtype != SGG_F32) return SGG_ERR_DTYPE;
    if ((long)E * H * 2 >= 0xffff0000L) return SGG_ERR_SPAN;
    const char* mw = getenv("SGG_IMP_MAX_WGS");
    const int rc = ctx_mfma_dispatch(true, e, nullptr, so, img_ptr, B, N, H, ctx2, max_edges, max_nodes, sum_ctx, dtype, mw ? atoi(mw) : 0,
                                     (hipStream_t)stream, v, node_dots, edge_dots, gate_b, e_in);
    return rc == 1 ? SGG_ERR_CAPACITY : rc;
}

// (graph, 64-channel slice) units from which sgg_imp_sliced_fwd hands bf16 graphs of <= 32 nodes to the persistent matrix-core step
// (256 workgroups, one per CU: four units each; measured on complete 32-node graphs: 96 images even, 128 images 64 against 72-75 us)
constexpr int IMP_STEP_MIN_UNITS = 1024;
extern "C" int sgg_imp_step_min_units(void) { return IMP_STEP_MIN_UNITS; }

// largest per-graph edge count the sliced kernel takes at this row width (0: rows too narrow for any slicing)
extern "C" int sgg_imp_sliced_capacity(int H, int dtype) {
    const int row = H * (dtype == SGG_BF16 ? 2 : 4);
    if (row % 32 || H <= 0) return 0;
    return SliceCfg<2>::EMAX;   // the narrowest pieces hold the most edges
}

extern "C" int sgg_imp_sliced_fwd(const void* v, const void* e, const int* so, const int* out_ptr, const int* in_ptr,
                                  const int* in_ids, const int* img_ptr, int B, int N, int E, int H, const float* node_dots,
                                  const float* edge_dots, const float* gate_b, void* e_in, void* ctx2, int max_edges, int max_nodes,
                                  int sum_ctx, int dtype, void* stream) {
    if (N == 0 || B == 0) return SGG_OK;
    if (!v || !e || !so || !out_ptr || !in_ptr || !in_ids || !img_ptr || !node_dots || !edge_dots || !gate_b || !e_in || !ctx2 ||
        N < 0 || E < 0 || B < 0 || H <= 0)
        return SGG_ERR_ARG;
    if (dtype != SGG_BF16 && dtype != SGG_F32) return SGG_ERR_DTYPE;
    const int row = H * (dtype == SGG_BF16 ? 2 : 4);
    if (row % 32) return SGG_ERR_ARG;
    if (max_nodes > SL_NMAX) return SGG_ERR_CAPACITY;
    hipStream_t s = (hipStream_t)stream;
    // piece width: 64-byte pieces (LP = 4) first -- a 992-edge graph then parks 62 KB, two workgroups fit a CU and one's loads
    // run under the other's gate / accumulate phases (measured: 9.1 vs 12.2 us at B=8, 70 vs 72 us at B=128 against 128-byte
    // pieces with one workgroup per CU); 32-byte pieces for graphs above LP = 4's capacity
    int lp = 0;
    if (row % 64 == 0 && max_edges <= SliceCfg<4>::EMAX) lp = 4;
    else if (max_edges <= SliceCfg<2>::EMAX) lp = 2;
    else return SGG_ERR_CAPACITY;
    static const char* force = getenv("SGG_IMP_LP");     // kernel experiments only
    if (force) {
        const int f = atoi(force);
        if ((f == 8 || f == 4 || f == 2) && row % (f * 16) == 0 &&
            max_edges <= (f == 8 ? SliceCfg<8>::EMAX : f == 4 ? SliceCfg<4>::EMAX : SliceCfg<2>::EMAX))
            lp = f;
    }
    // Default: the short-lived form (one workgroup per unit, two per CU) -- the fastest measured up to a few hundred units; from
    // IMP_STEP_MIN_UNITS (graph, 64-channel slice) units on, bf16 graphs of <= 32 nodes go to the persistent matrix-core step
    // (imp_ctx_mfma_kernel<STEP>, the kernel behind sgg_imp_step_fwd; DESIGN.md "IMP step").
    // SGG_IMP_STREAM=0: never route; =1: the register-prefetch persistent form, kept as a measured experiment and a cross-check
    // (tests/test_kernels_gpu.py).  SGG_IMP_MAX_WGS=n caps the persistent grids (tests: several units per workgroup on small batches).
    const char* st = getenv("SGG_IMP_STREAM");
    const char* mw = getenv("SGG_IMP_MAX_WGS");
    const int max_wgs = mw ? atoi(mw) : 0;
    if (!st && dtype == SGG_BF16 && row % 128 == 0 && (long)B * (row / 128) >= IMP_STEP_MIN_UNITS && (long)E * row < 0xffff0000L) {
        const int rc = ctx_mfma_dispatch(true, e, nullptr, so, img_ptr, B, N, H, ctx2, max_edges, max_nodes, sum_ctx, dtype, max_wgs, s, v,
                                         node_dots, edge_dots, gate_b, e_in);
        if (rc != 1) return rc;                              // 1: these graphs do not fit it
    }
    if (st && st[0] == '1') {
#define SGG_STREAM(T, LPV) \
    return launch_stream<T, LPV>(v, e, so, in_ptr, in_ids, img_ptr, B, N, H, node_dots, edge_dots, gate_b, e_in, ctx2, max_edges, \
                                 max_nodes, sum_ctx, max_wgs, s)
        if (dtype == SGG_BF16) {
            if (lp == 8) SGG_STREAM(bf16_t, 8);
            if (lp == 4) SGG_STREAM(bf16_t, 4);
            SGG_STREAM(bf16_t, 2);
        } else {
            if (lp == 8) SGG_STREAM(float, 8);
            if (lp == 4) SGG_STREAM(float, 4);
            SGG_STREAM(float, 2);
        }
#undef SGG_STREAM
    }
#define SGG_SLICED(T, LPV) \
    return launch_sliced<T, LPV>(v, e, so, out_ptr, in_ptr, in_ids, img_ptr, B, N, H, node_dots, edge_dots, gate_b, e_in, ctx2, \
                                 max_edges, sum_ctx, s)
    if (dtype == SGG_BF16) {
        if (lp == 8) SGG_SLICED(bf16_t, 8);
        if (lp == 4) SGG_SLICED(bf16_t, 4);
        SGG_SLICED(bf16_t, 2);
    } else {
        if (lp == 8) SGG_SLICED(float, 8);
        if (lp == 4) SGG_SLICED(float, 4);
        SGG_SLICED(float, 2);
    }
#undef SGG_SLICED
}

extern "C" int sgg_imp_fused_fwd(const void* v, const void* e, const int* so, const int* flags, const int* out_ptr,
                                 const int* out_ids, const int* in_ptr, const int* in_ids, int N, int E, int H,
                                 const void* gate_w, const float* gate_b, void* e_in, void* ctx2, int dtype, void* stream) {
    if (N == 0) return SGG_OK;
    if (!v || !e || !so || !flags || !out_ptr || !out_ids || !in_ptr || !in_ids || !gate_w || !gate_b || !e_in || !ctx2 || N < 0 ||
        E < 0 || H <= 0 || (H & 7) || H > MAXH)
        return SGG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int units = 2 * N;
    // enough units to give every CU >= 8 single-wave units: one wave per unit; otherwise one workgroup (4 waves) per unit
    if (units >= 256 * 8) {
        const dim3 grid(min((units + 3) / 4, 256 * 3)), blk(256);
        SGG_DISPATCH_T(dtype,
            hipLaunchKernelGGL((imp_fused_kernel<bf16_t, 1>), grid, blk, 0, s, (const bf16_t*)v, (const bf16_t*)e, so, flags, out_ptr, out_ids, in_ptr, in_ids, N, H, (const bf16_t*)gate_w, gate_b, (bf16_t*)e_in, (bf16_t*)ctx2),
            hipLaunchKernelGGL((imp_fused_kernel<float, 1>), grid, blk, 0, s, (const float*)v, (const float*)e, so, flags, out_ptr, out_ids, in_ptr, in_ids, N, H, (const float*)gate_w, gate_b, (float*)e_in, (float*)ctx2));
    } else {
        const dim3 grid(units), blk(256);
        SGG_DISPATCH_T(dtype,
            hipLaunchKernelGGL((imp_fused_kernel<bf16_t, 4>), grid, blk, 0, s, (const bf16_t*)v, (const bf16_t*)e, so, flags, out_ptr, out_ids, in_ptr, in_ids, N, H, (const bf16_t*)gate_w, gate_b, (bf16_t*)e_in, (bf16_t*)ctx2),
            hipLaunchKernelGGL((imp_fused_kernel<float, 4>), grid, blk, 0, s, (const float*)v, (const float*)e, so, flags, out_ptr, out_ids, in_ptr, in_ids, N, H, (const float*)gate_w, gate_b, (float*)e_in, (float*)ctx2));
    }
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}
