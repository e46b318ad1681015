// Iterative message passing (IMP): the obj<->edge gather / gate / scatter loop and the GRU pointwise part.
// Reference: RelModelStanford.message_pass, sgg_models/rel_model_stanford.py:48-94.
//
// HBM view (per iteration, E edges, N nodes, H channels, s bytes/element): read e_i (E*H*s) once for the edge
// kernel, gather v rows (N*H*s, each re-used ~2(n-1) times -> L2), write e_in (E*H*s), write ctx (N*H*s).
// A lane owns 8 consecutive channels, so every row access is 16-byte (bf16) / 32-byte (f32) pieces of one
// contiguous H-row: a wave reads/writes one whole row (1-2 KiB for H=512) per instruction group.
#include "common.h"

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
                                                       T* __restrict__ h_out, long total, int H) {
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
                                int H, int g_dtype, int dtype, void* stream) {
    if (M == 0) return SGG_OK;
    if (!gi || !h_out || M < 0 || H <= 0 || (H & 7)) return SGG_ERR_ARG;
    if (gh ? !h_prev : !b_hh) return SGG_ERR_ARG;
    const long total = (long)M * (H / 8);
    const dim3 grid((unsigned)((total + 255) / 256)), blk(256);
    hipStream_t s = (hipStream_t)stream;
    if (g_dtype == SGG_F32 && dtype == SGG_BF16)
        hipLaunchKernelGGL((gru_gate_kernel<float, bf16_t>), grid, blk, 0, s, (const float*)gi, (const float*)gh, b_hh, (const bf16_t*)h_prev, (bf16_t*)h_out, total, H);
    else if (g_dtype == SGG_F32 && dtype == SGG_F32)
        hipLaunchKernelGGL((gru_gate_kernel<float, float>), grid, blk, 0, s, (const float*)gi, (const float*)gh, b_hh, (const float*)h_prev, (float*)h_out, total, H);
    else if (g_dtype == SGG_BF16 && dtype == SGG_BF16)
        hipLaunchKernelGGL((gru_gate_kernel<bf16_t, bf16_t>), grid, blk, 0, s, (const bf16_t*)gi, (const bf16_t*)gh, b_hh, (const bf16_t*)h_prev, (bf16_t*)h_out, total, H);
    else
        return SGG_ERR_DTYPE;
    SGG_CHECK_LAUNCH();
    return SGG_OK;
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
