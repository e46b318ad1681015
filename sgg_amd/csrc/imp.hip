// Iterative message passing (IMP): the obj<->edge gather / gate / scatter loop and the GRU pointwise part.
// Reference: RelModelStanford.message_pass, sgg_models/rel_model_stanford.py:48-94.
//
// Per iteration i the reference forms, for every edge e = (s, o),
//     e_in[e] = g_sub(e) v_i[s] + g_obj(e) v_i[o]                         (:76-81)
//     ctx[n]  = sum_{s(e)=n} g_out(e) e_i[e] + sum_{o(e)=n} g_in(e) e_i[e]  (:86-91)
// and feeds e_in to the edge GRU, ctx to the node GRU.  e_in is consumed ONLY by the GRU's input projection, and the gates are
// scalars, so   W_ih e_in[e] = g_sub(e) (W_ih v_i[s]) + g_obj(e) (W_ih v_i[o]):   the 32 node rows of an image are projected once
// (P = v_i W_ih^T, [N, 3H] f32, L2-resident) and the edge GRU's gate kernel forms its input pre-activations from P[s], P[o] on the
// fly (gru_gate_proj_kernel).  e_in is never written or read, the [E, 3H] input pre-activations never exist, and three of the
// seven E-row GRU GEMMs per forward are gone.  What is left of the "gather / gate / scatter" step is its READ stream: every edge
// row once, two weighted sums per node out (imp_ctx_* kernels below).
//
// The four gate pre-activations arrive as dot products: the kernel that writes a state row (gru_gate_kernel /
// gru_gate_proj_kernel) also emits w_k . h' for the four gates while the row is in registers (16 bytes per row), so no kernel
// here needs a whole row to make a gate:  g_k(e) = sigmoid(ndots[node, k] + edots[e, k] + b_k)  with
// k = 0 sub_vert (node s), 1 obj_vert (node o), 2 out_edge (node s), 3 in_edge (node o)   (rel_model_stanford.py:41-45, 78-89).
#include <cstdlib>

#include "common.h"
#include "gemm_args.h"

namespace {

constexpr int MAXH = 512;  // one wave covers H <= 512 with 8 channels per lane

// GRU pointwise part, ATen's formulation: r=s(ir+hr) z=s(iz+hz) n=tanh(in + r*hn) h'=(h-n)*z+n  (sigmoid and tanh on v_exp_f32 /
// v_rcp_f32, common.h: the precise expf / division / tanhf made this HBM-streaming kernel VALU-bound -- 18.4 -> 14.1 us at 8 images).
// `o` (the new state, 8 channels of row m) -> h_out; with dot_w also the four gate dot products of the row AS STORED.
// H/8 lanes (a power of two <= 64) hold one row, rows never straddle a wave, whole rows are active or inactive together.
// x from the lane's DPP partner, 0 where the control selects no lane / the row mask excludes the row (VALU only: no LDS crossbar)
template <int CTRL, int ROW_MASK = 0xF> __device__ __forceinline__ float dpp_get(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xF, false));
}

// Sum of x over a group of G consecutive lanes (G a power of two <= 64, groups aligned), valid in the group's LAST lane.  DPP only:
// quad permutes, half-row / row mirrors, then row_bcast15 / row_bcast31 carry a row's total into the next row(s).  (Until round 5 this
// was a ds_bpermute butterfly; tools/gate_race.py showed that form returning a wrong partial in ~15 % of the launches of the f32 gate
// kernel while an f16 MFMA GEMM of another stream shared the chip -- the x3 mode's non-reproducible logits)
__device__ __forceinline__ float group_sum_last_lane(float x, int G) {
    if (G >= 2) x += dpp_get<0xB1>(x);             // quad_perm [1,0,3,2]
    if (G >= 4) x += dpp_get<0x4E>(x);             // quad_perm [2,3,0,1]: every lane of a quad holds the quad's sum
    if (G >= 8) x += dpp_get<0x141>(x);            // row_half_mirror: ... of 8 lanes
    if (G >= 16) x += dpp_get<0x140>(x);           // row_mirror: every lane of a 16-lane row holds the row's sum
    if (G >= 32) x += dpp_get<0x142, 0xA>(x);      // row_bcast15 into rows 1 and 3: rows 0+1, rows 2+3
    if (G >= 64) x += dpp_get<0x143, 0xC>(x);      // row_bcast31 into rows 2 and 3: row 3 holds all four
    return x;
}

template <typename T>
__device__ __forceinline__ void gru_store_with_dots(float (&o)[8], T* __restrict__ h_out, long m, int c, int H,
                                                    const float* __restrict__ dot_w, int dot_ld, float* __restrict__ dots) {
    store8(h_out + m * H + c, o);
    if (!dots) return;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = round_as<T>(o[j]);
    float p[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float w[8];
        load8(dot_w + (long)k * dot_ld + c, w);
        p[k] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) p[k] = fmaf(w[j], o[j], p[k]);
    }
    const int G = H >> 3;
#pragma unroll
    for (int k = 0; k < 4; ++k) p[k] = group_sum_last_lane(p[k], G);
    if (c == H - 8) *reinterpret_cast<f32x4*>(dots + m * 4) = f32x4{p[0], p[1], p[2], p[3]};
}

__device__ __forceinline__ void gru_cell(const float (&ir)[8], const float (&iz)[8], const float (&in_)[8], const float (&hr)[8],
                                         const float (&hz)[8], const float (&hn)[8], const float (&hp)[8], float (&o)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float r = sigmoidf_(ir[j] + hr[j]);
        const float z = sigmoidf_(iz[j] + hz[j]);
        const float n = tanh_fast(in_[j] + r * hn[j]);
        o[j] = (hp[j] - n) * z + n;
    }
}

// x and h pre-activations both given (gi = W_ih x + b_ih, gh = W_hh h + b_hh); gh == nullptr: h = 0, gh = b_hh
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
    gru_cell(ir, iz, in_, hr, hz, hn, hp, o);
    gru_store_with_dots<T>(o, h_out, m, c, H, dot_w, dot_ld, dots);
}

// The edge GRU of a message-passing iteration (rel_model_stanford.py:76-83) without its input rows:
//   gi[e] = g_sub(e) P[s(e)] + g_obj(e) P[o(e)] + b_ih,   P = v_i W_ih^T  (f32 [N, 3H]),   gh = W_hh e_i + b_hh  (f32 [E, 3H])
//   e_{i+1}[e] = GRU pointwise(gi[e], gh[e], e_i[e])
// One lane = 8 channels of one edge; the two gates are per-row scalars made from the dot products (every lane of the row computes
// them: two exps against 24 for the cell).  P rows are shared by the 2(n-1) edges of a node: L1 / L2 hits (196 KB per image).
template <typename T, typename TG>
__global__ __launch_bounds__(256) void gru_gate_proj_kernel(const TG* __restrict__ gh, const float* __restrict__ P,
                                                            const float* __restrict__ b_ih, const int* __restrict__ so,
                                                            const float* __restrict__ ndots, const float* __restrict__ edots,
                                                            const float* __restrict__ gb, const T* __restrict__ h_prev,
                                                            T* __restrict__ h_out, long total, int H,
                                                            const float* __restrict__ dot_w, int dot_ld, float* __restrict__ dots) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;  // over M*H/8
    if (i >= total) return;
    const int h8 = H >> 3;
    const long m = i / h8;
    const int c = (int)(i - m * h8) * 8;
    const int s = so[2 * m], ob = so[2 * m + 1];
    const float g_sub = sigmoidf_(ndots[4L * s] + edots[4 * m] + gb[0]);
    const float g_obj = sigmoidf_(ndots[4L * ob + 1] + edots[4 * m + 1] + gb[1]);
    float hr[8], hz[8], hn[8], hp[8], o[8], gi3[3][8];
    const TG* ghm = gh + m * 3 * H + c;       // (gh in the compute type: 16-bit halves the kernel's largest stream)
    load8(ghm, hr);
    load8(ghm + H, hz);
    load8(ghm + 2 * H, hn);
    load8(h_prev + m * H + c, hp);
    const float* ps = P + (long)s * 3 * H + c;
    const float* po = P + (long)ob * 3 * H + c;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        float a[8], b[8], bi[8];
        load8(ps + q * H, a);
        load8(po + q * H, b);
        load8(b_ih + q * H + c, bi);
#pragma unroll
        for (int j = 0; j < 8; ++j) gi3[q][j] = fmaf(g_sub, a[j], fmaf(g_obj, b[j], bi[j]));
    }
    gru_cell(gi3[0], gi3[1], gi3[2], hr, hz, hn, hp, o);
    gru_store_with_dots<T>(o, h_out, m, c, H, dot_w, dot_ld, dots);
}

// ------------------------------------------------------------------------------------------------
// The read stream for ANY edge list (CSR lists from sgg_edge_csr), one workgroup per (node, 512-channel block):
//   out[0][n] = sum over n's out-edges of g_a(e) x[e],  out[1][n] = sum over its in-edges of g_b(e) x[e]      (or their sum)
// with (a, b) = (pair, pair + 1): pair 2 = (out_edge, in_edge) -> the context sums of the forward (rel_model_stanford.py:86-91);
// pair 0 = (sub_vert, obj_vert) on x = d_gi -> the gradient of the node projection P in the backward.
// Every row is read twice (once from each list); the graphs of the path itself go through imp_ctx_sliced_kernel instead.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void imp_ctx_lists_kernel(const T* __restrict__ x, const int* __restrict__ out_ptr,
                                                            const int* __restrict__ out_ids, const int* __restrict__ in_ptr,
                                                            const int* __restrict__ in_ids, const float* __restrict__ ndots,
                                                            const float* __restrict__ edots, const float* __restrict__ gb, int pair,
                                                            int N, int H, T* __restrict__ out, int sum_ctx) {
    __shared__ float red[4][MAXH];
    const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cb = blockIdx.y * MAXH, c0 = cb + lane * 8;
    const bool act = c0 < H;
    const int W = min(MAXH, H - cb);
    float acc[2][8];
#pragma unroll
    for (int side = 0; side < 2; ++side) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[side][j] = 0.f;
        const int* ptr = side ? in_ptr : out_ptr;
        const int* ids = side ? in_ids : out_ids;
        const int beg = ptr[n], end = ptr[n + 1], k_ = pair + side;
        const float nd = ndots[4L * n + k_] + gb[k_];
        for (int k = beg + wave; k < end; k += 16) {
            float xv[4][8], gk[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kk = k + 4 * u;
                gk[u] = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[u][j] = 0.f;
                if (kk < end) {
                    const int id = ids[kk];
                    gk[u] = sigmoidf_(nd + edots[4L * id + k_]);
                    if (act) load8(x + (long)id * H + c0, xv[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[side][j] = fmaf(gk[u], xv[u][j], acc[side][j]);
        }
    }
    if (sum_ctx) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[0][j] += acc[1][j];
    }
    const int rounds = sum_ctx ? 1 : 2;
    for (int r = 0; r < rounds; ++r) {           // the four waves' partials meet in a fixed order
        if (r) __syncthreads();
        if (act) {
#pragma unroll
            for (int j = 0; j < 8; ++j) red[wave][lane * 8 + j] = acc[r][j];
        }
        __syncthreads();
        for (int c = threadIdx.x; c < W; c += 256)
            Elem<T>::st(out + ((long)r * N + n) * H + cb + c, (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]));
    }
}

// ------------------------------------------------------------------------------------------------
// The read stream on the path's own graphs, every row read ONCE (imp_ctx_sliced_kernel).  A workgroup owns (graph g, channel
// slice): LP lanes x 16 B = one PIECE of a row (64 B: 32 16-bit / 16 f32 channels at LP = 4).  Nothing needs a whole row: the
// gates are made from the dot products.
//   phase 0  the graph's vertex dots, list offsets and in-list ids -> LDS
//   phase 1  a lane group owns (subject n, part of n's out-list): per edge it loads the row piece (U edges in flight), makes
//            the two gates on 2 lanes (shared through DPP quad broadcasts, no LDS), adds g_a * row to its out-sum (registers)
//            and parks the row piece and g_b in LDS
//   phase 2  a lane group owns (object n, part of n's in-list): in-sum = sum g_b * parked piece
//   the P lane groups of a node sit in one wave: their partials are added with xor-shuffles (fixed order), part 0 stores
// HBM traffic = the rows once + 16 B of dots per edge and slice.  Needs the edge list sorted by (graph, subject) -- out-lists
// are ranges -- and the graph to fit: edges <= EMAX, nodes <= SL_NMAX.  No atomics: bit-reproducible.
// History (DESIGN.md "IMP step"): until round 3 this kernel also formed and wrote e_in = g_sub v[s] + g_obj v[o] (half of its
// bytes); the node projection (top of this file) removed that stream.
// ------------------------------------------------------------------------------------------------
constexpr int SL_THREADS = 512;   // 8 waves: two workgroups share a CU when the parked pieces leave room (measured best)
constexpr int SL_NMAX = 64;

// 16 bytes of a row: 8 16-bit or 4 f32 channels
template <typename T> struct Piece16 {
    u32x4 r;
    __device__ __forceinline__ void get(float (&x)[8]) const { unpack8<T>(r, x); }
    static __device__ __forceinline__ void store(T* p, const float (&x)[8]) { *reinterpret_cast<u32x4*>(p) = pack8<T>(x); }
};
template <> struct Piece16<float> {
    f32x4 r;
    __device__ __forceinline__ void get(float (&x)[4]) const { x[0] = r.x; x[1] = r.y; x[2] = r.z; x[3] = r.w; }
    static __device__ __forceinline__ void store(float* p, const float (&x)[4]) {
        *reinterpret_cast<f32x4*>(p) = f32x4{x[0], x[1], x[2], x[3]};
    }
};

template <int LP> struct SliceCfg;
template <> struct SliceCfg<4> { static constexpr int EMAX = 1792; };   // 112 KB
template <> struct SliceCfg<2> { static constexpr int EMAX = 3072; };   //  96 KB

// LDS bytes for graphs of at most `emax` edges (emax a multiple of 8)
template <int LP> constexpr int slice_lds_bytes(int emax) {
    return emax * (LP * 16 + 4 + 2) + SL_NMAX * 16 + (SL_NMAX + 4) * 4;
}

// value of one lane of the quad for every lane of the quad (DPP quad_perm: VALU only, no LDS crossbar)
template <int CTRL> __device__ __forceinline__ float quad_bcast(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}

template <typename T, int LP>
__global__ __launch_bounds__(SL_THREADS) void imp_ctx_sliced_kernel(const T* __restrict__ x, const int* __restrict__ so,
                                                                    const int* __restrict__ in_ptr, const int* __restrict__ in_ids,
                                                                    const int* __restrict__ img_ptr, int B, int N, int H,
                                                                    const float* __restrict__ ndots, const float* __restrict__ edots,
                                                                    const float* __restrict__ gb, int pair, T* __restrict__ out,
                                                                    int EMAX, int sum_ctx) {
    constexpr int PIECE = LP * 16, CHL = 16 / (int)sizeof(T), GROUPS = SL_THREADS / LP;
    constexpr int U = 8;                         // edges in flight per lane group: their loads are issued before any is used
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* stage = smem;                                                        // [EMAX][PIECE] parked row pieces
    float* gin = reinterpret_cast<float*>(stage + (long)EMAX * PIECE);         // [EMAX] object-side gate of each edge
    unsigned short* in_loc = reinterpret_cast<unsigned short*>(gin + EMAX);    // [EMAX] in-list entries, graph-local
    float* nd = reinterpret_cast<float*>(in_loc + EMAX);                       // [SL_NMAX][4]
    int* iptr = reinterpret_cast<int*>(nd + SL_NMAX * 4);                       // [SL_NMAX + 1], graph-local
    const int S = H * (int)sizeof(T) / PIECE;                                   // slices per graph
    const int L = xcd_remap((int)blockIdx.x, B * S);                            // a graph's slices share an XCD (dots, lists in its L2)
    const int g = L / S, slice = L - g * S;
    const int tid = threadIdx.x, sub = tid % LP, grp = tid / LP;
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
        for (int k = grp; k < Nn; k += GROUPS) {
            Piece16<T>::store(out + (long)(n0 + k) * H + col, nanv);
            if (!sum_ctx) Piece16<T>::store(out + ((long)N + n0 + k) * H + col, nanv);
        }
        return;
    }
    // the two gates of an edge sit on the lanes of a lane pair: even lane = subject side (pair), odd lane = object side (pair + 1)
    const int gk = pair + (sub & 1);
    const float bias = gb[gk];
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
    f32x4 p_nd = {0, 0, 0, 0};
    int p_ip = 0;
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
    float de[U];
    int on[U];
    auto issue = [&](int kb) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ec = e0 + max(min(kb + u, k1 - 1), 0);
            row[u].r = *reinterpret_cast<const decltype(row[u].r)*>(x + (long)ec * H + col);
            de[u] = edots[(long)ec * 4 + gk];
            on[u] = so[2 * (long)ec + 1] - n0;
        }
    };
    if (k0 < k1) issue(k0);
    // ---- phase 0: park
    if (tid < Nn) *reinterpret_cast<f32x4*>(nd + tid * 4) = p_nd;
    if (tid <= Nn) iptr[tid] = p_ip;
#pragma unroll
    for (int q = 0; q < INL; ++q)
        if (tid + q * SL_THREADS < Ee) in_loc[tid + q * SL_THREADS] = (unsigned short)p_in[q];
    for (int k = tid + INL * SL_THREADS; k < Ee; k += SL_THREADS) in_loc[k] = (unsigned short)(in_ids[i0 + k] - e0);
    __syncthreads();
    // ---- phase 1: out-lists
    float out_sum[CHL];                           // sum_ctx: the node's finished out-sum, kept for the single store after phase 2
#pragma unroll
    for (int j = 0; j < CHL; ++j) out_sum[j] = 0.f;
    if (has_node) {
        float acc[CHL];
#pragma unroll
        for (int j = 0; j < CHL; ++j) acc[j] = 0.f;
        const float ndn = nd[n * 4 + gk];         // the subject's dot (even lanes use it)
        for (int kb = k0; kb < k1; kb += U) {
            if (kb != k0) issue(kb);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int el = kb + u;
                const bool live = el < k1;       // lanes of a quad share `live`: the DPP broadcasts below stay inside a lane group
                const float vd = (sub & 1) ? nd[on[u] * 4 + gk] : ndn;
                const float gate = sigmoidf_(vd + de[u] + bias);
                float ga, gbv;
                if constexpr (LP >= 4) {
                    ga = quad_bcast<0x00>(gate);
                    gbv = quad_bcast<0x55>(gate);
                } else {                          // LP = 2: a quad holds two lane groups
                    ga = quad_bcast<0xA0>(gate);
                    gbv = quad_bcast<0xF5>(gate);
                }
                if (live) {
                    if (sub == 1) gin[el] = gbv;
                    *reinterpret_cast<decltype(row[u].r)*>(stage + (long)el * PIECE + sub * 16) = row[u].r;
                    float xx[CHL];
                    row[u].get(xx);
#pragma unroll
                    for (int j = 0; j < CHL; ++j) acc[j] = fmaf(ga, xx[j], acc[j]);
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
            Piece16<T>::store(out + (long)(n0 + n) * H + col, acc);
        }
    }
    __syncthreads();
    // ---- phase 2: in-lists, from the parked pieces
    if (has_node) {
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
                float xx[CHL];
                rw[u].get(xx);
#pragma unroll
                for (int j = 0; j < CHL; ++j) acc[j] = fmaf(gv[u], xx[j], acc[j]);
            }
        }
        for (int off = LP; off < P * LP; off <<= 1) {
#pragma unroll
            for (int j = 0; j < CHL; ++j) acc[j] += __shfl_xor(acc[j], off, 64);
        }
        if (sum_ctx) {                            // one [N,H] tensor (training: ctx is an operand of d W_ih; backward: dP)
#pragma unroll
            for (int j = 0; j < CHL; ++j) acc[j] += out_sum[j];
            if (part == 0) Piece16<T>::store(out + (long)(n0 + n) * H + col, acc);
        } else if (part == 0) {
            Piece16<T>::store(out + ((long)N + n0 + n) * H + col, acc);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Persistent LDS-DMA form of the read stream for launches that fill the chip (imp_ctx_mfma_kernel below).
// The DMA is issued from inline asm on purpose: hipcc treats a known LDS-DMA as an LDS write that may alias every later LDS
// read and puts `s_waitcnt vmcnt(0)` in front of each, which would serialise a chunk's compute behind the next chunk's DMA.
// Rule kept from round 2's experiments: a load whose completion the compiler cannot see may only target LDS (DMA), never a
// register that lives across other code.
constexpr int DM_THREADS = 1024;
constexpr int DM_EMAX = DM_THREADS;     // one edge per thread for the coalesced fetches
constexpr int DM_LDS_MAX = 160 * 1024;
constexpr int ctx_epad(int emax) { return (emax + 63) & ~63; }

typedef __attribute__((address_space(3))) char lds_char_t;

// 64 lanes x 16 bytes from per-lane global addresses into 1 KiB of LDS at `lds_base` (wave-uniform)
__device__ __forceinline__ void dma16_to_lds(const void* gptr, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_base), "v"(gptr) : "memory");
}
// 64 lanes x 4 bytes from per-lane global addresses into 256 bytes of LDS at `lds_base` (wave-uniform)
__device__ __forceinline__ void dma4_to_lds(const void* gptr, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(lds_base), "v"(gptr) : "memory");
}

// ---------------------------------------------------------------------------------------------------------------------
// The read stream as a block-sparse matrix product on the matrix cores (imp_ctx_mfma_kernel; 16-bit rows, graphs of <= 32 nodes).
//   out_a[n, :] = sum_e G_a[n, e] R[e, :],  G_a[n, e] = g_a(e) [s(e) == n]        out_b likewise with g_b, o(e)
// Persistent workgroups (one per CU) walk (graph, 64-channel slice) units.  R = the unit's row pieces [edges x 64 channels] in an LDS
// RING of NBUF buffers of EB consecutive edges, filled by LDS-DMA NBUF-1 chunks ahead of the compute, ACROSS unit boundaries, with
// counted waits: a wave tracks how many vector-memory operations it has issued (`ci`) and the value of that counter after each thing
// it will wait for; `s_waitcnt vmcnt(ci - mark)` returns as soon as that thing -- and, in issue order, everything older -- has landed.
// K = 32 edges per v_mfma_f32_16x16x32 step; the B fragment (edges x 16 channels: the reduction index is the SLOW axis in LDS) comes
// out through ds_read_b64_tr_b16, the A fragment (16 nodes x 32 edges of the gate matrix) is built in registers from the edge's
// gate and node id: a byte compare and a half-word select per element (SDWA), no unpacking, no list walking, no cross-lane reduction.
// Graph structure enters only through (s, o) of every edge, so any edge order inside a graph works.
// Roles of the 16 waves: (sum: a / b) x (node tile: 0-15 / 16-31) x (K quarter: every fourth 32-edge step); a wave builds each
// of its A fragments ONCE and multiplies it with all four 16-channel tiles (four 16x16 accumulators, kept for the whole unit); at
// the end of the unit the K quarters (and, for the summed output, the two sums) meet through the ring slot that was consumed last.
// The per-edge small arrays (4 gate dot products, s, o) and the graph's vertex dots go global -> LDS by DMA one unit ahead; a pass at
// the start of a unit (thread = edge) turns them into the two gates as 16-bit values (relative error <= 2^-9 / 2^-12 per term, below the
// rounding of the output) and byte node ids: 24 bytes per lane and K step.
// Rows are swizzled on the DMA's source side, slot' = slot ^ mf_key(row), so that the 16 rows one transposing read touches spread
// over the banks.
// Measured (bf16, B=128, 127 MB of rows, round 2, gates precomputed): 40 us; DMA and barriers alone: 31-32.  Until round 3 a second
// mode of this kernel also made and stored the edge inputs (e_in) while a chunk's DMA was landing (62-66 us for the whole step); the
// node projection removed that stream (top of this file).
typedef __attribute__((ext_vector_type(8))) __bf16 mf_bf16x8;
typedef __attribute__((ext_vector_type(4))) short mf_s16x4;
typedef __attribute__((ext_vector_type(8))) short mf_s16x8;
typedef __attribute__((address_space(3))) mf_s16x4 mf_lds_s16x4;
constexpr int MF_NBUF = 4, MF_NODES = 32;
// LDS beside the ring: raw per-edge arrays (4 dots + s + o, 32-bit each), packed (2 x 16-bit gates, 2 x byte ids), the graph's vertex dots
constexpr int mfma_fixed_bytes(int emax) { return 6 * ctx_epad(emax) * 4 + 6 * ctx_epad(emax) + 1024; }
constexpr int mfma_chunk_edges(int emax) {
    const int room = (DM_LDS_MAX - mfma_fixed_bytes(emax)) / (MF_NBUF * 128);
    const int need = (emax + 31) / 32 * 32;
    const int eb = (room < need ? room : need) / 32 * 32;
    return eb > 256 ? 256 : eb;                               // 8 K steps per chunk: two for each of the four K-quarter waves
}
static_assert(mfma_chunk_edges(992) == 256, "imp_ctx_mfma_kernel: a 992-edge graph goes through in four 256-row chunks");

// swizzle key of a chunk-local row: the 16 rows one transposing read touches (4 K blocks x 4 rows) get 8 distinct keys per row parity
__device__ __forceinline__ int mf_key(int r) { return ((r >> 1) ^ ((r >> 3) & 3)) & 7; }

__device__ __forceinline__ void dma16_to_lds_s(const void* sbase, unsigned voff, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_base), "v"(voff), "s"(sbase) : "memory");
}

template <typename T> struct MfmaGate;      // the gate matrix's element type = the rows' type
template <> struct MfmaGate<bf16_t> {
    static __device__ __forceinline__ f32x4 mfma(const u32x4& a, const mf_s16x8& b, const f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mf_bf16x8, a), __builtin_bit_cast(mf_bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct MfmaGate<f16_t> {
    static __device__ __forceinline__ f32x4 mfma(const u32x4& a, const mf_s16x8& b, const f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};

template <typename T>
__global__ __launch_bounds__(DM_THREADS) void imp_ctx_mfma_kernel(
    const T* __restrict__ e, const int* __restrict__ so, const int* __restrict__ img_ptr, int B, int N, int H, T* __restrict__ ctx2,
    int EMAX, int EB, int sum_ctx, const float* __restrict__ ndots, const float* __restrict__ edots, const float* __restrict__ gb,
    int pair) {
    static_assert(sizeof(T) == 2, "16-bit rows");
    constexpr int PIECE = 128, CHAN = 64, NBUF = MF_NBUF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int EPAD = ctx_epad(EMAX), SETW = 6 * EPAD;
    char* const ring = smem;                                                          // [NBUF][EB][128 B], rows swizzled
    // raw per-edge arrays as the DMA leaves them: the edge's four gate dot products [EPAD][4] f32, then subject, object (global ids)
    int* const sets = reinterpret_cast<int*>(ring + (long)NBUF * EB * PIECE);
    // what the K steps read, made from the raw arrays at the start of a unit: gates in the rows' 16-bit format (what the A fragment
    // holds anyway) and node ids as bytes (graph-local; 0xff = no node)
    unsigned short* const gbf = reinterpret_cast<unsigned short*>(sets + SETW);       // [2][EPAD] g_a | g_b
    unsigned char* const nid = reinterpret_cast<unsigned char*>(gbf + 2 * EPAD);      // [2][EPAD] subject | object
    float* const ndl = reinterpret_cast<float*>(nid + 2 * EPAD);                      // [64][4] vertex dots (one DMA instruction)
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
    // the per-edge arrays of h's graph -> raw: lane t of the workgroup <-> edge t; wave 0 also fetches the graph's vertex dots
    auto issue_small = [&](const Hdr& h) __attribute__((always_inline)) {
        if (wv * 64 < h.Ee) {
            const long et = h.e0 + min(tid, h.Ee - 1);
            dma16_to_lds(edots + 4 * et, __builtin_amdgcn_readfirstlane(sets_lds + (unsigned)(wv * 1024)));
            dma4_to_lds(so + 2 * et, __builtin_amdgcn_readfirstlane(sets_lds + (unsigned)((4 * EPAD + wv * 64) * 4)));
            dma4_to_lds(so + 2 * et + 1, __builtin_amdgcn_readfirstlane(sets_lds + (unsigned)((5 * EPAD + wv * 64) * 4)));
            ci += 3;
        }
        if (wv == 0 && h.Ee >= 0 && h.Nn > 0) {                 // vertex dots: lane n <-> node n
            dma16_to_lds(ndots + 4L * (h.n0 + min(lane, h.Nn - 1)), __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(lds_char_t*)ndl));
            ++ci;
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
        issue_dma(hp, lo, hi, pr);
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
    issue_small(h0);
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

    // thread t: edge t of unit hx -> what the K steps read.  Reads the raw arrays this thread's own wave DMA'd and the vertex dots wave 0
    // DMA'd: the caller has waited for them and passed a barrier.  (The per-edge arrays hold EPAD entries.)
    const float gb_a = gb[pair], gb_b = gb[pair + 1];
    auto convert = [&](const Hdr& hx) __attribute__((always_inline)) {
        if (tid < EPAD) {
            const bool live = tid < hx.Ee;
            const int sl = live ? sets[4 * EPAD + tid] - hx.n0 : 0, ol = live ? sets[5 * EPAD + tid] - hx.n0 : 0;
            const f32x4 de = reinterpret_cast<const f32x4*>(sets)[tid];
            const f32x4 ns = reinterpret_cast<const f32x4*>(ndl)[sl], no = reinterpret_cast<const f32x4*>(ndl)[ol];
            const float ga = sigmoidf_((pair ? ns.z + de.z : ns.x + de.x) + gb_a);     // subject side: sub_vert / out_edge
            const float gbv = sigmoidf_((pair ? no.w + de.w : no.y + de.y) + gb_b);    // object side: obj_vert / in_edge
            gbf[tid] = live ? (unsigned short)(H16<T>::pack(ga, 0.f) & 0xffffu) : (unsigned short)0;
            gbf[EPAD + tid] = live ? (unsigned short)(H16<T>::pack(gbv, 0.f) & 0xffffu) : (unsigned short)0;
            nid[tid] = live ? (unsigned char)sl : (unsigned char)0xff;
            nid[EPAD + tid] = live ? (unsigned char)ol : (unsigned char)0xff;
        }
    };
    // ---- first unit: its small arrays -> packed; then the second unit's small arrays are requested
    wait_mark(mark_small);
    __syncthreads();
    convert(h0);
    __syncthreads();
    if (nunits > 1) {
        issue_small(h1);
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
            wait_mark(mark_of(cr));
            __syncthreads();                                     // (D) chunk visible; every wave has left the last chunk (and unit)
            if (issued_chunks - consumed < NBUF) produce();
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
                // select per element, straight into the packed 16-bit registers (SDWA operand selects: no unpacking, no conversion)
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
                    c[t] = MfmaGate<T>::mfma(ap, bp, c[t]);
                }
            };
            // this wave's K steps of the chunk: every fourth one; two per turn on separate accumulators
            for (int ks = kq; ks < nks && blo + ks * 32 < bhi; ks += 8) {
                kstep(ks, acc[0]);
                if (ks + 4 < nks && blo + (ks + 4) * 32 < bhi) kstep(ks + 4, acc[1]);
            }
            last_slot = cr;
            cr = cr + 1 == NBUF ? 0 : cr + 1;
            ++consumed;
        }
        // ---- unit done: the four K quarters (and, for the summed ctx, the two sums) meet through LDS
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[0][t] += acc[1][t];
        lds_reads_done_barrier();                                // (E1) every wave has left the unit's last chunk
        char* const fixed_end = reinterpret_cast<char*>(ndl + 256);
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
                issue_small(h2);
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
            T* out = ctx2 + ((long)(dir && !sum_ctx ? N : 0) + hc.n0 + mt * 16 + kb * 4) * H + hc.slice * CHAN + m16;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float v[4] = {acc[0][t].x, acc[0][t].y, acc[0][t].z, acc[0][t].w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (mt * 16 + kb * 4 + i < hc.Nn) Elem<T>::st(out + (long)i * H + t * 16, v[i]);
            }
        }
        if (hc.Ee < 0) {                                         // broken promise: this graph's slice of the outputs is NaN
            u32x2 nn;
            asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(nn) : "s"(img_ptr + __builtin_amdgcn_readfirstlane(hc.g)) : "memory");
            const int n0 = (int)nn.x, Nn = (int)nn.y - n0;
            for (int k = tid; k < Nn * CHAN; k += DM_THREADS) {
                const long o = (long)(n0 + k / CHAN) * H + hc.slice * CHAN + k % CHAN;
                Elem<T>::st(ctx2 + o, __builtin_nanf(""));
                if (!sum_ctx) Elem<T>::st(ctx2 + (long)N * H + o, __builtin_nanf(""));
            }
        }
        // (the scratch slot is refilled by a DMA only after the next unit's first barrier (D): every reader has consumed its reads by then)
        h0 = h1;
        h1 = h2;
        if (loaded < nunits) {
            h2 = load_hdr(loaded);
            ++loaded;
        }
    }
}
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
int launch_ctx_sliced(const void* x, const int* so, const int* in_ptr, const int* in_ids, const int* img_ptr, int B, int N, int H,
                      const float* ndots, const float* edots, const float* gb, int pair, void* out, int max_edges, int sum_ctx,
                      hipStream_t s) {
    auto k = imp_ctx_sliced_kernel<T, LP>;
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
    hipLaunchKernelGGL(k, dim3(B * S), dim3(SL_THREADS), slice_lds_bytes<LP>(emax), s, (const T*)x, so, in_ptr, in_ids, img_ptr, B, N, H,
                       ndots, edots, gb, pair, (T*)out, emax, sum_ctx);
    return hipGetLastError() == hipSuccess ? SGG_OK : SGG_ERR_LAUNCH;
}

// the matrix-core form (16-bit rows, <= 32 nodes and <= 1024 edges per graph, rows a multiple of 128 bytes); 1 when it does not apply
template <typename T>
int launch_ctx_mfma(const void* x, const int* so, const int* img_ptr, int B, int N, int H, const float* ndots, const float* edots,
                    const float* gb, int pair, void* out, int max_edges, int sum_ctx, int max_wgs, hipStream_t s) {
    const int emax = (max(max_edges, 32) + 31) & ~31;
    int eb = mfma_chunk_edges(emax);
    const char* ebs = getenv("SGG_IMP_EB");     // tests: small chunks (many per unit on small graphs)
    if (ebs && atoi(ebs) >= 32) eb = min(eb, atoi(ebs) & ~31);
    if (eb < 32) return 1;
    auto k = imp_ctx_mfma_kernel<T>;
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, DM_LDS_MAX) != hipSuccess)
            return SGG_ERR_LAUNCH;
        configured = true;
    }
    const int units = B * (H * 2 / 128);
    int grid = min(units, 256);
    if (max_wgs > 0) grid = min(grid, max_wgs);
    const int smem = MF_NBUF * eb * 128 + mfma_fixed_bytes(emax) + (eb * 128 >= 32768 ? 0 : 32768 + 16);
    if (smem > DM_LDS_MAX) return 1;
    hipLaunchKernelGGL(k, dim3(grid), dim3(DM_THREADS), smem, s, (const T*)x, so, img_ptr, B, N, H, (T*)out, emax, eb, sum_ctx, ndots,
                       edots, gb, pair);
    return hipGetLastError() == hipSuccess ? SGG_OK : SGG_ERR_LAUNCH;
}
}  // namespace

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
    if (g_dtype == SGG_F32) {
        SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL((gru_gate_kernel<float, T>), grid, blk, 0, s, (const float*)gi, (const float*)gh, b_hh,
                                                (const T*)h_prev, (T*)h_out, total, H, dot_w, dot_ld, dots));
    } else if (g_dtype == dtype) {
        SGG_FOR_DTYPE16(dtype, hipLaunchKernelGGL((gru_gate_kernel<T, T>), grid, blk, 0, s, (const T*)gi, (const T*)gh, b_hh,
                                                  (const T*)h_prev, (T*)h_out, total, H, dot_w, dot_ld, dots));
    } else {
        return SGG_ERR_DTYPE;
    }
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_gru_gate_proj_fwd(const void* gh, const float* P, const float* b_ih, const int* so, const float* node_dots,
                                     const float* edge_dots, const float* gate_b, const void* h_prev, void* h_out, int M, int H,
                                     const float* dot_w, int dot_ld, float* dots, int dtype, int gh_dtype, void* stream) {
    if (M == 0) return SGG_OK;
    if (!gh || !P || !b_ih || !so || !node_dots || !edge_dots || !gate_b || !h_prev || !h_out || M < 0 || H <= 0 || (H & 7)) return SGG_ERR_ARG;
    if (gh_dtype != SGG_F32 && gh_dtype != dtype) return SGG_ERR_DTYPE;      // gh: f32, or the state's own element type
    if (dots) {
        const int h8 = H / 8;
        if (!dot_w || dot_ld < H || h8 > 64 || (h8 & (h8 - 1))) return SGG_ERR_ARG;
    }
    const long total = (long)M * (H / 8);
    const dim3 grid((unsigned)((total + 255) / 256)), blk(256);
    hipStream_t s = (hipStream_t)stream;
    if (gh_dtype == SGG_F32) {
        SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL((gru_gate_proj_kernel<T, float>), grid, blk, 0, s, (const float*)gh, P, b_ih, so, node_dots, edge_dots, gate_b,
                                                (const T*)h_prev, (T*)h_out, total, H, dot_w, dot_ld, dots));
    } else {
        SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL((gru_gate_proj_kernel<T, T>), grid, blk, 0, s, (const T*)gh, P, b_ih, so, node_dots, edge_dots, gate_b,
                                                (const T*)h_prev, (T*)h_out, total, H, dot_w, dot_ld, dots));
    }
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

extern "C" int sgg_graph_ptr(const int64_t* im_inds, int N, int B, const int* out_ptr, int* img_ptr, void* stream) {
    if (!im_inds || !img_ptr || !out_ptr || N < 0 || B < 0) return SGG_ERR_ARG;
    hipLaunchKernelGGL(graph_ptr_kernel, dim3((B + 256) / 256), dim3(256), 0, (hipStream_t)stream, im_inds, N, B, out_ptr, img_ptr);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

// (graph, 64-channel slice) units from which sgg_imp_ctx_fwd hands 16-bit graphs of <= 32 nodes to the persistent matrix-core kernel
// (256 workgroups, one per CU: eight units each).  Measured (bf16, complete 32-node graphs, H = 512; sliced / matrix-core, us):
// 8 images 7.1 / 13.0, 32: 12.7 / 13.8, 128: 39.0 / 41.1, 512: 172 / 152 -- the short-lived kernel's 64-byte requests stream rows
// that sit in the Infinity Cache faster, the persistent kernel's 128-byte DMA pieces win once the rows come from HBM.
constexpr int IMP_MFMA_MIN_UNITS = 2048;
extern "C" int sgg_imp_ctx_mfma_min_units(void) { return IMP_MFMA_MIN_UNITS; }

// largest per-graph edge count the sliced kernel takes at this row width (0: rows too narrow for any slicing)
extern "C" int sgg_imp_sliced_capacity(int H, int dtype) {
    const int row = H * sgg_elem_size(dtype);
    if (row % 32 || H <= 0) return 0;
    return SliceCfg<2>::EMAX;   // the narrowest pieces hold the most edges
}

// The read stream of one message-passing step:  out[0][n] = sum_{s(e)=n} g_a(e) x[e],  out[1][n] = sum_{o(e)=n} g_b(e) x[e]  (sum_ctx: their
// sum in out[0]) with the gates made from the dot products (pair 2: out_edge / in_edge = the context sums; pair 0: sub_vert / obj_vert =
// the gradient of the node projection).  img_ptr (sgg_graph_ptr) + max_edges / max_nodes promise graphs sorted by (graph, subject) that
// fit the sliced kernels; without them (or above their capacity) the CSR lists are walked (any edge list).
extern "C" int sgg_imp_ctx_fwd(const void* x, const int* so, const int* out_ptr, const int* out_ids, const int* in_ptr, const int* in_ids,
                               const int* img_ptr, int B, int N, int E, int H, const float* node_dots, const float* edge_dots,
                               const float* gate_b, int pair, void* out, int max_edges, int max_nodes, int sum_ctx, int dtype,
                               void* stream) {
    if (N == 0) return SGG_OK;
    if (!x || !out_ptr || !out_ids || !in_ptr || !in_ids || !node_dots || !edge_dots || !gate_b || !out || N < 0 || E < 0 || H <= 0 ||
        (H & 7) || (pair != 0 && pair != 2))
        return SGG_ERR_ARG;
    if (!sgg_is_dtype(dtype)) return SGG_ERR_DTYPE;
    hipStream_t s = (hipStream_t)stream;
    const int row = H * sgg_elem_size(dtype);
    const char* form = getenv("SGG_IMP_CTX");            // tests / kernel experiments: 'l' lists, 's' sliced, 'm' matrix-core wherever they apply
    const char* mw = getenv("SGG_IMP_MAX_WGS");          // tests: several units per persistent workgroup on small batches
    const bool sliced_ok = img_ptr && so && B > 0 && row % 32 == 0 && max_nodes <= SL_NMAX && max_edges <= SliceCfg<2>::EMAX;
    if (sliced_ok && !(form && form[0] == 'l')) {
        const bool mfma_ok = dtype != SGG_F32 && row % 128 == 0 && max_nodes <= MF_NODES && max_edges <= DM_EMAX && (long)E * row < 0xffff0000L;
        if (mfma_ok && ((form && form[0] == 'm') || (!form && (long)B * (row / 128) >= IMP_MFMA_MIN_UNITS))) {
            int rc = 1;
            SGG_FOR_DTYPE16(dtype, rc = launch_ctx_mfma<T>(x, so, img_ptr, B, N, H, node_dots, edge_dots, gate_b, pair, out, max_edges, sum_ctx,
                                                           mw ? atoi(mw) : 0, s));
            if (rc != 1) return rc;                          // 1: these graphs do not fit it
        }
        // piece width: 64-byte pieces (LP = 4) first -- a 992-edge graph then parks 62 KB and two workgroups fit a CU (one's loads run
        // under the other's accumulate phases); 32-byte pieces for graphs above LP = 4's capacity
        const int lp = (row % 64 == 0 && max_edges <= SliceCfg<4>::EMAX) ? 4 : 2;
        if (lp == 4) {
            SGG_FOR_DTYPE(dtype, return (launch_ctx_sliced<T, 4>(x, so, in_ptr, in_ids, img_ptr, B, N, H, node_dots, edge_dots, gate_b, pair, out,
                                                                 max_edges, sum_ctx, s)));
        }
        SGG_FOR_DTYPE(dtype, return (launch_ctx_sliced<T, 2>(x, so, in_ptr, in_ids, img_ptr, B, N, H, node_dots, edge_dots, gate_b, pair, out,
                                                             max_edges, sum_ctx, s)));
    }
    const dim3 grid(N, (H + MAXH - 1) / MAXH), blk(256);
    SGG_FOR_DTYPE(dtype, hipLaunchKernelGGL(imp_ctx_lists_kernel<T>, grid, blk, 0, s, (const T*)x, out_ptr, out_ids, in_ptr, in_ids, node_dots,
                                            edge_dots, gate_b, pair, N, H, (T*)out, sum_ctx));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}
