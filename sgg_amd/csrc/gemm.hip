// MFMA tile kernels for gfx950: nn.Linear-shaped GEMM and 3x3 convolution as implicit GEMM.
//
//   C[m][n] = post_scale[n] * act( sum_k A[m][k] * W[n][k] + bias[n] ) + post_shift[n]
//
// Both operands are K-contiguous (activations [M,K], weights [N,K] = nn.Linear / [Cout][3][3][Cin]), so one
// 128-byte LDS row holds a K-slab of 64 bf16 or 32 f32 for one m (or n).  Tiles are staged HBM->LDS with
// global_load_lds_dwordx4 (no VGPR round trip); the LDS image is lane-linear, so the bank swizzle
// (16-byte slot ^= (row>>1)&7, conflict-free for the 32-row ds_read_b128 fragment reads) is applied on the
// per-lane SOURCE address and again on the read.  Each of the 4 waves owns a 64x64 output tile = 2x2 MFMA
// 32x32 tiles (v_mfma_f32_32x32x16_bf16, or v_mfma_f32_32x32x2_f32 for the exact-fp32 parity mode), computed
// transposed (weights as the MFMA A operand) so that a lane holds 4 consecutive n.  The epilogue goes through
// LDS so that global stores are row-contiguous 16-byte pieces.
//
// Replaces (reference): nn.Linear calls at sgg_models/rel_model_stanford.py:29-37,103-107 and
// sgg_models/rel_model_base.py:110-111; [3P] cuDNN 3x3 convs of vgg16.features (rel_model_base.py:184).
#include <stdlib.h>

#include "gemm_args.h"

int sgg_launch_pingpong(const GemmArgs& g, int dt, bool conv, hipStream_t s);  // gemm256.hip
int sgg_launch_pingpong_tn(const GemmArgs& g, int dt, int splits, hipStream_t s);                 // gemm256.hip: TN form (g.nt in 32-row K-tiles)
int sgg_launch_pingpong_splitk(const GemmArgs& g, int dt, int splits, hipStream_t s);  // gemm256.hip
int sgg_launch_conv1_block(const float* img, const void* w1, const float* b1, const void* w2, const float* b2, void* out, int out_pad, int B,
                           int H, int W, int dt, int pool, hipStream_t s);  // conv_spatial.hip
int sgg_launch_conv1_pack(const float* w1, void* frags, int dt, hipStream_t s);
int sgg_launch_conv_pp(const void* in, const void* w, const float* bias, void* out, int out_pad, int B, int H, int W, int Cin, int Cout,
                       int dt, int pool, int form, int tw, hipStream_t s);  // conv_pp.hip
int sgg_launch_conv_pp_x3(const void* in, const void* w3, const float* bias, void* out, int out_pad, int B, int H, int W, int Cpl, int Cout,
                          int pool, hipStream_t s);   // conv_pp.hip
int sgg_launch_conv_spatial_x3(const void* in, const void* w3, const float* bias, void* out, int out_pad, int B, int H, int W, int Cpl, int Cout,
                               int pool, hipStream_t s);   // conv_spatial.hip
int sgg_launch_conv_spatial(const void* in, const void* w, const float* bias, void* out, int out_pad, int B, int H, int W,
                            int Cin, int Cout, int dt, int pool, hipStream_t s);            // conv_spatial.hip

namespace {

constexpr int ROWB = 128;  // bytes of K per LDS row

// WM x WN waves of 64x64; block tile (64*WM) x (64*WN); 256 threads.
template <int DT, int WM, int WN, bool CONV>
__global__ __launch_bounds__(256) void mfma_tile_kernel(const GemmArgs g) {
    static_assert(WM * WN == 4, "4 waves");
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    constexpr int LA = 2 * WM, LB = 2 * WN;  // 8-row load instructions per wave per tile
    constexpr int ESZ = DT == SGG_F32 ? 4 : 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WN, wc = wave % WN;

    const int tilesM = (g.M - g.m_base + BM - 1) / BM, tilesN = (g.N + BN - 1) / BN;
    int tm, tn;
    tile_coords(blockIdx.x, tilesM, tilesN, tm, tn);
    const int m0 = g.m_base + tm * BM, n0 = tn * BN;

    // ---- per-lane source pointers (16-byte chunk of a 128-byte row, swizzled)
    const int lrow = lane >> 3;  // row inside an 8-row load
    // per-lane 32-bit offsets against the uniform operand bases (glds16_su: scalar-base LDS-DMA, no vector ALU per piece)
    unsigned arp[LA];
    unsigned arp2[LA];
    unsigned brp[LB];
    unsigned brp2[LB];
#pragma unroll
    for (int j = 0; j < LA; ++j) {
        const int r = (wave * LA + j) * 8 + lrow;
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        const int m = min(m0 + r, g.M - 1);
        if constexpr (CONV) {
            const int hw = g.H * g.W;
            const int b = m / hw, rem = m - b * hw;
            const int y = rem / g.W, x = rem - y * g.W;
            arp[j] = (unsigned)(((long)(b * (g.H + 2) + y) * (g.W + 2) + x) * (g.cin_px ? g.cin_px : g.Cin) * ESZ + chunk * 16);
            arp2[j] = 0;
        } else {
            arp[j] = (unsigned)((long)m * g.lda_b + chunk * 16);
            arp2[j] = g.A2 ? (unsigned)((long)m * g.lda2_b + chunk * 16) : 0;
        }
    }
#pragma unroll
    for (int j = 0; j < LB; ++j) {
        const int r = (wave * LB + j) * 8 + lrow;
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        const int n = min(n0 + r, g.N - 1);
        brp[j] = (unsigned)((long)n * g.ldw_b + chunk * 16);
        brp2[j] = (!CONV && g.W2) ? (unsigned)((long)n * g.ldw2_b + chunk * 16) : 0;
    }
    const int tpc = CONV ? (g.Cin * ESZ) / ROWB : 1;  // k-tiles per conv tap

    auto stage = [&](int kt, int buf) {
        char* sa = smem + buf * STAGE;
        char* sb = sa + A_BYTES;
        long koff, woff;
        bool seg2 = false, wseg2 = false;
        if (g.x3t) {
            // pair operands [hi | lo]: segment 0 = (A hi, W hi), 1 = (A hi, W lo), 2 = (A lo, W hi); in conv mode per tap
            if constexpr (CONV) {
                const int tap = kt / (3 * tpc), r = kt - tap * 3 * tpc;
                const int seg = r / tpc, c0 = r - seg * tpc;
                const int ky = tap / 3, kx = tap - ky * 3;
                koff = ((long)(ky * (g.W + 2) + kx) * g.cin_px + (seg == 2 ? g.Cin : 0)) * ESZ + c0 * ROWB;
                woff = ((long)tap * g.cin_px + (seg == 1 ? g.Cin : 0)) * ESZ + c0 * ROWB;
            } else {
                int seg, kk;
                x3_tile(g, kt, seg, kk);
                koff = (long)((seg == 2 ? g.x3t : 0) + kk) * ROWB;
                woff = (long)((seg == 1 ? g.x3t : 0) + kk) * ROWB;
            }
        } else {
            if constexpr (CONV) {
                const int tap = kt / tpc, c0 = kt - tap * tpc;
                const int ky = tap / 3, kx = tap - ky * 3;
                koff = ((long)(ky * (g.W + 2) + kx) * g.Cin) * ESZ + c0 * ROWB;
            } else {
                seg2 = kt >= g.nt1;
                koff = (long)(seg2 ? kt - g.nt1 : kt) * ROWB;
            }
            wseg2 = !CONV && g.W2 && kt >= g.nt1;
            woff = (long)(wseg2 ? kt - g.nt1 : kt) * ROWB;
        }
        const char* ua = uniform_ptr((seg2 ? g.A2 : g.A) + koff);
#pragma unroll
        for (int j = 0; j < LA; ++j) glds16_su(ua, seg2 ? arp2[j] : arp[j], sa + (wave * LA + j) * 8 * ROWB);
        const char* ub = uniform_ptr((wseg2 ? g.W2 : g.Wt) + woff);
#pragma unroll
        for (int j = 0; j < LB; ++j) glds16_su(ub, wseg2 ? brp2[j] : brp[j], sb + (wave * LB + j) * 8 * ROWB);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read offsets: row = base + (lane&31), logical slot = 2*s + (lane>>5)
    const int fr = lane & 31, fh = lane >> 5;
    int aoff[2], boff[2], akey[2], bkey[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra = wr * 64 + i * 32 + fr, rb = wc * 64 + i * 32 + fr;
        aoff[i] = ra * ROWB;
        boff[i] = rb * ROWB;
        akey[i] = (ra >> 1) & 7;
        bkey[i] = (rb >> 1) & 7;
    }

    // split-K (gridDim.y > 1): this block reduces k-tiles [kt0, kt1) into its own fp32 partial output
    const int nsplit = gridDim.y;
    const int kt0 = (int)((long)g.nt * blockIdx.y / nsplit), kt1 = (int)((long)g.nt * (blockIdx.y + 1) / nsplit);
    stage(kt0, kt0 & 1);
    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < kt1) {
            stage(kt + 1, buf ^ 1);
            wait_vmcnt<LA + LB>();  // tile kt landed (this wave's part); tile kt+1 stays in flight
        } else {
            wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        const char* sa = smem + buf * STAGE;
        const char* sb = sa + A_BYTES;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int slot = 2 * s + fh;
            u32x4 av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                av[i] = *reinterpret_cast<const u32x4*>(sa + aoff[i] + ((slot ^ akey[i]) << 4));
                bv[i] = *reinterpret_cast<const u32x4*>(sb + boff[i] + ((slot ^ bkey[i]) << 4));
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    if constexpr (DT != SGG_F32) {
                        acc[mi][ni] = mfma_32x32x16<DT>(bv[ni], av[mi], acc[mi][ni]);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                __uint_as_float(bv[ni][q]), __uint_as_float(av[mi][q]), acc[mi][ni], 0, 0, 0);
                    }
                }
        }
        lds_reads_done_barrier();          // the stage this tile was read from is refilled right after (common.h)
    }

    // ---- epilogue through LDS: per wave a [32][64] f32 staging tile (row stride 272 B), two halves (mi)
    constexpr int ESTRIDE = 272;
    char* est = smem + wave * (32 * ESTRIDE);
    const bool vec_ok = CONV || ((g.ldc & 7) == 0);
    GemmArgs ge = g;
    ge.C = g.C + (long)blockIdx.y * g.splitk_stride;
    // The staging tile is private to the wave (est = smem + wave * ...): after ONE workgroup barrier (every wave has
    // finished reading operand tiles out of this memory) the wave's own LDS write -> read order is all that is needed.
    const ChanVec8 cv = load_chanvec8(g, n0 + wc * 64 + (lane & 7) * 8);   // the lane's 8 channels: the same in every store below
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = {acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]};
                *reinterpret_cast<f32x4*>(est + fr * ESTRIDE + (ni * 32 + 8 * q + 4 * fh) * 4) = v;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own LDS writes landed (same wave reads them back)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int rl = (lane >> 3) + 8 * it, cl = (lane & 7) * 8;
            const int m = m0 + wr * 64 + mi * 32 + rl;
            const int n = n0 + wc * 64 + cl;
            if (m >= g.M || n >= g.N) continue;
            float v[8];
            const f32x4 lo = *reinterpret_cast<const f32x4*>(est + rl * ESTRIDE + cl * 4);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(est + rl * ESTRIDE + cl * 4 + 16);
            v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
            epilogue_store8(ge, cv, v, m, n, out_offset<CONV>(g, m, n), vec_ok);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// TN form for weight gradients: C[n][k] = sum_m A[m][n] * B[m][k]  (A = dY [Mred, N], B = X [Mred, K], both row-major,
// the reduction index m is the SLOW axis of both).  The NT kernel above needs A^T and B^T materialised first (a
// read+write pass over each operand per GEMM: ~1 ms of the train step); here the K-tile is 64 reduction rows of each
// operand staged AS THEY LIE -- per operand two [64 rows][128 B] column halves in the usual swizzled layout, filled by
// the same lane-linear global->LDS DMA -- and the MFMA fragments (8 consecutive reduction elements of one column per
// lane) come out of LDS through ds_read_b64_tr_b16, the gfx950 transposing read: within a 16-lane group lane p hands in
// the address of 4 consecutive bf16 and lane i receives element i%4 of lanes i/4, i/4+4, i/4+8, i/4+12 (probed on
// hardware: tools/exp/tr16_probe.hip).  Which 4 stage rows form one read is free as long as both operands agree
// (a sum does not care about the order of its terms): rows {a, a+1, a+8, a+9} differ in row parity (128 B apart) and in
// bit 2 of the swizzle key, so the 4 x 64 B a 32-lane pass touches cover all 64 banks exactly once.
// 16-bit operands only (tr_b16 moves 16-bit elements).  128x128 output tile, 4 waves of 64x64, split-K over the reduction rows.
// ------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void mfma_tile_tn_kernel(const GemmArgs g) {
    constexpr int HALF = 64 * ROWB;                 // one [64 rows][128 B] column half
    constexpr int STAGE = 4 * HALF;                 // A: 2 halves (128 n), B: 2 halves (128 k)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;        // wave's 64 n / 64 k = column half wr of A / wc of B
    const int tilesM = g.M / 128, tilesN = g.N / 128;
    int tm, tn;
    tile_coords(blockIdx.x, tilesM, tilesN, tm, tn);
    const int n0 = tm * 128, k0 = tn * 128;         // output rows (A columns) / output columns (B columns)

    // ---- DMA sources: instruction i of a half covers stage rows 8i..8i+7; lane -> (row 8i + lane/8, 16-byte chunk)
    const int lrow = lane >> 3;
    unsigned asrc[4];
    unsigned bsrc[4];
    int ldst[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = wave * 4 + j;               // 16 instructions per operand: half = idx / 8, row block = idx % 8
        const int half = idx >> 3, r = (idx & 7) * 8 + lrow;
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        asrc[j] = (unsigned)((long)r * g.lda_b + (long)(n0 + half * 64) * 2 + chunk * 16);
        bsrc[j] = (unsigned)((long)r * g.ldw_b + (long)(k0 + half * 64) * 2 + chunk * 16);
        ldst[j] = half * HALF + (idx & 7) * 8 * ROWB;
    }
    auto stage = [&](int kt, int buf) {
        char* sa = smem + buf * STAGE;
        char* sb = sa + 2 * HALF;
        const char* ua = uniform_ptr(g.A + (long)kt * 64 * g.lda_b);
        const char* ub = uniform_ptr(g.Wt + (long)kt * 64 * g.ldw_b);
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16_su(ua, asrc[j], sa + ldst[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16_su(ub, bsrc[j], sb + ldst[j]);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- transposing fragment reads.  lane: kg = lane>>5 (which 8 of the k-step's 16 reduction rows), g16 = (lane>>4)&1
    // (columns 16*g16.. of the 32-column block), p = lane&15 -> stage row offset (0,1,8,9)[p>>2], 4 columns 4*(p&3)..
    const int kg = lane >> 5, g16 = (lane >> 4) & 1, p = lane & 15;
    const int rsel = ((p >> 2) & 1) + ((p >> 3) << 3);          // (0,1,8,9)[p>>2]
    const int nsplit = gridDim.y;
    const int kt0 = (int)((long)g.nt * blockIdx.y / nsplit), kt1 = (int)((long)g.nt * (blockIdx.y + 1) / nsplit);
    // workgroup-RELATIVE LDS byte offset for the ds_read asm: through an address_space(3) pointer.  (The low 32 bits of the
    // generic pointer are only right for the first workgroup on a CU: a co-resident one got garbage.)
    const unsigned lds0 = (unsigned)(unsigned long)(lds_void_t*)smem;
    stage(kt0, kt0 & 1);
    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < kt1) {
            stage(kt + 1, buf ^ 1);
            wait_vmcnt<8>();
        } else {
            wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        const unsigned sa = lds0 + buf * STAGE + wr * HALF, sb = lds0 + buf * STAGE + 2 * HALF + wc * HALF;
        // The four k-steps are software-pipelined inside the wave: the transposing reads of step s+1 are issued BEFORE the MFMAs of
        // step s and waited for after them (two register sets), so LDS latency runs under the wave's own matrix work.
        unsigned long long a_lo[2][2], a_hi[2][2], b_lo[2][2], b_hi[2][2];      // [set][i]
#define SGG_TN_READ(SET, S)                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                          \
        _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                                      \
            const int row = 16 * (S) + 2 * (2 * kg + h) + rsel;                                                               \
            const int col = i * 32 + g16 * 16 + 4 * (p & 3);                                                                  \
            const unsigned off = row * ROWB + ((((col >> 3) ^ ((row >> 1) & 7)) << 4) | (((col >> 2) & 1) << 3));             \
            if (h == 0) {                                                                                                     \
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(a_lo[SET][i]) : "v"(sa + off) : "memory");                    \
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(b_lo[SET][i]) : "v"(sb + off) : "memory");                    \
            } else {                                                                                                          \
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(a_hi[SET][i]) : "v"(sa + off) : "memory");                    \
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(b_hi[SET][i]) : "v"(sb + off) : "memory");                    \
            }                                                                                                                 \
        }                                                                                                                     \
    }
        // the waitcnt TAKES the eight results of a set as in/out operands: the compiler only sees register outputs of the asm reads
        // and would otherwise be free to schedule a use of them in front of this wait (it did, once two workgroups shared a CU)
#define SGG_TN_WAIT(SET)                                                                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                       \
                 : "+v"(a_lo[SET][0]), "+v"(a_lo[SET][1]), "+v"(a_hi[SET][0]), "+v"(a_hi[SET][1]), "+v"(b_lo[SET][0]),        \
                   "+v"(b_lo[SET][1]), "+v"(b_hi[SET][0]), "+v"(b_hi[SET][1])                                                 \
                 :                                                                                                            \
                 : "memory");
#define SGG_TN_MFMA(SET)                                                                                                      \
    _Pragma("unroll") for (int mi = 0; mi < 2; ++mi) _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) {                      \
        const u32x4 av = {(unsigned)a_lo[SET][mi], (unsigned)(a_lo[SET][mi] >> 32), (unsigned)a_hi[SET][mi],                  \
                          (unsigned)(a_hi[SET][mi] >> 32)};                                                                   \
        const u32x4 bv = {(unsigned)b_lo[SET][ni], (unsigned)(b_lo[SET][ni] >> 32), (unsigned)b_hi[SET][ni],                  \
                          (unsigned)(b_hi[SET][ni] >> 32)};                                                                   \
        acc[mi][ni] = mfma_32x32x16<DT>(bv, av, acc[mi][ni]);                                                                \
    }
        SGG_TN_READ(0, 0)
        SGG_TN_WAIT(0)
        SGG_TN_READ(1, 1)
        __builtin_amdgcn_sched_barrier(0);
        SGG_TN_MFMA(0)
        __builtin_amdgcn_sched_barrier(0);
        SGG_TN_WAIT(1)
        SGG_TN_READ(0, 2)
        __builtin_amdgcn_sched_barrier(0);
        SGG_TN_MFMA(1)
        __builtin_amdgcn_sched_barrier(0);
        SGG_TN_WAIT(0)
        SGG_TN_READ(1, 3)
        __builtin_amdgcn_sched_barrier(0);
        SGG_TN_MFMA(0)
        __builtin_amdgcn_sched_barrier(0);
        SGG_TN_WAIT(1)
        SGG_TN_MFMA(1)
#undef SGG_TN_READ
#undef SGG_TN_WAIT
#undef SGG_TN_MFMA
        __builtin_amdgcn_s_barrier();
    }

    // ---- epilogue (as the NT kernel): per wave a [32][64] f32 staging tile, rows = output rows n, 64 output columns k
    constexpr int ESTRIDE = 272;
    char* est = smem + wave * (32 * ESTRIDE);
    GemmArgs ge = g;
    ge.C = g.C + (long)blockIdx.y * g.splitk_stride;
    const ChanVec8 cv = load_chanvec8(ge, k0 + wc * 64 + (lane & 7) * 8);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int efr = lane & 31, efh = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = {acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]};
                *reinterpret_cast<f32x4*>(est + efr * ESTRIDE + (ni * 32 + 8 * q + 4 * efh) * 4) = v;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int rl = (lane >> 3) + 8 * it, cl = (lane & 7) * 8;
            const int m = n0 + wr * 64 + mi * 32 + rl;
            const int n = k0 + wc * 64 + cl;
            float v[8];
            const f32x4 lo = *reinterpret_cast<const f32x4*>(est + rl * ESTRIDE + cl * 4);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(est + rl * ESTRIDE + cl * 4 + 16);
            v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
            epilogue_store8(ge, cv, v, m, n, (long)m * g.ldc + n, true);
        }
    }
}

template <int DT, int WM, int WN, bool CONV>
int launch(const GemmArgs& g, hipStream_t s, int splits = 1) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int smem = 2 * (BM + BN) * ROWB;
    static_assert(smem >= 4 * 32 * 272, "epilogue staging fits");
    const int tilesM = (g.M - g.m_base + BM - 1) / BM, tilesN = (g.N + BN - 1) / BN;
    auto k = mfma_tile_kernel<DT, WM, WN, CONV>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return SGG_ERR_LAUNCH;
        attr_done = true;
    }
    hipLaunchKernelGGL(k, dim3(tilesM * tilesN, splits), dim3(256), smem, s, g);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

// Kernel choice: the 256x256 ping-pong kernel when the problem fills the chip with 256-wide tiles
// (N >= 256, M large); 256x64 tiles for narrow N; 128x128 otherwise.  SGG_GEMM_FORCE=128 disables the big kernel.
constexpr int N_CU_CHIP = 256;   // MI355X: one 256x256 ping-pong tile per CU and round
template <bool CONV>
int dispatch(GemmArgs g, int dt, hipStream_t s) {
    static const char* force = getenv("SGG_GEMM_FORCE");
    const bool allow256 = !(force && force[0] == '1' && force[1] == '2');
    const long tiles256 = (long)((g.M + 255) / 256) * ((g.N + 255) / 256);
    if (allow256 && g.N >= 256 && tiles256 >= 128) {
        if constexpr (CONV) {
            // One 256x256 tile per CU and round: a tile count a little above a multiple of the CU count costs a whole round for a few
            // tiles (conv4_1 / conv4_2 at 8 frames: 172 x 2 = 344 tiles = 1.34 rounds).  When the last round would be less than
            // half full and is made of whole tile rows, those pixels go to the 128x128 kernel (two workgroups per CU: 4 x rem tiles
            // fill the chip once at most) and the main launch is an exact number of rounds.  Same K order in both kernels.
            static const char* notail = getenv("SGG_CONV_NOTAIL");
            static const char* tmax = getenv("SGG_CONV_TAIL_MAX");        // experiments: largest last round (tiles) that becomes a tail
            const long tail_max = tmax ? atol(tmax) : 112;
            const int tN = (g.N + 255) / 256;
            const long rem = tiles256 % N_CU_CHIP;
            if (dt != SGG_F32 && !notail && g.m_base == 0 && tiles256 > N_CU_CHIP && rem > 0 && rem <= tail_max && (tiles256 - rem) % tN == 0) {
                GemmArgs tail = g;
                tail.m_base = (int)((tiles256 - rem) / tN) * 256;
                g.M = tail.m_base;
                g.nt *= 2;
                g.nt1 *= 2;
                g.x3t *= 2;
                g.x3c *= 2;
                const int rc = sgg_launch_pingpong(g, dt, CONV, s);
                if (rc != SGG_OK) return rc;
                return dt == SGG_BF16 ? launch<SGG_BF16, 2, 2, CONV>(tail, s) : launch<SGG_F16, 2, 2, CONV>(tail, s);
            }
        }
        g.nt *= 2;   // 64-byte K-tiles
        g.nt1 *= 2;
        g.x3t *= 2;
        g.x3c *= 2;
        return sgg_launch_pingpong(g, dt, CONV, s);
    }
    const bool narrow = g.N < 128;
    if (dt == SGG_BF16) return narrow ? launch<SGG_BF16, 4, 1, CONV>(g, s) : launch<SGG_BF16, 2, 2, CONV>(g, s);
    if (dt == SGG_F16) return narrow ? launch<SGG_F16, 4, 1, CONV>(g, s) : launch<SGG_F16, 2, 2, CONV>(g, s);
    return narrow ? launch<SGG_F32, 4, 1, CONV>(g, s) : launch<SGG_F32, 2, 2, CONV>(g, s);
}

}  // namespace

static int gemm_impl(const void* A, int lda, const void* A2, int lda2, int K1, const void* W, int ldw, const void* W2,
                     int ldw2, const float* bias, const float* post_scale, const float* post_shift, void* C, int ldc, int M,
                     int N, int K, int act, int in_dtype, int out_dtype, const float* add_rows, int ld_add, const int* add_idx, void* stream,
                     const float* gadd = nullptr, int ld_gadd = 0, int ggroup = 1, int gcol0 = 0) {
    // SGG_PAIR16 operands: A [M, >= 2 K] and W [N, >= 2 K] hold f16 planes [hi (K) | lo (K)], K per plane (a multiple of 64); the kernels
    // walk hi.hi + hi.lo + lo.hi.  SGG_PAIR16 output: C row = [hi (ldc / 2) | lo (ldc / 2)].
    const bool pair_in = in_dtype == SGG_PAIR16, pair_out = out_dtype == SGG_PAIR16;
    if (pair_in) in_dtype = SGG_F16;
    if (!sgg_is_dtype(in_dtype) || !(pair_out || sgg_is_dtype(out_dtype))) return SGG_ERR_DTYPE;
    if (M == 0 || N == 0) return SGG_OK;
    const int esz = sgg_elem_size(in_dtype);
    const int bke = ROWB / esz;
    if (!A || !W || !C || M < 0 || N < 0 || K <= 0 || K % bke) return SGG_ERR_ARG;
    if (pair_in && (A2 || W2 || lda < 2 * K || ldw < 2 * K)) return SGG_ERR_ARG;
    if (pair_out && ((ldc & 15) || ldc / 2 < N || gadd)) return SGG_ERR_ARG;
    if (!A2) K1 = K;
    if (K1 <= 0 || K1 > K || K1 % bke) return SGG_ERR_ARG;
    if ((lda & 7) || (ldw & 7) || (A2 && (lda2 & 7)) || (W2 && (ldw2 & 7))) return SGG_ERR_ARG;
    if (W2 && (!A2 || (((uintptr_t)W2) & 15) || ldw2 < K - K1 || ldw < K1)) return SGG_ERR_ARG;
    if (((uintptr_t)A | (uintptr_t)W | (uintptr_t)(A2 ? A2 : A)) & 15) return SGG_ERR_ARG;
    if (lda < K1 || (!W2 && ldw < K) || (!pair_out && ldc < N) || (A2 && lda2 < K - K1)) return SGG_ERR_ARG;
    // the kernels address operand rows as (uniform base + 32-bit lane offset): every operand must span < 4 GiB
    const long lim = 0xffff0000L;
    if ((long)M * lda * esz > lim || (long)N * ldw * esz > lim || (A2 && (long)M * lda2 * esz > lim) || (W2 && (long)N * ldw2 * esz > lim))
        return SGG_ERR_SPAN;
    GemmArgs g{};
    g.A = (const char*)A; g.A2 = (const char*)A2; g.Wt = (const char*)W; g.W2 = (const char*)W2;
    g.lda_b = (long)lda * esz; g.lda2_b = (long)lda2 * esz; g.ldw_b = (long)ldw * esz; g.ldw2_b = (long)ldw2 * esz;
    g.nt1 = K1 / bke; g.nt = K / bke;
    if (pair_in) {
        g.x3t = K / bke; g.nt = 3 * g.x3t; g.nt1 = g.nt;
        g.x3c = x3_chunk_tiles(g.x3t, 8);           // chunks of <= 512 elements per plane
    }
    g.bias = bias; g.pscale = post_scale; g.pshift = post_shift;
    g.C = (char*)C; g.ldc = ldc; g.M = M; g.N = N; g.act = act; g.out_dt = out_dtype;
    g.pair_off = pair_out ? ldc / 2 : 0;
    g.add_rows = add_rows; g.add_idx = add_idx; g.ld_add = ld_add;
    g.gadd = gadd; g.ld_gadd = ld_gadd; g.ggroup = ggroup; g.gcol0 = gcol0;
    return dispatch<false>(g, in_dtype, (hipStream_t)stream);
}

extern "C" int sgg_gemm(const void* A, int lda, const void* A2, int lda2, int K1, const void* W, int ldw, const void* W2,
                        int ldw2, const float* bias, const float* post_scale, const float* post_shift, void* C, int ldc, int M,
                        int N, int K, int act, int in_dtype, int out_dtype, void* stream) {
    return gemm_impl(A, lda, A2, lda2, K1, W, ldw, W2, ldw2, bias, post_scale, post_shift, C, ldc, M, N, K, act, in_dtype, out_dtype, nullptr, 0,
                     nullptr, stream);
}

// C = act(A . W^T + add_rows[add_idx[m]] + bias): sgg_gemm with a gathered f32 row added to every output row before the bias and the
// activation (add_idx NULL: row m itself).  Used where a long contraction is shared by several output rows (fc6 on the unordered box
// pairs, once; this call adds it to the per-edge rect term of both directions).
extern "C" int sgg_gemm_addrows(const void* A, int lda, const void* W, int ldw, const float* bias, const float* add_rows, int ld_add,
                                const int* add_idx, void* C, int ldc, int M, int N, int K, int act, int in_dtype, int out_dtype,
                                void* stream) {
    if (!add_rows || ld_add < N) return SGG_ERR_ARG;
    return gemm_impl(A, lda, nullptr, 0, K, W, ldw, nullptr, 0, bias, nullptr, nullptr, C, ldc, M, N, K, act, in_dtype, out_dtype, add_rows,
                     ld_add, add_idx, stream);
}

// C[m][n] = A . W^T + gadd[m][(n + col0) / group]: sgg_gemm with one f32 addend per output row and group of `group` consecutive
// columns (fc6's weight gradient on the unordered pairs: the rect term's share, constant over the 49 positions of a channel).
extern "C" int sgg_gemm_groupadd(const void* A, int lda, const void* W, int ldw, const float* gadd, int ld_gadd, int group, int col0,
                                 void* C, int ldc, int M, int N, int K, int in_dtype, int out_dtype, void* stream) {
    if (!gadd || group <= 0 || col0 < 0 || (long)ld_gadd * group < (long)N + col0) return SGG_ERR_ARG;
    return gemm_impl(A, lda, nullptr, 0, K, W, ldw, nullptr, 0, nullptr, nullptr, nullptr, C, ldc, M, N, K, SGG_ACT_NONE, in_dtype, out_dtype,
                     nullptr, 0, nullptr, stream, gadd, ld_gadd, group, col0);
}

namespace {
// out[m][n] = post_scale * act(sum_s ws[s][m][n] + bias) + post_shift
template <typename TO>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int S, long MN, int N, const float* __restrict__ bias,
                                                            int act, const float* __restrict__ ps, const float* __restrict__ pt,
                                                            TO* __restrict__ out, long ldc) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= MN) return;
    float acc[8], t[8];
    load8(ws + i, acc);
    for (int s = 1; s < S; ++s) {
        load8(ws + (long)s * MN + i, t);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += t[k];
    }
    const int n = (int)(i % N);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float v = acc[k] + (bias ? bias[n + k] : 0.f);
        if (act == SGG_ACT_RELU) v = fmaxf(v, 0.f);
        if (ps) v *= ps[n + k];
        if (pt) v += pt[n + k];
        acc[k] = v;
    }
    store8(out + (i / N) * ldc + n, acc);          // N % 8 == 0: the 8 values lie in one row
}
}  // namespace

// Split-K form for short-M GEMMs (e.g. fc6 on the 256 object rows: 64 tiles x K = 25088): `splits` blocks per output
// tile, each reducing a K range into workspace[s] (fp32 [M,N]), then one reduce + epilogue pass.  N % 8 == 0, ldc % 8 == 0, C 16-byte aligned.
extern "C" int sgg_gemm_splitk(const void* A, int lda, const void* W, int ldw, const float* bias, const float* post_scale,
                               const float* post_shift, void* C, int ldc, int M, int N, int K, int act, int in_dtype, int out_dtype,
                               int splits, float* workspace, void* stream) {
    const bool pair_in = in_dtype == SGG_PAIR16;       // pair operands [hi (K) | lo (K)], K per plane: the K range that is split is 3 K long
    if (pair_in) in_dtype = SGG_F16;
    if (!sgg_is_dtype(in_dtype) || !sgg_is_dtype(out_dtype)) return SGG_ERR_DTYPE;
    if (M == 0 || N == 0) return SGG_OK;
    const int esz = sgg_elem_size(in_dtype);
    const int bke = ROWB / esz;
    if (!A || !W || !C || !workspace || M < 0 || N <= 0 || (N & 7) || K <= 0 || K % bke || splits < 1 || splits > K / bke) return SGG_ERR_ARG;
    if ((lda & 7) || (ldw & 7) || lda < (pair_in ? 2 : 1) * K || ldw < (pair_in ? 2 : 1) * K || (((uintptr_t)A | (uintptr_t)W) & 15)) return SGG_ERR_ARG;
    if (ldc < N || (ldc & 7) || ((uintptr_t)C & 15)) return SGG_ERR_ARG;
    if ((long)M * lda * esz > 0xffff0000L || (long)N * ldw * esz > 0xffff0000L) return SGG_ERR_SPAN;   // 32-bit lane offsets
    GemmArgs g{};
    g.A = (const char*)A; g.Wt = (const char*)W;
    g.lda_b = (long)lda * esz; g.ldw_b = (long)ldw * esz;
    g.nt = K / bke; g.nt1 = g.nt;
    if (pair_in) {
        g.x3t = g.nt; g.nt = 3 * g.x3t; g.nt1 = g.nt;
        // split-K slices cut the 3 x3t tiles at arbitrary tile boundaries: any order of the (segment, K-tile) pairs sums the same products
        g.x3c = x3_chunk_tiles(g.x3t, 8);
    }
    g.C = (char*)workspace; g.ldc = N; g.M = M; g.N = N; g.act = SGG_ACT_NONE; g.out_dt = SGG_F32;
    g.splitk_stride = (long)M * N * 4;
    hipStream_t s = (hipStream_t)stream;
    // whole 256x256 tiles whose slices fill at least half of the chip (the last tile columns of fc6's weight gradient: 32 tiles x 8 slices):
    // the ping-pong kernel (one slice per CU); otherwise the 128x128 kernel
    const long tiles256 = (long)(M / 256) * (N / 256);
    static const char* no256 = getenv("SGG_SPLITK_128");
    int rc;
    if (!(no256 && no256[0] == '1') && M % 256 == 0 && N % 256 == 0 && tiles256 * splits >= N_CU_CHIP / 2 && tiles256 * splits <= 2 * N_CU_CHIP &&
        g.nt * 2 / splits >= 8) {
        g.nt *= 2;          // 64-byte K-tiles
        g.nt1 = g.nt;
        g.x3t *= 2;
        g.x3c *= 2;
        rc = sgg_launch_pingpong_splitk(g, in_dtype, splits, s);
    } else {
        rc = in_dtype == SGG_BF16 ? launch<SGG_BF16, 2, 2, false>(g, s, splits)
             : in_dtype == SGG_F16 ? launch<SGG_F16, 2, 2, false>(g, s, splits) : launch<SGG_F32, 2, 2, false>(g, s, splits);
    }
    if (rc != SGG_OK) return rc;
    const long MN = (long)M * N;
    const dim3 grid((unsigned)((MN / 8 + 255) / 256)), blk(256);
    SGG_FOR_DTYPE(out_dtype, hipLaunchKernelGGL(splitk_reduce_kernel<T>, grid, blk, 0, s, workspace, splits, MN, N, bias, act, post_scale, post_shift, (T*)C, (long)ldc));
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

// C[N, K] = A[Mred, N]^T . B[Mred, K]  (16-bit in, f32 or 16-bit out): the weight-gradient contraction without transposed
// copies of its operands.  Mred % 64 == 0, N % 128 == 0, K % 128 == 0, lda/ldb/ldc multiples of 8.  splits > 1: the
// reduction rows are split over `splits` workgroups per tile, partial sums in workspace f32[splits, N, K], then one reduce.
extern "C" int sgg_gemm_tn(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int Mred, int N, int K,
                           int in_dtype, int out_dtype, int splits, float* workspace, void* stream) {
    if ((in_dtype != SGG_BF16 && in_dtype != SGG_F16) || !sgg_is_dtype(out_dtype)) return SGG_ERR_DTYPE;
    if (!A || !B || !C || Mred <= 0 || N <= 0 || K <= 0 || (Mred & 63) || (N & 127) || (K & 127) || lda < N || ldb < K ||
        ldc < K || ((lda | ldb | ldc) & 7) || (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) || splits < 1 ||
        splits > Mred / 64 || (splits > 1 && !workspace))
        return SGG_ERR_ARG;
    if ((long)Mred * lda * 2 > 0xffff0000L || (long)Mred * ldb * 2 > 0xffff0000L) return SGG_ERR_SPAN;   // 32-bit lane offsets
    GemmArgs g{};
    g.A = (const char*)A; g.Wt = (const char*)B;
    g.lda_b = (long)lda * 2; g.ldw_b = (long)ldb * 2;
    g.nt = Mred / 64; g.nt1 = g.nt;
    g.M = N; g.N = K; g.act = SGG_ACT_NONE;
    hipStream_t s = (hipStream_t)stream;
    // whole 256x256 tiles and enough of them: the ping-pong kernel's TN form (gemm256.hip).  SGG_TN_PP=0 keeps the 128x128 kernel, 2 takes
    // the ping-pong form whenever the shapes allow it.
    static const int tn_pp = getenv("SGG_TN_PP") ? atoi(getenv("SGG_TN_PP")) : 1;
    if (tn_pp && splits == 1 && !(N & 255) && !(K & 255) && ((long)(N / 256) * (K / 256) >= 128 || tn_pp == 2)) {
        g.nt = Mred / 32; g.nt1 = g.nt;
        g.C = (char*)C; g.ldc = ldc; g.out_dt = out_dtype;
        return sgg_launch_pingpong_tn(g, in_dtype, 1, s);
    }
    // split form on the ping-pong kernel for few 256x256 tiles with a long reduction (unary / GRU weight gradients): measured (tools/tn_bench.py,
    // us, this route / the 128x128 kernel at its best split) GRU [31744 x 1536]^T [31744 x 512] 73 / 89, unary [7936 x 512]^T [7936 x 4096] 55 / 50 --
    // the fp32 partials (tiles x splits x 256 KB) cost what the faster tile loop gains; off unless SGG_TN_PP_SPLIT=1
    static const int tn_pp_split = getenv("SGG_TN_PP_SPLIT") ? atoi(getenv("SGG_TN_PP_SPLIT")) : 0;
    const long t256 = (long)(N / 256) * (K / 256);
    if (tn_pp && tn_pp_split && splits > 1 && !(N & 255) && !(K & 255) && t256 * splits >= 128 && t256 * splits <= 512 && Mred / 32 / splits >= 8) {
        g.nt = Mred / 32; g.nt1 = g.nt;
        g.C = (char*)workspace; g.ldc = K; g.out_dt = SGG_F32; g.splitk_stride = (long)N * K * 4;
        const int rc = sgg_launch_pingpong_tn(g, in_dtype, splits, s);
        if (rc != SGG_OK) return rc;
        const long MN = (long)N * K;
        const dim3 grid((unsigned)((MN / 8 + 255) / 256)), blk(256);
        SGG_FOR_DTYPE(out_dtype, hipLaunchKernelGGL(splitk_reduce_kernel<T>, grid, blk, 0, s, workspace, splits, MN, K, (const float*)nullptr, SGG_ACT_NONE,
                                                    (const float*)nullptr, (const float*)nullptr, (T*)C, (long)ldc));
        SGG_CHECK_LAUNCH();
        return SGG_OK;
    }
    constexpr int smem = 2 * 4 * 64 * ROWB;
    auto kern = in_dtype == SGG_BF16 ? mfma_tile_tn_kernel<SGG_BF16> : mfma_tile_tn_kernel<SGG_F16>;
    static bool attr_done[2] = {false, false};
    if (!attr_done[in_dtype == SGG_F16]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return SGG_ERR_LAUNCH;
        attr_done[in_dtype == SGG_F16] = true;
    }
    if (splits == 1) {
        g.C = (char*)C; g.ldc = ldc; g.out_dt = out_dtype;
    } else {
        g.C = (char*)workspace; g.ldc = K; g.out_dt = SGG_F32; g.splitk_stride = (long)N * K * 4;
    }
    hipLaunchKernelGGL(kern, dim3((N / 128) * (K / 128), splits), dim3(256), smem, s, g);
    SGG_CHECK_LAUNCH();
    if (splits > 1) {
        const long MN = (long)N * K;
        const dim3 grid((unsigned)((MN / 8 + 255) / 256)), blk(256);
        SGG_FOR_DTYPE(out_dtype, hipLaunchKernelGGL(splitk_reduce_kernel<T>, grid, blk, 0, s, workspace, splits, MN, K, (const float*)nullptr, SGG_ACT_NONE,
                                                    (const float*)nullptr, (const float*)nullptr, (T*)C, (long)ldc));
        SGG_CHECK_LAUNCH();
    }
    return SGG_OK;
}

namespace {
// the last (Mred mod 32) reduction rows of both operands, copied into one zero-padded K-tile: ws = [32][N] then [32][K] (16-bit elements)
__global__ __launch_bounds__(256) void tn_tail_pad_kernel(const char* __restrict__ A, long lda_b, const char* __restrict__ B, long ldb_b,
                                                          int row0, int Mred, int N, int K, char* __restrict__ ws) {
    const int per_row = (N + K) / 8;                              // 16-byte pieces per padded row
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 32 * per_row) return;
    const int r = i / per_row, c = (i - r * per_row) * 8;
    u32x4 v = {0, 0, 0, 0};
    if (row0 + r < Mred) v = c < N ? *reinterpret_cast<const u32x4*>(A + (long)(row0 + r) * lda_b + (long)c * 2)
                                   : *reinterpret_cast<const u32x4*>(B + (long)(row0 + r) * ldb_b + (long)(c - N) * 2);
    char* dst = c < N ? ws + ((long)r * N + c) * 2 : ws + (long)32 * N * 2 + ((long)r * K + (c - N)) * 2;
    *reinterpret_cast<u32x4*>(dst) = v;
}
}  // namespace

// The ping-pong kernel's TN form behind its own entry: C[N,K] = A[Mred,N]^T . B[Mred,K] (+ gadd[n][(k + col0) / group]) for whole 256x256
// output tiles and ANY number of reduction rows (the last Mred mod 32 rows go through pad_ws, 64 (N + K) bytes, as a zero-padded K-tile).
extern "C" int sgg_gemm_tn256(const void* A, int lda, const void* B, int ldb, const float* gadd, int ld_gadd, int group, int col0, void* C,
                              int ldc, int Mred, int N, int K, int in_dtype, int out_dtype, void* pad_ws, void* stream) {
    // SGG_PAIR16 operands (x3 mode): A [Mred, >= 2 N] = [hi (N) | lo (N)] per row, B [Mred, >= 2 K] likewise; the kernel walks hi.hi + hi.lo + lo.hi
    // over the same reduction rows.  Whole K-tiles only (Mred % 32 == 0).
    const bool pair_in = in_dtype == SGG_PAIR16;
    if (pair_in) in_dtype = SGG_F16;
    if ((in_dtype != SGG_BF16 && in_dtype != SGG_F16) || !sgg_is_dtype(out_dtype)) return SGG_ERR_DTYPE;
    if (!A || !B || !C || Mred <= 0 || N <= 0 || K <= 0 || (N & 255) || (K & 255) || lda < N || ldb < K || ldc < K || ((lda | ldb | ldc) & 7) ||
        (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)pad_ws) & 15) || ((Mred & 31) && !pad_ws))
        return SGG_ERR_ARG;
    if (pair_in && ((Mred & 31) || lda < 2 * N || ldb < 2 * K)) return SGG_ERR_ARG;
    if (gadd && (group <= 0 || col0 < 0 || (long)ld_gadd * group < (long)K + col0)) return SGG_ERR_ARG;
    if (32L * lda * 2 + 2L * N > 0xffff0000L || 32L * ldb * 2 + 2L * K > 0xffff0000L) return SGG_ERR_SPAN;     // 32-bit lane offsets inside a K-tile
    hipStream_t s = (hipStream_t)stream;
    GemmArgs g{};
    g.A = (const char*)A; g.Wt = (const char*)B;
    g.lda_b = (long)lda * 2; g.ldw_b = (long)ldb * 2;
    g.nt1 = Mred / 32; g.nt = (Mred + 31) / 32;
    if (pair_in) {
        g.x3t = g.nt; g.nt *= 3; g.nt1 = g.nt;
        g.x3c = x3_chunk_tiles(g.x3t, 16);
    }
    if (Mred & 31) {
        g.A2 = (const char*)pad_ws; g.lda2_b = (long)N * 2;
        g.W2 = (const char*)pad_ws + 32L * N * 2; g.ldw2_b = (long)K * 2;
        const int pieces = 32 * ((N + K) / 8);
        hipLaunchKernelGGL(tn_tail_pad_kernel, dim3((pieces + 255) / 256), dim3(256), 0, s, g.A, g.lda_b, g.Wt, g.ldw_b, g.nt1 * 32, Mred, N, K, (char*)pad_ws);
        SGG_CHECK_LAUNCH();
    }
    g.M = N; g.N = K; g.act = SGG_ACT_NONE;
    g.C = (char*)C; g.ldc = ldc; g.out_dt = out_dtype;
    g.gadd = gadd; g.ld_gadd = ld_gadd; g.ggroup = group; g.gcol0 = col0;
    return sgg_launch_pingpong_tn(g, in_dtype, 1, s);
}

// The first block of VGG-16 in one launch (16-bit modes): conv1_1 (3 -> 64) + ReLU + conv1_2 (64 -> 64) + ReLU [+ MaxPool2d(2)].
// in_nhwc4: the normalised image plane [B, H+2, W+2, 4] f32 with zero border (sgg_image_prep*); w1_frags: conv1_1's weights as packed by
// sgg_conv1_pack_weights (4 KiB, `dtype`); w2 [64][3][3][64] in `dtype`; out: [B, H+2p, W+2p, 64], or the pooled plane
// [B, H/2+2p, W/2+2p, 64] with pool = 1 (H, W even).
extern "C" int sgg_conv1_pack_weights(const float* w1 /*[64][27], k = (ky*3+kx)*3 + c*/, void* frags /*4096 bytes*/, int dtype, void* stream) {
    if (dtype != SGG_BF16 && dtype != SGG_F16) return SGG_ERR_DTYPE;
    if (!w1 || !frags || ((uintptr_t)frags & 15)) return SGG_ERR_ARG;
    return sgg_launch_conv1_pack(w1, frags, dtype, (hipStream_t)stream);
}

extern "C" int sgg_conv1_block(const float* in_nhwc4, const void* w1, const float* b1, const void* w2, const float* b2, void* out, int out_pad,
                               int B, int H, int W, int pool, int dtype, void* stream) {
    if (dtype != SGG_BF16 && dtype != SGG_F16) return SGG_ERR_DTYPE;
    if (!in_nhwc4 || !w1 || ((uintptr_t)w1 & 15) || ((uintptr_t)b1 & 15) || !b1 || !w2 || !b2 || !out || B <= 0 || H <= 0 || W <= 0 || (out_pad != 0 && out_pad != 1)) return SGG_ERR_ARG;
    if (pool && ((H | W) & 1)) return SGG_ERR_ARG;
    if ((long)B * (H + 2) * (W + 2) * 16 > 0xffff0000L) return SGG_ERR_SPAN;
    const int rc = sgg_launch_conv1_block(in_nhwc4, w1, b1, w2, b2, out, out_pad, B, H, W, dtype, pool, (hipStream_t)stream);
    return rc <= 0 ? rc : SGG_ERR_ARG;
}

// The x3 mode's 3x3 convolution + ReLU [+ MaxPool2d(2)] on the patch kernel (conv_pp.hip): `in` a PAIR plane [B, H+2, W+2, 2 Cin] (SGG_PAIR16), `w3`
// f16 [Cout][9][3 Cin] with tap = [hi | lo | hi] (sgg_split3, weight form), `out` a PAIR plane [B, H+2p, W+2p, 2 Cout] (pool: of the pooled size).
// Maps of >= 64 cells a side, Cout % 128 == 0, Cin % 32 == 0; SGG_ERR_ARG otherwise (the caller takes sgg_conv3x3_relu's pair form).
extern "C" int sgg_conv3x3_relu_x3(const void* in, const void* w3, const float* bias, void* out, int out_pad, int B, int H, int W, int Cin, int Cout,
                                   int pool, void* stream) {
    if (!in || !w3 || !bias || !out || B <= 0 || H < 64 || W < 64 || Cin % 32 || Cout % 64 || (out_pad != 0 && out_pad != 1)) return SGG_ERR_ARG;
    if (pool && ((H | W) & 1)) return SGG_ERR_ARG;
    if ((long)B * (H + 2) * (W + 2) * Cin * 4 > 0xffff0000L || 27L * Cin * Cout * 2 > 0xffff0000L) return SGG_ERR_SPAN;
    if ((((uintptr_t)in | (uintptr_t)w3 | (uintptr_t)out) & 15)) return SGG_ERR_ARG;
    int rc = Cout % 128 == 0 ? sgg_launch_conv_pp_x3(in, w3, bias, out, out_pad, B, H, W, Cin, Cout, pool, (hipStream_t)stream) : 1;
    if (rc > 0) rc = sgg_launch_conv_spatial_x3(in, w3, bias, out, out_pad, B, H, W, Cin, Cout, pool, (hipStream_t)stream);     // (64-channel layers)
    return rc <= 0 ? rc : SGG_ERR_ARG;
}

extern "C" int sgg_conv3x3_relu(const void* in, const void* w, const float* bias, void* out, int out_pad, int B, int H,
                                int W, int Cin, int Cout, int pool, int dtype, int out_dtype, void* stream) {
    // SGG_PAIR16 input: `in` is a pair plane [B, H+2, W+2, 2 Cin] (pixel = [hi (Cin) | lo (Cin)]), `w` [Cout][9][2 Cin] likewise; the
    // implicit-GEMM kernels walk hi.hi + hi.lo + lo.hi per tap (x3 mode).  SGG_PAIR16 output: [B, H+2p, W+2p, 2 Cout].
    const bool pair_in = dtype == SGG_PAIR16, pair_out = out_dtype == SGG_PAIR16;
    if (pair_in) dtype = SGG_F16;
    if (!sgg_is_dtype(dtype) || !(pair_out || sgg_is_dtype(out_dtype))) return SGG_ERR_DTYPE;
    if ((pair_in || out_dtype != dtype) && pool) return SGG_ERR_ARG;       // another output type: the implicit-GEMM kernels only (no fused pool)
    const int esz = sgg_elem_size(dtype);
    const int bke = ROWB / esz;
    if (!in || !w || !out || B <= 0 || H <= 0 || W <= 0 || Cin % bke || Cout % 64 || (out_pad != 0 && out_pad != 1))
        return SGG_ERR_ARG;
    if ((long)B * H * W > 0x7fffffffL) return SGG_ERR_ARG;
    if ((long)B * (H + 2) * (W + 2) * Cin * esz * (pair_in ? 2 : 1) > 0xffff0000L || 9L * Cin * Cout * esz * (pair_in ? 2 : 1) > 0xffff0000L) return SGG_ERR_SPAN;   // 32-bit lane offsets
    if (!pair_in) {
        // wide-spatial layers: LDS-resident input patch kernel (conv_spatial.hip); small maps: implicit GEMM.
        // SGG_CONV_FORCE=gemm|spatial|old overrides (experiments only).
        static const char* force = getenv("SGG_CONV_FORCE");
        // measured (round 2, after the scalar-base DMA went into every kernel; TFLOP/s spatial vs implicit GEMM at B=8): conv1_2 620 / 477,
        // conv2_1 594 / 558, conv2_2 781 / 750, conv3_1 749 / 826, conv3_2 881 / 981, conv4_1 781 / 795, conv4_2 849 / 890, conv5 713 / 688
        // (conv5: 37x37 there, 38x38 in the detector): the narrow layers on the patch kernel, the >= 256-channel ones on the GEMM
        const bool same_out = out_dtype == dtype;
        const bool want = same_out && (force ? (force[0] == 's' || (force[0] == 'o' && H >= 64 && W >= 64)) : (H >= 64 && W >= 64 && Cout < 256));   // 'o': round 1's rule
        if (pool && ((H | W) & 1)) return SGG_ERR_ARG;
        {
            const char* pp = getenv("SGG_CONV_PP");                 // experiments: 0 = off, 42 / 22 / 24 = force the workgroup form (conv_pp.hip)
            const char* ptw = getenv("SGG_CONV_PP_TW");             //              16 / 32 = force the tile width
            const int nq = pp ? atoi(pp) : 0, tw = ptw ? atoi(ptw) : 0;
            if (same_out && !(pp && pp[0] == '0') && !force && H >= 64 && W >= 64 && Cout >= 128) {
                const int rc = sgg_launch_conv_pp(in, w, bias, out, out_pad, B, H, W, Cin, Cout, dtype, pool, nq == 42 || nq == 22 || nq == 24 ? nq : 0, tw == 16 || tw == 32 ? tw : 0, (hipStream_t)stream);
                if (rc <= 0) return rc;
            }
        }
        if (want || pool) {
            const int rc = sgg_launch_conv_spatial(in, w, bias, out, out_pad, B, H, W, Cin, Cout, dtype, pool, (hipStream_t)stream);
            if (rc <= 0) return rc;
        }
        if (pool) return SGG_ERR_ARG;    // the fused pool lives in the spatial kernel only
    }
    GemmArgs g{};
    g.A = (const char*)in; g.Wt = (const char*)w;
    g.ldw_b = (long)9 * Cin * esz * (pair_in ? 2 : 1);
    g.nt = 9 * Cin / bke; g.nt1 = g.nt;
    if (pair_in) {
        g.x3t = Cin / bke; g.nt *= 3; g.nt1 = g.nt; g.cin_px = 2 * Cin;
        g.x3c = g.x3t;                               // (conv mode walks the three segments per tap: its own chunking)
    }
    g.bias = bias; g.C = (char*)out; g.M = B * H * W; g.N = Cout; g.act = SGG_ACT_RELU;
    g.out_dt = out_dtype;
    if (pair_out) {
        g.cout_px = 2 * Cout; g.pair_off = Cout;
    }
    g.H = H; g.W = W; g.Cin = Cin; g.out_pad = out_pad;
    return dispatch<true>(g, dtype, (hipStream_t)stream);
}
