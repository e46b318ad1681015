// 256x256 ping-pong MFMA tile kernel (gfx950): the large-N GEMMs (fc6/fc7/GRU/unary) and the wide VGG convs.
//
// 8 waves = two groups of four (one wave of each group per SIMD).  Each wave owns a 128(M) x 64(N) slab of the
// 256x256 block tile: 4x2 MFMA 32x32 tiles = 128 accumulator registers.  K advances in 64-byte slabs per row
// (32 bf16 / 16 f32): one K-tile = 2 MFMA k-steps = 16 MFMAs per wave.  The two groups run the same K-tile
// sequence one PHASE apart, one s_barrier per phase:
//
//   slot      0        1        2        3        4   ...
//   group 0   LOAD 0   MFMA 0   LOAD 1   MFMA 1   LOAD 2
//   group 1   -        LOAD 0   MFMA 0   LOAD 1   MFMA 1
//
// LOAD = 12 ds_read_b128 (fragments, single-buffered in registers); MFMA = 16 MFMAs with this wave's 4 global->LDS
// DMA pieces of the K-tile three ahead issued between them (an LDS-DMA issue costs ~60 cycles in an MFMA phase,
// 100-185 in a LOAD phase: measured 1053 vs 963 TFLOP/s).  A 4-deep LDS ring (4 x 32 KiB) keeps 2-3 K-tiles of
// DMA in flight across the barriers (counted vmcnt, never 0 in the main loop).
// Ablation on fc6 (7936x4096x25600 bf16, random normal data): full 1055 TF; without the DMA 1398; without
// ds_reads 1099; MFMA only 1519 (the matrix pipe is then 82 % busy at 1.88 GHz: the chip clocks down under load).
//
// LDS image per stage: A rows [256][64 B] then W rows [256][64 B], lane-linear for global_load_lds
// (16 rows per 1-KiB wave instruction); 16-byte slot swizzle phys = slot ^ ((row>>2)&3) applied on the source
// address and on the ds_read_b128 (conflict-free for the 32-row fragment reads).
//
// TN form (template flag, 16-bit only; round 4): C[m][n] = sum_k A[k][m] * W[k][n] -- both operands lie with the REDUCTION index as
// their slow axis (weight gradients dW = dY^T X: no transposed copies of dY and X).  Same schedule, ring and epilogue; a stage holds,
// per operand, four [32 k-rows][128 B] column blocks (64 columns each, the 128x128 TN kernel's swizzled image in gemm.hip) filled by
// the same lane-linear DMA (an instruction = 8 k-rows x 128 B), and a fragment (8 reduction elements of one column per lane) is two
// ds_read_b64_tr_b16 of stage rows {a, a+1, a+8, a+9} -- both operands use the same row order, which is all a sum needs.
#include "gemm_args.h"

namespace {

constexpr int ROW = 64;                 // bytes of K per LDS row
constexpr int STAGE = 512 * ROW;        // A 256 rows + W 256 rows = 32 KiB
constexpr int NSTAGE = 4;
constexpr int SMEM = NSTAGE * STAGE;    // 128 KiB

// byte offset of K-tile kt inside its source rows and which of the two K segments it belongs to (uniform values)
template <bool CONV, bool TN = false>
__device__ __forceinline__ long tile_koff(const GemmArgs& g, bool loads_a, int kt, int tpc, int esz, bool& seg2) {
    seg2 = false;
    if constexpr (TN) {     // 32 reduction rows further down; K-tiles from nt1 on come out of the second pair of operands (the padded tail rows)
        if (g.x3t) {
            // x3 mode on PAIR operands, TN form: both operands are [Mred][hi (cols) | lo (cols)]; the three segments walk the SAME reduction rows
            // with another column plane: 0 = (A hi, B hi), 1 = (A hi, B lo), 2 = (A lo, B hi); plane stride = the operand's column count
            int seg, kk;
            x3_tile(g, kt, seg, kk);
            const long plane = (loads_a ? seg == 2 : seg == 1) ? (long)(loads_a ? g.M : g.N) * 2 : 0;
            return (long)kk * 32 * (loads_a ? g.lda_b : g.ldw_b) + plane;
        }
        seg2 = g.A2 && kt >= g.nt1;
        return seg2 ? (long)(kt - g.nt1) * 32 * (loads_a ? g.lda2_b : g.ldw2_b) : (long)kt * 32 * (loads_a ? g.lda_b : g.ldw_b);
    }
#ifdef SGG_GEMM_ABL_HOT   // experiment: every K-tile re-reads tile 0 (operands stay in L2): what the memory side costs
    kt = 0;
#endif
    if (g.x3t) {    // pair operands [hi | lo]: segment 0 = (A hi, W hi), 1 = (A hi, W lo), 2 = (A lo, W hi); in conv mode per tap
        if constexpr (CONV) {
            const int tap = kt / (3 * tpc), r = kt - tap * 3 * tpc;
            const int seg = r / tpc, c0 = r - seg * tpc;
            if (loads_a) {
                const int ky = tap / 3, kx = tap - ky * 3;
                return ((long)(ky * (g.W + 2) + kx) * g.cin_px + (seg == 2 ? g.Cin : 0)) * esz + c0 * ROW;
            }
            return ((long)tap * g.cin_px + (seg == 1 ? g.Cin : 0)) * esz + c0 * ROW;
        } else {
            int seg, kk;
            x3_tile(g, kt, seg, kk);
            return (long)((seg == (loads_a ? 2 : 1) ? g.x3t : 0) + kk) * ROW;
        }
    }
    if (loads_a) {
        if constexpr (CONV) {
            const int tap = kt / tpc, c0 = kt - tap * tpc;
            const int ky = tap / 3, kx = tap - ky * 3;
            return ((long)(ky * (g.W + 2) + kx) * g.Cin) * esz + c0 * ROW;
        } else {
            seg2 = kt >= g.nt1;
            return (long)(seg2 ? kt - g.nt1 : kt) * ROW;
        }
    }
    seg2 = !CONV && g.W2 && kt >= g.nt1;
    return (long)(seg2 ? kt - g.nt1 : kt) * ROW;
}

template <int DT, bool CONV, bool TN = false>
__global__ __launch_bounds__(512) void mfma_pingpong_kernel(const GemmArgs g) {
    static_assert(!TN || (!CONV && DT != SGG_F32), "TN form: plain GEMM on 16-bit operands (ds_read_b64_tr_b16 moves 16-bit elements)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ESZ = DT == SGG_F32 ? 4 : 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, q = wave & 3;  // grp: which half of M; q: which 64-wide N slab

    const int tilesM = (g.M - g.m_base + 255) / 256, tilesN = (g.N + 255) / 256;
    int tm, tn;
    tile_coords(blockIdx.x, tilesM, tilesN, tm, tn);
    const int m0 = g.m_base + tm * 256, n0 = tn * 256;

    // ---- DMA duty: waves 0-3 stage the A rows, waves 4-7 the W rows; 4 instructions of 16 rows each
    const bool loads_a = wave < 4;
    const char* rp[4];
    const char* rp2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if constexpr (TN) {   // piece j = k-rows 8j .. 8j+7 of column block (wave & 3): lane -> (row 8j + lane/8, swizzled 16-byte chunk)
            const int r = j * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((r >> 1) & 7);
            rp[j] = (loads_a ? g.A + (long)r * g.lda_b + (long)m0 * 2 : g.Wt + (long)r * g.ldw_b + (long)n0 * 2) + (wave & 3) * 128 + chunk * 16;
            rp2[j] = !g.A2 ? nullptr
                           : (loads_a ? g.A2 + (long)r * g.lda2_b + (long)m0 * 2 : g.W2 + (long)r * g.ldw2_b + (long)n0 * 2) + (wave & 3) * 128 + chunk * 16;
            continue;
        }
        const int r = (wave & 3) * 64 + j * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ ((r >> 2) & 3);
        if (loads_a) {
            const int m = min(m0 + r, g.M - 1);
            rp[j] = a_row_ptr<CONV>(g, m, ESZ) + chunk * 16;
            rp2[j] = (!CONV && g.A2) ? g.A2 + (long)m * g.lda2_b + chunk * 16 : nullptr;
        } else {
            const int n = min(n0 + r, g.N - 1);
            rp[j] = g.Wt + (long)n * g.ldw_b + chunk * 16;
            rp2[j] = (!CONV && g.W2) ? g.W2 + (long)n * g.ldw2_b + chunk * 16 : nullptr;
        }
    }
    const int tpc = CONV ? (g.Cin * ESZ) / ROW : 1;
    const int lds_rows0 = (loads_a ? 0 : 256 * ROW) + (wave & 3) * 64 * ROW;
    // DMA addressing that costs the issuing wave nothing but the instruction itself: global = SGPR base (uniform, advanced
    // with scalar adds per K-tile) + a per-lane 32-bit offset that never changes + the instruction's immediate; the same
    // immediate (j KiB) places piece j in LDS behind ONE M0 value per K-tile.  The immediate applies to both addresses, so
    // the lane offset carries -j KiB (+3 KiB on the lane, -3 KiB on the base keep it unsigned).
    const char* sbase = loads_a ? (CONV ? g.A : g.A) : g.Wt;
    const char* sbase2 = loads_a ? g.A2 : g.W2;
    // (named scalars, not arrays: indexed through a runtime choice they would be demoted to scratch memory)
#define SGG_VO(j) (unsigned)(rp[j] - sbase) + 3072u - 1024u * j
#define SGG_VO2(j) (rp2[j] ? (unsigned)(rp2[j] - sbase2) + 3072u - 1024u * j : 0u)
    const unsigned vo_0 = SGG_VO(0), vo_1 = SGG_VO(1), vo_2 = SGG_VO(2), vo_3 = SGG_VO(3);
    const unsigned vo2_0 = SGG_VO2(0), vo2_1 = SGG_VO2(1), vo2_2 = SGG_VO2(2), vo2_3 = SGG_VO2(3);
#undef SGG_VO
#undef SGG_VO2
    // x3 GEMM on pair operands (plain, not conv): the K-tiles are requested strictly in order (prologue kb .. kb+2, then kt + 3 per iteration),
    // so (segment, K-tile inside the plane) is a running state instead of two integer divisions per K-tile inside the MFMA phase
    const bool x3_run = !CONV && !TN && g.x3t != 0;
    int x3_seg = 0, x3_r = 0, x3_base = 0;          // next tile: chunk base x3_base, segment x3_seg, offset x3_r inside the chunk
    auto x3_next_koff = [&]() -> long {
        const int kk = x3_base + x3_r;
        const long off = (long)((x3_seg == (loads_a ? 2 : 1) ? g.x3t : 0) + kk) * ROW;
        if (++x3_r == g.x3c) {
            x3_r = 0;
            if (++x3_seg == 3) {
                x3_seg = 0;
                x3_base += g.x3c;
            }
        }
        return off;
    };
    auto issue = [&](int kt) {
        bool seg2 = false;
        const long koff = x3_run ? x3_next_koff() : tile_koff<CONV, TN>(g, loads_a, kt, tpc, ESZ, seg2);
        const char* ub = uniform_ptr((seg2 ? sbase2 : sbase) + koff - 3072);      // the -3 KiB pairs with the +3 KiB inside vo_*
        char* dst = smem + (kt & (NSTAGE - 1)) * STAGE + lds_rows0;
        glds16_off<0>(ub + (seg2 ? vo2_0 : vo_0), dst);
        glds16_off<1024>(ub + (seg2 ? vo2_1 : vo_1), dst);
        glds16_off<2048>(ub + (seg2 ? vo2_2 : vo_2), dst);
        glds16_off<3072>(ub + (seg2 ? vo2_3 : vo_3), dst);
    };

    // ---- fragment addressing: row = base + (lane&31); logical slot = 2*s + (lane>>5)
    const int fr = lane & 31, fh = lane >> 5;
    int aoff[4], akey[4], boff[2], bkey[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = grp * 128 + i * 32 + fr;
        aoff[i] = r * ROW;
        akey[i] = (r >> 2) & 3;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = q * 64 + i * 32 + fr;
        boff[i] = 256 * ROW + r * ROW;
        bkey[i] = (r >> 2) & 3;
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    u32x4 af[4][2], bf[2][2];

    // TN: transposing reads.  kg = lane>>5: which 8 of the k-step's 16 reduction rows; 16-lane group g16 = (lane>>4)&1: columns 16 g16.. of the
    // 32-column block; p = lane&15 hands in stage row a + (0,1,8,9)[p>>2], 4 columns from 4 (p&3), and receives column p's 4 elements
    unsigned tn_a[2][2][4], tn_b[2][2][2];            // [k-step][lo / hi 4 of the 8][block]: workgroup-relative LDS byte offsets in a stage
    if constexpr (TN) {
        const int kg = lane >> 5, g16 = (lane >> 4) & 1, p = lane & 15;
        const int rsel = ((p >> 2) & 1) + ((p >> 3) << 3);
        const unsigned lds0 = (unsigned)(unsigned long)(lds_void_t*)smem;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int row = 16 * s + 2 * (2 * kg + h) + rsel, key = (row >> 1) & 7;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int col = (i & 1) * 32 + g16 * 16 + 4 * (p & 3);
                    tn_a[s][h][i] = lds0 + (grp * 2 + (i >> 1)) * 4096 + row * 128 + ((((col >> 3) ^ key) << 4) | (((col >> 2) & 1) << 3));
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int col = i * 32 + g16 * 16 + 4 * (p & 3);
                    tn_b[s][h][i] = lds0 + 256 * ROW + q * 4096 + row * 128 + ((((col >> 3) ^ key) << 4) | (((col >> 2) & 1) << 3));
                }
            }
    }
    auto load_frags = [&](int kt) {
        if constexpr (TN) {
            const unsigned st = (kt & (NSTAGE - 1)) * STAGE;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    unsigned long long lo, hi;
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(tn_b[s][0][i] + st) : "memory");
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(tn_b[s][1][i] + st) : "memory");
                    bf[i][s] = u32x4{(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    unsigned long long lo, hi;
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(tn_a[s][0][i] + st) : "memory");
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(tn_a[s][1][i] + st) : "memory");
                    af[i][s] = u32x4{(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
                }
            }
            return;
        }
        const char* st = smem + (kt & (NSTAGE - 1)) * STAGE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int slot = 2 * s + fh;
#pragma unroll
            for (int i = 0; i < 2; ++i) bf[i][s] = *reinterpret_cast<const u32x4*>(st + boff[i] + ((slot ^ bkey[i]) << 4));
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i][s] = *reinterpret_cast<const u32x4*>(st + aoff[i] + ((slot ^ akey[i]) << 4));
        }
    };

    // 16 MFMAs; the 4 DMA pieces of K-tile `pf` (DMA = false: none) are issued between them, where the wave is matrix-pipe
    // bound and has free issue slots.  Everything a piece needs beyond the instruction itself is resolved BEFORE the first
    // MFMA (uniform base in SGPRs, segment choice, M0): no branch, no vector ALU work and no M0 write sits between MFMAs.
#define SGG_MFMA_PAIR(s, mi)                                                                                              \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) {                                                                    \
        if constexpr (DT != SGG_F32) {                                                                                    \
            acc[mi][ni] = mfma_32x32x16<DT>(bf[ni][s], af[mi][s], acc[mi][ni]);                                           \
        } else {                                                                                                          \
            _Pragma("unroll") for (int c = 0; c < 4; ++c) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(             \
                __uint_as_float(bf[ni][s][c]), __uint_as_float(af[mi][s][c]), acc[mi][ni], 0, 0, 0);                      \
        }                                                                                                                 \
    }
    auto compute_dma = [&](int pf) {
        bool seg2 = false;
        const long koff = x3_run ? x3_next_koff() : tile_koff<CONV, TN>(g, loads_a, pf, tpc, ESZ, seg2);
        const char* ub = uniform_ptr((seg2 ? sbase2 : sbase) + koff - 3072);
        char* dst = smem + (pf & (NSTAGE - 1)) * STAGE + lds_rows0;
        const unsigned o0 = seg2 ? vo2_0 : vo_0, o1 = seg2 ? vo2_1 : vo_1, o2 = seg2 ? vo2_2 : vo_2, o3 = seg2 ? vo2_3 : vo_3;
        __builtin_amdgcn_s_setprio(1);
        SGG_MFMA_PAIR(0, 0)
        glds16_off<0>(ub + o0, dst);
        __builtin_amdgcn_sched_barrier(0);
        SGG_MFMA_PAIR(0, 1)
        glds16_off<1024>(ub + o1, dst);
        __builtin_amdgcn_sched_barrier(0);
        SGG_MFMA_PAIR(0, 2)
        glds16_off<2048>(ub + o2, dst);
        __builtin_amdgcn_sched_barrier(0);
        SGG_MFMA_PAIR(0, 3)
        glds16_off<3072>(ub + o3, dst);
        __builtin_amdgcn_sched_barrier(0);
        SGG_MFMA_PAIR(1, 0)
        SGG_MFMA_PAIR(1, 1)
        SGG_MFMA_PAIR(1, 2)
        SGG_MFMA_PAIR(1, 3)
        __builtin_amdgcn_s_setprio(0);
    };
    auto compute_plain = [&]() {
        __builtin_amdgcn_s_setprio(1);
        SGG_MFMA_PAIR(0, 0)
        SGG_MFMA_PAIR(0, 1)
        SGG_MFMA_PAIR(0, 2)
        SGG_MFMA_PAIR(0, 3)
        SGG_MFMA_PAIR(1, 0)
        SGG_MFMA_PAIR(1, 1)
        SGG_MFMA_PAIR(1, 2)
        SGG_MFMA_PAIR(1, 3)
        __builtin_amdgcn_s_setprio(0);
    };

    // wait until this wave's DMA pieces of tile kt+1 have landed; `later` = tiles it issued after kt+1
    auto wait_tiles_in_flight = [&](int later) {
        if (later >= 2) wait_vmcnt<8>();
        else if (later == 1) wait_vmcnt<4>();
        else wait_vmcnt<0>();
    };

    // K range of this workgroup: all of it, or -- split-K launches (gridDim.y > 1: a few output tiles with a long reduction, e.g. the
    // last two tile columns of fc6's weight gradient) -- slice blockIdx.y of the K-tiles, summed into the fp32 partial output
    // g.C + blockIdx.y * splitk_stride that a reduce pass adds up (sgg_gemm_splitk)
    int kb = 0, nt = g.nt;
    if (gridDim.y > 1) {
        const int per = (g.nt + (int)gridDim.y - 1) / (int)gridDim.y;
        kb = min((int)blockIdx.y * per, g.nt);
        nt = min(kb + per, g.nt);
    }
    if (x3_run && kb > 0) {          // a split-K slice starts inside the sequence
        const int c3 = 3 * g.x3c, chunk = kb / c3, r = kb - chunk * c3;
        x3_seg = (r >= g.x3c) + (r >= 2 * g.x3c);
        x3_r = r - x3_seg * g.x3c;
        x3_base = chunk * g.x3c;
    }
    // ---- prologue: tiles kb..kb+2 in flight, tile kb landed and visible
#pragma unroll
    for (int t = 0; t < 3; ++t)
        if (kb + t < nt) issue(kb + t);
    if (nt - kb >= 3) wait_vmcnt<8>();
    else if (nt - kb == 2) wait_vmcnt<4>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();

    // Two s_barriers per K-tile (measured better than one: 1053 vs 1008 TFLOP/s on fc6 -- with a single barrier the
    // two waves of a SIMD drift into the same phase).  Tile kt+3 goes into the stage tile kt-1 occupied; its last
    // readers (group 1, LOAD kt-1) retired their ds_reads before the barrier that ended slot 2kt-1.
    // The K loop is peeled: nt-3 steady iterations whose MFMA phase carries the DMA of tile kt+3, then the last three without
    // any -- one copy of the MFMA block per loop, no branch inside a phase.
    const int nsteady = max(nt - 3, kb);
    if (grp == 0) {
        for (int kt = kb; kt < nsteady; ++kt) {
            load_frags(kt);                                  // slot 2kt : LOAD
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            compute_dma(kt + 3);                             // slot 2kt+1 : MFMA + DMA issue of tile kt+3
            __builtin_amdgcn_sched_barrier(0);
            wait_vmcnt<8>();                                 // tile kt+1 landed; kt+2, kt+3 still in flight
            __builtin_amdgcn_s_barrier();
        }
        for (int kt = nsteady; kt < nt; ++kt) {
            load_frags(kt);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            compute_plain();
            __builtin_amdgcn_sched_barrier(0);
            wait_tiles_in_flight(min(2, nt - 2 - kt));       // tile kt+1 landed; nothing issued after kt+3 <= nt-1
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_s_barrier();  // slot 2nt: group 1 finishes its last MFMA phase
    } else {
        __builtin_amdgcn_s_barrier();  // slot 0: idle
        for (int kt = kb; kt < nsteady; ++kt) {
            load_frags(kt);                                  // slot 2kt+1 : LOAD
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            wait_vmcnt<4>();                                 // tile kt+1 landed; issued so far: .. kt+2
            __builtin_amdgcn_s_barrier();
            compute_dma(kt + 3);                             // slot 2kt+2 : MFMA + DMA issue of tile kt+3
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
        for (int kt = nsteady; kt < nt; ++kt) {
            load_frags(kt);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            wait_tiles_in_flight(min(1, nt - 2 - kt));
            __builtin_amdgcn_s_barrier();
            compute_plain();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
    }

    // ---- epilogue through LDS: per wave a [32][64] f32 staging tile (row stride 272 B), one 32-row block at a time
    constexpr int ESTRIDE = 272;
    char* est = smem + wave * (32 * ESTRIDE);
    GemmArgs gs = g;                                                     // (split-K: this slice's partial output)
    gs.C = g.C + (long)blockIdx.y * g.splitk_stride;
    const GemmArgs& g_ = gs;
    // Lean form for the common case (an interior tile of a plain GEMM: bias, ReLU, optionally the group addend of fc6's weight gradient).
    // s_memrealtime stamps (round 4) timed the general form below at 10-15 us of a 140-us tile -- not memory, VALU ISSUE: ~4000 instructions per
    // wave (64-bit address arithmetic and bounds tests per store, an integer division per group-addend column, loads between the
    // stores that each wait for the previous store's round trip) at 4 cycles each and two waves per SIMD.  Here: every global load
    // (bias, the lane's 16 x 2 group addends) BEFORE the first store; ONE 32-bit lane offset into a buffer descriptor of C, the row
    // block in the instruction's scalar offset; no bounds tests; ~600 instructions.
    {
        const int esz_out = g.out_dt == SGG_F32 ? 4 : 2;
        const bool fast = !CONV && g.out_dt != SGG_PAIR16 && !g.add_rows && !g.pscale && !g.pshift && (g.ldc & 7) == 0 && m0 + 256 <= g.M && n0 + 256 <= g.N &&
                          (!g.gadd || g.ggroup >= 8) && (long)g.M * g.ldc * esz_out < 0xffff0000L && (reinterpret_cast<uintptr_t>(g_.C) & 15) == 0;
        if (fast) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(g_.C, 0, (int)min((long)g.M * g.ldc * esz_out, 0xffffffffL), 0x00020000);
            const int rl = lane >> 3, cl = (lane & 7) * 8;
            const int row0 = m0 + grp * 128 + rl, col0 = n0 + q * 64 + cl;
            const unsigned voff = ((unsigned)row0 * (unsigned)g.ldc + (unsigned)col0) * (unsigned)esz_out;
            const unsigned s8 = 8u * (unsigned)g.ldc * (unsigned)esz_out;           // 8 rows further down
            float bias8[8];
            load_chan8(g.bias, col0, g.N, bias8, 0.f);
            float ga[16][2];
            int gsplit = 8;
            if (g.gadd) {
                const int col = col0 + g.gcol0, gi = col / g.ggroup;
                gsplit = min(8, g.ggroup - (col - gi * g.ggroup));                  // columns of the lane's 8 inside its first group
                const int gi1 = min(gi + 1, (int)g.ld_gadd - 1);
#pragma unroll
                for (int k = 0; k < 16; ++k) {                                      // row block k = 4 mi + it: rows row0 + 8 k
                    const float* r = g.gadd + (long)(row0 + 8 * k) * g.ld_gadd;
                    ga[k][0] = r[gi];
                    ga[k][1] = r[gi1];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        f32x4 v = {acc[mi][ni][4 * c], acc[mi][ni][4 * c + 1], acc[mi][ni][4 * c + 2], acc[mi][ni][4 * c + 3]};
                        *reinterpret_cast<f32x4*>(est + fr * ESTRIDE + (ni * 32 + 8 * c + 4 * fh) * 4) = v;
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(est + (rl + 8 * it) * ESTRIDE + cl * 4);
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(est + (rl + 8 * it) * ESTRIDE + cl * 4 + 16);
                    float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        float t = v[k];
                        if (g.gadd) t += k < gsplit ? ga[mi * 4 + it][0] : ga[mi * 4 + it][1];    // (same order as the general form: addend, bias, act)
                        t += bias8[k];
                        if (g.act == SGG_ACT_RELU) t = fmaxf(t, 0.f);
                        v[k] = t;
                    }
                    const unsigned so = (unsigned)(mi * 4 + it) * s8;
                    if (g.out_dt == SGG_F32) {
                        const u32x4 a = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                        const u32x4 b = {__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])};
                        __builtin_amdgcn_raw_buffer_store_b128(a, rs, voff, so, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(b, rs, voff + 16, so, 0);
                    } else if (g.out_dt == SGG_BF16) {
                        __builtin_amdgcn_raw_buffer_store_b128(pack8<bf16_t>(v), rs, voff, so, 0);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b128(pack8<f16_t>(v), rs, voff, so, 0);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            return;
        }
    }
    const bool vec_ok = CONV || ((g.ldc & 7) == 0);
    // The staging tile is private to the wave (est = smem + wave * ...): after ONE workgroup barrier (every wave has
    // finished reading operand tiles out of this memory) the wave's own LDS write -> read order is all that is needed.
    const ChanVec8 cv = load_chanvec8(g, n0 + q * 64 + (lane & 7) * 8);   // the lane's 8 channels: the same in every store below
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                f32x4 v = {acc[mi][ni][4 * c], acc[mi][ni][4 * c + 1], acc[mi][ni][4 * c + 2], acc[mi][ni][4 * c + 3]};
                *reinterpret_cast<f32x4*>(est + fr * ESTRIDE + (ni * 32 + 8 * c + 4 * fh) * 4) = v;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own LDS writes landed (same wave reads them back)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int rl = (lane >> 3) + 8 * it, cl = (lane & 7) * 8;
            const int m = m0 + grp * 128 + mi * 32 + rl;
            const int n = n0 + q * 64 + cl;
            if (m >= g.M || n >= g.N) continue;
            float v[8];
            const f32x4 lo = *reinterpret_cast<const f32x4*>(est + rl * ESTRIDE + cl * 4);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(est + rl * ESTRIDE + cl * 4 + 16);
            v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
            epilogue_store8(g_, cv, v, m, n, out_offset<CONV>(g_, m, n), vec_ok);
        }
    }
}

template <int DT, bool CONV, bool TN = false>
int launch256(const GemmArgs& g, hipStream_t s, int splits = 1) {
    const int tilesM = (g.M - g.m_base + 255) / 256, tilesN = (g.N + 255) / 256;
    auto k = mfma_pingpong_kernel<DT, CONV, TN>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess)
            return SGG_ERR_LAUNCH;
        attr_done = true;
    }
    hipLaunchKernelGGL(k, dim3(tilesM * tilesN, splits), dim3(512), SMEM, s, g);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

}  // namespace

// g.nt / g.nt1 are in units of 64-byte K-tiles here
int sgg_launch_pingpong(const GemmArgs& g, int dt, bool conv, hipStream_t s) {
    if (dt == SGG_BF16) return conv ? launch256<SGG_BF16, true>(g, s) : launch256<SGG_BF16, false>(g, s);
    if (dt == SGG_F16) return conv ? launch256<SGG_F16, true>(g, s) : launch256<SGG_F16, false>(g, s);
    return conv ? launch256<SGG_F32, true>(g, s) : launch256<SGG_F32, false>(g, s);
}

// split-K launch of the ping-pong kernel: `splits` K slices per 256x256 tile into fp32 partials at g.C + slice * g.splitk_stride
// (g.out_dt = SGG_F32, no bias / activation: the caller's reduce pass applies them).  g.nt in 64-byte K-tiles.
int sgg_launch_pingpong_splitk(const GemmArgs& g, int dt, int splits, hipStream_t s) {
    if (dt == SGG_BF16) return launch256<SGG_BF16, false>(g, s, splits);
    if (dt == SGG_F16) return launch256<SGG_F16, false>(g, s, splits);
    return launch256<SGG_F32, false>(g, s, splits);
}

// TN form: g.A = [Kred][M] (lda_b bytes per reduction row), g.Wt = [Kred][N]; g.nt = K-tiles of 32 reduction rows, the first g.nt1 of them
// in A / Wt, the rest (the zero-padded tail rows) in A2 / W2; M, N multiples of 256; 16-bit operands.
// splits > 1: K slices into fp32 partials (as sgg_launch_pingpong_splitk).
int sgg_launch_pingpong_tn(const GemmArgs& g, int dt, int splits, hipStream_t s) {
    if (dt == SGG_BF16) return launch256<SGG_BF16, false, true>(g, s, splits);
    if (dt == SGG_F16) return launch256<SGG_F16, false, true>(g, s, splits);
    return SGG_ERR_DTYPE;
}
