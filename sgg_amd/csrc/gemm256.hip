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
// SK = true: the persistent stream-K form (tile_sched.h) -- grid = CU count, each workgroup walks its list of work items
// (head part of a split tile, one tile per data-parallel round, whole stream-K tiles, tail part of a split tile) through the same
// K loop; a tile's K order, and with it every output bit, is the plain launch's.
#include "gemm_args.h"
#include "tile_sched.h"

#include <map>
#include <mutex>

namespace {

constexpr int ROW = 64;                 // bytes of K per LDS row
constexpr int STAGE = 512 * ROW;        // A 256 rows + W 256 rows = 32 KiB
constexpr int NSTAGE = 4;
constexpr int SMEM = NSTAGE * STAGE;    // 128 KiB

// byte offset of K-tile kt inside its source rows and which of the two K segments it belongs to (uniform values)
template <bool CONV>
__device__ __forceinline__ long tile_koff(const GemmArgs& g, bool loads_a, int kt, int tpc, int esz, bool& seg2) {
    seg2 = false;
#ifdef SGG_GEMM_ABL_HOT   // experiment: every K-tile re-reads tile 0 (operands stay in L2): what the memory side costs
    kt = 0;
#endif
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

template <int DT, bool CONV, bool SK>
__global__ __launch_bounds__(512) void mfma_pingpong_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ESZ = DT == SGG_F32 ? 4 : 2;
    const int tid0 = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    int tid = tid0, lane = tid0 & 63;
    const int grp = wave >> 2, q = wave & 3;  // grp: which half of M; q: which 64-wide N slab

    const int tilesM = (g.M - g.m_base + 255) / 256, tilesN = (g.N + 255) / 256;
    int m0, n0;

    // ---- DMA duty: waves 0-3 stage the A rows, waves 4-7 the W rows; 4 instructions of 16 rows each
    const bool loads_a = wave < 4;
    // DMA addressing that costs the issuing wave nothing but the instruction itself: global = SGPR base (uniform, advanced
    // with scalar adds per K-tile) + a per-lane 32-bit offset that never changes + the instruction's immediate; the same
    // immediate (j KiB) places piece j in LDS behind ONE M0 value per K-tile.  The immediate applies to both addresses, so
    // the lane offset carries -j KiB (+3 KiB on the lane, -3 KiB on the base keep it unsigned).
    const char* sbase = loads_a ? (CONV ? g.A : g.A) : g.Wt;
    const char* sbase2 = loads_a ? g.A2 : g.W2;
    // (named scalars, not arrays: indexed through a runtime choice they would be demoted to scratch memory)
    unsigned vo_0, vo_1, vo_2, vo_3, vo2_0, vo2_1, vo2_2, vo2_3;
    auto set_tile = [&](int tm, int tn) {       // per-lane source offsets of the tile's rows (the only per-tile addressing state)
        m0 = g.m_base + tm * 256;
        n0 = tn * 256;
        const char* rp[4];
        const char* rp2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
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
#define SGG_VO(j) (unsigned)(rp[j] - sbase) + 3072u - 1024u * j
#define SGG_VO2(j) (rp2[j] ? (unsigned)(rp2[j] - sbase2) + 3072u - 1024u * j : 0u)
        vo_0 = SGG_VO(0); vo_1 = SGG_VO(1); vo_2 = SGG_VO(2); vo_3 = SGG_VO(3);
        vo2_0 = SGG_VO2(0); vo2_1 = SGG_VO2(1); vo2_2 = SGG_VO2(2); vo2_3 = SGG_VO2(3);
#undef SGG_VO
#undef SGG_VO2
    };
    const int tpc = CONV ? (g.Cin * ESZ) / ROW : 1;
    const int lds_rows0 = (loads_a ? 0 : 256 * ROW) + (wave & 3) * 64 * ROW;
    auto issue = [&](int kt) {
        bool seg2;
        const long koff = tile_koff<CONV>(g, loads_a, kt, tpc, ESZ, seg2);
        const char* ub = uniform_ptr((seg2 ? sbase2 : sbase) + koff - 3072);      // the -3 KiB pairs with the +3 KiB inside vo_*
        char* dst = smem + (kt & (NSTAGE - 1)) * STAGE + lds_rows0;
        glds16_off<0>(ub + (seg2 ? vo2_0 : vo_0), dst);
        glds16_off<1024>(ub + (seg2 ? vo2_1 : vo_1), dst);
        glds16_off<2048>(ub + (seg2 ? vo2_2 : vo_2), dst);
        glds16_off<3072>(ub + (seg2 ? vo2_3 : vo_3), dst);
    };

    // ---- fragment addressing: row = base + (lane&31); logical slot = 2*s + (lane>>5)
    int fr, fh;
    int aoff[4], akey[4], boff[2], bkey[2];
    auto fill_frag_addr = [&]() {
        fr = lane & 31; fh = lane >> 5;
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
    };

    f32x16 acc[4][2];

    u32x4 af[4][2], bf[2][2];

    auto load_frags = [&](int kt) {
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
        bool seg2;
        const long koff = tile_koff<CONV>(g, loads_a, pf, tpc, ESZ, seg2);
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

    // K-tiles [kb, nt) of the tile whose first three K-tiles are already in flight (issue()) and whose accumulators are zero; from_slot >= 0:
    // the tile's head part was published by another workgroup (stream-K) -- its accumulator registers are added to the zeros first.
    // Two s_barriers per K-tile (measured better than one: 1053 vs 1008 TFLOP/s on fc6 -- with a single barrier the
    // two waves of a SIMD drift into the same phase).  Tile kt+3 goes into the stage tile kt-1 occupied; its last
    // readers (group 1, LOAD kt-1) retired their ds_reads before the barrier that ended slot 2kt-1.
    // The K loop is peeled: nt-3 steady iterations whose MFMA phase carries the DMA of tile kt+3, then the last three without
    // any -- one copy of the MFMA block per loop, no branch inside a phase.
    auto k_loop = [&](int kb, int nt, int from_slot) {
        // The tail part of a split tile continues the chain its head part's workgroup published: that workgroup's accumulator
        // registers, read write-through-coherent (sc1) and ADDED to the zeroed accumulators (0 + x = x).  Called at the head of each
        // wave group's K loop, not before the branch: defined there, the 128 loaded registers are spilled and reloaded around the
        // prologue's barrier and the group branch (the allocator cannot carry them into four loops' phis; zeros it re-materialises).
        // In chunks of SK_CHUNK MFMA tiles with the compiler's own wait builtin between them: it neither hoists loads across it (all
        // 128 registers at once = 128 temporaries beside the accumulators) nor puts a vmcnt(0) in front of every MFMA of the K loop.
        auto add_partial = [&]() {
            if (SK && from_slot >= 0) {
                const __amdgpu_buffer_rsrc_t rs = sk_slot_rsrc(g.sk_ws, from_slot);
                constexpr int SK_CHUNK = 4;
#pragma unroll
                for (int c0 = 0; c0 < 8; c0 += SK_CHUNK) {
                    f32x4 t[SK_CHUNK][4];
#pragma unroll
                    for (int u = 0; u < SK_CHUNK; ++u)
#pragma unroll
                        for (int c = 0; c < 4; ++c) t[u][c] = sk_load16(rs, (c0 + u) * 4 + c, tid);
                    __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0)
#pragma unroll
                    for (int u = 0; u < SK_CHUNK; ++u)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            f32x16& a = acc[(c0 + u) >> 1][(c0 + u) & 1];
                            a[4 * c] += t[u][c].x; a[4 * c + 1] += t[u][c].y; a[4 * c + 2] += t[u][c].z; a[4 * c + 3] += t[u][c].w;
                        }
                }
            }
        };
        const int nsteady = max(nt - 3, kb);
        if (grp == 0) {
            add_partial();
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
            add_partial();
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
    };
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    auto prologue = [&](int kb, int nt) {                     // K-tiles kb .. kb+2 in flight
#pragma unroll
        for (int t = 0; t < 3; ++t)
            if (kb + t < nt) issue(kb + t);
    };
    auto prologue_wait = [&](int kb, int nt) {                // ... K-tile kb landed
        if (nt - kb >= 3) wait_vmcnt<8>();
        else if (nt - kb == 2) wait_vmcnt<4>();
        else wait_vmcnt<0>();
    };

    if constexpr (!SK) {
        // ---- plain launch: the one tile of blockIdx.x
        int tm, tn;
        tile_coords(blockIdx.x, tilesM, tilesN, tm, tn);
        set_tile(tm, tn);
        fill_frag_addr();
        const int nt = g.nt;
        prologue(0, nt);
        zero_acc();
        prologue_wait(0, nt);
        __builtin_amdgcn_s_barrier();
        k_loop(0, nt, -1);

        // ---- epilogue through LDS: per wave a [32][64] f32 staging tile (row stride 272 B), one 32-row block at a time
        constexpr int ESTRIDE = 272;
        char* est = smem + wave * (32 * ESTRIDE);
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
                epilogue_store8(g, cv, v, m, n, out_offset<CONV>(g, m, n), vec_ok);
            }
        }
    } else {
        // ---- stream-K launch (tile_sched.h): this workgroup's work items -- head part of a split tile FIRST (its accumulators are
        // published for the workgroup that owns the tile's tail), one tile per data-parallel round, the whole tiles of its K-unit
        // range, the tail part of a split tile LAST (continues the published chain).  The items are software-pipelined: when an item's
        // K loop ends, the NEXT item's first three K-tiles are put in flight (ring stages kb..kb+2) BEFORE this item's epilogue, which
        // stages its accumulators through the one ring stage they leave free ((kb + 3) & 3: 4 KiB per wave, a 32 x 32 f32 block at a
        // time) and leaves its global stores in flight behind it: the counted vmcnt waits of the next K loop only get more
        // conservative from older operations in the queue (csrc/imp.hip's step kernel relies on the same in-order rule).
        const int G = gridDim.x;
        const int lc = xcd_remap(blockIdx.x, G);                  // XCD-contiguous: neighbours in tile order share an L2
        const int sk_base = g.sk_dp_rounds * G;
        SkRange rg{};
        if (g.sk_tiles > 0) rg = sk_range(lc, G, g.sk_tiles, g.nt);
        const int has_head = rg.head_ke > 0;
        const int n_items = has_head + g.sk_dp_rounds + rg.n_whole + (rg.tail_kb > 0);
        volatile unsigned* const flag_word = reinterpret_cast<volatile unsigned*>(smem + SMEM);   // (the 16 bytes behind the ring)

        int kb = 0, ke = 0, to_slot = -1, from_slot = -1;
        // item -> (tile, K range, hand-over slots); sets the tile's addressing and, for the tail part, looks for the published head
        auto setup = [&](int item) {
            int L, j = item;
            kb = 0; ke = g.nt; to_slot = -1; from_slot = -1;
            if (has_head && j == 0) {
                L = sk_base + rg.head_tile; ke = rg.head_ke; to_slot = lc;
            } else {
                j -= has_head;
                if (j < g.sk_dp_rounds) {
                    L = j * G + lc;
                } else {
                    j -= g.sk_dp_rounds;
                    if (j < rg.n_whole) L = sk_base + rg.whole0 + j;
                    else { L = sk_base + rg.tail_tile; kb = rg.tail_kb; from_slot = lc - 1; }
                }
            }
            // (epoch bit 31: test mode 3, no consumer finds its head part)
            if (from_slot >= 0 && ((g.sk_epoch >> 31) || !sk_ready(g.sk_flags, from_slot, g.sk_epoch, tid, flag_word))) {
                from_slot = -1;                                   // the head part is not there (its workgroup has not run yet): the whole
                kb = 0;                                           // chain here, from K-tile 0 -- same order, same bits, nobody waits
            }
            int tm, tn;
            tile_coords_logical(L, tilesM, tilesN, tm, tn);
            set_tile(tm, tn);
        };
        if (n_items > 0) {
            setup(0);
            prologue(kb, ke);
        }
        for (int item = 0; item < n_items; ++item) {
            tid = tid0;
            asm volatile("" : "+v"(tid));                         // (keeps per-lane addressing from being hoisted out of the item loop,
            lane = tid & 63;                                      //  where it would sit in registers across every K loop)
            fill_frag_addr();
            const int cur_kb = kb, cur_ke = ke, cur_from = from_slot, cur_to = to_slot, em0 = m0, en0 = n0;
            zero_acc();
            if (item == 0) prologue_wait(cur_kb, cur_ke);        // (later items: their first K-tile was waited for inside the previous epilogue)
            __builtin_amdgcn_s_barrier();
            k_loop(cur_kb, cur_ke, cur_from);

            if (cur_to >= 0) {
                // head part of a split tile: hand the accumulators to the tail part's workgroup as they are (no epilogue); the next
                // item's prologue follows the publication (its drain would wait for the DMA as well)
                const __amdgpu_buffer_rsrc_t rs = sk_slot_rsrc(g.sk_ws, cur_to);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const f32x4 v = {acc[i][j][4 * c], acc[i][j][4 * c + 1], acc[i][j][4 * c + 2], acc[i][j][4 * c + 3]};
                            sk_store16(rs, (i * 2 + j) * 4 + c, tid, v);
                        }
                sk_publish(g.sk_flags, cur_to, g.sk_epoch, tid);
                if (item + 1 < n_items) {
                    setup(item + 1);
                    prologue(kb, ke);
                    prologue_wait(kb, ke);
                }
                continue;
            }

            // ---- epilogue.  The per-channel vectors first (ordinary loads: the compiler waits for them in order, i.e. for everything
            // older -- so they go before the next item's DMA), then the next item's prologue, then the staged stores.
            const bool vec_ok = CONV || ((g.ldc & 7) == 0);
            const int lane_e = lane, fr_e = lane_e & 31, fh_e = lane_e >> 5;
            ChanVec8 cv[2];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) cv[ni] = load_chanvec8(g, en0 + q * 64 + ni * 32 + (lane_e & 3) * 8);
            int free_stage = 3;
            const bool more = item + 1 < n_items;
            if (more) {
                setup(item + 1);
                prologue(kb, ke);
                free_stage = (kb + 3) & (NSTAGE - 1);
            }
            // per wave a [32 rows][32 columns] f32 block (4 KiB, rows of eight 16-byte chunks, chunk ^= row & 7: the 64 lanes of a write
            // and the 16 rows of a read each cover all banks once)
            char* est = smem + free_stage * STAGE + wave * 4096;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        f32x4 v = {acc[mi][ni][4 * c], acc[mi][ni][4 * c + 1], acc[mi][ni][4 * c + 2], acc[mi][ni][4 * c + 3]};
                        *reinterpret_cast<f32x4*>(est + fr_e * 128 + (((2 * c + fh_e) ^ (fr_e & 7)) << 4)) = v;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own LDS writes landed (same wave reads them back)
                    if (more && ni == 0 && mi == 0) prologue_wait(kb, ke);   // next item's first K-tile landed -- before any store is queued behind it
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const int rl = (lane_e >> 2) + 16 * it, ch = (lane_e & 3) * 2;
                        const int m = em0 + grp * 128 + mi * 32 + rl;
                        const int n = en0 + q * 64 + ni * 32 + (lane_e & 3) * 8;
                        float v[8];
                        const f32x4 lo = *reinterpret_cast<const f32x4*>(est + rl * 128 + ((ch ^ (rl & 7)) << 4));
                        const f32x4 hi = *reinterpret_cast<const f32x4*>(est + rl * 128 + (((ch + 1) ^ (rl & 7)) << 4));
                        if (m >= g.M || n >= g.N) continue;
                        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
                        epilogue_store8(g, cv[ni], v, m, n, out_offset<CONV>(g, m, n), vec_ok);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the block is rewritten by the next one
                }
            }
            // (the barrier at the head of the next item's K loop: every wave is done with its staging block before K-tile kb+3 lands there)
        }
    }
}

template <int DT, bool CONV>
int launch256(const GemmArgs& g, hipStream_t s) {
    const int tilesM = (g.M - g.m_base + 255) / 256, tilesN = (g.N + 255) / 256;
    auto k = mfma_pingpong_kernel<DT, CONV, false>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess)
            return SGG_ERR_LAUNCH;
        attr_done = true;
    }
    hipLaunchKernelGGL(k, dim3(tilesM * tilesN), dim3(512), SMEM, s, g);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

template <int DT, bool CONV>
int launch256_sk(const GemmArgs& g, int grid, hipStream_t s) {
    auto k = mfma_pingpong_kernel<DT, CONV, true>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM + 16) != hipSuccess)
            return SGG_ERR_LAUNCH;
        attr_done = true;
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), SMEM + 16, s, g);      // + the word sk_ready broadcasts through
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

std::mutex sk_mu;
std::map<void*, SkWorkspace> sk_reg;      // stream -> its workspace (launches of one stream are ordered: one set of slots suffices)

}  // namespace

SkWorkspace* sgg_sk_workspace_of(void* stream) {
    std::lock_guard<std::mutex> lk(sk_mu);
    auto it = sk_reg.find(stream);
    return it == sk_reg.end() ? nullptr : &it->second;
}

int sgg_sk_grid() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            return 0;
        n = cus < SK_MAX_GRID ? cus : SK_MAX_GRID;
    }
    return n;
}

// Registers (ws != NULL) or forgets (ws == NULL) the stream-K workspace of `stream`: bytes >= sgg_streamk_workspace_bytes(), 256-byte
// aligned, its flag words ZEROED once by the caller (they carry launch epochs afterwards and are never cleared again).
extern "C" int64_t sgg_streamk_workspace_bytes(void) { return (int64_t)SK_MAX_GRID * SK_SLOT_BYTES + (int64_t)SK_MAX_GRID * 4; }

extern "C" int sgg_streamk_workspace(void* stream, void* ws, int64_t bytes) {
    std::lock_guard<std::mutex> lk(sk_mu);
    if (!ws) {
        sk_reg.erase(stream);
        return SGG_OK;
    }
    if (((uintptr_t)ws & 255) || bytes < sgg_streamk_workspace_bytes()) return SGG_ERR_ARG;
    SkWorkspace w;
    w.flags = reinterpret_cast<unsigned*>(ws);                    // SK_MAX_GRID words first (the part the caller zeroes) ...
    w.slots = reinterpret_cast<char*>(ws) + SK_MAX_GRID * 4;      // ... then the slots
    w.epoch = 0;
    w.max_grid = SK_MAX_GRID;
    sk_reg[stream] = w;
    return SGG_OK;
}

static int sk_mode_value = -1;
static int sk_mode() {
    if (sk_mode_value < 0) {
        const char* env = getenv("SGG_STREAMK");
        sk_mode_value = env && (env[0] == '0' || env[0] == '1') ? env[0] - '0' : 2;
    }
    return sk_mode_value;
}
// 0: never stream-K; 1: wherever it applies; 2: where a plain launch would leave >= 4 % of its tile slots empty (the default; the
// environment variable SGG_STREAMK = 0 / 1 sets the initial value).  Returns the previous mode.
extern "C" int sgg_streamk_mode(int mode) {
    const int prev = sk_mode();
    if (mode >= 0 && mode <= 3) sk_mode_value = mode;          // (3, tests: as 1, and no consumer finds its head part published)
    return prev;
}

// 1 when a ping-pong launch of `tiles` tiles with `nt` (64-byte) K-tiles on `stream` would take the stream-K form (callers that
// otherwise cut a nearly empty last round off into another launch ask first)
int sgg_pingpong_streamk(long tiles, int nt, void* stream, int* dp_rounds, int* sk_tiles) {
    const int mode = sk_mode();                                  // 0: never; 1: whenever it applies; 2: when a round is >= 4 % idle
    if (mode == 0) return 0;
    const int G = sgg_sk_grid();
    if (G <= 0 || !sgg_sk_workspace_of(stream)) return 0;
    int dp = 0, sk = 0;
    sk_split(tiles, G, dp, sk);
    if (sk == 0 || nt < 2 * SK_MIN_SEG) return 0;
    const long rounds = (tiles + G - 1) / G;
    const double idle = 1.0 - (double)tiles / ((double)rounds * G);
    if (mode == 2 && idle < 0.04) return 0;
    if (dp_rounds) *dp_rounds = dp;
    if (sk_tiles) *sk_tiles = sk;
    return 1;
}

// g.nt / g.nt1 are in units of 64-byte K-tiles here
int sgg_launch_pingpong(const GemmArgs& g_in, int dt, bool conv, hipStream_t s) {
    GemmArgs g = g_in;
    const long tiles = (long)((g.M - g.m_base + 255) / 256) * ((g.N + 255) / 256);
    int dp = 0, sk = 0;
    if (g.m_base == 0 && sgg_pingpong_streamk(tiles, g.nt, (void*)s, &dp, &sk)) {
        SkWorkspace* w = sgg_sk_workspace_of((void*)s);
        const int G = sgg_sk_grid();
        w->epoch = (w->epoch + 1) & 0x7fffffffu;
        if (w->epoch == 0) w->epoch = 1;
        g.sk_ws = w->slots; g.sk_flags = w->flags; g.sk_epoch = w->epoch | (sk_mode() == 3 ? 0x80000000u : 0u); g.sk_dp_rounds = dp; g.sk_tiles = sk;
        if (dt == SGG_BF16) return conv ? launch256_sk<SGG_BF16, true>(g, G, s) : launch256_sk<SGG_BF16, false>(g, G, s);
        if (dt == SGG_F16) return conv ? launch256_sk<SGG_F16, true>(g, G, s) : launch256_sk<SGG_F16, false>(g, G, s);
        return conv ? launch256_sk<SGG_F32, true>(g, G, s) : launch256_sk<SGG_F32, false>(g, G, s);
    }
    if (dt == SGG_BF16) return conv ? launch256<SGG_BF16, true>(g, s) : launch256<SGG_BF16, false>(g, s);
    if (dt == SGG_F16) return conv ? launch256<SGG_F16, true>(g, s) : launch256<SGG_F16, false>(g, s);
    return conv ? launch256<SGG_F32, true>(g, s) : launch256<SGG_F32, false>(g, s);
}
