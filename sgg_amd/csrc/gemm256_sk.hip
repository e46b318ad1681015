// Persistent stream-K form of the 256x256 ping-pong MFMA kernel (gemm256.hip holds the plain one-tile-per-workgroup form and the
// schedule's description; tile_sched.h how a launch is cut into work items and how a split tile stays ONE accumulation chain).
//
// What this file adds to the plain kernel's K loop is what happens BETWEEN tiles.  tools/sk_trace.py (s_memrealtime stamps) on the plain
// structure: of a 140-us tile of the fc6 weight gradient, 4 us go to the epilogue's loads, 10-15 us to its staged stores and 2-3 us to
// the next tile's prologue -- and every one of these is queue ORDER, not bandwidth: vmcnt retires in order, so an ordinary load issued
// after a store waits for that store's round trip, a DMA piece issued after the stores is "landed" only when they are, and a register
// the allocator spills in the epilogue is reloaded behind all of them.  Hence:
//   * ONE continuous K-tile stream per workgroup: the last three MFMA phases of an item issue the first three K-tiles of the NEXT item
//     (ring stages keep rotating, sequence numbers instead of K indices), so a tile's epilogue begins with the next tile's operands
//     already in LDS and issues nothing but stores;
//   * the epilogue's inputs (bias, the group addend of fc6's weight gradient) are LDS-DMA'd into 9 KiB behind the ring at the START of
//     the item, its per-lane addressing is rebuilt from an opaque lane id per item (nothing is hoisted across the K loop): no global
//     load and no spill sits between the stores;
//   * the accumulators go through the one ring stage the next item's first three K-tiles leave free (4 KiB per wave, a 32 x 32 f32
//     block at a time) and the stores stay in flight behind the wave; the first two waits of the next K loop count them in
//     (interior tiles issue exactly 16 / 32 store instructions per wave; an edge tile drains instead).
#include "gemm_args.h"
#include "tile_sched.h"

#include <map>
#include <mutex>

#ifdef SGG_SK_TRACE
// experiment build only (make trace; tools/sk_trace.py): s_memrealtime stamps (100 MHz) of workgroup 0 / wave 0 at the phases of every item
__device__ unsigned long long* sk_trace_buf = nullptr;
extern "C" int sgg_sk_trace_buffer(void* p) { return hipMemcpyToSymbol(HIP_SYMBOL(sk_trace_buf), &p, sizeof(p)) == hipSuccess ? 0 : -3; }
#define SK_STAMP(slot) do { if (blockIdx.x == 0 && tid0 == 0 && sk_trace_buf) sk_trace_buf[(item) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SK_STAMP(slot) do { } while (0)
#endif

namespace {

constexpr int ROW = 64;                 // bytes of K per LDS row
constexpr int STAGE = 512 * ROW;        // A 256 rows + W 256 rows = 32 KiB
constexpr int NSTAGE = 4;
constexpr int RING = NSTAGE * STAGE;    // 128 KiB
constexpr int X_FLAG = RING;            // the word sk_ready broadcasts through
constexpr int X_BIAS = RING + 64;       // f32[256]: the tile's bias
constexpr int X_GADD = X_BIAS + 1024;   // f32[8 groups][256 rows]: the tile's group addends
constexpr int SMEM_SK = X_GADD + 8192;  // 137.1 KiB

// byte offset of K-tile kt inside its source rows and which of the two K segments it belongs to (uniform values)
template <bool CONV>
__device__ __forceinline__ long sk_koff(const GemmArgs& g, bool loads_a, int kt, int tpc, int esz, bool& seg2) {
    seg2 = false;
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

// the only per-tile addressing state of the DMA: per-lane source offsets of this wave's four 16-row pieces (both K segments)
struct TileAddr {
    unsigned v0, v1, v2, v3, w0, w1, w2, w3;
    int m0, n0;
};

__device__ __forceinline__ void wait_vm_rt(int n) {      // s_waitcnt vmcnt(n) for a run-time (uniform) n: twice per work item
    switch (n) {
#define SK_W(N) case N: wait_vmcnt<N>(); break;
        SK_W(1) SK_W(2) SK_W(3) SK_W(4) SK_W(5) SK_W(6) SK_W(7) SK_W(8) SK_W(9) SK_W(10) SK_W(11) SK_W(12) SK_W(13) SK_W(14) SK_W(15) SK_W(16)
        SK_W(17) SK_W(18) SK_W(19) SK_W(20) SK_W(21) SK_W(22) SK_W(23) SK_W(24) SK_W(25) SK_W(26) SK_W(27) SK_W(28) SK_W(29) SK_W(30) SK_W(31)
        SK_W(32) SK_W(33) SK_W(34) SK_W(35) SK_W(36) SK_W(37) SK_W(38) SK_W(39) SK_W(40) SK_W(41) SK_W(42) SK_W(43) SK_W(44) SK_W(45) SK_W(46)
        SK_W(47) SK_W(48)
#undef SK_W
        default: wait_vmcnt<0>(); break;
    }
}

template <int DT, bool CONV>
__global__ __launch_bounds__(512) void mfma_pingpong_sk_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ESZ = DT == SGG_F32 ? 4 : 2;
    const int tid0 = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    int tid = tid0, lane = tid0 & 63;
    const int grp = wave >> 2, q = wave & 3;  // grp: which half of M; q: which 64-wide N slab
    const int tilesM = (g.M + 255) / 256, tilesN = (g.N + 255) / 256;

    // ---- DMA duty: waves 0-3 stage the A rows, waves 4-7 the W rows; 4 instructions of 16 rows each (addressing: gemm256.hip)
    const bool loads_a = wave < 4;
    const char* sbase = loads_a ? g.A : g.Wt;
    const char* sbase2 = loads_a ? g.A2 : g.W2;
    auto set_tile = [&](TileAddr& t, int tm, int tn) {
        t.m0 = tm * 256;
        t.n0 = tn * 256;
        const char* rp[4];
        const char* rp2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = (wave & 3) * 64 + j * 16 + (lane >> 2);
            const int chunk = (lane & 3) ^ ((r >> 2) & 3);
            if (loads_a) {
                const int m = min(t.m0 + r, g.M - 1);
                rp[j] = a_row_ptr<CONV>(g, m, ESZ) + chunk * 16;
                rp2[j] = (!CONV && g.A2) ? g.A2 + (long)m * g.lda2_b + chunk * 16 : nullptr;
            } else {
                const int n = min(t.n0 + r, g.N - 1);
                rp[j] = g.Wt + (long)n * g.ldw_b + chunk * 16;
                rp2[j] = (!CONV && g.W2) ? g.W2 + (long)n * g.ldw2_b + chunk * 16 : nullptr;
            }
        }
#define SGG_VO(j) (unsigned)(rp[j] - sbase) + 3072u - 1024u * j
#define SGG_VO2(j) (rp2[j] ? (unsigned)(rp2[j] - sbase2) + 3072u - 1024u * j : 0u)
        t.v0 = SGG_VO(0); t.v1 = SGG_VO(1); t.v2 = SGG_VO(2); t.v3 = SGG_VO(3);
        t.w0 = SGG_VO2(0); t.w1 = SGG_VO2(1); t.w2 = SGG_VO2(2); t.w3 = SGG_VO2(3);
#undef SGG_VO
#undef SGG_VO2
    };
    const int tpc = CONV ? (g.Cin * ESZ) / ROW : 1;
    const int lds_rows0 = (loads_a ? 0 : 256 * ROW) + (wave & 3) * 64 * ROW;
    // the four pieces of K-tile kt of tile t into ring stage seq & 3
    auto issue = [&](const TileAddr& t, int kt, int seq) {
        bool seg2;
        const long koff = sk_koff<CONV>(g, loads_a, kt, tpc, ESZ, seg2);
        const char* ub = uniform_ptr((seg2 ? sbase2 : sbase) + koff - 3072);      // the -3 KiB pairs with the +3 KiB inside the lane offsets
        char* dst = smem + (seq & (NSTAGE - 1)) * STAGE + lds_rows0;
        glds16_off<0>(ub + (seg2 ? t.w0 : t.v0), dst);
        glds16_off<1024>(ub + (seg2 ? t.w1 : t.v1), dst);
        glds16_off<2048>(ub + (seg2 ? t.w2 : t.v2), dst);
        glds16_off<3072>(ub + (seg2 ? t.w3 : t.v3), dst);
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
    auto load_frags = [&](int seq) {
        const char* st = smem + (seq & (NSTAGE - 1)) * STAGE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int slot = 2 * s + fh;
#pragma unroll
            for (int i = 0; i < 2; ++i) bf[i][s] = *reinterpret_cast<const u32x4*>(st + boff[i] + ((slot ^ bkey[i]) << 4));
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i][s] = *reinterpret_cast<const u32x4*>(st + aoff[i] + ((slot ^ akey[i]) << 4));
        }
    };

#define SGG_MFMA_PAIR(s, mi)                                                                                              \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) {                                                                    \
        if constexpr (DT != SGG_F32) {                                                                                    \
            acc[mi][ni] = mfma_32x32x16<DT>(bf[ni][s], af[mi][s], acc[mi][ni]);                                           \
        } else {                                                                                                          \
            _Pragma("unroll") for (int c = 0; c < 4; ++c) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(             \
                __uint_as_float(bf[ni][s][c]), __uint_as_float(af[mi][s][c]), acc[mi][ni], 0, 0, 0);                      \
        }                                                                                                                 \
    }
    // 16 MFMAs with the 4 DMA pieces of K-tile `pf` of tile t (ring stage seq_pf & 3) between them (gemm256.hip: compute_dma)
    auto compute_dma = [&](const TileAddr& t, int pf, int seq_pf) {
        bool seg2;
        const long koff = sk_koff<CONV>(g, loads_a, pf, tpc, ESZ, seg2);
        const char* ub = uniform_ptr((seg2 ? sbase2 : sbase) + koff - 3072);
        char* dst = smem + (seq_pf & (NSTAGE - 1)) * STAGE + lds_rows0;
        const unsigned o0 = seg2 ? t.w0 : t.v0, o1 = seg2 ? t.w1 : t.v1, o2 = seg2 ? t.w2 : t.v2, o3 = seg2 ? t.w3 : t.v3;
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
#undef SGG_MFMA_PAIR
    // wait until this wave's DMA pieces of tile kt+1 have landed; `later` = tiles it issued after kt+1 (last item only)
    auto wait_tiles_in_flight = [&](int later) {
        if (later >= 2) wait_vmcnt<8>();
        else if (later == 1) wait_vmcnt<4>();
        else wait_vmcnt<0>();
    };

    // ---- this workgroup's work items (tile_sched.h): head part of a split tile FIRST (its accumulators are published for the workgroup
    // that owns the tile's tail), one tile per data-parallel round, the whole tiles of its K-unit range, the tail part of a split
    // tile LAST (continues the published chain, or -- if that is not there yet -- computes the whole chain itself)
    const int G = gridDim.x;
    const int lc = xcd_remap(blockIdx.x, G);                  // XCD-contiguous: neighbours in tile order share an L2
    const int sk_base = g.sk_dp_rounds * G;
    SkRange rg{};
    if (g.sk_tiles > 0) rg = sk_range(lc, G, g.sk_tiles, g.nt);
    const int has_head = rg.head_ke > 0;
    const int n_items = has_head + g.sk_dp_rounds + rg.n_whole + (rg.tail_kb > 0);
    volatile unsigned* const flag_word = reinterpret_cast<volatile unsigned*>(smem + X_FLAG);
    float* const xb = reinterpret_cast<float*>(smem + X_BIAS);
    float* const xg = reinterpret_cast<float*>(smem + X_GADD);
    const int x_ops = (g.bias ? 1 : 0) + (g.gadd ? 4 : 0);   // LDS-DMA instructions per wave that fetch an item's epilogue inputs

    // item -> (tile, K range, hand-over slots)
    struct Item { int kb, ke, to_slot, from_slot, tm, tn; };
    auto setup = [&](int item) -> Item {
        Item it{0, g.nt, -1, -1, 0, 0};
        int L, j = item;
        if (has_head && j == 0) {
            L = sk_base + rg.head_tile; it.ke = rg.head_ke; it.to_slot = lc;
        } else {
            j -= has_head;
            if (j < g.sk_dp_rounds) {
                L = j * G + lc;
            } else {
                j -= g.sk_dp_rounds;
                if (j < rg.n_whole) L = sk_base + rg.whole0 + j;
                else { L = sk_base + rg.tail_tile; it.kb = rg.tail_kb; it.from_slot = lc - 1; }
            }
        }
        tile_coords_logical(L, tilesM, tilesN, it.tm, it.tn);
        return it;
    };

    // Is the head part of this workgroup's tail item published?  ONE wave polls the flag (relaxed, agent scope) and leaves the answer in
    // LDS; the others read it behind a barrier they pass anyway.  The producer ran its head part at the very start of the launch, so
    // the answer is normally yes long before it is asked; when the consumer gets there first (few items per workgroup) it waits up to
    // the time the head part needs -- waiting is never dearer than computing those K-tiles again -- and past that (the producer is
    // not resident) gives up: the caller then runs the whole chain itself from K-tile 0, the same order and bits.  No unbounded wait.
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
    auto poll_head = [&](int slot) {                          // called by ONE wave
        unsigned ok = 0;
        if (!(g.sk_epoch >> 31)) {                            // (epoch bit 31: test mode 3, no consumer finds its head part)
            const unsigned long long deadline = t_start + (unsigned long long)rg.tail_kb * 200 + 2000;   // 100 MHz ticks: 2 us per K-tile + 20 us
            for (;;) {
                ok = __builtin_amdgcn_readfirstlane(__hip_atomic_load((sk_gu32*)(g.sk_flags + slot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == g.sk_epoch ? 1u : 0u;   // (kept scalar: a per-lane loop condition here would make the K loop around it divergent)
                if (ok || __builtin_amdgcn_s_memrealtime() > deadline) break;
                __builtin_amdgcn_s_sleep(32);
            }
        }
        if (lane == 0) *flag_word = ok;
    };
    auto take_answer = [&](Item& it) {                        // every wave, behind a barrier after poll_head
        if (__builtin_amdgcn_readfirstlane(*flag_word) == 0u) {
            it.from_slot = -1;
            it.kb = 0;
        }
    };

    if (n_items == 0) return;
    TileAddr pf;                                              // addressing of the tile being PREFETCHED: this item's, and from its
    Item ci = setup(0), ni{};                                 // third-last iteration on the next item's
    if (ci.from_slot >= 0) {                                  // the tail part is this workgroup's first item
        if (wave == 0) poll_head(ci.from_slot);
        __syncthreads();
        take_answer(ci);
        __syncthreads();
    }
    set_tile(pf, ci.tm, ci.tn);
    int seq0 = 0;                                             // sequence number (ring position) of the item's first K-tile
#pragma unroll
    for (int t = 0; t < 3; ++t) issue(pf, ci.kb + t, seq0 + t);        // (every part is >= SK_MIN_SEG K-tiles long)
    int extra = 0;                                            // vm operations this wave queued between the item's third and fourth K-tile

    for (int item = 0; item < n_items; ++item) {
        tid = tid0;
        asm volatile("" : "+v"(tid));                         // (an opaque lane id per item: nothing per-lane is hoisted out of the item loop,
        lane = tid & 63;                                      //  where it would sit in registers across every K loop)
        fill_frag_addr();
        const int kb = ci.kb, ke = ci.ke, n = ke - kb, em0 = ci.tm * 256, en0 = ci.tn * 256;
        const bool more = item + 1 < n_items;
        SK_STAMP(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        if (item == 0) wait_vmcnt<8>();                       // first K-tile landed (later items: the previous K loop waited for it)
        __builtin_amdgcn_s_barrier();                         // ... and visible; every wave is past the previous item's epilogue
        // the epilogue's inputs, fetched NOW by LDS-DMA (nothing but stores will sit in the queue at epilogue time): bias[256] and the
        // [8 groups][256 rows] of group addends the tile's columns touch; the same number of instructions on every wave
        const int g0 = g.gadd ? (en0 + g.gcol0) / g.ggroup : 0;
        if (ci.to_slot < 0) {
            if (g.bias) {
                const int col = min(en0 + (wave & 3) * 64 + lane, g.N - 1);
                __builtin_amdgcn_global_load_lds((glb_void_t*)(g.bias + col), (lds_void_t*)(smem + X_BIAS + (wave & 3) * 256), 4, 0, 0);
            } else if (tid < 256) {
                xb[tid] = 0.f;
            }
            if (g.gadd) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int idx = wave * 4 + i, j = idx >> 2, rb = idx & 3;
                    const int m = min(em0 + rb * 64 + lane, g.M - 1), gi = min(g0 + j, (int)g.ld_gadd - 1);
                    __builtin_amdgcn_global_load_lds((glb_void_t*)(g.gadd + (long)m * g.ld_gadd + gi), (lds_void_t*)(smem + X_GADD + (j * 256 + rb * 64) * 4), 4, 0, 0);
                }
            }
            extra += x_ops;
        }
        SK_STAMP(1);

        // ---- K loop: K-tiles [kb, ke) at ring positions seq0 .., the stream running on into the next item's first three K-tiles.
        // Two s_barriers per K-tile, group 1 one phase behind group 0 (gemm256.hip).  extra = what the wave queued between this item's
        // K-tiles kb+2 and kb+3 (previous epilogue's stores + the fetches above): counted into the first two waits, which look
        // for K-tiles OLDER than those operations.
        auto add_partial = [&]() {
            // the tail part of a split tile continues the chain its head part's workgroup published: that workgroup's accumulator registers,
            // read write-through-coherent (sc1) and ADDED to the zeroed accumulators (0 + x = x) at the head of each wave group's loop
            // (defined before the group branch the allocator spills all 128 around it), in chunks with the compiler's own wait builtin
            // between them (it neither hoists loads across it nor re-waits in front of every MFMA)
            if (ci.from_slot >= 0) {
                const __amdgpu_buffer_rsrc_t rs = sk_slot_rsrc(g.sk_ws, ci.from_slot);
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
        if (more) ni = setup(item + 1);
        const bool next_is_tail = more && ni.from_slot >= 0;
        const int ex = extra;
        // one loop per group for every iteration that prefetches: K-tile kt+3 of this tile, or -- in the last three iterations of an item
        // that has a successor -- the successor's K-tile kt+3-ke (a select between two sets of lane offsets, made before the first MFMA);
        // the last item's final three iterations prefetch nothing
        const int k_dma_end = more ? ke : ke - 3;
        int kshift = 0;
        // third-last iteration of an item with a successor: from here on the prefetches are the successor's K-tiles kb', kb'+1, kb'+2 --
        // the ONE set of lane offsets is rebuilt for its tile (every K-tile of this tile has been issued), K indices shift by kb' - ke
        auto switch_tile = [&]() {
            if (next_is_tail) take_answer(ni);
            set_tile(pf, ni.tm, ni.tn);
            kshift = ni.kb - ke;
        };
        if (grp == 0) {
            add_partial();
#pragma nounroll
            for (int kt = kb; kt < k_dma_end; ++kt) {
                const int sq = seq0 + kt - kb;
                if (next_is_tail && kt == ke - 4 && wave == 0) poll_head(ni.from_slot);   // (the answer is read behind the barriers below)
                if (kt == ke - 3) switch_tile();                 // (only reached when the item has a successor)
                load_frags(sq);                                  // slot 2kt : LOAD
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                compute_dma(pf, kt + 3 + kshift, sq + 3);        // slot 2kt+1 : MFMA + DMA issue of K-tile kt+3
                __builtin_amdgcn_sched_barrier(0);
                if (kt < kb + 2) wait_vm_rt(8 + ex); else wait_vmcnt<8>();   // tile kt+1 landed; kt+2, kt+3 still in flight
                __builtin_amdgcn_s_barrier();
            }
#pragma nounroll
            for (int kt = k_dma_end; kt < ke; ++kt) {
                load_frags(seq0 + kt - kb);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                compute_plain();
                __builtin_amdgcn_sched_barrier(0);
                wait_tiles_in_flight(min(2, ke - 2 - kt));       // tile kt+1 landed; nothing is issued any more
                __builtin_amdgcn_s_barrier();
            }
            __builtin_amdgcn_s_barrier();  // slot 2nt: group 1 finishes its last MFMA phase
        } else {
            add_partial();
            __builtin_amdgcn_s_barrier();  // slot 0: idle
#pragma nounroll
            for (int kt = kb; kt < k_dma_end; ++kt) {
                const int sq = seq0 + kt - kb;
                if (kt == ke - 3) switch_tile();
                load_frags(sq);                                  // slot 2kt+1 : LOAD
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                if (kt < kb + 2) wait_vm_rt(4 + ex); else wait_vmcnt<4>();   // tile kt+1 landed; issued so far: .. kt+2
                __builtin_amdgcn_s_barrier();
                compute_dma(pf, kt + 3 + kshift, sq + 3);        // slot 2kt+2 : MFMA + DMA issue of K-tile kt+3
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
            }
            if (more) wait_vmcnt<8>();     // (group 0's last wait, made here too: the next item's first K-tile has landed on every wave)
#pragma nounroll
            for (int kt = k_dma_end; kt < ke; ++kt) {
                load_frags(seq0 + kt - kb);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                wait_tiles_in_flight(min(1, ke - 2 - kt));
                __builtin_amdgcn_s_barrier();
                compute_plain();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
            }
        }
        SK_STAMP(2);
        const int seq_next = seq0 + n;
        extra = 0;

        if (ci.to_slot >= 0) {
            // head part of a split tile: hand the accumulators to the tail part's workgroup as they are (no epilogue)
            const __amdgpu_buffer_rsrc_t rs = sk_slot_rsrc(g.sk_ws, ci.to_slot);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const f32x4 v = {acc[i][j][4 * c], acc[i][j][4 * c + 1], acc[i][j][4 * c + 2], acc[i][j][4 * c + 3]};
                        sk_store16(rs, (i * 2 + j) * 4 + c, tid, v);
                    }
            sk_publish(g.sk_flags, ci.to_slot, g.sk_epoch, tid);     // (drains the queue: extra stays 0)
        } else {
            // ---- epilogue: the accumulators through the free ring stage (the one the next item's first three K-tiles do not use), a
            // [32 rows][32 columns] f32 block per wave at a time (4 KiB, rows of eight 16-byte chunks, chunk ^= row & 7); bias and group
            // addends from LDS; the stores are left in flight.
            const bool vec_ok = CONV || ((g.ldc & 7) == 0);
            const bool interior = vec_ok && em0 + 256 <= g.M && en0 + 256 <= g.N;
            const int fr_e = lane & 31, fh_e = lane >> 5;
            char* est = smem + ((seq_next + 3) & (NSTAGE - 1)) * STAGE + wave * 4096;
            SK_STAMP(3);
#pragma unroll
            for (int nn = 0; nn < 2; ++nn) {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        f32x4 v = {acc[mi][nn][4 * c], acc[mi][nn][4 * c + 1], acc[mi][nn][4 * c + 2], acc[mi][nn][4 * c + 3]};
                        *reinterpret_cast<f32x4*>(est + fr_e * 128 + (((2 * c + fh_e) ^ (fr_e & 7)) << 4)) = v;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own LDS writes landed (same wave reads them back)
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const int rl = (lane >> 2) + 16 * it, ch = (lane & 3) * 2;
                        const int rt = grp * 128 + mi * 32 + rl;                    // row inside the tile
                        const int ncol = q * 64 + nn * 32 + (lane & 3) * 8;         // first of the lane's 8 columns inside the tile
                        const int m = em0 + rt, nc = en0 + ncol;
                        float v[8];
                        const f32x4 lo = *reinterpret_cast<const f32x4*>(est + rl * 128 + ((ch ^ (rl & 7)) << 4));
                        const f32x4 hi = *reinterpret_cast<const f32x4*>(est + rl * 128 + (((ch + 1) ^ (rl & 7)) << 4));
                        const f32x4 b0 = *reinterpret_cast<const f32x4*>(xb + ncol), b1 = *reinterpret_cast<const f32x4*>(xb + ncol + 4);
                        float ga0 = 0.f, ga1 = 0.f;
                        int gsplit = 8;
                        if (g.gadd) {
                            const int col = min(nc, g.N - 1) + g.gcol0, gi = col / g.ggroup;
                            gsplit = min(8, g.ggroup - (col - gi * g.ggroup));
                            ga0 = xg[min(gi - g0, 7) * 256 + rt];
                            ga1 = xg[min(gi - g0 + 1, 7) * 256 + rt];
                        }
                        if (!interior && (m >= g.M || nc >= g.N)) continue;
                        v[0] = lo.x + b0.x; v[1] = lo.y + b0.y; v[2] = lo.z + b0.z; v[3] = lo.w + b0.w;
                        v[4] = hi.x + b1.x; v[5] = hi.y + b1.y; v[6] = hi.z + b1.z; v[7] = hi.w + b1.w;
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            float t = v[k];
                            if (g.gadd) t += k < gsplit ? ga0 : ga1;
                            if (g.act == SGG_ACT_RELU) t = fmaxf(t, 0.f);
                            v[k] = t;
                        }
                        const long off = out_offset<CONV>(g, m, nc);
                        if (interior || (vec_ok && nc + 8 <= g.N)) {
#ifdef SGG_SK_NOSTORE
                            if (v[0] != 123456.f) continue;
#endif
                            if (g.out_dt == SGG_BF16) store8(reinterpret_cast<bf16_t*>(g.C) + off, v);
                            else if (g.out_dt == SGG_F16) store8(reinterpret_cast<f16_t*>(g.C) + off, v);
                            else store8(reinterpret_cast<float*>(g.C) + off, v);
                        } else {
                            const int nv = min(8, g.N - nc);
                            for (int k = 0; k < nv; ++k) {
                                if (g.out_dt == SGG_BF16) reinterpret_cast<bf16_t*>(g.C)[off + k] = f32_to_bf16(v[k]);
                                else if (g.out_dt == SGG_F16) reinterpret_cast<f16_t*>(g.C)[off + k] = (f16_t)v[k];
                                else reinterpret_cast<float*>(g.C)[off + k] = v[k];
                            }
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the block is rewritten by the next one
                }
            }
            if (interior) {
                extra = g.out_dt == SGG_F32 ? 32 : 16;            // store instructions this wave left in the queue: 16 rows x 16 / 32 bytes per lane
            } else {
                wait_vmcnt<0>();                                  // an edge tile's store count depends on its shape: drain (extra stays 0)
            }
            SK_STAMP(4);
        }
        if (more) {
            ci = ni;
            seq0 = seq_next;
        }
        // (the barrier at the head of the next item: every wave is done with its staging block before the next item's fourth K-tile lands there)
    }
}

template <int DT, bool CONV>
int launch256_sk(const GemmArgs& g, int grid, hipStream_t s) {
    auto k = mfma_pingpong_sk_kernel<DT, CONV>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_SK) != hipSuccess)
            return SGG_ERR_LAUNCH;
        attr_done = true;
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), SMEM_SK, s, g);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

std::mutex sk_mu;
// (device, stream) -> its workspace (launches of one stream are ordered: one set of slots suffices).  The device is part of the key: the
// default stream is the same handle (0) on every device of a process.
std::map<std::pair<int, void*>, SkWorkspace> sk_reg;
std::pair<int, void*> sk_key(void* stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    return std::make_pair(dev, stream);
}
int sk_mode_value = -1;
int sk_mode() {
    if (sk_mode_value < 0) {
        const char* env = getenv("SGG_STREAMK");
        // default 0: measured (tools/streamk_bench.py, profiles/r04_streamk.txt) the persistent form is bit-identical but not faster yet
        sk_mode_value = env && env[0] >= '0' && env[0] <= '3' ? env[0] - '0' : 0;
    }
    return sk_mode_value;
}

}  // namespace

SkWorkspace* sgg_sk_workspace_of(void* stream) {
    std::lock_guard<std::mutex> lk(sk_mu);
    auto it = sk_reg.find(sk_key(stream));
    return it == sk_reg.end() ? nullptr : &it->second;
}

// The workspace of `stream` with its launch epoch advanced (under the registry's lock), by value; false: no workspace
static bool sk_next_epoch(void* stream, SkWorkspace& out) {
    std::lock_guard<std::mutex> lk(sk_mu);
    auto it = sk_reg.find(sk_key(stream));
    if (it == sk_reg.end()) return false;
    SkWorkspace& w = it->second;
    w.epoch = (w.epoch + 1) & 0x7fffffffu;
    if (w.epoch == 0) w.epoch = 1;
    out = w;
    return true;
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
        sk_reg.erase(sk_key(stream));
        return SGG_OK;
    }
    if (((uintptr_t)ws & 255) || bytes < sgg_streamk_workspace_bytes()) return SGG_ERR_ARG;
    SkWorkspace w;
    w.flags = reinterpret_cast<unsigned*>(ws);                    // SK_MAX_GRID words first (the part the caller zeroes) ...
    w.slots = reinterpret_cast<char*>(ws) + SK_MAX_GRID * 4;      // ... then the slots
    w.epoch = 0;
    w.max_grid = SK_MAX_GRID;
    sk_reg[sk_key(stream)] = w;
    return SGG_OK;
}

// 0: never stream-K (the default); 1: wherever it applies; 2: where a plain launch would leave >= 4 % of its tile slots empty; 3 (tests):
// as 1, and no consumer finds its head part published.  The environment variable SGG_STREAMK = 0..3 sets the initial value; any other
// argument only queries.  Returns the previous mode.
extern "C" int sgg_streamk_mode(int mode) {
    const int prev = sk_mode();
    if (mode >= 0 && mode <= 3) sk_mode_value = mode;
    return prev;
}

// 1 when a ping-pong launch of `tiles` tiles with `nt` (64-byte) K-tiles on `stream` would take the stream-K form (callers that
// otherwise cut a nearly empty last round off into another launch ask first)
int sgg_pingpong_streamk(long tiles, int nt, void* stream, int* dp_rounds, int* sk_tiles) {
    const int mode = sk_mode();
    if (mode == 0) return 0;
    const int G = sgg_sk_grid();
    if (G <= 0 || !sgg_sk_workspace_of(stream)) return 0;
    // The hand-over flags carry the launch's EPOCH, a host counter that travels in the kernel arguments: a launch captured into a
    // hipGraph would replay with the epoch it was captured with, find the flags of its previous replay "published" and read slots the
    // current replay is still writing.  A capturing stream therefore always gets the plain launch (same bits, one kernel per round).
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return 0;
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

// The stream-K launch of a ping-pong GEMM / implicit-GEMM convolution, or 1 when this launch does not take that form (the caller then
// launches the plain kernel).  g.nt / g.nt1 in 64-byte K-tiles.
int sgg_launch_pingpong_sk(const GemmArgs& g_in, int dt, bool conv, hipStream_t s) {
    GemmArgs g = g_in;
    const long tiles = (long)((g.M + 255) / 256) * ((g.N + 255) / 256);
    int dp = 0, sk = 0;
    // the stream-K form's epilogue carries bias, activation and a group addend whose groups span >= 40 columns (256 columns touch <= 8)
    const bool epilogue_ok = !g.pscale && !g.pshift && !g.add_rows && (!g.gadd || g.ggroup >= 40);
    if (g.x3t || g.out_dt == SGG_PAIR16) return 1;       // pair operands / outputs (x3 mode): the plain ping-pong kernel walks the plane segments
    if (g.m_base != 0 || !epilogue_ok || !sgg_pingpong_streamk(tiles, g.nt, (void*)s, &dp, &sk)) return 1;
    SkWorkspace w;
    if (!sk_next_epoch((void*)s, w)) return 1;
    const int G = sgg_sk_grid();
    g.sk_ws = w.slots; g.sk_flags = w.flags; g.sk_epoch = w.epoch | (sk_mode() == 3 ? 0x80000000u : 0u); g.sk_dp_rounds = dp; g.sk_tiles = sk;
    if (dt == SGG_BF16) return conv ? launch256_sk<SGG_BF16, true>(g, G, s) : launch256_sk<SGG_BF16, false>(g, G, s);
    if (dt == SGG_F16) return conv ? launch256_sk<SGG_F16, true>(g, G, s) : launch256_sk<SGG_F16, false>(g, G, s);
    return conv ? launch256_sk<SGG_F32, true>(g, G, s) : launch256_sk<SGG_F32, false>(g, G, s);
}
