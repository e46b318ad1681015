// 3x3 convolution, LDS-resident input patch + ping-pong MFMA schedule (gfx950, 16-bit element types): the VGG layers with >= 128 output
// channels on maps of 64 cells and more.
//
// Two kernels came before this one and each is bound by what the other does well:
//   * conv_spatial.hip stages the 18x18 input patch of a 16x16 output tile ONCE per channel slab and reads the nine taps at shifted
//     LDS rows (a third of the global->LDS traffic of an implicit GEMM), but its eight waves run in lock step: after every barrier
//     they all read fragments, then they all issue MFMAs -- ablations: 222 us as built, 95 us without the MFMAs, ~143 us of MFMA work;
//     the two hardly overlap.  64 x 64 outputs per wave = one ds_read_b128 per MFMA keeps the LDS pipe as busy as the matrix pipe.
//   * gemm256.hip runs two groups of waves one phase apart (while one group reads fragments the other feeds the matrix pipe, one wave
//     of each group per SIMD) with 128 x 64 outputs per wave (0.75 reads per MFMA) and reaches 1.3 PFLOP/s on fc6 -- but as an implicit
//     GEMM it re-stages every input pixel once per tap and the LDS-DMA issue rate bounds it (~0.95-1.0 PFLOP/s on conv3 / conv4).
// This kernel is the second schedule on the first one's data layout.  A workgroup owns a 16x16 output tile x CN = 64 NQ output channels
// (NQ = 4: eight waves; NQ = 2: four waves, two workgroups per CU).  Wave (grp, q): image rows 8 grp .. 8 grp + 7 of the tile (128
// pixels = four 32-pixel MFMA blocks of 2 rows x 16 columns) x channels 64 q .. 64 q + 63: 4 x 2 MFMA 32x32 tiles, 128 accumulators.
// K advances in K-tiles = (32-channel slab c [64 B per LDS row], tap t), t fastest: 2 MFMA k-steps = 16 MFMAs and 12 ds_read_b128 per
// wave.  Per K-tile only the weight slab [CN][64 B] is streamed (two 1-KiB LDS-DMA instructions per wave, ring of four stages, issued
// three K-tiles ahead from inside the MFMA phase); the patch of slab c+1 (21 instructions per workgroup) rides along with the
// weight pieces of a few taps of slab c into the other of two patch buffers.  Every wave issues the same number of DMA instructions
// per K-tile position, so the counted vmcnt waits are compile-time constants; K-tiles past the end re-load the last one (no peeled
// tail).  Slot schedule and barriers as in gemm256.hip:
//
//   slot      0        1        2        3        4   ...
//   group 0   LOAD 0   MFMA 0   LOAD 1   MFMA 1   LOAD 2
//   group 1   -        LOAD 0   MFMA 0   LOAD 1   MFMA 1
//
// LDS rows are 64 B with the 16-byte slot swizzle phys = slot ^ ((row >> 2) & 3) on the DMA source address and on the read; a 32-pixel
// fragment block gives each ds_read_b128 hardware lane group one whole image row (16 consecutive patch rows: every (row & 3, key) pair
// once -- conflict-free for any tile position and tap).  Activations are zero-bordered NHWC planes; tiles that overhang the right /
// bottom edge clamp their loads inside the plane and mask their stores.  Epilogue (bias, ReLU, optional fused 2x2 max pool) through a
// per-wave LDS staging tile as in conv_spatial.hip.
#include <cstdlib>
#include <type_traits>
#include "gemm_args.h"

namespace {

constexpr int PROW = 64;                                                        // bytes of K per LDS row (32 channels)
constexpr int NSTG = 4;
// The 256-pixel output tile is TH rows x TW columns (16 x 16 or 8 x 32): the launcher picks the shape that wastes fewer MFMAs on pixels
// outside the map and fills the last round of workgroups better (152 x 152 maps, batch 8: 800 tiles of 16 x 16 = 3.1 rounds of 256 CUs
// with 15 % of the pixels outside; 760 tiles of 8 x 32 = 2.97 rounds with 5 % outside).
template <int TW, int NPG>
struct TileGeom {
    static constexpr int TH = 128 * NPG / TW, PW = TW + 2, ROWS = (TH + 2) * PW;   // patch: (TH + 2) x (TW + 2) rows of 64 B
    static constexpr int PIECES = (ROWS + 15) / 16;                             // DMA instructions of 16 rows (21 / 22; 39 for 512 pixels)
    static constexpr int BYTES = PIECES * 1024;                                 // per patch buffer
    static constexpr int CB = TW / 16;                                          // 32-pixel blocks (2 rows x 16 columns) side by side
};

struct ConvPPArgs {
    const char* in;      // [B, H+2, W+2, Cin]
    const char* w;       // [Cout][9][Cin]
    const float* bias;
    char* out;           // [B, H+2p, W+2p, Cout]   (pool: [B, H/2+2p, W/2+2p, Cout])
    int B, H, W, Cin, Cout, out_pad, tiles_x, tiles_y, pool;
    // X3 form (the x3 mode on PAIR planes, include/sgg_hip.h SGG_PAIR16): `in` is [B, H+2, W+2, 2 Cpl] with pixel = [hi (Cpl) | lo (Cpl)], `w` is
    // [Cout][9][3 Cpl] with tap = [hi | lo | hi] (sgg_split3's weight form), Cin = 3 Cpl VIRTUAL channels: slab v of 32 reads plane channels
    // (v mod Cpl/32) of the hi plane for v < 2 Cpl/32 and of the lo plane after -- hi.hi + hi.lo + lo.hi in the accumulator; `out` is a pair
    // plane [.., 2 Cout] written from the fp32 accumulator (bias, ReLU, optional 2x2 max, THEN the split)
    int Cpl;
};

// MFMA column (lane & 31) -> pixel of a 2 x 16 block: each ds_read_b128 lane group ({0-3,12-15,20-27} / {4-11,16-19,28-31}) gets one image row
__device__ __forceinline__ int pp_py(int c) { return ((c >= 4 && c < 12) || (c >= 16 && c < 20) || c >= 28) ? 1 : 0; }
__device__ __forceinline__ int pp_px(int c) { return c < 4 ? c : c < 12 ? c - 4 : c < 20 ? c - 8 : c < 28 ? c - 12 : c - 16; }

template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N <= 63, "vmcnt range");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_vm_n(int n) {          // folds to one instruction when n is a constant after unrolling
    switch (n) {
        case 1: wait_vm<1>(); break;
        case 2: wait_vm<2>(); break;
        case 3: wait_vm<3>(); break;
        case 4: wait_vm<4>(); break;
        case 5: wait_vm<5>(); break;
        case 6: wait_vm<6>(); break;
        default: wait_vm<0>(); break;
    }
}

// NQ channel slabs of 64 x NPG pixel groups of 128 = NW waves: (4, 2) 256 px x 256 ch; (2, 2) 256 px x 128 ch, four waves, two workgroups
// per CU; (2, 4) 512 px x 128 ch (the 128-channel layers with eight waves: half the weight traffic per MFMA of the four-wave form).
template <int DT, int NQ, int NPG, int TW, bool X3 = false>
__global__ __launch_bounds__(64 * NQ * NPG, 2) void conv3x3_pp_kernel(const ConvPPArgs g) {     // two waves per SIMD (four-wave form: from two workgroups)
    static_assert(DT == SGG_BF16 || DT == SGG_F16, "16-bit element types");
    using TG = TileGeom<TW, NPG>;
    constexpr int PT_PW = TG::PW, PT_ROWS = TG::ROWS, PATCH_PIECES = TG::PIECES, PATCH_B = TG::BYTES;
    constexpr int NW = NQ * NPG, CN = 64 * NQ;
    constexpr int WST = CN * PROW;                            // bytes per weight stage
    constexpr int WI = CN / 16 / NW;                          // weight-slab DMA instructions per wave and K-tile (2 or 1)
    static_assert(WI * 16 * NW == CN, "weight rows divide over the waves");
    constexpr int PTAPS = (PATCH_PIECES + NW - 1) / NW;       // taps 3 .. 3 + PTAPS - 1 of a slab carry the next slab's patch (one piece per wave)
    static_assert(3 + PTAPS <= 9, "patch pieces fit the taps of one slab");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const pbuf = smem;                                  // 2 x PATCH_B
    char* const wring = smem + 2 * PATCH_B;                   // NSTG x WST

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = wave / NQ, q = wave % NQ;                 // pixel group (128 pixels), channel slab
    const int grp = wave / (NW / 2);                         // phase group: waves w and w + 4 share a SIMD and must be in opposite phases

    // block -> (image, tile, channel block); channel blocks of one tile are adjacent (they share the input patch in L2)
    const int ncb = g.Cout / CN;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int cb = bid % ncb;
    bid /= ncb;
    const int tx = bid % g.tiles_x;
    bid /= g.tiles_x;
    const int ty = bid % g.tiles_y;
    const int b = bid / g.tiles_y;
    const int y0 = ty * TG::TH, x0 = tx * TW, n0 = cb * CN;
    const int nch = g.Cin / 32;

    // ---- DMA duty.  Weights: wave w stages rows [32 w, 32 w + 32) of the slab, two instructions of 16 rows.  Patch: pieces wave, wave + NW, ..
    unsigned wsrc[WI];
#pragma unroll
    for (int j = 0; j < WI; ++j) {
        const int r = wave * (16 * WI) + j * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ ((r >> 2) & 3);
        wsrc[j] = (unsigned)((long)(n0 + r) * 9 * g.Cin * 2 + chunk * 16);
    }
    unsigned psrc[PTAPS];
#pragma unroll
    for (int j = 0; j < PTAPS; ++j) {
        const int piece = min(wave + j * NW, PATCH_PIECES - 1);           // (surplus slots repeat the last piece: same bytes, same place)
        const int r = min(piece * 16 + (lane >> 2), PT_ROWS - 1);
        const int py = r / PT_PW, px = r - py * PT_PW;
        const int gy = min(y0 + py, g.H + 1), gx = min(x0 + px, g.W + 1);
        const int chunk = (lane & 3) ^ ((r >> 2) & 3);
        psrc[j] = (unsigned)((((long)b * (g.H + 2) + gy) * (g.W + 2) + gx) * (X3 ? 2 * g.Cpl : g.Cin) * 2 + chunk * 16);
    }
    // weight pieces of K-tile kt (clamped to the last one) into ring stage kt & 3; with them, at taps 3 .. 3+PTAPS-1, one patch piece of the NEXT slab
    auto w_base = [&](int slab, int tap) -> const char* {                 // (slab, tap) past the end: the last K-tile again
        const bool past = slab >= nch;
        const int c = past ? nch - 1 : slab, t = past ? 8 : tap;
        return uniform_ptr(g.w + ((long)t * g.Cin + c * 32) * 2);
    };
    auto issue_w = [&](const char* ub, int kt, int j) {
        glds16_su(ub, wsrc[j], wring + (kt & (NSTG - 1)) * WST + (wave * (16 * WI) + j * 16) * PROW);
    };
    auto issue_p = [&](int slab, int j) {                                  // piece j of this wave, patch of `slab` (clamped) into buffer slab & 1
        const int sl = min(slab, nch - 1);
        long poff = (long)sl * PROW;
        if constexpr (X3) {
            const int nc = g.Cpl / 32;                                      // slabs per plane
            const int seg = (sl >= nc) + (sl >= 2 * nc);
            poff = (long)(sl - seg * nc) * PROW + (seg == 2 ? (long)g.Cpl * 2 : 0);
        }
        const char* ub = uniform_ptr(g.in + poff);
        glds16_su(ub, psrc[j], pbuf + (slab & 1) * PATCH_B + min(wave + j * NW, PATCH_PIECES - 1) * 1024);
    };

    // ---- fragment addressing
    const int fr = lane & 31, fh = lane >> 5;
    int prow0[4];          // patch row (tap 0,0) of this lane's pixel in the four 32-pixel blocks of the wave
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int blk = pg * 4 + i;                             // block -> (row pair, 16-column part)
        prow0[i] = ((blk / TG::CB) * 2 + pp_py(fr)) * PT_PW + (blk % TG::CB) * 16 + pp_px(fr);
    }
    int boff[2], bkey[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = q * 64 + i * 32 + fr;
        boff[i] = r * PROW;
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

    auto load_frags = [&](int kt, int slab, int tap) {
        const char* ws = wring + (kt & (NSTG - 1)) * WST;
        const char* pb = pbuf + (slab & 1) * PATCH_B;
        const int ky = tap / 3, kx = tap - ky * 3;
        const int dp = ky * PT_PW + kx;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int s = 0; s < 2; ++s) bf[i][s] = *reinterpret_cast<const u32x4*>(ws + boff[i] + (((2 * s + fh) ^ bkey[i]) << 4));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int pr = prow0[i];
            asm volatile("" : "+v"(pr));          // keeps the nine taps' address arithmetic inside the loop (hoisted it costs 72 registers)
            const int r = pr + dp;
            const int x = (fh ^ ((r >> 2) & 3)) << 4;                      // physical 16-byte slot of logical slot fh; slot 2 + fh is x ^ 32
            af[i][0] = *reinterpret_cast<const u32x4*>(pb + r * PROW + x);
            af[i][1] = *reinterpret_cast<const u32x4*>(pb + r * PROW + (x ^ 32));
        }
    };

#define PP_MFMA_PAIR(s, mi) \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = mfma_32x32x16<DT>(bf[ni][s], af[mi][s], acc[mi][ni]);
    // 16 MFMAs with the DMA pieces of K-tile kt + 3 between them (its tap position decides whether a patch piece rides along)
    auto compute_dma = [&](int kt, int c, int tap) {                       // tap: a constant after unrolling
        const int tap3 = (tap + 3) % 9, c3 = c + (tap + 3) / 9;               // position of K-tile kt + 3
        const char* ub = w_base(c3, tap3);
        const bool with_patch = tap3 >= 3 && tap3 < 3 + PTAPS;
        const int slab3 = c3 + 1;
        __builtin_amdgcn_s_setprio(1);
        PP_MFMA_PAIR(0, 0)
        issue_w(ub, kt + 3, 0);
        __builtin_amdgcn_sched_barrier(0);
        PP_MFMA_PAIR(0, 1)
        if constexpr (WI > 1) issue_w(ub, kt + 3, 1);
        __builtin_amdgcn_sched_barrier(0);
        PP_MFMA_PAIR(0, 2)
        if (with_patch) {
#pragma unroll
            for (int j = 0; j < PTAPS; ++j)
                if (tap3 - 3 == j) issue_p(slab3, j);
        }
        __builtin_amdgcn_sched_barrier(0);
        PP_MFMA_PAIR(0, 3)
        PP_MFMA_PAIR(1, 0)
        PP_MFMA_PAIR(1, 1)
        PP_MFMA_PAIR(1, 2)
        PP_MFMA_PAIR(1, 3)
        __builtin_amdgcn_s_setprio(0);
    };
    // DMA instructions a wave issues with K-tile position `tap`
    auto n_at = [](int tap) { tap %= 9; return WI + ((tap >= 3 && tap < 3 + PTAPS) ? 1 : 0); };

    // ---- prologue: the patch of slab 0, K-tiles 0..2 in flight; patch and tile 0 landed and visible
#pragma unroll
    for (int j = 0; j < PTAPS; ++j) issue_p(0, j);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const char* ub = w_base(0, t);
        issue_w(ub, t, 0);
        if constexpr (WI > 1) issue_w(ub, t, 1);
    }
    wait_vm<2 * WI>();
    __builtin_amdgcn_s_barrier();

    if (grp == 0) {
        for (int c = 0; c < nch; ++c) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kt = c * 9 + tap;
                load_frags(kt, c, tap);                          // slot 2kt : LOAD
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                compute_dma(kt, c, tap);                         // slot 2kt+1 : MFMA + DMA issue of K-tile kt+3
                __builtin_amdgcn_sched_barrier(0);
                wait_vm_n(n_at(tap + 2) + n_at(tap + 3));        // K-tile kt+1 landed; kt+2, kt+3 still in flight
                __builtin_amdgcn_s_barrier();
            }
        }
        __builtin_amdgcn_s_barrier();  // slot 2nt: group 1 finishes its last MFMA phase
    } else {
        __builtin_amdgcn_s_barrier();  // slot 0: idle
        for (int c = 0; c < nch; ++c) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kt = c * 9 + tap;
                load_frags(kt, c, tap);                          // slot 2kt+1 : LOAD
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                wait_vm_n(n_at(tap + 2));                        // K-tile kt+1 landed; issued so far: .. kt+2
                __builtin_amdgcn_s_barrier();
                compute_dma(kt, c, tap);                         // slot 2kt+2 : MFMA + DMA issue of K-tile kt+3
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
            }
        }
    }
#undef PP_MFMA_PAIR

    // ---- epilogue through LDS: per wave a [32 px][64 ch] f32 staging tile (row stride 272 B), one 32-pixel block at a time
    constexpr int ESTRIDE = 272;
    char* est = smem + wave * (32 * ESTRIDE);
    float bias8[8];   // the lane's 8 output channels are the same in every store below
    load8(g.bias + n0 + q * 64 + (lane & 7) * 8, bias8);
    wait_vm<0>();                                                // the re-loads past the last K-tile are still landing in the ring
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    using TO = typename std::conditional<DT == SGG_BF16, bf16_t, f16_t>::type;
    TO* const outp = reinterpret_cast<TO*>(g.out);
    const int cl = (lane & 7) * 8, n = n0 + q * 64 + cl, op = g.out_pad;
    const int opx = X3 ? 2 * g.Cout : g.Cout;                   // elements per output pixel
    auto put8 = [&](long pixel, float (&v)[8]) {                // 8 channels of one output pixel (pair plane: hi, and lo Cout further)
        if constexpr (X3) {
            float lo[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float h = round_as<f16_t>(v[k]);
                lo[k] = v[k] - h;
                v[k] = h;
            }
            store8(outp + pixel * opx + n, v);
            store8(outp + pixel * opx + g.Cout + n, lo);
        } else {
            store8(outp + pixel * opx + n, v);
        }
    };
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                f32x4 v = {acc[mi][ni][4 * c4], acc[mi][ni][4 * c4 + 1], acc[mi][ni][4 * c4 + 2], acc[mi][ni][4 * c4 + 3]};
                *reinterpret_cast<f32x4*>(est + fr * ESTRIDE + (ni * 32 + 8 * c4 + 4 * fh) * 4) = v;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own LDS writes landed (same wave reads them back)
        const int blk = pg * 4 + mi;
        const int yb = y0 + (blk / TG::CB) * 2, xb = x0 + (blk % TG::CB) * 16;     // first image row / column of this 2 x 16 block
        if (g.pool) {
            // fused MaxPool2d(2): the block is 8 complete 2x2 windows; max first, then bias + ReLU (they commute with max)
            const int wb = lane >> 3;                          // window = columns 2 wb, 2 wb + 1 of both rows
            const int yo = yb >> 1, xo = (xb >> 1) + wb;
            if (2 * yo < g.H && 2 * xo < g.W) {
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = -3.0e38f;
#pragma unroll
                for (int p4 = 0; p4 < 4; ++p4) {
                    const int px = 2 * wb + (p4 & 1), py = p4 >> 1;
                    const int col = py == 0 ? (px < 4 ? px : px < 8 ? px + 8 : px + 12) : (px < 8 ? px + 4 : px < 12 ? px + 8 : px + 16);   // inverse of (pp_py, pp_px)
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(est + col * ESTRIDE + cl * 4);
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(est + col * ESTRIDE + cl * 4 + 16);
                    v[0] = fmaxf(v[0], lo.x); v[1] = fmaxf(v[1], lo.y); v[2] = fmaxf(v[2], lo.z); v[3] = fmaxf(v[3], lo.w);
                    v[4] = fmaxf(v[4], hi.x); v[5] = fmaxf(v[5], hi.y); v[6] = fmaxf(v[6], hi.z); v[7] = fmaxf(v[7], hi.w);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k] + bias8[k], 0.f);
                const int Ho = g.H >> 1, Wo = g.W >> 1;
                put8(((long)b * (Ho + 2 * op) + yo + op) * (Wo + 2 * op) + xo + op, v);
            }
        } else {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int pl = (lane >> 3) + 8 * it;           // pixel inside the 32-block
                const int y = yb + pp_py(pl), x = xb + pp_px(pl);
                if (y >= g.H || x >= g.W) continue;
                float v[8];
                const f32x4 lo = *reinterpret_cast<const f32x4*>(est + pl * ESTRIDE + cl * 4);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(est + pl * ESTRIDE + cl * 4 + 16);
                v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k] + bias8[k], 0.f);
                put8(((long)b * (g.H + 2 * op) + y + op) * (g.W + 2 * op) + x + op, v);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the staging tile is rewritten by the next block
    }
}

template <int DT, int NQ, int NPG, int TW, bool X3 = false>
int launch_pp(ConvPPArgs g, hipStream_t s) {
    constexpr int NW = NQ * NPG, CN = 64 * NQ;
    using TG = TileGeom<TW, NPG>;
    constexpr int smem_main = 2 * TG::BYTES + NSTG * CN * PROW, smem_epi = NW * 32 * 272;
    g.tiles_x = (g.W + TW - 1) / TW;
    g.tiles_y = (g.H + TG::TH - 1) / TG::TH;
    constexpr int smem = smem_main > smem_epi ? smem_main : smem_epi;
    static_assert(smem <= 160 * 1024, "fits the CU's LDS");
    auto k = conv3x3_pp_kernel<DT, NQ, NPG, TW, X3>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return SGG_ERR_LAUNCH;
        attr_done = true;
    }
    const int blocks = g.B * g.tiles_y * g.tiles_x * (g.Cout / CN);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * NW), smem, s, g);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

template <int DT, bool X3 = false>
int launch_pp_dt(const ConvPPArgs& g, int nq, int npg, int tw, hipStream_t s) {
    if (nq == 4) return tw == 32 ? launch_pp<DT, 4, 2, 32, X3>(g, s) : launch_pp<DT, 4, 2, 16, X3>(g, s);
    if (npg == 4) return tw == 32 ? launch_pp<DT, 2, 4, 32, X3>(g, s) : launch_pp<DT, 2, 4, 16, X3>(g, s);
    return tw == 32 ? launch_pp<DT, 2, 2, 32, X3>(g, s) : launch_pp<DT, 2, 2, 16, X3>(g, s);
}

// share of the launch's MFMA work that lands on map pixels, times how full its rounds of workgroups are
double pp_efficiency(int B, int H, int W, int Cout, int nq, int npg, int tw) {
    const int th = 128 * npg / tw, ty = (H + th - 1) / th, tx = (W + tw - 1) / tw;
    const long wgs = (long)B * ty * tx * (Cout / (64 * nq));
    const long slots = 256L * (nq * npg == 4 ? 2 : 1);           // workgroups the chip holds at once (four-wave form: two per CU)
    const long rounds = (wgs + slots - 1) / slots;
    return ((double)H * W / ((double)ty * th * tx * tw)) * ((double)wgs / (double)(rounds * slots));
}

}  // namespace

static int pick_pp_form(int B, int H, int W, int Cout, int form, int tw, int& bq, int& bg, int& bt) {
    double best = -1.0;
    bq = bg = bt = 0;
    const int cand[6][3] = {{4, 2, 16}, {4, 2, 32}, {2, 4, 16}, {2, 4, 32}, {2, 2, 16}, {2, 2, 32}};
    for (const auto& c : cand) {
        if (c[0] == 4 && Cout % 256) continue;
        if (form && form != c[0] * 10 + c[1]) continue;
        if (tw && tw != c[2]) continue;
        const double e = pp_efficiency(B, H, W, Cout, c[0], c[1], c[2]);
        if (e > best + 0.02) best = e, bq = c[0], bg = c[1], bt = c[2];
    }
    return bq;
}

// X3 form (pair planes in and out, weights [Cout][9][3 Cpl] = [hi | lo | hi] per tap): SGG_OK, or 1 if the shape is not handled
int sgg_launch_conv_pp_x3(const void* in, const void* w3, const float* bias, void* out, int out_pad, int B, int H, int W, int Cpl, int Cout,
                          int pool, hipStream_t s) {
    if (Cpl % 32 || Cout % 128) return 1;
    ConvPPArgs g{};
    g.in = (const char*)in; g.w = (const char*)w3; g.bias = bias; g.out = (char*)out;
    g.B = B; g.H = H; g.W = W; g.Cin = 3 * Cpl; g.Cpl = Cpl; g.Cout = Cout; g.out_pad = out_pad; g.pool = pool;
    int bq, bg, bt;
    if (!pick_pp_form(B, H, W, Cout, 0, 0, bq, bg, bt)) return 1;
    return launch_pp_dt<SGG_F16, true>(g, bq, bg, bt, s);
}

// returns SGG_OK, or 1 if the shape is not handled here (the caller falls through to the other convolution kernels).
// form: 0 = choose; 42 = 256 px x 256 ch; 22 = 256 px x 128 ch (four waves); 24 = 512 px x 128 ch.  tw: 16 / 32 = tile width, 0 = choose.
int sgg_launch_conv_pp(const void* in, const void* w, const float* bias, void* out, int out_pad, int B, int H, int W, int Cin, int Cout,
                       int dt, int pool, int form, int tw, hipStream_t s) {
    if ((dt != SGG_BF16 && dt != SGG_F16) || Cin % 32 || Cout % 128) return 1;
    ConvPPArgs g{};
    g.in = (const char*)in; g.w = (const char*)w; g.bias = bias; g.out = (char*)out;
    g.B = B; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.out_pad = out_pad; g.pool = pool;
    if (form == 42 && Cout % 256) return 1;
    double best = -1.0;
    int bq = 0, bg = 0, bt = 0;
    // candidates in order of preference (eight-wave forms first); a later one must be clearly better to win
    const int cand[6][3] = {{4, 2, 16}, {4, 2, 32}, {2, 4, 16}, {2, 4, 32}, {2, 2, 16}, {2, 2, 32}};
    for (const auto& c : cand) {
        if (c[0] == 4 && Cout % 256) continue;
        if (form && form != c[0] * 10 + c[1]) continue;
        if (tw && tw != c[2]) continue;
        const double e = pp_efficiency(B, H, W, Cout, c[0], c[1], c[2]);
        if (e > best + 0.02) best = e, bq = c[0], bg = c[1], bt = c[2];
    }
    if (!bq) return 1;
    return dt == SGG_BF16 ? launch_pp_dt<SGG_BF16>(g, bq, bg, bt, s) : launch_pp_dt<SGG_F16>(g, bq, bg, bt, s);
}
