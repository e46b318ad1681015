// 3x3 convolution with an LDS-resident input patch (gfx950): the wide-spatial VGG layers (conv1_2 .. conv3_3).
//
// The implicit-GEMM kernels re-stage every input pixel once per filter tap (9x) -- and the global->LDS DMA issue
// rate, not the matrix pipe, is what bounds those kernels (gemm256.hip ablation).  Here a workgroup owns a 16x16
// output tile x CN output channels and, per 64-byte... per 128-byte input-channel slab (64 bf16 / 32 f32 channels),
// stages the 18x18 input patch ONCE; the nine taps read it at shifted LDS rows.  Only the per-tap weight slab
// [CN][128 B] is streamed (double-buffered, prefetched one tap ahead).  DMA per MFMA drops 3x vs the 128x128
// implicit GEMM (0.16 vs 0.5 KiB per v_mfma_f32_32x32x16_bf16).
//
// 4*WN waves: wave (wm, wn) computes pixels rows 4wm..4wm+3 (64 px = two 32-px MFMA column blocks) x channels
// 64wn..64wn+63 (two 32-row blocks): 2x2 MFMA 32x32 tiles, weights as the A operand (lane holds 4 consecutive n).
// LDS rows are 128 B; 16-byte slot swizzle phys = slot ^ ((row>>1)&7) on the DMA source address and on the read.
// Activations are zero-bordered NHWC planes, so the patch never needs border tests; tiles that overhang the right /
// bottom edge clamp their loads inside the plane and mask their stores.
#include <cstdlib>
#include <type_traits>
#include "gemm_args.h"

namespace {

constexpr int TILE = 16, PW = TILE + 2, PROWS = PW * PW;   // 18x18 = 324 patch rows
constexpr int PROWS_PAD = 328;                              // multiple of 8 (one DMA instruction = 8 rows)
constexpr int RB = 128;                                     // bytes per LDS row

// MFMA column (lane & 31) -> pixel of the wave's 2x16 block.  ds_read_b128 is served in the lane groups
// {0-3,12-15,20-27} and {4-11,16-19,28-31} (MI355X_MICROARCH.md, LDS): each group must touch 16 distinct
// (row & 15) classes of the swizzled patch.  Patch rows of one image row are consecutive, those of the next row are
// 18 (not 16) further, so the natural map (row = c >> 4, x = c & 15) collides twice per group (measured: 32 % of all
// LDS cycles were bank conflicts).  Giving each hardware group one whole image row makes the reads conflict-free.
__device__ __forceinline__ int frag_py(int c) { return ((c >= 4 && c < 12) || (c >= 16 && c < 20) || c >= 28) ? 1 : 0; }
__device__ __forceinline__ int frag_px(int c) {
    return c < 4 ? c : c < 12 ? c - 4 : c < 20 ? c - 8 : c < 28 ? c - 12 : c - 16;
}

struct ConvArgs {
    const char* in;      // [B, H+2, W+2, Cin]
    const char* w;       // [Cout][9][Cin]
    const float* bias;
    char* out;           // [B, H+2p, W+2p, Cout]
    int B, H, W, Cin, Cout, out_pad, tiles_x, tiles_y;
    int pool;            // 1: out is the 2x2-max-pooled plane [B, H/2+2p, W/2+2p, Cout] (H, W even)
    // FUSE1 kernels (conv1_1 computed inside conv1_2): the normalised image plane [B, H+2, W+2, 4] f32, conv1_1's weights as MFMA
    // fragments [2 nt][2 k-steps][64 lanes] x 16 B (conv1_pack_kernel) and its bias; `in` is unused
    const float* img;
    const char* w1f;
    const float* b1;
    // X3 form (x3 mode on PAIR planes, as conv_pp.hip's): `in` [B, H+2, W+2, 2 Cpl] with pixel = [hi (Cpl) | lo (Cpl)], `w` [Cout][9][3 Cpl] with
    // tap = [hi | lo | hi], Cin = 3 Cpl virtual channels; `out` a pair plane [.., 2 Cout] from the fp32 accumulator (bias, ReLU, pool, split)
    int Cpl;
};

#ifndef SGG_CONV_WPE
#define SGG_CONV_WPE 1      // kernel experiments only: minimum waves per SIMD the register allocation must allow
#endif
// FUSE1 (conv1_2 of VGG-16, Cin = 64 = one 128-byte slab): the 18x18x64 input patch is not loaded but COMPUTED -- conv1_1 (3 -> 64, K = 27
// padded to 32 = two MFMA k-steps per 32 pixels and 32 channels) + bias + ReLU of the 20x20 image patch, written straight into the
// swizzled LDS patch; patch pixels outside the image are conv1_2's zero padding.  conv1_1's full-resolution output (the largest
// activation of the network: 0.36 GB per 8 frames, written once and read 1.27x) never exists; +6 % MFMA work for the halo.
template <int DT, int WN, int NI, bool ONEBAR, bool FUSE1 = false, bool X3 = false>
__global__ __launch_bounds__(256 * WN, SGG_CONV_WPE) void conv3x3_spatial_kernel(const ConvArgs g) {
    static_assert(!X3 || (DT == SGG_F16 && !FUSE1), "X3 form: f16 planes");
    constexpr int NW = 4 * WN, CNW = 32 * NI, CN = CNW * WN;   // channels per wave / per block
    constexpr int ESZ = DT == SGG_F32 ? 4 : 2;
    constexpr int PATCH_B = PROWS_PAD * RB, WSLAB_B = CN * RB;
    constexpr int PI = PROWS_PAD / 8;                        // 41 patch DMA instructions
    constexpr int PI_W = (PI + NW - 1) / NW;                 // per wave
    constexpr int WI_W = (CN / 8) / NW;                      // weight-slab DMA instructions per wave (= 2)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* patch = smem;
    char* wbuf = smem + PATCH_B;                             // 2 x WSLAB_B

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // block -> (image, tile, channel block); channel blocks of one tile are adjacent (they share the input patch in L2)
    const int ncb = g.Cout / CN;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int cb = bid % ncb;
    bid /= ncb;
    const int tx = bid % g.tiles_x;
    bid /= g.tiles_x;
    const int ty = bid % g.tiles_y;
    const int b = bid / g.tiles_y;
    const int y0 = ty * TILE, x0 = tx * TILE, n0 = cb * CN;

    // ---- DMA source pointers
    // per-lane 32-bit offsets against uniform bases: the LDS-DMA then takes its scalar-base form and costs no vector ALU work
    unsigned psrc[PI_W];
#pragma unroll
    for (int j = 0; j < PI_W; ++j) {
        const int instr = min(wave + j * NW, PI - 1);
        const int r = min(instr * 8 + (lane >> 3), PROWS - 1);
        const int py = r / PW, px = r - py * PW;
        const int gy = min(y0 + py, g.H + 1), gx = min(x0 + px, g.W + 1);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        psrc[j] = (unsigned)((((long)b * (g.H + 2) + gy) * (g.W + 2) + gx) * (X3 ? 2 * g.Cpl : g.Cin) * ESZ + chunk * 16);
    }
    unsigned wsrc[WI_W];
#pragma unroll
    for (int j = 0; j < WI_W; ++j) {
        const int r = (wave * WI_W + j) * 8 + (lane >> 3);   // row = output channel inside the block
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        wsrc[j] = (unsigned)((long)(n0 + r) * 9 * g.Cin * ESZ + chunk * 16);
    }
    auto stage_patch = [&](int chunk) {
        long poff = (long)chunk * RB;
        if constexpr (X3) {                                   // virtual slab -> (plane, slab inside the plane)
            const int nc = g.Cpl * ESZ / RB;
            const int seg = (chunk >= nc) + (chunk >= 2 * nc);
            poff = (long)(chunk - seg * nc) * RB + (seg == 2 ? (long)g.Cpl * ESZ : 0);
        }
        const char* ub = uniform_ptr(g.in + poff);
#pragma unroll
        for (int j = 0; j < PI_W; ++j)
            if (wave + j * NW < PI) glds16_su(ub, psrc[j], patch + (wave + j * NW) * 8 * RB);
    };
    auto stage_w = [&](int tap, int chunk, int buf) {
        const char* ub = uniform_ptr(g.w + ((long)tap * g.Cin) * ESZ + (long)chunk * RB);
#pragma unroll
        for (int j = 0; j < WI_W; ++j) glds16_su(ub, wsrc[j], wbuf + buf * WSLAB_B + (wave * WI_W + j) * 8 * RB);
    };

    // ---- fragment addressing
    const int fr = lane & 31, fh = lane >> 5;
    int prow0[2];          // patch row of this lane's pixel (tap 0,0) for the two 32-pixel column blocks
#pragma unroll
    for (int j = 0; j < 2; ++j) prow0[j] = (wm * 4 + j * 2 + frag_py(fr)) * PW + frag_px(fr);
    int woff[NI], wkey[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int r = wn * CNW + i * 32 + fr;
        woff[i] = r * RB;
        wkey[i] = (r >> 1) & 7;
    }

    f32x16 acc[2][NI];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#ifndef SGG_CONV_ABL
#define SGG_CONV_ABL 0   // kernel experiments only: 1 no main loop (tile prologue + epilogue cost), 2 no output stores,
                         // 3 no barriers / waits (wrong results), 4 no LDS fragment reads, 5 no MFMA, 6 no global->LDS staging
#endif
    const int nchunk = SGG_CONV_ABL == 1 ? 0 : (g.Cin * ESZ) / RB;
    if constexpr (FUSE1 && DT != SGG_F32) {
        using TT = typename std::conditional<DT == SGG_BF16, bf16_t, f16_t>::type;
        char* ipatch = wbuf + 2 * WSLAB_B;                   // 20 x 20 pixels x (4 f32)
        stage_w(0, 0, 0);                                    // conv1_2's first weight slab streams under the producer
        for (int i = tid; i < 400; i += 256 * WN) {
            const int iy = i / 20, ix = i - iy * 20;
            const int gy = min(max(y0 - 1 + iy, 0), g.H + 1), gx = min(max(x0 - 1 + ix, 0), g.W + 1);
            *reinterpret_cast<f32x4*>(ipatch + i * 16) = *reinterpret_cast<const f32x4*>(g.img + (((long)b * (g.H + 2) + gy) * (g.W + 2) + gx) * 4);
        }
        // conv1_1 weight fragments (packed once per weight update: four coalesced 16-byte loads per lane) and the lane's bias groups
        u32x4 wf[2][2];
        f32x4 bq[2][4];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) wf[nt][s2] = *reinterpret_cast<const u32x4*>(g.w1f + ((nt * 2 + s2) * 64 + lane) * 16);
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) bq[nt][q4] = *reinterpret_cast<const f32x4*>(g.b1 + nt * 32 + 8 * q4 + 4 * fh);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // the image patch is visible
        constexpr int NBLK = (PROWS + 31) / 32;              // 11 blocks of 32 patch pixels
        for (int blk = wave; blk < NBLK; blk += NW) {
            const int p = blk * 32 + fr, pc = min(p, PROWS - 1);
            const int py = pc / PW, px = pc - py * PW;
            float V[32];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(ipatch + ((py + ky) * 20 + px + kx) * 16);
                    V[(ky * 3 + kx) * 3 + 0] = t4.x;
                    V[(ky * 3 + kx) * 3 + 1] = t4.y;
                    V[(ky * 3 + kx) * 3 + 2] = t4.z;
                }
#pragma unroll
            for (int k = 27; k < 32; ++k) V[k] = 0.f;
            u32x4 af1[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float t8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) t8[j] = fh ? V[s2 * 16 + 8 + j] : V[s2 * 16 + j];
                af1[s2] = pack8<TT>(t8);
            }
            const int Y = y0 - 1 + py, X = x0 - 1 + px;
            const bool inside = Y >= 0 && Y < g.H && X >= 0 && X < g.W;     // else: conv1_2's zero padding
            const int key = (pc >> 1) & 7;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                f32x16 a1;
#pragma unroll
                for (int r = 0; r < 16; ++r) a1[r] = 0.f;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) a1 = mfma_32x32x16<DT>(wf[nt][s2], af1[s2], a1);
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x4 bb = bq[nt][q4];
                    const float o0 = inside ? fmaxf(a1[4 * q4] + bb.x, 0.f) : 0.f, o1 = inside ? fmaxf(a1[4 * q4 + 1] + bb.y, 0.f) : 0.f;
                    const float o2 = inside ? fmaxf(a1[4 * q4 + 2] + bb.z, 0.f) : 0.f, o3 = inside ? fmaxf(a1[4 * q4 + 3] + bb.w, 0.f) : 0.f;
                    const u32x2 pk = {H16<TT>::pack(o0, o1), H16<TT>::pack(o2, o3)};
                    if (p < PROWS) *reinterpret_cast<u32x2*>(patch + pc * RB + (((nt * 4 + q4) ^ key) << 4) + 8 * fh) = pk;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this wave's patch rows are written before it reaches the first tap's barrier
    }
    for (int ch = 0; ch < nchunk; ++ch) {
        if (ONEBAR && ch) lds_reads_done_barrier();      // every wave is done with the previous patch and slab 0's buffer
        if (SGG_CONV_ABL != 6 && !FUSE1) {
            stage_patch(ch);
            stage_w(0, ch, 0);
        }
        for (int tap = 0; tap < 9; ++tap) {
            if constexpr (SGG_CONV_ABL == 3) {
                if (tap + 1 < 9) stage_w(tap + 1, ch, (tap + 1) & 1);
            } else if constexpr (ONEBAR) {
                // One barrier per tap: after it, slab `tap` (and the patch) have landed for every wave, and every wave
                // has finished tap-1 -- the buffer tap-1 used is free and the slab for tap+1 can be streamed into it.
                wait_vmcnt<0>();
                lds_reads_done_barrier();                // (the reads of tap-1 out of the buffer that is refilled next)
                if (tap + 1 < 9) stage_w(tap + 1, ch, (tap + 1) & 1);
            } else {
                if (tap + 1 < 9) {
                    if (SGG_CONV_ABL != 6) stage_w(tap + 1, ch, (tap + 1) & 1);
                    wait_vmcnt<WI_W>();                  // everything but the slab just issued has landed
                } else {
                    wait_vmcnt<0>();
                }
                __builtin_amdgcn_s_barrier();
            }
            const char* wb = wbuf + (tap & 1) * WSLAB_B;
            const int ky = tap / 3, kx = tap - ky * 3;
            const int dp = ky * PW + kx;
            int poff[2], pkey[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int pr = prow0[j] + dp;
                poff[j] = pr * RB;
                pkey[j] = (pr >> 1) & 7;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int slot = 2 * s + fh;
                u32x4 av[NI], bv[2];
                if constexpr (SGG_CONV_ABL == 4) {
#pragma unroll
                    for (int i = 0; i < NI; ++i) av[i] = u32x4{(unsigned)slot, (unsigned)tap, 1u, 2u};
#pragma unroll
                    for (int i = 0; i < 2; ++i) bv[i] = u32x4{(unsigned)ch, 3u, (unsigned)slot, 4u};
                } else {
#pragma unroll
                for (int i = 0; i < NI; ++i) av[i] = *reinterpret_cast<const u32x4*>(wb + woff[i] + ((slot ^ wkey[i]) << 4));
#pragma unroll
                for (int i = 0; i < 2; ++i) bv[i] = *reinterpret_cast<const u32x4*>(patch + poff[i] + ((slot ^ pkey[i]) << 4));
                }
                if constexpr (SGG_CONV_ABL == 5) {
#pragma unroll
                    for (int i = 0; i < NI; ++i) asm volatile("" ::"v"(av[i]));
#pragma unroll
                    for (int i = 0; i < 2; ++i) asm volatile("" ::"v"(bv[i]));
                    continue;
                }
#pragma unroll
                for (int pj = 0; pj < 2; ++pj)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        if constexpr (DT != SGG_F32) {
                            acc[pj][ni] = mfma_32x32x16<DT>(av[ni], bv[pj], acc[pj][ni]);
                        } else {
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                acc[pj][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                    __uint_as_float(av[ni][q]), __uint_as_float(bv[pj][q]), acc[pj][ni], 0, 0, 0);
                        }
                    }
            }
            if constexpr (!ONEBAR && SGG_CONV_ABL != 3) lds_reads_done_barrier();       // slab buffer (tap&1) and, after tap 8, the patch are free
        }
    }

    // ---- epilogue through LDS: per wave a [32 px][CNW ch] f32 tile (row stride CNW*4+16 B), one pixel block at a time
    constexpr int ESTRIDE = CNW * 4 + 16;
    constexpr int LPR = CNW / 8;                  // lanes per pixel row (8 channels each)
    constexpr int PPI = 64 / LPR;                 // pixels per pass
    char* est = smem + wave * (32 * ESTRIDE);
    float bias8[8];   // the lane's 8 output channels are the same in every store below: two 16-byte loads, once
    load8(g.bias + n0 + wn * CNW + (lane % LPR) * 8, bias8);
    // The staging tile is private to the wave (est = smem + wave * ...): after ONE workgroup barrier (every wave has
    // finished reading operand tiles out of this memory) the wave's own LDS write -> read order is all that is needed.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int pj = 0; pj < 2; ++pj) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = {acc[pj][ni][4 * q], acc[pj][ni][4 * q + 1], acc[pj][ni][4 * q + 2], acc[pj][ni][4 * q + 3]};
                *reinterpret_cast<f32x4*>(est + fr * ESTRIDE + (ni * 32 + 8 * q + 4 * fh) * 4) = v;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own LDS writes landed (same wave reads them back)
        if (g.pool) {
            // fused MaxPool2d(2): a wave's 32-pixel block is 2 image rows x 16 columns, i.e. 8 complete 2x2 windows;
            // max first, then bias + ReLU (they commute with max).  Column c of the staged tile holds pixel
            // (frag_py(c), frag_px(c)); window b = pixels x in {2b, 2b+1} of both rows.
            constexpr int WPI = 64 / LPR;                 // windows per pass
#pragma unroll
            for (int it = 0; it < (8 + WPI - 1) / WPI; ++it) {
                const int wb = lane / LPR + WPI * it, cl = (lane % LPR) * 8;
                if (wb >= 8) continue;
                const int yo = (y0 + wm * 4 + pj * 2) >> 1, xo = (x0 >> 1) + wb;
                if (2 * yo >= g.H || 2 * xo >= g.W) continue;
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = -3.0e38f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int px = 2 * wb + (q & 1), py = q >> 1;
                    // inverse of (frag_py, frag_px): column index of pixel (py, px)
                    const int col = py == 0 ? (px < 4 ? px : px < 8 ? px + 8 : px + 12) : (px < 8 ? px + 4 : px < 12 ? px + 8 : px + 16);
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(est + col * ESTRIDE + cl * 4);
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(est + col * ESTRIDE + cl * 4 + 16);
                    v[0] = fmaxf(v[0], lo.x); v[1] = fmaxf(v[1], lo.y); v[2] = fmaxf(v[2], lo.z); v[3] = fmaxf(v[3], lo.w);
                    v[4] = fmaxf(v[4], hi.x); v[5] = fmaxf(v[5], hi.y); v[6] = fmaxf(v[6], hi.z); v[7] = fmaxf(v[7], hi.w);
                }
                const int n = n0 + wn * CNW + cl;
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k] + bias8[k], 0.f);
                const int op = g.out_pad, Ho = g.H >> 1, Wo = g.W >> 1;
                if constexpr (X3) {
                    float lo8[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float h = round_as<f16_t>(v[k]);
                        lo8[k] = v[k] - h;
                        v[k] = h;
                    }
                    f16_t* o = reinterpret_cast<f16_t*>(g.out) + (((long)b * (Ho + 2 * op) + yo + op) * (Wo + 2 * op) + xo + op) * 2 * g.Cout + n;
                    store8(o, v);
                    store8(o + g.Cout, lo8);
                    continue;
                }
                const long off = (((long)b * (Ho + 2 * op) + yo + op) * (Wo + 2 * op) + xo + op) * g.Cout + n;
                if constexpr (DT == SGG_BF16) store8(reinterpret_cast<bf16_t*>(g.out) + off, v);
                else if constexpr (DT == SGG_F16) store8(reinterpret_cast<f16_t*>(g.out) + off, v);
                else store8(reinterpret_cast<float*>(g.out) + off, v);
            }
            continue;
        }
#pragma unroll
        for (int it = 0; it < 32 / PPI; ++it) {
            const int pl = lane / LPR + PPI * it, cl = (lane % LPR) * 8;   // pixel inside the 32-block, channel chunk
            const int y = y0 + wm * 4 + pj * 2 + frag_py(pl), x = x0 + frag_px(pl);
            if (y >= g.H || x >= g.W) continue;
            float v[8];
            const f32x4 lo = *reinterpret_cast<const f32x4*>(est + pl * ESTRIDE + cl * 4);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(est + pl * ESTRIDE + cl * 4 + 16);
            v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
            const int n = n0 + wn * CNW + cl;
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k] + bias8[k], 0.f);
            const int op = g.out_pad;
            if constexpr (X3) {
                float lo8[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float h = round_as<f16_t>(v[k]);
                    lo8[k] = v[k] - h;
                    v[k] = h;
                }
                f16_t* o = reinterpret_cast<f16_t*>(g.out) + (((long)b * (g.H + 2 * op) + y + op) * (g.W + 2 * op) + x + op) * 2 * g.Cout + n;
                store8(o, v);
                store8(o + g.Cout, lo8);
                continue;
            }
            const long off = (((long)b * (g.H + 2 * op) + y + op) * (g.W + 2 * op) + x + op) * g.Cout + n;
            if (SGG_CONV_ABL == 2) continue;
            if constexpr (DT == SGG_BF16) store8(reinterpret_cast<bf16_t*>(g.out) + off, v);
            else if constexpr (DT == SGG_F16) store8(reinterpret_cast<f16_t*>(g.out) + off, v);
            else store8(reinterpret_cast<float*>(g.out) + off, v);
        }
    }
}

template <int DT, int WN, int NI, bool ONEBAR = false, bool FUSE1 = false, bool X3 = false>
int launch_spatial(const ConvArgs& g, hipStream_t s) {
    constexpr int CN = 32 * NI * WN;
    constexpr int smem_main = PROWS_PAD * RB + 2 * CN * RB + (FUSE1 ? 400 * 16 : 0), smem_epi = 4 * WN * 32 * (32 * NI * 4 + 16);   // operand tiles | epilogue staging
    constexpr int smem = smem_main > smem_epi ? smem_main : smem_epi;
    static_assert(smem <= 160 * 1024, "fits the CU's LDS");
    auto k = conv3x3_spatial_kernel<DT, WN, NI, ONEBAR, FUSE1, X3>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return SGG_ERR_LAUNCH;
        attr_done = true;
    }
    const int blocks = g.B * g.tiles_y * g.tiles_x * (g.Cout / CN);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256 * WN), smem, s, g);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

}  // namespace

// returns SGG_OK, or 1 if the shape is not handled here (caller falls through to the implicit-GEMM kernels)
int sgg_launch_conv_spatial(const void* in, const void* w, const float* bias, void* out, int out_pad, int B, int H, int W,
                            int Cin, int Cout, int dt, int pool, hipStream_t s) {
    ConvArgs g{};
    g.pool = pool;
    g.in = (const char*)in; g.w = (const char*)w; g.bias = bias; g.out = (char*)out;
    g.B = B; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.out_pad = out_pad;
    g.tiles_x = (W + TILE - 1) / TILE;
    g.tiles_y = (H + TILE - 1) / TILE;
    // (4 waves x [64 px x 128 ch] per wave -- NI = 4, a quarter fewer LDS reads -- measured 10-20 % slower than
    //  8 waves x [64 x 64]: the kernel is latency-, not LDS-bandwidth-bound, and the extra waves hide more of it.)
    static const bool onebar = [] { const char* e = getenv("SGG_CONV_ONEBAR"); return e && e[0] == '1'; }();
    if (Cout % 128 == 0 && dt == SGG_BF16 && onebar) return launch_spatial<SGG_BF16, 2, 2, true>(g, s);
    static const int wide = [] { const char* e = getenv("SGG_CONV_WIDE"); return e ? atoi(e) : 0; }();     // experiments: 64 px x 128 ch per wave
    if (wide && Cout % 256 == 0 && dt != SGG_F32)
        return dt == SGG_BF16 ? launch_spatial<SGG_BF16, 2, 4>(g, s) : launch_spatial<SGG_F16, 2, 4>(g, s);
    if (Cout % 128 == 0)
        return dt == SGG_BF16 ? launch_spatial<SGG_BF16, 2, 2>(g, s) : dt == SGG_F16 ? launch_spatial<SGG_F16, 2, 2>(g, s) : launch_spatial<SGG_F32, 2, 2>(g, s);
    if (Cout % 64 == 0)
        return dt == SGG_BF16 ? launch_spatial<SGG_BF16, 1, 2>(g, s) : dt == SGG_F16 ? launch_spatial<SGG_F16, 1, 2>(g, s) : launch_spatial<SGG_F32, 1, 2>(g, s);
    return 1;
}

// X3 form (pair planes, weights [hi | lo | hi] per tap): the 64-channel layer of VGG-16 (conv1_2) and anything else the patch kernel of
// conv_pp.hip does not take.  SGG_OK, or 1 if the shape is not handled.
int sgg_launch_conv_spatial_x3(const void* in, const void* w3, const float* bias, void* out, int out_pad, int B, int H, int W, int Cpl, int Cout,
                               int pool, hipStream_t s) {
    if (Cpl % 64 || Cout % 64) return 1;
    ConvArgs g{};
    g.pool = pool;
    g.in = (const char*)in; g.w = (const char*)w3; g.bias = bias; g.out = (char*)out;
    g.B = B; g.H = H; g.W = W; g.Cin = 3 * Cpl; g.Cpl = Cpl; g.Cout = Cout; g.out_pad = out_pad;
    g.tiles_x = (W + TILE - 1) / TILE;
    g.tiles_y = (H + TILE - 1) / TILE;
    if (Cout % 128 == 0) return launch_spatial<SGG_F16, 2, 2, false, false, true>(g, s);
    return launch_spatial<SGG_F16, 1, 2, false, false, true>(g, s);
}

namespace {
// conv1_1's weights [64][27] f32 -> the A-operand fragments of its two MFMA k-steps: frag[nt][s][lane] = 8 values k = s*16 + (lane>>5)*8 + j of
// channel nt*32 + (lane & 31), zero for k >= 27, in the 16-bit type T
template <typename T>
__global__ __launch_bounds__(256) void conv1_pack_kernel(const float* __restrict__ w1, char* __restrict__ frags) {
    const int i = threadIdx.x, lane = i & 63, s = (i >> 6) & 1, nt = i >> 7;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = s * 16 + (lane >> 5) * 8 + j;
        v[j] = k < 27 ? w1[(nt * 32 + (lane & 31)) * 27 + k] : 0.f;
    }
    *reinterpret_cast<u32x4*>(frags + i * 16) = pack8<T>(v);
}
}  // namespace

int sgg_launch_conv1_pack(const float* w1, void* frags, int dt, hipStream_t s) {
    if (dt == SGG_BF16) hipLaunchKernelGGL(conv1_pack_kernel<bf16_t>, dim3(1), dim3(256), 0, s, w1, (char*)frags);
    else hipLaunchKernelGGL(conv1_pack_kernel<f16_t>, dim3(1), dim3(256), 0, s, w1, (char*)frags);
    SGG_CHECK_LAUNCH();
    return SGG_OK;
}

// conv1_1 + ReLU + conv1_2 + ReLU (+ MaxPool2d(2)) of VGG-16 in one launch (16-bit modes): see FUSE1 above.  img: [B, H+2, W+2, 4] f32 (zero border).
int sgg_launch_conv1_block(const float* img, const void* w1, const float* b1, const void* w2, const float* b2, void* out, int out_pad, int B,
                           int H, int W, int dt, int pool, hipStream_t s) {
    if (dt != SGG_BF16 && dt != SGG_F16) return 1;
    ConvArgs g{};
    g.pool = pool;
    g.in = nullptr; g.w = (const char*)w2; g.bias = b2; g.out = (char*)out;
    g.img = img; g.w1f = (const char*)w1; g.b1 = b1;
    g.B = B; g.H = H; g.W = W; g.Cin = 64; g.Cout = 64; g.out_pad = out_pad;
    g.tiles_x = (W + TILE - 1) / TILE;
    g.tiles_y = (H + TILE - 1) / TILE;
    return dt == SGG_BF16 ? launch_spatial<SGG_BF16, 1, 2, false, true>(g, s) : launch_spatial<SGG_F16, 1, 2, false, true>(g, s);
}
