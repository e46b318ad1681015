// Stream-K tile scheduler shared by the 256x256 ping-pong GEMM (gemm256.hip) and the ping-pong 3x3 convolution (conv_pp.hip).
//
// Both kernels run ONE output tile per workgroup with one workgroup per CU, so a launch costs ceil(tiles / CUs) rounds whatever the
// remainder is: the fc6 weight gradient's 1568 tiles are 6.125 rounds and cost 7 (round 2 cut the last 32 tiles off into a split-K
// launch: 130 us for 2 % of the work), conv4_x's 400 tiles are 1.56 rounds and cost 2.  Here the launch is persistent (grid = CU count)
// and the unit of work is (tile, K-tile):
//
//   * the first dp_rounds * G tiles (logical tile order, G = grid) run data-parallel, tile r * G + lc on workgroup lc in round r --
//     the same XCD-contiguous order as the plain launch, so the L2 picture does not change;
//   * the last sk_tiles = G + (tiles mod G) tiles are ONE sequence of sk_tiles * nt K-tile units cut into G equal ranges
//     [B(lc), B(lc+1)).  Since sk_tiles >= G a range is at least one tile long: it holds at most the TAIL part of its first tile
//     (K-tiles [kb, nt)), whole tiles, and the HEAD part of its last tile (K-tiles [0, ke)) -- never a middle part.
//
// A split tile is computed as ONE accumulation chain: the workgroup with the head part runs it FIRST, publishes its raw accumulator
// registers (fp32, 256 KB per tile) and goes on; the workgroup with the tail part runs it LAST, starts from those registers and
// writes the tile.  The K order of every output element is therefore the plain kernel's, and so are the bits (tests compare them).
// The hand-over is the guide's R1 recipe (cdna_hip_programming.md 6 G16): 16-byte write-through (sc1) stores, every storing wave
// drains vmcnt, __syncthreads, one lane stores the flag with a relaxed agent-scope atomic; the consumer polls that one word relaxed
// and reads the slot with sc1 loads.  Nobody ever WAITS: if the flag is not there when the tail part starts (the producer ran its
// part at the very start of the launch, so this means it was not resident yet), the consumer computes the whole tile itself from
// K-tile 0 -- the same chain, the same bits, and no residency assumption.  Flags carry the launch's epoch (a per-workspace counter
// of the host side), so nothing is cleared between launches.
#pragma once
#include "common.h"

constexpr int SK_MIN_SEG = 8;                       // K-tiles: no part is shorter (boundaries closer to a tile edge snap onto it)
constexpr int SK_THREADS = 512;
constexpr int SK_SLOT_BYTES = SK_THREADS * 128 * 4; // a workgroup's accumulators: 128 fp32 registers per thread
constexpr int SK_MAX_GRID = 256;                    // slots / flags a workspace holds (MI355X: 256 CUs)

// first K-tile unit of range i (of G) over `units` = sk_tiles * nt units; multiples of 4 inside a tile, 0 / units at the ends
__host__ __device__ inline long sk_bound(int i, int G, long units, int nt) {
    long b = (long)i * units / G;
    long t = b / nt;
    int off = (int)(b - t * nt);
    if (off < SK_MIN_SEG) off = 0;
    else if (nt - off < SK_MIN_SEG) { off = 0; t += 1; }
    else off &= ~3;
    return t * nt + off;
}

struct SkRange {
    int tail_tile, tail_kb;     // tail part: K-tiles [tail_kb, nt) of SK tile tail_tile (tail_kb == 0: none)
    int whole0, n_whole;        // whole SK tiles [whole0, whole0 + n_whole)
    int head_tile, head_ke;     // head part: K-tiles [0, head_ke) of SK tile head_tile (head_ke == 0: none)
};

__host__ __device__ inline SkRange sk_range(int lc, int G, int sk_tiles, int nt) {
    const long units = (long)sk_tiles * nt;
    const long b0 = sk_bound(lc, G, units, nt), b1 = sk_bound(lc + 1, G, units, nt);
    SkRange r;
    r.tail_tile = (int)(b0 / nt);
    r.tail_kb = (int)(b0 - (long)r.tail_tile * nt);
    r.head_tile = (int)(b1 / nt);
    r.head_ke = (int)(b1 - (long)r.head_tile * nt);
    r.whole0 = r.tail_tile + (r.tail_kb > 0 ? 1 : 0);
    r.n_whole = r.head_tile - r.whole0;
    return r;
}

// How a launch of `tiles` tiles on G workgroups is cut: data-parallel rounds + stream-K region.  sk_tiles == 0: plain rounds only.
inline void sk_split(long tiles, int G, int& dp_rounds, int& sk_tiles) {
    const long R = tiles / G, rem = tiles % G;
    if (rem == 0 || R == 0) { dp_rounds = (int)R; sk_tiles = 0; return; }
    dp_rounds = (int)R - 1;
    sk_tiles = G + (int)rem;
}

#ifdef __HIPCC__
typedef __attribute__((address_space(1))) unsigned sk_gu32;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t sk_slot_rsrc(char* ws, int slot) {
    return __builtin_amdgcn_make_buffer_rsrc(ws + (long)slot * SK_SLOT_BYTES, 0, SK_SLOT_BYTES, 0x00020000);
}
// 16 bytes of this thread's accumulators, write-through (aux 16 = sc1): visible to every XCD without a release fence
__device__ __forceinline__ void sk_store16(__amdgpu_buffer_rsrc_t rs, int idx, int tid, const f32x4& v) {
    u32x4 u;
    u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
    __builtin_amdgcn_raw_buffer_store_b128(u, rs, tid * 16, idx * SK_THREADS * 16, 16);   // (piece index in the SCALAR offset: one lane offset for all 32 pieces)
}
__device__ __forceinline__ f32x4 sk_load16(__amdgpu_buffer_rsrc_t rs, int idx, int tid) {
    const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs, tid * 16, idx * SK_THREADS * 16, 16);
    f32x4 v = {__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w)};
    return v;
}
// after the slot's stores: every storing wave drains, the workgroup meets, ONE lane publishes the epoch
__device__ __forceinline__ void sk_publish(unsigned* flags, int slot, unsigned epoch, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store((sk_gu32*)(flags + slot), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one relaxed look at the flag, broadcast to the workgroup through LDS word `lds_word` (the kernel's own LDS array: a second
// __shared__ object would de-pipeline the K loop, guide 5.7); true = the slot holds this launch's partial
__device__ __forceinline__ bool sk_ready(unsigned* flags, int slot, unsigned epoch, int tid, volatile unsigned* lds_word) {
    __syncthreads();
    if (tid == 0) *lds_word = __hip_atomic_load((sk_gu32*)(flags + slot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch ? 1u : 0u;
    __syncthreads();
    const bool ok = __builtin_amdgcn_readfirstlane(*lds_word) != 0u;     // (an LDS read is per-lane to the compiler: keep the answer in an SGPR,
                                                                         //  or every address and loop bound derived from it turns divergent)
    __syncthreads();
    return ok;
}
#endif

// host side (gemm256.hip): the workspace registered for a stream, or NULL
struct SkWorkspace {
    char* slots;
    unsigned* flags;
    unsigned epoch;
    int max_grid;
};
SkWorkspace* sgg_sk_workspace_of(void* stream);
int sgg_sk_grid();                                  // workgroups of a persistent launch = CUs of the device
