// tile_kernels.hip — LDS-tiled passes: up to 10 radix-2 levels per trip through HBM.
//
// A workgroup owns a tile of T = 2^LOGT blocks (block stride 2^s) x W words.  Each lane keeps R = 2^LOGR
// words of ONE column in VGPRs and the tile visits LDS only to change which blocks a lane holds:
//
//   layout A  lane holds blocks q = [half*T/2 +] j*G + g   (j < R)  -> the high LOGR levels are in-thread
//   layout B  lane holds blocks q = g'*R + k                (k < R)  -> the low L2 = log2(G) levels are in-thread
//
// (G = waves per workgroup, g = wave id.)  One LDS round trip (ds_write_b32 / ds_read_b32, rows of W
// consecutive words -> conflict-free) turns A into B or back.  Twiddles depend only on the block index,
// never on the column, so they stay wave-uniform and are fetched with scalar loads.
//
// PAIR variant (W = 32): a wave covers two half-tiles, lanes 0-31 hold block q and lanes 32-63 block
// q + T/2 — butterfly partners at the top level.  That level is done across lanes with
// v_permlane32_swap: swapping registers (ja, jb) gives the low half-wave both operands of pair ja and
// the high half-wave both operands of pair jb, so every lane does one full butterfly per two registers
// (no redundant work) and all remaining levels again see wave-uniform twiddles.  Only ONE swap per pair is
// needed: on the DIF side the blocks are LOADED in that "paired" order (register 2i: block (2i+half)*G+g,
// register 2i+1: the same + T/2) and swapped into layout A after the butterfly; on the DIT side they are
// swapped out of layout A before it and STORED from the paired order.  This is what makes a
// 1024-block tile fit: 1024 x 32 words = 128 KiB of the CU's 160 KiB LDS, in 128-byte row segments.
//
// Occupancy variants (chosen by the host plan):
//   SPLIT = 2   an exchange never mixes columns, so it can run 16 columns at a time through a 64 KiB buffer:
//               the 1024-block MID tile then needs 55 VGPRs / 64 KiB and two workgroups (32 waves) share a CU
//               (measured 2.01 -> 1.66 ms for the same arithmetic);
//   LOGR = 4    "slim" 8/9-level tiles keep 16 words per lane (<= 64 VGPRs, 1024 threads): 32 waves per CU in
//               the HBM-bound outer passes.
//
//   DIF  load A -> [pair level] -> LOGR levels -> A=>B -> L2 levels -> store B
//   DIT  load B -> L2 levels -> B=>A -> LOGR levels -> [pair level] -> store A
//   MID  DIF half, multiply block p by D[bitrev(p)] (RS.cpp:51-59), DIT half: 2*LOGT levels per HBM trip
//
// With N = 2^19 the encode is 3 launches: DIF over levels 18..10, MID over 9..0 twice, DIT over 10..18
// (the reference needs 2 x (3 sweeps + twiddle sweep), ntt.cpp:412-446).
//
// Decoder modes (the split transform of decode.hip; same tiles, same level code):
//   DIF_ROWS     DIF whose input blocks are multiplied by per-block factors on the way in (zero = block not in use); optionally only the first
//                block groups of the pass are launched
//   DIF_IMPULSE  the 1024-block DIF tile at s = 0 when only its first few blocks are non-zero: the six levels before the exchange as one
//                multiply-add per word and block in use
//   MID_ADD      MID with "+ addend[p] * factor[p]" between its halves (optionally also storing the tile as it is after the first half)
//   MID_UP       that second half alone, from such a stored tile (fastecc_repair's second chain)
//   DIT_ROWS     DIT that stores only the blocks with a non-zero factor, times that factor (the scatter)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gf.hpp"
#include "kernels.hpp"
#include "ntt_device.hpp"

namespace fastecc {

template <int LOGT, int LOGR, bool PAIR, int SPLIT = 1>
struct TileCfg {
    static constexpr int T = 1 << LOGT;
    static constexpr int R = 1 << LOGR;
    static constexpr int L2 = LOGT - LOGR - (PAIR ? 1 : 0);  // levels done in layout B
    static constexpr int G = 1 << L2;                        // waves per workgroup
    static constexpr int W = PAIR ? 32 : 64;                 // words per tile row
    static constexpr int THREADS = G * 64;
    // An exchange only ever moves data between registers of the SAME column, so it can be done SPLIT
    // column groups at a time through an LDS buffer SPLIT times smaller (lanes of the other groups idle
    // during a round).  SPLIT = 2 brings the 1024-block pair tile down to 64 KiB: two workgroups per CU.
    static constexpr int WS = W / SPLIT;                     // words per LDS row
    static constexpr int LDS_BYTES = T * WS * 4;
    static_assert(L2 >= 1 && L2 <= LOGR, "tile shape");
    static_assert(THREADS <= 1024, "workgroup size");
};

// Top level of a PAIR tile (level sl + LOGR as seen from layout A: partner distance "2^LOGR registers" is
// the other half-wave).  Its 2^LOGR twiddles, one per register, are contiguous in the level table; they
// are fetched 16 at a time to bound SGPR pressure.  The low half-wave needs w[ja], the high one w[jb]:
// selected with (wa ^ ((wa ^ wb) & upper_mask)) — one scalar xor, two vector ops — because writing it as a
// ?: on SGPR array elements makes the compiler build a 32-way compare/select chain.
template <int LOGR>
__device__ __forceinline__ uint32_t pair_twiddle(uint32_t wa, uint32_t wb, uint32_t upper_mask)
{
    return wa ^ ((wa ^ wb) & upper_mask);
}

// Decimation in frequency: (a, b) -> (a + b, (a - b) * w).
template <int LOGR>
__device__ __forceinline__ void pair_level_dif(uint32_t (&x)[1 << LOGR][1], const uint32_t* __restrict__ twl, uint32_t off, int sl,
                                               uint32_t upper_mask)
{
    constexpr int R = 1 << LOGR, CH = R < 16 ? R : 16;
    const_u32_ptr p = as_constant(twl) + (1u << (sl + LOGR)) + (off << LOGR);
#pragma unroll
    for (int c0 = 0; c0 < R; c0 += CH) {
        uint32_t w[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) w[i] = p[c0 + i];
#pragma unroll
        for (int i = 0; i < CH; i += 2) {
            // x arrives in the PAIRED order (see the kernel): this lane already holds both operands of one pair
            const int ja = c0 + i, jb = ja + 1;
            const uint32_t a = x[ja][0], b = x[jb][0];
            const uint32_t sum = gf::add(a, b);
            const uint32_t dif = gf::mul_mont(gf::sub(a, b), pair_twiddle<LOGR>(w[i], w[i + 1], upper_mask));
            // one swap turns (sum, dif) of pair ja [low lanes] / pair jb [high lanes] into layout A
            const auto o = __builtin_amdgcn_permlane32_swap(sum, dif, false, false);
            x[ja][0] = o[0];
            x[jb][0] = o[1];
        }
    }
}

// Decimation in time: (a, b) -> (a + b*w, a - b*w).
template <int LOGR>
__device__ __forceinline__ void pair_level_dit(uint32_t (&x)[1 << LOGR][1], const uint32_t* __restrict__ twl, uint32_t off, int sl,
                                               uint32_t upper_mask)
{
    constexpr int R = 1 << LOGR, CH = R < 16 ? R : 16;
    const_u32_ptr p = as_constant(twl) + (1u << (sl + LOGR)) + (off << LOGR);
#pragma unroll
    for (int c0 = 0; c0 < R; c0 += CH) {
        uint32_t w[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) w[i] = p[c0 + i];
#pragma unroll
        for (int i = 0; i < CH; i += 2) {
            // one swap gives this lane both operands of one pair; the results stay in the PAIRED order and are
            // stored from there (see the kernel)
            const int ja = c0 + i, jb = ja + 1;
            const auto r = __builtin_amdgcn_permlane32_swap(x[ja][0], x[jb][0], false, false);
            const uint32_t a = r[0];
            const uint32_t b = gf::mul_mont(r[1], pair_twiddle<LOGR>(w[i], w[i + 1], upper_mask));
            x[ja][0] = gf::add(a, b);
            x[jb][0] = gf::sub(a, b);
        }
    }
}

// Per-tile addressing.  Global memory is reached through raw buffer descriptors (one per tile, built from
// wave-uniform values only) so that each access is ONE instruction:
//     buffer_load_dword  v, v_lane_off, s[desc], s_block_off offen
// v_lane_off  = 32-bit byte offset of the lane inside the tile (column + half-tile of a PAIR tile); it
//               does not depend on the tile, and is set past num_records for lanes whose column does not
//               exist (col >= S): the hardware bounds check then returns 0 / drops the store, so ragged
//               block sizes need no branches;
// s_block_off = 32-bit byte offset of the tile block, wave-uniform (SGPR).
// The host only selects a tile pass when a tile spans < 2^32 - 2^16 bytes (tile_fits, plan.hip), so live lane
// offset + block offset stay below num_records = 2^32-1 and "offset | dead_mask" = 2^32-1 is always out of range.
// Workgroup barrier for the LDS exchanges.  __syncthreads() also fences global memory (s_waitcnt
// vmcnt(0)), which would drain the previous tile's stores at every exchange;
// the exchange only needs this wave's LDS traffic to have completed.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

struct TileView {
    uint32_t lo, hi;          // tile group: stripe block of tile block q is (hi << (s+LOGT)) + lo + (q << s)
    uint32_t dead_mask;       // all ones in lanes whose column does not exist
    __amdgpu_buffer_rsrc_t in, out;
    __amdgpu_buffer_rsrc_t add;  // MODE_MID_ADD / MODE_MID_UP: the tile's blocks of TileArgs::addend
    __amdgpu_buffer_rsrc_t keep; // MODE_MID_ADD with TileArgs::keep: the tile's blocks there
    // WIDE tiles only: descriptors of the upper half of the tile's blocks (block T/2 onwards).  A tile whose blocks span
    // up to 2^33 bytes is then addressed as two windows of < 2^32 bytes each.
    __amdgpu_buffer_rsrc_t in_hi, out_hi;
    // MULTI tiles only: where the tile starts, descriptors are built per window (nothing but two pointers stays live)
    const uint32_t* in_base;
    uint32_t* out_base;
    uint32_t first_block;  // stripe block of tile block 0 (for the in_rows / out_rows bounds)
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_desc(const uint32_t* p, uint32_t num_records = 0xFFFFFFFFu)
{
    // the pointer is wave-uniform; passing its halves through readfirstlane makes that provable to the
    // compiler, which otherwise wraps every buffer op in a waterfall loop (cdna_hip_programming.md T20)
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    void* q = reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(num_records), 0x00020000);
}

// SZ: what the launcher knows about TileArgs::s — 1: it is zero, 0: it is not, -1: decided in the kernel (a uniform branch between the two
// versions of the low levels: fine where registers are plentiful; in the 1024-block outer tiles, 32 values per lane under a cap of 64 VGPRs,
// the allocator parked twelve of the values in scratch memory around that branch).
template <int LOGT, int LOGR, bool PAIR, int MODE, int SPLIT = 1, int NWIN = 1, int SZ = -1>
__global__ __launch_bounds__((TileCfg<LOGT, LOGR, PAIR>::THREADS), (LOGR <= 4 || SPLIT > 1 ? 8 : 4)) void ntt_tile_kernel(const TileArgs a)
{
    // NWIN address windows per tile: 1 = one buffer descriptor (blocks span < 2^32 bytes), 2 = WIDE (two descriptors kept in
    // SGPRs, < 2^33), 4 / 8 / 16 = MULTI (descriptors built per window from the tile's base pointers, < 2^34 .. 2^36)
    constexpr bool WIDE = NWIN == 2, MULTI = NWIN > 2;
    constexpr bool DIFK = MODE == MODE_DIF || MODE == MODE_DIF_ROWS || MODE == MODE_DIF_IMPULSE, MIDK = MODE == MODE_MID || MODE == MODE_MID_ADD || MODE == MODE_MID_UP;
    static_assert(MODE != MODE_DIF_IMPULSE || (PAIR && NWIN == 1 && LOGT == 10 && LOGR == 5), "impulse form: the 1024-block pair tile");
    static_assert(NWIN == 1 || (PAIR && !MIDK), "windows: outer pair tiles only");
    static_assert((MODE != MODE_DIF_ROWS && MODE != MODE_DIT_ROWS) || PAIR, "per-block factors: pair tiles");
    constexpr bool ADDK = MODE == MODE_MID_ADD || MODE == MODE_MID_UP;
    static_assert(!ADDK || PAIR, "addend: pair tiles");
    static_assert(NWIN == 1 || NWIN == 2 || NWIN == 4 || NWIN == 8 || NWIN == 16, "1, 2, 4, 8 or 16 windows");
    static_assert(!MULTI || (NWIN <= TileCfg<LOGT, LOGR, PAIR>::G && NWIN <= TileCfg<LOGT, LOGR, PAIR>::R), "a wave's blocks must fit one window");
    using C = TileCfg<LOGT, LOGR, PAIR, SPLIT>;
    using View = TileView;
    constexpr int R = C::R, G = C::G, W = C::W, L2 = C::L2, T = C::T;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];

    const uint32_t g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t c = PAIR ? (lane & 31u) : lane;
    const uint32_t half = PAIR ? (lane >> 5) : 0u;
    const uint32_t upper_mask = 0u - half;  // all ones in the high half-wave of a PAIR tile
    const int s = MIDK ? 0 : a.s;
    // MID with fewer parity than data blocks: only output positions that are multiples of 2^fold are kept, stored
    // at position >> fold.  fold <= L2, so whether a wave's blocks survive depends on g alone.
    const int fold = MIDK ? a.fold : 0;
    // Decoder's first pass (WIDE DIF tiles only): input block u is block u/2 of `in` (u even) or `in_odd` (u odd) times
    // row_factor[u].  s >= 1, so a whole tile reads ONE of the two buffers, as a tile with half the block stride.
    const bool gather = WIDE && MODE == MODE_DIF && a.row_factor != nullptr;  // uniform

    // Block held in register j (layout A) / k (layout B), split into a wave-uniform part (SGPRs) and the
    // half-wave part of a PAIR tile, which is folded once into per-lane offsets.
    const uint32_t qa_u = g;                        // + j*G   (+ half*T/2 per lane)
    const uint32_t qb_u = (PAIR ? 2u * g : g) * R;  // + k     (+ half*R   per lane)
    const uint32_t qa_l = half * (T / 2), qb_l = half * R;
    constexpr int WS = C::WS;
    const uint32_t my_round = c / WS;  // the exchange round this lane's column takes part in (always 0 when SPLIT == 1)
    uint32_t* lds_a = lds + qa_l * WS + (c % WS);
    uint32_t* lds_b = lds + qb_l * WS + (c % WS);
    const uint32_t lane_a = (((qa_l << s) * a.ld) + c) * 4u;
    const uint32_t lane_b = (((qb_l << s) * a.ld) + c) * 4u;
    const uint32_t lane_p = ((((half * G) << s) * a.ld) + c) * 4u;  // paired order: high half-wave is G blocks further
    const uint32_t row_bytes = (a.ld * 4u) << s;    // distance between consecutive tile blocks
    const int sl = s + L2;                          // layout A as seen by dif_levels/dit_levels

    auto view_of = [&](uint32_t tile) {
        View v;
        // `tile` is wave-uniform by construction; readfirstlane makes it provably so, which keeps every
        // twiddle fetch below a SCALAR load.
        tile = __builtin_amdgcn_readfirstlane(tile);
        uint32_t cc = tile % a.col_chunks;
        uint32_t grp = tile / a.col_chunks;
        // XCD-aware order (speed only): workgroup b runs on XCD b % 8.
        //   1: each XCD takes a CONTIGUOUS run of column chunks of a block group instead of every 8th one — its
        //      requests to a 4 KiB block fall into one 512-byte stretch and arrive close together;
        //   2: each XCD takes whole block groups (all column chunks, consecutive in its own dispatch order).
        if (a.xcd_swizzle == 1 && (a.col_chunks & 7u) == 0) {
            cc = (cc & 7u) * (a.col_chunks >> 3) + (cc >> 3);
        } else if (a.xcd_swizzle == 2 && ((a.tiles / a.col_chunks) & 7u) == 0) {
            const uint32_t x = tile & 7u, i = tile >> 3;
            cc = i % a.col_chunks;
            grp = (i / a.col_chunks) * 8u + x;
        }
        v.lo = grp & ((1u << s) - 1u);
        v.hi = grp >> s;
        v.dead_mask = (cc * W + c < a.S) ? 0u : 0xFFFFFFFFu;
        const size_t origin = (size_t)((v.hi << (s + LOGT)) + v.lo) * a.ld + cc * W;
        // A stripe that holds only `rows` blocks (zero-extended data, truncated parity): the descriptor ends after the
        // last existing block of this column chunk, so loads beyond return 0 and stores beyond are dropped.
        auto window = [&](uint32_t rows, size_t first_block) -> uint32_t {
            if (rows == 0) return 0xFFFFFFFFu;
            if (first_block >= rows) return 0u;
            const uint64_t bytes = ((uint64_t)(rows - first_block) * a.ld - cc * W) * 4u;
            return bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)bytes;
        };
        const size_t block0 = (size_t)(v.hi << (s + LOGT)) + v.lo;
        const size_t out_block0 = fold ? (size_t)((v.hi << LOGT) >> fold) : block0;
        v.in = make_desc(a.in + origin, window(a.in_rows, block0));
        v.out = make_desc(a.out + (fold ? out_block0 * a.ld + cc * W : origin), window(a.out_rows, out_block0));
        if constexpr (ADDK) v.add = make_desc(a.addend + (size_t)((v.hi << LOGT) >> a.addend_shift) * a.ld + cc * W);  // (MID: s = 0, lo = 0)
        if constexpr (MODE == MODE_MID_ADD) v.keep = make_desc((a.keep ? a.keep : a.out) + origin);
        if constexpr (MULTI) {
            v.in_base = a.in + origin;
            v.out_base = a.out + origin;
            v.first_block = (uint32_t)block0;
        }
        if constexpr (WIDE) {
            const size_t upper = origin + ((size_t)(T / 2) << s) * a.ld;
            const size_t upper_block0 = block0 + ((size_t)(T / 2) << s);
            v.in_hi = make_desc(a.in + upper, window(a.in_rows, upper_block0));
            v.out_hi = make_desc(a.out + upper, window(a.out_rows, upper_block0));
            if (gather) {
                const uint32_t* half_stripe = (v.lo & 1u) ? a.in_odd : a.in;
                const size_t o = (size_t)((v.hi << (s + LOGT - 1)) + (v.lo >> 1)) * a.ld + cc * W;
                v.in = make_desc(half_stripe + o);
                v.in_hi = make_desc(half_stripe + o + ((size_t)(T / 2) << (s - 1)) * a.ld);
            }
        }
        return v;
    };
    // WIDE, layout B (the only one that goes through load_rows/store_rows in a pair tile): a wave's blocks
    // [2g*R, 2g*R + 2R) lie entirely in one window
    const bool upper_wave = WIDE && g >= G / 2;
    // MULTI: window w covers tile blocks [w*T/NWIN, (w+1)*T/NWIN)
    auto window_desc = [&](const uint32_t* base, uint32_t rows, const View& v, uint32_t w) {
        const size_t first = (size_t)v.first_block + ((size_t)(w * (T / NWIN)) << s);
        uint32_t nrec = 0xFFFFFFFFu;
        if (rows != 0) {
            const uint32_t cc_words = (uint32_t)((base - (v.first_block * (size_t)a.ld + (base == v.in_base ? a.in : a.out))));  // = cc * W
            const uint64_t bytes = first >= rows ? 0 : ((uint64_t)(rows - first) * a.ld - cc_words) * 4u;
            nrec = bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)bytes;
        }
        return make_desc(base + ((size_t)(w * (T / NWIN)) << s) * a.ld, nrec);
    };
    const uint32_t my_window = MULTI ? g * NWIN / G : 0u;  // layout B: a wave's blocks [2g*R, 2g*R + 2R) lie in one window
    auto load_rows = [&](uint32_t (&r)[R][1], const View& vv, uint32_t lane_off, uint32_t q0, uint32_t qstep) {
        View v = vv;
        if constexpr (WIDE) {
            if (upper_wave) v.in = vv.in_hi, q0 -= T / 2;
        }
        if constexpr (MULTI) {
            v.in = window_desc(vv.in_base, a.in_rows, vv, my_window);
            q0 -= my_window * (T / NWIN);
        }
        const uint32_t voff = lane_off | v.dead_mask;
        // running block offset kept in ONE SGPR: the empty asm stops the compiler from materialising all R
        // offsets up front (they would sit in SGPRs for the whole kernel and spill to VGPR lanes)
        uint32_t soff = q0 * row_bytes;
        const uint32_t step = qstep * row_bytes;
#define FASTECC_LOAD_LOOP(AUX)                                                         \
    _Pragma("unroll") for (int j = 0; j < R; ++j) {                                     \
        r[j][0] = __builtin_amdgcn_raw_buffer_load_b32(v.in, voff, soff, AUX);          \
        soff += step;                                                                   \
        asm volatile("" : "+s"(soff));                                                  \
    }
        // every stripe word is read once and written once per pass: the non-temporal policy (aux bit 1) keeps the
        // stream from displacing itself in L2/MALL — measured +4..6 % on the whole encode (profiles/r01)
        if (a.cache_policy & 1) { FASTECC_LOAD_LOOP(2) } else { FASTECC_LOAD_LOOP(0) }
#undef FASTECC_LOAD_LOOP
    };
    auto store_rows = [&](const uint32_t (&r)[R][1], const View& vv, uint32_t lane_off, uint32_t q0, uint32_t qstep) {
        View v = vv;
        if constexpr (WIDE) {
            if (upper_wave) v.out = vv.out_hi, q0 -= T / 2;
        }
        if constexpr (MULTI) {
            v.out = window_desc(vv.out_base, a.out_rows, vv, my_window);
            q0 -= my_window * (T / NWIN);
        }
        const uint32_t voff = lane_off | v.dead_mask;
        uint32_t soff = q0 * row_bytes;
        const uint32_t step = qstep * row_bytes;
#define FASTECC_STORE_LOOP(AUX)                                                        \
    _Pragma("unroll") for (int j = 0; j < R; ++j) {                                     \
        __builtin_amdgcn_raw_buffer_store_b32(r[j][0], v.out, voff, soff, AUX);         \
        soff += step;                                                                   \
        asm volatile("" : "+s"(soff));                                                  \
    }
        if (a.cache_policy & 2) { FASTECC_STORE_LOOP(2) } else { FASTECC_STORE_LOOP(0) }
#undef FASTECC_STORE_LOOP
    };
    // Paired order of a PAIR tile: register 2i <-> block g + 2i*G (+ G in the high half-wave), register 2i+1 <-> that
    // block + T/2.  Two running scalar offsets.
    auto load_paired = [&](uint32_t (&r)[R][1], const View& v) {
        if constexpr (MULTI) {
            // register 2i + e <-> tile block g + 2i*G (+ G) + e*T/2: window e*NWIN/2 + i / (R/NWIN), and inside it the same
            // running offset as the single-window form, restarted
            constexpr int PER = R / NWIN;
            const uint32_t voff = lane_p | v.dead_mask, step = 2 * G * row_bytes;
#pragma unroll
            for (int w = 0; w < NWIN; ++w) {
                const __amdgpu_buffer_rsrc_t d = window_desc(v.in_base, a.in_rows, v, w);
                uint32_t soff = g * row_bytes;
#pragma unroll
                for (int ii = 0; ii < PER; ++ii) {
                    const int j = 2 * ((w % (NWIN / 2)) * PER + ii) + w / (NWIN / 2);
                    r[j][0] = (a.cache_policy & 1) ? __builtin_amdgcn_raw_buffer_load_b32(d, voff, soff, 2)
                                                   : __builtin_amdgcn_raw_buffer_load_b32(d, voff, soff, 0);
                    soff += step;
                    asm volatile("" : "+s"(soff));
                }
            }
            return;
        }
        const uint32_t rb = gather ? row_bytes >> 1 : row_bytes;  // gather: the half stripes have half the block stride
        const uint32_t voff = (gather ? ((((half * G) << (s - 1)) * a.ld) + c) * 4u : lane_p) | v.dead_mask;
        uint32_t soff = g * rb;
        const uint32_t far = (T / 2) * rb, step = 2 * G * rb;
#define FASTECC_LOAD_LOOP(AUX)                                                                 \
    _Pragma("unroll") for (int j = 0; j < R; j += 2) {                                          \
        r[j][0] = __builtin_amdgcn_raw_buffer_load_b32(v.in, voff, soff, AUX);                  \
        if constexpr (WIDE) r[j + 1][0] = __builtin_amdgcn_raw_buffer_load_b32(v.in_hi, voff, soff, AUX);       \
        else                r[j + 1][0] = __builtin_amdgcn_raw_buffer_load_b32(v.in, voff, soff + far, AUX);    \
        soff += step;                                                                           \
        asm volatile("" : "+s"(soff));                                                          \
    }
        if (a.cache_policy & 1) { FASTECC_LOAD_LOOP(2) } else { FASTECC_LOAD_LOOP(0) }
#undef FASTECC_LOAD_LOOP
    };
    auto store_paired = [&](const uint32_t (&r)[R][1], const View& v) {
        if constexpr (MULTI) {
            constexpr int PER = R / NWIN;
            const uint32_t voff = lane_p | v.dead_mask, step = 2 * G * row_bytes;
#pragma unroll
            for (int w = 0; w < NWIN; ++w) {
                const __amdgpu_buffer_rsrc_t d = window_desc(v.out_base, a.out_rows, v, w);
                uint32_t soff = g * row_bytes;
#pragma unroll
                for (int ii = 0; ii < PER; ++ii) {
                    const int j = 2 * ((w % (NWIN / 2)) * PER + ii) + w / (NWIN / 2);
                    if (a.cache_policy & 2) __builtin_amdgcn_raw_buffer_store_b32(r[j][0], d, voff, soff, 2);
                    else                    __builtin_amdgcn_raw_buffer_store_b32(r[j][0], d, voff, soff, 0);
                    soff += step;
                    asm volatile("" : "+s"(soff));
                }
            }
            return;
        }
        // with fold: block q = g + 2i*G (+ G) (+ T/2) goes to q >> fold; G, T/2 and (for the waves that store) g are multiples of 2^fold
        const uint32_t voff = (fold ? ((half * (G >> fold)) * a.ld + c) * 4u : lane_p) | v.dead_mask;
        uint32_t soff = (g >> fold) * row_bytes;
        const uint32_t far = ((T / 2) >> fold) * row_bytes, step = ((2 * G) >> fold) * row_bytes;
#define FASTECC_STORE_LOOP(AUX)                                                                \
    _Pragma("unroll") for (int j = 0; j < R; j += 2) {                                          \
        __builtin_amdgcn_raw_buffer_store_b32(r[j][0], v.out, voff, soff, AUX);                 \
        if constexpr (WIDE) __builtin_amdgcn_raw_buffer_store_b32(r[j + 1][0], v.out_hi, voff, soff, AUX);      \
        else                __builtin_amdgcn_raw_buffer_store_b32(r[j + 1][0], v.out, voff, soff + far, AUX);   \
        soff += step;                                                                           \
        asm volatile("" : "+s"(soff));                                                          \
    }
        if (a.cache_policy & 2) { FASTECC_STORE_LOOP(2) } else { FASTECC_STORE_LOOP(0) }
#undef FASTECC_STORE_LOOP
    };
    // Change the set of blocks a lane holds: write the registers in one layout, read them back in the other.
    // The caller guarantees that nobody still reads the LDS buffer when this starts.
    auto exchange = [&](uint32_t (&r)[R][1], uint32_t* wbase, uint32_t wq0, uint32_t wstep, const uint32_t* rbase, uint32_t rq0,
                        uint32_t rstep) {
#pragma unroll
        for (int round = 0; round < SPLIT; ++round) {
            const bool mine = SPLIT == 1 || my_round == (uint32_t)round;
            if (mine) {
#pragma unroll
                for (int j = 0; j < R; ++j) wbase[(wq0 + j * wstep) * WS] = r[j][0];
            }
            lds_barrier();
            if (mine) {
#pragma unroll
                for (int j = 0; j < R; ++j) r[j][0] = rbase[(rq0 + j * rstep) * WS];
            }
            if (round + 1 < SPLIT) lds_barrier();  // the buffer is reused by the next column group
        }
    };
    constexpr bool LOAD_A = MODE != MODE_DIT && MODE != MODE_DIT_ROWS && MODE != MODE_MID_UP;  // DIF and MID start in layout A, DIT (and MID's second half alone) in layout B
    auto load_tile = [&](uint32_t (&r)[R][1], const View& v) {
        if constexpr (MODE == MODE_DIF_IMPULSE) {
            // Blocks impulse_rows.. of the tile are zero, impulse_rows <= 16 IMPULSE_MAX: of this wave's 2R blocks g + 16 c only c < IMPULSE_MAX may
            // be non-zero, and what the pair level and the LOGR in-thread levels (strides 512 ... 16) make of block g + 16 t alone is a fixed
            // vector of 64 factors (row_factor[t][g][c]; t = 0: w_1024^(g * bitrev6(c)), a zero partner at every level) — by linearity the
            // 2R words are sum_t block(g + 16 t) * factor_t: one multiply-add per word and block in use instead of six levels.  Both half-waves
            // read the blocks; register j of half-wave h is c = 32 h + j (layout A).
            const uint32_t m = (a.impulse_rows + 15u) / 16u;
#pragma unroll
            for (int j = 0; j < R; ++j) r[j][0] = 0;
#pragma unroll
            for (int t = 0; t < IMPULSE_MAX; ++t) {
                if ((uint32_t)t >= m) break;  // uniform
                uint32_t xt = 0;
                if (g + 16u * t < a.impulse_rows) xt = __builtin_amdgcn_raw_buffer_load_b32(v.in, (c * 4u) | v.dead_mask, (g + 16u * t) * row_bytes, 0);
                const_u32_ptr tab = as_constant(a.row_factor) + ((uint32_t)t * 16u + g) * 64u;
                constexpr int CHI = 16;
#pragma unroll
                for (int j0 = 0; j0 < R; j0 += CHI) {
                    uint32_t tl[CHI], th[CHI];
#pragma unroll
                    for (int i = 0; i < CHI; ++i) tl[i] = tab[j0 + i], th[i] = tab[R + j0 + i];
#pragma unroll
                    for (int i = 0; i < CHI; ++i) r[j0 + i][0] = gf::add(r[j0 + i][0], gf::mul_mont(xt, pair_twiddle<LOGR>(tl[i], th[i], upper_mask)));
                }
            }
            return;
        }
        if constexpr (LOAD_A && PAIR) load_paired(r, v);
        else if constexpr (LOAD_A)    load_rows(r, v, lane_a, qa_u, G);
        else                          load_rows(r, v, lane_b, qb_u, 1);
    };

    // Persistent workgroup (128-KiB tiles only): tiles blockIdx.x, blockIdx.x + gridDim.x, ...  (A next-tile register prefetch
    // existed in rounds 1-2; it measured neutral once the kernels turned out VALU-bound and its variants spilled to scratch.)
    // Only those tiles loop: for the others PERSIST = false makes the body straight-line code, which keeps the compiler from
    // hoisting every level's table address out of a "loop" that runs once (55 SGPR spills in the 1024-block MID tile, round 2).
    constexpr bool PERSIST = C::LDS_BYTES > 80 * 1024;
    uint32_t tile = blockIdx.x;
    if (tile >= a.tiles) return;
    View v = view_of(tile);
    uint32_t x[R][1];
    load_tile(x, v);

    for (;;) {
        const uint32_t next = tile + gridDim.x;
        const bool has_next = next < a.tiles;  // uniform

        const uint32_t off = (g << s) + v.lo;
        const bool compute = !(a.debug & 1u), stores = !(a.debug & 2u);  // uniform; always true outside experiments
        if constexpr (DIFK || MIDK) {
            if constexpr (MODE == MODE_DIF_ROWS) {
                // the split decoder's first pass: every input block times its factor (a zero factor: an erased or unused block).  Factors in TILE
                // order like the gather's below.
                const_u32_ptr f = as_constant(a.row_factor) + (((size_t)(v.hi << s) + v.lo) * G + g) * (2 * R);
#pragma unroll
                for (int j = 0; j < R; j += 2) {
                    const uint32_t f0 = pair_twiddle<LOGR>(f[2 * j + 0], f[2 * j + 1], upper_mask);
                    const uint32_t f1 = pair_twiddle<LOGR>(f[2 * j + 2], f[2 * j + 3], upper_mask);
                    x[j][0] = gf::mul_mont(x[j][0], f0);
                    x[j + 1][0] = gf::mul_mont(x[j + 1][0], f1);
                }
            }
            if constexpr (WIDE && MODE == MODE_DIF) {
                if (gather) {
                    // paired order: register 2i holds tile block g + 2i*G (+ G in the high half-wave), register 2i+1 that
                    // block + T/2; a zero factor (erased block) turns whatever was read into 0
                    // row_factor is stored in TILE order (encode.hip, gather_tile_order): the 2R factors of a wave are contiguous,
                    // [j/2][+T/2][high half-wave], so they arrive with a few wide scalar loads
                    const_u32_ptr f = as_constant(a.row_factor) + (((size_t)(v.hi << s) + v.lo) * G + g) * (2 * R);
#pragma unroll
                    for (int j = 0; j < R; j += 2) {
                        const uint32_t f0 = pair_twiddle<LOGR>(f[2 * j + 0], f[2 * j + 1], upper_mask);
                        const uint32_t f1 = pair_twiddle<LOGR>(f[2 * j + 2], f[2 * j + 3], upper_mask);
                        x[j][0] = gf::mul_mont(x[j][0], f0);
                        x[j + 1][0] = gf::mul_mont(x[j + 1][0], f1);
                    }
                }
            }
            if (compute && MODE != MODE_DIF_IMPULSE && MODE != MODE_MID_UP) {  // (the impulse form's load has done these levels; MID_UP starts after them)
                if constexpr (PAIR) pair_level_dif<LOGR>(x, a.tw_dif, off, sl, upper_mask);
                dif_levels<LOGR, 1, false>(x, a.tw_dif, off, sl);
            }
            if constexpr (MODE != MODE_MID_UP) exchange(x, lds_a, qa_u, G, lds_b, qb_u, 1);
            if constexpr (DIFK) {
                if (compute) {
                    if constexpr (SZ == 1)      dif_levels<LOGR, 1, true, L2>(x, a.tw_dif, 0u, 0);
                    else if constexpr (SZ == 0) dif_levels<LOGR, 1, false, L2>(x, a.tw_dif, v.lo, s);
                    else if (s == 0)            dif_levels<LOGR, 1, true, L2>(x, a.tw_dif, 0u, 0);
                    else                        dif_levels<LOGR, 1, false, L2>(x, a.tw_dif, v.lo, s);
                }
                if (stores) store_rows(x, v, lane_b, qb_u, 1);
            } else {
                // MODE_MID_ADD: the first blocks of the addend are requested before the low levels, each further run of blocks before the
                // arithmetic of the one before (two runs of CHA registers in flight)
                constexpr int CHA = 8;
                uint32_t ya[2][CHA];
                // addend_shift = h > 0: tile block q is block q >> h of the addend buffer — 2^h consecutive registers read the same block (the
                // repeats are cache hits; HBM sees 1 / 2^h of the stripe).  qb_u and the half-wave's R are multiples of 32 >= 2^h.
                auto fetch_addend = [&](uint32_t (&y)[CHA], int k0) {
                    const uint32_t h = a.addend_shift, hmask = (1u << h) - 1u;
                    const uint32_t voff = (lane_b - ((qb_l - (qb_l >> h)) * a.ld) * 4u) | v.dead_mask;
                    uint32_t soff = ((qb_u + k0) >> h) * row_bytes;
#pragma unroll
                    for (int i = 0; i < CHA; ++i) {
                        y[i] = __builtin_amdgcn_raw_buffer_load_b32(v.add, voff, soff, 2);
                        soff += (((uint32_t)(k0 + i + 1) & hmask) == 0u) ? row_bytes : 0u;
                        asm volatile("" : "+s"(soff));
                    }
                };
                if constexpr (MODE == MODE_MID_ADD) fetch_addend(ya[0], 0);  // (MID_UP: the tile's own loads are still in flight here)
                if constexpr (MODE != MODE_MID_UP) dif_levels<LOGR, 1, true, L2>(x, a.tw_dif, 0u, 0);
                if constexpr (MODE == MODE_MID_ADD) {
                    if (a.keep) {  // uniform: fastecc_repair's second chain starts from here (MODE_MID_UP)
                        View vk = v;
                        vk.out = v.keep;
                        store_rows(x, vk, lane_b, qb_u, 1);
                    }
                }
                // position p = hi*T + q holds coefficient bitrev_n(p); dscale is stored in position order, so
                // the R factors of a lane are contiguous.  In a PAIR tile the two half-waves hold different
                // blocks: both runs are fetched (scalar) and selected per lane like the pair-level twiddles.
                // (a batch stores its stripes back to back: v.hi counts tiles across all of them, the factor table repeats)
                const uint32_t stripe_tile = a.dscale_whole ? v.hi : (v.hi & ((1u << (a.n - LOGT)) - 1u));
                const_u32_ptr d = as_constant(a.dscale) + ((size_t)stripe_tile << LOGT) + qb_u;
                constexpr int CH = 8;
#pragma unroll
                for (int k0 = 0; k0 < R; k0 += CH) {
                    uint32_t dl[CH], dh[CH];
#pragma unroll
                    for (int i = 0; i < CH; ++i) dl[i] = d[k0 + i];
                    if constexpr (PAIR) {
#pragma unroll
                        for (int i = 0; i < CH; ++i) dh[i] = d[R + k0 + i];
                    }
#pragma unroll
                    for (int i = 0; i < CH; ++i) {
                        const uint32_t f = PAIR ? pair_twiddle<LOGR>(dl[i], dh[i], upper_mask) : dl[i];
                        x[k0 + i][0] = gf::mul_mont(x[k0 + i][0], f);
                    }
                }
                if constexpr (ADDK) {
                    // + addend[p] * addend_factor[p]: the other half of the split decoder's coefficient vector (layout B: this lane's blocks
                    // qb_u + k (+ R in the high half-wave), the same run of positions as the factors above)
                    const_u32_ptr e = as_constant(a.addend_factor) + ((size_t)v.hi << LOGT) + qb_u;
                    if constexpr (MODE == MODE_MID_UP) fetch_addend(ya[0], 0);
#pragma unroll
                    for (int k0 = 0; k0 < R; k0 += CHA) {
                        const int cur = (k0 / CHA) & 1;
                        if (k0 + CHA < R) fetch_addend(ya[cur ^ 1], k0 + CHA);
                        uint32_t el[CHA], eh[CHA];
#pragma unroll
                        for (int i = 0; i < CHA; ++i) el[i] = e[k0 + i], eh[i] = e[R + k0 + i];
#pragma unroll
                        for (int i = 0; i < CHA; ++i)
                            x[k0 + i][0] = gf::add(x[k0 + i][0], gf::mul_mont(ya[cur][i], pair_twiddle<LOGR>(el[i], eh[i], upper_mask)));
                    }
                }
                dit_levels<LOGR, 1, true, L2>(x, a.tw_dit, 0u, 0);
                lds_barrier();  // every lane has finished reading the first exchange
                exchange(x, lds_b, qb_u, 1, lds_a, qa_u, G);
                if (fold == 0 || (g & ((1u << fold) - 1u)) == 0) {  // uniform: does this wave hold surviving blocks?
                    dit_levels<LOGR, 1, false>(x, a.tw_dit, off, sl);
                    if constexpr (PAIR) {
                        pair_level_dit<LOGR>(x, a.tw_dit, off, sl, upper_mask);
                        store_paired(x, v);
                    } else {
                        store_rows(x, v, lane_a, qa_u >> fold, G >> fold);
                    }
                }
            }
        } else {
            if constexpr (SZ == 1)      dit_levels<LOGR, 1, true, L2>(x, a.tw_dit, 0u, 0);
            else if constexpr (SZ == 0) dit_levels<LOGR, 1, false, L2>(x, a.tw_dit, v.lo, s);
            else if (s == 0)            dit_levels<LOGR, 1, true, L2>(x, a.tw_dit, 0u, 0);
            else                        dit_levels<LOGR, 1, false, L2>(x, a.tw_dit, v.lo, s);
            exchange(x, lds_b, qb_u, 1, lds_a, qa_u, G);
            dit_levels<LOGR, 1, false>(x, a.tw_dit, off, sl);
            if constexpr (MODE == MODE_DIT_ROWS) {
                // the decoder's scatter: only the blocks being rebuilt leave the tile, each times its factor; the others' stores get an
                // offset beyond the descriptor and are dropped.  Factors in tile order (as MODE_DIF_ROWS), paired layout as store_paired.
                pair_level_dit<LOGR>(x, a.tw_dit, off, sl, upper_mask);
                const_u32_ptr f = as_constant(a.row_factor) + (((size_t)(v.hi << s) + v.lo) * G + g) * (2 * R);
                const uint32_t voff = lane_p | v.dead_mask;
                uint32_t soff = g * row_bytes;
                const uint32_t far = (T / 2) * row_bytes, step = 2 * G * row_bytes;
                if constexpr (MULTI) {  // windows as in store_paired: register j's factor is the pair (f[2j], f[2j + 1]) of the two half-waves
                    constexpr int PER = R / NWIN;
#pragma unroll
                    for (int w = 0; w < NWIN; ++w) {
                        const __amdgpu_buffer_rsrc_t dsc = window_desc(v.out_base, a.out_rows, v, w);
                        uint32_t so = g * row_bytes;
#pragma unroll
                        for (int ii = 0; ii < PER; ++ii) {
                            const int j = 2 * ((w % (NWIN / 2)) * PER + ii) + w / (NWIN / 2);
                            const uint32_t fj = pair_twiddle<LOGR>(f[2 * j], f[2 * j + 1], upper_mask);
                            __builtin_amdgcn_raw_buffer_store_b32(gf::mul_mont(x[j][0], fj), dsc, fj ? voff : 0xFFFFFFFFu, so, 0);
                            so += step;
                            asm volatile("" : "+s"(so));
                        }
                    }
                } else
#pragma unroll
                for (int j = 0; j < R; j += 2) {
                    const uint32_t f0 = pair_twiddle<LOGR>(f[2 * j + 0], f[2 * j + 1], upper_mask);
                    const uint32_t f1 = pair_twiddle<LOGR>(f[2 * j + 2], f[2 * j + 3], upper_mask);
                    __builtin_amdgcn_raw_buffer_store_b32(gf::mul_mont(x[j][0], f0), v.out, f0 ? voff : 0xFFFFFFFFu, soff, 0);
                    if constexpr (WIDE) __builtin_amdgcn_raw_buffer_store_b32(gf::mul_mont(x[j + 1][0], f1), v.out_hi, f1 ? voff : 0xFFFFFFFFu, soff, 0);
                    else                __builtin_amdgcn_raw_buffer_store_b32(gf::mul_mont(x[j + 1][0], f1), v.out, f1 ? voff : 0xFFFFFFFFu, soff + far, 0);
                    soff += step;
                    asm volatile("" : "+s"(soff));
                }
            } else if constexpr (PAIR) {
                pair_level_dit<LOGR>(x, a.tw_dit, off, sl, upper_mask);
                store_paired(x, v);
            } else {
                store_rows(x, v, lane_a, qa_u, G);
            }
        }

        if constexpr (!PERSIST) break;
        if (!has_next) break;
        tile = next;
        v = view_of(tile);
        load_tile(x, v);
        lds_barrier();  // the LDS tile is rewritten by the next iteration
    }
}

// ------------------------------------------------------------------------------------------------
// launcher
// ------------------------------------------------------------------------------------------------
template <int LOGT, int LOGR, bool PAIR, int MODE, int SPLIT = 1, int NWIN = 1, int SZ = -1>
static hipError_t launch_one(const TileArgs& a, hipStream_t st)
{
    using C = TileCfg<LOGT, LOGR, PAIR, SPLIT>;
    auto kern = ntt_tile_kernel<LOGT, LOGR, PAIR, MODE, SPLIT, NWIN, SZ>;
    // > 64 KiB of dynamic LDS must be enabled per kernel AND per device; remember which devices are done.  Smaller tiles need no attribute — and
    // the call costs ~0.2 ms per instantiation: the decoder's first fastecc_decode_prepare touches a dozen tile shapes (profiles/r06/prepare_trace.txt)
    static bool configured[64] = {};
    int dev = 0;
    if (C::LDS_BYTES > 64 * 1024 && hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    if (C::LDS_BYTES > 64 * 1024 && (dev < 0 || dev >= 64 || !configured[dev])) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) configured[dev] = true;
    }
    TileArgs b = a;
    b.col_chunks = (a.S + C::W - 1) / C::W;
    uint64_t tiles = ((uint64_t)(a.batch > 1 ? a.batch : 1u) << (a.n - LOGT)) * b.col_chunks;
    if (a.groups) {  // the first block groups only: tile index = group * col_chunks + column chunk in every order but the XCD order 2
        if (a.batch > 1 || (uint64_t)a.groups * b.col_chunks > tiles) return hipErrorInvalidValue;
        tiles = (uint64_t)a.groups * b.col_chunks;
        if (b.xcd_swizzle == 2) b.xcd_swizzle = 0;
    }
    if (tiles == 0 || tiles > 0x7FFFFFFFull) return hipErrorInvalidValue;
    b.tiles = (uint32_t)tiles;
    uint64_t blocks = tiles;
    // Persistent workgroups pay off when only ONE workgroup fits a CU (128 KiB tiles): the grid is sized to
    // the machine and each workgroup walks its tiles.  Smaller tiles leave room for two or more resident
    // workgroups, and the hardware dispatcher interleaving them measured faster (sweep table).
    if (a.persistent_cus > 0 && C::LDS_BYTES > 80 * 1024) {
        // resident workgroups per CU: LDS (160 KiB) and 16 waves (4 per SIMD at <= 128 VGPRs)
        const int by_lds = (160 * 1024) / C::LDS_BYTES, by_waves = (LOGR <= 4 || SPLIT > 1 ? 32 : 16) / C::G;
        const int per_cu = by_lds < by_waves ? (by_lds < 1 ? 1 : by_lds) : (by_waves < 1 ? 1 : by_waves);
        const uint64_t cap = (uint64_t)a.persistent_cus * per_cu;
        if (blocks > cap) blocks = cap;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(C::THREADS), C::LDS_BYTES, st, b);
    return hipGetLastError();
}

template <int LOGT, int LOGR, bool PAIR>
static hipError_t launch_mode(int mode, const TileArgs& a, hipStream_t st)
{
    if (a.wide) {
        // several address windows per tile (blocks spanning up to 2^33 / 2^34 / 2^35 bytes): outer passes of the shapes the plans use
        if constexpr (LOGT == 10 && PAIR && LOGR == 5) {
            // (the two versions of the low levels — s = 0 or not — as two kernels: see SZ)
            if (a.wide == 2 && mode == MODE_DIF) return a.s == 0 ? launch_one<LOGT, LOGR, PAIR, MODE_DIF, 2, 2, 1>(a, st) : launch_one<LOGT, LOGR, PAIR, MODE_DIF, 2, 2, 0>(a, st);
            if (a.wide == 2 && mode == MODE_DIT) return a.s == 0 ? launch_one<LOGT, LOGR, PAIR, MODE_DIT, 2, 2, 1>(a, st) : launch_one<LOGT, LOGR, PAIR, MODE_DIT, 2, 2, 0>(a, st);
        } else if constexpr (LOGR == 4) {
            // (the split decoder's first and last pass on tiles of several windows: 8 ... 64 KB blocks at k = 2^19)
            if (mode == MODE_DIF_ROWS || mode == MODE_DIT_ROWS) {
                const bool down = mode == MODE_DIF_ROWS;
                if (a.wide == 2) return down ? launch_one<LOGT, LOGR, PAIR, MODE_DIF_ROWS, 1, 2>(a, st) : launch_one<LOGT, LOGR, PAIR, MODE_DIT_ROWS, 1, 2>(a, st);
                if (a.wide == 4) return down ? launch_one<LOGT, LOGR, PAIR, MODE_DIF_ROWS, 1, 4>(a, st) : launch_one<LOGT, LOGR, PAIR, MODE_DIT_ROWS, 1, 4>(a, st);
                if (a.wide == 8) return down ? launch_one<LOGT, LOGR, PAIR, MODE_DIF_ROWS, 1, 8>(a, st) : launch_one<LOGT, LOGR, PAIR, MODE_DIT_ROWS, 1, 8>(a, st);
                if constexpr (LOGT == 9) {
                    if (a.wide == 16) return down ? launch_one<LOGT, LOGR, PAIR, MODE_DIF_ROWS, 1, 16>(a, st) : launch_one<LOGT, LOGR, PAIR, MODE_DIT_ROWS, 1, 16>(a, st);
                }
                return hipErrorInvalidValue;
            }
            if (mode == MODE_DIF) {
                if (a.wide == 2) return launch_one<LOGT, LOGR, PAIR, MODE_DIF, 1, 2>(a, st);
                if (a.wide == 4) return launch_one<LOGT, LOGR, PAIR, MODE_DIF, 1, 4>(a, st);
                if (a.wide == 8) return launch_one<LOGT, LOGR, PAIR, MODE_DIF, 1, 8>(a, st);
                if constexpr (LOGT == 9) {
                    if (a.wide == 16) return launch_one<LOGT, LOGR, PAIR, MODE_DIF, 1, 16>(a, st);
                }
            }
            if (mode == MODE_DIT) {
                if (a.wide == 2) return launch_one<LOGT, LOGR, PAIR, MODE_DIT, 1, 2>(a, st);
                if (a.wide == 4) return launch_one<LOGT, LOGR, PAIR, MODE_DIT, 1, 4>(a, st);
                if (a.wide == 8) return launch_one<LOGT, LOGR, PAIR, MODE_DIT, 1, 8>(a, st);
                if constexpr (LOGT == 9) {
                    if (a.wide == 16) return launch_one<LOGT, LOGR, PAIR, MODE_DIT, 1, 16>(a, st);
                }
            }
        }
        return hipErrorInvalidValue;
    }
    if constexpr (LOGT == 10 && PAIR && LOGR == 5) {
        if (a.split2) {  // 1024-block tiles through a 64 KiB buffer: two workgroups per CU, never persistent
            switch (mode) {
                case MODE_DIF: return a.s == 0 ? launch_one<LOGT, LOGR, PAIR, MODE_DIF, 2, 1, 1>(a, st) : launch_one<LOGT, LOGR, PAIR, MODE_DIF, 2, 1, 0>(a, st);
                case MODE_DIT: return a.s == 0 ? launch_one<LOGT, LOGR, PAIR, MODE_DIT, 2, 1, 1>(a, st) : launch_one<LOGT, LOGR, PAIR, MODE_DIT, 2, 1, 0>(a, st);
                case MODE_MID_ADD: return launch_one<LOGT, LOGR, PAIR, MODE_MID_ADD, 2>(a, st);
                case MODE_MID_UP: return launch_one<LOGT, LOGR, PAIR, MODE_MID_UP, 2>(a, st);
                case MODE_DIF_ROWS: case MODE_DIT_ROWS: return hipErrorInvalidValue;
                case MODE_DIF_IMPULSE: return a.s == 0 && a.impulse_rows <= 16u * IMPULSE_MAX ? launch_one<LOGT, LOGR, PAIR, MODE_DIF_IMPULSE, 2, 1, 1>(a, st) : hipErrorInvalidValue;
                default:       return launch_one<LOGT, LOGR, PAIR, MODE_MID, 2>(a, st);
            }
        }
    }
    if constexpr (LOGT == 9 && !PAIR && LOGR == 5) {
        // 512 blocks x 64 words through a 64 KiB buffer (two rounds of 32 columns): 256-byte pieces of a block per request, two workgroups per CU
        if (a.split2 && mode == MODE_DIF) return a.s == 0 ? launch_one<LOGT, LOGR, PAIR, MODE_DIF, 2, 1, 1>(a, st) : launch_one<LOGT, LOGR, PAIR, MODE_DIF, 2, 1, 0>(a, st);
        if (a.split2 && mode == MODE_DIT) return a.s == 0 ? launch_one<LOGT, LOGR, PAIR, MODE_DIT, 2, 1, 1>(a, st) : launch_one<LOGT, LOGR, PAIR, MODE_DIT, 2, 1, 0>(a, st);
    }
    if constexpr (LOGT == 9 && PAIR && LOGR == 5) {
        // 512-block MID tiles through a 32 KiB buffer (two column rounds): four workgroups of eight waves per CU instead of two
        if (a.split2 && mode == MODE_MID) return launch_one<LOGT, LOGR, PAIR, MODE_MID, 2>(a, st);
    }
    // the split decoder's shapes (tile_split_supported): slim outer pair tiles with per-block factors; the addend MID only as above
    if (mode == MODE_MID_ADD || mode == MODE_MID_UP || mode == MODE_DIF_IMPULSE) return hipErrorInvalidValue;
    constexpr bool ROWS_SHAPE = PAIR && (LOGR == 4 || LOGT == 7);  // the outer tiles of the default plans at k = 2^19, 2^18 (slim) and 2^17
    if (mode == MODE_DIF_ROWS) {
        if constexpr (ROWS_SHAPE) return launch_one<LOGT, LOGR, PAIR, MODE_DIF_ROWS>(a, st);
        else return hipErrorInvalidValue;
    }
    if (mode == MODE_DIT_ROWS) {
        if constexpr (ROWS_SHAPE) return launch_one<LOGT, LOGR, PAIR, MODE_DIT_ROWS>(a, st);
        else return hipErrorInvalidValue;
    }
    // (32 values per lane: the two versions of the low levels as two kernels where the single one spills — see SZ)
    constexpr bool TWO = LOGR == 5 && ((LOGT == 10 && PAIR) || (LOGT == 9 && !PAIR));
    switch (mode) {
        case MODE_DIF:
            if constexpr (TWO) return a.s == 0 ? launch_one<LOGT, LOGR, PAIR, MODE_DIF, 1, 1, 1>(a, st) : launch_one<LOGT, LOGR, PAIR, MODE_DIF, 1, 1, 0>(a, st);
            else return launch_one<LOGT, LOGR, PAIR, MODE_DIF>(a, st);
        case MODE_DIT:
            if constexpr (TWO) return a.s == 0 ? launch_one<LOGT, LOGR, PAIR, MODE_DIT, 1, 1, 1>(a, st) : launch_one<LOGT, LOGR, PAIR, MODE_DIT, 1, 1, 0>(a, st);
            else return launch_one<LOGT, LOGR, PAIR, MODE_DIT>(a, st);
        default:
            if constexpr (LOGR == 5) return launch_one<LOGT, LOGR, PAIR, MODE_MID>(a, st);
            else return hipErrorInvalidValue;
    }
}

// Shapes with a two-window (WIDE) DIF/DIT instantiation, see launch_mode.
// Returns the largest window count of the shape (0 = none): see launch_mode.
int tile_max_windows(int logt, bool pair, int logr)
{
    if (!pair) return 0;
    if (logr == 5 && logt == 10) return 2;
    if (logr == 4 && logt == 9) return 16;  // G = R = 16: one register pair per window at most
    if (logr == 4 && logt == 8) return 8;
    return 0;
}
bool tile_wide_supported(int logt, bool pair, int logr) { return tile_max_windows(logt, pair, logr) >= 2; }

// Largest fold a MID tile supports: 2^fold must divide its wave stride G = 2^L2.
int tile_max_fold(int logt, bool pair, int logr) { return logt - logr - (pair ? 1 : 0); }

// Tile shapes that are instantiated.  logr = registers per lane (log2): 5 everywhere; 4 additionally for the
// 8- and 9-level pair tiles of the outer passes (16 words per lane, <= 64 VGPRs, twice the waves per CU).
bool tile_supported(int logt, bool pair, int logr)
{
    if (logr == 5) return pair ? (logt >= 7 && logt <= 10) : (logt >= 6 && logt <= 9);
    if (logr == 4) return pair && (logt == 8 || logt == 9);
    return false;
}

hipError_t launch_tile(int logt, bool pair, int logr, int mode, const TileArgs& a, hipStream_t st)
{
    if (!tile_supported(logt, pair, logr) || a.n < logt) return hipErrorInvalidValue;
    if (a.fold < 0 || (a.fold > 0 && (mode != MODE_MID || a.fold > tile_max_fold(logt, pair, logr)))) return hipErrorInvalidValue;
    if (logr == 4) {
        if (mode == MODE_MID || mode == MODE_MID_ADD || mode == MODE_MID_UP) return hipErrorInvalidValue;
        return logt == 8 ? launch_mode<8, 4, true>(mode, a, st) : launch_mode<9, 4, true>(mode, a, st);
    }
    if (pair) {
        switch (logt) {
            case 7: return launch_mode<7, 5, true>(mode, a, st);
            case 8: return launch_mode<8, 5, true>(mode, a, st);
            case 9: return launch_mode<9, 5, true>(mode, a, st);
            default: return launch_mode<10, 5, true>(mode, a, st);
        }
    }
    switch (logt) {
        case 6: return launch_mode<6, 5, false>(mode, a, st);
        case 7: return launch_mode<7, 5, false>(mode, a, st);
        case 8: return launch_mode<8, 5, false>(mode, a, st);
        default: return launch_mode<9, 5, false>(mode, a, st);
    }
}

void preload_tile_kernels()
{
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(ntt_tile_kernel<9, 5, true, MODE_MID, 2>)) != hipSuccess) (void)hipGetLastError();  // (speed only)
}

}  // namespace fastecc
