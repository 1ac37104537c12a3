// kernels.hpp — launcher interface between the C-ABI host code (api.hip) and the gfx950 kernels.
#pragma once
#include <vector>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fastecc {

// MODE_DIF_ROWS / MODE_MID_ADD: tile passes of the decoder's split transform (tile_kernels.hip; decode.hip "even / odd split"):
// a DIF tile whose input blocks are multiplied by per-block factors (TileArgs::row_factor, tile order), and a MID tile that adds
// addend[p] * addend_factor[p] to block p between its two halves; MODE_DIT_ROWS: a DIT tile that stores only the blocks with a non-zero
// factor (row_factor, tile order), times that factor — the decoder's scatter in its last pass; MODE_DIF_IMPULSE: a 1024-block DIF tile at s = 0
// whose input is zero from block `impulse_rows` <= 16 IMPULSE_MAX on: what its first six levels make of each of the few blocks in use is a
// fixed vector of factors (row_factor = those tables, [IMPULSE_MAX][16][64]), so a multiply-add per word and block replaces them.
constexpr int IMPULSE_MAX = 3;
enum { MODE_DIF = 0, MODE_DIT = 1, MODE_MID = 2, MODE_DIF_ROWS = 3, MODE_MID_ADD = 4, MODE_DIT_ROWS = 5, MODE_DIF_IMPULSE = 6, MODE_MID_UP = 7 };

// Arguments of one register pass (kernels.hip: ntt_pass_kernel).
struct PassArgs {
    const uint32_t* in;      // stripe X[N][S] read by this pass
    uint32_t* out;           // stripe written (may equal `in`)
    const uint32_t* tw_dif;  // Montgomery-form roots for DIF levels: tw[e] = w^e * 2^32 mod p, e < N/2
    const uint32_t* tw_dit;  // same for DIT levels
    const uint32_t* dscale;  // MID only: D[bitrev_n(pos)] in Montgomery form, indexed by position
    uint32_t S;              // words per block
    uint32_t ld;             // words between consecutive blocks in memory (>= S; the row pitch)
    int n;                   // log2 N
    int s;                   // log2 of the smallest stride of this pass
    uint32_t col_chunks;     // filled by the launcher
    uint64_t items;          // filled by the launcher
    int fold;                // MID only: keep every 2^fold-th output block, written compactly (fewer parity than data blocks)
    uint32_t batch;          // > 1: that many stripes stored back to back are transformed by one launch
    uint32_t in_rows;        // > 0: `in` holds only that many blocks, the rest of the stripe reads as zero (zero-extended codes)
    uint32_t out_rows;       // > 0: only that many blocks of the result exist in `out`, the rest is not written
    uint32_t dscale_whole;   // MID with batch > 1: != 0 = `dscale` covers all stripes of the batch (batch * 2^n entries by
                             // position: the q sub-transforms of an order q * 2^n transform), 0 = it repeats per stripe
    // DIF only, optional (the decoder's first pass): input block u is block u/2 of `in` (u even) or of `in_odd` (u odd),
    // multiplied by row_factor[u] (Montgomery form); a zero factor means "erased": the block is not read at all
    const uint32_t* in_odd;
    const uint32_t* row_factor;
};

// Arguments of one LDS-tiled pass (tile_kernels.hip: ntt_tile_kernel); fields as in PassArgs.
struct TileArgs {
    const uint32_t* in;
    uint32_t* out;
    const uint32_t* tw_dif;
    const uint32_t* tw_dit;
    const uint32_t* dscale;
    uint32_t S;           // words per block covered by this launch (a column slab may be narrower than a block)
    uint32_t ld;          // words between consecutive blocks in memory (the full block size)
    int n;
    int s;
    uint32_t col_chunks;  // filled by the launcher
    uint32_t tiles;       // filled by the launcher: tiles in this pass (persistent workgroups loop over them)
    uint32_t debug;       // ablation switches (profiles/r01/ablation_dif_tiles.md); always 0 in the product
    int persistent_cus;   // > 0: 128-KiB tiles run as persistent workgroups sized for that many CUs
    bool split2;          // 1024-block pair tiles exchange 16 columns at a time (64 KiB LDS, 2 workgroups per CU)
    int cache_policy;     // bit 0: non-temporal stripe loads, bit 1: non-temporal stripe stores
    const uint32_t* in_odd;      // wide DIF tiles, optional: as PassArgs::in_odd / row_factor (the decoder's first pass)
    const uint32_t* row_factor;
    uint32_t batch;       // > 1: that many stripes stored back to back are transformed by one launch
    uint32_t in_rows;     // > 0: `in` holds only that many blocks, the rest reads as zero (buffer bounds check does it)
    uint32_t out_rows;    // > 0: only that many blocks exist in `out`, stores beyond are dropped (same mechanism)
    int wide;             // DIF/DIT pair tiles whose blocks span >= 2^32 bytes: address windows per tile (2, 4, 8; 0 = one)
    int fold;             // MID only: keep the blocks whose position is a multiple of 2^fold, stored at position >> fold
    int xcd_swizzle;      // 0 off, 1 contiguous column chunks per XCD, 2 whole block groups per XCD (workgroup b -> XCD b % 8)
    uint32_t dscale_whole;  // as PassArgs::dscale_whole
    const uint32_t* addend;         // MODE_MID_ADD: a stripe in the position order MID's first half leaves (block p = coefficient bitrev(p))
    const uint32_t* addend_factor;  // ... and its per-position factors (Montgomery form), laid out like dscale
    uint32_t addend_shift;          // 0..5: block p of the addend is block p >> addend_shift of the buffer (every 2^shift positions share one block:
                                    // the transform of a stripe that is zero outside the multiples of 2^shift, stored once — decode.hip)
    uint32_t groups;                // > 0: only the first `groups` block groups of the pass are run (MODE_DIF_ROWS: the others are known to be zero)
    uint32_t* keep;                 // MODE_MID_ADD, optional: the tile after MID's first half (before any factor) is also stored here — MODE_MID_UP's input
    uint32_t impulse_rows;          // MODE_DIF_IMPULSE: blocks [impulse_rows, T) of every tile are zero and not read (<= 16 IMPULSE_MAX)
};

// Arguments of the odd-radix pass of a transform of order q * 2^m (mixed_kernels.hip: radix_kernel).
struct RadixArgs {
    const uint32_t* in;   // stripe of q * M blocks read by this pass
    uint32_t* out;        // stripe written (may equal `in`: a wave reads and writes the same q rows)
    const uint32_t* dft;  // constants of the q-point transform (radix_dft_table), Montgomery form
    const uint32_t* tw;   // M x (q-1): w_(q*M)^(+-i2*j), j = 1..q-1, Montgomery form
    uint32_t S;           // words per block
    uint32_t ld;          // words between consecutive blocks
    uint32_t M;           // 2^m: the block distance of the radix-q butterflies
    uint32_t in_rows;     // > 0: `in` holds only that many blocks, the rest reads as zero
    uint32_t out_rows;    // > 0: only that many blocks of the result are written
    uint32_t col_chunks;  // filled by the launcher
    uint64_t items;       // filled by the launcher
};
// Arguments of the odd-radix level fused with the outermost power-of-two tile (mixed_kernels.hip: fused_radix_kernel).
struct FusedArgs {
    const uint32_t* in;
    uint32_t* out;
    const uint32_t* dft;  // radix_dft_table of the direction
    const uint32_t* tw;   // M x (q-1) twiddles of the odd-radix level (as in RadixArgs)
    const uint32_t* twl;  // level-packed twiddles of the power-of-two levels (DIF or DIT table of the plan)
    uint32_t S, ld, M;
    uint32_t in_rows, out_rows;
    int s;                // the tile covers levels [s, s + A) of the size-M transforms
    uint32_t col_chunks;  // filled by the launcher
};
// the register run length the fused kernel uses for (q, A levels), 0 if that shape is not instantiated
int fused_rlog(int q, int levels);
hipError_t launch_fused(int q, int levels, bool dit, FusedArgs a, hipStream_t st);
bool fused_batch_fits(uint64_t ld, uint64_t S, uint64_t rows);  // can the fused kernel address this batch (one descriptor, 32-bit offsets)?
bool radix_supported(int q);
std::vector<uint32_t> radix_dft_table(int q, uint32_t wq);  // host: wq = the primitive q-th root of the direction
hipError_t launch_radix(int q, bool dit, int vec, RadixArgs a, hipStream_t st);

hipError_t launch_pass(int logr, int vec, int mode, PassArgs a, hipStream_t st);
// The code objects of the two engine translation units (kernels.hip, tile_kernels.hip) are loaded by the runtime at the first launch of one of
// their kernels — 1.3 ms for kernels.hip, found inside the first fastecc_decode_prepare (profiles/r06/prepare_trace_warm.txt).  These load them
// at the first fastecc_create on a device instead (once per process and device).
void preload_pass_kernels();
void preload_tile_kernels();
bool tile_supported(int logt, bool pair, int logr = 5);
int tile_max_fold(int logt, bool pair, int logr = 5);
bool tile_wide_supported(int logt, bool pair, int logr = 5);
int tile_max_windows(int logt, bool pair, int logr = 5);
hipError_t launch_tile(int logt, bool pair, int logr, int mode, const TileArgs& a, hipStream_t st);
hipError_t launch_bitrev_rows(uint32_t* data, uint32_t S, int n, int vec, hipStream_t st);
hipError_t launch_scale_rows(uint32_t* data, const uint32_t* factor, uint32_t S, uint64_t rows, int vec, hipStream_t st);
hipError_t launch_count_out_of_range(const uint32_t* x, uint64_t count, unsigned long long* bad, hipStream_t st);
// pack_kernels.hip: GF.md:72-104 recoding, `words` <= 1024 raw words per block <-> words + 1 packed words at row pitch ld
hipError_t launch_pack_blocks(const uint32_t* raw, uint32_t* packed, uint32_t words, uint32_t ld, uint64_t blocks, hipStream_t st);
hipError_t launch_unpack_blocks(const uint32_t* packed, uint32_t* raw, uint32_t words, uint32_t ld, uint64_t blocks,
                                unsigned long long* bad_blocks, hipStream_t st);
hipError_t launch_gf_binary(int op, const uint32_t* x, const uint32_t* y, uint32_t* out, uint64_t count, hipStream_t st);

}  // namespace fastecc
