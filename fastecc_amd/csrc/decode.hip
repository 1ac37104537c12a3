// decode.hip — erasure decoding, the "fastest" scheme of README.md:102-119 / RS.md:42-79, for every code of this library.
//
// The reference documents this algorithm and does not implement it.  For the reference's (2k,k) code the codeword is f
// on the 2k-th roots of unity: position u <-> point w^u (w = w_2k), even positions are the data blocks (u = 2i), odd
// ones the parity blocks (u = 2j+1, RS.cpp:51-54).  With E the erased positions (|E| <= k) and
// l(x) = prod_{e in E} (x - w^e):
//
//   p = f * l has degree < 2k and KNOWN values everywhere: c[u] * l(w^u) at surviving positions, 0 at erased ones;
//   p'(w^e) = f(w^e) * l'(w^e) at an erased position, so  f(w^e) = [x p'(x)](w^e) / (w^e * l'(w^e)).
//
// x p'(x) = sum m p_m x^m needs no coefficient shift, which makes the data-parallel part the SAME pipeline as the
// encoder one size up: inverse transform of size 2k, block holding coefficient m times m / 2k, forward transform —
// i.e. create_transform_ctx(2k, factor[m] = m / 2k) with fold = 1, because only the even (data) positions are wanted.
// Around it: a gather (codeword blocks times l(w^u), zeros at erasures; fused into the transform's first pass for the
// (2k,k) layout) and one pass that multiplies the recovered rows by 1 / (w^e l'(w^e)).
//
// Even / odd split (codes with n <= 2k on power-of-two orders, k >= 2^18; option "decode_split"): recovering e data blocks takes e parity
// blocks, so the other surviving parity blocks may count as erased too (ST_UNUSED: roots of l like the lost ones).  With q~ = DIF_k(data * l)
// and r~ = DIF_k(parity * l), unnormalised inverse transforms of k points over the even and the odd positions, the 2k coefficients are
// P[m] = (q~[m] + w^-m r~[m]) / 2k and P[m+k] = (q~[m] - w^-m r~[m]) / 2k, and because w^(2j(m+k)) = w^(2jm) the values of x p'(x) at the data
// positions are the k-point forward transform of  g[m] = m P[m] + (m+k) P[m+k] = (2m+k)/2k q~[m] - 1/2 w^-m r~[m].  So the 2k-point
// pipeline becomes: the encoder's own three passes over the data half (blocks times l(w^2i) on the way in, g's second term added between
// the halves of MID, only the rebuilt blocks stored on the way out) plus r~ — the first pass over the few parity block groups in use, and
// the low levels over a stripe that is zero elsewhere (run_split_decode in encode.hip; tile modes in tile_kernels.hip).
//
// The other codes are the same thing on the (k << e)-th roots of unity (fastecc_decode_prepare): positions that hold no
// block of the code count as erased, zero-extended data blocks as known zeros, the transform has fold = e.
//
// Everything that depends only on the erasure PATTERN is done once in fastecc_decode_prepare, on the device: the locator by
// a product tree whose every level is ONE batch of cyclic products through the library's own transforms (all polynomials
// of a level side by side as the word columns of a stripe), its values and its derivative's values by one transform of a
// two-column stripe, the inverses by Fermat powers.  The host only classifies the positions (one pass over the flags).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "gf.hpp"
#include "gf61_path.hpp"
#include "internal.hpp"
#include "ntt_device.hpp"

namespace fastecc {

struct DecodeState {
    fastecc_ctx* transform = nullptr;  // size-2k transform context, fold 1
    fastecc_ctx* pattern_ntt = nullptr;  // same length, 2 words per block: l and l' are evaluated on the device
    uint32_t* pattern_buf = nullptr;     // its stripe
    uint32_t* fin = nullptr;           // 2k factors by codeword position: l(w^u) (Montgomery) or 0 if erased
    uint32_t* fin_first_pass = nullptr;  // the same in the order the transform's first pass reads them (may equal fin)
    uint32_t* srcmap = nullptr;        // per codeword position: the block that sits there (row, bit 31 = parity stripe)
    uint32_t* gout = nullptr;          // k factors by data block: 1 / (w^2i l'(w^2i)) (Montgomery) if erased, else 0
    // fastecc_repair of the (2k,k) layout in ONE transform: the same x p'(x) evaluated at ALL 2k positions (fold 0) gives the lost parity
    // blocks as well, f(w^u) = (x p')(w^u) / (w^u l'(w^u)) at odd u — instead of decoding the data and encoding it once more
    fastecc_ctx* transform_full = nullptr;
    uint32_t* gout_par = nullptr;       // k factors by parity block: 1 / (w^(2q+1) l'(w^(2q+1))) (Montgomery) if erased, else 0
    uint32_t* recovered_full = nullptr; // 2k blocks: x p'(x) at every position (lazy: 4 GiB at the headline size)
    bool full_ok = false;               // transform_full's first pass reads the factors in the same order as transform's
    uint32_t* recovered = nullptr;     // k blocks: x p'(x) at the data positions
    // (2k,k) layout, the transform as two half-size ones ("even / odd split" below)
    fastecc_ctx* split = nullptr;           // k blocks, per-block factor (2m + k) / 2k
    uint32_t* split_order = nullptr;        // k words: the block each slot of its first pass holds (gather_tile_order)
    uint32_t* split_rows_data = nullptr;    // k words, that order: l(w^2i) of the surviving data blocks (0: lost)
    uint32_t* split_rows_parity = nullptr;  // k words, that order: l(w^(2i+1)) of the parity blocks in use (0: lost or unused)
    uint32_t* split_rows_out = nullptr;     // k words, that order: gout of the block (0: not lost)
    uint32_t* split_pos_parity = nullptr;   // k words by position: -w^(-m) / 2 at position bitrev(m)
    uint32_t* split_impulse = nullptr;      // [IMPULSE_MAX][16][64]: what six DIF levels make of a lone block of a 1024-block tile (run_split_decode)
    uint32_t* split_r1 = nullptr;           // k blocks: the parity half after its first pass (zero outside the groups in use)
    uint32_t* split_r2 = nullptr;           // k blocks: ... after all DIF levels
    uint32_t* split_q2 = nullptr;           // fastecc_repair: k blocks, the data chain's MID output while the top-level result serves the parity chain (lazy)
    uint32_t* split_pos_data_odd = nullptr;   // k words by position: -w^m / 2 at position bitrev(m) (the parity chain's factor of the data half)
    uint32_t* split_rows_out_parity = nullptr;  // k words, first-pass order: gout_par of the block (0: not lost)
    bool split_repair_ready = false;        // this pattern's lost parity blocks can come from the split transform too
    uint32_t* split_r0 = nullptr;           // codes with fewer parity blocks (fold > 0): parity block j copied to its place j << fold of a k-block stripe (lazy)
    uint32_t split_groups = 0;              // block groups of the parity stripe this pattern reads
    // "small" form of the parity half: the parity blocks in use are those at multiples of 2^split_shift of the parity half (1 <= shift <= 5), so r~
    // is the transform of (k >> shift) rows, each result block standing for 2^shift positions: no k-block stripes r1 / r2, no impulse pass
    uint32_t split_shift = 0;
    fastecc_ctx* split_small[6] = {};       // stand-alone transform contexts of k >> shift blocks (lazy, by shift)
    uint32_t* split_small_buf = nullptr;    // (k >> shift) blocks: those rows times l, then transformed in place
    uint64_t split_small_blocks = 0;
    uint32_t split_dirty = 0;               // groups of split_r1 that may hold non-zero rows
    bool split_ready = false;               // this pattern decodes through the split transform
    bool split_unavailable = false;         // it could not be built on this context (plan shape, memory): the 2k-point transform serves
    uint32_t* parity_dev = nullptr;    // staging for FASTECC_MEM_HOST calls (lazy)
    // FASTECC_MEM_HOST: only the rebuilt blocks travel back — their row numbers (host, and a device copy), valid for pattern `host_lists_of`
    std::vector<uint32_t> host_lost_data, host_lost_parity;
    std::vector<uint32_t> host_parity_used;  // few losses: the parity blocks the direct path reads (all it needs staged of a host parity stripe)
    uint64_t pattern_serial = 0, host_lists_of = ~0ull;
    uint32_t* lost_rows_dev = nullptr;  // the two lists back to back
    uint64_t lost_rows_cap = 0;
    uint32_t* pack_dev = nullptr;       // the rebuilt blocks, packed
    uint64_t pack_words = 0;
    uint32_t* pack_host = nullptr;      // pinned landing buffer of that copy (kept between calls; the blocks go to their places from here)
    uint64_t pack_host_words = 0;
    // fastecc_decode_prepare's device state (lazy): the product tree of the locator
    uint64_t tree_T = 0;                   // padded number of roots: the smallest power of two >= the most losses a code tolerates
    std::vector<fastecc_ctx*> tree_ctx;    // level k (polynomials of degree d = 2^k): transforms of length 2d, T/d columns
    fastecc_ctx* tree_top = nullptr;       // few-column levels (chunk_transform_kernel): the upper row bits, 2T / CHUNK rows of CHUNK words
    bool pattern_narrow = false;           // pattern_ntt is such a context too (2 NC / CHUNK rows)
    uint32_t* tree_x = nullptr;            // 2T words: the level's polynomials, [coefficient][polynomial]
    uint32_t* tree_f = nullptr;            // 2T words: their transforms
    uint32_t* tree_y = nullptr;            // 2T words: the next level's polynomials (swaps roles with tree_x)
    uint32_t* tree_p = nullptr;            // 2T words: pairwise products
    uint32_t* wpow = nullptr;              // NC words: w^u (plain)
    uint32_t* roots = nullptr;             // T words: the erased points, zero-padded
    uint32_t* dev_state = nullptr;         // NC bytes (as words/4): LOST / HELD / ZERO per position
    uint32_t* dev_erased = nullptr;        // T words: erased positions
    uint8_t* dev_present = nullptr;        // (2k,k) layout: the caller's two presence arrays, k bytes each (the pattern is scanned on the device)
    uint32_t* dev_counts = nullptr;        // ... and what presence_counts_kernel counts in them (8 words)
    int tree_low = 0;                      // levels below this one are done by tree_low_levels_kernel (0: the per-thread leaves of degree 2^LEAF_LOG)
    uint32_t* tile_order = nullptr;        // NC words: first-pass order of the factors (only for the (2k,k) layout)
    bool tile_order_valid = false;
    // fastecc_repair: which parity blocks are lost (one word each), and the stripe the re-encode writes to
    uint32_t* parity_lost = nullptr;
    uint32_t* parity_again = nullptr;
    uint64_t erased_parity = 0;
    uint64_t erased_data = 0, erased_total = 0;
    // few losses (any layout: the reference's, zero extension, sub-/extra cosets, mixed radix): every lost data block is a fixed linear
    // combination of the surviving data blocks and as many surviving parity blocks (interpolation on N nodes); for repair the lost parity
    // blocks follow from the complete data (direct.hip)
    DirectPass* direct_data = nullptr;
    DirectPass* direct_parity = nullptr;
    DirectPass* direct_both = nullptr;   // data AND parity lost: fastecc_repair's single pass (the lost parity blocks as further outputs on direct_data's nodes)
    bool sub_both = false;               // ... is built for this pattern
    bool sub_only_both = false;          // ... and is the only pass (few outputs: the pass is bound by the read of the survivors, fastecc_decode runs it too and drops the parity outputs)
    int sub_lost_data = 0, sub_lost_parity = 0;
    bool sub = false;
    int direct_kernel = 0;             // 0 choose, 1 VALU, 2 MFMA (option "direct_kernel")
    uint64_t positions = 0;            // code length on the roots of unity: k << log2(n / k) rounded up to powers of two
    bool mixed = false;                // mixed-radix code: `recovered` is the whole work stripe (all positions), transformed in place
    bool standard = false;             // the reference's (2k,k) layout: position u = data u/2 or parity u/2, every block in memory
    bool ready = false;
};

void destroy_decode_state(DecodeState* d)
{
    if (!d) return;
    if (d->transform) fastecc_destroy(d->transform);
    if (d->transform_full) fastecc_destroy(d->transform_full);
    if (d->split) fastecc_destroy(d->split);
    for (fastecc_ctx* sc : d->split_small)
        if (sc) fastecc_destroy(sc);
    if (d->split_small_buf) (void)hipFree(d->split_small_buf);
    for (uint32_t* b : {d->split_order, d->split_rows_data, d->split_rows_parity, d->split_rows_out, d->split_pos_parity, d->split_impulse, d->split_r1, d->split_r2, d->split_r0, d->split_q2, d->split_pos_data_odd,
                        d->split_rows_out_parity})
        if (b) (void)hipFree(b);
    if (d->gout_par) (void)hipFree(d->gout_par);
    if (d->dev_present) (void)hipFree(d->dev_present);
    if (d->dev_counts) (void)hipFree(d->dev_counts);
    if (d->recovered_full) (void)hipFree(d->recovered_full);
    if (d->pattern_ntt) fastecc_destroy(d->pattern_ntt);
    if (d->pattern_buf) (void)hipFree(d->pattern_buf);
    if (d->fin_first_pass && d->fin_first_pass != d->fin) (void)hipFree(d->fin_first_pass);
    if (d->fin) (void)hipFree(d->fin);
    if (d->srcmap) (void)hipFree(d->srcmap);
    if (d->gout) (void)hipFree(d->gout);
    if (d->recovered) (void)hipFree(d->recovered);
    if (d->parity_dev) (void)hipFree(d->parity_dev);
    if (d->lost_rows_dev) (void)hipFree(d->lost_rows_dev);
    if (d->pack_dev) (void)hipFree(d->pack_dev);
    if (d->pack_host) (void)hipHostFree(d->pack_host);
    for (fastecc_ctx* t : d->tree_ctx)
        if (t) fastecc_destroy(t);
    if (d->tree_top) fastecc_destroy(d->tree_top);
    direct_pass_free(d->direct_data);
    direct_pass_free(d->direct_parity);
    direct_pass_free(d->direct_both);
    for (uint32_t* b : {d->parity_lost, d->parity_again, d->tree_x, d->tree_f, d->tree_y, d->tree_p, d->wpow, d->roots, d->dev_state, d->dev_erased, d->tile_order})
        if (b) (void)hipFree(b);
    delete d;
}

namespace {

// ------------------------------------------------------------------------------------------------
// fastecc_decode_prepare on the device.  All values are plain representatives unless a table is consumed by
// gf::mul_mont, in which case it is stored in Montgomery form (x * 2^32 mod p = gf::mul(x, MONT_ONE)).
// ------------------------------------------------------------------------------------------------
// ST_UNUSED: a surviving parity block the split transform does not read — a root of the locator like a lost one, but nothing to rebuild
enum : uint32_t { ST_LOST = 0, ST_HELD = 1, ST_ZERO = 2, ST_UNUSED = 3 };
constexpr int LEAF_LOG = 4, LEAF = 1 << LEAF_LOG;
constexpr int TREE_LOW = 10;  // tall trees: the polynomials of 2^TREE_LOW roots come from one kernel (schoolbook products in LDS) instead of six more levels  // the lowest levels of the tree are one schoolbook kernel: 16 roots per thread

__device__ __forceinline__ uint32_t dev_pow(uint32_t x, uint32_t e)
{
    uint32_t r = 1;
    for (; e; e >>= 1) {
        if (e & 1u) r = gf::mul(r, x);
        x = gf::mul(x, x);
    }
    return r;
}

// wpow[u] = w^u
__global__ __launch_bounds__(256) void wpow_kernel(uint32_t* __restrict__ wpow, uint32_t w, uint32_t count)
{
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u < count) wpow[u] = dev_pow(w, u);
}

// (2k,k) layout: position u is data block u / 2 (u even) or parity block u / 2 (bit 31) — the block map of the table-driven gather, which
// serves the plans whose first pass cannot read the two stripes itself
__global__ __launch_bounds__(256) void standard_srcmap_kernel(const uint8_t* __restrict__ state, uint32_t NC, uint32_t* __restrict__ srcmap)
{
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u < NC) srcmap[u] = state[u] == ST_HELD ? ((u >> 1) | ((u & 1u) << 31)) : 0u;
}

// erased[] = the positions whose state is LOST or UNUSED, in any order; *counter (zero on entry) ends as their number.  A thread takes 16
// positions (one 16-byte load of the state), a workgroup 4096: ONE atomic per workgroup (one per wave measured 187 us at NC = 2^20 — 16384
// atomics on one address).
__global__ __launch_bounds__(256) void erased_list_kernel(const uint8_t* __restrict__ state, uint32_t NC, uint32_t* __restrict__ erased, uint32_t* __restrict__ counter)
{
    __shared__ uint32_t wave_sum[4], block_base;
    const uint32_t u0 = (blockIdx.x * blockDim.x + threadIdx.x) * 16u;
    uint32_t bits = 0;
    if (u0 + 16u <= NC) {
        const uint4 v = *reinterpret_cast<const uint4*>(state + u0);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t st = (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
            bits |= (uint32_t)(st == ST_LOST || st == ST_UNUSED) << i;
        }
    } else {
        for (uint32_t i = 0; i < 16u && u0 + i < NC; ++i) bits |= (uint32_t)(state[u0 + i] == ST_LOST || state[u0 + i] == ST_UNUSED) << i;
    }
    const uint32_t mine = (uint32_t)__builtin_popcount(bits), lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = mine;  // inclusive prefix sum over the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d);
        if (lane >= (uint32_t)d) incl += t;
    }
    if (lane == 63u) wave_sum[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
        block_base = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    uint32_t at = block_base + incl - mine;
    for (uint32_t w = 0; w < wave; ++w) at += wave_sum[w];
    for (uint32_t b = bits; b; b &= b - 1u) erased[at++] = u0 + (uint32_t)__builtin_ctz(b);
}

// The reference's (2k,k) layout, pattern scan on the device: counts[0] = lost data blocks, [1] = lost parity blocks, [2 + h] = surviving parity
// blocks at multiples of 2^h of the parity half (h = 1..5: the split transform's "small form" reads those alone).
__global__ __launch_bounds__(256) void presence_counts_kernel(const uint8_t* __restrict__ pd, const uint8_t* __restrict__ pp, uint32_t N, uint32_t* __restrict__ counts)
{
    __shared__ uint32_t acc[7];
    if (threadIdx.x < 7) acc[threadIdx.x] = 0;
    __syncthreads();
    uint32_t c[7] = {};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const uint32_t held_p = pp[i] != 0;
        c[0] += pd[i] == 0;
        c[1] += 1u - held_p;
        const int tz = __builtin_ctz(i | 32u);
#pragma unroll
        for (int h = 1; h <= 5; ++h) c[1 + h] += held_p & (uint32_t)(tz >= h);
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        uint32_t v = c[j];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
        if ((threadIdx.x & 63u) == 0 && v) atomicAdd(&acc[j], v);
    }
    __syncthreads();
    if (threadIdx.x < 7 && acc[threadIdx.x]) atomicAdd(&counts[threadIdx.x], acc[threadIdx.x]);
}

// ... and the per-position state from the two presence arrays: position 2i = data block i, 2i + 1 = parity block i; surviving parity blocks
// off the multiples of `unused_mask` + 1 are UNUSED (roots of the locator like lost ones).  parity_lost[i] = 1 for a lost parity block.
__global__ __launch_bounds__(256) void standard_state_kernel(const uint8_t* __restrict__ pd, const uint8_t* __restrict__ pp, uint32_t N, uint32_t unused_mask,
                                                             uint8_t* __restrict__ state, uint32_t* __restrict__ parity_lost)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const uint32_t held_p = pp[i] != 0;
    const uint32_t sd = pd[i] != 0 ? ST_HELD : ST_LOST;
    const uint32_t sp = !held_p ? ST_LOST : (i & unused_mask) ? ST_UNUSED : ST_HELD;
    reinterpret_cast<uint16_t*>(state)[i] = (uint16_t)(sd | (sp << 8));
    parity_lost[i] = 1u - held_p;
}

// roots[i] = w^erased[i] for i < n_erased, 0 for the padding up to T (a factor x: it only shifts the locator)
__global__ __launch_bounds__(256) void roots_kernel(uint32_t* __restrict__ roots, const uint32_t* __restrict__ erased,
                                                    const uint32_t* __restrict__ wpow, uint32_t n_erased, uint32_t T)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < T) roots[i] = i < n_erased ? wpow[erased[i]] : 0u;
}

// Leaves: polynomial p = prod_{j < leaf} (x - roots[p*leaf + j]), monic of degree `leaf`; its other coefficients go to
// x[i * m + p], i < leaf (m = T / leaf polynomials side by side).
__global__ __launch_bounds__(256) void leaf_products_kernel(const uint32_t* __restrict__ roots, uint32_t* __restrict__ x, uint32_t leaf, uint32_t m)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= m) return;
    uint32_t c[LEAF + 1];
#pragma unroll
    for (int i = 0; i <= LEAF; ++i) c[i] = i == 0 ? 1u : 0u;
    for (uint32_t j = 0; j < leaf; ++j) {
        const uint32_t r = roots[p * leaf + j];
#pragma unroll
        for (int i = LEAF; i >= 1; --i) c[i] = gf::sub(c[i - 1], gf::mul(r, c[i]));  // c <- c * (x - r)
        c[0] = gf::sub(0u, gf::mul(r, c[0]));
    }
    for (uint32_t i = 0; i < leaf; ++i) x[i * m + p] = c[i];
}

// The lowest LOW levels of the product tree in ONE kernel (the per-level form costs six launches of 5-9 us per level whatever the size):
// a workgroup takes 2^LOW roots and multiplies the monic polynomials pairwise in LDS by the schoolbook rule, degree 1 -> 2 -> ... -> 2^LOW
// ((x^d + a)(x^d + b) = x^2d + x^d (a + b) + a b; 2^(LOW-1) (2^LOW - 1) products per workgroup, 524 K at LOW = 10), and writes the result where
// level LOW of the tree expects it: coefficient i of polynomial p at x[i * m + p], m = T >> LOW (the upper half of the 2 * 2^LOW rows is zero).
template <int LOW>
__global__ __launch_bounds__(1 << LOW) void tree_low_levels_kernel(const uint32_t* __restrict__ roots, uint32_t* __restrict__ x, uint32_t m)
{
    constexpr uint32_t R = 1u << LOW;  // one thread per coefficient
    __shared__ uint32_t buf[2][R];
    const uint32_t p = blockIdx.x, o = threadIdx.x;
    buf[0][o] = gf::sub(0u, roots[p * R + o]);  // x - r
    __syncthreads();
    int cur = 0;
    for (uint32_t d = 1; d < R; d <<= 1) {
        const uint32_t t = o & (2u * d - 1u);
        const uint32_t* a = buf[cur] + (o - t);
        const uint32_t* b = a + d;
        const uint32_t lo_i = t >= d ? t - d + 1u : 0u, hi_i = t < d ? t : d - 1u;
        // the products are summed as exact integers (96 bits: at most 2^(LOW-1) terms below 2^64) and reduced once per coefficient
        uint64_t lo = t >= d ? (uint64_t)a[t - d] + b[t - d] : 0ull;
        uint32_t hi = 0;
        uint32_t i = lo_i;
        for (; i + 3u <= hi_i; i += 4u) {  // four products per turn: the LDS reads of a turn are in flight together
            const uint32_t a0 = a[i], a1 = a[i + 1], a2 = a[i + 2], a3 = a[i + 3];
            const uint32_t b0 = b[t - i], b1 = b[t - i - 1], b2 = b[t - i - 2], b3 = b[t - i - 3];
            uint64_t pr = (uint64_t)a0 * b0;
            lo += pr, hi += lo < pr;
            pr = (uint64_t)a1 * b1;
            lo += pr, hi += lo < pr;
            pr = (uint64_t)a2 * b2;
            lo += pr, hi += lo < pr;
            pr = (uint64_t)a3 * b3;
            lo += pr, hi += lo < pr;
        }
        for (; i <= hi_i; ++i) {  // (t = 2d - 1: lo_i > hi_i, no product)
            const uint64_t pr = (uint64_t)a[i] * b[t - i];
            lo += pr, hi += lo < pr;
        }
        const uint32_t l0 = (uint32_t)lo, l1 = (uint32_t)(lo >> 32);
        uint32_t acc = gf::add(l0 >= gf::P ? l0 - gf::P : l0, gf::mul(l1 >= gf::P ? l1 - gf::P : l1, gf::MONT_ONE));  // 2^32 mod p
        acc = gf::add(acc, gf::mul(hi, gf::MONT_R2));                                                              // 2^64 mod p
        buf[cur ^ 1][o] = acc;
        __syncthreads();
        cur ^= 1;
    }
    x[(size_t)o * m + p] = buf[cur][o];
}

// y[i][q] = f[i][2q] * f[i][2q+1] * scale (rows of pitch m, m/2 results per row); scale = 1 / (2d) in Montgomery form
__global__ __launch_bounds__(256) void pointwise_pairs_kernel(const uint32_t* __restrict__ f, uint32_t* __restrict__ y, uint32_t m, uint64_t total,
                                                              uint32_t scale_mont)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const uint32_t half = m >> 1;
    const uint64_t i = t / half;
    const uint32_t q = (uint32_t)(t - i * half);
    const uint2 v = *reinterpret_cast<const uint2*>(f + i * m + 2 * q);
    y[i * m + q] = gf::mul_mont(gf::mul(v.x, v.y), scale_mont);
}

// (x^d + a)(x^d + b) = x^2d + x^d (a + b) + a b: the next level's polynomials from the cyclic products a b (y, rows of
// pitch m) and this level's a, b (xold, [d][m]); xnew is [4d][m/2] with the upper 2d rows zero (room for the next product)
__global__ __launch_bounds__(256) void combine_kernel(const uint32_t* __restrict__ y, const uint32_t* __restrict__ xold, uint32_t* __restrict__ xnew,
                                                      uint32_t d, uint32_t m, uint64_t total, bool top)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const uint32_t half = m >> 1;
    const uint64_t i = t / half;
    const uint32_t q = (uint32_t)(t - i * half);
    uint32_t v = 0;
    if (i < 2ull * d) {
        v = y[i * m + q];
        if (i >= d) v = gf::add(v, gf::add(xold[(i - d) * m + 2 * q], xold[(i - d) * m + 2 * q + 1]));
    } else if (top) {
        return;  // the last level has no upper half
    }
    xnew[i * half + q] = v;
}

// ---- transforms of FEW columns (the locator's own: 2; the tree's top levels: 2, 4, 8 polynomials side by side) ----
// The stripe kernels give a wave 64 or more words of one row; with E words per row most lanes idle and a row is a 4 E-byte access.  The rows x E
// array is taken as N1 x (N2 E), N2 E = CHUNK words, as in the four-step method: the N1-point transform over the upper row bits is the stripe kernels'
// (a chunk's words are its columns: uniform twiddles, wide accesses); this kernel, one workgroup per chunk c = bitrev(k1), does the rest inside LDS with
// per-lane twiddles from the decoder's w^u table: DIF (natural rows in, bit-reversed out; runs AFTER the stripe kernels): times w_N^(i2 k1), then
// the N2-point DIF over i2; DIT with the inverse roots (bit-reversed in, natural out; runs BEFORE them): the N2-point DIT, then times w_N^(-i2 k1).
// Either way rows end up where the stripe kernels' passes alone would leave them.
constexpr int CHUNK_LOG = 12, CHUNK = 1 << CHUNK_LOG;
constexpr uint64_t NARROW_COLUMNS = 4;  // tree levels of at most this many polynomials go this way
template <bool DIT>
__global__ __launch_bounds__(256) void chunk_transform_kernel(uint32_t* __restrict__ data, const uint32_t* __restrict__ wpow, int logE, int logN1, uint32_t step_n,
                                                              uint32_t step_n2, uint32_t nc_mask)
{
    __shared__ uint32_t lds[CHUNK], tw[CHUNK / 2];  // the chunk; w_N2^(+-j) for j < N2 / 2 in Montgomery form
    const uint32_t tid = threadIdx.x, k1 = __brev(blockIdx.x) >> (32 - logN1);
    uint32_t* base = data + (size_t)CHUNK * blockIdx.x;
    auto root = [&](uint32_t ex) { return gf::mul_mont(wpow[DIT ? (0u - ex) & nc_mask : ex], gf::MONT_R2); };
    const int levels = CHUNK_LOG - logE;
    const uint32_t half_rows = 1u << (levels - 1);
    for (uint32_t j = tid; j < half_rows; j += 256u) tw[j] = root(j * step_n2);
#pragma unroll
    for (int r = 0; r < CHUNK / 256; ++r) {
        const uint32_t i = r * 256u + tid;
        uint32_t v = base[i];
        if (!DIT) v = gf::mul_mont(v, root((i >> logE) * k1 * step_n));
        lds[i] = v;
    }
    __syncthreads();
    for (int s = 0; s < levels; ++s) {
        // rows r and r + h, h = 2^s (DIT, bottom up) or N2 >> (s + 1) (DIF, top down); twiddle w_2h^(r mod h) = w_N2^((r mod h) N2 / 2h)
        const int hb = DIT ? logE + s : CHUNK_LOG - 1 - s, sh = DIT ? levels - 1 - s : s;
        const uint32_t low = (1u << hb) - 1u;
#pragma unroll
        for (int r = 0; r < CHUNK / 512; ++r) {
            const uint32_t b = r * 256u + tid, i = ((b >> hb) << (hb + 1)) | (b & low), j = i + low + 1u;
            const uint32_t x = lds[i], y = lds[j], w = tw[((i & low) >> logE) << sh];
            if (DIT) {
                const uint32_t t = gf::mul_mont(y, w);
                lds[i] = gf::add(x, t);
                lds[j] = gf::sub(x, t);
            } else {
                lds[i] = gf::add(x, y);
                lds[j] = gf::mul_mont(gf::sub(x, y), w);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < CHUNK / 256; ++r) {
        const uint32_t i = r * 256u + tid;
        uint32_t v = lds[i];
        if (DIT) v = gf::mul_mont(v, root((i >> logE) * k1 * step_n));
        base[i] = v;
    }
}

// lv[m][0] = c_m, lv[m][1] = m c_m for the locator L = x^T + sum_{m<T} c_m x^m taken modulo x^NC - 1 (exact on the NC-th
// roots of unity): the two columns whose transforms are L(w^u) and (x L')(w^u)
__global__ __launch_bounds__(256) void locator_columns_kernel(const uint32_t* __restrict__ c, uint32_t* __restrict__ lv, uint32_t T, uint32_t NC)
{
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= NC) return;
    uint32_t v0 = m < T ? c[m] : 0u;
    uint32_t v1 = gf::mul(m, v0);
    if (m == T % NC) {  // the monic term x^T (T == NC wraps onto x^0)
        v0 = gf::add(v0, 1u);
        v1 = gf::add(v1, T % gf::P);
    }
    lv[2 * m] = v0;
    lv[2 * m + 1] = v1;
}

// From the values L(w^u), (x L')(w^u) of the padded locator L = x^pad l to the decoder's tables:
//   fin[u]  = l(w^u) (Montgomery) on surviving positions, 0 elsewhere           l(w^u) = L(w^u) w^(-u pad)
//   gout[i] = 1 / (w^u l'(w^u)) (Montgomery) for erased data block i at u = i << e   (x l')(w^u) = (x L')(w^u) w^(-u pad) there
__global__ __launch_bounds__(256) void finish_tables_kernel(const uint32_t* __restrict__ lv, const uint32_t* __restrict__ state,
                                                            const uint32_t* __restrict__ wpow, uint32_t* __restrict__ fin, uint32_t* __restrict__ gout,
                                                            uint32_t NC, uint32_t pad, int e, uint32_t user_k, uint32_t q, int lg2,
                                                            uint32_t* __restrict__ gout_par = nullptr, bool lv_bitrev = false)
{
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= NC) return;
    const uint32_t st = (state[u >> 2] >> (8 * (u & 3u))) & 0xFFu;
    const uint32_t back = (uint32_t)(((uint64_t)u * pad) % NC);
    const uint32_t corr = wpow[back == 0 ? 0 : NC - back];  // w^(-u pad)
    // where the two values for w^u sit in lv: position u for the power-of-two transform (natural order out); the way down
    // of a mixed-radix context (mixed_dif) leaves the value at w^(-v), v = q * bitrev(r) + j1, in block j1 * 2^lg2 + r
    uint32_t at = lv_bitrev ? (__brev(u) >> (32 - lg2)) : u;  // (power of two, transform_bitrev: the value for w^u sits at the bit-reversed position)
    if (q > 1) {
        const uint32_t v = u == 0 ? 0u : NC - u;
        at = (v % q << lg2) + (__brev(v / q) >> (32 - lg2));
    }
    fin[u] = st == ST_HELD ? gf::mul(gf::mul(lv[2 * at], corr), gf::MONT_ONE) : 0u;
    if ((u & ((1u << e) - 1u)) == 0) {
        const uint32_t i = u >> e;
        uint32_t g = 0;
        if (st == ST_LOST && i < user_k) g = gf::mul(dev_pow(gf::mul(lv[2 * at + 1], corr), gf::P - 2u), gf::MONT_ONE);
        gout[i] = g;
    } else if (gout_par) {  // (2k,k) layout: odd u is parity block u >> 1
        gout_par[u >> 1] = st == ST_LOST ? gf::mul(dev_pow(gf::mul(lv[2 * at + 1], corr), gf::P - 2u), gf::MONT_ONE) : 0u;
    }
}

// split transform: the per-block factors of the two half stripes in the order the first pass reads them, from fin (by codeword position)
__global__ __launch_bounds__(256) void split_rows_kernel(const uint32_t* __restrict__ fin, const uint32_t* __restrict__ gout, const uint32_t* __restrict__ order,
                                                         uint32_t* __restrict__ rows_data, uint32_t* __restrict__ rows_parity, uint32_t* __restrict__ rows_out,
                                                         uint32_t k, const uint32_t* __restrict__ gout_par = nullptr, uint32_t* __restrict__ rows_out_parity = nullptr)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= k) return;
    const uint32_t i = order[slot];
    rows_data[slot] = fin[2u * i];
    rows_parity[slot] = fin[2u * i + 1u];
    rows_out[slot] = gout[i];  // the last pass has the first one's tile shape, so its blocks come in the same order
    if (rows_out_parity) rows_out_parity[slot] = gout_par[i];
}
// ... and the factor of the parity half's coefficients, by position: -w^(-m) / 2 (Montgomery) at position bitrev(m); wpow[u] = w^u, w of order 2k
// (forward: w^m instead — the factor of the data half's coefficients in the parity chain of fastecc_repair)
__global__ __launch_bounds__(256) void split_pos_kernel(const uint32_t* __restrict__ wpow, uint32_t* __restrict__ pos, uint32_t k, int lg, uint32_t neg_half,
                                                        bool forward)
{
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= k) return;
    const uint32_t w = wpow[forward ? m : (m == 0 ? 0 : 2u * k - m)];
    pos[__brev(m) >> (32 - lg)] = gf::mul(gf::mul(w, neg_half), gf::MONT_ONE);
}

__global__ __launch_bounds__(256) void permute_kernel(const uint32_t* __restrict__ src, const uint32_t* __restrict__ order, uint32_t* __restrict__ dst,
                                                      uint32_t count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dst[i] = src[order[i]];
}

// ------------------------------------------------------------------------------------------------
// kernels: one wave per (block row, 64*V-word column chunk); the row's factor is a scalar
// ------------------------------------------------------------------------------------------------
// work[u] = block at codeword position u, times fin[u].  srcmap[u] names the block: row index, bit 31 set = parity
// stripe.  Positions without a surviving block have fin == 0: they are written as zeros and nothing is read for them.
template <int V>
__global__ __launch_bounds__(256) void decode_gather_kernel(const uint32_t* __restrict__ data, const uint32_t* __restrict__ parity,
                                                            uint32_t* __restrict__ work, const uint32_t* __restrict__ fin,
                                                            const uint32_t* __restrict__ srcmap, uint32_t S, uint32_t ld, uint32_t ld_work,
                                                            uint32_t col_chunks, uint64_t items)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t item = (uint64_t)blockIdx.x * 4u + wave;
    if (item >= items) return;
    const uint32_t cc = (uint32_t)(item % col_chunks);
    const uint32_t u = (uint32_t)(item / col_chunks);
    const uint32_t col = (cc * 64u + lane) * V;
    if (col >= S) return;
    const uint32_t f = as_constant(fin)[u];
    uint32_t x[V];
    if (f != 0) {
        const uint32_t m = as_constant(srcmap)[u];
        const uint32_t* src = ((m >> 31) ? parity : data) + (size_t)(m & 0x7FFFFFFFu) * ld + col;
        load_vec<V>(x, src);
#pragma unroll
        for (int v = 0; v < V; ++v) x[v] = gf::mul_mont(x[v], f);
    } else {
#pragma unroll
        for (int v = 0; v < V; ++v) x[v] = 0;
    }
    store_vec<V>(work + (size_t)u * ld_work + col, x);
}

// data[i] = recovered[i] * gout[i] for the erased data blocks (gout != 0); surviving blocks are not touched
template <int V>
__global__ __launch_bounds__(256) void decode_scatter_kernel(const uint32_t* __restrict__ recovered, uint32_t* __restrict__ data,
                                                             const uint32_t* __restrict__ gout, uint32_t S, uint32_t ld_rec, uint32_t ld,
                                                             uint32_t col_chunks, uint64_t items)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t item = (uint64_t)blockIdx.x * 4u + wave;
    if (item >= items) return;
    const uint32_t cc = (uint32_t)(item % col_chunks);
    const uint32_t i = (uint32_t)(item / col_chunks);
    const uint32_t f = as_constant(gout)[i];
    if (f == 0) return;  // wave-uniform
    const uint32_t col = (cc * 64u + lane) * V;
    if (col >= S) return;
    uint32_t x[V];
    load_vec<V>(x, recovered + (size_t)i * ld_rec + col);
#pragma unroll
    for (int v = 0; v < V; ++v) x[v] = gf::mul_mont(x[v], f);
    store_vec<V>(data + (size_t)i * ld + col, x);
}

// split transform of a code with fewer parity blocks: stage[q << fold] = parity[q] for the parity blocks in use (fin != 0 at their position); the
// rest of the stage is never read with a non-zero factor
template <int V>
__global__ __launch_bounds__(256) void split_stage_kernel(const uint32_t* __restrict__ parity, uint32_t* __restrict__ stage, const uint32_t* __restrict__ fin,
                                                          uint32_t S, int fold, uint32_t col_chunks, uint64_t items)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t item = (uint64_t)blockIdx.x * 4u + wave;
    if (item >= items) return;
    const uint32_t cc = (uint32_t)(item % col_chunks);
    const uint32_t q = (uint32_t)(item / col_chunks);
    if (as_constant(fin)[2u * (q << fold) + 1u] == 0) return;  // wave-uniform
    const uint32_t col = (cc * 64u + lane) * V;
    if (col >= S) return;
    uint32_t x[V];
    load_vec<V>(x, parity + (size_t)q * S + col);
    store_vec<V>(stage + (size_t)(q << fold) * S + col, x);
}

// split transform, small form: row m of the work stripe = parity block at position m << shift of the parity half (block (m << shift) >> fold of
// the parity stripe) times l at its codeword position — zero where that block is lost, not in use, or does not exist
template <int V>
__global__ __launch_bounds__(256) void split_small_gather_kernel(const uint32_t* __restrict__ parity, uint32_t* __restrict__ work, const uint32_t* __restrict__ fin,
                                                                 uint32_t S, int shift, int fold, uint32_t parity_blocks, uint32_t col_chunks, uint64_t items)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t item = (uint64_t)blockIdx.x * 4u + wave;
    if (item >= items) return;
    const uint32_t cc = (uint32_t)(item % col_chunks);
    const uint32_t m = (uint32_t)(item / col_chunks);
    const uint32_t h = m << shift, q = h >> fold;
    const uint32_t f = ((h & ((1u << fold) - 1u)) == 0 && q < parity_blocks) ? as_constant(fin)[2u * h + 1u] : 0u;  // wave-uniform
    const uint32_t col = (cc * 64u + lane) * V;
    if (col >= S) return;
    uint32_t x[V];
#pragma unroll
    for (int v = 0; v < V; ++v) x[v] = 0;
    if (f != 0) {
        load_vec<V>(x, parity + (size_t)q * S + col);
#pragma unroll
        for (int v = 0; v < V; ++v) x[v] = gf::mul_mont(x[v], f);
    }
    store_vec<V>(work + (size_t)m * S + col, x);
}

// packed[r] = stripe[rows[r]]: the rebuilt blocks side by side, for one copy to the host
template <int V>
__global__ __launch_bounds__(256) void pack_rows_kernel(const uint32_t* __restrict__ stripe, const uint32_t* __restrict__ rows, uint32_t* __restrict__ packed,
                                                        uint32_t S, uint32_t col_chunks, uint64_t items)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t item = (uint64_t)blockIdx.x * 4u + wave;
    if (item >= items) return;
    const uint32_t cc = (uint32_t)(item % col_chunks);
    const uint32_t r = (uint32_t)(item / col_chunks);
    const uint32_t col = (cc * 64u + lane) * V;
    if (col >= S) return;
    uint32_t x[V];
    load_vec<V>(x, stripe + (size_t)as_constant(rows)[r] * S + col);
    store_vec<V>(packed + (size_t)r * S + col, x);
}

// parity[q] = again[q] for the parity blocks that were lost (lost[q] != 0); the others are not touched
template <int V>
__global__ __launch_bounds__(256) void restore_parity_kernel(const uint32_t* __restrict__ again, uint32_t* __restrict__ parity,
                                                             const uint32_t* __restrict__ lost, uint32_t S, uint32_t col_chunks, uint64_t items)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t item = (uint64_t)blockIdx.x * 4u + wave;
    if (item >= items) return;
    const uint32_t cc = (uint32_t)(item % col_chunks);
    const uint32_t q = (uint32_t)(item / col_chunks);
    if (as_constant(lost)[q] == 0) return;  // wave-uniform
    const uint32_t col = (cc * 64u + lane) * V;
    if (col >= S) return;
    uint32_t x[V];
    load_vec<V>(x, again + (size_t)q * S + col);
    store_vec<V>(parity + (size_t)q * S + col, x);
}

int hip_code(const char* what, hipError_t e)
{
    set_error_detail(what, e);
    return e == hipErrorOutOfMemory ? FASTECC_E_NOMEM : FASTECC_E_DEVICE;
}

// FASTECC_TRACE_PREPARE=1: wall-clock of the phases of fastecc_decode_prepare on stderr (where does a first call spend its time)
struct PhaseTimer {
    bool on = getenv("FASTECC_TRACE_PREPARE") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void mark(const char* what)
    {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[fastecc prepare] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

#define DEC_TRY(expr)                                      \
    do {                                                   \
        hipError_t e_ = (expr);                            \
        if (e_ != hipSuccess) return hip_code(#expr, e_);  \
    } while (0)

struct DeviceScope {
    int prev = -1;
    bool ok = false;
    explicit DeviceScope(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceScope()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace

}  // namespace fastecc

using namespace fastecc;

extern "C" {

static int decode_prepare_impl(fastecc_ctx* c, const uint8_t* data_present, const uint8_t* parity_present);

// No exception crosses the ABI: the set-up and the host staging use std::vector; an allocation failure there is FASTECC_E_NOMEM (the call
// locks of the context are scoped objects, so they are released on the way out).
int fastecc_decode_prepare(fastecc_ctx* c, const uint8_t* data_present, const uint8_t* parity_present)
{
    try {
        return decode_prepare_impl(c, data_present, parity_present);
    } catch (const std::bad_alloc&) {
        return FASTECC_E_NOMEM;
    } catch (...) {
        return FASTECC_E_DEVICE;
    }
}

static int decode_prepare_impl(fastecc_ctx* c, const uint8_t* data_present, const uint8_t* parity_present)
{
    if (!c || !data_present || !parity_present) return FASTECC_E_INVAL;
    if (sharded_of(c)) return sharded_decode_prepare(c, data_present, parity_present);
    const CtxInfo ci = info_of(c);
    if (ci.field == FASTECC_FIELD_GF_P61_SQUARED) {
        // the 64-bit field has its own decoder (gf61_decode.hip); its contexts are (2k,k), (4k,k) or (8k,k) with k a power of two (and their zero-extended relatives)
        int e61 = 1;
        while ((1 << e61) < ci.cosets + 1) e61++;  // n = 4k / 8k: the same decoder on the (k << e)-th roots of unity (gf61_decode.hip)
        DeviceScope ds61(ci.device);
        if (!ds61.ok) return FASTECC_E_DEVICE;
        CallScope call61(c);
        {
            const int rc0 = call61.wait_idle();  // a decode still using the previous pattern
            if (rc0 != FASTECC_OK) return rc0;
        }
        char detail[160] = "";
        // codes other than (2N,N): the decoder sees the (2N,N) codeword — data blocks beyond the caller's k are surviving zero blocks,
        // parity positions the code does not use are lost (which is what limits the losses to n - k)
        std::vector<uint8_t> dfull, pfull;
        if (ci.zero_extended) {
            dfull.assign(ci.k, 1);
            pfull.assign(ci.k, 0);
            for (uint64_t i = 0; i < ci.user_k; i++) dfull[i] = data_present[i] ? 1 : 0;
            for (uint64_t q = 0; q < ci.user_m; q++) pfull[q * (uint64_t)ci.p61_stride] = parity_present[q] ? 1 : 0;
            data_present = dfull.data();
            parity_present = pfull.data();
        }
        const int rc = p61::decode_prepare(&decoder61_of(c), ci.log2k, ci.words / 4, data_present, parity_present, ci.direct_max, detail, sizeof detail, ci.decode_split, e61);
        if (rc != FASTECC_OK && detail[0]) set_error_detail(detail, hipErrorUnknown);
        return rc;
    }
    if (ci.field != FASTECC_FIELD_GF_FFF00001) return FASTECC_E_UNSUPPORTED;
    if (ci.pitch != ci.words) return FASTECC_E_UNSUPPORTED;
    // mixed-radix codes (fastecc_create_ex): the same scheme on the (2 q 2^m)-th roots of unity; the decoder's transform is a
    // mixed-radix context one size up, the locator's values come from the way down of another one (mixed_dif)
    const bool mixed = ci.q > 1;
    const uint64_t N = mixed ? (uint64_t)ci.q * ci.k : ci.k;

    // Every code is f on a subset of the NC-th roots of unity, NC = N << e (position u <-> w_NC^u): data block i at
    // i << e (blocks k..N-1 of a zero-extended code are known zero blocks), parity at the positions fastecc_create
    // documents — odd multiples of 2^fold for the codes inside (2N,N), the cosets' offsets for n = 4k / 8k.  Positions
    // that hold no block of the code count as erased, which is exactly what limits the losses to n - k.
    int e = 1;
    while ((1 << e) < ci.cosets + 1) e++;
    const uint64_t NC = N << e;
    const int lgc = ci.log2k + e;
    // the matrix-core kernel recomputes 256 blocks in a third of the transform path's time, the VALU kernel breaks even near 128
    // (profiles/r03/direct_bench.jsonl); stripes the MFMA kernel cannot take (odd or short rows) stop at 96 unless a kernel was asked for — at 80
    // where the split transform (4.3 ms instead of 7.2 at k = 2^19 x 4 KB) is the alternative
    int direct_limit = std::min(ci.direct_max, direct_cap());
    const bool split_applies = ci.decode_split && ci.q <= 1 && ci.cosets == 1 && ci.log2k >= 17;
    if (ci.direct_kernel == 0 && !direct_mfma_applies(nullptr, nullptr, ci.words)) direct_limit = std::min(direct_limit, split_applies ? 80 : 96);
    {
        // orders above 2^20 (mixed radix): the locator tree is padded to 2^20 roots whatever the pattern
        uint64_t T = 1;
        while (T < NC - N) T <<= 1;
        // their tree costs a 2^20-point product: the direct path first, up to the caller's "decode_direct_max" (the 80 / 96 cap of rows the
        // matrix-core kernel cannot take is a speed trade-off against a transform path that is much dearer here, so it does not apply)
        if (T > (1ull << 20) && ci.direct_max > 0) direct_limit = std::max(direct_limit, std::min(ci.direct_max, direct_cap()));
    }
    auto parity_position = [&](uint64_t q) -> uint64_t {
        if (ci.cosets > 1) {
            const uint64_t t = q / N, j = q % N;  // coset t = generator w_(N << jj)^c, see fastecc_create
            int jj = 1;
            while ((1ull << jj) - 1 <= t) jj++;
            const uint64_t odd = 2 * (t + 1 - (1ull << (jj - 1))) + 1;
            return (odd << (e - jj)) + (j << e);
        }
        return ((q << ci.fold) << 1) + 1;
    };
    // ---- few losses (decided before any per-position table is built): interpolation on the surviving data points + a few parity points ----
    if (direct_limit > 0 && ci.user_k < 0xFFFFFFF0ull) {
        PhaseTimer ptd;
        std::vector<uint32_t> R, Pl, A;
        bool over = false;
        // (eight flags per step: a word without a zero byte holds no lost block — byte by byte the two scans took 0.2-0.4 ms of a 0.3-0.5 ms call at k = 2^19)
        auto each_lost = [](const uint8_t* flags, uint64_t count, auto&& lost) {  // lost(i) returns false to stop
            uint64_t i = 0;
            for (; i + 8 <= count; i += 8) {
                uint64_t v;
                memcpy(&v, flags + i, 8);
                if (((v - 0x0101010101010101ull) & ~v & 0x8080808080808080ull) == 0) continue;
                for (uint64_t j = i; j < i + 8; j++)
                    if (!flags[j] && !lost(j)) return;
            }
            for (; i < count; i++)
                if (!flags[i] && !lost(i)) return;
        };
        each_lost(data_present, ci.user_k, [&](uint64_t i) { R.push_back((uint32_t)i); return !(over = (int)R.size() > direct_limit); });
        if (!over) each_lost(parity_present, ci.user_m, [&](uint64_t q) { Pl.push_back((uint32_t)q); return !(over = (int)(R.size() + Pl.size()) > direct_limit); });
        for (uint64_t q = 0; q < ci.user_m && !over && A.size() < R.size(); q++)
            if (parity_present[q]) A.push_back((uint32_t)q);
        if (!over && R.size() + Pl.size() >= 1 && A.size() == R.size()) {
            const int ed = (int)R.size(), ep = (int)Pl.size();
            ptd.mark("few losses: pattern scan");
            DeviceScope ds(ci.device);
            if (!ds.ok) return FASTECC_E_DEVICE;
            CallScope call(c);
            DecodeState*& slot = decoder_of(c);
            if (!slot) {
                slot = new (std::nothrow) DecodeState();
                if (!slot) return FASTECC_E_NOMEM;
            }
            DecodeState* d = slot;
            d->ready = false;
            d->sub = false;
            d->erased_data = ed;
            d->erased_parity = ep;
            d->erased_total = ed + ep;
            d->positions = NC;
            d->standard = !mixed && ci.cosets == 1 && ci.fold == 0 && !ci.zero_extended;
            d->mixed = mixed;
            d->direct_kernel = ci.direct_kernel;
            const uint32_t K = (uint32_t)ci.user_k;
            const uint32_t w = gf::h_root((uint32_t)NC), wd = gf::h_pow(w, 1ull << e);  // data row i sits at wd^i
            auto fsub = [](uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a + gf::P - b) % gf::P); };
            {
                const int rc = call.wait_idle();  // a decode still using the previous pattern
                if (rc != FASTECC_OK) return rc;
            }
            ptd.mark("few losses: lock, idle");
            int rc = FASTECC_OK;
            // (up to 32 outputs a pass costs the read of the survivors whatever it computes, profiles/r03/direct_bench.jsonl: one table then serves decode and repair)
            d->sub_only_both = ed > 0 && ep > 0 && ed + ep <= 32;
            if (ed > 0 && !d->sub_only_both) {
                std::vector<uint32_t> xr(ed), ya(ed);
                for (int r = 0; r < ed; r++) xr[r] = gf::h_pow(w, (uint64_t)R[r] << e);
                for (int a = 0; a < ed; a++) ya[a] = gf::h_pow(w, parity_position(A[a]));
                if (!d->direct_data && !(d->direct_data = direct_pass_new())) return FASTECC_E_NOMEM;
                rc = direct_build_interp(d->direct_data, wd, N, K, R, xr, A, ya, nullptr);
            }
            ptd.mark("few losses: data table");
            d->sub_both = false;
            if (rc == FASTECC_OK && ep > 0) {
                std::vector<uint32_t> yt(ep), ct(ep), pos(ep);
                const uint32_t inv_N = gf::h_inv((uint32_t)(N % gf::P));
                for (int t = 0; t < ep; t++) {
                    yt[t] = gf::h_pow(w, parity_position(Pl[t]));
                    ct[t] = gf::h_mul(fsub(gf::h_pow(yt[t], N), 1u), inv_N);  // (y_t^N - 1) / N
                    pos[t] = 2u * Pl[t] + 1u;
                }
                if (d->sub_only_both) {
                    // data lost as well, few outputs: fastecc_repair reads the survivors ONCE — the lost parity blocks are further outputs on the data
                    // pass's nodes (the surviving data and as many parity blocks), not a second pass over the repaired data.  (Above 32 outputs the
                    // matrix cores bound the pass, not the read: 128 + 128 lost take 2.40 ms in one pass, 2.48 in two, and the set-up of the second
                    // 256-output table costs 0.9 ms.)
                    std::vector<uint32_t> xr(ed), ya(ed);
                    for (int r = 0; r < ed; r++) xr[r] = gf::h_pow(w, (uint64_t)R[r] << e);
                    for (int a = 0; a < ed; a++) ya[a] = gf::h_pow(w, parity_position(A[a]));
                    if (!d->direct_both && !(d->direct_both = direct_pass_new())) return FASTECC_E_NOMEM;
                    rc = direct_build_interp(d->direct_both, wd, N, K, R, xr, A, ya, nullptr, &yt, &pos);
                    d->sub_both = rc == FASTECC_OK;
                } else {
                    if (!d->direct_parity && !(d->direct_parity = direct_pass_new())) return FASTECC_E_NOMEM;
                    rc = direct_build_lagrange(d->direct_parity, wd, K, yt, ct, pos, nullptr);
                }
            }
            ptd.mark("few losses: parity table");
            if (rc == FASTECC_OK) {
                d->sub = true;
                d->sub_lost_data = ed;
                d->sub_lost_parity = ep;
                d->host_lost_data = R;  // (FASTECC_MEM_HOST calls: the rows that travel back)
                d->host_lost_parity = Pl;
                d->host_parity_used = A;
                d->host_lists_of = ++d->pattern_serial;
                d->ready = true;
                return FASTECC_OK;
            }
            if (rc != FASTECC_E_NOMEM) return rc;
            // no memory for the weight tables: the transform path below needs none of them
        }
    }
    PhaseTimer pt;
    enum : uint8_t { LOST = ST_LOST, HELD = ST_HELD, ZERO = ST_ZERO };
    const bool standard_layout = !mixed && ci.cosets == 1 && ci.fold == 0 && !ci.zero_extended;
    uint64_t erased_data = 0;
    uint32_t split_groups = 0, split_shift = 0;
    // ---- the reference's (2k,k) layout: the pattern is scanned on the DEVICE (the host loops below took 1.2-1.4 ms of a 2.6 ms call at k = 2^19:
    // four passes over a million byte-sized coin flips).  Two 512 KB uploads, one counting kernel, 32 bytes back; the per-position state and the
    // lost-parity flags are then written by a kernel.  Patterns the "small form" of the split transform does not take (too few surviving
    // parity blocks at multiples of 2^h) keep the host path. ----
    bool device_scan = false;
    uint64_t device_erased_count = 0, device_erased_parity = 0;
    DeviceScope ds(ci.device);
    if (!ds.ok) return FASTECC_E_DEVICE;
    CallScope call(c);  // (held from here on: the scan below already writes the context's pattern state)
    DecodeState*& slot = decoder_of(c);
    if (!slot) {
        slot = new (std::nothrow) DecodeState();
        if (!slot) return FASTECC_E_NOMEM;
    }
    if (standard_layout && N <= 0x7FFFFFFFull) {
        DecodeState* d0 = slot;
        if (!d0->dev_present) DEC_TRY(hipMalloc((void**)&d0->dev_present, 2 * N));
        if (!d0->dev_counts) DEC_TRY(hipMalloc((void**)&d0->dev_counts, 8 * 4));
        if (!d0->dev_state) DEC_TRY(hipMalloc((void**)&d0->dev_state, NC));
        if (!d0->parity_lost) DEC_TRY(hipMalloc((void**)&d0->parity_lost, ci.user_m * 4));
        {
            const int rc = call.wait_idle();  // a decode or repair still reading the previous pattern's state
            if (rc != FASTECC_OK) return rc;
        }
        d0->ready = false;
        hipStream_t st0 = nullptr;
        DEC_TRY(hipMemcpyAsync(d0->dev_present, data_present, N, hipMemcpyHostToDevice, st0));
        DEC_TRY(hipMemcpyAsync(d0->dev_present + N, parity_present, N, hipMemcpyHostToDevice, st0));
        DEC_TRY(hipMemsetAsync(d0->dev_counts, 0, 8 * 4, st0));
        hipLaunchKernelGGL(presence_counts_kernel, dim3(128), dim3(256), 0, st0, d0->dev_present, d0->dev_present + N, (uint32_t)N, d0->dev_counts);
        DEC_TRY(hipGetLastError());
        uint32_t counts[8] = {};
        DEC_TRY(hipMemcpy(counts, d0->dev_counts, sizeof counts, hipMemcpyDeviceToHost));
        erased_data = counts[0];
        device_erased_parity = counts[1];
        const bool want_split0 = ci.decode_split && ci.log2k >= 17 && erased_data != 0;
        if (want_split0 && ci.decode_split != 2)
            for (int h = 5; h >= 1 && split_shift == 0; h--)
                if (counts[1 + h] >= erased_data) split_shift = (uint32_t)h;
        if (!want_split0 || split_shift != 0) {
            device_scan = true;
            const uint64_t unused = split_shift ? (N - device_erased_parity) - counts[1 + split_shift] : 0;
            device_erased_count = erased_data + device_erased_parity + unused;
            split_groups = split_shift ? 1u : 0u;  // (non-zero: "this pattern goes through the split transform" for the code below)
            hipLaunchKernelGGL(standard_state_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st0, d0->dev_present, d0->dev_present + N, (uint32_t)N,
                               split_shift ? (1u << split_shift) - 1u : 0u, (uint8_t*)d0->dev_state, d0->parity_lost);
            DEC_TRY(hipGetLastError());
        } else {
            erased_data = 0;  // the host path counts again
            split_shift = 0;
        }
    }
    // (branch-free loops: on a random pattern every "if (present)" is a coin flip — 2^20 mispredictions were most of this call's time at 50 % loss)
    std::vector<uint8_t> state(device_scan ? 0 : NC, LOST);
    // the block map serves the table-driven gather only: the (2k,k) layout reads its two stripes by position
    std::vector<uint32_t> srcmap(standard_layout ? 0 : NC, 0);
    if (device_scan) {
        // nothing to do on the host
    } else if (standard_layout) {
        for (uint64_t i = 0; i < N; i++) {
            const uint32_t held_d = data_present[i] != 0, held_p = parity_present[i] != 0;
            state[2 * i] = held_d ? HELD : LOST;
            state[2 * i + 1] = held_p ? HELD : LOST;
            erased_data += 1u - held_d;
        }
    } else {
        for (uint64_t i = 0; i < ci.user_k; i++) {
            const uint64_t u = i << e;
            const uint32_t held = data_present[i] != 0;
            state[u] = held ? HELD : LOST;
            srcmap[u] = (uint32_t)i & (0u - held);
            erased_data += 1u - held;
        }
        for (uint64_t i = ci.user_k; i < N; i++) state[i << e] = ZERO;
        for (uint64_t q = 0; q < ci.user_m; q++) {
            const uint64_t u = parity_position(q);
            const uint32_t held = parity_present[q] != 0;
            state[u] = held ? HELD : LOST;
            srcmap[u] = ((uint32_t)q | 0x80000000u) & (0u - held);
        }
    }
    // (2k,k) layout, split transform: recovering e lost data blocks takes e parity blocks, not all of them — the surviving parity blocks of
    // the first few block groups of the parity stripe (group g = blocks g + (t << 10): what one tile of the first pass reads).  The others
    // are left unread: roots of the locator like the lost ones.
    // (also the zero-extended codes inside (2N,N): data block i at position 2i, parity block j at 2j + 1, fewer blocks than N in either stripe;
    // and the codes with fewer parity blocks: parity block j at position 2 (j << fold) + 1, i.e. block j << fold of the parity half)
    const bool split_layout = !mixed && ci.cosets == 1;
    const bool want_split = !device_scan && ci.decode_split && split_layout && ci.log2k >= 17 && erased_data != 0;
    if (want_split) {
        // first choice: the surviving parity blocks at multiples of 2^h of the parity half, the largest h <= 5 that still leaves as many as there are
        // lost data blocks (2 % of the codeword lost: h = 5) — r~ is then the transform of k >> h rows (see DecodeState::split_shift)
        uint64_t at_multiple[6] = {};
        for (uint64_t q = 0; q < ci.user_m; q++) {
            if (!parity_present[q]) continue;
            const uint64_t hpos = q << ci.fold;
            for (int h = 1; h <= 5 && (hpos & ((1ull << h) - 1ull)) == 0; h++) at_multiple[h]++;
        }
        for (int h = 5; h >= 1 && split_shift == 0; h--)
            if (at_multiple[h] >= erased_data && ci.decode_split != 2) split_shift = (uint32_t)h;
        if (split_shift != 0) {
            const uint64_t mask = (1ull << split_shift) - 1ull;
            for (uint64_t q = 0; q < N; q++)
                if ((q & mask) != 0 && state[2 * q + 1] == HELD) state[2 * q + 1] = (uint8_t)ST_UNUSED;
            split_groups = 1;  // (non-zero: "this pattern goes through the split transform" for the code below)
        } else {
        constexpr uint32_t GROUPS = 1024;
        uint32_t held_in[GROUPS] = {};
        for (uint64_t q = 0; q < ci.user_m; q++) held_in[(q << ci.fold) & (GROUPS - 1u)] += parity_present[q] != 0;
        uint64_t have = 0;
        while (split_groups < GROUPS && have < erased_data) have += held_in[split_groups++];
        if (have >= erased_data) {
            for (uint64_t q0 = 0; q0 < N; q0 += GROUPS)  // (the blocks of the groups in use keep their state)
                for (uint64_t q = q0 + split_groups; q < q0 + GROUPS; q++) state[2 * q + 1] = state[2 * q + 1] == HELD ? (uint8_t)ST_UNUSED : state[2 * q + 1];
        } else {
            split_groups = 0;  // not decodable: refused below
        }
        }
    }
    if (split_groups != 0 && !srcmap.empty())
        for (uint64_t u = 1; u < NC; u += 2) srcmap[u] &= 0u - (uint32_t)(state[u] != ST_UNUSED);
    // the erased positions themselves are listed on the device (erased_list_kernel): the host needs their number only
    uint64_t erased_count = device_erased_count;
    if (!device_scan)
        for (uint64_t u = 0; u < NC; u++) erased_count += (unsigned)(state[u] == LOST) + (unsigned)(state[u] == ST_UNUSED);
    if (erased_count > NC - N) return FASTECC_E_INVAL;  // fewer than k blocks survive: not decodable
    pt.mark(device_scan ? "pattern scan (device)" : "pattern scan (host)");

    DecodeState* d = slot;
    d->ready = false;
    d->erased_data = erased_data;
    d->erased_total = erased_count;
    d->positions = NC;
    d->standard = !mixed && ci.cosets == 1 && ci.fold == 0 && !ci.zero_extended;
    d->mixed = mixed;
    pt.mark("lock, state");
    if (device_scan) {
        d->erased_parity = device_erased_parity;  // (the flags were written by standard_state_kernel)
    } else {
        std::vector<uint32_t> plost(ci.user_m);
        d->erased_parity = 0;
        for (uint64_t q = 0; q < ci.user_m; q++) d->erased_parity += (plost[q] = parity_present[q] ? 0u : 1u);
        if (!d->parity_lost) DEC_TRY(hipMalloc((void**)&d->parity_lost, ci.user_m * 4));
        const int rc = call.wait_idle();  // a repair still reading the previous pattern
        if (rc != FASTECC_OK) return rc;
        DEC_TRY(hipMemcpy(d->parity_lost, plost.data(), ci.user_m * 4, hipMemcpyHostToDevice));
    }
    pt.mark("lost-parity flags");
    d->sub = false;
    ++d->pattern_serial;  // (the lists of rebuilt rows for FASTECC_MEM_HOST calls are made when such a call comes: decode_impl)
    if (erased_data == 0) {  // no data block to recover
        d->ready = true;
        return FASTECC_OK;
    }

    // ---- device state of the decoder (built once) ----
    // T = padded root count: the smallest power of two that holds the most losses the code tolerates, NC - N
    // (the top of the product tree is a cyclic product of length T, so w_T must exist: T <= 2^20.  Orders above 2^20 — mixed radix —
    // tolerate more losses than that; there T = 2^20 and patterns with more erasures than T are refused.)
    uint64_t T = 1;
    while (T < NC - N && T < (1ull << 20)) T <<= 1;
    int lgT = 0;
    while ((1ull << lgT) < T) lgT++;
    if (erased_count > T) return FASTECC_E_UNSUPPORTED;
    // the levels below 2^TREE_LOW roots per polynomial are one kernel (tree_low_levels_kernel) when the tree is tall enough to have them
    const int leaf_log = lgT >= TREE_LOW + 2 ? TREE_LOW : std::min(LEAF_LOG, lgT), leaf = 1 << leaf_log;
    const uint32_t w = gf::h_root((uint32_t)NC);
    hipStream_t st = nullptr;  // the set-up is synchronous: it runs on the default stream and ends with a synchronise
    const bool narrow = !mixed && (1ull << lgc) == NC;  // (the w^u table then holds every root the chunks need)
    if (!d->pattern_ntt) {
        int rc;
        if (mixed) {
            const std::vector<uint32_t> ones(NC, 1u);
            rc = create_mixed_transform_ctx(&d->pattern_ntt, ci.q, lgc, 8, ones.data(), ci.device);
        } else {
            // only its stand-alone transform is used; long ones as the upper row bits of a four-step transform (chunk_transform_kernel)
            d->pattern_narrow = narrow && lgc + 1 - CHUNK_LOG >= 1;
            rc = d->pattern_narrow ? create_ntt_ctx(&d->pattern_ntt, lgc + 1 - CHUNK_LOG, 4 * CHUNK, ci.device) : create_ntt_ctx(&d->pattern_ntt, lgc, 8, ci.device);
        }
        if (rc != FASTECC_OK) return rc;
    }
    pt.mark("pattern_ntt context");
    if (!d->pattern_buf) DEC_TRY(hipMalloc((void**)&d->pattern_buf, 2 * NC * 4));
    if (!d->transform) {
        // x p'(x): coefficient m times m, and the 1/NC of the inverse transform.  fold e: only the data positions (multiples of 2^e) are
        // evaluated (mixed radix: all positions, the even ones are used)
        const uint32_t inv_nc = gf::h_inv((uint32_t)NC);
        int rc;
        if (mixed) {
            std::vector<uint32_t> factor(NC);
            const uint32_t inv_nc_m = gf::h_to_mont(inv_nc);
            for (uint64_t m = 0; m < NC; m++) factor[m] = gf::h_mont_mul((uint32_t)m, inv_nc_m);
            rc = create_mixed_transform_ctx(&d->transform, ci.q, lgc, ci.words * 4, factor.data(), ci.device);
        } else {
            rc = create_ramp_transform_ctx(&d->transform, lgc, ci.words * 4, e, inv_nc, ci.device);
        }
        if (rc != FASTECC_OK) return rc;
    }
    pt.mark("transform context");
    if (d->tree_T != T || d->tree_low != leaf_log) {
        // level k >= leaf_log multiplies pairs of degree-2^k polynomials: transforms of length 2^(k+1) on T / 2^k columns
        for (fastecc_ctx* t : d->tree_ctx)
            if (t) fastecc_destroy(t);
        if (d->tree_top) fastecc_destroy(d->tree_top);
        d->tree_top = nullptr;
        d->tree_ctx.assign(lgT, nullptr);
        const bool narrow_tree = narrow && lgT + 1 - CHUNK_LOG >= 1 && lgT > leaf_log;
        for (int k = leaf_log; k < lgT; k++) {
            if (narrow_tree && (T >> k) <= NARROW_COLUMNS) continue;  // tree_top + chunk_transform_kernel
            const int rc = create_ntt_ctx(&d->tree_ctx[k], k + 1, 4 * (T >> k), ci.device);
            if (rc != FASTECC_OK) return rc;
        }
        if (narrow_tree) {
            const int rc = create_ntt_ctx(&d->tree_top, lgT + 1 - CHUNK_LOG, 4 * CHUNK, ci.device);
            if (rc != FASTECC_OK) return rc;
        }
        for (uint32_t** b : {&d->tree_x, &d->tree_f, &d->tree_y, &d->tree_p, &d->roots, &d->dev_erased}) {
            if (*b) (void)hipFree(*b);
            *b = nullptr;
        }
        DEC_TRY(hipMalloc((void**)&d->tree_x, 2 * T * 4));
        DEC_TRY(hipMalloc((void**)&d->tree_f, 2 * T * 4));
        DEC_TRY(hipMalloc((void**)&d->tree_y, 2 * T * 4));
        DEC_TRY(hipMalloc((void**)&d->tree_p, 2 * T * 4));
        DEC_TRY(hipMalloc((void**)&d->roots, T * 4));
        DEC_TRY(hipMalloc((void**)&d->dev_erased, (T + 1) * 4));  // + the counter of erased_list_kernel
        d->tree_T = T;
        d->tree_low = leaf_log;
    }
    pt.mark("tree contexts + buffers");
    if (!d->wpow) {
        // the table becomes visible to later calls only once the kernel that fills it has been launched without error
        // (an unfilled table behind a non-null pointer would give silently wrong weights on the next prepare)
        uint32_t* fresh = nullptr;
        DEC_TRY(hipMalloc((void**)&fresh, NC * 4));
        hipLaunchKernelGGL(wpow_kernel, dim3((unsigned)((NC + 255) / 256)), dim3(256), 0, st, fresh, w, (uint32_t)NC);
        const hipError_t e_fill = hipGetLastError();
        if (e_fill != hipSuccess) {
            (void)hipFree(fresh);
            return hip_code("wpow_kernel", e_fill);
        }
        d->wpow = fresh;
    }
    if (!d->dev_state) DEC_TRY(hipMalloc((void**)&d->dev_state, NC));
    if (!d->fin) DEC_TRY(hipMalloc((void**)&d->fin, NC * 4));
    if (!d->srcmap) DEC_TRY(hipMalloc((void**)&d->srcmap, NC * 4));
    if (!d->gout) DEC_TRY(hipMalloc((void**)&d->gout, N * 4));
    // parity block j at position 2j + 1 (the (2k,k) layout and its zero-extended relatives with fold 0): a pattern that loses data AND parity
    // gets the factors of its lost parity blocks too — fastecc_repair then needs no second encode
    const bool parity_factors = d->erased_parity != 0 && (d->standard || (split_groups != 0 && ci.fold == 0));
    if (parity_factors && !d->gout_par) DEC_TRY(hipMalloc((void**)&d->gout_par, N * 4));
    // mixed radix: the work stripe of all NC positions, transformed in place; else the N recovered data positions
    if (!d->recovered) DEC_TRY(hipMalloc((void**)&d->recovered, (mixed ? NC : N) * ci.words * 4));
    d->split_ready = false;
    d->split_repair_ready = false;
    if (split_groups != 0 && !d->split_unavailable && !d->split) {
        // ---- the split transform's context and tables (once).  Anything missing — a plan without the tile shapes, no memory for the two extra
        // stripes — leaves the 2k-point transform in charge; the pattern's unused parity blocks are unused there as well. ----
        const int rc_split = [&]() -> int {
            // per-block factor (2m + k) / 2k = m / k + 1 / 2
            int rc = create_ramp_transform_ctx(&d->split, ci.log2k, ci.words * 4, 0, gf::h_inv((uint32_t)N), ci.device, gf::h_inv(2u));
            if (rc != FASTECC_OK) return rc;
            if (!split_decode_supported(d->split) || split_decode_groups(d->split) != 1024u) return FASTECC_E_UNSUPPORTED;
            for (uint32_t** b : {&d->split_order, &d->split_rows_data, &d->split_rows_parity, &d->split_rows_out, &d->split_pos_parity, &d->split_pos_data_odd,
                                 &d->split_rows_out_parity})
                DEC_TRY(hipMalloc((void**)b, N * 4));
            if (!gather_tile_order_device(d->split, d->split_order, st)) return FASTECC_E_UNSUPPORTED;
            {
                // the parity half's low levels when few block groups are in use (run_split_decode): what the DIF levels with strides 512 ... 16
                // make of a 1024-block tile in which block q0 alone is 1 — simulated here exactly as the tile does them, (a, b) -> (a + b,
                // (a - b) w_2s^i) with the inverse roots; entry [t][g][c] = block g + 16 c for q0 = g + 16 t
                const uint32_t w1024_inv = gf::h_pow(gf::h_inv(gf::h_mul(w, w)), N / 1024);
                const uint32_t tables = (uint32_t)split_impulse_max();
                std::vector<uint32_t> table((size_t)tables * 16 * 64), v(1024), tw(512);
                for (uint32_t q0 = 0; q0 < 16u * tables; q0++) {
                    std::fill(v.begin(), v.end(), 0u);
                    v[q0] = 1;
                    for (uint32_t sdist = 512; sdist >= 16; sdist >>= 1) {
                        const uint32_t root = gf::h_pow(w1024_inv, 512 / sdist);  // order 2 * sdist
                        tw[0] = 1;
                        for (uint32_t i = 1; i < sdist; i++) tw[i] = gf::h_mul(tw[i - 1], root);
                        for (uint32_t base = 0; base < 1024; base += 2 * sdist)
                            for (uint32_t i = 0; i < sdist; i++) {
                                const uint32_t lo = v[base + i], hi = v[base + i + sdist];
                                if ((lo | hi) == 0) continue;
                                v[base + i] = (uint32_t)(((uint64_t)lo + hi) % gf::P);
                                v[base + i + sdist] = gf::h_mul((uint32_t)(((uint64_t)lo + gf::P - hi) % gf::P), tw[i]);
                            }
                    }
                    const uint32_t t = q0 / 16, g0 = q0 % 16;
                    for (uint32_t cc = 0; cc < 64; cc++) table[(t * 16 + g0) * 64 + cc] = gf::h_to_mont(v[g0 + 16 * cc]);
                }
                DEC_TRY(hipMalloc((void**)&d->split_impulse, table.size() * 4));
                DEC_TRY(hipMemcpy(d->split_impulse, table.data(), table.size() * 4, hipMemcpyHostToDevice));
            }
            const uint32_t neg_half = (uint32_t)(gf::P - gf::h_inv(2u));
            hipLaunchKernelGGL(split_pos_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, d->wpow, d->split_pos_parity, (uint32_t)N, ci.log2k, neg_half, false);
            hipLaunchKernelGGL(split_pos_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, d->wpow, d->split_pos_data_odd, (uint32_t)N, ci.log2k, neg_half, true);
            DEC_TRY(hipGetLastError());
            return FASTECC_OK;
        }();
        if (rc_split == FASTECC_OK) d->split_dirty = 0;
        if (rc_split != FASTECC_OK) {
            // nothing half-built stays behind: a later call either builds all of it or none
            (void)hipGetLastError();
            if (d->split) fastecc_destroy(d->split);
            d->split = nullptr;
            for (uint32_t** b : {&d->split_order, &d->split_rows_data, &d->split_rows_parity, &d->split_rows_out, &d->split_pos_parity, &d->split_impulse, &d->split_r1,
                                 &d->split_r2, &d->split_pos_data_odd, &d->split_rows_out_parity}) {
                if (*b) (void)hipFree(*b);
                *b = nullptr;
            }
            if (rc_split != FASTECC_E_NOMEM && rc_split != FASTECC_E_UNSUPPORTED) return rc_split;
            d->split_unavailable = true;
        }
    }
    if (split_groups != 0 && d->split) {
        // the parity half's work buffers, by form.  No memory for them: the 2k-point transform serves this pattern (as for a missing tile shape).
        hipError_t e = hipSuccess;
        if (split_shift != 0) {
            const uint64_t rows = N >> split_shift;
            if (!d->split_small[split_shift]) {
                const int rc = create_ntt_ctx(&d->split_small[split_shift], ci.log2k - (int)split_shift, ci.words * 4, ci.device);
                if (rc != FASTECC_OK && rc != FASTECC_E_NOMEM && rc != FASTECC_E_UNSUPPORTED) return rc;
                if (rc != FASTECC_OK) e = hipErrorOutOfMemory;
            }
            if (e == hipSuccess && d->split_small_blocks < rows) {
                if (d->split_small_buf) (void)hipFree(d->split_small_buf);
                d->split_small_buf = nullptr;
                d->split_small_blocks = 0;
                e = hipMalloc((void**)&d->split_small_buf, rows * ci.words * 4);
                if (e == hipSuccess) d->split_small_blocks = rows;
            }
        } else if (!d->split_r1 || !d->split_r2) {
            if (!d->split_r1) e = hipMalloc((void**)&d->split_r1, N * ci.words * 4);
            if (e == hipSuccess && !d->split_r2) e = hipMalloc((void**)&d->split_r2, N * ci.words * 4);
            if (e == hipSuccess) e = hipMemsetAsync(d->split_r1, 0, N * ci.words * 4, st);
            d->split_dirty = 0;
            if (e != hipSuccess) {
                for (uint32_t** b : {&d->split_r1, &d->split_r2}) {
                    if (*b) (void)hipFree(*b);
                    *b = nullptr;
                }
            }
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            split_groups = 0;  // (the pattern's unused parity blocks stay unused: the 2k-point transform reads the same factors)
            split_shift = 0;
        }
    }
    if (d->standard && d->erased_parity != 0 && !(split_groups != 0 && d->split)) {
        // repair in one transform (see DecodeState::transform_full): the form for patterns or plans the split transform does not take
        if (!d->transform_full) {
            const int rc = create_ramp_transform_ctx(&d->transform_full, lgc, ci.words * 4, 0, gf::h_inv((uint32_t)NC), ci.device);
            if (rc != FASTECC_OK && rc != FASTECC_E_NOMEM) return rc;
            d->full_ok = d->transform_full && same_tile_order(d->transform, d->transform_full);
            if (getenv("FASTECC_TRACE_PREPARE"))
                fprintf(stderr, "[fastecc prepare] one-transform repair: context %s, same first-pass order %d (%s | %s)\n", d->transform_full ? "built" : "none",
                        (int)d->full_ok, fastecc_plan_string(d->transform), d->transform_full ? fastecc_plan_string(d->transform_full) : "");
        }
    }
    if (d->standard && !d->tile_order_valid) {
        if (same_tile_order(d->transform, d->transform)) {  // (a tile first pass: the order exists)
            DEC_TRY(hipMalloc((void**)&d->tile_order, NC * 4));
            if (!gather_tile_order_device(d->transform, d->tile_order, st)) return FASTECC_E_DEVICE;
            DEC_TRY(hipMalloc((void**)&d->fin_first_pass, NC * 4));
        } else {
            d->fin_first_pass = d->fin;
        }
        d->tile_order_valid = true;
    }
    if (!d->standard) d->fin_first_pass = d->fin;
    {
        const int rc = call.wait_idle();  // a decode still using the previous pattern
        if (rc != FASTECC_OK) return rc;
    }

    pt.mark("tables, tile order");
    // ---- this pattern ----
    if (!device_scan) DEC_TRY(hipMemcpyAsync(d->dev_state, state.data(), NC, hipMemcpyHostToDevice, st));
    auto grid = [](uint64_t items) { return dim3((unsigned)((items + 255) / 256)); };
    if (!srcmap.empty()) DEC_TRY(hipMemcpyAsync(d->srcmap, srcmap.data(), NC * 4, hipMemcpyHostToDevice, st));
    else hipLaunchKernelGGL(standard_srcmap_kernel, grid(NC), dim3(256), 0, st, (const uint8_t*)d->dev_state, (uint32_t)NC, d->srcmap);
    // the list of erased positions (any order: the locator is their product); its counter sits behind the list
    DEC_TRY(hipMemsetAsync(d->dev_erased + T, 0, 4, st));
    hipLaunchKernelGGL(erased_list_kernel, grid((NC + 15) / 16), dim3(256), 0, st, (const uint8_t*)d->dev_state, (uint32_t)NC, d->dev_erased, d->dev_erased + T);
    hipLaunchKernelGGL(roots_kernel, grid(T), dim3(256), 0, st, d->roots, d->dev_erased, d->wpow, (uint32_t)erased_count, (uint32_t)T);
    // leaves: T / leaf polynomials of degree `leaf`, side by side ([coefficient][polynomial]); the upper half of the
    // 2*leaf rows the first product needs is zero
    DEC_TRY(hipMemsetAsync(d->tree_x, 0, 2 * T * 4, st));
    if (leaf_log == TREE_LOW) hipLaunchKernelGGL(tree_low_levels_kernel<TREE_LOW>, dim3((unsigned)(T >> leaf_log)), dim3(1 << TREE_LOW), 0, st, d->roots, d->tree_x, (uint32_t)(T >> leaf_log));
    else hipLaunchKernelGGL(leaf_products_kernel, grid(T >> leaf_log), dim3(256), 0, st, d->roots, d->tree_x, (uint32_t)leaf, (uint32_t)(T >> leaf_log));
    DEC_TRY(hipGetLastError());
    // 2^log_rows rows of 2^logE words (2^log_rows divides NC): DIF with the forward roots (natural -> bit-reversed rows) or DIT with the inverse
    // roots (bit-reversed -> natural); `top` has 2^(log_rows + logE) / CHUNK rows of CHUNK words
    auto narrow_transform = [&](fastecc_ctx* top, int log_rows, int logE, const uint32_t* in, uint32_t* out, bool dit) -> int {
        const int logN1 = log_rows + logE - CHUNK_LOG;
        const uint32_t step_n = (uint32_t)(NC >> log_rows), step_n2 = (uint32_t)(NC >> (CHUNK_LOG - logE));
        if (!dit) {
            const int rc = transform_bitrev(top, in, out, false, false, CHUNK, st);
            if (rc != FASTECC_OK) return rc;
            hipLaunchKernelGGL(chunk_transform_kernel<false>, dim3(1u << logN1), dim3(256), 0, st, out, d->wpow, logE, logN1, step_n, step_n2, (uint32_t)(NC - 1));
            return hipGetLastError() == hipSuccess ? FASTECC_OK : FASTECC_E_DEVICE;
        }
        if (in != out) return FASTECC_E_INVAL;
        hipLaunchKernelGGL(chunk_transform_kernel<true>, dim3(1u << logN1), dim3(256), 0, st, out, d->wpow, logE, logN1, step_n, step_n2, (uint32_t)(NC - 1));
        if (hipGetLastError() != hipSuccess) return FASTECC_E_DEVICE;
        return transform_bitrev(top, out, out, true, true, CHUNK, st);
    };
    uint32_t* x = d->tree_x;
    uint32_t* spare = d->tree_y;  // x / spare swap roles level by level; tree_f always holds the transforms
    for (int k = leaf_log; k < lgT; k++) {
        const uint64_t deg = 1ull << k, m = T >> k;  // m polynomials of degree deg in x: [2 deg][m], rows deg.. are zero
        fastecc_ctx* t = d->tree_ctx[k];  // (none: few columns, narrow_transform)
        int rc = t ? transform_bitrev(t, x, d->tree_f, false, false, (uint32_t)m, st)                   // all of them at once
                   : narrow_transform(d->tree_top, k + 1, lgT - k, x, d->tree_f, false);
        if (rc != FASTECC_OK) return rc;
        const uint32_t scale = gf::h_to_mont(gf::h_inv((uint32_t)(2 * deg)));
        hipLaunchKernelGGL(pointwise_pairs_kernel, grid(2 * deg * (m / 2)), dim3(256), 0, st, d->tree_f, d->tree_p, (uint32_t)m, 2 * deg * (m / 2), scale);
        DEC_TRY(hipGetLastError());
        rc = t ? transform_bitrev(t, d->tree_p, d->tree_p, true, true, (uint32_t)(m / 2), st)          // the products, back in natural order
               : narrow_transform(d->tree_top, k + 1, lgT - k, d->tree_p, d->tree_p, true);              // (the unused columns ride along)
        if (rc != FASTECC_OK) return rc;
        const bool top = k + 1 == lgT;
        const uint64_t rows = top ? 2 * deg : 4 * deg;
        hipLaunchKernelGGL(combine_kernel, grid(rows * (m / 2)), dim3(256), 0, st, d->tree_p, x, spare, (uint32_t)deg, (uint32_t)m, rows * (m / 2), top);
        DEC_TRY(hipGetLastError());
        std::swap(x, spare);
    }
    // x now holds the T lower coefficients of L = x^pad * l (monic of degree T), pad = T - |E|
    hipLaunchKernelGGL(locator_columns_kernel, grid(NC), dim3(256), 0, st, x, d->pattern_buf, (uint32_t)T, (uint32_t)NC);
    DEC_TRY(hipGetLastError());
    {
        // (power of two: the values stay in bit-reversed order, finish_tables_kernel reads them there — the reordering pass of fastecc_ntt was 87 us)
        const int rc = mixed               ? mixed_dif(d->pattern_ntt, d->pattern_buf, d->pattern_buf, st)
                       : d->pattern_narrow ? narrow_transform(d->pattern_ntt, lgc, 1, d->pattern_buf, d->pattern_buf, false)
                                           : transform_bitrev(d->pattern_ntt, d->pattern_buf, d->pattern_buf, false, false, 2, st);
        if (rc != FASTECC_OK) return rc;
    }
    hipLaunchKernelGGL(finish_tables_kernel, grid(NC), dim3(256), 0, st, d->pattern_buf, d->dev_state, d->wpow, d->fin, d->gout, (uint32_t)NC,
                       (uint32_t)(T - erased_count), e, (uint32_t)ci.user_k, (uint32_t)(mixed ? ci.q : 1), lgc,
                       parity_factors ? d->gout_par : nullptr, !mixed);
    DEC_TRY(hipGetLastError());
    if (d->fin_first_pass != d->fin) {
        hipLaunchKernelGGL(permute_kernel, grid(NC), dim3(256), 0, st, d->fin, d->tile_order, d->fin_first_pass, (uint32_t)NC);
        DEC_TRY(hipGetLastError());
    }
    if (split_groups != 0 && d->split) {
        // (fastecc_repair in the (2k,k) layout: the lost parity blocks' factors too — gout_par is filled above for such patterns)
        const bool with_parity = parity_factors && d->gout_par != nullptr;
        hipLaunchKernelGGL(split_rows_kernel, grid(N), dim3(256), 0, st, d->fin, d->gout, d->split_order, d->split_rows_data, d->split_rows_parity, d->split_rows_out,
                           (uint32_t)N, with_parity ? d->gout_par : nullptr, with_parity ? d->split_rows_out_parity : nullptr);
        d->split_repair_ready = with_parity;
        DEC_TRY(hipGetLastError());
        d->split_shift = split_shift;
        if (split_shift == 0 && d->split_dirty > split_groups) {
            // rows of groups this pattern does not write any more: group g = blocks g + (t << 10)
            const size_t row = ci.words * 4;
            DEC_TRY(hipMemset2DAsync(d->split_r1 + (size_t)split_groups * ci.words, 1024 * row, 0, (d->split_dirty - split_groups) * row,
                                     split_decode_group_rows(d->split), st));
            d->split_dirty = split_groups;
        }
        d->split_groups = split_shift == 0 ? split_groups : 0;
        d->split_ready = true;
    }
    DEC_TRY(hipStreamSynchronize(st));
    pt.mark("this pattern (device)");
    d->ready = true;
    return FASTECC_OK;
}

static int decode_impl(fastecc_ctx* c, void* data, const void* parity, int mem_kind, void* stream, void* parity_out);

int fastecc_decode(fastecc_ctx* c, void* data, const void* parity, int mem_kind, void* stream)
{
    try {
        return decode_impl(c, data, parity, mem_kind, stream, nullptr);
    } catch (const std::bad_alloc&) {
        return FASTECC_E_NOMEM;
    } catch (...) {
        return FASTECC_E_DEVICE;
    }
}

int fastecc_repair(fastecc_ctx* c, void* data, void* parity, int mem_kind, void* stream)
{
    try {
        return decode_impl(c, data, parity, mem_kind, stream, parity);
    } catch (const std::bad_alloc&) {
        return FASTECC_E_NOMEM;
    } catch (...) {
        return FASTECC_E_DEVICE;
    }
}

// parity_out != null (== parity): also rebuild the lost parity blocks from the repaired data
static int decode_impl(fastecc_ctx* c, void* data, const void* parity, int mem_kind, void* stream, void* parity_out)
{
    if (!c || !data || !parity || (((uintptr_t)data | (uintptr_t)parity) & 3u)) return FASTECC_E_INVAL;
    if (sharded_of(c)) return sharded_decode_stripe(c, data, const_cast<void*>(parity), mem_kind, parity_out != nullptr, (hipStream_t)stream);
    if (mem_kind == FASTECC_MEM_HOST_PINNED) mem_kind = FASTECC_MEM_HOST;  // the same staging; the copies are simply faster from pinned memory
    if (mem_kind != FASTECC_MEM_HOST && mem_kind != FASTECC_MEM_DEVICE) return FASTECC_E_INVAL;
    CallScope call(c);
    if (info_of(c).field == FASTECC_FIELD_GF_P61_SQUARED) {
        if ((((uintptr_t)data | (uintptr_t)parity) & 15u)) return FASTECC_E_INVAL;
        p61::Decoder* d61 = decoder61_of(c);
        if (!p61::decoder_ready(d61)) return FASTECC_E_INVAL;
        DeviceScope ds61(info_of(c).device);
        if (!ds61.ok) return FASTECC_E_DEVICE;
        int rc61 = call.begin((hipStream_t)stream);  // the decoder's work stripe and tables are internal buffers
        if (rc61 != FASTECC_OK) return rc61;
        const CtxInfo ci61 = info_of(c);
        void* prof61 = nullptr;
        const p61::LaunchHooks* hooks61 = p61_profile_hooks(c, &prof61);
        if (ci61.zero_extended) {
            // codes other than (2N,N): decode the padded (2N,N) codeword in the context's two work stripes and copy the caller's blocks back
            if (mem_kind != FASTECC_MEM_DEVICE) {
                (void)call.end((hipStream_t)stream);
                return FASTECC_E_UNSUPPORTED;
            }
            hipStream_t st61 = (hipStream_t)stream;
            const size_t row = ci61.words * 4, prow = row * (size_t)ci61.p61_stride;
            uint64_t *wd = nullptr, *wp = nullptr;
            rc61 = p61_work_stripes(c, &wd, &wp);
            auto step = [&](hipError_t e, const char* what) {
                if (rc61 == FASTECC_OK && e != hipSuccess) rc61 = hip_code(what, e);
            };
            if (rc61 == FASTECC_OK) {
                step(hipMemcpyAsync(wd, data, ci61.user_k * row, hipMemcpyDeviceToDevice, st61), "hipMemcpyAsync(data)");
                step(hipMemsetAsync((char*)wd + ci61.user_k * row, 0, (ci61.k - ci61.user_k) * row, st61), "hipMemsetAsync");
                step(hipMemcpy2DAsync(wp, prow, parity, row, row, ci61.user_m, hipMemcpyDeviceToDevice, st61), "hipMemcpy2DAsync(parity)");
            }
            if (rc61 == FASTECC_OK) rc61 = p61::decode(d61, wd, wp, parity_out ? p61_path_of(c) : nullptr, st61, hooks61);
            if (rc61 == FASTECC_OK) {
                step(hipMemcpyAsync(data, wd, ci61.user_k * row, hipMemcpyDeviceToDevice, st61), "hipMemcpyAsync(data back)");
                if (parity_out) step(hipMemcpy2DAsync(parity_out, row, wp, prow, row, ci61.user_m, hipMemcpyDeviceToDevice, st61), "hipMemcpy2DAsync(parity back)");
            }
            p61_profile_done(prof61);
            const int rc_end61 = call.end((hipStream_t)stream);
            return rc61 != FASTECC_OK ? rc61 : rc_end61;
        }
        rc61 = mem_kind == FASTECC_MEM_DEVICE
                   ? p61::decode(d61, (uint64_t*)data, (uint64_t*)const_cast<void*>(parity), parity_out ? p61_path_of(c) : nullptr, (hipStream_t)stream, hooks61)
                   : p61::decode_host(d61, data, const_cast<void*>(parity), parity_out ? p61_path_of(c) : nullptr, (hipStream_t)stream, hooks61);
        p61_profile_done(prof61);
        const int rc_end = call.end((hipStream_t)stream);
        return rc61 != FASTECC_OK ? rc61 : rc_end;
    }
    DecodeState* d = decoder_of(c);
    if (!d || !d->ready) return FASTECC_E_INVAL;  // fastecc_decode_prepare first
    const bool rebuild = parity_out != nullptr && d->erased_parity != 0;
    if (d->erased_data == 0 && !rebuild) return FASTECC_OK;
    const CtxInfo ci = info_of(c);
    if (ci.pitch != ci.words) return FASTECC_E_UNSUPPORTED;  // the gather / scatter passes address contiguous stripes
    DeviceScope ds(ci.device);
    if (!ds.ok) return FASTECC_E_DEVICE;
    hipStream_t st = (hipStream_t)stream;
    struct Marker {  // the decoder's work stripes are internal buffers: order their uses between streams
        CallScope& s;
        hipStream_t st;
        ~Marker() { (void)s.end(st); }
    };
    {
        const int rc0 = call.begin(st);
        if (rc0 != FASTECC_OK) return rc0;
    }
    Marker marker{call, st};
    const uint64_t N = d->mixed ? (uint64_t)ci.q * ci.k : ci.k;
    const size_t block = ci.words * 4, data_bytes = ci.user_k * block, parity_bytes = ci.user_m * block;

    uint32_t* ddata = (uint32_t*)data;
    const uint32_t* dparity = (const uint32_t*)parity;
    bool host_rows_only = false;  // FASTECC_MEM_HOST: only the rebuilt blocks are copied back (few enough of them, their rows known)
    if (mem_kind == FASTECC_MEM_HOST) {
        // stage the codeword — of the parity stripe only what the decoder will read where that is known to be a few block groups (the split
        // transform of the (2k,k) layout: groups g < split_groups = blocks g + 1024 t, one strided copy)
        if (!d->parity_dev) DEC_TRY(hipMalloc((void**)&d->parity_dev, parity_bytes + data_bytes));
        // the rows that will travel back: known from the set-up (few losses), else read off the decoder's tables once per pattern
        if (d->host_lists_of != d->pattern_serial) {
            d->host_lost_data.clear();
            d->host_lost_parity.clear();
            if (d->erased_data + d->erased_parity <= (ci.user_k + ci.user_m) / 8 && d->parity_lost && (d->erased_data == 0 || d->gout)) {
                std::vector<uint32_t> flags(std::max(ci.user_k, ci.user_m));
                if (d->erased_data != 0) {
                    DEC_TRY(hipMemcpy(flags.data(), d->gout, ci.user_k * 4, hipMemcpyDeviceToHost));
                    for (uint64_t i = 0; i < ci.user_k; i++)
                        if (flags[i] != 0) d->host_lost_data.push_back((uint32_t)i);
                }
                DEC_TRY(hipMemcpy(flags.data(), d->parity_lost, ci.user_m * 4, hipMemcpyDeviceToHost));
                for (uint64_t q = 0; q < ci.user_m; q++)
                    if (flags[q] != 0) d->host_lost_parity.push_back((uint32_t)q);
                if (d->host_lost_data.size() != d->erased_data) d->host_lost_data.clear(), d->host_lost_parity.clear();  // (tables of another shape: whole stripes back)
            }
            d->host_lists_of = d->pattern_serial;
        }
        const uint64_t back = (d->erased_data != 0 ? d->host_lost_data.size() : 0) + (rebuild ? d->host_lost_parity.size() : 0);
        host_rows_only = back != 0 && (d->erased_data == 0 || d->host_lost_data.size() == d->erased_data) &&
                         (!rebuild || d->host_lost_parity.size() == d->erased_parity) && back <= (ci.user_k + ci.user_m) / 8;
        // (a repair that copies the whole parity stripe back must have staged all of it)
        if (!d->sub && d->split_ready && d->split_shift != 0 && d->standard && d->erased_data != 0 && (!rebuild || host_rows_only)) {
            // small form: the parity blocks at multiples of 2^shift are all the decoder reads — one strided copy of every 2^shift-th block
            const size_t pitch = block << d->split_shift;
            DEC_TRY(hipMemcpy2DAsync(d->parity_dev, pitch, parity, pitch, block, (ci.user_m + (1ull << d->split_shift) - 1) >> d->split_shift, hipMemcpyHostToDevice, st));
        } else if (!d->sub && d->split_ready && d->split_shift == 0 && d->standard && d->erased_data != 0 && d->split_groups < 512 && (!rebuild || host_rows_only)) {
            DEC_TRY(hipMemcpy2DAsync(d->parity_dev, 1024 * block, parity, 1024 * block, (size_t)d->split_groups * block, ci.user_m / 1024, hipMemcpyHostToDevice, st));
        } else if (d->sub && (!rebuild || host_rows_only)) {
            // few losses: the direct path reads as many parity blocks as data blocks are lost (at most 256 copies of a block)
            for (uint32_t q : d->host_parity_used)
                DEC_TRY(hipMemcpyAsync(d->parity_dev + (size_t)q * ci.words, (const char*)parity + (size_t)q * block, block, hipMemcpyHostToDevice, st));
        } else {
            DEC_TRY(hipMemcpyAsync(d->parity_dev, parity, parity_bytes, hipMemcpyHostToDevice, st));
        }
        DEC_TRY(hipMemcpyAsync(d->parity_dev + ci.user_m * ci.words, data, data_bytes, hipMemcpyHostToDevice, st));
        dparity = d->parity_dev;
        ddata = d->parity_dev + ci.user_m * ci.words;
    }

    if (d->sub) {
        // any layout, few losses: the lost data from the surviving data + a few parity blocks, then (repair) the lost parity from the data
        const uint32_t S = (uint32_t)ci.words;
        uint32_t* dpar_out = mem_kind == FASTECC_MEM_HOST ? d->parity_dev : (uint32_t*)parity_out;
        // (profile: one "direct_pass" per read of the stripe)
        auto pass = [&](DirectPass* p, const uint32_t* par_in, uint32_t* data_to, uint32_t* par_to) -> int {
            void* scope = profile_scope_begin(c, st, "direct_pass", (ci.user_k + (uint64_t)d->sub_lost_data) * block);
            const int rc = direct_run(p, ddata, par_in, data_to, par_to, S, d->direct_kernel, st);
            profile_scope_end(scope);
            return rc;
        };
        if (d->sub_both && (rebuild || d->sub_only_both)) {
            // data and parity lost: one pass over the survivors writes both (fastecc_decode: the data only)
            const int rc = pass(d->direct_both, dparity, ddata, rebuild ? dpar_out : nullptr);
            if (rc != FASTECC_OK) return rc;
        } else {
            if (d->sub_lost_data > 0) {
                const int rc = pass(d->direct_data, dparity, ddata, nullptr);
                if (rc != FASTECC_OK) return rc;
            }
            if (rebuild) {
                const int rc = pass(d->direct_parity, nullptr, nullptr, dpar_out);
                if (rc != FASTECC_OK) return rc;
            }
        }
    } else {
    bool repaired_in_one_pass = false;
    // split transform, small form: the rows of the parity half in use (multiples of 2^shift), times l, and their stand-alone DIF — in place
    auto small_parity_half = [&]() -> int {
        const uint32_t S0 = (uint32_t)ci.words, rows = (uint32_t)(N >> d->split_shift);
        const bool w4 = (S0 % 4) == 0 && ((((uintptr_t)dparity | (uintptr_t)d->split_small_buf) & 15u) == 0);
        const uint32_t chunks = (S0 + (w4 ? 256 : 64) - 1) / (w4 ? 256 : 64);
        const uint64_t items = (uint64_t)rows * chunks;
        const dim3 grid((unsigned)((items + 3) / 4));
        if (w4) hipLaunchKernelGGL(split_small_gather_kernel<4>, grid, dim3(256), 0, st, dparity, d->split_small_buf, d->fin, S0, (int)d->split_shift, ci.fold, (uint32_t)ci.user_m, chunks, items);
        else    hipLaunchKernelGGL(split_small_gather_kernel<1>, grid, dim3(256), 0, st, dparity, d->split_small_buf, d->fin, S0, (int)d->split_shift, ci.fold, (uint32_t)ci.user_m, chunks, items);
        DEC_TRY(hipGetLastError());
        return transform_bitrev(d->split_small[d->split_shift], d->split_small_buf, d->split_small_buf, false, true, S0, st);
    };
    if (rebuild && d->erased_data != 0 && d->split_ready && d->split_repair_ready) {
        // fastecc_repair through the split transform: the data chain as in fastecc_decode, and a second MID + DIT over the same two halves for
        // x p'(x) at the odd positions — the lost parity blocks, written straight into the parity stripe.  (No room for the extra k-block
        // stripe: the forms below.)
        hipError_t e_alloc = hipSuccess;
        if (!d->split_q2) e_alloc = hipMalloc((void**)&d->split_q2, N * block);
        if (e_alloc != hipSuccess) {
            (void)hipGetLastError();
            d->split_q2 = nullptr;
        } else {
            uint32_t* dpar_out = mem_kind == FASTECC_MEM_HOST ? d->parity_dev : (uint32_t*)parity_out;
            const SplitRepair odd{d->split_q2, d->split_pos_data_odd, d->split_rows_out_parity, dpar_out};
            d->split_dirty = std::max(d->split_dirty, d->split_groups);  // (before the launches: a failure half way must not hide written groups)
            void* scope = profile_scope_begin(c, st, "repair_split_transform", (5 * N + (uint64_t)d->split_groups * split_decode_group_rows(d->split)) * block);
            int rc;
            if (d->split_shift != 0) {
                rc = small_parity_half();
                if (rc == FASTECC_OK)
                    rc = run_split_decode(d->split, ddata, nullptr, d->split_rows_data, nullptr, 0, d->split_pos_parity, d->recovered, nullptr, nullptr, d->split_rows_out, ddata,
                                          nullptr, (uint32_t)ci.user_k, (uint32_t)ci.user_m, st, &odd, d->split_small_buf, d->split_shift);
            } else {
                rc = run_split_decode(d->split, ddata, dparity, d->split_rows_data, d->split_rows_parity, d->split_groups, d->split_pos_parity, d->recovered,
                                      d->split_r1, d->split_r2, d->split_rows_out, ddata, d->split_impulse, (uint32_t)ci.user_k, (uint32_t)ci.user_m, st, &odd);
            }
            profile_scope_end(scope);
            if (rc != FASTECC_OK && rc != FASTECC_E_UNSUPPORTED) return rc;
            if (rc == FASTECC_OK) repaired_in_one_pass = true;
        }
    }
    if (!repaired_in_one_pass && rebuild && d->erased_data != 0 && d->standard && d->transform_full && d->full_ok && d->gout_par) {
        // fastecc_repair, (2k,k) layout: x p'(x) at all 2k positions — the even rows give the lost data, the odd rows the lost parity
        const uint32_t S = (uint32_t)ci.words;
        hipError_t e_alloc = hipSuccess;
        if (!d->recovered_full) e_alloc = hipMalloc((void**)&d->recovered_full, d->positions * (size_t)S * 4);
        if (e_alloc != hipSuccess) (void)hipGetLastError();  // no room for the 2k-block stripe: the two-step form below
        const int rc = e_alloc == hipSuccess ? run_gathered(d->transform_full, ddata, dparity, d->fin_first_pass, d->recovered_full, st) : FASTECC_E_UNSUPPORTED;
        if (rc != FASTECC_OK && rc != FASTECC_E_UNSUPPORTED) return rc;
        if (rc == FASTECC_OK) {
            uint32_t* dpar_out = mem_kind == FASTECC_MEM_HOST ? d->parity_dev : (uint32_t*)parity_out;
            const bool v4 = (S % 4) == 0 && ((((uintptr_t)ddata | (uintptr_t)dpar_out | (uintptr_t)d->recovered_full) & 15u) == 0);
            const uint32_t col_chunks = (S + (v4 ? 256 : 64) - 1) / (v4 ? 256 : 64);
            const uint64_t items = N * col_chunks;
            const dim3 grid((unsigned)((items + 3) / 4));
            if (v4) {
                hipLaunchKernelGGL(decode_scatter_kernel<4>, grid, dim3(256), 0, st, d->recovered_full, ddata, d->gout, S, 2u * S, S, col_chunks, items);
                hipLaunchKernelGGL(decode_scatter_kernel<4>, grid, dim3(256), 0, st, d->recovered_full + S, dpar_out, d->gout_par, S, 2u * S, S, col_chunks, items);
            } else {
                hipLaunchKernelGGL(decode_scatter_kernel<1>, grid, dim3(256), 0, st, d->recovered_full, ddata, d->gout, S, 2u * S, S, col_chunks, items);
                hipLaunchKernelGGL(decode_scatter_kernel<1>, grid, dim3(256), 0, st, d->recovered_full + S, dpar_out, d->gout_par, S, 2u * S, S, col_chunks, items);
            }
            DEC_TRY(hipGetLastError());
            repaired_in_one_pass = true;
        }
    }
    if (d->erased_data != 0 && !repaired_in_one_pass) {
    // The (2k,k) layout lets the transform's first pass read the two halves of the codeword itself (no gather pass).
    // The other codes do not hold every position in memory: they take the table-driven gather, which never touches a
    // position whose factor is zero, instead of a tile that reads first and multiplies by zero afterwards.
    int rc = FASTECC_E_UNSUPPORTED;
    bool scattered = false;
    if (d->split_ready) {
        // two half-size transforms instead of one of size 2k (see "even / odd split")
        const uint32_t* parity_half = dparity;
        uint32_t parity_half_blocks = (uint32_t)ci.user_m;
        bool staged_ok = true;
        if (d->split_shift != 0) {
            // (the gather of the small form reads parity block (h >> fold) for position h itself: no staging stripe for codes with fewer parity blocks)
            void* scope = profile_scope_begin(c, st, "decode_split_transform", (3 * N + 3 * (N >> d->split_shift)) * block);
            rc = small_parity_half();
            if (rc == FASTECC_OK)
                rc = run_split_decode(d->split, ddata, nullptr, d->split_rows_data, nullptr, 0, d->split_pos_parity, d->recovered, nullptr, nullptr, d->split_rows_out, ddata,
                                      nullptr, (uint32_t)ci.user_k, (uint32_t)ci.user_m, st, nullptr, d->split_small_buf, d->split_shift);
            profile_scope_end(scope);
            scattered = rc == FASTECC_OK;
            staged_ok = false;  // (done, or unsupported: nothing more to try in this branch)
        } else
        if (ci.fold > 0 && !d->split_r0 && hipMalloc((void**)&d->split_r0, N * block) != hipSuccess) {
            (void)hipGetLastError();  // no room for the staging stripe: the 2k-point transform below
            d->split_r0 = nullptr;
            staged_ok = false;
        }
        void* scope = staged_ok ? profile_scope_begin(c, st, "decode_split_transform", (3 * N + (uint64_t)d->split_groups * split_decode_group_rows(d->split)) * block)
                                : nullptr;
        if (ci.fold > 0 && staged_ok) {
            // fewer parity blocks than data blocks: block j belongs at j << fold of the parity half — the blocks in use are copied there
            const uint32_t S0 = (uint32_t)ci.words;
            const bool w4 = (S0 % 4) == 0 && ((((uintptr_t)dparity | (uintptr_t)d->split_r0) & 15u) == 0);
            const uint32_t chunks = (S0 + (w4 ? 256 : 64) - 1) / (w4 ? 256 : 64);
            const uint64_t items = ci.user_m * chunks;
            const dim3 grid((unsigned)((items + 3) / 4));
            if (w4) hipLaunchKernelGGL(split_stage_kernel<4>, grid, dim3(256), 0, st, dparity, d->split_r0, d->fin, S0, ci.fold, chunks, items);
            else    hipLaunchKernelGGL(split_stage_kernel<1>, grid, dim3(256), 0, st, dparity, d->split_r0, d->fin, S0, ci.fold, chunks, items);
            parity_half = d->split_r0;
            parity_half_blocks = (uint32_t)N;
        }
        if (staged_ok) d->split_dirty = std::max(d->split_dirty, d->split_groups);  // (before the launches, as above)
        if (staged_ok) rc = run_split_decode(d->split, ddata, parity_half, d->split_rows_data, d->split_rows_parity, d->split_groups, d->split_pos_parity, d->recovered, d->split_r1,
                              d->split_r2, d->split_rows_out, ddata, d->split_impulse, (uint32_t)ci.user_k, parity_half_blocks, st);  // ... whose last pass writes the rebuilt blocks straight into the data stripe
        profile_scope_end(scope);
        scattered = rc == FASTECC_OK;
    }
    if (rc == FASTECC_E_UNSUPPORTED && d->standard) {
        void* scope = profile_scope_begin(c, st, "decode_transform_2k", 3 * N * block);
        rc = run_gathered(d->transform, ddata, dparity, d->fin_first_pass, d->recovered, st);
        profile_scope_end(scope);
    }
    const bool fused = rc == FASTECC_OK;
    if (!fused && rc != FASTECC_E_UNSUPPORTED) return rc;
    uint32_t* work = d->recovered;
    if (!d->mixed && !fused) {  // (the transform context's scratch stripe: only the unfused form gathers into it)
        rc = scratch_of(d->transform, &work);
        if (rc != FASTECC_OK) return rc;
    }
    const uint32_t S = (uint32_t)ci.words;
    const uint32_t ld_rec = d->mixed ? 2u * S : S;  // mixed radix: data position i is row 2i of the transformed work stripe
    const bool v4 = (S % 4) == 0 && ((((uintptr_t)ddata | (uintptr_t)dparity | (uintptr_t)work | (uintptr_t)d->recovered) & 15u) == 0);
    const uint32_t col_chunks = (S + (v4 ? 256 : 64) - 1) / (v4 ? 256 : 64);
    if (!fused) {
        const uint64_t items = d->positions * col_chunks;
        const dim3 grid((unsigned)((items + 3) / 4));
        if (v4) hipLaunchKernelGGL(decode_gather_kernel<4>, grid, dim3(256), 0, st, ddata, dparity, work, d->fin, d->srcmap, S, S, S, col_chunks, items);
        else    hipLaunchKernelGGL(decode_gather_kernel<1>, grid, dim3(256), 0, st, ddata, dparity, work, d->fin, d->srcmap, S, S, S, col_chunks, items);
        DEC_TRY(hipGetLastError());
        rc = fastecc_encode(d->transform, work, d->mixed ? work : d->recovered, FASTECC_MEM_DEVICE, st);
        if (rc != FASTECC_OK) return rc;
    }
    if (!scattered) {
        const uint64_t items = N * col_chunks;
        const dim3 grid((unsigned)((items + 3) / 4));
        if (v4) hipLaunchKernelGGL(decode_scatter_kernel<4>, grid, dim3(256), 0, st, d->recovered, ddata, d->gout, S, ld_rec, S, col_chunks, items);
        else    hipLaunchKernelGGL(decode_scatter_kernel<1>, grid, dim3(256), 0, st, d->recovered, ddata, d->gout, S, ld_rec, S, col_chunks, items);
        DEC_TRY(hipGetLastError());
    }
    }
    if (rebuild && !repaired_in_one_pass) {
        // the lost parity blocks are whatever the encoder makes of the (now complete) data: one more encode into a stripe
        // of the decoder's, from which only the lost blocks are copied — the surviving ones are left as they are
        if (!d->parity_again) DEC_TRY(hipMalloc((void**)&d->parity_again, parity_bytes));
        const int rc = encode_unlocked(c, ddata, d->parity_again, st);
        if (rc != FASTECC_OK) return rc;
        const uint32_t S = (uint32_t)ci.words;
        uint32_t* dpar_out = mem_kind == FASTECC_MEM_HOST ? d->parity_dev : (uint32_t*)parity_out;
        const bool v4 = (S % 4) == 0 && ((((uintptr_t)dpar_out | (uintptr_t)d->parity_again) & 15u) == 0);
        const uint32_t col_chunks = (S + (v4 ? 256 : 64) - 1) / (v4 ? 256 : 64);
        const uint64_t items = ci.user_m * col_chunks;
        const dim3 grid((unsigned)((items + 3) / 4));
        if (v4) hipLaunchKernelGGL(restore_parity_kernel<4>, grid, dim3(256), 0, st, d->parity_again, dpar_out, d->parity_lost, S, col_chunks, items);
        else    hipLaunchKernelGGL(restore_parity_kernel<1>, grid, dim3(256), 0, st, d->parity_again, dpar_out, d->parity_lost, S, col_chunks, items);
        DEC_TRY(hipGetLastError());
    }
    }  // transform path
    bool rows_copied = false;
    if (mem_kind == FASTECC_MEM_HOST && host_rows_only) {
        // the rebuilt blocks packed side by side on the device, one copy into a pinned landing buffer, and a memcpy per block on the host.
        // The repair has already run: if one of the buffers of this shortcut cannot be had, the whole-stripe copy below still delivers it.
        const uint32_t S = (uint32_t)ci.words;
        const uint64_t nd = d->erased_data != 0 ? d->host_lost_data.size() : 0, np = rebuild ? d->host_lost_parity.size() : 0;
        bool have = true;
        if (d->lost_rows_cap < nd + np) {
            if (d->lost_rows_dev) (void)hipFree(d->lost_rows_dev);
            d->lost_rows_dev = nullptr;
            d->lost_rows_cap = 0;
            if (hipMalloc((void**)&d->lost_rows_dev, (nd + np) * 4) == hipSuccess) d->lost_rows_cap = nd + np;
            else have = false;
        }
        if (have && d->pack_words < (nd + np) * S) {
            if (d->pack_dev) (void)hipFree(d->pack_dev);
            d->pack_dev = nullptr;
            d->pack_words = 0;
            if (hipMalloc((void**)&d->pack_dev, (nd + np) * S * 4) == hipSuccess) d->pack_words = (nd + np) * S;
            else have = false;
        }
        if (have && d->pack_host_words < (nd + np) * S) {
            if (d->pack_host) (void)hipHostFree(d->pack_host);
            d->pack_host = nullptr;
            d->pack_host_words = 0;
            if (hipHostMalloc((void**)&d->pack_host, (nd + np) * S * 4, hipHostMallocDefault) == hipSuccess) d->pack_host_words = (nd + np) * S;
            else have = false;
        }
        if (!have) {
            (void)hipGetLastError();  // out of memory for the shortcut only
        } else {
            if (nd) DEC_TRY(hipMemcpyAsync(d->lost_rows_dev, d->host_lost_data.data(), nd * 4, hipMemcpyHostToDevice, st));
            if (np) DEC_TRY(hipMemcpyAsync(d->lost_rows_dev + nd, d->host_lost_parity.data(), np * 4, hipMemcpyHostToDevice, st));
            const bool v4 = (S % 4) == 0 && ((((uintptr_t)ddata | (uintptr_t)d->parity_dev | (uintptr_t)d->pack_dev) & 15u) == 0);
            const uint32_t col_chunks = (S + (v4 ? 256 : 64) - 1) / (v4 ? 256 : 64);
            auto pack = [&](const uint32_t* stripe, const uint32_t* rows, uint32_t* out, uint64_t count) {
                const uint64_t items = count * col_chunks;
                const dim3 grid((unsigned)((items + 3) / 4));
                if (v4) hipLaunchKernelGGL(pack_rows_kernel<4>, grid, dim3(256), 0, st, stripe, rows, out, S, col_chunks, items);
                else    hipLaunchKernelGGL(pack_rows_kernel<1>, grid, dim3(256), 0, st, stripe, rows, out, S, col_chunks, items);
            };
            if (nd) pack(ddata, d->lost_rows_dev, d->pack_dev, nd);
            if (np) pack(d->parity_dev, d->lost_rows_dev + nd, d->pack_dev + nd * S, np);
            DEC_TRY(hipGetLastError());
            DEC_TRY(hipMemcpyAsync(d->pack_host, d->pack_dev, (nd + np) * (size_t)S * 4, hipMemcpyDeviceToHost, st));
            DEC_TRY(hipStreamSynchronize(st));
            for (uint64_t r = 0; r < nd; r++) memcpy((char*)data + (size_t)d->host_lost_data[r] * block, d->pack_host + r * S, block);
            for (uint64_t r = 0; r < np; r++) memcpy((char*)parity_out + (size_t)d->host_lost_parity[r] * block, d->pack_host + (nd + r) * S, block);
            rows_copied = true;
        }
    }
    if (mem_kind == FASTECC_MEM_HOST && !rows_copied) {
        if (d->erased_data != 0) {
            const int rc = download_pageable(c, data, ddata, data_bytes, st);
            if (rc != FASTECC_OK) return rc;
        }
        if (rebuild) {
            const int rc = download_pageable(c, parity_out, d->parity_dev, parity_bytes, st);
            if (rc != FASTECC_OK) return rc;
        }
        DEC_TRY(hipStreamSynchronize(st));
    }
    return FASTECC_OK;
}

}  // extern "C"
