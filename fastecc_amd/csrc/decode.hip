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
    uint32_t* recovered = nullptr;     // k blocks: x p'(x) at the data positions
    uint32_t* parity_dev = nullptr;    // staging for FASTECC_MEM_HOST calls (lazy)
    // fastecc_decode_prepare's device state (lazy): the product tree of the locator
    uint64_t tree_T = 0;                   // padded number of roots: the smallest power of two >= the most losses a code tolerates
    std::vector<fastecc_ctx*> tree_ctx;    // level k (polynomials of degree d = 2^k): transforms of length 2d, T/d columns
    uint32_t* tree_x = nullptr;            // 2T words: the level's polynomials, [coefficient][polynomial]
    uint32_t* tree_f = nullptr;            // 2T words: their transforms
    uint32_t* tree_y = nullptr;            // 2T words: the next level's polynomials (swaps roles with tree_x)
    uint32_t* tree_p = nullptr;            // 2T words: pairwise products
    uint32_t* wpow = nullptr;              // NC words: w^u (plain)
    uint32_t* roots = nullptr;             // T words: the erased points, zero-padded
    uint32_t* dev_state = nullptr;         // NC bytes (as words/4): LOST / HELD / ZERO per position
    uint32_t* dev_erased = nullptr;        // T words: erased positions
    uint32_t* tile_order = nullptr;        // NC words: first-pass order of the factors (only for the (2k,k) layout)
    bool tile_order_valid = false;
    // fastecc_repair: which parity blocks are lost (one word each), and the stripe the re-encode writes to
    uint32_t* parity_lost = nullptr;
    uint32_t* parity_again = nullptr;
    uint64_t erased_parity = 0;
    uint64_t erased_data = 0, erased_total = 0;
    // few losses: every lost block is a fixed linear combination of surviving ones (the direct path)
    uint32_t* direct_coef = nullptr;   // [K + lost data][pad]: weight of data block i (then of the parity blocks used as nodes) in lost data block r, Montgomery
    uint32_t* direct_partial = nullptr;  // [row chunks + 32][pad][words]: partial sums, and the staging rows of the second summation step
    uint64_t direct_partial_words = 0;
    // any layout (the reference's, zero extension, sub-/extra cosets, mixed radix): interpolation on N nodes — the surviving data
    // points plus as many surviving parity points as data blocks are lost — then, for repair, the lost parity from the complete data
    int sub_lost_data = 0, sub_lost_parity = 0, sub_pad_data = 0, sub_pad_parity = 0;
    bool sub = false;
    uint32_t* sub_coef_parity = nullptr;  // [K][sub_pad_parity]: lost parity block t from data block i
    uint32_t* sub_lists = nullptr;        // 4 x 16 words: parity rows used as nodes | data output positions | parity output positions | spare
    uint32_t* sub_params = nullptr;       // 8 x 16 words of field elements for the coefficient kernels
    uint64_t direct_coef_words = 0;
    uint64_t positions = 0;            // code length on the roots of unity: k << log2(n / k) rounded up to powers of two
    bool mixed = false;                // mixed-radix code: `recovered` is the whole work stripe (all positions), transformed in place
    bool standard = false;             // the reference's (2k,k) layout: position u = data u/2 or parity u/2, every block in memory
    bool ready = false;
};

void destroy_decode_state(DecodeState* d)
{
    if (!d) return;
    if (d->transform) fastecc_destroy(d->transform);
    if (d->pattern_ntt) fastecc_destroy(d->pattern_ntt);
    if (d->pattern_buf) (void)hipFree(d->pattern_buf);
    if (d->fin_first_pass && d->fin_first_pass != d->fin) (void)hipFree(d->fin_first_pass);
    if (d->fin) (void)hipFree(d->fin);
    if (d->srcmap) (void)hipFree(d->srcmap);
    if (d->gout) (void)hipFree(d->gout);
    if (d->recovered) (void)hipFree(d->recovered);
    if (d->parity_dev) (void)hipFree(d->parity_dev);
    for (fastecc_ctx* t : d->tree_ctx)
        if (t) fastecc_destroy(t);
    for (uint32_t* b : {d->sub_coef_parity, d->sub_lists, d->sub_params, d->direct_coef, d->direct_partial, d->parity_lost, d->parity_again, d->tree_x, d->tree_f, d->tree_y, d->tree_p, d->wpow, d->roots, d->dev_state, d->dev_erased, d->tile_order})
        if (b) (void)hipFree(b);
    delete d;
}

namespace {

// ------------------------------------------------------------------------------------------------
// fastecc_decode_prepare on the device.  All values are plain representatives unless a table is consumed by
// gf::mul_mont, in which case it is stored in Montgomery form (x * 2^32 mod p = gf::mul(x, MONT_ONE)).
// ------------------------------------------------------------------------------------------------
enum : uint32_t { ST_LOST = 0, ST_HELD = 1, ST_ZERO = 2 };
constexpr int LEAF_LOG = 4, LEAF = 1 << LEAF_LOG;  // the lowest levels of the tree are one schoolbook kernel: 16 roots per thread

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
                                                            uint32_t NC, uint32_t pad, int e, uint32_t user_k, uint32_t q, int lg2)
{
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= NC) return;
    const uint32_t st = (state[u >> 2] >> (8 * (u & 3u))) & 0xFFu;
    const uint32_t back = (uint32_t)(((uint64_t)u * pad) % NC);
    const uint32_t corr = wpow[back == 0 ? 0 : NC - back];  // w^(-u pad)
    // where the two values for w^u sit in lv: position u for the power-of-two transform (natural order out); the way down
    // of a mixed-radix context (mixed_dif) leaves the value at w^(-v), v = q * bitrev(r) + j1, in block j1 * 2^lg2 + r
    uint32_t at = u;
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
    }
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

// ------------------------------------------------------------------------------------------------
// Few losses: the lost blocks are fixed linear combinations of surviving ones — no locator tree, no transform, one read of N + |lost data|
// blocks (sub_coef_*_kernel below state the weights).  Partial sums over 512 rows per wave, then a two-step sum.
// ------------------------------------------------------------------------------------------------
constexpr int DIRECT_MAX = 16;        // lost blocks per pattern on this path
constexpr uint32_t DIRECT_ROWS = 512; // codeword positions per partial sum

// partial[chunk][j][col] = sum over the chunk's positions u of block(u)[col] * coef[u][j]; a wave owns (chunk, 64*V-word column
// chunk) and keeps four rows in flight
template <int EB, int V>
__global__ __launch_bounds__(256) void direct_accumulate_kernel(const uint32_t* __restrict__ data, const uint32_t* __restrict__ parity,
                                                                const uint32_t* __restrict__ coef, uint32_t* __restrict__ partial, uint32_t S,
                                                                uint32_t NC, uint32_t col_chunks, uint64_t items, const uint32_t* __restrict__ extra = nullptr,
                                                                uint32_t data_rows = 0)
{
    // parity == nullptr: the NC "positions" are the rows of `data` (the encoder for few parity blocks, below); extra != nullptr:
    // rows [0, data_rows) of `data`, then the rows extra[0..] of `parity` (the decoder for any layout)
    constexpr int U = V == 4 ? 4 : 8;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t item = (uint64_t)blockIdx.x * 4u + wave;
    if (item >= items) return;
    const uint32_t cc = (uint32_t)(item % col_chunks);
    const uint32_t chunk = (uint32_t)(item / col_chunks);
    const uint32_t col = (cc * 64u + lane) * V;
    const bool live = col < S;
    uint32_t acc[EB][V];
#pragma unroll
    for (int j = 0; j < EB; ++j)
#pragma unroll
        for (int v = 0; v < V; ++v) acc[j][v] = 0;
    const uint32_t u0 = chunk * DIRECT_ROWS, u1 = min(u0 + DIRECT_ROWS, NC);
    for (uint32_t ub = u0; ub < u1; ub += U) {
        uint32_t w[U][EB], x[U][V];
        bool use[U];
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const uint32_t u = ub + i;
            uint32_t any = 0;
            if (u < u1) {
                const_u32_ptr cf = as_constant(coef) + (size_t)u * EB;
#pragma unroll
                for (int j = 0; j < EB; ++j) any |= (w[i][j] = cf[j]);
            }
            use[i] = any != 0;  // a lost block has no coefficients: whatever is stored in its place is not used (wave-uniform)
#pragma unroll
            for (int v = 0; v < V; ++v) x[i][v] = 0;
            // the load does not wait for the coefficients: all U rows are in flight at once
            if (u < u1 && live) {
                const uint32_t* row;
                if (extra) row = u < data_rows ? data + (size_t)u * S : parity + (size_t)as_constant(extra)[u - data_rows] * S;
                else if (parity) row = ((u & 1u) ? parity : data) + (size_t)(u >> 1) * S;
                else row = data + (size_t)u * S;
                load_vec<V>(x[i], row + col);
            }
        }
#pragma unroll
        for (int i = 0; i < U; ++i) {
            if (!use[i]) continue;
#pragma unroll
            for (int j = 0; j < EB; ++j)
#pragma unroll
                for (int v = 0; v < V; ++v) acc[j][v] = gf::add(acc[j][v], gf::mul_mont(x[i][v], w[i][j]));
        }
    }
    if (live) {
#pragma unroll
        for (int j = 0; j < EB; ++j) store_vec<V>(partial + ((size_t)chunk * EB + j) * S + col, acc[j]);
    }
}

// Sum of the partial sums in two steps.  Step 1 (to == nullptr ... see args): segment `seg` of the chunks -> stage[seg][j][col];
// step 2: the DIRECT_SEGS stage rows -> lost block j, written where it belongs: data (even positions) or — only for repair — parity.
constexpr uint32_t DIRECT_SEGS = 32;
__global__ __launch_bounds__(256) void direct_reduce1_kernel(const uint32_t* __restrict__ partial, uint32_t* __restrict__ stage, uint32_t S, uint32_t chunks,
                                                             int pad, int e)
{
    const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    const uint32_t seg = blockIdx.z;
    if (col >= S || j >= e) return;
    const uint32_t per = (chunks + DIRECT_SEGS - 1) / DIRECT_SEGS;
    const uint32_t c0 = seg * per, c1 = min(c0 + per, chunks);
    uint32_t v = 0;
#pragma unroll 8
    for (uint32_t c = c0; c < c1; ++c) v = gf::add(v, partial[((size_t)c * pad + j) * S + col]);
    stage[((size_t)seg * pad + j) * S + col] = v;
}
__global__ __launch_bounds__(256) void direct_reduce2_kernel(const uint32_t* __restrict__ stage, const uint32_t* __restrict__ epos, uint32_t* __restrict__ data,
                                                             uint32_t* __restrict__ parity, uint32_t S, int pad, int e, bool with_parity)
{
    const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (col >= S || j >= e) return;
    const uint32_t pos = epos ? epos[j] : 2u * (uint32_t)j + 1u;  // no list: output j is row j of `parity` (the encoder below)
    if ((pos & 1u) && !with_parity) return;
    uint32_t v = 0;
#pragma unroll
    for (uint32_t g = 0; g < DIRECT_SEGS; ++g) v = gf::add(v, stage[((size_t)g * pad + j) * S + col]);
    ((pos & 1u) ? parity : data)[(size_t)(pos >> 1) * S + col] = v;
}

// Any layout with the data at the N-th roots of unity x_i = w^(i << e) (N a power of two or q 2^m), R the lost data blocks and A as many
// surviving parity points y_a: the N nodes {x_i : i not in R} + {y_a} interpolate f, and with l(x) = (x^N - 1) A(x) / R(x),
// A(x) = prod_a (x - y_a), R(x) = prod_r (x - x_r), R_r = R / (x - x_r):
//     weight of data block i in lost block r   = -(x_i / x_r) * A(x_r) / R_r(x_r) * R_r(x_i) / A(x_i)
//     weight of parity block a in lost block r = N x_r^-1 A_a(x_r) R(y_a) / (R_r(x_r) (y_a^N - 1) A_a(y_a))     (host: sub-table rows K..)
// params: [0..16) x_r, [16..32) y_a, [32..48) C_r = -A(x_r) / (x_r R_r(x_r)), [48] = number of lost data blocks; all plain
__global__ __launch_bounds__(256) void sub_coef_data_kernel(uint32_t* __restrict__ coef, const uint32_t* __restrict__ wpow, const uint32_t* __restrict__ params,
                                                            const uint32_t* __restrict__ lost_rows, uint32_t K, int shift, int ed, int pad)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    bool lost = false;
    for (int r = 0; r < ed; ++r) lost = lost || lost_rows[r] == i;
    const uint32_t xi = wpow[(size_t)i << shift];
    uint32_t base = 0;
    if (!lost) {
        uint32_t a = 1;
        for (int t = 0; t < ed; ++t) a = gf::mul(a, gf::sub(xi, params[16 + t]));
        base = gf::mul(xi, dev_pow(a, gf::P - 2u));  // x_i / A(x_i)
    }
    for (int r = 0; r < pad; ++r) {
        uint32_t v = 0;
        if (!lost && r < ed) {
            v = gf::mul(params[32 + r], base);
            for (int t = 0; t < ed; ++t)
                if (t != r) v = gf::mul(v, gf::sub(xi, params[t]));
            v = gf::mul(v, gf::MONT_ONE);
        }
        coef[(size_t)i * pad + r] = v;
    }
}
// lost parity block t (point y_t) from the complete data: L_i(y_t) = (y_t^N - 1) x_i / (N (y_t - x_i)); params: [0..16) y_t, [16..32) (y_t^N - 1) / N
__global__ __launch_bounds__(256) void sub_coef_parity_kernel(uint32_t* __restrict__ coef, const uint32_t* __restrict__ wpow, const uint32_t* __restrict__ params,
                                                              uint32_t K, int shift, int ep, int pad)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    const uint32_t xi = wpow[(size_t)i << shift];
    for (int t = 0; t < pad; ++t) {
        uint32_t v = 0;
        if (t < ep) v = gf::mul(gf::mul(gf::mul(params[16 + t], xi), dev_pow(gf::sub(params[t], xi), gf::P - 2u)), gf::MONT_ONE);
        coef[(size_t)i * pad + t] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// The same idea for ENCODING when a code has few parity blocks (n - k <= 8): with the data points x_i = w_N^i and
// L_i(x) = (x^N - 1) x_i / (N (x - x_i)) the Lagrange basis, parity block j = f(y_j) = sum_i data_i * L_i(y_j), y_j = w_2N^(odd): y_j^N = -1,
// so coef[i][j] = -2 x_i / (N (y_j - x_i)) — one read of the data instead of three trips of the transform pipeline.  Exactly the
// polynomial evaluation the transform computes (RS.cpp:40-63), hence the same parity bits.
// ------------------------------------------------------------------------------------------------
constexpr int DIRECT_ENC_MAX = 8;
__global__ __launch_bounds__(256) void encode_coef_kernel(uint32_t* __restrict__ coef, uint32_t w2n, uint32_t minus_two_over_n, uint32_t K, int m, int pad, int fold)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    const uint32_t xi = dev_pow(w2n, 2u * i);
    for (int j = 0; j < pad; ++j) {
        uint32_t v = 0;
        if (j < m) {
            const uint32_t yj = dev_pow(w2n, (((uint32_t)j << fold) << 1) + 1u);
            v = gf::mul(gf::mul(minus_two_over_n, xi), dev_pow(gf::sub(yj, xi), gf::P - 2u));
            v = gf::mul(v, gf::MONT_ONE);
        }
        coef[(size_t)i * pad + j] = v;
    }
}

int hip_code(const char* what, hipError_t e)
{
    set_error_detail(what, e);
    return e == hipErrorOutOfMemory ? FASTECC_E_NOMEM : FASTECC_E_DEVICE;
}

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

struct DirectEncode {
    uint32_t* coef = nullptr;     // [K][pad], Montgomery form
    uint32_t* partial = nullptr;  // [chunks + DIRECT_SEGS][pad][S]
    uint32_t K = 0, S = 0;
    int m = 0, pad = 0;
};

void direct_encode_destroy(DirectEncode* de)
{
    if (!de) return;
    if (de->coef) (void)hipFree(de->coef);
    if (de->partial) (void)hipFree(de->partial);
    delete de;
}

int direct_encode_max() { return DIRECT_ENC_MAX; }

// N data points (a power of two, or q 2^m for the mixed-radix codes), K <= N existing data blocks, m <= 8 parity blocks at the odd positions ((j << fold) << 1) + 1 of the
// 2N-th roots of unity (fastecc_create's layout).  The current device is the context's.
int direct_encode_build(DirectEncode** out, uint64_t N, uint64_t K, uint64_t m, int fold, uint64_t words)
{
    *out = nullptr;
    if (m < 1 || m > DIRECT_ENC_MAX || K < 1 || K > N || N < 2 || ((gf::P - 1ull) % (2 * N)) != 0) return FASTECC_E_UNSUPPORTED;
    DirectEncode* de = new (std::nothrow) DirectEncode();
    if (!de) return FASTECC_E_NOMEM;
    de->K = (uint32_t)K;
    de->S = (uint32_t)words;
    de->m = (int)m;
    de->pad = 1;
    while (de->pad < de->m) de->pad <<= 1;
    const uint64_t chunks = (K + DIRECT_ROWS - 1) / DIRECT_ROWS;
    auto bail = [&](int rc) {
        direct_encode_destroy(de);
        return rc;
    };
    hipError_t e = hipMalloc((void**)&de->coef, K * de->pad * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&de->partial, (chunks + DIRECT_SEGS) * de->pad * words * 4);
    if (e != hipSuccess) return bail(hip_code("hipMalloc(direct encode)", e));
    const uint32_t w2n = gf::h_root((uint32_t)(2 * N));
    const uint32_t c0 = gf::h_mul(gf::P - 2u, gf::h_inv((uint32_t)(N % gf::P)));  // -2 / N
    hipLaunchKernelGGL(encode_coef_kernel, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, nullptr, de->coef, w2n, c0, (uint32_t)K, de->m, de->pad, fold);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) return bail(hip_code("encode_coef_kernel", e));
    *out = de;
    return FASTECC_OK;
}

// parity[j] = sum_i data[i] * coef[i][j]; data: K rows of S words, parity: m rows (may be the first m rows of data)
int direct_encode_run(DirectEncode* de, const uint32_t* data, uint32_t* parity, hipStream_t st)
{
    const uint32_t S = de->S, K = de->K;
    const bool v4 = (S % 4) == 0 && ((((uintptr_t)data | (uintptr_t)de->partial) & 15u) == 0);
    const uint32_t col_chunks = (S + (v4 ? 255u : 63u)) / (v4 ? 256u : 64u), chunks = (K + DIRECT_ROWS - 1) / DIRECT_ROWS;
    const uint64_t items = (uint64_t)chunks * col_chunks;
    const dim3 grid((unsigned)((items + 3) / 4));
#define FASTECC_DIRECT(EB, V) hipLaunchKernelGGL((direct_accumulate_kernel<EB, V>), grid, dim3(256), 0, st, data, (const uint32_t*)nullptr, de->coef, de->partial, S, K, col_chunks, items)
    switch (de->pad) {
        case 1: if (v4) FASTECC_DIRECT(1, 4); else FASTECC_DIRECT(1, 1); break;
        case 2: if (v4) FASTECC_DIRECT(2, 4); else FASTECC_DIRECT(2, 1); break;
        case 4: if (v4) FASTECC_DIRECT(4, 4); else FASTECC_DIRECT(4, 1); break;
        default: if (v4) FASTECC_DIRECT(8, 4); else FASTECC_DIRECT(8, 1); break;
    }
#undef FASTECC_DIRECT
    DEC_TRY(hipGetLastError());
    uint32_t* stage = de->partial + (size_t)chunks * de->pad * S;
    hipLaunchKernelGGL(direct_reduce1_kernel, dim3((S + 255) / 256, (unsigned)de->m, DIRECT_SEGS), dim3(256), 0, st, de->partial, stage, S, chunks, de->pad, de->m);
    hipLaunchKernelGGL(direct_reduce2_kernel, dim3((S + 255) / 256, (unsigned)de->m), dim3(256), 0, st, stage, (const uint32_t*)nullptr, (uint32_t*)nullptr, parity, S,
                       de->pad, de->m, true);
    DEC_TRY(hipGetLastError());
    return FASTECC_OK;
}

}  // namespace fastecc

using namespace fastecc;

extern "C" {

int fastecc_decode_prepare(fastecc_ctx* c, const uint8_t* data_present, const uint8_t* parity_present)
{
    if (!c || !data_present || !parity_present) return FASTECC_E_INVAL;
    if (sharded_of(c)) return sharded_decode_prepare(c, data_present, parity_present);
    const CtxInfo ci = info_of(c);
    if (ci.field == FASTECC_FIELD_GF_P61_SQUARED) {
        // the 64-bit field has its own decoder (gf61_decode.hip); its contexts are always (2k,k) with k a power of two
        DeviceScope ds61(ci.device);
        if (!ds61.ok) return FASTECC_E_DEVICE;
        CallScope call61(c);
        {
            const int rc0 = call61.wait_idle();  // a decode still using the previous pattern
            if (rc0 != FASTECC_OK) return rc0;
        }
        char detail[160] = "";
        const int rc = p61::decode_prepare(&decoder61_of(c), ci.log2k, ci.words / 4, data_present, parity_present, ci.direct_max, detail, sizeof detail);
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
    const int direct_limit = std::min(ci.direct_max, (int)DIRECT_MAX);
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
        std::vector<uint32_t> R, Pl, A;
        bool over = false;
        for (uint64_t i = 0; i < ci.user_k && !over; i++)
            if (!data_present[i]) R.push_back((uint32_t)i), over = (int)R.size() > direct_limit;
        for (uint64_t q = 0; q < ci.user_m && !over; q++)
            if (!parity_present[q]) Pl.push_back((uint32_t)q), over = (int)(R.size() + Pl.size()) > direct_limit;
        for (uint64_t q = 0; q < ci.user_m && !over && A.size() < R.size(); q++)
            if (parity_present[q]) A.push_back((uint32_t)q);
        if (!over && R.size() + Pl.size() >= 1 && A.size() == R.size()) {
            const int ed = (int)R.size(), ep = (int)Pl.size();
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
            const uint32_t K = (uint32_t)ci.user_k;
            int padd = 1, padp = 1;
            while (padd < ed) padd <<= 1;
            while (padp < ep) padp <<= 1;
            const uint32_t w = gf::h_root((uint32_t)NC);
            auto fsub = [](uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a + gf::P - b) % gf::P); };
            const uint32_t Nf = (uint32_t)(N % gf::P);
            std::vector<uint32_t> xr(ed), ya(ed), yaN1(ed), params(8 * 16, 0), lists(4 * 16, 0);
            for (int r = 0; r < ed; r++) xr[r] = gf::h_pow(w, (uint64_t)R[r] << e);
            for (int a = 0; a < ed; a++) {
                ya[a] = gf::h_pow(w, parity_position(A[a]));
                yaN1[a] = fsub(gf::h_pow(ya[a], N), 1u);
            }
            // per lost data block: C_r, and the weights of the parity nodes
            std::vector<uint32_t> node_rows((size_t)ed * padd, 0);
            for (int r = 0; r < ed; r++) {
                uint32_t Ar = 1, Rr = 1;
                for (int a = 0; a < ed; a++) Ar = gf::h_mul(Ar, fsub(xr[r], ya[a]));
                for (int t = 0; t < ed; t++)
                    if (t != r) Rr = gf::h_mul(Rr, fsub(xr[r], xr[t]));
                const uint32_t inv_xr_Rr = gf::h_inv(gf::h_mul(xr[r], Rr));
                params[r] = xr[r];
                params[32 + r] = fsub(0u, gf::h_mul(Ar, inv_xr_Rr));  // -A(x_r) / (x_r R_r(x_r))
                for (int a = 0; a < ed; a++) {
                    uint32_t Aa_xr = 1, Aa_ya = 1, R_ya = 1;
                    for (int t = 0; t < ed; t++) {
                        if (t != a) Aa_xr = gf::h_mul(Aa_xr, fsub(xr[r], ya[t])), Aa_ya = gf::h_mul(Aa_ya, fsub(ya[a], ya[t]));
                        R_ya = gf::h_mul(R_ya, fsub(ya[a], xr[t]));
                    }
                    const uint32_t num = gf::h_mul(gf::h_mul(Nf, Aa_xr), R_ya);
                    const uint32_t den = gf::h_mul(gf::h_mul(gf::h_mul(xr[r], Rr), yaN1[a]), Aa_ya);
                    node_rows[(size_t)a * padd + r] = gf::h_to_mont(gf::h_mul(num, gf::h_inv(den)));
                }
            }
            for (int a = 0; a < ed; a++) params[16 + a] = ya[a], lists[a] = A[a];
            for (int r = 0; r < ed; r++) lists[16 + r] = 2u * R[r];           // reduce: even "position" 2r -> data row r
            for (int t = 0; t < ep; t++) lists[32 + t] = 2u * Pl[t] + 1u;     // odd -> parity row
            const uint32_t inv_N = gf::h_inv(Nf);
            for (int t = 0; t < ep; t++) {
                const uint32_t yt = gf::h_pow(w, parity_position(Pl[t]));
                params[64 + t] = yt;
                params[80 + t] = gf::h_mul(fsub(gf::h_pow(yt, N), 1u), inv_N);  // (y_t^N - 1) / N
            }
            hipStream_t st = nullptr;
            if (!d->wpow) {
                DEC_TRY(hipMalloc((void**)&d->wpow, NC * 4));
                hipLaunchKernelGGL(wpow_kernel, dim3((unsigned)((NC + 255) / 256)), dim3(256), 0, st, d->wpow, w, (uint32_t)NC);
                DEC_TRY(hipGetLastError());
            }
            const uint64_t coef_words = ((uint64_t)K + DIRECT_MAX) * DIRECT_MAX;
            if (d->direct_coef_words < coef_words) {
                if (d->direct_coef) (void)hipFree(d->direct_coef);
                d->direct_coef = nullptr;
                d->direct_coef_words = 0;
                DEC_TRY(hipMalloc((void**)&d->direct_coef, coef_words * 4));
                d->direct_coef_words = coef_words;
            }
            if (!d->sub_coef_parity) DEC_TRY(hipMalloc((void**)&d->sub_coef_parity, (uint64_t)K * DIRECT_MAX * 4));
            if (!d->sub_lists) DEC_TRY(hipMalloc((void**)&d->sub_lists, 4 * 16 * 4));
            if (!d->sub_params) DEC_TRY(hipMalloc((void**)&d->sub_params, 8 * 16 * 4));
            const uint64_t chunks = ((uint64_t)K + DIRECT_MAX + DIRECT_ROWS - 1) / DIRECT_ROWS;
            const uint64_t need = (chunks + 32 /* DIRECT_SEGS */) * std::max(padd, padp) * ci.words;
            if (d->direct_partial_words < need) {
                if (d->direct_partial) (void)hipFree(d->direct_partial);
                d->direct_partial = nullptr;
                d->direct_partial_words = 0;
                DEC_TRY(hipMalloc((void**)&d->direct_partial, need * 4));
                d->direct_partial_words = need;
            }
            {
                const int rc = call.wait_idle();  // a decode still using the previous pattern
                if (rc != FASTECC_OK) return rc;
            }
            DEC_TRY(hipMemcpyAsync(d->sub_params, params.data(), params.size() * 4, hipMemcpyHostToDevice, st));
            DEC_TRY(hipMemcpyAsync(d->sub_lists, lists.data(), lists.size() * 4, hipMemcpyHostToDevice, st));
            std::vector<uint32_t> lost_rows(16, 0xFFFFFFFFu);  // (outlives the asynchronous copy below)
            if (ed > 0) {
                // lost data rows as a list for the kernel: the spare quarter of the lists
                for (int r = 0; r < ed; r++) lost_rows[r] = R[r];
                DEC_TRY(hipMemcpyAsync(d->sub_lists + 48, lost_rows.data(), 16 * 4, hipMemcpyHostToDevice, st));
                hipLaunchKernelGGL(sub_coef_data_kernel, dim3((K + 255) / 256), dim3(256), 0, st, d->direct_coef, d->wpow, d->sub_params, d->sub_lists + 48, K, e, ed,
                                   padd);
                DEC_TRY(hipGetLastError());
                DEC_TRY(hipMemcpyAsync(d->direct_coef + (size_t)K * padd, node_rows.data(), node_rows.size() * 4, hipMemcpyHostToDevice, st));
            }
            if (ep > 0) {
                hipLaunchKernelGGL(sub_coef_parity_kernel, dim3((K + 255) / 256), dim3(256), 0, st, d->sub_coef_parity, d->wpow, d->sub_params + 64, K, e, ep, padp);
                DEC_TRY(hipGetLastError());
            }
            DEC_TRY(hipStreamSynchronize(st));  // the host vectors above go out of scope
            d->sub = true;
            d->sub_lost_data = ed;
            d->sub_lost_parity = ep;
            d->sub_pad_data = padd;
            d->sub_pad_parity = padp;
            d->ready = true;
            return FASTECC_OK;
        }
    }
    enum : uint8_t { LOST = ST_LOST, HELD = ST_HELD, ZERO = ST_ZERO };
    std::vector<uint8_t> state(NC, LOST);
    std::vector<uint32_t> srcmap(NC, 0);
    uint64_t erased_data = 0;
    for (uint64_t i = 0; i < N; i++) {
        const uint64_t u = i << e;
        if (i >= ci.user_k) state[u] = ZERO;
        else if (data_present[i]) state[u] = HELD, srcmap[u] = (uint32_t)i;
        else erased_data++;
    }
    for (uint64_t q = 0; q < ci.user_m; q++) {
        const uint64_t u = parity_position(q);
        if (parity_present[q]) state[u] = HELD, srcmap[u] = (uint32_t)q | 0x80000000u;
    }
    std::vector<uint32_t> erased;
    for (uint64_t u = 0; u < NC; u++)
        if (state[u] == LOST) erased.push_back((uint32_t)u);
    if (erased.size() > NC - N) return FASTECC_E_INVAL;  // fewer than k blocks survive: not decodable

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
    d->erased_data = erased_data;
    d->erased_total = erased.size();
    d->positions = NC;
    d->standard = !mixed && ci.cosets == 1 && ci.fold == 0 && !ci.zero_extended;
    d->mixed = mixed;
    {
        std::vector<uint32_t> plost(ci.user_m);
        d->erased_parity = 0;
        for (uint64_t q = 0; q < ci.user_m; q++) d->erased_parity += (plost[q] = parity_present[q] ? 0u : 1u);
        if (!d->parity_lost) DEC_TRY(hipMalloc((void**)&d->parity_lost, ci.user_m * 4));
        const int rc = call.wait_idle();  // a repair still reading the previous pattern
        if (rc != FASTECC_OK) return rc;
        DEC_TRY(hipMemcpy(d->parity_lost, plost.data(), ci.user_m * 4, hipMemcpyHostToDevice));
    }
    d->sub = false;
    if (erased_data == 0) {  // no data block to recover
        d->ready = true;
        return FASTECC_OK;
    }

    // ---- device state of the decoder (built once) ----
    // T = padded root count: the smallest power of two that holds the most losses the code tolerates, NC - N
    uint64_t T = 1;
    while (T < NC - N) T <<= 1;
    int lgT = 0;
    while ((1ull << lgT) < T) lgT++;
    if (lgT > 20) return FASTECC_E_UNSUPPORTED;  // the top of the product tree is a cyclic product of length T: w_T must exist
    const int leaf_log = std::min(LEAF_LOG, lgT), leaf = 1 << leaf_log;
    const uint32_t w = gf::h_root((uint32_t)NC);
    hipStream_t st = nullptr;  // the set-up is synchronous: it runs on the default stream and ends with a synchronise
    if (!d->pattern_ntt) {
        const std::vector<uint32_t> ones(NC, 1u);
        const int rc = mixed ? create_mixed_transform_ctx(&d->pattern_ntt, ci.q, lgc, 8, ones.data(), ci.device)
                             : create_transform_ctx(&d->pattern_ntt, lgc, 8, 0, ones.data(), ci.device);
        if (rc != FASTECC_OK) return rc;
    }
    if (!d->pattern_buf) DEC_TRY(hipMalloc((void**)&d->pattern_buf, 2 * NC * 4));
    if (!d->transform) {
        std::vector<uint32_t> factor(NC);
        const uint32_t inv_nc = gf::h_inv((uint32_t)NC);
        for (uint64_t m = 0; m < NC; m++) factor[m] = gf::h_mul((uint32_t)m, inv_nc);  // x p'(x): coefficient m times m, and the 1/NC of the inverse transform
        // fold e: only the data positions (multiples of 2^e) are evaluated (mixed radix: all positions, the even ones are used)
        const int rc = mixed ? create_mixed_transform_ctx(&d->transform, ci.q, lgc, ci.words * 4, factor.data(), ci.device)
                             : create_transform_ctx(&d->transform, lgc, ci.words * 4, e, factor.data(), ci.device);
        if (rc != FASTECC_OK) return rc;
    }
    if (d->tree_T != T) {
        // level k >= leaf_log multiplies pairs of degree-2^k polynomials: transforms of length 2^(k+1) on T / 2^k columns
        for (fastecc_ctx* t : d->tree_ctx)
            if (t) fastecc_destroy(t);
        d->tree_ctx.assign(lgT, nullptr);
        const std::vector<uint32_t> ones((size_t)T, 1u);
        for (int k = leaf_log; k < lgT; k++) {
            const int rc = create_transform_ctx(&d->tree_ctx[k], k + 1, 4 * (T >> k), 0, ones.data(), ci.device);
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
        DEC_TRY(hipMalloc((void**)&d->dev_erased, T * 4));
        d->tree_T = T;
    }
    if (!d->wpow) {
        DEC_TRY(hipMalloc((void**)&d->wpow, NC * 4));
        hipLaunchKernelGGL(wpow_kernel, dim3((unsigned)((NC + 255) / 256)), dim3(256), 0, st, d->wpow, w, (uint32_t)NC);
        DEC_TRY(hipGetLastError());
    }
    if (!d->dev_state) DEC_TRY(hipMalloc((void**)&d->dev_state, NC));
    if (!d->fin) DEC_TRY(hipMalloc((void**)&d->fin, NC * 4));
    if (!d->srcmap) DEC_TRY(hipMalloc((void**)&d->srcmap, NC * 4));
    if (!d->gout) DEC_TRY(hipMalloc((void**)&d->gout, N * 4));
    // mixed radix: the work stripe of all NC positions, transformed in place; else the N recovered data positions
    if (!d->recovered) DEC_TRY(hipMalloc((void**)&d->recovered, (mixed ? NC : N) * ci.words * 4));
    if (d->standard && !d->tile_order_valid) {
        std::vector<uint32_t> order;
        if (gather_tile_order(d->transform, order)) {
            DEC_TRY(hipMalloc((void**)&d->tile_order, NC * 4));
            DEC_TRY(hipMemcpy(d->tile_order, order.data(), NC * 4, hipMemcpyHostToDevice));
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

    // ---- this pattern ----
    DEC_TRY(hipMemcpyAsync(d->dev_state, state.data(), NC, hipMemcpyHostToDevice, st));
    DEC_TRY(hipMemcpyAsync(d->srcmap, srcmap.data(), NC * 4, hipMemcpyHostToDevice, st));
    DEC_TRY(hipMemcpyAsync(d->dev_erased, erased.data(), erased.size() * 4, hipMemcpyHostToDevice, st));
    auto grid = [](uint64_t items) { return dim3((unsigned)((items + 255) / 256)); };
    hipLaunchKernelGGL(roots_kernel, grid(T), dim3(256), 0, st, d->roots, d->dev_erased, d->wpow, (uint32_t)erased.size(), (uint32_t)T);
    // leaves: T / leaf polynomials of degree `leaf`, side by side ([coefficient][polynomial]); the upper half of the
    // 2*leaf rows the first product needs is zero
    DEC_TRY(hipMemsetAsync(d->tree_x, 0, 2 * T * 4, st));
    hipLaunchKernelGGL(leaf_products_kernel, grid(T >> leaf_log), dim3(256), 0, st, d->roots, d->tree_x, (uint32_t)leaf, (uint32_t)(T >> leaf_log));
    DEC_TRY(hipGetLastError());
    uint32_t* x = d->tree_x;
    uint32_t* spare = d->tree_y;  // x / spare swap roles level by level; tree_f always holds the transforms
    for (int k = leaf_log; k < lgT; k++) {
        const uint64_t deg = 1ull << k, m = T >> k;  // m polynomials of degree deg in x: [2 deg][m], rows deg.. are zero
        fastecc_ctx* t = d->tree_ctx[k];
        int rc = transform_bitrev(t, x, d->tree_f, false, false, (uint32_t)m, st);                     // all of them at once
        if (rc != FASTECC_OK) return rc;
        const uint32_t scale = gf::h_to_mont(gf::h_inv((uint32_t)(2 * deg)));
        hipLaunchKernelGGL(pointwise_pairs_kernel, grid(2 * deg * (m / 2)), dim3(256), 0, st, d->tree_f, d->tree_p, (uint32_t)m, 2 * deg * (m / 2), scale);
        DEC_TRY(hipGetLastError());
        rc = transform_bitrev(t, d->tree_p, d->tree_p, true, true, (uint32_t)(m / 2), st);             // the products, back in natural order
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
        const int rc = mixed ? mixed_dif(d->pattern_ntt, d->pattern_buf, d->pattern_buf, st)
                             : fastecc_ntt(d->pattern_ntt, d->pattern_buf, 0, FASTECC_MEM_DEVICE, st);
        if (rc != FASTECC_OK) return rc;
    }
    hipLaunchKernelGGL(finish_tables_kernel, grid(NC), dim3(256), 0, st, d->pattern_buf, d->dev_state, d->wpow, d->fin, d->gout, (uint32_t)NC,
                       (uint32_t)(T - erased.size()), e, (uint32_t)ci.user_k, (uint32_t)(mixed ? ci.q : 1), lgc);
    DEC_TRY(hipGetLastError());
    if (d->fin_first_pass != d->fin) {
        hipLaunchKernelGGL(permute_kernel, grid(NC), dim3(256), 0, st, d->fin, d->tile_order, d->fin_first_pass, (uint32_t)NC);
        DEC_TRY(hipGetLastError());
    }
    DEC_TRY(hipStreamSynchronize(st));
    d->ready = true;
    return FASTECC_OK;
}

static int decode_impl(fastecc_ctx* c, void* data, const void* parity, int mem_kind, void* stream, void* parity_out);

int fastecc_decode(fastecc_ctx* c, void* data, const void* parity, int mem_kind, void* stream)
{
    return decode_impl(c, data, parity, mem_kind, stream, nullptr);
}

int fastecc_repair(fastecc_ctx* c, void* data, void* parity, int mem_kind, void* stream)
{
    return decode_impl(c, data, parity, mem_kind, stream, parity);
}

// parity_out != null (== parity): also rebuild the lost parity blocks from the repaired data
static int decode_impl(fastecc_ctx* c, void* data, const void* parity, int mem_kind, void* stream, void* parity_out)
{
    if (!c || !data || !parity || (((uintptr_t)data | (uintptr_t)parity) & 3u)) return FASTECC_E_INVAL;
    if (sharded_of(c)) return sharded_decode_stripe(c, data, const_cast<void*>(parity), mem_kind, parity_out != nullptr, (hipStream_t)stream);
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
        rc61 = mem_kind == FASTECC_MEM_DEVICE
                   ? p61::decode(d61, (uint64_t*)data, (uint64_t*)const_cast<void*>(parity), parity_out ? p61_path_of(c) : nullptr, (hipStream_t)stream, nullptr)
                   : p61::decode_host(d61, data, const_cast<void*>(parity), parity_out ? p61_path_of(c) : nullptr, (hipStream_t)stream, nullptr);
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
    if (mem_kind == FASTECC_MEM_HOST) {
        // stage both parts of the codeword
        if (!d->parity_dev) DEC_TRY(hipMalloc((void**)&d->parity_dev, parity_bytes + data_bytes));
        DEC_TRY(hipMemcpyAsync(d->parity_dev, parity, parity_bytes, hipMemcpyHostToDevice, st));
        DEC_TRY(hipMemcpyAsync(d->parity_dev + ci.user_m * ci.words, data, data_bytes, hipMemcpyHostToDevice, st));
        dparity = d->parity_dev;
        ddata = d->parity_dev + ci.user_m * ci.words;
    }

    if (d->sub) {
        // any layout, few losses: the lost data from the surviving data + a few parity blocks, then (repair) the lost parity from the data
        const uint32_t S = (uint32_t)ci.words, K = (uint32_t)ci.user_k;
        uint32_t* dpar_out = mem_kind == FASTECC_MEM_HOST ? d->parity_dev : (uint32_t*)parity_out;
        auto pass = [&](const uint32_t* coef, int pad, int outputs, uint32_t rows, const uint32_t* extra, const uint32_t* epos, bool to_parity) -> int {
            const bool v4 = (S % 4) == 0 && ((((uintptr_t)ddata | (uintptr_t)dparity | (uintptr_t)d->direct_partial) & 15u) == 0) && pad <= 8;
            const uint32_t col_chunks = (S + (v4 ? 255u : 63u)) / (v4 ? 256u : 64u), chunks = (rows + DIRECT_ROWS - 1) / DIRECT_ROWS;
            const uint64_t items = (uint64_t)chunks * col_chunks;
            const dim3 grid((unsigned)((items + 3) / 4));
            const uint32_t* par = extra ? dparity : nullptr;
#define FASTECC_DIRECT(EB, V) hipLaunchKernelGGL((direct_accumulate_kernel<EB, V>), grid, dim3(256), 0, st, ddata, par, coef, d->direct_partial, S, rows, col_chunks, items, extra, K)
            switch (pad) {
                case 1: if (v4) FASTECC_DIRECT(1, 4); else FASTECC_DIRECT(1, 1); break;
                case 2: if (v4) FASTECC_DIRECT(2, 4); else FASTECC_DIRECT(2, 1); break;
                case 4: if (v4) FASTECC_DIRECT(4, 4); else FASTECC_DIRECT(4, 1); break;
                case 8: if (v4) FASTECC_DIRECT(8, 4); else FASTECC_DIRECT(8, 1); break;
                default: FASTECC_DIRECT(16, 1); break;
            }
#undef FASTECC_DIRECT
            DEC_TRY(hipGetLastError());
            uint32_t* stage = d->direct_partial + (size_t)chunks * pad * S;
            hipLaunchKernelGGL(direct_reduce1_kernel, dim3((S + 255) / 256, (unsigned)outputs, DIRECT_SEGS), dim3(256), 0, st, d->direct_partial, stage, S, chunks, pad, outputs);
            hipLaunchKernelGGL(direct_reduce2_kernel, dim3((S + 255) / 256, (unsigned)outputs), dim3(256), 0, st, stage, epos, ddata, dpar_out, S, pad, outputs, to_parity);
            DEC_TRY(hipGetLastError());
            return FASTECC_OK;
        };
        if (d->sub_lost_data > 0) {
            const int rc = pass(d->direct_coef, d->sub_pad_data, d->sub_lost_data, K + (uint32_t)d->sub_lost_data, d->sub_lists, d->sub_lists + 16, false);
            if (rc != FASTECC_OK) return rc;
        }
        if (rebuild) {
            const int rc = pass(d->sub_coef_parity, d->sub_pad_parity, d->sub_lost_parity, K, nullptr, d->sub_lists + 32, true);
            if (rc != FASTECC_OK) return rc;
        }
    } else {
    if (d->erased_data != 0) {
    // The (2k,k) layout lets the transform's first pass read the two halves of the codeword itself (no gather pass).
    // The other codes do not hold every position in memory: they take the table-driven gather, which never touches a
    // position whose factor is zero, instead of a tile that reads first and multiplies by zero afterwards.
    int rc = d->standard ? run_gathered(d->transform, ddata, dparity, d->fin_first_pass, d->recovered, st) : FASTECC_E_UNSUPPORTED;
    const bool fused = rc == FASTECC_OK;
    if (!fused && rc != FASTECC_E_UNSUPPORTED) return rc;
    uint32_t* work = d->recovered;
    if (!d->mixed) {
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
    {
        const uint64_t items = N * col_chunks;
        const dim3 grid((unsigned)((items + 3) / 4));
        if (v4) hipLaunchKernelGGL(decode_scatter_kernel<4>, grid, dim3(256), 0, st, d->recovered, ddata, d->gout, S, ld_rec, S, col_chunks, items);
        else    hipLaunchKernelGGL(decode_scatter_kernel<1>, grid, dim3(256), 0, st, d->recovered, ddata, d->gout, S, ld_rec, S, col_chunks, items);
        DEC_TRY(hipGetLastError());
    }
    }
    if (rebuild) {
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
    if (mem_kind == FASTECC_MEM_HOST) {
        if (d->erased_data != 0) DEC_TRY(hipMemcpyAsync(data, ddata, data_bytes, hipMemcpyDeviceToHost, st));
        if (rebuild) DEC_TRY(hipMemcpyAsync(parity_out, d->parity_dev, parity_bytes, hipMemcpyDeviceToHost, st));
        DEC_TRY(hipStreamSynchronize(st));
    }
    return FASTECC_OK;
}

}  // extern "C"
