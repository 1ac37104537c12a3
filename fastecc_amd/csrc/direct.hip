// direct.hip — blocks as fixed linear combinations of other blocks: the decoder's few-loss path and the encoder for codes with few
// parity blocks (README.md:83-119 "recover the lost data", RS.md:42-79; RS.cpp:40-63 evaluates the same polynomial by transforms).
//
//   out[j][col] = sum_u  w[u][j] * block_u[col]   (mod p = 0xFFF00001),   j < E outputs,  u < rows,  col < S word columns
//
// is a dense contraction over the block index: a skinny matrix product [E x rows] . [rows x S] with exact modular arithmetic.  Two
// kernels compute it, both bit-exact (the sums are exact integers, reduced once):
//
//   * direct_accumulate_kernel (VALU): a wave owns a run of rows x 64*V columns; every product x*w goes into a 96-bit accumulator
//     with v_mad_u64_u32 + v_addc_co_u32 — two VALU instructions per term — and is reduced mod p once per run of rows, not per term.
//     Outputs beyond 16 are handled in sweeps of 16 (grid.y).  Any S, any alignment.
//   * direct_mfma_kernel (matrix cores, E >= 16): x and w are cut into signed base-256 digits (x or x - p, whichever fits the
//     balanced range), k' = (row, digit of x) is the contraction index and (output, digit of 256^a w mod p) the M index of
//     v_mfma_i32_32x32x32_i8, so the data words themselves are the B operand — no transposition, one VALU fix-up per word — and the
//     i32 sums of the four output digits land in four accumulator registers of one lane.  Weight fragments come from a table built
//     once per pattern (16 bytes per row and output), staged through LDS for the four waves of a workgroup.
//     16 digit products per term on ~4 POPS of i8 instead of 2 VALU instructions: up to ~100 outputs the pass costs one read of the
//     data.
//
// Weight tables (per erasure pattern / per code) are built on the device from closed forms of the Lagrange basis in O(E) products
// per row (prefix / suffix products through the output row; one inversion per row).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <new>
#include <vector>

#include "gf.hpp"
#include "internal.hpp"
#include "ntt_device.hpp"

namespace fastecc {

namespace {

constexpr int DIRECT_CAP = 256;          // outputs per pass: bound of the small parameter tables
constexpr uint32_t DIRECT_ROWS = 512;    // rows per partial sum (times the number of sweeps, up to 8)
constexpr uint32_t DIRECT_SEGS = 32;     // first summation step: the partial sums in this many segments
constexpr uint32_t MFMA_ROWS = 4096;     // most rows per partial sum of the MFMA kernel (i32 digit sums: < 2^19 per 8 rows, so < 2^28)
constexpr uint32_t TAIL_ROWS = 16;       // rows per partial sum for the few rows the MFMA kernel leaves to the VALU kernel
constexpr int MFMA_G = 4;                // 8-row steps per LDS stage of weight fragments

__device__ __forceinline__ uint32_t dev_pow(uint32_t x, uint32_t e)
{
    uint32_t r = 1;
    for (; e; e >>= 1) {
        if (e & 1u) r = gf::mul(r, x);
        x = gf::mul(x, x);
    }
    return r;
}

// Weight table layout: [rows][pad] for pad <= 16; beyond that [pad / 16][rows][16] — one 64-byte run per row and sweep of 16 outputs, so that
// the builders (one thread per row, walking the outputs) and the VALU kernel (one scalar load per row and sweep) both touch whole 64-byte
// runs.  ([rows][pad] with pad up to 256 made every access of the row-walking builders its own cache line: 15 GB of HBM traffic for a 0.5 GB table.)
__host__ __device__ __forceinline__ size_t coef_index(uint32_t row, uint32_t out, uint32_t rows, uint32_t pad)
{
    return pad <= 16u ? (size_t)row * pad + out : ((size_t)(out >> 4) * rows + row) * 16u + (out & 15u);
}

// ------------------------------------------------------------------------------------------------
// VALU form
// ------------------------------------------------------------------------------------------------
// (hi:lo) += x * w, w wave-uniform: the 64-bit multiply-add delivers its carry in an SGPR pair, the add-with-carry consumes it
__device__ __forceinline__ void mac96(uint64_t& lo, uint32_t& hi, uint32_t x, uint32_t w)
{
    uint64_t c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(lo), "=s"(c) : "v"(x), "s"(w));
    asm("v_addc_co_u32_e64 %0, %1, 0, %0, %1" : "+v"(hi), "+s"(c));
}
// (hi * 2^64 + lo) / 2^32 mod p: the sum of products with Montgomery-form weights, as a plain representative
__device__ __forceinline__ uint32_t reduce96(uint64_t lo, uint32_t hi)
{
    const uint32_t l0 = (uint32_t)lo, l1 = (uint32_t)(lo >> 32);
    uint32_t r = gf::mul_mont(l0, 1u);                            // l0 / 2^32
    r = gf::add(r, l1 >= gf::P ? l1 - gf::P : l1);                // l1
    r = gf::add(r, gf::mul(hi >= gf::P ? hi - gf::P : hi, gf::MONT_ONE));  // hi * 2^32
    return r;
}

struct AccArgs {
    const uint32_t* data;    // rows [0, data_rows)
    const uint32_t* parity;  // rows data_rows.. are its rows extra[u - data_rows] (may be null when rows == data_rows)
    const uint32_t* extra;
    const uint32_t* coef;    // [rows][pad], Montgomery form
    uint32_t* partial;       // [chunks][ppad][S]
    uint32_t S, rows, data_rows, pad, rows_per_chunk, col_chunks;
    uint64_t items;          // chunks * col_chunks
    uint32_t row_begin;      // the pass covers rows [row_begin, rows) (the MFMA kernel takes the bulk of the data stripe, this one the rest)
    uint32_t chunk_base;     // its partial sums go to chunk chunk_base + ...
    uint32_t ppad;           // outputs per chunk in `partial`
};

// partial[chunk][sweep * EB + j][col] = sum over the chunk's rows u of block_u[col] * coef[u][sweep * EB + j]
template <int EB, int V>
__global__ __launch_bounds__(256) void direct_accumulate_kernel(const AccArgs a)
{
    constexpr int U = EB >= 16 ? 4 : 8;  // rows in flight; U * EB weights live in SGPRs
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t item = (uint64_t)blockIdx.x * 4u + wave;
    if (item >= a.items) return;
    const uint32_t cc = (uint32_t)(item % a.col_chunks);
    const uint32_t chunk = (uint32_t)(item / a.col_chunks);
    const uint32_t sweep = blockIdx.y;
    const uint32_t col = (cc * 64u + lane) * V;
    const bool live = col < a.S;
    uint64_t lo[EB][V];
    uint32_t hi[EB][V];
#pragma unroll
    for (int j = 0; j < EB; ++j)
#pragma unroll
        for (int v = 0; v < V; ++v) lo[j][v] = 0, hi[j][v] = 0;
    const uint32_t u0 = a.row_begin + chunk * a.rows_per_chunk, u1 = min(u0 + a.rows_per_chunk, a.rows);
    // pad <= 16: one sweep, rows of `pad` weights; else sweep-major runs of 16 (coef_index)
    const uint32_t cstride = a.pad <= 16u ? a.pad : 16u;
    const_u32_ptr coef = as_constant(a.coef) + (size_t)sweep * a.rows * 16u;
    for (uint32_t ub = u0; ub < u1; ub += U) {
        uint32_t w[U][EB], x[U][V];
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const uint32_t u = ub + i;
            const bool in = u < u1;  // wave-uniform
#pragma unroll
            for (int j = 0; j < EB; ++j) w[i][j] = 0;
            if (in) {
                const_u32_ptr cf = coef + (size_t)u * cstride;
#pragma unroll
                for (int j = 0; j < EB; ++j) w[i][j] = cf[j];
            }
#pragma unroll
            for (int v = 0; v < V; ++v) x[i][v] = 0;
            if (in && live) {  // a lost row holds anything: its weights are zero
                const uint32_t* row = u < a.data_rows ? a.data + (size_t)u * a.S : a.parity + (size_t)as_constant(a.extra)[u - a.data_rows] * a.S;
                load_vec<V>(x[i], row + col);
            }
        }
#pragma unroll
        for (int i = 0; i < U; ++i)
#pragma unroll
            for (int j = 0; j < EB; ++j)
#pragma unroll
                for (int v = 0; v < V; ++v) mac96(lo[j][v], hi[j][v], x[i][v], w[i][j]);
    }
    if (live) {
#pragma unroll
        for (int j = 0; j < EB; ++j) {
            uint32_t r[V];
#pragma unroll
            for (int v = 0; v < V; ++v) r[v] = reduce96(lo[j][v], hi[j][v]);
            store_vec<V>(a.partial + ((size_t)(a.chunk_base + chunk) * a.ppad + sweep * EB + j) * a.S + col, r);
        }
    }
}

// Sum of the partial sums in two steps.  Step 1: segment `seg` of the chunks -> stage[seg][j][col]; step 2: the DIRECT_SEGS stage
// rows -> output j, written where pos[j] says: data row pos >> 1 (even) or parity row pos >> 1 (odd).
__global__ __launch_bounds__(256) void direct_reduce1_kernel(const uint32_t* __restrict__ partial, uint32_t* __restrict__ stage, uint32_t S, uint32_t chunks, int pad, int e)
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
__global__ __launch_bounds__(256) void direct_reduce2_kernel(const uint32_t* __restrict__ stage, const uint32_t* __restrict__ pos, uint32_t* __restrict__ data,
                                                             uint32_t* __restrict__ parity, uint32_t S, int pad, int e)
{
    const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (col >= S || j >= e) return;
    const uint32_t p = pos[j];
    uint32_t v = 0;
#pragma unroll
    for (uint32_t g = 0; g < DIRECT_SEGS; ++g) v = gf::add(v, stage[((size_t)g * pad + j) * S + col]);
    uint32_t* out = (p & 1u) ? parity : data;
    if (out) out[(size_t)(p >> 1) * S + col] = v;  // (no stripe given for that kind of output: fastecc_decode on a pass that also holds the lost parity blocks)
}

// ------------------------------------------------------------------------------------------------
// MFMA form
// ------------------------------------------------------------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// Signed base-256 digits, packed: x = sum_a s_a 256^a (mod p) with s_a in [-128, 127], s_a = byte a of the result as int8.
// y = x + 0x80808080 biases every digit by 128; where that would carry out of 32 bits (x >= 0x7F7F7F80) the value is taken as x - p
// instead (2^32 = 2^20 - 1 mod p: y + 0xFFFFF, which cannot carry again).  Bytes of y are the biased digits; flipping bit 7 un-biases.
__device__ __forceinline__ uint32_t balanced_digits(uint32_t x)
{
    const uint32_t bias = x >= 0x7F7F7F80u ? 0x8090807Fu : 0x80808080u;
    return (x + bias) ^ 0x80808080u;
}

// Weight fragments for v_mfma_i32_32x32x32_i8, A operand: step ks covers rows [8 ks, 8 ks + 8), M-tile mt the outputs [8 mt, 8 mt + 8).
// Lane l: m = l & 31 = 4 j' + b (output 8 mt + j', digit b of the weight), half = l >> 5; its 16 bytes, index 4 i + a: digit b of
// 256^a * w[row 8 ks + 4 half + i][output] — k' = (row, a) pairs with digit a of the data word of that row (B operand, same index).
// frag[(ks * MTtot + mt) * 64 + l] as uint4.  coef: [rows][pad] Montgomery form.
__global__ __launch_bounds__(256) void mfma_weights_kernel(const uint32_t* __restrict__ coef, uint4* __restrict__ frag, uint32_t rows, uint32_t table_rows, uint32_t pad,
                                                           uint32_t mt_total, uint64_t total)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const uint32_t l = (uint32_t)(t & 63u);
    const uint64_t tile = t >> 6;
    const uint32_t mt = (uint32_t)(tile % mt_total);
    const uint64_t ks = tile / mt_total;
    const uint32_t m = l & 31u, half = l >> 5, b = m & 3u, out = 8u * mt + (m >> 2);
    uint32_t dw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint64_t u = 8ull * ks + 4u * half + i;
        uint32_t w = (u < rows && out < pad) ? gf::mul_mont(coef[coef_index((uint32_t)u, out, table_rows, pad)], 1u) : 0u;  // plain representative
        uint32_t packed = 0;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const uint32_t d = (balanced_digits(w) >> (8 * b)) & 0xFFu;
            packed |= d << (8 * a);
            w = gf::mul(w, 256u);
        }
        dw[i] = packed;
    }
    frag[t] = make_uint4(dw[0], dw[1], dw[2], dw[3]);
}

struct MfmaArgs {
    const uint32_t* data;    // rows [0, rows): the bulk of the data stripe, rows a multiple of 8 * MFMA_G (the rest: the VALU kernel)
    const uint4* frag;       // [rows / 8 + MFMA_G][mt_total][64], the last MFMA_G steps zero
    uint32_t* partial;       // [chunks (+ the VALU kernel's)][pad][S]
    uint32_t S, rows, pad, mt_total, chunks, col_groups, sweeps;
    uint32_t chunk_rows;     // rows per chunk: a multiple of 8 * MFMA_G, at most MFMA_ROWS
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t direct_desc(const void* p, uint32_t bytes)
{
    // the pointer is wave-uniform; readfirstlane makes that provable (no waterfall loop around the buffer ops)
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

// A workgroup = 4 waves owns (chunk of MFMA_ROWS rows, 256 columns), a wave 64 of them as two 32-column N-tiles (lane n = l & 31 holds
// columns 2n, 2n+1: one dwordx2 per row; lanes 32-63 the rows 4 further down).  a sweep = 8*MT outputs.  At the top of a
// stage (G steps of 8 rows) everything the NEXT stage needs is requested — its weight fragments into registers (parked in the other half
// of the LDS buffer at the end of the stage) and its rows into a second register set — and lands while this stage's MFMAs run.
// Addresses: one buffer descriptor per chunk for the rows, one for the fragments; a lane-constant offset plus a scalar one per request.
// ABL (builds with -DFASTECC_DIRECT_ABLATION only; results are then WRONG on purpose): timing ablations that name the kernel's limiter —
// bit 0: no row loads inside the loop, bit 1: no LDS reads of A fragments inside the loop, bit 2: no fragment staging (loads, LDS writes,
// barrier) inside the loop, bit 3: no digit arithmetic.  profiles/r05/direct_mfma_ablation.jsonl.
// MG = 2 (round 6): TWO groups of four waves per workgroup share the rows' columns and the staged fragments — waves 0-3 take the first MT M-tiles
// of the sweep, waves 4-7 the next MT — so that 8 M-tiles (64 outputs) are one sweep at 128 accumulator registers per wave and TWO waves per SIMD:
// the second group's row requests are the first group's (same addresses, the same CU's L1), and each wave reads only its own group's fragments.
template <int MT, int ABL = 0, int NBX = 0, int MG = 1>
__global__ __launch_bounds__(256 * MG, (MG == 2 ? 2 : MT <= 2 ? 3 : MT <= 4 ? 2 : 1)) void direct_mfma_kernel(const MfmaArgs a)
{
    constexpr int G = MFMA_G, MTW = MT * MG, WN = G * MT / 4;  // M-tiles per workgroup; uint4 of weight fragments per thread and stage
    static_assert((G * MT) % 4 == 0, "stage size");
    __shared__ uint4 wl[2][G * MTW * 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = (tid >> 6) & 3u, mg = tid >> 8;
    // workgroup b runs on XCD b % 8: all workgroups of a chunk — the column groups share its weight fragments, the sweeps its rows —
    // go to the same XCD (one L2) and are dispatched next to each other
    const uint32_t b = blockIdx.x, xcd = b & 7u, idx = b >> 3;
    const uint32_t sweep = idx % a.sweeps, cg = (idx / a.sweeps) % a.col_groups;
    const uint32_t chunk = (idx / (a.sweeps * a.col_groups)) * 8u + xcd;
    if (chunk >= a.chunks) return;
    const uint32_t n = lane & 31u, half = lane >> 5;
    const uint32_t col = cg * 256u + wave * 64u + 2u * n;
    const bool live = col < a.S;
    const uint32_t row_bytes = a.S * 4u;
    const uint32_t voff = 4u * half * row_bytes + 4u * (live ? col : 0u);  // dead lanes read column 0: their D columns are never stored
    const uint32_t per_chunk = a.chunk_rows / 8u;
    const uint32_t ks0 = chunk * per_chunk, ks1 = min(ks0 + per_chunk, a.rows / 8u);
    const uint32_t stages = (ks1 - ks0) / G;
    const uint32_t step_bytes = a.mt_total * 1024u;  // fragments of one step, all M-tiles
    const __amdgpu_buffer_rsrc_t rows_desc = direct_desc(a.data + (size_t)ks0 * 8u * a.S, (ks1 - ks0) * 8u * row_bytes);
    const __amdgpu_buffer_rsrc_t frag_desc = direct_desc(a.frag + ((size_t)ks0 * a.mt_total + (size_t)sweep * MTW) * 64u, (ks1 - ks0 + G) * step_bytes);
    uint32_t wv[WN];  // where this thread's share of a stage's fragments sits, relative to the stage's first step
#pragma unroll
    for (int q = 0; q < WN; ++q) {
        const uint32_t e = q * (256u * MG) + tid;
        wv[q] = (e / (MTW * 64u)) * step_bytes + (e % (MTW * 64u)) * 16u;
    }
    v16i acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][c][r] = 0;

    // NB row buffers: stage s lives in buffer s % NB and, once its steps have been turned into B fragments, the buffer is refilled with
    // stage s + NB — that many stages of rows (NB * G * 4 KB per wave) are in flight, which is what hides the HBM latency when only one or
    // two waves fit a SIMD.  Stages past the end re-read the chunk's first stages (never used), so no bound has to be checked.
    constexpr int NB = NBX ? NBX : MT >= 4 ? 2 : 1;
    v2u x[NB][G][4];
    v4u wreg[WN];
#pragma unroll
    for (int q = 0; q < WN; ++q) wreg[q] = __builtin_amdgcn_raw_buffer_load_b128(frag_desc, wv[q], 0, 0);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const uint32_t st0 = (uint32_t)nb < stages ? nb : 0;
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) x[nb][g][i] = __builtin_amdgcn_raw_buffer_load_b64(rows_desc, voff, (8u * (st0 * G + g) + i) * row_bytes, 0);
    }
#pragma unroll
    for (int q = 0; q < WN; ++q) wl[0][q * (256u * MG) + tid] = make_uint4(wreg[q][0], wreg[q][1], wreg[q][2], wreg[q][3]);
    __syncthreads();
    for (uint32_t s0 = 0; s0 < stages; s0 += NB) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const uint32_t s = s0 + nb;
            if (s < stages) {  // workgroup-uniform
                const uint32_t sw = s + 1 < stages ? s + 1 : 0, sx = s + NB < stages ? s + NB : 0;
                if (!(ABL & 4)) {
#pragma unroll
                    for (int q = 0; q < WN; ++q) wreg[q] = __builtin_amdgcn_raw_buffer_load_b128(frag_desc, wv[q], sw * G * step_bytes, 0);
                }
                const uint4* wcur = wl[(ABL & 4) ? 0 : (s & 1u)];
                // A fragments: PF of them on their way from LDS ahead of the MFMAs that use them — with ONE wave per SIMD (MT = 8) nothing else
                // runs while this wave sits in an s_waitcnt, and one fragment ahead (two MFMAs = 64 cycles) is about the latency of a ds_read_b128
                constexpr int PF = 3, RING = 4;
                uint4 af[RING];
                auto frag_at = [&](int t) { return wcur[((t / MT) * MTW + mg * MT + (t % MT)) * 64 + lane]; };  // step t / MT, this group's M-tile t % MT
#pragma unroll
                for (int t = 0; t < PF && t < G * MT; ++t) af[t] = frag_at(t);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    v4i bf[2];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        bf[0][i] = (int)((ABL & 8) ? x[nb][g][i][0] : balanced_digits(x[nb][g][i][0]));
                        bf[1][i] = (int)((ABL & 8) ? x[nb][g][i][1] : balanced_digits(x[nb][g][i][1]));
                    }
                    if (!(ABL & 1)) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) x[nb][g][i] = __builtin_amdgcn_raw_buffer_load_b64(rows_desc, voff, (8u * (sx * G + g) + i) * row_bytes, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);  // the requests stay ahead of the arithmetic
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const int t = g * MT + mt;
                        if (t + PF < G * MT && !(ABL & 2)) af[(t + PF) % RING] = frag_at(t + PF);
                        v4i av;
                        av[0] = (int)af[t % RING].x; av[1] = (int)af[t % RING].y; av[2] = (int)af[t % RING].z; av[3] = (int)af[t % RING].w;
                        acc[mt][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bf[0], acc[mt][0], 0, 0, 0);
                        acc[mt][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bf[1], acc[mt][1], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!(ABL & 4)) {
#pragma unroll
                    for (int q = 0; q < WN; ++q) wl[(s + 1) & 1u][q * (256u * MG) + tid] = make_uint4(wreg[q][0], wreg[q][1], wreg[q][2], wreg[q][3]);
                    __syncthreads();
                }
            }
        }
    }
    // D: column n = lane & 31, row m = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) = 4 j' + b: registers 4 q .. 4 q + 3 of a lane are the
    // four digit sums of output j' = 2 q + half.  value = sum_b d_b 256^b, |d_b| < 2^28 here; + p * 2^24 makes it positive.
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t r[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int64_t v = (int64_t)acc[mt][c][4 * q] + ((int64_t)acc[mt][c][4 * q + 1] << 8) + ((int64_t)acc[mt][c][4 * q + 2] << 16) +
                                  ((int64_t)acc[mt][c][4 * q + 3] << 24);
                const uint64_t t = (uint64_t)(v + ((int64_t)gf::P << 24));
                const uint32_t tl = (uint32_t)t, th = (uint32_t)(t >> 32);  // th < 2^26
                r[c] = gf::add(gf::mul(th, gf::MONT_ONE), tl >= gf::P ? tl - gf::P : tl);
            }
            const uint32_t out = sweep * (8u * MTW) + 8u * (mg * MT + mt) + 2u * q + half;
            if (live) *reinterpret_cast<uint2*>(a.partial + ((size_t)chunk * a.pad + out) * a.S + col) = make_uint2(r[0], r[1]);
            __builtin_amdgcn_sched_barrier(0);  // one output at a time: 256 accumulators are not all read out before the first store
        }
}

// ------------------------------------------------------------------------------------------------
// Weight tables.  Data point of row i: x_i = wd^i (wd of order N).  params: field elements, plain, DIRECT_CAP per array.
// ------------------------------------------------------------------------------------------------
// Output t = f(y_t) from ALL data rows (the code's parity points, or lost parity blocks once the data is complete):
// L_i(y) = (y^N - 1) x_i / (N (y - x_i)), so coef[i][t] = c_t x_i / (y_t - x_i) with c_t = (y_t^N - 1) / N.
// The `outputs` inversions of a row are one: prefix products are parked in the output row, then unwound.
// params: [0, CAP) y_t, [CAP, 2 CAP) c_t
// A row's weights are written and read in runs of 16 — one 64-byte piece of the table per row and run (coef_index), as four 16-byte
// accesses per lane with the lanes' pieces back to back: a single word per lane and instruction touched a whole 64-byte segment for
// 4 bytes of it, 15 GB of traffic for a 0.5 GB table (profiles/r03).  Tables of fewer than 16 outputs per row are one short run.
__device__ __forceinline__ void coef_store_run(uint32_t* coef, uint32_t row, int t0, uint32_t rows, int pad, const uint32_t (&v)[16])
{
    uint32_t* p = coef + coef_index(row, (uint32_t)t0, rows, (uint32_t)pad);
    if (pad >= 16) {
        uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = make_uint4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (j < pad) p[j] = v[j];
    }
}
__device__ __forceinline__ void coef_load_run(const uint32_t* coef, uint32_t row, int t0, uint32_t rows, int pad, uint32_t (&v)[16])
{
    const uint32_t* p = coef + coef_index(row, (uint32_t)t0, rows, (uint32_t)pad);
    if (pad >= 16) {
        const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint4 w = q[j];
            v[4 * j] = w.x, v[4 * j + 1] = w.y, v[4 * j + 2] = w.z, v[4 * j + 3] = w.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = j < pad ? p[j] : 0u;
    }
}

__global__ __launch_bounds__(256) void lagrange_coef_kernel(uint32_t* __restrict__ coef, const uint32_t* __restrict__ params, uint32_t wd, uint32_t K, int outputs, int pad)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    const uint32_t xi = dev_pow(wd, i);
    uint32_t run = 1;
    for (int t0 = 0; t0 < pad; t0 += 16) {  // prefix products, parked in the output row
        uint32_t v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            v[j] = 0;
            if (t0 + j < outputs) {
                v[j] = run;
                run = gf::mul(run, gf::sub(params[t0 + j], xi));  // never zero: a parity point is not a data point
            }
        }
        coef_store_run(coef, i, t0, K, pad, v);
    }
    uint32_t inv = dev_pow(run, gf::P - 2u);
    for (int t0 = ((pad - 1) / 16) * 16; t0 >= 0; t0 -= 16) {
        uint32_t v[16];
        coef_load_run(coef, i, t0, K, pad, v);
#pragma unroll
        for (int j = 15; j >= 0; --j) {
            const int t = t0 + j;
            if (t < outputs) {
                const uint32_t d = gf::sub(params[t], xi);
                const uint32_t invd = gf::mul(inv, v[j]);  // 1 / (y_t - x_i)
                inv = gf::mul(inv, d);
                v[j] = gf::mul(gf::mul(gf::mul(params[DIRECT_CAP + t], xi), invd), gf::MONT_ONE);
            } else {
                v[j] = 0;
            }
        }
        coef_store_run(coef, i, t0, K, pad, v);
    }
}

// Interpolation on the N nodes {x_i : i not lost} + {y_a}: R the lost data rows (points x_r), A as many surviving parity points y_a,
// A(x) = prod_a (x - y_a), R(x) = prod_r (x - x_r), R_r = R / (x - x_r); node polynomial M = (x^N - 1) A / R:
//     weight of data row i in lost row r   = C_r x_i R_r(x_i) / A(x_i),   C_r = -A(x_r) / (x_r R_r(x_r))
//     weight of parity node a in lost row r = N A_a(x_r) R(y_a) / (x_r R_r(x_r) (y_a^N - 1) A_a(y_a)),  A_a = A / (x - y_a)
// and, for fastecc_repair in ONE pass, further targets y_t (the lost parity blocks) on the same nodes — L_u(y) = M(y) / ((y - u) M'(u)):
//     weight of data row i in target t      = c_t x_i R(x_i) / (A(x_i) (y_t - x_i)),   c_t = (y_t^N - 1) A(y_t) / (N R(y_t))
//     weight of parity node a in target t   = (y_t^N - 1) A_a(y_t) R(y_a) / (R(y_t) (y_a^N - 1) A_a(y_a))
// params: [0, CAP) the targets: x_r (ed of them), then y_t (ep); [CAP, 2 CAP) y_a; [2 CAP, 3 CAP) C_r (interp_params_kernel), then c_t (host)
__global__ __launch_bounds__(256) void interp_params_kernel(uint32_t* __restrict__ params, int ed)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= ed) return;
    const uint32_t xr = params[r];
    uint32_t A = 1, Rr = 1;
    for (int t = 0; t < ed; ++t) {
        A = gf::mul(A, gf::sub(xr, params[DIRECT_CAP + t]));
        if (t != r) Rr = gf::mul(Rr, gf::sub(xr, params[t]));
    }
    params[2 * DIRECT_CAP + r] = gf::sub(0u, gf::mul(A, dev_pow(gf::mul(xr, Rr), gf::P - 2u)));
}
__global__ __launch_bounds__(256) void interp_coef_kernel(uint32_t* __restrict__ coef, const uint32_t* __restrict__ params, uint32_t wd, uint32_t K, uint32_t rows, int ed,
                                                          int ep, int pad)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    const uint32_t xi = dev_pow(wd, i);
    // forward: the products over the targets before t — of (x_i - x_s) for the lost rows, of (y_t - x_i) for the further targets
    uint32_t A = 1, run = 1, far = 1;
    for (int t0 = 0; t0 < pad; t0 += 16) {
        uint32_t v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int t = t0 + j;
            v[j] = 0;
            if (t < ed) {
                A = gf::mul(A, gf::sub(xi, params[DIRECT_CAP + t]));
                v[j] = run;  // prod_{s < t} (x_i - x_s)
                run = gf::mul(run, gf::sub(xi, params[t]));
            } else if (t < ed + ep) {
                v[j] = far;  // prod_{ed <= s < t} (y_s - x_i)
                far = gf::mul(far, gf::sub(params[t], xi));
            }
        }
        coef_store_run(coef, i, t0, rows, pad, v);
    }
    const bool node = run != 0;  // x_i is one of the lost points: the row is not a node, its weights are zero
    // one inversion for both denominators: A(x_i) and prod_t (y_t - x_i) (neither is zero: parity points are no data points)
    const uint32_t inv = dev_pow(gf::mul(A, far), gf::P - 2u);
    const uint32_t base = node ? gf::mul(xi, gf::mul(inv, far)) : 0u;  // x_i / A(x_i)
    const uint32_t base_far = gf::mul(gf::mul(base, run), gf::mul(inv, A));  // x_i R(x_i) / (A(x_i) prod_t (y_t - x_i))
    uint32_t suf = 1, suf_far = 1;
    for (int t0 = ((pad - 1) / 16) * 16; t0 >= 0; t0 -= 16) {
        uint32_t v[16];
        coef_load_run(coef, i, t0, rows, pad, v);
#pragma unroll
        for (int j = 15; j >= 0; --j) {
            const int t = t0 + j;
            if (t < ed && node) {
                const uint32_t w = gf::mul(gf::mul(gf::mul(params[2 * DIRECT_CAP + t], base), v[j]), suf);
                suf = gf::mul(suf, gf::sub(xi, params[t]));
                v[j] = gf::mul(w, gf::MONT_ONE);
            } else if (t >= ed && t < ed + ep && node) {
                const uint32_t w = gf::mul(gf::mul(gf::mul(params[2 * DIRECT_CAP + t], base_far), v[j]), suf_far);
                suf_far = gf::mul(suf_far, gf::sub(params[t], xi));
                v[j] = gf::mul(w, gf::MONT_ONE);
            } else {
                v[j] = 0;
            }
        }
        coef_store_run(coef, i, t0, rows, pad, v);
    }
}
// thread (a, t): the weight of parity node a in target t -> coef[K + a][t]
__global__ __launch_bounds__(256) void interp_node_kernel(uint32_t* __restrict__ coef, const uint32_t* __restrict__ params, uint32_t K, uint32_t N, int ed, int ep, int pad)
{
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (uint32_t)ed * pad) return;
    const int a = idx / pad, t = idx % pad;
    uint32_t v = 0;
    if (t < ed + ep) {
        const uint32_t z = params[t], ya = params[DIRECT_CAP + a];  // the target: x_r or y_t
        uint32_t Aa_z = 1, Aa_ya = 1, R_ya = 1, R_z = 1;  // (R_z: without the target's own factor when it is a lost row)
        for (int s = 0; s < ed; ++s) {
            const uint32_t ys = params[DIRECT_CAP + s], xs = params[s];
            if (s != a) Aa_z = gf::mul(Aa_z, gf::sub(z, ys)), Aa_ya = gf::mul(Aa_ya, gf::sub(ya, ys));
            R_ya = gf::mul(R_ya, gf::sub(ya, xs));
            if (s != t) R_z = gf::mul(R_z, gf::sub(z, xs));
        }
        // lost row: (z^N - 1) / R(z) -> N / (z R_r(z));  further target: (z^N - 1) / R(z) as it stands
        const uint32_t num = gf::mul(gf::mul(t < ed ? N : gf::sub(dev_pow(z, N), 1u), Aa_z), R_ya);
        const uint32_t den = gf::mul(gf::mul(gf::mul(t < ed ? z : 1u, R_z), gf::sub(dev_pow(ya, N), 1u)), Aa_ya);
        v = gf::mul(gf::mul(num, dev_pow(den, gf::P - 2u)), gf::MONT_ONE);
    }
    coef[coef_index(K + (uint32_t)a, (uint32_t)t, K + (uint32_t)ed, (uint32_t)pad)] = v;
}

int hip_code(const char* what, hipError_t e)
{
    set_error_detail(what, e);
    return e == hipErrorOutOfMemory ? FASTECC_E_NOMEM : FASTECC_E_DEVICE;
}
#define DIR_TRY(expr)                                      \
    do {                                                   \
        hipError_t e_ = (expr);                            \
        if (e_ != hipSuccess) return hip_code(#expr, e_);  \
    } while (0)

// grow-only device buffer
template <class T> int ensure(T*& p, uint64_t& have, uint64_t need)
{
    if (have >= need && p) return FASTECC_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    have = 0;
    DIR_TRY(hipMalloc((void**)&p, need * sizeof(T)));
    have = need;
    return FASTECC_OK;
}

int pad_of(int outputs)
{
    int pad = 1;
    while (pad < outputs && pad < 16) pad <<= 1;
    return outputs <= 16 ? pad : (outputs + 15) / 16 * 16;
}

}  // namespace

struct DirectPass {
    uint32_t* coef = nullptr;     // [rows][pad], Montgomery form
    uint64_t coef_words = 0;
    uint32_t* params = nullptr;   // 3 * DIRECT_CAP field elements for the table kernels
    uint32_t* lists = nullptr;    // [0, CAP) parity rows used as nodes; [CAP, 2 CAP) output positions (row << 1 | parity)
    uint32_t* partial = nullptr;  // [chunks + DIRECT_SEGS][pad][S]
    uint64_t partial_words = 0;
    uint4* frag = nullptr;        // MFMA weight fragments
    uint64_t frag_count = 0;
    bool frag_valid = false;
    int mfma_pad = 0;             // outputs rounded up to the M-tiles of the MFMA kernel in use
    uint32_t rows = 0, data_rows = 0;
    int outputs = 0, pad = 0;
    bool built = false;
};

DirectPass* direct_pass_new() { return new (std::nothrow) DirectPass(); }
void direct_pass_free(DirectPass* p)
{
    if (!p) return;
    for (void* b : {(void*)p->coef, (void*)p->params, (void*)p->lists, (void*)p->partial, (void*)p->frag})
        if (b) (void)hipFree(b);
    delete p;
}
int direct_cap() { return DIRECT_CAP; }
int direct_pass_outputs(const DirectPass* p) { return p && p->built ? p->outputs : 0; }

static int pass_common(DirectPass* p, uint32_t rows, uint32_t data_rows, int outputs)
{
    if (outputs < 1 || outputs > DIRECT_CAP) return FASTECC_E_UNSUPPORTED;
    p->built = false;
    p->frag_valid = false;
    p->rows = rows;
    p->data_rows = data_rows;
    p->outputs = outputs;
    p->pad = pad_of(outputs);
    uint64_t fixed = p->params ? 3 * DIRECT_CAP : 0;
    int rc = ensure(p->params, fixed, 3 * DIRECT_CAP);
    if (rc != FASTECC_OK) return rc;
    fixed = p->lists ? 2 * DIRECT_CAP : 0;
    rc = ensure(p->lists, fixed, 2 * DIRECT_CAP);
    if (rc != FASTECC_OK) return rc;
    return ensure(p->coef, p->coef_words, (uint64_t)rows * p->pad);
}

int direct_build_lagrange(DirectPass* p, uint32_t wd, uint32_t K, const std::vector<uint32_t>& y, const std::vector<uint32_t>& c, const std::vector<uint32_t>& out_pos,
                          hipStream_t st)
{
    const int outputs = (int)y.size();
    if (c.size() != y.size() || out_pos.size() != y.size() || K < 1) return FASTECC_E_INVAL;
    const int rc = pass_common(p, K, K, outputs);
    if (rc != FASTECC_OK) return rc;
    DIR_TRY(hipMemcpyAsync(p->params, y.data(), outputs * 4, hipMemcpyHostToDevice, st));
    DIR_TRY(hipMemcpyAsync(p->params + DIRECT_CAP, c.data(), outputs * 4, hipMemcpyHostToDevice, st));
    DIR_TRY(hipMemcpyAsync(p->lists + DIRECT_CAP, out_pos.data(), outputs * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(lagrange_coef_kernel, dim3((K + 255) / 256), dim3(256), 0, st, p->coef, p->params, wd, K, outputs, p->pad);
    DIR_TRY(hipGetLastError());
    DIR_TRY(hipStreamSynchronize(st));  // the host vectors may go out of scope
    p->built = true;
    return FASTECC_OK;
}

int direct_build_interp(DirectPass* p, uint32_t wd, uint64_t N, uint32_t K, const std::vector<uint32_t>& lost_rows, const std::vector<uint32_t>& lost_points,
                        const std::vector<uint32_t>& node_rows, const std::vector<uint32_t>& node_points, hipStream_t st, const std::vector<uint32_t>* more_points,
                        const std::vector<uint32_t>* more_pos)
{
    const int ed = (int)lost_rows.size(), ep = more_points ? (int)more_points->size() : 0;
    if (lost_points.size() != lost_rows.size() || node_rows.size() != lost_rows.size() || node_points.size() != lost_rows.size() || K < 1 || ed < 1) return FASTECC_E_INVAL;
    if (ep != 0 && (!more_pos || more_pos->size() != more_points->size() || ed + ep > DIRECT_CAP)) return FASTECC_E_INVAL;
    const int rc = pass_common(p, K + (uint32_t)ed, K, ed + ep);
    if (rc != FASTECC_OK) return rc;
    std::vector<uint32_t> pos(ed + ep), targets(ed + ep), cfar(ep);
    for (int r = 0; r < ed; r++) pos[r] = 2u * lost_rows[r], targets[r] = lost_points[r];
    auto fsub = [](uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a + gf::P - b) % gf::P); };
    const uint32_t inv_n = gf::h_inv((uint32_t)(N % gf::P));
    for (int t = 0; t < ep; t++) {
        const uint32_t y = (*more_points)[t];
        uint32_t A = 1, R = 1;
        for (int s = 0; s < ed; s++) A = gf::h_mul(A, fsub(y, node_points[s])), R = gf::h_mul(R, fsub(y, lost_points[s]));
        if (A == 0 || R == 0) return FASTECC_E_INVAL;  // a target that is a node or a lost data point
        pos[ed + t] = (*more_pos)[t];
        targets[ed + t] = y;
        cfar[t] = gf::h_mul(gf::h_mul(gf::h_mul(fsub(gf::h_pow(y, N), 1u), A), inv_n), gf::h_inv(R));  // (y^N - 1) A(y) / (N R(y))
    }
    DIR_TRY(hipMemcpyAsync(p->params, targets.data(), (ed + ep) * 4, hipMemcpyHostToDevice, st));
    DIR_TRY(hipMemcpyAsync(p->params + DIRECT_CAP, node_points.data(), ed * 4, hipMemcpyHostToDevice, st));
    if (ep) DIR_TRY(hipMemcpyAsync(p->params + 2 * DIRECT_CAP + ed, cfar.data(), ep * 4, hipMemcpyHostToDevice, st));
    DIR_TRY(hipMemcpyAsync(p->lists, node_rows.data(), ed * 4, hipMemcpyHostToDevice, st));
    DIR_TRY(hipMemcpyAsync(p->lists + DIRECT_CAP, pos.data(), (ed + ep) * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(interp_params_kernel, dim3((ed + 255) / 256), dim3(256), 0, st, p->params, ed);
    hipLaunchKernelGGL(interp_coef_kernel, dim3((K + 255) / 256), dim3(256), 0, st, p->coef, p->params, wd, K, K + (uint32_t)ed, ed, ep, p->pad);
    hipLaunchKernelGGL(interp_node_kernel, dim3((unsigned)(((uint64_t)ed * p->pad + 255) / 256)), dim3(256), 0, st, p->coef, p->params, K, (uint32_t)(N % gf::P), ed, ep,
                       p->pad);
    DIR_TRY(hipGetLastError());
    DIR_TRY(hipStreamSynchronize(st));
    p->built = true;
    return FASTECC_OK;
}

// VALU kernel over rows [row_begin, p->rows): partial sums to chunks chunk_base.. of a [..][ppad][S] buffer; returns the chunks it used
static int launch_accumulate(DirectPass* p, const uint32_t* data, const uint32_t* parity, uint32_t S, uint32_t row_begin, uint32_t chunk_base, uint32_t ppad,
                             uint32_t rows_per_chunk, hipStream_t st)
{
    const int pad = p->pad, sweeps = pad > 16 ? pad / 16 : 1, eb = std::min(pad, 16);
    const uint32_t chunks = (p->rows - row_begin + rows_per_chunk - 1) / rows_per_chunk;
    const int vmax = eb <= 4 ? 4 : eb == 8 ? 2 : 1;
    int v = vmax;
    const uintptr_t align = (uintptr_t)data | (uintptr_t)parity | (uintptr_t)p->partial;
    while (v > 1 && ((S % v) != 0 || (align & (4u * v - 1u)) != 0)) v >>= 1;
    AccArgs a{data, parity, p->lists, p->coef, p->partial, S, p->rows, p->data_rows, (uint32_t)pad, rows_per_chunk, (S + 64u * v - 1u) / (64u * v), 0, row_begin, chunk_base, ppad};
    a.items = (uint64_t)chunks * a.col_chunks;
    const dim3 grid((unsigned)((a.items + 3) / 4), (unsigned)sweeps);
#define FASTECC_ACC(EB, V) hipLaunchKernelGGL((direct_accumulate_kernel<EB, V>), grid, dim3(256), 0, st, a)
    switch (eb * 8 + v) {
        case 1 * 8 + 4: FASTECC_ACC(1, 4); break;
        case 1 * 8 + 2: FASTECC_ACC(1, 2); break;
        case 1 * 8 + 1: FASTECC_ACC(1, 1); break;
        case 2 * 8 + 4: FASTECC_ACC(2, 4); break;
        case 2 * 8 + 2: FASTECC_ACC(2, 2); break;
        case 2 * 8 + 1: FASTECC_ACC(2, 1); break;
        case 4 * 8 + 4: FASTECC_ACC(4, 4); break;
        case 4 * 8 + 2: FASTECC_ACC(4, 2); break;
        case 4 * 8 + 1: FASTECC_ACC(4, 1); break;
        case 8 * 8 + 2: FASTECC_ACC(8, 2); break;
        case 8 * 8 + 1: FASTECC_ACC(8, 1); break;
        default: FASTECC_ACC(16, 1); break;
    }
#undef FASTECC_ACC
    DIR_TRY(hipGetLastError());
    return FASTECC_OK;
}

// kernel: 0 = choose, 1 = VALU kernel, 2 = MFMA kernel (when its alignment requirements hold)
// Experiment knobs of the matrix-core pass (FASTECC_DIRECT_MIN_WGS, FASTECC_DIRECT_MG, FASTECC_DIRECT_NB): read once; a set one announces itself on stderr,
// so that a stray exported variable cannot change production timings silently.
static int experiment_knob(const char* name, int fallback)
{
    const char* e = getenv(name);
    if (!e) return fallback;
    fprintf(stderr, "[fastecc direct] EXPERIMENT KNOB ACTIVE: %s=%s (default %d)\n", name, e, fallback);
    return atoi(e);
}

int direct_run(DirectPass* p, const uint32_t* data, const uint32_t* parity, uint32_t* data_out, uint32_t* parity_out, uint32_t S, int kernel, hipStream_t st)
{
    if (!p || !p->built) return FASTECC_E_INVAL;
    const uint32_t rows = p->rows;
    const uint32_t bulk = p->data_rows / (8u * MFMA_G) * (8u * MFMA_G);  // the MFMA kernel's share: whole stages of the data stripe
    const bool mfma_ok = (S % 2) == 0 && (((uintptr_t)data) & 7u) == 0 && S >= 32 && bulk > 0 && (uint64_t)S * 4u * MFMA_ROWS < (1ull << 32);  // a chunk's rows within one buffer descriptor
    const bool use_mfma = kernel == 2 ? mfma_ok : kernel == 1 ? false : (mfma_ok && p->outputs >= 5 && S >= 64 && bulk >= 4096);
    uint32_t chunks;
    int pad;
    if (use_mfma) {
        // M-tiles per workgroup (measured: profiles/r03/direct_bench.jsonl; round 4: 8 tiles at one wave per SIMD and 4 tiles at two waves per SIMD with
        // twice the sweeps take the same time — 0.70 / 1.23 / 2.33 ms for 64 / 128 / 256 outputs — so latency is not what holds the kernel at ~45 % of the MFMA rate)
#ifdef FASTECC_DIRECT_ABLATION
        static const int mt_env = [] { const char* e = getenv("FASTECC_DIRECT_MT"); return e ? atoi(e) : 0; }();
        const int mt = mt_env ? mt_env : p->outputs <= 16 ? 2 : p->outputs <= 32 ? 4 : 8;
#else
        const int mt = p->outputs <= 16 ? 2 : p->outputs <= 32 ? 4 : 8;
#endif
        // above 32 outputs: 8 M-tiles per wave, one wave per SIMD.  FASTECC_DIRECT_MG=2 (experiments): 8 M-tiles per workgroup as two groups of 4
        // (512 threads, two waves per SIMD, still ONE sweep over the rows) — measured SLOWER, 0.74 against 0.67 ms at 64 outputs, 1.33 against 1.23 at
        // 128 (profiles/r06/direct_mfma_two_groups.jsonl): two waves on a SIMD take matrix-pipe time from each other, as the butterfly stage found
        static const int mg_env = experiment_knob("FASTECC_DIRECT_MG", 1);
        const bool two_groups = mt == 8 && mg_env == 2;
        pad = (p->outputs + 8 * mt - 1) / (8 * mt) * (8 * mt);
        const uint32_t mt_total = (uint32_t)pad / 8u;
        const uint32_t steps_alloc = bulk / 8u + MFMA_G;  // zero steps at the end: the prefetch of a stage never leaves the table
        if (!p->frag_valid || p->mfma_pad != pad) {
            const uint64_t count = (uint64_t)steps_alloc * mt_total * 64u;
            const int rc = ensure(p->frag, p->frag_count, count);
            if (rc != FASTECC_OK) return rc;
            hipLaunchKernelGGL(mfma_weights_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, p->coef, p->frag, bulk, rows, (uint32_t)p->pad, mt_total, count);
            DIR_TRY(hipGetLastError());
            p->frag_valid = true;
            p->mfma_pad = pad;
        }
        // rows per chunk: at least 512 workgroups (two rounds per CU: one workgroup of MT = 8 fills a CU) when the stripe is large enough for that.
        // (1024 until round 6: half the prologues and epilogues — 256 accumulators read out per lane — are 4.5 % of the 64-output pass,
        //  0.713 -> 0.681 ms; 256 is no better.  FASTECC_DIRECT_MIN_WGS overrides, for experiments.)
        const uint32_t col_groups = (S + 255u) / 256u, sweeps = (uint32_t)(pad / (8 * mt));
        uint32_t chunk_rows = 8u * MFMA_G * 8u;
        static const unsigned min_wgs = (unsigned)experiment_knob("FASTECC_DIRECT_MIN_WGS", 512);
        while (chunk_rows < MFMA_ROWS && (uint64_t)(bulk / (2u * chunk_rows)) * col_groups * sweeps >= min_wgs) chunk_rows *= 2u;
        const uint32_t mchunks = (bulk + chunk_rows - 1) / chunk_rows;
        const uint32_t tail_chunks = rows > bulk ? (rows - bulk + TAIL_ROWS - 1) / TAIL_ROWS : 0;
        chunks = mchunks + tail_chunks;
        int rc = ensure(p->partial, p->partial_words, ((uint64_t)chunks + DIRECT_SEGS) * pad * S);
        if (rc != FASTECC_OK) return rc;
        MfmaArgs a{data, p->frag, p->partial, S, bulk, (uint32_t)pad, mt_total, mchunks, col_groups, sweeps, chunk_rows};
        const dim3 grid((mchunks + 7u) / 8u * 8u * a.col_groups * a.sweeps);
        switch (mt) {
            case 2: hipLaunchKernelGGL(direct_mfma_kernel<2>, grid, dim3(256), 0, st, a); break;
            case 4: hipLaunchKernelGGL(direct_mfma_kernel<4>, grid, dim3(256), 0, st, a); break;
            default:
#ifdef FASTECC_DIRECT_ABLATION
            {
                static const int abl = [] { const char* e = getenv("FASTECC_DIRECT_ABLATE"); return e ? atoi(e) : 0; }();
                switch (abl) {
                    case 1: hipLaunchKernelGGL((direct_mfma_kernel<8, 1>), grid, dim3(256), 0, st, a); break;
                    case 2: hipLaunchKernelGGL((direct_mfma_kernel<8, 2>), grid, dim3(256), 0, st, a); break;
                    case 3: hipLaunchKernelGGL((direct_mfma_kernel<8, 3>), grid, dim3(256), 0, st, a); break;
                    case 4: hipLaunchKernelGGL((direct_mfma_kernel<8, 4>), grid, dim3(256), 0, st, a); break;
                    case 5: hipLaunchKernelGGL((direct_mfma_kernel<8, 5>), grid, dim3(256), 0, st, a); break;
                    case 6: hipLaunchKernelGGL((direct_mfma_kernel<8, 6>), grid, dim3(256), 0, st, a); break;
                    case 7: hipLaunchKernelGGL((direct_mfma_kernel<8, 7>), grid, dim3(256), 0, st, a); break;
                    case 8: hipLaunchKernelGGL((direct_mfma_kernel<8, 8>), grid, dim3(256), 0, st, a); break;
                    case 15: hipLaunchKernelGGL((direct_mfma_kernel<8, 15>), grid, dim3(256), 0, st, a); break;
                    case 103: hipLaunchKernelGGL((direct_mfma_kernel<8, 0, 3>), grid, dim3(256), 0, st, a); break;
                    case 104: hipLaunchKernelGGL((direct_mfma_kernel<8, 0, 4>), grid, dim3(256), 0, st, a); break;
                    case 105: hipLaunchKernelGGL((direct_mfma_kernel<8, 0, 5>), grid, dim3(256), 0, st, a); break;
                    default: hipLaunchKernelGGL((direct_mfma_kernel<8, 0>), grid, dim3(256), 0, st, a); break;
                }
                break;
            }
#else
                if (two_groups) {
                    static const int nb_env = experiment_knob("FASTECC_DIRECT_NB", 1);  // (row buffers in flight)
                    if (nb_env == 2) hipLaunchKernelGGL((direct_mfma_kernel<4, 0, 2, 2>), grid, dim3(512), 0, st, a);
                    else hipLaunchKernelGGL((direct_mfma_kernel<4, 0, 1, 2>), grid, dim3(512), 0, st, a);
                }
                else hipLaunchKernelGGL(direct_mfma_kernel<8>, grid, dim3(256), 0, st, a);
                break;
#endif
        }
        DIR_TRY(hipGetLastError());
        if (tail_chunks) {  // the last data rows and the parity rows used as nodes
            rc = launch_accumulate(p, data, parity, S, bulk, mchunks, (uint32_t)pad, TAIL_ROWS, st);
            if (rc != FASTECC_OK) return rc;
        }
    } else {
        pad = p->pad;
        const int sweeps = pad > 16 ? pad / 16 : 1;
        const uint32_t rows_per_chunk = DIRECT_ROWS * (uint32_t)std::min(sweeps, 8);
        chunks = (rows + rows_per_chunk - 1) / rows_per_chunk;
        int rc = ensure(p->partial, p->partial_words, ((uint64_t)chunks + DIRECT_SEGS) * pad * S);
        if (rc != FASTECC_OK) return rc;
        rc = launch_accumulate(p, data, parity, S, 0, 0, (uint32_t)pad, rows_per_chunk, st);
        if (rc != FASTECC_OK) return rc;
    }
    uint32_t* stage = p->partial + (size_t)chunks * pad * S;
    hipLaunchKernelGGL(direct_reduce1_kernel, dim3((S + 255) / 256, (unsigned)p->outputs, DIRECT_SEGS), dim3(256), 0, st, p->partial, stage, S, chunks, pad, p->outputs);
    hipLaunchKernelGGL(direct_reduce2_kernel, dim3((S + 255) / 256, (unsigned)p->outputs), dim3(256), 0, st, stage, p->lists + DIRECT_CAP, data_out, parity_out, S, pad,
                       p->outputs);
    DIR_TRY(hipGetLastError());
    return FASTECC_OK;
}

// ------------------------------------------------------------------------------------------------
// ENCODING when a code has few parity blocks: with the data points x_i = w_N^i the parity block j = f(y_j) = sum_i data_i * L_i(y_j),
// y_j = w_2N^(odd): y_j^N = -1, so coef[i][j] = -2 x_i / (N (y_j - x_i)) — one read of the data instead of the three trips of the
// transform pipeline.  Exactly the polynomial evaluation the transform computes (RS.cpp:40-63), hence the same parity bits.
// ------------------------------------------------------------------------------------------------
struct DirectEncode {
    DirectPass* pass = nullptr;
    uint32_t S = 0;
};

void direct_encode_destroy(DirectEncode* de)
{
    if (!de) return;
    direct_pass_free(de->pass);
    delete de;
}

int direct_encode_max() { return DIRECT_CAP; }

// N data points (a power of two, or q 2^m for the mixed-radix codes), K <= N existing data blocks, m parity blocks at the odd positions
// ((j << fold) << 1) + 1 of the 2N-th roots of unity (fastecc_create's layout).  The current device is the context's.
int direct_encode_build(DirectEncode** out, uint64_t N, uint64_t K, uint64_t m, int fold, uint64_t words)
{
    *out = nullptr;
    if (m < 1 || m > (uint64_t)DIRECT_CAP || K < 1 || K > N || N < 2 || K > 0xFFFFFFF0ull || ((gf::P - 1ull) % (2 * N)) != 0) return FASTECC_E_UNSUPPORTED;
    DirectEncode* de = new (std::nothrow) DirectEncode();
    if (!de) return FASTECC_E_NOMEM;
    de->S = (uint32_t)words;
    de->pass = direct_pass_new();
    if (!de->pass) {
        delete de;
        return FASTECC_E_NOMEM;
    }
    const uint32_t w2n = gf::h_root((uint32_t)(2 * N));
    const uint32_t c0 = gf::h_mul(gf::P - 2u, gf::h_inv((uint32_t)(N % gf::P)));  // -2 / N
    std::vector<uint32_t> y(m), cc(m, c0), pos(m);
    for (uint64_t j = 0; j < m; j++) {
        y[j] = gf::h_pow(w2n, ((j << fold) << 1) + 1);
        pos[j] = (uint32_t)(2 * j + 1);
    }
    const int rc = direct_build_lagrange(de->pass, gf::h_mul(w2n, w2n), (uint32_t)K, y, cc, pos, nullptr);
    if (rc != FASTECC_OK) {
        direct_encode_destroy(de);
        return rc;
    }
    *out = de;
    return FASTECC_OK;
}

// parity[j] = sum_i data[i] * coef[i][j]; data: K rows of S words, parity: m rows (may be the first m rows of data)
int direct_encode_run(DirectEncode* de, const uint32_t* data, uint32_t* parity, int kernel, hipStream_t st)
{
    return direct_run(de->pass, data, nullptr, nullptr, parity, de->S, kernel, st);
}
// whether the MFMA kernel can take these stripes (else many outputs cost 2 VALU instructions per word and output)
bool direct_mfma_applies(const void* data, const void* parity, uint64_t words)
{
    return (words % 2) == 0 && words >= 64 && ((((uintptr_t)data | (uintptr_t)parity) & 7u) == 0);
}

}  // namespace fastecc
