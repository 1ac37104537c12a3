// microbench_mfma_dft.hip — VERDICT r05 item 1 (a): is a radix-32 butterfly STAGE on the matrix cores faster than five radix-2 levels on the VALU?
// Isolated (no HBM traffic), bit-exact, on the layout the tile kernels already use.  Not part of the library.
//
// What a stage is.  A wave holds 32 blocks x 32 word columns in the pair layout of csrc/tile_kernels.hip: lane = (half h, column c), 16 VGPRs =
// blocks 16 h + r.  Five DIF levels on these 32 blocks followed by the per-block factors of the next stage are, per column,
//     y[o] = tw[o] * sum_i W32^(i o) x[i]   (mod p),   p = 0xFFF00001 (ntt.cpp:251-284 runs the same as 5 x 16 butterflies per column).
// The sum is a 32 x 32 matrix times the 32 x 32 tile of the wave: a contraction.  It is done EXACTLY on v_mfma_i32_32x32x32_i8 (the trick of
// csrc/direct.hip, here with the data VGPRs themselves as the B operand):
//   * x[i] = sum_j b_ij 256^j: the four bytes of a data VGPR are four K-slots of B (after one v_xor 0x80808080: b - 128 is a signed byte);
//   * A holds, for K-slot (i, j), the four BALANCED base-256 digits a_0..a_3 in [-128,127] of the representative of W32^(i o) 256^j mod p in
//     [-128 S, 127 S], S = (256^4-1)/255 (2^32 consecutive integers, p < 2^32 of them needed): four digit planes x four K-chunks = 16 MFMAs;
//   * plane sums T_d = sum_k a_d,k b_k are in [-4177920, 4145280]; L = T_0 + 256 T_1 and H = T_2 + 256 T_3 (one v_lshl_add_u32 each), both
//     shifted into [0, 2^31) by constants in the accumulators' initial values which also carry the +128 sum(a) correction of the xor:
//     cL + 2^16 cH = 0 (mod p), so the shifts cancel;
//   * y = (L tw + H tw 2^16) mod p is ONE Montgomery reduction of the 64-bit sum of two products (L, H < 2^31: the sum is below 2^32 p):
//     2 v_mad_u64_u32 + v_lshl_add_u32 + v_mul_hi_u32 + v_sub_co + v_add + v_cndmask.  The result is canonical.
// VALU per value and stage: 1 + 2 + 7 = 10 (+ what it costs to hold tw per half-wave) against 5 x 6 = 30.
//
// Probes (one JSON line each): "valu5" five radix-2 levels on 16 registers (gf.hpp arithmetic; the baseline), "mfma" the stage above with
//   twiddle source TW: 0 = the same SGPR constants for both half-waves (lower bound, not usable), 1 = SGPR constants selected per half-wave,
//                      2 = per-lane global loads (two distinct 16-byte pieces per instruction), 3 = per-lane LDS loads;
//   A source: 0 = the 16 fragments stay in registers, 1 = ds_read_b128 per use.
// Usage: microbench_mfma_dft [waves_per_simd ...]      (default 1 2 3 4); verifies every variant against host arithmetic first.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "gf.hpp"

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

// y = (L w1 + H w2) / 2^32 mod p, L and H below 2^31, w1 and w2 below p: one reduction for both products.
__device__ __forceinline__ uint32_t mont_sum2(uint32_t L, uint32_t w1, uint32_t H, uint32_t w2)
{
    const uint64_t t = (uint64_t)L * w1 + (uint64_t)H * w2;
    const uint32_t lo = (uint32_t)t, hi = (uint32_t)(t >> 32);
    const uint32_t m = lo + (lo << 20);
    const uint32_t q = __umulhi(m, gf::P);
    uint32_t r;
    const bool borrow = __builtin_usub_overflow(hi, q, &r);
    return borrow ? r + gf::P : r;
}

struct StageArgs {
    const v4i* a_frag;     // [4 planes][4 chunks][64 lanes] fragments of the digit planes
    const v4i* init;       // [2 (L, H)][4][64 lanes]: 16 initial accumulator values per lane as four v4i
    const uint32_t* tw;    // [8 sets][2 halves][16 registers][2 (w1, w2)] Montgomery constants
    uint32_t* io;          // [waves][16][64]: in / out for the verification run
    int iters;
    int verify;
};

template <int TW, int ALDS>
__global__ __launch_bounds__(256) void mfma_stage_kernel(StageArgs g)
{
    __shared__ v4i s_a[ALDS ? 16 * 64 : 1];
    __shared__ uint32_t s_tw[TW == 3 ? 8 * 64 : 1];
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int half = lane >> 5;
    if (ALDS)
        for (int i = threadIdx.x; i < 16 * 64; i += 256) s_a[i] = g.a_frag[i];
    if (TW == 3)
        for (int i = threadIdx.x; i < 8 * 64; i += 256) s_tw[i] = g.tw[i];
    __syncthreads();
    v4i a[16];
    if (!ALDS)
#pragma unroll
        for (int f = 0; f < 16; ++f) a[f] = g.a_frag[f * 64 + lane];
    v16i initL, initH;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const v4i l4 = g.init[q * 64 + lane], h4 = g.init[(4 + q) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            initL[4 * q + e] = l4[e];
            initH[4 * q + e] = h4[e];
        }
    }
    uint32_t x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
        x[r] = g.verify ? g.io[((size_t)wave * 16 + r) * 64 + lane] : (uint32_t)((lane * 2654435761u + r * 40503u + wave) % gf::P);

    for (int it = 0; it < g.iters; ++it) {
        const int set = it & 7;
        if (ALDS) asm volatile("" ::: "memory");  // the fragments are read from LDS again in every stage, as a tile kernel would
        v4i b[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) b[m][e] = (int)(x[4 * m + e] ^ 0x80808080u);
        v16i acc[4];
        acc[0] = initL;
        acc[2] = initH;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[1][e] = 0, acc[3][e] = 0;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const v4i af = ALDS ? s_a[(d * 4 + m) * 64 + lane] : a[d * 4 + m];
                acc[d] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, b[m], acc[d], 0, 0, 0);
            }
        // twiddles for this "tile"
        uint32_t w1[16], w2[16];
        if constexpr (TW == 0 || TW == 1) {
            const uint32_t* __restrict__ t = g.tw + set * 64;  // uniform address: scalar loads
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if constexpr (TW == 0) {
                    w1[r] = t[2 * r];
                    w2[r] = t[2 * r + 1];
                } else {
                    w1[r] = half ? t[32 + 2 * r] : t[2 * r];
                    w2[r] = half ? t[32 + 2 * r + 1] : t[2 * r + 1];
                }
            }
        } else {
            const v4u* t4 = TW == 2 ? (const v4u*)(g.tw + set * 64 + half * 32) : (const v4u*)(s_tw + set * 64 + half * 32);
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const v4u q = t4[r >> 1];
                w1[r] = q[0], w2[r] = q[1], w1[r + 1] = q[2], w2[r + 1] = q[3];
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t L = (uint32_t)acc[0][r] + ((uint32_t)acc[1][r] << 8);
            const uint32_t H = (uint32_t)acc[2][r] + ((uint32_t)acc[3][r] << 8);
            x[r] = mont_sum2(L, w1[r], H, w2[r]);
        }
    }
    if (g.verify) {
#pragma unroll
        for (int r = 0; r < 16; ++r) g.io[((size_t)wave * 16 + r) * 64 + lane] = x[r];
    } else {
        uint32_t s = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) s ^= x[r];
        if (s == 0x12345679u) g.io[lane] = s;
    }
}

// The same stage as a two-stream pipeline inside ONE wave: the 16 MFMAs of wave-tile B are issued right after wave-tile A's four plane sums have
// been folded into L and H (32 VALU), so they run on the matrix pipe while the VALU does A's Montgomery sums (7 per word) — and the other way round.
// Twiddles: per-lane global loads, two words per value.  A source as above.
__device__ __forceinline__ void issue_mfmas(v16i (&acc)[4], const uint32_t (&x)[16], const v4i* a_reg, const v4i* a_lds, int lane, bool alds,
                                            const v16i& initL, const v16i& initH)
{
    v4i b[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) b[m][e] = (int)(x[4 * m + e] ^ 0x80808080u);
    acc[0] = initL;
    acc[2] = initH;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[1][e] = 0, acc[3][e] = 0;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const v4i af = alds ? a_lds[(d * 4 + m) * 64 + lane] : a_reg[d * 4 + m];
            acc[d] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, b[m], acc[d], 0, 0, 0);
        }
}

template <int ALDS>
__global__ __launch_bounds__(256) void mfma_pipe_kernel(StageArgs g)
{
    __shared__ v4i s_a[ALDS ? 16 * 64 : 1];
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int half = lane >> 5;
    if (ALDS)
        for (int i = threadIdx.x; i < 16 * 64; i += 256) s_a[i] = g.a_frag[i];
    __syncthreads();
    v4i a[ALDS ? 1 : 16];
    if (!ALDS)
#pragma unroll
        for (int f = 0; f < 16; ++f) a[f] = g.a_frag[f * 64 + lane];
    v16i initL, initH;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const v4i l4 = g.init[q * 64 + lane], h4 = g.init[(4 + q) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            initL[4 * q + e] = l4[e];
            initH[4 * q + e] = h4[e];
        }
    }
    uint32_t xa[16], xb[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        xa[r] = g.verify ? g.io[((size_t)wave * 16 + r) * 64 + lane] : (uint32_t)((lane * 2654435761u + r * 40503u + wave) % gf::P);
        xb[r] = (uint32_t)((lane * 40503u + r * 2654435761u + wave + 7u) % gf::P);
    }
    v16i acc[4];
    uint32_t L[16], H[16];
    auto fold = [&]() {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            L[r] = (uint32_t)acc[0][r] + ((uint32_t)acc[1][r] << 8);
            H[r] = (uint32_t)acc[2][r] + ((uint32_t)acc[3][r] << 8);
        }
    };
    auto finish = [&](uint32_t (&x)[16], int set) {
        const v4u* t4 = (const v4u*)(g.tw + set * 64 + half * 32);
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const v4u q = t4[r >> 1];
            x[r] = mont_sum2(L[r], q[0], H[r], q[1]);
            x[r + 1] = mont_sum2(L[r + 1], q[2], H[r + 1], q[3]);
        }
    };
    issue_mfmas(acc, xa, a, s_a, lane, ALDS, initL, initH);
    for (int it = 0; it < g.iters; ++it) {
        if (ALDS) asm volatile("" ::: "memory");
        fold();                                                   // A's plane sums -> L, H: the accumulators are free again
        issue_mfmas(acc, xb, a, s_a, lane, ALDS, initL, initH);   // B on the matrix pipe ...
        finish(xa, it & 7);                                       // ... while the VALU finishes A
        fold();
        if (it + 1 < g.iters || !g.verify) issue_mfmas(acc, xa, a, s_a, lane, ALDS, initL, initH);
        finish(xb, (it + 3) & 7);
    }
    if (g.verify) {
#pragma unroll
        for (int r = 0; r < 16; ++r) g.io[((size_t)wave * 16 + r) * 64 + lane] = xa[r];
    } else {
        uint32_t s = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) s ^= xa[r] ^ xb[r];
        if (s == 0x12345679u) g.io[lane] = s;
    }
}

// The arithmetic floor of the pipelined form: no memory traffic at all in the loop (fragments in registers, uniform twiddles in SGPRs), the
// instruction order pinned with sched_group_barrier: one MFMA, then PER VALU instructions of the OTHER stream's Montgomery sums, sixteen times.
template <int PER>
__global__ __launch_bounds__(256) void mfma_pipe_floor_kernel(StageArgs g)
{
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    v4i a[16];
#pragma unroll
    for (int f = 0; f < 16; ++f) a[f] = g.a_frag[f * 64 + lane];
    v16i initL, initH;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const v4i l4 = g.init[q * 64 + lane], h4 = g.init[(4 + q) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            initL[4 * q + e] = l4[e];
            initH[4 * q + e] = h4[e];
        }
    }
    uint32_t xa[16], xb[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        xa[r] = g.verify ? g.io[((size_t)wave * 16 + r) * 64 + lane] : (uint32_t)((lane * 2654435761u + r * 40503u + wave) % gf::P);
        xb[r] = (uint32_t)((lane * 40503u + r * 2654435761u + wave + 7u) % gf::P);
    }
    v16i acc[4];
    uint32_t L[16], H[16];
    const uint32_t* __restrict__ tws = g.tw;
    auto half_iteration = [&](uint32_t (&xin)[16], uint32_t (&xout)[16], int set) {
        // fold the finished plane sums of `xout`'s stream, start `xin`'s MFMAs, finish `xout`
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            L[r] = (uint32_t)acc[0][r] + ((uint32_t)acc[1][r] << 8);
            H[r] = (uint32_t)acc[2][r] + ((uint32_t)acc[3][r] << 8);
        }
        v4i b[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) b[m][e] = (int)(xin[4 * m + e] ^ 0x80808080u);
        __builtin_amdgcn_sched_barrier(0);  // the accumulators are free from here on: nothing below reads the old plane sums
        acc[0] = initL;
        acc[2] = initH;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[1][e] = 0, acc[3][e] = 0;
        const uint32_t* __restrict__ t = tws + set * 64;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                acc[d] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[d * 4 + m], b[m], acc[d], 0, 0, 0);
                const int r = m * 4 + d;
                xout[r] = mont_sum2(L[r], t[2 * r], H[r], t[2 * r + 1]);
            }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, PER, 0);
        }
    };
    // prologue: stream A's MFMAs
    {
        v4i b[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) b[m][e] = (int)(xa[4 * m + e] ^ 0x80808080u);
        acc[0] = initL;
        acc[2] = initH;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[1][e] = 0, acc[3][e] = 0;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int d = 0; d < 4; ++d) acc[d] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[d * 4 + m], b[m], acc[d], 0, 0, 0);
    }
    for (int it = 0; it < g.iters; ++it) {
        half_iteration(xb, xa, it & 7);        // B's MFMAs beside A's Montgomery sums
        half_iteration(xa, xb, (it + 3) & 7);  // A's next MFMAs beside B's
    }
    uint32_t s = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) s ^= xa[r] ^ xb[r] ^ (uint32_t)acc[0][r];
    if (g.verify) {
#pragma unroll
        for (int r = 0; r < 16; ++r) g.io[((size_t)wave * 16 + r) * 64 + lane] = xa[r];
    } else if (s == 0x12345679u) g.io[lane] = s;
}

// Baseline: five radix-2 DIF levels on the 16 registers of a lane (the fifth level pairs registers again instead of half-waves: the
// arithmetic is the same, the v_permlane32_swap of the real tile is left out — in favour of the baseline).
__global__ __launch_bounds__(256) void valu5_kernel(uint32_t* out, const uint32_t* __restrict__ tw, int iters)
{
    const int lane = threadIdx.x & 63;
    uint32_t x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = (uint32_t)((lane * 2654435761u + r * 40503u + blockIdx.x) % gf::P);
    for (int it = 0; it < iters; ++it) {
        const uint32_t* __restrict__ t = tw + (it & 7) * 64;
#pragma unroll
        for (int lv = 0; lv < 5; ++lv) {
            const int d = 1 << (lv & 3);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (!(r & d)) {
                    const uint32_t u = x[r], v = x[r + d];
                    x[r] = gf::add(u, v);
                    x[r + d] = gf::mul_mont(gf::sub(u, v), t[(lv * 8 + (r & 7)) & 63]);
                }
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) s ^= x[r];
    if (s == 0x12345679u) out[lane] = s;
}

// ---- host ----
static uint64_t rng_state = 0x1234;
static uint64_t splitmix()
{
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct Tables {
    std::vector<int8_t> a;        // [plane][chunk][lane][16]
    std::vector<uint32_t> init;   // [2][4][64][4]
    std::vector<uint32_t> tw;     // [8][2][16][2] Montgomery constants
    std::vector<uint32_t> twraw;  // [8][2][16] the factors themselves
    uint32_t W[32][32];
};

static inline int row_of(int half, int r) { return 8 * (r >> 2) + 4 * half + (r & 3); }  // D layout of the 32x32 MFMAs: VGPR r of a lane

static void build(Tables& T)
{
    const uint32_t w32 = gf::h_root(32);
    for (int o = 0; o < 32; ++o)
        for (int i = 0; i < 32; ++i) T.W[o][i] = gf::h_pow(w32, (uint64_t)((o * i) & 31));
    T.a.assign(4 * 4 * 64 * 16, 0);
    int64_t sumA[4][32] = {};
    for (int m = 0; m < 4; ++m)
        for (int lane = 0; lane < 64; ++lane)
            for (int t = 0; t < 16; ++t) {
                const int rho = lane & 31, hb = lane >> 5, r = t >> 2, j = t & 3, in = 16 * hb + 4 * m + r;
                uint32_t val = T.W[rho][in];
                for (int s = 0; s < j; ++s) val = gf::h_mul(val, 256);
                int64_t bal = val <= 2139062143u ? (int64_t)val : (int64_t)val - (int64_t)gf::P;
                for (int d = 0; d < 4; ++d) {
                    int64_t dig = ((bal % 256) + 256) % 256;
                    if (dig >= 128) dig -= 256;
                    bal = (bal - dig) / 256;
                    T.a[(((size_t)d * 4 + m) * 64 + lane) * 16 + t] = (int8_t)dig;
                    sumA[d][rho] += dig;
                }
                if (bal != 0) {
                    fprintf(stderr, "balanced digits do not close\n");
                    exit(1);
                }
            }
    // shifts: cL + 2^16 cH = 0 (mod p), both in [1073725440, 1082146688)
    uint32_t cL = 0, cH = 0;
    for (uint32_t h = 1073725440u; h < 1082146688u; ++h) {
        const uint32_t l = (uint32_t)((gf::P - (uint32_t)(((uint64_t)h << 16) % gf::P)) % gf::P);
        if (l >= 1073725440u && l < 1082146688u) {
            cL = l, cH = h;
            break;
        }
    }
    if (!cH) {
        fprintf(stderr, "no shift pair\n");
        exit(1);
    }
    T.init.assign(2 * 4 * 64 * 4, 0);
    for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 16; ++r) {
            const int rho = row_of(lane >> 5, r);
            const uint32_t iL = (uint32_t)(128 * (sumA[0][rho] + 256 * sumA[1][rho])) + cL;
            const uint32_t iH = (uint32_t)(128 * (sumA[2][rho] + 256 * sumA[3][rho])) + cH;
            T.init[(((size_t)0 * 4 + (r >> 2)) * 64 + lane) * 4 + (r & 3)] = iL;
            T.init[(((size_t)1 * 4 + (r >> 2)) * 64 + lane) * 4 + (r & 3)] = iH;
        }
    T.tw.resize(8 * 64);
    T.twraw.resize(8 * 32);
    for (int s = 0; s < 8; ++s)
        for (int hb = 0; hb < 2; ++hb)
            for (int r = 0; r < 16; ++r) {
                const uint32_t f = (uint32_t)(splitmix() % gf::P);
                T.twraw[(s * 2 + hb) * 16 + r] = f;
                T.tw[s * 64 + hb * 32 + 2 * r] = gf::h_to_mont(f);
                T.tw[s * 64 + hb * 32 + 2 * r + 1] = gf::h_to_mont(gf::h_mul(f, 65536));
            }
}

template <int TW, int ALDS>
static bool verify(const Tables& T, StageArgs g, uint32_t* d_io)
{
    const int waves = 8;
    std::vector<uint32_t> in((size_t)waves * 16 * 64), out(in.size());
    for (size_t i = 0; i < in.size(); ++i) {
        const uint64_t z = splitmix();
        in[i] = (z & 0xF00) == 0 ? gf::P - 1 : (z & 0xF00) == 0x100 ? 0 : (uint32_t)(z % gf::P);
    }
    for (int c = 0; c < 64 * 16; ++c) in[c] = gf::P - 1;  // a whole wave of p - 1
    CK(hipMemcpy(d_io, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    g.io = d_io;
    g.iters = 1;
    g.verify = 1;
    hipLaunchKernelGGL((mfma_stage_kernel<TW, ALDS>), dim3(waves / 4), dim3(256), 0, nullptr, g);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out.data(), d_io, out.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (int w = 0; w < waves; ++w)
        for (int c = 0; c < 32; ++c) {
            uint32_t xin[32];
            for (int hb = 0; hb < 2; ++hb)
                for (int r = 0; r < 16; ++r) xin[16 * hb + r] = in[((size_t)w * 16 + r) * 64 + hb * 32 + c];
            for (int hb = 0; hb < 2; ++hb)
                for (int r = 0; r < 16; ++r) {
                    const int o = row_of(hb, r);
                    uint64_t s = 0;
                    for (int i = 0; i < 32; ++i) s = (s + (uint64_t)gf::h_mul(T.W[o][i], xin[i])) % gf::P;
                    const uint32_t f = T.twraw[(0 * 2 + (TW == 0 ? 0 : hb)) * 16 + r];
                    const uint32_t want = gf::h_mul((uint32_t)s, f);
                    if (out[((size_t)w * 16 + r) * 64 + hb * 32 + c] != want) ++bad;
                }
        }
    if (bad) fprintf(stderr, "mfma stage TW=%d ALDS=%d: %zu of %zu words differ\n", TW, ALDS, bad, in.size());
    return bad == 0;
}

template <int TW, int ALDS>
static void run_mfma(const Tables& T, StageArgs g, uint32_t* d_io, int cus, int wps, double valu5_ns)
{
    const bool ok = verify<TW, ALDS>(T, g, d_io);
    const int iters = 2048, blocks = cus * wps;
    g.verify = 0;
    g.io = d_io;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    g.iters = 64;
    hipLaunchKernelGGL((mfma_stage_kernel<TW, ALDS>), dim3(blocks), dim3(256), 0, nullptr, g);
    CK(hipDeviceSynchronize());
    g.iters = iters;
    CK(hipEventRecord(e0));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((mfma_stage_kernel<TW, ALDS>), dim3(blocks), dim3(256), 0, nullptr, g);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    const double ns_per_stage_per_simd = ms * 1e6 / ((double)iters * wps);  // wave-stages a SIMD completes one after the other
    const double gvals = (double)blocks * 4 * 1024 * iters * 5 / (ms * 1e6);  // G value-levels per second, chip wide
    printf("{\"probe\":\"mfma_stage\",\"twiddle_source\":%d,\"a_from_lds\":%d,\"waves_per_simd\":%d,\"bit_exact\":%s,\"ms\":%.4f,"
           "\"ns_per_wave_stage_per_simd\":%.1f,\"G_value_levels_per_s\":%.0f,\"speedup_vs_valu5\":%.3f}\n",
           TW, ALDS, wps, ok ? "true" : "false", ms, ns_per_stage_per_simd, gvals, valu5_ns > 0 ? valu5_ns / ns_per_stage_per_simd : 0.0);
    fflush(stdout);
}

template <int ALDS>
static void run_pipe(const Tables& T, StageArgs g, uint32_t* d_io, int cus, int wps, double valu5_ns)
{
    // verification: one iteration = one stage on stream A (set 0), as in verify<2, ALDS>
    const int waves = 8;
    std::vector<uint32_t> in((size_t)waves * 16 * 64), out(in.size());
    for (size_t i = 0; i < in.size(); ++i) in[i] = (uint32_t)(splitmix() % gf::P);
    CK(hipMemcpy(d_io, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    g.io = d_io;
    g.iters = 1;
    g.verify = 1;
    hipLaunchKernelGGL((mfma_pipe_kernel<ALDS>), dim3(waves / 4), dim3(256), 0, nullptr, g);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out.data(), d_io, out.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (int w = 0; w < waves; ++w)
        for (int c = 0; c < 32; ++c) {
            uint32_t xin[32];
            for (int hb = 0; hb < 2; ++hb)
                for (int r = 0; r < 16; ++r) xin[16 * hb + r] = in[((size_t)w * 16 + r) * 64 + hb * 32 + c];
            for (int hb = 0; hb < 2; ++hb)
                for (int r = 0; r < 16; ++r) {
                    const int o = row_of(hb, r);
                    uint64_t sm = 0;
                    for (int i = 0; i < 32; ++i) sm = (sm + (uint64_t)gf::h_mul(T.W[o][i], xin[i])) % gf::P;
                    const uint32_t want = gf::h_mul((uint32_t)sm, T.twraw[hb * 16 + r]);
                    if (out[((size_t)w * 16 + r) * 64 + hb * 32 + c] != want) ++bad;
                }
        }
    const int iters = 1024, blocks = cus * wps;  // an iteration is TWO wave-stages
    g.verify = 0;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    g.iters = 64;
    hipLaunchKernelGGL((mfma_pipe_kernel<ALDS>), dim3(blocks), dim3(256), 0, nullptr, g);
    CK(hipDeviceSynchronize());
    g.iters = iters;
    CK(hipEventRecord(e0));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((mfma_pipe_kernel<ALDS>), dim3(blocks), dim3(256), 0, nullptr, g);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    const double ns = ms * 1e6 / ((double)iters * 2 * wps);
    printf("{\"probe\":\"mfma_stage_two_streams_in_one_wave\",\"twiddle_source\":2,\"a_from_lds\":%d,\"waves_per_simd\":%d,\"bit_exact\":%s,\"ms\":%.4f,"
           "\"ns_per_wave_stage_per_simd\":%.1f,\"speedup_vs_valu5\":%.3f}\n",
           ALDS, wps, bad == 0 ? "true" : "false", ms, ns, valu5_ns / ns);
    fflush(stdout);
}

template <int PER>
static void run_floor(StageArgs g, uint32_t* d_io, int cus, int wps, double valu5_ns)
{
    const int iters = 1024, blocks = cus * wps;  // an iteration is TWO wave-stages
    g.verify = 0;
    g.io = d_io;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    g.iters = 64;
    hipLaunchKernelGGL((mfma_pipe_floor_kernel<PER>), dim3(blocks), dim3(256), 0, nullptr, g);
    CK(hipDeviceSynchronize());
    g.iters = iters;
    CK(hipEventRecord(e0));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((mfma_pipe_floor_kernel<PER>), dim3(blocks), dim3(256), 0, nullptr, g);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    const double ns = ms * 1e6 / ((double)iters * 2 * wps);
    printf("{\"probe\":\"mfma_stage_pipelined_floor\",\"what\":\"no memory traffic, fragments in registers, uniform twiddles, order pinned: 1 MFMA then %d VALU\","
           "\"waves_per_simd\":%d,\"ms\":%.4f,\"ns_per_wave_stage_per_simd\":%.1f,\"speedup_vs_valu5\":%.3f}\n",
           PER, wps, ms, ns, valu5_ns / ns);
    fflush(stdout);
}

static double run_valu5(uint32_t* d_io, const uint32_t* d_tw, int cus, int wps)
{
    const int iters = 2048, blocks = cus * wps;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(valu5_kernel, dim3(blocks), dim3(256), 0, nullptr, d_io, d_tw, 64);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(valu5_kernel, dim3(blocks), dim3(256), 0, nullptr, d_io, d_tw, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    const double ns = ms * 1e6 / ((double)iters * wps);
    printf("{\"probe\":\"valu5\",\"waves_per_simd\":%d,\"ms\":%.4f,\"ns_per_wave_stage_per_simd\":%.1f,\"G_value_levels_per_s\":%.0f,"
           "\"G_butterflies_per_s\":%.0f}\n",
           wps, ms, ns, (double)blocks * 4 * 1024 * iters * 5 / (ms * 1e6), (double)blocks * 4 * 512 * iters * 5 / (ms * 1e6));
    fflush(stdout);
    return ns;
}

int main(int argc, char** argv)
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("{\"probe\":\"device\",\"name\":\"%s\",\"cus\":%d,\"clock_mhz\":%d}\n", prop.name, cus, prop.clockRate / 1000);
    Tables T;
    build(T);
    v4i *d_a, *d_init;
    uint32_t *d_tw, *d_io;
    CK(hipMalloc(&d_a, T.a.size()));
    CK(hipMalloc(&d_init, T.init.size() * 4));
    CK(hipMalloc(&d_tw, T.tw.size() * 4));
    CK(hipMalloc(&d_io, (size_t)8 * 16 * 64 * 4 + 4096));
    CK(hipMemcpy(d_a, T.a.data(), T.a.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_init, T.init.data(), T.init.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tw, T.tw.data(), T.tw.size() * 4, hipMemcpyHostToDevice));
    StageArgs g{d_a, d_init, d_tw, d_io, 1, 1};
    std::vector<int> occ;
    for (int i = 1; i < argc; ++i) occ.push_back(atoi(argv[i]));
    if (occ.empty()) occ = {1, 2, 3, 4};
    // the baseline at the occupancy the tile kernels run at (8 waves per SIMD) and at the probes' own
    const double base8 = run_valu5(d_io, d_tw, cus, 8);
    for (int wps : occ) {
        const double base = run_valu5(d_io, d_tw, cus, wps);
        const double ref = base8 < base ? base8 : base;  // both are time per wave-stage and SIMD: the better baseline counts
        run_mfma<0, 0>(T, g, d_io, cus, wps, ref);
        run_mfma<1, 0>(T, g, d_io, cus, wps, ref);
        run_mfma<2, 0>(T, g, d_io, cus, wps, ref);
        run_mfma<3, 0>(T, g, d_io, cus, wps, ref);
        run_mfma<2, 1>(T, g, d_io, cus, wps, ref);
        run_mfma<3, 1>(T, g, d_io, cus, wps, ref);
        run_floor<7>(g, d_io, cus, wps, ref);
        run_floor<9>(g, d_io, cus, wps, ref);
        run_pipe<0>(T, g, d_io, cus, wps, ref);
        run_pipe<1>(T, g, d_io, cus, wps, ref);
    }
    return 0;
}
