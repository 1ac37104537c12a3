// proto_mid_mfma.hip — VERDICT r05 item 1 (b): the encode's MID pass (512-block tiles: 9 DIF levels, the per-block factor, 9 DIT levels; what
// csrc/tile_kernels.hip runs as ntt_tile_kernel<9,5,true,MODE_MID,2>) with the butterflies on the MATRIX CORES.  Stand-alone: builds its own tables,
// runs the pass over a 2 GiB stripe in place, checks sampled columns bit for bit against radix-2 integer arithmetic on the host (the levels of
// ntt.cpp:251-284) and reports the time.  tools/microbench_mfma_dft.hip has the stage arithmetic and why it is exact.
//
// A run of radix-2 levels on 32 blocks is a 32 x 32 matrix F applied per word column, followed (DIF) or preceded (DIT) by per-block factors:
//     DIF levels l+4..l :  y = diag(c_b) F x          DIT levels l..l+4 :  y = F diag(d_b) x          (b = position below the run's stride)
// F does not depend on b.  The factors of a DIT run are moved into the run before it, so every STAGE is "y[rho] = f[element] * sum_i F[rho][i] x[i]"
// with f = (own DIF factors) x (per-block factor D between the halves) x (the next run's DIT factors).  MID over 512 blocks = four stages:
//     S1  DIF levels 8..4 (blocks q = wt + 16 i)        f = w_512^-(b bitrev5(rho)), b = wt                       -> LDS
//     S2  DIF levels 3..0 (blocks q = 32 u + i, two 16-point groups as ONE block-diagonal 32 x 32 matrix)   f = D[position]   -> LDS (same blocks)
//     S3  DIT levels 0..3 (same blocks)                  f = S4's input factors w_512^(b bitrev5(i4))            -> LDS
//     S4  DIT levels 4..8 (blocks q = wt + 16 i)        f = 1                                                    -> HBM
// F and the factors are derived NUMERICALLY from the host's radix-2 levels (unit vectors), not from closed forms.
//
// Kernel: a workgroup of 4 waves owns a tile of 512 blocks x 32 words in 64 KiB of LDS (+ 16 KiB for the stage's matrix fragments): two workgroups
// per CU.  Per stage a wave takes 4 of the 16 wave-tiles (32 blocks x 32 words: lane = (half h, column c), register r = block 16 h + r of the run;
// the results come out in the D layout of v_mfma_i32_32x32x32_i8: register r of half h = row 8 (r / 4) + 4 h + r % 4).
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

constexpr uint32_t CSHIFT = 1u << 30;  // both accumulator pairs start from this: L' = S_L + 2^30, H' = S_H + 2^30 in (0, 2^31)

struct MidArgs {
    const uint32_t* in;
    uint32_t* out;
    const v4i* frag[4];   // per stage: [4 planes][4 chunks][64 lanes]
    const v4u* fac[4];    // per stage: {f 2^32, f 2^48, K, 0} per element in the order the lanes consume them: [tile & mask][wave-tile][half][16]
    uint32_t fac_mask[4]; // tile-independent tables: 0
    uint32_t S, ld;
    uint32_t col_chunks, tiles;
};

__device__ __forceinline__ int row_of(int half, int r) { return 8 * (r >> 2) + 4 * half + (r & 3); }

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// One stage on one wave-tile: x (16 registers, canonical or any uint32) -> y (canonical), in the D layout.
__device__ __forceinline__ void mfma_stage(uint32_t (&x)[16], const v4i (&a)[16], const v4u* __restrict__ fac, const v16i& cinit)
{
    v4i b[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) b[m][e] = (int)(x[4 * m + e] ^ 0x80808080u);
    v16i acc[4];
    acc[0] = cinit;
    acc[2] = cinit;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[1][e] = 0, acc[3][e] = 0;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int d = 0; d < 4; ++d) acc[d] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[d * 4 + m], b[m], acc[d], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const v4u q = fac[r];
        const uint32_t L = (uint32_t)acc[0][r] + ((uint32_t)acc[1][r] << 8);
        const uint32_t H = (uint32_t)acc[2][r] + ((uint32_t)acc[3][r] << 8);
        uint64_t t = (((uint64_t)q[3]) << 32) | q[2];
        t += (uint64_t)L * q[0];
        t += (uint64_t)H * q[1];
        const uint32_t lo = (uint32_t)t, hi = (uint32_t)(t >> 32);
        const uint32_t mq = lo + (lo << 20);
        const uint32_t qq = __umulhi(mq, gf::P);
        uint32_t res;
        const bool borrow = __builtin_usub_overflow(hi, qq, &res);
        x[r] = borrow ? res + gf::P : res;
    }
}

constexpr int LDS_WORDS = 512 * 32 + 16 * 64 * 4;

__global__ __launch_bounds__(256) void mid9_mfma_kernel(const MidArgs g)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* tile = lds;
    v4i* stage_a = reinterpret_cast<v4i*>(lds + 512 * 32);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u, c = lane & 31u, half = lane >> 5;
    const uint32_t t = blockIdx.x;
    const uint32_t cc = t % g.col_chunks, grp = t / g.col_chunks;
    const size_t origin = (size_t)grp * 512 * g.ld + cc * 32 + c;
    v16i cinit;
#pragma unroll
    for (int e = 0; e < 16; ++e) cinit[e] = (int)CSHIFT;
    v4i a[16];
    uint32_t x[16];

    auto load_matrix = [&](int st, bool first) {
        if (!first) lds_barrier();  // every wave has finished the previous stage: its tile writes are in LDS and its fragments are in registers
#pragma unroll
        for (int i = 0; i < 4; ++i) stage_a[threadIdx.x + 256 * i] = g.frag[st][threadIdx.x + 256 * i];
        __syncthreads();
#pragma unroll
        for (int f = 0; f < 16; ++f) a[f] = stage_a[f * 64 + lane];
    };
    auto fac_of = [&](int st, uint32_t wt) { return g.fac[st] + ((((size_t)(grp & g.fac_mask[st]) * 16 + wt) * 2 + half) * 16); };

    // S1: blocks q = wt + 16 i, i = 16 half + r, from HBM
    load_matrix(0, true);
#pragma unroll 1
    for (int j = 0; j < 4; ++j) {
        const uint32_t wt = wave * 4 + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = g.in[origin + (size_t)(wt + 16 * (16 * half + r)) * g.ld];
        mfma_stage(x, a, fac_of(0, wt), cinit);
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[(wt + 16 * row_of(half, r)) * 32 + c] = x[r];
    }
    // S2, S3: blocks q = 32 u + i, in place
    for (int st = 1; st <= 2; ++st) {
        load_matrix(st, false);
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
            const uint32_t u = wave * 4 + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = tile[(32 * u + 16 * half + r) * 32 + c];
            mfma_stage(x, a, fac_of(st, u), cinit);
#pragma unroll
            for (int r = 0; r < 16; ++r) tile[(32 * u + row_of(half, r)) * 32 + c] = x[r];
        }
    }
    // S4: blocks q = wt + 16 i, to HBM
    load_matrix(3, false);
#pragma unroll 1
    for (int j = 0; j < 4; ++j) {
        const uint32_t wt = wave * 4 + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = tile[(wt + 16 * (16 * half + r)) * 32 + c];
        mfma_stage(x, a, fac_of(3, wt), cinit);
#pragma unroll
        for (int r = 0; r < 16; ++r) g.out[origin + (size_t)(wt + 16 * row_of(half, r)) * g.ld] = x[r];
    }
}

// ---------------------------------------------------------------- host ----------------------------------------------------------------
static inline int h_row_of(int half, int r) { return 8 * (r >> 2) + 4 * half + (r & 3); }
static inline uint32_t h_add(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a + b) % gf::P); }
static inline uint32_t h_sub(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a + gf::P - b) % gf::P); }

// radix-2 levels on a vector of `len` blocks (one word column), exactly as the tile kernels run them (ntt.cpp:251-284):
// level with half-size h: pairs (p, p + h), twiddle (root of order 2h)^(p mod h)
static void dif_level(std::vector<uint32_t>& v, int h, uint32_t root_2h)
{
    const int len = (int)v.size();
    for (int b0 = 0; b0 < len; b0 += 2 * h) {
        uint32_t w = 1;
        for (int i = 0; i < h; i++) {
            const uint32_t u = v[b0 + i], x = v[b0 + i + h];
            v[b0 + i] = h_add(u, x);
            v[b0 + i + h] = gf::h_mul(h_sub(u, x), w);
            w = gf::h_mul(w, root_2h);
        }
    }
}
static void dit_level(std::vector<uint32_t>& v, int h, uint32_t root_2h)
{
    const int len = (int)v.size();
    for (int b0 = 0; b0 < len; b0 += 2 * h) {
        uint32_t w = 1;
        for (int i = 0; i < h; i++) {
            const uint32_t u = v[b0 + i], x = gf::h_mul(v[b0 + i + h], w);
            v[b0 + i] = h_add(u, x);
            v[b0 + i + h] = h_sub(u, x);
            w = gf::h_mul(w, root_2h);
        }
    }
}

struct HostStage {
    uint32_t F[32][32];           // F[rho][i]
    std::vector<int8_t> frag;     // [4][4][64][16]
    uint32_t abar[32];            // sum over the K-slots of the balanced representatives (mod p) x 128: the xor's share per output row
};

// digit planes of F for the B layout "lane (half, c), register r = input i = 16 half + r", output row rho in the D layout
static void make_fragments(HostStage& st)
{
    st.frag.assign(4 * 4 * 64 * 16, 0);
    for (int rho = 0; rho < 32; ++rho) st.abar[rho] = 0;
    for (int m = 0; m < 4; ++m)
        for (int lane = 0; lane < 64; ++lane)
            for (int t = 0; t < 16; ++t) {
                const int rho = lane & 31, hb = lane >> 5, r = t >> 2, j = t & 3, in = 16 * hb + 4 * m + r;
                uint32_t val = st.F[rho][in];
                for (int s = 0; s < j; ++s) val = gf::h_mul(val, 256);
                int64_t bal = val <= 2139062143u ? (int64_t)val : (int64_t)val - (int64_t)gf::P;
                st.abar[rho] = h_add(st.abar[rho], val);
                for (int d = 0; d < 4; ++d) {
                    int64_t dig = ((bal % 256) + 256) % 256;
                    if (dig >= 128) dig -= 256;
                    bal = (bal - dig) / 256;
                    st.frag[(((size_t)d * 4 + m) * 64 + lane) * 16 + t] = (int8_t)dig;
                }
                if (bal != 0) {
                    fprintf(stderr, "balanced digits do not close\n");
                    exit(1);
                }
            }
}

// {f 2^32, f 2^48, K, 0}: y = (L' f~1 + H' f~2 + K) / 2^32 with L' = S_L + 2^30, H' = S_H + 2^30 and the xor's share 128 abar[rho]:
//   K = f~1 (128 abar[rho] - 2^30 (1 + 2^16))  (mod p)
static void make_factor(uint32_t f, uint32_t abar_rho, uint32_t (&q)[4])
{
    const uint32_t f1 = gf::h_to_mont(f);
    q[0] = f1;
    q[1] = gf::h_to_mont(gf::h_mul(f, 65536));
    const uint32_t shift = gf::h_mul(CSHIFT % gf::P, 65537);
    q[2] = gf::h_mul(f1, h_sub(gf::h_mul(128, abar_rho), shift));
    q[3] = 0;
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 19;
    const uint32_t S = argc > 2 ? atoi(argv[2]) : 1024;
    const int reps = argc > 3 ? atoi(argv[3]) : 20;
    const size_t N = (size_t)1 << n;
    const uint32_t w_dit = gf::h_root(512), w_dif = gf::h_inv(w_dit);
    auto root = [&](uint32_t w512, int order) { return gf::h_pow(w512, 512 / order); };

    // ---- stage matrices from unit vectors through the host's own levels ----
    HostStage st[4];
    std::vector<uint32_t> f1tab(512), f3tab(512);  // S1's own factors and S4's input factors by position q
    {
        // S1: DIF levels h = 256..16 on 512 points, b = q & 15, i = q >> 4
        for (int i = 0; i < 32; ++i) {
            std::vector<uint32_t> v(512, 0);
            v[16 * i] = 1;
            for (int h = 256; h >= 16; h >>= 1) dif_level(v, h, root(w_dif, 2 * h));
            for (int rho = 0; rho < 32; ++rho) st[0].F[rho][i] = v[16 * rho];
        }
        for (int b = 0; b < 16; ++b) {
            std::vector<uint32_t> v(512, 0);
            v[b] = 1;  // i = 0: F[rho][0] = 1, so what arrives at 16 rho + b is the factor itself
            for (int h = 256; h >= 16; h >>= 1) dif_level(v, h, root(w_dif, 2 * h));
            for (int rho = 0; rho < 32; ++rho) f1tab[16 * rho + b] = v[16 * rho + b];
        }
        // S2: DIF levels h = 8..1 on 32 consecutive points (two groups of 16)
        for (int i = 0; i < 32; ++i) {
            std::vector<uint32_t> v(32, 0);
            v[i] = 1;
            for (int h = 8; h >= 1; h >>= 1) dif_level(v, h, root(w_dif, 2 * h));
            for (int rho = 0; rho < 32; ++rho) st[1].F[rho][i] = v[rho];
        }
        // S3: DIT levels h = 1..8 on 32 consecutive points
        for (int i = 0; i < 32; ++i) {
            std::vector<uint32_t> v(32, 0);
            v[i] = 1;
            for (int h = 1; h <= 8; h <<= 1) dit_level(v, h, root(w_dit, 2 * h));
            for (int rho = 0; rho < 32; ++rho) st[2].F[rho][i] = v[rho];
        }
        // S4: DIT levels h = 16..256 on 512 points: y = F diag(d_b) x
        for (int i = 0; i < 32; ++i) {
            std::vector<uint32_t> v(512, 0);
            v[16 * i] = 1;
            for (int h = 16; h <= 256; h <<= 1) dit_level(v, h, root(w_dit, 2 * h));
            for (int rho = 0; rho < 32; ++rho) st[3].F[rho][i] = v[16 * rho];
        }
        for (int q = 0; q < 512; ++q) {
            std::vector<uint32_t> v(512, 0);
            v[q] = 1;
            for (int h = 16; h <= 256; h <<= 1) dit_level(v, h, root(w_dit, 2 * h));
            f3tab[q] = v[q & 15];  // row rho = 0 of F is all ones
        }
        for (int s = 0; s < 4; ++s) make_fragments(st[s]);
    }
    // per-block factor D by position (random: the pass must work for any table)
    std::vector<uint32_t> dplain(N);
    uint64_t s = 99;
    for (size_t i = 0; i < N; i++) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        dplain[i] = (uint32_t)((s >> 16) % gf::P);
    }
    // factor tables in consumption order: [tile][wave-tile][half][r]
    const size_t ntiles = N >> 9;
    std::vector<uint32_t> fac[4];
    const uint32_t mask[4] = {0u, (uint32_t)(ntiles - 1), 0u, 0u};
    for (int stg = 0; stg < 4; ++stg) {
        const size_t tl = (size_t)mask[stg] + 1;
        fac[stg].resize(tl * 16 * 2 * 16 * 4);
        for (size_t t = 0; t < tl; ++t)
            for (int wt = 0; wt < 16; ++wt)
                for (int hb = 0; hb < 2; ++hb)
                    for (int r = 0; r < 16; ++r) {
                        const int rho = h_row_of(hb, r);
                        const int q = (stg == 0 || stg == 3) ? wt + 16 * rho : 32 * wt + rho;  // the block this result is
                        uint32_t f = 1;
                        if (stg == 0) f = f1tab[q];
                        if (stg == 1) f = dplain[t * 512 + q];
                        if (stg == 2) f = f3tab[q];
                        uint32_t qd[4];
                        make_factor(f, st[stg].abar[rho], qd);
                        memcpy(&fac[stg][((((t * 16 + wt) * 2 + hb) * 16) + r) * 4], qd, 16);
                    }
    }

    // data: pseudo-random words, plus tiles of extreme values
    std::vector<uint32_t> host(N * S);
    for (size_t i = 0; i < host.size(); i++) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        host[i] = (uint32_t)((s >> 16) % gf::P);
    }
    for (size_t t = 0; t < ntiles && t < 24; t++)
        for (size_t q = 0; q < 512; q++)
            for (uint32_t col = 0; col < S; col++) {
                const uint32_t msk = (uint32_t)(t * 37 + 1) & 511u;
                const bool on = t == 0 ? true : (__builtin_popcount((uint32_t)q & msk) & 1);
                if (t < 12) host[(t * 512 + q) * S + col] = on ? gf::P - 1 : 0;
                else if (col & 1) host[(t * 512 + q) * S + col] = on ? gf::P - 1 : (uint32_t)(q * 2654435761u) % gf::P;
            }
    uint32_t* d_x;
    CK(hipMalloc(&d_x, host.size() * 4));
    CK(hipMemcpy(d_x, host.data(), host.size() * 4, hipMemcpyHostToDevice));
    MidArgs a{};
    a.in = d_x;
    a.out = d_x;
    for (int stg = 0; stg < 4; ++stg) {
        void *df, *dq;
        CK(hipMalloc(&df, st[stg].frag.size()));
        CK(hipMemcpy(df, st[stg].frag.data(), st[stg].frag.size(), hipMemcpyHostToDevice));
        CK(hipMalloc(&dq, fac[stg].size() * 4));
        CK(hipMemcpy(dq, fac[stg].data(), fac[stg].size() * 4, hipMemcpyHostToDevice));
        a.frag[stg] = (const v4i*)df;
        a.fac[stg] = (const v4u*)dq;
        a.fac_mask[stg] = mask[stg];
    }
    a.S = S;
    a.ld = S;
    a.col_chunks = S / 32;
    a.tiles = (uint32_t)(ntiles * a.col_chunks);
    const int lds_bytes = LDS_WORDS * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mid9_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipLaunchKernelGGL(mid9_mfma_kernel, dim3(a.tiles), dim3(256), lds_bytes, nullptr, a);
    CK(hipDeviceSynchronize());

    std::vector<uint32_t> got(host.size());
    CK(hipMemcpy(got.data(), d_x, got.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0, checked = 0;
    std::vector<uint32_t> col(512);
    for (size_t t = 0; t < ntiles; t += (t < 24 ? 1 : ntiles / 7 + 1))
        for (uint32_t cix = 0; cix < S; cix += (t < 24 ? 7 : 131)) {
            for (int q = 0; q < 512; q++) col[q] = host[(t * 512 + q) * S + cix];
            for (int h = 256; h >= 1; h >>= 1) dif_level(col, h, root(w_dif, 2 * h));
            for (int q = 0; q < 512; q++) col[q] = gf::h_mul(col[q], dplain[(t << 9) + q]);
            for (int h = 1; h <= 256; h <<= 1) dit_level(col, h, root(w_dit, 2 * h));
            for (int q = 0; q < 512; q++) {
                bad += got[(t * 512 + q) * S + cix] != col[q];
                checked++;
            }
        }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(mid9_mfma_kernel, dim3(a.tiles), dim3(256), lds_bytes, nullptr, a);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(mid9_mfma_kernel, dim3(a.tiles), dim3(256), lds_bytes, nullptr, a);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double bytes = 2.0 * N * S * 4;
    printf("{\"probe\":\"proto_mid9_mfma\",\"log2_blocks\":%d,\"words_per_block\":%u,\"bit_exact\":%s,\"words_checked\":%zu,\"words_wrong\":%zu,"
           "\"ms\":%.4f,\"algorithmic_TBps\":%.3f,\"workgroup\":\"4 waves, 80 KiB LDS\",\"stages\":4}\n",
           n, S, bad == 0 ? "true" : "false", checked, bad, ms, bytes / ms / 1e9);
    return bad == 0 ? 0 : 1;
}
