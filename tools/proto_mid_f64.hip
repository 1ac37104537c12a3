// proto_mid_f64.hip — prototype of the encode's MID pass (512-block tiles: 9 DIF levels, the per-block factor, 9 DIT levels) with the butterflies on
// the FP64 pipe (tools/microbench_f64.hip has the arithmetic and why it is exact).  Stand-alone: builds its own tables, runs the pass over a
// 2 GiB stripe in place, checks sampled columns against exact integer arithmetic on the host and reports the time beside the library's
// integer MID9 (1.55 ms on the same stripe).
//
// A tile is 512 blocks x 32 words; a workgroup of 16 waves holds 16 values per lane as doubles.  Both halves use the butterfly
// (a, b) -> (a + w b, a - w b), whose outputs grow by p/2 per level (no reduction inside a run of levels):
//   first half   natural order in, bit-reversed out, twiddle chosen by the HIGH bits of the block index (the remainder tree of X^512 - 1):
//                the half-wave bit is block bit 0, so no twiddle depends on it before the last level
//   second half  bit-reversed in, natural out, twiddle chosen by the LOW bits (the usual decimation in time): half-wave bit = block bit 8
// Values pass through LDS as int32 in (-p/2, p/2].
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

struct TwF64 {
    double w, wp;  // w balanced (|w| <= p/2), wp = RN(-w 2^32 / p)
};
using const_f64_ptr = const double __attribute__((address_space(4)))*;
struct const_tw_ptr {  // twiddle pairs read through the constant address space: wave-uniform indices become scalar loads
    const_f64_ptr p;
    __device__ __forceinline__ TwF64 operator[](size_t i) const { return TwF64{p[2 * i], p[2 * i + 1]}; }
    __device__ __forceinline__ const_tw_ptr operator+(size_t i) const { return const_tw_ptr{p + 2 * i}; }
};

struct MidF64Args {
    const uint32_t* in;
    uint32_t* out;
    const TwF64* tw;
    const double* ptw;   // the twiddles of the two cross-lane levels as the workgroup keeps them in LDS: [first half, second half][wave][half-wave][8] pairs
    const double* dfac;  // per position of one stripe, balanced
    uint32_t S, ld;
    int n;
    uint32_t col_chunks, tiles;
    int cache_policy;
};
enum { DIFA = 0, DIFB = 16, DITB = 272, DITA = 288, TW_ENTRIES = 544, PTW_PAIRS = 512, LDS_BYTES = 512 * 32 * 4 + PTW_PAIRS * 16 + 512 * 8 };

static constexpr double PD = 4293918721.0;
#define F64_MAGIC 0x1.8p84
#define F64_CFOLD (-1048575.0 / 4294967296.0)
#define F64_NPINV32 (-(4294967296.0 / 4293918721.0))
#define F64_P_OVER (4293918721.0 / 4294967296.0)

__device__ __forceinline__ double mulmod(double b, double w, double wp)
{
    const double u = __builtin_fma(b, wp, F64_MAGIC);
    const double qn = u - F64_MAGIC;
    const double t1 = __builtin_fma(b, w, qn);
    return __builtin_fma(qn, F64_CFOLD, t1);
}
__device__ __forceinline__ double reduce(double x)
{
    const double u = __builtin_fma(x, F64_NPINV32, F64_MAGIC);
    const double qn = u - F64_MAGIC;
    return __builtin_fma(qn, F64_P_OVER, x);
}
__device__ __forceinline__ void swap_halves(double& v0, double& v1)
{
    const uint64_t a = __builtin_bit_cast(uint64_t, v0), b = __builtin_bit_cast(uint64_t, v1);
    const auto lo = __builtin_amdgcn_permlane32_swap((uint32_t)a, (uint32_t)b, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((uint32_t)(a >> 32), (uint32_t)(b >> 32), false, false);
    v0 = __builtin_bit_cast(double, ((uint64_t)hi[0] << 32) | lo[0]);
    v1 = __builtin_bit_cast(double, ((uint64_t)hi[1] << 32) | lo[1]);
}

// four levels, twiddle by the block index above the partner bit (entry 2^d + block)
template <bool UNIT0>
__device__ __forceinline__ void high_levels(double (&x)[16], const_tw_ptr T)
{
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int half = 8 >> d;
#pragma unroll
        for (int cb = 0; cb < (1 << d); ++cb) {
            const TwF64 t = T[(1 << d) + cb];
#pragma unroll
            for (int i = 0; i < half; ++i) {
                const int ja = cb * 2 * half + i, jb = ja + half;
                const double a = x[ja];
                const double m = (UNIT0 && cb == 0) ? x[jb] : mulmod(x[jb], t.w, t.wp);
                x[ja] = a + m;
                x[jb] = a - m;
            }
        }
    }
}
// four levels, twiddle by the index below the partner bit (entry 2^t + m)
template <bool UNIT0>
__device__ __forceinline__ void low_levels(double (&x)[16], const_tw_ptr T)
{
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int half = 1 << t;
#pragma unroll
        for (int m = 0; m < half; ++m) {
            const TwF64 tw = T[half + m];
#pragma unroll
            for (int j0 = 0; j0 < 16; j0 += 2 * half) {
                const int ja = j0 + m, jb = ja + half;
                const double a = x[ja];
                const double b = (UNIT0 && m == 0) ? x[jb] : mulmod(x[jb], tw.w, tw.wp);
                x[ja] = a + b;
                x[jb] = a - b;
            }
        }
    }
}

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_desc(const uint32_t* p)
{
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    void* q = reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, 0xFFFFFFFFu, 0x00020000);
}

__global__ __launch_bounds__(1024, 8) void mid9_f64_kernel(const MidF64Args a)
{
    extern __shared__ __attribute__((aligned(16))) int32_t lds[];  // [512][32] words, then the cross-lane twiddles, then the tile's 512 factors
    double* const lds_ptw = reinterpret_cast<double*>(lds + 512 * 32);
    double* const lds_d = lds_ptw + PTW_PAIRS * 2;
    const uint32_t g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u, c = lane & 31u, hf = lane >> 5;
    const uint32_t tile = __builtin_amdgcn_readfirstlane(blockIdx.x);
    uint32_t cc = tile % a.col_chunks;
    const uint32_t hi = tile / a.col_chunks;
    if ((a.col_chunks & 7u) == 0) cc = (cc & 7u) * (a.col_chunks >> 3) + (cc >> 3);  // an XCD takes a contiguous run of column chunks
    const uint32_t dead = (cc * 32 + c < a.S) ? 0u : 0xFFFFFFFFu;
    const size_t origin = ((size_t)hi << 9) * a.ld + cc * 32;
    const __amdgpu_buffer_rsrc_t din = make_desc(a.in + origin), dout = make_desc(a.out + origin);
    const uint32_t row_bytes = a.ld * 4u;
    const const_tw_ptr T{(const_f64_ptr)(reinterpret_cast<uintptr_t>(a.tw))};

    // what depends on the half-wave goes through LDS: a lane reads its own copy, no per-lane selects (read after the first exchange's barrier)
    lds_ptw[threadIdx.x] = a.ptw[threadIdx.x];
    if (threadIdx.x < 512) lds_d[threadIdx.x] = a.dfac[(((size_t)(hi & ((1u << (a.n - 9)) - 1u))) << 9) + threadIdx.x];
    double x[16];
    {  // block j*32 + 2g + hf
        const uint32_t voff = ((hf * a.ld + c) * 4u) | dead;
        uint32_t soff = 2u * g * row_bytes;
        const uint32_t step = 32u * row_bytes;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            x[j] = (double)__builtin_amdgcn_raw_buffer_load_b32(din, voff, soff, 2);
            soff += step;
            asm volatile("" : "+s"(soff));
        }
    }
    high_levels<true>(x, T + DIFA);
#pragma unroll
    for (int j = 0; j < 16; ++j) lds[(j * 32 + 2 * g + hf) * 32 + c] = (int32_t)reduce(x[j]);
    lds_barrier();
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = (double)lds[(g * 32 + 2 * k + hf) * 32 + c];
    high_levels<false>(x, T + (DIFB + g * 16));
    double y[16];
    {
        const TwF64* Pt = reinterpret_cast<const TwF64*>(lds_ptw) + (g * 2 + hf) * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            double va = x[k], vb = x[k + 8];
            swap_halves(va, vb);  // low half-wave: blocks 2c, 2c + 1 of pair c = 16 g + k; high: of pair c + 8
            const TwF64 t = Pt[k];
            const double m = mulmod(vb, t.w, t.wp);
            y[2 * k] = va + m;  // block 32 g + 16 hf + 2k
            y[2 * k + 1] = va - m;
        }
    }
    {
        const double* D = lds_d + g * 32 + hf * 16;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const double d = D[kk];
            y[kk] = mulmod(y[kk], d, d * F64_NPINV32);
        }
    }
    low_levels<true>(y, T + DITB);
    lds_barrier();  // every lane has read the first exchange
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) lds[(g * 32 + hf * 16 + kk) * 32 + c] = (int32_t)reduce(y[kk]);
    lds_barrier();
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = (double)lds[(hf * 256 + j * 16 + g) * 32 + c];
    low_levels<false>(x, T + (DITA + g * 16));
    {
        const TwF64* Pt = reinterpret_cast<const TwF64*>(lds_ptw) + 256 + (g * 2 + hf) * 8;
        const uint32_t voff = ((hf * 16u * a.ld + c) * 4u) | dead;
        uint32_t soff = g * row_bytes;
        const uint32_t step = 32u * row_bytes, far = 256u * row_bytes;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            double va = x[2 * i], vb = x[2 * i + 1];
            swap_halves(va, vb);  // low half-wave: blocks 16 (2i) + g and + 256; high: 16 (2i + 1) + g and + 256
            const TwF64 t = Pt[i];
            const double m = mulmod(vb, t.w, t.wp);
            const int32_t r0 = (int32_t)reduce(va + m), r1 = (int32_t)reduce(va - m);
            const uint32_t o0 = (uint32_t)r0 + ((uint32_t)(r0 >> 31) & gf::P), o1 = (uint32_t)r1 + ((uint32_t)(r1 >> 31) & gf::P);
            __builtin_amdgcn_raw_buffer_store_b32(o0, dout, voff, soff, 2);
            __builtin_amdgcn_raw_buffer_store_b32(o1, dout, voff, soff + far, 2);
            soff += step;
            asm volatile("" : "+s"(soff));
        }
    }
}

// ---- host ----
static TwF64 make_tw(uint32_t w)
{
    int64_t v = w;
    if (v > (int64_t)(gf::P >> 1)) v -= gf::P;
    TwF64 t;
    t.w = (double)v;
    t.wp = (double)(-((long double)v * 4294967296.0L / (long double)gf::P));
    return t;
}
static uint32_t brev(uint32_t v, int bits)
{
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((v >> i) & 1u) << (bits - 1 - i);
    return r;
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 19;
    const uint32_t S = argc > 2 ? atoi(argv[2]) : 1024;
    const size_t N = (size_t)1 << n;
    const uint32_t w_dit = gf::h_root(512), w_dif = gf::h_inv(w_dit);
    std::vector<TwF64> tw(TW_ENTRIES), ptw(PTW_PAIRS);
    auto rdif = [&](int order_log, uint32_t e) { return gf::h_pow(gf::h_pow(w_dif, 1u << (9 - order_log)), e); };  // (root of order 2^order_log)^e
    auto rdit = [&](int order_log, uint32_t e) { return gf::h_pow(gf::h_pow(w_dit, 1u << (9 - order_log)), e); };
    for (int d = 0; d < 4; d++)
        for (uint32_t cb = 0; cb < (1u << d); cb++) tw[DIFA + (1 << d) + cb] = make_tw(rdif(d + 1, brev(cb, d)));
    for (uint32_t g = 0; g < 16; g++) {
        for (int dd = 0; dd < 4; dd++)
            for (uint32_t cl = 0; cl < (1u << dd); cl++) {
                const int d = 4 + dd;
                tw[DIFB + g * 16 + (1 << dd) + cl] = make_tw(rdif(d + 1, brev((g << dd) | cl, d)));
            }
        for (uint32_t h = 0; h < 2; h++)
            for (uint32_t i = 0; i < 8; i++) {
                ptw[(g * 2 + h) * 8 + i] = make_tw(rdif(9, brev(g * 16 + i + 8 * h, 8)));
                ptw[256 + (g * 2 + h) * 8 + i] = make_tw(rdit(9, (2 * i + h) * 16 + g));
            }
        for (int t = 0; t < 4; t++)
            for (uint32_t m = 0; m < (1u << t); m++) tw[DITA + g * 16 + (1 << t) + m] = make_tw(rdit(4 + t + 1, (m << 4) + g));
    }
    for (int t = 0; t < 4; t++)
        for (uint32_t m = 0; m < (1u << t); m++) tw[DITB + (1 << t) + m] = make_tw(rdit(t + 1, m));
    std::vector<uint32_t> dplain(N);
    std::vector<double> dfac(N);
    uint64_t s = 99;
    for (size_t i = 0; i < N; i++) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        dplain[i] = (uint32_t)((s >> 16) % gf::P);
        dfac[i] = make_tw(dplain[i]).w;
    }
    // data: pseudo-random words, plus tiles of extreme values (all p - 1; rows alternating 0 / p - 1 by bit patterns of the block index)
    std::vector<uint32_t> host(N * S);
    for (size_t i = 0; i < host.size(); i++) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        host[i] = (uint32_t)((s >> 16) % gf::P);
    }
    const size_t ntiles = N >> 9;
    for (size_t t = 0; t < ntiles && t < 24; t++)
        for (size_t q = 0; q < 512; q++)
            for (uint32_t col = 0; col < S; col++) {
                const uint32_t mask = (uint32_t)(t * 37 + 1) & 511u;
                const bool on = t == 0 ? true : (__builtin_popcount((uint32_t)q & mask) & 1);
                if (t < 12) host[(t * 512 + q) * S + col] = on ? gf::P - 1 : 0;
                else if (col & 1) host[(t * 512 + q) * S + col] = on ? gf::P - 1 : (uint32_t)(q * 2654435761u) % gf::P;
            }
    uint32_t* d_x;
    TwF64 *d_tw, *d_ptw;
    double* d_df;
    CK(hipMalloc(&d_x, host.size() * 4));
    CK(hipMalloc(&d_tw, tw.size() * sizeof(TwF64)));
    CK(hipMalloc(&d_df, dfac.size() * 8));
    CK(hipMalloc(&d_ptw, ptw.size() * sizeof(TwF64)));
    CK(hipMemcpy(d_ptw, ptw.data(), ptw.size() * sizeof(TwF64), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_x, host.data(), host.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tw, tw.data(), tw.size() * sizeof(TwF64), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_df, dfac.data(), dfac.size() * 8, hipMemcpyHostToDevice));
    MidF64Args a{};
    a.in = d_x;
    a.out = d_x;
    a.tw = d_tw;
    a.dfac = d_df;
    a.ptw = reinterpret_cast<const double*>(d_ptw);
    a.S = S;
    a.ld = S;
    a.n = n;
    a.col_chunks = (S + 31) / 32;
    a.tiles = (uint32_t)(ntiles * a.col_chunks);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mid9_f64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipLaunchKernelGGL(mid9_f64_kernel, dim3(a.tiles), dim3(1024), LDS_BYTES, nullptr, a);
    CK(hipDeviceSynchronize());
    // check: every column of the extreme tiles' first chunk, a few columns of others
    std::vector<uint32_t> got(host.size());
    CK(hipMemcpy(got.data(), d_x, got.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0, checked = 0;
    std::vector<uint32_t> col(512);
    for (size_t t = 0; t < ntiles; t += (t < 24 ? 1 : ntiles / 7 + 1))
        for (uint32_t cix = 0; cix < S; cix += (t < 24 ? 7 : 131)) {
            for (int q = 0; q < 512; q++) col[q] = host[(t * 512 + q) * S + cix];
            // decimation in frequency, natural in, bit-reversed out
            for (int h = 256; h >= 1; h >>= 1) {
                const uint32_t root = gf::h_pow(w_dif, 256 / h);
                for (int b0 = 0; b0 < 512; b0 += 2 * h) {
                    uint32_t w = 1;
                    for (int i = 0; i < h; i++) {
                        const uint32_t u = col[b0 + i], v = col[b0 + i + h];
                        col[b0 + i] = (uint32_t)(((uint64_t)u + v) % gf::P);
                        col[b0 + i + h] = gf::h_mul((uint32_t)(((uint64_t)u + gf::P - v) % gf::P), w);
                        w = gf::h_mul(w, root);
                    }
                }
            }
            for (int q = 0; q < 512; q++) col[q] = gf::h_mul(col[q], dplain[((t << 9) + q) & (N - 1)]);
            for (int h = 1; h <= 256; h <<= 1) {
                const uint32_t root = gf::h_pow(w_dit, 256 / h);
                for (int b0 = 0; b0 < 512; b0 += 2 * h) {
                    uint32_t w = 1;
                    for (int i = 0; i < h; i++) {
                        const uint32_t u = col[b0 + i], v = gf::h_mul(col[b0 + i + h], w);
                        col[b0 + i] = (uint32_t)(((uint64_t)u + v) % gf::P);
                        col[b0 + i + h] = (uint32_t)(((uint64_t)u + gf::P - v) % gf::P);
                        w = gf::h_mul(w, root);
                    }
                }
            }
            for (int q = 0; q < 512; q++) {
                bad += got[(t * 512 + q) * S + cix] != col[q];
                checked++;
            }
        }
    // timing (in place on whatever the buffer holds now: values stay canonical)
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(mid9_f64_kernel, dim3(a.tiles), dim3(1024), LDS_BYTES, nullptr, a);
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(mid9_f64_kernel, dim3(a.tiles), dim3(1024), LDS_BYTES, nullptr, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("{\"probe\":\"mid9_f64\",\"log2_blocks\":%d,\"words_per_block\":%u,\"checked_words\":%zu,\"mismatches\":%zu,\"ms\":%.4f,\"GBps_algorithmic\":%.1f}\n", n, S, checked, bad,
           ms, 2.0 * N * S * 4 / ms / 1e6);
    return bad ? 1 : 0;
}
