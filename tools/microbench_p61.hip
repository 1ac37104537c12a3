// microbench_p61.hip — measured inputs for the question "would a 3-trip plan pay for GF((2^61-1)^2)?" (BASELINE configs[4]; not part of the library).
//
// The 5-trip plan (dif7, dif6, mid6, dit6, dit7) keeps every twiddle WAVE-UNIFORM: a lane is one 16-byte element column, the wave's
// blocks are the same for all lanes, the limbs of a twiddle live in SGPRs.  A 3-trip plan needs 1024-block tiles; with 16-byte
// elements that only fits LDS when a row of the tile is 8 element columns (128 bytes), i.e. when a wave holds 8 columns x 8 BLOCKS.
// Then, in every tile half,
//   * the 3 levels whose partners sit in other lanes need a cross-lane exchange of 16-byte elements, and
//   * the twiddles of the levels above them depend on the lane's block: they are per-LANE values — fetched by vector loads and
//     split into limbs per lane instead of once per wave on the scalar unit.
// Probes (all exact GF((2^61-1)^2) arithmetic from csrc/gf61.hpp, 8 independent butterflies per lane and iteration):
//   uniform      (a, b) -> (a + b, (a - b) w), w wave-uniform (SGPR limbs)                        — what the 5-trip kernels do
//   lane_limbs   the same with w's nine limbs in VGPRs, split once outside the loop               — per-lane twiddle reused over many butterflies
//   lane_split   the same, the twiddle a fresh per-lane (c, d) every use (limb split inside)       — per-lane twiddle used once (collected twiddles)
//   lane_load    lane_split with (c, d) fetched from a per-lane table in memory (L2-resident)       — plus the vector loads
//   swap32/16    a butterfly whose partner is lane ^ 32 / lane ^ 16: v_permlane32_swap / v_permlane16_swap of the four words
//   dpp8         partner lane ^ 8 via ds_bpermute-free DPP row moves (row_ror:8 inside a row of 16)
// Output: one JSON line per probe with G butterflies / s chip-wide; tools/p61_three_trip_projection.py turns them into a time.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <functional>
#include <vector>

#include "gf61.hpp"

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));                \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

using gf61::Elem;

static float time_ms(int reps, const std::function<void()>& fn)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    fn();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(e0, nullptr));
        fn();
        CK(hipEventRecord(e1, nullptr));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    return best;
}

enum { UNIFORM, LANE_LIMBS, LANE_SPLIT, LANE_LOAD, SWAP32, SWAP16, DPP8, NVAR };
static const char* NAME[] = {"uniform twiddle (SGPR limbs): the 5-trip kernels' butterfly",
                             "per-lane twiddle, limbs in VGPRs (split once, reused)",
                             "per-lane twiddle, split into limbs at every use",
                             "per-lane twiddle, loaded from a table and split at every use",
                             "uniform twiddle, partner in lane ^ 32 (8 x v_permlane32_swap per butterfly: operand in, result back)",
                             "uniform twiddle, partner in lane ^ 16 (8 x v_permlane16_swap per butterfly)",
                             "uniform twiddle, partner in lane ^ 8 (8 x DPP row_ror:8 per butterfly)"};

__device__ __forceinline__ uint64_t next61(uint64_t z)  // cheap per-iteration change of a twiddle word, stays < p
{
    z = z * 3u + 1u;
    return (z & gf61::P) == gf61::P ? 1 : (z & gf61::P);
}

template <int LANES_XOR>
__device__ __forceinline__ uint32_t xor_lane(uint32_t mine, uint32_t& other_out)
{
    // returns the partner lane's word; v_permlane*_swap exchange a register between the two halves (32) / between rows of 16
    if constexpr (LANES_XOR == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(mine, mine, false, false);
        other_out = r[1];
        return (threadIdx.x & 32u) ? r[0] : r[1];
    } else if constexpr (LANES_XOR == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(mine, mine, false, false);
        other_out = r[1];
        return (threadIdx.x & 16u) ? r[0] : r[1];
    } else {
        // lane ^ 8 inside a row of 16 lanes = rotate the row by 8
        other_out = 0;
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0x128 /* row_ror:8 */, 0xF, 0xF, false);
    }
}

template <int LANES_XOR>
__device__ __forceinline__ Elem partner(Elem x)
{
    uint32_t dummy;
    Elem y;
    const uint32_t a = xor_lane<LANES_XOR>((uint32_t)x.re, dummy), b = xor_lane<LANES_XOR>((uint32_t)(x.re >> 32), dummy);
    const uint32_t c = xor_lane<LANES_XOR>((uint32_t)x.im, dummy), d = xor_lane<LANES_XOR>((uint32_t)(x.im >> 32), dummy);
    y.re = ((uint64_t)b << 32) | a;
    y.im = ((uint64_t)d << 32) | c;
    return y;
}

template <int VAR>
__global__ __launch_bounds__(256) void bfly61_kernel(uint64_t* out, const uint64_t* __restrict__ table, int iters, uint64_t c0, uint64_t d0)
{
    const gf61::Opaque k = gf61::make_opaque();
    Elem a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        a[i] = Elem{(threadIdx.x * 0x9E3779B97F4A7C15ull + i * 40503u) & gf61::P, (threadIdx.x * 40503ull + i * 0xBF58476D1CE4E5B9ull) & (gf61::P >> 1)};
        b[i] = Elem{(threadIdx.x * 0x94D049BB133111EBull + i * 77u) & (gf61::P >> 1), (threadIdx.x * 12345ull + i * 0x9E3779B97F4A7C15ull + 5u) & (gf61::P >> 1)};
    }
    uint64_t c = c0, d = d0;                                   // wave-uniform (SALU)
    uint64_t lc = (c0 + threadIdx.x * 977u) & (gf61::P >> 1);  // per-lane
    uint64_t ld = (d0 + threadIdx.x * 131u) & (gf61::P >> 1);
    const gf61::Twiddle wl = gf61::make_twiddle(lc, ld);       // LANE_LIMBS: limbs in VGPRs for the whole loop
    const uint64_t* mine = table + (size_t)(threadIdx.x & 255u) * 2;
    for (int it = 0; it < iters; it++) {
        const gf61::Twiddle wu = gf61::make_twiddle(c, d);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            Elem x = a[i], y = b[i];
            // cross-lane levels: the second operand comes from the partner lane and the second result goes back to it — 4 words each way
            // (a pair tile swaps two registers of the two half-waves before and after the butterfly: the same eight moves per butterfly)
            if constexpr (VAR == SWAP32) y = partner<32>(y);
            if constexpr (VAR == SWAP16) y = partner<16>(y);
            if constexpr (VAR == DPP8) y = partner<8>(y);
            const Elem dlt = gf61::sub_raw(x, y);
            a[i] = gf61::add(x, y, k);
            if constexpr (VAR == UNIFORM || VAR >= SWAP32) b[i] = gf61::mul_raw(dlt, wu, k);
            if constexpr (VAR == SWAP32) b[i] = partner<32>(b[i]);
            if constexpr (VAR == SWAP16) b[i] = partner<16>(b[i]);
            if constexpr (VAR == DPP8) b[i] = partner<8>(b[i]);
            if constexpr (VAR == LANE_LIMBS) b[i] = gf61::mul_raw(dlt, wl, k);
            if constexpr (VAR == LANE_SPLIT) {
                b[i] = gf61::mul_raw(dlt, gf61::make_twiddle(lc, ld), k);
                lc = next61(lc);
            }
            if constexpr (VAR == LANE_LOAD) {
                const uint64_t tc = mine[((it * 8 + i) & 63) * 512], td = mine[((it * 8 + i) & 63) * 512 + 1];  // 64 x 256 entries of 16 bytes, L2-resident
                b[i] = gf61::mul_raw(dlt, gf61::make_twiddle(tc, td), k);
            }
        }
        c = next61(c);
        d = next61(d);
    }
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) acc ^= gf61::canon(a[i].re) ^ gf61::canon(a[i].im) ^ gf61::canon(b[i].re) ^ gf61::canon(b[i].im);
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int VAR>
static void run(uint64_t* d_out, const uint64_t* d_table, int blocks)
{
    const int iters = 256;
    const float ms = time_ms(5, [&] {
        hipLaunchKernelGGL(bfly61_kernel<VAR>, dim3(blocks), dim3(256), 0, nullptr, d_out, d_table, iters, 0x123456789ABCDEFull & gf61::P, 0x0FEDCBA987654321ull & gf61::P);
    });
    const double bf = (double)blocks * 256 * iters * 8;
    printf("{\"probe\":\"p61_bfly\",\"variant\":\"%s\",\"ms\":%.4f,\"Gbfly_per_s\":%.1f}\n", NAME[VAR], ms, bf / ms / 1e6);
    fflush(stdout);
}

int main()
{
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    printf("{\"probe\":\"device\",\"name\":\"%s\",\"cus\":%d,\"clock_mhz\":%d}\n", pr.name, pr.multiProcessorCount, pr.clockRate / 1000);
    const int blocks = pr.multiProcessorCount * 8;
    uint64_t *d_out, *d_table;
    CK(hipMalloc((void**)&d_out, (size_t)blocks * 256 * 8));
    std::vector<uint64_t> t(64 * 512);
    for (size_t i = 0; i < t.size(); i++) t[i] = (i * 0x9E3779B97F4A7C15ull + 12345u) & (gf61::P >> 1);
    CK(hipMalloc((void**)&d_table, t.size() * 8));
    CK(hipMemcpy(d_table, t.data(), t.size() * 8, hipMemcpyHostToDevice));
    run<UNIFORM>(d_out, d_table, blocks);
    run<LANE_LIMBS>(d_out, d_table, blocks);
    run<LANE_SPLIT>(d_out, d_table, blocks);
    run<LANE_LOAD>(d_out, d_table, blocks);
    run<SWAP32>(d_out, d_table, blocks);
    run<SWAP16>(d_out, d_table, blocks);
    run<DPP8>(d_out, d_table, blocks);
    run<UNIFORM>(d_out, d_table, blocks);  // again: warm clocks
    return 0;
}
