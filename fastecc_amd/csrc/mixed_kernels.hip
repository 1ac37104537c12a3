// mixed_kernels.hip — the odd-radix passes for q in {2, 3, 5, 7, 9, 13, 15}: dispatch, and the host-side constant tables of every q.
#include <algorithm>
#include <functional>

#include "mixed_device.hpp"

namespace fastecc {

// The fused kernel reaches the whole batch of `rows` blocks through one buffer descriptor with 32-bit byte offsets.
bool fused_batch_fits(uint64_t ld, uint64_t S, uint64_t rows) { return rows * std::max(ld, S) * 4ull < 0xFFFF0000ull; }

hipError_t launch_fused(int q, int levels, bool dit, FusedArgs a, hipStream_t st)
{
    if (fused_rlog(q, levels) == 0 || a.M == 0 || (a.M >> levels) == 0) return hipErrorInvalidValue;
    if (!fused_batch_fits(a.ld, a.S, (uint64_t)q * a.M)) return hipErrorInvalidValue;
    a.col_chunks = (a.S + 63u) / 64u;
    const uint64_t tiles = (uint64_t)(a.M >> levels) * a.col_chunks;
    if (tiles == 0 || tiles > 0x7FFFFFFFull) return hipErrorInvalidValue;
    switch (q) {
        case 2: return dit ? launch_fused_q<2, true>(levels, a, (unsigned)tiles, st) : launch_fused_q<2, false>(levels, a, (unsigned)tiles, st);
        case 3: return dit ? launch_fused_q<3, true>(levels, a, (unsigned)tiles, st) : launch_fused_q<3, false>(levels, a, (unsigned)tiles, st);
        case 5: return dit ? launch_fused_q<5, true>(levels, a, (unsigned)tiles, st) : launch_fused_q<5, false>(levels, a, (unsigned)tiles, st);
        case 7: return dit ? launch_fused_q<7, true>(levels, a, (unsigned)tiles, st) : launch_fused_q<7, false>(levels, a, (unsigned)tiles, st);
        case 9: return dit ? launch_fused_q<9, true>(levels, a, (unsigned)tiles, st) : launch_fused_q<9, false>(levels, a, (unsigned)tiles, st);
        case 13: return dit ? launch_fused_q<13, true>(levels, a, (unsigned)tiles, st) : launch_fused_q<13, false>(levels, a, (unsigned)tiles, st);
        case 15: return dit ? launch_fused_q<15, true>(levels, a, (unsigned)tiles, st) : launch_fused_q<15, false>(levels, a, (unsigned)tiles, st);
        default: return launch_fused_pfa(q, levels, dit, a, (unsigned)tiles, st);
    }
}

// Host: the constants small_dft<q> reads, for the primitive q-th root wq (forward or inverse), Montgomery form.
std::vector<uint32_t> radix_dft_table(int q, uint32_t wq)
{
    const uint32_t inv2 = (uint32_t)((gf::P + 1ull) / 2);
    auto sym = [&](std::vector<uint32_t>& t, int n, uint32_t w) {  // C then S tables of the n-point transform with root w
        const int h2 = (n - 1) / 2;
        const size_t base = t.size();
        t.resize(base + 2 * (size_t)h2 * h2);
        const uint32_t wi = gf::h_inv(w);
        for (int i = 1; i <= h2; i++)
            for (int j = 1; j <= h2; j++) {
                const uint32_t a = gf::h_pow(w, (uint64_t)i * j), b = gf::h_pow(wi, (uint64_t)i * j);
                const uint32_t sum = (uint32_t)(((uint64_t)a + b) % gf::P), dif = (uint32_t)(((uint64_t)a + gf::P - b) % gf::P);
                t[base + (size_t)(i - 1) * h2 + (j - 1)] = gf::h_to_mont(gf::h_mul(sum, inv2));
                t[base + (size_t)h2 * h2 + (size_t)(i - 1) * h2 + (j - 1)] = gf::h_to_mont(gf::h_mul(dif, inv2));
            }
    };
    std::vector<uint32_t> t;
    std::function<void(int, uint32_t)> append = [&](int n, uint32_t w) {  // the constants of the n-point transform with root w, as small_dft_at<n> reads them
        if (n == 9) {
            sym(t, 3, gf::h_pow(w, 3));
            for (int i2 = 1; i2 <= 2; i2++)
                for (int j1 = 1; j1 <= 2; j1++) t.push_back(gf::h_to_mont(gf::h_pow(w, (uint64_t)i2 * j1)));
            // order read by the kernel: slots 3+1 (i2=1,j1=1), 6+1 (i2=1,j1=2), 3+2 (i2=2,j1=1), 6+2 (i2=2,j1=2)
        } else if (pfa_first(n)) {
            const int qa = pfa_first(n), qb = n / qa;
            append(qa, gf::h_pow(w, (uint64_t)qb));
            append(qb, gf::h_pow(w, (uint64_t)qa));
        } else {
            sym(t, n, w);
        }
    };
    if (q == 2) t.push_back(0);  // the two-point transform has no constants
    else append(q, wq);
    return t;
}

template <int V>
static hipError_t launch_v(int q, bool dit, const RadixArgs& a, dim3 grid, hipStream_t st)
{
    switch (q) {
        case 2: return launch_q<2, V>(dit, a, grid, st);
        case 3: return launch_q<3, V>(dit, a, grid, st);
        case 5: return launch_q<5, V>(dit, a, grid, st);
        case 7: return launch_q<7, V>(dit, a, grid, st);
        case 9: return launch_q<9, V>(dit, a, grid, st);
        case 13:
            if constexpr (V == 1) return launch_q<13, 1>(dit, a, grid, st);
            else return hipErrorInvalidValue;
        case 15:
            if constexpr (V == 1) return launch_q<15, 1>(dit, a, grid, st);
            else return hipErrorInvalidValue;
        default:
            if constexpr (V == 1) return launch_radix_pfa(q, dit, a, grid, st);
            else return hipErrorInvalidValue;
    }
}

bool radix_supported(int q) { return q == 2 || q == 3 || q == 5 || q == 7 || q == 9 || q == 13 || (q >= 15 && q <= 117 && pfa_first(q) != 0); }
int fused_rlog(int q, int levels) { return radix_supported(q) ? fused_shape_rlog(q, levels) : 0; }

hipError_t launch_radix(int q, bool dit, int vec, RadixArgs a, hipStream_t st)
{
    if (!radix_supported(q) || a.M == 0) return hipErrorInvalidValue;
    if (q > 9) vec = 1;
    a.col_chunks = (a.S + 64u * vec - 1u) / (64u * vec);
    a.items = (uint64_t)a.col_chunks * a.M;
    const uint64_t blocks = (a.items + 3u) / 4u;
    if (blocks == 0 || blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const dim3 grid((unsigned)blocks);
    switch (vec) {
        case 1: return launch_v<1>(q, dit, a, grid, st);
        case 2: return launch_v<2>(q, dit, a, grid, st);
        case 4: return launch_v<4>(q, dit, a, grid, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace fastecc
