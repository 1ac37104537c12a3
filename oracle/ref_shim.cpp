// ref_shim.cpp — ORACLE BUILD ONLY (test infrastructure).
//
// Wraps the UNMODIFIED FastECC reference in extern "C" entry points so that tests can compare our
// restatement (fastecc_oracle.c) and the HIP path with the real thing, and so that bench.py can time
// the reference's own CPU path (cpu_baseline.kind == "reference").
//
// No reference source is copied: the reference's translation unit is #included from where it lies
// (-I/root/reference, see oracle/Makefile), with its main() renamed out of the way.  The output of
// this file only ever goes to oracle/_ref/ (git-ignored).
#define main fastecc_reference_ntt_main   // main.cpp:338 — keep the CLI entry from clashing
#include "main.cpp"                        // pulls in GF(p).cpp, ntt.cpp, LargePages.cpp, wall_clock_timer.h
#undef main

#include <vector>

namespace {
using T = uint32_t;
constexpr T P = 0xFFF00001u;   // RS.cpp:86

// The reference API works on an array of block pointers (ntt.cpp:348-350).
std::vector<T*> block_pointers(T* base, size_t N, size_t size)
{
    std::vector<T*> p(N);
    for (size_t i = 0; i < N; i++) p[i] = base + i * size;
    return p;
}

// A single MFA_NTT/Rec_NTT call leaves the pointer array permuted; gather the blocks
// back so that logical block j is physically block j.
void gather_logical(T* base, const std::vector<T*>& ptrs, size_t size)
{
    const size_t N = ptrs.size();
    bool identity = true;
    for (size_t i = 0; i < N; i++) identity &= (ptrs[i] == base + i * size);
    if (identity) return;
    std::vector<T> tmp(N * size);
    for (size_t i = 0; i < N; i++) memcpy(tmp.data() + i * size, ptrs[i], size * sizeof(T));
    memcpy(base, tmp.data(), N * size * sizeof(T));
}
}  // namespace

extern "C" {

uint32_t ref_gf_add(uint32_t x, uint32_t y) { return GF_Add<T, P>(x, y); }
uint32_t ref_gf_sub(uint32_t x, uint32_t y) { return GF_Sub<T, P>(x, y); }
uint32_t ref_gf_mul(uint32_t x, uint32_t y) { return GF_Mul<T, P>(x, y); }
uint32_t ref_gf_pow(uint32_t x, uint32_t n) { return GF_Pow<T, P>(x, n); }
uint32_t ref_gf_root(uint32_t n) { return GF_Root<T, P>(n); }
uint32_t ref_gf_inv(uint32_t x) { return GF_Inv<T, P>(x); }

// which: 0 = MFA_NTT (ntt.cpp:382), 1 = Rec_NTT (ntt.cpp:349), 2 = Slow_NTT (ntt.cpp:451)
void ref_ntt(uint32_t* data, size_t N, size_t size, int inverse, int which)
{
    if (which == 2) {
        Slow_NTT<T, P>(data, N, size, inverse != 0);
        return;
    }
    auto ptrs = block_pointers(data, N, size);
    if (which == 1) Rec_NTT<T, P>(ptrs.data(), N, size, inverse != 0);
    else            MFA_NTT<T, P>(ptrs.data(), N, size, inverse != 0);
    gather_logical(data, ptrs, size);
}

// The encode the reference benchmarks (RS.cpp:39-67), as a call sequence on the reference's own
// functions: inverse transform, per-block factor root(2N)^i / N, forward transform.
void ref_encode(uint32_t* data, size_t N, size_t size)
{
    auto ptrs = block_pointers(data, N, size);
    T** blk = ptrs.data();
    MFA_NTT<T, P>(blk, N, size, true);
    const T w2N = GF_Root<T, P>(T(2 * N));
    const T invN = GF_Inv<T, P>(T(N));
#pragma omp parallel for
    for (ptrdiff_t i = 0; i < ptrdiff_t(N); i++) {
        const T f = GF_Mul<T, P>(invN, GF_Pow<T, P>(w2N, T(i)));
        T* b = blk[i];
        for (size_t k = 0; k < size; k++) b[k] = GF_Mul<T, P>(b[k], f);
    }
    MFA_NTT<T, P>(blk, N, size, false);
    gather_logical(data, ptrs, size);
}

// The reference's small-order codelets, f[0..order) in place: NTT2 / NTT4 (ntt.cpp:16-22, 50-62; natural order out — NTT4 ends
// with the swap of its two middle values) and the odd-order ones (ntt.cpp:25-44, 113-146), which none of its drivers reaches.
// order 2, 3, 4 or 9; anything else leaves f untouched and returns -1.
int ref_small_ntt(uint32_t* f, int order, int inverse)
{
    if (order == 2) {  // its own inverse up to the factor 1/2, which the reference's transforms never apply
        NTT2<T, P>(f[0], f[1]);
        return 0;
    }
    if (order == 4) {
        if (inverse) NTT4<T, P, true>(f[0], f[1], f[2], f[3]);
        else         NTT4<T, P, false>(f[0], f[1], f[2], f[3]);
        return 0;
    }
    if (order == 3) {
        if (inverse) NTT3<T, P, true>(f[0], f[1], f[2]);
        else         NTT3<T, P, false>(f[0], f[1], f[2]);
        return 0;
    }
    if (order == 9) {
        if (inverse) NTT9<T, P, true>(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7], f[8]);
        else         NTT9<T, P, false>(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7], f[8]);
        return 0;
    }
    return -1;
}

// main.cpp:202-212 through the reference's own template.
uint32_t ref_hash(uint32_t* data, size_t N, size_t size)
{
    auto ptrs = block_pointers(data, N, size);
    return hash<T>(ptrs.data(), N, size);
}

int ref_simd_level(void)
{
#ifdef SIMD
    return SIMD;
#else
    return 0;
#endif
}

}  // extern "C"
