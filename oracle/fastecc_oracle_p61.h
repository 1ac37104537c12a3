/*
 * fastecc_oracle_p61.h — CPU ORACLE for the 64-bit field configuration (TEST INFRASTRUCTURE ONLY).
 *
 * BASELINE.json configs[4] asks for "(n,k)=(2^20,2^19), 64 KB blocks, GF(p=2^61-1) 64-bit field".
 * The reference has NO code for it: RS.cpp:86 instantiates GF(0xFFF00001) only, the closest thing is the
 * ring mod 2^64-1 (GF(p).cpp:202-222, root 283-290) whose largest power-of-two order is 65536, and its
 * documentation only names the idea (README.md:178 "GF(p^2) with p = 2^61-1", GF.md:30-31).
 * GF(2^61-1) itself has no element of order 2^20 (p-1 = 2*(2^60-1)); its quadratic extension
 * GF(p^2) = GF(p)[i]/(i^2+1) (p = 3 mod 4) has multiplicative order p^2-1 = 2^62*(2^60-1), so the same
 * NTT encoder runs there with 128-bit elements.
 *
 * PARITY STATUS: UNPINNED.  No reference implementation, test or golden vector exists for this
 * configuration.  What this oracle restates is the reference's COMPOSITION (RS.cpp:40-63: unscaled
 * inverse transform, block i *= w_2N^i / N, forward transform; transform definition ntt.cpp:451-483)
 * over the field above, with the conventions we had to choose ourselves:
 *     element   (re, im) = two consecutive little-endian uint64 words, both canonical in [0, p)
 *     generator g = 4 + i   (smallest a >= 0 such that a + i is a non-square; norm 17)
 *     w_(2^62)  = g^(2^60-1),   root of order 2^t = w_(2^62)^(2^(62-t))        [w_4 = i, w_8 = 2^30(1+i)]
 * The oracle itself is checked against an independent pure-Python big-integer statement of the same
 * definition (tests/golden/make_golden_p61.py -> tests/golden/golden_p61.json) and against the
 * mathematical contract parity[j] = f(w_2N^(2j+1)) evaluated directly in O(N^2).
 */
#ifndef FASTECC_ORACLE_P61_H
#define FASTECC_ORACLE_P61_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC61_P 0x1FFFFFFFFFFFFFFFull /* 2^61 - 1 */

/* ---- GF(p) ---- */
uint64_t orc61_add(uint64_t x, uint64_t y);
uint64_t orc61_sub(uint64_t x, uint64_t y);
uint64_t orc61_mul(uint64_t x, uint64_t y); /* 128-bit product, % p */

/* ---- GF(p^2); z[0] = re, z[1] = im; out may alias an input ---- */
void orc61c_mul(const uint64_t x[2], const uint64_t y[2], uint64_t out[2]);
void orc61c_pow(const uint64_t x[2], uint64_t e, uint64_t out[2]);
void orc61c_inv(const uint64_t x[2], uint64_t out[2]);
/* Root of unity of order `order` (a power of two <= 2^62); (0,0) when there is none. */
void orc61c_root(uint64_t order, uint64_t out[2]);

/* ---- transforms over a block-major matrix data[N][elems] of GF(p^2) elements (2*elems u64 per block) ---- */
void orc61_slow_ntt(uint64_t *data, size_t N, size_t elems, int inverse); /* O(N^2) definition, ntt.cpp:451-483 */
void orc61_ntt(uint64_t *data, size_t N, size_t elems, int inverse);      /* radix-2, natural order out, unscaled */
void orc61_scale_blocks(uint64_t *data, size_t N, size_t elems, const uint64_t scale[2], const uint64_t base[2]);
void orc61_encode(uint64_t *data, size_t N, size_t elems);                /* RS.cpp:40-63 composition, in place */
void orc61_encode_by_definition(const uint64_t *data, uint64_t *parity, size_t N, size_t elems);

/* O(N^2) Lagrange erasure decoder of the (2N,N) code (position u <-> w_2N^u; data at even, parity at odd positions):
 * rewrites the erased data blocks in place from the first N survivors; -1 when fewer than N blocks survive */
int orc61_decode(uint64_t *data, const uint64_t *parity, const uint8_t *data_present, const uint8_t *parity_present, size_t N, size_t elems);

/* splitmix64 % p fill (same generator as the 32-bit oracle) */
void orc61_fill_splitmix(uint64_t *data, size_t nwords, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
