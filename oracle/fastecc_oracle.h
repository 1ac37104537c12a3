/*
 * fastecc_oracle.h — CPU ORACLE (TEST INFRASTRUCTURE ONLY).
 *
 * Plain-C restatement of the FastECC encode path over GF(0xFFF00001).  It exists so
 * that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can check the
 * HIP path bit-for-bit.  Nothing under fastecc_amd/ may include, link or call it.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle.py) against
 *   - the reference's published known-answer hash  (Benchmarks.md:491,499,507),
 *   - golden hashes/values recorded from the unmodified reference (SURVEY.md App. B,
 *     tests/golden/), and
 *   - the reference itself compiled here into oracle/_ref/ (see oracle/Makefile).
 */
#ifndef FASTECC_ORACLE_H
#define FASTECC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_P 0xFFF00001u /* 2^32 - 2^20 + 1 (GF.md:20) */

/* ---- field arithmetic, follows GF(p).cpp ---- */
uint32_t orc_gf_add(uint32_t x, uint32_t y);  /* GF(p).cpp:44-48  */
uint32_t orc_gf_sub(uint32_t x, uint32_t y);  /* GF(p).cpp:37-42  */
uint32_t orc_gf_mul(uint32_t x, uint32_t y);  /* GF(p).cpp:110-127 (Barrett-32 form) */
uint32_t orc_gf_mul_wide(uint32_t x, uint32_t y); /* GF(p).cpp:99-104 (64-bit %) */
uint32_t orc_gf_pow(uint32_t x, uint32_t n);  /* GF(p).cpp:254-264 */
uint32_t orc_gf_root(uint32_t order);         /* GF(p).cpp:268-276, generator 19 */
uint32_t orc_gf_inv(uint32_t x);              /* GF(p).cpp:293-297 */

/* ---- transforms over a block-major matrix data[N][size] (size u32 words per block) ---- */
/* O(N^2) definition, natural order in and out, unscaled (ntt.cpp:451-483). */
void orc_slow_ntt(uint32_t *data, size_t N, size_t size, int inverse);
/* Radix-2 DIT with bit-reversed block addressing (ntt.cpp:251-318), natural order out, unscaled. */
void orc_ntt(uint32_t *data, size_t N, size_t size, int inverse);
/* Same transform, one column at a time on a transposed copy (fast enough for N=2^19). */
void orc_ntt_fast(uint32_t *data, size_t N, size_t size, int inverse);

/* block i *= scale * base^i  (the RS.cpp:51-59 loop with scale=1/N, base=root(2N)). */
void orc_scale_blocks(uint32_t *data, size_t N, size_t size, uint32_t scale, uint32_t base);

/* In-place Reed-Solomon encode: data[N][size] -> parity[N][size] (RS.cpp:40-63). */
void orc_encode(uint32_t *data, size_t N, size_t size);
/* Same result through orc_ntt_fast (used for large N). */
void orc_encode_fast(uint32_t *data, size_t N, size_t size);
/* any order N | p-1 (mixed radix): the composition of RS.cpp:40-63 by the O(N^2) definition, and through a transform of
 * order q 2^m with the odd factor outermost; orc_ntt_mixed is that transform (natural order in and out) */
void orc_encode_slow(uint32_t *data, size_t N, size_t size);
void orc_ntt_mixed(uint32_t *data, size_t N, size_t size, int inverse);
void orc_encode_mixed(uint32_t *data, size_t N, size_t size);
/* Parity straight from the mathematical contract parity[j] = f(w_2N^(2j+1)), O(N^2). */
void orc_encode_by_definition(const uint32_t *data, uint32_t *parity, size_t N, size_t size);

/* Rolling 32-bit checksum over blocks in logical order (main.cpp:202-212). */
uint32_t orc_hash(const uint32_t *data, size_t nwords);

/* Input generators used by the reference and by SURVEY.md Appendix B. */
void orc_fill_linear(uint32_t *data, size_t nwords);                 /* i % P (RS.cpp:28-29) */
void orc_fill_splitmix(uint32_t *data, size_t nwords, uint64_t seed); /* splitmix64 % P        */

/* Data packing GF.md:72-104 (prose only upstream: format defined in include/fastecc.h, parity unpinned).
 * raw: `words` arbitrary uint32 (words <= 1024); packed: words + 1 uint32, all < P. */
void orc_pack_block(const uint32_t *raw, size_t words, uint32_t *packed);
int orc_unpack_block(const uint32_t *packed, size_t words, uint32_t *raw); /* -1: not a packer output */
void orc_pack_blocks(const uint32_t *raw, size_t N, size_t words, uint32_t *packed);
size_t orc_unpack_blocks(const uint32_t *packed, size_t N, size_t words, uint32_t *raw); /* count of bad blocks */

/* Erasure decoding by plain Lagrange interpolation (checker for fastecc_decode; the reference has no decoder).
 * Erased data blocks (data_present[i] == 0) are recomputed in place from the first N surviving positions. */
int orc_decode(uint32_t *data, const uint32_t *parity, const uint8_t *data_present, const uint8_t *parity_present, size_t N,
               size_t size);

int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
