/*
 * fastecc.h — C ABI of libfastecc_hip.so: the MI355X (gfx950) NTT Reed-Solomon encode path.
 *
 * This is the drop-in boundary for FastECC's encode hot path.  The reference has no ABI: the path is
 * C++ templates #included into RS.cpp.  Each entry point below names the reference code it replaces
 * (file:line in Bulat-Ziganshin/FastECC); INTEGRATION.md shows the host-side binding.
 *
 * Conventions
 *   - Field: GF(p), p = 0xFFF00001 (RS.cpp:86).  Words are native little-endian uint32, every input
 *     word must be < p (README.md:160-162); every output word is the canonical value in [0,p).
 *   - Code: the reference's (n,k) = (2N,N), N = 2^m, 1 <= m <= 19 (RS.cpp:36,82; GF.md:20).  A "block" is
 *     block_bytes/4 consecutive words; a stripe is N blocks back to back (block-major), exactly the
 *     layout RS.cpp:28-33 builds.  parity block j = f(w_2N^(2j+1)) where f interpolates the data
 *     blocks at the powers of w_N = 19^((p-1)/N)   (RS.cpp:40-63).  Other (n,k) — see fastecc_create — evaluate the same
 *     f on other points; a second field (FASTECC_FIELD_GF_P61_SQUARED) exists for 64-bit words.
 *   - All functions return FASTECC_OK (0) or a negative FASTECC_E_* code; no exceptions cross the
 *     ABI and nothing is printed.  The reference returns void and has no error path (RS.cpp:26 prints
 *     and returns on allocation failure); preconditions it leaves implicit are checked here.
 *   - A context is bound to one HIP device (fastecc_create) or to a set of devices (fastecc_create_sharded) and
 *     one (n,k,block_bytes).  Calls on one context are serialised internally (a second thread blocks until the
 *     first call has returned) and nothing about a call is kept in the context, so a context may be shared
 *     between threads and streams; device work of calls that go through the context's internal buffers is
 *     ordered between streams by the library.  Distinct contexts are independent.
 *   - There is NO CPU fallback: without a usable HIP device fastecc_create fails with
 *     FASTECC_E_DEVICE.
 */
#ifndef FASTECC_H
#define FASTECC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FASTECC_VERSION 200 /* 0.2.0 */

enum {
    FASTECC_OK = 0,
    FASTECC_E_INVAL = -1,       /* bad argument (n != 2k, k not a power of two, block_bytes % 4, null pointer, ...) */
    FASTECC_E_NOMEM = -2,       /* host or device allocation failed */
    FASTECC_E_DEVICE = -3,      /* no HIP device / HIP runtime error */
    FASTECC_E_UNSUPPORTED = -4  /* field or size outside what GF(0xFFF00001) admits (k > 2^19), or an entry point the
                                   context kind does not offer */
};

enum {
    FASTECC_FIELD_GF_FFF00001 = 0,   /* the only field RS.cpp instantiates (RS.cpp:86) */
    /*
     * GF(p^2), p = 2^61 - 1: the "64-bit field" of the 64 KB-block configuration.  The reference has NO code for
     * it (README.md:178 and GF.md:30-31 only name the idea; GF(2^61-1) itself has no roots of unity of order
     * > 2), so the conventions are defined HERE and nothing upstream pins them:
     *   - an element is (re, im) = two consecutive little-endian uint64 words, i^2 = -1, both words < p;
     *     a block of block_bytes bytes holds block_bytes / 16 elements (block_bytes % 16 == 0);
     *   - roots of unity: w_(2^62) = (4 + i)^(2^60 - 1)  (4 + i is the first non-square a + i), the root of
     *     order 2^t is w_(2^62)^(2^(62-t)); so w_4 = i and w_8 = 2^30 (1 + i);
     *   - the encoder is the same composition as RS.cpp:40-63 over this field: parity block j is the value at
     *     w_2N^(2j+1) of the polynomial of degree < N whose value at w_N^i is data block i.
     * Supported by create/destroy/encode/encode_blocks/ntt/check_range/decode_prepare/decode/repair/profile/plan_string and
     * fastecc_set_plan (0 = default: LDS tiles; 1..4 = register passes with that many radix-2 levels; 10+L / 20+L = tiles with a
     * 64 / 128 KiB exchange buffer); the 32-bit-word entry points
     * (scale_blocks, gf_binary, set_option other than "decode_direct_max" and "decode_split") return FASTECC_E_UNSUPPORTED.
     * Codes: (2k,k) with k = 2^m, 1 <= m <= 24, run on the caller's stripes.  Any other k <= 2^24 with n - k <= N = 2^ceil(log2 k) follows
     * the rules fastecc_create documents for GF(0xFFF00001) — the data zero-extended to N blocks, parity block j = block j * 2^fold of the
     * (2N,N) parity, fold = min(log2 N - ceil(log2(n-k)), 4) — through padded copies of the stripes inside the context (two N-block work
     * stripes: encode, decode_prepare, decode and repair on device memory, encode also on host memory; not encode_columns / ntt /
     * check_range).  n = 4k and n = 8k with k = 2^m (more parity than data blocks, fastecc_create's coset rule and nesting order with this
     * field's roots: w_2k; w_4k, w_4k^3; w_8k, w_8k^3, w_8k^5, w_8k^7) are native transforms — the DIF half once into a k-block work stripe
     * of the context, MID and the DIT half once per coset: fastecc_encode on device and host memory (out of place only), set_plan, profile,
     * and decode_prepare / decode / repair on device and host memory: the decoder's scheme on the n-th roots of unity, any n - k lost blocks
     * (one transform of n points; neither the few-loss direct path nor the even / odd split, which are the (2k,k) code's).
     */
    FASTECC_FIELD_GF_P61_SQUARED = 1
};
enum {
    FASTECC_MEM_HOST = 0,        /* ordinary host memory: staged through HBM with copies, the call is synchronous */
    FASTECC_MEM_DEVICE = 1,      /* device memory of the context's device */
    /* pinned host memory (hipHostMalloc / hipHostRegister): fastecc_encode pipelines the stripe through HBM in column
     * slabs — strided copy-engine uploads, kernels and downloads of different slabs overlap, using both directions of the
     * link at once — and is asynchronous on `stream` like FASTECC_MEM_DEVICE.  n = 2k over GF(0xFFF00001) only; the
     * other entry points treat this kind as invalid. */
    FASTECC_MEM_HOST_PINNED = 2
};

typedef struct fastecc_ctx fastecc_ctx;

/* Human-readable text for an error code. */
const char *fastecc_strerror(int code);
/* Which HIP call failed behind the last FASTECC_E_DEVICE / _NOMEM on this thread ("" if none). */
const char *fastecc_last_error_detail(void);
int fastecc_version(void);

/*
 * Create an encoder for (n,k) with `block_bytes`-byte blocks on HIP device `device`.
 * Replaces the set-up part of EncodeReedSolomon<T,P>(N,SIZE) (RS.cpp:22-37) plus the per-call root
 * tables MFA_NTT rebuilds (ntt.cpp:397-402) and the per-block GF_Pow of RS.cpp:51-54: all twiddle
 * tables are built once here and live in HBM.
 * Requires block_bytes > 0 and a multiple of 4, 1 <= k <= 2^19 and n > k.  With N = the smallest power of two >= k:
 *   n = 2k, k = N          the reference's configuration (RS.cpp:22): parity block j = f(w_2k^(2j+1));
 *   n = k + N/2^d, d<=4    fewer parity blocks (RS.md:13-33 "output some M values"): parity block j is block j*2^d of
 *                          the (2N,N) parity, i.e. f on the coset w_2N * <w_(N/2^d)>.  The DIF half is unchanged, the
 *                          evaluation half shrinks to a size-(n-k) transform;
 *   n = 4k or 8k, k = N    more parity blocks (needs n <= 2^20): the n - k parity blocks are f on the 2^e - 1 cosets of
 *                          the data points inside the n-th roots of unity, k blocks per coset, ordered so that codes
 *                          nest: w_2k (= the (2k,k) parity), w_4k, w_4k^3, w_8k, w_8k^3, w_8k^5, w_8k^7; block j of coset
 *                          g is f(g * w_k^j).  parity == data (in place) and fastecc_encode_blocks are not possible;
 *   any other k, n-k <= N  zero extension, exactly RS.md:23-33: the k data blocks are the first k of N (blocks k..N-1 are
 *                          zero and never exist in memory), M = max(N/16, 2^ceil(log2(n-k))) parity blocks of the
 *                          (N + M, N) code above are computed and the first n - k of them are the parity (no extra
 *                          copies: the first and last kernels bound their reads / writes to the existing blocks).
 *                          GF(0xFFF00001) only; encode, encode_blocks, check_range and decode (no ntt / scale_blocks /
 *                          pack / encode_batch).
 * `parity` buffers hold n - k blocks.  Codes with n - k <= 160 (GF(0xFFF00001), option "encode_direct_max") are encoded without the
 * transform — one read of the data, the parity blocks being a dense product of the data with a weight matrix, on the matrix cores (or, for
 * rows they cannot take, up to 32 parity blocks on the VALU); the parity is the same.
 */
int fastecc_create(fastecc_ctx **out, uint64_t n, uint64_t k, uint64_t block_bytes, int field, int device);
/*
 * fastecc_create with flags.  FASTECC_CODE_MIXED_RADIX: the transform order is the smallest N1 = q * 2^m >= k with q in
 * {1, 3, 5, 7, 9, 13, 15} instead of the next power of two — the reference's roadmap for block counts that are not powers of two
 * (NTT.md:43-46, README.md:175; its NTT3 / NTT9 codelets, ntt.cpp:25-146, are never reached by its drivers).  The data
 * points are then the powers of w_N1 = 19^((p-1)/N1) and parity block j = f(w_(2 N1)^(2j+1)): the composition of
 * RS.cpp:40-63 with N1 for N (k < N1: zero extension as in RS.md:23-33; the first n - k <= N1 of the N1 parity blocks
 * are the parity).  With q = 1 this IS fastecc_create; otherwise it is a different code than the zero-extended power-of-
 * two one fastecc_create builds for the same (n,k) — a stripe must be decoded with the flags it was encoded with.
 * k <= 15 * 2^19.  GF(0xFFF00001); encode, encode_blocks, check_range, set_plan, profile, and decode_prepare / decode / repair
 * (orders above 2^20: patterns of up to 2^20 erasures — the locator's product tree needs w_T for its T padded roots and this field
 * stops at order 2^20; patterns of at most 256 lost blocks do not use the tree at any order).  The odd-radix
 * level is fused into the outermost tile passes where a shape exists (2^m with m <= 16..18 depending on q: three trips through
 * HBM, like the power-of-two orders; option "fuse_radix" = 0 gives it its own two passes); measured against zero extension in
 * profiles/r02/mixed_radix_bench.jsonl: faster than zero extension for q <= 9.
 */
#define FASTECC_CODE_MIXED_RADIX 1u
/* The same with the composite odd factors of the prime-factor map as well (NTT.md:43-46 "PFA NTT as well as NTT kernels of orders
 * 3,5,7,9,13"): q in {1, 3, 5, 7, 9, 13, 15, 21, 35, 39, 45, 63, 65, 91, 105, 117}, again the smallest q * 2^m >= k — the next order is
 * then at most 8.4 % above k (20 % with the seven q of FASTECC_CODE_MIXED_RADIX).  The q-point transforms are prime-factor compositions of
 * the 3-, 5-, 7-, 9- and 13-point ones, in registers (3 * 7, 5 * 7, 3 * 13, 9 * 5, 9 * 7, 5 * 13, 7 * 13, 15 * 7, 9 * 13: no twiddles
 * between the two factors).  Three trips through HBM up to m = 10, and for q <= 63 up to m = 13 (q = 63), 14 (35, 39, 45) or 16 (21) (the fused outer tile);
 * beyond that the odd-radix level has its own two passes.  A different code than FASTECC_CODE_MIXED_RADIX builds whenever the orders differ; the
 * other 8 odd divisors of 4095 (195 ... 4095: that many blocks per lane do not fit the registers) are not offered. */
#define FASTECC_CODE_MIXED_RADIX_PFA 4u
/* A/B experiment (same code, same results as fastecc_create): for n = 2k, k = 2^m >= 2^12 the top level of the transform is handled
 * like an odd radix with q = 2 — fused with the next levels in mixed_kernels.hip's kernel instead of tile_kernels.hip's outer tile. */
#define FASTECC_CODE_TOP_RADIX2 2u
int fastecc_create_ex(fastecc_ctx **out, uint64_t n, uint64_t k, uint64_t block_bytes, int field, int device, unsigned flags);
void fastecc_destroy(fastecc_ctx *ctx);

/*
 * encode(n,k,block_bytes, data -> parity).  Replaces the timed lambda RS.cpp:39-67:
 *     MFA_NTT(data,N,SIZE,true); block_i *= root(2N)^i / N; MFA_NTT(data,N,SIZE,false);
 * data   : k blocks, block-major, k*block_bytes bytes (read only unless parity == data)
 * parity : n - k blocks (k for the reference's (2k,k) code), same layout; parity == data gives the reference's in-place behaviour
 *          and needs n - k <= k.  Codes with few parity blocks are evaluated directly (option "encode_direct_max"), same bits.
 * mem_kind FASTECC_MEM_DEVICE: both pointers are device memory on the context's device; the work is
 *          enqueued on `stream` (a hipStream_t, NULL = default stream) and the call does not
 *          synchronise — not even the first call on a context: its twiddle tables are written by a kernel on `stream` in front of the
 *          first pass (a use on another stream later waits for that kernel on the device), so calls may be captured into a hipGraph.  FASTECC_MEM_HOST: pointers are host memory; the call stages through HBM and
 *          returns when `parity` is complete.
 */
int fastecc_encode(fastecc_ctx *ctx, const void *data, void *parity, int mem_kind, void *stream);

/*
 * The same for the word columns [col0_words, col0_words + width_words) of every block only (DEVICE memory, both
 * pointers are the stripes' base addresses).  The columns of a stripe are independent transforms (ntt.cpp:348-350),
 * so a host can cut a stripe into column slabs and overlap their encodes with whatever moves the slabs — which is
 * what fastecc_encode does for FASTECC_MEM_HOST_PINNED stripes and fastecc_encode_sharded for the xGMI gather.
 * n = 2k = 2^m in either field (the 64-bit field: ranges of whole 16-byte elements, i.e. multiples of 4 words);
 * FASTECC_E_UNSUPPORTED for the other codes (they work through whole-stripe scratch buffers), and the caller encodes the
 * stripe in one piece.  Any range is accepted; multiples of 32 words keep every 128-byte row segment whole.
 */
int fastecc_encode_columns(fastecc_ctx *ctx, const void *data, void *parity, uint64_t col0_words, uint64_t width_words, void *stream);

/*
 * One stripe on several GPUs (BASELINE configs[3]; the reference has no multi-device code — SURVEY.md §8e).
 * The word columns of a stripe are independent transforms, so GPU g of G takes words [g*S/G, (g+1)*S/G) of EVERY
 * block (S = block_bytes/4): a "column slab" of k blocks x block_bytes/G bytes, encoded by an ordinary per-device
 * context with no communication; the only exchange is moving slabs (xGMI peer copies, or each GPU's own host link).
 *
 *   fastecc_create_sharded : gpu_ids[0] is the ROOT device.  block_bytes must divide by 4*n_gpus (16*n_gpus for the
 *       64-bit field).  Ids may repeat (several slabs on one device; used by the single-GPU tests).  Every device
 *       must be able to access the root's memory (hipDeviceCanAccessPeer), else FASTECC_E_UNSUPPORTED.  All (n,k)
 *       of fastecc_create are accepted.  One host process drives all devices; for one-process-per-GPU hosts see
 *       fastecc_encode_columns and fastecc_amd/sharding.py (RCCL gather).
 *   fastecc_encode on such a context, same arguments as ever:
 *       FASTECC_MEM_DEVICE       data / parity are full stripes in the ROOT device's memory.  Slab g is pulled by GPU g
 *                                with a strided peer copy, encoded, and its parity pushed back into `parity`; enqueued
 *                                on internal streams, the call behaves as one operation on `stream` (a stream of the
 *                                root device) and does not synchronise.
 *       FASTECC_MEM_HOST_PINNED  full stripes in pinned host memory: every GPU moves its own slab over its own host
 *                                link (strided copies), so G links work in parallel and no xGMI traffic exists.
 *       FASTECC_MEM_HOST         the same for pageable memory, synchronous: every slab has a host thread that moves its columns through that
 *                                slab's rings of pinned slots (128 MiB of pinned memory per GPU), all slabs side by side.
 *   fastecc_encode_sharded : the data is ALREADY sharded — data_slabs[g] is slab g ([k][block_bytes/G], contiguous)
 *       in GPU g's memory.  parity_slabs (optional): G device pointers, slab g of the parity stays on GPU g.
 *       parity (optional): full parity stripe in the ROOT's memory, gathered over xGMI ("a final gather").  At least
 *       one of the two; with parity_slabs == NULL the library uses its own slab buffers.
 *   Both pipeline in column sub-slabs ("sub_slabs" option, default 2; 64 words at 4 KB blocks on 8 GPUs): the copy of
 *   one sub-slab runs behind the kernels of the next.  "gather_mode": 1 = copy engines (hipMemcpy2DAsync, default),
 *   2 = a copy kernel on the sending GPU storing straight into the root's memory.
 *   fastecc_decode_prepare / fastecc_decode / fastecc_repair work on a sharded context as well: a lost block is lost in every
 *   slab, so each slab repairs its own columns with the same pattern; full stripes (root or host memory) are moved slab-wise
 *   like for the encode.
 * Other entry points on a sharded context: destroy, set_option (the two above; anything else is forwarded to every
 * device context), set_plan, plan_string, profile_* (device 0's kernels); the rest return FASTECC_E_UNSUPPORTED.
 */
int fastecc_create_sharded(fastecc_ctx **out, uint64_t n, uint64_t k, uint64_t block_bytes, int field, const int *gpu_ids,
                           int n_gpus);
int fastecc_encode_sharded(fastecc_ctx *ctx, const void *const *data_slabs, void *const *parity_slabs, void *parity, void *stream);
/*
 * The block-distributed form of the same encode — the one that scales.  A gather to one root pushes (G-1)/G of the stripe through ONE
 * GPU's links; here the result stays distributed the way RS.md:13-33 thinks of a codeword (N data and M parity blocks, separately stored
 * units): GPU g ends with parity blocks [g*M/G, (g+1)*M/G) WHOLE in parity_blocks[g] ([M/G][block_bytes], device memory of GPU g).  The
 * exchange is an all-to-all — every GPU sends rows [d*M/G, (d+1)*M/G) of its parity slab into its columns of GPU d's blocks — so each
 * xGMI link carries 1/G^2 of the stripe per direction and no GPU is a hot spot.
 *   data_layout FASTECC_SHARD_SLABS  : data[g] = column slab g on GPU g, as for fastecc_encode_sharded;
 *               FASTECC_SHARD_BLOCKS : the data is block-distributed too — data[g] = data blocks [g*k/G, (g+1)*k/G) whole on GPU g
 *                                      ([k/G][block_bytes]); the mirror all-to-all (whole blocks -> column slabs) runs in front of the
 *                                      encode, sub-slab h+1 travelling while sub-slab h is in the kernels.
 * Needs n - k (and, for FASTECC_SHARD_BLOCKS, k) divisible by the number of GPUs (FASTECC_E_INVAL otherwise) and peer access between every
 * pair of the context's devices (FASTECC_E_UNSUPPORTED otherwise; checked and enabled at the first call).  parity_blocks must not overlap
 * data.  "sub_slabs" and "gather_mode" apply: 2 = one kernel per GPU and sub-slab storing through all peer mappings at once, 1 = G pitched
 * copies on the copy engines, one stream per destination.  Enqueued on internal streams; the call behaves as one operation on `stream`
 * (a stream of the root device) and does not synchronise.
 */
#define FASTECC_SHARD_SLABS 0
#define FASTECC_SHARD_BLOCKS 1
int fastecc_encode_sharded_blocks(fastecc_ctx *ctx, const void *const *data, int data_layout, void *const *parity_blocks, void *stream);
/* Geometry of a sharded context: slabs (= n_gpus given at creation), bytes of a block that one slab holds, and the
 * device of slab g (any pointer may be NULL).  FASTECC_E_INVAL on an ordinary context. */
int fastecc_shard_info(const fastecc_ctx *ctx, int *n_slabs, uint64_t *slab_block_bytes, int *devices, int cap);

/*
 * Many stripes at once: `count` stripes of k blocks stored back to back in DEVICE memory (stripe b at data +
 * b*k*block_bytes, its parity at parity + b*k*block_bytes), encoded by ONE launch per pass.  Same result as `count`
 * fastecc_encode calls; meant for small codes — a (256,128) x 4 KB stripe is 1 MiB and a single launch of it is
 * launch-bound, a batch of them runs at the rate of the large stripes.  n = 2k = 2^m over GF(0xFFF00001) only;
 * parity == data encodes in place; enqueued on `stream` without synchronising.
 */
int fastecc_encode_batch(fastecc_ctx *ctx, const void *data, void *parity, uint64_t count, void *stream);

/*
 * The reference's own calling form: an array of k HOST block pointers, transformed in place
 * (T** data of RS.cpp:31-33 / ntt.cpp:348-350).  On return blocks[j] holds parity block j (the
 * pointer array itself is left untouched, which is also what two MFA_NTT calls leave behind,
 * SURVEY.md §8 a1).  Synchronous.  The blocks travel through rings of pinned slots that helper threads fill from / empty into
 * the caller's blocks while the copy engine moves the previous slot: k = 2^19 blocks of 4 KB (2 GiB up, 2 GiB down) in 88-110 ms =
 * 39-49 GB/s of data + parity; a table that points into one buffer, block after block (RS.cpp's own), goes up as one copy: 82-88 ms
 * (profiles/r04/encode_blocks_bench.jsonl).
 */
int fastecc_encode_blocks(fastecc_ctx *ctx, void *const *blocks);

/*
 * Length-k number-theoretic transform of the stripe, natural order in and out, unscaled, forward
 * root w_k (inverse != 0: w_k^-1).  Replaces MFA_NTT<T,P>(data,N,SIZE,InvNTT) (ntt.cpp:382-447) and
 * Rec_NTT (ntt.cpp:349-378) followed by reading the blocks through the permuted pointer array.
 * In place on `data` (k*block_bytes bytes).
 */
int fastecc_ntt(fastecc_ctx *ctx, void *data, int inverse, int mem_kind, void *stream);

/*
 * block i *= scale * base^i for i in [0,k): the per-block twiddle multiply of RS.cpp:51-59
 * (scale = 1/N, base = root(2N)) and, with other arguments, the MFA twiddle of ntt.cpp:421-431.
 * In place on device or host memory as for fastecc_encode.
 */
int fastecc_scale_blocks(fastecc_ctx *ctx, void *data, uint32_t scale, uint32_t base, int mem_kind, void *stream);

/*
 * Element-wise field kernels over `count` words of DEVICE memory (parity tests of the device
 * GF_Mul/GF_Add/GF_Sub against GF(p).cpp:37-48,110-127).  op: 0 = add, 1 = sub, 2 = mul.
 */
int fastecc_gf_binary(fastecc_ctx *ctx, int op, const uint32_t *x, const uint32_t *y, uint32_t *out, uint64_t count,
                      void *stream);

/*
 * Count the words of a stripe that are not field elements (>= p).  The reference documents "all words < p"
 * as a precondition (README.md:160-162) and never checks it; a host can call this before encoding
 * untrusted data.  Synchronises `stream`.  *bad_words == 0 means the stripe is encodable.
 */
int fastecc_check_range(fastecc_ctx *ctx, const void *data, int mem_kind, void *stream, uint64_t *bad_words);

/*
 * Erasure decoding: recover the erased DATA blocks from any k or more surviving blocks of the codeword.  Both fields:
 * GF(0xFFF00001) as described below; GF((2^61-1)^2) for its (2k,k) codes with the same scheme on 16-byte elements
 * (gf61_decode.hip).  The reference describes the algorithm (README.md:102-119 "Fastest", RS.md:42-79: erasure locator l,
 * p = f*l known everywhere, f(e) = p'(e) / l'(e)) and does not implement it; the data-parallel part here is one
 * transform pipeline of size 2N (the encoder's kernels, N = 2^ceil(log2 k)) between a gather and a scale pass — for
 * the codes with n <= 2k and k >= 2^17 as two pipelines of size k, the data half and the few parity blocks a pattern needs
 * (option "decode_split").
 * Works for every GF(0xFFF00001) code fastecc_create accepts: a code is f on a subset of the (N << e)-th roots of unity
 * (e = 1, or 2 / 3 for n = 4k / 8k); positions that hold none of its blocks count as erased, zero-extended data blocks
 * as known, so exactly n - k losses are tolerated.  Mixed-radix codes (fastecc_create_ex) are decoded the same way on the
 * (2 q 2^m)-th roots of unity with mixed-radix transforms, for orders q 2^m <= 2^20.
 *   fastecc_decode_prepare : set the erasure pattern, k data flags and n - k parity flags (non-zero = block survives).
 *                            The host classifies the positions; the locator's product tree, its two size-NC transforms
 *                            and the inversions run on the device (2.2-2.4 ms at (2^20,2^19); the first call also builds the
 *                            decoder's contexts).  Synchronous.  FASTECC_E_INVAL if fewer than k blocks survive.
 *                            Reusable for any number of stripes.
 *   fastecc_decode         : data (k blocks; the erased ones are overwritten with the recovered content, the others
 *                            are not written) and parity (n - k blocks, read only; content of erased blocks is ignored).
 *                            DEVICE pointers: enqueued on `stream`, no synchronisation.  HOST / HOST_PINNED: staged, synchronous (on sharded contexts
 *                            too) — the data
 *                            stripe and the parity blocks the decoder reads travel up, the rebuilt blocks back (whole stripes
 *                            when more than an eighth of the codeword is lost).
 * fastecc_decode leaves erased parity blocks alone; fastecc_repair rebuilds them too.
 * Patterns with at most 256 lost blocks (option "decode_direct_max", 0..256, default 256; 32 for GF((2^61-1)^2)) take a direct path: every
 * lost block is a fixed linear combination of surviving ones, so prepare builds weight tables (0.2-3.5 ms, no transform contexts) and decode
 * is one read of the data plus a few parity blocks — 0.4 ms for up to 16 lost blocks of a 2 GiB stripe, 0.7 ms for 64, 2.6 ms for 256 (matrix
 * cores; option "direct_kernel") against 3.8-5.7 ms on the transform path (repair: the lost parity blocks in the same pass when at most 32 blocks are lost in all, else in a second read); every GF(0xFFF00001)
 * code, and the (2k,k) codes of GF((2^61-1)^2); identical results.  n = 4k / 8k over GF((2^61-1)^2): the data and the first coset are a (2k,k) code,
 * and up to 32 losses among THOSE 2k blocks (lost blocks of the other cosets do not count; repair re-encodes them) take that code's direct path.
 */
int fastecc_decode_prepare(fastecc_ctx *ctx, const uint8_t *data_present, const uint8_t *parity_present);
int fastecc_decode(fastecc_ctx *ctx, void *data, const void *parity, int mem_kind, void *stream);
/* fastecc_decode, then the erased PARITY blocks as well: the repaired data is encoded once more and the lost parity blocks
 * (only those) are written into `parity` — the whole codeword is whole again ("repair").  Same arguments otherwise. */
int fastecc_repair(fastecc_ctx *ctx, void *data, void *parity, int mem_kind, void *stream);

/*
 * Data packing (GF.md:72-104 "Efficient data packing", README.md:160-163): RS.cpp only encodes words < p, so
 * arbitrary bytes are first recoded with one extra word per block — 4096-byte sectors become the 4100-byte
 * blocks the encoder then works on.  The reference describes this in prose and has NO code for it; the exact
 * format is therefore defined here (nothing upstream pins it):
 *   a word is (digit << 20) | low20, digit = its top 12 bits.  Only words with digit 0xFFF can be >= p.
 *   raw block   : W = block_bytes/4 - 1 arbitrary uint32 words (1 <= W <= 1024)
 *   packed block: W + 1 words, all < 0xFFF00000 < p.  Word j keeps low20 of raw word j; the digit string is
 *     flag word (word W) = 0 : no raw digit is 0xFFF, digits unchanged;
 *     flag word          = 1 : [one 11-bit entry per 0xFFF digit, in increasing position: position | 0x400 if
 *                               another entry follows][all other digits in their original order].
 * The context must be GF(0xFFF00001) with block_bytes = 4 * (W + 1) (e.g. 4100); raw stripes are k * 4W bytes,
 * packed stripes k * block_bytes bytes, both block-major and contiguous — except that DEVICE packed stripes use
 * the context's "row_pitch_words" option like fastecc_encode does (e.g. 1056: rows padded to 4224 bytes, which
 * is what makes 4100-byte blocks encode at full speed).  DEVICE pointers: enqueued on `stream`.
 * fastecc_unpack_blocks also validates: a block no packer produces (flag > 1, a 0xFFF digit left, entries not
 * strictly increasing / out of range / never ending) is copied through unchanged and counted in *bad_blocks
 * (may be NULL = not wanted; non-NULL makes the call synchronise `stream`).
 */
int fastecc_pack_blocks(fastecc_ctx *ctx, const void *raw, void *packed, int mem_kind, void *stream);
int fastecc_unpack_blocks(fastecc_ctx *ctx, const void *packed, void *raw, int mem_kind, void *stream, uint64_t *bad_blocks);

/* Host-side field helpers (GF(p).cpp:254-297), used to build tables and by bindings. */
uint32_t fastecc_gf_mul(uint32_t x, uint32_t y);
uint32_t fastecc_gf_pow(uint32_t x, uint32_t e);
uint32_t fastecc_gf_root(uint32_t order); /* 19^((p-1)/order); order must divide 2^20 */
uint32_t fastecc_gf_inv(uint32_t x);
/* The same for FASTECC_FIELD_GF_P61_SQUARED: z[0] = re, z[1] = im (inputs are reduced mod p); out may alias.
 * fastecc_gf61_root returns FASTECC_E_INVAL unless order is a power of two <= 2^62. */
int fastecc_gf61_mul(const uint64_t x[2], const uint64_t y[2], uint64_t out[2]);
int fastecc_gf61_pow(const uint64_t x[2], uint64_t e, uint64_t out[2]);
int fastecc_gf61_inv(const uint64_t x[2], uint64_t out[2]);
int fastecc_gf61_root(uint64_t order, uint64_t out[2]);

/*
 * Introspection for the benchmark: time the last `fastecc_encode` enqueued on DEVICE memory spent in
 * kernels, measured with HIP events on the stream it ran on.  Enable before the encode calls.
 *   fastecc_profile_enable(ctx, 1)  -> every kernel launch is bracketed by hipEvents
 *   fastecc_profile_read(ctx, names, ms, launches, cap) -> per-kernel accumulated ms and launch counts
 *                                      since the last reset (synchronises the stream); returns the
 *                                      number of distinct kernels written (<= cap)
 */
int fastecc_profile_enable(fastecc_ctx *ctx, int on);
int fastecc_profile_read(fastecc_ctx *ctx, const char **names, double *ms, uint64_t *launches, int cap);
/* As above, plus the accumulated ALGORITHMIC bytes of those launches (each launch reads its part of the stripe
 * once and writes it once): bytes[i] / launches[i] is the per-launch figure the roofline uses. */
int fastecc_profile_read_bytes(fastecc_ctx *ctx, const char **names, double *ms, uint64_t *launches, uint64_t *bytes, int cap);
int fastecc_profile_reset(fastecc_ctx *ctx);

/* Plan description, e.g. "dif5,dif5,...|mid...|dit..." — for logs and DESIGN.md tables. */
const char *fastecc_plan_string(fastecc_ctx *ctx);
/*
 * Tuning options (all bit-exact; defaults are the measured best on MI355X).
 *   "cache_policy" 0..15: bit 0/1 non-temporal stripe loads/stores in the outer passes, bit 2/3 in MID (default 15);
 *   "xcd_swizzle"  0..2 : tile order per XCD (default 1);
 *   "row_pitch_words" = L >= block_bytes/4 (0 = contiguous): DEVICE stripes given to fastecc_encode are [k][L]
 *                  words, the first block_bytes/4 of each row valid, the rest untouched.  Lets a host that owns its
 *                  HBM layout pad odd block sizes (2052, 4100 bytes) to a multiple of 128 bytes: +50 % throughput.
 *                  fastecc_pack_blocks / _unpack_blocks follow the pitch on their packed side; the other entry
 *                  points return FASTECC_E_UNSUPPORTED while a pitch is set;
 *   "host_slabs" = 1 .. 32, a power of two (default 8): column slabs of the FASTECC_MEM_HOST_PINNED pipeline (and of the FASTECC_MEM_HOST one);
 *   "stage_threads" = 0 .. 64 (default 0 = min(6, hardware threads / 4)): helper threads per staging ring (pageable host memory: FASTECC_MEM_HOST
 *                  results, fastecc_encode_blocks); 4 to 12 measured the same on a 16-CPU share (profiles/r04/encode_blocks_bench.jsonl);
 *   "host_pipeline" = 0 / 1 (default 0): 1 = FASTECC_MEM_HOST encodes of (2k,k) stripes of 256 MiB and more move pageable memory through two
 *                  rings of pinned slots (64 MiB each, allocated at the first such call) served by helper threads, slab h going up while
 *                  slab h - 1 comes down; 0 = upload, encode, download one after the other (the download still through its ring).  The
 *                  pipeline is bounded by what the host's cores copy, not by the link: 65-100 ms against 81-97 (by box) for 2 + 2 GiB in
 *                  a 16-CPU share of an EPYC 9575F (profiles/r04/host_pageable_pipeline.jsonl) — worth it on a host with cores to spare;
 *   "encode_direct_max" = 0..256 (default 160): codes with at most this many parity blocks (n - k) are encoded straight from the Lagrange
 *                  basis — one read of the data (0.4 ms up to 16 parity blocks ... 1.4 ms for 128 at k = 2^19 x 4 KB) instead of the transform
 *                  pipeline (2.4 ms); same parity bits.  Rows the matrix-core kernel cannot take (odd length, < 64 words, not 8-byte aligned)
 *                  stop at 32;
 *   "decode_direct_max" = 0..256 (default 256; 0..32 for GF((2^61-1)^2)): lost blocks up to which the decoder's direct path is used (next
 *                  decode_prepare); rows the matrix-core kernel cannot take stop at 96 (not for mixed-radix orders above 2^20, whose transform
 *                  path is much dearer: there the option alone decides);
 *   "decode_split" = 0 / 1 / 2 (default 1; codes over GF(0xFFF00001) with n <= 2k and k >= 2^17 (power-of-two orders), next decode_prepare): the
 *                  decoder's 2k-point transform as two transforms of k points — the data half, and of the parity half only the blocks needed: the
 *                  survivors at multiples of 2^h of that half (largest h <= 5 that leaves as many as there are lost data blocks), whose transform is
 *                  one of k >> h rows (2 % of the codeword lost: decode 7.1 -> 3.8 ms at k = 2^19 x 4 KB), else the surviving blocks of the first
 *                  few block groups (what round 3 always did: 2 = that form only); 0 = one transform of 2k points.  Same bits.
 *                  GF((2^61-1)^2), k >= 2^11: 0 / 1 (default 1) — the same split in its k >> h form (h = 5 .. 1): decode 7.2 -> 4.5 ms, repair 9.2 -> 6.8 ms (a second MID + DIT chain for the lost parity blocks) at
 *                  k = 2^19 x 4 KB and 2 % lost, 72 ms = 1.14 x the encode at 64 KB blocks; patterns it does not take run the folded 2k-point transform;
 *   "direct_kernel" = 0 / 1 / 2 (default 0 = choose): the kernel of those direct paths — 1 = VALU (96-bit lazy accumulation, any rows),
 *                  2 = MFMA (i8 digits; falls back to 1 where it cannot run).  Same bits either way;
 *   "fuse_radix" = 0 / 1 (default 1; mixed-radix contexts): the odd-radix level fused into the outer tile passes, or as its own passes;
 *   "slabs" = H (1..32): encode H column slabs of the stripe on internal streams, each one pass
 * behind the previous, so that different kinds of passes overlap on the GPU, or with "slab_mode" = 1 one
 * after the other on `stream` (a memory-side-cache experiment: profiles/r02/slab_cache_sweep.md — neither pays).  The call
 * still behaves as one operation on `stream`: it starts after prior work on `stream` and later work on `stream` waits for it.
 */
int fastecc_set_option(fastecc_ctx *ctx, const char *name, int value);

/* Select the kernel plan (0 = default).  Exposed so bench.py can A/B plans; see DESIGN.md §5.  3100 is the split of rounds 1-4 (dif9 / mid10 / dit9
 * at k = 2^19); the default gives MID one level fewer for the power-of-two codes from k = 2^16 up (3090: dif10 / mid9 / dit10; 4090 at 2^18:
 * outer tiles of 64-word rows).  Every plan computes the same words. */
int fastecc_set_plan(fastecc_ctx *ctx, int plan);

/*
 * Host-only introspection (no HIP device is touched): the pass plan that (k, block_bytes, plan) selects,
 * and the level-packed twiddle table that goes with it — what replaces the roots[] array of
 * ntt.cpp:397-402.  which: 0 = encode/interpolate (inverse roots), 1 = encode/evaluate (forward roots),
 * 2 = fastecc_ntt forward, 3 = fastecc_ntt inverse.  `out` receives k words (Montgomery form, w*2^32 mod p;
 * level l at [2^l, 2^(l+1))); level_stride (optional, log2(k) ints) the log2 stride of the register run
 * that executes each level.
 */
int fastecc_plan_describe(uint64_t k, uint64_t block_bytes, int plan, char *buf, size_t cap);
int fastecc_plan_twiddles(uint64_t k, uint64_t block_bytes, int plan, int which, uint32_t *out, int32_t *level_stride);

#ifdef __cplusplus
}
#endif
#endif /* FASTECC_H */
