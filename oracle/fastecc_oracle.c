/*
 * fastecc_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE ONLY; see fastecc_oracle.h).
 *
 * A from-scratch plain-C restatement of what the FastECC reference computes on the
 * encode path, written from the reference's documented maths, with the file:line of
 * the reference code each routine follows.  It is deliberately simple (radix-2 only,
 * no cache blocking): it is the checker, never the thing measured or shipped.
 *
 * Parity status: PINNED (see header).  Build: oracle/Makefile -> oracle/libfastecc_oracle.so
 */
#include "fastecc_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define P ORC_P

/* ------------------------------------------------------------------------------------------
 * GF(P) arithmetic.  All results are the canonical representative in [0,P).
 * ---------------------------------------------------------------------------------------- */

/* GF(p).cpp:37-42: subtract, add P back when the unsigned subtraction wrapped. */
uint32_t orc_gf_sub(uint32_t x, uint32_t y)
{
    uint32_t d = x - y;
    if (d > x) d += P;
    return d;
}

/* GF(p).cpp:44-48: x + y == x - (P - y). */
uint32_t orc_gf_add(uint32_t x, uint32_t y)
{
    return orc_gf_sub(x, P - y);
}

/* GF(p).cpp:99-104: the textbook form, product in 64 bits then %. */
uint32_t orc_gf_mul_wide(uint32_t x, uint32_t y)
{
    return (uint32_t)(((uint64_t)x * y) % P);
}

/* GF(p).cpp:110-127: Barrett reduction with a 32-bit reciprocal.
 * recip = floor(2^64 / P) - 2^32 = 0x001000FF; the quotient estimate is
 * ((t + hi(t)*recip) >> 32) and undershoots by at most one, fixed by one conditional subtract. */
uint32_t orc_gf_mul(uint32_t x, uint32_t y)
{
    const uint64_t recip = 0x001000FFu;
    uint64_t t = (uint64_t)x * y;
    uint64_t q = (t + (t >> 32) * recip) >> 32;
    t -= q * P;
    if (t >= P) t -= P;
    return (uint32_t)t;
}

/* GF(p).cpp:254-264: right-to-left square and multiply. */
uint32_t orc_gf_pow(uint32_t x, uint32_t n)
{
    uint32_t acc = 1;
    while (n) {
        if (n & 1u) acc = orc_gf_mul(acc, x);
        x = orc_gf_mul(x, x);
        n >>= 1;
    }
    return acc;
}

/* GF(p).cpp:268-276: 19 generates the multiplicative group; an order-`order` root is 19^((P-1)/order). */
uint32_t orc_gf_root(uint32_t order)
{
    return orc_gf_pow(19u, (P - 1u) / order);
}

/* GF(p).cpp:293-297: Fermat inverse x^(P-2). */
uint32_t orc_gf_inv(uint32_t x)
{
    return orc_gf_pow(x, P - 2u);
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------
 * Transforms.  data is block-major: block b occupies words [b*size, (b+1)*size).  The transform
 * runs down the block index; the `size` word columns are independent (ntt.cpp:348-350).
 * ---------------------------------------------------------------------------------------- */

/* ntt.cpp:451-483: X_j = sum_i x_i * w^(i*j), w = root(N) or its inverse; no 1/N scaling. */
void orc_slow_ntt(uint32_t *data, size_t N, size_t size, int inverse)
{
    uint32_t w = orc_gf_root((uint32_t)N);
    if (inverse) w = orc_gf_inv(w);
    uint32_t *out = (uint32_t *)malloc(N * size * sizeof(uint32_t));
    uint32_t wj = 1; /* w^j */
    for (size_t j = 0; j < N; j++) {
#pragma omp parallel for
        for (ptrdiff_t k = 0; k < (ptrdiff_t)size; k++) {
            uint32_t acc = 0, wij = 1; /* w^(i*j) */
            for (size_t i = 0; i < N; i++) {
                acc = orc_gf_add(acc, orc_gf_mul(wij, data[i * size + k]));
                wij = orc_gf_mul(wij, wj);
            }
            out[j * size + k] = acc;
        }
        wj = orc_gf_mul(wj, w);
    }
    memcpy(data, out, N * size * sizeof(uint32_t));
    free(out);
}

static size_t bit_reverse(size_t v, int bits)
{
    size_t r = 0;
    for (int b = 0; b < bits; b++) {
        r = (r << 1) | (v & 1u);
        v >>= 1;
    }
    return r;
}

static int ilog2(size_t n)
{
    int l = 0;
    while (((size_t)1 << l) < n) l++;
    return l;
}

/* ntt.cpp:292-318 + 251-284: the reference bit-reverses its array of block POINTERS and then runs
 * log2(N) decimation-in-time levels; level with half-size h combines blocks (x+i, x+i+h) using the
 * twiddle (root of order 2h)^i.  Here an index table `where[]` plays the pointer array's role, and at
 * the end blocks are moved so that logical block j is physically block j again. */
void orc_ntt(uint32_t *data, size_t N, size_t size, int inverse)
{
    if (N < 2) return;
    const int n = ilog2(N);
    size_t *where = (size_t *)malloc(N * sizeof(size_t));
    for (size_t j = 0; j < N; j++) where[j] = bit_reverse(j, n);

    uint32_t wN = orc_gf_root((uint32_t)N);
    if (inverse) wN = orc_gf_inv(wN);

    for (size_t h = 1; h < N; h *= 2) {
        /* root of order 2h = wN^(N/(2h)) */
        uint32_t w2h = orc_gf_pow(wN, (uint32_t)(N / (2 * h)));
#pragma omp parallel for schedule(static)
        for (ptrdiff_t x = 0; x < (ptrdiff_t)N; x += 2 * (ptrdiff_t)h) {
            uint32_t tw = 1;
            for (size_t i = 0; i < h; i++) {
                uint32_t *lo = data + where[x + i] * size;
                uint32_t *hi = data + where[x + i + h] * size;
                for (size_t k = 0; k < size; k++) {
                    uint32_t u = lo[k];
                    uint32_t v = orc_gf_mul(hi[k], tw);
                    lo[k] = orc_gf_add(u, v);
                    hi[k] = orc_gf_sub(u, v);
                }
                tw = orc_gf_mul(tw, w2h);
            }
        }
    }

    /* put logical block j at physical position j */
    uint32_t *tmp = (uint32_t *)malloc(N * size * sizeof(uint32_t));
    for (size_t j = 0; j < N; j++) memcpy(tmp + j * size, data + where[j] * size, size * sizeof(uint32_t));
    memcpy(data, tmp, N * size * sizeof(uint32_t));
    free(tmp);
    free(where);
}

/* One length-N column held contiguously; same radix-2 DIT recurrence as orc_ntt. */
static void column_ntt(uint32_t *col, size_t N, int n, const uint32_t *stage_root /* [n]: root of order 2^(l+1) */)
{
    for (size_t j = 0; j < N; j++) {
        size_t r = bit_reverse(j, n);
        if (r > j) {
            uint32_t t = col[j];
            col[j] = col[r];
            col[r] = t;
        }
    }
    int level = 0;
    for (size_t h = 1; h < N; h *= 2, level++) {
        const uint32_t w2h = stage_root[level];
        for (size_t x = 0; x < N; x += 2 * h) {
            uint32_t tw = 1;
            for (size_t i = 0; i < h; i++) {
                uint32_t u = col[x + i];
                uint32_t v = orc_gf_mul(col[x + i + h], tw);
                col[x + i] = orc_gf_add(u, v);
                col[x + i + h] = orc_gf_sub(u, v);
                tw = orc_gf_mul(tw, w2h);
            }
        }
    }
}

void orc_ntt_fast(uint32_t *data, size_t N, size_t size, int inverse)
{
    if (N < 2) return;
    const int n = ilog2(N);
    uint32_t wN = orc_gf_root((uint32_t)N);
    if (inverse) wN = orc_gf_inv(wN);
    uint32_t stage_root[32];
    for (int l = 0; l < n; l++) stage_root[l] = orc_gf_pow(wN, (uint32_t)(N >> (l + 1)));

#pragma omp parallel
    {
        uint32_t *col = (uint32_t *)malloc(N * sizeof(uint32_t));
#pragma omp for schedule(dynamic, 4)
        for (ptrdiff_t k = 0; k < (ptrdiff_t)size; k++) {
            for (size_t i = 0; i < N; i++) col[i] = data[i * size + k];
            column_ntt(col, N, n, stage_root);
            for (size_t i = 0; i < N; i++) data[i * size + k] = col[i];
        }
        free(col);
    }
}

/* RS.cpp:51-59 generalised: block i is multiplied by scale * base^i. */
void orc_scale_blocks(uint32_t *data, size_t N, size_t size, uint32_t scale, uint32_t base)
{
#pragma omp parallel for schedule(static)
    for (ptrdiff_t i = 0; i < (ptrdiff_t)N; i++) {
        uint32_t f = orc_gf_mul(scale, orc_gf_pow(base, (uint32_t)i));
        uint32_t *blk = data + (size_t)i * size;
        for (size_t k = 0; k < size; k++) blk[k] = orc_gf_mul(blk[k], f);
    }
}

/* RS.cpp:40-63: unscaled inverse transform (interpolate), multiply coefficient i by
 * root(2N)^i / N, forward transform (evaluate at the odd powers of root(2N)). */
static void encode_with(void (*ntt)(uint32_t *, size_t, size_t, int), uint32_t *data, size_t N, size_t size)
{
    ntt(data, N, size, 1);
    orc_scale_blocks(data, N, size, orc_gf_inv((uint32_t)N), orc_gf_root((uint32_t)(2 * N)));
    ntt(data, N, size, 0);
}

void orc_encode(uint32_t *data, size_t N, size_t size) { encode_with(orc_ntt, data, N, size); }
void orc_encode_fast(uint32_t *data, size_t N, size_t size) { encode_with(orc_ntt_fast, data, N, size); }

/* The same composition by the O(N^2) definition (ntt.cpp:451-483): valid for ANY order N | p-1, which is what pins the
 * mixed-radix codes (N = q 2^m, q odd) — the reference's Slow_NTT is the only transform of it that accepts such N. */
void orc_encode_slow(uint32_t *data, size_t N, size_t size) { encode_with(orc_slow_ntt, data, N, size); }

/* Transform of order N = q * M, M = 2^m, q odd (NTT.md:43-46), natural order in and out: Cooley-Tukey with the odd factor
 * outermost, X[q j2 + j1] = sum_i2 w_M^(i2 j2) [ w_N^(i2 j1) sum_i1 x[i1 M + i2] w_q^(i1 j1) ].  The inner odd-order sums
 * are taken by definition (for q = 3, 9 the reference has the codelets NTT3 / NTT9, ntt.cpp:25-44, 113-146: same values,
 * tests/test_oracle.py checks them against each other through oracle/_ref). */
void orc_ntt_mixed(uint32_t *data, size_t N, size_t size, int inverse)
{
    size_t M = 1;
    while ((N % (2 * M)) == 0) M *= 2;
    const size_t q = N / M;
    if (q == 1) {
        orc_ntt_fast(data, N, size, inverse);
        return;
    }
    const int m = ilog2(M);
    uint32_t wN = orc_gf_root((uint32_t)N);
    if (inverse) wN = orc_gf_inv(wN);
    const uint32_t wq = orc_gf_pow(wN, (uint32_t)M), wM = orc_gf_pow(wN, (uint32_t)q);
    uint32_t stage_root[32];
    for (int l = 0; l < m; l++) stage_root[l] = orc_gf_pow(wM, (uint32_t)(M >> (l + 1)));
#pragma omp parallel
    {
        uint32_t *col = (uint32_t *)malloc(N * sizeof(uint32_t)), *y = (uint32_t *)malloc(N * sizeof(uint32_t));
#pragma omp for schedule(dynamic, 4)
        for (ptrdiff_t k = 0; k < (ptrdiff_t)size; k++) {
            for (size_t i = 0; i < N; i++) col[i] = data[i * size + k];
            for (size_t i2 = 0; i2 < M; i2++) {
                const uint32_t t = orc_gf_pow(wN, (uint32_t)i2);
                uint32_t tj = 1; /* w_N^(i2 j1) */
                for (size_t j1 = 0; j1 < q; j1++) {
                    const uint32_t wj = orc_gf_pow(wq, (uint32_t)j1);
                    uint32_t acc = 0, wij = 1; /* w_q^(i1 j1) */
                    for (size_t i1 = 0; i1 < q; i1++) {
                        acc = orc_gf_add(acc, orc_gf_mul(wij, col[i1 * M + i2]));
                        wij = orc_gf_mul(wij, wj);
                    }
                    y[j1 * M + i2] = orc_gf_mul(acc, tj);
                    tj = orc_gf_mul(tj, t);
                }
            }
            for (size_t j1 = 0; j1 < q; j1++) {
                if (M > 1) column_ntt(y + j1 * M, M, m, stage_root);
                for (size_t j2 = 0; j2 < M; j2++) data[(q * j2 + j1) * size + k] = y[j1 * M + j2];
            }
        }
        free(col);
        free(y);
    }
}

void orc_encode_mixed(uint32_t *data, size_t N, size_t size) { encode_with(orc_ntt_mixed, data, N, size); }

/* SURVEY.md §0.6 (derived from RS.cpp:40-63, ntt.cpp:450-483): with f the degree<N polynomial
 * through f(w_N^m) = data[m], parity[j] = f(w_2N^(2j+1)).  Evaluated directly via Lagrange-free
 * route: coefficients c = (1/N) * sum_m data[m] w_N^(-m i), then Horner-free O(N^2) evaluation. */
void orc_encode_by_definition(const uint32_t *data, uint32_t *parity, size_t N, size_t size)
{
    const uint32_t wN = orc_gf_root((uint32_t)N), w2N = orc_gf_root((uint32_t)(2 * N));
    const uint32_t wNinv = orc_gf_inv(wN), invN = orc_gf_inv((uint32_t)N);
    uint32_t *coef = (uint32_t *)malloc(N * sizeof(uint32_t));
    for (size_t k = 0; k < size; k++) {
        for (size_t i = 0; i < N; i++) {
            uint32_t acc = 0;
            const uint32_t wi = orc_gf_pow(wNinv, (uint32_t)i);
            uint32_t wim = 1;
            for (size_t m = 0; m < N; m++) {
                acc = orc_gf_add(acc, orc_gf_mul(data[m * size + k], wim));
                wim = orc_gf_mul(wim, wi);
            }
            coef[i] = orc_gf_mul(acc, invN);
        }
        for (size_t j = 0; j < N; j++) {
            const uint32_t pt = orc_gf_pow(w2N, (uint32_t)(2 * j + 1));
            uint32_t acc = 0, pw = 1;
            for (size_t i = 0; i < N; i++) {
                acc = orc_gf_add(acc, orc_gf_mul(coef[i], pw));
                pw = orc_gf_mul(pw, pt);
            }
            parity[j * size + k] = acc;
        }
    }
    free(coef);
}

/* main.cpp:202-212: h = (h + word) * 123456791 + (h >> 17), seeded with 314159253, 32-bit wrap. */
uint32_t orc_hash(const uint32_t *data, size_t nwords)
{
    uint32_t h = 314159253u;
    for (size_t i = 0; i < nwords; i++) h = (h + data[i]) * 123456791u + (h >> 17);
    return h;
}

/* RS.cpp:28-29 / main.cpp:249-250 */
void orc_fill_linear(uint32_t *data, size_t nwords)
{
    for (size_t i = 0; i < nwords; i++) data[i] = (uint32_t)(i % P);
}

/* SURVEY.md Appendix B "rand": splitmix64 stream reduced mod P, filled in increasing index. */
void orc_fill_splitmix(uint32_t *data, size_t nwords, uint64_t seed)
{
    uint64_t s = seed;
    for (size_t i = 0; i < nwords; i++) {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        data[i] = (uint32_t)(z % P);
    }
}

/* ------------------------------------------------------------------------------------------
 * Data packing, GF.md:72-104 ("Efficient data packing"): arbitrary 32-bit words -> words < P at the cost of
 * one extra word per block.  The reference has prose only (no code), so the exact format is OURS (stated in
 * include/fastecc.h) and PARITY IS UNPINNED; this is the plain sequential statement of it.
 *
 * A word is (digit << 20) | low20.  Words with digit 0xFFF are the ones that may be >= P = 0xFFF00001, so the
 * block's `words` digits (base 4096) are recoded so that none equals 0xFFF; the low 20 bits never move.
 *   flag 0: no digit is 0xFFF, digits unchanged.
 *   flag 1: digits = [index entries][the digits != 0xFFF in order]; entry t = position of the t-th 0xFFF digit
 *           (10 bits) | 0x400 if another entry follows.
 * The flag is stored as word number `words` of the packed block (GF.md:98-100: the 1025th source word).
 * ---------------------------------------------------------------------------------------- */
#define PK_DIGIT(w) ((w) >> 20)
#define PK_LOW(w) ((w) & 0xFFFFFu)

void orc_pack_block(const uint32_t *raw, size_t words, uint32_t *packed)
{
    uint32_t digits[1024];
    size_t m = 0, out = 0;
    for (size_t j = 0; j < words; j++) m += PK_DIGIT(raw[j]) == 0xFFFu;
    if (m == 0) {
        memcpy(packed, raw, words * 4);
        packed[words] = 0;
        return;
    }
    size_t t = 0;
    for (size_t j = 0; j < words; j++)
        if (PK_DIGIT(raw[j]) == 0xFFFu) {
            t++;
            digits[out++] = (uint32_t)j | (t < m ? 0x400u : 0u);
        }
    for (size_t j = 0; j < words; j++)
        if (PK_DIGIT(raw[j]) != 0xFFFu) digits[out++] = PK_DIGIT(raw[j]);
    for (size_t j = 0; j < words; j++) packed[j] = (digits[j] << 20) | PK_LOW(raw[j]);
    packed[words] = 1;
}

/* Returns 0, or -1 for a block no packer produces (raw then gets the first `words` packed words unchanged). */
int orc_unpack_block(const uint32_t *packed, size_t words, uint32_t *raw)
{
    const uint32_t flag = packed[words];
    int bad = flag > 1;
    if (!bad && flag == 0) {
        for (size_t j = 0; j < words; j++) bad |= PK_DIGIT(packed[j]) == 0xFFFu;
    } else if (!bad) {
        uint8_t is_fff[1024];
        memset(is_fff, 0, sizeof is_fff);
        size_t m = 0;
        long prev = -1;
        for (;;) {
            if (m == words) { bad = 1; break; }
            const uint32_t e = PK_DIGIT(packed[m]);
            const long idx = (long)(e & 0x3FFu);
            if ((e & 0x800u) || idx <= prev || (size_t)idx >= words) { bad = 1; break; }
            is_fff[idx] = 1;
            prev = idx;
            m++;
            if (!(e & 0x400u)) break;
        }
        if (!bad) {
            size_t next = m;
            for (size_t j = m; j < words; j++) bad |= PK_DIGIT(packed[j]) == 0xFFFu;
            if (!bad)
                for (size_t j = 0; j < words; j++) {
                    const uint32_t d = is_fff[j] ? 0xFFFu : PK_DIGIT(packed[next++]);
                    raw[j] = (d << 20) | PK_LOW(packed[j]);
                }
        }
    }
    if (bad) {
        memcpy(raw, packed, words * 4);
        return -1;
    }
    if (flag == 0) memcpy(raw, packed, words * 4);
    return 0;
}

void orc_pack_blocks(const uint32_t *raw, size_t N, size_t words, uint32_t *packed)
{
#pragma omp parallel for
    for (size_t i = 0; i < N; i++) orc_pack_block(raw + i * words, words, packed + i * (words + 1));
}

size_t orc_unpack_blocks(const uint32_t *packed, size_t N, size_t words, uint32_t *raw)
{
    size_t bad = 0;
#pragma omp parallel for reduction(+ : bad)
    for (size_t i = 0; i < N; i++) bad += orc_unpack_block(packed + i * (words + 1), words, raw + i * words) != 0;
    return bad;
}

/* ------------------------------------------------------------------------------------------
 * Erasure decoding, checker only.  The reference documents decoding (README.md:83-119, RS.md:42-79) and does not
 * implement it; the product uses the O(N log N) derivative scheme.  This is the independent textbook statement:
 * the codeword is f on the 2N-th roots of unity (position u <-> w_2N^u, data at even u, parity at odd u,
 * RS.cpp:51-54), deg f < N, so f is the Lagrange interpolant through ANY N surviving positions and an erased data
 * block is its value at w_2N^(2i).  O(N^2) scalar work plus O(E*N) per word.  Returns -1 if fewer than N survive.
 * ---------------------------------------------------------------------------------------- */
int orc_decode(uint32_t *data, const uint32_t *parity, const uint8_t *data_present, const uint8_t *parity_present, size_t N,
               size_t size)
{
    const size_t N2 = 2 * N;
    uint32_t *pt = malloc(N2 * 4);      /* w_2N^u */
    size_t *pos = malloc(N * sizeof *pos); /* the N surviving positions used */
    uint32_t *den = malloc(N * 4), *wgt = malloc(N * 4);
    const uint32_t w = orc_gf_root((uint32_t)N2);
    size_t cnt = 0;
    pt[0] = 1;
    for (size_t u = 1; u < N2; u++) pt[u] = orc_gf_mul(pt[u - 1], w);
    for (size_t u = 0; u < N2 && cnt < N; u++)
        if ((u & 1) ? parity_present[u >> 1] : data_present[u >> 1]) pos[cnt++] = u;
    int rc = cnt == N ? 0 : -1;
    if (rc == 0) {
        for (size_t a = 0; a < N; a++) { /* prod_{b != a} (x_a - x_b) */
            uint32_t d = 1;
            for (size_t b = 0; b < N; b++)
                if (b != a) d = orc_gf_mul(d, orc_gf_sub(pt[pos[a]], pt[pos[b]]));
            den[a] = d;
        }
        uint32_t *row = malloc(size * 4);
        for (size_t i = 0; i < N; i++) {
            if (data_present[i]) continue;
            const uint32_t xe = pt[2 * i];
            uint32_t full = 1;
            for (size_t b = 0; b < N; b++) full = orc_gf_mul(full, orc_gf_sub(xe, pt[pos[b]]));
            for (size_t a = 0; a < N; a++)
                wgt[a] = orc_gf_mul(full, orc_gf_inv(orc_gf_mul(orc_gf_sub(xe, pt[pos[a]]), den[a])));
            memset(row, 0, size * 4);
            for (size_t a = 0; a < N; a++) {
                const size_t u = pos[a];
                const uint32_t *src = (u & 1) ? parity + (u >> 1) * size : data + (u >> 1) * size;
                for (size_t s = 0; s < size; s++) row[s] = orc_gf_add(row[s], orc_gf_mul(wgt[a], src[s]));
            }
            memcpy(data + i * size, row, size * 4); /* erased blocks are never read as sources */
        }
        free(row);
    }
    free(pt);
    free(pos);
    free(den);
    free(wgt);
    return rc;
}
