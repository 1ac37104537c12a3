/*
 * fastecc_oracle_p61.c — CPU ORACLE for the GF((2^61-1)^2) configuration (TEST INFRASTRUCTURE ONLY).
 * See fastecc_oracle_p61.h: PARITY UNPINNED (the reference has no code for this field); the composition
 * follows RS.cpp:40-63 and the transform definition ntt.cpp:451-483.  Deliberately simple: 128-bit
 * products with %, one column at a time.
 */
#include "fastecc_oracle_p61.h"

#include <stdlib.h>
#include <string.h>

#define P ORC61_P
typedef unsigned __int128 u128;

uint64_t orc61_add(uint64_t x, uint64_t y)
{
    uint64_t s = x + y; /* < 2^62 */
    return s >= P ? s - P : s;
}

uint64_t orc61_sub(uint64_t x, uint64_t y)
{
    return x >= y ? x - y : x + P - y;
}

uint64_t orc61_mul(uint64_t x, uint64_t y)
{
    return (uint64_t)(((u128)x * y) % P);
}

/* (a+bi)(c+di) = (ac - bd) + (ad + bc)i,  i^2 = -1 */
void orc61c_mul(const uint64_t x[2], const uint64_t y[2], uint64_t out[2])
{
    const uint64_t a = x[0], b = x[1], c = y[0], d = y[1];
    const uint64_t re = orc61_sub(orc61_mul(a, c), orc61_mul(b, d));
    const uint64_t im = orc61_add(orc61_mul(a, d), orc61_mul(b, c));
    out[0] = re;
    out[1] = im;
}

void orc61c_pow(const uint64_t x[2], uint64_t e, uint64_t out[2])
{
    uint64_t r[2] = {1, 0}, b[2] = {x[0], x[1]};
    for (; e; e >>= 1) {
        if (e & 1) orc61c_mul(r, b, r);
        orc61c_mul(b, b, b);
    }
    out[0] = r[0];
    out[1] = r[1];
}

/* 1/(a+bi) = (a - bi) / (a^2 + b^2); the norm is inverted in GF(p) by Fermat. */
void orc61c_inv(const uint64_t x[2], uint64_t out[2])
{
    const uint64_t norm = orc61_add(orc61_mul(x[0], x[0]), orc61_mul(x[1], x[1]));
    uint64_t inv = 1, b = norm, e = P - 2;
    for (; e; e >>= 1) {
        if (e & 1) inv = orc61_mul(inv, b);
        b = orc61_mul(b, b);
    }
    out[0] = orc61_mul(x[0], inv);
    out[1] = orc61_mul(orc61_sub(0, x[1]), inv);
}

void orc61c_root(uint64_t order, uint64_t out[2])
{
    out[0] = out[1] = 0;
    if (order == 0 || (order & (order - 1)) != 0 || order > (1ull << 62)) return;
    const uint64_t g[2] = {4, 1};
    uint64_t w[2];
    orc61c_pow(g, (1ull << 60) - 1, w);           /* order exactly 2^62 */
    orc61c_pow(w, (1ull << 62) / order, out);
}

/* ------------------------------------------------------------------------------------------ */

static void column_get(const uint64_t *data, size_t N, size_t elems, size_t c, uint64_t *col)
{
    for (size_t i = 0; i < N; i++) {
        col[2 * i] = data[(i * elems + c) * 2];
        col[2 * i + 1] = data[(i * elems + c) * 2 + 1];
    }
}

static void column_put(uint64_t *data, size_t N, size_t elems, size_t c, const uint64_t *col)
{
    for (size_t i = 0; i < N; i++) {
        data[(i * elems + c) * 2] = col[2 * i];
        data[(i * elems + c) * 2 + 1] = col[2 * i + 1];
    }
}

/* X_j = sum_i x_i w^(ij)  (ntt.cpp:451-483), w = root(N) or its inverse; unscaled either way. */
void orc61_slow_ntt(uint64_t *data, size_t N, size_t elems, int inverse)
{
    uint64_t w[2];
    orc61c_root(N, w);
    if (inverse) orc61c_inv(w, w);
    uint64_t *col = malloc(N * 16), *res = malloc(N * 16);
    for (size_t c = 0; c < elems; c++) {
        column_get(data, N, elems, c, col);
        for (size_t j = 0; j < N; j++) {
            uint64_t wj[2], t[2] = {1, 0}, acc[2] = {0, 0}, m[2];
            orc61c_pow(w, j, wj);
            for (size_t i = 0; i < N; i++) {
                orc61c_mul(&col[2 * i], t, m);
                acc[0] = orc61_add(acc[0], m[0]);
                acc[1] = orc61_add(acc[1], m[1]);
                orc61c_mul(t, wj, t);
            }
            res[2 * j] = acc[0];
            res[2 * j + 1] = acc[1];
        }
        column_put(data, N, elems, c, res);
    }
    free(col);
    free(res);
}

/* In-place radix-2 decimation in time on one contiguous column (bit-reversal, then levels of
 * half-size h = 1, 2, .. N/2 with twiddle (root of order 2h)^i, the structure of ntt.cpp:251-318). */
static void column_ntt(uint64_t *col, size_t N, const uint64_t *tw /* tw[i] = w^i, i < N/2 */)
{
    int bits = 0;
    while (((size_t)1 << bits) < N) bits++;
    for (size_t i = 0; i < N; i++) {
        size_t r = 0;
        for (int b = 0; b < bits; b++) r |= ((i >> b) & 1) << (bits - 1 - b);
        if (r > i) {
            uint64_t t0 = col[2 * i], t1 = col[2 * i + 1];
            col[2 * i] = col[2 * r];
            col[2 * i + 1] = col[2 * r + 1];
            col[2 * r] = t0;
            col[2 * r + 1] = t1;
        }
    }
    for (size_t h = 1; h < N; h *= 2) {
        const size_t step = N / (2 * h);
        for (size_t base = 0; base < N; base += 2 * h) {
            for (size_t i = 0; i < h; i++) {
                uint64_t *u = &col[2 * (base + i)], *v = &col[2 * (base + i + h)], m[2];
                orc61c_mul(v, &tw[2 * i * step], m);
                v[0] = orc61_sub(u[0], m[0]);
                v[1] = orc61_sub(u[1], m[1]);
                u[0] = orc61_add(u[0], m[0]);
                u[1] = orc61_add(u[1], m[1]);
            }
        }
    }
}

void orc61_ntt(uint64_t *data, size_t N, size_t elems, int inverse)
{
    if (N < 2) return;
    uint64_t w[2];
    orc61c_root(N, w);
    if (inverse) orc61c_inv(w, w);
    uint64_t *tw = malloc((N / 2) * 16);
    tw[0] = 1;
    tw[1] = 0;
    for (size_t i = 1; i < N / 2; i++) orc61c_mul(&tw[2 * (i - 1)], w, &tw[2 * i]);
#pragma omp parallel
    {
        uint64_t *col = malloc(N * 16);
#pragma omp for
        for (size_t c = 0; c < elems; c++) {
            column_get(data, N, elems, c, col);
            column_ntt(col, N, tw);
            column_put(data, N, elems, c, col);
        }
        free(col);
    }
    free(tw);
}

/* block i *= scale * base^i  (RS.cpp:51-59 with scale = 1/N, base = root(2N)) */
void orc61_scale_blocks(uint64_t *data, size_t N, size_t elems, const uint64_t scale[2], const uint64_t base[2])
{
    uint64_t f[2] = {scale[0], scale[1]};
    for (size_t i = 0; i < N; i++) {
        for (size_t c = 0; c < elems; c++) orc61c_mul(&data[(i * elems + c) * 2], f, &data[(i * elems + c) * 2]);
        orc61c_mul(f, base, f);
    }
}

/* RS.cpp:40-63: inverse transform, multiply block i by w_2N^i / N, forward transform. */
void orc61_encode(uint64_t *data, size_t N, size_t elems)
{
    uint64_t n_elem[2] = {(uint64_t)N % P, 0}, inv_n[2], w2n[2];
    orc61c_inv(n_elem, inv_n);
    orc61c_root(2 * N, w2n);
    orc61_ntt(data, N, elems, 1);
    orc61_scale_blocks(data, N, elems, inv_n, w2n);
    orc61_ntt(data, N, elems, 0);
}

/* parity[j] = f(w_2N^(2j+1)) where f is the polynomial of degree < N with f(w_N^i) = data[i]:
 * coefficients by the O(N^2) inverse transform and 1/N, then Horner-free direct evaluation. */
void orc61_encode_by_definition(const uint64_t *data, uint64_t *parity, size_t N, size_t elems)
{
    uint64_t *coef = malloc(N * elems * 16);
    memcpy(coef, data, N * elems * 16);
    orc61_slow_ntt(coef, N, elems, 1);
    uint64_t n_elem[2] = {(uint64_t)N % P, 0}, inv_n[2], w2n[2], one[2] = {1, 0};
    orc61c_inv(n_elem, inv_n);
    orc61_scale_blocks(coef, N, elems, inv_n, one);
    orc61c_root(2 * N, w2n);
    for (size_t j = 0; j < N; j++) {
        uint64_t x[2];
        orc61c_pow(w2n, 2 * j + 1, x);
        for (size_t c = 0; c < elems; c++) {
            uint64_t acc[2] = {0, 0}, t[2] = {1, 0}, m[2];
            for (size_t i = 0; i < N; i++) {
                orc61c_mul(&coef[(i * elems + c) * 2], t, m);
                acc[0] = orc61_add(acc[0], m[0]);
                acc[1] = orc61_add(acc[1], m[1]);
                orc61c_mul(t, x, t);
            }
            parity[(j * elems + c) * 2] = acc[0];
            parity[(j * elems + c) * 2 + 1] = acc[1];
        }
    }
    free(coef);
}

void orc61_fill_splitmix(uint64_t *data, size_t nwords, uint64_t seed)
{
    uint64_t s = seed;
    for (size_t i = 0; i < nwords; i++) {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        data[i] = z % P;
    }
}

/* ------------------------------------------------------------------------------------------
 * Erasure decoding by Lagrange interpolation, O(N^2): the oracle of gf61_decode.hip (the reference only describes
 * decoding, README.md:83-119).  Codeword position u <-> w_2N^u: data block i at u = 2i, parity block j at u = 2j+1.  The
 * erased data blocks are rebuilt from the first N surviving positions; returns -1 when fewer than N survive.  Same
 * structure as orc_decode over GF(0xFFF00001).
 * ---------------------------------------------------------------------------------------- */
static void c_sub(const uint64_t x[2], const uint64_t y[2], uint64_t out[2])
{
    out[0] = orc61_sub(x[0], y[0]);
    out[1] = orc61_sub(x[1], y[1]);
}

int orc61_decode(uint64_t *data, const uint64_t *parity, const uint8_t *data_present, const uint8_t *parity_present, size_t N, size_t elems)
{
    const size_t N2 = 2 * N;
    uint64_t(*pt)[2] = malloc(N2 * 16), (*den)[2] = malloc(N * 16), (*wgt)[2] = malloc(N * 16);
    size_t *pos = malloc(N * sizeof *pos);
    uint64_t w[2];
    orc61c_root(N2, w);
    pt[0][0] = 1;
    pt[0][1] = 0;
    for (size_t u = 1; u < N2; u++) orc61c_mul(pt[u - 1], w, pt[u]);
    size_t cnt = 0;
    for (size_t u = 0; u < N2 && cnt < N; u++)
        if ((u & 1) ? parity_present[u >> 1] : data_present[u >> 1]) pos[cnt++] = u;
    const int rc = cnt == N ? 0 : -1;
    if (rc == 0) {
        for (size_t a = 0; a < N; a++) { /* prod_{b != a} (x_a - x_b) */
            uint64_t d[2] = {1, 0}, t[2];
            for (size_t b = 0; b < N; b++)
                if (b != a) {
                    c_sub(pt[pos[a]], pt[pos[b]], t);
                    orc61c_mul(d, t, d);
                }
            den[a][0] = d[0];
            den[a][1] = d[1];
        }
        uint64_t *row = malloc(elems * 16);
        for (size_t i = 0; i < N; i++) {
            if (data_present[i]) continue;
            const uint64_t *xe = pt[2 * i];
            uint64_t full[2] = {1, 0}, t[2];
            for (size_t b = 0; b < N; b++) {
                c_sub(xe, pt[pos[b]], t);
                orc61c_mul(full, t, full);
            }
            for (size_t a = 0; a < N; a++) {
                c_sub(xe, pt[pos[a]], t);
                orc61c_mul(t, den[a], t);
                orc61c_inv(t, t);
                orc61c_mul(full, t, wgt[a]);
            }
            memset(row, 0, elems * 16);
            for (size_t a = 0; a < N; a++) {
                const size_t u = pos[a];
                const uint64_t *src = (u & 1) ? parity + (u >> 1) * elems * 2 : data + (u >> 1) * elems * 2;
                for (size_t s = 0; s < elems; s++) {
                    orc61c_mul(wgt[a], src + 2 * s, t);
                    row[2 * s] = orc61_add(row[2 * s], t[0]);
                    row[2 * s + 1] = orc61_add(row[2 * s + 1], t[1]);
                }
            }
            memcpy(data + i * elems * 2, row, elems * 16); /* erased blocks are never read as sources */
        }
        free(row);
    }
    free(pt);
    free(den);
    free(wgt);
    free(pos);
    return rc;
}
