// decode.hip — erasure decoding, the "fastest" scheme of README.md:102-119 / RS.md:42-79, for every code of this library.
//
// The reference documents this algorithm and does not implement it.  For the reference's (2k,k) code the codeword is f
// on the 2k-th roots of unity: position u <-> point w^u (w = w_2k), even positions are the data blocks (u = 2i), odd
// ones the parity blocks (u = 2j+1, RS.cpp:51-54).  With E the erased positions (|E| <= k) and
// l(x) = prod_{e in E} (x - w^e):
//
//   p = f * l has degree < 2k and KNOWN values everywhere: c[u] * l(w^u) at surviving positions, 0 at erased ones;
//   p'(w^e) = f(w^e) * l'(w^e) at an erased position, so  f(w^e) = [x p'(x)](w^e) / (w^e * l'(w^e)).
//
// x p'(x) = sum m p_m x^m needs no coefficient shift, which makes the data-parallel part the SAME pipeline as the
// encoder one size up: inverse transform of size 2k, block holding coefficient m times m / 2k, forward transform —
// i.e. create_transform_ctx(2k, factor[m] = m / 2k) with fold = 1, because only the even (data) positions are wanted.
// Around it: a gather (codeword blocks times l(w^u), zeros at erasures; fused into the transform's first pass for the
// (2k,k) layout) and one pass that multiplies the recovered rows by 1 / (w^e l'(w^e)).
//
// The other codes are the same thing on the (k << e)-th roots of unity (fastecc_decode_prepare): positions that hold no
// block of the code count as erased, zero-extended data blocks as known zeros, the transform has fold = e.
//
// Everything that depends only on the erasure PATTERN is done once in fastecc_decode_prepare: l by a product tree on the
// host, its values and its derivative's values by one device transform of a two-column stripe, one batch inversion.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstring>
#include <new>
#include <vector>

#include "gf.hpp"
#include "internal.hpp"
#include "ntt_device.hpp"

namespace fastecc {

struct DecodeState {
    fastecc_ctx* transform = nullptr;  // size-2k transform context, fold 1
    fastecc_ctx* pattern_ntt = nullptr;  // same length, 2 words per block: l and l' are evaluated on the device
    uint32_t* pattern_buf = nullptr;     // its stripe
    uint32_t* fin = nullptr;           // 2k factors by codeword position: l(w^u) (Montgomery) or 0 if erased
    uint32_t* fin_first_pass = nullptr;  // the same in the order the transform's first pass reads them (may equal fin)
    uint32_t* srcmap = nullptr;        // per codeword position: the block that sits there (row, bit 31 = parity stripe)
    uint32_t* gout = nullptr;          // k factors by data block: 1 / (w^2i l'(w^2i)) (Montgomery) if erased, else 0
    uint32_t* recovered = nullptr;     // k blocks: x p'(x) at the data positions
    uint32_t* parity_dev = nullptr;    // staging for FASTECC_MEM_HOST calls (lazy)
    uint64_t erased_data = 0, erased_total = 0;
    uint64_t positions = 0;            // code length on the roots of unity: k << log2(n / k) rounded up to powers of two
    bool standard = false;             // the reference's (2k,k) layout: position u = data u/2 or parity u/2, every block in memory
    bool ready = false;
};

void destroy_decode_state(DecodeState* d)
{
    if (!d) return;
    if (d->transform) fastecc_destroy(d->transform);
    if (d->pattern_ntt) fastecc_destroy(d->pattern_ntt);
    if (d->pattern_buf) (void)hipFree(d->pattern_buf);
    if (d->fin_first_pass && d->fin_first_pass != d->fin) (void)hipFree(d->fin_first_pass);
    if (d->fin) (void)hipFree(d->fin);
    if (d->srcmap) (void)hipFree(d->srcmap);
    if (d->gout) (void)hipFree(d->gout);
    if (d->recovered) (void)hipFree(d->recovered);
    if (d->parity_dev) (void)hipFree(d->parity_dev);
    delete d;
}

namespace {

// ------------------------------------------------------------------------------------------------
// host arithmetic for the pattern-only part: plain values, products through a Montgomery step
// ------------------------------------------------------------------------------------------------
using gf::P;

inline uint32_t h_add(uint32_t a, uint32_t b)
{
    const uint64_t s = (uint64_t)a + b;
    return (uint32_t)(s >= P ? s - P : s);
}
inline uint32_t h_sub(uint32_t a, uint32_t b) { return a >= b ? a - b : a + P - b; }
// x * w for w given as wm = w * 2^32 mod p (same reduction as gf::mul_mont on the device)
inline uint32_t h_mont(uint32_t x, uint32_t wm)
{
    const uint64_t t = (uint64_t)x * wm;
    const uint32_t lo = (uint32_t)t, hi = (uint32_t)(t >> 32);
    const uint32_t m = lo + (lo << 20);
    const uint32_t q = (uint32_t)(((uint64_t)m * P) >> 32);
    return hi >= q ? hi - q : hi - q + P;
}

// Transforms of any power-of-two size up to 2^20 from one table of w_(2^20)^i (Montgomery form), i < 2^19.
struct HostNtt {
    static constexpr int MAXLOG = 20;
    std::vector<uint32_t> fwd, inv;
    HostNtt() : fwd(1u << (MAXLOG - 1)), inv(1u << (MAXLOG - 1))
    {
        const uint32_t w = gf::h_root(1u << MAXLOG), wi = gf::h_inv(w);
        uint32_t a = 1, b = 1;
        for (size_t i = 0; i < fwd.size(); i++) {
            fwd[i] = gf::h_to_mont(a);
            inv[i] = gf::h_to_mont(b);
            a = gf::h_mul(a, w);
            b = gf::h_mul(b, wi);
        }
    }
    // decimation in frequency: natural order in, bit-reversed order out, unscaled
    void dif(uint32_t* x, int logn, bool inverse) const
    {
        const std::vector<uint32_t>& tw = inverse ? inv : fwd;
        const size_t n = (size_t)1 << logn;
        for (size_t h = n >> 1; h >= 1; h >>= 1) {
            const size_t step = (fwd.size() / h);  // (root of order 2h)^i = w_(2^20)^(i * 2^19 / h)
            for (size_t base = 0; base < n; base += 2 * h)
                for (size_t i = 0; i < h; i++) {
                    const uint32_t a = x[base + i], b = x[base + i + h];
                    x[base + i] = h_add(a, b);
                    x[base + i + h] = h_mont(h_sub(a, b), tw[i * step]);
                }
        }
    }
    // decimation in time: bit-reversed order in, natural order out, unscaled
    void dit(uint32_t* x, int logn, bool inverse) const
    {
        const std::vector<uint32_t>& tw = inverse ? inv : fwd;
        const size_t n = (size_t)1 << logn;
        for (size_t h = 1; h < n; h <<= 1) {
            const size_t step = (fwd.size() / h);
            for (size_t base = 0; base < n; base += 2 * h)
                for (size_t i = 0; i < h; i++) {
                    const uint32_t a = x[base + i], b = h_mont(x[base + i + h], tw[i * step]);
                    x[base + i] = h_add(a, b);
                    x[base + i + h] = h_sub(a, b);
                }
        }
    }
};

const HostNtt& host_ntt()
{
    static const HostNtt t;
    return t;
}

// c = a * b (coefficient vectors, lowest degree first)
std::vector<uint32_t> poly_mul(const std::vector<uint32_t>& a, const std::vector<uint32_t>& b)
{
    const size_t need = a.size() + b.size() - 1;
    std::vector<uint32_t> c(need, 0);
    if (std::min(a.size(), b.size()) <= 32) {
        for (size_t i = 0; i < a.size(); i++) {
            if (!a[i]) continue;
            const uint32_t am = gf::h_to_mont(a[i]);
            for (size_t j = 0; j < b.size(); j++) c[i + j] = h_add(c[i + j], h_mont(b[j], am));
        }
        return c;
    }
    // Two monic polynomials of the same power-of-two degree d (every product of the tree except at its ragged edge):
    // (x^d + a')(x^d + b') = x^2d + x^d (a' + b') + a' b', and a' b' has degree < 2d, so a cyclic product of length 2d
    // is enough — half the transform length of the general case below.
    const size_t d = a.size() - 1;
    const bool monic_pair = a.size() == b.size() && (d & (d - 1)) == 0 && a[d] == 1u && b[d] == 1u;
    const size_t cyc = monic_pair ? 2 * d : need;
    int logn = 0;
    while (((size_t)1 << logn) < cyc) logn++;
    const size_t n = (size_t)1 << logn;
    std::vector<uint32_t> fa(n, 0), fb(n, 0);
    std::copy(a.begin(), a.end() - (monic_pair ? 1 : 0), fa.begin());
    std::copy(b.begin(), b.end() - (monic_pair ? 1 : 0), fb.begin());
    const HostNtt& t = host_ntt();
    t.dif(fa.data(), logn, false);
    t.dif(fb.data(), logn, false);
    // fa * fb / n through two Montgomery steps: (fa fb / R) * (R^2 / n) / R
    const uint32_t scale = gf::h_to_mont(gf::h_to_mont(gf::h_inv((uint32_t)n)));
    for (size_t i = 0; i < n; i++) fa[i] = h_mont(h_mont(fa[i], fb[i]), scale);
    t.dit(fa.data(), logn, true);  // the bit-reversed products go straight back: no permutation anywhere
    if (monic_pair) {
        std::copy(fa.begin(), fa.begin() + 2 * d, c.begin());
        for (size_t i = 0; i < d; i++) c[d + i] = h_add(c[d + i], h_add(a[i], b[i]));
        c[2 * d] = 1u;
    } else {
        std::copy(fa.begin(), fa.begin() + need, c.begin());
    }
    return c;
}

// l(x) = prod (x - roots[i]) by a balanced product tree: O(M log^2 M)
std::vector<uint32_t> poly_from_roots(const std::vector<uint32_t>& roots)
{
    std::vector<std::vector<uint32_t>> level;
    level.reserve(roots.size());
    for (uint32_t r : roots) level.push_back({h_sub(0, r), 1u});
    if (level.empty()) return {1u};
    while (level.size() > 1) {
        std::vector<std::vector<uint32_t>> next;
        next.reserve((level.size() + 1) / 2);
        for (size_t i = 0; i + 1 < level.size(); i += 2) next.push_back(poly_mul(level[i], level[i + 1]));
        if (level.size() & 1) next.push_back(std::move(level.back()));
        level.swap(next);
    }
    return level[0];
}

// ------------------------------------------------------------------------------------------------
// kernels: one wave per (block row, 64*V-word column chunk); the row's factor is a scalar
// ------------------------------------------------------------------------------------------------
// work[u] = block at codeword position u, times fin[u].  srcmap[u] names the block: row index, bit 31 set = parity
// stripe.  Positions without a surviving block have fin == 0: they are written as zeros and nothing is read for them.
template <int V>
__global__ __launch_bounds__(256) void decode_gather_kernel(const uint32_t* __restrict__ data, const uint32_t* __restrict__ parity,
                                                            uint32_t* __restrict__ work, const uint32_t* __restrict__ fin,
                                                            const uint32_t* __restrict__ srcmap, uint32_t S, uint32_t ld, uint32_t ld_work,
                                                            uint32_t col_chunks, uint64_t items)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t item = (uint64_t)blockIdx.x * 4u + wave;
    if (item >= items) return;
    const uint32_t cc = (uint32_t)(item % col_chunks);
    const uint32_t u = (uint32_t)(item / col_chunks);
    const uint32_t col = (cc * 64u + lane) * V;
    if (col >= S) return;
    const uint32_t f = as_constant(fin)[u];
    uint32_t x[V];
    if (f != 0) {
        const uint32_t m = as_constant(srcmap)[u];
        const uint32_t* src = ((m >> 31) ? parity : data) + (size_t)(m & 0x7FFFFFFFu) * ld + col;
        load_vec<V>(x, src);
#pragma unroll
        for (int v = 0; v < V; ++v) x[v] = gf::mul_mont(x[v], f);
    } else {
#pragma unroll
        for (int v = 0; v < V; ++v) x[v] = 0;
    }
    store_vec<V>(work + (size_t)u * ld_work + col, x);
}

// data[i] = recovered[i] * gout[i] for the erased data blocks (gout != 0); surviving blocks are not touched
template <int V>
__global__ __launch_bounds__(256) void decode_scatter_kernel(const uint32_t* __restrict__ recovered, uint32_t* __restrict__ data,
                                                             const uint32_t* __restrict__ gout, uint32_t S, uint32_t ld_rec, uint32_t ld,
                                                             uint32_t col_chunks, uint64_t items)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t item = (uint64_t)blockIdx.x * 4u + wave;
    if (item >= items) return;
    const uint32_t cc = (uint32_t)(item % col_chunks);
    const uint32_t i = (uint32_t)(item / col_chunks);
    const uint32_t f = as_constant(gout)[i];
    if (f == 0) return;  // wave-uniform
    const uint32_t col = (cc * 64u + lane) * V;
    if (col >= S) return;
    uint32_t x[V];
    load_vec<V>(x, recovered + (size_t)i * ld_rec + col);
#pragma unroll
    for (int v = 0; v < V; ++v) x[v] = gf::mul_mont(x[v], f);
    store_vec<V>(data + (size_t)i * ld + col, x);
}

int hip_code(const char* what, hipError_t e)
{
    set_error_detail(what, e);
    return e == hipErrorOutOfMemory ? FASTECC_E_NOMEM : FASTECC_E_DEVICE;
}

#define DEC_TRY(expr)                                      \
    do {                                                   \
        hipError_t e_ = (expr);                            \
        if (e_ != hipSuccess) return hip_code(#expr, e_);  \
    } while (0)

struct DeviceScope {
    int prev = -1;
    bool ok = false;
    explicit DeviceScope(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceScope()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace
}  // namespace fastecc

using namespace fastecc;

extern "C" {

int fastecc_decode_prepare(fastecc_ctx* c, const uint8_t* data_present, const uint8_t* parity_present)
{
    if (!c || !data_present || !parity_present) return FASTECC_E_INVAL;
    if (sharded_of(c)) return FASTECC_E_UNSUPPORTED;
    const CtxInfo ci = info_of(c);
    if (ci.field != FASTECC_FIELD_GF_FFF00001 || ci.q > 1) return FASTECC_E_UNSUPPORTED;  // the decoder's transform is a power of two
    if (ci.pitch != ci.words) return FASTECC_E_UNSUPPORTED;
    const uint64_t N = ci.k;

    // Every code is f on a subset of the NC-th roots of unity, NC = N << e (position u <-> w_NC^u): data block i at
    // i << e (blocks k..N-1 of a zero-extended code are known zero blocks), parity at the positions fastecc_create
    // documents — odd multiples of 2^fold for the codes inside (2N,N), the cosets' offsets for n = 4k / 8k.  Positions
    // that hold no block of the code count as erased, which is exactly what limits the losses to n - k.
    int e = 1;
    while ((1 << e) < ci.cosets + 1) e++;
    const uint64_t NC = N << e;
    const int lgc = ci.log2k + e;
    enum : uint8_t { LOST = 0, HELD = 1, ZERO = 2 };
    std::vector<uint8_t> state(NC, LOST);
    std::vector<uint32_t> srcmap(NC, 0);
    uint64_t erased_data = 0;
    for (uint64_t i = 0; i < N; i++) {
        const uint64_t u = i << e;
        if (i >= ci.user_k) state[u] = ZERO;
        else if (data_present[i]) state[u] = HELD, srcmap[u] = (uint32_t)i;
        else erased_data++;
    }
    for (uint64_t q = 0; q < ci.user_m; q++) {
        uint64_t u;
        if (ci.cosets > 1) {
            const uint64_t t = q / N, j = q % N;  // coset t = generator w_(N << jj)^c, see fastecc_create
            int jj = 1;
            while ((1ull << jj) - 1 <= t) jj++;
            const uint64_t odd = 2 * (t + 1 - (1ull << (jj - 1))) + 1;
            u = (odd << (e - jj)) + (j << e);
        } else {
            u = ((q << ci.fold) << 1) + 1;
        }
        if (parity_present[q]) state[u] = HELD, srcmap[u] = (uint32_t)q | 0x80000000u;
    }
    std::vector<uint32_t> erased;
    for (uint64_t u = 0; u < NC; u++)
        if (state[u] == LOST) erased.push_back((uint32_t)u);
    if (erased.size() > NC - N) return FASTECC_E_INVAL;  // fewer than k blocks survive: not decodable

    DeviceScope ds(ci.device);
    if (!ds.ok) return FASTECC_E_DEVICE;
    CallScope call(c);
    DecodeState*& slot = decoder_of(c);
    if (!slot) {
        slot = new (std::nothrow) DecodeState();
        if (!slot) return FASTECC_E_NOMEM;
    }
    DecodeState* d = slot;
    d->ready = false;
    d->erased_data = erased_data;
    d->erased_total = erased.size();
    d->positions = NC;
    d->standard = ci.cosets == 1 && ci.fold == 0 && !ci.zero_extended;
    if (erased_data == 0) {  // nothing to recover
        d->ready = true;
        return FASTECC_OK;
    }

    // ---- pattern-only scalars ----
    const uint32_t w = gf::h_root((uint32_t)NC);
    std::vector<uint32_t> wpow(NC);  // w^u
    {
        uint32_t a = 1;
        for (uint64_t u = 0; u < NC; u++) {
            wpow[u] = a;
            a = gf::h_mul(a, w);
        }
    }
    std::vector<uint32_t> roots(erased.size());
    for (size_t i = 0; i < erased.size(); i++) roots[i] = wpow[erased[i]];
    const std::vector<uint32_t> l = poly_from_roots(roots);  // degree |E| <= NC - N < NC
    // values of l and l' on all NC points: one forward transform of a two-column stripe (column 0 = l, column 1 = l'),
    // on the device — the same kernels as everything else, natural order in and out
    std::vector<uint32_t> lv(2 * NC, 0);
    for (size_t m = 0; m < l.size(); m++) lv[2 * m] = l[m];
    for (size_t m = 0; m + 1 < l.size(); m++) lv[2 * m + 1] = gf::h_mul((uint32_t)((m + 1) % P), l[m + 1]);
    if (!d->pattern_ntt) {
        const std::vector<uint32_t> ones(NC, 1u);
        const int rc = create_transform_ctx(&d->pattern_ntt, lgc, 8, 0, ones.data(), ci.device);
        if (rc != FASTECC_OK) return rc;
    }
    if (!d->pattern_buf) DEC_TRY(hipMalloc((void**)&d->pattern_buf, 2 * NC * 4));
    DEC_TRY(hipMemcpy(d->pattern_buf, lv.data(), 2 * NC * 4, hipMemcpyHostToDevice));
    {
        const int rc = fastecc_ntt(d->pattern_ntt, d->pattern_buf, 0, FASTECC_MEM_DEVICE, nullptr);
        if (rc != FASTECC_OK) return rc;
    }
    DEC_TRY(hipMemcpy(lv.data(), d->pattern_buf, 2 * NC * 4, hipMemcpyDeviceToHost));

    std::vector<uint32_t> fin(NC, 0), gout(N, 0);
    for (uint64_t u = 0; u < NC; u++)  // zero blocks contribute 0 * l(w^u): factor 0 keeps every kernel from reading them
        if (state[u] == HELD) fin[u] = gf::h_to_mont(lv[2 * u]);
    {
        // 1 / (w^u l'(w^u)) for the erased data positions with ONE inversion (prefix products)
        std::vector<uint32_t> den, prefix;
        std::vector<uint64_t> who;
        for (uint64_t i = 0; i < ci.user_k; i++) {
            if (data_present[i]) continue;
            const uint64_t u = i << e;
            const uint32_t v = gf::h_mul(wpow[u], lv[2 * u + 1]);
            if (v == 0) return FASTECC_E_INVAL;  // cannot happen: l has simple roots
            den.push_back(v);
            who.push_back(i);
        }
        prefix.resize(den.size());
        uint32_t acc = 1;
        for (size_t j = 0; j < den.size(); j++) {
            prefix[j] = acc;
            acc = gf::h_mul(acc, den[j]);
        }
        uint32_t inv = gf::h_inv(acc);
        for (size_t j = den.size(); j-- > 0;) {
            gout[who[j]] = gf::h_to_mont(gf::h_mul(inv, prefix[j]));
            inv = gf::h_mul(inv, den[j]);
        }
    }

    // ---- device state ----
    if (!d->transform) {
        std::vector<uint32_t> factor(NC);
        const uint32_t inv_nc = gf::h_inv((uint32_t)NC);
        for (uint64_t m = 0; m < NC; m++) factor[m] = gf::h_mul((uint32_t)m, inv_nc);  // x p'(x): coefficient m times m, and the 1/NC of the inverse transform
        // fold e: only the data positions (multiples of 2^e) are evaluated
        const int rc = create_transform_ctx(&d->transform, lgc, ci.words * 4, e, factor.data(), ci.device);
        if (rc != FASTECC_OK) return rc;
    }
    if (!d->fin) DEC_TRY(hipMalloc((void**)&d->fin, NC * 4));
    if (!d->srcmap) DEC_TRY(hipMalloc((void**)&d->srcmap, NC * 4));
    if (!d->gout) DEC_TRY(hipMalloc((void**)&d->gout, N * 4));
    if (!d->recovered) DEC_TRY(hipMalloc((void**)&d->recovered, N * ci.words * 4));
    {
        const int rc = call.wait_idle();  // a decode still using the previous pattern
        if (rc != FASTECC_OK) return rc;
    }
    DEC_TRY(hipMemcpy(d->fin, fin.data(), NC * 4, hipMemcpyHostToDevice));
    DEC_TRY(hipMemcpy(d->srcmap, srcmap.data(), NC * 4, hipMemcpyHostToDevice));
    {
        std::vector<uint32_t> order;
        if (d->standard && gather_tile_order(d->transform, order)) {
            std::vector<uint32_t> tiled(NC);
            for (uint64_t i = 0; i < NC; i++) tiled[i] = fin[order[i]];
            if (!d->fin_first_pass || d->fin_first_pass == d->fin) DEC_TRY(hipMalloc((void**)&d->fin_first_pass, NC * 4));
            DEC_TRY(hipMemcpy(d->fin_first_pass, tiled.data(), NC * 4, hipMemcpyHostToDevice));
        } else {
            d->fin_first_pass = d->fin;
        }
    }
    DEC_TRY(hipMemcpy(d->gout, gout.data(), N * 4, hipMemcpyHostToDevice));
    d->ready = true;
    return FASTECC_OK;
}

int fastecc_decode(fastecc_ctx* c, void* data, const void* parity, int mem_kind, void* stream)
{
    if (!c || !data || !parity || (((uintptr_t)data | (uintptr_t)parity) & 3u)) return FASTECC_E_INVAL;
    if (mem_kind != FASTECC_MEM_HOST && mem_kind != FASTECC_MEM_DEVICE) return FASTECC_E_INVAL;
    if (sharded_of(c)) return FASTECC_E_UNSUPPORTED;
    CallScope call(c);
    DecodeState* d = decoder_of(c);
    if (!d || !d->ready) return FASTECC_E_INVAL;  // fastecc_decode_prepare first
    if (d->erased_data == 0) return FASTECC_OK;
    const CtxInfo ci = info_of(c);
    if (ci.pitch != ci.words) return FASTECC_E_UNSUPPORTED;  // the gather / scatter passes address contiguous stripes
    DeviceScope ds(ci.device);
    if (!ds.ok) return FASTECC_E_DEVICE;
    hipStream_t st = (hipStream_t)stream;
    struct Marker {  // the decoder's work stripes are internal buffers: order their uses between streams
        CallScope& s;
        hipStream_t st;
        ~Marker() { (void)s.end(st); }
    };
    {
        const int rc0 = call.begin(st);
        if (rc0 != FASTECC_OK) return rc0;
    }
    Marker marker{call, st};
    const uint64_t N = ci.k;
    const size_t block = ci.words * 4, data_bytes = ci.user_k * block, parity_bytes = ci.user_m * block;

    uint32_t* ddata = (uint32_t*)data;
    const uint32_t* dparity = (const uint32_t*)parity;
    if (mem_kind == FASTECC_MEM_HOST) {
        // stage both parts of the codeword
        if (!d->parity_dev) DEC_TRY(hipMalloc((void**)&d->parity_dev, parity_bytes + data_bytes));
        DEC_TRY(hipMemcpyAsync(d->parity_dev, parity, parity_bytes, hipMemcpyHostToDevice, st));
        DEC_TRY(hipMemcpyAsync(d->parity_dev + ci.user_m * ci.words, data, data_bytes, hipMemcpyHostToDevice, st));
        dparity = d->parity_dev;
        ddata = d->parity_dev + ci.user_m * ci.words;
    }

    // The (2k,k) layout lets the transform's first pass read the two halves of the codeword itself (no gather pass).
    // The other codes do not hold every position in memory: they take the table-driven gather, which never touches a
    // position whose factor is zero, instead of a tile that reads first and multiplies by zero afterwards.
    int rc = d->standard ? run_gathered(d->transform, ddata, dparity, d->fin_first_pass, d->recovered, st) : FASTECC_E_UNSUPPORTED;
    const bool fused = rc == FASTECC_OK;
    if (!fused && rc != FASTECC_E_UNSUPPORTED) return rc;
    uint32_t* work = nullptr;
    rc = scratch_of(d->transform, &work);
    if (rc != FASTECC_OK) return rc;
    const uint32_t S = (uint32_t)ci.words;
    const bool v4 = (S % 4) == 0 && ((((uintptr_t)ddata | (uintptr_t)dparity | (uintptr_t)work | (uintptr_t)d->recovered) & 15u) == 0);
    const uint32_t col_chunks = (S + (v4 ? 256 : 64) - 1) / (v4 ? 256 : 64);
    if (!fused) {
        const uint64_t items = d->positions * col_chunks;
        const dim3 grid((unsigned)((items + 3) / 4));
        if (v4) hipLaunchKernelGGL(decode_gather_kernel<4>, grid, dim3(256), 0, st, ddata, dparity, work, d->fin, d->srcmap, S, S, S, col_chunks, items);
        else    hipLaunchKernelGGL(decode_gather_kernel<1>, grid, dim3(256), 0, st, ddata, dparity, work, d->fin, d->srcmap, S, S, S, col_chunks, items);
        DEC_TRY(hipGetLastError());
        rc = fastecc_encode(d->transform, work, d->recovered, FASTECC_MEM_DEVICE, st);
        if (rc != FASTECC_OK) return rc;
    }
    {
        const uint64_t items = N * col_chunks;
        const dim3 grid((unsigned)((items + 3) / 4));
        if (v4) hipLaunchKernelGGL(decode_scatter_kernel<4>, grid, dim3(256), 0, st, d->recovered, ddata, d->gout, S, S, S, col_chunks, items);
        else    hipLaunchKernelGGL(decode_scatter_kernel<1>, grid, dim3(256), 0, st, d->recovered, ddata, d->gout, S, S, S, col_chunks, items);
        DEC_TRY(hipGetLastError());
    }
    if (mem_kind == FASTECC_MEM_HOST) {
        DEC_TRY(hipMemcpyAsync(data, ddata, data_bytes, hipMemcpyDeviceToHost, st));
        DEC_TRY(hipStreamSynchronize(st));
    }
    return FASTECC_OK;
}

}  // extern "C"
