// gf61_kernels.hip — the encode path over GF((2^61-1)^2): gfx950 kernels, pass plan and tables.
//
// Same operation as the 32-bit path (RS.cpp:40-63: unscaled inverse transform, block i *= w_2N^i / N, forward
// transform; butterfly ntt.cpp:16-22 / 251-284) over the field of gf61.hpp, for BASELINE.json configs[4]
// (64 KB blocks).  The reference has no code for this field, see gf61.hpp / include/fastecc.h.
//
// Data: a stripe is X[N][E] of 16-byte elements (re, im), block-major.  Mapping:
//
//      lane  <->  one element of a block               (a wave reads 64 * 16 = 1024 contiguous bytes per block)
//      wave  <->  R = 2^r blocks of one 64-element column chunk, held in VGPRs (4 per element)
//      twiddles are wave-uniform: fetched with scalar loads, their 31/30-bit limbs live in SGPRs (gf61.hpp)
//
// A pass runs r consecutive radix-2 levels in registers (DIF going down, DIT coming up, MID = the lowest
// levels of both with the per-block factor in between), exactly like the register passes of kernels.hip;
// encode = DIF passes over all levels, factor, DIT passes, and no permutation pass.  Values stay lazy
// (< 2^61 + 16) between passes; the last pass of a transform writes canonical words.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdio>
#include <new>
#include <string>
#include <vector>

#include "../../include/fastecc.h"
#include "gf61.hpp"
#include "gf61_path.hpp"

namespace fastecc {
namespace p61 {

namespace {

enum { MODE_DIF = 0, MODE_DIT = 1, MODE_MID = 2 };

struct PassArgs {
    const uint64_t* in;
    uint64_t* out;
    const uint64_t* tw_dif;  // level-packed twiddles for the DIF levels (16 bytes per entry)
    const uint64_t* tw_dit;  // ... for the DIT levels
    const uint64_t* dscale;  // position p -> w_2N^bitrev(p) / N
    uint32_t elems;          // elements per block
    uint32_t col_chunks;     // ceil(elems / 64)
    uint64_t items;          // (N >> r) * col_chunks
    int s;                   // log2 of the smallest stride of the pass
};

using gf61::Elem;
typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
using const_u64_ptr = const uint64_t __attribute__((address_space(4)))*;

__device__ __forceinline__ const_u64_ptr as_constant(const uint64_t* p) { return (const_u64_ptr)(reinterpret_cast<uintptr_t>(p)); }

__device__ __forceinline__ Elem load_elem(const uint64_t* p)
{
    // every element is read once and written once per pass: non-temporal keeps the stream from displacing itself in L2/MALL
    const u64x2 t = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(p));  // global_load_dwordx4 ... nt
    return Elem{t.x, t.y};
}

__device__ __forceinline__ void store_elem(uint64_t* p, Elem e)
{
    u64x2 t;
    t.x = e.re;
    t.y = e.im;
    __builtin_nontemporal_store(t, reinterpret_cast<u64x2*>(p));
}

// Twiddles of level l = sl + T for a lane set holding blocks (.. + j*2^sl + off): entries
// 2^l + (off << T) + m, m < 2^T, of the level-packed table (same packing as ntt_device.hpp).
template <int LOGR, bool LO_ZERO, int T>
__device__ __forceinline__ void dif_one_level(Elem (&x)[1 << LOGR], const uint64_t* twl, uint32_t off, int sl)
{
    constexpr int R = 1 << LOGR, half = 1 << T;
    const_u64_ptr p = as_constant(twl) + 2 * (((size_t)1 << (sl + T)) + ((size_t)off << T));
#pragma unroll
    for (int m = 0; m < half; ++m) {
        const bool unit = LO_ZERO && m == 0;  // exponent 0 (ntt.cpp:259-267)
        const gf61::Twiddle w = gf61::make_twiddle(p[2 * m], p[2 * m + 1]);
#pragma unroll
        for (int j0 = 0; j0 < R; j0 += 2 * half) {
            const int ja = j0 + m, jb = ja + half;
            const Elem a = x[ja], b = x[jb];
            x[ja] = gf61::add(a, b);
            const Elem d = gf61::sub(a, b);
            x[jb] = unit ? d : gf61::mul(d, w);
        }
    }
}

template <int LOGR, bool LO_ZERO, int T>
__device__ __forceinline__ void dit_one_level(Elem (&x)[1 << LOGR], const uint64_t* twl, uint32_t off, int sl)
{
    constexpr int R = 1 << LOGR, half = 1 << T;
    const_u64_ptr p = as_constant(twl) + 2 * (((size_t)1 << (sl + T)) + ((size_t)off << T));
#pragma unroll
    for (int m = 0; m < half; ++m) {
        const bool unit = LO_ZERO && m == 0;
        const gf61::Twiddle w = gf61::make_twiddle(p[2 * m], p[2 * m + 1]);
#pragma unroll
        for (int j0 = 0; j0 < R; j0 += 2 * half) {
            const int ja = j0 + m, jb = ja + half;
            const Elem a = x[ja];
            const Elem b = unit ? x[jb] : gf61::mul(x[jb], w);
            x[ja] = gf61::add(a, b);
            x[jb] = gf61::sub(a, b);
        }
    }
}

template <int LOGR, bool LO_ZERO>
__device__ __forceinline__ void dif_levels(Elem (&x)[1 << LOGR], const uint64_t* twl, uint32_t off, int sl)
{
    if constexpr (LOGR >= 5) dif_one_level<LOGR, LO_ZERO, 4>(x, twl, off, sl);
    if constexpr (LOGR >= 4) dif_one_level<LOGR, LO_ZERO, 3>(x, twl, off, sl);
    if constexpr (LOGR >= 3) dif_one_level<LOGR, LO_ZERO, 2>(x, twl, off, sl);
    if constexpr (LOGR >= 2) dif_one_level<LOGR, LO_ZERO, 1>(x, twl, off, sl);
    dif_one_level<LOGR, LO_ZERO, 0>(x, twl, off, sl);
}

template <int LOGR, bool LO_ZERO>
__device__ __forceinline__ void dit_levels(Elem (&x)[1 << LOGR], const uint64_t* twl, uint32_t off, int sl)
{
    dit_one_level<LOGR, LO_ZERO, 0>(x, twl, off, sl);
    if constexpr (LOGR >= 2) dit_one_level<LOGR, LO_ZERO, 1>(x, twl, off, sl);
    if constexpr (LOGR >= 3) dit_one_level<LOGR, LO_ZERO, 2>(x, twl, off, sl);
    if constexpr (LOGR >= 4) dit_one_level<LOGR, LO_ZERO, 3>(x, twl, off, sl);
    if constexpr (LOGR >= 5) dit_one_level<LOGR, LO_ZERO, 4>(x, twl, off, sl);
}

// One register pass.  Work item = (block group g, column chunk cc); a wave owns one work item.
template <int LOGR, int MODE, bool CANON>
__global__ __launch_bounds__(256, (LOGR <= 4 ? 3 : 1)) void p61_pass_kernel(const PassArgs a)
{
    constexpr int R = 1 << LOGR;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t item = (uint64_t)blockIdx.x * 4u + wave;
    if (item >= a.items) return;  // wave-uniform
    const uint32_t cc = (uint32_t)(item % a.col_chunks);
    const uint32_t g = (uint32_t)(item / a.col_chunks);
    const uint32_t col = cc * 64u + lane;
    const bool live = col < a.elems;

    const int s = MODE == MODE_MID ? 0 : a.s;
    const uint32_t lo = g & ((1u << s) - 1u);
    const uint32_t hi = g >> s;
    const uint64_t base = ((uint64_t)hi << (s + LOGR)) + lo;  // first block of this group
    const uint64_t row_words = 2ull * a.elems;

    Elem x[R];
    if (live) {
#pragma unroll
        for (int j = 0; j < R; ++j) x[j] = load_elem(a.in + (base + ((uint64_t)j << s)) * row_words + 2u * col);
    } else {
#pragma unroll
        for (int j = 0; j < R; ++j) x[j] = Elem{0, 0};
    }

    if constexpr (MODE == MODE_DIF) {
        if (a.s == 0) dif_levels<LOGR, true>(x, a.tw_dif, 0u, 0);
        else          dif_levels<LOGR, false>(x, a.tw_dif, lo, s);
    } else if constexpr (MODE == MODE_DIT) {
        if (a.s == 0) dit_levels<LOGR, true>(x, a.tw_dit, 0u, 0);
        else          dit_levels<LOGR, false>(x, a.tw_dit, lo, s);
    } else {
        dif_levels<LOGR, true>(x, a.tw_dif, 0u, 0);
        // position p = hi*R + j holds coefficient bitrev_n(p); dscale is stored in position order
        const_u64_ptr d = as_constant(a.dscale) + 2 * ((size_t)hi * R);
#pragma unroll
        for (int j = 0; j < R; ++j) x[j] = gf61::mul(x[j], gf61::make_twiddle(d[2 * j], d[2 * j + 1]));
        dit_levels<LOGR, true>(x, a.tw_dit, 0u, 0);
    }

    if (live) {
#pragma unroll
        for (int j = 0; j < R; ++j)
            store_elem(a.out + (base + ((uint64_t)j << s)) * row_words + 2u * col, CANON ? gf61::canon(x[j]) : x[j]);
    }
}

// Swap block j with block bitrev(j): only the stand-alone transform needs it (the reference permutes
// pointers instead, ntt.cpp:292-309).
__global__ __launch_bounds__(256) void p61_bitrev_rows_kernel(uint64_t* data, uint32_t elems, int n, uint32_t col_chunks, uint64_t items)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t item = (uint64_t)blockIdx.x * 4u + wave;
    if (item >= items) return;
    const uint32_t cc = (uint32_t)(item % col_chunks);
    const uint32_t j = (uint32_t)(item / col_chunks);
    const uint32_t rj = __brev(j) >> (32 - n);
    if (rj <= j) return;
    const uint32_t col = cc * 64u + lane;
    if (col >= elems) return;
    uint64_t* pa = data + ((uint64_t)j * elems + col) * 2;
    uint64_t* pb = data + ((uint64_t)rj * elems + col) * 2;
    const Elem va = load_elem(pa), vb = load_elem(pb);
    store_elem(pa, vb);
    store_elem(pb, va);
}

__global__ __launch_bounds__(256) void p61_count_out_of_range_kernel(const uint64_t* data, uint64_t words, unsigned long long* counter)
{
    unsigned long long bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (uint64_t)gridDim.x * blockDim.x)
        bad += data[i] >= gf61::P;
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_down(bad, o);
    if ((threadIdx.x & 63u) == 0 && bad) atomicAdd(counter, bad);
}

template <int LOGR, int MODE>
hipError_t launch_canon(bool canon, const PassArgs& a, dim3 grid, hipStream_t st)
{
    if (canon) hipLaunchKernelGGL((p61_pass_kernel<LOGR, MODE, true>), grid, dim3(256), 0, st, a);
    else       hipLaunchKernelGGL((p61_pass_kernel<LOGR, MODE, false>), grid, dim3(256), 0, st, a);
    return hipGetLastError();
}

template <int LOGR>
hipError_t launch_mode(int mode, bool canon, const PassArgs& a, dim3 grid, hipStream_t st)
{
    switch (mode) {
    case MODE_DIF: return launch_canon<LOGR, MODE_DIF>(canon, a, grid, st);
    case MODE_DIT: return launch_canon<LOGR, MODE_DIT>(canon, a, grid, st);
    default:       return launch_canon<LOGR, MODE_MID>(canon, a, grid, st);
    }
}

struct Pass {
    int mode, logr, s;
    bool canon;  // last pass of the transform: write canonical words
};

}  // namespace

struct Path {
    int n = 0;
    uint64_t N = 0, elems = 0;
    int levels = DEFAULT_LEVELS;  // levels per register pass
    std::vector<Pass> enc, fwd;  // encode plan; stand-alone transform plan (all DIF, then the block permutation)
    uint64_t* tw_fwd = nullptr;  // forward roots, level-packed
    uint64_t* tw_inv = nullptr;  // inverse roots
    uint64_t* dscale = nullptr;
    std::string text;
};

namespace {

int fail(char* detail, size_t cap, hipError_t e, const char* what)
{
    if (detail && cap) snprintf(detail, cap, "%s: %s", what, hipGetErrorString(e));
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? FASTECC_E_NOMEM : FASTECC_E_DEVICE;
}

// MID takes the lowest min(n, L) levels; the rest is split into near-equal chunks of at most L levels.
void build_plans(Path* p)
{
    const int n = p->n, L = p->levels;
    const int m = std::min(n, L), rest = n - m;
    std::vector<int> chunks;
    if (rest > 0) {
        const int q = (rest + L - 1) / L;
        for (int i = 0; i < q; i++) chunks.push_back(rest / q + (i < rest % q ? 1 : 0));
    }
    p->enc.clear();
    p->fwd.clear();
    int top = n;
    for (int c : chunks) {
        p->enc.push_back(Pass{MODE_DIF, c, top - c, false});
        top -= c;
    }
    p->fwd = p->enc;
    p->fwd.push_back(Pass{MODE_DIF, m, 0, true});
    p->enc.push_back(Pass{MODE_MID, m, 0, false});
    for (size_t i = chunks.size(); i-- > 0;) {
        p->enc.push_back(Pass{MODE_DIT, chunks[i], top, false});
        top += chunks[i];
    }
    p->enc.back().canon = true;

    p->text.clear();
    char buf[32];
    for (const Pass& q : p->enc) {
        snprintf(buf, sizeof buf, "%s%s%d@%d", p->text.empty() ? "" : ",", q.mode == MODE_DIF ? "dif" : q.mode == MODE_DIT ? "dit" : "mid",
                 q.logr, q.s);
        p->text += buf;
    }
    p->text += " gf61^2";
}

// Level l is executed by a register run whose smallest stride is 2^sl[l] (see ntt_device.hpp).
std::vector<int> level_strides(const Path* p)
{
    std::vector<int> sl(p->n, 0);
    for (const Pass& q : p->enc) {
        if (q.mode == MODE_DIT) continue;
        for (int l = q.s; l < q.s + q.logr; l++) sl[l] = q.s;
    }
    return sl;
}

std::vector<uint64_t> build_level_table(int n, gf61::Elem root_of_order_N, const std::vector<int>& sl)
{
    std::vector<uint64_t> tab(2 * std::max<size_t>((size_t)1 << n, 2), 0);
    for (int l = 0; l < n; l++) {
        const uint64_t h = 1ull << l;
        const gf61::Elem root = gf61::h_pow(root_of_order_N, 1ull << (n - 1 - l));
        const int t = l - sl[l];
        const uint64_t lowmask = (1ull << sl[l]) - 1;
        gf61::Elem w{1, 0};
        for (uint64_t i = 0; i < h; i++) {
            const uint64_t e = h + (((i & lowmask) << t) | (i >> sl[l]));
            tab[2 * e] = w.re;
            tab[2 * e + 1] = w.im;
            w = gf61::h_mul(w, root);
        }
    }
    return tab;
}

int upload(uint64_t** dst, const std::vector<uint64_t>& src, char* detail, size_t cap)
{
    hipError_t e = hipSuccess;
    if (!*dst) e = hipMalloc((void**)dst, src.size() * 8);
    if (e != hipSuccess) return fail(detail, cap, e, "hipMalloc(gf61 table)");
    e = hipMemcpy(*dst, src.data(), src.size() * 8, hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail(detail, cap, e, "hipMemcpy(gf61 table)");
    return FASTECC_OK;
}

int upload_tables(Path* p, char* detail, size_t cap)
{
    const gf61::Elem wN = gf61::h_root(p->N), wNi = gf61::h_inv(wN);
    const std::vector<int> sl = level_strides(p);
    int rc = upload(&p->tw_fwd, build_level_table(p->n, wN, sl), detail, cap);
    if (rc == FASTECC_OK) rc = upload(&p->tw_inv, build_level_table(p->n, wNi, sl), detail, cap);
    return rc;
}

struct Scope {
    const LaunchHooks* h;
    hipStream_t st;
    Scope(const LaunchHooks* h_, hipStream_t st_, const char* name, uint64_t bytes) : h(h_), st(st_)
    {
        if (h) h->begin(h->user, st, name, bytes);
    }
    ~Scope()
    {
        if (h) h->end(h->user, st);
    }
};

int run_passes(Path* p, const std::vector<Pass>& plan, const uint64_t* in, uint64_t* out, const uint64_t* tw_dif,
               const uint64_t* tw_dit, hipStream_t st, const LaunchHooks* hooks)
{
    const uint64_t* src = in;
    for (const Pass& q : plan) {
        PassArgs a{};
        a.in = src;
        a.out = out;
        a.tw_dif = tw_dif;
        a.tw_dit = tw_dit;
        a.dscale = p->dscale;
        a.elems = (uint32_t)p->elems;
        a.col_chunks = (uint32_t)((p->elems + 63) / 64);
        a.items = (p->N >> q.logr) * a.col_chunks;
        a.s = q.s;
        const uint64_t blocks = (a.items + 3) / 4;
        if (blocks > 0x7FFFFFFFull) return FASTECC_E_UNSUPPORTED;
        const dim3 grid((unsigned)blocks);
        char name[32];
        snprintf(name, sizeof name, "p61_%s%d", q.mode == MODE_DIF ? "dif" : q.mode == MODE_DIT ? "dit" : "mid", q.logr);
        Scope sc(hooks, st, name, 2ull * p->N * p->elems * 16ull);
        hipError_t e;
        switch (q.logr) {
        case 1: e = launch_mode<1>(q.mode, q.canon, a, grid, st); break;
        case 2: e = launch_mode<2>(q.mode, q.canon, a, grid, st); break;
        case 3: e = launch_mode<3>(q.mode, q.canon, a, grid, st); break;
        case 4: e = launch_mode<4>(q.mode, q.canon, a, grid, st); break;
        case 5: e = launch_mode<5>(q.mode, q.canon, a, grid, st); break;
        default: return FASTECC_E_UNSUPPORTED;
        }
        if (e != hipSuccess) return fail(nullptr, 0, e, "p61 pass");
        src = out;  // after the first pass everything is in place on `out`
    }
    return FASTECC_OK;
}

}  // namespace

int create(Path** out, int n, uint64_t elems, char* detail, size_t cap)
{
    *out = nullptr;
    if (n < 1 || n > MAX_LOG2_K || elems == 0 || elems > 0x7FFFFFFFull) return FASTECC_E_UNSUPPORTED;
    Path* p = new (std::nothrow) Path();
    if (!p) return FASTECC_E_NOMEM;
    p->n = n;
    p->N = 1ull << n;
    p->elems = elems;
    build_plans(p);

    // per-block factors w_2N^i / N (RS.cpp:51-54), stored by position: position q holds coefficient bitrev(q)
    std::vector<uint64_t> dsc(2 * p->N);
    const gf61::Elem w2N = gf61::h_root(2 * p->N);
    gf61::Elem d = gf61::h_inv(gf61::Elem{p->N % gf61::P, 0});
    for (uint64_t i = 0; i < p->N; i++) {
        uint64_t r = 0;
        for (int b = 0; b < n; b++) r |= ((i >> b) & 1ull) << (n - 1 - b);
        dsc[2 * r] = d.re;
        dsc[2 * r + 1] = d.im;
        d = gf61::h_mul(d, w2N);
    }
    int rc = upload_tables(p, detail, cap);
    if (rc == FASTECC_OK) rc = upload(&p->dscale, dsc, detail, cap);
    if (rc != FASTECC_OK) {
        destroy(p);
        return rc;
    }
    *out = p;
    return FASTECC_OK;
}

void destroy(Path* p)
{
    if (!p) return;
    if (p->tw_fwd) (void)hipFree(p->tw_fwd);
    if (p->tw_inv) (void)hipFree(p->tw_inv);
    if (p->dscale) (void)hipFree(p->dscale);
    delete p;
}

int encode(Path* p, const uint64_t* data, uint64_t* parity, hipStream_t st, const LaunchHooks* hooks)
{
    // inverse roots on the way down (interpolate), forward roots on the way up (evaluate) — RS.cpp:41,63
    return run_passes(p, p->enc, data, parity, p->tw_inv, p->tw_fwd, st, hooks);
}

int ntt(Path* p, uint64_t* data, bool inverse, hipStream_t st, const LaunchHooks* hooks)
{
    const uint64_t* tw = inverse ? p->tw_inv : p->tw_fwd;
    const int rc = run_passes(p, p->fwd, data, data, tw, tw, st, hooks);
    if (rc != FASTECC_OK) return rc;
    if (p->n >= 2) {
        const uint32_t col_chunks = (uint32_t)((p->elems + 63) / 64);
        const uint64_t items = p->N * col_chunks;
        const uint64_t blocks = (items + 3) / 4;
        if (blocks > 0x7FFFFFFFull) return FASTECC_E_UNSUPPORTED;
        Scope sc(hooks, st, "p61_bitrev_rows", 2ull * p->N * p->elems * 16ull);
        hipLaunchKernelGGL(p61_bitrev_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, data, (uint32_t)p->elems, p->n, col_chunks, items);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(nullptr, 0, e, "p61_bitrev_rows");
    }
    return FASTECC_OK;
}

int count_out_of_range(Path* p, const uint64_t* data, unsigned long long* counter, hipStream_t st)
{
    const uint64_t words = 2 * p->N * p->elems;
    const unsigned blocks = (unsigned)std::min<uint64_t>((words + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(p61_count_out_of_range_kernel, dim3(blocks), dim3(256), 0, st, data, words, counter);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, 0, e, "p61_count_out_of_range");
    return FASTECC_OK;
}

int set_levels_per_pass(Path* p, int levels, char* detail, size_t cap)
{
    if (levels < 1 || levels > 5) return FASTECC_E_INVAL;
    p->levels = levels;
    build_plans(p);
    return upload_tables(p, detail, cap);
}

const char* plan_string(const Path* p) { return p->text.c_str(); }

}  // namespace p61
}  // namespace fastecc
