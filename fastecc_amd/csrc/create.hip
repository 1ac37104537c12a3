// create.hip — contexts: fastecc_create / fastecc_create_ex, the context behind every code family and the decoder's internal transform
// contexts.  Split from api.hip in round 6 (no change of behaviour).
#include "drivers.hpp"

using namespace fastecc;

namespace fastecc {

extern const uint32_t* const NTT_ONLY = reinterpret_cast<const uint32_t*>(~(uintptr_t)0);  // custom_factor value: see create_ntt_ctx

// Turns a fresh (2N, N) context into the power-of-two core of a transform of order q * N: the per-block factors for all
// q stripes (position j1*N + r holds coefficient q*bitrev(r) + j1 -> w_(2qN)^coefficient / (qN), RS.cpp:51-54 with qN for
// N) and the tables of the two odd-radix passes (mixed_kernels.hip).
int setup_mixed(fastecc_ctx* c, int q, uint64_t k_user, uint64_t m_user, const uint32_t* custom_factor)
{
    DeviceGuard dg(c->device);
    if (!dg.ok) return hip_fail(hipErrorInvalidDevice, "hipSetDevice");
    const uint64_t M = c->N, N1 = (uint64_t)q * M;
    const uint32_t wN1 = gf::h_root((uint32_t)N1), wN1i = gf::h_inv(wN1);
    const uint32_t inv = gf::h_inv((uint32_t)N1);
    std::vector<uint32_t> dsc(N1), twd((size_t)M * (q - 1)), twu((size_t)M * (q - 1)), dfi, dff;
    {
        std::vector<uint32_t> pw(N1);  // w_(2 N1)^j / N1 by coefficient index, or the caller's factors (a transform context)
        if (custom_factor) {
            for (uint64_t j = 0; j < N1; j++) pw[j] = gf::h_to_mont(custom_factor[j] % gf::P);
        } else {
            const uint32_t w2 = gf::h_root((uint32_t)(2 * N1));
            uint32_t d = inv;
            for (uint64_t j = 0; j < N1; j++) {
                pw[j] = gf::h_to_mont(d);
                d = gf::h_mul(d, w2);
            }
        }
        for (uint64_t j1 = 0; j1 < (uint64_t)q; j1++)
            for (uint64_t r = 0; r < M; r++) dsc[j1 * M + r] = pw[(uint64_t)q * bitrev_host((uint32_t)r, c->n) + j1];
    }
    for (uint64_t i2 = 0; i2 < M; i2++) {
        const uint32_t a = gf::h_pow(wN1, i2), b = gf::h_pow(wN1i, i2);
        uint32_t x = 1, y = 1;
        for (int j = 1; j < q; j++) {
            x = gf::h_mul(x, a);
            y = gf::h_mul(y, b);
            twu[i2 * (q - 1) + j - 1] = gf::h_to_mont(x);
            twd[i2 * (q - 1) + j - 1] = gf::h_to_mont(y);
        }
    }
    const uint32_t wq = gf::h_pow(wN1, M);
    dff = radix_dft_table(q, wq);
    dfi = radix_dft_table(q, gf::h_inv(wq));
    (void)hipFree(c->dscale);  // sized for one stripe by create_impl
    c->dscale = nullptr;
    int rc = upload_table(&c->dscale, dsc);
    if (rc == FASTECC_OK) rc = upload_table(&c->q_tw_dif, twd);
    if (rc == FASTECC_OK) rc = upload_table(&c->q_tw_dit, twu);
    if (rc == FASTECC_OK) rc = upload_table(&c->q_dft_inv, dfi);
    if (rc == FASTECC_OK) rc = upload_table(&c->q_dft_fwd, dff);
    if (rc != FASTECC_OK) return rc;
    c->q = q;
    c->K = k_user;
    c->Mu = m_user;
    c->stripe_bytes = (size_t)N1 * c->S * 4;  // the staging stripe of the host-memory calls holds all q * N blocks
    c->parity_bytes = (size_t)m_user * c->S * 4;
    build_plans(c);  // the odd-radix level joins the plan: its own two passes, or fused into the outer tiles (other run lengths)
    return upload_twiddles(c);
}

}  // namespace fastecc

extern "C" {

int fastecc_create(fastecc_ctx** out, uint64_t n, uint64_t k, uint64_t block_bytes, int field, int device)
{
    if (!out) return FASTECC_E_INVAL;
    *out = nullptr;
    const bool f61 = field == FASTECC_FIELD_GF_P61_SQUARED;
    if (field != FASTECC_FIELD_GF_FFF00001 && !f61) return FASTECC_E_UNSUPPORTED;
    if (k < 1 || n <= k || block_bytes == 0 || (block_bytes % (f61 ? 16 : 4)) != 0) return FASTECC_E_INVAL;
    // transform size: the next power of two (RS.md:23-27 "find N1 >= N ... extend input vector with zeroes")
    int lg = 1;
    while ((1ull << lg) < k && lg < 63) lg++;
    const uint64_t N1 = 1ull << lg, m = n - k;
    const bool pow2 = N1 == k;
    // parity blocks: k (the reference's configuration), 3k or 7k (further cosets), or any m <= N1: then the smallest
    // power-of-two count >= m (at least N1/16) is computed and the first m blocks are the parity
    int fold = 0, cosets = 1;
    if (pow2 && (n == 4 * k || n == 8 * k)) {
        cosets = (int)(n / k) - 1;
    } else {
        if (m > N1) return FASTECC_E_UNSUPPORTED;
        int lgm = 0;
        while ((1ull << lgm) < m) lgm++;
        fold = std::min(lg - lgm, 4);
    }
    // The 64-bit field always runs the (2 N1, N1) transform; other (n,k) of the rules above (zero extension, fewer parity blocks) work on
    // padded copies of the stripes, and parity block j is block j * 2^fold of the full parity — the same code definition as for
    // GF(0xFFF00001), without the kernels' bounds handling (RS.md:23-33 spells out exactly this: extend with zeroes, output some values).
    int p61_stride = 1;
    if (f61) {
        p61_stride = 1 << fold;
        fold = 0;
    }
    // root(2N) must exist: 2N | 2^20 (GF.md:20, RS.cpp:51); in GF(p61^2) 2N | 2^62, the bound is table memory
    if (lg > (f61 ? p61::MAX_LOG2_K : 19)) return FASTECC_E_UNSUPPORTED;
    if (!f61 && cosets > 1 && n > (1ull << 20)) return FASTECC_E_UNSUPPORTED;  // w_n must exist
    if (block_bytes / 4 > 0xFFFFFFFFull / 2) return FASTECC_E_UNSUPPORTED;
    const uint64_t n_internal = cosets > 1 ? n : N1 + (N1 >> fold);
    const int rc = create_impl(out, n_internal, N1, lg, block_bytes, field, device, fold, cosets, nullptr);
    if (rc == FASTECC_OK) {
        (*out)->K = k;
        (*out)->Mu = m;
        (*out)->p61_stride = p61_stride;
    }
    return rc;

}

int fastecc_create_ex(fastecc_ctx** out, uint64_t n, uint64_t k, uint64_t block_bytes, int field, int device, unsigned flags)
{
    if (!out) return FASTECC_E_INVAL;
    *out = nullptr;
    if (flags & ~(unsigned)(FASTECC_CODE_MIXED_RADIX | FASTECC_CODE_TOP_RADIX2 | FASTECC_CODE_MIXED_RADIX_PFA)) return FASTECC_E_INVAL;
    if (flags & FASTECC_CODE_TOP_RADIX2) {
        if (flags != FASTECC_CODE_TOP_RADIX2 || field != FASTECC_FIELD_GF_FFF00001) return FASTECC_E_UNSUPPORTED;
        int lg = 0;
        while ((1ull << lg) < k) lg++;
        if (n != 2 * k || (1ull << lg) != k || lg < 12 || lg > 19 || block_bytes == 0 || (block_bytes % 4) != 0) return FASTECC_E_UNSUPPORTED;
        const uint64_t M = k / 2;
        int rc = create_impl(out, 2 * M, M, lg - 1, block_bytes, field, device, 0, 1, nullptr);
        if (rc != FASTECC_OK) return rc;
        rc = setup_mixed(*out, 2, k, k);
        if (rc != FASTECC_OK) {
            fastecc_destroy(*out);
            *out = nullptr;
        }
        return rc;
    }
    if (!(flags & (FASTECC_CODE_MIXED_RADIX | FASTECC_CODE_MIXED_RADIX_PFA))) return fastecc_create(out, n, k, block_bytes, field, device);
    if (field != FASTECC_FIELD_GF_FFF00001) return FASTECC_E_UNSUPPORTED;
    if (k < 1 || n <= k || block_bytes == 0 || (block_bytes % 4) != 0) return FASTECC_E_INVAL;
    // transform order: the smallest q * 2^m >= k with q in {1, 3, 5, 7, 9, 13, 15} — with FASTECC_CODE_MIXED_RADIX_PFA also the products
    // of coprime factors 21 ... 117 —, m >= 1 (NTT.md:43-46: "the next divider of 0xFFF00000 is only a few percents larger than N itself");
    // w_(2 q 2^m) must exist: 2^(m+1) | 2^20
    uint64_t best = 0;
    int bq = 1, bm = 0;
    for (int q : {1, 3, 5, 7, 9, 13, 15, 21, 35, 39, 45, 63, 65, 91, 105, 117}) {
        if (q > 15 && !(flags & FASTECC_CODE_MIXED_RADIX_PFA)) break;
        for (int m = 1; m <= 19; m++) {
            const uint64_t N1 = (uint64_t)q << m;
            if (N1 >= k && (best == 0 || N1 < best)) best = N1, bq = q, bm = m;
        }
    }
    if (best == 0 || n - k > best) return FASTECC_E_UNSUPPORTED;
    if (bq == 1) return fastecc_create(out, n, k, block_bytes, field, device);
    if (block_bytes / 4 > 0xFFFFFFFFull / 2) return FASTECC_E_UNSUPPORTED;
    const uint64_t M = 1ull << bm;
    int rc = create_impl(out, 2 * M, M, bm, block_bytes, field, device, 0, 1, nullptr);
    if (rc != FASTECC_OK) return rc;
    rc = setup_mixed(*out, bq, k, n - k);
    if (rc != FASTECC_OK) {
        fastecc_destroy(*out);
        *out = nullptr;
    }
    return rc;
}

}  // extern "C"

namespace fastecc {

// Everything after argument validation.  custom_factor (k plain values by coefficient index) replaces the encoder's
// w_2k^m / k table (create_transform_ctx).
int create_impl(fastecc_ctx** out, uint64_t n, uint64_t k, int lg, uint64_t block_bytes, int field, int device, int fold,
                       int cosets, const uint32_t* custom_factor)
{
    const bool f61 = field == FASTECC_FIELD_GF_P61_SQUARED;
    int ndev = 0;
    {
        const hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev <= 0) return hip_fail(e == hipSuccess ? hipErrorNoDevice : e, "hipGetDeviceCount");
    }
    if (device < 0 || device >= ndev) return FASTECC_E_INVAL;

    fastecc_ctx* c = new (std::nothrow) fastecc_ctx();
    if (!c) return FASTECC_E_NOMEM;
    c->device = device;
    c->field = field;
    c->N = k;
    c->n = lg;
    c->S = block_bytes / 4;
    c->ld = c->S;
    c->stripe_bytes = (size_t)k * block_bytes;
    c->fold = fold;
    c->cosets = cosets;
    c->M = n - k;
    c->K = k;
    c->Mu = n - k;
    c->parity_bytes = (size_t)(n - k) * block_bytes;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->cus = cus;
        else (void)hipGetLastError();
    }
    c->classic_plan = custom_factor != nullptr;  // the decoder's transform contexts (and the stand-alone transform's): see build_plans
    if (!f61) build_plans(c);

    DeviceGuard dg(device);
    if (dg.ok && !f61) {  // the engine's code objects, loaded once per process and device (kernels.hpp)
        static std::mutex preload_mu;
        static bool preloaded[64] = {};
        std::lock_guard<std::mutex> lk(preload_mu);
        if (device < 64 && !preloaded[device]) {
            preload_pass_kernels();
            preload_tile_kernels();
            preloaded[device] = true;
        }
    }
    if (!dg.ok) {
        delete c;
        return hip_fail(hipErrorInvalidDevice, "hipSetDevice");
    }
    if (f61) {
        int rc = cosets > 1 ? p61::create_cosets(&c->p61, lg, block_bytes / 16, cosets, g_detail, sizeof g_detail)
                            : p61::create(&c->p61, lg, block_bytes / 16, g_detail, sizeof g_detail);
        if (rc == FASTECC_OK) {
            const hipError_t e = hipMalloc((void**)&c->factor, 8);  // counter of fastecc_check_range
            if (e != hipSuccess) rc = hip_fail(e, "hipMalloc(counter)");
        }
        if (rc != FASTECC_OK) {
            fastecc_destroy(c);
            return rc;
        }
        c->plan_text = p61::plan_string(c->p61);
        *out = c;
        return FASTECC_OK;
    }

    // ---- tables: per-level twiddles for the plan, and the per-block factors w_2N^i / N of RS.cpp:51-54 ----
    const uint64_t N = k;
    const bool ntt_only = custom_factor == NTT_ONLY;  // create_ntt_ctx: the stand-alone transform's passes only, no per-block factors
    if (ntt_only) custom_factor = nullptr;
    std::vector<uint32_t> dsc(ntt_only ? 0 : N * cosets);
    const uint32_t invN = gf::h_inv((uint32_t)N);
    if (custom_factor || ntt_only) c->encode_direct_max = 0;  // a transform context is not the encoder's polynomial evaluation: always the pipeline
    for (int t = 0; t < cosets && custom_factor; t++) {
        for (uint64_t i = 0; i < N; i++) {
            const uint32_t f = custom_factor[i] >= gf::P ? custom_factor[i] - gf::P : custom_factor[i];
            dsc[bitrev_host((uint32_t)i, lg)] = gf::h_mont_mul(f, gf::MONT_R2);
        }
    }
    for (int t = 0; t < cosets && !custom_factor && !ntt_only; t++) {
        // coset t: generator w_(2^j k)^c with j = floor(log2(t + 1)) + 1 and c the (t + 2 - 2^(j-1))-th odd number
        int j = 1;
        while ((1 << j) - 1 <= t) j++;
        const uint32_t cth_odd = 2u * (uint32_t)(t + 1 - (1 << (j - 1))) + 1u;
        const uint32_t gen = gf::h_pow(gf::h_root((uint32_t)(N << j)), cth_odd);
        const uint32_t gen_m = gf::h_to_mont(gen);
        uint32_t d = gf::h_to_mont(invN);  // Montgomery form throughout
        for (uint64_t i = 0; i < N; i++) {
            dsc[t * N + bitrev_host((uint32_t)i, lg)] = d;  // by position: position p holds coefficient bitrev(p)
            d = gf::h_mont_mul(d, gen_m);
        }
    }
    int rc = upload_twiddles(c);
    if (rc == FASTECC_OK && !ntt_only) rc = upload_table(&c->dscale, dsc);
    if (rc == FASTECC_OK) {
        const hipError_t e = hipMalloc((void**)&c->factor, (ntt_only ? 2 : N) * 4);
        if (e != hipSuccess) rc = hip_fail(e, "hipMalloc(factor)");
    }
    if (rc != FASTECC_OK) {
        fastecc_destroy(c);
        return rc;
    }
    *out = c;
    return FASTECC_OK;
}

int create_transform_ctx(fastecc_ctx** out, int log2k, uint64_t block_bytes, int fold, const uint32_t* factor, int device)
{
    if (!out || !factor || log2k < 1 || log2k > 20 || fold < 0 || fold > 4 || fold > log2k || block_bytes == 0 || (block_bytes % 4)) return FASTECC_E_INVAL;
    *out = nullptr;
    const uint64_t k = 1ull << log2k;
    return create_impl(out, k + (k >> fold), k, log2k, block_bytes, FASTECC_FIELD_GF_FFF00001, device, fold, 1, factor);
}

int create_ntt_ctx(fastecc_ctx** out, int log2k, uint64_t block_bytes, int device)
{
    if (!out || log2k < 1 || log2k > 20 || block_bytes == 0 || (block_bytes % 4)) return FASTECC_E_INVAL;
    *out = nullptr;
    const uint64_t k = 1ull << log2k;
    return create_impl(out, 2 * k, k, log2k, block_bytes, FASTECC_FIELD_GF_FFF00001, device, 0, 1, NTT_ONLY);
}

namespace {
// dscale[bitrev(m)] = m * scale + offset in Montgomery form; scale_mm = scale * 2^64 mod p (mul_mont divides by 2^32 once), offset_m = offset * 2^32
__global__ __launch_bounds__(256) void ramp_factor_kernel(uint32_t* __restrict__ dsc, uint32_t N, int lg, uint32_t scale_mm, uint32_t offset_m)
{
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < N) dsc[__brev(m) >> (32 - lg)] = gf::add(gf::mul_mont(m, scale_mm), offset_m);
}
}  // namespace

int create_ramp_transform_ctx(fastecc_ctx** out, int log2k, uint64_t block_bytes, int fold, uint32_t scale, int device, uint32_t offset)
{
    if (!out || log2k < 1 || log2k > 20 || fold < 0 || fold > 4 || fold > log2k || block_bytes == 0 || (block_bytes % 4)) return FASTECC_E_INVAL;
    *out = nullptr;
    const uint64_t k = 1ull << log2k;
    int rc = create_impl(out, k + (k >> fold), k, log2k, block_bytes, FASTECC_FIELD_GF_FFF00001, device, fold, 1, NTT_ONLY);
    if (rc != FASTECC_OK) return rc;
    fastecc_ctx* c = *out;
    DeviceGuard dg(device);
    hipError_t e = dg.ok ? hipMalloc((void**)&c->dscale, k * 4) : hipErrorInvalidDevice;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(ramp_factor_kernel, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, nullptr, c->dscale, (uint32_t)k, log2k,
                           gf::h_to_mont(gf::h_to_mont(scale % gf::P)), gf::h_to_mont(offset % gf::P));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) {
        rc = hip_fail(e, "create_ramp_transform_ctx");
        fastecc_destroy(c);
        *out = nullptr;
    }
    return rc;
}

int create_mixed_transform_ctx(fastecc_ctx** out, int q, int log2m, uint64_t block_bytes, const uint32_t* factor, int device)
{
    if (!out || !factor || !radix_supported(q) || log2m < 1 || log2m > 20 || block_bytes == 0 || (block_bytes % 4)) return FASTECC_E_INVAL;
    *out = nullptr;
    const uint64_t M = 1ull << log2m;
    const std::vector<uint32_t> ones((size_t)M, 1u);  // replaced by setup_mixed below
    int rc = create_impl(out, 2 * M, M, log2m, block_bytes, FASTECC_FIELD_GF_FFF00001, device, 0, 1, ones.data());
    if (rc != FASTECC_OK) return rc;
    rc = setup_mixed(*out, q, (uint64_t)q * M, (uint64_t)q * M, factor);
    if (rc != FASTECC_OK) {
        fastecc_destroy(*out);
        *out = nullptr;
    }
    return rc;
}

}  // namespace fastecc
