// microbench_f64.hip — could the GF(0xFFF00001) butterfly run on the FP64 pipe instead of the integer one?
//
// The encode is bound by VALU issue (profiles/r05/pmc_valu_default_plan.json): 12 integer instructions per butterfly, two of them 32 x 32
// multiplies.  v_fma_f64 is a full-rate instruction on this chip, and a double holds any integer below 2^53 exactly, so a product modulo p
// can be formed from FMAs on integer-valued doubles (w a twiddle, |w| <= p/2, wp = RN(-w 2^32 / p) stored beside it):
//     u  = fma(b, wp, M)        M = 1.5 * 2^84: the sum is rounded to a multiple of 2^32, i.e. u - M = -q 2^32 with q = rndne(b w / p)
//     Qn = u - M                exact
//     t1 = fma(b, w, Qn)        = b w - q 2^32: an integer of magnitude <= |b w| 2^-12 + 2^31, exact while that is below 2^53
//     t  = fma(Qn, C, t1)       C = -(2^20 - 1) 2^-32:  t = b w - q (2^32 - 2^20 + 1) = b w - q p,  |t| <= p/2 (+ the quotient's error)
// and the butterfly's two outputs are a + t and a - t, unreduced: values stay "lazy" (|x| grows by p/2 per level) until an exchange
// through LDS or the store, where they are reduced to (-p/2, p/2] and converted to int32.  6 FP64 instructions per butterfly plus
// the reduction every LV levels, against 12 integer ones.  This probe measures whether that is faster on the device, in a burst and
// sustained (power-capped), and checks the values against 128-bit integer arithmetic on the host.
//
//   variant "dit LV":  LV butterfly levels, then every value reduced + converted to int32 and back (the exchange / store model)
//   variant "pure":    butterflies only (a reduced every 6 levels without the conversions): the loop's own rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <vector>

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

static constexpr uint32_t P32 = 0xFFF00001u;
static constexpr double PD = 4293918721.0;
static constexpr double MAGIC = 0x1.8p84;                 // 1.5 * 2^84: ulp = 2^32
static constexpr double CFOLD = -1048575.0 / 4294967296.0;  // -(2^20 - 1) / 2^32, exact
static constexpr double P_OVER_2_32 = PD / 4294967296.0;   // exact (p has 32 bits)
static constexpr int NTW = 64;

struct Twiddle {
    double w, wp;
};

__device__ __forceinline__ double mulmod(double b, double w, double wp, double magic, double cfold)
{
    const double u = __builtin_fma(b, wp, magic);
    const double qn = u - magic;
    const double t1 = __builtin_fma(b, w, qn);
    return __builtin_fma(qn, cfold, t1);
}

// x -> the representative of x mod p in [-p/2, p/2]
__device__ __forceinline__ double reduce(double x, double npinv32, double magic, double p_over)
{
    const double u = __builtin_fma(x, npinv32, magic);
    const double qn = u - magic;  // -rndne(x / p) 2^32
    return __builtin_fma(qn, p_over, x);
}

template <int LV, bool CVT>
__global__ __launch_bounds__(256) void bfly_f64_kernel(uint32_t* out, const Twiddle* __restrict__ tw, int iters, double magic, double cfold, double npinv32,
                                                       double p_over)
{
    double a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        a[i] = (double)(int32_t)((threadIdx.x * 2654435761u + i * 40503u) % P32 - (P32 >> 1));
        b[i] = (double)(int32_t)((threadIdx.x * 40503u + i * 2654435761u + 7u) % P32 - (P32 >> 1));
    }
    for (int it = 0; it < iters; it += LV) {
#pragma unroll
        for (int l = 0; l < LV; l++) {
            const Twiddle t = tw[(it + l) & (NTW - 1)];  // uniform: scalar loads
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const double m = mulmod(b[i], t.w, t.wp, magic, cfold);
                const double x = a[i];
                a[i] = x + m;
                b[i] = x - m;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            a[i] = reduce(a[i], npinv32, magic, p_over);
            if (CVT) {
                b[i] = reduce(b[i], npinv32, magic, p_over);
                a[i] = (double)(int32_t)a[i];  // v_cvt_i32_f64 + v_cvt_f64_i32: what an exchange of 4-byte words through LDS costs
                b[i] = (double)(int32_t)b[i];
            }
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int32_t ra = (int32_t)reduce(a[i], npinv32, magic, p_over), rb = (int32_t)reduce(b[i], npinv32, magic, p_over);
        acc ^= (uint32_t)(ra < 0 ? ra + P32 : ra) ^ (uint32_t)(rb < 0 ? rb + P32 : rb);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// the same chain in exact integers
static uint32_t host_chain(unsigned tid, const std::vector<int64_t>& w, int iters)
{
    auto mod = [](__int128 x) {
        int64_t r = (int64_t)(x % (__int128)P32);
        return (uint32_t)(r < 0 ? r + P32 : r);
    };
    uint32_t acc = 0;
    for (int i = 0; i < 8; i++) {
        __int128 a = (int32_t)((tid * 2654435761u + i * 40503u) % P32 - (P32 >> 1));
        __int128 b = (int32_t)((tid * 40503u + i * 2654435761u + 7u) % P32 - (P32 >> 1));
        a = mod(a), b = mod(b);
        for (int it = 0; it < iters; it++) {
            const __int128 m = mod(b * w[it & (NTW - 1)]);
            const __int128 x = a;
            a = mod(x + m);
            b = mod(x - m);
        }
        acc ^= (uint32_t)a ^ (uint32_t)b;
    }
    return acc;
}

template <int LV, bool CVT>
static void run(const char* name, uint32_t* d_out, const Twiddle* d_tw, const std::vector<int64_t>& w, int blocks, double sustained_seconds)
{
    const int iters = 2040 / LV * LV;
    hipStream_t st = nullptr;
    const double npinv32 = -(4294967296.0 / PD);
    auto launch = [&] {
        hipLaunchKernelGGL((bfly_f64_kernel<LV, CVT>), dim3(blocks), dim3(256), 0, st, d_out, d_tw, iters, MAGIC, CFOLD, npinv32, P_OVER_2_32);
    };
    launch();
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> h(256);
    CK(hipMemcpy(h.data(), d_out, 1024, hipMemcpyDeviceToHost));
    bool ok = true;
    for (unsigned t = 0; t < 256; t += 17) ok = ok && h[t] == host_chain(t, w, iters);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < 5; r++) launch();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5;
    const double bf = (double)blocks * 256 * iters * 8;
    printf("{\"probe\":\"bfly_f64\",\"variant\":\"%s\",\"ms\":%.4f,\"Gbfly_per_s\":%.1f,\"agrees_with_128bit_integers\":%s}\n", name, ms, bf / ms / 1e6, ok ? "true" : "false");
    fflush(stdout);
    if (sustained_seconds <= 0) return;
    const auto t0 = std::chrono::steady_clock::now();
    auto elapsed = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    while (elapsed() < sustained_seconds / 2) {
        for (int i = 0; i < 50; i++) launch();
        CK(hipStreamSynchronize(st));
    }
    long launches = 0;
    char smi[256] = "";
    bool sampled = false;
    CK(hipEventRecord(e0, st));
    while (elapsed() < sustained_seconds) {
        for (int i = 0; i < 200; i++) launch();
        launches += 200;
        if (!sampled) {
            FILE* f = popen("rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Package Power|sclk' | sed -e 's/.*: //' | tr '\\n' ' '", "r");
            if (f) {
                if (!fgets(smi, sizeof smi, f)) smi[0] = 0;
                pclose(f);
            }
            sampled = true;
        }
        CK(hipStreamSynchronize(st));
    }
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    for (char* c = smi; *c; ++c)
        if (*c == '"' || *c == '\n') *c = ' ';
    printf("{\"probe\":\"bfly_f64_sustained\",\"variant\":\"%s\",\"seconds\":%.1f,\"Gbfly_per_s\":%.1f,\"rocm_smi\":\"%s\"}\n", name, ms / 1e3, bf * launches / ms / 1e6, smi);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    const double sustained = argc > 1 ? atof(argv[1]) : 0.0;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 8;  // 8 workgroups of 4 waves per CU: 8 waves per SIMD, as the integer probe (tools/microbench.hip)
    std::vector<Twiddle> tw(NTW);
    std::vector<int64_t> w(NTW);
    uint64_t s = 12345;
    for (int i = 0; i < NTW; i++) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        int64_t v = (int64_t)((s >> 20) % P32);
        if (v > (int64_t)(P32 >> 1)) v -= P32;  // balanced: |w| <= p/2
        w[i] = v;
        tw[i].w = (double)v;
        tw[i].wp = -((double)v * 4294967296.0 / PD);  // two roundings (2^-52 relative): the quotient may be off by one part in 2^17 of a unit at |b| = 2^34
    }
    Twiddle* d_tw;
    uint32_t* d_out;
    CK(hipMalloc(&d_tw, sizeof(Twiddle) * NTW));
    CK(hipMemcpy(d_tw, tw.data(), sizeof(Twiddle) * NTW, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_out, (size_t)blocks * 256 * 4));
    printf("{\"probe\":\"device\",\"name\":\"%s\",\"cus\":%d,\"clock_mhz\":%d}\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000);
    run<6, false>("pure: 6 FP64 instructions per butterfly (a reduced every 6 levels)", d_out, d_tw, w, blocks, sustained);
    run<5, true>("dit 5: 5 levels, then every value reduced and through int32 (LDS exchange / store model)", d_out, d_tw, w, blocks, sustained);
    run<3, true>("dit 3: 3 levels between reductions", d_out, d_tw, w, blocks, sustained);
    run<10, true>("dit 10: 10 levels between reductions (|b| up to 5.5 p: past the exactness bound unless it agrees)", d_out, d_tw, w, blocks, 0);
    return 0;
}
