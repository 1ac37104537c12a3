// microbench_mfma_valu_overlap.hip — which VALU instructions issue beside a running v_mfma_i32_32x32x32_i8 on the same SIMD?
// Every wave loops over 16 independent MFMAs (4 accumulators in turn) followed by / interleaved with N VALU instructions of ONE kind on registers the
// MFMAs do not touch.  Reported: time per iteration per SIMD against the MFMAs alone (16 x 32 cycles) and the VALU alone.  One JSON line per case.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

enum { K_ADD, K_XOR, K_LSHL_ADD, K_MAD64, K_MULHI, K_CNDMASK, K_COUNT };
static const char* KNAME[] = {"v_add_u32", "v_xor_b32", "v_lshl_add_u32", "v_mad_u64_u32", "v_mul_hi_u32", "v_sub_co+v_cndmask"};

// PER = VALU instructions after each MFMA (so 16 * PER per iteration); MFMA = 0 leaves the MFMAs out, VALU = 0 the VALU
template <int KIND, int PER, bool MFMA, bool VALU>
__global__ __launch_bounds__(256) void overlap_kernel(uint32_t* out, int iters, uint32_t k)
{
    const int lane = threadIdx.x & 63;
    v4i a = {lane, lane + 1, lane + 2, lane + 3}, b = {lane * 3, lane * 5, lane * 7, lane * 9};
    v16i acc[4];
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0;
    uint32_t x[8];
    uint64_t y[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 8 + i + k, y[i] = x[i];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (MFMA) acc[m & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[m & 3], 0, 0, 0);
            if (VALU) {
#pragma unroll
                for (int p = 0; p < PER; ++p) {
                    const int i = (m * PER + p) & 7;
                    if constexpr (KIND == K_ADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(k));
                    if constexpr (KIND == K_XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(k));
                    if constexpr (KIND == K_LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 20, %1" : "+v"(x[i]) : "v"(k));
                    if constexpr (KIND == K_MAD64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(y[i]) : "v"(x[i]), "v"(k) : "vcc");
                    if constexpr (KIND == K_MULHI) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(k));
                    if constexpr (KIND == K_CNDMASK)
                        asm volatile("v_sub_co_u32 %0, vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(k) : "vcc");
                }
            }
        }
    }
    uint32_t s = 0;
    for (int j = 0; j < 4; ++j) s ^= (uint32_t)acc[j][0];
    for (int i = 0; i < 8; ++i) s ^= x[i] ^ (uint32_t)y[i] ^ (uint32_t)(y[i] >> 32);
    if (s == 0x12345678u) out[threadIdx.x] = s;
}

template <int KIND, int PER, bool MFMA, bool VALU>
static double time_one(uint32_t* d, int blocks)
{
    const int iters = 2048;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((overlap_kernel<KIND, PER, MFMA, VALU>), dim3(blocks), dim3(256), 0, nullptr, d, 64, 3u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((overlap_kernel<KIND, PER, MFMA, VALU>), dim3(blocks), dim3(256), 0, nullptr, d, iters, 3u);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 3 * 1e6 / iters;  // ns per iteration of one wave
}

template <int KIND, int PER>
static void run(uint32_t* d, int cus, int wps)
{
    const int blocks = cus * wps;
    const double both = time_one<KIND, PER, true, true>(d, blocks), mf = time_one<KIND, PER, true, false>(d, blocks), va = time_one<KIND, PER, false, true>(d, blocks);
    const int n = 16 * PER * (KIND == K_CNDMASK ? 2 : 1);
    printf("{\"probe\":\"mfma_valu_overlap\",\"valu\":\"%s\",\"valu_per_mfma\":%d,\"valu_per_iteration\":%d,\"waves_per_simd\":%d,\"ns_per_iteration_per_simd\":{\"both\":%.1f,"
           "\"mfma_only\":%.1f,\"valu_only\":%.1f},\"hidden_fraction_of_the_shorter\":%.2f}\n",
           KNAME[KIND], PER * (KIND == K_CNDMASK ? 2 : 1), n, wps, both / 1 * 1.0 / 1, mf, va, (mf + va - both) / (mf < va ? mf : va));
    fflush(stdout);
}

int main(int argc, char** argv)
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t* d;
    CK(hipMalloc(&d, 4096));
    for (int wps = 1; wps <= 2; ++wps) {
        run<K_ADD, 5>(d, cus, wps);
        run<K_ADD, 10>(d, cus, wps);
        run<K_XOR, 10>(d, cus, wps);
        run<K_LSHL_ADD, 5>(d, cus, wps);
        run<K_LSHL_ADD, 10>(d, cus, wps);
        run<K_MAD64, 3>(d, cus, wps);
        run<K_MAD64, 6>(d, cus, wps);
        run<K_MULHI, 3>(d, cus, wps);
        run<K_MULHI, 6>(d, cus, wps);
        run<K_CNDMASK, 3>(d, cus, wps);
    }
    return 0;
}
