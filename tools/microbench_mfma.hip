// microbench_mfma.hip — issue rate of the matrix-core instructions the direct path (csrc/direct.hip) could use, measured on the device:
// every wave runs a loop of INDEPENDENT MFMAs (ACC accumulators in turn, so no instruction waits for its own result), W waves per SIMD,
// and the kernel reports shader cycles (s_memtime) per MFMA and SIMD.  One JSON line per instruction and occupancy.
//
// Why: direct_mfma_kernel runs at one v_mfma_i32_32x32x32_i8 per 64 cycles and SIMD whatever else is changed around it (profiles/r05).
// If that is the instruction's own rate, the kernel is matrix-core bound and "5 POPS of i8" is not what this chip has.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

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
typedef int v4i_acc __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef long v2l __attribute__((ext_vector_type(2)));

enum { I8_32x32x32 = 0, I8_16x16x64 = 1, BF16_32x32x16 = 2, I8_32x32x16 = 3 };

template <int OP, int ACC>
__global__ __launch_bounds__(256) void mfma_rate_kernel(uint64_t* __restrict__ cycles, int* __restrict__ sink, int iters)
{
    const int lane = threadIdx.x & 63;
    v4i a4 = {lane, lane + 1, lane + 2, lane + 3}, b4 = {lane * 3, lane * 5, lane * 7, lane * 9};
    long a1 = lane * 0x0101010101010101L, b1 = lane * 0x0301030103010301L;
    v8bf ab, bb;
    for (int i = 0; i < 8; ++i) {
        ab[i] = (__bf16)(float)(lane + i);
        bb[i] = (__bf16)(float)(lane - i);
    }
    v16i acc[ACC];
    v16f accf[ACC];
    v4i_acc acc4[ACC];
    for (int j = 0; j < ACC; ++j)
        for (int r = 0; r < 16; ++r) {
            acc[j][r] = 0;
            accf[j][r] = 0.f;
            if (r < 4) acc4[j][r] = 0;
        }
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < ACC; ++j) {
            if (OP == I8_32x32x32) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4, b4, acc[j], 0, 0, 0);
            if (OP == I8_16x16x64) acc4[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4, b4, acc4[j], 0, 0, 0);
            if (OP == BF16_32x32x16) accf[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, accf[j], 0, 0, 0);
            if (OP == I8_32x32x16) acc[j] = __builtin_amdgcn_mfma_i32_32x32x16_i8(a1, b1, acc[j], 0, 0, 0);
        }
    }
    // the results are needed: wait for them, then stop the clock
    int s = 0;
    for (int j = 0; j < ACC; ++j) s += acc[j][0] + acc4[j][0] + (int)accf[j][0];
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (s == 0x7fffffff) sink[0] = s;
    if (lane == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP, int ACC>
static void run(const char* name, int macs_per_mfma, int waves_per_simd, uint64_t* d_cycles, int* d_sink, int cus, double ghz)
{
    const int iters = 4096;
    const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD; `waves_per_simd` blocks per CU
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((mfma_rate_kernel<OP, ACC>), dim3(blocks), dim3(256), 0, nullptr, d_cycles, d_sink, 64);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((mfma_rate_kernel<OP, ACC>), dim3(blocks), dim3(256), 0, nullptr, d_cycles, d_sink, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint64_t> h((size_t)blocks * 4);
    CK(hipMemcpy(h.data(), d_cycles, h.size() * 8, hipMemcpyDeviceToHost));
    double sum = 0;
    for (uint64_t v : h) sum += (double)v;
    // s_memtime ticks at the constant 100 MHz reference on this chip family, so cycles come from the wall time and the shader clock instead;
    // the tick count is kept as a cross-check of the wall time of one wave
    const double mfma_per_simd = (double)iters * ACC * waves_per_simd;
    const double ns_per_mfma = ms * 1e6 / mfma_per_simd;
    const double tops = 2.0 * macs_per_mfma * mfma_per_simd * 4 * cus / (ms * 1e-3) / 1e12;
    printf("{\"probe\":\"mfma_rate\",\"instruction\":\"%s\",\"independent_accumulators\":%d,\"waves_per_simd\":%d,\"ms\":%.4f,\"ns_per_mfma_per_simd\":%.2f,"
           "\"cycles_per_mfma_per_simd_at_%.2f_GHz\":%.1f,\"chip_Tops_per_s\":%.0f,\"avg_memtime_ticks_per_wave\":%.0f}\n",
           name, ACC, waves_per_simd, ms, ns_per_mfma, ghz, ns_per_mfma * ghz, tops, sum / h.size());
    fflush(stdout);
}

int main(int argc, char** argv)
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate / 1e6;
    printf("{\"probe\":\"device\",\"name\":\"%s\",\"cus\":%d,\"clock_mhz\":%d}\n", prop.name, cus, prop.clockRate / 1000);
    uint64_t* d_cycles;
    int* d_sink;
    CK(hipMalloc(&d_cycles, (size_t)cus * 8 * 4 * 8));
    CK(hipMalloc(&d_sink, 64));
    for (int w = 1; w <= 2; ++w) {
        run<I8_32x32x32, 8>("v_mfma_i32_32x32x32_i8", 32 * 32 * 32, w, d_cycles, d_sink, cus, ghz);
        run<I8_32x32x16, 8>("v_mfma_i32_32x32x16_i8", 32 * 32 * 16, w, d_cycles, d_sink, cus, ghz);
        run<I8_16x16x64, 8>("v_mfma_i32_16x16x64_i8", 16 * 16 * 64, w, d_cycles, d_sink, cus, ghz);
        run<BF16_32x32x16, 8>("v_mfma_f32_32x32x16_bf16", 32 * 32 * 16, w, d_cycles, d_sink, cus, ghz);
    }
    return 0;
}
