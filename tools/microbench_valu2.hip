// microbench_valu2.hip — per-instruction VALU cost on gfx950 with the loop overhead amortised:
// 64 instructions per iteration over 16 independent registers, 8 waves per SIMD.
// Reports cycles per wave-instruction per SIMD assuming the clock printed by rocm (GRBM not read here):
//   cycles = (SIMDs * clock * time) / (waves * instructions)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define REP64(X) REP16(X) REP16(X) REP16(X) REP16(X)

#define DEF_KERNEL(NAME, ASMSTR, ...)                                                         \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, int iters, uint32_t k, uint32_t k2) \
    {                                                                                          \
        uint32_t x[16];                                                                        \
        uint64_t y[16];                                                                        \
        _Pragma("unroll") for (int i = 0; i < 16; i++) { x[i] = threadIdx.x * 16 + i + k; y[i] = x[i]; } \
        uint32_t sk = __builtin_amdgcn_readfirstlane(k2);                                      \
        for (int it = 0; it < iters; it++) {                                                   \
            _Pragma("unroll") for (int r = 0; r < 4; r++) {                                    \
                _Pragma("unroll") for (int i = 0; i < 16; i++) { asm volatile(ASMSTR : __VA_ARGS__); } \
            }                                                                                  \
        }                                                                                      \
        uint32_t acc = 0;                                                                      \
        _Pragma("unroll") for (int i = 0; i < 16; i++) acc ^= x[i] ^ (uint32_t)y[i] ^ (uint32_t)(y[i] >> 32); \
        if (acc == 0x12345678u) out[threadIdx.x] = acc + sk;                                   \
    }

DEF_KERNEL(k_add_e32, "v_add_u32_e32 %0, %1, %0", "+v"(x[i]) : "v"(k))
DEF_KERNEL(k_add_e64, "v_add_u32_e64 %0, %1, %0", "+v"(x[i]) : "s"(sk))
DEF_KERNEL(k_add_lit, "v_add_u32_e32 %0, 0xfff00001, %0", "+v"(x[i]) : "v"(k))
DEF_KERNEL(k_addco_e32, "v_add_co_u32_e32 %0, vcc, %1, %0", "+v"(x[i]) : "v"(k) : "vcc")
DEF_KERNEL(k_addco_e64, "v_add_co_u32_e64 %0, s[10:11], %1, %0", "+v"(x[i]) : "v"(k) : "s10", "s11")
DEF_KERNEL(k_subco_e32, "v_sub_co_u32_e32 %0, vcc, %0, %1", "+v"(x[i]) : "v"(k) : "vcc")
DEF_KERNEL(k_cnd_e32, "v_cndmask_b32_e32 %0, %0, %1, vcc", "+v"(x[i]) : "v"(k) : "vcc")
DEF_KERNEL(k_cnd_e64, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]", "+v"(x[i]) : "v"(k) : "s10", "s11")
DEF_KERNEL(k_and, "v_and_b32_e32 %0, %1, %0", "+v"(x[i]) : "v"(k))
DEF_KERNEL(k_xor, "v_xor_b32_e32 %0, %1, %0", "+v"(x[i]) : "v"(k))
DEF_KERNEL(k_mov, "v_mov_b32_e32 %0, %1", "+v"(x[i]) : "v"(k))
DEF_KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 20, %1", "+v"(x[i]) : "v"(k))
DEF_KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %1", "+v"(x[i]) : "v"(k))
DEF_KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %1, %1 bitop3:0xe4", "+v"(x[i]) : "v"(k))
DEF_KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1", "+v"(x[i]) : "v"(k))
DEF_KERNEL(k_mul_hi, "v_mul_hi_u32 %0, %0, %1", "+v"(x[i]) : "v"(k))
DEF_KERNEL(k_mad64, "v_mad_u64_u32 %0, s[10:11], %1, %2, %0", "+v"(y[i]) : "v"(x[i]), "v"(k) : "s10", "s11")
DEF_KERNEL(k_mad64_s, "v_mad_u64_u32 %0, s[10:11], %1, %2, 0", "+v"(y[i]) : "v"(x[i]), "s"(sk) : "s10", "s11")
DEF_KERNEL(k_mul24, "v_mul_u32_u24_e32 %0, %0, %1", "+v"(x[i]) : "v"(k))
DEF_KERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %1, %1", "+v"(x[i]) : "v"(k))
DEF_KERNEL(k_addc, "v_addc_co_u32_e32 %0, vcc, %0, %1, vcc", "+v"(x[i]) : "v"(k) : "vcc")
DEF_KERNEL(k_lshl_add64, "v_lshl_add_u64 %0, %0, 0, %0", "+v"(y[i]) : "v"(k))
DEF_KERNEL(k_perm32, "v_permlane32_swap_b32_e32 %0, %1", "+v"(x[i]), "+v"(x[(i + 1) & 15]) : )
DEF_KERNEL(k_sub_e32, "v_sub_u32_e32 %0, %0, %1", "+v"(x[i]) : "v"(k))
DEF_KERNEL(k_min, "v_min_u32_e32 %0, %0, %1", "+v"(x[i]) : "v"(k))

typedef void (*kern_t)(uint32_t*, int, uint32_t, uint32_t);

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 8;
    const double simds = prop.multiProcessorCount * 4.0;
    const double clock = prop.clockRate * 1e3;
    uint32_t* d;
    CK(hipMalloc(&d, 4096));
    struct { const char* name; kern_t k; } list[] = {
        {"v_add_u32_e32", k_add_e32}, {"v_add_u32_e64(sgpr)", k_add_e64}, {"v_add_u32_e32+literal", k_add_lit}, {"v_sub_u32_e32", k_sub_e32},
        {"v_add_co_u32_e32(vcc)", k_addco_e32}, {"v_add_co_u32_e64(sgpr pair)", k_addco_e64}, {"v_sub_co_u32_e32", k_subco_e32},
        {"v_addc_co_u32_e32", k_addc}, {"v_cndmask_b32_e32(vcc)", k_cnd_e32}, {"v_cndmask_b32_e64(sgpr pair)", k_cnd_e64},
        {"v_and_b32", k_and}, {"v_xor_b32", k_xor}, {"v_min_u32", k_min}, {"v_mov_b32", k_mov}, {"v_lshl_add_u32", k_lshl_add}, {"v_add3_u32", k_add3},
        {"v_bitop3_b32", k_bitop3}, {"v_lshl_add_u64", k_lshl_add64}, {"v_mul_lo_u32", k_mul_lo}, {"v_mul_hi_u32", k_mul_hi},
        {"v_mad_u64_u32(vvv)", k_mad64}, {"v_mad_u64_u32(v,s,0)", k_mad64_s}, {"v_mul_u32_u24", k_mul24}, {"v_mad_u32_u24", k_mad24},
        {"v_permlane32_swap", k_perm32},
    };
    const int iters = 2048;
    for (auto& e : list) {
        hipEvent_t a, b;
        CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d, iters, 3u, 5u);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a, 0));
        for (int r = 0; r < 3; r++) hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d, iters, 3u, 5u);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
        const double waves = blocks * 4.0, instr = (double)iters * 64;
        const double cyc = simds * clock * (ms * 1e-3) / (waves * instr);
        printf("{\"probe\":\"valu2\",\"op\":\"%s\",\"ms\":%.4f,\"cycles_per_wave_instr_at_%.0fMHz\":%.2f,\"Tlaneops_per_s\":%.1f}\n", e.name, ms, clock / 1e6, cyc,
               waves * 64 * instr / ms / 1e9);
        fflush(stdout);
    }
    return 0;
}
