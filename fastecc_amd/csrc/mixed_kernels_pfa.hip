// mixed_kernels_pfa.hip — the odd-radix passes for the composite q = 21, 35, 39, 45, 63, 65, 91, 105, 117 (products of coprime factors from
// {3, 5, 7, 9, 13}: the prime-factor map of mixed_device.hpp; NTT.md:43-46 "PFA NTT as well as NTT kernels of orders 3,5,7,9,13 ... for
// random N the next divider of 0xFFF00000 is only a few percents larger than N itself").  A translation unit of its own: these kernels are
// long straight-line code (up to 126 values per lane) and compile beside mixed_kernels.hip (the fused kernels of q = 39, 45, 63 in two more units).
//   radix_kernel<Q, DIT, 1>:          every q above; the pass around a power-of-two pipeline (5 trips through HBM, 3 up to 2^10 blocks per stripe)
//   fused_radix_kernel<Q, A, RLOG, .> q <= 63, up to 4-6 outer levels (fused_shape_rlog): 3 trips
#include "mixed_device.hpp"

namespace fastecc {

hipError_t launch_fused_pfa(int q, int levels, bool dit, const FusedArgs& a, unsigned tiles, hipStream_t st)
{
    switch (q) {
        case 21: return launch_fused_dir<21>(levels, dit, a, tiles, st);
        case 35: return launch_fused_dir<35>(levels, dit, a, tiles, st);
        case 39:
        case 45: return launch_fused_pfa2(q, levels, dit, a, tiles, st);
        case 63: return launch_fused_pfa3(q, levels, dit, a, tiles, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_radix_pfa(int q, bool dit, const RadixArgs& a, dim3 grid, hipStream_t st)
{
    switch (q) {
        case 21: return launch_q<21, 1>(dit, a, grid, st);
        case 35: return launch_q<35, 1>(dit, a, grid, st);
        case 39: return launch_q<39, 1>(dit, a, grid, st);
        case 45: return launch_q<45, 1>(dit, a, grid, st);
        case 63: return launch_q<63, 1>(dit, a, grid, st);
        case 65: return launch_q<65, 1>(dit, a, grid, st);
        case 91: return launch_q<91, 1>(dit, a, grid, st);
        case 105: return launch_q<105, 1>(dit, a, grid, st);
        case 117: return launch_q<117, 1>(dit, a, grid, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace fastecc
