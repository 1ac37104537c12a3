// mixed_kernels_pfa3.hip — fused_radix_kernel for q = 63 (see mixed_kernels_pfa.hip; a unit of its own for compile time).
#include "mixed_device.hpp"

namespace fastecc {

hipError_t launch_fused_pfa3(int q, int levels, bool dit, const FusedArgs& a, unsigned tiles, hipStream_t st)
{
    return q == 63 ? launch_fused_dir<63>(levels, dit, a, tiles, st) : hipErrorInvalidValue;
}

}  // namespace fastecc
