// mixed_kernels_pfa2.hip — fused_radix_kernel for q = 39 and 45 (see mixed_kernels_pfa.hip; a unit of its own for compile time).
#include "mixed_device.hpp"

namespace fastecc {

hipError_t launch_fused_pfa2(int q, int levels, bool dit, const FusedArgs& a, unsigned tiles, hipStream_t st)
{
    switch (q) {
        case 39: return launch_fused_dir<39>(levels, dit, a, tiles, st);
        case 45: return launch_fused_dir<45>(levels, dit, a, tiles, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace fastecc
