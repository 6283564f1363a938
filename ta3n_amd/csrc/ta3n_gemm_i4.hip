// Explicit instantiations of ta3n::gemm_tiles, part 4 of 5: the register-blocked tiles and the kind-specialised kernels.
#include "ta3n_gemm_kernel.h"
namespace ta3n {
TA3N_BLOCKED_CONFIGS(TA3N_INSTANTIATE_BLOCKED)
TA3N_KIND_CONFIGS(TA3N_INSTANTIATE_KIND)
}  // namespace ta3n
