// Explicit instantiations of ta3n::gemm_tiles, part 3 of 5 (ta3n_gemm_kernel.h; split so that the parts compile in parallel).
#include "ta3n_gemm_kernel.h"
namespace ta3n {
#define TA3N_PART_CONFIGS(X) X(2, 2, 1) X(2, 2, 2)
TA3N_PART_CONFIGS(TA3N_INSTANTIATE)
}  // namespace ta3n
