// Explicit instantiations of ta3n::gemm_tiles, part 5: the half-stage kernels (bf16 twins in 64-k stages).
#include "ta3n_gemm_kernel.h"
namespace ta3n {
TA3N_HS_CONFIGS(TA3N_INSTANTIATE_HS)
}  // namespace ta3n
