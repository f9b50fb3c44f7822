// One variant of lin_fused_kernel (lin_fused.h) per translation unit: the fully unrolled k-loops compile in parallel.
#include "lin_fused.h"

namespace kpgnn {
int lin_launch_bwd2_reduce(const LinFParams& p, hipStream_t s) { return lin_fused_launch<3, 2>(p, s); }
}  // namespace kpgnn
