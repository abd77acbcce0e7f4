// stage_a_kernel, float16 rows.
#define DCTFP_TEMPLATES_ONLY
#include "launch.h"

namespace dctfp_host {

#include "k_stage_a.inc"

void launch_a_f16(const AParams& p, int vec, int n, int waves, int unroll) {
    if (vec == 8) launch_a_n<_Float16, 8>(p, n, waves, unroll);
    else if (vec == 4) launch_a_cfg<_Float16, 3, 4>(p, waves, unroll);  // (n = 3 fused walks only)
    else launch_a_n<_Float16, 1>(p, n, waves, unroll);
}

}  // namespace dctfp_host
