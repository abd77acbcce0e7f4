// stage_a_kernel, bfloat16 rows.
#define DCTFP_TEMPLATES_ONLY
#include "launch.h"

namespace dctfp_host {

#include "k_stage_a.inc"

void launch_a_bf16(const AParams& p, int vec, int n, int waves, int unroll) {
    if (vec == 8) launch_a_n<bf16_t, 8>(p, n, waves, unroll);
    else if (vec == 4) launch_a_cfg<bf16_t, 3, 4>(p, waves, unroll);  // (n = 3 fused walks only)
    else launch_a_n<bf16_t, 1>(p, n, waves, unroll);
}

}  // namespace dctfp_host
