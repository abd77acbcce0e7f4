// stage_a_kernel, float64 rows.
#define DCTFP_TEMPLATES_ONLY
#include "launch.h"

namespace dctfp_host {

#include "k_stage_a.inc"

void launch_a_f64(const AParams& p, int vec, int n, int waves, int unroll) {
    if (vec == 2) launch_a_n<double, 2>(p, n, waves, unroll);
    else launch_a_n<double, 1>(p, n, waves, unroll);
}

}  // namespace dctfp_host
