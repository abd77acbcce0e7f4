// stage_a_kernel, bfloat16 rows.
#define DCTFP_TEMPLATES_ONLY
#include "launch.h"

namespace dctfp_host {

template <typename T, int N, int VEC, int WAVES, int UNROLL>
void launch_a_impl(const AParams& p) {
    static const InvTab<N> inv = make_inv<N>();
    if (p.fused)
        hipLaunchKernelGGL((stage_a_kernel<T, N, VEC, WAVES, UNROLL, true>), dim3(p.grid), dim3(WAVES * 64), 0, p.stream,
                           p.jobs, p.walks, p.pieces, p.yprime, p.job_bytes, p.packed, p.n_cols, p.ld, p.ldy, p.n_slabs,
                           inv, p.degenerate);
    else
        hipLaunchKernelGGL((stage_a_kernel<T, N, VEC, WAVES, UNROLL, false>), dim3(p.grid), dim3(WAVES * 64), 0, p.stream,
                           p.jobs, p.walks, p.pieces, p.yprime, p.job_bytes, p.packed, p.n_cols, p.ld, p.ldy, p.n_slabs,
                           inv, p.degenerate);
}

template <typename T, int N, int VEC>
void launch_a_cfg(const AParams& p, int waves, int unroll) {
    if (waves == 1) {
        if (unroll == 4) launch_a_impl<T, N, VEC, 1, 4>(p);
        else launch_a_impl<T, N, VEC, 1, 8>(p);
    } else if (waves == 2) {
        if (unroll == 4) launch_a_impl<T, N, VEC, 2, 4>(p);
        else launch_a_impl<T, N, VEC, 2, 8>(p);
    } else if (waves == 8) {
        if (unroll == 4) launch_a_impl<T, N, VEC, 8, 4>(p);
        else launch_a_impl<T, N, VEC, 8, 8>(p);
    } else if (waves == 16) {
        if (unroll == 4) launch_a_impl<T, N, VEC, 16, 4>(p);
        else launch_a_impl<T, N, VEC, 16, 8>(p);
    } else {
        if (unroll == 4) launch_a_impl<T, N, VEC, 4, 4>(p);
        else launch_a_impl<T, N, VEC, 4, 8>(p);
    }
}

template <typename T, int VEC>
void launch_a_n(const AParams& p, int n, int waves, int unroll) {
    switch (n) {
        case 2: launch_a_cfg<T, 2, VEC>(p, waves, unroll); break;
        case 3: launch_a_cfg<T, 3, VEC>(p, waves, unroll); break;
        case 4: launch_a_impl<T, 4, VEC, 4, 4>(p); break;
        case 5: launch_a_impl<T, 5, VEC, 4, 4>(p); break;
        case 6: launch_a_impl<T, 6, VEC, 4, 4>(p); break;
        case 7: launch_a_impl<T, 7, VEC, 4, 4>(p); break;
        default: launch_a_impl<T, 8, VEC, 4, 4>(p); break;
    }
}

void launch_a_bf16(const AParams& p, int vec, int n, int waves, int unroll) {
    if (vec == 8) launch_a_n<bf16_t, 8>(p, n, waves, unroll);
    else if (vec == 4) launch_a_cfg<bf16_t, 3, 4>(p, waves, unroll);  // (n = 3 fused walks only)
    else launch_a_n<bf16_t, 1>(p, n, waves, unroll);
}

}  // namespace dctfp_host
