// walk_ab_kernel.
#define DCTFP_TEMPLATES_ONLY
#include "launch.h"

namespace dctfp_host {

template <typename T, int S, int G, int NT, int UNROLL>
void launch_walk_impl(const WParams& p, bool fused, bool mfma_a = false) {
    static const InvTab<3> inv = make_inv<3>();
#ifdef DCTFP_EXPERIMENTS
    if constexpr (sizeof(T) == 4 && UNROLL == 8 && G == 4 && NT == 5) {
        if (fused && mfma_a) {  // stage A on the matrix pipe (experiment of round 3: DESIGN.md section 4)
            hipLaunchKernelGGL((walk_ab_kernel<T, S, G, NT, UNROLL, true, true>), dim3(p.grid), dim3(S * 64), 0, p.stream, p.jobs, p.jobb,
                               p.walks, p.runs, p.pieces, p.stf, p.out, p.n_cols, p.ld, p.m, inv, p.degenerate);
            return;
        }
    }
#else
    (void)mfma_a;
#endif
    if constexpr (sizeof(T) == 4 && UNROLL == 8 && G == 4 && NT == 5) {
        if (p.two_source) {  // pieces that are the mean of two windows' rows (dctfp_quantize_windows): builds of their own
            if (fused)
                hipLaunchKernelGGL((walk_ab_kernel<T, S, G, NT, UNROLL, true, false, true>), dim3(p.grid), dim3(S * 64), 0, p.stream, p.jobs, p.jobb,
                                   p.walks, p.runs, p.pieces, p.stf, p.out, p.n_cols, p.ld, p.m, inv, p.degenerate);
            else
                hipLaunchKernelGGL((walk_ab_kernel<T, S, G, NT, UNROLL, false, false, true>), dim3(p.grid), dim3(S * 64), 0, p.stream, p.jobs, p.jobb,
                                   p.walks, p.runs, p.pieces, p.stf, p.out, p.n_cols, p.ld, p.m, inv, p.degenerate);
            return;
        }
    }
    if (fused)
        hipLaunchKernelGGL((walk_ab_kernel<T, S, G, NT, UNROLL, true>), dim3(p.grid), dim3(S * 64), 0, p.stream, p.jobs, p.jobb, p.walks,
                           p.runs, p.pieces, p.stf, p.out, p.n_cols, p.ld, p.m, inv, p.degenerate);
    else
        hipLaunchKernelGGL((walk_ab_kernel<T, S, G, NT, UNROLL, false>), dim3(p.grid), dim3(S * 64), 0, p.stream, p.jobs, p.jobb, p.walks,
                           p.runs, p.pieces, p.stf, p.out, p.n_cols, p.ld, p.m, inv, p.degenerate);
}

template <int S, int G>
int launch_walk_u(const WParams& p, int unroll, bool fused, bool mfma_a) {
#ifdef DCTFP_EXPERIMENTS
    // rows in flight other than 8: A/B builds only (option ab_unroll, which libdctfp.so does not know)
    if (unroll == 4) launch_walk_impl<float, S, G, 5, 4>(p, fused);
    else if (unroll == 6) launch_walk_impl<float, S, G, 5, 6>(p, fused);
    else if (unroll == 12 && G == 4) launch_walk_impl<float, S, 4, 5, 12>(p, fused);
    else if (unroll == 16 && G == 4) launch_walk_impl<float, S, 4, 5, 16>(p, fused);
    else
#else
    (void)unroll;
#endif
    launch_walk_impl<float, S, G, 5, 8>(p, fused, mfma_a);
    return DCTFP_OK;
}

// Instantiated shapes: S waves cover up to 256 S channels; G = jobs per flush = 4, the rows of an MFMA tile (a flush costs
// the same MFMAs for 1..4 jobs; the LDS -- 2304 B per wave and job -- leaves room for 17 waves per CU).  G = 3 and other
// numbers of rows in flight exist in libdctfp_experiments.so only (options ab_group / ab_unroll): the product library holds
// the 3 widths x {plain, fused} x {float32, float16, bfloat16} = 18 builds it can reach.
int launch_walk(const WParams& p, int dtype, int s, int g, int unroll, bool fused, bool mfma_a, LaunchError* err) {
    if (dtype == DCTFP_F16 || dtype == DCTFP_BF16) {
        const bool h = dtype == DCTFP_F16;
        if (s == 3) h ? launch_walk_impl<_Float16, 3, 4, 5, 8>(p, fused) : launch_walk_impl<bf16_t, 3, 4, 5, 8>(p, fused);
        else if (s == 5) h ? launch_walk_impl<_Float16, 5, 4, 5, 8>(p, fused) : launch_walk_impl<bf16_t, 5, 4, 5, 8>(p, fused);
        else h ? launch_walk_impl<_Float16, 10, 4, 5, 8>(p, fused) : launch_walk_impl<bf16_t, 10, 4, 5, 8>(p, fused);
        return DCTFP_OK;
    }
    if (p.m > 80) {   // six column groups (80 < m <= 96: PROST's [3, 85]): float32 rows, four jobs per flush, eight rows in flight
        if (g != 4 || p.two_source) return launch_fail(err, DCTFP_ERR_INVALID, "walk kernel: m = %d with %d jobs per flush / two-source pieces", p.m, g);
        if (s == 3) launch_walk_impl<float, 3, 4, 6, 8>(p, fused);
        else if (s == 5) launch_walk_impl<float, 5, 4, 6, 8>(p, fused);
        else launch_walk_impl<float, 10, 4, 6, 8>(p, fused);
        return DCTFP_OK;
    }
    if (s == 3 && g == 4) return launch_walk_u<3, 4>(p, unroll, fused, mfma_a);
    if (s == 5 && g == 4) return launch_walk_u<5, 4>(p, unroll, fused, mfma_a);
    if (s == 10 && g == 4) return launch_walk_u<10, 4>(p, unroll, fused, mfma_a);
#ifdef DCTFP_EXPERIMENTS
    if (s == 3 && g == 3) return launch_walk_u<3, 3>(p, unroll, fused, mfma_a);
    if (s == 5 && g == 3) return launch_walk_u<5, 3>(p, unroll, fused, mfma_a);
    if (s == 10 && g == 3) return launch_walk_u<10, 3>(p, unroll, fused, mfma_a);
#endif
    return launch_fail(err, DCTFP_ERR_INVALID, "walk kernel: no build for %d waves x %d jobs per flush", s, g);
}


}  // namespace dctfp_host
