// walk_gen_kernel.
#define DCTFP_TEMPLATES_ONLY
#include "launch.h"

namespace dctfp_host {

template <typename T, int N, int VEC, bool FUSED, int NTC>
int launch_gen_ntc(const GParams& p, LaunchError* err) {
    static const InvTab<N> inv = make_inv<N>();
    static bool attr_set = false;
    if (!attr_set) {  // dynamic LDS above 64 KB has to be asked for, once per kernel
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&walk_gen_kernel<T, N, VEC, FUSED, NTC>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kGenLdsBudget + 1024));
        if (e != hipSuccess) return launch_fail(err, DCTFP_ERR_HIP, "hipFuncSetAttribute(walk_gen_kernel): %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL((walk_gen_kernel<T, N, VEC, FUSED, NTC>), dim3(p.grid), dim3(p.waves * 64), p.lds_bytes, p.stream, p.jobs, p.jobb, p.walks, p.runs,
                       p.pieces, p.stp, p.out, p.n_cols, p.ld, p.m, p.n_slots, inv, p.degenerate);
    return DCTFP_OK;
}

template <typename T, int N, int VEC, bool FUSED = false>
int launch_gen_impl(const GParams& p, LaunchError* err) {   // (m <= 64: the build that holds four column groups of fragments per step)
    return p.m <= 64 ? launch_gen_ntc<T, N, VEC, FUSED, 4>(p, err) : launch_gen_ntc<T, N, VEC, FUSED, 8>(p, err);
}

template <typename T, int VEC>
int launch_gen_n(const GParams& p, int n, LaunchError* err) {
    if (p.fused) {   // fused walks: builds for n <= 5 (gen_fused_shape)
        switch (n) {
            case 2: return launch_gen_impl<T, 2, VEC, true>(p, err);
            case 3: return launch_gen_impl<T, 3, VEC, true>(p, err);
            case 4: return launch_gen_impl<T, 4, VEC, true>(p, err);
            case 5: return launch_gen_impl<T, 5, VEC, true>(p, err);
            default: return launch_fail(err, DCTFP_ERR_INVALID, "walk_gen_kernel: fused walks at n = %d", n);
        }
    }
    switch (n) {
        case 2: return launch_gen_impl<T, 2, VEC>(p, err);
        case 3: return launch_gen_impl<T, 3, VEC>(p, err);
        case 4: return launch_gen_impl<T, 4, VEC>(p, err);
        case 5: return launch_gen_impl<T, 5, VEC>(p, err);
        case 6: return launch_gen_impl<T, 6, VEC>(p, err);
        case 7: return launch_gen_impl<T, 7, VEC>(p, err);
        case 8: return launch_gen_impl<T, 8, VEC>(p, err);
        default: return launch_fail(err, DCTFP_ERR_INVALID, "walk_gen_kernel: n = %d", n);
    }
}

int launch_gen(const GParams& p, int dtype, int vec, int n, LaunchError* err) {
    if (dtype == DCTFP_F32) return vec == 4 ? launch_gen_n<float, 4>(p, n, err) : launch_gen_n<float, 1>(p, n, err);
    if (dtype == DCTFP_F64) return vec == 2 ? launch_gen_n<double, 2>(p, n, err) : launch_gen_n<double, 1>(p, n, err);
    return launch_fail(err, DCTFP_ERR_INVALID, "walk_gen_kernel: float32 or float64 rows");
}


}  // namespace dctfp_host
