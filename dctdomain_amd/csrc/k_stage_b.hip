// stage_b_mfma_kernel.
#define DCTFP_TEMPLATES_ONLY
#include "launch.h"

namespace dctfp_host {

void launch_b_mfma(int nt, bool packed, unsigned grid, hipStream_t s, const char* yp, int64_t job_bytes, int64_t rows,
                   int ldy, const double* st, const JobB* jobs, int n, int m, int8_t* out) {
#define DCTFP_B_CASE(NT)                                                                                             \
    case NT:                                                                                                         \
        if (packed)                                                                                                  \
            hipLaunchKernelGGL((stage_b_mfma_kernel<NT, true>), dim3(grid), dim3(kBWaves * 64), 0, s, yp, job_bytes, rows, ldy, \
                               st, jobs, n, m, out);                                                                 \
        else                                                                                                         \
            hipLaunchKernelGGL((stage_b_mfma_kernel<NT, false>), dim3(grid), dim3(kBWaves * 64), 0, s, yp, job_bytes, rows,   \
                               ldy, st, jobs, n, m, out);                                                            \
        break;
    switch (nt) {
        DCTFP_B_CASE(1)
        DCTFP_B_CASE(2)
        DCTFP_B_CASE(3)
        DCTFP_B_CASE(4)
        DCTFP_B_CASE(5)
        DCTFP_B_CASE(6)
        DCTFP_B_CASE(7)
        default:
            if (packed)
                hipLaunchKernelGGL((stage_b_mfma_kernel<8, true>), dim3(grid), dim3(kBWaves * 64), 0, s, yp, job_bytes, rows, ldy,
                                   st, jobs, n, m, out);
            else
                hipLaunchKernelGGL((stage_b_mfma_kernel<8, false>), dim3(grid), dim3(kBWaves * 64), 0, s, yp, job_bytes, rows, ldy,
                                   st, jobs, n, m, out);
            break;
    }
#undef DCTFP_B_CASE
}


}  // namespace dctfp_host
