// reccut_kernel: the domain cutter's recursion on the GPU (one workgroup per protein), three LDS classes.
#define DCTFP_TEMPLATES_ONLY
#include "launch.h"
#include "reccut_kernel.hip.h"

namespace dctfp_host {

template <int CAP, int TH, int RB, int KMAX>
static int launch_class(const dctfp::CutJob* jobs, unsigned n, double cut1, double cut2, hipStream_t stream, LaunchError* err) {
    constexpr int ECAP = 6 * CAP;
    constexpr size_t bytes = dctfp::reccut_lds_bytes(CAP, ECAP, RB, KMAX, TH);
    static bool raised = false;
    if (!raised && bytes > 48 * 1024) {   // (dynamic LDS above 48 KB must be asked for once per kernel)
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&dctfp::reccut_kernel<CAP, ECAP, TH, RB, KMAX>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)bytes) != hipSuccess)
            return launch_fail(err, DCTFP_ERR_HIP, "reccut kernel: %zu bytes of LDS refused", bytes);
        raised = true;
    }
    hipLaunchKernelGGL((dctfp::reccut_kernel<CAP, ECAP, TH, RB, KMAX>), dim3(n), dim3(TH), bytes, stream, jobs, cut1, cut2);
    return DCTFP_OK;
}

static constexpr int kClassCap[kCutClasses] = {512, 1024, 1536, 2048};

int reccut_class_of(int n_res, int64_t n_contacts) {
    for (int c = 0; c < kCutClasses; ++c) {
        const int cap = kClassCap[c];
        if (n_res <= cap && n_contacts + 3ll * n_res <= 6ll * cap) return c;
    }
    return kCutClasses - 1;   // (the largest class writes status -1 for what it cannot hold)
}

int launch_reccut(int cls, const dctfp::CutJob* jobs, unsigned n, double cut1, double cut2, hipStream_t stream, LaunchError* err) {
    if (n == 0) return DCTFP_OK;
    // (class: residues, threads, rows of the byte tile, bands of the scan -- KMAX - 2 start vectors of CAP ints of LDS beside the
    //  one that lives in `pos`: 49 / 80 / 119 / 155 KB per workgroup, i.e. three / two / one / one per CU)
    if (cls == 0) return launch_class<512, 512, 32, 4>(jobs, n, cut1, cut2, stream, err);
    if (cls == 1) return launch_class<1024, 1024, 16, 4>(jobs, n, cut1, cut2, stream, err);
    if (cls == 2) return launch_class<1536, 1024, 16, 4>(jobs, n, cut1, cut2, stream, err);
    return launch_class<2048, 1024, 16, 4>(jobs, n, cut1, cut2, stream, err);   // (155 KB of LDS)
}

}  // namespace dctfp_host
