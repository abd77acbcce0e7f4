// Microbenchmark / layout probe for the two f64 MFMA shapes of gfx950:
//   v_mfma_f64_16x16x4_f64      (one 16x16 tile, K = 4)
//   v_mfma_f64_4x4x4_4b_f64     (four independent 4x4 blocks, K = 4)
// 1. layout: for every lane la, A = one-hot at la, B = 1 + lane  ->  which output lanes light up and whose B they show.
//    Printed as (la: [ld<-lb ...]); the A/B/D lane maps of the 4x4x4 form are read off that table.
// 2. rate: back-to-back MFMAs with independent accumulators, all CUs, 1 / 2 / 4 waves per SIMD -> cycles per instruction.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/microbench/mfma_f64_probe.hip && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void layout_4x4(double* out) {  // out[la][ld]
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la) {
        const double a = lane == la ? 1.0 : 0.0;
        const double b = 1.0 + lane;
        const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
        out[la * 64 + lane] = d;
    }
}

template <int SHAPE, int NACC>
__global__ __launch_bounds__(256) void rate_kernel(double* out, int iters, double seed) {
    double a = seed + threadIdx.x * 1e-3, b = seed * 0.5 + threadIdx.x * 1e-4;
    if constexpr (SHAPE == 16) {
        v4d acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        double s = 0;
#pragma unroll
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        double acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
        }
        double s = 0;
#pragma unroll
        for (int i = 0; i < NACC; ++i) s += acc[i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
}

template <int SHAPE, int NACC>
void rate(const char* name, int wg_per_cu, double* dout) {
    const int iters = 20000;
    const int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((rate_kernel<SHAPE, NACC>), dim3(grid), dim3(256), 0, 0, dout, 100, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((rate_kernel<SHAPE, NACC>), dim3(grid), dim3(256), 0, 0, dout, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double n_inst_per_simd = (double)iters * NACC * wg_per_cu;  // 4 waves per WG -> one per SIMD
    const double flop_per_inst = SHAPE == 16 ? 16.0 * 16 * 4 * 2 : 4 * 4.0 * 4 * 4 * 2;
    const double tf = n_inst_per_simd * 1024.0 * flop_per_inst / (ms * 1e-3) / 1e12;
    printf("%-28s waves/SIMD %d  acc %d: %8.3f ms  %6.1f TFLOP/s  %6.1f ns per instruction per SIMD (= %5.1f cycles at 2.4 GHz)\n", name,
           wg_per_cu, NACC, ms, tf, ms * 1e6 / n_inst_per_simd, ms * 1e6 / n_inst_per_simd * 2.4);
}

int main() {
    double* dout;
    hipMalloc(&dout, sizeof(double) * 256 * 8 * 256);
    std::vector<double> h(64 * 64);
    hipLaunchKernelGGL(layout_4x4, dim3(1), dim3(64), 0, 0, dout);
    hipMemcpy(h.data(), dout, sizeof(double) * 64 * 64, hipMemcpyDeviceToHost);
    printf("v_mfma_f64_4x4x4_4b_f64 layout: A one-hot at lane la, B = 1 + lane; entries 'ld<-lb'\n");
    for (int la = 0; la < 64; ++la) {
        printf("la %2d:", la);
        for (int ld = 0; ld < 64; ++ld)
            if (h[la * 64 + ld] != 0.0) printf(" %d<-%d", ld, (int)h[la * 64 + ld] - 1);
        printf("\n");
    }
    rate<16, 1>("v_mfma_f64_16x16x4_f64", 1, dout);
    rate<16, 4>("v_mfma_f64_16x16x4_f64", 1, dout);
    rate<16, 4>("v_mfma_f64_16x16x4_f64", 2, dout);
    rate<4, 1>("v_mfma_f64_4x4x4_4b_f64", 1, dout);
    rate<4, 4>("v_mfma_f64_4x4x4_4b_f64", 1, dout);
    rate<4, 8>("v_mfma_f64_4x4x4_4b_f64", 1, dout);
    rate<4, 8>("v_mfma_f64_4x4x4_4b_f64", 2, dout);
    rate<4, 8>("v_mfma_f64_4x4x4_4b_f64", 4, dout);
    return 0;
}
