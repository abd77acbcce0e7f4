// Microbenchmark: does a VALU instruction get cheaper when whole 16-lane groups of the wave are switched off (EXEC)?
// The third wave of a D = 640 workgroup of the walk kernel has lanes 16..31 and 48..63 without work.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/exec_mask tools/microbench/exec_mask.hip && /tmp/exec_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int KIND>  // 0 v_fma_f64, 1 v_cvt_f64_f32, 2 v_add_u32
__global__ __launch_bounds__(256) void k(int iters, uint64_t mask, double seed, double* out) {
    const int lane = threadIdx.x & 63;
    double a[16];
    float f[16];
    uint32_t u[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        a[i] = seed + i;
        f[i] = (float)seed + i;
        u[i] = (uint32_t)i;
    }
    if ((mask >> lane) & 1) {  // the loop runs with EXEC = mask
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a[i]) : "v"(seed));
                if (KIND == 1) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
                if (KIND == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(i));
            }
        }
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + u[i];
    if (s == 12345.678) out[blockIdx.x] = s;
}

template <int KIND>
static float run(int iters, uint64_t mask, double* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND>), dim3(1024), dim3(256), 0, 0, iters, mask, 1.0, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND>), dim3(1024), dim3(256), 0, 0, iters, mask, 1.0, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    double* out;
    hipMalloc(&out, 1 << 20);
    const int iters = 20000;
    struct M { uint64_t m; const char* name; } masks[] = {
        {0xffffffffffffffffull, "all 64 lanes"},
        {0x00000000ffffffffull, "lanes 0..31"},
        {0x0000ffff0000ffffull, "lanes 0..15 and 32..47 (the padded wave)"},
        {0x000000000000ffffull, "lanes 0..15"},
        {0x0000000000000001ull, "lane 0"},
    };
    for (auto& m : masks) {
        printf("%-44s  v_fma_f64 %7.3f ms   v_cvt_f64_f32 %7.3f ms   v_add_u32 %7.3f ms\n", m.name, run<0>(iters, m.m, out), run<1>(iters, m.m, out),
               run<2>(iters, m.m, out));
        fflush(stdout);
    }
    return 0;
}
