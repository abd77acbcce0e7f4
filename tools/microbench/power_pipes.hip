// Microbenchmark behind DESIGN.md section 8: the fused walks run the package into its power limit.  Does the SAME float64
// multiply-add work cost less power on the matrix pipe than on the vector pipe?  Every variant streams a buffer far larger
// than the Infinity Cache at full rate in the walk kernel's shape (a wave reads 1-KiB rows, 8 in flight, `nt`) and does, per
// 16 bytes a lane loads (4 float32 channels):
//     read    nothing but a checksum add
//     valu16  4 cvt + 4 sub + 8 v_fma_f64          (the plain walk: 2 coefficients per channel)
//     valu24  4 cvt + 4 sub + 16 v_fma_f64         (the fused walk: part + whole protein)
//     mfma    4 cvt + 4 sub + 4 v_mfma_f64_4x4x4_4b (256 multiply-adds each = the 16 FMAs x 64 lanes of valu24, on the
//             matrix pipe; operand layout is NOT the DCT's -- this measures power and rate, not a result)
// The program runs one variant for the given number of seconds and prints GB/s; tools/power_pipes.sh samples
// `rocm-smi --showclocks --showpower` beside it.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/power_pipes tools/microbench/power_pipes.hip && /tmp/power_pipes valu24 6
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>  // 0 read, 1 valu16, 2 valu24, 3 mfma
__global__ __launch_bounds__(320) void stream_kernel(const float* __restrict__ x, size_t rows_per_wave, double c0, double c1, double c2,
                                                      double c3, double* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const float* base = x + wave * rows_per_wave * 256 + lane * 4;
    typedef const v4f __attribute__((address_space(1))) * GP;
    double acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0;
    const double ref = (double)lane;
    for (size_t r = 0; r + 8 <= rows_per_wave; r += 8) {
        v4f v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load((GP)(uintptr_t)(base + (r + u) * 256));
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) {
                acc[u & 3] += (double)v[u][0];
            } else {
                double d[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) d[q] = (double)v[u][q] - ref;
                if (MODE == 1 || MODE == 2) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[q] = fma(c0, d[q], acc[q]);
                        acc[4 + q] = fma(c1, d[q], acc[4 + q]);
                        if (MODE == 2) {
                            acc[8 + q] = fma(c2, d[q], acc[8 + q]);
                            acc[12 + q] = fma(c3, d[q], acc[12 + q]);
                        }
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(c0, d[q], acc[q], 0, 0, 0);
                }
            }
        }
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (s == 1.2345e300) sink[wave] = s;  // (never true: keeps the arithmetic alive)
}

// ESM-like values (a zero buffer would flatter every variant: multipliers and data lines that see zeros do not toggle)
__global__ void fill_kernel(float* __restrict__ x, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t h = i * 0x9E3779B97F4A7C15ull + 0x1234567;
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        const float u = (float)(h & 0xffffff) * (1.0f / 16777216.0f), v = (float)((h >> 24) & 0xffffff) * (1.0f / 16777216.0f);
        x[i] = 5.0f * (u + v - 1.0f) * (1.0f + (float)((i >> 2) & 63) * 0.1f) + (float)((i & 255) % 7);
    }
}

int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "valu24";
    const double seconds = argc > 2 ? atof(argv[2]) : 6.0;
    const int m = !strcmp(mode, "read") ? 0 : (!strcmp(mode, "valu16") ? 1 : (!strcmp(mode, "valu24") ? 2 : 3));
    const int wgs = 256 * 3, waves_per_wg = 5;                   // the walk kernel's residency at D = 1280
    const size_t rows_per_wave = 8192;                            // 8 MiB per wave, 31.5 GB in all
    const size_t n_waves = (size_t)wgs * waves_per_wg;
    const size_t bytes = n_waves * rows_per_wave * 1024;
    float* x = nullptr;
    double* sink = nullptr;
    if (hipMalloc(&x, bytes) != hipSuccess || hipMalloc(&sink, n_waves * sizeof(double)) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipLaunchKernelGGL(fill_kernel, dim3(256 * 16), dim3(256), 0, 0, x, bytes / sizeof(float));
    (void)hipDeviceSynchronize();
    auto launch = [&]() {
        switch (m) {
            case 0: hipLaunchKernelGGL(stream_kernel<0>, dim3(wgs), dim3(320), 0, 0, x, rows_per_wave, 0.3, 0.5, 0.7, 0.9, sink); break;
            case 1: hipLaunchKernelGGL(stream_kernel<1>, dim3(wgs), dim3(320), 0, 0, x, rows_per_wave, 0.3, 0.5, 0.7, 0.9, sink); break;
            case 2: hipLaunchKernelGGL(stream_kernel<2>, dim3(wgs), dim3(320), 0, 0, x, rows_per_wave, 0.3, 0.5, 0.7, 0.9, sink); break;
            default: hipLaunchKernelGGL(stream_kernel<3>, dim3(wgs), dim3(320), 0, 0, x, rows_per_wave, 0.3, 0.5, 0.7, 0.9, sink); break;
        }
    };
    launch();
    (void)hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    int n = 0;
    double el = 0.0;
    do {
        for (int i = 0; i < 10; ++i) launch();
        (void)hipDeviceSynchronize();
        n += 10;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (el < seconds);
    printf("%-7s %d launches in %.2f s: %.3f ms per launch = %.0f GB/s\n", mode, n, el, 1e3 * el / n, (double)bytes * n / el / 1e9);
    (void)hipFree(x);
    (void)hipFree(sink);
    return 0;
}
