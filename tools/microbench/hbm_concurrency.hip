// How much HBM bandwidth does the chip deliver for a given number of bytes in flight?  Workgroups of 5 waves stream a
// [rows][1280 float32] matrix in the walk kernel's access pattern (two 512-byte segments of every row per wave); the
// arguments vary the workgroups resident per CU (1, 2, 3 = 5, 10, 15 waves) and the rows a wave keeps in flight (4, 8, 16
// = 4, 8, 16 KiB).  Nothing is computed but a checksum.  Prints GB/s per combination: the curve the walk kernel's
// workloads sit on (C2: ~15 waves x 8 KiB, 77 % of the time streaming; c4: 10 waves x 8 KiB, 67 %).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_concurrency tools/microbench/hbm_concurrency.hip && /tmp/hbm_concurrency
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int D = 1280, WAVES = 5;

__device__ inline v4f load16(__amdgpu_buffer_rsrc_t rs, int lane_bytes, int uniform_bytes) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    const v4i r = __builtin_amdgcn_raw_buffer_load_b128(rs, lane_bytes, uniform_bytes, 2);
    return __builtin_bit_cast(v4f, r);
}

template <int ROWS>
__global__ __launch_bounds__(WAVES * 64) void read_kernel(const float* __restrict__ x, int rows_per_wg, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const float* base = x + (size_t)blockIdx.x * rows_per_wg * D;
    const __amdgpu_buffer_rsrc_t rows = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
    const int pair0 = wave * 128 + 4 * (lane & 31);
    const int colc = lane >= 32 ? D - 4 - pair0 : pair0;
    float acc = 0.f;
    for (int r = 0; r + ROWS <= rows_per_wg; r += ROWS) {
        v4f v[ROWS];
#pragma unroll
        for (int u = 0; u < ROWS; ++u) v[u] = load16(rows, colc * 4, (r + u) * D * 4);
#pragma unroll
        for (int u = 0; u < ROWS; ++u) acc += v[u][0];
    }
    if (acc == 1.2345e30f) sink[blockIdx.x] = acc;
}

int main() {
    const size_t total_rows = (size_t)768 * 8192;  // 32.2 GB
    const size_t bytes = total_rows * D * 4;
    float *x = nullptr, *sink = nullptr;
    if (hipMalloc(&x, bytes) != hipSuccess || hipMalloc(&sink, 4096 * sizeof(float)) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    (void)hipMemset(x, 1, bytes);
    for (int per_cu = 1; per_cu <= 3; ++per_cu)
        for (int rows = 4; rows <= 16; rows *= 2) {
            // a fixed amount of LDS per workgroup caps the residency: 160 KiB / per_cu
            const int wgs = 256 * per_cu;
            const int rows_per_wg = (int)(total_rows / wgs);
            const size_t lds = 160 * 1024 / per_cu - 1024;
            auto launch = [&]() {
                if (rows == 4) hipLaunchKernelGGL(read_kernel<4>, dim3(wgs), dim3(WAVES * 64), lds, 0, x, rows_per_wg, sink);
                else if (rows == 8) hipLaunchKernelGGL(read_kernel<8>, dim3(wgs), dim3(WAVES * 64), lds, 0, x, rows_per_wg, sink);
                else hipLaunchKernelGGL(read_kernel<16>, dim3(wgs), dim3(WAVES * 64), lds, 0, x, rows_per_wg, sink);
            };
            launch();
            if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
            const auto t0 = std::chrono::steady_clock::now();
            int n = 0;
            double el = 0.0;
            do {
                for (int i = 0; i < 5; ++i) launch();
                (void)hipDeviceSynchronize();
                n += 5;
                el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            } while (el < 1.5);
            printf("%2d waves per CU x %2d KiB in flight per wave = %4d KiB per CU: %.3f ms per pass = %.0f GB/s\n", per_cu * WAVES, rows,
                   per_cu * WAVES * rows, 1e3 * el / n, (double)bytes * n / el / 1e9);
            fflush(stdout);
        }
    return 0;
}
