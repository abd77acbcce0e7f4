// Microbenchmark for "stage A of the fused walk on the matrix pipe" (DESIGN.md section 8, item 1): the SAME matrix
// [rows][1280 float32] streamed by workgroups of 5 waves in the walk kernel's access pattern (a wave owns the channels
// [128 w, 128 w + 128) and their mirror images: two 512-byte segments of every row), 8 rows in flight, with the fused
// walk's arithmetic per element done two ways:
//     valu24       one row per load instruction (lanes 0..31 the first segment, 32..63 the mirrors), per element
//                  1 cvt + 1 v_sub_f64 + 4 v_fma_f64 (coefficients from the scalar cache)            = the shipped kernel
//     mfma_direct  one load instruction = 4 consecutive rows x 64 channels (lane = (q = lane & 15, g = lane >> 4): row
//                  r + g, 16 bytes at channel 4 q of the chunk), per element 1 cvt + 1 v_cmp_neq_f32 (-> s_or: "is this
//                  channel constant") and per 4 rows x 16 channels one v_mfma_f64_4x4x4_4b whose A operand (4 outputs x 4
//                  rows, one double per lane) is loaded per 4-row step from a table in L2
//     read         loads only (valu24's shape);   read_direct   loads only (mfma_direct's shape)
// Prints GB/s; tools/stage_a_pipes.sh samples clocks and package power beside each variant.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/stage_a_pipes tools/microbench/stage_a_pipes.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int D = 1280, WAVES = 5;

__device__ inline __amdgpu_buffer_rsrc_t wave_buffer(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ inline v4f load16(__amdgpu_buffer_rsrc_t rs, int lane_bytes, int uniform_bytes) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    const v4i r = __builtin_amdgcn_raw_buffer_load_b128(rs, lane_bytes, uniform_bytes, 2);
    return __builtin_bit_cast(v4f, r);
}

template <int MODE>  // 0 read, 1 valu24, 2 read_direct, 3 mfma_direct
__global__ __launch_bounds__(WAVES * 64, 3) void stream_kernel(const float* __restrict__ x, int rows_per_wg, const double* __restrict__ tab,
                                                               double* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const float* base = x + (size_t)blockIdx.x * rows_per_wg * D;
    const __amdgpu_buffer_rsrc_t rows = wave_buffer(base);
    constexpr int LD = D * 4;
    double s = 0.0;
    if constexpr (MODE <= 1) {
        const int pair0 = wave * 128 + 4 * (lane & 31);
        const int colc = lane >= 32 ? D - 4 - pair0 : pair0;
        double acc[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[k][v] = 0.0;
        const v4f r0 = load16(rows, colc * 4, 0);
        const double ref[4] = {r0[0], r0[1], r0[2], r0[3]};
        typedef const double __attribute__((address_space(4))) * CT;
        const CT ct = (CT)(uintptr_t)tab;
        for (int r = 0; r + 8 <= rows_per_wg; r += 8) {
            v4f v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = load16(rows, colc * 4, (r + u) * LD);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (MODE == 0) {
                    acc[0][u & 3] += (double)v[u][0];
                } else {
                    const int t = (r + u) & 1023;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double d = (double)v[u][q] - ref[q];
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[k][q] = fma(ct[4 * t + k], d, acc[k][q]);
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int v = 0; v < 4; ++v) s += acc[k][v];
    } else {
        const int q = lane & 15, g = lane >> 4;
        // chunks 0, 1: the wave's first segment; chunks 2, 3: the mirror segment, lanes in descending channel order
        const int off_a = g * LD + (wave * 128 + 4 * q) * 4;                 // + 256 h
        const int off_m = g * LD + (D - 4 - wave * 128 - 64 - 4 * q) * 4;   // chunk 3; chunk 2 = + 256
        double acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0;
        v4f ref[4];
        ref[0] = load16(rows, off_a - g * LD, 0);
        ref[1] = load16(rows, off_a - g * LD + 256, 0);
        ref[2] = load16(rows, off_m - g * LD + 256, 0);
        ref[3] = load16(rows, off_m - g * LD, 0);
        uint64_t differs[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) differs[i] = 0;
        const double* ap = tab + 4 * g + (lane & 3);  // A[i = lane & 3][k = lane >> 4] of step t: tab[16 t + 4 k + i]
        for (int r = 0; r + 8 <= rows_per_wg; r += 8) {
            v4f v[2][4];
            double a[2];
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                v[st][0] = load16(rows, off_a, (r + 4 * st) * LD);
                v[st][1] = load16(rows, off_a + 256, (r + 4 * st) * LD);
                v[st][2] = load16(rows, off_m + 256, (r + 4 * st) * LD);
                v[st][3] = load16(rows, off_m, (r + 4 * st) * LD);
                a[st] = ap[16 * (((r >> 2) + st) & 255)];
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
#pragma unroll
                for (int h = 0; h < 4; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (MODE == 2) {
                            if (e == 0) acc[4 * h + (st & 1)] += (double)v[st][h][0];
                        } else {
                            differs[4 * h + e] |= __builtin_amdgcn_ballot_w64(v[st][h][e] != ref[h][e]);
                            acc[4 * h + e] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[st], (double)v[st][h][e], acc[4 * h + e], 0, 0, 0);
                        }
                    }
            }
        }
        uint64_t any = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            s += acc[i];
            any += differs[i];
        }
        if (any == 0x123456789abcull) s += 1.0;
    }
    if (s == 1.2345e300) sink[blockIdx.x] = s;  // (never true: keeps the arithmetic alive)
}

// ESM-like values (zeros would flatter every variant: multipliers and data lines that see zeros do not toggle)
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
    const char* names[4] = {"read", "valu24", "read_direct", "mfma_direct"};
    int m = 1;
    for (int i = 0; i < 4; ++i)
        if (!strcmp(mode, names[i])) m = i;
    const int wgs = 256 * 3, rows_per_wg = 8192;  // 3 workgroups of 5 waves per CU; 40 MiB per workgroup, 32.2 GB in all
    const size_t bytes = (size_t)wgs * rows_per_wg * D * 4;
    float* x = nullptr;
    double *sink = nullptr, *tab = nullptr;
    if (hipMalloc(&x, bytes) != hipSuccess || hipMalloc(&sink, wgs * sizeof(double)) != hipSuccess || hipMalloc(&tab, 4096 * sizeof(double)) != hipSuccess) {
        printf("hipMalloc failed\n");
        return 1;
    }
    double htab[4096];
    for (int i = 0; i < 4096; ++i) htab[i] = 0.9 * __builtin_cos(0.37 * i + 0.1);
    (void)hipMemcpy(tab, htab, sizeof htab, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(fill_kernel, dim3(256 * 16), dim3(256), 0, 0, x, bytes / sizeof(float));
    (void)hipDeviceSynchronize();
    auto launch = [&]() {
        switch (m) {
            case 0: hipLaunchKernelGGL(stream_kernel<0>, dim3(wgs), dim3(WAVES * 64), 0, 0, x, rows_per_wg, tab, sink); break;
            case 1: hipLaunchKernelGGL(stream_kernel<1>, dim3(wgs), dim3(WAVES * 64), 0, 0, x, rows_per_wg, tab, sink); break;
            case 2: hipLaunchKernelGGL(stream_kernel<2>, dim3(wgs), dim3(WAVES * 64), 0, 0, x, rows_per_wg, tab, sink); break;
            default: hipLaunchKernelGGL(stream_kernel<3>, dim3(wgs), dim3(WAVES * 64), 0, 0, x, rows_per_wg, tab, sink); break;
        }
    };
    launch();
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    const auto t0 = std::chrono::steady_clock::now();
    int n = 0;
    double el = 0.0;
    do {
        for (int i = 0; i < 10; ++i) launch();
        (void)hipDeviceSynchronize();
        n += 10;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (el < seconds);
    printf("%-11s %d launches in %.2f s: %.3f ms per launch = %.0f GB/s\n", names[m], n, el, 1e3 * el / n, (double)bytes * n / el / 1e9);
    (void)hipFree(x);
    (void)hipFree(sink);
    (void)hipFree(tab);
    return 0;
}
