// The read-only ceiling of the walk kernel's three shapes: workgroups of S waves stream a [rows][D float32] matrix the way the
// walk kernel does (wave w: channels [128 w, 128 w + 128) and their mirror images, lanes past D / 2 masked off; 8 rows in
// flight, nt), with the residency the kernel has (5 / 3 / 1 workgroups per CU at D = 640 / 1280 / 2560).  Nothing is
// computed; optionally every workgroup pauses after `job_rows` rows for `pause_us` (a stand-in for epilogue + flush: no loads
// in flight on that workgroup).  Prints GB/s.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_shapes tools/microbench/hbm_shapes.hip && /tmp/hbm_shapes
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ inline v4f load16(__amdgpu_buffer_rsrc_t rs, int lane_bytes, int uniform_bytes) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    const v4i r = __builtin_amdgcn_raw_buffer_load_b128(rs, lane_bytes, uniform_bytes, 2);
    return __builtin_bit_cast(v4f, r);
}

template <int D, int S, int WORK, int FRAG>
__global__ __launch_bounds__(S * 64, S >= 10 ? 3 : 4) void read_kernel(const float* __restrict__ x, int rows_per_wg, int job_rows, int pause_ticks,
                                                       float* __restrict__ sink, const double* __restrict__ frag_tab) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const float* base = x + (size_t)blockIdx.x * rows_per_wg * D;
    const __amdgpu_buffer_rsrc_t rows = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
    const int pair0 = wave * 128 + 4 * (lane & 31);
    const bool pad = pair0 >= D / 2;
    const int colc = pad ? 0 : (lane >= 32 ? D - 4 - pair0 : pair0);
    float acc = 0.f;
    double f[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) f[k][q] = 0.0;
    double zacc[15];
#pragma unroll
    for (int i = 0; i < 15; ++i) zacc[i] = 0.0;
    const double c0 = 0.3 + 1e-9 * lane, c1 = 0.5, c2 = 0.7, c3 = 0.9, ref = 0.25;
    int since = 0;
    for (int r = 0; r + 8 <= rows_per_wg; r += 8) {
        if (!pad) {
            v4f v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = load16(rows, colc * 4, (r + u) * D * 4);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (WORK == 0) {
                    acc += v[u][0];
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double d = (double)v[u][q] - ref;
                        f[0][q] = fma(c0, d, f[0][q]);
                        f[1][q] = fma(c1, d, f[1][q]);
                        f[2][q] = fma(c2, d, f[2][q]);
                        f[3][q] = fma(c3, d, f[3][q]);
                    }
                }
            }
        }
        since += 8;
        if (job_rows > 0 && since >= job_rows) {  // every wave of the workgroup idles together, like a flush
            since = 0;
            if (FRAG == 0) {
                __syncthreads();
                const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
                while ((int)(__builtin_amdgcn_s_memrealtime() - t0) < pause_ticks) __builtin_amdgcn_s_sleep(8);
                __syncthreads();
            } else {
                // a flush's contraction: 32 k-steps x 5 fragments of 512 bytes per wave from a table every workgroup reads
                // (L2), 4 k-steps in flight, 15 MFMAs per k-step; then the two barriers
                const __amdgpu_buffer_rsrc_t ft = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(frag_tab) + (size_t)wave * 32 * 5 * 64, 0, 0x7fffffff, 0x00020000);
                typedef unsigned v2u __attribute__((ext_vector_type(2)));
                double bq[4][5];
#pragma unroll
                for (int st = 0; st < 4; ++st)
#pragma unroll
                    for (int c = 0; c < 5; ++c) bq[st][c] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(ft, lane * 8, (st * 5 + c) * 512, 0));
                for (int q = 0; q < 8; ++q) {
#pragma unroll
                    for (int st = 0; st < 4; ++st) {
                        const double y = f[st][0] + 1.0;
#pragma unroll
                        for (int j = 0; j < 3; ++j)
#pragma unroll
                            for (int c = 0; c < 5; ++c) zacc[j * 5 + c] = __builtin_amdgcn_mfma_f64_4x4x4f64(y, bq[st][c], zacc[j * 5 + c], 0, 0, 0);
                        const int nxt = (q * 4 + st + 4) & 31;
#pragma unroll
                        for (int c = 0; c < 5; ++c) bq[st][c] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(ft, lane * 8, (nxt * 5 + c) * 512, 0));
                    }
                }
                __syncthreads();
                __syncthreads();
            }
        }
    }
    double t = acc;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) t += f[k][q];
#pragma unroll
    for (int i = 0; i < 15; ++i) t += zacc[i];
    if (t == 1.2345e300) sink[blockIdx.x] = (float)t;
}

template <int D, int S, int WORK = 0, int FRAG = 0>
void run(const float* x, size_t bytes, float* sink, int per_cu, int job_rows, double pause_us, const double* frag_tab = nullptr) {
    const int wgs = 256 * per_cu * 4;  // four rounds of workgroups
    const int rows_per_wg = (int)(bytes / 4 / D / wgs) / 8 * 8;
    const size_t used = (size_t)wgs * rows_per_wg * D * 4;
    const size_t lds = 160 * 1024 / per_cu - 1024;
    auto launch = [&]() { hipLaunchKernelGGL((read_kernel<D, S, WORK, FRAG>), dim3(wgs), dim3(S * 64), lds, 0, x, rows_per_wg, job_rows, (int)(pause_us * 100), sink, frag_tab); };
    launch();
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return; }
    const auto t0 = std::chrono::steady_clock::now();
    int n = 0;
    double el = 0.0;
    do {
        for (int i = 0; i < 5; ++i) launch();
        (void)hipDeviceSynchronize();
        n += 5;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (el < 1.5);
    printf("D = %4d, %2d waves per workgroup, %d workgroups per CU, %s, %s %5.1f us every %3d rows: %.3f ms per pass = %.0f GB/s\n", D, S, per_cu,
           WORK ? "fused walk's float64 work per element" : "no arithmetic", FRAG ? "a flush's fragment loads + MFMAs instead of a pause of" : "pause", pause_us,
           job_rows, 1e3 * el / n, (double)used * n / el / 1e9);
    fflush(stdout);
}

int main() {
    const size_t bytes = (size_t)30 << 30;
    float *x = nullptr, *sink = nullptr;
    if (hipMalloc(&x, bytes) != hipSuccess || hipMalloc(&sink, 1 << 20) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    (void)hipMemset(x, 1, bytes);
    run<640, 3>(x, bytes, sink, 5, 0, 0);
    run<1280, 5>(x, bytes, sink, 3, 0, 0);
    run<2560, 10>(x, bytes, sink, 1, 0, 0);
    // jobs of 112 rows (c5: 26 us of stream), a quarter of a 40-us flush per job; D = 2560: 88 rows, 3.75 us per job
    run<640, 3>(x, bytes, sink, 5, 112, 10.0);
    run<640, 3>(x, bytes, sink, 5, 448, 40.0);
    run<2560, 10>(x, bytes, sink, 1, 88, 3.75);
    run<2560, 10>(x, bytes, sink, 1, 352, 15.0);
    run<1280, 5>(x, bytes, sink, 3, 496, 40.0);
    // towards the real kernel at D = 2560 and D = 640: + the arithmetic, + the flush's memory traffic and MFMAs
    double* ft = nullptr;
    if (hipMalloc(&ft, (size_t)10 * 32 * 5 * 64 * 8 + 4096) != hipSuccess) return 1;
    (void)hipMemset(ft, 0, (size_t)10 * 32 * 5 * 64 * 8 + 4096);
    run<2560, 10, 1, 0>(x, bytes, sink, 1, 0, 0);
    run<2560, 10, 1, 0>(x, bytes, sink, 1, 352, 15.0);
    run<2560, 10, 0, 1>(x, bytes, sink, 1, 352, 15.0, ft);
    run<2560, 10, 1, 1>(x, bytes, sink, 1, 352, 15.0, ft);
    run<640, 3, 1, 0>(x, bytes, sink, 5, 0, 0);
    run<640, 3, 1, 0>(x, bytes, sink, 5, 448, 40.0);
    run<640, 3, 1, 1>(x, bytes, sink, 5, 448, 40.0, ft);
    run<1280, 5, 1, 1>(x, bytes, sink, 3, 496, 40.0, ft);
    return 0;
}
