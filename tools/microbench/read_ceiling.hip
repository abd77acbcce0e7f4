// Microbenchmark: what does MI355X give a kernel that ONLY reads, in the access shapes the fingerprint kernels use?
// The 8 TB/s of the data sheet is not reachable; this measures what is, so that "85 % of peak" can be read against it.
//   flat   : every workgroup reads a contiguous span, 16 B per lane, U loads in flight per wave
//   walk   : the walk kernel's shape -- a workgroup of S waves owns jobs of `rows` x (S x 1 KiB); wave w reads the
//            1 KiB segment w of every row (row stride = S KiB), U rows in flight; `jobs_per_wg` consecutive jobs
//   slab   : the two-kernel stage A shape -- workgroup = (job, 1 KiB slab), its 4 waves take rows r = wave, wave + 4, ...
// Per load the lanes do one float64 add per element (so the compiler cannot drop anything) and nothing else.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/read_ceiling tools/microbench/read_ceiling.hip && /tmp/read_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef const v4f __attribute__((address_space(1))) * GP;

template <int NT>
__device__ inline v4f ld(const float* p) {
    if (NT) return __builtin_nontemporal_load((GP)(uintptr_t)p);
    return *(GP)(uintptr_t)p;
}

template <int U, int NT>
__global__ __launch_bounds__(256) void flat_kernel(const float* __restrict__ in, int64_t floats_per_wg, double* __restrict__ out) {
    const float* base = in + (int64_t)blockIdx.x * floats_per_wg + threadIdx.x * 4;
    double acc = 0.0;
    for (int64_t o = 0; o < floats_per_wg; o += (int64_t)U * 1024) {
        v4f x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = ld<NT>(base + o + u * 1024);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += (double)x[u][0] + (double)x[u][1] + (double)x[u][2] + (double)x[u][3];
    }
    if (acc == 12345.678) out[blockIdx.x] = acc;
}

template <int S, int U, int NT>
__global__ __launch_bounds__(S * 64) void walk_kernel(const float* __restrict__ in, int rows, int jobs_per_wg, double* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t ld_f = (int64_t)S * 256;
    double acc = 0.0;
    for (int g = 0; g < jobs_per_wg; ++g) {
        const int64_t job = (int64_t)blockIdx.x * jobs_per_wg + g;
        const float* base = in + job * rows * ld_f + wave * 256 + lane * 4;
        int r = 0;
        for (; r + U <= rows; r += U) {
            v4f x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = ld<NT>(base + (int64_t)(r + u) * ld_f);
#pragma unroll
            for (int u = 0; u < U; ++u) acc += (double)x[u][0] + (double)x[u][1] + (double)x[u][2] + (double)x[u][3];
        }
        for (; r < rows; ++r) {
            v4f x = ld<NT>(base + (int64_t)r * ld_f);
            acc += (double)x[0] + (double)x[1] + (double)x[2] + (double)x[3];
        }
    }
    if (acc == 12345.678) out[blockIdx.x] = acc;
}

template <int U, int NT>
__global__ __launch_bounds__(256) void slab_kernel(const float* __restrict__ in, int rows, int n_slabs, double* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t ld_f = (int64_t)n_slabs * 256;
    const int64_t job = blockIdx.x / n_slabs;
    const int slab = blockIdx.x % n_slabs;
    const float* base = in + job * rows * ld_f + slab * 256 + lane * 4;
    double acc = 0.0;
    int r = wave;
    for (; r + 4 * (U - 1) < rows; r += 4 * U) {
        v4f x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = ld<NT>(base + (int64_t)(r + 4 * u) * ld_f);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += (double)x[u][0] + (double)x[u][1] + (double)x[u][2] + (double)x[u][3];
    }
    for (; r < rows; r += 4) {
        v4f x = ld<NT>(base + (int64_t)r * ld_f);
        acc += (double)x[0] + (double)x[1] + (double)x[2] + (double)x[3];
    }
    if (acc == 12345.678) out[blockIdx.x] = acc;
}


// The walk shape with the arithmetic of the real kernels: per element one conversion, K/2 subtractions (first-row shift)
// and K float64 FMAs into K accumulators (K = 2: plain walk, K = 4: fused walk = part + whole protein).  Dynamic LDS
// limits the workgroups per CU so that the occupancy of the real kernel (3 x 5 waves per CU) can be reproduced.
template <int S, int U, int K, int FIRST, int MIRROR = 0>
__global__ __launch_bounds__(S * 64) void walk_fma_kernel(const float* __restrict__ in, int rows, int jobs_per_wg, double* __restrict__ out,
                                                          double c0, double c1) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t ld_f = (int64_t)S * 256;
    double acc[K][4];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[k][e] = 0.0;
    double first[K / 2][4];
#pragma unroll
    for (int k = 0; k < K / 2; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) first[k][e] = c0 * (k + 1) + (double)(lane * 4 + e);
    for (int g = 0; g < jobs_per_wg; ++g) {
        const int64_t job = (int64_t)blockIdx.x * jobs_per_wg + g;
        // MIRROR: the wave owns two 512-byte segments per row, channels [128 w, 128 w + 128) and their mirror images
        const float* base = in + job * rows * ld_f + (MIRROR ? (lane < 32 ? wave * 128 + lane * 4 : S * 256 - 128 * (wave + 1) + (lane - 32) * 4)
                                                             : wave * 256 + lane * 4);
        int r = 0;
        for (; r + U <= rows; r += U) {
            v4f x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = ld<1>(base + (int64_t)(r + u) * ld_f);
            if (FIRST) __builtin_amdgcn_sched_barrier(0);  // all U loads issued before the first use; without it hipcc
                                                           // serialises load/use pairs: 1 load in flight, 66 VGPRs
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double ca = c0 + (r + u) * c1, cb = c1 - (r + u) * c0;  // stand-ins for the cosines (scalar unit)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double v = (double)x[u][e];
#pragma unroll
                    for (int k = 0; k < K / 2; ++k) {
                        const double d = v - first[k][e];
                        acc[2 * k][e] = fma(d, ca, acc[2 * k][e]);
                        acc[2 * k + 1][e] = fma(d, cb, acc[2 * k + 1][e]);
                    }
                }
            }
        }
    }
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) t += acc[k][e];
    if (t == 12345.678) out[blockIdx.x] = t + lds[threadIdx.x];
}

// Half-precision rows in the walk shape: 4 channels per lane = 8 bytes, two 256-byte mirror segments per row and wave.
typedef float v2f_ __attribute__((ext_vector_type(2)));
template <int S, int U, int K>
__global__ __launch_bounds__(S * 64) void walk_half_kernel(const float* __restrict__ in, int rows, int jobs_per_wg, double* __restrict__ out,
                                                           double c0, double c1) {
    extern __shared__ double lds[];
    typedef const v2f_ __attribute__((address_space(1))) * GP2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t ld_f = (int64_t)S * 128;  // a row of S x 256 halves = S x 128 floats' worth of bytes
    double acc[K][4];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[k][e] = 0.0;
    for (int g = 0; g < jobs_per_wg; ++g) {
        const int64_t job = (int64_t)blockIdx.x * jobs_per_wg + g;
        const float* base = in + job * rows * ld_f + (lane < 32 ? wave * 64 + lane * 2 : S * 128 - 64 * (wave + 1) + (lane - 32) * 2);
        for (int r = 0; r + U <= rows; r += U) {
            v2f_ x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = __builtin_nontemporal_load((GP2)(uintptr_t)(base + (int64_t)(r + u) * ld_f));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double ca = c0 + (r + u) * c1, cb = c1 - (r + u) * c0;
                const uint32_t w0 = __float_as_uint(x[u][0]), w1 = __float_as_uint(x[u][1]);
                const _Float16 h[4] = {__builtin_bit_cast(_Float16, (unsigned short)(w0 & 0xffff)), __builtin_bit_cast(_Float16, (unsigned short)(w0 >> 16)),
                                       __builtin_bit_cast(_Float16, (unsigned short)(w1 & 0xffff)), __builtin_bit_cast(_Float16, (unsigned short)(w1 >> 16))};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double v = (double)(float)h[e];
#pragma unroll
                    for (int k = 0; k < K / 2; ++k) {
                        const double d = v - (double)(lane + k);
                        acc[2 * k][e] = fma(d, ca, acc[2 * k][e]);
                        acc[2 * k + 1][e] = fma(d, cb, acc[2 * k + 1][e]);
                    }
                }
            }
        }
    }
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) t += acc[k][e];
    if (t == 12345.678) out[blockIdx.x] = t + lds[threadIdx.x];
}

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);      \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

int main() {
    const int64_t total_bytes = 40ll << 30;  // well past the 256 MB of infinity cache; every byte read once per launch
    float* in;
    double* out;
    CK(hipMalloc(&in, total_bytes));
    CK(hipMemset(in, 0, total_bytes));
    CK(hipMalloc(&out, 8 << 20));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto report = [&](const char* name, int64_t bytes, int reps) {
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-100s %7.3f ms  %6.0f GB/s\n", name, ms / reps, bytes / (ms / reps * 1e-3) / 1e9);
        fflush(stdout);
    };
    const int reps = 3;
    char name[128];
#define TIME(label, bytes, ...)                              \
    do {                                                     \
        __VA_ARGS__;                                         \
        CK(hipDeviceSynchronize());                          \
        CK(hipEventRecord(e0));                              \
        for (int i_ = 0; i_ < reps; ++i_) { __VA_ARGS__; }   \
        CK(hipEventRecord(e1));                              \
        CK(hipEventSynchronize(e1));                         \
        CK(hipGetLastError());                               \
        report(label, bytes, reps);                          \
    } while (0)

    // flat: spans of 1, 4 and 16 MiB per workgroup
    for (int64_t span_kib : {256ll, 1024ll, 4096ll, 16384ll}) {
        const int64_t fpw = span_kib * 256;  // floats per workgroup
        const int64_t n_wg = total_bytes / (fpw * 4);
        snprintf(name, sizeof name, "flat  span %5lld KiB/wg  U=8 nt", (long long)span_kib);
        TIME(name, n_wg * fpw * 4, hipLaunchKernelGGL((flat_kernel<8, 1>), dim3((unsigned)n_wg), dim3(256), 0, 0, in, fpw, out));
        snprintf(name, sizeof name, "flat  span %5lld KiB/wg  U=8 default policy", (long long)span_kib);
        TIME(name, n_wg * fpw * 4, hipLaunchKernelGGL((flat_kernel<8, 0>), dim3((unsigned)n_wg), dim3(256), 0, 0, in, fpw, out));
        snprintf(name, sizeof name, "flat  span %5lld KiB/wg  U=16 nt", (long long)span_kib);
        TIME(name, n_wg * fpw * 4, hipLaunchKernelGGL((flat_kernel<16, 1>), dim3((unsigned)n_wg), dim3(256), 0, 0, in, fpw, out));
    }
    // walk shape: L = 500 rows
    {
        const int rows = 500;
        {
            const int64_t job_bytes = (int64_t)rows * 5 * 1024;
            const int64_t n_jobs = total_bytes / job_bytes / 8 * 8;
            TIME("walk  S=5 (D=1280) L=500 U=8  8 jobs/wg nt", n_jobs * job_bytes,
                 hipLaunchKernelGGL((walk_kernel<5, 8, 1>), dim3((unsigned)(n_jobs / 8)), dim3(320), 0, 0, in, rows, 8, out));
            TIME("walk  S=5 (D=1280) L=500 U=16 8 jobs/wg nt", n_jobs * job_bytes,
                 hipLaunchKernelGGL((walk_kernel<5, 16, 1>), dim3((unsigned)(n_jobs / 8)), dim3(320), 0, 0, in, rows, 8, out));
            TIME("walk  S=5 (D=1280) L=500 U=4  8 jobs/wg nt", n_jobs * job_bytes,
                 hipLaunchKernelGGL((walk_kernel<5, 4, 1>), dim3((unsigned)(n_jobs / 8)), dim3(320), 0, 0, in, rows, 8, out));
            TIME("walk  S=5 (D=1280) L=500 U=8  1 job/wg  nt", n_jobs * job_bytes,
                 hipLaunchKernelGGL((walk_kernel<5, 8, 1>), dim3((unsigned)n_jobs), dim3(320), 0, 0, in, rows, 1, out));
            TIME("walk  S=5 (D=1280) L=500 U=8  8 jobs/wg default policy", n_jobs * job_bytes,
                 hipLaunchKernelGGL((walk_kernel<5, 8, 0>), dim3((unsigned)(n_jobs / 8)), dim3(320), 0, 0, in, rows, 8, out));
        }
        {
            const int64_t job_bytes = (int64_t)rows * 3 * 1024;
            const int64_t n_jobs = total_bytes / job_bytes / 8 * 8;
            TIME("walk  S=3 (D=768)  L=500 U=8  8 jobs/wg nt", n_jobs * job_bytes,
                 hipLaunchKernelGGL((walk_kernel<3, 8, 1>), dim3((unsigned)(n_jobs / 8)), dim3(192), 0, 0, in, rows, 8, out));
        }
        {
            const int64_t job_bytes = (int64_t)rows * 10 * 1024;
            const int64_t n_jobs = total_bytes / job_bytes / 8 * 8;
            TIME("walk  S=10 (D=2560) L=500 U=8 8 jobs/wg nt", n_jobs * job_bytes,
                 hipLaunchKernelGGL((walk_kernel<10, 8, 1>), dim3((unsigned)(n_jobs / 8)), dim3(640), 0, 0, in, rows, 8, out));
        }
    }
    // walk shape with short jobs: L = 110 rows (database-build mix)
    {
        const int rows = 110;
        const int64_t job_bytes = (int64_t)rows * 3 * 1024;
        const int64_t n_jobs = total_bytes / job_bytes / 8 * 8;
        TIME("walk  S=3 (D=768)  L=110 U=8  8 jobs/wg nt", n_jobs * job_bytes,
             hipLaunchKernelGGL((walk_kernel<3, 8, 1>), dim3((unsigned)(n_jobs / 8)), dim3(192), 0, 0, in, rows, 8, out));
    }
    // slab shape (stage A of the two-kernel path): D = 1280 -> 5 slabs, L = 500
    {
        const int rows = 500;
        const int64_t job_bytes = (int64_t)rows * 5 * 1024;
        const int64_t n_jobs = total_bytes / job_bytes;
        TIME("slab  D=1280 L=500 U=4 nt", n_jobs * job_bytes,
             hipLaunchKernelGGL((slab_kernel<4, 1>), dim3((unsigned)(n_jobs * 5)), dim3(256), 0, 0, in, rows, 5, out));
        TIME("slab  D=1280 L=500 U=8 nt", n_jobs * job_bytes,
             hipLaunchKernelGGL((slab_kernel<8, 1>), dim3((unsigned)(n_jobs * 5)), dim3(256), 0, 0, in, rows, 5, out));
    }

    // arithmetic + occupancy of the real walk kernels (S = 5: 3 workgroups per CU through 48 KB of LDS each)
    {
        const int rows = 496;
        const int64_t job_bytes = (int64_t)rows * 5 * 1024;
        const int64_t n_jobs = total_bytes / job_bytes / 2 * 2;
#define WF(U_, K_, F_, label)                                                                                                         \
    snprintf(name, sizeof name, "walk+fma S=5 L=496 %s lds %2d KB/wg", label, lds_kb);                                               \
    TIME(name, n_jobs* job_bytes,                                                                                                     \
         hipLaunchKernelGGL((walk_fma_kernel<5, U_, K_, F_>), dim3((unsigned)(n_jobs / 2)), dim3(320), lds_kb * 1024, 0, in, rows, 2, \
                            out, 0.5, 0.25))
        for (int lds_kb : {1, 32, 48, 64}) {  // 160 KB per CU: 1 KB -> limited by registers only; 48 -> 3 workgroups; 64 -> 2
            WF(8, 2, 1, "K=2 plain U=8  loads first ");
            WF(8, 2, 0, "K=2 plain U=8  serialised  ");
            WF(4, 4, 1, "K=4 fused U=4  loads first ");
            WF(8, 4, 1, "K=4 fused U=8  loads first ");
            WF(16, 4, 1, "K=4 fused U=16 loads first ");
            WF(8, 4, 0, "K=4 fused U=8  serialised  ");
            snprintf(name, sizeof name, "walk+fma S=5 L=496 K=2 plain U=8 loads first, 2 x 512 B mirror segments lds %2d KB/wg", lds_kb);
            TIME(name, n_jobs* job_bytes,
                 hipLaunchKernelGGL((walk_fma_kernel<5, 8, 2, 1, 1>), dim3((unsigned)(n_jobs / 2)), dim3(320), lds_kb * 1024, 0, in, rows, 2,
                                    out, 0.5, 0.25));
            snprintf(name, sizeof name, "walk+fma S=5 L=496 K=4 fused U=8 loads first, 2 x 512 B mirror segments lds %2d KB/wg", lds_kb);
            TIME(name, n_jobs* job_bytes,
                 hipLaunchKernelGGL((walk_fma_kernel<5, 8, 4, 1, 1>), dim3((unsigned)(n_jobs / 2)), dim3(320), lds_kb * 1024, 0, in, rows, 2,
                                    out, 0.5, 0.25));
        }
    }
    // half-precision rows: 8 bytes per lane, walk shape (D = 1280 halves: S = 5), 496 rows
    {
        const int rows = 496;
        const int64_t job_bytes = (int64_t)rows * 5 * 512;
        const int64_t n_jobs = total_bytes / job_bytes / 2 * 2;
        for (int lds_kb : {1, 48}) {
            snprintf(name, sizeof name, "walk half S=5 L=496 K=2 plain U=8  (8 B per lane) lds %2d KB/wg", lds_kb);
            TIME(name, n_jobs * job_bytes,
                 hipLaunchKernelGGL((walk_half_kernel<5, 8, 2>), dim3((unsigned)(n_jobs / 2)), dim3(320), lds_kb * 1024, 0, in, rows, 2, out, 0.5, 0.25));
            snprintf(name, sizeof name, "walk half S=5 L=496 K=2 plain U=16 (8 B per lane) lds %2d KB/wg", lds_kb);
            TIME(name, n_jobs * job_bytes,
                 hipLaunchKernelGGL((walk_half_kernel<5, 16, 2>), dim3((unsigned)(n_jobs / 2)), dim3(320), lds_kb * 1024, 0, in, rows, 2, out, 0.5, 0.25));
            snprintf(name, sizeof name, "walk half S=5 L=496 K=4 fused U=8  (8 B per lane) lds %2d KB/wg", lds_kb);
            TIME(name, n_jobs * job_bytes,
                 hipLaunchKernelGGL((walk_half_kernel<5, 8, 4>), dim3((unsigned)(n_jobs / 2)), dim3(320), lds_kb * 1024, 0, in, rows, 2, out, 0.5, 0.25));
        }
    }
    return 0;
}
