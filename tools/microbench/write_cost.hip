// Microbenchmark: what do small result writes cost a kernel that streams HBM at full rate?
// Each workgroup (256 threads) streams `rows` x 1 KiB from its own region (16 B per lane, nt, 4 loads in flight per wave)
// and then writes `wbytes` of results.  Modes vary where and when the results go.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/write_cost tools/microbench/write_cost.hip && /tmp/write_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int POLICY>  // 0 write-back, 1 sc1, 2 nt
__device__ inline void st(double* p, double v) {
    if (POLICY == 0) *p = v;
    else if (POLICY == 1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __builtin_nontemporal_store(v, p);
}

// group: jobs whose results one workgroup collects before writing them (contiguously) -- a workgroup handles
// `group` consecutive jobs; results of a job = wdoubles doubles.
template <int POLICY>
__global__ __launch_bounds__(256) void stream_kernel(const float* __restrict__ in, int rows, int64_t job_floats, double* __restrict__ out,
                                                      int wdoubles, int64_t out_stride_doubles, int group, int defer) {
    typedef const v4f __attribute__((address_space(1))) * GP;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double acc = 0.0;
    for (int g = 0; g < group; ++g) {
        const int64_t job = (int64_t)blockIdx.x * group + g;
        const float* base = in + job * job_floats + lane * 4;
        for (int r = wave; r + 12 < rows; r += 16) {
            v4f x0 = __builtin_nontemporal_load((GP)(uintptr_t)(base + (int64_t)r * 256));
            v4f x1 = __builtin_nontemporal_load((GP)(uintptr_t)(base + (int64_t)(r + 4) * 256));
            v4f x2 = __builtin_nontemporal_load((GP)(uintptr_t)(base + (int64_t)(r + 8) * 256));
            v4f x3 = __builtin_nontemporal_load((GP)(uintptr_t)(base + (int64_t)(r + 12) * 256));
            acc += (double)x0[0] + (double)x1[1] + (double)x2[2] + (double)x3[3];
        }
        if (!defer) {
            double* o = out + job * out_stride_doubles;
            for (int i = threadIdx.x; i < wdoubles; i += 256) st<POLICY>(o + i, acc);
        }
    }
    if (defer) {  // all results of the group at the end, one contiguous piece
        double* o = out + (int64_t)blockIdx.x * group * out_stride_doubles;
        for (int i = threadIdx.x; i < wdoubles * group; i += 256) st<POLICY>(o + i, acc);
    }
}

int main(int argc, char** argv) {
    const int64_t total_bytes = 32ll << 30;
    float* in;
    double* out;
    hipMalloc(&in, total_bytes);
    hipMemset(in, 0, total_bytes);
    const int64_t out_bytes = 12ll << 30;
    hipMalloc(&out, out_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    struct Cfg { int rows, wdoubles, stride_mult, group, defer, policy; const char* name; };
    std::vector<Cfg> cfgs;
    for (int rows : {512, 128, 32}) {
        cfgs.push_back({rows, 0, 1, 1, 0, 0, "no writes"});
        for (int pol : {0, 1, 2}) cfgs.push_back({rows, 256, 5, 1, 0, pol, "2 KiB per job, stride 10 KiB"});
        cfgs.push_back({rows, 256, 1, 1, 0, 1, "2 KiB per job, contiguous over jobs"});
        cfgs.push_back({rows, 256, 1, 8, 0, 1, "8 jobs per workgroup, 2 KiB after each"});
        cfgs.push_back({rows, 256, 1, 8, 1, 1, "8 jobs per workgroup, 16 KiB at the end"});
        cfgs.push_back({rows, 256, 1, 32, 1, 1, "32 jobs per workgroup, 64 KiB at the end"});
        cfgs.push_back({rows, 32, 1, 1, 0, 1, "256 B per job"});
        cfgs.push_back({rows, 1024, 1, 1, 0, 1, "8 KiB per job"});
    }
    for (const Cfg& c : cfgs) {
        const int64_t job_floats = (int64_t)c.rows * 256;
        int64_t n_jobs = total_bytes / (job_floats * 4);
        n_jobs -= n_jobs % c.group;
        const int64_t stride = (int64_t)c.wdoubles * c.stride_mult;
        if (stride * n_jobs * 8 > out_bytes) { printf("skip %s\n", c.name); continue; }
        const unsigned grid = (unsigned)(n_jobs / c.group);
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0, 0);
            if (c.policy == 0) hipLaunchKernelGGL(stream_kernel<0>, dim3(grid), dim3(256), 0, 0, in, c.rows, job_floats, out, c.wdoubles, stride ? stride : 1, c.group, c.defer);
            else if (c.policy == 1) hipLaunchKernelGGL(stream_kernel<1>, dim3(grid), dim3(256), 0, 0, in, c.rows, job_floats, out, c.wdoubles, stride ? stride : 1, c.group, c.defer);
            else hipLaunchKernelGGL(stream_kernel<2>, dim3(grid), dim3(256), 0, 0, in, c.rows, job_floats, out, c.wdoubles, stride ? stride : 1, c.group, c.defer);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        const double rd = (double)n_jobs * job_floats * 4;
        printf("rows/job %4d  %-42s policy %d : %7.3f ms  read %6.0f GB/s  (writes %.2f %% of bytes)\n", c.rows, c.name, c.policy, best,
               rd / best / 1e6, 100.0 * c.wdoubles * 8 / (job_floats * 4.0));
        fflush(stdout);
    }
    return 0;
}
