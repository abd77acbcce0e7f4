// Microbenchmark: which float64 instructions share an execution unit on gfx950?
// The walk kernel streams rows with v_cvt_f64_f32 / v_add_f64 / v_fma_f64 and contracts with v_mfma_f64_4x4x4_4b; if the
// matrix instruction runs on the same DP units as the vector FMAs, the two costs add up instead of overlapping.
// One workgroup of 512 threads per CU: every wave reads the SIMD it sits on (HW_ID) and takes a ticket there; odd tickets
// run mode A, even tickets mode B, so that each SIMD holds both kinds.  The time of (A, B) against (A, idle) and (idle, B)
// says whether the two overlap.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/f64_pipes tools/microbench/f64_pipes.hip && /tmp/f64_pipes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

enum Mode { IDLE = 0, FMA = 1, ADD = 2, CVT = 3, MFMA = 4, I32 = 5, CVT_BITS = 6 };

template <int MODE>
__device__ inline double work(int iters, double seed) {
    double a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + i;
    if (MODE == FMA) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a[i]) : "v"(seed));
        }
    } else if (MODE == ADD) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
        }
    } else if (MODE == CVT) {
        float f[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) f[i] = (float)seed + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
        }
    } else if (MODE == MFMA) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(seed, seed, a[i], 0, 0, 0);
        }
    } else if (MODE == I32) {
        uint32_t u[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) u[i] = (uint32_t)seed + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(i));
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = u[i];
    } else if (MODE == CVT_BITS) {
        // exact float32 -> float64 of a normal number with integer instructions: 3 full-rate ops instead of one conversion
        uint32_t u[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) u[i] = __float_as_uint((float)seed + i);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                uint32_t lo, hi, t;
                asm volatile("v_lshlrev_b32 %0, 29, %1" : "=v"(lo) : "v"(u[i]));
                asm volatile("v_ashrrev_i32 %0, 3, %1" : "=v"(t) : "v"(u[i]));
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(hi) : "v"(t), "v"(0x38000000u));  // (sign fix-up left out)
                a[i] = __hiloint2double((int)hi, (int)lo);
            }
        }
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    return s;
}

template <int MA, int MB>
__global__ __launch_bounds__(512) void pipes_kernel(int iters, double seed, double* out) {
    __shared__ int tickets[4];
    __shared__ int placed[8];
    if (threadIdx.x < 4) tickets[threadIdx.x] = 0;
    __syncthreads();
    uint32_t hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const int simd = (hw >> 4) & 3;
    int ticket = 0;
    if ((threadIdx.x & 63) == 0) {
        ticket = atomicAdd(&tickets[simd], 1);
        placed[threadIdx.x >> 6] = simd * 16 + ticket;
    }
    ticket = __shfl(ticket, 0);
    __syncthreads();
    double r;
    if (ticket & 1) r = work<MB>(iters, seed);
    else r = work<MA>(iters, seed);
    if (r == 12345.678) out[blockIdx.x] = r;
    if (blockIdx.x == 0 && threadIdx.x < 8) out[1024 + threadIdx.x] = placed[threadIdx.x];
}

template <int MA, int MB>
static float run(int iters, double* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((pipes_kernel<MA, MB>), dim3(256), dim3(512), 0, 0, iters, 1.0, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((pipes_kernel<MA, MB>), dim3(256), dim3(512), 0, 0, iters, 1.0, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    static bool shown = false;
    if (!shown) {
        double h[8];
        hipMemcpy(h, out + 1024, sizeof h, hipMemcpyDeviceToHost);
        printf("placement of the 8 waves of workgroup 0 (simd.ticket):");
        for (int i = 0; i < 8; ++i) printf(" %d.%d", (int)h[i] / 16, (int)h[i] % 16);
        printf("\n");
        shown = true;
    }
    return ms;
}

int main() {
    double* out;
    hipMalloc(&out, 1 << 20);
    const int iters = 20000;
    const double n_inst = 16.0 * iters;  // per wave
#define R(A, B, label)                                                                                                \
    do {                                                                                                              \
        float ms = run<A, B>(iters, out);                                                                             \
        printf("%-44s %8.3f ms   %6.2f ns per instruction of one wave (%.1f cycles at 2.4 GHz)\n", label, ms, ms * 1e6 / n_inst, \
               ms * 1e6 / n_inst * 2.4);                                                                              \
        fflush(stdout);                                                                                               \
    } while (0)
    R(FMA, IDLE, "v_fma_f64 | idle");
    R(FMA, FMA, "v_fma_f64 | v_fma_f64");
    R(ADD, IDLE, "v_add_f64 | idle");
    R(CVT, IDLE, "v_cvt_f64_f32 | idle");
    R(CVT, CVT, "v_cvt_f64_f32 | v_cvt_f64_f32");
    R(I32, IDLE, "v_add_u32 | idle");
    R(I32, I32, "v_add_u32 | v_add_u32");
    R(CVT_BITS, IDLE, "f32->f64 by 3 integer ops | idle");
    R(MFMA, IDLE, "v_mfma_f64_4x4x4_4b | idle");
    R(MFMA, MFMA, "v_mfma_f64_4x4x4_4b | same");
    R(FMA, MFMA, "v_fma_f64 | v_mfma_f64_4x4x4_4b");
    R(CVT, MFMA, "v_cvt_f64_f32 | v_mfma_f64_4x4x4_4b");
    R(I32, MFMA, "v_add_u32 | v_mfma_f64_4x4x4_4b");
    R(FMA, CVT, "v_fma_f64 | v_cvt_f64_f32");
    R(FMA, I32, "v_fma_f64 | v_add_u32");
    return 0;
}
