// Microbenchmark: what does v_sad_u8 sustain on gfx950, alone and fed from LDS the way l1_matrix_kernel feeds it?
//   mode 0: 64 accumulators per lane, operands in registers (the issue ceiling of the instruction)
//   mode 1: per k-step 16 ds_read_b32 (8 broadcast "a" rows, 8 "b" rows at an odd row stride) + 64 v_sad_u8  -- the round-3 kernel's loop
//   mode 2: per 4 k-steps 16 ds_read_b128 + 256 v_sad_u8 (rows stored with a 36-dword stride)
//   mode 3: mode 2 with 8 KB of int32 results written per thread-tile at the end of every 120 k-steps (the real output rate)
// Workgroups of 256 threads, WG_PER_CU x 256 of them; prints T byte-differences per second against 157.3 T/s
// (256 CUs x 64 lanes x 4 bytes per clock at 2.4 GHz).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/sad_rate tools/microbench/sad_rate.hip && /tmp/sad_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void sad_kernel(uint32_t* __restrict__ out, int ksteps, int reps, uint32_t seed) {
    __shared__ uint32_t sa[128 * 36];
    __shared__ uint32_t sb[128 * 36];
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    for (int i = threadIdx.x; i < 128 * 36; i += 256) {
        sa[i] = seed * 2654435761u + i;
        sb[i] = seed * 40503u + 7u * i;
    }
    __syncthreads();
    uint32_t acc[8][8] = {};
    for (int rep = 0; rep < reps; ++rep) {
        if (MODE == 0) {
            uint32_t av[8], bv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { av[i] = sa[ty * 8 + i]; bv[i] = sb[tx * 8 + i]; }
            for (int k = 0; k < ksteps; ++k) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(av[i]), "v"(bv[j]));
            }
        } else if (MODE == 1) {
#pragma unroll 2
            for (int k = 0; k < ksteps; ++k) {
                const int kk = k & 31;
                uint32_t av[8], bv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) av[i] = sa[(ty * 8 + i) * 33 + kk];
#pragma unroll
                for (int j = 0; j < 8; ++j) bv[j] = sb[(j * 16 + tx) * 33 + kk];
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_sad_u8(av[i], bv[j], acc[i][j]);
            }
        } else {
            for (int k = 0; k < ksteps; k += 4) {
                const int kk = k & 31;
                v4u av[8], bv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) av[i] = *reinterpret_cast<const v4u*>(&sa[(ty * 8 + i) * 36 + kk]);
#pragma unroll
                for (int j = 0; j < 8; ++j) bv[j] = *reinterpret_cast<const v4u*>(&sb[(j * 16 + tx) * 36 + kk]);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int i = 0; i < 8; ++i)
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_sad_u8(av[i][q], bv[j][q], acc[i][j]);
            }
            if (MODE == 3) {
                uint32_t* o = out + ((size_t)blockIdx.x * 128 + ty * 8) * 128 + tx * 8;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    *reinterpret_cast<v4u*>(o + i * 128) = (v4u){acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
                    *reinterpret_cast<v4u*>(o + i * 128 + 4) = (v4u){acc[i][4], acc[i][5], acc[i][6], acc[i][7]};
                }
            }
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += acc[i][j];
    if (s == 0x12345678u) out[threadIdx.x] = s;
}

template <int MODE>
int run(const char* name, uint32_t* out, int wg_per_cu) {
    const int ksteps = 120, reps = 200, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    sad_kernel<MODE><<<grid, 256>>>(out, ksteps, 2, 1u);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int t = 0; t < 3; ++t) {
        CHECK(hipEventRecord(e0));
        sad_kernel<MODE><<<grid, 256>>>(out, ksteps, reps, 1u + t);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double diffs = (double)grid * 256 * 64 * 4 * ksteps * reps;
    printf("%-58s %d WG/CU: %8.3f ms  %6.1f T differences/s = %.2f of 157.3\n", name, wg_per_cu, best, diffs / best / 1e9, diffs / best / 1e9 / 157.3);
    return 0;
}

int main() {
    uint32_t* out;
    CHECK(hipMalloc(&out, (size_t)256 * 4 * 128 * 128 * 4));
    for (int w : {2, 4}) {
        if (run<0>("registers only", out, w)) return 1;
        if (run<1>("16 ds_read_b32 per 64 (stride 33)", out, w)) return 1;
        if (run<2>("16 ds_read_b128 per 256 (stride 36)", out, w)) return 1;
        if (run<3>("the same + 64 KB of int32 written per tile of 120 k-steps", out, w)) return 1;
    }
    return 0;
}
