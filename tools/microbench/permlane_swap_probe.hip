#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* o) {
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    v2u r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    v2u s = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1]; o[128 + threadIdx.x] = s[0]; o[192 + threadIdx.x] = s[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* n[4] = {"p16 vdst", "p16 src ", "p32 vdst", "p32 src "};
    for (int r = 0; r < 4; ++r) { printf("%s:", n[r]); for (int i = 0; i < 64; i += 4) printf(" %3u", h[64 * r + i]); printf("\n"); }
    return 0;
}
