// What the chip gives a kernel with the walk kernel's WORK on the job lengths of the multi-domain mixes -- and nothing of its
// logic: workgroups of S waves stream random float32 rows the way walk_ab_kernel does (wave w: channels [128 w, 128 w + 128) and
// their mirror images, 8 rows in flight, nt), with the fused walk's float64 arithmetic per element (two accumulator sets), and
//   * after every JOB (lengths drawn like bench.py's c4 / c5 domain lists: proteins cut at multiples of 25 rows, ~ 86-row parts at
//     c4, ~ 110-row parts at c5): an epilogue of the real one's cost -- per lane four times {6 fma, min / max, one float64
//     division, a dozen selects} and a 40-byte LDS write;
//   * after every fourth job: a flush of the real one's cost -- 32 k-steps x 5 fragments of 512 bytes per wave from a table in
//     L2 (4 k-steps in flight), 15 MFMAs per k-step, the partial blocks through LDS, and per job three rows of {S LDS reads,
//     wave min / max, two divisions} by one wave each, all waves meeting at a barrier before and after.
// No job tables, no cosine tables, no pieces, no tickets, no int8 output: the ceiling for THIS work on THESE job lengths.
// ("if it does not move, commit the same-work microbenchmark that shows the ceiling for these job lengths".)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/walk_same_work tools/microbench/walk_same_work.hip && /tmp/walk_same_work
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ inline v4f load16(__amdgpu_buffer_rsrc_t rs, int lane_bytes, int uniform_bytes) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    const v4i r = __builtin_amdgcn_raw_buffer_load_b128(rs, lane_bytes, uniform_bytes, 2);
    return __builtin_bit_cast(v4f, r);
}

__global__ void fill_kernel(float* __restrict__ x, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)(i * 2654435761u) ^ (uint32_t)(i >> 17);
        h ^= h >> 13;
        h *= 0x5bd1e995u;
        h ^= h >> 15;
        x[i] = (float)(int32_t)h * 4.6e-10f + 5.0f;
    }
}

struct WgJobs {
    uint32_t first;   // first entry of this workgroup in the job-length list
    uint32_t n;       // jobs (parts; a whole-protein job streams nothing and is counted in the flush only)
    uint64_t row0;    // first row of the workgroup
};

template <int D, int S>
__global__ __launch_bounds__(S * 64, S >= 10 ? 3 : 4) void same_work_kernel(const float* __restrict__ x, const WgJobs* __restrict__ wgs,
                                                                            const uint16_t* __restrict__ job_rows, float* __restrict__ sink,
                                                                            const double* __restrict__ frag_tab) {
    __shared__ double lds_t[S][4][256];
    __shared__ uint32_t lds_c[S][4][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const WgJobs wg = wgs[blockIdx.x];
    const int pair0 = wave * 128 + 4 * (lane & 31);
    const bool pad = pair0 >= D / 2;
    const int colc = pad ? 0 : (lane >= 32 ? D - 4 - pair0 : pair0);
    double zacc[15];
#pragma unroll
    for (int i = 0; i < 15; ++i) zacc[i] = 0.0;
    double keep = 0.0;
    uint64_t row = wg.row0;
    int pending = 0;
    for (uint32_t jb = 0; jb < wg.n; ++jb) {
        const int nrows = (int)job_rows[wg.first + jb];
        const bool whole = nrows == 0xffff;   // a whole-protein job: its coefficients came with the parts
        double f[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) f[k][q] = 1e-3 * (k + q);
        if (!whole && !pad) {
            const __amdgpu_buffer_rsrc_t rows = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + row * D), 0, 0x7fffffff, 0x00020000);
            const double c0 = 0.3 + 1e-9 * lane, c1 = 0.5, c2 = 0.7, c3 = 0.9;
            const v4f r0 = load16(rows, colc * 4, 0);
            int r = 0;
            for (; r + 8 <= nrows; r += 8) {
                v4f v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = load16(rows, colc * 4, (r + u) * D * 4);
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double d = (double)v[u][q] - (double)r0[q];
                        f[0][q] = fma(c0, d, f[0][q]);
                        f[1][q] = fma(c1, d, f[1][q]);
                        f[2][q] = fma(c2, d, f[2][q]);
                        f[3][q] = fma(c3, d, f[3][q]);
                    }
            }
            for (; r < nrows; ++r) {
                const v4f v = load16(rows, colc * 4, r * D * 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double d = (double)v[q] - (double)r0[q];
                    f[0][q] = fma(c0, d, f[0][q]);
                    f[1][q] = fma(c1, d, f[1][q]);
                    f[2][q] = fma(c2, d, f[2][q]);
                    f[3][q] = fma(c3, d, f[3][q]);
                }
            }
        }
        if (!whole) row += (uint64_t)nrows;
        // ---- epilogue: per channel three resampled values, min / max, one division, a state word; the packed row into LDS
        {
            double tv[4];
            uint32_t c4 = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double a = whole ? f[2][q] : f[0][q], b = whole ? f[3][q] : f[1][q];
                const double y0 = fma(0.866, a, 0.5 * b), y1 = fma(0.0, a, -1.0 * b), y2 = fma(-0.866, a, 0.5 * b);
                const double mn = fmin(y0, fmin(y1, y2)), mx = fmax(y0, fmax(y1, y2));
                const double den = mx - mn;
                const double n0 = y0 - mn, n1 = y1 - mn, n2 = y2 - mn;
                const double mid = (n0 != 0.0 && n0 != den) ? n0 : ((n1 != 0.0 && n1 != den) ? n1 : n2);
                uint32_t code = (n0 == den ? 1u : (n0 == 0.0 ? 0u : 2u)) | (n1 == den ? 1u : (n1 == 0.0 ? 0u : 2u)) << 2 |
                                (n2 == den ? 1u : (n2 == 0.0 ? 0u : 2u)) << 4;
                tv[q] = mid / den;
                c4 |= code << (8 * q);
                __builtin_amdgcn_sched_barrier(0);
            }
            *reinterpret_cast<double4*>(&lds_t[wave][pending][4 * lane]) = make_double4(tv[0], tv[1], tv[2], tv[3]);
            lds_c[wave][pending][lane] = c4;
        }
        ++pending;
        if (pending < 4 && jb + 1 < wg.n) continue;
        // ---- flush: fragments from L2, 480 MFMAs per wave, partial blocks through LDS, rows by `pending` waves
        {
            const __amdgpu_buffer_rsrc_t ft = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(frag_tab) + (size_t)wave * 32 * 5 * 64, 0, 0x7fffffff, 0x00020000);
            double bq[4][5];
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int c = 0; c < 5; ++c) bq[st][c] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(ft, lane * 8, (st * 5 + c) * 512, 0));
            for (int q = 0; q < 8; ++q) {
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const double y = lds_t[wave][lane & 3][(q * 16 + st * 4 + (lane >> 4)) & 255] + (double)(lds_c[wave][lane & 3][q] & 3u);
#pragma unroll
                    for (int j = 0; j < 3; ++j)
#pragma unroll
                        for (int c = 0; c < 5; ++c) zacc[j * 5 + c] = __builtin_amdgcn_mfma_f64_4x4x4f64(y + j, bq[st][c], zacc[j * 5 + c], 0, 0, 0);
                    const int nxt = (q * 4 + st + 4) & 31;
#pragma unroll
                    for (int c = 0; c < 5; ++c) bq[st][c] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(ft, lane * 8, (nxt * 5 + c) * 512, 0));
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int c = 0; c < 5; ++c) lds_t[wave][lane >> 4][j * 80 + c * 16 + (lane & 15)] = zacc[j * 5 + c];
            __syncthreads();
            if (wave < pending) {   // one job's three rows: sums over the waves, min / max over the row, two divisions
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    double ze = 0.0, zo = 0.0;
#pragma unroll
                    for (int w = 0; w < S; ++w) {
                        ze += lds_t[w][wave][j * 80 + (lane & 31)];
                        zo += lds_t[w][wave][j * 80 + 40 + (lane & 31)];
                    }
                    double mn = fmin(ze + zo, ze - zo), mx = fmax(ze + zo, ze - zo);
#pragma unroll
                    for (int o = 32; o >= 1; o >>= 1) {
                        mn = fmin(mn, __shfl_xor(mn, o));
                        mx = fmax(mx, __shfl_xor(mx, o));
                    }
                    keep += (ze + zo - mn) / (mx - mn) + (ze - zo - mn) / (mx - mn);
                }
            }
            __syncthreads();
            pending = 0;
        }
    }
    double t = keep;
#pragma unroll
    for (int i = 0; i < 15; ++i) t += zacc[i];
    if (t == 1.2345e300) sink[blockIdx.x] = (float)t;
}

// Job lists as bench.py's c4 (L ~ U[100, 500], 1-6 parts cut at multiples of 25 + whole protein) / c5 (pfam-like lengths, ~ 110-row parts)
static void make_jobs(int mix, int jobs_per_wg, int n_wg, std::vector<WgJobs>& wgs, std::vector<uint16_t>& rows, uint64_t& total_rows) {
    std::mt19937_64 rng(2024);
    std::vector<uint16_t> stream;   // parts and whole-protein markers in protein order
    uint64_t at = 0;
    wgs.clear();
    rows.clear();
    while ((int)wgs.size() < n_wg) {
        WgJobs w;
        w.first = (uint32_t)rows.size();
        w.row0 = at;
        uint32_t n = 0;
        while (n < (uint32_t)jobs_per_wg) {
            int L, k;
            if (mix == 4) {
                L = 100 + (int)(rng() % 401);
                k = 1 + (int)(rng() % 6);
            } else {
                std::gamma_distribution<double> g(2.2, 170.0);
                L = std::min(1330, std::max(81, (int)g(rng)));
                std::normal_distribution<double> nd(0.0, 0.7);
                k = std::max(1, (int)std::lround(L / 110.0 + nd(rng)));
            }
            k = std::max(1, std::min(k, L / 30));
            if (k == 1) {
                rows.push_back((uint16_t)L);
                at += L;
                n += 1;
                continue;
            }
            std::vector<int> cuts;
            while ((int)cuts.size() < k - 1) {
                const int c = 25 * (1 + (int)(rng() % (L / 25 - 1)));
                bool dup = false;
                for (int v : cuts) dup |= v == c;
                if (!dup) cuts.push_back(c);
            }
            std::sort(cuts.begin(), cuts.end());
            int prev = 0;
            for (int c : cuts) {
                rows.push_back((uint16_t)(c - prev));
                prev = c;
            }
            rows.push_back((uint16_t)(L - prev));
            rows.push_back(0xffff);   // the whole protein
            at += L;
            n += (uint32_t)k + 1;
        }
        w.n = n;
        wgs.push_back(w);
    }
    total_rows = at;
}

template <int D, int S>
void run(int mix, int per_cu, int jobs_per_wg, float* x, size_t cap_bytes, float* sink, const double* ft) {
    std::vector<WgJobs> wgs;
    std::vector<uint16_t> rows;
    uint64_t total_rows = 0;
    int n_wg = 256 * per_cu * 10;   // ten rounds of workgroups, as the kernel's runs are cut
    make_jobs(mix, jobs_per_wg, n_wg, wgs, rows, total_rows);
    while (total_rows * D * 4 > cap_bytes) {
        n_wg = n_wg * 3 / 4;
        make_jobs(mix, jobs_per_wg, n_wg, wgs, rows, total_rows);
    }
    WgJobs* dw = nullptr;
    uint16_t* dr = nullptr;
    (void)hipMalloc(&dw, wgs.size() * sizeof(WgJobs));
    (void)hipMalloc(&dr, rows.size() * 2 + 16);
    (void)hipMemcpy(dw, wgs.data(), wgs.size() * sizeof(WgJobs), hipMemcpyHostToDevice);
    (void)hipMemcpy(dr, rows.data(), rows.size() * 2, hipMemcpyHostToDevice);
    auto launch = [&]() { hipLaunchKernelGGL((same_work_kernel<D, S>), dim3((unsigned)wgs.size()), dim3(S * 64), 0, 0, x, dw, dr, sink, ft); };
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
    } while (el < 2.0);
    size_t n_parts = 0;
    for (uint16_t v : rows) n_parts += v != 0xffff;
    printf("c%d-like job lengths, D = %4d, %2d waves per workgroup, %d per CU, %d jobs per workgroup: %zu workgroups, %zu jobs (%.1f rows per part), "
           "%.2f GB: %.3f ms per pass = %.0f GB/s = %.3f of 8 TB/s\n", mix, D, S, per_cu, jobs_per_wg, wgs.size(), rows.size(), (double)total_rows / n_parts,
           total_rows * D * 4 / 1e9, 1e3 * el / n, (double)total_rows * D * 4 * n / el / 1e9, (double)total_rows * D * 4 * n / el / 8e12);
    fflush(stdout);
    (void)hipFree(dw);
    (void)hipFree(dr);
}

int main() {
    const size_t bytes = (size_t)60 << 30;
    float *x = nullptr, *sink = nullptr;
    if (hipMalloc(&x, bytes) != hipSuccess || hipMalloc(&sink, 4 << 20) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, x, bytes / 4);
    double* ft = nullptr;
    if (hipMalloc(&ft, (size_t)10 * 32 * 5 * 64 * 8 + 4096) != hipSuccess) return 1;
    (void)hipMemset(ft, 0, (size_t)10 * 32 * 5 * 64 * 8 + 4096);
    (void)hipDeviceSynchronize();
    run<2560, 10>(4, 1, 16, x, bytes, sink, ft);
    run<2560, 10>(4, 1, 8, x, bytes, sink, ft);
    run<640, 3>(5, 5, 16, x, bytes, sink, ft);
    run<1280, 5>(4, 3, 16, x, bytes, sink, ft);
    return 0;
}
