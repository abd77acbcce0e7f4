// Launchers of the gfx950 kernels, one translation unit per kernel family so that they compile side by side
// (build_ext.py): the stage-A instantiations alone are two thirds of the device code.  dctfp.hip -- the host side of the C
// ABI -- sees the parameter blocks and the launcher declarations below and no kernel template of these families.
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstddef>
#include <cstdint>

#include "dctfp.h"

#ifndef DCTFP_MAX_N_K
#define DCTFP_MAX_N_K DCTFP_MAX_N
#define DCTFP_MAX_M_K DCTFP_MAX_M
#endif
#include "kernels.hip.h"

namespace dctfp {
struct CutJob;   // reccut_kernel.hip.h
}

namespace dctfp_host {

using namespace dctfp;

// A failure inside a launcher: the code and the message (at most 255 bytes) it wants dctfp_last_error() to carry.
struct LaunchError {
    int code = 0;
    char msg[256] = "";
};
int launch_fail(LaunchError* err, int code, const char* fmt, ...);

// cos(pi p / q) in long double with exact integer argument reduction (p >= 0, q > 0).
inline long double cospi_ratio_host(int64_t p, int64_t q) {
    p %= 2 * q;
    if (p > q) p = 2 * q - p;
    long double sign = 1.0L;
    if (2 * p > q) {
        p = q - p;
        sign = -1.0L;
    }
    const long double pi = 3.141592653589793238462643383279502884L;
    long double r = (4 * p > q) ? sinl(pi * (long double)(q - 2 * p) / (long double)(2 * q))
                                : cosl(pi * (long double)p / (long double)q);
    return sign * r;
}

template <int N>
inline InvTab<N> make_inv() {
    InvTab<N> t;
    if constexpr (N > 1) {
        for (int j = 0; j < N; ++j)
            for (int k = 1; k < N; ++k)
                t.c[j * (N - 1) + (k - 1)] = (double)cospi_ratio_host((int64_t)k * (2 * j + 1), 2 * (int64_t)N);
    } else {
        t.c[0] = 0.0;
    }
    return t;
}

struct AParams {
    const JobA* jobs;
    const Walk* walks;
    bool fused;
    const PieceA* pieces;
    unsigned long long* degenerate;
    char* yprime;
    int64_t job_bytes;
    int packed;
    int n_cols;
    int64_t ld;
    int ldy;
    int n_slabs;
    unsigned grid;
    hipStream_t stream;
};

struct WParams {
    const JobA* jobs;
    const JobB* jobb;
    const Walk* walks;
    const Run* runs;
    const PieceA* pieces;
    const double* stf;
    int8_t* out;
    int n_cols;
    int64_t ld;
    int m;
    unsigned long long* degenerate;
    unsigned grid;
    hipStream_t stream;
    bool two_source = false;  // some piece has PieceA::ptr2 (float32 rows, 8 rows in flight, 4 jobs per flush only)
};

// walk_gen_kernel: the shapes walk_ab_kernel does not take.
struct GParams {
    const JobA* jobs;
    const JobB* jobb;
    const Walk* walks;   // fused: the walks of the runs (parts + whole protein); else unused
    bool fused;
    const Run* runs;
    const PieceA* pieces;
    const double* stp;
    int8_t* out;
    int n_cols;
    int64_t ld;
    int m;
    int n_slots;
    unsigned long long* degenerate;
    unsigned grid;
    unsigned waves;
    size_t lds_bytes;
    hipStream_t stream;
};

constexpr size_t kGenLdsBudget = 150 * 1024;  // of the 160 KB of a CU
constexpr int kGenFusedMaxN = 5;              // fused walks of the general kernel: two sets of n - 1 accumulators per channel ...
constexpr int kGenFusedMaxWaves = 10;         // ... in builds bounded to ten waves per workgroup (168 registers)

// LDS of one slot: Y'[N][CH] float64; the partial Z blocks [S][N][cp] reuse it unless a wave's columns are too few
inline size_t gen_slot_bytes(int n, int m, int waves, int vec) {
    const size_t ch = (size_t)waves * 64 * vec, cp = ((size_t)m + 15) / 16 * 16;
    return ((size_t)n * ch + (cp <= (size_t)64 * vec ? 0 : (size_t)waves * n * cp)) * sizeof(double);
}


// stage A (k_stage_a_*.hip: one unit per storage type)
void launch_a_f32(const AParams& p, int vec, int n, int waves, int unroll);
void launch_a_f64(const AParams& p, int vec, int n, int waves, int unroll);
void launch_a_f16(const AParams& p, int vec, int n, int waves, int unroll);
void launch_a_bf16(const AParams& p, int vec, int n, int waves, int unroll);
// stage B on the matrix pipe (k_stage_b.hip)
void launch_b_mfma(int nt, bool packed, unsigned grid, hipStream_t s, const char* yp, int64_t job_bytes, int64_t rows,
                   int ldy, const double* st, const JobB* jobs, int n, int m, int8_t* out);
// the walk kernels (k_walk.hip, k_gen.hip)
int launch_walk(const WParams& p, int dtype, int s, int g, int unroll, bool fused, bool mfma_a, LaunchError* err);
int launch_gen(const GParams& p, int dtype, int vec, int n, LaunchError* err);
// the domain cutter's recursion (k_reccut.hip): LDS class of a protein (0 .. 3: 512 / 1024 / 1536 / 2048 residues), launch of one class
constexpr int kCutClasses = 4;
int reccut_class_of(int n_res, int64_t n_contacts);
int launch_reccut(int cls, const dctfp::CutJob* jobs, unsigned n, double cut1, double cut2, hipStream_t stream, LaunchError* err);

}  // namespace dctfp_host
