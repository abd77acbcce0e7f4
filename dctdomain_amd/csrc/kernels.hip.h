// Device code of libdctfp.so (gfx950 only).  See DESIGN.md for the data layout and
// the roofline of each kernel.  Math reference: mgtools/DCTdomain
// src/fingerprint.py:126-142 (idct_quant), :110-123 (scale), :194-195 (int8 cast).
//
// With C_N[k,t] = s_k cos(pi k (2t+1) / (2N)) (scipy DCT-II, norm='ortho'),
// idct_quant(., K) along an axis of length N is   y = C_K^T . C_N[:K] . x   followed by
// a min-max scale of the K resampled values.  The scale removes the k = 0 term
// (a per-vector constant) and every common positive factor (s_k is the same for all
// k >= 1), so the kernels accumulate only
//     F_k = sum_t cos(pi k (2t+1) / (2N)) * (x_t - x_0),   k = 1 .. K-1
// in float64 and resample with the bare cosines.  Subtracting x_0 is exact in float64
// for float32 data of similar magnitude and makes a constant vector give exactly
// F_k = 0 -> 0/0 = NaN, as the reference's FFT does.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace dctfp {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v4h __attribute__((ext_vector_type(4)));
typedef unsigned short v4us __attribute__((ext_vector_type(4)));
typedef unsigned short v8us __attribute__((ext_vector_type(8)));

// Cosine tables are read-only for the kernels and always addressed wave-uniformly: a constant-address-space view lets the
// compiler fetch them through the scalar cache (s_load_dwordx4) -- a pointer loaded from a job record is otherwise a
// generic pointer and every cosine becomes a 64-lane vector load.
typedef const double __attribute__((address_space(4))) * CosTab;
__device__ inline CosTab cos_tab(const double* p) { return (CosTab)(uintptr_t)p; }

struct bf16_t {  // storage-only bfloat16 (the upper half of a float32)
    unsigned short bits;
};

struct JobA {              // one (layer, domain) matrix of stage A
    uint32_t piece_begin;  // first PieceA of this job
    uint32_t n_pieces;
    uint32_t n_rows;       // L_d
    uint32_t reserved;
    const double* basis;   // this length's cosine table (context-owned cache, one table per distinct length)
    const double* w_basis; // fused groups: cosine table of the whole protein ...
    const void* w_ref;     // ... and the whole protein's first row
};

struct PieceA {
    const void* ptr;  // first element of the piece's first row
    uint32_t n_rows;
    uint32_t t0;      // position of the piece's first row inside the domain matrix
    uint32_t w0;      // position of the piece's first row inside the whole protein
    uint32_t reserved;
    // Rows that exist only as the overlap of two WINDOWS of the language model (Embedding.embed_seq, src/embedding.py:185-187:
    // run[-olp:] = (run[-olp:] + new[:olp]) / 2): row r of the piece is the float32 (ptr row r + ptr2 row r) / 2, taken in the
    // row load of walk_ab_kernel -- the stitched matrix is never written (dctfp_quantize_windows).  nullptr: an ordinary piece.
    // Only walk_ab_kernel (float32 rows) reads it; the host sends two-source pieces nowhere else.
    const void* ptr2;
};

struct JobB {
    int64_t out_off;  // byte offset of this (layer, domain) block in the int8 output
};

template <int N>
struct InvTab {  // c[j * (N-1) + k-1] = cos(pi k (2j+1) / (2N)), j < N, 1 <= k < N
    double c[N > 1 ? N * (N - 1) : 1];
};

// cos(pi * p / q) with the argument reduced exactly in integers (q > 0, p >= 0).
__device__ inline double cospi_ratio(uint64_t p, uint64_t q) {
    p %= 2 * q;
    if (p > q) p = 2 * q - p;
    double sign = 1.0;
    if (2 * p > q) {
        p = q - p;
        sign = -1.0;
    }
    double r = (4 * p > q) ? sinpi((double)(q - 2 * p) / (double)(2 * q)) : cospi((double)p / (double)q);
    return sign * r;
}

// ---------------------------------------------------------------------------
// K0: cosine tables for stage A, one per distinct domain length (filled once per context, then reused by
// every later call).  Table of length L:   tab[t*NK + (k-1)] = cos(pi k (2t+1) / (2L)),  t < L,
// followed by the prefix sums              pre[t*NK + (k-1)] = sum_{t' < t} cos(pi k (2t'+1) / (2L)),  t <= L
// (closed form sin(pi k t / L) / (2 sin(pi k / 2L)); pre = tab + L*NK).  The prefix sums let a fused walk accumulate the
// whole protein against the PART's first row -- one subtraction per element feeds both accumulator sets -- and restore
// the whole protein's own shift once per part (walk_ab_kernel).
// ---------------------------------------------------------------------------
struct BasisJob {
    double* tab;
    uint32_t len;
    uint32_t reserved;
};

// sin(pi * p / q) with the argument reduced exactly in integers (q > 0, p >= 0).
__device__ inline double sinpi_ratio(uint64_t p, uint64_t q) {
    p %= 2 * q;
    double sign = 1.0;
    if (p > q) {
        p -= q;
        sign = -1.0;
    }
    if (2 * p > q) p = q - p;
    const double r = (4 * p > q) ? cospi((double)(q - 2 * p) / (double)(2 * q)) : sinpi((double)p / (double)q);
    return sign * r;
}

#ifndef DCTFP_TEMPLATES_ONLY   // (a plain kernel: defined in dctfp.hip only, the kernel-family units see the templates)
__global__ void basis_kernel(const BasisJob* __restrict__ tabs, int nk) {
    const BasisJob tj = tabs[blockIdx.y];
    const uint64_t n_cos = (uint64_t)tj.len * nk;
    const uint64_t total = n_cos + ((uint64_t)tj.len + 1) * nk;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        if (i < n_cos) {
            const uint64_t t = i / nk;
            const uint64_t k = i % nk + 1;
            tj.tab[i] = cospi_ratio(k * (2 * t + 1), 2 * (uint64_t)tj.len);
        } else {
            const uint64_t t = (i - n_cos) / nk;
            const uint64_t k = (i - n_cos) % nk + 1;
            // (k < 2L always: k <= n - 1 <= L - 1, so the denominator is not zero)
            tj.tab[i] = sinpi_ratio(k * t, tj.len) / (2.0 * sinpi_ratio(k, 2 * (uint64_t)tj.len));
        }
    }
}
#endif

// Raw (unconverted) register image of one row segment, so that UNROLL loads can be in
// flight before the first conversion.  16 bytes per lane: 4 x float32, 2 x float64, 8 x float16 / bfloat16.
template <typename T, int VEC>
struct Raw;
template <>
struct Raw<float, 4> { typedef v4f type; };
template <>
struct Raw<float, 1> { typedef float type; };
template <>
struct Raw<double, 2> { typedef v2d type; };
template <>
struct Raw<double, 1> { typedef double type; };
template <>
struct Raw<_Float16, 8> { typedef v8h type; };
template <>
struct Raw<_Float16, 4> { typedef v4h type; };  // 8 bytes per lane: fused walks of half-precision rows (register budget)
template <>
struct Raw<_Float16, 1> { typedef _Float16 type; };
template <>
struct Raw<bf16_t, 8> { typedef v8us type; };
template <>
struct Raw<bf16_t, 4> { typedef v4us type; };
template <>
struct Raw<bf16_t, 1> { typedef unsigned short type; };

// Streaming load: global address space (global_load_*, counted vmcnt waits) + nt policy.
template <typename T, int VEC>
__device__ inline typename Raw<T, VEC>::type load_raw(const T* p) {
    typedef typename Raw<T, VEC>::type R;
    typedef const R __attribute__((address_space(1))) * GP;
    return __builtin_nontemporal_load((GP)(uintptr_t)p);  // +6..8 % over the default cache policy (profiles/r01)
}
// The same through a buffer descriptor: base in scalar registers, ONE 32-bit lane offset for every load of a piece, the row
// (or table step) offset in a scalar register -- no 64-bit address arithmetic on the vector pipe and no address registers
// (the walk kernel issues a load per 25 vector instructions and its flush wants many fragment loads in flight).
typedef unsigned v2u32 __attribute__((ext_vector_type(2)));
typedef unsigned v4u32 __attribute__((ext_vector_type(4)));
__device__ inline __amdgpu_buffer_rsrc_t wave_buffer(const void* base) {  // `base` must be wave-uniform; no range check
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000);
}
template <typename R, bool NT>
__device__ inline R buffer_load_raw(__amdgpu_buffer_rsrc_t rs, int lane_bytes, int uniform_bytes) {
#ifndef DCTFP_STREAM_AUX
#define DCTFP_STREAM_AUX 2
#endif
    constexpr int aux = NT ? DCTFP_STREAM_AUX : 0;  // gfx94x / gfx950 cache-policy bits: 1 = sc0, 2 = nt, 16 = sc1
    if constexpr (sizeof(R) == 16) {
        const v4u32 r = __builtin_amdgcn_raw_buffer_load_b128(rs, lane_bytes, uniform_bytes, aux);
        return __builtin_bit_cast(R, r);
    } else {
        static_assert(sizeof(R) == 8, "8 or 16 bytes per lane");
        const v2u32 r = __builtin_amdgcn_raw_buffer_load_b64(rs, lane_bytes, uniform_bytes, aux);
        return __builtin_bit_cast(R, r);
    }
}
// Element v of a raw image as float64 (every storage type converts exactly).
template <typename T, int VEC, typename R>
__device__ inline double raw_elem(const R& r, int v) {
    if constexpr (sizeof(T) == 2 && !__is_same(T, _Float16)) {  // bf16_t
        unsigned short b;
        if constexpr (VEC == 1) b = r;
        else b = r[v];
        return (double)__uint_as_float((uint32_t)b << 16);
    } else if constexpr (__is_same(T, _Float16)) {
        if constexpr (VEC == 1) return (double)(float)r;
        else return (double)(float)r[v];
    } else {
        if constexpr (VEC == 1) return (double)r;
        else return (double)r[v];
    }
}

// Stores of stage A's results (Y').  They are 0.5 % of the bytes the kernel moves, but as ordinary write-back
// stores they cost 8 % of its time at C2 and up to 30 % on 25-row jobs: with the stores removed (arithmetic kept) the
// kernel streams at 7.0 TB/s for every job length, and issuing the same stores at the start of the workgroup
// instead of its end changes nothing -- it is the write traffic, not a wait (profiles/r01/stage_a_store_experiments.txt).
// Written through at agent scope (`sc1`) they cost about half of that; nt and sc0 sc1 measure the same within 1 %.
template <typename V>
__device__ inline void store_through(V* p, V v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------
// Stage-A epilogue of one channel: resample the K-1 accumulated coefficients to N points and
// min-max scale them (scale(): src/fingerprint.py:110-123, applied per channel at :139-140).
// `degenerate` counts the channels whose N resampled values are all equal (an exactly constant
// channel): 0/0 = NaN here and wherever the reference's FFT cancels exactly, but round-off noise in
// the reference at the other lengths (tests/golden/fence_golden.json) -- the one input class where the
// result is reported instead of matched.
// ---------------------------------------------------------------------------
// counters[0] += 1, and a plain store of 1 into the context's host-mapped flag word (its device address sits in
// counters[kFlagSlot]): the host learns "this call saw one" from its own memory after the stream synchronisation it
// does anyway -- no copy, no extra wait (Fingerprint.quantize / make_db log a warning with the protein's id).
constexpr int kFlagSlot = 15;
__device__ __noinline__ void note_degenerate(unsigned long long* __restrict__ counters) {
    atomicAdd(counters, 1ull);
    volatile uint32_t* flag = reinterpret_cast<volatile uint32_t*>((uintptr_t)counters[kFlagSlot]);
    if (flag) *flag = 1u;
}

template <int N>
__device__ inline void scale_channel(const double (&f)[N > 1 ? N - 1 : 1], const InvTab<N>& inv, bool pad, double (&z)[N],
                                     unsigned long long* __restrict__ degenerate) {
    constexpr int NK = N - 1;
    double y[N];
    double mn = INFINITY, mx = -INFINITY;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < NK; ++k) s = fma(inv.c[j * NK + k], f[k], s);
        y[j] = s;
        bad |= (s != s);
        mn = fmin(mn, s);
        mx = fmax(mx, s);
    }
    const double den = mx - mn;
    if (!bad && den == 0.0 && !pad) note_degenerate(degenerate);
#pragma unroll
    for (int j = 0; j < N; ++j) z[j] = pad ? 0.0 : (bad ? __builtin_nan("") : (y[j] - mn) / den);
}

// n = 3: the scaled values of a channel are {0, t, 1} (or NaN): one float64 t and a 2-bit state per row
// (0 -> 0.0, 1 -> 1.0, 2 -> t, 3 -> NaN) carry them exactly: 9 bytes instead of 24.
__device__ inline void pack_channel(const double (&z)[3], double& t, unsigned& code) {
    t = 0.0;
    code = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {  // selects only: this runs per channel with divergent values
        const bool is_nan = z[j] != z[j], is_zero = z[j] == 0.0, is_one = z[j] == 1.0;
        const unsigned st = is_nan ? 3u : (is_zero ? 0u : (is_one ? 1u : 2u));
        t = (st == 2u) ? z[j] : t;
        code |= st << (2 * j);
    }
}

// scale_channel + pack_channel for n = 3 with ONE division instead of three: (y - min) / (max - min) is exactly 0 for the
// minimum and exactly 1 for the maximum whenever max - min is finite and positive, NaN or 0 by the IEEE rules otherwise;
// only the middle value needs the divider.  Same states and the same t, bit for bit, as pack_channel(scale_channel()).
__device__ inline void scale_pack3(const double (&f)[2], const InvTab<3>& inv, bool pad, double& t, unsigned& code,
                                   unsigned long long* __restrict__ degenerate) {
    double y[3];
    double mn = INFINITY, mx = -INFINITY;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double s = fma(inv.c[j * 2 + 1], f[1], fma(inv.c[j * 2], f[0], 0.0));
        y[j] = s;
        bad |= (s != s);
        mn = fmin(mn, s);
        mx = fmax(mx, s);
    }
    const double den = mx - mn;
    if (!bad && den == 0.0 && !pad) note_degenerate(degenerate);
    const bool zero_ok = den > 0.0;                    // 0 / den = 0   (else 0 / 0 or 0 / NaN = NaN)
    const bool one_ok = zero_ok && den < INFINITY;     // den / den = 1 (else inf / inf = NaN)
    const bool den_inf = den == INFINITY;              // finite / inf = 0
    double mid = 0.0;
    code = 0;
    bool any_nan = false;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double num = y[j] - mn;
        const bool is0 = num == 0.0, is1 = num == den;
        unsigned st = den_inf ? 0u : 2u;
        st = is1 ? (one_ok ? 1u : 3u) : st;
        st = is0 ? (zero_ok ? 0u : 3u) : st;
        st = (bad || num != num) ? 3u : st;
        mid = (st == 2u) ? num : mid;
        // A NaN row is stored as state 2 with t = NaN: states 2 and 3 never meet in one channel (a channel with a finite,
        // positive range has the states 0, 1, 2 only; any other range leaves 0 and 3 only), and the flush's unpacking loses
        // its NaN case (5 instead of 9 instructions per value).
        any_nan |= st == 3u;
        st = st == 3u ? 2u : st;
        code |= st << (2 * j);
    }
    t = mid / den;
    t = any_nan ? __builtin_nan("") : t;
    if (pad) {
        t = 0.0;
        code = 0;
    }
}

// ... and stores Y'[j][col], j < N (two-kernel path).  Padding columns (col >= n_cols) are written as 0.
template <int N>
__device__ inline void finish_channel(const double (&f)[N > 1 ? N - 1 : 1], const InvTab<N>& inv, double* __restrict__ o,
                                      int ldy, bool pad, bool packed, void* pk, unsigned long long* __restrict__ degenerate) {
    double z[N];
    scale_channel<N>(f, inv, pad, z, degenerate);
    if constexpr (N == 3) {
        if (packed) {
            double t;
            unsigned code;
            pack_channel(z, t, code);
            store_through(o, t);
            store_through(reinterpret_cast<uint8_t*>(pk), (uint8_t)code);
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < N; ++j) store_through(o + (size_t)j * ldy, z[j]);
}

// One value of a packed Y' row: state bits -> 0, 1, t or NaN (bit selects, no branches).
__device__ inline double unpack_y(unsigned st, double t) {
    unsigned long long v = (st == 2u) ? (unsigned long long)__double_as_longlong(t) : 0ull;
    v = (st == 1u) ? 0x3FF0000000000000ull : v;
    v = (st == 3u) ? 0x7FF8000000000000ull : v;
    return __longlong_as_double((long long)v);
}

// The same for the walk kernel's slots, from a state word with the channel's byte at a static position: masks instead of
// compares and selects (the flush unpacks 2 x 4 x 128 values per job and wave).
//   st = 0 -> 0, 1 -> 1.0, 2 -> t (a NaN row carries t = NaN, see scale_pack3): two bit-field extracts, an and, an and-or, an and
__device__ inline double unpack_bits(uint32_t word, int pos, double t) {
    const int m1 = __builtin_amdgcn_sbfe((int)word, (unsigned)pos, 1u);      // -1 where bit 0 of the state is set
    const int m2 = __builtin_amdgcn_sbfe((int)word, (unsigned)pos + 1u, 1u);  // -1 where bit 1 is set
    const unsigned long long tb = (unsigned long long)__double_as_longlong(t);
    const uint32_t hi = ((uint32_t)(tb >> 32) & (uint32_t)m2) | ((uint32_t)m1 & 0x3FF00000u);
    const uint32_t lo = (uint32_t)tb & (uint32_t)m2;
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// ---------------------------------------------------------------------------
// K1: stage A -- HBM-streaming L-axis contraction + per-channel min-max scale.
//   grid  : n_walks * n_slabs workgroups; workgroup = (walk, slab of 64*VEC channels).
//           A walk is a run of consecutive jobs streamed one after the other by the same
//           workgroup: a single job, or -- FUSED -- the parts of a protein whose parts tile it
//           exactly and whose last domain is the whole protein (what RecCut emits,
//           src/fingerprint.py:103-107).  While the parts stream by, a second accumulator set
//           (the whole protein's cosine table and first row) collects the whole-protein
//           coefficients, so every embedding row is read once instead of twice.
//   block : WAVES waves; wave w streams rows w, w+WAVES, ... of every piece of the job,
//           UNROLL rows (UNROLL x 16 B per lane) in flight; the cosines of a row are
//           wave-uniform and come through the scalar cache (s_load) into SGPR operands.
//   out   : yprime[(job*N + j) * ldy + col]  float64, j < N (padding columns = 0)
// ---------------------------------------------------------------------------
struct Walk {
    uint32_t job_begin;  // first job (chunk relative)
    uint32_t n_parts;    // consecutive jobs streamed by this workgroup
    int32_t whole_job;   // FUSED: the whole-protein job all parts feed (chunk relative), else -1
    uint32_t reserved;
};

// (second launch bound = waves per SIMD the register allocation must allow: the fused variant of the main
//  configuration sits at the 96-VGPR edge -- 5 waves per SIMD -- and is held there)
template <typename T, int N, int VEC, int WAVES, int UNROLL, bool FUSED>
__global__ __launch_bounds__(WAVES * 64, (FUSED && N == 3 && VEC == 4 && UNROLL == 4) ? 5 : 1) void stage_a_kernel(const JobA* __restrict__ jobs,
                                                              const Walk* __restrict__ walks,
                                                              const PieceA* __restrict__ pieces,
                                                              char* __restrict__ yprime, int64_t job_bytes, int packed,
                                                              int n_cols, int64_t ld, int ldy, int n_slabs, InvTab<N> inv,
                                                              unsigned long long* __restrict__ degenerate) {
    // yprime: per job either N rows of ldy float64, or (packed, N = 3) ldy float64 t values followed by
    // ldy state bytes (finish_channel)
    constexpr int NK = N - 1;
    __shared__ double red[WAVES][NK * VEC][64];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const Walk wk = walks[blockIdx.x / (uint32_t)n_slabs];
    const int slab = (int)(blockIdx.x % (uint32_t)n_slabs);
    const int col0 = (slab * 64 + lane) * VEC;
    const int colc = (col0 < n_cols) ? col0 : 0;  // out-of-range lanes stream column 0 and are discarded
    const bool has_w = FUSED && wk.whole_job >= 0;

    double wacc[FUSED ? NK : 1][VEC];
    double wref[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        wref[v] = 0.0;
#pragma unroll
        for (int k = 0; k < (FUSED ? NK : 1); ++k) wacc[k][v] = 0.0;
    }
    if (has_w) {
        auto w0 = load_raw<T, VEC>(reinterpret_cast<const T*>(jobs[wk.job_begin].w_ref) + colc);
#pragma unroll
        for (int v = 0; v < VEC; ++v) wref[v] = raw_elem<T, VEC>(w0, v);
    }

    // LDS combine of the waves' partial sums in a fixed order + the per-channel epilogue
    auto finish_job = [&](const double (&part)[NK][VEC], uint32_t job_id, bool first) {
        if (!first) __syncthreads();  // the previous epilogue has finished reading `red`
#pragma unroll
        for (int k = 0; k < NK; ++k)
#pragma unroll
            for (int v = 0; v < VEC; ++v) red[wave][k * VEC + v][lane] = part[k][v];
        __syncthreads();
        for (int cl = threadIdx.x; cl < 64 * VEC; cl += WAVES * 64) {
            const int ln = cl / VEC, v = cl % VEC;
            const int col = slab * 64 * VEC + cl;
            if (col >= ldy) continue;
            double f[NK];
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                double s = red[0][k * VEC + v][ln];
#pragma unroll
                for (int w = 1; w < WAVES; ++w) s += red[w][k * VEC + v][ln];
                f[k] = s;
            }
            char* __restrict__ jb = yprime + (size_t)job_id * job_bytes;
            finish_channel<N>(f, inv, reinterpret_cast<double*>(jb) + col, ldy, col >= n_cols, packed != 0,
                              jb + (size_t)ldy * sizeof(double) + col, degenerate);
        }
    };

    for (uint32_t part = 0; part < wk.n_parts; ++part) {
        const uint32_t job_id = wk.job_begin + part;
        const JobA job = jobs[job_id];
        const PieceA* __restrict__ pc = pieces + job.piece_begin;
        const CosTab bt = cos_tab(job.basis);
        const CosTab wt = cos_tab(job.w_basis);

        double acc[NK][VEC];
        double ref[VEC];
#pragma unroll
        for (int k = 0; k < NK; ++k)
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[k][v] = 0.0;
        {
            auto r0 = load_raw<T, VEC>(reinterpret_cast<const T*>(pc[0].ptr) + colc);
#pragma unroll
            for (int v = 0; v < VEC; ++v) ref[v] = raw_elem<T, VEC>(r0, v);
        }

        // (the test of `has_w` stays outside the row loops: a branch per row would pin the whole protein's cosine load next to
        //  its use and expose its latency row after row)
        auto stream_piece = [&](auto hw_tag, const PieceA& piece) {
            constexpr bool HW = decltype(hw_tag)::value;
            const T* __restrict__ base = reinterpret_cast<const T*>(piece.ptr) + colc;
            const CosTab btp = bt + (size_t)piece.t0 * NK;
            const CosTab wtp = wt + (size_t)piece.w0 * NK;
            auto row_update = [&](const typename Raw<T, VEC>::type& x, uint32_t r) {
                const CosTab c = btp + (size_t)r * NK;
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const double d = raw_elem<T, VEC>(x, v) - ref[v];
#pragma unroll
                    for (int k = 0; k < NK; ++k) acc[k][v] = fma(c[k], d, acc[k][v]);
                }
                if constexpr (HW) {
                    const CosTab cw = wtp + (size_t)r * NK;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const double dw = raw_elem<T, VEC>(x, v) - wref[v];
#pragma unroll
                        for (int k = 0; k < NK; ++k) wacc[FUSED ? k : 0][v] = fma(cw[k], dw, wacc[FUSED ? k : 0][v]);
                    }
                }
            };
            // this wave's share of the piece's rows, UNROLL loads in flight
            uint32_t r = (uint32_t)wave;
            for (; r + (UNROLL - 1) * WAVES < piece.n_rows; r += UNROLL * WAVES) {
                typename Raw<T, VEC>::type xv[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) xv[u] = load_raw<T, VEC>(base + (size_t)(r + u * WAVES) * ld);
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) row_update(xv[u], r + u * WAVES);
            }
            for (; r < piece.n_rows; r += WAVES) {
                auto x1 = load_raw<T, VEC>(base + (size_t)r * ld);
                row_update(x1, r);
            }
        };
        for (uint32_t p = 0; p < job.n_pieces; ++p) {
            const PieceA piece = pc[p];
            if (FUSED && has_w) stream_piece(std::integral_constant<bool, FUSED>{}, piece);
            else stream_piece(std::false_type{}, piece);
        }
        finish_job(acc, job_id, part == 0);
    }
    if constexpr (FUSED) {
        if (has_w) finish_job(wacc, (uint32_t)wk.whole_job, false);
    }
}

// ---------------------------------------------------------------------------
// K1s: stage A split over the rows -- for calls too small to fill the chip (a protein per call, the reference's calling
// pattern: 2 layers x 5 slabs = 10 workgroups would stream 5 MB on 10 of 256 CUs).  Workgroup = (job, chunk of
// `chunk_rows` rows, slab): same arithmetic as stage_a_kernel (first-row shift of the JOB, the job's cosine table), but the
// partial sums go to a scratch; stage_a_combine_kernel adds the chunks in order (deterministic) and runs the epilogue.
// ---------------------------------------------------------------------------
template <typename T, int N, int VEC, int WAVES, int UNROLL>
__global__ __launch_bounds__(WAVES * 64) void stage_a_split_kernel(const JobA* __restrict__ jobs, const PieceA* __restrict__ pieces,
                                                                    double* __restrict__ partial, int n_chunks, uint32_t chunk_rows,
                                                                    int n_cols, int64_t ld, int ldy, int n_slabs) {
    constexpr int NK = N - 1;
    __shared__ double red[WAVES][NK * VEC][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t slab = blockIdx.x % (uint32_t)n_slabs;
    const uint32_t jc = blockIdx.x / (uint32_t)n_slabs;
    const uint32_t chunk = jc % (uint32_t)n_chunks, job_id = jc / (uint32_t)n_chunks;
    const int col0 = ((int)slab * 64 + lane) * VEC;
    const int colc = (col0 < n_cols) ? col0 : 0;
    const JobA job = jobs[job_id];
    const PieceA* __restrict__ pc = pieces + job.piece_begin;
    const uint32_t lo = chunk * chunk_rows;
    const uint32_t hi = min(job.n_rows, lo + chunk_rows);  // this workgroup's rows of the job: [lo, hi)
    double acc[NK][VEC];
    double ref[VEC];
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[k][v] = 0.0;
    if (lo < hi) {
        {
            auto r0 = load_raw<T, VEC>(reinterpret_cast<const T*>(pc[0].ptr) + colc);
#pragma unroll
            for (int v = 0; v < VEC; ++v) ref[v] = raw_elem<T, VEC>(r0, v);
        }
        const CosTab bt = cos_tab(job.basis);
        for (uint32_t p = 0; p < job.n_pieces; ++p) {
            const PieceA piece = pc[p];
            const uint32_t a = max(lo, piece.t0), b = min(hi, piece.t0 + piece.n_rows);  // job rows of this piece in the chunk
            if (a >= b) continue;
            const T* __restrict__ base = reinterpret_cast<const T*>(piece.ptr) + colc;
            auto row_update = [&](const typename Raw<T, VEC>::type& x, uint32_t t) {  // t = row of the job
                const CosTab c = bt + (size_t)t * NK;
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const double d = raw_elem<T, VEC>(x, v) - ref[v];
#pragma unroll
                    for (int k = 0; k < NK; ++k) acc[k][v] = fma(c[k], d, acc[k][v]);
                }
            };
            uint32_t t = a + (uint32_t)wave;
            for (; t + (UNROLL - 1) * WAVES < b; t += UNROLL * WAVES) {
                typename Raw<T, VEC>::type xv[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) xv[u] = load_raw<T, VEC>(base + (size_t)(t + u * WAVES - piece.t0) * ld);
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) row_update(xv[u], t + u * WAVES);
            }
            for (; t < b; t += WAVES) {
                auto x1 = load_raw<T, VEC>(base + (size_t)(t - piece.t0) * ld);
                row_update(x1, t);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
        for (int v = 0; v < VEC; ++v) red[wave][k * VEC + v][lane] = acc[k][v];
    __syncthreads();
    for (int cl = threadIdx.x; cl < 64 * VEC; cl += WAVES * 64) {
        const int ln = cl / VEC, v = cl % VEC;
        const int col = (int)slab * 64 * VEC + cl;
        if (col >= ldy) continue;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            double sum = red[0][k * VEC + v][ln];
#pragma unroll
            for (int w = 1; w < WAVES; ++w) sum += red[w][k * VEC + v][ln];
            partial[((size_t)jc * NK + k) * ldy + col] = sum;
        }
    }
}

template <int N>
__global__ __launch_bounds__(256) void stage_a_combine_kernel(const double* __restrict__ partial, int n_chunks, char* __restrict__ yprime,
                                                               int64_t job_bytes, int packed, int n_cols, int ldy, InvTab<N> inv,
                                                               unsigned long long* __restrict__ degenerate) {
    constexpr int NK = N - 1;
    const int col = blockIdx.x * 256 + threadIdx.x;
    const uint32_t job_id = blockIdx.y;
    if (col >= ldy) return;
    double f[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) f[k] = 0.0;
    for (int c = 0; c < n_chunks; ++c)
#pragma unroll
        for (int k = 0; k < NK; ++k) f[k] += partial[(((size_t)job_id * n_chunks + c) * NK + k) * ldy + col];
    char* __restrict__ jb = yprime + (size_t)job_id * job_bytes;
    finish_channel<N>(f, inv, reinterpret_cast<double*>(jb) + col, ldy, col >= n_cols, packed != 0,
                      jb + (size_t)ldy * sizeof(double) + col, degenerate);
}

// ---------------------------------------------------------------------------
// Shared epilogue helper: trunc(127 z) with NaN / out-of-range -> 0
// ((ddct*127).astype('int8'), src/fingerprint.py:195; x86 numpy gives 0 for NaN).
// ---------------------------------------------------------------------------
__device__ inline int8_t quant127(double num, double den, bool bad) {
    const double v = (num / den) * 127.0;
    return (!bad && v >= 0.0 && v <= 127.0) ? (int8_t)(int)v : (int8_t)0;
}

// ---------------------------------------------------------------------------
// K2 (MFMA): stage B -- rows (job, j) x K = D contraction against St (D x m), then per-row
// min-max scale and int8 truncation.  v_mfma_f64_16x16x4_f64; workgroup = 4 waves x 16 rows,
// all NT column tiles per wave; St staged through LDS in 32-row chunks (row stride CP+4
// doubles keeps the four k-groups of a ds_read_b64 on distinct banks).
//   f64 MFMA layouts (cdna_hip_programming.md section 3): A[i = l&15][k = l>>4],
//   B[k = l>>4][j = l&15], C/D reg i: row = (l>>4) + 4 i, col = l & 15.
// ---------------------------------------------------------------------------
constexpr int kBWaves = 4;  // waves (16-row tiles) per stage-B workgroup
constexpr int kBKRows = 16;  // K rows of St per LDS stage
// Register budget as waves per SIMD: up to 5 column tiles (m <= 80, the reference's setting) fit 96 VGPRs with
// 16-row LDS stages -- the footprint of a fused stage-A wave, so a stage-B wave fits wherever one of those leaves.
constexpr int stage_b_min_waves(int nt) { return nt <= 5 ? 5 : (nt <= 7 ? 3 : 2); }
template <int NT, bool PACKED, int KB = kBKRows, int MINW = stage_b_min_waves(NT)>
__global__ __launch_bounds__(kBWaves * 64, MINW) void stage_b_mfma_kernel(const char* __restrict__ ypb, int64_t job_bytes,
                                                            int64_t n_rows_total, int ldy, const double* __restrict__ st,
                                                            const JobB* __restrict__ jobs, int n, int m,
                                                            int8_t* __restrict__ out) {
    constexpr int CP = NT * 16;
    constexpr int LDS_LD = CP + 4;
    constexpr int NQ = KB / 16;     // 16-deep A fragments per stage
    __shared__ double bs[KB][LDS_LD];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane >> 4, r16 = lane & 15;
    constexpr int BT = kBWaves * 64;  // threads per workgroup
    const int64_t row0 = (int64_t)blockIdx.x * (kBWaves * 16) + wave * 16;
    int64_t arow = row0 + r16;
    if (arow >= n_rows_total) arow = n_rows_total - 1;
    // A operand source: row (job, j) of Y'.  Plain: float64 row.  PACKED (n = 3): the job's t row + state bytes.
    const int64_t ajob = arow / n;
    const int aj = (int)(arow - ajob * n);
    const char* __restrict__ jbase = ypb + (size_t)ajob * job_bytes;
    const double* __restrict__ ap = reinterpret_cast<const double*>(jbase) + (PACKED ? 0 : (size_t)aj * ldy) + 4 * g;
    const uint8_t* __restrict__ cp8 = reinterpret_cast<const uint8_t*>(jbase) + (size_t)ldy * sizeof(double) + 4 * g;
    // raw A fragments are fetched one step ahead and decoded only when they are consumed, so the wait for
    // them sits a whole step (40 MFMAs) after their issue
    struct RawA {
        v4d t;
        uint32_t c;
    };
    auto fetch_a = [&](int k) -> RawA {
        RawA r;
        r.t = *reinterpret_cast<const v4d*>(ap + k);
        r.c = 0;
        if constexpr (PACKED) r.c = *reinterpret_cast<const uint32_t*>(cp8 + k);
        return r;
    };
    auto decode_a = [&](const RawA& r) -> v4d {
        v4d t = r.t;
        if constexpr (PACKED) {
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] = unpack_y((r.c >> (8 * q + 2 * aj)) & 3u, t[q]);
        }
        return t;
    };

    v4d acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};

    // Software pipeline: the St chunk and the A fragments of step kb+1 are fetched into registers
    // while the MFMAs of step kb run, so no global-memory latency sits between two barriers.
    constexpr int PER = (KB * CP / 2 + BT - 1) / BT;  // v2d pieces of an St chunk per thread
    v2d nxt[PER];
    auto fetch_st = [&](int kb) {
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const int i = threadIdx.x + e * BT;
            if (i < KB * CP / 2) {
                const int kk = i / (CP / 2), cc = (i % (CP / 2)) * 2;
                nxt[e] = *reinterpret_cast<const v2d*>(st + (size_t)(kb + kk) * CP + cc);
            }
        }
    };
    fetch_st(0);
    RawA araw[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) araw[q] = fetch_a(16 * q);
    for (int kb = 0; kb < ldy; kb += KB) {
        __syncthreads();  // the MFMAs of the previous step have read bs
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const int i = threadIdx.x + e * BT;
            if (i < KB * CP / 2) {
                const int kk = i / (CP / 2), cc = (i % (CP / 2)) * 2;
                bs[kk][cc] = nxt[e][0];
                bs[kk][cc + 1] = nxt[e][1];
            }
        }
        __syncthreads();
        v4d cur[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) cur[q] = decode_a(araw[q]);
        if (kb + KB < ldy) {
            fetch_st(kb + KB);
#pragma unroll
            for (int q = 0; q < NQ; ++q) araw[q] = fetch_a(kb + KB + 16 * q);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double a = cur[q][r];
                const int kk = q * 16 + 4 * g + r;
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bs[kk][t * 16 + r16], acc[t], 0, 0, 0);
            }
        }
    }

#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t grow = row0 + g + 4 * i;
        double mn = INFINITY, mx = -INFINITY;
        int bad = 0;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int col = t * 16 + r16;
            if (col < m) {
                const double v = acc[t][i];
                bad |= (v != v) ? 1 : 0;
                mn = fmin(mn, v);
                mx = fmax(mx, v);
            }
        }
#pragma unroll
        for (int s = 1; s < 16; s <<= 1) {
            mn = fmin(mn, __shfl_xor(mn, s));
            mx = fmax(mx, __shfl_xor(mx, s));
            bad |= __shfl_xor(bad, s);
        }
        if (grow < n_rows_total) {
            const int64_t job = grow / n;
            const int j = (int)(grow - job * n);
            int8_t* __restrict__ o = out + jobs[job].out_off + (int64_t)j * m;
            const double den = mx - mn;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int col = t * 16 + r16;
                if (col < m) o[col] = quant127(acc[t][i] - mn, den, bad != 0);  // (written through: no difference)
            }
        }
    }
}

// ---------------------------------------------------------------------------
// K2 (VALU): same result as the MFMA kernel, one workgroup per job, plain FMAs.
// Kept as the cross-check of the MFMA fragment layout and for A/B timing.
// ---------------------------------------------------------------------------
#ifndef DCTFP_TEMPLATES_ONLY   // (a plain kernel: defined in dctfp.hip only, the kernel-family units see the templates)
__global__ __launch_bounds__(1024) void stage_b_valu_kernel(const double* __restrict__ yp, int ldy, int n_cols,
                                                             const double* __restrict__ st, int cp,
                                                             const JobB* __restrict__ jobs, int n, int m,
                                                             int8_t* __restrict__ out) {
    // 1024 threads = 4 groups x 256: group kg takes every fourth 64-channel stretch of a staged chunk, so the D-long sums
    // of a job -- the whole latency of a small call -- run four abreast; the partial sums meet in LDS in group order.
    constexpr int DCH = 256;  // channels staged per pass
    constexpr int KG = 4;
    __shared__ double ys[DCTFP_MAX_N_K][DCH];
    __shared__ double part[KG][DCTFP_MAX_N_K * DCTFP_MAX_M_K];
    __shared__ double bl[DCTFP_MAX_N_K * DCTFP_MAX_M_K];
    const int job = blockIdx.x;
    const double* __restrict__ yj = yp + (size_t)job * n * ldy;
    const int n_out = n * m;
    const int kg = threadIdx.x >> 8, tx = threadIdx.x & 255;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int d0 = 0; d0 < n_cols; d0 += DCH) {
        const int dn = min(DCH, n_cols - d0);
        __syncthreads();
        for (int i = threadIdx.x; i < n * DCH; i += 1024) {
            const int j = i / DCH, d = i % DCH;
            ys[j][d] = (d < dn) ? yj[(size_t)j * ldy + d0 + d] : 0.0;
        }
        __syncthreads();
        const int da = kg * (DCH / KG), db = min(dn, da + DCH / KG);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int o = tx + s * 256;
            if (o < n_out) {
                const int j = o / m, c = o % m;
                double a0 = 0.0, a1 = 0.0;
                int d = da;
                for (; d + 1 < db; d += 2) {
                    a0 = fma(ys[j][d], st[(size_t)(d0 + d) * cp + c], a0);
                    a1 = fma(ys[j][d + 1], st[(size_t)(d0 + d + 1) * cp + c], a1);
                }
                if (d < db) a0 = fma(ys[j][d], st[(size_t)(d0 + d) * cp + c], a0);
                acc[s] += a0 + a1;
            }
        }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int o = tx + s * 256;
        if (o < n_out) part[kg][o] = acc[s];
    }
    __syncthreads();
    for (int o = threadIdx.x; o < n_out; o += 1024) bl[o] = ((part[0][o] + part[1][o]) + part[2][o]) + part[3][o];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = wave; j < n; j += 16) {
        double mn = INFINITY, mx = -INFINITY;
        int bad = 0;
        for (int c = lane; c < m; c += 64) {
            const double v = bl[j * m + c];
            bad |= (v != v) ? 1 : 0;
            mn = fmin(mn, v);
            mx = fmax(mx, v);
        }
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            mn = fmin(mn, __shfl_xor(mn, s));
            mx = fmax(mx, __shfl_xor(mx, s));
            bad |= __shfl_xor(bad, s);
        }
        int8_t* __restrict__ o = out + jobs[job].out_off + (int64_t)j * m;
        const double den = mx - mn;
        for (int c = lane; c < m; c += 64) o[c] = quant127(bl[j * m + c] - mn, den, bad != 0);
    }
}
#endif

// ---------------------------------------------------------------------------
// K2s: stage B of a small call (a protein per call: one or two jobs).  One workgroup per job spent 46 us on the D-long
// sums -- three quarters of the call's GPU time -- so the channels are spread over workgroups of 64: each adds its slab's
// share of the job's n x m block (FROM_SPLIT: after adding the row chunks of stage_a_split_kernel and scaling its 64
// channels itself, which saves the combine launch), stage_b_finish_kernel adds the slabs in slab order (deterministic),
// scales the rows and writes the int8.
// ---------------------------------------------------------------------------
constexpr int kSlabChannels = 64;

template <bool FROM_SPLIT>
__global__ __launch_bounds__(256) void stage_b_slab_kernel(const double* __restrict__ yp, int ldy, const double* __restrict__ partial,
                                                            int n_chunks, InvTab<3> inv, unsigned long long* __restrict__ degenerate,
                                                            int n_cols, const double* __restrict__ st, int cp, int n, int m,
                                                            double* __restrict__ zpart) {
    __shared__ double ys[DCTFP_MAX_N_K][kSlabChannels];
    const int job = blockIdx.y, ks = blockIdx.x;
    const int d0 = ks * kSlabChannels;
    const int dn = min(kSlabChannels, n_cols - d0);
    if constexpr (FROM_SPLIT) {  // n = 3: my channel's chunk sums -> scaled values
        if (threadIdx.x < kSlabChannels) {
            const int d = d0 + (int)threadIdx.x;
            double f[2] = {0.0, 0.0};
            if (d < ldy)
                for (int c = 0; c < n_chunks; ++c)
#pragma unroll
                    for (int k = 0; k < 2; ++k) f[k] += partial[(((size_t)job * n_chunks + c) * 2 + k) * ldy + d];
            double z[3];
            scale_channel<3>(f, inv, d >= n_cols, z, degenerate);
#pragma unroll
            for (int j = 0; j < 3; ++j) ys[j][threadIdx.x] = z[j];
        }
    } else {
        const double* __restrict__ yj = yp + (size_t)job * n * ldy;
        for (int i = threadIdx.x; i < n * kSlabChannels; i += 256) {
            const int j = i / kSlabChannels, t = i % kSlabChannels;
            ys[j][t] = (t < dn) ? yj[(size_t)j * ldy + d0 + t] : 0.0;
        }
    }
    __syncthreads();
    const int n_out = n * m;
    double* __restrict__ zp = zpart + ((size_t)job * gridDim.x + ks) * n_out;
    for (int o = threadIdx.x; o < n_out; o += 256) {
        const int j = o / m, c = o % m;
        const double* __restrict__ sc = st + (size_t)d0 * cp + c;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int t = 0;
        for (; t + 3 < dn; t += 4) {
            a0 = fma(ys[j][t], sc[(size_t)t * cp], a0);
            a1 = fma(ys[j][t + 1], sc[(size_t)(t + 1) * cp], a1);
            a2 = fma(ys[j][t + 2], sc[(size_t)(t + 2) * cp], a2);
            a3 = fma(ys[j][t + 3], sc[(size_t)(t + 3) * cp], a3);
        }
        for (; t < dn; ++t) a0 = fma(ys[j][t], sc[(size_t)t * cp], a0);
        zp[o] = (a0 + a1) + (a2 + a3);
    }
}

#ifndef DCTFP_TEMPLATES_ONLY   // (a plain kernel: defined in dctfp.hip only, the kernel-family units see the templates)
__global__ __launch_bounds__(256) void stage_b_finish_kernel(const double* __restrict__ zpart, int n_kslabs, const JobB* __restrict__ jobs,
                                                              int n, int m, int8_t* __restrict__ out) {
    __shared__ double bl[DCTFP_MAX_N_K * DCTFP_MAX_M_K];
    const int job = blockIdx.x;
    const int n_out = n * m;
    const double* __restrict__ zj = zpart + (size_t)job * n_kslabs * n_out;
    for (int o = threadIdx.x; o < n_out; o += 256) {
        double sum = 0.0;
        for (int ks = 0; ks < n_kslabs; ++ks) sum += zj[(size_t)ks * n_out + o];
        bl[o] = sum;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = wave; j < n; j += 4) {
        double mn = INFINITY, mx = -INFINITY;
        int bad = 0;
        for (int c = lane; c < m; c += 64) {
            const double v = bl[j * m + c];
            bad |= (v != v) ? 1 : 0;
            mn = fmin(mn, v);
            mx = fmax(mx, v);
        }
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            mn = fmin(mn, __shfl_xor(mn, s));
            mx = fmax(mx, __shfl_xor(mx, s));
            bad |= __shfl_xor(bad, s);
        }
        int8_t* __restrict__ o = out + jobs[job].out_off + (int64_t)j * m;
        const double den = mx - mn;
        for (int c = lane; c < m; c += 64) o[c] = quant127(bl[j * m + c] - mn, den, bad != 0);
    }
}
#endif

// ---------------------------------------------------------------------------
// K3: stage A and stage B in ONE kernel ("walk" kernel; n = 3, 64 < m <= 80; 4 channels per lane: float32 rows read as
// 16 B per lane, float16 / bfloat16 rows as 8 B).
//
// The two-kernel path sends Y' (9 B per channel and job) through HBM.  That is 0.45 % of the bytes at the headline
// shape but costs 5 % there and up to 30 % on 25-row domains: writes beside a saturated read stream (DESIGN.md
// section 4, K1).  Here a workgroup owns ALL channels of its jobs, so the scaled channels never leave the CU: they are
// packed into the wave's own LDS slot, and every G jobs the workgroup contracts their rows against the stage-B basis
// with v_mfma_f64_4x4x4_4b_f64 and writes 240 bytes per job.  What reaches HBM is the int8 result alone.
//
//   grid  : one workgroup per run = a few consecutive walks (a walk = one job, or the parts of a protein + the
//           whole protein fed from the same rows, as in stage_a_kernel); the jobs of a run are consecutive.
//   block : S = ceil(D / 256) waves, no row split.  Wave w owns the 128 channel PAIRS (d, D-1-d), d in [128 w, 128 w + 128):
//           lanes 0..31 stream channels [128 w, 128 w + 128) of every row, lanes 32..63 their mirror images (two 512-byte
//           segments per row), UNROLL rows in flight, cosines through the scalar cache.  No barrier per job -- the
//           epilogue of a job is wave-private.
//   flush : every G jobs (or at the end of the run), no workgroup barrier (tickets, see its end); stage B in even / odd halves (see "flush" in the kernel): with
//           u = y[d] + y[D-1-d], v = y[d] - y[D-1-d]: ZE[c] = sum u E[d][c], ZO[c] = sum v O[d][c] over the D/2 pairs and
//           the m/2 left columns, Z[c] = ZE + ZO, Z[m-1-c] = ZE - ZO.  v_mfma_f64_4x4x4_4b_f64 (measured 16.8 cycles,
//           75 TFLOP/s with 8 independent accumulators; tools/microbench/mfma_f64_probe.hip): per block b = (lane >> 2) & 3
//               A[i = lane & 3][k = lane >> 4],  B[k = lane >> 4][j = lane & 3],  D[i = lane >> 4][j = lane & 3]
//           The 4 rows i of a tile are the (up to) 4 jobs of the flush, one tile per resampled position j = 0, 1, 2 and
//           16-slot group: D lane = Z part of job lane >> 4, slot 16 c + (lane & 15).  Each wave contracts its own 128
//           pairs (K split over the waves); the partial 3 x 80 blocks are summed through LDS in wave order
//           (deterministic), then ZE +- ZO, per row min / max, scale, x 127, truncate.
//   stf   : [E | O] (80 slots: E columns 0..39, O columns 0..39) in fragment order (host: get_st),
//           stf[((q * 4 + r) * NT + c) * 64 + lane] = Tab[16 q + 4 (lane >> 4) + r][16 c + (lane & 15)]  (q = 16-pair group,
//           r = k-step inside it, c = 16-slot group): the B operand of one MFMA is 512 contiguous bytes, the NT operands
//           of a k-step follow each other.  The pair order inside a group (k = 4 (lane >> 4) + r) is the one the packed
//           Y' is read in.
// ---------------------------------------------------------------------------
// Minimum and maximum over the lanes 0 .. 47 of a wave (a fingerprint row: m <= 80 values, two per lane in lanes < 40) by
// DPP rotations inside the 16-lane rows and three v_readlane per value -- a dependent chain of ~20 short instructions where
// the xor-shuffle (ds_bpermute: a trip through the LDS crossbar per level and value) took most of the 1 500+ cycles a
// fingerprint row cost the wave that wrote it.  Lanes without a value pass +inf / -inf.
template <int CTRL>
__device__ inline double dpp_rotate(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ inline double lane_value(double v, int src_lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src_lane), __builtin_amdgcn_readlane(__double2loint(v), src_lane));
}
__device__ inline void wave_min_max48(double& mn, double& mx) {
    // row_ror:8, 4, 2, 1 -> every lane of a 16-lane row holds the row's extremes
    mn = fmin(mn, dpp_rotate<0x128>(mn));
    mx = fmax(mx, dpp_rotate<0x128>(mx));
    mn = fmin(mn, dpp_rotate<0x124>(mn));
    mx = fmax(mx, dpp_rotate<0x124>(mx));
    mn = fmin(mn, dpp_rotate<0x122>(mn));
    mx = fmax(mx, dpp_rotate<0x122>(mx));
    mn = fmin(mn, dpp_rotate<0x121>(mn));
    mx = fmax(mx, dpp_rotate<0x121>(mx));
    mn = fmin(fmin(lane_value(mn, 0), lane_value(mn, 16)), lane_value(mn, 32));
    mx = fmax(fmax(lane_value(mx, 0), lane_value(mx, 16)), lane_value(mx, 32));
}

// Transposition of a 4 x 4 arrangement across the four 16-lane rows of a wave: in, lane row g holds r[h]; out, lane row g
// holds o[i] = (what lane row i held in r[g]), same lane inside the row.  v_permlane32_swap exchanges the upper half of its
// first operand with the lower half of the second, v_permlane16_swap the odd rows of the first with the even rows of the
// second (tools/microbench/permlane_swap_probe.hip): two of each per 32-bit register quartet.
__device__ inline void lane_rows_swap32(uint32_t& a, uint32_t& b) {
    const v2u32 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}
__device__ inline void lane_rows_swap16(uint32_t& a, uint32_t& b) {
    const v2u32 r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}
template <int NOUT>  // the first NOUT (2 or 4) of the transposed values are wanted
__device__ inline void lane_rows_transpose(const double (&r)[4], double (&o)[NOUT]) {
    uint32_t w[4][2];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        w[h][0] = (uint32_t)__double2loint(r[h]);
        w[h][1] = (uint32_t)__double2hiint(r[h]);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        lane_rows_swap32(w[0][c], w[2][c]);  // w0 = [r0.0 r0.1 r2.0 r2.1]   w2 = [r0.2 r0.3 r2.2 r2.3]   (rX.g = lane row g of r[X])
        lane_rows_swap32(w[1][c], w[3][c]);  // w1 = [r1.0 r1.1 r3.0 r3.1]   w3 = [r1.2 r1.3 r3.2 r3.3]
        lane_rows_swap16(w[0][c], w[1][c]);  // w0 = [r0.0 r1.0 r2.0 r3.0]   w1 = [r0.1 r1.1 r2.1 r3.1]
        if constexpr (NOUT > 2) lane_rows_swap16(w[2][c], w[3][c]);  // w2 = [r0.2 r1.2 r2.2 r3.2]   w3 = [r0.3 r1.3 r2.3 r3.3]
    }
#pragma unroll
    for (int i = 0; i < NOUT; ++i) o[i] = __hiloint2double((int)w[i][1], (int)w[i][0]);
}

// The row two overlapping windows share, as Embedding.embed_seq leaves it (src/embedding.py:185-187): float32 (old + new) / 2 --
// the sum rounded to float32, the halving exact -- the very expression of stitch_rows_kernel.
template <typename R>
__device__ inline R window_mean(const R& a, const R& b) {
    static_assert(std::is_same<R, v4f>::value, "windows are stitched in float32 (as dctfp_stitch)");
    return (R){(a[0] + b[0]) / 2.0f, (a[1] + b[1]) / 2.0f, (a[2] + b[2]) / 2.0f, (a[3] + b[3]) / 2.0f};
}

struct Run {
    uint32_t walk_begin;  // first Walk of this workgroup
    uint32_t n_walks;
    uint32_t job_begin;   // first job (the jobs of a run are consecutive)
    uint32_t n_jobs;
};


// Build-time knobs of the walk kernel (A/B builds: tools/build_variant.sh).
#ifndef DCTFP_WALK_MIN_WAVES
#define DCTFP_WALK_MIN_WAVES 4       // waves per SIMD the register allocation is held to (4 -> 128 VGPRs)
#endif
#ifndef DCTFP_WALK_B_DEPTH
#define DCTFP_WALK_B_DEPTH 0         // k-steps of stage-B fragments in flight during a flush: 0 = by shape (below), else 1, 2, 4
#endif

// Instrumented build (tools/walk_timeline.py; never the shipped library): every wave adds the time (s_memrealtime: the
// constant 100 MHz counter -- the shader clock moves with the power management, by 20 % between variants) it
// spends per phase to degenerate[1 + phase] -- 0 stream (job record -> last row accumulated), 1 epilogue, 2 flush contraction
// (unpack + MFMA), 3 the last arrivals' wait for the others (round 3: every wave's wait at the barrier before the cross-wave
// sum), 4 ticket, cross-wave sum + int8, 5 wait for the slots of the last flush group before the first write of the next (round
// 3: the barrier that freed them), 6 everything else, 7 wave lifetime; 8 waves, 9 jobs, 10 flushes.
#ifdef DCTFP_WALK_TIMELINE
#define DCTFP_TL_DECL                                   \
    uint64_t tl_prev = __builtin_amdgcn_s_memrealtime();    \
    const uint64_t tl_begin = tl_prev;                  \
    uint64_t tl_acc[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define DCTFP_TL_MARK(i)                                         \
    do {                                                         \
        const uint64_t now_ = __builtin_amdgcn_s_memrealtime();      \
        tl_acc[i] += now_ - tl_prev;                             \
        tl_prev = now_;                                          \
    } while (0)
#define DCTFP_TL_COUNT(i) tl_acc[i] += 1
#define DCTFP_TL_ANCHOR(x) asm volatile("" ::"v"(x))
#else
#define DCTFP_TL_DECL
#define DCTFP_TL_MARK(i)
#define DCTFP_TL_COUNT(i)
#define DCTFP_TL_ANCHOR(x)
#endif

// WIN: builds that also take two-source pieces (PieceA::ptr2; float32 rows) -- builds of their own, because the fused variants sit
// exactly at their register budget and the second row stream costs the ordinary calls nothing this way.
template <typename T, int S, int G, int NT, int UNROLL, bool FUSED, bool MA = false, bool WIN = false>
__global__ __launch_bounds__(S * 64, S >= 10 ? 3 : DCTFP_WALK_MIN_WAVES) void walk_ab_kernel(const JobA* __restrict__ jobs, const JobB* __restrict__ jobb,
                                                          const Walk* __restrict__ walks, const Run* __restrict__ runs,
                                                          const PieceA* __restrict__ pieces, const double* __restrict__ stf,
                                                          int8_t* __restrict__ out, int n_cols, int64_t ld, int m,
                                                          InvTab<3> inv, unsigned long long* __restrict__ degenerate) {
    constexpr int VEC = 4, NK = 2;
    constexpr int WCH = 64 * VEC;  // channels per wave
    // per wave and slot: 256 t values (later reused for the wave's partial 3 x (NT*16) block) + 256 state bytes
    // (NT = 6 -- 80 < m <= 96, PROST's [3, 85] -- : the partial block is 288 values, the slot grows with it)
    constexpr int SLOT = 3 * NT * 16 > WCH ? 3 * NT * 16 : WCH;
    constexpr int HS = NT * 8;     // slots of the table's E half (and of its O half): 40 at NT = 5, 48 at NT = 6
    static_assert(NT == 5 || NT == 6, "five or six 16-column groups: [E | O] halves of 40 or 48 slots");
    __shared__ double lds_t[S][G][SLOT];
    __shared__ uint32_t lds_c[S][G][WCH / 4];
    // The flush has no workgroup barrier (see its end): `lds_arrived` counts the waves that have stored their partial blocks,
    // `lds_done` the jobs whose rows are written -- over all flushes of the workgroup so far.
    __shared__ uint32_t lds_arrived, lds_done;
    uint32_t flush_seq = 0, done_expected = 0;
    if (threadIdx.x == 0) {
        lds_arrived = 0;
        lds_done = 0;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // Ten waves on four SIMDs sit 3-3-2-2, and a SIMD issues its oldest wave first: the youngest wave of a three-wave SIMD falls
    // behind, and at every flush the other nine wait for it (tools/wave_wait_probe.py: waves 0-3 spent 24 % of their time at
    // that barrier, 4-7 16 %, 8-9 0.5 %; a fixed higher priority for the later waves only moved the role to waves 0 and 1).
    // The three waves w, w + 4, w + 8 of a SIMD therefore take turns at the issue priority, job by job: c4 +0.9 %.
    uint32_t prio_turn = (uint32_t)wave >> 2;
    const Run run = runs[blockIdx.x];
    // A wave owns 128 channel pairs (d, D-1-d): lanes 0..31 stream channels [128 w, 128 w + 128), lanes 32..63 their mirror
    // images -- two 512-byte segments of every row (as fast as one of 1 KiB: tools/microbench/read_ceiling.hip) -- so that
    // the even/odd fold of the flush finds both channels of a pair in the wave's own slots.
    const int half = n_cols >> 1;
    const int pair0 = wave * (WCH / 2) + VEC * (lane & 31);  // the first of my 4 pairs
    const bool mirror = lane >= 32;
    const bool pad = pair0 >= half;  // out-of-range lanes stream column 0 and are discarded
    const int colc = pad ? 0 : (mirror ? n_cols - VEC - pair0 : pair0);
    // D % 8 == 4: the last 4 channels before D/2 are 2 pairs; both lanes that read them hold all four Y', the flush takes
    // each pair once -- and the degenerate-channel counter must see each channel once
    auto channel_counts = [&](int v) { return !pad && pair0 + (mirror ? VEC - 1 - v : v) < half; };
    // 16-pair groups of this wave that hold real pairs
    const int n_q = min(WCH / 32, max(0, (half + 15) / 16 - wave * (WCH / 32)));
    typedef typename Raw<T, VEC>::type Rw;  // 4 channels per lane: 16 bytes of float32, 8 of float16 / bfloat16
    const int col_bytes = colc * (int)sizeof(T);       // my lane's offset inside every row
    const int ld_bytes = (int)(ld * (int64_t)sizeof(T));  // (the host sends only layers whose pieces stay below 2^31 bytes here)

    uint32_t pending = 0;            // jobs whose Y' sits in LDS
    uint32_t group_job = run.job_begin;  // job of slot 0
    DCTFP_TL_DECL

    // MA, rare path: bit e = my channel e differs somewhere in the job from the job's first row (an exactly constant
    // channel has no bit: its F must be exactly 0).  One more pass over the job's rows, by the whole wave.
    auto differs_from_first_row = [&](const JobA& jb) {
        const PieceA* __restrict__ pcs = pieces + jb.piece_begin;
        const Rw r0 = load_raw<T, VEC>(reinterpret_cast<const T*>(pcs[0].ptr) + colc);
        uint32_t dif = 0;
        for (uint32_t p = 0; p < jb.n_pieces; ++p) {
            const PieceA piece = pcs[p];
            const __amdgpu_buffer_rsrc_t rows = wave_buffer(piece.ptr);
            for (uint32_t r = 0; r < piece.n_rows; ++r) {
                const Rw x = buffer_load_raw<Rw, false>(rows, col_bytes, (int)r * ld_bytes);
#pragma unroll
                for (int e = 0; e < VEC; ++e) dif |= (raw_elem<T, VEC>(x, e) != raw_elem<T, VEC>(r0, e) ? 1u : 0u) << e;
            }
        }
        return dif;
    };

    for (uint32_t wi = 0; wi < run.n_walks; ++wi) {
        const Walk wk = walks[run.walk_begin + wi];
        const bool has_w = FUSED && wk.whole_job >= 0;
        const uint32_t n_walk_jobs = wk.n_parts + (has_w ? 1u : 0u);
        double wacc[FUSED ? NK : 1][VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v)
#pragma unroll
            for (int k = 0; k < (FUSED ? NK : 1); ++k) wacc[k][v] = 0.0;
        // rows of the whole protein = offset of the prefix sums behind its cosine table
        const uint32_t w_rows = has_w ? jobs[wk.whole_job].n_rows : 0u;
        uint32_t w_suspect = 0xfu;  // MA: bit e = my channel e was within the round-off bound of a constant channel in every part so far

        for (uint32_t part = 0; part < n_walk_jobs; ++part) {
            double f[NK][VEC];
            if constexpr (S >= 10) {
                const uint32_t turn = prio_turn % 3u;
                if (turn == 0) __builtin_amdgcn_s_setprio(0);
                else if (turn == 1) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(2);
                ++prio_turn;
            }
            DCTFP_TL_MARK(6);
            DCTFP_TL_COUNT(9);
            if (part < wk.n_parts) {
                // ---- stage A of one job: every row of this wave's 256 channels
                const JobA job = jobs[wk.job_begin + part];
                const PieceA* __restrict__ pc = pieces + job.piece_begin;
                if constexpr (MA) {
                // ---- the multiply-adds of stage A on the matrix pipe (fused walks of float32 rows).  The fused walks run the
                // package into its power limit; a float64 multiply-add costs 2.6 x less energy in an MFMA
                // than in v_fma_f64 (tools/microbench/power_pipes.hip, stage_a_pipes.hip).  One load instruction fetches 4
                // consecutive rows x 64 channels: lane (q, g) = (lane & 15, lane >> 4) reads row r + g, the 16 bytes lane
                // 16 h + q of the vector layout owns (chunk h = 0..3: the same two 512-byte segments per row and wave as
                // before, in 256-byte halves) -- which IS the B operand layout B[k = lane >> 4][j = lane & 3] of the four
                // 4 x 4 x 4 blocks, no shuffle.  The A operand A[i = lane & 3][k = lane >> 4] is one double per lane and 4-row
                // step: cos of {part k=1, part k=2, whole k=1, whole k=2} at row r + k, straight from the cosine tables.
                // D[i = lane >> 4][j]: 16 MFMAs per step leave 4 outputs x 256 channels in macc.  Per element that is one
                // v_cvt_f64_f32 and a sixteenth of an MFMA -- no subtraction: the cosines of a job sum to zero, so the first-row
                // shift of the vector path only matters for an exactly constant channel (which must give exactly 0, not the
                // round-off of sum c(t) x).  That case is caught afterwards: |F| below the round-off bound of a constant channel
                // -> the wave re-reads the job and compares (differs_from_first_row, above); no healthy channel comes near the bound.
                // Outcome: same bytes, a quarter of the vector instructions, 6-11 % more shader clock -- and the
                // same kernel time to 0.1 %: the fused walks do not hang on the clock.  Kept as an experiment (libdctfp_experiments.so,
                // option ab_mfma_a); at four waves per SIMD it spills, so A/B it in a build with -DDCTFP_WALK_MIN_WAVES=3.
                static_assert(!MA || (FUSED && sizeof(T) == 4 && VEC == 4 && UNROLL % 4 == 0), "matrix-pipe stage A: fused walks of float32 rows");
                const int mq = lane & 15, mg = lane >> 4;
                int mcol[4];  // my 16 bytes inside a row, per chunk
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const int p0 = wave * (WCH / 2) + 64 * (h & 1) + VEC * mq;
                    mcol[h] = (p0 >= half ? 0 : (h >= 2 ? n_cols - VEC - p0 : p0)) * (int)sizeof(T);
                }
                const int n_act = (half - wave * (WCH / 2) + 63) / 64;  // chunk pairs (h, h + 2) that hold channels: <= 0, 1, >= 2
                // macc[h][e]: lane (q, i = lane >> 4) holds output i (0, 1: the part's F1, F2; 2, 3: the whole protein's share of this
                // part) of channel colc(lane 16 h + q) + e
                double macc[4][4];
#pragma unroll
                for (int h = 0; h < 4; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e) macc[h][e] = 0.0;
                // (my channels' first row, for the round-off bound at the end: asked for now, its latency under the stream's)
                const Rw rp = load_raw<T, VEC>(reinterpret_cast<const T*>(pc[0].ptr) + colc);
                typedef const double __attribute__((address_space(1))) * GD;
                auto stream_piece = [&](auto nch_tag, const PieceA& piece) {
                    constexpr int NCH = decltype(nch_tag)::value;
                    const __amdgpu_buffer_rsrc_t rows = wave_buffer(piece.ptr);
                    const bool whole_lane = has_w && (lane & 2);  // (a walk without whole protein: outputs 2, 3 repeat 0, 1, unused)
                    const GD ap = (GD)(uintptr_t)((whole_lane ? job.w_basis + (size_t)piece.w0 * NK : job.basis + (size_t)piece.t0 * NK) +
                                                  mg * NK + (lane & 1));
                    auto load_step = [&](Rw (&x)[4], int lane_rows, uint32_t r) {
#pragma unroll
                        for (int h = 0; h < 4; ++h)
                            if ((h & 1) < NCH) x[h] = buffer_load_raw<Rw, true>(rows, mcol[h] + lane_rows, (int)r * ld_bytes);
                    };
                    auto step = [&](const Rw (&x)[4], double a) {
#pragma unroll
                        for (int h = 0; h < 4; ++h)
                            if ((h & 1) < NCH) {
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    macc[h][e] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, (double)x[h][e], macc[h][e], 0, 0, 0);
                            }
                    };
                    constexpr int STEPS = UNROLL / 4;
                    const int my_rows = mg * ld_bytes;
                    uint32_t r = 0;
                    for (; r + 4 * STEPS <= piece.n_rows; r += 4 * STEPS) {
                        Rw x[STEPS][4];
                        double a[STEPS];
#pragma unroll
                        for (int st = 0; st < STEPS; ++st) {
                            load_step(x[st], my_rows, r + 4 * st);
                            a[st] = ap[(size_t)(r + 4 * st) * NK];
                        }
#pragma unroll
                        for (int st = 0; st < STEPS; ++st) step(x[st], a[st]);
                    }
                    if (r < piece.n_rows) {  // the last 1-7 rows in one round of loads: the lanes past the end repeat the piece's
                        const int rem = (int)(piece.n_rows - r);  // last row with cosine 0
                        const int rr0 = min(mg, rem - 1), rr1 = min(4 + mg, rem - 1);
                        Rw x[2][4];
                        load_step(x[0], rr0 * ld_bytes, r);
                        const double a0 = ap[((int64_t)r + rr0 - mg) * NK];
                        double a1 = 0.0;
                        if (rem > 4) {
                            load_step(x[1], rr1 * ld_bytes, r);
                            a1 = ap[((int64_t)r + rr1 - mg) * NK];
                        }
                        step(x[0], mg < rem ? a0 : 0.0);
                        if (rem > 4) step(x[1], 4 + mg < rem ? a1 : 0.0);
                    }
                };
                for (uint32_t p = 0; p < job.n_pieces; ++p) {
                    const PieceA piece = pc[p];
                    if (n_act >= 2) stream_piece(std::integral_constant<int, 2>{}, piece);
                    else if (n_act == 1) stream_piece(std::integral_constant<int, 1>{}, piece);
                }
                // back to the vector layout: lane l = 16 g + q owns colc(l) + e = the channels of chunk g -> after the lane-row
                // transposition of macc[0..3][e] every lane holds the four outputs of its own channel e.
                uint32_t suspect = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double r4[4] = {macc[0][e], macc[1][e], macc[2][e], macc[3][e]};
                    double o[4];
                    lane_rows_transpose<4>(r4, o);
                    f[0][e] = o[0];
                    f[1][e] = o[1];
                    wacc[0][e] += o[2];
                    wacc[1][e] += o[3];
                    // round-off of an exactly constant channel: |sum c(t) x| <= |x| (L 2^-51 + L^2 2^-54); 16 x that
                    const double bound = fabs((double)rp[e]) * ((double)job.n_rows * (double)job.n_rows) * 0x1p-50;
                    suspect |= (fabs(o[0]) <= bound && fabs(o[1]) <= bound ? 1u : 0u) << e;
                }
                w_suspect &= suspect;  // a channel constant over the whole protein is constant in every part
                if (__builtin_amdgcn_ballot_w64(suspect != 0 && !pad) != 0) {
                    const uint32_t dif = differs_from_first_row(job);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (!((dif >> e) & 1)) f[0][e] = f[1][e] = 0.0;
                }
                } else {
                double ref[VEC];
#pragma unroll
                for (int k = 0; k < NK; ++k)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) f[k][v] = 0.0;
                // Lanes without channels (D = 640: half of the third wave) sit the stream out with their EXEC bit off.  They
                // used to stream column 0 and throw the result away: the same issue time either way -- but this kernel runs the
                // chip into its power limit (the shader clock falls from 2.38 GHz on whole-protein batches to 1.9 GHz on the
                // D = 640 database-build mix, tools/clock_probe.py), and a masked lane does not pay for 24 float64 operations per row.
                if (!pad) {
                {
                    Rw r0 = load_raw<T, VEC>(reinterpret_cast<const T*>(pc[0].ptr) + colc);
                    if constexpr (WIN) {  // a job that starts inside the overlap of two windows: its first row is their mean
                        const T* __restrict__ p2 = reinterpret_cast<const T*>(pc[0].ptr2);
                        if (p2) r0 = window_mean(load_raw<T, VEC>(p2 + colc), r0);
                    }
#pragma unroll
                    for (int v = 0; v < VEC; ++v) ref[v] = raw_elem<T, VEC>(r0, v);
                }
                // The stream of one piece, once with and once without the whole-protein accumulation: the test of `has_w`
                // must not sit inside the row loop -- a branch per row keeps the whole protein's cosine load (s_load) next
                // to its use, its latency exposed row after row (that was 20 % of the fused walks).
                // The whole protein is accumulated against the PART's first row, so that one subtraction per element feeds
                // both accumulator sets (24 instead of 28 float64 instructions per row and lane), and its own shift is
                // restored once per part:  sum_t cw(t) (x_t - r_w) = sum_t cw(t) (x_t - r_p) + (r_p - r_w) sum_t cw(t),
                // the last sum from the prefix sums behind the whole protein's table, r_w read again at the end of the part
                // (kept in registers through the stream it cost 5 more spilled registers per lane: the scratch WRITES beside
                // the row stream made the variant 3 % slower at D <= 1280 although it issues 9 % fewer instructions -- PMC:
                // profiles/r02/pmc_single_shift_*.md).  An exactly constant channel still gives exactly 0: both terms vanish.
                double cwsum[NK] = {0.0, 0.0};
                auto stream_piece = [&](auto hw_tag, auto two_tag, const PieceA& piece) {
                    constexpr bool HW = decltype(hw_tag)::value;
                    constexpr bool TWO = decltype(two_tag)::value;  // rows = the float32 mean of two windows' rows (PieceA::ptr2)
                    const __amdgpu_buffer_rsrc_t rows = wave_buffer(piece.ptr);
                    auto load_row = [&](uint32_t r) { return buffer_load_raw<Rw, true>(rows, col_bytes, (int)r * ld_bytes); };
                    const CosTab btp = cos_tab(job.basis) + (size_t)piece.t0 * NK;
                    const CosTab wtp = cos_tab(job.w_basis) + (size_t)piece.w0 * NK;
                    auto row_update = [&](const Rw& x, uint32_t r) {
                        const CosTab c = btp + (size_t)r * NK;
                        const CosTab cw = wtp + (size_t)r * NK;
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            const double d = raw_elem<T, VEC>(x, v) - ref[v];
#pragma unroll
                            for (int k = 0; k < NK; ++k) f[k][v] = fma(c[k], d, f[k][v]);
                            if constexpr (HW) {
#pragma unroll
                                for (int k = 0; k < NK; ++k) wacc[FUSED ? k : 0][v] = fma(cw[k], d, wacc[FUSED ? k : 0][v]);
                            }
                        }
                    };
                    uint32_t r = 0;
                    if constexpr (TWO) {
                        // UNROLL / 2 row PAIRS in flight (the same bytes as UNROLL rows), each pair reduced to its float32 mean --
                        // (old + new) / 2 as the reference's torch expression and stitch_rows_kernel do it -- before the promotion
                        // to float64: per KiB read, half the float64 work of an ordinary piece.
                        const __amdgpu_buffer_rsrc_t rows2 = wave_buffer(piece.ptr2);
                        auto load_row2 = [&](uint32_t rr) { return buffer_load_raw<Rw, true>(rows2, col_bytes, (int)rr * ld_bytes); };
                        constexpr int PAIRS = HW ? UNROLL / 4 : UNROLL / 2;  // (the fused variants have no registers to spare)
                        for (; r + PAIRS <= piece.n_rows; r += PAIRS) {
                            Rw xa[PAIRS], xb[PAIRS];
#pragma unroll
                            for (int u = 0; u < PAIRS; ++u) {
                                xa[u] = load_row(r + u);
                                xb[u] = load_row2(r + u);
                            }
#pragma unroll
                            for (int u = 0; u < PAIRS; ++u) row_update(window_mean(xa[u], xb[u]), r + u);
                        }
                        if (r < piece.n_rows) {  // the last 1 .. PAIRS - 1 pairs, their loads issued together
                            Rw xa[PAIRS - 1], xb[PAIRS - 1];
#pragma unroll
                            for (int u = 0; u < PAIRS - 1; ++u)
                                if (r + u < piece.n_rows) {
                                    xa[u] = load_row(r + u);
                                    xb[u] = load_row2(r + u);
                                }
#pragma unroll
                            for (int u = 0; u < PAIRS - 1; ++u)
                                if (r + u < piece.n_rows) row_update(window_mean(xa[u], xb[u]), r + u);
                        }
                    } else {
                    for (; r + UNROLL <= piece.n_rows; r += UNROLL) {
                        Rw xv[UNROLL];
#pragma unroll
                        for (int u = 0; u < UNROLL; ++u) xv[u] = load_row(r + u);
#pragma unroll
                        for (int u = 0; u < UNROLL; ++u) row_update(xv[u], r + u);
                    }
                    if constexpr (UNROLL > 4) {  // what is left of the piece: groups of 4 (one at most up to 8 rows in flight) ...
                        for (int g4 = 0; g4 < (UNROLL - 1) / 4; ++g4) {
                            if (r + 4 > piece.n_rows) break;
                            Rw xv[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) xv[u] = load_row(r + u);
#pragma unroll
                            for (int u = 0; u < 4; ++u) row_update(xv[u], r + u);
                            r += 4;
                        }
                    }
                    if (r < piece.n_rows) {  // ... and the last 1-3 rows, their loads issued together too
                        Rw xv[3];
#pragma unroll
                        for (int u = 0; u < 3; ++u)
                            if (r + u < piece.n_rows) xv[u] = load_row(r + u);
#pragma unroll
                        for (int u = 0; u < 3; ++u)
                            if (r + u < piece.n_rows) row_update(xv[u], r + u);
                    }
                    }  // !TWO
                    if constexpr (HW) {  // prefix sums past this piece's last and at its first row
                        const CosTab wpre = wtp + (size_t)w_rows * NK;
#pragma unroll
                        for (int k = 0; k < NK; ++k) cwsum[k] += wpre[(size_t)piece.n_rows * NK + k] - wpre[k];
                    }
                };
                for (uint32_t p = 0; p < job.n_pieces; ++p) {
                    const PieceA piece = pc[p];
                    if constexpr (WIN) {
                        if (piece.ptr2) {
                            if (FUSED && has_w) stream_piece(std::integral_constant<bool, FUSED>{}, std::true_type{}, piece);
                            else stream_piece(std::false_type{}, std::true_type{}, piece);
                            continue;
                        }
                    }
                    if (FUSED && has_w) stream_piece(std::integral_constant<bool, FUSED>{}, std::false_type{}, piece);
                    else stream_piece(std::false_type{}, std::false_type{}, piece);
                }
                if (FUSED && has_w) {
                    const Rw w0 = load_raw<T, VEC>(reinterpret_cast<const T*>(job.w_ref) + colc);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const double dr = ref[v] - raw_elem<T, VEC>(w0, v);
#pragma unroll
                        for (int k = 0; k < NK; ++k) wacc[FUSED ? k : 0][v] = fma(dr, cwsum[k], wacc[FUSED ? k : 0][v]);
                    }
                }
                }  // !pad
                }  // !MA
            } else if constexpr (MA) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    f[0][e] = wacc[0][e];
                    f[1][e] = wacc[1][e];
                }
                if (__builtin_amdgcn_ballot_w64(w_suspect != 0 && !pad) != 0) {
                    const uint32_t dif = differs_from_first_row(jobs[wk.whole_job]);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (!((dif >> e) & 1)) f[0][e] = f[1][e] = 0.0;
                }
            } else {
#pragma unroll
                for (int k = 0; k < NK; ++k)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) f[k][v] = wacc[FUSED ? k : 0][v];
            }

            // ---- epilogue of the job: scale each of my 4 channels, pack into my slot (wave-private, no barrier)
            DCTFP_TL_ANCHOR(f[0][0]);
            DCTFP_TL_ANCHOR(f[1][VEC - 1]);
            DCTFP_TL_MARK(0);
            {
                double tv[VEC];
                uint32_t c4 = 0;
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const double fk[NK] = {f[0][v], f[1][v]};
                    unsigned code;
                    scale_pack3(fk, inv, !channel_counts(v), tv[v], code, degenerate);
                    c4 |= code << (8 * v);
                    __builtin_amdgcn_sched_barrier(0);  // one channel at a time (register pressure)
                }
                int sl = lane;
                if constexpr (MA) asm volatile("" : "+v"(sl));
                if (pending == 0 && done_expected != 0) {  // first write of a flush group: the rows of the last one must be out
                    DCTFP_TL_MARK(1);
                    while (__hip_atomic_load(&lds_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != done_expected)
                        __builtin_amdgcn_s_sleep(1);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    DCTFP_TL_MARK(5);
                }
                *reinterpret_cast<v4d*>(&lds_t[wave][pending][VEC * sl]) = (v4d){tv[0], tv[1], tv[2], tv[3]};
                lds_c[wave][pending][sl] = c4;
            }
            ++pending;
            DCTFP_TL_MARK(1);
            const bool last = (wi + 1 == run.n_walks) && (part + 1 == n_walk_jobs);
            if (pending < (uint32_t)G && !last) continue;
            DCTFP_TL_COUNT(10);

            // ---- flush: stage B of the `pending` jobs in LDS
            // Both cosine factors of St[d][c] = sum_k cos_m(k, c) cos_D(k, d) are mirror (anti)symmetric:
            //   St[d][c] = E[d][c] + O[d][c],  St[D-1-d][c] = St[d][m-1-c] = E[d][c] - O[d][c]     (E: even k, O: odd k)
            // so with u = y[d] + y[D-1-d], v = y[d] - y[D-1-d] over the D/2 channel pairs and the m/2 left columns
            //   ZE[c] = sum_d u_d E[d][c],  ZO[c] = sum_d v_d O[d][c],  Z[c] = ZE[c] + ZO[c],  Z[m-1-c] = ZE[c] - ZO[c]
            // -- half the multiply-adds of the plain product.  The MFMAs of a flush are issue time the stream does not get
            // (tools/microbench/f64_pipes.hip): on 70..110-row jobs they were 17 % (D = 2560) of the kernel.
            // The table (`stf`, host: get_st) holds [E | O] in fragment order: 80 slots = E columns 0..39, O columns 0..39.
            // Both channels of a pair sit in the wave's own slot: entries 0..127 the channels d, 128..255 their mirrors in
            // ascending channel order (pair p of the wave <-> entries p and 128 + (p ^ 3)).
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // The 4 rows of an MFMA tile are the 4 jobs of the flush: lane (i = lane & 3, k = lane >> 4) unpacks job i's value
            // of pair k for row j, three tiles (j = 0, 1, 2) per column group -- 15 MFMAs per k-step whatever `pending` is
            // (rows of jobs that are not there are computed on stale slots and never stored).
            {
                // (NT = 5: slots 0..39 = E, 40..79 = O -- the middle column group is half and half; NT = 6: 0..47 = E, 48..95 = O)
                static_assert(G >= 1 && G <= 4, "one MFMA row per job of the flush");
                // (the lane index through an opaque copy: everything the flush derives from it -- slot, masks, LDS and fragment
                //  offsets -- would otherwise be hoisted to the top of the kernel and held in registers through the row stream,
                //  where the fused variants have none to spare: 3 spilled registers per lane, 42 MB of scratch per launch)
                int fl = lane;
                asm volatile("" : "+v"(fl));
                const int g4 = fl >> 4;
                const int gs = min(fl & 3, G - 1);                 // my job's slot
                const double fold_sign = (fl & 8) ? -1.0 : 1.0;  // blocks 2, 3 of the middle column group belong to O
                double acc[3][NT];
#pragma unroll
                for (int jr = 0; jr < 3; ++jr)
#pragma unroll
                    for (int c = 0; c < NT; ++c) acc[jr][c] = 0.0;
                // B fragments of k-step (q, r): NT column groups x 8 bytes per lane.  The wave's part of the table is a uniform
                // base (scalar registers), the lane adds 8 bytes: one address register for the whole flush.
                const __amdgpu_buffer_rsrc_t frag = wave_buffer(stf + (size_t)wave * (WCH / 32) * 4 * NT * 64);
                auto fetch_b = [&](double (&b)[NT], int step) {
#pragma unroll
                    for (int c = 0; c < NT; ++c) b[c] = buffer_load_raw<double, false>(frag, fl * 8, (step * NT + c) * 512);
                };
                // DEPTH k-steps of fragments are in flight ahead of their use: a slot is refilled for step + DEPTH right after
                // its MFMAs.  The table ends with two groups of zeros (host: get_st), so the requests past the wave's last
                // step need no clamp (they are never used).  What a flush waits for is this latency: with one step in flight a
                // k-step took 2 600 cycles against 255 of its 15 MFMAs (tools/walk_timeline.py, profiles/r03).
                // As deep as the registers allow without spilling (the loads go through a buffer descriptor: no address
                // registers): 4 steps where the budget is 168 registers (10 waves), 2 where the kernel carries no second accumulator set.
                constexpr int DEPTH = DCTFP_WALK_B_DEPTH ? DCTFP_WALK_B_DEPTH : (S >= 10 ? (MA ? 2 : 4) : (FUSED ? 1 : 2));
                static_assert(DEPTH == 1 || DEPTH == 2 || DEPTH == 4, "slot of a k-step must be static under the 4-step unroll");
                double bq[DEPTH][NT];
                if (n_q > 0) {
#pragma unroll
                    for (int r = 0; r < DEPTH; ++r) fetch_b(bq[r], r);
                }
                for (int qi = 0; qi < n_q; ++qi) {
                    const int pl0 = 16 * qi + 4 * g4;  // my pairs of this group: pl0 + r
                    const int p0 = wave * (WCH / 2) + pl0;
                    // pairs past D/2 (D % 32 != 0) get state 0 = value 0, whatever the slots hold: byte r of the word of the
                    // channels d, byte 3 - r of the mirrors' word
                    const int nl = min(4, max(0, half - p0));
                    const uint32_t lm = nl >= 4 ? 0xffffffffu : ((1u << (8 * nl)) - 1u);
                    const uint32_t lmm = nl >= 4 ? 0xffffffffu : (nl <= 0 ? 0u : ~((1u << (8 * (4 - nl))) - 1u));
                    const uint32_t cw = lds_c[wave][gs][pl0 >> 2] & lm;
                    const uint32_t cwm = lds_c[wave][gs][WCH / 8 + (pl0 >> 2)] & lmm;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double t = lds_t[wave][gs][pl0 + r], tm = lds_t[wave][gs][WCH / 2 + pl0 + 3 - r];
#pragma unroll
                        for (int jr = 0; jr < 3; ++jr) {
                            const double y = unpack_bits(cw, 8 * r + 2 * jr, t);
                            const double ym = unpack_bits(cwm, 8 * (3 - r) + 2 * jr, tm);
                            const double au = y + ym, av = y - ym;
                            const double ax = fma(fold_sign, ym, y);  // blocks 0, 1 of the middle column group: u, blocks 2, 3: v
                            // 3 * NT independent accumulators between two uses of one
#pragma unroll
                            for (int c = 0; c < NT; ++c) {
                                if constexpr (NT == 5)
                                    acc[jr][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(c < 2 ? au : (c == 2 ? ax : av), bq[r % DEPTH][c], acc[jr][c], 0, 0, 0);
                                else
                                    acc[jr][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(c < NT / 2 ? au : av, bq[r % DEPTH][c], acc[jr][c], 0, 0, 0);
                            }
                        }
                        fetch_b(bq[r % DEPTH], 4 * qi + r + DEPTH);
                    }
                }
                // partial blocks -> my slots (the Y' in them is consumed): tile row = lane >> 4 = job, zp[row jr][slot]
                __builtin_amdgcn_wave_barrier();
                if ((uint32_t)(fl >> 4) < pending) {
#pragma unroll
                    for (int jr = 0; jr < 3; ++jr)
#pragma unroll
                        for (int c = 0; c < NT; ++c) lds_t[wave][fl >> 4][jr * (NT * 16) + c * 16 + (fl & 15)] = acc[jr][c];
                }
            }
            DCTFP_TL_MARK(2);
            // the three fingerprint rows of one job of the flush: sum over the waves in wave order, Z[c] = ZE[c] + ZO[c] and
            // Z[m-1-c] = ZE[c] - ZO[c], per-row min-max scale, int8 (src/fingerprint.py:193-195); lane c < ceil(m / 2) holds both.
            // Where the registers allow, the three rows go through together: their dependent chains (S additions, the DPP
            // reduction, two divisions) interleave, so a job costs one wave little more than a single row did.  The fused
            // float32 variants at the 128-register budget take them one by one (together they spill 4 registers per lane).
            constexpr int ROWS = (FUSED && S < 10 && sizeof(T) == 4) ? 1 : 3;
            auto finish_job = [&](uint32_t g) {
                // (MA: the lane index through an opaque copy, like the flush above -- the store offsets and masks derived from it
                //  would otherwise be computed at the top of the kernel and held in ~20 registers through the row stream)
                int lane = threadIdx.x & 63;
                if constexpr (MA) asm volatile("" : "+v"(lane));
                const int hm = (m + 1) >> 1;
                const bool valid0 = lane < hm;
                const bool valid1 = lane < hm && (m - 1 - lane) != lane;  // odd m: the middle column is its own mirror (O = 0 there)
                int8_t* __restrict__ o = out + jobb[group_job + g].out_off;
                static_assert(DCTFP_MAX_M_K <= 128 && HS <= 48, "lanes 0 .. 47 hold a row (wave_min_max48)");
                for (int j0 = 0; j0 < 3; j0 += ROWS) {
                    double v0[ROWS], v1[ROWS], mn[ROWS], mx[ROWS];
                    bool nan_here[ROWS];
#pragma unroll
                    for (int jj = 0; jj < ROWS; ++jj) {
                        const int j = j0 + jj;
                        double ze = 0.0, zo = 0.0;
                        if (valid0) {
#pragma unroll
                            for (int w = 0; w < S; ++w) {
                                ze += lds_t[w][g][j * (NT * 16) + lane];
                                zo += lds_t[w][g][j * (NT * 16) + HS + lane];
                            }
                        }
                        v0[jj] = ze + zo;
                        v1[jj] = ze - zo;
                        mn[jj] = valid0 ? fmin(v0[jj], valid1 ? v1[jj] : v0[jj]) : INFINITY;
                        mx[jj] = valid0 ? fmax(v0[jj], valid1 ? v1[jj] : v0[jj]) : -INFINITY;
                        nan_here[jj] = valid0 && (v0[jj] != v0[jj] || v1[jj] != v1[jj]);
                    }
#pragma unroll
                    for (int jj = 0; jj < ROWS; ++jj) wave_min_max48(mn[jj], mx[jj]);
#pragma unroll
                    for (int jj = 0; jj < ROWS; ++jj) {
                        const int j = j0 + jj;
                        const bool bad = __builtin_amdgcn_ballot_w64(nan_here[jj]) != 0;  // a NaN anywhere in the row: the whole row is 0
                        const double den = mx[jj] - mn[jj];
                        if (valid0) o[j * m + lane] = quant127(v0[jj] - mn[jj], den, bad);
                        if (valid1) o[j * m + m - 1 - lane] = quant127(v1[jj] - mn[jj], den, bad);
                    }
                }
            };
            // No workgroup barrier.  A wave that has stored its partial blocks draws a ticket and goes on streaming; the LAST
            // min(S, pending) waves to arrive -- the ones nobody would have to wait for -- write the rows, each one job
            // (g, g + nf, ...), once every wave has arrived.  The sum over the partial blocks runs in wave order whoever does
            // it, so the bytes do not depend on the order of arrival.  The slots are reused only after `lds_done` says the rows
            // are out (checked before the first write of the next group, a whole job's stream later), so the waves of a
            // workgroup may drift apart by up to a job before anyone waits for anyone -- and no wave draws a ticket of flush
            // n + 1 before every wave has drawn its ticket of flush n (its first write of group n + 1 waits for rows that the
            // last arrival of flush n has to release).
            // Round 4 measured what that buys: +0.3 % (c4) .. +0.4 % (C2).  The waves DID wait 9-17 % of their time at the two
            // barriers this replaces (profiles/r03/wave_wait_probe_shipped_kernel.txt) -- and the rate does not care: what a
            // wave waits for there is issue time that the other waves of its SIMD are using (profiles/r04/experiments/).
            {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                uint32_t ticket = 0;
                if (lane == 0) ticket = __hip_atomic_fetch_add(&lds_arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket);
                const uint32_t idx = ticket - (uint32_t)S * flush_seq;       // my place among the arrivals of this flush
                const uint32_t nf = pending < (uint32_t)S ? pending : (uint32_t)S;
                if (idx >= (uint32_t)S - nf) {
                    const uint32_t all = (uint32_t)S * (flush_seq + 1u);
                    while (__hip_atomic_load(&lds_arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != all)
                        __builtin_amdgcn_s_sleep(1);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    DCTFP_TL_MARK(3);
                    for (uint32_t g = idx - ((uint32_t)S - nf); g < pending; g += nf) {
                        finish_job(g);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0) __hip_atomic_fetch_add(&lds_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                ++flush_seq;
                done_expected += pending;
                DCTFP_TL_MARK(4);
            }
            group_job += pending;
            pending = 0;
        }
    }
#ifdef DCTFP_WALK_TIMELINE
    DCTFP_TL_MARK(6);
    tl_acc[7] = tl_prev - tl_begin;
    tl_acc[8] = 1;
    if (lane == 0) {
        // per-wave trace (option walk_trace, tools/walk_trace.py): begin, end, where it ran, which workgroup -- instead of the
        // phase sums (100 000 waves adding to the same eleven words stretch the launch they are meant to describe)
        unsigned long long* __restrict__ trace = reinterpret_cast<unsigned long long*>((uintptr_t)degenerate[17]);
        const unsigned long long idx = (unsigned long long)blockIdx.x * S + wave;
        if (!trace) {
#pragma unroll
            for (int i = 0; i < 11; ++i) atomicAdd(degenerate + 1 + i, (unsigned long long)tl_acc[i]);
        } else if (idx < degenerate[16]) {
            const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_ID, XCC_ID
            trace[4 * idx + 0] = tl_begin;
            trace[4 * idx + 1] = tl_prev;
            trace[4 * idx + 2] = ((unsigned long long)xcc << 32) | hw;
            trace[4 * idx + 3] = ((unsigned long long)tl_acc[3] << 32) | ((unsigned long long)(blockIdx.x & 0xffffffu) << 8) | (unsigned)wave;  // barrier wait (ticks), workgroup, wave
        }
    }
#endif
}

// ---------------------------------------------------------------------------
// K3g: the walk kernel for every OTHER shape -- any n = 2 .. 8, any m <= 128, any width the LDS holds, float32 or float64
// rows, 16-byte aligned or not (round 4; VERDICT r3 "one fused kernel for all of quantize").  Same idea as walk_ab_kernel
// (a workgroup owns all channels of its jobs, Y' never leaves the CU, the int8 block is all that reaches HBM), without the
// shape-specific machinery: no packed {0, t, 1} rows, no even / odd fold, no fused walks, one job per flush.
//
//   grid  : one workgroup per run of consecutive jobs (the host's Run table; walks are ignored: every job streams its rows).
//   block : S = ceil(D / (64 VEC)) waves; lane l of wave w owns the channels [(64 w + l) VEC, +VEC) in natural order.
//   LDS   : (dynamic) n_slots x { Y'[N][CH] float64, CH = 64 S VEC } + counters; a wave's partial Z [N][cp], cp = 16 ceil(m / 16),
//           overwrites its own columns of Y' (one channel per lane and cp > 64: a region of its own behind Y').
//           Two slots where they fit: a wave writes the next job's Y' while the last arrival of this job still sums.
//   job   : stage A as everywhere (first-row shift, cosines through the scalar cache, UNROLL rows in flight), then
//           scale_channel<N> per channel -> Y' into the slot (own channels: wave-private), then the wave contracts ITS
//           channels against the plain stage-B basis with v_mfma_f64_4x4x4 -- tile rows = the N resampled rows (1 or 2
//           tiles of 4), 16 columns per MFMA, K = 4 channels per step -- and stores its partial N x cp block.  The last wave
//           to arrive (ticket, as in walk_ab_kernel: nobody waits) sums the partial blocks in wave order, scales every row
//           over its m values and writes the int8 block.
//   stp   : the plain basis in fragment order (host: get_st): stp[(q NT + c) 64 + lane] = St[4 q + (lane >> 4)][16 c + (lane & 15)],
//           zero beyond D and m, for q < CH / 4.
// ---------------------------------------------------------------------------
template <typename R>
__device__ inline R buffer_load_any(__amdgpu_buffer_rsrc_t rs, int lane_bytes, int uniform_bytes) {
    if constexpr (sizeof(R) == 4) {
        const unsigned r = __builtin_amdgcn_raw_buffer_load_b32(rs, lane_bytes, uniform_bytes, DCTFP_STREAM_AUX);
        return __builtin_bit_cast(R, r);
    } else {
        return buffer_load_raw<R, true>(rs, lane_bytes, uniform_bytes);
    }
}

__device__ inline void wave_min_max64(double& mn, double& mx) {  // over all 64 lanes (wave_min_max48 + the fourth lane row)
    mn = fmin(mn, dpp_rotate<0x128>(mn));
    mx = fmax(mx, dpp_rotate<0x128>(mx));
    mn = fmin(mn, dpp_rotate<0x124>(mn));
    mx = fmax(mx, dpp_rotate<0x124>(mx));
    mn = fmin(mn, dpp_rotate<0x122>(mn));
    mx = fmax(mx, dpp_rotate<0x122>(mx));
    mn = fmin(mn, dpp_rotate<0x121>(mn));
    mx = fmax(mx, dpp_rotate<0x121>(mx));
    mn = fmin(fmin(lane_value(mn, 0), lane_value(mn, 16)), fmin(lane_value(mn, 32), lane_value(mn, 48)));
    mx = fmax(fmax(lane_value(mx, 0), lane_value(mx, 16)), fmax(lane_value(mx, 32), lane_value(mx, 48)));
}

// FUSED (round 5): the walks of walk_ab_kernel -- the parts of a protein whose last domain is the whole protein (what RecCut emits,
// src/fingerprint.py:103-107) stream their rows ONCE: a second accumulator set collects the whole protein's coefficients against the
// part's first row (one subtraction per element feeds both sets), its own shift restored once per part from the prefix sums
// behind the whole protein's cosine table (see walk_ab_kernel); the whole protein is then a job that streams nothing.  Builds for
// n <= 5 (two sets of n - 1 accumulators per channel); UNROLL 4 rows in flight there, as the fused variants of the tuned kernel.
// NTC: column groups of 16 the build holds registers for (4: m <= 64, 8: m <= 128) -- the stage-B fragments of DEPTH k-steps are
// in flight ahead of their use, NTC x 8 bytes per lane and step.
// Launch bounds: the fused builds carry two accumulator sets (at n = 5 they spilled 20 registers per lane under the 128-register
// budget of 16 waves); their workgroups have at most ten waves (host: gen_can_fuse), as the tuned kernel's -- 168 registers.
template <typename T, int N, int VEC, bool FUSED = false, int NTC = 8>
__global__ __launch_bounds__(FUSED ? 640 : 1024) void walk_gen_kernel(const JobA* __restrict__ jobs, const JobB* __restrict__ jobb,
                                                         const Walk* __restrict__ walks, const Run* __restrict__ runs,
                                                         const PieceA* __restrict__ pieces,
                                                         const double* __restrict__ stp, int8_t* __restrict__ out, int n_cols,
                                                         int64_t ld, int m, int n_slots, InvTab<N> inv,
                                                         unsigned long long* __restrict__ degenerate) {
    constexpr int NK = N - 1, TILES = (N + 3) / 4, UNROLL = FUSED ? 4 : 8;
    extern __shared__ double lds_dyn[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int S = (int)(blockDim.x >> 6);
    const int CH = S * 64 * VEC;
    const int NT = (m + 15) >> 4, cp = 16 * NT;
    // A wave's partial block lives in its OWN columns of the slot (row j, columns [64 VEC w, +cp): the Y' there is consumed
    // by then) wherever they hold it; with one channel per lane and cp > 64 it gets a region of its own behind Y'.
    const bool alias = cp <= 64 * VEC;
    const int slot_doubles = N * CH + (alias ? 0 : S * N * cp);
    // per slot: [s] waves that have stored their partial block, [2 + s] jobs whose rows are written -- over all jobs of the
    // workgroup that used the slot (per slot, so that a wave one job ahead cannot be mistaken for an arrival of this job)
    uint32_t* const counters = reinterpret_cast<uint32_t*>(lds_dyn + (size_t)n_slots * slot_doubles);
    if (threadIdx.x < 4) counters[threadIdx.x] = 0;
    __syncthreads();

    const Run run = runs[blockIdx.x];
    const int ch0 = (wave * 64 + lane) * VEC;          // my first channel
    const bool pad = ch0 >= n_cols;                      // (D % VEC == 0: a lane has all of its channels or none)
    typedef typename Raw<T, VEC>::type Rw;
    const int col_bytes = (pad ? 0 : ch0) * (int)sizeof(T);
    const int ld_bytes = (int)(ld * (int64_t)sizeof(T));

    uint32_t jn = 0;   // jobs of this run behind us (the jobs of a run are consecutive: job = run.job_begin + jn)
    const uint32_t n_walks = FUSED ? run.n_walks : run.n_jobs;   // (not FUSED: every job is a walk of its own, the table is not read)
    for (uint32_t wi = 0; wi < n_walks; ++wi) {
        uint32_t n_parts = 1;
        int32_t whole_job = -1;
        if constexpr (FUSED) {
            const Walk wk = walks[run.walk_begin + wi];
            n_parts = wk.n_parts;
            whole_job = wk.whole_job;
        }
        const bool has_w = FUSED && whole_job >= 0;
        const uint32_t n_walk_jobs = n_parts + (has_w ? 1u : 0u);
        double wacc[FUSED ? (NK > 0 ? NK : 1) : 1][VEC];
#pragma unroll
        for (int k = 0; k < (FUSED ? NK : 1); ++k)
#pragma unroll
            for (int v = 0; v < VEC; ++v) wacc[k][v] = 0.0;
        const uint32_t w_rows = has_w ? jobs[whole_job].n_rows : 0u;   // rows of the whole protein = offset of the prefix sums behind its cosine table
    for (uint32_t part = 0; part < n_walk_jobs; ++part, ++jn) {
        const uint32_t slot = jn % (uint32_t)n_slots, before = jn / (uint32_t)n_slots;   // jobs that used this slot before
        double* const ys = lds_dyn + (size_t)slot * slot_doubles;                         // Y'[N][CH]
        // partial Z of wave w, row j: pz(w, j)[0 .. cp)
        auto pz = [&](int w, int j) { return alias ? ys + j * CH + w * 64 * VEC : ys + N * CH + (w * N + j) * cp; };
        double f[NK][VEC];
#pragma unroll
        for (int k = 0; k < NK; ++k)
#pragma unroll
            for (int v = 0; v < VEC; ++v) f[k][v] = 0.0;
        if (part >= n_parts) {   // the whole protein of a fused walk: its coefficients have been collected while the parts streamed by
#pragma unroll
            for (int k = 0; k < NK; ++k)
#pragma unroll
                for (int v = 0; v < VEC; ++v) f[k][v] = wacc[FUSED ? k : 0][v];
        } else if (!pad) {
            // ---- stage A: every row of my channels
            const JobA job = jobs[run.job_begin + jn];
            const PieceA* __restrict__ pc = pieces + job.piece_begin;
            double ref[VEC];
            {
                const Rw r0 = load_raw<T, VEC>(reinterpret_cast<const T*>(pc[0].ptr) + ch0);
#pragma unroll
                for (int v = 0; v < VEC; ++v) ref[v] = raw_elem<T, VEC>(r0, v);
            }
            double cwsum[NK > 0 ? NK : 1];
#pragma unroll
            for (int k = 0; k < NK; ++k) cwsum[k] = 0.0;
            // (the test of `has_w` must not sit inside the row loop: see walk_ab_kernel)
            auto stream_piece = [&](auto hw_tag, const PieceA& piece) {
                constexpr bool HW = decltype(hw_tag)::value;
                const __amdgpu_buffer_rsrc_t rows = wave_buffer(piece.ptr);
                const CosTab btp = cos_tab(job.basis) + (size_t)piece.t0 * NK;
                const CosTab wtp = cos_tab(job.w_basis) + (size_t)piece.w0 * NK;
                auto load_row = [&](uint32_t r) { return buffer_load_any<Rw>(rows, col_bytes, (int)r * ld_bytes); };
                auto row_update = [&](const Rw& x, uint32_t r) {
                    const CosTab c = btp + (size_t)r * NK;
                    const CosTab cw = wtp + (size_t)r * NK;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const double d = raw_elem<T, VEC>(x, v) - ref[v];
#pragma unroll
                        for (int k = 0; k < NK; ++k) f[k][v] = fma(c[k], d, f[k][v]);
                        if constexpr (HW) {
#pragma unroll
                            for (int k = 0; k < NK; ++k) wacc[FUSED ? k : 0][v] = fma(cw[k], d, wacc[FUSED ? k : 0][v]);
                        }
                    }
                };
                uint32_t r = 0;
                for (; r + UNROLL <= piece.n_rows; r += UNROLL) {
                    Rw xv[UNROLL];
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u) xv[u] = load_row(r + u);
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u) row_update(xv[u], r + u);
                }
                if constexpr (UNROLL > 4) {
                    if (r + 4 <= piece.n_rows) {
                        Rw xv[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) xv[u] = load_row(r + u);
#pragma unroll
                        for (int u = 0; u < 4; ++u) row_update(xv[u], r + u);
                        r += 4;
                    }
                }
                if (r < piece.n_rows) {
                    Rw xv[3];
#pragma unroll
                    for (int u = 0; u < 3; ++u)
                        if (r + u < piece.n_rows) xv[u] = load_row(r + u);
#pragma unroll
                    for (int u = 0; u < 3; ++u)
                        if (r + u < piece.n_rows) row_update(xv[u], r + u);
                }
                if constexpr (HW) {  // prefix sums past this piece's last and at its first row
                    const CosTab wpre = wtp + (size_t)w_rows * NK;
#pragma unroll
                    for (int k = 0; k < NK; ++k) cwsum[k] += wpre[(size_t)piece.n_rows * NK + k] - wpre[k];
                }
            };
            for (uint32_t p = 0; p < job.n_pieces; ++p) {
                const PieceA piece = pc[p];
                if (FUSED && has_w) stream_piece(std::integral_constant<bool, FUSED>{}, piece);
                else stream_piece(std::false_type{}, piece);
            }
            if (FUSED && has_w) {   // the whole protein's own shift:  sum_t cw(t) (x_t - r_w) = sum_t cw(t) (x_t - r_p) + (r_p - r_w) sum_t cw(t)
                const Rw w0 = load_raw<T, VEC>(reinterpret_cast<const T*>(job.w_ref) + ch0);
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const double dr = ref[v] - raw_elem<T, VEC>(w0, v);
#pragma unroll
                    for (int k = 0; k < NK; ++k) wacc[FUSED ? k : 0][v] = fma(dr, cwsum[k], wacc[FUSED ? k : 0][v]);
                }
            }
        }
        // ---- epilogue: scale my channels; Y' into the slot (its last user, job jn - n_slots, must have its rows out)
        if (before != 0) {
            while (__hip_atomic_load(&counters[2 + slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != before)
                __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            double fk[NK > 0 ? NK : 1], z[N];
#pragma unroll
            for (int k = 0; k < NK; ++k) fk[k] = f[k][v];
            scale_channel<N>(fk, inv, pad, z, degenerate);
#pragma unroll
            for (int j = 0; j < N; ++j) ys[j * CH + ch0 + v] = z[j];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- stage B of this job: my channels against the basis
        {
            double acc[TILES][NTC];
#pragma unroll
            for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
                for (int c = 0; c < NTC; ++c) acc[tl][c] = 0.0;
            const int i4 = lane & 3, k4 = lane >> 4;
            const int wch = wave * 64 * VEC;
            const int steps = min(16 * VEC, max(0, (n_cols - wch + 3) >> 2));   // k-steps of 4 channels that hold channels
            // The B fragments of a k-step are NT x 512 bytes per wave from L2 -- the same for every job, D x cp x 8 bytes per job in
            // all: on ~100-row jobs as many bytes as the job's own rows.  With one step's loads issued right before their MFMAs
            // the contraction was a chain of 16 VEC L2 round trips per job, and the general kernel streamed the c4 / c5 mixes at
            // 2.1-2.5 TB/s whatever it read (profiles/r05/gen_probe_fused_first.txt).  DEPTH steps are now in flight ahead of their
            // use, through a buffer descriptor (wave-uniform base and step offset: no address registers).
            // (32 registers of fragments at most; where the accumulators leave less -- n >= 6, fused n = 5 -- fewer steps, so that nothing spills)
            constexpr int DEPTH = N > 5 ? 1 : ((FUSED && N == 5) ? (NTC > 4 ? 1 : 2) : (NTC > 4 ? 2 : 4));
            const __amdgpu_buffer_rsrc_t frag = wave_buffer(stp + ((size_t)(wch >> 2) * NT) * 64);
            int fl = lane;
            asm volatile("" : "+v"(fl));   // (keeps the lane's byte offset from being hoisted above the row stream)
            auto fetch_b = [&](double (&b)[NTC], int q) {
#pragma unroll
                for (int c = 0; c < NTC; ++c)
                    if (c < NT) b[c] = buffer_load_raw<double, false>(frag, fl * 8, (q * NT + c) * 512);
            };
            double bq[DEPTH][NTC];
#pragma unroll
            for (int r = 0; r < DEPTH; ++r)
                if (r < steps) fetch_b(bq[r], r);
            for (int q0 = 0; q0 < steps; q0 += DEPTH) {
#pragma unroll
                for (int r = 0; r < DEPTH; ++r) {
                    const int q = q0 + r;
                    if (q < steps) {
                        double a[TILES];
#pragma unroll
                        for (int tl = 0; tl < TILES; ++tl) a[tl] = ys[min(4 * tl + i4, N - 1) * CH + wch + 4 * q + k4];
#pragma unroll
                        for (int c = 0; c < NTC; ++c) {
                            if (c < NT) {
#pragma unroll
                                for (int tl = 0; tl < TILES; ++tl) acc[tl][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[tl], bq[r][c], acc[tl][c], 0, 0, 0);
                            }
                        }
                        if (q + DEPTH < steps) fetch_b(bq[r], q + DEPTH);
                    }
                }
            }
            // D[i = lane >> 4][j = lane & 3] of block (lane >> 2) & 3: row 4 tl + (lane >> 4), column 16 c + (lane & 15)
            __builtin_amdgcn_wave_barrier();   // (the partial block overwrites Y' this wave has just read)
#pragma unroll
            for (int tl = 0; tl < TILES; ++tl) {
                const int row = 4 * tl + (lane >> 4);
                if (row < N) {
#pragma unroll
                    for (int c = 0; c < NTC; ++c)
                        if (c < NT) pz(wave, row)[16 * c + (lane & 15)] = acc[tl][c];
                }
            }
        }
        // ---- the last wave to arrive writes the rows
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        uint32_t ticket = 0;
        if (lane == 0) ticket = __hip_atomic_fetch_add(&counters[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket);
        if (ticket == (uint32_t)S * (before + 1u) - 1u) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            int8_t* __restrict__ o = out + jobb[run.job_begin + jn].out_off;
            for (int j = 0; j < N; ++j) {
                double v0 = 0.0, v1 = 0.0;
                const bool ok0 = lane < m, ok1 = lane + 64 < m;
                for (int w = 0; w < S; ++w) {   // wave order: the bytes do not depend on who arrives last
                    if (lane < cp) v0 += pz(w, j)[lane];
                    if (lane + 64 < cp) v1 += pz(w, j)[lane + 64];
                }
                double mn = fmin(ok0 ? v0 : INFINITY, ok1 ? v1 : INFINITY);
                double mx = fmax(ok0 ? v0 : -INFINITY, ok1 ? v1 : -INFINITY);
                const bool nan_here = (ok0 && v0 != v0) || (ok1 && v1 != v1);
                wave_min_max64(mn, mx);
                const bool bad = __builtin_amdgcn_ballot_w64(nan_here) != 0;   // a NaN anywhere in the row: the whole row is 0
                const double den = mx - mn;
                if (ok0) o[j * m + lane] = quant127(v0 - mn, den, bad);
                if (ok1) o[j * m + lane + 64] = quant127(v1 - mn, den, bad);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_fetch_add(&counters[2 + slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    }
}

// ---------------------------------------------------------------------------
// K1s: a SMALL call -- a protein at a time, the reference's own calling pattern (src/make_db.py:29-30) -- in ONE launch (round 5).
// Such a call is the latency of a dependent chain, not bandwidth: stage_a_split_kernel -> stage_b_slab_kernel ->
// stage_b_finish_kernel were three launches, ~ 22 us on the GPU of a 56-us call (profiles/r05/pcie_inclusive_rate.txt).  Here the
// same three steps hand over inside one grid, by tickets: nobody waits for anybody (a workgroup that is not the last of its group
// to arrive is done), so no assumption about which workgroups are resident together.
//
//   grid   : jobs x row chunks x 256-channel slabs (as stage_a_split_kernel), 8 waves each.
//   step 1 : every workgroup: the partial sums of its rows for its 256 channels -> `partial` (global), fence, ticket of
//            (job, slab).
//   step 2 : the LAST workgroup of a (job, slab) to arrive: adds the chunks in chunk order, scales its 256 channels
//            (scale_channel<3>) into LDS, contracts them against the stage-B basis -- a wave per 32 channels, all eight k-steps of
//            fragments in flight at once (one L2 round trip), v_mfma_f64_4x4x4, rows = the 3 resampled rows -- adds the eight
//            waves' blocks in wave order -> `zpart` (global), fence, ticket of the job.
//   step 3 : the LAST slab of a job to arrive: adds the slabs in slab order, scales the three rows over their m values, writes
//            the int8 block (into pinned host memory for a one-protein call).
// Every sum runs in a fixed order, whoever does it: the bytes do not depend on the order of arrival.  The tickets are reset by
// their last taker; the host zeroes them once, when it allocates them.
// OUTCOME (profiles/r05/pcie_rate_one_launch.txt against pcie_rate_three_launches.txt): bit-exact, and SLOWER -- 86 us per
// one-protein call against 57 with the three kernels.  The eight XCDs of the chip do not share an L2: what one workgroup wrote
// reaches a workgroup on another XCD through an agent-scope release (L2 write-back) and acquire (L2 invalidate) per workgroup
// and step, and the tickets are memory-side atomics -- dearer than the two launch boundaries they replace, which do the same
// once for the whole grid.  Kept behind the option "small_one" (default 0) with its parity test; the dispatch does not pick it.
//   stp    : the plain basis in fragment order (host: get_st_plain), NT = cp / 16 column groups (m <= 80: NT <= 5).
// ---------------------------------------------------------------------------
template <int WAVES, int UNROLL>
__global__ __launch_bounds__(WAVES * 64) void small_call_kernel(const JobA* __restrict__ jobs, const JobB* __restrict__ jobb,
                                                                 const PieceA* __restrict__ pieces, double* __restrict__ partial,
                                                                 double* __restrict__ zpart, uint32_t* __restrict__ tickets, int n_jobs,
                                                                 int n_chunks, uint32_t chunk_rows, int n_cols, int64_t ld, int ldy,
                                                                 int n_slabs, const double* __restrict__ stp, int m, InvTab<3> inv,
                                                                 unsigned long long* __restrict__ degenerate, int8_t* __restrict__ out) {
    typedef float T;
    constexpr int N = 3, NK = 2, VEC = 4, NTC = 5, SLAB = 64 * VEC;
    static_assert(WAVES == 8, "a wave per 32 channels of the slab in step 2");
    __shared__ double red[WAVES][NK * VEC][64];      // step 1: per-wave sums; step 2: Y'[3][256] + the waves' partial blocks [8][3][80]
    __shared__ uint32_t last_flag;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t slab = blockIdx.x % (uint32_t)n_slabs;
    const uint32_t jc = blockIdx.x / (uint32_t)n_slabs;
    const uint32_t chunk = jc % (uint32_t)n_chunks, job_id = jc / (uint32_t)n_chunks;
    const int col0 = ((int)slab * 64 + lane) * VEC;
    const int colc = (col0 < n_cols) ? col0 : 0;
    const JobA job = jobs[job_id];
    // ---- step 1 (stage_a_split_kernel's body)
    {
        const PieceA* __restrict__ pc = pieces + job.piece_begin;
        const uint32_t lo = chunk * chunk_rows;
        const uint32_t hi = min(job.n_rows, lo + chunk_rows);  // this workgroup's rows of the job: [lo, hi)
        double acc[NK][VEC];
        double ref[VEC];
#pragma unroll
        for (int k = 0; k < NK; ++k)
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[k][v] = 0.0;
        if (lo < hi) {
            {
                auto r0 = load_raw<T, VEC>(reinterpret_cast<const T*>(pc[0].ptr) + colc);
#pragma unroll
                for (int v = 0; v < VEC; ++v) ref[v] = raw_elem<T, VEC>(r0, v);
            }
            const CosTab bt = cos_tab(job.basis);
            for (uint32_t p = 0; p < job.n_pieces; ++p) {
                const PieceA piece = pc[p];
                const uint32_t a = max(lo, piece.t0), b = min(hi, piece.t0 + piece.n_rows);  // job rows of this piece in the chunk
                if (a >= b) continue;
                const T* __restrict__ base = reinterpret_cast<const T*>(piece.ptr) + colc;
                auto row_update = [&](const typename Raw<T, VEC>::type& x, uint32_t t) {  // t = row of the job
                    const CosTab c = bt + (size_t)t * NK;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const double d = raw_elem<T, VEC>(x, v) - ref[v];
#pragma unroll
                        for (int k = 0; k < NK; ++k) acc[k][v] = fma(c[k], d, acc[k][v]);
                    }
                };
                uint32_t t = a + (uint32_t)wave;
                for (; t + (UNROLL - 1) * WAVES < b; t += UNROLL * WAVES) {
                    typename Raw<T, VEC>::type xv[UNROLL];
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u) xv[u] = load_raw<T, VEC>(base + (size_t)(t + u * WAVES - piece.t0) * ld);
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u) row_update(xv[u], t + u * WAVES);
                }
                for (; t < b; t += WAVES) {
                    auto x1 = load_raw<T, VEC>(base + (size_t)(t - piece.t0) * ld);
                    row_update(x1, t);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NK; ++k)
#pragma unroll
            for (int v = 0; v < VEC; ++v) red[wave][k * VEC + v][lane] = acc[k][v];
        __syncthreads();
        for (int cl = threadIdx.x; cl < SLAB; cl += WAVES * 64) {
            const int ln = cl / VEC, v = cl % VEC;
            const int col = (int)slab * SLAB + cl;
            if (col >= ldy) continue;
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                double sum = red[0][k * VEC + v][ln];
#pragma unroll
                for (int w = 1; w < WAVES; ++w) sum += red[w][k * VEC + v][ln];
                store_through(&partial[((size_t)jc * NK + k) * ldy + col], sum);
            }
        }
    }
    // ---- ticket of (job, slab): the last of its n_chunks workgroups goes on
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t* __restrict__ tk = tickets + (size_t)job_id * n_slabs + slab;
        const uint32_t t = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = t == (uint32_t)n_chunks - 1u;
        if (last) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (for the next call)
        last_flag = last ? 1u : 0u;
    }
    __syncthreads();
    if (last_flag == 0) return;
    __threadfence();
    // ---- step 2: my 256 channels of the job, scaled, against the basis
    double* const ys = &red[0][0][0];                 // [3][SLAB]
    double* const pzb = ys + N * SLAB;                // [WAVES][3][16 NTC]
    static_assert((size_t)N * SLAB + (size_t)WAVES * N * 16 * NTC <= (size_t)WAVES * NK * VEC * 64, "step 2 fits the reduction buffer");
    const int NT = (m + 15) >> 4, cp = 16 * NT;
    const int d0 = (int)slab * SLAB;
    if ((int)threadIdx.x < SLAB) {
        const int d = d0 + (int)threadIdx.x;
        double f[2] = {0.0, 0.0};
        if (d < ldy) {
            const double* __restrict__ pj = partial + ((size_t)job_id * n_chunks) * NK * ldy + d;
            // chunk order; sixteen chunks' loads in flight at a time (a loop of unknown length issued them one by one, each a round
            // trip to where the other workgroups' write-through stores went: 16 x 2 dependent latencies in the chain of a call)
            for (int c0 = 0; c0 < n_chunks; c0 += 16) {
                double pv[16][NK];
#pragma unroll
                for (int i = 0; i < 16; ++i)
#pragma unroll
                    for (int k = 0; k < NK; ++k)
                        pv[i][k] = (c0 + i < n_chunks) ? __builtin_nontemporal_load(pj + ((size_t)(c0 + i) * NK + k) * ldy) : 0.0;
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (c0 + i < n_chunks) {
#pragma unroll
                        for (int k = 0; k < NK; ++k) f[k] += pv[i][k];
                    }
            }
        }
        double z[3];
        scale_channel<3>(f, inv, d >= n_cols, z, degenerate);
#pragma unroll
        for (int j = 0; j < N; ++j) ys[j * SLAB + threadIdx.x] = z[j];
    }
    __syncthreads();
    {
        const int i4 = lane & 3, k4 = lane >> 4;
        const int wch = d0 + 32 * wave;                                       // my 32 channels: k-steps wch / 4 + 0 .. 7
        const int steps = min(8, max(0, (n_cols - wch + 3) >> 2));
        const __amdgpu_buffer_rsrc_t frag = wave_buffer(stp + ((size_t)(wch >> 2) * NT) * 64);
        double b[8][NTC];
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int c = 0; c < NTC; ++c)
                b[q][c] = (q < steps && c < NT) ? buffer_load_raw<double, false>(frag, lane * 8, (q * NT + c) * 512) : 0.0;
        double acc[NTC];
#pragma unroll
        for (int c = 0; c < NTC; ++c) acc[c] = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const double a = ys[min(i4, N - 1) * SLAB + 32 * wave + 4 * q + k4];   // (a channel past D holds 0, a step past `steps` meets b = 0)
#pragma unroll
            for (int c = 0; c < NTC; ++c) acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b[q][c], acc[c], 0, 0, 0);
        }
        // D[i = lane >> 4][j = lane & 3] of block (lane >> 2) & 3: row lane >> 4, column 16 c + (lane & 15)
        const int row = lane >> 4;
        if (row < N) {
#pragma unroll
            for (int c = 0; c < NTC; ++c)
                if (c < NT) pzb[((size_t)wave * N + row) * (16 * NTC) + 16 * c + (lane & 15)] = acc[c];
        }
    }
    __syncthreads();
    const int n_out = N * m;
    double* __restrict__ zp = zpart + ((size_t)job_id * n_slabs + slab) * (N * 16 * NTC);
    for (int o = threadIdx.x; o < N * cp; o += WAVES * 64) {
        const int j = o / cp, c = o % cp;
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) sum += pzb[((size_t)w * N + j) * (16 * NTC) + c];   // wave order
        store_through(&zp[j * (16 * NTC) + c], sum);
    }
    // ---- ticket of the job: the last of its n_slabs slabs goes on
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t* __restrict__ tk = tickets + (size_t)n_jobs * n_slabs + job_id;
        const uint32_t t = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = t == (uint32_t)n_slabs - 1u;
        if (last) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = last ? 1u : 0u;
    }
    __syncthreads();
    if (last_flag == 0) return;
    __threadfence();
    // ---- step 3: the slabs in slab order, per-row min-max scale, int8 (src/fingerprint.py:193-195)
    double* const bl = ys;                            // [3][m]  (everybody is past the partial blocks: the barrier above)
    for (int o = threadIdx.x; o < n_out; o += WAVES * 64) {
        const int j = o / m, c = o % m;
        const double* __restrict__ zj = zpart + (size_t)job_id * n_slabs * (N * 16 * NTC) + j * (16 * NTC) + c;
        double sum = 0.0;
        for (int s0 = 0; s0 < n_slabs; s0 += 8) {   // slab order, eight loads in flight
            double zv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) zv[i] = (s0 + i < n_slabs) ? __builtin_nontemporal_load(zj + (size_t)(s0 + i) * (N * 16 * NTC)) : 0.0;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (s0 + i < n_slabs) sum += zv[i];
        }
        bl[o] = sum;
    }
    __syncthreads();
    if (wave < N) {
        const int j = wave;
        const bool ok0 = lane < m, ok1 = lane + 64 < m;
        const double v0 = ok0 ? bl[j * m + lane] : 0.0, v1 = ok1 ? bl[j * m + lane + 64] : 0.0;
        double mn = fmin(ok0 ? v0 : INFINITY, ok1 ? v1 : INFINITY);
        double mx = fmax(ok0 ? v0 : -INFINITY, ok1 ? v1 : -INFINITY);
        const bool nan_here = (ok0 && v0 != v0) || (ok1 && v1 != v1);
        wave_min_max64(mn, mx);
        const bool bad = __builtin_amdgcn_ballot_w64(nan_here) != 0;   // a NaN anywhere in the row: the whole row is 0
        const double den = mx - mn;
        int8_t* __restrict__ o = out + jobb[job_id].out_off + (int64_t)j * m;
        if (ok0) o[lane] = quant127(v0 - mn, den, bad);
        if (ok1) o[lane + 64] = quant127(v1 - mn, den, bad);
    }
}

// ---------------------------------------------------------------------------
// Zero fill of (layer, domain) blocks whose n or m is 1: the single resampled value
// scales to 0/0 = NaN -> 0 (golden case qdim_n1).
// ---------------------------------------------------------------------------
#ifndef DCTFP_TEMPLATES_ONLY   // (a plain kernel: defined in dctfp.hip only, the kernel-family units see the templates)
__global__ void fill_zero_kernel(const JobB* __restrict__ jobs, int64_t n_jobs, int block_bytes,
                                 int8_t* __restrict__ out) {
    const int64_t total = n_jobs * block_bytes;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        out[jobs[i / block_bytes].out_off + i % block_bytes] = 0;
}
#endif

// ---------------------------------------------------------------------------
// Generic (any num) idct_quant pieces -- NOT a hot path; backs dctfp_idct_quant.
//   G1: fs[k][c] = sum_t cos(pi k (2t+1)/(2N)) (x[t][c] - x[0][c])  (k >= 1);  fs[0][c] = sum_t x[t][c]
//       coef[c][k] = s_k * (k ? fs[k][c] : fs[0][c])
//   G2: y_j = sum_{k>=1} cos(pi k (2j+1)/(2 num)) fs[k][c], min-max scaled over j.
// ---------------------------------------------------------------------------
template <typename T>
__global__ void generic_forward_kernel(const T* __restrict__ x, int64_t n_rows, int64_t n_cols, int64_t ld, int num,
                                       double* __restrict__ fs, double* __restrict__ coef) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y;
    if (c >= n_cols) return;
    const double x0 = (double)x[c];
    double s = 0.0;
    if (k == 0) {
        for (int64_t t = 0; t < n_rows; ++t) s += (double)x[t * ld + c];
    } else {
        for (int64_t t = 0; t < n_rows; ++t)
            s = fma(cospi_ratio((uint64_t)k * (2 * (uint64_t)t + 1), 2 * (uint64_t)n_rows), (double)x[t * ld + c] - x0, s);
    }
    fs[(size_t)k * n_cols + c] = s;
    if (coef) coef[(size_t)c * num + k] = s * (k == 0 ? sqrt(1.0 / (double)n_rows) : sqrt(2.0 / (double)n_rows));
}

#ifndef DCTFP_TEMPLATES_ONLY   // (a plain kernel: defined in dctfp.hip only, the kernel-family units see the templates)
__global__ void generic_inverse_kernel(const double* __restrict__ fs, int64_t n_cols, int num,
                                       double* __restrict__ scaled) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cols) return;
    double mn = INFINITY, mx = -INFINITY;
    bool bad = false;
    for (int j = 0; j < num; ++j) {
        double s = 0.0;
        for (int k = 1; k < num; ++k)
            s = fma(cospi_ratio((uint64_t)k * (2 * (uint64_t)j + 1), 2 * (uint64_t)num), fs[(size_t)k * n_cols + c], s);
        scaled[(size_t)j * n_cols + c] = s;
        bad |= (s != s);
        mn = fmin(mn, s);
        mx = fmax(mx, s);
    }
    const double den = mx - mn;
    for (int j = 0; j < num; ++j) {
        const double s = scaled[(size_t)j * n_cols + c];
        scaled[(size_t)j * n_cols + c] = bad ? __builtin_nan("") : (s - mn) / den;
    }
}
#endif

// scale(): one workgroup, any length.
#ifndef DCTFP_TEMPLATES_ONLY   // (a plain kernel: defined in dctfp.hip only, the kernel-family units see the templates)
__global__ __launch_bounds__(256) void scale_kernel(const double* __restrict__ v, int64_t n, double* __restrict__ out) {
    __shared__ double smn[4], smx[4];
    __shared__ int sbad[4];
    double mn = INFINITY, mx = -INFINITY;
    int bad = 0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const double x = v[i];
        bad |= (x != x) ? 1 : 0;
        mn = fmin(mn, x);
        mx = fmax(mx, x);
    }
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        mn = fmin(mn, __shfl_xor(mn, s));
        mx = fmax(mx, __shfl_xor(mx, s));
        bad |= __shfl_xor(bad, s);
    }
    if ((threadIdx.x & 63) == 0) {
        smn[threadIdx.x >> 6] = mn;
        smx[threadIdx.x >> 6] = mx;
        sbad[threadIdx.x >> 6] = bad;
    }
    __syncthreads();
    mn = fmin(fmin(smn[0], smn[1]), fmin(smn[2], smn[3]));
    mx = fmax(fmax(smx[0], smx[1]), fmax(smx[2], smx[3]));
    bad = sbad[0] | sbad[1] | sbad[2] | sbad[3];
    const double den = mx - mn;
    for (int64_t i = threadIdx.x; i < n; i += 256) out[i] = bad ? __builtin_nan("") : (v[i] - mn) / den;
}
#endif

// get_doms row gather + float64 promotion; one PieceA per piece, t0 = destination row.
template <typename T>
__global__ void gather_rows_kernel(const PieceA* __restrict__ pieces, int n_pieces, int64_t n_cols, int64_t ld,
                                   double* __restrict__ out) {
    const PieceA pc = pieces[blockIdx.y];
    const T* __restrict__ src = reinterpret_cast<const T*>(pc.ptr);
    const int64_t total = (int64_t)pc.n_rows * n_cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / n_cols, c = i % n_cols;
        out[((int64_t)pc.t0 + r) * n_cols + c] = (double)src[r * ld + c];
    }
}


// ---------------------------------------------------------------------------
// Contact top-k (Fingerprint.writece's selection, src/fingerprint.py:54-67): among the pairs
// (i, j), j >= i + 5, of an L x L float32 contact map keep the k largest values; ties are
// broken by (i, j) ascending (Python's stable sort with reverse=True).  One workgroup per
// protein: 4-pass radix select on an order-preserving key, then a collection pass; only
// when more elements tie at the threshold than are needed, wave 0 walks the candidates in
// (i, j) order to take the first ones.  Output order is unspecified (the host sorts).
// ---------------------------------------------------------------------------
struct TopkJob {
    const float* map;
    int64_t ld;
    int32_t n_res;
    int32_t k;
    int64_t out_off;
    int32_t orig;  // index of the protein in the caller's arrays (out_n)
    int32_t reserved;
};

__device__ inline uint32_t topk_key(float v) {
    if (v == 0.0f) v = 0.0f;  // -0.0 and +0.0 compare equal in the reference's sort
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

#ifndef DCTFP_TEMPLATES_ONLY   // (a plain kernel: defined in dctfp.hip only, the kernel-family units see the templates)
__global__ __launch_bounds__(1024) void contact_topk_kernel(const TopkJob* __restrict__ jobs,
                                                             int32_t* __restrict__ out_i, int32_t* __restrict__ out_j,
                                                             float* __restrict__ out_v, int32_t* __restrict__ out_n, int redo_only) {
    __shared__ int hist[256];
    __shared__ uint32_t s_prefix;
    __shared__ int s_need, s_eq, s_cnt;
    const TopkJob job = jobs[blockIdx.x];
    if (redo_only && out_n[job.orig] != -1) return;  // (launched behind contact_topk2_kernel: only what that one handed back)
    const int L = job.n_res;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int last_row = L - 6;  // rows 0 .. L-6 have at least one j >= i + 5
    if (threadIdx.x == 0) {
        s_prefix = 0;
        s_need = job.k;
        s_cnt = 0;
        s_eq = 0;
    }
    __syncthreads();
    if (job.k <= 0 || last_row < 0) {
        if (threadIdx.x == 0) out_n[job.orig] = 0;
        return;
    }
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int b = threadIdx.x; b < 256; b += blockDim.x) hist[b] = 0;
        __syncthreads();
        const uint32_t prefix = s_prefix;
        for (int i = wave; i <= last_row; i += nwaves) {
            const float* __restrict__ row = job.map + (size_t)i * job.ld;
            for (int j = i + 5 + lane; j < L; j += 64) {
                const uint32_t key = topk_key(row[j]);
                if (shift == 24 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(key >> shift) & 255u], 1);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int need = s_need, b = 255;
            for (; b > 0; --b) {
                if (hist[b] >= need) break;
                need -= hist[b];
            }
            s_prefix = prefix | ((uint32_t)b << shift);
            s_need = need;  // elements still to take from bin b
            s_eq = hist[b];
        }
        __syncthreads();
    }
    const uint32_t thr = s_prefix;
    const int need_eq = s_need;
    const bool all_ties = (s_eq == need_eq);
    int32_t* __restrict__ oi = out_i + job.out_off;
    int32_t* __restrict__ oj = out_j + job.out_off;
    float* __restrict__ ov = out_v + job.out_off;
    for (int i = wave; i <= last_row; i += nwaves) {
        const float* __restrict__ row = job.map + (size_t)i * job.ld;
        for (int j = i + 5 + lane; j < L; j += 64) {
            const float v = row[j];
            const uint32_t key = topk_key(v);
            if (key > thr || (all_ties && key == thr)) {
                const int pos = atomicAdd(&s_cnt, 1);
                oi[pos] = i;
                oj[pos] = j;
                ov[pos] = v;
            }
        }
    }
    __syncthreads();
    if (!all_ties && wave == 0) {  // ordered walk for the first need_eq ties
        int base = s_cnt, taken = 0;
        for (int i = 0; i <= last_row && taken < need_eq; ++i) {
            const float* __restrict__ row = job.map + (size_t)i * job.ld;
            for (int j0 = i + 5; j0 < L && taken < need_eq; j0 += 64) {
                const int j = j0 + lane;
                const float v = (j < L) ? row[j] : 0.0f;
                const bool hit = (j < L) && topk_key(v) == thr;
                const unsigned long long m = __ballot(hit);
                const int before = __popcll(m & ((1ull << lane) - 1ull));
                if (hit && taken + before < need_eq) {
                    const int pos = base + taken + before;
                    oi[pos] = i;
                    oj[pos] = j;
                    ov[pos] = v;
                }
                taken += __popcll(m);
            }
        }
    }
    if (threadIdx.x == 0) out_n[job.orig] = job.k;
}
#endif


// ---------------------------------------------------------------------------
// Order of the selected contacts (the CON line of the .ce file, src/fingerprint.py:58-61, :69-73): value descending,
// ties in (i, j) ascending order -- Python's stable sort with reverse=True over pairs appended i-major.  One workgroup
// per protein sorts its k entries in LDS (bitonic network over 64-bit keys: the inverted order-preserving image of the
// value above (i << 16 | j)) and writes them back in place; the value is re-read from the map, so its bits are the
// original ones (-0.0 prints as "-0.000000" in the reference too).  N = capacity of the network (a power of two >= k).
// ---------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(1024) void contact_sort_kernel(const TopkJob* __restrict__ jobs, int32_t* __restrict__ out_i,
                                                             int32_t* __restrict__ out_j, float* __restrict__ out_v) {
    __shared__ unsigned long long keys[N];
    const TopkJob job = jobs[blockIdx.x];
    const int k = job.k;
    if (k <= 1 || k > N) return;  // (the host sends a protein to a network that holds it)
    int32_t* __restrict__ oi = out_i + job.out_off;
    int32_t* __restrict__ oj = out_j + job.out_off;
    float* __restrict__ ov = out_v + job.out_off;
    for (int p = threadIdx.x; p < N; p += blockDim.x) {
        unsigned long long key = ~0ull;
        if (p < k) key = ((unsigned long long)(~topk_key(ov[p])) << 32) | ((uint32_t)oi[p] << 16) | (uint32_t)oj[p];
        keys[p] = key;
    }
    __syncthreads();
    for (int size = 2; size <= N; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int q = threadIdx.x; q < N / 2; q += blockDim.x) {
                const int lo = ((q & ~(stride - 1)) << 1) | (q & (stride - 1));
                const int hi = lo | stride;
                const bool up = (lo & size) == 0;
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a > b) == up) {
                    keys[lo] = b;
                    keys[hi] = a;
                }
            }
            __syncthreads();
        }
    }
    for (int p = threadIdx.x; p < k; p += blockDim.x) {
        const uint32_t ij = (uint32_t)keys[p];
        const int i = (int)(ij >> 16), j = (int)(ij & 0xffffu);
        oi[p] = i;
        oj[p] = j;
        ov[p] = job.map[(size_t)i * job.ld + j];
    }
}

// ---------------------------------------------------------------------------
// Contact top-k of LONG proteins (L >= ~1500): the same selection, spread over many workgroups.  One workgroup per
// (protein, stripe of rows) -- stripes hold about the same number of candidate pairs -- and one launch per step instead of
// one workgroup walking the whole L x L map five times (13 ms for a 5 000-residue protein):
//   4 x [ topk_hist_kernel : stripe histogram of the current radix digit  -> global histogram of the protein
//         topk_pick_kernel : the bin that holds the k-th largest key       -> prefix / need of the next digit ]
//   topk_collect_kernel    : everything above the threshold (and the threshold ties, if all of them are wanted) + the
//                            number of ties per stripe
//   topk_ties_kernel       : only if fewer ties are wanted than exist: every stripe walks its own rows in (i, j) order and
//                            takes its ties while their global rank (ties of the earlier stripes first) is below the need --
//                            Python's stable sort, as in contact_topk_kernel.
// ---------------------------------------------------------------------------
struct TopkStripe {
    int32_t job;        // protein (index into the TopkJob array of the long proteins)
    int32_t row_begin;  // rows [row_begin, row_end) of the map
    int32_t row_end;
    int32_t stripe;     // index of this stripe inside its protein
};

struct TopkState {  // per long protein, in global memory
    uint32_t prefix;   // digits fixed so far
    int32_t need;      // elements still wanted from the bin under the prefix
    int32_t eq;        // elements in that bin
    int32_t count;     // output cursor
    int32_t hist[256];
};

#ifndef DCTFP_TEMPLATES_ONLY   // (a plain kernel: defined in dctfp.hip only, the kernel-family units see the templates)
__global__ __launch_bounds__(1024) void topk_hist_kernel(const TopkJob* __restrict__ jobs, const TopkStripe* __restrict__ stripes,
                                                          TopkState* __restrict__ state, int shift) {
    __shared__ int hist[256];
    const TopkStripe sp = stripes[blockIdx.x];
    const TopkJob job = jobs[sp.job];
    const int L = job.n_res;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    for (int b = threadIdx.x; b < 256; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    const uint32_t prefix = state[sp.job].prefix;
    for (int i = sp.row_begin + wave; i < sp.row_end; i += nwaves) {
        const float* __restrict__ row = job.map + (size_t)i * job.ld;
        for (int j = i + 5 + lane; j < L; j += 64) {
            const uint32_t key = topk_key(row[j]);
            if (shift == 24 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(key >> shift) & 255u], 1);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < 256; b += blockDim.x)
        if (hist[b]) atomicAdd(&state[sp.job].hist[b], hist[b]);
}
#endif

#ifndef DCTFP_TEMPLATES_ONLY   // (a plain kernel: defined in dctfp.hip only, the kernel-family units see the templates)
__global__ __launch_bounds__(64) void topk_pick_kernel(const TopkJob* __restrict__ jobs, TopkState* __restrict__ state, int shift,
                                                        int32_t* __restrict__ out_n) {
    TopkState& st = state[blockIdx.x];
    if (threadIdx.x == 0) {
        if (shift == 0) out_n[jobs[blockIdx.x].orig] = jobs[blockIdx.x].k;
        int need = st.need, b = 255;
        for (; b > 0; --b) {
            if (st.hist[b] >= need) break;
            need -= st.hist[b];
        }
        st.prefix |= (uint32_t)b << shift;
        st.need = need;
        st.eq = st.hist[b];
        for (int i = 0; i < 256; ++i) st.hist[i] = 0;
    }
}
#endif

#ifndef DCTFP_TEMPLATES_ONLY   // (a plain kernel: defined in dctfp.hip only, the kernel-family units see the templates)
__global__ __launch_bounds__(1024) void topk_collect_kernel(const TopkJob* __restrict__ jobs, const TopkStripe* __restrict__ stripes,
                                                             TopkState* __restrict__ state, int32_t* __restrict__ stripe_ties,
                                                             int32_t* __restrict__ out_i, int32_t* __restrict__ out_j,
                                                             float* __restrict__ out_v) {
    __shared__ int s_ties;
    const TopkStripe sp = stripes[blockIdx.x];
    const TopkJob job = jobs[sp.job];
    const int L = job.n_res;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    TopkState& st = state[sp.job];
    const uint32_t thr = st.prefix;
    const bool all_ties = st.eq == st.need;
    if (threadIdx.x == 0) s_ties = 0;
    __syncthreads();
    int32_t* __restrict__ oi = out_i + job.out_off;
    int32_t* __restrict__ oj = out_j + job.out_off;
    float* __restrict__ ov = out_v + job.out_off;
    int my_ties = 0;
    for (int i = sp.row_begin + wave; i < sp.row_end; i += nwaves) {
        const float* __restrict__ row = job.map + (size_t)i * job.ld;
        for (int j = i + 5 + lane; j < L; j += 64) {
            const float v = row[j];
            const uint32_t key = topk_key(v);
            if (key > thr || (all_ties && key == thr)) {
                const int pos = atomicAdd(&st.count, 1);
                oi[pos] = i;
                oj[pos] = j;
                ov[pos] = v;
            } else if (key == thr) {
                ++my_ties;
            }
        }
    }
    if (my_ties) atomicAdd(&s_ties, my_ties);
    __syncthreads();
    if (threadIdx.x == 0) stripe_ties[blockIdx.x] = s_ties;
}
#endif

// grid = stripes; `first_stripe[job]` = index of the protein's first stripe.  Runs after topk_collect_kernel has finished.
#ifndef DCTFP_TEMPLATES_ONLY   // (a plain kernel: defined in dctfp.hip only, the kernel-family units see the templates)
__global__ __launch_bounds__(64) void topk_ties_kernel(const TopkJob* __restrict__ jobs, const TopkStripe* __restrict__ stripes,
                                                        const TopkState* __restrict__ state, const int32_t* __restrict__ stripe_ties,
                                                        const int32_t* __restrict__ first_stripe, int32_t* __restrict__ out_i,
                                                        int32_t* __restrict__ out_j, float* __restrict__ out_v) {
    const TopkStripe sp = stripes[blockIdx.x];
    const TopkJob job = jobs[sp.job];
    const TopkState& st = state[sp.job];
    if (st.eq == st.need) return;  // every tie was wanted and has been collected
    const int need_eq = st.need;
    const int L = job.n_res;
    const int lane = threadIdx.x;
    int before = 0;  // ties in the earlier stripes of this protein
    for (int s = first_stripe[sp.job]; s < (int)blockIdx.x; ++s) before += stripe_ties[s];
    if (before >= need_eq) return;
    const uint32_t thr = st.prefix;
    const int base = job.k - need_eq;  // the elements above the threshold fill the front of the output
    int32_t* __restrict__ oi = out_i + job.out_off;
    int32_t* __restrict__ oj = out_j + job.out_off;
    float* __restrict__ ov = out_v + job.out_off;
    int taken = before;
    for (int i = sp.row_begin; i < sp.row_end && taken < need_eq; ++i) {
        const float* __restrict__ row = job.map + (size_t)i * job.ld;
        for (int j0 = i + 5; j0 < L && taken < need_eq; j0 += 64) {
            const int j = j0 + lane;
            const float v = (j < L) ? row[j] : 0.0f;
            const bool hit = (j < L) && topk_key(v) == thr;
            const unsigned long long mk = __ballot(hit);
            const int rank = taken + __popcll(mk & ((1ull << lane) - 1ull));
            if (hit && rank < need_eq) {
                oi[base + rank] = i;
                oj[base + rank] = j;
                ov[base + rank] = v;
            }
            taken += __popcll(mk);
        }
    }
}
#endif


// ---------------------------------------------------------------------------
// Chunk stitcher (Embedding.embed_seq / combine_contacts, src/embedding.py:123-150, :185-188).
// A long sequence is embedded in windows of maxlen residues starting every maxlen - 200;
// window i's first 200 rows are averaged with the running matrix' last 200 rows, the rest is
// appended.  One launch per window index ("level") over all sequences of a batch, so the
// reference's sequential semantics hold for any maxlen.  float32, (a + b) / 2 as there.
// ---------------------------------------------------------------------------
struct StitchJob {
    const float* src;  // window matrix
    float* dst;        // where the window's first row / top-left corner lands in the output
    int32_t n_rows;    // window rows (embedding) or side (contacts)
    int32_t n_avg;     // leading rows (embedding) / leading square side (contacts) that are averaged
    int64_t ld_src;
    int64_t ld_dst;
    // One launch for all windows (round 4, embeddings whose windows overlap their neighbours only): the rows a window shares
    // with its predecessor are averaged from the two WINDOWS -- `prev` = the predecessor's row that meets my row 0 -- not from
    // the running result, and the rows my successor will average (`n_skip` at my end) are left to it: every output row is
    // written once, every window row read once, no launch waits for another.  prev == nullptr: the sequential form above.
    const float* prev;
    int64_t ld_prev;
    int32_t n_skip;
    int32_t reserved;
};

// grid.x = row blocks of 16, grid.y = job;  block = 256 threads over the columns.
// VEC4: 16 B per lane when every row of src and dst is 16-byte aligned (checked on the host).
template <bool VEC4>
__global__ __launch_bounds__(256) void stitch_rows_kernel(const StitchJob* __restrict__ jobs, int n_cols) {
    const StitchJob job = jobs[blockIdx.y];
    const int r0 = blockIdx.x * 16;
    if (r0 >= job.n_rows) return;
    const int r1 = min(job.n_rows - job.n_skip, r0 + 16);
    for (int r = r0; r < r1; ++r) {
        const float* __restrict__ s = job.src + (size_t)r * job.ld_src;
        float* __restrict__ d = job.dst + (size_t)r * job.ld_dst;
        const bool avg = r < job.n_avg;
        // what my row is averaged with: the running result (sequential form), or the predecessor window's own row
        const float* __restrict__ o = job.prev ? job.prev + (size_t)r * job.ld_prev : d;
        if (VEC4) {
            const v4f* __restrict__ s4 = reinterpret_cast<const v4f*>(s);
            const v4f* __restrict__ o4 = reinterpret_cast<const v4f*>(o);
            v4f* __restrict__ d4 = reinterpret_cast<v4f*>(d);
            for (int c = threadIdx.x; c < n_cols / 4; c += 256) {
                const v4f a = s4[c];
                if (avg) {
                    const v4f b = o4[c];
                    d4[c] = (v4f){(b[0] + a[0]) / 2.0f, (b[1] + a[1]) / 2.0f, (b[2] + a[2]) / 2.0f, (b[3] + a[3]) / 2.0f};
                } else {
                    d4[c] = a;
                }
            }
        } else if (avg) {
            for (int c = threadIdx.x; c < n_cols; c += 256) d[c] = (o[c] + s[c]) / 2.0f;
        } else {
            for (int c = threadIdx.x; c < n_cols; c += 256) d[c] = s[c];
        }
    }
}

// contacts: window square n_rows x n_rows; the leading n_avg x n_avg corner overlaps the
// running map and is averaged, everything else of the square is new (the running map is 0
// there: new_mat = zeros, src/embedding.py:143).
#ifndef DCTFP_TEMPLATES_ONLY   // (a plain kernel: defined in dctfp.hip only, the kernel-family units see the templates)
__global__ __launch_bounds__(256) void stitch_contacts_kernel(const StitchJob* __restrict__ jobs) {
    const StitchJob job = jobs[blockIdx.y];
    const int r0 = blockIdx.x * 16;
    if (r0 >= job.n_rows) return;
    const int r1 = min(job.n_rows, r0 + 16);
    for (int r = r0; r < r1; ++r) {
        const float* __restrict__ s = job.src + (size_t)r * job.ld_src;
        float* __restrict__ d = job.dst + (size_t)r * job.ld_dst;
        for (int c = threadIdx.x; c < job.n_rows; c += 256) {
            if (r < job.n_avg && c < job.n_avg) d[c] = (d[c] + s[c]) / 2.0f;
            else d[c] = d[c] + s[c];
        }
    }
}
#endif


// ---------------------------------------------------------------------------
// Fingerprint similarity (consumers: src/dct-sim.py:12-50, src/query_db.py:57,76).
// L1 distance matrix between two sets of int8 fingerprints with v_sad_u8 (4 bytes per
// instruction); 128 x 128 distances per workgroup of 256 threads, 8 x 8 per thread, both operand
// tiles staged in LDS KC = 32 dwords (128 bytes of the fingerprints) at a time, row stride 33 dwords,
// the b rows stored in the order the lanes read them (no bank conflict; a wave writes whole 512-byte
// row segments of the result).
// ---------------------------------------------------------------------------
__device__ inline uint32_t load_bytes4(const int8_t* p, int n_valid) {  // n_valid in 1..4
    uint32_t v = 0;
    for (int i = 0; i < n_valid; ++i) v |= (uint32_t)(uint8_t)p[i] << (8 * i);
    for (int i = n_valid; i < 4; ++i) v |= 0x80u << (8 * i);  // xor'ed back to 0 below
    return v;
}

template <bool ALIGNED>
__global__ __launch_bounds__(256) void l1_matrix_kernel(const int8_t* __restrict__ a, int64_t na, int64_t lda,
                                                         const int8_t* __restrict__ b, int64_t nb, int64_t ldb, int d,
                                                         int32_t* __restrict__ out, int64_t ldo) {
    // 128 x 128 distances per workgroup, 8 x 8 per thread: 16 LDS reads per 64 v_sad_u8 (the 4 x 4 tile of round 1 read 8 per
    // 16 and sat on the LDS).  Thread (ty, tx) owns rows 8 ty .. 8 ty + 7 and columns 8 tx .. 8 tx + 7 -- eight consecutive
    // int32 per row, so a wave writes whole 512-byte row segments -- and the b rows sit in LDS in the order the lanes read them
    // (row r at slot (r % 8) * 16 + r / 8: the 16 lanes of a row group read 16 consecutive slots, no bank conflict at the odd
    // row stride; the a rows are broadcast reads).
    constexpr int KC = 32, TILE = 128;  // dwords per chunk, rows / columns per workgroup
    __shared__ uint32_t sa[TILE][KC + 1];
    __shared__ uint32_t sb[TILE][KC + 1];
    const int64_t r0 = (int64_t)blockIdx.y * TILE, c0 = (int64_t)blockIdx.x * TILE;
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    uint32_t acc[8][8] = {};
    const int nd = (d + 3) / 4;
    for (int k0 = 0; k0 < nd; k0 += KC) {
        const int kn = min(KC, nd - k0);
        __syncthreads();
        {   // thread -> dword k = tid & 31 of the rows tid >> 5, + 8, + 16, ...: 128 consecutive bytes of a row per 32 lanes
            const int k = threadIdx.x & (KC - 1);
            const int byte0 = (k0 + k) * 4;
            const int valid = min(4, d - byte0);
            if (k < kn) {
#pragma unroll 4
                for (int r = threadIdx.x >> 5; r < TILE; r += 256 / KC) {
                    uint32_t va = 0x80808080u, vb = 0x80808080u;
                    if (r0 + r < na) {
                        const int8_t* p = a + (r0 + r) * lda + byte0;
                        va = (ALIGNED && valid == 4) ? *reinterpret_cast<const uint32_t*>(p) : load_bytes4(p, valid);
                    }
                    if (c0 + r < nb) {
                        const int8_t* p = b + (c0 + r) * ldb + byte0;
                        vb = (ALIGNED && valid == 4) ? *reinterpret_cast<const uint32_t*>(p) : load_bytes4(p, valid);
                    }
                    sa[r][k] = va ^ 0x80808080u;  // signed -> unsigned order, |x - y| unchanged
                    sb[(r & 7) * 16 + (r >> 3)][k] = vb ^ 0x80808080u;
                }
            }
        }
        __syncthreads();
#pragma unroll 2
        for (int k = 0; k < kn; ++k) {
            uint32_t av[8], bv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) av[i] = sa[ty * 8 + i][k];
#pragma unroll
            for (int j = 0; j < 8; ++j) bv[j] = sb[j * 16 + tx][k];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_sad_u8(av[i], bv[j], acc[i][j]);
        }
    }
    const int64_t c = c0 + tx * 8;
    const bool wide = c + 8 <= nb && (ldo & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int64_t r = r0 + ty * 8 + i;
        if (r >= na) continue;
        int32_t* __restrict__ o = out + r * ldo + c;
        if (wide) {
            typedef int32_t v4i32 __attribute__((ext_vector_type(4)));
            *reinterpret_cast<v4i32*>(o) = (v4i32){(int32_t)acc[i][0], (int32_t)acc[i][1], (int32_t)acc[i][2], (int32_t)acc[i][3]};
            *reinterpret_cast<v4i32*>(o + 4) = (v4i32){(int32_t)acc[i][4], (int32_t)acc[i][5], (int32_t)acc[i][6], (int32_t)acc[i][7]};
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (c + j < nb) o[j] = (int32_t)acc[i][j];
        }
    }
}

// The same tile for rows that start on 16-byte boundaries (lda, ldb and both bases multiples of 16: every fingerprint file and
// tensor this library produces), round 4.  tools/microbench/sad_rate.hip: v_sad_u8 from registers sustains 0.92 of its
// 157 T/s, fed by 16 ds_read_b32 per 64 instructions (the kernel above) 0.83, by 16 ds_read_b128 per 256 0.91 -- and the
// kernel above reached 0.59: beside the narrow LDS reads it fills its tiles with 64 four-byte global loads and as many
// ds_write_b32 per thread and chunk, through per-byte tail code in the same loop.  Here: 16 bytes per lane from HBM / L2 to
// LDS (8 loads + 8 ds_write_b128 per thread and chunk of 128 fingerprint bytes), row stride 36 dwords (16 lanes reading
// 16 bytes each of 16 different rows hit 64 different banks), 4 k-steps per round of LDS reads.
#ifndef DCTFP_TEMPLATES_ONLY   // (a plain kernel: defined in dctfp.hip only, the kernel-family units see the templates)
__global__ __launch_bounds__(256, 4) void l1_matrix16_kernel(const int8_t* __restrict__ a, int64_t na, int64_t lda,
                                                             const int8_t* __restrict__ b, int64_t nb, int64_t ldb, int d,
                                                             int32_t* __restrict__ out, int64_t ldo) {
    constexpr int KC = 32, TILE = 128, LD = KC + 4;  // dwords per chunk, rows / columns per workgroup, LDS row stride
    __shared__ uint32_t sa[TILE * LD];
    __shared__ uint32_t sb[TILE * LD];
    const int64_t r0 = (int64_t)blockIdx.y * TILE, c0 = (int64_t)blockIdx.x * TILE;
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    uint32_t acc[8][8] = {};
    // fill: thread -> 16-byte segment (tid & 7) of the rows tid >> 3, + 32, + 64, + 96.  Addresses = a uniform 64-bit tile base
    // + a 32-bit lane offset (the host checks 128 * ld < 2^31): eight 64-bit row pointers held through the k loop would not fit
    // beside the 64 accumulators.
    const int seg = threadIdx.x & 7, frow = threadIdx.x >> 3;
    const v4u32 flip = {0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};  // signed -> unsigned order, |x - y| unchanged
    const int8_t* __restrict__ abase = a + r0 * lda;
    const int8_t* __restrict__ bbase = b + c0 * ldb;
    const int rows_a = (int)min((int64_t)TILE, na - r0), rows_b = (int)min((int64_t)TILE, nb - c0);
    const uint32_t lda32 = (uint32_t)lda, ldb32 = (uint32_t)ldb;
    // 4 k-steps: 16 bytes of 8 a rows (broadcast reads) and, two at a time, of my 8 b rows from LDS -> 256 v_sad_u8
    auto contract = [&](int kn) {
        for (int k = 0; k < kn; k += 4) {
            v4u32 av[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) av[i] = *reinterpret_cast<const v4u32*>(&sa[(ty * 8 + i) * LD + k]);
#pragma unroll
            for (int h = 0; h < 4; ++h) {   // (64 + 32 + 8 registers)
                v4u32 bv[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) bv[j] = *reinterpret_cast<const v4u32*>(&sb[((2 * h + j) * 16 + tx) * LD + k]);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int i = 0; i < 8; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][2 * h + j] = __builtin_amdgcn_sad_u8(av[i][q], bv[j][q], acc[i][2 * h + j]);
            }
        }
    };
    // the b rows sit in LDS in the order the lanes read them (column c at slot (c % 8) * 16 + c / 8: the 16 lanes of a row group
    // read 16 consecutive slots)
    auto b_slot = [](int r) { return (r & 7) * 16 + (r >> 3); };
    const int d16 = d & ~15;   // whole 16-byte segments; what is left (d % 16 != 0) goes through one more, narrow round below
    for (int byte0 = 0; byte0 < d16; byte0 += KC * 4) {
        const int my0 = byte0 + seg * 16;   // first byte of my segment
        const bool have = my0 < d16;        // (else past the end: both sides equal, no difference)
        __syncthreads();
        {
            v4u32 va[TILE / 32], vb[TILE / 32];
#pragma unroll
            for (int i = 0; i < TILE / 32; ++i) {
                const int r = frow + 32 * i;
                va[i] = flip;
                vb[i] = flip;
                if (have && r < rows_a) va[i] = *reinterpret_cast<const v4u32*>(abase + ((uint32_t)r * lda32 + (uint32_t)my0));
                if (have && r < rows_b) vb[i] = *reinterpret_cast<const v4u32*>(bbase + ((uint32_t)r * ldb32 + (uint32_t)my0));
            }
#pragma unroll
            for (int i = 0; i < TILE / 32; ++i) {
                const int r = frow + 32 * i;
                *reinterpret_cast<v4u32*>(&sa[r * LD + seg * 4]) = va[i] ^ flip;
                *reinterpret_cast<v4u32*>(&sb[b_slot(r) * LD + seg * 4]) = vb[i] ^ flip;
            }
        }
        __syncthreads();
        contract(min(KC, (d16 - byte0) >> 2));
    }
    if (d16 < d) {   // the 1..15 bytes the fingerprints end with: byte loads, one 16-byte segment per row
        __syncthreads();
        if (threadIdx.x < TILE) {
            const int r = threadIdx.x;
            v4u32 va = flip, vb = flip;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = min(4, d - d16 - 4 * q);
                if (n > 0 && r < rows_a) va[q] = load_bytes4(abase + ((uint32_t)r * lda32 + (uint32_t)(d16 + 4 * q)), n);
                if (n > 0 && r < rows_b) vb[q] = load_bytes4(bbase + ((uint32_t)r * ldb32 + (uint32_t)(d16 + 4 * q)), n);
            }
            *reinterpret_cast<v4u32*>(&sa[r * LD]) = va ^ flip;
            *reinterpret_cast<v4u32*>(&sb[b_slot(r) * LD]) = vb ^ flip;
        }
        __syncthreads();
        contract(4);
    }
    const int64_t c = c0 + tx * 8;
    const bool wide = c + 8 <= nb && (ldo & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int64_t r = r0 + ty * 8 + i;
        if (r >= na) continue;
        int32_t* __restrict__ o = out + r * ldo + c;
        if (wide) {
            typedef int32_t v4i32 __attribute__((ext_vector_type(4)));
            *reinterpret_cast<v4i32*>(o) = (v4i32){(int32_t)acc[i][0], (int32_t)acc[i][1], (int32_t)acc[i][2], (int32_t)acc[i][3]};
            *reinterpret_cast<v4i32*>(o + 4) = (v4i32){(int32_t)acc[i][4], (int32_t)acc[i][5], (int32_t)acc[i][6], (int32_t)acc[i][7]};
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (c + j < nb) o[j] = (int32_t)acc[i][j];
        }
    }
}

#endif

// min over every (protein_a, protein_b) block of the distance matrix + the block's last entry
// (domain_sim, src/dct-sim.py:28-50: the max similarity over domain pairs and the similarity
// of the two last = whole-protein fingerprints).
#ifndef DCTFP_TEMPLATES_ONLY   // (a plain kernel: defined in dctfp.hip only, the kernel-family units see the templates)
__global__ void block_min_kernel(const int32_t* __restrict__ dist, int64_t ldo, const int64_t* __restrict__ idx_a,
                                 int64_t npa, const int64_t* __restrict__ idx_b, int64_t npb,
                                 int32_t* __restrict__ out_min, int32_t* __restrict__ out_last) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= npa * npb) return;
    const int64_t pa = t / npb, pb = t % npb;
    const int64_t a0 = idx_a[pa], a1 = idx_a[pa + 1], b0 = idx_b[pb], b1 = idx_b[pb + 1];
    int32_t mn = 0x7fffffff, last = 0x7fffffff;
    for (int64_t r = a0; r < a1; ++r)
        for (int64_t c = b0; c < b1; ++c) {
            last = dist[r * ldo + c];
            mn = min(mn, last);
        }
    out_min[t] = mn;
    out_last[t] = last;
}
#endif


// ---------------------------------------------------------------------------
// k smallest entries of every row of an int32 matrix, ties to the lower column (what a flat L1
// index returns, src/query_db.py:87).  One workgroup per row: 4-pass radix select for the k-th
// smallest value, collection of everything below it, and an ordered ballot walk for the ties at
// the threshold.  Output order within a row is unspecified (the host sorts k entries).
// ---------------------------------------------------------------------------
#ifndef DCTFP_TEMPLATES_ONLY   // (a plain kernel: defined in dctfp.hip only, the kernel-family units see the templates)
__global__ __launch_bounds__(1024) void row_select_kernel(const int32_t* __restrict__ dist, int64_t ld, int64_t n_cols,
                                                           int k, int32_t* __restrict__ out_val,
                                                           int32_t* __restrict__ out_idx) {
    __shared__ int hist[256];
    __shared__ uint32_t s_prefix;
    __shared__ int s_need, s_eq, s_cnt;
    const int32_t* __restrict__ row = dist + (size_t)blockIdx.x * ld;
    int32_t* __restrict__ ov = out_val + (size_t)blockIdx.x * k;
    int32_t* __restrict__ oi = out_idx + (size_t)blockIdx.x * k;
    if (threadIdx.x == 0) {
        s_prefix = 0;
        s_need = k;
        s_cnt = 0;
        s_eq = 0;
    }
    __syncthreads();
    // order-preserving key, smallest first: flip the sign bit
    auto key_of = [](int32_t v) -> uint32_t { return (uint32_t)v ^ 0x80000000u; };
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int b = threadIdx.x; b < 256; b += blockDim.x) hist[b] = 0;
        __syncthreads();
        const uint32_t prefix = s_prefix;
        for (int64_t c = threadIdx.x; c < n_cols; c += blockDim.x) {
            const uint32_t key = key_of(row[c]);
            if (shift == 24 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(key >> shift) & 255u], 1);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int need = s_need, b = 0;
            for (; b < 255; ++b) {
                if (hist[b] >= need) break;
                need -= hist[b];
            }
            s_prefix = prefix | ((uint32_t)b << shift);
            s_need = need;
            s_eq = hist[b];
        }
        __syncthreads();
    }
    const uint32_t thr = s_prefix;
    const int need_eq = s_need;
    const bool all_ties = (s_eq == need_eq);
    for (int64_t c = threadIdx.x; c < n_cols; c += blockDim.x) {
        const int32_t v = row[c];
        const uint32_t key = key_of(v);
        if (key < thr || (all_ties && key == thr)) {
            const int pos = atomicAdd(&s_cnt, 1);
            ov[pos] = v;
            oi[pos] = (int32_t)c;
        }
    }
    __syncthreads();
    if (!all_ties && threadIdx.x < 64) {
        const int lane = threadIdx.x;
        int base = s_cnt, taken = 0;
        for (int64_t c0 = 0; c0 < n_cols && taken < need_eq; c0 += 64) {
            const int64_t c = c0 + lane;
            const int32_t v = (c < n_cols) ? row[c] : 0;
            const bool hit = (c < n_cols) && key_of(v) == thr;
            const unsigned long long m = __ballot(hit);
            const int before = __popcll(m & ((1ull << lane) - 1ull));
            if (hit && taken + before < need_eq) {
                ov[base + taken + before] = v;
                oi[base + taken + before] = (int32_t)c;
            }
            taken += __popcll(m);
        }
    }
}
#endif


// The same selection with the row held in registers (round 4): one workgroup of TH threads loads up to TH * PER entries
// ONCE (the radix select above reads its row six times and counts through LDS atomics that all hit a few bins: 2.6 ms for
// 6 700 rows of 40 000 against 1.2 ms for the distance matrix itself; this kernel: 0.29 ms).
//   1. every thread keeps the minimum of each of its 1024 / TH equal shares of entries; the k-th smallest of those 1024 minima,
//      B, is an upper bound of the k-th smallest entry T (the minima are a subset of the row) -- found by one wave, 16 minima
//      per lane, by bisection;
//   2. one compare per register counts the entries below B.  Fewer than k: T = B, and the entries below B plus the first
//      ties at B (by column) are the answer.  Else the entries below B -- a few more than k -- move to LDS with their columns,
//   3. and T is found among those by bisection on the value, kSelectCap / TH per thread (count = the compare's lane mask, popcount on the
//      scalar unit, one LDS add per wave and step); where more entries equal T than are still wanted, the last column to
//      take by bisection on the column.  More than kSelectCap entries below B (a row of few distinct values): the same
//      bisection over the registers themselves.
// No atomics on the data, no second read.  Rows longer than a workgroup holds go through in segments (grid = rows x
// segments): each leaves its k candidates (value, column) in a scratch, and a second launch (COLS: columns come with the
// values) selects among those.  A segment with fewer than k entries pads its candidates with (INT32_MAX, INT32_MAX), which
// lose every tie.
constexpr int kSelectCap = 4096;   // candidates the LDS holds
constexpr int kSelectCounters = 128;
template <int NWAVES>
struct SelectSharedT {
    uint32_t cnt[kSelectCounters];  // one counter per counting step: no reset, one barrier per step
    uint32_t bound, row_min, pos, fill;
    uint32_t hist[2][NWAVES * 64];  // kth_smallest: a 64-bin histogram per wave, two sets in turn
    uint32_t wave_val[NWAVES], wave_cnt[NWAVES];
};
typedef SelectSharedT<16> SelectShared;
__device__ inline uint32_t lane_votes(bool p) { return (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(p)); }

// The kk-th smallest of E keys per thread in [lo, hi] (which must hold it; n_less = count(key < lo) on entry, = count(key <
// result) on return).  Slots without an entry hold 0xffffffff.
// By HISTOGRAM passes (round 5): a pass sorts the keys of [lo, hi] into 64 equal bins -- every wave into a histogram of its own
// in LDS (ds_add without return: no wave waits for another's counter, no two waves share one), one barrier, then every wave
// adds the histograms bin by bin (lane = bin), scans them across its lanes and keeps the bin that holds the answer: six bits of
// the range per barrier.  The binary search this replaces took one bit per barrier (32 steps from the full key range: a third of
// the contact top-k's time, most of the row select's after its single read); counting fifteen thresholds per step on the scalar
// unit (compare, lane mask, popcount, add per key and threshold) was slower still -- the scalar unit is one per CU.
template <int E, typename SH>
__device__ inline uint32_t kth_smallest(const uint32_t (&key)[E], uint32_t lo, uint32_t hi, uint32_t& n_less, uint32_t kk, SH& sh,
                                        int& step) {
    const int lane = threadIdx.x & 63, wave = (int)(threadIdx.x >> 6);
    const int n_waves = (int)((blockDim.x + 63u) >> 6);
    while (lo < hi) {
        const uint32_t span = hi - lo;
        const int sft = max(0, 26 - (int)__builtin_clz(span));       // bit_width(span) - 6: bin = (key - lo) >> sft < 64
        uint32_t* __restrict__ h = sh.hist[step & 1];
        h[wave * 64 + lane] = 0;   // (the LDS operations of one wave complete in order: my wave's row is clear before it counts)
#pragma unroll
        for (int j = 0; j < E; ++j)
            if (key[j] >= lo && key[j] <= hi) atomicAdd(&h[wave * 64 + (int)((key[j] - lo) >> sft)], 1u);
        __syncthreads();
        uint32_t cum = 0;
        for (int w = 0; w < n_waves; ++w) cum += h[w * 64 + lane];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {   // inclusive prefix sums over the bins
            const uint32_t up = (uint32_t)__shfl_up((int)cum, off);
            cum += lane >= off ? up : 0u;
        }
        const unsigned long long ok = __builtin_amdgcn_ballot_w64(n_less + cum >= kk);   // (not empty: count(key <= hi) >= kk)
        const int b = ok != 0 ? (int)__builtin_ctzll(ok) : 63;
        if (b > 0) n_less += (uint32_t)__builtin_amdgcn_readlane((int)cum, b - 1);
        lo += (uint32_t)b << sft;
        hi = min(hi, lo + ((1u << sft) - 1u));
        ++step;   // (the next pass takes the other set of histograms: a wave still adding up this one is not disturbed)
    }
    return lo;
}

// The same for ONE wave on its own keys (no barrier; `h` = 64 counters of its own), at most `passes` passes: returns the upper
// end of the range the answer is known to lie in -- the answer itself once the range has shrunk to one key.
template <int E>
__device__ inline uint32_t wave_kth_upper(const uint32_t (&key)[E], uint32_t lo, uint32_t hi, uint32_t kk, uint32_t* __restrict__ h, int passes) {
    const int lane = threadIdx.x & 63;
    uint32_t n_less = 0;
    for (int p = 0; p < passes && lo < hi; ++p) {
        const uint32_t span = hi - lo;
        const int sft = max(0, 26 - (int)__builtin_clz(span));
        h[lane] = 0;
#pragma unroll
        for (int j = 0; j < E; ++j)
            if (key[j] >= lo && key[j] <= hi) atomicAdd(&h[(int)((key[j] - lo) >> sft)], 1u);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t cum = h[lane];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)cum, off);
            cum += lane >= off ? up : 0u;
        }
        const unsigned long long ok = __builtin_amdgcn_ballot_w64(n_less + cum >= kk);
        const int b = ok != 0 ? (int)__builtin_ctzll(ok) : 63;
        if (b > 0) n_less += (uint32_t)__builtin_amdgcn_readlane((int)cum, b - 1);
        lo += (uint32_t)b << sft;
        hi = min(hi, lo + ((1u << sft) - 1u));
        __builtin_amdgcn_wave_barrier();
    }
    return hi;
}

// The kk smallest of E (key, column) pairs per thread, ties to the lowest columns: T = the kk-th smallest key by bisection
// in [lo, hi] (n_less = count(key < lo) on entry), then emit(output slot, key, column) for each of them.  Slots without an entry hold
// key 0xffffffff and a column > col_hi; `valid(j)` tells them from real entries of that value.
template <int E, typename SH, typename ColF, typename ValidF, typename EmitF>
__device__ inline void bisect_emit(const uint32_t (&key)[E], ColF col_of, ValidF valid, uint32_t lo, uint32_t hi, uint32_t n_less,
                                   uint32_t col_hi, uint32_t kk, SH& sh, int& step, EmitF emit) {
    const int lane = threadIdx.x & 63;
    auto total = [&](uint32_t w) {   // wave counts -> sum over the workgroup; every thread gets it
        if (lane == 0) atomicAdd(&sh.cnt[step], w);
        __syncthreads();
        const uint32_t t = sh.cnt[step];
        ++step;
        return t;
    };
    const uint32_t thr = kth_smallest<E, SH>(key, lo, hi, n_less, kk, sh, step);
    const uint32_t need = kk - n_less;   // entries equal to thr still wanted (>= 1): the ones in the lowest columns
    uint32_t ties = 0;
#pragma unroll
    for (int j = 0; j < E; ++j) ties += lane_votes(valid(j) && key[j] == thr);
    ties = total(ties);
    uint32_t last_col = 0xffffffffu;   // take the ties up to this column
    if (ties > need) {
        uint32_t clo = 0, chi = col_hi;
        while (clo < chi) {
            const uint32_t mid = clo + ((chi - clo) >> 1);
            uint32_t c = 0;   // (empty slots: column > col_hi > mid)
#pragma unroll
            for (int j = 0; j < E; ++j) c += lane_votes(key[j] == thr && col_of(j) <= mid);
            c = total(c);
            if (c >= need) chi = mid;
            else clo = mid + 1;
        }
        last_col = clo;
    }
    // output slots: one LDS add per wave and register (the lanes that take an entry rank themselves inside the wave's block)
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const bool take = valid(j) && (key[j] < thr || (key[j] == thr && col_of(j) <= last_col));
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(take);
        if (mask != 0) {   // (wave-uniform)
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&sh.pos, (uint32_t)__builtin_popcountll(mask));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (take) emit(base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u)), key[j], col_of(j));
        }
    }
}

// TH threads hold PER entries each (TH * PER = the segment).  1024 x 40 needs 118 registers: one workgroup per CU, whose load
// and counting phases nobody overlaps; 512 x 80 fits two workgroups per CU at the same 128-register budget, and one loads
// while the other counts.  Every thread gives 1024 / TH minima (over equal shares of its entries) to the bound.
template <int PER, bool COLS, int TH>
__global__ __launch_bounds__(TH, 4) void row_select_reg_kernel(const int32_t* __restrict__ src_val, const int32_t* __restrict__ src_col,
                                                                           int64_t ld, int64_t n_cols, int64_t seg_cols, int64_t n_seg, int k,
                                                                           int32_t* __restrict__ out_val, int32_t* __restrict__ out_idx) {
    __shared__ SelectShared sh;
    __shared__ uint32_t s_min[1024];
    __shared__ uint32_t s_ckey[kSelectCap], s_ccol[kSelectCap];
    const int64_t row = (int64_t)blockIdx.x / n_seg, seg = (int64_t)blockIdx.x % n_seg;
    const int64_t c_first = seg * seg_cols;
    const int n = (int)min(seg_cols, n_cols - c_first);
    const uint32_t kk = (uint32_t)min(k, n);
    const int32_t* __restrict__ v = src_val + row * ld + c_first;
    const int32_t* __restrict__ cs = COLS ? src_col + row * ld + c_first : nullptr;
    int32_t* __restrict__ ov = out_val + ((size_t)row * n_seg + seg) * k;
    int32_t* __restrict__ oi = out_idx + ((size_t)row * n_seg + seg) * k;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int q = tid; q < kSelectCounters; q += TH) sh.cnt[q] = 0;
    if (tid == 0) {
        sh.pos = 0;
        sh.fill = 0;
    }
    // order-preserving keys, smallest first: sign bit flipped.  Element j of thread t is entry j * TH + t of the segment.
    constexpr int NMIN = 1024 / TH, SHARE = PER / NMIN;   // minima per thread, entries behind each
    static_assert(TH * NMIN == 1024 && SHARE * NMIN == PER, "1024 minima over equal shares");
    uint32_t key[PER];
    uint32_t col[COLS ? PER : 1];
    uint32_t kmin[NMIN];
#pragma unroll
    for (int q = 0; q < NMIN; ++q) kmin[q] = 0xffffffffu;
    // All loads first, unconditionally: a load under `if (c < n)` is waited for before the next one is issued -- 40 round
    // trips to HBM per wave.  Through a buffer descriptor that ends with the segment: one lane offset for all of them
    // (40 clamped addresses would cost 40 more registers), a slot past the end reads 0 and is overwritten below.
    auto load_row = [&](uint32_t (&kk_)[PER], uint32_t (&cc_)[COLS ? PER : 1]) {
        const __amdgpu_buffer_rsrc_t vb = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(v), 0, n * 4, 0x00020000);
#pragma unroll
        for (int j = 0; j < PER; ++j) kk_[j] = __builtin_amdgcn_raw_buffer_load_b32(vb, tid * 4, j * TH * 4, DCTFP_STREAM_AUX);
        if (COLS) {
            const __amdgpu_buffer_rsrc_t cb = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(cs), 0, n * 4, 0x00020000);
#pragma unroll
            for (int j = 0; j < PER; ++j) cc_[j] = __builtin_amdgcn_raw_buffer_load_b32(cb, tid * 4, j * TH * 4, 0);
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const bool have = j * TH + tid < n;
            kk_[j] = have ? kk_[j] ^ 0x80000000u : 0xffffffffu;
            if (COLS) cc_[j] = have ? cc_[j] : 0xffffffffu;
        }
    };
    load_row(key, col);
#pragma unroll
    for (int j = 0; j < PER; ++j) kmin[j / SHARE] = min(kmin[j / SHARE], key[j]);
    const int my_n = tid < n ? (n - tid + TH - 1) / TH : 0;  // my entries: j < my_n
    auto col_of = [&](int j) { return COLS ? col[j] : (uint32_t)(j * TH + tid); };   // (segment-local, or as it came)
    const uint32_t col_hi = COLS ? 0x7fffffffu : (uint32_t)(n - 1);
    auto emit = [&](uint32_t pos, uint32_t key_e, uint32_t col_e) {
        ov[pos] = (int32_t)(key_e ^ 0x80000000u);
        oi[pos] = COLS ? (int32_t)col_e : (int32_t)(c_first + col_e);
    };
#pragma unroll
    for (int q = 0; q < NMIN; ++q) s_min[q * TH + tid] = kmin[q];
    __syncthreads();
#if defined(DCTFP_RSEL_STOP) && DCTFP_RSEL_STOP == 1
    if (tid == 0) ov[0] = (int32_t)s_min[5];
    return;
#endif
    if (tid < 64) {   // 1.: the kk-th smallest of the 1024 minima.  With one minimum per thread at least kk of them are real (kk <=
        // min(n, 1024)); with two, a short segment may fill fewer than kk shares: the bound is then 0xffffffff -- an upper bound
        // all the same, and what follows tells real entries of that value from empty slots.
        uint32_t lo = 0xffffffffu, hi = 0;   // (the minima stay in LDS: the wave's registers hold its 40 entries)
        for (int i = 0; i < 16; ++i) {
            const uint32_t m = s_min[i * 64 + lane];
            lo = min(lo, m);
            hi = max(hi, m);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            lo = min(lo, (uint32_t)__shfl_xor((int)lo, off));
            hi = max(hi, (uint32_t)__shfl_xor((int)hi, off));
        }
        if (lane == 0) sh.row_min = lo;
        // (any value with at least kk minima up to it is a bound: the bisection stops at 1/256 of the minima's range -- eight
        //  steps instead of sixteen for a fraction of a candidate more -- and hands out the upper end)
        const uint32_t tol = (hi - lo) >> 8;
        while (hi - lo > tol) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            uint32_t c = 0;
            for (int i = 0; i < 16; ++i) c += lane_votes(s_min[i * 64 + lane] <= mid);
            if (c >= kk) hi = mid;
            else lo = mid + 1;
        }
        if (lane == 0) sh.bound = hi;
    }
    __syncthreads();
    const uint32_t bound = sh.bound, row_min = sh.row_min;
#if defined(DCTFP_RSEL_STOP) && DCTFP_RSEL_STOP == 2
    if (tid == 0) ov[0] = (int32_t)(bound + key[3]);
    return;
#endif
    int step = 0;
    // 2.: the entries below the bound, counted and moved to LDS in ONE pass over the registers: every wave appends to a part of the
    // buffer of its own (kSelectCap / waves slots; its fill count is a wave-uniform register: no LDS counter, no atomics, no wait
    // between two registers) -- counting them first (a compare, a lane mask, popcount and add per register) and compacting them
    // through one shared LDS counter afterwards was 0.2 of the kernel's 0.29 ms, more than its read of the row.  (Empty slots hold
    // 0xffffffff: never below.)
    constexpr int NWV = TH / 64, WCAP = kSelectCap / NWV;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t w_fill = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const bool take = key[j] < bound;
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(take);
        if (mask != 0) {   // (wave-uniform)
            const uint32_t at = w_fill + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            if (take && at < (uint32_t)WCAP) {
                s_ckey[wave * WCAP + at] = key[j];
                s_ccol[wave * WCAP + at] = col_of(j);
            }
            w_fill += (uint32_t)__builtin_popcountll(mask);
        }
    }
    if (lane == 0) sh.wave_cnt[wave] = w_fill;
    __syncthreads();
    uint32_t below = 0;
    bool fits = true;
#pragma unroll
    for (int w = 0; w < NWV; ++w) {
        below += sh.wave_cnt[w];
        fits = fits && sh.wave_cnt[w] <= (uint32_t)WCAP;
    }
#if defined(DCTFP_RSEL_STOP) && DCTFP_RSEL_STOP == 3
    if (tid == 0) ov[0] = (int32_t)below;
    return;
#endif
    if (below >= kk && fits) {                 // 3.: T < bound, among the `below` entries under the bound
        constexpr int E = kSelectCap / TH;
        uint32_t ck[E], cc[E];
        bool have[E];
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const uint32_t e = (uint32_t)(i * TH + tid);
            have[i] = (e % (uint32_t)WCAP) < sh.wave_cnt[e / (uint32_t)WCAP];
            ck[i] = have[i] ? s_ckey[e] : 0xffffffffu;
            cc[i] = have[i] ? s_ccol[e] : 0xffffffffu;
        }
        bisect_emit<E>(ck, [&](int i) { return cc[i]; }, [&](int i) { return have[i]; }, row_min, bound - 1u, 0u,
                       col_hi, kk, sh, step, emit);
    } else {
        // Rare: fewer than kk entries below the bound (T = the bound itself), or a row of few distinct values whose entries below
        // the bound do not fit the buffer.  The row is READ AGAIN for it: held in its 80 registers across the selection above,
        // it cost that path 40 spilled registers per lane.
        uint32_t key2[PER];
        uint32_t col2[COLS ? PER : 1];
        load_row(key2, col2);
        auto col_of2 = [&](int j) { return COLS ? col2[j] : (uint32_t)(j * TH + tid); };
        if (below < kk) bisect_emit<PER>(key2, col_of2, [&](int j) { return j < my_n; }, bound, bound, below, col_hi, kk, sh, step, emit);
        else bisect_emit<PER>(key2, col_of2, [&](int j) { return j < my_n; }, row_min, bound - 1u, 0u, col_hi, kk, sh, step, emit);
    }
    for (uint32_t pos = kk + (uint32_t)tid; pos < (uint32_t)k; pos += TH) {   // a segment shorter than k: candidates that lose every tie
        ov[pos] = 0x7fffffff;
        oi[pos] = 0x7fffffff;
    }
}


// The k entries a row select left for a row, in the order a flat index scan reports them: value ascending, ties by column
// (round 4: numpy's lexsort of 6 700 x 100 survivors took 134 ms, and as long again to apply -- against 1.5 ms for the distance
// matrix and the selection together).  One workgroup of N / 2 threads per row: 64-bit keys (order-preserving value above the
// column) through a bitonic network in LDS, written back in place.  N = capacity (a power of two >= k).
template <int N>
__global__ __launch_bounds__(N / 2) void row_order_kernel(int32_t* __restrict__ val, int32_t* __restrict__ idx, int k) {
    __shared__ unsigned long long keys[N];
    int32_t* __restrict__ v = val + (size_t)blockIdx.x * k;
    int32_t* __restrict__ c = idx + (size_t)blockIdx.x * k;
    for (int p = threadIdx.x; p < N; p += N / 2)
        keys[p] = p < k ? ((unsigned long long)((uint32_t)v[p] ^ 0x80000000u) << 32) | (uint32_t)c[p] : ~0ull;
    __syncthreads();
    for (int size = 2; size <= N; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const int q = threadIdx.x;
            const int lo = ((q & ~(stride - 1)) << 1) | (q & (stride - 1));
            const int hi = lo | stride;
            const bool up = (lo & size) == 0;
            const unsigned long long a = keys[lo], b = keys[hi];
            if ((a > b) == up) {
                keys[lo] = b;
                keys[hi] = a;
            }
            __syncthreads();
        }
    }
    for (int p = threadIdx.x; p < k; p += N / 2) {
        v[p] = (int32_t)((uint32_t)(keys[p] >> 32) ^ 0x80000000u);
        c[p] = (int32_t)(uint32_t)keys[p];
    }
}

// ---------------------------------------------------------------------------
// Contact top-k in two reads of the map (round 4; contact_topk_kernel above reads it six times and counts through LDS
// atomics that mostly hit one bin).  Keys are the inverted order-preserving images of the values, so "largest value first,
// ties in (i, j) order" is "smallest key first, ties to the lowest i << 16 | j" -- the selection of row_select_reg_kernel.
//   1. first read: every thread keeps the M smallest keys it sees (M = 2, 4, 8 by k; a lane's position inside a 64-entry
//      chunk rotates with the row, so that the near-diagonal band of a contact map does not land on the same few lanes);
//      the k-th smallest of those 1024 * M keys bounds the k-th smallest key of the map from above (they are a subset);
//   2. second read: the entries up to the bound -- a few more than k -- go to LDS with their (i, j);
//   3. the k smallest among them, ties by (i, j): bisect_emit.  Values are read back from the map (their original bits).
// More than kTopkCap entries up to the bound (plateaus of equal values), or k > 6144: out_n = -1, and the radix select,
// launched behind this kernel with redo_only, does that protein.
// ---------------------------------------------------------------------------
constexpr int kTopkCap = 8192;
template <int M>
__device__ inline void topk2_run(const TopkJob& job, SelectShared& sh, uint32_t* __restrict__ s_key, uint32_t* __restrict__ s_ij,
                                 int32_t* __restrict__ out_i, int32_t* __restrict__ out_j, float* __restrict__ out_v,
                                 int32_t* __restrict__ out_n) {
    const int L = job.n_res, last_row = L - 6;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t kk = (uint32_t)job.k;
    auto ikey = [](float v) { return ~topk_key(v); };
    // The candidates of a map, a wave per row pair: row i has L - 5 - i of them, so the wave takes its n-th row from the top
    // together with its n-th from the bottom -- about (L - 5) / 64 + 1 chunks of 64 per pair whatever n is -- and requests G of
    // them at once (one row at a time left 4 useful loads in flight at L = 500: a round trip to HBM per row and wave).
    // f(i, j, value, valid) sees every candidate once; a lane's place inside a chunk rotates with the row.
    // Loads go through a buffer descriptor that ends with the map (a chunk may reach past the end of its row, and of the last
    // rows' allocation: those lanes read what follows or 0 and are not counted): row and chunk in the scalar offset, the lane's
    // rotated place in one register per row.
    const __amdgpu_buffer_rsrc_t mb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(job.map), 0, (int)((int64_t)L * job.ld * 4), 0x00020000);
    auto stream = [&](auto f) {
        constexpr int G = 12;
        const int n_rows = last_row >= wave ? (last_row - wave) / 16 + 1 : 0;   // rows wave, wave + 16, ...
        for (int n = 0; 2 * n < n_rows; ++n) {
            const int ra = wave + 16 * n, rb = wave + 16 * (n_rows - 1 - n);
            const int ca = (L - 5 - ra + 63) >> 6, cb = rb != ra ? (L - 5 - rb + 63) >> 6 : 0;
            const int la = (lane + ra * 13) & 63, lb = (lane + rb * 13) & 63;
            for (int s0 = 0; s0 < ca + cb; s0 += G) {
                float v[G];
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const int sl = s0 + u;                                    // (everything about a slot but the lane is wave-uniform)
                    const int i = sl < ca ? ra : rb, c = sl < ca ? sl : sl - ca;
                    v[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(mb, (sl < ca ? la : lb) * 4,
                                                                                          (int)(((int64_t)i * job.ld + i + 5 + 64 * c) * 4), 0));
                }
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const int sl = s0 + u;
                    const int i = sl < ca ? ra : rb, c = sl < ca ? sl : sl - ca;
                    const int j = i + 5 + 64 * c + (sl < ca ? la : lb);
                    f(i, j, v[u], sl < ca + cb && j < L);
                }
            }
        }
    };
    // ---- 1. my M smallest keys, ascending
    uint32_t t[M];
#pragma unroll
    for (int m = 0; m < M; ++m) t[m] = 0xffffffffu;
    stream([&](int, int, float v, bool valid) {
        uint32_t x = valid ? ikey(v) : 0xffffffffu;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const uint32_t lo = min(t[m], x);
            x = max(t[m], x);
            t[m] = lo;
        }
    });
    int step = 0;
    uint32_t n_less = 0;
    const uint32_t bound = kth_smallest<M>(t, 0u, 0xffffffffu, n_less, kk, sh, step);
    // ---- 2. the entries up to the bound -> LDS
    stream([&](int i, int j, float v, bool valid) {
        const uint32_t x = ikey(v);
        const bool take = valid && x <= bound;
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(take);
        if (mask != 0) {   // (wave-uniform)
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&sh.fill, (uint32_t)__builtin_popcountll(mask));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            const uint32_t at = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            if (take && at < (uint32_t)kTopkCap) {
                s_key[at] = x;
                s_ij[at] = ((uint32_t)i << 16) | (uint32_t)j;
            }
        }
    });
    __syncthreads();
    const uint32_t n_cand = sh.fill;
    if (n_cand > (uint32_t)kTopkCap) {   // plateaus: the radix select behind this kernel
        if (tid == 0) out_n[job.orig] = -1;
        return;
    }
    // ---- 3. the kk smallest of the candidates, ties to the lowest (i, j)
    constexpr int E = kTopkCap / 1024;
    uint32_t ck[E], cij[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const uint32_t p = (uint32_t)(e * 1024 + tid);
        ck[e] = p < n_cand ? s_key[p] : 0xffffffffu;
        cij[e] = p < n_cand ? s_ij[p] : 0xffffffffu;
    }
    int32_t* __restrict__ oi = out_i + job.out_off;
    int32_t* __restrict__ oj = out_j + job.out_off;
    float* __restrict__ ov = out_v + job.out_off;
    bisect_emit<E>(ck, [&](int e) { return cij[e]; }, [&](int e) { return (uint32_t)(e * 1024 + tid) < n_cand; }, 0u, bound, 0u, 0xfffffffeu, kk, sh,
                   step, [&](uint32_t pos, uint32_t, uint32_t ij) {
                       const int i = (int)(ij >> 16), j = (int)(ij & 0xffffu);
                       oi[pos] = i;
                       oj[pos] = j;
                       ov[pos] = job.map[(size_t)i * job.ld + j];
                   });
    if (tid == 0) out_n[job.orig] = job.k;
}

#ifndef DCTFP_TOPK_WAVES
#define DCTFP_TOPK_WAVES 8   // waves per SIMD the allocation is held to: 8 = two workgroups per CU (nine set-up values spilled)
#endif
#ifndef DCTFP_TEMPLATES_ONLY   // (a plain kernel: defined in dctfp.hip only, the kernel-family units see the templates)
__global__ __launch_bounds__(1024, DCTFP_TOPK_WAVES) void contact_topk2_kernel(const TopkJob* __restrict__ jobs, int32_t* __restrict__ out_i,
                                                              int32_t* __restrict__ out_j, float* __restrict__ out_v,
                                                              int32_t* __restrict__ out_n, int redo_only) {
    __shared__ SelectShared sh;
    __shared__ uint32_t s_key[kTopkCap], s_ij[kTopkCap];
    const TopkJob job = jobs[blockIdx.x];
    if (redo_only && out_n[job.orig] >= 0) return;   // (behind contact_topk1_kernel: only what that one handed back)
    if (threadIdx.x < kSelectCounters) sh.cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        sh.pos = 0;
        sh.fill = 0;
    }
    __syncthreads();
    if (job.k <= 0 || job.n_res < 6) {
        if (threadIdx.x == 0) out_n[job.orig] = 0;
        return;
    }
    if (job.k > 6144 || job.n_res > 65535 || (int64_t)job.n_res * job.ld >= ((int64_t)1 << 29)) {   // (more than the thread minima bound
        // tightly / (i, j) beyond 16 bits each / a map beyond the 32-bit byte offsets of the buffer loads)
        if (threadIdx.x == 0) out_n[job.orig] = -1;
        return;
    }
    if (job.k <= 1400) topk2_run<2>(job, sh, s_key, s_ij, out_i, out_j, out_v, out_n);
    else if (job.k <= 3000) topk2_run<4>(job, sh, s_key, s_ij, out_i, out_j, out_v, out_n);
    else topk2_run<8>(job, sh, s_key, s_ij, out_i, out_j, out_v, out_n);
}
#endif

// ---------------------------------------------------------------------------
// Contact top-k in ONE read of the map (round 5; the kernel above streams the candidate triangle twice, 4 bytes per lane).
//   1. a SAMPLE of the triangle stays in registers: four 16-byte loads per lane (VEC4) = 16 384 entries per map, from four of
//      every wave's row pairs spread over its rows, a different column chunk of each (the waves interleave rows, so the sample
//      is stratified over the whole map and over the distance from the diagonal).  With n_s valid entries among them and N in
//      the triangle, the r-th smallest sample key, r = ceil(2 kk n_s / N), is a bound b0 that about 2 kk keys of the map stay
//      under (sixteen-way selection in registers: 8 barriers) -- the whole map in the sample: r = kk, b0 exact;
//   2. the rest of the triangle streams by ONCE (the sampled slots are not read again): keys <= b0 go to LDS with their (i, j),
//      the sampled ones straight from their registers;
//   3. kk <= candidates <= kTopkCap: the kk smallest among them, ties by (i, j), exactly as in the two-read kernel (every key
//      equal to the kk-th is <= b0, so all ties are among the candidates).  Fewer than kk (the sample misjudged the map: for
//      independent entries a nine-sigma event, any map is possible) or more than the LDS holds (plateaus of equal values),
//      k > 3000, L > 65 535: out_n = -1, and the two-read kernel launched behind this one (redo_only) does that protein.
// A wave takes its n-th row from the top together with its n-th from the bottom as above; the slots (row, chunk) of all its
// pairs form one sequence, G of them requested at once whatever pair they belong to (at L = 500 a pair has three 256-column
// chunks: per-pair batches would leave three loads in flight).
// ---------------------------------------------------------------------------
template <bool VEC4, int NW>
struct TopkSlots {   // the (row, chunk) sequence of one wave; everything here is wave-uniform
    static constexpr int W = VEC4 ? 4 : 1, CH = 64 * W;
    int L, wave, n_rows, n_pairs;
    int n, sl, ra, rb, ca, cb;
    __device__ inline static int first_col(int i) { return VEC4 ? ((i + 5) & ~3) : i + 5; }
    __device__ inline int chunks(int i) const { return (L - first_col(i) + CH - 1) / CH; }
    __device__ inline void set_pair(int pair) {
        n = pair;
        sl = 0;
        if (pair < n_pairs) {
            ra = wave + NW * pair;
            rb = wave + NW * (n_rows - 1 - pair);
            ca = chunks(ra);
            cb = rb != ra ? chunks(rb) : 0;
        } else {
            ra = rb = 0;
            ca = cb = 0;
        }
    }
    __device__ inline void init(int L_, int wave_) {
        L = L_;
        wave = wave_;
        const int last_row = L - 6;
        n_rows = last_row >= wave ? (last_row - wave) / NW + 1 : 0;   // rows wave, wave + NW, ...
        n_pairs = (n_rows + 1) / 2;
        set_pair(0);
    }
    __device__ inline bool done() const { return n >= n_pairs; }
    __device__ inline int row() const { return sl < ca ? ra : rb; }
    __device__ inline int col0() const { return first_col(row()) + CH * (sl < ca ? sl : sl - ca); }   // lane 0's first column
    __device__ inline void next() {
        if (++sl >= ca + cb) set_pair(n + 1);
    }
};

template <bool VEC4, int NW>
__device__ inline void topk1_run(const TopkJob& job, SelectSharedT<NW>& sh, uint32_t* __restrict__ s_key, uint32_t* __restrict__ s_ij,
                                 uint16_t* __restrict__ s_rowt, uint32_t* __restrict__ s_rowp, int32_t* __restrict__ out_i,
                                 int32_t* __restrict__ out_j, float* __restrict__ out_v, int32_t* __restrict__ out_n) {
    constexpr int kRowTies = NW == 8 ? 576 : 1216;   // rows whose ties at the bound are counted (k <= 1 400 -> L <= 538; k <= 3 000 -> L <= 1 153)
    constexpr int W = VEC4 ? 4 : 1, NS = 4;   // entries per lane and load; sample loads per lane
    constexpr int WCAP = 512;                // candidates a wave may keep (its own part of the LDS buffer: no atomics)
    constexpr int TH = NW * 64;
    typedef typename std::conditional<VEC4, v4f, float>::type LV;
    const int L = job.n_res;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t kk = (uint32_t)job.k;
    auto ikey = [](float v) { return ~topk_key(v); };
    // A load = 64 lanes x W columns of ONE row through a descriptor that ends with that row: the lanes past column L -- a chunk
    // is 256 columns, a row's candidates rarely a multiple of it -- read 0 WITHOUT a memory access (through a descriptor of the
    // whole map they fetched the head of the next row: + 35 % of HBM traffic at L = 500).  i < 0: no such slot, nothing read.
    auto load = [&](int i, int c0) -> LV {   // lane l: columns c0 + W l .. + W - 1 of row i
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(job.map) + (int64_t)max(i, 0) * job.ld, 0,
                                                                               i >= 0 ? L * 4 : 0, 0x00020000);
        if constexpr (VEC4) return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rb, lane * 16, c0 * 4, 0));
        else return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, lane * 4, c0 * 4, 0));
    };
    auto elem = [](const LV& v, int e) -> float {
        if constexpr (VEC4) return v[e];
        else return v;
    };
    // ---- 1. the sample: slot (u + 1) mod slots-of-the-pair of pair u * stride, u < NS
    TopkSlots<VEC4, NW> it;
    it.init(L, wave);
    const int stride = it.n_pairs >= NS ? it.n_pairs / NS : 1;
    int s_row[NS], s_c0[NS], s_pair[NS], s_slot[NS];
    LV sv[NS];
    // (slots first, loads after them in straight-line code: a load under a branch -- even a wave-uniform one -- is waited for
    //  before the next is issued, a round trip to HBM per slot; a slot that does not exist reads past the end of the buffer: 0,
    //  no memory access)
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        TopkSlots<VEC4, NW> p = it;
        p.set_pair(u * stride);
        s_pair[u] = s_slot[u] = -1;
        s_row[u] = -1;
        s_c0[u] = 0;
        if (!p.done()) {
            const int want = (u + 1) % (p.ca + p.cb);
            for (int q = 0; q < want; ++q) p.next();
            s_pair[u] = p.n;
            s_slot[u] = p.sl;
            s_row[u] = p.row();
            s_c0[u] = p.col0();
        }
    }
#pragma unroll
    for (int u = 0; u < NS; ++u) sv[u] = load(s_row[u], s_c0[u]);
    uint32_t sk[NS * W];
    uint32_t n_w = 0, s_valid = 0;   // valid sample entries of this wave; bit u W + e: that sample slot of mine holds a candidate
    uint32_t kmin = 0xffffffffu, kmax = 0;
#pragma unroll
    for (int u = 0; u < NS; ++u)
#pragma unroll
        for (int e = 0; e < W; ++e) {
            const int j = s_c0[u] + W * lane + e;
            const bool valid = s_pair[u] >= 0 && j >= s_row[u] + 5 && j < L;
            const uint32_t x = ikey(elem(sv[u], e));
            sk[u * W + e] = valid ? x : 0xffffffffu;
            s_valid |= (valid ? 1u : 0u) << (u * W + e);
            n_w += lane_votes(valid);
            kmin = min(kmin, valid ? x : 0xffffffffu);
            kmax = max(kmax, valid ? x : 0u);
        }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, off));
        kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off));
    }
    // Every wave bounds the map from ITS sample (its rows are spread over the whole map): with n_w valid entries of the N in the
    // triangle, the r-th smallest key of the sample, r = ceil(2 kk n_w / N), is a value that about 2 kk keys of the map stay
    // under.  Wave-local (three histogram passes over the wave's own 64 counters: the range to 1 / 2^18, its upper end taken),
    // no barrier; the workgroup then takes the 12th smallest of its 16 wave bounds.  The whole map in the samples (L < ~140):
    // exact selection below instead.
    const uint64_t n_all = (uint64_t)(L - 5) * (uint64_t)(L - 4) / 2;
    {
        uint32_t rank = (uint32_t)min((uint64_t)n_w, (3ull * kk * n_w + 2 * n_all - 1) / (2 * n_all));
        rank = max(rank, min(n_w, 4u));
        const uint32_t wb = n_w > 0 && kmin <= kmax ? wave_kth_upper<NS * W>(sk, kmin, kmax, rank, &sh.hist[0][wave * 64], 3) : 0xffffffffu;
        if (lane == 0) {
            sh.wave_val[wave] = wb;
            sh.wave_cnt[wave] = n_w;
        }
    }
    __syncthreads();
    uint32_t n_s = 0;
    for (int w = 0; w < NW; ++w) n_s += sh.wave_cnt[w];
    const bool whole = n_s >= n_all;   // every candidate of the map is in the sample
    int step = 0;
    uint32_t bound;
    if (whole) {
        uint32_t n_less = 0;
        bound = kth_smallest<NS * W>(sk, 0u, 0xffffffffu, n_less, kk, sh, step);
    } else {   // the wave bound three quarters up their sorted list (ties by wave): a little above their median
        const uint32_t v = sh.wave_val[lane % NW];
        uint32_t below = 0;
#pragma unroll
        for (int m = 0; m < NW; ++m) {
            const uint32_t vm = (uint32_t)__builtin_amdgcn_readlane((int)v, m);
            below += (vm < v || (vm == v && m < lane % NW)) ? 1u : 0u;
        }
        const unsigned long long pick = __builtin_amdgcn_ballot_w64(lane < NW && below == (uint32_t)(3 * NW / 4 - 1));
        bound = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)__builtin_ctzll(pick));
    }
#if defined(DCTFP_TOPK1_STOP) && DCTFP_TOPK1_STOP == 1
    if (tid == 0) out_n[job.orig] = (int)bound;
    return;
#endif
    // ---- 2. keys BELOW the bound -> this wave's part of the LDS buffer: the sample from its registers, the rest of the triangle
    // as it streams by.  The stream tests the VALUE against the bound's value (one compare: above it, or not comparable -- NaNs
    // pass and are judged by their key like everything that passes); key, validity and position only for what passed.
    // Entries EQUAL to the bound's value are only counted, per row (a row belongs to one wave: no atomics): a contact map with
    // plateaus -- probabilities out of a half-precision model, quantised maps -- has thousands of them, they would overflow
    // any buffer, and which of them belong to the answer is decided by position alone (step 4).
    const uint32_t tkey = ~bound;   // topk_key of the bound
    const float thr = __uint_as_float((tkey & 0x80000000u) ? (tkey & 0x7fffffffu) : ~tkey);
    // (only where the sample says the map has them: four or more of its 16 384 keys equal to the bound -- a map of real-valued
    //  probabilities has one, the bound itself, and keeps the cheaper test: one compare and one lane mask per register)
    uint32_t ties_s = 0;
#pragma unroll
    for (int q = 0; q < NS * W; ++q) ties_s += lane_votes(((s_valid >> q) & 1u) && sk[q] == bound);
    if (lane == 0) atomicAdd(&sh.cnt[kSelectCounters - 1], ties_s);
    for (int q = tid; q < kRowTies; q += TH) s_rowt[q] = 0;
    __syncthreads();
    const bool count_ties = !whole && L <= kRowTies && sh.cnt[kSelectCounters - 1] >= 4u;
    uint32_t* __restrict__ wk = s_key + wave * WCAP;
    uint32_t* __restrict__ wij = s_ij + wave * WCAP;
    uint32_t w_fill = 0;   // (wave-uniform)
    auto keep = [&](uint32_t x, bool take, int i, int j) {
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(take);
        const uint32_t at = w_fill + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        if (take && at < (uint32_t)WCAP) {
            wk[at] = x;
            wij[at] = ((uint32_t)i << 16) | (uint32_t)j;
        }
        w_fill += (uint32_t)__builtin_popcountll(mask);
    };
    auto offer = [&](float x, auto valid_of, int i, int j) {   // (`i` wave-uniform; validity only asked for what passes the value test)
        if (__builtin_amdgcn_ballot_w64(!(x < thr)) == 0) return;   // (two thirds of a map's registers: nobody in the wave)
        const bool valid = valid_of();
        if (count_ties) {
            const bool above = valid && !(x <= thr);
            if (__builtin_amdgcn_ballot_w64(above) != 0) keep(__float_as_uint(x), above, i, j);
            const unsigned long long eq = __builtin_amdgcn_ballot_w64(valid && x == thr);
            if (eq != 0 && lane == 0) s_rowt[i] = (uint16_t)(s_rowt[i] + (uint32_t)__builtin_popcountll(eq));
        } else {
            const bool take = valid && !(x < thr);
            if (__builtin_amdgcn_ballot_w64(take) != 0) keep(__float_as_uint(x), take, i, j);
        }
    };
#pragma unroll
    for (int u = 0; u < NS; ++u)
#pragma unroll
        for (int e = 0; e < W; ++e) offer(elem(sv[u], e), [&]() { return (bool)((s_valid >> (u * W + e)) & 1u); }, max(s_row[u], 0), s_c0[u] + W * lane + e);
    constexpr int G = VEC4 ? 8 : 12;
    if (!whole) {
        auto sampled = [&](const TopkSlots<VEC4, NW>& p) {
            bool hit = false;
#pragma unroll
            for (int u = 0; u < NS; ++u) hit |= p.n == s_pair[u] && p.sl == s_slot[u];
            return hit;
        };
        while (!it.done()) {
            LV v[G];
            int vi[G], vc[G];
#pragma unroll
            for (int u = 0; u < G; ++u) {   // the next G slots (scalar work only) ...
                while (!it.done() && sampled(it)) it.next();
                vi[u] = -1;
                vc[u] = 0;
                if (!it.done()) {
                    vi[u] = it.row();
                    vc[u] = it.col0();
                    it.next();
                }
            }
#pragma unroll
            for (int u = 0; u < G; ++u) v[u] = load(vi[u], vc[u]);   // ... their loads, all in flight together ...
#pragma unroll
            for (int u = 0; u < G; ++u) {                                      // ... and what they hold
                if (vi[u] < 0) continue;   // (wave-uniform)
#pragma unroll
                for (int e = 0; e < W; ++e) {
                    const int j = vc[u] + W * lane + e;
                    offer(elem(v[u], e), [&]() { return j >= vi[u] + 5 && j < L; }, vi[u], j);
                }
            }
        }
    }
    if (lane == 0) sh.wave_cnt[wave] = w_fill;
    __syncthreads();
    uint32_t n_cand = 0;
    bool over = false;
    for (int w = 0; w < NW; ++w) {
        n_cand += sh.wave_cnt[w];
        over |= sh.wave_cnt[w] > (uint32_t)WCAP;
    }
#if defined(DCTFP_TOPK1_STOP) && DCTFP_TOPK1_STOP == 2
    if (tid == 0) out_n[job.orig] = (int)n_cand;
    return;
#endif
    if (over || (!count_ties && n_cand < kk)) {   // more candidates than the buffer holds, or the samples misjudged a map whose
        if (tid == 0) out_n[job.orig] = -1;         // ties were not counted: the two-read kernel behind this one
        return;
    }
    // ---- 3. the candidates' keys; how many there really are (what passed the value test and is not below the bound by its KEY
    // -- a NaN of the wrong sign -- is no candidate)
    constexpr int E = WCAP / 64;   // (NW * WCAP candidate slots over NW * 64 threads)
    uint32_t ck[E], cij[E];
    bool have[E];
    uint32_t cmin = 0xffffffffu;
    const uint32_t key_hi = count_ties ? (bound == 0 ? 0u : bound - 1u) : bound;   // the largest key a candidate may have
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const uint32_t p = (uint32_t)(e * TH + tid);
        have[e] = (p % (uint32_t)WCAP) < sh.wave_cnt[p / (uint32_t)WCAP];
        const uint32_t xk = have[e] ? ikey(__uint_as_float(s_key[p])) : 0xffffffffu;
        have[e] = have[e] && xk <= key_hi && (!count_ties || bound != 0);
        ck[e] = have[e] ? xk : 0xffffffffu;
        cij[e] = have[e] ? s_ij[p] : 0xffffffffu;
        cmin = min(cmin, ck[e]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) cmin = min(cmin, (uint32_t)__shfl_xor((int)cmin, off));
    uint32_t n_real = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) n_real += lane_votes(have[e]);
    if (lane == 0) {
        atomicMin(&sh.row_min, cmin);
        atomicAdd(&sh.fill, n_real);
    }
    __syncthreads();
    cmin = sh.row_min;
    const uint32_t n_less = sh.fill;   // candidates (keys below the bound when ties are counted, up to it otherwise)
    int32_t* __restrict__ oi = out_i + job.out_off;
    int32_t* __restrict__ oj = out_j + job.out_off;
    float* __restrict__ ov = out_v + job.out_off;
    auto emit = [&](uint32_t pos, uint32_t key_e, uint32_t ij) {
        const int i = (int)(ij >> 16), j = (int)(ij & 0xffffu);
        oi[pos] = i;
        oj[pos] = j;
        // the value from its key (1 300 scattered 4-byte reads per map were a fifth of the kernel's HBM traffic);
        // only a zero is read back: -0.0 and +0.0 share a key and the caller gets the map's own bits
        const uint32_t tk = ~key_e;
        const float val = __uint_as_float((tk & 0x80000000u) ? (tk & 0x7fffffffu) : ~tk);
        ov[pos] = val == 0.0f ? job.map[(size_t)i * job.ld + j] : val;
    };
    if (n_less >= kk) {   // the kk smallest of the candidates, ties to the lowest (i, j)
        bisect_emit<E>(ck, [&](int e) { return cij[e]; }, [&](int e) { return have[e]; }, cmin, key_hi, 0u, 0xfffffffeu, kk, sh, step, emit);
        if (tid == 0) out_n[job.orig] = job.k;
        return;
    }
    if (!count_ties) {
        if (tid == 0) out_n[job.orig] = -1;
        return;
    }
    // ---- 4. fewer than kk keys below the bound: the answer is all of them + the first kk - n_less entries EQUAL to the bound in
    // (i, j) order -- a plateau.  Rows before r* give all their ties, row r* its first few: from the per-row counts; the rows
    // up to r* are streamed once more and every tie goes straight to its place (no buffer, no selection).
    const uint32_t need = kk - n_less;
    if (wave == 0) {   // exclusive prefix sums of the per-row counts, in place (uint16 wraps are harmless below: compared against need)
        uint32_t carry = 0;
        for (int r0 = 0; r0 < L; r0 += 64) {
            const int r = r0 + lane;
            const uint32_t c = r < L ? s_rowt[r] : 0u;
            uint32_t incl = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, off);
                incl += lane >= off ? up : 0u;
            }
            if (r < L) s_rowp[r] = carry + incl - c;
            carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        if (lane == 0) sh.bound = carry;   // all ties of the map
    }
    __syncthreads();
    if (sh.bound < need) {   // (the samples misjudged the map: even with its ties there are fewer than kk entries up to the bound)
        if (tid == 0) out_n[job.orig] = -1;
        return;
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {   // the candidates, wherever there is room among the first n_less places
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(have[e]);
        if (mask != 0) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&sh.pos, (uint32_t)__builtin_popcountll(mask));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (have[e]) emit(base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u)), ck[e], cij[e]);
        }
    }
    {
        TopkSlots<VEC4, NW> it2;
        it2.init(L, wave);
        int cur_row = -1;
        uint32_t row_run = 0;   // ties of the current row seen in its earlier chunks
        while (!it2.done()) {
            LV v[G];
            int vi[G], vc[G];
#pragma unroll
            for (int u = 0; u < G; ++u) {
                while (!it2.done() && s_rowp[it2.row()] >= need) it2.next();   // (rows past r*: none of their ties belongs to the answer)
                vi[u] = -1;
                vc[u] = 0;
                if (!it2.done()) {
                    vi[u] = it2.row();
                    vc[u] = it2.col0();
                    it2.next();
                }
            }
#pragma unroll
            for (int u = 0; u < G; ++u) v[u] = load(vi[u], vc[u]);
#pragma unroll
            for (int u = 0; u < G; ++u) {
                if (vi[u] < 0) continue;
                if (vi[u] != cur_row) {
                    cur_row = vi[u];
                    row_run = 0;
                }
                unsigned long long m[W];
                uint32_t total = 0;
#pragma unroll
                for (int e = 0; e < W; ++e) {
                    const int j = vc[u] + W * lane + e;
                    m[e] = __builtin_amdgcn_ballot_w64(j >= vi[u] + 5 && j < L && elem(v[u], e) == thr);
                    total += (uint32_t)__builtin_popcountll(m[e]);
                }
                if (total == 0) continue;
                const unsigned long long lower = (1ull << lane) - 1ull;
                uint32_t before_lane = 0;   // ties of this chunk in the lanes before mine (all their columns are smaller)
#pragma unroll
                for (int e = 0; e < W; ++e) before_lane += (uint32_t)__builtin_popcountll(m[e] & lower);
                uint32_t mine = 0;
#pragma unroll
                for (int e = 0; e < W; ++e) {
                    if ((m[e] >> lane) & 1ull) {
                        const uint32_t at = s_rowp[vi[u]] + row_run + before_lane + mine;   // place among the map's ties in (i, j) order
                        if (at < need) {
                            const int j = vc[u] + W * lane + e;
                            oi[n_less + at] = vi[u];
                            oj[n_less + at] = j;
                            ov[n_less + at] = elem(v[u], e);
                        }
                        ++mine;
                    }
                }
                row_run += total;
            }
        }
    }
    if (tid == 0) out_n[job.orig] = job.k;
}

// NW = 8: maps up to k = 1 280 (L <= 492 at t = 2.6 ... see the host) in workgroups of 512 threads, 37 KB of LDS: FOUR per CU, so
// that three stream while one samples or selects (two workgroups of 1024 left the memory pipe idle a third of the time);
// NW = 16: up to k = 3 000.  A job outside a build's range is left to the other (each launch sees every job).
template <int NW>
__global__ __launch_bounds__(NW * 64, DCTFP_TOPK_WAVES) void contact_topk1_kernel(const TopkJob* __restrict__ jobs, int32_t* __restrict__ out_i,
                                                              int32_t* __restrict__ out_j, float* __restrict__ out_v,
                                                              int32_t* __restrict__ out_n, int k_from, int k_to) {
    __shared__ SelectSharedT<NW> sh;
    __shared__ uint32_t s_key[NW * 512], s_ij[NW * 512];
    __shared__ uint16_t s_rowt[NW == 8 ? 576 : 1216];
    const TopkJob job = jobs[blockIdx.x];
    if (job.k > 0 && job.n_res >= 6 && (job.k < k_from || job.k > k_to)) return;
    for (int q = threadIdx.x; q < kSelectCounters; q += NW * 64) sh.cnt[q] = 0;
    if (threadIdx.x == 0) {
        sh.pos = 0;
        sh.fill = 0;
        sh.row_min = 0xffffffffu;
    }
    __syncthreads();
    if (job.k <= 0 || job.n_res < 6) {
        if (threadIdx.x == 0) out_n[job.orig] = 0;
        return;
    }
    if (job.k > 3000 || job.n_res > 65535 || (int64_t)job.n_res * job.ld >= ((int64_t)1 << 29)) {   // (one and a half times k candidates must
        // fit the LDS / (i, j) beyond 16 bits each / a map beyond the 32-bit byte offsets of the buffer loads)
        if (threadIdx.x == 0) out_n[job.orig] = -1;
        return;
    }
    if ((reinterpret_cast<uintptr_t>(job.map) & 15u) == 0 && (job.ld & 3) == 0) topk1_run<true, NW>(job, sh, s_key, s_ij, s_rowt, s_ij, out_i, out_j, out_v, out_n);
    else topk1_run<false, NW>(job, sh, s_key, s_ij, s_rowt, s_ij, out_i, out_j, out_v, out_n);
}

}  // namespace dctfp
