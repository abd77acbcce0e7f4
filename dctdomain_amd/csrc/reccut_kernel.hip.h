// recursiveMaxCut of the reference's domain cutter (mgtools/DCTdomain src/RecCut.cpp:150-351) on the GPU: one workgroup per
// protein, the contact graph as sparse adjacency lists, the recursion driven from a small stack in device memory.
//
// The host library (csrc/reccut.cpp, libreccut.so) stays the definition of what must come out; this kernel computes the same
// integers and the same double expressions, so that the same cuts are chosen:
//   * graph weights (int)(strtod("%.6f" % p) * 100 + 0.5), |i - j| <= 3 forced to 100 (readGraph, :354-397) -- the text round
//     trip reproduced in exact arithmetic (contact_weight_exact below);
//   * pre / post weights of every vertex in the CURRENT vertex order, n1 / n2 / cutv as running sums with the reference's quirk
//     that n2[0] = sum is never reduced by vertex 0's edges (:178-183);
//   * single cut: ave = (double)cutv * sum / n1[i] / n2[i], first minimum below 2.0 (:186-201);
//   * double cut over i in [10, V - 10), j in [i + 21, V - 10): ave = (double)cv * sum / ns1 / ns2 with ns2 = the weight inside the
//     vertex range [i, j], first minimum in ascending (i, j) order (:243-260);
//   * accept / recurse rules with 0.08 / 0.07 and Min_Size 22 (:263-350), the segment / cut-site bookkeeping of SplitDomain and
//     SplitDomain_2cuts (:16-148) statement for statement as in reccut.cpp (its quirks decide the printed strings).
//
// What is different is how the O(V^2) double-cut scan is done.  inner[i][j] (weight inside [i, j]) obeys
//   inner[i][j] = inner[i+1][j] + sum_{b in (i, j]} a[i][b],
// and a contact graph has ~ 11 edges per vertex: a lane owns a COLUMN j and walks the rows i downwards; what row i adds to its
// column is the sum over i's forward neighbours q <= j -- a list of ~ 6 entries, the same for all lanes of the wave (LDS
// broadcast) -- no V x V table, no prefix scan across lanes, no barrier inside the scan.  Every candidate pair first meets a
// single-precision test against the lane's best so far (all integers below 2^23 are exact in float; margin 1e-4); the
// reference's own double expression (two divisions) is evaluated only where the test cannot rule the pair out.
//
// A protein this kernel does not take -- more residues or edges than its LDS class holds, more segments per domain or pending
// nodes than its tables, a contact outside the protein, a non-finite probability, or a step where the reference would index
// outside its segment table (undefined behaviour there) -- gets status -1: the caller runs libreccut.so on it.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dctfp {

constexpr int kCutMinTerminal = 10;  // src/RecCut.cpp:10
constexpr int kCutMinSize = 22;      // src/RecCut.cpp:14
constexpr int kCutMaxSeg = 24;       // segments (and cut sites) a node's tables hold
constexpr int kCutStack = 24;        // pending nodes
constexpr int kCutNodeInts = 4 + 2 * kCutMaxSeg + kCutMaxSeg;   // {first, V, n_seg, n_site, segs, sites}

struct CutJob {
    const int32_t* ci;   // contacts of this protein (device)
    const int32_t* cj;
    const float* cv;
    uint32_t* adj;       // scratch: 2 * n_contacts + 6 * n_res entries (neighbour | weight << 16)
    int32_t* stack;      // scratch: kCutStack * kCutNodeInts
    int32_t* out;        // {n_domains or -1, then per domain: n_segs, (first, last) ...}, 0-based residues
    unsigned long long* timing;   // instrumented build (-DDCTFP_CUT_TIMING): 11 phase totals of the whole launch (100 MHz ticks), else unused
    int32_t n_contacts;
    int32_t n_res;
    int32_t out_cap;
    int32_t reserved;
};

// The integer edge weight of a contact probability as the reference's binary reads it from the .ce file: the float32 printed with
// "%.6f" (src/fingerprint.py:72), parsed back (strtod) and turned into (int)(v * 100 + 0.5) (src/RecCut.cpp:384) -- in exact
// arithmetic: p * 10^6 is exact in double (24 + 20 bits), printf rounds that exact value half-to-even (rint), strtod returns the
// double nearest to n / 10^6 (a correctly rounded division).  Same function on the host (reccut.cpp checks it against the
// snprintf / strtod round trip).
__host__ __device__ inline int contact_weight_exact(float p) {
    const double n6 = rint((double)p * 1.0e6);
    const double v = n6 / 1.0e6;
#ifdef __HIP_DEVICE_COMPILE__
    return (int)__dadd_rn(__dmul_rn(v, 100.0), 0.5);   // (no fused multiply-add: the reference rounds the product)
#else
    volatile double prod = v * 100.0;
    return (int)(prod + 0.5);
#endif
}

// Inclusive prefix sums of v[0 .. n) in place (n <= capacity of the LDS arrays), by the whole workgroup; `carry` = one LDS int.
template <int kCutThreads>
__device__ inline void block_scan_inclusive(int32_t* __restrict__ v, int n, int32_t* __restrict__ wave_tot) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = kCutThreads / 64;
    int base = 0;
    for (int c0 = 0; c0 < n; c0 += kCutThreads) {   // (uniform trip count)
        const int p = c0 + tid;
        int x = p < n ? v[p] : 0;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(x, off);
            x += lane >= off ? up : 0;
        }
        if (lane == 63) wave_tot[wave] = x;
        __syncthreads();
        int before = base;
        for (int w = 0; w < wave; ++w) before += wave_tot[w];
        int total = 0;
        for (int w = 0; w < NW; ++w) total += wave_tot[w];
        if (p < n) v[p] = x + before;
        base += total;
        __syncthreads();
    }
}

// (ave, i, j) in the reference's order of preference: smaller ave, then smaller i, then smaller j
__device__ inline bool cut_better(double a, int i, int j, double b, int bi, int bj) {
    return a < b || (a == b && (i < bi || (i == bi && j < bj)));
}

// LDS (ints), CAP = residues of the class, ECAP = forward edges of a node:
//   idx[CAP] tmp[CAP] pos[CAP] off[CAP + 1] foff[CAP + 1] A[CAP + 1] B[CAP + 2] fwd[ECAP] + node tables
// kCutThreads: 512 for the 512-residue class, 1024 above -- a wave per 64 columns of the top node, so that a protein's scan is one
// pass of all its waves (with 256 threads a 1 300-residue protein took 5.7 ms on its own: the latency of one workgroup)
template <int CAP, int ECAP, int kCutThreads, int RB, int KMAX>
__global__ __launch_bounds__(kCutThreads, 4 /* waves per SIMD: two 512-thread workgroups per CU, or one of 1 024 */) void reccut_kernel(const CutJob* __restrict__ jobs, double cut1, double cut2) {
    extern __shared__ int32_t lds[];
    int32_t* __restrict__ idx = lds;
    int32_t* __restrict__ tmp = idx + CAP;
    int32_t* __restrict__ pos = tmp + CAP;
    int32_t* __restrict__ off = pos + CAP;          // residue-space adjacency offsets
    int32_t* __restrict__ foff = off + CAP + 1;     // position-space forward lists
    int32_t* __restrict__ A = foff + CAP + 1;       // pre -> n1 -> t
    int32_t* __restrict__ B = A + CAP + 1;          // post -> n2 (V + 1 entries) -> c
    uint32_t* __restrict__ fwd = reinterpret_cast<uint32_t*>(B + CAP + 2);
    int32_t* __restrict__ node = reinterpret_cast<int32_t*>(fwd + ECAP);   // current node: {first, V, n_seg, n_site, segs[2 MAXSEG], sites[MAXSEG]}
    int32_t* __restrict__ kid = node + kCutNodeInts;                       // two children being built
    int32_t* __restrict__ wtot = kid + 2 * kCutNodeInts;                   // 16 wave totals of the scans
    int32_t* __restrict__ U = wtot + 32;                                   // [RB][CAP / 64 + 1]: weight of a row's edges before each column group
    uint8_t* __restrict__ M = reinterpret_cast<uint8_t*>(U + RB * (CAP / 64 + 1));   // [RB][CAP]: forward weights of a block of rows, dense
    // what the bands of the scan start from, [KMAX - 1][CAP]: band 1's vector lives in `pos` (nobody reads pos between the forward
    // lists and the end of the node, where it is reset -- wholesale then), the others behind the tile
    int32_t* __restrict__ INITX = reinterpret_cast<int32_t*>(M + RB * CAP);
    auto INIT = [&](int b) -> int32_t* { return b == 0 ? pos : INITX + (b - 1) * CAP; };   // b = band - 1
    int32_t* __restrict__ WT = INITX + (KMAX > 2 ? KMAX - 2 : 0) * CAP;              // [KMAX - 1][waves]: wave totals of the scan over the vectors
    constexpr int RBS = RB / KMAX;                                                    // rows of a band per iteration (the tile holds RBS rows of each band)
    // tasks (band, column group) per wave: band b holds the groups from its lowest row's on -- K NG / 2 + NG / 2 + K pairs at most
    constexpr int SLOTS = (KMAX * (CAP / 64) / 2 + (CAP / 64) / 2 + 2 * KMAX + kCutThreads / 64 - 1) / (kCutThreads / 64);
    static_assert(RB % KMAX == 0 && RBS % 4 == 0 && RBS * KMAX * 1 <= kCutThreads, "tile rows: RBS rows of each of KMAX bands");
    int32_t* __restrict__ misc = wtot + 16;                                // [4] status, [5] action, [6] cuts, [7] cuts1, [8] cuts2, [9] out cursor,
                                                                           // [10] n_domains, [11] stack depth
    __shared__ double red_ave[kCutThreads / 64];
    __shared__ int red_i[kCutThreads / 64], red_j[kCutThreads / 64];

    const CutJob job = jobs[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = job.n_res;
    int32_t* __restrict__ out = job.out;
#ifdef DCTFP_CUT_TIMING
    unsigned long long tl_prev = __builtin_amdgcn_s_memrealtime(), tl_acc[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define CUT_T(k)                                                          \
    do {                                                                  \
        const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); \
        tl_acc[k] += now_ - tl_prev;                                      \
        tl_prev = now_;                                                   \
    } while (0)
#define CUT_T_OUT()                                                                     \
    do {                                                                                \
        if (tid == 0 && job.timing)                                                     \
            for (int q_ = 0; q_ < 11; ++q_) atomicAdd(job.timing + q_, tl_acc[q_]);      \
    } while (0)
#else
#define CUT_T(k)
#define CUT_T_OUT()
#endif
    if (L < kCutMinSize) {   // "Protein has length of 0" never happens from the caller: a short protein is one domain (:446-447)
        if (tid == 0 && job.out_cap >= 4) {
            out[0] = 1;
            out[1] = 1;
            out[2] = 0;
            out[3] = L - 1;
        } else if (tid == 0 && job.out_cap >= 1) {
            out[0] = -1;
        }
        return;
    }
    if (L > CAP || (int64_t)job.n_contacts + 3ll * L > ECAP || job.out_cap < 4) {
        if (tid == 0 && job.out_cap >= 1) out[0] = -1;
        return;
    }
    if (tid < 16) misc[tid] = 0;
    for (int q = tid; q < RB * CAP / 4; q += kCutThreads) reinterpret_cast<uint32_t*>(M)[q] = 0;
    for (int q = tid; q < RB * (CAP / 64 + 1); q += kCutThreads) U[q] = 0;
    __syncthreads();

    // ---- the graph: residue-space adjacency (both directions) in the job's scratch
    for (int r = tid; r < L; r += kCutThreads) {
        int d = 0;
        for (int s = 1; s <= 3; ++s) d += (r - s >= 0 ? 1 : 0) + (r + s < L ? 1 : 0);
        tmp[r] = d;
        idx[r] = r;
        pos[r] = -1;
    }
    __syncthreads();
    for (int c = tid; c < job.n_contacts; c += kCutThreads) {
        const int i = job.ci[c], j = job.cj[c];
        const float p = job.cv[c];
        if (i < 0 || j < 0 || i >= L || j >= L || !(fabsf(p) < 1.0e6f)) {
            misc[4] = 1;   // a contact outside the protein / a value the text round trip does not define: the host library reports it
            continue;
        }
        const int d = i > j ? i - j : j - i;
        if (d == 0 || d <= 3) continue;   // (the band overwrites these; the diagonal carries no edge)
        if (contact_weight_exact(p) == 0) continue;
        atomicAdd(&tmp[i], 1);
        atomicAdd(&tmp[j], 1);
    }
    __syncthreads();
    for (int r = tid; r < L; r += kCutThreads) A[r] = tmp[r];
    __syncthreads();
    block_scan_inclusive<kCutThreads>(A, L, wtot);
    for (int r = tid; r < L; r += kCutThreads) {
        const int begin = A[r] - tmp[r];
        off[r] = begin;
        if (r == L - 1) off[L] = A[r];
        int at = begin;   // the band first (this thread alone writes residue r's list so far)
        for (int s = 1; s <= 3; ++s) {
            if (r - s >= 0) job.adj[at++] = (uint32_t)(r - s) | (100u << 16);
            if (r + s < L) job.adj[at++] = (uint32_t)(r + s) | (100u << 16);
        }
        B[r] = at;   // cursor
    }
    __syncthreads();
    if (misc[4] == 0)
        for (int c = tid; c < job.n_contacts; c += kCutThreads) {
            const int i = job.ci[c], j = job.cj[c];
            const int d = i > j ? i - j : j - i;
            if (d <= 3) continue;
            const int w = contact_weight_exact(job.cv[c]);
            if (w == 0) continue;
            if (w < 0 || w > 255) {
                misc[4] = 1;   // (the scan keeps a block of rows as bytes: other weights go to the host library)
                continue;
            }
            job.adj[atomicAdd(&B[i], 1)] = (uint32_t)j | ((uint32_t)w << 16);
            job.adj[atomicAdd(&B[j], 1)] = (uint32_t)i | ((uint32_t)w << 16);
        }
    __threadfence_block();
    __syncthreads();
    if (misc[4] != 0) {
        if (tid == 0) out[0] = -1;
        return;
    }
    CUT_T(0);   // graph
    if (tid == 0) {
        node[0] = 0;
        node[1] = L;
        node[2] = 1;
        node[3] = 0;
        node[4] = 0;
        node[5] = L - 1;
        misc[9] = 1;    // out cursor
        misc[10] = 0;   // domains
        misc[11] = 0;   // stack depth
    }
    __syncthreads();

    for (;;) {   // one node of the recursion per turn; every branch below is workgroup-uniform
        const int first = node[0], V = node[1];
        int32_t* __restrict__ ix = idx + first;
        for (int p = tid; p < V; p += kCutThreads) pos[ix[p]] = p;
        __syncthreads();
        // ---- pre / post weight of every vertex in the current order; forward degree
        for (int p = tid; p < V; p += kCutThreads) {
            const int r = ix[p];
            int pre = 0, post = 0, fd = 0;
            for (int e = off[r]; e < off[r + 1]; ++e) {
                const uint32_t a = job.adj[e];
                const int q = pos[a & 0xffffu];
                if (q >= 0) {
                    const int w = (int)(a >> 16);
                    if (q < p) pre += w;
                    else {
                        post += w;
                        ++fd;
                    }
                }
            }
            A[p] = pre;
            B[p] = post;
            foff[p] = fd;
            tmp[p] = fd;
        }
        __syncthreads();
        block_scan_inclusive<kCutThreads>(foff, V, wtot);
        const int n_fwd = foff[V - 1];
        if (n_fwd > ECAP) {   // (cannot happen: n_fwd <= n_contacts + 3 L)
            if (tid == 0) out[0] = -1;
            return;
        }
        for (int p = tid; p < V; p += kCutThreads) {
            const int r = ix[p];
            int at = foff[p] - tmp[p];
            for (int e = off[r]; e < off[r + 1]; ++e) {
                const uint32_t a = job.adj[e];
                const int q = pos[a & 0xffffu];
                if (q > p) fwd[at++] = (uint32_t)q | (a & 0xffff0000u);
            }
        }
        __syncthreads();
        for (int p = tid; p < V; p += kCutThreads) foff[p] = foff[p] - tmp[p];   // -> exclusive (start of the list); tmp keeps the length
        CUT_T(1);   // pre / post / forward lists
        // sum = all post weights; pre0 / post0 are needed after the scans
        const int post0 = B[0];
        __syncthreads();
        if (tid == 0) {
            A[0] = 0;   // n1[0] = 0: vertex 0 has no pre weight anyway
            B[0] = 0;   // ... and its post weight is not subtracted from n2 (the reference's quirk): handled through post0 below
        }
        __syncthreads();
        block_scan_inclusive<kCutThreads>(A, V, wtot);   // A[i] = n1[i]
        block_scan_inclusive<kCutThreads>(B, V, wtot);   // B[i] = sum_{1 <= x <= i} post[x]
        const int sum = B[V - 1] + post0;
        // single cut: cutv[i] = post0 + sum_{1..i} (post - pre); ave over 1 <= i < V - 2
        double b1 = 2.0;
        int b1i = 0x7fffffff;
        for (int i = tid; i < V; i += kCutThreads) {
            const int n1 = A[i], spost = B[i];
            const int n2 = sum - spost;
            if (i >= 1 && i < V - 2) {
                const int cutv = post0 + spost - n1;
                const double ave = ((double)cutv) * sum / n1 / n2;
                if (ave < 2.0 && cut_better(ave, i, 0, b1, b1i, 0)) {
                    b1 = ave;
                    b1i = i;
                }
            }
        }
        __syncthreads();
        // B -> n2 (V + 1 entries), then A -> t[j] = n1[j] - n2[j + 1] and B -> c[i] = n2[i] - n1[i - 1]
        for (int i = tid; i < V; i += kCutThreads) B[i] = sum - B[i];
        if (tid == 0) B[V] = 0;
        __syncthreads();
        int tj_keep[(CAP + kCutThreads - 1) / kCutThreads], ci_keep[(CAP + kCutThreads - 1) / kCutThreads];
#pragma unroll
        for (int m = 0; m < (CAP + kCutThreads - 1) / kCutThreads; ++m) {
            const int i = m * kCutThreads + tid;
            tj_keep[m] = i < V ? A[i] - B[i + 1] : 0;
            ci_keep[m] = i >= 1 && i < V ? B[i] - A[i - 1] : 0;
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < (CAP + kCutThreads - 1) / kCutThreads; ++m) {
            const int i = m * kCutThreads + tid;
            if (i < V) {
                A[i] = tj_keep[m];
                B[i] = ci_keep[m];
            }
        }
        // reduce the single cut
        {
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                const double oa = __shfl_xor(b1, o);
                const int oi = __shfl_xor(b1i, o);
                if (cut_better(oa, oi, 0, b1, b1i, 0)) {
                    b1 = oa;
                    b1i = oi;
                }
            }
            if (lane == 0) {
                red_ave[wave] = b1;
                red_i[wave] = b1i;
            }
        }
        __syncthreads();
        double best1 = 2.0;
        int cuts = 0;
        {
            int bi = 0x7fffffff;
            for (int w = 0; w < kCutThreads / 64; ++w)
                if (cut_better(red_ave[w], red_i[w], 0, best1, bi, 0)) {
                    best1 = red_ave[w];
                    bi = red_i[w];
                }
            if (bi != 0x7fffffff) cuts = bi + 1;
        }
        __syncthreads();
        CUT_T(2);   // scans + single cut
        // ---- the double-cut scan: a lane per column j in [31, V - 10), rows downwards in blocks of RB.
        // For a block the forward weights go into a dense byte tile M[RB][CAP] in LDS (M[t][q] = weight of the edge from row hi - t
        // to position q > row), and per row the weight that lies before each 64-column group into U[t][g].  What row a adds to
        // column j of group g -- the sum over a's forward neighbours q <= j -- is then U[t][g] + the inclusive scan of the tile's
        // 64 bytes across the lanes (six DPP additions): no list walk, no branch per edge.  A loop over the edge lists with a
        // compare per lane and entry was 140 us per node, 90 % of the kernel (tools/cut_timing_probe.py) -- a dependent chain of
        // ~ 150 instructions per row on a wave alone on its SIMD.
        double b2 = 2.0;
        int b2i = 0x7fffffff, b2j = 0x7fffffff;
        const bool small = sum > 0 && sum < (1 << 23);
        const float sum_f = (float)sum;
        const int j_end = V - kCutMinTerminal;   // candidates: j < j_end
        constexpr int NG = CAP / 64, NWV = kCutThreads / 64;
        // ---- the scan in row BANDS (round 5).  Through the first version a wave owned a 64-column group and walked all of its rows
        // one after the other: the wave of the last group had V rows to go, the wave of the first 40, and a node took as long as
        // its longest wave (tools/cut_timing_probe.py: wave 0 waited for the slowest one for 2.4 x its own scan; on 250-residue
        // nodes three of eight waves had no group at all).  The rows [10, j_end - 2] are now cut into K bands of `band_rows`; a
        // task = (band, column group), active where the band has a row below the group's last column; the tasks go round the
        // waves, so every wave walks about the same number of rows whatever V is.  What a band needs to start from -- for column
        // j the weight inside [top + 1, j], top = the row above the band -- comes from the forward lists: INIT[b][q] collects
        // the weight of the edges (a, q) with a above band b's top (a row's edges go to the first band below it, LDS atomics;
        // then a running sum over the bands), and an inclusive scan over q makes it the weight inside [top_b + 1, j].
        const int hi_top = j_end - 2;
        const int n_scan_rows = hi_top - kCutMinTerminal + 1;
        int K = n_scan_rows / 16;
        K = K < 1 ? 1 : (K > KMAX ? KMAX : K);
        int band_rows = n_scan_rows > 0 ? (n_scan_rows + K - 1) / K : RBS;
        band_rows = (band_rows + RBS - 1) / RBS * RBS;
        const int n_iter = n_scan_rows > 0 ? band_rows / RBS : 0;
        const int NGa = (j_end + 63) >> 6;   // column groups with a candidate column
        auto band_top = [&](int b) { return hi_top - b * band_rows; };
        auto band_low = [&](int b) { return max(kCutMinTerminal, hi_top - (b + 1) * band_rows + 1); };
        if (KMAX > 1 && K > 1) {
            // INIT[b - 1][q], b = 1 .. K - 1
            for (int b = 0; b < K - 1; ++b)
                for (int q = tid; q < V; q += kCutThreads) INIT(b)[q] = 0;
            __syncthreads();
            for (int a = tid; a <= hi_top; a += kCutThreads) {
                if (a < kCutMinTerminal + 1) continue;                       // (inside [top + 1, j] means a >= top + 1 >= 11)
                const int bb = (hi_top - a) / band_rows + 1;                  // the first band whose top lies below row a
                if (bb >= K) continue;
                const int e1 = foff[a] + tmp[a];
                for (int e = foff[a]; e < e1; ++e) {
                    const uint32_t ent = fwd[e];
                    atomicAdd(&INIT(bb - 1)[(int)(ent & 0xffffu)], (int)(ent >> 16));
                }
            }
            __syncthreads();
            // running sum over the bands (an edge above band bb is above every later band), then the scan over q of all bands at once
            {
                constexpr int PER = (CAP + kCutThreads - 1) / kCutThreads;
                int carry[KMAX > 1 ? KMAX - 1 : 1];
#pragma unroll
                for (int b = 0; b < KMAX - 1; ++b) carry[b] = 0;
#pragma unroll
                for (int m = 0; m < PER; ++m) {
                    const int q = m * kCutThreads + tid;
                    int x[KMAX > 1 ? KMAX - 1 : 1];
                    int run = 0;
#pragma unroll
                    for (int b = 0; b < KMAX - 1; ++b) {
                        run += (b < K - 1 && q < V) ? INIT(b)[q] : 0;
                        x[b] = run;
                    }
#pragma unroll
                    for (int off2 = 1; off2 < 64; off2 <<= 1) {
#pragma unroll
                        for (int b = 0; b < KMAX - 1; ++b) {
                            const int up = __shfl_up(x[b], off2);
                            x[b] += lane >= off2 ? up : 0;
                        }
                    }
                    if (lane == 63) {
#pragma unroll
                        for (int b = 0; b < KMAX - 1; ++b) WT[b * NWV + wave] = x[b];
                    }
                    __syncthreads();
#pragma unroll
                    for (int b = 0; b < KMAX - 1; ++b) {
                        int before = carry[b], total = 0;
                        for (int w = 0; w < NWV; ++w) {
                            const int t_w = WT[b * NWV + w];
                            before += w < wave ? t_w : 0;
                            total += t_w;
                        }
                        if (b < K - 1 && q < V) INIT(b)[q] = x[b] + before;
                        carry[b] += total;
                    }
                    __syncthreads();
                }
            }
        }
        // tasks: band b holds the groups g_lo(b) .. NGa - 1; task tau -> (b, g) by the running counts
        int n_tasks = 0;
        int cum[KMAX + 1];
#pragma unroll
        for (int b = 0; b < KMAX; ++b) {
            cum[b] = n_tasks;
            if (b < K && n_scan_rows > 0 && band_top(b) >= kCutMinTerminal) {
                const int g_lo = (band_low(b) + 1) >> 6;
                n_tasks += max(0, NGa - g_lo);
            }
        }
        cum[KMAX] = n_tasks;
        int task_b[SLOTS], task_g[SLOTS], below[SLOTS];
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            const int tau = wave + sl * NWV;
            int b = -1, g = 0;
            if (tau < n_tasks) {
#pragma unroll
                for (int bb = 0; bb < KMAX; ++bb)
                    if (tau >= cum[bb] && tau < cum[bb + 1]) {
                        b = bb;
                        g = ((band_low(bb) + 1) >> 6) + tau - cum[bb];
                    }
            }
            task_b[sl] = b;
            task_g[sl] = g;
            const int j = g * 64 + lane;
            below[sl] = (b >= 1 && j < j_end) ? INIT(b - 1)[j] : 0;
        }
        if (n_tasks > SLOTS * NWV) {   // (cannot happen: SLOTS is sized for the largest node of the class)
            if (tid == 0) out[0] = -1;
            return;
        }
        float best_f = 2.0f * 1.0001f;
        for (int it = 0; it < n_iter; ++it) {
            // Tile row (b, t) = row band_top(b) - it RBS - t, filled by FPR threads (entry e of the row by thread e mod FPR: the byte
            // into M, its weight added to the row's per-group total U[.][group]) and emptied again by the same threads after the
            // scan -- tile and totals are all zero between iterations and between nodes, nothing is cleared wholesale.
            constexpr int FPR = kCutThreads / (KMAX * RBS) >= 8 ? 8 : kCutThreads / (KMAX * RBS);
            const int fill_t = tid / FPR, fill_b = fill_t / RBS;
            int fill_row = -1;
            if (fill_b < K) {
                const int r = band_top(fill_b) - it * RBS - (fill_t % RBS);
                if (r >= band_low(fill_b)) fill_row = r;
            }
            if (fill_row >= 0) {
                const int e1 = foff[fill_row] + tmp[fill_row];
                for (int e = foff[fill_row] + tid % FPR; e < e1; e += FPR) {
                    const uint32_t ent = fwd[e];
                    const int q = (int)(ent & 0xffffu);
                    M[fill_t * CAP + q] = (uint8_t)(ent >> 16);
                    atomicAdd(&U[fill_t * (NG + 1) + (q >> 6)], (int)(ent >> 16));
                }
            }
            __syncthreads();
            CUT_T(3);    // tile fill + its barrier
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) {
                const int b = task_b[sl], g = task_g[sl];
                if (b < 0) continue;
                const int hi = band_top(b) - it * RBS;                       // rows hi, hi - 1, ... (t = 0 ..) of this band
                const int n_rows = min(RBS, hi - band_low(b) + 1);
                const int g0 = g * 64;
                if (n_rows <= 0 || g0 + 63 <= hi - n_rows + 1) continue;     // (band done / all columns of the group at or before every row: nothing to add)
                const int j = g0 + lane;
                const bool col_ok = j < j_end;
                const int tj = col_ok ? A[j] : 0;
                const int tr0 = b * RBS;                                     // my band's rows of the tile
                int m_u = 0;   // lane t: weight of row hi - t's edges before this group's first column
                if (lane < n_rows)
                    for (int k = 0; k < g; ++k) m_u += U[(tr0 + lane) * (NG + 1) + k];
                const int m_c = lane < n_rows ? B[hi - lane] : 0;
                int bl = below[sl];
                static_assert(RBS % 4 == 0, "rows go through four at a time");
                for (int t = 0; t < n_rows; t += 4) {   // four rows at a time: their scans and tests are independent chains
                    int x[4], ns2[4], cv[4], ns1[4];
                    bool maybe[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) x[k] = (int)M[(tr0 + t + k) * CAP + j];   // (rows past the band's last: zero)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        // inclusive scan over the wave's 64 lanes: four steps inside the 16-lane rows, two across them
                        x[k] += __builtin_amdgcn_update_dpp(0, x[k], 0x111, 0xf, 0xf, true);
                        x[k] += __builtin_amdgcn_update_dpp(0, x[k], 0x112, 0xf, 0xf, true);
                        x[k] += __builtin_amdgcn_update_dpp(0, x[k], 0x114, 0xf, 0xf, true);
                        x[k] += __builtin_amdgcn_update_dpp(0, x[k], 0x118, 0xf, 0xf, true);
                        x[k] += __builtin_amdgcn_update_dpp(0, x[k], 0x142, 0xa, 0xf, false);
                        x[k] += __builtin_amdgcn_update_dpp(0, x[k], 0x143, 0xc, 0xf, false);
                    }
                    bool any = false;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        bl += x[k] + __builtin_amdgcn_readlane(m_u, min(t + k, 63));
                        // The candidate (i, j): first a single-precision test against the best score ANY lane of this wave has found
                        // so far; the reference's expression (two float64 divisions) only for what the test cannot rule out -- a
                        // pair that ties the best passes (margin 1e-4 against 3e-7 of rounding).
                        const int i = hi - t - k;
                        ns2[k] = bl;
                        cv[k] = tj + __builtin_amdgcn_readlane(m_c, min(t + k, 63)) - 2 * bl;
                        ns1[k] = sum - cv[k] - bl;
                        maybe[k] = t + k < n_rows && col_ok && j - i >= kCutMinSize - 1 && ns1[k] > 0 && ns2[k] > 0;
                        if (small) maybe[k] = maybe[k] && (float)cv[k] * sum_f <= best_f * (float)ns1[k] * (float)ns2[k];
                        any |= maybe[k];
                    }
                    if (__builtin_amdgcn_ballot_w64(any) != 0) {   // (wave-uniform)
                        bool improved = false;
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (maybe[k]) {
                                const double ave = ((double)cv[k]) * sum / ns1[k] / ns2[k];
                                if (ave < 2.0 && cut_better(ave, hi - t - k, j, b2, b2i, b2j)) {
                                    b2 = ave;
                                    b2i = hi - t - k;
                                    b2j = j;
                                    improved = true;
                                }
                            }
                        if (__builtin_amdgcn_ballot_w64(improved) != 0) {
                            double wmin = b2;
#pragma unroll
                            for (int o = 32; o >= 1; o >>= 1) wmin = fmin(wmin, __shfl_xor(wmin, o));
                            best_f = (float)(wmin * 1.0001);
                        }
                    }
                }
                below[sl] = bl;
            }
            CUT_T(8);    // this wave's rows
            __syncthreads();   // (every wave has read the tile)
            CUT_T(9);    // the wait for the slowest wave of the block
            if (fill_row >= 0) {
                const int e1 = foff[fill_row] + tmp[fill_row];
                for (int e = foff[fill_row] + tid % FPR; e < e1; e += FPR) {
                    const int q = (int)(fwd[e] & 0xffffu);
                    M[fill_t * CAP + q] = 0;
                    U[fill_t * (NG + 1) + (q >> 6)] = 0;
                }
            }
            // (the next iteration's fill by the same threads follows without a barrier only for their own entries: a barrier for the rest)
            __syncthreads();
            CUT_T(10);   // tile cleared
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const double oa = __shfl_xor(b2, o);
            const int oi = __shfl_xor(b2i, o), oj = __shfl_xor(b2j, o);
            if (cut_better(oa, oi, oj, b2, b2i, b2j)) {
                b2 = oa;
                b2i = oi;
                b2j = oj;
            }
        }
        if (lane == 0) {
            red_ave[wave] = b2;
            red_i[wave] = b2i;
            red_j[wave] = b2j;
        }
        __syncthreads();
        CUT_T(3);   // (instrumented build: 3 = tile fills, 8 = wave 0's rows, 9 = its wait for the block's slowest wave, 10 = tile clears)
        // ---- decision and bookkeeping: thread 0 (segments and cut sites are a handful of ints)
        if (tid == 0) {
            double best2 = 2.0;
            int cuts1 = 0x7fffffff, cuts2 = 0x7fffffff;
            for (int w = 0; w < kCutThreads / 64; ++w)
                if (cut_better(red_ave[w], red_i[w], red_j[w], best2, cuts1, cuts2)) {
                    best2 = red_ave[w];
                    cuts1 = red_i[w];
                    cuts2 = red_j[w];
                }
            if (cuts1 == 0x7fffffff) cuts1 = cuts2 = 0;
            const int n_seg = node[2], n_site = node[3];
            const int32_t* __restrict__ segs = node + 4;
            const int32_t* __restrict__ sites = node + 4 + 2 * kCutMaxSeg;
            int action = 0;   // 0 = this node is a domain, 1 = single cut, 2 = double cut, -1 = not for this kernel
            int32_t* __restrict__ k1 = kid;
            int32_t* __restrict__ k2 = kid + kCutNodeInts;
            int n1s = 0, n1c = 0, n2s = 0, n2c = 0;   // segments / cut sites of the two children
            bool bad = false;
            auto seg_ok = [&](int c) { return c >= 0 && c < n_seg; };
            auto push_seg = [&](int32_t* __restrict__ k, int& n, int a, int b) {
                if (n >= kCutMaxSeg) {
                    bad = true;
                    return;
                }
                k[4 + 2 * n] = a;
                k[4 + 2 * n + 1] = b;
                ++n;
            };
            auto push_site = [&](int32_t* __restrict__ k, int& n, int v) {
                if (n >= kCutMaxSeg) {
                    bad = true;
                    return;
                }
                k[4 + 2 * kCutMaxSeg + n] = v;
                ++n;
            };
            if (best1 - cut1 <= best2 - cut2) {
                if (!(best1 > cut1 || cuts < kCutMinSize || V - cuts < kCutMinSize)) {
                    action = 1;
                    // SplitDomain (src/RecCut.cpp:16-63), as reccut.cpp's split_one
                    int c = 0;
                    while (c < n_site && sites[c] < cuts) {
                        if (!seg_ok(c)) bad = true;
                        else push_seg(k1, n1s, segs[2 * c], segs[2 * c + 1]);
                        push_site(k1, n1c, sites[c]);
                        ++c;
                        if (bad) break;
                    }
                    if (!bad && !seg_ok(c)) bad = true;
                    if (!bad) {
                        const int len1 = c == 0 ? cuts : cuts - sites[c - 1];
                        const int mf = segs[2 * c], ms = segs[2 * c + 1];
                        push_seg(k1, n1s, mf, mf + len1 - 1);
                        const int len2 = (n_site == 0 || c == n_site) ? V - cuts : sites[c] - cuts;
                        if (len2 > 0) {
                            push_seg(k2, n2s, ms - len2 + 1, ms);
                            if (c < n_site) push_site(k2, n2c, len2);
                        }
                        for (++c; c < n_seg; ++c) {
                            push_seg(k2, n2s, segs[2 * c], segs[2 * c + 1]);
                            if (c < n_site && len1 > 0) push_site(k2, n2c, len1);
                        }
                    }
                    k1[1] = cuts;
                    k2[1] = V - cuts;
                }
            } else {
                const int length = cuts2 - cuts1;
                if (!(best2 > cut2 || length < kCutMinSize || V - length < kCutMinSize)) {
                    action = 2;
                    // SplitDomain_2cuts (src/RecCut.cpp:65-148), as reccut.cpp's split_two
                    const int ns = n_site;
                    const int tail = V - cuts2;
                    int c1 = -1, c2 = 0;
                    for (int i = ns - 1; i >= 0; --i) {
                        if (sites[i] < cuts2) c2 = i + 1;
                        if (sites[i] <= cuts1) {
                            c1 = i;
                            break;
                        }
                    }
                    // cs_at: index -1 reads 0 (with a non-empty table), any other index outside the table is undefined
                    auto cs_at = [&](int i) {
                        if (i == -1 && ns > 0) return 0;
                        if (i < 0 || i >= ns) {
                            bad = true;
                            return 0;
                        }
                        return sites[i];
                    };
                    const int len2 = c2 == 0 ? cuts2 : cuts2 - cs_at(c2 - 1);
                    if (!seg_ok(c2)) bad = true;
                    if (!bad) {
                        push_seg(k1, n1s, segs[2 * c2] + len2, segs[2 * c2 + 1]);
                        for (int i = c2 + 1; i <= ns && !bad; ++i) {
                            if (!seg_ok(i)) bad = true;
                            else push_seg(k1, n1s, segs[2 * i], segs[2 * i + 1]);
                        }
                        for (int i = c2; i < ns; ++i) push_site(k1, n1c, sites[i] - cuts2);
                        push_site(k1, n1c, tail);
                        for (int i = 0; i <= c1 && !bad; ++i) {
                            if (!seg_ok(i)) bad = true;
                            else push_seg(k1, n1s, segs[2 * i], segs[2 * i + 1]);
                            if (sites[i] != cuts1) push_site(k1, n1c, sites[i] + tail);
                        }
                    }
                    if (!bad) {
                        const int len1 = c1 >= 0 ? cuts1 - sites[c1] : cuts1;
                        ++c1;
                        if (len1 > 0) {
                            if (!seg_ok(c1)) bad = true;
                            else push_seg(k1, n1s, segs[2 * c1], segs[2 * c1] + len1 - 1);
                        }
                        if (!bad && c2 > c1) {
                            const int lhs = cs_at(c1 - 1) + len1, rhs = cs_at(c1);
                            if (!bad && lhs < rhs) {
                                if (!seg_ok(c1)) bad = true;
                                else push_seg(k2, n2s, segs[2 * c1] + len1, segs[2 * c1 + 1]);
                            }
                        }
                        for (int i = c1 + 1; i < c2 && !bad; ++i) {
                            if (!seg_ok(i)) bad = true;
                            else push_seg(k2, n2s, segs[2 * i], segs[2 * i + 1]);
                        }
                        if (!bad) {
                            if (c2 > c1 && len2 > 0) {
                                if (!seg_ok(c2)) bad = true;
                                else push_seg(k2, n2s, segs[2 * c2], segs[2 * c2] + len2 - 1);
                            } else if (c1 == c2) {
                                if (!seg_ok(c2)) bad = true;
                                else push_seg(k2, n2s, segs[2 * c2] + len1, segs[2 * c2] + len2 - 1);
                            }
                        }
                        for (int i = c1; i < c2 && !bad; ++i) {
                            const int v = cs_at(i) - cuts1;
                            if (!bad && v > 0) push_site(k2, n2c, v);
                        }
                    }
                    k1[1] = V - length;
                    k2[1] = length;
                }
            }
            if (action != 0) {
                if (bad || misc[11] >= kCutStack) action = -1;
                else {
                    k1[0] = first;
                    k1[2] = n1s;
                    k1[3] = n1c;
                    k2[0] = first + k1[1];
                    k2[2] = n2s;
                    k2[3] = n2c;
                }
            } else {   // this node is a domain: its segments go out in order
                int at = misc[9];
                if (at + 1 + 2 * n_seg > job.out_cap) action = -1;
                else {
                    out[at++] = n_seg;
                    for (int s = 0; s < n_seg; ++s) {
                        out[at++] = segs[2 * s];
                        out[at++] = segs[2 * s + 1];
                    }
                    misc[9] = at;
                    misc[10] += 1;
                }
            }
            misc[5] = action;
            misc[6] = cuts;
            misc[7] = cuts1;
            misc[8] = cuts2;
        }
        __syncthreads();
        CUT_T(4);   // decision
        const int action = misc[5];
        for (int p = tid; p < V; p += kCutThreads) pos[p] = -1;       // (band 1's start vector sat in pos[0 .. V))
        __syncthreads();
        for (int p = tid; p < V; p += kCutThreads) pos[ix[p]] = -1;
        if (action < 0) {
            if (tid == 0) out[0] = -1;
            return;
        }
        if (action == 2) {   // the two flanks (tail first, then head) become the first child, the middle the second
            const int c1 = misc[7], c2 = misc[8];
            const int n_tail = V - c2;
            for (int p = tid; p < V; p += kCutThreads) {
                int to;
                if (p >= c2) to = p - c2;
                else if (p < c1) to = n_tail + p;
                else to = n_tail + c1 + (p - c1);
                tmp[to] = ix[p];
            }
            __syncthreads();
            for (int p = tid; p < V; p += kCutThreads) ix[p] = tmp[p];
        }
        __syncthreads();
        if (action == 0) {   // a domain is out: the next pending node, or done
            if (misc[11] == 0) {
                if (tid == 0) out[0] = misc[10];
                CUT_T(5);
                CUT_T_OUT();
                return;
            }
            const int32_t* __restrict__ src = job.stack + (size_t)(misc[11] - 1) * kCutNodeInts;
            for (int q = tid; q < kCutNodeInts; q += kCutThreads) node[q] = src[q];
            __syncthreads();
            if (tid == 0) misc[11] -= 1;
        } else {             // the second child waits, the first is next
            int32_t* __restrict__ dst = job.stack + (size_t)misc[11] * kCutNodeInts;
            for (int q = tid; q < kCutNodeInts; q += kCutThreads) {
                dst[q] = kid[kCutNodeInts + q];
                node[q] = kid[q];
            }
            __threadfence_block();
            __syncthreads();
            if (tid == 0) misc[11] += 1;
        }
        __syncthreads();
        CUT_T(5);   // node hand-over
#ifdef DCTFP_CUT_TIMING
        tl_acc[6] += 1;   // nodes
        tl_acc[7] += (unsigned long long)V * V;
#endif
    }
}

constexpr size_t reccut_lds_bytes(int cap, int ecap, int rb, int kmax, int threads) {
    return ((size_t)7 * cap + 8 + ecap + 3 * kCutNodeInts + 32 + (size_t)rb * (cap / 64 + 1)) * 4 + (size_t)rb * cap +
           ((size_t)(kmax > 2 ? kmax - 2 : 0) * cap + (size_t)(kmax > 1 ? kmax - 1 : 1) * (threads / 64)) * 4;
}

}  // namespace dctfp
