// libreccut.so -- in-process restatement of the reference's domain cutter
// (mgtools/DCTdomain src/RecCut.cpp; C ABI in include/reccut.h).
//
// Written from the algorithm, not from the file's structure: the contact graph is one
// dense byte matrix and every recursion level works on an index list into it (the
// reference copies sub-matrices); range weights come from running sums produced four rows
// at a time and scanned as they appear (no V x V table of ints), and whole chunks of the
// O(V^2) double-cut scan are skipped by a conservative vectorised test (round 4: 4 x faster
// than the straightforward loops, same strings on every golden and fuzz case).  What must be
// -- and is -- identical are the integers and the double expressions that decide a cut:
//   * graph weights, src/RecCut.cpp:384-393;
//   * single cut score  (double)cutv * sum / N1[i] / N2[i], strict '<', first minimum
//     (:186-201), including N2[0] = sum never being reduced by vertex 0's edges (:178-183);
//   * double cut score over i in [10, V-10), j in [i+21, V-10)  (:243-260);
//   * the accept / recurse rules with thresholds 0.08 / 0.07 and Min_Size 22 (:263-350);
//   * the segment / cut-site bookkeeping of SplitDomain and SplitDomain_2cuts (:16-148),
//     including its quirks (see split_one / split_two), because it decides the printed
//     domain strings.
#include "reccut.h"

#include <emmintrin.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <thread>
#include <utility>
#include <vector>

namespace {

constexpr int kMinTerminal = 10;  // src/RecCut.cpp:10
constexpr int kMinSize = 22;      // src/RecCut.cpp:14

typedef std::pair<int, int> Seg;  // [first, second] original 0-based residues
typedef std::vector<Seg> Domain;

struct Undefined {};  // thrown where the reference would index outside a vector

// The contact graph as a dense n0 x n0 matrix of weights.  W = uint8_t whenever every weight fits (a probability gives
// 0..100: a quarter of the bytes, and the sums below vectorise), int32_t for whatever else a caller hands in.
template <typename W>
struct Ctx {
    int n0;                  // residues of the protein
    std::vector<W> w;        // dense n0 x n0 weights
    double cut1, cut2;
};

inline const Seg& seg_at(const Domain& d, long i) {
    if (i < 0 || i >= (long)d.size()) throw Undefined();
    return d[(size_t)i];
}

// cut-site read with the reference's out-of-range behaviour: index -1 reads the word in
// front of a std::vector<int> buffer, which is 0 with glibc's allocator (upper half of the
// chunk size); any other out-of-range index has no defined value.
inline int cs_at(const std::vector<int>& c, long i) {
    if (i == -1 && !c.empty()) return 0;
    if (i < 0 || i >= (long)c.size()) throw Undefined();
    return c[(size_t)i];
}

// SplitDomain (src/RecCut.cpp:16-63): one cut at `cuts` inside a domain made of `segs`
// joined at `sites`.
void split_one(int cuts, const Domain& segs, const std::vector<int>& sites, int length, Domain& d1, Domain& d2,
               std::vector<int>& s1, std::vector<int>& s2) {
    long c = 0;
    const long ns = (long)sites.size();
    while (c < ns && sites[(size_t)c] < cuts) {
        d1.push_back(seg_at(segs, c));
        s1.push_back(sites[(size_t)c]);
        ++c;
    }
    const int len1 = (c == 0) ? cuts : cuts - sites[(size_t)(c - 1)];
    const Seg& mid = seg_at(segs, c);
    d1.push_back(Seg(mid.first, mid.first + len1 - 1));
    const int len2 = (ns == 0 || c == ns) ? length - cuts : sites[(size_t)c] - cuts;
    if (len2 > 0) {
        d2.push_back(Seg(mid.second - len2 + 1, mid.second));
        if (c < ns) s2.push_back(len2);
    }
    for (++c; c < (long)segs.size(); ++c) {
        d2.push_back(segs[(size_t)c]);
        if (c < ns && len1 > 0) s2.push_back(len1);  // the reference records len1 here, not a running offset
    }
}

// SplitDomain_2cuts (src/RecCut.cpp:65-148): the part between cuts1 and cuts2 becomes the
// second domain, the two flanks (tail first, then head) the first.
void split_two(int cuts1, int cuts2, const Domain& segs, const std::vector<int>& sites, int length, Domain& d1,
               Domain& d2, std::vector<int>& s1, std::vector<int>& s2) {
    const long ns = (long)sites.size();
    const int tail = length - cuts2;
    long c1 = -1, c2 = 0;
    for (long i = ns - 1; i >= 0; --i) {
        if (sites[(size_t)i] < cuts2) c2 = i + 1;  // keeps being overwritten on the way down
        if (sites[(size_t)i] <= cuts1) {
            c1 = i;
            break;
        }
    }
    const int len2 = (c2 == 0) ? cuts2 : cuts2 - cs_at(sites, c2 - 1);
    {
        const Seg& s = seg_at(segs, c2);
        d1.push_back(Seg(s.first + len2, s.second));
    }
    for (long i = c2 + 1; i <= ns; ++i) d1.push_back(seg_at(segs, i));
    for (long i = c2; i < ns; ++i) s1.push_back(sites[(size_t)i] - cuts2);
    s1.push_back(tail);
    for (long i = 0; i <= c1; ++i) {
        d1.push_back(seg_at(segs, i));
        if (sites[(size_t)i] != cuts1) s1.push_back(sites[(size_t)i] + tail);
    }
    const int len1 = (c1 >= 0) ? cuts1 - sites[(size_t)c1] : cuts1;
    ++c1;
    if (len1 > 0) {
        const Seg& s = seg_at(segs, c1);
        d1.push_back(Seg(s.first, s.first + len1 - 1));
    }
    if (c2 > c1 && cs_at(sites, c1 - 1) + len1 < cs_at(sites, c1)) {
        const Seg& s = seg_at(segs, c1);
        d2.push_back(Seg(s.first + len1, s.second));
    }
    for (long i = c1 + 1; i < c2; ++i) d2.push_back(seg_at(segs, i));
    if (c2 > c1 && len2 > 0) {
        const Seg& s = seg_at(segs, c2);
        d2.push_back(Seg(s.first, s.first + len2 - 1));
    } else if (c1 == c2) {
        const Seg& s = seg_at(segs, c2);
        d2.push_back(Seg(s.first + len1, s.first + len2 - 1));
    }
    for (long i = c1; i < c2; ++i) {
        const int v = cs_at(sites, i) - cuts1;
        if (v > 0) s2.push_back(v);
    }
}

// Sum of n weights.  Bytes go 16 at a time through psadbw (SSE2: part of every x86-64).
// The last 1..15 bytes come in one masked 16-byte load (every buffer handed in here is followed by 16 readable bytes): a
// scalar tail was most of the time of a 150-byte range.
alignas(16) static const uint8_t kTailMask[32] = {255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
                                                   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0};
inline int sum_range(const uint8_t* p, int n) {
    __m128i acc = _mm_setzero_si128();
    int j = 0;
    for (; j + 16 <= n; j += 16) acc = _mm_add_epi64(acc, _mm_sad_epu8(_mm_loadu_si128((const __m128i*)(p + j)), _mm_setzero_si128()));
    if (j < n) {
        const __m128i m = _mm_loadu_si128((const __m128i*)(kTailMask + 16 - (n - j)));
        acc = _mm_add_epi64(acc, _mm_sad_epu8(_mm_and_si128(_mm_loadu_si128((const __m128i*)(p + j)), m), _mm_setzero_si128()));
    }
    return _mm_cvtsi128_si32(acc) + _mm_cvtsi128_si32(_mm_srli_si128(acc, 8));
}
inline int sum_range(const int32_t* p, int n) {
    int s = 0;
    for (int j = 0; j < n; ++j) s += p[j];
    return s;
}

// Conservative single-precision test over the candidates j in [j0, j1) of row i of the double-cut scan: true when some pair
// MIGHT satisfy `ave < best2` (so the caller runs the reference's own double expression over the chunk), false when none
// can.  All integer quantities are below 2^23 (checked by the caller), so their float images are exact and the three
// float products are off by < 3e-7 relative -- against a margin of 1e-4 in `best_f`.  Written so that the compiler
// vectorises it (no branch, an OR reduction); target_clones gives AVX2 / AVX-512 hosts their own build behind one symbol.
__attribute__((target_clones("avx2", "default")))
int chunk_may_win(const int* __restrict__ inner_row, const int* __restrict__ t, int c_i, int sum, float sum_f, float best_f,
                  int j0, int j1) {
    int any = 0;
    for (int j = j0; j < j1; ++j) {
        const int ns2 = inner_row[j];
        const int cv = t[j] + c_i - 2 * ns2;
        const int ns1 = sum - cv - ns2;
        const float num = (float)cv * sum_f;
        const float rhs = best_f * (float)ns1 * (float)ns2;
        any |= (int)(ns1 > 0) & (int)(ns2 > 0) & (int)(num <= rhs);
    }
    return any;
}

// recursiveMaxCut (src/RecCut.cpp:150-351) on the vertices idx[0..V) of the protein graph.
// `out` receives the final domains of this subtree in the reference's order.
template <typename W>
void cut_rec(const Ctx<W>& cx, const std::vector<int>& idx, const std::vector<int>& sites, const Domain& segs,
             std::vector<Domain>& out) {
    const int V = (int)idx.size();
    if (V < kMinSize) throw Undefined();  // the binary prints "Protein has length of 0" and exits (-1)

    // The weights in current vertex order: where the vertices are one run of consecutive residues (the whole protein, both
    // sides of a single cut, the middle of a double cut) that is a window of the graph itself -- rows n0 apart, no copy;
    // otherwise (the two flanks of a double cut joined) a local dense copy.
    const bool window = idx[V - 1] - idx[0] == V - 1;
    std::vector<W> copy;
    const W* a = window ? &cx.w[(size_t)idx[0] * cx.n0 + idx[0]] : nullptr;
    const size_t lda = window ? (size_t)cx.n0 : (size_t)V;
    if (!window) {
        copy.resize((size_t)V * V + 16);   // (+16: see sum_range)
        // idx is a handful of runs of consecutive residues: copy run by run instead of element by element
        std::vector<std::pair<int, int>> runs;  // (first position in idx, length)
        for (int j = 0; j < V;) {
            int e = j + 1;
            while (e < V && idx[e] == idx[e - 1] + 1) ++e;
            runs.emplace_back(j, e - j);
            j = e;
        }
        for (int i = 0; i < V; ++i) {
            const W* row = &cx.w[(size_t)idx[i] * cx.n0];
            W* dst = &copy[(size_t)i * V];
            for (const auto& r : runs) std::memcpy(dst + r.first, row + idx[r.first], (size_t)r.second * sizeof(W));
        }
        a = copy.data();
    }

    std::vector<int> pre(V, 0), post(V, 0);
    int sum = 0;
    for (int i = 0; i < V; ++i) {
        const W* ai = a + (size_t)i * lda;
        const int p = sum_range(ai, i), q = sum_range(ai + i + 1, V - i - 1);
        pre[i] = p;
        post[i] = q;
        sum += q;
    }
    std::vector<int> n1(V), n2(V + 1);
    n1[0] = 0;
    n2[0] = sum;
    int cutv = post[0];
    double best1 = 2.0;
    int cuts = 0;
    for (int i = 1; i < V; ++i) {
        n1[i] = n1[i - 1] + pre[i];
        n2[i] = n2[i - 1] - post[i];
        if (i < V - 2) {
            cutv = cutv + post[i] - pre[i];
            const double ave = ((double)cutv) * sum / n1[i] / n2[i];
            if (ave < best1) {
                best1 = ave;
                cuts = i + 1;
            }
        }
    }
    n2[V] = 0;  // (never read by a candidate: j + 1 <= V - kMinTerminal)

    // cv(i, j) = n1[j] + n2[i] - n1[i-1] - n2[j+1] - 2 inner[i][j] = t[j] + c_i - 2 inner[i][j]
    std::vector<int> t(V, 0);
    for (int j = 0; j + 1 <= V; ++j) t[j] = n1[j] - n2[j + 1];
    // The double-cut scan.  inner[i][j] = total weight inside the vertex range [i, j] (S of the reference) obeys
    //   inner[i][j] = inner[i+1][j] + sum_{b = i+1..j} a[i][b],
    // so its rows are produced from the bottom up, FOUR AT A TIME IN LOCKSTEP (the running sum of a row is a dependent
    // chain of additions: four rows give the core four chains), and each row is scanned as soon as it exists -- five rows
    // of ints live at any time instead of a V x V matrix that does not fit the cache.  Scanning the rows in descending
    // order keeps the reference's answer (first minimum in ascending (i, j) order, strict '<' from 2.0): within a row the
    // first strict minimum, and a later row (smaller i) takes over on `<=`.
    std::vector<int> rows((size_t)5 * V, 0);
    int* below = rows.data();                      // row ib (all zero for ib = V - 1)
    int* h[4] = {below + V, below + 2 * V, below + 3 * V, below + 4 * V};
    double best2 = 2.0;
    bool found = false;
    int cuts1 = 0, cuts2 = 0;
    const bool small = sum > 0 && sum < (1 << 23);  // every integer of the scan is exact in float
    const float sum_f = (float)sum;
    constexpr int kChunk = 64;
    for (int ib = V - 1; ib > kMinTerminal;) {
        const int nrow = std::min(4, ib - kMinTerminal);          // rows ib-1 .. ib-nrow
        const W* ar[4];
        int r[4] = {0, 0, 0, 0};
        for (int k = 0; k < nrow; ++k) ar[k] = a + (size_t)(ib - 1 - k) * lda;
        for (int k = 1; k < nrow; ++k) {                          // the entries left of column ib: a few per row
            const int i = ib - 1 - k;
            for (int j = i + 1; j < ib; ++j) {
                r[k] += ar[k][j];
                h[k][j] = (j > i + 1 ? h[k - 1][j] : 0) + r[k];
            }
        }
        if (nrow == 4) {
            int r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
            const W *a0 = ar[0], *a1 = ar[1], *a2 = ar[2], *a3 = ar[3];
            int *h0 = h[0], *h1 = h[1], *h2 = h[2], *h3 = h[3];
            for (int j = ib; j < V; ++j) {
                r0 += a0[j];
                r1 += a1[j];
                r2 += a2[j];
                r3 += a3[j];
                const int v0 = below[j] + r0, v1 = v0 + r1, v2 = v1 + r2, v3 = v2 + r3;
                h0[j] = v0;
                h1[j] = v1;
                h2[j] = v2;
                h3[j] = v3;
            }
        } else {
            for (int j = ib; j < V; ++j) {
                int v = below[j];
                for (int k = 0; k < nrow; ++k) {
                    r[k] += ar[k][j];
                    v += r[k];
                    h[k][j] = v;
                }
            }
        }
        for (int k = 0; k < nrow; ++k) {
            const int i = ib - 1 - k;
            if (i >= V - kMinTerminal) continue;
            const int* row = h[k];
            const int c_i = n2[i] - n1[i - 1];
            double row_best = 1e300;
            int row_j = -1;
            for (int j0 = i + kMinSize - 1; j0 < V - kMinTerminal; j0 += kChunk) {
                const int j1 = std::min(j0 + kChunk, V - kMinTerminal);
                const double thr = std::min(best2, row_best);
                // The two divisions per candidate pair dominated the run time: whole chunks of pairs that cannot win are
                // skipped by a conservative test (single precision, margin 1e-4, vectorised); every pair of a chunk that may
                // hold a winner goes through the reference's own double expression.
                if (small && !chunk_may_win(row, t.data(), c_i, sum, sum_f, (float)(thr * 1.0001), j0, j1)) continue;
                for (int j = j0; j < j1; ++j) {
                    const int ns2 = row[j];
                    const int cv = n1[j] + n2[i] - n1[i - 1] - n2[j + 1] - ns2 * 2;
                    const int ns1 = sum - cv - ns2;
                    if (ns1 > 0 && ns2 > 0) {
                        const double num = ((double)cv) * sum;
                        if (num > best2 * (double)ns1 * (double)ns2 * 1.000000001) continue;  // (ave > best2: cannot win)
                        const double ave = num / ns1 / ns2;
                        if (ave < row_best && (found ? ave <= best2 : ave < best2)) {
                            row_best = ave;
                            row_j = j;
                        }
                    }
                }
            }
            if (row_j >= 0) {
                best2 = row_best;
                cuts1 = i;
                cuts2 = row_j;
                found = true;
            }
        }
        std::swap(below, h[nrow - 1]);                            // the lowest row of this block is `below` of the next
        ib -= nrow;
    }

    Domain d1, d2;
    std::vector<int> s1, s2;
    std::vector<int> idx1, idx2;
    if (best1 - cx.cut1 <= best2 - cx.cut2) {
        if (best1 > cx.cut1 || cuts < kMinSize || V - cuts < kMinSize) {
            out.push_back(segs);
            return;
        }
        idx1.assign(idx.begin(), idx.begin() + cuts);
        idx2.assign(idx.begin() + cuts, idx.end());
        split_one(cuts, segs, sites, V, d1, d2, s1, s2);
    } else {
        const int length = cuts2 - cuts1;
        if (best2 > cx.cut2 || length < kMinSize || V - length < kMinSize) {
            out.push_back(segs);
            return;
        }
        idx1.assign(idx.begin() + cuts2, idx.end());  // tail flank first ...
        idx1.insert(idx1.end(), idx.begin(), idx.begin() + cuts1);  // ... then the head flank
        idx2.assign(idx.begin() + cuts1, idx.begin() + cuts2);
        split_two(cuts1, cuts2, segs, sites, V, d1, d2, s1, s2);
    }
    copy.clear();
    copy.shrink_to_fit();
    rows.clear();
    rows.shrink_to_fit();
    cut_rec(cx, idx1, s1, d1, out);
    cut_rec(cx, idx2, s2, d2, out);
}

// (int)(strtod("%.6f" % p) * 100 + 0.5): the text round trip only matters next to a rounding boundary
int contact_weight(float p) {
    const double x = (double)p * 100 + 0.5;
    const double fl = std::floor(x);
    // printing to 6 decimals moves x by at most 5e-5 (+ a few ulp): away from an integer by 1e-3 nothing can change
    if (x - fl > 1e-3 && fl + 1 - x > 1e-3 && x > -2e9 && x < 2e9) return (int)x;
    char buf[64];
    snprintf(buf, sizeof buf, "%.6f", (double)p);
    const double v = strtod(buf, nullptr);
    return (int)(v * 100 + 0.5);
}

// Fills the dense graph (readGraph, src/RecCut.cpp:354-397) and runs the recursion; false = a contact outside the protein.
template <typename W>
int cut_protein(int32_t n_res, const int32_t* ci, const int32_t* cj, const int32_t* wgt, int64_t n_contacts, double cut1,
                double cut2, const Domain& whole, std::vector<Domain>& doms) {
    Ctx<W> cx;
    cx.n0 = n_res;
    cx.cut1 = cut1;
    cx.cut2 = cut2;
    cx.w.assign((size_t)n_res * n_res + 16, 0);   // (+16: sum_range's last load may reach past the last row)
    for (int64_t k = 0; k < n_contacts; ++k) {
        const int i = ci[k], j = cj[k];
        cx.w[(size_t)i * n_res + j] = (W)wgt[k];
        cx.w[(size_t)j * n_res + i] = (W)wgt[k];
    }
    for (int i = 0; i < n_res; ++i)
        for (int d = 1; d <= 3 && i + d < n_res; ++d) {
            cx.w[(size_t)i * n_res + i + d] = 100;
            cx.w[(size_t)(i + d) * n_res + i] = 100;
        }
    std::vector<int> idx(n_res);
    for (int i = 0; i < n_res; ++i) idx[i] = i;
    try {
        cut_rec(cx, idx, std::vector<int>(), whole, doms);
    } catch (const Undefined&) {
        return RECCUT_ERR_UNDEFINED;
    }
    return RECCUT_OK;
}

int predict_impl(int32_t n_res, const int32_t* ci, const int32_t* cj, const float* prob, int64_t n_contacts,
                 double cut1, double cut2, std::string& text, int32_t* n_domains) {
    if (n_res <= 0 || n_contacts < 0 || (n_contacts > 0 && (!ci || !cj || !prob))) return RECCUT_ERR_INVALID;
    std::vector<Domain> doms;
    Domain whole(1, Seg(0, n_res - 1));
    if (n_res >= kMinSize) {
        std::vector<int32_t> wgt((size_t)n_contacts);
        bool bytes = true;
        for (int64_t k = 0; k < n_contacts; ++k) {
            const int i = ci[k], j = cj[k];
            if (i < 0 || j < 0 || i >= n_res || j >= n_res) return RECCUT_ERR_INVALID;
            wgt[(size_t)k] = contact_weight(prob[k]);  // the value as the .ce file carries it
            bytes = bytes && wgt[(size_t)k] >= 0 && wgt[(size_t)k] <= 255;
        }
        const int rc = bytes ? cut_protein<uint8_t>(n_res, ci, cj, wgt.data(), n_contacts, cut1, cut2, whole, doms)
                             : cut_protein<int32_t>(n_res, ci, cj, wgt.data(), n_contacts, cut1, cut2, whole, doms);
        if (rc != RECCUT_OK) return rc;
    } else {
        doms.push_back(whole);
    }
    text.clear();
    char buf[48];
    for (const Domain& d : doms) {
        for (size_t s = 0; s < d.size(); ++s) {
            snprintf(buf, sizeof buf, "%s%d-%d", s ? "," : "", d[s].first + 1, d[s].second + 1);
            text += buf;
        }
        text += ';';
    }
    if (n_domains) *n_domains = (int32_t)doms.size();
    return RECCUT_OK;
}

}  // namespace

extern "C" {

int32_t reccut_contact_weight(float prob) { return contact_weight(prob); }

/* The same in exact arithmetic, no text round trip (what the GPU kernel computes: csrc/reccut_kernel.hip.h): p * 10^6 is exact
 * in double, printf rounds it half-to-even, strtod returns the double nearest to n / 10^6. */
int32_t reccut_contact_weight_exact(float prob) {
    const double n6 = std::rint((double)prob * 1.0e6);
    const double v = n6 / 1.0e6;
    volatile double prod = v * 100.0;
    return (int)(prod + 0.5);
}

/* Strings of dctfp_reccut's encoded results (include/dctfp.h), packed like reccut_predict_packed's. */
int reccut_format_packed(int64_t n_prot, const int32_t* enc, const int64_t* enc_off, char* out, int64_t out_cap, int64_t* out_off,
                         int32_t* n_domains, uint8_t* needs_host) {
    if (n_prot < 0 || !enc || !enc_off || !out || !out_off || !needs_host) return RECCUT_ERR_INVALID;
    int64_t at = 0;
    char buf[48];
    for (int64_t p = 0; p < n_prot; ++p) {
        out_off[p] = at;
        const int32_t* e = enc + enc_off[p];
        const int64_t room = enc_off[p + 1] - enc_off[p];
        needs_host[p] = 0;
        if (n_domains) n_domains[p] = 0;
        if (room < 1 || e[0] < 1) {
            needs_host[p] = 1;
            continue;
        }
        int64_t q = 1;
        bool ok = true;
        const int64_t begin = at;
        for (int32_t d = 0; d < e[0] && ok; ++d) {
            if (q >= room) { ok = false; break; }
            const int32_t ns = e[q++];
            if (ns < 1 || q + 2 * (int64_t)ns > room) { ok = false; break; }
            for (int32_t s = 0; s < ns; ++s) {
                const int n = snprintf(buf, sizeof buf, "%s%d-%d", s ? "," : "", e[q] + 1, e[q + 1] + 1);
                q += 2;
                if (at + n + 1 > out_cap) return RECCUT_ERR_BUFFER;
                memcpy(out + at, buf, (size_t)n);
                at += n;
            }
            out[at++] = ';';
        }
        if (!ok) {
            at = begin;
            needs_host[p] = 1;
            continue;
        }
        if (n_domains) n_domains[p] = e[0];
    }
    out_off[n_prot] = at;
    return RECCUT_OK;
}

int reccut_predict(int32_t n_res, const int32_t* ci, const int32_t* cj, const float* prob, int64_t n_contacts,
                   double cut1, double cut2, char* out, int64_t out_cap, int32_t* n_domains) {
    if (!out || out_cap < 1) return RECCUT_ERR_INVALID;
    std::string text;
    int rc;
    try {
        rc = predict_impl(n_res, ci, cj, prob, n_contacts, cut1, cut2, text, n_domains);
    } catch (...) {
        return RECCUT_ERR_INVALID;
    }
    if (rc != RECCUT_OK) return rc;
    if ((int64_t)text.size() + 1 > out_cap) return RECCUT_ERR_BUFFER;
    memcpy(out, text.c_str(), text.size() + 1);
    return RECCUT_OK;
}

int reccut_predict_batch(int64_t n_prot, const int32_t* n_res, const int64_t* offs, const int32_t* ci,
                         const int32_t* cj, const float* prob, double cut1, double cut2, char* out,
                         int64_t out_stride, int32_t* n_domains, int32_t* rc, int32_t n_threads) {
    if (n_prot < 0 || !n_res || !offs || !out || !rc || out_stride < 1) return RECCUT_ERR_INVALID;
    if (n_threads < 1) n_threads = 1;
    auto work = [&](int64_t p0, int64_t step) {
        for (int64_t p = p0; p < n_prot; p += step) {
            const int64_t b = offs[p], e = offs[p + 1];
            rc[p] = reccut_predict(n_res[p], ci ? ci + b : nullptr, cj ? cj + b : nullptr, prob ? prob + b : nullptr,
                                   e - b, cut1, cut2, out + p * out_stride, out_stride,
                                   n_domains ? n_domains + p : nullptr);
        }
    };
    if (n_threads == 1 || n_prot < 2) {
        work(0, 1);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < n_threads; ++t) pool.emplace_back(work, (int64_t)t, (int64_t)n_threads);
        for (auto& t : pool) t.join();
    }
    return RECCUT_OK;
}

/* reccut_predict_batch with the strings packed back to back (see include/reccut.h). */
int reccut_predict_packed(int64_t n_prot, const int32_t* n_res, const int64_t* offs, const int32_t* ci, const int32_t* cj,
                          const float* prob, double cut1, double cut2, char* out, int64_t out_cap, int64_t* out_off,
                          int32_t* n_domains, int32_t* rc, int32_t n_threads) {
    if (n_prot < 0 || !n_res || !offs || !out || !out_off || !rc || out_cap < 0) return RECCUT_ERR_INVALID;
    if (n_threads < 1) n_threads = 1;
    try {
        std::vector<std::string> text((size_t)n_prot);
        std::atomic<int64_t> next(0);
        auto work = [&]() {
            for (;;) {
                const int64_t p = next.fetch_add(1);
                if (p >= n_prot) return;
                const int64_t b = offs[p], e = offs[p + 1];
                try {
                    rc[p] = predict_impl(n_res[p], ci ? ci + b : nullptr, cj ? cj + b : nullptr, prob ? prob + b : nullptr, e - b,
                                         cut1, cut2, text[(size_t)p], n_domains ? n_domains + p : nullptr);
                } catch (...) {
                    rc[p] = RECCUT_ERR_INVALID;
                }
                if (rc[p] != RECCUT_OK) text[(size_t)p].clear();
            }
        };
        if (n_threads == 1 || n_prot < 2) {
            work();
        } else {
            std::vector<std::thread> pool;
            for (int t = 0; t < n_threads; ++t) pool.emplace_back(work);
            for (auto& t : pool) t.join();
        }
        int64_t at = 0;
        for (int64_t p = 0; p < n_prot; ++p) {
            out_off[p] = at;
            const std::string& s = text[(size_t)p];
            if (at + (int64_t)s.size() > out_cap) return RECCUT_ERR_BUFFER;
            memcpy(out + at, s.data(), s.size());
            at += (int64_t)s.size();
        }
        out_off[n_prot] = at;
    } catch (...) {
        return RECCUT_ERR_INVALID;
    }
    return RECCUT_OK;
}

}  // extern "C"
