// libreccut.so -- in-process restatement of the reference's domain cutter
// (mgtools/DCTdomain src/RecCut.cpp; C ABI in include/reccut.h).
//
// Written from the algorithm, not from the file's structure: the contact graph is one
// dense int matrix and every recursion level works on an index list into it (the
// reference copies sub-matrices); range weights come from running sums.  What must be
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

struct Ctx {
    int n0;                  // residues of the protein
    std::vector<int> w;      // dense n0 x n0 weights
    double cut1, cut2;
    int at(int a, int b) const { return w[(size_t)a * n0 + b]; }
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

// recursiveMaxCut (src/RecCut.cpp:150-351) on the vertices idx[0..V) of the protein graph.
// `out` receives the final domains of this subtree in the reference's order.
void cut_rec(const Ctx& cx, const std::vector<int>& idx, const std::vector<int>& sites, const Domain& segs,
             std::vector<Domain>& out) {
    const int V = (int)idx.size();
    if (V < kMinSize) throw Undefined();  // the binary prints "Protein has length of 0" and exits (-1)

    // local dense copy in current vertex order
    std::vector<int> a((size_t)V * V);
    {
        // idx is a handful of runs of consecutive residues: copy run by run instead of element by element
        std::vector<std::pair<int, int>> runs;  // (first position in idx, length)
        for (int j = 0; j < V;) {
            int e = j + 1;
            while (e < V && idx[e] == idx[e - 1] + 1) ++e;
            runs.emplace_back(j, e - j);
            j = e;
        }
        for (int i = 0; i < V; ++i) {
            const int* row = &cx.w[(size_t)idx[i] * cx.n0];
            int* dst = &a[(size_t)i * V];
            for (const auto& r : runs) std::memcpy(dst + r.first, row + idx[r.first], (size_t)r.second * sizeof(int));
        }
    }
    auto A = [&](int i, int j) { return a[(size_t)i * V + j]; };

    std::vector<int> pre(V, 0), post(V, 0);
    int sum = 0;
    for (int i = 0; i < V; ++i) {
        int p = 0, q = 0;
        for (int j = 0; j < i; ++j) p += A(i, j);
        for (int j = i + 1; j < V; ++j) q += A(i, j);
        pre[i] = p;
        post[i] = q;
        sum += q;
    }
    std::vector<int> n1(V), n2(V);
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

    // inner[i][j] = total weight inside the vertex range [i, j]  (S of the reference)
    std::vector<int> inner((size_t)V * V, 0);
    for (int i = V - 2; i >= 0; --i) {  // row i from row i + 1, both walked contiguously
        const int* ai = &a[(size_t)i * V];
        const int* below = &inner[(size_t)(i + 1) * V];
        int* here = &inner[(size_t)i * V];
        int r = 0;  // sum_{b = i+1..j} a[i][b]
        for (int j = i + 1; j < V; ++j) {
            r += ai[j];
            here[j] = below[j] + r;
        }
    }
    double best2 = 2.0;
    int cuts1 = 0, cuts2 = 0;
    for (int i = kMinTerminal; i < V - kMinTerminal; ++i) {
        for (int j = i + kMinSize - 1; j < V - kMinTerminal; ++j) {
            const int ns2 = inner[(size_t)i * V + j];
            const int cv = n1[j] + n2[i] - n1[i - 1] - n2[j + 1] - ns2 * 2;
            const int ns1 = sum - cv - ns2;
            if (ns1 > 0 && ns2 > 0) {
                // two divisions per candidate pair dominate the run time: skip the pairs that cannot win.  The test is
                // conservative (margin 1e-9 against a rounding error of 1e-15), so every pair that could satisfy
                // `ave < best2` still goes through the reference's own expression below
                const double num = ((double)cv) * sum;
                if (num > best2 * (double)ns1 * (double)ns2 * 1.000000001) continue;
                const double ave = num / ns1 / ns2;
                if (ave < best2) {
                    best2 = ave;
                    cuts1 = i;
                    cuts2 = j;
                }
            }
        }
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
    a.clear();
    a.shrink_to_fit();
    inner.clear();
    inner.shrink_to_fit();
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

int predict_impl(int32_t n_res, const int32_t* ci, const int32_t* cj, const float* prob, int64_t n_contacts,
                 double cut1, double cut2, std::string& text, int32_t* n_domains) {
    if (n_res <= 0 || n_contacts < 0 || (n_contacts > 0 && (!ci || !cj || !prob))) return RECCUT_ERR_INVALID;
    Ctx cx;
    cx.n0 = n_res;
    cx.cut1 = cut1;
    cx.cut2 = cut2;
    std::vector<Domain> doms;
    Domain whole(1, Seg(0, n_res - 1));
    if (n_res >= kMinSize) {
        cx.w.assign((size_t)n_res * n_res, 0);
        for (int64_t k = 0; k < n_contacts; ++k) {
            const int i = ci[k], j = cj[k];
            if (i < 0 || j < 0 || i >= n_res || j >= n_res) return RECCUT_ERR_INVALID;
            const int wgt = contact_weight(prob[k]);  // the value as the .ce file carries it
            cx.w[(size_t)i * n_res + j] = wgt;
            cx.w[(size_t)j * n_res + i] = wgt;
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

}  // extern "C"
