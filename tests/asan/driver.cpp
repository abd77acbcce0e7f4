// Drives the C ABI of libdctfp's HOST code over seeded random batches under AddressSanitizer, against the HIP stand-in of
// hip_stub.cpp (no GPU, no kernels: see there).  What it covers: every table dctfp_quantize builds (jobs, pieces, walks,
// runs, cosine-table list), the staging / device copies it sizes, fused-group detection, the split at giant domains, the
// chunk plan of the two-kernel path, the option surface, the cosine-table cache (failure injection, arena restart).
// Usage: driver [rounds] [seed]
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "dctfp.h"

extern "C" unsigned long dctfp_stub_counter(int which);

// ---- allocation-failure hook: the N-th operator new from now on throws std::bad_alloc (once).  Nothing may throw across
// the C ABI -- an exception that left dctfp_quantize would end this process in std::terminate (SIGABRT on the calling
// thread) -- so the call has to come back with DCTFP_ERR_NOMEM (or have succeeded, if it needed fewer allocations).
#include <new>
static long g_fail_after = -1;   // < 0: never
static long g_failed = 0;
static void* hooked_alloc(size_t n, size_t align) {
    if (g_fail_after >= 0 && g_fail_after-- == 0) {
        ++g_failed;
        throw std::bad_alloc();
    }
    void* p = align > 16 ? aligned_alloc(align, (n + align - 1) / align * align) : malloc(n ? n : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
void* operator new(size_t n) { return hooked_alloc(n, 0); }
void* operator new[](size_t n) { return hooked_alloc(n, 0); }
void* operator new(size_t n, std::align_val_t a) { return hooked_alloc(n, (size_t)a); }
void* operator new[](size_t n, std::align_val_t a) { return hooked_alloc(n, (size_t)a); }
void* operator new(size_t n, const std::nothrow_t&) noexcept { try { return hooked_alloc(n, 0); } catch (...) { return nullptr; } }
void* operator new[](size_t n, const std::nothrow_t&) noexcept { try { return hooked_alloc(n, 0); } catch (...) { return nullptr; } }
void* operator new(size_t n, std::align_val_t a, const std::nothrow_t&) noexcept { try { return hooked_alloc(n, (size_t)a); } catch (...) { return nullptr; } }
void* operator new[](size_t n, std::align_val_t a, const std::nothrow_t&) noexcept { try { return hooked_alloc(n, (size_t)a); } catch (...) { return nullptr; } }
void operator delete(void* p) noexcept { free(p); }
void operator delete[](void* p) noexcept { free(p); }
void operator delete(void* p, size_t) noexcept { free(p); }
void operator delete[](void* p, size_t) noexcept { free(p); }
void operator delete(void* p, std::align_val_t) noexcept { free(p); }
void operator delete[](void* p, std::align_val_t) noexcept { free(p); }
void operator delete(void* p, size_t, std::align_val_t) noexcept { free(p); }
void operator delete[](void* p, size_t, std::align_val_t) noexcept { free(p); }
void operator delete(void* p, const std::nothrow_t&) noexcept { free(p); }
void operator delete[](void* p, const std::nothrow_t&) noexcept { free(p); }
void operator delete(void* p, std::align_val_t, const std::nothrow_t&) noexcept { free(p); }
void operator delete[](void* p, std::align_val_t, const std::nothrow_t&) noexcept { free(p); }

#define CHECK(expr)                                                                         \
    do {                                                                                    \
        const int rc_ = (expr);                                                             \
        if (rc_ != DCTFP_OK) {                                                              \
            fprintf(stderr, "%s -> %d: %s\n", #expr, rc_, dctfp_last_error());              \
            return 1;                                                                       \
        }                                                                                   \
    } while (0)

// ---- the entry points either side of dctfp_quantize: argument checks, table builds (top-k jobs and stripes, sorting-network
// groups, window jobs by level and in the one-launch form, candidate scratch of the row select), the host-only parsers.  The
// kernels themselves are no-ops under the stub; what runs is every line of host code, with allocation failures injected.
static int other_entry_points(dctfp_ctx* ctx, std::mt19937_64& rng, int rounds, int* n_calls, int* n_expected) {
    auto uni = [&](int lo, int hi) { return (int)(lo + rng() % (uint64_t)(hi - lo + 1)); };
    auto ok_or_expected = [&](int rc, const char* what) {
        ++*n_calls;
        if (rc == DCTFP_OK) return true;
        if (rc == DCTFP_ERR_NOMEM || rc == DCTFP_ERR_SHAPE || rc == DCTFP_ERR_INVALID || rc == DCTFP_ERR_LIMIT || rc == DCTFP_ERR_UNSUPPORTED) { ++*n_expected; return true; }
        fprintf(stderr, "%s -> %d: %s\n", what, rc, dctfp_last_error());
        return false;
    };
    for (int round = 0; round < rounds; ++round) {
        const bool starve = uni(0, 4) == 0;
        // ---- domain strings -> pieces (host only)
        {
            const int n_seq = uni(1, 40);
            std::string text;
            std::vector<int32_t> counts(n_seq);
            std::vector<int64_t> rows(n_seq);
            int64_t n_str = 0, commas = 0;
            for (int s = 0; s < n_seq; ++s) {
                rows[s] = uni(1, 900);
                counts[s] = uni(0, 6);
                for (int d = 0; d < counts[s]; ++d, ++n_str) {
                    if (n_str) text += '\n';
                    const int pieces = uni(1, 4);
                    for (int q = 0; q < pieces; ++q) {
                        if (q) { text += ','; ++commas; }
                        const int kind = uni(0, 19);
                        if (kind == 0) text += "x-7";                                   // not digits: left to the caller's parser
                        else if (kind == 1) text += std::to_string(uni(0, 5)) + "-";    // nothing behind the dash
                        else if (kind == 2) text += "0-" + std::to_string(uni(1, 50));  // the [-1:end] slice
                        else text += std::to_string(uni(1, 1000)) + "-" + std::to_string(uni(1, 1100));
                    }
                }
            }
            const int64_t cap = n_str + commas + (uni(0, 9) == 0 ? -1 : 2);               // (sometimes too small: an error, not a write)
            std::vector<dctfp_piece> pieces((size_t)std::max<int64_t>(cap, 1));
            std::vector<int32_t> str_row((size_t)std::max<int64_t>(n_str, 1));
            std::vector<int64_t> str_len((size_t)std::max<int64_t>(n_str, 1));
            std::vector<uint8_t> changed((size_t)std::max<int64_t>(n_str, 1));
            std::vector<char> keys(text.size() + (size_t)n_str + 1);
            int64_t n_pieces = 0, key_len = 0, n_dom = 0, n_other = 0;
            if (starve) g_fail_after = uni(0, 6);
            const int rc = dctfp_build_pieces(text.data(), (int64_t)text.size(), counts.data(), rows.data(), n_seq, pieces.data(), std::max<int64_t>(cap, 0),
                                              &n_pieces, str_row.data(), str_len.data(), changed.data(), keys.data(), (int64_t)keys.size(), &key_len,
                                              &n_dom, &n_other);
            g_fail_after = -1;
            if (!ok_or_expected(rc, "dctfp_build_pieces")) return 1;
            if (rc == DCTFP_OK && (n_pieces > std::max<int64_t>(cap, 0) || key_len > (int64_t)keys.size() || n_dom > n_str)) {
                fprintf(stderr, "dctfp_build_pieces: counts beyond the buffers\n");
                return 1;
            }
        }
        // ---- dctfp_reccut_pieces: well-formed records of the cutter, and garbage in their place (exact-size buffers: any write past what
        // the header promises is a report)
        {
            const int n_prot = uni(1, 30);
            std::vector<int64_t> rows(n_prot), enc_off(n_prot + 1, 0);
            for (int p = 0; p < n_prot; ++p) {
                rows[p] = uni(0, 12) == 0 ? uni(0, 21) : uni(22, 900);
                enc_off[p + 1] = enc_off[p] + std::max<int64_t>(uni(0, 15) == 0 ? uni(0, 3) : dctfp_reccut_room((int32_t)rows[p]), 0);
            }
            std::vector<int32_t> enc((size_t)std::max<int64_t>(enc_off[n_prot], 1), -1);
            int64_t segs = 0;
            for (int p = 0; p < n_prot; ++p) {
                int32_t* e = enc.data() + enc_off[p];
                const int64_t room = enc_off[p + 1] - enc_off[p];
                const int kind = uni(0, 9);
                if (kind == 0 || room < 4) {           // garbage
                    for (int64_t q = 0; q < room; ++q) e[q] = uni(0, 3) == 0 ? uni(-5, 2000000) : uni(-2, 40);
                    continue;
                }
                const int L = (int)std::max<int64_t>(rows[p], 1);
                int64_t q = 1;
                int nd = 0;
                const int want = uni(1, 6);
                for (int d = 0; d < want; ++d) {
                    const int ns = uni(1, 3);
                    if (q + 1 + 2 * ns > room) break;
                    e[q++] = ns;
                    for (int sg = 0; sg < ns; ++sg) {
                        const int a = uni(0, L - 1), b2 = kind == 1 ? uni(0, L + 3) : uni(a, L - 1);   // (kind 1: some segments outside / reversed)
                        e[q++] = a;
                        e[q++] = b2;
                    }
                    ++nd;
                    segs += ns;
                }
                e[0] = nd;
            }
            const int64_t piece_cap = uni(0, 12) == 0 ? uni(0, 3) : enc_off[n_prot] / 2 + n_prot + 1;
            const int64_t text_cap = uni(0, 12) == 0 ? uni(0, 40) : 12 * enc_off[n_prot] + 32 * n_prot + 64;
            std::vector<dctfp_piece> pieces((size_t)std::max<int64_t>(piece_cap, 1));
            std::vector<char> text((size_t)std::max<int64_t>(text_cap, 1));
            std::vector<int32_t> counts(n_prot);
            int64_t text_len = 0, n_pieces = 0, n_dom = 0, n_undone = 0;
            if (starve) g_fail_after = uni(0, 4);
            const int rc = dctfp_reccut_pieces(n_prot, enc.data(), enc_off.data(), rows.data(), text.data(), text_cap, &text_len, counts.data(),
                                               pieces.data(), piece_cap, &n_pieces, &n_dom, &n_undone);
            g_fail_after = -1;
            if (!ok_or_expected(rc, "dctfp_reccut_pieces")) return 1;
            if (rc == DCTFP_OK && (n_pieces > piece_cap || text_len > text_cap || n_undone > n_prot || n_dom > n_pieces)) {
                fprintf(stderr, "dctfp_reccut_pieces: counts beyond the buffers\n");
                return 1;
            }
            (void)segs;
        }
        // ---- contact top-k + order: short and long proteins (one workgroup / stripes), every t
        {
            const int n_prot = uni(1, 40);
            const double t = (double[]){0.0, 0.5, 2.6, 6.0, 40.0}[uni(0, 4)];
            std::vector<int32_t> n_res(n_prot);
            std::vector<int64_t> ld(n_prot), offs(n_prot + 1, 0);
            std::vector<std::vector<float>> maps(n_prot);
            std::vector<const void*> ptrs(n_prot);
            for (int p = 0; p < n_prot; ++p) {
                const int c = uni(0, 19);
                n_res[p] = c == 0 ? uni(0, 5) : (c == 1 ? uni(1500, 2100) : uni(6, 300));
                ld[p] = n_res[p] + (uni(0, 3) == 0 ? 3 : 0);
                maps[p].assign((size_t)std::max<int64_t>(1, (int64_t)n_res[p] * ld[p]), 0.5f);
                ptrs[p] = maps[p].data();
                offs[p + 1] = offs[p] + dctfp_contact_count(n_res[p], t);
            }
            const size_t total = (size_t)std::max<int64_t>(offs[n_prot], 1);
            std::vector<int32_t> oi(total), oj(total), on(n_prot);
            std::vector<float> ov(total);
            std::vector<uint8_t> sorted(n_prot);
            if (starve) g_fail_after = uni(0, 10);
            int rc = dctfp_contact_topk(ctx, ptrs.data(), ld.data(), n_res.data(), n_prot, t, oi.data(), oj.data(), ov.data(), offs.data(), on.data(), nullptr);
            g_fail_after = -1;
            if (!ok_or_expected(rc, "dctfp_contact_topk")) return 1;
            if (starve) g_fail_after = uni(0, 10);
            rc = dctfp_contact_sort(ctx, ptrs.data(), ld.data(), n_res.data(), n_prot, t, oi.data(), oj.data(), ov.data(), offs.data(), sorted.data(), nullptr);
            g_fail_after = -1;
            if (!ok_or_expected(rc, "dctfp_contact_sort")) return 1;
            // ---- the domain cutter on the selected contacts: job table, scratch sizes, the three size classes on their streams
            std::vector<int64_t> enc_off(n_prot + 1, 0);
            for (int p = 0; p < n_prot; ++p) enc_off[p + 1] = enc_off[p] + dctfp_reccut_room(n_res[p]) - (uni(0, 30) == 0 ? 100 : 0);   // (sometimes no room: an error)
            std::vector<int32_t> enc((size_t)std::max<int64_t>(enc_off[n_prot], 1));
            if (starve) g_fail_after = uni(0, 8);
            rc = enc_off[n_prot] >= 0 ? dctfp_reccut(ctx, n_res.data(), n_prot, oi.data(), oj.data(), ov.data(), offs.data(), 0.08, 0.07, enc.data(), enc_off.data(), nullptr)
                                      : DCTFP_OK;
            g_fail_after = -1;
            if (!ok_or_expected(rc, "dctfp_reccut")) return 1;
        }
        // ---- Fingerprint.quantize over windows: the geometry of every piece (one window's rows / the rows two windows share), the
        // refusals, the walk kernel's two-source builds (the stub touches both rows of every shared piece)
        {
            const int D = (int[]){640, 1280, 2560, 96, 1280}[uni(0, 4)];
            const int dtype = uni(0, 7) == 0 ? DCTFP_F16 : DCTFP_F32;
            const size_t esz = dtype == DCTFP_F32 ? 4 : 2;
            const int overlap = (int[]){200, 200, 30}[uni(0, 2)], maxlen = overlap == 200 ? (int[]){500, 500, 400, 300}[uni(0, 3)] : uni(60, 90);
            const int n_seq = uni(0, 3) == 0 ? uni(1, 6) : uni(130, 220);
            const int n_layers = uni(1, 2);
            const int64_t ld = D + (uni(0, 4) == 0 ? 8 : 0);
            std::vector<int64_t> seq_win(n_seq + 1, 0), rows_of(n_seq);
            std::vector<int32_t> win_rows;
            std::vector<dctfp_piece> pieces;
            int32_t n_domains = 0;
            for (int s = 0; s < n_seq; ++s) {
                const int L = uni(0, 40) == 0 ? uni(1, overlap) : uni(3, 6 * maxlen);
                int n = 0;
                if (L <= maxlen) {
                    win_rows.push_back(L);
                    n = 1;
                } else {
                    for (int i = 0; i < L; i += maxlen - overlap) {
                        const int len = std::min(maxlen, L - i);
                        if (len > overlap) { win_rows.push_back(len); ++n; }
                    }
                }
                if (uni(0, 60) == 0 && n > 1) win_rows.back() = uni(1, overlap);          // not longer than the overlap: ERR_SHAPE
                seq_win[s + 1] = seq_win[s] + n;
            }
            int rc = dctfp_stitch_sizes(win_rows.data(), seq_win.data(), n_seq, overlap, 0, rows_of.data());
            if (!ok_or_expected(rc, "dctfp_stitch_sizes (windows)")) return 1;
            const bool geometry_ok = rc == DCTFP_OK;
            for (int s = 0; s < n_seq; ++s) {
                const int64_t L = geometry_ok ? rows_of[s] : 50;
                auto add = [&](int64_t start, int64_t rows, int32_t dom) { pieces.push_back({start, (int32_t)rows, dom, s, 0}); };
                const int kind = uni(0, 9);
                if (kind < 5 || L < 12) add(0, L, n_domains++);
                else if (kind < 9) {                                                    // parts tiling the sequence (cuts anywhere) + the whole
                    const int k = (int)std::min<int64_t>(uni(2, 6), L / 4);
                    std::vector<int64_t> cut{0};
                    for (int i = 1; i < k; ++i) cut.push_back(cut.back() + std::max<int64_t>(3, (L - cut.back()) / (k - i + 1) + uni(-1, 1)));
                    cut.push_back(L);
                    bool ok = true;
                    for (size_t i = 1; i < cut.size(); ++i) ok = ok && cut[i] - cut[i - 1] >= 3;
                    if (!ok) { add(0, L, n_domains++); continue; }
                    for (int i = 0; i < k; ++i) add(cut[i], cut[i + 1] - cut[i], n_domains++);
                    add(0, L, n_domains++);
                } else {
                    const int64_t a = uni(0, (int)(L / 2)), b = std::min<int64_t>(L + (uni(0, 20) == 0 ? 5 : 0), a + uni(3, (int)L));   // (sometimes past the end: ERR_INVALID)
                    add(a, b - a, n_domains++);
                }
            }
            const size_t n_win = win_rows.size();
            std::vector<std::vector<char*>> data(n_layers, std::vector<char*>(n_win));
            std::vector<std::vector<const void*>> wp(n_layers, std::vector<const void*>(n_win));
            for (int l = 0; l < n_layers; ++l)
                for (size_t w = 0; w < n_win; ++w) {
                    const size_t bytes = ((size_t)std::max(win_rows[w] - 1, 0) * ld + D) * esz;
                    data[l][w] = (char*)aligned_alloc(64, (bytes + 63) / 64 * 64);
                    wp[l][w] = data[l][w];
                }
            std::vector<dctfp_layer> layers(n_layers);
            int32_t off = 0;
            static const int wq[][2] = {{3, 80}, {3, 80}, {3, 80}, {5, 44}};
            const int* q = wq[uni(0, 3)];
            for (int l = 0; l < n_layers; ++l) {
                layers[l] = {wp[l].data(), ld, D, dtype, q[0], q[1], off, 0};
                off += q[0] * q[1];
            }
            std::vector<int8_t> out((size_t)std::max(n_domains, 1) * off);
            const int path = uni(0, 2);
            if (dctfp_set_option(ctx, "path", path) != DCTFP_OK) return 1;
            if (starve) g_fail_after = uni(0, 30);
            rc = dctfp_quantize_windows(ctx, layers.data(), n_layers, n_seq, seq_win.data(), win_rows.data(), overlap, pieces.data(), (int64_t)pieces.size(),
                                        n_domains, out.data(), off, nullptr);
            g_fail_after = -1;
            if (dctfp_set_option(ctx, "path", 0) != DCTFP_OK) return 1;
            for (int l = 0; l < n_layers; ++l)
                for (size_t w = 0; w < n_win; ++w) free(data[l][w]);
            if (!ok_or_expected(rc, "dctfp_quantize_windows")) return 1;
        }
        // ---- window stitching: geometry, the one-launch form, the sequential form, malformed windows
        {
            const int square = uni(0, 3) == 0;
            const int n_seq = uni(1, 30), n_cols = (int[]){32, 50, 64, 1280}[uni(0, 3)];
            const int step = square ? uni(20, 60) : (int[]){200, 50, 7}[uni(0, 2)];
            std::vector<int64_t> seq_win(n_seq + 1, 0), win_ld;
            std::vector<int32_t> win_rows;
            for (int s = 0; s < n_seq; ++s) {
                const int n = uni(1, 6);
                for (int w = 0; w < n; ++w) {
                    const int c = uni(0, 29);
                    int rows = square ? uni(step, 2 * step + 10) : uni(step + 1, 3 * step + 50);
                    if (c == 0) rows = uni(0, step);                                   // not longer than the overlap: ERR_SHAPE
                    if (c == 1 && !square) rows = uni(step + 1, 2 * step - 1);         // three windows meet: the sequential form
                    win_rows.push_back(rows);
                    win_ld.push_back((square ? rows : n_cols) + (uni(0, 4) == 0 ? 4 : 0));
                }
                seq_win[s + 1] = seq_win[s] + n;
            }
            std::vector<int64_t> sizes(n_seq);
            int rc = dctfp_stitch_sizes(win_rows.data(), seq_win.data(), n_seq, step, square, sizes.data());
            if (!ok_or_expected(rc, "dctfp_stitch_sizes")) return 1;
            if (rc == DCTFP_OK) {
                std::vector<std::vector<float>> wins(win_rows.size()), outs(n_seq);
                std::vector<const void*> wp(win_rows.size());
                std::vector<void*> dp(n_seq);
                std::vector<int64_t> dld(n_seq);
                for (size_t w = 0; w < win_rows.size(); ++w) {
                    wins[w].assign((size_t)std::max<int64_t>(1, win_rows[w] * win_ld[w]), 1.0f);
                    wp[w] = wins[w].data();
                }
                for (int s = 0; s < n_seq; ++s) {
                    dld[s] = square ? sizes[s] : n_cols;
                    outs[s].assign((size_t)std::max<int64_t>(1, sizes[s] * dld[s]), 0.0f);
                    dp[s] = outs[s].data();
                }
                if (starve) g_fail_after = uni(0, 8);
                rc = dctfp_stitch_sequences(ctx, wp.data(), win_rows.data(), win_ld.data(), seq_win.data(), n_seq, dp.data(), dld.data(), n_cols, step, square, nullptr);
                g_fail_after = -1;
                if (!ok_or_expected(rc, "dctfp_stitch_sequences")) return 1;
                // the same windows as explicit jobs (what dctfp_stitch takes), one of them malformed now and then
                std::vector<dctfp_stitch_job> jobs;
                for (int s = 0; s < n_seq; ++s)
                    for (int64_t w = seq_win[s]; w < seq_win[s + 1]; ++w) {
                        dctfp_stitch_job j{wp[w], dp[s], win_ld[w], dld[s], win_rows[w], w > seq_win[s] ? std::min(step, win_rows[w]) : 0, (int32_t)(w - seq_win[s]), 0};
                        if (uni(0, 200) == 0) j.n_avg = j.n_rows + 1;
                        jobs.push_back(j);
                    }
                if (starve) g_fail_after = uni(0, 8);
                rc = dctfp_stitch(ctx, jobs.data(), (int64_t)jobs.size(), n_cols, square, nullptr);
                g_fail_after = -1;
                if (!ok_or_expected(rc, "dctfp_stitch")) return 1;
            }
        }
        // ---- similarity consumers: every alignment class of the L1 matrix, block minima, the row select in one segment,
        // in several (candidate scratch), with k beyond the register kernel
        {
            const int na = uni(1, 300), nb = uni(1, 300), d = (int[]){480, 470, 16, 3, 129}[uni(0, 4)];
            const int64_t lda = d + (int[]){0, 0, 10, 16}[uni(0, 3)], ldb = d + (int[]){0, 0, 6, 16}[uni(0, 3)];
            std::vector<int8_t> a((size_t)na * lda + 16), b((size_t)nb * ldb + 16);
            std::vector<int32_t> dist((size_t)na * nb);
            if (!ok_or_expected(dctfp_l1_matrix(ctx, a.data() + uni(0, 1), na, lda, b.data(), nb, ldb, d, dist.data(), nb, nullptr), "dctfp_l1_matrix")) return 1;
            std::vector<int64_t> ia{0}, ib{0};
            while (ia.back() < na) ia.push_back(std::min<int64_t>(na, ia.back() + uni(0, 7)));
            while (ib.back() < nb) ib.push_back(std::min<int64_t>(nb, ib.back() + uni(0, 7)));
            std::vector<int32_t> mn((ia.size() - 1) * (ib.size() - 1) + 1), last(mn.size());
            if (!ok_or_expected(dctfp_block_min(ctx, dist.data(), nb, ia.data(), (int64_t)ia.size() - 1, ib.data(), (int64_t)ib.size() - 1, mn.data(), last.data(),
                                                nullptr), "dctfp_block_min")) return 1;
            const int64_t n_rows = uni(1, 20), n_cols = (int64_t[]){1, 50, 40960, 40961, 130000}[uni(0, 4)];
            const int32_t k = (int32_t)std::min<int64_t>(n_cols, (int64_t[]){1, 100, 1024, 1025, 5000}[uni(0, 4)]);
            std::vector<int32_t> wide((size_t)n_rows * n_cols), val((size_t)n_rows * k), idx((size_t)n_rows * k);
            if (starve) g_fail_after = uni(0, 4);
            const int rc = dctfp_row_select(ctx, wide.data(), n_rows, n_cols, n_cols, k, val.data(), idx.data(), nullptr);
            g_fail_after = -1;
            if (!ok_or_expected(rc, "dctfp_row_select")) return 1;
            if (!ok_or_expected(dctfp_row_order(ctx, val.data(), idx.data(), n_rows, k, nullptr), "dctfp_row_order")) return 1;   // (k > 1024: the limit error)
        }
    }
    return 0;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 300;
    std::mt19937_64 rng(argc > 2 ? strtoull(argv[2], nullptr, 10) : 12345);
    auto uni = [&](int lo, int hi) { return (int)(lo + rng() % (uint64_t)(hi - lo + 1)); };
    dctfp_ctx* ctx = nullptr;
    CHECK(dctfp_create(0, &ctx));
    int n_calls = 0, n_errors_expected = 0;
    for (int round = 0; round < rounds; ++round) {
        // ---- shape of the call
        static const int widths[] = {96, 512, 640, 768, 1000, 1280, 2048, 2560, 324, 2564};
        const int D = widths[uni(0, 9)];
        const int dtype = uni(0, 9) < 6 ? DCTFP_F32 : (uni(0, 2) == 0 ? DCTFP_F64 : (uni(0, 1) ? DCTFP_F16 : DCTFP_BF16));
        const size_t esz = dtype == DCTFP_F64 ? 8 : (dtype == DCTFP_F32 ? 4 : 2);
        const int n_layers = uni(1, 3);
        const int size_class = uni(0, 9);
        const int n_seq = size_class < 3 ? uni(1, 3) : (size_class < 7 ? uni(20, 120) : uni(200, 500));
        const int64_t ld = D + (uni(0, 3) == 0 ? 8 : 0);
        std::vector<int64_t> seq_rows(n_seq);
        std::vector<dctfp_piece> pieces;
        int32_t n_domains = 0;
        const bool short_jobs = uni(0, 3) == 0;
        for (int s = 0; s < n_seq; ++s) {
            int64_t L = short_jobs ? uni(3, 40) : uni(3, 700);
            if (uni(0, 60) == 0) L = uni(8193, 9000);                       // a giant
            seq_rows[s] = L;
            const int kind = uni(0, 9);
            auto add = [&](int64_t start, int64_t rows, int32_t dom) { pieces.push_back({start, (int32_t)rows, dom, s, 0}); };
            if (kind < 4 || L < 9) {
                add(0, L, n_domains++);                                       // whole sequence
            } else if (kind < 8) {                                            // parts tiling the protein + whole (fused group)
                const int k = std::min<int64_t>(uni(2, 7), L / 3);
                std::vector<int64_t> cut{0};
                for (int i = 1; i < k; ++i) cut.push_back(cut.back() + std::max<int64_t>(3, (L - cut.back()) / (k - i + 1) + uni(-2, 2)));
                cut.push_back(L);
                bool ok = true;
                for (size_t i = 1; i < cut.size(); ++i) ok = ok && cut[i] - cut[i - 1] >= 3;
                if (!ok) { add(0, L, n_domains++); continue; }
                const bool disc = k >= 3 && uni(0, 2) == 0;                   // first + last part as one discontinuous domain
                if (disc) {
                    add(cut[0], cut[1] - cut[0], n_domains);
                    add(cut[k - 1], cut[k] - cut[k - 1], n_domains++);
                    for (int i = 1; i < k - 1; ++i) add(cut[i], cut[i + 1] - cut[i], n_domains++);
                } else {
                    for (int i = 0; i < k; ++i) add(cut[i], cut[i + 1] - cut[i], n_domains++);
                }
                add(0, L, n_domains++);
            } else {                                                          // an inner window, and a clipped two-piece domain
                const int64_t a = uni(0, (int)(L / 3)), b = std::min<int64_t>(L, a + uni(3, (int)std::max<int64_t>(3, L / 2)));
                if (b - a >= 3) add(a, b - a, n_domains++);
                else add(0, L, n_domains++);
                if (L >= 12 && uni(0, 1)) {
                    add(0, 3, n_domains);
                    add(L - 4, 4, n_domains++);
                }
            }
        }
        // ---- buffers ("device" memory is host memory under the stub: ASan checks what the kernels would touch)
        std::vector<std::vector<char*>> data(n_layers, std::vector<char*>(n_seq));
        std::vector<std::vector<const void*>> ptrs(n_layers, std::vector<const void*>(n_seq));
        const bool one_tensor = uni(0, 1);                                    // all sequences back to back, or one tensor each
        std::vector<char*> big(n_layers, nullptr);
        int64_t total = 0;
        for (int s = 0; s < n_seq; ++s) total += seq_rows[s];
        for (int l = 0; l < n_layers; ++l) {
            if (one_tensor) {
                big[l] = (char*)aligned_alloc(64, ((size_t)total * ld * esz + 63) / 64 * 64);
                int64_t r0 = 0;
                for (int s = 0; s < n_seq; ++s) { ptrs[l][s] = big[l] + (size_t)r0 * ld * esz; r0 += seq_rows[s]; }
            } else {
                for (int s = 0; s < n_seq; ++s) {
                    // (the last row of a tensor is D wide, not ld: what torch allocates for a view with a padded stride)
                    const size_t bytes = ((size_t)(seq_rows[s] - 1) * ld + D) * esz;
                    data[l][s] = (char*)aligned_alloc(64, (bytes + 63) / 64 * 64);
                    ptrs[l][s] = data[l][s];
                }
            }
        }
        std::vector<dctfp_layer> layers(n_layers);
        int32_t off = 0;
        static const int qd[][2] = {{3, 80}, {3, 80}, {3, 80}, {5, 44}, {3, 85}, {3, 65}, {2, 80}, {8, 128}, {1, 80}, {3, 1}, {4, 64}};
        const int qsel = uni(0, 9) < 6 ? 0 : uni(0, 10);                        // mostly the reference's [3, 80]
        for (int l = 0; l < n_layers; ++l) {
            const int* q = qd[(uni(0, 3) == 0) ? uni(0, 10) : qsel];
            layers[l] = {ptrs[l].data(), ld, D, dtype, q[0], q[1], off, 0};
            off += q[0] * q[1];
        }
        const int64_t out_stride = off + (uni(0, 2) == 0 ? 16 : 0);
        std::vector<int8_t> out((size_t)n_domains * out_stride);
        // ---- options
        static const char* names[] = {"path", "fuse", "ab_run_jobs", "ab_longest_first", "workspace_mb", "overlap", "ab_group", "pack_y", "small_b_jobs", "a_waves", "gen_fuse"};
        std::vector<std::pair<const char*, int64_t>> saved;
        for (int i = 0; i < uni(0, 3); ++i) {
            const char* nm = names[uni(0, 10)];
            int64_t v = 0, old = 0;
            if (!strcmp(nm, "path")) v = uni(0, 2);
            else if (!strcmp(nm, "fuse") || !strcmp(nm, "pack_y") || !strcmp(nm, "gen_fuse")) v = uni(0, 1);
            else if (!strcmp(nm, "ab_run_jobs")) v = (int64_t[]){0, 1, 4, 16, 64}[uni(0, 4)];
            else if (!strcmp(nm, "ab_longest_first")) v = uni(0, 2);
            else if (!strcmp(nm, "workspace_mb")) v = (int64_t[]){16, 64, 4096}[uni(0, 2)];
            else if (!strcmp(nm, "overlap")) v = uni(1, 8);
            else if (!strcmp(nm, "ab_group")) v = (int64_t[]){0, 3, 4}[uni(0, 2)];
            else if (!strcmp(nm, "small_b_jobs")) v = (int64_t[]){0, 512, 1 << 20}[uni(0, 2)];
            else v = (int64_t[]){0, 2, 4, 8, 16}[uni(0, 4)];
            CHECK(dctfp_get_option(ctx, nm, &old));
            CHECK(dctfp_set_option(ctx, nm, v));
            saved.push_back({nm, old});
        }
        if (uni(0, 40) == 0) CHECK(dctfp_set_option(ctx, "basis_cap_kb", 64));
        const bool inject = uni(0, 25) == 0;
        if (inject) CHECK(dctfp_set_option(ctx, "test_fail_once", 1));
        // ---- the call (twice: the second is served from the caches); one call in six with a failing allocation inside
        const bool starve = uni(0, 5) == 0;
        for (int rep = 0; rep < 2; ++rep) {
            if (starve) g_fail_after = uni(0, rep == 0 ? 40 : 12);
            const int rc = dctfp_quantize(ctx, layers.data(), n_layers, n_seq, seq_rows.data(), pieces.data(), (int64_t)pieces.size(),
                                          n_domains, out.data(), out_stride, nullptr);
            g_fail_after = -1;
            ++n_calls;
            if (rc == DCTFP_ERR_SHAPE || rc == DCTFP_ERR_NOMEM) { ++n_errors_expected; continue; }   // D < m / L < n / the injected failure
            if (rc != DCTFP_OK) {
                fprintf(stderr, "round %d: dctfp_quantize -> %d: %s\n", round, rc, dctfp_last_error());
                return 1;
            }
        }
        CHECK(dctfp_set_option(ctx, "test_fail_once", 0));
        CHECK(dctfp_set_option(ctx, "basis_cap_kb", 1 << 20));
        for (auto& kv : saved) CHECK(dctfp_set_option(ctx, kv.first, kv.second));
        for (int l = 0; l < n_layers; ++l) {
            free(big[l]);
            for (int s = 0; s < n_seq; ++s) free(data[l][s]);
        }
    }
    int n_other_calls = 0, n_other_expected = 0;
    if (other_entry_points(ctx, rng, std::max(20, rounds / 3), &n_other_calls, &n_other_expected)) return 1;
    printf("the other entry points: %d calls (%d ended in an expected error)\n", n_other_calls, n_other_expected);
    CHECK(dctfp_destroy(ctx));
    printf("allocation failures injected and reported as DCTFP_ERR_NOMEM: %ld\n", g_failed);
    printf("asan driver: %d calls over %d rounds (%d ended in an expected error), %lu walk-kernel launches, %lu stage-A launches, "
           "%lu jobs walked by the table emulation: no memory error\n", n_calls, rounds, n_errors_expected, dctfp_stub_counter(0),
           dctfp_stub_counter(1), dctfp_stub_counter(2));
    return 0;
}
