// Drives the C ABI of libdctfp's HOST code over seeded random batches under AddressSanitizer, against the HIP stand-in of
// hip_stub.cpp (no GPU, no kernels: see there).  What it covers: every table dctfp_quantize builds (jobs, pieces, walks,
// runs, cosine-table list), the staging / device copies it sizes, fused-group detection, the split at giant domains, the
// chunk plan of the two-kernel path, the option surface, the cosine-table cache (failure injection, arena restart).
// Usage: driver [rounds] [seed]
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
        static const char* names[] = {"path", "fuse", "ab_run_jobs", "ab_longest_first", "workspace_mb", "overlap", "ab_group", "pack_y", "small_b_jobs", "a_waves"};
        std::vector<std::pair<const char*, int64_t>> saved;
        for (int i = 0; i < uni(0, 3); ++i) {
            const char* nm = names[uni(0, 9)];
            int64_t v = 0, old = 0;
            if (!strcmp(nm, "path")) v = uni(0, 2);
            else if (!strcmp(nm, "fuse") || !strcmp(nm, "pack_y")) v = uni(0, 1);
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
    CHECK(dctfp_destroy(ctx));
    printf("allocation failures injected and reported as DCTFP_ERR_NOMEM: %ld\n", g_failed);
    printf("asan driver: %d calls over %d rounds (%d ended in an expected error), %lu walk-kernel launches, %lu stage-A launches, "
           "%lu jobs walked by the table emulation: no memory error\n", n_calls, rounds, n_errors_expected, dctfp_stub_counter(0),
           dctfp_stub_counter(1), dctfp_stub_counter(2));
    return 0;
}
