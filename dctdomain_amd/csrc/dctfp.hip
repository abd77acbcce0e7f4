// libdctfp.so -- host side of the C ABI declared in include/dctfp.h.
// Builds the job tables of a ragged batch, owns the cosine bases and the float64
// scratch, and launches the gfx950 kernels of kernels.hip.h.  No CPU compute path:
// every entry point needs the GPU.
#include "dctfp.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "launch.h"   // (kernels.hip.h, the parameter blocks and the launchers of the kernel families built in their own units)
#include "reccut_kernel.hip.h"   // (CutJob, table sizes; the kernel itself is instantiated in k_reccut.hip)

using namespace dctfp;
using namespace dctfp_host;

namespace {

// Message of the calling thread's last failure.  A fixed buffer: reporting an out-of-memory condition must not allocate.
thread_local char g_err[512] = "";

void set_err(const char* msg) {
    snprintf(g_err, sizeof g_err, "%s", msg);
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail(DCTFP_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return DCTFP_OK;
        if (p) {
            // (growth is rare; kernels of ANOTHER stream may still be reading the old block -- a flush's cutter on its side stream --
            //  and hipFree is only known to wait for the blocking streams)
            (void)hipDeviceSynchronize();
            hipError_t e = hipFree(p);  // device-synchronising: earlier launches are done with it
            p = nullptr;
            cap = 0;
            if (e != hipSuccess) return fail(DCTFP_ERR_HIP, "hipFree: %s", hipGetErrorString(e));
        }
        size_t want = bytes + bytes / 4;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            want = bytes;
            e = hipMalloc(&p, want);
        }
        if (e != hipSuccess) {
            p = nullptr;
            return fail(DCTFP_ERR_NOMEM, "hipMalloc(%zu bytes): %s", want, hipGetErrorString(e));
        }
        cap = want;
        return DCTFP_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct Staging {  // pinned host buffer + the event after its last H2D copy (or after its last reader on the GPU)
    void* p = nullptr;
    void* dev = nullptr;  // the same buffer as the GPU addresses it (nullptr: not mapped, copy instead)
    size_t cap = 0;
    hipEvent_t ev = nullptr;
    bool pending = false;
    int ensure(size_t bytes) {
        if (pending) {
            hipError_t e = hipEventSynchronize(ev);
            pending = false;
            if (e != hipSuccess) return fail(DCTFP_ERR_HIP, "hipEventSynchronize: %s", hipGetErrorString(e));
        }
        if (!ev) {
            hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e != hipSuccess) return fail(DCTFP_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(e));
        }
        if (bytes <= cap) return DCTFP_OK;
        if (p) {
            (void)hipDeviceSynchronize();   // (a kernel reading the tables straight from this buffer, on whatever stream)
            (void)hipHostFree(p);
        }
        p = nullptr;
        dev = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 2 + 4096;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(DCTFP_ERR_NOMEM, "hipHostMalloc(%zu): %s", want, hipGetErrorString(e));
        }
        cap = want;
        dev = nullptr;
        if (hipHostGetDevicePointer(&dev, p, 0) != hipSuccess) {
            dev = nullptr;
            (void)hipGetLastError();
        }
        return DCTFP_OK;
    }
    void release() {
        if (pending && ev) (void)hipEventSynchronize(ev);
        if (p) (void)hipHostFree(p);
        if (ev) (void)hipEventDestroy(ev);
        p = nullptr;
        ev = nullptr;
        cap = 0;
    }
};

struct StEntry {  // stage-B basis  St[d][c] (ldy x cp), zero padded
    double* dev = nullptr;
    double* frag = nullptr;  // even/odd halves of the basis in MFMA-fragment order (walk_ab_kernel), `frag_groups` 16-pair groups
    int frag_groups = 0;
    double* fragp = nullptr; // the plain basis in MFMA-fragment order (walk_gen_kernel), made when that kernel first wants it
    int ldy = 0, cp = 0;
    uint64_t last_use = 0;
};

struct BasisSlab {  // a piece of the context's cosine-table arena
    double* dev = nullptr;
    size_t cap = 0, used = 0;  // doubles
};

struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    int which = 0;
};

}  // namespace

constexpr int kMaxSlots = 8;
constexpr size_t kCounterBytes = 24 * sizeof(unsigned long long);  // [0] degenerate channels; [1..11] phase times of an instrumented build; [15] device address of the host flag; [16], [17] trace of an instrumented build

struct dctfp_ctx {
    int device = 0;
    int64_t opt_overlap = 4;
    hipStream_t side = nullptr;
    hipStream_t copy = nullptr;                 // table upload + cosine tables, ahead of the caller's stream
    hipEvent_t ev_tab_free[2] = {}, ev_tab_ready = nullptr;
    hipEvent_t ev_ws_free = nullptr;            // after the last reader of the Y' scratch (any stream, any call)
    hipEvent_t ev_basis = nullptr;              // after the last basis_kernel (cosine tables are shared by all later calls)
    hipStream_t basis_stream = nullptr;
    bool basis_valid = false;
    bool tab_busy[2] = {false, false};
    bool ws_busy = false;
    std::mutex mu;                              // one host thread at a time inside a context
    int ensure_copy() {
        if (copy) return DCTFP_OK;
        if (hipStreamCreateWithFlags(&copy, hipStreamNonBlocking) != hipSuccess) { copy = nullptr; set_err("hipStreamCreate(copy) failed"); return DCTFP_ERR_HIP; }
        if (hipEventCreateWithFlags(&ev_tab_ready, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ev_ws_free, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ev_basis, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ev_tab_free[0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ev_tab_free[1], hipEventDisableTiming) != hipSuccess) { set_err("hipEventCreate failed"); return DCTFP_ERR_HIP; }
        return DCTFP_OK;
    }
    hipEvent_t ev_a[kMaxSlots] = {}, ev_b[kMaxSlots] = {};
    int ensure_side() {
        if (side) return DCTFP_OK;
        hipError_t e = hipStreamCreateWithFlags(&side, hipStreamNonBlocking);
        if (e != hipSuccess) { side = nullptr; snprintf(g_err, sizeof g_err, "hipStreamCreate: %s", hipGetErrorString(e)); return DCTFP_ERR_HIP; }
        for (int i = 0; i < kMaxSlots; ++i) {
            if (hipEventCreateWithFlags(&ev_a[i], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&ev_b[i], hipEventDisableTiming) != hipSuccess) {
                set_err("hipEventCreate failed");
                return DCTFP_ERR_HIP;
            }
        }
        return DCTFP_OK;
    }
    int64_t opt_stage_b = 1, opt_a_waves = 0, opt_a_unroll = 4, opt_profile = 0, opt_ws_mb = 4096;
    DevBuf tables[2];
    Staging staging[2];
    int flip = 0;
    DevBuf ws;       // yprime
    int64_t opt_fuse = 1, opt_pack_y = 1;
    DevBuf scratch;  // generic idct_quant fs
    DevBuf split_ws; // partial sums of the row-split stage A (small calls)
    uint32_t* small_tickets = nullptr;  // small_call_kernel: arrival counters, zeroed once (their last taker resets them)
    // "small_one": 1 = a small call of the production shape in ONE launch (small_call_kernel, round 5).  Off by default: measured
    // 86 us per one-protein call against 57 through the three kernels (profiles/r05/pcie_rate_one_launch.txt, _three_launches.txt) --
    // what the launches cost is what the hand-over inside a grid costs too on this chip: a workgroup on another XCD sees the
    // partial sums only through agent-scope release / acquire (L2 write-back and invalidate per workgroup) and memory-side atomics.
    int64_t opt_small_one = 0;
    int64_t last_small_one = 0;         // read only ("last_small_one"): the last dctfp_quantize went through small_call_kernel
    // stage-A cosine tables, one per (domain length, n - 1): filled once, kept for the life of the context
    std::vector<BasisSlab> basis_slabs;
    std::unordered_map<uint64_t, double*> basis_tabs;
    size_t basis_doubles = 0;
    unsigned long long* degenerate = nullptr;  // device counter: exactly constant channels seen (see dctfp.h)
    uint32_t* flag_host = nullptr;             // pinned + mapped word the kernels set when they see one (option degenerate_seen)
    int n_cu = 256;  // compute units of the device (workgroup slots of the walk kernel = n_cu x workgroups per CU)
    void *trace_dev = nullptr, *trace_host = nullptr;  // instrumented build only (walk_trace)
    int64_t trace_waves = 0;
    DevBuf cut_ws;   // dctfp_reccut: adjacency lists and node stacks of a batch
    hipStream_t cut_stream[kCutClasses - 1] = {};   // ... its larger size classes run beside the small one
    hipEvent_t cut_ev[kCutClasses] = {};
    hipEvent_t ev_cut_ws_free = nullptr;            // after the last dctfp_reccut that used cut_ws (calls on different streams share it)
    bool cut_ws_busy = false;
    int ensure_cut_streams() {
        if (cut_stream[0]) return DCTFP_OK;
        // (the highest priority the device offers: what runs on them -- the long proteins' workgroups, the striped selection -- is
        //  what a flush waits for, and its workgroups should take the CUs the short proteins' ones leave)
        int prio_low = 0, prio_high = 0;
        if (hipDeviceGetStreamPriorityRange(&prio_low, &prio_high) != hipSuccess) {
            (void)hipGetLastError();
            prio_high = 0;
        }
        for (int i = 0; i < kCutClasses - 1; ++i)
            if (hipStreamCreateWithPriority(&cut_stream[i], hipStreamNonBlocking, prio_high) != hipSuccess) { cut_stream[i] = nullptr; set_err("hipStreamCreate(cut) failed"); return DCTFP_ERR_HIP; }
        for (int i = 0; i < kCutClasses; ++i)
            if (hipEventCreateWithFlags(&cut_ev[i], hipEventDisableTiming) != hipSuccess) { set_err("hipEventCreate failed"); return DCTFP_ERR_HIP; }
        return DCTFP_OK;
    }
    int64_t opt_path = 0, opt_ab_group = 0, opt_ab_unroll = 0, opt_ab_run_jobs = 0, opt_small_b_jobs = 512, opt_ab_longest_first = 0, opt_ab_mfma_a = 0, opt_ab_taper = 4, opt_ab_align = 2, opt_l1_kernel = 0, opt_row_select = 0, opt_stitch_once = 0, opt_topk_kernel = 0;
    int64_t last_path = 0;  // which kernels the last dctfp_quantize launched: 1 = stage A + stage B, 2 = walk kernel
    int64_t last_gen_fused = 0;  // ... and whether that was the general walk kernel streaming fused walks (parts + whole protein)
    int64_t last_walk_groups = 0;  // ... or walk_ab_kernel: its build's 16-column groups (5: m <= 80, 6: 80 < m <= 96), 0 = another kernel
    int64_t opt_gen_fuse = 1;    // "gen_fuse": fused walks through the general walk kernel (n <= 5); 0 = the two kernels, as through round 4
    int64_t walk_launches = 0;  // walk-kernel launches so far (a call split at a giant domain ends on the two-kernel path)
    int64_t test_fail_once = 0;                    // test hook: the next dctfp_quantize fails after its table lookups
    int64_t basis_cap_doubles = (int64_t)1 << 27;  // 1 GiB of cosine tables, then the arena starts over (test hook: basis_cap_kb)
    int64_t basis_restarts = 0;                    // how often it did
    std::map<std::pair<int, int>, StEntry> st_cache;
    uint64_t tick = 0;
    std::vector<EventPair> events;
    size_t events_used = 0;
    double prof_ms[2] = {0, 0};
    int64_t prof_n[2] = {0, 0};
};

namespace {

// Records that `stream` has (enqueued) the last reader of table buffer `buf`: a later dctfp_quantize uploads into it
// from the copy stream only after this point.
int mark_table_used(dctfp_ctx* ctx, int buf, hipStream_t stream) {
    int rc = ctx->ensure_copy();
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ctx->ev_tab_free[buf], stream));
    ctx->tab_busy[buf] = true;
    return DCTFP_OK;
}

inline size_t dtype_size(int dtype) { return dtype == DCTFP_F64 ? 8 : (dtype == DCTFP_F32 ? 4 : 2); }

// length -> cosine-table offset.  A flat array while the longest domain is moderate, a hash map beyond.
struct LenTable {
    static constexpr uint32_t kNone = 0xffffffffu;
    std::vector<uint32_t> flat;
    std::unordered_map<uint32_t, uint32_t> map;
    bool use_flat;
    explicit LenTable(uint32_t max_len) : use_flat(max_len <= (1u << 22)) {
        if (use_flat) flat.assign((size_t)max_len + 1, kNone);
    }
    bool has(uint32_t len) const { return use_flat ? flat[len] != kNone : map.find(len) != map.end(); }
    void set(uint32_t len, uint32_t off) {
        if (use_flat) flat[len] = off;
        else map[len] = off;
    }
    uint32_t operator[](uint32_t len) const { return use_flat ? flat[len] : map.at(len); }
};

int get_st(dctfp_ctx* ctx, int n_cols, int m, StEntry** out) {
    auto key = std::make_pair(n_cols, m);
    auto it = ctx->st_cache.find(key);
    if (it != ctx->st_cache.end()) {
        it->second.last_use = ++ctx->tick;
        *out = &it->second;
        return DCTFP_OK;
    }
    if (ctx->st_cache.size() >= 32) {  // drop the least recently used basis
        auto victim = ctx->st_cache.begin();
        for (auto i = ctx->st_cache.begin(); i != ctx->st_cache.end(); ++i)
            if (i->second.last_use < victim->second.last_use) victim = i;
        HIP_TRY(hipDeviceSynchronize());
        (void)hipFree(victim->second.dev);
        if (victim->second.frag) (void)hipFree(victim->second.frag);
        if (victim->second.fragp) (void)hipFree(victim->second.fragp);
        ctx->st_cache.erase(victim);
    }
    const int ldy = (int)align_up((size_t)n_cols, 32);
    const int cp = (int)align_up((size_t)m, 16);
    // S[c][d] = sum_{k=1}^{m-1} cos(pi k (2c+1) / (2m)) cos(pi k (2d+1) / (2D)): forward DCT-II
    // over the D channels, keep m, inverse DCT of length m -- k = 0 and the common
    // normalisation dropped (the per-row min-max scale removes both).
    std::vector<long double> ca((size_t)m * m), cb((size_t)m * n_cols);
    for (int k = 1; k < m; ++k) {
        for (int c = 0; c < m; ++c) ca[(size_t)k * m + c] = cospi_ratio_host((int64_t)k * (2 * c + 1), 2 * (int64_t)m);
        for (int d = 0; d < n_cols; ++d)
            cb[(size_t)k * n_cols + d] = cospi_ratio_host((int64_t)k * (2 * (int64_t)d + 1), 2 * (int64_t)n_cols);
    }
    std::vector<double> host((size_t)ldy * cp, 0.0);
    std::vector<long double> col(m);
    for (int d = 0; d < n_cols; ++d) {
        for (int c = 0; c < m; ++c) col[c] = 0.0L;
        for (int k = 1; k < m; ++k) {
            const long double b = cb[(size_t)k * n_cols + d];
            const long double* a = &ca[(size_t)k * m];
            for (int c = 0; c < m; ++c) col[c] += a[c] * b;
        }
        for (int c = 0; c < m; ++c) host[(size_t)d * cp + c] = (double)col[c];
    }
    StEntry e;
    e.ldy = ldy;
    e.cp = cp;
    // (an exception between here and the insertion into the cache -- std::bad_alloc from one of the host vectors below --
    //  must not leave the device allocations behind: found by the allocation-failure hook of tests/asan/driver.cpp)
    struct DevGuard {
        StEntry& e;
        bool armed = true;
        ~DevGuard() {
            if (!armed) return;
            if (e.dev) (void)hipFree(e.dev);
            if (e.frag) (void)hipFree(e.frag);
        }
    } dev_guard{e};
    HIP_TRY(hipMalloc((void**)&e.dev, host.size() * sizeof(double)));
    hipError_t err = hipMemcpy(e.dev, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice);
    if (err != hipSuccess) return fail(DCTFP_ERR_HIP, "hipMemcpy(St): %s", hipGetErrorString(err));
    if ((cp == 80 || cp == 96) && n_cols % 4 == 0) {
        // walk_ab_kernel contracts the even and the odd half of the basis apart (kernels.hip.h, "flush"):
        //   E[d][c] = sum over even k, O[d][c] = sum over odd k of cos_m(k, c) cos_D(k, d),  d < D/2, c < ceil(m/2)
        // Tab[d] = [E[d][0..39] | O[d][0..39]] (zero past ceil(m/2); odd m: O is exactly 0 in the middle column), in
        // fragment order: frag[((q * 4 + r) * 5 + c) * 64 + lane] = Tab[16 q + 4 (lane >> 4) + r][16 c + (lane & 15)]
        // (cp = 96 -- 80 < m <= 96, PROST's [3, 85] --: six column groups, halves of 48 slots)
        const int nt = cp / 16, hs = cp / 2, half = n_cols / 2, hm = (m + 1) / 2;
        e.frag_groups = (half + 15) / 16;
        std::vector<double> tab((size_t)half * cp, 0.0);
        std::vector<long double> ce(hm), co(hm);
        for (int d = 0; d < half; ++d) {
            for (int c = 0; c < hm; ++c) ce[c] = co[c] = 0.0L;
            for (int k = 1; k < m; ++k) {
                const long double b = cb[(size_t)k * n_cols + d];
                const long double* a = &ca[(size_t)k * m];
                long double* dst = (k & 1) ? co.data() : ce.data();
                for (int c = 0; c < hm; ++c) dst[c] += a[c] * b;
            }
            for (int c = 0; c < hm; ++c) {
                tab[(size_t)d * cp + c] = (double)ce[c];
                tab[(size_t)d * cp + hs + c] = (m - 1 - c == c) ? 0.0 : (double)co[c];
            }
        }
        // (two groups of zeros behind the last one: the flush requests its fragments a few k-steps ahead without a clamp)
        std::vector<double> fr((size_t)(e.frag_groups + 2) * nt * 256, 0.0);
        for (int q = 0; q < e.frag_groups; ++q)
            for (int r = 0; r < 4; ++r)
                for (int c = 0; c < nt; ++c)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int d = 16 * q + 4 * (lane >> 4) + r, col = 16 * c + (lane & 15);
                        if (d < half) fr[(((size_t)q * 4 + r) * nt + c) * 64 + lane] = tab[(size_t)d * cp + col];
                    }
        err = hipMalloc((void**)&e.frag, fr.size() * sizeof(double));
        if (err == hipSuccess) err = hipMemcpy(e.frag, fr.data(), fr.size() * sizeof(double), hipMemcpyHostToDevice);
        if (err != hipSuccess) return fail(DCTFP_ERR_HIP, "St fragments: %s", hipGetErrorString(err));
    }
    e.last_use = ++ctx->tick;
    auto ins = ctx->st_cache.emplace(key, e);
    dev_guard.armed = false;
    *out = &ins.first->second;
    return DCTFP_OK;
}

// The plain basis of (D, m) in the fragment order of walk_gen_kernel, from the row-major table get_st made:
//   fragp[(q NT + c) 64 + lane] = St[4 q + (lane >> 4)][16 c + (lane & 15)],  zero beyond D;  q < ceil(D / 4) + 64
// (the kernel's waves stop at the last k-step that holds a channel; the slack keeps a wave's base address inside the table).
int get_st_plain(dctfp_ctx* ctx, StEntry* st, int n_cols) {
    if (st->fragp) return DCTFP_OK;
    const int nt = st->cp / 16, steps = (n_cols + 3) / 4 + 64;
    std::vector<double> host((size_t)st->ldy * st->cp);
    HIP_TRY(hipMemcpy(host.data(), st->dev, host.size() * sizeof(double), hipMemcpyDeviceToHost));
    std::vector<double> fr((size_t)steps * nt * 64, 0.0);
    for (int q = 0; q < steps; ++q)
        for (int c = 0; c < nt; ++c)
            for (int lane = 0; lane < 64; ++lane) {
                const int d = 4 * q + (lane >> 4), col = 16 * c + (lane & 15);
                if (d < n_cols) fr[((size_t)q * nt + c) * 64 + lane] = host[(size_t)d * st->cp + col];
            }
    double* dev = nullptr;
    HIP_TRY(hipMalloc((void**)&dev, fr.size() * sizeof(double)));
    const hipError_t err = hipMemcpy(dev, fr.data(), fr.size() * sizeof(double), hipMemcpyHostToDevice);
    if (err != hipSuccess) {
        (void)hipFree(dev);
        return fail(DCTFP_ERR_HIP, "St plain fragments: %s", hipGetErrorString(err));
    }
    st->fragp = dev;
    (void)ctx;
    return DCTFP_OK;
}

constexpr size_t kBasisSlabDoubles = (size_t)1 << 20;   // 8 MB pieces

// Drops every cosine table (only between calls: nothing may be in flight).
int basis_purge(dctfp_ctx* ctx) {
    HIP_TRY(hipDeviceSynchronize());
    for (auto& sl : ctx->basis_slabs) (void)hipFree(sl.dev);
    ctx->basis_slabs.clear();
    ctx->basis_tabs.clear();
    ctx->basis_doubles = 0;
    ctx->basis_valid = false;
    ctx->basis_restarts += 1;
    return DCTFP_OK;
}

// Device address of the cosine table of (len, nk); a table seen for the first time gets its room in the arena and is
// appended to `fresh`.  It is NOT in the cache yet: the caller launches basis_kernel for the fresh tables and only then
// publishes them (basis_publish); every error exit in between gives the room back (BasisRollback) -- a length cached
// before its table is filled would hand uninitialised cosines to every later call.
int basis_lookup(dctfp_ctx* ctx, uint32_t len, int nk, double** out, std::vector<BasisJob>& fresh) {
    const uint64_t key = ((uint64_t)nk << 32) | len;
    auto it = ctx->basis_tabs.find(key);
    if (it != ctx->basis_tabs.end()) {
        *out = it->second;
        return DCTFP_OK;
    }
    const size_t need = align_up((size_t)len * nk + ((size_t)len + 1) * nk, 2);  // cosines + prefix sums; 16-byte granules (s_load_dwordx4)
    if (ctx->basis_slabs.empty() || ctx->basis_slabs.back().used + need > ctx->basis_slabs.back().cap) {
        BasisSlab sl;
        sl.cap = std::max(need, kBasisSlabDoubles);
        hipError_t e = hipMalloc((void**)&sl.dev, sl.cap * sizeof(double));
        if (e != hipSuccess) return fail(DCTFP_ERR_NOMEM, "hipMalloc(cosine tables, %zu bytes): %s", sl.cap * sizeof(double), hipGetErrorString(e));
        ctx->basis_slabs.push_back(sl);
    }
    BasisSlab& sl = ctx->basis_slabs.back();
    double* tab = sl.dev + sl.used;
    sl.used += need;
    ctx->basis_doubles += need;
    BasisJob bj;
    bj.tab = tab;
    bj.len = len;
    bj.reserved = 0;
    fresh.push_back(bj);
    *out = tab;
    return DCTFP_OK;
}

// All of `fresh` or none of it: an insertion that fails half way (std::bad_alloc) takes the entries already made back out,
// because the caller's BasisRollback then gives their room in the arena back -- a cached length whose table has been
// freed would hand a dangling pointer to every later call (found by the allocation-failure hook of tests/asan/driver.cpp).
void basis_publish(dctfp_ctx* ctx, const std::vector<BasisJob>& fresh, int nk) {
    size_t done = 0;
    try {
        for (; done < fresh.size(); ++done) ctx->basis_tabs.emplace(((uint64_t)nk << 32) | fresh[done].len, fresh[done].tab);
    } catch (...) {
        for (size_t i = 0; i < done; ++i) ctx->basis_tabs.erase(((uint64_t)nk << 32) | fresh[i].len);
        throw;
    }
}

// State of the arena before a call reserves room for fresh tables; restore() undoes the reservations.
struct BasisRollback {
    dctfp_ctx* ctx;
    size_t n_slabs, used_last, doubles;
    bool armed = true;
    explicit BasisRollback(dctfp_ctx* c)
        : ctx(c), n_slabs(c->basis_slabs.size()), used_last(c->basis_slabs.empty() ? 0 : c->basis_slabs.back().used),
          doubles(c->basis_doubles) {}
    void restore() {
        if (ctx->basis_slabs.size() > n_slabs) {
            (void)hipDeviceSynchronize();
            while (ctx->basis_slabs.size() > n_slabs) {
                (void)hipFree(ctx->basis_slabs.back().dev);
                ctx->basis_slabs.pop_back();
            }
        }
        if (n_slabs > 0) ctx->basis_slabs[n_slabs - 1].used = used_last;
        ctx->basis_doubles = doubles;
    }
    ~BasisRollback() {
        if (armed) restore();
    }
};

}  // namespace

namespace dctfp_host {
int launch_fail(LaunchError* err, int code, const char* fmt, ...) {
    if (err) {
        err->code = code;
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(err->msg, sizeof err->msg, fmt, ap);
        va_end(ap);
    }
    return code;
}
}  // namespace dctfp_host

namespace {

// stage A of the two-kernel path, by storage type (each in its own translation unit)
void launch_a(const AParams& p, int dtype, int vec, int n, int waves, int unroll) {
    if (dtype == DCTFP_F32) launch_a_f32(p, vec, n, waves, unroll);
    else if (dtype == DCTFP_F64) launch_a_f64(p, vec, n, waves, unroll);
    else if (dtype == DCTFP_F16) launch_a_f16(p, vec, n, waves, unroll);
    else launch_a_bf16(p, vec, n, waves, unroll);
}

// a launcher's failure -> this thread's error message
int launcher_rc(int rc, const LaunchError& err) {
    return rc ? fail(err.code ? err.code : rc, "%s", err.msg) : DCTFP_OK;
}

int prof_begin(dctfp_ctx* ctx, int which, hipStream_t s, EventPair** ep) {
    *ep = nullptr;
    if (!ctx->opt_profile) return DCTFP_OK;
    if (ctx->events_used == ctx->events.size()) {
        EventPair e;
        HIP_TRY(hipEventCreate(&e.a));
        HIP_TRY(hipEventCreate(&e.b));
        ctx->events.push_back(e);
    }
    EventPair& e = ctx->events[ctx->events_used++];
    e.which = which;
    HIP_TRY(hipEventRecord(e.a, s));
    *ep = &e;
    return DCTFP_OK;
}

int prof_end(EventPair* ep, hipStream_t s) {
    if (!ep) return DCTFP_OK;
    HIP_TRY(hipEventRecord(ep->b, s));
    return DCTFP_OK;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Nothing may throw across the C ABI: an exception that leaves an extern "C" function ends the process in std::terminate
// (SIGABRT on the calling thread).  Every entry point is a function-try-block that ends here; the vectors of the table
// build are the only things that can throw (std::bad_alloc, std::length_error).
int guard_exception(const char* where) noexcept {
    try {
        throw;
    } catch (const std::bad_alloc&) {
        return fail(DCTFP_ERR_NOMEM, "%s: out of host memory", where);
    } catch (const std::exception& e) {
        return fail(DCTFP_ERR_INVALID, "%s: %s", where, e.what());
    } catch (...) {
        return fail(DCTFP_ERR_INVALID, "%s: unknown C++ exception", where);
    }
}
#define DCTFP_GUARD(name) catch (...) { return guard_exception(name); }

}  // namespace

extern "C" {

int dctfp_version(void) { return DCTFP_VERSION; }

const char* dctfp_last_error(void) { return g_err; }

int dctfp_create(int device, dctfp_ctx** out) try {
    if (!out) return fail(DCTFP_ERR_INVALID, "dctfp_create: out is NULL");
    *out = nullptr;
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return fail(DCTFP_ERR_INVALID, "dctfp_create: device %d of %d", device, count);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(DCTFP_ERR_HIP, "dctfp_create: device %d is %s; this library is built for gfx950 only", device,
                    prop.gcnArchName);
    dctfp_ctx* ctx = new (std::nothrow) dctfp_ctx();
    if (!ctx) return fail(DCTFP_ERR_NOMEM, "dctfp_create: out of host memory");
    ctx->device = device;
    ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    hipError_t e = hipMalloc((void**)&ctx->degenerate, kCounterBytes);
    if (e == hipSuccess) e = hipMemset(ctx->degenerate, 0, kCounterBytes);
    if (e == hipSuccess) e = hipHostMalloc((void**)&ctx->flag_host, 64, hipHostMallocMapped);
    if (e == hipSuccess) {
        *ctx->flag_host = 0;
        void* flag_dev = nullptr;
        e = hipHostGetDevicePointer(&flag_dev, ctx->flag_host, 0);
        const unsigned long long addr = (unsigned long long)(uintptr_t)flag_dev;
        if (e == hipSuccess) e = hipMemcpy(ctx->degenerate + kFlagSlot, &addr, sizeof addr, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        if (ctx->degenerate) (void)hipFree(ctx->degenerate);
        if (ctx->flag_host) (void)hipHostFree(ctx->flag_host);
        delete ctx;
        return fail(DCTFP_ERR_HIP, "dctfp_create: %s", hipGetErrorString(e));
    }
    *out = ctx;
    return DCTFP_OK;
} DCTFP_GUARD("dctfp_create")

int dctfp_destroy(dctfp_ctx* ctx) try {
    if (!ctx) return DCTFP_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (auto& t : ctx->tables) t.release();
    for (auto& s : ctx->staging) s.release();
    ctx->ws.release();
    ctx->scratch.release();
    ctx->split_ws.release();
    if (ctx->small_tickets) (void)hipFree(ctx->small_tickets);
    ctx->small_tickets = nullptr;
    ctx->cut_ws.release();
    for (auto& kv : ctx->st_cache) {
        (void)hipFree(kv.second.dev);
        if (kv.second.frag) (void)hipFree(kv.second.frag);
        if (kv.second.fragp) (void)hipFree(kv.second.fragp);
    }
    for (auto& sl : ctx->basis_slabs) (void)hipFree(sl.dev);
    if (ctx->degenerate) (void)hipFree(ctx->degenerate);
    if (ctx->flag_host) (void)hipHostFree(ctx->flag_host);
    for (auto& e : ctx->events) {
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    if (ctx->copy) {
        (void)hipStreamDestroy(ctx->copy);
        (void)hipEventDestroy(ctx->ev_tab_ready);
        (void)hipEventDestroy(ctx->ev_ws_free);
        (void)hipEventDestroy(ctx->ev_basis);
        (void)hipEventDestroy(ctx->ev_tab_free[0]);
        (void)hipEventDestroy(ctx->ev_tab_free[1]);
    }
    if (ctx->side) {
        (void)hipStreamDestroy(ctx->side);
        for (int i = 0; i < kMaxSlots; ++i) {
            if (ctx->ev_a[i]) (void)hipEventDestroy(ctx->ev_a[i]);
            if (ctx->ev_b[i]) (void)hipEventDestroy(ctx->ev_b[i]);
        }
    }
    for (int i = 0; i < kCutClasses; ++i)
        if (ctx->cut_ev[i]) (void)hipEventDestroy(ctx->cut_ev[i]);
    if (ctx->ev_cut_ws_free) (void)hipEventDestroy(ctx->ev_cut_ws_free);
    for (int i = 0; i < kCutClasses - 1; ++i)
        if (ctx->cut_stream[i]) (void)hipStreamDestroy(ctx->cut_stream[i]);
    delete ctx;
    return DCTFP_OK;
} DCTFP_GUARD("dctfp_destroy")

int dctfp_set_option(dctfp_ctx* ctx, const char* name, int64_t value) try {
    if (!ctx || !name) return fail(DCTFP_ERR_INVALID, "dctfp_set_option: NULL argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    std::string n(name);
    // ---- what a user of the library sets (include/dctfp.h)
    if (n == "path") {
        if (value < 0 || value > 2) return fail(DCTFP_ERR_INVALID, "path must be 0 (auto), 1 (stage A -> Y' -> stage B) or 2 (walk kernel wherever its shapes allow)");
        ctx->opt_path = value;
    } else if (n == "fuse") {
        ctx->opt_fuse = value ? 1 : 0;
    } else if (n == "gen_fuse") {
        ctx->opt_gen_fuse = value ? 1 : 0;
    } else if (n == "small_one") {
        ctx->opt_small_one = value ? 1 : 0;
    } else if (n == "workspace_mb") {
        if (value < 16) return fail(DCTFP_ERR_INVALID, "workspace_mb must be >= 16");
        ctx->opt_ws_mb = value;
    } else if (n == "profile") {
        ctx->opt_profile = value ? 1 : 0;
    } else if (n == "degenerate_channels") {
        if (value != 0) return fail(DCTFP_ERR_INVALID, "degenerate_channels can only be reset to 0");
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemset(ctx->degenerate, 0, kFlagSlot * sizeof(unsigned long long)));  // (slot 15 keeps the flag's address)
        *ctx->flag_host = 0;
    }
#ifdef DCTFP_WALK_TIMELINE
    else if (n == "walk_trace") {  // instrumented build: per-wave {begin, end, HW_ID, workgroup} of the next walk launch (value = waves)
        if (value < 0 || value > (1 << 24)) return fail(DCTFP_ERR_INVALID, "walk_trace must be 0 .. 2^24 waves");
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(hipDeviceSynchronize());
        if (ctx->trace_dev) (void)hipFree(ctx->trace_dev);
        ctx->trace_dev = nullptr;
        free(ctx->trace_host);
        ctx->trace_host = nullptr;
        ctx->trace_waves = value;
        unsigned long long slots[2] = {(unsigned long long)value, 0};
        if (value > 0) {
            HIP_TRY(hipMalloc((void**)&ctx->trace_dev, (size_t)value * 32));
            HIP_TRY(hipMemset(ctx->trace_dev, 0, (size_t)value * 32));
            slots[1] = (unsigned long long)(uintptr_t)ctx->trace_dev;
        }
        HIP_TRY(hipMemcpy(ctx->degenerate + 16, slots, sizeof slots, hipMemcpyHostToDevice));
    }
#endif
#ifdef DCTFP_EXPERIMENTS
    // ---- engineering knobs and test hooks: libdctfp_experiments.so only (A/B tools, kernel-variant parity tests, cache tests)
    else if (n == "stage_b") {
        if (value != 0 && value != 1) return fail(DCTFP_ERR_INVALID, "stage_b must be 0 or 1");
        ctx->opt_stage_b = value;
    } else if (n == "a_waves") {
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8 && value != 16)
            return fail(DCTFP_ERR_INVALID, "a_waves must be 0 (auto), 1, 2, 4, 8 or 16");
        ctx->opt_a_waves = value;
    } else if (n == "a_unroll") {
        if (value != 4 && value != 8) return fail(DCTFP_ERR_INVALID, "a_unroll must be 4 or 8");
        ctx->opt_a_unroll = value;
    } else if (n == "pack_y") {
        ctx->opt_pack_y = value ? 1 : 0;
    } else if (n == "ab_group") {
        if (value != 0 && value != 3 && value != 4) return fail(DCTFP_ERR_INVALID, "ab_group must be 0 (auto), 3 or 4 jobs per flush");
        ctx->opt_ab_group = value;
    } else if (n == "ab_unroll") {
        if (value != 0 && value != 4 && value != 6 && value != 8 && value != 12 && value != 16)
            return fail(DCTFP_ERR_INVALID, "ab_unroll must be 0 (auto), 4, 6, 8, 12 or 16");
        ctx->opt_ab_unroll = value;
    } else if (n == "small_b_jobs") {
        if (value < 0 || value > 1 << 20) return fail(DCTFP_ERR_INVALID, "small_b_jobs must be 0 .. 2^20");
        ctx->opt_small_b_jobs = value;
    } else if (n == "ab_longest_first") {
        if (value < 0 || value > 2) return fail(DCTFP_ERR_INVALID, "ab_longest_first must be 0 (auto), 1 (on) or 2 (off)");
        ctx->opt_ab_longest_first = value;
    } else if (n == "ab_taper") {
        if (value < 0 || value > 64) return fail(DCTFP_ERR_INVALID, "ab_taper must be 0 (off) .. 64 (quarters of a round of workgroups whose jobs go out in short runs)");
        ctx->opt_ab_taper = value;
    } else if (n == "ab_align") {
        if (value < 0 || value > 8) return fail(DCTFP_ERR_INVALID, "ab_align must be 0 (off) .. 8 (walks to look ahead for a run that ends on a full flush)");
        ctx->opt_ab_align = value;
    } else if (n == "l1_kernel") {
        if (value != 0 && value != 1) return fail(DCTFP_ERR_INVALID, "l1_kernel must be 0 (by alignment) or 1 (the 4-byte kernel whatever the alignment)");
        ctx->opt_l1_kernel = value;
    } else if (n == "stitch_once") {
        if (value != 0 && value != 2) return fail(DCTFP_ERR_INVALID, "stitch_once must be 0 (one launch where the windows allow it) or 2 (one launch per window index)");
        ctx->opt_stitch_once = value;
    } else if (n == "topk_kernel") {
        if (value < 0 || value > 2)
            return fail(DCTFP_ERR_INVALID, "topk_kernel must be 0 (one read; two reads, then the radix select behind it), 1 (radix select only) or 2 (two reads first)");
        ctx->opt_topk_kernel = value;
    } else if (n == "row_select") {
        if (value != 0 && value != 1) return fail(DCTFP_ERR_INVALID, "row_select must be 0 (by shape) or 1 (the radix select whatever the shape)");
        ctx->opt_row_select = value;
    } else if (n == "ab_mfma_a") {
        ctx->opt_ab_mfma_a = value ? 1 : 0;
    } else if (n == "ab_run_jobs") {
        if (value < 0 || value > 4096) return fail(DCTFP_ERR_INVALID, "ab_run_jobs must be 0 (auto) .. 4096");
        ctx->opt_ab_run_jobs = value;
    } else if (n == "overlap") {
        if (value < 1 || value > kMaxSlots) return fail(DCTFP_ERR_INVALID, "overlap must be 1..%d", kMaxSlots);
        ctx->opt_overlap = value;
    } else if (n == "test_fail_once") {
        ctx->test_fail_once = value ? 1 : 0;
    } else if (n == "basis_cap_kb") {
        if (value < 1) return fail(DCTFP_ERR_INVALID, "basis_cap_kb must be >= 1");
        ctx->basis_cap_doubles = value * 128;
    }
#endif
    else {
        return fail(DCTFP_ERR_INVALID, "unknown option '%s'", name);
    }
    return DCTFP_OK;
} DCTFP_GUARD("dctfp_set_option")

int dctfp_get_option(dctfp_ctx* ctx, const char* name, int64_t* value) try {
    if (!ctx || !name || !value) return fail(DCTFP_ERR_INVALID, "dctfp_get_option: NULL argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    std::string n(name);
    if (n == "path") *value = ctx->opt_path;
    else if (n == "last_path") *value = ctx->last_path;
    else if (n == "last_gen_fused") *value = ctx->last_gen_fused;
    else if (n == "last_walk_groups") *value = ctx->last_walk_groups;
    else if (n == "gen_fuse") *value = ctx->opt_gen_fuse;
    else if (n == "small_one") *value = ctx->opt_small_one;
    else if (n == "last_small_one") *value = ctx->last_small_one;
    else if (n == "walk_launches") *value = ctx->walk_launches;
    else if (n == "fuse") *value = ctx->opt_fuse;
    else if (n == "workspace_mb") *value = ctx->opt_ws_mb;
    else if (n == "profile") *value = ctx->opt_profile;
    else if (n == "degenerate_channels") {  // synchronises the device
        unsigned long long v = 0;
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(&v, ctx->degenerate, sizeof v, hipMemcpyDeviceToHost));
        *value = (int64_t)v;
    } else if (n == "degenerate_seen") {  // no device synchronisation: meaningful once the caller has waited for its call
        // one exchange: a kernel of another stream that raises the flag between a read and a clear would be lost
        *value = __atomic_exchange_n(ctx->flag_host, 0u, __ATOMIC_ACQ_REL) ? 1 : 0;
    }
#ifdef DCTFP_EXPERIMENTS
    else if (n == "stage_b") *value = ctx->opt_stage_b;
    else if (n == "a_waves") *value = ctx->opt_a_waves;
    else if (n == "a_unroll") *value = ctx->opt_a_unroll;
    else if (n == "overlap") *value = ctx->opt_overlap;
    else if (n == "ab_group") *value = ctx->opt_ab_group;
    else if (n == "ab_unroll") *value = ctx->opt_ab_unroll;
    else if (n == "ab_run_jobs") *value = ctx->opt_ab_run_jobs;
    else if (n == "ab_longest_first") *value = ctx->opt_ab_longest_first;
    else if (n == "small_b_jobs") *value = ctx->opt_small_b_jobs;
    else if (n == "ab_mfma_a") *value = ctx->opt_ab_mfma_a;
    else if (n == "ab_taper") *value = ctx->opt_ab_taper;
    else if (n == "ab_align") *value = ctx->opt_ab_align;
    else if (n == "l1_kernel") *value = ctx->opt_l1_kernel;
    else if (n == "row_select") *value = ctx->opt_row_select;
    else if (n == "topk_kernel") *value = ctx->opt_topk_kernel;
    else if (n == "stitch_once") *value = ctx->opt_stitch_once;
    else if (n == "pack_y") *value = ctx->opt_pack_y;
    else if (n == "basis_cap_kb") *value = ctx->basis_cap_doubles / 128;
    else if (n == "basis_restarts") *value = ctx->basis_restarts;
    else if (n == "basis_tables") *value = (int64_t)ctx->basis_tabs.size();
#endif
#ifdef DCTFP_WALK_TIMELINE
    else if (n.rfind("walk_timeline_", 0) == 0) {  // instrumented build: cycles per phase (kernels.hip.h), synchronises the device
        const int i = atoi(n.c_str() + 14);
        if (i < 0 || i > 10) return fail(DCTFP_ERR_INVALID, "walk_timeline_0 .. walk_timeline_10");
        unsigned long long v = 0;
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(&v, ctx->degenerate + 1 + i, sizeof v, hipMemcpyDeviceToHost));
        *value = (int64_t)v;
    } else if (n == "walk_trace_host") {  // address of a host copy of the trace (4 x uint64 per wave), valid until the next walk_trace
        if (!ctx->trace_dev) return fail(DCTFP_ERR_INVALID, "walk_trace_host: no trace (set walk_trace first)");
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(hipDeviceSynchronize());
        free(ctx->trace_host);
        ctx->trace_host = malloc((size_t)ctx->trace_waves * 32);
        if (!ctx->trace_host) return fail(DCTFP_ERR_NOMEM, "walk_trace_host: out of memory");
        HIP_TRY(hipMemcpy(ctx->trace_host, ctx->trace_dev, (size_t)ctx->trace_waves * 32, hipMemcpyDeviceToHost));
        *value = (int64_t)(uintptr_t)ctx->trace_host;
    }
#endif
    else return fail(DCTFP_ERR_INVALID, "unknown option '%s'", name);
    return DCTFP_OK;
} DCTFP_GUARD("dctfp_get_option")

int dctfp_profile(dctfp_ctx* ctx, double ms[2], int64_t launches[2]) try {
    if (!ctx || !ms || !launches) return fail(DCTFP_ERR_INVALID, "dctfp_profile: NULL argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    for (size_t i = 0; i < ctx->events_used; ++i) {
        EventPair& e = ctx->events[i];
        HIP_TRY(hipEventSynchronize(e.b));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, e.a, e.b));
        ctx->prof_ms[e.which] += t;
        ctx->prof_n[e.which] += 1;
    }
    ctx->events_used = 0;
    for (int i = 0; i < 2; ++i) {
        ms[i] = ctx->prof_ms[i];
        launches[i] = ctx->prof_n[i];
        ctx->prof_ms[i] = 0;
        ctx->prof_n[i] = 0;
    }
    return DCTFP_OK;
} DCTFP_GUARD("dctfp_profile")

}  // extern "C"

namespace {

constexpr uint32_t kWalkMaxRows = 8192;  // longest domain a wave of the walk kernel streams on its own

// Shapes the walk kernel takes (the rest of its conditions -- alignment, job count -- are judged per call in quantize_impl).
bool walk_shape(const dctfp_layer& ly) {
#ifdef DCTFP_GEN_ALL
    return false;
#endif
    if (ly.dtype == DCTFP_F64) return false;  // float64 rows: the general kernel (6.73 against 5.73 TB/s through this one's double build)
    // (80 < m <= 96 -- PROST's [3, 85] --: the builds with six column groups, float32 rows only; round 5)
    const int m_max = ly.dtype == DCTFP_F32 ? 96 : 80;
    return ly.n_keep == 3 && ly.m_keep > 64 && ly.m_keep <= m_max && ly.n_cols >= 512 && ly.n_cols <= 2560 && ly.n_cols % 4 == 0;
}

// Shapes the general walk kernel (walk_gen_kernel) takes: float32 / float64 rows, n = 2 .. 8, whatever width and m fit the LDS
// with one slot at 4 (float32) / 2 (float64) channels per lane -- the widest layout; an unaligned layer needs 1 channel per
// lane and is judged again per call.
bool gen_shape(const dctfp_layer& ly) {
    if (ly.n_keep < 2 || ly.m_keep < 2 || (ly.dtype != DCTFP_F32 && ly.dtype != DCTFP_F64)) return false;
    const int vec = ly.dtype == DCTFP_F32 ? 4 : 2;
    const int waves = (ly.n_cols + 64 * vec - 1) / (64 * vec);
    return waves <= 16 && gen_slot_bytes(ly.n_keep, ly.m_keep, waves, vec) + 64 <= kGenLdsBudget;
}

// Window geometry of one sequence (src/embedding.py:123-150, :185-188): see dctfp_stitch_sizes / dctfp_stitch_sequences.
// Returns the rows (embeddings) or the side (contact maps) of the stitched result, or -1 where the reference's torch
// expression would fail to broadcast.  `emit(window, dst_row_offset, n_avg)` is called per window when given.
template <typename Emit>
int64_t stitch_geometry(const int32_t* rows, int64_t n_win, int32_t step, bool square, Emit emit) {
    if (n_win < 1 || rows[0] < 1) return -1;
    int64_t size = rows[0];
    emit((int64_t)0, (int64_t)0, (int32_t)0);
    for (int64_t w = 1; w < n_win; ++w) {
        if (rows[w] < 1) return -1;
        if (!square) {      // run[-olp:] = (run[-olp:] + new[:olp]) / 2; cat(new[olp:])
            if (rows[w] <= step || size < step) return -1;
            emit(w, size - step, step);
            size += rows[w] - step;
        } else {            // combine_contacts: the window lands at offset inc * w, its overlap with the running map is averaged
            const int64_t off = (int64_t)step * w;
            if (off > size) return -1;
            emit(w, off, (int32_t)std::min<int64_t>(size - off, rows[w]));
            size = off + rows[w];
        }
    }
    return size;
}
// The walk kernels address the rows of a piece through a 32-bit buffer offset and stream a whole domain per wave.
bool walk_rows_ok(const dctfp_ctx* ctx, const dctfp_layer& g, uint32_t max_len_all) {
    return max_len_all <= kWalkMaxRows && (size_t)g.ld * dtype_size(g.dtype) <= ((size_t)1 << 31) / kWalkMaxRows && ctx->opt_stage_b == 1;
}
// ... and take a call by the "path" option: always (2), or from 256 jobs (a smaller call is latency-bound: two kernels)
bool walk_by_path(const dctfp_ctx* ctx, int64_t n_jobs) { return ctx->opt_path == 2 || (ctx->opt_path == 0 && n_jobs >= 256); }

// 16 bytes per lane where every row of every sequence (window) allows it.  seq_rows == nullptr: every entry counts.
bool rows_aligned16(const dctfp_layer* group, int ng, int64_t n_data, const int64_t* seq_rows) {
    const dctfp_layer& g = group[0];
    const size_t esz = dtype_size(g.dtype);
    if (((size_t)g.ld * esz) % 16 != 0 || g.n_cols % (int)(16 / esz) != 0) return false;
    for (int li = 0; li < ng; ++li)
        for (int64_t sq = 0; sq < n_data; ++sq)
            if ((!seq_rows || seq_rows[sq] > 0) && !aligned16(group[li].seq_data[sq])) return false;
    return true;
}

// Where the rows of a piece come from when the sequences exist only as the language model's overlapping windows
// (dctfp_quantize_windows): row r of the piece is row row_a + r of window a -- or, b >= 0, the float32 mean of that row and row
// row_b + r of window b (the rows two windows share, src/embedding.py:185-187).  seq_data is then indexed by WINDOW.
struct PieceSrc {
    int64_t a, row_a;
    int64_t b, row_b;
};

// Set by dctfp_quantize_one around its dctfp_quantize: the caller waits for the stream before anything else can touch the context.
thread_local bool tl_sync_call = false;

// dctfp_quantize proper.  `out_row` (optional): the output row of every domain of THIS piece table (a call that
// dctfp_quantize has split in two); without it domain d writes row d.  `src` (optional, one per piece; n_data = windows):
// see PieceSrc -- pieces, seq_rows and seq then speak of the STITCHED sequences.  The caller holds the context's mutex.
int quantize_impl(dctfp_ctx* ctx, const dctfp_layer* layers, int32_t n_layers, int32_t n_seq, const int64_t* seq_rows,
                  const dctfp_piece* pieces, int64_t n_pieces, int64_t n_domains, int8_t* out, int64_t out_stride,
                  hipStream_t stream, const int64_t* out_row, const PieceSrc* src = nullptr, int64_t n_data = 0) {
    if (!src) n_data = n_seq;
    bool two_source = false;  // some piece is the mean of two windows' rows: only walk_ab_kernel reads those
    if (src)
        for (int64_t i = 0; i < n_pieces && !two_source; ++i) two_source = src[i].b >= 0;

    // ---- validate the piece table, domain lengths --------------------------------
    std::vector<uint32_t> dom_len((size_t)n_domains, 0), dom_first((size_t)n_domains, 0), dom_np((size_t)n_domains, 0);
    {
        int64_t prev = -1;
        for (int64_t i = 0; i < n_pieces; ++i) {
            const dctfp_piece& pc = pieces[i];
            if (pc.domain < 0 || pc.domain >= n_domains || pc.domain < prev)
                return fail(DCTFP_ERR_INVALID, "piece %lld: domain %d out of order or range", (long long)i, pc.domain);
            if (pc.seq < 0 || pc.seq >= n_seq)
                return fail(DCTFP_ERR_INVALID, "piece %lld: sequence %d out of range", (long long)i, pc.seq);
            if (pc.n_rows <= 0 || pc.row_start < 0 || pc.row_start + pc.n_rows > seq_rows[pc.seq])
                return fail(DCTFP_ERR_INVALID, "piece %lld: rows [%lld, +%d) outside sequence %d of %lld rows",
                            (long long)i, (long long)pc.row_start, pc.n_rows, pc.seq, (long long)seq_rows[pc.seq]);
            if (pc.domain != prev) dom_first[pc.domain] = (uint32_t)i;
            if ((uint64_t)dom_len[pc.domain] + (uint64_t)pc.n_rows > 0x7fffffffu)
                return fail(DCTFP_ERR_LIMIT, "domain %d longer than 2^31 rows", pc.domain);
            dom_len[pc.domain] += (uint32_t)pc.n_rows;
            dom_np[pc.domain] += 1;
            prev = pc.domain;
        }
        for (int64_t d = 0; d < n_domains; ++d)
            if (dom_np[d] == 0) return fail(DCTFP_ERR_INVALID, "domain %lld has no piece", (long long)d);
    }
    uint32_t min_len = 0xffffffffu, max_len_all = 0;
    for (int64_t d = 0; d < n_domains; ++d) {
        min_len = std::min(min_len, dom_len[d]);
        max_len_all = std::max(max_len_all, dom_len[d]);
    }

    for (int32_t l = 0; l < n_layers; ++l) {
        const dctfp_layer& ly = layers[l];
        if (!ly.seq_data) return fail(DCTFP_ERR_INVALID, "layer %d: seq_data is NULL", l);
        if (ly.dtype < DCTFP_F32 || ly.dtype > DCTFP_BF16) return fail(DCTFP_ERR_INVALID, "layer %d: dtype %d", l, ly.dtype);
        if (ly.n_cols <= 0 || ly.ld < ly.n_cols) return fail(DCTFP_ERR_INVALID, "layer %d: n_cols %d ld %lld", l, ly.n_cols, (long long)ly.ld);
        if (ly.n_keep < 1 || ly.m_keep < 1) return fail(DCTFP_ERR_INVALID, "layer %d: qdim (%d, %d)", l, ly.n_keep, ly.m_keep);
        if (ly.n_keep > DCTFP_MAX_N || ly.m_keep > DCTFP_MAX_M)
            return fail(DCTFP_ERR_LIMIT, "layer %d: qdim (%d, %d) above the supported (%d, %d)", l, ly.n_keep, ly.m_keep, DCTFP_MAX_N, DCTFP_MAX_M);
        if (ly.out_offset < 0 || (int64_t)ly.out_offset + (int64_t)ly.n_keep * ly.m_keep > out_stride)
            return fail(DCTFP_ERR_INVALID, "layer %d: block [%d, +%d) outside out_stride %lld", l, ly.out_offset, ly.n_keep * ly.m_keep, (long long)out_stride);
        for (int64_t s = 0; s < n_data; ++s)
            if (!ly.seq_data[s] && (src || seq_rows[s] > 0))
                return fail(DCTFP_ERR_INVALID, src ? "layer %d: window %lld has no data" : "layer %d: sequence %lld has no data", l, (long long)s);
        // the reference's reshape failure (src/fingerprint.py:194): L_d < n or D < m
        if ((int64_t)min_len < ly.n_keep) {
            for (int64_t d = 0; d < n_domains; ++d)
                if ((int64_t)dom_len[d] < ly.n_keep)
                    return fail(DCTFP_ERR_SHAPE, "cannot reshape array of size %lld into shape (%d,) [domain %lld has %u rows < n = %d]",
                                (long long)dom_len[d] * std::min(ly.m_keep, ly.n_cols), ly.n_keep * ly.m_keep, (long long)d, dom_len[d], ly.n_keep);
        }
        if (ly.n_cols < ly.m_keep)
            return fail(DCTFP_ERR_SHAPE, "cannot reshape array of size %d into shape (%d,) [layer %d has %d channels < m = %d]",
                        ly.n_keep * ly.n_cols, ly.n_keep * ly.m_keep, l, ly.n_cols, ly.m_keep);
    }

    // ---- fused groups: consecutive domains of one sequence whose last domain is the whole
    // sequence and whose other domains tile it exactly (RecCut's output shape).  Their rows are
    // streamed once: every part also accumulates the whole-protein coefficients.
    std::vector<int32_t> grp_start((size_t)n_domains), grp_end((size_t)n_domains, -1);
    std::vector<uint8_t> is_whole((size_t)n_domains, 0);
    int64_t n_groups = 0;
    for (int64_t d = 0; d < n_domains; ++d) grp_start[d] = (int32_t)d;
    if (ctx->opt_fuse) {
        std::vector<std::pair<int64_t, int64_t>> runs;  // scratch: (row_start, n_rows) of the parts
        int64_t d = 0;
        while (d < n_domains) {
            const int32_t s0 = pieces[dom_first[d]].seq;
            int64_t e = d;  // run [d, e] of domains of sequence s0
            bool one_seq = true;
            while (true) {
                for (uint32_t k = 0; k < dom_np[e]; ++k)
                    if (pieces[dom_first[e] + k].seq != s0) one_seq = false;
                if (e + 1 < n_domains && pieces[dom_first[e + 1]].seq == s0) ++e;
                else break;
            }
            bool ok = one_seq && e > d;
            if (ok) {
                const dctfp_piece& w = pieces[dom_first[e]];
                ok = dom_np[e] == 1 && w.row_start == 0 && w.n_rows == seq_rows[s0];
                if (src && !ok) {  // the whole protein of a windowed sequence: one piece per window region, back to back
                    int64_t pos = 0;
                    ok = true;
                    for (uint32_t k = 0; k < dom_np[e] && ok; ++k) {
                        ok = pieces[dom_first[e] + k].row_start == pos;
                        pos += pieces[dom_first[e] + k].n_rows;
                    }
                    ok = ok && pos == seq_rows[s0];
                }
            }
            if (ok) {
                runs.clear();
                for (int64_t q = d; q < e; ++q)
                    for (uint32_t k = 0; k < dom_np[q]; ++k)
                        runs.emplace_back(pieces[dom_first[q] + k].row_start, (int64_t)pieces[dom_first[q] + k].n_rows);
                std::sort(runs.begin(), runs.end());
                int64_t pos = 0;
                for (auto& r : runs) {
                    if (r.first != pos) { ok = false; break; }
                    pos += r.second;
                }
                ok = ok && pos == seq_rows[s0];
            }
            if (ok) {
                for (int64_t q = d; q <= e; ++q) {
                    grp_start[q] = (int32_t)d;
                    grp_end[q] = (int32_t)e;
                }
                is_whole[e] = 1;
                ++n_groups;
            }
            d = e + 1;
        }
    }

    // ---- groups of consecutive layers with the same geometry ----------------------
    if ((int64_t)ctx->basis_doubles > ctx->basis_cap_doubles) {  // the cosine-table arena starts over (nothing of this call uses it yet)
        int rcp = basis_purge(ctx);
        if (rcp) return rcp;
    }
    int32_t l0 = 0;
    while (l0 < n_layers) {
        int32_t l1 = l0 + 1;
        const dctfp_layer& g = layers[l0];
        while (l1 < n_layers && layers[l1].n_cols == g.n_cols && layers[l1].dtype == g.dtype && layers[l1].ld == g.ld &&
               layers[l1].n_keep == g.n_keep && layers[l1].m_keep == g.m_keep)
            ++l1;
        const int ng = l1 - l0;
        const int n = g.n_keep, m = g.m_keep, nk = n - 1;
        const int64_t n_jobs = (int64_t)ng * n_domains;
        const size_t esz = dtype_size(g.dtype);
        const bool trivial = (n == 1 || m == 1);  // single resampled value -> 0/0 -> 0

        // staging layout (the run and cosine-table lists are bounded by the job count)
        const size_t off_jobb = 0;
        const size_t off_joba = align_up(off_jobb + (size_t)n_jobs * sizeof(JobB), 16);
        const size_t off_piece = align_up(off_joba + (size_t)n_jobs * sizeof(JobA), 16);
        const size_t off_walk = align_up(off_piece + (size_t)ng * n_pieces * sizeof(PieceA), 16);
        const size_t off_run = align_up(off_walk + (size_t)n_jobs * sizeof(Walk), 16);
        const size_t off_btab = align_up(off_run + (size_t)n_jobs * sizeof(Run), 16);
        const size_t max_bytes = align_up(off_btab + (size_t)n_domains * sizeof(BasisJob), 16);
        const int buf = ctx->flip;
        Staging& stg = ctx->staging[buf];
        DevBuf& tab = ctx->tables[buf];
        ctx->flip ^= 1;
        int rc = stg.ensure(max_bytes);
        if (rc) return rc;
        char* h = (char*)stg.p;
        JobB* hjb = (JobB*)(h + off_jobb);
        JobA* hja = (JobA*)(h + off_joba);
        PieceA* hpc = (PieceA*)(h + off_piece);
        Walk* hwalk = (Walk*)(h + off_walk);
        Run* hrun = (Run*)(h + off_run);
        BasisJob* hbt = (BasisJob*)(h + off_btab);

        // one cosine table per distinct domain length, from the context's cache
        BasisRollback basis_guard(ctx);  // until the fresh tables are filled and published
        std::vector<BasisJob> fresh;
        std::vector<double*> dom_tab((size_t)n_domains, nullptr);
        if (!trivial) {
            LenTable seen(max_len_all);  // length -> index of the first domain with it
            for (int64_t d = 0; d < n_domains; ++d) {
                if (seen.has(dom_len[d])) {
                    dom_tab[d] = dom_tab[seen[dom_len[d]]];
                } else {
                    rc = basis_lookup(ctx, dom_len[d], nk, &dom_tab[d], fresh);
                    if (rc) return rc;
                    seen.set(dom_len[d], (uint32_t)d);
                }
            }
        }
        for (size_t i = 0; i < fresh.size(); ++i) hbt[i] = fresh[i];

        const int ldy_pre = (int)align_up((size_t)g.n_cols, 32);
        // 16 bytes per lane where every row of every sequence allows it
        const bool vec_ok = rows_aligned16(layers + l0, ng, n_data, src ? nullptr : seq_rows);
        const int vec_want = (int)(16 / esz);  // 16 bytes per lane
        // ---- which kernels.  The walk kernel (stage A + B in one launch, nothing but int8 written) takes the production
        // shapes: n = 3, 64 < m <= 80 (five 16-column groups), rows read 4 channels per lane, 512 <= D <= 2560, no giant domain
        // (a wave streams all rows of its channels).  Every other shape of float32 / float64 rows that fits the LDS goes to the
        // general walk kernel (round 4); the rest -- and calls too small to fill the chip -- run stage A -> Y' -> stage B.
        // (rows are addressed through a 32-bit buffer offset: a piece of at most kWalkMaxRows rows stays below 2^31 bytes)
        const bool rows_ok = walk_rows_ok(ctx, g, max_len_all);
        const bool walk_ok = !trivial && walk_shape(g) && vec_ok && rows_ok;
        const bool use_walk = walk_ok && walk_by_path(ctx, n_jobs);
        if (two_source && !(use_walk && g.dtype == DCTFP_F32 && m <= 80))  // (dctfp_quantize_windows has asked takes_two_sources() before anything was launched)
            return fail(DCTFP_ERR_UNSUPPORTED, "internal: two-source pieces outside the walk kernel");
        int gen_vec = 0, gen_waves = 0, gen_slots = 0;
        if (!trivial && !walk_ok && rows_ok && n >= 2 && m >= 2 && (g.dtype == DCTFP_F32 || g.dtype == DCTFP_F64)) {
            gen_vec = vec_ok ? vec_want : 1;
            gen_waves = (g.n_cols + 64 * gen_vec - 1) / (64 * gen_vec);
            const size_t slot = gen_slot_bytes(n, m, gen_waves, gen_vec);
            gen_slots = gen_waves <= 16 && slot + 64 <= kGenLdsBudget ? 1 : 0;
            // A second slot (a wave writes the next job's Y' while the last arrival of this one still sums) only where it
            // costs no workgroup per CU: resident workgroups hide the end of a job (epilogue, contraction, row sums), and a
            // wave reaches its next write a whole job's stream after the last one anyway.
            if (gen_slots == 1) {
                const size_t by_waves = std::max<size_t>(1, 20 / (size_t)gen_waves);   // (the kernel's builds hold 5 .. 7 waves per SIMD)
                const size_t one = std::min(by_waves, kGenLdsBudget / (slot + 64)), two = std::min(by_waves, kGenLdsBudget / (2 * slot + 64));
                if (two >= one) gen_slots = 2;
            }
        }
        // (a shape whose slot leaves fewer than eight waves resident per CU -- [8, 128] at D = 1280: one workgroup of five -- streams
        //  at 3.8 TB/s there against 5.4 through the two kernels: only when asked for)
        const int64_t gen_resident = gen_slots > 0 ? gen_waves * std::min<int64_t>(std::max<int64_t>(1, 20 / gen_waves), (int64_t)(kGenLdsBudget / (gen_slots * gen_slot_bytes(n, m, gen_waves, gen_vec) + 64))) : 0;
        // (... and it streams every job on its own: where proteins come as parts + whole protein, the fused stage A of the two
        //  kernels reads the rows once -- 3.6-4.7 against 2.1-2.3 TB/s on the c4 / c5 mixes at [5, 44] / [3, 85] / [4, 80],
        //  tools/gen_probe.py; on whole-protein batches the general kernel is 2-6 % ahead)
        // Round 5: the general kernel has FUSED builds for n <= 5 (its walks then read the rows of a protein once, as the tuned
        // kernel's); with them and four k-steps of stage-B fragments in flight it streams the c4 / c5 mixes at [5, 44] / [3, 85] /
        // [4, 80] at 2.6-3.3 TB/s (2.1-2.5 before) -- and the two kernels at 3.6-5.0 (tools/gen_probe.py,
        // profiles/r05/gen_probe_fused.txt): a flush per ~ 100-row job, its Y' slot in LDS holding the workgroups per CU down, is
        // not how short jobs want to be run.  So such batches still go to the two kernels by default; "path" = 2 gets the fused walks.
        const bool would_fuse = ctx->opt_fuse && n_groups > 0;
        const bool use_gen = gen_slots > 0 && (ctx->opt_path == 2 || (ctx->opt_path == 0 && n_jobs >= 256 && gen_resident >= 8 && !would_fuse));
        const bool gen_fuse = use_gen && would_fuse && n <= kGenFusedMaxN && gen_waves <= kGenFusedMaxWaves && ctx->opt_gen_fuse && n_jobs >= 64;
        // (a small call wants parallelism, not fewer bytes: every job on its own workgroups)
        const bool fuse = !trivial && n_groups > 0 && n_jobs >= 64 && (!use_gen || gen_fuse);
        for (int li = 0; li < ng; ++li) {
            const dctfp_layer& ly = layers[l0 + li];
            for (int64_t d = 0; d < n_domains; ++d) {
                const int64_t job = (int64_t)li * n_domains + d;
                hjb[job].out_off = (out_row ? out_row[d] : d) * out_stride + ly.out_offset;
                hja[job].piece_begin = (uint32_t)((int64_t)li * n_pieces + dom_first[d]);
                hja[job].n_pieces = dom_np[d];
                hja[job].n_rows = dom_len[d];
                hja[job].reserved = 0;
                hja[job].basis = dom_tab[d];
                hja[job].w_basis = nullptr;
                hja[job].w_ref = nullptr;
                if (fuse && grp_end[d] >= 0 && !is_whole[d]) {
                    const int64_t w = grp_end[d];  // the whole-protein domain closes the group
                    hja[job].w_basis = dom_tab[w];
                    // (windows: row 0 of a sequence is row 0 of its first window -- a window is longer than the overlap)
                    hja[job].w_ref = src ? ly.seq_data[src[dom_first[w]].a] : ly.seq_data[pieces[dom_first[w]].seq];
                }
            }
            uint32_t t0 = 0;
            int32_t prev_dom = -1;
            for (int64_t i = 0; i < n_pieces; ++i) {
                const dctfp_piece& pc = pieces[i];
                if (pc.domain != prev_dom) t0 = 0;
                prev_dom = pc.domain;
                PieceA& o = hpc[(int64_t)li * n_pieces + i];
                if (src) {
                    const PieceSrc& ps = src[i];
                    o.ptr = (const char*)ly.seq_data[ps.a] + (size_t)ps.row_a * (size_t)ly.ld * esz;
                    o.ptr2 = ps.b >= 0 ? (const char*)ly.seq_data[ps.b] + (size_t)ps.row_b * (size_t)ly.ld * esz : nullptr;
                } else {
                    o.ptr = (const char*)ly.seq_data[pc.seq] + (size_t)pc.row_start * (size_t)ly.ld * esz;
                    o.ptr2 = nullptr;
                }
                o.n_rows = (uint32_t)pc.n_rows;
                o.t0 = t0;
                o.w0 = (uint32_t)pc.row_start;
                o.reserved = 0;
                t0 += (uint32_t)pc.n_rows;
            }
        }
        int vec = vec_ok ? vec_want : 1;
        // Fused walks of half-precision rows: 8 channels per lane mean two accumulator sets of 8 -- 27..37 registers per lane
        // spilled, and scratch writes beside the row stream cost far more than their bytes (the c5 mix in float16 took 19 ms
        // against 12.6 in float32).  4 channels per lane (8-byte loads) fit the registers.
        if (fuse && n == 3 && vec == 8) vec = 4;

        // Measured (profiles/r02): the walk kernel wins at every width it takes -- D = 2560 (10-wave workgroups, one per CU)
        // since its flush contracts the even and odd halves of the basis apart: 5.3 against 4.9-5.25 TB/s on config 4.
        // A small call (a protein at a time, the reference's calling pattern) is latency-bound: there the two-kernel path,
        // which spreads one job over slabs x 8 waves, finishes first.
        // walks of ALL jobs (walk kernel) -- the two-kernel path builds its walks per chunk below
        int64_t n_walks = 0, n_runs = 0;
        int walk_s = 0, walk_g = 0;
        // workgroups of the kernel the chip holds at once, per CU (LDS: 5 / 3 / 1 at 3 / 5 / 10 waves of the walk kernel)
        int64_t wg_per_cu = 1;
        if (use_walk || use_gen) {
            walk_s = use_gen ? gen_waves : (g.n_cols <= 768 ? 3 : (g.n_cols <= 1280 ? 5 : 10));
            wg_per_cu = use_gen ? std::max<int64_t>(1, std::min<int64_t>(20 / gen_waves, (int64_t)(kGenLdsBudget / (gen_slots * gen_slot_bytes(n, m, gen_waves, gen_vec) + 64))))
                                : (walk_s == 3 ? 5 : (walk_s == 5 ? 3 : 1));
            // jobs per flush: 4 = the rows of an MFMA tile (a flush costs the same MFMAs for 1..4 jobs)
            walk_g = ctx->opt_ab_group && !two_source && m <= 80 ? (int)ctx->opt_ab_group : 4;
            for (int64_t j = 0; j < n_jobs;) {
                const int64_t d = j % n_domains;
                Walk& wk = hwalk[n_walks++];
                wk.job_begin = (uint32_t)j;
                wk.reserved = 0;
                if (fuse && grp_end[d] >= 0) {  // d is the first part of a fused group
                    wk.n_parts = (uint32_t)(grp_end[d] - d);
                    wk.whole_job = (int32_t)(j + (grp_end[d] - d));
                    j += grp_end[d] - d + 1;
                } else {
                    wk.n_parts = 1;
                    wk.whole_job = -1;
                    j += 1;
                }
            }
            // runs: consecutive walks until a run holds `want` jobs (a multiple of the flush group, so that most
            // flushes are full); fewer jobs per run when the batch is small, to keep every CU busy
            int64_t want = ctx->opt_ab_run_jobs;
            bool longest_first = ctx->opt_ab_longest_first == 1;
            if (want == 0) {
                // Rows per job decide (profiles/r02/path_probe_run_jobs.log): long jobs (whole proteins) want ONE per
                // workgroup -- a flush of one job costs the MFMAs of four, nothing beside 500 rows, and 4 x as many, smaller
                // workgroups drain the chip more evenly at the end (C2 6.79 -> 6.97 TB/s, C3 6.70 -> 6.90); short jobs
                // (domains) want many per workgroup, so that few flushes are partial (c4 5.80 -> 5.91 at 16).
                int64_t rows = 0;
                for (int64_t d = 0; d < n_domains; ++d)
                    if (!(fuse && is_whole[d])) rows += dom_len[d];
                // (rows of 4-byte elements: a half-precision job of 500 rows weighs like 250 -- 4.40 ms per C2 batch with four
                //  jobs per workgroup, 4.85 with one)
                const int64_t job_rows = rows * (int64_t)esz / 4 / std::max<int64_t>(1, n_domains);  // whole-protein jobs of fused walks stream nothing
                const int64_t by_rows = job_rows >= 384 ? 1 : 4 * walk_g;
                if (ctx->opt_ab_longest_first == 0) longest_first = by_rows >= walk_g && walk_s == 10 && !use_gen;
                const int64_t slots = (int64_t)ctx->n_cu * wg_per_cu;
                if (n_jobs <= 6 * slots * by_rows) {
                    // fewer than a handful of rounds at that size: ONE round of equal workgroups instead (a second, partly
                    // filled round costs as much as a full one: 1 024 whole-protein jobs 509 us as 1 024 workgroups, 477 as 512)
                    want = std::max<int64_t>(1, (n_jobs + slots - 1) / slots);
                    if (by_rows >= walk_g && want > 1) want = (want + walk_g - 1) / walk_g * walk_g;  // short jobs: full flushes
                    want = std::min<int64_t>(want, 4 * walk_g);
                } else {
                    want = by_rows;
                    while (want > walk_g && n_jobs / want < 8192) want -= walk_g;  // ... but ten rounds of workgroups at least
                }
            }
            // The end of the launch: its last workgroups run on a chip that is emptying (tools/walk_trace.py: the last 5 % of a
            // c5 launch hold 15 % of the waves), for as long as ONE workgroup lives.  The jobs of the last round of workgroups
            // therefore go out in runs of one flush group: four times as many workgroups, a quarter as long.
            int64_t taper_from = n_jobs;  // runs that start at or after this job are short
            if (want > walk_g && ctx->opt_ab_taper) {
                const int64_t slots = (int64_t)ctx->n_cu * wg_per_cu;
                taper_from = std::max<int64_t>(0, n_jobs - slots * want * ctx->opt_ab_taper / 4);
            }
            int64_t jobs_done = 0;
            for (int64_t w = 0; w < n_walks;) {
                Run& rn = hrun[n_runs++];
                rn.walk_begin = (uint32_t)w;
                rn.job_begin = hwalk[w].job_begin;
                uint32_t jobs_in = 0;
                const int64_t want_here = jobs_done >= taper_from ? walk_g : want;
                while (w < n_walks && (jobs_in == 0 || (int64_t)jobs_in < want_here)) {
                    jobs_in += hwalk[w].n_parts + (hwalk[w].whole_job >= 0 ? 1u : 0u);
                    ++w;
                }
                // A run whose job count is no multiple of the flush group ends in a partial flush, which costs the MFMAs of a
                // full one (walks of k parts + whole protein rarely add up).  Look one or two walks further for a count that is:
                // c4 +1.9 %, c5 +0.2 %; looking further or longer runs gain nothing (profiles/r04/experiments/ab_run_alignment_and_length.txt).
                if (ctx->opt_ab_align && want_here > walk_g && jobs_in % (uint32_t)walk_g != 0) {
                    uint32_t more = jobs_in;
                    for (int64_t x = w; x < n_walks && x < w + ctx->opt_ab_align && more < jobs_in + 2u * (uint32_t)walk_g; ++x) {
                        more += hwalk[x].n_parts + (hwalk[x].whole_job >= 0 ? 1u : 0u);
                        if (more % (uint32_t)walk_g == 0) {
                            jobs_in = more;
                            w = x + 1;
                            break;
                        }
                    }
                }
                rn.n_walks = (uint32_t)(w - rn.walk_begin);
                rn.n_jobs = jobs_in;
                jobs_done += jobs_in;
            }
            // Longest run first, for batches of domains at D > 1280 (one workgroup per CU: 256 slots, so the last round
            // weighs most): the workgroups that start last are the short ones and the chip drains together (c4 +2.8 %).
            // With 1 280 slots (D = 640) it gains nothing in the kernel and costs 70 us of host time per 340 000 jobs; on
            // whole-protein batches it would put every short protein -- the jobs whose flush weighs most -- at the end
            // together (C3 -1.5 %).  Counting sort on the rows a run streams, 16-row buckets, input order within a bucket.
            if (longest_first && n_runs > 1) {
                constexpr uint32_t kBuckets = 4096;
                std::vector<uint32_t> key((size_t)n_runs), start(kBuckets + 1, 0);
                for (int64_t r = 0; r < n_runs; ++r) {
                    uint64_t rows = 0;
                    for (uint32_t w = hrun[r].walk_begin; w < hrun[r].walk_begin + hrun[r].n_walks; ++w)
                        for (uint32_t p = 0; p < hwalk[w].n_parts; ++p) rows += (uint64_t)dom_len[(hwalk[w].job_begin + p) % n_domains];
                    key[r] = kBuckets - 1 - (uint32_t)std::min<uint64_t>(rows >> 4, kBuckets - 1);  // descending
                    ++start[key[r] + 1];
                }
                for (uint32_t b = 0; b < kBuckets; ++b) start[b + 1] += start[b];
                std::vector<Run> sorted((size_t)n_runs);
                for (int64_t r = 0; r < n_runs; ++r) sorted[start[key[r]]++] = hrun[r];
                std::memcpy(hrun, sorted.data(), (size_t)n_runs * sizeof(Run));
            }
        }

        // ---- chunk plan of the two-kernel path.  The float64 scratch is a ring of `slots` regions of `sub` jobs
        // each; a chunk never splits a fused group (its whole-protein job needs every part's slab).
        struct Chunk { int64_t j0, j1, w0, wn; };
        int64_t avg_rows = 0;  // rows per streamed job (launch-shape heuristic)
        {
            int64_t rows = 0, cnt = 0;
            for (int64_t d = 0; d < n_domains; ++d)
                if (!(fuse && is_whole[d])) {
                    rows += dom_len[d];
                    ++cnt;
                }
            avg_rows = cnt ? rows / cnt : 0;
        }
        std::vector<Chunk> plan;
        // Y' per job: n float64 rows, or (n = 3 with the MFMA stage B) one float64 t row + one state byte per channel.
        // Below 512 jobs the MFMA stage B (a few workgroups walking the D channels in 80 dependent steps: ~80 us of
        // latency) loses to stage B over 64-channel slabs (stage_b_slab_kernel): 57 against 147 us at 8 jobs, 195 against
        // 245 us at 256, even at 512 (profiles/r02/midsize_probe.txt); the walk kernel takes over from 256 jobs.
        const bool small_b = n_jobs < ctx->opt_small_b_jobs && ctx->opt_stage_b == 1;
        const bool packed = ctx->opt_pack_y && n == 3 && ctx->opt_stage_b == 1 && !small_b;
        const size_t job_bytes = packed ? (size_t)ldy_pre * 9 : (size_t)n * ldy_pre * sizeof(double);
        const int n_slabs = (ldy_pre + 64 * vec - 1) / (64 * vec);
        int slots = 1;
        int64_t sub = 1;
        if (!trivial && !use_walk && !use_gen) {
            const int64_t budget_jobs = std::max<int64_t>(1, (int64_t)(((size_t)ctx->opt_ws_mb << 20) / job_bytes));
            if (ctx->opt_overlap > 1 && n_jobs >= 2048 && budget_jobs >= 2048) slots = (int)std::min<int64_t>(ctx->opt_overlap, kMaxSlots);
            int64_t max_group = 1;
            if (fuse)
                for (int64_t d = 0; d < n_domains; ++d) max_group = std::max<int64_t>(max_group, d - grp_start[d] + 1);
            // jobs one ring region may hold (a fused group always fits one region)
            int64_t region = std::max<int64_t>(1, budget_jobs / slots);
            region = std::min<int64_t>(region, (int64_t)0x7fffffff / n_slabs);
            region = std::max<int64_t>(region, max_group);
            // Chunk boundaries: equal shares (fixed-size cuts would leave a short extra chunk whose stage B runs on its
            // own at the end); when the scratch budget is the limit, as many equal chunks as needed.
            int64_t nck = slots;
            if (n_jobs / nck + max_group + 1 > region) nck = (n_jobs + region - 1) / region;
            const double shares = (double)nck;
            auto group_start = [&](int64_t j) {  // a chunk never splits a fused group
                if (fuse && j < n_jobs) {
                    const int64_t d = j % n_domains;
                    if (d != 0 && grp_start[d] < d) j -= d - grp_start[d];
                }
                return j;
            };
            std::vector<int64_t> cuts;
            for (int64_t k = 1; k < nck; ++k) {
                const int64_t prev = cuts.empty() ? 0 : cuts.back();
                int64_t j1 = std::min<int64_t>((int64_t)((double)n_jobs * (double)k / shares + 0.5), n_jobs);
                j1 = group_start(std::min<int64_t>(j1, prev + region));
                if (j1 > prev && j1 < n_jobs) cuts.push_back(j1);
            }
            cuts.push_back(n_jobs);
            int64_t nw = 0;
            size_t next_cut = 0;
            for (int64_t j0 = 0; j0 < n_jobs;) {
                while (cuts[next_cut] <= j0) ++next_cut;
                const int64_t j1 = std::max<int64_t>(j0 + 1, group_start(std::min<int64_t>(cuts[next_cut], j0 + region)));
                sub = std::max<int64_t>(sub, j1 - j0);
                Chunk ck{j0, j1, nw, 0};
                for (int64_t j = j0; j < j1;) {
                    const int64_t d = j % n_domains;
                    Walk& wk = hwalk[nw++];
                    wk.job_begin = (uint32_t)(j - j0);
                    wk.reserved = 0;
                    if (fuse && grp_end[d] >= 0) {  // d is the first part of a fused group
                        wk.n_parts = (uint32_t)(grp_end[d] - d);
                        wk.whole_job = (int32_t)(j - j0 + (grp_end[d] - d));
                        j += grp_end[d] - d + 1;
                    } else {
                        wk.n_parts = 1;
                        wk.whole_job = -1;
                        j += 1;
                    }
                    ++ck.wn;
                }
                plan.push_back(ck);
                j0 = j1;
            }
        }

        const size_t tab_bytes = align_up(off_btab + fresh.size() * sizeof(BasisJob), 16);
        rc = tab.ensure(tab_bytes);
        if (rc) return rc;
        // The tables go up on the context's copy stream, so the upload of this call overlaps the kernels of
        // the previous one; the copy waits until the last user of this table buffer (two calls ago) is done.
        rc = ctx->ensure_copy();
        if (rc) return rc;
        // (a small call keeps everything on the caller's stream: the hop through the copy stream costs two event waits,
        //  more than the upload itself)
        const bool inline_tables = tab_bytes <= (64u << 10) && n_jobs < 512;
        // ... and read the few hundred bytes of tables straight from the pinned staging buffer: an upload through the copy
        // engine costs more latency than the kernels of such a call take
        const bool zero_copy = inline_tables && stg.dev != nullptr;
        hipStream_t ts = inline_tables ? stream : ctx->copy;
        if (!zero_copy) {
            if (ctx->tab_busy[buf]) HIP_TRY(hipStreamWaitEvent(ts, ctx->ev_tab_free[buf], 0));
            HIP_TRY(hipMemcpyAsync(tab.p, stg.p, tab_bytes, hipMemcpyHostToDevice, ts));
            HIP_TRY(hipEventRecord(stg.ev, ts));
            stg.pending = true;
        }
        char* dt = zero_copy ? (char*)stg.dev : (char*)tab.p;
        const JobB* djb = (const JobB*)(dt + off_jobb);
        const JobA* dja = (const JobA*)(dt + off_joba);
        const PieceA* dpc = (const PieceA*)(dt + off_piece);
        const Walk* dwalk = (const Walk*)(dt + off_walk);
        const Run* drun = (const Run*)(dt + off_run);
        const BasisJob* dbt = (const BasisJob*)(dt + off_btab);

        if (trivial) {
            if (!inline_tables) {
                HIP_TRY(hipEventRecord(ctx->ev_tab_ready, ctx->copy));
                HIP_TRY(hipStreamWaitEvent(stream, ctx->ev_tab_ready, 0));
            }
            const int64_t total = n_jobs * n * m;
            const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 4096);
            hipLaunchKernelGGL(fill_zero_kernel, dim3(grid), dim3(256), 0, stream, djb, n_jobs, n * m, out);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(ctx->ev_tab_free[buf], stream));
            ctx->tab_busy[buf] = true;
            if (zero_copy) {  // the kernels read the staging buffer itself: it is free again after them
                HIP_TRY(hipEventRecord(stg.ev, stream));
                stg.pending = true;
            }
            l0 = l1;
            continue;
        }

        StEntry* st = nullptr;
        rc = get_st(ctx, g.n_cols, m, &st);
        if (rc) return rc;
        const int ldy = st->ldy;

#ifdef DCTFP_EXPERIMENTS
        if (ctx->test_fail_once) {  // test hook: an allocation failure between the table lookup and the fill kernel
            ctx->test_fail_once = 0;
            return fail(DCTFP_ERR_NOMEM, "injected failure (option test_fail_once)");
        }
#endif
        if (!fresh.empty()) {  // cosine tables this context has not seen yet (grid.y is limited to 65535)
            // tables cached by earlier calls may have been filled on another stream: chain the events, so that whoever
            // waits for the new ev_basis also has the older fills behind it
            if (ctx->basis_valid && ctx->basis_stream != ts) HIP_TRY(hipStreamWaitEvent(ts, ctx->ev_basis, 0));
            uint32_t max_len = 0;
            for (const BasisJob& bj : fresh) max_len = std::max(max_len, bj.len);
            const unsigned gx = (unsigned)std::min<uint64_t>(((2 * (uint64_t)max_len + 1) * nk + 255) / 256, 1024);
            for (size_t b0 = 0; b0 < fresh.size(); b0 += 65535) {
                const unsigned ny = (unsigned)std::min<size_t>(fresh.size() - b0, 65535);
                hipLaunchKernelGGL(basis_kernel, dim3(gx, ny), dim3(256), 0, ts, dbt + b0, nk);
                HIP_TRY(hipGetLastError());
            }
        }
        if (!fresh.empty()) {
            HIP_TRY(hipEventRecord(ctx->ev_basis, ts));
            ctx->basis_stream = ts;
            ctx->basis_valid = true;
            basis_publish(ctx, fresh, nk);
        }
        basis_guard.armed = false;
        if (!inline_tables) {
            HIP_TRY(hipEventRecord(ctx->ev_tab_ready, ctx->copy));
            HIP_TRY(hipStreamWaitEvent(stream, ctx->ev_tab_ready, 0));
        }
        // tables cached by an earlier call may have been filled on another stream
        if (ctx->basis_valid && ctx->basis_stream != stream) HIP_TRY(hipStreamWaitEvent(stream, ctx->ev_basis, 0));

        ctx->last_path = (use_walk || use_gen) ? 2 : 1;
        ctx->last_gen_fused = gen_fuse ? 1 : 0;
        ctx->last_walk_groups = use_walk ? (m > 80 ? 6 : 5) : 0;
        ctx->walk_launches += (use_walk || use_gen) ? 1 : 0;
        if (use_gen) {
            // one launch of the general walk kernel: stage A + stage B per workgroup, int8 out
            rc = get_st_plain(ctx, st, g.n_cols);
            if (rc) return rc;
            EventPair* ep = nullptr;
            rc = prof_begin(ctx, 0, stream, &ep);
            if (rc) return rc;
            GParams gp;
            gp.jobs = dja;
            gp.jobb = djb;
            gp.walks = dwalk;
            gp.fused = gen_fuse;
            gp.runs = drun;
            gp.pieces = dpc;
            gp.stp = st->fragp;
            gp.out = out;
            gp.n_cols = g.n_cols;
            gp.ld = g.ld;
            gp.m = m;
            gp.n_slots = gen_slots;
            gp.degenerate = ctx->degenerate;
            gp.grid = (unsigned)n_runs;
            gp.waves = (unsigned)gen_waves;
            gp.lds_bytes = (size_t)gen_slots * gen_slot_bytes(n, m, gen_waves, gen_vec) + 64;
            gp.stream = stream;
            {
                LaunchError le;
                rc = launcher_rc(launch_gen(gp, g.dtype, gen_vec, n, &le), le);
            }
            if (rc) return rc;
            HIP_TRY(hipGetLastError());
            rc = prof_end(ep, stream);
            if (rc) return rc;
            HIP_TRY(hipEventRecord(ctx->ev_tab_free[buf], stream));
            ctx->tab_busy[buf] = true;
            if (zero_copy) {  // the kernels read the staging buffer itself: it is free again after them
                HIP_TRY(hipEventRecord(stg.ev, stream));
                stg.pending = true;
            }
            l0 = l1;
            continue;
        }
        if (use_walk) {
            // one launch: stage A + stage B per workgroup, int8 out
            EventPair* ep = nullptr;
            rc = prof_begin(ctx, 0, stream, &ep);
            if (rc) return rc;
            WParams wp;
            wp.jobs = dja;
            wp.jobb = djb;
            wp.walks = dwalk;
            wp.runs = drun;
            wp.pieces = dpc;
            wp.stf = st->frag;
            wp.out = out;
            wp.n_cols = g.n_cols;
            wp.ld = g.ld;
            wp.m = m;
            wp.degenerate = ctx->degenerate;
            wp.grid = (unsigned)n_runs;
            wp.stream = stream;
            wp.two_source = two_source;
            {
                LaunchError le;
                rc = launcher_rc(launch_walk(wp, g.dtype, walk_s, walk_g, ctx->opt_ab_unroll && !two_source ? (int)ctx->opt_ab_unroll : 8, fuse,
                                             ctx->opt_ab_mfma_a != 0 && !two_source, &le), le);
            }
            if (rc) return rc;
            HIP_TRY(hipGetLastError());
            rc = prof_end(ep, stream);
            if (rc) return rc;
            HIP_TRY(hipEventRecord(ctx->ev_tab_free[buf], stream));
            ctx->tab_busy[buf] = true;
            if (zero_copy) {  // the kernels read the staging buffer itself: it is free again after them
                HIP_TRY(hipEventRecord(stg.ev, stream));
                stg.pending = true;
            }
            l0 = l1;
            continue;
        }

        // Stage A of chunk c runs on the caller's stream, stage B of it on the context's side
        // stream, so the MFMA-bound stage B of one chunk overlaps the HBM-bound stage A of the next.
        if (ldy != ldy_pre) return fail(DCTFP_ERR_INVALID, "internal: basis width mismatch");
        rc = ctx->ws.ensure((size_t)sub * slots * job_bytes);
        if (rc) return rc;
        const bool side = slots > 1;
        if (side) {
            rc = ctx->ensure_side();
            if (rc) return rc;
        }
        hipStream_t sb = side ? ctx->side : stream;
        // the scratch is the context's: wait for whatever call used it last, on whatever stream
        if (ctx->ws_busy) HIP_TRY(hipStreamWaitEvent(stream, ctx->ev_ws_free, 0));

        int64_t c = 0;
        for (const Chunk& ck : plan) {
            const int64_t j0 = ck.j0, jn = ck.j1 - ck.j0;
            const int slot = (int)(c % slots);
            char* yprime = (char*)ctx->ws.p + (size_t)slot * sub * job_bytes;
            if (side && c >= slots) HIP_TRY(hipStreamWaitEvent(stream, ctx->ev_b[slot], 0));  // slot free again?
            EventPair* ep = nullptr;
            rc = prof_begin(ctx, 0, stream, &ep);
            if (rc) return rc;
            // small call: stage B over 64-channel slabs (stage_b_slab_kernel), their partial blocks in the split scratch
            const int n_kslabs = (g.n_cols + kSlabChannels - 1) / kSlabChannels;
            auto zpart_bytes = [&](int64_t jobs_here) { return (size_t)jobs_here * n_kslabs * n * m * sizeof(double); };
            double* zpart = nullptr;
            bool one_launch = false;   // small_call_kernel has done stage A, stage B and the int8 rows of this chunk
            {
                AParams ap;
                ap.jobs = dja + j0;
                ap.walks = dwalk + ck.w0;
                ap.fused = fuse;
                ap.pieces = dpc;
                ap.degenerate = ctx->degenerate;
                ap.yprime = yprime;
                ap.job_bytes = (int64_t)job_bytes;
                ap.packed = packed ? 1 : 0;
                ap.n_cols = g.n_cols;
                ap.ld = g.ld;
                ap.ldy = ldy;
                ap.n_slabs = n_slabs;
                ap.grid = (unsigned)(ck.wn * n_slabs);
                ap.stream = stream;
                // a call that cannot fill the chip: split the rows of every job over workgroups (stage_a_split_kernel)
                const bool split = !fuse && n == 3 && g.dtype == DCTFP_F32 && vec == 4 && ctx->opt_a_waves == 0 &&
                                   jn * n_slabs < 128 && avg_rows >= 128 && max_len_all <= (1u << 24);
                if (split) {
                    int64_t want_chunks = std::min<int64_t>(32, std::max<int64_t>(2, 384 / (jn * n_slabs)));
                    uint32_t chunk_rows = (uint32_t)((max_len_all + want_chunks - 1) / want_chunks);
                    chunk_rows = std::max<uint32_t>(32, (chunk_rows + 31) / 32 * 32);  // 8 waves x 4 rows in flight
                    const int n_chunks = (int)((max_len_all + chunk_rows - 1) / chunk_rows);
                    const size_t partial_bytes = (size_t)jn * n_chunks * nk * ldy * sizeof(double);
                    // (small_call_kernel keeps a 3 x 80 block per job and 256-channel slab behind the partial sums)
                    rc = ctx->split_ws.ensure(partial_bytes + (small_b ? std::max(zpart_bytes(jn), (size_t)jn * n_slabs * 3 * 80 * sizeof(double)) : 0));
                    if (rc) return rc;
                    static const InvTab<3> inv3 = make_inv<3>();
                    // Round 5: the three steps of a small call in ONE launch (small_call_kernel: they hand over by tickets).
                    // n = 3, m <= 80 (five column groups of fragments per k-step in registers): the production shape.
                    if (small_b && m <= 80 && ctx->opt_small_one) {
                        rc = get_st_plain(ctx, st, g.n_cols);
                        if (rc) return rc;
                        constexpr size_t kTickets = 1024;   // jn * n_slabs < 128 here: (n_slabs + 1) counters per job
                        if (!ctx->small_tickets) {
                            HIP_TRY(hipMalloc((void**)&ctx->small_tickets, kTickets * sizeof(uint32_t)));
                            HIP_TRY(hipMemset(ctx->small_tickets, 0, kTickets * sizeof(uint32_t)));
                        }
                        if ((size_t)jn * (n_slabs + 1) <= kTickets) {
                            zpart = (double*)((char*)ctx->split_ws.p + partial_bytes);   // (jn x n_slabs blocks of 3 x 80: below zpart_bytes(jn))
                            hipLaunchKernelGGL((small_call_kernel<8, 4>), dim3((unsigned)(jn * n_chunks * n_slabs)), dim3(512), 0, stream,
                                               dja + j0, djb + j0, dpc, (double*)ctx->split_ws.p, zpart, ctx->small_tickets, (int)jn, n_chunks,
                                               chunk_rows, g.n_cols, g.ld, ldy, n_slabs, (const double*)st->fragp, m, inv3, ctx->degenerate, out);
                            HIP_TRY(hipGetLastError());
                            one_launch = true;
                        }
                    }
                    if (!one_launch)
                    hipLaunchKernelGGL((stage_a_split_kernel<float, 3, 4, 8, 4>), dim3((unsigned)(jn * n_chunks * n_slabs)), dim3(512), 0, stream,
                                       dja + j0, dpc, (double*)ctx->split_ws.p, n_chunks, chunk_rows, g.n_cols, g.ld, ldy, n_slabs);
                    HIP_TRY(hipGetLastError());
                    if (one_launch) {
                    } else if (small_b) {  // the slabs of stage B add the chunks and scale their channels themselves
                        zpart = (double*)((char*)ctx->split_ws.p + partial_bytes);
                        hipLaunchKernelGGL((stage_b_slab_kernel<true>), dim3((unsigned)n_kslabs, (unsigned)jn), dim3(256), 0, stream,
                                           (const double*)nullptr, ldy, (const double*)ctx->split_ws.p, n_chunks, inv3, ctx->degenerate,
                                           g.n_cols, (const double*)st->dev, st->cp, n, m, zpart);
                    } else {
                        hipLaunchKernelGGL((stage_a_combine_kernel<3>), dim3((unsigned)((ldy + 255) / 256), (unsigned)jn), dim3(256), 0, stream,
                                           (const double*)ctx->split_ws.p, n_chunks, yprime, (int64_t)job_bytes, packed ? 1 : 0, g.n_cols, ldy,
                                           inv3, ctx->degenerate);
                    }
                    HIP_TRY(hipGetLastError());
                }
                int waves = (int)ctx->opt_a_waves;
                int unroll = (int)ctx->opt_a_unroll;
                if (waves == 0) {  // auto: short walks want more, smaller workgroups per CU
                    waves = avg_rows >= 320 ? 8 : (avg_rows >= 160 ? 4 : 2);
                    // a call that cannot fill the chip (a protein at a time) is bound by the latency of one workgroup:
                    // as many waves and rows in flight as a workgroup can have
                    if (ck.wn * n_slabs < 256 && avg_rows >= 128 && vec == 4) {
                        waves = 16;
                        if (!fuse) unroll = 8;
                    }
                    if (vec == 8 && waves > 4) waves = 4;  // 8 channels per lane: keep the LDS reduction buffer small
                }
                if (!split) launch_a(ap, g.dtype, vec, n, waves, unroll);
                HIP_TRY(hipGetLastError());
            }
            rc = prof_end(ep, stream);
            if (rc) return rc;
            if (side) {
                HIP_TRY(hipEventRecord(ctx->ev_a[slot], stream));
                HIP_TRY(hipStreamWaitEvent(sb, ctx->ev_a[slot], 0));
            }

            rc = prof_begin(ctx, 1, sb, &ep);
            if (rc) return rc;
            ctx->last_small_one = one_launch ? 1 : 0;
            if (one_launch) {
            } else if (ctx->opt_stage_b == 1 && !small_b) {
                const int64_t rows = jn * n;
                launch_b_mfma(st->cp / 16, packed, (unsigned)((rows + kBWaves * 16 - 1) / (kBWaves * 16)), sb, yprime, (int64_t)job_bytes, rows, ldy,
                              st->dev, djb + j0, n, m, out);
            } else if (small_b) {
                if (!zpart) {  // stage A wrote Y' (no row split): the slabs read it
                    rc = ctx->split_ws.ensure(zpart_bytes(jn));
                    if (rc) return rc;
                    zpart = (double*)ctx->split_ws.p;
                    static const InvTab<3> inv3 = make_inv<3>();
                    hipLaunchKernelGGL((stage_b_slab_kernel<false>), dim3((unsigned)n_kslabs, (unsigned)jn), dim3(256), 0, sb,
                                       (const double*)yprime, ldy, (const double*)nullptr, 0, inv3, ctx->degenerate, g.n_cols,
                                       (const double*)st->dev, st->cp, n, m, zpart);
                    HIP_TRY(hipGetLastError());
                }
                hipLaunchKernelGGL(stage_b_finish_kernel, dim3((unsigned)jn), dim3(256), 0, sb, (const double*)zpart, n_kslabs, djb + j0, n, m, out);
            } else {
                hipLaunchKernelGGL(stage_b_valu_kernel, dim3((unsigned)jn), dim3(1024), 0, sb, (const double*)yprime, ldy, g.n_cols,
                                   st->dev, st->cp, djb + j0, n, m, out);
            }
            HIP_TRY(hipGetLastError());
            rc = prof_end(ep, sb);
            if (rc) return rc;
            if (side) HIP_TRY(hipEventRecord(ctx->ev_b[slot], sb));
            ++c;
        }
        if (side) {  // the caller's stream continues only after every stage B of this group
            const int64_t used = std::min<int64_t>(c, slots);
            for (int64_t k = 0; k < used; ++k) HIP_TRY(hipStreamWaitEvent(stream, ctx->ev_b[k], 0));
        }
        // (dctfp_quantize_one waits for the stream before it returns: scratch, tables and staging buffer ARE free for whoever
        //  comes next, on whatever stream -- three event records, 4-5 us of a 56-us call, say nothing it does not already know)
        // -- for the LAST layer group of the call only: an earlier group's buffers are taken again by a later group of the same call
        // (five layers of five geometries: the staging buffer of group 1 is group 3's), long before the call's wait.
        if (!(tl_sync_call && l1 == n_layers)) {
            HIP_TRY(hipEventRecord(ctx->ev_ws_free, stream));  // ... and the scratch is free for the next call after this point
            ctx->ws_busy = true;
            HIP_TRY(hipEventRecord(ctx->ev_tab_free[buf], stream));  // this table buffer may be overwritten after this point
            ctx->tab_busy[buf] = true;
            if (zero_copy) {
                HIP_TRY(hipEventRecord(stg.ev, stream));
                stg.pending = true;
            }
        }
        l0 = l1;
    }
    return DCTFP_OK;
}

}  // namespace

extern "C" {

int dctfp_quantize(dctfp_ctx* ctx, const dctfp_layer* layers, int32_t n_layers, int32_t n_seq,
                   const int64_t* seq_rows, const dctfp_piece* pieces, int64_t n_pieces, int64_t n_domains,
                   int8_t* out, int64_t out_stride, void* stream_v) try {
    if (!ctx) return fail(DCTFP_ERR_INVALID, "dctfp_quantize: ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (n_layers < 0 || n_seq < 0 || n_pieces < 0 || n_domains < 0)
        return fail(DCTFP_ERR_INVALID, "dctfp_quantize: negative count");
    if (n_layers == 0 || n_domains == 0) return DCTFP_OK;
    if (!layers || !seq_rows || !pieces || !out) return fail(DCTFP_ERR_INVALID, "dctfp_quantize: NULL argument");
    if (n_pieces >= (int64_t)1 << 31 || n_domains >= (int64_t)1 << 31)
        return fail(DCTFP_ERR_LIMIT, "dctfp_quantize: more than 2^31 pieces or domains in one call");
    hipStream_t stream = (hipStream_t)stream_v;
    HIP_TRY(hipSetDevice(ctx->device));

    // A domain above kWalkMaxRows rows (a titin in a flush of 10 000 proteins) must not take the whole call off the walk
    // kernel: such domains are cut out into a call of their own (two-kernel path), everything else stays where it was.
    // Only when every layer has the walk kernel's shape and the rest of the call is large enough to be sent there.
    bool all_walk = ctx->opt_path != 1;
    for (int32_t l = 0; l < n_layers && all_walk; ++l) all_walk = walk_shape(layers[l]) || gen_shape(layers[l]);
    if (all_walk && n_domains * n_layers >= 256) {
        std::vector<uint32_t> len((size_t)n_domains, 0);
        bool table_ok = true;
        for (int64_t i = 0; i < n_pieces && table_ok; ++i) {
            const dctfp_piece& pc = pieces[i];
            table_ok = pc.domain >= 0 && pc.domain < n_domains && pc.n_rows > 0 && (uint64_t)len[pc.domain] + (uint64_t)pc.n_rows <= 0x7fffffffu;
            if (table_ok) len[pc.domain] += (uint32_t)pc.n_rows;
        }
        int64_t n_giant = 0;
        for (int64_t d = 0; d < n_domains && table_ok; ++d) n_giant += len[d] > kWalkMaxRows ? 1 : 0;
        if (table_ok && n_giant > 0 && n_giant < n_domains) {  // (a broken table goes to quantize_impl as it is: it reports the error)
            std::vector<dctfp_piece> part[2];
            std::vector<int64_t> rows[2], new_id((size_t)n_domains);
            for (int64_t d = 0; d < n_domains; ++d) {
                const int w = len[d] > kWalkMaxRows ? 1 : 0;
                new_id[d] = (int64_t)rows[w].size();
                rows[w].push_back(d);
            }
            for (int64_t i = 0; i < n_pieces; ++i) {
                dctfp_piece pc = pieces[i];
                const int w = len[pc.domain] > kWalkMaxRows ? 1 : 0;
                pc.domain = (int32_t)new_id[pc.domain];
                part[w].push_back(pc);
            }
            for (int w = 0; w < 2; ++w) {
                const int rc = quantize_impl(ctx, layers, n_layers, n_seq, seq_rows, part[w].data(), (int64_t)part[w].size(),
                                             (int64_t)rows[w].size(), out, out_stride, stream, rows[w].data());
                if (rc) return rc;
            }
            return DCTFP_OK;
        }
    }
    return quantize_impl(ctx, layers, n_layers, n_seq, seq_rows, pieces, n_pieces, n_domains, out, out_stride, stream, nullptr);
} DCTFP_GUARD("dctfp_quantize")

int dctfp_quantize_windows(dctfp_ctx* ctx, const dctfp_layer* layers, int32_t n_layers, int32_t n_seq, const int64_t* seq_win,
                           const int32_t* win_rows, int32_t overlap, const dctfp_piece* pieces, int64_t n_pieces,
                           int64_t n_domains, int8_t* out, int64_t out_stride, void* stream_v) try {
    if (!ctx) return fail(DCTFP_ERR_INVALID, "dctfp_quantize_windows: ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (n_layers < 0 || n_seq < 0 || n_pieces < 0 || n_domains < 0 || overlap < 0)
        return fail(DCTFP_ERR_INVALID, "dctfp_quantize_windows: negative count");
    if (n_layers == 0 || n_domains == 0) return DCTFP_OK;
    if (!layers || !seq_win || !win_rows || !pieces || !out) return fail(DCTFP_ERR_INVALID, "dctfp_quantize_windows: NULL argument");
    if (n_pieces >= (int64_t)1 << 30 || n_domains >= (int64_t)1 << 31)
        return fail(DCTFP_ERR_LIMIT, "dctfp_quantize_windows: too many pieces or domains in one call");
    hipStream_t stream = (hipStream_t)stream_v;
    HIP_TRY(hipSetDevice(ctx->device));

    // ---- where every window lands in its stitched sequence (the geometry of dctfp_stitch_sequences)
    if (seq_win[0] != 0) return fail(DCTFP_ERR_INVALID, "dctfp_quantize_windows: seq_win[0] must be 0");
    for (int32_t s = 0; s < n_seq; ++s)
        if (seq_win[s + 1] < seq_win[s]) return fail(DCTFP_ERR_INVALID, "dctfp_quantize_windows: seq_win must not decrease");
    const int64_t n_win = seq_win[n_seq];
    std::vector<int64_t> seq_rows((size_t)n_seq), win_off((size_t)n_win);
    bool simple = true;  // every row is one window's row or the mean of two
    for (int32_t s = 0; s < n_seq; ++s) {
        const int64_t w0 = seq_win[s], n = seq_win[s + 1] - w0;
        const int64_t size = n > 0 ? stitch_geometry(win_rows + w0, n, overlap, false, [&](int64_t w, int64_t off, int32_t) { win_off[(size_t)(w0 + w)] = off; }) : 0;
        if (size < 0) return fail(DCTFP_ERR_SHAPE, "sequence %d: a window is not longer than the overlap", s);
        seq_rows[(size_t)s] = size;
        for (int64_t w = w0 + 1; w + 1 < w0 + n; ++w)
            if (win_rows[w] < 2 * (int64_t)overlap) simple = false;
    }
    if (!simple)
        return fail(DCTFP_ERR_UNSUPPORTED, "dctfp_quantize_windows: three windows meet in one row (a window between two others has fewer than "
                                           "2 x %d rows): stitch with dctfp_stitch_sequences, then dctfp_quantize", overlap);

    // ---- the caller's pieces (stitched rows) cut at the region borders: a run of one window's own rows, or of rows two share
    std::vector<dctfp_piece> sub;
    std::vector<PieceSrc> src;
    sub.reserve((size_t)n_pieces + (size_t)n_pieces / 2);
    src.reserve((size_t)n_pieces + (size_t)n_pieces / 2);
    bool two_source = false;
    for (int64_t i = 0; i < n_pieces; ++i) {
        const dctfp_piece& pc = pieces[i];
        if (pc.seq < 0 || pc.seq >= n_seq) return fail(DCTFP_ERR_INVALID, "piece %lld: sequence %d out of range", (long long)i, pc.seq);
        if (pc.n_rows <= 0 || pc.row_start < 0 || pc.row_start + pc.n_rows > seq_rows[(size_t)pc.seq])
            return fail(DCTFP_ERR_INVALID, "piece %lld: rows [%lld, +%d) outside sequence %d of %lld stitched rows", (long long)i,
                        (long long)pc.row_start, pc.n_rows, pc.seq, (long long)seq_rows[(size_t)pc.seq]);
        const int64_t w0 = seq_win[pc.seq], n = seq_win[pc.seq + 1] - w0;
        const int64_t* off = win_off.data() + w0;
        // window w owns the stitched rows [off[w], off[w + 1]): its first `overlap` rows (w > 0) shared with w - 1, the rest its own
        int64_t w = std::upper_bound(off, off + n, pc.row_start) - off - 1;
        int64_t pos = pc.row_start, left = pc.n_rows;
        while (left > 0) {
            while (w + 1 < n && off[w + 1] <= pos) ++w;
            const int64_t local = pos - off[w];
            dctfp_piece sp = pc;
            PieceSrc ps;
            int64_t take;
            if (w > 0 && local < overlap) {  // (old + new) / 2: old = the predecessor's tail, new = this window's head
                take = std::min<int64_t>(left, overlap - local);
                ps.a = w0 + w - 1;
                ps.row_a = win_rows[w0 + w - 1] - overlap + local;
                ps.b = w0 + w;
                ps.row_b = local;
                two_source = true;
            } else {
                const int64_t own_end = win_rows[w0 + w] - (w + 1 < n ? overlap : 0);
                take = std::min<int64_t>(left, own_end - local);
                ps.a = w0 + w;
                ps.row_a = local;
                ps.b = -1;
                ps.row_b = 0;
            }
            if (take <= 0) return fail(DCTFP_ERR_INVALID, "internal: window geometry of sequence %d", pc.seq);
            sp.row_start = pos;
            sp.n_rows = (int32_t)take;
            sub.push_back(sp);
            src.push_back(ps);
            pos += take;
            left -= take;
        }
    }

    // ---- only walk_ab_kernel averages two windows in its row load: a call it would not get is refused before anything runs
    if (two_source) {
        std::vector<uint32_t> len((size_t)n_domains, 0);
        uint32_t max_len = 0;
        for (const dctfp_piece& pc : sub)
            if (pc.domain >= 0 && pc.domain < n_domains) max_len = std::max(max_len, len[(size_t)pc.domain] += (uint32_t)pc.n_rows);
        for (int32_t l0 = 0; l0 < n_layers;) {
            const dctfp_layer& g = layers[l0];
            int32_t l1 = l0 + 1;
            while (l1 < n_layers && layers[l1].n_cols == g.n_cols && layers[l1].dtype == g.dtype && layers[l1].ld == g.ld &&
                   layers[l1].n_keep == g.n_keep && layers[l1].m_keep == g.m_keep)
                ++l1;
            const char* why = nullptr;
            if (!g.seq_data || g.dtype != DCTFP_F32) why = "windows are averaged in float32 (as dctfp_stitch)";
            else if (!walk_shape(g) || g.m_keep > 80) why = "kept sizes / width outside the one-launch kernel's (n = 3, 64 < m <= 80, 512 <= D <= 2560, D % 4 == 0)";
            else if (!rows_aligned16(layers + l0, l1 - l0, n_win, nullptr)) why = "rows are not 16-byte aligned";
            else if (!walk_rows_ok(ctx, g, max_len)) why = "a domain above 8 192 rows";
            else if (!walk_by_path(ctx, (int64_t)(l1 - l0) * n_domains)) why = "fewer than 256 jobs (layers x domains) in the call";
            if (why)
                return fail(DCTFP_ERR_UNSUPPORTED, "dctfp_quantize_windows: layer %d: %s -- stitch with dctfp_stitch_sequences, then dctfp_quantize", l0, why);
            l0 = l1;
        }
    }
    return quantize_impl(ctx, layers, n_layers, n_seq, seq_rows.data(), sub.data(), (int64_t)sub.size(), n_domains, out, out_stride, stream,
                         nullptr, src.data(), n_win);
} DCTFP_GUARD("dctfp_quantize_windows")

int dctfp_quantize_one(dctfp_ctx* ctx, const dctfp_layer* layers, int32_t n_layers, int64_t n_rows, const char* dom_text,
                       int64_t text_len, int32_t n_strings, int8_t* out, int64_t out_rows, int64_t out_stride, int32_t* str_row,
                       uint8_t* str_changed, char* key_text, int64_t key_cap, int64_t* key_len, int64_t* n_domains,
                       int64_t* n_other, int32_t* degenerate_seen, void* stream_v) try {
    if (!ctx || !dom_text || !str_row || !str_changed || !key_len || !n_domains || !n_other)
        return fail(DCTFP_ERR_INVALID, "dctfp_quantize_one: NULL argument");
    if (n_strings < 0 || n_rows < 0 || text_len < 0) return fail(DCTFP_ERR_INVALID, "dctfp_quantize_one: negative size");
    *n_domains = *n_other = *key_len = 0;
    if (degenerate_seen) *degenerate_seen = 0;
    // the domain strings of this protein -> pieces (the reference's get_doms clean-up, src/fingerprint.py:163-169)
    int64_t commas = 0;
    for (int64_t i = 0; i < text_len; ++i) commas += dom_text[i] == ',';
    std::vector<dctfp_piece> pieces((size_t)(n_strings + commas + 1));
    std::vector<int64_t> str_len((size_t)std::max(n_strings, 1));
    int64_t n_pieces = 0;
    const int32_t count = n_strings;
    int rc = dctfp_build_pieces(dom_text, text_len, &count, &n_rows, 1, pieces.data(), (int64_t)pieces.size(), &n_pieces, str_row, str_len.data(),
                                str_changed, key_text, key_cap, key_len, n_domains, n_other);
    if (rc) return rc;
    if (*n_other > 0 || *n_domains == 0 || n_layers == 0) return DCTFP_OK;   // (strings for the caller's own parser / nothing to do)
    if (*n_domains > out_rows) return fail(DCTFP_ERR_LIMIT, "dctfp_quantize_one: %lld domains, room for %lld", (long long)*n_domains, (long long)out_rows);
    hipStream_t stream = (hipStream_t)stream_v;
    {
        // (the constant-channel flag is the context's: what an earlier caller left unread is not about THIS protein)
        std::lock_guard<std::mutex> lock(ctx->mu);
        if (ctx->flag_host) __atomic_store_n(ctx->flag_host, 0u, __ATOMIC_RELEASE);
    }
    tl_sync_call = true;
    rc = dctfp_quantize(ctx, layers, n_layers, 1, &n_rows, pieces.data(), n_pieces, *n_domains, out, out_stride, stream_v);
    tl_sync_call = false;
    if (rc) {
        (void)hipStreamSynchronize(stream);   // (whatever was enqueued before the failure must not outlive the buffers it reads)
        return rc;
    }
    HIP_TRY(hipStreamSynchronize(stream));
    if (degenerate_seen && ctx->flag_host) *degenerate_seen = (int32_t)__atomic_exchange_n(ctx->flag_host, 0u, __ATOMIC_ACQ_REL);
    return DCTFP_OK;
} DCTFP_GUARD("dctfp_quantize_one")

int dctfp_idct_quant(dctfp_ctx* ctx, const void* vec, int32_t dtype, int64_t n_rows, int64_t n_cols, int64_t ld,
                     int32_t num, double* scaled_out, double* coef_out, void* stream_v) try {
    if (!ctx || !vec) return fail(DCTFP_ERR_INVALID, "dctfp_idct_quant: NULL argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (dtype != DCTFP_F32 && dtype != DCTFP_F64) return fail(DCTFP_ERR_INVALID, "dctfp_idct_quant: dtype %d", dtype);
    if (n_rows < 1 || n_cols < 1 || ld < n_cols || num < 1) return fail(DCTFP_ERR_INVALID, "dctfp_idct_quant: bad shape");
    if (num > n_rows) return fail(DCTFP_ERR_SHAPE, "dctfp_idct_quant: num %d > %lld rows", num, (long long)n_rows);
    if (num > 65535) return fail(DCTFP_ERR_LIMIT, "dctfp_idct_quant: num %d", num);
    hipStream_t stream = (hipStream_t)stream_v;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = ctx->scratch.ensure((size_t)num * n_cols * sizeof(double));
    if (rc) return rc;
    double* fs = (double*)ctx->scratch.p;
    dim3 grid((unsigned)((n_cols + 63) / 64), (unsigned)num);
    if (dtype == DCTFP_F32)
        hipLaunchKernelGGL((generic_forward_kernel<float>), grid, dim3(64), 0, stream, (const float*)vec, n_rows, n_cols, ld, num, fs, coef_out);
    else
        hipLaunchKernelGGL((generic_forward_kernel<double>), grid, dim3(64), 0, stream, (const double*)vec, n_rows, n_cols, ld, num, fs, coef_out);
    HIP_TRY(hipGetLastError());
    if (scaled_out) {
        hipLaunchKernelGGL(generic_inverse_kernel, dim3((unsigned)((n_cols + 63) / 64)), dim3(64), 0, stream, fs, n_cols, num, scaled_out);
        HIP_TRY(hipGetLastError());
    }
    return DCTFP_OK;
} DCTFP_GUARD("dctfp_idct_quant")

int dctfp_scale(dctfp_ctx* ctx, const double* vec, int64_t n, double* out, void* stream_v) try {
    if (!ctx || !vec || !out) return fail(DCTFP_ERR_INVALID, "dctfp_scale: NULL argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (n < 1) return fail(DCTFP_ERR_INVALID, "dctfp_scale: empty vector");
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(scale_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream_v, vec, n, out);
    HIP_TRY(hipGetLastError());
    return DCTFP_OK;
} DCTFP_GUARD("dctfp_scale")

int dctfp_gather_rows(dctfp_ctx* ctx, const void* embed, int32_t dtype, int64_t n_rows, int64_t n_cols, int64_t ld,
                      const dctfp_piece* pieces, int64_t n_pieces, double* out, void* stream_v) try {
    if (!ctx || !embed || !pieces || !out) return fail(DCTFP_ERR_INVALID, "dctfp_gather_rows: NULL argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (dtype != DCTFP_F32 && dtype != DCTFP_F64) return fail(DCTFP_ERR_INVALID, "dctfp_gather_rows: dtype %d", dtype);
    if (n_pieces < 1 || n_pieces > 65535 || n_cols < 1 || ld < n_cols) return fail(DCTFP_ERR_INVALID, "dctfp_gather_rows: bad shape");
    hipStream_t stream = (hipStream_t)stream_v;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t esz = dtype == DCTFP_F32 ? 4 : 8;
    const int buf = ctx->flip;
    Staging& stg = ctx->staging[buf];
    DevBuf& tab = ctx->tables[buf];
    ctx->flip ^= 1;
    int rc = stg.ensure((size_t)n_pieces * sizeof(PieceA));
    if (rc) return rc;
    PieceA* h = (PieceA*)stg.p;
    uint64_t t0 = 0;
    uint32_t max_rows = 0;
    for (int64_t i = 0; i < n_pieces; ++i) {
        const dctfp_piece& pc = pieces[i];
        if (pc.n_rows <= 0 || pc.row_start < 0 || pc.row_start + pc.n_rows > n_rows)
            return fail(DCTFP_ERR_INVALID, "dctfp_gather_rows: piece %lld outside the matrix", (long long)i);
        h[i].ptr = (const char*)embed + (size_t)pc.row_start * (size_t)ld * esz;
        h[i].n_rows = (uint32_t)pc.n_rows;
        h[i].t0 = (uint32_t)t0;
        t0 += (uint64_t)pc.n_rows;
        max_rows = std::max(max_rows, (uint32_t)pc.n_rows);
        if (t0 > 0x7fffffffu) return fail(DCTFP_ERR_LIMIT, "dctfp_gather_rows: more than 2^31 rows");
    }
    rc = tab.ensure((size_t)n_pieces * sizeof(PieceA));
    if (rc) return rc;
    // (the table buffer may still be read by kernels another stream runs -- a flush's cutter on its side stream while the caller's
    //  stream stitches the next proteins: overwrite it only behind them)
    if (ctx->tab_busy[buf]) HIP_TRY(hipStreamWaitEvent(stream, ctx->ev_tab_free[buf], 0));
    HIP_TRY(hipMemcpyAsync(tab.p, stg.p, (size_t)n_pieces * sizeof(PieceA), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(stg.ev, stream));
    stg.pending = true;
    const unsigned gx = (unsigned)std::min<int64_t>(((int64_t)max_rows * n_cols + 255) / 256, 2048);
    if (dtype == DCTFP_F32)
        hipLaunchKernelGGL((gather_rows_kernel<float>), dim3(gx, (unsigned)n_pieces), dim3(256), 0, stream, (const PieceA*)tab.p, (int)n_pieces, n_cols, ld, out);
    else
        hipLaunchKernelGGL((gather_rows_kernel<double>), dim3(gx, (unsigned)n_pieces), dim3(256), 0, stream, (const PieceA*)tab.p, (int)n_pieces, n_cols, ld, out);
    HIP_TRY(hipGetLastError());
    return mark_table_used(ctx, buf, stream);
} DCTFP_GUARD("dctfp_gather_rows")


int dctfp_contact_topk(dctfp_ctx* ctx, const void* const* maps, const int64_t* ld, const int32_t* n_res,
                       int32_t n_prot, double t, int32_t* out_i, int32_t* out_j, float* out_v,
                       const int64_t* out_offs, int32_t* out_n, void* stream_v) try {
    if (!ctx || !maps || !ld || !n_res || !out_i || !out_j || !out_v || !out_offs || !out_n)
        return fail(DCTFP_ERR_INVALID, "dctfp_contact_topk: NULL argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (n_prot < 0) return fail(DCTFP_ERR_INVALID, "dctfp_contact_topk: negative count");
    if (n_prot == 0) return DCTFP_OK;
    hipStream_t stream = (hipStream_t)stream_v;
    HIP_TRY(hipSetDevice(ctx->device));
    // long proteins (a million candidate pairs and more; a quarter of that when the call is too small to fill the chip
    // with one workgroup per protein) are spread over many workgroups, the others get one each
    // (round 5: a quarter of a million pairs whatever the size of the call -- L >= 730.  Up to a million pairs such a protein used
    //  to get one workgroup: the one-read kernel's sample misjudges long stitched maps now and then (their top contacts sit in a
    //  narrow band of a mostly empty triangle), the two-read kernel's thread minima bound them loosely near the top of its k
    //  ranges, and what both hand back met the radix select on ONE workgroup -- 4.5 ms for a 1 100-residue protein, with the whole
    //  flush waiting for it.  Spread over stripes it is 6 reads by many workgroups.)
    const int64_t kLongPairs = (int64_t)1 << 18;
    constexpr int64_t kStripePairs = (int64_t)1 << 17;
    // (... and whatever the one-read kernel does not take -- k > 3 000: L > 1 153 at t = 2.6 -- with half a million pairs or more:
    //  behind it such a protein met the two-read kernel, whose candidates overflow on banded maps, and then the radix select on
    //  ONE workgroup: 4.5 ms for a 1 300-residue protein, the whole flush waiting for it)
    auto is_long = [&](int64_t L, int64_t cand) {
        const int64_t k = dctfp_contact_count((int32_t)L, t);
        return k > 0 && (cand >= kLongPairs || (k > 3000 && cand >= ((int64_t)1 << 19)));
    };
    std::vector<int32_t> order((size_t)n_prot);
    int32_t n_short = 0, n_long = 0;
    int64_t n_stripes = 0;
    for (int32_t p = 0; p < n_prot; ++p) {
        const int64_t L = n_res[p];
        if (L < 0 || (L > 0 && (!maps[p] || ld[p] < L)))
            return fail(DCTFP_ERR_INVALID, "dctfp_contact_topk: protein %d: bad map", p);
        const int64_t cand = L >= 6 ? (L - 5) * (L - 4) / 2 : 0;
        if (is_long(L, cand)) {
            ++n_long;
            n_stripes += std::min<int64_t>(512, (cand + kStripePairs - 1) / kStripePairs);
        } else {
            ++n_short;
        }
    }
    {
        int32_t a = 0, b = n_short;
        for (int32_t p = 0; p < n_prot; ++p) {
            const int64_t L = n_res[p];
            const int64_t cand = L >= 6 ? (L - 5) * (L - 4) / 2 : 0;
            if (is_long(L, cand)) order[b++] = p;
            else order[a++] = p;
        }
    }
    const size_t off_stripe = align_up((size_t)n_prot * sizeof(TopkJob), 16);
    const size_t off_first = align_up(off_stripe + (size_t)n_stripes * sizeof(TopkStripe), 16);
    const size_t off_state = align_up(off_first + (size_t)n_long * sizeof(int32_t), 16);
    const size_t up_bytes = align_up(off_state + (size_t)n_long * sizeof(TopkState), 16);
    const size_t off_ties = up_bytes;
    const size_t all_bytes = off_ties + (size_t)n_stripes * sizeof(int32_t);
    const int buf = ctx->flip;
    Staging& stg = ctx->staging[buf];
    DevBuf& tab = ctx->tables[buf];
    ctx->flip ^= 1;
    int rc = stg.ensure(up_bytes);
    if (rc) return rc;
    TopkJob* h = (TopkJob*)stg.p;
    TopkStripe* hs = (TopkStripe*)((char*)stg.p + off_stripe);
    int32_t* hfirst = (int32_t*)((char*)stg.p + off_first);
    TopkState* hstate = (TopkState*)((char*)stg.p + off_state);
    int64_t s_fill = 0;
    for (int32_t q = 0; q < n_prot; ++q) {
        const int32_t p = order[q];
        const int64_t L = n_res[p];
        h[q].map = (const float*)maps[p];
        h[q].ld = ld[p];
        h[q].n_res = (int32_t)L;
        h[q].k = (int32_t)dctfp_contact_count((int32_t)L, t);
        h[q].out_off = out_offs[p];
        h[q].orig = p;
        h[q].reserved = 0;
        if (out_offs[p + 1] - out_offs[p] < h[q].k)
            return fail(DCTFP_ERR_INVALID, "dctfp_contact_topk: protein %d: output room %lld < %d", p,
                        (long long)(out_offs[p + 1] - out_offs[p]), h[q].k);
        if (q >= n_short) {  // stripes of about equal numbers of candidate pairs (row i has L - 5 - i of them)
            const int32_t lj = q - n_short;
            const int64_t cand = (L - 5) * (L - 4) / 2;
            const int64_t ns = std::min<int64_t>(512, (cand + kStripePairs - 1) / kStripePairs);
            hfirst[lj] = (int32_t)s_fill;
            memset(&hstate[lj], 0, sizeof(TopkState));
            hstate[lj].need = h[q].k;
            int64_t acc = 0;
            int32_t row = 0, stripe = 0;
            const int32_t last_row = (int32_t)L - 5;  // rows 0 .. L-6 have candidates
            for (int64_t k = 0; k < ns; ++k) {
                const int64_t goal = cand * (k + 1) / ns;
                const int32_t begin = row;
                while (row < last_row && (acc < goal || k + 1 == ns)) {
                    acc += L - 5 - row;
                    ++row;
                }
                if (row > begin) {
                    TopkStripe& sp = hs[s_fill++];
                    sp.job = lj;
                    sp.row_begin = begin;
                    sp.row_end = row;
                    sp.stripe = stripe++;
                }
            }
        }
    }
    const int64_t used_stripes = s_fill;  // (a stripe can come out empty when rows are long: it is simply not emitted)
    rc = tab.ensure(all_bytes);
    if (rc) return rc;
    // (the table buffer may still be read by kernels another stream runs -- a flush's cutter on its side stream while the caller's
    //  stream stitches the next proteins: overwrite it only behind them)
    if (ctx->tab_busy[buf]) HIP_TRY(hipStreamWaitEvent(stream, ctx->ev_tab_free[buf], 0));
    HIP_TRY(hipMemcpyAsync(tab.p, stg.p, up_bytes, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(stg.ev, stream));
    stg.pending = true;
    const TopkJob* djobs = (const TopkJob*)tab.p;
    // The selection of the long proteins (stripes: four histogram passes, the collection, the ties) and that of the short ones
    // (one workgroup each: one read, what it hands back, ...) touch different proteins: they run side by side -- the long chain on
    // a stream of its own, forked here (the tables are up) and joined before the caller's stream goes on.  A flush of 2 048
    // proteins spent 2.8 ms in the two chains one after the other, 0.9 ms of kernels each and a dozen launch gaps
    // (profiles/r05/flush_kernel_stats_tiefree.txt).
    const bool long_beside = n_long > 0 && used_stripes > 0 && n_short > 0;
    const hipStream_t short_stream = stream;
    if (long_beside) {
        rc = ctx->ensure_cut_streams();
        if (rc) return rc;
        HIP_TRY(hipEventRecord(ctx->cut_ev[0], stream));
        HIP_TRY(hipStreamWaitEvent(ctx->cut_stream[0], ctx->cut_ev[0], 0));
    }
    if (n_short > 0) {
        if (ctx->opt_topk_kernel != 1) {   // one read of the map (0; 2 = straight to the two-read kernel of round 4); what a kernel hands
            // back (out_n = -1) the next one redoes: the two-read kernel, then the radix select
            if (ctx->opt_topk_kernel == 0) {   // (each build leaves the jobs outside its range of k to the other)
                constexpr int kSmallK = 1400;
                bool any_small = false, any_large = false;
                for (int32_t q = 0; q < n_short; ++q) (h[q].k <= kSmallK ? any_small : any_large) = true;
                if (any_small)
                    hipLaunchKernelGGL((contact_topk1_kernel<8>), dim3((unsigned)n_short), dim3(512), 0, stream, djobs, out_i, out_j, out_v, out_n, 0, kSmallK);
                if (any_large)
                    hipLaunchKernelGGL((contact_topk1_kernel<16>), dim3((unsigned)n_short), dim3(1024), 0, stream, djobs, out_i, out_j, out_v, out_n,
                                       kSmallK + 1, 0x7fffffff);
            }
            hipLaunchKernelGGL(contact_topk2_kernel, dim3((unsigned)n_short), dim3(1024), 0, stream, djobs, out_i, out_j, out_v, out_n,
                               ctx->opt_topk_kernel == 0 ? 1 : 0);
            hipLaunchKernelGGL(contact_topk_kernel, dim3((unsigned)n_short), dim3(1024), 0, stream, djobs, out_i, out_j, out_v, out_n, 1);
        } else {
            hipLaunchKernelGGL(contact_topk_kernel, dim3((unsigned)n_short), dim3(1024), 0, stream, djobs, out_i, out_j, out_v, out_n, 0);
        }
        HIP_TRY(hipGetLastError());
    }
    if (long_beside) stream = ctx->cut_stream[0];
    if (n_long > 0 && used_stripes > 0) {
        const TopkJob* dlong = djobs + n_short;
        const TopkStripe* dstripes = (const TopkStripe*)((char*)tab.p + off_stripe);
        const int32_t* dfirst = (const int32_t*)((char*)tab.p + off_first);
        TopkState* dstate = (TopkState*)((char*)tab.p + off_state);
        int32_t* dties = (int32_t*)((char*)tab.p + off_ties);
        for (int shift = 24; shift >= 0; shift -= 8) {
            hipLaunchKernelGGL(topk_hist_kernel, dim3((unsigned)used_stripes), dim3(1024), 0, stream, dlong, dstripes, dstate, shift);
            hipLaunchKernelGGL(topk_pick_kernel, dim3((unsigned)n_long), dim3(64), 0, stream, dlong, dstate, shift, out_n);
        }
        hipLaunchKernelGGL(topk_collect_kernel, dim3((unsigned)used_stripes), dim3(1024), 0, stream, dlong, dstripes, dstate, dties, out_i,
                           out_j, out_v);
        hipLaunchKernelGGL(topk_ties_kernel, dim3((unsigned)used_stripes), dim3(64), 0, stream, dlong, dstripes, dstate, dties, dfirst,
                           out_i, out_j, out_v);
        HIP_TRY(hipGetLastError());
    }
    if (long_beside) {
        HIP_TRY(hipEventRecord(ctx->cut_ev[1], stream));
        stream = short_stream;
        HIP_TRY(hipStreamWaitEvent(stream, ctx->cut_ev[1], 0));
    }
    return mark_table_used(ctx, buf, stream);
} DCTFP_GUARD("dctfp_contact_topk")

// The order of the CON line (src/fingerprint.py:58-61) on the device: see include/dctfp.h.
int dctfp_contact_sort(dctfp_ctx* ctx, const void* const* maps, const int64_t* ld, const int32_t* n_res, int32_t n_prot,
                       double t, int32_t* out_i, int32_t* out_j, float* out_v, const int64_t* out_offs, uint8_t* sorted,
                       void* stream_v) try {
    if (!ctx || !maps || !ld || !n_res || !out_i || !out_j || !out_v || !out_offs || !sorted)
        return fail(DCTFP_ERR_INVALID, "dctfp_contact_sort: NULL argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (n_prot < 0) return fail(DCTFP_ERR_INVALID, "dctfp_contact_sort: negative count");
    if (n_prot == 0) return DCTFP_OK;
    hipStream_t stream = (hipStream_t)stream_v;
    HIP_TRY(hipSetDevice(ctx->device));
    // proteins grouped by the size of the sorting network that holds their k entries
    static const int kNet[4] = {2048, 4096, 8192, 16384};
    std::vector<int32_t> group[4];
    for (int32_t p = 0; p < n_prot; ++p) {
        const int64_t k = dctfp_contact_count(n_res[p], t);
        sorted[p] = 1;
        if (k <= 1) continue;                       // nothing to order
        if (n_res[p] > 65536 || k > kNet[3]) {      // (i, j) do not fit 16 bits each / more entries than the LDS holds
            sorted[p] = 0;
            continue;
        }
        if (!maps[p] || ld[p] < n_res[p]) return fail(DCTFP_ERR_INVALID, "dctfp_contact_sort: protein %d: bad map", p);
        int g = 0;
        while (k > kNet[g]) ++g;
        group[g].push_back(p);
    }
    const size_t n_jobs = group[0].size() + group[1].size() + group[2].size() + group[3].size();
    if (n_jobs == 0) return DCTFP_OK;
    const int buf = ctx->flip;
    Staging& stg = ctx->staging[buf];
    DevBuf& tab = ctx->tables[buf];
    ctx->flip ^= 1;
    int rc = stg.ensure(n_jobs * sizeof(TopkJob));
    if (rc) return rc;
    rc = tab.ensure(n_jobs * sizeof(TopkJob));
    if (rc) return rc;
    TopkJob* h = (TopkJob*)stg.p;
    size_t q = 0;
    for (int g = 0; g < 4; ++g)
        for (int32_t p : group[g]) {
            h[q].map = (const float*)maps[p];
            h[q].ld = ld[p];
            h[q].n_res = n_res[p];
            h[q].k = (int32_t)dctfp_contact_count(n_res[p], t);
            h[q].out_off = out_offs[p];
            h[q].orig = p;
            h[q].reserved = 0;
            ++q;
        }
    if (ctx->tab_busy[buf]) HIP_TRY(hipStreamWaitEvent(stream, ctx->ev_tab_free[buf], 0));
    HIP_TRY(hipMemcpyAsync(tab.p, stg.p, n_jobs * sizeof(TopkJob), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(stg.ev, stream));
    stg.pending = true;
    const TopkJob* d = (const TopkJob*)tab.p;
    size_t first = 0;
    for (int g = 0; g < 4; ++g) {
        const unsigned n = (unsigned)group[g].size();
        if (n == 0) continue;
        if (g == 0) hipLaunchKernelGGL((contact_sort_kernel<2048>), dim3(n), dim3(1024), 0, stream, d + first, out_i, out_j, out_v);
        else if (g == 1) hipLaunchKernelGGL((contact_sort_kernel<4096>), dim3(n), dim3(1024), 0, stream, d + first, out_i, out_j, out_v);
        else if (g == 2) hipLaunchKernelGGL((contact_sort_kernel<8192>), dim3(n), dim3(1024), 0, stream, d + first, out_i, out_j, out_v);
        else hipLaunchKernelGGL((contact_sort_kernel<16384>), dim3(n), dim3(1024), 0, stream, d + first, out_i, out_j, out_v);
        HIP_TRY(hipGetLastError());
        first += n;
    }
    return mark_table_used(ctx, buf, stream);
} DCTFP_GUARD("dctfp_contact_sort")

int64_t dctfp_reccut_room(int32_t n_res) {
    // {status / n_domains} + per domain {n_segs} + per segment {first, last}: domains hold >= 22 residues, a cut adds at most two
    // segments
    const int64_t doms = n_res > 0 ? n_res / kCutMinSize + 1 : 1;
    return 2 + doms + 2 * (2 * doms + 1);
}

int dctfp_reccut(dctfp_ctx* ctx, const int32_t* n_res, int32_t n_prot, const int32_t* ci, const int32_t* cj, const float* cv,
                 const int64_t* offs, double cut1, double cut2, int32_t* out, const int64_t* out_offs, void* stream_v) try {
    if (!ctx || !n_res || !offs || !out || !out_offs) return fail(DCTFP_ERR_INVALID, "dctfp_reccut: NULL argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (n_prot < 0) return fail(DCTFP_ERR_INVALID, "dctfp_reccut: negative count");
    if (n_prot == 0) return DCTFP_OK;
    hipStream_t stream = (hipStream_t)stream_v;
    HIP_TRY(hipSetDevice(ctx->device));
    // jobs by LDS class (512 / 1024 / 1536 / 2048 residues), each class one launch; scratch: adjacency lists + node stacks
    std::vector<int32_t> order((size_t)n_prot);
    int32_t count[kCutClasses] = {};
    std::vector<uint8_t> cls((size_t)n_prot);
    size_t adj_total = 0;
    for (int32_t p = 0; p < n_prot; ++p) {
        const int64_t nc = offs[p + 1] - offs[p];
        if (n_res[p] < 0 || nc < 0 || nc > 0x7fffffff || (nc > 0 && (!ci || !cj || !cv)))
            return fail(DCTFP_ERR_INVALID, "dctfp_reccut: protein %d: bad sizes", p);
        if (out_offs[p + 1] - out_offs[p] < 4) return fail(DCTFP_ERR_INVALID, "dctfp_reccut: protein %d: output room below 4", p);
        cls[(size_t)p] = (uint8_t)reccut_class_of(n_res[p], nc);
        ++count[cls[(size_t)p]];
        adj_total += 2 * (size_t)nc + 6 * (size_t)std::max(n_res[p], 0);
    }
    int32_t first[kCutClasses];
    {
        int32_t at[kCutClasses], run = 0;
        for (int c = 0; c < kCutClasses; ++c) {
            first[c] = at[c] = run;
            run += count[c];
        }
        for (int32_t p = 0; p < n_prot; ++p) order[(size_t)at[cls[(size_t)p]]++] = p;
    }
    const size_t stack_ints = (size_t)kCutStack * kCutNodeInts;
    const size_t ws_bytes = align_up(adj_total * sizeof(uint32_t), 16) + (size_t)n_prot * stack_ints * sizeof(int32_t);
    int rc = ctx->cut_ws.ensure(ws_bytes);
    if (rc) return rc;
    // (adjacency lists and node stacks are the context's: a call on another stream waits for the kernels of the last one)
    if (!ctx->ev_cut_ws_free) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_cut_ws_free, hipEventDisableTiming));
    if (ctx->cut_ws_busy) HIP_TRY(hipStreamWaitEvent((hipStream_t)stream_v, ctx->ev_cut_ws_free, 0));
    const int buf = ctx->flip;
    Staging& stg = ctx->staging[buf];
    DevBuf& tab = ctx->tables[buf];
    ctx->flip ^= 1;
    rc = stg.ensure((size_t)n_prot * sizeof(CutJob));
    if (rc) return rc;
    rc = tab.ensure((size_t)n_prot * sizeof(CutJob));
    if (rc) return rc;
    CutJob* h = (CutJob*)stg.p;
    uint32_t* adj = (uint32_t*)ctx->cut_ws.p;
    int32_t* stacks = (int32_t*)((char*)ctx->cut_ws.p + align_up(adj_total * sizeof(uint32_t), 16));
    size_t adj_at = 0;
    for (int32_t q = 0; q < n_prot; ++q) {
        const int32_t p = order[(size_t)q];
        const int64_t nc = offs[p + 1] - offs[p];
        CutJob& j = h[q];
        j.ci = ci ? ci + offs[p] : nullptr;
        j.cj = cj ? cj + offs[p] : nullptr;
        j.cv = cv ? cv + offs[p] : nullptr;
        j.adj = adj + adj_at;
        adj_at += 2 * (size_t)nc + 6 * (size_t)std::max(n_res[p], 0);
        j.stack = stacks + (size_t)q * stack_ints;
        j.out = out + out_offs[p];
        j.timing = nullptr;
#ifdef DCTFP_CUT_TIMING
        j.timing = ctx->degenerate + 1;   // (instrumented build: the context's spare device counters)
#endif
        j.n_contacts = (int32_t)nc;
        j.n_res = n_res[p];
        j.out_cap = (int32_t)std::min<int64_t>(out_offs[p + 1] - out_offs[p], 0x7fffffff);
        j.reserved = 0;
    }
    // (the table buffer may still be read by kernels another stream runs -- a flush's cutter on its side stream while the caller's
    //  stream stitches the next proteins: overwrite it only behind them)
    if (ctx->tab_busy[buf]) HIP_TRY(hipStreamWaitEvent(stream, ctx->ev_tab_free[buf], 0));
    HIP_TRY(hipMemcpyAsync(tab.p, stg.p, (size_t)n_prot * sizeof(CutJob), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(stg.ev, stream));
    stg.pending = true;
    const CutJob* d = (const CutJob*)tab.p;
    // A class's launch lasts as long as its slowest protein (one workgroup each): the larger classes run on streams of their
    // own beside the small one, and the caller's stream continues after all of them.
    int n_used = 0;
    for (int c = 0; c < kCutClasses; ++c) n_used += count[c] > 0 ? 1 : 0;
    const bool beside = n_used > 1;
    if (beside) {
        rc = ctx->ensure_cut_streams();
        if (rc) return rc;
        HIP_TRY(hipEventRecord(ctx->cut_ev[0], stream));
    }
    // (the small class first: its many short workgroups fill the chip three per CU and drain within a few hundred microseconds; the
    //  long classes, on high-priority streams, take the CUs as they come free.  The other way round the 80-155 KB workgroups of the
    //  long classes held the LDS of most CUs and the small class ran in what was left: 3.3 ms for kernels of 1.8 / 1.7 / 0.8 ms)
    for (int c = 0; c < kCutClasses; ++c) {
        if (count[c] == 0) continue;
        hipStream_t s = beside && c > 0 ? ctx->cut_stream[c - 1] : stream;
        if (s != stream) HIP_TRY(hipStreamWaitEvent(s, ctx->cut_ev[0], 0));
        for (int32_t done = 0; done < count[c]; done += 65535 * 16) {
            const unsigned n = (unsigned)std::min<int32_t>(count[c] - done, 65535 * 16);
            LaunchError le;
            rc = launcher_rc(launch_reccut(c, d + first[c] + done, n, cut1, cut2, s, &le), le);
            if (rc) return rc;
            HIP_TRY(hipGetLastError());
        }
        if (s != stream) {
            HIP_TRY(hipEventRecord(ctx->cut_ev[c], s));
            HIP_TRY(hipStreamWaitEvent(stream, ctx->cut_ev[c], 0));
        }
    }
    HIP_TRY(hipEventRecord(ctx->ev_cut_ws_free, stream));
    ctx->cut_ws_busy = true;
    return mark_table_used(ctx, buf, stream);
} DCTFP_GUARD("dctfp_reccut")

}  // extern "C"

namespace {
// dctfp_stitch proper; the caller holds the context's mutex.
// `once` (embeddings only): per window the predecessor's row that meets its row 0, that window's row stride, and the rows at its
// end that its successor writes -- all windows then go out in one launch (see StitchJob).
struct StitchOnce {
    const float* prev;
    int64_t ld_prev;
    int32_t n_skip;
};
int stitch_impl(dctfp_ctx* ctx, const dctfp_stitch_job* jobs, int64_t n_jobs, int32_t n_cols, int32_t square, void* stream_v,
                const StitchOnce* once = nullptr) {
    if (n_jobs < 0 || (!square && n_cols < 1)) return fail(DCTFP_ERR_INVALID, "dctfp_stitch: bad count");
    if (n_jobs == 0) return DCTFP_OK;
    if (n_jobs > 65535 * 64) return fail(DCTFP_ERR_LIMIT, "dctfp_stitch: too many windows in one call");
    hipStream_t stream = (hipStream_t)stream_v;
    HIP_TRY(hipSetDevice(ctx->device));
    int32_t max_level = 0;
    bool vec4 = !square && n_cols % 4 == 0;
    for (int64_t i = 0; i < n_jobs; ++i) {
        const dctfp_stitch_job& j = jobs[i];
        if (!aligned16(j.src) || !aligned16(j.dst) || (j.ld_src % 4) || (j.ld_dst % 4)) vec4 = false;
        if (once && once[i].prev && (!aligned16(once[i].prev) || (once[i].ld_prev % 4))) vec4 = false;
        if (!j.src || !j.dst || j.n_rows < 1 || j.n_avg < 0 || j.n_avg > j.n_rows || j.level < 0 ||
            j.ld_src < (square ? j.n_rows : n_cols) || j.ld_dst < (square ? j.n_rows : n_cols))
            return fail(DCTFP_ERR_INVALID, "dctfp_stitch: window %lld is malformed", (long long)i);
        if (j.level == 0 && j.n_avg != 0) return fail(DCTFP_ERR_INVALID, "dctfp_stitch: window %lld: level 0 cannot average", (long long)i);
        max_level = std::max(max_level, j.level);
    }
    if (once) max_level = 0;   // every window in the one launch
    const int buf = ctx->flip;
    Staging& stg = ctx->staging[buf];
    DevBuf& tab = ctx->tables[buf];
    ctx->flip ^= 1;
    int rc = stg.ensure((size_t)n_jobs * sizeof(StitchJob));
    if (rc) return rc;
    rc = tab.ensure((size_t)n_jobs * sizeof(StitchJob));
    if (rc) return rc;
    // bucket the windows by level (stable), one launch per level in ascending order
    StitchJob* h = (StitchJob*)stg.p;
    std::vector<int64_t> start((size_t)max_level + 2, 0);
    for (int64_t i = 0; i < n_jobs; ++i) start[(size_t)(once ? 0 : jobs[i].level) + 1] += 1;
    for (int32_t l = 0; l <= max_level; ++l) start[(size_t)l + 1] += start[(size_t)l];
    std::vector<int64_t> fill(start.begin(), start.end() - 1);
    std::vector<int32_t> max_rows((size_t)max_level + 1, 0);
    for (int64_t i = 0; i < n_jobs; ++i) {
        const dctfp_stitch_job& j = jobs[i];
        const int32_t level = once ? 0 : j.level;
        StitchJob& o = h[fill[(size_t)level]++];
        o.src = (const float*)j.src;
        o.dst = (float*)j.dst;
        o.n_rows = j.n_rows;
        o.n_avg = j.n_avg;
        o.ld_src = j.ld_src;
        o.ld_dst = j.ld_dst;
        o.prev = once ? once[i].prev : nullptr;
        o.ld_prev = once ? once[i].ld_prev : 0;
        o.n_skip = once ? once[i].n_skip : 0;
        o.reserved = 0;
        max_rows[(size_t)level] = std::max(max_rows[(size_t)level], j.n_rows);
    }
    // (the table buffer may still be read by kernels another stream runs -- a flush's cutter on its side stream while the caller's
    //  stream stitches the next proteins: overwrite it only behind them)
    if (ctx->tab_busy[buf]) HIP_TRY(hipStreamWaitEvent(stream, ctx->ev_tab_free[buf], 0));
    HIP_TRY(hipMemcpyAsync(tab.p, stg.p, (size_t)n_jobs * sizeof(StitchJob), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(stg.ev, stream));
    stg.pending = true;
    const StitchJob* d = (const StitchJob*)tab.p;
    for (int32_t l = 0; l <= max_level; ++l) {
        int64_t cnt = start[(size_t)l + 1] - start[(size_t)l];
        int64_t done = 0;
        while (done < cnt) {  // grid.y is limited to 65535
            const unsigned ny = (unsigned)std::min<int64_t>(cnt - done, 65535);
            dim3 grid((unsigned)((max_rows[(size_t)l] + 15) / 16), ny);
            if (square) hipLaunchKernelGGL(stitch_contacts_kernel, grid, dim3(256), 0, stream, d + start[(size_t)l] + done);
            else if (vec4) hipLaunchKernelGGL((stitch_rows_kernel<true>), grid, dim3(256), 0, stream, d + start[(size_t)l] + done, n_cols);
            else hipLaunchKernelGGL((stitch_rows_kernel<false>), grid, dim3(256), 0, stream, d + start[(size_t)l] + done, n_cols);
            HIP_TRY(hipGetLastError());
            done += ny;
        }
    }
    return mark_table_used(ctx, buf, stream);
}

}  // namespace

extern "C" {

int dctfp_stitch(dctfp_ctx* ctx, const dctfp_stitch_job* jobs, int64_t n_jobs, int32_t n_cols, int32_t square,
                 void* stream_v) try {
    if (!ctx || (!jobs && n_jobs > 0)) return fail(DCTFP_ERR_INVALID, "dctfp_stitch: NULL argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    return stitch_impl(ctx, jobs, n_jobs, n_cols, square, stream_v);
} DCTFP_GUARD("dctfp_stitch")

int dctfp_stitch_sizes(const int32_t* win_rows, const int64_t* seq_win, int64_t n_seq, int32_t step, int32_t square,
                       int64_t* out_rows) try {
    if (!win_rows || !seq_win || !out_rows || n_seq < 0 || step < 0) return fail(DCTFP_ERR_INVALID, "dctfp_stitch_sizes: bad argument");
    for (int64_t s = 0; s < n_seq; ++s) {
        const int64_t n = seq_win[s + 1] - seq_win[s];
        out_rows[s] = stitch_geometry(win_rows + seq_win[s], n, step, square != 0, [](int64_t, int64_t, int32_t) {});
        if (out_rows[s] < 0)
            return fail(DCTFP_ERR_SHAPE, square ? "sequence %lld: window offset beyond the running contact map"
                                                : "sequence %lld: a window is not longer than the overlap", (long long)s);
    }
    return DCTFP_OK;
} DCTFP_GUARD("dctfp_stitch_sizes")

int dctfp_stitch_sequences(dctfp_ctx* ctx, const void* const* win, const int32_t* win_rows, const int64_t* win_ld,
                           const int64_t* seq_win, int64_t n_seq, void* const* dst, const int64_t* dst_ld, int32_t n_cols,
                           int32_t step, int32_t square, void* stream_v) try {
    if (!ctx || !win || !win_rows || !win_ld || !seq_win || !dst || !dst_ld) return fail(DCTFP_ERR_INVALID, "dctfp_stitch_sequences: NULL argument");
    if (n_seq < 0 || step < 0) return fail(DCTFP_ERR_INVALID, "dctfp_stitch_sequences: bad count");
    std::lock_guard<std::mutex> lock(ctx->mu);
    std::vector<dctfp_stitch_job> jobs;
    jobs.reserve((size_t)(n_seq > 0 ? seq_win[n_seq] - seq_win[0] : 0));
    // Embeddings: where no window's averaged head reaches back beyond its predecessor's own rows that nobody else averaged
    // (rows >= 2 * step for every window with a successor and a predecessor -- every shape Embedding.embed_seq produces unless
    // maxlen < 2 * 200), the rows two windows share are averaged from the two windows and all of them go out in one launch.
    bool simple = !square && ctx->opt_stitch_once != 2;
    for (int64_t s = 0; s < n_seq && simple; ++s)
        for (int64_t w = seq_win[s] + 1; w + 1 < seq_win[s + 1]; ++w)
            if (win_rows[w] < 2 * (int64_t)step) simple = false;
    std::vector<StitchOnce> once;
    for (int64_t s = 0; s < n_seq; ++s) {
        const int64_t w0 = seq_win[s], n = seq_win[s + 1] - w0;
        const int64_t size = stitch_geometry(win_rows + w0, n, step, square != 0, [&](int64_t w, int64_t off, int32_t n_avg) {
            dctfp_stitch_job j;
            j.src = win[w0 + w];
            j.dst = (char*)dst[s] + (size_t)(square ? off * dst_ld[s] + off : off * dst_ld[s]) * sizeof(float);
            j.ld_src = win_ld[w0 + w];
            j.ld_dst = dst_ld[s];
            j.n_rows = win_rows[w0 + w];
            j.n_avg = n_avg;
            j.level = (int32_t)w;
            j.reserved = 0;
            jobs.push_back(j);
            if (simple) {
                StitchOnce o;
                o.prev = w > 0 ? (const float*)win[w0 + w - 1] + (size_t)(win_rows[w0 + w - 1] - n_avg) * (size_t)win_ld[w0 + w - 1] : nullptr;
                o.ld_prev = w > 0 ? win_ld[w0 + w - 1] : 0;
                o.n_skip = w + 1 < n ? step : 0;
                once.push_back(o);
            }
        });
        if (size < 0)
            return fail(DCTFP_ERR_SHAPE, square ? "sequence %lld: window offset beyond the running contact map"
                                                : "sequence %lld: a window is not longer than the overlap", (long long)s);
    }
    return stitch_impl(ctx, jobs.data(), (int64_t)jobs.size(), n_cols, square, stream_v, simple ? once.data() : nullptr);
} DCTFP_GUARD("dctfp_stitch_sequences")

int dctfp_l1_matrix(dctfp_ctx* ctx, const int8_t* a, int64_t na, int64_t lda, const int8_t* b, int64_t nb, int64_t ldb,
                    int32_t d, int32_t* out, int64_t ldo, void* stream_v) try {
    if (!ctx || !a || !b || !out) return fail(DCTFP_ERR_INVALID, "dctfp_l1_matrix: NULL argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (na < 0 || nb < 0 || d < 1 || lda < d || ldb < d || ldo < nb) return fail(DCTFP_ERR_INVALID, "dctfp_l1_matrix: bad shape");
    if (na == 0 || nb == 0) return DCTFP_OK;
    if ((na + 127) / 128 > 65535) return fail(DCTFP_ERR_LIMIT, "dctfp_l1_matrix: more than 8M rows per call");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t stream = (hipStream_t)stream_v;
    dim3 grid((unsigned)((nb + 127) / 128), (unsigned)((na + 127) / 128));  // 128 x 128 distances per workgroup (l1_matrix_kernel)
    const bool aligned = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | (uintptr_t)lda | (uintptr_t)ldb) & 3u) == 0;
    const bool aligned16 = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | (uintptr_t)lda | (uintptr_t)ldb) & 15u) == 0 &&
                           lda < (1 << 24) && ldb < (1 << 24);   // (the kernel addresses a tile's rows with 32-bit offsets)
    if (aligned16 && ctx->opt_l1_kernel != 1) hipLaunchKernelGGL(l1_matrix16_kernel, grid, dim3(256), 0, stream, a, na, lda, b, nb, ldb, d, out, ldo);
    else if (aligned) hipLaunchKernelGGL((l1_matrix_kernel<true>), grid, dim3(256), 0, stream, a, na, lda, b, nb, ldb, d, out, ldo);
    else hipLaunchKernelGGL((l1_matrix_kernel<false>), grid, dim3(256), 0, stream, a, na, lda, b, nb, ldb, d, out, ldo);
    HIP_TRY(hipGetLastError());
    return DCTFP_OK;
} DCTFP_GUARD("dctfp_l1_matrix")

int dctfp_block_min(dctfp_ctx* ctx, const int32_t* dist, int64_t ldo, const int64_t* idx_a, int64_t npa,
                    const int64_t* idx_b, int64_t npb, int32_t* out_min, int32_t* out_last, void* stream_v) try {
    if (!ctx || !dist || !idx_a || !idx_b || !out_min || !out_last) return fail(DCTFP_ERR_INVALID, "dctfp_block_min: NULL argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (npa < 0 || npb < 0) return fail(DCTFP_ERR_INVALID, "dctfp_block_min: negative count");
    if (npa == 0 || npb == 0) return DCTFP_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const int64_t total = npa * npb;
    hipLaunchKernelGGL(block_min_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_v, dist, ldo,
                       idx_a, npa, idx_b, npb, out_min, out_last);
    HIP_TRY(hipGetLastError());
    return DCTFP_OK;
} DCTFP_GUARD("dctfp_block_min")

int dctfp_row_select(dctfp_ctx* ctx, const int32_t* dist, int64_t n_rows, int64_t n_cols, int64_t ld, int32_t k,
                     int32_t* out_val, int32_t* out_idx, void* stream_v) try {
    if (!ctx || !dist || !out_val || !out_idx) return fail(DCTFP_ERR_INVALID, "dctfp_row_select: NULL argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (n_rows < 0 || n_cols < 1 || ld < n_cols || k < 1 || k > n_cols) return fail(DCTFP_ERR_INVALID, "dctfp_row_select: bad shape");
    if (n_rows == 0) return DCTFP_OK;
    if (n_rows > 0x7fffffff) return fail(DCTFP_ERR_LIMIT, "dctfp_row_select: too many rows");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t stream = (hipStream_t)stream_v;
    // The row (or a segment of it) in the registers of one workgroup, read once (row_select_reg_kernel); longer rows in segments
    // whose k candidates each a second launch selects from.  Large k, or more candidates than one workgroup holds: the radix
    // select, which reads its row six times but has no limit.
    constexpr int kPer = 40;
    constexpr int64_t kSeg = 1024 * kPer;
    const int64_t n_seg = (n_cols + kSeg - 1) / kSeg;
    const bool reg_ok = ctx->opt_row_select != 1 && k <= 1024 && n_rows * n_seg <= 0x7fffffff && (n_seg == 1 || n_seg * k <= kSeg);
    if (reg_ok && n_seg == 1) {
        hipLaunchKernelGGL((row_select_reg_kernel<2 * kPer, false, 512>), dim3((unsigned)n_rows), dim3(512), 0, stream, dist, (const int32_t*)nullptr,
                           ld, n_cols, kSeg, (int64_t)1, (int)k, out_val, out_idx);
    } else if (reg_ok) {
        const int64_t n_in = n_seg * k;                       // candidates per row
        int rc = ctx->scratch.ensure((size_t)n_rows * n_in * 2 * sizeof(int32_t));
        if (rc) return rc;
        int32_t* cand_val = (int32_t*)ctx->scratch.p;
        int32_t* cand_idx = cand_val + (size_t)n_rows * n_in;
        hipLaunchKernelGGL((row_select_reg_kernel<2 * kPer, false, 512>), dim3((unsigned)(n_rows * n_seg)), dim3(512), 0, stream, dist,
                           (const int32_t*)nullptr, ld, n_cols, kSeg, n_seg, (int)k, cand_val, cand_idx);
        hipLaunchKernelGGL((row_select_reg_kernel<kPer, true, 1024>), dim3((unsigned)n_rows), dim3(1024), 0, stream, (const int32_t*)cand_val,
                           (const int32_t*)cand_idx, n_in, n_in, n_in, (int64_t)1, (int)k, out_val, out_idx);
    } else {
        hipLaunchKernelGGL(row_select_kernel, dim3((unsigned)n_rows), dim3(1024), 0, stream, dist, ld, n_cols, k, out_val, out_idx);
    }
    HIP_TRY(hipGetLastError());
    return DCTFP_OK;
} DCTFP_GUARD("dctfp_row_select")

int dctfp_row_order(dctfp_ctx* ctx, int32_t* val, int32_t* idx, int64_t n_rows, int32_t k, void* stream_v) try {
    if (!ctx || !val || !idx) return fail(DCTFP_ERR_INVALID, "dctfp_row_order: NULL argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (n_rows < 0 || k < 1) return fail(DCTFP_ERR_INVALID, "dctfp_row_order: bad shape");
    if (k > 1024) return fail(DCTFP_ERR_LIMIT, "dctfp_row_order: more than 1024 entries per row (order them on the host)");
    if (n_rows > 0x7fffffff) return fail(DCTFP_ERR_LIMIT, "dctfp_row_order: too many rows");
    if (n_rows == 0 || k == 1) return DCTFP_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t stream = (hipStream_t)stream_v;
    if (k <= 128) hipLaunchKernelGGL((row_order_kernel<128>), dim3((unsigned)n_rows), dim3(64), 0, stream, val, idx, (int)k);
    else if (k <= 256) hipLaunchKernelGGL((row_order_kernel<256>), dim3((unsigned)n_rows), dim3(128), 0, stream, val, idx, (int)k);
    else hipLaunchKernelGGL((row_order_kernel<1024>), dim3((unsigned)n_rows), dim3(512), 0, stream, val, idx, (int)k);
    HIP_TRY(hipGetLastError());
    return DCTFP_OK;
} DCTFP_GUARD("dctfp_row_order")

int dctfp_host_device_pointer(void* host, void** dev) try {
    if (!host || !dev) return fail(DCTFP_ERR_INVALID, "dctfp_host_device_pointer: NULL argument");
    *dev = nullptr;
    hipError_t e = hipHostGetDevicePointer(dev, host, 0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *dev = nullptr;
        return fail(DCTFP_ERR_HIP, "hipHostGetDevicePointer: %s", hipGetErrorString(e));
    }
    return DCTFP_OK;
} DCTFP_GUARD("dctfp_host_device_pointer")

int dctfp_stream_synchronize(void* stream) try {
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return DCTFP_OK;
} DCTFP_GUARD("dctfp_stream_synchronize")

int64_t dctfp_contact_count(int32_t n_res, double t) {
    if (n_res < 6) return 0;
    const int64_t cand = (int64_t)(n_res - 5) * (n_res - 4) / 2;  // pairs with j >= i + 5
    int64_t tot = (int64_t)(t * (double)n_res);                   // int(t * slen), src/fingerprint.py:65
    if (tot < 0) tot = 0;
    return tot > cand ? cand : tot;
}


// ---- host-side table builders (no GPU work): what the Python front end used to do per domain / per window in loops ----

/* Fingerprint.get_doms' domain-string rules (src/fingerprint.py:163-169) for a whole batch -> dctfp_piece[].
 * See include/dctfp.h.  Only strings of the plain form  digits-digits[,digits-digits]*  are handled here (what RecCut
 * prints and every reference test uses); anything else -- signs, blanks, underscores, empty fields: whatever Python's
 * int() and str.split would accept or reject -- is counted in *n_other and left to the caller's own parser. */
int dctfp_build_pieces(const char* text, int64_t text_len, const int32_t* str_count, const int64_t* seq_rows, int32_t n_seq,
                       dctfp_piece* pieces, int64_t piece_cap, int64_t* n_pieces, int32_t* str_row, int64_t* str_len,
                       uint8_t* str_changed, char* key_text, int64_t key_cap, int64_t* key_len, int64_t* n_domains,
                       int64_t* n_other) try {
    if (!text || !str_count || !seq_rows || !pieces || !n_pieces || !str_row || !str_len || !str_changed || !key_len || !n_domains || !n_other)
        return fail(DCTFP_ERR_INVALID, "dctfp_build_pieces: NULL argument");
    if (n_seq < 0 || text_len < 0 || piece_cap < 0 || key_cap < 0) return fail(DCTFP_ERR_INVALID, "dctfp_build_pieces: negative size");
    struct Part { const char* p; int32_t len; int64_t beg, end; };
    std::vector<Part> parts;
    int64_t np = 0, nd = 0, kl = 0, other = 0, istr = 0;
    const char* cur = text;
    const char* const fin = text + text_len;
    for (int32_t s = 0; s < n_seq; ++s) {
        const int64_t L = seq_rows[s];
        for (int32_t k = 0; k < str_count[s]; ++k, ++istr) {
            if (cur > fin) return fail(DCTFP_ERR_INVALID, "dctfp_build_pieces: fewer strings in the text than str_count says");
            const char* e = cur;
            while (e < fin && *e != '\n') ++e;
            // ---- split at ',' and parse "digits-digits"
            parts.clear();
            bool plain = e > cur;
            for (const char* q = cur; plain && q <= e;) {
                const char* f = q;
                while (f < e && *f != ',') ++f;
                Part pt{q, (int32_t)(f - q), 0, 0};
                const char* c = q;
                int digits = 0;
                while (c < f && *c >= '0' && *c <= '9' && digits < 18) { pt.beg = pt.beg * 10 + (*c - '0'); ++c; ++digits; }
                if (digits == 0 || c >= f || *c != '-') { plain = false; break; }
                ++c;
                digits = 0;
                while (c < f && *c >= '0' && *c <= '9' && digits < 18) { pt.end = pt.end * 10 + (*c - '0'); ++c; ++digits; }
                if (digits == 0 || c != f) { plain = false; break; }
                parts.push_back(pt);
                q = f + 1;
                if (f == e) break;
            }
            str_row[istr] = -1;
            str_len[istr] = 0;
            str_changed[istr] = 0;
            if (!plain) {
                str_changed[istr] = 2;
                ++other;
                cur = e + 1;
                continue;
            }
            // ---- the reference's loop: `for r in regions: if (beg or end) > L: regions.remove(r)` -- the list shrinks
            // under its own iterator, so the part after a removed one is never looked at but stays in the key; remove()
            // takes out the FIRST equal string
            const int64_t first_piece = np;
            int64_t rows = 0;
            bool changed = false;
            for (size_t i = 0; i < parts.size();) {
                const Part pt = parts[i];
                if ((pt.beg ? pt.beg : pt.end) > L) {
                    size_t victim = i;
                    for (size_t v = 0; v < i; ++v)
                        if (parts[v].len == pt.len && memcmp(parts[v].p, pt.p, (size_t)pt.len) == 0) { victim = v; break; }
                    parts.erase(parts.begin() + (ptrdiff_t)victim);
                    changed = true;
                    ++i;  // (the iterator advances regardless)
                    continue;
                }
                // embed[beg - 1 : end] with Python's slice rules
                int64_t start = pt.beg - 1, stop = pt.end;
                if (start < 0) { start += L; if (start < 0) start = 0; }
                if (start > L) start = L;
                if (stop > L) stop = L;
                if (stop > start) {
                    if (np >= piece_cap) return fail(DCTFP_ERR_INVALID, "dctfp_build_pieces: piece_cap %lld too small", (long long)piece_cap);
                    if (stop - start > 0x7fffffff) return fail(DCTFP_ERR_LIMIT, "dctfp_build_pieces: piece longer than 2^31 rows");
                    pieces[np].row_start = start;
                    pieces[np].n_rows = (int32_t)(stop - start);
                    pieces[np].domain = (int32_t)nd;
                    pieces[np].seq = s;
                    pieces[np].reserved = 0;
                    ++np;
                    rows += stop - start;
                }
                ++i;
            }
            if (changed) {  // the cleaned key: the surviving parts joined by ','
                str_changed[istr] = 1;
                for (size_t i = 0; i < parts.size(); ++i) {
                    if (kl + parts[i].len + 1 > key_cap) return fail(DCTFP_ERR_INVALID, "dctfp_build_pieces: key_cap too small");
                    if (i) key_text[kl++] = ',';
                    memcpy(key_text + kl, parts[i].p, (size_t)parts[i].len);
                    kl += parts[i].len;
                }
                if (kl + 1 > key_cap) return fail(DCTFP_ERR_INVALID, "dctfp_build_pieces: key_cap too small");
                key_text[kl++] = '\n';
            }
            if (np > first_piece) {
                if (nd >= 0x7fffffff) return fail(DCTFP_ERR_LIMIT, "dctfp_build_pieces: more than 2^31 domains");
                str_row[istr] = (int32_t)nd;
                str_len[istr] = rows;
                ++nd;
            }
            cur = e + 1;
        }
    }
    if (istr > 0 && cur != fin + 1) ++other;  // more line breaks in the text than strings: one of them had a '\n' inside
    *n_pieces = np;
    *n_domains = nd;
    *key_len = kl;
    *n_other = other;
    return DCTFP_OK;
} DCTFP_GUARD("dctfp_build_pieces")

/* The domain lists of a flush straight from dctfp_reccut's encoded results: the strings + the piece table in one pass, no text
 * round trip (format -> split -> join -> parse).  See include/dctfp.h. */
int dctfp_reccut_pieces(int32_t n_prot, const int32_t* enc, const int64_t* enc_off, const int64_t* seq_rows, char* text,
                        int64_t text_cap, int64_t* text_len, int32_t* str_count, dctfp_piece* pieces, int64_t piece_cap,
                        int64_t* n_pieces, int64_t* n_domains, int64_t* n_undone) try {
    if (!enc || !enc_off || !seq_rows || !text || !text_len || !str_count || !pieces || !n_pieces || !n_domains || !n_undone)
        return fail(DCTFP_ERR_INVALID, "dctfp_reccut_pieces: NULL argument");
    if (n_prot < 0 || text_cap < 0 || piece_cap < 0) return fail(DCTFP_ERR_INVALID, "dctfp_reccut_pieces: negative size");
    int64_t at = 0, np = 0, nd = 0, undone = 0;
    auto put_int = [&](int64_t v) {   // decimal digits of v >= 0 (room checked by the caller: 20 bytes)
        char tmp[24];
        int k = 0;
        do {
            tmp[k++] = (char)('0' + v % 10);
            v /= 10;
        } while (v);
        while (k) text[at++] = tmp[--k];
    };
    for (int32_t p = 0; p < n_prot; ++p) {
        const int32_t* e = enc + enc_off[p];
        const int64_t room = enc_off[p + 1] - enc_off[p];
        const int64_t L = seq_rows[p];
        str_count[p] = 0;
        // ---- first pass: is the record whole, every segment inside the protein?  (status -1, or anything get_doms' clean-up
        // rules would have to judge, is left to the caller: it has the host library and the string parser for those)
        bool ok = room >= 1 && e[0] >= 1 && L >= 1;
        int64_t q = 1, n_seg_all = 0;
        for (int32_t d = 0; ok && d < e[0]; ++d) {
            if (q >= room) { ok = false; break; }
            const int32_t ns = e[q++];
            if (ns < 1 || q + 2 * (int64_t)ns > room) { ok = false; break; }
            for (int32_t sg = 0; sg < ns; ++sg, q += 2)
                if (e[q] < 0 || e[q + 1] < e[q] || e[q + 1] >= L) ok = false;
            n_seg_all += ns;
        }
        if (!ok) {
            ++undone;
            continue;
        }
        const bool several = e[0] > 1;
        if (np + n_seg_all + (several ? 1 : 0) > piece_cap) return fail(DCTFP_ERR_INVALID, "dctfp_reccut_pieces: piece_cap %lld too small", (long long)piece_cap);
        if (at + 24 * n_seg_all + 32 > text_cap) return fail(DCTFP_ERR_INVALID, "dctfp_reccut_pieces: text_cap %lld too small", (long long)text_cap);
        if (nd + e[0] + 1 > 0x7fffffff) return fail(DCTFP_ERR_LIMIT, "dctfp_reccut_pieces: more than 2^31 domains");
        q = 1;
        for (int32_t d = 0; d < e[0]; ++d) {
            const int32_t ns = e[q++];
            for (int32_t sg = 0; sg < ns; ++sg, q += 2) {
                if (sg) text[at++] = ',';
                put_int((int64_t)e[q] + 1);      // 1-based, inclusive: what the binary prints (src/RecCut.cpp:440-444)
                text[at++] = '-';
                put_int((int64_t)e[q + 1] + 1);
                pieces[np].row_start = e[q];     // embed[beg - 1 : end] (src/fingerprint.py:168)
                pieces[np].n_rows = e[q + 1] - e[q] + 1;
                pieces[np].domain = (int32_t)nd;
                pieces[np].seq = p;
                pieces[np].reserved = 0;
                ++np;
            }
            text[at++] = ';';
            ++nd;
        }
        if (several) {                           // `if len(domains) > 1: self.domains.append(f'1-{len(self.seq)}')` (src/fingerprint.py:106-107)
            text[at++] = '1';
            text[at++] = '-';
            put_int(L);
            text[at++] = ';';
            pieces[np].row_start = 0;
            pieces[np].n_rows = (int32_t)std::min<int64_t>(L, 0x7fffffff);
            pieces[np].domain = (int32_t)nd;
            pieces[np].seq = p;
            pieces[np].reserved = 0;
            ++np;
            ++nd;
        }
        str_count[p] = e[0] + (several ? 1 : 0);
    }
    *text_len = at;
    *n_pieces = np;
    *n_domains = nd;
    *n_undone = undone;
    return DCTFP_OK;
} DCTFP_GUARD("dctfp_reccut_pieces")

// ---- diagnostics: which HIP / HSA runtime this library is bound to, and a crash handler that names the failing frame ----

/* Writes one line per fact into buf: the HIP version this library was compiled against, the version of the runtime it is
 * bound to in this process (hipRuntimeGetVersion / hipDriverGetVersion), the file that runtime was loaded from (dladdr of
 * hipMalloc) and every libamdhip64 / libhsa-runtime64 / libamd_comgr mapped into the process (/proc/self/maps).
 * libdctfp.so is compiled by /opt/rocm's hipcc but must run on the runtime of the process that owns the device pointers
 * and streams it is given (torch's bundled one): exactly one of each may be mapped.  Returns the number of distinct
 * libamdhip64 files mapped (1 = healthy) or a negative error. */
int dctfp_runtime_info(char* buf, int64_t cap) try {
    if (!buf || cap < 64) return fail(DCTFP_ERR_INVALID, "dctfp_runtime_info: buffer of at least 64 bytes");
    size_t used = 0;
    auto put = [&](const char* fmt, ...) {
        if (used + 1 >= (size_t)cap) return;
        va_list ap;
        va_start(ap, fmt);
        const int n = vsnprintf(buf + used, (size_t)cap - used, fmt, ap);
        va_end(ap);
        if (n > 0) used = std::min((size_t)cap - 1, used + (size_t)n);
    };
    int rt = 0, drv = 0;
    const hipError_t e1 = hipRuntimeGetVersion(&rt), e2 = hipDriverGetVersion(&drv);
    put("libdctfp compiled against HIP %d.%d.%d; bound runtime reports hipRuntimeGetVersion=%d (%s) hipDriverGetVersion=%d (%s)\n",
        HIP_VERSION_MAJOR, HIP_VERSION_MINOR, HIP_VERSION_PATCH, rt, hipGetErrorName(e1), drv, hipGetErrorName(e2));
    Dl_info info;
    if (dladdr((void*)&hipRuntimeGetVersion, &info) && info.dli_fname) put("hipRuntimeGetVersion resolves into %s\n", info.dli_fname);
    if (dladdr((void*)&dctfp_runtime_info, &info) && info.dli_fname) put("this library: %s\n", info.dli_fname);
    std::vector<std::string> seen;
    int n_hip = 0;
    if (FILE* f = fopen("/proc/self/maps", "r")) {
        char line[1024];
        while (fgets(line, sizeof line, f)) {
            const char* path = strchr(line, '/');
            if (!path) continue;
            if (!strstr(path, "libamdhip64") && !strstr(path, "libhsa-runtime64") && !strstr(path, "libamd_comgr")) continue;
            std::string ps(path);
            while (!ps.empty() && (ps.back() == '\n' || ps.back() == ' ')) ps.pop_back();
            if (std::find(seen.begin(), seen.end(), ps) != seen.end()) continue;
            seen.push_back(ps);
            if (ps.find("libamdhip64") != std::string::npos) ++n_hip;
            put("mapped: %s\n", ps.c_str());
        }
        fclose(f);
    }
    return n_hip;
} DCTFP_GUARD("dctfp_runtime_info")

namespace {
struct sigaction g_prev_abrt, g_prev_segv, g_prev_bus;
bool g_crash_installed = false;

// SIGABRT / SIGSEGV / SIGBUS: the native frames to stderr (backtrace_symbols_fd does not allocate), then the handler that was
// installed before (Python's faulthandler prints the Python stack and re-raises), so an abort names itself in the log of the
// ordinary run instead of needing a second one under a debugger.
void crash_handler(int sig, siginfo_t* si, void* uc) {
    static const char head[] = "\n*** libdctfp: fatal signal; native backtrace of the failing thread:\n";
    (void)!write(2, head, sizeof head - 1);
    void* frames[64];
    const int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    static const char tail[] = "*** end of native backtrace\n";
    (void)!write(2, tail, sizeof tail - 1);
    struct sigaction* prev = sig == SIGABRT ? &g_prev_abrt : (sig == SIGSEGV ? &g_prev_segv : &g_prev_bus);
    sigaction(sig, prev, nullptr);
    if ((prev->sa_flags & SA_SIGINFO) && prev->sa_sigaction) {
        prev->sa_sigaction(sig, si, uc);
    } else if (prev->sa_handler != SIG_DFL && prev->sa_handler != SIG_IGN && prev->sa_handler) {
        prev->sa_handler(sig);
    }
    raise(sig);  // (back under the previous disposition: ends the process the way it would have ended without us)
}
}  // namespace

/* enable = 1: install the handler above (idempotent); 0: put the previous handlers back.  Installed at load time when the
 * environment has DCTFP_CRASH_BACKTRACE=1 (tests/conftest.py, bench.py, __graft_entry__.smoke and the tools/ wrappers set it),
 * and always by libdctfp_experiments.so. */
int dctfp_crash_handler(int enable) {
    if (enable && !g_crash_installed) {
        // (libdctfp.so and libdctfp_experiments.so may both be loaded: one handler per process)
        if (getenv("DCTFP_CRASH_HANDLER_INSTALLED")) return DCTFP_OK;
        setenv("DCTFP_CRASH_HANDLER_INSTALLED", "1", 1);
        void* warm[4];
        (void)backtrace(warm, 4);  // loads libgcc's unwinder now, not inside the handler
        struct sigaction sa;
        memset(&sa, 0, sizeof sa);
        sa.sa_sigaction = crash_handler;
        sa.sa_flags = SA_SIGINFO | SA_NODEFER;
        sigemptyset(&sa.sa_mask);
        sigaction(SIGABRT, &sa, &g_prev_abrt);
        sigaction(SIGSEGV, &sa, &g_prev_segv);
        sigaction(SIGBUS, &sa, &g_prev_bus);
        g_crash_installed = true;
    } else if (!enable && g_crash_installed) {
        sigaction(SIGABRT, &g_prev_abrt, nullptr);
        sigaction(SIGSEGV, &g_prev_segv, nullptr);
        sigaction(SIGBUS, &g_prev_bus, nullptr);
        unsetenv("DCTFP_CRASH_HANDLER_INSTALLED");
        g_crash_installed = false;
    }
    return DCTFP_OK;
}

}  // extern "C"

namespace {
__attribute__((constructor)) void dctfp_on_load() {
#ifdef DCTFP_EXPERIMENTS
    const bool on = true;
#else
    const char* v = getenv("DCTFP_CRASH_BACKTRACE");
    const bool on = v && v[0] == '1';
#endif
    if (on) (void)dctfp_crash_handler(1);
}
}  // namespace
