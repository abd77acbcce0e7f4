// Host-only stand-in for the HIP runtime, for AddressSanitizer runs of libdctfp's host code on a machine WITHOUT a GPU
// (tests/test_host_asan.py; GPU ASan is not available on the pool).  "Device" memory is host memory, so every table the
// host builds, every copy it sizes and every pointer it hands to a kernel is checked by ASan; kernels are not executed --
// instead `hipLaunchKernel` walks the job tables of the main kernels the way their waves do and touches every address they
// would (first / last byte of every row range, cosine table, output block): an index that is wrong on the host side
// shows up here as a heap-buffer-overflow instead of as a GPU fault on the box.
// Test infrastructure only: never linked into the product.
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>

// The job-table records of dctdomain_amd/csrc/kernels.hip.h, restated (that header is device code: a second translation
// unit including it would define its kernels twice).  tests/test_host_asan.py checks that the two stay in step.
struct JobA {
    uint32_t piece_begin, n_pieces, n_rows, reserved;
    const double* basis;
    const double* w_basis;
    const void* w_ref;
};
struct PieceA {
    const void* ptr;
    uint32_t n_rows, t0, w0, reserved;
    const void* ptr2;
};
struct JobB {
    int64_t out_off;
};
struct Walk {
    uint32_t job_begin, n_parts;
    int32_t whole_job;
    uint32_t reserved;
};
struct Run {
    uint32_t walk_begin, n_walks, job_begin, n_jobs;
};
struct BasisJob {
    double* tab;
    uint32_t len, reserved;
};
static_assert(sizeof(JobA) == 40 && sizeof(PieceA) == 32 && sizeof(JobB) == 8 && sizeof(Walk) == 16 && sizeof(Run) == 16 && sizeof(BasisJob) == 16,
              "job-table records as in kernels.hip.h");

namespace {
std::map<const void*, std::string>& names() {
    static std::map<const void*, std::string> m;
    return m;
}
volatile unsigned char g_sink;
inline void touch(const void* p, size_t bytes) {  // ASan checks both ends of [p, p + bytes)
    if (!bytes) return;
    const volatile unsigned char* c = (const volatile unsigned char*)p;
    g_sink ^= c[0];
    g_sink ^= c[bytes - 1];
}
inline void touch_w(void* p, size_t bytes) {
    if (!bytes) return;
    volatile unsigned char* c = (volatile unsigned char*)p;
    c[0] = c[0];
    c[bytes - 1] = c[bytes - 1];
}
size_t elem_size(const std::string& n, const char* kernel) {
    const size_t at = n.find(kernel);
    const std::string t = n.substr(at + strlen(kernel), 12);
    if (t.rfind("If", 0) == 0) return 4;
    if (t.rfind("Id", 0) == 0) return 8;
    return 2;  // _Float16 / bf16_t
}
int template_int(const std::string& n, const char* kernel, int index) {  // the index-th "Li<k>E" after the kernel name
    size_t at = n.find(kernel);
    for (int i = 0; i <= index; ++i) {
        at = n.find("Li", at + 1);
        if (at == std::string::npos) return -1;
    }
    return atoi(n.c_str() + at + 2);
}
unsigned long g_walk_launches = 0, g_stage_a_launches = 0, g_jobs_walked = 0;

void emulate_walk(const std::string& name, dim3 grid, void** a) {
    const JobA* jobs = *(const JobA**)a[0];
    const JobB* jobb = *(const JobB**)a[1];
    const Walk* walks = *(const Walk**)a[2];
    const Run* runs = *(const Run**)a[3];
    const PieceA* pieces = *(const PieceA**)a[4];
    const double* stf = *(const double**)a[5];
    int8_t* out = *(int8_t**)a[6];
    const int n_cols = *(int*)a[7];
    const int64_t ld = *(int64_t*)a[8];
    const int m = *(int*)a[9];
    const size_t esz = elem_size(name, "walk_ab_kernel");
    const int S = template_int(name, "walk_ab_kernel", 0), G = template_int(name, "walk_ab_kernel", 1);
    if (S * 256 < n_cols) { fprintf(stderr, "stub: %d waves cannot cover %d channels\n", S, n_cols); abort(); }
    const int groups = (n_cols / 2 + 15) / 16;
    const int nt_w = template_int(name, "walk_ab_kernel", 2);   // five or six column groups ([E | O] halves of 40 / 48 slots)
    if ((nt_w != 5 && nt_w != 6) || m > nt_w * 16 || (nt_w == 6 && m <= 80)) { fprintf(stderr, "stub: walk_ab_kernel build of %d column groups for m = %d\n", nt_w, m); abort(); }
    touch(stf, (size_t)(groups * 4 + 4) * nt_w * 64 * sizeof(double));  // fragments up to 4 k-steps past the last group
    ++g_walk_launches;
    for (unsigned b = 0; b < grid.x; ++b) {
        const Run run = runs[b];
        uint32_t pending = 0, group_job = run.job_begin, jobs_seen = 0;
        for (uint32_t wi = 0; wi < run.n_walks; ++wi) {
            const Walk wk = walks[run.walk_begin + wi];
            const bool has_w = wk.whole_job >= 0;
            const uint32_t w_rows = has_w ? jobs[wk.whole_job].n_rows : 0;
            const uint32_t n_walk_jobs = wk.n_parts + (has_w ? 1u : 0u);
            for (uint32_t part = 0; part < n_walk_jobs; ++part) {
                const uint32_t job_id = part < wk.n_parts ? wk.job_begin + part : (uint32_t)wk.whole_job;
                if (job_id != group_job + pending) {  // slot g of a flush <-> job group_job + g (walk_ab_kernel, epilogue)
                    fprintf(stderr, "stub: run %u walk %u part %u is job %u, the flush expects %u\n", b, wi, part, job_id, group_job + pending);
                    abort();
                }
                if (part < wk.n_parts) {
                    const JobA job = jobs[job_id];
                    uint32_t rows = 0;
                    for (uint32_t p = 0; p < job.n_pieces; ++p) {
                        const PieceA pc = pieces[job.piece_begin + p];
                        if (pc.n_rows == 0 || pc.t0 != rows) { fprintf(stderr, "stub: piece table of job %u broken\n", job_id); abort(); }
                        touch(pc.ptr, (size_t)n_cols * esz);
                        touch((const char*)pc.ptr + (size_t)(pc.n_rows - 1) * (size_t)ld * esz, (size_t)n_cols * esz);
                        if (pc.ptr2) {  // the mean of two windows' rows: only the builds with the last template flag set read it
                            if (name.find("Lb1EEEvPK") == std::string::npos || esz != 4) { fprintf(stderr, "stub: two-source piece sent to %s\n", name.c_str()); abort(); }
                            touch(pc.ptr2, (size_t)n_cols * esz);
                            touch((const char*)pc.ptr2 + (size_t)(pc.n_rows - 1) * (size_t)ld * esz, (size_t)n_cols * esz);
                        }
                        touch(job.basis + (size_t)pc.t0 * 2, (size_t)pc.n_rows * 2 * sizeof(double));
                        // the compiler merges the cosines of up to four rows into one scalar load: 8 doubles from any row
                        touch(job.basis + (size_t)(pc.t0 + pc.n_rows - 1) * 2, 8 * sizeof(double));
                        if (has_w) {
                            touch(job.w_basis + (size_t)pc.w0 * 2, (size_t)pc.n_rows * 2 * sizeof(double));
                            touch(job.w_basis + (size_t)(pc.w0 + pc.n_rows - 1) * 2, 8 * sizeof(double));
                            touch(job.w_basis + ((size_t)w_rows + pc.w0) * 2, ((size_t)pc.n_rows + 1) * 2 * sizeof(double));  // prefix sums
                        }
                        rows += pc.n_rows;
                    }
                    if (rows != job.n_rows || rows < 3) { fprintf(stderr, "stub: job %u has %u rows, its pieces %u\n", job_id, job.n_rows, rows); abort(); }
                    if (has_w) touch(job.w_ref, (size_t)n_cols * esz);
                }
                ++pending;
                ++jobs_seen;
                ++g_jobs_walked;
                const bool last = wi + 1 == run.n_walks && part + 1 == n_walk_jobs;
                if (pending < (uint32_t)G && !last) continue;
                for (uint32_t g = 0; g < pending; ++g) touch_w(out + jobb[group_job + g].out_off, (size_t)3 * m);
                group_job += pending;
                pending = 0;
            }
        }
        if (jobs_seen != run.n_jobs) { fprintf(stderr, "stub: run %u walks %u jobs, says %u\n", b, jobs_seen, run.n_jobs); abort(); }
    }
}

void emulate_stage_a(const std::string& name, dim3 grid, void** a) {
    const JobA* jobs = *(const JobA**)a[0];
    const Walk* walks = *(const Walk**)a[1];
    const PieceA* pieces = *(const PieceA**)a[2];
    char* yprime = *(char**)a[3];
    const int64_t job_bytes = *(int64_t*)a[4];
    const int n_cols = *(int*)a[6];
    const int64_t ld = *(int64_t*)a[7];
    const int n_slabs = *(int*)a[9];
    const size_t esz = elem_size(name, "stage_a_kernel");
    const int nk = template_int(name, "stage_a_kernel", 0) - 1;
    ++g_stage_a_launches;
    for (unsigned w = 0; w < grid.x / (unsigned)n_slabs; ++w) {
        const Walk wk = walks[w];
        const bool has_w = wk.whole_job >= 0;
        for (uint32_t part = 0; part < wk.n_parts; ++part) {
            const JobA job = jobs[wk.job_begin + part];
            for (uint32_t p = 0; p < job.n_pieces; ++p) {
                const PieceA pc = pieces[job.piece_begin + p];
                if (pc.ptr2) { fprintf(stderr, "stub: two-source piece sent to %s\n", name.c_str()); abort(); }
                touch(pc.ptr, (size_t)n_cols * esz);
                touch((const char*)pc.ptr + (size_t)(pc.n_rows - 1) * (size_t)ld * esz, (size_t)n_cols * esz);
                touch(job.basis + (size_t)pc.t0 * nk, (size_t)pc.n_rows * nk * sizeof(double));
                if (has_w) touch(job.w_basis + (size_t)pc.w0 * nk, (size_t)pc.n_rows * nk * sizeof(double));
            }
            touch_w(yprime + (size_t)(wk.job_begin + part) * job_bytes, (size_t)job_bytes);
            ++g_jobs_walked;
        }
        if (has_w) {
            touch(jobs[wk.job_begin].w_ref, (size_t)n_cols * esz);
            touch_w(yprime + (size_t)wk.whole_job * job_bytes, (size_t)job_bytes);
        }
    }
}

// walk_gen_kernel (the general walk kernel): every job of every run streams its pieces, reads its cosine table, the plain
// fragment table up to the last k-step that holds a channel, and writes its n x m block.
void emulate_walk_gen(const std::string& name, dim3 grid, dim3 block, size_t shmem, void** a) {
    const JobA* jobs = *(const JobA**)a[0];
    const JobB* jobb = *(const JobB**)a[1];
    const Walk* walks = *(const Walk**)a[2];
    const Run* runs = *(const Run**)a[3];
    const PieceA* pieces = *(const PieceA**)a[4];
    const double* stp = *(const double**)a[5];
    int8_t* out = *(int8_t**)a[6];
    const int n_cols = *(int*)a[7];
    const int64_t ld = *(int64_t*)a[8];
    const int m = *(int*)a[9];
    const int n_slots = *(int*)a[10];
    const size_t esz = elem_size(name, "walk_gen_kernel");
    // template <typename T, int N, int VEC, bool FUSED, int NTC>: ... Li<N>E Li<VEC>E Lb<FUSED>E Li<NTC>E
    const int n = template_int(name, "walk_gen_kernel", 0), vec = template_int(name, "walk_gen_kernel", 1), ntc = template_int(name, "walk_gen_kernel", 2);
    const bool fused = name.find("Lb1ELi") != std::string::npos;
    const int waves = (int)(block.x / 64), nt = (m + 15) / 16;
    if (block.x % 64 || waves * 64 * vec < n_cols || n_cols % vec || n_slots < 1 || n_slots > 2 || nt > ntc || (fused && (waves > 10 || n > 5)) ||
        shmem < (size_t)n_slots * ((size_t)n * waves * 64 * vec + (nt * 16 <= 64 * vec ? 0 : (size_t)waves * n * nt * 16)) * 8 + 16 || shmem > 160 * 1024) {
        fprintf(stderr, "stub: walk_gen_kernel launch shape: %d waves x %d channels per lane for D = %d, m = %d (build: %d column groups, fused %d, n %d), %d slots, %zu bytes of LDS\n",
                waves, vec, n_cols, m, ntc, (int)fused, n, n_slots, shmem);
        abort();
    }
    touch(stp, (size_t)((n_cols + 3) / 4) * nt * 64 * sizeof(double));
    ++g_walk_launches;
    const int nk = n - 1;
    for (unsigned b = 0; b < grid.x; ++b) {
        const Run run = runs[b];
        uint32_t jn = 0;
        const uint32_t n_walks = fused ? run.n_walks : run.n_jobs;
        for (uint32_t wi = 0; wi < n_walks; ++wi) {
            Walk wk{run.job_begin + wi, 1, -1, 0};
            if (fused) wk = walks[run.walk_begin + wi];
            const bool has_w = fused && wk.whole_job >= 0;
            const uint32_t w_rows = has_w ? jobs[wk.whole_job].n_rows : 0;
            const uint32_t n_walk_jobs = wk.n_parts + (has_w ? 1u : 0u);
            for (uint32_t part = 0; part < n_walk_jobs; ++part, ++jn) {
                const uint32_t job_id = run.job_begin + jn;      // (the kernel takes the jobs of a run in order)
                if (fused && job_id != (part < wk.n_parts ? wk.job_begin + part : (uint32_t)wk.whole_job)) {
                    fprintf(stderr, "stub: run %u walk %u part %u is not job %u\n", b, wi, part, job_id);
                    abort();
                }
                if (part < wk.n_parts) {
                    const JobA job = jobs[job_id];
                    uint32_t rows = 0;
                    for (uint32_t p = 0; p < job.n_pieces; ++p) {
                        const PieceA pc = pieces[job.piece_begin + p];
                        if (pc.n_rows == 0 || pc.t0 != rows) { fprintf(stderr, "stub: piece table of job %u broken\n", job_id); abort(); }
                        if (pc.ptr2) { fprintf(stderr, "stub: two-source piece sent to %s\n", name.c_str()); abort(); }
                        touch(pc.ptr, (size_t)n_cols * esz);
                        touch((const char*)pc.ptr + (size_t)(pc.n_rows - 1) * (size_t)ld * esz, (size_t)n_cols * esz);
                        touch(job.basis + (size_t)pc.t0 * nk, (size_t)pc.n_rows * nk * sizeof(double));
                        if (has_w) {
                            touch(job.w_basis + (size_t)pc.w0 * nk, (size_t)pc.n_rows * nk * sizeof(double));
                            touch(job.w_basis + ((size_t)w_rows + pc.w0) * nk, ((size_t)pc.n_rows + 1) * nk * sizeof(double));  // prefix sums
                        }
                        rows += pc.n_rows;
                    }
                    if (rows != job.n_rows || (int)rows < n) { fprintf(stderr, "stub: job %u has %u rows, its pieces %u\n", job_id, job.n_rows, rows); abort(); }
                    if (has_w) touch(job.w_ref, (size_t)n_cols * esz);
                }
                touch_w(out + jobb[job_id].out_off, (size_t)n * m);
                ++g_jobs_walked;
            }
        }
        if (jn != run.n_jobs) { fprintf(stderr, "stub: run %u of the general kernel walks %u jobs, says %u\n", b, jn, run.n_jobs); abort(); }
    }
}

void emulate_basis(dim3 grid, void** a) {
    const BasisJob* tabs = *(const BasisJob**)a[0];
    const int nk = *(int*)a[1];
    for (unsigned y = 0; y < grid.y; ++y) touch_w(tabs[y].tab, ((size_t)tabs[y].len * nk + ((size_t)tabs[y].len + 1) * nk) * sizeof(double));
}

void emulate_stage_b(void** a) {
    const int64_t rows = *(int64_t*)a[2];
    const JobB* jobs = *(const JobB**)a[5];
    const int n = *(int*)a[6], m = *(int*)a[7];
    int8_t* out = *(int8_t**)a[8];
    const char* ypb = *(const char**)a[0];
    const int64_t job_bytes = *(int64_t*)a[1];
    for (int64_t j = 0; j < rows / n; ++j) {
        touch(ypb + (size_t)j * job_bytes, (size_t)job_bytes);
        touch_w(out + jobs[j].out_off, (size_t)n * m);
    }
}
}  // namespace

extern "C" {
unsigned long dctfp_stub_counter(int which) { return which == 0 ? g_walk_launches : (which == 1 ? g_stage_a_launches : g_jobs_walked); }

hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_t* p, int) {
    memset(p, 0, sizeof *p);
    strcpy(p->gcnArchName, "gfx950:sramecc+:xnack-");
    p->multiProcessorCount = 256;
    return hipSuccess;
}
hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 1; *hi = -1; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 1.0f; return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "stub"; }
const char* hipGetErrorName(hipError_t) { return "stub"; }
hipError_t hipRuntimeGetVersion(int* v) { *v = 0; return hipSuccess; }
hipError_t hipDriverGetVersion(int* v) { *v = 0; return hipSuccess; }

void** __hipRegisterFatBinary(const void*) { static void* h; return &h; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void* host_fn, char*, const char* device_name, unsigned, void*, void*, void*, void*, int*) {
    names()[host_fn] = device_name;
}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
namespace { dim3 g_grid, g_block; size_t g_shmem; hipStream_t g_stream; }
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream) {
    g_grid = grid; g_block = block; g_shmem = shmem; g_stream = stream;
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3* grid, dim3* block, size_t* shmem, hipStream_t* stream) {
    *grid = g_grid; *block = g_block; *shmem = g_shmem; *stream = g_stream;
    return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipLaunchKernel(const void* fn, dim3 grid, dim3 block, void** args, size_t shmem, hipStream_t) {
    const auto it = names().find(fn);
    if (it == names().end()) { fprintf(stderr, "stub: launch of an unregistered kernel\n"); abort(); }
    const std::string& n = it->second;
    if (grid.x == 0 || grid.y == 0 || block.x == 0 || block.x > 1024) { fprintf(stderr, "stub: bad launch shape of %s\n", n.c_str()); abort(); }
    if (n.find("walk_gen_kernel") != std::string::npos) emulate_walk_gen(n, grid, block, shmem, args);
    else if (n.find("walk_ab_kernel") != std::string::npos) emulate_walk(n, grid, args);
    else if (n.find("stage_a_kernel") != std::string::npos) emulate_stage_a(n, grid, args);
    else if (n.find("basis_kernel") != std::string::npos) emulate_basis(grid, args);
    else if (n.find("stage_b_mfma_kernel") != std::string::npos) emulate_stage_b(args);
    return hipSuccess;
}
}
