// Context, error text, device-buffer helpers and the contraction test hooks of the C-ABI.
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <mutex>

#include <map>

#include "crm_internal.h"
#include "eigh.h"

namespace crm {

static thread_local std::string g_error;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
}

const char* last_error_text() { return g_error.c_str(); }

namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_LAZY | RTLD_GLOBAL);
        if (!h) h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_LAZY | RTLD_GLOBAL);
        if (!h) return;
        push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
};
const Roctx& roctx() {
    static Roctx r;   // (only looked up when ranges are asked for: CRM_ROCTX=1 or a rocprofiler tool in the process)
    return r;
}
bool ranges_on() {
    static const bool on = getenv("CRM_ROCTX") || getenv("ROCP_TOOL_LIBRARIES") || getenv("ROCPROFILER_LIBRARY_CTOR");
    return on;
}
}  // namespace

void trace_push(const char* name) {
    if (ranges_on() && roctx().push) roctx().push(name);
}
void trace_pop() {
    if (ranges_on() && roctx().pop) roctx().pop();
}

// CRM_POISON=1 (GPU AddressSanitizer is not available for this target): every fresh allocation is filled with 0xFF bytes
// -- NaN as a double, -1 as an int -- instead of zeros, so that a read of memory nobody wrote shows up as a NaN result or as
// a range-checked index, and it is followed by a red zone of REDZONE bytes of the same fill that is inspected when the
// buffer is released: a kernel that wrote past the end of its buffer is counted (crm_test_overruns) and reported on stderr.
static bool poison_mode() {
    static const bool on = getenv("CRM_POISON") && atoi(getenv("CRM_POISON")) != 0;
    return on;
}
static constexpr size_t REDZONE = 4096;
static std::atomic<long> g_overruns{0};
long overruns_detected() { return g_overruns.load(); }

static void check_redzone(const void* ptr, size_t bytes) {
    if (!ptr || !poison_mode()) return;
    std::vector<unsigned char> rz(REDZONE);
    if (hipMemcpy(rz.data(), static_cast<const char*>(ptr) + bytes, REDZONE, hipMemcpyDeviceToHost) != hipSuccess) return;
    for (size_t i = 0; i < REDZONE; i++) {
        if (rz[i] != 0xFF) {
            g_overruns++;
            fprintf(stderr, "[crm] CRM_POISON: write past the end of a %zu-byte device buffer (first touched byte at +%zu)\n",
                    bytes, i);
            break;
        }
    }
}

// The fill of a fresh allocation runs on a stream of the library's own (non-blocking, one per device) and only that
// stream is waited for: on the legacy null stream every allocation would be an ordering point for all blocking streams of
// the process (torch's default stream among them).
static hipStream_t fill_stream() {
    static std::mutex mu;
    static std::map<int, hipStream_t> streams;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    auto it = streams.find(dev);
    if (it != streams.end()) return it->second;
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) st = nullptr;   // (null stream as a last resort)
    streams[dev] = st;
    return st;
}

int DevBuf::ensure(size_t need) {
    if (need <= bytes) return CRM_OK;
    if (ptr) {
        check_redzone(ptr, bytes);
        CRM_HIP(hipFree(ptr));
        ptr = nullptr;
        bytes = 0;
    }
    // (the 4 KiB behind the buffer are always there: zeros in a normal run -- a tile that over-reads its operand by a few
    // hundred bytes, as the test-forced 160-column tile of round 2 did, then stays inside the allocation instead of
    // depending on what the allocator mapped behind it -- and the inspected 0xFF red zone under CRM_POISON=1)
    const size_t total = need + REDZONE;
    hipError_t e = hipMalloc(&ptr, total);
    if (e == hipErrorOutOfMemory && trim_idle_workspaces() > 0) {
        (void)hipGetLastError();
        e = hipMalloc(&ptr, total);
    }
    if (e != hipSuccess) {
        ptr = nullptr;
        (void)hipGetLastError();
        set_error("device allocation of %zu bytes failed: %s", need, hipGetErrorString(e));
        return CRM_ERR_HIP;
    }
    bytes = need;
    // Fresh allocations never carry the previous tenant's bytes into a kernel: they are zero-filled (poison mode: 0xFF,
    // above).  The fill runs on the null stream and is waited for here: the contexts' streams are non-blocking and would
    // not order themselves behind it.
    hipStream_t fs = fill_stream();
    CRM_HIP(hipMemsetAsync(ptr, poison_mode() ? 0xFF : 0, total, fs));
    CRM_HIP(hipStreamSynchronize(fs));
    return CRM_OK;
}

static std::mutex g_ctx_mutex;
static std::vector<crm_ctx*> g_contexts;

static size_t trim_context(crm_ctx* c) {
    if (!c->eigh_ws || c->eigh_ws_busy) return 0;
    size_t freed = 0;
    for (const DevBuf* b : {&c->eigh_ws->A, &c->eigh_ws->Vt, &c->eigh_ws->Vc, &c->eigh_ws->QA, &c->eigh_ws->QB}) freed += b->bytes;
    eigh_free(*c->eigh_ws);
    return freed;
}

EighWork* acquire_eigh_workspace(crm_ctx* c) {
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    if (!c->eigh_ws) c->eigh_ws = new EighWork();
    c->eigh_ws_busy = true;
    return c->eigh_ws;
}

void release_eigh_workspace(crm_ctx* c) {
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    c->eigh_ws_busy = false;
}

size_t trim_idle_workspaces() {
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    size_t freed = 0;
    for (crm_ctx* c : g_contexts) freed += trim_context(c);
    return freed;
}
void DevBuf::release() {
    if (ptr) check_redzone(ptr, bytes);
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
}

// Upload a row-major host matrix [rows x cols] (leading dimension ld_src) into a zero-padded
// device matrix [rows_pad x ld_dst].
int upload_padded(hipStream_t st, double* dst, long ld_dst, long rows_pad, const double* src,
                  long ld_src, long rows, long cols) {
    CRM_HIP(hipMemsetAsync(dst, 0, sizeof(double) * rows_pad * ld_dst, st));
    if (rows > 0 && cols > 0)
        CRM_HIP(hipMemcpy2DAsync(dst, ld_dst * sizeof(double), src, ld_src * sizeof(double),
                                 cols * sizeof(double), rows, hipMemcpyHostToDevice, st));
    return CRM_OK;
}

}  // namespace crm

using namespace crm;

extern "C" {

const char* crm_last_error(void) { return last_error_text(); }
const char* crm_version(void) { return "0.6.0"; }

int crm_ctx_create(int device, crm_ctx** out) {
    return crm::guarded("crm_ctx_create", [&]() -> int {
    if (!out) return CRM_ERR_ARG;
    *out = nullptr;
    int count = 0;
    CRM_HIP(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) {
        set_error("crm_ctx_create: device %d not present (%d visible)", device, count);
        return CRM_ERR_ARG;
    }
    CRM_HIP(hipSetDevice(device));
    crm_ctx* c = new crm_ctx();
    c->device = device;
    // (the persistent form of large Khatri-Rao launches, the rotation routes and the tile walk are set through the C-ABI:
    // crm_test_set_contraction_sync, crm_set_fast_rotation)
    if (const char* e = getenv("CRM_KIN_ROUTE")) c->kin_route = std::max(0, std::min(2, atoi(e)));
    CRM_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    CRM_HIP(hipStreamCreateWithFlags(&c->upload_stream, hipStreamNonBlocking));
    CRM_HIP(hipEventCreate(&c->ev0));
    CRM_HIP(hipEventCreate(&c->ev1));
    {
        std::lock_guard<std::mutex> lock(g_ctx_mutex);
        g_contexts.push_back(c);
    }
    *out = c;
    return CRM_OK;
    });
}

int crm_ctx_trim(crm_ctx* c) {
    return crm::guarded_on("crm_ctx_trim", c, [&]() -> int {
    if (!c) return CRM_ERR_ARG;
    CRM_HIP(hipSetDevice(c->device));
    CRM_HIP(hipStreamSynchronize(c->stream));
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    (void)trim_context(c);
    return CRM_OK;
    });
}

void crm_ctx_destroy(crm_ctx* c) {
    try {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    {
        std::lock_guard<std::mutex> lock(g_ctx_mutex);
        g_contexts.erase(std::remove(g_contexts.begin(), g_contexts.end(), c), g_contexts.end());
        if (c->eigh_ws) {
            eigh_free(*c->eigh_ws);
            delete c->eigh_ws;
            c->eigh_ws = nullptr;
        }
    }
    for (auto* b : c->all_bufs()) b->release();
    for (auto& e : c->timed) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    (void)hipEventDestroy(c->ev0);
    (void)hipEventDestroy(c->ev1);
    (void)hipStreamDestroy(c->stream);
    if (c->upload_stream) (void)hipStreamDestroy(c->upload_stream);
    if (c->sync_timeouts_host) (void)hipHostFree(c->sync_timeouts_host);
    delete c;
    } catch (...) {  // (nothing may unwind into the caller; a destroy has no status to return)
    }
}

int crm_ctx_synchronize(crm_ctx* c) {
    return crm::guarded_on("crm_ctx_synchronize", c, [&]() -> int {
    if (!c) return CRM_ERR_ARG;
    CRM_HIP(hipSetDevice(c->device));
    CRM_HIP(hipStreamSynchronize(c->stream));
    return CRM_OK;
    });
}

int crm_set_block_variants(crm_ctx* c, int variants) {
    return crm::guarded_on("crm_set_block_variants", c, [&]() -> int {
    if (!c || variants < 0) return CRM_ERR_ARG;
    c->block_variants = variants == 0 ? 0 : (int)round_up(variants, 128);
    return CRM_OK;
    });
}

int crm_set_null_fit_polish(crm_ctx* c, int on) {
    return crm::guarded_on("crm_set_null_fit_polish", c, [&]() -> int {
    if (!c) return CRM_ERR_ARG;
    c->polish = on != 0;
    return CRM_OK;
    });
}

int crm_set_progress_callback(crm_ctx* c, void (*callback)(long, long, void*), void* user) {
    return crm::guarded_on("crm_set_progress_callback", c, [&]() -> int {
    if (!c) return CRM_ERR_ARG;
    if (callback) c->progress[std::this_thread::get_id()] = crm_ctx::Progress{callback, user};
    else c->progress.erase(std::this_thread::get_id());
    return CRM_OK;
    });
}

int crm_set_fast_rotation(crm_ctx* c, int on) {
    return crm::guarded_on("crm_set_fast_rotation", c, [&]() -> int {
    if (!c) return CRM_ERR_ARG;
    c->fast_T = on != 0;
    return CRM_OK;
    });
}

extern "C++" {
namespace crm {
namespace {
struct FormSlot { const char* name; int value; bool set; };
FormSlot g_forms[] = {{"gram_staged", 0, false}, {"kr_no_tail", 0, false}, {"nullfit_per_wave", 0, false}, {"kin_fold", 0, false},
                      {"eigh_one_stage", 0, false}, {"nullfit_exact", 0, false}, {"donor_pairs", 0, false},
                      {"pairs_without_kinship_term", 0, false}, {"flat_kappa_milli", 0, false}, {"chase_abort", 0, false}, {"nullfit_one_per_wave", 0, false}};
std::mutex g_forms_mu;
}  // namespace
int form(const char* name, int otherwise) {
    std::lock_guard<std::mutex> lock(g_forms_mu);
    for (const FormSlot& f : g_forms)
        if (strcmp(f.name, name) == 0) return f.set ? f.value : otherwise;
    return otherwise;
}
}  // namespace crm
}  // extern "C++"

int crm_test_set_form(const char* name, int value, int reset) {
    return crm::guarded("crm_test_set_form", [&]() -> int {
    if (!name) return CRM_ERR_ARG;
    std::lock_guard<std::mutex> lock(crm::g_forms_mu);
    for (crm::FormSlot& f : crm::g_forms)
        if (strcmp(f.name, name) == 0) {
            f.value = value;
            f.set = reset == 0;
            return CRM_OK;
        }
    crm::set_error("crm_test_set_form: no kernel form is called '%s'", name);
    return CRM_ERR_ARG;
    });
}

int crm_kernel_timer_reset(crm_ctx* c) {
    return crm::guarded_on("crm_kernel_timer_reset", c, [&]() -> int {
    if (!c) return CRM_ERR_ARG;
    CRM_HIP(hipSetDevice(c->device));
    CRM_HIP(hipStreamSynchronize(c->stream));
    c->timed_used = 0;
    c->kr_flops = 0.0;
    c->timing = true;
    return CRM_OK;
    });
}

int crm_kernel_timer_stop(crm_ctx* c) {
    return crm::guarded_on("crm_kernel_timer_stop", c, [&]() -> int {
    if (!c) return CRM_ERR_ARG;
    c->timing = false;  // later scans record nothing; the pairs recorded so far stay readable
    return CRM_OK;
    });
}

int crm_kernel_timer_read(crm_ctx* c, double* kr_ms, long* kr_launches, double* kr_flops,
                          double* total_ms) {
    return crm::guarded_on("crm_kernel_timer_read", c, [&]() -> int {
    if (!c) return CRM_ERR_ARG;
    CRM_HIP(hipSetDevice(c->device));
    CRM_HIP(hipStreamSynchronize(c->stream));
    double ms = 0.0;
    for (size_t i = 0; i < c->timed_used; i++) {
        float t = 0.f;
        CRM_HIP(hipEventElapsedTime(&t, c->timed[i].first, c->timed[i].second));
        ms += t;
    }
    if (kr_ms) *kr_ms = ms;
    if (kr_launches) *kr_launches = (long)c->timed_used;
    if (kr_flops) *kr_flops = c->kr_flops;
    if (total_ms) *total_ms = 0.0;
    return CRM_OK;
    });
}

// ---- single-kernel hooks -------------------------------------------------------------
int crm_test_set_contraction(crm_ctx* c, int tile_width, int lds_dma) {
    return crm::guarded_on("crm_test_set_contraction", c, [&]() -> int {
    if (!c || (tile_width != 0 && tile_width != 64 && tile_width != 128 && tile_width != 160)) return CRM_ERR_ARG;
    c->tune.bn = tile_width;
    c->tune.glds = lds_dma ? 1 : 0;
    return CRM_OK;
    });
}

long crm_test_sync_fallbacks(const crm_ctx* c) { return c ? c->sync_fallbacks : -1; }
long crm_test_overruns(void) { return crm::overruns_detected(); }

int crm_test_set_kinship_route(crm_ctx* c, int on) {
    return crm::guarded_on("crm_test_set_kinship_route", c, [&]() -> int {
    if (!c) return CRM_ERR_ARG;
    c->kin_route = on < 0 ? 0 : (on > 2 ? 2 : on);
    return CRM_OK;
    });
}

int crm_test_overrun_selftest(crm_ctx* c) {
    return crm::guarded_on("crm_test_overrun_selftest", c, [&]() -> int {
    if (!c) return CRM_ERR_ARG;
    CRM_HIP(hipSetDevice(c->device));
    const long before = overruns_detected();
    {
        ScopedBuf b;
        CRM_TRY(b.ensure(1000));
        if (poison_mode()) CRM_HIP(hipMemset(static_cast<char*>(b.ptr) + 1000, 0, 8));   // eight bytes past the end, on purpose
        CRM_HIP(hipDeviceSynchronize());
    }
    return (int)(overruns_detected() - before);
    });
}

int crm_test_check_context(crm_ctx* c) {
    return crm::guarded_on("crm_test_check_context", c, [&]() -> int {
    if (!c) return CRM_ERR_ARG;
    CRM_HIP(hipSetDevice(c->device));
    CRM_HIP(hipStreamSynchronize(c->stream));
    for (const DevBuf* b : c->all_bufs()) check_redzone(b->ptr, b->bytes);
    if (c->eigh_ws)
        for (const DevBuf* b : {&c->eigh_ws->A, &c->eigh_ws->Vt, &c->eigh_ws->Vc, &c->eigh_ws->QA, &c->eigh_ws->QB, &c->eigh_ws->d,
                                &c->eigh_ws->e, &c->eigh_ws->tau, &c->eigh_ws->lam, &c->eigh_ws->small})
            check_redzone(b->ptr, b->bytes);
    return CRM_OK;
    });
}

int crm_test_null_fit_probe(crm_ctx* c, int on, double x) {
    return crm::guarded_on("crm_test_null_fit_probe", c, [&]() -> int {
    if (!c) return CRM_ERR_ARG;
    c->probe_on = on != 0;
    c->probe_x = x;
    return CRM_OK;
    });
}

int crm_test_null_fit_probe_read(crm_ctx* c, double* out, long capacity) {
    return crm::guarded_on("crm_test_null_fit_probe_read", c, [&]() -> int {
    if (!c || !out || capacity < (long)c->probe_out.size()) return CRM_ERR_ARG;
    std::copy(c->probe_out.begin(), c->probe_out.end(), out);
    return (int)c->probe_out.size();
    });
}

int crm_test_set_contraction_sync(crm_ctx* c, int every) {
    return crm::guarded_on("crm_test_set_contraction_sync", c, [&]() -> int {
    if (!c || every < 0) return CRM_ERR_ARG;
    c->tune.sync = every;
    return CRM_OK;
    });
}

int crm_test_contract(crm_ctx* c, long cells, int M, int N, const double* X, const double* Y,
                      double* C, int ksplit) {
    return crm::guarded_on("crm_test_contract", c, [&]() -> int {
    if (!c || cells <= 0 || M <= 0 || N <= 0 || !X || !Y || !C || ksplit < 1) return CRM_ERR_ARG;
    CRM_HIP(hipSetDevice(c->device));
    const long cp = round_up(cells, GEMM_BK * (long)ksplit);
    const long ldx = round_up(M, 128), ldy = round_up(N, 128);
    ScopedBuf bx, by, bc, bp;
    CRM_TRY(bx.ensure(sizeof(double) * cp * ldx));
    CRM_TRY(by.ensure(sizeof(double) * cp * ldy));
    const long cstride = (long)M * ldy;
    CRM_TRY(bc.ensure(sizeof(double) * cstride * ksplit));
    CRM_TRY(bp.ensure(sizeof(GemmProblem)));
    CRM_TRY(upload_padded(c->stream, bx.as<double>(), ldx, cp, X, M, cells, M));
    CRM_TRY(upload_padded(c->stream, by.as<double>(), ldy, cp, Y, N, cells, N));
    GemmProblem p{};
    p.X = bx.as<double>(); p.Y = by.as<double>(); p.C = bc.as<double>();
    p.ldx = ldx; p.ldy = ldy; p.ldc = ldy; p.M = M; p.N = N;
    CRM_HIP(hipMemcpyAsync(bp.ptr, &p, sizeof p, hipMemcpyHostToDevice, c->stream));
    CRM_TRY(launch_gemm_tn(c, bp.as<GemmProblem>(), 1, M, N, cp, false, 0, ksplit, cstride));
    CRM_TRY(launch_reduce_splits(c->stream, bc.as<double>(), cstride, ksplit, cstride));
    CRM_HIP(hipMemcpy2DAsync(C, N * sizeof(double), bc.ptr, ldy * sizeof(double), N * sizeof(double),
                             M, hipMemcpyDeviceToHost, c->stream));
    CRM_HIP(hipStreamSynchronize(c->stream));
    return CRM_OK;
    });
}

int crm_test_contract_kr(crm_ctx* c, long cells, int B, int k0, int N, const double* G,
                         const double* E, const double* Y, double* C) {
    return crm::guarded_on("crm_test_contract_kr", c, [&]() -> int {
    if (!c || cells <= 0 || B <= 0 || k0 <= 0 || N <= 0 || !G || !E || !Y || !C) return CRM_ERR_ARG;
    CRM_HIP(hipSetDevice(c->device));
    const long cp = round_up(cells, GEMM_BK);
    const long ldg = round_up(B, 128) + 128, lde = round_up(k0, 32), ldy = round_up(N, 128);
    const int M = B * k0;
    ScopedBuf bg, be, by, bc, bp;
    CRM_TRY(bg.ensure(sizeof(double) * cp * ldg));
    CRM_TRY(be.ensure(sizeof(double) * cp * lde));
    CRM_TRY(by.ensure(sizeof(double) * cp * ldy));
    CRM_TRY(bc.ensure(sizeof(double) * (long)M * ldy));
    CRM_TRY(bp.ensure(sizeof(GemmProblem)));
    CRM_TRY(upload_padded(c->stream, bg.as<double>(), ldg, cp, G, B, cells, B));
    CRM_TRY(upload_padded(c->stream, be.as<double>(), lde, cp, E, k0, cells, k0));
    CRM_TRY(upload_padded(c->stream, by.as<double>(), ldy, cp, Y, N, cells, N));
    GemmProblem p{};
    p.X = bg.as<double>(); p.E = be.as<double>(); p.Y = by.as<double>(); p.C = bc.as<double>();
    p.ldx = ldg; p.lde = lde; p.ldy = ldy; p.ldc = ldy; p.M = M; p.N = N; p.k0 = k0;
    CRM_HIP(hipMemcpyAsync(bp.ptr, &p, sizeof p, hipMemcpyHostToDevice, c->stream));
    CRM_TRY(launch_gemm_tn(c, bp.as<GemmProblem>(), 1, M, N, cp, true, k0, 1, 0));
    CRM_HIP(hipMemcpy2DAsync(C, N * sizeof(double), bc.ptr, ldy * sizeof(double), N * sizeof(double),
                             M, hipMemcpyDeviceToHost, c->stream));
    CRM_HIP(hipStreamSynchronize(c->stream));
    return CRM_OK;
    });
}

// transposed store: CT is N x (B*k0) row-major
int crm_test_contract_kr_t(crm_ctx* c, long cells, int B, int k0, int N, const double* G, const double* E,
                           const double* Y, double* CT) {
    return crm::guarded_on("crm_test_contract_kr_t", c, [&]() -> int {
    if (!c || cells <= 0 || B <= 0 || k0 <= 0 || N <= 0 || !G || !E || !Y || !CT) return CRM_ERR_ARG;
    CRM_HIP(hipSetDevice(c->device));
    const long cp = round_up(cells, GEMM_BK);
    const long ldg = round_up(B, 128) + 128, lde = round_up(k0, 32), ldy = round_up(N, 128);
    const int M = B * k0;
    const long ldc = round_up(M, 128);
    ScopedBuf bg, be, by, bc, bp;
    CRM_TRY(bg.ensure(sizeof(double) * cp * ldg));
    CRM_TRY(be.ensure(sizeof(double) * cp * lde));
    CRM_TRY(by.ensure(sizeof(double) * cp * ldy));
    CRM_TRY(bc.ensure(sizeof(double) * (long)N * ldc));
    CRM_TRY(bp.ensure(sizeof(GemmProblem)));
    CRM_TRY(upload_padded(c->stream, bg.as<double>(), ldg, cp, G, B, cells, B));
    CRM_TRY(upload_padded(c->stream, be.as<double>(), lde, cp, E, k0, cells, k0));
    CRM_TRY(upload_padded(c->stream, by.as<double>(), ldy, cp, Y, N, cells, N));
    GemmProblem p{};
    p.X = bg.as<double>(); p.E = be.as<double>(); p.Y = by.as<double>(); p.C = bc.as<double>();
    p.ldx = ldg; p.lde = lde; p.ldy = ldy; p.ldc = ldc; p.M = M; p.N = N; p.k0 = k0;
    CRM_HIP(hipMemcpyAsync(bp.ptr, &p, sizeof p, hipMemcpyHostToDevice, c->stream));
    CRM_TRY(launch_kr_transposed(c, bp.as<GemmProblem>(), 1, M, N, cp, k0));
    CRM_HIP(hipMemcpy2DAsync(CT, M * sizeof(double), bc.ptr, ldc * sizeof(double), M * sizeof(double), N,
                             hipMemcpyDeviceToHost, c->stream));
    CRM_HIP(hipStreamSynchronize(c->stream));
    return CRM_OK;
    });
}

}  // extern "C"

// ---- eigenvalue / Davies hooks ---------------------------------------------------------------
#include "nullfit.h"

extern "C" {

int crm_test_eigvalsh(crm_ctx* c, int count, int k, const double* F, double* lambda) {
    return crm::guarded_on("crm_test_eigvalsh", c, [&]() -> int {
    if (!c || count <= 0 || k <= 0 || !F || !lambda) return CRM_ERR_ARG;
    CRM_HIP(hipSetDevice(c->device));
    ScopedBuf bF, bQ, bL, bP;
    CRM_TRY(bF.ensure(sizeof(double) * (size_t)count * k * k));
    CRM_TRY(bQ.ensure(sizeof(double) * count));
    CRM_TRY(bL.ensure(sizeof(double) * (size_t)count * k));
    CRM_TRY(bP.ensure(sizeof(double) * count));
    CRM_HIP(hipMemcpyAsync(bF.ptr, F, sizeof(double) * (size_t)count * k * k, hipMemcpyHostToDevice, c->stream));
    CRM_HIP(hipMemsetAsync(bQ.ptr, 0, sizeof(double) * count, c->stream));
    ScopedBuf bS;
    CRM_TRY(bS.ensure(sizeof(double) * eig_scratch_doubles(count, k)));
    CRM_TRY(launch_eig_davies(c->stream, bF.as<double>(), bQ.as<double>(), count, k, bL.as<double>(),
                              bP.as<double>(), nullptr, nullptr, true, bS.as<double>()));
    CRM_HIP(hipMemcpyAsync(lambda, bL.ptr, sizeof(double) * (size_t)count * k, hipMemcpyDeviceToHost, c->stream));
    CRM_HIP(hipStreamSynchronize(c->stream));
    return CRM_OK;
    });
}

int crm_test_davies(crm_ctx* c, int count, int k, const double* Q, const double* lambda, double* pvalue,
                    int* ifault, double* liu) {
    return crm::guarded_on("crm_test_davies", c, [&]() -> int {
    if (!c || count <= 0 || k <= 0 || !Q || !lambda || !pvalue) return CRM_ERR_ARG;
    CRM_HIP(hipSetDevice(c->device));
    ScopedBuf bQ, bL, bP, bI, bU;
    CRM_TRY(bQ.ensure(sizeof(double) * count));
    CRM_TRY(bL.ensure(sizeof(double) * (size_t)count * k));
    CRM_TRY(bP.ensure(sizeof(double) * count));
    CRM_TRY(bI.ensure(sizeof(int) * count));
    CRM_TRY(bU.ensure(sizeof(double) * count));
    CRM_HIP(hipMemcpyAsync(bQ.ptr, Q, sizeof(double) * count, hipMemcpyHostToDevice, c->stream));
    CRM_HIP(hipMemcpyAsync(bL.ptr, lambda, sizeof(double) * (size_t)count * k, hipMemcpyHostToDevice, c->stream));
    CRM_TRY(launch_eig_davies(c->stream, nullptr, bQ.as<double>(), count, k, bL.as<double>(), bP.as<double>(),
                              bI.as<int>(), bU.as<double>(), false));
    CRM_HIP(hipMemcpyAsync(pvalue, bP.ptr, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
    if (ifault) CRM_HIP(hipMemcpyAsync(ifault, bI.ptr, sizeof(int) * count, hipMemcpyDeviceToHost, c->stream));
    if (liu) CRM_HIP(hipMemcpyAsync(liu, bU.ptr, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
    CRM_HIP(hipStreamSynchronize(c->stream));
    return CRM_OK;
    });
}

}  // extern "C"
