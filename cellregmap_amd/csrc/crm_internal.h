// Internal object layouts behind the opaque C-ABI handles.
#pragma once
#include <map>
#include <mutex>
#include <thread>
#include <utility>

#include "crm_common.h"

namespace crm {

constexpr int CRM_DEFAULT_BLOCK = 1024;  // variants per internal batch of the association scans
constexpr int CRM_MAX_AUTO_BLOCK = 4096;  // interaction scan: largest automatic batch (see scan_core)
constexpr int CRM_MAX_RHO = 16;    // rho grid points (the reference uses 1 or 11)
constexpr int CRM_MAX_COV = 8;    // columns of W the register null-fit kernel is instantiated for
constexpr int CRM_MAX_COV_WIDE = 62;  // columns of W of the interaction scan (beyond CRM_MAX_COV: the LDS null-fit kernel)
constexpr int CRM_MAX_COV_XWIDE = 128; // columns of W of the association scans and LMM fits (beyond 62: nullfit_xwide.hip)
constexpr int CRM_MAX_K0 = 256;   // contexts (columns of E0); past 128 through slower forms of four kernels
constexpr int CRM_MAX_GRAM_ROWS = 288;   // contexts + covariate columns + 2 in the interaction scan (Gram kernel's LDS image)
constexpr int DT_SUMS_LD = 136;        // columns of the per-donor sums table (1 + 1 + c, c <= CRM_MAX_COV_XWIDE = 128)
constexpr int BLOCK_SLACK_MAX = 4096;  // groups of a donor-constant panel

struct DevBuf {
    void* ptr = nullptr;
    size_t bytes = 0;
    int ensure(size_t need);
    void release();
    template <class T>
    T* as() const { return static_cast<T*>(ptr); }
};

// temporary device buffer released on scope exit (error paths included)
struct ScopedBuf : DevBuf {
    ScopedBuf() = default;
    ScopedBuf(const ScopedBuf&) = delete;
    ScopedBuf& operator=(const ScopedBuf&) = delete;
    ~ScopedBuf() { release(); }
};

int upload_padded(hipStream_t st, double* dst, long ld_dst, long rows_pad, const double* src,
                  long ld_src, long rows, long cols);

struct EighWork;
// Release the idle cached workspaces of every context (the eigen-solver's); returns the bytes handed back.
// DevBuf::ensure calls it once before giving up on an allocation.
size_t trim_idle_workspaces();
// CRM_POISON=1: device buffers whose red zone was found overwritten when they were released
long overruns_detected();
}  // namespace crm
struct crm_ctx;
namespace crm {
// The context's cached eigen-solver workspace, marked busy / idle under the same lock the trimming takes (a context that
// runs out of memory on one thread must not free the buffers another thread's constructor has just picked up).
EighWork* acquire_eigh_workspace(crm_ctx* ctx);
void release_eigh_workspace(crm_ctx* ctx);

}  // namespace crm

struct crm_ctx {
    // Every entry point that works on a context (directly or through a background / gene / panel of it) holds this lock
    // for the duration of the call: calls on one context are serialised by the library, whatever threads they come from
    // (recursive: an entry may call another one; a scan started from a progress callback is refused separately, in_scan).
    std::recursive_mutex mu;
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t upload_stream = nullptr;   // crm_panel_create copies on it, outside the context's lock: a panel can be
                                           // uploaded from one thread while another one runs the constructor
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int block_variants = 0;  // 0 = automatic
    bool fast_T = true;    // T(rho) through the mixing matrices when the background offers them
    int kin_route = 1;          // H'(g o E0) donor by donor when the background knows its kinship structure: 0 never
                                // (CRM_KIN_ROUTE=0), 1 when its flop count pays (scan.hip), 2 always
    bool fast_gene_rot = true;  // Q0(rho)'[y, W] of a gene through the mixing matrices as well (else against Q0 itself)
    bool collapse = true;  // use the donor-collapsed path for grouped panels
    // Progress callbacks, one per calling thread (crm_set_progress_callback installs it for the thread that calls it; a
    // scan reports to the callback of the thread it runs on): two threads scanning on one device each see their own bar.
    struct Progress { void (*fn)(long done, long total, void* user) = nullptr; void* user = nullptr; };
    std::map<std::thread::id, Progress> progress;
    bool progress_muted = false;   // a scan's second (dense) pass over variants the first one already reported
    void report(long done, long total) {   // (called with the context's lock held)
        if (progress_muted) return;
        auto it = progress.find(std::this_thread::get_id());
        if (it != progress.end() && it->second.fn) it->second.fn(done, total, it->second.user);
    }
    bool in_scan = false;  // a scan is running on this context (its work buffers are in use: no second one from a callback)
    bool polish = false;  // opt-in: refine the null-fit optimum beyond Brent's 1e-6 (nullfit.hip)
    bool probe_on = false;       // crm_test_null_fit_probe: scans stop after the null-fit kernels and keep the trial records
    double probe_x = 0.0;
    std::vector<double> probe_out;   // [variants x nrho x 2]: lml, scale at probe_x (last scanned block)
    bool nullfit_exact = false;  // null-fit likelihood in the reference's own operations (IEEE division, one log per entry)
    crm::GemmTune tune;   // contraction kernel variant (test hooks only change it)
    crm::DevBuf sync_counters;  // per-XCD generation counters of the persistent contraction form ([8]: waits that ran out)
    unsigned* sync_timeouts_host = nullptr;  // pinned copy of counters[8] of the last persistent launch
    long sync_fallbacks = 0;                 // times the context left the persistent form because its waits timed out
    // Work buffers of the constructor's eigen-solver, kept between constructor calls: handing 45 GB (config 5: five slabs
    // of 11 x 10 050^2) back to the driver costs 1.3 s per call and mapping them again up to as much; per-SNP
    // backgrounds of the effect-size path call the constructor once per variant.  Released by crm_ctx_trim,
    // crm_ctx_destroy, or when another allocation would otherwise fail.
    long dense_repeats = 0;   // variants a collapsed scan repeated on the dense path (nearly collinear with W)
    long tail_launches = 0;   // blocks whose last columns took the 160-column-tile launch (crm_test_tail_launches)
    long spectrum_tail_launches = 0;   // blocks whose last few columns of the spectrum went through the skinny one-pass kernel
    long donor_pair_blocks = 0;   // blocks whose per-donor sums came from the symmetric pair features (crm_test_donor_pair_blocks)
    long tests_without_pair = 0;  // (phenotype, variant) tests whose fit has no kinship term to speak of: no A~ formed for them
    // crm_scan_interaction_permuted: what the scan of a block computes BEFORE the permutation hooks enter -- the eleven
    // rotations T(rho) = G'Q0(rho), the null fits and rho* (cellregmap/_cellregmap.py:345-357 sit above the hooks at
    // :398-413) -- is recorded by the first permutation's pass (per block: the fit records and the rows T(rho*)) and
    // replayed by the passes of the other permutations, which visit the same blocks in the same order.
    struct ReplayBlock {
        long col0 = 0;
        int nb = 0;
        bool collapsed = false;
        std::vector<char> fit;   // NullFitOut[nb]
        crm::DevBuf T;           // [nb x ldq]: row b = Q0(rho*(b))' g_b
    };
    int replay_mode = 0;         // 0 off, 1 record, 2 replay
    size_t replay_cursor = 0;
    std::vector<ReplayBlock*> replay_blocks;
    void replay_clear() {
        for (ReplayBlock* b : replay_blocks) { b->T.release(); delete b; }
        replay_blocks.clear();
        replay_cursor = 0;
        replay_mode = 0;
    }
    crm::EighWork* eigh_ws = nullptr;
    bool eigh_ws_busy = false;
    // per-launch event pairs around the dominant kernel (bench.py's roofline leg)
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timed;
    size_t timed_used = 0;
    double kr_flops = 0.0;
    // scan workspace (grown on demand, reused across calls)
    crm::DevBuf ws_T, ws_A, ws_Gb, ws_Gx, ws_Gs, ws_G2, ws_GG, ws_Gt, ws_Z, ws_small, ws_probs, ws_F, ws_Gext, ws_TH, ws_AH, ws_XG;
    crm::DevBuf ws_xwide;  // scratch of the 63..128-column null-fit kernel (nullfit_xwide.hip)
    crm::DevBuf ws_Tcut;   // rotations: the few small products taken out of the batched launch (cut along the contraction axis)
    crm::DevBuf ws_Gk, ws_S, ws_S2;   // kinship-structure route: the block in donor order, the per-donor sums (step 6 / step 3)
    crm::DevBuf ws_Pd;                // ... the per-donor products against the symmetric pair features (step 6, donor pairs)
    crm::DevBuf ws_Anone;             // a row of zeros: A~ of the tests whose fit has no kinship term (AssembleArgs::A_none)
    std::vector<crm::DevBuf*> all_bufs() {
        return {&sync_counters, &ws_xwide, &ws_Tcut, &ws_S, &ws_S2, &ws_Gk, &ws_Pd, &ws_Anone, &ws_AH, &ws_XG, &ws_TH, &ws_T, &ws_A, &ws_Gb, &ws_Gx, &ws_Gs, &ws_G2, &ws_GG, &ws_Gt, &ws_Z, &ws_small, &ws_probs, &ws_F, &ws_Gext};
    }
};

namespace crm {
// crm::guarded (no C++ exception across the C-ABI) with the context's lock held around the body; ctx may be null (the
// body then reports the bad argument itself).
template <class Body>
inline int guarded_on(const char* entry, crm_ctx* ctx, Body&& body) noexcept {
    try {
        if (!ctx) return guarded(entry, body);
        std::lock_guard<std::recursive_mutex> lock(ctx->mu);
        return guarded(entry, body);
    } catch (...) {   // (std::system_error from the lock)
        set_error("%s: could not take the context's lock", entry);
        return CRM_ERR_INTERNAL;
    }
}
}  // namespace crm
