// Stage 2 of the two-stage eigen-solver of the background constructor: band -> tridiagonal by bulge chasing, per grid
// point, and the driver of the whole family solve.  (Stage 1, the family argument and the reference lines this replaces:
// eigh2_band.hip; the back-transformation through the chase's reflectors: eigh2_back.hip.  numpy statement:
// tools/eigh2_prototype.py -- chase().)
//
// Chase (column-wise elimination; lower band storage AB[c][row - c], half-width w = 64, fill up to 2w - 1):
//   sweep s annihilates column s below its sub-diagonal with a reflector on rows s+1 .. s+w, then walks down the band:
//   at chain position k (rows r = s + 1 + k w ..) it applies the current reflector to the diagonal block from both sides
//   and to the block below from the right, which fills that block; the fill's first column is annihilated by the next
//   reflector, applied to the rest of the block from the left -- and so on to the end of the band.  Sweep s + 1 may run
//   position k as soon as sweep s has finished position k + 1, so the sweeps of one matrix are pipelined over workgroups
//   (workgroup g takes sweeps g, g + G, ...): a progress counter per sweep, written after the step's stores have
//   drained and polled by the successor (MI355X_MICROARCH.md, inter-workgroup visibility: every hand-off byte stored and
//   loaded sc1, the flag an sc1 store of one lane, a workgroup barrier between the poll and the loads; one workgroup per
//   CU).  All workgroups of a launch must be co-resident; a wait that runs out (a shared device) raises an abort flag and
//   the caller falls back to the one-stage solver.
#include <chrono>

#include "eigh.h"

namespace crm {
namespace {

typedef unsigned long long u64;
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int W = E2_W;            // 64
constexpr int CH_LD = 65;          // LDS row stride of the chase's blocks
constexpr int PROG_DONE = 1 << 30;

__device__ inline void st_sc1(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<u64*>(p), (u64)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline double ld_sc1(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const u64*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
// 16-byte accesses that another CU can see / that bypass this CU's L1: buffer instructions with the sc1 bit (aux = 16)
__device__ inline v2d ld_sc1_x2(__amdgpu_buffer_rsrc_t rs, int byte_off) {
    return __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16));
}
__device__ inline void st_sc1_x2(__amdgpu_buffer_rsrc_t rs, int byte_off, double x, double y) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, (v2d){x, y}), rs, byte_off, 0, 16);
}
// barrier of a workgroup for its LDS traffic only: global loads in flight (prefetches) stay in flight
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct ChaseArgs {
    double* AB;  long ab_slab;      // [batch][dimp + 128][128]
    double* V;   long v_slab;       // [batch][npos][dimp][64]
    double* tau; long tau_slab;     // [batch][npos][dimp]
    int* prog;   long prog_slab;    // [batch][dimp]
    int* abort_flag;
    long n, dimp;
    unsigned long long wait_ticks;  // bound of a workgroup's wait for its predecessor, in ticks of the 100 MHz wall clock
};

// dlarfg on a wavefront: lane i holds x_i (zero beyond len); returns v_i (v_0 = 1), tau and beta on every lane
__device__ inline double wave_house(double x, int lane, int len, double* tau_out, double* beta_out) {
    double ss = (lane >= 1 && lane < len) ? x * x : 0.0;
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    const double alpha = __shfl(x, 0, 64);
    double tau = 0.0, beta = alpha, scale = 0.0;
    if (ss > 0.0) {
        const double nrm = sqrt(alpha * alpha + ss);
        beta = alpha >= 0.0 ? -nrm : nrm;
        tau = (beta - alpha) / beta;
        scale = 1.0 / (alpha - beta);
    }
    *tau_out = tau;
    *beta_out = beta;
    if (lane == 0) return 1.0;
    return lane < len ? x * scale : 0.0;
}

// grid (G, batch), 256 threads, one workgroup per CU (the dynamic LDS request sees to that).
// Latencies kept off the critical path of a step: the workgroup's barriers wait for LDS traffic only (a flag store or a
// prefetch in flight does not hold them up); the predecessor's progress is re-read in the background once per step, so the
// blocking poll is the exception; and when that progress already covers the NEXT step, its two blocks -- which this step
// does not touch -- are fetched into registers while this step computes.
__global__ __launch_bounds__(256) void e2_chase_kernel(ChaseArgs a) {
    extern __shared__ double sm[];
    double* Dm = sm;                    // [64][CH_LD] diagonal block, both triangles
    double* Bm = Dm + W * CH_LD;        // [64][CH_LD] the block below
    double* v = Bm + W * CH_LD;         // [64] current reflector
    double* v1 = v + W;                 // [64] next reflector
    double* pu = v1 + W;                // [128] (D; B) v
    double* part = pu + 2 * W;          // [2][128]
    double* qv = part + 4 * W;          // [64]
    double* zv = qv + W;                // [64]
    double* zpart = zv + W;             // [4][64]
    double* scal = zpart + 4 * W;       // [8]
    __shared__ int wait_val, wait_ok;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = gridDim.x, b = blockIdx.y;
    const long n = a.n;
    double* AB = a.AB + (size_t)b * a.ab_slab;
    double* Vout = a.V + (size_t)b * a.v_slab;
    double* tauout = a.tau + (size_t)b * a.tau_slab;
    int* prog = a.prog + (size_t)b * a.prog_slab;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(AB, 0, (int)(a.ab_slab * 8), 0x00020000);

    for (long s = blockIdx.x; s < n - 2; s += G) {
        int seen = s > 0 ? 0 : PROG_DONE;      // progress of sweep s - 1 as last observed (the same on every thread)
        // blocking form: poll until the predecessor has finished `need` steps
        auto ensure = [&](int need) -> bool {
            if (seen >= need) return true;
            if (tid == 0) {
                int ok = 1, val;
                long spins = 0;
                // (bounded by wall clock -- the constant 100 MHz counter: a whole chase takes 0.1 - 0.4 s, a workgroup whose
                // predecessor is not resident -- a GPU shared with another process -- gives up after CHASE_WAIT_TICKS and the
                // constructor falls back to the one-stage solver; a spin count would take close to a minute to get there)
                const unsigned long long t_start = wall_clock64();
                while ((val = __hip_atomic_load(prog + s - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < need) {
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 1023) == 0) {
                        if (wall_clock64() - t_start > a.wait_ticks || __hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                            __hip_atomic_store(a.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ok = 0;
                            break;
                        }
                    }
                }
                wait_val = val;
                wait_ok = ok;
            }
            lds_barrier();
            seen = wait_val;
            const bool ok = wait_ok != 0;
            lds_barrier();
            return ok;
        };
        auto fetch = [&](long r_, v2d (&x)[16]) {
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int e2 = tid + 256 * q, j = e2 >> 6, o = (e2 & 63) * 2;
                x[q] = ld_sc1_x2(rs, (int)(((r_ + j) * 128 + o) * 8));
            }
        };
        if (!ensure(2)) return;
        long r = s + 1;
        int L = (int)min((long)W, n - r);
        // the sweep's own reflector from column s
        if (wave == 0) {
            const double x = lane < L ? ld_sc1(AB + s * 128 + 1 + lane) : 0.0;
            double t, beta;
            const double vv = wave_house(x, lane, L, &t, &beta);
            v[lane] = vv;
            if (lane == 0) { scal[0] = t; st_sc1(AB + s * 128 + 1, beta); }
            else if (lane < L) st_sc1(AB + s * 128 + 1 + lane, 0.0);
        }
        lds_barrier();
        double tau = scal[0];
        int k = 0;
        bool have = false;
        v2d xr[16];
        while (true) {
            const long r1 = r + L;
            const int L1 = (int)max(0L, min((long)W, n - r1));
            if (!have) {
                if (!ensure(k + 2)) return;
                fetch(r, xr);
            }
            int peek = 0;
            if (tid == 0 && s > 0) peek = __hip_atomic_load(prog + s - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // ---- the two blocks into LDS (the diagonal one mirrored) -------------------------------------------------------
            if (L < W || L1 < W) {      // (the end of the band: partial blocks, the rest of the LDS image must read zero)
                for (int e = tid; e < W * CH_LD; e += 256) { Dm[e] = 0.0; Bm[e] = 0.0; }
                lds_barrier();
            }
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int e2 = tid + 256 * q, j = e2 >> 6, o = (e2 & 63) * 2;
                if (j >= L) continue;
                for (int h = 0; h < 2; h++) {
                    const int i = j + o + h;
                    if (i >= L + L1) continue;
                    if (i < L) { Dm[i * CH_LD + j] = xr[q][h]; Dm[j * CH_LD + i] = xr[q][h]; }
                    else Bm[(i - L) * CH_LD + j] = xr[q][h];
                }
            }
            // the next step's blocks, if the predecessor is known to be far enough ahead already
            have = L1 > 0 && seen >= k + 3;
            if (have) fetch(r1, xr);
            lds_barrier();
            // ---- (D; B) v ----------------------------------------------------------------------------------------------
            {
                const int i = tid & 127, half = tid >> 7;
                const double* row = i < W ? Dm + i * CH_LD : Bm + (i - W) * CH_LD;
                double acc = 0.0;
                for (int j = half * 32; j < half * 32 + 32; j++) acc += row[j] * v[j];
                part[half * 128 + i] = acc;
            }
            lds_barrier();
            if (tid < 128) pu[tid] = part[tid] + part[128 + tid];
            lds_barrier();
            if (wave == 0) {    // p = tau D v,  q = p - 1/2 tau (p'v) v
                const double p = tau * pu[lane];
                double dot = p * v[lane];
                for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
                qv[lane] = p - 0.5 * tau * dot * v[lane];
                if (lane == 0 && s > 0) wait_val = peek;      // (the background read of the predecessor's progress has landed)
            }
            lds_barrier();
            if (s > 0) seen = max(seen, wait_val);
            {
                const int i = tid & 63, cq = tid >> 6;
                const double vi = v[i], qi = qv[i], ui = tau * pu[W + i];
                for (int j = cq * 16; j < cq * 16 + 16; j++) {
                    Dm[i * CH_LD + j] -= vi * qv[j] + qi * v[j];
                    Bm[i * CH_LD + j] -= ui * v[j];
                }
            }
            lds_barrier();
            // ---- the fill's first column -> next reflector; the rest of the block from the left -------------------------
            double tau1 = 0.0;
            if (L1 > 0) {
                if (wave == 0) {
                    const double x = lane < L1 ? Bm[lane * CH_LD] : 0.0;
                    double t, beta;
                    const double vv = wave_house(x, lane, L1, &t, &beta);
                    v1[lane] = vv;
                    if (lane < L1) Bm[lane * CH_LD] = lane == 0 ? beta : 0.0;
                    if (lane == 0) scal[1] = t;
                }
                lds_barrier();
                tau1 = scal[1];
                {
                    const int j = tid & 63, rq = tid >> 6;
                    double acc = 0.0;
                    for (int i = rq * 16; i < rq * 16 + 16; i++) acc += v1[i] * Bm[i * CH_LD + j];
                    zpart[rq * 64 + j] = acc;
                }
                lds_barrier();
                if (tid < 64) zv[tid] = tid >= 1 ? tau1 * (zpart[tid] + zpart[64 + tid] + zpart[128 + tid] + zpart[192 + tid]) : 0.0;
                lds_barrier();
                {
                    const int i = tid & 63, cq = tid >> 6;
                    const double vi = v1[i];
                    for (int j = cq * 16; j < cq * 16 + 16; j++) Bm[i * CH_LD + j] -= vi * zv[j];
                }
                lds_barrier();
            }
            // ---- store: the band (slots past the two blocks hold zeros: nothing fills them), the reflector of this step -----
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int e2 = tid + 256 * q, j = e2 >> 6, o = (e2 & 63) * 2, i = j + o;
                if (j >= L || i >= L + L1) continue;
                double x[2];
                for (int h = 0; h < 2; h++) {
                    const int ih = i + h;
                    x[h] = ih < L ? Dm[ih * CH_LD + j] : (ih < L + L1 ? Bm[(ih - L) * CH_LD + j] : 0.0);
                }
                st_sc1_x2(rs, (int)(((r + j) * 128 + o) * 8), x[0], x[1]);
            }
            if (tid < W) Vout[((size_t)k * a.dimp + s) * W + tid] = tid < L ? v[tid] : 0.0;
            if (tid == 0) tauout[(size_t)k * a.dimp + s] = tau;
            // every wave's stores drained, then ONE lane raises the sweep's progress
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
            if (tid == 0) __hip_atomic_store(prog + s, L1 > 0 ? k + 1 : PROG_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (L1 <= 0) break;
            if (tid < W) v[tid] = v1[tid];
            lds_barrier();
            tau = tau1;
            r = r1;
            L = L1;
            k++;
        }
        lds_barrier();
    }
}

__global__ void e2_diag_kernel(const double* __restrict__ AB, long ab_slab, long n, double* __restrict__ d, double* __restrict__ e,
                               long ld) {
    const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (c >= ld) return;
    d[(size_t)b * ld + c] = c < n ? AB[(size_t)b * ab_slab + c * 128] : 0.0;
    e[(size_t)b * ld + c] = c + 1 < n ? AB[(size_t)b * ab_slab + c * 128 + 1] : 0.0;
}

}  // namespace

static inline int chase_positions(long n) { return (int)((n - 1 + W - 1) / W); }   // chain positions 0 .. npos - 1

int eigh2_chase(crm_ctx* ctx, EighWork& w) {
    hipStream_t st = ctx->stream;
    const long n = w.dim, dimp = w.dimp;
    const int B = w.batch, npos = chase_positions(n);
    ChaseArgs a{};
    a.ab_slab = (long)(dimp + 128) * 128;
    a.v_slab = (long)npos * dimp * W;
    a.tau_slab = (long)npos * dimp;
    a.prog_slab = dimp;
    CRM_TRY(w.Vbc.ensure(sizeof(double) * (size_t)a.v_slab * B));
    CRM_TRY(w.taubc.ensure(sizeof(double) * (size_t)a.tau_slab * B));
    CRM_TRY(w.sync.ensure(sizeof(int) * ((size_t)B * dimp + 64)));
    a.AB = w.AB.as<double>(); a.V = w.Vbc.as<double>(); a.tau = w.taubc.as<double>();
    a.abort_flag = w.sync.as<int>();
    a.prog = w.sync.as<int>() + 64;
    a.n = n; a.dimp = dimp;
    // (test form "chase_abort": no patience at all -- the first workgroup that has to wait raises the flag, and the
    // constructor must come out of it through the one-stage solver with the same spectra)
    a.wait_ticks = form("chase_abort", 0) ? 0ull : E2_WAIT_TICKS;
    CRM_HIP(hipMemsetAsync(w.sync.ptr, 0, sizeof(int) * ((size_t)B * dimp + 64), st));
    CRM_HIP(hipMemsetAsync(w.Vbc.ptr, 0, sizeof(double) * (size_t)a.v_slab * B, st));
    CRM_HIP(hipMemsetAsync(w.taubc.ptr, 0, sizeof(double) * (size_t)a.tau_slab * B, st));
    // one workgroup per CU: all of them resident at once
    hipDeviceProp_t prop;
    CRM_HIP(hipGetDeviceProperties(&prop, ctx->device));
    const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    int G = std::max(1, cus / B);
    G = (int)std::min<long>(G, std::max<long>(1, (n - 2)));
    const size_t lds = 86 * 1024;   // (more than half of a CU's 160 KB: a second workgroup does not fit beside it)
    CRM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&e2_chase_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (n > 2) hipLaunchKernelGGL(e2_chase_kernel, dim3(G, B), dim3(256), lds, st, a);
    hipLaunchKernelGGL(e2_diag_kernel, dim3((unsigned)((w.ld + 255) / 256), B), dim3(256), 0, st, w.AB.as<double>(), a.ab_slab, n,
                       w.d.as<double>(), w.e.as<double>(), w.ld);
    CRM_HIP(hipGetLastError());
    int aborted = 0;
    CRM_HIP(hipMemcpyAsync(&aborted, a.abort_flag, sizeof(int), hipMemcpyDeviceToHost, st));
    CRM_HIP(hipStreamSynchronize(st));
    if (aborted) {
        set_error("two-stage eigen-solver: the chase's workgroups were not co-resident (a shared device?)");
        return CRM_ERR_UNSUPPORTED;
    }
    return CRM_OK;
}

int eigh2_family(crm_ctx* ctx, EighWork& w, const double* wa, const double* wb, double* lam_host, double** Zt) {
    const bool trace = getenv("CRM_TRACE_SETUP") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!trace) return;
        (void)hipStreamSynchronize(ctx->stream);
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[crm eigh2 %d x %ld] %-28s %.3f s\n", w.batch, w.dim, what, std::chrono::duration<double>(now - t0).count());
        t0 = now;
    };
    {
        TraceRange r("crm eigh2 dense to band");
        CRM_TRY(eigh2_to_band(ctx, w));
    }
    lap("dense to band (once)");
    {
        TraceRange r("crm eigh2 chase");
        CRM_TRY(eigh2_scale_band(ctx, w, wa, wb));
        CRM_TRY(eigh2_chase(ctx, w));
    }
    lap("band to tridiagonal");
    double* Qt = nullptr;
    {
        TraceRange r("crm eigh divide & conquer");
        CRM_TRY(eigh_dc(ctx, w, lam_host, &Qt));
    }
    lap("divide & conquer");
    {
        TraceRange r("crm eigh2 back-transformation (chase)");
        CRM_TRY(eigh_rows_to_columns(ctx, w, Qt));
        CRM_TRY(eigh2_back_chase(ctx, w, w.A.as<double>()));
    }
    lap("back-transformation 2");
    {
        TraceRange r("crm eigh2 back-transformation (band)");
        w.v_shared = true;
        const int rc = eigh_back_transform(ctx, w, Qt, Zt, true);
        w.v_shared = false;
        CRM_TRY(rc);
    }
    lap("back-transformation 1");
    return CRM_OK;
}

}  // namespace crm

// ---- test hook: the family solver on a host matrix ---------------------------------------------------------------------
// C: dim x dim (row-major, symmetric), whose first 64 coordinates are the leading block; wa / wb: nq weights.
// stage 0: everything (lam: nq x dim ascending, Z: nq x dim x dim, column j = eigenvector j; may be NULL);
// stage 1: dense -> band only (band_out: dim x dim, the lower triangle of Q1' C Q1);
// stage 2: ... and the chase (d_out / e_out: nq x dim).
extern "C" int crm_test_eigh2(crm_ctx* ctx, int nq, int dim, const double* C, const double* wa, const double* wb, double* lam,
                              double* Z, int stage, double* d_out, double* e_out, double* band_out) {
    return crm::guarded_on("crm_test_eigh2", ctx, [&]() -> int {
    using namespace crm;
    if (!ctx || nq < 1 || dim < 1 || !C || !wa || !wb) return CRM_ERR_ARG;
    CRM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    EighWork w;
    struct Guard { EighWork& w; ~Guard() { eigh_free(w); } } guard{w};
    CRM_TRY(eigh_alloc(w, nq, dim));
    CRM_HIP(hipMemsetAsync(w.A.ptr, 0, sizeof(double) * (size_t)nq * w.slab, st));
    CRM_HIP(hipMemcpy2DAsync(w.A.as<double>(), w.ld * sizeof(double), C, dim * sizeof(double), dim * sizeof(double), dim,
                             hipMemcpyHostToDevice, st));
    if (stage == 1 || stage == 2) {
        CRM_TRY(eigh2_to_band(ctx, w));
        if (band_out) CRM_HIP(hipMemcpy2DAsync(band_out, dim * sizeof(double), w.A.ptr, w.ld * sizeof(double), dim * sizeof(double), dim,
                                               hipMemcpyDeviceToHost, st));
        CRM_HIP(hipStreamSynchronize(st));
        if (stage == 1) return CRM_OK;
        CRM_TRY(eigh2_scale_band(ctx, w, wa, wb));
        CRM_TRY(eigh2_chase(ctx, w));
        if (d_out) CRM_HIP(hipMemcpy2DAsync(d_out, dim * sizeof(double), w.d.ptr, w.ld * sizeof(double), dim * sizeof(double), nq,
                                            hipMemcpyDeviceToHost, st));
        if (e_out) CRM_HIP(hipMemcpy2DAsync(e_out, dim * sizeof(double), w.e.ptr, w.ld * sizeof(double), dim * sizeof(double), nq,
                                            hipMemcpyDeviceToHost, st));
        CRM_HIP(hipStreamSynchronize(st));
        return CRM_OK;
    }
    if (!lam) return CRM_ERR_ARG;
    double* Zt = nullptr;
    CRM_TRY(eigh2_family(ctx, w, wa, wb, lam, &Zt));
    if (Z) {
        std::vector<double> rows((size_t)dim * dim);
        for (int b = 0; b < nq; b++) {
            CRM_HIP(hipMemcpy2D(rows.data(), dim * sizeof(double), Zt + (size_t)b * w.slab, w.ld * sizeof(double),
                                dim * sizeof(double), dim, hipMemcpyDeviceToHost));
            double* out = Z + (size_t)b * dim * dim;
            for (int j = 0; j < dim; j++)
                for (int r = 0; r < dim; r++) out[(size_t)r * dim + j] = rows[(size_t)j * dim + r];
        }
    }
    return CRM_OK;
    });
}
