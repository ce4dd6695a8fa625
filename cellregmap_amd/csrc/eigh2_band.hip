// Stage 1 of the two-stage eigen-solver of the background constructor: dense -> band, ONCE for all grid points.
//
// The grid points of a background are one family  A(rho) = D(rho) C D(rho),  D = diag(sqrt(rho) on the contexts' block,
// sqrt(1 - rho) on the rest)  (cellregmap/_cellregmap.py:101-131: hS(rho) = [sqrt(rho) E1, sqrt(1 - rho) B]; C is the Gram
// matrix of [E1, B]; numpy_sugar.economic_qs_linear, in-tree twin _math.py:238-256, decomposes each one from scratch).
// The one-stage tridiagonalisation (eigh_trd.hip) streams half of every trailing matrix once per column: n^3 / 6 doubles
// per grid point, HBM-bound.  Here C is reduced to a band of half-width E2_W = 64 by panel QR + two-sided compact-WY
// updates -- three products per panel on the FP64 matrix pipe -- and because the first panel is exactly the leading
// block's 64 columns, every reflector acts on rows >= 64 only: Q1 = diag(I_64, Q2) commutes with D(rho), so
//     Q1' A(rho) Q1 = D(rho) (Q1' C Q1) D(rho)
// is the SAME band rescaled.  The n^3 part of the reduction is done once instead of once per grid point; what remains
// per grid point is O(n^2 w) (eigh2_chase.hip).  tools/eigh2_prototype.py is the numpy statement of the whole scheme.
//
// Per panel (columns c0 .. c0 + 63, reflectors on rows r0 = c0 + 64 ..):
//   e2_panel_qr    Householder QR of the m x 64 panel, rows split over workgroups that keep their 256 rows in LDS; ONE
//                  grid-wide reduction per column (the sums over the rows BELOW the pivot of column j against columns
//                  j .. 63 give the norm and every v'P_k at once), then V'V the same way and the dlarft recurrence -> T.
//   contraction    W = A22' V                              (2 m^2 64 flops, FP64 MFMA, gemm_tn*.hip)
//   e2_vtw/_small/_z   K = T'(V'W)T,  Z = W T - 1/2 V K,  panels [V; Z]' and [Z; V]' as rows
//   contraction    A22 -= V Z' + Z V'   ==  [V; Z]' [Z; V]   (GEMM_SUBTRACT, both triangles)
#include <chrono>

#include "eigh.h"

namespace crm {
namespace {

constexpr int QR_ROWS = 256;   // rows of the panel per workgroup
constexpr int QR_LD = 65;      // LDS row stride (doubles)
constexpr int CH_ROWS = 128;   // rows per workgroup of the small panel kernels

typedef unsigned long long u64;

__device__ inline void st_sc1(double* p, double v) {   // write-through store another CU can see (agent scope)
    __hip_atomic_store(reinterpret_cast<u64*>(p), (u64)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline double ld_sc1(const double* p) {     // load that bypasses this CU's L1
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const u64*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// Barrier over the workgroups of one launch (all co-resident: one per CU, at most 64 of them).  Every wave drains its
// stores, one lane adds to the counter and polls it (MI355X_MICROARCH.md, inter-workgroup visibility: sc1 payload,
// agent-scope counter, workgroup barrier before the loads).  Bounded: a wait that runs out raises *abort and returns.
__device__ inline bool grid_barrier(unsigned* ctr, unsigned target, int* abort_flag) {
    __shared__ int gb_ok;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 1;
        long spins = 0;
        const unsigned long long t_start = wall_clock64();   // (100 MHz: the wait is bounded by wall clock, E2_WAIT_TICKS)
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 1023) == 0) {
                if (wall_clock64() - t_start > E2_WAIT_TICKS || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0;
                    break;
                }
            }
        }
        gb_ok = ok;
    }
    __syncthreads();
    return gb_ok != 0;
}

struct QrArgs {
    double* A;        // the matrix (slab 0), leading dimension ld
    double* Vc;       // column c0 + k = reflector k of the panel (dimp x ld)
    double* Vt;       // row c0 + k = the same
    double* tau;      // [ld]
    double* T;        // this panel's 64 x 64 (row-major, upper triangular)
    double* part;     // [64][maxwg][64]   partial sums per column, workgroup
    double* toprow;   // [64][64]          pivot rows as the columns are reached
    double* spart;    // [maxwg][64 * 64]  partial V'V
    unsigned* bar;    // two counters
    int* abort_flag;
    long ld, dim, c0, r0;
    int maxwg;
};

// one workgroup per 256 rows of the panel, 1024 threads
__global__ __launch_bounds__(1024) void e2_panel_qr_kernel(QrArgs a) {
    extern __shared__ double sm[];
    double* P = sm;                          // [QR_ROWS][QR_LD]
    double* red = P + QR_ROWS * QR_LD;       // [16][64]
    double* cvec = red + 16 * 64;            // [64] column sums
    double* wvec = cvec + 64;                // [64]
    double* top = wvec + 64;                 // [64]
    double* taus = top + 64;                 // [64]
    double* scal = taus + 64;                // [4]: scale, beta
    const int tid = threadIdx.x, wg = blockIdx.x, nwg = gridDim.x;
    const int k = tid & 63, rg = tid >> 6;   // column, row group (16 rows each)
    const long row0 = a.r0 + (long)wg * QR_ROWS;
    const int nrows = (int)min((long)QR_ROWS, a.dim - row0);
    for (int i = 0; i < 16; i++) {
        const int r = rg * 16 + i;
        P[r * QR_LD + k] = r < nrows ? a.A[(row0 + r) * a.ld + a.c0 + k] : 0.0;
    }
    __syncthreads();
    for (int j = 0; j < E2_W; j++) {
        // sums over the rows strictly below the pivot row r0 + j
        const int lo = wg == 0 ? j + 1 : 0;
        double acc = 0.0;
        if (k >= j) {
            for (int i = 0; i < 16; i++) {
                const int r = rg * 16 + i;
                if (r >= lo && r < nrows) acc += P[r * QR_LD + j] * P[r * QR_LD + k];
            }
        }
        red[rg * 64 + k] = acc;
        __syncthreads();
        if (tid < 64) {
            double s = 0.0;
            for (int g = 0; g < 16; g++) s += red[g * 64 + tid];
            st_sc1(a.part + ((long)j * a.maxwg + wg) * 64 + tid, s);
            if (wg == 0) st_sc1(a.toprow + j * 64 + tid, P[j * QR_LD + tid]);
        }
        if (!grid_barrier(a.bar, (unsigned)(nwg * (j + 1)), a.abort_flag)) return;
        // every workgroup forms the same sums in the same order
        {
            double s = 0.0;
            for (int g = rg; g < nwg; g += 16) s += ld_sc1(a.part + ((long)j * a.maxwg + g) * 64 + k);
            red[rg * 64 + k] = s;
            if (rg == 0) top[k] = ld_sc1(a.toprow + j * 64 + k);
        }
        __syncthreads();
        if (tid < 64) {
            double s = 0.0;
            for (int g = 0; g < 16; g++) s += red[g * 64 + tid];
            cvec[tid] = s;
        }
        __syncthreads();
        if (tid < 64) {
            const double alpha = top[j], xnorm2 = cvec[j];
            double tau = 0.0, beta = alpha, scale = 0.0;
            if (xnorm2 > 0.0) {
                const double nrm = sqrt(alpha * alpha + xnorm2);
                beta = alpha >= 0.0 ? -nrm : nrm;
                tau = (beta - alpha) / beta;
                scale = 1.0 / (alpha - beta);
            }
            wvec[tid] = tid > j ? tau * (top[tid] + scale * cvec[tid]) : 0.0;
            if (tid == 0) {
                scal[0] = scale;
                scal[1] = beta;
                taus[j] = tau;
                if (wg == 0) a.tau[a.c0 + j] = tau;
            }
        }
        __syncthreads();
        {
            const double scale = scal[0], beta = scal[1], wk = wvec[k];
            for (int i = 0; i < 16; i++) {
                const int r = rg * 16 + i;
                if (r >= nrows) continue;
                if (wg == 0 && r == j) {             // the pivot row: v = 1
                    if (k > j) P[r * QR_LD + k] -= wk;
                    else if (k == j) P[r * QR_LD + k] = beta;
                } else if (r >= lo) {
                    const double vr = P[r * QR_LD + j] * scale;   // (the lane k == j of this wave overwrites it below)
                    if (k > j) P[r * QR_LD + k] -= vr * wk;
                    else if (k == j) P[r * QR_LD + k] = vr;
                }
            }
        }
        __syncthreads();
    }
    // R (and zeros) back into the panel's place; V in both layouts
    for (int i = 0; i < 16; i++) {
        const int r = rg * 16 + i;
        if (r >= nrows) continue;
        const bool topblock = wg == 0 && r < E2_W;
        const double x = P[r * QR_LD + k];
        a.A[(row0 + r) * a.ld + a.c0 + k] = topblock && k >= r ? x : 0.0;
        const double v = topblock ? (k < r ? x : (k == r ? 1.0 : 0.0)) : x;
        a.Vc[(row0 + r) * a.ld + a.c0 + k] = v;
    }
    __syncthreads();
    // LDS now holds V proper (unit lower trapezoidal)
    if (wg == 0) {
        for (int i = 0; i < 16; i++) {
            const int r = rg * 16 + i;
            if (r < E2_W && r < nrows) {
                const double x = P[r * QR_LD + k];
                P[r * QR_LD + k] = k < r ? x : (k == r ? 1.0 : 0.0);
            }
        }
    }
    __syncthreads();
    for (int q = 0; q < 16; q++) {          // Vt rows: 256 consecutive rows per reflector
        const int kk = q * 4 + (tid >> 8), r = tid & 255;
        if (r < nrows) a.Vt[(a.c0 + kk) * a.ld + row0 + r] = P[r * QR_LD + kk];
    }
    // partial V'V: thread (i, four columns)
    {
        const int i = tid >> 4, kq = (tid & 15) * 4;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        for (int r = 0; r < nrows; r++) {
            const double vi = P[r * QR_LD + i];
            s0 += vi * P[r * QR_LD + kq];
            s1 += vi * P[r * QR_LD + kq + 1];
            s2 += vi * P[r * QR_LD + kq + 2];
            s3 += vi * P[r * QR_LD + kq + 3];
        }
        double* sp = a.spart + (long)wg * 4096 + i * 64 + kq;
        st_sc1(sp, s0); st_sc1(sp + 1, s1); st_sc1(sp + 2, s2); st_sc1(sp + 3, s3);
    }
    if (!grid_barrier(a.bar + 1, (unsigned)nwg, a.abort_flag)) return;
    if (wg != 0) return;
    // T by the dlarft recurrence (forward, columnwise) from S = V'V:  T[l][i] = -tau_i sum_{m = l}^{i-1} T[l][m] S[m][i]
    double* S = P;                           // [64][QR_LD]
    double* T = P + 64 * QR_LD;              // [64][QR_LD]
    __syncthreads();
    for (int e = tid; e < 4096; e += 1024) {
        double s = 0.0;
        for (int g = 0; g < nwg; g++) s += ld_sc1(a.spart + (long)g * 4096 + e);
        S[(e >> 6) * QR_LD + (e & 63)] = s;
        T[(e >> 6) * QR_LD + (e & 63)] = 0.0;
    }
    __syncthreads();
    for (int i = 0; i < E2_W; i++) {
        const double ti = taus[i];
        if (tid < i) {
            double acc = 0.0;
            for (int m = tid; m < i; m++) acc += T[tid * QR_LD + m] * S[m * QR_LD + i];
            T[tid * QR_LD + i] = -ti * acc;     // (column i of T is not read in this step)
        }
        if (tid == i) T[i * QR_LD + i] = ti;
        __syncthreads();
    }
    for (int e = tid; e < 4096; e += 1024) a.T[e] = T[(e >> 6) * QR_LD + (e & 63)];
}

// partial G = V'W over 128 rows:  thread (i, four columns)
__global__ __launch_bounds__(1024) void e2_vtw_kernel(const double* __restrict__ Vc, long ld, long r0, long c0, long dim,
                                                      const double* __restrict__ W, double* __restrict__ gpart) {
    extern __shared__ double sm[];
    double* Vs = sm;                       // [CH_ROWS][QR_LD]
    double* Ws = Vs + CH_ROWS * QR_LD;
    const int tid = threadIdx.x, k = tid & 63, rg = tid >> 6;
    const long row0 = r0 + (long)blockIdx.x * CH_ROWS;
    const int nrows = (int)min((long)CH_ROWS, dim - row0);
    for (int i = 0; i < 8; i++) {
        const int r = rg * 8 + i;
        Vs[r * QR_LD + k] = r < nrows ? Vc[(row0 + r) * ld + c0 + k] : 0.0;
        Ws[r * QR_LD + k] = r < nrows ? W[(row0 - r0 + r) * 64 + k] : 0.0;
    }
    __syncthreads();
    const int i = tid >> 4, kq = (tid & 15) * 4;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int r = 0; r < nrows; r++) {
        const double vi = Vs[r * QR_LD + i];
        s0 += vi * Ws[r * QR_LD + kq];
        s1 += vi * Ws[r * QR_LD + kq + 1];
        s2 += vi * Ws[r * QR_LD + kq + 2];
        s3 += vi * Ws[r * QR_LD + kq + 3];
    }
    double* gp = gpart + (long)blockIdx.x * 4096 + i * 64 + kq;
    gp[0] = s0; gp[1] = s1; gp[2] = s2; gp[3] = s3;
}

// K = T' (sum of the partial G) T   (one workgroup)
__global__ __launch_bounds__(1024) void e2_small_kernel(const double* __restrict__ gpart, int nparts, const double* __restrict__ T,
                                                        double* __restrict__ K) {
    extern __shared__ double sm[];
    double* G = sm;                  // [64][QR_LD]
    double* Ts = G + 64 * QR_LD;
    double* M1 = Ts + 64 * QR_LD;
    const int tid = threadIdx.x;
    for (int e = tid; e < 4096; e += 1024) {
        double s = 0.0;
        for (int p = 0; p < nparts; p++) s += gpart[(long)p * 4096 + e];
        G[(e >> 6) * QR_LD + (e & 63)] = s;
        Ts[(e >> 6) * QR_LD + (e & 63)] = T[e];
    }
    __syncthreads();
    for (int e = tid; e < 4096; e += 1024) {     // M1 = G T
        const int i = e >> 6, j = e & 63;
        double s = 0.0;
        for (int m = 0; m <= j; m++) s += G[i * QR_LD + m] * Ts[m * QR_LD + j];
        M1[i * QR_LD + j] = s;
    }
    __syncthreads();
    for (int e = tid; e < 4096; e += 1024) {     // K = T' M1
        const int i = e >> 6, j = e & 63;
        double s = 0.0;
        for (int m = 0; m <= i; m++) s += Ts[m * QR_LD + i] * M1[m * QR_LD + j];
        K[e] = s;
    }
}

// Z = W T - 1/2 V K over 64 rows; rows of the update's operand panels:  PX = [V'; Z'],  PY = [Z'; V']
constexpr int Z_ROWS = 64;
__global__ __launch_bounds__(512) void e2_z_kernel(const double* __restrict__ Vc, long ld, long r0, long c0, long dim,
                                                   const double* __restrict__ W, const double* __restrict__ T,
                                                   const double* __restrict__ K, double* __restrict__ PX, double* __restrict__ PY) {
    extern __shared__ double sm[];
    double* Vs = sm;                         // [Z_ROWS][QR_LD]
    double* Ws = Vs + Z_ROWS * QR_LD;
    double* Ts = Ws + Z_ROWS * QR_LD;        // [64][QR_LD]
    double* Ks = Ts + 64 * QR_LD;
    const int tid = threadIdx.x, k = tid & 63, rg = tid >> 6;   // 8 row groups of 8 rows
    const long row0 = r0 + (long)blockIdx.x * Z_ROWS;
    const int nrows = (int)min((long)Z_ROWS, dim - row0);
    for (int i = 0; i < 8; i++) {
        const int r = rg * 8 + i;
        Vs[r * QR_LD + k] = r < nrows ? Vc[(row0 + r) * ld + c0 + k] : 0.0;
        Ws[r * QR_LD + k] = r < nrows ? W[(row0 - r0 + r) * 64 + k] : 0.0;
    }
    for (int e = tid; e < 4096; e += 512) {
        Ts[(e >> 6) * QR_LD + (e & 63)] = T[e];
        Ks[(e >> 6) * QR_LD + (e & 63)] = K[e];
    }
    __syncthreads();
    // thread (row r = tid % 64, eight columns): the transposed stores are then contiguous along r
    const int r = tid & 63, jq = (tid >> 6) * 8;
    double z[8];
    for (int q = 0; q < 8; q++) z[q] = 0.0;
    for (int m = 0; m < 64; m++) {
        const double wv = Ws[r * QR_LD + m], vv = -0.5 * Vs[r * QR_LD + m];
        for (int q = 0; q < 8; q++) z[q] += wv * Ts[m * QR_LD + jq + q] + vv * Ks[m * QR_LD + jq + q];
    }
    if (r < nrows) {
        for (int q = 0; q < 8; q++) {
            const int j = jq + q;
            const double v = Vs[r * QR_LD + j];
            PX[(long)j * ld + row0 + r] = v;
            PX[(long)(64 + j) * ld + row0 + r] = z[q];
            PY[(long)j * ld + row0 + r] = z[q];
            PY[(long)(64 + j) * ld + row0 + r] = v;
        }
    }
}

// AB[q][c][o] = w(c + o) w(c) A[c + o][c]  for o <= 64 (zero beyond: room for the chase's fill), w = wa on the first
// E2_W coordinates, wb on the rest
__global__ void e2_band_kernel(const double* __restrict__ A, long ld, long dim, double wa, double wb, double* __restrict__ AB) {
    const long c = blockIdx.x;
    const int o = threadIdx.x;   // 128 threads
    double v = 0.0;
    if (c < dim && o <= E2_W && c + o < dim) {
        const double wr = (c + o) < E2_W ? wa : wb, wc = c < E2_W ? wa : wb;
        v = wr * wc * A[(c + o) * ld + c];
    }
    AB[c * 128 + o] = v;
}

}  // namespace

bool eigh2_serves(long dim, int batch) {
    return !form("eigh_one_stage", 0) && batch >= 1 && dim >= 1024;
}

int eigh2_to_band(crm_ctx* ctx, EighWork& w) {
    hipStream_t st = ctx->stream;
    const long dim = w.dim, ld = w.ld, dimp = w.dimp;
    if (dim <= E2_W + 2) return CRM_OK;
    const int npanels = (int)((dim - 2 - E2_W) / E2_W + 1);     // panels with at least two rows to reflect
    const int maxwg = (int)((dim - E2_W + QR_ROWS - 1) / QR_ROWS);
    const int maxch = (int)((dim - E2_W + CH_ROWS - 1) / CH_ROWS);
    if (maxwg > 64) {
        set_error("two-stage eigen-solver: order %ld beyond the panel kernel's 64 co-resident workgroups", dim);
        return CRM_ERR_UNSUPPORTED;
    }
    // carve the scratch buffer
    size_t nd = 0;
    auto take = [&](size_t count) { size_t at = nd; nd += (count + 31) / 32 * 32; return at; };
    const size_t oT = take((size_t)npanels * 4096), oPart = take((size_t)64 * maxwg * 64), oTop = take(4096),
                 oSpart = take((size_t)maxwg * 4096), oGpart = take((size_t)maxch * 4096), oK = take(4096),
                 oPX = take((size_t)128 * ld + 256), oPY = take((size_t)128 * ld + 256);
    // W = A22' V in slices along the contraction axis
    const long cells0 = dimp - E2_W;
    const int ksplit_max = 16;
    const size_t wsz = (size_t)dimp * 64 + 256;
    const size_t oW = take(wsz * ksplit_max);
    const size_t bytes = sizeof(double) * nd + sizeof(GemmProblem) * 2 * (size_t)npanels + sizeof(unsigned) * 2 * (size_t)npanels + 256;
    CRM_TRY(w.s1.ensure(bytes));
    double* base = w.s1.as<double>();
    GemmProblem* d_probs = reinterpret_cast<GemmProblem*>(base + nd);
    unsigned* bars = reinterpret_cast<unsigned*>(d_probs + 2 * (size_t)npanels);
    CRM_TRY(w.sync.ensure(sizeof(int) * ((size_t)w.batch * w.ld + 64)));
    int* abort_flag = w.sync.as<int>();
    CRM_HIP(hipMemsetAsync(abort_flag, 0, sizeof(int) * 16, st));
    CRM_HIP(hipMemsetAsync(bars, 0, sizeof(unsigned) * 2 * (size_t)npanels, st));
    CRM_HIP(hipMemsetAsync(base + oPX, 0, sizeof(double) * 2 * ((size_t)128 * ld + 256), st));
    CRM_HIP(hipMemsetAsync(w.Vt.ptr, 0, sizeof(double) * w.slab, st));
    CRM_HIP(hipMemsetAsync(w.Vc.ptr, 0, sizeof(double) * w.slab, st));
    CRM_HIP(hipMemsetAsync(w.tau.ptr, 0, sizeof(double) * ld, st));
    double* A = w.A.as<double>();
    double* Vc = w.Vc.as<double>();
    double* Wb = base + oW;
    std::vector<GemmProblem> probs(2 * (size_t)npanels);
    std::vector<int> splits(npanels);
    for (int p = 0; p < npanels; p++) {
        const long c0 = (long)p * E2_W, r0 = c0 + E2_W, m = dim - r0;
        GemmProblem g{};      // W (m x 64) = A22' V
        g.X = A + r0 * ld + r0; g.ldx = ld;
        g.Y = Vc + r0 * ld + c0; g.ldy = ld;
        g.C = Wb; g.ldc = 64;
        g.M = (int)m; g.N = 64;
        probs[2 * (size_t)p] = g;
        {   // slices of whole stages, every one non-empty, at most ksplit_max of them
            const long stages = (dimp - r0) / GEMM_BK;
            const long want = std::min<long>(ksplit_max, split_for(dimp - r0, (m + 127) / 128));
            const long per = (stages + want - 1) / want;
            splits[p] = (int)((stages + per - 1) / per);
        }
        GemmProblem u{};      // A22 -= [V; Z]' [Z; V]
        u.X = base + oPX + r0; u.ldx = ld;
        u.Y = base + oPY + r0; u.ldy = ld;
        u.C = A + r0 * ld + r0; u.ldc = ld;
        u.M = (int)m; u.N = (int)m;
        u.flags = GEMM_SUBTRACT;
        probs[2 * (size_t)p + 1] = u;
    }
    (void)cells0;
    CRM_HIP(hipMemcpyAsync(d_probs, probs.data(), sizeof(GemmProblem) * probs.size(), hipMemcpyHostToDevice, st));
    const size_t lds_qr = sizeof(double) * (QR_ROWS * QR_LD + 16 * 64 + 64 * 4 + 8);
    const size_t lds_vtw = sizeof(double) * (2 * CH_ROWS * QR_LD);
    const size_t lds_small = sizeof(double) * (3 * 64 * QR_LD);
    const size_t lds_z = sizeof(double) * (2 * Z_ROWS * QR_LD + 2 * 64 * QR_LD);
    CRM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&e2_panel_qr_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_qr));
    CRM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&e2_vtw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_vtw));
    CRM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&e2_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_small));
    CRM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&e2_z_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_z));
    for (int p = 0; p < npanels; p++) {
        const long c0 = (long)p * E2_W, r0 = c0 + E2_W, m = dim - r0;
        QrArgs a{};
        a.A = A; a.Vc = Vc; a.Vt = w.Vt.as<double>(); a.tau = w.tau.as<double>();
        a.T = base + oT + (size_t)p * 4096; a.part = base + oPart; a.toprow = base + oTop; a.spart = base + oSpart;
        a.bar = bars + 2 * (size_t)p; a.abort_flag = abort_flag;
        a.ld = ld; a.dim = dim; a.c0 = c0; a.r0 = r0; a.maxwg = maxwg;
        const int nwg = (int)((m + QR_ROWS - 1) / QR_ROWS);
        hipLaunchKernelGGL(e2_panel_qr_kernel, dim3(nwg), dim3(1024), lds_qr, st, a);
        const int ks = splits[p];
        CRM_TRY(launch_gemm_tn(ctx, d_probs + 2 * (size_t)p, 1, (int)m, 64, dimp - r0, false, 0, ks, (long)wsz));
        CRM_TRY(launch_reduce_splits(st, Wb, m * 64, ks, (long)wsz));
        const int nch = (int)((m + CH_ROWS - 1) / CH_ROWS);
        hipLaunchKernelGGL(e2_vtw_kernel, dim3(nch), dim3(1024), lds_vtw, st, Vc, ld, r0, c0, dim, Wb, base + oGpart);
        hipLaunchKernelGGL(e2_small_kernel, dim3(1), dim3(1024), lds_small, st, base + oGpart, nch, a.T, base + oK);
        hipLaunchKernelGGL(e2_z_kernel, dim3((unsigned)((m + Z_ROWS - 1) / Z_ROWS)), dim3(512), lds_z, st, Vc, ld, r0, c0, dim, Wb,
                           a.T, base + oK, base + oPX, base + oPY);
        CRM_HIP(hipGetLastError());
        CRM_TRY(launch_gemm_tn(ctx, d_probs + 2 * (size_t)p + 1, 1, (int)m, (int)m, 128, false, 0, 1, 0));
    }
    int aborted = 0;
    CRM_HIP(hipMemcpyAsync(&aborted, abort_flag, sizeof(int), hipMemcpyDeviceToHost, st));
    CRM_HIP(hipStreamSynchronize(st));
    if (aborted) {
        set_error("two-stage eigen-solver: the panel kernel's workgroups were not co-resident (a shared device?)");
        return CRM_ERR_UNSUPPORTED;
    }
    return CRM_OK;
}

int eigh2_scale_band(crm_ctx* ctx, EighWork& w, const double* wa, const double* wb) {
    hipStream_t st = ctx->stream;
    const size_t ab_slab = (size_t)(w.dimp + 128) * 128;
    CRM_TRY(w.AB.ensure(sizeof(double) * ab_slab * w.batch));
    for (int q = 0; q < w.batch; q++)
        hipLaunchKernelGGL(e2_band_kernel, dim3((unsigned)(w.dimp + 128)), dim3(128), 0, st, w.A.as<double>(), w.ld, w.dim, wa[q],
                           wb[q], w.AB.as<double>() + (size_t)q * ab_slab);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

}  // namespace crm
