// Stage 2 of the two-stage eigen-solver of the background constructor: band -> tridiagonal by bulge chasing, per grid
// point, and the back-transformation of the tridiagonal eigenvectors through the chase's reflectors.
// (Stage 1, the family argument and the reference lines this replaces: eigh2_band.hip.  numpy statement of both:
// tools/eigh2_prototype.py -- chase(), back2().)
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
// Back-transformation: Q2 = prod_s prod_k H(s, k).  H(s + 1, k) overlaps only H(s, k) and H(s, k + 1), so the product can
//   be regrouped into blocks of 64 consecutive sweeps at one chain position -- compact-WY blocks I - V T V' over windows
//   of 127 rows -- applied to the eigenvector rows sweep blocks last to first, chain positions ascending
//   (tools/eigh2_prototype.py: back2).  One workgroup keeps 48 eigenvectors' window in LDS and runs three small products
//   per block on the FP64 matrix pipe (v_mfma_f64_16x16x4_f64), skipping the parallelogram's zero k-steps.
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
                while ((val = __hip_atomic_load(prog + s - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < need) {
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 4095) == 0) {
                        if (spins > (1L << 26) || __hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
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

// T of the group (sweep block S, chain position k): reflector j is v(S 64 + j, k) placed at rows j .. j + 63 of the
// group's window of 127 rows.  S = V'V over the window (V kept in window coordinates in LDS: lanes run along a row), then
// the dlarft recurrence.
__global__ __launch_bounds__(256) void e2_group_larft_kernel(const double* __restrict__ V, long v_slab, const double* __restrict__ tau,
                                                             long tau_slab, long dimp, long n, int npos, double* __restrict__ Tout,
                                                             long t_slab) {
    extern __shared__ double lsm[];
    constexpr int LW = W + 1;
    double* Vw = lsm;                  // [128][LW]  Vw[c][j] = v_j[c - j]
    double* Ss = Vw + 128 * LW;        // [64][LW]
    double* Ts = Ss + W * LW;          // [64][LW]
    double* taus = Ts + W * LW;        // [64]
    const int k = blockIdx.x, S = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    if ((long)S * W + 1 + (long)k * W >= n) return;      // no reflector of this group exists
    const double* Vg = V + (size_t)b * v_slab + ((size_t)k * dimp + (size_t)S * W) * W;
    for (int e = tid; e < 128 * LW; e += 256) Vw[e] = 0.0;
    if (tid < W) taus[tid] = tau[(size_t)b * tau_slab + (size_t)k * dimp + (size_t)S * W + tid];
    __syncthreads();
    for (int e = tid; e < W * W; e += 256) {
        const int j = e >> 6, o = e & 63;
        Vw[(j + o) * LW + j] = Vg[e];
    }
    __syncthreads();
    for (int e = tid; e < W * W; e += 256) {
        const int j1 = e >> 6, j2 = e & 63;          // (a wavefront shares j1)
        const int lo = j1 > j2 ? j1 : j2, hi = (j1 < j2 ? j1 : j2) + W;   // rows where both reflectors live
        double s = 0.0;
        for (int c = lo; c < hi; c++) s += Vw[c * LW + j1] * Vw[c * LW + j2];
        Ss[j1 * LW + j2] = s;
        Ts[j1 * LW + j2] = 0.0;
    }
    __syncthreads();
    for (int i = 0; i < W; i++) {
        const double ti = taus[i];
        if (tid < i) {
            double acc = 0.0;
            for (int m = tid; m < i; m++) acc += Ts[tid * LW + m] * Ss[m * LW + i];
            Ts[tid * LW + i] = -ti * acc;
        }
        if (tid == i) Ts[i * LW + i] = ti;
        __syncthreads();
    }
    double* Tg = Tout + (size_t)b * t_slab + ((size_t)S * npos + k) * W * W;
    for (int e = tid; e < W * W; e += 256) Tg[e] = Ts[(e >> 6) * LW + (e & 63)];
}

// ---- back-transformation through the chase's reflectors --------------------------------------------------------------
constexpr int BT_ROWS = 48;     // eigenvectors per workgroup
constexpr int ZS_LD = 132;      // window of 128 coordinates
constexpr int VC_G = 18;        // zero guard on both sides of a reflector's 64 entries: the products read V[c][j] = Vc[j][c - j]
constexpr int VC_LD = 101;      // for every c of a k-step, inside the band or not, without a select
constexpr int TS_LD = 68;
constexpr int WS_LD = 68;

// The three products of one group on the matrix pipe, for wavefront WV of the workgroup.  FP64 MFMAs and VALU
// instructions do not overlap on this chip, so the loops carry no address arithmetic: every LDS address is a base
// formed once per group plus a compile-time offset (hence the wavefront as a template parameter), and the parallelogram's
// zeros come from guard bands in LDS instead of selects.
//   W1[e][j]  = sum_c Z[e][c] V[c][j]            wave -> j in [16 WV, 16 WV + 16), k-steps c in [16 WV, 16 WV + 80)
//   W2[e][j'] = sum_j W1[e][j] T[j'][j]          wave -> j' tile WV, k-steps j >= 16 WV (T upper triangular)
//   Z[e][c]  -= sum_j W2[e][j] V[c][j]           wave -> two column tiles of c (twenty k-steps together)
template <int WV>
__device__ __forceinline__ void bt2_products(double* __restrict__ Zs, const double* __restrict__ Vc, const double* __restrict__ Ts,
                                             double* __restrict__ Ws, int par, int l15, int lq) {
    v4d acc[3];
    double* zh[2];      // logical halves of the window
    zh[0] = Zs + (par ? 64 : 0);
    zh[1] = Zs + (par ? 0 : 64);
    {
        const double* bB = Vc + (16 * WV + l15) * VC_LD + VC_G + lq - l15;
        const double* aA0 = zh[0] + l15 * ZS_LD + lq;
        const double* aA1 = zh[1] + l15 * ZS_LD + lq;
#pragma unroll
        for (int m = 0; m < 3; m++) acc[m] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < 20; i++) {
            const int c0 = 16 * WV + 4 * i;
            const double* aA = (c0 >> 6) ? aA1 : aA0;
            const double bv = bB[4 * i];
#pragma unroll
            for (int m = 0; m < 3; m++)
                acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(aA[16 * m * ZS_LD + (c0 & 63)], bv, acc[m], 0, 0, 0);
        }
        double* w = Ws + lq * WS_LD + 16 * WV + l15;
#pragma unroll
        for (int m = 0; m < 3; m++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) w[(16 * m + 4 * reg) * WS_LD] = acc[m][reg];
    }
    lds_barrier();
    {
        const double* bT = Ts + (16 * WV + l15) * TS_LD + 16 * WV + lq;
        const double* aW = Ws + l15 * WS_LD + 16 * WV + lq;
#pragma unroll
        for (int m = 0; m < 3; m++) acc[m] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < 16 - 4 * WV; i++) {
            const double bv = bT[4 * i];
#pragma unroll
            for (int m = 0; m < 3; m++) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(aW[16 * m * WS_LD + 4 * i], bv, acc[m], 0, 0, 0);
        }
    }
    lds_barrier();
    {
        double* w = Ws + lq * WS_LD + 16 * WV + l15;
#pragma unroll
        for (int m = 0; m < 3; m++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) w[(16 * m + 4 * reg) * WS_LD] = acc[m][reg];
    }
    lds_barrier();
#pragma unroll
    for (int half = 0; half < 2; half++) {
        constexpr int NTA = WV == 0 ? 0 : WV == 1 ? 1 : WV == 2 ? 4 : 5;
        constexpr int NTB = WV == 0 ? 3 : WV == 1 ? 2 : WV == 2 ? 7 : 6;
        const int nt = half == 0 ? NTA : NTB;
        const int jlo = 16 * nt - 63 > 0 ? 16 * nt - 63 : 0, jhi = 16 * nt + 15 < 63 ? 16 * nt + 15 : 63;
        const int ks0 = jlo >> 2, steps = (jhi >> 2) - ks0 + 1;
        const double* bB = Vc + (4 * ks0 + lq) * VC_LD + VC_G + 16 * nt + l15 - 4 * ks0 - lq;
        const double* aW = Ws + l15 * WS_LD + 4 * ks0 + lq;
#pragma unroll
        for (int m = 0; m < 3; m++) acc[m] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (i < steps) {
                const double bv = bB[i * (4 * VC_LD - 4)];
#pragma unroll
                for (int m = 0; m < 3; m++) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(aW[16 * m * WS_LD + 4 * i], bv, acc[m], 0, 0, 0);
            }
        }
        double* z = zh[(16 * nt) >> 6] + lq * ZS_LD + ((16 * nt) & 63) + l15;
#pragma unroll
        for (int m = 0; m < 3; m++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) z[(16 * m + 4 * reg) * ZS_LD] -= acc[m][reg];
    }
    lds_barrier();
}

struct Bt2Args {
    double* Qt; long slab, ld;          // rows = eigenvectors
    const double* V; long v_slab;
    const double* T; long t_slab;
    long n, dimp;
    int npos, nS, tasks_per_matrix;
};

// grid: (tasks_per_matrix * batch), 256 threads.  Per sweep block the 128-column window slides down the eigenvectors 64
// columns per group: the half that leaves is stored, the half that stays keeps its place in LDS (the halves swap roles by
// parity), and the next group's reflectors, T and 64 new columns are fetched into registers while this group's products
// run on the matrix pipe.
__global__ __launch_bounds__(256) void e2_bt2_kernel(Bt2Args a) {
    extern __shared__ double sm[];
    double* Zs = sm;                          // [48][ZS_LD]  two halves of 64 columns
    double* Vc = Zs + BT_ROWS * ZS_LD;        // [64][VC_LD]   Vc[j][o] = v_j[o]
    double* Ts = Vc + W * VC_LD;              // [64][TS_LD]
    double* Ws = Ts + W * TS_LD;              // [48][WS_LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
    const int b = blockIdx.x / a.tasks_per_matrix, task = blockIdx.x % a.tasks_per_matrix;
    const long e0 = (long)task * BT_ROWS;
    const int ne = (int)min((long)BT_ROWS, a.n - e0);
    double* Q = a.Qt + (size_t)b * a.slab + (size_t)e0 * a.ld;
    const double* Vb = a.V + (size_t)b * a.v_slab;
    const double* Tb = a.T + (size_t)b * a.t_slab;
    const long n = a.n;
    auto phys = [](int c, int par) { return (c & 63) | ((((c >> 6) ^ par) & 1) << 6); };
    for (int e = tid; e < W * VC_LD; e += 256) Vc[e] = 0.0;     // (the guard bands stay zero)
    for (int S = a.nS - 1; S >= 0; S--) {
        const long s0 = (long)S * W, c00 = s0 + 1;
        if (c00 >= n) continue;
        const int kc = (int)((n - c00 + W - 1) / W);      // groups with a window inside the vectors
        // ---- the first window of the block, its reflectors and T ------------------------------------------------------------
        __syncthreads();
        for (int e = tid; e < BT_ROWS * 128; e += 256) {
            const int r = e >> 7, c = e & 127;
            Zs[r * ZS_LD + c] = (r < ne && c00 + c < n) ? Q[(size_t)r * a.ld + c00 + c] : 0.0;
        }
        {
            const v2d* Vg = reinterpret_cast<const v2d*>(Vb + (size_t)s0 * W);
            const v2d* Tg = reinterpret_cast<const v2d*>(Tb + (size_t)S * a.npos * W * W);
            for (int q = 0; q < 8; q++) {
                const int e2 = (tid + 256 * q) * 2;
                const v2d x = Vg[tid + 256 * q], y = Tg[tid + 256 * q];
                Vc[(e2 >> 6) * VC_LD + VC_G + (e2 & 63)] = x[0]; Vc[(e2 >> 6) * VC_LD + VC_G + (e2 & 63) + 1] = x[1];
                Ts[(e2 >> 6) * TS_LD + (e2 & 63)] = y[0]; Ts[(e2 >> 6) * TS_LD + (e2 & 63) + 1] = y[1];
            }
        }
        __syncthreads();
        for (int k = 0; k < kc; k++) {
            const long c0 = c00 + (long)k * W;
            const int par = k & 1;
            const bool more = k + 1 < kc;
            // ---- (i) the next group's data on its way into registers ------------------------------------------------------
            v2d pv[8], pt[8];
            double pz[12];
            if (more) {
                const v2d* Vg = reinterpret_cast<const v2d*>(Vb + ((size_t)(k + 1) * a.dimp + (size_t)s0) * W);
                const v2d* Tg = reinterpret_cast<const v2d*>(Tb + ((size_t)S * a.npos + k + 1) * W * W);
#pragma unroll
                for (int q = 0; q < 8; q++) { pv[q] = Vg[tid + 256 * q]; pt[q] = Tg[tid + 256 * q]; }
#pragma unroll
                for (int q = 0; q < 12; q++) {
                    const int e = tid + 256 * q, r = e >> 6, cc = e & 63;
                    pz[q] = (r < ne && c0 + 128 + cc < n) ? Q[(size_t)r * a.ld + c0 + 128 + cc] : 0.0;
                }
            }
            // ---- (ii) the three products ------------------------------------------------------------------------------------
            switch (wave) {
                case 0: bt2_products<0>(Zs, Vc, Ts, Ws, par, l15, lq); break;
                case 1: bt2_products<1>(Zs, Vc, Ts, Ws, par, l15, lq); break;
                case 2: bt2_products<2>(Zs, Vc, Ts, Ws, par, l15, lq); break;
                default: bt2_products<3>(Zs, Vc, Ts, Ws, par, l15, lq); break;
            }
            // ---- (iii) the half that leaves the window goes home; (iv) its place takes the columns that enter ---------------
#pragma unroll
            for (int q = 0; q < 12; q++) {
                const int e = tid + 256 * q, r = e >> 6, cc = e & 63;
                double* z = Zs + r * ZS_LD + phys(cc, par);
                if (r < ne && c0 + cc < n) Q[(size_t)r * a.ld + c0 + cc] = *z;
                if (more) *z = pz[q];
                else if (r < ne && c0 + 64 + cc < n) Q[(size_t)r * a.ld + c0 + 64 + cc] = Zs[r * ZS_LD + phys(64 + cc, par)];
            }
            if (more) {
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int e2 = (tid + 256 * q) * 2;
                    Vc[(e2 >> 6) * VC_LD + VC_G + (e2 & 63)] = pv[q][0]; Vc[(e2 >> 6) * VC_LD + VC_G + (e2 & 63) + 1] = pv[q][1];
                    Ts[(e2 >> 6) * TS_LD + (e2 & 63)] = pt[q][0]; Ts[(e2 >> 6) * TS_LD + (e2 & 63) + 1] = pt[q][1];
                }
            }
            lds_barrier();
        }
    }
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

int eigh2_back_chase(crm_ctx* ctx, EighWork& w, double* Qt) {
    hipStream_t st = ctx->stream;
    const long n = w.dim, dimp = w.dimp;
    if (n <= 2) return CRM_OK;
    const int B = w.batch, npos = chase_positions(n);
    const int nS = (int)((n - 2 + W - 1) / W);
    const long t_slab = (long)nS * npos * W * W;
    CRM_TRY(w.Tbc.ensure(sizeof(double) * (size_t)t_slab * B));
    const size_t lds_t = sizeof(double) * (4 * W * (W + 1) + W);
    CRM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&e2_group_larft_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t));
    hipLaunchKernelGGL(e2_group_larft_kernel, dim3(npos, nS, B), dim3(256), lds_t, st, w.Vbc.as<double>(), (long)npos * dimp * W,
                       w.taubc.as<double>(), (long)npos * dimp, dimp, n, npos, w.Tbc.as<double>(), t_slab);
    Bt2Args a{};
    a.Qt = Qt; a.slab = w.slab; a.ld = w.ld;
    a.V = w.Vbc.as<double>(); a.v_slab = (long)npos * dimp * W;
    a.T = w.Tbc.as<double>(); a.t_slab = t_slab;
    a.n = n; a.dimp = dimp; a.npos = npos; a.nS = nS;
    a.tasks_per_matrix = (int)((n + BT_ROWS - 1) / BT_ROWS);
    const size_t lds = sizeof(double) * (BT_ROWS * ZS_LD + W * VC_LD + W * TS_LD + BT_ROWS * WS_LD);
    CRM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&e2_bt2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(e2_bt2_kernel, dim3((unsigned)(a.tasks_per_matrix * B)), dim3(256), lds, st, a);
    CRM_HIP(hipGetLastError());
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
        CRM_TRY(eigh2_back_chase(ctx, w, Qt));
    }
    lap("back-transformation 2");
    {
        TraceRange r("crm eigh2 back-transformation (band)");
        w.v_shared = true;
        const int rc = eigh_back_transform(ctx, w, Qt, Zt);
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
