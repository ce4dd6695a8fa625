// Stage 2 of the two-stage eigen-solver, second half: the back-transformation of the tridiagonal eigenvectors through
// the chase's reflectors (eigh2_chase.hip; numpy statement: tools/eigh2_prototype.py -- back2()).
//
// Q2 = prod_s prod_k H(s, k).  H(s + 1, k) overlaps only H(s, k) and H(s, k + 1), so the product regroups into blocks of 64
// consecutive sweeps at one chain position -- compact-WY blocks I - V T V' over windows of 127 coordinates -- applied to
// the eigenvectors sweep blocks last to first, chain positions ascending.
//
// e2_back_kernel keeps the eigenvectors' window in REGISTERS.  Z is stored with column e = eigenvector e; a wavefront owns
// sixteen eigenvectors and the window of 128 coordinates as eight 16 x 16 accumulator tiles of v_mfma_f64_16x16x4_f64.  An
// accumulator tile holds row 4 reg + (lane / 16), column lane % 16 -- which is exactly the B operand of k-step `reg` of a
// product that contracts over the tile's rows.  So the three products of a group chain through registers:
//     W1 = V' Z      A = V' from LDS,  B = the window's tiles          (80 MFMAs: 4 tiles of W1, 20 k-steps each)
//     W2 = -T W1     A = -T from LDS,  B = W1's tiles                  (40: T is upper triangular)
//     Z += V W2      A = V from LDS,   B = W2's tiles, C = the window  (80: the parallelogram's zero k-steps skipped)
// with no LDS round trip for Z or W and no barrier inside a group.  LDS holds only what the wavefronts of a workgroup
// share -- the group's 64 reflectors (zero guard bands instead of selects: every address is a per-lane base plus a
// compile-time offset, FP64 MFMAs and VALU instructions do not overlap on this chip) and -T packed by tile rows -- twice,
// so the next group's copy is written while this group's products run: one workgroup barrier per group.  After a group
// the window slides 64 coordinates: the upper four tiles are stored, the lower four take their place (register renaming
// by parity) and four new tiles arrive from a prefetch issued a group earlier.  Eight wavefronts per workgroup, one
// workgroup per CU: two wavefronts per SIMD cover each other's operand latencies.
#include <algorithm>
#include <vector>

#include "eigh.h"

namespace crm {
namespace {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
constexpr int W = E2_W;            // 64

// barrier of a workgroup for its LDS traffic only: global loads and stores in flight stay in flight
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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

// ---- the back-transformation ---------------------------------------------------------------------------------------------
constexpr int R_WAVES = 8;
constexpr int R_VG = 18;            // zero guard on both sides of a reflector's 64 entries: the products read V[c][j] =
constexpr int R_VLD = 101;          // Vs[j][c - j] for every c of a k-step, inside the band or not, without a select
constexpr int R_VSZ = W * R_VLD;
// -T by tile rows: rows 16 t .. 16 t + 15 keep their columns from 16 t on (upper triangular), row stride 68 - 16 t
__host__ __device__ constexpr int r_tld(int t) { return 68 - 16 * t; }
__host__ __device__ constexpr int r_toff(int t) { return 1088 * t - 128 * t * (t - 1); }
constexpr int R_TSZ = r_toff(4);
constexpr int R_BUF = R_VSZ + R_TSZ;   // doubles per copy (9 280: two copies are 145 KB of LDS)

__device__ __forceinline__ double z_load(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)byte_off, 0, 0));
}
__device__ __forceinline__ void z_store(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, double v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, v), rs, (int)byte_off, 0, 0);
}

struct BackArgs {
    double* Z; long slab, ld;           // column e = eigenvector e
    const double* V; long v_slab;
    const double* T; long t_slab;
    long n, dimp;
    int npos, nS;
    const int* tasks;                   // per workgroup: matrix, first tile of sixteen eigenvectors, tiles (<= R_WAVES)
};

// One group on the window lo (coordinates c0 .. c0 + 63) / hi (c0 + 64 .. c0 + 127).  `more`: the sweep block goes on --
// lo is stored and refilled with the coordinates c0 + 128 .. c0 + 191 (it is the next group's hi); otherwise both halves
// are stored.  Vn / Tn: the next group's reflectors and T in global memory (of the next sweep block if this one ends),
// copied into `nxt` between the products; null after the very last group.
__device__ __forceinline__ void r_group(v4d (&lo)[4], v4d (&hi)[4], const double* __restrict__ cur, double* __restrict__ nxt,
                                        const v2d* __restrict__ Vn, const v2d* __restrict__ Tn, __amdgpu_buffer_rsrc_t rs,
                                        unsigned zoff, unsigned row_bytes, long c0, bool more, bool active, int tid, int l15,
                                        int lq) {
    v2d pv[4], pt[4];
    v4d pz[4];
    if (Vn) {
#pragma unroll
        for (int q = 0; q < 4; q++) { pv[q] = Vn[tid + 512 * q]; pt[q] = Tn[tid + 512 * q]; }
    }
    if (more && active) {
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int i = 0; i < 4; i++) pz[t][i] = z_load(rs, zoff + (unsigned)(c0 + 128 + 16 * t + 4 * i) * row_bytes);
    }
    v4d w1[4], w2[4];
    if (active) {
        // W1[j][e] = sum_c V[c][j] Z[c][e]:  A[m = j][k = c] = Vs[j][c - j], c = 16 jt + 4 i + lq
        const double* a1 = cur + l15 * (R_VLD - 1) + lq + R_VG;
        const v4d zero = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < 20; i++)
#pragma unroll
            for (int jt = 0; jt < 4; jt++) {
                const int ks = 4 * jt + i, tile = ks >> 2, reg = ks & 3;
                const double b = tile < 4 ? lo[tile & 3][reg] : hi[tile & 3][reg];
                w1[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[jt * 16 * R_VLD + 4 * i], b, i == 0 ? zero : w1[jt], 0, 0, 0);
            }
    }
    if (Vn) {
        double* tn = nxt + R_VSZ;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int e2 = (tid + 512 * q) * 2, r = e2 >> 6, c = e2 & 63, tr = r >> 4;
            double* v = nxt + r * R_VLD + R_VG + c;
            v[0] = pv[q][0]; v[1] = pv[q][1];
            if (c >= 16 * tr) {
                double* t = tn + (1088 * tr - 128 * tr * (tr - 1)) + (r & 15) * (68 - 16 * tr) + (c - 16 * tr);
                t[0] = -pt[q][0]; t[1] = -pt[q][1];
            }
        }
    }
    if (active) {
        // W2[j'][e] = sum_j (-T[j'][j]) W1[j][e]:  A[m = j'][k = j], j = 4 ks + lq >= 16 jt'
        const double* tc = cur + R_VSZ + lq;
        const double* a2[4];
#pragma unroll
        for (int jt = 0; jt < 4; jt++) a2[jt] = tc + r_toff(jt) + l15 * r_tld(jt);
        const v4d zero = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 16; ks++)
#pragma unroll
            for (int jt = 0; jt < 4; jt++)
                if (ks >= 4 * jt)
                    w2[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[jt][4 * ks - 16 * jt], w1[ks >> 2][ks & 3],
                                                                ks == 4 * jt ? zero : w2[jt], 0, 0, 0);
        // Z[c][e] += sum_j V[c][j] W2[j][e]:  A[m = c][k = j] = Vs[j][c - j], c = 16 ct + l15, j = 4 ks + lq
        const double* a3 = cur + lq * (R_VLD - 1) + l15 + R_VG;
#pragma unroll
        for (int ks = 0; ks < 16; ks++)
#pragma unroll
            for (int ct = 0; ct < 8; ct++) {
                const int jlo = 16 * ct - 63 > 0 ? 16 * ct - 63 : 0, jhi = 16 * ct + 15 < 63 ? 16 * ct + 15 : 63;
                if (ks >= (jlo >> 2) && ks <= (jhi >> 2)) {
                    const double a = a3[4 * ks * (R_VLD - 1) + 16 * ct];
                    if (ct < 4) lo[ct & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, w2[ks >> 2][ks & 3], lo[ct & 3], 0, 0, 0);
                    else hi[ct & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, w2[ks >> 2][ks & 3], hi[ct & 3], 0, 0, 0);
                }
            }
        // the half that leaves the window goes home (rows past the end fall outside the buffer's range: dropped)
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int i = 0; i < 4; i++) z_store(rs, zoff + (unsigned)(c0 + 16 * t + 4 * i) * row_bytes, lo[t][i]);
        if (more) {
#pragma unroll
            for (int t = 0; t < 4; t++) lo[t] = pz[t];
        } else {
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int i = 0; i < 4; i++) z_store(rs, zoff + (unsigned)(c0 + 64 + 16 * t + 4 * i) * row_bytes, hi[t][i]);
        }
    }
    lds_barrier();
}

// grid: one workgroup per task, 512 threads, 2 * R_BUF doubles of dynamic LDS
__global__ __launch_bounds__(512) void e2_back_kernel(BackArgs a) {
    extern __shared__ double sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
    const int b = a.tasks[3 * blockIdx.x], tile0 = a.tasks[3 * blockIdx.x + 1], tiles = a.tasks[3 * blockIdx.x + 2];
    const long n = a.n;
    const long e0 = 16L * (tile0 + (wave < tiles ? wave : 0));
    const bool active = wave < tiles;                 // (a wavefront without a tile only helps with the copies)
    // rows < n of this matrix: loads past the end return zero, stores there are dropped
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(a.Z + (size_t)b * a.slab, 0, (int)(unsigned)((size_t)n * a.ld * 8), 0x00020000);
    const unsigned row_bytes = (unsigned)(a.ld * 8);
    const unsigned zoff = (unsigned)(((long)lq * a.ld + e0 + l15) * 8);
    const double* Vb = a.V + (size_t)b * a.v_slab;
    const double* Tb = a.T + (size_t)b * a.t_slab;
    auto v_of = [&](long S, long k) { return reinterpret_cast<const v2d*>(Vb + ((size_t)k * a.dimp + (size_t)S * W) * W); };
    auto t_of = [&](long S, long k) { return reinterpret_cast<const v2d*>(Tb + ((size_t)S * a.npos + k) * W * W); };
    for (int e = tid; e < 2 * R_BUF; e += 512) sm[e] = 0.0;      // (the guard bands stay zero)
    int S = a.nS - 1;
    while (S >= 0 && (long)S * W + 1 >= n) S--;
    if (S < 0) return;
    __syncthreads();
    {   // the first group's reflectors and T
        const v2d* Vg = v_of(S, 0);
        const v2d* Tg = t_of(S, 0);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int e2 = (tid + 512 * q) * 2, r = e2 >> 6, c = e2 & 63, tr = r >> 4;
            const v2d x = Vg[tid + 512 * q], y = Tg[tid + 512 * q];
            sm[r * R_VLD + R_VG + c] = x[0]; sm[r * R_VLD + R_VG + c + 1] = x[1];
            if (c >= 16 * tr) {
                double* t = sm + R_VSZ + (1088 * tr - 128 * tr * (tr - 1)) + (r & 15) * (68 - 16 * tr) + (c - 16 * tr);
                t[0] = -y[0]; t[1] = -y[1];
            }
        }
    }
    __syncthreads();
    int g = 0;
    v4d lo[4], hi[4];
    for (; S >= 0; S--) {
        const long c00 = (long)S * W + 1;
        const int kc = (int)((n - c00 + W - 1) / W);      // groups with a window inside the vectors
        // this wavefront's own stores of the block before must have landed before their rows are read again
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (active) {
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    lo[t][i] = z_load(rs, zoff + (unsigned)(c00 + 16 * t + 4 * i) * row_bytes);
                    hi[t][i] = z_load(rs, zoff + (unsigned)(c00 + 64 + 16 * t + 4 * i) * row_bytes);
                }
        }
        for (int k = 0; k < kc; k++, g++) {
            const bool more = k + 1 < kc;
            const v2d* Vn = more ? v_of(S, k + 1) : S > 0 ? v_of(S - 1, 0) : nullptr;
            const v2d* Tn = more ? t_of(S, k + 1) : S > 0 ? t_of(S - 1, 0) : nullptr;
            const double* cur = sm + (g & 1) * R_BUF;
            double* nxt = sm + ((g + 1) & 1) * R_BUF;
            const long c0 = c00 + (long)k * W;
            if (k & 1) r_group(hi, lo, cur, nxt, Vn, Tn, rs, zoff, row_bytes, c0, more, active, tid, l15, lq);
            else r_group(lo, hi, cur, nxt, Vn, Tn, rs, zoff, row_bytes, c0, more, active, tid, l15, lq);
        }
    }
}

}  // namespace

static inline int chase_positions(long n) { return (int)((n - 1 + W - 1) / W); }   // chain positions 0 .. npos - 1

// Tasks of e2_back_kernel: (matrix, first tile of sixteen eigenvectors, tiles <= R_WAVES) per workgroup.  A wavefront's tile
// costs the same wherever it is, and a CU runs one workgroup at a time, two wavefronts per SIMD: whole rounds of
// eight-tile workgroups first; what is left goes out as four-tile workgroups (one wavefront per SIMD: half the time of a
// round) when a single round of those takes it.
static std::vector<int> back_tasks(int B, long n, long cus) {
    const long nt = (n + 15) / 16;
    const long full = (long)B * nt / (R_WAVES * cus) * cus;          // eight-tile workgroups of the whole rounds
    std::vector<int> tasks, tail;
    auto put = [](std::vector<int>& v, int b, long t0, long count) { v.push_back(b); v.push_back((int)t0); v.push_back((int)count); };
    for (int b = 0; b < B; b++) {
        const long mine = std::min(nt / R_WAVES, full / B + (b < full % B ? 1 : 0));
        for (long q = 0; q < mine; q++) put(tasks, b, q * R_WAVES, R_WAVES);
        for (long t0 = mine * R_WAVES; t0 < nt; t0 += 4) put(tail, b, t0, std::min(4L, nt - t0));
    }
    if ((long)tail.size() / 3 > cus) {    // too many for one short round: eight-tile workgroups throughout
        tasks.clear();
        tail.clear();
        for (int b = 0; b < B; b++)
            for (long t0 = 0; t0 < nt; t0 += R_WAVES) put(tasks, b, t0, std::min<long>(R_WAVES, nt - t0));
    }
    tasks.insert(tasks.end(), tail.begin(), tail.end());
    return tasks;
}

// Z: [batch] x slab, column e (of matrix b) = eigenvector e of the tridiagonal; Z <- Q2 Z in place.
int eigh2_back_chase(crm_ctx* ctx, EighWork& w, double* Z) {
    hipStream_t st = ctx->stream;
    const long n = w.dim, dimp = w.dimp;
    if (n <= 2) return CRM_OK;
    const int B = w.batch, npos = chase_positions(n);
    const int nS = (int)((n - 2 + W - 1) / W);
    const long t_slab = (long)nS * npos * W * W;
    hipDeviceProp_t prop;
    CRM_HIP(hipGetDeviceProperties(&prop, ctx->device));
    const std::vector<int> tasks = back_tasks(B, n, prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256);
    const size_t t_bytes = sizeof(double) * (size_t)t_slab * B;
    CRM_TRY(w.Tbc.ensure(t_bytes + sizeof(int) * tasks.size()));
    int* d_tasks = reinterpret_cast<int*>(w.Tbc.as<char>() + t_bytes);
    CRM_HIP(hipMemcpyAsync(d_tasks, tasks.data(), sizeof(int) * tasks.size(), hipMemcpyHostToDevice, st));
    const size_t lds_t = sizeof(double) * (4 * W * (W + 1) + W);
    CRM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&e2_group_larft_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t));
    hipLaunchKernelGGL(e2_group_larft_kernel, dim3(npos, nS, B), dim3(256), lds_t, st, w.Vbc.as<double>(), (long)npos * dimp * W,
                       w.taubc.as<double>(), (long)npos * dimp, dimp, n, npos, w.Tbc.as<double>(), t_slab);
    BackArgs a{};
    a.Z = Z; a.slab = w.slab; a.ld = w.ld;
    a.V = w.Vbc.as<double>(); a.v_slab = (long)npos * dimp * W;
    a.T = w.Tbc.as<double>(); a.t_slab = t_slab;
    a.n = n; a.dimp = dimp; a.npos = npos; a.nS = nS;
    a.tasks = d_tasks;
    const size_t lds = sizeof(double) * 2 * R_BUF;
    CRM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&e2_back_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(e2_back_kernel, dim3((unsigned)(tasks.size() / 3)), dim3(512), lds, st, a);
    CRM_HIP(hipGetLastError());
    CRM_HIP(hipStreamSynchronize(st));     // (the task table is a host vector)
    return CRM_OK;
}

}  // namespace crm

// ---- test hook (host only: no GPU touched): the task table of the back-transformation ------------------------------------
extern "C" int crm_test_back_tasks(int batch, long dim, int cus, int* tasks, int capacity, int* count) {
    return crm::guarded("crm_test_back_tasks", [&]() -> int {
    if (batch < 1 || dim < 1 || cus < 1 || !tasks || !count) return CRM_ERR_ARG;
    const std::vector<int> t = crm::back_tasks(batch, dim, cus);
    *count = (int)(t.size() / 3);
    if ((int)t.size() > capacity) return CRM_ERR_ARG;
    std::copy(t.begin(), t.end(), tasks);
    return CRM_OK;
    });
}

