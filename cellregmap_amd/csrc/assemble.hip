// Score statistic Q and mixture matrix F per variant.
//
// Reference: QSCov / PMat / ScoreStatistic, cellregmap/_math.py:40-128, called at
// cellregmap/_cellregmap.py:379-435 with K0 = v0 * Q0 S0 Q0' + v1 * I, X = [W, g],
// half_dK = diag(gtest) E0.  The reference applies K0^-1 to n-vectors with three products
// against Q0; here every bilinear form is taken in the rotated space instead,
//
//     u' K0^-1 v = ( u'v - sum_j d_j (Q0'u)_j (Q0'v)_j ) / v1 ,   d_j = v0 S0_j / (v0 S0_j + v1)
//
// so that with A~ = Q0' diag(gtest) E0 (from the Khatri-Rao contraction) and the n-length
// reductions Z1..Z3 nothing of size n is touched here.  Two kernels:
//   gram_ext   : S S' for S = sqrt(d) o [A~ rows ; Q0'W ; Q0'g ; Q0'y]  (one pass over r, FP64 MFMA)
//   finalize   : X'K^-1X etc., the (c+1)x(c+1) solve, Q = 1/2 |u|^2,
//                F = 1/2 (D'K^-1 D - D'K^-1 X (X'K^-1X)^-1 X'K^-1 D)
#include <type_traits>

#include "nullfit.h"

namespace crm {

namespace {

constexpr int CH = 64;   // spectrum entries staged per step
constexpr int SLD = 66;  // LDS row length: 16 rows x 2 columns of a fragment read land on 32 distinct bank pairs

typedef double v4d __attribute__((ext_vector_type(4)));

// S S' with S = sqrt(d) o [A~ rows ; Q0'W ; Q0'g ; Q0'y] (KT x r) on the FP64 matrix pipe.  One
// workgroup per variant; S is staged through LDS in chunks of 64 spectrum entries (scaled on the way
// in, next chunk prefetched into registers during the MFMAs).  Only the upper triangle of 16x16
// output tiles is computed; the tiles are dealt round-robin to the four wavefronts, rotated by the
// block index so that the SIMDs of a CU see the same load.  NG > 1 (more than 144 rows: the slower form for many contexts +
// covariates): NG workgroups per variant, workgroup g of them takes the tiles g, g + NG, ... of the list -- every one
// stages all rows, the accumulators of a wavefront stay within its registers.
template <int NTL, int NG = 1>
__global__ __launch_bounds__(256) void gram_ext_kernel(AssembleArgs a, double* __restrict__ Gext, int KT) {
    constexpr int ROWS = 16 * NTL, NTILES = NTL * (NTL + 1) / 2, NLOC = (NTILES + NG - 1) / NG, MAXT = (NLOC + 3) / 4,
                  RPT = ROWS / 4;
    extern __shared__ double Ss[];  // [ROWS][SLD]
    __shared__ int tile_ij[NTILES];
    __shared__ const double* tail_ptr[CRM_MAX_COV_XWIDE + 2];
    const int b = blockIdx.x;
    const NullFitOut fit = a.fit[b];
    const AssembleRho R = a.rho[fit.rho_index];
    const int r = R.r;
    const double ratio = fit.v0 / fit.v1;
    const long pos = a.sorted_pos[b];
    const double* __restrict__ Arows = pos >= 0 ? a.A + pos * a.k0 * a.ldA : a.A_none;
    const long a_stride = pos >= 0 ? a.ldA : 0;      // (no position: every row is the row of zeros)
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int k0 = a.k0, c = a.c;

    for (int e = tid; e < NTILES; e += 256) {
        int ti = 0, rem = e;
        while (rem >= NTL - ti) { rem -= NTL - ti; ti++; }
        tile_ij[e] = ti | ((ti + rem) << 8);
    }
    if (tid < c + 2) {
        const double* p;
        if (tid < c) p = R.tW + (long)tid * R.ldW;
        else if (tid == c) p = R.T + (long)b * R.ldT;
        else p = R.ty;
        tail_ptr[tid] = p;
    }
    __syncthreads();
    // tiles of this wavefront: the entries first, first + 4, ... of this workgroup's list (MAXT or MAXT - 1 of them)
    const int grp = NG > 1 ? (int)blockIdx.y : 0;
    const int first = (wave + b) & 3;
    const int nloc = (NTILES - grp + NG - 1) / NG;
    const int ntw = (nloc - first + 3) / 4;
    int t_i[MAXT], t_j[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; t++) {
        const int idx = grp + NG * (first + 4 * t);
        const int ij = tile_ij[idx < NTILES ? idx : 0];
        t_i[t] = ((ij & 255) * 16 + l15) * SLD + lq;
        t_j[t] = ((ij >> 8) * 16 + l15) * SLD + lq;
    }
    v4d acc[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; t++) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};

    const int cc = lane;          // column of the chunk this thread stages
    auto fetch = [&](int c0, double (&v)[RPT]) __attribute__((always_inline)) {
        const int j = c0 + cc;
        const bool okj = j < r;
        double sdv = 0.0;
        if (okj) {
            const double s = ratio * R.S0[j];
            sdv = sqrt(s / (1.0 + s));
        }
#pragma unroll
        for (int i = 0; i < RPT; i++) {
            const int row = wave + 4 * i;
            double x = 0.0;
            if (okj && row < KT) {
                const double* __restrict__ p = row < k0 ? Arows + (long)row * a_stride : tail_ptr[row - k0];
                x = p[j];
            }
            v[i] = x * sdv;
        }
    };
    auto stash = [&](const double (&v)[RPT]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < RPT; i++) Ss[(wave + 4 * i) * SLD + cc] = v[i];
    };
    // MFMAs of one staged chunk for NT tiles; operand fragments are read one k-step ahead
    auto chunk_mma = [&](auto nt_tag) __attribute__((always_inline)) {
        constexpr int NT = decltype(nt_tag)::value;
        if constexpr (NT > 0) {
            double fa[2][NT], fb[2][NT];
#pragma unroll
            for (int t = 0; t < NT; t++) {
                fa[0][t] = Ss[t_i[t]];
                fb[0][t] = Ss[t_j[t]];
            }
#pragma unroll
            for (int ks = 0; ks < CH / 4; ks++) {
                const int cur = ks & 1, nx = cur ^ 1;
                if (ks + 1 < CH / 4) {
#pragma unroll
                    for (int t = 0; t < NT; t++) {
                        fa[nx][t] = Ss[t_i[t] + 4 * (ks + 1)];
                        fb[nx][t] = Ss[t_j[t] + 4 * (ks + 1)];
                    }
                }
#pragma unroll
                for (int t = 0; t < NT; t++)
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[cur][t], fb[cur][t], acc[t], 0, 0, 0);
            }
        }
    };

    // chunk ch sits in LDS while chunk ch+1 (and, with DEEP, ch+2) are in registers / in flight: the
    // kernel streams 8 (KT r) bytes per variant and needs the loads outstanding during the MFMAs
    constexpr bool DEEP = false;  // measured: two chunks ahead costs occupancy (146 VGPRs) and gains nothing
    const int nchunks = (r + CH - 1) / CH;
    auto run_mma = [&]() __attribute__((always_inline)) {
        if (ntw == MAXT) chunk_mma(std::integral_constant<int, MAXT>{});
        else chunk_mma(std::integral_constant<int, MAXT - 1>{});
    };
    double bA[RPT], bB[DEEP ? RPT : 1];
    fetch(0, bA);
    stash(bA);
    __syncthreads();
    fetch(CH, bA);
    if constexpr (DEEP) {
        fetch(2 * CH, bB);
        for (int ch = 0; ch < nchunks; ch += 2) {
            run_mma();
            __syncthreads();
            if (ch + 1 < nchunks) {
                stash(bA);
                __syncthreads();
                fetch((ch + 3) * CH, bA);
                run_mma();
                __syncthreads();
                if (ch + 2 < nchunks) {
                    stash(bB);
                    __syncthreads();
                    fetch((ch + 4) * CH, bB);
                }
            }
        }
    } else {
        for (int ch = 0; ch < nchunks; ch++) {
            run_mma();
            __syncthreads();
            if (ch + 1 < nchunks) {
                stash(bA);
                __syncthreads();
                fetch((ch + 2) * CH, bA);
            }
        }
    }
    double* __restrict__ out = Gext + (long)b * KT * KT;
#pragma unroll
    for (int t = 0; t < MAXT; t++) {
        const int idx = grp + NG * (first + 4 * t);
        if (idx >= NTILES) continue;
        const int ij = tile_ij[idx];
        const int ri = (ij & 255) * 16, cj = (ij >> 8) * 16;
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int row = ri + lq + 4 * reg, col = cj + l15;
            if (row < KT && col < KT) {
                out[(long)row * KT + col] = acc[t][reg];
                if (ri != cj) out[(long)col * KT + row] = acc[t][reg];
            }
        }
    }
}

// The same Gram with the operand rows brought in by direct-to-LDS loads (global_load_lds_dwordx4), for 16 NTL rows
// with NTL even.  The staged kernel above spends a third of its issue slots on the way into LDS -- per chunk and
// thread sixteen 8-byte loads with 64-bit address arithmetic, a square root and a division for sqrt(d_j), sixteen
// products and sixteen ds_write -- none of which can overlap an FP64 MFMA, and two barriers per chunk.  Here a
// wave-instruction carries two rows of a chunk (lanes 0-31 row q, lanes 32-63 row q + ROWS/2, 64 columns each) straight
// into LDS, two buffers deep, one barrier per chunk; the spectrum weights go in as d_j (not sqrt d_j) on ONE operand,
// a product per fragment read:  S S' = sum_j a_ij d_j a_i'j.  LDS image: instruction q at q * 130 doubles (its two
// rows 64 doubles apart), so the sixteen rows of a fragment read start two 8-byte slots apart -- conflict-free like the
// 66-double rows of the staged kernel.  A last, partial chunk is loaded through registers with per-element guards.
typedef const double __attribute__((address_space(1))) * gram_gptr_t;
typedef __attribute__((address_space(3))) void* gram_lptr_t;

// (a __device__ function: the builtin does not exist in the host pass, which would drop the kernel's host stub)
__device__ __forceinline__ void gram_dma16(gram_gptr_t src, double* lds_wave_uniform) {
    __builtin_amdgcn_global_load_lds(src, (gram_lptr_t)lds_wave_uniform, 16, 0, 0);
}

template <int NTL>
__global__ __launch_bounds__(256) void gram_ext_dma_kernel(AssembleArgs a, double* __restrict__ Gext, int KT) {
    static_assert(NTL % 2 == 0, "row pairs of one DMA instruction must fall into different 16-row tiles");
    constexpr int ROWS = 16 * NTL, HALF = ROWS / 2, NTILES = NTL * (NTL + 1) / 2;
    constexpr int QLD = 130;                 // doubles per DMA instruction slot (2 x 64 + 2 of padding)
    constexpr int BUF = HALF * QLD;          // doubles per chunk buffer
    constexpr int IPW = HALF / 4;            // DMA instructions per wavefront and chunk
    constexpr int KSW = CH / 16;             // k-steps per wavefront and chunk (the four wavefronts split the chunk's columns)
    static_assert(2 * BUF >= 2 * NTILES * 256, "the accumulators of two wavefronts must fit the chunk buffers");
    extern __shared__ double Ss[];           // [2][BUF] + d [2][64]
    double* const dS = Ss + 2 * BUF;
    const int b = blockIdx.x;
    const NullFitOut fit = a.fit[b];
    const AssembleRho R = a.rho[fit.rho_index];
    const int r = R.r;
    const double ratio = fit.v0 / fit.v1;
    const long pos = a.sorted_pos[b];
    const double* __restrict__ Arows = pos >= 0 ? a.A + pos * a.k0 * a.ldA : a.A_none;
    const long a_stride = pos >= 0 ? a.ldA : 0;      // (no position: every row is the row of zeros)
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int k0 = a.k0, c = a.c;

    // row -> pointer (rows past KT read row 0: their outputs are never stored)
    auto row_ptr = [&](int row) -> const double* {
        if (row >= KT) row = 0;
        if (row < k0) return Arows + (long)row * a_stride;
        const int t = row - k0;
        return t < c ? R.tW + (long)t * R.ldW : (t == c ? R.T + (long)b * R.ldT : R.ty);
    };
    auto row_off = [](int row) { return row < HALF ? row * QLD : (row - HALF) * QLD + 64; };
    // this lane's source addresses (column pair 2 * (lane & 31) of the row its half-wave carries)
    gram_gptr_t src[IPW];
#pragma unroll
    for (int q = 0; q < IPW; q++) {
        const int inst = wave + 4 * q;
        src[q] = (gram_gptr_t)(row_ptr(inst + (lane >> 5) * HALF) + 2 * (lane & 31));
    }
    // Every wavefront accumulates ALL upper-triangle tiles over its quarter of the columns of each chunk (an even
    // split whatever the tile count; the fragment of a 16-row block serves as the A operand and, times d_j, as the B
    // operand of every tile that touches the block: NTL reads and NTL products per k-step for NTILES MFMAs); the four
    // partial sums meet through LDS once per variant, in a fixed order.
    int f_off[NTL];
#pragma unroll
    for (int t = 0; t < NTL; t++) f_off[t] = row_off(16 * t + l15) + 4 * KSW * wave + lq;
    v4d acc[NTILES];
#pragma unroll
    for (int t = 0; t < NTILES; t++) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};

    const int nfull = r / CH, nchunks = (r + CH - 1) / CH;
    auto weights = [&](int ch, int buf) __attribute__((always_inline)) {   // d_j of chunk ch (wavefront 0)
        if (wave == 0) {
            const int j = ch * CH + lane;
            double d = 0.0;
            if (j < r) {
                const double s = ratio * R.S0[j];
                d = s / (1.0 + s);
            }
            dS[buf * CH + lane] = d;
        }
    };
    auto issue = [&](int ch, int buf) __attribute__((always_inline)) {
        if (ch < nfull) {
#pragma unroll
            for (int q = 0; q < IPW; q++)
                gram_dma16(src[q] + (long)ch * CH, Ss + buf * BUF + (wave + 4 * q) * QLD);
        } else {   // the partial chunk: guarded element loads, zeros past r
            for (int e = tid; e < ROWS * CH; e += 256) {
                const int row = e / CH, col = e - row * CH;
                const int j = ch * CH + col;
                Ss[buf * BUF + row_off(row) + col] = (j < r && row < KT) ? row_ptr(row)[j] : 0.0;
            }
        }
    };
    auto chunk_mma = [&](int buf) __attribute__((always_inline)) {
        const double* __restrict__ S = Ss + buf * BUF;
        const double* __restrict__ D = dS + buf * CH + 4 * KSW * wave + lq;
        double fa[2][NTL], fb[2][NTL];
        {
            const double dk = D[0];
#pragma unroll
            for (int t = 0; t < NTL; t++) {
                fa[0][t] = S[f_off[t]];
                fb[0][t] = fa[0][t] * dk;
            }
        }
#pragma unroll
        for (int ks = 0; ks < KSW; ks++) {
            const int cur = ks & 1, nx = cur ^ 1;
            if (ks + 1 < KSW) {
                const double dk = D[4 * (ks + 1)];
#pragma unroll
                for (int t = 0; t < NTL; t++) {
                    fa[nx][t] = S[f_off[t] + 4 * (ks + 1)];
                    fb[nx][t] = fa[nx][t] * dk;
                }
            }
            int idx = 0;
#pragma unroll
            for (int ti = 0; ti < NTL; ti++)
#pragma unroll
                for (int tj = ti; tj < NTL; tj++, idx++)
                    acc[idx] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[cur][ti], fb[cur][tj], acc[idx], 0, 0, 0);
        }
    };
    issue(0, 0);
    weights(0, 0);
    __syncthreads();   // (drains the LDS-DMA: vmcnt(0) precedes the barrier)
    for (int ch = 0; ch < nchunks; ch++) {
        const int buf = ch & 1;
        if (ch + 1 < nchunks) {
            issue(ch + 1, buf ^ 1);
            weights(ch + 1, buf ^ 1);
        }
        chunk_mma(buf);
        __syncthreads();
    }
    // (wave 0 + wave 2) + (wave 1 + wave 3): two exchanges through the chunk buffers
    double* const red = Ss + (wave & 1) * NTILES * 256;
    if (wave >= 2) {
#pragma unroll
        for (int t = 0; t < NTILES; t++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) red[(t * 4 + reg) * 64 + lane] = acc[t][reg];
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int t = 0; t < NTILES; t++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) acc[t][reg] += red[(t * 4 + reg) * 64 + lane];
    }
    __syncthreads();
    if (wave == 1) {
#pragma unroll
        for (int t = 0; t < NTILES; t++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) red[(t * 4 + reg) * 64 + lane] = acc[t][reg];
    }
    __syncthreads();
    if (wave != 0) return;
    double* __restrict__ out = Gext + (long)b * KT * KT;
    int idx = 0;
#pragma unroll
    for (int ti = 0; ti < NTL; ti++) {
#pragma unroll
        for (int tj = ti; tj < NTL; tj++, idx++) {
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const double v = acc[idx][reg] + (Ss + NTILES * 256)[(idx * 4 + reg) * 64 + lane];
                const int row = 16 * ti + lq + 4 * reg, col = 16 * tj + l15;
                if (row < KT && col < KT) {
                    out[(long)row * KT + col] = v;
                    if (ti != tj) out[(long)col * KT + row] = v;
                }
            }
        }
    }
}

constexpr int PMAX = CRM_MAX_COV_XWIDE + 1;

__global__ __launch_bounds__(128) void finalize_kernel(AssembleArgs a, const double* __restrict__ Gext,
                                                        int KT, double* __restrict__ rows) {
    // per variant: everything below is k0- or (c+1)-sized; LDS carved from the dynamic segment:
    // L [P][P] (Cholesky factor of X'K^-1X), xky [P], uvec [k0], vv, ww [P] each, then dkx [k0][P] (D'K^-1X) and
    // sol [k0][P] -- or, when those two do not fit (many contexts x many covariates), in `rows` (global memory,
    // 2 k0 P doubles per variant: the slower form)
    extern __shared__ double fsm[];
    __shared__ int ok_flag, keep_flag;
    __shared__ double rescale;     // 1 / scale where the fit record asks for the scale to be derived here (fit.scale < 0)
    const int Pd = a.c + 1;
    double* Lm = fsm;
    double* xky = Lm + Pd * Pd;
    double* uvec = xky + Pd;
    double* vv = uvec + a.k0;
    double* ww = vv + Pd;
    double* dkxm = rows ? rows + (size_t)blockIdx.x * 2 * a.k0 * Pd : ww + Pd;
    double* solm = dkxm + a.k0 * Pd;
#define L(i, j) Lm[(i) * Pd + (j)]
#define dkx(j, i) dkxm[(j) * Pd + (i)]
#define sol(j, i) solm[(j) * Pd + (i)]
    const int b = blockIdx.x;
    const NullFitOut fit = a.fit[b];
    const int k0 = a.k0, c = a.c;
    const int P = c + 1;
    const double v1 = fit.v1;
    const double inv = 1.0 / v1;
    const double* __restrict__ Ge = Gext + (long)b * KT * KT;
    const int tid = threadIdx.x;
    auto plain_xx = [&](int i, int j) -> double {  // X'X entries, X = [W, g]
        if (i > j) { int t = i; i = j; j = t; }
        if (j < c) return a.WW[i * c + j];
        if (i < c) return a.gW[(long)b * a.ld_gW + i];
        return a.gg[b];
    };
    auto plain_xy = [&](int i) -> double { return i < c ? a.Wy[i] : a.gy[b]; };

    if (tid == 0) {
        bool ok = true;
        for (int i = 0; i < P; i++) {
            for (int j = 0; j <= i; j++)
                L(i, j) = (plain_xx(i, j) - Ge[(long)(k0 + i) * KT + (k0 + j)]) * inv;
            xky[i] = (plain_xy(i) - Ge[(long)(k0 + i) * KT + (k0 + c + 1)]) * inv;
        }
        // (whether the variant is a column of the projection is PMat's own decision, below -- not the null fit's use_g:
        // the reference's LMM and its PMat apply different rank rules to [W, g])
        bool keep = true;
        for (int j = 0; j < P && ok; j++) {
            double d = L(j, j);
            for (int k = 0; k < j; k++) d -= L(j, k) * L(j, k);
            if (!(d > 0.0)) {
                if (j == c) keep = false;   // no component outside span(W) at all
                else ok = false;
                break;
            }
            const double l = sqrt(d);
            L(j, j) = l;
            for (int i = j + 1; i < P; i++) {
                double s = L(i, j);
                for (int k = 0; k < j; k++) s -= L(i, k) * L(j, k);
                L(i, j) = s / l;
            }
        }
        if (ok && keep) {
            // PMat solves with numpy's lstsq(rcond=None) on X'K^-1X in the basis of X = [W, g] AS GIVEN (_math.py:33-37,
            // :91-93): a singular value below eps * (c + 1) * the largest is cut off -- a rule of its own, apart from the
            // LMM's economic_svd (absolute sqrt(eps) on the singular values of X, use_g above): with columns of norm
            // ~ sqrt(n) a variant can be kept by the null fit and still be cut here.  The matrix at hand is the one of
            // [W', x] with x = g - W' a (a: the block's projection coefficients; none on the collapsed path), i.e.
            // A_given = T' (L L') T, T = [[I, a], [0, 1]] up to an orthogonal change of W's basis, which leaves the
            // singular values alone.  Smallest one: the Rayleigh quotient of the near-null vector (-(a + b), 1), b the
            // K-metric regression of x on W' -- s / (1 + |a + b|^2) with s the Schur complement (last pivot squared);
            // largest one: a few power iterations through the factor.  Cutting that direction leaves the projection of W
            // alone (to the order of the singular-value ratio).
            double nrm2 = 1.0;
            {
                // b = L_WW^-T L(c, 0..c-1)'
                for (int i = c - 1; i >= 0; i--) {
                    double sacc = L(c, i);
                    for (int k = i + 1; k < c; k++) sacc -= L(k, i) * vv[k];
                    vv[i] = sacc / L(i, i);
                }
                for (int j = 0; j < c; j++) {
                    const double t = vv[j] + (a.coef ? a.coef[(long)j * a.ld_coef + b] : 0.0);
                    nrm2 += t * t;
                }
            }
            const double lam_min = L(c, c) * L(c, c) / nrm2;
            // power iteration on T' L L' T
            for (int i = 0; i < P; i++) vv[i] = 1.0;
            double lam_max = 0.0;
            for (int it = 0; it < 12; it++) {
                // u = T v
                const double vl = vv[c];
                for (int j = 0; j < c; j++) ww[j] = vv[j] + (a.coef ? a.coef[(long)j * a.ld_coef + b] : 0.0) * vl;
                ww[c] = vl;
                // w = L' u (in vv), then u = L w (in ww)
                for (int i = 0; i < P; i++) {
                    double sacc = 0.0;
                    for (int k = i; k < P; k++) sacc += L(k, i) * ww[k];
                    vv[i] = sacc;
                }
                for (int i = P - 1; i >= 0; i--) {
                    double sacc = 0.0;
                    for (int k = 0; k <= i; k++) sacc += L(i, k) * vv[k];
                    ww[i] = sacc;
                }
                // v = T' u
                double last = ww[c];
                for (int j = 0; j < c; j++) last += (a.coef ? a.coef[(long)j * a.ld_coef + b] : 0.0) * ww[j];
                double n2 = last * last;
                for (int j = 0; j < c; j++) n2 += ww[j] * ww[j];
                lam_max = sqrt(n2);
                if (!(lam_max > 0.0)) break;
                for (int j = 0; j < c; j++) vv[j] = ww[j] / lam_max;
                vv[c] = last / lam_max;
            }
            if (lam_min <= 2.220446049250313e-16 * (double)P * lam_max) keep = false;
        }
        if (!keep) {  // the projection is the one of W alone
            for (int j = 0; j < c; j++) L(c, j) = 0.0;
            L(c, c) = 1.0;
            xky[c] = 0.0;
        }
        keep_flag = keep ? 1 : 0;
        rescale = 1.0;
        if (ok) {
            for (int i = 0; i < P; i++) {
                double s = xky[i];
                for (int k = 0; k < i; k++) s -= L(i, k) * xky[k];
                xky[i] = s / L(i, i);
            }
            if (fit.scale < 0.0) {
                // A record of (v0, v1) = (1 - delta, delta) without its scale (scan.hip: the probes of the flat-optimum flag):
                // the REML scale at that delta is y'Py / df with P of the unit-scale covariance, from what is at hand --
                // y'K^-1y and the forward-substituted X'K^-1y (glimix-core: LMM.scale; nullfit.hip evaluates the same).
                double r = (a.yy - Ge[(long)(k0 + c + 1) * KT + (k0 + c + 1)]) * inv;
                const int used = c + ((fit.use_g && keep) ? 1 : 0);
                for (int i = 0; i < used; i++) r -= xky[i] * xky[i];
                const double df = (double)a.n - (double)(c + (fit.use_g ? 1 : 0));
                rescale = 1.0 / fmax(r / df, 1.4901161193847656e-08);
            }
            for (int i = P - 1; i >= 0; i--) {
                double s = xky[i];
                for (int k = i + 1; k < P; k++) s -= L(k, i) * xky[k];
                xky[i] = s / L(i, i);
            }
        }
        ok_flag = ok ? 1 : 0;
    }
    __syncthreads();
    const bool ok = ok_flag != 0;
    const bool keep_g = keep_flag != 0;
    // D'K^-1 X rows and their solves, one context per thread
    for (int j = tid; j < k0; j += blockDim.x) {
        double row[PMAX];
        for (int i = 0; i < P; i++) {
            double plain;
            if (i < c) plain = a.Z1[(long)b * a.ldZ1 + (long)(1 + i) * k0 + j];
            else plain = a.Z2[(long)b * a.ldZ2 + j];
            double v = (plain - Ge[(long)j * KT + (k0 + i)]) * inv;
            if (i == c && !keep_g) v = 0.0;
            row[i] = v;
            dkx(j, i) = v;
        }
        for (int i = 0; i < P; i++) {
            double s = row[i];
            for (int k = 0; k < i; k++) s -= L(i, k) * row[k];
            row[i] = s / L(i, i);
        }
        for (int i = P - 1; i >= 0; i--) {
            double s = row[i];
            for (int k = i + 1; k < P; k++) s -= L(k, i) * row[k];
            row[i] = s / L(i, i);
        }
        for (int i = 0; i < P; i++) sol(j, i) = row[i];
        const double dky = (a.Z1[(long)b * a.ldZ1 + j] - Ge[(long)j * KT + (k0 + c + 1)]) * inv;
        double u = dky;
        for (int i = 0; i < P; i++) u -= dkx(j, i) * xky[i];
        uvec[j] = u * rescale;
    }
    __syncthreads();
    if (tid == 0) {
        double q = 0.0;
        for (int j = 0; j < k0; j++) q += uvec[j] * uvec[j];
        a.Q[b] = ok ? 0.5 * q : NAN;
    }
    double* __restrict__ F = a.F + (long)b * k0 * k0;
    for (int e = tid; e < k0 * k0; e += blockDim.x) {
        const int j = e / k0, jp = e - j * k0;
        const int lo = j < jp ? j : jp, hi = j < jp ? jp : j;
        // pair index of (lo, hi) in the row-major upper triangle
        const long pidx = (long)lo * k0 - (long)lo * (lo - 1) / 2 + (hi - lo);
        double v = (a.Z3[(long)b * a.ldZ3 + pidx] - Ge[(long)j * KT + jp]) * inv;
        for (int i = 0; i < P; i++) v -= dkx(j, i) * sol(jp, i);
        F[e] = ok ? 0.5 * v * rescale : NAN;
    }
}
#undef L
#undef dkx
#undef sol

}  // namespace

size_t assemble_rows_scratch_doubles(int variants, int k0, int c) {
    const size_t P = (size_t)c + 1;
    const size_t lds = sizeof(double) * (P * P + 3 * P + k0 + 2 * (size_t)k0 * P);
    return lds > 150 * 1024 ? (size_t)variants * 2 * k0 * P : 0;
}

int launch_assemble(hipStream_t st, const AssembleArgs& a, int variants, double* Gext, double* fin_rows) {
    if (variants <= 0) return CRM_OK;
    const int KT = a.k0 + a.c + 2;
    const int P = a.c + 1;
    const size_t fin_small = sizeof(double) * ((size_t)P * P + 3 * P + a.k0);
    size_t fin_lds = fin_small + sizeof(double) * 2 * (size_t)a.k0 * P;
    const bool rows_global = fin_lds > 150 * 1024;
    if (rows_global) fin_lds = fin_small;
    if (a.k0 > CRM_MAX_K0 || a.c > CRM_MAX_COV_XWIDE || KT > CRM_MAX_GRAM_ROWS || (rows_global && !fin_rows)) {
        set_error("assemble: k0=%d, c=%d outside the supported range (k0 <= %d, c <= %d, k0 + c + 2 <= %d)", a.k0, a.c,
                  CRM_MAX_K0, CRM_MAX_COV_XWIDE, CRM_MAX_GRAM_ROWS);
        return CRM_ERR_UNSUPPORTED;
    }
    const int ts = (KT + 15) / 16;
    // LDS-DMA form: 16-byte loads, so every row must start on a 16-byte boundary and hold an even number of doubles
    bool dma = ts <= 4 && a.ldA % 2 == 0 && (reinterpret_cast<uintptr_t>(a.A) & 15) == 0 && !form("gram_staged", 0);
    for (int i = 0; dma && i < CRM_MAX_RHO; i++) {
        const AssembleRho& R = a.rho[i];
        if (R.r <= 0 && !R.ty) continue;
        dma = R.ldW % 2 == 0 && R.ldT % 2 == 0 && ((reinterpret_cast<uintptr_t>(R.ty) | reinterpret_cast<uintptr_t>(R.tW) |
                                                    reinterpret_cast<uintptr_t>(R.T)) & 15) == 0;
    }
#define CRM_GRAM_DMA(NTL)                                                                                     \
    do {                                                                                                      \
        const size_t lds = sizeof(double) * (2 * (16 * NTL / 2) * 130 + 2 * CH);                              \
        if (lds > 60 * 1024)                                                                                  \
            CRM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gram_ext_dma_kernel<NTL>),             \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));               \
        hipLaunchKernelGGL(gram_ext_dma_kernel<NTL>, dim3(variants), dim3(256), lds, st, a, Gext, KT);        \
    } while (0)
    if (dma) {
        if (ts <= 2) CRM_GRAM_DMA(2);
        else CRM_GRAM_DMA(4);   // (k0 + c + 2 > 64: 21 tiles per wavefront would leave one wavefront per SIMD -- staged kernel)
    }
#undef CRM_GRAM_DMA
#define CRM_GRAM(NTL)                                                                                         \
    do {                                                                                                      \
        const size_t lds = sizeof(double) * 16 * NTL * SLD;                                                   \
        if (lds > 60 * 1024)                                                                                  \
            CRM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gram_ext_kernel<NTL>),                 \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));               \
        hipLaunchKernelGGL(gram_ext_kernel<NTL>, dim3(variants), dim3(256), lds, st, a, Gext, KT);            \
    } while (0)
#define CRM_GRAM_GROUPS(NTL, NG)                                                                              \
    do {                                                                                                      \
        const size_t lds = sizeof(double) * 16 * NTL * SLD;                                                   \
        CRM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gram_ext_kernel<NTL, NG>),                 \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                   \
        hipLaunchKernelGGL((gram_ext_kernel<NTL, NG>), dim3(variants, NG), dim3(256), lds, st, a, Gext, KT);  \
    } while (0)
    if (dma) {
    } else if (ts <= 2) CRM_GRAM(2);
    else if (ts <= 4) CRM_GRAM(4);
    else if (ts <= 6) CRM_GRAM(6);
    else if (ts <= 9) CRM_GRAM(9);
    else if (ts <= 12) CRM_GRAM_GROUPS(12, 2);
    else if (ts <= 15) CRM_GRAM_GROUPS(15, 4);
    else CRM_GRAM_GROUPS(18, 4);
#undef CRM_GRAM
#undef CRM_GRAM_GROUPS
    CRM_HIP(hipGetLastError());
    if (fin_lds > 60 * 1024)
        CRM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&finalize_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)fin_lds));
    hipLaunchKernelGGL(finalize_kernel, dim3(variants), dim3(128), fin_lds, st, a, Gext, KT, rows_global ? fin_rows : nullptr);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

}  // namespace crm
