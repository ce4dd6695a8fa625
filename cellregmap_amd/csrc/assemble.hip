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
//   gram_ext   : S S' for S = sqrt(d) o [A~ rows ; Q0'W ; Q0'g ; Q0'y]  (one pass over r)
//   finalize   : X'K^-1X etc., the (c+1)x(c+1) solve, Q = 1/2 |u|^2,
//                F = 1/2 (D'K^-1 D - D'K^-1 X (X'K^-1X)^-1 X'K^-1 D)
#include "nullfit.h"

namespace crm {

namespace {

constexpr int CH = 64;  // spectrum entries staged per step

template <int TS>
__global__ __launch_bounds__(256) void gram_ext_kernel(AssembleArgs a, double* __restrict__ Gext,
                                                        int KT) {
    __shared__ double Ss[16 * TS][CH + 1];
    __shared__ double sd[CH];
    const int b = blockIdx.x;
    const NullFitOut fit = a.fit[b];
    const AssembleRho R = a.rho[fit.rho_index];
    const int r = R.r;
    const double ratio = fit.v0 / fit.v1;
    const long pos = a.sorted_pos[b];
    const double* __restrict__ Arows = a.A + pos * a.k0 * a.ldA;
    const double* __restrict__ tg = R.T + (long)b * R.ldT;
    const int tid = threadIdx.x;
    const int ti = tid >> 4, tj = tid & 15;
    const int k0 = a.k0, c = a.c;

    double acc[TS][TS];
#pragma unroll
    for (int i = 0; i < TS; i++)
#pragma unroll
        for (int j = 0; j < TS; j++) acc[i][j] = 0.0;

    for (int c0 = 0; c0 < r; c0 += CH) {
        if (tid < CH) {
            const int j = c0 + tid;
            double v = 0.0;
            if (j < r) {
                const double s = ratio * R.S0[j];
                v = sqrt(s / (1.0 + s));
            }
            sd[tid] = v;
        }
        __syncthreads();
        for (int e = tid; e < 16 * TS * CH; e += 256) {
            const int row = e / CH, cc = e - row * CH;
            const int j = c0 + cc;
            double v = 0.0;
            if (row < KT && j < r) {
                if (row < k0) v = Arows[(long)row * a.ldA + j];
                else if (row < k0 + c) v = R.tW[(long)(row - k0) * R.ldW + j];
                else if (row == k0 + c) v = tg[j];
                else v = R.ty[j];
                v *= sd[cc];
            }
            Ss[row][cc] = v;
        }
        __syncthreads();
#pragma unroll 4
        for (int cc = 0; cc < CH; cc++) {
            double x[TS], y[TS];
#pragma unroll
            for (int i = 0; i < TS; i++) {
                x[i] = Ss[ti + 16 * i][cc];
                y[i] = Ss[tj + 16 * i][cc];
            }
#pragma unroll
            for (int i = 0; i < TS; i++)
#pragma unroll
                for (int j = 0; j < TS; j++) acc[i][j] += x[i] * y[j];
        }
        __syncthreads();
    }
    double* __restrict__ out = Gext + (long)b * KT * KT;
#pragma unroll
    for (int i = 0; i < TS; i++) {
        const int row = ti + 16 * i;
        if (row >= KT) continue;
#pragma unroll
        for (int j = 0; j < TS; j++) {
            const int col = tj + 16 * j;
            if (col < KT) out[(long)row * KT + col] = acc[i][j];
        }
    }
}

constexpr int PMAX = CRM_MAX_COV_WIDE + 1;

__global__ __launch_bounds__(128) void finalize_kernel(AssembleArgs a, const double* __restrict__ Gext,
                                                        int KT) {
    // per variant: everything below is k0- or (c+1)-sized; LDS carved from the dynamic segment:
    // L [P][P] (Cholesky factor of X'K^-1X), xky [P], dkx [k0][P] (D'K^-1X), sol [k0][P], uvec [k0]
    extern __shared__ double fsm[];
    __shared__ int ok_flag;
    const int Pd = a.c + 1;
    double* Lm = fsm;
    double* xky = Lm + Pd * Pd;
    double* dkxm = xky + Pd;
    double* solm = dkxm + a.k0 * Pd;
    double* uvec = solm + a.k0 * Pd;
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
        if (!fit.use_g) {  // g in span(W): the projection is the one of W alone
            for (int j = 0; j < c; j++) L(c, j) = 0.0;
            L(c, c) = 1.0;
            xky[c] = 0.0;
        }
        for (int j = 0; j < P && ok; j++) {
            double d = L(j, j);
            for (int k = 0; k < j; k++) d -= L(j, k) * L(j, k);
            if (!(d > 0.0)) { ok = false; break; }
            const double l = sqrt(d);
            L(j, j) = l;
            for (int i = j + 1; i < P; i++) {
                double s = L(i, j);
                for (int k = 0; k < j; k++) s -= L(i, k) * L(j, k);
                L(i, j) = s / l;
            }
        }
        if (ok) {
            for (int i = 0; i < P; i++) {
                double s = xky[i];
                for (int k = 0; k < i; k++) s -= L(i, k) * xky[k];
                xky[i] = s / L(i, i);
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
    // D'K^-1 X rows and their solves, one context per thread
    for (int j = tid; j < k0; j += blockDim.x) {
        double row[PMAX];
        for (int i = 0; i < P; i++) {
            double plain;
            if (i < c) plain = a.Z1[(long)b * a.ldZ1 + (long)(1 + i) * k0 + j];
            else plain = a.Z2[(long)b * a.ldZ2 + j];
            double v = (plain - Ge[(long)j * KT + (k0 + i)]) * inv;
            if (i == c && !fit.use_g) v = 0.0;
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
        uvec[j] = u;
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
        F[e] = ok ? 0.5 * v : NAN;
    }
}
#undef L
#undef dkx
#undef sol

}  // namespace

int launch_assemble(hipStream_t st, const AssembleArgs& a, int variants, double* Gext) {
    if (variants <= 0) return CRM_OK;
    const int KT = a.k0 + a.c + 2;
    const int P = a.c + 1;
    const size_t fin_lds = sizeof(double) * ((size_t)P * P + P + 2 * (size_t)a.k0 * P + a.k0);
    if (a.k0 > CRM_MAX_K0 || a.c > CRM_MAX_COV_WIDE || KT > 144 || fin_lds > 150 * 1024) {
        set_error("assemble: k0=%d, c=%d outside the supported range (k0 <= %d, c <= %d, k0 + c + 2 <= 144, "
                  "(c+1)(2 k0 + c + 2) <= 19000)", a.k0, a.c, CRM_MAX_K0, CRM_MAX_COV_WIDE);
        return CRM_ERR_UNSUPPORTED;
    }
    const int ts = (KT + 15) / 16;
    if (ts <= 2) hipLaunchKernelGGL(gram_ext_kernel<2>, dim3(variants), dim3(256), 0, st, a, Gext, KT);
    else if (ts <= 4) hipLaunchKernelGGL(gram_ext_kernel<4>, dim3(variants), dim3(256), 0, st, a, Gext, KT);
    else if (ts <= 6) hipLaunchKernelGGL(gram_ext_kernel<6>, dim3(variants), dim3(256), 0, st, a, Gext, KT);
    else hipLaunchKernelGGL(gram_ext_kernel<9>, dim3(variants), dim3(256), 0, st, a, Gext, KT);
    CRM_HIP(hipGetLastError());
    if (fin_lds > 60 * 1024)
        CRM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&finalize_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)fin_lds));
    hipLaunchKernelGGL(finalize_kernel, dim3(variants), dim3(128), fin_lds, st, a, Gext, KT);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

}  // namespace crm
