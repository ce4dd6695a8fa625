// Association (persistent-effect) LRT paths, SURVEY 8a row a13:
//   scan_association       cellregmap/_cellregmap.py:246-281  (ML refit per SNP at the null's rho)
//   scan_association_fast  cellregmap/_cellregmap.py:284-314  (glimix-core FastScanner: delta frozen
//                          at the null, beta and one scale re-estimated per SNP in closed form)
//   lrt_pvalues            cellregmap/_cellregmap.py:443-469  (chi2_1 survival with the clips)
// The null fit and the per-SNP refits reuse the null-fit kernels (ML mode); this file holds the
// FastScanner closed form and the LRT.
#include "nullfit.h"
#include "objects.h"

namespace crm {

namespace {

constexpr double LOG2PI = 1.8378770664093453;
constexpr double EPS_SMALL = 1.4901161193847656e-08;
constexpr double DBL_TINY = 2.2250738585072014e-308;   // numpy_sugar.epsilon.super_tiny
constexpr double DBL_EPS = 2.220446049250313e-16;      // numpy_sugar.epsilon.tiny
constexpr int CMAX = CRM_MAX_COV_XWIDE;   // layout constant of the prep record (effects.hip reads it too)

__device__ inline double block_sum(double v, double* red) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    const double out = red[0];
    __syncthreads();
    return out;
}

// Null-model quantities at the frozen delta: L = chol(W'K^-1W), zy = L^-1 W'K^-1y,
// rss0 = y'K^-1y - zy'zy, logdet K, and the spectrum weights w_j = 1/((1-d) S0_j + d).
// prep layout: [0] rss0, [1] logdetK, [2] delta, [3] ok, [8 .. 8+c) zy, [8+CMAX ..) L (c x c, ld CMAX)
__global__ __launch_bounds__(256) void fastscan_prep_kernel(AssocArgs a, double* __restrict__ prep,
                                                             double* __restrict__ wts) {
    __shared__ double red[256];
    __shared__ double hy[CMAX];
    double* const H = prep + 8 + CMAX;   // (c x c, ld CMAX: in the record itself -- 128 x 128 doubles do not fit static LDS;
                                         // one workgroup, its own writes read back past the L1 by thread 0 only)
    const int tid = threadIdx.x;
    const int c = a.c, r = a.r;
    const double delta = a.delta0, inv_d = 1.0 / delta, omd = 1.0 - delta;
    double lpart = 0.0;
    for (int j = tid; j < r; j += 256) {
        const double D = omd * a.S0[j] + delta;
        wts[j] = 1.0 / D;
        lpart += log(D);
    }
    const double logdetK = block_sum(lpart, red) + ((double)a.n - (double)r) * log(delta);
    __syncthreads();
    // weighted and plain sums for all pairs among (W.., y)
    for (int u = 0; u <= c; u++) {
        const double* tu = u < c ? a.tW + (long)u * a.ldW : a.ty;
        for (int v = u; v <= c; v++) {
            const double* tv = v < c ? a.tW + (long)v * a.ldW : a.ty;
            double sw = 0.0, s1 = 0.0;
            for (int j = tid; j < r; j += 256) {
                const double p = tu[j] * tv[j];
                sw += p * wts[j];
                s1 += p;
            }
            sw = block_sum(sw, red);
            s1 = block_sum(s1, red);
            if (tid == 0) {
                double plain;
                if (v < c) plain = a.WW[u * c + v];
                else if (u < c) plain = a.Wy[u];
                else plain = a.yy;
                const double k = sw + (plain - s1) * inv_d;
                if (v < c) { H[u * CMAX + v] = k; H[v * CMAX + u] = k; }
                else if (u < c) hy[u] = k;
                else red[255] = k;  // y'K^-1 y
            }
            __syncthreads();
        }
    }
    if (tid == 0) {
        const double yKy = red[255];
        bool ok = true;
        for (int j = 0; j < c && ok; j++) {
            double d = H[j * CMAX + j];
            for (int k = 0; k < j; k++) d -= H[j * CMAX + k] * H[j * CMAX + k];
            if (!(d > 0.0)) { ok = false; break; }
            const double l = sqrt(d);
            H[j * CMAX + j] = l;
            for (int i = j + 1; i < c; i++) {
                double s = H[i * CMAX + j];
                for (int k = 0; k < j; k++) s -= H[i * CMAX + k] * H[j * CMAX + k];
                H[i * CMAX + j] = s / l;
            }
        }
        double rss0 = yKy;
        for (int i = 0; i < c && ok; i++) {
            double s = hy[i];
            for (int k = 0; k < i; k++) s -= H[i * CMAX + k] * prep[8 + k];
            s /= H[i * CMAX + i];
            prep[8 + i] = s;
            rss0 -= s * s;
        }
        prep[0] = rss0;
        prep[1] = logdetK;
        prep[2] = delta;
        prep[3] = ok ? 1.0 : 0.0;
    }
}

// Per SNP: h_gg, h_gy, h_gW in the frozen metric, Schur complement against W, ML log-likelihood.
__global__ __launch_bounds__(256) void fastscan_kernel(AssocArgs a, const double* __restrict__ prep,
                                                        const double* __restrict__ wts,
                                                        double* __restrict__ alt_lml) {
    __shared__ double red[256];
    __shared__ double hgW[CMAX];
    __shared__ double hsc[2];
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int c = a.c, r = a.r;
    const double inv_d = 1.0 / a.delta0;
    const double* __restrict__ tg = a.T + (long)b * a.ldT;
    for (int u = 0; u < c + 2; u++) {
        // u < c: W_u ; u == c: g ; u == c+1: y
        const double* tu = u < c ? a.tW + (long)u * a.ldW : (u == c ? tg : a.ty);
        double sw = 0.0, s1 = 0.0;
        for (int j = tid; j < r; j += 256) {
            const double p = tg[j] * tu[j];
            sw += p * wts[j];
            s1 += p;
        }
        sw = block_sum(sw, red);
        s1 = block_sum(s1, red);
        if (tid == 0) {
            double plain;
            if (u < c) plain = a.gW[(long)b * a.ld_gW + u];
            else if (u == c) plain = a.gg[b];
            else plain = a.gy[b];
            const double k = sw + (plain - s1) * inv_d;
            if (u < c) hgW[u] = k;
            else hsc[u - c] = k;
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double n = (double)a.n;
        const double rss0 = prep[0], logdetK = prep[1];
        const double* zy = prep + 8;
        const double* L = prep + 8 + CMAX;
        double z[CMAX];
        double zz = 0.0, zzy = 0.0;
        for (int i = 0; i < c; i++) {
            double s = hgW[i];
            for (int k = 0; k < i; k++) s -= L[i * CMAX + k] * z[k];
            s /= L[i * CMAX + i];
            z[i] = s;
            zz += s * s;
            zzy += s * zy[i];
        }
        const double schur = hsc[0] - zz;   // g'K^-1g - g'K^-1W (W'K^-1W)^-1 W'K^-1g
        const double num = hsc[1] - zzy;    // g'K^-1 y after removing W
        double rss = rss0;
        if (schur > 1e-12 * hsc[0]) rss -= num * num / schur;  // else g in span(W): lstsq drops it
        const double s = fmax(rss / n, EPS_SMALL);
        alt_lml[b] = prep[3] != 0.0 ? -0.5 * (n * LOG2PI + n + n * log(s) + logdetK) : NAN;
    }
}

// lrt_pvalues (_cellregmap.py:443-469) with dof = 1: sf(x) = erfc(sqrt(x / 2))
__global__ void lrt_kernel(const double* __restrict__ alt_lml, double null_lml, int count,
                           double* __restrict__ pv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    double lrs = -2.0 * null_lml + 2.0 * alt_lml[i];
    if (lrs < DBL_TINY) lrs = DBL_TINY;      // clip(lrs, super_tiny, inf); NaN stays NaN
    double p = erfc(sqrt(0.5 * lrs));
    if (p < DBL_TINY) p = DBL_TINY;
    if (p > 1.0 - DBL_EPS) p = 1.0 - DBL_EPS;
    pv[i] = p;
}

__global__ void gather_trial_lml(const NullFitTrial* __restrict__ trial, int count, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = trial[i].lml;
}

}  // namespace

int launch_fastscan_prep(hipStream_t st, const AssocArgs& a, double* prep, double* wts) {
    if (a.c > CMAX) {
        set_error("association: %d covariate columns (supported up to %d)", a.c, CMAX);
        return CRM_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(fastscan_prep_kernel, dim3(1), dim3(256), 0, st, a, prep, wts);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}
size_t fastscan_prep_doubles() { return 8 + CMAX + (size_t)CMAX * CMAX; }

int launch_fastscan(hipStream_t st, const AssocArgs& a, const double* prep, const double* wts,
                    int variants, double* alt_lml) {
    if (variants <= 0) return CRM_OK;
    hipLaunchKernelGGL(fastscan_kernel, dim3(variants), dim3(256), 0, st, a, prep, wts, alt_lml);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

int launch_lrt(hipStream_t st, const double* alt_lml, double null_lml, int count, double* pv) {
    if (count <= 0) return CRM_OK;
    hipLaunchKernelGGL(lrt_kernel, dim3((count + 255) / 256), dim3(256), 0, st, alt_lml, null_lml, count, pv);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

int launch_gather_trial_lml(hipStream_t st, const NullFitTrial* trial, int count, double* out) {
    if (count <= 0) return CRM_OK;
    hipLaunchKernelGGL(gather_trial_lml, dim3((count + 255) / 256), dim3(256), 0, st, trial, count, out);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

}  // namespace crm
