// Blocked Householder tridiagonalisation (lower), batched over the grid points of a background:
// A = Q T Q' with Q = H_0 H_1 ... H_{dim-2}, H_j = I - tau_j v_j v_j' (LAPACK dsytrd / dlatrd conventions).
//
// Per column j (panel position i): three small launches that cover every matrix of the batch
//   trd_column : a_j <- a_j - V W[j,:]' - W V[j,:]'   (pending updates of the panel), Householder vector
//   trd_symv   : p = A_trailing v over the LOWER triangle in 64 x 64 tiles -- each tile is read once and used
//                for the row part (in registers) and for the column part (partial vectors), so the HBM
//                stream is half the matrix per column; all CUs take part (block rows x batch)
//   trd_w      : w = tau (p - V (W'v) - W (V'v)),  w += -1/2 tau (w'v) v
// and per panel of TRD_NB columns one contraction  A_trailing -= [V W]' [W V]'  on the FP64 matrix pipe
// (gemm_tn*.hip with GEMM_SUBTRACT; the panel vectors are kept as ROWS so that this is a plain X'Y).
// Both triangles of A are updated; column j is read from the upper one (a contiguous row).
#include "eigh.h"

namespace crm {
namespace {

struct TrdArgs {
    double* A;      // [batch] slabs
    double* Vt;     // [batch] slabs, row j = v_j
    double* PV;     // [batch][2 nb x ld]  rows 0..nb-1: v_i, rows nb..2nb-1: w_i
    double* PW;     // [batch][2 nb x ld]  rows 0..nb-1: w_i, rows nb..2nb-1: v_i
    double* rowpart;  // [batch][ld]
    double* colpart;  // [batch][tiles x ld]
    double *d, *e, *tau;  // [batch][ld]
    long slab, ld, dim, dimp;
    long colpart_stride;
    int nb;
};

__device__ inline double block_sum_1024(double v, double* red) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = 0.0;
    const int nw = blockDim.x >> 6;
    for (int k = 0; k < nw; k++) s += red[k];  // same order on every thread
    return s;
}

// one workgroup per matrix
__global__ __launch_bounds__(1024) void trd_column_kernel(TrdArgs a, long j, int i) {
    __shared__ double cw[TRD_NB], cv[TRD_NB], red[16], bc[2];
    const int b = blockIdx.x, tid = threadIdx.x;
    const long ld = a.ld, dim = a.dim;
    double* A = a.A + (long)b * a.slab;
    double* Vrow = a.Vt + (long)b * a.slab + j * ld;
    double* PV = a.PV + (long)b * 2 * a.nb * ld;
    double* PW = a.PW + (long)b * 2 * a.nb * ld;
    const int nb = a.nb;
    if (tid < i) {
        cv[tid] = PV[(long)tid * ld + j];          // V[j, l]
        cw[tid] = PV[(long)(nb + tid) * ld + j];   // W[j, l]
    }
    __syncthreads();
    double ss = 0.0;
    for (long r = j + tid; r < dim; r += blockDim.x) {
        double x = A[j * ld + r];
        for (int l = 0; l < i; l++) x -= PV[(long)l * ld + r] * cw[l] + PV[(long)(nb + l) * ld + r] * cv[l];
        Vrow[r] = x;
        if (r >= j + 2) ss += x * x;
    }
    const double xnorm2 = block_sum_1024(ss, red);
    __syncthreads();
    if (tid == 0) {
        const double dj = Vrow[j];
        a.d[(long)b * ld + j] = dj;
        double tau = 0.0, scale = 0.0, beta = 0.0;
        if (j + 1 < dim) {
            const double alpha = Vrow[j + 1];
            beta = alpha;
            if (xnorm2 > 0.0) {
                const double nrm = sqrt(alpha * alpha + xnorm2);
                beta = alpha >= 0.0 ? -nrm : nrm;
                tau = (beta - alpha) / beta;
                scale = 1.0 / (alpha - beta);
            }
            a.e[(long)b * ld + j] = beta;
            a.tau[(long)b * ld + j] = tau;
        }
        bc[0] = tau;
        bc[1] = scale;
    }
    __syncthreads();
    if (j + 1 >= dim) return;
    const double scale = bc[1];
    for (long r = j + tid; r < dim; r += blockDim.x) {
        double v = 0.0;
        if (r == j + 1) v = 1.0;
        else if (r > j + 1) v = Vrow[r] * scale;   // (tau = 0: scale = 0, v = e_1, H = I)
        Vrow[r] = v;
        PV[(long)i * ld + r] = v;
        PW[(long)(nb + i) * ld + r] = v;
    }
}

// grid (block rows of the trailing part, batch); 256 threads.  Block row I (absolute 64-row blocks) takes the
// tiles (I, J), J0 <= J <= I.  x = v_j (zero up to j, so the columns left of the trailing part drop out).
__global__ __launch_bounds__(256) void trd_symv_kernel(TrdArgs a, long j, int i, int I0) {
    __shared__ double red[4][64];
    const int b = blockIdx.y;
    const int I = I0 + (int)(gridDim.x - 1 - blockIdx.x);  // longest block rows first
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rr = lane >> 3, cp = lane & 7;
    const long ld = a.ld;
    const double* __restrict__ A = a.A + (long)b * a.slab;
    const double* __restrict__ x = a.PV + (long)b * 2 * a.nb * ld + (long)i * ld;  // v_j as a row
    double* __restrict__ colpart = a.colpart + (long)b * a.colpart_stride + (long)(I - I0) * ld;
    typedef double v2d __attribute__((ext_vector_type(2)));
    const long row0 = (long)I * 64;
    // x over this block row (8 entries per lane), loaded once
    double xi[8];
#pragma unroll
    for (int q = 0; q < 8; q++) xi[q] = x[row0 + 8 * q + rr];
    double racc[8];
#pragma unroll
    for (int q = 0; q < 8; q++) racc[q] = 0.0;
    const int ccol = 16 * wave + 2 * cp;  // this lane's column pair inside a tile
    const double* __restrict__ base = A + (row0 + rr) * ld + ccol;
    v2d cur[8], nxt[8];
    auto load = [&](int J, v2d (&t)[8]) __attribute__((always_inline)) {
        const double* p = base + (long)J * 64;
#pragma unroll
        for (int q = 0; q < 8; q++) t[q] = *reinterpret_cast<const v2d*>(p + (long)(8 * q) * ld);
    };
    load(I0, cur);
    for (int J = I0; J <= I; J++) {
        if (J < I) load(J + 1, nxt);
        const double x0 = x[(long)J * 64 + ccol], x1 = x[(long)J * 64 + ccol + 1];
        double p0 = 0.0, p1 = 0.0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            racc[q] += cur[q][0] * x0 + cur[q][1] * x1;
            p0 += cur[q][0] * xi[q];
            p1 += cur[q][1] * xi[q];
        }
        if (J < I) {  // column part of an off-diagonal tile: sum over the 64 rows (8 in-lane x 8 lanes apart)
#pragma unroll
            for (int off = 8; off < 64; off <<= 1) {
                p0 += __shfl_xor(p0, off, 64);
                p1 += __shfl_xor(p1, off, 64);
            }
            if (rr == 0) {
                colpart[(long)J * 64 + ccol] = p0;
                colpart[(long)J * 64 + ccol + 1] = p1;
            }
#pragma unroll
            for (int q = 0; q < 8; q++) cur[q] = nxt[q];
        }
    }
    // row part: sum over this wavefront's 16 columns (8 lanes), then over the four wavefronts
#pragma unroll
    for (int q = 0; q < 8; q++) {
        double v = racc[q];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        if (cp == 0) red[wave][8 * q + rr] = v;
    }
    __syncthreads();
    if (tid < 64) a.rowpart[(long)b * ld + row0 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// one workgroup per matrix
__global__ __launch_bounds__(1024) void trd_w_kernel(TrdArgs a, long j, int i, int I0, int Iend) {
    __shared__ double t1[TRD_NB], t2[TRD_NB], red[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long ld = a.ld, dim = a.dim, o = j + 1;
    const int nb = a.nb;
    double* PV = a.PV + (long)b * 2 * nb * ld;
    double* PW = a.PW + (long)b * 2 * nb * ld;
    const double* __restrict__ v = PV + (long)i * ld;
    double* __restrict__ w = PV + (long)(nb + i) * ld;   // w_i, built in place
    double* __restrict__ w2 = PW + (long)i * ld;
    const double tau = a.tau[(long)b * ld + j];
    if (tau == 0.0) {  // H = I: no contribution to the panel update
        for (long r = o + tid; r < dim; r += blockDim.x) { w[r] = 0.0; w2[r] = 0.0; }
        return;
    }
    const double* __restrict__ rowpart = a.rowpart + (long)b * ld;
    const double* __restrict__ colpart = a.colpart + (long)b * a.colpart_stride;
    // p = A v from the row part and the column parts of the block rows below
    for (long r = o + tid; r < dim; r += blockDim.x) {
        double s = rowpart[r];
        const int Jc = (int)(r >> 6);
        for (int I = Jc + 1; I < Iend; I++) s += colpart[(long)(I - I0) * ld + r];
        w[r] = s;
    }
    // t1[l] = W_l' v, t2[l] = V_l' v for the columns already in the panel (one wavefront per l)
    for (int l = wave; l < i; l += 16) {
        const double* __restrict__ Wl = PV + (long)(nb + l) * ld;
        const double* __restrict__ Vl = PV + (long)l * ld;
        double s1 = 0.0, s2 = 0.0;
        for (long r = o + lane; r < dim; r += 64) {
            const double vr = v[r];
            s1 += Wl[r] * vr;
            s2 += Vl[r] * vr;
        }
        for (int off = 32; off > 0; off >>= 1) {
            s1 += __shfl_xor(s1, off, 64);
            s2 += __shfl_xor(s2, off, 64);
        }
        if (lane == 0) { t1[l] = s1; t2[l] = s2; }
    }
    __syncthreads();
    double dot = 0.0;
    for (long r = o + tid; r < dim; r += blockDim.x) {
        double s = w[r];
        for (int l = 0; l < i; l++) s -= PV[(long)l * ld + r] * t1[l] + PV[(long)(nb + l) * ld + r] * t2[l];
        s *= tau;
        w[r] = s;
        dot += s * v[r];
    }
    dot = block_sum_1024(dot, red);
    const double alpha = -0.5 * tau * dot;
    for (long r = o + tid; r < dim; r += blockDim.x) {
        const double s = w[r] + alpha * v[r];
        w[r] = s;
        w2[r] = s;
    }
}

}  // namespace

int eigh_tridiagonalise(crm_ctx* ctx, EighWork& w) {
    hipStream_t st = ctx->stream;
    const long dim = w.dim, ld = w.ld, dimp = w.dimp;
    const int B = w.batch, nb = TRD_NB;
    const int tiles = (int)(dimp / 64);
    // carve the small buffer
    const size_t panel = (size_t)2 * nb * ld;
    const int npanels = (int)((dim + nb - 1) / nb);
    const size_t need = sizeof(double) * ((size_t)B * (2 * panel + ld + (size_t)tiles * ld)) +
                        sizeof(GemmProblem) * (size_t)B * npanels;
    CRM_TRY(w.small.ensure(need));
    TrdArgs a{};
    a.A = w.A.as<double>();
    a.Vt = w.Vt.as<double>();
    a.PV = w.small.as<double>();
    a.PW = a.PV + (size_t)B * panel;
    a.rowpart = a.PW + (size_t)B * panel;
    a.colpart = a.rowpart + (size_t)B * ld;
    a.colpart_stride = (long)tiles * ld;
    GemmProblem* d_probs = reinterpret_cast<GemmProblem*>(a.colpart + (size_t)B * tiles * ld);
    a.d = w.d.as<double>(); a.e = w.e.as<double>(); a.tau = w.tau.as<double>();
    a.slab = w.slab; a.ld = ld; a.dim = dim; a.dimp = dimp; a.nb = nb;
    CRM_HIP(hipMemsetAsync(w.Vt.ptr, 0, sizeof(double) * (size_t)B * w.slab, st));
    CRM_HIP(hipMemsetAsync(w.d.ptr, 0, sizeof(double) * (size_t)B * ld, st));
    CRM_HIP(hipMemsetAsync(w.e.ptr, 0, sizeof(double) * (size_t)B * ld, st));
    CRM_HIP(hipMemsetAsync(w.tau.ptr, 0, sizeof(double) * (size_t)B * ld, st));
    // the trailing updates of all panels:  A[o2:, o2:] -= V W' + W V'  ==  [V; W]' [W; V]  as X'Y over the
    // 2 nb panel rows (one record per panel and matrix, uploaded once)
    std::vector<GemmProblem> probs((size_t)B * npanels);
    for (int pi = 0; pi < npanels; pi++) {
        const long o2 = std::min<long>((long)(pi + 1) * nb, dim);
        for (int b = 0; b < B; b++) {
            GemmProblem p{};
            p.X = a.PV + (size_t)b * panel + o2; p.ldx = ld;
            p.Y = a.PW + (size_t)b * panel + o2; p.ldy = ld;
            p.C = a.A + (size_t)b * w.slab + o2 * ld + o2; p.ldc = ld;
            p.M = (int)(dim - o2); p.N = (int)(dim - o2);
            p.flags = GEMM_SUBTRACT;
            probs[(size_t)pi * B + b] = p;
        }
    }
    CRM_HIP(hipMemcpyAsync(d_probs, probs.data(), sizeof(GemmProblem) * probs.size(), hipMemcpyHostToDevice, st));
    CRM_HIP(hipStreamSynchronize(st));
    for (long j0 = 0; j0 < dim; j0 += nb) {
        const int cols = (int)std::min<long>(nb, dim - j0);
        CRM_HIP(hipMemsetAsync(a.PV, 0, sizeof(double) * (size_t)B * 2 * panel, st));  // PV and PW are adjacent
        for (int i = 0; i < cols; i++) {
            const long j = j0 + i;
            hipLaunchKernelGGL(trd_column_kernel, dim3(B), dim3(1024), 0, st, a, j, i);
            if (j + 1 >= dim) break;
            const int I0 = (int)((j + 1) / 64), Iend = (int)((dim + 63) / 64);
            hipLaunchKernelGGL(trd_symv_kernel, dim3(Iend - I0, B), dim3(256), 0, st, a, j, i, I0);
            hipLaunchKernelGGL(trd_w_kernel, dim3(B), dim3(1024), 0, st, a, j, i, I0, Iend);
        }
        CRM_HIP(hipGetLastError());
        const long o2 = j0 + cols;
        if (o2 >= dim) break;
        CRM_TRY(launch_gemm_tn(ctx, d_probs + (size_t)(j0 / nb) * B, B, (int)(dim - o2), (int)(dim - o2), 2 * nb, false,
                               0, 1, 0));
    }
    return CRM_OK;
}

}  // namespace crm
