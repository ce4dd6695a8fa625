// Blocked Householder tridiagonalisation (lower), batched over the grid points of a background:
// A = Q T Q' with Q = H_0 H_1 ... H_{dim-2}, H_j = I - tau_j v_j v_j' (LAPACK dsytrd / dlatrd conventions).
//
// Per column j (panel position i) three launches that cover every matrix of the batch:
//   trd_column : Householder vector v_j of the updated column (one workgroup per matrix; the pending updates
//                of the panel have already been applied by the previous column's trd_w, except the newest)
//   trd_symv   : p = A_trailing v over the LOWER triangle in 64 x 64 tiles -- each tile is read once and used
//                for the row part (in registers) and for the column part (partial vectors), so the HBM
//                stream is half the matrix per column; all CUs take part (block rows x batch).  Extra
//                workgroups of the same launch take the dot products W_l'v, V_l'v of the panel columns.
//   trd_w      : w' = tau (p - V (W'v) - W (V'v)) in chunks of 512 rows over many workgroups, which also
//                prepare the next column (a_{j+1} minus the panel's pending updates) from the same loads.
// The last step of dlatrd, w = w' + alpha v with alpha = -1/2 tau w''v, needs a dot product over the whole
// vector; instead of a grid-wide wait the panel carries (w', alpha) -- every use of w is linear in it, so
// alpha enters the coefficients (2 alpha V_l[j] etc.) -- and w is formed once per panel, before the rank-2k
// update of the trailing matrix  A -= [V; W]' [W; V]  on the FP64 matrix pipe (gemm_tn*.hip, GEMM_SUBTRACT;
// the panel vectors are kept as ROWS so that this is a plain X'Y).  Both triangles of A are updated; column j
// is read from the upper one (a contiguous row).
#include "eigh.h"

namespace crm {
namespace {

constexpr int TRD_CH = 64;    // rows per workgroup of trd_w (four wavefronts share the sums over the panel columns)
constexpr int TRD_FCH = 512;  // rows per workgroup of trd_finalize
constexpr int TRD_SEG = 8;    // tiles per workgroup of trd_symv (a block row is cut into segments of this many)

struct TrdArgs {
    double* A;      // [batch] slabs
    double* Vt;     // [batch] slabs, row j = v_j
    double* PV;     // [batch][2 nb x ld]  rows 0..nb-1: v_l, rows nb..2nb-1: w'_l (w_l once the panel is final)
    double* PW;     // [batch][2 nb x ld]  rows 0..nb-1: w_l (panel final), rows nb..2nb-1: v_l
    double* rowpart;  // [batch][nseg x ld]  row parts per segment of a block row
    double* colpart;  // [batch][tiles x ld]
    double* xnext;    // [batch][ld]  the next column with the pending updates of all but the newest panel column
    double* alpha;    // [batch][nb]
    double* tdots;    // [batch][2 nb]  W'_l'v, V_l'v
    double* dotpart;  // [batch][maxch] partial sums of w''v per chunk
    double *d, *e, *tau;  // [batch][ld]
    long slab, ld, dim, dimp;
    long colpart_stride, rowpart_stride;
    int nb, maxch;
};

__device__ inline double block_sum(double v, double* red) {
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
    __shared__ double red[16], bc[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const long ld = a.ld, dim = a.dim;
    const int nb = a.nb;
    const double* A = a.A + (long)b * a.slab;
    double* Vrow = a.Vt + (long)b * a.slab + j * ld;
    double* PV = a.PV + (long)b * 2 * nb * ld;
    double* PW = a.PW + (long)b * 2 * nb * ld;
    const double* xnext = a.xnext + (long)b * ld;
    double dsum = 0.0;
    if (i > 0) {
        // alpha of the previous column from the chunk sums of its trd_w
        const int nch = (int)((dim - j + TRD_CH - 1) / TRD_CH);
        for (int c = tid; c < nch; c += blockDim.x) dsum += a.dotpart[(long)b * a.maxch + c];
        dsum = block_sum(dsum, red);
        __syncthreads();
    }
    if (i > 0 && tid == 0) {
        const double al = -0.5 * a.tau[(long)b * ld + j - 1] * dsum;
        a.alpha[(long)b * nb + i - 1] = al;
        const double cv = PV[(long)(i - 1) * ld + j];                       // V_{i-1}[j]
        bc[2] = PV[(long)(nb + i - 1) * ld + j] + 2.0 * al * cv;            // W_{i-1}[j] + alpha V_{i-1}[j] (twice: both terms)
        bc[3] = cv;
    }
    __syncthreads();
    const double cw = i > 0 ? bc[2] : 0.0, cv = i > 0 ? bc[3] : 0.0;
    const double* Vp = PV + (long)(i > 0 ? i - 1 : 0) * ld;
    const double* Wp = PV + (long)(nb + (i > 0 ? i - 1 : 0)) * ld;
    double ss = 0.0;
    for (long r = j + tid; r < dim; r += blockDim.x) {
        const double x = i > 0 ? xnext[r] - Vp[r] * cw - Wp[r] * cv : A[j * ld + r];
        Vrow[r] = x;
        if (r >= j + 2) ss += x * x;
    }
    const double xnorm2 = block_sum(ss, red);
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

// grid (segments of block rows of the trailing part + i, batch); 256 threads.  Block row I (absolute 64-row
// blocks) has the tiles (I, J), I0 <= J <= I, cut into segments of TRD_SEG tiles -- one workgroup each, so that no
// workgroup streams more than 256 KB (the longest block row alone would otherwise set the pace of a launch).
// x = v_j (zero up to j, so the columns left of the trailing part drop out).
// Workgroups past the segments: W'_l'v and V_l'v for panel column l.
__global__ __launch_bounds__(256) void trd_symv_kernel(TrdArgs a, long j, int i, int I0, int nsegs) {
    __shared__ double red[4][64];
    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long ld = a.ld;
    const double* __restrict__ x = a.PV + (long)b * 2 * a.nb * ld + (long)i * ld;  // v_j as a row
    if ((int)blockIdx.x >= nsegs) {
        const int l = (int)blockIdx.x - nsegs;
        const double* __restrict__ Wl = a.PV + (long)b * 2 * a.nb * ld + (long)(a.nb + l) * ld;
        const double* __restrict__ Vl = a.PV + (long)b * 2 * a.nb * ld + (long)l * ld;
        double s1 = 0.0, s2 = 0.0;
        for (long r = j + 1 + tid; r < a.dim; r += 256) {
            const double vr = x[r];
            s1 += Wl[r] * vr;
            s2 += Vl[r] * vr;
        }
        s1 = block_sum(s1, &red[0][0]);
        s2 = block_sum(s2, &red[1][0]);
        if (tid == 0) {
            a.tdots[(long)b * 2 * a.nb + 2 * l] = s1;
            a.tdots[(long)b * 2 * a.nb + 2 * l + 1] = s2;
        }
        return;
    }
    // segment index -> (block row, segment): block rows 8a .. 8a + 7 (relative to I0) have a + 1 segments each,
    // 4 a (a + 1) segments lie before them
    int aa = 0;
    const int xid = (int)blockIdx.x;
    while (4 * (aa + 1) * (aa + 2) <= xid) aa++;
    const int rem = xid - 4 * aa * (aa + 1);
    const int Irel = TRD_SEG * aa + rem / (aa + 1), seg = rem % (aa + 1);
    const int I = I0 + Irel;
    const int Jb = I0 + seg * TRD_SEG, Je = min(Jb + TRD_SEG - 1, I);   // tiles Jb .. Je
    const int rr = lane >> 3, cp = lane & 7;
    const double* __restrict__ A = a.A + (long)b * a.slab;
    double* __restrict__ colpart = a.colpart + (long)b * a.colpart_stride + (long)Irel * ld;
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
    load(Jb, cur);
    for (int J = Jb; J <= Je; J++) {
        if (J < Je) load(J + 1, nxt);
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
        }
        if (J < Je) {
#pragma unroll
            for (int q = 0; q < 8; q++) cur[q] = nxt[q];
        }
    }
    // row part of this segment: sum over this wavefront's 16 columns (8 lanes), then over the four wavefronts
#pragma unroll
    for (int q = 0; q < 8; q++) {
        double v = racc[q];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        if (cp == 0) red[wave][8 * q + rr] = v;
    }
    __syncthreads();
    if (tid < 64)
        a.rowpart[(long)b * a.rowpart_stride + (long)seg * ld + row0 + tid] =
            (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// grid (chunks of TRD_CH = 64 rows from j + 1 on, batch); 256 threads: lane = row of the chunk, the four wavefronts
// take every fourth term of the sums over the block rows below and over the panel columns (short dependent
// chains of loads instead of one long one), combined through LDS in a fixed order
__global__ __launch_bounds__(256) void trd_w_kernel(TrdArgs a, long j, int i, int I0, int Iend, int next) {
    __shared__ double c1[TRD_NB], c2[TRD_NB], cwn[TRD_NB], cvn[TRD_NB], ps[4][64], px[4][64];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long ld = a.ld, dim = a.dim, o = j + 1;
    const int nb = a.nb;
    double* PV = a.PV + (long)b * 2 * nb * ld;
    const double* __restrict__ v = PV + (long)i * ld;
    double* __restrict__ w = PV + (long)(nb + i) * ld;
    const double tau = a.tau[(long)b * ld + j];
    if (tid < i) {
        const double al = a.alpha[(long)b * nb + tid];
        const double t1 = a.tdots[(long)b * 2 * nb + 2 * tid], t2 = a.tdots[(long)b * 2 * nb + 2 * tid + 1];
        c1[tid] = t1 + 2.0 * al * t2;    // W_l'v + alpha_l V_l'v, and the alpha_l V_l part of W_l[r]
        c2[tid] = t2;
        if (next) {
            const double vn = PV[(long)tid * ld + j + 1];
            cvn[tid] = vn;
            cwn[tid] = PV[(long)(nb + tid) * ld + j + 1] + 2.0 * al * vn;
        }
    }
    __syncthreads();
    const double* __restrict__ rowpart = a.rowpart + (long)b * a.rowpart_stride;
    const double* __restrict__ colpart = a.colpart + (long)b * a.colpart_stride;
    const long r = o + (long)blockIdx.x * TRD_CH + lane;
    const bool live = r < dim;
    double s = 0.0, xn = 0.0;
    if (live) {
        // p = A v: row parts of the segments of this row's block row, column parts of the block rows below
        const int Jc = (int)(r >> 6);
        const int nseg = (Jc - I0) / TRD_SEG + 1;
        for (int g = wave; g < nseg; g += 4) s += rowpart[(long)g * ld + r];
        for (int I = Jc + 1 + wave; I < Iend; I += 4) s += colpart[(long)(I - I0) * ld + r];
        for (int l = wave; l < i; l += 4) {
            const double vl = PV[(long)l * ld + r], wl = PV[(long)(nb + l) * ld + r];
            s -= vl * c1[l] + wl * c2[l];
            if (next) xn -= vl * cwn[l] + wl * cvn[l];
        }
    }
    ps[wave][lane] = s;
    px[wave][lane] = xn;
    __syncthreads();
    if (wave != 0) return;
    double dot = 0.0;
    if (live) {
        s = (ps[0][lane] + ps[1][lane]) + (ps[2][lane] + ps[3][lane]);
        s = tau != 0.0 ? tau * s : 0.0;   // (H = I: no contribution to the panel update)
        w[r] = s;
        if (next) {
            xn = (px[0][lane] + px[1][lane]) + (px[2][lane] + px[3][lane]);
            a.xnext[(long)b * ld + r] = a.A[(long)b * a.slab + (j + 1) * ld + r] + xn;
        }
        dot = s * v[r];
    }
    for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
    if (lane == 0) a.dotpart[(long)b * a.maxch + blockIdx.x] = dot;
}

// panel complete: w_l = w'_l + alpha_l v_l into both operand panels of the rank-2k update
__global__ __launch_bounds__(256) void trd_finalize_kernel(TrdArgs a, long j_last, int cols) {
    __shared__ double al[TRD_NB];
    const int b = blockIdx.y, tid = threadIdx.x;
    const long ld = a.ld, dim = a.dim;
    const int nb = a.nb;
    if (tid < cols - 1) al[tid] = a.alpha[(long)b * nb + tid];
    if (tid == cols - 1) {   // the last column's alpha has not been formed yet (same order of summation everywhere)
        const int nch = (int)((dim - (j_last + 1) + TRD_CH - 1) / TRD_CH);
        double s = 0.0;
        for (int c = 0; c < nch; c++) s += a.dotpart[(long)b * a.maxch + c];
        al[tid] = -0.5 * a.tau[(long)b * ld + j_last] * s;
    }
    __syncthreads();
    double* PV = a.PV + (long)b * 2 * nb * ld;
    double* PW = a.PW + (long)b * 2 * nb * ld;
    for (int q = 0; q < TRD_FCH / 256; q++) {
        const long r = (long)blockIdx.x * TRD_FCH + q * 256 + tid;
        if (r >= dim) continue;
        for (int l = 0; l < cols; l++) {
            const double wv = PV[(long)(nb + l) * ld + r] + al[l] * PV[(long)l * ld + r];
            PV[(long)(nb + l) * ld + r] = wv;
            PW[(long)l * ld + r] = wv;
        }
    }
}

}  // namespace

int eigh_tridiagonalise(crm_ctx* ctx, EighWork& w) {
    hipStream_t st = ctx->stream;
    const long dim = w.dim, ld = w.ld, dimp = w.dimp;
    const int B = w.batch, nb = TRD_NB;
    const int tiles = (int)(dimp / 64);
    const int maxch = (int)((dimp + TRD_CH - 1) / TRD_CH);
    // carve the small buffer
    const size_t panel = (size_t)2 * nb * ld;
    const int npanels = (int)((dim + nb - 1) / nb);
    const int nsegmax = tiles / TRD_SEG + 1;
    const size_t ndouble = (size_t)B * (2 * panel + (size_t)(1 + nsegmax) * ld + (size_t)tiles * ld + 3 * nb + maxch);
    const size_t need = sizeof(double) * ndouble + sizeof(GemmProblem) * (size_t)B * npanels + 64;
    CRM_TRY(w.small.ensure(need));
    TrdArgs a{};
    a.A = w.A.as<double>();
    a.Vt = w.Vt.as<double>();
    a.PV = w.small.as<double>();
    a.PW = a.PV + (size_t)B * panel;
    a.rowpart = a.PW + (size_t)B * panel;
    a.rowpart_stride = (long)nsegmax * ld;
    a.xnext = a.rowpart + (size_t)B * nsegmax * ld;
    a.colpart = a.xnext + (size_t)B * ld;
    a.colpart_stride = (long)tiles * ld;
    a.alpha = a.colpart + (size_t)B * tiles * ld;
    a.tdots = a.alpha + (size_t)B * nb;
    a.dotpart = a.tdots + (size_t)B * 2 * nb;
    GemmProblem* d_probs = reinterpret_cast<GemmProblem*>((reinterpret_cast<uintptr_t>(a.dotpart + (size_t)B * maxch) + 15) & ~(uintptr_t)15);
    a.d = w.d.as<double>(); a.e = w.e.as<double>(); a.tau = w.tau.as<double>();
    a.slab = w.slab; a.ld = ld; a.dim = dim; a.dimp = dimp; a.nb = nb; a.maxch = maxch;
    CRM_HIP(hipMemsetAsync(w.small.ptr, 0, sizeof(double) * ndouble, st));
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
        long j_last = j0;
        for (int i = 0; i < cols; i++) {
            const long j = j0 + i;
            hipLaunchKernelGGL(trd_column_kernel, dim3(B), dim3(1024), 0, st, a, j, i);
            if (j + 1 >= dim) break;
            j_last = j;
            const int I0 = (int)((j + 1) / 64), Iend = (int)((dim + 63) / 64);
            int nsegs = 0;   // segments of all block rows: (t + 8) / 8 for the block row with t + 1 tiles
            for (int t = 0; t < Iend - I0; t++) nsegs += t / TRD_SEG + 1;
            hipLaunchKernelGGL(trd_symv_kernel, dim3(nsegs + i, B), dim3(256), 0, st, a, j, i, I0, nsegs);
            const int nch = (int)((dim - (j + 1) + TRD_CH - 1) / TRD_CH);
            const int next = (i + 1 < cols) ? 1 : 0;
            hipLaunchKernelGGL(trd_w_kernel, dim3(nch, B), dim3(256), 0, st, a, j, i, I0, Iend, next);
        }
        CRM_HIP(hipGetLastError());
        const long o2 = j0 + cols;
        if (o2 >= dim) break;
        hipLaunchKernelGGL(trd_finalize_kernel, dim3((unsigned)((dim + TRD_FCH - 1) / TRD_FCH), B), dim3(256), 0, st, a, j_last,
                           cols);
        CRM_TRY(launch_gemm_tn(ctx, d_probs + (size_t)(j0 / nb) * B, B, (int)(dim - o2), (int)(dim - o2), 2 * nb, false,
                               0, 1, 0));
    }
    return CRM_OK;
}

}  // namespace crm
