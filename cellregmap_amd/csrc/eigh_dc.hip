// Divide & conquer for the symmetric tridiagonal eigenproblems of a batch (Cuppen / Gu-Eisenstat, the
// scheme of LAPACK dstedc: dlaed0-dlaed4), eigenvectors kept as ROWS so that every merge product is the
// contraction kernel's native shape  Qt_new = U' Qt_old  (sum over the old pole index = rows of both).
//
//   leaves   (<= 32)     parallel-order Jacobi in LDS, one wavefront per leaf
//   per level            z = last / first components of the two halves (kernel) -> host: sort, deflation
//                        (dlaed2: tiny |rho z_i| and nearly equal poles, O(k) per merge) -> kernels: Givens
//                        rotations of the deflated pairs, row gather, secular equation (one root per thread,
//                        safeguarded two-pole rational iteration around the nearer pole, so d_i - lambda_j is
//                        known to full relative accuracy), Loewner re-computation of the update vector
//                        (orthogonality of the eigenvectors to working precision whatever the accuracy of the
//                        roots), normalised eigenvector block U -> merge products on the FP64 matrix pipe.
// The tree is the same for every matrix of the batch (they share `dim`); deflation differs per matrix.
#include <algorithm>
#include <cmath>
#include <numeric>

#include "eigh.h"

namespace crm {
namespace {

constexpr double DC_EPS = 2.220446049250313e-16;

struct DcNode { int s, m, t; };   // merge of [s, m) and [m, t); m == t: block passed through unchanged

struct DcDesc {   // one per (matrix, node) of a level
    int s, m, t, k;          // k poles survive the deflation
    int rot_first, rot_count;
    double rho;              // of the normalised problem  D + rho w w', |w| = 1
};

struct DcRot { int a, b; double c, s; };  // rows (absolute); x' = c x + s y, y' = c y - s x

// ---- leaves ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void dc_leaf_kernel(const double* __restrict__ dcut, const double* __restrict__ e,
                                                     long ld, const int* __restrict__ leaf_s,
                                                     const int* __restrict__ leaf_t, double* __restrict__ lam,
                                                     double* __restrict__ Qt, long slab) {
    __shared__ double A[DC_LEAF][DC_LEAF + 1], V[DC_LEAF][DC_LEAF + 1];
    __shared__ double cs[DC_LEAF / 2], sn[DC_LEAF / 2];
    __shared__ int pp[DC_LEAF / 2], qq[DC_LEAF / 2];
    const int b = blockIdx.y, lane = threadIdx.x;
    const int s = leaf_s[blockIdx.x], t = leaf_t[blockIdx.x];
    const int n = t - s, m = (n + 1) & ~1;   // even order for the tournament (a padded index stays isolated)
    const double* d = dcut + (long)b * ld;
    const double* eb = e + (long)b * ld;
    for (int idx = lane; idx < DC_LEAF * DC_LEAF; idx += 64) {
        const int r = idx / DC_LEAF, c = idx - r * DC_LEAF;
        double v = 0.0;
        if (r < n && c < n) {
            if (r == c) v = d[s + r];
            else if (r == c + 1) v = eb[s + c];
            else if (c == r + 1) v = eb[s + r];
        }
        A[r][c] = v;
        V[r][c] = r == c ? 1.0 : 0.0;
    }
    __syncthreads();
    for (int sweep = 0; sweep < 40; sweep++) {
        double off = 0.0, dia = 0.0;
        for (int idx = lane; idx < m * m; idx += 64) {
            const int r = idx / m, c = idx - r * m;
            const double v = A[r][c];
            if (r == c) dia += v * v; else off += v * v;
        }
        for (int o = 32; o > 0; o >>= 1) { off += __shfl_xor(off, o, 64); dia += __shfl_xor(dia, o, 64); }
        if (off <= 1e-300 || off <= (1e-3 * DC_EPS) * (1e-3 * DC_EPS) * dia) break;
        for (int round = 0; round < m - 1; round++) {
            if (lane < m / 2) {
                int p, q;
                if (lane == 0) { p = m - 1; q = round; }
                else { p = (round + lane) % (m - 1); q = (round - lane + (m - 1)) % (m - 1); }
                if (p > q) { const int tmp = p; p = q; q = tmp; }
                const double apq = A[p][q];
                double c = 1.0, sv = 0.0;
                if (fabs(apq) > 1e-300) {
                    const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                    const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(1.0 + theta * theta));
                    c = 1.0 / sqrt(1.0 + tt * tt);
                    sv = tt * c;
                }
                pp[lane] = p; qq[lane] = q; cs[lane] = c; sn[lane] = sv;
            }
            __syncthreads();
            // A <- A J  (columns p, q of every row)
            for (int idx = lane; idx < m * (m / 2); idx += 64) {
                const int r = idx / (m / 2), k = idx - r * (m / 2);
                const int p = pp[k], q = qq[k];
                const double c = cs[k], sv = sn[k];
                const double x = A[r][p], y = A[r][q];
                A[r][p] = c * x - sv * y;
                A[r][q] = sv * x + c * y;
            }
            __syncthreads();
            // A <- J' A, V <- J' V  (rows p, q)
            for (int idx = lane; idx < m * (m / 2); idx += 64) {
                const int col = idx / (m / 2), k = idx - col * (m / 2);
                const int p = pp[k], q = qq[k];
                const double c = cs[k], sv = sn[k];
                double x = A[p][col], y = A[q][col];
                A[p][col] = c * x - sv * y;
                A[q][col] = sv * x + c * y;
                x = V[p][col]; y = V[q][col];
                V[p][col] = c * x - sv * y;
                V[q][col] = sv * x + c * y;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    double* L = lam + (long)b * ld;
    double* Q = Qt + (long)b * slab;
    for (int r = lane; r < n; r += 64) L[s + r] = A[r][r];
    for (int idx = lane; idx < n * n; idx += 64) {
        const int r = idx / n, c = idx - r * n;
        Q[(long)(s + r) * ld + s + c] = V[r][c];
    }
}

// ---- per level ----------------------------------------------------------------------------------------
// z[r] = Qt[r][m - 1] for the rows of the left half, Qt[r][m] for the right half
__global__ void dc_gather_z_kernel(const double* __restrict__ Qt, long slab, long ld, const int* __restrict__ row_node,
                                   const DcNode* __restrict__ nodes, long dim, double* __restrict__ z) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (r >= dim) return;
    const DcNode nd = nodes[row_node[r]];
    double v = 0.0;
    if (nd.m < nd.t) v = Qt[(long)b * slab + r * ld + (r < nd.m ? nd.m - 1 : nd.m)];
    z[(long)b * ld + r] = v;
}

// the Givens rotations that deflate nearly equal poles, in the order the host generated them
__global__ __launch_bounds__(256) void dc_rotate_kernel(double* __restrict__ Qt, long slab, long ld,
                                                        const DcDesc* __restrict__ desc, int nnodes,
                                                        const DcRot* __restrict__ rots) {
    const int b = blockIdx.y;
    const DcDesc D = desc[(long)b * nnodes + blockIdx.x];
    if (D.rot_count == 0) return;
    double* Q = Qt + (long)b * slab;
    for (int i = 0; i < D.rot_count; i++) {
        const DcRot R = rots[D.rot_first + i];
        double* x = Q + (long)R.a * ld;
        double* y = Q + (long)R.b * ld;
        for (int c = D.s + threadIdx.x; c < D.t; c += blockDim.x) {
            const double xv = x[c], yv = y[c];
            x[c] = R.c * xv + R.s * yv;
            y[c] = R.c * yv - R.s * xv;
        }
        __syncthreads();
    }
}

// destination row r of its node: the first k rows (poles that survive, in pole order) go to the operand
// buffer of the merge product, the others (deflated, or a block passed through) straight to the next buffer
__global__ __launch_bounds__(256) void dc_gather_rows_kernel(const double* __restrict__ cur, double* __restrict__ nxt,
                                                             double* __restrict__ ysrc, long slab, long ld,
                                                             const int* __restrict__ row_node,
                                                             const DcDesc* __restrict__ desc, int nnodes,
                                                             const int* __restrict__ src_row, long dim) {
    const long r = blockIdx.x;
    const int b = blockIdx.y;
    const DcDesc D = desc[(long)b * nnodes + row_node[r]];
    const long src = src_row[(long)b * ld + r];
    const double* in = cur + (long)b * slab + src * ld;
    double* out = ((r - D.s) < D.k ? ysrc : nxt) + (long)b * slab + r * ld;
    for (int c = D.s + threadIdx.x; c < D.t; c += blockDim.x) out[c] = in[c];
}

// One root per WAVEFRONT: the 64 lanes share the sums over the poles (terms i = lane, lane + 64, ...; butterfly
// totals, identical on all lanes, so the iteration is wave-uniform) -- a root of the top-level merge sums over
// ~5 000 poles per evaluation, and one thread per root left most of the chip idle behind chains of divisions.
// dl: poles of the node in ascending order at [s, s + k); w: normalised update vector.
// Root j lies in (dl_j, dl_j+1) (the last one in (dl_k-1, dl_k-1 + rho)); it is represented as
// dl[origin] + tau with the origin at the nearer pole.
__device__ inline double wave_total(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ __launch_bounds__(256) void dc_secular_kernel(const DcDesc* __restrict__ desc, int nnodes, long ld,
                                                         const double* __restrict__ dl_all,
                                                         const double* __restrict__ w_all, int* __restrict__ org_all,
                                                         double* __restrict__ tau_all, double* __restrict__ lam_next) {
    const int b = blockIdx.z;
    const DcDesc D = desc[(long)b * nnodes + blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int k = D.k;
    if (j >= k) return;
    const double* __restrict__ dl = dl_all + (long)b * ld + D.s;
    const double* __restrict__ w = w_all + (long)b * ld + D.s;
    const double rho = D.rho;
    const bool last = j == k - 1;
    int o;
    double lo, hi;
    if (last) {
        o = k - 1;
        lo = 0.0;
        hi = rho;   // |w| = 1
    } else {
        const double gap = dl[j + 1] - dl[j];
        const double mid = 0.5 * gap, dj = dl[j];
        double g = 0.0;
        for (int i = lane; i < k; i += 64) g += rho * w[i] * w[i] / ((dl[i] - dj) - mid);
        g = 1.0 + wave_total(g);
        if (g > 0.0) { o = j; lo = 0.0; hi = mid; }
        else { o = j + 1; lo = -mid; hi = 0.0; }
    }
    const double dorg = dl[o];
    double tau = 0.5 * (lo + hi);
    for (int it = 0; it < 400; it++) {
        double psi = 0.0, dpsi = 0.0, phi = 0.0, dphi = 0.0;
        for (int i = lane; i < k; i += 64) {
            const double rd = 1.0 / ((dl[i] - dorg) - tau);
            const double t = rho * w[i] * w[i] * rd;
            if (i <= j) { psi += t; dpsi += t * rd; }
            else { phi += t; dphi += t * rd; }
        }
        psi = wave_total(psi); dpsi = wave_total(dpsi); phi = wave_total(phi); dphi = wave_total(dphi);
        const double g = 1.0 + psi + phi;
        const double err = 8.0 * DC_EPS * (1.0 + fabs(psi) + fabs(phi)) + DC_EPS * fabs(tau) * (dpsi + dphi);
        if (fabs(g) <= err) break;
        if (g < 0.0) lo = tau; else hi = tau;
        if (hi - lo <= 2.0 * DC_EPS * fmax(fabs(lo), fabs(hi))) {
            tau = fabs(lo) > 0.0 ? lo : hi;
            if (tau == 0.0) tau = 0.5 * (lo + hi);
            break;
        }
        // two-pole model matching psi, psi', phi, phi' at tau; s = step from tau
        const double dj = (dl[j] - dorg) - tau;
        const double a = dpsi * dj * dj, sp = psi - dpsi * dj;
        double nw;
        if (last) {
            const double c = 1.0 + sp;
            nw = c > 0.0 ? tau + dj + a / c : INFINITY;
        } else {
            const double dj1 = (dl[j + 1] - dorg) - tau;
            const double bb = dphi * dj1 * dj1, sf = phi - dphi * dj1;
            const double c = 1.0 + sp + sf;
            const double A2 = c, B2 = -(c * (dj + dj1) + a + bb), C2 = c * dj * dj1 + a * dj1 + bb * dj;
            double s = NAN;
            if (A2 == 0.0) {
                if (B2 != 0.0) s = -C2 / B2;
            } else {
                const double disc = B2 * B2 - 4.0 * A2 * C2;
                if (disc >= 0.0) {
                    const double q = -0.5 * (B2 + copysign(sqrt(disc), B2));
                    const double r1 = q / A2, r2 = q != 0.0 ? C2 / q : NAN;
                    if (r1 > dj && r1 < dj1) s = r1;
                    else if (r2 > dj && r2 < dj1) s = r2;
                }
            }
            nw = tau + s;
        }
        if (!(nw > lo && nw < hi)) nw = 0.5 * (lo + hi);   // (also catches NaN / infinity)
        tau = nw;
    }
    if (lane == 0) {
        org_all[(long)b * ld + D.s + j] = o;
        tau_all[(long)b * ld + D.s + j] = tau;
        lam_next[(long)b * ld + D.s + j] = dorg + tau;
    }
}

// zhat_i^2 = prod_j (lam_j - dl_i) / (rho prod_{j != i} (dl_j - dl_i)), sign of w_i
__global__ __launch_bounds__(256) void dc_zhat_kernel(const DcDesc* __restrict__ desc, int nnodes, long ld,
                                                      const double* __restrict__ dl_all, const double* __restrict__ w_all,
                                                      const int* __restrict__ org_all, const double* __restrict__ tau_all,
                                                      double* __restrict__ zhat_all) {
    const int b = blockIdx.z;
    const DcDesc D = desc[(long)b * nnodes + blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = D.k;
    if (i >= k) return;
    const double* __restrict__ dl = dl_all + (long)b * ld + D.s;
    const int* __restrict__ org = org_all + (long)b * ld + D.s;
    const double* __restrict__ tau = tau_all + (long)b * ld + D.s;
    const double di = dl[i];
    double p = (dl[org[i]] - di) + tau[i];
    for (int j = 0; j < k; j++) {
        if (j == i) continue;
        p *= ((dl[org[j]] - di) + tau[j]) / (dl[j] - di);
    }
    const double wi = w_all[(long)b * ld + D.s + i];
    zhat_all[(long)b * ld + D.s + i] = copysign(sqrt(fabs(p) / D.rho), wi);
}

// U[i][j] = zhat_i / (dl_i - lam_j), columns normalised; rows [k, round_up(k, 16)) zero.  U lives in the
// A slab of its matrix at rows [s, s + k), leading dimension ld.
__global__ __launch_bounds__(256) void dc_vectors_kernel(const DcDesc* __restrict__ desc, int nnodes, long ld, long slab,
                                                         const double* __restrict__ dl_all, const int* __restrict__ org_all,
                                                         const double* __restrict__ tau_all,
                                                         const double* __restrict__ zhat_all, double* __restrict__ U_all) {
    const int b = blockIdx.z;
    const DcDesc D = desc[(long)b * nnodes + blockIdx.y];
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = D.k;
    if (j >= k) return;
    const double* __restrict__ dl = dl_all + (long)b * ld + D.s;
    const double* __restrict__ zh = zhat_all + (long)b * ld + D.s;
    const double dorg = dl[org_all[(long)b * ld + D.s + j]], tau = tau_all[(long)b * ld + D.s + j];
    double ss = 0.0;
    for (int i = 0; i < k; i++) {
        const double v = zh[i] / ((dl[i] - dorg) - tau);
        ss += v * v;
    }
    const double inv = 1.0 / sqrt(ss);
    double* U = U_all + (long)b * slab + (long)D.s * ld;
    for (int i = 0; i < k; i++) U[(long)i * ld + j] = zh[i] / ((dl[i] - dorg) - tau) * inv;
    const int kp = (k + 15) & ~15;
    for (int i = k; i < kp; i++) U[(long)i * ld + j] = 0.0;
}

// out[r] = in[order[r]] (rows, all dim columns)
__global__ __launch_bounds__(256) void dc_sort_rows_kernel(const double* __restrict__ in, double* __restrict__ out, long slab,
                                                           long ld, const int* __restrict__ order, long dim) {
    const long r = blockIdx.x;
    const int b = blockIdx.y;
    const double* src = in + (long)b * slab + (long)order[(long)b * ld + r] * ld;
    double* dst = out + (long)b * slab + r * ld;
    for (long c = threadIdx.x; c < dim; c += blockDim.x) dst[c] = src[c];
}

// ---- host: deflation of one merge (dlaed2) ---------------------------------------------------------------
struct MergePlan {
    int k = 0;
    double rho = 0.0;
    std::vector<int> nondefl, defl;     // local row indices (0 .. t - s) in destination order
    std::vector<double> dl, w, lam_defl;
    std::vector<DcRot> rots;            // local rows
};

void plan_merge(const double* lam, const double* z, int n1, int n, double beta, MergePlan& P) {
    // D + rho z z' with z = [last components of the left vectors ; +-first components of the right ones] / sqrt 2
    std::vector<double> D(lam, lam + n), zz(n);
    const double sgn = beta < 0.0 ? -1.0 : 1.0, r2 = 0.7071067811865476;
    for (int i = 0; i < n; i++) zz[i] = (i < n1 ? z[i] : sgn * z[i]) * r2;
    const double rho = 2.0 * fabs(beta);
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return D[a] < D[b]; });
    double dmax = 0.0, zmax = 0.0;
    for (int i = 0; i < n; i++) { dmax = std::max(dmax, fabs(D[i])); zmax = std::max(zmax, fabs(zz[i])); }
    const double tol = 8.0 * DC_EPS * std::max(dmax, zmax);
    P = MergePlan();
    std::vector<int> nd, df;   // positions in sorted order -> local rows through `order`
    if (rho * zmax <= tol) {
        for (int i = 0; i < n; i++) df.push_back(order[i]);
    } else {
        int pj = -1;
        for (int jj = 0; jj < n; jj++) {
            const int rj = order[jj];
            if (rho * fabs(zz[rj]) <= tol) { df.push_back(rj); continue; }
            if (pj < 0) { pj = rj; continue; }
            double s = zz[pj], c = zz[rj];
            const double tau = hypot(c, s);
            const double t = D[rj] - D[pj];
            c /= tau;
            s = -s / tau;
            if (fabs(t * c * s) <= tol) {
                zz[rj] = tau;
                zz[pj] = 0.0;
                P.rots.push_back(DcRot{pj, rj, c, s});
                const double tn = D[pj] * c * c + D[rj] * s * s;
                D[rj] = D[pj] * s * s + D[rj] * c * c;
                D[pj] = tn;
                df.push_back(pj);
                pj = rj;
            } else {
                nd.push_back(pj);
                pj = rj;
            }
        }
        if (pj >= 0) nd.push_back(pj);
    }
    // the rotations may move a pole by a hair past its neighbour: keep the poles strictly increasing by
    // re-sorting the survivors (ties are impossible: equal poles would have been deflated)
    std::stable_sort(nd.begin(), nd.end(), [&](int a, int b) { return D[a] < D[b]; });
    P.k = (int)nd.size();
    P.nondefl = nd;
    P.defl = df;
    double nrm2 = 0.0;
    for (int r : nd) nrm2 += zz[r] * zz[r];
    const double nrm = sqrt(nrm2);
    P.rho = rho * nrm2;
    for (int r : nd) { P.dl.push_back(D[r]); P.w.push_back(zz[r] / nrm); }
    for (int r : df) P.lam_defl.push_back(D[r]);
}

void build_tree(int s, int t, int& height_out, std::vector<std::vector<DcNode>>& by_height, std::vector<DcNode>& leaves) {
    // block boundaries on multiples of DC_ALIGN (tile loads of the merge products are 16-byte aligned)
    const int units = (t - s + DC_ALIGN - 1) / DC_ALIGN;
    if (t - s <= DC_LEAF) {
        leaves.push_back(DcNode{s, t, t});
        height_out = 0;
        return;
    }
    const int m = s + (units / 2) * DC_ALIGN;
    int hl = 0, hr = 0;
    build_tree(s, m, hl, by_height, leaves);
    build_tree(m, t, hr, by_height, leaves);
    const int h = std::max(hl, hr) + 1;
    if ((int)by_height.size() < h + 1) by_height.resize(h + 1);
    by_height[h].push_back(DcNode{s, m, t});
    height_out = h;
}

}  // namespace

int eigh_dc(crm_ctx* ctx, EighWork& w, double* lam_host, double** Qt_out) {
    hipStream_t st = ctx->stream;
    const long dim = w.dim, ld = w.ld, slab = w.slab;
    const int B = w.batch;
    // ---- tree ---------------------------------------------------------------------------------------
    std::vector<std::vector<DcNode>> by_height(1);
    std::vector<DcNode> leaves;
    int height = 0;
    build_tree(0, (int)dim, height, by_height, leaves);
    // ---- cuts: d[c-1] -= |e[c-1]|, d[c] -= |e[c-1]| at every internal boundary --------------------------
    std::vector<double> hd((size_t)B * ld), he((size_t)B * ld);
    CRM_HIP(hipMemcpyAsync(hd.data(), w.d.ptr, sizeof(double) * hd.size(), hipMemcpyDeviceToHost, st));
    CRM_HIP(hipMemcpyAsync(he.data(), w.e.ptr, sizeof(double) * he.size(), hipMemcpyDeviceToHost, st));
    CRM_HIP(hipStreamSynchronize(st));
    for (int b = 0; b < B; b++)
        for (size_t li = 0; li + 1 < leaves.size(); li++) {
            const int c = leaves[li].t;
            const double bt = fabs(he[(size_t)b * ld + c - 1]);
            hd[(size_t)b * ld + c - 1] -= bt;
            hd[(size_t)b * ld + c] -= bt;
        }
    // ---- device scratch -------------------------------------------------------------------------------
    const int max_nodes = (int)leaves.size();
    ScopedBuf dDcut, dLeaf, dLamA, dLamB, dZ, dDl, dW, dZhat, dTau, dOrg, dSrc, dRowNode, dNodes, dDesc, dRots, dProbs;
    CRM_TRY(dDcut.ensure(sizeof(double) * (size_t)B * ld));
    CRM_TRY(dLeaf.ensure(sizeof(int) * 2 * leaves.size()));
    for (ScopedBuf* bf : {&dLamA, &dLamB, &dZ, &dDl, &dW, &dZhat, &dTau}) CRM_TRY(bf->ensure(sizeof(double) * (size_t)B * ld));
    CRM_TRY(dOrg.ensure(sizeof(int) * (size_t)B * ld));
    CRM_TRY(dSrc.ensure(sizeof(int) * (size_t)B * ld));
    CRM_TRY(dRowNode.ensure(sizeof(int) * ld));
    CRM_TRY(dNodes.ensure(sizeof(DcNode) * max_nodes));
    CRM_TRY(dDesc.ensure(sizeof(DcDesc) * (size_t)B * max_nodes));
    CRM_TRY(dProbs.ensure(sizeof(GemmProblem) * (size_t)B * max_nodes));
    CRM_HIP(hipMemcpyAsync(dDcut.ptr, hd.data(), sizeof(double) * hd.size(), hipMemcpyHostToDevice, st));
    {
        std::vector<int> ls(2 * leaves.size());
        for (size_t i = 0; i < leaves.size(); i++) { ls[i] = leaves[i].s; ls[leaves.size() + i] = leaves[i].t; }
        CRM_HIP(hipMemcpyAsync(dLeaf.ptr, ls.data(), sizeof(int) * ls.size(), hipMemcpyHostToDevice, st));
        CRM_HIP(hipStreamSynchronize(st));
    }
    double* cur = w.QA.as<double>();
    double* nxt = w.QB.as<double>();
    double* ysrc = w.Vc.as<double>();
    double* Uall = w.A.as<double>();
    CRM_HIP(hipMemsetAsync(cur, 0, sizeof(double) * (size_t)B * slab, st));
    CRM_HIP(hipMemsetAsync(nxt, 0, sizeof(double) * (size_t)B * slab, st));
    CRM_HIP(hipMemsetAsync(ysrc, 0, sizeof(double) * (size_t)B * slab, st));
    CRM_HIP(hipMemsetAsync(Uall, 0, sizeof(double) * (size_t)B * slab, st));
    double* lamc = dLamA.as<double>();
    double* lamn = dLamB.as<double>();
    hipLaunchKernelGGL(dc_leaf_kernel, dim3((unsigned)leaves.size(), B), dim3(64), 0, st, dDcut.as<double>(),
                       w.e.as<double>(), ld, dLeaf.as<int>(), dLeaf.as<int>() + leaves.size(), lamc, cur, slab);
    CRM_HIP(hipGetLastError());

    // ---- levels -------------------------------------------------------------------------------------------
    std::vector<DcNode> blocks = leaves;   // current blocks, ascending; node.m == node.t
    std::vector<double> hlam((size_t)B * ld), hz((size_t)B * ld), hdl((size_t)B * ld), hw((size_t)B * ld),
        hlamn((size_t)B * ld);
    std::vector<int> hsrc((size_t)B * ld), hrow(ld);
    for (int h = 1; h <= height; h++) {
        // nodes of this level: the merges of height h; every other current block is passed through
        std::vector<DcNode> nodes;
        {
            std::vector<DcNode> merges = by_height[h];
            std::sort(merges.begin(), merges.end(), [](const DcNode& a, const DcNode& b) { return a.s < b.s; });
            size_t mi = 0;
            for (size_t bi = 0; bi < blocks.size();) {
                if (mi < merges.size() && merges[mi].s == blocks[bi].s) {
                    nodes.push_back(merges[mi]);
                    while (bi < blocks.size() && blocks[bi].t <= merges[mi].t) bi++;
                    mi++;
                } else {
                    nodes.push_back(DcNode{blocks[bi].s, blocks[bi].t, blocks[bi].t});
                    bi++;
                }
            }
        }
        const int nn = (int)nodes.size();
        for (int ni = 0; ni < nn; ni++)
            for (int r = nodes[ni].s; r < nodes[ni].t; r++) hrow[r] = ni;
        CRM_HIP(hipMemcpyAsync(dNodes.ptr, nodes.data(), sizeof(DcNode) * nn, hipMemcpyHostToDevice, st));
        CRM_HIP(hipMemcpyAsync(dRowNode.ptr, hrow.data(), sizeof(int) * dim, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(dc_gather_z_kernel, dim3((unsigned)((dim + 255) / 256), B), dim3(256), 0, st, cur, slab, ld,
                           dRowNode.as<int>(), dNodes.as<DcNode>(), dim, dZ.as<double>());
        CRM_HIP(hipGetLastError());
        CRM_HIP(hipMemcpyAsync(hlam.data(), lamc, sizeof(double) * hlam.size(), hipMemcpyDeviceToHost, st));
        CRM_HIP(hipMemcpyAsync(hz.data(), dZ.ptr, sizeof(double) * hz.size(), hipMemcpyDeviceToHost, st));
        CRM_HIP(hipStreamSynchronize(st));
        // host: plan every merge
        std::vector<DcDesc> desc((size_t)B * nn);
        std::vector<DcRot> rots;
        std::vector<GemmProblem> probs;
        int kmax = 0, nmax = 1;
        long cells_max = 0;
        MergePlan P;
        for (int b = 0; b < B; b++) {
            const size_t ob = (size_t)b * ld;
            for (int ni = 0; ni < nn; ni++) {
                const DcNode nd = nodes[ni];
                DcDesc D{};
                D.s = nd.s; D.m = nd.m; D.t = nd.t;
                D.rot_first = (int)rots.size();
                if (nd.m == nd.t) {   // passed through
                    for (int r = nd.s; r < nd.t; r++) { hsrc[ob + r] = r; hlamn[ob + r] = hlam[ob + r]; }
                } else {
                    plan_merge(&hlam[ob + nd.s], &hz[ob + nd.s], nd.m - nd.s, nd.t - nd.s, he[ob + nd.m - 1], P);
                    D.k = P.k;
                    D.rho = P.rho;
                    for (const DcRot& R : P.rots) rots.push_back(DcRot{R.a + nd.s, R.b + nd.s, R.c, R.s});
                    for (int i = 0; i < P.k; i++) {
                        hsrc[ob + nd.s + i] = nd.s + P.nondefl[i];
                        hdl[ob + nd.s + i] = P.dl[i];
                        hw[ob + nd.s + i] = P.w[i];
                    }
                    for (size_t i = 0; i < P.defl.size(); i++) {
                        hsrc[ob + nd.s + P.k + i] = nd.s + P.defl[i];
                        hlamn[ob + nd.s + P.k + i] = P.lam_defl[i];
                    }
                    if (P.k > 0) {
                        GemmProblem g{};
                        g.X = Uall + (size_t)b * slab + (size_t)nd.s * ld; g.ldx = ld;
                        g.Y = ysrc + (size_t)b * slab + (size_t)nd.s * ld + nd.s; g.ldy = ld;
                        g.C = nxt + (size_t)b * slab + (size_t)nd.s * ld + nd.s; g.ldc = ld;
                        g.M = P.k; g.N = nd.t - nd.s;
                        g.cells = (P.k + 15) / 16 * 16;
                        probs.push_back(g);
                        kmax = std::max(kmax, P.k);
                        nmax = std::max(nmax, nd.t - nd.s);
                        cells_max = std::max(cells_max, g.cells);
                    }
                }
                D.rot_count = (int)rots.size() - D.rot_first;
                desc[(size_t)b * nn + ni] = D;
            }
        }
        CRM_HIP(hipMemcpyAsync(dDesc.ptr, desc.data(), sizeof(DcDesc) * desc.size(), hipMemcpyHostToDevice, st));
        CRM_HIP(hipMemcpyAsync(dSrc.ptr, hsrc.data(), sizeof(int) * hsrc.size(), hipMemcpyHostToDevice, st));
        CRM_HIP(hipMemcpyAsync(dDl.ptr, hdl.data(), sizeof(double) * hdl.size(), hipMemcpyHostToDevice, st));
        CRM_HIP(hipMemcpyAsync(dW.ptr, hw.data(), sizeof(double) * hw.size(), hipMemcpyHostToDevice, st));
        CRM_HIP(hipMemcpyAsync(lamn, hlamn.data(), sizeof(double) * hlamn.size(), hipMemcpyHostToDevice, st));
        if (!rots.empty()) {
            CRM_TRY(dRots.ensure(sizeof(DcRot) * rots.size()));
            CRM_HIP(hipMemcpyAsync(dRots.ptr, rots.data(), sizeof(DcRot) * rots.size(), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(dc_rotate_kernel, dim3(nn, B), dim3(256), 0, st, cur, slab, ld, dDesc.as<DcDesc>(), nn,
                               dRots.as<DcRot>());
        }
        hipLaunchKernelGGL(dc_gather_rows_kernel, dim3((unsigned)dim, B), dim3(256), 0, st, cur, nxt, ysrc, slab, ld,
                           dRowNode.as<int>(), dDesc.as<DcDesc>(), nn, dSrc.as<int>(), dim);
        if (kmax > 0) {
            const dim3 grid((unsigned)((kmax + 255) / 256), nn, B);
            hipLaunchKernelGGL(dc_secular_kernel, dim3((unsigned)((kmax + 3) / 4), nn, B), dim3(256), 0, st, dDesc.as<DcDesc>(), nn, ld, dDl.as<double>(),
                               dW.as<double>(), dOrg.as<int>(), dTau.as<double>(), lamn);
            hipLaunchKernelGGL(dc_zhat_kernel, grid, dim3(256), 0, st, dDesc.as<DcDesc>(), nn, ld, dDl.as<double>(),
                               dW.as<double>(), dOrg.as<int>(), dTau.as<double>(), dZhat.as<double>());
            hipLaunchKernelGGL(dc_vectors_kernel, grid, dim3(256), 0, st, dDesc.as<DcDesc>(), nn, ld, slab,
                               dDl.as<double>(), dOrg.as<int>(), dTau.as<double>(), dZhat.as<double>(), Uall);
            CRM_HIP(hipGetLastError());
            CRM_HIP(hipMemcpyAsync(dProbs.ptr, probs.data(), sizeof(GemmProblem) * probs.size(), hipMemcpyHostToDevice, st));
            CRM_TRY(launch_gemm_tn(ctx, dProbs.as<GemmProblem>(), (int)probs.size(), kmax, nmax, cells_max, false, 0, 1, 0));
        }
        CRM_HIP(hipGetLastError());
        CRM_HIP(hipStreamSynchronize(st));   // host vectors of this level are reused by the next
        std::swap(cur, nxt);
        std::swap(lamc, lamn);
        blocks.clear();
        for (const DcNode& nd : nodes) blocks.push_back(DcNode{nd.s, nd.t, nd.t});
    }
    // ---- ascending order ----------------------------------------------------------------------------------
    CRM_HIP(hipMemcpyAsync(hlam.data(), lamc, sizeof(double) * hlam.size(), hipMemcpyDeviceToHost, st));
    CRM_HIP(hipStreamSynchronize(st));
    for (int b = 0; b < B; b++) {
        int* ord = &hsrc[(size_t)b * ld];
        std::iota(ord, ord + dim, 0);
        const double* L = &hlam[(size_t)b * ld];
        std::stable_sort(ord, ord + dim, [&](int a, int c) { return L[a] < L[c]; });
        for (long j = 0; j < dim; j++) lam_host[(size_t)b * dim + j] = L[ord[j]];
    }
    CRM_HIP(hipMemcpyAsync(dSrc.ptr, hsrc.data(), sizeof(int) * hsrc.size(), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(dc_sort_rows_kernel, dim3((unsigned)dim, B), dim3(256), 0, st, cur, nxt, slab, ld, dSrc.as<int>(), dim);
    CRM_HIP(hipGetLastError());
    CRM_HIP(hipStreamSynchronize(st));
    *Qt_out = nxt;
    return CRM_OK;
}

}  // namespace crm

// ---- test hook (host only: no GPU touched): the deflation plan of one merge ------------------------------------------
extern "C" int crm_test_dc_plan(const double* lam, const double* z, int n1, int n, double beta, int* k, double* rho,
                                int* rows, double* dl, double* w, int* nrot, double* rots) {
    return crm::guarded("crm_test_dc_plan", [&]() -> int {
    if (!lam || !z || n1 < 1 || n <= n1 || !k || !rho || !rows || !dl || !w || !nrot || !rots) return CRM_ERR_ARG;
    crm::MergePlan P;
    crm::plan_merge(lam, z, n1, n, beta, P);
    *k = P.k;
    *rho = P.rho;
    for (int i = 0; i < P.k; i++) { rows[i] = P.nondefl[i]; dl[i] = P.dl[i]; w[i] = P.w[i]; }
    for (size_t i = 0; i < P.defl.size(); i++) { rows[P.k + i] = P.defl[i]; dl[P.k + i] = P.lam_defl[i]; w[P.k + i] = 0.0; }
    *nrot = (int)P.rots.size();
    for (size_t i = 0; i < P.rots.size(); i++) {
        rots[4 * i] = P.rots[i].a; rots[4 * i + 1] = P.rots[i].b; rots[4 * i + 2] = P.rots[i].c; rots[4 * i + 3] = P.rots[i].s;
    }
    return CRM_OK;
    });
}
