// Streaming helpers around one block of variants: column statistics of G, gathers
// (permutation hooks idx_E / idx_G of cellregmap/_cellregmap.py:398-413 and the rho*-sorted
// order the Khatri-Rao contraction consumes), elementwise products feeding the side
// contractions.  All HBM-bound, coalesced along the variant / column axis.
#include "nullfit.h"

namespace crm {

namespace {

constexpr int STAT_SPLITS = 64;

// The row-by-row passes below launch one workgroup per (row, chunk of blockDim.x columns) with the CHUNK as the fast
// index of a one-dimensional grid: workgroups dispatched together then stream neighbouring memory.  With the row as the fast
// index (a dim3(rows, chunks) grid) neighbouring workgroups sit a whole row apart, and a 20 096 x 4096 block of doubles is
// copied at 5.0 TB/s instead of 5.9 (tools/diag/micro/copy_shapes.hip).
__device__ __forceinline__ void row_and_chunk(int chunks, long& row, int& chunk) {
    row = blockIdx.x / (unsigned)chunks;
    chunk = (int)(blockIdx.x - (unsigned)row * (unsigned)chunks);
}
inline dim3 row_chunk_grid(long rows, int chunks) { return dim3((unsigned)(rows * chunks)); }

// partial[split][q][b]: q = 0 gg, 1 gy, 2.. gW_i.  y and W are columns of one row-major
// matrix (leading dimension ldw).
template <int C>
__global__ __launch_bounds__(256) void variant_stats_kernel(const double* __restrict__ G, long ldg,
                                                           long cells, int variants,
                                                           const double* __restrict__ y,
                                                           const double* __restrict__ W, long ldw,
                                                           double* __restrict__ partial) {
    __shared__ double red[4][C + 2][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int b = blockIdx.x * 64 + tx;
    const long per = (cells + STAT_SPLITS - 1) / STAT_SPLITS;
    const long i0 = (long)blockIdx.y * per;
    const long i1 = i0 + per < cells ? i0 + per : cells;
    double acc[C + 2];
#pragma unroll
    for (int q = 0; q < C + 2; q++) acc[q] = 0.0;
    if (b < variants) {
        for (long i = i0 + ty; i < i1; i += 4) {
            const double g = G[i * ldg + b];
            acc[0] += g * g;
            acc[1] += g * y[i * ldw];
#pragma unroll
            for (int q = 0; q < C; q++) acc[2 + q] += g * W[i * ldw + q];
        }
    }
#pragma unroll
    for (int q = 0; q < C + 2; q++) red[ty][q][tx] = acc[q];
    __syncthreads();
    if (ty == 0 && b < variants) {
#pragma unroll
        for (int q = 0; q < C + 2; q++) {
            const double s = ((red[0][q][tx] + red[1][q][tx]) + red[2][q][tx]) + red[3][q][tx];
            partial[((long)blockIdx.y * (C + 2) + q) * variants + b] = s;
        }
    }
}

// any number of covariate columns: one thread per (variant, quantity); q = 0 gg, 1 gy, 2.. gW_i
__global__ __launch_bounds__(64) void variant_stats_general_kernel(const double* __restrict__ G, long ldg,
                                                                  long cells, int variants,
                                                                  const double* __restrict__ yW, long ldw,
                                                                  double* __restrict__ gg,
                                                                  double* __restrict__ gy,
                                                                  double* __restrict__ gW, long ld_gW) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    const int q = blockIdx.y;
    if (b >= variants) return;
    double acc = 0.0;
    if (q == 0) {
        for (long i = 0; i < cells; i++) {
            const double g = G[i * ldg + b];
            acc += g * g;
        }
        gg[b] = acc;
    } else {
        const double* col = yW + (q - 1);  // column 0 = y, 1.. = W
        for (long i = 0; i < cells; i++) acc += G[i * ldg + b] * col[i * ldw];
        if (q == 1) gy[b] = acc;
        else gW[(long)b * ld_gW + (q - 2)] = acc;
    }
}

__global__ void variant_stats_finish(const double* __restrict__ partial, int variants, int c,
                                     double* __restrict__ gg, double* __restrict__ gy,
                                     double* __restrict__ gW, long ld_gW) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= variants) return;
    for (int q = 0; q < c + 2; q++) {
        double s = 0.0;
        for (int k = 0; k < STAT_SPLITS; k++) s += partial[((long)k * (c + 2) + q) * variants + b];
        if (q == 0) gg[b] = s;
        else if (q == 1) gy[b] = s;
        else gW[(long)b * ld_gW + (q - 2)] = s;
    }
}

__global__ void gather_block_kernel(const double* __restrict__ src, long ld_src, long cells_pad,
                                    long cells, const int* __restrict__ row_index,
                                    const int* __restrict__ col_index, int variants,
                                    double* __restrict__ dst, long ld_dst, int dst_cols, int chunks) {
    long i;
    int chunk;
    row_and_chunk(chunks, i, chunk);
    const int j = chunk * blockDim.x + threadIdx.x;
    if (j >= dst_cols) return;
    double v = 0.0;
    if (i < cells && j < variants) {
        const long si = row_index ? row_index[i] : i;
        const long sj = col_index ? col_index[j] : j;
        v = src[si * ld_src + sj];
    }
    dst[i * ld_dst + j] = v;
}

__global__ void expand_block_kernel(const double* __restrict__ Gd, long ld_gd, const int* __restrict__ group,
                                    long cells, const int* __restrict__ row_index, int variants,
                                    double* __restrict__ dst, long ld_dst, int dst_cols, int chunks) {
    long i;
    int chunk;
    row_and_chunk(chunks, i, chunk);
    const int j = chunk * blockDim.x + threadIdx.x;
    if (j >= dst_cols) return;
    double v = 0.0;
    if (i < cells && j < variants) {
        const long si = row_index ? row_index[i] : i;
        v = Gd[(long)group[si] * ld_gd + j];
    }
    dst[i * ld_dst + j] = v;
}

__global__ void indicator_kernel(const int* __restrict__ group, long cells, int m, double* __restrict__ Z,
                                 long ldz) {
    const int d = blockIdx.y * blockDim.x + threadIdx.x;
    const long i = blockIdx.x;
    if (d >= ldz) return;
    Z[i * ldz + d] = (i < cells && d < m && group[i] == d) ? 1.0 : 0.0;
}

__global__ void donor_stats_kernel(const double* __restrict__ Gam, long ld_gam, int m, int variants,
                                   const double* __restrict__ sums, int c, double* __restrict__ gg,
                                   double* __restrict__ gy, double* __restrict__ gW, long ld_gW) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const int q = blockIdx.y;  // 0 gg, 1 gy, 2.. gW
    if (b >= variants) return;
    double acc = 0.0;
    for (int d = 0; d < m; d++) {
        const double g = Gam[(long)d * ld_gam + b];
        acc += (q == 0 ? g * g : g) * sums[d * DT_SUMS_LD + q];
    }
    if (q == 0) gg[b] = acc;
    else if (q == 1) gy[b] = acc;
    else gW[(long)b * ld_gW + (q - 2)] = acc;
}

__global__ void square_block_kernel(const double* __restrict__ Gt, const double* __restrict__ G,
                                    long ldg, long ldg_t, int cols, double* __restrict__ G2,
                                    double* __restrict__ GG, long ld_out, int chunks) {
    long i;
    int chunk;
    row_and_chunk(chunks, i, chunk);
    const int j = chunk * blockDim.x + threadIdx.x;
    if (j >= cols) return;
    const double t = Gt[i * ldg_t + j];
    G2[i * ld_out + j] = t * t;
    if (GG) GG[i * ld_out + j] = t * G[i * ldg + j];
}

__global__ void context_features_kernel(const double* __restrict__ E, long lde,
                                        const int* __restrict__ row_index, long cells, int k0,
                                        const double* __restrict__ y, const double* __restrict__ W,
                                        long ldw, int c, double* __restrict__ Ep, long ld_ep,
                                        double* __restrict__ YE, long ld_ye, double* __restrict__ EE,
                                        long ld_ee) {
    extern __shared__ double row[];  // k0 entries of the (permuted) context row
    const long i = blockIdx.x;
    const bool live = i < cells;
    const long si = live ? (row_index ? row_index[i] : i) : 0;
    for (int j = threadIdx.x; j < k0; j += blockDim.x) row[j] = live ? E[si * lde + j] : 0.0;
    __syncthreads();
    if (Ep)
        for (int j = threadIdx.x; j < ld_ep; j += blockDim.x) Ep[i * ld_ep + j] = j < k0 ? row[j] : 0.0;
    const int nye = k0 * (1 + c);
    for (int q = threadIdx.x; q < ld_ye; q += blockDim.x) {
        double v = 0.0;
        if (q < nye && live) {
            const int u = q / k0, j = q - u * k0;
            v = (u == 0 ? y[i * ldw] : W[i * ldw + (u - 1)]) * row[j];
        }
        YE[i * ld_ye + q] = v;
    }
    // pairs (j, j'), j <= j', row-major over the upper triangle
    const int npair = k0 * (k0 + 1) / 2;
    if (!EE) return;
    for (int q = threadIdx.x; q < ld_ee; q += blockDim.x) {
        double v = 0.0;
        if (q < npair) {
            // invert q -> (j, j'): rows of the triangle have lengths k0, k0-1, ...
            int j = 0, rem = q;
            while (rem >= k0 - j) {
                rem -= k0 - j;
                j++;
            }
            v = row[j] * row[j + rem];
        }
        EE[i * ld_ee + q] = v;
    }
}

}  // namespace

int launch_variant_stats(hipStream_t st, const double* G, long ldg, long cells, int variants,
                         const double* y, const double* W, long ldw, int c, double* partial,
                         double* gg, double* gy, double* gW, long ld_gW) {
    if (variants <= 0) return CRM_OK;
    dim3 grid((variants + 63) / 64, STAT_SPLITS);
#define CRM_STATS(CC)                                                                          \
    case CC:                                                                                   \
        hipLaunchKernelGGL(variant_stats_kernel<CC>, grid, dim3(256), 0, st, G, ldg, cells,    \
                           variants, y, W, ldw, partial);                                      \
        break;
    switch (c) {
        CRM_STATS(1) CRM_STATS(2) CRM_STATS(3) CRM_STATS(4)
        CRM_STATS(5) CRM_STATS(6) CRM_STATS(7) CRM_STATS(8)
        default: {
            // y points at column 0 of the packed [y | W] matrix
            dim3 g2((variants + 63) / 64, c + 2);
            hipLaunchKernelGGL(variant_stats_general_kernel, g2, dim3(64), 0, st, G, ldg, cells, variants, y,
                               ldw, gg, gy, gW, ld_gW);
            CRM_HIP(hipGetLastError());
            return CRM_OK;
        }
    }
#undef CRM_STATS
    CRM_HIP(hipGetLastError());
    hipLaunchKernelGGL(variant_stats_finish, dim3((variants + 127) / 128), dim3(128), 0, st, partial,
                       variants, c, gg, gy, gW, ld_gW);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

size_t variant_stats_workspace(int variants, int c) {
    return sizeof(double) * (size_t)STAT_SPLITS * (c + 2) * variants;
}

// dst[h, p*k0 + j] = src[h, ord[p]*k0 + j]: k0-wide column slabs reordered (pair order), zero beyond
__global__ __launch_bounds__(256) void gather_slabs_kernel(const double* __restrict__ src, long ld_src,
                                                            const int* __restrict__ ord, int npairs, int k0,
                                                            double* __restrict__ dst, long ld_dst, int dst_cols) {
    const long h = blockIdx.x;
    const int q = blockIdx.y * 256 + threadIdx.x;
    if (q >= dst_cols) return;
    double v = 0.0;
    const int p = q / k0;
    if (p < npairs) v = src[h * ld_src + (long)ord[p] * k0 + (q - p * k0)];
    dst[h * ld_dst + q] = v;
}

int launch_gather_slabs(hipStream_t st, const double* src, long ld_src, long rows, const int* ord, int npairs,
                        int k0, double* dst, long ld_dst, int dst_cols) {
    if (rows <= 0 || dst_cols <= 0) return CRM_OK;
    dim3 grid((unsigned)rows, (dst_cols + 255) / 256);
    hipLaunchKernelGGL(gather_slabs_kernel, grid, dim3(256), 0, st, src, ld_src, ord, npairs, k0, dst, ld_dst,
                       dst_cols);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

// ---- donor order of a background's kinship structure (objects.h: crm_background::kin*) ------------------------------
// dst[k, j] = src[map[k], j] (map[k] < 0: a padding row of zeros), j < cols
__global__ void gather_rows_kernel(const double* __restrict__ src, long ld_src, const int* __restrict__ map, int cols,
                                   double* __restrict__ dst, long ld_dst, int chunks) {
    long k;
    int chunk;
    row_and_chunk(chunks, k, chunk);
    const int j = chunk * blockDim.x + threadIdx.x;
    if (j >= cols) return;
    const int c = map[k];
    dst[k * ld_dst + j] = c >= 0 ? src[(long)c * ld_src + j] : 0.0;
}

int launch_gather_rows(hipStream_t st, const double* src, long ld_src, const int* map, long rows, int cols, double* dst,
                       long ld_dst) {
    if (rows <= 0 || cols <= 0) return CRM_OK;
    const int chunks = (cols + 255) / 256;
    hipLaunchKernelGGL(gather_rows_kernel, row_chunk_grid(rows, chunks), dim3(256), 0, st, src, ld_src, map, cols, dst, ld_dst,
                       chunks);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

// Y[k, 0 .. k2) = U[map[k], :], Y[k, k2 .. k2 + k1) = H[map[k], 0 .. k1) (the E1 columns of the half factor), zero elsewhere
__global__ void kin_operand_kernel(const double* __restrict__ U, int k2, const double* __restrict__ H, long ldh, int k1,
                                   const int* __restrict__ map, double* __restrict__ Y, long ldy) {
    const long k = blockIdx.x;
    const int j = threadIdx.x;
    if (j >= ldy) return;
    const int c = map[k];
    double v = 0.0;
    if (c >= 0) {
        if (j < k2) v = U[(long)c * k2 + j];
        else if (j < k2 + k1) v = H[(long)c * ldh + (j - k2)];
    }
    Y[k * ldy + j] = v;
}

int launch_kin_operand(hipStream_t st, const double* U, int k2, const double* H, long ldh, int k1, const int* map, long rows,
                       double* Y, long ldy) {
    if (rows <= 0) return CRM_OK;
    hipLaunchKernelGGL(kin_operand_kernel, dim3((unsigned)rows), dim3((unsigned)ldy), 0, st, U, k2, H, ldh, k1, map, Y, ldy);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

// Is the half factor what the announced structure says, H[c, k1 + j m + d] = U[c, j] hKd[group(c), d]?  out[0] = largest
// difference, out[1] = largest |H| over those columns (bit patterns: non-negative doubles order like integers).
__global__ void kin_verify_kernel(const double* __restrict__ H, long ldh, int k1, const double* __restrict__ U, int k2,
                                  const int* __restrict__ group, const double* __restrict__ hKd, long ldk, long m,
                                  unsigned long long* __restrict__ out) {
    const long c = blockIdx.x;
    const long g = group[c];
    double dmax = 0.0, hmax = 0.0;
    for (long e = threadIdx.x; e < (long)k2 * m; e += blockDim.x) {
        const long j = e / m, d = e - j * m;
        const double h = H[c * ldh + k1 + e];
        const double want = U[c * k2 + j] * hKd[g * ldk + d];
        const double diff = fabs(h - want);
        dmax = diff > dmax || diff != diff ? diff : dmax;      // (a NaN counts as a difference)
        hmax = fmax(hmax, fabs(h));
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(dmax, off);
        dmax = o > dmax || o != o ? o : dmax;
        hmax = fmax(hmax, __shfl_xor(hmax, off));
    }
    if ((threadIdx.x & 63) == 0) {
        if (dmax != dmax) dmax = INFINITY;
        atomicMax(&out[0], (unsigned long long)__double_as_longlong(dmax));
        atomicMax(&out[1], (unsigned long long)__double_as_longlong(hmax));
    }
}

int launch_kin_verify(hipStream_t st, const double* H, long ldh, int k1, const double* U, int k2, const int* group,
                      const double* hKd, long ldk, long m, long n, unsigned long long* out) {
    if (n <= 0) return CRM_OK;
    hipLaunchKernelGGL(kin_verify_kernel, dim3((unsigned)n), dim3(256), 0, st, H, ldh, k1, U, k2, group, hKd, ldk, m, out);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

// AH[a, c] = sum over the donors d' of S[(d' KT + k2 + a), c]: the E1 rows of H'(g o E0) from the per-donor sums
__global__ void kin_sum_e1_kernel(const double* __restrict__ S, long ld_s, int KT, int k2, int groups, long cols,
                                  double* __restrict__ AH, long ld_ah) {
    const int a = blockIdx.y;
    const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    double acc = 0.0;
    for (int d = 0; d < groups; d++) acc += S[((long)d * KT + k2 + a) * ld_s + c];
    AH[(long)a * ld_ah + c] = acc;
}

int launch_kin_sum_e1(hipStream_t st, const double* S, long ld_s, int KT, int k2, int k1, int groups, long cols, double* AH,
                      long ld_ah) {
    if (k1 <= 0 || cols <= 0) return CRM_OK;
    dim3 grid((unsigned)((cols + 255) / 256), (unsigned)k1);
    hipLaunchKernelGGL(kin_sum_e1_kernel, grid, dim3(256), 0, st, S, ld_s, KT, k2, groups, cols, AH, ld_ah);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

// ---- the variant in the fixed effects' own basis (cellregmap/_cellregmap.py:345-352) ----------------------------------
// The reference's LMM never works with X = [W, g] as given: glimix-core reduces it by numpy_sugar.economic_svd (an
// orthonormal basis of its column space, singular values below sqrt(eps) dropped) before anything is rotated or solved.
// The same here: W arrives with mutually orthogonal columns (crm_gene_create), and each block's variants are made
// orthogonal to them IN THE CELL AXIS before the rotations and the n-length sums --
//     gx = g - W a,   a_j = W_j'g / W_j'W_j
// so that a variant nearly collinear with the covariates costs eps sqrt(cond) like the reference's basis, not the
// eps cond of a Cholesky factorisation of [W, g]'K^-1[W, g] in the raw basis.  span([W, gx]) = span([W, g]): the
// projection matrix of the score test and every likelihood are unchanged; only the fixed effects' role of g is
// affected (the test direction g o E0 keeps the variant as given).
//
// coef[j * ld_coef + b] = a_j of variant b (zero in the padding columns); thr[b] = the reference's rank rule turned into
// a bound on |gx|^2: the smallest singular value of [W, g] lies below sqrt(eps) (economic_svd then drops that
// direction) exactly when  |gx|^2 < eps (1 + sum_j h_j^2 / (d_j^2 (d_j^2 - eps))),  h = V'(W'g), (V, d^2) the
// eigen-decomposition of W'W -- the secular equation of the arrowhead matrix [W, g]'[W, g] in W's own orthogonal
// basis, evaluated at eps.  proj (crm_gene::Wproj): (W'W)^-1 [c x c], V [c x c] (column j = eigenvector j), d^2 [c];
// for the mutually orthogonal columns the Python host passes all three are diagonal.
__global__ void ortho_coef_kernel(const double* __restrict__ gW, long ld_gW, const double* __restrict__ proj, int c,
                                  int variants, int cols, double* __restrict__ coef, long ld_coef,
                                  double* __restrict__ thr) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= cols) return;
    constexpr double EPS = 2.220446049250313e-16;
    const double* __restrict__ inv = proj;
    const double* __restrict__ V = proj + (long)c * c;
    const double* __restrict__ d2 = V + (long)c * c;
    const double* __restrict__ h = gW + (long)b * ld_gW;
    double t = 1.0;
    for (int j = 0; j < c; j++) {
        double a = 0.0;
        if (b < variants) {
            double hv = 0.0;
            for (int k = 0; k < c; k++) {
                a = fma(inv[j * c + k], h[k], a);
                hv = fma(V[k * c + j], h[k], hv);
            }
            t += (hv / d2[j]) * (hv / (d2[j] - EPS));
        }
        coef[(long)j * ld_coef + b] = a;
    }
    if (b < variants) thr[b] = EPS * t;
}

__global__ __launch_bounds__(256) void ortho_apply_kernel(const double* __restrict__ G, long ldg,
                                                         const double* __restrict__ W, long ldw, int c,
                                                         const double* __restrict__ coef, long ld_coef, int cols,
                                                         double* __restrict__ Gx, long ldx, int chunks) {
    long i;
    int chunk;
    row_and_chunk(chunks, i, chunk);
    const int b = chunk * blockDim.x + threadIdx.x;
    if (b >= cols) return;
    double v = G[i * ldg + b];
    for (int j = 0; j < c; j++) v = fma(-W[i * ldw + j], coef[(long)j * ld_coef + b], v);
    Gx[i * ldx + b] = v;
}

// drop[b] = 1 when the variant's own direction falls under the reference's rank rule (gg: |gx|^2 of the block)
__global__ void ortho_rank_kernel(const double* __restrict__ gg, const double* __restrict__ thr, int variants,
                                  int* __restrict__ drop) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < variants) drop[b] = gg[b] < thr[b] ? 1 : 0;
}

// Collapsed path (donor-level sums only): near[b] = 1 when the variant keeps less than `tau` of its squared norm
// outside span(W) -- the scan then repeats such variants on the dense path, where they are orthogonalised in the cell axis.
__global__ void collinear_flag_kernel(const double* __restrict__ gg, const double* __restrict__ gW, long ld_gW,
                                      const double* __restrict__ proj, int c, int variants, double tau,
                                      int* __restrict__ near) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= variants) return;
    const double* __restrict__ h = gW + (long)b * ld_gW;
    double rest = gg[b];
    for (int j = 0; j < c; j++) {
        double a = 0.0;
        for (int k = 0; k < c; k++) a = fma(proj[j * c + k], h[k], a);
        rest -= h[j] * a;
    }
    near[b] = rest > tau * gg[b] ? 0 : 1;
}

int launch_ortho_block(hipStream_t st, const double* G, long ldg, long cells_pad, int variants, int cols,
                       const double* W, long ldw, int c, const double* proj, const double* gW, long ld_gW,
                       double* coef, long ld_coef, double* thr, double* Gx, long ldx) {
    if (cols <= 0) return CRM_OK;
    hipLaunchKernelGGL(ortho_coef_kernel, dim3((cols + 255) / 256), dim3(256), 0, st, gW, ld_gW, proj, c, variants, cols,
                       coef, ld_coef, thr);
    CRM_HIP(hipGetLastError());
    const int chunks = (cols + 255) / 256;
    hipLaunchKernelGGL(ortho_apply_kernel, row_chunk_grid(cells_pad, chunks), dim3(256), 0, st, G, ldg, W,
                       ldw, c, coef, ld_coef, cols, Gx, ldx, chunks);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

int launch_ortho_rank(hipStream_t st, const double* gg, const double* thr, int variants, int* drop) {
    if (variants <= 0) return CRM_OK;
    hipLaunchKernelGGL(ortho_rank_kernel, dim3((variants + 255) / 256), dim3(256), 0, st, gg, thr, variants, drop);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

int launch_collinear_flag(hipStream_t st, const double* gg, const double* gW, long ld_gW, const double* proj, int c,
                          int variants, double tau, int* near) {
    if (variants <= 0) return CRM_OK;
    hipLaunchKernelGGL(collinear_flag_kernel, dim3((variants + 255) / 256), dim3(256), 0, st, gg, gW, ld_gW, proj, c,
                       variants, tau, near);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

// ---- E1 rows of the folded kinship-structure form through pair products (scan.hip, step 6) --------------------------
// P[c, a k0 + i] = H[c, a] * Ep[c, i]  (a < k1: the E1 columns of the half factor; Ep: the scan's (permuted) contexts)
__global__ void pair_features_kernel(const double* __restrict__ H, long ldh, int k1, const double* __restrict__ Ep,
                                     long ld_ep, int k0, double* __restrict__ P, long ldp) {
    const long c = blockIdx.x;
    for (int q = threadIdx.x; q < ldp; q += blockDim.x) {
        const int a = q / k0, i = q - a * k0;
        P[c * ldp + q] = a < k1 ? H[c * ldh + a] * Ep[c * ld_ep + i] : 0.0;
    }
}

int launch_pair_features(hipStream_t st, const double* H, long ldh, int k1, const double* Ep, long ld_ep, int k0,
                         long cells_pad, double* P, long ldp) {
    if (cells_pad <= 0) return CRM_OK;
    hipLaunchKernelGGL(pair_features_kernel, dim3((unsigned)cells_pad), dim3(256), 0, st, H, ldh, k1, Ep, ld_ep, k0, P, ldp);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

// flag[0] |= 1 when H[c, a] differs (bitwise) from Ep[c, a] for some cell c < cells, a < k: are the background's E1 columns
// the scan's own contexts (the reference's default E1 = E, no context permutation)?
__global__ void same_columns_kernel(const double* __restrict__ H, long ldh, const double* __restrict__ Ep, long ld_ep,
                                    long cells, int k, int* __restrict__ flag) {
    const long c = blockIdx.x;
    if (c >= cells) return;
    bool diff = false;
    for (int a = threadIdx.x; a < k; a += blockDim.x)
        diff |= __double_as_longlong(H[c * ldh + a]) != __double_as_longlong(Ep[c * ld_ep + a]);
    if (diff) atomicOr(flag, 1);
}

int launch_same_columns(hipStream_t st, const double* H, long ldh, const double* Ep, long ld_ep, long cells, int k, int* flag) {
    if (cells <= 0 || k <= 0) return CRM_OK;
    hipLaunchKernelGGL(same_columns_kernel, dim3((unsigned)cells), dim3(64), 0, st, H, ldh, Ep, ld_ep, cells, k, flag);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

// S[a, b k0 + i] = C[b, pair(min(a, i), max(a, i))]: the E1 rows from G'(E (x) E) when E1 is E itself (pairs j <= j' in
// row-major order of the upper triangle, as context_features writes them)
__global__ void pair_rows_sym_kernel(const double* __restrict__ C, long ldc, int variants, int k0, double* __restrict__ S,
                                     long lds) {
    const int a = blockIdx.y;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;   // b k0 + i
    if (e >= (long)variants * k0) return;
    const long b = e / k0;
    const int i = (int)(e - b * k0);
    const int lo = a < i ? a : i, hi = a < i ? i : a;
    const long pidx = (long)lo * k0 - (long)lo * (lo - 1) / 2 + (hi - lo);
    S[(long)a * lds + e] = C[b * ldc + pidx];
}

int launch_pair_rows_sym(hipStream_t st, const double* C, long ldc, int variants, int k0, double* S, long lds) {
    if (variants <= 0 || k0 <= 0) return CRM_OK;
    dim3 grid((unsigned)(((long)variants * k0 + 255) / 256), (unsigned)k0);
    hipLaunchKernelGGL(pair_rows_sym_kernel, grid, dim3(256), 0, st, C, ldc, variants, k0, S, lds);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

// ---- per-donor sums of the folded kinship-structure form from the symmetric pair features (scan.hip, step 6) ----------
// P[d][b][pair(j, i)] = sum over the cells of donor d of g_b E_j E_i (j <= i; one batched product against E (x) E in donor
// order) holds everything of S_d = sum_c g_c e_c e_c' when the contexts of the kinship term are the scan's own (the
// reference's default E2 = E).  A workgroup takes four variants and a range of donors and walks them:
//     S[(k1 + d k0 + j), b k0 + i] = P[d][b][pair(min(j, i), max(j, i))]      4 k0 contiguous doubles per row of S
// and its sum over the donors -- the E1 rows' G'(E (x) E) over all cells when E1 = E as well -- goes to Psum[split][b][pair].
constexpr int DP_VARIANTS = 4, DP_MAX_SLOTS = 32;   // 4 * npair <= 256 * slots and 4 * k0 <= 256: k0 <= 63

template <int DP_SLOTS>     // entries of the four pair rows per thread (registers: as few as the context count needs)
__global__ __launch_bounds__(256) void donor_pairs_expand_kernel(const double* __restrict__ P, long p_slab, long ldp, int donors,
                                                                  int variants, int k0, int k1, double* __restrict__ S, long lds,
                                                                  double* __restrict__ Psum, long psum_slab, long row_d, long row_j) {
    extern __shared__ double tile[];              // [DP_VARIANTS][npair]
    const int npair = k0 * (k0 + 1) / 2, total = DP_VARIANTS * npair;
    const int v0 = blockIdx.x * DP_VARIANTS, tid = threadIdx.x;
    const int per = (donors + gridDim.y - 1) / gridDim.y;
    const int d0 = blockIdx.y * per, d1 = min(donors, d0 + per);
    if (d0 >= d1) return;
    // reading: entry e = tid + 256 q of the four variants' pair rows, the same for every donor
    double acc[DP_SLOTS], x[DP_SLOTS];
    int src[DP_SLOTS];
#pragma unroll
    for (int q = 0; q < DP_SLOTS; q++) {
        const int e = tid + 256 * q, v = e / npair, pr = e - v * npair;
        acc[q] = 0.0;
        src[q] = e < total && v0 + v < variants ? (int)((long)(v0 + v) * ldp + pr) : -1;   // (< 2^31: rows x 2 048 at most)
    }
    auto fetch = [&](int d) {
        const double* __restrict__ Pd = P + (size_t)d * p_slab;
#pragma unroll
        for (int q = 0; q < DP_SLOTS; q++)
            if (tid + 256 * q < total) x[q] = src[q] >= 0 ? Pd[src[q]] : 0.0;
    };
    // writing: thread t < 4 k0 owns column t of the four variants' run in every row j of the donor: (variant, i) fixed
    const int row_len = DP_VARIANTS * k0;
    const int vw = tid / k0, iw = tid - vw * k0;
    const bool writer = tid < row_len && v0 + vw < variants;
    const int base_i = iw * k0 - iw * (iw - 1) / 2 - iw;          // pair(i, j) = base_i + j for j >= i
    const double* __restrict__ mine = tile + vw * npair;
    fetch(d0);
    for (int d = d0; d < d1; d++) {
#pragma unroll
        for (int q = 0; q < DP_SLOTS; q++)
            if (tid + 256 * q < total) { acc[q] += x[q]; tile[tid + 256 * q] = x[q]; }
        __syncthreads();
        if (d + 1 < d1) fetch(d + 1);      // (in flight while this donor's rows are written)
        if (writer) {
            // (row of (d, j): k1 + d row_d + j row_j -- the folded form's k1 + d k0 + j, or k1 + j m + d where the donors' index
            // is a column of the kinship factor and the rows are those of the half factor itself)
            double* __restrict__ Sd = S + (size_t)(k1 + (long)d * row_d) * lds + (long)v0 * k0 + tid;
            for (int j = 0; j < k0; j++) {
                const int pidx = j <= iw ? j * k0 - j * (j - 1) / 2 + (iw - j) : base_i + j;
                Sd[(long)j * row_j * lds] = mine[pidx];
            }
        }
        __syncthreads();
    }
    double* __restrict__ out = Psum + (size_t)blockIdx.y * psum_slab;
#pragma unroll
    for (int q = 0; q < DP_SLOTS; q++)
        if (tid + 256 * q < total && src[q] >= 0) out[src[q]] = acc[q];
}

bool donor_pairs_serves(int k0) { return k0 >= 2 && DP_VARIANTS * (k0 * (k0 + 1) / 2) <= 256 * DP_MAX_SLOTS && DP_VARIANTS * k0 <= 256; }

int launch_donor_pairs_expand(hipStream_t st, const double* P, long p_slab, long ldp, int donors, int variants, int k0, int k1,
                              double* S, long lds, double* Psum, long psum_slab, int splits, long row_d, long row_j) {
    if (row_d <= 0) { row_d = k0; row_j = 1; }
    if (variants <= 0 || donors <= 0) return CRM_OK;
    if (!donor_pairs_serves(k0) || splits < 1) {
        set_error("donor pairs: k0=%d outside the supported range", k0);
        return CRM_ERR_UNSUPPORTED;
    }
    const size_t lds_bytes = sizeof(double) * DP_VARIANTS * (size_t)(k0 * (k0 + 1) / 2);
    dim3 grid((unsigned)((variants + DP_VARIANTS - 1) / DP_VARIANTS), (unsigned)splits);
    const int slots = (DP_VARIANTS * (k0 * (k0 + 1) / 2) + 255) / 256;
#define CRM_DP_LAUNCH(SL)                                                                                                        \
    hipLaunchKernelGGL(donor_pairs_expand_kernel<SL>, grid, dim3(256), lds_bytes, st, P, p_slab, ldp, donors, variants, k0, k1, \
                       S, lds, Psum, psum_slab, row_d, row_j)
    if (slots <= 4) CRM_DP_LAUNCH(4);
    else if (slots <= 8) CRM_DP_LAUNCH(8);
    else if (slots <= 16) CRM_DP_LAUNCH(16);
    else if (slots <= 24) CRM_DP_LAUNCH(24);
    else CRM_DP_LAUNCH(32);
#undef CRM_DP_LAUNCH
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

// dst[k, j] = src[k, j] * scale[k * ld_scale]  (j < cols): the contexts times the single column of us, donor order
__global__ void scale_rows_kernel(const double* __restrict__ src, long ld_src, const double* __restrict__ scale, long ld_scale,
                                  int cols, double* __restrict__ dst, long ld_dst) {
    const long k = blockIdx.x;
    const double f = scale[k * ld_scale];
    for (int j = threadIdx.x; j < cols; j += blockDim.x) dst[k * ld_dst + j] = src[k * ld_src + j] * f;
}

int launch_scale_rows(hipStream_t st, const double* src, long ld_src, const double* scale, long ld_scale, long rows, int cols,
                      double* dst, long ld_dst) {
    if (rows <= 0 || cols <= 0) return CRM_OK;
    hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)rows), dim3(128), 0, st, src, ld_src, scale, ld_scale, cols, dst, ld_dst);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

// S[a, b k0 + i] = C[b, a k0 + i]  (b < variants, a < k1, i < k0): k0-long runs are contiguous on both sides
__global__ void pair_rows_kernel(const double* __restrict__ C, long ldc, int variants, int k1, int k0,
                                 double* __restrict__ S, long lds) {
    const int a = blockIdx.y;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;   // b k0 + i
    if (e >= (long)variants * k0) return;
    const long b = e / k0;
    const int i = (int)(e - b * k0);
    S[(long)a * lds + e] = C[b * ldc + (long)a * k0 + i];
}

int launch_pair_rows(hipStream_t st, const double* C, long ldc, int variants, int k1, int k0, double* S, long lds) {
    if (variants <= 0 || k1 <= 0) return CRM_OK;
    dim3 grid((unsigned)(((long)variants * k0 + 255) / 256), (unsigned)k1);
    hipLaunchKernelGGL(pair_rows_kernel, grid, dim3(256), 0, st, C, ldc, variants, k1, k0, S, lds);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

int launch_gather_block(hipStream_t st, const double* src, long ld_src, long cells_pad, long cells,
                        const int* row_index, const int* col_index, int variants, double* dst,
                        long ld_dst, int dst_cols) {
    const int chunks = (dst_cols + 255) / 256;
    hipLaunchKernelGGL(gather_block_kernel, row_chunk_grid(cells_pad, chunks), dim3(256), 0, st, src, ld_src, cells_pad, cells,
                       row_index, col_index, variants, dst, ld_dst, dst_cols, chunks);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

int launch_expand_block(hipStream_t st, const double* Gd, long ld_gd, const int* group, long cells_pad,
                        long cells, const int* row_index, int variants, double* dst, long ld_dst,
                        int dst_cols) {
    const int chunks = (dst_cols + 255) / 256;
    hipLaunchKernelGGL(expand_block_kernel, row_chunk_grid(cells_pad, chunks), dim3(256), 0, st, Gd, ld_gd, group, cells,
                       row_index, variants, dst, ld_dst, dst_cols, chunks);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

int launch_indicator(hipStream_t st, const int* group, long cells, long cells_pad, int m, double* Z, long ldz) {
    dim3 grid((unsigned)cells_pad, (unsigned)((ldz + 127) / 128));
    hipLaunchKernelGGL(indicator_kernel, grid, dim3(128), 0, st, group, cells, m, Z, ldz);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

int launch_donor_stats(hipStream_t st, const double* Gam, long ld_gam, int m, int variants,
                       const double* sums, int c, double* gg, double* gy, double* gW, long ld_gW) {
    if (variants <= 0) return CRM_OK;
    dim3 grid((variants + 127) / 128, c + 2);
    hipLaunchKernelGGL(donor_stats_kernel, grid, dim3(128), 0, st, Gam, ld_gam, m, variants, sums, c, gg, gy, gW,
                       ld_gW);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

int launch_square_block(hipStream_t st, const double* Gt, const double* G, long ldg, long ldg_t,
                        long cells_pad, int cols, double* G2, double* GG, long ld_out) {
    const int chunks = (cols + 255) / 256;
    hipLaunchKernelGGL(square_block_kernel, row_chunk_grid(cells_pad, chunks), dim3(256), 0, st, Gt, G, ldg, ldg_t, cols, G2, GG,
                       ld_out, chunks);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

int launch_context_features(hipStream_t st, const double* E, long lde, const int* row_index,
                            long cells, long cells_pad, int k0, const double* y, const double* W,
                            long ldw, int c, double* Ep, long ld_ep, double* YE, long ld_ye,
                            double* EE, long ld_ee) {
    hipLaunchKernelGGL(context_features_kernel, dim3((unsigned)cells_pad), dim3(256),
                       sizeof(double) * k0, st, E, lde, row_index, cells, k0, y, W, ldw, c, Ep, ld_ep,
                       YE, ld_ye, EE, ld_ee);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

}  // namespace crm
