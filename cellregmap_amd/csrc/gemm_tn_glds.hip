// FP64 MFMA contraction C = X' * Y (see gemm_tn.hip) with the operand tiles brought in by
// direct-to-LDS loads (global_load_lds_dwordx4): no staging registers and no ds_write pass for the
// Q0 / context / plain-X tiles, which leaves the register file to the accumulators and to operand
// fragments that are read a full k-step ahead of their MFMAs.
//
// A wave-instruction of the LDS-DMA writes 64 x 16 B contiguously (wave-uniform base + lane * 16), so
// the tiles are stored dense, one 1 KiB row of 128 doubles per instruction, and the bank spread that
// padding gave the register-staged kernel comes from the SOURCE side instead: for odd rows a lane
// fetches the 16-byte granule (lane ^ 8), i.e. the two 128-byte halves of every 256-byte bank window
// are swapped, and fragment reads apply the same XOR (column ^ 16 doubles on odd rows).  Consecutive
// rows read by the two 16-lane halves of a ds_read_b64 group then fall on disjoint banks.
#include <algorithm>

#include "crm_common.h"

namespace crm {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef const double __attribute__((address_space(1))) * gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__host__ __device__ inline int glds_kr_variants(int k0) {
    int nb = GEMM_BM / k0 + 2;
    return nb > GEMM_BM ? GEMM_BM : nb;
}

// ECQ: context columns staged per row in units of 32 (round_up(k0, 32) / 32); 0 for the plain kernel
// TR:  store the transpose, C'[n][m] (ldc = row length of C'): the MFMA operands swap roles, so the
//      accumulator tiles come out transposed and the stores stay 128-byte contiguous
template <bool KR, int KRQ, int ECQ, bool TR = false>
__global__ __launch_bounds__(256, 2) void gemm_tn_glds_kernel(const GemmProblem* __restrict__ probs,
                                                               int mtiles_max, long cells_per_split,
                                                               long split_stride, int k0) {
    extern __shared__ __align__(16) double smem[];
    constexpr int BN = 128, NT = 4, LD = 128;
    const GemmProblem P = probs[blockIdx.z];
    const int tile = xcd_tile_id((int)blockIdx.x, (int)gridDim.x);
    const int mtile = tile % mtiles_max;
    const int ntile = tile / mtiles_max;
    const int m0 = mtile * GEMM_BM;
    const int n0 = ntile * BN;
    if (m0 >= P.M || n0 >= P.N) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, lq = lane >> 4;
    const long cell_begin = (long)blockIdx.y * cells_per_split;
    const int stages = (int)(cells_per_split / GEMM_BK);

    // ---- LDS carve-up ------------------------------------------------------------------------
    const int nb = KR ? glds_kr_variants(k0) : 0;
    constexpr int EC = 32 * ECQ;                        // context columns staged per row
    double* Ys = smem;                                  // [2][BK][128]
    double* Xs = Ys + 2 * GEMM_BK * LD;                 // plain: [2][BK][128]
    double* Es = Xs;                                    // KR: [2][BK][EC]
    double* Gs = Es + 2 * GEMM_BK * EC;                 // KR: [2][BK][nb]
    const int b0 = KR ? (m0 / k0) : 0;

    gptr_t Yg = (gptr_t)P.Y + cell_begin * P.ldy + n0;
    gptr_t Xg = KR ? (gptr_t)P.X + cell_begin * P.ldx + b0 : (gptr_t)P.X + cell_begin * P.ldx + m0;
    gptr_t Eg = KR ? (gptr_t)P.E + cell_begin * P.lde : nullptr;
    double rg[KRQ];

    // LDS-DMA of stage s into buffer `buf`: each wavefront issues the rows / pieces w, w+4, ...
    auto issue = [&](int s, int buf) {
        const long roff = (long)s * GEMM_BK;
#pragma unroll
        for (int q = 0; q < GEMM_BK / 4; q++) {
            const int r = wave + 4 * q;
            const int gsw = (lane ^ ((r & 1) << 3)) * 2;  // source granule swap on odd rows
            __builtin_amdgcn_global_load_lds(Yg + (roff + r) * P.ldy + gsw,
                                             (lptr_t)(Ys + (buf * GEMM_BK + r) * LD), 16, 0, 0);
            if (!KR)
                __builtin_amdgcn_global_load_lds(Xg + (roff + r) * P.ldx + gsw,
                                                 (lptr_t)(Xs + (buf * GEMM_BK + r) * LD), 16, 0, 0);
        }
        if (KR) {
            constexpr int ppr = EC / 2;  // 16-byte pieces per row
#pragma unroll
            for (int q = 0; q < ECQ; q++) {       // EC/8 wave-instructions, ECQ per wavefront
                const int ii = wave + 4 * q;
                const int p = ii * 64 + lane;
                const int r = p / ppr, g = p - r * ppr;
                __builtin_amdgcn_global_load_lds(Eg + (roff + r) * P.lde + ((g ^ ((r & 1) << 3)) << 1),
                                                 (lptr_t)(Es + buf * GEMM_BK * EC + ii * 128), 16, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < KRQ; q++) {
                const int e = tid + 256 * q;
                const int row = e / nb, col = e - row * nb;
                if (row < GEMM_BK) rg[q] = Xg[(roff + row) * P.ldx + col];
            }
        }
    };
    auto stash_g = [&](int buf) {
        if (KR) {
#pragma unroll
            for (int q = 0; q < KRQ; q++) {
                const int e = tid + 256 * q;
                if (e < GEMM_BK * nb) Gs[buf * GEMM_BK * nb + e] = rg[q];
            }
        }
    };

    // ---- per-lane fragment addressing ------------------------------------------------------------
    int xa[4], xg[4], xe[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int mloc = wm * 64 + t * 16 + l15;
        xa[t] = mloc;
        if (KR) {
            const int m = m0 + mloc;
            const int b = m / k0;
            const int bl = b - b0;
            xg[t] = bl < nb ? bl : nb - 1;
            xe[t] = m - b * k0;
        }
    }
    const int yb = wn * 64 + l15;

    auto load_raw = [&](int buf, int ks, double (&a)[4], double (&e)[4], double (&b)[NT]) {
        const int row = buf * GEMM_BK + ks * 4 + lq;
        const int sw = (lq & 1) << 4;  // rows ks*4 + lq: parity of the row = parity of lq
#pragma unroll
        for (int t = 0; t < NT; t++) b[t] = Ys[row * LD + ((yb + t * 16) ^ sw)];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if (KR) {
                a[t] = Gs[row * nb + xg[t]];
                e[t] = Es[row * EC + (xe[t] ^ sw)];
            } else {
                a[t] = Xs[row * LD + (xa[t] ^ sw)];
            }
        }
    };
    auto finish = [&](double (&a)[4], const double (&e)[4]) {
        if (KR) {
#pragma unroll
            for (int t = 0; t < 4; t++) a[t] *= e[t];
        }
    };

    v4d acc[4][NT];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < NT; j++) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
    auto mma = [&](const double (&a)[4], const double (&b)[NT]) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < NT; j++)
                acc[i][j] = TR ? __builtin_amdgcn_mfma_f64_16x16x4f64(b[j], a[i], acc[i][j], 0, 0, 0)
                               : __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    };

    constexpr int KS = GEMM_BK / 4;
    double fa[2][4], fb[2][NT], fe[4];
    issue(0, 0);
    stash_g(0);
    __syncthreads();  // drains the LDS-DMA (vmcnt) and publishes the tiles
    load_raw(0, 0, fa[0], fe, fb[0]);
    finish(fa[0], fe);

    for (int s = 0; s < stages; s++) {
        const int buf = s & 1;
        const bool more = s + 1 < stages;
        // buffer buf^1 was last read before the barrier of the previous stage: free for the DMA now
        if (more) issue(s + 1, buf ^ 1);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const int cur = ks & 1, nxt = cur ^ 1;
            const bool have_next = ks + 1 < KS || more;
            if (ks + 1 < KS) {
                load_raw(buf, ks + 1, fa[nxt], fe, fb[nxt]);
            } else {
                if (more) stash_g(buf ^ 1);
                __syncthreads();
                if (more) load_raw(buf ^ 1, 0, fa[nxt], fe, fb[nxt]);
            }
            mma(fa[cur], fb[cur]);
            if (have_next) finish(fa[nxt], fe);
        }
    }

    double* Cb = P.C + (long)blockIdx.y * split_stride;
    if (TR) {
#pragma unroll
        for (int j = 0; j < NT; j++) {
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int nn = n0 + wn * 64 + j * 16 + lq + 4 * reg;
                if (nn < P.N) {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int m = m0 + wm * 64 + i * 16 + l15;
                        if (m < P.M) Cb[(long)nn * P.ldc + m] = acc[i][j][reg];
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int m = m0 + wm * 64 + i * 16 + lq + 4 * reg;
            if (m < P.M) {
#pragma unroll
                for (int j = 0; j < NT; j++) {
                    const int n = n0 + wn * 64 + j * 16 + l15;
                    if (n < P.N) Cb[(long)m * P.ldc + n] = acc[i][j][reg];
                }
            }
        }
    }
}

int launch_gemm_tn_glds(hipStream_t st, const GemmProblem* probs_dev, int nz, int mt, int nt, long cells,
                        bool khatri_rao, int k0, int ksplit, long split_stride, bool transposed_out) {
    if (transposed_out && !khatri_rao) {
        set_error("contraction: the transposed store is only built for the Khatri-Rao form");
        return CRM_ERR_UNSUPPORTED;
    }
    dim3 grid((unsigned)(mt * nt), (unsigned)ksplit, (unsigned)nz);
    size_t lds = (size_t)2 * GEMM_BK * 128 * sizeof(double);
    if (khatri_rao) {
        const int EC = (k0 + 31) / 32 * 32;
        const int nb = glds_kr_variants(k0);
        lds += (size_t)2 * GEMM_BK * (EC + nb) * sizeof(double);
        constexpr int KRQ_BIG = (GEMM_BK * GEMM_BM + 255) / 256;
        const bool small = GEMM_BK * nb <= 256;
#define CRM_GLDS_T(Q, T)                                                                                      \
    do {                                                                                                      \
        if (small)                                                                                            \
            hipLaunchKernelGGL((gemm_tn_glds_kernel<true, 1, Q, T>), grid, dim3(256), lds, st, probs_dev, mt, \
                               cells / ksplit, split_stride, k0);                                             \
        else                                                                                                  \
            hipLaunchKernelGGL((gemm_tn_glds_kernel<true, KRQ_BIG, Q, T>), grid, dim3(256), lds, st,          \
                               probs_dev, mt, cells / ksplit, split_stride, k0);                              \
    } while (0)
#define CRM_GLDS(Q)                              \
    do {                                         \
        if (transposed_out) CRM_GLDS_T(Q, true); \
        else CRM_GLDS_T(Q, false);               \
    } while (0)
        switch (EC / 32) {
            case 1: CRM_GLDS(1); break;
            case 2: CRM_GLDS(2); break;
            case 3: CRM_GLDS(3); break;
            default: CRM_GLDS(4); break;
        }
#undef CRM_GLDS
#undef CRM_GLDS_T
    } else {
        lds += (size_t)2 * GEMM_BK * 128 * sizeof(double);
        hipLaunchKernelGGL((gemm_tn_glds_kernel<false, 1, 0>), grid, dim3(256), lds, st, probs_dev, mt,
                           cells / ksplit, split_stride, 0);
    }
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

}  // namespace crm
