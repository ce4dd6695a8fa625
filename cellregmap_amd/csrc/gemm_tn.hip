// FP64 MFMA contraction over the cell axis:  C = X' * Y   (gfx950 / CDNA4).
//
// Every large product of the score-test path has this one shape -- both operands are
// row-major [cells x columns] matrices and the sum runs over cells:
//     T(rho)  = G_block'            * Q0(rho)      (null-fit rotations)
//     A~      = KR(G_block, E0)'    * Q0(rho*)     (the dominant term, 2*n*r*k0 flop/variant, a10)
//     side reductions  G'[y o E, W o E],  (G o G)'[E, E (x) E]            (a9, a10)
// KR(G, E)[i, b*k0 + j] = G[i, b] * E[i, j] (Khatri-Rao columns) is never materialised:
// the G and E tiles are staged in LDS separately and multiplied while the A fragment of
// v_mfma_f64_16x16x4_f64 is assembled.
//
// Tiling: 128 x 128 output tile per 256-thread workgroup, 4 wavefronts as 2 x 2, each
// wavefront owns 64 x 64 = 4 x 4 MFMA tiles (16 accumulators of 4 f64 -> 128 registers);
// cell axis in stages of 8 rows, LDS double-buffered, next stage prefetched into registers
// while the current one feeds the matrix pipe.  Operand tiles are stored [stage row][column]
// with a row stride of 144 doubles (= 128 B mod 256 B) so that the four 16-lane groups of a
// ds_read_b64 fragment read hit disjoint bank halves.
#include "crm_internal.h"

namespace crm {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int LDT = GEMM_BM + 16;  // LDS row stride of a plain operand tile (doubles)

__host__ __device__ inline int kr_variants_per_tile(int k0) {
    int nb = GEMM_BM / k0 + 2;
    return nb > GEMM_BM ? GEMM_BM : nb;
}
__host__ __device__ inline int kr_e_stride(int k0) {
    // smallest stride >= k0 with stride % 32 == 16 (doubles): consecutive rows fall on
    // opposite halves of the 64 LDS banks
    int s = (k0 + 15) / 32 * 32 + 16;
    if (s < k0) s += 32;
    return s;
}

typedef const double __attribute__((address_space(1))) * gptr_t;   // global (not flat) loads
typedef const v2d __attribute__((address_space(1))) * gptr2_t;

// KRQ: 8-byte prefetch slots per thread for the G tile of the Khatri-Rao operand
// EW: 1 = context tiles of up to 128 columns; 2 = up to 256 (the slower form for 129 .. 256 contexts, one workgroup per CU)
// transposed: store C' (N x M, leading dimension ldc)
template <bool KR, int KRQ, int BN, int EW = 1>
__global__ __launch_bounds__(256, (EW == 2 ? 1 : BN == 64 ? 3 : 2)) void gemm_tn_kernel(const GemmProblem* __restrict__ probs,
                                                          int mtiles_max, long cells_per_split, long cells_total,
                                                          long split_stride, int k0, int transposed) {
    extern __shared__ __align__(16) double smem[];
    const GemmProblem P = probs[blockIdx.z];
    const int tile = xcd_tile_id((int)blockIdx.x, (int)gridDim.x);
    const int mtile = tile % mtiles_max;
    const int ntile = tile / mtiles_max;
    const int m0 = mtile * GEMM_BM;
    const int n0 = ntile * BN;
    constexpr int NT = BN / 32;          // 16-wide B fragments per wavefront (wave tile 64 x BN/2)
    constexpr int LDY = BN + 16;         // LDS row stride of the Y tile
    if (m0 >= P.M || n0 >= P.N) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, lq = lane >> 4;

    const long cell_begin = (long)blockIdx.y * cells_per_split;
    // (the last slice of a split over the cell axis may be shorter)
    const long cells_mine = P.cells > 0 ? P.cells : cells_total;
    const int stages = (int)(std::min(cells_per_split, cells_mine - cell_begin) / GEMM_BK);

    // ---- LDS carve-up (two stages of every tile) --------------------------------------
    const int nb = KR ? kr_variants_per_tile(k0) : 0;
    const int lde_s = KR ? kr_e_stride(k0) : 0;
    double* Ys = smem;                                  // [2][BK][LDY]
    double* Xs = Ys + 2 * GEMM_BK * LDY;                // plain: [2][BK][LDT]
    double* Et = Xs;                                    // KR: [2][BK][lde_s]
    double* Gt = Et + 2 * GEMM_BK * lde_s;              // KR: [2][BK][nb]

    // ---- global -> register prefetch ----------------------------------------------------
    // plain tiles: BK x 128 doubles = BK*64 16-byte pieces, 256 threads -> BK/4 pieces each
    constexpr int NPF = GEMM_BK / 4;                 // X tile: 64 pieces per row, 4 rows per pass
    constexpr int NPY = GEMM_BK * BN / 512;          // Y tile: BN/2 pieces per row
    constexpr int YROWS = 512 / BN;                  // rows covered per pass (4 or 8)
    const int ld_row = tid >> 6;        // + 4 per piece
    const int ld_col = (tid & 63) * 2;
    const int ly_row = tid / (BN / 2);
    const int ly_col = (tid % (BN / 2)) * 2;
    v2d ry[NPY], rx[NPF];
    // KR context tile: BK rows x round32(k0) columns (P.lde >= that, zero padded), 16-byte pieces
    const int e_pieces_row = KR ? ((k0 + 31) / 32 * 16) : 1;  // 16-byte pieces per row (k0 rounded to 32)
    constexpr int NPE = EW * GEMM_BK / 4;                   // pieces per thread (lde <= 128 EW)
    v2d re[NPE];
    double rg[KRQ];
    const int b0 = KR ? (m0 / k0) : 0;

    gptr_t Yp = (gptr_t)P.Y + (cell_begin + ly_row) * P.ldy + n0 + ly_col;
    gptr_t Xp = KR ? (gptr_t)P.X + cell_begin * P.ldx + b0
                   : (gptr_t)P.X + (cell_begin + ld_row) * P.ldx + m0 + ld_col;
    gptr_t Ep = KR ? (gptr_t)P.E + cell_begin * P.lde : nullptr;

    auto fetch = [&](int s) {
        const long roff = (long)s * GEMM_BK;
#pragma unroll
        for (int q = 0; q < NPY; q++) ry[q] = *(gptr2_t)(Yp + (roff + YROWS * q) * P.ldy);
        if (!KR) {
#pragma unroll
            for (int q = 0; q < NPF; q++) rx[q] = *(gptr2_t)(Xp + (roff + 4 * q) * P.ldx);
        }
        if (KR) {
#pragma unroll
            for (int q = 0; q < NPE; q++) {
                const int e = tid + 256 * q;          // piece index inside the BK x lde tile
                const int row = e / e_pieces_row, pc = e - row * e_pieces_row;
                if (row < GEMM_BK) re[q] = *(gptr2_t)(Ep + (roff + row) * P.lde + 2 * pc);
            }
#pragma unroll
            for (int q = 0; q < KRQ; q++) {
                const int e = tid + 256 * q;
                const int row = e / nb, col = e - row * nb;
                if (row < GEMM_BK) rg[q] = Xp[(roff + row) * P.ldx + col];
            }
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int q = 0; q < NPY; q++)
            *reinterpret_cast<v2d*>(Ys + (buf * GEMM_BK + ly_row + YROWS * q) * LDY + ly_col) = ry[q];
        if (!KR) {
#pragma unroll
            for (int q = 0; q < NPF; q++)
                *reinterpret_cast<v2d*>(Xs + (buf * GEMM_BK + ld_row + 4 * q) * LDT + ld_col) = rx[q];
        }
        if (KR) {
#pragma unroll
            for (int q = 0; q < NPE; q++) {
                const int e = tid + 256 * q;
                const int row = e / e_pieces_row, pc = e - row * e_pieces_row;
                if (row < GEMM_BK && 2 * pc < lde_s)
                    *reinterpret_cast<v2d*>(Et + (buf * GEMM_BK + row) * lde_s + 2 * pc) = re[q];
            }
#pragma unroll
            for (int q = 0; q < KRQ; q++) {
                const int e = tid + 256 * q;
                if (e < GEMM_BK * nb) Gt[buf * GEMM_BK * nb + e] = rg[q];
            }
        }
    };

    // ---- per-lane fragment addressing ----------------------------------------------------
    int xa[4], xg[4], xe[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        int mloc = wm * 64 + t * 16 + l15;
        xa[t] = mloc;
        if (KR) {
            int m = m0 + mloc;
            int b = m / k0;
            int j = m - b * k0;
            int bl = b - b0;
            xg[t] = bl < nb ? bl : nb - 1;  // rows past the group's end are never stored
            xe[t] = j;
        }
    }
    const int yb = wn * (BN / 2) + l15;

    // Fragment loads are split in two so that the LDS reads of k-step ks+1 can be issued before the
    // MFMAs of k-step ks and the Khatri-Rao products taken after them (the products would otherwise
    // sit, with their LDS latency, in front of the matrix instructions).
    auto load_raw = [&](int buf, int ks, double (&a)[4], double (&e)[4], double (&b)[NT]) {
        const int row = buf * GEMM_BK + ks * 4 + lq;
#pragma unroll
        for (int t = 0; t < NT; t++) b[t] = Ys[row * LDY + yb + t * 16];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if (KR) {
                a[t] = Gt[row * nb + xg[t]];
                e[t] = Et[row * lde_s + xe[t]];
            } else {
                a[t] = Xs[row * LDT + xa[t]];
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
                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    };

    // Software pipeline, one barrier per stage: global loads of stage s+1 are issued first, land in
    // LDS (other buffer) after the last-but-one k-step of the stage, the barrier sits before the last
    // k-step, and the first fragments of stage s+1 are read while that k-step runs.
    constexpr int KS = GEMM_BK / 4;  // k-steps per stage
    constexpr int STASH_AFTER = KS - 2;  // latest point that still precedes the stage's barrier
    static_assert(KS >= 2, "pipeline needs at least two k-steps per stage");
    double fa[2][4], fb[2][NT], fe[4];
    fetch(0);
    stash(0);
    __syncthreads();
    load_raw(0, 0, fa[0], fe, fb[0]);
    finish(fa[0], fe);

    for (int s = 0; s < stages; s++) {
        const int buf = s & 1;
        const bool more = s + 1 < stages;
        if (more) fetch(s + 1);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const int cur = ks & 1, nxt = cur ^ 1;
            const bool have_next = ks + 1 < KS || more;
            if (ks + 1 < KS) {
                load_raw(buf, ks + 1, fa[nxt], fe, fb[nxt]);
            } else {
                __syncthreads();  // every wave has stashed stage s+1 and is done reading `buf^1`
                if (more) load_raw(buf ^ 1, 0, fa[nxt], fe, fb[nxt]);
            }
            mma(fa[cur], fb[cur]);
            if (have_next) finish(fa[nxt], fe);
            if (ks == STASH_AFTER && more) stash(buf ^ 1);
        }
    }

    // ---- epilogue: D[row = lq + 4*reg][col = l15] per 16x16 tile ---------------------------
    double* Cb = P.C + (long)blockIdx.y * split_stride;
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int m = m0 + wm * 64 + i * 16 + lq + 4 * reg;
            if (m < P.M) {
#pragma unroll
                for (int j = 0; j < NT; j++) {
                    const int n = n0 + wn * (BN / 2) + j * 16 + l15;
                    if (n < P.N) {
                        double* cp = transposed ? Cb + (long)n * P.ldc + m : Cb + (long)m * P.ldc + n;
                        *cp = (P.flags & GEMM_SUBTRACT) ? *cp - acc[i][j][reg] : acc[i][j][reg];
                    }
                }
            }
        }
    }
}

__global__ void reduce_splits_kernel(double* __restrict__ C, long count, int ksplit, long stride) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    double s = C[i];
    for (int k = 1; k < ksplit; k++) s += C[i + k * stride];
    C[i] = s;
}

// the same for a band of columns [col0, col0 + ncols) of a row-major matrix with leading dimension ld
__global__ void reduce_splits_band_kernel(double* __restrict__ C, long rows, long ld, int col0, int ncols, int ksplit,
                                          long stride) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * ncols) return;
    const long row = i / ncols;
    const int col = (int)(i - row * ncols);
    double* p = C + row * ld + col0 + col;
    double s = *p;
    for (int k = 1; k < ksplit; k++) s += p[k * stride];
    *p = s;
}

int split_for(long cells_pad, long blocks_without_split) {
    // enough workgroups to cover the 256 CUs twice, at least eight stages per slice, every slice non-empty
    const long stages = cells_pad / GEMM_BK;
    // (long contractions: four rounds' worth, which also moves the launch onto the 128-wide LDS-DMA tiles -- measured
    // +2 % on a mode-B step at config 3; nothing to gain at 5 000 cells)
    const long target = cells_pad >= 16384 ? 1024 : 512;
    long want = (target + blocks_without_split - 1) / std::max<long>(blocks_without_split, 1);
    want = std::min<long>(std::min<long>(want, 64), std::max<long>(stages / 8, 1));
    if (want <= 1) return 1;
    const long per = (stages + want - 1) / want;
    return (int)((stages + per - 1) / per);
}

// Output-tile width of a launch: 128-wide tiles (two workgroups per CU) give the Khatri-Rao operand the most MFMAs per
// LDS read; 64-wide tiles (three per CU) fill the chip better when a launch has few tiles or a narrow output.
int contraction_tile_width(const crm_ctx* ctx, int mt, int max_n, int nz, int ksplit, bool khatri_rao) {
    int bn = ctx->tune.bn;
    if (bn != 64 && bn != 128 && bn != 160) {
        const long tiles128 = (long)mt * ((max_n + 127) / 128) * nz * ksplit;
        bn = tiles128 < 1024 ? 64 : 128;
        // narrow outputs (modes A / B: N = k1 + m columns): 64-wide tiles when they waste less padding, one 160-wide
        // tile for 129 .. 160 columns (mode B at config 3: 150 columns fill it to 94 %, three 64-wide tiles to 78 %)
        if (khatri_rao && ((max_n + 63) / 64) * 64 * 100 <= ((max_n + 127) / 128) * 128 * 85) bn = 64;
        if (khatri_rao && ctx->tune.glds && max_n > 128 && max_n <= 160 && (long)mt * nz * ksplit >= 256) bn = 160;
    }
    if (bn == 160 && !(khatri_rao && ctx->tune.glds)) bn = 128;
    return bn;
}

// Slices along the cell axis for a Khatri-Rao launch of `tiles128` row tiles x column tiles: a launch of a few rounds
// of workgroups loses the unfilled part of its last round (mode B at config 3: 1600 workgroups on 512 slots = 3.1
// rounds, a quarter of the time in a round that is 12 % full); cutting the cell axis makes the rounds shorter.
int kr_split_for(const crm_ctx* ctx, long row_tiles, int max_n, int nz, long cells_pad, int max_split) {
    const int bn = contraction_tile_width(ctx, (int)std::min<long>(row_tiles, 1 << 20), max_n, nz, 1, true);
    const long slots = 256L * (bn == 64 ? 3 : 2);
    const long tiles = row_tiles * ((max_n + bn - 1) / bn);
    if (tiles <= 0 || tiles > 6 * slots) return 1;
    const long stages = cells_pad / GEMM_BK;
    int best = 1;
    double best_eff = 0.0;
    for (int ks : {1, 2, 3, 4, 6, 8}) {
        if (ks > max_split || stages / ks < 32) break;
        const long w = tiles * ks, rounds = (w + slots - 1) / slots;
        const double eff = (double)w / (double)(rounds * slots);
        if (eff > best_eff + (ks == 1 ? 0.0 : 0.04)) { best_eff = eff; best = ks; }
    }
    return best;
}

// 129 .. CRM_MAX_K0 contexts: the staged kernel with context tiles twice as wide (no LDS-DMA form; one workgroup per CU).  The
// slower correct path for context counts past the tiles of the fast kernels.
static int launch_kr_wide(crm_ctx* ctx, const GemmProblem* probs_dev, int nz, int max_m, int max_n, long cells, int k0,
                          int ksplit, long split_stride, bool transposed) {
    const int bn = max_n <= 64 ? 64 : 128;
    const int mt = (max_m + GEMM_BM - 1) / GEMM_BM, nt = (max_n + bn - 1) / bn;
    const long total_stages = cells / GEMM_BK;
    const long cps = (total_stages + ksplit - 1) / ksplit * GEMM_BK;
    dim3 grid((unsigned)(mt * nt), (unsigned)ksplit, (unsigned)nz);
    const size_t lds = sizeof(double) * 2 * GEMM_BK * ((size_t)(bn + 16) + kr_variants_per_tile(k0) + kr_e_stride(k0));
    if (bn == 64) {
        CRM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_kernel<true, 1, 64, 2>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((gemm_tn_kernel<true, 1, 64, 2>), grid, dim3(256), lds, ctx->stream, probs_dev, mt, cps, cells,
                           split_stride, k0, transposed ? 1 : 0);
    } else {
        CRM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_kernel<true, 1, 128, 2>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((gemm_tn_kernel<true, 1, 128, 2>), grid, dim3(256), lds, ctx->stream, probs_dev, mt, cps, cells,
                           split_stride, k0, transposed ? 1 : 0);
    }
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

int launch_kr_transposed(crm_ctx* ctx, const GemmProblem* probs_dev, int nz, int max_m, int max_n, long cells,
                         int k0) {
    if (nz <= 0 || max_m <= 0 || max_n <= 0) return CRM_OK;
    if (cells % GEMM_BK != 0 || k0 < 1 || k0 > CRM_MAX_K0) {
        set_error("transposed Khatri-Rao contraction: cells=%ld, k0=%d", cells, k0);
        return CRM_ERR_ARG;
    }
    if (k0 > 128) return launch_kr_wide(ctx, probs_dev, nz, max_m, max_n, cells, k0, 1, 0, true);
    // outputs of at most 64 columns (the [us | E1] rows of the kinship-structure route at config 2: 40, in mode B: 51) through
    // the 64-wide tile: a 128-wide one would be less than half full
    const int bn = max_n <= 64 && ctx->tune.bn != 128 ? 64 : 128;
    return launch_gemm_tn_glds(ctx, probs_dev, nz, (max_m + GEMM_BM - 1) / GEMM_BM, (max_n + bn - 1) / bn, cells, true,
                               k0, 1, 0, true, bn);
}

int launch_gemm_tn(crm_ctx* ctx, const GemmProblem* probs_dev, int nz, int max_m, int max_n,
                   long cells, bool khatri_rao, int k0, int ksplit, long split_stride) {
    if (nz <= 0 || max_m <= 0 || max_n <= 0) return CRM_OK;
    hipStream_t st = ctx->stream;
    if (ksplit < 1) ksplit = 1;
    // slices of ceil(stages / ksplit) stages; every slice must hold at least one (see split_for)
    const long total_stages = cells / GEMM_BK;
    const long cps = (total_stages + ksplit - 1) / ksplit * GEMM_BK;
    if (cells % GEMM_BK != 0 || (long)(ksplit - 1) * cps >= cells) {
        set_error("contraction: %ld cells cannot be cut into %d slices of whole stages", cells, ksplit);
        return CRM_ERR_ARG;
    }
    if (khatri_rao && (k0 < 1 || k0 > CRM_MAX_K0)) {
        set_error("Khatri-Rao contraction supports 1 <= k0 <= %d (got %d)", CRM_MAX_K0, k0);
        return CRM_ERR_UNSUPPORTED;
    }
    if (khatri_rao && k0 > 128) return launch_kr_wide(ctx, probs_dev, nz, max_m, max_n, cells, k0, ksplit, split_stride, false);
    const int mt = (max_m + GEMM_BM - 1) / GEMM_BM;
    const int bn = contraction_tile_width(ctx, mt, max_n, nz, ksplit, khatri_rao);
    const int nt = (max_n + bn - 1) / bn;
    if (ctx->tune.glds && (bn == 128 || khatri_rao)) {
        return launch_gemm_tn_glds(ctx, probs_dev, nz, mt, nt, cells, khatri_rao, k0, ksplit, split_stride, false, bn);
    }
    dim3 grid((unsigned)(mt * nt), (unsigned)ksplit, (unsigned)nz);
    size_t lds = (size_t)2 * GEMM_BK * (bn + 16) * sizeof(double);
    if (khatri_rao) {
        lds += (size_t)2 * GEMM_BK * (kr_variants_per_tile(k0) + kr_e_stride(k0)) * sizeof(double);
        constexpr int KRQ_BIG = (GEMM_BK * GEMM_BM + 255) / 256;
        const bool small = GEMM_BK * kr_variants_per_tile(k0) <= 256;
        if (bn == 64) {
            if (small)
                hipLaunchKernelGGL((gemm_tn_kernel<true, 1, 64>), grid, dim3(256), lds, st, probs_dev, mt,
                                   cps, cells, split_stride, k0, 0);
            else
                hipLaunchKernelGGL((gemm_tn_kernel<true, KRQ_BIG, 64>), grid, dim3(256), lds, st, probs_dev, mt,
                                   cps, cells, split_stride, k0, 0);
        } else {
            if (small)
                hipLaunchKernelGGL((gemm_tn_kernel<true, 1, 128>), grid, dim3(256), lds, st, probs_dev, mt,
                                   cps, cells, split_stride, k0, 0);
            else
                hipLaunchKernelGGL((gemm_tn_kernel<true, KRQ_BIG, 128>), grid, dim3(256), lds, st, probs_dev, mt,
                                   cps, cells, split_stride, k0, 0);
        }
    } else {
        lds += (size_t)2 * GEMM_BK * LDT * sizeof(double);
        if (bn == 64)
            hipLaunchKernelGGL((gemm_tn_kernel<false, 1, 64>), grid, dim3(256), lds, st, probs_dev, mt,
                               cps, cells, split_stride, 0, 0);
        else
            hipLaunchKernelGGL((gemm_tn_kernel<false, 1, 128>), grid, dim3(256), lds, st, probs_dev, mt,
                               cps, cells, split_stride, 0, 0);
    }
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

// ---- C = X'Y for a narrow Y (N <= 16 columns) -------------------------------------------------------------------------
// The last few columns of a product whose width is a little more than a multiple of the 128-column tile (scan.hip: the
// spectrum's 5 000 = 39 tiles + 8) would cost a whole column of tiles -- 1/40 of the launch -- in the tiled kernels.  Here
// they are one pass over X: a wavefront takes 32 columns of X (16-byte loads: columns m, m + 1 per lane, four rows per
// instruction) as the A operands of two v_mfma_f64_16x16x4_f64 per k-step against the same four rows of Y; bound by the
// traffic of X.  grid (ceil(max_m / 128), problems), 256 threads.
typedef double sk_v4d __attribute__((ext_vector_type(4)));
typedef double sk_v2d __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void skinny_tn_kernel(const GemmProblem* __restrict__ probs, long cells_total) {
    const GemmProblem P = probs[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
    const long m0 = (long)blockIdx.x * 128 + 32 * wave;
    if (m0 >= P.M) return;
    const long cells = P.cells > 0 ? P.cells : cells_total;
    const long m = m0 + 2 * l15;                       // this lane's two columns of X (ldx even, X 16-byte aligned)
    const bool two = m + 1 < P.M, one = m < P.M, nok = l15 < P.N;
    const double* __restrict__ x = P.X + (long)lq * P.ldx + m;
    const double* __restrict__ y = P.Y + (long)lq * P.ldy + l15;
    sk_v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    constexpr int U = 8;                               // k-steps per trip, every load issued before the first MFMA
    long k = 0;
    for (; k + 4 * U <= cells; k += 4 * U) {
        sk_v2d a[U];
        double b[U];
#pragma unroll
        for (int q = 0; q < U; q++) {
            const double* xr = x + (k + 4 * q) * P.ldx;
            a[q] = two ? *reinterpret_cast<const sk_v2d*>(xr) : (sk_v2d){one ? xr[0] : 0.0, 0.0};
            b[q] = nok ? y[(k + 4 * q) * P.ldy] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < U; q++) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][0], b[q], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][1], b[q], acc1, 0, 0, 0);
        }
    }
    for (; k < cells; k += 4) {
        const bool kok = k + lq < cells;
        const double* xr = x + k * P.ldx;
        const double a0 = kok && one ? xr[0] : 0.0, a1 = kok && two ? xr[1] : 0.0;
        const double b = kok && nok ? y[k * P.ldy] : 0.0;
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b, acc1, 0, 0, 0);
    }
    // accumulator rows are the A operand's row index: row (lq + 4 reg) <-> columns m0 + 2 (lq + 4 reg) (+ 1) of X
    if (l15 < P.N) {
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const long row = m0 + 2 * (lq + 4 * reg);
            if (row < P.M) P.C[row * P.ldc + l15] = acc0[reg];
            if (row + 1 < P.M) P.C[(row + 1) * P.ldc + l15] = acc1[reg];
        }
    }
}

int launch_skinny_tn(hipStream_t st, const GemmProblem* probs_dev, int nz, int max_m, long cells) {
    if (nz <= 0 || max_m <= 0) return CRM_OK;
    hipLaunchKernelGGL(skinny_tn_kernel, dim3((unsigned)((max_m + 127) / 128), (unsigned)nz), dim3(256), 0, st, probs_dev, cells);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

int launch_reduce_splits(hipStream_t st, double* C, long count, int ksplit, long split_stride) {
    if (ksplit <= 1 || count <= 0) return CRM_OK;
    hipLaunchKernelGGL(reduce_splits_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st,
                       C, count, ksplit, split_stride);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

int launch_reduce_splits_band(hipStream_t st, double* C, long rows, long ld, int col0, int ncols, int ksplit,
                              long split_stride) {
    if (ksplit <= 1 || rows <= 0 || ncols <= 0) return CRM_OK;
    hipLaunchKernelGGL(reduce_splits_band_kernel, dim3((unsigned)((rows * ncols + 255) / 256)), dim3(256), 0, st, C, rows,
                       ld, col0, ncols, ksplit, split_stride);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

}  // namespace crm
