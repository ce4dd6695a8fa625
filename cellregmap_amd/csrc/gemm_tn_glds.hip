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
#include <type_traits>

#include "crm_internal.h"

namespace crm {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef const double __attribute__((address_space(1))) * gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef const double __attribute__((address_space(3))) * lcptr_t;

typedef const char __attribute__((address_space(1))) * gbptr_t;
__device__ inline gptr_t at_bytes(gptr_t base, unsigned byte_off) { return (gptr_t)((gbptr_t)base + byte_off); }
// a wave-uniform global pointer pinned to scalar registers (keeps `base + lane offset` in the
// saddr + voffset form of the load instead of a per-lane 64-bit add)
// a pointer the compiler already keeps in scalar registers, made opaque there: without it the 64-bit adds of
// `base + stage offset + lane offset` are re-associated into per-lane vector adds and the saddr form is lost
__device__ inline gptr_t opaque_scalar(gptr_t p) {
    unsigned long v = (unsigned long)p;
    asm volatile("" : "+s"(v));
    return (gptr_t)v;
}
__device__ inline gptr_t scalar_ptr(gptr_t p) {
    const unsigned long v = (unsigned long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (gptr_t)(((unsigned long)hi << 32) | lo);
}

__host__ __device__ inline int glds_kr_variants(int k0) {
    int nb = GEMM_BM / k0 + 2;
    return nb > GEMM_BM ? GEMM_BM : nb;
}

// ECQ: context columns staged per row in units of 32 (round_up(k0, 32) / 32); 0 for the plain kernel
// KRQ: 1 when at most 16 genotype columns are staged per row (k0 >= 10), else the 128-column form
// TR:  store the transpose, C'[n][m] (ldc = row length of C'): the MFMA operands swap roles, so the
//      accumulator tiles come out transposed and the stores stay 128-byte contiguous
//
// Instruction-level layout of the main loop.  rocprofv3 shows SQ_VALU_MFMA_COEXEC_CYCLES = 0 for this
// kernel: an FP64 MFMA and a vector-ALU instruction never overlap, so every VALU instruction in the loop
// is taken out of the matrix pipe's time.  Hence: the stage loop is unrolled by two so that the LDS
// buffer index is a compile-time constant and every fragment read is `ds_read_b64 v, vbase offset:imm`
// off twelve loop-invariant per-lane bases; the global side uses wave-uniform bases advanced on the
// scalar unit plus loop-invariant 32-bit lane offsets (`saddr + voffset`); what is left per stage of 64
// MFMAs is the eight Khatri-Rao operand products (wave tile 32 x 128: two operand fragments per k-step).
// The reads / operand products of the next k-step are spread between the sixteen MFMAs of the current one
// (sched_group_barrier).
template <int KRQ>
struct GldsGeno {
    static constexpr int LD = KRQ == 1 ? 16 : 128;  // LDS row length of the staged genotype columns
};

template <bool KR, int KRQ, int ECQ, bool TR, int BN>
__device__ __forceinline__ void glds_tile(double* smem, const GemmProblem& P, int mtile, int ntile, int slice,
                                          long cells_per_split, long cells_total, long split_stride, int k0) {
    static_assert(BN == 128 || ((BN == 64 || BN == 160) && KR), "64- and 160-wide tiles are built for the Khatri-Rao form");
    constexpr int LD = BN;
    // Wave tile: 64 x 64 (wavefronts 2 x 2) for the plain product -- fewest fragment reads per MFMA; 32 x
    // 128 (wavefronts 4 x 1) for the Khatri-Rao form -- per k-step two operand products instead of four
    // (each one a VALU instruction that the FP64 matrix pipe cannot overlap), same twelve reads.
    constexpr int MT = KR ? 2 : 4, NT = KR ? BN / 16 : 4;   // 16-row / 16-column fragments per wavefront
    constexpr int WROWS = 16 * MT, WCOLS = 16 * NT;
    constexpr int EC = 32 * ECQ;                        // context columns staged per row
    constexpr int GLD = GldsGeno<KRQ>::LD;
    constexpr int KS = GEMM_BK / 4;
    const int m0 = mtile * GEMM_BM;
    const int n0 = ntile * BN;
    if (m0 >= P.M || n0 >= P.N) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = KR ? wave : wave >> 1, wn = KR ? 0 : wave & 1;
    const int l15 = lane & 15, lq = lane >> 4;
    const long cell_begin = (long)slice * cells_per_split;
    // (the last slice of a split over the cell axis may be shorter)
    const long cells_mine = P.cells > 0 ? P.cells : cells_total;
    const int stages = (int)(std::min(cells_per_split, cells_mine - cell_begin) / GEMM_BK);

    // ---- LDS carve-up ------------------------------------------------------------------------
    const int nb = KR ? glds_kr_variants(k0) : 0;
    double* const Ys = smem;                            // [2][BK][128]
    double* const Xs = Ys + 2 * GEMM_BK * LD;           // plain: [2][BK][128]
    double* const Es = Xs;                              // KR: [2][BK][EC]
    double* const Gs = Es + 2 * GEMM_BK * EC;           // KR: [2][BK][GLD]
    const int b0 = KR ? (m0 / k0) : 0;

    gptr_t Yg = (gptr_t)P.Y + cell_begin * P.ldy + n0;
    gptr_t Xg = KR ? (gptr_t)P.X + cell_begin * P.ldx + b0 : (gptr_t)P.X + cell_begin * P.ldx + m0;
    gptr_t Eg = KR ? (gptr_t)P.E + cell_begin * P.lde : nullptr;
    double rg[KRQ];

    // LDS-DMA of stage s into buffer BUF: each wavefront issues the rows / pieces w, w+4, ...
    // Every global address is a wave-uniform 64-bit base (advanced on the scalar unit) plus a per-lane
    // 32-bit offset fixed before the loop -- the `saddr + voffset` form, no vector arithmetic per stage.
    // bytes; source granule swap on odd rows.  64-wide tiles: one wave-instruction carries two 512-byte rows
    // (lanes 0-31 the even one, lanes 32-63 the odd one)
    const unsigned y_lane = BN == 128 ? 8u * (unsigned)((lane ^ ((wave & 1) << 3)) * 2)
                                      : (unsigned)(lane >> 5) * (unsigned)(P.ldy * 8) +
                                            16u * (unsigned)((lane & 31) ^ ((lane >> 5) << 3));
    // 160-wide tiles (outputs of 129 .. 160 columns, e.g. mode B's k1 + m = 150): a stage of the Y tile is 16 rows x
    // 1280 bytes = 20 wave-instructions of 64 consecutive 16-byte pieces of the dense LDS image, 5 per wavefront;
    // piece p sits in row p / 80 (1280 bytes = 5 bank windows, so the odd-row granule swap works as for 128)
    constexpr int YQ = BN == 160 ? 5 : 1;
    unsigned y_piece[YQ];
    if (BN == 160) {
#pragma unroll
        for (int q = 0; q < YQ; q++) {
            const int p = (wave + 4 * q) * 64 + lane;
            const int r = p / 80, g = p - r * 80;
            // A 160-column tile on rows shorter than n0 + 160 doubles (a 128-column operand forced through this
            // kernel) would read past the row -- on the operand's last row past its allocation, an illegal access
            // whenever the buffer ends on a page boundary (round 2's intermittent abort of the GPU suite).  Pieces
            // beyond the row fetch its first granule instead: they only feed output columns >= N, never stored.
            const int col = (g ^ ((r & 1) << 3)) << 1;
            y_piece[q] = 8u * (unsigned)(r * (int)P.ldy + ((long)n0 + col + 2 <= P.ldy ? col : 0));
        }
    }
    constexpr int ECN = ECQ > 0 ? ECQ : 1;
    unsigned e_lane[ECN];
    unsigned g_lane[KRQ];
    if (KR) {
        constexpr int ppr = EC > 0 ? EC / 2 : 1;  // 16-byte pieces per row
#pragma unroll
        for (int q = 0; q < ECQ; q++) {
            const int p = (wave + 4 * q) * 64 + lane;
            const int r = p / ppr, g = p - r * ppr;
            e_lane[q] = 8u * (unsigned)(r * (int)P.lde + ((g ^ ((r & 1) << 3)) << 1));
        }
#pragma unroll
        for (int q = 0; q < KRQ; q++) {
            const int e = tid + 256 * q;
            const int row = e / GLD, col = e - row * GLD;
            g_lane[q] = 8u * (unsigned)(row * (int)P.ldx + (col < nb ? col : 0));  // columns >= nb: never read back
        }
    }
    // row bases of stage 0 for the rows this wavefront fetches, pinned to scalar registers once: inside the loop they
    // advance by whole stages on the scalar unit (left to the compiler, the wavefront's own row offset stayed in vector
    // registers and every row cost a 64-bit vector add and two v_readfirstlane -- 30 vector instructions per stage of the
    // plain kernel, each taken out of the matrix pipe's time)
    constexpr int YROWS_W = BN == 128 ? 1 : 2, YQN = BN == 160 ? 1 : GEMM_BK / (4 * YROWS_W);
    gptr_t Yw[YQN], Xw[YQN];
#pragma unroll
    for (int q = 0; q < YQN; q++) {
        const int r = (wave + 4 * q) * YROWS_W;
        Yw[q] = scalar_ptr(Yg + (long)r * P.ldy);
        Xw[q] = KR ? nullptr : scalar_ptr(Xg + (long)r * P.ldx);
    }
    auto issue = [&](int BUF, int s) __attribute__((always_inline)) {
        const long roff = (long)s * GEMM_BK;
        // (re-defined inside the loop body so that the zero-extension stays next to the load and the
        // instruction selector can fold it into the voffset operand)
        unsigned yl = y_lane;
        asm volatile("" : "+v"(yl));
        constexpr int YROWS = BN == 128 ? 1 : 2;  // rows per wave-instruction
        if (BN == 160) {
            gptr_t ybase = scalar_ptr(Yg + roff * P.ldy);
#pragma unroll
            for (int q = 0; q < YQ; q++) {
                unsigned yo_q = y_piece[q];
                asm volatile("" : "+v"(yo_q));
                __builtin_amdgcn_global_load_lds(at_bytes(ybase, yo_q), (lptr_t)(Ys + BUF * GEMM_BK * LD + (wave + 4 * q) * 128), 16,
                                                 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < (BN == 160 ? 0 : GEMM_BK / (4 * YROWS)); q++) {
            const int r = (wave + 4 * q) * YROWS;
            gptr_t yrow = opaque_scalar(Yw[q] + roff * P.ldy);
            __builtin_amdgcn_global_load_lds(at_bytes(yrow, yl), (lptr_t)(Ys + (BUF * GEMM_BK + r) * LD), 16, 0, 0);
            if (!KR) {
                gptr_t xrow = opaque_scalar(Xw[q] + roff * P.ldx);
                __builtin_amdgcn_global_load_lds(at_bytes(xrow, yl), (lptr_t)(Xs + (BUF * GEMM_BK + r) * LD), 16, 0, 0);
            }
        }
        if (KR) {
            gptr_t erow = scalar_ptr(Eg + roff * P.lde);
            gptr_t grow = scalar_ptr(Xg + roff * P.ldx);
#pragma unroll
            for (int q = 0; q < ECQ; q++) {        // EC/8 wave-instructions, ECQ per wavefront
                unsigned eoffq = e_lane[q];
                asm volatile("" : "+v"(eoffq));
                __builtin_amdgcn_global_load_lds(at_bytes(erow, eoffq),
                                                 (lptr_t)(Es + BUF * GEMM_BK * EC + (wave + 4 * q) * 128), 16, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < KRQ; q++) {
                unsigned goffq = g_lane[q];
                asm volatile("" : "+v"(goffq));
                rg[q] = *at_bytes(grow, goffq);
            }
        }
    };
    auto stash_g = [&](int BUF) __attribute__((always_inline)) {
        if (KR) {
#pragma unroll
            for (int q = 0; q < KRQ; q++) Gs[BUF * GEMM_BK * GLD + tid + 256 * q] = rg[q];  // (columns >= nb hold column 0; never read)
        }
    };

    // ---- per-lane fragment base offsets (doubles; buffer 0, k-step 0) --------------------------
    const int sw = (lq & 1) << 4;  // rows ks*4 + lq: parity of the row = parity of lq
    // LDS byte addresses (segment and region bases folded in)
    const unsigned lds0 = (unsigned)(unsigned long)(lptr_t)smem;
    const unsigned XS_BASE = lds0 + 8u * 2 * GEMM_BK * LD, GS_BASE = XS_BASE + 8u * 2 * GEMM_BK * EC;
    unsigned yo[NT], xo[MT], go[MT], eo[MT];
#pragma unroll
    for (int t = 0; t < NT; t++) yo[t] = lds0 + 8u * (lq * LD + ((wn * WCOLS + l15 + t * 16) ^ sw));
#pragma unroll
    for (int t = 0; t < MT; t++) {
        const int mloc = wm * WROWS + t * 16 + l15;
        xo[t] = XS_BASE + 8u * (lq * LD + (mloc ^ sw));
        go[t] = 0;
        eo[t] = 0;
        if (KR) {
            const int m = m0 + mloc;
            const int b = m / k0;
            const int bl = b - b0;
            go[t] = GS_BASE + 8u * (lq * GLD + (bl < nb ? bl : nb - 1));
            eo[t] = XS_BASE + 8u * (lq * EC + ((m - b * k0) ^ sw));
        }
    }
    auto lds_at = [](unsigned addr, int imm_doubles) __attribute__((always_inline)) {
        return ((lcptr_t)(unsigned long)addr)[imm_doubles];
    };

    auto load_raw = [&](auto buf_tag, auto ks_tag, double (&a)[MT], double (&e)[MT], double (&b)[NT])
                        __attribute__((always_inline)) {
        constexpr int R0 = decltype(buf_tag)::value * GEMM_BK + decltype(ks_tag)::value * 4;
#pragma unroll
        for (int t = 0; t < NT; t++) b[t] = lds_at(yo[t], R0 * LD);
#pragma unroll
        for (int t = 0; t < MT; t++) {
            if (KR) {
                a[t] = lds_at(go[t], R0 * GLD);
                e[t] = lds_at(eo[t], R0 * EC);
            } else {
                a[t] = lds_at(xo[t], R0 * LD);
            }
        }
    };
    auto finish = [&](double (&a)[MT], const double (&e)[MT]) __attribute__((always_inline)) {
        if (KR) {
#pragma unroll
            for (int t = 0; t < MT; t++) a[t] *= e[t];
        }
    };

    v4d acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
    auto mma = [&](const double (&a)[MT], const double (&b)[NT]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MT; i++)
#pragma unroll
            for (int j = 0; j < NT; j++)
                acc[i][j] = TR ? __builtin_amdgcn_mfma_f64_16x16x4f64(b[j], a[i], acc[i][j], 0, 0, 0)
                               : __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    };
    // issue order of one k-step: the fragment reads of the next k-step behind the first MFMAs, two per
    // MFMA; the operand products behind the last ones (measured: within +-0.5 % of the compiler's own
    // order once the address arithmetic was gone; kept because it pins a known-good schedule)
    auto interleave = [&]() __attribute__((always_inline)) {
        constexpr int READS = KR ? 2 * MT + NT : MT + NT, PRODUCTS = KR ? MT : 0;
#pragma unroll
        for (int i = 0; i < READS / 2; i++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, MT * NT - READS / 2 - PRODUCTS, 0);
#pragma unroll
        for (int i = 0; i < PRODUCTS; i++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
        }
    };

    double fa[2][MT], fb[2][NT], fe[MT];
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    using K3 = std::integral_constant<int, 3>;
    static_assert(KS == 4, "the stage body is written out for four k-steps");
    issue(0, 0);
    stash_g(0);
    __syncthreads();  // drains the LDS-DMA (vmcnt) and publishes the tiles
    load_raw(K0{}, K0{}, fa[0], fe, fb[0]);
    finish(fa[0], fe);

    // One stage out of buffer BUF (a compile-time constant: the stage loop is unrolled by two, so every
    // LDS address is a loop-invariant per-lane base plus an immediate).  The fragments of its k-step 0
    // are in fa[0] / fb[0] on entry, those of the next stage's k-step 0 on exit.
    auto stage = [&](auto buf_tag, int s) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_tag)::value;
        using Other = std::integral_constant<int, BUF ^ 1>;
        const bool more = s + 1 < stages;
        // buffer BUF^1 was last read before the barrier of the previous stage: free for the DMA now
        if (more) issue(BUF ^ 1, s + 1);
        load_raw(buf_tag, K1{}, fa[1], fe, fb[1]);
        mma(fa[0], fb[0]);
        finish(fa[1], fe);
        interleave();
        load_raw(buf_tag, K2{}, fa[0], fe, fb[0]);
        mma(fa[1], fb[1]);
        finish(fa[0], fe);
        interleave();
        load_raw(buf_tag, K3{}, fa[1], fe, fb[1]);
        mma(fa[0], fb[0]);
        finish(fa[1], fe);
        interleave();
        // every read of this buffer has been issued; the next stage's tiles must have landed
        if (more) stash_g(BUF ^ 1);
        __syncthreads();
        // (after the last stage these reads fetch stale tiles of the other buffer; nothing uses them)
        load_raw(Other{}, K0{}, fa[0], fe, fb[0]);
        mma(fa[1], fb[1]);
        finish(fa[0], fe);
        interleave();
    };
    int s = 0;
    for (; s + 1 < stages; s += 2) {
        stage(K0{}, s);
        stage(K1{}, s + 1);
    }
    if (s < stages) stage(K0{}, s);

    double* Cb = P.C + (long)slice * split_stride;
    if (TR) {
#pragma unroll
        for (int j = 0; j < NT; j++) {
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int nn = n0 + wn * WCOLS + j * 16 + lq + 4 * reg;
                if (nn < P.N) {
#pragma unroll
                    for (int i = 0; i < MT; i++) {
                        const int m = m0 + wm * WROWS + i * 16 + l15;
                        if (m < P.M) Cb[(long)nn * P.ldc + m] = acc[i][j][reg];
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < MT; i++) {
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int m = m0 + wm * WROWS + i * 16 + lq + 4 * reg;
            if (m < P.M) {
#pragma unroll
                for (int j = 0; j < NT; j++) {
                    const int n = n0 + wn * WCOLS + j * 16 + l15;
                    if (n < P.N) {
                        double* cp = Cb + (long)m * P.ldc + n;
                        *cp = (P.flags & GEMM_SUBTRACT) ? *cp - acc[i][j][reg] : acc[i][j][reg];
                    }
                }
            }
        }
    }
}

template <bool KR, int KRQ, int ECQ, bool TR = false, int BN = 128, int TAG = 0>   // TAG: see GemmTune::tag
__global__ __launch_bounds__(256, (BN == 64 ? 3 : 2)) void gemm_tn_glds_kernel(const GemmProblem* __restrict__ probs,
                                                               int mtiles_max, long cells_per_split, long cells_total,
                                                               long split_stride, int k0) {
    extern __shared__ __align__(16) double smem[];
    const GemmProblem P = probs[blockIdx.z];
    // The grid is sized for the largest problem of the launch; every problem numbers ITS OWN tiles 0 .. mt_p * nt_p - 1 and
    // spreads them over the 8 XCDs by itself (workgroup b runs on XCD b & 7 whatever the problem).  Numbered against the
    // largest problem's tile count instead, the real tiles of a small problem sit at the start of each band and whole XCDs
    // get none of them: a multi-phenotype pass, whose (variant, rho*) groups differ in size by an order of magnitude, ran its
    // MixK products at 57 TFLOP/s against 73 for one phenotype.
    const int mt_p = (P.M + GEMM_BM - 1) / GEMM_BM, nt_p = (P.N + BN - 1) / BN;
    const int nwg_p = mt_p * nt_p;
    (void)mtiles_max;
    if ((int)blockIdx.x >= nwg_p) return;
    const int tile = xcd_tile_id((int)blockIdx.x, nwg_p);
    int tm = tile % mt_p, tn = tile / mt_p;
    if constexpr (!KR) {
        // plain products carry the band height of the tile walk in k0 (GemmTune::band): bands of `k0` column tiles of Y,
        // walked column tile first, so that the workgroups in flight on an XCD (a window of consecutive tile numbers)
        // cover a near-square block of the output and share both operands' panels in that XCD's L2
        if (k0 > 1 && nt_p >= 2 * k0) {
            const int per_band = k0 * mt_p, band = tile / per_band, left = tile - band * per_band;
            const int bh = min(k0, nt_p - band * k0);
            tn = band * k0 + left % bh;
            tm = left / bh;
        }
    }
    glds_tile<KR, KRQ, ECQ, TR, BN>(smem, P, tm, tn, (int)blockIdx.y, cells_per_split, cells_total, split_stride, k0);
}

// Persistent form with a soft per-XCD generation sync (the default for Khatri-Rao launches of more than 1024 tiles;
// crm_test_set_contraction_sync / CRM_CONTRACTION_SYNC=0 switch it off): 8 x 64 workgroups; the workgroups that
// share an XCD (blockIdx % 8) walk a contiguous run of the flattened (problem, slice, tile) list 64 tiles at a time
// and wait -- bounded, so no assumption about residency can deadlock -- until the whole group has finished a
// generation before starting the next: the 64 tiles of a generation then stream the same Q0 column tile and the
// same context rows at the same time.  The wait assumes the 512 workgroups are co-resident; when the GPU is shared
// (a second process, another stream's kernels) they may not be and every generation would sit out its bound
// (~5 ms).  A wait that runs out is counted in counters[8]; the host reads the count back with the next launch
// and returns to one workgroup per tile for the rest of the context's life (crm_test_sync_fallbacks).
template <bool KR, int KRQ, int ECQ, bool TR>
__global__ __launch_bounds__(256, 2) void gemm_tn_glds_sync_kernel(const GemmProblem* __restrict__ probs,
                                                                    int mtiles_max, int tiles_per_slice, int slices,
                                                                    int nprob, long cells_per_split, long cells_total,
                                                                    long split_stride, int k0,
                                                                    unsigned* __restrict__ counters, int every) {
    extern __shared__ __align__(16) double smem[];
    const int group = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const long total = (long)tiles_per_slice * slices * nprob;
    const long q = total >> 3, r = total & 7;
    const long start = group < r ? group * (q + 1) : r * (q + 1) + (group - r) * q;
    const long cnt = q + (group < r ? 1 : 0);
    for (long gen = 0; gen * slots < cnt; gen++) {
        const long idx = gen * slots + slot;
        if (idx < cnt) {
            const long w = start + idx;
            const int tile = (int)(w % tiles_per_slice);
            const int sl = (int)((w / tiles_per_slice) % slices);
            const int z = (int)(w / ((long)tiles_per_slice * slices));
            const GemmProblem P = probs[z];
            glds_tile<KR, KRQ, ECQ, TR, 128>(smem, P, tile % mtiles_max, tile / mtiles_max, sl, cells_per_split, cells_total,
                                        split_stride, k0);
        }
        if (threadIdx.x == 0 && (gen + 1) % every == 0) {
            __hip_atomic_fetch_add(&counters[group], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(slots * ((gen + 1) / every));
            bool met = false;
            for (int spin = 0; spin < 20000 && !met; spin++) {
                met = __hip_atomic_load(&counters[group], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want;
                if (!met) __builtin_amdgcn_s_sleep(8);
            }
            if (!met) __hip_atomic_fetch_add(&counters[8], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
}

int launch_gemm_tn_glds(crm_ctx* ctx, const GemmProblem* probs_dev, int nz, int mt, int nt, long cells,
                        bool khatri_rao, int k0, int ksplit, long split_stride, bool transposed_out, int bn) {
    if (bn != 128 && !(bn == 64 && khatri_rao) && !(bn == 160 && khatri_rao && !transposed_out)) {
        set_error("contraction: %d-wide LDS-DMA tiles are not built for this form", bn);
        return CRM_ERR_UNSUPPORTED;
    }
    if (transposed_out && !khatri_rao) {
        set_error("contraction: the transposed store is only built for the Khatri-Rao form");
        return CRM_ERR_UNSUPPORTED;
    }
    hipStream_t st = ctx->stream;
    dim3 grid((unsigned)(mt * nt), (unsigned)ksplit, (unsigned)nz);
    // waits of the previous persistent launch that ran out (read back asynchronously: a late value only delays the
    // fallback by a launch): the workgroups are not co-resident on this GPU right now -- one workgroup per tile from here
    if (ctx->tune.sync > 0 && ctx->sync_timeouts_host && *ctx->sync_timeouts_host >= 8) {
        ctx->tune.sync = 0;
        ctx->sync_fallbacks++;
        if (getenv("CRM_TRACE_SETUP"))
            fprintf(stderr, "[crm] persistent contraction: %u generation waits timed out (GPU shared?) -- falling back to one "
                            "workgroup per tile\n", *ctx->sync_timeouts_host);
    }
    const int sync_every = ctx->tune.sync;
    // (long contractions only: a generation of short tiles -- the per-donor launches of the kinship-structure route, 13 stages --
    // does not pay for its re-alignment wait: 22 ms in this form against 17 ms with one workgroup per tile; nor does the plain
    // Mix(rho*)' product of that route, 231 against 228 ms per block)
    const bool sync = sync_every > 0 && khatri_rao && bn == 128 && (long)mt * nt * ksplit * nz > 1024 && cells / ksplit >= 1024;
    unsigned* sync_counters = nullptr;
    if (sync) {
        CRM_TRY(ctx->sync_counters.ensure(64));
        sync_counters = ctx->sync_counters.as<unsigned>();
        CRM_HIP(hipMemsetAsync(sync_counters, 0, 64, st));
        if (!ctx->sync_timeouts_host) {
            CRM_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->sync_timeouts_host), sizeof(unsigned), hipHostMallocDefault));
            *ctx->sync_timeouts_host = 0;
        }
    }
    const long cps = (cells / GEMM_BK + ksplit - 1) / ksplit * GEMM_BK;  // validated by launch_gemm_tn
    size_t lds = (size_t)2 * GEMM_BK * bn * sizeof(double);
    if (khatri_rao) {
        const int EC = (k0 + 31) / 32 * 32;
        const int nb = glds_kr_variants(k0);
        constexpr int KRQ_BIG = (GEMM_BK * GEMM_BM + 255) / 256;
        const bool small = nb <= GldsGeno<1>::LD;
        lds += (size_t)2 * GEMM_BK * (EC + (small ? GldsGeno<1>::LD : GldsGeno<KRQ_BIG>::LD)) * sizeof(double);
#define CRM_GLDS_T(Q, T)                                                                                      \
    do {                                                                                                      \
        if (sync && small)                                                                                    \
            hipLaunchKernelGGL((gemm_tn_glds_sync_kernel<true, 1, Q, T>), dim3(512), dim3(256), lds, st,      \
                               probs_dev, mt, mt * nt, ksplit, nz, cps, cells, split_stride, k0,              \
                               sync_counters, sync_every);                                          \
        else if (small)                                                                                       \
            hipLaunchKernelGGL((gemm_tn_glds_kernel<true, 1, Q, T>), grid, dim3(256), lds, st, probs_dev, mt, \
                               cps, cells, split_stride, k0);                                             \
        else                                                                                                  \
            hipLaunchKernelGGL((gemm_tn_glds_kernel<true, KRQ_BIG, Q, T>), grid, dim3(256), lds, st,          \
                               probs_dev, mt, cps, cells, split_stride, k0);                              \
    } while (0)
#define CRM_GLDS_64(Q, T)                                                                                     \
    do {                                                                                                      \
        if (small)                                                                                            \
            hipLaunchKernelGGL((gemm_tn_glds_kernel<true, 1, Q, T, 64>), grid, dim3(256), lds, st,            \
                               probs_dev, mt, cps, cells, split_stride, k0);                              \
        else                                                                                                  \
            hipLaunchKernelGGL((gemm_tn_glds_kernel<true, KRQ_BIG, Q, T, 64>), grid, dim3(256), lds, st,      \
                               probs_dev, mt, cps, cells, split_stride, k0);                              \
    } while (0)
#define CRM_GLDS_160(Q)                                                                                       \
    do {                                                                                                      \
        if (small)                                                                                            \
            hipLaunchKernelGGL((gemm_tn_glds_kernel<true, 1, Q, false, 160>), grid, dim3(256), lds, st,       \
                               probs_dev, mt, cps, cells, split_stride, k0);                                  \
        else                                                                                                  \
            hipLaunchKernelGGL((gemm_tn_glds_kernel<true, KRQ_BIG, Q, false, 160>), grid, dim3(256), lds, st, \
                               probs_dev, mt, cps, cells, split_stride, k0);                                  \
    } while (0)
#define CRM_GLDS(Q)                              \
    do {                                         \
        if (bn == 64 && transposed_out) CRM_GLDS_64(Q, true); \
        else if (bn == 64) CRM_GLDS_64(Q, false); \
        else if (bn == 160) CRM_GLDS_160(Q);     \
        else if (transposed_out) CRM_GLDS_T(Q, true); \
        else CRM_GLDS_T(Q, false);               \
    } while (0)
        switch (EC / 32) {
            case 1: CRM_GLDS(1); break;
            case 2: CRM_GLDS(2); break;
            case 3: CRM_GLDS(3); break;
            default: CRM_GLDS(4); break;
        }
#undef CRM_GLDS
#undef CRM_GLDS_160
#undef CRM_GLDS_64
#undef CRM_GLDS_T
    } else {
        lds += (size_t)2 * GEMM_BK * 128 * sizeof(double);
        const int band = nt >= 2 * ctx->tune.band ? ctx->tune.band : 0;
        if (ctx->tune.tag)
            hipLaunchKernelGGL((gemm_tn_glds_kernel<false, 1, 0, false, 128, 1>), grid, dim3(256), lds, st, probs_dev, mt,
                               cps, cells, split_stride, band);
        else
            hipLaunchKernelGGL((gemm_tn_glds_kernel<false, 1, 0>), grid, dim3(256), lds, st, probs_dev, mt,
                               cps, cells, split_stride, band);
    }
    CRM_HIP(hipGetLastError());
    if (sync)
        CRM_HIP(hipMemcpyAsync(ctx->sync_timeouts_host, sync_counters + 8, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    return CRM_OK;
}

}  // namespace crm
