// Diagnostic: how fast can the lower triangle of a batch of large row-major matrices be STREAMED in the access shape of
// the tridiagonalisation's symv (eigh_trd.hip: tiles of R rows x C columns, a workgroup takes a run of tiles in one block
// row and reads every tile once, 16 bytes per lane) -- for row pieces of 512 B (64 x 64 tiles, what trd_symv_kernel uses),
// 1 KiB, 2 KiB and 4 KiB -- against one contiguous sweep of the same bytes?  Loads only (a sum keeps them alive); the
// answer bounds what a re-tiled symv could gain.  Not part of the product.   hipcc --offload-arch=gfx950 -O3 ... && ./symv_stream_probe [dim 10050] [batch 10]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double v2d __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Seg { int row0, col0, ntiles; };   // a run of tiles (R x C each) in one block row, starting at (row0, col0)

// 256 threads; a tile is R x C doubles = 32 KiB: thread t loads 8 pieces of 16 B per tile.  Lane layout: C/2 lanes across a
// row piece, 256 / (C/2) rows per pass, R / rows_per_pass passes.
template <int R, int C>
__global__ __launch_bounds__(256) void stream_tiles(const double* __restrict__ A, long ld, long slab, const Seg* __restrict__ segs,
                                                    double* __restrict__ out) {
    constexpr int LPR = C / 2;            // lanes per row piece
    constexpr int RPP = 256 / LPR;        // rows per pass
    constexpr int PASSES = R / RPP;       // = 8 for every shape (R * C = 4096)
    const Seg s = segs[blockIdx.x];
    const double* base = A + (long)blockIdx.y * slab + (long)(s.row0 + threadIdx.x / LPR) * ld + s.col0 + 2 * (threadIdx.x % LPR);
    v2d cur[PASSES], nxt[PASSES];
    double acc = 0.0;
    auto load = [&](int t, v2d (&v)[PASSES]) {
#pragma unroll
        for (int q = 0; q < PASSES; q++) v[q] = *reinterpret_cast<const v2d*>(base + (long)t * C + (long)(q * RPP) * ld);
    };
    load(0, cur);
    for (int t = 0; t < s.ntiles; t++) {
        if (t + 1 < s.ntiles) load(t + 1, nxt);
#pragma unroll
        for (int q = 0; q < PASSES; q++) acc += cur[q][0] + cur[q][1];
        if (t + 1 < s.ntiles) {
#pragma unroll
            for (int q = 0; q < PASSES; q++) cur[q] = nxt[q];
        }
    }
    out[(long)blockIdx.y * gridDim.x * 256 + (long)blockIdx.x * 256 + threadIdx.x] = acc;
}

// the same bytes as one contiguous sweep: every workgroup reads 256 KiB in a row
__global__ __launch_bounds__(256) void stream_flat(const double* __restrict__ A, long chunks, double* __restrict__ out) {
    const long c = blockIdx.x;
    if (c >= chunks) return;
    const v2d* p = reinterpret_cast<const v2d*>(A) + c * 16384 + threadIdx.x;   // 256 KiB = 16384 x 16 B
    double acc = 0.0;
#pragma unroll 8
    for (int i = 0; i < 64; i++) { const v2d v = p[i * 256]; acc += v[0] + v[1]; }
    out[c * 256 + threadIdx.x] = acc;
}

template <int R, int C>
static int run(const double* A, long dim, long ld, long slab, int batch, double* out, hipEvent_t e0, hipEvent_t e1) {
    // lower triangle in R x C tiles: block row I covers rows [I R, (I+1) R), tiles left of and on the diagonal
    std::vector<Seg> segs;
    long tiles = 0;
    const int per = 8;   // tiles per workgroup (256 KiB)
    for (long r0 = 0; r0 + R <= dim; r0 += R) {
        const int nt = (int)((r0 + R + C - 1) / C);   // tiles up to the diagonal of this block row's last row
        for (int t0 = 0; t0 < nt; t0 += per) segs.push_back({(int)r0, t0 * C, std::min(per, nt - t0)});
        tiles += nt;
    }
    Seg* d_segs;
    CK(hipMalloc(&d_segs, sizeof(Seg) * segs.size()));
    CK(hipMemcpy(d_segs, segs.data(), sizeof(Seg) * segs.size(), hipMemcpyHostToDevice));
    const double bytes = (double)tiles * R * C * 8.0 * batch;
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((stream_tiles<R, C>), dim3((unsigned)segs.size(), batch), dim3(256), 0, 0, A, ld, slab, d_segs, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    printf("tiles %3d rows x %3d columns (row pieces of %4d B): %8.1f GB in %7.3f ms = %5.2f TB/s   (%zu workgroups x %d matrices)\n", R, C, C * 8,
           bytes * 1e-9, best, bytes / best * 1e-9, segs.size(), batch);
    CK(hipFree(d_segs));
    return 0;
}

int main(int argc, char** argv) {
    const long dim = argc > 1 ? atol(argv[1]) : 10050;
    const int batch = argc > 2 ? atoi(argv[2]) : 10;
    const long ld = (dim + 511) / 512 * 512, slab = ld * ld;
    double *A, *out;
    CK(hipMalloc(&A, sizeof(double) * slab * batch));
    CK(hipMemset(A, 0, sizeof(double) * slab * batch));
    CK(hipMalloc(&out, sizeof(double) * 256 * 400000));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("dim %ld, ld %ld, %d matrices of %.0f MB\n", dim, ld, batch, slab * 8e-6);
    if (run<64, 64>(A, dim, ld, slab, batch, out, e0, e1)) return 1;
    if (run<32, 128>(A, dim, ld, slab, batch, out, e0, e1)) return 1;
    if (run<16, 256>(A, dim, ld, slab, batch, out, e0, e1)) return 1;
    if (run<8, 512>(A, dim, ld, slab, batch, out, e0, e1)) return 1;
    const long chunks = slab * batch / 2 / 32768;   // half of every matrix, contiguous
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(stream_flat, dim3((unsigned)chunks), dim3(256), 0, 0, A, chunks, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    printf("contiguous sweep of %.1f GB: %7.3f ms = %5.2f TB/s\n", chunks * 262144e-9, best, chunks * 262144.0 / best * 1e-9);
    return 0;
}
