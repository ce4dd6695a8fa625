// Stand-alone probe: (1) bare v_mfma_f64_16x16x4_f64 issue rate, (2) throughput of the
// contraction kernel (plain and Khatri-Rao) at config-3 shapes.  Not part of the product.
//   sh tools/build_probes.sh   (links against the built libcrm_hip.so)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "../cellregmap_amd/csrc/crm_internal.h"

typedef double v4d __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mfma_rate(double* out, int iters) {
    v4d acc[8];
    for (int i = 0; i < 8; i++) acc[i] = (v4d){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void fill(double* p, long n, unsigned seed) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned)(i * 2654435761u) ^ seed;
    x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
    p[i] = ((double)(x & 0xffffff) / 8388608.0) - 1.0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    crm_ctx* ctx = nullptr;
    if (crm_ctx_create(0, &ctx) != CRM_OK) { printf("no context: %s\n", crm_last_error()); return 1; }
    hipStream_t st = ctx->stream;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms;
    const long cells = 20000, r = 5120;
    const int k0 = 50;
    int Bs[] = {128, 1024, 4096};   // 4096 = one automatic block of the scan at config 3
    const long GLD = 4096 + 128;
    double *Q0, *G, *E, *C;
    CK(hipMalloc(&Q0, sizeof(double) * cells * r));
    CK(hipMalloc(&G, sizeof(double) * cells * GLD));
    CK(hipMalloc(&E, sizeof(double) * cells * 64));
    CK(hipMalloc(&C, sizeof(double) * 4096L * k0 * r));
    hipLaunchKernelGGL(fill, dim3((cells * r + 255) / 256), dim3(256), 0, st, Q0, cells * r, 1u);
    hipLaunchKernelGGL(fill, dim3((unsigned)((cells * GLD + 255) / 256)), dim3(256), 0, st, G, cells * GLD, 2u);
    hipLaunchKernelGGL(fill, dim3((cells * 64 + 255) / 256), dim3(256), 0, st, E, cells * 64, 3u);
    crm::GemmProblem* pd;
    CK(hipMalloc(&pd, sizeof(crm::GemmProblem)));
    if (argc > 1) ctx->tune.bn = atoi(argv[1]);
    if (argc > 2) ctx->tune.glds = atoi(argv[2]);
    if (argc > 3) ctx->tune.sync = atoi(argv[3]);
    printf("tile width %d\n", ctx->tune.bn);
    for (int B : Bs) {
        // plain: T = G' Q0  (M = B, N = r)
        crm::GemmProblem p{};
        p.X = G; p.Y = Q0; p.C = C; p.ldx = GLD; p.ldy = r; p.ldc = r; p.M = B; p.N = (int)r;
        CK(hipMemcpy(pd, &p, sizeof p, hipMemcpyHostToDevice));
        crm::launch_gemm_tn(ctx, pd, 1, B, (int)r, cells, false, 0, 1, 0);
        CK(hipEventRecord(e0, st));
        crm::launch_gemm_tn(ctx, pd, 1, B, (int)r, cells, false, 0, 1, 0);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("plain  B=%4d: %.3f ms  %.2f TFLOP/s\n", B, ms, 2.0 * cells * r * B / ms * 1e-9);
        // KR: A = KR(G, E)' Q0   (M = B*k0, N = r)
        p.E = E; p.lde = 64; p.M = B * k0; p.k0 = k0;
        CK(hipMemcpy(pd, &p, sizeof p, hipMemcpyHostToDevice));
        crm::launch_gemm_tn(ctx, pd, 1, B * k0, (int)r, cells, true, k0, 1, 0);
        CK(hipEventRecord(e0, st));
        crm::launch_gemm_tn(ctx, pd, 1, B * k0, (int)r, cells, true, k0, 1, 0);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("KR     B=%4d: %.3f ms  %.2f TFLOP/s  (%.1f variants/s on this term)\n", B, ms,
               2.0 * cells * r * B * k0 / ms * 1e-9, B / (ms * 1e-3));
    }
    printf("last error: '%s'\n", crm::last_error_text());
    return 0;
}
