// Diagnostic: sustained v_mfma_f64_16x16x4_f64 rate and in-kernel clock (s_memtime / s_memrealtime)
// for waves-per-SIMD in {1, 2} and operands in {random, zero}.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(double* out, unsigned long long* stamps, int iters, double scale) {
    v4d acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = (v4d){0, 0, 0, 0};
    double a = (threadIdx.x * 1e-3 + 0.5) * scale, b = (1.0 + threadIdx.x * 1e-4) * scale;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
    double* out; unsigned long long* st;
    CK(hipMalloc(&out, sizeof(double) * 1024 * 256));
    CK(hipMalloc(&st, sizeof(unsigned long long) * 2048));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wps : {1, 2}) for (double scale : {1.0, 0.0}) {
        int blocks = 256 * wps, iters = 40000;
        hipLaunchKernelGGL(mfma_loop<8>, dim3(blocks), dim3(256), 0, 0, out, st, 2000, scale);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(mfma_loop<8>, dim3(blocks), dim3(256), 0, 0, out, st, iters, scale);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[2048]; CK(hipMemcpy(h, st, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost));
        double cyc = 0, rt = 0; for (int i = 0; i < blocks; i++) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
        double clk_ghz = cyc / rt * 0.1;  // memrealtime ticks at 100 MHz
        double flops = (double)blocks * 4 * iters * 8 * 2048.0;
        double cyc_per_mfma = (cyc / blocks) / ((double)iters * 8 * wps);  // per SIMD
        printf("waves/SIMD=%d operands=%s: %.2f ms %.2f TFLOP/s  in-kernel clock %.3f GHz  cycles per MFMA per SIMD %.1f\n",
               wps, scale == 0.0 ? "zero" : "nonzero", ms, flops / ms * 1e-9, clk_ghz, cyc_per_mfma);
    }
    return 0;
}
