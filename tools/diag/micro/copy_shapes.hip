// Micro-benchmark behind blockops.hip's launch shapes: a rows x cols block of doubles copied (a) one row per blockIdx.x, one
// element per thread (the round-5 shape of gather_block / square_block / ortho_apply / gather_rows), (b) the same with
// the column chunk as the fast block index, (c) two and (d) four elements per thread, column chunk fast.
//   hipcc -O3 --offload-arch=gfx950 copy_shapes.hip -o copy_shapes && ./copy_shapes [rows 5024] [cols 4096]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void copy_row_fast(const double* __restrict__ s, double* __restrict__ d, long ld, int cols) {
    const int j = blockIdx.y * blockDim.x + threadIdx.x;
    const long i = blockIdx.x;
    if (j < cols) d[i * ld + j] = s[i * ld + j];
}
__global__ void copy_col_fast(const double* __restrict__ s, double* __restrict__ d, long ld, int cols, int chunks) {
    const long i = blockIdx.x / chunks;
    const int j = (blockIdx.x % chunks) * blockDim.x + threadIdx.x;
    if (j < cols) d[i * ld + j] = s[i * ld + j];
}
__global__ void copy_col_fast2(const double* __restrict__ s, double* __restrict__ d, long ld, int cols, int chunks) {
    const long i = blockIdx.x / chunks;
    const int j = 2 * ((blockIdx.x % chunks) * blockDim.x + threadIdx.x);
    if (j < cols) *reinterpret_cast<double2*>(d + i * ld + j) = *reinterpret_cast<const double2*>(s + i * ld + j);
}
__global__ void copy_col_fast4(const double* __restrict__ s, double* __restrict__ d, long ld, int cols, int chunks) {
    const long i = blockIdx.x / chunks;
    const int j = 2 * ((blockIdx.x % chunks) * 2 * blockDim.x + threadIdx.x);
    if (j < cols) *reinterpret_cast<double2*>(d + i * ld + j) = *reinterpret_cast<const double2*>(s + i * ld + j);
    const int j2 = j + 2 * blockDim.x;
    if (j2 < cols) *reinterpret_cast<double2*>(d + i * ld + j2) = *reinterpret_cast<const double2*>(s + i * ld + j2);
}
__global__ void copy_flat2(const double* __restrict__ s, double* __restrict__ d, long n2) {   // grid-stride, 16 B
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n2; e += (long)gridDim.x * blockDim.x)
        reinterpret_cast<double2*>(d)[e] = reinterpret_cast<const double2*>(s)[e];
}

int main(int argc, char** argv) {
    const long rows = argc > 1 ? atol(argv[1]) : 5024;
    const int cols = argc > 2 ? atoi(argv[2]) : 4096;
    const long ld = cols;
    double *s, *d;
    hipMalloc(&s, rows * ld * 8);
    hipMalloc(&d, rows * ld * 8);
    hipMemset(s, 0, rows * ld * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const double gb = 2.0 * rows * cols * 8 / 1e9;
    for (int form = 0; form < 5; form++) {
        float best = 1e9;
        for (int rep = 0; rep < 6; rep++) {
            hipEventRecord(e0);
            if (form == 0) copy_row_fast<<<dim3(rows, (cols + 255) / 256), 256>>>(s, d, ld, cols);
            if (form == 1) { const int ch = (cols + 255) / 256; copy_col_fast<<<rows * ch, 256>>>(s, d, ld, cols, ch); }
            if (form == 2) { const int ch = (cols + 511) / 512; copy_col_fast2<<<rows * ch, 256>>>(s, d, ld, cols, ch); }
            if (form == 3) { const int ch = (cols + 1023) / 1024; copy_col_fast4<<<rows * ch, 256>>>(s, d, ld, cols, ch); }
            if (form == 4) copy_flat2<<<256 * 16, 256>>>(s, d, rows * ld / 2);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        const char* names[] = {"row per blockIdx.x, 8 B", "column chunk fast, 8 B", "column chunk fast, 16 B", "column chunk fast, 2 x 16 B", "grid-stride flat, 16 B"};
        printf("%-30s %.3f ms  %.2f TB/s\n", names[form], best, gb / best);
    }
    return 0;
}
