// Register-only bf16 MFMA throughput probe: what the chip sustains on v_mfma_f32_32x32x16_bf16 (the instruction the
// split-bf16 GEMM issues) for NACC independent accumulators used round-robin.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(a0 + threadIdx.x * 1e-3f + j); b[j] = (__bf16)(b0 - threadIdx.x * 1e-3f - j); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u % NACC], 0, 0, 0);
        a[0] = (__bf16)((float)a[0] + 1e-3f);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(float* d, int blocks) {
    int iters = 4000;
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, d, 100, 1.f, 2.f);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.f, 2.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)blocks * 4 * iters * 32.0 * (32.0 * 32 * 16 * 2);
    printf("bf16 32x32x16 nacc %d blocks %4d: %.3f ms  %.1f TFLOP/s\n", NACC, blocks, ms, flop / ms / 1e9);
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 4096 * 4);
    for (int blocks : {256, 512}) { run<1>(d, blocks); run<2>(d, blocks); run<4>(d, blocks); }
    return 0;
}
