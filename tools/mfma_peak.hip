// Register-only fp32 MFMA throughput probe (what the chip sustains on v_mfma_f32_32x32x2_f32 with nothing else going on).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        a += 1e-6f;
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, 256 * 4096 * 4);
    for (int blocks : {256, 512, 1024, 2048}) {
        int iters = 4000;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 100, 1.f, 2.f);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 1.f, 2.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flop = (double)blocks * 4 /*waves*/ * iters * 32.0 * (32.0 * 32 * 2 * 2);
        printf("blocks %4d: %.3f ms  %.1f TFLOP/s\n", blocks, ms, flop / ms / 1e9);
    }
    return 0;
}
