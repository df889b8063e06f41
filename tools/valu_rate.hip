// VALU issue-rate probe: scalar v_fma_f32 vs packed v_pk_fma_f32 vs the transcendental unit (v_exp_f32), 8 independent chains per lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
    float a[8]; v2f p[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3f + i; p[i] = (v2f){a[i], a[i] + 0.5f}; }
    const float m = 0.999f + s * 1e-9f; const v2f mm = {m, m}, cc = {1e-3f, 2e-3f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) a[u] = __builtin_fmaf(a[u], m, 1e-3f);
            if (MODE == 1) p[u] = __builtin_elementwise_fma(p[u], mm, cc);
            if (MODE == 2) a[u] = __builtin_amdgcn_exp2f(a[u] * 1e-3f);
        }
    }
    float r = 0.f;
    for (int i = 0; i < 8; ++i) r += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int MODE> void run(float* d, const char* name, double flop_per_op) {
    const int blocks = 256 * 8, iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.f);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)blocks * 4 * iters * 8;        // wave-instructions
    printf("%-14s %.3f ms  %.1f G wave-instr/s  (%.1f T lane-ops/s, %.1f TFLOP/s)\n", name, ms, insts / ms / 1e6, insts * 64 / ms / 1e9, insts * 64 * flop_per_op / ms / 1e9);
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>(d, "v_fma_f32", 2); run<1>(d, "v_pk_fma_f32", 4); run<2>(d, "v_exp_f32+mul", 1);
    return 0;
}
