// Can VALU work hide under MFMAs?  One dependent chain of v_mfma_f32_32x32x16_bf16 with K independent VALU instructions
// (v_fma_f32 / v_pk_fma_f32 / v_exp_f32 / v_cvt_pk_bf16_f32 mixes) issued after every MFMA, 1 or 2 waves per SIMD.
// Prints time per MFMA slot in core cycles (s_memtime), so the answer does not depend on the clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// KIND 0: v_fma_f32, 1: v_exp_f32 (quarter rate), 2: v_cvt_pk_bf16_f32 + v_and + v_sub (split-like), 3: no MFMA at all (VALU only, fma)
template <int K, int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters, float a0, float b0) {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(a0 + threadIdx.x * 1e-3f + j); b[j] = (__bf16)(b0 - threadIdx.x * 1e-3f - j); }
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = a0 * (j + 1) + threadIdx.x * 1e-4f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (KIND != 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < K; ++q) {
                float& x = v[q & 7];
                if (KIND == 0 || KIND == 3) x = __builtin_fmaf(x, 1.0001f, 0.5f);
                else if (KIND == 1) x = __builtin_amdgcn_exp2f(x) * 0.f + x;
                else { unsigned p; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(p) : "v"(x)); x = x - __uint_as_float(p & 0xffff0000u) + 1.f; }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int K, int KIND> void run(float* d, unsigned long long* c, int blocks) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<K, KIND>), dim3(blocks), dim3(256), 0, 0, d, c, 50, 1.f, 2.f);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<K, KIND>), dim3(blocks), dim3(256), 0, 0, d, c, iters, 1.f, 2.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8]; (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    const char* kn[] = {"fma", "exp2+fma", "cvt_pk+and+sub+add", "fma, NO mfma"};
    printf("K %2d %-20s blocks %4d (%d waves/SIMD): %.3f ms  %.1f cycles per slot  (%.2f GHz)\n", K, kn[KIND], blocks, blocks / 256, ms,
           (double)h[0] / (iters * 16.0), (double)h[0] / (ms * 1e6));
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 4096 * 4);
    unsigned long long* c; (void)hipMalloc(&c, 4096 * 8);
    for (int blocks : {256, 512}) {
        run<0, 0>(d, c, blocks); run<2, 0>(d, c, blocks); run<4, 0>(d, c, blocks); run<8, 0>(d, c, blocks); run<16, 0>(d, c, blocks); run<32, 0>(d, c, blocks);
        run<8, 3>(d, c, blocks); run<16, 3>(d, c, blocks); run<32, 3>(d, c, blocks);
        run<1, 1>(d, c, blocks); run<2, 1>(d, c, blocks); run<4, 1>(d, c, blocks);
        run<2, 2>(d, c, blocks); run<4, 2>(d, c, blocks); run<8, 2>(d, c, blocks);
    }
    return 0;
}
