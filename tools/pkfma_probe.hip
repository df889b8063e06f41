// Probe (tools only): issue rate of v_pk_fma_f32 / v_pk_mul_f32 against v_fma_f32 in pure-VALU code, 1..8 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o tools/pkfma_probe tools/pkfma_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float m = 0.999f, c = 0.001f;
    const f2 pm = {m, m}, pc = {c, c};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (MODE == 1) {
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm), "v"(pc));
            } else if (MODE == 2) {
                asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm));
            } else if (MODE == 3) {
                asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                             "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (MODE == 5) {   // ONE dependent chain: 8 v_fma_f32, each on the previous result
                asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             : "+v"(a0) : "v"(m), "v"(c));
            } else if (MODE == 6) {   // TWO interleaved dependent chains
                asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                             "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                             : "+v"(a0), "+v"(a1) : "v"(m), "v"(c));
            } else if (MODE == 7) {   // ONE dependent chain of v_pk_fma_f32
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n"
                             : "+v"(p0) : "v"(pm), "v"(pc));
            } else if (MODE == 8) {   // dependent chain exp -> fma -> exp -> fma
                asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %0, %0, %1, %2\n v_exp_f32 %0, %0\n v_fma_f32 %0, %0, %1, %2\n"
                             "v_exp_f32 %0, %0\n v_fma_f32 %0, %0, %1, %2\n v_exp_f32 %0, %0\n v_fma_f32 %0, %0, %1, %2\n"
                             : "+v"(a0) : "v"(m), "v"(c));
            } else if (MODE == 9) {   // one MFMA 32x32x16 f16 + one dependent chain of 6 v_fma_f32 (a GELU slice per MFMA gap)
                typedef _Float16 h8 __attribute__((ext_vector_type(8)));
                typedef float f16v __attribute__((ext_vector_type(16)));
                static __device__ f16v acc;
                h8 x = {1, 1, 1, 1, 1, 1, 1, 1};
                f16v z = {};
                z[0] = a1;
                z = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, x, z, 0, 0, 0);
                asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             : "+v"(a0) : "v"(m), "v"(c));
                a1 = z[0] * 1e-30f;
            } else if (MODE == 4) {   // 4 exp + 4 pk_fma interleaved
                asm volatile("v_exp_f32 %0, %0\n v_pk_fma_f32 %4, %4, %8, %9\n v_exp_f32 %1, %1\n v_pk_fma_f32 %5, %5, %8, %9\n"
                             "v_exp_f32 %2, %2\n v_pk_fma_f32 %6, %6, %8, %9\n v_exp_f32 %3, %3\n v_pk_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm), "v"(pc));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int MODE>
void run(const char* name, int instr_per_iter, int lanes_elems) {
    float* out; hipMalloc(&out, 256 * 1024 * 4 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps = 1; wps <= 8; wps = wps < 4 ? wps + 1 : wps * 2) {
        const int threads = 256, blocks = 256 * wps;     // 4 waves per block = 1 per SIMD
        const int iters = 4096;
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 16);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double inst = (double)iters * 16 * instr_per_iter * wps;     // per SIMD
        printf("%-28s waves/SIMD %d: %.3f ms, %.2f ns per instruction per SIMD (at 2.4 GHz: %.2f cycles)\n", name, wps, ms, ms * 1e6 / inst, ms * 1e6 / inst * 2.4);
    }
}
int main() {
    run<0>("v_fma_f32", 8, 1);
    run<1>("v_pk_fma_f32", 4, 2);
    run<2>("v_pk_mul_f32", 4, 2);
    run<3>("v_exp_f32", 8, 1);
    run<4>("v_exp_f32+v_pk_fma_f32 pairs", 8, 1);
    run<5>("v_fma_f32, 1 dependent chain", 8, 1);
    run<6>("v_fma_f32, 2 chains", 8, 1);
    run<7>("v_pk_fma_f32, 1 dependent chain", 4, 2);
    run<8>("exp->fma dependent chain", 8, 1);
    run<9>("mfma + 6 dependent v_fma (per 7)", 7, 1);
    return 0;
}
