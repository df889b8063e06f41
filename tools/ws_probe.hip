// Probe (tools only): does a wave that issues ONLY v_mfma (one per SIMD) share the SIMD with a wave that issues ONLY vector instructions?
// 8-wave workgroups, one per CU: waves 0-3 run a chain of MFMAs, waves 4-7 a stream of v_fma_f32; kernel time for each alone and for both.
// hipcc --offload-arch=gfx950 -O3 -o tools/ws_probe tools/ws_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int NOP>
__global__ __launch_bounds__(512) void k(float* out, int n_mfma, int n_valu, int roles) {
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        if (!(roles & 1)) return;
        h8 x = {1, 1, 1, 1, 1, 1, 1, 1};
        f16v z = {};
        for (int i = 0; i < n_mfma; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                z = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, x, z, 0, 0, 0);
                if (NOP == 16) asm volatile("s_nop 15");
                if (NOP == 24) asm volatile("s_nop 15\n s_nop 7");
                if (NOP == 28) asm volatile("s_nop 15\n s_nop 11");
                if (NOP == 1) asm volatile("s_sleep 1");
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = z[0] + z[5];
    } else {
        if (!(roles & 2)) return;
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
        const float m = 0.999f, c = 0.001f;
        for (int i = 0; i < n_valu; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));
        }
        out[blockIdx.x * 512 + threadIdx.x] = a0 + a1 + a2 + a3;
    }
}
template <int NOP>
float run(float* out, int nm, int nv, int roles) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NOP>, dim3(256), dim3(512), 0, 0, out, 8, 8, roles);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<NOP>, dim3(256), dim3(512), 0, 0, out, nm, nv, roles);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
template <int NOP>
void test(float* out) {
    const int nm = 4096, nv = 4096 * 3;      // 32768 MFMAs (x 32 cycles = 1.05 M cycles) beside 393216 v_fma (x ~5 cycles single wave = 2 M cycles ... )
    const float a = run<NOP>(out, nm, nv, 1), b = run<NOP>(out, nm, nv, 2), c = run<NOP>(out, nm, nv, 3);
    printf("gap after each MFMA: %2d scalar idle cycles%s | MFMA waves alone %.3f ms (%.1f cycles per MFMA at 2.4 GHz), vector waves alone %.3f ms (%.2f cycles per v_fma), both %.3f ms (sum %.3f, max %.3f)\n",
           NOP == 1 ? 64 : NOP, NOP == 1 ? " (s_sleep 1)" : "", a, a * 2.4e6 / (nm * 8), b, b * 2.4e6 / (nv * 32.0), c, a + b, a > b ? a : b);
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    test<0>(out); test<16>(out); test<24>(out); test<28>(out); test<1>(out);
    return 0;
}
