// Probe (tools only, round 6): does vector-ALU work issued by the SAME wave run under that wave's matrix instructions?
// tools/coexec_probe.hip (round 1) answered "no" for a compiler-scheduled loop; this one fixes the instruction order in inline asm (nothing the
// compiler can move, no hazard nops inside the block) and separates the cases the fused block tail could use at ONE wave per SIMD:
//   CH = 1  one dependent chain of v_mfma_f32_32x32x16_f16 (every instruction accumulates into the previous one's result)
//   CH = 2  two independent accumulators alternating,  CH = 3: three (the fc2 phase of the fused tail has NT = 3 of them at C = 96)
//   K       independent v_fma_f32 (K distinct registers, so no vector dependence either) after every matrix instruction
//   ACC     accumulators in AccVGPRs (a[...]) instead of v[...]
// One workgroup of 4 waves per CU (100 KB of LDS requested so that two never share a CU) = 1 wave per SIMD;  WPS = 2: 8 waves per workgroup.
// Prints cycles per matrix-instruction slot (s_memtime of wave 0 of workgroup 0) and the wall time.
// hipcc --offload-arch=gfx950 -O3 -o tools/coissue_probe tools/coissue_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

#define FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define V1 FMA(0)
#define V2 V1 FMA(1)
#define V3 V2 FMA(2)
#define V4 V3 FMA(3)
#define V6 V4 FMA(4) FMA(5)
#define V8 V6 FMA(6) FMA(7)
#define V12 V8 FMA(0) FMA(1) FMA(2) FMA(3)
#define V16 V8 V8
#define MV(acc) "v_mfma_f32_32x32x16_f16 v[" acc "], %10, %11, v[" acc "]\n"
#define MA(acc) "v_mfma_f32_32x32x16_f16 a[" acc "], %10, %11, a[" acc "]\n"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// the loop body: 12 matrix-instruction slots (a multiple of 1, 2, 3 chains); VS = the vector instructions after each
#define BODY1(M, VS) M("0:15") VS M("0:15") VS M("0:15") VS M("0:15") VS M("0:15") VS M("0:15") VS M("0:15") VS M("0:15") VS M("0:15") VS M("0:15") VS M("0:15") VS M("0:15") VS
#define BODY2(M, VS) M("0:15") VS M("16:31") VS M("0:15") VS M("16:31") VS M("0:15") VS M("16:31") VS M("0:15") VS M("16:31") VS M("0:15") VS M("16:31") VS M("0:15") VS M("16:31") VS
#define BODY3(M, VS) M("0:15") VS M("16:31") VS M("32:47") VS M("0:15") VS M("16:31") VS M("32:47") VS M("0:15") VS M("16:31") VS M("32:47") VS M("0:15") VS M("16:31") VS M("32:47") VS
#define NOMFMA(acc) ""
#define BODY0(M, VS) VS VS VS VS VS VS VS VS VS VS VS VS

#define CLOB_V "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47"
#define CLOB_A "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47"

#define KERNEL(NAME, BODY, M, VS, CLOB)                                                                                                        \
    __global__ __launch_bounds__(512) void NAME(float* out, unsigned long long* cyc, int iters) {                                              \
        extern __shared__ unsigned char lds[];                                                                                                 \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                     \
        const float m = 0.999f, c = 0.001f;                                                                                                    \
        h8 x, y;                                                                                                                               \
        for (int j = 0; j < 8; ++j) { x[j] = (_Float16)(1.f + threadIdx.x * 1e-3f); y[j] = (_Float16)(1e-3f * j); }                            \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                                            \
        for (int i = 0; i < iters; ++i)                                                                                                        \
            asm volatile(BODY(M, VS) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c), "v"(x), "v"(y) : CLOB); \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                    \
        if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;                                                                             \
    }

// the accumulator registers v0..v47 are clobbers, so the fma operands live above them
KERNEL(k_m1_v0, BODY1, MV, "", CLOB_V)
KERNEL(k_m1_v4, BODY1, MV, V4, CLOB_V)
KERNEL(k_m1_v8, BODY1, MV, V8, CLOB_V)
KERNEL(k_m1_v12, BODY1, MV, V12, CLOB_V)
KERNEL(k_m2_v0, BODY2, MV, "", CLOB_V)
KERNEL(k_m2_v4, BODY2, MV, V4, CLOB_V)
KERNEL(k_m2_v8, BODY2, MV, V8, CLOB_V)
KERNEL(k_m2_v12, BODY2, MV, V12, CLOB_V)
KERNEL(k_m3_v0, BODY3, MV, "", CLOB_V)
KERNEL(k_m3_v8, BODY3, MV, V8, CLOB_V)
KERNEL(k_m3_v12, BODY3, MV, V12, CLOB_V)
KERNEL(k_a1_v0, BODY1, MA, "", CLOB_A)
KERNEL(k_a1_v8, BODY1, MA, V8, CLOB_A)
KERNEL(k_a2_v8, BODY2, MA, V8, CLOB_A)
KERNEL(k_a3_v8, BODY3, MA, V8, CLOB_A)
KERNEL(k_a3_v12, BODY3, MA, V12, CLOB_A)
KERNEL(k_m0_v4, BODY0, NOMFMA, V4, CLOB_V)
KERNEL(k_m0_v8, BODY0, NOMFMA, V8, CLOB_V)
KERNEL(k_m0_v12, BODY0, NOMFMA, V12, CLOB_V)

typedef void (*kern_t)(float*, unsigned long long*, int);
static void run(const char* name, kern_t k, int wps) {
    static float* out = nullptr; static unsigned long long* cyc = nullptr;
    if (!out) { (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 64); }
    const int iters = 4000, threads = 256 * wps;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 100 * 1024, 0, out, cyc, 20);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 100 * 1024, 0, out, cyc, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-34s %d wave/SIMD: %.3f ms, %6.1f s_memtime ticks per slot, %6.1f ns per slot\n", name, wps, ms, (double)h / (iters * 12.0), ms * 1e6 / (iters * 12.0));
}
#define RUN(k, desc) run(desc, k, wps)
int main() {
    for (int wps = 1; wps <= 2; ++wps) {
        RUN(k_m1_v0, "1 chain (v), 0 fma");
        RUN(k_m1_v4, "1 chain (v), 4 fma");
        RUN(k_m1_v8, "1 chain (v), 8 fma");
        RUN(k_m1_v12, "1 chain (v), 12 fma");
        RUN(k_m2_v0, "2 chains (v), 0 fma");
        RUN(k_m2_v4, "2 chains (v), 4 fma");
        RUN(k_m2_v8, "2 chains (v), 8 fma");
        RUN(k_m2_v12, "2 chains (v), 12 fma");
        RUN(k_m3_v0, "3 chains (v), 0 fma");
        RUN(k_m3_v8, "3 chains (v), 8 fma");
        RUN(k_m3_v12, "3 chains (v), 12 fma");
        RUN(k_a1_v0, "1 chain (acc regs), 0 fma");
        RUN(k_a1_v8, "1 chain (acc regs), 8 fma");
        RUN(k_a2_v8, "2 chains (acc regs), 8 fma");
        RUN(k_a3_v8, "3 chains (acc regs), 8 fma");
        RUN(k_a3_v12, "3 chains (acc regs), 12 fma");
        RUN(k_m0_v4, "no mfma, 4 fma");
        RUN(k_m0_v8, "no mfma, 8 fma");
        RUN(k_m0_v12, "no mfma, 12 fma");
    }
    return 0;
}
