// Stand-alone micro-benchmark and checker of the ring dense engine (xpoint_amd/csrc/ring_core.h): no torch, no Python — one binary per XP_RING_DBG value.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I xpoint_amd/csrc -I include tools/ring_bench.hip -o tools/ring_bench
//   tools/ring_bench [planes=1|2|0(both)] [rounds] [check=1|0]
// Prints, per shape and tile variant, the median / minimum launch time of interleaved rounds (cdna_hip_programming.md §5.4 rule 24), TF/s on the EXECUTED
// fp16 products, the L2 -> LDS bytes per second the DMA moved, and the maximum deviation from an fp64 reference computed from the same operand images.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "ring_core.h"

void xp_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }

#define CK(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { fprintf(stderr, "%s failed: %s (%s:%d)\n", #call, hipGetErrorString(e__), __FILE__, __LINE__); exit(2); } } while (0)

__device__ __forceinline__ float hash_unit(unsigned long long i, unsigned seed) {      // uniform in [-1, 1)
    unsigned long long z = i * 0x9E3779B97F4A7C15ull + seed * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    return (float)((z >> 40) & 0xFFFFFF) * (1.f / 8388608.f) - 1.f;
}

// operand image of a (rows, K) matrix: PLANES 1: [row][K] fp16; PLANES 2: [row][slab][plane][32] (slab_major = 0) or [slab][row][plane][32] (slab_major = 1)
__global__ void fill_image(_Float16* img, int rows, int K, int planes, int slab_major, float amp, unsigned seed) {
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long long)rows * K) return;
    const int r = (int)(id / K), k = (int)(id - (long long)r * K);
    const float v = hash_unit(id, seed) * amp;
    if (planes == 1) { img[id] = (_Float16)v; return; }
    const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
    const int slab = k >> 5, kk = k & 31, nslab = K >> 5;
    const long long base = slab_major ? ((long long)slab * rows + r) * 64 : ((long long)r * nslab + slab) * 64;
    img[base + kk] = hi; img[base + 32 + kk] = lo;
}
__global__ void fill_f32(float* x, long long n, float amp, unsigned seed) {
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id < n) x[id] = hash_unit(id, seed) * amp;
}

// fp64 reference of the same operand images: one thread per output
__global__ void ref_kernel(const _Float16* A, const _Float16* W, double* R, int M, int N, int K, int planes) {
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long long)M * N) return;
    const int m = (int)(id / N), n = (int)(id - (long long)m * N);
    double s = 0.0;
    if (planes == 1) {
        for (int k = 0; k < K; ++k) s += (double)(float)A[(long long)m * K + k] * (double)(float)W[(long long)n * K + k];
    } else {
        const int nslab = K >> 5;
        for (int sl = 0; sl < nslab; ++sl) {
            const _Float16* a = A + ((long long)m * nslab + sl) * 64;
            const _Float16* w = W + ((long long)sl * N + n) * 64;
            for (int k = 0; k < 32; ++k) {
                const double ah = (double)(float)a[k], al = (double)(float)a[32 + k], wh = (double)(float)w[k], wl = (double)(float)w[32 + k];
                s += al * wh + ah * wl + ah * wh;
            }
        }
    }
    R[id] = s;
}

__global__ void cmp_kernel(const void* C, int out_fmt, const double* R, const float* bias, const float* res, int M, int N, int act, double* maxerr, double* maxref) {
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long long)M * N) return;
    const int m = (int)(id / N), n = (int)(id - (long long)m * N);
    double v = R[id] + (bias ? (double)bias[n] : 0.0);
    if (act == 1) v = 0.5 * v * (1.0 + erf(v * 0.70710678118654752440));
    if (res) v += (double)res[id];
    double c;
    if (out_fmt == RG_F32) c = (double)reinterpret_cast<const float*>(C)[id];
    else if (out_fmt == RG_F16) c = (double)(float)reinterpret_cast<const _Float16*>(C)[id];
    else {
        const _Float16* p = reinterpret_cast<const _Float16*>(C) + ((long long)m * (N >> 5) + (n >> 5)) * 64 + (n & 31);
        c = (double)(float)p[0] + (double)(float)p[32];
    }
    const double e = fabs(c - v);
    // max via atomics on the bit pattern of non-negative doubles
    atomicMax(reinterpret_cast<unsigned long long*>(maxerr), (unsigned long long)__double_as_longlong(e));
    atomicMax(reinterpret_cast<unsigned long long*>(maxref), (unsigned long long)__double_as_longlong(fabs(v)));
}

struct Variant { const char* name; int BM, BN; size_t lds; void (*launch)(const RingParams&, int grid, hipStream_t); };

template <int GM, int GN, int TM, int TN, int PL, int S>
void launch_v(const RingParams& p, int grid, hipStream_t s) {
    using T = RingTile<GM, GN, TM, TN, PL, S>;
    static bool attr = false;
    if (!attr) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ring_gemm_kernel<GM, GN, TM, TN, PL, S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::kLdsBytes)); attr = true; }
    hipLaunchKernelGGL((ring_gemm_kernel<GM, GN, TM, TN, PL, S>), dim3(grid), dim3(512), T::kLdsBytes, s, p);
}
#define VARIANT(name, GM, GN, TM, TN, PL, S) Variant{name, RingTile<GM, GN, TM, TN, PL, S>::BM, RingTile<GM, GN, TM, TN, PL, S>::BN, RingTile<GM, GN, TM, TN, PL, S>::kLdsBytes, &launch_v<GM, GN, TM, TN, PL, S>}

template <int PL>
std::vector<Variant> variants() {
    return {
        VARIANT("256x128 w64x64 S3", 2, 2, 2, 2, PL, 3),
        VARIANT("128x128 w32x64 S3", 2, 2, 1, 2, PL, 3),
        VARIANT("128x128 w32x64 S4", 2, 2, 1, 2, PL, 4),
        VARIANT("128x128 w64x32 S4", 1, 4, 2, 1, PL, 4),
        VARIANT("128x256 w32x128 S3", 2, 2, 1, 4, PL, 3),
        VARIANT("256x256 w64x128 S2", 2, 2, 2, 4, PL, 2),
        VARIANT("128x64 w32x32 S4", 2, 2, 1, 1, PL, 4),
        VARIANT("128x128 w64x32 S2 (2 WG/CU)", 1, 4, 2, 1, PL, 2),
        VARIANT("128x128 w32x64 S2 (2 WG/CU)", 2, 2, 1, 2, PL, 2),
        VARIANT("128x64 w32x32 S2 (3 WG/CU)", 2, 2, 1, 1, PL, 2),
    };
}

struct Shape { int M, N, K, act, res; };

int main(int argc, char** argv) {
    const int planes_sel = argc > 1 ? atoi(argv[1]) : 0;
    const int rounds = argc > 2 ? atoi(argv[2]) : 7;
    const int check = argc > 3 ? atoi(argv[3]) : 1;
    const char* only = getenv("RB_VARIANTS");       // comma-separated variant indices
    std::vector<Shape> shapes = {
        {19200, 384, 384, 0, 0}, {19200, 1536, 384, 1, 0}, {19200, 384, 1536, 0, 1},
        {4800, 768, 768, 0, 0}, {4800, 3072, 768, 1, 0}, {4800, 768, 3072, 0, 1},
        {76800, 192, 768, 0, 1}, {76800, 768, 192, 1, 0}, {76800, 192, 192, 0, 0}, {307200, 96, 384, 0, 1},
        {8192, 4096, 4096, 0, 0},
    };
    if (getenv("RB_SHAPES")) {      // "M,N,K,act,res;..."
        shapes.clear();
        std::string s = getenv("RB_SHAPES");
        size_t pos = 0;
        while (pos < s.size()) {
            size_t e = s.find(';', pos); if (e == std::string::npos) e = s.size();
            Shape sh{}; sscanf(s.substr(pos, e - pos).c_str(), "%d,%d,%d,%d,%d", &sh.M, &sh.N, &sh.K, &sh.act, &sh.res);
            shapes.push_back(sh); pos = e + 1;
        }
    }
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("XP_RING_DBG = %d\n", XP_RING_DBG);
    for (int planes = 1; planes <= 2; ++planes) {
        if (planes_sel && planes_sel != planes) continue;
        std::vector<Variant> vs = planes == 1 ? variants<1>() : variants<2>();
        for (const Shape& sh : shapes) {
            const int M = sh.M, N = sh.N, K = sh.K;
            const int bk = planes == 1 ? 64 : 32;
            if (K % bk) continue;
            const size_t a_elems = (size_t)M * K * planes, w_elems = (size_t)N * K * planes;
            _Float16 *A, *W; float *bias, *res, *wscale; void* C; double *R, *stats;
            CK(hipMalloc(&A, a_elems * 2)); CK(hipMalloc(&W, w_elems * 2)); CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&wscale, N * 4));
            CK(hipMalloc(&res, (size_t)M * N * 4)); CK(hipMalloc(&C, (size_t)M * N * 4)); CK(hipMalloc(&stats, 16));
            fill_image<<<(unsigned)(((size_t)M * K + 255) / 256), 256, 0, st>>>(A, M, K, planes, 0, 1.f, 1u);
            fill_image<<<(unsigned)(((size_t)N * K + 255) / 256), 256, 0, st>>>(W, N, K, planes, 1, 0.05f, 2u);
            fill_f32<<<(N + 255) / 256, 256, 0, st>>>(bias, N, 0.5f, 3u);
            fill_f32<<<(unsigned)(((size_t)M * N + 255) / 256), 256, 0, st>>>(res, (long long)M * N, 1.f, 4u);
            R = nullptr;
            const bool do_check = check && (double)M * N * K < 4e10;
            if (do_check) {
                CK(hipMalloc(&R, (size_t)M * N * 8));
                ref_kernel<<<(unsigned)(((size_t)M * N + 255) / 256), 256, 0, st>>>(A, W, R, M, N, K, planes);
            }
            CK(hipStreamSynchronize(st));
            RingParams p{};
            p.A = (const char*)A; p.W = (const char*)W;
            if (planes == 1) { p.a_row = (int64_t)K * 2; p.a_slab = 128; p.w_row = (int64_t)K * 2; p.w_slab = 128; }
            else { p.a_row = (int64_t)(K / 32) * 128; p.a_slab = 128; p.w_row = 128; p.w_slab = (int64_t)N * 128; }
            p.M = M; p.N = N; p.T = K / bk;
            p.C = C; p.ldc = N; p.wscale = nullptr; p.bias = bias; p.scale = nullptr; p.shift = nullptr;
            p.res = sh.res ? res : nullptr; p.ldres = N; p.res_fmt = RG_F32; p.act = sh.act; p.r16 = 0; p.ngroup = 0;
            printf("\nplanes %d  M %d N %d K %d act %d res %d   (%.2f GF%s, operands %.1f MB, out %.1f MB)\n", planes, M, N, K, sh.act, sh.res, 2.0 * M * N * K * 1e-9,
                   planes == 2 ? " x3 products" : "", (a_elems + w_elems) * 2e-6, (double)M * N * 4e-6);
            std::vector<int> sel;
            for (int v = 0; v < (int)vs.size(); ++v) {
                if (only) { bool in = false; std::string o = only; size_t q = 0; while (q < o.size()) { size_t e = o.find(',', q); if (e == std::string::npos) e = o.size(); if (atoi(o.substr(q, e - q).c_str()) == v) in = true; q = e + 1; } if (!in) continue; }
                sel.push_back(v);
            }
            std::vector<std::vector<float>> times(vs.size());
            for (int fmt_pass = 0; fmt_pass < 1; ++fmt_pass) {
                // correctness, every output format once per variant
                for (int v : sel) {
                    const int grid = ((M + vs[v].BM - 1) / vs[v].BM) * ((N + vs[v].BN - 1) / vs[v].BN);
                    if (do_check && XP_RING_DBG == 0) {
                        for (int fmt = 0; fmt < 3; ++fmt) {
                            if (fmt == RG_P32 && N % 32) continue;
                            p.out_fmt = fmt;
                            CK(hipMemsetAsync(C, 0xff, (size_t)M * N * 4, st));
                            CK(hipMemsetAsync(stats, 0, 16, st));
                            vs[v].launch(p, grid, st);
                            cmp_kernel<<<(unsigned)(((size_t)M * N + 255) / 256), 256, 0, st>>>(C, fmt, R, bias, p.res ? res : nullptr, M, N, sh.act, stats, stats + 1);
                            double h[2]; CK(hipMemcpyAsync(h, stats, 16, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
                            const double tol = fmt == RG_F16 ? 2e-3 * h[1] + 1e-3 : (planes == 1 ? 1e-4 * h[1] + 2e-4 : 2e-6 * h[1] + 2e-6) * (sh.act ? 4 : 1);
                            if (!(h[0] <= tol)) printf("  !! %-20s fmt %d: max |err| %.3e (max |ref| %.3e) exceeds %.1e\n", vs[v].name, fmt, h[0], h[1], tol);
                            else if (fmt == 0) printf("  ok %-20s max |err| %.2e of %.2e\n", vs[v].name, h[0], h[1]);
                        }
                    }
                }
            }
            p.out_fmt = planes == 1 ? RG_F16 : RG_F32;
            const int reps = 10;
            for (int r = 0; r < rounds + 1; ++r)
                for (int v : sel) {
                    const int grid = ((M + vs[v].BM - 1) / vs[v].BM) * ((N + vs[v].BN - 1) / vs[v].BN);
                    CK(hipEventRecord(e0, st));
                    for (int i = 0; i < reps; ++i) vs[v].launch(p, grid, st);
                    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (r > 0) times[v].push_back(ms / reps);
                }
            for (int v : sel) {
                std::sort(times[v].begin(), times[v].end());
                const double med = times[v][times[v].size() / 2] * 1e-3, mn = times[v][0] * 1e-3;
                const int gm = (M + vs[v].BM - 1) / vs[v].BM, gn = (N + vs[v].BN - 1) / vs[v].BN;
                const double flops = 2.0 * M * N * K * (planes == 2 ? 3 : 1);
                const double dma = (double)gm * gn * (vs[v].BM + vs[v].BN) * 128.0 * p.T;
                printf("  %-20s grid %5d (%.2f rounds)  median %7.1f us  min %7.1f us   %7.1f TF/s executed (%.3f of 2.5 PF)   DMA %6.2f TB/s = %5.1f B/clk/CU at 2.1 GHz\n", vs[v].name, gm * gn,
                       gm * gn / 256.0, med * 1e6, mn * 1e6, flops / med * 1e-12, flops / med / 2.5e15, dma / med * 1e-12, dma / med / 256 / 2.1e9);
            }
            CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(bias)); CK(hipFree(wscale)); CK(hipFree(res)); CK(hipFree(C)); CK(hipFree(stats)); if (R) CK(hipFree(R));
        }
    }
    return 0;
}
