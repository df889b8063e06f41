// Entry points of the ring dense engine (ring_core.h): the split-fp16 GEMM on PRE-SPLIT activations ("h2s": the f32-grade class, stages 2 - 3 of the encoder:
// reference VMamba.py:649,663 in/out_proj, :110-128 Mlp), the producer-side f32 -> P32 converter, and the one-product fp16 instance the fast mixed-precision
// class's xp_gemm_nt_f16 dispatches its long-K layers to.
//
//   C[m, n] = epilogue( 2^-k_n * sum_k A[m, k] * (2^k_n W[n, k]) )
//   A: "P32" image  [m][slab = k / 32][plane 0..1][32] fp16, A = plane0 + plane1 (xp_split_activations_h2, xp_layernorm_p32, the P32 outputs of
//      xp_ss2d_core_fwd_ex and of this GEMM itself): 4 bytes per element like the f32 tensor it replaces; K % 32 == 0
//   W: xp_split_weights_h2's [slab][n][plane][32] + the N inverse row scales (gemm_h2.hip)
// Every tile shape walks K in the same order with one accumulator per output element (no split K), so the result never depends on the tile and the tile may
// be chosen from M (the batch): the largest of 256 x 256 / 256 x 128 / 128 x 128 that still gives the chip >= 192 workgroups (tools/ring_bench.hip,
// profiles/r5_ring_microbench.txt).  Summation order = gemm_h2_core.h's tile engine: bit-identical to xp_gemm_nt_h2 wherever that runs its tile kernel.
#include <stdlib.h>

#include <string>

#include "ring_core.h"
#include "../../include/xpoint_hip.h"

namespace {

template <int GM, int GN, int TM, int TN, int PL, int S>
void ring_launch(const RingParams& p, hipStream_t s, const char* name, double flops, double bytes) {
    using T = RingTile<GM, GN, TM, TN, PL, S>;
    static XpPerDeviceOnce attr_once;
    if (attr_once.need()) XP_HIP_WARN(hipFuncSetAttribute(reinterpret_cast<const void*>(&ring_gemm_kernel<GM, GN, TM, TN, PL, S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::kLdsBytes));
    const int grid = xp_cdiv(p.M, T::BM) * xp_cdiv(p.N, T::BN);
    static const bool by_shape = getenv("XP_PROF_SHAPES") != nullptr;
    std::string tag = std::string(name) + "_" + std::to_string(T::BM) + "x" + std::to_string(T::BN);
    if (by_shape) tag += "_M" + std::to_string(p.M) + "_N" + std::to_string(p.N) + "_K" + std::to_string(p.T * (PL == 1 ? 64 : 32)) + (p.act == 1 ? "_gelu" : "");
    XpProfScope prof(tag.c_str(), s, flops, bytes);
    hipLaunchKernelGGL((ring_gemm_kernel<GM, GN, TM, TN, PL, S>), dim3(grid), dim3(512), T::kLdsBytes, s, p);
}

template <int PL>
void ring_dispatch(RingParams& p, hipStream_t s, const char* name, double flops, double bytes) {
    static const int force = getenv("XP_RING_TILE") ? atoi(getenv("XP_RING_TILE")) : -1;      // tuning experiments only: 0 = 256 x 256, 1 = 256 x 128, 2 = 128 x 128
    auto tiles = [&](int bm, int bn) { return (int64_t)xp_cdiv(p.M, bm) * xp_cdiv(p.N, bn); };
    int sel = tiles(256, 256) >= 192 && p.N > 128 ? 0 : tiles(256, 128) >= 192 ? 1 : 2;
    if (force >= 0) sel = force;
    // column-tile groups: the weight slabs of one group of column tiles should stay in an XCD's L2 (4 MB) next to the activation rows streaming past them
    {
        const int bn = sel == 0 ? 256 : 128, ntn = xp_cdiv(p.N, bn);
        const int64_t wbytes_per_tile = (int64_t)bn * p.T * 128;
        p.ngroup = 0;
        if ((int64_t)ntn * wbytes_per_tile > (3 << 20) && ntn > 2) {
            const int gmax = (int)std::max<int64_t>(1, (int64_t)(2 << 20) / wbytes_per_tile);
            const int ngroups = xp_cdiv(ntn, gmax);
            p.ngroup = ngroups > 1 ? xp_cdiv(ntn, ngroups) : 0;
        }
    }
    switch (sel) {
        case 0: ring_launch<2, 2, 2, 4, PL, 2>(p, s, name, flops, bytes); break;
        case 1: ring_launch<2, 2, 2, 2, PL, 3>(p, s, name, flops, bytes); break;
        default: ring_launch<1, 4, 2, 1, PL, 4>(p, s, name, flops, bytes); break;
    }
}

// f32 rows -> P32 image: one thread per 8 consecutive elements (two 16-byte loads, two 16-byte stores)
__global__ __launch_bounds__(256) void split_activations_kernel(const float* __restrict__ x, unsigned char* __restrict__ out, int64_t M, int K, int ldx) {
    const int k8 = K >> 3;
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= M * k8) return;
    const int64_t m = id / k8;
    const int k = (int)(id - m * k8) * 8;
    const float4 a = *reinterpret_cast<const float4*>(x + m * ldx + k), b = *reinterpret_cast<const float4*>(x + m * ldx + k + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    rg_h8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) { hi[e] = (_Float16)v[e]; lo[e] = (_Float16)(v[e] - (float)hi[e]); }
    unsigned char* o = out + (m * (K >> 5) + (k >> 5)) * 128 + (k & 31) * 2;
    *reinterpret_cast<rg_h8*>(o) = hi;
    *reinterpret_cast<rg_h8*>(o + 64) = lo;
}

}  // namespace

static int ring_enabled() {
    static const int on = getenv("XP_RING") ? atoi(getenv("XP_RING")) : 1;      // A/B runs: XP_RING=0 keeps every layer on the round-4 kernels
    return on;
}

extern "C" size_t xp_p32_bytes(int64_t M, int K) { return M > 0 && K > 0 ? (size_t)M * ((K + 31) / 32) * 128 : 0; }

extern "C" int xp_split_activations_h2(const float* x, void* out_p32, int64_t M, int K, int ldx, void* stream) {
    XP_CHECK_ARG(x && out_p32, "xp_split_activations_h2: null pointer");
    XP_CHECK_ARG(M >= 0 && K > 0 && K % 32 == 0 && ldx >= K && ldx % 4 == 0, "xp_split_activations_h2: K must be a multiple of 32 and ldx a multiple of 4 (got K = %d, ldx = %d)", K, ldx);
    XP_CHECK_ARG((((uintptr_t)x | (uintptr_t)out_p32) & 15) == 0, "xp_split_activations_h2: buffers must be 16-byte aligned");
    if (M == 0) return XP_OK;
    XpProfScope prof("split_activations_h2", (hipStream_t)stream, 0.0, 8.0 * M * K);
    hipLaunchKernelGGL(split_activations_kernel, dim3((unsigned)xp_cdiv(M * (K / 8), 256)), dim3(256), 0, (hipStream_t)stream, x, (unsigned char*)out_p32, M, K, ldx);
    XP_LAUNCH_CHECK();
    return XP_OK;
}

// Does the split class route this layer to the ring engine?  A property of the LAYER (N, K) only — the producers of the layer's input must know whether to
// write the P32 image, and a result must not depend on the batch (both engines give the same bits only where gemm_h2 runs its tile kernel).
extern "C" int xp_gemm_nt_h2s_applies(int N, int K) {
    return ring_enabled() && !xp_amp_value() && K % 32 == 0 && K >= 256 && N % 8 == 0 && N >= 256;
}

extern "C" int xp_gemm_nt_h2s(const void* A_p32, const void* Wh2, void* C, int out_fmt, const float* bias, const float* scale, const float* shift,
                              const float* res, int M, int N, int K, int ldc, int ldres, int act, void* stream) {
    XP_CHECK_ARG(A_p32 && Wh2 && C, "xp_gemm_nt_h2s: null pointer");
    XP_CHECK_ARG(M > 0 && N > 0 && K > 0, "xp_gemm_nt_h2s: bad shape %d %d %d", M, N, K);
    XP_CHECK_ARG(K % 32 == 0, "xp_gemm_nt_h2s: K must be a multiple of 32 (whole P32 slabs; got %d)", K);
    XP_CHECK_ARG(N % 8 == 0, "xp_gemm_nt_h2s: N must be a multiple of 8 (got %d)", N);
    XP_CHECK_ARG(out_fmt == RG_F32 || out_fmt == RG_P32, "xp_gemm_nt_h2s: out_fmt must be 0 (f32 rows) or 2 (P32 image)");
    XP_CHECK_ARG(out_fmt != RG_P32 || (N % 32 == 0 && ldc == N), "xp_gemm_nt_h2s: a P32 output needs N %% 32 == 0 and ldc == N (got %d, %d)", N, ldc);
    XP_CHECK_ARG(ldc % 4 == 0 && (!res || ldres % 4 == 0), "xp_gemm_nt_h2s: ldc / ldres must be multiples of 4");
    XP_CHECK_ARG((((uintptr_t)A_p32 | (uintptr_t)Wh2 | (uintptr_t)C | (uintptr_t)res) & 15) == 0, "xp_gemm_nt_h2s: buffers must be 16-byte aligned");
    XP_CHECK_ARG((scale == nullptr) == (shift == nullptr), "xp_gemm_nt_h2s: scale and shift go together");
    XP_CHECK_ARG(act >= 0 && act <= 3, "xp_gemm_nt_h2s: bad act %d", act);
    XP_CHECK_ARG(!xp_amp_value(), "xp_gemm_nt_h2s: the f32-container mixed-precision class (xp_set_amp_mode) runs on xp_gemm_nt_h2");
    const int T = K / 32;
    XP_CHECK_ARG((int64_t)M * T * 128 < (1ll << 32) && (int64_t)N * T * 128 < (1ll << 32), "xp_gemm_nt_h2s: operand images must stay below 4 GB");
    RingParams p{};
    p.A = (const char*)A_p32; p.W = (const char*)Wh2;
    p.a_row = (int64_t)T * 128; p.a_slab = 128; p.w_row = 128; p.w_slab = (int64_t)N * 128;
    p.M = M; p.N = N; p.T = T;
    p.C = C; p.ldc = ldc; p.out_fmt = out_fmt;
    p.wscale = reinterpret_cast<const float*>(reinterpret_cast<const char*>(Wh2) + (size_t)N * T * 128);      // the inverse row scales sit right behind the planes
    p.bias = bias; p.scale = scale; p.shift = shift; p.res = res; p.ldres = ldres; p.res_fmt = RG_F32; p.act = act; p.r16 = 0;
    ring_dispatch<2>(p, (hipStream_t)stream, "gemm_ring_h2s", 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N * (res ? 2 : 1)));
    XP_LAUNCH_CHECK();
    return XP_OK;
}
