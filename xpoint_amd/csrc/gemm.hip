// fp32 MFMA GEMM / implicit-GEMM 3x3 convolution with fused epilogue, for the encoder's dense
// layers (reference VMamba.py:649,663 in/out_proj; :110-128 Mlp; :605 x_proj; :1405-1440 strided
// 3x3 convs; XPoint.py:112-138 head convs).
//
//   C[m, n] = epilogue( sum_k A'[m, k] * Wt[n, k] )        A' = A (row-major M x K)  or  im2col(NHWC image)
//   epilogue(v) = (act(v + bias[n]) * scale[n] + shift[n]) + res[m, n]
//
// Tile engine: gemm_core.h.  MFMA-bound for the deep stages, HBM-bound at stage 0 (K = N = 96).
#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <string>
#include <vector>

#include "gemm_core.h"
#include "gemm_epilogue.h"

namespace {

template <int WM, int WN, int TM, int TN, int MODE>
__global__ __launch_bounds__(WM * WN * 64) void gemm_kernel(GemmParams p) {
    using T = GemmTile<WM, WN, TM, TN>;
    extern __shared__ __align__(16) float lds[];
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so block b runs
    // on XCD b % 8.  Give every XCD a contiguous range of logical tiles with the N-tiles of one M-tile adjacent, so
    // the A rows shared by those tiles are fetched once per XCD instead of once per tile (bijective remap;
    // placement only affects speed, never results).
    const int ntn = (p.N + T::BN - 1) / T::BN;
    const int total = gridDim.x;
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int q = total >> 3, r = total & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    const int m0 = (logical / ntn) * T::BM, n0 = (logical % ntn) * T::BN;

    // De-phase the two workgroups that share a CU.  Co-resident workgroups start together and take equal time, so they
    // run in lockstep: both in their MFMA phase (sharing the matrix pipe), then both in their epilogue (pipe idle).
    // Delaying the second resident wave of workgroups (ids 256..511 fill the second slot of the 256 CUs) by about half
    // a workgroup period makes one's epilogue overlap the other's MFMAs; later workgroups inherit the offset because
    // slots are refilled as workgroups retire.  Timing only — results do not depend on it.
    if (p.stagger_cycles > 0 && bid < 1024) {
        const long long delay = ((long long)((bid * 40503u) & 0xffffu) * p.stagger_cycles) >> 16;   // pseudo-random phase in [0, stagger)
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while ((long long)(__builtin_amdgcn_s_memtime() - t0) < delay) __builtin_amdgcn_s_sleep(8);
    }

    unsigned long long st0 = 0, st1 = 0, st2 = 0;
    if (p.stamps) st0 = __builtin_amdgcn_s_memtime();
    // Staging loads are branch-free: out-of-range rows / k are redirected to a valid address and zeroed by a select,
    // so the 8 loads of a slab issue back to back (divergent "if (ok) load" regions made hipcc serialise them).
    const float* a_ptr[T::A_LD];     // plain: row base; conv: image base
    int a_oh[T::A_LD], a_ow[T::A_LD];
    bool a_ok[T::A_LD];
#pragma unroll
    for (int s = 0; s < T::A_LD; ++s) {
        const int m = m0 + T::slot_row(s);
        a_ok[s] = m < p.M;
        const int mc = a_ok[s] ? m : 0;
        if (MODE == 0) {
            a_ptr[s] = p.A + (int64_t)mc * p.lda; a_oh[s] = a_ow[s] = 0;
        } else {
            const int hw = p.Ho * p.Wo;
            const int b = mc / hw, rr = mc - b * hw;
            a_oh[s] = (rr / p.Wo) * p.stride - 1; a_ow[s] = (rr % p.Wo) * p.stride - 1;
            a_ptr[s] = p.A + (int64_t)b * p.Hi * p.Wi * p.Ci;
        }
    }
    const float* wrow[T::B_LD];
    bool w_ok[T::B_LD];
#pragma unroll
    for (int s = 0; s < T::B_LD; ++s) {
        const int n = n0 + T::slot_row(s);
        w_ok[s] = n < p.N;
        wrow[s] = p.Wt + (int64_t)(w_ok[s] ? n : 0) * p.K;
    }
    const int kmax = p.K - 4;
    // conv mode: (tap, ci) of each staging slot advance by one slab per call — no integer division in the K loop
    int a_tap[T::A_LD], a_ci[T::A_LD];
#pragma unroll
    for (int s = 0; s < T::A_LD; ++s) {
        const int k = T::slot_kq(s) * 4;
        a_tap[s] = (MODE == 1) ? k / p.Ci : 0;
        a_ci[s] = (MODE == 1) ? k - a_tap[s] * p.Ci : 0;
    }
    auto ldA = [&](int s, int k, bool& ok) -> float4 {
        ok = a_ok[s] && k < p.K;
        const int kc = k < p.K ? k : kmax;
        float4 v;
        if (MODE == 0) {
            v = *reinterpret_cast<const float4*>(a_ptr[s] + kc);
        } else {
            int tap = a_tap[s], ci = a_ci[s];
            if (k >= p.K) { tap = 8; ci = p.Ci - 4; }            // past the end: any valid address, value is zeroed
            a_ci[s] += XP_BK;                                      // state for the next slab (calls come in slab order)
            while (a_ci[s] >= p.Ci) { a_ci[s] -= p.Ci; ++a_tap[s]; }
            int ih = a_oh[s] + tap / 3, iw = a_ow[s] + tap % 3;
            if (p.reflect) {
                ih = ih < 0 ? -ih : (ih >= p.Hi ? 2 * p.Hi - 2 - ih : ih);
                iw = iw < 0 ? -iw : (iw >= p.Wi ? 2 * p.Wi - 2 - iw : iw);
            } else {
                ok = ok && ih >= 0 && ih < p.Hi && iw >= 0 && iw < p.Wi;
                ih = ih < 0 ? 0 : (ih >= p.Hi ? p.Hi - 1 : ih);
                iw = iw < 0 ? 0 : (iw >= p.Wi ? p.Wi - 1 : iw);
            }
            v = *reinterpret_cast<const float4*>(a_ptr[s] + ((int64_t)ih * p.Wi + iw) * p.Ci + ci);
        }
        return v;
    };
    auto ldB = [&](int s, int k, bool& ok) -> float4 {
        ok = w_ok[s] && k < p.K;
        return *reinterpret_cast<const float4*>(wrow[s] + (k < p.K ? k : kmax));
    };

    f32x16 acc[TM][TN];
    T::run(lds, p.K, ldA, ldB, acc);
    if (p.stamps) st1 = __builtin_amdgcn_s_memtime();

    gemm_epilogue<T, TM, TN>(p, m0, n0, acc);
    if (p.stamps) {
        st2 = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0) {
            unsigned long long* o = p.stamps + (size_t)blockIdx.x * 4;
            o[0] = st0; o[1] = st1; o[2] = st2; o[3] = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) /* HW_REG_XCC_ID */;
        }
    }
}

template <int WM, int WN, int TM, int TN>
void launch(const GemmParams& p, hipStream_t s) {
    using T = GemmTile<WM, WN, TM, TN>;
    static XpPerDeviceOnce attr_once;
    if (T::kLdsBytes > 64 * 1024 && attr_once.need()) {
        XP_HIP_WARN(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<WM, WN, TM, TN, 0>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::kLdsBytes));
        XP_HIP_WARN(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<WM, WN, TM, TN, 1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::kLdsBytes));
    }
    dim3 grid(xp_cdiv(p.N, T::BN) * xp_cdiv(p.M, T::BM));
    static const bool by_shape = getenv("XP_PROF_SHAPES") != nullptr;
    // one tag per kernel instantiation (tile x mode), so the HIP-event averages line up with rocprofv3's per-kernel rows
    std::string tag = std::string(p.mode ? "conv3x3_f32_mfma_" : "gemm_f32_mfma_") + std::to_string(T::BM) + "x" + std::to_string(T::BN);
    if (by_shape) tag += "_M" + std::to_string(p.M) + "_N" + std::to_string(p.N) + "_K" + std::to_string(p.K) + (p.act == 1 ? "_gelu" : "");
    const double in_elems = p.mode == 0 ? (double)p.M * p.K : (double)p.M / (p.Ho * p.Wo) * p.Hi * p.Wi * p.Ci;
    XpProfScope prof(tag.c_str(), s, 2.0 * p.M * p.N * p.K,
                     4.0 * (in_elems + (double)p.N * p.K + (double)p.M * p.N * (p.res ? 2 : 1)));
    GemmParams q = p;
    static const bool want_stamps = getenv("XP_GEMM_STAMPS") != nullptr;
    static unsigned long long* stamp_buf = nullptr;
    if (want_stamps) {
        if (!stamp_buf) (void)hipMalloc(&stamp_buf, sizeof(unsigned long long) * 4 * 65536);
        if (grid.x <= 65536) q.stamps = stamp_buf;
    }
    if (p.mode == 0) hipLaunchKernelGGL((gemm_kernel<WM, WN, TM, TN, 0>), grid, dim3(T::NT), T::kLdsBytes, s, q);
    else hipLaunchKernelGGL((gemm_kernel<WM, WN, TM, TN, 1>), grid, dim3(T::NT), T::kLdsBytes, s, q);
    if (q.stamps) {   // debug only: synchronises
        (void)hipStreamSynchronize(s);
        std::vector<unsigned long long> h((size_t)grid.x * 4);
        (void)hipMemcpy(h.data(), stamp_buf, h.size() * 8, hipMemcpyDeviceToHost);
        double kloop = 0, epi = 0; unsigned long long tmin = ~0ull, tmax = 0;
        for (unsigned b = 0; b < grid.x; ++b) {
            kloop += (double)(h[b * 4 + 1] - h[b * 4]); epi += (double)(h[b * 4 + 2] - h[b * 4 + 1]);
            tmin = std::min(tmin, h[b * 4]); tmax = std::max(tmax, h[b * 4 + 2]);
        }
        fprintf(stderr, "[stamps] %s grid %u: prologue+K-loop %.0f cyc, epilogue %.0f cyc per workgroup; kernel span %.0f cyc (s_memtime ticks)\n",
                tag.c_str(), grid.x, kloop / grid.x, epi / grid.x, (double)(tmax - tmin));
    }
}

int dispatch(GemmParams p, hipStream_t s) {
    static const int stagger_env = getenv("XP_GEMM_STAGGER") ? atoi(getenv("XP_GEMM_STAGGER")) : -1;
    if (stagger_env >= 0) p.stagger_cycles = stagger_env;
    // tile choice by N (the encoder's N are 32..3072; M is large except at the last stage)
    const int N = p.N;
    if (N <= 32) launch<4, 1, 1, 1>(p, s);                                   // 128 x 32
    else if (N <= 64) launch<4, 1, 1, 2>(p, s);                              // 128 x 64
    else if (N <= 96 || (N % 96 == 0 && (N / 96) % 4 != 0)) launch<4, 1, 1, 3>(p, s);   // 128 x 96  (N = 65..96, 192)
    else if (p.M <= 8192 && N >= 512) launch<2, 2, 1, 2>(p, s);              // 64 x 128: more blocks when M is small
    else launch<2, 2, 2, 2>(p, s);                                           // 128 x 128
    XP_LAUNCH_CHECK();
    return XP_OK;
}

}  // namespace

extern "C" int xp_gemm_nt(const float* A, const float* Wt, float* C, const float* bias, const float* scale,
                          const float* shift, const float* res, int M, int N, int K, int lda, int ldc, int ldres,
                          int act, void* stream) {
    XP_CHECK_ARG(A && Wt && C, "xp_gemm_nt: null pointer");
    XP_CHECK_ARG(M > 0 && N > 0 && K > 0, "xp_gemm_nt: bad shape %d %d %d", M, N, K);
    XP_CHECK_ARG(K % 4 == 0 && lda % 4 == 0, "xp_gemm_nt: K and lda must be multiples of 4 (got %d, %d)", K, lda);
    XP_CHECK_ARG((scale == nullptr) == (shift == nullptr), "xp_gemm_nt: scale and shift go together");
    XP_CHECK_ARG(act >= 0 && act <= 3, "xp_gemm_nt: bad act %d", act);
    GemmParams p{};
    p.A = A; p.Wt = Wt; p.C = C; p.bias = bias; p.scale = scale; p.shift = shift; p.res = res;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldc = ldc; p.ldres = ldres; p.act = act; p.mode = 0;
    return dispatch(p, (hipStream_t)stream);
}

extern "C" int xp_conv3x3_nhwc(const float* x, const float* Wt, float* y, const float* bias, const float* scale,
                               const float* shift, int batch, int Hi, int Wi, int Ci, int Co, int stride,
                               int reflect_pad, int act, void* stream) {
    XP_CHECK_ARG(x && Wt && y, "xp_conv3x3_nhwc: null pointer");
    XP_CHECK_ARG(Ci % 4 == 0, "xp_conv3x3_nhwc: Ci must be a multiple of 4 (got %d)", Ci);
    XP_CHECK_ARG(stride == 1 || stride == 2, "xp_conv3x3_nhwc: stride 1 or 2");
    XP_CHECK_ARG((scale == nullptr) == (shift == nullptr), "xp_conv3x3_nhwc: scale and shift go together");
    XP_CHECK_ARG(!reflect_pad || (Hi >= 2 && Wi >= 2), "xp_conv3x3_nhwc: reflection pad needs H,W >= 2");
    GemmParams p{};
    p.A = x; p.Wt = Wt; p.C = y; p.bias = bias; p.scale = scale; p.shift = shift; p.res = nullptr;
    p.Hi = Hi; p.Wi = Wi; p.Ci = Ci; p.stride = stride; p.reflect = reflect_pad;
    p.Ho = (Hi + 2 - 3) / stride + 1; p.Wo = (Wi + 2 - 3) / stride + 1;
    p.M = batch * p.Ho * p.Wo; p.N = Co; p.K = 9 * Ci; p.lda = 0; p.ldc = Co; p.ldres = 0; p.act = act; p.mode = 1;
    return dispatch(p, (hipStream_t)stream);
}
