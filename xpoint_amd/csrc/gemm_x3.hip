// fp32-accurate GEMM / implicit-GEMM 3x3 convolution on the bf16 matrix pipe (split operands, 6 partial products;
// gemm_x3_core.h) with the same fused epilogue and the same call sites as gemm.hip (reference VMamba.py:649,663
// in/out_proj; :110-128 Mlp; :605 x_proj; :1405-1440 strided 3x3 convs; XPoint.py:112-138 head convs).
//
//   C[m, n] = epilogue( sum_k A'[m, k] * W[n, k] )     A' = A (row-major M x K, f32) or im2col(NHWC f32 image)
//   W is given pre-split: xp_split_weights_x3 turns the (N, K) f32 matrix into
//       Wx3[slab = k / 16][n][plane 0..2][16] bf16      (K zero-padded to a multiple of 16)
//   once per weight upload: slab-major, so the BN rows a workgroup needs for one slab are BN * 96 contiguous bytes.
#include <stdlib.h>

#include <string>

#include "gemm_x3_core.h"

namespace {

__global__ void split_weights_kernel(const float* __restrict__ W, uint4* __restrict__ out, int N, int K, int nslab) {
    // one thread per (n, slab, octet): 8 floats -> 3 x 16 B
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (int64_t)N * nslab * 2) return;
    const int oct = (int)(id & 1);
    const int64_t ns = id >> 1;                       // n * nslab + slab
    const int n = (int)(ns / nslab), slab = (int)(ns - (int64_t)n * nslab);
    const int k = slab * X3_BK + oct * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (k + j < K) ? W[(int64_t)n * K + k + j] : 0.f;
    uint4 p0, p1, p2;
    xp_split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), p0, p1, p2);
    uint4* o = out + ((int64_t)slab * N + n) * X3_SLAB_UNITS + oct;
    o[0] = p0; o[2] = p1; o[4] = p2;
}

template <int WM, int WN, int TM, int TN, int MODE, int NP>
__global__ __launch_bounds__(WM * WN * 64) void gemm_x3_kernel(GemmParams p) {
    using T = GemmTileX3<WM, WN, TM, TN, NP>;
    extern __shared__ __align__(16) unsigned char lds_x3[];
    // XCD-aware tile order (same bijective remap as gemm.hip: each XCD gets a contiguous run of logical tiles, the
    // N-tiles of one M-tile adjacent, so the A rows they share come from that XCD's L2)
    const int ntn = (p.N + T::BN - 1) / T::BN;
    const int total = gridDim.x;
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int q = total >> 3, r = total & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    const int m0 = (logical / ntn) * T::BM, n0 = (logical % ntn) * T::BN;

    const float* a_ptr[T::A_LD];     // plain: row base; conv: image base
    int a_oh[T::A_LD], a_ow[T::A_LD], a_tap[T::A_LD], a_ci[T::A_LD];
#pragma unroll
    for (int s = 0; s < T::A_LD; ++s) {
        const int m = m0 + T::a_row(s);
        const int mc = m < p.M ? m : 0;          // rows past M only feed output rows that are never stored
        if (MODE == 0) {
            a_ptr[s] = p.A + (int64_t)mc * p.lda; a_oh[s] = a_ow[s] = a_tap[s] = a_ci[s] = 0;
        } else {
            const int hw = p.Ho * p.Wo;
            const int b = mc / hw, rr = mc - b * hw;
            a_oh[s] = (rr / p.Wo) * p.stride - 1; a_ow[s] = (rr % p.Wo) * p.stride - 1;
            a_ptr[s] = p.A + (int64_t)b * p.Hi * p.Wi * p.Ci;
            const int k = T::a_quad(s) * 4;      // (tap, ci) advance by one slab per call: no division in the K loop
            a_tap[s] = k / p.Ci; a_ci[s] = k - a_tap[s] * p.Ci;
        }
    }
    const int nslab = (p.K + X3_BK - 1) / X3_BK;
    const uint4* w_unit[T::B_LD];
#pragma unroll
    for (int s = 0; s < T::B_LD; ++s) {
        const int n = n0 + T::b_row(s);
        w_unit[s] = reinterpret_cast<const uint4*>(p.Wt) + (int64_t)(n < p.N ? n : 0) * X3_SLAB_UNITS + T::b_unit(s);
    }
    const int kmax = p.K - 4;
    const int64_t w_slab = (int64_t)p.N * X3_SLAB_UNITS;     // 16-byte units per slab of the whole weight matrix
    auto ldA = [&](int s, int k, float4& v) -> bool {
        bool ok = k < p.K;
        const float* src;
        if (MODE == 0) {
            src = a_ptr[s] + (ok ? k : kmax);
        } else {
            int tap = a_tap[s], ci = a_ci[s];
            if (!ok) { tap = 8; ci = p.Ci - 4; }
            a_ci[s] += X3_BK;                              // Ci >= 4: at most four wraps per 16-wide slab, as selects
#pragma unroll
            for (int w = 0; w < 4; ++w) { const bool wrap = a_ci[s] >= p.Ci; a_ci[s] -= wrap ? p.Ci : 0; a_tap[s] += wrap ? 1 : 0; }
            int ih = a_oh[s] + tap / 3, iw = a_ow[s] + tap % 3;
            if (p.reflect) {
                ih = ih < 0 ? -ih : (ih >= p.Hi ? 2 * p.Hi - 2 - ih : ih);
                iw = iw < 0 ? -iw : (iw >= p.Wi ? 2 * p.Wi - 2 - iw : iw);
            } else {
                ok = ok && ih >= 0 && ih < p.Hi && iw >= 0 && iw < p.Wi;
                ih = ih < 0 ? 0 : (ih >= p.Hi ? p.Hi - 1 : ih);
                iw = iw < 0 ? 0 : (iw >= p.Wi ? p.Wi - 1 : iw);
            }
            src = a_ptr[s] + ((int64_t)ih * p.Wi + iw) * p.Ci + ci;
        }
        v = *reinterpret_cast<const float4*>(src);
        return ok;
    };
    auto ldB = [&](int s, int t) -> uint4 { return w_unit[s][(t < nslab ? t : nslab - 1) * w_slab]; };

    f32x16 acc[TM][TN];
    T::run(lds_x3, p.K, ldA, ldB, acc);
    gemm_epilogue<T, TM, TN>(p, m0, n0, acc);
}

template <int WM, int WN, int TM, int TN, int NP>
void launch_np(const GemmParams& p, hipStream_t s) {
    using T = GemmTileX3<WM, WN, TM, TN, NP>;
    static XpPerDeviceOnce attr_once;
    if (T::kLdsBytes > 48 * 1024 && attr_once.need()) {
        XP_HIP_WARN(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x3_kernel<WM, WN, TM, TN, 0, NP>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::kLdsBytes));
        XP_HIP_WARN(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x3_kernel<WM, WN, TM, TN, 1, NP>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::kLdsBytes));
    }
    dim3 grid(xp_cdiv(p.N, T::BN) * xp_cdiv(p.M, T::BM));
    static const bool by_shape = getenv("XP_PROF_SHAPES") != nullptr;
    std::string tag = std::string(p.mode ? "conv3x3_x3_mfma_" : "gemm_x3_mfma_") + std::to_string(T::BM) + "x" + std::to_string(T::BN);
    if (NP != 6) tag += "_np" + std::to_string(NP);
    if (by_shape) tag += "_M" + std::to_string(p.M) + "_N" + std::to_string(p.N) + "_K" + std::to_string(p.K) + (p.act == 1 ? "_gelu" : "");
    const double in_elems = p.mode == 0 ? (double)p.M * p.K : (double)p.M / (p.Ho * p.Wo) * p.Hi * p.Wi * p.Ci;
    // flops = algorithmic 2MNK (f32-equivalent); the matrix pipe executes NP x that in bf16
    XpProfScope prof(tag.c_str(), s, 2.0 * p.M * p.N * p.K,
                     4.0 * (in_elems + 1.5 * (double)p.N * p.K + (double)p.M * p.N * (p.res ? 2 : 1)));
    if (p.mode == 0) hipLaunchKernelGGL((gemm_x3_kernel<WM, WN, TM, TN, 0, NP>), grid, dim3(T::NT), T::kLdsBytes, s, p);
    else hipLaunchKernelGGL((gemm_x3_kernel<WM, WN, TM, TN, 1, NP>), grid, dim3(T::NT), T::kLdsBytes, s, p);
}

template <int WM, int WN, int TM, int TN>
void launch(const GemmParams& p, hipStream_t s) {
    switch (xp_dense_products_value()) {
        case 1: launch_np<WM, WN, TM, TN, 1>(p, s); break;
        case 3: launch_np<WM, WN, TM, TN, 3>(p, s); break;
        default: launch_np<WM, WN, TM, TN, 6>(p, s); break;
    }
}

int dispatch(const GemmParams& p, hipStream_t s) {
    const int N = p.N;
    static const int force = getenv("XP_X3_TILE") ? atoi(getenv("XP_X3_TILE")) : -1;   // tuning experiments only
    if (force >= 0) {
        switch (force) {
            case 0: launch<4, 1, 1, 1>(p, s); break;
            case 1: launch<4, 1, 1, 2>(p, s); break;
            case 2: launch<4, 1, 1, 3>(p, s); break;
            case 3: launch<2, 2, 1, 2>(p, s); break;
            default: launch<2, 2, 2, 2>(p, s); break;
        }
        XP_LAUNCH_CHECK();
        return XP_OK;
    }
    if (N <= 32) launch<4, 1, 1, 1>(p, s);                                   // 128 x 32
    else if (N <= 64) launch<4, 1, 1, 2>(p, s);                              // 128 x 64
    else if (N <= 96 || (N % 96 == 0 && (N / 96) % 4 != 0)) launch<4, 1, 1, 3>(p, s);   // 128 x 96  (N = 65..96, 192)
    else if (p.M <= 8192 && N >= 512 && (int64_t)xp_cdiv(p.M, 128) * xp_cdiv(N, 128) < 512)
        launch<2, 2, 1, 2>(p, s);                                            // 64 x 128: more blocks when 128 x 128 tiles would not fill the 2 x 256 slots once
    else launch<2, 2, 2, 2>(p, s);                                           // 128 x 128
    XP_LAUNCH_CHECK();
    return XP_OK;
}

}  // namespace

extern "C" size_t xp_split_weights_x3_bytes(int N, int K) {
    if (N <= 0 || K <= 0) return 0;
    return (size_t)N * ((K + X3_BK - 1) / X3_BK) * X3_SLAB_UNITS * 16;
}

extern "C" int xp_split_weights_x3(const float* W, void* out, int N, int K, void* stream) {
    XP_CHECK_ARG(W && out, "xp_split_weights_x3: null pointer");
    XP_CHECK_ARG(N > 0 && K > 0, "xp_split_weights_x3: bad shape %d %d", N, K);
    XP_CHECK_ARG(((uintptr_t)out & 15) == 0, "xp_split_weights_x3: out must be 16-byte aligned");
    const int nslab = (K + X3_BK - 1) / X3_BK;
    const int64_t n = (int64_t)N * nslab * 2;
    hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)xp_cdiv(n, (int64_t)256)), dim3(256), 0, (hipStream_t)stream,
                       W, (uint4*)out, N, K, nslab);
    XP_LAUNCH_CHECK();
    return XP_OK;
}

extern "C" int xp_gemm_nt_x3(const float* A, const void* Wx3, float* C, const float* bias, const float* scale,
                             const float* shift, const float* res, int M, int N, int K, int lda, int ldc, int ldres,
                             int act, void* stream) {
    XP_CHECK_ARG(A && Wx3 && C, "xp_gemm_nt_x3: null pointer");
    XP_CHECK_ARG(M > 0 && N > 0 && K > 0, "xp_gemm_nt_x3: bad shape %d %d %d", M, N, K);
    XP_CHECK_ARG(K % 4 == 0 && lda % 4 == 0, "xp_gemm_nt_x3: K and lda must be multiples of 4 (got %d, %d)", K, lda);
    XP_CHECK_ARG((scale == nullptr) == (shift == nullptr), "xp_gemm_nt_x3: scale and shift go together");
    XP_CHECK_ARG(act >= 0 && act <= 3, "xp_gemm_nt_x3: bad act %d", act);
    GemmParams p{};
    p.A = A; p.Wt = (const float*)Wx3; p.C = C; p.bias = bias; p.scale = scale; p.shift = shift; p.res = res;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldc = ldc; p.ldres = ldres; p.act = act; p.mode = 0;
    return dispatch(p, (hipStream_t)stream);
}

extern "C" int xp_conv3x3_nhwc_x3(const float* x, const void* Wx3, float* y, const float* bias, const float* scale,
                                  const float* shift, int batch, int Hi, int Wi, int Ci, int Co, int stride,
                                  int reflect_pad, int act, void* stream) {
    XP_CHECK_ARG(x && Wx3 && y, "xp_conv3x3_nhwc_x3: null pointer");
    XP_CHECK_ARG(Ci % 4 == 0, "xp_conv3x3_nhwc_x3: Ci must be a multiple of 4 (got %d)", Ci);
    XP_CHECK_ARG(stride == 1 || stride == 2, "xp_conv3x3_nhwc_x3: stride 1 or 2");
    XP_CHECK_ARG((scale == nullptr) == (shift == nullptr), "xp_conv3x3_nhwc_x3: scale and shift go together");
    XP_CHECK_ARG(!reflect_pad || (Hi >= 2 && Wi >= 2), "xp_conv3x3_nhwc_x3: reflection pad needs H,W >= 2");
    GemmParams p{};
    p.A = x; p.Wt = (const float*)Wx3; p.C = y; p.bias = bias; p.scale = scale; p.shift = shift; p.res = nullptr;
    p.Hi = Hi; p.Wi = Wi; p.Ci = Ci; p.stride = stride; p.reflect = reflect_pad;
    p.Ho = (Hi + 2 - 3) / stride + 1; p.Wo = (Wi + 2 - 3) / stride + 1;
    p.M = batch * p.Ho * p.Wo; p.N = Co; p.K = 9 * Ci; p.lda = 0; p.ldc = Co; p.ldres = 0; p.act = act; p.mode = 1;
    return dispatch(p, (hipStream_t)stream);
}
