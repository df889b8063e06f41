// fp32-grade GEMM / implicit-GEMM 3x3 convolution on the f16 matrix pipe: operands split two ways into fp16, three partial
// products (gemm_h2_core.h), the same fused epilogue and call sites as gemm_x3.hip / gemm.hip (reference VMamba.py:649,663
// in/out_proj; :110-128 Mlp; :605 x_proj; :1405-1440 strided 3x3 convs; XPoint.py:112-138 head convs).
//
//   C[m, n] = epilogue( 2^-k_n * sum_k A'[m, k] * (2^k_n W[n, k]) )     A' = A (row-major M x K, f32) or im2col(NHWC f32 image)
//   W is given pre-split: xp_split_weights_h2 turns the (N, K) f32 matrix into
//       Wh2[slab = k / 32][n][plane 0..1][32] fp16   (K zero-padded to a multiple of 32; row n scaled by 2^k_n so that its largest
//       element lies in [2^13, 2^14))   followed by the N factors 2^-k_n (f32)
//   once per weight upload: slab-major, so the BN rows a workgroup needs for one slab are BN * 128 contiguous bytes.
#include <stdlib.h>

#include <algorithm>
#include <string>

#include "gemm_h2_core.h"

bool xp_gemm_h2p_applies(const GemmParams& p);      // gemm_h2p.hip: ping-pong schedule of the 128 x 128 tile
int xp_gemm_h2p_launch(const GemmParams& p, hipStream_t s);

namespace {

// one wave per weight row: 2^-k_n with max|W[n, :]| * 2^k_n in [2^13, 2^14) (1 for an all-zero row)
__global__ __launch_bounds__(256) void h2_row_scale_kernel(const float* __restrict__ W, float* __restrict__ inv_scale, int N, int K) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const int lane = threadIdx.x & 63;
    float m = 0.f;
    for (int k = lane; k < K; k += 64) m = fmaxf(m, fabsf(W[(int64_t)n * K + k]));
    m = xp_wave_max(m);
    if (lane == 0) {
        int e = 0;
        if (m > 0.f && m < INFINITY) (void)frexpf(m, &e); else e = 14;      // m = f * 2^e, f in [0.5, 1)  ->  m * 2^(14 - e) in [2^13, 2^14)
        // a nearly dead row (largest |w| < 2^-100) is scaled like one at 2^-100: its elements then sit at or below fp16's smallest
        // values and contribute what they are worth, ~0; without the clamp 2^(e-14) underflows, 1 / inv_scale = inf and the row's
        // planes become inf / NaN (0 * inf) — poisoning every output column
        e = e < -100 ? -100 : e;
        inv_scale[n] = ldexpf(1.f, e - 14);
    }
}

__global__ void h2_split_weights_kernel(const float* __restrict__ W, const float* __restrict__ inv_scale, uint4* __restrict__ out, int N, int K,
                                        int nslab) {
    // one thread per (n, slab, octet): 8 floats -> 2 x 16 B
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (int64_t)N * nslab * 4) return;
    const int oct = (int)(id & 3);
    const int64_t ns = id >> 2;                       // n * nslab + slab
    const int n = (int)(ns / nslab), slab = (int)(ns - (int64_t)n * nslab);
    const int k = slab * H2_BK + oct * 8;
    const float sc = 1.f / inv_scale[n];              // both are powers of two: exact
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (k + j < K) ? W[(int64_t)n * K + k + j] * sc : 0.f;
    uint2 a0, a1, b0, b1;
    h2_split4(make_float4(v[0], v[1], v[2], v[3]), a0, a1);
    h2_split4(make_float4(v[4], v[5], v[6], v[7]), b0, b1);
    uint4* o = out + ((int64_t)slab * N + n) * H2_SLAB_UNITS + oct;
    o[0] = make_uint4(a0.x, a0.y, b0.x, b0.y);
    o[4] = make_uint4(a1.x, a1.y, b1.x, b1.y);
}

template <int WM, int WN, int TM, int TN, int MODE>
__global__ __launch_bounds__(WM * WN * 64) void gemm_h2_kernel(GemmParams p) {
    using T = GemmTileH2<WM, WN, TM, TN>;
    extern __shared__ __align__(16) unsigned char lds_h2[];
    // XCD-aware tile order (the bijective remap of gemm.hip: each XCD gets a contiguous run of logical tiles, the N-tiles of one
    // M-tile adjacent, so the A rows they share come from that XCD's L2)
    const int ntn = (p.N + T::BN - 1) / T::BN;
    const int total = gridDim.x;
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int q = total >> 3, r = total & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    // Wide layers (the weights do not fit an XCD's 4 MB L2: fc1 of the deep stages): column tiles in groups of p.ngroup — all row blocks of a
    // group before the next group — so that a group's weight slabs stay in L2 while the activations stream past (tools/gemm_traffic.sh:
    // 275 -> MB fetched per launch for M 4800, N 3072, K 768 with the plain N-fastest order: every row block pulled all of W again).
    int mt, nt;
    if (p.ngroup > 0) {
        const int ntm = (p.M + T::BM - 1) / T::BM;
        const int per = ntm * p.ngroup;
        const int g = logical / per, rem = logical - g * per;
        const int gw = min(p.ngroup, ntn - g * p.ngroup);
        mt = rem / gw; nt = g * p.ngroup + (rem - mt * gw);
    } else {
        mt = logical / ntn; nt = logical - mt * ntn;
    }
    const int m0 = mt * T::BM, n0 = nt * T::BN;

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
    const int nslab = (p.K + H2_BK - 1) / H2_BK;
    const uint4* w_unit[T::B_LD];
#pragma unroll
    for (int s = 0; s < T::B_LD; ++s) {
        const int n = n0 + T::b_row(s);
        w_unit[s] = reinterpret_cast<const uint4*>(p.Wt) + (int64_t)(n < p.N ? n : 0) * H2_SLAB_UNITS + T::b_unit(s);
    }
    const int kmax = p.K - 4;
    const int64_t w_slab = (int64_t)p.N * H2_SLAB_UNITS;     // 16-byte units per slab of the whole weight matrix
    auto ldA = [&](int s, int k, float4& v) -> bool {
        bool ok = k < p.K;
        const float* src;
        if (MODE == 0) {
            src = a_ptr[s] + (ok ? k : kmax);
        } else {
            int tap = a_tap[s], ci = a_ci[s];
            if (!ok) { tap = 8; ci = p.Ci - 4; }
            a_ci[s] += H2_BK;                              // Ci >= 4: at most eight wraps per 32-wide slab, as selects
#pragma unroll
            for (int w = 0; w < 8; ++w) { const bool wrap = a_ci[s] >= p.Ci; a_ci[s] -= wrap ? p.Ci : 0; a_tap[s] += wrap ? 1 : 0; }
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
    T::run(lds_h2, p.K, ldA, ldB, acc);
    gemm_epilogue<T, TM, TN, true>(p, m0, n0, acc);
}

// Row-stationary variant (GemmTileH2R): A fragments straight from global memory, only the weights in LDS.  K % 8 == 0 (conv: Ci % 8 == 0).
template <int TN, int MODE>
__global__ __launch_bounds__(256) void gemm_h2r_kernel(GemmParams p) {
    using T = GemmTileH2R<TN>;
    extern __shared__ __align__(16) unsigned char lds_h2[];
    const int ntn = (p.N + T::BN - 1) / T::BN;
    const int total = gridDim.x;
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int q = total >> 3, r = total & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    const int m0 = (logical / ntn) * T::BM, n0 = (logical % ntn) * T::BN;
    const int fh = (threadIdx.x & 63) >> 5;

    const int m = m0 + T::a_row();
    const int mc = m < p.M ? m : 0;              // rows past M only feed output rows that are never stored
    const float* a_ptr;
    int a_oh = 0, a_ow = 0, a_tap[2] = {0, 0}, a_ci[2] = {0, 0};
    if (MODE == 0) a_ptr = p.A + (int64_t)mc * p.lda;
    else {
        const int hw = p.Ho * p.Wo;
        const int b = mc / hw, rr = mc - b * hw;
        a_oh = (rr / p.Wo) * p.stride - 1; a_ow = (rr % p.Wo) * p.stride - 1;
        a_ptr = p.A + (int64_t)b * p.Hi * p.Wi * p.Ci;
#pragma unroll
        for (int s = 0; s < 2; ++s) { const int k = s * 16 + fh * 8; a_tap[s] = k / p.Ci; a_ci[s] = k - a_tap[s] * p.Ci; }
    }
    const int nslab = (p.K + H2_BK - 1) / H2_BK;
    const uint4* w_unit[T::B_LD];
#pragma unroll
    for (int s = 0; s < T::B_LD; ++s) {
        const int n = n0 + T::b_row(s);
        w_unit[s] = reinterpret_cast<const uint4*>(p.Wt) + (int64_t)(n < p.N ? n : 0) * H2_SLAB_UNITS + T::b_unit(s);
    }
    const int64_t w_slab = (int64_t)p.N * H2_SLAB_UNITS;
    int which = 0;         // the engine asks for k-step 0, then k-step 1 of every slab, slabs in order
    auto ldA8 = [&](int k, float4& lo, float4& hi) -> bool {
        bool ok = k < p.K;
        const float* src;
        if (MODE == 0) {
            src = a_ptr + (ok ? k : p.K - 8);
        } else {
            const int s = which; which ^= 1;
            int tap = a_tap[s], ci = a_ci[s];
            if (!ok) { tap = 8; ci = p.Ci - 8; }
            a_ci[s] += H2_BK;                              // Ci >= 8: at most four wraps per 32-wide slab, as selects
#pragma unroll
            for (int w = 0; w < 4; ++w) { const bool wrap = a_ci[s] >= p.Ci; a_ci[s] -= wrap ? p.Ci : 0; a_tap[s] += wrap ? 1 : 0; }
            int ih = a_oh + tap / 3, iw = a_ow + tap % 3;
            if (p.reflect) {
                ih = ih < 0 ? -ih : (ih >= p.Hi ? 2 * p.Hi - 2 - ih : ih);
                iw = iw < 0 ? -iw : (iw >= p.Wi ? 2 * p.Wi - 2 - iw : iw);
            } else {
                ok = ok && ih >= 0 && ih < p.Hi && iw >= 0 && iw < p.Wi;
                ih = ih < 0 ? 0 : (ih >= p.Hi ? p.Hi - 1 : ih);
                iw = iw < 0 ? 0 : (iw >= p.Wi ? p.Wi - 1 : iw);
            }
            src = a_ptr + ((int64_t)ih * p.Wi + iw) * p.Ci + ci;
        }
        lo = *reinterpret_cast<const float4*>(src);
        hi = *reinterpret_cast<const float4*>(src + 4);
        return ok;
    };
    auto ldB = [&](int s, int t) -> uint4 { return w_unit[s][(t < nslab ? t : nslab - 1) * w_slab]; };

    f32x16 acc[1][TN];
    T::run(lds_h2, p.K, ldA8, ldB, acc);
    gemm_epilogue<T, 1, TN, true>(p, m0, n0, acc);
}

template <int TN>
void launch_r(const GemmParams& p, hipStream_t s) {
    using T = GemmTileH2R<TN>;
    dim3 grid(xp_cdiv(p.N, T::BN) * xp_cdiv(p.M, T::BM));
    static const bool by_shape = getenv("XP_PROF_SHAPES") != nullptr;
    std::string tag = std::string(p.mode ? "conv3x3_h2r_mfma_" : "gemm_h2r_mfma_") + std::to_string(T::BM) + "x" + std::to_string(T::BN);
    if (by_shape) tag += "_M" + std::to_string(p.M) + "_N" + std::to_string(p.N) + "_K" + std::to_string(p.K) + (p.act == 1 ? "_gelu" : "");
    const double in_elems = p.mode == 0 ? (double)p.M * p.K : (double)p.M / (p.Ho * p.Wo) * p.Hi * p.Wi * p.Ci;
    XpProfScope prof(tag.c_str(), s, 2.0 * p.M * p.N * p.K, 4.0 * (in_elems + (double)p.N * p.K + (double)p.M * p.N * (p.res ? 2 : 1)));
    if (p.mode == 0) hipLaunchKernelGGL((gemm_h2r_kernel<TN, 0>), grid, dim3(T::NT), T::kLdsBytes, s, p);
    else hipLaunchKernelGGL((gemm_h2r_kernel<TN, 1>), grid, dim3(T::NT), T::kLdsBytes, s, p);
}

template <int WM, int WN, int TM, int TN>
void launch(const GemmParams& p, hipStream_t s) {
    using T = GemmTileH2<WM, WN, TM, TN>;
    static XpPerDeviceOnce attr_once;
    if (T::kLdsBytes > 48 * 1024 && attr_once.need()) {
        XP_HIP_WARN(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h2_kernel<WM, WN, TM, TN, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::kLdsBytes));
        XP_HIP_WARN(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h2_kernel<WM, WN, TM, TN, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::kLdsBytes));
    }
    dim3 grid(xp_cdiv(p.N, T::BN) * xp_cdiv(p.M, T::BM));
    static const bool by_shape = getenv("XP_PROF_SHAPES") != nullptr;
    std::string tag = std::string(p.mode ? "conv3x3_h2_mfma_" : "gemm_h2_mfma_") + std::to_string(T::BM) + "x" + std::to_string(T::BN);
    if (by_shape) tag += "_M" + std::to_string(p.M) + "_N" + std::to_string(p.N) + "_K" + std::to_string(p.K) + (p.act == 1 ? "_gelu" : "");
    const double in_elems = p.mode == 0 ? (double)p.M * p.K : (double)p.M / (p.Ho * p.Wo) * p.Hi * p.Wi * p.Ci;
    // flops = algorithmic 2MNK (f32-equivalent); the matrix pipe executes 3 x that in fp16
    XpProfScope prof(tag.c_str(), s, 2.0 * p.M * p.N * p.K, 4.0 * (in_elems + (double)p.N * p.K + (double)p.M * p.N * (p.res ? 2 : 1)));
    if (p.mode == 0) hipLaunchKernelGGL((gemm_h2_kernel<WM, WN, TM, TN, 0>), grid, dim3(T::NT), T::kLdsBytes, s, p);
    else hipLaunchKernelGGL((gemm_h2_kernel<WM, WN, TM, TN, 1>), grid, dim3(T::NT), T::kLdsBytes, s, p);
}

int dispatch(const GemmParams& p_in, hipStream_t s) {
    GemmParams p = p_in;
    const int N = p.N;
    // column-tile groups (tile engine): when W (two fp16 planes = 4 bytes per element) exceeds what an XCD's L2 can keep next to the activations,
    // process the column tiles in groups whose weight slabs take ~2 MB (XP_H2_NGROUP overrides; 0 = plain N-fastest order)
    {
        static const int force_g = getenv("XP_H2_NGROUP") ? atoi(getenv("XP_H2_NGROUP")) : -1;
        const int ntn128 = xp_cdiv(N, 128);
        int g = 0;
        if (p.mode == 0 && (int64_t)N * p.K * 4 > (2 << 20) && ntn128 > 2 && p.K <= 1024) {      // long K with few columns: the activations are the big operand — N-fastest shares them
            const int gmax = (int)std::max<int64_t>(1, (int64_t)(2 << 20) / ((int64_t)128 * p.K * 4));
            const int ngroups = xp_cdiv(ntn128, gmax);
            g = ngroups > 1 ? xp_cdiv(ntn128, ngroups) : 0;                                        // balanced group widths
        }
        p.ngroup = force_g >= 0 ? force_g : g;
    }
    static const int force = getenv("XP_H2_TILE") ? atoi(getenv("XP_H2_TILE")) : -1;   // tuning experiments only
    // 64 x 128 tiles (twice the workgroups) only when there are fewer than 128 tiles of 128 x 128 — i.e. at small batches, where the launch would leave most
    // CUs idle.  Rounds 2-3 used "< 256, or < 512 for wide N at M <= 8192", tuned on stand-alone launches; in the overlapped step the other streams fill the idle
    // CUs and the 128 x 128 tile's lower cost per FLOP wins: +0.5 to +1.5 % at 8 pairs, equal at 1 - 2 pairs (XP_H2_NO64 = 0 old rule, 1 never, 2 this rule).
    // Every tile and both engines walk K in the same order: the choice never changes a result bit (tools checked by CRC), so it MAY depend on M.
    static const int no64 = getenv("XP_H2_NO64") ? atoi(getenv("XP_H2_NO64")) : 2;
    const int64_t t128 = (int64_t)xp_cdiv(p.M, 128) * xp_cdiv(N, 128);
    const bool want64 = no64 == 1 ? false : no64 == 2 ? t128 < 128 : ((p.M <= 8192 && N >= 512 && t128 < 512) || t128 < 256);
    const int sel = force >= 0 ? force
                  : N <= 32 ? 0 : N <= 64 ? 1 : (N <= 96 || (N % 96 == 0 && (N / 96) % 4 != 0)) ? 2
                  : want64 ? 3 : 4;       // fewer 128 x 128 tiles than CUs (x_proj of the deep stages: N = 104 / 200): 64 x 128
    // Row-stationary engine for the implicit-GEMM convolutions (measured: 0.55 vs 0.71 ms for the four big convs of a step; the gathered
    // A rows cost the tile engine an LDS round trip they do not need), the tile engine for plain GEMMs (equal at K >= 384, 15 % faster at
    // K = 96 where the row-stationary lane-per-row loads touch 32 cache lines per instruction).  XP_H2_ENGINE = rs | lds forces one (A/B).
    static const char* eng = getenv("XP_H2_ENGINE");
    const bool want_rs = eng ? (eng[0] == 'r') : p.mode == 1;
    const bool rs_ok = want_rs && p.K % 8 == 0 && (p.mode == 0 ? p.lda % 4 == 0 : p.Ci % 8 == 0) && sel != 3;
    if (rs_ok) {
        switch (sel) {
            case 0: launch_r<1>(p, s); break;
            case 1: launch_r<2>(p, s); break;
            case 2: launch_r<3>(p, s); break;
            default: launch_r<4>(p, s); break;
        }
        XP_LAUNCH_CHECK();
        return XP_OK;
    }
    if (sel >= 3 && xp_gemm_h2p_applies(p)) return xp_gemm_h2p_launch(p, s);
    switch (sel) {
        case 0: launch<4, 1, 1, 1>(p, s); break;       // 128 x 32
        case 1: launch<4, 1, 1, 2>(p, s); break;       // 128 x 64
        case 2: launch<4, 1, 1, 3>(p, s); break;       // 128 x 96  (N = 65..96, 192)
        case 3: launch<2, 2, 1, 2>(p, s); break;       // 64 x 128: more blocks when 128 x 128 tiles would not fill the 2 x 256 slots once
        default: launch<2, 2, 2, 2>(p, s); break;      // 128 x 128
    }
    XP_LAUNCH_CHECK();
    return XP_OK;
}

}  // namespace

extern "C" size_t xp_split_weights_h2_bytes(int N, int K) {
    if (N <= 0 || K <= 0) return 0;
    return (size_t)N * ((K + H2_BK - 1) / H2_BK) * H2_SLAB_UNITS * 16 + (((size_t)N * 4 + 15) & ~(size_t)15);
}

// planes first, the N inverse row scales (f32) right behind them
static const float* h2_scales(const void* Wh2, int N, int K) {
    return reinterpret_cast<const float*>(reinterpret_cast<const char*>(Wh2) + (size_t)N * ((K + H2_BK - 1) / H2_BK) * H2_SLAB_UNITS * 16);
}

extern "C" int xp_split_weights_h2(const float* W, void* out, int N, int K, void* stream) {
    XP_CHECK_ARG(W && out, "xp_split_weights_h2: null pointer");
    XP_CHECK_ARG(N > 0 && K > 0, "xp_split_weights_h2: bad shape %d %d", N, K);
    XP_CHECK_ARG(((uintptr_t)out & 15) == 0, "xp_split_weights_h2: out must be 16-byte aligned");
    const int nslab = (K + H2_BK - 1) / H2_BK;
    float* inv = const_cast<float*>(h2_scales(out, N, K));
    hipLaunchKernelGGL(h2_row_scale_kernel, dim3(xp_cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, W, inv, N, K);
    const int64_t n = (int64_t)N * nslab * 4;
    hipLaunchKernelGGL(h2_split_weights_kernel, dim3((unsigned)xp_cdiv(n, (int64_t)256)), dim3(256), 0, (hipStream_t)stream, W, inv, (uint4*)out, N, K, nslab);
    XP_LAUNCH_CHECK();
    return XP_OK;
}

extern "C" int xp_gemm_nt_h2(const float* A, const void* Wh2, float* C, const float* bias, const float* scale, const float* shift,
                             const float* res, int M, int N, int K, int lda, int ldc, int ldres, int act, void* stream) {
    XP_CHECK_ARG(A && Wh2 && C, "xp_gemm_nt_h2: null pointer");
    XP_CHECK_ARG(M > 0 && N > 0 && K > 0, "xp_gemm_nt_h2: bad shape %d %d %d", M, N, K);
    XP_CHECK_ARG(K % 4 == 0 && lda % 4 == 0, "xp_gemm_nt_h2: K and lda must be multiples of 4 (got %d, %d)", K, lda);
    XP_CHECK_ARG((scale == nullptr) == (shift == nullptr), "xp_gemm_nt_h2: scale and shift go together");
    XP_CHECK_ARG(act >= 0 && act <= 3, "xp_gemm_nt_h2: bad act %d", act);
    GemmParams p{};
    p.A = A; p.Wt = (const float*)Wh2; p.C = C; p.bias = bias; p.scale = scale; p.shift = shift; p.res = res; p.wscale = h2_scales(Wh2, N, K);
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldc = ldc; p.ldres = ldres; p.act = act; p.mode = 0;
    p.r16 = xp_amp_value();
    return dispatch(p, (hipStream_t)stream);
}

extern "C" int xp_conv3x3_nhwc_h2(const float* x, const void* Wh2, float* y, const float* bias, const float* scale, const float* shift,
                                  int batch, int Hi, int Wi, int Ci, int Co, int stride, int reflect_pad, int act, void* stream) {
    XP_CHECK_ARG(x && Wh2 && y, "xp_conv3x3_nhwc_h2: null pointer");
    XP_CHECK_ARG(Ci % 4 == 0, "xp_conv3x3_nhwc_h2: Ci must be a multiple of 4 (got %d)", Ci);
    XP_CHECK_ARG(stride == 1 || stride == 2, "xp_conv3x3_nhwc_h2: stride 1 or 2");
    XP_CHECK_ARG((scale == nullptr) == (shift == nullptr), "xp_conv3x3_nhwc_h2: scale and shift go together");
    XP_CHECK_ARG(!reflect_pad || (Hi >= 2 && Wi >= 2), "xp_conv3x3_nhwc_h2: reflection pad needs H,W >= 2");
    GemmParams p{};
    p.A = x; p.Wt = (const float*)Wh2; p.C = y; p.bias = bias; p.scale = scale; p.shift = shift; p.res = nullptr; p.wscale = h2_scales(Wh2, Co, 9 * Ci);
    p.Hi = Hi; p.Wi = Wi; p.Ci = Ci; p.stride = stride; p.reflect = reflect_pad;
    p.Ho = (Hi + 2 - 3) / stride + 1; p.Wo = (Wi + 2 - 3) / stride + 1;
    p.M = batch * p.Ho * p.Wo; p.N = Co; p.K = 9 * Ci; p.lda = 0; p.ldc = Co; p.ldres = 0; p.act = act; p.mode = 1;
    p.r16 = xp_amp_value();
    return dispatch(p, (hipStream_t)stream);
}
