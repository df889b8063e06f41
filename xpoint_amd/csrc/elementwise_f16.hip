// HBM-bound glue kernels of the fast mixed-precision class: the same operations as elementwise.hip with the tensors between them stored
// as fp16 — the reference's `mixed_precision: true` deployment (XPoint.py:182 autocast) keeps half tensors between its operations, so every store below IS
// one of the recipe's rounding points and nothing else is rounded.  Arithmetic in f32 on the (exact) half inputs, like torch's half kernels (opmath float).
//   stem      image (f32, cast to half by autocast's convolution) -> conv3x3 s2 (half) -> LayerNorm (half) -> GELU (half)      VMamba.py:1411-1416
//   layernorm half -> half, statistics in f32                                                                                     VMamba.py:1222-1234
//   dwconv    depthwise 3x3 (half out) -> SiLU (half out)                                                                         VMamba.py:655-658
//   depth_to_space half -> f32 (the `encoder_output` the API returns; values fp16-exact) + a half copy for the head convolution  VMamba.py:1500-1505
#include <stdlib.h>

#include <string>

#include "xp_common.h"
#include "../../include/xpoint_hip.h"

typedef _Float16 e16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 e16x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float e16_r(float v) { return (float)(_Float16)v; }

// ---- stem (1 -> CO channels), one thread per output pixel, output staged through LDS for coalesced 2-byte stores ----
template <int CO>
__global__ __launch_bounds__(64) void stem_f16_kernel(const float* __restrict__ img, const float* __restrict__ w9, const float* __restrict__ bias,
                                                      const float* __restrict__ lnw, const float* __restrict__ lnb, _Float16* __restrict__ y,
                                                      int B, int H, int W, float eps) {
    __shared__ float s_w[12 * CO];
    __shared__ _Float16 s_o[64 * (CO + 2)];
    for (int i = threadIdx.x; i < 9 * CO; i += 64) s_w[i] = w9[i];
    for (int i = threadIdx.x; i < CO; i += 64) { s_w[9 * CO + i] = bias[i]; s_w[10 * CO + i] = lnw[i]; s_w[11 * CO + i] = lnb[i]; }
    __syncthreads();
    const int Ho = H / 2 + (H & 1), Wo = W / 2 + (W & 1);
    const int64_t total = (int64_t)B * Ho * Wo;
    const int64_t p0 = (int64_t)blockIdx.x * 64, pix = p0 + threadIdx.x;
    if (pix < total) {
        const int ow = (int)(pix % Wo), oh = (int)((pix / Wo) % Ho);
        const int64_t b = pix / ((int64_t)Wo * Ho);
        float xin[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ih = oh * 2 + kh - 1, iw = ow * 2 + kw - 1;
                xin[kh * 3 + kw] = e16_r((ih >= 0 && ih < H && iw >= 0 && iw < W) ? img[(b * H + ih) * W + iw] : 0.f);
            }
        float acc[CO];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            float a = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) a = fmaf(xin[t], s_w[t * CO + c], a);
            a = e16_r(a + s_w[9 * CO + c]);
            acc[c] = a; s += a;
        }
        const float mean = s / (float)CO;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < CO; ++c) { const float d = acc[c] - mean; q = fmaf(d, d, q); }
        const float rstd = 1.f / sqrtf(q / (float)CO + eps);
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            const float v = e16_r((acc[c] - mean) * rstd * s_w[10 * CO + c] + s_w[11 * CO + c]);
            s_o[threadIdx.x * (CO + 2) + c] = (_Float16)xp_gelu_fast(v);
        }
    }
    __syncthreads();
    const int64_t nvalid = (total - p0 < 64) ? (total - p0) : 64;
    for (int i = threadIdx.x; i < nvalid * CO; i += 64) {
        const int pl = i / CO, c = i - pl * CO;
        y[(p0 + pl) * CO + c] = s_o[pl * (CO + 2) + c];
    }
}

// ---- LayerNorm: LPR lanes share a row, each lane owns NV chunks of 8 halves (16-byte accesses); two-pass mean / variance in f32 ----
template <int LPR, int NV>
__global__ __launch_bounds__(256) void layernorm_f16_kernel(const _Float16* __restrict__ x, _Float16* __restrict__ y, const float* __restrict__ w,
                                                            const float* __restrict__ b, int64_t M, int C, float eps) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, sub = lane % LPR;
    const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const int C8 = C >> 3;
    const bool rok = row < M;
    const e16x8* xr = reinterpret_cast<const e16x8*>(x + (rok ? row : 0) * C);
    float v[NV][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c8 = sub + i * LPR;
        e16x8 t = {};
        if (c8 < C8) t = xr[c8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[i][e] = (float)t[e]; }
        s += ((v[i][0] + v[i][1]) + (v[i][2] + v[i][3])) + ((v[i][4] + v[i][5]) + (v[i][6] + v[i][7]));
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (sub + i * LPR < C8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q = fmaf(d, d, q); }
        }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = 1.f / sqrtf(q / (float)C + eps);
    if (!rok) return;
    e16x8* yr = reinterpret_cast<e16x8*>(y + row * C);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c8 = sub + i * LPR;
        if (c8 < C8) {
            const float4 w0 = reinterpret_cast<const float4*>(w)[2 * c8], w1 = reinterpret_cast<const float4*>(w)[2 * c8 + 1];
            const float4 b0 = reinterpret_cast<const float4*>(b)[2 * c8], b1 = reinterpret_cast<const float4*>(b)[2 * c8 + 1];
            const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w}, bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            e16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (_Float16)((v[i][e] - mean) * rstd * wv[e] + bv[e]);
            yr[c8] = o;
        }
    }
}

// ---- depthwise 3x3 (zero pad 1, no bias) + SiLU: a 4 x 4 pixel block x 4 channels per thread (elementwise.hip's scheme), 8-byte accesses.
//      (An 8-channel / 16-byte form with packed-half operands and mixed-precision FMAs was built and measured SLOWER: 0.245 vs 0.219 ms per step at 172 registers;
//      the kernel is not load-width-bound.) ----
#ifndef XP_DWH_PW
#define XP_DWH_PW 4
#endif
#ifndef XP_DWH_PH
#define XP_DWH_PH 4
#endif
constexpr int DWH_PW = XP_DWH_PW, DWH_PH = XP_DWH_PH;      // (-D overrides: tools/dwconv16_dbg.sh)
template <bool F32COPY>
__global__ __launch_bounds__(256) void dwconv3x3_silu_f16_kernel(const _Float16* __restrict__ x, const float* __restrict__ w, _Float16* __restrict__ y,
                                                                 float* __restrict__ y32, int B, int H, int W, int C) {
    const int C4 = C >> 2;
    const int WG = (W + DWH_PW - 1) / DWH_PW, HG = (H + DWH_PH - 1) / DWH_PH;
    const int64_t total = (int64_t)B * HG * WG * C4;
    const unsigned nb = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, q = nb >> 3, r = nb & 7;
    const unsigned blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;        // an XCD takes a band of image rows (shared halo rows in its L2)
    const int64_t idx = (int64_t)blk * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c4 = (int)(idx % C4);
    const int64_t g = idx / C4;
    const int w0 = (int)(g % WG) * DWH_PW, h0 = (int)((g / WG) % HG) * DWH_PH;
    const int64_t b = g / ((int64_t)WG * HG);
    const e16x4* xv = reinterpret_cast<const e16x4*>(x);
    float4 wt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[t] = reinterpret_cast<const float4*>(w)[t * C4 + c4];
    float4 acc[DWH_PH][DWH_PW];
#pragma unroll
    for (int rr = 0; rr < DWH_PH; ++rr)
#pragma unroll
        for (int p = 0; p < DWH_PW; ++p) acc[rr][p] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ir = 0; ir < DWH_PH + 2; ++ir) {
        const int ih = h0 + ir - 1;
        if (ih < 0 || ih >= H) continue;
        float4 col[DWH_PW + 2];
#pragma unroll
        for (int cx = 0; cx < DWH_PW + 2; ++cx) {
            const int iw = w0 + cx - 1;
            e16x4 t = {};
            if (iw >= 0 && iw < W) t = xv[((b * H + ih) * W + iw) * C4 + c4];
            col[cx] = make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
        }
#pragma unroll
        for (int rr = 0; rr < DWH_PH; ++rr) {
            const int kh = ir - rr;
            if (kh < 0 || kh > 2) continue;
#pragma unroll
            for (int p = 0; p < DWH_PW; ++p)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int iw = w0 + p + kw - 1;
                    if (iw < 0 || iw >= W) continue;
                    const float4 xin = col[p + kw], wv = wt[kh * 3 + kw];
                    acc[rr][p].x = fmaf(xin.x, wv.x, acc[rr][p].x); acc[rr][p].y = fmaf(xin.y, wv.y, acc[rr][p].y);
                    acc[rr][p].z = fmaf(xin.z, wv.z, acc[rr][p].z); acc[rr][p].w = fmaf(xin.w, wv.w, acc[rr][p].w);
                }
        }
    }
#pragma unroll
    for (int rr = 0; rr < DWH_PH; ++rr) {
        if (h0 + rr >= H) break;
#pragma unroll
        for (int p = 0; p < DWH_PW; ++p) {
            if (w0 + p >= W) break;
            const float4 a = acc[rr][p];
            e16x4 o;
            o[0] = (_Float16)xp_silu(e16_r(a.x)); o[1] = (_Float16)xp_silu(e16_r(a.y)); o[2] = (_Float16)xp_silu(e16_r(a.z)); o[3] = (_Float16)xp_silu(e16_r(a.w));
            const int64_t oi = ((b * H + h0 + rr) * W + w0 + p) * C4 + c4;
            reinterpret_cast<e16x4*>(y)[oi] = o;
            if (F32COPY) reinterpret_cast<float4*>(y32)[oi] = make_float4((float)o[0], (float)o[1], (float)o[2], (float)o[3]);
        }
    }
}

// ---- depth_to_space(bs) of the half residual stream: f32 output (API) + half copy (head convolution operand) + range / finiteness status ----
__global__ __launch_bounds__(256) void depth_to_space_f16_kernel(const e16x4* __restrict__ x, float4* __restrict__ y32, e16x4* __restrict__ y16,
                                                                 int B, int H, int W, int C4, int bs, int* __restrict__ status) {
    const int Cq4 = C4 / (bs * bs);
    const unsigned total = (unsigned)B * H * W * C4;
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    e16x4 v = {};
    if (idx < total) v = x[idx];
    const float4 f = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    if (status) {
        const bool bad = !(fabsf(f.x) < INFINITY) || !(fabsf(f.y) < INFINITY) || !(fabsf(f.z) < INFINITY) || !(fabsf(f.w) < INFINITY);   // a half overflow is +-inf
        if (__ballot(bad && idx < total) != 0ull && (threadIdx.x & 63) == 0) atomicOr(status, XP_STATUS_ENC);
    }
    if (idx >= total) return;
    const unsigned ch = idx % (unsigned)C4, pix = idx / (unsigned)C4;
    const unsigned w = pix % (unsigned)W, hn = pix / (unsigned)W, h = hn % (unsigned)H, n = hn / (unsigned)H;
    const unsigned blk = ch / (unsigned)Cq4, c = ch - blk * Cq4;
    const unsigned i = blk / (unsigned)bs, j = blk - i * bs;
    const unsigned o = ((n * (H * bs) + (h * bs + i)) * (unsigned)(W * bs) + (w * bs + j)) * Cq4 + c;
    y32[o] = f;
    y16[o] = v;
}

}  // namespace

extern "C" int xp_stem_conv_ln_gelu_f16(const float* img, const float* w9co, const float* bias, const float* ln_w, const float* ln_b, void* y,
                                        int batch, int H, int W, int CO, float eps, void* stream) {
    XP_CHECK_ARG(img && w9co && bias && ln_w && ln_b && y, "xp_stem_conv_ln_gelu_f16: null pointer");
    const int Ho = H / 2 + (H & 1), Wo = W / 2 + (W & 1);
    const int64_t total = (int64_t)batch * Ho * Wo;
    XpProfScope prof("stem_conv_ln_gelu_f16", (hipStream_t)stream, 0.0, 4.0 * batch * H * W + 2.0 * total * CO);
    const dim3 grid((unsigned)((total + 63) / 64));
    _Float16* yh = reinterpret_cast<_Float16*>(y);
    if (CO == 48) hipLaunchKernelGGL(stem_f16_kernel<48>, grid, dim3(64), 0, (hipStream_t)stream, img, w9co, bias, ln_w, ln_b, yh, batch, H, W, eps);
    else if (CO == 16) hipLaunchKernelGGL(stem_f16_kernel<16>, grid, dim3(64), 0, (hipStream_t)stream, img, w9co, bias, ln_w, ln_b, yh, batch, H, W, eps);
    else { xp_set_error("xp_stem_conv_ln_gelu_f16: CO must be 48 or 16 (got %d)", CO); return XP_ERR_ARG; }
    XP_LAUNCH_CHECK();
    return XP_OK;
}

extern "C" int xp_layernorm_f16(const void* x, void* y, const float* w, const float* b, int64_t rows, int C, float eps, void* stream) {
    XP_CHECK_ARG(x && y && w && b, "xp_layernorm_f16: null pointer");
    XP_CHECK_ARG(C > 0 && C % 8 == 0 && C <= 1024, "xp_layernorm_f16: C must be a multiple of 8 in [8, 1024] (got %d)", C);
    XP_CHECK_ARG((((uintptr_t)x | (uintptr_t)y | (uintptr_t)w | (uintptr_t)b) & 15) == 0, "xp_layernorm_f16: buffers must be 16-byte aligned");
    if (rows == 0) return XP_OK;
    static const bool by_shape = getenv("XP_PROF_SHAPES") != nullptr;
    XpProfScope prof(by_shape ? ("layernorm_f16_C" + std::to_string(C)).c_str() : "layernorm_f16", (hipStream_t)stream, 8.0 * rows * C, 4.0 * rows * C);
    hipStream_t s = (hipStream_t)stream;
    const _Float16* xh = reinterpret_cast<const _Float16*>(x); _Float16* yh = reinterpret_cast<_Float16*>(y);
    const int C8 = C / 8;
#define XP_LNH(LPR, NV) hipLaunchKernelGGL((layernorm_f16_kernel<LPR, NV>), dim3(xp_cdiv(rows, 4 * (64 / LPR))), dim3(256), 0, s, xh, yh, w, b, rows, C, eps)
    if (C8 <= 4) XP_LNH(4, 1);
    else if (C8 <= 8) XP_LNH(8, 1);
    else if (C8 <= 16) XP_LNH(16, 1);
    else if (C8 <= 32) XP_LNH(32, 1);
    else if (C8 <= 64) XP_LNH(64, 1);
    else XP_LNH(64, 2);
#undef XP_LNH
    XP_LAUNCH_CHECK();
    return XP_OK;
}

extern "C" int xp_dwconv3x3_silu_f16(const void* x, const float* w9c, void* y, float* y_f32_copy, int batch, int H, int W, int C, void* stream) {
    XP_CHECK_ARG(x && w9c && y, "xp_dwconv3x3_silu_f16: null pointer");
    XP_CHECK_ARG(C % 4 == 0, "xp_dwconv3x3_silu_f16: C must be a multiple of 4 (got %d)", C);
    const int WG = (W + DWH_PW - 1) / DWH_PW, HG = (H + DWH_PH - 1) / DWH_PH;
    const int64_t total = (int64_t)batch * HG * WG * (C / 4);
    XpProfScope prof("dwconv3x3_silu_f16", (hipStream_t)stream, 0.0, (4.0 + (y_f32_copy ? 4.0 : 0.0)) * batch * H * W * C);
    const dim3 grid((unsigned)((total + 255) / 256));
    const _Float16* xh = reinterpret_cast<const _Float16*>(x); _Float16* yh = reinterpret_cast<_Float16*>(y);
    if (y_f32_copy) hipLaunchKernelGGL(dwconv3x3_silu_f16_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, xh, w9c, yh, y_f32_copy, batch, H, W, C);
    else hipLaunchKernelGGL(dwconv3x3_silu_f16_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, xh, w9c, yh, (float*)nullptr, batch, H, W, C);
    XP_LAUNCH_CHECK();
    return XP_OK;
}

extern "C" int xp_depth_to_space_nhwc_f16(const void* x, float* y_f32, void* y_f16, int batch, int H, int W, int C, int bs, int* status, void* stream) {
    XP_CHECK_ARG(x && y_f32 && y_f16, "xp_depth_to_space_nhwc_f16: null pointer");
    XP_CHECK_ARG(bs > 0 && C % (bs * bs) == 0 && (C / (bs * bs)) % 4 == 0, "xp_depth_to_space_nhwc_f16: C / bs^2 must be a multiple of 4");
    const int64_t n4 = (int64_t)batch * H * W * (C / 4);
    XP_CHECK_ARG(n4 < (1ll << 31), "xp_depth_to_space_nhwc_f16: tensor too large");
    XpProfScope prof("depth_to_space_f16", (hipStream_t)stream, 0.0, 8.0 * 4.0 * n4);
    hipLaunchKernelGGL(depth_to_space_f16_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const e16x4*>(x),
                       reinterpret_cast<float4*>(y_f32), reinterpret_cast<e16x4*>(y_f16), batch, H, W, C / 4, bs, status);
    XP_LAUNCH_CHECK();
    return XP_OK;
}
