// HBM-bound glue kernels of the encoder and heads, NHWC layout (channels contiguous):
// LayerNorm, depthwise 3x3 + SiLU, the 1-channel stem convolution (+LN+GELU), depth-to-space,
// detector softmax + pixel shuffle, descriptor L2 normalisation, NHWC->NCHW export.
#include <stdlib.h>

#include <string>

#include <algorithm>

#include "xp_common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// LayerNorm over the last dim (C <= 1024), one wave per row, two-pass mean/variance in registers.
// Reference: nn.LayerNorm eps 1e-5, biased variance (VMamba.py:1222-1234, :1405-1440).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        int64_t M, int C, float eps, int gelu) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + row * C;
    float v[16];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + 64 * i;
        v[i] = (c < C) ? xr[c] : 0.f;
        s += v[i];
    }
    const float mean = xp_wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + 64 * i;
        const float d = (c < C) ? v[i] - mean : 0.f;
        q = fmaf(d, d, q);
    }
    const float rstd = 1.f / sqrtf(xp_wave_sum(q) / (float)C + eps);
    float* yr = y + row * C;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + 64 * i;
        if (c < C) {
            float o = (v[i] - mean) * rstd * w[c] + b[c];
            if (gelu) o = xp_gelu(o);
            yr[c] = o;
        }
    }
}

// Vectorised form for C % 4 == 0: LPR lanes (a power of two) share a row, each lane owns NV float4s, so a wave handles
// 64/LPR rows per pass with 16-byte accesses (C = 96: two rows per wave, 24 of 32 lanes active, instead of one row on
// 24 scalar lanes).  Same two-pass mean / variance.
// P32: y is the two-plane fp16 image [row][c / 32][plane][32] of the result (ring_core.h: the operand format of xp_gemm_nt_h2s) instead of f32 rows.
template <int LPR, int NV, int RPG = 1, bool P32 = false>
__global__ __launch_bounds__(256) void layernorm_vec_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            int64_t M, int C, float eps, int gelu) {
    // RPG rows per lane group and pass (rows r, r + RPW, ...): RPG x NV independent 16-byte loads in flight per lane; every row's arithmetic — the lane
    // partial, the xor butterfly, the normalisation — is exactly the RPG = 1 sequence, so the results are bit-identical
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, sub = lane % LPR;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (RPW * RPG) + lane / LPR;
    const int C4 = C >> 2;
    float4 v[RPG][NV];
    bool rok[RPG];
#pragma unroll
    for (int r = 0; r < RPG; ++r) {
        const int64_t row = row0 + r * RPW;
        rok[r] = row < M;
        const float4* xr = reinterpret_cast<const float4*>(x + (rok[r] ? row : 0) * C);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c4 = sub + i * LPR;
            v[r][i] = (c4 < C4) ? xr[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
#pragma unroll
    for (int r = 0; r < RPG; ++r) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) s += (v[r][i].x + v[r][i].y) + (v[r][i].z + v[r][i].w);
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mean = s / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (sub + i * LPR < C4) {
                const float dx = v[r][i].x - mean, dy = v[r][i].y - mean, dz = v[r][i].z - mean, dw = v[r][i].w - mean;
                q = fmaf(dx, dx, q); q = fmaf(dy, dy, q); q = fmaf(dz, dz, q); q = fmaf(dw, dw, q);
            }
        }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        const float rstd = 1.f / sqrtf(q / (float)C + eps);
        if (!rok[r]) continue;
        float4* yr = reinterpret_cast<float4*>(y + (row0 + r * RPW) * C);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c4 = sub + i * LPR;
            if (c4 < C4) {
                const float4 wv = reinterpret_cast<const float4*>(w)[c4], bv = reinterpret_cast<const float4*>(b)[c4];
                float4 o;
                o.x = (v[r][i].x - mean) * rstd * wv.x + bv.x; o.y = (v[r][i].y - mean) * rstd * wv.y + bv.y;
                o.z = (v[r][i].z - mean) * rstd * wv.z + bv.z; o.w = (v[r][i].w - mean) * rstd * wv.w + bv.w;
                if (gelu) { o.x = xp_gelu(o.x); o.y = xp_gelu(o.y); o.z = xp_gelu(o.z); o.w = xp_gelu(o.w); }
                if (P32) {
                    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                    const h4 hi = {(_Float16)o.x, (_Float16)o.y, (_Float16)o.z, (_Float16)o.w};
                    const h4 lo = {(_Float16)(o.x - (float)hi[0]), (_Float16)(o.y - (float)hi[1]), (_Float16)(o.z - (float)hi[2]), (_Float16)(o.w - (float)hi[3])};
                    const int c = c4 * 4;
                    unsigned char* op = reinterpret_cast<unsigned char*>(y) + ((row0 + r * RPW) * (C >> 5) + (c >> 5)) * 128 + (c & 31) * 2;
                    *reinterpret_cast<h4*>(op) = hi;
                    *reinterpret_cast<h4*>(op + 64) = lo;
                } else
                yr[c4] = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Depthwise 3x3 (zero pad 1, no bias) + SiLU, NHWC.  Reference VMamba.py:655-658.
// weight layout [9][C] (tap-major) so that a lane's 4 channels are one 16-B load.
// ---------------------------------------------------------------------------------------------
// Each thread produces a PH x PW = 4 x 4 block of pixels x 4 channels, streaming the PH + 2 input rows of its 6-column window
// once (36 float4 loads per 16 outputs; a thread per output row re-reads two of its three rows: 72).  0.285 -> 0.244 ms per step.  Every output still
// accumulates its taps in (kh, kw) order with padded taps skipped — input rows arrive in ascending order, and an input row ih
// is tap kh = ih - oh + 1 of output row oh — so results are bit-identical to the one-row form.
#ifndef XP_DW_PW
#define XP_DW_PW 4
#endif
#ifndef XP_DW_PH
#define XP_DW_PH 4
#endif
constexpr int DW_PW = XP_DW_PW, DW_PH = XP_DW_PH;      // (-D overrides: tools/instep_ab.sh)
__device__ __forceinline__ float ew_r16(float v) { return (float)(_Float16)v; }
// AMP (xp_set_amp_mode): the convolution's output and the SiLU's output are half tensors under autocast: both rounded to fp16
template <bool AMP>
__global__ __launch_bounds__(256) void dwconv3x3_silu_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             float* __restrict__ y, int B, int H, int W, int C) {
    const int C4 = C >> 2;
    const int WG = (W + DW_PW - 1) / DW_PW, HG = (H + DW_PH - 1) / DW_PH;
    const int64_t total = (int64_t)B * HG * WG * C4;
    // XCD-aware block order (workgroup ids go round-robin to the 8 XCDs, one L2 each): every XCD takes one contiguous run of blocks = a band of
    // image rows, so the halo rows that vertically adjacent blocks share are fetched once per band, not once per block (144 -> MB per launch mix)
    const unsigned nb = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, q = nb >> 3, r = nb & 7;
    const unsigned blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    const int64_t idx = (int64_t)blk * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c4 = (int)(idx % C4);
    const int64_t g = idx / C4;
    const int w0 = (int)(g % WG) * DW_PW;
    const int h0 = (int)((g / WG) % HG) * DW_PH;
    const int64_t b = g / ((int64_t)WG * HG);
    const float4* xv = reinterpret_cast<const float4*>(x);
    float4 wt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[t] = reinterpret_cast<const float4*>(w)[t * C4 + c4];
    float4 acc[DW_PH][DW_PW];
#pragma unroll
    for (int r = 0; r < DW_PH; ++r)
#pragma unroll
        for (int p = 0; p < DW_PW; ++p) acc[r][p] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ir = 0; ir < DW_PH + 2; ++ir) {
        const int ih = h0 + ir - 1;
        if (ih < 0 || ih >= H) continue;          // zero padding: a skipped row adds nothing (same sum order for the rest)
        float4 col[DW_PW + 2];
#pragma unroll
        for (int cx = 0; cx < DW_PW + 2; ++cx) {
            const int iw = w0 + cx - 1;
            col[cx] = (iw >= 0 && iw < W) ? xv[((b * H + ih) * W + iw) * C4 + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int r = 0; r < DW_PH; ++r) {
            const int kh = ir - r;                // this input row is tap kh of output row h0 + r
            if (kh < 0 || kh > 2) continue;
#pragma unroll
            for (int p = 0; p < DW_PW; ++p)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int iw = w0 + p + kw - 1;
                    if (iw < 0 || iw >= W) continue;   // keep the reference's accumulation order: padded taps are skipped, not added as 0
                    const float4 xin = col[p + kw], wv = wt[kh * 3 + kw];
                    acc[r][p].x = fmaf(xin.x, wv.x, acc[r][p].x); acc[r][p].y = fmaf(xin.y, wv.y, acc[r][p].y);
                    acc[r][p].z = fmaf(xin.z, wv.z, acc[r][p].z); acc[r][p].w = fmaf(xin.w, wv.w, acc[r][p].w);
                }
        }
    }
#pragma unroll
    for (int r = 0; r < DW_PH; ++r) {
        if (h0 + r >= H) break;
#pragma unroll
        for (int p = 0; p < DW_PW; ++p) {
            if (w0 + p >= W) break;
            float4 o = acc[r][p];
            if (AMP) { o.x = ew_r16(o.x); o.y = ew_r16(o.y); o.z = ew_r16(o.z); o.w = ew_r16(o.w); }
            o.x = xp_silu(o.x); o.y = xp_silu(o.y); o.z = xp_silu(o.z); o.w = xp_silu(o.w);
            if (AMP) { o.x = ew_r16(o.x); o.y = ew_r16(o.y); o.z = ew_r16(o.z); o.w = ew_r16(o.w); }
            reinterpret_cast<float4*>(y)[((b * H + h0 + r) * W + w0 + p) * C4 + c4] = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Stem: gray image (B,1,H,W) -> conv3x3 stride 2 pad 1 (1 -> CO channels; the reference replicates gray
// to 3 channels, VMamba.py:1509-1510, so the 3 input-channel weights are pre-summed on the host)
// + bias + LayerNorm(CO) + GELU -> NHWC (B,H/2,W/2,CO).  Reference VMamba.py:1411-1416.
// One thread per output pixel, CO accumulators in registers; output staged through LDS for coalescing.
// ---------------------------------------------------------------------------------------------
// AMP: the image is cast to half by autocast's convolution, and conv, LayerNorm and GELU each return a half tensor
template <int CO, bool AMP>
__global__ __launch_bounds__(64) void stem_conv_ln_gelu_kernel(const float* __restrict__ img, const float* __restrict__ w9,
                                                               const float* __restrict__ bias, const float* __restrict__ lnw,
                                                               const float* __restrict__ lnb, float* __restrict__ y,
                                                               int B, int H, int W, float eps) {
    __shared__ float s_w[9 * CO + 3 * CO];
    __shared__ float s_o[64 * (CO + 1)];
    for (int i = threadIdx.x; i < 9 * CO; i += 64) s_w[i] = w9[i];          // [tap][CO]
    for (int i = threadIdx.x; i < CO; i += 64) { s_w[9 * CO + i] = bias[i]; s_w[10 * CO + i] = lnw[i]; s_w[11 * CO + i] = lnb[i]; }
    __syncthreads();
    const int Ho = H / 2 + (H & 1), Wo = W / 2 + (W & 1);   // floor((H+2-3)/2)+1
    const int64_t total = (int64_t)B * Ho * Wo;
    const int64_t p0 = (int64_t)blockIdx.x * 64;
    const int64_t pix = p0 + threadIdx.x;
    if (pix < total) {
        const int ow = (int)(pix % Wo), oh = (int)((pix / Wo) % Ho);
        const int64_t b = pix / ((int64_t)Wo * Ho);
        float xin[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ih = oh * 2 + kh - 1, iw = ow * 2 + kw - 1;
                xin[kh * 3 + kw] = (ih >= 0 && ih < H && iw >= 0 && iw < W) ? img[(b * H + ih) * W + iw] : 0.f;
                if (AMP) xin[kh * 3 + kw] = ew_r16(xin[kh * 3 + kw]);
            }
        float acc[CO];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            float a = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) a = fmaf(xin[t], s_w[t * CO + c], a);
            a += s_w[9 * CO + c];
            if (AMP) a = ew_r16(a);
            acc[c] = a; s += a;
        }
        const float mean = s / (float)CO;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < CO; ++c) { const float d = acc[c] - mean; q = fmaf(d, d, q); }
        const float rstd = 1.f / sqrtf(q / (float)CO + eps);
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            float v = (acc[c] - mean) * rstd * s_w[10 * CO + c] + s_w[11 * CO + c];
            if (AMP) v = ew_r16(v);
            v = xp_gelu_fast(v);
            if (AMP) v = ew_r16(v);
            s_o[threadIdx.x * (CO + 1) + c] = v;
        }
    }
    __syncthreads();
    const int64_t nvalid = (total - p0 < 64) ? (total - p0) : 64;
    for (int i = threadIdx.x; i < nvalid * CO; i += 64) {
        const int pl = i / CO, c = i - pl * CO;
        y[(p0 + pl) * CO + c] = s_o[pl * (CO + 1) + c];
    }
}

// ---------------------------------------------------------------------------------------------
// depth_to_space(4) in NHWC: out[n, bs*h+i, bs*w+j, c] = x[n, h, w, (bs*i+j)*Cq + c]  (VMamba.py:1500-1505;
// block-major channel order, NOT PixelShuffle order).
// ---------------------------------------------------------------------------------------------
// status != NULL: bit XP_STATUS_ENC is OR-ed into *status when an element is non-finite or |x| >= limit (one atomic per wave that saw one).
__global__ __launch_bounds__(256) void depth_to_space_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             int B, int H, int W, int C, int bs, float limit, int* __restrict__ status) {
    const int Cq = C / (bs * bs);
    const int64_t total = (int64_t)B * H * W * C;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (status) {
        const bool bad = idx < total && !(fabsf(x[idx]) < limit);        // NaN compares false
        if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(status, XP_STATUS_ENC);
    }
    if (idx >= total) return;
    const int ch = (int)(idx % C);
    const int64_t pix = idx / C;
    const int w = (int)(pix % W), h = (int)((pix / W) % H);
    const int64_t n = pix / ((int64_t)W * H);
    const int blk = ch / Cq, c = ch - blk * Cq;
    const int i = blk / bs, j = blk - i * bs;
    y[((n * (H * bs) + (h * bs + i)) * (int64_t)(W * bs) + (w * bs + j)) * Cq + c] = x[idx];
}

// float4 form (Cq % 4 == 0, 16-byte aligned buffers, < 2^31 elements): a thread moves four channels of one block — 32-bit index arithmetic, 16-byte accesses
__global__ __launch_bounds__(256) void depth_to_space_vec_kernel(const float4* __restrict__ x, float4* __restrict__ y,
                                                                 int B, int H, int W, int C4, int bs, float limit, int* __restrict__ status) {
    const int Cq4 = C4 / (bs * bs);
    const unsigned total = (unsigned)B * H * W * C4;
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    const float4 v = idx < total ? x[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (status) {
        const bool bad = !(fabsf(v.x) < limit) || !(fabsf(v.y) < limit) || !(fabsf(v.z) < limit) || !(fabsf(v.w) < limit);        // NaN compares false
        if (__ballot(bad && idx < total) != 0ull && (threadIdx.x & 63) == 0) atomicOr(status, XP_STATUS_ENC);
    }
    if (idx >= total) return;
    const unsigned ch = idx % (unsigned)C4, pix = idx / (unsigned)C4;
    const unsigned w = pix % (unsigned)W, hn = pix / (unsigned)W, h = hn % (unsigned)H, n = hn / (unsigned)H;
    const unsigned blk = ch / (unsigned)Cq4, c = ch - blk * Cq4;
    const unsigned i = blk / (unsigned)bs, j = blk - i * bs;
    y[((n * (H * bs) + (h * bs + i)) * (unsigned)(W * bs) + (w * bs + j)) * Cq4 + c] = v;
}

// ---------------------------------------------------------------------------------------------
// Detector tail: softmax over the 65 logits of a cell, drop the dustbin, PixelShuffle(r):
// prob[b, r*h+i, r*w+j] = p[b, h, w, r*i+j]   (XPoint.py:356-358).  One wave per cell.
// mode 1 = SuperPointMagicLeap heat-map: exp(x)/(sum+1e-5), no max subtraction (SuperPointMagicLeap.py:73-74).
// ---------------------------------------------------------------------------------------------
constexpr int SMX_CPW = 4;       // cells per wave: the loads of all four are issued first, the four (independent) reductions interleave; per cell the arithmetic is unchanged
__global__ __launch_bounds__(256) void softmax_shuffle_kernel(const float* __restrict__ logits, float* __restrict__ prob,
                                                              int B, int Hc, int Wc, int r, int ld, int mode, int* __restrict__ status) {
    const int lane = threadIdx.x & 63;
    const int64_t cell0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * SMX_CPW;
    const int64_t ncell = (int64_t)B * Hc * Wc;
    if (cell0 >= ncell) return;
    const int nch = r * r + 1;
    float v0[SMX_CPW], v1[SMX_CPW];
#pragma unroll
    for (int q = 0; q < SMX_CPW; ++q) {
        const int64_t cell = cell0 + q < ncell ? cell0 + q : ncell - 1;
        const float* lp = logits + cell * ld;
        v0[q] = (lane < nch) ? lp[lane] : -INFINITY;
        v1[q] = (lane + 64 < nch) ? lp[lane + 64] : -INFINITY;
    }
#pragma unroll
    for (int q = 0; q < SMX_CPW; ++q) {
        const int64_t cell = cell0 + q;
        if (cell >= ncell) break;                                    // wave-uniform
        float e0, e1, inv;
        if (mode == 0) {
            const float mx = xp_wave_max(fmaxf(v0[q], v1[q]));
            e0 = (lane < nch) ? expf(v0[q] - mx) : 0.f;
            e1 = (lane + 64 < nch) ? expf(v1[q] - mx) : 0.f;
            inv = 1.f / xp_wave_sum(e0 + e1);
            e0 *= inv; e1 *= inv;
            if (status && !(inv <= 1.f) && lane == 0) atomicOr(status, XP_STATUS_PROB);     // a NaN / +-inf logit makes the sum NaN (1 <= sum <= 65 otherwise)
        } else {
            e0 = (lane < nch) ? expf(v0[q]) : 0.f;
            e1 = (lane + 64 < nch) ? expf(v1[q]) : 0.f;
            const float den = xp_wave_sum(e0 + e1) + 0.00001f;
            e0 = e0 / den; e1 = e1 / den;
        }
        const int w = (int)(cell % Wc), h = (int)((cell / Wc) % Hc);
        const int64_t b = cell / ((int64_t)Wc * Hc);
        const int64_t Wf = (int64_t)Wc * r;
        for (int k = 0; k < 2; ++k) {
            const int ch = lane + 64 * k;
            if (ch < r * r) {
                const int i = ch / r, j = ch - i * r;
                prob[(b * Hc * r + (h * r + i)) * Wf + (w * r + j)] = k ? e1 : e0;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// L2 normalise rows (F.normalize p=2 dim=channel, eps 1e-12: x / max(||x||, eps); XPoint.py:365-366).
// eps < 0 selects the SuperPoint form x / ||x|| without clamp (SuperPointMagicLeap.py:59-60).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t M, int C,
                                                          float eps, int* __restrict__ status) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) { const float v = xr[c]; s = fmaf(v, v, s); }
    float nrm = sqrtf(xp_wave_sum(s));
    if (status && !(nrm < INFINITY) && lane == 0) atomicOr(status, XP_STATUS_DESC);      // NaN or inf anywhere in the row
    if (eps >= 0.f) nrm = fmaxf(nrm, eps);
    float* yr = y + row * C;
    for (int c = lane; c < C; c += 64) yr[c] = xr[c] / nrm;
}

// NHWC (B, HW, C) -> NCHW (B, C, HW) through a padded 32x32 LDS tile.
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int HW, int C) {
    __shared__ float t[32][33];
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int p = p0 + i, c = c0 + tx;
        t[i][tx] = (p < HW && c < C) ? x[((int64_t)b * HW + p) * C + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, p = p0 + tx;
        if (p < HW && c < C) y[((int64_t)b * C + c) * HW + p] = t[tx][i];
    }
}

__global__ __launch_bounds__(256) void mul_mask_kernel(const float* __restrict__ x, const uint8_t* __restrict__ m,
                                                       float* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = x[i] * (m[i] ? 1.f : 0.f);
}

__global__ __launch_bounds__(256) void mul_mask4_kernel(const float4* __restrict__ x, const unsigned* __restrict__ m, float4* __restrict__ y, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4 v = x[i];
    const unsigned k = m[i];
    y[i] = make_float4(v.x * ((k & 0xffu) ? 1.f : 0.f), v.y * ((k & 0xff00u) ? 1.f : 0.f), v.z * ((k & 0xff0000u) ? 1.f : 0.f), v.w * ((k & 0xff000000u) ? 1.f : 0.f));
}

// 2x2 max pool stride 2, NHWC (conv backbones: XPoint.py:451-466, SuperPointMagicLeap.py:42-48).
__global__ __launch_bounds__(256) void maxpool2_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2;
    const int64_t total = (int64_t)B * Ho * Wo * C;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const int64_t pix = idx / C;
    const int ow = (int)(pix % Wo), oh = (int)((pix / Wo) % Ho);
    const int64_t b = pix / ((int64_t)Wo * Ho);
    const float* p = x + ((b * H + oh * 2) * W + ow * 2) * (int64_t)C + c;
    y[idx] = fmaxf(fmaxf(p[0], p[C]), fmaxf(p[(int64_t)W * C], p[(int64_t)W * C + C]));
}

// ---------------------------------------------------------------------------------------------
// Data ingest (reference datasets/ImagePairDataset.py:199-208, :254-274 folder mode): a decoded 8-bit image ->
// gray -> / 255 -> crop, straight into a slot of the (B, 1, h, w) f32 batch.  Gray = OpenCV's COLOR_BGR2GRAY for 8-bit
// images, fixed point with 14 fractional bits: (B*1868 + G*9617 + R*4899 + 8192) >> 14.  `lut` = float32(k / 255.0)
// for k = 0..255 computed by the host in double precision, as numpy does for the reference.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ingest_u8_kernel(const uint8_t* __restrict__ src, int H0, int W0, int ch, int top, int left,
                                                        int h, int w, const float* __restrict__ lut, float* __restrict__ dst) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const uint8_t* px = src + ((int64_t)(top + y) * W0 + (left + x)) * ch;
    int g;
    if (ch == 1) g = px[0];
    else g = (px[2] * 1868 + px[1] * 9617 + px[0] * 4899 + 8192) >> 14;      // src is R, G, B(, A) interleaved
    dst[(int64_t)y * w + x] = lut[g];
}

}  // namespace

// y = fp16-round(x) kept in f32 containers (the mixed-precision class's inter-op tensors): n floats, 16-byte aligned when n % 4 == 0
__global__ __launch_bounds__(256) void round_f16_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        float4 v = *reinterpret_cast<const float4*>(x + i);
        v.x = ew_r16(v.x); v.y = ew_r16(v.y); v.z = ew_r16(v.z); v.w = ew_r16(v.w);
        *reinterpret_cast<float4*>(y + i) = v;
    } else {
        for (int64_t j = i; j < n; ++j) y[j] = ew_r16(x[j]);
    }
}
extern "C" int xp_round_f16(const float* x, float* y, int64_t n, void* stream) {
    XP_CHECK_ARG(x && y && n >= 0, "xp_round_f16: bad args");
    XP_CHECK_ARG((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "xp_round_f16: buffers must be 16-byte aligned");
    if (n == 0) return XP_OK;
    XpProfScope prof("round_f16", (hipStream_t)stream, 0.0, 8.0 * n);
    hipLaunchKernelGGL(round_f16_kernel, dim3(xp_cdiv(n, 1024)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    XP_LAUNCH_CHECK();
    return XP_OK;
}

static int layernorm_impl(const float* x, float* y, const float* w, const float* b, int64_t rows, int C, float eps, int gelu, void* stream);
extern "C" int xp_layernorm(const float* x, float* y, const float* w, const float* b, int64_t rows, int C, float eps,
                            int gelu, void* stream) {
    // mixed-precision class: LayerNorm of a half tensor returns a half tensor (statistics in f32, as here).  Only the form the model uses is
    // implemented for it — no fused GELU (autocast rounds BETWEEN LayerNorm and GELU: the stem kernel does that itself), C a multiple of 4, y 16-byte
    // aligned (xp_round_f16's vector form) — and anything else is refused instead of silently skipping a rounding point.
    if (xp_amp_value()) {
        XP_CHECK_ARG(gelu == 0, "xp_layernorm: the fused GELU has no mixed-precision form (the LayerNorm -> GELU boundary would not be rounded); use xp_stem_conv_ln_gelu or gelu = 0");
        XP_CHECK_ARG(C % 4 == 0 && ((uintptr_t)y & 15) == 0, "xp_layernorm: mixed-precision mode needs C %% 4 == 0 and a 16-byte aligned output (C = %d)", C);
    }
    const int rc = layernorm_impl(x, y, w, b, rows, C, eps, gelu, stream);
    if (rc == XP_OK && xp_amp_value() && rows > 0) return xp_round_f16(y, y, rows * C, stream);
    return rc;
}
static int layernorm_impl(const float* x, float* y, const float* w, const float* b, int64_t rows, int C, float eps,
                          int gelu, void* stream) {
    XP_CHECK_ARG(x && y && w && b, "xp_layernorm: null pointer");
    XP_CHECK_ARG(C > 0 && C <= 1024, "xp_layernorm: C must be in [1,1024] (got %d)", C);
    if (rows == 0) return XP_OK;
    static const bool by_shape = getenv("XP_PROF_SHAPES") != nullptr;
    XpProfScope prof(by_shape ? ("layernorm_C" + std::to_string(C)).c_str() : "layernorm", (hipStream_t)stream, 8.0 * rows * C, 8.0 * rows * C);
    hipStream_t s = (hipStream_t)stream;
    const bool vec = (C % 4 == 0) && ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)w | (uintptr_t)b) & 15) == 0);
    const int C4 = C / 4;
#define XP_LN_LAUNCH(LPR, NV) hipLaunchKernelGGL((layernorm_vec_kernel<LPR, NV, (NV == 1 ? 4 : 1)>), dim3(xp_cdiv(rows, 4 * (64 / LPR) * (NV == 1 ? 4 : 1))), dim3(256), 0, s, x, y, w, b, rows, C, eps, gelu)
    // (A lane mapping that covers 96 / 192 / 384-channel rows exactly — LPR x 3 — was measured in round 3: C = 96 54 -> 44 us, 13 us per step, but it changes the
    // order of the two row sums, i.e. the last bit of some outputs and with it WHICH near-tied keypoints agree with the reference; removed in round 6 so that
    // xp_layernorm and xp_layernorm_p32 cannot drift apart.)
    if (!vec) hipLaunchKernelGGL(layernorm_kernel, dim3(xp_cdiv(rows, 4)), dim3(256), 0, s, x, y, w, b, rows, C, eps, gelu);
    else if (C4 <= 4) XP_LN_LAUNCH(4, 1);
    else if (C4 <= 8) XP_LN_LAUNCH(8, 1);
    else if (C4 <= 16) XP_LN_LAUNCH(16, 1);
    else if (C4 <= 32) XP_LN_LAUNCH(32, 1);
    else if (C4 <= 64) XP_LN_LAUNCH(64, 1);
    else if (C4 <= 128) XP_LN_LAUNCH(64, 2);
    else if (C4 <= 192) XP_LN_LAUNCH(64, 3);
    else XP_LN_LAUNCH(64, 4);
#undef XP_LN_LAUNCH
    XP_LAUNCH_CHECK();
    return XP_OK;
}

extern "C" int xp_layernorm_p32(const float* x, void* y_p32, const float* w, const float* b, int64_t rows, int C, float eps, void* stream) {
    XP_CHECK_ARG(x && y_p32 && w && b, "xp_layernorm_p32: null pointer");
    XP_CHECK_ARG(C >= 32 && C <= 1024 && C % 32 == 0, "xp_layernorm_p32: C must be a multiple of 32 in [32, 1024] (got %d)", C);
    XP_CHECK_ARG((((uintptr_t)x | (uintptr_t)y_p32 | (uintptr_t)w | (uintptr_t)b) & 15) == 0, "xp_layernorm_p32: buffers must be 16-byte aligned");
    XP_CHECK_ARG(!xp_amp_value(), "xp_layernorm_p32: no mixed-precision form (the P32 image feeds the f32-grade ring GEMM only)");
    if (rows == 0) return XP_OK;
    XpProfScope prof("layernorm_p32", (hipStream_t)stream, 8.0 * rows * C, 8.0 * rows * C);
    hipStream_t s = (hipStream_t)stream;
    float* y = reinterpret_cast<float*>(y_p32);
    const int C4 = C / 4;
    // the lane / row geometry of xp_layernorm for the same C: the statistics and the normalised values are the same bits, only the store differs
#define XP_LNP_LAUNCH(LPR, NV) hipLaunchKernelGGL((layernorm_vec_kernel<LPR, NV, (NV == 1 ? 4 : 1), true>), dim3(xp_cdiv(rows, 4 * (64 / LPR) * (NV == 1 ? 4 : 1))), dim3(256), 0, s, x, y, w, b, rows, C, eps, 0)
    if (C4 <= 8) XP_LNP_LAUNCH(8, 1);
    else if (C4 <= 16) XP_LNP_LAUNCH(16, 1);
    else if (C4 <= 32) XP_LNP_LAUNCH(32, 1);
    else if (C4 <= 64) XP_LNP_LAUNCH(64, 1);
    else if (C4 <= 128) XP_LNP_LAUNCH(64, 2);
    else if (C4 <= 192) XP_LNP_LAUNCH(64, 3);
    else XP_LNP_LAUNCH(64, 4);
#undef XP_LNP_LAUNCH
    XP_LAUNCH_CHECK();
    return XP_OK;
}

extern "C" int xp_dwconv3x3_silu(const float* x, const float* w9c, float* y, int batch, int H, int W, int C, void* stream) {
    XP_CHECK_ARG(x && w9c && y, "xp_dwconv3x3_silu: null pointer");
    XP_CHECK_ARG(C % 4 == 0, "xp_dwconv3x3_silu: C %% 4 != 0");
    const int64_t total = (int64_t)batch * ((H + DW_PH - 1) / DW_PH) * ((W + DW_PW - 1) / DW_PW) * (C / 4);
    XpProfScope prof("dwconv3x3_silu", (hipStream_t)stream, 22.0 * batch * H * W * C, 8.0 * batch * H * W * C);
    if (xp_amp_value()) hipLaunchKernelGGL(dwconv3x3_silu_kernel<true>, dim3(xp_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, w9c, y, batch, H, W, C);
    else hipLaunchKernelGGL(dwconv3x3_silu_kernel<false>, dim3(xp_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, w9c, y, batch, H, W, C);
    XP_LAUNCH_CHECK();
    return XP_OK;
}

extern "C" int xp_stem_conv_ln_gelu(const float* img, const float* w9co, const float* bias, const float* ln_w,
                                    const float* ln_b, float* y, int batch, int H, int W, int Co, float eps, void* stream) {
    XP_CHECK_ARG(img && w9co && bias && ln_w && ln_b && y, "xp_stem_conv_ln_gelu: null pointer");
    const int Ho = H / 2 + (H & 1), Wo = W / 2 + (W & 1);
    const int64_t total = (int64_t)batch * Ho * Wo;
    dim3 grid(xp_cdiv(total, 64)), block(64);
    hipStream_t s = (hipStream_t)stream;
    XpProfScope prof("stem_conv_ln_gelu", (hipStream_t)stream, (double)total * Co * 30.0, 4.0 * ((double)batch * H * W + (double)total * Co));
    const bool amp = xp_amp_value() != 0;
    if (Co == 48 && amp) hipLaunchKernelGGL((stem_conv_ln_gelu_kernel<48, true>), grid, block, 0, s, img, w9co, bias, ln_w, ln_b, y, batch, H, W, eps);
    else if (Co == 16 && amp) hipLaunchKernelGGL((stem_conv_ln_gelu_kernel<16, true>), grid, block, 0, s, img, w9co, bias, ln_w, ln_b, y, batch, H, W, eps);
    else if (Co == 48) hipLaunchKernelGGL((stem_conv_ln_gelu_kernel<48, false>), grid, block, 0, s, img, w9co, bias, ln_w, ln_b, y, batch, H, W, eps);
    else if (Co == 16) hipLaunchKernelGGL((stem_conv_ln_gelu_kernel<16, false>), grid, block, 0, s, img, w9co, bias, ln_w, ln_b, y, batch, H, W, eps);
    else { xp_set_error("xp_stem_conv_ln_gelu: Co must be 48 or 16 (EMBED_DIM 96 / 32), got %d", Co); return XP_ERR_ARG; }
    XP_LAUNCH_CHECK();
    return XP_OK;
}

int xp_depth_to_space_nhwc_st(const float* x, float* y, int batch, int H, int W, int C, int bs, float limit, int* status, void* stream) {
    XP_CHECK_ARG(x && y && C % (bs * bs) == 0, "xp_depth_to_space_nhwc: bad args");
    const int64_t total = (int64_t)batch * H * W * C;
    XpProfScope prof("depth_to_space", (hipStream_t)stream, 0.0, 8.0 * total);
    const bool vec = (C / (bs * bs)) % 4 == 0 && total < (1ll << 31) && ((((uintptr_t)x | (uintptr_t)y) & 15) == 0);
    if (vec) hipLaunchKernelGGL(depth_to_space_vec_kernel, dim3(xp_cdiv(total / 4, 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(x),
                                reinterpret_cast<float4*>(y), batch, H, W, C / 4, bs, limit, status);
    else hipLaunchKernelGGL(depth_to_space_kernel, dim3(xp_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, batch, H, W, C, bs, limit, status);
    XP_LAUNCH_CHECK();
    return XP_OK;
}
extern "C" int xp_depth_to_space_nhwc(const float* x, float* y, int batch, int H, int W, int C, int bs, void* stream) {
    return xp_depth_to_space_nhwc_st(x, y, batch, H, W, C, bs, 0.f, nullptr, stream);
}

extern "C" int xp_softmax_shuffle(const float* logits, float* prob, int batch, int Hc, int Wc, int r, int ld, int mode, void* stream) {
    return xp_softmax_shuffle_st(logits, prob, batch, Hc, Wc, r, ld, mode, nullptr, stream);
}
int xp_softmax_shuffle_st(const float* logits, float* prob, int batch, int Hc, int Wc, int r, int ld, int mode, int* status, void* stream) {
    XP_CHECK_ARG(logits && prob, "xp_softmax_shuffle: null pointer");
    XP_CHECK_ARG(r * r + 1 <= 128 && ld >= r * r + 1, "xp_softmax_shuffle: r*r+1 must be <= 128 and <= ld");
    const int64_t cells = (int64_t)batch * Hc * Wc;
    XpProfScope prof("softmax_shuffle", (hipStream_t)stream, (double)cells * 4.0 * (r * r + 1), 4.0 * cells * (2.0 * r * r + 1));
    hipLaunchKernelGGL(softmax_shuffle_kernel, dim3(xp_cdiv(cells, 4 * SMX_CPW)), dim3(256), 0, (hipStream_t)stream, logits, prob, batch, Hc, Wc, r, ld, mode, status);
    XP_LAUNCH_CHECK();
    return XP_OK;
}

extern "C" int xp_l2norm_rows(const float* x, float* y, int64_t rows, int C, float eps, void* stream) {
    return xp_l2norm_rows_st(x, y, rows, C, eps, nullptr, stream);
}
int xp_l2norm_rows_st(const float* x, float* y, int64_t rows, int C, float eps, int* status, void* stream) {
    XP_CHECK_ARG(x && y, "xp_l2norm_rows: null pointer");
    if (rows == 0) return XP_OK;
    XpProfScope prof("l2norm_rows", (hipStream_t)stream, 3.0 * rows * C, 8.0 * rows * C);
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3(xp_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, y, rows, C, eps, status);
    XP_LAUNCH_CHECK();
    return XP_OK;
}

extern "C" int xp_nhwc_to_nchw(const float* x, float* y, int batch, int HW, int C, void* stream) {
    XP_CHECK_ARG(x && y, "xp_nhwc_to_nchw: null pointer");
    dim3 grid(xp_cdiv(HW, 32), xp_cdiv(C, 32), batch);
    XpProfScope prof("nhwc_to_nchw", (hipStream_t)stream, 0.0, 8.0 * batch * HW * C);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, HW, C);
    XP_LAUNCH_CHECK();
    return XP_OK;
}

// two f32 blocks (and optionally two u8 blocks) of n elements each copied into the halves of the batch buffers; 16-byte units, scalar tails
__global__ __launch_bounds__(256) void stage_pair_batch_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                               const uint8_t* __restrict__ ma, const uint8_t* __restrict__ mb, uint8_t* __restrict__ mout,
                                                               int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n4 = n >> 2;
    const bool al = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(out) | (uintptr_t)(n * 4)) & 15) == 0;
    if (al) {
        for (int64_t i = t0; i < 2 * n4; i += stride) {
            const bool second = i >= n4;
            const int64_t j = second ? i - n4 : i;
            reinterpret_cast<float4*>(out + (second ? n : 0))[j] = reinterpret_cast<const float4*>(second ? b : a)[j];
        }
        for (int64_t i = 4 * n4 + t0; i < n; i += stride) { out[i] = a[i]; out[n + i] = b[i]; }
    } else {
        for (int64_t i = t0; i < n; i += stride) { out[i] = a[i]; out[n + i] = b[i]; }
    }
    if (!mout) return;
    const int64_t n16 = n >> 4;
    const bool alm = ((reinterpret_cast<uintptr_t>(ma) | reinterpret_cast<uintptr_t>(mb) | reinterpret_cast<uintptr_t>(mout) | (uintptr_t)n) & 15) == 0;
    if (alm) {
        for (int64_t i = t0; i < 2 * n16; i += stride) {
            const bool second = i >= n16;
            const int64_t j = second ? i - n16 : i;
            reinterpret_cast<uint4*>(mout + (second ? n : 0))[j] = reinterpret_cast<const uint4*>(second ? mb : ma)[j];
        }
        for (int64_t i = 16 * n16 + t0; i < n; i += stride) { mout[i] = ma[i]; mout[n + i] = mb[i]; }
    } else {
        for (int64_t i = t0; i < n; i += stride) { mout[i] = ma[i]; mout[n + i] = mb[i]; }
    }
}

extern "C" int xp_stage_pair_batch(const float* optical, const float* thermal, float* images, const uint8_t* mask_optical,
                                   const uint8_t* mask_thermal, uint8_t* masks, int64_t n, void* stream) {
    XP_CHECK_ARG(n >= 0, "xp_stage_pair_batch: negative size");
    if (n == 0) return XP_OK;
    XP_CHECK_ARG(optical && thermal && images, "xp_stage_pair_batch: null pointer");
    XP_CHECK_ARG((mask_optical != nullptr) == (masks != nullptr) && (mask_thermal != nullptr) == (masks != nullptr),
                 "xp_stage_pair_batch: pass both masks and the mask buffer, or none of them");
    XpProfScope prof("stage_pair_batch", (hipStream_t)stream, 0.0, (masks ? 20.0 : 16.0) * n);
    const int64_t units = (n + 1) / 2;      // 2n / 4 sixteen-byte units of image data
    hipLaunchKernelGGL(stage_pair_batch_kernel, dim3((unsigned)std::min<int64_t>(xp_cdiv(units, (int64_t)256), 8192)), dim3(256), 0, (hipStream_t)stream,
                       optical, thermal, images, mask_optical, mask_thermal, masks, n);
    XP_LAUNCH_CHECK();
    return XP_OK;
}

extern "C" int xp_mul_mask(const float* x, const uint8_t* mask, float* y, int64_t n, void* stream) {
    XP_CHECK_ARG(x && mask && y, "xp_mul_mask: null pointer");
    if (n == 0) return XP_OK;
    XpProfScope prof("mul_mask", (hipStream_t)stream, 1.0 * n, 9.0 * n);
    if (n % 4 == 0 && ((((uintptr_t)x | (uintptr_t)y) & 15) == 0) && (((uintptr_t)mask & 3) == 0))
        hipLaunchKernelGGL(mul_mask4_kernel, dim3(xp_cdiv(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(x),
                           reinterpret_cast<const unsigned*>(mask), reinterpret_cast<float4*>(y), n / 4);
    else hipLaunchKernelGGL(mul_mask_kernel, dim3(xp_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, mask, y, n);
    XP_LAUNCH_CHECK();
    return XP_OK;
}

// ---------------------------------------------------------------------------------------------
// RegNet cost volume + global average pool without the volume (reference RegNet.py:44-52: cv = bmm(x1^T, x2) (hw, hw), then
// adaptive_avg_pool2d(cv.view(N, hw, H', W'), 1) = the mean over the SECOND image's positions):
//     v[b][p] = (1 / hw) sum_q  a[b][p] . b[b][q]  =  a[b][p] . mean_q b[b][q]
// — O(hw C) instead of O(hw^2 C), one launch for the whole batch instead of two GEMM launches per sample.  One workgroup per sample: column means of b
// (thread = channel, rows in ascending order), then a wave per row of a (lane partial sums over c = lane, lane + 64, ..., wave reduction).
// ---------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void costvolume_mean_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ v, int hw, int C) {
    extern __shared__ float s_mean[];                     // C floats
    const int n = blockIdx.x;
    const float* bb = b + (int64_t)n * hw * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
        for (int q = 0; q < hw; ++q) s += bb[(int64_t)q * C + c];
        s_mean[c] = s / (float)hw;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const float* aa = a + (int64_t)n * hw * C;
    for (int p = wave; p < hw; p += nw) {
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s = fmaf(aa[(int64_t)p * C + c], s_mean[c], s);
        s = xp_wave_sum(s);
        if (lane == 0) v[(int64_t)n * hw + p] = s;
    }
}
}  // namespace

extern "C" int xp_costvolume_mean(const float* a, const float* b, float* v, int batch, int hw, int C, void* stream) {
    XP_CHECK_ARG(a && b && v, "xp_costvolume_mean: null pointer");
    XP_CHECK_ARG(batch > 0 && hw > 0 && C > 0 && C <= 8192, "xp_costvolume_mean: bad shape batch %d, hw %d, C %d", batch, hw, C);
    XpProfScope prof("costvolume_mean", (hipStream_t)stream, 4.0 * batch * hw * (double)C, 4.0 * batch * hw * (2.0 * C + 1.0));
    hipLaunchKernelGGL(costvolume_mean_kernel, dim3(batch), dim3(256), (size_t)C * sizeof(float), (hipStream_t)stream, a, b, v, hw, C);
    XP_LAUNCH_CHECK();
    return XP_OK;
}

extern "C" int xp_maxpool2_nhwc(const float* x, float* y, int batch, int H, int W, int C, void* stream) {
    XP_CHECK_ARG(x && y, "xp_maxpool2_nhwc: null pointer");
    const int64_t total = (int64_t)batch * (H / 2) * (W / 2) * C;
    XpProfScope prof("maxpool2", (hipStream_t)stream, 3.0 * total, 20.0 * total);
    hipLaunchKernelGGL(maxpool2_kernel, dim3(xp_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, batch, H, W, C);
    XP_LAUNCH_CHECK();
    return XP_OK;
}

extern "C" int xp_ingest_u8(const uint8_t* src, int H0, int W0, int channels, int top, int left, int h, int w,
                            const float* lut256, float* dst, void* stream) {
    XP_CHECK_ARG(src && lut256 && dst, "xp_ingest_u8: null pointer");
    XP_CHECK_ARG(channels == 1 || channels == 3 || channels == 4, "xp_ingest_u8: channels must be 1 (gray), 3 (RGB) or 4 (RGBA), got %d", channels);
    XP_CHECK_ARG(h > 0 && w > 0 && top >= 0 && left >= 0 && top + h <= H0 && left + w <= W0,
                 "xp_ingest_u8: crop (%d,%d)+(%dx%d) outside the %dx%d image", top, left, h, w, H0, W0);
    XpProfScope prof("ingest_u8", (hipStream_t)stream, 6.0 * h * w, (double)h * w * (channels + 4));
    hipLaunchKernelGGL(ingest_u8_kernel, dim3(xp_cdiv(w, 256), h), dim3(256), 0, (hipStream_t)stream, src, H0, W0, channels, top, left, h, w, lut256, dst);
    XP_LAUNCH_CHECK();
    return XP_OK;
}

// 8-bit gray batch -> f32 / 255 (the reference loader's `img.astype(float32) / 255`, xpoint/datasets/ImagePairDataset.py:254-274: the same IEEE division, so
// the same bits as a host conversion): the device half of the streaming path's 8-bit upload (predict.PairPipeline._stage_inputs) — 4 bytes read, 16 written per lane.
namespace {
__global__ __launch_bounds__(256) void u8_to_unit_f32_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 4 <= n) {
        const uchar4 v = *reinterpret_cast<const uchar4*>(src + i);
        *reinterpret_cast<float4*>(dst + i) = make_float4((float)v.x / 255.f, (float)v.y / 255.f, (float)v.z / 255.f, (float)v.w / 255.f);
    } else {
        for (int64_t j = i; j < n; ++j) dst[j] = (float)src[j] / 255.f;
    }
}
}  // namespace

extern "C" int xp_u8_to_unit_f32(const uint8_t* src, float* dst, int64_t n, void* stream) {
    XP_CHECK_ARG(src && dst && n >= 0, "xp_u8_to_unit_f32: bad args");
    XP_CHECK_ARG((((uintptr_t)src & 3) | ((uintptr_t)dst & 15)) == 0, "xp_u8_to_unit_f32: src must be 4-byte, dst 16-byte aligned");
    if (n == 0) return XP_OK;
    XpProfScope prof("u8_to_unit_f32", (hipStream_t)stream, (double)n, 5.0 * n);
    hipLaunchKernelGGL(u8_to_unit_f32_kernel, dim3((unsigned)xp_cdiv(n, 1024)), dim3(256), 0, (hipStream_t)stream, src, dst, n);
    XP_LAUNCH_CHECK();
    return XP_OK;
}

// Result lists -> pinned (device-mapped) host memory by a KERNEL instead of the runtime's copy engines.  Why: in the streaming loop the device-to-host copies
// of step i (queued behind its last kernels) and the host-to-device image upload of step i + 2 share a copy-engine queue, and the upload waits behind a copy
// that itself waits for kernels — the encoder of step i + 2 starts late (bench.py, profiles/r5_streaming_parts.txt: uploads alone -2 %, downloads alone 0 %,
// both -11 %).  The lists are a few MB: 16-byte stores over the link from a handful of workgroups, ordered like any other kernel of the stream.
namespace {
__global__ __launch_bounds__(256) void copy_to_host_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16, size_t bytes) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
    if (blockIdx.x == 0) {
        const unsigned char* sb = reinterpret_cast<const unsigned char*>(src); unsigned char* db = reinterpret_cast<unsigned char*>(dst);
        for (size_t j = n16 * 16 + threadIdx.x; j < bytes; j += 256) db[j] = sb[j];
    }
}
}  // namespace

extern "C" int xp_copy_to_mapped_host(const void* src_dev, void* dst_host, size_t bytes, void* stream) {
    XP_CHECK_ARG(src_dev && dst_host, "xp_copy_to_mapped_host: null pointer");
    XP_CHECK_ARG((((uintptr_t)src_dev | (uintptr_t)dst_host) & 15) == 0, "xp_copy_to_mapped_host: buffers must be 16-byte aligned");
    if (bytes == 0) return XP_OK;
    const size_t n16 = bytes / 16;
    const unsigned grid = (unsigned)std::min<size_t>(64, std::max<size_t>(1, (n16 + 255) / 256));
    hipLaunchKernelGGL(copy_to_host_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4*)src_dev, (uint4*)dst_host, n16, bytes);
    XP_LAUNCH_CHECK();
    return XP_OK;
}
