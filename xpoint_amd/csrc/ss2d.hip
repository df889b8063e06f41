// Fused SS2D core in pixel (NHWC) layout — the MI355X-first form of reference
// VMamba.py:601-646 (forward_corev2: cross_scan -> x_proj/dt_proj -> selective_scan -> cross_merge ->
// out_norm) for d_state = 1.
//
// The four scan routes (csm_triton.py:22-53) are permutations of the same pixels and x_proj / dt_proj
// are pointwise in L, so nothing is materialised per route: `xdbl` (M, 4*(R+2)) is ONE GEMM in pixel
// layout, dt_proj + softplus + exp run inside the scan, and the four routes become four traversals
// (row-major fwd/bwd, column-major fwd/bwd) of the same NHWC buffers.  With lanes over channels every
// access is a contiguous C*4-byte segment in either traversal, so the column routes need no transpose.
//
// The recurrence is made parallel over L by chunks of T pixels:
//   pass 1  per (image, route pair, chunk, channel): a-product P and local end state S of both routes
//   pass 2  per (image, route, channel): sequential carry over chunks -> state entering each chunk
//   pass 3  per (image, route pair, chunk, channel): re-run with the true start state, forward route
//           into an LDS tile, backward route added to it; the column pair adds the row pair's result
//           in the reference's order (y0 + y2) + (y1 + y3) (csm_triton.py:60-62) and applies out_norm
//           (LayerNorm over C, VMamba.py:644) before the single store.
// Directions are stored in the order (0, 2, 1, 3) so that a pair's operands are adjacent.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <string>
#include <type_traits>

#include "gemm_h2_core.h"
#include "xp_common.h"

#ifndef XP_SS2D_DBG
#define XP_SS2D_DBG 0   /* timing experiments only (wrong results): 1 no out_norm tail, 2 no loads of the row pair's partial sums, 4 no step arithmetic (projection, softplus, exp), 8 no u loads */
#endif

// L bound of the sequential (one wave per route) form, per image: see ss2d_core_impl
#define XP_SS2D_SEQ_DEFAULT_MAXL 8192

namespace {

constexpr float XP_L2E = 1.44269504088896340736f;   // log2(e): folded into the dt weights, the dt bias and A where they are loaded

struct SS2DParams {
    const float* u;      // (B, H, W, C)   after dwconv + SiLU
    const float* xdbl;   // (B*H*W, 4*(R+2))  [pair][dir][dtr(R), B, C]
    const float* wdt;    // (4, R, C)  directions in the order (0,2,1,3); channel-contiguous so that a wave's weight loads coalesce
    const float* dtb;    // (4, C)
    const float* A;      // (4, C)    = -exp(A_logs)
    const float* Dp;     // (4, C)
    const float* ln_w;   // (C) out_norm
    const float* ln_b;
    float* wsP;          // (B, 2, nc, 2, C)
    float* wsS;          // same; pass 2 overwrites S with the chunk start state
    float* ya;           // (B, H, W, C) row-pair partial
    float* out;          // (B, H, W, C)
    int out_p32;         // sequential form only: out is the P32 image [row][c / 32][plane][32] of the result (ring_core.h), not f32 rows
    int Bn, H, W, C, T, nc, cpb;
    float eps;
};

__device__ __forceinline__ int pixel_of(int l, int L, int H, int W, bool colmajor) {
    if (l >= L) return -1;
    if (!colmajor) return l;
    int w = l / H, h = l - w * H;   // route 1: l = w*H + h   (x.transpose(2,3).flatten)
    return h * W + w;
}

// One scan step's operands for one route: xr = [dt_rank values, B, C] (R + 2 floats, 8-byte aligned in LDS; 16-byte
// aligned when (R + 2) % 4 == 0).  Read with the widest aligned LDS loads (b128 / b64 broadcasts), not R + 2 b32 reads.
// AMP (mixed-precision class, xp_set_amp_mode): the dt projection is a half convolution under autocast — its output is rounded to fp16 BEFORE
// the f32 bias is added (csms6s.py:47-50 adds delta_bias to the half tensor in f32) — so w and bias arrive WITHOUT the log2(e) factor
template <int R, bool AMP = false>
__device__ __forceinline__ void step_vals(const float* __restrict__ xr, const float (&w)[R], float bias, float A,
                                          float u, float& a, float& b, float& Cv) {
    float xv[R + 2];
    if constexpr ((R + 2) % 4 == 0) {
#pragma unroll
        for (int q = 0; q < (R + 2) / 4; ++q) {
            const float4 t = *reinterpret_cast<const float4*>(xr + 4 * q);
            xv[4 * q] = t.x; xv[4 * q + 1] = t.y; xv[4 * q + 2] = t.z; xv[4 * q + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int q = 0; q < (R + 2) / 2; ++q) {
            const float2 t = *reinterpret_cast<const float2*>(xr + 2 * q);
            xv[2 * q] = t.x; xv[2 * q + 1] = t.y;
        }
    }
    if (XP_SS2D_DBG & 4) { a = 0.5f; b = xv[R] * u; Cv = xv[R + 1]; return; }
    // w, bias and A arrive pre-multiplied by log2(e) (XP_L2E at their loads): the chain below is x * log2(e) directly
    float dt;
    if constexpr (!AMP) {
        dt = fmaf(w[0], xv[0], bias);
#pragma unroll
        for (int r = 1; r < R; ++r) dt = fmaf(w[r], xv[r], dt);
    } else {
        dt = w[0] * xv[0];
#pragma unroll
        for (int r = 1; r < R; ++r) dt = fmaf(w[r], xv[r], dt);
        dt = ((float)(_Float16)dt + bias) * XP_L2E;
    }
    float delta;
    xp_softplus_decay_l2(dt, A, delta, a);          // softplus + exp(delta * A): xp_common.h
    b = delta * xv[R] * u;
    Cv = xv[R + 1];
}

// ---- dt projection of a 32-pixel tile on the matrix pipe (dt_rank >= 12: the deep stages) ------------------------------------------
//   dt[pixel][channel] = bias[channel] + sum_r x[pixel][r] Wdt[channel][r]   for the wave's 64 channels: two 32x32 MFMA tiles x ceil(R / 16)
// slabs, both operands split into two fp16 planes, three products per slab (gemm_h2_core.h: operand error <= 2^-23, per product <= 2^-21 worst / ~2^-25 typical — f32-grade) instead of R
// FMAs per step and lane.  The weight fragments stay in registers (as many as the R scalar weights they replace), the pixel rows are split
// from the LDS tile.  The accumulator layout (lane = column = channel, 16 registers = rows {0-3, 8-11, 16-19, 24-27} + 4 (lane >> 5))
// becomes "lane = channel, 32 registers = the tile's 32 pixels" with one v_permlane32_swap per register pair of the two tiles — the
// scan's own layout: dt of tile pixel j is dtv[(j >> 2) & 1][(j & 3) + 4 (j >> 3)].
template <int R>
struct DtWeights {
    static constexpr int KS = (R + 15) / 16;
    f16x8_t bw[2][KS][2];                                               // [channel tile][k slab][plane]
    float bias[2];
    float bias_lane;                                                    // AMP: the unscaled bias of channel cbase + lane (the layout AFTER dt_tile's lane swap)
};
__device__ __forceinline__ void dt_split8(const float (&v)[8], f16x8_t& hi, f16x8_t& lo) {
    uint2 a0, a1, b0, b1;
    h2_split4(make_float4(v[0], v[1], v[2], v[3]), a0, a1);
    h2_split4(make_float4(v[4], v[5], v[6], v[7]), b0, b1);
    union { uint4 u; f16x8_t h; } x, y;
    x.u = make_uint4(a0.x, a0.y, b0.x, b0.y); y.u = make_uint4(a1.x, a1.y, b1.x, b1.y);
    hi = x.h; lo = y.h;
}
// wdt_dir: (R, C) weights of one direction, dtb_dir: (C); cbase: first of the wave's 64 channels.  log2(e) folded in (see step_vals).
// AMP (mixed-precision classes): the projection is a half convolution whose output is rounded to fp16 BEFORE the f32 bias is added (see step_vals): weights
// and bias arrive without the log2(e) factor, the accumulator starts at 0, and dt_tile finishes with (r16(acc) + bias) * log2(e)
template <int R, bool AMP = false>
__device__ __forceinline__ void dt_load_weights(DtWeights<R>& w, const float* __restrict__ wdt_dir, const float* __restrict__ dtb_dir, int C, int cbase) {
    constexpr float WL2E = AMP ? 1.f : XP_L2E;
    const int fr = threadIdx.x & 31, fh = (threadIdx.x >> 5) & 1;
    w.bias_lane = dtb_dir[cbase + (threadIdx.x & 63)];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int cj = cbase + 32 * j + fr;
        w.bias[j] = AMP ? 0.f : XP_L2E * dtb_dir[cj];
#pragma unroll
        for (int ks = 0; ks < DtWeights<R>::KS; ++ks) {
            float wv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int r = ks * 16 + fh * 8 + e;                     // clamped unconditional load + select: no branch per element
                const float t = wdt_dir[(int64_t)(r < R ? r : 0) * C + cj];
                wv[e] = r < R ? WL2E * t : 0.f;
            }
            dt_split8(wv, w.bw[j][ks][0], w.bw[j][ks][1]);
        }
    }
}
// rows: LDS, tile pixel i's dt_rank values at rows + i * stride (floats, 8-byte aligned)
template <int R, bool AMP = false>
__device__ __forceinline__ void dt_tile(const DtWeights<R>& w, const float* rows, int stride, float (&dtv)[2][16]) {
    const int fr = threadIdx.x & 31, fh = (threadIdx.x >> 5) & 1;
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = w.bias[j];
#pragma unroll
    for (int ks = 0; ks < DtWeights<R>::KS; ++ks) {
        const int k0 = ks * 16 + fh * 8;
        const float* ar = rows + fr * stride + (k0 < R ? k0 : 0);      // a half slab wholly past R reads the row start instead and is cleared
        float v[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float2 t = *reinterpret_cast<const float2*>(ar + 2 * q);
            v[2 * q] = k0 + 2 * q < R ? t.x : 0.f;                      // (R is even: a float2 is inside or outside as a whole)
            v[2 * q + 1] = k0 + 2 * q < R ? t.y : 0.f;
        }
        f16x8_t hi, lo;
        dt_split8(v, hi, lo);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(lo, w.bw[j][ks][0], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hi, w.bw[j][ks][1], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hi, w.bw[j][ks][0], acc[j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[0][e]), __float_as_uint(acc[1][e]), false, false);
        dtv[0][e] = __uint_as_float(sw[0]); dtv[1][e] = __uint_as_float(sw[1]);
        if constexpr (AMP) {
            dtv[0][e] = ((float)(_Float16)dtv[0][e] + w.bias_lane) * XP_L2E;
            dtv[1][e] = ((float)(_Float16)dtv[1][e] + w.bias_lane) * XP_L2E;
        }
    }
}
#define XP_DTV(dtv, j) (dtv)[((j) >> 2) & 1][((j) & 3) + 4 * ((j) >> 3)]

// Shared staging of one block's chunk(s): pixel indices and the xdbl rows of this route pair.
// UT: element type of u / xdbl / out in HBM — float, or _Float16 in the fast mixed-precision class (fp16 storage; every value is converted on load, and the
// conversion on the out_norm store IS the recipe's `y.to(x.dtype)`, VMamba.py:646)
template <int R, typename UT = float>
__device__ __forceinline__ void stage_chunk(const SS2DParams& p, int b, int pair, int chunk0, int* s_pix, int* s_off, float* s_x) {
    const int L = p.H * p.W;
    const int npx = p.cpb * p.T;
    for (int i = threadIdx.x; i < npx; i += blockDim.x) {
        int cl = i / p.T, ii = i - cl * p.T;
        int chunk = chunk0 + cl;
        const int px = (chunk < p.nc) ? pixel_of(chunk * p.T + ii, L, p.H, p.W, pair == 1) : -1;
        s_pix[i] = px;
        s_off[i] = px >= 0 ? px * p.C : -1;      // element offset of the pixel's row: the step loops address u / ya / out with it (no v_mul_lo_u32 per step)
    }
    __syncthreads();
    constexpr int XW = 2 * (R + 2);
    const int XD = 2 * XW;
    for (int i = threadIdx.x; i < npx * XW; i += blockDim.x) {
        int pi = i / XW, e = i - pi * XW;
        int px = s_pix[pi];
        s_x[i] = (px >= 0) ? (float)reinterpret_cast<const UT*>(p.xdbl)[((int64_t)b * L + px) * XD + pair * XW + e] : 0.f;
    }
    __syncthreads();
}

// FULL: every chunk of every block is complete (L % T == 0 and nc % cpb == 0 — all shapes of the 480 x 640 model): the per-step
// validity selects (compare + two cndmask per step) are compiled out
template <int R, bool FULL, bool AMP = false, typename UT = float>
__global__ __launch_bounds__(768) void ss2d_pass1(SS2DParams p) {
    constexpr float WL2E = AMP ? 1.f : XP_L2E;        // factor folded into the dt weights and bias where they are loaded (see step_vals)
    extern __shared__ __align__(16) char smem[];
    const int npx = p.cpb * p.T;
    int* s_pix = reinterpret_cast<int*>(smem);
    int* s_off = s_pix + npx;
    float* s_x = reinterpret_cast<float*>(smem + sizeof(int) * 2 * npx);
    constexpr int XW = 2 * (R + 2);
    const int b = blockIdx.y, pair = blockIdx.z, chunk0 = blockIdx.x * p.cpb;
    stage_chunk<R, UT>(p, b, pair, chunk0, s_pix, s_off, s_x);
    const int cl = threadIdx.x / p.C, c = threadIdx.x - cl * p.C;
    const int chunk = chunk0 + cl;
    if (chunk >= p.nc) return;
    const int L = p.H * p.W;
    const UT* ub = reinterpret_cast<const UT*>(p.u) + (int64_t)b * L * p.C + c;
    float P0 = 1.f, S0 = 0.f, Q1 = 1.f, S1 = 0.f;
    if constexpr (R < 16) {
        // small dt_rank (the wide, HBM-bound stages): both routes in one sweep over u
        float w0[R], w1[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            w0[r] = WL2E * p.wdt[((int64_t)(pair * 2 + 0) * R + r) * p.C + c];
            w1[r] = WL2E * p.wdt[((int64_t)(pair * 2 + 1) * R + r) * p.C + c];
        }
        const float b0 = WL2E * p.dtb[(pair * 2 + 0) * p.C + c], b1 = WL2E * p.dtb[(pair * 2 + 1) * p.C + c];
        const float A0 = XP_L2E * p.A[(pair * 2 + 0) * p.C + c], A1 = XP_L2E * p.A[(pair * 2 + 1) * p.C + c];
        for (int i0 = 0; i0 < p.T; i0 += 4) {
            float uv[4];
            int px[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                px[k] = s_off[cl * p.T + i0 + k];                       // element offset of the pixel row, -1 past the end
                const float t = (XP_SS2D_DBG & 8) ? 1.f : (float)ub[(FULL || px[k] >= 0) ? px[k] : 0];   // 32-bit offsets (host checks B*L*C < 2^31)
                uv[k] = (FULL || px[k] >= 0) ? t : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // Pixels past the end of the last chunk run as u = 0 steps (their xdbl row is zero-filled): b = 0, and their
                // decay factor only reaches states that nobody reads (forward: after the last pixel; backward: multiplied
                // into the zero initial state), so no branch is needed.
                const float* xr = s_x + (cl * p.T + i0 + k) * XW;
                float a, bb, cv;
                step_vals<R, AMP>(xr, w0, b0, A0, uv[k], a, bb, cv);
                S0 = a * S0 + bb; P0 *= a;
                step_vals<R, AMP>(xr + (R + 2), w1, b1, A1, uv[k], a, bb, cv);
                S1 = fmaf(Q1, bb, S1); Q1 *= a;     // backward route accumulated in forward order
            }
        }
    } else {
    // large dt_rank: the two routes run one after the other so that only one route's R projection weights are live at a time: with
    // both (2R = 96 registers at C = 768) a 768-thread workgroup fills the CU's register file by itself and every
    // load latency is exposed.  u is re-read for the second route (L1 / L2 hits; these stages are small).
#pragma unroll 1
    for (int route = 0; route < 2; ++route) {
        float w[R];
#pragma unroll
        for (int r = 0; r < R; ++r) w[r] = WL2E * p.wdt[((int64_t)(pair * 2 + route) * R + r) * p.C + c];
        const float bias = WL2E * p.dtb[(pair * 2 + route) * p.C + c], Av = XP_L2E * p.A[(pair * 2 + route) * p.C + c];
        float Pr = 1.f, Sr = 0.f;
        for (int i0 = 0; i0 < p.T; i0 += 4) {
            float uv[4];
            int px[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                px[k] = s_off[cl * p.T + i0 + k];                       // element offset of the pixel row, -1 past the end
                const float t = (XP_SS2D_DBG & 8) ? 1.f : (float)ub[(FULL || px[k] >= 0) ? px[k] : 0];   // 32-bit offsets (host checks B*L*C < 2^31)
                uv[k] = (FULL || px[k] >= 0) ? t : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // Pixels past the end of the last chunk run as u = 0 steps (their xdbl row is zero-filled): b = 0, and their
                // decay factor only reaches states that nobody reads (forward: after the last pixel; backward: multiplied
                // into the zero initial state), so no branch is needed.
                const float* xr = s_x + (cl * p.T + i0 + k) * XW + route * (R + 2);
                float a, bb, cv;
                step_vals<R, AMP>(xr, w, bias, Av, uv[k], a, bb, cv);
                if (route == 0) Sr = a * Sr + bb;
                else Sr = fmaf(Pr, bb, Sr);                 // backward route accumulated in forward order
                Pr *= a;
            }
        }
        if (route == 0) { P0 = Pr; S0 = Sr; } else { Q1 = Pr; S1 = Sr; }
    }
    }
    const int64_t o = ((((int64_t)b * 2 + pair) * p.nc + chunk) * 2) * p.C + c;
    p.wsP[o] = P0; p.wsS[o] = S0;
    p.wsP[o + p.C] = Q1; p.wsS[o + p.C] = S1;
}

// pass 2: carry over chunks, per (image, pair, direction, channel).  Two-level so that the sequential depth is
// ~2*nc/G + G instead of nc: G groups of consecutive chunks are composed in parallel, a short serial pass gives the
// state entering each group, then every group re-walks its chunks storing the state entering each chunk (over S).
#ifndef XP_SS2D_P2G
#define XP_SS2D_P2G 16
#endif
constexpr int P2_G = XP_SS2D_P2G;      // (-D override: tools/instep_ab.sh)
__global__ __launch_bounds__(64 * P2_G) void ss2d_pass2(SS2DParams p) {
    __shared__ float s_P[P2_G][64], s_S[P2_G][64];
    const int lane = threadIdx.x, g = threadIdx.y;
    const int c = blockIdx.x * 64 + lane;
    const int bp = blockIdx.y >> 1, dir = blockIdx.y & 1;    // bp = b*2 + pair
    const bool ok = c < p.C;
    const int per = (p.nc + P2_G - 1) / P2_G;
    const int j0 = g * per, j1 = min(p.nc, j0 + per);
    auto off = [&](int jj) -> int64_t {
        const int j = dir ? (p.nc - 1 - jj) : jj;
        return (((int64_t)bp * p.nc + j) * 2 + dir) * p.C + c;
    };
    float P = 1.f, S = 0.f;
    if (ok) {
        int jj = j0;
        for (; jj + 4 <= j1; jj += 4) {
            float Pv[4], Sv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int64_t o = off(jj + k); Pv[k] = p.wsP[o]; Sv[k] = p.wsS[o]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) { S = fmaf(Pv[k], S, Sv[k]); P *= Pv[k]; }
        }
        for (; jj < j1; ++jj) { const int64_t o = off(jj); const float Pj = p.wsP[o], Sj = p.wsS[o]; S = fmaf(Pj, S, Sj); P *= Pj; }
    }
    s_P[g][lane] = P; s_S[g][lane] = S;
    __syncthreads();
    float h = 0.f;
    for (int gg = 0; gg < g; ++gg) h = fmaf(s_P[gg][lane], h, s_S[gg][lane]);
    if (!ok) return;
    int jj = j0;
    for (; jj + 4 <= j1; jj += 4) {
        float Pv[4], Sv[4];
        int64_t ov[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { ov[k] = off(jj + k); Pv[k] = p.wsP[ov[k]]; Sv[k] = p.wsS[ov[k]]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) { p.wsS[ov[k]] = h; h = fmaf(Pv[k], h, Sv[k]); }
    }
    for (; jj < j1; ++jj) { const int64_t o = off(jj); const float Pj = p.wsP[o], Sj = p.wsS[o]; p.wsS[o] = h; h = fmaf(Pj, h, Sj); }
}

template <int R, bool COLPAIR, bool FULL, bool AMP = false, typename UT = float>
__global__ __launch_bounds__(768) void ss2d_pass3(SS2DParams p) {
    constexpr float WL2E = AMP ? 1.f : XP_L2E;
    extern __shared__ __align__(16) char smem[];
    const int npx = p.cpb * p.T;
    int* s_pix = reinterpret_cast<int*>(smem);
    int* s_off = s_pix + npx;
    float* s_x = reinterpret_cast<float*>(smem + sizeof(int) * 2 * npx);
    constexpr int XW = 2 * (R + 2);
    float* s_y = s_x + npx * XW;   // [npx][C + 8]: the pad keeps the 16-lanes-per-pixel LayerNorm reads of 4 rows off the same banks
    const int SY = p.C + 8;
    const int pair = COLPAIR ? 1 : 0;
    const int b = blockIdx.y, chunk0 = blockIdx.x * p.cpb;
    stage_chunk<R, UT>(p, b, pair, chunk0, s_pix, s_off, s_x);
    const int cl = threadIdx.x / p.C, c = threadIdx.x - cl * p.C;
    const int chunk = chunk0 + cl;
    const int L = p.H * p.W;
    if (chunk < p.nc) {
        // one route's projection weights live at a time (see pass 1)
        float w0[R];
#pragma unroll
        for (int r = 0; r < R; ++r) w0[r] = WL2E * p.wdt[((int64_t)(pair * 2 + 0) * R + r) * p.C + c];
        const float b0 = WL2E * p.dtb[(pair * 2 + 0) * p.C + c];
        const float A0 = XP_L2E * p.A[(pair * 2 + 0) * p.C + c];
        const float D0 = p.Dp[(pair * 2 + 0) * p.C + c];
        const UT* ub = reinterpret_cast<const UT*>(p.u) + (int64_t)b * L * p.C + c;
        const int64_t o = ((((int64_t)b * 2 + pair) * p.nc + chunk) * 2) * p.C + c;
        float h = p.wsS[o];
        // forward route
        for (int i0 = 0; i0 < p.T; i0 += 4) {
            float uv[4];
            int px[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                px[k] = s_off[cl * p.T + i0 + k];
                const float t = (XP_SS2D_DBG & 8) ? 1.f : (float)ub[(FULL || px[k] >= 0) ? px[k] : 0];
                uv[k] = (FULL || px[k] >= 0) ? t : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int pi = cl * p.T + i0 + k;
                const float* xr = s_x + pi * XW;
                float a, bb, cv;
                step_vals<R, AMP>(xr, w0, b0, A0, uv[k], a, bb, cv);
                h = a * h + bb;
                s_y[pi * SY + c] = cv * h + D0 * uv[k];   // y = C*h + D*u (csms6s.py:61,67); rows past the end are never read
            }
        }
        // backward route over the same pixels
        asm volatile("" ::: "memory");      // keep the second route's weight loads below the forward loop
        float w1[R];
#pragma unroll
        for (int r = 0; r < R; ++r) w1[r] = WL2E * p.wdt[((int64_t)(pair * 2 + 1) * R + r) * p.C + c];
        const float b1 = WL2E * p.dtb[(pair * 2 + 1) * p.C + c];
        const float A1 = XP_L2E * p.A[(pair * 2 + 1) * p.C + c];
        const float D1 = p.Dp[(pair * 2 + 1) * p.C + c];
        h = p.wsS[o + p.C];
        const float* prev = COLPAIR ? (p.ya + (int64_t)b * L * p.C + c) : nullptr;
        float* dst = COLPAIR ? nullptr : (p.ya + (int64_t)b * L * p.C + c);
        for (int i0 = p.T - 4; i0 >= 0; i0 -= 4) {
            float uv[4], pv[4];
            int px[4];
#pragma unroll
            for (int k = 3; k >= 0; --k) {
                px[k] = s_off[cl * p.T + i0 + k];
                const int po = (FULL || px[k] >= 0) ? px[k] : 0;
                const float t = (XP_SS2D_DBG & 8) ? 1.f : (float)ub[po];
                uv[k] = (FULL || px[k] >= 0) ? t : 0.f;
                if (COLPAIR) pv[k] = (XP_SS2D_DBG & 2) ? 0.f : prev[po];
            }
#pragma unroll
            for (int k = 3; k >= 0; --k) {
                const int pi = cl * p.T + i0 + k;
                const float* xr = s_x + pi * XW + (R + 2);
                float a, bb, cv;
                step_vals<R, AMP>(xr, w1, b1, A1, uv[k], a, bb, cv);
                h = a * h + bb;                                      // u = 0 steps before the image's last pixel keep h = 0
                const float y2 = cv * h + D1 * uv[k];
                const float tot = s_y[pi * SY + c] + y2;           // y_fwd + flip(y_bwd)
                if (COLPAIR) s_y[pi * SY + c] = pv[k] + tot;      // (y0+y2) + (y1+y3)
                else if (FULL || px[k] >= 0) dst[px[k]] = tot;
            }
        }
    }
    if (!COLPAIR) return;
    __syncthreads();
    if (XP_SS2D_DBG & 1) { if (threadIdx.x < 4) p.out[(int64_t)b * L * p.C + blockIdx.x * 4 + threadIdx.x] = s_y[threadIdx.x]; return; }
    // out_norm: LayerNorm over C per pixel (two-pass mean / variance).  Narrow stages (C <= 192): 16 lanes per pixel, four
    // pixels per wave — with a whole wave per pixel two thirds of the lanes idle at C = 96 and the two 6-step shuffle
    // reductions dominate (this part was a third of the column pass at stage 0).  Wide stages: one wave per pixel.
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    if (p.C <= 192) {
        // 8 lanes per pixel, eight pixels per wave; a lane owns the float4 chunks sub + 8 j (C / 32 <= 6 of them): the row is read from LDS
        // once (ds_read_b128), both reductions run on registers, every store instruction writes full 128-byte lines, and the out_norm
        // weights are loaded once per lane.  (The first form — 16 lanes per pixel, scalar elements, three LDS reads of the row and the
        // weights fetched per pixel — was 38 % of the column pass at stage 0.)
        const int sub = lane & 7, grp = lane >> 3;
        const int nj = p.C >> 5;
        const float invC = 1.f / (float)p.C;
        float4 gw[6], gb[6];
#pragma unroll
        for (int j = 0; j < 6; ++j)
            if (j < nj) {
                gw[j] = *reinterpret_cast<const float4*>(p.ln_w + 4 * (sub + 8 * j));
                gb[j] = *reinterpret_cast<const float4*>(p.ln_b + 4 * (sub + 8 * j));
            }
        for (int pi = wave * 8 + grp; pi < npx; pi += nw * 8) {      // an 8-lane group skips as a whole: the shuffles stay inside the group
            const int px = s_pix[pi];
            if (px < 0) continue;
            const float* row = s_y + pi * SY;
            float4 r[6];
            float sm = 0.f;
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (j < nj) { r[j] = *reinterpret_cast<const float4*>(row + 4 * (sub + 8 * j)); sm += (r[j].x + r[j].y) + (r[j].z + r[j].w); }
            sm = xp_row8_sum(sm);
            const float mean = sm * invC;
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (j < nj) {
                    const float dx = r[j].x - mean, dy = r[j].y - mean, dz = r[j].z - mean, dw = r[j].w - mean;
                    v = fmaf(dx, dx, v); v = fmaf(dy, dy, v); v = fmaf(dz, dz, v); v = fmaf(dw, dw, v);
                }
            v = xp_row8_sum(v);
            const float rstd = 1.f / sqrtf(v * invC + p.eps);
            UT* orow = reinterpret_cast<UT*>(p.out) + ((int64_t)b * L + px) * p.C;
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (j < nj) {
                    float4 o;
                    o.x = (r[j].x - mean) * rstd * gw[j].x + gb[j].x; o.y = (r[j].y - mean) * rstd * gw[j].y + gb[j].y;
                    o.z = (r[j].z - mean) * rstd * gw[j].z + gb[j].z; o.w = (r[j].w - mean) * rstd * gw[j].w + gb[j].w;
                    if constexpr (std::is_same<UT, float>::value) *reinterpret_cast<float4*>(orow + 4 * (sub + 8 * j)) = o;
                    else {
                        union { _Float16 h[4]; uint2 u; } hh;
                        hh.h[0] = (_Float16)o.x; hh.h[1] = (_Float16)o.y; hh.h[2] = (_Float16)o.z; hh.h[3] = (_Float16)o.w;
                        *reinterpret_cast<uint2*>(orow + 4 * (sub + 8 * j)) = hh.u;
                    }
                }
        }
        return;
    }
    for (int pi = wave; pi < npx; pi += nw) {
        const int px = s_pix[pi];
        if (px < 0) continue;
        const float* row = s_y + pi * SY;
        float s = 0.f;
        for (int cc = lane; cc < p.C; cc += 64) s += row[cc];
        const float mean = xp_wave_sum(s) / (float)p.C;
        float v = 0.f;
        for (int cc = lane; cc < p.C; cc += 64) { float d = row[cc] - mean; v = fmaf(d, d, v); }
        const float rstd = 1.f / sqrtf(xp_wave_sum(v) / (float)p.C + p.eps);
        UT* orow = reinterpret_cast<UT*>(p.out) + ((int64_t)b * L + px) * p.C;
        for (int cc = lane; cc < p.C; cc += 64) orow[cc] = (UT)((row[cc] - mean) * rstd * p.ln_w[cc] + p.ln_b[cc]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Sequential form for the deepest stage (short sequences, many channels: L = 300 x C = 768 at 480 x 640; measured slower than the
// chunked form at L = 1200 x C = 384, where 1200 dependent steps of one wave per SIMD outlast the three passes).  With 3072-float LDS tiles the chunks above shrink to 8 pixels there, every step is evaluated twice (summary
// pass + re-run) and the three passes cost as much as at stage 1 for a half / a quarter of the steps.  Here one wave
// walks ONE route of 64 channels of one image from end to end — every step evaluated once, no carry pass — and writes the
// route's y; a second kernel adds the four routes in the reference's order (y0 + y2) + (y1 + y3) (csm_triton.py:60-62) and applies
// out_norm.  batch * 4 * C / 64 independent waves (768 at both deep stages of a 16-image batch); u and the route's xdbl rows
// are fetched a 32-pixel tile ahead.
// ---------------------------------------------------------------------------------------------------------------
constexpr int SEQ_TP = 32;
template <int R>
__global__ __launch_bounds__(64) void ss2d_seq_scan(SS2DParams p, float* __restrict__ ys) {
    constexpr int RW = R + 2, HW2 = RW / 2;                             // floats / float2 pieces per xdbl row of one route
    static_assert(HW2 <= 32, "two rows per staging instruction");
    __shared__ __align__(16) float s_x[2][SEQ_TP * RW];
    const int lane = threadIdx.x, d = blockIdx.y, b = blockIdx.z;
    const int pair = d >> 1, back = d & 1;                              // directions in the stored order (0, 2, 1, 3)
    const int c = blockIdx.x * 64 + lane;                               // C % 64 == 0 (host)
    const int L = p.H * p.W, XD = 4 * RW;
    float w[R];
#pragma unroll
    for (int r = 0; r < R; ++r) w[r] = XP_L2E * p.wdt[((int64_t)d * R + r) * p.C + c];
    const float bias = XP_L2E * p.dtb[d * p.C + c], Av = XP_L2E * p.A[d * p.C + c], Dv = p.Dp[d * p.C + c];
    const float* ub = p.u + (int64_t)b * L * p.C + c;
    const float* xb = p.xdbl + (int64_t)b * L * XD + pair * 2 * RW + back * RW + 2 * (lane & 31);
    float* yb = ys + ((int64_t)d * p.Bn + b) * L * p.C + c;
    // sequence index i -> pixel: forward routes walk l = i, backward routes l = L - 1 - i (the flipped sequence, csm_triton.py:33-36);
    // column-major routes need (w, h) = (l / H, l % H): one division per tile, then stepping (a division per step and use —
    // three per step — was a third of this kernel).  Indices past the end repeat the last pixel (loaded, never stored).
    auto tile_pixels = [&](int t, int (&px)[SEQ_TP]) {
        const int i0 = t * SEQ_TP;
        if (pair == 0) {
#pragma unroll
            for (int j = 0; j < SEQ_TP; ++j) { const int i = min(i0 + j, L - 1); px[j] = back ? L - 1 - i : i; }
        } else {
            const int l0 = back ? L - 1 - i0 : i0;
            int ww = l0 / p.H, hh = l0 - ww * p.H;
#pragma unroll
            for (int j = 0; j < SEQ_TP; ++j) {
                px[j] = hh * p.W + ww;
                if (i0 + j < L - 1) {
                    if (back) { if (--hh < 0) { hh = p.H - 1; --ww; } }
                    else { if (++hh == p.H) { hh = 0; ++ww; } }
                }
            }
        }
    };
    const int ntile = (L + SEQ_TP - 1) / SEQ_TP;
    float ucur[SEQ_TP], unext[SEQ_TP];
    int pxc[SEQ_TP], pxn[SEQ_TP];          // wave-uniform
    float2 xreg[SEQ_TP / 2];
    auto load_u = [&](const int (&px)[SEQ_TP], float (&uv)[SEQ_TP]) {
#pragma unroll
        for (int j = 0; j < SEQ_TP; ++j) uv[j] = ub[px[j] * p.C];      // 32-bit offsets (host checks H*W*C < 2^31)
    };
    // two xdbl rows per instruction: lanes 0..31 row 2q, lanes 32..63 row 2q+1, float2 piece lane & 31 (lanes past the row
    // length load a valid neighbour and are not stored)
    const bool xok = (lane & 31) < HW2;
    auto load_x = [&](const int (&px)[SEQ_TP]) {
#pragma unroll
        for (int q = 0; q < SEQ_TP / 2; ++q) {
            const int pj = (lane >> 5) ? px[2 * q + 1] : px[2 * q];
            xreg[q] = *reinterpret_cast<const float2*>(xb + (xok ? (int64_t)pj * XD : 0));
        }
    };
    const int xdst = (lane >> 5) * RW + 2 * (lane & 31);
    auto store_x = [&](int buf) {
        if (xok) {
#pragma unroll
            for (int q = 0; q < SEQ_TP / 2; ++q) *reinterpret_cast<float2*>(&s_x[buf][2 * q * RW + xdst]) = xreg[q];
        }
    };
    tile_pixels(0, pxc);
    load_u(pxc, ucur); load_x(pxc); store_x(0);
    float h = 0.f;
    // four steps at a time: their operand evaluation (projection, softplus, exp) is independent and interleaves; only the
    // h update is a chain.  Full tiles carry no bounds tests; the last, partial tile evaluates on valid LDS rows and masks its stores.
    auto run_tile = [&](const float* sx, int nj, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int j0 = 0; j0 < SEQ_TP; j0 += 4) {
            if (FULL || j0 < nj) {                                      // uniform
                float a[4], bb[4], cv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) step_vals<R>(sx + (j0 + k) * RW, w, bias, Av, ucur[j0 + k], a[k], bb[k], cv[k]);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    h = a[k] * h + bb[k];
                    if (FULL || j0 + k < nj) yb[pxc[j0 + k] * p.C] = cv[k] * h + Dv * ucur[j0 + k];   // y = C*h + D*u (csms6s.py:61,67)
                }
            }
        }
    };
    for (int t = 0; t < ntile; ++t) {
        if (t + 1 < ntile) { tile_pixels(t + 1, pxn); load_u(pxn, unext); load_x(pxn); }
        const int nj = L - t * SEQ_TP;
        if (nj >= SEQ_TP) run_tile(s_x[t & 1], SEQ_TP, std::true_type{});
        else run_tile(s_x[t & 1], nj, std::false_type{});
        if (t + 1 < ntile) {
            store_x((t + 1) & 1);
#pragma unroll
            for (int j = 0; j < SEQ_TP; ++j) { ucur[j] = unext[j]; pxc[j] = pxn[j]; }
        }
    }
}

// Sequential form for dt_rank >= 16 (the deep stages), written for INSTRUCTION COUNT: with one wave per SIMD every instruction of any
// kind — scalar address arithmetic, branches, LDS reads — is issue time, and the first kernel above spends ~3 500 of them per 32-pixel
// tile (64-bit scalar address chains per access, a branchy scalar walk of the pixel order, R FMAs per step).  Here
//  * the dt projection of a tile runs on the matrix pipe:  dt[pixel][channel] = bias[channel] + sum_r x[pixel][r] Wdt[channel][r]  as two
//    32x32 MFMA tiles (64 channels) x ceil(R / 16) slabs, both operands split into two fp16 planes, three products per slab
//    (gemm_h2_core.h: operand error <= 2^-23 — f32-grade); the weight fragments stay in registers (as many as the R scalar weights they
//    replace), the pixel rows are split from the LDS tile.  The accumulator layout (lane = column = channel, 16 registers = rows
//    {0-3, 8-11, 16-19, 24-27} + 4 (lane >> 5)) becomes "lane = channel, 32 registers = the tile's 32 pixels" with one
//    v_permlane32_swap per register pair of the two tiles — the scan's own layout;
//  * a tile's pixel offsets are computed once by 32 lanes (vector arithmetic, one reciprocal multiply for the column-major routes) and
//    reach the memory instructions as the SCALAR offset of a buffer access (v_readlane + buffer_load / buffer_store: two instructions
//    per access); pixels past the end get an out-of-range offset, which the buffer hardware turns into a dropped store / a zero load,
//    so there is one code path for full and partial tiles;
//  * the tile's xdbl rows are fetched one row per lane pair with immediate offsets (no address arithmetic).
template <int R, bool AMP = false>
__global__ __launch_bounds__(64) void ss2d_seq_scan2(SS2DParams p, float* __restrict__ ys) {
    constexpr int RW = R + 2, NP = RW / 2, NPL = (NP + 1) / 2;          // float2 pieces per xdbl row of one route; pieces fetched by one lane
    __shared__ __align__(16) float s_x[2][SEQ_TP * RW];
    const int lane = threadIdx.x, d = blockIdx.y, b = blockIdx.z;
    const int pair = d >> 1, back = d & 1;                              // directions in the stored order (0, 2, 1, 3)
    const int fr = lane & 31, fh = lane >> 5;
    const int L = p.H * p.W, XD = 4 * RW;
    const int c = blockIdx.x * 64 + lane;                               // C % 64 == 0 (host)
    DtWeights<R> dw;
    dt_load_weights<R, AMP>(dw, p.wdt + (int64_t)d * R * p.C, p.dtb + d * p.C, p.C, blockIdx.x * 64);
    const float Av = XP_L2E * p.A[d * p.C + c], Dv = p.Dp[d * p.C + c];
    const int plane_bytes = L * p.C * 4;                                // host: < 2^31
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc((void*)(p.u + (int64_t)b * L * p.C), 0, plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(ys + ((int64_t)d * p.Bn + b) * L * p.C), 0, plane_bytes, 0x00020000);
    const int cbyte = c * 4;
    const char* xb = reinterpret_cast<const char*>(p.xdbl + (int64_t)b * L * XD + pair * 2 * RW + back * RW) + fh * (NPL * 8);
    const float invH = 1.f / (float)p.H;
    // tile t, pixel (lane & 31) of the route's sequence: byte offset of its row in a (L, C) plane (out of range past the end) and of its xdbl row
    auto tile_offsets = [&](int t, int& off_uy, int& off_x) {
        const int i = t * SEQ_TP + fr;
        const int ic = min(i, L - 1);
        const int l = back ? L - 1 - ic : ic;                           // backward routes walk the flipped sequence (csm_triton.py:33-36)
        int px = l;
        if (pair == 1) {                                                // column-major routes: l = w * H + h
            const int ww = (int)(((float)l + 0.5f) * invH);             // exact: (l + 0.5) / H is at least 0.5 / H away from an integer
            px = (l - ww * p.H) * p.W + ww;
        }
        off_uy = i < L ? px * (p.C * 4) : 0x7fffffff;
        off_x = px * (XD * 4);
    };
    float ucur[SEQ_TP], unext[SEQ_TP];
    float2 xreg[NPL];
    int offc, offn, offx;
    auto load_tile = [&](int off_uy, int off_x, float (&uv)[SEQ_TP]) {
#pragma unroll
        for (int j = 0; j < SEQ_TP; ++j)
            uv[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ru, cbyte, __builtin_amdgcn_readlane(off_uy, j), 0));
        const char* xr = xb + off_x;
#pragma unroll
        for (int e = 0; e < NPL; ++e) {
            // the upper half-wave fetches pieces NPL .. NP-1; when NP is odd its last slot repeats piece NP-1 (same data, same LDS address)
            const int pe = (e == NPL - 1 && 2 * NPL > NP) ? (fh ? e - 1 : e) : e;
            xreg[e] = *reinterpret_cast<const float2*>(xr + pe * 8);
        }
    };
    const int xdst = fr * RW + fh * (NPL * 2);
    auto store_x = [&](int buf) {
#pragma unroll
        for (int e = 0; e < NPL; ++e) {
            const int pe = (e == NPL - 1 && 2 * NPL > NP) ? (fh ? e - 1 : e) : e;
            *reinterpret_cast<float2*>(&s_x[buf][xdst + pe * 2]) = xreg[e];
        }
    };
    const int ntile = (L + SEQ_TP - 1) / SEQ_TP;
    tile_offsets(0, offc, offx);
    load_tile(offc, offx, ucur);
    store_x(0);
    float h = 0.f;
    for (int t = 0; t < ntile; ++t) {
        if (t + 1 < ntile) { tile_offsets(t + 1, offn, offx); load_tile(offn, offx, unext); }
        const float* sx = s_x[t & 1];
        float dtv[2][16];
        dt_tile<R, AMP>(dw, sx, RW, dtv);                               // dt of the tile on the matrix pipe
        // four steps at a time: their operand evaluation (softplus, exp) is independent and interleaves; only the h update is a chain
#pragma unroll
        for (int j0 = 0; j0 < SEQ_TP; j0 += 4) {
            float a[4], bb[4], cv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float2 bc = *reinterpret_cast<const float2*>(sx + (j0 + k) * RW + R);
                float delta;
                xp_softplus_decay_l2_nb(XP_DTV(dtv, j0 + k), Av, delta, a[k]);
                bb[k] = delta * bc.x * ucur[j0 + k];
                cv[k] = bc.y;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                h = a[k] * h + bb[k];
                const float y = cv[k] * h + Dv * ucur[j0 + k];          // y = C*h + D*u (csms6s.py:61,67)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), ry, cbyte, __builtin_amdgcn_readlane(offc, j0 + k), 0);
            }
        }
        if (t + 1 < ntile) {
            store_x((t + 1) & 1);
#pragma unroll
            for (int j = 0; j < SEQ_TP; ++j) ucur[j] = unext[j];
            offc = offn;
        }
    }
}

// The same sequential recurrence PIPELINED OVER THE WAVES OF A WORKGROUP (round 4).  ss2d_seq_scan2 is one wave per (image, route, 64 channels): its time is the
// route's length — ~100 instructions per step of issue latency, 117 us for L = 1 200 whatever the batch — although only two of those instructions (h = a h + b,
// y = C h + D u) form the chain; the projection, softplus and exp of a step depend on nothing before it.  Here NW waves share the route: wave w takes the tiles
// w, w + NW, ... — loads, matrix-pipe dt projection and the (a, b, C, D u) of all 32 steps of its tile into registers, in parallel with the other waves' tiles —
// then waits for its turn (an LDS counter), takes the running state from LDS, runs the 32 chain steps + stores, and passes the state on.  The chain itself is the
// same instruction sequence in the same order (h = a * h + b; y = C * h + D u with the same operands), so the results are bit-identical to ss2d_seq_scan2;
// what changes is that the chain's length, not the evaluation's, sets the time.  All NW waves of a workgroup are resident together, so the turn wait cannot
// deadlock; a waiting wave sleeps (s_sleep) on a SIMD nobody else of the workgroup uses.
template <int R, bool AMP, int NW>
__global__ __launch_bounds__(64 * NW) void ss2d_seq_scan3(SS2DParams p, float* __restrict__ ys) {
    constexpr int RW = R + 2, NP = RW / 2, NPL = (NP + 1) / 2;
    __shared__ __align__(16) float s_x[NW][2][SEQ_TP * RW];
    __shared__ float s_h[64];
    __shared__ int s_turn;
    const int lane = threadIdx.x & 63, d = blockIdx.y, b = blockIdx.z;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = d >> 1, back = d & 1;
    const int fr = lane & 31, fh = lane >> 5;
    const int L = p.H * p.W, XD = 4 * RW;
    const int c = blockIdx.x * 64 + lane;
    if (threadIdx.x < 64) s_h[lane] = 0.f;
    if (threadIdx.x == 0) s_turn = 0;
    __syncthreads();
    DtWeights<R> dw;
    dt_load_weights<R, AMP>(dw, p.wdt + (int64_t)d * R * p.C, p.dtb + d * p.C, p.C, blockIdx.x * 64);
    const float Av = XP_L2E * p.A[d * p.C + c], Dv = p.Dp[d * p.C + c];
    const int plane_bytes = L * p.C * 4;
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc((void*)(p.u + (int64_t)b * L * p.C), 0, plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(ys + ((int64_t)d * p.Bn + b) * L * p.C), 0, plane_bytes, 0x00020000);
    const int cbyte = c * 4;
    const char* xb = reinterpret_cast<const char*>(p.xdbl + (int64_t)b * L * XD + pair * 2 * RW + back * RW) + fh * (NPL * 8);
    const float invH = 1.f / (float)p.H;
    auto tile_offsets = [&](int t, int& off_uy, int& off_x) {           // as ss2d_seq_scan2
        const int i = t * SEQ_TP + fr;
        const int ic = min(i, L - 1);
        const int l = back ? L - 1 - ic : ic;
        int px = l;
        if (pair == 1) {
            const int ww = (int)(((float)l + 0.5f) * invH);
            px = (l - ww * p.H) * p.W + ww;
        }
        off_uy = i < L ? px * (p.C * 4) : 0x7fffffff;
        off_x = px * (XD * 4);
    };
    float ucur[SEQ_TP], unext[SEQ_TP];
    float2 xreg[NPL];
    int offc = 0, offn = 0, offx = 0;
    auto load_tile = [&](int off_uy, int off_x, float (&uv)[SEQ_TP]) {
#pragma unroll
        for (int j = 0; j < SEQ_TP; ++j)
            uv[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ru, cbyte, __builtin_amdgcn_readlane(off_uy, j), 0));
        const char* xr = xb + off_x;
#pragma unroll
        for (int e = 0; e < NPL; ++e) {
            const int pe = (e == NPL - 1 && 2 * NPL > NP) ? (fh ? e - 1 : e) : e;
            xreg[e] = *reinterpret_cast<const float2*>(xr + pe * 8);
        }
    };
    const int xdst = fr * RW + fh * (NPL * 2);
    auto store_x = [&](int buf) {
#pragma unroll
        for (int e = 0; e < NPL; ++e) {
            const int pe = (e == NPL - 1 && 2 * NPL > NP) ? (fh ? e - 1 : e) : e;
            *reinterpret_cast<float2*>(&s_x[wave][buf][xdst + pe * 2]) = xreg[e];
        }
    };
    const int ntile = (L + SEQ_TP - 1) / SEQ_TP;
    if (wave < ntile) {
        tile_offsets(wave, offc, offx);
        load_tile(offc, offx, ucur);
        store_x(0);
    }
    int it = 0;
    for (int t = wave; t < ntile; t += NW, ++it) {
        if (t + NW < ntile) { tile_offsets(t + NW, offn, offx); load_tile(offn, offx, unext); }
        const float* sx = s_x[wave][it & 1];                            // (written and read by this wave only: LDS operations of a wave execute in order)
        float dtv[2][16];
        dt_tile<R, AMP>(dw, sx, RW, dtv);
        float a[SEQ_TP], bb[SEQ_TP], cv[SEQ_TP];
#pragma unroll
        for (int j = 0; j < SEQ_TP; ++j) {
            const float2 bc = *reinterpret_cast<const float2*>(sx + j * RW + R);
            float delta;
            xp_softplus_decay_l2_nb(XP_DTV(dtv, j), Av, delta, a[j]);
            bb[j] = delta * bc.x * ucur[j];
            cv[j] = bc.y;
            ucur[j] = Dv * ucur[j];                                     // D u: the second product of y = C h + D u, rounded as before
        }
        // (pin the tile's operand evaluation BEFORE the turn wait: they are pure arithmetic, which the compiler may otherwise sink past the wait, into the chain)
#pragma unroll
        for (int j = 0; j < SEQ_TP; ++j) asm volatile("" :: "v"(a[j]), "v"(bb[j]), "v"(cv[j]), "v"(ucur[j]));
        // my turn: every tile before t has passed its state on
        // (LDS only, and LDS operations of a wave execute in order: plain volatile accesses + compiler barriers.  An acquire / release pair at workgroup
        //  scope would also drain the wave's outstanding GLOBAL stores and prefetch loads — s_waitcnt vmcnt(0) — into the chain: measured 1.5x instead of 3x)
        while (*reinterpret_cast<volatile int*>(&s_turn) != t) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        float h = *reinterpret_cast<volatile float*>(&s_h[lane]);
#pragma unroll
        for (int j = 0; j < SEQ_TP; ++j) {
            h = a[j] * h + bb[j];
            const float y = cv[j] * h + ucur[j];                        // y = C*h + D*u (csms6s.py:61,67)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), ry, cbyte, __builtin_amdgcn_readlane(offc, j), 0);
        }
        *reinterpret_cast<volatile float*>(&s_h[lane]) = h;
        asm volatile("" ::: "memory");
        *reinterpret_cast<volatile int*>(&s_turn) = t + 1;
        if (t + NW < ntile) {
            store_x((it + 1) & 1);
#pragma unroll
            for (int j = 0; j < SEQ_TP; ++j) ucur[j] = unext[j];
            offc = offn;
        }
    }
}

// out[b][px][:] = LayerNorm_C((y0 + y2) + (y1 + y3)); one wave per pixel, the row held in registers (C <= 768)
template <typename OT = float>
__global__ __launch_bounds__(256) void ss2d_seq_merge_ln(SS2DParams p, const float* __restrict__ ys) {
    const int lane = threadIdx.x & 63;
    const int64_t M = (int64_t)p.Bn * p.H * p.W;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int64_t plane = M * p.C;
    float v[12];
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 12; ++q) {
        const int cc = q * 64 + lane;
        v[q] = 0.f;
        if (cc < p.C) {
            const float* y = ys + row * p.C + cc;
            v[q] = (y[0] + y[plane]) + (y[2 * plane] + y[3 * plane]);
            s += v[q];
        }
    }
    const float mean = xp_wave_sum(s) / (float)p.C;
    float q2 = 0.f;
#pragma unroll
    for (int q = 0; q < 12; ++q) if (q * 64 + lane < p.C) { const float dd = v[q] - mean; q2 = fmaf(dd, dd, q2); }
    const float rstd = 1.f / sqrtf(xp_wave_sum(q2) / (float)p.C + p.eps);
    if (std::is_same<OT, float>::value && p.out_p32) {
        // the two-plane image the ring GEMM (out_proj) loads by DMA: lane pairs own adjacent channels, so a wave stores whole 64-byte plane segments
        _Float16* prow = reinterpret_cast<_Float16*>(p.out) + row * p.C * 2;
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const int cc = q * 64 + lane;
            if (cc < p.C) {
                const float o = (v[q] - mean) * rstd * p.ln_w[cc] + p.ln_b[cc];
                const _Float16 hi = (_Float16)o, lo = (_Float16)(o - (float)hi);
                _Float16* d = prow + (cc >> 5) * 64 + (cc & 31);
                d[0] = hi; d[32] = lo;
            }
        }
        return;
    }
    OT* orow = reinterpret_cast<OT*>(p.out) + row * p.C;
#pragma unroll
    for (int q = 0; q < 12; ++q) {
        const int cc = q * 64 + lane;
        if (cc < p.C) orow[cc] = (OT)((v[q] - mean) * rstd * p.ln_w[cc] + p.ln_b[cc]);
    }
}

// ss2d_seq_scan2 serves dt_rank >= 16 (whole 8-value fragment halves) while a plane's byte offsets fit the 32-bit buffer offsets
bool seq_scan2_applies(int R, int H, int W, int C) {
    static const bool old_seq = getenv("XP_SS2D_SEQ_V1") && atoi(getenv("XP_SS2D_SEQ_V1")) != 0;      // A/B: first kernel (dt projection on the vector ALU)
    return !old_seq && R >= 16 && R % 8 == 0 && (int64_t)H * W * C < (1ll << 29) && (int64_t)H * W * 4 * (R + 2) < (1ll << 29);
}

// half_out: the fast mixed-precision class — u / xdbl are f32 COPIES of the half tensors (written beside them by their producers: the deep stages are
// small), the dt projection rounds like step_vals<R, AMP>, out is stored as fp16
template <int R>
int launch_ss2d_seq(const SS2DParams& p, float* ys, hipStream_t s, bool half_out = false) {
    const double MC = (double)p.Bn * p.H * p.W * p.C, MX = (double)p.Bn * p.H * p.W * 4 * (R + 2);
    static const bool by_shape = getenv("XP_PROF_SHAPES") != nullptr;
    const std::string sfx = by_shape ? "_C" + std::to_string(p.C) : std::string();
    {   // every route reads u and its quarter of xdbl, writes its y
        XpProfScope prof(("ss2d_seq_scan" + sfx).c_str(), s, 4.0 * MC * (2.0 * R + 14.0), 4.0 * (8.0 * MC + MX));
        const bool v2 = seq_scan2_applies(R, p.H, p.W, p.C);
        if constexpr (R >= 16 && R % 8 == 0) {
            // waves per route (ss2d_seq_scan3): XP_SS2D_SEQ_NW = 1 selects the one-wave kernel (bit-identical results)
            // as many waves per route as the chip has idle SIMDs for: 4 while 4 x routes fit the SIMDs once, else 2, else the one-wave kernel (more waves than
            // SIMDs only add turn waits: 16 images at C = 384: 142 / 99 / 109 us with 1 / 2 / 4; at C = 768: 54 / 62 / 66; 2 images at C = 384: 116 / 70 / 46).
            // Bit-identical for every NW (tools/ss2d_bench.py prints the CRC), so the choice may depend on the batch.
            static const int force_nw = getenv("XP_SS2D_SEQ_NW") ? atoi(getenv("XP_SS2D_SEQ_NW")) : 0;
            static int n_simd = 0;
            if (n_simd == 0) {
                int dev = 0, v = 0;
                if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
                n_simd = 4 * v;
            }
            const dim3 grid(xp_cdiv(p.C, 64), 4, p.Bn);
            const int64_t routes = (int64_t)grid.x * grid.y * grid.z;
            const int nw = force_nw ? force_nw : (routes * 4 <= n_simd ? 4 : (routes * 2 <= n_simd ? 2 : 1));
            if (v2 && nw == 4) {
                if (half_out) hipLaunchKernelGGL((ss2d_seq_scan3<R, true, 4>), grid, dim3(256), 0, s, p, ys);
                else hipLaunchKernelGGL((ss2d_seq_scan3<R, false, 4>), grid, dim3(256), 0, s, p, ys);
            } else if (v2 && nw == 2) {
                if (half_out) hipLaunchKernelGGL((ss2d_seq_scan3<R, true, 2>), grid, dim3(128), 0, s, p, ys);
                else hipLaunchKernelGGL((ss2d_seq_scan3<R, false, 2>), grid, dim3(128), 0, s, p, ys);
            } else if (v2 && half_out) hipLaunchKernelGGL((ss2d_seq_scan2<R, true>), grid, dim3(64), 0, s, p, ys);
            else if (v2) hipLaunchKernelGGL((ss2d_seq_scan2<R, false>), grid, dim3(64), 0, s, p, ys);
        }
        if (!v2) {
            if (half_out) { xp_set_error("xp_ss2d_core_fwd_f16: the sequential form needs dt_rank >= 16 (got %d)", R); return XP_ERR_ARG; }
            hipLaunchKernelGGL(ss2d_seq_scan<R>, dim3(xp_cdiv(p.C, 64), 4, p.Bn), dim3(64), 0, s, p, ys);
        }
    }
    {
        XpProfScope prof(("ss2d_seq_merge_ln" + sfx).c_str(), s, 12.0 * MC, 4.0 * 5.0 * MC);
        if (half_out) hipLaunchKernelGGL(ss2d_seq_merge_ln<_Float16>, dim3(xp_cdiv((int64_t)p.Bn * p.H * p.W, 4)), dim3(256), 0, s, p, ys);
        else hipLaunchKernelGGL(ss2d_seq_merge_ln<float>, dim3(xp_cdiv((int64_t)p.Bn * p.H * p.W, 4)), dim3(256), 0, s, p, ys);
    }
    XP_LAUNCH_CHECK();
    return XP_OK;
}

template <int R>
int launch_ss2d(const SS2DParams& p, hipStream_t s, bool half_io = false) {
    constexpr int XW = 2 * (R + 2);
    const int npx = p.cpb * p.T;
    const int threads = p.cpb * p.C;
    const bool full = (p.H * p.W) % p.T == 0 && p.nc % p.cpb == 0;
    const bool amp = xp_amp_value() != 0;           // mixed-precision class: the general (bounds-tested) instances with the dt rounding of step_vals
    const size_t sm1 = sizeof(int) * 2 * npx + sizeof(float) * npx * XW;
    const size_t sm3 = sm1 + sizeof(float) * (size_t)npx * (p.C + 8);
    dim3 grid1(xp_cdiv(p.nc, p.cpb), p.Bn, 2), grid3(xp_cdiv(p.nc, p.cpb), p.Bn, 1);
    const double MC = (double)p.Bn * p.H * p.W * p.C, MX = (double)p.Bn * p.H * p.W * XW;
    static const bool by_shape = getenv("XP_PROF_SHAPES") != nullptr;
    const std::string sfx = by_shape ? "_C" + std::to_string(p.C) : std::string();
    const double el = 4.0 * MC;   // (pixel, channel, direction) scan elements of the whole core
    {   // reads u + its half of xdbl for each of the two route pairs
        XpProfScope prof(("ss2d_pass1" + sfx).c_str(), s, el * (2.0 * R + 12.0) / 2.0, 4.0 * 2.0 * (MC + MX));
        if (half_io && full) hipLaunchKernelGGL((ss2d_pass1<R, true, true, _Float16>), grid1, dim3(threads), sm1, s, p);
        else if (half_io) hipLaunchKernelGGL((ss2d_pass1<R, false, true, _Float16>), grid1, dim3(threads), sm1, s, p);
        else if (amp) hipLaunchKernelGGL((ss2d_pass1<R, false, true>), grid1, dim3(threads), sm1, s, p);
        else if (full) hipLaunchKernelGGL((ss2d_pass1<R, true>), grid1, dim3(threads), sm1, s, p);
        else hipLaunchKernelGGL((ss2d_pass1<R, false>), grid1, dim3(threads), sm1, s, p);
    }
    {
        XpProfScope prof(("ss2d_pass2" + sfx).c_str(), s, 0.0, 4.0 * 3.0 * (double)p.Bn * 4 * p.nc * p.C);
        hipLaunchKernelGGL(ss2d_pass2, dim3(xp_cdiv(p.C, 64), p.Bn * 4), dim3(64, P2_G), 0, s, p);
    }
    {   // read u, xdbl half; write ya
        XpProfScope prof(("ss2d_pass3_row" + sfx).c_str(), s, el * (2.0 * R + 14.0) / 4.0, 4.0 * (2.0 * MC + MX));
        if (half_io && full) hipLaunchKernelGGL((ss2d_pass3<R, false, true, true, _Float16>), grid3, dim3(threads), sm3, s, p);
        else if (half_io) hipLaunchKernelGGL((ss2d_pass3<R, false, false, true, _Float16>), grid3, dim3(threads), sm3, s, p);
        else if (amp) hipLaunchKernelGGL((ss2d_pass3<R, false, false, true>), grid3, dim3(threads), sm3, s, p);
        else if (full) hipLaunchKernelGGL((ss2d_pass3<R, false, true>), grid3, dim3(threads), sm3, s, p);
        else hipLaunchKernelGGL((ss2d_pass3<R, false, false>), grid3, dim3(threads), sm3, s, p);
    }
    {   // read u, ya, xdbl half; write out (after out_norm)
        XpProfScope prof(("ss2d_pass3_col_ln" + sfx).c_str(), s, el * (2.0 * R + 14.0) / 4.0, 4.0 * (3.0 * MC + MX));
        if (half_io && full) hipLaunchKernelGGL((ss2d_pass3<R, true, true, true, _Float16>), grid3, dim3(threads), sm3, s, p);
        else if (half_io) hipLaunchKernelGGL((ss2d_pass3<R, true, false, true, _Float16>), grid3, dim3(threads), sm3, s, p);
        else if (amp) hipLaunchKernelGGL((ss2d_pass3<R, true, false, true>), grid3, dim3(threads), sm3, s, p);
        else if (full) hipLaunchKernelGGL((ss2d_pass3<R, true, true>), grid3, dim3(threads), sm3, s, p);
        else hipLaunchKernelGGL((ss2d_pass3<R, true, false>), grid3, dim3(threads), sm3, s, p);
    }
    XP_LAUNCH_CHECK();
    return XP_OK;
}

}  // namespace

// -1 (default): chosen per call shape; 0: chunked three-pass form; 1: sequential form.  XP_SS2D_SEQ=0/1 presets it.
static std::atomic<int> g_ss2d_mode{getenv("XP_SS2D_SEQ") ? atoi(getenv("XP_SS2D_SEQ")) : -1};
extern "C" int xp_ss2d_core_set_mode(int mode) {
    XP_CHECK_ARG(mode >= -1 && mode <= 1, "xp_ss2d_core_set_mode: mode must be -1 (auto), 0 (chunked) or 1 (sequential)");
    g_ss2d_mode.store(mode);
    return XP_OK;
}

// THE decision "does this per-image shape take the sequential (deep-stage) form" — the one copy of it: ss2d_core_impl, xp_ss2d_core_p32_supported and
// xp_ss2d_core_f16_wants_f32_copies all call it, so the predicates the host plans with can never disagree with what the launch does.
// Per-image quantities only (never the batch).  half_io: the fp16-storage class, which needs f32 copies of u / xdbl for this form (have_copies).
static bool ss2d_takes_seq(int H, int W, int C, int R, bool half_io, bool have_copies) {
    static const int max_l = getenv("XP_SS2D_SEQ_MAXL") ? atoi(getenv("XP_SS2D_SEQ_MAXL")) : -1;
    if (!(C <= 768 && C % 64 == 0)) return false;
    switch (R) { case 2: case 4: case 6: case 8: case 12: case 16: case 24: case 48: break; default: return false; }
    if (!half_io && xp_amp_value()) return false;          // the f32-container mixed-precision class: chunked form only (its dt rounding lives in step_vals)
    const bool seq2 = seq_scan2_applies(R, H, W, C);
    if (half_io && !(seq2 && have_copies)) return false;
    const int mode = g_ss2d_mode.load();
    return mode >= 0 ? mode != 0 : (seq2 && H * W <= (max_l >= 0 ? max_l : XP_SS2D_SEQ_DEFAULT_MAXL));
}

extern "C" size_t xp_ss2d_core_workspace_bytes(int batch, int H, int W, int C) {
    // P and S: (B, 2, nc, 2, C) each with the smallest chunk (T = 8) -> upper bound; + ya (B,H,W,C)
    const int64_t L = (int64_t)H * W;
    const int64_t nc = (L + 7) / 8;
    const int64_t chunked = 2 * (int64_t)batch * 2 * nc * 2 * C + (int64_t)batch * L * C;
    const int64_t sequential = 4 * (int64_t)batch * L * C;     // y of the four routes (deep stages, ss2d_seq_scan)
    return (size_t)(chunked > sequential ? chunked : sequential) * sizeof(float);
}

static int ss2d_core_impl(const void* u, const void* xdbl, const float* u32, const float* xdbl32, const float* wdt, const float* dt_bias,
                          const float* A, const float* Ds, const float* ln_w, const float* ln_b, void* out,
                          float* workspace, size_t workspace_bytes, int batch, int H, int W, int C, int R,
                          int dstate, float eps, bool half_io, void* stream, int out_p32 = 0) {
    const char* who = half_io ? "xp_ss2d_core_fwd_f16" : "xp_ss2d_core_fwd";
    XP_CHECK_ARG(u && xdbl && wdt && dt_bias && A && Ds && ln_w && ln_b && out && workspace, "%s: null pointer", who);
    XP_CHECK_ARG(dstate == 1, "%s: only d_state == 1 (the XPoint config) is implemented in the fused core; "
                              "use xp_selective_scan_fwd for general d_state (got %d)", who, dstate);
    XP_CHECK_ARG(batch > 0 && H > 0 && W > 0, "%s: bad shape", who);
    XP_CHECK_ARG(C % 32 == 0 && C >= 32 && C <= 768, "%s: C must be a multiple of 32 in [32,768] (got %d)", who, C);
    XP_CHECK_ARG(workspace_bytes >= xp_ss2d_core_workspace_bytes(batch, H, W, C), "%s: workspace too small", who);
    XP_CHECK_ARG((int64_t)H * W * C < (1ll << 31), "%s: one image plane (H*W*C) must stay below 2^31 elements", who);
    SS2DParams p;
    p.u = (const float*)u; p.xdbl = (const float*)xdbl; p.wdt = wdt; p.dtb = dt_bias; p.A = A; p.Dp = Ds; p.ln_w = ln_w; p.ln_b = ln_b;
    p.out = (float*)out; p.out_p32 = out_p32; p.Bn = batch; p.H = H; p.W = W; p.C = C; p.eps = eps;
    // threads = cpb * C must be a multiple of 64 and <= 768
    int cpb = 1;
    static const int blk_threads = getenv("XP_SS2D_THREADS") ? atoi(getenv("XP_SS2D_THREADS")) : 192;      // experiments: threads per workgroup of the chunked passes
    if (C < blk_threads) { cpb = blk_threads / C; while ((cpb * C) % 64) ++cpb; }
    XP_CHECK_ARG((cpb * C) % 64 == 0 && cpb * C <= 768, "%s: unsupported C=%d", who, C);
    p.cpb = cpb;
    // chunk length per LAYER (C only, never the batch).  Round 4 re-measured it in the overlapped step instead of alone: 32 pixels at C <= 96, 16 at C = 192 / 384
    // (round 1-3: 16 / 16 / 8, chosen on the stand-alone core) — alone the core times are equal within 2 % at stages 0 - 1, but half as many chunks are half as
    // many workgroups, carry entries and aggregate stores, and with three encoders in flight that is what the step pays for: 1 674 - 1 677 -> 1 707 - 1 708
    // pairs/s on one box (64 / 32 / 16: 1 662; 64 / 64 / 32: 1 653).  XP_SS2D_TBUDGET / XP_SS2D_T ("96:16,384:8") override for A/B runs.
    static const int t_budget = getenv("XP_SS2D_TBUDGET") ? atoi(getenv("XP_SS2D_TBUDGET")) : 6144;
    int T = t_budget / (cpb * C);
    T = T >= 64 ? 64 : (T >= 32 ? 32 : (T >= 16 ? 16 : 8));
    if (!getenv("XP_SS2D_TBUDGET") && cpb == 1 && T > 16) T = 16;          // C = 192: 16 (32 measured equal in the step, slower alone)
    if (const char* tl = getenv("XP_SS2D_T")) {          // experiments: "96:32,192:16,384:16" = chunk length per channel count
        for (const char* q = tl; q && *q;) {
            int cc = 0, tt = 0;
            if (sscanf(q, "%d:%d", &cc, &tt) == 2 && cc == C && (tt == 8 || tt == 16 || tt == 32 || tt == 64)) T = tt;
            q = strchr(q, ',');
            if (q) ++q;
        }
    }
    p.T = T;
    const int L = H * W;
    p.nc = xp_cdiv(L, T);
    const int64_t nps = (int64_t)batch * 2 * p.nc * 2 * C;
    p.wsP = workspace; p.wsS = workspace + nps; p.ya = workspace + 2 * nps;
    hipStream_t s = (hipStream_t)stream;
    // Deep stages -> the sequential form.  The choice uses per-image quantities only (dt_rank, L), never the batch: the two forms differ
    // by rounding (~1e-7), and a result that changed with the number of images sharing a call would make grouped / split / single-image
    // runs of the same image disagree.  tools/ss2d_bench.py: C = 768, L = 300: 44 / 46 / 56 / 103 us for 2 / 4 / 16 / 32 images against
    // 51 / 65 / 169 / 292 us chunked; C = 768, L = 1024, 8 images: 131 vs 248 us.  C = 384, L = 1200 (stage 2 of a 480 x 640 image): 148 vs 195 us alone at
    // 16 images but 117 vs 47 us at 2 — the sequential form's time is the route's length, whatever the batch.  Rounds 2-3 called it a draw in the step and kept
    // it chunked (bound 512); re-measured in round 4 in the three-stream step it is +1.6 % at 8 pairs per call (1 671-1 673 -> 1 698, three alternations), 0 at
    // 4, -0.9 % at 2 and -4.8 % at ONE pair per call (0.99 -> 1.03 ms per pair).  The bound is now 2 048 for every dt_rank >= 16 shape — the configurations
    // this path is measured on run >= 8 pairs per GPU; the wave-pipelined kernel (ss2d_seq_scan3) then removed the one-pair cost as well (965 -> 1 029 pairs/s,
    // above the chunked form's 1 022), and the bound went to 8 192: stage 2 of a 1024 x 1024 image (L = 4 096) through the pipelined kernel is +2.4 % on
    // config C4 at 4 pairs per call, +2 % at 2, +0.8 % at 1.  xp_ss2d_core_set_mode / XP_SS2D_SEQ force one form.
    // (the f32-container mixed-precision class always takes the chunked form; the fp16-storage class takes the sequential form when the caller supplied f32
    //  copies of u and xdbl — xp_ss2d_core_f16_wants_f32_copies says when it will)
    const bool seq = ss2d_takes_seq(H, W, C, R, half_io, u32 && xdbl32);
    XP_CHECK_ARG(!out_p32 || (seq && !half_io), "%s: a P32 output exists for the sequential form only (xp_ss2d_core_p32_supported)", who);
    if (seq) {
        if (half_io) { p.u = u32; p.xdbl = xdbl32; }
        switch (R) {
            case 2: return launch_ss2d_seq<2>(p, workspace, s, half_io);
            case 4: return launch_ss2d_seq<4>(p, workspace, s, half_io);
            case 6: return launch_ss2d_seq<6>(p, workspace, s, half_io);
            case 8: return launch_ss2d_seq<8>(p, workspace, s, half_io);
            case 12: return launch_ss2d_seq<12>(p, workspace, s, half_io);
            case 16: return launch_ss2d_seq<16>(p, workspace, s, half_io);
            case 24: return launch_ss2d_seq<24>(p, workspace, s, half_io);
            case 48: return launch_ss2d_seq<48>(p, workspace, s, half_io);
            default: break;
        }
    }
    switch (R) {
        case 2: return launch_ss2d<2>(p, s, half_io);
        case 4: return launch_ss2d<4>(p, s, half_io);
        case 6: return launch_ss2d<6>(p, s, half_io);
        case 8: return launch_ss2d<8>(p, s, half_io);
        case 12: return launch_ss2d<12>(p, s, half_io);
        case 16: return launch_ss2d<16>(p, s, half_io);
        case 24: return launch_ss2d<24>(p, s, half_io);
        case 48: return launch_ss2d<48>(p, s, half_io);
        default: xp_set_error("%s: dt_rank %d not instantiated (2,4,6,8,12,16,24,48)", who, R); return XP_ERR_ARG;
    }
}

extern "C" int xp_ss2d_core_fwd(const float* u, const float* xdbl, const float* wdt, const float* dt_bias,
                                const float* A, const float* Ds, const float* ln_w, const float* ln_b, float* out,
                                float* workspace, size_t workspace_bytes, int batch, int H, int W, int C, int R,
                                int dstate, float eps, void* stream) {
    return ss2d_core_impl(u, xdbl, nullptr, nullptr, wdt, dt_bias, A, Ds, ln_w, ln_b, out, workspace, workspace_bytes, batch, H, W, C, R, dstate, eps, false, stream);
}

// out_fmt 0: f32 rows (= xp_ss2d_core_fwd); 2: the P32 image of the result (the operand format of xp_gemm_nt_h2s: out_proj loads it by DMA) — only where the
// sequential form runs (xp_ss2d_core_p32_supported, a per-image predicate)
extern "C" int xp_ss2d_core_p32_supported(int H, int W, int C, int R) { return ss2d_takes_seq(H, W, C, R, false, false) ? 1 : 0; }
extern "C" int xp_ss2d_core_fwd_ex(const float* u, const float* xdbl, const float* wdt, const float* dt_bias,
                                   const float* A, const float* Ds, const float* ln_w, const float* ln_b, void* out, int out_fmt,
                                   float* workspace, size_t workspace_bytes, int batch, int H, int W, int C, int R,
                                   int dstate, float eps, void* stream) {
    XP_CHECK_ARG(out_fmt == 0 || out_fmt == 2, "xp_ss2d_core_fwd_ex: out_fmt must be 0 (f32 rows) or 2 (P32 image)");
    return ss2d_core_impl(u, xdbl, nullptr, nullptr, wdt, dt_bias, A, Ds, ln_w, ln_b, out, workspace, workspace_bytes, batch, H, W, C, R, dstate, eps, false, stream, out_fmt == 2);
}

// Would xp_ss2d_core_fwd_f16 take the sequential (deep-stage) form for this per-image shape if it is given f32 copies of u and xdbl?  (A per-image
// predicate like the f32 class's: never depends on the batch.)
extern "C" int xp_ss2d_core_f16_wants_f32_copies(int H, int W, int C, int R) { return ss2d_takes_seq(H, W, C, R, true, true) ? 1 : 0; }

// The fast mixed-precision class's core: u (batch,H,W,C) and xdbl (batch*H*W, 4*(R+2)) are fp16, out is fp16 (= out_norm's f32 result cast to half,
// VMamba.py:646); the scan state, softplus / exp and out_norm run in f32 (csms6s.py:47-67); the dt projection's output is rounded to fp16 before the f32
// bias (a half convolution under autocast).  u_f32 / xdbl_f32: optional f32 copies of the same tensors — when given and the shape is a deep-stage one
// (xp_ss2d_core_f16_wants_f32_copies) the one-wave-per-route sequential kernels run on them.
extern "C" int xp_ss2d_core_fwd_f16(const void* u, const void* xdbl, const float* u_f32, const float* xdbl_f32, const float* wdt, const float* dt_bias,
                                    const float* A, const float* Ds, const float* ln_w, const float* ln_b, void* out,
                                    float* workspace, size_t workspace_bytes, int batch, int H, int W, int C, int R,
                                    int dstate, float eps, void* stream) {
    return ss2d_core_impl(u, xdbl, u_f32, xdbl_f32, wdt, dt_bias, A, Ds, ln_w, ln_b, out, workspace, workspace_bytes, batch, H, W, C, R, dstate, eps, true, stream);
}
