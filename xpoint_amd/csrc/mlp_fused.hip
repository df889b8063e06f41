// Fused VSS-block MLP for the wide stages:   x <- x + fc2(GELU(fc1(LN(x)) + b1)) + b2     (reference VMamba.py:1230-1234 VSSBlock
// second residual branch, :110-128 Mlp, nn.LayerNorm eps 1e-5, exact-erf GELU), in place, one launch instead of
// layernorm + fc1 GEMM + fc2 GEMM.  The (M, 4C) hidden tensor never exists: not in HBM (472 MB written and read back per
// block at stage 0 of a 16-image 480x640 batch), not in LDS either.
//
// Arithmetic = the split-bf16 engine of gemm_x3_core.h (every f32 operand is the exact sum of three bf16 values, six
// bf16 MFMA partial products per multiply, f32 accumulate), so results agree with the unfused path to f32 rounding.
//
// A wave owns 32 rows of x.  Per 32-wide chunk c of hidden units:
//   fc1   hT[h][m] = b1[h] + sum_k W1[h][k] LN(x)[m][k]   A = W1 fragments from LDS, B = the wave's LN(x) rows, split into
//                                                          planes ONCE and held in registers for the whole kernel
//   GELU  on the accumulator registers, then the 16 values of a lane are split into bf16 planes
//   fc2   out[m][n] += sum_h hid[m][h] W2[n][h]           A = those registers, B = W2 fragments from LDS
// The accumulator of a 32x32x16 MFMA holds a column (here: the row m of x) on the lane and 16 rows in the registers, which
// is exactly an A operand over k = hidden unit: lane-half g, register r <-> row (r&3) + 8(r>>2) + 4g.  Storing the W1 rows
// of a chunk in LDS with bits 2 and 3 of the row index swapped makes that "row" the hidden unit 8g + (r&7) + 16(r>>3), i.e.
// registers 0..7 / 8..15 are two natural 16-wide k slabs and W2 needs no permutation.
//
// Issue order (one wave): the fc1 MFMAs of chunk c+1 with the GELU of chunk c between them, then the fc2 MFMAs of chunk c with the
// bf16 split of its second half between them; two accumulator sets for the hidden chunk alternate.  (Inside ONE wave VALU work
// does not hide under a dependent MFMA chain on this part — tools/coexec_probe.hip — the order only evens out the instruction mix
// that the second resident wave of the SIMD overlaps with.)
//
// The same phases also serve the rest of the block (template MODE): MODE 1 first folds in the SS2D out_proj and the block's first
// residual (x += t W0^T: fc1-type phases with the t rows as the resident operand; the result lands in the lane layout of the loaded
// x rows), MODE 2 is LayerNorm + one projection only (the block's norm + in_proj).  NP = 6 / 3 / 1 partial products per multiply
// (xp_set_dense_products).
//
// Only the weights go through LDS, shared by the 4 waves of a workgroup (128 rows).  xp_mlp_fused_x3_pack lays them out once
// per weight upload as the exact sequence of LDS images the kernel consumes — W1(0), W1(1), W2(0), W1(2), W2(1), ... — each
// 2C rows x 112 B in the padded row format of gemm_x3_core.h (conflict-free ds_read_b128 fragments) with the W1 rows already
// permuted, so an image is one contiguous block and a wave's LDS-DMA (global_load_lds_dwordx4, no staging registers)
// needs a scalar base and lane * 16.  Images land two phases ahead in a 3-slot ring (their pieces issued between the MFMA groups of
// the phase); one counted wait + barrier per phase.
#include <stdlib.h>

#include <string>

#include "gemm_x3_core.h"
#include "gemm_h2_core.h"

#ifndef XP_MLP_FRAG_DEPTH
#define XP_MLP_FRAG_DEPTH 2   /* LDS fragment look-ahead of the split-fp16 instances, in k slabs (1 = round 4) */
#endif
#ifndef XP_MLP_DBG
#define XP_MLP_DBG 0   /* timing experiments only (wrong results): 1 no GELU, 2 no LDS-DMA after the prologue, 4 no barriers, 8 no bf16 split of the hidden values, 16 no MFMA */
#endif

namespace {

struct MlpParams {
    float* X;                 // (M, C) in / out
    const float* T1;          // PRE: (M, C) input of the preceding bias-free projection  X <- X + T1 W0^T  (SS2D out_proj), else null
    float* Out;               // projection-only mode: (M, Nout) = LN(X) W0^T, X read-only
    int Nout;
    const float* ln_w; const float* ln_b;
    const unsigned char* Wpack;   // xp_mlp_fused_x3_pack output
    const float* b1;
    const float* b2;
    int M, H4;
    float eps;
    // split-fp16 ("h2") instances only: inverse power-of-two row scales of the offline weight split (gemm_h2_core.h) —
    // s0 (rows of W0), s1 (H4 rows of fc1.weight), s2 (C rows of fc2.weight); null for the split-bf16 instances
    const float* s0; const float* s1; const float* s2;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int C, bool H2 = false>
struct MlpTile {
    static constexpr int KS = C / 16;            // k slabs of fc1
    static constexpr int NT = C / 32;            // 32-wide output tiles of fc2
    static constexpr int ROWS = 2 * C;           // rows of a chunk image (W1: KS x 32, W2: 2 x C)
    static constexpr int NPL = H2 ? 2 : 3;       // operand planes (h2: two fp16 planes, x3: three bf16 planes), 32 B each per row and slab
    static constexpr int UPR = 2 * NPL + 1;      // 16-byte units per image row incl. its pad unit
    static constexpr int ROWB = UPR * 16;        // 112 B (x3) / 80 B (h2): both conflict-free for ds_read_b128 fragment reads
    static constexpr int UNITS = ROWS * UPR;     // 16-byte units per image
    static constexpr int NI = (UNITS * 16 + 4095) / 4096;     // DMA instructions per wave and image (4 waves x 1 KiB each)
    static constexpr int IMGP = NI * 4096;       // image stride in the packed stream and in the LDS ring (>= UNITS * 16)
    static_assert(C % 32 == 0, "C must be a multiple of 32");
};

__device__ __forceinline__ int mlp_swap23(int r) { return (r & 0x13) | ((r & 4) << 1) | ((r & 8) >> 1); }

// One thread per 16-byte unit of the packed stream (see the header comment for the image order).
// source unit of (weight row n of N, 16-wide k slab s, unit k = plane * 2 + octet) in the offline split layouts
template <bool H2>
__device__ __forceinline__ uint4 mlp_src_unit(const uint4* __restrict__ W, int N, int n, int s, int k) {
    if (H2) return W[((int64_t)(s >> 1) * N + n) * H2_SLAB_UNITS + (k >> 1) * 4 + (s & 1) * 2 + (k & 1)];       // [slab32][n][plane][4 octets]
    return W[((int64_t)s * N + n) * X3_SLAB_UNITS + k];                                                         // [slab16][n][plane][2 octets]
}

template <int C, bool H2>
__global__ void mlp_pack_kernel(const uint4* __restrict__ W1, const uint4* __restrict__ W2, const uint4* __restrict__ W0, uint4* __restrict__ out, int H4,
                                int n0) {      // n0 = rows of W0 (C for out_proj ahead of an MLP; any multiple of 32 for a projection-only stream)
    using T = MlpTile<C, H2>;
    constexpr int UPR = T::UPR, KD = T::UPR - 1;
    const int NC = H4 / 32, NPRE = W0 ? n0 / 32 : 0;
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int upi = T::IMGP / 16;
    if (id >= (int64_t)(NPRE + 2 * NC) * upi) return;
    int n = (int)(id / upi);
    const int u = (int)(id - (int64_t)n * upi);
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    const int row = u / UPR, k = u - row * UPR;
    if (n < NPRE) {          // the (C, C) projection ahead of the MLP, as NT images in the W1 format (32 permuted output rows x all k slabs)
        if (u < T::UNITS && k < KD) {
            const int s = row >> 5, h = 32 * n + mlp_swap23(row & 31);
            v = mlp_src_unit<H2>(W0, n0, h, s, k);
        }
        out[id] = v;
        return;
    }
    n -= NPRE;
    if (u < T::UNITS && k < KD) {
        // x3 (fc1 of chunk c+1 runs ahead of fc2 of chunk c): image n: 0 -> W1(0); odd n < 2NC-1 -> W1((n+1)/2); even n > 0 -> W2(n/2 - 1);
        // n = 2NC-1 -> W2(NC-1).   h2 (one hidden accumulator, chunk after chunk): even n -> W1(n/2), odd n -> W2(n/2).
        const bool is_w1 = H2 ? !(n & 1) : ((n == 0) || ((n & 1) && n < 2 * NC - 1));
        if (is_w1) {
            const int c = H2 ? n >> 1 : (n + 1) >> 1, s = row >> 5, h = 32 * c + mlp_swap23(row & 31);
            v = mlp_src_unit<H2>(W1, H4, h, s, k);
        } else {
            const int c = H2 ? n >> 1 : ((n == 2 * NC - 1) ? NC - 1 : (n >> 1) - 1), j = row / C, nn = row - j * C;
            v = mlp_src_unit<H2>(W2, C, nn, 2 * c + j, k);
        }
    }
    out[id] = v;
}

// GELU(erf) as max(x, 0) - 0.5 |x| erfc(|x| / sqrt 2): the same erfc approximation as xp_gelu_fast (Abramowitz-Stegun 7.1.26 on
// the hardware rcp / exp2 units) without the sign select; the factor 0.5 sqrt 2 is folded into the polynomial.
__device__ __forceinline__ float mlp_gelu(float x) {
    if (XP_MLP_DBG & 1) return x;
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
    constexpr float k = 0.70710678118654752440f;
    float q = fmaf(1.061405429f * k, t, -1.453152027f * k);
    q = fmaf(q, t, 1.421413741f * k);
    q = fmaf(q, t, -0.284496736f * k);
    q = fmaf(q, t, 0.254829592f * k);
    const float e = q * t * __builtin_amdgcn_exp2f(z * z * -1.44269504088896340736f);     // 0.5 sqrt2 erfc(z)
    return fmaf(-z, e, fmaxf(x, 0.f));
}

// MODE 0: MLP branch; 1: out_proj + first residual, then the MLP branch; 2: LayerNorm + one bias-free projection only
// (Out = LN(X) W0^T: the block's norm + in_proj, VMamba.py:1229 / :649), the same row-stationary phases without the MLP.
// 16-bit operand fragments of either engine travel as raw bits; the MFMA is chosen by the engine
typedef unsigned frag_bits __attribute__((ext_vector_type(4)));
template <bool H2>
__device__ __forceinline__ f32x16 mlp_mfma(frag_bits a, frag_bits b, f32x16 c) {
    if constexpr (H2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// eight f32 -> operand planes (x3: three exact bf16 planes; h2: two fp16 planes, operand error <= 2^-23)
template <bool H2>
__device__ __forceinline__ void mlp_split8(const float4& lo, const float4& hi, frag_bits (&pl)[3]) {
    if constexpr (H2) {
        uint2 a0, a1, b0, b1;
        h2_split4(lo, a0, a1); h2_split4(hi, b0, b1);
        pl[0] = frag_bits{a0.x, a0.y, b0.x, b0.y}; pl[1] = frag_bits{a1.x, a1.y, b1.x, b1.y}; pl[2] = pl[0];
    } else {
        uint4 c0, c1, c2;
        xp_split8(lo, hi, c0, c1, c2);
        pl[0] = frag_bits{c0.x, c0.y, c0.z, c0.w}; pl[1] = frag_bits{c1.x, c1.y, c1.z, c1.w}; pl[2] = frag_bits{c2.x, c2.y, c2.z, c2.w};
    }
}
template <bool H2>
__device__ __forceinline__ void mlp_split2(float x, float y, unsigned& p0, unsigned& p1, unsigned& p2) {
    if constexpr (H2) {
        union { f16x2_t h; unsigned u; } a, b;
        a.h = f16x2_t{(_Float16)x, (_Float16)y};
        b.h = f16x2_t{(_Float16)(x - (float)a.h[0]), (_Float16)(y - (float)a.h[1])};
        p0 = a.u; p1 = b.u; p2 = a.u;
    } else xp_split2(x, y, p0, p1, p2);
}

template <int C, int NW, int MODE, int NP, bool H2>
__global__ __launch_bounds__(NW * 64, H2 ? 2 : ((C <= 96 ? 8 : 4) / NW)) void mlp_fused_kernel(MlpParams p) {      // h2: two waves per SIMD at every C (two planes resident)
    static_assert(NP == 6 || NP == 3 || NP == 1, "partial products per multiply (gemm_x3_core.h)");
    static_assert(!H2 || NP == 3, "the split-fp16 engine always forms its three products");
    constexpr bool PRE = MODE == 1, PROJ_ONLY = MODE == 2;
    using T = MlpTile<C, H2>;
    constexpr int ROWB = T::ROWB;
    constexpr int KS = T::KS, NT = T::NT;
    // HALF2 (round 6, split-fp16 MLP instances wider than 96 channels): GELU + split of the SECOND half of a hidden chunk run between the fc2 matrix instructions of
    // its first half instead of ahead of fc2 — no extra registers, same instructions per row in the same order (bit-identical); C = 192: 250 -> 242 - 246 us
    // per launch, C = 96: no change (left off there).  profiles/r6_mlp_pipelined_agpr.txt
    constexpr bool HALF2 = H2 && MODE != 2 && C > 96;
    constexpr int NI = T::NI * 4 / NW;          // DMA instructions per wave and image
    static_assert(T::NI * 4 % NW == 0, "image pieces must divide among the waves");
    extern __shared__ __align__(16) unsigned char lds[];       // [3 image slots][b1 (H4 floats)][h2: 1 / row scale of fc1 (H4 floats)] — ONE array (LDS-DMA waits)
    unsigned char* const bias_lds = lds + 3 * T::IMGP;
    unsigned char* const inv1_lds = bias_lds + (size_t)p.H4 * 4;
    const int lane = threadIdx.x & 63, fr = lane & 31, g = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m0 = blockIdx.x * (NW * 32) + wave * 32;
    const int mrow = (m0 + fr < p.M) ? m0 + fr : p.M - 1;      // rows past M are computed on a copy of the last row, never stored
    const int NC = p.H4 / 32, NIMG = PROJ_ONLY ? p.Nout / 32 : 2 * NC + (PRE ? T::NT : 0);

    // image n of the packed stream -> ring slot; every wave moves NI KiB-sized pieces (source: scalar base + lane * 16)
    auto issue_image = [&](int n, int slot) {
        n = n < NIMG ? n : NIMG - 1;               // past the end: the last image again, into a slot nobody reads (uniform wait counts)
        const unsigned char* src = p.Wpack + (size_t)n * T::IMGP + wave * 1024 + lane * 16;
        unsigned char* dst = lds + slot * T::IMGP + wave * 1024;
#pragma unroll
        for (int i = 0; i < NI; ++i) __builtin_amdgcn_global_load_lds(src + i * (NW * 1024), (lds_ptr_t)(dst + i * (NW * 1024)), 16, 0, 0);
    };
    // one piece (i < NI) of image n -> ring slot: lets a phase spread its DMA issue between its MFMA groups (XP_MLP_SPREAD_DMA)
    auto issue_piece = [&](int n, int slot, int i) {
        n = n < NIMG ? n : NIMG - 1;
        const unsigned char* src = p.Wpack + (size_t)n * T::IMGP + wave * 1024 + lane * 16 + i * (NW * 1024);
        unsigned char* dst = lds + slot * T::IMGP + wave * 1024 + i * (NW * 1024);
        __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)dst, 16, 0, 0);
    };
    // all but the most recently issued image have landed
    auto wait_images = [&]() {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NI) : "memory");
        static_assert(NI <= 63, "wait_images");
    };
    auto barrier = [&]() { if (!(XP_MLP_DBG & 4)) __builtin_amdgcn_s_barrier(); };

    // ---- prologue: this lane's half (k = 16 s + 8 g .. + 7) of row mrow, LayerNorm, split into planes ----
    float4 xv[KS][2];
    {
        const float* xr = p.X + (int64_t)mrow * C + 8 * g;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            xv[s][0] = *reinterpret_cast<const float4*>(xr + 16 * s);
            xv[s][1] = *reinterpret_cast<const float4*>(xr + 16 * s + 4);
        }
    }
    if (!PROJ_ONLY)
        for (int i = threadIdx.x; i < p.H4 / 4; i += NW * 64) {
            float4 bv = reinterpret_cast<const float4*>(p.b1)[i];
            if constexpr (H2) {      // accumulators carry the row scale 2^k of fc1.weight: start them at b1 * 2^k, multiply by 2^-k before the GELU (both exact)
                const float4 iv = reinterpret_cast<const float4*>(p.s1)[i];
                reinterpret_cast<float4*>(inv1_lds)[i] = iv;
                bv = make_float4(bv.x / iv.x, bv.y / iv.y, bv.z / iv.z, bv.w / iv.w);
            }
            reinterpret_cast<float4*>(bias_lds)[i] = bv;
        }
    frag_bits xp[KS][3];       // planes of the B operand of the fc1-type MFMAs: LN(x) of this wave's rows (PRE: first the T1 rows)
    auto layer_norm_split = [&]() {
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s)
            sum += ((xv[s][0].x + xv[s][0].y) + (xv[s][0].z + xv[s][0].w)) + ((xv[s][1].x + xv[s][1].y) + (xv[s][1].z + xv[s][1].w));
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum / (float)C;
        float q2 = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float dx = xv[s][e].x - mean, dy = xv[s][e].y - mean, dz = xv[s][e].z - mean, dw = xv[s][e].w - mean;
                q2 = fmaf(dx, dx, q2); q2 = fmaf(dy, dy, q2); q2 = fmaf(dz, dz, q2); q2 = fmaf(dw, dw, q2);
            }
        q2 += __shfl_xor(q2, 32, 64);
        const float rstd = 1.f / sqrtf(q2 / (float)C + p.eps);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float4 w0 = *reinterpret_cast<const float4*>(p.ln_w + 16 * s + 8 * g), w1 = *reinterpret_cast<const float4*>(p.ln_w + 16 * s + 8 * g + 4);
            const float4 c0 = *reinterpret_cast<const float4*>(p.ln_b + 16 * s + 8 * g), c1 = *reinterpret_cast<const float4*>(p.ln_b + 16 * s + 8 * g + 4);
            float4 lo, hi;
            lo.x = (xv[s][0].x - mean) * rstd * w0.x + c0.x; lo.y = (xv[s][0].y - mean) * rstd * w0.y + c0.y;
            lo.z = (xv[s][0].z - mean) * rstd * w0.z + c0.z; lo.w = (xv[s][0].w - mean) * rstd * w0.w + c0.w;
            hi.x = (xv[s][1].x - mean) * rstd * w1.x + c1.x; hi.y = (xv[s][1].y - mean) * rstd * w1.y + c1.y;
            hi.z = (xv[s][1].z - mean) * rstd * w1.z + c1.z; hi.w = (xv[s][1].w - mean) * rstd * w1.w + c1.w;
            mlp_split8<H2>(lo, hi, xp[s]);
        }
    };
    if (PRE) {                 // rows of T1 in the same lane layout, split into planes
        const float* tr = p.T1 + (int64_t)mrow * C + 8 * g;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float4 lo = *reinterpret_cast<const float4*>(tr + 16 * s), hi = *reinterpret_cast<const float4*>(tr + 16 * s + 4);
            mlp_split8<H2>(lo, hi, xp[s]);
        }
    } else {
        layer_norm_split();
    }
    // every ordinary global load above has been consumed: from here to the epilogue the only VMEM traffic is LDS-DMA
    // (PRE: plus the stores of the updated rows)
    issue_image(0, 0);
    issue_image(1, 1);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    barrier();

    f32x16 oacc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};     // smallest partial products first (h2 = the last three: a1 b0, a0 b1, a0 b0)
    constexpr int NPLD = H2 ? 2 : 3;                                          // planes read from an image row
    const int frag = fr * ROWB + 16 * g;
    // h2: 1 / row scale of the hidden units held by this lane's accumulator registers (registers 8j .. 8j+7 = units 32c + 16j + 8g + 0..7)
    auto load_inv1 = [&](int c, float (&iv)[16]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float4 lo = *reinterpret_cast<const float4*>(inv1_lds + (32 * c + 16 * j + 8 * g) * 4);
            const float4 hi = *reinterpret_cast<const float4*>(inv1_lds + (32 * c + 16 * j + 8 * g + 4) * 4);
            iv[8 * j + 0] = lo.x; iv[8 * j + 1] = lo.y; iv[8 * j + 2] = lo.z; iv[8 * j + 3] = lo.w;
            iv[8 * j + 4] = hi.x; iv[8 * j + 5] = hi.y; iv[8 * j + 6] = hi.z; iv[8 * j + 7] = hi.w;
        }
    };
    // The same 16 floats by inline-asm LDS reads.  hipcc guards an ordinary ds_read of these arrays with s_waitcnt vmcnt(0) whenever an LDS-DMA is in
    // flight (it cannot tell the bias region from the image ring: one __shared__ array, and it must be one — cdna_hip_programming.md §5 trap (a)): the image
    // issued a phase ago would be waited for a phase early.  An asm load is invisible to that pass; its own completion is waited for right here.
    auto lds_read16_asm = [&](const unsigned char* base, int c, float (&v)[16]) {
        typedef __attribute__((address_space(3))) const unsigned char* lds_cptr;
        const unsigned addr = (unsigned)(size_t)(lds_cptr)(base + (32 * c + 8 * g) * 4);
        float4 q0, q1, q2, q3;
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:64\n\tds_read_b128 %3, %4 offset:80\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(addr) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w; v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
        v[8] = q2.x; v[9] = q2.y; v[10] = q2.z; v[11] = q2.w; v[12] = q3.x; v[13] = q3.y; v[14] = q3.z; v[15] = q3.w;
    };
    float inv_cur[16];                     // h2: scales of the chunk whose GELU is being evaluated
    // 8 consecutive inverse row scales of a projection's output columns col0 .. col0 + 7 (h2; the x3 planes are unscaled)
    auto load_inv8 = [&](const float* sc, int col0, float (&iv)[8]) {
        const float4 lo = *reinterpret_cast<const float4*>(sc + col0), hi = *reinterpret_cast<const float4*>(sc + col0 + 4);
        iv[0] = lo.x; iv[1] = lo.y; iv[2] = lo.z; iv[3] = lo.w; iv[4] = hi.x; iv[5] = hi.y; iv[6] = hi.z; iv[7] = hi.w;
    };

    // accumulator start = b1 of the chunk: registers 8j .. 8j+7 = hidden units 32c + 16j + 8g + 0..7
    auto load_bias = [&](int c, f32x16& h) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float4 lo = *reinterpret_cast<const float4*>(bias_lds + (32 * c + 16 * j + 8 * g) * 4);
            const float4 hi = *reinterpret_cast<const float4*>(bias_lds + (32 * c + 16 * j + 8 * g + 4) * 4);
            h[8 * j + 0] = lo.x; h[8 * j + 1] = lo.y; h[8 * j + 2] = lo.z; h[8 * j + 3] = lo.w;
            h[8 * j + 4] = hi.x; h[8 * j + 5] = hi.y; h[8 * j + 6] = hi.z; h[8 * j + 7] = hi.w;
        }
    };
    unsigned hp[2][3][4];                  // GELU(hidden chunk) as planes (slab j, plane, 4 x 2 bf16): the A operand of fc2
    auto hfrag = [&](int j, int pl) { return frag_bits{hp[j][pl][0], hp[j][pl][1], hp[j][pl][2], hp[j][pl][3]}; };
    int n = 0, slot = 0;                   // image used by the current phase and its ring slot; image n + 2 goes to slot - 1 (mod 3)
    // The DMA pieces of image n + 2 are issued BETWEEN the MFMA groups of phase n (one piece per k slab / output step), not in one
    // burst at its start: a few per cent faster (363 vs 375 us at C = 192) — an LDS-DMA issued among MFMAs costs less than one issued
    // next to other pieces and fragment reads.
    auto spread = [&](int step, int nsteps) {      // called after MFMA group `step` of `nsteps`: the pieces due by then
        if (XP_MLP_DBG & 2) return;
#pragma unroll
        for (int i = 0; i < NI; ++i) if (i * nsteps / NI == step) issue_piece(n + 2, slot == 0 ? 2 : slot - 1, i);
    };
    // VALU slices placed between MFMAs.  Slices 0..15: GELU of element r in place; 16..19 / 20..23: bf16 split of pair q of half 0 / 1.
    auto slice = [&](int k, f32x16& h) {
        // HALF2 (below): a slice's result is pinned where the slice stands — hipcc otherwise SINKS this register-only code past the sched_barriers to its
        // consumer and nothing runs between the matrix instructions (round 6: the ISA of the first interleaved build; profiles/r6_mlp_pipelined_agpr.txt)
        if (k < 16) {
            float v = mlp_gelu(H2 ? h[k] * inv_cur[k] : h[k]);
            if (HALF2) asm volatile("" : "+v"(v));
            h[k] = v;
            return;
        }
        const int j = (k - 16) >> 2, q = (k - 16) & 3;
        if (XP_MLP_DBG & 8) { hp[j][0][q] = __float_as_uint(h[8 * j + 2 * q]); hp[j][1][q] = __float_as_uint(h[8 * j + 2 * q + 1]); hp[j][2][q] = hp[j][0][q]; }
        else mlp_split2<H2>(h[8 * j + 2 * q], h[8 * j + 2 * q + 1], hp[j][0][q], hp[j][1][q], hp[j][2][q]);
        if (HALF2) asm volatile("" : "+v"(hp[j][0][q]), "+v"(hp[j][1][q]));
    };
    // fc1 of one chunk from the W1 image in `slot` into nxt (preloaded with the bias); between the MFMAs, slices [0, NSL) of the
    // previous chunk's accumulators `cur` (NSL = 0: none)
    auto fc1 = [&](int slot, f32x16& nxt, f32x16& cur, auto nsl_tag) {
        constexpr int NSL = decltype(nsl_tag)::value;
        const unsigned char* img = lds + slot * T::IMGP + frag;
        // fragment ring FD + 1 deep: the reads of slab s + FD are requested before the MFMAs of slab s (round 5: FD = 2 for the split-fp16 instances — with
        // one slab of look-ahead the ds_read latency exceeded the 96 cycles of a slab's three MFMAs and every slab stalled: stamps in profiles/r5_mlp_pingpong.txt; C = 96: 262.7 - 268.7 -> 258.8 us, same bits)
        constexpr int FD = (H2 && C <= 128 && XP_MLP_FRAG_DEPTH > 1 && KS > 2) ? 2 : 1;      // (C = 192 is at the register limit: 8 -> 18 spills, no gain)
        frag_bits a[FD + 1][3];
#pragma unroll
        for (int d = 0; d < FD; ++d)
#pragma unroll
            for (int pl = 0; pl < NPLD; ++pl) a[d][pl] = *reinterpret_cast<const frag_bits*>(img + d * 32 * ROWB + pl * 32);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s + FD < KS) {
#pragma unroll
                for (int pl = 0; pl < NPLD; ++pl) a[(s + FD) % (FD + 1)][pl] = *reinterpret_cast<const frag_bits*>(img + (s + FD) * 32 * ROWB + pl * 32);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pp = 6 - NP; pp < 6; ++pp) {
                const int m = s * NP + pp - (6 - NP);
                if (XP_MLP_DBG & 16) { if (pp == 5) nxt[0] += __uint_as_float(a[s % (FD + 1)][0][0]) * __uint_as_float(xp[s][0][0]); }
                else nxt = mlp_mfma<H2>(a[s % (FD + 1)][PA[pp]], xp[s][PB[pp]], nxt);
#pragma unroll
                for (int k = (m * NSL + NP * KS - 1) / (NP * KS); k < ((m + 1) * NSL + NP * KS - 1) / (NP * KS); ++k) slice(k, cur);
                __builtin_amdgcn_sched_barrier(0);
            }
            spread(s, KS);
        }
    };
    // fc2 of one chunk from the W2 image in `slot`: hidden slab 0 for every output tile first, with the split of half 1
    // (slices 20..23 of `cur`) between those MFMAs, then hidden slab 1
    auto fc2 = [&](int slot, f32x16& cur) {
        const unsigned char* img = lds + slot * T::IMGP + frag;
        constexpr int FD = (H2 && C <= 128 && XP_MLP_FRAG_DEPTH > 1 && 2 * NT > 2) ? 2 : 1;       // fragment look-ahead, as in fc1
        frag_bits b[FD + 1][3];
        auto b_addr = [&](int i) { return img + ((i / NT) * C + (i % NT) * 32) * ROWB; };
#pragma unroll
        for (int d = 0; d < FD; ++d)
#pragma unroll
            for (int pl = 0; pl < NPLD; ++pl) b[d][pl] = *reinterpret_cast<const frag_bits*>(b_addr(d) + pl * 32);
#pragma unroll
        for (int i = 0; i < 2 * NT; ++i) {          // step i = (hidden slab j = i / NT, output tile t = i % NT)
            const int j = i / NT, t = i % NT;
            if (i + FD < 2 * NT) {
#pragma unroll
                for (int pl = 0; pl < NPLD; ++pl) b[(i + FD) % (FD + 1)][pl] = *reinterpret_cast<const frag_bits*>(b_addr(i + FD) + pl * 32);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pp = 6 - NP; pp < 6; ++pp) {
                if (XP_MLP_DBG & 16) { if (pp == 5) oacc[t][0] += __uint_as_float(hp[j][0][0]) * __uint_as_float(b[i % (FD + 1)][0][0]); }
                else oacc[t] = mlp_mfma<H2>(hfrag(j, PA[pp]), b[i % (FD + 1)][PB[pp]], oacc[t]);
                if (HALF2) {
                    // hidden half 1 (GELU of elements 8..15, then their split) spread over the MFMAs of hidden half 0
                    constexpr int ORDER[12] = {8, 9, 20, 10, 11, 21, 12, 13, 22, 14, 15, 23};
                    const int m = i * NP + pp - (6 - NP);
                    if (i < NT) {
#pragma unroll
                        for (int k = (m * 12 + NT * NP - 1) / (NT * NP); k < ((m + 1) * 12 + NT * NP - 1) / (NT * NP); ++k) slice(ORDER[k], cur);
                    }
                } else
                if (i == 0 && pp - (6 - NP) < (NP >= 4 ? 4 : 1)) {      // split of half 1: one pair per MFMA (NP = 6), else all at once
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (NP >= 4 ? q == pp - (6 - NP) : true) slice(20 + q, cur);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            spread(i, 2 * NT);
        }
    };
    auto next_slot = [](int s) { return s == 2 ? 0 : s + 1; };

    f32x16 h0, h1;
    auto end_phase = [&]() { wait_images(); barrier(); ++n; slot = next_slot(slot); };
    if constexpr (PROJ_ONLY) {
        // one fc1-type phase per 32 output columns; registers 8jj .. 8jj+7 of lane-half g are columns 32t + 16jj + 8g + 0..7 of row
        // m0 + fr (the permuted weight rows again), stored as two float4 per jj
        float* ow = p.Out + (int64_t)(m0 + fr) * p.Nout + 8 * g;
        const bool rok = m0 + fr < p.M;
        for (int t = 0; t < NIMG; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) h0[r] = 0.f;
            fc1(slot, h0, h1, std::integral_constant<int, 0>{});
            end_phase();
            if (rok) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    if constexpr (H2) {      // undo the row scale of the projection weight (exact)
                        float iv[8];
                        load_inv8(p.s0, 32 * t + 16 * jj + 8 * g, iv);
#pragma unroll
                        for (int e = 0; e < 8; ++e) h0[8 * jj + e] *= iv[e];
                    }
                    *reinterpret_cast<float4*>(ow + 32 * t + 16 * jj) = make_float4(h0[8 * jj + 0], h0[8 * jj + 1], h0[8 * jj + 2], h0[8 * jj + 3]);
                    *reinterpret_cast<float4*>(ow + 32 * t + 16 * jj + 4) = make_float4(h0[8 * jj + 4], h0[8 * jj + 5], h0[8 * jj + 6], h0[8 * jj + 7]);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the look-ahead DMAs past the last image
        return;
    }
    if (PRE) {
        // x <- x + T1 W0^T (reference VMamba.py:663 out_proj, :1229 first residual): one fc1-type phase per 32 output channels.  With
        // the permuted weight rows the accumulator registers 8jj .. 8jj+7 of lane-half g are channels 16 (2t + jj) + 8g + 0..7 — the
        // lane layout of xv — so the sum never leaves the registers; the updated rows are stored for the epilogue's residual read.
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) h0[r] = 0.f;
            fc1(slot, h0, h1, std::integral_constant<int, 0>{});
            end_phase();
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                if constexpr (H2) {          // undo the row scale of out_proj.weight (exact): channels 16 (2t + jj) + 8g + 0..7
                    float iv[8];
                    load_inv8(p.s0, 16 * (2 * t + jj) + 8 * g, iv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) h0[8 * jj + e] *= iv[e];
                }
                float4& lo = xv[2 * t + jj][0]; float4& hi = xv[2 * t + jj][1];
                lo.x = lo.x + h0[8 * jj + 0]; lo.y = lo.y + h0[8 * jj + 1]; lo.z = lo.z + h0[8 * jj + 2]; lo.w = lo.w + h0[8 * jj + 3];
                hi.x = hi.x + h0[8 * jj + 4]; hi.y = hi.y + h0[8 * jj + 5]; hi.z = hi.z + h0[8 * jj + 6]; hi.w = hi.w + h0[8 * jj + 7];
            }
        }
        if (m0 + fr < p.M) {
            float* xw = p.X + (int64_t)(m0 + fr) * C + 8 * g;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                *reinterpret_cast<float4*>(xw + 16 * s) = xv[s][0];
                *reinterpret_cast<float4*>(xw + 16 * s + 4) = xv[s][1];
            }
        }
        layer_norm_split();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // stores and plain loads retired: the counted waits below see only LDS-DMA
    }
    // phase 0 of the MLP: fc1 of chunk 0
    // (the bias reads come BEFORE the phase's DMA issue: hipcc guards these ds_reads — not the fragment reads — with s_waitcnt vmcnt(0),
    // which after the issue would wait for the image that was just requested; before it, it waits for one requested a phase ago)
    load_bias(0, h0);
    __builtin_amdgcn_sched_barrier(0);
    fc1(slot, h0, h1, std::integral_constant<int, 0>{});
    end_phase();
    // chunk c: phase A = fc1(c+1) with GELU(c) (all 16 elements) and the split of half 0 between its MFMAs (image 1 + 2c);
    //          phase B = fc2(c) (image 2 + 2c)
    auto iter = [&](int c, f32x16& cur, f32x16& nxt) {
        if constexpr (H2) load_inv1(c, inv_cur);
        load_bias(c + 1, nxt);
        __builtin_amdgcn_sched_barrier(0);
        fc1(slot, nxt, cur, std::integral_constant<int, 20>{});
        end_phase();
        fc2(slot, cur);
        end_phase();
    };
    int c = 0;
    if constexpr (H2) {
        // split-fp16 instances: ONE hidden accumulator, chunk after chunk — fc1(c), GELU + split, fc2(c) — over a stream packed in that
        // order.  The x3 schedule below (fc1 of chunk c+1 with the GELU of chunk c between its MFMAs, two accumulator sets) spills 149
        // registers at C = 192 with two planes resident and eight waves (8 here); measured 314 -> 272 us at C = 192, 278 -> 266 us at C = 96.
        for (; c < NC; ++c) {
#ifndef XP_MLP_ASM_LDS
#define XP_MLP_ASM_LDS 1   /* 0: the bias / scale reads as ordinary ds_reads (hipcc then puts s_waitcnt vmcnt(0) in front of them: the image issued a phase ago is waited for a phase early) */
#endif
            if (c > 0) {
                if (XP_MLP_ASM_LDS) {
                    float bv[16];
                    lds_read16_asm(bias_lds, c, bv);
#pragma unroll
                    for (int r = 0; r < 16; ++r) h0[r] = bv[r];
                } else load_bias(c, h0);
                __builtin_amdgcn_sched_barrier(0);
                fc1(slot, h0, h0, std::integral_constant<int, 0>{});
                end_phase();
            }
            if constexpr (H2) { if (XP_MLP_ASM_LDS) lds_read16_asm(inv1_lds, c, inv_cur); else load_inv1(c, inv_cur); }
#pragma unroll
            for (int k = 0; k < 20; ++k) if (!HALF2 || k < 8 || k >= 16) slice(k, h0);
            fc2(slot, h0);
            if (c + 1 < NC) end_phase();
        }
        c = NC;
    }
    if constexpr (!H2)
    for (; c + 2 < NC; c += 2) { iter(c, h0, h1); iter(c + 1, h1, h0); }
    auto tail = [&](f32x16& cur) {         // last chunk: nothing left to overlap the GELU with; image 2 NC - 1
        if constexpr (H2) load_inv1(NC - 1, inv_cur);
#pragma unroll
        for (int k = 0; k < 20; ++k) slice(k, cur);
        fc2(slot, cur);
    };
    if (c == NC) {} else if (c + 1 < NC) { iter(c, h0, h1); tail(h1); } else tail(h0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the dummy DMAs of the last phases

    // ---- epilogue: x[m][n] = x[m][n] + (acc + b2[n]); lane = column n, registers = rows (r&3) + 8(r>>2) + 4g ----
    float* xb = p.X + (int64_t)m0 * C;
    auto epilogue = [&](auto interior_tag) {
        constexpr bool INTERIOR = decltype(interior_tag)::value;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int col = t * 32 + fr;
            const float bi = p.b2[col];
            const float ws = H2 ? p.s2[col] : 1.f;          // h2: 1 / row scale of fc2.weight (exact)
            float rv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (r & 3) + 8 * (r >> 2) + 4 * g;
                const float* rp = xb + ((INTERIOR || m0 + rl < p.M) ? rl : 0) * C + col;
                rv[r] = PRE ? __hip_atomic_load(rp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *rp;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (r & 3) + 8 * (r >> 2) + 4 * g;
                const float v = H2 ? oacc[t][r] * ws + bi : oacc[t][r] + bi;
                if (INTERIOR || m0 + rl < p.M) xb[rl * C + col] = rv[r] + v;
            }
        }
    };
    if (m0 + 32 <= p.M) epilogue(std::true_type{}); else epilogue(std::false_type{});
}

template <int C, int NW, int MODE, int NP, bool H2 = false>
int launch_mlp_np(const MlpParams& p, hipStream_t s) {
    constexpr bool PRE = MODE == 1;
    using T = MlpTile<C, H2>;
    const size_t lds_bytes = 3 * (size_t)T::IMGP + (size_t)p.H4 * 4 * (H2 ? 2 : 1);
    static XpPerDeviceOnce attr_once;
    if (attr_once.need()) {
        XP_HIP_WARN(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fused_kernel<C, NW, MODE, NP, H2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  3 * T::IMGP + 4096 * 4 * (H2 ? 2 : 1)));
    }
    static const bool by_shape = getenv("XP_PROF_SHAPES") != nullptr;
    const char* eng = H2 ? "_h2_c" : "_x3_c";
    std::string tag = std::string(MODE == 2 ? "ln_proj" : PRE ? "proj_mlp_fused" : "mlp_fused") + eng + std::to_string(C);      // one tag per kernel instance
    if (NP != 6 && !H2) tag += "_np" + std::to_string(NP);
    if (by_shape) tag += "_M" + std::to_string(p.M);
    // flops = algorithmic 2*M*C*H4 per GEMM (f32-equivalent); bytes: x read twice (LN input, residual) and written once
    XpProfScope prof(tag.c_str(), s, MODE == 2 ? 2.0 * p.M * C * (double)p.Nout : 4.0 * p.M * C * (double)p.H4 + (PRE ? 2.0 * p.M * C * (double)C : 0.0),
                     MODE == 2 ? 4.0 * p.M * (C + (double)p.Nout) + 6.0 * C * (double)p.Nout : (PRE ? 20.0 : 12.0) * p.M * C + 12.0 * C * (double)p.H4);
    hipLaunchKernelGGL((mlp_fused_kernel<C, NW, MODE, NP, H2>), dim3(xp_cdiv(p.M, NW * 32)), dim3(NW * 64), lds_bytes, s, p);
    XP_LAUNCH_CHECK();
    return XP_OK;
}

template <int C, int NW, int MODE>
int launch_mlp_pre(const MlpParams& p, hipStream_t s) {
    // split-fp16 instances (scales present).  C = 192: one 8-wave workgroup per CU (its three 32 KB image slots leave no room for a
    // second one) puts two waves on every SIMD; the narrower ones run two 4-wave workgroups per CU.
    static const bool nw4 = getenv("XP_MLP_H2_NW4") != nullptr && atoi(getenv("XP_MLP_H2_NW4")) != 0;      // A/B: 0.716 (8 waves) vs 0.747 ms (4 waves) per two launches
    if (p.s2 || (MODE == 2 && p.s0)) {
        if (C == 192 && !nw4) {
            // One 256-row workgroup per CU: M = 76 800 rows (16 images of 480 x 640 at stage 1) is 300 workgroups = one full round of the chip + 44 workgroups
            // that take as long again.  When the last round would be less than half full, its rows run as 128-row (4-wave) workgroups in a second launch —
            // twice as many CUs for them, 0.69 of the time per round (XP_MLP_H2_NW4 A/B).  Every wave owns its 32 rows end to end in both instances, so the
            // split never changes a result bit (and the batch-invariance tests compare exactly such calls); XP_MLP_TAIL=0 turns it off.
            static const bool tail_split = !(getenv("XP_MLP_TAIL") && atoi(getenv("XP_MLP_TAIL")) == 0);
            static int n_cu = 0;
            if (n_cu == 0) {
                int dev = 0, v = 0;
                if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
                n_cu = v;
            }
            auto main_launch = [&](const MlpParams& q) {
                return launch_mlp_np<C, (C == 192 ? 8 : 4), MODE, 3, true>(q, s);
            };
            const int round_rows = n_cu * 256;
            // Small calls (round 6: the 256 x 256 crops of the C5 head, single pairs): when 256-row workgroups would occupy less than half of the CUs, 128-row
            // (4-wave) workgroups fill twice as many and a workgroup's latency — the kernel's time at this size — is no longer (same bits: see above)
            if (p.M <= round_rows / 2) return launch_mlp_np<C, 4, MODE, 3, true>(p, s);
            const int rem = p.M % round_rows;
            if (tail_split && p.M > round_rows && rem > 0 && rem <= round_rows / 2) {
                MlpParams a = p, b = p;
                a.M = p.M - rem;
                b.M = rem;
                b.X = p.X + (int64_t)a.M * C;
                if (p.T1) b.T1 = p.T1 + (int64_t)a.M * C;
                if (p.Out) b.Out = p.Out + (int64_t)a.M * p.Nout;
                const int rc = main_launch(a);
                return rc != XP_OK ? rc : launch_mlp_np<C, 4, MODE, 3, true>(b, s);
            }
            return main_launch(p);
        }
        return launch_mlp_np<C, 4, MODE, 3, true>(p, s);
    }
    switch (xp_dense_products_value()) {      // precision class of the dense kernels (xp_set_dense_products)
        case 1: return launch_mlp_np<C, 4, MODE, 1>(p, s);
        case 3: return launch_mlp_np<C, 4, MODE, 3>(p, s);
        default: return launch_mlp_np<C, NW, MODE, 6>(p, s);
    }
}

template <int C, int NW>
int launch_mlp(const MlpParams& p, hipStream_t s) {
    if (p.Out) return launch_mlp_pre<C, 4, 2>(p, s);
    return p.T1 ? launch_mlp_pre<C, NW, 1>(p, s) : launch_mlp_pre<C, NW, 0>(p, s);
}

template <int C, bool H2 = false>
size_t pack_bytes(int H4, int with_proj) { return (size_t)(2 * (H4 / 32) + (with_proj ? MlpTile<C, H2>::NT : 0)) * MlpTile<C, H2>::IMGP; }

template <bool H2>
int pack_launch(const void* W1, const void* W2, const void* W0, void* out, int C, int H4, int n0, size_t bytes, hipStream_t s) {
    const int64_t units = (int64_t)(bytes / 16);
    const dim3 grid((unsigned)xp_cdiv(units, (int64_t)256)), block(256);
    const uint4 *w1 = (const uint4*)W1, *w2 = (const uint4*)W2, *w0 = (const uint4*)W0;
    if (C == 32) hipLaunchKernelGGL((mlp_pack_kernel<32, H2>), grid, block, 0, s, w1, w2, w0, (uint4*)out, H4, n0);
    else if (C == 64) hipLaunchKernelGGL((mlp_pack_kernel<64, H2>), grid, block, 0, s, w1, w2, w0, (uint4*)out, H4, n0);
    else if (C == 96) hipLaunchKernelGGL((mlp_pack_kernel<96, H2>), grid, block, 0, s, w1, w2, w0, (uint4*)out, H4, n0);
    else if (C == 128) hipLaunchKernelGGL((mlp_pack_kernel<128, H2>), grid, block, 0, s, w1, w2, w0, (uint4*)out, H4, n0);
    else hipLaunchKernelGGL((mlp_pack_kernel<192, H2>), grid, block, 0, s, w1, w2, w0, (uint4*)out, H4, n0);
    XP_LAUNCH_CHECK();
    return XP_OK;
}

// the N inverse row scales behind the planes of an xp_split_weights_h2 buffer (gemm_h2.hip)
const float* h2_scales_of(const void* Wh2, int N, int K) {
    return reinterpret_cast<const float*>(reinterpret_cast<const char*>(Wh2) + (size_t)N * ((K + H2_BK - 1) / H2_BK) * H2_SLAB_UNITS * 16);
}

int run_mlp(const MlpParams& p, int C, hipStream_t s) {
    static const bool nw8 = getenv("XP_MLP_NW8") != nullptr && atoi(getenv("XP_MLP_NW8")) != 0;     // tuning experiment
    switch (C) {
        case 32: return nw8 ? launch_mlp<32, 8>(p, s) : launch_mlp<32, 4>(p, s);
        case 64: return nw8 ? launch_mlp<64, 8>(p, s) : launch_mlp<64, 4>(p, s);
        case 96: return nw8 ? launch_mlp<96, 8>(p, s) : launch_mlp<96, 4>(p, s);
        case 128: return launch_mlp<128, 4>(p, s);
        default: return launch_mlp<192, 4>(p, s);
    }
}

}  // namespace

extern "C" int xp_mlp_fused_x3_supported(int C, int H4) {
    return (C == 32 || C == 64 || C == 96 || C == 128 || C == 192) && H4 >= 64 && H4 % 32 == 0 && H4 <= 4096;
}

extern "C" size_t xp_mlp_fused_x3_pack_bytes(int C, int H4, int with_proj) {
    if (!xp_mlp_fused_x3_supported(C, H4)) return 0;
    const int w = with_proj;
    return C == 32 ? pack_bytes<32>(H4, w) : C == 64 ? pack_bytes<64>(H4, w) : C == 96 ? pack_bytes<96>(H4, w) : C == 128 ? pack_bytes<128>(H4, w) : pack_bytes<192>(H4, w);
}

extern "C" int xp_mlp_fused_x3_pack(const void* W1x3, const void* W2x3, const void* W0x3, void* out, int C, int H4, void* stream) {
    XP_CHECK_ARG(W1x3 && W2x3 && out, "xp_mlp_fused_x3_pack: null pointer");
    XP_CHECK_ARG(xp_mlp_fused_x3_supported(C, H4), "xp_mlp_fused_x3_pack: unsupported shape C = %d, hidden = %d", C, H4);
    XP_CHECK_ARG((((uintptr_t)W1x3 | (uintptr_t)W2x3 | (uintptr_t)W0x3 | (uintptr_t)out) & 15) == 0, "xp_mlp_fused_x3_pack: pointers must be 16-byte aligned");
    return pack_launch<false>(W1x3, W2x3, W0x3, out, C, H4, C, xp_mlp_fused_x3_pack_bytes(C, H4, W0x3 != nullptr), (hipStream_t)stream);
}

extern "C" int xp_mlp_fused_x3(float* X, const float* T1, const float* ln_w, const float* ln_b, const void* Wpack, const float* b1,
                               const float* b2, int M, int C, int H4, float eps, void* stream) {
    XP_CHECK_ARG(X && ln_w && ln_b && Wpack && b1 && b2, "xp_mlp_fused_x3: null pointer");
    XP_CHECK_ARG(M > 0, "xp_mlp_fused_x3: bad M %d", M);
    XP_CHECK_ARG(xp_mlp_fused_x3_supported(C, H4), "xp_mlp_fused_x3: unsupported shape C = %d, hidden = %d (C in {32, 64, 96, 128, 192}, hidden %% 32 == 0, 64 <= hidden <= 4096)", C, H4);
    XP_CHECK_ARG((((uintptr_t)X | (uintptr_t)T1 | (uintptr_t)Wpack | (uintptr_t)b1 | (uintptr_t)ln_w | (uintptr_t)ln_b) & 15) == 0,
                 "xp_mlp_fused_x3: pointers must be 16-byte aligned");
    XP_CHECK_ARG(T1 != X, "xp_mlp_fused_x3: T1 must not alias X");
    MlpParams p{X, T1, nullptr, 0, ln_w, ln_b, (const unsigned char*)Wpack, b1, b2, M, H4, eps, nullptr, nullptr, nullptr};
    return run_mlp(p, C, (hipStream_t)stream);
}

extern "C" size_t xp_ln_proj_x3_pack_bytes(int C, int N) {
    if (!xp_mlp_fused_x3_supported(C, 64) || N <= 0 || N % 32) return 0;
    return xp_mlp_fused_x3_pack_bytes(C, 64, 0) / 4 * (size_t)(N / 32);      // one image per 32 output columns (the 64-wide MLP stream has 4)
}

extern "C" int xp_ln_proj_x3_pack(const void* W0x3, void* out, int C, int N, void* stream) {
    XP_CHECK_ARG(W0x3 && out, "xp_ln_proj_x3_pack: null pointer");
    XP_CHECK_ARG(xp_ln_proj_x3_pack_bytes(C, N) != 0, "xp_ln_proj_x3_pack: unsupported shape C = %d, N = %d (C in {32, 64, 96, 128, 192}, N %% 32 == 0)", C, N);
    XP_CHECK_ARG((((uintptr_t)W0x3 | (uintptr_t)out) & 15) == 0, "xp_ln_proj_x3_pack: pointers must be 16-byte aligned");
    return pack_launch<false>(W0x3, W0x3, W0x3, out, C, 0, N, xp_ln_proj_x3_pack_bytes(C, N), (hipStream_t)stream);
}

extern "C" int xp_ln_proj_x3(const float* X, const float* ln_w, const float* ln_b, const void* Wpack, float* Out, int M, int C, int N,
                             float eps, void* stream) {
    XP_CHECK_ARG(X && ln_w && ln_b && Wpack && Out, "xp_ln_proj_x3: null pointer");
    XP_CHECK_ARG(M > 0, "xp_ln_proj_x3: bad M %d", M);
    XP_CHECK_ARG(xp_ln_proj_x3_pack_bytes(C, N) != 0, "xp_ln_proj_x3: unsupported shape C = %d, N = %d (C in {32, 64, 96, 128, 192}, N %% 32 == 0)", C, N);
    XP_CHECK_ARG((((uintptr_t)X | (uintptr_t)Out | (uintptr_t)Wpack | (uintptr_t)ln_w | (uintptr_t)ln_b) & 15) == 0, "xp_ln_proj_x3: pointers must be 16-byte aligned");
    XP_CHECK_ARG((const float*)Out != X, "xp_ln_proj_x3: Out must not alias X");
    MlpParams p{const_cast<float*>(X), nullptr, Out, N, ln_w, ln_b, (const unsigned char*)Wpack, nullptr, nullptr, M, 64, eps, nullptr, nullptr, nullptr};
    return run_mlp(p, C, (hipStream_t)stream);
}

// ---- split-fp16 ("h2") instances: the same kernels on two fp16 planes and three products (gemm_h2_core.h); the weight streams are
// packed from xp_split_weights_h2 buffers, whose row scales the kernels undo (fc1: before the GELU; fc2 / projections: at the end)
extern "C" size_t xp_mlp_fused_h2_pack_bytes(int C, int H4, int with_proj) {
    if (!xp_mlp_fused_x3_supported(C, H4)) return 0;
    const int w = with_proj;
    return C == 32 ? pack_bytes<32, true>(H4, w) : C == 64 ? pack_bytes<64, true>(H4, w) : C == 96 ? pack_bytes<96, true>(H4, w)
         : C == 128 ? pack_bytes<128, true>(H4, w) : pack_bytes<192, true>(H4, w);
}

extern "C" int xp_mlp_fused_h2_pack(const void* W1h2, const void* W2h2, const void* W0h2, void* out, int C, int H4, void* stream) {
    XP_CHECK_ARG(W1h2 && W2h2 && out, "xp_mlp_fused_h2_pack: null pointer");
    XP_CHECK_ARG(xp_mlp_fused_x3_supported(C, H4), "xp_mlp_fused_h2_pack: unsupported shape C = %d, hidden = %d", C, H4);
    XP_CHECK_ARG((((uintptr_t)W1h2 | (uintptr_t)W2h2 | (uintptr_t)W0h2 | (uintptr_t)out) & 15) == 0, "xp_mlp_fused_h2_pack: pointers must be 16-byte aligned");
    return pack_launch<true>(W1h2, W2h2, W0h2, out, C, H4, C, xp_mlp_fused_h2_pack_bytes(C, H4, W0h2 != nullptr), (hipStream_t)stream);
}

// W1h2 (hidden, C), W2h2 (C, hidden), W0h2 (C, C) or NULL: the xp_split_weights_h2 buffers the stream was packed from (their row scales
// are read here); T1 != NULL iff the stream was packed with W0h2.
extern "C" int xp_mlp_fused_h2(float* X, const float* T1, const float* ln_w, const float* ln_b, const void* Wpack, const void* W1h2,
                               const void* W2h2, const void* W0h2, const float* b1, const float* b2, int M, int C, int H4, float eps, void* stream) {
    XP_CHECK_ARG(X && ln_w && ln_b && Wpack && W1h2 && W2h2 && b1 && b2, "xp_mlp_fused_h2: null pointer");
    XP_CHECK_ARG((T1 != nullptr) == (W0h2 != nullptr), "xp_mlp_fused_h2: T1 and W0h2 go together");
    XP_CHECK_ARG(M > 0, "xp_mlp_fused_h2: bad M %d", M);
    XP_CHECK_ARG(xp_mlp_fused_x3_supported(C, H4), "xp_mlp_fused_h2: unsupported shape C = %d, hidden = %d", C, H4);
    XP_CHECK_ARG((((uintptr_t)X | (uintptr_t)T1 | (uintptr_t)Wpack | (uintptr_t)b1 | (uintptr_t)ln_w | (uintptr_t)ln_b) & 15) == 0,
                 "xp_mlp_fused_h2: pointers must be 16-byte aligned");
    XP_CHECK_ARG(T1 != X, "xp_mlp_fused_h2: T1 must not alias X");
    MlpParams p{X, T1, nullptr, 0, ln_w, ln_b, (const unsigned char*)Wpack, b1, b2, M, H4, eps,
                W0h2 ? h2_scales_of(W0h2, C, C) : nullptr, h2_scales_of(W1h2, H4, C), h2_scales_of(W2h2, C, H4)};
    return run_mlp(p, C, (hipStream_t)stream);
}

extern "C" size_t xp_ln_proj_h2_pack_bytes(int C, int N) {
    if (!xp_mlp_fused_x3_supported(C, 64) || N <= 0 || N % 32) return 0;
    return xp_mlp_fused_h2_pack_bytes(C, 64, 0) / 4 * (size_t)(N / 32);
}

extern "C" int xp_ln_proj_h2_pack(const void* W0h2, void* out, int C, int N, void* stream) {
    XP_CHECK_ARG(W0h2 && out, "xp_ln_proj_h2_pack: null pointer");
    XP_CHECK_ARG(xp_ln_proj_h2_pack_bytes(C, N) != 0, "xp_ln_proj_h2_pack: unsupported shape C = %d, N = %d", C, N);
    XP_CHECK_ARG((((uintptr_t)W0h2 | (uintptr_t)out) & 15) == 0, "xp_ln_proj_h2_pack: pointers must be 16-byte aligned");
    return pack_launch<true>(W0h2, W0h2, W0h2, out, C, 0, N, xp_ln_proj_h2_pack_bytes(C, N), (hipStream_t)stream);
}

extern "C" int xp_ln_proj_h2(const float* X, const float* ln_w, const float* ln_b, const void* Wpack, const void* W0h2, float* Out, int M, int C,
                             int N, float eps, void* stream) {
    XP_CHECK_ARG(X && ln_w && ln_b && Wpack && W0h2 && Out, "xp_ln_proj_h2: null pointer");
    XP_CHECK_ARG(M > 0, "xp_ln_proj_h2: bad M %d", M);
    XP_CHECK_ARG(xp_ln_proj_h2_pack_bytes(C, N) != 0, "xp_ln_proj_h2: unsupported shape C = %d, N = %d", C, N);
    XP_CHECK_ARG((((uintptr_t)X | (uintptr_t)Out | (uintptr_t)Wpack | (uintptr_t)ln_w | (uintptr_t)ln_b) & 15) == 0, "xp_ln_proj_h2: pointers must be 16-byte aligned");
    XP_CHECK_ARG((const float*)Out != X, "xp_ln_proj_h2: Out must not alias X");
    MlpParams p{const_cast<float*>(X), nullptr, Out, N, ln_w, ln_b, (const unsigned char*)Wpack, nullptr, nullptr, M, 64, eps,
                h2_scales_of(W0h2, N, C), nullptr, nullptr};
    return run_mlp(p, C, (hipStream_t)stream);
}
