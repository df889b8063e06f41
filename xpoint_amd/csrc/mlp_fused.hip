// Fused VSS-block MLP for the wide stages:   x <- x + fc2(GELU(fc1(LN(x)) + b1)) + b2     (reference VMamba.py:1230-1234 VSSBlock
// second residual branch, :110-128 Mlp, nn.LayerNorm eps 1e-5, exact-erf GELU), in place, one launch instead of
// layernorm + fc1 GEMM + fc2 GEMM.  The (M, 4C) hidden tensor never exists: not in HBM (472 MB written and read back per
// block at stage 0 of a 16-image 480x640 batch), not in LDS either.
//
// Arithmetic = the split-bf16 engine of gemm_x3_core.h (every f32 operand is the exact sum of three bf16 values, six
// bf16 MFMA partial products per multiply, f32 accumulate), so results agree with the unfused path to f32 rounding.
//
// A wave owns 32 rows of x.  Per 32-wide chunk of hidden units:
//   fc1   hT[h][m] = sum_k W1[h][k] LN(x)[m][k]     A = W1 fragments from LDS, B = the wave's LN(x) rows, split into planes
//                                                    ONCE and held in registers for the whole kernel (C/16 slabs x 3 planes)
//   GELU  on the accumulator registers (+ b1), then the 16 values of a lane are split into bf16 planes in place
//   fc2   out[m][n] += sum_h hid[m][h] W2[n][h]     A = those registers, B = W2 fragments from LDS
// The accumulator of a 32x32x16 MFMA holds a column (here: the row m of x) on the lane and 16 rows in the registers, which
// is exactly an A operand over k = hidden unit: lane-half g, register r <-> row (r&3) + 8(r>>2) + 4g.  Storing the W1 rows
// of a chunk in LDS with bits 2 and 3 of the row index swapped makes that "row" the hidden unit 8g + (r&7) + 16(r>>3), i.e.
// registers 0..7 / 8..15 are two natural 16-wide k slabs and W2 needs no permutation.
//
// Only the weights go through LDS, shared by the 4 waves of a workgroup (128 rows): chunk images of W1 (C/16 slabs x 32
// rows) and W2 (2 slabs x C rows), both 2C rows x 112 B in the padded row format of gemm_x3_core.h (conflict-free
// ds_read_b128 fragments), filled by LDS-DMA (global_load_lds_dwordx4: no staging registers; the pad unit of a row is a
// duplicate load) one phase ahead into a 2-slot ring, one counted wait + barrier per phase.
#include <stdlib.h>

#include <string>

#include "gemm_x3_core.h"

namespace {

struct MlpParams {
    float* X;                 // (M, C) in / out
    const float* ln_w; const float* ln_b;
    const uint4* W1;          // xp_split_weights_x3 layout of fc1.weight (H4, C)
    const float* b1;
    const uint4* W2;          // xp_split_weights_x3 layout of fc2.weight (C, H4)
    const float* b2;
    int M, H4;
    float eps;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int C>
struct MlpTile {
    static constexpr int KS = C / 16;            // k slabs of fc1
    static constexpr int NT = C / 32;            // 32-wide output tiles of fc2
    static constexpr int ROWS = 2 * C;           // rows of a chunk image (W1: KS x 32, W2: 2 x C)
    static constexpr int UNITS = ROWS * 7;       // 16-byte units per image incl. the pad unit of every row
    static constexpr int IMG = UNITS * 16;       // bytes
    static constexpr int NWI = UNITS / 64;       // wave-level DMA instructions per image (1 KiB each)
    static constexpr int NI = (NWI + 3) / 4;     // per wave
    static_assert(C % 32 == 0 && UNITS % 64 == 0, "C must be a multiple of 32");
};

template <int C>
__global__ __launch_bounds__(256, 2) void mlp_fused_kernel(MlpParams p) {
    using T = MlpTile<C>;
    constexpr int KS = T::KS, NT = T::NT;
    extern __shared__ __align__(16) unsigned char lds[];       // [2 image slots][b1 (H4 floats)] — ONE array (LDS-DMA waits)
    unsigned char* const bias_lds = lds + 2 * T::IMG;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 31, g = lane >> 5;
    const int m0 = blockIdx.x * 128 + wave * 32;
    const int mrow = (m0 + fr < p.M) ? m0 + fr : p.M - 1;      // rows past M are computed on a copy of the last row, never stored
    const int NC = p.H4 / 32;

    // chunk image `ph` (even: W1 of chunk ph/2, odd: W2 of chunk ph/2) -> ring slot ph & 1
    auto issue_image = [&](int ph) {
        const int c = ph >> 1;
        unsigned char* slot = lds + (ph & 1) * T::IMG;
#pragma unroll
        for (int i = 0; i < T::NI; ++i) {
            int q = wave + 4 * i;                               // wave-level instruction index inside the image
            q = q < T::NWI ? q : T::NWI - 1;                    // surplus instructions repeat the last one (same bytes, same place)
            const int u = q * 64 + lane;
            const int row = u / 7, k = u - row * 7;
            const int unit = k < 6 ? k : 5;                     // pad unit: any valid address
            int64_t src;
            if (ph & 1) {                                       // W2: rows = (slab j, n)
                const int j = row / C, n = row - j * C;
                src = ((int64_t)(2 * c + j) * C + n) * X3_SLAB_UNITS + unit;
            } else {                                            // W1: rows = (slab s, permuted hidden row)
                const int s = row >> 5, hp = row & 31;
                const int h = 32 * c + ((hp & 0x13) | ((hp & 4) << 1) | ((hp & 8) >> 1));
                src = ((int64_t)s * p.H4 + h) * X3_SLAB_UNITS + unit;
            }
            const uint4* gp = ((ph & 1) ? p.W2 : p.W1) + src;
            __builtin_amdgcn_global_load_lds(gp, (lds_ptr_t)(slot + q * 1024), 16, 0, 0);
        }
    };

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
    for (int i = threadIdx.x; i < p.H4 / 4; i += 256)
        reinterpret_cast<float4*>(bias_lds)[i] = reinterpret_cast<const float4*>(p.b1)[i];
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
    bf16x8 xp[KS][3];                                           // LN(x) planes: the B operand of every fc1 MFMA of this wave
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const float4 w0 = *reinterpret_cast<const float4*>(p.ln_w + 16 * s + 8 * g), w1 = *reinterpret_cast<const float4*>(p.ln_w + 16 * s + 8 * g + 4);
        const float4 c0 = *reinterpret_cast<const float4*>(p.ln_b + 16 * s + 8 * g), c1 = *reinterpret_cast<const float4*>(p.ln_b + 16 * s + 8 * g + 4);
        float4 lo, hi;
        lo.x = (xv[s][0].x - mean) * rstd * w0.x + c0.x; lo.y = (xv[s][0].y - mean) * rstd * w0.y + c0.y;
        lo.z = (xv[s][0].z - mean) * rstd * w0.z + c0.z; lo.w = (xv[s][0].w - mean) * rstd * w0.w + c0.w;
        hi.x = (xv[s][1].x - mean) * rstd * w1.x + c1.x; hi.y = (xv[s][1].y - mean) * rstd * w1.y + c1.y;
        hi.z = (xv[s][1].z - mean) * rstd * w1.z + c1.z; hi.w = (xv[s][1].w - mean) * rstd * w1.w + c1.w;
        union { uint4 u; bf16x8 v; } c[3];
        xp_split8(lo, hi, c[0].u, c[1].u, c[2].u);
        xp[s][0] = c[0].v; xp[s][1] = c[1].v; xp[s][2] = c[2].v;
    }
    // every ordinary global load above has been consumed: from here to the epilogue the only VMEM traffic is LDS-DMA
    issue_image(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    f32x16 oacc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};     // smallest partial products first
    const int frag = fr * X3_ROWB + 16 * g;

    for (int c = 0; c < NC; ++c) {
        // ---- phase A: fc1 of chunk c (image 2c in slot 0), while W2 of chunk c lands in slot 1 ----
        issue_image(2 * c + 1);
        f32x16 hacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) hacc[r] = 0.f;
        {
            const unsigned char* img = lds + frag;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                bf16x8 a[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[pl] = *reinterpret_cast<const bf16x8*>(img + s * 32 * X3_ROWB + pl * 32);
#pragma unroll
                for (int pp = 0; pp < 6; ++pp) hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[pp]], xp[s][PB[pp]], hacc, 0, 0, 0);
            }
        }
        // bias + GELU + split: registers 8j .. 8j+7 = hidden units 32c + 16j + 8g + 0..7
        bf16x8 hp[2][3];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float4 b0 = *reinterpret_cast<const float4*>(bias_lds + (32 * c + 16 * j + 8 * g) * 4);
            const float4 b1v = *reinterpret_cast<const float4*>(bias_lds + (32 * c + 16 * j + 8 * g + 4) * 4);
            float4 lo, hi;
            lo.x = xp_gelu_fast(hacc[8 * j + 0] + b0.x); lo.y = xp_gelu_fast(hacc[8 * j + 1] + b0.y);
            lo.z = xp_gelu_fast(hacc[8 * j + 2] + b0.z); lo.w = xp_gelu_fast(hacc[8 * j + 3] + b0.w);
            hi.x = xp_gelu_fast(hacc[8 * j + 4] + b1v.x); hi.y = xp_gelu_fast(hacc[8 * j + 5] + b1v.y);
            hi.z = xp_gelu_fast(hacc[8 * j + 6] + b1v.z); hi.w = xp_gelu_fast(hacc[8 * j + 7] + b1v.w);
            union { uint4 u; bf16x8 v; } cc[3];
            xp_split8(lo, hi, cc[0].u, cc[1].u, cc[2].u);
            hp[j][0] = cc[0].v; hp[j][1] = cc[1].v; hp[j][2] = cc[2].v;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ---- phase B: fc2 of chunk c (image 2c+1 in slot 1), while W1 of chunk c+1 lands in slot 0 ----
        if (c + 1 < NC) issue_image(2 * c + 2);
        {
            const unsigned char* img = lds + T::IMG + frag;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    bf16x8 b[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) b[pl] = *reinterpret_cast<const bf16x8*>(img + (j * C + t * 32) * X3_ROWB + pl * 32);
#pragma unroll
                    for (int pp = 0; pp < 6; ++pp) oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hp[j][PA[pp]], b[PB[pp]], oacc[t], 0, 0, 0);
                }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- epilogue: x[m][n] = x[m][n] + (acc + b2[n]); lane = column n, registers = rows (r&3) + 8(r>>2) + 4g ----
    float* xb = p.X + (int64_t)m0 * C;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = t * 32 + fr;
        const float bi = p.b2[col];
        float rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = (r & 3) + 8 * (r >> 2) + 4 * g;
            rv[r] = xb[((m0 + rl < p.M) ? rl : 0) * C + col];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = (r & 3) + 8 * (r >> 2) + 4 * g;
            const float v = oacc[t][r] + bi;
            if (m0 + rl < p.M) xb[rl * C + col] = rv[r] + v;
        }
    }
}

template <int C>
int launch_mlp(const MlpParams& p, hipStream_t s) {
    using T = MlpTile<C>;
    const size_t lds_bytes = 2 * (size_t)T::IMG + (size_t)p.H4 * 4;
    static bool attr_set = false;
    if (!attr_set && lds_bytes > 48 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fused_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        attr_set = true;
    }
    static const bool by_shape = getenv("XP_PROF_SHAPES") != nullptr;
    std::string tag = "mlp_fused_x3";
    if (by_shape) tag += "_M" + std::to_string(p.M) + "_C" + std::to_string(C);
    // flops = algorithmic 2*M*C*H4 per GEMM (f32-equivalent); bytes: x read twice (LN input, residual) and written once
    XpProfScope prof(tag.c_str(), s, 4.0 * p.M * C * (double)p.H4, 12.0 * p.M * C + 12.0 * C * (double)p.H4);
    hipLaunchKernelGGL((mlp_fused_kernel<C>), dim3(xp_cdiv(p.M, 128)), dim3(256), lds_bytes, s, p);
    XP_LAUNCH_CHECK();
    return XP_OK;
}

}  // namespace

extern "C" int xp_mlp_fused_x3_supported(int C, int H4) {
    return (C == 32 || C == 64 || C == 96) && H4 > 0 && H4 % 32 == 0 && H4 <= 4096;
}

extern "C" int xp_mlp_fused_x3(float* X, const float* ln_w, const float* ln_b, const void* W1x3, const float* b1,
                               const void* W2x3, const float* b2, int M, int C, int H4, float eps, void* stream) {
    XP_CHECK_ARG(X && ln_w && ln_b && W1x3 && b1 && W2x3 && b2, "xp_mlp_fused_x3: null pointer");
    XP_CHECK_ARG(M > 0, "xp_mlp_fused_x3: bad M %d", M);
    XP_CHECK_ARG(xp_mlp_fused_x3_supported(C, H4), "xp_mlp_fused_x3: unsupported shape C = %d, hidden = %d (C in {32, 64, 96}, hidden %% 32 == 0)", C, H4);
    XP_CHECK_ARG((((uintptr_t)X | (uintptr_t)W1x3 | (uintptr_t)W2x3 | (uintptr_t)b1 | (uintptr_t)ln_w | (uintptr_t)ln_b) & 15) == 0,
                 "xp_mlp_fused_x3: pointers must be 16-byte aligned");
    MlpParams p{X, ln_w, ln_b, (const uint4*)W1x3, b1, (const uint4*)W2x3, b2, M, H4, eps};
    hipStream_t s = (hipStream_t)stream;
    switch (C) {
        case 32: return launch_mlp<32>(p, s);
        case 64: return launch_mlp<64>(p, s);
        default: return launch_mlp<96>(p, s);
    }
}
