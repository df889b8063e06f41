// Shared by the exact-f32 (gemm.hip) and split-bf16 (gemm_x3.hip) GEMM kernels: launch parameters and the fused
// epilogue   C[m, n] = (act(acc + bias[n]) * scale[n] + shift[n]) + res[m, n].
#pragma once
#include <type_traits>

#include "xp_common.h"

#ifndef XP_EPI_DBG
#define XP_EPI_DBG 0   /* timing experiment: 1 = compute the epilogue but store (almost) nothing */
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmParams {
    const float* A;
    const float* Wt;      // (N, K) row-major f32, or the split-bf16 planes written by xp_split_weights_x3
    float* C;
    const float* bias;    // (N) or null
    const float* scale;   // (N) or null   (applied after the activation; eval-mode BatchNorm)
    const float* shift;
    const float* res;     // (M, ldres) or null
    const float* wscale;  // (N) or null: per-output-column factor applied to the accumulator before the bias (split-fp16 engine: undoes the
                          // power-of-two row scale of the offline weight split, exactly)
    int M, N, K;
    int lda, ldc, ldres;
    int act;              // 0 none, 1 GELU(erf), 2 ReLU (before scale/shift), 3 ReLU after scale/shift
    int ngroup;           // split-fp16 tile engine: column tiles per group of the tile order (0 = N-fastest), gemm_h2.hip
    int mode;             // 0: plain A; 1: implicit 3x3 conv over NHWC (K order = kh, kw, ci)
    int Hi, Wi, Ci, Ho, Wo, stride, reflect;
    int r16;              // split-fp16 engine only: round the value to fp16 after every operation autocast would end in a half tensor — bias add,
                          // GELU, the BatchNorm affine, the residual add (xp_set_amp_mode; the mixed-precision class of DESIGN.md §3e)
    int stagger_cycles;   // start-up delay of the second workgroup slot of every CU (see gemm_kernel)
    unsigned long long* stamps;   // debug (XP_GEMM_STAMPS): 4 s_memtime stamps per workgroup, else null
};


// tile engines whose accumulators carry a per-column power-of-two factor declare `static constexpr bool kRowScale = true`
template <class T, class = void> struct tile_has_row_scale : std::false_type {};
template <class T> struct tile_has_row_scale<T, std::enable_if_t<T::kRowScale>> : std::true_type {};

__device__ __forceinline__ float xp_r16(float v) { return (float)(_Float16)v; }       // round to nearest even to fp16 and back (v_cvt_f16_f32, v_cvt_f32_f16)

// T: tile engine providing BM, BN, row_of(i, r), col_of(j).  acc[i][j] are the lane's 32x32 accumulator tiles.
// ALLOW_R16: also instantiate the p.r16 form (the split-fp16 kernels; the other engines never set it).
template <class T, int TM, int TN, bool ALLOW_R16 = false>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, int m0, int n0, f32x16 (&acc)[TM][TN]) {
    // Epilogue, straight-line: the activation is a compile-time tag and interior tiles skip every bounds test (per-element
    // runtime switches made hipcc emit ~3 scalar branches per element: ~190 cycles per stored value).  Absent scale /
    // shift / residual are the exact identities (x*1+0, +0).  Two phases so that every residual load is in flight before
    // the first store (C may alias res: a load-add-store chain per element would serialise the L2 round trips).
    const bool interior = (m0 + T::BM <= p.M) && (n0 + T::BN <= p.N);
    auto epilogue = [&](auto act_tag, auto interior_tag, auto r16_tag) {
        constexpr int ACT = decltype(act_tag)::value;
        constexpr bool INTERIOR = decltype(interior_tag)::value;
        constexpr bool R16 = decltype(r16_tag)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                // one 32x32 accumulator tile at a time: its 16 residual loads are in flight together, then its 16 stores
                // (keeps the live registers at one tile; C may alias res, so loads of the next tile stay behind these stores)
                // addresses = uniform 64-bit tile base (SGPRs) + 32-bit lane offset: no 64-bit VALU per element
                const int cl = T::col_of(j), col = n0 + cl;
                const bool cok = INTERIOR || col < p.N;
                const int clc = cok ? cl : 0;
                const float bi = p.bias ? p.bias[n0 + clc] : 0.f;
                const float ws = tile_has_row_scale<T>::value ? p.wscale[n0 + clc] : 1.f;
                const float sc = p.scale ? p.scale[n0 + clc] : 1.f;
                const float sh = p.shift ? p.shift[n0 + clc] : 0.f;
                const float* resb = p.res ? p.res + (int64_t)m0 * p.ldres + n0 : nullptr;
                float* cb = p.C + (int64_t)m0 * p.ldc + n0;
                float rv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = T::row_of(i, r);
                    const int rlc = (INTERIOR || m0 + rl < p.M) ? rl : 0;
                    rv[r] = resb ? resb[rlc * p.ldres + clc] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = tile_has_row_scale<T>::value ? acc[i][j][r] * ws + bi : acc[i][j][r] + bi;    // ws is a power of two: the product is exact
                    if (R16) v = xp_r16(v);                                     // the layer's half output
                    if (ACT == 1) { v = xp_gelu_fast(v); if (R16) v = xp_r16(v); }
                    if (ACT == 2) v = fmaxf(v, 0.f);
                    v = v * sc + sh;
                    if (R16) { if (p.scale) v = xp_r16(v); }                    // eval BatchNorm on a half tensor returns a half tensor
                    if (ACT == 3) v = fmaxf(v, 0.f);
                    rv[r] = rv[r] + v;
                    if (R16) { if (resb) rv[r] = xp_r16(rv[r]); }               // half + half -> half
                }
                if (cok && !(XP_EPI_DBG && rv[0] != 12345.678f)) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rl = T::row_of(i, r);
                        if (INTERIOR || m0 + rl < p.M) cb[rl * p.ldc + cl] = rv[r];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    auto by_act = [&](auto interior_tag, auto r16_tag) {
        switch (p.act) {
            case 1: epilogue(std::integral_constant<int, 1>{}, interior_tag, r16_tag); break;
            case 2: epilogue(std::integral_constant<int, 2>{}, interior_tag, r16_tag); break;
            case 3: epilogue(std::integral_constant<int, 3>{}, interior_tag, r16_tag); break;
            default: epilogue(std::integral_constant<int, 0>{}, interior_tag, r16_tag); break;
        }
    };
    if constexpr (ALLOW_R16) {
        if (p.r16) {       // the rounding class is rare: the general (bounds-tested) form serves interior tiles too
            by_act(std::false_type{}, std::true_type{});
            return;
        }
    }
    if (interior) by_act(std::true_type{}, std::false_type{}); else by_act(std::false_type{}, std::false_type{});
}
