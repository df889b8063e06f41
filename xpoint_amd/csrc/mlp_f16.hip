// Fused MLP of a VSS block in the fast mixed-precision class:   x += fc2( GELU( fc1( a ) ) )   with a = LayerNorm(x) given (half),
// reference xpoint/models/vmamba_src/VMamba.py:110-128 (Mlp) + :1230-1234 (residual) under autocast: fc1 -> half, GELU -> half, fc2 -> half, add -> half.
// The (M, 4C) hidden activation never leaves the CU: stage 0 writes and re-reads 236 MB of it per block as two GEMMs (fc1 + GELU 136 us, fc2 + residual 92 us
// at 16 images of 480 x 640), and both of those launches are bound by their epilogues, not by the matrix pipe.
//
// CDNA4 mapping.  A workgroup = 4 waves x 32 rows (128 rows of a / x); a wave owns its rows end to end ("row stationary").  The hidden dimension is walked
// in chunks of HC units:
//   fc1, TRANSPOSED:  hT[hid][m] = sum_k W1[hid][k] a[m][k]   — W1 chunk rows as the MFMA "A" operand, the wave's own a rows as "B": the accumulator lane then
//                     holds ROW m and four consecutive hidden units per register group, i.e. exactly what the second GEMM's A operand needs contiguous;
//   epilogue 1:       + b1 -> r16 -> GELU -> r16, four halves packed per ds_write_b64 into the wave's PRIVATE LDS tile H[m][hid] (no barrier: one wave
//                     writes and reads it, LDS operations of a wave execute in order);
//   fc2:              acc2[m][n] += sum_hid H[m][hid] W2[n][hid]   — accumulators (32 rows x C columns per wave: C / 32 tiles) live across all chunks;
//   epilogue 2:       + b2 -> r16 -> + x -> r16, staged through LDS (the wave's own rows of the a image, no longer needed) for 16-byte row-contiguous
//                     residual loads and stores, as csrc/gemm_f16.hip does.
// Operands arrive by LDS-DMA (global_load_lds_dwordx4) into PADDED row images — row stride = row bytes + 16, an odd number of 16-byte slots, so every
// ds_read_b128 of 16 consecutive rows covers all banks; the DMA writes lane-linear, so the lanes that land on a pad slot fetch from a zero page.  The a image
// (128 x C) exists only in the prologue: every wave lifts its rows' operand fragments into registers (C / 16 x 4 VGPRs, resident for all chunks) and the image's
// bytes then serve as the double-buffered W1 / W2 chunk images (the DMA of chunk c + 1 in flight behind the MFMAs and the GELUs of chunk c, one barrier per
// chunk) and, at the end, as the output staging tiles.  32-wide hidden chunks.  LDS 42.5 KB / 124 registers at C = 96 (three workgroups per CU), 70.7 KB / 196
// registers at C = 192 (two): one wave's GELUs run under another's MFMAs (VALU and MFMA of ONE wave do not overlap, coexec_probe).  With the a image resident
// in LDS (first version: 67 / 118 KB, two / one workgroups per CU) the launches took 166 / 196 us at 16 images of 480 x 640; now 143 / 141 us.  Bound by the
// GELU's vector work (118 M evaluations per stage-0 block; identity instead of GELU: 111 us at C = 96, tools/mlp16_dbg.sh), not by HBM.
#include <string>

#include "xp_common.h"
#include "../../include/xpoint_hip.h"

typedef _Float16 m16x8 __attribute__((ext_vector_type(8)));
typedef float m16acc __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* m16_lds_ptr_t;

#ifndef XP_MLP16_DBG
#define XP_MLP16_DBG 0   /* timing experiments only (wrong results): 1 GELU replaced by the identity */
#endif

namespace {

__device__ __attribute__((aligned(64))) unsigned int g_m16_zero_page[16];
__device__ __forceinline__ float m16_r(float v) { return (float)(_Float16)v; }

struct Mlp16Params {
    const _Float16* A;      // (M, C)  LayerNorm(x)
    _Float16* X;            // (M, C)  residual in, result out
    const _Float16* W1;     // (4C, C)
    const _Float16* W2;     // (C, 4C)
    const float* b1; const float* b2;
    int M;
    const float* ln_w; const float* ln_b; float eps;      // ln_w != NULL: a = LayerNorm(X) is computed here (A unused) — VMamba.py:1230 `self.norm2(x)` folded in
};

template <int C>
struct Mlp16Cfg {
    static constexpr int H4 = 4 * C;
    static constexpr int HC = 32;                                // hidden units per chunk (64 was measured with one workgroup per CU: 221 us at C = 96)
    static constexpr int NCH = H4 / HC;
    static constexpr int SPA = C / 8 + 1;                        // 16-byte slots per row of a K = C image (a, W1), incl. the pad slot
    static constexpr int SPH = HC / 8 + 1;                       // ... of a K = HC image (H, W2)
    static constexpr int A_BYTES = 128 * SPA * 16;
    static constexpr int W1_BYTES = HC * SPA * 16;
    static constexpr int W2_BYTES = C * SPH * 16;
    static constexpr int H_BYTES = 4 * 32 * SPH * 16;
    static constexpr int B1_BYTES = H4 * 4;
    static constexpr int pieces(int bytes) { return (bytes + 1023) / 1024; }
    // image offsets; DMA pieces are whole KB, so every image is padded up to one.  The a image lives only in the PROLOGUE (each wave lifts its rows' MFMA
    // fragments into registers: C / 16 x 4 VGPRs) and the 4 x 32 x C output staging only in the EPILOGUE: both alias the W1 / W2 chunk buffers of the main loop.
    static constexpr int W_REGION = 2 * pieces(W1_BYTES) * 1024 + 2 * pieces(W2_BYTES) * 1024;
    static constexpr int Z_BYTES = pieces(A_BYTES) * 1024 > W_REGION ? pieces(A_BYTES) * 1024 : W_REGION;
    static constexpr int OFF_A = 0;
    static constexpr int OFF_W1 = 0;
    static constexpr int OFF_W2 = 2 * pieces(W1_BYTES) * 1024;
    static constexpr int OFF_H = Z_BYTES;
    static constexpr int OFF_B1 = OFF_H + H_BYTES;
    static constexpr int LDS_BYTES = OFF_B1 + B1_BYTES;
    static_assert(4 * 32 * SPA * 16 <= Z_BYTES, "output staging must fit the aliased region");
};

// LayerNorm of a wave's own 32 rows (rows m0 + 32 wave ..) of x straight into its rows of a padded a image: two lanes per row, C / 16 chunks of 8 halves each;
// two-pass statistics in f32 (as xp_layernorm_f16), the store to LDS rounds to fp16 = the half tensor LayerNorm returns under autocast.  Rows past M are zeros.
template <int C>
__device__ __forceinline__ void mlp16_ln_rows(const _Float16* __restrict__ X, const float* __restrict__ ln_w, const float* __restrict__ ln_b, float eps, int M,
                                              int m0, int wave, int lane, unsigned char* a_image, int SPA) {
    constexpr int CPL = C / 16;
    const int r = lane >> 1, hs = lane & 1, grow = m0 + wave * 32 + r;
    const bool rok = grow < M;
    const m16x8* xr = reinterpret_cast<const m16x8*>(X + (int64_t)(rok ? grow : 0) * C) + hs * CPL;
    float v[CPL][8];
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const m16x8 t = xr[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] = (float)t[e];
        sm += ((v[i][0] + v[i][1]) + (v[i][2] + v[i][3])) + ((v[i][4] + v[i][5]) + (v[i][6] + v[i][7]));
    }
    sm += xp_dpp_mov<0xB1>(sm);                              // the row's other lane (quad_perm [1,0,3,2])
    const float mean = sm / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q = fmaf(d, d, q); }
    q += xp_dpp_mov<0xB1>(q);
    const float rstd = 1.f / sqrtf(q / (float)C + eps);
    unsigned char* dst = a_image + (wave * 32 + r) * (SPA * 16) + hs * CPL * 16;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int c8 = hs * CPL + i;
        const float4 w0 = reinterpret_cast<const float4*>(ln_w)[2 * c8], w1v = reinterpret_cast<const float4*>(ln_w)[2 * c8 + 1];
        const float4 b0 = reinterpret_cast<const float4*>(ln_b)[2 * c8], b1v = reinterpret_cast<const float4*>(ln_b)[2 * c8 + 1];
        const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1v.x, w1v.y, w1v.z, w1v.w}, bv[8] = {b0.x, b0.y, b0.z, b0.w, b1v.x, b1v.y, b1v.z, b1v.w};
        m16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = rok ? (_Float16)((v[i][e] - mean) * rstd * wv[e] + bv[e]) : (_Float16)0.f;
        *reinterpret_cast<m16x8*>(dst + i * 16) = o;
    }
}

#ifndef XP_MLP16_WG96
#define XP_MLP16_WG96 3
#endif
#ifndef XP_MLP16_WG192
#define XP_MLP16_WG192 2
#endif
template <int C>
__global__ __launch_bounds__(256, C <= 96 ? XP_MLP16_WG96 : XP_MLP16_WG192) void mlp_f16_kernel(Mlp16Params p) {
    using T = Mlp16Cfg<C>;
    constexpr int HC = T::HC, SPA = T::SPA, SPH = T::SPH, JN = C / 32, JH = HC / 32;
    extern __shared__ __align__(16) unsigned char lds[];
    const int lane = threadIdx.x & 63, fr = lane & 31, fh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m0 = blockIdx.x * 128;
    const char* zero = reinterpret_cast<const char*>(g_m16_zero_page);

    // generic image fill: `rows` rows of `spr - 1` real 16-byte slots (+ 1 pad slot) each, row r from base + r * ld_bytes; rows >= valid are zero
    auto fill = [&](unsigned char* dst, const char* base, int64_t ld_bytes, int rows, int valid, int spr) {
        const int np = (rows * spr * 16 + 1023) / 1024;
        for (int pc = wave; pc < np; pc += 4) {
            const int slot = pc * 64 + lane;
            const int r = slot / spr, c = slot - r * spr;
            const char* src = (r < valid && c < spr - 1) ? base + (int64_t)r * ld_bytes + c * 16 : zero;
            __builtin_amdgcn_global_load_lds(src, (m16_lds_ptr_t)(dst + pc * 1024), 16, 0, 0);
        }
    };
    const int mvalid = min(128, p.M - m0);
    if (p.ln_w == nullptr) fill(lds + T::OFF_A, reinterpret_cast<const char*>(p.A + (int64_t)m0 * C), C * 2, 128, mvalid, SPA);
    else {
        mlp16_ln_rows<C>(p.X, p.ln_w, p.ln_b, p.eps, p.M, m0, wave, lane, lds + T::OFF_A, SPA);
    }
    auto issue_chunk = [&](int ch, int buf) {
        fill(lds + T::OFF_W1 + buf * T::pieces(T::W1_BYTES) * 1024, reinterpret_cast<const char*>(p.W1 + (int64_t)ch * HC * C), C * 2, HC, HC, SPA);
        fill(lds + T::OFF_W2 + buf * T::pieces(T::W2_BYTES) * 1024, reinterpret_cast<const char*>(p.W2 + (int64_t)ch * HC), T::H4 * 2, C, C, SPH);
    };
    for (int i = threadIdx.x; i < T::H4; i += 256) reinterpret_cast<float*>(lds + T::OFF_B1)[i] = p.b1[i];

    m16acc acc2[JN];
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;

    // the wave's a rows as MFMA operand fragments, resident for the whole kernel: lane = row fr, k = 16 ks + 8 fh ..+8
    m16x8 afrag[C / 16];
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                        // the a image is complete (DMA pieces come from every wave)
        const unsigned char* a_rows = lds + T::OFF_A + (wave * 32 + fr) * (SPA * 16) + fh * 16;
#pragma unroll
        for (int ks = 0; ks < C / 16; ++ks) afrag[ks] = *reinterpret_cast<const m16x8*>(a_rows + ks * 32);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                                        // everybody has its fragments: the image's bytes become the W chunk buffers
    }
    issue_chunk(0, 0);
    unsigned char* h_tile = lds + T::OFF_H + wave * 32 * (SPH * 16);
    const float* b1s = reinterpret_cast<const float*>(lds + T::OFF_B1);
    for (int ch = 0; ch < T::NCH; ++ch) {
        const int buf = ch & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of chunk ch (and, the first time, of the a image) have landed
        __syncthreads();                                        // ... everybody's; everybody is done with the other W buffers (chunk ch - 1)
        if (ch + 1 < T::NCH) issue_chunk(ch + 1, buf ^ 1);
        const unsigned char* w1 = lds + T::OFF_W1 + buf * T::pieces(T::W1_BYTES) * 1024 + fr * (SPA * 16) + fh * 16;
        const unsigned char* w2 = lds + T::OFF_W2 + buf * T::pieces(T::W2_BYTES) * 1024 + fr * (SPH * 16) + fh * 16;
        // ---- fc1 (transposed): hT[jh] = W1chunk . a^T ----
        m16acc hT[JH];
#pragma unroll
        for (int jh = 0; jh < JH; ++jh)
#pragma unroll
            for (int r = 0; r < 16; ++r) hT[jh][r] = 0.f;
        {   // W1 fragments of k-step ks + 1 are read before the MFMAs of ks (two register sets); the a fragments are registers
            m16x8 wfr[2][JH];
            auto rd = [&](int ks, int set) {
#pragma unroll
                for (int jh = 0; jh < JH; ++jh) wfr[set][jh] = *reinterpret_cast<const m16x8*>(w1 + jh * 32 * (SPA * 16) + ks * 32);
            };
            rd(0, 0);
#pragma unroll
            for (int ks = 0; ks < C / 16; ++ks) {
                if (ks + 1 < C / 16) rd(ks + 1, (ks + 1) & 1);
#pragma unroll
                for (int jh = 0; jh < JH; ++jh) hT[jh] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfr[ks & 1][jh], afrag[ks], hT[jh], 0, 0, 0);
            }
        }
        // ---- epilogue 1: register r of tile jh = hidden unit 32 jh + (r & 3) + 8 (r >> 2) + 4 fh of row m = fr ----
#pragma unroll
        for (int jh = 0; jh < JH; ++jh)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int hid0 = jh * 32 + 8 * g4 + 4 * fh;
                const float4 bb = *reinterpret_cast<const float4*>(b1s + ch * HC + hid0);
                const float bv[4] = {bb.x, bb.y, bb.z, bb.w};
                union { _Float16 h[4]; uint2 u; } pk;
#pragma unroll
                for (int q = 0; q < 4; ++q) pk.h[q] = (XP_MLP16_DBG & 1) ? (_Float16)(hT[jh][4 * g4 + q] + bv[q]) : (_Float16)xp_gelu_fast(m16_r(hT[jh][4 * g4 + q] + bv[q]));
                *reinterpret_cast<uint2*>(h_tile + fr * (SPH * 16) + hid0 * 2) = pk.u;
            }
        // ---- fc2: acc2[jn] += H . W2chunk^T ----
        {   // H fragments of both k-steps up front; W2 fragments of output tile jn + 1 are read before the MFMAs of tile jn (two register sets — with the
            // a fragments resident, all 2 JN of them up front would not fit two workgroups' registers at C = 192)
            m16x8 hfr[HC / 16], wfr[2][HC / 16];
#pragma unroll
            for (int ks = 0; ks < HC / 16; ++ks) hfr[ks] = *reinterpret_cast<const m16x8*>(h_tile + fr * (SPH * 16) + fh * 16 + ks * 32);
            auto rd = [&](int jn, int set) {
#pragma unroll
                for (int ks = 0; ks < HC / 16; ++ks) wfr[set][ks] = *reinterpret_cast<const m16x8*>(w2 + jn * 32 * (SPH * 16) + ks * 32);
            };
            rd(0, 0);
#pragma unroll
            for (int jn = 0; jn < JN; ++jn) {
                if (jn + 1 < JN) rd(jn + 1, (jn + 1) & 1);
#pragma unroll
                for (int ks = 0; ks < HC / 16; ++ks) acc2[jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hfr[ks], wfr[jn & 1][ks], acc2[jn], 0, 0, 0);
            }
        }
    }
    // ---- epilogue 2: the wave's 32 x C tile -> its own 32 staging rows (the W buffers' bytes, free now) as halves, then row-contiguous + residual -> x ----
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                                                      // every wave is done with the last chunk's W2 image: its bytes become the staging tiles
    unsigned char* ot = lds + T::OFF_A + wave * 32 * (SPA * 16);          // 32 rows x (SPA * 16) bytes: row stride SPA * 16 >= C * 2
    constexpr int RS = SPA * 16;
#pragma unroll
    for (int jn = 0; jn < JN; ++jn) {
        const int cl = jn * 32 + fr;
        const float bi = p.b2[cl];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = (r & 3) + 8 * (r >> 2) + 4 * fh;
            *reinterpret_cast<_Float16*>(ot + rl * RS + cl * 2) = (_Float16)(acc2[jn][r] + bi);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    constexpr int CPR = C / 8;
#pragma unroll
    for (int it = 0; it < (32 * CPR + 63) / 64; ++it) {
        const int idx = it * 64 + lane;
        if ((32 * CPR) % 64 != 0 && idx >= 32 * CPR) break;
        const int rl = idx / CPR, cc = idx - rl * CPR;
        const int grow = m0 + wave * 32 + rl;
        if (grow >= p.M) continue;
        m16x8 v = *reinterpret_cast<const m16x8*>(ot + rl * RS + cc * 16);
        _Float16* xp = p.X + (int64_t)grow * C + cc * 8;
        const m16x8 rv = *reinterpret_cast<const m16x8*>(xp);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (_Float16)((float)v[e] + (float)rv[e]);
        *reinterpret_cast<m16x8*>(xp) = v;
    }
}

template <int C>
int mlp16_launch(const Mlp16Params& p, hipStream_t s) {
    using T = Mlp16Cfg<C>;
    static XpPerDeviceOnce attr_once;
    if (attr_once.need()) {
        XP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_f16_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES));
    }
    const std::string tag = std::string(p.ln_w ? "ln_mlp_fused_f16_c" : "mlp_fused_f16_c") + std::to_string(C);
    XpProfScope prof(tag.c_str(), s, 2.0 * p.M * C * 8.0 * C, 2.0 * 3.0 * p.M * C);
    hipLaunchKernelGGL(mlp_f16_kernel<C>, dim3(xp_cdiv(p.M, 128)), dim3(256), T::LDS_BYTES, s, p);
    XP_LAUNCH_CHECK();
    return XP_OK;
}

// ---- norm + in_proj of a VSS block in one launch (VMamba.py:1225, :649):  y = LayerNorm(x) W^T  (no bias), y (M, N) half.  Same row-stationary layout: the
//      wave's LayerNorm-ed rows go through a padded a image into operand fragments in registers; the image's bytes then take the WHOLE weight matrix
//      (N x C, N <= 192: <= 77 KB) by one LDS-DMA fill and finally the 32 x N staging tile of every wave for 16-byte stores: ONE LDS region, three uses —
//      26 KB at C = 96 (four workgroups per CU), 77 KB at C = 192 (two; 128 KB and one with separate images). ----
struct LnProj16Params { const _Float16* X; _Float16* Y; const _Float16* W; const float* ln_w; const float* ln_b; float eps; int M; };

template <int C>
__global__ __launch_bounds__(256, C <= 96 ? 4 : 2) void ln_proj_f16_kernel(LnProj16Params p) {
    constexpr int N = C, SPA = C / 8 + 1, JN = N / 32;
    extern __shared__ __align__(16) unsigned char lds[];      // ONE region: the a image (prologue) -> the weight image (MFMA phase) -> the output staging tiles
    const int lane = threadIdx.x & 63, fr = lane & 31, fh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m0 = blockIdx.x * 128;
    const char* zero = reinterpret_cast<const char*>(g_m16_zero_page);
    mlp16_ln_rows<C>(p.X, p.ln_w, p.ln_b, p.eps, p.M, m0, wave, lane, lds, SPA);
    // the wave's LayerNorm-ed rows as operand fragments in registers (each wave wrote and reads only its own rows: LDS operations of a wave execute in order)
    m16x8 afrag[C / 16];
    {
        const unsigned char* a_rows = lds + (wave * 32 + fr) * (SPA * 16) + fh * 16;
#pragma unroll
        for (int ks = 0; ks < C / 16; ++ks) afrag[ks] = *reinterpret_cast<const m16x8*>(a_rows + ks * 32);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                                       // every wave has its fragments: the image's bytes become the weight image
    }
    {   // weight image: N rows of SPA slots
        constexpr int np = (N * SPA * 16 + 1023) / 1024;
        for (int pc = wave; pc < np; pc += 4) {
            const int slot = pc * 64 + lane;
            const int r = slot / SPA, c = slot - r * SPA;
            const char* src = (r < N && c < SPA - 1) ? reinterpret_cast<const char*>(p.W) + (int64_t)r * (C * 2) + c * 16 : zero;
            __builtin_amdgcn_global_load_lds(src, (m16_lds_ptr_t)(lds + pc * 1024), 16, 0, 0);
        }
    }
    m16acc acc[JN];
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned char* w_rows = lds + fr * (SPA * 16) + fh * 16;
    {
        m16x8 bfr[2][JN];
        auto rd = [&](int ks, int set) {
#pragma unroll
            for (int j = 0; j < JN; ++j) bfr[set][j] = *reinterpret_cast<const m16x8*>(w_rows + j * 32 * (SPA * 16) + ks * 32);
        };
        rd(0, 0);
#pragma unroll
        for (int ks = 0; ks < C / 16; ++ks) {
            if (ks + 1 < C / 16) rd(ks + 1, (ks + 1) & 1);
#pragma unroll
            for (int j = 0; j < JN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrag[ks], bfr[ks & 1][j], acc[j], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                                           // everybody is done with the weight image: its bytes become the staging tiles
    unsigned char* ot = lds + wave * 32 * (SPA * 16);          // the wave's own 32 staging rows
    constexpr int RS = SPA * 16;
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = (r & 3) + 8 * (r >> 2) + 4 * fh;
            *reinterpret_cast<_Float16*>(ot + rl * RS + (j * 32 + fr) * 2) = (_Float16)acc[j][r];
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    constexpr int CPR = N / 8;
#pragma unroll
    for (int it = 0; it < (32 * CPR + 63) / 64; ++it) {
        const int idx = it * 64 + lane;
        if ((32 * CPR) % 64 != 0 && idx >= 32 * CPR) break;
        const int rl = idx / CPR, cc = idx - rl * CPR;
        const int grow = m0 + wave * 32 + rl;
        if (grow >= p.M) continue;
        *reinterpret_cast<m16x8*>(p.Y + (int64_t)grow * N + cc * 8) = *reinterpret_cast<const m16x8*>(ot + rl * RS + cc * 16);
    }
}

template <int C>
int lnproj16_launch(const LnProj16Params& p, hipStream_t s) {
    constexpr int SPA = C / 8 + 1;
    constexpr int A_IMG = (128 * SPA * 16 + 1023) / 1024 * 1024, W_IMG = (C * SPA * 16 + 1023) / 1024 * 1024;
    constexpr int LDS = A_IMG > W_IMG ? A_IMG : W_IMG;         // one region, three uses (see the kernel)
    static XpPerDeviceOnce attr_once;
    if (attr_once.need()) {
        XP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ln_proj_f16_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    }
    const std::string tag = "ln_proj_f16_c" + std::to_string(C);
    XpProfScope prof(tag.c_str(), s, 2.0 * p.M * C * (double)C, 2.0 * 2.0 * p.M * C);
    hipLaunchKernelGGL(ln_proj_f16_kernel<C>, dim3(xp_cdiv(p.M, 128)), dim3(256), LDS, s, p);
    XP_LAUNCH_CHECK();
    return XP_OK;
}

}  // namespace

extern "C" int xp_mlp_fused_f16_supported(int C, int H4) { return (C == 96 || C == 192 || C == 32 || C == 64) && H4 == 4 * C; }

extern "C" int xp_mlp_fused_f16(const void* a, void* x, const void* W1, const float* b1, const void* W2, const float* b2, int M, int C, int H4, void* stream) {
    XP_CHECK_ARG(a && x && W1 && b1 && W2 && b2, "xp_mlp_fused_f16: null pointer");
    XP_CHECK_ARG(M > 0 && xp_mlp_fused_f16_supported(C, H4), "xp_mlp_fused_f16: unsupported shape C = %d, H4 = %d (C in {32, 64, 96, 192}, H4 = 4 C)", C, H4);
    XP_CHECK_ARG((((uintptr_t)a | (uintptr_t)x | (uintptr_t)W1 | (uintptr_t)W2 | (uintptr_t)b1 | (uintptr_t)b2) & 15) == 0, "xp_mlp_fused_f16: buffers must be 16-byte aligned");
    Mlp16Params p{(const _Float16*)a, (_Float16*)x, (const _Float16*)W1, (const _Float16*)W2, b1, b2, M, nullptr, nullptr, 0.f};
    hipStream_t s = (hipStream_t)stream;
    switch (C) {
        case 32: return mlp16_launch<32>(p, s);
        case 64: return mlp16_launch<64>(p, s);
        case 96: return mlp16_launch<96>(p, s);
        default: return mlp16_launch<192>(p, s);
    }
}

// The same with norm2 folded in:  x += fc2(GELU(fc1(LayerNorm(x))))  (VMamba.py:1230-1234) — one launch and one pass over x less per block.
extern "C" int xp_ln_mlp_fused_f16(void* x, const float* ln_w, const float* ln_b, float eps, const void* W1, const float* b1, const void* W2, const float* b2,
                                   int M, int C, int H4, void* stream) {
    XP_CHECK_ARG(x && ln_w && ln_b && W1 && b1 && W2 && b2, "xp_ln_mlp_fused_f16: null pointer");
    XP_CHECK_ARG(M > 0 && xp_mlp_fused_f16_supported(C, H4), "xp_ln_mlp_fused_f16: unsupported shape C = %d, H4 = %d (C in {32, 64, 96, 192}, H4 = 4 C)", C, H4);
    XP_CHECK_ARG((((uintptr_t)x | (uintptr_t)W1 | (uintptr_t)W2 | (uintptr_t)b1 | (uintptr_t)b2 | (uintptr_t)ln_w | (uintptr_t)ln_b) & 15) == 0, "xp_ln_mlp_fused_f16: buffers must be 16-byte aligned");
    Mlp16Params p{nullptr, (_Float16*)x, (const _Float16*)W1, (const _Float16*)W2, b1, b2, M, ln_w, ln_b, eps};
    hipStream_t s = (hipStream_t)stream;
    switch (C) {
        case 32: return mlp16_launch<32>(p, s);
        case 64: return mlp16_launch<64>(p, s);
        case 96: return mlp16_launch<96>(p, s);
        default: return mlp16_launch<192>(p, s);
    }
}

// norm + in_proj:  y = LayerNorm(x) W^T,  x (M, C) and y (M, C) fp16, W (C, C) fp16; C in {32, 64, 96, 192}.
extern "C" int xp_ln_proj_f16(const void* x, const float* ln_w, const float* ln_b, float eps, const void* W, void* y, int M, int C, void* stream) {
    XP_CHECK_ARG(x && ln_w && ln_b && W && y, "xp_ln_proj_f16: null pointer");
    XP_CHECK_ARG(M > 0 && (C == 32 || C == 64 || C == 96 || C == 192), "xp_ln_proj_f16: C must be 32, 64, 96 or 192 (got %d)", C);
    XP_CHECK_ARG((((uintptr_t)x | (uintptr_t)y | (uintptr_t)W | (uintptr_t)ln_w | (uintptr_t)ln_b) & 15) == 0, "xp_ln_proj_f16: buffers must be 16-byte aligned");
    LnProj16Params p{(const _Float16*)x, (_Float16*)y, (const _Float16*)W, ln_w, ln_b, eps, M};
    hipStream_t s = (hipStream_t)stream;
    switch (C) {
        case 32: return lnproj16_launch<32>(p, s);
        case 64: return lnproj16_launch<64>(p, s);
        case 96: return lnproj16_launch<96>(p, s);
        default: return lnproj16_launch<192>(p, s);
    }
}
