// fp16-storage dense engine of the FAST mixed-precision class ("amp16f"): activations live in HBM as fp16, weights are the fp16 tensors
// autocast casts them to, every multiply is ONE exact fp16 x fp16 product on v_mfma_f32_32x32x16_f16 with f32 accumulation, and the epilogue rounds to fp16
// wherever the reference's autocast recipe (XPoint.py:182 `torch.cuda.amp.autocast()`; op list in DESIGN.md §3e) ends in a half tensor:
//     C = r16( r16( act( r16(acc + bias) ) * scale + shift ) + res )          (act: GELU -> r16, ReLU; absent terms drop out with their rounding)
// i.e. bit for bit the values of the round-3 "amp16" parity class (f32 containers, three-product kernels whose low planes are zero) up to the order of the
// f32 accumulation — the pin is the same fixture, tests/golden/g20 (real reference under float16 autocast, op-level taps).
//
// Why a new kernel and not a mode of gemm_h2: the split engines are bound by f32 -> two-plane staging (split VALU, two planes through LDS, profiles/
// r3_gemm_h2_stalls.txt); with half operands already in HBM none of that exists.  Both operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no
// staging registers, no ds_write, no address VALU beyond one 64-bit add per piece and slab), 64-wide K slabs, two LDS buffers, one LDS-only barrier per slab
// with the DMA of slab t+1 in flight behind the 16 MFMAs of slab t.  LDS image: 128-byte rows (64 halves), the eight 16-byte slots of row r XOR-permuted by
// (r >> 1) & 7 — the permutation is applied on the per-lane GLOBAL address (the DMA writes lane-linear: wave-uniform base + lane * 16) and again on the
// fragment read, so every ds_read_b128 of 16 consecutive rows covers all 64 banks.  Rows / columns / k past the matrix are fetched from a 64-byte zero page
// (per-lane source addresses make that free), so there is no tail code and no masking in the loop.
// Epilogue: the wave's tile goes to LDS as fp16 (after the K loop the staging buffers are free) and comes back row-contiguous, so residual loads and output
// stores are 16 bytes per lane in full 128-byte row segments (an accumulator lane holds a COLUMN: storing from registers would write 2-byte elements).
#include <string>
#include <type_traits>

#include "xp_common.h"
#include "../../include/xpoint_hip.h"

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f16acc __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* f16_lds_ptr_t;

namespace {

// Slab depth BK (halves): 64 (128-byte LDS rows, 8 slots, rows permuted by (r >> 1) & 7) for the K loops that keep the matrix pipe busy; 32 (64-byte rows, 4 slots,
// permuted by (r >> 2) & 3) for the SHORT contractions (K <= 192: stage 0 / 1, where a tile's life is DMA latency + epilogue): half the LDS per workgroup — four
// workgroups per CU instead of two — and K = 96 is three whole slabs instead of one and a half.

struct F16Params {
    const _Float16* A; const _Float16* W; void* C;
    const float* bias; const float* scale; const float* shift; const _Float16* res;
    int M, N, K, lda, ldc, ldres, act, c_f32;
    // implicit 3x3 convolution over NHWC halves (MODE 1): K order (kh, kw, ci), zero or reflection padding
    int Hi, Wi, Ci, Ho, Wo, stride, reflect;
    unsigned ci_magic;        // ceil(2^32 / Ci): k / Ci = umulhi(k, ci_magic) for every k < 2^16
};

__device__ __attribute__((aligned(64))) unsigned int g_f16_zero_page[16];      // zero-initialised: the source of every out-of-range 16-byte piece

__device__ __forceinline__ float f16_r(float v) { return (float)(_Float16)v; }

template <int WM, int WN, int TM, int TN, int BK>
struct F16Tile {
    static constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64, NW = WM * WN;
    static constexpr int ROWB = BK * 2, SLOTS = BK / 8, RPP = 1024 / ROWB;       // bytes per LDS row, 16-byte slots per row, rows per 1-KB DMA piece
    static constexpr int PA = BM / RPP, PB = BN / RPP;              // DMA pieces per slab
    static constexpr int PA_W = (PA + NW - 1) / NW, PB_W = (PB + NW - 1) / NW;
    static constexpr int kBufBytes = (BM + BN) * ROWB;
    // slot permutation of row r (the same involution on the DMA source address and on the fragment read): 16 consecutive rows at one logical slot cover all banks
    __device__ static __forceinline__ int swz(int r) { return BK == 64 ? ((r >> 1) & 7) : ((r >> 2) & 3); }
    static constexpr int WROWS = TM * 32, WCOLS = TN * 32;          // a wave's output tile
    static constexpr int kOutBytes = NW * WROWS * WCOLS * 2;
    static constexpr size_t kLdsBytes = (2 * kBufBytes > kOutBytes ? 2 * kBufBytes : kOutBytes);
};

// MODE 0: plain A (M, lda).  MODE 1: implicit 3x3 convolution.
template <int WM, int WN, int TM, int TN, int MODE, int BK>
__global__ __launch_bounds__(WM * WN * 64, (BK == 64 ? 2 : 4) / (WM * WN > 4 ? 2 : 1)) void gemm_f16_kernel(F16Params p) {
    using T = F16Tile<WM, WN, TM, TN, BK>;
    constexpr int F16_BK = BK, F16_ROWB = T::ROWB;
    extern __shared__ __align__(16) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile order (cdna_hip_programming.md T1, bijective form): workgroup ids go round-robin to the 8 XCDs; every XCD gets one contiguous run of
    // logical tiles, column tiles of one row tile adjacent (they share the activation rows in that XCD's L2)
    const int ntn = (p.N + T::BN - 1) / T::BN, ntm = (p.M + T::BM - 1) / T::BM, nt = ntn * ntm;
    int tile;
    {
        const int b = blockIdx.x, xcd = b & 7, q = nt >> 3, r = nt & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int m0 = (tile / ntn) * T::BM, n0 = (tile % ntn) * T::BN;
    const char* zero = reinterpret_cast<const char*>(g_f16_zero_page);

    // ---- DMA plan: piece = 1 KB = RPP tile rows; lane l of the issuing wave writes LDS row (l / SLOTS), physical slot (l % SLOTS), and therefore FETCHES the
    //      logical slot (l % SLOTS) ^ swz(row) of that row ----
    const char* a_ptr[T::PA_W]; int a_klim[T::PA_W];          // MODE 0: byte pointer to (row, logical slot) at k = 0; k limit: slab t is real iff 64 t < klim
    const char* b_ptr[T::PB_W]; int b_klim[T::PB_W];
    int a_ls[T::PA_W];                                         // logical slot (MODE 1 needs it per slab)
    int a_pix[T::PA_W];                                        // MODE 1: (image, oh, ow) packed as linear output pixel, -1 past M
#pragma unroll
    for (int i = 0; i < T::PA_W; ++i) {
        const int piece = wave + i * T::NW;
        const int row = piece * T::RPP + lane / T::SLOTS;
        const int ls = (lane % T::SLOTS) ^ T::swz(row);
        const int grow = m0 + row;
        const bool ok = piece < T::PA && grow < p.M;
        a_ls[i] = ls;
        if (MODE == 0) {
            a_ptr[i] = reinterpret_cast<const char*>(p.A + (int64_t)(ok ? grow : 0) * p.lda + ls * 8);
            a_klim[i] = ok ? p.K - ls * 8 : 0;
        } else {
            a_ptr[i] = nullptr; a_klim[i] = 0;
            a_pix[i] = ok ? grow : -1;
        }
    }
#pragma unroll
    for (int i = 0; i < T::PB_W; ++i) {
        const int piece = wave + i * T::NW;
        const int row = piece * T::RPP + lane / T::SLOTS;
        const int ls = (lane % T::SLOTS) ^ T::swz(row);
        const int gcol = n0 + row;
        const bool ok = piece < T::PB && gcol < p.N;
        b_ptr[i] = reinterpret_cast<const char*>(p.W + (int64_t)(ok ? gcol : 0) * p.K + ls * 8);
        b_klim[i] = ok ? p.K - ls * 8 : 0;
    }
    // MODE 1: decompose the output pixel once
    int c_n[T::PA_W], c_oh[T::PA_W], c_ow[T::PA_W];
    if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < T::PA_W; ++i) {
            const int g = a_pix[i] < 0 ? 0 : a_pix[i];
            c_n[i] = g / (p.Ho * p.Wo);
            const int rem = g - c_n[i] * (p.Ho * p.Wo);
            c_oh[i] = rem / p.Wo; c_ow[i] = rem - c_oh[i] * p.Wo;
        }
    }
    auto issue = [&](int t, int buf) {
        unsigned char* base = lds + buf * T::kBufBytes;
#pragma unroll
        for (int i = 0; i < T::PA_W; ++i) {
            const int piece = wave + i * T::NW;
            if (T::PA % T::NW != 0 && piece >= T::PA) continue;
            const char* src;
            if (MODE == 0) src = t * F16_BK < a_klim[i] ? a_ptr[i] + (int64_t)t * (F16_BK * 2) : zero;
            else {
                const int k = t * F16_BK + a_ls[i] * 8;                 // 8 consecutive k = 8 channels of one tap (Ci % 8 == 0)
                const int tap = (int)__umulhi((unsigned)k, p.ci_magic), ci = k - tap * p.Ci;
                const int kh = (tap * 11) >> 5, kw = tap - kh * 3;                // tap / 3 for tap in 0..8 (taps past 8 only occur with k >= K)
                int ih = c_oh[i] * p.stride - 1 + kh, iw = c_ow[i] * p.stride - 1 + kw;
                bool ok = a_pix[i] >= 0 && k < p.K;
                if (p.reflect) { ih = ih < 0 ? -ih : (ih >= p.Hi ? 2 * p.Hi - 2 - ih : ih); iw = iw < 0 ? -iw : (iw >= p.Wi ? 2 * p.Wi - 2 - iw : iw); }
                else ok = ok && ih >= 0 && ih < p.Hi && iw >= 0 && iw < p.Wi;
                src = ok ? reinterpret_cast<const char*>(p.A + (((int64_t)c_n[i] * p.Hi + ih) * p.Wi + iw) * p.Ci + ci) : zero;
            }
            __builtin_amdgcn_global_load_lds(src, (f16_lds_ptr_t)(base + piece * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < T::PB_W; ++i) {
            const int piece = wave + i * T::NW;
            if (T::PB % T::NW != 0 && piece >= T::PB) continue;
            const char* src = t * F16_BK < b_klim[i] ? b_ptr[i] + (int64_t)t * (F16_BK * 2) : zero;
            __builtin_amdgcn_global_load_lds(src, (f16_lds_ptr_t)(base + T::BM * F16_ROWB + piece * 1024), 16, 0, 0);
        }
    };

    f16acc acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses: lane (fr, h) reads the 8 halves k = 16 ks + 8 h .. + 7 of row (base + fr): logical slot 2 ks + h, physical slot = logical ^ swz(fr)
    // (tile row bases are multiples of 32, so swz(row) = swz(fr))
    const int fr = lane & 31, fh = lane >> 5;
    const int cx = (fh ^ T::swz(fr)) << 4;
    const int a_frag = (wm * TM * 32 + fr) * F16_ROWB, b_frag = (T::BM + wn * TN * 32 + fr) * F16_ROWB;

    const int nslab = (p.K + F16_BK - 1) / F16_BK;
    issue(0, 0);
    for (int t = 0; t < nslab; ++t) {
        const int buf = t & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's pieces of slab t have landed ...
        __builtin_amdgcn_s_barrier();                            // ... and everybody's; everybody is done reading the other buffer (slab t - 1)
        if (t + 1 < nslab) issue(t + 1, buf ^ 1);
        const unsigned char* bb = lds + buf * T::kBufBytes;
        // fragments of k-step ks + 1 are read before the MFMAs of k-step ks (two register sets): the LDS latency sits behind matrix work
        h16x8 af[2][TM], bf[2][TN];
        auto frags = [&](int ks, int set) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[set][i] = *reinterpret_cast<const h16x8*>(bb + a_frag + i * 32 * F16_ROWB + (cx ^ (ks << 5)));
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[set][j] = *reinterpret_cast<const h16x8*>(bb + b_frag + j * 32 * F16_ROWB + (cx ^ (ks << 5)));
        };
        frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            if (ks + 1 < BK / 16) frags(ks + 1, (ks + 1) & 1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks & 1][i], bf[ks & 1][j], acc[i][j], 0, 0, 0);
        }
    }
    // ---- epilogue ----
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                // the staging buffers are free: every wave turns its part of them into an output tile
    constexpr int RS = T::WCOLS * 2;                             // bytes per row of the wave's fp16 tile
    unsigned char* ot = lds + wave * (T::WROWS * RS);
    auto run = [&](auto act_tag) {
        constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int cl = j * 32 + fr, col = n0 + wn * T::WCOLS + cl;
                const int cc = col < p.N ? col : p.N - 1;
                const float bi = p.bias ? p.bias[cc] : 0.f;
                const float sc = p.scale ? p.scale[cc] : 1.f, sh = p.shift ? p.shift[cc] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    float v = f16_r(acc[i][j][r] + bi);                       // the layer's half output (f32 accumulate + bias, one rounding)
                    if (ACT == 1) v = f16_r(xp_gelu_fast(v));
                    if (ACT == 2) v = fmaxf(v, 0.f);
                    if (p.scale) v = f16_r(v * sc + sh);                      // eval BatchNorm on a half tensor returns a half tensor
                    if (ACT == 3) v = fmaxf(v, 0.f);
                    // rows r and r + 4 (the two lane halves) sit 4 rows apart: XOR bit 6 of the byte offset with bit 2 of the row so they use different banks
                    const int off = rl * RS + ((cl * 2) ^ (RS % 128 == 0 ? ((rl >> 2) & 1) << 6 : 0));
                    *reinterpret_cast<_Float16*>(ot + off) = (_Float16)v;
                }
            }
    };
    switch (p.act) {
        case 1: run(std::integral_constant<int, 1>{}); break;
        case 2: run(std::integral_constant<int, 2>{}); break;
        case 3: run(std::integral_constant<int, 3>{}); break;
        default: run(std::integral_constant<int, 0>{}); break;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the wave reads back only its own tile: no barrier
    constexpr int CPR = T::WCOLS / 8;                            // 16-byte chunks per tile row
    const int mrow0 = m0 + wm * T::WROWS, ncol0 = n0 + wn * T::WCOLS;
    const bool vec = (p.ldc % 8 == 0) && (!p.res || p.ldres % 8 == 0);      // 16-byte accesses to C / res rows are aligned
#pragma unroll
    for (int it = 0; it < (T::WROWS * CPR + 63) / 64; ++it) {
        const int idx = it * 64 + lane;
        if ((T::WROWS * CPR) % 64 != 0 && idx >= T::WROWS * CPR) break;
        const int rl = idx / CPR, ch = idx - rl * CPR;
        const int grow = mrow0 + rl, gcol = ncol0 + ch * 8;
        if (grow >= p.M || gcol >= p.N) continue;
        const int off = rl * RS + ((ch * 16) ^ (RS % 128 == 0 ? ((rl >> 2) & 1) << 6 : 0));
        h16x8 v = *reinterpret_cast<const h16x8*>(ot + off);
        const bool full = gcol + 8 <= p.N;
        if (p.res) {
            const _Float16* rp = p.res + (int64_t)grow * p.ldres + gcol;
            if (full && vec) {
                const h16x8 rv = *reinterpret_cast<const h16x8*>(rp);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (_Float16)((float)v[e] + (float)rv[e]);        // half + half -> half
            } else {
                for (int e = 0; e < 8 && gcol + e < p.N; ++e) v[e] = (_Float16)((float)v[e] + (float)rp[e]);
            }
        }
        if (p.c_f32) {
            float* cp = reinterpret_cast<float*>(p.C) + (int64_t)grow * p.ldc + gcol;
            if (full && p.ldc % 4 == 0) {
                *reinterpret_cast<float4*>(cp) = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
                *reinterpret_cast<float4*>(cp + 4) = make_float4((float)v[4], (float)v[5], (float)v[6], (float)v[7]);
            } else {
                for (int e = 0; e < 8 && gcol + e < p.N; ++e) cp[e] = (float)v[e];
            }
        } else {
            _Float16* cp = reinterpret_cast<_Float16*>(p.C) + (int64_t)grow * p.ldc + gcol;
            if (full && vec) *reinterpret_cast<h16x8*>(cp) = v;
            else for (int e = 0; e < 8 && gcol + e < p.N; ++e) cp[e] = v[e];
        }
    }
}

template <int WM, int WN, int TM, int TN, int BK>
void f16_launch(const F16Params& p, hipStream_t s) {
    using T = F16Tile<WM, WN, TM, TN, BK>;
    static XpPerDeviceOnce attr_once;
    if (attr_once.need()) {
        XP_HIP_WARN(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16_kernel<WM, WN, TM, TN, 0, BK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::kLdsBytes));
        XP_HIP_WARN(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16_kernel<WM, WN, TM, TN, 1, BK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::kLdsBytes));
    }
    const int grid = xp_cdiv(p.M, T::BM) * xp_cdiv(p.N, T::BN);
    static const bool by_shape = getenv("XP_PROF_SHAPES") != nullptr;
    std::string tag = std::string(p.Ci ? "conv3x3_f16_mfma_" : "gemm_f16_mfma_") + std::to_string(T::BM) + "x" + std::to_string(T::BN) + (BK == 32 ? "_k32" : "");
    if (by_shape) tag += "_M" + std::to_string(p.M) + "_N" + std::to_string(p.N) + "_K" + std::to_string(p.K);
    const double in_elems = p.Ci ? (double)p.M / (p.Ho * p.Wo) * p.Hi * p.Wi * p.Ci : (double)p.M * p.K;
    XpProfScope prof(tag.c_str(), s, 2.0 * p.M * p.N * p.K, 2.0 * (in_elems + (double)p.N * p.K + (double)p.M * p.N * (p.res ? 2 : 1)) + (p.c_f32 ? 2.0 * p.M * p.N : 0.0));
    if (p.Ci) hipLaunchKernelGGL((gemm_f16_kernel<WM, WN, TM, TN, 1, BK>), dim3(grid), dim3(T::NT), T::kLdsBytes, s, p);
    else hipLaunchKernelGGL((gemm_f16_kernel<WM, WN, TM, TN, 0, BK>), dim3(grid), dim3(T::NT), T::kLdsBytes, s, p);
}

int f16_dispatch(const F16Params& p, hipStream_t s) {
    static const int force = getenv("XP_F16_TILE") ? atoi(getenv("XP_F16_TILE")) : -1;      // tuning experiments only
    const int N = p.N;
    // Tile by LAYER (N, K) only — every tile walks K in the same slab order, so the choice never changes a result bit, and it never depends on the batch.
    // Measured on the deep-stage layers (tools/gemm_bench.py, GB_F16=1, XP_F16_TILE=3/4/5): 128 x 192 wins where it covers N in fewer column tiles at short K
    // (N 192, K 768: 54.1 -> 49.8 us; N 1536, K 384 + GELU: 64.3 -> 57.4), 256 x 128 where K is long and N narrow (N 384, K 1536: 40.5 -> 35.7: the L2-read-bound
    // case of profiles/r4_03_gemm_l2_counters.txt, A re-read halves); 128 x 128 everywhere else (M 4 800 rows: larger tiles leave CUs idle).
    int sel = N <= 32 ? 0 : N <= 64 ? 1 : (N <= 96 || (N % 96 == 0 && (N / 96) % 4 != 0)) ? 2 : 3;
    if (!p.Ci && p.K > 192) {
        // x_proj of the deep stages (N = 4 (dt_rank + 2) = 104 / 200: one or two 128-wide column tiles over 19 200 / 4 800 rows leave most CUs idle): 128 x 32
        // tiles — 9.7 -> 8.1 us and 12.8 -> 8.4 us
        if (N > 96 && N < 256 && N % 32 != 0) sel = 0;
        if ((N == 192 && p.K >= 512) || (N == 1536 && p.K <= 512)) sel = 4;
        else if (N == 384 && p.K >= 1024) sel = 5;
    }
    if (force >= 0) sel = force;
    // slab depth: a property of the layer (K), never of the batch.  K <= 192 (plain GEMMs of stages 0 / 1, whose tiles live on DMA latency + epilogue): 32
    static const int force_bk = getenv("XP_F16_BK") ? atoi(getenv("XP_F16_BK")) : 0;
    const bool k32 = force_bk ? force_bk == 32 : (p.K <= 192 && !p.Ci);
    if (k32) {
        switch (sel) {
            case 0: f16_launch<4, 1, 1, 1, 32>(p, s); break;
            case 1: f16_launch<4, 1, 1, 2, 32>(p, s); break;
            case 2: f16_launch<4, 1, 1, 3, 32>(p, s); break;
            default: f16_launch<2, 2, 2, 2, 32>(p, s); break;
        }
    } else {
        switch (sel) {
            case 0: f16_launch<4, 1, 1, 1, 64>(p, s); break;       // 128 x 32
            case 1: f16_launch<4, 1, 1, 2, 64>(p, s); break;       // 128 x 64
            case 2: f16_launch<4, 1, 1, 3, 64>(p, s); break;       // 128 x 96
            case 4: f16_launch<2, 2, 2, 3, 64>(p, s); break;       // 128 x 192 (experiment, XP_F16_TILE=4): A re-read N / 192 times instead of N / 128
            case 5: f16_launch<4, 2, 2, 2, 64>(p, s); break;       // 256 x 128, 8 waves (experiment, XP_F16_TILE=5)
            default: f16_launch<2, 2, 2, 2, 64>(p, s); break;      // 128 x 128
        }
    }
    XP_LAUNCH_CHECK();
    return XP_OK;
}

__global__ __launch_bounds__(256) void f32_to_f16_kernel(const float* __restrict__ x, _Float16* __restrict__ y, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 4 <= n) {
        const float4 v = *reinterpret_cast<const float4*>(x + i);
        union { _Float16 h[4]; uint2 u; } o;
        o.h[0] = (_Float16)v.x; o.h[1] = (_Float16)v.y; o.h[2] = (_Float16)v.z; o.h[3] = (_Float16)v.w;
        *reinterpret_cast<uint2*>(y + i) = o.u;
    } else {
        for (int64_t j = i; j < n; ++j) y[j] = (_Float16)x[j];
    }
}

}  // namespace

extern "C" int xp_f32_to_f16(const float* x, void* y, int64_t n, void* stream) {
    XP_CHECK_ARG(x && y && n >= 0, "xp_f32_to_f16: bad args");
    XP_CHECK_ARG((((uintptr_t)x & 15) | ((uintptr_t)y & 7)) == 0, "xp_f32_to_f16: x must be 16-byte, y 8-byte aligned");
    if (n == 0) return XP_OK;
    hipLaunchKernelGGL(f32_to_f16_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, x, reinterpret_cast<_Float16*>(y), n);
    XP_LAUNCH_CHECK();
    return XP_OK;
}

extern "C" int xp_gemm_nt_f16(const void* A, const void* W, void* C, int c_f32, const float* bias, const float* scale, const float* shift, const void* res,
                              int M, int N, int K, int lda, int ldc, int ldres, int act, void* stream) {
    XP_CHECK_ARG(A && W && C, "xp_gemm_nt_f16: null pointer");
    XP_CHECK_ARG(M > 0 && N > 0 && K > 0, "xp_gemm_nt_f16: bad shape %d %d %d", M, N, K);
    XP_CHECK_ARG(K % 8 == 0 && lda % 8 == 0, "xp_gemm_nt_f16: K and lda must be multiples of 8 halves (got %d, %d)", K, lda);
    XP_CHECK_ARG((((uintptr_t)A | (uintptr_t)W) & 15) == 0, "xp_gemm_nt_f16: A and W must be 16-byte aligned");
    XP_CHECK_ARG((scale == nullptr) == (shift == nullptr), "xp_gemm_nt_f16: scale and shift go together");
    XP_CHECK_ARG(act >= 0 && act <= 3, "xp_gemm_nt_f16: bad act %d", act);
    XP_CHECK_ARG(((uintptr_t)C & 15) == 0 && (!res || ((uintptr_t)res & 15) == 0), "xp_gemm_nt_f16: C and res must be 16-byte aligned");
    F16Params p{};
    p.A = (const _Float16*)A; p.W = (const _Float16*)W; p.C = C; p.bias = bias; p.scale = scale; p.shift = shift; p.res = (const _Float16*)res;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldc = ldc; p.ldres = ldres; p.act = act; p.c_f32 = c_f32;
    return f16_dispatch(p, (hipStream_t)stream);
}

extern "C" int xp_conv3x3_nhwc_f16(const void* x, const void* W, void* y, int y_f32, const float* bias, const float* scale, const float* shift,
                                   int batch, int Hi, int Wi, int Ci, int Co, int stride, int reflect_pad, int act, void* stream) {
    XP_CHECK_ARG(x && W && y, "xp_conv3x3_nhwc_f16: null pointer");
    XP_CHECK_ARG(Ci % 8 == 0, "xp_conv3x3_nhwc_f16: Ci must be a multiple of 8 (got %d)", Ci);
    XP_CHECK_ARG(stride == 1 || stride == 2, "xp_conv3x3_nhwc_f16: stride 1 or 2");
    XP_CHECK_ARG((scale == nullptr) == (shift == nullptr), "xp_conv3x3_nhwc_f16: scale and shift go together");
    XP_CHECK_ARG(!reflect_pad || (Hi >= 2 && Wi >= 2), "xp_conv3x3_nhwc_f16: reflection pad needs H,W >= 2");
    XP_CHECK_ARG((((uintptr_t)x | (uintptr_t)W | (uintptr_t)y) & 15) == 0, "xp_conv3x3_nhwc_f16: buffers must be 16-byte aligned");
    F16Params p{};
    p.A = (const _Float16*)x; p.W = (const _Float16*)W; p.C = y; p.bias = bias; p.scale = scale; p.shift = shift; p.res = nullptr;
    p.Hi = Hi; p.Wi = Wi; p.Ci = Ci; p.stride = stride; p.reflect = reflect_pad;
    p.Ho = (Hi + 2 - 3) / stride + 1; p.Wo = (Wi + 2 - 3) / stride + 1;
    p.ci_magic = (unsigned)((0x100000000ull + (unsigned)Ci - 1) / (unsigned)Ci);
    XP_CHECK_ARG(9 * Ci + 64 < 65536, "xp_conv3x3_nhwc_f16: Ci too large");
    p.M = batch * p.Ho * p.Wo; p.N = Co; p.K = 9 * Ci; p.lda = 0; p.ldc = Co; p.ldres = 0; p.act = act; p.c_f32 = y_f32;
    return f16_dispatch(p, (hipStream_t)stream);
}
