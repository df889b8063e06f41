// "Ring" dense engine (round 5): one tile kernel for BOTH dense classes — the one-product fp16 class (amp16f) and the three-product split-fp16 class
// (f32-grade, the headline class) — in which NOTHING but matrix instructions runs between the LDS and the accumulators.
//
// Operand images.  Every operand reaches the kernel as rows of 128-byte "slab rows": (row, slab) -> 8 slots of 16 bytes.
//     PLANES = 1 (fp16 class):  slab = 64 consecutive k of a row-major fp16 matrix; slot s = k 8s .. 8s+7.
//     PLANES = 2 (split class): slab = 32 consecutive k as TWO fp16 planes [hi: 32 halves][lo: 32 halves], x = hi + lo to within one f32 ulp
//                               (gemm_h2_core.h); slots 0..3 = hi, 4..7 = lo.  Weights: the offline layout of xp_split_weights_h2 ([slab][n][plane][32]).
//                               Activations: the "P32" image [row][slab][plane][32] written by the PRODUCER's epilogue (4 bytes per element, like f32):
//                               the f32 -> two-plane split that the round-2..4 kernels did in the K loop (VALU + ds_write staging, the measured
//                               bottleneck: profiles/r3_gemm_h2_stage_removal.txt) now happens once, where the value is produced.
// Both operands therefore go global -> LDS by LDS-DMA (global_load_lds_dwordx4; per-lane source address, lane-linear destination), 1 KB pieces of 8 rows,
// into a ring of S stage buffers; the slot permutation slot ^ ((row >> 1) & 7) is applied on the source address and again on the fragment read
// (cdna_hip_programming.md rule 21), which makes every ds_read_b128 of 32 consecutive rows conflict-free.
//
// Schedule ("ping-pong", cdna_hip_programming.md §5 8-phase template reduced to two phases per slab).  A workgroup is 8 waves = two groups of four;
// wave w and wave w + 4 share a SIMD.  Group 0 owns the upper half of the tile's rows, group 1 the lower half; both walk every slab, half a period apart:
//     phase 2t   : group 0 MEM(t)  = fragment reads of slab t into registers      | group 1 MFMA(t-1) = matrix instructions on registers only
//     phase 2t+1 : group 0 MFMA(t)                                                | group 1 MEM(t)
// separated by workgroup barriers, so each SIMD's matrix pipe always has one wave feeding it and the other wave's LDS reads, DMA issue and waits sit
// in that shadow.  DMA of slab t + S - 1 (group 0) / t + S (group 1) is issued INSIDE the wave's MFMA phase, one 1-KB piece per matrix-instruction gap
// (saddr + voffset addressing: no vector arithmetic per piece), into the buffer whose last reader finished at least a barrier earlier; each wave retires
// its own pieces with a COUNTED vmcnt one phase before the first read of the slab (never vmcnt(0) in the loop, raw s_barrier: the DMA stays in flight
// across barriers).
//
// Accumulation order per output element: slabs in order; inside a slab k-step 0 then 1; inside a k-step lo*hi, hi*lo, hi*hi (PLANES = 2) — the order of
// gemm_h2_core.h's tile engine, so the split class reproduces that engine's sums bit for bit.
//
// Epilogue: accumulators -> LDS (f32, per wave) -> row-contiguous 8 columns per lane: bias / activation / BN affine / residual on 16- and 32-byte
// accesses, output as f32 rows, fp16 rows or the P32 image of the next layer's operand.
#pragma once
#include <type_traits>

#include "xp_common.h"

typedef _Float16 rg_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 rg_h2 __attribute__((ext_vector_type(2)));
typedef float rg_acc __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* rg_lds_ptr;

#ifndef XP_RING_DBG
#define XP_RING_DBG 0   /* timing experiments only (wrong results): 1 no MFMA, 2 no DMA after the prologue, 4 no fragment reads after slab 0, 8 no output stores */
#endif

enum { RG_F32 = 0, RG_F16 = 1, RG_P32 = 2 };

struct RingParams {
    const char* A; const char* W;        // operand images
    int64_t a_row, a_slab;               // byte strides: row -> row, slab -> slab
    int64_t w_row, w_slab;
    int M, N, T;                         // rows, columns, slabs
    void* C; int ldc; int out_fmt;       // ldc in ELEMENTS (RG_P32: elements = padded K of the consumer, a multiple of 32)
    const float* wscale;                 // (N) or null: per-column factor on the accumulator (the split weights' power-of-two row scale, exact)
    const float* bias; const float* scale; const float* shift;      // (N) or null
    const void* res; int ldres; int res_fmt;                        // residual (M, ldres): RG_F32 or RG_F16
    int act;                             // 0 none, 1 GELU(erf), 2 ReLU before the affine, 3 ReLU after it
    int r16;                             // round to fp16 after every operation autocast would end in a half tensor (the fp16 class; also "amp16")
    int ngroup;                          // column tiles per group of the tile order (0: all column tiles of a row tile adjacent)
};

__device__ __forceinline__ float rg_r16(float v) { return (float)(_Float16)v; }

// GM x GN waves per group, TM x TN MFMA tiles (32 x 32) per wave, S ring stages
template <int GM, int GN, int TM, int TN, int PLANES, int S>
struct RingTile {
    static_assert(GM * GN == 4, "four waves per group");
    static constexpr int BM = 2 * GM * TM * 32, BN = GN * TN * 32;
    static constexpr int ROWB = 128;
    static constexpr int STAGE = (BM + BN) * ROWB;
    static constexpr int PIECES = STAGE / 1024;
    static constexpr int PW = PIECES / 8;                          // DMA pieces per wave and slab
    static_assert(PIECES % 8 == 0, "whole pieces per wave");
    static constexpr int WCOLS = TN * 32;
    static constexpr int ESTRIDE = WCOLS + 32;                     // f32 per staged epilogue row: == 32 (mod 64) so that a ds_read_b128 group's four rows use four bank quarters
    static constexpr int EPI_WAVE = 32 * ESTRIDE * 4;              // bytes per wave: one 32-row block at a time
    static constexpr size_t kLdsBytes = (size_t)S * STAGE > (size_t)8 * EPI_WAVE ? (size_t)S * STAGE : (size_t)8 * EPI_WAVE;
    static_assert(kLdsBytes <= 160 * 1024, "LDS");
    static_assert(PW * (S - 1) < 64, "vmcnt field");
};

template <int N> __device__ __forceinline__ void rg_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void rg_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

template <int GM, int GN, int TM, int TN, int PLANES, int S>
__global__ __launch_bounds__(512) void ring_gemm_kernel(RingParams p) {
    using T = RingTile<GM, GN, TM, TN, PLANES, S>;
    extern __shared__ __align__(16) unsigned char rg_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2, w4 = wave & 3, gm = w4 / GN, gn = w4 % GN;

    // ---- tile order: workgroup ids go round-robin to the 8 XCDs; every XCD gets one contiguous run of logical tiles (cdna_hip_programming.md T1, bijective
    //      form); inside a run the column tiles of one row tile (or of one group of p.ngroup column tiles) are adjacent: they share A rows in that XCD's L2
    const int ntn = (p.N + T::BN - 1) / T::BN, ntm = (p.M + T::BM - 1) / T::BM, nt = ntn * ntm;
    int mt, ntile;
    {
        const int b = blockIdx.x, xcd = b & 7, q = nt >> 3, r = nt & 7;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
        if (p.ngroup > 0 && p.ngroup < ntn) {
            const int per = ntm * p.ngroup;
            const int g = logical / per, rem = logical - g * per;
            const int gw = min(p.ngroup, ntn - g * p.ngroup);
            mt = rem / gw; ntile = g * p.ngroup + (rem - mt * gw);
        } else {
            mt = logical / ntn; ntile = logical - mt * ntn;
        }
    }
    const int m0 = mt * T::BM, n0 = ntile * T::BN;

    // ---- DMA plan: piece = 1 KB = 8 stage rows; lane l writes stage row 8 piece + (l >> 3), physical slot l & 7, and so FETCHES logical slot (l & 7) ^ swz(row).
    //      Rows past M / N are clamped to the last real row (valid memory; they only reach outputs that are never stored).  A source address is a UNIFORM
    //      64-bit base (operand + slab offset: scalar registers) plus a per-lane 32-bit byte offset fixed for the whole kernel (host: operand images < 4 GB):
    //      the global_load_lds takes its saddr + voffset form and a piece costs no vector arithmetic at all.
    unsigned voff[T::PW];
#pragma unroll
    for (int i = 0; i < T::PW; ++i) {
        const int piece = wave + i * 8;
        const int row = piece * 8 + (lane >> 3);
        const int ls = (lane & 7) ^ ((row >> 1) & 7);
        if (piece * 8 < T::BM) voff[i] = (unsigned)min(m0 + row, p.M - 1) * (unsigned)p.a_row + ls * 16;
        else voff[i] = (unsigned)min(n0 + row - T::BM, p.N - 1) * (unsigned)p.w_row + ls * 16;
    }
    auto issue_piece = [&](int t, int i) {          // piece i of this wave's share of slab t -> ring buffer t % S
        if ((XP_RING_DBG & 2) && t >= S) return;
        const int piece = wave + i * 8;
        const char* base = piece * 8 < T::BM ? p.A + (int64_t)t * p.a_slab : p.W + (int64_t)t * p.w_slab;      // uniform
        asm volatile("" : "+s"(base));          // opaque scalar: keeps hipcc from folding the slab offset into a per-lane 64-bit pointer (v_mad_u64_u32 per piece)
        // (the local copy matters: with a captured array element passed straight to the builtin, hipcc 7.2's host pass silently drops the kernel stub)
        const char* s = base + voff[i];
        __builtin_amdgcn_global_load_lds(s, (rg_lds_ptr)(rg_lds + (t % S) * T::STAGE + piece * 1024), 16, 0, 0);
    };
    auto issue = [&](int t) {
#pragma unroll
        for (int i = 0; i < T::PW; ++i) issue_piece(t, i);
    };

    // ---- fragments: lane (fr, fh) reads the 8 halves of unit u (16 k) of row base + fr: logical slot 2u + fh, physical slot ^ swz(fr) (row bases are multiples of 32)
    const int fr = lane & 31, fh = lane >> 5;
    const int cx = (fh ^ ((fr >> 1) & 7)) << 4;
    const int a_frag = ((grp * GM + gm) * TM * 32 + fr) * T::ROWB;
    const int b_frag = (T::BM + gn * TN * 32 + fr) * T::ROWB;
    rg_h8 af[4][TM], bf[4][TN];
    auto frags = [&](int t) {
        if ((XP_RING_DBG & 4) && t > 0) return;
        const unsigned char* bb = rg_lds + (t % S) * T::STAGE;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[u][i] = *reinterpret_cast<const rg_h8*>(bb + a_frag + i * 32 * T::ROWB + (cx ^ (u << 5)));
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[u][j] = *reinterpret_cast<const rg_h8*>(bb + b_frag + j * 32 * T::ROWB + (cx ^ (u << 5)));
        }
    };
    rg_acc acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // the MFMAs of one slab; fill(n) runs after the n-th of them (n = 0, 1, ...): DMA pieces issued one per matrix-instruction gap ride in the pipe's shadow
    auto mfmas = [&](auto&& fill) {
        if (XP_RING_DBG & 1) {
            acc[0][0][0] += (float)af[0][0][0] * (float)bf[3][TN - 1][1] + (float)af[3][TM - 1][2] * (float)bf[0][0][3];
#pragma unroll
            for (int n = 0; n < T::PW; ++n) fill(n);
            return;
        }
        int n = 0;
        if (PLANES == 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) { acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[u][i], bf[u][j], acc[i][j], 0, 0, 0); fill(n++); }
        } else {
            constexpr int PA[3] = {2, 0, 0}, PB[3] = {0, 2, 0};          // unit offsets of the planes: lo * hi, hi * lo, hi * hi (smallest partial products first)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int pp = 0; pp < 3; ++pp)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[PA[pp] + ks][i], bf[PB[pp] + ks][j], acc[i][j], 0, 0, 0);
                            fill(n++);
                        }
        }
    };
    static_assert(T::PW <= (PLANES == 1 ? 4 : 6) * TM * TN, "one DMA piece per matrix-instruction gap");

    // ---- K loop ----
    const int nslab = p.T;
#pragma unroll
    for (int t = 0; t < S; ++t)
        if (t < nslab) issue(t);
    if (nslab >= S) rg_wait_vm<(S - 1) * T::PW>(); else rg_wait_vm<0>();        // this wave's pieces of slab 0 have landed
    rg_barrier();                                                               // B0: everybody's
    if (grp == 1) rg_barrier();                                                 // group 1 runs one phase behind
    for (int t = 0; t < nslab; ++t) {
        frags(t);
        if (grp == 1) {                                                         // slab t + 1 must be complete before group 0 reads it after the next barrier
            if (t + S - 1 < nslab) rg_wait_vm<(S - 2) * T::PW>(); else rg_wait_vm<0>();
        }
        rg_barrier();
        // DMA of this phase, one piece per matrix-instruction gap: group 0 refills the buffer of slab t - 1 (group 1 finished reading it two barriers ago)
        // with slab t + S - 1, group 1 the buffer of slab t (it was its last reader, a barrier ago) with slab t + S
        const int ti = t + S - 1 + grp;
        const bool do_issue = ti < nslab && (grp == 1 || t > 0);
        __builtin_amdgcn_s_setprio(1);
        mfmas([&](int n) {
            if (n < T::PW) {
                if (do_issue) issue_piece(ti, n);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        __builtin_amdgcn_s_setprio(0);
        if (grp == 0) {
            if (t + S - 1 < nslab) rg_wait_vm<(S - 2) * T::PW>(); else rg_wait_vm<0>();
            rg_barrier();
        } else if (t + 1 < nslab) rg_barrier();
    }

    // ---- epilogue: one 32-row block of the wave's tile at a time through the wave's own LDS region (the ring is free: the last fragment reads and the last
    //      DMA completed before the last barrier this wave passed) ----
    float* el = reinterpret_cast<float*>(rg_lds + wave * T::EPI_WAVE);
    constexpr int LPR = T::WCOLS / 8;                 // lanes per staged row (8 columns each)
    constexpr int RPP = 64 / LPR;                     // rows per pass
    const int er = lane / LPR, ec = (lane % LPR) * 8; // this lane's row inside a pass, its first column inside the wave's tile
    const int ncol0 = n0 + gn * T::WCOLS + ec;
    const bool cok = ncol0 < p.N;                     // N % 8 == 0 (host): a lane's 8 columns are all real or all past N
    const int ncc = cok ? ncol0 : 0;
    float wsv[8], biv[8], scv[8], shv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        wsv[e] = p.wscale ? p.wscale[ncc + e] : 1.f;
        biv[e] = p.bias ? p.bias[ncc + e] : 0.f;
        scv[e] = p.scale ? p.scale[ncc + e] : 1.f;
        shv[e] = p.shift ? p.shift[ncc + e] : 0.f;
    }
    auto run = [&](auto act_tag, auto r16_tag) {
        constexpr int ACT = decltype(act_tag)::value;
        constexpr bool R16 = decltype(r16_tag)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            // accumulators -> LDS: lane (fr, fh), register r of tile j is row (r & 3) + 8 (r >> 2) + 4 fh, column 32 j + fr
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) el[((r & 3) + 8 * (r >> 2) + 4 * fh) * T::ESTRIDE + j * 32 + fr] = acc[i][j][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the wave reads back only what it wrote itself
            const int mrow0 = m0 + ((grp * GM + gm) * TM + i) * 32;
#pragma unroll
            for (int ps = 0; ps < 32 / RPP; ++ps) {
                const int rl = ps * RPP + er, m = mrow0 + rl;
                const float4 v0 = *reinterpret_cast<const float4*>(el + rl * T::ESTRIDE + ec);
                const float4 v1 = *reinterpret_cast<const float4*>(el + rl * T::ESTRIDE + ec + 4);
                float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                const bool ok = cok && m < p.M;
                float rv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) rv[e] = 0.f;
                if (p.res && ok) {
                    if (p.res_fmt == RG_F32) {
                        const float* rp = reinterpret_cast<const float*>(p.res) + (int64_t)m * p.ldres + ncol0;
                        const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
                        rv[0] = r0.x; rv[1] = r0.y; rv[2] = r0.z; rv[3] = r0.w; rv[4] = r1.x; rv[5] = r1.y; rv[6] = r1.z; rv[7] = r1.w;
                    } else {
                        const rg_h8 rh = *reinterpret_cast<const rg_h8*>(reinterpret_cast<const _Float16*>(p.res) + (int64_t)m * p.ldres + ncol0);
#pragma unroll
                        for (int e = 0; e < 8; ++e) rv[e] = (float)rh[e];
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float x = p.wscale ? v[e] * wsv[e] + biv[e] : v[e] + biv[e];          // wscale is a power of two: the product is exact
                    if (R16) x = rg_r16(x);
                    if (ACT == 1) { x = xp_gelu_fast(x); if (R16) x = rg_r16(x); }
                    if (ACT == 2) x = fmaxf(x, 0.f);
                    if (!R16 || p.scale) x = x * scv[e] + shv[e];
                    if (R16) { if (p.scale) x = rg_r16(x); }
                    if (ACT == 3) x = fmaxf(x, 0.f);
                    if (!R16 || p.res) x = rv[e] + x;                                     // (the f32 classes add the 0 of an absent residual, like gemm_epilogue.h: -0 -> +0)
                    if (R16) { if (p.res) x = rg_r16(x); }
                    v[e] = x;
                }
                if (!ok || (XP_RING_DBG & 8)) continue;
                if (p.out_fmt == RG_F32) {
                    float* cp = reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + ncol0;
                    *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>(cp + 4) = make_float4(v[4], v[5], v[6], v[7]);
                } else if (p.out_fmt == RG_F16) {
                    rg_h8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (_Float16)v[e];
                    *reinterpret_cast<rg_h8*>(reinterpret_cast<_Float16*>(p.C) + (int64_t)m * p.ldc + ncol0) = o;
                } else {
                    // P32 image of the consumer: row m = ldc / 32 slabs of [hi: 32 halves][lo: 32 halves]; columns ncol0 .. + 7 lie inside one slab
                    rg_h8 hi, lo;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { hi[e] = (_Float16)v[e]; lo[e] = (_Float16)(v[e] - (float)hi[e]); }
                    unsigned char* cp = reinterpret_cast<unsigned char*>(p.C) + (int64_t)m * p.ldc * 4 + (ncol0 >> 5) * 128 + (ncol0 & 31) * 2;
                    *reinterpret_cast<rg_h8*>(cp) = hi;
                    *reinterpret_cast<rg_h8*>(cp + 64) = lo;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the read-back is done before the next block overwrites the region
        }
    };
    auto by_act = [&](auto r16_tag) {
        switch (p.act) {
            case 1: run(std::integral_constant<int, 1>{}, r16_tag); break;
            case 2: run(std::integral_constant<int, 2>{}, r16_tag); break;
            case 3: run(std::integral_constant<int, 3>{}, r16_tag); break;
            default: run(std::integral_constant<int, 0>{}, r16_tag); break;
        }
    };
    if (p.r16) by_act(std::true_type{}); else by_act(std::false_type{});
}
