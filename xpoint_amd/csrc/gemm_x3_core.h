// Split-bf16 tile engine: an fp32-accurate GEMM on the bf16 matrix pipe.
//
// Every fp32 operand value is the EXACT sum of three bf16 values (8 significand bits each, 3 x 8 = 24):
//     x = x0 + x1 + x2,   x0 = bf16(x),  x1 = bf16(x - x0),  x2 = x - x0 - x1   (the two differences are exact in f32)
// and a product a*b is evaluated as the six partial products with weight >= 2^-16,
//     a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0),
// each on v_mfma_f32_32x32x16_bf16 (bf16 x bf16 products are exact in f32; f32 accumulate).  The dropped terms
// (a1 b2, a2 b1, a2 b2) are <= 2^-24 |a b| each — below the rounding of a plain f32 FMA chain (measured: relative
// error 7e-9 of sum|a||b| for this truncation vs 3e-7 for an f32 chain at K = 96).  Six bf16 MFMAs
// (16 k each, 8 passes) replace eight f32 MFMAs (2 k each, 16 passes): 2.67x the matrix-pipe rate.
//
// A workgroup of WM x WN waves owns a (WM*TM*32) x (WN*TN*32) tile; K is walked in 16-wide slabs = one MFMA k-step.
// The A operand (activations, f32 in HBM) is split in registers while it is staged (4 lanes x 16 B cover the 64 contiguous
// bytes a tile row contributes to a slab); the B operand (weights) is split once, offline (xp_split_weights_x3), into a
// slab-major plane layout so that a tile's slab is one contiguous run and staging B is a straight 16-byte copy.  LDS holds two
// slabs (double buffer, (BM + BN) rows x 112 B each: 56 KB for 128 x 128, two workgroups of 8 waves per CU), one LDS-only
// barrier per slab; global loads run two slabs ahead in registers.  An LDS row is [plane][16 bf16] + 16 B pad = 112 B:
// with ds_read_b128's lane groups the 16 lanes of a group land on 16 distinct 4-bank slots (28*row mod 64), conflict-free.
// MFMA operand layout (32x32x16): lane l holds 8 consecutive k (= 8*(l>>5) .. +7) of row/col l & 31.
#pragma once
#include "gemm_epilogue.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#ifndef XP_X3_DBG
#define XP_X3_DBG 0   /* timing experiments only (wrong results): 1 no split VALU, 2 no global loads after the prologue, 4 no MFMA, 8 no LDS store, 16 no barrier, 32 no fragment reads */
#endif
constexpr int X3_BK = 16;          // k per slab
constexpr int X3_ROWB = 112;       // LDS bytes per tile row: 3 planes x 16 bf16 (= the offline weight layout) + 16 B pad
constexpr int X3_SLAB_UNITS = 6;   // 16-byte units per (weight row, slab) in the offline layout: 3 planes x 2 octets

// Workgroup barrier that orders LDS traffic only: waits for this wave's LDS operations (lgkmcnt) and NOT for its
// outstanding global loads (vmcnt) — __syncthreads() drains both, which would cut the register prefetch to zero slabs.
__device__ __forceinline__ void xp_lds_barrier() {
    if (XP_X3_DBG & 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// exact three-way bf16 split of two floats; returns the packed pairs (low half = x, high half = y) per plane
// round-to-nearest-even conversion of two floats to a packed bf16 pair (low half = x): ONE v_cvt_pk_bf16_f32 (written as asm because
// hipcc otherwise converts the low element a second time to extract it: 11 instead of 9 instructions per split pair)
__device__ __forceinline__ unsigned xp_cvt_pk_bf16(float x, float y) {
#if defined(XP_X3_NO_ASM_CVT)
    union { bf16x2 v; unsigned u; } c;
    c.v[0] = (__bf16)x; c.v[1] = (__bf16)y;
    return c.u;
#else
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
#endif
}
__device__ __forceinline__ void xp_split2(float x, float y, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = xp_cvt_pk_bf16(x, y);
    const float rx = x - __uint_as_float(p0 << 16), ry = y - __uint_as_float(p0 & 0xffff0000u);
    p1 = xp_cvt_pk_bf16(rx, ry);
    const float sx = rx - __uint_as_float(p1 << 16), sy = ry - __uint_as_float(p1 & 0xffff0000u);
    p2 = xp_cvt_pk_bf16(sx, sy);
}

__device__ __forceinline__ void xp_split4(const float4& v, uint2& p0, uint2& p1, uint2& p2) {
    xp_split2(v.x, v.y, p0.x, p1.x, p2.x);
    xp_split2(v.z, v.w, p0.y, p1.y, p2.y);
}

__device__ __forceinline__ void xp_split8(const float4& lo, const float4& hi, uint4& p0, uint4& p1, uint4& p2) {
    xp_split2(lo.x, lo.y, p0.x, p1.x, p2.x);
    xp_split2(lo.z, lo.w, p0.y, p1.y, p2.y);
    xp_split2(hi.x, hi.y, p0.z, p1.z, p2.z);
    xp_split2(hi.z, hi.w, p0.w, p1.w, p2.w);
}

// NP = partial products per multiply: 6 (default; every product of weight >= 2^-16: f32-grade), 3 (a0 b0 + a0 b1 + a1 b0: error
// <= 3 * 2^-16 * sum|a||b|, half the matrix work) or 1 (a0 b0: plain bf16 operands, f32 accumulate — the "autocast" class).
// Planes that no product reads are neither split, stored nor loaded.
template <int WM, int WN, int TM, int TN, int NP = 6>
struct GemmTileX3 {
    static_assert(NP == 6 || NP == 3 || NP == 1, "NP");
    static constexpr int NPL = NP == 6 ? 3 : (NP == 3 ? 2 : 1);     // operand planes in use
    static constexpr int B_UNITS = 2 * NPL;                          // 16-byte units of a weight row per slab that are staged
    static constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
    static constexpr int A_TOT = BM * 4;                    // (row, k-quad) staging slots per slab: 16 B of f32 each
    static constexpr int B_TOT = BN * B_UNITS;              // 16-byte units per slab
    static constexpr int A_LD = (A_TOT + NT - 1) / NT, B_LD = (B_TOT + NT - 1) / NT;
    static constexpr int kBufBytes = (BM + BN) * X3_ROWB;
    static constexpr size_t kLdsBytes = 2 * (size_t)kBufBytes;
    // Staging work is dealt round-robin; when a round is only partly needed the surplus threads repeat the last slot
    // (same data to the same LDS address), which keeps the K loop free of divergent regions — those made hipcc drain
    // every outstanding global load at each barrier.
    __device__ static __forceinline__ int a_id(int s) { const int id = (int)threadIdx.x + s * NT; return (s + 1) * NT <= A_TOT ? id : (id < A_TOT ? id : A_TOT - 1); }
    __device__ static __forceinline__ int b_id(int s) { const int id = (int)threadIdx.x + s * NT; return (s + 1) * NT <= B_TOT ? id : (id < B_TOT ? id : B_TOT - 1); }
    __device__ static __forceinline__ int a_row(int s) { return a_id(s) >> 2; }
    __device__ static __forceinline__ int a_quad(int s) { return a_id(s) & 3; }
    __device__ static __forceinline__ int b_row(int s) { return b_id(s) / B_UNITS; }
    __device__ static __forceinline__ int b_unit(int s) { return b_id(s) % B_UNITS; }   // plane * 2 + octet

    struct RawA { float4 a[A_LD]; bool ok[A_LD]; };      // f32 A values of one slab as loaded
    struct SplitA { uint2 p[A_LD][3]; };                 // the same slab split into its three bf16 planes
    struct RawB { uint4 b[B_LD]; };                      // offline-split weights of one slab as loaded

    // Tile rows are permuted inside aligned groups of 8 when stored to LDS: with the 112-B row stride that puts the 4 rows
    // of a ds_write_b64 lane group (4 rows x 4 quads) on four different bank octets AND keeps the ds_read_b128 fragment
    // reads conflict-free (both verified by enumeration; SQ_LDS_BANK_CONFLICT was 36 % of the LDS cycles without it).
    __device__ static __forceinline__ int lds_row(int r) { return (r & ~7) | ((0x35712460u >> ((r & 7) * 4)) & 7); }

    // ldA(slot, k, v) -> ok: the 4 consecutive f32 starting at absolute k of the slot's row, loaded unconditionally from a
    //                        valid address; ok = whether they are real (else the slot is stored as zeros).
    //                        Called once per slot and slab, in slab order.
    // ldB(slot, slab)    -> the slot's 16-byte unit of the offline-split weights (rows past N are clamped: they only feed
    //                        output columns that are never stored; k past K is zero in the offline layout)
    //
    // Software pipeline, per slab t (one LDS-only barrier each; everything below is straight-line code, slabs past K
    // contribute exact zeros):
    //     ds_read   fragments of slab t                      (its buffer was completed during slab t-1)
    //     MFMA      first third of slab t
    //     ds_write  slab t+1: A planes split during slab t-1, B as loaded during slab t-2  -> the other buffer
    //     global    loads of B slab t+3 and A slab t+5 into registers that were just consumed
    //     MFMA      rest of slab t, with the split arithmetic of A slab t+2 (loaded during slab t-3) in its shadow
    // so no LDS store and no global load is waited for right before a barrier.
    template <class LA, class LB>
    __device__ static __forceinline__ void run(unsigned char* lds, int K, LA ldA, LB ldB, f32x16 (&acc)[TM][TN]) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int wm = wave / WN, wn = wave % WN;
        const int fr = lane & 31, fh = lane >> 5;
        constexpr bool LOADS = !(XP_X3_DBG & 2), STORES = !(XP_X3_DBG & 8);
        auto gloadA = [&](RawA& r, int t) {
#pragma unroll
            for (int s = 0; s < A_LD; ++s) r.ok[s] = ldA(s, t * X3_BK + a_quad(s) * 4, r.a[s]);
        };
        auto gloadB = [&](RawB& r, int t) {
#pragma unroll
            for (int s = 0; s < B_LD; ++s) r.b[s] = ldB(s, t);
        };
        auto split = [&](const RawA& r, SplitA& o) {
#pragma unroll
            for (int s = 0; s < A_LD; ++s) {
                // slots that are not real (k past K, conv zero padding) become zeros by masking the INPUT bits: selects,
                // not a branch
                const unsigned m = r.ok[s] ? 0xffffffffu : 0u;
                auto mk = [&](float v) { return __uint_as_float(__float_as_uint(v) & m); };
                const float4 v = make_float4(mk(r.a[s].x), mk(r.a[s].y), mk(r.a[s].z), mk(r.a[s].w));
                if (XP_X3_DBG & 1) {
                    o.p[s][0] = make_uint2(__float_as_uint(v.x), __float_as_uint(v.y)); o.p[s][1] = make_uint2(__float_as_uint(v.z), __float_as_uint(v.w));
                    o.p[s][2] = o.p[s][0];
                } else {
                    xp_split4(v, o.p[s][0], o.p[s][1], o.p[s][2]);
                }
            }
        };
        int a_dst[A_LD], b_dst[B_LD];       // LDS byte offsets of this thread's staging slots inside a buffer
#pragma unroll
        for (int s = 0; s < A_LD; ++s) a_dst[s] = lds_row(a_row(s)) * X3_ROWB + a_quad(s) * 8;
#pragma unroll
        for (int s = 0; s < B_LD; ++s) b_dst[s] = BM * X3_ROWB + b_row(s) * X3_ROWB + b_unit(s) * 16;   // a straight copy of the offline layout
        auto lstore = [&](const SplitA& sa, const RawB& rb, unsigned char* buf) {
            if (!STORES) return;
#pragma unroll
            for (int s = 0; s < A_LD; ++s) {
                unsigned char* d = buf + a_dst[s];
                *reinterpret_cast<uint2*>(d) = sa.p[s][0];
                if (NPL > 1) *reinterpret_cast<uint2*>(d + 32) = sa.p[s][1];
                if (NPL > 2) *reinterpret_cast<uint2*>(d + 64) = sa.p[s][2];
            }
#pragma unroll
            for (int s = 0; s < B_LD; ++s) *reinterpret_cast<uint4*>(buf + b_dst[s]) = rb.b[s];
        };
        const int a_frag = lds_row(wm * TM * 32 + fr) * X3_ROWB + 16 * fh;      // lds_row permutes inside groups of 8: + i * 32 rows commutes
        const int b_frag = BM * X3_ROWB + (wn * TN * 32 + fr) * X3_ROWB + 16 * fh;
        bf16x8 af[3][TM], bf[3][TN];
        auto frags = [&](const unsigned char* buf) {
            if ((XP_X3_DBG & 32) && buf != lds) return;
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[pl][i] = *reinterpret_cast<const bf16x8*>(buf + a_frag + pl * 32 + i * 32 * X3_ROWB);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[pl][j] = *reinterpret_cast<const bf16x8*>(buf + b_frag + pl * 32 + j * 32 * X3_ROWB);
            }
        };
        auto mfmas = [&](int pp0, int pp1) {
            // smallest partial products first; the TM*TN independent accumulators separate dependent MFMAs
            constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};      // NP = 3: the last three, NP = 1: the last
#pragma unroll
            for (int pp = pp0 + (6 - NP); pp < pp1 + (6 - NP); ++pp)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if (XP_X3_DBG & 4) { if (pp == 6 - NP) acc[i][j][0] += (float)af[0][i][0] * (float)bf[0][j][0] + (float)af[1][i][1] * (float)bf[1][j][1] + (float)af[2][i][2] * (float)bf[2][j][2]; }
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[pp]][i], bf[PB[pp]][j], acc[i][j], 0, 0, 0);
                    }
        };
        // Issue order inside a slab (one scheduling region between two barriers); a wave issues in order, so whatever is
        // left after the last MFMA runs with the matrix pipe idle, and global loads issued back to back by every wave
        // at once queue up in the texture-address unit with the MFMAs stuck behind them.
        auto pipeline = [&]() {      // for the second part of a slab: 4 partial products with the loads and the split VALU between them
#if !defined(XP_X3_NO_SGB)
            constexpr int NMFMA = (NP - NP / 3) * TM * TN, NLOAD = A_LD + B_LD, NVALU = 30 * A_LD + 8 + 4 * NLOAD;
            constexpr int VPER = (NVALU + NMFMA - 1) / NMFMA;
            constexpr int LEVERY = NMFMA / (2 * NLOAD) > 0 ? NMFMA / (2 * NLOAD) : 1;       // loads in the first half of this MFMA stream
#pragma unroll
            for (int i = 0; i < NMFMA; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    // one MFMA
                if (i % LEVERY == LEVERY - 1 && i / LEVERY < NLOAD) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // one VMEM read
                __builtin_amdgcn_sched_group_barrier(0x002, VPER, 0);                 // a few VALU
            }
#endif
        };
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        unsigned char* bufs[2] = {lds, lds + kBufBytes};
        const int nslab = ((K + X3_BK - 1) / X3_BK + 1) & ~1;       // even; a slab past K adds exact zeros
        RawA ra[4]; RawB rb[2]; SplitA sp;
        // slab t: A f32 values live in ra[t % 4] from their load (issued during slab t-5) until they are split (during
        // slab t-2); B planes live in rb[t % 2] from their load (slab t-3) until they are stored (slab t-1).
        gloadA(ra[0], 0); gloadB(rb[0], 0);
        gloadA(ra[1], 1); gloadB(rb[1], 1);
        gloadA(ra[2], 2); gloadA(ra[3], 3);
        split(ra[0], sp);
        lstore(sp, rb[0], bufs[0]);
        gloadB(rb[0], 2); gloadA(ra[0], 4);
        split(ra[1], sp);
        xp_lds_barrier();
        auto step = [&](int t, auto u_tag) {
            constexpr int U = decltype(u_tag)::value;          // t % 4, compile-time so that every register index is static
            frags(bufs[U & 1]);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(0, NP / 3);                                   // the LDS is busy with every wave's fragment reads right now:
            __builtin_amdgcn_sched_barrier(0);                  // the stores of slab t+1 go out once a third of the MFMAs are queued
            lstore(sp, rb[(U + 1) & 1], bufs[(U + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (LOADS) { gloadB(rb[(U + 1) & 1], t + 3); gloadA(ra[(U + 1) & 3], t + 5); }
            mfmas(NP / 3, NP);
            split(ra[(U + 2) & 3], sp);
            pipeline();
            xp_lds_barrier();
        };
        int t = 0;
        for (; t + 4 <= nslab; t += 4) {
            step(t, std::integral_constant<int, 0>{}); step(t + 1, std::integral_constant<int, 1>{});
            step(t + 2, std::integral_constant<int, 2>{}); step(t + 3, std::integral_constant<int, 3>{});
        }
        if (t < nslab) { step(t, std::integral_constant<int, 0>{}); step(t + 1, std::integral_constant<int, 1>{}); }
    }

    // Both operands already split (xp_split_weights_x3 layout): staging is a straight copy for A as well — no VALU at all.
    // ldAu(slot, slab) / ldBu(slot, slab) -> the slot's 16-byte unit.  Used by the descriptor matcher (match.hip), whose
    // two operands are split once per image by its prepare kernel.
    static constexpr int AU_TOT = BM * X3_SLAB_UNITS, AU_LD = (AU_TOT + NT - 1) / NT;
    __device__ static __forceinline__ int au_id(int s) { const int id = (int)threadIdx.x + s * NT; return (s + 1) * NT <= AU_TOT ? id : (id < AU_TOT ? id : AU_TOT - 1); }
    __device__ static __forceinline__ int au_row(int s) { return au_id(s) / X3_SLAB_UNITS; }
    __device__ static __forceinline__ int au_unit(int s) { return au_id(s) % X3_SLAB_UNITS; }
    // NPROD = 6: all partial products of weight >= 2^-16 (f32-grade result); NPROD = 3: a0 b0 + a0 b1 + a1 b0 only — the dropped
    // terms are bounded by 3 * 2^-16 * sum_k |a_k||b_k| (|x1| <= 2^-8 |x|, |x2| <= 2^-16 |x|), half the matrix work: for consumers
    // that only need a bounded-error result (the matcher's nomination pass).
    template <int NPROD = 6, class LA, class LB>
    __device__ static __forceinline__ void run_presplit(unsigned char* lds, int K, LA ldAu, LB ldBu, f32x16 (&acc)[TM][TN]) {
        static_assert(NPROD == 6 || NPROD == 3, "NPROD");
        static_assert(NP == 6, "run_presplit stages all six units of both operands: use the default tile");
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int wm = wave / WN, wn = wave % WN;
        const int fr = lane & 31, fh = lane >> 5;
        struct RawAu { uint4 a[AU_LD]; };
        int a_dst[AU_LD], b_dst[B_LD];
#pragma unroll
        for (int s = 0; s < AU_LD; ++s) a_dst[s] = lds_row(au_row(s)) * X3_ROWB + au_unit(s) * 16;
#pragma unroll
        for (int s = 0; s < B_LD; ++s) b_dst[s] = BM * X3_ROWB + b_row(s) * X3_ROWB + b_unit(s) * 16;
        const int a_frag = lds_row(wm * TM * 32 + fr) * X3_ROWB + 16 * fh;
        const int b_frag = BM * X3_ROWB + (wn * TN * 32 + fr) * X3_ROWB + 16 * fh;
        bf16x8 af[3][TM], bf[3][TN];
        auto gload = [&](RawAu& ra, RawB& rb, int t) {
#pragma unroll
            for (int s = 0; s < AU_LD; ++s) ra.a[s] = ldAu(s, t);
#pragma unroll
            for (int s = 0; s < B_LD; ++s) rb.b[s] = ldBu(s, t);
        };
        auto lstore = [&](const RawAu& ra, const RawB& rb, unsigned char* buf) {
#pragma unroll
            for (int s = 0; s < AU_LD; ++s) *reinterpret_cast<uint4*>(buf + a_dst[s]) = ra.a[s];
#pragma unroll
            for (int s = 0; s < B_LD; ++s) *reinterpret_cast<uint4*>(buf + b_dst[s]) = rb.b[s];
        };
        auto frags = [&](const unsigned char* buf) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[pl][i] = *reinterpret_cast<const bf16x8*>(buf + a_frag + pl * 32 + i * 32 * X3_ROWB);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[pl][j] = *reinterpret_cast<const bf16x8*>(buf + b_frag + pl * 32 + j * 32 * X3_ROWB);
            }
        };
        auto mfmas = [&](int pp0, int pp1) {
            constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};      // NPROD = 3: the last three
#pragma unroll
            for (int pp = pp0 + (6 - NPROD); pp < pp1 + (6 - NPROD); ++pp)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[pp]][i], bf[PB[pp]][j], acc[i][j], 0, 0, 0);
        };
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        unsigned char* bufs[2] = {lds, lds + kBufBytes};
        const int nslab = ((K + X3_BK - 1) / X3_BK + 1) & ~1;       // callers pad K to a multiple of 32 (zero planes)
        RawAu ra[2]; RawB rb[2];
        gload(ra[0], rb[0], 0);
        gload(ra[1], rb[1], 1);
        lstore(ra[0], rb[0], bufs[0]);
        gload(ra[0], rb[0], 2);
        xp_lds_barrier();
        auto step = [&](int t, auto u_tag) {
            constexpr int U = decltype(u_tag)::value;          // t % 2
            frags(bufs[U]);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(0, NPROD / 3);
            __builtin_amdgcn_sched_barrier(0);
            lstore(ra[U ^ 1], rb[U ^ 1], bufs[U ^ 1]);       // slab t+1, loaded during slab t-2
            __builtin_amdgcn_sched_barrier(0);
            gload(ra[U ^ 1], rb[U ^ 1], t + 3);
            mfmas(NPROD / 3, NPROD);
            xp_lds_barrier();
        };
        for (int t = 0; t < nslab; t += 2) { step(t, std::integral_constant<int, 0>{}); step(t + 1, std::integral_constant<int, 1>{}); }
    }

    // element (i, j, r) of this lane's accumulators is C[row_of(i, r)][col_of(j)] within the tile
    __device__ static __forceinline__ int row_of(int i, int r) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        return ((wave / WN) * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    }
    __device__ static __forceinline__ int col_of(int j) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        return ((wave % WN) * TN + j) * 32 + (lane & 31);
    }
};
