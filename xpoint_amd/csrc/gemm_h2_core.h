// Split-fp16 tile engine ("h2"): an fp32-grade GEMM on the f16 matrix pipe with THREE partial products per multiply.
//
// An f32 value x (24 significant bits) is written as x = h0 + h1 + r.  fp16 has an 11-bit significand (unit roundoff 2^-11):
//     h0 = fp16(x)  (round to nearest: |x - h0| <= 2^-11 |x|),   h1 = fp16(x - h0)  (the difference is exact in f32 and has at most 12
//     significant bits, of which fp16 keeps 11:  |r| = |x - h0 - h1| <= 2^-23 |x| worst case, 2^-25.5 |x| on average, measured)
// i.e. the pair reproduces x to within ONE f32 ulp in the worst case, and a product a*b is evaluated as
//     a0 b0 + a0 b1 + a1 b0          (dropped: a1 b1 <= 2^-22 |a b| worst case: both operands at an fp16 rounding midpoint)
// on v_mfma_f32_32x32x16_f16 (fp16 x fp16 products are exact in f32; f32 accumulate).  Per product: <= 2^-23 + 2^-23 + 2^-22 = 2^-21 |ab| in the
// worst case (operands at rounding midpoints with aligned signs: tests/test_gpu_h2.py::test_h2_adversarial_midpoints), ~2^-25 typically — the size of
// the f32 rounding a plain f32 FMA chain, the reference's arithmetic, commits per product (2^-24).  Half the matrix work of the six-product bf16
// split (gemm_x3_core.h), which is exact to 2^-24 per operand at the price of six products.
//
// fp16 has 5 exponent bits, so the split is only that good while h1 stays a NORMAL number (|h1| >= 2^-14, i.e. |x| >~ 2^-3 ...
// below that h1 is rounded to a multiple of 2^-24: an ABSOLUTE error <= 2^-25) and |x| < 65504:
//   weights     are split once, offline, with every row scaled by a power of two so that its largest element lies in
//               [2^13, 2^14): exact, elements down to 2^-17 of the row's largest keep a normal h1, and the scale is undone
//               exactly in the epilogue (acc * 2^-k, fused with the bias add);
//   activations are split in registers while they are staged, unscaled: LayerNorm / GELU / SiLU outputs of this network are
//               O(1) (measured per layer: medians 0.08 .. 0.9, maxima <= 14 on the synthetic weights); elements below 2^-3 carry
//               an absolute error <= 2^-25 = 3e-8 each, elements beyond 65504 would overflow to infinity (xp_xpoint_forward
//               documents the limit; the bf16 six-product class has no such limit and stays selectable).
// Measured (oracle emulation of the whole network, tools/ + DESIGN.md §3c): reference prob error 5.9e-6 — the same as the f32
// oracle itself at a different thread count (4.8e-6) — against 5.9e-5 for a three-product bf16 split.
//
// Tile engine.  A workgroup of WM x WN waves owns a (WM*TM*32) x (WN*TN*32) tile; K is walked in 32-wide slabs = two MFMA
// k-steps x three products (24 MFMAs per wave and barrier for a 2 x 2 wave tile, as many as the six-product engine has per
// 16-wide slab).  LDS row = [plane 0: 32 fp16][plane 1: 32 fp16] + 16 B pad = 144 B: ds_read_b128 fragment reads of 32
// consecutive rows and the ds_write_b64 / b128 staging stores are conflict-free (rows are permuted inside groups of 8 for the
// b64 stores).  Two buffers, one LDS-only barrier per slab, global loads two to three slabs ahead in registers, the split
// arithmetic of the next slab in the shadow of the second k-step's MFMAs.
#pragma once
#include "gemm_epilogue.h"

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

#ifndef XP_H2_DBG
#define XP_H2_DBG 0   /* timing experiments only (wrong results), tile engine: 1 no split VALU, 2 no global loads after the prologue, 4 no MFMA, 8 no LDS stores, 16 no barrier, 32 no fragment reads after the first slab */
#endif
constexpr int H2_BK = 32;           // k per slab
constexpr int H2_ROWB = 144;        // LDS bytes per tile row
constexpr int H2_SLAB_UNITS = 8;    // 16-byte units per (weight row, slab) in the offline layout: 2 planes x 4 octets

__device__ __forceinline__ void h2_lds_barrier() {
    if (XP_H2_DBG & 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Two-way fp16 split of four floats: planes as packed fp16 quads.  Two vector instructions per value (round 2 had four: the compiler turned
// `(float)h0` + subtract into v_cvt_f32_f16 + v_sub_f32 and converted h0 twice): h0 pair = one v_cvt_pk_f16_f32 (round to nearest even), the residual
// x - h0 = ONE v_fma_mix_f32 per value reading the packed half in place (fma(h0, -1, x): the difference is exactly representable, so the single rounding
// is exact), h1 pair = one v_cvt_pk_f16_f32.  The K loop of the tile engine went from 4.0 to 2.x vector instructions per MFMA (profiles/r3_gemm_h2_stalls.txt:
// the waves were waiting to ISSUE, not for LDS).
__device__ __forceinline__ float h2_resid_lo(unsigned h, float x) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x)); return r; }
__device__ __forceinline__ float h2_resid_hi(unsigned h, float x) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x)); return r; }
__device__ __forceinline__ void h2_split4(const float4& v, uint2& p0, uint2& p1) {
    union { f16x2_t h; unsigned u; } a, b, c, d;
    a.h = f16x2_t{(_Float16)v.x, (_Float16)v.y};
    b.h = f16x2_t{(_Float16)v.z, (_Float16)v.w};
    c.h = f16x2_t{(_Float16)h2_resid_lo(a.u, v.x), (_Float16)h2_resid_hi(a.u, v.y)};
    d.h = f16x2_t{(_Float16)h2_resid_lo(b.u, v.z), (_Float16)h2_resid_hi(b.u, v.w)};
    p0 = make_uint2(a.u, b.u);
    p1 = make_uint2(c.u, d.u);
}

template <int WM, int WN, int TM, int TN>
struct GemmTileH2 {
    static constexpr int BK = H2_BK;
    static constexpr bool kRowScale = true;                 // accumulators carry the weight rows' power-of-two scale (gemm_epilogue.h)
    static constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
    static constexpr int A_TOT = BM * 8;                    // (row, k-quad) staging slots per slab: 16 B of f32 each
    static constexpr int B_TOT = BN * H2_SLAB_UNITS;        // 16-byte units per slab
    static constexpr int A_LD = (A_TOT + NT - 1) / NT, B_LD = (B_TOT + NT - 1) / NT;
    static constexpr int kBufBytes = (BM + BN) * H2_ROWB;
    static constexpr size_t kLdsBytes = 2 * (size_t)kBufBytes;
    // staging slots are dealt round-robin; surplus threads of a partly needed round repeat the last slot (same data, same LDS
    // address): no divergent region in the K loop
    __device__ static __forceinline__ int a_id(int s) { const int id = (int)threadIdx.x + s * NT; return (s + 1) * NT <= A_TOT ? id : (id < A_TOT ? id : A_TOT - 1); }
    __device__ static __forceinline__ int b_id(int s) { const int id = (int)threadIdx.x + s * NT; return (s + 1) * NT <= B_TOT ? id : (id < B_TOT ? id : B_TOT - 1); }
    // slot -> tile row: eight consecutive slots cover the eight k-quads of one row; the rows of an aligned group of 8 are taken in the
    // order 0,4,1,5,2,6,3,7 so that the two rows of a 16-lane ds_write_b64 group sit 16 banks apart (row stride 36 dwords)
    __device__ static __forceinline__ int a_row(int s) { const int r = a_id(s) >> 3; return (r & ~7) | ((r & 7) >> 1) | ((r & 1) << 2); }
    __device__ static __forceinline__ int a_quad(int s) { return a_id(s) & 7; }
    __device__ static __forceinline__ int b_row(int s) { return b_id(s) / H2_SLAB_UNITS; }
    __device__ static __forceinline__ int b_unit(int s) { return b_id(s) % H2_SLAB_UNITS; }   // plane * 4 + octet

    struct RawA { float4 a[A_LD]; bool ok[A_LD]; };      // f32 A values of one slab as loaded
    struct SplitA { uint2 p[A_LD][2]; };                 // the same slab as its two fp16 planes
    struct RawB { uint4 b[B_LD]; };                      // offline-split weights of one slab as loaded

    // ldA(slot, k, v) -> ok: the 4 consecutive f32 starting at absolute k of the slot's row, loaded unconditionally from a valid
    //                        address; ok = whether they are real (else the slot is staged as zeros).  Called once per slot and slab,
    //                        in slab order.
    // ldB(slot, slab)     -> the slot's 16-byte unit of the offline-split weights (rows past N clamped by the caller; k past K is
    //                        zero in the offline layout)
    // Per slab t (buffers alternate; slabs past K contribute exact zeros):
    //     ds_read   fragments of k-step 0 of slab t, 12 MFMAs
    //     ds_write  slab t+1 (A planes split during slab t-1, B as loaded during slab t-1) -> the other buffer
    //     global    loads of B slab t+2 and A slab t+3 into the registers just freed
    //     ds_read   fragments of k-step 1, 12 MFMAs with the split arithmetic of A slab t+2 in their shadow
    template <class LA, class LB>
    __device__ static __forceinline__ void run(unsigned char* lds, int K, LA ldA, LB ldB, f32x16 (&acc)[TM][TN]) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int wm = wave / WN, wn = wave % WN;
        const int fr = lane & 31, fh = lane >> 5;
        auto gloadA = [&](RawA& r, int t) {
            if ((XP_H2_DBG & 2) && t > 3) return;
#pragma unroll
            for (int s = 0; s < A_LD; ++s) r.ok[s] = ldA(s, t * H2_BK + a_quad(s) * 4, r.a[s]);
        };
        auto gloadB = [&](RawB& r, int t) {
            if ((XP_H2_DBG & 2) && t > 3) return;
#pragma unroll
            for (int s = 0; s < B_LD; ++s) r.b[s] = ldB(s, t);
        };
        auto split = [&](const RawA& r, SplitA& o) {
            if (XP_H2_DBG & 1) {
#pragma unroll
                for (int s = 0; s < A_LD; ++s) { o.p[s][0] = make_uint2(__float_as_uint(r.a[s].x), __float_as_uint(r.a[s].y)); o.p[s][1] = make_uint2(__float_as_uint(r.a[s].z), __float_as_uint(r.a[s].w)); }
                return;
            }
#pragma unroll
            for (int s = 0; s < A_LD; ++s) {
                const unsigned m = r.ok[s] ? 0xffffffffu : 0u;       // not-real slots become zeros by masking the input bits
                auto mk = [&](float v) { return __uint_as_float(__float_as_uint(v) & m); };
                h2_split4(make_float4(mk(r.a[s].x), mk(r.a[s].y), mk(r.a[s].z), mk(r.a[s].w)), o.p[s][0], o.p[s][1]);
            }
        };
        int a_dst[A_LD], b_dst[B_LD];
#pragma unroll
        for (int s = 0; s < A_LD; ++s) a_dst[s] = a_row(s) * H2_ROWB + a_quad(s) * 8;
#pragma unroll
        for (int s = 0; s < B_LD; ++s) b_dst[s] = BM * H2_ROWB + b_row(s) * H2_ROWB + b_unit(s) * 16;   // a straight copy of the offline layout
        auto lstore = [&](const SplitA& sa, const RawB& rb, unsigned char* buf) {
            if ((XP_H2_DBG & 8) && buf != lds) return;
#pragma unroll
            for (int s = 0; s < A_LD; ++s) {
                *reinterpret_cast<uint2*>(buf + a_dst[s]) = sa.p[s][0];
                *reinterpret_cast<uint2*>(buf + a_dst[s] + 64) = sa.p[s][1];
            }
#pragma unroll
            for (int s = 0; s < B_LD; ++s) *reinterpret_cast<uint4*>(buf + b_dst[s]) = rb.b[s];
        };
        const int a_frag = (wm * TM * 32 + fr) * H2_ROWB + 16 * fh;
        const int b_frag = BM * H2_ROWB + (wn * TN * 32 + fr) * H2_ROWB + 16 * fh;
        f16x8_t af[2][TM], bf[2][TN];
        bool frag_first = true;
        auto frags = [&](const unsigned char* buf, int ks) {
            if ((XP_H2_DBG & 32) && !frag_first) return;
            frag_first = false;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[pl][i] = *reinterpret_cast<const f16x8_t*>(buf + a_frag + pl * 64 + ks * 32 + i * 32 * H2_ROWB);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[pl][j] = *reinterpret_cast<const f16x8_t*>(buf + b_frag + pl * 64 + ks * 32 + j * 32 * H2_ROWB);
            }
        };
        auto mfmas = [&]() {
            constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};          // smallest partial products first
#pragma unroll
            for (int pp = 0; pp < 3; ++pp)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if (XP_H2_DBG & 4) { if (pp == 0) acc[i][j][0] += (float)af[0][i][0] * (float)bf[0][j][0] + (float)af[1][i][1] * (float)bf[1][j][1]; }
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[PA[pp]][i], bf[PB[pp]][j], acc[i][j], 0, 0, 0);
                    }
        };
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        unsigned char* bufs[2] = {lds, lds + kBufBytes};
        const int nslab = ((K + H2_BK - 1) / H2_BK + 1) & ~1;       // even; a slab past K adds exact zeros
        RawA ra[2]; RawB rb; SplitA sp;
        // slab t: its A values are loaded during slab t-3 into ra[t & 1] and split during slab t-2; its B planes are loaded
        // during slab t-2 into rb and stored, with the A planes, during slab t-1
        gloadA(ra[0], 0); gloadB(rb, 0);
        gloadA(ra[1], 1);
        split(ra[0], sp);
        lstore(sp, rb, bufs[0]);
        gloadB(rb, 1); gloadA(ra[0], 2);
        split(ra[1], sp);
        h2_lds_barrier();
        auto step = [&](int t, auto u_tag) {
            constexpr int U = decltype(u_tag)::value;          // t & 1
            frags(bufs[U], 0);
            __builtin_amdgcn_sched_barrier(0);
            mfmas();
            __builtin_amdgcn_sched_barrier(0);
            lstore(sp, rb, bufs[U ^ 1]);                        // slab t+1
            __builtin_amdgcn_sched_barrier(0);
            gloadB(rb, t + 2); gloadA(ra[U ^ 1], t + 3);
            frags(bufs[U], 1);
            mfmas();
            split(ra[U], sp);                                   // A slab t+2 (loaded during slab t-1)
            h2_lds_barrier();
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

// ---------------------------------------------------------------------------------------------------------------------
// Row-stationary variant: the A operand never touches LDS.  In the tile engine above the LDS pipe is the busiest unit of the CU
// (per 32-wide slab and CU: 64 KB of A/B staging stores + 128 KB of fragment reads against 1536 matrix-pipe cycles: ~85 % busy).
// Here a workgroup is 4 waves x 32 rows (BM = 128), every wave owns ALL TN * 32 columns of its rows, so no other wave needs its A
// rows: each lane loads the 8 consecutive f32 of its row that make up its MFMA fragment (lane l: row l & 31, k = 8 (l >> 5) .. + 7)
// straight from global memory, splits them in registers and feeds the MFMAs from there.  Only the weights go through LDS
// (straight 16-byte copies of the offline layout, shared by the four waves): LDS traffic per slab and CU drops to a third.
// ---------------------------------------------------------------------------------------------------------------------
template <int TN>
struct GemmTileH2R {
    static constexpr int BK = H2_BK;
    static constexpr bool kRowScale = true;
    static constexpr int BM = 128, BN = TN * 32, NT = 256;
    static constexpr int B_TOT = BN * H2_SLAB_UNITS;
    static constexpr int B_LD = (B_TOT + NT - 1) / NT;       // = TN
    static constexpr int kBufBytes = BN * H2_ROWB;
    static constexpr size_t kLdsBytes = 2 * (size_t)kBufBytes;
    __device__ static __forceinline__ int b_id(int s) { const int id = (int)threadIdx.x + s * NT; return (s + 1) * NT <= B_TOT ? id : (id < B_TOT ? id : B_TOT - 1); }
    __device__ static __forceinline__ int b_row(int s) { return b_id(s) / H2_SLAB_UNITS; }
    __device__ static __forceinline__ int b_unit(int s) { return b_id(s) % H2_SLAB_UNITS; }
    __device__ static __forceinline__ int a_row() { return (int)(threadIdx.x >> 6) * 32 + (int)(threadIdx.x & 31); }   // this lane's tile row

    struct RawA { float4 lo[2], hi[2]; bool ok[2]; };     // the lane's 2 x 8 floats of one slab (k-steps 0 and 1)
    struct PlanesA { unsigned p[2][2][4]; };              // [k-step][plane] as MFMA operand bits
    struct RawB { uint4 b[B_LD]; };

    // ldA8(k, lo, hi) -> ok: the 8 consecutive f32 of THIS LANE's row starting at absolute k (a multiple of 8), loaded unconditionally
    //                        from a valid address; ok = real (else they count as zeros).  Called twice per slab, in k order.
    // ldB(slot, slab)      : as in GemmTileH2.
    template <class LA, class LB>
    __device__ static __forceinline__ void run(unsigned char* lds, int K, LA ldA8, LB ldB, f32x16 (&acc)[1][TN]) {
        const int lane = threadIdx.x & 63;
        const int fr = lane & 31, fh = lane >> 5;
        auto gloadA = [&](RawA& r, int t) {
#pragma unroll
            for (int s = 0; s < 2; ++s) r.ok[s] = ldA8(t * H2_BK + s * 16 + fh * 8, r.lo[s], r.hi[s]);
        };
        auto gloadB = [&](RawB& r, int t) {
#pragma unroll
            for (int s = 0; s < B_LD; ++s) r.b[s] = ldB(s, t);
        };
        auto split = [&](const RawA& r, PlanesA& o) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const unsigned m = r.ok[s] ? 0xffffffffu : 0u;
                auto mk = [&](float v) { return __uint_as_float(__float_as_uint(v) & m); };
                uint2 a0, a1, b0, b1;
                h2_split4(make_float4(mk(r.lo[s].x), mk(r.lo[s].y), mk(r.lo[s].z), mk(r.lo[s].w)), a0, a1);
                h2_split4(make_float4(mk(r.hi[s].x), mk(r.hi[s].y), mk(r.hi[s].z), mk(r.hi[s].w)), b0, b1);
                o.p[s][0][0] = a0.x; o.p[s][0][1] = a0.y; o.p[s][0][2] = b0.x; o.p[s][0][3] = b0.y;
                o.p[s][1][0] = a1.x; o.p[s][1][1] = a1.y; o.p[s][1][2] = b1.x; o.p[s][1][3] = b1.y;
            }
        };
        int b_dst[B_LD];
#pragma unroll
        for (int s = 0; s < B_LD; ++s) b_dst[s] = b_row(s) * H2_ROWB + b_unit(s) * 16;
        auto lstoreB = [&](const RawB& rb, unsigned char* buf) {
#pragma unroll
            for (int s = 0; s < B_LD; ++s) *reinterpret_cast<uint4*>(buf + b_dst[s]) = rb.b[s];
        };
        const int b_frag = fr * H2_ROWB + 16 * fh;
        typedef unsigned ubits4 __attribute__((ext_vector_type(4)));
        auto kstep = [&](const unsigned char* buf, const PlanesA& ap, int ks) {
            f16x8_t bf[2][TN];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[pl][j] = *reinterpret_cast<const f16x8_t*>(buf + b_frag + pl * 64 + ks * 32 + j * 32 * H2_ROWB);
            const f16x8_t a0 = __builtin_bit_cast(f16x8_t, ubits4{ap.p[ks][0][0], ap.p[ks][0][1], ap.p[ks][0][2], ap.p[ks][0][3]});
            const f16x8_t a1 = __builtin_bit_cast(f16x8_t, ubits4{ap.p[ks][1][0], ap.p[ks][1][1], ap.p[ks][1][2], ap.p[ks][1][3]});
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bf[0][j], acc[0][j], 0, 0, 0);      // smallest products first
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bf[1][j], acc[0][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bf[0][j], acc[0][j], 0, 0, 0);
        };
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;

        unsigned char* bufs[2] = {lds, lds + kBufBytes};
        const int nslab = ((K + H2_BK - 1) / H2_BK + 1) & ~1;       // even; a slab past K adds exact zeros
        RawA ra[2]; RawB rb; PlanesA ap[2];
        // slab t: A loaded during slab t-2 into ra[t & 1], split during slab t-1 into ap[t & 1]; B loaded during slab t-2, stored during t-1
        gloadA(ra[0], 0); gloadB(rb, 0);
        gloadA(ra[1], 1);
        split(ra[0], ap[0]);
        lstoreB(rb, bufs[0]);
        gloadB(rb, 1); gloadA(ra[0], 2);
        h2_lds_barrier();
        auto step = [&](int t, auto u_tag) {
            constexpr int U = decltype(u_tag)::value;          // t & 1
            kstep(bufs[U], ap[U], 0);
            __builtin_amdgcn_sched_barrier(0);
            lstoreB(rb, bufs[U ^ 1]);                           // B slab t+1
            gloadB(rb, t + 2);
            kstep(bufs[U], ap[U], 1);
            split(ra[U ^ 1], ap[U ^ 1]);                        // A slab t+1 (loaded during slab t-1) ...
            gloadA(ra[U ^ 1], t + 3);                           // ... whose registers then take slab t+3
            h2_lds_barrier();
        };
        for (int t = 0; t < nslab; t += 2) { step(t, std::integral_constant<int, 0>{}); step(t + 1, std::integral_constant<int, 1>{}); }
    }

    __device__ static __forceinline__ int row_of(int, int r) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        return wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    }
    __device__ static __forceinline__ int col_of(int j) { return j * 32 + (int)(threadIdx.x & 31); }
};
