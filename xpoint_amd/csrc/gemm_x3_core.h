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
// The A operand (activations, f32 in HBM) is split in registers while it is staged; the B operand (weights) is split once,
// offline (xp_split_weights_x3), into slab-interleaved planes so that staging B is a straight 16-byte copy.  LDS holds two
// slabs (double buffer, (BM + BN) rows x 112 B each: 56 KB for 128 x 128, two workgroups of 8 waves per CU), one LDS-only
// barrier per slab; global loads run two slabs ahead in registers.  An LDS row is [plane][16 bf16] + 16 B pad = 112 B:
// with ds_read_b128's lane groups the 16 lanes of a group land on 16 distinct 4-bank slots (28*row mod 64), conflict-free.
// MFMA operand layout (32x32x16): lane l holds 8 consecutive k (= 8*(l>>5) .. +7) of row/col l & 31.
#pragma once
#include "gemm_epilogue.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#ifndef XP_X3_DBG
#define XP_X3_DBG 0   /* timing experiments only (wrong results): 1 no split VALU, 2 no global loads after the prologue, 4 no MFMA, 8 no LDS store */
#endif
constexpr int X3_BK = 16;          // k per slab
constexpr int X3_ROWB = 112;       // LDS bytes per tile row: 3 planes x 16 bf16 (= the offline weight layout) + 16 B pad
constexpr int X3_SLAB_UNITS = 6;   // 16-byte units per (weight row, slab) in the offline layout: 3 planes x 2 octets

// Workgroup barrier that orders LDS traffic only: waits for this wave's LDS operations (lgkmcnt) and NOT for its
// outstanding global loads (vmcnt) — __syncthreads() drains both, which would cut the register prefetch to zero slabs.
__device__ __forceinline__ void xp_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// exact three-way bf16 split of two floats; returns the packed pairs (low half = x, high half = y) per plane
__device__ __forceinline__ void xp_split2(float x, float y, unsigned& p0, unsigned& p1, unsigned& p2) {
    union { bf16x2 v; unsigned u; } c;
    c.v[0] = (__bf16)x; c.v[1] = (__bf16)y; p0 = c.u;
    const float rx = x - __uint_as_float(p0 << 16), ry = y - __uint_as_float(p0 & 0xffff0000u);
    c.v[0] = (__bf16)rx; c.v[1] = (__bf16)ry; p1 = c.u;
    const float sx = rx - __uint_as_float(p1 << 16), sy = ry - __uint_as_float(p1 & 0xffff0000u);
    c.v[0] = (__bf16)sx; c.v[1] = (__bf16)sy; p2 = c.u;
}

__device__ __forceinline__ void xp_split8(const float4& lo, const float4& hi, uint4& p0, uint4& p1, uint4& p2) {
    xp_split2(lo.x, lo.y, p0.x, p1.x, p2.x);
    xp_split2(lo.z, lo.w, p0.y, p1.y, p2.y);
    xp_split2(hi.x, hi.y, p0.z, p1.z, p2.z);
    xp_split2(hi.z, hi.w, p0.w, p1.w, p2.w);
}

template <int WM, int WN, int TM, int TN>
struct GemmTileX3 {
    static constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
    static constexpr int A_TOT = BM * 2;                    // (row, k-octet) staging slots per slab
    static constexpr int B_TOT = BN * X3_SLAB_UNITS;        // 16-byte units per slab
    static constexpr int A_LD = (A_TOT + NT - 1) / NT, B_LD = (B_TOT + NT - 1) / NT;
    static constexpr int kBufBytes = (BM + BN) * X3_ROWB;
    static constexpr size_t kLdsBytes = 2 * (size_t)kBufBytes;
    // Staging work is dealt round-robin; when a round is only partly needed the surplus threads repeat the last slot
    // (same data to the same LDS address), which keeps the K loop free of divergent regions — those made hipcc drain
    // every outstanding global load at each barrier.
    __device__ static __forceinline__ int a_id(int s) { const int id = (int)threadIdx.x + s * NT; return (s + 1) * NT <= A_TOT ? id : (id < A_TOT ? id : A_TOT - 1); }
    __device__ static __forceinline__ int b_id(int s) { const int id = (int)threadIdx.x + s * NT; return (s + 1) * NT <= B_TOT ? id : (id < B_TOT ? id : B_TOT - 1); }
    __device__ static __forceinline__ int a_row(int s) { return a_id(s) >> 1; }
    __device__ static __forceinline__ int a_oct(int s) { return a_id(s) & 1; }
    __device__ static __forceinline__ int b_row(int s) { return b_id(s) / X3_SLAB_UNITS; }
    __device__ static __forceinline__ int b_unit(int s) { return b_id(s) % X3_SLAB_UNITS; }   // plane * 2 + octet

    struct Stage { float4 lo[A_LD], hi[A_LD]; bool ok[A_LD]; uint4 b[B_LD]; };

    // ldA(slot, k, lo, hi) -> ok: the 8 consecutive f32 starting at absolute k of the slot's row, loaded unconditionally
    //                             from a valid address; ok = whether they are real (else the slot is stored as zeros).
    //                             Called once per slot and slab, in slab order.
    // ldB(slot, slab)      -> the slot's 16-byte unit of the offline-split weights (rows past N are clamped: they only
    //                             feed output columns that are never stored; k past K is zero in the offline layout)
    template <class LA, class LB>
    __device__ static __forceinline__ void run(unsigned char* lds, int K, LA ldA, LB ldB, f32x16 (&acc)[TM][TN]) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int wm = wave / WN, wn = wave % WN;
        const int fr = lane & 31, fh = lane >> 5;
        auto gload = [&](Stage& r, int t) {
#pragma unroll
            for (int s = 0; s < A_LD; ++s) r.ok[s] = ldA(s, t * X3_BK + a_oct(s) * 8, r.lo[s], r.hi[s]);
#pragma unroll
            for (int s = 0; s < B_LD; ++s) r.b[s] = ldB(s, t);
        };
        auto lstore = [&](const Stage& r, unsigned char* buf) {
            unsigned char* As = buf;
            unsigned char* Bs = buf + BM * X3_ROWB;
#pragma unroll
            for (int s = 0; s < A_LD; ++s) {
                uint4 p0, p1, p2;
                // slots that are not real (k past K, conv zero padding) become zeros by masking the INPUT bits: selects,
                // not a branch
                const unsigned m = r.ok[s] ? 0xffffffffu : 0u;
                auto mk = [&](float v) { return __uint_as_float(__float_as_uint(v) & m); };
                const float4 lo = make_float4(mk(r.lo[s].x), mk(r.lo[s].y), mk(r.lo[s].z), mk(r.lo[s].w));
                const float4 hi = make_float4(mk(r.hi[s].x), mk(r.hi[s].y), mk(r.hi[s].z), mk(r.hi[s].w));
                if (XP_X3_DBG & 1) {
                    p0 = make_uint4(__float_as_uint(lo.x), __float_as_uint(lo.y), __float_as_uint(lo.z), __float_as_uint(lo.w));
                    p1 = make_uint4(__float_as_uint(hi.x), __float_as_uint(hi.y), __float_as_uint(hi.z), __float_as_uint(hi.w));
                    p2 = p0;
                } else {
                    xp_split8(lo, hi, p0, p1, p2);
                }
                unsigned char* d = As + a_row(s) * X3_ROWB + a_oct(s) * 16;
                *reinterpret_cast<uint4*>(d) = p0;
                *reinterpret_cast<uint4*>(d + 32) = p1;
                *reinterpret_cast<uint4*>(d + 64) = p2;
            }
#pragma unroll
            for (int s = 0; s < B_LD; ++s) {
                const int u = b_unit(s);
                *reinterpret_cast<uint4*>(Bs + b_row(s) * X3_ROWB + u * 16) = r.b[s];   // a straight copy of the offline layout
            }
        };
        auto compute = [&](const unsigned char* buf) {
            const unsigned char* Ab = buf + (wm * TM * 32 + fr) * X3_ROWB + 16 * fh;
            const unsigned char* Bb = buf + BM * X3_ROWB + (wn * TN * 32 + fr) * X3_ROWB + 16 * fh;
            bf16x8 af[3][TM], bf[3][TN];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[pl][i] = *reinterpret_cast<const bf16x8*>(Ab + pl * 32 + i * 32 * X3_ROWB);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[pl][j] = *reinterpret_cast<const bf16x8*>(Bb + pl * 32 + j * 32 * X3_ROWB);
            }
            // smallest partial products first; the TM*TN independent accumulators separate dependent MFMAs
            constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int pp = 0; pp < 6; ++pp)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if (XP_X3_DBG & 4) { if (pp == 0) acc[i][j][0] += (float)af[0][i][0] * (float)bf[0][j][0] + (float)af[1][i][1] * (float)bf[1][j][1] + (float)af[2][i][2] * (float)bf[2][j][2]; }
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[pp]][i], bf[PB[pp]][j], acc[i][j], 0, 0, 0);
                    }
        };
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        unsigned char* buf0 = lds;
        unsigned char* buf1 = lds + kBufBytes;
        // Slabs are processed in pairs with no conditional code: a slab past K stages zeros for A (ldA reports !ok) and
        // a clamped copy for B, so it adds exactly 0; loads run two slabs ahead and are never waited for at a barrier.
        const int npair = ((K + X3_BK - 1) / X3_BK + 1) / 2;
        constexpr bool LOADS = !(XP_X3_DBG & 2), STORES = !(XP_X3_DBG & 8);
        Stage ra, rb;      // even / odd slabs
        gload(ra, 0);
        gload(rb, 1);
        lstore(ra, buf0);
        xp_lds_barrier();
        // Issue order inside a slab (one scheduling region between two barriers): the global loads of slab t+2 and the
        // fragment reads of slab t first, then the MFMAs with the split arithmetic and the LDS stores of slab t+1 spread
        // between them — a wave issues in order, so anything left after the last MFMA would run with the matrix pipe idle.
        auto pipeline = [&]() {
#if !defined(XP_X3_NO_SGB)
            __builtin_amdgcn_sched_group_barrier(0x020, 2 * A_LD + B_LD, 0);          // VMEM reads
            __builtin_amdgcn_sched_group_barrier(0x100, 3 * (TM + TN), 0);            // DS reads
            constexpr int NMFMA = 6 * TM * TN, NVALU = 64 * A_LD + 8, NDSW = 3 * A_LD + B_LD;
            constexpr int VPER = (NVALU + NMFMA - 1) / NMFMA, WEVERY = NMFMA / NDSW > 0 ? NMFMA / NDSW : 1;
#pragma unroll
            for (int i = 0; i < NMFMA; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, VPER, 0);                 // a few VALU
                if (i % WEVERY == WEVERY - 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // one DS write
            }
#endif
        };
        for (int t = 0; t < 2 * npair; t += 2) {
            if (LOADS) gload(ra, t + 2);
            compute(buf0);
            if (STORES) lstore(rb, buf1);
            pipeline();
            xp_lds_barrier();
            if (LOADS) gload(rb, t + 3);
            compute(buf1);
            if (STORES) lstore(ra, buf0);
            pipeline();
            xp_lds_barrier();
        }
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
