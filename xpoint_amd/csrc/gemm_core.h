// fp32 MFMA tile engine shared by the dense-layer GEMM / implicit-GEMM conv (gemm.hip) and the
// descriptor-distance kernel (match.hip).
//
//   acc[i][j] (32x32 f32 tiles) += sum_k A[m, k] * B[n, k]      both operands K-contiguous ("NT")
//
// CDNA4 mapping: exact-f32 MFMA v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate; bit-for-bit a k-ordered
// fmaf chain).  A workgroup of WM x WN waves owns a (WM*TM*32) x (WN*TN*32) tile; K is walked in 32-wide
// slabs staged through double-buffered LDS with register prefetch (the global loads of slab t+1 are in
// flight during the MFMAs of slab t; one barrier per slab).  LDS rows are padded to 36 floats so that the
// ds_read_b128 fragment reads are bank-conflict-free (slot = (9*row + s) mod 16 is a bijection on each
// 16-lane group).  Each lane reads 4 consecutive k per ds_read_b128: MFMA step j of lane-half h consumes
// k = 8q + 4h + j for BOTH operands — a permutation of the k order, which the sum does not care about.
#pragma once
#include "xp_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef XP_BK_VALUE
#define XP_BK_VALUE 32
#endif
constexpr int XP_BK = XP_BK_VALUE;
constexpr int XP_LDS_STRIDE = XP_BK + 4;
#ifndef XP_STORE_AT_Q
#define XP_STORE_AT_Q (-1)   /* -1: after the last MFMA group (measured best); q: before MFMA group q */
#endif

template <int WM, int WN, int TM, int TN>
struct GemmTile {
    static constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
    static constexpr int A_TOT = BM * (XP_BK / 4), B_TOT = BN * (XP_BK / 4);      // float4 staging slots per slab
    static constexpr int A_LD = (A_TOT + NT - 1) / NT, B_LD = (B_TOT + NT - 1) / NT;
    __device__ static __forceinline__ bool a_slot_ok(int s) { return A_TOT % NT == 0 || (int)threadIdx.x + s * NT < A_TOT; }
    __device__ static __forceinline__ bool b_slot_ok(int s) { return B_TOT % NT == 0 || (int)threadIdx.x + s * NT < B_TOT; }
    static constexpr size_t kLdsBytes = sizeof(float) * 2 * (BM + BN) * XP_LDS_STRIDE;

    // staging slot s of this thread covers tile row slot_row(s), k-quad slot_kq(s)
    __device__ static __forceinline__ int slot_row(int s) { return (threadIdx.x + s * NT) / (XP_BK / 4); }
    __device__ static __forceinline__ int slot_kq(int s) { return (threadIdx.x + s * NT) % (XP_BK / 4); }

    // ldA(slot, k, ok) / ldB(slot, k, ok): float4 of 4 consecutive k starting at absolute k, loaded UNCONDITIONALLY
    // from a valid (possibly redirected) address; ok = whether the value is real.  Invalid values are zeroed when the
    // registers are written to LDS, i.e. AFTER the MFMAs of the current slab: the loads stay in flight behind the
    // matrix work (a select right after the load would make hipcc wait for the data before the MFMAs).
    template <class LA, class LB>
    __device__ static __forceinline__ void run(float* lds, int K, LA ldA, LB ldB, f32x16 (&acc)[TM][TN]) {
        float* As = lds;
        float* Bs = lds + 2 * BM * XP_LDS_STRIDE;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int wm = wave / WN, wn = wave % WN;
        const int fr = lane & 31, fh = lane >> 5;
        float4 ra[A_LD], rb[B_LD];
        bool oka[A_LD], okb[B_LD];
        auto gload = [&](int k0) {
#pragma unroll
            for (int s = 0; s < A_LD; ++s) if (a_slot_ok(s)) ra[s] = ldA(s, k0 + slot_kq(s) * 4, oka[s]);
#pragma unroll
            for (int s = 0; s < B_LD; ++s) if (b_slot_ok(s)) rb[s] = ldB(s, k0 + slot_kq(s) * 4, okb[s]);
        };
        auto lstore = [&](int buf) {
            __builtin_amdgcn_sched_barrier(0);   // keep the selects (first use of the loaded registers) below the earlier MFMAs
#pragma unroll
            for (int s = 0; s < A_LD; ++s) {
                if (!a_slot_ok(s)) continue;
                float4 v = ra[s];
                v.x = oka[s] ? v.x : 0.f; v.y = oka[s] ? v.y : 0.f; v.z = oka[s] ? v.z : 0.f; v.w = oka[s] ? v.w : 0.f;
                *reinterpret_cast<float4*>(As + (buf * BM + slot_row(s)) * XP_LDS_STRIDE + slot_kq(s) * 4) = v;
            }
#pragma unroll
            for (int s = 0; s < B_LD; ++s) {
                if (!b_slot_ok(s)) continue;
                float4 v = rb[s];
                v.x = okb[s] ? v.x : 0.f; v.y = okb[s] ? v.y : 0.f; v.z = okb[s] ? v.z : 0.f; v.w = okb[s] ? v.w : 0.f;
                *reinterpret_cast<float4*>(Bs + (buf * BN + slot_row(s)) * XP_LDS_STRIDE + slot_kq(s) * 4) = v;
            }
            __builtin_amdgcn_sched_barrier(0);
        };
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        const int nt = (K + XP_BK - 1) / XP_BK;
        gload(0);
        lstore(0);
        __syncthreads();
        for (int t = 0; t < nt; ++t) {
            const int buf = t & 1;
            if (t + 1 < nt) gload((t + 1) * XP_BK);
            const float* Ab = As + (buf * BM + wm * TM * 32 + fr) * XP_LDS_STRIDE + 4 * fh;
            const float* Bb = Bs + (buf * BN + wn * TN * 32 + fr) * XP_LDS_STRIDE + 4 * fh;
#pragma unroll
            for (int q = 0; q < XP_BK / 8; ++q) {
                // The prefetched slab goes to LDS BEFORE the last MFMA group (its buffer was released by the barrier
                // that ended the previous slab), so the selects, the ds_writes and their latency sit under 16 MFMAs
                // instead of between the last MFMA and the barrier.
                if (XP_STORE_AT_Q >= 0 && q == XP_STORE_AT_Q && t + 1 < nt) lstore(buf ^ 1);
                float4 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * XP_LDS_STRIDE + q * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4*>(Bb + j * 32 * XP_LDS_STRIDE + q * 8);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                    }
            }
            if (XP_STORE_AT_Q < 0 && t + 1 < nt) lstore(buf ^ 1);
            __syncthreads();
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
