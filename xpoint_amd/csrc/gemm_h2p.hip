// Ping-pong form of the split-fp16 GEMM (gemm_h2.hip): same arithmetic, operand formats and epilogue — the two waves of every SIMD take TURNS.
//
// Why (profiles/r3_gemm_h2_stage_removal.txt).  In the tile kernel the stage-removal deltas of the K loop add up to the whole loop and the
// matrix pipe is busy 0.33 of the time: inside one wave nothing overlaps a dependent MFMA chain (the cycles are the sums), and the second
// wave of a SIMD — a wave of ANOTHER workgroup running the same program — drifts into the same phase, so both want the matrix pipe, then both want LDS.
// Here the two waves of a SIMD belong to ONE 512-thread workgroup and are locked half a period apart by the workgroup barrier:
//     group 0 = waves 0..3, group 1 = waves 4..7 (workgroup waves go to the SIMDs cyclically: wave w and wave w + 4 share a SIMD);
//     phase p:  group p % 2       COMPUTE  the 24 MFMAs of slab p on fragments already in registers — nothing else;
//               the other group   MEMORY   ds_read the fragments of slab p + 1 (its next slab), split + ds_write slab p + 2 into the buffer slab p
//                                          was read from, issue the global loads of a slab three of its turns ahead;
//     s_barrier, roles swap.
// Every group accumulates the slabs of ITS parity (an in-workgroup split of K in two): wave w and wave w + 4 hold partial sums of the same 64 x 64
// sub-tile and exchange halves through LDS once, after the loop — each finishes 32 rows x 64 columns in the shared epilogue.  A 128 x 128 tile per
// workgroup, 2 x 36 KB of LDS, one workgroup (8 waves) per CU.  Summation order differs from the tile kernel (even slabs + odd slabs); same class.
#include <stdlib.h>

#include "gemm_h2_core.h"

namespace {

constexpr int P_BM = 128, P_BN = 128;
constexpr int P_BUF = (P_BM + P_BN) * H2_ROWB;           // [A rows: 2 planes][B rows: 2 planes], 144-byte rows
constexpr size_t P_LDS = 2 * (size_t)P_BUF;              // 73 728 B
constexpr int P_ALD = 4, P_BLD = 4;                      // staging slots per thread of the staging group and slab (256 threads: 1024 A quads, 1024 B units)

#ifndef XP_H2P_DBG
#define XP_H2P_DBG 0   /* timing experiments only (wrong results): 1 no MFMA, 2 no staging (split, stores, loads), 4 no fragment reads, 8 no split VALU, 16 no LDS stores after the prologue, 32 no global loads after the prologue */
#endif

// Waves w and w + 4 of a workgroup share a SIMD (measured: pairing w with w ^ 1 doubles the MFMA-only time), so group = wave >> 2.  The group index is
// deliberately NOT passed through readfirstlane: with a provably uniform branch hipcc's register allocation starved group 0 in both roles (124 vs 84 us).
__device__ __forceinline__ int p_grp() { return (int)(threadIdx.x >> 8); }            // which half of the workgroup a wave is in
__device__ __forceinline__ int p_w4() { return (int)(threadIdx.x >> 6) & 3; }         // its index inside the half

// what the shared epilogue needs to know: after the exchange every wave holds ONE 32-row block x 64 columns of the tile
struct PEpiTile {
    static constexpr int BM = P_BM, BN = P_BN;
    static constexpr bool kRowScale = true;
    __device__ static __forceinline__ int row_of(int, int r) {
        const int lane = threadIdx.x & 63;
        return (((p_w4() >> 1) * 2 + p_grp()) * 32) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    }
    __device__ static __forceinline__ int col_of(int j) {
        const int lane = threadIdx.x & 63;
        return ((p_w4() & 1) * 2 + j) * 32 + (lane & 31);
    }
};

__device__ unsigned long long g_h2p_stamps[8][128];     // XP_H2P_DBG & 64: s_memtime at every barrier entry / exit of the waves of workgroup 0
__device__ __forceinline__ void p_stamp(int& n) {
    if ((XP_H2P_DBG & 64) && blockIdx.x == 0 && (threadIdx.x & 63) == 0 && n < 128) g_h2p_stamps[threadIdx.x >> 6][n] = __builtin_amdgcn_s_memtime();
    ++n;
}
__device__ __forceinline__ void p_barrier_raw() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

typedef unsigned p_u32x4 __attribute__((ext_vector_type(4)));
struct PRawA { p_u32x4 a[P_ALD]; };      // f32 bits
struct PRawB { p_u32x4 b[P_BLD]; };
struct PFrags { f16x8_t a[2][2][2], b[2][2][2]; };       // [k-step][plane][block]

__global__ __launch_bounds__(512) void gemm_h2p_kernel(GemmParams p) {
    extern __shared__ __align__(16) unsigned char lds_h2p[];
    // XCD-aware tile order and column-tile groups: as gemm_h2_kernel
    const int ntn = (p.N + P_BN - 1) / P_BN;
    const int total = gridDim.x;
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int q = total >> 3, rr8 = total & 7;
    const int logical = (xcd < rr8 ? xcd * (q + 1) : rr8 * (q + 1) + (xcd - rr8) * q) + slot;
    int mt, nt;
    if (p.ngroup > 0) {
        const int ntm = (p.M + P_BM - 1) / P_BM;
        const int per = ntm * p.ngroup;
        const int g = logical / per, rem = logical - g * per;
        const int gw = min(p.ngroup, ntn - g * p.ngroup);
        mt = rem / gw; nt = g * p.ngroup + (rem - mt * gw);
    } else {
        mt = logical / ntn; nt = logical - mt * ntn;
    }
    const int m0 = mt * P_BM, n0 = nt * P_BN;

    int n_stamp = 0;
    auto p_barrier = [&]() { if (XP_H2P_DBG & 64) p_stamp(n_stamp); p_barrier_raw(); if (XP_H2P_DBG & 64) p_stamp(n_stamp); };
    const int tid = threadIdx.x, lane = tid & 63, grp = p_grp(), w4 = p_w4(), t256 = w4 * 64 + lane;
    const int wm = w4 >> 1, wn = w4 & 1, fr = lane & 31, fh = lane >> 5;
    // staging slots of this thread inside its group (the layout of GemmTileH2: eight consecutive slots = the eight k-quads of a row, rows of a
    // group of 8 in the order 0,4,1,5,2,6,3,7 for conflict-free ds_write_b64)
    auto a_row = [&](int s) { const int r = (t256 + s * 256) >> 3; return (r & ~7) | ((r & 7) >> 1) | ((r & 1) << 2); };
    auto a_quad = [&](int s) { return (t256 + s * 256) & 7; };
    auto b_row = [&](int s) { return (t256 + s * 256) >> 3; };
    auto b_unit = [&](int s) { return (t256 + s * 256) & 7; };
    // byte offsets of this thread's staging slots from p.A / p.Wt at slab 0 (rows past M / N clamped to row 0: they only reach outputs that are never stored);
    // slot s is slot 0 moved down 32 tile rows, so the LDS side needs one address per operand plus immediates
    unsigned a_offc[P_ALD], w_offc[P_BLD];
#pragma unroll
    for (int s = 0; s < P_ALD; ++s) {
        const int m = m0 + a_row(s), n = n0 + b_row(s);
        a_offc[s] = (unsigned)((m < p.M ? m : 0) * p.lda + a_quad(s) * 4) * 4u;
        w_offc[s] = (unsigned)((n < p.N ? n : 0) * H2_SLAB_UNITS + b_unit(s)) * 16u;
    }
    const int a_dst0 = a_row(0) * H2_ROWB + a_quad(0) * 8;
    const int b_dst0 = P_BM * H2_ROWB + b_row(0) * H2_ROWB + b_unit(0) * 16;
    constexpr int kSlotLds = 32 * H2_ROWB;
    const int nslab = p.K / H2_BK;                                                  // K % 64 == 0 (host): whole slabs, an even number of them
    const unsigned w_slab_bytes = (unsigned)p.N * (H2_SLAB_UNITS * 16u);
    auto gload = [&](PRawA& ra, PRawB& rb, int t) {           // slab t; the look-ahead past the last slab re-reads the last one (staged, never multiplied)
        if ((XP_H2P_DBG & (2 | 32)) && t > 5) return;
        const int tc = t < nslab ? t : nslab - 1;
        const unsigned sa = (unsigned)tc * (H2_BK * 4u), sw = (unsigned)tc * w_slab_bytes;      // scalar slab offsets
#pragma unroll
        for (int s = 0; s < P_ALD; ++s) ra.a[s] = *reinterpret_cast<const p_u32x4*>(reinterpret_cast<const char*>(p.A) + ((size_t)a_offc[s] + sa));
#pragma unroll
        for (int s = 0; s < P_BLD; ++s) rb.b[s] = *reinterpret_cast<const p_u32x4*>(reinterpret_cast<const char*>(p.Wt) + ((size_t)w_offc[s] + sw));
    };
    auto stage = [&](const PRawA& ra, const PRawB& rb, unsigned char* buf, bool first = false) {     // split + ds_write of one slab
#pragma unroll
        for (int s = 0; s < P_ALD; ++s) {
            uint2 p0, p1;
            if (XP_H2P_DBG & 8) { p0 = make_uint2(ra.a[s][0], ra.a[s][1]); p1 = make_uint2(ra.a[s][2], ra.a[s][3]); }
            else h2_split4(make_float4(__uint_as_float(ra.a[s][0]), __uint_as_float(ra.a[s][1]), __uint_as_float(ra.a[s][2]), __uint_as_float(ra.a[s][3])), p0, p1);
            if ((XP_H2P_DBG & 16) && !first) { if (p0.x == 0x12345u && p1.y == 0x54321u) *reinterpret_cast<uint2*>(buf + a_dst0 + s * kSlotLds) = p0; continue; }
            *reinterpret_cast<uint2*>(buf + a_dst0 + s * kSlotLds) = p0;
            *reinterpret_cast<uint2*>(buf + a_dst0 + s * kSlotLds + 64) = p1;
        }
#pragma unroll
        for (int s = 0; s < P_BLD; ++s) {
            if ((XP_H2P_DBG & 16) && !first) { if (rb.b[s][0] == 0x12345u) *reinterpret_cast<p_u32x4*>(buf + b_dst0 + s * kSlotLds) = rb.b[s]; continue; }
            *reinterpret_cast<p_u32x4*>(buf + b_dst0 + s * kSlotLds) = rb.b[s];
        }
    };
    const int a_frag = (wm * 64 + fr) * H2_ROWB + 16 * fh;
    const int b_frag = P_BM * H2_ROWB + (wn * 64 + fr) * H2_ROWB + 16 * fh;
    auto read_frags = [&](const unsigned char* buf, PFrags& f) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    f.a[ks][pl][i] = *reinterpret_cast<const f16x8_t*>(buf + a_frag + pl * 64 + ks * 32 + i * 32 * H2_ROWB);
                    f.b[ks][pl][i] = *reinterpret_cast<const f16x8_t*>(buf + b_frag + pl * 64 + ks * 32 + i * 32 * H2_ROWB);
                }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto mfmas = [&](const PFrags& f) {
        if (XP_H2P_DBG & 1) { acc[0][0][0] += (float)f.a[0][0][0][0] * (float)f.b[1][1][1][1]; return; }
        constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};              // smallest partial products first
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int pp = 0; pp < 3; ++pp)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[ks][PA[pp]][i], f.b[ks][PB[pp]][j], acc[i][j], 0, 0, 0);
    };

    // This group STAGES the slabs of the other parity (st(i) = (grp ^ 1) + 2 i: group 1 stages 0, 2, 4 .., group 0 stages 1, 3, 5 ..) into the buffer of that
    // parity and CONSUMES the slabs of its own parity from the other buffer.  The global loads of a staging turn are issued at the START OF THE COMPUTE
    // PHASE one and a half turns earlier (ring of two register sets): the texture path works through them in the shadow of the wave's own MFMAs, and the
    // memory phase is left with fragments, split and LDS stores only (with the loads at its end every memory phase ran ~500 cycles over its partner's MFMAs).
    unsigned char* const wbuf = lds_h2p + (grp ^ 1) * P_BUF;
    const unsigned char* const rbuf = lds_h2p + grp * P_BUF;
    const int st0 = grp ^ 1;
    const int turns = nslab >> 1;                                        // per group: one slab of its parity per turn
    PRawA ra0, ra1; PRawB rb0, rb1; PFrags fr_;
    gload(ra0, rb0, st0);
    gload(ra1, rb1, st0 + 2);
    stage(ra0, rb0, wbuf, true);
    if (grp) gload(ra0, rb0, st0 + 4);
    p_barrier();                                                         // slabs 0 and 1 are in LDS
    if (grp == 0) read_frags(rbuf, fr_);                                 // slab 0
    p_barrier();                                                         // ... before group 1 overwrites buffer 0 with slab 2
    auto memory_phase = [&](PRawA& ra, PRawB& rb) {                      // fragments of this group's next slab, then one staging turn
        if (!(XP_H2P_DBG & 4)) read_frags(rbuf, fr_);
        if (!(XP_H2P_DBG & 2)) stage(ra, rb, wbuf);
    };
    // the 24 MFMAs of the slab in registers, the global loads of slab t issued ahead of them.  Measured neutral and left out: s_setprio 1 here (79.6 us) or in
    // the memory phase (82.1), the loads interleaved one per two MFMAs (80.4), buffer loads with a scalar slab offset (93.2).
    auto compute_phase = [&](PRawA& ra, PRawB& rb, int t) {
        __builtin_amdgcn_sched_barrier(0);
        gload(ra, rb, t);
        mfmas(fr_);
        __builtin_amdgcn_sched_barrier(0);
    };
    if (grp == 0) {
        // turn j: compute slab 2j (loads of staging turn j + 2 first) | memory: fragments of slab 2j + 2, staging turn j + 1 (slab 2j + 3)
        int j = 0;
        for (; j + 1 < turns; j += 2) {
            compute_phase(ra0, rb0, st0 + 2 * (j + 2)); p_barrier(); memory_phase(ra1, rb1); p_barrier();
            compute_phase(ra1, rb1, st0 + 2 * (j + 3)); p_barrier(); memory_phase(ra0, rb0); p_barrier();
        }
        if (j < turns) { mfmas(fr_); p_barrier(); p_barrier(); }
    } else {
        // turn j: memory: fragments of slab 2j + 1, staging turn j + 1 (slab 2j + 2) | compute slab 2j + 1 (loads of staging turn j + 3 first)
        int j = 0;
        for (; j + 1 < turns; j += 2) {
            memory_phase(ra1, rb1); p_barrier(); compute_phase(ra1, rb1, st0 + 2 * (j + 3)); p_barrier();
            memory_phase(ra0, rb0); p_barrier(); compute_phase(ra0, rb0, st0 + 2 * (j + 4)); p_barrier();
        }
        if (j < turns) { read_frags(rbuf, fr_); p_barrier(); mfmas(fr_); p_barrier(); }
    }

    // exchange: group 0 finishes block row 0 of the pair's 64 x 64 sub-tile, group 1 block row 1
    float* xl = reinterpret_cast<float*>(lds_h2p) + w4 * 4096;          // 16 KB per wave pair: [sender group][j][r][lane]
    f32x16 keep[1][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            xl[grp * 2048 + (j * 16 + r) * 64 + lane] = grp ? acc[0][j][r] : acc[1][j][r];
            keep[0][j][r] = grp ? acc[1][j][r] : acc[0][j][r];
        }
    p_barrier();
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) keep[0][j][r] += xl[(grp ^ 1) * 2048 + (j * 16 + r) * 64 + lane];
    // the leading dimensions pass through an opaque scalar move: otherwise the epilogue's element offsets are computed before the K loop and spilled across it
    GemmParams pe = p;
    asm volatile("" : "+s"(pe.ldc), "+s"(pe.ldres));
    gemm_epilogue<PEpiTile, 1, 2, true>(pe, m0, n0, keep);
}

}  // namespace

// DEFAULT for K >= 768 and N >= 384 (XP_H2P=0 turns it off, XP_H2P=2 widens it to K >= 128 and N >= 96).  Measured, round 3 (tools/h2p_dbg.sh,
// profiles/r3_gemm_h2p_pingpong.txt): alone on the GPU it beats the tile kernel where the K loop dominates (M 19200 N 384 K 1536: 79.4 vs 91.6-99.7 us,
// 285 TF/s; M 4800 N 768 K 3072: 86.9 vs 102.4) and loses where prologue and epilogue dominate (K = 384: 32.3 vs 28.9, fc1 + GELU 131 vs 123: one workgroup
// per CU has nothing to hide them behind) or the tile count is small (M 4800 N 200: 25 vs 20 us).  In the pair step: +0.9 .. 1.2 %.
// The summation order differs from the tile kernel (even slabs + odd slabs), so the choice must NOT depend on M (= images x L, a batch quantity): round 3's
// predicate had a tile-count term and a pair run alone no longer agreed bit for bit with the same pair inside a batch of 8.  The predicate
// is now a property of the LAYER (K, N, lda) only: every batch size takes the same kernel for the same layer (tests/test_gpu_model.py::test_batch_invariance_480x640).
bool xp_gemm_h2p_applies(const GemmParams& p) {
    static const int on = getenv("XP_H2P") ? atoi(getenv("XP_H2P")) : 1;
    return on && p.mode == 0 && p.K % 64 == 0 && p.K >= (on > 1 ? 128 : 768) && p.lda % 4 == 0 && p.N >= (on > 1 ? 96 : 384) &&
           (int64_t)p.M * p.lda < (1ll << 30) && (int64_t)p.N * H2_SLAB_UNITS * 16 * (p.K / H2_BK) < (1ll << 32);
}

int xp_gemm_h2p_launch(const GemmParams& p, hipStream_t s) {
    static XpPerDeviceOnce attr_once;
    if (attr_once.need()) {
        XP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h2p_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)P_LDS));
    }
    const int grid = xp_cdiv(p.M, P_BM) * xp_cdiv(p.N, P_BN);
    static const bool by_shape = getenv("XP_PROF_SHAPES") != nullptr;
    std::string tag = "gemm_h2p_mfma_128x128";
    if (by_shape) tag += "_M" + std::to_string(p.M) + "_N" + std::to_string(p.N) + "_K" + std::to_string(p.K) + (p.act == 1 ? "_gelu" : "");
    XpProfScope prof(tag.c_str(), s, 2.0 * p.M * p.N * p.K, 4.0 * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N * (p.res ? 2 : 1)));
    hipLaunchKernelGGL(gemm_h2p_kernel, dim3(grid), dim3(512), P_LDS, s, p);
    XP_LAUNCH_CHECK();
    if (XP_H2P_DBG & 64) {      // debug builds only: synchronises
        static int printed = 0;
        static unsigned long long h[8][128];
        if (printed < 3 && p.K >= 1024 && hipDeviceSynchronize() == hipSuccess && hipMemcpyFromSymbol(h, HIP_SYMBOL(g_h2p_stamps), sizeof(h)) == hipSuccess) {
            ++printed;
            // stamps 2k / 2k+1 = entry / exit of barrier k; barriers 0, 1 belong to the prologue, then two per turn
            for (int w = 0; w < 8; w += 4) {
                double work[2] = {0, 0}, wait[2] = {0, 0}; int cnt = 0;
                for (int k = 10; k + 2 < 40; k += 2, ++cnt)
                    for (int h2 = 0; h2 < 2; ++h2) { work[h2] += (double)(h[w][2 * (k + h2)] - h[w][2 * (k + h2) - 1]); wait[h2] += (double)(h[w][2 * (k + h2) + 1] - h[w][2 * (k + h2)]); }
                fprintf(stderr, "[h2p stamps] M %d N %d K %d wave %d: first phase of a turn: work %.0f + barrier wait %.0f; second phase: work %.0f + wait %.0f (s_memtime ticks, mean of %d turns)\n",
                        p.M, p.N, p.K, w, work[0] / cnt, wait[0] / cnt, work[1] / cnt, wait[1] / cnt, cnt);
            }
        }
    }
    return XP_OK;
}
