// Dense descriptor matching: all-pairs L2 distance + nearest neighbour in both directions + mutual
// (cross-check) test.  Replaces cv2.BFMatcher(NORM_L2, crossCheck=True).match as called from
// reference xpoint/utils/matching.py:4-36 (and the in-repo NNMatcher, matching.py:38-75).
//
// CDNA4 mapping (round 2).  The index result must be that of EXACT arithmetic, so the matrix pipe only has to NOMINATE:
// every (q, t) whose approximate score is within a proven error bound of its row (column) optimum becomes a candidate,
// and a second kernel re-evaluates the candidates in direct form with fp64 accumulation (first index wins ties).
// That freedom is spent on the cheapest matrix arithmetic whose error can still be bounded: ONE fp16 product per
// multiply (v_mfma_f32_32x32x16_f16, exact products, f32 accumulate) on descriptors rounded once to fp16.
//   score e(q,t) = a.b - |a|^2/2 - |b|^2/2  (= -d^2/2; maximised).  The two norm terms ride in the contraction itself: every
//   descriptor image carries one extra 16-wide k slab, [ -|a|^2/2 as hi+lo fp16, 1, 1, 0.. ] on the query side and
//   [ 1, 1, -|b|^2/2 as hi+lo, 0.. ] on the target side, so the accumulator IS the score and the epilogue is a max.
//   Descriptors are pre-multiplied by one power of two S per call (largest norm in [1, 2): exact, keeps fp16 in range).
//   Rows past a pair's count are written as zero descriptors with score -30000: they never win and never qualify, so
//   the kernels carry no bounds masks at all.
// Two passes of the same Gram loop (the matrix work is cheap now: 2 x 1 product instead of 3 bf16 products + a heavy
// epilogue): pass 1 leaves the approximate row / column maxima (running row maxima live in registers across the whole
// column range of a workgroup, one atomicMax per row and workgroup; column maxima by one coalesced atomicMax per 32
// columns and wave); pass 2 recomputes the tile and nominates everything within the error bound of those FINAL maxima.
// A workgroup = 4 waves x 64 query rows; its A operand (64 x 272 fp16 per wave) stays in REGISTERS for the whole
// kernel, so LDS only carries the target tiles (64 descriptors, LDS-DMA, double buffered, row stride 560 B = conflict-
// free ds_read_b128), each fragment read feeding two MFMAs.  Grid = (column splits, 256-row strips, pairs), sized by
// the capacity; strips / splits past the device-side counts exit at once.
//
// Error bound (scaled units, |a^|, |b^| <= 2).  fp16 has an 11-bit significand: round to nearest commits u = 2^-11 relative per operand
// (absolute 2^-25 in the subnormal range), so |a~.b~ - a.b| <= (2u + u^2) sum|a_k||b_k| <= 2^-11 (|a^|^2 + |b^|^2) + 2e-6 (2|a||b| <= |a|^2 +
// |b|^2); norm terms as hi + lo: 2^-22 relative; f32 accumulation over 272 terms <= 8.5e-6 (|a^|^2 + |b^|^2).  One score is therefore off by at
// most err(q,t) = (2^-11 + 8.5e-6)(|a^_q|^2 + |b^_t|^2) + 2e-6 ~= 4.97e-4 (..) + 2e-6, and the TRUE optimum t* of row q against the approximate
// one t~ satisfies e~(q,t*) >= rowmax~ - err(q,t*) - err(q,t~) >= rowmax~ - E_q with
//     E_q = 1.0e-3 (|a^_q|^2 + M) + 4e-6,   M = the largest scaled squared norm of the call
// (round 2 had derived this with u = 2^-12 and used half the window: the index result was then not proven exact for sparse / low-D descriptors,
// ADVICE r2).  Only the number of candidates depends on the bound (measured on 480x640 pairs: ~1.5 per row), never the result.
#include "xp_common.h"
#include "../../include/xpoint_hip.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* mt_lds_ptr_t;

#ifndef XP_MT_DBG
#define XP_MT_DBG 0   /* timing experiments only (wrong results): 1 no MFMA, 2 no target-tile DMA after the first, 4 no epilogue, 8 no barrier, 16 no fragment reads */
#endif

namespace {

constexpr int CAND_CAP = 64;                // inline candidates per row / column (one per lane of the refining wave); longer lists -> overflow list -> match_overflow_kernel
constexpr float MT_EPS_REL = 1.0e-3f, MT_EPS_ABS = 4e-6f;    // 2 (2^-11 + 8.5e-6) = 9.94e-4, see the bound above
constexpr float MT_DEAD = -30000.f;          // score term of rows past the count (finite in fp16)
constexpr int MT_STRIP = 256, MT_TILE = 64;  // query rows per workgroup, target descriptors per LDS tile
constexpr int MT_QCAP = 1020;                // pass 2: entries of the workgroup's LDS hit queue (4 KB with its count word)

__host__ __device__ constexpr int mt_rowb(int ksd) { return (ksd * 16 + 16) * 2 + 16; }   // fp16 image row: D values, ext slab, 16 B pad

__device__ __forceinline__ int count_of(const int* p, int idx, int cap) { int n = p ? p[idx] : cap; return n < cap ? n : cap; }
// order-preserving float <-> uint map (atomicMax on scores of either sign)
__device__ __forceinline__ unsigned mt_ord(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float mt_unord(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }
__device__ __forceinline__ unsigned long long pack_key(float d2, int idx) {
    return ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned int)idx;
}
__device__ __forceinline__ float mt_max(float a, float b) { float d; asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float mt_max3(float a, float b, float c) { float d; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }

// (a kernel, not hipMemsetAsync: see nms_reset_counters_kernel in postproc.hip)
__global__ void match_reset_kernel(unsigned* __restrict__ maxbits) { if (threadIdx.x < 8) maxbits[threadIdx.x] = 0u; }

// ---- pass 0a: squared norms of both descriptor sets and the largest one of the call ----
__global__ __launch_bounds__(256) void match_norms_kernel(const float* __restrict__ d1, const float* __restrict__ d2, const int* __restrict__ cnt,
                                                          int cnt_stride, int which1, int which2, int cap1, int cap2, int D,
                                                          float* __restrict__ na, float* __restrict__ nb, unsigned* __restrict__ maxbits) {
    __shared__ float s_m[4];
    const int pair = blockIdx.y, side = blockIdx.z;
    const int cap = side ? cap2 : cap1;
    const int n = count_of(cnt, pair * cnt_stride + (side ? which2 : which1), cap);
    constexpr int NG = 4;                                          // row groups of four per wave: 16 independent 16-byte loads in flight per lane (was 4:
                                                                   // 2.0 TB/s); every row's sum is formed exactly as before (lane partial in c order, wave sum)
    if (blockIdx.x * (16 * NG) >= n) return;
    const int lane = threadIdx.x & 63;
    const int i0 = blockIdx.x * (16 * NG) + (threadIdx.x >> 6) * (4 * NG);
    float acc4[4 * NG];
#pragma unroll
    for (int j = 0; j < 4 * NG; ++j) acc4[j] = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        float4 v[4 * NG];
#pragma unroll
        for (int j = 0; j < 4 * NG; ++j) {
            const int i = i0 + j < n ? i0 + j : n - 1;
            v[j] = *reinterpret_cast<const float4*>((side ? d2 : d1) + ((int64_t)pair * cap + i) * D + c);
        }
#pragma unroll
        for (int j = 0; j < 4 * NG; ++j) { acc4[j] = fmaf(v[j].x, v[j].x, acc4[j]); acc4[j] = fmaf(v[j].y, v[j].y, acc4[j]); acc4[j] = fmaf(v[j].z, v[j].z, acc4[j]); acc4[j] = fmaf(v[j].w, v[j].w, acc4[j]); }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4 * NG; ++j) {
        const float t = xp_wave_sum(acc4[j]);
        if (i0 + j < n) { if (lane == 0) (side ? nb : na)[(int64_t)pair * cap + i0 + j] = t; s = fmaxf(s, t); }
    }
    if (lane == 0) s_m[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        // one atomic per workgroup at most, and only when it can raise the maximum: tens of thousands of atomics on ONE word
        // serialise (0.7 ms for 65 k rows); the relaxed agent-scope load may lag, which only costs a redundant atomic
        const float m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
        const unsigned cur = __hip_atomic_load(maxbits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__float_as_uint(m) > cur) atomicMax(maxbits, __float_as_uint(m));          // m >= 0: uint order = float order
    }
}

// ---- pass 0b: fp16 images (scaled by the call's power of two), extension slab, dead rows; resets keys and counters ----
template <int KSD>
__global__ __launch_bounds__(256) void match_convert_kernel(const float* __restrict__ d1, const float* __restrict__ d2, const int* __restrict__ cnt,
                                                            int cnt_stride, int which1, int which2, int cap1, int cap2, int rows1P, int rows2P, int D,
                                                            const float* __restrict__ na, const float* __restrict__ nb, const unsigned* __restrict__ maxbits,
                                                            unsigned char* __restrict__ imgA, unsigned char* __restrict__ imgB,
                                                            unsigned* __restrict__ rowkey, unsigned* __restrict__ colkey, int* __restrict__ rcnt, int* __restrict__ ccnt,
                                                            unsigned long long* __restrict__ rowkey2) {
    constexpr int ROWB = mt_rowb(KSD);
    const int pair = blockIdx.y, side = blockIdx.z;
    const int cap = side ? cap2 : cap1, rowsP = side ? rows2P : rows1P;
    const int n = count_of(cnt, pair * cnt_stride + (side ? which2 : which1), cap);
    const int pad = side ? MT_TILE : MT_STRIP;
    const int nP = (n + pad - 1) / pad * pad;            // rows the Gram kernels may touch
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= nP || i >= rowsP) return;
    const int lane = threadIdx.x & 63;
    // S = 2^-floor(log2(sqrt(max norm^2))): the largest scaled norm lies in [1, 2)
    const float mx = __uint_as_float(*maxbits);
    int ex = 0;
    if (mx > 0.f) { (void)frexpf(sqrtf(mx), &ex); ex -= 1; }
    const float S = ldexpf(1.f, -ex), S2 = S * S;
    unsigned char* row = (side ? imgB : imgA) + ((int64_t)pair * rowsP + i) * ROWB;
    const bool live = i < n;
    const float* r = (side ? d2 : d1) + ((int64_t)pair * cap + (live ? i : 0)) * D;
    for (int c = lane * 4; c < KSD * 16; c += 256) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live && c < D) v = *reinterpret_cast<const float4*>(r + c);
        union { _Float16 h[4]; uint2 u; } o;
        o.h[0] = (_Float16)(v.x * S); o.h[1] = (_Float16)(v.y * S); o.h[2] = (_Float16)(v.z * S); o.h[3] = (_Float16)(v.w * S);
        *reinterpret_cast<uint2*>(row + c * 2) = o.u;
    }
    if (lane == 0) {
        const float hn = live ? -0.5f * S2 * (side ? nb : na)[(int64_t)pair * cap + i] : MT_DEAD;
        const _Float16 hi = (_Float16)hn, lo = (_Float16)(hn - (float)hi), one = (_Float16)1.f, z = (_Float16)0.f;
        union { _Float16 h[24]; uint4 u[3]; } e;
        for (int k = 0; k < 24; ++k) e.h[k] = z;
        if (side == 0) { e.h[0] = hi; e.h[1] = lo; e.h[2] = one; e.h[3] = one; }
        else           { e.h[0] = one; e.h[1] = one; e.h[2] = hi; e.h[3] = lo; }
        uint4* dst = reinterpret_cast<uint4*>(row + KSD * 32);
        dst[0] = e.u[0]; dst[1] = e.u[1]; dst[2] = e.u[2];
        if (side == 0) { rowkey[(int64_t)pair * rows1P + i] = 0u; rowkey2[(int64_t)pair * rows1P + i] = 0ull; if (i < cap1) rcnt[(int64_t)pair * cap1 + i] = 0; }
        else           { colkey[(int64_t)pair * rows2P + i] = 0u; if (i < cap2) ccnt[(int64_t)pair * cap2 + i] = 0; }
    }
}

struct GramParams {
    const unsigned char* A; const unsigned char* B;      // fp16 images [pair][rowsP][ROWB]
    int rows1P, rows2P;
    const int* counts; int cnt_stride, which1, which2, cap1, cap2;
    unsigned* rowkey; unsigned* colkey;                   // ordered-uint approximate maxima, [pair][rowsP]
    const float* na; const float* nb; const unsigned* maxbits;
    int* rcnt; int* ccnt; int* rcand; int* ccand;        // candidate lists (pairs, cap[, CAND_CAP])
    int csplit;
    // knn (xp_match_knn2): rows only, thresholds hang from the SECOND largest approximate score of the row
    unsigned long long* rowkey2;                          // [pair][rows1P]: ordered-uint (largest << 32 | second largest)
    int knn;
    // threshold matcher (xp_match_threshold, PASS 3): nominate a.b > c0 - E
    float thr_c0; int* hit_count; int2* hits; int hit_cap;
};

// ---- passes 1 and 2: the Gram loop ----
// PASS 1: approximate row / column maxima (TOP2: the two largest per row, rows only — one wave per SIMD, 32 more resident registers);
// PASS 2: nomination against them; PASS 3: nomination against a fixed inner-product threshold (ThresholdMatcher).
__device__ __forceinline__ float mt_med3(float a, float b, float c) { float d; asm("v_med3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }

template <int KSD, int PASS, bool TOP2 = false>
__global__ __launch_bounds__(256, TOP2 ? 1 : 2) void match_gram_kernel(GramParams p) {
    constexpr int KS = KSD + 1, ROWB = mt_rowb(KSD), TILEB = MT_TILE * ROWB, NPIECE = TILEB / 1024;
    static_assert(TILEB % 1024 == 0, "a tile must be whole LDS-DMA pieces");
    extern __shared__ __align__(16) unsigned char lds[];          // two target tiles | pass 2: hit queue (count word + MT_QCAP entries)
    unsigned* const q_count = reinterpret_cast<unsigned*>(lds + 2 * TILEB);
    unsigned* const q_entry = q_count + 4;
    float* const row_thr = reinterpret_cast<float*>(lds + 2 * TILEB + 16 + 4 * MT_QCAP);     // pass 2: [MT_STRIP]
    if (PASS >= 2 && threadIdx.x == 0) *q_count = 0u;             // visible to everybody after the first tile's barrier
    const int pair = blockIdx.z, strip = blockIdx.y, split = blockIdx.x;
    const int n1 = count_of(p.counts, pair * p.cnt_stride + p.which1, p.cap1);
    const int n2 = count_of(p.counts, pair * p.cnt_stride + p.which2, p.cap2);
    const int r0 = strip * MT_STRIP;
    if (r0 >= n1 || n2 <= 0) return;
    const int per = ((n2 + p.csplit - 1) / p.csplit + MT_TILE - 1) / MT_TILE * MT_TILE;
    const int c_begin = split * per;
    if (c_begin >= n2) return;
    const int c_end = min(c_begin + per, (n2 + MT_TILE - 1) / MT_TILE * MT_TILE);
    const int ntiles = (c_end - c_begin) / MT_TILE;
    const int lane = threadIdx.x & 63, fr = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr0 = r0 + wave * 64;                               // this wave's first query row

    const unsigned char* Bb = p.B + ((int64_t)pair * p.rows2P + c_begin) * ROWB;
    auto issue_tile = [&](int t, int buf) {
        const unsigned char* src = Bb + (int64_t)t * TILEB + lane * 16;
        unsigned char* dst = lds + buf * TILEB;
#pragma unroll
        for (int i = 0; i < (NPIECE + 3) / 4; ++i) {
            const int piece = wave + 4 * i;
            if (piece < NPIECE) __builtin_amdgcn_global_load_lds(src + piece * 1024, (mt_lds_ptr_t)(dst + piece * 1024), 16, 0, 0);
        }
    };
    issue_tile(0, 0);

    // the wave's 64 query rows as MFMA A fragments, resident for the whole kernel
    f16x8 af[KS][2];
    {
        const unsigned char* Ab = p.A + ((int64_t)pair * p.rows1P + wr0 + fr) * ROWB + h * 16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) af[ks][rb] = *reinterpret_cast<const f16x8*>(Ab + (int64_t)rb * 32 * ROWB + ks * 32);
    }
    // row of register r of row block rb (wave-relative): 32 rb + (r & 3) + 8 (r >> 2) + 4 h
    float rowv[2][16];     // pass 1 only: running row maxima over this workgroup's columns
    float rowv2[TOP2 ? 2 : 1][TOP2 ? 16 : 1];     // TOP2: running second largest
    float Mx = 0.f, S2 = 1.f;
    if (PASS == 1) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { rowv[rb][r] = -INFINITY; if (TOP2) rowv2[rb][r] = -INFINITY; }
    } else {
        // pass 2: the 256 row thresholds of the strip live in LDS (a broadcast ds_read_b128 per four accumulator registers): as
        // 32 more resident registers they pushed the kernel past the 256 of two waves per SIMD
        const float mxn = __uint_as_float(*p.maxbits);
        int ex = 0;
        if (mxn > 0.f) { (void)frexpf(sqrtf(mxn), &ex); ex -= 1; }
        S2 = ldexpf(1.f, -2 * ex);
        Mx = mxn * S2;
        const int row = r0 + (int)threadIdx.x;
        const float nas = p.na[(int64_t)pair * p.cap1 + (row < n1 ? row : n1 - 1)] * S2;
        const float Eq = MT_EPS_REL * (nas + Mx) + MT_EPS_ABS;
        float thr;
        if (PASS == 3) thr = p.thr_c0 * S2 - 0.5f * nas - Eq;             // a.b > c0  <=>  score + |b|^2 / 2 > c0 - |a|^2 / 2 (scaled units); the column term rides in cthr
        else if (p.knn) {                                                  // hang the window from the row's SECOND largest approximate score; a row with
            const unsigned lo = (unsigned)(p.rowkey2[(int64_t)pair * p.rows1P + row] & 0xffffffffull);      // one live target nominates every live one
            thr = lo ? fmaxf(mt_unord(lo) - Eq, -1000.f) : -1000.f;        // (live scores >= -8, dead ones <= -29000: with one live target the second
                                                                           //  largest IS a dead column's score and must not pull the dead ones in)
        } else thr = mt_unord(p.rowkey[(int64_t)pair * p.rows1P + row]) - Eq;
        row_thr[threadIdx.x] = row < n1 ? thr : INFINITY;                 // rows past the count never nominate
    }
    unsigned* ck = p.colkey + (int64_t)pair * p.rows2P + c_begin;
    int* rcnt = p.rcnt + (int64_t)pair * p.cap1; int* rcand = p.rcand + (int64_t)pair * p.cap1 * CAND_CAP;
    int* ccnt = p.ccnt + (int64_t)pair * p.cap2; int* ccand = p.ccand + (int64_t)pair * p.cap2 * CAND_CAP;

    // column thresholds (pass 2) of a tile's two 32-column blocks; loaded one tile ahead and BEFORE that tile's DMA is issued, so
    // that the wait for them never includes the DMA behind them (vmcnt retires in order)
    auto col_thresholds = [&](int t, float (&out)[2]) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int cl = t * MT_TILE + cb * 32 + fr, col = c_begin + cl;
            const int cc = col < n2 ? col : n2 - 1;               // unconditional loads, clamped (see the row thresholds)
            const float nbs = p.nb[(int64_t)pair * p.cap2 + cc] * S2;
            const float thr = PASS == 3 ? -0.5f * nbs : mt_unord(ck[cc - c_begin]) - (MT_EPS_REL * (nbs + Mx) + MT_EPS_ABS);
            out[cb] = (col < n2 && !(PASS == 2 && p.knn)) ? thr : INFINITY;
        }
    };
    float cthr[2] = {INFINITY, INFINITY}, cnext[2] = {INFINITY, INFINITY};
    if (PASS >= 2) col_thresholds(0, cnext);

    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        // This wave's pieces of tile t have landed.  Pass 1 leaves its two column-maximum atomics of the previous tile in flight
        // (issued after the DMA, so "all but the 2 youngest" covers the DMA): an atomic stays counted for thousands of cycles.
        if (PASS == 1 && t > 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(XP_MT_DBG & 8)) __builtin_amdgcn_s_barrier();       // ... and everybody's; everybody is done reading tile t - 1
        if (PASS >= 2) { cthr[0] = cnext[0]; cthr[1] = cnext[1]; col_thresholds(t + 1, cnext); }
        if (t + 1 < ntiles && !(XP_MT_DBG & 2)) issue_tile(t + 1, buf ^ 1);
        const unsigned char* tb = lds + buf * TILEB + fr * ROWB + h * 16;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            f32x16_t acc[2];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
            // target fragments are read two k-steps ahead of the MFMAs that consume them (LDS latency behind the matrix pipe)
            const unsigned char* tc = tb + cb * 32 * ROWB;
            f16x8 bq[2] = {*reinterpret_cast<const f16x8*>(tc), *reinterpret_cast<const f16x8*>(tc + 32)};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const f16x8 bf = bq[ks & 1];
                if (ks + 2 < KS && !(XP_MT_DBG & 16)) bq[ks & 1] = *reinterpret_cast<const f16x8*>(tc + (ks + 2) * 32);
                if (XP_MT_DBG & 1) { asm volatile("" :: "v"(bf), "v"(af[ks][0]), "v"(af[ks][1])); continue; }
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks][0], bf, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks][1], bf, acc[1], 0, 0, 0);
            }
            if (XP_MT_DBG & 4) { asm volatile("" :: "v"(acc[0]), "v"(acc[1])); continue; }
            // The epilogue reads the accumulators from inline asm (v_max_f32 / v_max3_f32 without the canonicalising moves hipcc puts in
            // front of fmaxf on MFMA results).  hipcc pads no hazards for an asm statement, so the MFMA -> VALU-read wait states
            // (up to 18 for a 16-pass XDL write) are spent here, once per block, in a statement that owns both accumulators.
            asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0]), "+v"(acc[1]));
            if (PASS == 1 && TOP2) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        rowv2[rb][r] = mt_med3(rowv[rb][r], rowv2[rb][r], acc[rb][r]);      // rowv2 <= rowv: the median of (largest, second, new) is the new second
                        rowv[rb][r] = mt_max(rowv[rb][r], acc[rb][r]);
                    }
            } else if (PASS == 1) {
                float cm = acc[0][0];
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        rowv[rb][r] = mt_max(rowv[rb][r], acc[rb][r]);
                        rowv[rb][r + 1] = mt_max(rowv[rb][r + 1], acc[rb][r + 1]);
                        cm = mt_max3(cm, acc[rb][r], acc[rb][r + 1]);
                    }
                cm = mt_max(cm, __shfl_xor(cm, 32, 64));
                if (h == 0) atomicMax(&ck[t * MT_TILE + cb * 32 + fr], mt_ord(cm));
            } else {
                // Nomination: one compare pair per element, merged into a wave-wide scalar mask; only when some lane hits (about one
                // element in 3000) do the hit lanes push (row, column, kind) into the workgroup's LDS queue.  The global candidate
                // lists are appended to after the column range, all entries in parallel: a returning global atomic per hit inside
                // this loop would stall the wave for microseconds each time.
                const int cl = t * MT_TILE + cb * 32 + fr;                   // column, relative to c_begin
                // One bit per accumulator register and lane, branch-free and without lane masks in scalar registers (64 compare masks
                // overflow them and get spilled lane by lane; a scalar branch per register drains the instruction buffer 32 times per
                // block): acc >= thr  <=>  sign bit of (acc - thr) clear, and v_alignbit shifts that sign bit into the mask.
                // Element i = 16 rb + r ends up, inverted, in bit 31 - i.
                unsigned rneg = 0u, cneg = PASS == 3 ? ~0u : 0u;
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const float4 rt = *reinterpret_cast<const float4*>(row_thr + wave * 64 + 32 * rb + 8 * g4 + 4 * h);   // rows of registers 4 g4 .. + 3
                        const float rtv[4] = {rt.x, rt.y, rt.z, rt.w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float v = acc[rb][4 * g4 + q];
                            if (PASS == 3) rneg = __builtin_amdgcn_alignbit(rneg, __float_as_uint((v - cthr[cb]) - rtv[q]), 31);
                            else {
                                rneg = __builtin_amdgcn_alignbit(rneg, __float_as_uint(v - rtv[q]), 31);
                                cneg = __builtin_amdgcn_alignbit(cneg, __float_as_uint(v - cthr[cb]), 31);
                            }
                        }
                    }
                const unsigned rbits = ~rneg, cbits = ~cneg;
                if (__builtin_amdgcn_ballot_w64((rbits | cbits) != 0u)) {          // about 70 % of the blocks hold one or two hits
                    unsigned bits = rbits | cbits;
                    while (bits) {
                        const int bp = __ffs(bits) - 1;
                        bits &= bits - 1u;
                        const bool hr = (rbits >> bp) & 1u, hc = (cbits >> bp) & 1u;
                        const int b = 31 - bp;                               // accumulator register 16 rb + r
                        const unsigned rl = wave * 64 + 32 * (b >> 4) + (b & 3) + 8 * ((b & 15) >> 2) + 4 * h;     // row, relative to r0
                        const unsigned e = (unsigned)cl | (rl << 16) | (hr ? 0x40000000u : 0u) | (hc ? 0x80000000u : 0u);
                        const unsigned pos = atomicAdd(q_count, 1u);
                        if (pos < MT_QCAP) q_entry[pos] = e;
                        else if (PASS == 3) {
                            const int k = atomicAdd(p.hit_count, 1);
                            if (k < p.hit_cap) p.hits[k] = make_int2(pair, ((r0 + (int)rl) << 16) | (c_begin + cl));
                        } else {     // queue full (heavily clustered descriptors): append to the global lists from here — slow (a returning global
                                     // atomic per hit) but complete: round 3 flagged the row as overflowed instead, which sent it to a full scan
                            if (hr) { const int row = r0 + (int)rl; const int k = atomicAdd(&rcnt[row], 1); if (k < CAND_CAP) rcand[(int64_t)row * CAND_CAP + k] = c_begin + cl; }
                            if (hc) { const int col = c_begin + cl; const int k = atomicAdd(&ccnt[col], 1); if (k < CAND_CAP) ccand[(int64_t)col * CAND_CAP + k] = r0 + (int)rl; }
                        }
                    }
                }
            }
        }
    }
    if (PASS >= 2) {
        __syncthreads();
        const unsigned nq = min(*q_count, (unsigned)MT_QCAP);
        for (unsigned i = threadIdx.x; i < nq; i += 256) {
            const unsigned e = q_entry[i];
            const int row = r0 + (int)((e >> 16) & 0x3fffu), col = c_begin + (int)(e & 0xffffu);
            if (PASS == 3) { const int k = atomicAdd(p.hit_count, 1); if (k < p.hit_cap) p.hits[k] = make_int2(pair, (row << 16) | col); continue; }
            if (e & 0x40000000u) { const int k = atomicAdd(&rcnt[row], 1); if (k < CAND_CAP) rcand[(int64_t)row * CAND_CAP + k] = col; }
            if (e & 0x80000000u) { const int k = atomicAdd(&ccnt[col], 1); if (k < CAND_CAP) ccand[(int64_t)col * CAND_CAP + k] = row; }
        }
    }
    if (PASS == 1 && TOP2) {
        // the two largest of every row over the 32 lanes (columns) of its half, then one 64-bit compare-and-swap merge per row and workgroup
        unsigned long long* rk2 = p.rowkey2 + (int64_t)pair * p.rows1P + wr0;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float m1 = rowv[rb][r], m2 = rowv2[rb][r];
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) {
                    const float o1 = __shfl_xor(m1, o, 64), o2 = __shfl_xor(m2, o, 64);
                    m2 = fmaxf(fminf(m1, o1), fmaxf(m2, o2)); m1 = fmaxf(m1, o1);
                }
                if (fr == 0) {
                    unsigned long long* w = &rk2[32 * rb + (r & 3) + 8 * (r >> 2) + 4 * h];
                    const unsigned a1 = mt_ord(m1), a2 = mt_ord(m2);           // -inf maps below every score, 0 = "none yet"
                    unsigned long long old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    for (;;) {
                        const unsigned b1 = (unsigned)(old >> 32), b2 = (unsigned)old;
                        const unsigned n1v = a1 > b1 ? a1 : b1, lo = a1 > b1 ? b1 : a1, n2v = max(lo, max(a2, b2));
                        const unsigned long long nw = ((unsigned long long)n1v << 32) | n2v;
                        if (nw == old) break;
                        const unsigned long long prev = atomicCAS(w, old, nw);
                        if (prev == old) break;
                        old = prev;
                    }
                }
            }
    } else if (PASS == 1) {
        // row maxima: reduce each register over the 32 lanes (columns) of its half, one atomicMax per row and workgroup
        unsigned* rk = p.rowkey + (int64_t)pair * p.rows1P + wr0;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = xp_row16_max(rowv[rb][r]);
                v = mt_max(v, __shfl_xor(v, 16, 64));
                if (fr == 0) atomicMax(&rk[32 * rb + (r & 3) + 8 * (r >> 2) + 4 * h], mt_ord(v));
            }
    }
}

// ---- refinement: exact nearest neighbour among the nominated candidates ------------------------------------------------------------------------
// Round 4.  Round 3 re-evaluated every candidate in fp64 and, when a list overflowed (CAND_CAP 16), let ONE wave scan every target
// in fp64: 2.5 us per target — a single overflowing row of a 4060-keypoint pair cost 10 ms (0.30 -> 10.7 ms per 8 pairs on clustered descriptors).
// Now three levels, each cheaper per element than the next:
//   level 1  fp16 one-product Gram nomination (above): window 2e-3 of the score;
//   level 2  f32 DIRECT form sum (a_k - b_k)^2 of every listed candidate (lists up to CAND_CAP = 64: one candidate per lane of the refining wave, eight
//            target rows in flight): relative error <= 14 x 2^-24 = 8.4e-7 (one rounding per difference, square, fma and reduction step; all terms >= 0),
//            so only candidates with s <= s_min (1 + MT_F32_REL) + MT_F32_ABS can be the exact minimum (or tie with it) — a window 400 x narrower;
//   level 3  fp64 direct form of those (usually one: the winner, whose fp64 distance is the reported one), first index wins exact ties.
// Rows whose list overflowed are appended to an overflow list and finished by match_overflow_kernel: a WORKGROUP per row scans all targets with the
// same level 2 / 3 logic (running minimum per wave), rows in parallel across the chip.  The result is the exact-arithmetic argmin as before.
constexpr int MT_OVF_GRID = 512;               // persistent workgroups of the overflow pass (two per CU)
constexpr float MT_F32_REL = 1.0e-5f, MT_F32_ABS = 1.0e-30f;    // 2 x (8.4e-7 rounded up 5 x); the absolute term covers squares that underflow

// fp64 direct-form squared distance between the lane-distributed query (a4 = its float4 at column 4 lane, zeros past D) and target row b
__device__ __forceinline__ double mt_dist64(const float4& a4, const float* __restrict__ b, int lane, int D) {
    double s = 0.0;
    if (lane * 4 < D) {
        const float4 v = *reinterpret_cast<const float4*>(b + lane * 4);
        double df = (double)a4.x - (double)v.x; s = fma(df, df, s);
        df = (double)a4.y - (double)v.y; s = fma(df, df, s);
        df = (double)a4.z - (double)v.z; s = fma(df, df, s);
        df = (double)a4.w - (double)v.w; s = fma(df, df, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    return s;
}
__device__ __forceinline__ float mt_part32(const float4& a4, const float4& v) {
    float d = a4.x - v.x, s = d * d;
    d = a4.y - v.y; s = fmaf(d, d, s);
    d = a4.z - v.z; s = fmaf(d, d, s);
    d = a4.w - v.w; s = fmaf(d, d, s);
    return s;
}

struct OvfList { int* count; int2* entry; };      // entry = (pair * 2 + direction, query row)

// One wave per query.  K2 (knn): the TWO nearest targets are returned (idx_out / dist_out hold 2 entries per query), see xp_match_knn.
template <bool K2>
__global__ __launch_bounds__(256) void match_refine_kernel(const float* __restrict__ dq, const float* __restrict__ dt, const int* __restrict__ nqp,
                                                           const int* __restrict__ ntp, int cnt_stride, int whichq, int whicht, int capq,
                                                           int capt, int D, const int* __restrict__ cnt, const int* __restrict__ cand,
                                                           int* __restrict__ idx_out, float* __restrict__ dist_out, OvfList ovf, int dir) {
    const int pair = blockIdx.y;
    const int nq = count_of(nqp, pair * cnt_stride + whichq, capq), nt = count_of(ntp, pair * cnt_stride + whicht, capt);
    const int q = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (q >= nq) return;
    const int lane = threadIdx.x & 63;
    const float* a = dq + ((int64_t)pair * capq + q) * D;
    const float* tb = dt + (int64_t)pair * capt * D;
    const int nc = cnt[(int64_t)pair * capq + q];
    if (nc > CAND_CAP) {
        if (lane == 0) { const int pos = atomicAdd(ovf.count, 1); ovf.entry[pos] = make_int2(pair * 2 + dir, q); }
        return;
    }
    const int* cl = cand + ((int64_t)pair * capq + q) * CAND_CAP;
    float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane * 4 < D) a4 = *reinterpret_cast<const float4*>(a + lane * 4);
    const int mine_t = lane < nc ? cl[lane] : 0;                   // lane c owns candidate c
    // level 2: f32 direct form, eight candidates in flight
    float mine = INFINITY;
    for (int c0 = 0; c0 < nc; c0 += 8) {
        float part[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = __builtin_amdgcn_readlane(mine_t, (c0 + j) & 63);      // lanes past nc hold 0: a valid row, its sum is discarded
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lane * 4 < D) v = *reinterpret_cast<const float4*>(tb + (int64_t)t * D + lane * 4);
            part[j] = mt_part32(a4, v);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float sum = xp_wave_sum(part[j]); if (lane == c0 + j) mine = sum; }
    }
    if (lane >= nc) mine = INFINITY;
    float m32 = -xp_wave_max(-mine);
    if (K2) {       // the window hangs from the SECOND smallest f32 sum (every candidate within it may be one of the exact two nearest)
        const unsigned long long at_min = __ballot(mine == m32);
        const int first = __ffsll((long long)at_min) - 1;
        const float second = -xp_wave_max(lane == first ? -INFINITY : -mine);
        m32 = second;       // INFINITY when there is a single candidate: everything passes
    }
    const float thr = m32 < INFINITY ? fmaf(m32, MT_F32_REL, m32) + MT_F32_ABS : INFINITY;
    unsigned long long pass = __ballot(lane < nc && mine <= thr);
    // level 3: fp64 on the survivors
    double best = INFINITY, best2 = INFINITY; int bi = -1, bi2 = -1;
    while (pass) {
        const int c = __ffsll((long long)pass) - 1;
        pass &= pass - 1ull;
        const int t = __builtin_amdgcn_readlane(mine_t, c);
        const double s = mt_dist64(a4, tb + (int64_t)t * D, lane, D);
        if (s < best || (s == best && t < bi)) { best2 = best; bi2 = bi; best = s; bi = t; }
        else if (K2 && (s < best2 || (s == best2 && t < bi2))) { best2 = s; bi2 = t; }
    }
    (void)nt;
    if (lane == 0) {
        if (K2) {
            int* io = idx_out + ((int64_t)pair * capq + q) * 2; float* dd = dist_out + ((int64_t)pair * capq + q) * 2;
            io[0] = bi; dd[0] = (float)sqrt(best); io[1] = bi2; dd[1] = (float)sqrt(best2);
        } else { idx_out[(int64_t)pair * capq + q] = bi; dist_out[(int64_t)pair * capq + q] = (float)sqrt(best); }
    }
}

// Rows whose candidate list overflowed: a workgroup (16 waves) per row walks every target — f32 direct form, sixteen... eight rows in flight per wave, a
// running minimum per wave; a target within the f32 window of the running minimum (a superset of the final window: the minimum only falls) is evaluated
// in fp64 on the spot (the branch is wave-uniform) — then the waves' (distance, index) pairs are reduced through LDS.  Persistent over the overflow
// list, so one fixed-size launch serves any number of rows (0 rows: every workgroup exits after one load).
template <bool K2>
__global__ __launch_bounds__(1024) void match_overflow_kernel(const float* __restrict__ d1, const float* __restrict__ d2, const int* __restrict__ counts,
                                                              int cnt_stride, int which1, int which2, int cap1, int cap2, int D, OvfList ovf,
                                                              int* __restrict__ idx12, float* __restrict__ dist12, int* __restrict__ idx21,
                                                              float* __restrict__ dist21) {
    __shared__ double s_best[2][16];
    __shared__ int s_idx[2][16];
    const int n_ovf = *ovf.count;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = blockIdx.x; e < n_ovf; e += gridDim.x) {
        const int2 ent = ovf.entry[e];
        const int pair = ent.x >> 1, dir = ent.x & 1, q = ent.y;
        const int capq = dir ? cap2 : cap1, capt = dir ? cap1 : cap2;
        const int nt = count_of(counts, pair * cnt_stride + (dir ? which1 : which2), capt);
        const float* a = (dir ? d2 : d1) + ((int64_t)pair * capq + q) * D;
        const float* tb = (dir ? d1 : d2) + (int64_t)pair * capt * D;
        float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane * 4 < D) a4 = *reinterpret_cast<const float4*>(a + lane * 4);
        float run1 = INFINITY, run2 = INFINITY;          // smallest (and, K2, second smallest) f32 sums this wave has seen
        double best = INFINITY, best2 = INFINITY; int bi = -1, bi2 = -1;
        for (int t0 = wave * 8; t0 < nt; t0 += 16 * 8) {
            float part[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int t = t0 + j < nt ? t0 + j : nt - 1;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (lane * 4 < D) v = *reinterpret_cast<const float4*>(tb + (int64_t)t * D + lane * 4);
                part[j] = mt_part32(a4, v);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float sum = xp_wave_sum(part[j]);
                const int t = t0 + j;
                if (t >= nt) continue;
                const float ref = K2 ? run2 : run1;
                if (sum < run1) { run2 = run1; run1 = sum; } else if (sum < run2) run2 = sum;
                if (!(sum <= fmaf(ref, MT_F32_REL, ref) + MT_F32_ABS) && ref < INFINITY) continue;       // wave-uniform
                const double s = mt_dist64(a4, tb + (int64_t)t * D, lane, D);
                if (s < best || (s == best && t < bi)) { best2 = best; bi2 = bi; best = s; bi = t; }
                else if (K2 && (s < best2 || (s == best2 && t < bi2))) { best2 = s; bi2 = t; }
            }
        }
        if (lane == 0) { s_best[0][wave] = best; s_idx[0][wave] = bi; s_best[1][wave] = best2; s_idx[1][wave] = bi2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double b1 = INFINITY, b2 = INFINITY; int i1 = -1, i2 = -1;
            for (int k = 0; k < 2; ++k)
                for (int w = 0; w < 16; ++w) {
                    const double s = s_best[k][w]; const int t = s_idx[k][w];
                    if (t < 0) continue;
                    if (s < b1 || (s == b1 && t < i1)) { b2 = b1; i2 = i1; b1 = s; i1 = t; }
                    else if (s < b2 || (s == b2 && t < i2)) { b2 = s; i2 = t; }
                }
            int* io = dir ? idx21 : idx12; float* dd = dir ? dist21 : dist12;
            if (K2) { io += ((int64_t)pair * capq + q) * 2; dd += ((int64_t)pair * capq + q) * 2; io[0] = i1; dd[0] = (float)sqrt(b1); io[1] = i2; dd[1] = (float)sqrt(b2); }
            else { io[(int64_t)pair * capq + q] = i1; dd[(int64_t)pair * capq + q] = (float)sqrt(b1); }
        }
        __syncthreads();
    }
}

// Candidate-list statistics of the latest call on a workspace (bench.py: match_candidates_per_row, match_overflow_rows): out[0] = sum of the list
// lengths of the live rows and columns, out[1] = the longest, out[2] = rows + columns that overflowed CAND_CAP, out[3] = live rows + columns.
__global__ __launch_bounds__(256) void match_stats_kernel(const int* __restrict__ rcnt, const int* __restrict__ ccnt, const int* __restrict__ counts,
                                                          int cnt_stride, int which1, int which2, int cap1, int cap2, int pairs,
                                                          unsigned long long* __restrict__ out) {
    const int64_t a = (int64_t)pairs * cap1, total = a + (int64_t)pairs * cap2;
    unsigned long long sum = 0, mx = 0, ov = 0, live = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const bool col = i >= a;
        const int64_t j = col ? i - a : i;
        const int cap = col ? cap2 : cap1, pair = (int)(j / cap), r = (int)(j % cap);
        if (r >= count_of(counts, pair * cnt_stride + (col ? which2 : which1), cap)) continue;
        const unsigned long long n = (unsigned long long)(col ? ccnt : rcnt)[j];
        sum += n; mx = n > mx ? n : mx; ov += n > (unsigned long long)CAND_CAP; live += 1;
    }
    atomicAdd(&out[0], sum); atomicMax(&out[1], mx); atomicAdd(&out[2], ov); atomicAdd(&out[3], live);
}

// Mutual test + ordered compaction (ascending queryIdx, like BFMatcher.match output order).
// mode 0 strict mutual NN: keep q iff idx21[idx12[q]] == q.
// mode 1 legacy cross-check (OpenCV <= 3.4.1 semantics): for every q the nearest t among
//        {t : idx21[t] == q} (strict <, ascending t).
__global__ __launch_bounds__(1024) void match_mutual_kernel(const int* __restrict__ idx12, const float* __restrict__ dist12,
                                                            const int* __restrict__ idx21, const float* __restrict__ dist21,
                                                            const int* __restrict__ n1p, const int* __restrict__ n2p, int cnt_stride,
                                                            int which1, int which2, int cap1, int cap2, int mode,
                                                            int* __restrict__ mq, int* __restrict__ mt, float* __restrict__ md,
                                                            int* __restrict__ mcount, unsigned long long* __restrict__ scratch) {
    __shared__ int s_wave[16];
    __shared__ int s_base;
    const int pair = blockIdx.x;
    const int n1 = count_of(n1p, pair * cnt_stride + which1, cap1), n2 = count_of(n2p, pair * cnt_stride + which2, cap2);
    const int* i12 = idx12 + (int64_t)pair * cap1; const int* i21 = idx21 + (int64_t)pair * cap2;
    const float* d12 = dist12 + (int64_t)pair * cap1; const float* d21 = dist21 + (int64_t)pair * cap2;
    unsigned long long* sk = scratch + (int64_t)pair * cap1;
    if (mode == 1) {
        for (int q = threadIdx.x; q < n1; q += 1024) sk[q] = ~0ull;
        __syncthreads();
        for (int t = threadIdx.x; t < n2; t += 1024) { const int q = i21[t]; if (q >= 0) atomicMin(&sk[q], pack_key(d21[t], t)); }
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    int* oq = mq + (int64_t)pair * cap1; int* ot = mt + (int64_t)pair * cap1; float* od = md + (int64_t)pair * cap1;
    for (int q0 = 0; q0 < n1; q0 += 1024) {
        const int q = q0 + threadIdx.x;
        bool hit = false; int t = -1; float d = 0.f;
        if (q < n1) {
            if (mode == 0) { t = i12[q]; hit = t >= 0 && i21[t] == q; d = d12[q]; }
            else { const unsigned long long k = sk[q]; hit = k != ~0ull; t = (int)(k & 0xffffffffu); d = __uint_as_float((unsigned int)(k >> 32)); }
        }
        const unsigned long long bal = __ballot(hit);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wave] = __popcll(bal);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; ++w) off += s_wave[w];
        if (hit) { oq[off + before] = q; ot[off + before] = t; od[off + before] = d; }
        __syncthreads();
        if (threadIdx.x == 0) { int s = 0; for (int w = 0; w < 16; ++w) s += s_wave[w]; s_base += s; }
        __syncthreads();
    }
    if (threadIdx.x == 0) mcount[pair] = s_base;
}

}  // namespace

static int mt_ksd(int D) { return D <= 64 ? 4 : (D <= 128 ? 8 : 16); }
static size_t mt_up(size_t n, size_t a) { return (n + a - 1) / a * a; }

extern "C" size_t xp_match_workspace_bytes(int pairs, int cap1, int cap2, int D) {
    const size_t a = (size_t)pairs * cap1, b = (size_t)pairs * cap2;
    const size_t aP = (size_t)pairs * mt_up(cap1, MT_STRIP), bP = (size_t)pairs * mt_up(cap2, MT_TILE);
    const size_t rowb = mt_rowb(mt_ksd(D));
    // rowkey, colkey (u32, padded rows), rowkey2 (u64) | scratch (u64) | na, nb | rcnt, ccnt | rcand, ccand | overflow list (int2) | max-norm / counter words | fp16 images
    return 4 * (aP + bP) + 8 * aP + 8 * a + 4 * (a + b) + 4 * (a + b) + 4 * CAND_CAP * (a + b) + 8 * (a + b) + 1024 + (aP + bP) * rowb;
}

// The carve-up of a workspace (shared by xp_match_mnn and xp_match_stats)
struct MatchWs {
    unsigned long long* scratch; unsigned long long* rowkey2; unsigned* rowkey; unsigned* colkey; float* na; float* nb; int* rcnt; int* ccnt; int* rcand; int* ccand;
    int2* ovf_entry; unsigned* words; unsigned char* A; unsigned char* B; int rows1P, rows2P;
};
static MatchWs mt_carve(void* workspace, int pairs, int cap1, int cap2, int D) {
    MatchWs m{};
    const size_t a = (size_t)pairs * cap1, b = (size_t)pairs * cap2;
    m.rows1P = (int)mt_up(cap1, MT_STRIP); m.rows2P = (int)mt_up(cap2, MT_TILE);
    const size_t aP = (size_t)pairs * m.rows1P, bP = (size_t)pairs * m.rows2P;
    char* w = (char*)workspace;
    m.scratch = (unsigned long long*)w; w += 8 * a;
    m.ovf_entry = (int2*)w; w += 8 * (a + b);
    m.rowkey2 = (unsigned long long*)w; w += 8 * aP;
    m.rowkey = (unsigned*)w; w += 4 * aP;
    m.colkey = (unsigned*)w; w += 4 * bP;
    m.na = (float*)w; w += 4 * a;
    m.nb = (float*)w; w += 4 * b;
    m.rcnt = (int*)w; w += 4 * a;
    m.ccnt = (int*)w; w += 4 * b;
    m.rcand = (int*)w; w += 4 * CAND_CAP * a;
    m.ccand = (int*)w; w += 4 * CAND_CAP * b;
    w = (char*)(((uintptr_t)w + 255) & ~(uintptr_t)255);
    m.words = (unsigned*)w; w += 256;          // [0] largest squared norm (bits), [2] overflow-list length, [8..15] statistics (xp_match_stats)
    m.A = (unsigned char*)w; w += aP * mt_rowb(mt_ksd(D));
    m.B = (unsigned char*)w;
    return m;
}

// kind 0: mutual NN (passes 1 + 2), 1: knn (pass 1 with the two largest per row + pass 2 on rows), 2: threshold matcher (pass 3)
template <int KSD>
static void mt_launch(const GramParams& g, const float* d1, const float* d2, int D, int pairs, int kind, hipStream_t s) {
    constexpr int ROWB = mt_rowb(KSD);
    constexpr size_t kLds = 2 * (size_t)MT_TILE * ROWB + 16 + 4 * MT_QCAP + 4 * MT_STRIP;
    static XpPerDeviceOnce attr_once;
    if (attr_once.need()) {
        XP_HIP_WARN(hipFuncSetAttribute(reinterpret_cast<const void*>(&match_gram_kernel<KSD, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
        XP_HIP_WARN(hipFuncSetAttribute(reinterpret_cast<const void*>(&match_gram_kernel<KSD, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
        XP_HIP_WARN(hipFuncSetAttribute(reinterpret_cast<const void*>(&match_gram_kernel<KSD, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
        XP_HIP_WARN(hipFuncSetAttribute(reinterpret_cast<const void*>(&match_gram_kernel<KSD, 3, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
    }
    const int rowsmax = g.rows1P > g.rows2P ? g.rows1P : g.rows2P;
    hipLaunchKernelGGL(match_convert_kernel<KSD>, dim3(xp_cdiv(rowsmax, 4), pairs, 2), dim3(256), 0, s, d1, d2, g.counts, g.cnt_stride, g.which1, g.which2,
                       g.cap1, g.cap2, g.rows1P, g.rows2P, D, g.na, g.nb, g.maxbits, const_cast<unsigned char*>(g.A), const_cast<unsigned char*>(g.B),
                       g.rowkey, g.colkey, g.rcnt, g.ccnt, g.rowkey2);
    const dim3 grid(g.csplit, g.rows1P / MT_STRIP, pairs);
    if (kind == 2) { hipLaunchKernelGGL((match_gram_kernel<KSD, 3, false>), grid, dim3(256), kLds, s, g); return; }
    if (kind == 1) hipLaunchKernelGGL((match_gram_kernel<KSD, 1, true>), grid, dim3(256), kLds, s, g);
    else hipLaunchKernelGGL((match_gram_kernel<KSD, 1, false>), grid, dim3(256), kLds, s, g);
    hipLaunchKernelGGL((match_gram_kernel<KSD, 2, false>), grid, dim3(256), kLds, s, g);
}

// Shared front end: argument checks, workspace carve-up, norms, fp16 images, Gram passes of `kind`.
static int mt_front(const char* who, const float* d1, const float* d2, const int* counts, int cnt_stride, int which1, int which2, int pairs, int cap1, int cap2,
                    int D, void* workspace, size_t workspace_bytes, int kind, GramParams& g, MatchWs& m, hipStream_t s) {
    XP_CHECK_ARG(d1 && d2 && workspace, "%s: null pointer", who);
    XP_CHECK_ARG(pairs > 0 && cap1 > 0 && cap2 > 0 && D > 0 && D % 4 == 0, "%s: bad shape (D must be a multiple of 4)", who);
    XP_CHECK_ARG(cap1 <= 65536 && cap2 <= 65536, "%s: at most 65536 descriptors per image", who);
    XP_CHECK_ARG(pairs <= (1 << 29), "%s: too many pairs", who);
    XP_CHECK_ARG(D <= 256, "%s: descriptor size %d > 256 (the query strip is register resident; the reference's models use 64 and 256)", who, D);
    XP_CHECK_ARG(workspace_bytes >= xp_match_workspace_bytes(pairs, cap1, cap2, D), "%s: workspace too small", who);
    XP_CHECK_ARG(((uintptr_t)workspace & 15) == 0, "%s: workspace must be 16-byte aligned", who);
    m = mt_carve(workspace, pairs, cap1, cap2, D);
    g.rows1P = m.rows1P; g.rows2P = m.rows2P; g.counts = counts; g.cnt_stride = cnt_stride; g.which1 = which1; g.which2 = which2;
    g.cap1 = cap1; g.cap2 = cap2;
    g.rowkey = m.rowkey; g.colkey = m.colkey; g.rowkey2 = m.rowkey2; g.na = m.na; g.nb = m.nb;
    g.rcnt = m.rcnt; g.ccnt = m.ccnt; g.rcand = m.rcand; g.ccand = m.ccand;
    g.maxbits = m.words; g.A = m.A; g.B = m.B; g.knn = kind == 1;
    // enough workgroups to fill the chip twice when the lists are half full; at least one 64-wide tile per split
    int csplit = xp_cdiv(1024, (int64_t)(m.rows1P / MT_STRIP) * pairs);
    csplit = csplit < 1 ? 1 : (csplit > 16 ? 16 : csplit);
    if (csplit > m.rows2P / MT_TILE) csplit = m.rows2P / MT_TILE;
    g.csplit = csplit;
    hipLaunchKernelGGL(match_reset_kernel, dim3(1), dim3(64), 0, s, m.words);
    const int capmax = cap1 > cap2 ? cap1 : cap2;
    hipLaunchKernelGGL(match_norms_kernel, dim3(xp_cdiv(capmax, 64), pairs, 2), dim3(256), 0, s, d1, d2, counts, cnt_stride, which1, which2, cap1, cap2, D,
                       m.na, m.nb, m.words);
    const int ksd = mt_ksd(D);
    if (ksd == 4) mt_launch<4>(g, d1, d2, D, pairs, kind, s);
    else if (ksd == 8) mt_launch<8>(g, d1, d2, D, pairs, kind, s);
    else mt_launch<16>(g, d1, d2, D, pairs, kind, s);
    return XP_OK;
}

// d1 (pairs, cap1, D), d2 (pairs, cap2, D); counts: device int array, n1 of pair i at counts[i*cnt_stride + which1]
// (null -> every pair has cap rows).  Outputs (pairs, cap1|cap2): idx12/dist12, idx21/dist21; matches
// (pairs, cap1) q/t/dist + count per pair.
extern "C" int xp_match_mnn(const float* d1, const float* d2, const int* counts, int cnt_stride, int which1, int which2,
                            int pairs, int cap1, int cap2, int D, int mode, int* idx12, float* dist12, int* idx21,
                            float* dist21, int* match_q, int* match_t, float* match_d, int* match_count, void* workspace,
                            size_t workspace_bytes, void* stream) {
    XP_CHECK_ARG(idx12 && dist12 && idx21 && dist21 && match_q && match_t && match_d && match_count, "xp_match_mnn: null pointer");
    XP_CHECK_ARG(mode == 0 || mode == 1, "xp_match_mnn: mode 0 (strict_mnn) or 1 (legacy_crosscheck)");
    hipStream_t s = (hipStream_t)stream;
    XpProfScope prof("match_mnn", s, 0.0, 0.0);   // work depends on device-side counts: bench.py prices it from the fetched counts
    GramParams g{}; MatchWs m{};
    const int rc = mt_front("xp_match_mnn", d1, d2, counts, cnt_stride, which1, which2, pairs, cap1, cap2, D, workspace, workspace_bytes, 0, g, m, s);
    if (rc != XP_OK) return rc;
    const OvfList ovf{reinterpret_cast<int*>(m.words + 2), m.ovf_entry};
    hipLaunchKernelGGL(match_refine_kernel<false>, dim3(xp_cdiv(cap1, 4), pairs), dim3(256), 0, s, d1, d2, counts, counts, cnt_stride, which1, which2,
                       cap1, cap2, D, g.rcnt, g.rcand, idx12, dist12, ovf, 0);
    hipLaunchKernelGGL(match_refine_kernel<false>, dim3(xp_cdiv(cap2, 4), pairs), dim3(256), 0, s, d2, d1, counts, counts, cnt_stride, which2, which1,
                       cap2, cap1, D, g.ccnt, g.ccand, idx21, dist21, ovf, 1);
    hipLaunchKernelGGL(match_overflow_kernel<false>, dim3(MT_OVF_GRID), dim3(1024), 0, s, d1, d2, counts, cnt_stride, which1, which2, cap1, cap2, D, ovf,
                       idx12, dist12, idx21, dist21);
    hipLaunchKernelGGL(match_mutual_kernel, dim3(pairs), dim3(1024), 0, s, idx12, dist12, idx21, dist21, counts, counts, cnt_stride, which1,
                       which2, cap1, cap2, mode, match_q, match_t, match_d, match_count, m.scratch);
    XP_LAUNCH_CHECK();
    return XP_OK;
}

// cv2.BFMatcher(NORM_L2).knnMatch(d1, d2, k = 2) of reference matching.py:20-21: the two nearest targets of every query in exact arithmetic (ties ->
// lower index).  idx2 / dist2 (pairs, cap1, 2); a pair with fewer than two targets gets idx -1 / dist +inf in the missing slots.
extern "C" int xp_match_knn2(const float* d1, const float* d2, const int* counts, int cnt_stride, int which1, int which2, int pairs, int cap1, int cap2,
                             int D, int* idx2, float* dist2, void* workspace, size_t workspace_bytes, void* stream) {
    XP_CHECK_ARG(idx2 && dist2, "xp_match_knn2: null pointer");
    hipStream_t s = (hipStream_t)stream;
    XpProfScope prof("match_knn2", s, 0.0, 0.0);
    GramParams g{}; MatchWs m{};
    const int rc = mt_front("xp_match_knn2", d1, d2, counts, cnt_stride, which1, which2, pairs, cap1, cap2, D, workspace, workspace_bytes, 1, g, m, s);
    if (rc != XP_OK) return rc;
    const OvfList ovf{reinterpret_cast<int*>(m.words + 2), m.ovf_entry};
    hipLaunchKernelGGL(match_refine_kernel<true>, dim3(xp_cdiv(cap1, 4), pairs), dim3(256), 0, s, d1, d2, counts, counts, cnt_stride, which1, which2,
                       cap1, cap2, D, g.rcnt, g.rcand, idx2, dist2, ovf, 0);
    hipLaunchKernelGGL(match_overflow_kernel<true>, dim3(MT_OVF_GRID), dim3(1024), 0, s, d1, d2, counts, cnt_stride, which1, which2, cap1, cap2, D, ovf,
                       idx2, dist2, (int*)nullptr, (float*)nullptr);
    XP_LAUNCH_CHECK();
    return XP_OK;
}

// ThresholdMatcher (reference matching.py:77-102): every (q, t) with sqrt(2 - 2 clip(<a_q, b_t>, -1, 1)) < threshold.  The matrix pass nominates pairs whose
// fp16 inner product is within the proven error of the threshold, this kernel decides each in fp64 (one wave per nominated pair) and appends the accepted ones.
__global__ __launch_bounds__(256) void match_threshold_verify_kernel(const float* __restrict__ d1, const float* __restrict__ d2, int cap1, int cap2, int D,
                                                                     const int* __restrict__ hit_count, const int2* __restrict__ hits, int hit_cap,
                                                                     double threshold, int* __restrict__ out_pairs, float* __restrict__ out_dist,
                                                                     int* __restrict__ out_count, int out_cap) {
    const int nh = min(*hit_count, hit_cap);
    const int lane = threadIdx.x & 63;
    for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < nh; i += gridDim.x * 4) {
        const int2 hte = hits[i];
        const int pair = hte.x, q = (int)((unsigned)hte.y >> 16), t = hte.y & 0xffff;
        const float* a = d1 + ((int64_t)pair * cap1 + q) * D; const float* b = d2 + ((int64_t)pair * cap2 + t) * D;
        double sdot = 0.0;
        for (int k = lane * 4; k < D; k += 256) {
            const float4 x = *reinterpret_cast<const float4*>(a + k), y = *reinterpret_cast<const float4*>(b + k);
            sdot = fma((double)x.x, (double)y.x, sdot); sdot = fma((double)x.y, (double)y.y, sdot);
            sdot = fma((double)x.z, (double)y.z, sdot); sdot = fma((double)x.w, (double)y.w, sdot);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sdot += __shfl_xor(sdot, o, 64);
        sdot = sdot > 1.0 ? 1.0 : (sdot < -1.0 ? -1.0 : sdot);
        const double dist = sqrt(2.0 - 2.0 * sdot);
        if (lane == 0 && dist < threshold) {
            const int k = atomicAdd(out_count, 1);
            if (k < out_cap) { out_pairs[3 * k] = pair; out_pairs[3 * k + 1] = q; out_pairs[3 * k + 2] = t; out_dist[k] = (float)dist; }
        }
    }
}

// out_pairs (out_cap, 3) = (pair, query, target) in NO particular order (the host wrapper sorts them into the reference's row-major order), out_dist (out_cap),
// out_count[0] = accepted pairs (may exceed out_cap: the list was truncated), out_count[1] = nominated pairs (if > hit_cap the call must be repeated with a
// larger hit list: accepted pairs may be missing).  hits: caller-owned (hit_cap, 2) int32 scratch.  threshold <= 2 (beyond it every pair matches).
extern "C" int xp_match_threshold(const float* d1, const float* d2, const int* counts, int cnt_stride, int which1, int which2, int pairs, int cap1, int cap2,
                                  int D, double threshold, int* hits, int hit_cap, int* out_pairs, float* out_dist, int* out_count, int out_cap,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    XP_CHECK_ARG(hits && out_pairs && out_dist && out_count && hit_cap > 0 && out_cap > 0, "xp_match_threshold: null pointer / empty output");
    XP_CHECK_ARG(threshold >= 0.0 && threshold <= 2.0, "xp_match_threshold: threshold %g outside [0, 2] (unit descriptors: distances lie in [0, 2])", threshold);
    hipStream_t s = (hipStream_t)stream;
    XpProfScope prof("match_threshold", s, 0.0, 0.0);
    GramParams g{}; MatchWs m{};
    g.thr_c0 = (float)(1.0 - 0.5 * threshold * threshold) - 1e-6f;        // a.b > c0 (rounded down: the window below only widens)
    g.hit_count = out_count + 1; g.hits = reinterpret_cast<int2*>(hits); g.hit_cap = hit_cap;
    hipLaunchKernelGGL(match_reset_kernel, dim3(1), dim3(64), 0, s, reinterpret_cast<unsigned*>(out_count));       // zeroes 8 words: out_count must hold 8 ints
    const int rc = mt_front("xp_match_threshold", d1, d2, counts, cnt_stride, which1, which2, pairs, cap1, cap2, D, workspace, workspace_bytes, 2, g, m, s);
    if (rc != XP_OK) return rc;
    hipLaunchKernelGGL(match_threshold_verify_kernel, dim3(1024), dim3(256), 0, s, d1, d2, cap1, cap2, D, out_count + 1, reinterpret_cast<const int2*>(hits), hit_cap,
                       threshold, out_pairs, out_dist, out_count, out_cap);
    XP_LAUNCH_CHECK();
    return XP_OK;
}

// Candidate-list statistics of the LATEST xp_match_mnn call that used this workspace with the same shapes: out4 (device, 4 x uint64) =
// [sum of list lengths over live rows + columns, longest list, rows + columns that overflowed the inline list (finished by the overflow pass), live rows + columns].
extern "C" int xp_match_stats(void* workspace, const int* counts, int cnt_stride, int which1, int which2, int pairs, int cap1, int cap2, int D,
                              unsigned long long* out4, void* stream) {
    XP_CHECK_ARG(workspace && out4, "xp_match_stats: null pointer");
    XP_CHECK_ARG(pairs > 0 && cap1 > 0 && cap2 > 0 && D > 0 && D <= 256, "xp_match_stats: bad shape");
    const MatchWs m = mt_carve(workspace, pairs, cap1, cap2, D);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(match_reset_kernel, dim3(1), dim3(64), 0, s, reinterpret_cast<unsigned*>(out4));          // 8 words (a kernel, not a memset node)
    hipLaunchKernelGGL(match_stats_kernel, dim3(256), dim3(256), 0, s, m.rcnt, m.ccnt, counts, cnt_stride, which1, which2, cap1, cap2, pairs, out4);
    XP_LAUNCH_CHECK();
    return XP_OK;
}

extern "C" int xp_match_cand_cap(void) { return CAND_CAP; }
