// Wave-specialised form of the split-fp16 GEMM (gemm_h2.hip): same arithmetic, same operand formats, same epilogue — a different schedule.
//
// Why (profiles/r3_gemm_h2_stage_removal.txt).  In the tile kernel every wave does everything — global loads, the f32 -> two-plane
// split, the LDS tile stores, the barrier, the fragment reads, the MFMAs — and the stage-removal deltas of its K loop ADD UP to the whole loop: one
// serial chain per workgroup (reads wait on the barrier, the barrier on the slowest wave's stores, the stores on loads) that two workgroups per CU
// overlap only to half; the matrix pipe is busy 0.33 of the kernel.  Here the chain is cut in two:
//   waves 4..7  PRODUCERS  global loads (two slabs ahead in registers), split, ds_write of a slab into a slot of an LDS RING, "full" signal;
//   waves 0..3  CONSUMERS  wait "full", read the slot's fragments into a second register set while the MFMAs of the slab before run, "free" signal.
// A wave and its partner four waves up share a SIMD (workgroup waves go to SIMDs cyclically), so every SIMD hosts one producer (vector / memory
// instructions) and one consumer (matrix instructions): the two pipes of the SIMD run side by side instead of taking turns inside one wave.
// Slots are handed over with two monotone counters per slot in LDS (full: +1 per producer wave, free: +1 per consumer wave) — no s_barrier in the
// loop.  One 512-thread workgroup per CU, persistent over the tiles of the launch: the producers run on into the next tile's slabs while the
// consumers write the epilogue of the one before, so prologue and epilogue (a quarter of the tile kernel's time at K = 1536) are covered too.
// Every wait is bounded (a spin that runs out sets an error word and leaves: wrong numbers, never a hung GPU; the host checks the word in tests).
#include <stdlib.h>

#include "gemm_h2_core.h"

namespace {

constexpr int W_BM = 128, W_BN = 128, W_NS = 4;                      // tile, ring slots
constexpr int W_SLOT = (W_BM + W_BN) * H2_ROWB;                      // 36 864 B per slot: [A rows: 2 planes][B rows: 2 planes], 144-byte rows
constexpr size_t W_LDS = (size_t)W_NS * W_SLOT;
constexpr int W_SPIN_MAX = 1 << 18;                                  // ~30 ms of polling per wait before a wave gives up for good

struct WParams {
    GemmParams g;
    const uint4* wplanes;      // slab-major planes (xp_split_weights_h2)
    int ntm, ntn, ntiles;
    int* err;                  // device word, set to 1 by a wait that ran out
};

typedef __attribute__((address_space(3))) int w_lds_int;
#ifndef XP_H2W_DBG
#define XP_H2W_DBG 0      /* 1: count poll iterations per role into err[1] (producers waiting for a free slot) and err[2] (consumers waiting for a full one) */
#endif
__device__ __forceinline__ void w_wait_ge(volatile w_lds_int* flag, int target, int* err, bool& dead) {
    // one lane polls (LDS read + s_sleep), the wave reconverges behind it; a wave whose wait ran out once never waits again (dead)
    if ((threadIdx.x & 63) == 0 && !dead) {
        int n = 0;
        while (*flag < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++n > W_SPIN_MAX) { if (err) *err = 1; dead = true; break; }
        }
        if (XP_H2W_DBG && n) atomicAdd(err + (threadIdx.x >= 256 ? 1 : 2), n);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void w_signal(int* flag) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // this wave's stores / reads of the slot are done
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__global__ __launch_bounds__(512) void gemm_h2w_kernel(WParams wp) {
    using T = GemmTileH2<2, 2, 2, 2>;                                 // fragment / accumulator geometry of the four consumer waves
    extern __shared__ __align__(16) unsigned char w_lds[];
    const GemmParams& p = wp.g;
    // hand-off counters as STATIC shared arrays: the compiler must know they are LDS (a generic pointer makes every poll a flat load, whose
    // s_waitcnt vmcnt(0) drains the producer's prefetched global loads at every slab: the first build ran four times slower than the tile kernel)
    __shared__ int full[W_NS];                                         // full[s]: producer waves that finished writing the slot's current use
    __shared__ int free_[W_NS];                                        // free_[s]: consumer waves that finished reading it
    if (threadIdx.x < W_NS) { full[threadIdx.x] = 0; free_[threadIdx.x] = 0; }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int nslab = p.K / H2_BK;                                    // K % 32 == 0 (host)
    const int ntiles = wp.ntiles;
    bool dead = false;
    // Tile schedule of this (persistent) workgroup.  Workgroup ids go round-robin to the 8 XCDs (one L2 each): every XCD takes ONE contiguous range
    // of the logical tiles (N-tiles of an M-tile adjacent), and its workgroups walk that range side by side — the column tiles that share a block of
    // activation rows are in flight together on the same L2.  (With tile = workgroup id + k * grid every workgroup streamed its rows from HBM on its
    // own: 691 MB per launch at 4.2 TB/s — that, not the producers' instruction count, is what starved the consumers in the first builds.)
    const int xcd = blockIdx.x & 7, wslot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;       // gridDim.x is a multiple of 8 (host)
    const int tq = ntiles >> 3, tr = ntiles & 7;
    const int t_base = xcd * tq + (xcd < tr ? xcd : tr), t_cnt = tq + (xcd < tr ? 1 : 0);
    const int my_tiles = wslot < t_cnt ? (t_cnt - wslot + per_xcd - 1) / per_xcd : 0;
    if (my_tiles == 0) return;
    auto tile_at = [&](int k) { return t_base + wslot + k * per_xcd; };                        // k-th tile of this workgroup

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers
        const int ptid = (int)threadIdx.x - 256;
        int a_row[4], a_quad[4], a_dst[4], b_rowi[4], b_dst[4], b_unit[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int id = ptid + s * 256;
            const int r = id >> 3;
            a_row[s] = (r & ~7) | ((r & 7) >> 1) | ((r & 1) << 2);    // rows of an aligned group of 8 in the order 0,4,1,5,2,6,3,7: conflict-free ds_write_b64 groups
            a_quad[s] = id & 7;
            a_dst[s] = a_row[s] * H2_ROWB + a_quad[s] * 8;
            b_rowi[s] = id >> 3; b_unit[s] = id & 7;
            b_dst[s] = W_BM * H2_ROWB + b_rowi[s] * H2_ROWB + b_unit[s] * 16;
        }
        // global slab sequence of this workgroup: (tile k, slab t) for its tiles in order; loads run two slabs ahead of the stores
        // Loads in flight per producer wave: the activations stream from HBM (1-2 us under load) and a CU has only these four waves to cover it, so they
        // run PFA = 6 slabs ahead (96 registers); the weights are L2-resident and run PFB = 2 ahead.  (With 3 + 3 the kernel was latency-bound at
        // 5.7 k cycles per slab.)
        constexpr int PFA = 6, PFB = 2;
        float4 ra[PFA][4]; uint4 rb[PFB][4];
        const int total = my_tiles * nslab;
        auto tile_of = [&](int q, int& m0, int& n0, int& t) {
            const int k = q / nslab; t = q - k * nslab;
            const int tile = tile_at(k);
            const int mt = tile / wp.ntn, nt = tile - mt * wp.ntn;
            m0 = mt * W_BM; n0 = nt * W_BN;
        };
        auto gloadA = [&](int q, float4 (&a)[4]) {
            if (q >= total) return;
            int m0, n0, t; tile_of(q, m0, n0, t);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int m = m0 + a_row[s];
                a[s] = *reinterpret_cast<const float4*>(p.A + (int64_t)(m < p.M ? m : 0) * p.lda + t * H2_BK + a_quad[s] * 4);
            }
        };
        auto gloadB = [&](int q, uint4 (&b)[4]) {
            if (q >= total) return;
            int m0, n0, t; tile_of(q, m0, n0, t);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int n = n0 + b_rowi[s];
                b[s] = wp.wplanes[((int64_t)t * p.N + (n < p.N ? n : 0)) * H2_SLAB_UNITS + b_unit[s]];
            }
        };
#pragma unroll
        for (int u = 0; u < PFA; ++u) gloadA(u, ra[u]);
#pragma unroll
        for (int u = 0; u < PFB; ++u) gloadB(u, rb[u]);
        // One slab per call.  The wave never drains its own LDS queue for the slab it has just written: it signals the slab BEFORE (whose 12 stores
        // are complete once at most this slab's 12 are outstanding: counted lgkmcnt), and it reads the "free" counter of its slot ahead of the split
        // arithmetic, so neither the store drain nor the poll's LDS latency sits on the critical path.
        auto produce = [&](int q, auto ua_tag, auto ub_tag) {
            constexpr int UA = decltype(ua_tag)::value, UB = decltype(ub_tag)::value;     // q % PFA, q % PFB: static register sets
            if (q >= total) return;
            const int slot = q % W_NS, gen = q / W_NS;
            unsigned char* buf = w_lds + (size_t)slot * W_SLOT;
            const int seen = gen > 0 ? *(volatile w_lds_int*)&free_[slot] : 0;       // early look (all lanes: one broadcast read)
            uint2 p0[4], p1[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) h2_split4(ra[UA][s], p0[s], p1[s]);
            if (gen > 0 && seen < 4 * gen) w_wait_ge((volatile w_lds_int*)&free_[slot], 4 * gen, wp.err, dead);      // rare: the ring is full
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                *reinterpret_cast<uint2*>(buf + a_dst[s]) = p0[s];
                *reinterpret_cast<uint2*>(buf + a_dst[s] + 64) = p1[s];
                *reinterpret_cast<uint4*>(buf + b_dst[s]) = rb[UB][s];
            }
            gloadA(q + PFA, ra[UA]);                                 // into the registers just stored
            gloadB(q + PFB, rb[UB]);
            if (q > 0) {
                asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");   // everything older than this slab's 12 stores has completed: slab q - 1 is in LDS
                if (lane == 0) __hip_atomic_fetch_add(&full[(q - 1) % W_NS], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (q == total - 1) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(&full[slot], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        };
        for (int q = 0; q < total; q += 6) {
            produce(q, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            produce(q + 1, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
            produce(q + 2, std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});
            produce(q + 3, std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{});
            produce(q + 4, std::integral_constant<int, 4>{}, std::integral_constant<int, 0>{});
            produce(q + 5, std::integral_constant<int, 5>{}, std::integral_constant<int, 1>{});
        }
        return;
    }

    // ---------------------------------------------------------------------- consumers
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 31, fh = lane >> 5;
    const int a_frag = (wm * 64 + fr) * H2_ROWB + 16 * fh;
    const int b_frag = W_BM * H2_ROWB + (wn * 64 + fr) * H2_ROWB + 16 * fh;
    struct Frags { f16x8_t a[2][2], b[2][2]; };                      // one k-step: [plane][tile] — two sets alternate (64 registers in all)
    auto read_frags = [&](const unsigned char* buf, int ks, Frags& f) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f.a[pl][i] = *reinterpret_cast<const f16x8_t*>(buf + a_frag + pl * 64 + ks * 32 + i * 32 * H2_ROWB);
                f.b[pl][i] = *reinterpret_cast<const f16x8_t*>(buf + b_frag + pl * 64 + ks * 32 + i * 32 * H2_ROWB);
            }
    };
    f32x16 acc[2][2];
    auto mfmas = [&](const Frags& f) {
        constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};          // smallest partial products first (as the tile kernel)
#pragma unroll
        for (int pp = 0; pp < 3; ++pp)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[PA[pp]][i], f.b[PB[pp]][j], acc[i][j], 0, 0, 0);
    };
    Frags f0, f1;                                                     // f0: k-step 0 of a slab, f1: k-step 1
    int q = 0;                                                        // global slab index of this workgroup (same sequence as the producers')
    auto slot_of = [&](int qq) { return w_lds + (size_t)(qq % W_NS) * W_SLOT; };
    auto wait_full = [&](int qq) { w_wait_ge((volatile w_lds_int*)&full[qq % W_NS], 4 * (qq / W_NS + 1), wp.err, dead); };
    auto release = [&](int qq) {
        if (lane == 0) __hip_atomic_fetch_add(&free_[qq % W_NS], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto lds_done = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
    // pipeline over k-steps: while one k-step multiplies, the fragments of the next one (same slab, or the next slab once it is full) are read
    wait_full(q);
    read_frags(slot_of(q), 0, f0);
    lds_done();
    for (int kt = 0; kt < my_tiles; ++kt) {
        const int tile = tile_at(kt);
        const int mt = tile / wp.ntn, nt = tile - mt * wp.ntn;
        const bool last_tile = kt + 1 >= my_tiles;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int t = 0; t < nslab; ++t, ++q) {
            read_frags(slot_of(q), 1, f1);                           // k-step 1 of slab q arrives while ...
            __builtin_amdgcn_sched_barrier(0);
            mfmas(f0);                                               // ... k-step 0 multiplies
            __builtin_amdgcn_sched_barrier(0);
            lds_done();
            release(q);                                              // both k-steps of the slot are in registers
            const bool more = t + 1 < nslab || !last_tile;           // the next slab of this workgroup's sequence (it may belong to the next tile)
            if (more) {
                wait_full(q + 1);
                read_frags(slot_of(q + 1), 0, f0);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfmas(f1);
            __builtin_amdgcn_sched_barrier(0);
            if (more) lds_done();
        }
        // the leading dimensions go through an opaque scalar move per tile: otherwise the epilogue's ~200 tile-invariant 64-bit element offsets are hoisted
        // out of the persistent loop and live across the K loop (1.4 KB of scratch per lane in the first build)
        GemmParams pe = p;
        asm volatile("" : "+s"(pe.ldc), "+s"(pe.ldres));
        gemm_epilogue<T, 2, 2, true>(pe, mt * W_BM, nt * W_BN, acc);
    }
}

}  // namespace

__device__ int g_h2w_err[4];   // [0] set by a bounded wait that ran out (never expected; read by xp_gemm_h2w_error); [1], [2]: poll counts of XP_H2W_DBG builds

// Applies to plain GEMMs whose K loop is long enough to fill the ring and whose K is a whole, even number of 32-wide slabs.
// OPT-IN (XP_H2W=1): measured 1.7x SLOWER than the tile kernel on the deep-stage shapes (M 19200 x N 384 x K 1536: 166 us against 99.5; round 3,
// tools/h2w_dbg.sh: the consumers poll 40x more than the producers — four producer waves per CU cannot stage a 128 x 128 slab (loads, split, 12
// ds_writes, addressing: ~1 k issue cycles per wave) in the 768 cycles its MFMAs take, even six slabs ahead; the tile kernel spreads the same staging
// over the 8 waves of two workgroups).  Kept as a tested alternative schedule with bounded hand-offs: what it needs next is eight producer waves
// (twelve waves per workgroup: consumers within 168 registers) or weights by LDS-DMA.
bool xp_gemm_h2w_applies(const GemmParams& p) {
    static const bool on = getenv("XP_H2W") != nullptr && atoi(getenv("XP_H2W")) != 0;
    return on && p.mode == 0 && p.K % 64 == 0 && p.K >= 256 && p.N >= 96 && p.lda % 4 == 0;
}

static int* h2w_err_ptr() {
    static int* ptr = nullptr;
    if (!ptr) { void* q = nullptr; if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_h2w_err)) == hipSuccess) ptr = (int*)q; }
    return ptr;
}
// 1 if a ring hand-off of gemm_h2w_kernel ever timed out in this process (results of that launch are wrong); synchronises the device
extern "C" int xp_gemm_h2w_error(void) {
    int v[4] = {0, 0, 0, 0};
    int* ptr = h2w_err_ptr();
    if (!ptr || hipMemcpy(v, ptr, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (XP_H2W_DBG) fprintf(stderr, "gemm_h2w polls: producers waiting for a free slot %d, consumers waiting for a full slot %d\n", v[1], v[2]);
    return v[0];
}

int xp_gemm_h2w_launch(const GemmParams& p, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        XP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h2w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)W_LDS));
        attr_set = true;
    }
    int* err_word = h2w_err_ptr();
    WParams wp;
    wp.g = p;
    wp.wplanes = reinterpret_cast<const uint4*>(p.Wt);
    wp.ntm = xp_cdiv(p.M, W_BM); wp.ntn = xp_cdiv(p.N, W_BN); wp.ntiles = wp.ntm * wp.ntn;
    wp.err = err_word;
    static int n_cu = 0;
    if (!n_cu) { int dev = 0; hipDeviceProp_t prop; (void)hipGetDevice(&dev); (void)hipGetDeviceProperties(&prop, dev); n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256; }
    int grid = wp.ntiles < n_cu ? (wp.ntiles + 7) / 8 * 8 : n_cu / 8 * 8;     // a multiple of 8: the same number of workgroups on every XCD
    if (grid < 8) grid = 8;
    static const bool by_shape = getenv("XP_PROF_SHAPES") != nullptr;
    std::string tag = "gemm_h2w_mfma_128x128";
    if (by_shape) tag += "_M" + std::to_string(p.M) + "_N" + std::to_string(p.N) + "_K" + std::to_string(p.K) + (p.act == 1 ? "_gelu" : "");
    XpProfScope prof(tag.c_str(), s, 2.0 * p.M * p.N * p.K, 4.0 * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N * (p.res ? 2 : 1)));
    hipLaunchKernelGGL(gemm_h2w_kernel, dim3(grid), dim3(512), W_LDS, s, wp);
    XP_LAUNCH_CHECK();
    return XP_OK;
}
