// Detection post-processing on the heat-map: box NMS, keypoint extraction, descriptor sampling.
//   box_nms            reference xpoint/utils/utils.py:148-192  (torchvision.ops.nms / batched_nms)
//   extract_keypoints  reference predict_align_image_pair.py:242-243, predict_keypoints.py:213-215
//   sample_descriptors reference xpoint/utils/utils.py:229-238  (F.grid_sample bilinear, align_corners)
#include <stdlib.h>

#include "xp_common.h"

extern "C" size_t xp_extract_keypoints_workspace_bytes(int batch, int H, int W);

namespace {

// ---------------------------------------------------------------------------------------------
// Box NMS.  Greedy NMS (sort by score desc, stable; a kept box suppresses every later box with
// IoU > thr) is sequential by definition.  Parallel formulation with the IDENTICAL result for the
// strict total order (score desc, row-major index asc): iterate to a fixed point
//     undecided p becomes SUPPRESSED if a KEPT box overlaps it,
//     undecided p becomes KEPT if every overlapping box of higher priority is SUPPRESSED
// (all boxes overlapping a kept box are suppressed: lower priority ones by it, higher priority ones
// by an earlier keep, otherwise it would not have been kept).  Boxes are size x size squares centred
// on integer pixels, so "overlap with IoU > thr" is a translation-invariant predicate of (|dy|,|dx|):
// the host evaluates it with torchvision's float32 arithmetic into a bit table (ovl[dy] bit dx).
// Tiles of 32x32 pixels with a halo iterate locally in LDS; launches repeat until no pixel in the
// batch is undecided (device counter).
// ---------------------------------------------------------------------------------------------
constexpr int NMS_TILE = 32;          // tile width: the row masks are 64-bit words, NMS_TILE + 2 * NMS_MAXR <= 64
#ifndef XP_NMS_TH
#define XP_NMS_TH 64   /* measured at 480 x 640, 16 images: 0.41 ms (32), 0.37-0.39 (64), 0.44 (128), 0.58 (192) — fewer tiles pay the fixed per-tile cost, taller ones serialise */
#endif
constexpr int NMS_TH = XP_NMS_TH;     // tile height (rows are not limited by the mask width)
constexpr int NMS_MAXR = 15;
constexpr int NMS_MAX_SWEEPS = 64;        // counters[] slots (one per sweep of a round)

struct NmsTable { unsigned long long win[NMS_MAXR + 1]; int reach; };   // win[|dy|]: bit (reach + dx) set iff overlap

enum : int { ST_NONE = 0, ST_UNDECIDED = 1, ST_KEPT = 2 };

// One sweep over every active 32x32 tile.  Tile state lives in LDS as per-row 64-bit masks (undecided / kept)
// plus the scores; the owned undecided candidates are compacted into a list so that lanes work on candidates,
// not pixels.  A candidate's test is 2R+1 mask tests for a kept neighbour and a walk over the set bits of the
// undecided neighbours (few) for the priority test.
//   first != 0: states are derived from prob (> min_prob) instead of being read; every tile is active.
//   sweep k > 0 exits at once when sweep k-1 left nothing undecided, and skips tiles with no undecided pixel.
// Each active tile writes its owned part of `out` (kept ? score : 0) and of `state`.
__global__ __launch_bounds__(256) void nms_sweep_kernel(const float* __restrict__ prob, uint8_t* __restrict__ state,
                                                        float* __restrict__ out, uint8_t* __restrict__ tile_active, int H, int W,
                                                        NmsTable tab, float min_prob, int* __restrict__ counters, int sweep,
                                                        int first, int local_iters) {
    if (sweep > 0 && counters[sweep - 1] == 0) return;   // converged in an earlier sweep (uniform early exit)
    const int tile_id = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (!first && !tile_active[tile_id]) return;
    constexpr int TWMAX = NMS_TILE + 2 * NMS_MAXR, THMAX = NMS_TH + 2 * NMS_MAXR;
    __shared__ float s_sc[TWMAX * THMAX];
    __shared__ unsigned long long m_und[THMAX], m_kept[THMAX];
    __shared__ unsigned short s_list[NMS_TILE * NMS_TH];
    __shared__ int s_n, s_changed;
    const int R = tab.reach;
    const int TW = NMS_TILE + 2 * R, TH = NMS_TH + 2 * R;
    const int b = blockIdx.z;
    const int y0 = blockIdx.y * NMS_TH - R, x0 = blockIdx.x * NMS_TILE - R;
    const float* pb = prob + (int64_t)b * H * W;
    uint8_t* sb = state + (int64_t)b * H * W;
    float* ob = out + (int64_t)b * H * W;
    for (int i = threadIdx.x; i < TH; i += 256) { m_und[i] = 0ull; m_kept[i] = 0ull; }
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < TW * TH; i += 256) {
        const int ty = i / TW, tx = i - ty * TW;
        const int y = y0 + ty, x = x0 + tx;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;
        const float sc = in ? pb[(int64_t)y * W + x] : 0.f;
        int st = ST_NONE;
        if (in) st = first ? (sc > min_prob ? ST_UNDECIDED : ST_NONE) : sb[(int64_t)y * W + x];
        s_sc[i] = sc;
        if (st == ST_UNDECIDED) {
            atomicOr(&m_und[ty], 1ull << tx);
            const bool own = ty >= R && ty < R + NMS_TH && tx >= R && tx < R + NMS_TILE;
            if (own) { const int pos = atomicAdd(&s_n, 1); s_list[pos] = (unsigned short)(ty * 64 + tx); }
        } else if (st == ST_KEPT) {
            atomicOr(&m_kept[ty], 1ull << tx);
        }
    }
    __syncthreads();
    const int n = s_n;
    if (n > 0) {
        for (int it = 0; it < local_iters; ++it) {
            if (threadIdx.x == 0) s_changed = 0;
            __syncthreads();
            for (int li = threadIdx.x; li < n; li += 256) {
                const int ty = s_list[li] >> 6, tx = s_list[li] & 63;
                const unsigned long long mybit = 1ull << tx;
                if (!(m_und[ty] & mybit)) continue;
                const float sc = s_sc[ty * TW + tx];
                const int sh = tx - R;
                // Concurrent lanes flip bits while we look.  A lane that keeps q sets q's KEPT bit BEFORE clearing its
                // UNDECIDED bit, so reading UNDECIDED first and KEPT afterwards can never miss q in both.  Order:
                // (1) cheap filter on KEPT, (2) priority walk over UNDECIDED, (3) KEPT again (the authoritative read).
                const volatile unsigned long long* vk = m_kept;
                const volatile unsigned long long* vu = m_und;
                bool suppressed = false;
                for (int dy = -R; dy <= R; ++dy)
                    if (vk[ty + dy] & (tab.win[dy < 0 ? -dy : dy] << sh)) { suppressed = true; break; }
                if (suppressed) { atomicAnd(&m_und[ty], ~mybit); s_changed = 1; continue; }
                bool blocked = false;
                for (int dy = -R; dy <= R && !blocked; ++dy) {
                    unsigned long long u = vu[ty + dy] & (tab.win[dy < 0 ? -dy : dy] << sh);
                    if (dy == 0) u &= ~mybit;
                    while (u) {
                        const int bx = __ffsll((long long)u) - 1;
                        u &= u - 1;
                        const float sq = s_sc[(ty + dy) * TW + bx];
                        // q has higher priority: larger score, or equal score and earlier row-major index
                        if (sq > sc || (sq == sc && (dy < 0 || (dy == 0 && bx < tx)))) { blocked = true; break; }
                    }
                }
                if (blocked) continue;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                for (int dy = -R; dy <= R; ++dy)
                    if (vk[ty + dy] & (tab.win[dy < 0 ? -dy : dy] << sh)) { suppressed = true; break; }
                if (suppressed) { atomicAnd(&m_und[ty], ~mybit); s_changed = 1; continue; }
                atomicOr(&m_kept[ty], mybit);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                atomicAnd(&m_und[ty], ~mybit);
                s_changed = 1;
            }
            __syncthreads();
            if (!s_changed) break;
            __syncthreads();
        }
    }
    int left = 0;
    for (int o = threadIdx.x; o < NMS_TILE * NMS_TH; o += 256) {
        const int ty = R + o / NMS_TILE, tx = R + o % NMS_TILE;
        const int y = y0 + ty, x = x0 + tx;
        if (y < H && x < W) {
            const bool kept = (m_kept[ty] >> tx) & 1ull, und = (m_und[ty] >> tx) & 1ull;
            sb[(int64_t)y * W + x] = (uint8_t)(kept ? ST_KEPT : (und ? ST_UNDECIDED : ST_NONE));
            ob[(int64_t)y * W + x] = kept ? s_sc[ty * TW + tx] : 0.f;
            left |= und;
        }
    }
    left = __syncthreads_or(left);
    if (threadIdx.x == 0) {
        tile_active[tile_id] = left ? 1 : 0;
        if (left) atomicAdd(&counters[sweep], 1);
    }
}


// ---------------------------------------------------------------------------------------------
// Box NMS, round-3 form: TWO launches, exact, convergence guaranteed inside the second one.
// Measured on a 480 x 640 heat map of this model (tools/nms_rounds.py): 31 k of 307 k pixels are candidates; the candidates that are
// maxima of their own overlap window (2.5 k) decide 22.5 k more in one step, and the remaining 6.4 k settle in three more keep rounds.
//   A  nms_localmax_kernel: one pass over the image in 32 x 64 tiles + halo — candidate mask, "kept in round 1" mask (a candidate with no
//      higher-priority candidate in its window; needs only the raw scores, so no tile ever waits for another) and the provisional output
//      (score where kept, 0 elsewhere).  Bit masks go to global memory as one uint32 per (row, 32 columns): a tile owns its words.
//   B  nms_finish_kernel: ONE workgroup per image holds the image's two bit masks in LDS (480 x 640: 87 KB), drops every candidate that a
//      round-1 keeper suppresses, ranks what is left (row-major rank = row base + word prefix + popcount) into a score table in LDS and runs
//      the keep / suppress fixed point as Jacobi rounds separated by workgroup barriers until nothing is undecided — no other workgroup, no
//      host round trip, no sweep count.  Later keepers patch the output sparsely.
// Images whose masks do not fit LDS (1024 x 1024) keep the sweep form above.
// ---------------------------------------------------------------------------------------------
#ifndef XP_NMS_DBG
#define XP_NMS_DBG 0   /* timing experiments only (wrong results), nms_localmax_kernel: 1 no candidate tests, 2 no out store, 4 no list building, 8 no loads, 16 no window walk */
#endif
template <int RT>
__global__ __launch_bounds__(256) void nms_localmax_kernel(const float* __restrict__ prob, float* __restrict__ out, unsigned* __restrict__ cand32,
                                                           unsigned* __restrict__ kept32, int H, int W, int wpr, NmsTable tab, float min_prob) {
    constexpr int RMAX = RT > 0 ? RT : NMS_MAXR;                     // compile-time reach: the tile image is sized for it (13.4 instead of 15.4 KB at reach 6: 7 instead of 4 workgroups per CU with the lists)
    constexpr int TWMAX = NMS_TILE + 2 * RMAX, THMAX = NMS_TH + 2 * RMAX;
    __shared__ float s_sc[TWMAX * THMAX];
    __shared__ unsigned long long m_cand[THMAX], m_kept[THMAX];
    __shared__ unsigned short s_list[NMS_TILE * NMS_TH], s_surv[NMS_TILE * NMS_TH];
    __shared__ int s_n, s_m;
    const int R = RT > 0 ? RT : tab.reach;          // RT > 0: compile-time reach (the window reads below unroll into straight-line code)
    const int TW = NMS_TILE + 2 * R, TH = NMS_TH + 2 * R;
    const int b = blockIdx.z;
    const int y0 = blockIdx.y * NMS_TH - R, x0 = blockIdx.x * NMS_TILE - R;
    const float* pb = prob + (int64_t)b * H * W;
    float* ob = out + (int64_t)b * H * W;
    for (int i = threadIdx.x; i < TH; i += 256) { m_cand[i] = 0ull; m_kept[i] = 0ull; }
    if (threadIdx.x == 0) { s_n = 0; s_m = 0; }
    __syncthreads();
    // tile + halo, a wave per tile row (lane = column, TW <= 62): the row's candidate mask is one ballot — no LDS atomics on the masks — and the
    // owned candidates of the row take their list slots from ONE atomicAdd per wave and row (popcount prefix inside the wave).  Batches of 8 rows per
    // wave: the loads of a batch are issued before the first is used.
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int x = x0 + lane;
        const bool xin = lane < TW && x >= 0 && x < W;
        const unsigned long long own_cols = ((1ull << NMS_TILE) - 1ull) << R;
        for (int r0 = wave; r0 < TH; r0 += 4 * 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int ty = r0 + 4 * k, y = y0 + ty;
                v[k] = (ty < TH && xin && y >= 0 && y < H && !((XP_NMS_DBG & 8) && ty > 2)) ? pb[(int64_t)y * W + x] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int ty = r0 + 4 * k;
                if (ty >= TH) continue;                              // wave-uniform
                if (lane < TW) s_sc[ty * TW + lane] = v[k];
                const unsigned long long m = __ballot(v[k] > min_prob);      // (elements outside the image were loaded as 0 <= min_prob)
                if (lane == 0) m_cand[ty] = m;
                const unsigned long long own = (ty >= R && ty < R + NMS_TH) ? (m & own_cols) : 0ull;
                if (own && !(XP_NMS_DBG & 4)) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&s_n, __popcll(own));
                    base = __shfl(base, 0, 64);
                    if ((own >> lane) & 1ull) s_list[base + __popcll(own & ((1ull << lane) - 1ull))] = (unsigned short)(ty * 64 + lane);
                }
            }
        }
    }
    __syncthreads();
    const int n = (XP_NMS_DBG & 1) ? 0 : s_n;
    // Phase 1, every candidate: quick reject — the 8 immediate neighbours are inside every overlap window (non-candidates score <= min_prob < sc:
    // harmless); eight independent LDS reads settle ~8 of 9 candidates.  The survivors are COMPACTED into a second list (one LDS atomic per wave and
    // batch) before the window walk: walking inside this loop, nearly every wave had a surviving lane and ran the whole data-dependent walk with
    // one lane in nine active (16 walk batches per tile on the bench's dense maps; now 2).
    for (int l0 = 0; l0 < n; l0 += 256) {
        const int li = l0 + threadIdx.x;
        bool top = li < n;
        unsigned short item = 0;
        if (top) {
            item = s_list[li];
            if (R >= 1) {
                const int ty = item >> 6, tx = item & 63;
                const float sc = s_sc[ty * TW + tx];
                const float* c = s_sc + ty * TW + tx;
                const float n0 = c[-TW - 1], n1 = c[-TW], n2 = c[-TW + 1], n3 = c[-1], n4 = c[1], n5 = c[TW - 1], n6 = c[TW], n7 = c[TW + 1];
                top = !(n0 >= sc || n1 >= sc || n2 >= sc || n3 >= sc || n4 > sc || n5 > sc || n6 > sc || n7 > sc);
            }
        }
        const unsigned long long sv = __ballot(top);
        if (sv) {
            const int lane = threadIdx.x & 63;
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_m, __popcll(sv));
            base = __shfl(base, 0, 64);
            if (top) s_surv[base + __popcll(sv & ((1ull << lane) - 1ull))] = item;
        }
    }
    __syncthreads();
    const int m = (XP_NMS_DBG & 16) ? 0 : s_m;
    // Phases 2 and 3, survivors only: is any pixel of the overlap window AHEAD of this one (higher score, or the same score earlier in row-major order)?
    if constexpr (RT > 0) {
        // Compile-time reach: every position of the (2 RT + 1)^2 square is read as a plain score at an immediate offset — straight-line code, all LDS
        // reads independent and in flight together; whether a position belongs to the overlap window is a uniform bit that masks the comparison, and a
        // non-candidate can never be ahead (its score <= min_prob < sc).  Phase 2 looks at the 7 x 7 neighbourhood (minus the 3 x 3 already seen) and
        // compacts again; phase 3 reads the rest of the square for what is left (the true maxima and a few more: about one wave per tile).  The
        // bit walk below (runtime reach) is a chain of ffs -> dependent LDS read -> compare -> branch, ~130 links for a true maximum, and every wave
        // holds one: 40 of the kernel's 65 us on the bench's dense maps.
        auto ahead_in = [&](int ty, int tx, float sc, auto rin_tag, auto rout_tag) {          // positions with RIN < max(|dy|, |dx|) <= ROUT
            constexpr int RIN = decltype(rin_tag)::value, ROUT = decltype(rout_tag)::value;
            const float* c = s_sc + ty * (NMS_TILE + 2 * RT) + tx;
            bool ahead = false;
#pragma unroll
            for (int dy = -ROUT; dy <= ROUT; ++dy) {
                const unsigned long long w = tab.win[dy < 0 ? -dy : dy] >> (RT - ROUT);
#pragma unroll
                for (int dx = -ROUT; dx <= ROUT; ++dx) {
                    if ((dy < 0 ? -dy : dy) <= RIN && (dx < 0 ? -dx : dx) <= RIN) continue;          // compile time
                    const float sq = c[dy * (NMS_TILE + 2 * RT) + dx];
                    const bool in = (w >> (dx + ROUT)) & 1ull;                                      // uniform
                    ahead |= in && ((dy < 0 || (dy == 0 && dx < 0)) ? sq >= sc : sq > sc);
                }
            }
            return ahead;
        };
        constexpr int R2 = RT < 3 ? RT : 3;
        if (threadIdx.x == 0) s_n = 0;                  // reused as the length of the second survivor list (s_list is free again)
        __syncthreads();
        for (int l0 = 0; l0 < m; l0 += 256) {
            const int li = l0 + threadIdx.x;
            bool top = li < m;
            unsigned short item = 0;
            if (top) {
                item = s_surv[li];
                const int ty = item >> 6, tx = item & 63;
                top = !ahead_in(ty, tx, s_sc[ty * TW + tx], std::integral_constant<int, 1>{}, std::integral_constant<int, R2>{});
            }
            const unsigned long long sv = __ballot(top);
            if (sv) {
                const int lane = threadIdx.x & 63;
                int base = 0;
                if (lane == 0) base = atomicAdd(&s_n, __popcll(sv));
                base = __shfl(base, 0, 64);
                if (top) s_list[base + __popcll(sv & ((1ull << lane) - 1ull))] = item;
            }
        }
        __syncthreads();
        const int m2 = s_n;
        for (int li = threadIdx.x; li < m2; li += 256) {
            const int ty = s_list[li] >> 6, tx = s_list[li] & 63;
            if (!ahead_in(ty, tx, s_sc[ty * TW + tx], std::integral_constant<int, R2>{}, std::integral_constant<int, RT>{})) atomicOr(&m_kept[ty], 1ull << tx);
        }
    } else {
    // runtime reach: walk over the candidate bits of the overlap window, early exit
    for (int li = threadIdx.x; li < m; li += 256) {
        const int ty = s_surv[li] >> 6, tx = s_surv[li] & 63;
        const float sc = s_sc[ty * TW + tx];
        const int sh = tx - R;
        bool top = true;
        for (int dy = -R; dy <= R && top; ++dy) {
            unsigned long long u = m_cand[ty + dy] & (tab.win[dy < 0 ? -dy : dy] << sh);
            if (dy == 0) u &= ~(1ull << tx);
            while (u) {
                const int bx = __ffsll((long long)u) - 1;
                u &= u - 1;
                const float sq = s_sc[(ty + dy) * TW + bx];
                if (sq > sc || (sq == sc && (dy < 0 || (dy == 0 && bx < tx)))) { top = false; break; }    // q has higher priority (score, then row-major index)
            }
        }
        if (top) atomicOr(&m_kept[ty], 1ull << tx);
    }
    }
    __syncthreads();
    for (int o = threadIdx.x; o < NMS_TILE * NMS_TH; o += 256) {      // NMS_TILE = 32: a wave instruction writes two full 128-byte row segments
        const int ty = R + o / NMS_TILE, tx = R + o % NMS_TILE;
        const int y = y0 + ty, x = x0 + tx;
        if (y < H && x < W && !((XP_NMS_DBG & 2) && o > 3)) ob[(int64_t)y * W + x] = ((m_kept[ty] >> tx) & 1ull) ? s_sc[ty * TW + tx] : 0.f;
    }
    for (int r = threadIdx.x; r < NMS_TH; r += 256) {
        const int y = y0 + R + r;
        if (y < H) {
            const int64_t wo = ((int64_t)b * H + y) * wpr + blockIdx.x;
            cand32[wo] = (unsigned)((m_cand[R + r] & ~m_kept[R + r]) >> R);     // the undecided set after round 1
            kept32[wo] = (unsigned)(m_kept[R + r] >> R);
        }
    }
}

// ---- pass A2: candidates that a round-1 keeper suppresses leave the undecided set.  Bit-mask work only; one workgroup per band of 32 rows.
constexpr int NMS_S1_ROWS = 32;
__host__ __device__ inline int rw_of(int W) { return (W + 31) / 32 + 2; }
__device__ __forceinline__ unsigned long long nms_window(const unsigned* __restrict__ row, int x, int R) {
    // bits [x - R, x - R + 63] of a padded mask row (one zero word on the left): bit j <-> column x - R + j
    const int xp = x - R + 32;
    const int wi = xp >> 5, sh = xp & 31;
    return (((unsigned long long)row[wi + 1] << 32) | row[wi]) >> sh;
}
// "does any bit of the padded mask M (row r of the image at M + (r + pad_rows) * rw) fall into the window of (y, x)"; RT > 0: compile-time reach,
// the 2 RT + 1 row reads are unrolled and issued together (one LDS round trip instead of 2 RT + 1 dependent ones)
template <int RT>
__device__ __forceinline__ bool nms_any_in_window(const unsigned* __restrict__ M, int rw, int yrow, int x, int R, const NmsTable& tab) {
    constexpr int NW = RT > 0 ? 2 * RT + 1 : 2 * NMS_MAXR + 1;
    unsigned long long acc = 0ull;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const int dy = k - (NW >> 1);
        if (RT > 0 || (dy >= -R && dy <= R)) acc |= nms_window(M + (yrow + dy) * rw, x, R) & tab.win[dy < 0 ? -dy : dy];
    }
    return acc != 0ull;
}

// One Jacobi round over the whole batch (all CUs): every undecided candidate reads the masks as the previous round left them —
//   a kept box in its window      -> suppressed (leaves the undecided set);
//   no undecided box of higher priority in its window -> kept (score written to `out`);
// and the band's rows of the next masks are written.  Round 1 after the local-maximum pass is pure suppression (22.5 k of 28.9 k candidates on a
// 480 x 640 map), rounds 2 and 3 leave ~700 of 6.4 k: what remains goes to the finisher.  A workgroup owns a band of 32 rows and keeps the band's
// rows + reach of both masks in LDS; neighbour scores come from the L2-resident heat map.
// SUP_ONLY: the suppression half alone (no neighbour-score reads) — round 1 after the local-maximum pass, where no candidate can become a keeper yet
// (its higher-priority neighbours are still undecided until this very round suppresses them).  The full form distributes candidates by mask word,
// which serialises dense blobs (32 undecided pixels in one word = 32 dependent chains in one lane: 95 us for round 2 on the model's maps against 11 us
// for the suppression half); the finisher's compacted list balances them, so by default only the suppression half runs wide (XP_NMS_WIDE_ROUNDS adds
// full rounds for experiments).
template <int RT, bool SUP_ONLY>
__global__ __launch_bounds__(256) void nms_round_kernel(const float* __restrict__ prob, float* __restrict__ out, const unsigned* __restrict__ und_in,
                                                        const unsigned* __restrict__ kept_in, unsigned* __restrict__ und_out, unsigned* __restrict__ kept_out,
                                                        int H, int W, int wpr, NmsTable tab) {
    __shared__ unsigned s_k[(NMS_S1_ROWS + 2 * NMS_MAXR) * (2 + 64)];      // kept rows of the band + reach, padded (wpr <= 64: host)
    __shared__ unsigned s_u[(NMS_S1_ROWS + 2 * NMS_MAXR) * (2 + 64)];      // undecided rows
    const int R = RT > 0 ? RT : tab.reach;
    constexpr int NW = RT > 0 ? 2 * RT + 1 : 2 * NMS_MAXR + 1;
    const int rw = wpr + 2, b = blockIdx.y, y0 = blockIdx.x * NMS_S1_ROWS;
    const int nrows = NMS_S1_ROWS + 2 * R;
    const float* pb = prob + (int64_t)b * H * W;
    float* ob = out + (int64_t)b * H * W;
    for (int i = threadIdx.x; i < nrows * rw; i += 256) {
        const int r = i / rw, wi = i - r * rw - 1, y = y0 - R + r;
        const bool in = y >= 0 && y < H && wi >= 0 && wi < wpr;
        s_k[i] = in ? kept_in[((int64_t)b * H + y) * wpr + wi] : 0u;
        s_u[i] = in ? und_in[((int64_t)b * H + y) * wpr + wi] : 0u;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NMS_S1_ROWS * wpr; i += 256) {
        const int r = i / wpr, wi = i - r * wpr, y = y0 + r;
        if (y >= H) continue;
        const int64_t wo = ((int64_t)b * H + y) * wpr + wi;
        unsigned u = s_u[(r + R) * rw + wi + 1], stay = 0u, keep = 0u;
        while (u) {
            const int bit = __ffs((int)u) - 1;
            u &= u - 1;
            const int x = wi * 32 + bit;
            if (nms_any_in_window<RT>(s_k, rw, r + R, x, R, tab)) continue;          // suppressed
            if (SUP_ONLY) { stay |= 1u << bit; continue; }
            const float sc = pb[(int64_t)y * W + x];
            bool blocked = false;
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                const int dy = k - (NW >> 1);
                unsigned long long v = (RT > 0 || (dy >= -R && dy <= R)) ? (nms_window(s_u + (r + R + dy) * rw, x, R) & tab.win[dy < 0 ? -dy : dy]) : 0ull;
                if (dy == 0) v &= ~(1ull << R);
                if (blocked) v = 0ull;
                while (v) {
                    const int j = __ffsll((long long)v) - 1;
                    v &= v - 1;
                    const int xx = x - R + j;
                    const float sq = pb[(int64_t)(y + dy) * W + xx];
                    if (sq > sc || (sq == sc && (dy < 0 || (dy == 0 && xx < x)))) { blocked = true; break; }
                }
            }
            if (blocked) stay |= 1u << bit;
            else { keep |= 1u << bit; ob[(int64_t)y * W + x] = sc; }
        }
        und_out[wo] = stay;
        kept_out[wo] = s_k[(r + R) * rw + wi + 1] | keep;
    }
}

// ---- pass B: the finisher.  LDS: two padded bit masks ((H + 2R) rows of wpr + 2 words) and the list of undecided positions in what is left.
// An image is handled by `bands` workgroups of `brows` rows each (1 band = the whole image: the fixed point completes inside the launch).  With more
// bands a workgroup sees the rows of its neighbours (reach R above and below) as they were when the launch started and cannot decide a candidate that
// waits on one of them: such launches are repeated (ping-pong masks) until a launch leaves nothing undecided — rare chains across a band border.
struct NmsFinishPlan { int wpr, rows, rw, bands, brows; size_t mask_bytes, list_off; int list_cap; };
__host__ __device__ inline NmsFinishPlan nms_finish_plan(int H, int W, int R, size_t lds_budget) {
    NmsFinishPlan p;
    p.wpr = (W + 31) / 32; p.rw = p.wpr + 2;
    const size_t per_row = (size_t)p.rw * 4 * 2;                         // both masks
    const size_t want = lds_budget * 5 / 8;                               // keep 3/8 of the budget for the (position, score) list
    int bands = 1;                                                        // as few as the LDS allows (more per image was measured: no gain, profiles/r5_nms_bands.txt)
    while (bands < 64 && (size_t)((H + bands - 1) / bands + 2 * R) * per_row > want) ++bands;
    p.bands = bands; p.brows = (H + bands - 1) / bands;
    p.rows = p.brows + 2 * R;
    p.mask_bytes = (size_t)p.rows * p.rw * 4;
    p.list_off = (2 * p.mask_bytes + 15) / 16 * 16;
    p.list_cap = lds_budget > p.list_off ? (int)((lds_budget - p.list_off) / 8) : 0;        // (position, score) per undecided candidate
    return p;
}
constexpr int NMS_FIN_THREADS = 1024;
constexpr size_t NMS_FIN_LDS = 158 * 1024;

template <int RT>
__global__ __launch_bounds__(NMS_FIN_THREADS) void nms_finish_kernel(const float* __restrict__ prob, float* __restrict__ out,
                                                                       const unsigned* __restrict__ und32, const unsigned* __restrict__ kept32,
                                                                       unsigned* __restrict__ und_out, unsigned* __restrict__ kept_out,
                                                                       int* __restrict__ glist, int H, int W, NmsTable tab, int* __restrict__ counters,
                                                                       int sweep, int* __restrict__ info) {
    extern __shared__ __align__(16) unsigned char fin_lds[];
    __shared__ int s_n;
    const int R = RT > 0 ? RT : tab.reach;
    constexpr int NW = RT > 0 ? 2 * RT + 1 : 2 * NMS_MAXR + 1;
    const NmsFinishPlan pl = nms_finish_plan(H, W, R, NMS_FIN_LDS);
    unsigned* mU = reinterpret_cast<unsigned*>(fin_lds);                          // undecided (bits cleared as decisions fall)
    unsigned* mK = reinterpret_cast<unsigned*>(fin_lds + pl.mask_bytes);          // kept
    if (sweep > 0 && counters[sweep - 1] == 0) return;                            // banded form: an earlier launch already left nothing undecided
    const int b = blockIdx.y, band = blockIdx.x, tid = threadIdx.x;
    const float* pb = prob + (int64_t)b * H * W;
    float* ob = out + (int64_t)b * H * W;
    const int wpr = pl.wpr, rw = pl.rw;
    const int yb0 = band * pl.brows, yb1 = min(H, yb0 + pl.brows);                // owned rows; LDS row r <-> image row yb0 - R + r
    const int nwords = (yb1 - yb0) * wpr;
    const int64_t img_words = (int64_t)b * H * wpr;
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
    auto stamp = [&](int slot) { if (tid == 0 && info && band == 0) info[8 * b + slot] = (int)(__builtin_amdgcn_s_memrealtime() - t_start); };     // 100 MHz ticks (diagnostics)
    for (int i = tid; i < pl.rows * rw; i += NMS_FIN_THREADS) { mU[i] = 0u; mK[i] = 0u; }
    if (tid == 0) s_n = 0;
    __syncthreads();
    // halo rows (reach R above and below the band, as the launch found them): read-only neighbours
    if (pl.bands > 1) {
        for (int i = tid; i < 2 * R * wpr; i += NMS_FIN_THREADS) {
            const int h = i / wpr, wi = i - h * wpr;
            const int r = h < R ? h : (pl.rows - 2 * R + h);                      // LDS rows 0..R-1 and rows-R..rows-1
            const int y = yb0 - R + r;
            if (y >= 0 && y < H && !(y >= yb0 && y < yb1)) {
                mU[r * rw + wi + 1] = und32[img_words + (int64_t)y * wpr + wi];
                mK[r * rw + wi + 1] = kept32[img_words + (int64_t)y * wpr + wi];
            }
        }
    }
    // masks in; undecided positions into the list (LDS when it fits, else the caller's workspace — same code through a flat pointer)
    int my_count = 0;
    for (int i = tid; i < nwords; i += NMS_FIN_THREADS) {
        const int yl = i / wpr, wi = i - yl * wpr;
        const unsigned u = und32[img_words + (int64_t)(yb0 + yl) * wpr + wi];
        mU[(yl + R) * rw + wi + 1] = u;
        mK[(yl + R) * rw + wi + 1] = kept32[img_words + (int64_t)(yb0 + yl) * wpr + wi];
        my_count += __popc(u);
    }
    int my_base = my_count ? atomicAdd(&s_n, my_count) : 0;
    __syncthreads();
    const int total = s_n;
    const int cap = total <= pl.list_cap ? pl.list_cap : pl.brows * W;
    int* list = total <= pl.list_cap ? reinterpret_cast<int*>(fin_lds + pl.list_off) : glist + ((int64_t)b * H + yb0) * W * 2;
    float* lscore = reinterpret_cast<float*>(list + cap);
    for (int i = tid; i < nwords; i += NMS_FIN_THREADS) {
        const int yl = i / wpr, wi = i - yl * wpr;
        unsigned u = mU[(yl + R) * rw + wi + 1];
        while (u) {
            const int bit = __ffs((int)u) - 1;
            u &= u - 1;
            lscore[my_base] = pb[(int64_t)(yb0 + yl) * W + wi * 32 + bit];
            list[my_base++] = (yl << 16) | (wi * 32 + bit);                     // band-local row
        }
    }
    __syncthreads();
    if (tid == 0 && info) { info[8 * b] = total; info[8 * b + 1] = total <= pl.list_cap ? 0 : 1; }
    stamp(3);
    // Fixed point, asynchronous inside the workgroup.  Races between lanes are harmless by ordering: a lane that keeps q sets q's KEPT bit BEFORE
    // clearing its UNDECIDED bit, and a reader looks at UNDECIDED first and at KEPT afterwards, so it can never miss q in both.
    // The whole image runs on ONE CU, so instructions — not latency — are the budget (16 waves share 4 SIMDs): after every round the list is
    // compacted to the still-undecided entries (a wave is then busy with 64 live candidates, not with one), and a candidate's work is a loop over
    // the few undecided neighbours it really has, not an unrolled worst case.
    volatile unsigned* vU = mU;
    __shared__ int s_cnt[NMS_FIN_THREADS / 64];
    int n_live = total;
    for (int round = 0; round < 1000000; ++round) {
        // entries [0, n_live) are live; this thread's slice is contiguous within the round: e = tid + k * 1024
        int nkeep = 0, progress = 0;
        int kpos[8]; float ksc[8];                                   // survivors of this thread (at most 8 per round: slices beyond that stay in place, see below)
        const bool small = n_live <= 8 * NMS_FIN_THREADS;
        for (int e = tid; e < n_live; e += NMS_FIN_THREADS) {
            const int pos = list[e];
            const float sc = lscore[e];
            const int y = pos >> 16, x = pos & 0xffff;
            const int wofs = (y + R) * rw + (x >> 5) + 1;
            const unsigned mybit = 1u << (x & 31);
            bool live = (vU[wofs] & mybit) != 0u;
            if (live && nms_any_in_window<RT>(mK, rw, y + R, x, R, tab)) { atomicAnd(&mU[wofs], ~mybit); live = false; progress = 1; }
            if (live) {
                bool blocked = false;
#pragma unroll
                for (int k = 0; k < NW; ++k) {
                    const int dy = k - (NW >> 1);
                    unsigned long long v = (RT > 0 || (dy >= -R && dy <= R)) ? (nms_window(mU + (y + R + dy) * rw, x, R) & tab.win[dy < 0 ? -dy : dy]) : 0ull;
                    if (dy == 0) v &= ~(1ull << R);
                    if (blocked) v = 0ull;
                    while (v) {
                        const int j = __ffsll((long long)v) - 1;
                        v &= v - 1;
                        const int xx = x - R + j;
                        const float sq = pb[(int64_t)(yb0 + y + dy) * W + xx];      // L2-resident heat map (1.2 MB per image)
                        if (sq > sc || (sq == sc && (dy < 0 || (dy == 0 && xx < x)))) { blocked = true; break; }
                    }
                }
                if (!blocked) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                    if (nms_any_in_window<RT>(mK, rw, y + R, x, R, tab)) { atomicAnd(&mU[wofs], ~mybit); }      // the authoritative read
                    else {
                        atomicOr(&mK[wofs], mybit);
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                        atomicAnd(&mU[wofs], ~mybit);
                        ob[(int64_t)(yb0 + y) * W + x] = sc;
                    }
                    live = false;
                    progress = 1;
                }
            }
            if (live && small) {
                // static register slots: select chain instead of a dynamically indexed array
#pragma unroll
                for (int q = 0; q < 8; ++q) if (q == nkeep) { kpos[q] = pos; ksc[q] = sc; }
                ++nkeep;
            } else if (live) {
                nkeep = 1;                                           // large lists are not compacted (they shrink below the limit within a round or two)
            }
        }
        // banded form: a round without any decision means the rest waits on a neighbouring band — leave it to the next launch.  (One band: the
        // undecided candidate of highest priority always decides, so every round makes progress.)
        const int any_progress = __syncthreads_or(progress);
        if (!small) {
            if (!__syncthreads_or(nkeep)) { n_live = 0; break; }
            if (!any_progress) break;
            continue;
        }
        // compaction: exclusive scan of the per-thread survivor counts (wave ballots + a scan of the 16 wave totals)
        const int lane = tid & 63, wave = tid >> 6;
        int incl = nkeep;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (lane == 63) s_cnt[wave] = incl;
        __syncthreads();                                             // (also: every read of list / lscore of this round is done)
        int base = 0, tot = 0;
#pragma unroll
        for (int wv = 0; wv < NMS_FIN_THREADS / 64; ++wv) { const int c = s_cnt[wv]; if (wv < wave) base += c; tot += c; }
        base += incl - nkeep;
#pragma unroll
        for (int q = 0; q < 8; ++q) if (q < nkeep) { list[base + q] = kpos[q]; lscore[base + q] = ksc[q]; }
        __syncthreads();
        n_live = tot;
        if (tid == 0 && info && band == 0) info[8 * b + 2] = round + 1;
        if (n_live == 0 || !any_progress) break;
    }
    if (pl.bands > 1) {
        // the band's rows of the next masks; a band that leaves something undecided asks for another launch
        __syncthreads();
        int left = 0;
        for (int i = tid; i < nwords; i += NMS_FIN_THREADS) {
            const int yl = i / wpr, wi = i - yl * wpr;
            const unsigned u = mU[(yl + R) * rw + wi + 1];
            und_out[img_words + (int64_t)(yb0 + yl) * wpr + wi] = u;
            kept_out[img_words + (int64_t)(yb0 + yl) * wpr + wi] = mK[(yl + R) * rw + wi + 1];
            left |= (u != 0u);
        }
        if (__syncthreads_or(left) && tid == 0) atomicAdd(&counters[sweep], 1);
    } else if (tid < NMS_MAX_SWEEPS && b == 0) {
        counters[tid] = 0;                                           // xp_box_nms_check: one band always converges inside the launch
    }

    stamp(7);
}

// Counter housekeeping as kernels, not hipMemsetAsync / hipMemcpyAsync: replayed from several hipGraphs on several streams
// (PairPipeline.capture with overlap) the API memset / copy nodes left the counters holding stale bytes on ROCm 7.2, kernel
// nodes do not.
__global__ void nms_reset_counters_kernel(int* __restrict__ counters) { if (threadIdx.x < NMS_MAX_SWEEPS) counters[threadIdx.x] = 0; }
__global__ void nms_publish_counter_kernel(int* __restrict__ counters, int from) { if (threadIdx.x == 0) counters[NMS_MAX_SWEEPS - 1] = counters[from]; }

// keep_top_k: rank of every survivor among the survivors of its image under (score desc, index asc);
// survivors with rank >= k are zeroed (utils.py:179-186: first k of the score-sorted survivors).
// Two launches: every rank is computed before any score is zeroed.
__global__ __launch_bounds__(256) void topk_rank_kernel(const float* __restrict__ out, const int* __restrict__ kp,
                                                        const int* __restrict__ counts, int* __restrict__ rank_out, int cap, int W,
                                                        int64_t HW) {
    __shared__ float s_sc[256];
    __shared__ int s_ix[256];
    const int b = blockIdx.y;
    const int n = min(counts[b], cap);
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float* ob = out + (int64_t)b * HW;
    const int* kb = kp + (int64_t)b * cap * 2;
    float sc = 0.f; int ix = 0;
    if (i < n) { ix = kb[2 * i] * W + kb[2 * i + 1]; sc = ob[ix]; }
    int rank = 0;
    for (int j0 = 0; j0 < n; j0 += 256) {
        const int j = j0 + threadIdx.x;
        if (j < n) { const int jx = kb[2 * j] * W + kb[2 * j + 1]; s_ix[threadIdx.x] = jx; s_sc[threadIdx.x] = ob[jx]; }
        __syncthreads();
        const int m = min(256, n - j0);
        if (i < n)
            for (int t = 0; t < m; ++t) rank += (s_sc[t] > sc || (s_sc[t] == sc && s_ix[t] < ix));
        __syncthreads();
    }
    if (i < n) rank_out[(int64_t)b * cap + i] = rank;
}

__global__ __launch_bounds__(256) void topk_zero_kernel(float* __restrict__ out, const int* __restrict__ kp, const int* __restrict__ counts,
                                                        const int* __restrict__ rank, int cap, int W, int64_t HW, int k) {
    const int b = blockIdx.y;
    const int n = min(counts[b], cap);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && rank[(int64_t)b * cap + i] >= k) {
        const int* kb = kp + (int64_t)b * cap * 2;
        out[(int64_t)b * HW + kb[2 * i] * W + kb[2 * i + 1]] = 0.f;
    }
}

// ---------------------------------------------------------------------------------------------
// Keypoint extraction: row-major (y, x) list of pixels with prob > thr (and mask != 0), per image — the order of
// torch.nonzero, which defines the match index space.  Ordered compaction in three stream-ordered launches:
// per-segment counts (1024 pixels per workgroup), exclusive scan of the segment counts per image, ordered write.
// ---------------------------------------------------------------------------------------------
constexpr int KP_SEG = 1024;

__device__ __forceinline__ bool kp_hit(const float* __restrict__ pb, const uint8_t* __restrict__ mb, int64_t i, int64_t HW, float thr) {
    return i < HW && pb[i] > thr && (!mb || mb[i]);
}

__global__ __launch_bounds__(KP_SEG) void kp_count_kernel(const float* __restrict__ prob, const uint8_t* __restrict__ mask, float thr,
                                                           int* __restrict__ seg_counts, int64_t HW, int nseg) {
    __shared__ int s_wave[KP_SEG / 64];
    const int b = blockIdx.y, seg = blockIdx.x;
    const int64_t i = (int64_t)seg * KP_SEG + threadIdx.x;
    const bool hit = kp_hit(prob + b * HW, mask ? mask + b * HW : nullptr, i, HW, thr);
    const unsigned long long bal = __ballot(hit);
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < KP_SEG / 64; ++w) t += s_wave[w]; seg_counts[b * nseg + seg] = t; }
}

// one workgroup per image: exclusive scan of the segment counts in place; total -> counts[b]
__global__ __launch_bounds__(1024) void kp_scan_kernel(int* __restrict__ seg_counts, int* __restrict__ counts, int nseg) {
    __shared__ int s_part[1024];
    const int b = blockIdx.x;
    int* sc = seg_counts + b * nseg;
    const int per = (nseg + 1023) / 1024;
    const int j0 = threadIdx.x * per, j1 = min(nseg, j0 + per);
    int t = 0;
    for (int j = j0; j < j1; ++j) t += sc[j];
    s_part[threadIdx.x] = t;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {          // Hillis-Steele inclusive scan over the 1024 partials
        const int v = threadIdx.x >= o ? s_part[threadIdx.x - o] : 0;
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = threadIdx.x ? s_part[threadIdx.x - 1] : 0;
    for (int j = j0; j < j1; ++j) { const int c = sc[j]; sc[j] = run; run += c; }
    if (threadIdx.x == 1023) counts[b] = s_part[1023];   // may exceed cap: the caller checks
}

__global__ __launch_bounds__(KP_SEG) void kp_write_kernel(const float* __restrict__ prob, const uint8_t* __restrict__ mask, float thr,
                                                           const int* __restrict__ seg_offsets, int* __restrict__ kp, int64_t HW, int W,
                                                           int nseg, int cap) {
    __shared__ int s_wave[KP_SEG / 64];
    const int b = blockIdx.y, seg = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)seg * KP_SEG + threadIdx.x;
    const bool hit = kp_hit(prob + b * HW, mask ? mask + b * HW : nullptr, i, HW, thr);
    const unsigned long long bal = __ballot(hit);
    if (lane == 0) s_wave[wave] = __popcll(bal);
    __syncthreads();
    if (!hit) return;
    int pos = seg_offsets[b * nseg + seg] + __popcll(bal & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) pos += s_wave[w];
    if (pos < cap) { int* kb = kp + ((int64_t)b * cap + pos) * 2; kb[0] = (int)(i / W); kb[1] = (int)(i % W); }
}

// Four pixels per thread (one 16-byte load; H * W % 4 == 0, 16-byte aligned maps): a workgroup covers 4 * KP_SEG pixels.  Same lists, same order: the
// rank of pixel j of lane l inside its wave = hits of the lanes before l (four ballots) + hits of l's own earlier pixels.
__device__ __forceinline__ unsigned kp_hit4(const float* __restrict__ pb, const uint8_t* __restrict__ mb, int64_t i, int64_t HW, float thr) {
    if (i >= HW) return 0u;
    const float4 v = *reinterpret_cast<const float4*>(pb + i);
    unsigned h = (v.x > thr ? 1u : 0u) | (v.y > thr ? 2u : 0u) | (v.z > thr ? 4u : 0u) | (v.w > thr ? 8u : 0u);
    if (mb) {
        const unsigned m = *reinterpret_cast<const unsigned*>(mb + i);
        h &= ((m & 0xffu) ? 1u : 0u) | ((m & 0xff00u) ? 2u : 0u) | ((m & 0xff0000u) ? 4u : 0u) | ((m & 0xff000000u) ? 8u : 0u);
    }
    return h;
}
__global__ __launch_bounds__(KP_SEG) void kp_count4_kernel(const float* __restrict__ prob, const uint8_t* __restrict__ mask, float thr,
                                                            int* __restrict__ seg_counts, int64_t HW, int nseg) {
    __shared__ int s_wave[KP_SEG / 64];
    const int b = blockIdx.y, seg = blockIdx.x;
    const int64_t i = ((int64_t)seg * KP_SEG + threadIdx.x) * 4;
    const unsigned h = kp_hit4(prob + b * HW, mask ? mask + b * HW : nullptr, i, HW, thr);
    const int c = __popcll(__ballot(h & 1u)) + __popcll(__ballot(h & 2u)) + __popcll(__ballot(h & 4u)) + __popcll(__ballot(h & 8u));
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < KP_SEG / 64; ++w) t += s_wave[w]; seg_counts[b * nseg + seg] = t; }
}
__global__ __launch_bounds__(KP_SEG) void kp_write4_kernel(const float* __restrict__ prob, const uint8_t* __restrict__ mask, float thr,
                                                            const int* __restrict__ seg_offsets, int* __restrict__ kp, int64_t HW, int W,
                                                            int nseg, int cap) {
    __shared__ int s_wave[KP_SEG / 64];
    const int b = blockIdx.y, seg = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = ((int64_t)seg * KP_SEG + threadIdx.x) * 4;
    const unsigned h = kp_hit4(prob + b * HW, mask ? mask + b * HW : nullptr, i, HW, thr);
    const unsigned long long before = (1ull << lane) - 1ull;
    const unsigned long long b0 = __ballot(h & 1u), b1 = __ballot(h & 2u), b2 = __ballot(h & 4u), b3 = __ballot(h & 8u);
    if (lane == 0) s_wave[wave] = __popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3);
    __syncthreads();
    if (!h) return;
    int pos = seg_offsets[b * nseg + seg] + __popcll(b0 & before) + __popcll(b1 & before) + __popcll(b2 & before) + __popcll(b3 & before);
    for (int w = 0; w < wave; ++w) pos += s_wave[w];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if ((h >> j) & 1u) {
            if (pos < cap) { int* kb = kp + ((int64_t)b * cap + pos) * 2; kb[0] = (int)((i + j) / W); kb[1] = (int)((i + j) % W); }
            ++pos;
        }
}

// ---------------------------------------------------------------------------------------------
// Descriptor sampling at keypoints: bilinear grid_sample (zeros padding, align_corners=True) of the
// NHWC descriptor volume followed by L2 normalisation, in ATen's fp32 operation order so the result
// matches torch to rounding:  g = k / (S*0.5) - 1;  i = (g + 1) * ((Sc - 1) / 2);  weights
// nw=(ixe-ix)(iye-iy), ne=(ix-ixw)(iye-iy), sw=(ixe-ix)(iy-iyn), se=(ix-ixw)(iy-iyn); sum nw,ne,sw,se.
// One wave per keypoint, lanes over the descriptor dimension.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sample_descriptors_kernel(const int* __restrict__ kp, const int* __restrict__ counts,
                                                                 const float* __restrict__ desc, float* __restrict__ out, int cap,
                                                                 int Hc, int Wc, int D, int H, int W) {
    const int b = blockIdx.y;
    const int n = min(counts[b], cap);
    // XCD-aware block order: consecutive workgroup ids go round-robin to the 8 XCDs, each with its own L2.  Keypoints arrive in raster
    // order and neighbours share descriptor rows (4 cells per keypoint, ~8 keypoints per cell row and column), so every XCD gets ONE
    // contiguous run of the image's keypoints (sized from the actual count, not the capacity) instead of every eighth group of four:
    // PMC reads 216 -> MB per 16 images for a 79 MB descriptor volume.
    int blk = blockIdx.x;
    if ((gridDim.x & 7) == 0) {
        const int per = ((n + 3) / 4 + 7) >> 3;              // 4-keypoint groups per XCD
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        if (slot >= per) return;
        blk = xcd * per + slot;
    }
    const int i = blk * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int lane = threadIdx.x & 63;
    const int* kb = kp + ((int64_t)b * cap + i) * 2;
    const float ky = (float)kb[0], kx = (float)kb[1];
    const float gy = ky / ((float)H * 0.5f) - 1.0f, gx = kx / ((float)W * 0.5f) - 1.0f;
    const float iy = (gy + 1.f) * ((float)(Hc - 1) / 2.f), ix = (gx + 1.f) * ((float)(Wc - 1) / 2.f);   // ATen CPU: (g + 1) * ((size-1)/2)
    const float iyn = floorf(iy), ixw = floorf(ix);
    const float iys = iyn + 1.f, ixe = ixw + 1.f;
    const float nw = (ixe - ix) * (iys - iy), ne = (ix - ixw) * (iys - iy);
    const float sw = (ixe - ix) * (iy - iyn), se = (ix - ixw) * (iy - iyn);
    const int y0 = (int)iyn, x0 = (int)ixw, y1 = y0 + 1, x1 = x0 + 1;
    const bool vy0 = y0 >= 0 && y0 < Hc, vy1 = y1 >= 0 && y1 < Hc, vx0 = x0 >= 0 && x0 < Wc, vx1 = x1 >= 0 && x1 < Wc;
    const float* db = desc + (int64_t)b * Hc * Wc * D;
    if (D == 256) {
        // the configured descriptor size: a lane owns 4 consecutive channels, a cell's row is ONE 16-byte-per-lane load (4 loads per keypoint
        // instead of 16); products and sums in the order of the general path below (nw, ne, sw, se), so results are identical
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 a = (vy0 && vx0) ? reinterpret_cast<const float4*>(db + ((int64_t)y0 * Wc + x0) * 256)[lane] : z;
        const float4 bq = (vy0 && vx1) ? reinterpret_cast<const float4*>(db + ((int64_t)y0 * Wc + x1) * 256)[lane] : z;
        const float4 cq = (vy1 && vx0) ? reinterpret_cast<const float4*>(db + ((int64_t)y1 * Wc + x0) * 256)[lane] : z;
        const float4 dq = (vy1 && vx1) ? reinterpret_cast<const float4*>(db + ((int64_t)y1 * Wc + x1) * 256)[lane] : z;
        auto mix = [&](float p0, float p1, float p2, float p3) {
            float o = 0.f;
            if (vy0 && vx0) o = p0 * nw;
            if (vy0 && vx1) o += p1 * ne;
            if (vy1 && vx0) o += p2 * sw;
            if (vy1 && vx1) o += p3 * se;
            return o;
        };
        float4 o4;
        o4.x = mix(a.x, bq.x, cq.x, dq.x); o4.y = mix(a.y, bq.y, cq.y, dq.y); o4.z = mix(a.z, bq.z, cq.z, dq.z); o4.w = mix(a.w, bq.w, cq.w, dq.w);
        // the general path sums the squares of channels lane, lane + 64, ... per lane; here a lane holds 4 neighbours — a different (equally
        // valid) association of the same 256 squares, so the norm may differ in the last bit
        float ss4 = fmaf(o4.x, o4.x, 0.f); ss4 = fmaf(o4.y, o4.y, ss4); ss4 = fmaf(o4.z, o4.z, ss4); ss4 = fmaf(o4.w, o4.w, ss4);
        const float nrm4 = fmaxf(sqrtf(xp_wave_sum(ss4)), 1e-12f);
        float4 r4; r4.x = o4.x / nrm4; r4.y = o4.y / nrm4; r4.z = o4.z / nrm4; r4.w = o4.w / nrm4;
        reinterpret_cast<float4*>(out + ((int64_t)b * cap + i) * 256)[lane] = r4;
        return;
    }
    float v[8];
    float ss = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int c = lane + 64 * t;
        float o = 0.f;
        if (c < D) {
            if (vy0 && vx0) o = db[((int64_t)y0 * Wc + x0) * D + c] * nw;
            if (vy0 && vx1) o += db[((int64_t)y0 * Wc + x1) * D + c] * ne;
            if (vy1 && vx0) o += db[((int64_t)y1 * Wc + x0) * D + c] * sw;
            if (vy1 && vx1) o += db[((int64_t)y1 * Wc + x1) * D + c] * se;
        }
        v[t] = o;
        ss = fmaf(o, o, ss);
    }
    const float nrm = fmaxf(sqrtf(xp_wave_sum(ss)), 1e-12f);
    float* ob = out + ((int64_t)b * cap + i) * D;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int c = lane + 64 * t;
        if (c < D) ob[c] = v[t] / nrm;
    }
}

bool make_nms_table(float size, float iou, NmsTable* t) {
    // torchvision float32 arithmetic on boxes [c - size/2, c + size/2]
    if (!(size > 0.f) || size > (float)(NMS_MAXR + 1) || (2.f * size) != floorf(2.f * size)) return false;
    const float half = size * 0.5f;
    const float area = ((0.f + half) - (0.f - half)) * ((0.f + half) - (0.f - half));
    uint32_t ovl[NMS_MAXR + 1];
    int reach = 0;
    for (int dy = 0; dy <= NMS_MAXR; ++dy) {
        uint32_t bits = 0;
        for (int dx = 0; dx <= NMS_MAXR; ++dx) {
            const float ay1 = -half, ay2 = half, ax1 = -half, ax2 = half;
            const float by1 = (float)dy - half, by2 = (float)dy + half, bx1 = (float)dx - half, bx2 = (float)dx + half;
            float w = fminf(ay2, by2) - fmaxf(ay1, by1); if (w < 0.f) w = 0.f;
            float h = fminf(ax2, bx2) - fmaxf(ax1, bx1); if (h < 0.f) h = 0.f;
            const float inter = w * h;
            const float ovr = inter / (area + area - inter);
            if (ovr > iou) { bits |= (1u << dx); if (dy > reach) reach = dy; if (dx > reach) reach = dx; }
        }
        ovl[dy] = bits;
    }
    t->reach = reach;
    for (int dy = 0; dy <= NMS_MAXR; ++dy) {
        unsigned long long m = 0;
        for (int dx = -reach; dx <= reach; ++dx)
            if ((ovl[dy] >> (dx < 0 ? -dx : dx)) & 1u) m |= 1ull << (reach + dx);
        t->win[dy] = m;
    }
    return true;
}

}  // namespace

namespace {
struct NmsWs { uint8_t* state; uint8_t* tile_active; int* counters; int* kp; float* gtab; int* info; unsigned* cand32; unsigned* kept32; };
size_t nms_ws_tail_offset(int batch, int H, int W, int cap);
NmsWs nms_ws(void* workspace, int batch, int H, int W, int cap = 0) {
    const size_t n = ((size_t)batch * H * W + 255) / 256 * 256;
    const size_t nt = ((size_t)batch * xp_cdiv(H, NMS_TH) * xp_cdiv(W, NMS_TILE) + 255) / 256 * 256;
    NmsWs w;
    w.state = (uint8_t*)workspace;
    w.tile_active = w.state + n;
    w.counters = (int*)(w.tile_active + nt);
    w.kp = w.counters + NMS_MAX_SWEEPS;
    // round-3 form: the bit masks live in the (otherwise unused) state bytes: 2 * H * ceil(W / 32) words per image <= H * W bytes for W >= 32
    w.cand32 = (unsigned*)w.state;                          // four mask arrays of batch * H * ceil(W / 32) words: (undecided, kept) x ping-pong
    w.kept32 = w.cand32 + (size_t)batch * H * xp_cdiv(W, 32);
    char* tail = (char*)workspace + nms_ws_tail_offset(batch, H, W, cap);
    w.gtab = (float*)tail;                                  // undecided-list overflow area: (position, score) per pixel (used only when an image's list outgrows LDS)
    w.info = (int*)(tail + 2 * sizeof(float) * (size_t)batch * H * W);
    return w;
}
}  // namespace

namespace {
size_t nms_ws_tail_offset(int batch, int H, int W, int cap) {
    const size_t n = ((size_t)batch * H * W + 255) / 256 * 256;
    const size_t nt = ((size_t)batch * xp_cdiv(H, NMS_TH) * xp_cdiv(W, NMS_TILE) + 255) / 256 * 256;
    const size_t head = n + nt + sizeof(int) * (NMS_MAX_SWEEPS + (size_t)batch * cap * 3 + batch + 64) + xp_extract_keypoints_workspace_bytes(batch, H, W);
    return (head + 255) / 256 * 256;
}
// the two-launch form applies when an image's padded masks, rank prefix and a minimal score table fit one workgroup's LDS, a thread owns at
// most NMS_FIN_MAXW mask words and the mask words fit the state bytes
bool nms_two_launch_applies(int H, int W, int R) {
    static const bool force_sweeps = getenv("XP_NMS_SWEEP") != nullptr && atoi(getenv("XP_NMS_SWEEP")) != 0;     // A/B: the round-1/2 sweep form
    if (force_sweeps || W < 32) return false;
    const NmsFinishPlan pl = nms_finish_plan(H, W, R, NMS_FIN_LDS);
    return pl.wpr <= 64 && H < 65536 && W < 65536 && pl.list_cap >= 2048 && pl.bands <= 32;
}
}  // namespace

extern "C" size_t xp_box_nms_workspace_bytes(int batch, int H, int W, int cap) {
    // state bytes (sweep form) / bit masks (two-launch form) + tile flags + counters + (top-k) keypoint list, counts, ranks + score-table overflow + info
    return nms_ws_tail_offset(batch, H, W, cap) + 2 * sizeof(float) * (size_t)batch * H * W + sizeof(int) * (8 * (size_t)batch + 64);
}

// Enqueue `sweeps` sweep launches (each exits at once when the previous one left nothing undecided).
// band_sweeps: launches of the finisher when an image is split into row bands (ignored for one band)
static int box_nms_two_launch(const float* prob, float* out, void* workspace, int batch, int H, int W, int cap, const NmsTable& tab, float min_prob,
                              int band_sweeps, hipStream_t s) {
    const NmsWs w = nms_ws(workspace, batch, H, W, cap);
    XpProfScope prof("box_nms", s, 0.0, 8.0 * (double)batch * H * W);   // 8*H*W bytes per image
    static XpPerDeviceOnce attr_once;
    if (attr_once.need()) {
        XP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(nms_finish_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)NMS_FIN_LDS));
        XP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(nms_finish_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)NMS_FIN_LDS));
    }
    const int wpr = xp_cdiv(W, 32);
    if (tab.reach == 6) hipLaunchKernelGGL(nms_localmax_kernel<6>, dim3(wpr, xp_cdiv(H, NMS_TH), batch), dim3(256), 0, s, prob, out, w.cand32, w.kept32, H, W, wpr, tab, min_prob);
    else hipLaunchKernelGGL(nms_localmax_kernel<0>, dim3(wpr, xp_cdiv(H, NMS_TH), batch), dim3(256), 0, s, prob, out, w.cand32, w.kept32, H, W, wpr, tab, min_prob);
    // wide Jacobi rounds (ping-pong masks), then the per-image finisher for what is left; the count only moves work between the two, never the result
    static const int wide_rounds = getenv("XP_NMS_WIDE_ROUNDS") ? atoi(getenv("XP_NMS_WIDE_ROUNDS")) : 0;
    const size_t mw = (size_t)batch * H * wpr;
    unsigned* um[2] = {w.cand32, w.cand32 + 2 * mw};
    unsigned* km[2] = {w.kept32, w.kept32 + 2 * mw};
    int cur = 0;
    const dim3 rgrid(xp_cdiv(H, NMS_S1_ROWS), batch);
    if (tab.reach == 6) hipLaunchKernelGGL((nms_round_kernel<6, true>), rgrid, dim3(256), 0, s, prob, out, um[cur], km[cur], um[cur ^ 1], km[cur ^ 1], H, W, wpr, tab);
    else hipLaunchKernelGGL((nms_round_kernel<0, true>), rgrid, dim3(256), 0, s, prob, out, um[cur], km[cur], um[cur ^ 1], km[cur ^ 1], H, W, wpr, tab);
    cur ^= 1;
    for (int r = 0; r < wide_rounds; ++r, cur ^= 1) {
        if (tab.reach == 6) hipLaunchKernelGGL((nms_round_kernel<6, false>), rgrid, dim3(256), 0, s, prob, out, um[cur], km[cur], um[cur ^ 1], km[cur ^ 1], H, W, wpr, tab);
        else hipLaunchKernelGGL((nms_round_kernel<0, false>), rgrid, dim3(256), 0, s, prob, out, um[cur], km[cur], um[cur ^ 1], km[cur ^ 1], H, W, wpr, tab);
    }
    const NmsFinishPlan pl = nms_finish_plan(H, W, tab.reach, NMS_FIN_LDS);
    const int launches = pl.bands > 1 ? (band_sweeps > 0 ? band_sweeps : 1) : 1;
    if (pl.bands > 1) hipLaunchKernelGGL(nms_reset_counters_kernel, dim3(1), dim3(64), 0, s, w.counters);
    for (int sw = 0; sw < launches; ++sw, cur ^= 1) {
        if (tab.reach == 6) hipLaunchKernelGGL(nms_finish_kernel<6>, dim3(pl.bands, batch), dim3(NMS_FIN_THREADS), NMS_FIN_LDS, s, prob, out, um[cur], km[cur], um[cur ^ 1], km[cur ^ 1], (int*)w.gtab, H, W, tab, w.counters, sw, w.info);
        else hipLaunchKernelGGL(nms_finish_kernel<0>, dim3(pl.bands, batch), dim3(NMS_FIN_THREADS), NMS_FIN_LDS, s, prob, out, um[cur], km[cur], um[cur ^ 1], km[cur ^ 1], (int*)w.gtab, H, W, tab, w.counters, sw, w.info);
    }
    if (pl.bands > 1 && launches != NMS_MAX_SWEEPS) hipLaunchKernelGGL(nms_publish_counter_kernel, dim3(1), dim3(64), 0, s, w.counters, launches - 1);
    XP_LAUNCH_CHECK();
    return XP_OK;
}

static int box_nms_enqueue(const float* prob, float* out, void* workspace, int batch, int H, int W, const NmsTable& tab,
                           float min_prob, int sweeps, bool first_round, hipStream_t s) {
    const NmsWs w = nms_ws(workspace, batch, H, W);
    XpProfScope prof("box_nms", s, 0.0, 8.0 * (double)batch * H * W);   // 8*H*W bytes per image
    hipLaunchKernelGGL(nms_reset_counters_kernel, dim3(1), dim3(64), 0, s, w.counters);
    dim3 grid(xp_cdiv(W, NMS_TILE), xp_cdiv(H, NMS_TH), batch);
    // Local fixed-point iterations per sweep.  Early sweeps are dominated by decisions that wait on a neighbouring tile, so
    // iterating long inside a tile is wasted there (measured: 8 iterations in sweep 0 cost 256 us, 2 cost 128 us, and the
    // number of sweeps to convergence is the same); late sweeps touch few tiles and finish them locally.
    // XP_NMS_SCHED="a,b,c,..." overrides (tuning).
    static int sched[NMS_MAX_SWEEPS];
    static bool sched_init = false;
    if (!sched_init) {
        for (int i = 0; i < NMS_MAX_SWEEPS; ++i) sched[i] = i < 3 ? 2 : 8;
        if (const char* e = getenv("XP_NMS_SCHED")) {
            int i = 0, last = 8;
            for (const char* q = e; *q && i < NMS_MAX_SWEEPS;) { last = atoi(q); sched[i++] = last; while (*q && *q != ',') ++q; if (*q == ',') ++q; }
            for (; i < NMS_MAX_SWEEPS; ++i) sched[i] = last;
        }
        sched_init = true;
    }
    for (int i = 0; i < sweeps; ++i)
        hipLaunchKernelGGL(nms_sweep_kernel, grid, dim3(256), 0, s, prob, w.state, out, w.tile_active, H, W, tab, min_prob, w.counters,
                           i, (first_round && i == 0) ? 1 : 0, sched[i]);
    XP_LAUNCH_CHECK();
    return XP_OK;
}

extern "C" size_t xp_extract_keypoints_workspace_bytes(int batch, int H, int W);
extern "C" int xp_extract_keypoints(const float* prob, const uint8_t* mask, float thr, int* kp, int* counts, int batch,
                                    int H, int W, int cap, void* workspace, size_t workspace_bytes, void* stream);

extern "C" int xp_box_nms(const float* prob, float* out, void* workspace, size_t workspace_bytes, int batch, int H, int W,
                          float size, float min_prob, float iou, int keep_top_k, int cap, int max_sweeps_async,
                          int* converged_host, void* stream) {
    XP_CHECK_ARG(prob && out && workspace, "xp_box_nms: null pointer");
    XP_CHECK_ARG(batch > 0 && H > 0 && W > 0, "xp_box_nms: bad shape");
    XP_CHECK_ARG(workspace_bytes >= xp_box_nms_workspace_bytes(batch, H, W, cap), "xp_box_nms: workspace too small");
    NmsTable tab;
    XP_CHECK_ARG(make_nms_table(size, iou, &tab), "xp_box_nms: size must be a positive multiple of 0.5 and <= %d (got %f)", NMS_MAXR + 1, size);
    hipStream_t s = (hipStream_t)stream;
    XP_CHECK_ARG(max_sweeps_async <= NMS_MAX_SWEEPS, "xp_box_nms: at most %d async sweeps", NMS_MAX_SWEEPS);
    const NmsWs w = nms_ws(workspace, batch, H, W, cap);
    if (nms_two_launch_applies(H, W, tab.reach)) {
        // exact, no sweep count when the image is one band (see nms_localmax_kernel): the same in the stream-ordered and in the synchronous mode.
        // Images split into row bands repeat the finisher (launches past convergence exit at once): max_sweeps_async of them stream-ordered, or —
        // synchronous mode — as many as there are counter slots, checked here.
        int rc = box_nms_two_launch(prob, out, workspace, batch, H, W, cap, tab, min_prob, max_sweeps_async > 0 ? max_sweeps_async : NMS_MAX_SWEEPS, s);
        if (rc) return rc;
        if (max_sweeps_async <= 0) {
            int left = 0;
            XP_HIP(hipMemcpyAsync(&left, w.counters + NMS_MAX_SWEEPS - 1, sizeof(int), hipMemcpyDeviceToHost, s));
            XP_HIP(hipStreamSynchronize(s));
            if (left != 0) { xp_set_error("xp_box_nms: %d row bands still undecided after %d finisher launches", left, NMS_MAX_SWEEPS); return XP_ERR_STATE; }
            if (converged_host) *converged_host = 1;
        }
    } else if (max_sweeps_async > 0) {
        // fire-and-forget: the caller checks the last counter later (xp_box_nms_check)
        int rc = box_nms_enqueue(prob, out, workspace, batch, H, W, tab, min_prob, max_sweeps_async, true, s);
        if (rc) return rc;
        if (max_sweeps_async != NMS_MAX_SWEEPS)
            hipLaunchKernelGGL(nms_publish_counter_kernel, dim3(1), dim3(64), 0, s, w.counters, max_sweeps_async - 1);
    } else {
        bool first = true;
        const int per_round = 4;
        for (int round = 0; round < 100000; ++round) {
            int rc = box_nms_enqueue(prob, out, workspace, batch, H, W, tab, min_prob, per_round, first, s);
            if (rc) return rc;
            first = false;
            int left = 0;
            XP_HIP(hipMemcpyAsync(&left, w.counters + per_round - 1, sizeof(int), hipMemcpyDeviceToHost, s));
            XP_HIP(hipStreamSynchronize(s));
            if (left == 0) break;
        }
        if (converged_host) *converged_host = 1;
    }
    if (keep_top_k > 0) {
        XP_CHECK_ARG(cap > 0, "xp_box_nms: keep_top_k needs cap > 0");
        int* kp = w.kp;
        int* counts = kp + (size_t)batch * cap * 2;
        int* rank = counts + batch;
        void* ews = rank + (size_t)batch * cap;
        int rc = xp_extract_keypoints(out, nullptr, 0.f, kp, counts, batch, H, W, cap, ews, xp_extract_keypoints_workspace_bytes(batch, H, W), stream);
        if (rc) return rc;
        dim3 grid(xp_cdiv(cap, 256), batch);
        hipLaunchKernelGGL(topk_rank_kernel, grid, dim3(256), 0, s, out, kp, counts, rank, cap, W, (int64_t)H * W);
        hipLaunchKernelGGL(topk_zero_kernel, grid, dim3(256), 0, s, out, kp, counts, rank, cap, W, (int64_t)H * W, keep_top_k);
        XP_LAUNCH_CHECK();
    }
    return XP_OK;
}

// After the stream has been synchronised: did the async NMS converge?  (device->host copy of one int)
extern "C" int xp_box_nms_check(const void* workspace, int batch, int H, int W, int* undecided_tiles, void* stream) {
    XP_CHECK_ARG(workspace && undecided_tiles, "xp_box_nms_check: null pointer");
    const NmsWs w = nms_ws(const_cast<void*>(workspace), batch, H, W);
    XP_HIP(hipMemcpyAsync(undecided_tiles, w.counters + NMS_MAX_SWEEPS - 1, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    XP_HIP(hipStreamSynchronize((hipStream_t)stream));
    return XP_OK;
}

extern "C" size_t xp_extract_keypoints_workspace_bytes(int batch, int H, int W) {
    return sizeof(int) * (size_t)batch * (size_t)xp_cdiv((int64_t)H * W, KP_SEG) + 256;
}

extern "C" int xp_extract_keypoints(const float* prob, const uint8_t* mask, float thr, int* kp, int* counts, int batch,
                                    int H, int W, int cap, void* workspace, size_t workspace_bytes, void* stream) {
    XP_CHECK_ARG(prob && kp && counts && workspace, "xp_extract_keypoints: null pointer");
    XP_CHECK_ARG(batch > 0 && cap > 0, "xp_extract_keypoints: bad batch/cap");
    XP_CHECK_ARG(workspace_bytes >= xp_extract_keypoints_workspace_bytes(batch, H, W), "xp_extract_keypoints: workspace too small");
    const int64_t HW = (int64_t)H * W;
    int* seg = (int*)workspace;
    hipStream_t s = (hipStream_t)stream;
    XpProfScope prof("extract_keypoints", s, 0.0, 8.0 * batch * H * W);
    const bool vec4 = HW % 4 == 0 && (((uintptr_t)prob & 15) == 0) && (!mask || ((uintptr_t)mask & 3) == 0);
    const int nseg = xp_cdiv(HW, vec4 ? 4 * KP_SEG : KP_SEG);
    if (vec4) hipLaunchKernelGGL(kp_count4_kernel, dim3(nseg, batch), dim3(KP_SEG), 0, s, prob, mask, thr, seg, HW, nseg);
    else hipLaunchKernelGGL(kp_count_kernel, dim3(nseg, batch), dim3(KP_SEG), 0, s, prob, mask, thr, seg, HW, nseg);
    hipLaunchKernelGGL(kp_scan_kernel, dim3(batch), dim3(1024), 0, s, seg, counts, nseg);
    if (vec4) hipLaunchKernelGGL(kp_write4_kernel, dim3(nseg, batch), dim3(KP_SEG), 0, s, prob, mask, thr, seg, kp, HW, W, nseg, cap);
    else hipLaunchKernelGGL(kp_write_kernel, dim3(nseg, batch), dim3(KP_SEG), 0, s, prob, mask, thr, seg, kp, HW, W, nseg, cap);
    XP_LAUNCH_CHECK();
    return XP_OK;
}

extern "C" int xp_sample_descriptors(const int* kp, const int* counts, const float* desc_nhwc, float* out, int batch,
                                     int cap, int Hc, int Wc, int D, int H, int W, void* stream) {
    XP_CHECK_ARG(kp && counts && desc_nhwc && out, "xp_sample_descriptors: null pointer");
    XP_CHECK_ARG(D > 0 && D <= 512, "xp_sample_descriptors: D must be in [1,512] (got %d)", D);
    dim3 grid(xp_cdiv(cap, 4), batch);
    XpProfScope prof("sample_descriptors", (hipStream_t)stream, 0.0, 0.0);
    hipLaunchKernelGGL(sample_descriptors_kernel, grid, dim3(256), 0, (hipStream_t)stream, kp, counts, desc_nhwc, out, cap, Hc, Wc, D, H, W);
    XP_LAUNCH_CHECK();
    return XP_OK;
}
