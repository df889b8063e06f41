// Detection post-processing on the heat-map: box NMS, keypoint extraction, descriptor sampling.
//   box_nms            reference xpoint/utils/utils.py:148-192  (torchvision.ops.nms / batched_nms)
//   extract_keypoints  reference predict_align_image_pair.py:242-243, predict_keypoints.py:213-215
//   sample_descriptors reference xpoint/utils/utils.py:229-238  (F.grid_sample bilinear, align_corners)
#include "xp_common.h"

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
constexpr int NMS_TILE = 32;
constexpr int NMS_MAXR = 15;

struct NmsTable { uint32_t ovl[NMS_MAXR + 1]; int reach; };

enum : int { ST_NONE = 0, ST_UNDECIDED = 1, ST_KEPT = 2 };

__global__ __launch_bounds__(256) void nms_init_kernel(const float* __restrict__ prob, uint8_t* __restrict__ state, int64_t n,
                                                       float min_prob, int* __restrict__ counters) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { counters[0] = 1; counters[1] = 0; }   // [0] = undecided seen by the previous sweep, [1] = accumulator
    if (i < n) state[i] = prob[i] > min_prob ? ST_UNDECIDED : ST_NONE;
}

__global__ __launch_bounds__(256) void nms_sweep_kernel(const float* __restrict__ prob, uint8_t* __restrict__ state, int H, int W,
                                                        NmsTable tab, int* __restrict__ counters, int local_iters) {
    if (counters[0] == 0) return;   // converged in an earlier launch (uniform early exit)
    __shared__ float s_sc[(NMS_TILE + 2 * NMS_MAXR) * (NMS_TILE + 2 * NMS_MAXR)];
    __shared__ int s_st[(NMS_TILE + 2 * NMS_MAXR) * (NMS_TILE + 2 * NMS_MAXR)];
    __shared__ int s_flag[2];
    const int R = tab.reach;
    const int TW = NMS_TILE + 2 * R;
    const int b = blockIdx.z;
    const int y0 = blockIdx.y * NMS_TILE - R, x0 = blockIdx.x * NMS_TILE - R;
    const float* pb = prob + (int64_t)b * H * W;
    uint8_t* sb = state + (int64_t)b * H * W;
    if (threadIdx.x < 2) s_flag[threadIdx.x] = 0;
    __syncthreads();
    int any = 0;
    for (int i = threadIdx.x; i < TW * TW; i += 256) {
        const int ty = i / TW, tx = i - ty * TW;
        const int y = y0 + ty, x = x0 + tx;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;
        const int st = in ? sb[(int64_t)y * W + x] : ST_NONE;
        s_st[i] = st;
        s_sc[i] = in ? pb[(int64_t)y * W + x] : 0.f;
        const bool own = ty >= R && ty < R + NMS_TILE && tx >= R && tx < R + NMS_TILE;
        any |= (own && st == ST_UNDECIDED);
    }
    if (any) s_flag[0] = 1;
    __syncthreads();
    if (!s_flag[0]) return;
    for (int it = 0; it < local_iters; ++it) {
        if (threadIdx.x == 0) s_flag[1] = 0;
        __syncthreads();
        for (int o = threadIdx.x; o < NMS_TILE * NMS_TILE; o += 256) {
            const int ty = R + o / NMS_TILE, tx = R + o % NMS_TILE;
            const int c = ty * TW + tx;
            if (s_st[c] != ST_UNDECIDED) continue;
            const float sc = s_sc[c];
            bool suppressed = false, blocked = false;
            for (int dy = -R; dy <= R && !suppressed; ++dy) {
                const uint32_t row = tab.ovl[dy < 0 ? -dy : dy];
                for (int dx = -R; dx <= R; ++dx) {
                    if (!((row >> (dx < 0 ? -dx : dx)) & 1u) || (dy == 0 && dx == 0)) continue;
                    const int q = c + dy * TW + dx;
                    const int st = s_st[q];
                    if (st == ST_KEPT) { suppressed = true; break; }
                    if (st == ST_UNDECIDED) {
                        const float sq = s_sc[q];
                        // q has higher priority: larger score, or equal score and earlier row-major index
                        if (sq > sc || (sq == sc && (dy < 0 || (dy == 0 && dx < 0)))) blocked = true;
                    }
                }
            }
            if (suppressed) { s_st[c] = ST_NONE; s_flag[1] = 1; }
            else if (!blocked) { s_st[c] = ST_KEPT; s_flag[1] = 1; }
        }
        __syncthreads();
        if (!s_flag[1]) break;
        __syncthreads();
    }
    int left = 0;
    for (int o = threadIdx.x; o < NMS_TILE * NMS_TILE; o += 256) {
        const int ty = R + o / NMS_TILE, tx = R + o % NMS_TILE;
        const int y = y0 + ty, x = x0 + tx;
        if (y < H && x < W) {
            const int st = s_st[ty * TW + tx];
            sb[(int64_t)y * W + x] = (uint8_t)st;
            left |= (st == ST_UNDECIDED);
        }
    }
    if (left) atomicAdd(&counters[1], 1);
}

__global__ void nms_roll_counter_kernel(int* counters) { counters[0] = counters[1]; counters[1] = 0; }

__global__ __launch_bounds__(256) void nms_write_kernel(const float* __restrict__ prob, const uint8_t* __restrict__ state,
                                                        float* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = state[i] == ST_KEPT ? prob[i] : 0.f;
}

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
// Keypoint extraction: row-major (y, x) list of pixels with prob > thr (and mask != 0), per image.
// One 1024-thread workgroup per image walks the image in order with a running offset, so the list
// order equals torch.nonzero's — it defines the match index space.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void extract_keypoints_kernel(const float* __restrict__ prob, const uint8_t* __restrict__ mask,
                                                                 float thr, int* __restrict__ kp, int* __restrict__ counts,
                                                                 int H, int W, int cap) {
    __shared__ int s_wave[16];
    __shared__ int s_base;
    const int b = blockIdx.x;
    const int64_t HW = (int64_t)H * W;
    const float* pb = prob + b * HW;
    const uint8_t* mb = mask ? mask + b * HW : nullptr;
    int* kb = kp + (int64_t)b * cap * 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (int64_t i0 = 0; i0 < HW; i0 += 1024) {
        const int64_t i = i0 + threadIdx.x;
        bool hit = false;
        if (i < HW) hit = (pb[i] > thr) && (!mb || mb[i]);
        const unsigned long long bal = __ballot(hit);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wave] = __popcll(bal);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; ++w) off += s_wave[w];
        if (hit) {
            const int pos = off + before;
            if (pos < cap) { kb[2 * pos] = (int)(i / W); kb[2 * pos + 1] = (int)(i % W); }
        }
        __syncthreads();
        if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += s_wave[w]; s_base += t; }
        __syncthreads();
    }
    if (threadIdx.x == 0) counts[b] = s_base;   // may exceed cap: the caller checks
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
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
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
        t->ovl[dy] = bits;
    }
    t->reach = reach;
    return true;
}

}  // namespace

extern "C" size_t xp_box_nms_workspace_bytes(int batch, int H, int W, int cap) {
    // state bytes + counters + (top-k) keypoint list, counts, ranks
    size_t n = (size_t)batch * H * W;
    n = (n + 255) / 256 * 256;
    return n + 256 + sizeof(int) * ((size_t)batch * cap * 3 + batch + 64);
}

// Enqueue-only form: runs `sweeps` sweep launches (each exits immediately once converged).  After the
// stream is synchronised, workspace[ state_bytes .. ] holds counters; *converged is NOT written here.
static int box_nms_enqueue(const float* prob, float* out, void* workspace, int batch, int H, int W, const NmsTable& tab,
                           float min_prob, int sweeps, bool init, hipStream_t s) {
    const int64_t n = (int64_t)batch * H * W;
    uint8_t* state = (uint8_t*)workspace;
    int* counters = (int*)((char*)workspace + ((n + 255) / 256 * 256));
    XpProfScope prof("box_nms", s, 0.0, 8.0 * n);   // 8*H*W bytes per image (read prob, write prob_nms)
    if (init) hipLaunchKernelGGL(nms_init_kernel, dim3(xp_cdiv(n, 256)), dim3(256), 0, s, prob, state, n, min_prob, counters);
    dim3 grid(xp_cdiv(W, NMS_TILE), xp_cdiv(H, NMS_TILE), batch);
    for (int i = 0; i < sweeps; ++i) {
        hipLaunchKernelGGL(nms_sweep_kernel, grid, dim3(256), 0, s, prob, state, H, W, tab, counters, 8);
        hipLaunchKernelGGL(nms_roll_counter_kernel, dim3(1), dim3(1), 0, s, counters);
    }
    hipLaunchKernelGGL(nms_write_kernel, dim3(xp_cdiv(n, 256)), dim3(256), 0, s, prob, state, out, n);
    XP_LAUNCH_CHECK();
    return XP_OK;
}

extern "C" int xp_extract_keypoints(const float* prob, const uint8_t* mask, float thr, int* kp, int* counts, int batch,
                                    int H, int W, int cap, void* stream);

extern "C" int xp_box_nms(const float* prob, float* out, void* workspace, size_t workspace_bytes, int batch, int H, int W,
                          float size, float min_prob, float iou, int keep_top_k, int cap, int max_sweeps_async,
                          int* converged_host, void* stream) {
    XP_CHECK_ARG(prob && out && workspace, "xp_box_nms: null pointer");
    XP_CHECK_ARG(batch > 0 && H > 0 && W > 0, "xp_box_nms: bad shape");
    XP_CHECK_ARG(workspace_bytes >= xp_box_nms_workspace_bytes(batch, H, W, cap), "xp_box_nms: workspace too small");
    NmsTable tab;
    XP_CHECK_ARG(make_nms_table(size, iou, &tab), "xp_box_nms: size must be a positive multiple of 0.5 and <= %d (got %f)", NMS_MAXR + 1, size);
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = (int64_t)batch * H * W;
    int* counters = (int*)((char*)workspace + ((n + 255) / 256 * 256));
    if (max_sweeps_async > 0) {
        // fire-and-forget: the caller checks counters later (xp_box_nms_check)
        int rc = box_nms_enqueue(prob, out, workspace, batch, H, W, tab, min_prob, max_sweeps_async, true, s);
        if (rc) return rc;
    } else {
        bool init = true;
        for (int round = 0; round < 4096; ++round) {
            int rc = box_nms_enqueue(prob, out, workspace, batch, H, W, tab, min_prob, 4, init, s);
            if (rc) return rc;
            init = false;
            int left = 0;
            XP_HIP(hipMemcpyAsync(&left, counters, sizeof(int), hipMemcpyDeviceToHost, s));
            XP_HIP(hipStreamSynchronize(s));
            if (left == 0) break;
        }
        if (converged_host) *converged_host = 1;
    }
    if (keep_top_k > 0) {
        XP_CHECK_ARG(cap > 0, "xp_box_nms: keep_top_k needs cap > 0");
        int* kp = counters + 64;
        int* counts = kp + (size_t)batch * cap * 2;
        int* rank = counts + batch;
        int rc = xp_extract_keypoints(out, nullptr, 0.f, kp, counts, batch, H, W, cap, stream);
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
    const int64_t n = (int64_t)batch * H * W;
    const int* counters = (const int*)((const char*)workspace + ((n + 255) / 256 * 256));
    XP_HIP(hipMemcpyAsync(undecided_tiles, counters, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    XP_HIP(hipStreamSynchronize((hipStream_t)stream));
    return XP_OK;
}

extern "C" int xp_extract_keypoints(const float* prob, const uint8_t* mask, float thr, int* kp, int* counts, int batch,
                                    int H, int W, int cap, void* stream) {
    XP_CHECK_ARG(prob && kp && counts, "xp_extract_keypoints: null pointer");
    XP_CHECK_ARG(batch > 0 && cap > 0, "xp_extract_keypoints: bad batch/cap");
    XpProfScope prof("extract_keypoints", (hipStream_t)stream, 0.0, 4.0 * batch * H * W);
    hipLaunchKernelGGL(extract_keypoints_kernel, dim3(batch), dim3(1024), 0, (hipStream_t)stream, prob, mask, thr, kp, counts, H, W, cap);
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
