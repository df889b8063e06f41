// Robust homography from point correspondences, batched over image pairs — the device-side stand-in for
//   cv2.findHomography(optical_pts, thermal_pts, cv2.USAC_MAGSAC, ransacReprojThreshold, confidence=0.9999, maxIters=10000)
// as called by the reference (predict_align_image_pair.py:291-303, xpoint/utils/benchmark_evaluation.py:796-812).
// SURVEY.md 8(f) rank 2.  OpenCV is absent here and its estimator is randomised, so this is NOT bit-comparable with it:
// the contract kept is "3x3 H (h33 = 1) mapping src -> dst, inlier mask at the reprojection threshold, no model below
// four correspondences".  Deterministic: hypotheses are drawn by a counter-based hash of (seed, pair, hypothesis).
//
// Algorithm: Hartley-normalised 4-point DLT hypotheses scored with the MSAC loss sum_i min(err_i^2, thr^2) over ALL
// correspondences (forward reprojection error in pixels, the quantity ransacReprojThreshold bounds); the best hypothesis
// is refined by three rounds of inlier selection + normalised linear least squares on the inliers (8 x 8 normal
// equations in f64).  10 000 hypotheses x ~1 200 correspondences x ~25 flops = 0.3 GFLOP per pair: a latency-sized job,
// one thread per hypothesis, correspondences streamed through LDS.
#include "xp_common.h"

namespace {

struct Norm { double sx, sy, s, dx, dy, d; };     // src: (x - sx) * s ; dst: (u - dx) * d

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// Solve the 8 x 8 system M h = r in place (Gaussian elimination, partial pivoting).  Returns false if singular.
__device__ bool solve8(double (&M)[8][9]) {
    for (int c = 0; c < 8; ++c) {
        int piv = c; double best = fabs(M[c][c]);
        for (int r = c + 1; r < 8; ++r) { const double v = fabs(M[r][c]); if (v > best) { best = v; piv = r; } }
        if (best < 1e-12) return false;
        if (piv != c) for (int k = c; k < 9; ++k) { const double t = M[c][k]; M[c][k] = M[piv][k]; M[piv][k] = t; }
        const double inv = 1.0 / M[c][c];
        for (int r = c + 1; r < 8; ++r) {
            const double f = M[r][c] * inv;
            if (f != 0.0) for (int k = c; k < 9; ++k) M[r][k] -= f * M[c][k];
        }
    }
    for (int c = 7; c >= 0; --c) {
        double v = M[c][8];
        for (int k = c + 1; k < 8; ++k) v -= M[c][k] * M[k][8];
        M[c][8] = v / M[c][c];
    }
    return true;
}

// normalised-frame h (8 values, h33 = 1) -> pixel-frame H (9 values, not yet scaled)
__device__ void denormalise(const double (&hn)[8], const Norm& nm, double (&H)[9]) {
    // H = Td^-1 * Hn * Ts,  Ts = [s 0 -s*sx; 0 s -s*sy; 0 0 1],  Td^-1 = [1/d 0 dx; 0 1/d dy; 0 0 1]
    const double a[9] = {hn[0], hn[1], hn[2], hn[3], hn[4], hn[5], hn[6], hn[7], 1.0};
    double b[9];
    for (int r = 0; r < 3; ++r) {
        b[r * 3 + 0] = a[r * 3 + 0] * nm.s;
        b[r * 3 + 1] = a[r * 3 + 1] * nm.s;
        b[r * 3 + 2] = -a[r * 3 + 0] * nm.s * nm.sx - a[r * 3 + 1] * nm.s * nm.sy + a[r * 3 + 2];
    }
    const double id = 1.0 / nm.d;
    for (int c = 0; c < 3; ++c) {
        H[c] = b[c] * id + nm.dx * b[6 + c];
        H[3 + c] = b[3 + c] * id + nm.dy * b[6 + c];
        H[6 + c] = b[6 + c];
    }
}

__device__ __forceinline__ float reproj_err2(const double (&H)[9], float x, float y, float u, float v) {
    const double w = H[6] * x + H[7] * y + H[8];
    const double iw = fabs(w) > 1e-12 ? 1.0 / w : 0.0;
    const double pu = (H[0] * x + H[1] * y + H[2]) * iw, pv = (H[3] * x + H[4] * y + H[5]) * iw;
    const double du = pu - u, dv = pv - v;
    return (float)(du * du + dv * dv);
}

__device__ __forceinline__ int count_of(const int* counts, int pair, int cap) { const int n = counts ? counts[pair] : cap; return n < cap ? (n < 0 ? 0 : n) : cap; }

// per pair: centroid and mean distance of both point sets (Hartley normalisation), best-score slot reset
__global__ __launch_bounds__(256) void hg_prepare_kernel(const float* __restrict__ src, const float* __restrict__ dst, const int* __restrict__ counts,
                                                         int cap, Norm* __restrict__ norms, unsigned long long* __restrict__ best) {
    __shared__ double s_red[4][4];
    const int pair = blockIdx.x, n = count_of(counts, pair, cap);
    const float* s = src + (size_t)pair * cap * 2; const float* d = dst + (size_t)pair * cap * 2;
    double acc[4] = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < n; i += 256) { acc[0] += s[2 * i]; acc[1] += s[2 * i + 1]; acc[2] += d[2 * i]; acc[3] += d[2 * i + 1]; }
    auto block_sum4 = [&](double (&a)[4]) {
        for (int k = 0; k < 4; ++k) { for (int o = 32; o > 0; o >>= 1) a[k] += __shfl_xor(a[k], o, 64); }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) for (int k = 0; k < 4; ++k) s_red[threadIdx.x >> 6][k] = a[k];
        __syncthreads();
        for (int k = 0; k < 4; ++k) a[k] = s_red[0][k] + s_red[1][k] + s_red[2][k] + s_red[3][k];
    };
    block_sum4(acc);
    const double inv = n > 0 ? 1.0 / n : 0.0;
    const double sx = acc[0] * inv, sy = acc[1] * inv, dx = acc[2] * inv, dy = acc[3] * inv;
    double dist[4] = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < n; i += 256) {
        const double a = s[2 * i] - sx, b = s[2 * i + 1] - sy, c = d[2 * i] - dx, e = d[2 * i + 1] - dy;
        dist[0] += sqrt(a * a + b * b); dist[1] += sqrt(c * c + e * e);
    }
    block_sum4(dist);
    if (threadIdx.x == 0) {
        Norm nm;
        nm.sx = sx; nm.sy = sy; nm.dx = dx; nm.dy = dy;
        const double ms = dist[0] * inv, md = dist[1] * inv;
        nm.s = ms > 1e-9 ? 1.4142135623730951 / ms : 1.0;
        nm.d = md > 1e-9 ? 1.4142135623730951 / md : 1.0;
        norms[pair] = nm;
        best[pair] = ~0ull;
    }
}

// 4-point DLT in the normalised frame for the correspondences idx[0..3]; false if degenerate
__device__ bool fit4(const float* __restrict__ s, const float* __restrict__ d, const Norm& nm, const int (&idx)[4], double (&hn)[8]) {
    double M[8][9];
    for (int k = 0; k < 4; ++k) {
        const double x = (s[2 * idx[k]] - nm.sx) * nm.s, y = (s[2 * idx[k] + 1] - nm.sy) * nm.s;
        const double u = (d[2 * idx[k]] - nm.dx) * nm.d, v = (d[2 * idx[k] + 1] - nm.dy) * nm.d;
        double* r0 = M[2 * k]; double* r1 = M[2 * k + 1];
        r0[0] = x; r0[1] = y; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -u * x; r0[7] = -u * y; r0[8] = u;
        r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = x; r1[4] = y; r1[5] = 1; r1[6] = -v * x; r1[7] = -v * y; r1[8] = v;
    }
    if (!solve8(M)) return false;
    for (int k = 0; k < 8; ++k) hn[k] = M[k][8];
    return true;
}

__device__ void sample4(unsigned seed, int pair, int it, int n, int (&idx)[4]) {
    unsigned ctr = hash32(seed ^ hash32((unsigned)pair * 0x9e3779b9u + 0x85ebca6bu) ^ hash32((unsigned)it + 0x27d4eb2fu));
    for (int k = 0; k < 4; ++k) {
        for (int tries = 0; tries < 64; ++tries) {
            ctr = hash32(ctr + 0x9e3779b9u);
            const int c = (int)(ctr % (unsigned)n);
            bool dup = false;
            for (int j = 0; j < k; ++j) dup |= (idx[j] == c);
            if (!dup) { idx[k] = c; break; }
            if (tries == 63) idx[k] = (k == 0 ? 0 : (idx[k - 1] + 1) % n);
        }
    }
}

// one thread per hypothesis; MSAC score over all correspondences (LDS tiles); best (score, hypothesis) per pair by atomicMin
__global__ __launch_bounds__(256) void hg_hypotheses_kernel(const float* __restrict__ src, const float* __restrict__ dst, const int* __restrict__ counts,
                                                            int cap, float thr2, int iters, unsigned seed, const Norm* __restrict__ norms,
                                                            unsigned long long* __restrict__ best) {
    __shared__ float s_pts[512][4];
    const int pair = blockIdx.y, n = count_of(counts, pair, cap);
    if (n < 4) return;
    const float* s = src + (size_t)pair * cap * 2; const float* d = dst + (size_t)pair * cap * 2;
    const Norm nm = norms[pair];
    const int it = blockIdx.x * 256 + threadIdx.x;
    bool ok = it < iters;
    double H[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (ok) {
        int idx[4]; double hn[8];
        sample4(seed, pair, it, n, idx);
        ok = fit4(s, d, nm, idx, hn);
        if (ok) denormalise(hn, nm, H);
    }
    float score = 0.f;
    for (int j0 = 0; j0 < n; j0 += 512) {
        const int m = min(512, n - j0);
        __syncthreads();
        for (int t = threadIdx.x; t < m; t += 256) { s_pts[t][0] = s[2 * (j0 + t)]; s_pts[t][1] = s[2 * (j0 + t) + 1]; s_pts[t][2] = d[2 * (j0 + t)]; s_pts[t][3] = d[2 * (j0 + t) + 1]; }
        __syncthreads();
        if (ok) for (int j = 0; j < m; ++j) score += fminf(reproj_err2(H, s_pts[j][0], s_pts[j][1], s_pts[j][2], s_pts[j][3]), thr2);
    }
    if (ok) atomicMin(&best[pair], ((unsigned long long)__float_as_uint(score) << 32) | (unsigned)it);
}

// one workgroup per pair: rebuild the best hypothesis, 3 x (inliers -> normalised least squares), final mask / count / H
__global__ __launch_bounds__(256) void hg_refine_kernel(const float* __restrict__ src, const float* __restrict__ dst, const int* __restrict__ counts,
                                                        int cap, float thr2, unsigned seed, const Norm* __restrict__ norms,
                                                        const unsigned long long* __restrict__ best, double* __restrict__ Hout,
                                                        uint8_t* __restrict__ mask, int* __restrict__ n_inl) {
    __shared__ double s_H[9];
    __shared__ double s_acc[4][44];
    __shared__ int s_cnt[4];
    __shared__ int s_ok;
    const int pair = blockIdx.x, n = count_of(counts, pair, cap);
    const float* s = src + (size_t)pair * cap * 2; const float* d = dst + (size_t)pair * cap * 2;
    uint8_t* mk = mask + (size_t)pair * cap;
    double* Ho = Hout + (size_t)pair * 9;
    const unsigned long long b = n >= 4 ? best[pair] : ~0ull;
    if (b == ~0ull) {     // fewer than 4 correspondences or no non-degenerate sample: no model
        for (int i = threadIdx.x; i < cap; i += 256) mk[i] = 0;
        if (threadIdx.x < 9) Ho[threadIdx.x] = (threadIdx.x % 4 == 0) ? 1.0 : 0.0;
        if (threadIdx.x == 0) n_inl[pair] = 0;
        return;
    }
    const Norm nm = norms[pair];
    if (threadIdx.x == 0) {
        int idx[4]; double hn[8], H[9];
        sample4(seed, pair, (int)(b & 0xffffffffu), n, idx);
        fit4(s, d, nm, idx, hn);
        denormalise(hn, nm, H);
        for (int k = 0; k < 9; ++k) s_H[k] = H[k];
    }
    __syncthreads();
    for (int round = 0; round < 3; ++round) {
        double H[9];
        for (int k = 0; k < 9; ++k) H[k] = s_H[k];
        // normal equations of the inliers in the normalised frame: unknowns h11..h32, two rows per correspondence
        double acc[44];
        for (int k = 0; k < 44; ++k) acc[k] = 0.0;
        int cnt = 0;
        for (int i = threadIdx.x; i < n; i += 256) {
            if (reproj_err2(H, s[2 * i], s[2 * i + 1], d[2 * i], d[2 * i + 1]) > thr2) continue;
            ++cnt;
            const double x = (s[2 * i] - nm.sx) * nm.s, y = (s[2 * i + 1] - nm.sy) * nm.s;
            const double u = (d[2 * i] - nm.dx) * nm.d, v = (d[2 * i + 1] - nm.dy) * nm.d;
            const double r0[8] = {x, y, 1, 0, 0, 0, -u * x, -u * y}, r1[8] = {0, 0, 0, x, y, 1, -v * x, -v * y};
            int q = 0;
            for (int a = 0; a < 8; ++a) for (int c = a; c < 8; ++c) acc[q++] += r0[a] * r0[c] + r1[a] * r1[c];
            for (int a = 0; a < 8; ++a) acc[36 + a] += r0[a] * u + r1[a] * v;
        }
        for (int k = 0; k < 44; ++k) for (int o = 32; o > 0; o >>= 1) acc[k] += __shfl_xor(acc[k], o, 64);
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) { for (int k = 0; k < 44; ++k) s_acc[threadIdx.x >> 6][k] = acc[k]; s_cnt[threadIdx.x >> 6] = cnt; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
            s_ok = 0;
            if (total >= 4) {
                double M[8][9];
                int q = 0;
                for (int a = 0; a < 8; ++a) for (int c = a; c < 8; ++c) { const double v = s_acc[0][q] + s_acc[1][q] + s_acc[2][q] + s_acc[3][q]; M[a][c] = v; M[c][a] = v; ++q; }
                for (int a = 0; a < 8; ++a) M[a][8] = s_acc[0][36 + a] + s_acc[1][36 + a] + s_acc[2][36 + a] + s_acc[3][36 + a];
                if (solve8(M)) {
                    double hn[8], Hn[9];
                    for (int k = 0; k < 8; ++k) hn[k] = M[k][8];
                    denormalise(hn, nm, Hn);
                    for (int k = 0; k < 9; ++k) s_H[k] = Hn[k];
                    s_ok = 1;
                }
            }
        }
        __syncthreads();
        if (!s_ok) break;      // keep the previous model
    }
    double H[9];
    for (int k = 0; k < 9; ++k) H[k] = s_H[k];
    int cnt = 0;
    for (int i = threadIdx.x; i < cap; i += 256) {
        const bool in = i < n && reproj_err2(H, s[2 * i], s[2 * i + 1], d[2 * i], d[2 * i + 1]) <= thr2;
        mk[i] = in ? 1 : 0;
        cnt += in;
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) n_inl[pair] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    const double sc = fabs(H[8]) > 1e-300 ? 1.0 / H[8] : 1.0;
    if (threadIdx.x < 9) Ho[threadIdx.x] = H[threadIdx.x] * sc;
}

// correspondences of the mutual matches: src = optical keypoint (x, y) of match_q, dst = thermal keypoint (x, y) of match_t
__global__ __launch_bounds__(256) void gather_match_points_kernel(const int* __restrict__ kp, const int* __restrict__ mq, const int* __restrict__ mt,
                                                                  const int* __restrict__ mcount, int pairs, int cap, float* __restrict__ src,
                                                                  float* __restrict__ dst) {
    const int pair = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cap) return;
    const int n = min(mcount[pair], cap);
    float2 a = make_float2(0.f, 0.f), b = a;
    if (i < n) {
        const int q = mq[(size_t)pair * cap + i], t = mt[(size_t)pair * cap + i];
        const int* ko = kp + ((size_t)pair * cap + q) * 2;                  // optical image `pair`
        const int* kt = kp + ((size_t)(pairs + pair) * cap + t) * 2;        // thermal image `pairs + pair`
        a = make_float2((float)ko[1], (float)ko[0]);                        // (y, x) -> (x, y), as cv2.KeyPoint(c[1], c[0], 1)
        b = make_float2((float)kt[1], (float)kt[0]);
    }
    reinterpret_cast<float2*>(src)[(size_t)pair * cap + i] = a;
    reinterpret_cast<float2*>(dst)[(size_t)pair * cap + i] = b;
}

}  // namespace

// kp (2*pairs, cap, 2) int32 (y, x): optical images first, then thermal (PairPipeline layout); match_q / match_t (pairs, cap),
// match_count (pairs) from xp_match_mnn -> src / dst (pairs, cap, 2) f32 (x, y) for xp_find_homography (counts = match_count).
extern "C" int xp_gather_match_points(const int* kp, const int* match_q, const int* match_t, const int* match_count, int pairs, int cap,
                                      float* src, float* dst, void* stream) {
    XP_CHECK_ARG(kp && match_q && match_t && match_count && src && dst, "xp_gather_match_points: null pointer");
    XP_CHECK_ARG(pairs > 0 && cap > 0, "xp_gather_match_points: bad shape");
    XpProfScope prof("gather_match_points", (hipStream_t)stream, 0.0, 0.0);
    hipLaunchKernelGGL(gather_match_points_kernel, dim3(xp_cdiv(cap, 256), pairs), dim3(256), 0, (hipStream_t)stream, kp, match_q, match_t,
                       match_count, pairs, cap, src, dst);
    XP_LAUNCH_CHECK();
    return XP_OK;
}

extern "C" size_t xp_find_homography_workspace_bytes(int pairs) { return pairs > 0 ? (size_t)pairs * (sizeof(Norm) + 8) + 64 : 0; }

// src / dst: (pairs, cap, 2) float32 (x, y) device arrays; counts (pairs) device int32 or NULL (= cap correspondences each).
// H (pairs, 9) f64 row-major with h33 = 1; mask (pairs, cap) uint8; n_inliers (pairs) int32 (0 = no model, H = identity).
extern "C" int xp_find_homography(const float* src, const float* dst, const int* counts, int pairs, int cap, float reproj_thr,
                                  int max_iters, unsigned seed, double* H, uint8_t* mask, int* n_inliers, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    XP_CHECK_ARG(src && dst && H && mask && n_inliers && workspace, "xp_find_homography: null pointer");
    XP_CHECK_ARG(pairs > 0 && cap > 0 && max_iters > 0 && reproj_thr > 0.f, "xp_find_homography: bad arguments");
    XP_CHECK_ARG(workspace_bytes >= xp_find_homography_workspace_bytes(pairs), "xp_find_homography: workspace too small");
    XP_CHECK_ARG(((uintptr_t)workspace & 7) == 0, "xp_find_homography: workspace must be 8-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    Norm* norms = (Norm*)workspace;
    unsigned long long* best = (unsigned long long*)(norms + pairs);
    const float thr2 = reproj_thr * reproj_thr;
    XpProfScope prof("find_homography", s, 0.0, 0.0);
    hipLaunchKernelGGL(hg_prepare_kernel, dim3(pairs), dim3(256), 0, s, src, dst, counts, cap, norms, best);
    hipLaunchKernelGGL(hg_hypotheses_kernel, dim3(xp_cdiv(max_iters, 256), pairs), dim3(256), 0, s, src, dst, counts, cap, thr2, max_iters, seed, norms, best);
    hipLaunchKernelGGL(hg_refine_kernel, dim3(pairs), dim3(256), 0, s, src, dst, counts, cap, thr2, seed, norms, best, H, mask, n_inliers);
    XP_LAUNCH_CHECK();
    return XP_OK;
}
