/* ORACLE — TEST INFRASTRUCTURE ONLY.  Plain-C CPU restatement of the sequential / integer parts
 * of the XPoint hot path.  Never linked into or called from the product (xpoint_amd); only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load this library.
 *
 * Each function cites the reference file:line it restates.
 * Build: oracle/build.py  (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * Selective scan recurrence.  Reference: xpoint/models/vmamba_src/csms6s.py:56-67
 *     x = deltaA[:, :, i, :] * x + deltaB_u[:, :, i, :]          (mul then add, two roundings)
 *     y = einsum('bdn,bdn->bd', x, C[:, :, :, i])
 * The elementwise prologue (softplus, exp, products: csms6s.py:47-55) is done by the caller in
 * torch so that libm differences do not enter.  Layouts: deltaA, deltaBu (B, D, L, N); Cmat
 * (B, G, N, L) with D % G == 0; y (B, D, L); last_state (B, D, N) optional.
 * ---------------------------------------------------------------------------------------- */
void xo_scan_recurrence(const float* deltaA, const float* deltaBu, const float* Cmat, float* y,
                        float* last_state, int64_t Bn, int64_t D, int64_t L, int64_t N, int64_t G) {
    int64_t rows = Bn * D;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
        int64_t b = r / D, d = r % D;
        int64_t g = d / (D / G);
        const float* a = deltaA + r * L * N;
        const float* bu = deltaBu + r * L * N;
        const float* c = Cmat + (b * G + g) * N * L;
        float x[256];
        for (int64_t n = 0; n < N; ++n) x[n] = 0.f;
        for (int64_t l = 0; l < L; ++l) {
            float acc = 0.f;
            for (int64_t n = 0; n < N; ++n) {
                float t = a[l * N + n] * x[n];
                x[n] = t + bu[l * N + n];
                float p = x[n] * c[n * L + l];
                acc = (n == 0) ? p : acc + p;
            }
            y[r * L + l] = acc;
        }
        if (last_state)
            for (int64_t n = 0; n < N; ++n) last_state[r * N + n] = x[n];
    }
}

/* ------------------------------------------------------------------------------------------
 * Greedy box NMS on a heat-map.  Reference: xpoint/utils/utils.py:148-192 (box_nms) which calls
 * torchvision.ops.nms / batched_nms (third-party, absent here: documented algorithm — stable
 * descending sort by score, box i suppresses later box j when inter/(a_i+a_j-inter) > iou; boxes
 * are [c - size/2, c + size/2] in float32; batched_nms == independent per image).
 * prob (H,W) one image; out (H,W) zero-filled by the caller; returns number kept.
 * keep_order (optional, capacity >= #candidates) receives flat pixel indices of survivors in
 * decreasing-score order (what torchvision returns), used for keep_top_k.
 * ---------------------------------------------------------------------------------------- */
typedef struct { float s; int32_t idx; } xo_cand;

static int xo_cmp(const void* a, const void* b) {
    const xo_cand* x = (const xo_cand*)a; const xo_cand* y = (const xo_cand*)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx); /* stable: earlier (row-major) index first */
}

int64_t xo_box_nms(const float* prob, float* out, int32_t* keep_order, int64_t H, int64_t W,
                   float size, float min_prob, float iou, int64_t keep_top_k) {
    int64_t n = 0;
    for (int64_t i = 0; i < H * W; ++i) n += (prob[i] > min_prob);
    if (n == 0) return 0;
    xo_cand* c = (xo_cand*)malloc(sizeof(xo_cand) * n);
    int64_t k = 0;
    for (int64_t i = 0; i < H * W; ++i)
        if (prob[i] > min_prob) { c[k].s = prob[i]; c[k].idx = (int32_t)i; ++k; }
    qsort(c, n, sizeof(xo_cand), xo_cmp);
    uint8_t* kept = (uint8_t*)calloc(H * W, 1);
    int64_t reach = (int64_t)ceilf(size); /* boxes further apart than `size` cannot intersect */
    float half = size * 0.5f;
    int64_t nkept = 0;
    for (int64_t q = 0; q < n; ++q) {
        int64_t y = c[q].idx / W, x = c[q].idx % W;
        float bx1 = (float)y - half, by1 = (float)x - half, bx2 = (float)y + half, by2 = (float)x + half;
        float area_q = (bx2 - bx1) * (by2 - by1);
        int sup = 0;
        for (int64_t yy = y - reach; yy <= y + reach && !sup; ++yy) {
            if (yy < 0 || yy >= H) continue;
            for (int64_t xx = x - reach; xx <= x + reach; ++xx) {
                if (xx < 0 || xx >= W || !kept[yy * W + xx]) continue;
                float kx1 = (float)yy - half, ky1 = (float)xx - half, kx2 = (float)yy + half, ky2 = (float)xx + half;
                float area_k = (kx2 - kx1) * (ky2 - ky1);
                float ix1 = bx1 > kx1 ? bx1 : kx1, iy1 = by1 > ky1 ? by1 : ky1;
                float ix2 = bx2 < kx2 ? bx2 : kx2, iy2 = by2 < ky2 ? by2 : ky2;
                float w = ix2 - ix1; if (w < 0.f) w = 0.f;
                float h = iy2 - iy1; if (h < 0.f) h = 0.f;
                float inter = w * h;
                float ovr = inter / (area_k + area_q - inter);
                if (ovr > iou) { sup = 1; break; }
            }
        }
        if (!sup) {
            kept[c[q].idx] = 1;
            if (keep_top_k <= 0 || nkept < keep_top_k) out[c[q].idx] = c[q].s; /* utils.py:179-190 */
            if (keep_order) keep_order[nkept] = c[q].idx;
            ++nkept;
        }
    }
    free(kept); free(c);
    return nkept;
}

/* ------------------------------------------------------------------------------------------
 * Brute-force L2 nearest neighbours in both directions, direct form, double accumulation
 * (the "true" argmin; first minimum wins exact ties).  Reference call site:
 * xpoint/utils/matching.py:4-36 (cv2.BFMatcher(NORM_L2, crossCheck=True).match — third-party,
 * absent) and matching.py:38-75 (NNMatcher: argmin rows / argmin cols / mutual test).
 * d1 (n1,dim), d2 (n2,dim) float32.  idx12[q] = argmin_t, dist12[q]; idx21[t] = argmin_q.
 * gap12[q] (optional) = second-best minus best distance, for near-tie reports.
 * ---------------------------------------------------------------------------------------- */
void xo_nn_both(const float* d1, int64_t n1, const float* d2, int64_t n2, int64_t dim,
                int32_t* idx12, double* dist12, double* gap12, int32_t* idx21, double* dist21) {
    double* col_best = (double*)malloc(sizeof(double) * n2);
    for (int64_t t = 0; t < n2; ++t) { col_best[t] = INFINITY; idx21[t] = -1; }
#pragma omp parallel
    {
        double* lbest = (double*)malloc(sizeof(double) * n2);
        int32_t* lidx = (int32_t*)malloc(sizeof(int32_t) * n2);
        for (int64_t t = 0; t < n2; ++t) { lbest[t] = INFINITY; lidx[t] = -1; }
#pragma omp for schedule(static)
        for (int64_t q = 0; q < n1; ++q) {
            double best = INFINITY, second = INFINITY; int32_t bi = -1;
            const float* a = d1 + q * dim;
            for (int64_t t = 0; t < n2; ++t) {
                const float* b = d2 + t * dim;
                double s = 0.0;
                for (int64_t k = 0; k < dim; ++k) { double df = (double)a[k] - (double)b[k]; s += df * df; }
                if (s < best) { second = best; best = s; bi = (int32_t)t; }
                else if (s < second) second = s;
                if (s < lbest[t]) { lbest[t] = s; lidx[t] = (int32_t)q; } /* q ascending within a thread */
            }
            idx12[q] = bi; dist12[q] = sqrt(best);
            if (gap12) gap12[q] = sqrt(second) - sqrt(best);
        }
#pragma omp critical
        {
            for (int64_t t = 0; t < n2; ++t)
                if (lidx[t] >= 0 && (lbest[t] < col_best[t] || (lbest[t] == col_best[t] && lidx[t] < idx21[t]))) {
                    col_best[t] = lbest[t]; idx21[t] = lidx[t];
                }
        }
        free(lbest); free(lidx);
    }
    if (dist21) for (int64_t t = 0; t < n2; ++t) dist21[t] = sqrt(col_best[t]);
    free(col_best);
}

/* ------------------------------------------------------------------------------------------
 * k = 2 nearest targets of every query (fp64 direct form; ties -> lower index first): what
 * cv2.BFMatcher(NORM_L2).knnMatch(d1, d2, 2) returns per query, reference
 * xpoint/utils/matching.py:20-27 (knn_matches + Lowe ratio).  idx (n1,2) = -1 / dist = inf where n2 < 2.
 * ---------------------------------------------------------------------------------------- */
void xo_knn2(const float* d1, int64_t n1, const float* d2, int64_t n2, int64_t dim, int32_t* idx, double* dist) {
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < n1; ++q) {
        double b1 = INFINITY, b2 = INFINITY; int32_t i1 = -1, i2 = -1;
        const float* a = d1 + q * dim;
        for (int64_t t = 0; t < n2; ++t) {
            const float* b = d2 + t * dim;
            double s = 0.0;
            for (int64_t k = 0; k < dim; ++k) { double df = (double)a[k] - (double)b[k]; s += df * df; }
            if (s < b1) { b2 = b1; i2 = i1; b1 = s; i1 = (int32_t)t; }       /* t ascending: strict < keeps the lower index on ties */
            else if (s < b2) { b2 = s; i2 = (int32_t)t; }
        }
        idx[2 * q] = i1; idx[2 * q + 1] = i2; dist[2 * q] = sqrt(b1); dist[2 * q + 1] = sqrt(b2);
    }
}

/* ------------------------------------------------------------------------------------------
 * ThresholdMatcher (reference xpoint/utils/matching.py:77-102): every (q, t) with
 * sqrt(2 - 2 clip(<a_q, b_t>, -1, 1)) < threshold, row-major order.  fp64 dot products.
 * Returns the number of pairs; writes at most `cap` of them.
 * ---------------------------------------------------------------------------------------- */
int64_t xo_threshold_pairs(const float* d1, int64_t n1, const float* d2, int64_t n2, int64_t dim, double threshold,
                           int32_t* pairs, double* dist, int64_t cap) {
    int64_t n = 0;
    for (int64_t q = 0; q < n1; ++q) {
        const float* a = d1 + q * dim;
        for (int64_t t = 0; t < n2; ++t) {
            const float* b = d2 + t * dim;
            double s = 0.0;
            for (int64_t k = 0; k < dim; ++k) s += (double)a[k] * (double)b[k];
            if (s > 1.0) s = 1.0;
            if (s < -1.0) s = -1.0;
            const double d = sqrt(2.0 - 2.0 * s);
            if (d < threshold) { if (n < cap) { pairs[2 * n] = (int32_t)q; pairs[2 * n + 1] = (int32_t)t; dist[n] = d; } ++n; }
        }
    }
    return n;
}

/* ------------------------------------------------------------------------------------------
 * Perspective warp.  Reference call: predict_align_image_pair.py:308 (and demo.py:225-249)
 *     cv2.warpPerspective(im_optical, H_est, (W, H), borderMode=cv2.BORDER_CONSTANT)
 * PARITY UNPINNED: OpenCV (opencv-python==4.10.0.82, requirements.txt:2) is absent from /root/reference and from the
 * image; this restates the published algorithm of its imgproc module (imgwarp.cpp: warpPerspective ->
 * WarpPerspectiveInvoker -> remap, INTER_LINEAR, INTER_BITS = 5, INTER_REMAP_COEF_BITS = 15):
 *   - M (forward map, src -> dst) is inverted by the closed 3x3 form of cv::invert (det and cofactors in double,
 *     multiplied by 1/det; a singular M gives the zero matrix) unless inverse_map;
 *   - destination pixels are walked in blocks bw0 = min(1024 / min(16, H), W) wide; per row of a block
 *         X0 = M0*x + M1*y + M2 (x = first column of the block), likewise Y0, W0;   per column x1 of the block
 *         W = W0 + M6*x1;  W = W ? 32/W : 0;  fX = max(INT_MIN, min(INT_MAX, (X0 + M0*x1)*W));  X = cvRound(fX)  (lrint)
 *         sx = saturate_cast<short>(X >> 5), alpha = (Y & 31)*32 + (X & 31)
 *   - remapBilinear, BORDER_CONSTANT, borderValue 0: the 2x2 taps at (sx, sy); taps outside the image read 0.
 *         u8 : integer weights itab = saturate_cast<short>(w * 32768) — for the bilinear table exact integers
 *              (32-ay | ay)*(32-ax | ax)*32 that sum to 32768 — and FixedPtCast: (sum + 16384) >> 15
 *         f32: float weights (1-fy)*(1-fx), (1-fy)*fx, fy*(1-fx), fy*fx; S0*w0 + S1*w1 + S2*w2 + S3*w3 left to right
 * src (Hs, Ws, C) interleaved, dst (Hd, Wd, C).  Compiled with -ffp-contract=off: every product and sum rounds separately.
 * ---------------------------------------------------------------------------------------- */
static void xo_warp_matrix(const double* S, int inverse_map, double* t) {
    if (inverse_map) { for (int k = 0; k < 9; ++k) t[k] = S[k]; return; }
    double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
    if (d != 0.0) {
        d = 1.0 / d;
        t[0] = (S[4] * S[8] - S[5] * S[7]) * d; t[1] = (S[2] * S[7] - S[1] * S[8]) * d; t[2] = (S[1] * S[5] - S[2] * S[4]) * d;
        t[3] = (S[5] * S[6] - S[3] * S[8]) * d; t[4] = (S[0] * S[8] - S[2] * S[6]) * d; t[5] = (S[2] * S[3] - S[0] * S[5]) * d;
        t[6] = (S[3] * S[7] - S[4] * S[6]) * d; t[7] = (S[1] * S[6] - S[0] * S[7]) * d; t[8] = (S[0] * S[4] - S[1] * S[3]) * d;
    } else {
        for (int k = 0; k < 9; ++k) t[k] = 0.0;
    }
}

static void xo_warp_coords(const double* m, int x, int y, int bw, int* sx, int* sy, int* ax, int* ay) {
    const int xb = x / bw * bw, x1 = x - xb;
    const double X0 = m[0] * xb + m[1] * y + m[2], Y0 = m[3] * xb + m[4] * y + m[5], W0 = m[6] * xb + m[7] * y + m[8];
    double W = W0 + m[6] * x1;
    W = W != 0.0 ? 32.0 / W : 0.0;
    double fX = (X0 + m[0] * x1) * W, fY = (Y0 + m[3] * x1) * W;
    fX = fX < 2147483647.0 ? fX : 2147483647.0; fX = fX > -2147483648.0 ? fX : -2147483648.0;      /* NaN -> INT_MIN by these comparisons ... */
    fY = fY < 2147483647.0 ? fY : 2147483647.0; fY = fY > -2147483648.0 ? fY : -2147483648.0;
    if (fX != fX) fX = 0.0;                                                                           /* ... so map it to 0 (the device's conversion of NaN) explicitly */
    if (fY != fY) fY = 0.0;
    const int X = (int)lrint(fX), Y = (int)lrint(fY);
    int a = X >> 5, b = Y >> 5;
    *sx = a < -32768 ? -32768 : (a > 32767 ? 32767 : a);
    *sy = b < -32768 ? -32768 : (b > 32767 ? 32767 : b);
    *ax = X & 31; *ay = Y & 31;
}

static int xo_warp_bw(int Hd, int Wd) { const int bh = Hd < 16 ? Hd : 16; const int bw = 1024 / bh; return bw < Wd ? bw : Wd; }

void xo_warp_perspective_u8(const uint8_t* src, uint8_t* dst, const double* M, int64_t Hs, int64_t Ws, int64_t Hd, int64_t Wd, int64_t C,
                            int inverse_map) {
    double m[9];
    xo_warp_matrix(M, inverse_map, m);
    const int bw = xo_warp_bw((int)Hd, (int)Wd);
#pragma omp parallel for schedule(static)
    for (int64_t y = 0; y < Hd; ++y)
        for (int64_t x = 0; x < Wd; ++x) {
            int sx, sy, ax, ay;
            xo_warp_coords(m, (int)x, (int)y, bw, &sx, &sy, &ax, &ay);
            const int w[4] = {(32 - ay) * (32 - ax) * 32, (32 - ay) * ax * 32, ay * (32 - ax) * 32, ay * ax * 32};
            for (int64_t c = 0; c < C; ++c) {
                int sum = 0;
                for (int k = 0; k < 4; ++k) {
                    const int64_t px = sx + (k & 1), py = sy + (k >> 1);
                    const int v = (px >= 0 && px < Ws && py >= 0 && py < Hs) ? src[(py * Ws + px) * C + c] : 0;
                    sum += v * w[k];
                }
                const int o = (sum + 16384) >> 15;
                dst[(y * Wd + x) * C + c] = (uint8_t)(o > 255 ? 255 : o);
            }
        }
}

void xo_warp_perspective_f32(const float* src, float* dst, const double* M, int64_t Hs, int64_t Ws, int64_t Hd, int64_t Wd, int64_t C,
                             int inverse_map) {
    double m[9];
    xo_warp_matrix(M, inverse_map, m);
    const int bw = xo_warp_bw((int)Hd, (int)Wd);
#pragma omp parallel for schedule(static)
    for (int64_t y = 0; y < Hd; ++y)
        for (int64_t x = 0; x < Wd; ++x) {
            int sx, sy, ax, ay;
            xo_warp_coords(m, (int)x, (int)y, bw, &sx, &sy, &ax, &ay);
            const float fx = (float)ax * 0.03125f, fy = (float)ay * 0.03125f;
            const float w[4] = {(1.f - fy) * (1.f - fx), (1.f - fy) * fx, fy * (1.f - fx), fy * fx};
            for (int64_t c = 0; c < C; ++c) {
                float t[4];
                for (int k = 0; k < 4; ++k) {
                    const int64_t px = sx + (k & 1), py = sy + (k >> 1);
                    t[k] = (px >= 0 && px < Ws && py >= 0 && py < Hs) ? src[(py * Ws + px) * C + c] : 0.f;
                }
                dst[(y * Wd + x) * C + c] = ((t[0] * w[0] + t[1] * w[1]) + t[2] * w[2]) + t[3] * w[3];
            }
        }
}
