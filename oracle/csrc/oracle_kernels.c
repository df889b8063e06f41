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
