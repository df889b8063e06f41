"""CPU restatement (numpy, fp64) of the device-side robust homography estimator xp_find_homography (csrc/homography.hip) — the
stand-in for cv2.findHomography(src, dst, cv2.USAC_MAGSAC, thr, ...) of reference predict_align_image_pair.py:287-303 and
xpoint/utils/benchmark_evaluation.py:796-812.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing in xpoint_amd).  OpenCV is absent from /root/reference and from the image
and its estimator is randomised, so the reference side of this row cannot be pinned ("parity unpinned" for cv2.findHomography);
what IS pinned here is the estimator the product ships: this file replays the same counter-hashed 4-point samples, the same
Hartley-normalised DLT (Gaussian elimination with partial pivoting), the same MSAC score (f32 sum of min(err^2, thr^2) in point
order), the same three rounds of inlier least squares, in plain numpy — an independent statement of the algorithm the HIP kernels
must reproduce hypothesis for hypothesis."""
import numpy as np

M32 = 0xFFFFFFFF


def hash32(x):
    x = np.asarray(x, dtype=np.uint64) & M32
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & M32
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & M32
    x ^= x >> np.uint64(16)
    return x


def sample4(seed, pair, it, n):
    """it: (K,) hypothesis numbers -> (K, 4) distinct correspondence indices (csrc/homography.hip: sample4)."""
    it = np.asarray(it, dtype=np.uint64)
    ctr = hash32(np.uint64(seed) ^ hash32((np.uint64(pair) * np.uint64(0x9e3779b9) + np.uint64(0x85ebca6b)) & M32) ^ hash32((it + np.uint64(0x27d4eb2f)) & M32))
    K = len(it)
    idx = np.zeros((K, 4), dtype=np.int64)
    for k in range(4):
        done = np.zeros(K, dtype=bool)
        for tries in range(64):
            ctr_new = hash32((ctr + np.uint64(0x9e3779b9)) & M32)
            ctr = np.where(done, ctr, ctr_new)
            c = (ctr % np.uint64(n)).astype(np.int64)
            dup = np.zeros(K, dtype=bool)
            for j in range(k):
                dup |= idx[:, j] == c
            take = ~done & ~dup
            idx[take, k] = c[take]
            if tries == 63:
                rest = ~done & dup
                idx[rest, k] = 0 if k == 0 else (idx[rest, k - 1] + 1) % n
                done |= rest
            done |= take
            if done.all():
                break
    return idx


def solve8(M):
    """(K, 8, 9) augmented systems, Gaussian elimination with partial pivoting as the device does it -> (h (K, 8), ok (K,))."""
    M = M.copy()
    K = M.shape[0]
    ok = np.ones(K, dtype=bool)
    ar = np.arange(K)
    for c in range(8):
        piv = c + np.argmax(np.abs(M[:, c:, c]), axis=1)          # first maximum, like the strict '>' scan
        best = np.abs(M[ar, piv, c])
        ok &= best >= 1e-12
        rc = M[ar, c, :].copy(); rp = M[ar, piv, :].copy()
        M[ar, c, :] = rp; M[ar, piv, :] = rc
        safe = np.where(np.abs(M[:, c, c]) > 0, M[:, c, c], 1.0)
        inv = 1.0 / safe
        for r in range(c + 1, 8):
            f = M[:, r, c] * inv
            M[:, r, c:] -= f[:, None] * M[:, c, c:]
    h = np.zeros((K, 8))
    for c in range(7, -1, -1):
        v = M[:, c, 8].copy()
        for k in range(c + 1, 8):
            v -= M[:, c, k] * h[:, k]
        h[:, c] = v / np.where(np.abs(M[:, c, c]) > 0, M[:, c, c], 1.0)
    return h, ok


def _norm(src, dst):
    n = len(src)
    inv = 1.0 / n if n else 0.0
    sx, sy = src[:, 0].sum() * inv, src[:, 1].sum() * inv
    dx, dy = dst[:, 0].sum() * inv, dst[:, 1].sum() * inv
    ms = np.sqrt((src[:, 0] - sx) ** 2 + (src[:, 1] - sy) ** 2).sum() * inv
    md = np.sqrt((dst[:, 0] - dx) ** 2 + (dst[:, 1] - dy) ** 2).sum() * inv
    s = np.sqrt(2.0) / ms if ms > 1e-9 else 1.0
    d = np.sqrt(2.0) / md if md > 1e-9 else 1.0
    return sx, sy, s, dx, dy, d


def _denorm(hn, nm):
    sx, sy, s, dx, dy, d = nm
    a = np.concatenate([hn, np.ones(hn.shape[:-1] + (1,))], -1).reshape(hn.shape[:-1] + (3, 3))
    b = np.empty_like(a)
    b[..., 0] = a[..., 0] * s
    b[..., 1] = a[..., 1] * s
    b[..., 2] = -a[..., 0] * s * sx - a[..., 1] * s * sy + a[..., 2]
    H = np.empty_like(a)
    H[..., 0, :] = b[..., 0, :] / d + dx * b[..., 2, :]
    H[..., 1, :] = b[..., 1, :] / d + dy * b[..., 2, :]
    H[..., 2, :] = b[..., 2, :]
    return H


def _err2(H, src, dst):
    """(..., 3, 3) x (n, 2) -> (..., n) squared forward reprojection error, f32 like the device returns it."""
    x, y = src[:, 0], src[:, 1]
    w = H[..., 2, 0, None] * x + H[..., 2, 1, None] * y + H[..., 2, 2, None]
    iw = np.where(np.abs(w) > 1e-12, 1.0 / np.where(w == 0, 1.0, w), 0.0)
    pu = (H[..., 0, 0, None] * x + H[..., 0, 1, None] * y + H[..., 0, 2, None]) * iw
    pv = (H[..., 1, 0, None] * x + H[..., 1, 1, None] * y + H[..., 1, 2, None]) * iw
    du, dv = pu - dst[:, 0], pv - dst[:, 1]
    return (du * du + dv * dv).astype(np.float32)


def _rows(src, dst, nm):
    sx, sy, s, dx, dy, d = nm
    x, y = (src[:, 0] - sx) * s, (src[:, 1] - sy) * s
    u, v = (dst[:, 0] - dx) * d, (dst[:, 1] - dy) * d
    z, o = np.zeros_like(x), np.ones_like(x)
    r0 = np.stack([x, y, o, z, z, z, -u * x, -u * y], -1)
    r1 = np.stack([z, z, z, x, y, o, -v * x, -v * y], -1)
    return r0, r1, u, v


def find_homography(src, dst, reproj_thr=3.0, max_iters=10000, seed=0, pair=0):
    """src, dst (n, 2) float32 (x, y).  Returns (H (3, 3) f64 with h33 = 1, mask (n,) uint8, n_inliers, best_hypothesis)."""
    src = np.asarray(src, dtype=np.float32).astype(np.float64); dst = np.asarray(dst, dtype=np.float32).astype(np.float64)
    n = len(src)
    if n < 4:
        return np.eye(3), np.zeros(n, np.uint8), 0, -1
    thr2 = np.float32(reproj_thr) * np.float32(reproj_thr)
    nm = _norm(src, dst)
    best_key, best_it = None, -1
    for i0 in range(0, max_iters, 2000):
        its = np.arange(i0, min(max_iters, i0 + 2000))
        idx = sample4(seed, pair, its, n)
        r0, r1, u, v = _rows(src, dst, nm)
        M = np.zeros((len(its), 8, 9))
        for k in range(4):
            M[:, 2 * k, :8] = r0[idx[:, k]]; M[:, 2 * k, 8] = u[idx[:, k]]
            M[:, 2 * k + 1, :8] = r1[idx[:, k]]; M[:, 2 * k + 1, 8] = v[idx[:, k]]
        hn, ok = solve8(M)
        H = _denorm(hn, nm)
        e = np.minimum(_err2(H, src, dst), thr2)                   # (K, n) f32
        score = np.zeros(len(its), dtype=np.float32)
        for j in range(n):                                         # the device's f32 running sum, in point order
            score = score + e[:, j]
        for kk in np.nonzero(ok)[0]:
            key = (int(score[kk].view(np.uint32)), int(its[kk]))
            if best_key is None or key < best_key:
                best_key, best_it = key, int(its[kk])
    if best_it < 0:
        return np.eye(3), np.zeros(n, np.uint8), 0, -1
    idx = sample4(seed, pair, np.array([best_it]), n)
    r0, r1, u, v = _rows(src, dst, nm)
    M = np.zeros((1, 8, 9))
    for k in range(4):
        M[0, 2 * k, :8] = r0[idx[0, k]]; M[0, 2 * k, 8] = u[idx[0, k]]
        M[0, 2 * k + 1, :8] = r1[idx[0, k]]; M[0, 2 * k + 1, 8] = v[idx[0, k]]
    hn, _ = solve8(M)
    H = _denorm(hn, nm)[0]
    for _ in range(3):
        inl = _err2(H, src, dst) <= thr2
        if inl.sum() < 4:
            break
        a0, a1 = r0[inl], r1[inl]
        N = a0.T @ a0 + a1.T @ a1
        rhs = a0.T @ u[inl] + a1.T @ v[inl]
        hn2, ok2 = solve8(np.concatenate([N, rhs[:, None]], 1)[None])
        if not ok2[0]:
            break
        H = _denorm(hn2, nm)[0]
    mask = (_err2(H, src, dst) <= thr2).astype(np.uint8)
    sc = 1.0 / H[2, 2] if abs(H[2, 2]) > 1e-300 else 1.0
    return H * sc, mask, int(mask.sum()), best_it
