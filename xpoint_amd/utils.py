"""Host-side mirror of the reference's `xpoint.utils` functions on the hot path, over the HIP C ABI:
`box_nms`, `interpolate_descriptors` (reference xpoint/utils/utils.py:148-192, 229-238),
`get_matches` (xpoint/utils/matching.py:4-36) plus the small dict helpers the callers use
(utils.py:73-113, 240-246).  Same names, argument meaning and error behaviour."""
from __future__ import annotations

import collections
import collections.abc
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import c_f, c_i, ptr


# ---------------------------------------------------------------- dict / data helpers (utils.py:73-113,240-246)
def dict_update(d, u):
    for k, v in u.items():
        if isinstance(v, collections.abc.Mapping):
            d[k] = dict_update(d.get(k, {}), v)
        else:
            d[k] = v
    return d


def data_to_device(data, device):
    for key in data.keys():
        if type(data[key]) is torch.Tensor:
            data[key] = data[key].to(device)
        elif type(data[key]) is dict:
            data[key] = data_to_device(data[key], device)
    return data


def data_unsqueeze(data, dim):
    for key in data.keys():
        if type(data[key]) is torch.Tensor:
            data[key] = data[key].unsqueeze(dim)
        elif type(data[key]) is dict:
            data[key] = data_unsqueeze(data[key], dim)
    return data


def fix_model_weigth_keys(weights):
    new_weights = collections.OrderedDict()
    for key, value in weights.items():
        new_weights[key.split('__')[-1]] = value
    return new_weights


# ---------------------------------------------------------------- box_nms (utils.py:148-192)
def box_nms(prob, size, min_prob, iou=0.1, keep_top_k=0, on_cpu=False):
    """prob (H,W) or (B,1,H,W) on the GPU -> same shape: surviving scores, zeros elsewhere.
    `on_cpu` (the reference moves the tensors to the CPU for torchvision's CPU NMS) is accepted and ignored:
    the result is defined by the algorithm, not by where it runs."""
    if not (len(prob.shape) == 2 or len(prob.shape) == 4):
        raise ValueError('The probability must be either 2D (H,W), or 4D (B, 1, H, W)')
    if not prob.is_cuda:
        raise RuntimeError("xpoint_amd.utils.box_nms runs on the GPU only (no CPU fallback)")
    if len(prob.shape) == 4 and prob.shape[1] != 1:
        raise ValueError('The probability must be either 2D (H,W), or 4D (B, 1, H, W)')
    p = prob.contiguous().float()
    H, W = p.shape[-2:]
    B = p.numel() // (H * W) if H * W else 0
    out = torch.zeros_like(p)
    if p.numel() == 0:
        return out
    cap = H * W if keep_top_k > 0 else 1
    lib = _lib.load()
    ws = torch.empty(lib.xp_box_nms_workspace_bytes(B, H, W, cap), dtype=torch.uint8, device=p.device)
    conv = c_i(0)
    _lib.check(lib.xp_box_nms(ptr(p), ptr(out), ptr(ws), ws.numel(), B, H, W, float(size), float(min_prob), float(iou),
                              int(keep_top_k), cap, 0, ctypes.byref(conv), _lib.current_stream()), "xp_box_nms")
    return out


def extract_keypoints(prob, thr, mask=None, cap=None):
    """torch.nonzero((prob > thr)[* mask]) for every image of prob (B,H,W) / (B,1,H,W) / (H,W):
    returns (kp (B,cap,2) int32 (y,x) row-major, counts (B) int32) on the GPU."""
    p = prob.contiguous().float()
    H, W = p.shape[-2:]
    B = p.numel() // (H * W)
    cap = int(cap or H * W)
    kp = torch.zeros((B, cap, 2), dtype=torch.int32, device=p.device)
    counts = torch.zeros((B,), dtype=torch.int32, device=p.device)
    m = None
    if mask is not None:
        m = mask.contiguous().to(torch.uint8)
    ws = torch.empty(_lib.load().xp_extract_keypoints_workspace_bytes(B, H, W), dtype=torch.uint8, device=p.device)
    _lib.call("xp_extract_keypoints", ptr(p), ptr(m), float(thr), ptr(kp), ptr(counts), B, H, W, cap, ptr(ws), ws.numel(),
              _lib.current_stream())
    return kp, counts


# ---------------------------------------------------------------- interpolate_descriptors (utils.py:229-238)
def interpolate_descriptors(keypoints, descriptors_lowres, H, W):
    """keypoints (N,2) (y,x) integer tensor; descriptors_lowres (C,Hc,Wc) as the reference passes it
    (or (Hc,Wc,C) NHWC with `descriptors_lowres.nhwc = True` via interpolate_descriptors_nhwc) -> (N,C)."""
    d = descriptors_lowres
    if d.dim() != 3:
        raise RuntimeError("descriptors_lowres must be (C,Hc,Wc)")
    return interpolate_descriptors_nhwc(keypoints, d.permute(1, 2, 0).contiguous(), H, W)


def interpolate_descriptors_nhwc(keypoints, desc_nhwc, H, W):
    if not desc_nhwc.is_cuda:
        raise RuntimeError("xpoint_amd.utils.interpolate_descriptors runs on the GPU only (no CPU fallback)")
    Hc, Wc, D = desc_nhwc.shape
    n = int(keypoints.shape[0])
    out = torch.empty((max(n, 1), D), device=desc_nhwc.device)
    if n == 0:
        return out[:0]
    kp = keypoints.to(device=desc_nhwc.device, dtype=torch.int32).contiguous().view(1, n, 2)
    counts = torch.tensor([n], dtype=torch.int32, device=desc_nhwc.device)
    dvol = desc_nhwc.contiguous().float()      # hold a reference: ptr() of a temporary could be reused before the launch
    _lib.call("xp_sample_descriptors", ptr(kp), ptr(counts), ptr(dvol), ptr(out), 1, n, Hc, Wc, D,
              int(H), int(W), _lib.current_stream())
    return out


# ---------------------------------------------------------------- get_matches (matching.py:4-36)
class DMatch:
    """cv2.DMatch stand-in: the fields the reference's callers read
    (predict_align_image_pair.py:283-284, benchmark_evaluation.py:666-675)."""
    __slots__ = ("queryIdx", "trainIdx", "distance", "imgIdx")

    def __init__(self, queryIdx=-1, trainIdx=-1, distance=0.0):
        self.queryIdx, self.trainIdx, self.distance, self.imgIdx = int(queryIdx), int(trainIdx), float(distance), 0

    def __repr__(self):
        return f"DMatch({self.queryIdx}, {self.trainIdx}, {self.distance:.6f})"


MATCH_MODES = {"strict_mnn": 0, "legacy_crosscheck": 1}


def match_descriptors(d1, d2, counts=None, mode="strict_mnn"):
    """Batched device-side matching.  d1 (P,cap1,D), d2 (P,cap2,D) GPU float32; counts: optional int32 (2P,)
    = [n1 of every pair..., n2 of every pair...].  Returns a dict of device tensors."""
    P, cap1, D = d1.shape
    cap2 = d2.shape[1]
    dev = d1.device
    lib = _lib.load()
    res = dict(idx12=torch.empty((P, cap1), dtype=torch.int32, device=dev), dist12=torch.empty((P, cap1), device=dev),
               idx21=torch.empty((P, cap2), dtype=torch.int32, device=dev), dist21=torch.empty((P, cap2), device=dev),
               match_q=torch.empty((P, cap1), dtype=torch.int32, device=dev), match_t=torch.empty((P, cap1), dtype=torch.int32, device=dev),
               match_d=torch.empty((P, cap1), device=dev), match_count=torch.zeros((P,), dtype=torch.int32, device=dev))
    ws = torch.empty(lib.xp_match_workspace_bytes(P, cap1, cap2, D), dtype=torch.uint8, device=dev)
    d1, d2 = d1.contiguous(), d2.contiguous()
    _lib.check(lib.xp_match_mnn(ptr(d1), ptr(d2), ptr(counts), 1, 0, P, P, cap1, cap2, D, MATCH_MODES[mode],
                                ptr(res["idx12"]), ptr(res["dist12"]), ptr(res["idx21"]), ptr(res["dist21"]), ptr(res["match_q"]),
                                ptr(res["match_t"]), ptr(res["match_d"]), ptr(res["match_count"]), ptr(ws), ws.numel(),
                                _lib.current_stream()), "xp_match_mnn")
    res["_ws"] = ws
    res["_call"] = (counts, P, cap1, cap2, D)
    return res


def match_stats(res):
    """Nomination statistics of a match_descriptors() result (xp_match_stats): dict(mean, max, overflow_rows, rows, cand_cap) over the live rows and
    columns of the call — what the matcher's run time depends on (its result never does)."""
    counts, P, cap1, cap2, D = res["_call"]
    lib = _lib.load()
    out = torch.zeros(4, dtype=torch.int64, device=res["_ws"].device)
    _lib.check(lib.xp_match_stats(ptr(res["_ws"]), ptr(counts), 1, 0, P, P, cap1, cap2, D, ptr(out), _lib.current_stream()), "xp_match_stats")
    s, mx, ov, rows = (int(v) for v in out.cpu())
    return dict(mean=s / max(rows, 1), max=mx, overflow_rows=ov, rows=rows, cand_cap=int(lib.xp_match_cand_cap()))


def knn2_descriptors(d1, d2, counts=None):
    """Batched device-side k = 2 nearest neighbours (xp_match_knn2): idx (P,cap1,2) int32, dist (P,cap1,2) float32."""
    P, cap1, D = d1.shape
    cap2 = d2.shape[1]
    lib = _lib.load()
    idx = torch.empty((P, cap1, 2), dtype=torch.int32, device=d1.device); dist = torch.empty((P, cap1, 2), device=d1.device)
    ws = torch.empty(lib.xp_match_workspace_bytes(P, cap1, cap2, D), dtype=torch.uint8, device=d1.device)
    d1, d2 = d1.contiguous(), d2.contiguous()
    _lib.check(lib.xp_match_knn2(ptr(d1), ptr(d2), ptr(counts), 1, 0, P, P, cap1, cap2, D, ptr(idx), ptr(dist), ptr(ws), ws.numel(), _lib.current_stream()),
               "xp_match_knn2")
    return idx, dist


def threshold_pairs(d1, d2, threshold, counts=None):
    """Batched device-side ThresholdMatcher (xp_match_threshold): (pair, query, target) int32 rows sorted row-major per pair, and their distances."""
    P, cap1, D = d1.shape
    cap2 = d2.shape[1]
    lib = _lib.load()
    dev = d1.device
    d1, d2 = d1.contiguous(), d2.contiguous()
    ws = torch.empty(lib.xp_match_workspace_bytes(P, cap1, cap2, D), dtype=torch.uint8, device=dev)
    cap = max(4096, 4 * P * max(cap1, cap2))
    while True:
        hits = torch.empty((cap, 2), dtype=torch.int32, device=dev)
        out = torch.empty((cap, 3), dtype=torch.int32, device=dev); dist = torch.empty((cap,), device=dev)
        cnt = torch.zeros(8, dtype=torch.int32, device=dev)
        _lib.check(lib.xp_match_threshold(ptr(d1), ptr(d2), ptr(counts), 1, 0, P, P, cap1, cap2, D, float(threshold), ptr(hits), cap, ptr(out), ptr(dist), ptr(cnt),
                                          cap, ptr(ws), ws.numel(), _lib.current_stream()), "xp_match_threshold")
        n_out, n_hit = int(cnt[0].item()), int(cnt[1].item())
        if n_hit <= cap and n_out <= cap:
            break
        cap = max(n_hit, n_out) + 1024                      # the lists were truncated: repeat with room for everything (counts are exact)
    out = out[:n_out].cpu().numpy(); dist = dist[:n_out].cpu().numpy()
    order = np.lexsort((out[:, 2], out[:, 1], out[:, 0]))
    return out[order], dist[order]


def _as_device_f32(d):
    dev = d.device if (torch.is_tensor(d) and d.is_cuda) else torch.device("cuda", torch.cuda.current_device())
    return torch.as_tensor(np.ascontiguousarray(d) if isinstance(d, np.ndarray) else d).to(dev).float()


def get_matches(desc_1, desc_2, method='bfmatcher', knn_matches=False, mode="strict_mnn", **kwargs):
    """desc_1 (N1,D), desc_2 (N2,D): numpy arrays (as the reference passes them) or torch tensors (matching.py:4-36).
    method 'bfmatcher' with crossCheck=True (configs/cipdp.yaml:57-61) -> list of DMatch in ascending queryIdx;
    `mode` picks the cross-check semantics: 'strict_mnn' (default) or 'legacy_crosscheck'.
    'nnmatcher' = strict mutual NN with the reference's distance threshold (matching.py:38-75).
    'thresholdmatcher' = every pair with sqrt(2 - 2 <a, b>) < threshold (default 0.4), row-major (matching.py:77-102).
    knn_matches=True ('bfmatcher'): the two nearest targets of every query + Lowe's ratio test `m.distance < 0.9 * n.distance` (matching.py:20-27).
    'flann' (cv2.FlannBasedMatcher: randomised kd-trees, approximate, no reference semantics to reproduce) raises."""
    if method not in ('bfmatcher', 'nnmatcher', 'thresholdmatcher'):
        if method == 'flann':
            raise NotImplementedError("matching method 'flann' is an approximate randomised search in OpenCV: there is no result to reproduce (use 'bfmatcher')")
        raise ValueError('unknown matching method')
    if method == 'nnmatcher' and float(kwargs.get('threshold', 0.7)) < 0.0 or method == 'thresholdmatcher' and float(kwargs.get('threshold', 0.4)) < 0.0:
        raise ValueError("'threshold' should be non-negative")                          # matching.py:41-42, :80-81
    if knn_matches:
        if method != 'bfmatcher':
            raise AttributeError(f"'{'NNMatcher' if method == 'nnmatcher' else 'ThresholdMatcher'}' object has no attribute 'knnMatch'")   # what the reference does
        if kwargs.get('crossCheck', False):
            raise RuntimeError("BFMatcher.knnMatch with crossCheck=True requires k == 1 (OpenCV asserts); the reference calls it with k = 2")
        if desc_1.shape[0] == 0:
            return []
        if desc_2.shape[0] < 2:       # knnMatch returns shorter lists and the reference's `for m, n in all_matches` fails to unpack
            raise ValueError(f"not enough values to unpack (expected 2, got {desc_2.shape[0]})")
        t1, t2 = _as_device_f32(desc_1), _as_device_f32(desc_2)
        with torch.cuda.device(t1.device):
            idx, dist = knn2_descriptors(t1.unsqueeze(0), t2.unsqueeze(0))
        idx = idx[0].cpu().numpy(); dist = dist[0].cpu().numpy().astype(np.float64)      # DMatch.distance is a float32 value; Python compares doubles
        keep = dist[:, 0] < 0.9 * dist[:, 1]                                              # ratio_thresh = 0.9 (matching.py:23-26)
        return [DMatch(q, idx[q, 0], dist[q, 0]) for q in np.nonzero(keep)[0]]
    if desc_1.shape[0] == 0 or desc_2.shape[0] == 0:
        return []
    # device tensors are matched where they live; host arrays (what the reference passes) go to the current device
    t1, t2 = _as_device_f32(desc_1), _as_device_f32(desc_2)
    dev = t1.device
    if method == 'thresholdmatcher':
        thr = float(kwargs.get('threshold', 0.4))
        if thr > 2.0:
            raise ValueError("thresholdmatcher: threshold > 2 accepts every pair of unit descriptors")
        with torch.cuda.device(dev):
            pairs, dist = threshold_pairs(t1.unsqueeze(0), t2.unsqueeze(0), thr)
        return [DMatch(a, b, c) for (_, a, b), c in zip(pairs, dist)]
    with torch.cuda.device(dev):
        res = match_descriptors(t1.unsqueeze(0), t2.unsqueeze(0), None, mode)
    if method == 'bfmatcher' and not kwargs.get('crossCheck', False):
        # BFMatcher default (crossCheck False): the nearest train descriptor of every query
        t = res["idx12"][0].cpu().numpy(); d = res["dist12"][0].cpu().numpy(); q = np.arange(len(t))
    else:
        n = int(res["match_count"][0].item())
        q = res["match_q"][0, :n].cpu().numpy(); t = res["match_t"][0, :n].cpu().numpy(); d = res["match_d"][0, :n].cpu().numpy()
    if method == 'nnmatcher':
        keep = d < float(kwargs.get('threshold', 0.7))
        q, t, d = q[keep], t[keep], d[keep]
    return [DMatch(a, b, c) for a, b, c in zip(q, t, d)]


# ---------------------------------------------------------------- homography estimation
def find_homography_batched(src, dst, counts=None, reproj_threshold=3.0, max_iters=10000, seed=0):
    """src, dst: (P, cap, 2) float32 device tensors of (x, y) correspondences (row i of src matches row i of dst);
    counts: optional (P,) int32 device tensor.  Returns device tensors H (P,3,3) float64, mask (P,cap) uint8,
    n_inliers (P,) int32 (0 = no model).  Device stand-in for cv2.findHomography(..., USAC_MAGSAC, thr): same contract,
    deterministic estimator (include/xpoint_hip.h: xp_find_homography)."""
    P, cap, _ = src.shape
    dev = src.device
    lib = _lib.load()
    src, dst = src.contiguous().float(), dst.contiguous().float()
    H = torch.empty((P, 3, 3), dtype=torch.float64, device=dev)
    mask = torch.empty((P, cap), dtype=torch.uint8, device=dev)
    n_inl = torch.empty((P,), dtype=torch.int32, device=dev)
    ws = torch.empty(lib.xp_find_homography_workspace_bytes(P) // 8 + 1, dtype=torch.float64, device=dev)
    cnt = counts.to(torch.int32).contiguous() if counts is not None else None
    _lib.check(lib.xp_find_homography(ptr(src), ptr(dst), ptr(cnt), P, cap, float(reproj_threshold), int(max_iters), int(seed) & 0xffffffff,
                                      ptr(H), ptr(mask), ptr(n_inl), ptr(ws), ws.numel() * 8, _lib.current_stream()), "xp_find_homography")
    return H, mask, n_inl


def find_homography(src_pts, dst_pts, reproj_threshold=3.0, max_iters=10000, seed=0):
    """cv2.findHomography-shaped call: src_pts / dst_pts (N,1,2) or (N,2) float (x, y) numpy arrays or tensors.
    Returns (H (3,3) float64 numpy or None, mask (N,1) uint8 numpy) like cv2 does."""
    import numpy as np
    s = torch.as_tensor(np.asarray(src_pts, dtype=np.float32) if not torch.is_tensor(src_pts) else src_pts).reshape(-1, 2).float()
    d = torch.as_tensor(np.asarray(dst_pts, dtype=np.float32) if not torch.is_tensor(dst_pts) else dst_pts).reshape(-1, 2).float()
    n = s.shape[0]
    if n < 4:
        return None, np.zeros((n, 1), np.uint8)
    H, mask, n_inl = find_homography_batched(s.cuda()[None], d.cuda()[None], None, reproj_threshold, max_iters, seed)
    if int(n_inl[0].item()) == 0:
        return None, np.zeros((n, 1), np.uint8)
    return H[0].cpu().numpy(), mask[0].cpu().numpy().reshape(n, 1)


# ---------------------------------------------------------------- image registration output
def warp_perspective(img, M, dsize=None, inverse_map=False, quantise_u8=False, dst_channels=None):
    """cv2.warpPerspective(img, M, dsize, flags=INTER_LINEAR [| WARP_INVERSE_MAP], borderMode=BORDER_CONSTANT) on the device
    (reference predict_align_image_pair.py:308, demo.py:225-249).

    img: device tensor uint8 or float32, (H, W), (H, W, C) or batched (B, H, W, C) with C <= 4 channel-interleaved as cv2 images are
    (a (B, 1, H, W) network input is accepted as B one-channel images); M: (3, 3) or (B, 3, 3) forward map src -> dst in (x, y)
    — numpy, or a float64 device tensor such as `find_homography_batched` returns (no host round trip); dsize = (width, height)
    as in cv2 (default: the source size).  quantise_u8: a float32 image in [0, 1] is quantised on load exactly as the reference
    builds `im_optical` ((np.clip(img, 0, 1) * 255.0).astype(np.uint8)) and the result is uint8; dst_channels=3 replicates a
    one-channel source (cv2.COLOR_GRAY2RGB ahead of the warp).  Returns a device tensor shaped like the input with (Hd, Wd).
    Arithmetic: OpenCV's documented 1/32-pixel fixed-point scheme (include/xpoint_hip.h: xp_warp_perspective; parity unpinned)."""
    import numpy as np
    if not (torch.is_tensor(img) and img.is_cuda):
        raise _lib.XPointHipError("warp_perspective needs a device tensor: xpoint_amd has no CPU fallback")
    if img.dtype not in (torch.uint8, torch.float32):
        raise ValueError(f"warp_perspective: uint8 or float32 images, got {img.dtype}")
    if quantise_u8 and img.dtype != torch.float32:
        raise ValueError("warp_perspective: quantise_u8 applies to float32 images")
    x = img
    shape_kind = x.dim()
    nchw1 = x.dim() == 4 and x.shape[1] == 1 and x.shape[3] > 4
    if x.dim() == 2:
        x = x[None, :, :, None]
    elif x.dim() == 3:
        x = x[None]
    elif x.dim() == 4:
        if nchw1:
            x = x.permute(0, 2, 3, 1)
    else:
        raise ValueError(f"warp_perspective: bad image shape {tuple(img.shape)}")
    x = x.contiguous()
    B, Hs, Ws, C = x.shape
    Wd, Hd = (Ws, Hs) if dsize is None else (int(dsize[0]), int(dsize[1]))
    Cd = C if dst_channels is None else int(dst_channels)
    if torch.is_tensor(M) and M.is_cuda:
        Md = M.to(torch.float64).reshape(-1, 9)
    else:
        Md = torch.from_numpy(np.ascontiguousarray(np.asarray(M.cpu() if torch.is_tensor(M) else M, dtype=np.float64).reshape(-1, 9))).to(x.device)
    if Md.shape[0] == 1 and B > 1:
        Md = Md.expand(B, 9)
    if Md.shape[0] != B:
        raise ValueError(f"warp_perspective: {Md.shape[0]} matrices for {B} images")
    Md = Md.contiguous()
    u8_out = x.dtype == torch.uint8 or quantise_u8
    out = torch.empty((B, Hd, Wd, Cd), dtype=torch.uint8 if u8_out else torch.float32, device=x.device)
    dtype = 0 if x.dtype == torch.uint8 else (2 if quantise_u8 else 1)
    with torch.cuda.device(x.device):
        _lib.call("xp_warp_perspective", ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(Md.data_ptr()),
                  B, Hs, Ws, Hd, Wd, C, Cd, dtype, 1 if inverse_map else 0, _lib.current_stream(x.device))
    if shape_kind == 2:
        return out[0, :, :, 0] if Cd == 1 else out[0]
    if shape_kind == 3:
        return out[0]
    return out.permute(0, 3, 1, 2) if (nchw1 and Cd == 1) else out


def checkerboard_visualization(img_visible, img_other, H, cell_size=50):
    """demo.py:222-234 `create_checkerboard_visualization`: the visible image warped into the other image's frame by H
    (`cv2.warpPerspective(img_visible, H, (W, H))`), composited with the other image in a checkerboard of `cell_size`-pixel cells
    (cells with odd (x // cell + y // cell) show the warped image).  img_* : (H, W) device tensors of one dtype (uint8 or float32); returns a
    device tensor like img_other.  The warp is `warp_perspective`; the select is elementwise glue."""
    if img_visible.dtype != img_other.dtype or img_other.dim() != 2 or img_visible.dim() != 2:
        raise ValueError("checkerboard_visualization: two (H, W) images of the same dtype")
    Ho, Wo = img_other.shape
    warped = warp_perspective(img_visible, H, (Wo, Ho))
    y = torch.arange(Ho, device=img_other.device)[:, None] // int(cell_size)
    x = torch.arange(Wo, device=img_other.device)[None, :] // int(cell_size)
    return torch.where(((x + y) % 2).bool(), warped, img_other)
