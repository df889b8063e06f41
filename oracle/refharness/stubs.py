"""Import shims that let the *real* reference package (`/root/reference/xpoint`) be imported
in the build container, where several of its third-party dependencies are absent.

TEST INFRASTRUCTURE ONLY (oracle pinning / golden-vector generation).  Nothing here is
imported by the product (`xpoint_amd`) and nothing here travels as reference code: the
reference sources stay under /root/reference and are only *imported* from there.

What is shimmed:
  * timm.models.layers   -> DropPath (identity in eval), trunc_normal_, to_2tuple
  * fvcore.nn            -> 4 unused names
  * yacs.config.CfgNode  -> attribute dict with clone/defrost/freeze/merge_from_file
  * cv2                  -> DMatch + __version__; perspectiveTransform (the textbook projective map, fp64) and a
                            BFMatcher(NORM_L2, crossCheck=...) whose match() is the exact (fp64, first-minimum)
                            nearest neighbour / mutual nearest neighbour.  Both are stand-ins written from the
                            documented behaviour, NOT OpenCV: they exist so that the reference's evaluation
                            harness (benchmark_evaluation.py) runs here; parity with OpenCV itself stays unpinned
  * h5py, kornia, matplotlib, GPUtil -> empty modules
  * torchvision.ops.nms / ops.boxes.batched_nms -> plain greedy restatement (documented
    algorithm: stable descending sort; suppress when inter/(a_i+a_j-inter) > iou)
  * torch.cuda.device    -> nullcontext for CPU devices (reference csm_triton.py:505-517
    wraps CPU tensors in `torch.cuda.device(x.device)` which raises on torch 2.10)
"""
import contextlib
import copy
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REFERENCE_ROOT = os.environ.get("XPOINT_REFERENCE_ROOT", "/root/reference")


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []  # behave like a package so that submodule imports resolve
    sys.modules[name] = m
    return m


class _DropPath(nn.Module):
    def __init__(self, drop_prob=0.0, *a, **k):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        assert not self.training, "harness DropPath shim is eval-only"
        return x


class CfgNode(dict):
    """Minimal stand-in for yacs.config.CfgNode (attribute access on a dict tree)."""

    def __init__(self, init=None):
        super().__init__()
        if init:
            for k, v in init.items():
                self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)

    def defrost(self):
        pass

    def freeze(self):
        pass

    def _merge(self, other):
        for k, v in other.items():
            if isinstance(v, dict):
                node = self.setdefault(k, CfgNode())
                if not isinstance(node, CfgNode):
                    node = CfgNode(node)
                    self[k] = node
                node._merge(v)
            else:
                self[k] = v

    def merge_from_file(self, path):
        import yaml
        with open(path, "r") as f:
            self._merge(yaml.safe_load(f) or {})

    def setdefault(self, k, default=None):
        if k not in self:
            self[k] = CfgNode(default) if isinstance(default, dict) and not isinstance(default, CfgNode) else default
        return self[k]


class DMatch:
    def __init__(self, queryIdx=-1, trainIdx=-1, distance=0.0):
        self.queryIdx = int(queryIdx)
        self.trainIdx = int(trainIdx)
        self.distance = float(distance)
        self.imgIdx = 0


def greedy_nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """Documented torchvision.ops.nms semantics (torchvision is not installed here).
    Returns kept indices sorted by decreasing score (stable)."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    b = boxes.detach().cpu().numpy().astype(np.float32)
    s = scores.detach().cpu().numpy()
    order = np.argsort(-s, kind="stable")
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    suppressed = np.zeros(len(b), dtype=bool)
    keep = []
    for _i in range(len(order)):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        xx1 = np.maximum(x1[i], x1[rest]); yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest]); yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(np.float32(0), xx2 - xx1); h = np.maximum(np.float32(0), yy2 - yy1)
        inter = w * h
        ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr > np.float32(iou_threshold)]] = True
    return torch.as_tensor(np.asarray(keep, dtype=np.int64))


def batched_nms(boxes, scores, idxs, iou_threshold):
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
    return greedy_nms(boxes + offsets[:, None], scores, iou_threshold)


def perspective_transform(src, m):
    """cv2.perspectiveTransform for an array of shape (1, N, 2): (x, y) -> (x', y') = (X/W, Y/W), [X Y W]^T = M [x y 1]^T."""
    src = np.asarray(src)
    m = np.asarray(m, dtype=np.float64)
    pts = src.reshape(-1, 2).astype(np.float64)
    hom = np.concatenate([pts, np.ones((pts.shape[0], 1))], 1) @ m.T
    out = hom[:, :2] / hom[:, 2:3]
    return out.reshape(src.shape).astype(src.dtype if src.dtype in (np.float32, np.float64) else np.float64)


class BFMatcher:
    """Stand-in for cv2.BFMatcher(NORM_L2, crossCheck): brute force in fp64, first minimum wins, results ordered by
    queryIdx; crossCheck=True keeps (q, t) iff t is q's nearest train descriptor and q is t's nearest query descriptor."""

    def __init__(self, norm=4, crossCheck=False):
        self.cross = bool(crossCheck)

    def match(self, d1, d2):
        d1 = np.asarray(d1, dtype=np.float64); d2 = np.asarray(d2, dtype=np.float64)
        if d1.shape[0] == 0 or d2.shape[0] == 0:
            return []
        d = np.sqrt(np.maximum(((d1[:, None, :] - d2[None, :, :]) ** 2).sum(-1), 0.0)) if d1.shape[0] * d2.shape[0] <= 1 << 22 else None
        if d is None:       # large case: row blocks
            nn12 = np.empty(d1.shape[0], np.int64); dm = np.empty(d1.shape[0]); best21 = np.full(d2.shape[0], np.inf); nn21 = np.zeros(d2.shape[0], np.int64)
            for i0 in range(0, d1.shape[0], 256):
                blk = np.sqrt(np.maximum(((d1[i0:i0 + 256, None, :] - d2[None, :, :]) ** 2).sum(-1), 0.0))
                nn12[i0:i0 + 256] = blk.argmin(1); dm[i0:i0 + 256] = blk.min(1)
                cm = blk.min(0); ca = blk.argmin(0) + i0
                upd = cm < best21
                best21[upd] = cm[upd]; nn21[upd] = ca[upd]
        else:
            nn12 = d.argmin(1); dm = d.min(1); nn21 = d.argmin(0)
        out = []
        for q in range(d1.shape[0]):
            t = int(nn12[q])
            if not self.cross or int(nn21[t]) == q:
                out.append(DMatch(q, t, float(np.float32(dm[q]))))
        return out


_installed = False


def install():
    global _installed
    if _installed:
        return
    _installed = True
    sys.dont_write_bytecode = True  # never drop __pycache__ into the reference tree
    os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")

    layers = dict(DropPath=_DropPath, trunc_normal_=nn.init.trunc_normal_,
                  to_2tuple=lambda x: x if isinstance(x, (tuple, list)) else (x, x))
    _mod("timm"); _mod("timm.models"); _mod("timm.models.layers", **layers)
    _mod("fvcore")
    _mod("fvcore.nn", FlopCountAnalysis=None, flop_count_str=None, flop_count=None, parameter_count=None)
    _mod("yacs"); _mod("yacs.config", CfgNode=CfgNode)
    _mod("cv2", DMatch=DMatch, __version__="0.0.0-harness-stub", perspectiveTransform=perspective_transform, BFMatcher=BFMatcher,
         NORM_L2=4)
    _mod("h5py")
    _mod("GPUtil")
    for name in ("matplotlib", "matplotlib.pyplot", "matplotlib.image"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                _mod(name)
    tv = _mod("torchvision")
    ops = _mod("torchvision.ops", nms=greedy_nms)
    boxes = _mod("torchvision.ops.boxes", batched_nms=batched_nms)
    tv.ops = ops; ops.boxes = boxes
    tv.models = _mod("torchvision.models")

    _orig_device = torch.cuda.device

    def _device(d):
        dev = torch.device(d) if not isinstance(d, torch.device) else d
        if dev.type == "cpu":
            return contextlib.nullcontext()
        return _orig_device(d)

    torch.cuda.device = _device

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "xpoint"))
