"""Host-side mirror of the reference model interface for the hot path:
`xpoint.models.XPoint` (reference xpoint/models/XPoint.py:28-214) — same constructor config,
`forward(data)`, `takes_pair()`, `get_encoder_downsample_ratio()`, `set_force_return_logits()`,
`load_state_dict()` with the reference's key names — over the HIP C ABI (include/xpoint_hip.h).

PyTorch here is plumbing only: device buffers, the weight re-layout at load time, streams.
There is no CPU path: forward() on CPU tensors raises.
"""
from __future__ import annotations

import collections
import copy
import ctypes
import os
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from ._lib import c_i, ptr
from .utils import dict_update
from .convmodels import SuperPointMagicLeap  # noqa: F401  (reference: xpoint.models.SuperPointMagicLeap)


class _ModelCfg(ctypes.Structure):
    _fields_ = [("embed_dim", c_i), ("n_stages", c_i), ("depths", c_i * 4), ("d_state", c_i), ("dt_rank", c_i),
                ("mlp_ratio", ctypes.c_float), ("head_channels", c_i), ("desc_size", c_i), ("det_channels", c_i)]


_LoadResult = collections.namedtuple("_IncompatibleKeys", ["missing_keys", "unexpected_keys"])

_DIR_ORDER = [0, 2, 1, 3]   # device order of the four scan routes (row pair, then column pair)
_DENSE_PRODUCTS = {"h2": 6, "x3": 6, "x2": 3, "bf16": 1, "f32": 6, "amp16": 6, "amp16f": 6}   # gemm_mode -> partial products of the split-bf16 kernels
_DENSE_ENGINE = {"h2": 1, "amp16": 1}                                 # gemm_mode -> xp_set_dense_engine (default 0 = x3)
# "amp16": the reference's mixed_precision deployment class (XPoint.py:182, autocast; half is its default dtype) — xp_set_amp_mode(1): fp16 rounding
# at every autocast boundary on the split-fp16 engine, weights of the convolutions / linear layers rounded to fp16 (what autocast casts).
# "amp16f": the same recipe with HALF STORAGE (xp_xpoint_forward_f16: fp16 tensors in HBM, one-product fp16 MFMA GEMMs, csrc/gemm_f16.hip) — the fast
# deployment class; same rounding points, same fixture (g20).


class XPoint(torch.nn.Module):
    # reference XPoint.py:29-59
    default_config = {
        'multispectral': True, 'descriptor_head': True, 'intepolation_mode': 'bilinear', 'descriptor_size': 256,
        'normalize_descriptors': True, 'final_batchnorm': True, 'reflection_pad': True, 'bn_first': False,
        'double_convolution': True, 'channel_version': 0, 'verbose': False, 'mixed_precision': False,
        'force_return_logits': False, 'takes_pair': False,
        'homography_regression_head': {'check': False, 'type': 'HomographyNet'},
        'use_attention': {'check': False, 'type': 'SimpleViT', 'height': 256, 'width': 256,
                          'pretrained': {'check': True, 'type_dir': "model_weights/swinv2-imagenet/base256"}},
    }

    def __init__(self, config=None):
        super().__init__()
        if config:
            self.config = dict_update(copy.deepcopy(self.default_config), config)
        else:
            self.config = copy.deepcopy(self.default_config)
        ua = self.config['use_attention']
        self._kind = "vmamba" if (ua['check'] and ua['type'] == 'VMamba') else ("conv" if not ua['check'] else None)
        if self._kind is None:
            raise NotImplementedError("xpoint_amd.models.XPoint implements the VMamba encoder (model_weights/XPoint-EXP1/params.yaml) "
                                      "and the conv encoder (model_weights/multipoint/params.yaml); SwinV2 is out of scope")
        if self.config['multispectral'] and self._kind != "vmamba":
            raise NotImplementedError("multispectral two-encoder routing (XPoint.py:284-305) is implemented for the VMamba encoder only")
        for k, v in (('reflection_pad', True), ('bn_first', False), ('final_batchnorm', True), ('descriptor_head', True),
                     ('normalize_descriptors', True)):
            if self.config[k] != v:
                raise NotImplementedError(f"config['{k}'] = {self.config[k]!r} is not implemented (XPoint-EXP1 uses {v!r})")
        if self.config['homography_regression_head']['check']:
            assert self.config['takes_pair'], "RegNet can only be used with takes_pair=True"       # XPoint.py:103
        self._ref_state: "collections.OrderedDict[str, torch.Tensor]" = collections.OrderedDict()
        self._blob: Optional[torch.Tensor] = None        # device-format weights (one float32 tensor); multispectral: the OPTICAL encoder + heads
        self._wsplit: Optional[torch.Tensor] = None      # split-bf16 copies of the GEMM weights, derived from _blob on the device
        self._blob_t: Optional[torch.Tensor] = None      # multispectral only: the THERMAL encoder + the same heads
        self._wsplit_t: Optional[torch.Tensor] = None
        self._amp_w: Dict[str, tuple] = {}               # gemm_mode "amp16": spectrum -> (blob with the autocast-cast tensors rounded to fp16, its split copies)
        # "h2" (default): dense layers on the f16 matrix pipe, f32 operands as two fp16 planes, 3 partial products (f32-grade: operand
        # error <= 2^-23, <= 2^-21 per product worst case, ~2^-25 typical, csrc/gemm_h2_core.h); "x3": bf16 matrix pipe, three exact planes,
        # 6 partial products (f32-grade, no range limit); both are pinned against the reference.  "f32": exact-f32 MFMA kernels;
        # "x2": 3 bf16 partial products (operands to 16 bits); "bf16": 1 product = bf16 operands, f32 accumulate.
        # h2 needs dense-layer operands below 65504: every forward reports a violation through a device status word
        # (xp_xpoint_forward_ex), and the host then re-runs on "x3" and stays there for this weight set (self._h2_off).
        self.gemm_mode = os.environ.get("XP_GEMM_MODE", "h2")
        # The reference wraps its forward in autocast when config['mixed_precision'] is set AND it runs on CUDA (XPoint.py:182); its CPU path — the parity
        # target — never does.  Default here: the f32 class whatever the flag says.  Opt in to the reference's GPU behaviour with
        # XP_HONOR_MIXED_PRECISION=1 (or net.use_config_precision()): mixed_precision: true then selects the fast half-storage class "amp16f".
        if os.environ.get("XP_HONOR_MIXED_PRECISION", "0") not in ("", "0") and "XP_GEMM_MODE" not in os.environ:
            self.use_config_precision()
        self._h2_off = False
        self._h2_mask = 0                 # dense launches of the forward that run on the x3 planes under the h2 engine (xp_set_dense_override)
        self._status: Dict[str, torch.Tensor] = {}
        # RegNet head beyond 256x256 (opt-in, NOT reference semantics: the reference's head only accepts 256x256 inputs, RegNet.py:38-52):
        # adaptive-average-pool the pooled cost-volume map to the 16x16 grid its FC layer was sized for (convmodels.regnet_forward)
        self.regnet_adaptive_pool = False
        self._device = torch.device("cpu")
        self._ws: Dict[str, torch.Tensor] = {}
        self._conv_impl = None
        self._regnet_w = None
        self.encoder_downsample_ratio = 8
        self.detector_head_last_dim = 65
        self.head_channels = 256
        self._ctx = None
        if self._kind == "conv":
            if self.config['channel_version'] != 0 or not self.config['double_convolution']:
                raise NotImplementedError("conv encoder: only channel_version 0 with double_convolution (multipoint params)")
            self.n_channels = [1, 64, 64, 128, 128]
            return
        vssm = ua['model_parameters']['MODEL']['VSSM']
        if vssm.get('SSM_FORWARDTYPE', 'v05_noz') != 'v05_noz' or float(vssm.get('SSM_RATIO', 1.0)) != 1.0:
            raise NotImplementedError("only SSM_FORWARDTYPE v05_noz / SSM_RATIO 1.0 (the XPoint config) is implemented")
        depths = list(vssm['DEPTHS'])
        if len(depths) != 4:
            # the heads expect EMBED_DIM / 2 channels at 1/8 resolution = depth_to_space(4) of a FOURTH stage (XPoint.py:112-125,
            # VMamba.py:1500-1505); the reference fails with a channel mismatch in the head convolution for any other depth list
            raise ValueError(f"XPoint: the VMamba encoder needs exactly 4 stages (DEPTHS has {len(depths)} entries)")
        cfg = _ModelCfg()
        cfg.embed_dim = int(vssm['EMBED_DIM']); cfg.n_stages = len(depths)
        for i, d in enumerate(depths):
            cfg.depths[i] = int(d)
        cfg.d_state = int(vssm['SSM_D_STATE'])
        cfg.dt_rank = 0 if vssm.get('SSM_DT_RANK', 'auto') == 'auto' else int(vssm['SSM_DT_RANK'])
        cfg.mlp_ratio = float(vssm.get('MLP_RATIO', 4.0))
        cfg.head_channels = 256; cfg.desc_size = int(self.config['descriptor_size']); cfg.det_channels = 65
        self._cfg = cfg
        self._ctx = ctypes.c_void_p()
        _lib.check(_lib.load().xp_ctx_create(ctypes.byref(cfg), ctypes.byref(self._ctx)), "xp_ctx_create")
        self._layout = self._read_layout()
        self.n_channels = [1, 64, 64, 128, cfg.embed_dim // 2]

    # ------------------------------------------------------------------ reference surface
    def takes_pair(self):
        return self.config['takes_pair']

    def get_encoder_downsample_ratio(self):
        return self.encoder_downsample_ratio

    def set_force_return_logits(self, value):
        if not isinstance(value, bool):
            raise ValueError('set_force_return_logits: The input value needs to be a bool')      # XPoint.py:175-179
        self.config['force_return_logits'] = value

    def __del__(self):
        try:
            if getattr(self, "_ctx", None):
                _lib.load().xp_ctx_destroy(self._ctx)
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def _read_layout(self):
        lib = _lib.load()
        out = collections.OrderedDict()
        name = ctypes.create_string_buffer(128)
        off = ctypes.c_size_t(); num = ctypes.c_size_t()
        for i in range(lib.xp_param_count(self._ctx)):
            _lib.check(lib.xp_param_info(self._ctx, i, name, 128, ctypes.byref(off), ctypes.byref(num)), "xp_param_info")
            out[name.value.decode()] = (off.value, num.value)
        return out

    def expected_keys(self):
        from .synth import conv_xpoint_state_spec, xpoint_state_spec
        return conv_xpoint_state_spec(self.config) if self._kind == "conv" else xpoint_state_spec(self.config)

    def state_dict(self, *a, **k):
        return collections.OrderedDict(self._ref_state)

    def load_state_dict(self, state_dict, strict=True):
        """Accepts the reference's key names.  Like nn.Module.load_state_dict:
        returns (missing_keys, unexpected_keys); raises on strict mismatch or shape mismatch."""
        spec = self.expected_keys()
        sd = {}
        for k, v in state_dict.items():
            # legacy VMamba renames (reference VMamba.py:1577-1583)
            k = k.replace(".ln_1.", ".norm.").replace(".self_attention.", ".op.")
            sd[k] = v
        missing = [k for k in spec if k not in sd]
        unexpected = [k for k in sd if k not in spec]
        for k, (shape, _) in spec.items():
            if k in sd and tuple(sd[k].shape) != tuple(shape):
                raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(sd[k].shape)} vs model {tuple(shape)}")
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:5]}... unexpected {unexpected[:5]}...")
        for k in spec:
            if k in sd:
                t = sd[k]
                t = torch.from_numpy(np.array(t, copy=True)) if isinstance(t, np.ndarray) else t
                self._ref_state[k] = t.detach().to("cpu").clone()
        self._blob = None
        self._wsplit = None
        self._blob_t = None
        self._wsplit_t = None
        self._amp_w = {}
        self._conv_impl = None
        self._regnet_w = None
        self._h2_off = False              # a new weight set gets the default engine back
        self._h2_mask = 0
        return _LoadResult(missing, unexpected)

    def _bn_affine(self, pre, eps=1e-5):
        s = self._ref_state
        scale = s[pre + "weight"].double() / torch.sqrt(s[pre + "running_var"].double() + eps)
        shift = s[pre + "bias"].double() - s[pre + "running_mean"].double() * scale
        return scale.float(), shift.float()

    # tensors autocast casts to half: the weight and bias arguments of every convolution / linear layer (incl. the conv1d forms of x_proj and
    # dt_projs, VMamba.py:605-610).  LayerNorm / BatchNorm parameters, dt_projs_bias, A_logs and Ds stay f32 (csms6s.py:47-52).
    _AMP_CAST = ("patch_embed.0.", "patch_embed.5.", "op.in_proj.weight", "op.conv2d.weight", "op.x_proj_weight", "op.dt_projs_weight",
                 "op.out_proj.weight", "mlp.fc1.", "mlp.fc2.", "downsample.1.", "_head_convolutions.1.", "_head_convolutions.4.")

    def _device_params(self, e="encoder.", amp=False):
        """reference state dict -> device-format tensors (names of csrc/model.cpp build_layout) for the encoder whose
        state_dict prefix is `e` ("encoder.", or "encoder_optical." / "encoder_thermal." when multispectral).
        amp: the tensors of _AMP_CAST rounded to fp16 first (the mixed-precision class: what autocast feeds the kernels)."""
        s = self._ref_state
        if amp:
            s = collections.OrderedDict((k, (v.to(torch.float16).to(torch.float32) if (v.is_floating_point() and any(t in k for t in self._AMP_CAST)) else v))
                                        for k, v in s.items())
        missing = [k for k, (_, kind) in self.expected_keys().items() if k not in s and kind != "bn_count"
                   and not k.startswith("hm_regressor.")]
        if missing:
            raise RuntimeError(f"XPoint: weights not loaded ({len(missing)} tensors missing, e.g. {missing[:3]}); "
                               "call load_state_dict first")
        out = {}
        w0 = s[e + "patch_embed.0.weight"].double().sum(dim=1)                 # gray replicated to 3 channels (VMamba.py:1509)
        out["stem.w"] = w0.permute(1, 2, 0).reshape(9, -1).float()
        out["stem.b"] = s[e + "patch_embed.0.bias"]
        out["stem.ln_w"] = s[e + "patch_embed.2.weight"]; out["stem.ln_b"] = s[e + "patch_embed.2.bias"]
        out["pe2.w"] = s[e + "patch_embed.5.weight"].permute(0, 2, 3, 1)
        out["pe2.b"] = s[e + "patch_embed.5.bias"]
        out["pe2.ln_w"] = s[e + "patch_embed.7.weight"]; out["pe2.ln_b"] = s[e + "patch_embed.7.bias"]
        for st in range(self._cfg.n_stages):
            for j in range(self._cfg.depths[st]):
                r = f"{e}layers.{st}.blocks.{j}."
                d = f"s{st}.b{j}."
                C = s[r + "norm.weight"].shape[0]
                out[d + "ln1_w"] = s[r + "norm.weight"]; out[d + "ln1_b"] = s[r + "norm.bias"]
                out[d + "in_w"] = s[r + "op.in_proj.weight"]
                out[d + "dw_w"] = s[r + "op.conv2d.weight"].reshape(C, 9).t()
                out[d + "xproj_w"] = s[r + "op.x_proj_weight"][_DIR_ORDER]
                out[d + "dt_w"] = s[r + "op.dt_projs_weight"][_DIR_ORDER].permute(0, 2, 1)          # (4, R, C): channel-contiguous for the scan kernels
                out[d + "dt_b"] = s[r + "op.dt_projs_bias"][_DIR_ORDER]
                out[d + "A"] = (-torch.exp(s[r + "op.A_logs"].float())).view(4, C, -1)[_DIR_ORDER]      # VMamba.py:619
                out[d + "D"] = s[r + "op.Ds"].view(4, C)[_DIR_ORDER]
                out[d + "onorm_w"] = s[r + "op.out_norm.weight"]; out[d + "onorm_b"] = s[r + "op.out_norm.bias"]
                out[d + "out_w"] = s[r + "op.out_proj.weight"]
                out[d + "ln2_w"] = s[r + "norm2.weight"]; out[d + "ln2_b"] = s[r + "norm2.bias"]
                out[d + "fc1_w"] = s[r + "mlp.fc1.weight"]; out[d + "fc1_b"] = s[r + "mlp.fc1.bias"]
                out[d + "fc2_w"] = s[r + "mlp.fc2.weight"]; out[d + "fc2_b"] = s[r + "mlp.fc2.bias"]
            if st < self._cfg.n_stages - 1:
                r = f"{e}layers.{st}.downsample."
                d = f"s{st}.ds."
                out[d + "w"] = s[r + "1.weight"].permute(0, 2, 3, 1); out[d + "b"] = s[r + "1.bias"]
                out[d + "ln_w"] = s[r + "3.weight"]; out[d + "ln_b"] = s[r + "3.bias"]
        det, dsc = "detector_head_convolutions.", "descriptor_head_convolutions."
        out["head.w"] = torch.cat([s[det + "1.weight"], s[dsc + "1.weight"]], 0).permute(0, 2, 3, 1)
        out["head.b"] = torch.cat([s[det + "1.bias"], s[dsc + "1.bias"]], 0)
        a, b = self._bn_affine(det + "3."), self._bn_affine(dsc + "3.")
        out["head.scale"] = torch.cat([a[0], b[0]]); out["head.shift"] = torch.cat([a[1], b[1]])
        out["det2.w"] = s[det + "4.weight"].reshape(s[det + "4.weight"].shape[0], -1); out["det2.b"] = s[det + "4.bias"]
        out["det2.scale"], out["det2.shift"] = self._bn_affine(det + "5.")
        out["desc2.w"] = s[dsc + "4.weight"].reshape(s[dsc + "4.weight"].shape[0], -1); out["desc2.b"] = s[dsc + "4.bias"]
        out["desc2.scale"], out["desc2.shift"] = self._bn_affine(dsc + "5.")
        return out

    def pack_weights(self, spectrum: Optional[str] = None, amp: bool = False) -> torch.Tensor:
        """The device-format blob on the CPU (one float32 vector); what rank 0 broadcasts over RCCL.
        multispectral models have two (spectrum = "optical" / "thermal": that encoder + the shared heads).
        amp=True: the blob of the mixed-precision class (see _device_params)."""
        total = _lib.load().xp_weights_numel(self._ctx)
        blob = torch.zeros(total, dtype=torch.float32)
        if self.config['multispectral']:
            if spectrum not in ("optical", "thermal"):
                raise RuntimeError("multispectral XPoint: pack_weights(spectrum='optical' | 'thermal')")
            dp = self._device_params(f"encoder_{spectrum}.", amp=amp)
        else:
            dp = self._device_params(amp=amp)
        for name, (off, num) in self._layout.items():
            t = dp[name].contiguous().float().reshape(-1)
            if t.numel() != num:
                raise RuntimeError(f"internal: {name} has {t.numel()} elements, layout expects {num}")
            blob[off:off + num] = t
        return blob

    def weights_numel(self) -> int:
        return int(_lib.load().xp_weights_numel(self._ctx))

    def set_weight_blob(self, blob: torch.Tensor):
        """Adopt an already packed device blob (e.g. received by RCCL broadcast).  Single-encoder models only."""
        if self.config['multispectral']:
            raise NotImplementedError("set_weight_blob: multispectral models hold two blobs; load the state_dict on every rank instead")
        assert blob.is_cuda and blob.dtype == torch.float32 and blob.numel() == self.weights_numel()
        self._blob = blob.contiguous()
        self._wsplit = None
        self._amp_w = {}                  # derived copies of the OLD weights (fp16-rounded blob, its split planes) must not survive a new blob
        self._h2_off = False              # and a new weight set gets the default engine back, as in load_state_dict
        self._h2_mask = 0
        self._device = blob.device

    def to(self, device=None, *a, **k):
        if device is not None:
            self._device = torch.device(device)
            if self._blob is not None and self._blob.device != self._device:
                self._blob = self._blob.to(self._device)
                self._wsplit = None
            if self._blob_t is not None and self._blob_t.device != self._device:
                self._blob_t = self._blob_t.to(self._device)
                self._wsplit_t = None
            self._conv_impl = None          # device copies of the conv-encoder / RegNet weights are rebuilt on the new device
            self._regnet_w = None
            self._ws.clear()
        return self

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    # ------------------------------------------------------------------ forward
    def _workspace(self, n_img, H, W, device):
        # one cached buffer per device, grown to the largest request seen (a mixed multispectral batch alternates between
        # two sub-batch sizes: no reallocation per call)
        nbytes = _lib.load().xp_forward_workspace_bytes(self._ctx, n_img, H, W)
        if nbytes == 0:
            raise RuntimeError(f"XPoint: invalid image size {H}x{W}")
        key = str(device)
        if key not in self._ws or self._ws[key].numel() < nbytes:
            self._ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return self._ws[key]

    def workspace_bytes(self, n_img, H, W) -> int:
        return int(_lib.load().xp_forward_workspace_bytes(self._ctx, n_img, H, W))

    # ------------------------------------------------------------------ range guard
    def effective_gemm_mode(self) -> str:
        """gemm_mode, with "h2" replaced by "x3" once this weight set has tripped the split-fp16 range guard."""
        return "x3" if (self.gemm_mode == "h2" and self._h2_off) else self.gemm_mode

    def use_config_precision(self):
        """Select the arithmetic class the REFERENCE would run on a GPU for this config: `mixed_precision: true` -> autocast (XPoint.py:182) = gemm_mode
        "amp16f" (VMamba encoder only; the conv backbones stay f32-grade), else the f32 class.  Returns self."""
        self.gemm_mode = "amp16f" if (self.config.get('mixed_precision') and self._kind == "vmamba") else "h2"
        return self

    def status_word(self, device) -> torch.Tensor:
        """The device status word (int32[1]) that every forward on `device` ORs its XP_STATUS_* bits into (sticky until cleared)."""
        device = torch.device(device)
        if device.index is None:                       # "cuda" and "cuda:0" must name the same word
            device = torch.device("cuda", torch.cuda.current_device())
        key = str(device)
        if key not in self._status:
            self._status[key] = torch.zeros(1, dtype=torch.int32, device=device)
        return self._status[key]

    _STATUS_BITS = ((1, "encoder output non-finite or beyond the dense engine's operand range"), (2, "non-finite heat-map logit"),
                    (4, "non-finite descriptor element"))

    N_DENSE_LAUNCHES = 47          # XP_DENSE_LAUNCHES of include/xpoint_hip.h (xp_set_dense_override numbering)
    MAX_LOCALISED = 6              # more out-of-range launches than this: the whole weight set moves to "x3" as before

    def engine_key(self) -> str:
        """What distinguishes two forwards of this model arithmetically: the effective gemm_mode plus, on "h2", the per-launch override mask.  A pipeline
        re-captures its graphs when this changes."""
        m = self.effective_gemm_mode()
        return f"h2+{self._h2_mask:x}" if (m == "h2" and self._h2_mask) else m

    def _localise_range_trip(self, retry) -> bool:
        """The range guard tripped on "h2".  Find the dense launches whose operands leave the fp16 range by BISECTION over the per-launch override mask
        (xp_set_dense_override: launches >= k on the x3 planes; the forward is clean exactly when every offender is among them) and keep only THOSE on
        x3 for this weight set — one layer re-routed instead of the whole weight set (−1 % instead of −23 %).  `retry()` re-runs the
        caller's forward with the model's current settings, synchronises, reads and clears the status word and returns it.  True = a mask that runs
        clean is installed (the caller re-runs once more to produce its results); False = nothing installed (genuinely non-finite, or more than
        MAX_LOCALISED offenders): the caller falls back to "x3"."""
        n = self.N_DENSE_LAUNCHES
        full = (1 << n) - 1
        keep = self._h2_mask

        def run(mask):
            self._h2_mask = mask
            return retry()
        try:
            if keep and run(keep) == 0:          # another user of this model has localised the trip since these forwards were enqueued
                return True
            found = keep
            if run(full) != 0:                   # every launch on x3 and still non-finite: not a range problem
                self._h2_mask = keep
                return False
            for _ in range(self.MAX_LOCALISED):
                lo, hi = 0, n                    # launches >= lo on x3: clean;  launches >= hi on x3: trips
                while hi - lo > 1:
                    mid = (lo + hi) // 2
                    if run(found | (full & ~((1 << mid) - 1))) == 0:
                        lo = mid
                    else:
                        hi = mid
                found |= 1 << lo                 # launch `lo` on h2 trips, with every later launch on x3: an offender
                if run(found) == 0:
                    self._h2_mask = found
                    return True
            self._h2_mask = keep
            return False
        except Exception:
            self._h2_mask = keep
            raise

    def handle_status(self, st: int, where: str, engine: str = None, retry=None) -> bool:
        """Host policy for a non-zero status.  On the split-fp16 engine: with `retry` (a callable that re-runs the caller's forward and returns its
        status) first try to LOCALISE the trip — keep the engine, send only the out-of-range launches to the x3 planes (_localise_range_trip; warning);
        otherwise, or when that fails, switch this weight set to "x3" (warning).  Either way return True = the caller re-runs.  On any other engine the
        outputs are genuinely non-finite: raise.
        engine: the engine the reporting forwards were ENQUEUED with (stream-ordered callers: a pipeline's steps in flight or its captured graphs) —
        default: the model's engine now.  A trip of forwards enqueued on "h2" is recoverable also when another caller has switched the model to
        "x3" in the meantime: no second switch, no second warning, the caller re-runs."""
        if st == 0:
            return False
        what = "; ".join(t for b, t in self._STATUS_BITS if st & b)
        eng = engine if engine is not None else self.effective_gemm_mode()
        if eng == "h2":
            if self.gemm_mode == "h2" and not self._h2_off:
                import warnings
                if retry is not None:
                    before = self._h2_mask
                    if self._localise_range_trip(retry):
                        if self._h2_mask != before:
                            ids = [i for i in range(self.N_DENSE_LAUNCHES) if (self._h2_mask >> i) & 1]
                            warnings.warn(f"xpoint_amd.XPoint ({where}): {what} on the split-fp16 dense engine (operands must stay below 65504); dense "
                                          f"launch(es) {ids} of the forward now run on the split-bf16 planes for this weight set, every other layer stays on "
                                          "'h2' (xp_set_dense_override)", RuntimeWarning, stacklevel=3)
                        return True
                warnings.warn(f"xpoint_amd.XPoint ({where}): {what} on the split-fp16 dense engine (operands must stay below 65504); "
                              "re-running on gemm_mode 'x3' (split-bf16, no range limit) and keeping it for this weight set", RuntimeWarning, stacklevel=3)
                self._h2_off = True
            return True
        raise RuntimeError(f"xpoint_amd.XPoint ({where}): {what} on gemm_mode {eng!r} — the weights or the input produce "
                           "non-finite activations")

    def forward_raw(self, images: torch.Tensor, want_prob=True, want_desc=True, want_logits=False, out=None, workspace=None,
                    is_optical=None, check=True, status=None):
        """See _forward_raw.  Runs with the images' device as the current device, so every launch below goes to THAT
        device's current stream (the model may live on cuda:N while another device is current).
        check=True (default; ignored during stream capture): read the forward's status word (one 4-byte download = a host
        synchronisation) and, if the split-fp16 engine's range guard tripped, run the forward again on "x3" — the call never returns
        overflowed results.  Stream-ordered callers (PairPipeline) pass check=False and test the word at their own synchronisation point.
        status: an int32[1] device tensor OWNED BY THE CALLER that this forward ORs its XP_STATUS_* bits into instead of the model's shared word
        (status_word): a pipeline with forwards in flight must not have its bits read and cleared by somebody else's check (another pipeline on the same
        model, an eager call); with check=True the given word is the one read and cleared."""
        if not images.is_cuda:
            raise RuntimeError("xpoint_amd.XPoint runs on the GPU only (no CPU fallback): move the data to 'cuda'")
        with torch.cuda.device(images.device):
            res = self._forward_raw(images, want_prob, want_desc, want_logits, out, workspace, is_optical, status)
            if check and not torch.cuda.is_current_stream_capturing():
                word = self.status_word(images.device) if status is None else status
                st = int(word.item())
                if st:
                    word.zero_()

                    def retry():
                        self._forward_raw(images, want_prob, want_desc, want_logits, res if out is None else out, workspace, is_optical, status)
                        v = int(word.item())
                        word.zero_()
                        return v
                    if self.handle_status(st, "forward", retry=retry):
                        res = self._forward_raw(images, want_prob, want_desc, want_logits, res if out is None else out, workspace, is_optical, status)
                        st = int(word.item())
                        word.zero_()
                        self.handle_status(st, "forward, second run")       # raises when "x3" is non-finite as well
            return res

    def _forward_raw(self, images: torch.Tensor, want_prob=True, want_desc=True, want_logits=False, out=None, workspace=None,
                     is_optical=None, status=None):
        """images (N,1,H,W) float32 on the GPU -> dict of NHWC device tensors (no layout exports):
        prob (N,H,W), desc_nhwc (N,Hc,Wc,D), enc_nhwc (N,Hc,Wc,E/2), logits_nhwc (N,Hc,Wc,65).
        out: a dict returned by an earlier call with the same shapes -> its tensors are overwritten instead of
        allocating new ones (VMamba branch; lets a caller double-buffer the outputs across streams).
        workspace: a uint8 device tensor of workspace_bytes(N, H, W) bytes owned by the caller (concurrent calls on
        different streams need one each); default: the model's own cached workspace.
        is_optical: multispectral models only (XPoint.py:284-305) — bool per image, True -> optical encoder, False ->
        thermal encoder (heads are shared); a uniform batch is one call, a mixed batch is gathered per spectrum."""
        if not images.is_cuda:
            raise RuntimeError("xpoint_amd.XPoint runs on the GPU only (no CPU fallback): move the data to 'cuda'")
        if self.training:
            raise RuntimeError("xpoint_amd.XPoint is inference-only: call .eval()")
        if images.dim() != 4 or images.shape[1] != 1:
            raise RuntimeError("image must be (B,1,H,W)")
        images = images.contiguous().float()
        dev = images.device
        if self._kind == "conv":
            if self._conv_impl is None:
                from .convmodels import ConvEncoderXPointImpl
                missing = [k for k, (_, kind) in self.expected_keys().items() if k not in self._ref_state and kind != "bn_count"]
                if missing:
                    raise RuntimeError(f"XPoint: weights not loaded ({len(missing)} tensors missing, e.g. {missing[:3]})")
                self._conv_impl = ConvEncoderXPointImpl(self._ref_state, dev)
            return self._conv_impl.forward_raw(images, want_logits=want_logits)
        thermal = False
        if self.config['multispectral']:
            if is_optical is None:
                raise RuntimeError("multispectral XPoint: forward_raw needs is_optical (one bool per image)")
            flags = torch.as_tensor(is_optical).reshape(-1).to(torch.bool).cpu()
            if flags.numel() != images.shape[0]:
                raise RuntimeError("is_optical must hold one flag per image")
            if bool(flags.all()):
                thermal = False
            elif not bool(flags.any()):
                thermal = True
            else:       # mixed batch: run each spectrum's images through its encoder and scatter the results back
                idx_o = torch.nonzero(flags).reshape(-1).to(dev); idx_t = torch.nonzero(~flags).reshape(-1).to(dev)
                ro = self._forward_raw(images[idx_o], want_prob, want_desc, want_logits, None, workspace, [True] * int(idx_o.numel()), status)
                rt = self._forward_raw(images[idx_t], want_prob, want_desc, want_logits, None, workspace, [False] * int(idx_t.numel()), status)
                res = {}
                for k, v in ro.items():
                    if v is None:
                        res[k] = None
                        continue
                    full = out[k] if (out is not None and out.get(k) is not None) else torch.empty((images.shape[0],) + tuple(v.shape[1:]), device=dev)
                    full[idx_o] = v; full[idx_t] = rt[k]
                    res[k] = full
                return res
        if thermal:
            if self._blob_t is None or self._blob_t.device != dev:
                self._blob_t = self.pack_weights("thermal").to(dev)
                self._wsplit_t = None
        elif self._blob is None or self._blob.device != dev:
            self._blob = self.pack_weights("optical" if self.config['multispectral'] else None).to(dev)
            self._wsplit = None
        n, _, H, W = images.shape
        lib = _lib.load()
        if self.gemm_mode not in _DENSE_PRODUCTS:
            raise RuntimeError(f"XPoint.gemm_mode must be one of {sorted(_DENSE_PRODUCTS)}, got {self.gemm_mode!r}")
        mode = self.effective_gemm_mode()
        split_mode = mode != "f32"
        amp = mode in ("amp16", "amp16f")
        fast16 = mode == "amp16f"
        blob = self._blob_t if thermal else self._blob
        ws_split = self._wsplit_t if thermal else self._wsplit
        if amp:
            key = ("thermal" if thermal else "optical") + str(dev)
            if key not in self._amp_w:
                if not self._ref_state:
                    raise RuntimeError("gemm_mode 'amp16' needs the reference state dict (load_state_dict): the fp16 rounding of the weights is applied to "
                                       "the reference tensors, not to the packed blob")
                b16 = self.pack_weights(("thermal" if thermal else "optical") if self.config['multispectral'] else None, amp=True).to(dev)
                self._amp_w[key] = (b16, None)
            blob, ws_split = self._amp_w[key]
        if fast16:
            # fp16 (N, K) copies of the GEMM weights, converted on the device from the amp blob (whose values are fp16-exact already)
            k16 = key + "/f16"
            if k16 not in self._amp_w:
                nb = lib.xp_f16_weights_bytes(self._ctx)
                w16 = torch.empty(nb, dtype=torch.uint8, device=dev)
                _lib.check(lib.xp_prepare_f16_weights(self._ctx, ptr(blob), ptr(w16), ctypes.c_size_t(nb), _lib.current_stream()), "xp_prepare_f16_weights")
                self._amp_w[k16] = (w16, None)
            w16 = self._amp_w[k16][0]
        if split_mode and ws_split is None and not fast16:
            nb = lib.xp_split_weights_bytes(self._ctx)
            ws_split = torch.empty(nb, dtype=torch.uint8, device=dev)
            _lib.check(lib.xp_prepare_split_weights(self._ctx, ptr(blob), ptr(ws_split), ctypes.c_size_t(nb),
                                                    _lib.current_stream()), "xp_prepare_split_weights")
            if amp:
                self._amp_w[key] = (blob, ws_split)
            elif thermal:
                self._wsplit_t = ws_split
            else:
                self._wsplit = ws_split
        wsplit = ptr(ws_split) if (split_mode and not fast16) else None
        Hc = c_i(); Wc = c_i(); Ce = c_i()
        _lib.check(lib.xp_forward_shapes(self._ctx, n, H, W, ctypes.byref(Hc), ctypes.byref(Wc), ctypes.byref(Ce)), "xp_forward_shapes")
        Hc, Wc, Ce = Hc.value, Wc.value, Ce.value
        ws = self._workspace(n, H, W, dev) if workspace is None else workspace
        if out is None:
            out = {"enc_nhwc": torch.empty((n, Hc, Wc, Ce), device=dev)}
            out["prob"] = torch.empty((n, H, W), device=dev) if want_prob else None
            out["desc_nhwc"] = torch.empty((n, Hc, Wc, self._cfg.desc_size), device=dev) if want_desc else None
            out["logits_nhwc"] = torch.empty((n, Hc, Wc, 65), device=dev) if want_logits else None
        elif tuple(out["enc_nhwc"].shape) != (n, Hc, Wc, Ce) or (want_prob and out.get("prob") is None) or \
                (want_desc and out.get("desc_nhwc") is None) or (want_logits and out.get("logits_nhwc") is None):
            raise RuntimeError("forward_raw(out=...): buffers do not match this call")
        # the precision class is process-wide in the library and read when a kernel is launched: set for the duration of this
        # (host-synchronous) enqueue, then back to the default.  Not safe against OTHER host threads enqueueing dense kernels at the
        # same time: one enqueueing thread per process (the reference's scripts are single-threaded; multi-GPU = one process per GPU)
        if fast16:      # its own entry point: no process-wide switches involved
            _lib.check(lib.xp_xpoint_forward_f16(self._ctx, ptr(blob), ptr(w16), ptr(images), n, H, W, ptr(ws), ctypes.c_size_t(ws.numel()),
                                                 ptr(out["prob"]), ptr(out["desc_nhwc"]), ptr(out["enc_nhwc"]), ptr(out["logits_nhwc"]),
                                                 ptr(self.status_word(dev) if status is None else status), _lib.current_stream()), "xp_xpoint_forward_f16")
            return out
        nprod = _DENSE_PRODUCTS[mode]
        engine = _DENSE_ENGINE.get(mode, 0)
        prev = int(lib.xp_get_dense_products())          # whatever XP_DENSE_PRODUCTS / an earlier caller left: restored afterwards
        prev_engine = int(lib.xp_get_dense_engine())
        if nprod != prev:
            _lib.call("xp_set_dense_products", nprod)
        if engine != prev_engine:
            _lib.call("xp_set_dense_engine", engine)
        prev_amp = int(lib.xp_get_amp_mode())
        if int(amp) != prev_amp:
            _lib.call("xp_set_amp_mode", int(amp))
        ovmask = self._h2_mask if mode == "h2" else 0          # launches of THIS weight set that left the fp16 range (handle_status)
        if ovmask:
            _lib.call("xp_set_dense_override", ovmask)
        try:
            _lib.check(lib.xp_xpoint_forward_ex(self._ctx, ptr(blob), wsplit, ptr(images), n, H, W, ptr(ws), ctypes.c_size_t(ws.numel()),
                                                ptr(out["prob"]), ptr(out["desc_nhwc"]), ptr(out["enc_nhwc"]), ptr(out["logits_nhwc"]),
                                                ptr(self.status_word(dev) if status is None else status), _lib.current_stream()), "xp_xpoint_forward_ex")
        finally:
            if ovmask:
                _lib.call("xp_set_dense_override", 0)
            if int(amp) != prev_amp:
                _lib.call("xp_set_amp_mode", prev_amp)
            if nprod != prev:
                _lib.call("xp_set_dense_products", prev)
            if engine != prev_engine:
                _lib.call("xp_set_dense_engine", prev_engine)
        return out

    @staticmethod
    def _nchw(t):
        n, h, w, c = t.shape
        y = torch.empty((n, c, h, w), device=t.device)
        _lib.call("xp_nhwc_to_nchw", ptr(t), ptr(y), c_i(n), c_i(h * w), c_i(c), _lib.current_stream())
        return y

    def _export(self, raw, sl):
        """reference forward_impl output dict (XPoint.py:311-323) for images raw[sl]."""
        logits_mode = self.config['force_return_logits']
        out = {'prob': None if logits_mode else raw["prob"][sl].unsqueeze(1),
               'logits': self._nchw(raw["logits_nhwc"][sl]) if logits_mode else None,
               'desc': self._nchw(raw["desc_nhwc"][sl]),
               'encoder_output': self._nchw(raw["enc_nhwc"][sl]),
               # extra (not in the reference): the NHWC descriptor volume the sampling kernel reads directly
               'desc_nhwc': raw["desc_nhwc"][sl]}
        return out

    def _flags(self, data):
        """is_optical column of a reference data dict (XPoint.py:293-296), or None for single-encoder models."""
        if not self.config['multispectral']:
            return None
        f = data['is_optical']
        if f.dim() == 1:
            f = f.unsqueeze(0)
        return f[:, 0]

    def forward_impl(self, data):
        lm = self.config['force_return_logits']
        raw = self.forward_raw(data['image'], want_prob=not lm, want_logits=lm, is_optical=self._flags(data))
        return self._export(raw, slice(0, data['image'].shape[0]))

    def forward(self, data):
        """reference XPoint.py:181-214.  With one shared encoder (multispectral False) the optical and thermal batches
        run as ONE 2B-image batch; with two encoders each spectrum's images go through its own (forward_raw)."""
        if not self.takes_pair():
            return self.forward_impl(data)
        io, it = data["optical"]["image"], data["thermal"]["image"]
        lm = self.config['force_return_logits']
        flags = None
        if self.config['multispectral']:
            flags = torch.cat([self._flags(data["optical"]).cpu(), self._flags(data["thermal"]).cpu()], 0)
        raw = self.forward_raw(torch.cat([io, it], 0), want_prob=not lm, want_logits=lm, is_optical=flags)
        B = io.shape[0]
        pred_optical = self._export(raw, slice(0, B))
        pred_thermal = self._export(raw, slice(B, B + it.shape[0]))
        pred_hm = None
        if self.config["homography_regression_head"]["check"]:
            from .convmodels import regnet_forward, regnet_weights
            if self.config["homography_regression_head"]["type"] != "RegNet":
                raise NotImplementedError("only homography_regression_head.type 'RegNet' (XPoint-EXP1 params) is implemented")
            if self._regnet_w is None:
                self._regnet_w = regnet_weights(self._ref_state, raw["enc_nhwc"].device)
            pred_hm = regnet_forward(self._regnet_w, raw["enc_nhwc"][:B], raw["enc_nhwc"][B:], adaptive_pool=self.regnet_adaptive_pool,
                                  enc_both=raw["enc_nhwc"] if raw["enc_nhwc"].shape[0] == 2 * B else None)
        return pred_optical, pred_thermal, pred_hm

    def predict_homography(self, optical, thermal):
        """Homography-regression head alone (the third value of forward(), reference XPoint.py:205-212 / RegNet.py:7-52) for callers that take
        keypoints and descriptors from another forward (bench config C5 runs the head on 256x256 crops, the only size the reference's RegNet
        is defined for): encoder + RegNet, the detector / descriptor heads are not evaluated.  optical, thermal: (B, 1, H, W) device tensors."""
        from .convmodels import regnet_forward, regnet_weights
        if not self.config["homography_regression_head"]["check"] or self.config["homography_regression_head"]["type"] != "RegNet":
            raise NotImplementedError("predict_homography needs homography_regression_head.check with type 'RegNet'")
        B = optical.shape[0]
        flags = [True] * B + [False] * thermal.shape[0] if self.config['multispectral'] else None
        raw = self.forward_raw(torch.cat([optical, thermal], 0), want_prob=False, want_desc=False, is_optical=flags)
        if self._regnet_w is None:
            self._regnet_w = regnet_weights(self._ref_state, raw["enc_nhwc"].device)
        return regnet_forward(self._regnet_w, raw["enc_nhwc"][:B], raw["enc_nhwc"][B:], adaptive_pool=self.regnet_adaptive_pool,
                                  enc_both=raw["enc_nhwc"] if raw["enc_nhwc"].shape[0] == 2 * B else None)
