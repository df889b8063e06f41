"""Data ingest for the inference path: folder mode of the reference's `ImagePairDataset`
(xpoint/datasets/ImagePairDataset.py:18-46 config, :48-74 folder structure, :122-128 member list,
:199-208 decode + gray + / 255, :254-274 crop to multiples of 32, :331-420 pair output dict).

What is mirrored: the config keys `foldername`, `height`, `width`, `single_image`, `random_pairs`, `return_name`,
the folder check (`optical/` + `thermal/` or `images/`), the same-shape check, the random crop (same `random.randint`
call order: i_h, then i_w) and the output structure `{'optical': {'image' (1,h,w) f32, 'valid_mask' (1,h,w) bool,
'is_optical'}, 'thermal': {...}, 'name'}` that `XPoint.forward` / `predict_align_image_pair` consume.
Not mirrored (training side, SURVEY.md 2 "out of scope"): the HDF5 file mode, keypoint label files, photometric /
homographic augmentation — a config that asks for them raises.

Decode: PIL (the image has no OpenCV).  Gray conversion and normalisation follow OpenCV's 8-bit `COLOR_BGR2GRAY`
(fixed point, 14 fractional bits) and numpy's `gray / 255.0` -> float32; restated, not pinned (cv2 absent).

Two ways to get a batch:
  * `ds[i]` — CPU tensors like the reference's `__getitem__` (host gray conversion; for DataLoader-style use);
  * `ds.load_batch(indices, device)` — host decode only, then ONE upload of the 8-bit pixels per image and the
    `xp_ingest_u8` kernel writes gray / 255 of the crop straight into the (B,1,h,w) batch tensors on the GPU.
"""
from __future__ import annotations

import copy
import ctypes
import os
import random

import numpy as np
import torch

from . import utils

# OpenCV's 8-bit BGR2GRAY coefficients (R2Y, G2Y, B2Y) * 2^14
_R2Y, _G2Y, _B2Y, _SHIFT = 4899, 9617, 1868, 14


def rgb_to_gray_u8(rgb: np.ndarray) -> np.ndarray:
    """(H, W, 3|4) uint8 R,G,B[,A] -> (H, W) uint8, = cv2.cvtColor(cv2.imread(f), cv2.COLOR_BGR2GRAY) of the same pixels."""
    if rgb.ndim == 2:
        return rgb
    r = rgb[..., 0].astype(np.int32); g = rgb[..., 1].astype(np.int32); b = rgb[..., 2].astype(np.int32)
    return ((b * _B2Y + g * _G2Y + r * _R2Y + (1 << (_SHIFT - 1))) >> _SHIFT).astype(np.uint8)


def gray_lut() -> np.ndarray:
    """float32(k / 255.0) for k = 0..255: numpy's `gray / 255.0` (float64) followed by `.astype(np.float32)`."""
    return (np.arange(256, dtype=np.float64) / 255.0).astype(np.float32)


class ImagePairDataset:
    default_config = {
        'filename': None, 'foldername': None, 'keypoints_filename': None,
        'height': -1, 'width': -1, 'raw_thermal': False,
        'single_image': False, 'random_pairs': False, 'return_name': True,
        'augmentation': {'photometric': {'enable': False}, 'homographic': {'enable': False}},
    }

    @staticmethod
    def check_folder_structure(folder_path):
        opt, th, img = (os.path.join(folder_path, d) for d in ("optical", "thermal", "images"))
        if os.path.isdir(opt) and os.path.isdir(th):
            return (opt, th)
        if os.path.isdir(img):
            return (img, img)
        raise ValueError(f"Folder structure is not correct.\nExpected:\n- {os.path.basename(folder_path)} (root)\n"
                         "  - optical\n  - thermal\nOR\n  - images")

    def __init__(self, config):
        self.config = utils.dict_update(copy.deepcopy(self.default_config), config or {})
        if self.config['filename'] is not None:
            raise NotImplementedError("ImagePairDataset: only the folder mode is implemented (h5py is not available here)")
        if self.config['foldername'] is None:
            raise ValueError("ImagePairDataset: The dataset filename XOR foldername needs to be present in the config file")
        if self.config['keypoints_filename'] is not None:
            raise NotImplementedError("ImagePairDataset: keypoint label files belong to the training path (out of scope)")
        aug = self.config['augmentation']
        if aug['photometric'].get('enable') or aug['homographic'].get('enable'):
            raise NotImplementedError("ImagePairDataset: augmentation belongs to the training path (out of scope)")
        if self.config['single_image']:
            raise NotImplementedError("ImagePairDataset: single_image mode belongs to the training path; inference consumes pairs")
        if not os.path.exists(self.config['foldername']):
            raise ValueError("The folder {} does not exists.".format(self.config['foldername']))
        self.data_path = self.check_folder_structure(self.config['foldername'])
        # the reference keeps os.listdir order (filesystem dependent); sorted here so that runs are reproducible
        self.memberslist = sorted(f for f in os.listdir(self.data_path[0]) if f.endswith(".jpg") or f.endswith(".png"))
        self.num_files = len(self.memberslist)
        self._lut_dev = {}

    def __len__(self):
        return self.num_files

    # ---- host side -------------------------------------------------------------------------------------------
    def _decode(self, index):
        """8-bit pixels of both images as decoded ((H, W) gray or (H, W, 3) RGB)."""
        from PIL import Image
        out = []
        for root in self.data_path:
            with Image.open(os.path.join(root, self.memberslist[index])) as im:
                if im.mode not in ("L", "RGB"):
                    im = im.convert("RGB")         # cv2.imread's default flag also yields 3 x 8 bit
                out.append(np.array(im))            # a writable, contiguous copy (torch.from_numpy needs one)
        if out[0].shape[:2] != out[1].shape[:2]:
            raise ValueError('ImagePairDataset: The optical and thermal image must have the same shape')
        return out

    def _crop_window(self, H0, W0):
        """Reference :254-274: crop size = requested size rounded down to a multiple of 32; random offset."""
        if self.config['height'] > 0 or self.config['width'] > 0:
            h = self.config['height'] // 32 * 32 if self.config['height'] > 0 else H0
            w = self.config['width'] // 32 * 32 if self.config['width'] > 0 else W0
            if w > W0 or h > H0:
                raise ValueError('ImagePairDataset: Requested height/width exceeds original image size')
            i_h = random.randint(0, H0 - h)
            i_w = random.randint(0, W0 - w)
            return i_h, i_w, h, w
        return 0, 0, H0, W0

    def _flags(self):
        o, t, swap_o, swap_t = True, False, False, False
        if self.config['random_pairs']:      # reference :345-354
            if bool(random.randint(0, 1)):
                o, swap_o = False, True
            if bool(random.randint(0, 1)):
                t, swap_t = True, True
        return o, t, swap_o, swap_t

    def __getitem__(self, index):
        opt8, th8 = self._decode(index)
        optical = rgb_to_gray_u8(opt8) / 255.0
        thermal = rgb_to_gray_u8(th8) / 255.0
        i_h, i_w, h, w = self._crop_window(*thermal.shape)
        optical = optical[i_h:i_h + h, i_w:i_w + w]
        thermal = thermal[i_h:i_h + h, i_w:i_w + w]
        o_flag, t_flag, swap_o, swap_t = self._flags()
        tmp_o, tmp_t = optical, thermal
        if swap_o:
            optical = tmp_t
        if swap_t:
            thermal = tmp_o
        out = {'optical': {}, 'thermal': {}}
        for key, img, flag in (('optical', optical, o_flag), ('thermal', thermal, t_flag)):
            out[key]['image'] = torch.from_numpy(np.expand_dims(img, 0).astype(np.float32))
            out[key]['valid_mask'] = torch.ones((1, h, w), dtype=torch.bool)
            out[key]['is_optical'] = torch.BoolTensor([flag])
        if self.config['return_name']:
            out['name'] = self.memberslist[index]
        return out

    # ---- device side -----------------------------------------------------------------------------------------
    def load_batch(self, indices, device="cuda:0"):
        """Decode on the host, convert / normalise / crop on the GPU: returns the batched data dict of `XPoint.forward`
        ({'optical': {'image' (B,1,h,w), 'valid_mask', 'is_optical' (B,1)}, 'thermal': {...}, 'name': [...]})."""
        from . import _lib as L
        device = torch.device(device)
        if device.type != "cuda":
            raise L.XPointHipError("ImagePairDataset.load_batch needs a GPU device (no CPU fallback); use ds[i] for CPU tensors")
        if device not in self._lut_dev:
            self._lut_dev[device] = torch.from_numpy(gray_lut()).to(device)
        lut = self._lut_dev[device]
        imgs = {'optical': None, 'thermal': None}
        flags = {'optical': [], 'thermal': []}
        names = []
        st = L.current_stream(device)
        keep = []
        for bi, index in enumerate(indices):
            opt8, th8 = self._decode(index)
            i_h, i_w, h, w = self._crop_window(*th8.shape[:2])
            o_flag, t_flag, swap_o, swap_t = self._flags()
            if imgs['optical'] is None:
                for k in imgs:
                    imgs[k] = torch.empty((len(indices), 1, h, w), dtype=torch.float32, device=device)
            elif tuple(imgs['optical'].shape[2:]) != (h, w):
                raise ValueError("ImagePairDataset.load_batch: all crops of a batch must have the same size (set height / width)")
            src = {'optical': th8 if swap_o else opt8, 'thermal': opt8 if swap_t else th8}
            for key in ('optical', 'thermal'):
                a = src[key]
                d = torch.from_numpy(a).to(device, non_blocking=True)
                keep.append(d)
                ch = 1 if a.ndim == 2 else a.shape[2]
                L.call("xp_ingest_u8", ctypes.c_void_p(d.data_ptr()), a.shape[0], a.shape[1], ch, i_h, i_w, h, w, L.ptr(lut),
                       ctypes.c_void_p(imgs[key][bi].data_ptr()), st)
            flags['optical'].append([o_flag]); flags['thermal'].append([t_flag])
            names.append(self.memberslist[index])
        torch.cuda.current_stream(device).synchronize()      # the staging buffers in `keep` may go now
        out = {}
        for key in ('optical', 'thermal'):
            out[key] = {'image': imgs[key], 'valid_mask': torch.ones_like(imgs[key], dtype=torch.bool),
                        'is_optical': torch.tensor(flags[key], dtype=torch.bool, device=device)}
        if self.config['return_name']:
            out['name'] = names
        return out
