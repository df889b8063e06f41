"""The reference's two prediction flows as functions (they are scripts upstream):
  predict_align_image_pair  — reference predict_align_image_pair.py:176-264 (= benchmark_evaluation.py:16-142,
                              demo.py:30-69,427-462): forward -> prob*mask -> box_nms -> nonzero ->
                              interpolate_descriptors -> get_matches.
  predict_keypoints         — reference predict_keypoints.py:144-216: NMS on the UNMASKED prob, mask applied at
                              extraction.
plus `PairPipeline`, the batched device-resident form of the first flow used by bench.py (no host
synchronisation between stages, preallocated buffers)."""
from __future__ import annotations

import copy
import ctypes

import torch

from . import _lib
from . import utils
from ._lib import c_i, ptr

# reference configs/cipdp.yaml:47-61
DEFAULT_PREDICTION = dict(detection_threshold=0.015, nms=8, cpu_nms=True, topk=0,
                          matching=dict(method="bfmatcher", knn_matches=False, method_kwargs=dict(crossCheck=True)))


def _cfg(pred):
    c = copy.deepcopy(DEFAULT_PREDICTION)
    if pred:
        utils.dict_update(c, pred)
    return c


def predict_align_image_pair(net, data, cfg_prediction=None, match_mode="strict_mnn", estimate_homography=False, warp_optical=None):
    """data: the reference pair dict ({'optical': {'image','valid_mask',...}, 'thermal': {...}}) on the GPU.
    Returns (out_optical, out_thermal, results) where results[i] = dict(kp_optical, kp_thermal (N,2) int64 (y,x),
    desc_optical, desc_thermal (N,D), matches [DMatch]).  estimate_homography=True adds the registration step of
    predict_align_image_pair.py:287-303: H_est (3,3) float64 mapping optical (x, y) to thermal (identity when fewer than
    4 matches, as the reference) and matchesMask, from `utils.find_homography` at `reprojection_threshold` (default 3), and — unless
    warp_optical=False — the script's final product (predict_align_image_pair.py:271, 308): `warped_optical`, the optical image quantised
    to uint8 RGB as the reference builds `im_optical` and warped by H_est into the thermal frame (`utils.warp_perspective`: INTER_LINEAR,
    BORDER_CONSTANT; (H, W, 3) uint8 on the device).  warp_optical=True without estimate_homography is an error."""
    if warp_optical and not estimate_homography:
        raise ValueError("predict_align_image_pair: warp_optical needs estimate_homography=True (the warp uses H_est)")
    if warp_optical is None:
        warp_optical = estimate_homography
    pred = _cfg(cfg_prediction)
    if net.takes_pair():
        out_o, out_t, _ = net(data)
    else:
        out_o, out_t = net(data['optical']), net(data['thermal'])
    thr = pred['detection_threshold']
    H, W = data['optical']['image'].shape[2:]
    for out, spec in ((out_o, 'optical'), (out_t, 'thermal')):
        if pred['nms'] > 0:          # the mask is applied only on the NMS branch (predict_align_image_pair.py:194-205)
            out['prob'] = utils.box_nms(out['prob'] * data[spec]['valid_mask'], pred['nms'], thr, keep_top_k=pred['topk'],
                                        on_cpu=pred['cpu_nms'])
    results = []
    for i in range(out_o['prob'].shape[0]):
        ko = torch.nonzero((out_o['prob'][i].squeeze() > thr).float())
        kt = torch.nonzero((out_t['prob'][i].squeeze() > thr).float())
        do = utils.interpolate_descriptors_nhwc(ko, out_o['desc_nhwc'][i], H, W)
        dt = utils.interpolate_descriptors_nhwc(kt, out_t['desc_nhwc'][i], H, W)
        ms = utils.get_matches(do, dt, pred['matching']['method'], pred['matching']['knn_matches'], mode=match_mode,
                               **pred['matching']['method_kwargs'])
        r = dict(kp_optical=ko, kp_thermal=kt, desc_optical=do, desc_thermal=dt, matches=ms)
        if estimate_homography:
            import numpy as np
            H_est, mask = None, None
            if len(ms) >= 4:
                qi = torch.tensor([m.queryIdx for m in ms], device=ko.device); ti = torch.tensor([m.trainIdx for m in ms], device=ko.device)
                H_est, mask = utils.find_homography(ko[qi].flip(-1).float(), kt[ti].flip(-1).float(),
                                                    float(pred.get('reprojection_threshold', 3.0)))
            r["H_est"] = H_est if H_est is not None else np.eye(3)
            r["matchesMask"] = mask.ravel().tolist() if mask is not None else []
            if warp_optical:
                r["warped_optical"] = utils.warp_perspective(data['optical']['image'][i, 0], r["H_est"], quantise_u8=True, dst_channels=3)
        results.append(r)
    return out_o, out_t, results


def predict_keypoints(net, data, cfg_prediction=None):
    """Returns ([kp per optical image], [kp per thermal image]); kp (N,2) int64 (y,x)."""
    pred = _cfg(cfg_prediction)
    if net.takes_pair():
        out_o, out_t, _ = net(data)
    else:
        out_o, out_t = net(data['optical']), net(data['thermal'])
    thr = pred['detection_threshold']
    res = []
    for out, spec in ((out_o, 'optical'), (out_t, 'thermal')):
        p = out['prob']
        if pred['nms'] > 0:
            p = utils.box_nms(p, pred['nms'], thr, keep_top_k=pred['topk'], on_cpu=pred['cpu_nms'])
        m = data[spec]['valid_mask']
        res.append([torch.nonzero((p[i].squeeze() > thr).float() * m[i].squeeze()) for i in range(p.shape[0])])
    return res[0], res[1]


class PairPipeline:
    """encode + detect + describe + match for a batch of B pairs, entirely stream-ordered on the device.

    run() enqueues: one 2B-image forward, prob*mask, box NMS (fixed number of sweeps, verified afterwards),
    keypoint extraction, descriptor sampling, mutual-NN matching.  Results stay on the GPU; `fetch()`
    synchronises, checks NMS convergence / capacity and returns host lists."""

    def __init__(self, net, batch, H, W, cap=8192, cfg_prediction=None, match_mode="strict_mnn", nms_sweeps=8, overlap=False,
                 split_encoder=False, estimate_homography=False, ransac_iters=10000, alternate_encoders=False, warp_optical=False):
        """overlap=True: two HIP streams — the encoder of call i+1 runs while the detection / matching kernels of call i
        (many small, latency-bound launches) are still in flight; encoder outputs are double-buffered.  Results of a
        call are complete after fetch() / torch.cuda.synchronize(), exactly as without overlap."""
        self.net, self.B, self.H, self.W, self.cap = net, int(batch), int(H), int(W), int(cap)
        self.overlap = bool(overlap)
        # split_encoder = S > 1: the 2B images go through the encoder as S groups on S streams (images are independent),
        # so the kernels of different groups fill each other's tails and latency-bound phases
        self.split_encoder = (int(split_encoder) if self.overlap else 0)
        if self.split_encoder in (0, 1) or (2 * self.B) % max(self.split_encoder, 1):
            self.split_encoder = 0
        # alternate_encoders: the encoders of consecutive calls run on two streams (call i on stream i & 1, whole batch, own
        # workspace), so TWO full-batch forwards are in flight next to the detection / matching of the call before — the
        # kernels keep their large-batch efficiency and fill each other's tails.  Results are those of the one-stream step.
        self.alternate = bool(alternate_encoders) and self.overlap and not self.split_encoder
        self.depth = (max(2, int(alternate_encoders)) if self.alternate else 2) if self.overlap else 1      # calls in flight = sets of input / encoder-output buffers
        self._call = 0
        self.pred = _cfg(cfg_prediction)
        self.mode = match_mode
        self.sweeps = int(nms_sweeps)
        dev = net._device if getattr(net, "_device", None) is not None and net._device.type == "cuda" else torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        with torch.cuda.device(dev):            # streams / events below belong to the model's device, whatever device is current
            self._init_buffers(net, H, W, estimate_homography, ransac_iters)
            # warp_optical: the registration output of predict_align_image_pair.py:308 for every pair, stream-ordered behind the robust fit —
            # the optical image (quantised to uint8 as the reference's im_optical, :271) warped by the DEVICE-resident H_est
            self.warp_optical = bool(warp_optical)
            if self.warp_optical:
                if not self.estimate_homography:
                    raise ValueError("PairPipeline: warp_optical needs estimate_homography=True")
                self.hg["warped"] = torch.empty((self.B, self.H, self.W), dtype=torch.uint8, device=dev)

    def _init_buffers(self, net, H, W, estimate_homography, ransac_iters):
        dev = self.device
        n = 2 * self.B
        lib = _lib.load()
        nbuf = self.depth
        self.images_b = [torch.empty((n, 1, H, W), device=dev) for _ in range(nbuf)]
        self.images = self.images_b[0]
        self.raw_b = [None] * nbuf
        self.mask_b = [None] * nbuf
        self.prob_nms = torch.empty((n, H, W), device=dev)
        self.prob_masked = torch.empty((n, H, W), device=dev)
        if self.overlap:
            self.enc_stream, self.post_stream = torch.cuda.Stream(), torch.cuda.Stream()
            if self.split_encoder:
                S = self.split_encoder
                self.enc_streams = [self.enc_stream] + [torch.cuda.Stream() for _ in range(S - 1)]
                self.encs_done = [[torch.cuda.Event() for _ in range(S)] for _ in range(nbuf)]
                self.group_ws = [torch.empty(net.workspace_bytes(n // S, H, W), dtype=torch.uint8, device=dev) for _ in range(S)]
            if self.alternate:
                self.enc_streams = [self.enc_stream] + [torch.cuda.Stream() for _ in range(nbuf - 1)]
                self.alt_ws = [torch.empty(net.workspace_bytes(n, H, W), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
            self.enc_done = [torch.cuda.Event() for _ in range(nbuf)]
            self.post_done = [torch.cuda.Event() for _ in range(nbuf)]
        self.nms_ws = torch.empty(lib.xp_box_nms_workspace_bytes(n, H, W, self.cap), dtype=torch.uint8, device=dev)
        self.kp = torch.zeros((n, self.cap, 2), dtype=torch.int32, device=dev)
        self.counts = torch.zeros((n,), dtype=torch.int32, device=dev)
        self.kp_ws = torch.empty(lib.xp_extract_keypoints_workspace_bytes(n, H, W), dtype=torch.uint8, device=dev)
        D = int(net.config['descriptor_size'])
        self.D = D
        self.desc = torch.zeros((n, self.cap, D), device=dev)
        P = self.B
        self.m = dict(idx12=torch.empty((P, self.cap), dtype=torch.int32, device=dev), dist12=torch.empty((P, self.cap), device=dev),
                      idx21=torch.empty((P, self.cap), dtype=torch.int32, device=dev), dist21=torch.empty((P, self.cap), device=dev),
                      match_q=torch.empty((P, self.cap), dtype=torch.int32, device=dev),
                      match_t=torch.empty((P, self.cap), dtype=torch.int32, device=dev),
                      match_d=torch.empty((P, self.cap), device=dev), match_count=torch.zeros((P,), dtype=torch.int32, device=dev))
        self.match_ws = torch.empty(lib.xp_match_workspace_bytes(P, self.cap, self.cap, D), dtype=torch.uint8, device=dev)
        # optional registration step (predict_align_image_pair.py:283-303) on the device: one batched robust fit per pair
        self.estimate_homography = bool(estimate_homography)
        self.ransac_iters = int(ransac_iters)
        if self.estimate_homography:
            self.hg = dict(src=torch.empty((P, self.cap, 2), device=dev), dst=torch.empty((P, self.cap, 2), device=dev),
                           H=torch.empty((P, 3, 3), dtype=torch.float64, device=dev), mask=torch.empty((P, self.cap), dtype=torch.uint8, device=dev),
                           n_inliers=torch.zeros((P,), dtype=torch.int32, device=dev),
                           ws=torch.empty(lib.xp_find_homography_workspace_bytes(P) // 8 + 1, dtype=torch.float64, device=dev))
        self.raw = None
        self.repaired = False        # set by verify(): the latest call was recomputed (range guard): earlier host downloads of it are stale
        self._last = None            # (buffer set, masked) of the latest call: what verify() re-runs when the range guard trips
        self._graphs = None
        self._capture_masked = None

    def run(self, optical, thermal, mask_optical=None, mask_thermal=None):
        """Input contract (also of capture() / the replay callable): optical and thermal are (B, 1, H, W) / (B, H, W) images of the SAME dtype —
        float32 in [0, 1] (device-resident: staged by one kernel; host / pinned: copied), or uint8 gray 0..255 (a quarter of the PCIe bytes; divided by 255
        on the device, the same f32 division the reference's loader does on the host).  A uint8 / float mix raises ValueError.  Masks: both or neither."""
        if (mask_optical is None) != (mask_thermal is None):
            raise ValueError("PairPipeline.run: pass both valid masks or neither")
        with torch.cuda.device(self.device):
            return self._run(optical, thermal, mask_optical, mask_thermal)

    def _run(self, optical, thermal, mask_optical, mask_thermal):
        if not self.overlap:
            self._stage_inputs(0, optical, thermal, mask_optical, mask_thermal)
            self._last = (0, mask_optical is not None)
            self._encode(0, None, None)
            return self._post(0, mask_optical is not None, None)
        k = self._call % self.depth
        self._call += 1
        self._last = (k, mask_optical is not None)
        cur = torch.cuda.current_stream()
        # inputs are taken over on the CALLER's stream (so the caller may reuse its tensors right after run() returns),
        # once the call before last has finished with buffer k
        cur.wait_event(self.post_done[k])
        self._stage_inputs(k, optical, thermal, mask_optical, mask_thermal)
        self.enc_stream.wait_stream(cur)
        if self.split_encoder:
            S = self.split_encoder
            if self.raw_b[k] is None:          # allocate the full-batch outputs once; the groups write into views of them
                with torch.cuda.stream(self.enc_stream):
                    self._encode(k, None, None)
                cur.wait_stream(self.enc_stream)
            g = 2 * self.B // S
            for h, stream in enumerate(self.enc_streams):
                stream.wait_stream(cur)
                with torch.cuda.stream(stream):
                    sl = slice(h * g, (h + 1) * g)
                    view = {kk: (v[sl] if v is not None else None) for kk, v in self.raw_b[k].items()}
                    fl = self._flags()
                    self.net.forward_raw(self.images_b[k][sl], want_prob=True, want_desc=True, out=view, workspace=self.group_ws[h],
                                         is_optical=None if fl is None else fl[sl], check=False, **self._status_kw())
                    self.encs_done[k][h].record()
                self.post_stream.wait_event(self.encs_done[k][h])
        else:
            stream = self.enc_streams[k] if self.alternate else self.enc_stream
            stream.wait_stream(cur)
            with torch.cuda.stream(stream):
                self._encode(k, None, None)
                self.enc_done[k].record()
            self.post_stream.wait_event(self.enc_done[k])
        with torch.cuda.stream(self.post_stream):
            self._post(k, mask_optical is not None, None)
            self.post_done[k].record()
        return self

    def _flags(self):
        """multispectral models: the first B images of the batch are optical, the last B thermal."""
        if not self.net.config.get('multispectral', False):
            return None
        return [True] * self.B + [False] * self.B

    def _stage_inputs(self, k, optical, thermal, mask_optical, mask_thermal):
        B, H, W = self.B, self.H, self.W
        img = self.images_b[k]
        masked = mask_optical is not None
        if masked and self.mask_b[k] is None:
            self.mask_b[k] = torch.empty((2 * B, H, W), dtype=torch.uint8, device=img.device)
        n = B * H * W

        def resident(t, dtype):
            return t.is_cuda and t.device == img.device and t.dtype == dtype and t.is_contiguous() and t.numel() == n
        if resident(optical, torch.float32) and resident(thermal, torch.float32) and \
                (not masked or (resident(mask_optical, torch.uint8) and resident(mask_thermal, torch.uint8))):
            # device-resident inputs: one launch instead of four runtime blits (a 9.8 MB image block takes the blit kernel 76 us)
            _lib.check(_lib.load().xp_stage_pair_batch(ptr(optical), ptr(thermal), ptr(img), ptr(mask_optical) if masked else None,
                                                       ptr(mask_thermal) if masked else None, ptr(self.mask_b[k]) if masked else None, n,
                                                       _lib.current_stream()), "xp_stage_pair_batch")
            return
        if (optical.dtype == torch.uint8) != (thermal.dtype == torch.uint8):
            # a uint8 image means "0..255 gray, divide by 255 on the device"; a float image means "already in [0, 1]".  One of each would put the two spectra
            # on different scales without any error: refuse it
            raise ValueError(f"PairPipeline: optical is {optical.dtype} and thermal is {thermal.dtype} — pass both images as uint8 (0..255, normalised on the "
                             "device) or both as float32 in [0, 1]")
        if optical.dtype == torch.uint8 and (optical.numel() != n or thermal.numel() != n):
            raise ValueError(f"PairPipeline: uint8 images must hold exactly {self.B} x {H} x {W} pixels each (got {optical.numel()} and {thermal.numel()})")
        if optical.dtype == torch.uint8:
            # 8-bit gray images (what a camera / decoder delivers): a quarter of the bytes over PCIe, gray / 255 on the device — the same f32 division the
            # reference's loader does on the host (xpoint/datasets/ImagePairDataset.py:254-274), so the staged images are the same bits
            if getattr(self, "_u8_b", None) is None:
                self._u8_b = [torch.empty((2 * B, H, W), dtype=torch.uint8, device=img.device) for _ in range(self.depth)]
            u8 = self._u8_b[k]
            u8[:B].copy_(optical.reshape(B, H, W), non_blocking=True)
            u8[B:].copy_(thermal.reshape(B, H, W), non_blocking=True)
            _lib.check(_lib.load().xp_u8_to_unit_f32(ctypes.c_void_p(u8.data_ptr()), ptr(img), 2 * n, _lib.current_stream()), "xp_u8_to_unit_f32")
        else:
            img[:B].copy_(optical, non_blocking=True)           # host (pinned) inputs, other dtypes / layouts: the runtime's copy engines
            img[B:].copy_(thermal, non_blocking=True)
        if masked:
            self.mask_b[k][:B].copy_(mask_optical.reshape(B, H, W))
            self.mask_b[k][B:].copy_(mask_thermal.reshape(B, H, W))

    def _encode(self, k, optical, thermal):
        if optical is not None:
            self._stage_inputs(k, optical, thermal, None, None)
        ws = self.alt_ws[k] if getattr(self, "alternate", False) else None
        # check=False: stream-ordered, no host synchronisation here; the forward's status word is read in verify() / fetch() / download_async()
        self._note_engine()
        self.raw_b[k] = self.net.forward_raw(self.images_b[k], want_prob=True, want_desc=True, out=self.raw_b[k], workspace=ws, is_optical=self._flags(),
                                             check=False, **self._status_kw())

    def _post(self, k, masked=False, _unused=None):
        B, H, W, n = self.B, self.H, self.W, 2 * self.B
        lib = _lib.load()
        st = _lib.current_stream()
        raw = self.raw_b[k]
        self.raw = raw
        self.images = self.images_b[k]
        prob = raw["prob"]
        thr = float(self.pred['detection_threshold'])
        if self.pred['nms'] > 0:
            if masked:          # the mask is applied on the NMS branch only, like predict_align_image_pair.py:194-205 (and predict_align_image_pair above)
                _lib.check(lib.xp_mul_mask(ptr(prob), ptr(self.mask_b[k]), ptr(self.prob_masked), n * H * W, st), "xp_mul_mask")
                prob = self.prob_masked
            _lib.check(lib.xp_box_nms(ptr(prob), ptr(self.prob_nms), ptr(self.nms_ws), self.nms_ws.numel(), n, H, W,
                                      float(self.pred['nms']), thr, 0.1, int(self.pred['topk']), self.cap, self.sweeps, None, st),
                       "xp_box_nms")
            prob = self.prob_nms
        _lib.check(lib.xp_extract_keypoints(ptr(prob), None, thr, ptr(self.kp), ptr(self.counts), n, H, W, self.cap, ptr(self.kp_ws),
                                            self.kp_ws.numel(), st),
                   "xp_extract_keypoints")
        d = raw["desc_nhwc"]
        _lib.check(lib.xp_sample_descriptors(ptr(self.kp), ptr(self.counts), ptr(d), ptr(self.desc), n, self.cap, d.shape[1],
                                             d.shape[2], self.D, H, W, st), "xp_sample_descriptors")
        m = self.m
        _lib.check(lib.xp_match_mnn(ptr(self.desc[:B]), ptr(self.desc[B:]), ptr(self.counts), 1, 0, B, B, self.cap, self.cap, self.D,
                                    utils.MATCH_MODES[self.mode], ptr(m["idx12"]), ptr(m["dist12"]), ptr(m["idx21"]), ptr(m["dist21"]),
                                    ptr(m["match_q"]), ptr(m["match_t"]), ptr(m["match_d"]), ptr(m["match_count"]),
                                    ptr(self.match_ws), self.match_ws.numel(), st), "xp_match_mnn")
        if self.estimate_homography:
            h = self.hg
            _lib.check(lib.xp_gather_match_points(ptr(self.kp), ptr(m["match_q"]), ptr(m["match_t"]), ptr(m["match_count"]), B, self.cap,
                                                  ptr(h["src"]), ptr(h["dst"]), st), "xp_gather_match_points")
            _lib.check(lib.xp_find_homography(ptr(h["src"]), ptr(h["dst"]), ptr(m["match_count"]), B, self.cap,
                                              float(self.pred.get('reprojection_threshold', 3.0)), self.ransac_iters, 0, ptr(h["H"]), ptr(h["mask"]),
                                              ptr(h["n_inliers"]), ptr(h["ws"]), h["ws"].numel() * 8, st), "xp_find_homography")
            if self.warp_optical:
                _lib.check(lib.xp_warp_perspective(ptr(self.images_b[k]), ctypes.c_void_p(h["warped"].data_ptr()), ctypes.c_void_p(h["H"].data_ptr()),
                                                   B, H, W, H, W, 1, 1, 2, 0, st), "xp_warp_perspective")
        return self

    def match_stats(self):
        """Nomination statistics of the latest step's matcher call (xp_match_stats): what its run time depends on (clustered descriptors nominate
        more), never its result.  Synchronises."""
        with torch.cuda.device(self.device):
            torch.cuda.synchronize()
            lib = _lib.load()
            out = torch.zeros(4, dtype=torch.int64, device=self.device)
            _lib.check(lib.xp_match_stats(ptr(self.match_ws), ptr(self.counts), 1, 0, self.B, self.B, self.cap, self.cap, self.D, ptr(out),
                                          _lib.current_stream()), "xp_match_stats")
            s, mx, ov, rows = (int(v) for v in out.cpu())
        return dict(mean=round(s / max(rows, 1), 3), max=mx, overflow_rows=ov, rows=rows, inline_capacity=int(lib.xp_match_cand_cap()))

    def wait(self):
        """Make the caller's current stream wait for everything run() has enqueued so far (overlapped mode: the
        detection / matching stream); after it, stream-ordered reads of the result tensors are safe."""
        if self.overlap and self._call:
            torch.cuda.current_stream().wait_event(self.post_done[(self._call - 1) % self.depth])
        return self

    def capture(self, optical, thermal, mask_optical=None, mask_thermal=None):
        """Capture the step into hipGraphs (fixed shapes, preallocated buffers) and return a replay callable that takes the same
        arguments as run().  The input copies stay outside the graphs (so the caller may pass different tensors, e.g. pinned
        host memory, at every replay).
        One stream: one graph of encoder + detection + matching.  Overlapped pipeline: per output buffer (two of them), one
        graph per encoder image group and one for the detection / matching kernels, each replayed on its own stream and
        chained by the same events as the eager schedule — the cross-step overlap survives capture."""
        with torch.cuda.device(self.device):
            for _ in range(self.depth):                         # warm-up outside capture: one-time allocations, every buffer set
                self._run(optical, thermal, mask_optical, mask_thermal)
            torch.cuda.synchronize()
            self._settle_engine()                               # a range-guard trip during warm-up switches the engine BEFORE anything is captured
            self._capture_masked = mask_optical is not None
            self._capture_graphs()
            if not self.overlap:
                def replay(optical, thermal, mask_optical=None, mask_thermal=None):
                    self._stage_inputs(0, optical, thermal, mask_optical, mask_thermal)
                    self._last = (0, mask_optical is not None)
                    self._note_engine(self._graph_engine)
                    self._graphs.replay()
                    return self
                return replay

            def replay(optical, thermal, mask_optical=None, mask_thermal=None):
                with torch.cuda.device(self.device):
                    k = self._call % self.depth
                    self._call += 1
                    self._last = (k, mask_optical is not None)
                    cur = torch.cuda.current_stream()
                    cur.wait_event(self.post_done[k])
                    self._stage_inputs(k, optical, thermal, mask_optical, mask_thermal)
                    enc_g, pg = self._graphs[k]
                    self._note_engine(self._graph_engine)
                    for h, stream in enumerate(self._streams_of(k)):
                        stream.wait_stream(cur)
                        with torch.cuda.stream(stream):
                            enc_g[h].replay()
                            self.encs_done[k][h].record() if self.split_encoder else self.enc_done[k].record()
                        self.post_stream.wait_event(self.encs_done[k][h] if self.split_encoder else self.enc_done[k])
                    with torch.cuda.stream(self.post_stream):
                        pg.replay()
                        self.post_done[k].record()
                    self.raw = self.raw_b[k]
                    self.images = self.images_b[k]
                return self
            return replay

    def _streams_of(self, k):
        return self.enc_streams if self.split_encoder else [self.enc_streams[k] if self.alternate else self.enc_stream]

    def _capture_graphs(self):
        """(Re)capture the step's graphs with the dense engine in force now; buffers exist (warm-up done), device synchronised."""
        masked = self._capture_masked
        self._graph_engine = self.net.effective_gemm_mode() if hasattr(self.net, "effective_gemm_mode") else None      # what every replay enqueues (_note_engine)
        pend = getattr(self, "_h2_pending", False)
        if not self.overlap:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._encode(0, None, None)
                self._post(0, masked, None)
            self._graphs = g
            self._h2_pending = pend                             # (capturing enqueues nothing)
            return
        S = max(self.split_encoder, 1)
        graphs = []
        for k in range(self.depth):
            enc_g = []
            gsz = 2 * self.B // S
            for h, stream in enumerate(self._streams_of(k)):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=stream):
                    if self.split_encoder:
                        sl = slice(h * gsz, (h + 1) * gsz)
                        view = {kk: (v[sl] if v is not None else None) for kk, v in self.raw_b[k].items()}
                        fl = self._flags()
                        self.net.forward_raw(self.images_b[k][sl], want_prob=True, want_desc=True, out=view, workspace=self.group_ws[h],
                                             is_optical=None if fl is None else fl[sl], check=False, **self._status_kw())
                    else:
                        self._encode(k, None, None)
                enc_g.append(g)
            pg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(pg, stream=self.post_stream):
                self._post(k, masked, None)
            graphs.append((enc_g, pg))
        self._graphs = graphs
        self._h2_pending = pend                                 # (capturing enqueues nothing)
        torch.cuda.synchronize()

    def status_word(self):
        """This pipeline's OWN forward status word (int32[1] on its device): its forwards OR their XP_STATUS_* bits into it and only its own
        verify() / fetch() reads and clears it — another pipeline on the same model, or an eager forward, cannot consume a trip raised by a step
        that is still in flight here.  Models without a status word (the conv back-bones) get the model's shared behaviour."""
        if getattr(self, "_status", None) is None:
            self._status = torch.zeros(1, dtype=torch.int32, device=self.device)
        return self._status

    def _status_kw(self):
        import inspect
        if not hasattr(self, "_status_ok"):
            self._status_ok = hasattr(self.net, "status_word") and "status" in inspect.signature(self.net.forward_raw).parameters
        return {"status": self.status_word()} if self._status_ok else {}

    def _note_engine(self, engine=None):
        """Record the dense engine of the forwards this pipeline enqueues (eager: the model's engine now; graph replay: the engine the graphs were
        captured on): a status trip is judged against THAT engine, not against whatever the shared model has been switched to since (_settle_engine)."""
        if engine is None:
            engine = self.net.effective_gemm_mode() if hasattr(self.net, "effective_gemm_mode") else None
        if engine == "h2":
            self._h2_pending = True
        self._engine_enqueued = engine

    def _handle_status_params(self):
        if getattr(self, "_hs_params", None) is None:
            import inspect
            self._hs_params = set(inspect.signature(self.net.handle_status).parameters)
        return self._hs_params

    def _settle_engine(self):
        """After a device-wide synchronisation: read and clear the forward status word.  Non-zero on the split-fp16 engine = an operand left
        the fp16 range: the model switches to "x3" for this weight set (warning), captured graphs are re-captured on it, and the latest call
        is computed again from its staged inputs — the results the caller reads next are never the overflowed ones.  Non-zero on any other
        engine raises (models.XPoint.handle_status).  Returns True when it re-ran."""
        if not hasattr(self.net, "status_word"):
            return False
        word = self.status_word()
        st = int(word.item())
        if st == 0:
            self._h2_pending = False                           # (device-wide synchronisation done: nothing enqueued on h2 is still in flight)
            return False
        word.zero_()
        # judged against the engine the pending forwards were enqueued with (some step in flight on "h2" -> recoverable, even when another user of
        # the same model has already switched it to "x3"); raises unless the h2 -> x3 switch applies
        pending = "h2" if getattr(self, "_h2_pending", False) else getattr(self, "_engine_enqueued", None)
        self._h2_pending = False
        retry = None
        if self._last is not None:
            def retry():          # the latest call's forward again (eager, from its staged inputs) on the model's current settings -> its status
                self._encode(self._last[0], None, None)
                torch.cuda.synchronize()
                v = int(word.item())
                word.zero_()
                return v
        # round 6: on "h2" the model first tries to LOCALISE the trip (only the out-of-range launches move to the x3 planes) before giving up the engine
        self.net.handle_status(st, "PairPipeline", pending, **({"retry": retry} if retry is not None and "retry" in self._handle_status_params() else {}))
        self._h2_pending = False                               # (the localisation's forwards were synchronised and read by retry())
        if self._graphs is not None:
            self._capture_graphs()
        if self._last is not None:
            k, masked = self._last
            self._encode(k, None, None)
            self._post(k, masked, None)
            torch.cuda.synchronize()
            st = int(word.item())
            word.zero_()
            self._h2_pending = False
            self.net.handle_status(st, "PairPipeline, second run")
        return True

    def download_async(self):
        """Streaming use: enqueue — right behind the detection / matching kernels of the last run() / replay — the copies of its result
        lists (keypoint counts and coordinates, match counts and index lists, distances; the homography fields when estimated) into
        pinned host buffers, and return (buffers, event).  The buffers are valid after event.synchronize(); the NEXT run() may be enqueued
        before that, so the device-to-host traffic of step i overlaps the steps behind it (`self.host_sets` = depth + 1 buffer sets rotate: a caller
        may keep `depth` steps in flight and must consume a set before the host_sets-th call after it).  The copies are ordered on the stream that
        writes the results, ahead of the next step's kernels."""
        with torch.cuda.device(self.device):
            if not hasattr(self, "_host"):
                def pin(t):
                    return torch.empty(t.shape, dtype=t.dtype, device="cpu").pin_memory()
                src = dict(counts=self.counts, kp=self.kp, match_count=self.m["match_count"], match_q=self.m["match_q"], match_t=self.m["match_t"],
                           match_d=self.m["match_d"])
                if hasattr(self.net, "status_word"):
                    src["status"] = self.status_word()                     # XP_STATUS_* bits of THIS pipeline's forwards so far: non-zero = call verify()
                if self.estimate_homography:
                    src.update(H_est=self.hg["H"], n_inliers=self.hg["n_inliers"], matchesMask=self.hg["mask"])
                self._host_src = src
                import os
                self._copy_engine_d2h = os.environ.get("XP_D2H_COPY_ENGINE", "0") == "1"      # A/B: the runtime's copy engines for the result lists
                self.host_sets = max(2, self.depth + 1)
                self._host = [{k: pin(v) for k, v in src.items()} for _ in range(self.host_sets)]
                self._host_ev = [torch.cuda.Event() for _ in range(self.host_sets)]
                self._host_i = 0
            i = self._host_i
            self._host_i = (i + 1) % self.host_sets
            stream = self.post_stream if self.overlap else torch.cuda.current_stream()
            with torch.cuda.stream(stream):
                # by a kernel, not by the copy engines: a device-to-host copy queued behind this step's kernels would hold up the NEXT steps' image
                # uploads in the engines' queue (xp_copy_to_mapped_host; bench.py's streaming loop: 1 553 -> see profiles/r5_streaming_parts.txt)
                lib = _lib.load()
                for k, v in self._host_src.items():
                    h = self._host[i][k]
                    if v.is_contiguous() and v.numel() > 0 and v.data_ptr() % 16 == 0 and h.data_ptr() % 16 == 0 and not self._copy_engine_d2h:
                        _lib.check(lib.xp_copy_to_mapped_host(ctypes.c_void_p(v.data_ptr()), ctypes.c_void_p(h.data_ptr()), v.numel() * v.element_size(),
                                                              _lib.current_stream()), "xp_copy_to_mapped_host")
                    else:
                        h.copy_(v, non_blocking=True)
                self._host_ev[i].record()
            return self._host[i], self._host_ev[i]

    def verify(self):
        """After a synchronisation point: the async NMS must have reached its fixed point and no list may have
        overflowed its capacity.  Raises otherwise (the caller can re-run with more sweeps / capacity).  Also settles the dense engine's
        range guard (see _settle_engine): never raises for an fp16-range overflow, falls back instead."""
        lib = _lib.load()
        # range guard of the split-fp16 dense engine (include/xpoint_hip.h, xp_xpoint_forward_ex): the forwards' status word.  A trip re-runs the
        # latest call on "x3" (and keeps that engine), so the checks below — and the caller — see the repaired results.
        # SCOPE of the repair: only the LATEST call (self._last) is recomputed; earlier calls still in flight when the guard tripped (depth 2-3
        # bursts, streaming) ran on the overflowed engine and are NOT recomputed — a caller that keeps several steps in flight must treat every step
        # since its last verify() as suspect.  Host buffers already filled by download_async() are stale after a repair: `self.repaired` is set and the
        # caller must call download_async() again for the latest step (bufs["status"] != 0 is the trigger, see download_async).
        self.repaired = self._settle_engine()
        if self.pred['nms'] > 0:
            left = c_i(-1)
            _lib.check(lib.xp_box_nms_check(ptr(self.nms_ws), 2 * self.B, self.H, self.W, ctypes.byref(left), _lib.current_stream()),
                       "xp_box_nms_check")
            while left.value != 0:
                # The stream-ordered NMS enqueues a fixed number of sweeps (sweeps past the fixed point exit at once); an image whose suppression
                # chains need more is NOT an error of the data: raise the count (kept for later calls), recompute the latest call's detection /
                # matching stages from its encoder outputs, and check again — the caller never sees a half-converged heat map.
                if self.sweeps >= 64 or self._last is None:
                    raise RuntimeError(f"PairPipeline: NMS not converged after {self.sweeps} sweeps ({left.value} tiles undecided)")
                import warnings
                warnings.warn(f"PairPipeline: NMS needed more than {self.sweeps} sweeps ({left.value} tiles undecided); re-running the step's "
                              f"post-processing with {min(64, 2 * self.sweeps)} and keeping that count", RuntimeWarning, stacklevel=2)
                self.sweeps = min(64, 2 * self.sweeps)
                if self._graphs is not None:
                    self._capture_graphs()
                k, masked = self._last
                self._post(k, masked, None)
                torch.cuda.synchronize()
                _lib.check(lib.xp_box_nms_check(ptr(self.nms_ws), 2 * self.B, self.H, self.W, ctypes.byref(left), _lib.current_stream()),
                           "xp_box_nms_check")
        mx = int(self.counts.max().item())
        if mx > self.cap:
            raise RuntimeError(f"PairPipeline: {mx} keypoints exceed capacity {self.cap}")

    def fetch(self):
        torch.cuda.synchronize()        # device-wide: covers both streams of the overlapped mode
        self.verify()
        B = self.B
        counts = self.counts.cpu().numpy()
        mc = self.m["match_count"].cpu().numpy()
        out = []
        for i in range(B):
            no, nt, nm = int(counts[i]), int(counts[B + i]), int(mc[i])
            extra = {}
            if self.estimate_homography:
                extra = dict(H_est=self.hg["H"][i].cpu().numpy(), matchesMask=self.hg["mask"][i, :nm].cpu().numpy(),
                             n_inliers=int(self.hg["n_inliers"][i].item()))
                if self.warp_optical:
                    extra["warped_optical"] = self.hg["warped"][i].cpu()
            out.append(dict(extra, kp_optical=self.kp[i, :no].cpu().long(), kp_thermal=self.kp[B + i, :nt].cpu().long(),
                            desc_optical=self.desc[i, :no].cpu(), desc_thermal=self.desc[B + i, :nt].cpu(),
                            match_q=self.m["match_q"][i, :nm].cpu().numpy(), match_t=self.m["match_t"][i, :nm].cpu().numpy(),
                            match_d=self.m["match_d"][i, :nm].cpu().numpy(),
                            idx12=self.m["idx12"][i, :no].cpu().numpy(), idx21=self.m["idx21"][i, :nt].cpu().numpy()))
        return out
