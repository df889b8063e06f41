"""BASELINE config C5 as a callable: the streaming, hipGraph-replayed pair step with the homography-regression head.

Every step takes its B pairs from (pinned) host memory, replays the captured encode + detect + describe + match graphs of an overlapped
`PairPipeline`, runs the reference's RegNet head (xpoint/models/RegNet.py:7-52) on the 256x256 top-left crop of every pair — the only input size
that head is defined for (its FC layer is sized for a 32x32 encoder map, RegNet.py:38-52) — from a second captured graph, and
enqueues the downloads of the result lists and of `hm` into pinned host buffers behind the step's last kernel.  The host may keep `pipe.depth`
steps in flight (depth + 1 pinned buffer sets rotate): it consumes step i - depth + 1 while the newer ones run.  `bench.py --config c5` times exactly this object; tests/test_gpu_configs.py checks it against the
reference fixtures g15 (lists) and g21 (hm)."""
from __future__ import annotations

import ctypes
import os

import torch

from . import _lib
from .predict import PairPipeline

CROP = 256


class StreamingRegistrationStep:
    def __init__(self, pipe: PairPipeline, net_hm, warm_optical, warm_thermal, mask_optical=None, mask_thermal=None):
        """pipe: an overlapped PairPipeline (not yet captured); net_hm: models.XPoint with homography_regression_head.check on a 256x256
        configuration (same encoder weights); warm_*: device images (B,1,H,W) used for the warm-up passes of the two captures."""
        self.pipe, self.net_hm = pipe, net_hm
        B, dev = pipe.B, pipe.device
        self.B = B
        with torch.cuda.device(dev), torch.no_grad():
            self.replay = pipe.capture(warm_optical, warm_thermal, mask_optical, mask_thermal)
            self.crop_o = torch.empty((B, 1, CROP, CROP), device=dev)
            self.crop_t = torch.empty((B, 1, CROP, CROP), device=dev)
            self._cut(warm_optical, warm_thermal)
            for _ in range(2):                              # warm-up outside capture (allocations; a range-guard trip settles the engine here)
                self.hm = net_hm.predict_homography(self.crop_o, self.crop_t)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.hm = net_hm.predict_homography(self.crop_o, self.crop_t)
            self._hm_engine = net_hm.effective_gemm_mode()
            self.host_sets = max(2, pipe.depth + 1)         # as PairPipeline.download_async: `depth` steps may be in flight on the host side
            self.hm_host = [torch.empty(self.hm.shape, dtype=self.hm.dtype).pin_memory() for _ in range(self.host_sets)]
            self.hm_ev = [torch.cuda.Event() for _ in range(self.host_sets)]
        self._i = 0
        self._head_on_caller = os.environ.get("XP_C5_HEAD_STREAM", "post") == "caller"     # A/B knob: the round-4 placement

    def _cut(self, o_img, t_img):
        self.crop_o.copy_(o_img[:, :, :CROP, :CROP]); self.crop_t.copy_(t_img[:, :, :CROP, :CROP])

    def __call__(self, optical, thermal, mask_optical=None, mask_thermal=None):
        """Enqueue one step.  Returns (bufs, event, hm_host, hm_event): pinned host buffers of THIS step, valid after the two events.
        bufs["status"] != 0 (forward status word, see PairPipeline.download_async) means the caller must run verify() — and, when verify() reports
        `pipe.repaired`, fetch the latest step's lists again (pipe.download_async()): the pinned buffers it already holds were filled BEFORE the repair.
        Steps before the latest one are not recomputed by verify() (PairPipeline.verify)."""
        pipe = self.pipe
        with torch.cuda.device(pipe.device), torch.no_grad():
            self.replay(optical, thermal, mask_optical, mask_thermal)
            # The head runs on the pipeline's detection / matching stream, behind this step's matching: the runtime maps streams onto
            # GPU_MAX_HW_QUEUES (4) hardware queues, and a fifth busy stream shares a queue with one of the other four (measured: head on the caller's
            # stream 1 241 pairs/s, with 6 queues 1 402).  The crops are cut outside the graph: their source alternates between the pipeline's
            # input buffers, which the step `depth` calls later overwrites only after post_done[k] — re-recorded below, behind the head.
            head_stream = pipe.post_stream if (pipe.overlap and not self._head_on_caller) else torch.cuda.current_stream()
            head_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(head_stream):
                self._cut(pipe.images[:self.B], pipe.images[self.B:])
                self.graph.replay()
                j = self._i % self.host_sets
                self._i += 1
                # like the result lists (PairPipeline.download_async): by a kernel, so that no copy-engine transfer of this step queues ahead of the next uploads
                if self.hm.is_contiguous() and self.hm.data_ptr() % 16 == 0 and not getattr(pipe, "_copy_engine_d2h", False):
                    _lib.check(_lib.load().xp_copy_to_mapped_host(ctypes.c_void_p(self.hm.data_ptr()), ctypes.c_void_p(self.hm_host[j].data_ptr()),
                                                                  self.hm.numel() * self.hm.element_size(), _lib.current_stream()), "xp_copy_to_mapped_host")
                else:
                    self.hm_host[j].copy_(self.hm, non_blocking=True)
                self.hm_ev[j].record()
                if pipe.overlap and not self._head_on_caller:
                    pipe.post_done[pipe._last[0]].record()
            bufs, ev = pipe.download_async()
        return bufs, ev, self.hm_host[j], self.hm_ev[j]

    def verify(self):
        """At a synchronisation point: PairPipeline.verify() (NMS convergence, capacities, the dense engine's range guard incl. its x3
        fallback) and the same guard for the head's forward: a trip re-captures the head graph on "x3" and recomputes `hm` for the latest crops."""
        torch.cuda.synchronize()
        self.pipe.verify()
        word = self.net_hm.status_word(self.pipe.device)
        st = int(word.item())
        if st:
            word.zero_()

            def retry():          # the head's forward again, eagerly, on the model's current settings -> its status (round 6: the trip is localised first)
                with torch.cuda.device(self.pipe.device), torch.no_grad():
                    flags = [True] * self.B + [False] * self.B if self.net_hm.config['multispectral'] else None      # as predict_homography
                    self.net_hm.forward_raw(torch.cat([self.crop_o, self.crop_t], 0), want_prob=False, want_desc=False, is_optical=flags, check=False)
                    torch.cuda.synchronize()
                v = int(word.item())
                word.zero_()
                return v
            if self.net_hm.handle_status(st, "StreamingRegistrationStep head", retry=retry):
                with torch.cuda.device(self.pipe.device), torch.no_grad():
                    self.graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self.graph):
                        self.hm = self.net_hm.predict_homography(self.crop_o, self.crop_t)
                    self.graph.replay()
                    torch.cuda.synchronize()
        return self
