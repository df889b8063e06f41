"""Shared by the CPU (oracle) and GPU (HIP) tests of the stand-alone cross-scan / cross-merge operators: walks the cases of fixture
tests/golden/g22_cross_scan_ops.npz (made by oracle/refharness/make_golden.py: gen_g22 through the REAL reference's cross_scan_fn /
cross_merge_fn, csm_triton.py:501-517) and compares a callable against it bit for bit."""
import zlib

import numpy as np
import torch

from oracle.refharness.make_golden import g22_inputs

DTYPES = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}
SHAPES = {"2x3x5x7": (2, 3, 5, 7), "1x2x33x58": (1, 2, 33, 58), "27x253x57x58": (27, 253, 57, 58)}


def _bits(t):
    t = t.detach().cpu().contiguous()
    return t.view(torch.int16).numpy() if t.dtype in (torch.float16, torch.bfloat16) else t.numpy()


def cases(g, shapes=None):
    """-> (tag, shape, dtype name, kwargs) for every (layout, one_by_one, scans, dtype) combination the fixture holds."""
    seen = []
    for key in g.files:
        parts = key.split("/")
        tag, lay, obo, sc, dt, what = parts[:6]
        if what != "scan" or (shapes and tag not in shapes):
            continue
        c = (tag, SHAPES[tag], dt, dict(in_channel_first=lay[2] == "1", out_channel_first=lay[6] == "1", one_by_one=obo[3] == "1", scans=int(sc[1])))
        if c not in seen:
            seen.append(c)
    return seen


def check(g, scan_fn, merge_fn, device="cpu", shapes=None):
    """scan_fn / merge_fn(tensor, in_channel_first=, out_channel_first=, one_by_one=, scans=) -> tensor.  Returns the number of cases checked."""
    n = 0
    cache = {}
    for tag, shape, dt, kw in cases(g, shapes):
        if (tag, dt) not in cache:
            cache.clear()
            cache[(tag, dt)] = g22_inputs(shape, DTYPES[dt])
        x, x4 = cache[(tag, dt)]
        key = f"{tag}/in{int(kw['in_channel_first'])}out{int(kw['out_channel_first'])}/obo{int(kw['one_by_one'])}/s{kw['scans']}/{dt}"
        src = x4 if kw["one_by_one"] else x
        if not kw["in_channel_first"]:
            src = (src.permute(0, 3, 4, 1, 2) if kw["one_by_one"] else src.permute(0, 2, 3, 1)).contiguous()
        yin = x4 if kw["out_channel_first"] else x4.permute(0, 3, 4, 1, 2).contiguous()
        for what, fn, inp in (("scan", scan_fn, src), ("merge", merge_fn, yin)):
            got = _bits(fn(inp.to(device), **kw))
            k = f"{key}/{what}"
            if k in g.files:
                # (one_by_one, scans 1: the reference returns `x.flatten(2, 3)` = (B, 4, C H, W) for a (B,4,C,H,W) input, csm_triton.py:103 — the same
                # bytes as (B, 4, C, L); only that quirk's shape is not required)
                assert got.shape == g[k].shape or (kw["one_by_one"] and kw["scans"] == 1 and got.size == g[k].size), (k, got.shape, g[k].shape)
                assert np.array_equal(got.reshape(-1), g[k].reshape(-1)), k
            else:
                crc, size = (int(v) for v in g[k + "/crc"])
                assert got.size == size and zlib.crc32(got.tobytes()) == crc, k
                assert np.array_equal(got.reshape(-1)[:16], g[k + "/head"]), k
            n += 1
    return n
