"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

The hot path shards by image pair (independent units): there is NO data-path collective.
The only collective is the one-time broadcast of the packed weight blob (81.6 MB fp32) from rank 0 —
the reference has nothing to mirror here (it only knows nn.DataParallel, train.py:149-151)."""
from __future__ import annotations

from typing import Tuple

import torch


def shard_pairs(total_pairs: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block of pair indices [first, first + count) owned by `rank`.  Remainders go to the low ranks."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, rem = divmod(int(total_pairs), world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def broadcast_weights(net, state_dict_fn=None, src: int = 0, device=None, group=None):
    """Rank `src` loads the reference-format state dict (state_dict_fn() -> dict) and packs it into the
    device-format blob; every other rank allocates an empty blob of the same size; one broadcast; every rank
    adopts the blob.  Works with gloo on CPU tensors (tests) and nccl/RCCL on GPU tensors (bench.py)."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    device = torch.device(device) if device is not None else torch.device("cpu")
    if rank == src:
        if state_dict_fn is not None:
            net.load_state_dict(state_dict_fn(), strict=True)
        blob = net.pack_weights().to(device)
    else:
        blob = torch.empty(net.weights_numel(), dtype=torch.float32, device=device)
    global last_first_broadcast_ms
    last_first_broadcast_ms = None
    if world > 1 or (dist.is_initialized() and blob.is_cuda):
        # the FIRST broadcast of the job, timed on its own: it carries the communicator's lazy set-up (ring / tree construction, xGMI
        # channel allocation) — timed_broadcast() below reports the steady-state re-broadcast separately
        import time
        if blob.is_cuda:
            torch.cuda.synchronize(blob.device)
        t0 = time.perf_counter()
        dist.broadcast(blob, src, group=group)
        if blob.is_cuda:
            torch.cuda.synchronize(blob.device)
        last_first_broadcast_ms = (time.perf_counter() - t0) * 1e3
    if blob.is_cuda:
        net.set_weight_blob(blob)
    return blob


last_first_broadcast_ms = None       # wall time of the most recent broadcast_weights() collective on this rank (None: no collective ran)


def gather_floats(values, device=None, group=None):
    """All-gather of a short list of floats per rank (e.g. a rank's own timed-region seconds): returns [rank][i].  The only collectives of
    the bench besides the weight broadcast and the result headers; never inside a timed region."""
    import torch.distributed as dist
    mine = torch.tensor([float(v) for v in values], dtype=torch.float64,
                        device=torch.device(device) if device is not None else torch.device("cpu"))
    if not dist.is_initialized():
        return [mine.tolist()]
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, mine, group=group)
    return [t.tolist() for t in out]


def timed_broadcast(blob: torch.Tensor, src: int = 0, repeats: int = 5, group=None) -> float:
    """Average wall time (ms) of one re-broadcast of `blob` (same bytes, so every rank's copy stays what it is):
    device events on the current stream for GPU tensors (nccl/RCCL), perf_counter for CPU tensors (gloo).
    The first, untimed broadcast absorbs communicator set-up.  With one rank the collective is RCCL's
    single-rank path (no link traffic): the number then only shows that the code path runs."""
    import time
    import torch.distributed as dist
    if not dist.is_initialized():
        return 0.0
    dist.broadcast(blob, src, group=group)
    if blob.is_cuda:
        torch.cuda.synchronize(blob.device)
        dist.barrier(group=group)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(repeats):
            dist.broadcast(blob, src, group=group)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / repeats
    dist.barrier(group=group)
    t0 = time.perf_counter()
    for _ in range(repeats):
        dist.broadcast(blob, src, group=group)
    return (time.perf_counter() - t0) * 1e3 / repeats


def gather_headers(first_pair: int, pairs: int, keypoints: int, matches: int, device=None, group=None):
    """All-gather of the fixed-size result header (first pair index, pairs, keypoints, matches) of every rank.
    Returns a list of 4-tuples indexed by rank (a one-element list without an initialised process group)."""
    import torch.distributed as dist
    mine = torch.tensor([first_pair, pairs, keypoints, matches], dtype=torch.int64,
                        device=torch.device(device) if device is not None else torch.device("cpu"))
    if not dist.is_initialized():
        return [tuple(int(v) for v in mine.tolist())]
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, mine, group=group)
    return [tuple(int(v) for v in t.tolist()) for t in out]


def gather_strings(text: str, device=None, group=None, width: int = 256):
    """All-gather of one short string per rank (e.g. the rank's CPU-affinity mask) as fixed-width byte tensors: returns [rank] -> str."""
    import torch.distributed as dist
    raw = text.encode()[:width]
    mine = torch.zeros(width, dtype=torch.uint8, device=torch.device(device) if device is not None else torch.device("cpu"))
    if raw:
        mine[:len(raw)] = torch.tensor(list(raw), dtype=torch.uint8)
    if not dist.is_initialized():
        return [text[:width]]
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, mine, group=group)
    return [bytes(t.cpu().tolist()).rstrip(b"\0").decode(errors="replace") for t in out]
