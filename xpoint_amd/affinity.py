"""CPU affinity of a rank: the cores next to its GPU.  Imports nothing heavy (no torch, no HIP): bench.py calls pin_rank() in every rank
BEFORE torch is imported, so the interpreter's threads and the runtime's helper threads inherit the mask; it is never a re-exec.

Nothing in the reference to mirror (single process, nn.DataParallel, train.py:149-151); the hot path's host side — kernel enqueue of three
streams per step, pinned-memory staging of the PCIe-inclusive and streaming (C5) paths — is latency-sensitive, and on a two-socket 8-GPU node a
rank scheduled on the far socket pays a cross-socket hop on every doorbell write and staging copy.

Topology source: KFD (`/sys/class/kfd/kfd/topology/nodes/*/properties`: a node with simd_count > 0 is a GPU; `domain` + `location_id` give its PCI
address) -> `/sys/bus/pci/devices/<addr>/local_cpulist` (the CPUs of the GPU's NUMA node).  GPUs that share a NUMA node split its CPUs evenly, in
device order.  Everything is intersected with the mask the process already has (a container's cpuset); if sysfs gives nothing, the allowed CPUs
are split evenly among the local ranks."""
from __future__ import annotations

import glob
import os
import re
from typing import Dict, List, Optional, Sequence, Set


def parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]."""
    out: List[int] = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return sorted(set(out))


def format_cpulist(cpus: Sequence[int]) -> str:
    cpus = sorted(set(int(c) for c in cpus))
    runs, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        runs.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(runs)


def _node_props(path: str) -> Dict[str, int]:
    props: Dict[str, int] = {}
    try:
        for line in open(path):
            f = line.split()
            if len(f) == 2 and re.fullmatch(r"-?\d+", f[1]):
                props[f[0]] = int(f[1])
    except OSError:
        pass
    return props


def gpu_local_cpus(sysfs_root: str = "/sys") -> List[Optional[List[int]]]:
    """Per GPU, in KFD node order (= HIP device order when no *_VISIBLE_DEVICES reordering is in force): the CPUs of its NUMA node, or None when
    sysfs does not say."""
    nodes = glob.glob(os.path.join(sysfs_root, "class/kfd/kfd/topology/nodes/*/properties"))
    nodes.sort(key=lambda p: int(os.path.basename(os.path.dirname(p))))
    out: List[Optional[List[int]]] = []
    for p in nodes:
        props = _node_props(p)
        if props.get("simd_count", 0) <= 0:
            continue
        loc, dom = props.get("location_id"), props.get("domain", 0)
        cpus = None
        if loc is not None:
            addr = f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7}"
            try:
                cpus = parse_cpulist(open(os.path.join(sysfs_root, "bus/pci/devices", addr, "local_cpulist")).read()) or None
            except (OSError, ValueError):
                cpus = None
        out.append(cpus)
    return out


def visible_device_indices(n_gpus: int, env=None) -> List[int]:
    """Physical (KFD-order) index of every visible device, honouring ROCR_ / HIP_ / CUDA_VISIBLE_DEVICES when they are plain index lists."""
    env = os.environ if env is None else env
    idx = list(range(n_gpus))
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = env.get(var)
        if v is None:
            continue
        toks = [t.strip() for t in v.split(",") if t.strip() != ""]
        if all(re.fullmatch(r"\d+", t) for t in toks):
            idx = [idx[int(t)] for t in toks if int(t) < len(idx)]
        else:                       # UUID lists: the order cannot be resolved without the runtime — keep the count, no mapping
            idx = idx[:len(toks)]
    return idx


def plan(local_world: int, allowed: Set[int], gpu_cpus: Sequence[Optional[Sequence[int]]], device_of_rank: Optional[Sequence[int]] = None) -> List[List[int]]:
    """CPU set of every local rank.  Rank r drives visible device r (physical GPU device_of_rank[r]).  GPUs whose NUMA node is known get that
    node's allowed CPUs, shared evenly (contiguous chunks, rank order) among the ranks on the same node; the others split what is left of the
    allowed set.  No rank ever gets an empty set: a rank whose share would be empty keeps the whole allowed set."""
    allowed_sorted = sorted(allowed)
    if device_of_rank is None:
        device_of_rank = list(range(local_world))
    node_of: List[Optional[tuple]] = []
    for r in range(local_world):
        d = device_of_rank[r] if r < len(device_of_rank) else None
        cpus = gpu_cpus[d] if d is not None and d < len(gpu_cpus) and gpu_cpus[d] else None
        usable = tuple(c for c in (cpus or ()) if c in allowed)
        node_of.append(usable or None)
    result: List[Optional[List[int]]] = [None] * local_world
    groups: Dict[tuple, List[int]] = {}
    for r, node in enumerate(node_of):
        if node is not None:
            groups.setdefault(node, []).append(r)
    taken: Set[int] = set()
    for node, ranks in groups.items():
        per = len(node) // len(ranks)
        for k, r in enumerate(ranks):
            share = list(node[k * per:(k + 1) * per]) if per > 0 else list(node)
            result[r] = share
            taken.update(share)
    rest_ranks = [r for r in range(local_world) if result[r] is None]
    if rest_ranks:
        free = [c for c in allowed_sorted if c not in taken] or allowed_sorted
        per = len(free) // len(rest_ranks)
        for k, r in enumerate(rest_ranks):
            result[r] = free[k * per:(k + 1) * per] if per > 0 else list(free)
    return [sorted(s) if s else allowed_sorted for s in result]


def pin_rank(local_rank: int, local_world: int, sysfs_root: str = "/sys", apply: bool = True) -> Dict[str, object]:
    """Pin the calling process to its rank's CPU set (os.sched_setaffinity on itself: allowed for an ordinary user).  Returns what the bench line
    reports: {"cpus": "0-15", "count": 16, "source": "kfd+pci local_cpulist" | "even split of the allowed set", "applied": bool, "previous": "0-127"}."""
    try:
        allowed = set(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return {"cpus": None, "count": 0, "source": "sched_getaffinity unavailable", "applied": False}
    gpus = gpu_local_cpus(sysfs_root)
    dev = visible_device_indices(len(gpus)) if gpus else None
    sets = plan(local_world, allowed, gpus, dev)
    mine = sets[local_rank] if 0 <= local_rank < len(sets) else sorted(allowed)
    override = os.environ.get("XP_RANK_CPUS")            # experiments: "0-63,128-191" pins every rank to that list (intersected with the allowed set)
    if override:
        mine = [c for c in parse_cpulist(override) if c in allowed] or mine
    known = bool(gpus) and dev is not None and local_rank < len(dev) and dev[local_rank] < len(gpus) and bool(gpus[dev[local_rank]])
    info = {"cpus": format_cpulist(mine), "count": len(mine), "source": "kfd + pci local_cpulist" if known else "even split of the allowed set",
            "applied": False, "previous": format_cpulist(sorted(allowed))}
    if apply and mine:
        try:
            os.sched_setaffinity(0, mine)
            info["applied"] = True
        except OSError as e:
            info["error"] = repr(e)
    return info


def restore(previous: str) -> None:
    """Back to an earlier mask (bench.py: the CPU baseline runs on all host cores, whatever the rank was pinned to)."""
    try:
        os.sched_setaffinity(0, parse_cpulist(previous))
    except (AttributeError, OSError, ValueError):
        pass
