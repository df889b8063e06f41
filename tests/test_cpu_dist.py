"""CPU, world_size 2, gloo: the N > 1 path of bench.py — pair sharding (no data-path collective) and the one-time
weight broadcast — is correct by construction."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from xpoint_amd import dist as xdist
from xpoint_amd import synth


def test_shard_pairs_partition():
    for total, world in [(64, 8), (8, 1), (10, 4), (3, 8), (0, 2)]:
        seen = []
        for r in range(world):
            first, cnt = xdist.shard_pairs(total, world, r)
            seen += list(range(first, first + cnt))
        assert seen == list(range(total))
    assert xdist.shard_pairs(64, 8, 3) == (24, 8)
    with pytest.raises(ValueError):
        xdist.shard_pairs(8, 2, 2)


def test_headers_without_process_group():
    assert xdist.gather_headers(5, 8, 33000, 9000) == [(5, 8, 33000, 9000)]
    assert xdist.timed_broadcast(torch.zeros(4)) == 0.0
    assert xdist.gather_strings("0-7") == ["0-7"]


def test_bench_self_launch_refuses_missing_gpus():
    """`python bench.py --gpus N` outside torch.distributed.run starts the ranks itself; with fewer than N GPUs
    visible (none in the CPU container) it must exit 3 with a clear message BEFORE touching any device."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    n = torch.cuda.device_count() + 1 if torch.cuda.device_count() >= 1 else 2
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "1"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode == 3, (r.returncode, r.stderr[-400:])
    assert f"needs {n} GPUs" in r.stderr


def test_bench_parent_stays_gpu_free_and_watchdog_kills_the_group(tmp_path):
    """The parent of an unwrapped `bench.py --gpus N` must never load the HIP / HSA runtime: it counts GPUs from the KFD
    topology in sysfs, does not even import torch (checked with -X importtime), answers in well under 2 s, honours *_VISIBLE_DEVICES; and its
    watchdog kills the child's whole process group on --launch-timeout (exit 124) — exercised with a stand-in child that sleeps."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, "-X", "importtime", os.path.join(root, "bench.py"), "--gpus", "64"], capture_output=True, text=True, env=env, timeout=60)
    assert r.returncode == 3 and time.perf_counter() - t0 < 2.0
    assert "torch" not in r.stderr.replace("torch.distributed.run", "")            # no torch import in the parent (importtime lists every import)
    sys.path.insert(0, root)
    import importlib
    bench = importlib.import_module("bench")
    os.environ["ROCR_VISIBLE_DEVICES"] = ""
    try:
        assert bench.visible_gpus() == 0
    finally:
        del os.environ["ROCR_VISIBLE_DEVICES"]
    # watchdog: make the "launcher" a sleeping python by pointing sys.executable's module at a stub through PYTHONPATH
    stub = tmp_path / "torch" / "distributed"
    stub.mkdir(parents=True)
    (tmp_path / "torch" / "__init__.py").write_text("")
    (stub / "__init__.py").write_text("")
    (stub / "run.py").write_text("import time, os\nopen(os.environ['XP_STUB_PID'], 'w').write(str(os.getpid()))\ntime.sleep(600)\n")
    code = ("import sys, types; sys.argv = ['bench.py', '--gpus', '2', '--launch-timeout', '2']; sys.path.insert(0, %r); import bench; "
            "bench.visible_gpus = lambda: 2; bench.main()" % root)
    env2 = dict(env, PYTHONPATH=str(tmp_path), XP_STUB_PID=str(tmp_path / "pid"))
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env2, timeout=60)
    assert r.returncode == 124, (r.returncode, r.stderr[-400:])
    assert time.perf_counter() - t0 < 30 and "killing its process group" in r.stderr
    pid = int((tmp_path / "pid").read_text())
    time.sleep(0.5)
    alive = False
    if os.path.exists(f"/proc/{pid}/status"):
        state = open(f"/proc/{pid}/status").read().split("State:")[1].split()[0]
        alive = state not in ("Z", "X")
    assert not alive, "the child of the timed-out job is still running"


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xpoint_amd import models
    cfg = synth.xpoint_exp1_config(64, 96, vssm={"EMBED_DIM": 32})
    net = models.XPoint(cfg).eval()
    blob = xdist.broadcast_weights(net, (lambda: synth.make_torch_state_dict(cfg)), src=0, device="cpu")
    # every rank ends with the same bytes; rank 1 never built the state dict
    ref = None
    if rank == 0:
        ref = net.pack_weights()
        assert torch.equal(ref, blob)
    else:
        assert len(net.state_dict()) == 0
    digest = torch.tensor([float(blob.double().sum()), float(blob.double().abs().sum()), float(blob.numel())], dtype=torch.float64)
    gathered = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(gathered, digest)
    first, cnt = xdist.shard_pairs(6, world, rank)
    ms = xdist.timed_broadcast(blob, src=0, repeats=2)
    assert ms > 0.0
    if rank == 0:
        assert torch.equal(ref, blob)                                   # re-broadcasts leave the bytes alone
    hdr = xdist.gather_headers(first, cnt, 100 + rank, 10 + rank)
    assert hdr == [(0, 3, 100, 10), (3, 3, 101, 11)]
    assert xdist.gather_strings(f"{rank * 16}-{rank * 16 + 15}") == ["0-15", "16-31"]
    q.put((rank, [g.tolist() for g in gathered], first, cnt))
    dist.destroy_process_group()


def test_weight_broadcast_and_sharding_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out.sort()
    for rank, gathered, first, cnt in out:
        assert gathered[0] == gathered[1] and gathered[0][2] > 1e5          # identical blobs on both ranks
    assert [(o[2], o[3]) for o in out] == [(0, 3), (3, 3)]                   # disjoint shards covering the batch


def _fake_sysfs(root, gpus, cpu_nodes=1):
    """A KFD + PCI sysfs tree: `gpus` = list of (bus, cpulist or None); KFD node 0.. are CPU nodes (simd_count 0), GPUs follow."""
    k = 0
    for _ in range(cpu_nodes):
        d = root / "class/kfd/kfd/topology/nodes" / str(k); d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\n")
        k += 1
    for bus, cpus in gpus:
        d = root / "class/kfd/kfd/topology/nodes" / str(k); d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count 0\nsimd_count 1024\ndomain 0\nlocation_id {bus << 8}\nname ignored_text\n")
        if cpus is not None:
            p = root / "bus/pci/devices" / f"0000:{bus:02x}:00.0"; p.mkdir(parents=True)
            (p / "local_cpulist").write_text(cpus + "\n")
        k += 1


def test_rank_cpu_affinity_plan(tmp_path, monkeypatch):
    """every rank is pinned to the CPUs of its GPU's NUMA node before torch is imported (xpoint_amd/affinity.py: KFD topology ->
    PCI local_cpulist; GPUs on one node split its CPUs; intersected with the allowed set; even split when sysfs is silent) — the mapping, on a fake
    sysfs tree of a two-socket 8-GPU node, plus the real call on this host (applies and restores)."""
    import importlib, sys
    from xpoint_amd import affinity as af
    assert "torch" not in getattr(af, "__dict__", {}) and af.parse_cpulist("0-3,8,10-11") == [0, 1, 2, 3, 8, 10, 11]
    assert af.format_cpulist([0, 1, 2, 3, 8, 10, 11]) == "0-3,8,10-11" and af.format_cpulist([]) == ""
    _fake_sysfs(tmp_path, [(0x11 + 0x10 * i, "0-31,64-95" if i < 4 else "32-63,96-127") for i in range(8)], cpu_nodes=2)
    gpus = af.gpu_local_cpus(str(tmp_path))
    assert len(gpus) == 8 and gpus[0] == af.parse_cpulist("0-31,64-95") and gpus[7] == af.parse_cpulist("32-63,96-127")
    allowed = set(range(128))
    sets = af.plan(8, allowed, gpus)
    assert all(len(s) == 16 for s in sets)                                           # 64 CPUs of a socket / 4 GPUs
    assert set().union(*sets[:4]) == set(gpus[0]) and set().union(*sets[4:]) == set(gpus[4])
    assert sum(len(s) for s in sets) == len(set().union(*sets)) == 128               # disjoint, nothing unused
    assert sets[0] == list(range(0, 16)) and sets[4] == list(range(32, 48))
    # a container cpuset that only allows socket 0: ranks on socket-1 GPUs fall back to an even share of what is left (never empty)
    sets = af.plan(8, set(range(0, 32)), gpus)
    assert all(s and set(s) <= set(range(32)) for s in sets) and [len(s) for s in sets[:4]] == [8, 8, 8, 8]
    # two ranks on a one-GPU node description / unknown NUMA node: even split of the allowed set
    assert af.plan(2, set(range(8)), [None]) == [[0, 1, 2, 3], [4, 5, 6, 7]]
    assert af.plan(3, {5}, []) == [[5], [5], [5]]                                    # fewer CPUs than ranks: everybody keeps the set
    # *_VISIBLE_DEVICES reorders / narrows the mapping
    assert af.visible_device_indices(8, {"HIP_VISIBLE_DEVICES": "4,5"}) == [4, 5]
    assert af.visible_device_indices(8, {"ROCR_VISIBLE_DEVICES": "2,3,4,5", "HIP_VISIBLE_DEVICES": "1,0"}) == [3, 2]
    assert af.plan(2, allowed, gpus, af.visible_device_indices(8, {"HIP_VISIBLE_DEVICES": "4,5"}))[0] == list(range(32, 64))
    # the real call: applies to this process and can be undone
    before = sorted(os.sched_getaffinity(0))
    info = af.pin_rank(0, 2, sysfs_root=str(tmp_path / "nothing_here"))
    assert info["applied"] and info["source"].startswith("even split") and info["count"] == (len(before) // 2 or len(before))
    assert sorted(os.sched_getaffinity(0)) == af.parse_cpulist(info["cpus"])
    af.restore(info["previous"])
    assert sorted(os.sched_getaffinity(0)) == before
    # the module pulls in no GPU runtime and no torch
    code = "import sys; import xpoint_amd.affinity as a; a.pin_rank(0, 1, apply=False); print(sorted(m for m in sys.modules if m.split('.')[0] in ('torch', 'numpy')))"
    import subprocess
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and r.stdout.strip() == "[]", (r.stdout, r.stderr)
