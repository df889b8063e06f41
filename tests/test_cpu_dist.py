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
