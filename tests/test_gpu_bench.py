"""bench.py as the driver runs it: `python bench.py --gpus N` (unwrapped, self-launching) — RCCL initialised at every
world size, one weight broadcast, an all-gather of the result headers, ONE JSON line from rank 0."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(n, extra=()):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    # the CPU baseline stays ON (bounded to one pair here): every line, at every world size, must carry it
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--cpu-pairs", "1",
           "--no-other-backend", *extra]
    return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)


def test_bench_launcher_rccl_world():
    world = min(2, torch.cuda.device_count())
    assert world >= 1
    r = _run(world)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["rccl_ranks"] == world, out
    assert "rccl_error" not in out
    assert out["weight_bcast_ms"] > 0 and out["weight_blob_mb"] > 80
    hdr = out["rank_headers"]
    assert [h["rank"] for h in hdr] == list(range(world))
    assert [h["first_pair"] for h in hdr] == [8 * i for i in range(world)] and all(h["pairs"] == 8 for h in hdr)
    assert all(h["keypoints"] > 8 * 2 * 3000 and h["matches"] > 8 * 500 for h in hdr), hdr
    assert out["value"] > 0 and out["scaling"] == "weak" and out["roofline"]["frac"] > 0
    assert "pcie_inclusive_pairs_per_s" in out
    # round 3: the K-step region is timed five times and the median reported; every rank's own rate; first vs steady-state weight broadcast
    tr = out["timed_regions"]
    assert tr["count"] == 5 and tr["reported"] == "median" and len(tr["pairs_per_s"]) == 5
    assert sorted(tr["pairs_per_s"])[2] == pytest.approx(out["value"], rel=1e-3)
    pr = out["per_rank_pairs_per_s"]
    assert 0 < pr["min"] <= pr["max"] and pr["max"] * world >= out["value"] * 0.99
    assert out["weight_bcast_first_ms"] > 0
    assert out["roofline"].get("traffic") or out["roofline"].get("traffic_error")        # a PMC lookup failure is reported, never swallowed
    # round 4: matcher nomination statistics, per-rank CPU masks, precision class label
    assert out["match_candidates_per_row"]["mean"] >= 1.0 and out["match_candidates_per_row"]["max"] >= 1 and out["match_overflow_rows"] >= 0
    aff = out["rank_cpu_affinity"]
    assert len(aff["cpus_per_rank"]) == world and aff["rank0"]["count"] >= 1 and aff["rank0"]["applied"] in (True, False)
    assert out["precision_class"] == "f32" and "NOT the headline" not in out["metric"]
    # round 6: the N-rank line is complete — the keys of the 1-GPU line at every world size (cpu_baseline runs on rank 0 after the process group is torn down)
    cb = out["cpu_baseline"]
    assert cb["value"] > 0 and cb["kind"] == "port" and cb["cores"] >= 1 and "sample" in cb, cb
    assert set(("roofline", "cpu_baseline", "rccl_ranks", "per_rank_pairs_per_s", "rank_cpu_affinity", "timed_regions", "rank_headers")) <= set(out)
    # and the counter numbers quoted in it are tied to what ran: source hash of this tree, stale flag (a missing / foreign summary is an error key)
    roof = out["roofline"]
    assert len(roof["kernel_source_hash"]) == 16
    assert "traffic_stale" in roof or "traffic_error" in roof, roof          # (a summary written before round 6 has no hash: quoted, flagged stale)
    if roof.get("traffic_source_hash") == roof["kernel_source_hash"]:
        assert roof["traffic_stale"] is False


def test_bench_two_rank_rehearsal_on_one_gpu():
    """the first real N-GPU line must not be the thing that fails.  No multi-GPU node exists for this run, so the N-rank code path is
    REHEARSED on the one GPU: two ranks (torch.distributed.run, started by bench.py itself) share GPU 0, their collectives run over gloo on host tensors
    (RCCL refuses two ranks on one device) — pair sharding, the weight broadcast to a rank that never built the state dict, the gathers of the timed
    regions / CPU masks / result headers, max-over-ranks timing, ONE JSON line with every key of the 1-GPU line.  Rates of this run mean nothing."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["XP_BENCH_REHEARSE_ON_ONE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--cpu-pairs", "1", "--no-other-backend", "--no-h2d"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and "rehearsal" in out and out["scaling"] == "weak"
    hdr = out["rank_headers"]
    assert [h["rank"] for h in hdr] == [0, 1] and [h["first_pair"] for h in hdr] == [0, 8] and all(h["pairs"] == 8 for h in hdr)
    assert all(h["keypoints"] > 8 * 2 * 3000 and h["matches"] > 8 * 500 for h in hdr), hdr
    assert (hdr[0]["keypoints"], hdr[0]["matches"]) != (hdr[1]["keypoints"], hdr[1]["matches"])      # rank 1 ran ITS pairs (8..15), with weights it only ever received by broadcast
    assert len(out["rank_cpu_affinity"]["cpus_per_rank"]) == 2
    pr = out["per_rank_pairs_per_s"]
    assert 0 < pr["min"] <= pr["max"] and out["value"] <= 2 * pr["min"] * 1.001      # whole-job value = all pairs / the SLOWEST rank's time
    assert out["timed_regions"]["count"] == 5
    assert out["cpu_baseline"]["value"] > 0 and out["roofline"]["frac"] > 0 and out["weight_bcast_ms"] > 0
    assert set(("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline")) <= set(out)


def test_bench_hygiene_keys_and_fast_class_label():
    """The default line carries the rotating-input and sustained-region rates (with the shader clock) and the reduced-precision classes incl. amp16f; a
    `--precision-class amp16f` run labels itself as not the headline."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "2", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["rotating_inputs_pairs_per_s"] > 0.8 * out["value"]
    assert out["sustained_pairs_per_s"] > 0.8 * out["value"] and out["sustained_region"]["seconds"] >= 10.0
    clk = out["sustained_region"]["shader_clock"]
    assert clk["mid_region_mhz"] is None or 500 <= clk["mid_region_mhz"] <= 3000, clk
    cls = out["reduced_precision_classes"]["pairs_per_s"]
    assert cls["amp16f"] > out["value"] and cls["amp16f"] > cls["amp16"], cls          # the half-storage class is the FAST one
    assert out["pcie_inclusive_pairs_per_s"] > 0.6 * out["value"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--no-h2d", "--precision-class", "amp16f"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["precision_class"] == "amp16f" and "NOT the headline" in out["metric"] and out["dtype"].startswith("f16 storage")
    # the class's dominant kernel is whichever tag holds most of the step: the one-product GEMM (matrix-pipe roofline, 2.5 PF/s) or, since the fused MLPs and
    # the GEMM tiles got faster, an SS2D pass (HBM roofline) — the two are within a few per cent of each other
    roof = out["roofline"]
    assert ("_f16" in roof["kernel"] and roof["peak"] == 2500.0 and roof["bound"] == "mfma") or (roof["kernel"].startswith("ss2d_") and roof["bound"] == "hbm"), roof
    assert 0 < roof["frac"] < 1


def test_bench_launcher_refuses_more_gpus_than_visible():
    n = torch.cuda.device_count() + 1
    r = _run(n)
    assert r.returncode == 3 and f"needs {n} GPUs" in r.stderr, (r.returncode, r.stderr[-500:])
