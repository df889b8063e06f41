#!/usr/bin/env python3
"""bench.py — XPoint hot path throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 works both ways: under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment), and as a plain `python bench.py --gpus N`, in which
case this process only counts the devices and starts that launcher as a CHILD process (it never initialises the GPU
itself and never exec()s), forwards its output and exits with its code; fewer than N visible GPUs is exit code 3.

A "step" is one pass of the whole hot path (encode both images with the VMamba encoder, detector +
descriptor heads, box NMS, keypoint extraction, descriptor sampling, mutual-NN matching) over one batch
of 8 synthetic 480x640 optical/thermal pairs per GPU (BASELINE.json configs[1]); inputs are resident in
HBM when the timed region starts.  Pairs are independent units: every rank runs the full path on its
own shard (weak scaling), the only collective is the one-time RCCL broadcast of the packed weights.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W, PAIRS = 480, 640, 8          # config c2 (the headline); c4 / c5 override them in main()
CONFIGS = {
    # BASELINE.json configs[1]: the configuration the metric is quoted on
    "c2": dict(H=480, W=640, pairs=8, cap=8192, topk=0, label="C2: XPoint VMamba encoder, 480x640 optical-thermal, batch=8 pairs/GPU"),
    # configs[3]: 1024x1024, keep_top_k 4096 -> a dense 4k x 4k x 256 descriptor match per pair
    "c4": dict(H=1024, W=1024, pairs=4, cap=16384, topk=4096, nms_sweeps=12, label="C4: XPoint 1024x1024 VIS-SAR pairs, keep_top_k 4096 (4k x 4k x 256 match)"),
    # configs[4]: streaming (images from pinned host memory every step, result lists back to the host), hipGraph-replayed step,
    # homography-regression head.  The reference's RegNet head is only defined for 256x256 inputs (its FC layer is sized for a
    # 32x32 encoder map: RegNet.py:38-52), so the head runs on the 256x256 top-left crop of every pair, in the step.
    "c5": dict(H=480, W=640, pairs=8, cap=8192, topk=0, label="C5: XPoint + RegNet head (on 256x256 crops), 480x640 optical-NIR, streaming 8-pair steps "
                                                                  "from pinned host memory, hipGraph-replayed step"),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: FP32 matrix peak
MFMA_BF16_PEAK_TFLOPS = 2500.0 # MI355X_MICROARCH.md: dense bf16 matrix peak (no sparsity)
X3_PRODUCTS = 6                # bf16 MFMA partial products per f32-accurate multiply in the split-bf16 GEMM (csrc/gemm_x3_core.h)
H2_PRODUCTS = 3                # fp16 MFMA partial products per f32-grade multiply in the split-fp16 GEMM (csrc/gemm_h2_core.h)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)     # 100 steps = 0.7 s timed: run-to-run spread of a 20-step region was +-5 %
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2", help="BASELINE.json configuration (c2 = the headline; c4, c5: see CONFIGS)")
    ap.add_argument("--pairs", type=int, default=0, help="pairs per GPU per step (default: the configuration's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--regions", type=int, default=5, help="the K-step timed region is run this many times inside one invocation; `value` is the MEDIAN "
                                                            "region (a single 20-step region is 0.1 s: +-5 % run to run); every region is listed in the line")
    ap.add_argument("--launch-timeout", type=float, default=1800.0, help="unwrapped --gpus N > 1: seconds the parent waits for the child job before "
                                                                          "killing its process group (exit code 124)")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured hipGraph (roofline events are then taken in a separate eager pass)")
    ap.add_argument("--no-h2d", action="store_true", help="skip the extra PCIe-inclusive pass (pcie_inclusive_pairs_per_s: images uploaded from pinned host memory and "
                                                          "keypoints / match lists downloaded every step; reported beside the headline, never as `value`)")
    ap.add_argument("--cpu-pairs", type=int, default=24, help="pairs in the bounded CPU-baseline sample")
    ap.add_argument("--split-encoder", type=int, default=0, help="encoder as S image groups on S streams instead of the default schedule (round 2 until the alternating "
                    "schedule: S = 2); needs overlap")
    ap.add_argument("--no-alternate", action="store_true", help="default schedule off: by default the whole-batch encoders of consecutive steps alternate between two streams "
                    "(two forwards in flight next to the detection / matching of the step before); with this flag and no --split-encoder: one encoder stream")
    ap.add_argument("--register", action="store_true", help="also run the registration step (robust homography per pair) inside every step; not part of the headline metric's definition")
    ap.add_argument("--no-overlap", action="store_true", help="one stream: no overlap of step i's detection / matching kernels with step i+1's encoder")
    ap.add_argument("--no-other-backend", action="store_true", help="skip the extra timing passes (other dense-layer back end, single-stream rate): keeps profiler output to the headline configuration")
    ap.add_argument("--precision-class", choices=["f32", "amp16f", "amp16"], default="f32",
                    help="f32 (default) = the headline class (the reference's CPU arithmetic, 1e-4 bar).  amp16f / amp16 run the WHOLE invocation in the reference's "
                         "mixed_precision deployment class (half storage / f32 containers) for profiling: the line then says so in `metric`, `dtype` and `precision_class` "
                         "and must not be read as the headline")
    ap.add_argument("--gemm", choices=["h2", "x3", "f32"], default=os.environ.get("XP_GEMM_MODE", "h2"),
                    help="dense-layer back end: h2 = f32-grade split-fp16 on the f16 matrix pipe (default), x3 = f32-grade split-bf16, f32 = exact-f32 MFMA "
                         "(the reduced-precision classes x2 / bf16 are timed as extra, labelled passes only: never the headline)")
    return ap.parse_args()


def cpu_baseline(n_pairs):
    """The oracle (CPU restatement: torch fp32 ops + C scan/NMS/matcher with OpenMP) timed on this host on a
    bounded sample of the same workload.  kind 'port': the reference's own Python path cannot travel here."""
    import torch
    from oracle import xpoint_oracle as xo
    from xpoint_amd import synth
    cfg = synth.xpoint_exp1_config(H, W)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg).items()}
    # threads actually used: torch intra-op + the OpenMP C kernels share one pool; more than ~16-32 threads makes this
    # path slower (many small ops), so the baseline uses min(host cores, XP_CPU_THREADS or 16)
    cores = min(os.cpu_count() or 1, int(os.environ.get("XP_CPU_THREADS", "16")))
    torch.set_num_threads(cores)
    data = synth.to_torch(synth.make_pair_batch(0, 1, H, W))
    with torch.no_grad():
        xo.predict_align_image_pair(data, sd)           # warm-up (also builds liboracle.so)
        t0 = time.perf_counter()
        for i in range(n_pairs):
            data = synth.to_torch(synth.make_pair_batch(i, 1, H, W))
            xo.predict_align_image_pair(data, sd)
        dt = time.perf_counter() - t0
    return {"value": n_pairs / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"{n_pairs} synthetic 480x640 pairs, batch 1, full path (oracle/xpoint_oracle.py predict_align_image_pair), "
                      f"torch {torch.get_num_threads()} threads + OpenMP C scan; reference's own CPU path measured 0.19 pairs/s "
                      f"on 8 cores (BASELINE.md)"}


def pmc_kernel_for_tag(tag, names):
    """HIP-event tag of the library (csrc/*: XpProfScope) -> kernel name as rocprofv3 prints it (tools/pmc_summary.py `short`).  Raises KeyError
    with the candidates when nothing (or more than one kernel) matches, so a renamed template shows up in the bench line instead of a null."""
    import re
    m16 = re.match(r"(ln_mlp_fused|mlp_fused|ln_proj)_f16_c(\d+)$", tag)
    if m16:
        # fast mixed-precision class (csrc/mlp_f16.hip): (ln_)mlp_fused_f16_c<C> <-> mlp_f16_kernel<C>, ln_proj_f16_c<C> <-> ln_proj_f16_kernel<C>
        pat = re.compile(rf"{'ln_proj_f16_kernel' if m16.group(1) == 'ln_proj' else 'mlp_f16_kernel'}<{m16.group(2)}>$")
    elif "mlp_fused" in tag or tag.startswith("ln_proj"):
        # (proj_)mlp_fused_{h2|x3}_c<C>, ln_proj_{h2|x3}_c<C>  <->  mlp_fused_kernel<C, NW, MODE, NP, H2>; MODE 0 = MLP, 1 = out_proj + MLP, 2 = LN + projection
        m = re.match(r"(proj_mlp_fused|mlp_fused|ln_proj)_(h2|x3)_c(\d+)$", tag)
        if not m:
            raise KeyError(f"unrecognised fused-kernel tag {tag!r}")
        mode = {"mlp_fused": 0, "proj_mlp_fused": 1, "ln_proj": 2}[m.group(1)]
        # (round 5: a trailing PP parameter — the ping-pong schedule, off by default — follows H2)
        pat = re.compile(rf"mlp_fused_kernel<{m.group(3)}, \d+, {mode}, \d+, {'true' if m.group(2) == 'h2' else 'false'}(, (true|false))?>$")
    elif tag.startswith("gemm_ring"):
        # ring dense engine (csrc/ring_core.h): ring_gemm_kernel<GM, GN, TM, TN, PLANES, S>; tag gemm_ring_{h2s|f16}_<BM>x<BN>
        m = re.match(r"gemm_ring_(h2s|f16)_(\d+)x(\d+)$", tag)
        if not m:
            raise KeyError(f"unrecognised ring tag {tag!r}")
        cfg = {("256", "256"): "2, 2, 2, 4", ("256", "128"): "2, 2, 2, 2", ("128", "128"): "1, 4, 2, 1"}[(m.group(2), m.group(3))]
        pat = re.compile(rf"ring_gemm_kernel<{cfg}, {2 if m.group(1) == 'h2s' else 1}, \d+>$")
    elif tag.startswith(("gemm_h2p", "gemm_h2w")):
        # ping-pong / wave-specialised schedules of the split-fp16 GEMM: one (non-template) kernel each
        pat = re.compile(rf"{tag.split('_mfma_')[0]}_kernel$")
    elif tag.startswith(("gemm", "conv3x3")):
        m = re.match(r"(gemm|conv3x3)(_h2r|_h2|_x3|_f16|)_mfma_(\d+)x(\d+)(_k32)?$", tag)
        if not m or (m.group(5) and m.group(2) != "_f16"):
            raise KeyError(f"unrecognised GEMM tag {tag!r}")
        cfgs = {("128", "128"): "2, 2, 2, 2", ("128", "96"): "4, 1, 1, 3", ("64", "128"): "2, 2, 1, 2", ("128", "64"): "4, 1, 1, 2", ("128", "32"): "4, 1, 1, 1",
                ("128", "192"): "2, 2, 2, 3", ("256", "128"): "4, 2, 2, 2"}
        conv = 1 if m.group(1) == "conv3x3" else 0
        eng = m.group(2)
        if eng == "_h2r":         # row-stationary split-fp16 kernel: gemm_h2r_kernel<TN, MODE>
            pat = re.compile(rf"gemm_h2r_kernel<{int(m.group(4)) // 32}, {conv}>$")
        else:
            base = {"_h2": "gemm_h2_kernel", "_x3": "gemm_x3_kernel", "": "gemm_kernel", "_f16": "gemm_f16_kernel"}[eng]
            bk = (", 32" if m.group(5) else ", 64") if eng == "_f16" else r"(, \d+)?"        # gemm_f16_kernel<WM, WN, TM, TN, MODE, BK>
            pat = re.compile(rf"{base}<{cfgs[(m.group(3), m.group(4))]}, {conv}{bk}>$")
    else:
        pat = re.compile(re.escape(tag) + r"(_kernel)?(<.*>)?$")
    hits = [n for n in names if pat.search(n)]
    if len(hits) > 1 and all(h.startswith("mlp_fused_kernel<") or "mlp_fused_kernel<" in h for h in hits):
        # the C = 192 split-fp16 instances run their last, partly filled round as 4-wave workgroups in a second launch (csrc/mlp_fused.hip): same tag,
        # two NW values — the 8-wave main launch is the one the tag's time is about
        hits = [max(hits, key=lambda h: int(re.search(r"mlp_fused_kernel<\d+, (\d+),", h).group(1)))]
    if len(hits) != 1:
        raise KeyError(f"tag {tag!r} matches {len(hits)} kernels of the PMC file (pattern {pat.pattern!r}; hits {hits[:4]})")
    return hits[0]


def pmc_family_for_tag(tag, names):
    """Tags that cover SEVERAL template instances in one step (the chunked SS2D passes run once per stage: dt_rank 6 / 12 / 24): all kernels of the family,
    under both spellings rocprofv3 prints (demangled `ss2d_pass3<6, true, ...>`, or the raw mangled name when the demangler gives up on _Float16)."""
    import re
    fam = {"ss2d_pass1": (r"ss2d_pass1<\d+,", r"ss2d_pass1ILi\d+E"), "ss2d_pass2": (r"ss2d_pass2$", r"ss2d_pass2E"),
           "ss2d_pass3_row": (r"ss2d_pass3<\d+, false,", r"ss2d_pass3ILi\d+ELb0E"), "ss2d_pass3_col_ln": (r"ss2d_pass3<\d+, true,", r"ss2d_pass3ILi\d+ELb1E")}.get(tag)
    if fam is None:
        return [pmc_kernel_for_tag(tag, names)]
    pats = [re.compile(f) for f in fam]
    hits = [n for n in names if any(p.search(n) for p in pats)]
    if not hits:
        raise KeyError(f"tag {tag!r} matches no kernel of the PMC file (patterns {fam})")
    return hits


def pmc_files(config, precision_class):
    """The PMC summaries a bench line of (config, precision class) may quote: collected on THAT workload (tools/profile_round.sh), never another one's."""
    sfx = ("" if config == "c2" else "_" + config) + ("" if precision_class == "f32" else "_" + precision_class)
    return os.path.join(ROOT, "profiles", f"pmc_traffic{sfx}.json"), os.path.join(ROOT, "profiles", f"pmc_mfma{sfx}.json")


def pmc_fields(dominant, bound, config, precision_class, current_hash=None, files=None):
    """roofline.traffic / frac_mfma_busy_pmc from the committed rocprofv3 --pmc summaries (counters cannot be collected from inside this process), tied
    to what ran: every summary carries the hash of the kernel sources it was collected on (xpoint_amd.build.source_hash); when it differs from the
    sources of THIS tree the numbers are still quoted but flagged `traffic_stale` / `frac_mfma_busy_pmc_stale` = true.  A missing file, a summary
    of another workload or a tag that matches no kernel is reported (`*_error`), never swallowed."""
    from xpoint_amd.build import source_hash
    cur = current_hash or source_hash()
    tf, mf_path = files or pmc_files(config, precision_class)
    want = config if precision_class == "f32" else precision_class
    out = {"traffic": None, "kernel_source_hash": cur}
    try:
        doc = json.load(open(tf))
        if doc.get("workload", "c2") != want:
            raise KeyError(f"{os.path.basename(tf)} holds workload {doc.get('workload')!r}, this line is {want!r}")
        pmc = doc["kernels"]
        knames = pmc_family_for_tag(dominant, list(pmc))
        nl = sum(pmc[k]["launches"] for k in knames)            # launch-weighted mean over the tag's template instances, like `achieved`
        out["traffic"] = round(sum(pmc[k]["hbm_bytes_per_launch"] * pmc[k]["launches"] for k in knames) / nl)
        out["traffic_kernel"] = knames[0] if len(knames) == 1 else knames
        out["traffic_source"] = f"profiles/{os.path.basename(tf)} (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE; (2F+W)*1024 bytes per launch)"
        out["traffic_source_hash"] = doc.get("source_hash")
        out["traffic_stale"] = doc.get("source_hash") != cur
    except Exception as e:
        out["traffic_error"] = f"{type(e).__name__}: {e}"
        sys.stderr.write(f"bench.py: roofline.traffic unavailable for {dominant}: {e}\n")
    try:
        doc = json.load(open(mf_path))
        if doc.get("workload", "c2") != want:
            raise KeyError(f"{os.path.basename(mf_path)} holds workload {doc.get('workload')!r}, this line is {want!r}")
        mf = doc["kernels"]
        kname = pmc_kernel_for_tag(dominant, list(mf))
        out["frac_mfma_busy_pmc"] = round(mf[kname]["mfma_busy_frac"], 4)
        out["frac_mfma_busy_pmc_source"] = (f"profiles/{os.path.basename(mf_path)} (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), "
                                            "tools/mfma_util.sh)")
        out["frac_mfma_busy_pmc_stale"] = doc.get("source_hash") != cur
    except Exception as e:
        if bound == "mfma":
            out["frac_mfma_busy_pmc_error"] = f"{type(e).__name__}: {e}"
    return out


def dense_engine_ceilings():
    """STATIC reference data (not a measurement of this run): the measured ceilings of the ring dense engine's load path and matrix-instruction stream
    (tools/ring_bench stage-removal builds), read from profiles/ring_ceilings.json with the source hash and box they were measured on.  Emitted only
    when the dominant kernel is a gemm_ring_* launch."""
    try:
        doc = json.load(open(os.path.join(ROOT, "profiles", "ring_ceilings.json")))
        doc["static_reference_data"] = True
        return doc
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def visible_gpus():
    """GPUs this process may use, counted WITHOUT loading the HIP / HSA runtime (the parent of a multi-rank job must stay GPU-free: it is
    never a rank).  KFD topology in sysfs: a node with simd_count > 0 is a GPU; the *_VISIBLE_DEVICES variables the runtime honours narrow it."""
    import glob
    n = 0
    for prop in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            for line in open(prop):
                f = line.split()
                if len(f) == 2 and f[0] == "simd_count" and int(f[1]) > 0:
                    n += 1
        except OSError:
            pass
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([t for t in v.split(",") if t.strip() != ""]))
    # sysfs can mislead in a container: every host GPU is listed although only some /dev/dri render nodes are mapped (count too high), or
    # /sys/class/kfd is not mounted (count 0).  Cross-check against the render nodes; if the two disagree or sysfs is silent, ask the runtime in a
    # SHORT-LIVED CHILD (torch.cuda.device_count() there), so that this process still never loads it.
    import glob as _glob
    render = len(_glob.glob("/dev/dri/renderD*"))
    if n == 0 and not os.path.exists("/dev/kfd"):
        return 0                                   # no compute driver node at all: nothing to ask (and the refusal stays instant)
    if n == 0 or (render and render < n):
        import subprocess
        try:
            r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=300)
            return int(r.stdout.strip().splitlines()[-1])
        except Exception:
            return min(n, render) if render else n
    return n


def self_launch(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the one-process-per-GPU job as a CHILD process group.  The parent
    never imports torch, never opens /dev/kfd, never exec()s; it counts GPUs from sysfs, forwards the child's output and exit code, and is
    the job's watchdog: past --launch-timeout seconds (or on Ctrl-C) the whole process group is killed and the exit code is non-zero.
    torch.distributed.run itself tears the other ranks down when the first one fails and returns that failure."""
    import signal
    import subprocess
    visible = visible_gpus()
    if os.environ.get("XP_BENCH_REHEARSE_ON_ONE_GPU") == "1" and visible >= 1:
        visible = args.gpus              # rehearsal of the N-rank code path on a 1-GPU box (see main): every rank shares GPU 0
    if visible < args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} needs {args.gpus} GPUs, {visible} visible on this node\n")
        raise SystemExit(3)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    child = subprocess.Popen(cmd, env=env, start_new_session=True)          # own process group: one killpg reaches the launcher and every rank

    def kill_group(sig=signal.SIGTERM):
        try:
            os.killpg(child.pid, sig)
        except ProcessLookupError:
            pass
    try:
        rc = child.wait(timeout=args.launch_timeout)
    except subprocess.TimeoutExpired:
        sys.stderr.write(f"bench.py: the {args.gpus}-rank job did not finish within --launch-timeout {args.launch_timeout:.0f} s; killing its process group\n")
        kill_group()
        try:
            child.wait(timeout=20)
        except subprocess.TimeoutExpired:
            kill_group(signal.SIGKILL)
            child.wait()
        raise SystemExit(124)
    except KeyboardInterrupt:
        kill_group()
        child.wait()
        raise SystemExit(130)
    raise SystemExit(rc if rc >= 0 else 128 - rc)


def dt_hint(region_dt):
    return sorted(region_dt)[(len(region_dt) - 1) // 2]


def gpu_clock_mhz(local=0):
    """Current shader clock of THIS rank's GPU, MHz: the visible GPU's KFD node (the nodes of GPUs that are not mapped into the container are unreadable) names
    its DRM render minor -> /sys/class/drm/renderD<minor>/device/pp_dpm_sclk, the level marked '*'; rocm-smi as a fallback; None when neither answers.
    Read before, in the middle of and after the sustained region: the matrix pipe's peak assumes 2.4 GHz, the part throttles under dense MFMA load."""
    import glob
    import re
    minors = []
    for prop in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"), key=lambda q: int(q.split("/")[-2])):
        try:
            kv = dict(line.split()[:2] for line in open(prop) if len(line.split()) >= 2)
        except OSError:
            continue                                   # another tenant's GPU
        if int(kv.get("simd_count", "0")) > 0 and "drm_render_minor" in kv:
            minors.append(int(kv["drm_render_minor"]))
    paths = [f"/sys/class/drm/renderD{minors[local]}/device/pp_dpm_sclk"] if local < len(minors) else []
    for path in paths:
        try:
            for line in open(path):
                if "*" in line:
                    m = re.search(r"(\d+)\s*[Mm][Hh]z", line)
                    if m:
                        return int(m.group(1))
        except OSError:
            pass
    try:
        import subprocess
        r = subprocess.run(["rocm-smi", "-d", str(local), "--showclocks"], capture_output=True, text=True, timeout=15)
        m = re.search(r"sclk[^\n]*?\((\d+)Mhz\)", r.stdout + r.stderr)
        return int(m.group(1)) if m else None
    except Exception:
        return None


def main():
    global H, W
    args = parse()
    conf = CONFIGS[args.config]
    H, W = conf["H"], conf["W"]
    if args.pairs <= 0:
        args.pairs = conf["pairs"]
    if args.config == "c5":
        args.graph = True
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    # every rank pins itself to the CPUs next to its GPU BEFORE torch (and with it the HIP runtime's helper threads) is loaded; never a re-exec
    from xpoint_amd import affinity
    pin = affinity.pin_rank(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))),
                            apply=os.environ.get("XP_BENCH_NO_PIN") is None)
    global torch
    import torch
    # stdout carries exactly ONE line, the JSON: keep the real stdout for it and point fd 1 at stderr for everything else (RCCL prints its
    # version banner to stdout from C code when NCCL_DEBUG asks for it)
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    os.environ.setdefault("NCCL_DEBUG", "WARN")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or run bench.py unwrapped)")
    # XP_BENCH_REHEARSE_ON_ONE_GPU=1: the N ranks of the job share GPU 0 and their collectives run over gloo on host tensors — a REHEARSAL of the N-rank
    # code path (pair sharding, weight broadcast, gathers, the line's per-rank keys) on a box with one GPU; RCCL refuses two ranks on one device.
    # The rates of such a run mean nothing and the line says so (tests/test_gpu_bench.py::test_bench_two_rank_rehearsal_on_one_gpu).
    rehearse = os.environ.get("XP_BENCH_REHEARSE_ON_ONE_GPU") == "1" and world > 1
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    coll_dev = torch.device("cpu") if rehearse else dev          # where the tensors of the (few, untimed) collectives live
    # RCCL is initialised at EVERY world size (world 1 included: a single-rank communicator), so that the weight
    # broadcast and the result-header all-gather below run through the same code on 1 and on 8 GPUs
    import torch.distributed as dist
    rccl = {}
    try:
        if world == 1 and os.environ.get("XP_BENCH_NO_RCCL"):       # A/B knob: the timed region without a live communicator
            raise RuntimeError("XP_BENCH_NO_RCCL set")
        if rehearse:
            dist.init_process_group("gloo")
        elif "MASTER_ADDR" in os.environ and "RANK" in os.environ:
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
    except Exception as e:
        if world > 1:
            raise
        rccl["rccl_error"] = repr(e)      # world 1 only: the measurement goes on, the JSON line says RCCL did not come up

    from xpoint_amd import models, synth
    from xpoint_amd.predict import PairPipeline
    from xpoint_amd import _lib

    from xpoint_amd import dist as xdist
    cfg = synth.xpoint_exp1_config(H, W)
    net = models.XPoint(cfg).eval()
    net.gemm_mode = args.gemm if args.precision_class == "f32" else args.precision_class
    # shared "pretrained" weights: rank 0 packs, ONE RCCL broadcast over xGMI to the other ranks (no data-path collective)
    blob = xdist.broadcast_weights(net, (lambda: synth.make_torch_state_dict(cfg)), src=0, device=coll_dev)
    if rehearse:
        net.set_weight_blob(blob.to(dev))
        rccl["rehearsal"] = f"{world} ranks share ONE GPU, collectives over gloo on host tensors: a check of the N-rank code path only — the rates of this line mean nothing"
    if dist.is_initialized():
        rccl["rccl_ranks"] = dist.get_world_size()
        rccl["weight_blob_mb"] = round(blob.numel() * 4 / 1e6, 1)
        if xdist.last_first_broadcast_ms is not None:
            rccl["weight_bcast_first_ms"] = round(xdist.last_first_broadcast_ms, 3)          # the job's first collective: includes RCCL's lazy set-up
        rccl["weight_bcast_ms"] = round(xdist.timed_broadcast(blob, src=0, repeats=5), 3)   # steady-state re-broadcasts of the same blob

    B = args.pairs
    first, _ = xdist.shard_pairs(world * B, world, rank)   # every rank owns its own contiguous block of pairs
    data = synth.to_torch(synth.make_pair_batch(first, B, H, W), dev)
    opt, thr = data["optical"]["image"], data["thermal"]["image"]
    mo, mt = data["optical"]["valid_mask"], data["thermal"]["valid_mask"]
    overlap = not args.no_overlap
    CAP = conf["cap"]
    pred = dict(topk=conf["topk"])
    sweeps = conf.get("nms_sweeps", 8)       # sweeps past the fixed point exit at once; 1024x1024 needs more than 480x640 (longer suppression chains)
    pipe = PairPipeline(net, B, H, W, cap=CAP, cfg_prediction=pred, nms_sweeps=sweeps, overlap=overlap, split_encoder=args.split_encoder,
                        estimate_homography=args.register, alternate_encoders=(int(os.environ.get("XP_BENCH_DEPTH", "3")) if not args.no_alternate and args.split_encoder in (0, 1) else 0))
    # single-stream twin for the per-kernel measurements: with several streams in flight a launch's HIP-event duration
    # includes the time it shares the GPU with other kernels, which says nothing about the kernel itself
    pipe1 = PairPipeline(net, B, H, W, cap=CAP, cfg_prediction=pred, nms_sweeps=sweeps, estimate_homography=args.register) if overlap else pipe
    # config c5: the homography-regression head on the 256x256 crops (a second, small forward through the same encoder weights)
    net_hm = None
    if args.config == "c5":
        cfg_hm = synth.xpoint_exp1_config(256, 256, hm_head=True)
        net_hm = models.XPoint(cfg_hm).eval()
        net_hm.gemm_mode = args.gemm
        net_hm.load_state_dict(synth.make_torch_state_dict(cfg_hm), strict=True)
        net_hm.to(dev)
        pin_o, pin_t = opt.cpu().pin_memory(), thr.cpu().pin_memory()

    def sync_all():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
            torch.cuda.synchronize()

    import ctypes
    lib = _lib.load()

    def prof_table():
        rows = []
        name = ctypes.create_string_buffer(64)
        ms = ctypes.c_double(); cnt = ctypes.c_int(); fl = ctypes.c_double(); by = ctypes.c_double()
        for i in range(lib.xp_prof_count()):
            lib.xp_prof_get(i, name, 64, ctypes.byref(ms), ctypes.byref(cnt), ctypes.byref(fl), ctypes.byref(by))
            rows.append(dict(tag=name.value.decode(), ms=ms.value, launches=cnt.value, flops=fl.value, bytes=by.value))
        return rows

    with torch.no_grad():
        for _ in range(max(args.warmup, 1)):
            pipe.run(opt, thr, mo, mt)
        torch.cuda.synchronize()
        pipe.verify()
        # one untimed single-stream pass with every kernel bracketed by HIP events: per-kernel breakdown, picks the dominant kernel
        pipe1.run(opt, thr, mo, mt); torch.cuda.synchronize()
        lib.xp_prof_reset(); lib.xp_prof_filter(None); lib.xp_prof_enable(1)
        pipe1.run(opt, thr, mo, mt)
        torch.cuda.synchronize()
        lib.xp_prof_enable(0)
        breakdown = sorted(prof_table(), key=lambda r: -r["ms"])
        dominant = breakdown[0]["tag"]
        if args.graph:
            # HIP events cannot be recorded inside a replayed graph: the dominant kernel is timed in eager single-stream passes first
            lib.xp_prof_reset(); lib.xp_prof_filter(dominant.encode()); lib.xp_prof_enable(1)
            for _ in range(3):
                pipe1.run(opt, thr, mo, mt)
            torch.cuda.synchronize(); lib.xp_prof_enable(0)
            if args.config == "c5":
                # streaming step (xpoint_amd/streaming.py; tests/test_gpu_configs.py checks the same object against the reference fixtures): images
                # from pinned host memory straight into the batch buffers, graphs replayed, RegNet head on the 256x256 crops from its own graph,
                # result lists + hm downloaded behind the step; the host consumes step i-1's results while step i runs
                from xpoint_amd.streaming import StreamingRegistrationStep
                sstep = StreamingRegistrationStep(pipe, net_hm, opt, thr, mo, mt)
                pending = []

                def step():
                    bufs, ev, hm_host, hm_ev = sstep(pin_o, pin_t, mo, mt)
                    pending.append((ev, hm_ev))
                    if len(pending) >= pipe.depth:                        # the oldest step in flight is on the host; depth - 1 newer ones keep the GPU busy
                        pev, phm = pending.pop(0)
                        pev.synchronize(); phm.synchronize()
            else:
                replay = pipe.capture(opt, thr, mo, mt)
                step = lambda: replay(opt, thr, mo, mt)
            for _ in range(2):
                step()
            torch.cuda.synchronize()
        elif overlap:
            step = lambda: pipe.run(opt, thr, mo, mt)
            # the dominant kernel's launches are timed in single-stream passes right here (same data, same kernels);
            # the timed region below runs without any events
            lib.xp_prof_reset(); lib.xp_prof_filter(dominant.encode()); lib.xp_prof_enable(1)
            for _ in range(3):
                pipe1.run(opt, thr, mo, mt)
            torch.cuda.synchronize(); lib.xp_prof_enable(0)
            pipe.run(opt, thr, mo, mt)
        else:
            step = lambda: pipe.run(opt, thr, mo, mt)
            # timed region: only the dominant kernel's launches carry events (on their launch stream)
            lib.xp_prof_reset(); lib.xp_prof_filter(dominant.encode()); lib.xp_prof_enable(1)
        # the timed region: K steps bracketed by barrier + synchronize on both sides, run `--regions` times back to back; the line reports the
        # MEDIAN region (max over ranks per region) and lists them all.  steps = K stays what one region times.
        region_dt = []
        for _ in range(max(1, args.regions)):
            sync_all()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            sync_all()
            region_dt.append(time.perf_counter() - t0)
        lib.xp_prof_enable(0)
        dom = [r for r in prof_table() if r["tag"] == dominant][0]
        pipe.verify()
        if args.config == "c5":
            sstep.verify()                  # the head's own status word (its forward runs from its own graph): read once, after the timed region
        want_hygiene = args.config == "c2" and world == 1 and not args.no_other_backend and not args.graph      # (before the override below: a precision-class
        if args.config != "c2" or world > 1 or args.precision_class != "f32":                                     #  run keeps the rotating-input / sustained regions)
            args.no_other_backend = True        # the extra passes (other back ends, precision classes, PCIe-inclusive) are single-GPU records: an N-rank run stays short
        if world > 1:
            args.no_h2d = True
        # the same K steps on ONE stream (no cross-step overlap), for the record
        single_rate = None
        if overlap and not args.no_other_backend:
            for _ in range(2):
                pipe1.run(opt, thr, mo, mt)
            sync_all()
            t3 = time.perf_counter()
            for _ in range(args.steps):
                pipe1.run(opt, thr, mo, mt)
            sync_all()
            single_rate = world * B * args.steps / (time.perf_counter() - t3)
            pipe1.verify()
        # the same step on the other dense-layer back end, for the record (never the headline value)
        other = "x3" if args.gemm == "h2" else "h2"
        other_rate = None
        class_rates = {}
        f32_rate = None
        if not args.graph and not args.no_other_backend:
            net.gemm_mode = other
            for _ in range(2):
                pipe.run(opt, thr, mo, mt)
            sync_all()
            t2 = time.perf_counter()
            for _ in range(args.steps):
                pipe.run(opt, thr, mo, mt)
            sync_all()
            other_rate = world * B * args.steps / (time.perf_counter() - t2)
            pipe.verify()
            net.gemm_mode = "f32"
            for _ in range(2):
                pipe.run(opt, thr, mo, mt)
            sync_all()
            t2 = time.perf_counter()
            for _ in range(args.steps):
                pipe.run(opt, thr, mo, mt)
            sync_all()
            f32_rate = world * B * args.steps / (time.perf_counter() - t2)
            # reduced-precision classes of the same kernels: NOT within the 1e-4 bar, reported beside the headline
            class_rates = {}
            for cls in ("x2", "bf16", "amp16", "amp16f"):
                net.gemm_mode = cls
                for _ in range(2):
                    pipe.run(opt, thr, mo, mt)
                sync_all()
                t3 = time.perf_counter()
                for _ in range(args.steps):
                    pipe.run(opt, thr, mo, mt)
                sync_all()
                class_rates[cls] = world * B * args.steps / (time.perf_counter() - t3)
            net.gemm_mode = args.gemm
            pipe.run(opt, thr, mo, mt)          # leave the buffers holding the headline back end's results
            torch.cuda.synchronize()
    hygiene = {}
    if want_hygiene:
        with torch.no_grad():
            # (a) rotating inputs: the 64 pairs of BASELINE config C3 resident in HBM, a different batch of 8 every step (the headline region replays one batch;
            #     other pairs carry other keypoint counts and NMS chains)
            batches = []
            for k in range(8):
                d = synth.to_torch(synth.make_pair_batch(8 * k, B, H, W), dev)
                batches.append((d["optical"]["image"], d["thermal"]["image"], d["optical"]["valid_mask"], d["thermal"]["valid_mask"]))
            for k in range(8):
                pipe.run(*batches[k])
            sync_all()
            t4 = time.perf_counter()
            for i in range(args.steps):
                pipe.run(*batches[i % 8])
            sync_all()
            hygiene["rotating_inputs_pairs_per_s"] = round(B * args.steps / (time.perf_counter() - t4), 2)
            pipe.verify()
            hygiene["rotating_inputs_note"] = "the same K steps cycling through pairs 0..63 (8 resident batches of 8), a different batch every step"
            # (b) sustained: one region of >= 10 s with the shader clock read before, in the middle and after (DVFS under sustained MFMA load)
            import threading
            n_sus = max(args.steps, int(10.5 / (dt_hint(region_dt) / args.steps)))
            clocks = {"before_mhz": gpu_clock_mhz(local)}
            timer = threading.Timer(5.0, lambda: clocks.__setitem__("mid_region_mhz", gpu_clock_mhz(local)))
            sync_all()
            timer.start()
            t5 = time.perf_counter()
            for _ in range(n_sus):
                pipe.run(opt, thr, mo, mt)
            sync_all()
            t_sus = time.perf_counter() - t5
            clocks["after_mhz"] = gpu_clock_mhz(local)
            timer.cancel()
            pipe.verify()
            hygiene["sustained_pairs_per_s"] = round(B * n_sus / t_sus, 2)
            hygiene["sustained_region"] = {"steps": n_sus, "seconds": round(t_sus, 2), "shader_clock": clocks,
                                           "note": "one timed region of >= 10 s on the headline schedule (same batch every step), shader clock from sysfs pp_dpm_sclk / rocm-smi"}
            pipe.run(opt, thr, mo, mt)
            torch.cuda.synchronize()
    ms = pipe.match_stats()
    hygiene["match_candidates_per_row"] = {"mean": ms["mean"], "max": ms["max"]}
    hygiene["match_overflow_rows"] = ms["overflow_rows"]
    hygiene["match_stats_note"] = (f"nomination lists of the last step's matcher call over {ms['rows']} live rows + columns (inline capacity {ms['inline_capacity']}; longer lists "
                                   "go through the parallel overflow pass): the matcher's time depends on them, its result never does")
    pcie = None
    pcie_u8 = None
    if not args.no_h2d and args.config == "c2":
        import collections
        ho, ht = opt.cpu().pin_memory(), thr.cpu().pin_memory()
        # the same images as 8-bit gray (what a camera / decoder delivers): a quarter of the upload; gray / 255 happens on the device (xp_u8_to_unit_f32)
        ho8, ht8 = (opt * 255.0).round().clamp(0, 255).to(torch.uint8).cpu().pin_memory(), (thr * 255.0).round().clamp(0, 255).to(torch.uint8).cpu().pin_memory()

        def stream_region(a, b):
            """K steps fed from pinned host memory with the result lists downloaded behind every step; the host keeps `pipe.depth` steps in flight
            (it waits for step i - depth + 1's download before it enqueues step i + 1: the pinned result buffers rotate over depth + 1 sets)"""
            with torch.no_grad():
                for _ in range(pipe.depth + 2):                                # untimed: the first download_async() calls allocate their pinned host buffers —
                    pipe.run(a, b, mo, mt)                                     # with the rank's CPU mask narrowed that allocation took ~0.3 s and sat inside this region
                    bufs, ev = pipe.download_async()                           # (PCIe-inclusive 1 455 -> 780 pairs/s for a 0.5 s region; the copies themselves run at
                    ev.synchronize()                                           # 52 GB/s either way: tools/h2d_affinity_probe.py, tools/pcie_affinity_probe.py)
                sync_all()
                t1 = time.perf_counter()
                pend = collections.deque()
                for _ in range(args.steps):
                    pipe.run(a, b, mo, mt)                                     # pinned host images uploaded straight into the batch buffers
                    bufs, ev = pipe.download_async()                           # counts, keypoints, match lists -> pinned host buffers behind the step
                    pend.append(ev)
                    if len(pend) >= pipe.depth:
                        pend.popleft().synchronize()                           # the oldest step in flight is on the host; depth - 1 newer ones keep the GPU busy
                sync_all()
                return world * B * args.steps / (time.perf_counter() - t1)
        pcie = stream_region(ho, ht)
        pcie_u8 = stream_region(ho8, ht8)
        if os.environ.get("XP_BENCH_PCIE_PARTS"):      # diagnosis only: which half of the streaming loop costs what (stderr)
            def region_parts(a, b, dl):
                with torch.no_grad():
                    for _ in range(4):
                        pipe.run(a, b, mo, mt)
                        if dl:
                            pipe.download_async()[1].synchronize()
                    sync_all()
                    t1 = time.perf_counter()
                    pend = collections.deque()
                    for _ in range(args.steps):
                        pipe.run(a, b, mo, mt)
                        if dl:
                            pend.append(pipe.download_async()[1])
                            if len(pend) >= pipe.depth:
                                pend.popleft().synchronize()
                    sync_all()
                    return world * B * args.steps / (time.perf_counter() - t1)
            sys.stderr.write("pcie parts: resident, no download %.1f | resident + download %.1f | host f32, no download %.1f | host u8, no download %.1f | host f32 + download %.1f\n" % (
                region_parts(opt, thr, False), region_parts(opt, thr, True), region_parts(ho, ht, False), region_parts(ho8, ht8, False), region_parts(ho, ht, True)))
        pipe.run(opt, thr, mo, mt)                                             # leave the buffers holding the headline inputs' results
        torch.cuda.synchronize()
    res = pipe.fetch()
    per_rank = xdist.gather_floats(region_dt, device=coll_dev) if dist.is_initialized() else [region_dt]
    # region r of the job = its slowest rank; the reported region = the median one
    job_dt = [max(per_rank[k][r] for k in range(len(per_rank))) for r in range(len(region_dt))]
    order = sorted(range(len(job_dt)), key=lambda r: job_dt[r])
    med = order[(len(order) - 1) // 2]
    dt = job_dt[med]
    rank_rates = [B * args.steps / per_rank[k][med] for k in range(len(per_rank))]
    rank_cpus = xdist.gather_strings(str(pin.get("cpus")), device=coll_dev) if dist.is_initialized() else [str(pin.get("cpus"))]
    if dist.is_initialized():
        # fixed-size result headers of every rank's last step, all-gathered (the only other collective)
        hdr = xdist.gather_headers(first, len(res), sum(len(r["kp_optical"]) + len(r["kp_thermal"]) for r in res),
                                   sum(len(r["match_q"]) for r in res), device=coll_dev)
        rccl["rank_headers"] = [dict(rank=i, first_pair=h[0], pairs=h[1], keypoints=h[2], matches=h[3]) for i, h in enumerate(hdr)]

    if rank == 0:
        avg_s = dom["ms"] / max(dom["launches"], 1) * 1e-3
        if dom["flops"] > 0 and dominant.startswith(("gemm", "conv3x3", "mlp_fused", "proj_mlp_fused")):
            ach = dom["flops"] / dom["launches"] / avg_s / 1e12
            x3 = "_x3" in dominant or "_h2" in dominant or "_f16" in dominant          # ("_h2" also matches the _h2r / _h2p schedules: same three-product arithmetic)
            nprod = 1 if "_f16" in dominant else (H2_PRODUCTS if "_h2" in dominant else X3_PRODUCTS)
            # split kernels: algorithmic (f32-equivalent) 2MNK flops against the 16-bit dense MFMA peak / partial products per multiply
            peak = MFMA_BF16_PEAK_TFLOPS / nprod if x3 else MFMA_F32_PEAK_TFLOPS
            roof = {"kernel": dominant, "bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": None,
                    "algorithmic_gflop_per_launch": round(dom["flops"] / dom["launches"] / 1e9, 3)}
            roof["frac_vs_f32_matrix_peak"] = round(ach / MFMA_F32_PEAK_TFLOPS, 4)      # orientation only: algorithmic flops against the exact-f32 MFMA peak
            if x3:
                roof["peak_note"] = (f"f32-equivalent: bf16 / fp16 dense MFMA peak {MFMA_BF16_PEAK_TFLOPS:.0f} TFLOP/s / {nprod} partial products per "
                                     f"f32-grade multiply; executed 16-bit MFMA rate = {ach * nprod:.0f} TFLOP/s = {ach * nprod / MFMA_BF16_PEAK_TFLOPS:.3f} of peak")
        else:
            ach = dom["bytes"] / dom["launches"] / avg_s / 1e9
            roof = {"kernel": dominant, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                    "algorithmic_mb_per_launch": round(dom["bytes"] / dom["launches"] / 1e6, 3)}
        roof.update(pmc_fields(dominant, roof["bound"], args.config, args.precision_class))
        if dominant.startswith("gemm_ring"):
            roof["dense_engine_ceilings"] = dense_engine_ceilings()
        roof.update({"avg_launch_us": round(avg_s * 1e6, 2), "launches_timed": dom["launches"]})
        if overlap or args.graph:
            roof["measured_in"] = ("3 single-stream eager passes of the same step next to the timed region: " +
                                   ("HIP events cannot be recorded inside a replayed graph" if args.graph else
                                    "in the timed region kernels of several streams run concurrently, so a launch's event duration includes time-sharing"))
        else:
            roof["share_of_step"] = round(dom["ms"] / (dt * 1e3), 4)
        tot = sum(r["ms"] for r in breakdown)
        sys.stderr.write("per-kernel breakdown of one step (HIP events, untimed pass):\n")
        for r in breakdown:
            extra = ""
            if r["flops"] > 0 and r["ms"] > 0:
                extra += f"  {r['flops'] / r['ms'] / 1e9:8.1f} TFLOP/s"
            if r["bytes"] > 0 and r["ms"] > 0:
                extra += f"  {r['bytes'] / r['ms'] / 1e6:8.1f} GB/s"
            sys.stderr.write(f"  {r['tag']:28s} {r['ms']:8.3f} ms {100 * r['ms'] / tot:5.1f}%  x{r['launches']:3d}{extra}\n")
        sys.stderr.write(f"  {'sum':28s} {tot:8.3f} ms\n")
        out = {
            "metric": "image-pairs/sec (detect+describe+match) 480x640 optical-thermal" + ("" if args.precision_class == "f32" else f" [NOT the headline: precision class {args.precision_class}]"),
            "value": round(world * B * args.steps / dt, 3), "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": ("f16 storage / f32 accumulate: the reference's mixed_precision (autocast) deployment recipe, NOT the headline class") if args.precision_class != "f32" else
                     ("f32 (dense layers: f32 operands as 2 fp16 terms (operand error <= 2^-23, dropped cross term <= 2^-22: per product <= 2^-21 worst case, ~2^-25 typical), 3 fp16-MFMA partial products, f32 accumulate — GEMMs, implicit-GEMM "
                      "convolutions and the fused block kernels alike; the deep-stage scan's dt projection on the same scheme; all other kernels f32)") if args.gemm == "h2" else
                     ("f32 (dense layers: f32 operands split exactly into 3 bf16 terms, 6 bf16-MFMA partial products, f32 accumulate; "
                      "all other kernels f32)") if args.gemm == "x3" else "f32",
            "precision_class": args.precision_class,
            "data": "synthetic",
            "config": {"workload": conf["label"] + f" (pairs/GPU/step = {B}): encode+detect(NMS 8, thr 0.015" + (f", keep_top_k {conf['topk']}" if conf["topk"] else "") +
                                   ")+describe+match(strict mutual NN)" + (", step replayed from hipGraphs" if args.graph else ""),
                       "pairs_per_gpu_per_step": B, "height": H, "width": W, "parallelism": f"pair-sharded x{world}, RCCL weight bcast",
                       "stream_overlap": (f"{pipe.depth + 1} HIP streams: the whole-batch encoders of {pipe.depth} consecutive steps (one stream each, taken in turn) overlap each "
                                          "other and the detection / matching kernels of the step before; all K steps complete inside the timed region") if (overlap and pipe.alternate) else
                                         (f"{1 + max(pipe.split_encoder, 1)} HIP streams: step i+1's encoder ({max(pipe.split_encoder, 1)} image group(s)) overlaps "
                                          "step i's detection / matching kernels; all K steps complete inside the timed region") if overlap else "none (one stream)",
                       "keypoints_per_image_mean": round(sum(len(r["kp_optical"]) + len(r["kp_thermal"]) for r in res) / (2 * len(res)), 1),
                       "matches_per_pair_mean": round(sum(len(r["match_q"]) for r in res) / len(res), 1)},
            "roofline": roof,
            "timed_regions": {"count": len(job_dt), "reported": "median", "pairs_per_s": [round(world * B * args.steps / t, 2) for t in job_dt],
                              "note": f"{len(job_dt)} regions of K = {args.steps} steps each (barrier + synchronize on both sides, max over ranks), back to back in this invocation"},
            "per_rank_pairs_per_s": {"min": round(min(rank_rates), 2), "max": round(max(rank_rates), 2),
                                     "note": "each rank's own pairs/s in the reported region (its own clock between the same barriers)"},
        }
        out.update(rccl)
        out.update(hygiene)
        out["rank_cpu_affinity"] = {"rank0": pin, "cpus_per_rank": rank_cpus, "note": "every rank pins itself (os.sched_setaffinity before torch is imported) to the CPUs of its GPU's NUMA node, "
                                                           "GPUs on one node splitting them (xpoint_amd/affinity.py); the CPU baseline runs on the unpinned mask"}
        # the other large kernels of the step, each against its own bound (from the untimed single-stream breakdown pass; the
        # `roofline` object above is the dominant one, timed next to the timed region)
        tot_ms = sum(r["ms"] for r in breakdown) or 1.0
        top = []
        for r in breakdown[:8]:
            if r["ms"] <= 0 or r["launches"] <= 0:
                continue
            mfma = r["flops"] > 0 and r["tag"].startswith(("gemm", "conv3x3", "mlp_fused", "proj_mlp_fused", "ln_proj"))
            if mfma:
                x3k = "_x3" in r["tag"] or "_h2" in r["tag"] or "_f16" in r["tag"]
                pk = MFMA_BF16_PEAK_TFLOPS / (1 if "_f16" in r["tag"] else H2_PRODUCTS if "_h2" in r["tag"] else X3_PRODUCTS) if x3k else MFMA_F32_PEAK_TFLOPS
                a = r["flops"] / r["ms"] / 1e9
                top.append({"kernel": r["tag"], "bound": "mfma", "achieved": round(a, 1), "peak": round(pk, 1), "unit": "TFLOP/s",
                            "frac": round(a / pk, 4), "share_of_single_stream_step": round(r["ms"] / tot_ms, 4), "launches": r["launches"]})
            elif r["bytes"] > 0:
                a = r["bytes"] / r["ms"] / 1e6
                top.append({"kernel": r["tag"], "bound": "hbm", "achieved": round(a, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(a / HBM_PEAK_GBS, 4), "share_of_single_stream_step": round(r["ms"] / tot_ms, 4), "launches": r["launches"]})
        out["roofline_top_kernels"] = top
        if single_rate is not None:
            out["single_stream"] = {"pairs_per_s": round(single_rate, 2), "note": "same steps enqueued on one HIP stream (bench.py --no-overlap)"}
        if other_rate is not None:
            out["other_gemm_backends"] = {"note": "same step with the dense layers on the other f32-grade back ends: x3 = split-bf16 (6 products), "
                                                  "h2 = split-fp16 (3 products), f32 = exact-f32 MFMA kernels",
                                          "pairs_per_s": {other: round(other_rate, 2), "f32": round(f32_rate, 2)}}
        if class_rates:
            out["reduced_precision_classes"] = {
                "note": "same step in the reduced-precision classes: outside the 1e-4 parity bar, never the headline value. x2 / bf16 = the split-bf16 kernels with "
                        "3 / 1 partial products (operands to 16 / 8 bits, f32 activations; reference prob error ~5e-5 / ~2e-2); amp16 = the reference's "
                        "mixed_precision recipe (XPoint.py:182 autocast, pinned op by op against the real reference under float16 autocast, tests/golden/g20): "
                        "fp16 rounding at every autocast boundary, fp16-rounded weights, scan / out_norm / softmax in f32 — values kept in f32 containers and "
                        "every block as separate launches, so it is a parity class, not a fast one; amp16f = the SAME recipe and rounding points with half "
                        "storage (fp16 tensors in HBM, one-product fp16 MFMA GEMMs fed by LDS-DMA, csrc/gemm_f16.hip): the fast deployment class, pinned by the same "
                        "fixture",
                "pairs_per_s": {k: round(v, 2) for k, v in class_rates.items()}}
        if pcie is not None:
            out["pcie_inclusive_pairs_per_s"] = round(pcie, 2)
            out["pcie_inclusive_u8_pairs_per_s"] = round(pcie_u8, 2)
        if dist.is_initialized():
            dist.destroy_process_group()          # the other ranks are done: the baseline below has the host to itself
        if not args.no_cpu_baseline:
            # rank 0, after the timed region, at every world size (bounded: 24 pairs at N = 1 = 10 - 30 s; 8 pairs = ~3 - 7 s beside an N-rank job, so that an
            # 8-GPU line carries the same keys as the 1-GPU line)
            try:
                if pin.get("applied"):
                    affinity.restore(pin["previous"])
                out["cpu_baseline"] = cpu_baseline(args.cpu_pairs if world == 1 else min(args.cpu_pairs, 8))
            except Exception as e:   # the baseline must never hide the measurement
                out["cpu_baseline"] = {"error": repr(e)}
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
