"""Build xpoint_amd/libxpoint_hip.so (gfx950 only) with hipcc.  In-tree so that it travels to the
GPU box with the repo snapshot.  `python -m xpoint_amd.build [--force]`."""
import concurrent.futures as cf
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libxpoint_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         "-I", os.path.join(HERE, "..", "include"), "-I", CSRC] + os.environ.get("XP_EXTRA_HIPCC_FLAGS", "").split()


def source_hash(csrc: str = CSRC, include: str = os.path.join(HERE, "..", "include")) -> str:
    """sha256 (first 16 hex digits) over the kernel sources the library is built from (csrc/*.hip, *.cpp, *.h except the knob-description registry xp_knobs.h, include/*.h; names + bytes, sorted).
    Stored next to every PMC summary under profiles/ (tools/pmc_summary.py, tools/mfma_util.sh) and compared by bench.py: counter numbers
    quoted in a bench line that were collected on OTHER kernel sources are flagged `traffic_stale` / `frac_mfma_busy_pmc_stale`."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.cpp")) + glob.glob(os.path.join(csrc, "*.h")) +
                   glob.glob(os.path.join(include, "*.h")))
    for f in files:
        if os.path.basename(f) == "xp_knobs.h":      # the registry of knob DESCRIPTIONS: text, not kernel code
            continue
        h.update(os.path.basename(f).encode()); h.update(b"\0")
        h.update(open(f, "rb").read()); h.update(b"\0")
    return h.hexdigest()[:16]


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def _headers_mtime():
    hs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return max([os.path.getmtime(h) for h in hs] + [0.0])


def _compile(src, force, hm):
    obj = os.path.join(OBJ, os.path.basename(src) + ".o")
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), hm):
        return obj, False
    cmd = ["hipcc", "-x", "hip", "-c", src, "-o", obj] + FLAGS
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(force: bool = False, jobs: int = 6) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hm = _headers_mtime()
    srcs = _sources()
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        res = list(ex.map(lambda s: _compile(s, force, hm), srcs))
    objs = [o for o, _ in res]
    if force or any(ch for _, ch in res) or not os.path.exists(LIB):
        cmd = ["hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    if "--source-hash" in sys.argv:
        print(source_hash())
    else:
        print(build(force="--force" in sys.argv))
