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
    print(build(force="--force" in sys.argv))
