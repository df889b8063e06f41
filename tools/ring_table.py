"""Condense gpurun_out/ring_*.txt (tools/ring_bench output) into one table: median us per shape x tile variant, main build and stage-removal builds."""
import re, sys, glob, os
files = sys.argv[1:] or sorted(glob.glob("gpurun_out/ring_*.txt"))
tabs = {}
for f in files:
    tag = os.path.basename(f).replace("ring_", "").replace(".txt", "")
    shape = None
    for line in open(f):
        m = re.match(r"planes (\d)  M (\d+) N (\d+) K (\d+) act (\d) res (\d)", line)
        if m: shape = "p%s %6s x%5s x%5s a%s r%s" % m.groups(); continue
        m = re.match(r"\s+(\S+ \S+ \S+)\s+grid\s+(\d+).*median\s+([\d.]+) us", line)
        if m and shape: tabs.setdefault(shape, {}).setdefault(m.group(1), {})[tag] = float(m.group(3))
        if "!!" in line: print(f, shape, line.strip())
for shape, vs in tabs.items():
    tags = sorted({t for v in vs.values() for t in v}, key=lambda t: (t != "main", t))
    print("\n" + shape + "    " + "  ".join("%8s" % t for t in tags))
    for v, d in vs.items():
        print("  %-22s" % v + "  ".join("%8.1f" % d[t] if t in d else "       -" for t in tags))
