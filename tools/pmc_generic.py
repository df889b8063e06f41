"""Per-kernel averages of arbitrary rocprofv3 --pmc passes.
usage: python tools/pmc_generic.py <dir-with-counter_collection.csv files (searched recursively)> [kernel-substring]"""
import collections, csv, glob, os, re, sys
root = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""
d = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"])[:70]
        if filt and filt not in k:
            continue
        e = d[k][r["Counter_Name"]]; e[0] += float(r["Counter_Value"]); e[1] += 1
for k in sorted(d):
    print(k)
    for c in sorted(d[k]):
        s, n = d[k][c]
        print(f"    {c:32s} {s / n:16.1f}   (n={n})")
