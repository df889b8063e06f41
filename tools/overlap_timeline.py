"""Timeline summary of a rocprofv3 --kernel-trace CSV of the overlapped pair step: how much of the wall time has 0 / 1 / 2 / 3+ kernels in flight, busy time
per queue, and the longest gaps.   usage: python tools/overlap_timeline.py <kernel_trace.csv> [skip_fraction]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
stop = float(sys.argv[3]) if len(sys.argv) > 3 else 0.6
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((s, e, r.get("Queue_Id", "?"), r["Kernel_Name"]))
ev.sort()
ev = ev[int(len(ev) * skip):int(len(ev) * stop)]          # a slice of the launch sequence (by kernel index): inside the long timed region
t0, t1 = ev[0][0], max(e for _, e, _, _ in ev)
pts = []
for s, e, q, k in ev:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
hist = collections.Counter(); cur = 0; last = t0; gaps = []
for t, d in pts:
    hist[min(cur, 4)] += t - last
    if cur == 0 and t - last > 0: gaps.append((t - last, last))
    cur += d; last = t
wall = t1 - t0
print(f"window {wall / 1e6:.3f} ms, {len(ev)} kernels")
for k in sorted(hist): print(f"  {k}{'+' if k == 4 else ' '} kernels in flight: {hist[k] / 1e6:8.3f} ms  {100.0 * hist[k] / wall:5.1f} %")
busy = collections.defaultdict(int)
for s, e, q, k in ev: busy[q] += e - s
for q in sorted(busy): print(f"  queue {q}: kernel time {busy[q] / 1e6:8.3f} ms  ({100.0 * busy[q] / wall:5.1f} % of the window)")
print(f"  sum of kernel durations / wall = {sum(busy.values()) / wall:.2f}")
gaps.sort(reverse=True)
print("  longest idle gaps (us):", ", ".join(f"{g / 1e3:.1f}" for g, _ in gaps[:10]), f"; total idle {sum(g for g, _ in gaps) / 1e6:.3f} ms")
dur = collections.defaultdict(lambda: [0, 0])
for s, e, q, k in ev:
    k = k.split("(")[0][-60:]; dur[k][0] += e - s; dur[k][1] += 1
print("  top kernels by summed duration inside the window:")
for k, (d, n) in sorted(dur.items(), key=lambda kv: -kv[1][0])[:14]: print(f"    {d / 1e6:8.3f} ms  x{n:4d}  avg {d / n / 1e3:8.1f} us  {k}")
