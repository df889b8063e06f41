"""GPU idle time and concurrency from a rocprofv3 kernel trace.  usage: python tools/trace_gaps.py <kernel_trace.csv> [skip_fraction]
Reports, over the last (1 - skip) part of the trace: wall span, time with >= 1 kernel running, idle time, sum of kernel durations."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * skip):]
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
busy = 0; depth = 0; last = t0; conc = {}
for t, d in ev:
    if depth > 0:
        busy += t - last
    conc[depth] = conc.get(depth, 0) + (t - last)
    depth += d; last = t
tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print(f"kernels {len(rows)}  span {(t1 - t0) / 1e6:.3f} ms  busy {busy / 1e6:.3f} ms  idle {(t1 - t0 - busy) / 1e6:.3f} ms ({100 * (t1 - t0 - busy) / (t1 - t0):.1f} %)  sum of durations {tot / 1e6:.3f} ms")
print("time by number of kernels in flight:", {k: round(v / 1e6, 3) for k, v in sorted(conc.items())})
