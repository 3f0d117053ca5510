"""tools/lanes_trace.py <dir with *_kernel_trace.csv> [trailing kernels to look at] — per-queue view of a traced frame-lanes run (tools/lanes_bench.py under
rocprofv3 --kernel-trace): kernels, busy time and gaps per hardware queue, and how much of the span has 0 / 1 / 2+ kernels
in flight."""
import csv
import glob
import os
import sys

for f in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "bounceKernel" in r["Kernel_Name"] or "flushKernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-int(sys.argv[2]):] if len(sys.argv) > 2 else rows[len(rows) // 2:]   # the last configuration traced, warm
    if not rows:
        continue
    t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
    per = {}
    for r in rows:
        per.setdefault(r.get("Queue_Id", "?"), []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    print(f, "kernels", len(rows), "span %.2f ms" % ((t1 - t0) / 1e6))
    for q, v in sorted(per.items()):
        busy = sum(e - s for s, e in v)
        gaps = [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
        print("   queue %-6s n=%5d busy %.2f ms (%.1f %% of span) mean gap %.2f us, max %.1f us" %
              (q, len(v), busy / 1e6, 100.0 * busy / (t1 - t0), sum(gaps) / max(1, len(gaps)) / 1e3, max(gaps) / 1e3 if gaps else 0))
    ev = sorted([(int(r["Start_Timestamp"]), 1) for r in rows] + [(int(r["End_Timestamp"]), -1) for r in rows])
    depth, last, hist = 0, t0, {}
    for t, d in ev:
        hist[min(depth, 3)] = hist.get(min(depth, 3), 0) + (t - last)
        depth += d
        last = t
    print("   kernels in flight: " + ", ".join("%d%s: %.1f %%" % (k, "+" if k == 3 else "", 100.0 * v / (t1 - t0)) for k, v in sorted(hist.items())))
