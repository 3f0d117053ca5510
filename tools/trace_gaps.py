"""tools/trace_gaps.py <dir with *_kernel_trace.csv> — busy time vs span of the kernels of a traced run: how much of the
wall time between the first and the last kernel is gaps between launches (launch-shaped workloads, S = 1)."""
import csv
import glob
import os
import sys

for f in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True)):
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
    rows = [r for r in rows if "bounceKernel" in r[2] or "flushKernel" in r[2] or "displayKernel" in r[2]]
    rows = rows[len(rows) // 4:]  # skip warm-up
    if not rows:
        continue
    span = rows[-1][1] - rows[0][0]
    busy = sum(e - s for s, e, _ in rows)
    gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
    per = {}
    for s, e, n in rows:
        k = n[n.index("ptss::"):][:60] if "ptss::" in n else n[:60]
        per.setdefault(k, []).append(e - s)
    print(f, "kernels", len(rows), "span %.2f ms busy %.2f ms (%.1f %%) mean gap %.2f us" % (span / 1e6, busy / 1e6, 100.0 * busy / span, sum(gaps) / len(gaps) / 1e3))
    for k, v in per.items():
        print("   %-62s n=%5d avg %8.2f us" % (k, len(v), sum(v) / len(v) / 1e3))
