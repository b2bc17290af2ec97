"""Idle gaps between consecutive kernels of each HIP stream (queue) in a rocprofv3 --kernel-trace CSV: how much of a chain's wall
time is kernel execution and how much is the dispatch gap between dependent launches.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python bench.py ...
    python tools/trace_gaps.py /tmp/tr
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root):
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    assert files, "no kernel trace under " + root
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    print(len(rows), "dispatches; columns:", list(rows[0].keys()))
    byq = defaultdict(list)
    for r in rows:
        byq[r.get("Queue_Id", "?")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    for q, ks in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        ks.sort()
        if len(ks) < 200:
            continue
        ks = ks[len(ks) // 2:]                         # the steady half (past warm-up, plan building)
        busy = sum(e - s for s, e, _ in ks)
        span = ks[-1][1] - ks[0][0]
        gaps = [(ks[i + 1][0] - ks[i][1], ks[i][2], ks[i + 1][2]) for i in range(len(ks) - 1)]
        pos = sorted(g for g, _, _ in gaps if g > 0)
        neg = [g for g, _, _ in gaps if g <= 0]
        print(f"queue {q}: {len(ks)} kernels over {span / 1e3:.0f} us; executing {busy / 1e3:.0f} us ({100 * busy / span:.1f} %), "
              f"{len(neg)} overlapped starts")
        if pos:
            pct = lambda p: pos[min(len(pos) - 1, int(p * len(pos)))] / 1e3
            print(f"   gaps: n {len(pos)}, sum {sum(pos) / 1e3:.0f} us, median {pct(0.5):.2f}, p10 {pct(0.1):.2f}, p90 {pct(0.9):.2f}, max {pos[-1] / 1e3:.1f} us")
        bykind = defaultdict(list)
        for g, a, b in gaps:
            bykind[b.split("(")[0][:50]].append(g)
        for k, v in sorted(bykind.items(), key=lambda kv: -sum(kv[1]))[:8]:
            print(f"   before {k:50s} n {len(v):5d} mean gap {sum(v) / len(v) / 1e3:6.2f} us")


if __name__ == "__main__":
    main(sys.argv[1])
