"""Rank kernels of a rocprofv3 --pmc run by a counter's total (and show a second counter's total next to it).

    python tools/pmc_rank.py counter_collection.csv SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"""
import csv, sys
from collections import defaultdict
path, c1, c2 = sys.argv[1], sys.argv[2], sys.argv[3]
tot = defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"][:80]
    if r["Counter_Name"] == c1: tot[k][0] += float(r["Counter_Value"]); tot[k][2] += 1
    elif r["Counter_Name"] == c2: tot[k][1] += float(r["Counter_Value"])
for k, (a, b, n) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"{a:16.0f} {b:16.0f} {a / b if b else 0:6.2f} {n:6d}  {k}")
