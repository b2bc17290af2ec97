"""Per-kernel means of rocprofv3 --pmc counters (one or more *_counter_collection.csv files), for kernels whose name contains a pattern.

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES ... --output-format csv -d gpurun_out/pmcA -o a -- python3 tools/prof_unet_train.py
    python tools/pmc_kernel.py k_wgrad gpurun_out/pmcA/a_counter_collection.csv [more.csv ...]"""
import csv
import sys
from collections import defaultdict

pat, paths = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(list))
for p in paths:
    with open(p) as f:
        for r in csv.DictReader(f):
            if pat in r["Kernel_Name"]:
                acc[(r["Kernel_Name"][:70], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:34s} n={len(v):4d} mean={sum(v) / len(v):16.1f}")
