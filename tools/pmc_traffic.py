"""HBM traffic per launch of a kernel from two rocprofv3 counter passes (MI355X_MICROARCH.md, HBM section: FETCH_SIZE and WRITE_SIZE
cannot share a pass; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced streaming reads -> doubled; both count KiB).

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py --no-cpu-baseline --no-train-step --no-fp32-mode --no-configs --steps 4 --warmup 1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py ...   (same command)
    python tools/pmc_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv \\
           k_tauleap_s256_b16 401408 profiles/r03_pmc_traffic_tauleap_s256_b16.json [algorithmic bytes per launch]

Only dispatches of the named kernel with the given grid size (threads; 401408 = the 1568 workgroups of a 256-sample launch) count.
Also writes per-kernel means of both counters next to the JSON (…_per_kernel.csv)."""
import csv
import json
import sys
from collections import defaultdict


def read(path, counter):
    per = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                per[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return per


def main():
    fpath, wpath, kname, grid, out = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
    algo = int(sys.argv[6]) if len(sys.argv) > 6 else None
    fetch, write = read(fpath, "FETCH_SIZE"), read(wpath, "WRITE_SIZE")
    f = [v for (k, g), vs in fetch.items() if kname in k and g == grid for v in vs]
    w = [v for (k, g), vs in write.items() if kname in k and g == grid for v in vs]
    if not f or not w:
        raise SystemExit(f"no dispatch of {kname} with grid {grid}: grids seen {sorted({g for (k, g) in fetch if kname in k})}")
    fm, wm = sum(f) / len(f), sum(w) / len(w)
    rec = {"kernel": kname, "grid_threads": grid, "launches_averaged": {"fetch_pass": len(f), "write_pass": len(w)},
           "workload": "bench.py MNIST tauLDR, batch 256 (rocprofv3 --pmc, separate passes, --steps 4 --warmup 1; the whole-batch launches of the roofline section)",
           "FETCH_SIZE_mean_KB": fm, "WRITE_SIZE_mean_KB": wm,
           "corrections": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section), KB = 1024 B; WRITE_SIZE as read",
           "read_bytes": 2 * fm * 1024, "write_bytes": wm * 1024, "hbm_bytes_per_launch": 2 * fm * 1024 + wm * 1024}
    if algo:
        rec["algorithmic_bytes_per_launch"] = algo
        rec["traffic_over_algorithmic"] = round(rec["hbm_bytes_per_launch"] / algo, 3)
    with open(out, "w") as fo:
        json.dump(rec, fo, indent=1)
    print(json.dumps(rec, indent=1))
    stem = out[:-5] if out.endswith(".json") else out
    with open(stem + "_per_kernel.csv", "w", newline="") as fo:
        wr = csv.writer(fo)
        wr.writerow(["Kernel_Name", "Grid_Size", "dispatches", "FETCH_SIZE_mean_KB(uncorrected)", "WRITE_SIZE_mean_KB"])
        keys = sorted(set(fetch) | set(write), key=lambda k: -sum(fetch.get(k, [0])))
        for k in keys[:60]:
            fv, wv = fetch.get(k, []), write.get(k, [])
            wr.writerow([k[0][:90], k[1], len(fv), round(sum(fv) / max(len(fv), 1), 2), round(sum(wv) / max(len(wv), 1), 2)])


if __name__ == "__main__":
    main()
