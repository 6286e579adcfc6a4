#!/usr/bin/env python3
"""HBM-side traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).

    python tools/make_traffic_json.py <dir with pmc_fetch/ and pmc_write/> [out.json]

rocprofv3 reports both counters in KiB. On gfx950 FETCH_SIZE counts 128-byte fabric read
requests as 64 bytes for 16-byte-per-lane loads (the loads these kernels issue), so it is
doubled; WRITE_SIZE is exact. Result: bytes per launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024,
averaged over the launches of each kernel (Infinity-Cache hits are included in these
counters, so for a cache-resident scene this is an upper bound on what HBM itself served)."""
import collections
import csv
import glob
import json
import sys
from pathlib import Path


def mean_by_kernel(d, counter):
    agg = collections.defaultdict(list)
    for path in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].replace("rtd::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            if name.startswith("k_"):
                agg[name].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


def main():
    d = Path(sys.argv[1])
    out = Path(sys.argv[2]) if len(sys.argv) > 2 else Path(__file__).resolve().parent.parent / "profiles" / "hbm_traffic.json"
    fetch, nf = mean_by_kernel(d / "pmc_fetch", "FETCH_SIZE")
    write, _ = mean_by_kernel(d / "pmc_write", "WRITE_SIZE")
    res = {"_note": "bytes per launch = 2*FETCH_SIZE_KiB*1024 + WRITE_SIZE_KiB*1024 (gfx950 FETCH_SIZE half-count "
                    "correction for 16-byte-per-lane loads); workload: tools/prof_workload.py --render 1 --batch 1"}
    for k in sorted(fetch):
        res[k] = 2.0 * fetch[k] * 1024.0 + write.get(k, 0.0) * 1024.0
        res[k + "__detail"] = {"fetch_size_kib": fetch[k], "write_size_kib": write.get(k, 0.0), "launches": nf[k]}
    out.write_text(json.dumps(res, indent=1) + "\n")
    print(out.read_text())


if __name__ == "__main__":
    main()
