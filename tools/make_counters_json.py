#!/usr/bin/env python3
"""profiles/rNN/counters.json from the outputs of tools/prof_round.sh (run here, after gpurun merged them):

    python tools/make_counters_json.py gpurun_out/r02_hippo gpurun_out/r02_deep profiles/r02

Per workload and traversal kernel: the mean of every PMC counter per launch (pmc_summary.txt), the kernel's
average duration from the kernel trace (kernel_stats.csv), the rays per launch of the profiled workload, and the
fast walk's visits per ray from the counting build. bench.py combines these with launch times it measures live.
FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE counts 128-byte requests as 64 bytes on gfx950 (double it)."""
import csv
import json
import re
import shutil
import subprocess
import sys
from pathlib import Path


def pmc(path):
    out, cur = {}, None
    for line in Path(path).read_text().splitlines():
        if line.startswith("k_"):
            cur = line.split(" calls ")[0].strip()
            out.setdefault(cur, {})
        elif cur and line.strip():
            parts = line.split()
            out[cur][parts[0]] = float(parts[2])
    return out


def kernel_ms(path):
    out = {}
    for r in csv.DictReader(open(path)):
        name = r["Name"].replace("rtd::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        out[name] = {"kernel_ms": float(r["AverageNs"]) / 1e6, "calls": int(r["Calls"])}
    return out


def visits(path):
    lines = Path(path).read_text().splitlines()
    for line in lines:
        if line.startswith("{\"fast_walk_rays\""):
            d = json.loads(line)
            out = {k: d[k] for k in ("node_steps_per_ray", "screened_per_ray", "exact_tests_per_ray")}
            for l in lines:  # "  leaf   rounds N  lanes per round L ...": one 128-byte pair record per lane and round
                m = re.match(r"\s+leaf\s+rounds\s+(\d+)\s+lanes per round\s+([0-9.]+)", l)
                if m:
                    out["leaf_visits_per_ray"] = round(int(m.group(1)) * float(m.group(2)) / d["fast_walk_rays"], 3)
            return out, d["fast_walk_rays"]
    raise SystemExit(f"no visit counts in {path}")


def rounds(path):
    return [l for l in Path(path).read_text().splitlines() if l.startswith("  ") and "rounds" in l]


def main():
    hippo, deep, dest = Path(sys.argv[1]), Path(sys.argv[2]), Path(sys.argv[3])
    dest.mkdir(parents=True, exist_ok=True)
    res = {"_note": "per-launch means; workloads: tools/prof_workload.py --render 1 --batch 1 (hippo: 16 samples per pixel at 1920x1080 in one pass "
                    "= 5 path launches; one 16,777,216-ray batch launch) and "
                    "--scene deep --batch 1; FETCH_SIZE / WRITE_SIZE in KiB (FETCH_SIZE x 2 on gfx950)"}
    try:  # the state of the tree the profiled library was built from
        root = Path(__file__).resolve().parent.parent
        head = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=root, capture_output=True, text=True).stdout.strip()
        dirty = subprocess.run(["git", "status", "--porcelain", "--", "ray-tracer_amd", "include"], cwd=root,
                               capture_output=True, text=True).stdout.strip()
        res["commit"] = head + (" + uncommitted changes" if dirty else "")
    except OSError:
        pass
    hp, hk = pmc(hippo / "pmc_summary.txt"), kernel_ms(hippo / "kernel_stats.csv")
    hv, hv_rays = visits(hippo / "visits_render.txt")
    bv, _ = visits(hippo / "visits_batch.txt")
    n_launch = max(1, hk["k_traverse_paths_fast"]["calls"] // 2)  # the statistics pass renders twice what the counter passes and the counting build render
    res["hippo_render_1080p"] = {
        "k_traverse_paths_fast": dict(hp["k_traverse_paths_fast"], rays_per_launch=hv_rays / n_launch,
                                      **hk["k_traverse_paths_fast"]),
        "k_shade": dict(hp.get("k_shade<false>", {}), **hk.get("k_shade<false>", {})),
        "fast_walk_visits_per_ray": hv}
    # the batch launches of a small scene run the two-rays-per-lane kernel
    bk = "k_intersect_fast2" if "k_intersect_fast2" in hp else "k_intersect_fast"
    res["hippo_batch_16m"] = {bk: dict(hp[bk], rays_per_launch=1 << 24, **hk[bk]), "kernel": bk,
                              "fast_walk_visits_per_ray": bv}
    dp, dk = pmc(deep / "pmc_summary.txt"), kernel_ms(deep / "kernel_stats.csv")
    dv, _ = visits(deep / "visits_batch.txt")
    res["deep_batch_16m"] = {"k_intersect_fast": dict(dp["k_intersect_fast"], rays_per_launch=1 << 24, **dk["k_intersect_fast"]),
                             "k_rewalk_batch": dict(dp.get("k_rewalk_batch", {}), **dk.get("k_rewalk_batch", {})),
                             "fast_walk_visits_per_ray": dv}
    (dest / "counters.json").write_text(json.dumps(res, indent=1) + "\n")
    for src, tag in ((hippo, "hippo"), (deep, "deep")):
        shutil.copy(src / "pmc_summary.txt", dest / f"pmc_summary_{tag}.txt")
        shutil.copy(src / "kernel_stats.csv", dest / f"kernel_stats_{tag}.csv")
    (dest / "fast_walk_rounds.txt").write_text(
        "hippo render (1080p, 16 spp):\n" + "\n".join(rounds(hippo / "visits_render.txt")) + "\nhippo batch:\n" +
        "\n".join(rounds(hippo / "visits_batch.txt")) + "\ndeep batch:\n" + "\n".join(rounds(deep / "visits_batch.txt")) + "\n")
    print(json.dumps(res, indent=1)[:3000])


if __name__ == "__main__":
    main()
