#!/usr/bin/env python3
"""Copies the judged summaries of the last scripts/gpu_profile.sh run from gpurun_out/ (scratch) to
profiles/<round>/ and derives profiles/pmc_grid_tiles.json, which bench.py reads to report
roofline.traffic. FETCH_SIZE is doubled (gfx950 reports half the bytes of wide streaming reads,
MI355X_MICROARCH.md section HBM); WRITE_SIZE is taken as is. Both are in KiB... the counters are in
kilobytes: hbm_bytes = value * 1024."""

import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")


def counters(name):
    rows = list(csv.DictReader(open(os.path.join(OUT, f"prof_{name}", f"{name}_counter_collection.csv"))))
    table = collections.defaultdict(list)
    for row in rows:
        table[(row["Kernel_Name"].split("(")[0], row["Counter_Name"])].append(float(row["Counter_Value"]))
    return table


def main():
    round_dir = os.path.join(ROOT, "profiles", sys.argv[1] if len(sys.argv) > 1 else "r02")
    os.makedirs(round_dir, exist_ok=True)
    shutil.copy(os.path.join(OUT, "prof_trace", "trace_kernel_stats.csv"),
                os.path.join(round_dir, "bench_kernel_stats.csv"))
    bench_line = [l for l in open(os.path.join(OUT, "prof_trace.log")) if l.startswith('{"metric"')][-1]
    open(os.path.join(round_dir, "bench_under_rocprof.json"), "w").write(bench_line)
    bench = json.loads(bench_line)
    summary = {}
    for name in ("fetch", "write"):
        for (kernel, counter), values in counters(name).items():
            summary.setdefault(kernel, {})[counter] = {
                "dispatches": len(values), "mean_kb": sum(values) / len(values), "max_kb": max(values)}
    json.dump(summary, open(os.path.join(round_dir, "bench_pmc_fetch_write.json"), "w"), indent=1, sort_keys=True)
    tiles = summary["mdb::k_grid_tiles"]
    config = bench["config"]
    launches_per_step = 1
    points = config["series_per_gpu"] * config["points_per_series"]
    segments = config["segments_per_gpu"]
    n_launch = tiles["WRITE_SIZE"]["dispatches"]
    write_bytes = tiles["WRITE_SIZE"]["mean_kb"] * 1024
    fetch_bytes = tiles["FETCH_SIZE"]["mean_kb"] * 1024
    # mean over dispatches; a dispatch covers points / dispatches_per_step points
    per_step = {"fetch": 1, "write": 1}
    steps_fetch = 1  # gpu_profile.sh runs the PMC passes with --steps 1 --warmup 0 (+ roofline loop of 1)
    dispatches_per_step = n_launch / 2.0
    pmc = {
        "series": config["series_per_gpu"], "points": config["points_per_series"],
        "dispatches_profiled": n_launch,
        "write_bytes_per_point": write_bytes * dispatches_per_step / points,
        "fetch_bytes_per_segment_reported": fetch_bytes * dispatches_per_step / segments,
        "fetch_bytes_per_segment_corrected": 2.0 * fetch_bytes * dispatches_per_step / segments,
        "source": f"profiles/{os.path.basename(round_dir)}/bench_pmc_fetch_write.json (rocprofv3 --pmc, separate passes)",
        "source_hash": open(os.path.join(OUT, "prof_source_hash.txt")).read().strip(),
    }
    json.dump(pmc, open(os.path.join(ROOT, "profiles", "pmc_grid_tiles.json"), "w"), indent=1)
    print(json.dumps(pmc, indent=1))
    for line in open(os.path.join(round_dir, "bench_kernel_stats.csv")).read().splitlines()[:6]:
        print(line[:160])


if __name__ == "__main__":
    main()
