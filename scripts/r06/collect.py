#!/usr/bin/env python3
"""Copies what scripts/r06/final_{a,b,c}.sh left under gpurun_out/r06 (scratch) to profiles/r06 and profiles/ (tracked),
and prints the bench lines' headline figures."""
import glob, json, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT, TO = os.path.join(ROOT, "gpurun_out", "r06"), os.path.join(ROOT, "profiles", "r06")
os.makedirs(TO, exist_ok=True)

def copy(source, target=None):
    path = os.path.join(OUT, source)
    if os.path.exists(path):
        shutil.copy(path, os.path.join(TO, target or os.path.basename(source)))

if os.path.exists(os.path.join(OUT, "pmc_fit_models.json")):
    shutil.copy(os.path.join(OUT, "pmc_fit_models.json"), os.path.join(ROOT, "profiles", "pmc_fit_models.json"))
copy("pmc_fit_models.txt", "pmc_fit_models_sq_counters.txt")
copy("pmc_mv_pieces_sq_counters.txt")
for name in ("segment_lengths.csv", "error_bound_sweep.csv", "macaque_streams.csv", "fit_few_chunks.csv"):
    copy(name)
for name in ("fit_latency", "mixed_fit", "mixed_fit_wave_counts", "mixed_grid", "segment_files_e2e", "irregular", "lossless_long_chunk",
             "pmc_wave_lossless"):
    copy(name + ".log", name + ".txt")
for name in ("timed_fit", "mixed", "fit_latency"):
    copy(os.path.join("prof_" + name, name + "_kernel_stats.csv"))
log = os.path.join(OUT, "prof_timed_fit.log")
if os.path.exists(log):
    lines = [l for l in open(log) if l.startswith('{"metric"')]
    if lines:
        open(os.path.join(TO, "bench_timed_fit_under_rocprof.json"), "w").write(lines[-1])
for path in sorted(glob.glob(os.path.join(OUT, "bench_*.json"))):
    if os.path.basename(path).startswith("bench_n"):
        continue  # (scratch runs of the round)
    lines = [l for l in open(path) if l.startswith('{"metric"')]
    if not lines:
        continue
    open(os.path.join(TO, os.path.basename(path)), "w").write(lines[-1])
    b = json.loads(lines[-1])
    if path.endswith("_detail.json"):
        continue  # (everything that run measured: kept as it is, the compact line is what is summarised below)
    r = b["roofline"]
    print(os.path.basename(path), b["metric"], f'{b["value"]:.4g}', "ms/step", round(b["ms_per_step"], 2), "|", r.get("kernel"),
          round(r["kernel_ms"], 2), "ms frac", round(r["frac"], 3), "traffic", r.get("traffic"), "valu_issue",
          (r.get("valu_issue") or {}).get("frac"), "| cpu", f'{b["cpu_baseline"]["value"]:.4g}')
