#!/usr/bin/env python3
"""grid() and the segment aggregates away from the bench's ~743 points per segment, and MacaqueV with
many streams. Three tables (CSV, kept under profiles/rNN/):

  segment_lengths.csv   runs of exactly 8 / 16 / 64 / 256 / 743 / 4096 points (constants and lines with a
                        jump between runs); grid and SUM+COUNT+MIN+MAX over the segments, achieved
                        GB/s against the algorithmic bytes 73*S + 12*N (grid: segment rows read + both
                        columns written) and 73*S (aggregates: nothing is written).
  error_bound_sweep.csv sine + uniform noise under a relative bound of 50 % ... 0.1 %: the same columns.
  macaque_streams.csv   lossless fit (every segment a MacaqueV stream) of 10^3 / 10^4 / 10^5 streams of
                        50 000 values: grid and SUM, values/s and GB/s against 12 B per value written
                        plus the stream bytes read.

Usage (on the GPU box): python3 scripts/profile_segment_lengths.py [out_dir]
"""
import csv
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import modelardb_rs_amd as mdb  # noqa: E402

HBM_PEAK_GBPS = 8000.0
SEGMENT_ROW_BYTES = 73  # one row of the compressed schema with short inline payloads
TARGETS = (8, 16, 64, 256, 743, 4096)
ALL = mdb.MDB_AGG_SUM | mdb.MDB_AGG_COUNT | mdb.MDB_AGG_MIN | mdb.MDB_AGG_MAX


def noisy(n, seed):
    rng = np.random.default_rng(seed)
    i = np.arange(n, dtype=np.float64)
    return (100.0 + 10.0 * np.sin(i / 2000.0) + rng.uniform(-0.5, 0.5, n)).astype(np.float32)


def timed(ctx, call, repetitions=3):
    call()
    ctx.profile_enable(True); ctx.profile_reset(); ctx.sync()
    started = time.perf_counter()
    for _ in range(repetitions):
        call()
    ctx.sync()
    wall = (time.perf_counter() - started) / repetitions
    kernels = {name: total / max(calls, 1) for name, (calls, total) in ctx.profile().items()}
    ctx.profile_enable(False)
    return wall, kernels


def fit(ctx, values_dev, n, chunk, eb):
    offsets = np.arange(0, n + chunk, chunk, dtype=np.uint64)
    offsets[-1] = n
    offsets_dev = ctx.upload_array(offsets)
    ctx.sync(); started = time.perf_counter()
    dev = ctx.compress_chunks_dev(0, values_dev, offsets_dev, len(offsets) - 1, eb, 0, 1000, 0)
    ctx.sync(); seconds = time.perf_counter() - started
    ctx.dev_free(offsets_dev)
    return dev, seconds


def runs(n, length, seed):
    """Runs of exactly `length` points: every other run a constant, the rest a line, a jump far outside
    the error bound between runs and a jitter well inside it within a run. The greedy fit ends a
    PMC-Mean or Swing segment at (nearly) every jump, so the mean segment length is close to `length`
    without MacaqueV taking over (on plain noisy data the first position where PMC-Mean and Swing both
    stop after one or two points hands the rest of the chunk to MacaqueV: see error_bound_sweep)."""
    rng = np.random.default_rng(seed)
    n_runs = (n + length - 1) // length
    level = np.repeat(rng.normal(0.0, 50.0, n_runs).astype(np.float64), length)[:n]
    slope = np.repeat(np.where(np.arange(n_runs) % 2 == 0, 0.0, rng.normal(0.0, 0.05, n_runs)), length)[:n]
    within = (np.arange(n, dtype=np.int64) % length).astype(np.float64)
    return (level + slope * within + rng.uniform(-0.002, 0.002, n)).astype(np.float32)


def payload_bytes(segments):
    return int(sum(int(col.lengths().astype(np.int64).sum()) for col in (segments.timestamps, segments.values,
                                                                           segments.residuals)))


def measure(ctx, dev, n_points):
    count = ctx.grid_count_dev(dev)
    assert count == n_points, (count, n_points)
    out_ts, out_val = ctx.dev_alloc(8 * count), ctx.dev_alloc(4 * count)
    grid_wall, grid_kernels = timed(ctx, lambda: ctx.grid_batch_dev(dev, out_ts, out_val, count))
    agg_wall, agg_kernels = timed(ctx, lambda: ctx.agg_batch_dev(dev, ALL))
    for pointer in (out_ts, out_val):
        ctx.dev_free(pointer)
    return grid_wall, grid_kernels, agg_wall, agg_kernels


def top(kernels):
    return " ".join(f"{name}={ms:.3f}" for name, ms in sorted(kernels.items(), key=lambda kv: -kv[1])[:4] if ms > 0.01)


def row_of(ctx, dev, n, fit_seconds, head):
    s = len(dev)
    mix = np.bincount(np.asarray(dev.download().model_type_id), minlength=3).tolist() if s <= 40_000_000 else None
    grid_wall, grid_kernels, agg_wall, agg_kernels = measure(ctx, dev, n)
    grid_bytes = SEGMENT_ROW_BYTES * s + 12 * n
    row = dict(head)
    row.update({
        "points": n, "segments": s, "mean_points_per_segment": f"{n / s:.1f}",
        "pmc_swing_macaquev": "/".join(map(str, mix)) if mix else "",
        "fit_ms": f"{fit_seconds * 1e3:.2f}",
        "grid_ms": f"{grid_wall * 1e3:.3f}", "grid_values_per_s": f"{n / grid_wall:.4g}",
        "grid_GBps": f"{grid_bytes / grid_wall / 1e9:.1f}",
        "grid_frac_of_hbm": f"{grid_bytes / grid_wall / 1e9 / HBM_PEAK_GBPS:.3f}",
        "agg_ms": f"{agg_wall * 1e3:.3f}", "agg_values_per_s": f"{n / agg_wall:.4g}",
        "agg_GBps": f"{SEGMENT_ROW_BYTES * s / agg_wall / 1e9:.1f}",
        "agg_frac_of_hbm": f"{SEGMENT_ROW_BYTES * s / agg_wall / 1e9 / HBM_PEAK_GBPS:.3f}",
        "grid_kernels_ms": top(grid_kernels), "agg_kernels_ms": top(agg_kernels),
    })
    print(row, flush=True)
    return row


def segment_lengths(ctx, out_dir, n=1 << 28):
    rows = []
    eb = mdb.error_bound("absolute", 0.01)
    for target in TARGETS:
        values_dev = ctx.upload_array(runs(n, target, 11 + target))
        dev, fit_seconds = fit(ctx, values_dev, n, 65536, eb)
        ctx.dev_free(values_dev)
        rows.append(row_of(ctx, dev, n, fit_seconds, {"run_length": target, "error_bound": "absolute 0.01"}))
        dev.free()
    write_csv(os.path.join(out_dir, "segment_lengths.csv"), rows)


def error_bound_sweep(ctx, out_dir, n=1 << 28):
    """Plain noisy data under a tightening relative bound: segments get shorter until MacaqueV takes
    whole chunks."""
    rows = []
    values_dev = ctx.upload_array(noisy(n, 11))
    for percent in (50.0, 20.0, 10.0, 5.0, 2.0, 1.0, 0.7, 0.5, 0.3, 0.1):
        dev, fit_seconds = fit(ctx, values_dev, n, 65536, mdb.error_bound("relative", percent))
        rows.append(row_of(ctx, dev, n, fit_seconds, {"relative_error_bound_percent": percent}))
        dev.free()
    ctx.dev_free(values_dev)
    write_csv(os.path.join(out_dir, "error_bound_sweep.csv"), rows)


def macaque_streams(ctx, out_dir, values_per_stream=50_000):
    rows = []
    eb = mdb.error_bound("lossless")
    for streams in (1_000, 10_000, 100_000):
        n = streams * values_per_stream
        values_dev = ctx.dev_alloc(4 * n)
        ctx.synth_values_dev(values_dev, 0, streams, values_per_stream)
        dev, fit_seconds = fit(ctx, values_dev, n, values_per_stream, eb)
        ctx.dev_free(values_dev)
        s = len(dev)
        seg = dev.seg
        stream_bytes = sum(int(seg.values.buffer_sizes[b]) for b in range(seg.values.n_buffers))
        grid_wall, grid_kernels, agg_wall, agg_kernels = measure(ctx, dev, n)
        grid_bytes = stream_bytes + SEGMENT_ROW_BYTES * s + 12 * n
        rows.append({
            "streams": s, "values_per_stream": values_per_stream, "values": n,
            "stream_bytes": stream_bytes, "bits_per_value": f"{8 * stream_bytes / n:.2f}",
            "fit_ms": f"{fit_seconds * 1e3:.2f}",
            "grid_ms": f"{grid_wall * 1e3:.3f}", "grid_values_per_s": f"{n / grid_wall:.4g}",
            "grid_GBps": f"{grid_bytes / grid_wall / 1e9:.1f}",
            "grid_frac_of_hbm": f"{grid_bytes / grid_wall / 1e9 / HBM_PEAK_GBPS:.3f}",
            "sum_ms": f"{agg_wall * 1e3:.3f}", "sum_values_per_s": f"{n / agg_wall:.4g}",
            "sum_GBps": f"{(stream_bytes + SEGMENT_ROW_BYTES * s) / agg_wall / 1e9:.1f}",
            "grid_kernels_ms": top(grid_kernels), "agg_kernels_ms": top(agg_kernels),
        })
        print(rows[-1], flush=True)
        dev.free()
    write_csv(os.path.join(out_dir, "macaque_streams.csv"), rows)


def write_csv(path, rows):
    with open(path, "w", newline="") as f:
        writer = csv.DictWriter(f, fieldnames=list(rows[0]))
        writer.writeheader()
        writer.writerows(rows)
    print("wrote", path, flush=True)


def main():
    out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "segment_lengths")
    os.makedirs(out_dir, exist_ok=True)
    ctx = mdb.Context(0)
    which = sys.argv[2] if len(sys.argv) > 2 else "all"
    if which in ("all", "lengths"):
        segment_lengths(ctx, out_dir)
    if which in ("all", "sweep"):
        error_bound_sweep(ctx, out_dir)
    if which in ("all", "streams"):
        macaque_streams(ctx, out_dir)


main()
