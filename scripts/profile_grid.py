#!/usr/bin/env python3
"""Measure the grid kernels on oracle-fitted segments (no GPU fitter needed): fits a few distinct
sine series with the CPU oracle, tiles their segments to many series on the host, uploads once and
times mdb_grid_batch_dev with the library's HIP-event profiler. Development tool, not the bench."""

import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import datagen  # noqa: E402
import modelardb_rs_amd as mdb  # noqa: E402
import oracle_lib as ora  # noqa: E402


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--distinct", type=int, default=8)
    parser.add_argument("--points", type=int, default=2_000_000)
    parser.add_argument("--tile", type=int, default=64)
    parser.add_argument("--steps", type=int, default=5)
    parser.add_argument("--error-bound", type=float, default=1.0)
    args = parser.parse_args()

    eb = mdb.error_bound("relative", args.error_bound) if args.error_bound > 0 else mdb.error_bound("lossless")
    timestamps = np.arange(args.points, dtype=np.int64) * 1000
    offsets = np.arange(0, args.points + 65536, 65536, dtype=np.uint64)
    offsets[-1] = args.points
    if offsets[-2] == offsets[-1]:
        offsets = offsets[:-1]
    t0 = time.perf_counter()
    parts = []
    for s in range(args.distinct):
        _, values = datagen.sine_series(s, args.points)
        parts.append(ora.compress_chunks(timestamps, values, offsets, eb, n_threads=8))
    base = mdb.SegmentBatch.concat(parts) if False else None
    print(f"oracle fit: {time.perf_counter() - t0:.2f}s", flush=True)

    def cat(name):
        return np.concatenate([getattr(p, name) for p in parts])

    # Concatenate without going through Python rows: rebase out-of-line offsets per part.
    views = {c: [] for c in ("timestamps", "values", "residuals")}
    buffers = {c: [] for c in ("timestamps", "values", "residuals")}
    for column in views:
        base_offset = 0
        for p in parts:
            col = getattr(p, column)
            v = col.views.copy()
            lengths = col.lengths()
            long_rows = lengths > 12
            if long_rows.any():
                off = v[long_rows, 12:16].copy().view(np.int32).reshape(-1) + base_offset
                v[long_rows, 12:16] = off.astype(np.int32).view(np.uint8).reshape(-1, 4)
            views[column].append(v)
            if col.buffers:
                buffers[column].append(col.buffers[0])
                base_offset += col.buffers[0].size
    cols = {c: mdb.BinaryViewColumn(np.concatenate(views[c]),
                                    [np.concatenate(buffers[c])] if buffers[c] else [])
            for c in views}
    one = mdb.SegmentBatch(cat("model_type_id"), cat("start_time"), cat("end_time"), cols["timestamps"],
                           cat("min_value"), cat("max_value"), cols["values"], cols["residuals"])
    reps = args.tile
    tiled = mdb.SegmentBatch(np.tile(one.model_type_id, reps), np.tile(one.start_time, reps),
                             np.tile(one.end_time, reps),
                             mdb.BinaryViewColumn(np.tile(one.timestamps.views, (reps, 1)), one.timestamps.buffers),
                             np.tile(one.min_value, reps), np.tile(one.max_value, reps),
                             mdb.BinaryViewColumn(np.tile(one.values.views, (reps, 1)), one.values.buffers),
                             mdb.BinaryViewColumn(np.tile(one.residuals.views, (reps, 1)), one.residuals.buffers))
    print(f"segments: {len(tiled)} types: {np.bincount(tiled.model_type_id, minlength=3)}", flush=True)

    context = mdb.Context(0)
    print(context.device_info(), flush=True)
    dev = context.upload_segments(tiled)
    total = context.grid_count_dev(dev)
    print(f"points: {total}  ({total / len(tiled):.1f} per segment)", flush=True)
    out_ts = context.dev_alloc(8 * total)
    out_val = context.dev_alloc(4 * total)
    context.grid_batch_dev(dev, out_ts, out_val, total)
    context.profile_enable(True)
    context.profile_reset()
    context.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n, metrics = context.grid_batch_dev(dev, out_ts, out_val, total)
    context.sync()
    wall = (time.perf_counter() - t0) / args.steps
    print(f"wall per step {wall * 1e3:.3f} ms -> {total / wall / 1e9:.2f} Gpoints/s, "
          f"{12 * total / wall / 1e9:.1f} GB/s of output", flush=True)
    for name, (launches, ms) in sorted(context.profile().items()):
        per = ms / launches
        extra = ""
        if name == "k_grid_tiles":
            extra = f"  {(12 * total + 73 * len(tiled)) / per / 1e6:.1f} GB/s algorithmic"
        print(f"  {name:18s} {per:9.4f} ms x{launches}{extra}")
    print(metrics)
    # spot check against the oracle on the first part
    ts = context.download_array(out_ts, 100_000, np.int64)
    val = context.download_array(out_val, 100_000, np.float32)
    exp = ora.grid_batch(parts[0])
    assert np.array_equal(ts, exp[0][:100_000]) and np.array_equal(val.view(np.uint32), exp[1][:100_000].view(np.uint32))
    print("spot check vs oracle ok")


if __name__ == "__main__":
    main()
