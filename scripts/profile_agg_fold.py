#!/usr/bin/env python3
"""One fold of the patched accumulators: mdb_agg_batch over 262 144 host segments of the mixed series (bench.py's
mixed_models recipe), SUM - the call's milliseconds and the kernels behind it, under a lossless bound and under 1 %.
Usage (on the GPU box): python3 scripts/profile_agg_fold.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import modelardb_rs_amd as mdb  # noqa: E402
import datagen  # noqa: E402

ctx = mdb.Context(0)
points, series = 1_000_000, 48
host_values = np.concatenate([datagen.mixed_series(points, 1000 + s, (1.0, 1.05) if s % 2 else None)[1] for s in range(series)])
values = ctx.upload_array(host_values)
starts = np.arange(0, points, 65536, dtype=np.uint64)
offsets = np.concatenate([s * points + starts for s in range(series)] + [np.array([series * points], dtype=np.uint64)]).astype(np.uint64)
first_index = np.tile(starts, series)
offsets_dev, first_index_dev = ctx.upload_array(offsets), ctx.upload_array(first_index)
for label, eb in (("lossless", mdb.error_bound("lossless")), ("relative 1 %", mdb.error_bound("relative", 1.0))):
    fitted = ctx.compress_chunks_dev(0, values, offsets_dev, len(offsets) - 1, eb, 0, 100, first_index_dev)
    batch = fitted.download()
    fitted.free()
    batch = batch.slice(0, min(len(batch), 262144))
    n_points = int(ctx.grid_count(batch))
    for mask_name, mask in (("SUM", mdb.MDB_AGG_SUM), ("COUNT", mdb.MDB_AGG_COUNT)):
        ctx.agg_batch(batch, mask)
        ctx.profile_enable(True); ctx.profile_reset()
        started = time.perf_counter()
        for _ in range(3):
            ctx.agg_batch(batch, mask)
        ms = (time.perf_counter() - started) / 3 * 1e3
        kernels = {k: round(v[1] / v[0], 3) for k, v in ctx.profile().items() if v[1] / v[0] > 0.02}
        ctx.profile_enable(False)
        print(f"{label:13s} {mask_name:5s} {len(batch)} segments, {n_points} points: {ms:.2f} ms per call; kernels {kernels}", flush=True)
ctx.close()
