#!/usr/bin/env python3
"""BASELINE configs[0]: 1 univariate series, 10^6 regular points, lossless: compress (16 chunks of 65 536 as the
server's buffers, and the whole series as one chunk as the embedded API hands it over), grid and SUM through the
host entry points (PCIe included) and on the resident segments; best of 5, milliseconds."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import datagen
import modelardb_rs_amd as mdb

ctx = mdb.Context(0)
n = 1_000_000
ts, values = datagen.sine_series(0, n)
mask = mdb.MDB_AGG_COUNT | mdb.MDB_AGG_MIN | mdb.MDB_AGG_MAX | mdb.MDB_AGG_SUM
def best(call, repetitions=5):
    call()
    times = []
    for _ in range(repetitions):
        started = time.perf_counter(); result = call(); times.append(time.perf_counter() - started)
    return 1e3 * min(times), result
for label, eb in (("lossless", mdb.error_bound("lossless")), ("relative 1 %", mdb.error_bound("relative", 1.0))):
    offsets = np.arange(0, n + 65536, 65536, dtype=np.uint64); offsets[-1] = n
    fit_chunks, segments = best(lambda: ctx.compress_chunks(ts, values, offsets, eb))
    fit_one, whole = best(lambda: ctx.try_compress_univariate_time_series(ts, values, eb))
    grid_ms, _ = best(lambda: ctx.grid_batch(segments))
    sum_ms, _ = best(lambda: ctx.agg_batch(segments, mask))
    resident = ctx.upload_segments(segments)
    grid_resident, _ = best(lambda: ctx.grid_resident(resident))
    sum_resident, _ = best(lambda: ctx.agg_batch_dev(resident, mask))
    resident.free()
    print(f"{label}: {len(segments)} segments ({len(whole)} as one chunk); fit {fit_chunks:.2f} ms in 16 chunks, {fit_one:.2f} ms as one; "
          f"grid {grid_ms:.2f} ms from host segments, {grid_resident:.2f} ms resident (+ download); SUM {sum_ms:.2f} / {sum_resident:.2f} ms", flush=True)
