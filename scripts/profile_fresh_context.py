#!/usr/bin/env python3
"""What a context made per query costs: a 2 M-point GridStream on a fresh context, three times in one process.
Development tool (run it under rocprofv3 --hip-runtime-trace --stats to see where the time goes)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, modelardb_rs_amd as mdb, oracle_lib as ora, cases  # noqa: E402
from modelardb_rs_amd import host, api  # noqa: E402

ts, v = cases.synthetic_series(2_000_000, False, (1.0, 1.05), 3)
batch = ora.try_compress_univariate_time_series(ts, v, cases.error_bounds()["rel1"])
for k in range(4):
    t0 = time.perf_counter()
    ctx = api.Context(0)
    t1 = time.perf_counter()
    n, s, b = host.measure_grid_stream(ctx, batch, 8192)
    t2 = time.perf_counter()
    ctx.close()
    t3 = time.perf_counter()
    print(f"fresh context {k}: init {1e3 * (t1 - t0):.2f} ms, stream of {n} points {1e3 * (t2 - t1):.2f} ms, close {1e3 * (t3 - t2):.2f} ms")
ctx = api.Context(0)
for k in range(4):
    t1 = time.perf_counter()
    n, s, b = host.measure_grid_stream(ctx, batch, 8192)
    t2 = time.perf_counter()
    print(f"kept context, new GridStream {k}: stream of {n} points {1e3 * (t2 - t1):.2f} ms")
ctx.close()
