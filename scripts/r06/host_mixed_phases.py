#!/usr/bin/env python3
"""The mixed series as host batches through the pipelined grid: every entry of the library's profile (kernels and the
`host:` phases), per call."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import datagen, modelardb_rs_amd as mdb
from modelardb_rs_amd import host
points, distinct, copies = 1_000_000, 32, int(os.environ.get("COPIES", 8))
ctx = mdb.Context(0)
host_values = np.tile(np.concatenate([datagen.mixed_series(points, 1000 + s, (1.0, 1.05) if s % 2 else None)[1] for s in range(distinct)]), copies)
total = len(host_values)
values = ctx.upload_array(host_values)
starts = np.arange(0, points, 65536, dtype=np.uint64)
series = distinct * copies
offsets = np.concatenate([(s * points + starts) for s in range(series)] + [np.array([total], dtype=np.uint64)]).astype(np.uint64)
offsets_dev, first_dev = ctx.upload_array(offsets), ctx.upload_array(np.tile(starts, series))
for label, eb in (("lossless", mdb.error_bound("lossless")), ("relative 1 %", mdb.error_bound("relative", 1.0))):
    dev = ctx.compress_chunks_dev(0, values, offsets_dev, len(offsets) - 1, eb, 0, 100, first_dev)
    batch = dev.download(); dev.free()
    host.measure_grid_stream(ctx, batch, 8192)
    best = None
    for _ in range(4):
        ctx.profile_enable(True); ctx.profile_reset()
        rows, seconds, nbytes = host.measure_grid_stream(ctx, batch, 8192)
        entries = sorted(ctx.profile().items(), key=lambda item: -item[1][1])
        ctx.profile_enable(False)
        if best is None or seconds < best[0]: best = (seconds, entries, rows, nbytes)
    seconds, entries, rows, nbytes = best
    print(label, f"{len(batch)} segments, {rows} points: {seconds * 1e3:.1f} ms = {rows / seconds:.3g} values/s, {nbytes / seconds / 1e9:.1f} GB/s down", flush=True)
    print("   ", ", ".join(f"{name} {calls}x {ms:.1f} ms" for name, (calls, ms) in entries[:26]), flush=True)
