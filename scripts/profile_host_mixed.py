#!/usr/bin/env python3
"""The pipelined grid of the boundary over host batches of the reference's acceptance series (all three model
types): values/s and the kernels behind it (one context: MDB_GRID_PIPELINE_CONTEXTS=1 is set here)."""
import os
import sys
import time

import numpy as np

os.environ.setdefault("MDB_GRID_PIPELINE_CONTEXTS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import datagen  # noqa: E402
import modelardb_rs_amd as mdb  # noqa: E402
from modelardb_rs_amd import host  # noqa: E402

points, distinct, copies = 1_000_000, 32, int(os.environ.get("COPIES", 4))
ctx = mdb.Context(0)
host_values = np.concatenate([datagen.mixed_series(points, 1000 + s, (1.0, 1.05) if s % 2 else None)[1] for s in range(distinct)])
host_values = np.tile(host_values, copies)
total = len(host_values)
values = ctx.upload_array(host_values)
starts = np.arange(0, points, 65536, dtype=np.uint64)
series = distinct * copies
offsets = np.concatenate([(s * points + starts) for s in range(series)] + [np.array([total], dtype=np.uint64)]).astype(np.uint64)
first_index = np.tile(starts, series)
offsets_dev, first_dev = ctx.upload_array(offsets), ctx.upload_array(first_index)
for label, eb in (("lossless", mdb.error_bound("lossless")), ("relative 1 %", mdb.error_bound("relative", 1.0))):
    dev = ctx.compress_chunks_dev(0, values, offsets_dev, len(offsets) - 1, eb, 0, 100, first_dev)
    batch = dev.download()
    dev.free()
    host.measure_grid_stream(ctx, batch, 8192)
    ctx.profile_enable(True); ctx.profile_reset()
    rows, seconds, _ = host.measure_grid_stream(ctx, batch, 8192)
    kernels = sorted(ctx.profile().items(), key=lambda item: -item[1][1])[:8]
    ctx.profile_enable(False)
    print(label, f"{len(batch)} segments, {rows} points: {seconds * 1e3:.1f} ms = {rows / seconds:.3g} values/s", flush=True)
    print("   ", ", ".join(f"{name} {calls}x {ms:.1f} ms" for name, (calls, ms) in kernels), flush=True)
