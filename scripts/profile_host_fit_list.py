#!/usr/bin/env python3
"""Where the time of mdb_compress_chunk_list goes (MDB_FIT_DEBUG=1 prints the library's own phases), next to
mdb_compress_chunks over the same points as one pair of arrays: 16 of the benchmark's series in host memory."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import datagen  # noqa: E402
import modelardb_rs_amd as mdb  # noqa: E402

series, points, chunk = int(os.environ.get("SERIES", 16)), 10_000_000, 65536
context = mdb.Context(0)
eb = mdb.error_bound("relative", 1.0)
dev = context.dev_alloc(4 * series * points)
context.synth_values_dev(dev, 0, series, points)
values = context.download_array(dev, series * points, np.float32)
ts = np.tile(np.arange(points, dtype=np.int64) * 1000, series)
offsets = np.array([s * points + c for s in range(series) for c in range(0, points, chunk)] + [series * points], dtype=np.uint64)
chunks = [(ts[int(a):int(b)], values[int(a):int(b)]) for a, b in zip(offsets[:-1], offsets[1:])]
for _ in range(3):
    context.compress_chunk_list(chunks, eb)
    listed = context.last_call_seconds
    context.compress_chunks(ts, values, offsets, eb)
    flat = context.last_call_seconds
    print(f"chunk list {1e3 * listed:.1f} ms = {series * points / listed:.3g} points/s; one pair of arrays {1e3 * flat:.1f} ms", flush=True)
