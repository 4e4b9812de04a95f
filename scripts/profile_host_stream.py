#!/usr/bin/env python3
"""The drop-in path (C++ GridStream over host segment batches) on 10^9 points of the bench's workload,
alone, so that rocprofv3 --kernel-trace --memory-copy-trace --stats of this script shows how long the copies
into page-locked memory take by themselves. Development tool.
Usage: python3 scripts/profile_host_stream.py [batch_size ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import modelardb_rs_amd as mdb  # noqa: E402
from modelardb_rs_amd import host  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [8192, 65536]
    series, points, chunk = 100, 10_000_000, 65536
    ctx = mdb.Context(0)
    total = series * points
    values = ctx.dev_alloc(4 * total)
    ctx.synth_values_dev(values, 0, series, points, 20260101)
    starts = np.arange(0, points, chunk, dtype=np.uint64)
    offsets = (np.arange(series, dtype=np.uint64)[:, None] * np.uint64(points) + starts[None, :]).reshape(-1)
    offsets = np.concatenate([offsets, np.array([total], dtype=np.uint64)])
    first_index = np.tile(starts, series)
    dev = ctx.compress_chunks_dev(0, values, ctx.upload_array(offsets), len(offsets) - 1,
                                  mdb.error_bound("relative", 1.0), 0, 1000, ctx.upload_array(first_index))
    ctx.dev_free(values)
    sample = dev.download()
    dev.free()
    host.measure_grid_stream(ctx, sample.slice(0, 65536), 8192)
    for tags in (None, {"tag": "wind-turbine-0042"}, {"tag": "wind-turbine-0042", "park": "north-sea-7", "country": "dk"}):
      for batch_size in sizes:
        for _ in range(2):
            n, seconds, down = host.measure_grid_stream(ctx, sample, batch_size, tags=tags)
            print(f"{len(tags or ())} tag columns,", end=" ")
            print(f"batch_size {batch_size}: {n} points, {len(sample)} segments in {seconds * 1e3:.1f} ms: "
                  f"{n / seconds / 1e9:.2f} Gvalues/s, {down / seconds / 1e9:.1f} GB/s into page-locked memory", flush=True)


main()
