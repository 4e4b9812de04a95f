#!/usr/bin/env python3
"""GridStream over host batches, PCIe included, under the environment it is started in: the headline's series
(synthetic sensor values, relative 1 %: PMC-Mean and Swing with short tails; 100 series x 10^7 points) and the mixed
series (all three model types, lossless and 1 %; 256 series x 10^6 points). Best of 3 after a warm-up, with the plain
page-locked copy of the same bytes as the yardstick."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bench, datagen, modelardb_rs_amd as mdb
from modelardb_rs_amd import host

ctx = mdb.Context(0)
link = bench.link_rates(0)
label_env = " ".join(f"{k}={os.environ[k]}" for k in ("MDB_GRID_PIPELINE_CONTEXTS", "MDB_HOST_GRID_PREFETCH") if k in os.environ) or "defaults"


def chunked(values_dev, n_series, points, eb):
    starts = np.arange(0, points, 65536, dtype=np.uint64)
    offsets = np.concatenate([(np.arange(n_series, dtype=np.uint64)[:, None] * np.uint64(points) + starts[None, :]).reshape(-1),
                              np.array([n_series * points], dtype=np.uint64)])
    offsets_dev, first_dev = ctx.upload_array(offsets), ctx.upload_array(np.tile(starts, n_series))
    dev = ctx.compress_chunks_dev(0, values_dev, offsets_dev, len(offsets) - 1, eb, 0, bench.INTERVAL_US, first_dev)
    batch = dev.download()
    dev.free()
    for pointer in (offsets_dev, first_dev):
        ctx.dev_free(pointer)
    return batch


def measure(label, batch):
    host.measure_grid_stream(ctx, batch, 8192)
    best = min((host.measure_grid_stream(ctx, batch, 8192) for _ in range(3)), key=lambda r: r[1])
    rows, seconds, nbytes = best
    print(f"{label_env}: {label}: {len(batch)} segments, {rows} points: {seconds * 1e3:.1f} ms, {nbytes / seconds / 1e9:.1f} GB/s down = "
          f"{nbytes / seconds / 1e9 / link['d2h_GB_per_s']:.2f} of a plain copy ({link['d2h_GB_per_s']:.1f} GB/s)", flush=True)


series, points = 100, 10_000_000
values = ctx.dev_alloc(4 * series * points)
ctx.synth_values_dev(values, 0, series, points, bench.SEED)
measure("headline series", chunked(values, series, points, mdb.error_bound("relative", 1.0)))
ctx.dev_free(values)
points, distinct, copies = 1_000_000, 32, 8
host_values = np.tile(np.concatenate([datagen.mixed_series(points, 1000 + s, (1.0, 1.05) if s % 2 else None)[1] for s in range(distinct)]), copies)
values = ctx.upload_array(host_values)
for label, eb in (("mixed lossless", mdb.error_bound("lossless")), ("mixed 1 %", mdb.error_bound("relative", 1.0))):
    measure(label, chunked(values, distinct * copies, points, eb))
