#!/usr/bin/env python3
"""How the CPU baseline (the oracle's threaded grid and fit) scales with its thread count on this box, and what
the box allows: CPUs visible, affinity mask, cgroup CPU quota. bench.py picks its thread count from this."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import datagen  # noqa: E402
import modelardb_rs_amd as mdb  # noqa: E402
import oracle_lib as ora  # noqa: E402

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(path):
        print(path, open(path).read().strip())
eb = mdb.error_bound("relative", 1.0)
series, n = int(os.environ.get("SERIES", 16)), 4_000_000
ts = np.tile(np.arange(n, dtype=np.int64) * 1000, series)
values = np.concatenate([datagen.bench_series(s, n) for s in range(series)])
offsets = np.array([s * n + c for s in range(series) for c in range(0, n, 65536)] + [series * n], dtype=np.uint64)
for pin in (True, False):
    for threads in (1, 2, 4, 8, 16, 32, 64, 128, 256):
        if threads > 2 * (os.cpu_count() or 1):
            break
        fitted, fit_seconds = ora.compress_chunks_timed(ts, values, offsets, eb, threads, repetitions=3, pin=pin)
        t, v, grid_seconds = ora.grid_batch_timed(fitted, threads, repetitions=3, pin=pin)
        print(f"pin={pin} threads={threads:4d} fit {series * n / np.median(fit_seconds) / 1e6:9.1f} Mpoints/s   "
              f"grid {len(t) / np.median(grid_seconds) / 1e6:9.1f} Mvalues/s", flush=True)
