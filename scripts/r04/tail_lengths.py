#!/usr/bin/env python3
"""Development tool: the residual tails of the error-bound sweep's short-segment rows (how many values, how long)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import modelardb_rs_amd as mdb  # noqa: E402

n = 1 << 24
rng = np.random.default_rng(3)
i = np.arange(n, dtype=np.float64)
values = (100.0 + 10.0 * np.sin(i / 2000.0) + rng.uniform(-0.5, 0.5, n)).astype(np.float32)
timestamps = 1000 * np.arange(n, dtype=np.int64)
offsets = np.arange(0, n + 65536, 65536, dtype=np.uint64)
ctx = mdb.Context(0)
for percent in (0.5, 0.3):
    segments = ctx.compress_chunks(timestamps, values, offsets, mdb.error_bound("relative", percent))
    lengths = segments.residuals.lengths()
    residuals = segments.residuals.to_bytes_list()
    counts = np.array([r[-1] if len(r) else 0 for r in residuals])
    with_tail = counts > 0
    print(f"{percent} %: {len(segments)} segments, {with_tail.mean():.2f} with a tail, {counts.sum() / n:.3f} of the points in tails, "
          f"tail values mean {counts[with_tail].mean():.1f} median {np.median(counts[with_tail])} p90 {np.percentile(counts[with_tail], 90)} "
          f"p99 {np.percentile(counts[with_tail], 99)} max {counts.max()}; bytes mean {lengths[with_tail].mean():.1f}, inline {np.mean(lengths[with_tail] <= 12):.2f}; "
          f"sum per 256 segments: mean {counts[: len(counts) // 256 * 256].reshape(-1, 256).sum(axis=1).mean():.0f} "
          f"over 3072: {(counts[: len(counts) // 256 * 256].reshape(-1, 256).sum(axis=1) > 3072).mean():.2f}")
