#!/usr/bin/env python3
"""Per-segment comparison of the aggregates of one soak case (debug tool)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import modelardb_rs_amd as mdb, oracle_lib as ora, test_gpu_soak as s  # noqa: E402

index = int(sys.argv[1])
rng = np.random.default_rng([0x50414B, index])
n = int(rng.choice([1, 2, 7, 8, 60, 700, 5000, 20_000, 70_000]))
n = max(1, int(n * rng.uniform(0.5, 1.0)))
ts, v, eb = s.random_timestamps(rng, n), s.random_values(rng, n), s.random_error_bound(rng)
n_chunks = int(rng.choice([1, 1, 2, 5]))
cuts = np.sort(rng.integers(0, n + 1, n_chunks - 1)) if n_chunks > 1 else np.zeros(0, dtype=np.int64)
offsets = np.concatenate([[0], cuts, [n]]).astype(np.uint64)
batch = ora.compress_chunks(ts, v, offsets, eb)
print("n", n, "eb", eb.kind, eb.value, "segments", len(batch))
ctx = mdb.Context(0)
bad = 0
for i in range(len(batch)):
    one = batch.slice(i, i + 1)
    g, e = ctx.agg_batch(one, s.ALL), ora.agg_batch(one, s.ALL)
    if g.sum != e.sum and not (np.isnan(g.sum) and np.isnan(e.sum)):
        bad += 1
        if bad <= 10:
            row = one.rows()[0]
            print(i, "type", row[0], "count", e.count, "gpu", repr(g.sum), "ora", repr(e.sum), "min/max", row[4], row[5],
                  "values", row[6].hex(), "residuals", len(row[7]))
print("segments with different sums:", bad)
g, e = ctx.agg_batch(batch, s.ALL), ora.agg_batch(batch, s.ALL)
print("whole:", repr(g.sum), repr(e.sum), g.count, e.count)
a, b = sorted(int(x) for x in rng.integers(0, n, 2))
t_lo = int(ts[a]) - int(rng.integers(0, 2))
t_hi = int(ts[b]) + int(rng.integers(0, 2))
g, e = ctx.agg_batch_range(batch, t_lo, t_hi, s.ALL), ora.agg_batch_range(batch, t_lo, t_hi, s.ALL)
print("range:", t_lo, t_hi, repr(g.sum), repr(e.sum), g.count, e.count)
for i in range(len(batch)):
    one = batch.slice(i, i + 1)
    g, e = ctx.agg_batch_range(one, t_lo, t_hi, s.ALL), ora.agg_batch_range(one, t_lo, t_hi, s.ALL)
    if g.sum != e.sum and not (np.isnan(g.sum) and np.isnan(e.sum)):
        row = one.rows()[0]
        print(" segment", i, "type", row[0], "start/end", row[1], row[2], "count", g.count, e.count, "gpu", repr(g.sum), "ora", repr(e.sum),
              "min/max", row[4], row[5], "values", row[6].hex(), "ts", row[3].hex(), "residuals", len(row[7]))
