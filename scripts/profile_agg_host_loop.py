import sys, time
sys.path[:0]=["/root/repo","/root/repo/tests"]
import numpy as np, modelardb_rs_amd as mdb, oracle_lib as ora, datagen
ctx = mdb.Context(0)
eb = mdb.error_bound("relative", 1.0)
n = 6_000_000
ts = np.arange(n, dtype=np.int64) * 1000
offs = np.arange(0, n + 65536, 65536, dtype=np.uint64); offs[-1] = n
batch = ora.compress_chunks(ts, datagen.sine_series(1, n)[1], offs, eb, n_threads=8).slice(0, 8192)
mask = mdb.MDB_AGG_COUNT | mdb.MDB_AGG_MIN | mdb.MDB_AGG_MAX | mdb.MDB_AGG_SUM
for _ in range(20): ctx.agg_batch(batch, mask)
t0=time.perf_counter()
for _ in range(300): ctx.agg_batch(batch, mask)
print("per call", (time.perf_counter()-t0)/300*1e6, "us", len(batch))
