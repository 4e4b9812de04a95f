"""Where a mdb_compress_chunk_list call of the bench's host-path size spends its time (MDB_FIT_DEBUG's line on stderr):
160 M points in 2 448 chunks of 65 536, every chunk with a timestamp array of its own. usage: python fit_host_phases.py [series]"""
import os, sys, time
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
if os.environ.get("NO_DEBUG") != "1": os.environ["MDB_FIT_DEBUG"] = "1"
import numpy as np
import modelardb_rs_amd as mdb, datagen
n_series = int(sys.argv[1]) if len(sys.argv) > 1 else 16
points, chunk = 10_000_000, 65536
ctx = mdb.Context(0)
eb = mdb.error_bound("relative", 1.0)
values = np.concatenate([datagen.bench_series(s, points) for s in range(n_series)])
ts = np.tile(np.arange(points, dtype=np.int64) * 1000, n_series)
chunks = []
for s in range(n_series):
    for a in range(0, points, chunk):
        b = min(a + chunk, points)
        chunks.append((ts[s * points + a: s * points + b], values[s * points + a: s * points + b]))
for _ in range(4):
    started = time.perf_counter()
    got = ctx.compress_chunk_list(chunks, eb)
    print("call %.2f ms, library %.2f ms, %d segments, %.3g points/s" % (1e3 * (time.perf_counter() - started), 1e3 * ctx.last_call_seconds,
                                                                     len(got), len(values) / ctx.last_call_seconds), flush=True)
