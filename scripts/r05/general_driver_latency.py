"""The general fit driver's fixed cost: calls of ONE 65 536-point chunk (MDB_FIT_SMALL=0: not the few-chunks path) and of
256 chunks, host to host; under `rocprofv3 --hip-trace --stats` the HIP calls behind it. usage: python general_driver_latency.py [calls]"""
import os, sys, time
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
os.environ["MDB_FIT_SMALL"] = "0"
import numpy as np
import modelardb_rs_amd as mdb, datagen
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ctx = mdb.Context(0)
eb = mdb.error_bound("relative", 1.0)
chunk = 65536
values = datagen.bench_series(3, 256 * chunk)
ts = np.arange(256 * chunk, dtype=np.int64) * 1000
for n in (1, 256):
    chunks = [(ts[k * chunk:(k + 1) * chunk], values[k * chunk:(k + 1) * chunk]) for k in range(n)]
    for _ in range(3):
        ctx.compress_chunk_list(chunks, eb)
    timings = []
    for _ in range(calls if n == 1 else max(3, calls // 10)):
        ctx.compress_chunk_list(chunks, eb)
        timings.append(ctx.last_call_seconds)
    print("%d chunks: min %.3f ms, median %.3f ms" % (n, 1e3 * min(timings), 1e3 * float(np.median(timings))), flush=True)
