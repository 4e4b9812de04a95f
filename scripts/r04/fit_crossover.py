#!/usr/bin/env python3
"""One univariate series of L points from host memory: mdb_compress_series (the path of a handful of chunks) against one
CPU thread of the port - where patch 0003's HIP_MINIMUM_UNIVARIATE_LENGTH belongs."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import modelardb_rs_amd as mdb  # noqa: E402
import datagen  # noqa: E402
import oracle_lib as ora  # noqa: E402

ctx = mdb.Context(0)
for kind, bound in (("relative", 1.0), ("relative", 10.0), ("lossless", 0.0)):
    eb = mdb.error_bound(kind, bound) if kind != "lossless" else mdb.error_bound("lossless")
    for length in (1024, 4096, 8192, 16384, 32768, 65536, 262144, 1048576):
        ts = np.arange(length, dtype=np.int64) * 1000
        values = datagen.bench_series(5, length, 0x4D44425F52454631)
        expected, cpu = ora.compress_chunks_timed(ts, values, np.array([0, length], dtype=np.uint64), eb, 1, repetitions=5)
        got = ctx.try_compress_univariate_time_series(ts, values, eb)
        assert got.identical(expected)
        gpu = []
        for _ in range(7):
            ctx.try_compress_univariate_time_series(ts, values, eb)
            gpu.append(ctx.last_call_seconds)
        print(f"{kind} {bound}: {length:8d} points, {len(expected):5d} segments: GPU {1e3 * min(gpu):.3f} ms, one CPU thread {1e3 * min(cpu):.3f} ms")
