#!/usr/bin/env python3
"""One univariate series of L points from host memory: mdb_compress_series against one CPU thread of the port (where
patch 0003's routing belongs), as scripts/r04/fit_crossover.py, plus - for the lossless bound, where the series is ONE
MacaqueV stream - the same with the long-segment blocks switched off (MDB_FIT_GAP_LONG_MIN_VALUES=off: round 4's one
wave per stream) and the kernels' share of the call."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import modelardb_rs_amd as mdb  # noqa: E402
import datagen  # noqa: E402
import oracle_lib as ora  # noqa: E402

ctx = mdb.Context(0)
lib = mdb.load_hip_library()


def timed(ts, values, eb, repetitions=7):
    best = []
    for _ in range(repetitions):
        ctx.try_compress_univariate_time_series(ts, values, eb)
        best.append(ctx.last_call_seconds)
    return min(best)


for kind, bound in (("relative", 1.0), ("relative", 10.0), ("lossless", 0.0)):
    eb = mdb.error_bound(kind, bound) if kind != "lossless" else mdb.error_bound("lossless")
    for length in (1024, 4096, 8192, 16384, 32768, 65536, 262144, 1048576):
        ts = np.arange(length, dtype=np.int64) * 1000
        values = datagen.bench_series(5, length, 0x4D44425F52454631)
        expected, cpu = ora.compress_chunks_timed(ts, values, np.array([0, length], dtype=np.uint64), eb, 1, repetitions=5)
        got = ctx.try_compress_univariate_time_series(ts, values, eb)
        assert got.identical(expected)
        gpu = timed(ts, values, eb)
        line = f"{kind} {bound}: {length:8d} points, {len(expected):5d} segments: GPU {1e3 * gpu:.3f} ms, one CPU thread {1e3 * min(cpu):.3f} ms"
        if kind == "lossless" and length >= 8192:
            lib.mdb_set_option(b"MDB_FIT_GAP_LONG_MIN_VALUES", b"off")
            one_wave = timed(ts, values, eb)
            lib.mdb_set_option(b"MDB_FIT_GAP_LONG_MIN_VALUES", None)
            line += f" (one wave per stream: {1e3 * one_wave:.3f} ms)"
        print(line, flush=True)

# noise under a lossless and a tight absolute bound (every code opens a window; every value is stored anew)
rng = np.random.default_rng(1)
for name, eb in (("lossless", mdb.error_bound("lossless")), ("absolute 0.01", mdb.error_bound("absolute", 0.01))):
    for length in (65536, 1048576):
        ts = np.arange(length, dtype=np.int64) * 1000
        values = rng.uniform(100.0, 200.0, length).astype(np.float32)
        expected, cpu = ora.compress_chunks_timed(ts, values, np.array([0, length], dtype=np.uint64), eb, 1, repetitions=3)
        got = ctx.try_compress_univariate_time_series(ts, values, eb)
        assert got.identical(expected)
        gpu = timed(ts, values, eb)
        lib.mdb_set_option(b"MDB_FIT_GAP_LONG_MIN_VALUES", b"off")
        one_wave = timed(ts, values, eb)
        lib.mdb_set_option(b"MDB_FIT_GAP_LONG_MIN_VALUES", None)
        ctx.profile_enable(True)
        ctx.profile_reset()
        ctx.try_compress_univariate_time_series(ts, values, eb)
        kernels = ", ".join(f"{k} {v[1]:.3f}" for k, v in sorted(ctx.profile().items(), key=lambda kv: -kv[1][1]) if v[1] >= 0.005)
        ctx.profile_enable(False)
        print(f"noise, {name}: {length:8d} points: GPU {1e3 * gpu:.3f} ms (one wave per stream: {1e3 * one_wave:.3f} ms), "
              f"one CPU thread {1e3 * min(cpu):.3f} ms; kernels (ms): {kernels}", flush=True)
