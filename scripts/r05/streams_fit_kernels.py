#!/usr/bin/env python3
"""Where the lossless fit of S streams of 50 000 values (every chunk ONE MacaqueV segment) spends its time: the
kernels of the call with the long segments cut into blocks (default) and with one wave per stream."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import modelardb_rs_amd as mdb  # noqa: E402

ctx = mdb.Context(0)
lib = mdb.load_hip_library()
eb = mdb.error_bound("lossless")
per_stream = 50_000
for streams in (1000, 10_000):
    n = streams * per_stream
    rng = np.random.default_rng(1)
    i = np.arange(n, dtype=np.float64)
    values = (100.0 + 10.0 * np.sin(i / 2000.0) + rng.uniform(-0.5, 0.5, n)).astype(np.float32)
    values_dev = ctx.upload_array(values)
    offsets = np.arange(0, n + per_stream, per_stream, dtype=np.uint64)
    offsets_dev = ctx.upload_array(offsets)
    for mode in ("blocks", "one wave per stream"):
        lib.mdb_set_option(b"MDB_FIT_GAP_LONG_MIN_VALUES", None if mode == "blocks" else b"off")
        best, kernels = 1e9, None
        for repetition in range(4):
            ctx.profile_enable(repetition == 3)
            ctx.profile_reset()
            ctx.sync(); started = time.perf_counter()
            dev = ctx.compress_chunks_dev(0, values_dev, offsets_dev, streams, eb, 0, 1000, 0)
            ctx.sync(); best = min(best, time.perf_counter() - started)
            if repetition == 3:
                kernels = ", ".join(f"{k} {v[1]:.3f}" for k, v in sorted(ctx.profile().items(), key=lambda kv: -kv[1][1]) if v[1] >= 0.01)
            dev.free()
        ctx.profile_enable(False)
        print(f"{streams} streams, {mode}: {1e3 * best:.2f} ms; kernels (ms, profiled run): {kernels}", flush=True)
    lib.mdb_set_option(b"MDB_FIT_GAP_LONG_MIN_VALUES", None)
    ctx.dev_free(values_dev); ctx.dev_free(offsets_dev)
