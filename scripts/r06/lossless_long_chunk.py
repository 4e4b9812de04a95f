#!/usr/bin/env python3
"""One series handed over as ONE chunk under a lossless bound (the embedded API's call): how long a chunk the wave
kernel should keep (MDB_FIT_WAVE_MAX_CHUNK_POINTS) now that a lossless wave walks a chunk by comparisons."""
import os, sys, time, statistics
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import modelardb_rs_amd as mdb, datagen
ctx = mdb.Context(0)
eb = mdb.error_bound("lossless")
for label, maker in (("sine+noise (all MacaqueV)", lambda n: datagen.sine_series(9, n)), ("mixed recipe", lambda n: datagen.mixed_series(n, 1000, None, interval=1000))):
    for n in (262_144, 1_000_000, 4_000_000):
        ts, v = maker(n)
        offs = np.array([0, n], dtype=np.uint64)
        ctx.compress_chunks(ts, v, offs, eb)
        ctx.profile_enable(True); ctx.profile_reset()
        seconds = []
        for _ in range(3):
            t0 = time.perf_counter(); got = ctx.compress_chunks(ts, v, offs, eb); seconds.append(time.perf_counter() - t0)
        kernels = {k: round(x[1] / 3, 2) for k, x in ctx.profile().items() if x[1] / 3 > 0.05}
        ctx.profile_enable(False)
        print(f"{label}, {n} points, MAX_CHUNK_POINTS={os.environ.get('MDB_FIT_WAVE_MAX_CHUNK_POINTS', 'default')}: {1e3 * statistics.median(seconds):.2f} ms host to host, {len(got)} segments {kernels}", flush=True)
