#!/usr/bin/env python3
"""A call that is smooth (models of thousands of points) with rough windows in it (noise of the bound's size in a share of
the 1 024-point windows): the library's choice against split mode for every chunk (what the probe would send it to if it
looked at the pace alone) and against no probe. 4 096 chunks of 65 536 points, relative 2 %."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import modelardb_rs_amd as mdb
mdb._abi.RELOAD_OPTIONS_BEFORE_EVERY_CALL = True
ctx = mdb.Context(0)
chunks, points = 4096, 65536
n = chunks * points
rng = np.random.default_rng(3)
i = np.arange(n, dtype=np.float64)
smooth = 100.0 + 10.0 * np.sin(i / 20000.0)
eb = mdb.error_bound("relative", 2.0)
offsets_dev = ctx.upload_array(np.arange(0, n + points, points, dtype=np.uint64))
for share in (0.0, 0.1, 0.3, 0.6):
    rough = rng.uniform(size=n // 1024) < share
    values = (smooth + np.repeat(rough, 1024) * rng.uniform(-2.5, 2.5, n)).astype(np.float32)
    values_dev = ctx.upload_array(values)
    row = []
    for label, env in (("default", {}), ("no probe", {"MDB_FIT_WAVE_PROBE": "0"}), ("split mode, pieces of 1 024", {"MDB_FIT_PIECE_POINTS": "1024"}),
                       ("a wave per chunk to the end", {"MDB_FIT_WAVE": "1"}), ("windows of 8 192", {"MDB_FIT_WAVE_WINDOW_POINTS": "8192"})):
        for name in ("MDB_FIT_WAVE_PROBE", "MDB_FIT_PIECE_POINTS", "MDB_FIT_WAVE", "MDB_FIT_WAVE_WINDOW_POINTS"):
            os.environ.pop(name, None)
        os.environ.update(env)
        best, kernels = None, {}
        for repetition in range(3):
            ctx.profile_enable(True); ctx.profile_reset(); ctx.sync()
            started = time.perf_counter()
            dev = ctx.compress_chunks_dev(0, values_dev, offsets_dev, chunks, eb, 0, 1000, 0)
            ctx.sync()
            seconds = time.perf_counter() - started
            segments = len(dev); dev.free()
            if repetition and (best is None or seconds < best):
                best, kernels = seconds, {k: round(v[1], 2) for k, v in ctx.profile().items() if v[1] > 0.3}
            ctx.profile_enable(False)
        row.append(f"{label}: {best * 1e3:.1f} ms {kernels}")
    print(f"rough windows {share:.0%}, {segments} segments | " + " | ".join(row), flush=True)
    ctx.dev_free(values_dev)
