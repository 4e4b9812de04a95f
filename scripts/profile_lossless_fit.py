import os, sys, time
import numpy as np
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import datagen, modelardb_rs_amd as mdb
ctx = mdb.Context(0)
ts, v = datagen.sine_series(9, 1_000_000)
eb = mdb.error_bound("lossless")
for label, offs in (("one call", np.array([0, 1_000_000], dtype=np.uint64)),
                    ("16 chunks", np.append(np.arange(0, 1_000_000, 65536), 1_000_000).astype(np.uint64))):
    ctx.compress_chunks(ts, v, offs, eb)
    ctx.profile_enable(True); ctx.profile_reset()
    t0 = time.perf_counter(); got = ctx.compress_chunks(ts, v, offs, eb); dt = time.perf_counter() - t0
    print(label, f"{dt*1e3:.1f} ms", len(got), {k: round(x[1], 2) for k, x in ctx.profile().items() if x[1] > 0.05})
    ctx.profile_enable(False)
