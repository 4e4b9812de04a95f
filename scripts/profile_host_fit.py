import os, sys, time
import numpy as np
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import modelardb_rs_amd as mdb
ctx = mdb.Context(0)
eb = mdb.error_bound("relative", 1.0)
series, points, chunk = 20, 10_000_000, 65536
total = series*points
dev = ctx.dev_alloc(4*total); ctx.synth_values_dev(dev, 0, series, points)
values = ctx.download_array(dev, total, np.float32); ctx.dev_free(dev)
ts = np.tile(np.arange(points, dtype=np.int64)*1000, series)
offsets = np.array([s*points + c for s in range(series) for c in range(0, points, chunk)] + [total], dtype=np.uint64)
for rep in range(3):
    ctx.profile_enable(True); ctx.profile_reset()
    t0=time.perf_counter(); got = ctx.compress_chunks(ts, values, offsets, eb); dt=time.perf_counter()-t0
    k={n:round(v[1],2) for n,v in ctx.profile().items() if v[1]>0.5}
    print(f"host fit {total/1e6:.0f} M points: {dt*1e3:.1f} ms = {total/dt/1e9:.2f} Gpts/s ({12*total/dt/1e9:.1f} GB/s of input) {len(got)} segments; kernels {k}", flush=True)
