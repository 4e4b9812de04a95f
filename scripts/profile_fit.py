#!/usr/bin/env python3
"""Time the fit kernels on device-generated sine series (development tool)."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import modelardb_rs_amd as mdb  # noqa: E402

def main():
    p = argparse.ArgumentParser()
    p.add_argument("--series", type=int, default=200)
    p.add_argument("--points", type=int, default=10_000_000)
    p.add_argument("--chunk", type=int, default=65536)
    p.add_argument("--error-bound", type=float, default=1.0)
    p.add_argument("--materialise-ts", action="store_true")
    p.add_argument("--irregular", action="store_true", help="materialised timestamps with random gaps")
    p.add_argument("--gaps", action="store_true", help="materialised timestamps: a fixed interval, 1 %% of the samples missing")
    a = p.parse_args()
    ctx = mdb.Context(0)
    eb = mdb.error_bound("relative", a.error_bound) if a.error_bound > 0 else mdb.error_bound("lossless")
    total = a.series * a.points
    values = ctx.dev_alloc(4 * total)
    ctx.synth_values_dev(values, 0, a.series, a.points)
    cps = (a.points + a.chunk - 1) // a.chunk
    offsets = np.array([s * a.points + c * a.chunk for s in range(a.series) for c in range(cps)] + [total], dtype=np.uint64)
    first = np.array([c * a.chunk for s in range(a.series) for c in range(cps)], dtype=np.uint64)
    off_dev, first_dev = ctx.upload_array(offsets), ctx.upload_array(first)
    ts_dev = 0
    if a.irregular:
        rng = np.random.default_rng(5)
        one = np.cumsum(rng.integers(900, 1100, a.points).astype(np.int64))
        ts_dev = ctx.upload_array(np.tile(one, a.series))
    elif a.gaps:
        rng = np.random.default_rng(5)
        one = np.cumsum(np.where(rng.random(a.points) < 0.01, 2000, 1000).astype(np.int64))
        ts_dev = ctx.upload_array(np.tile(one, a.series))
    elif a.materialise_ts:
        ts_dev = ctx.upload_array(np.tile(np.arange(a.points, dtype=np.int64) * 1000, a.series))
    for rep in range(2):
        ctx.profile_enable(True); ctx.profile_reset(); ctx.sync()
        t0 = time.perf_counter()
        dev = ctx.compress_chunks_dev(ts_dev, values, off_dev, len(offsets) - 1, eb, 0, 1000, first_dev)
        ctx.sync(); dt = time.perf_counter() - t0
        print(f"rep {rep}: {dt*1e3:.2f} ms  {total/dt/1e9:.2f} Gpts/s  {len(dev)} segments ({total/len(dev):.0f} pts/seg), {len(offsets)-1} chunks")
        for name, (n, ms) in sorted(ctx.profile().items()):
            print(f"   {name:16s} {ms/n:9.3f} ms x{n}")
        dev.free()
main()
