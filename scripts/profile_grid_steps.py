#!/usr/bin/env python3
"""Does k_grid_tiles slow down over consecutive steps (power / clock management)? Fits the benchmark
workload on the GPU, then times every step of three bursts separated by idle pauses. Development tool."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import modelardb_rs_amd as mdb  # noqa: E402

def main():
    series, points, chunk = 1000, 10_000_000, 65536
    ctx = mdb.Context(0)
    eb = mdb.error_bound("relative", 1.0)
    total = series * points
    values = ctx.dev_alloc(4 * total)
    ctx.synth_values_dev(values, 0, series, points)
    cps = (points + chunk - 1) // chunk
    offsets = np.array([s * points + c * chunk for s in range(series) for c in range(cps)] + [total], dtype=np.uint64)
    first = np.array([c * chunk for s in range(series) for c in range(cps)], dtype=np.uint64)
    off_dev, first_dev = ctx.upload_array(offsets), ctx.upload_array(first)
    dev = ctx.compress_chunks_dev(0, values, off_dev, len(offsets) - 1, eb, 0, 1000, first_dev)
    ctx.dev_free(values)
    out_ts, out_val = ctx.dev_alloc(8 * total), ctx.dev_alloc(4 * total)
    ctx.profile_enable(True)
    for pause in (0.0, 3.0, 10.0, 0.0):
        time.sleep(pause)
        times = []
        for _ in range(10):
            ctx.profile_reset()
            ctx.grid_batch_dev(dev, out_ts, out_val, total)
            times.append(ctx.profile()["k_grid_tiles"][1])
        print(f"after {pause:4.1f} s idle: " + " ".join(f"{t:.2f}" for t in times), flush=True)
main()
