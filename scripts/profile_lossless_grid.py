#!/usr/bin/env python3
"""grid() of lossless (MacaqueV) segments when there are too many streams for the parallel decoder:
one lane per stream in k_grid_serial. Development tool."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import modelardb_rs_amd as mdb  # noqa: E402
mdb._abi.RELOAD_OPTIONS_BEFORE_EVERY_CALL = True  # (this script changes MDB_* switches between calls: the library reads them once otherwise)

def main():
    ctx = mdb.Context(0)
    eb = mdb.error_bound("lossless")
    for series, points, chunk in ((20000, 20000, 2000), (2000, 200_000, 65536), (256, 65536, 65536)):
        total = series * points
        values = ctx.dev_alloc(4 * total)
        ctx.synth_values_dev(values, 0, series, points)
        cps = (points + chunk - 1) // chunk
        offsets = np.array([s * points + c * chunk for s in range(series) for c in range(cps)] + [total], dtype=np.uint64)
        off_dev = ctx.upload_array(offsets)
        dev = ctx.compress_chunks_dev(0, values, off_dev, len(offsets) - 1, eb, 0, 1000, 0)
        n = ctx.grid_count_dev(dev)
        out_ts, out_val = ctx.dev_alloc(8 * n), ctx.dev_alloc(4 * n)
        for setting in ("off", None):
            if setting is None:
                os.environ.pop("MDB_GRID_MV_MIN_VALUES", None)
            else:
                os.environ["MDB_GRID_MV_MIN_VALUES"] = setting
            ctx.grid_batch_dev(dev, out_ts, out_val, n)
            ctx.profile_enable(True); ctx.profile_reset(); ctx.sync(); t0 = time.perf_counter()
            for _ in range(3):
                ctx.grid_batch_dev(dev, out_ts, out_val, n)
            ctx.sync(); dt = (time.perf_counter() - t0) / 3
            kernels = {k: round(v[1] / v[0], 3) for k, v in ctx.profile().items() if v[1] / v[0] > 0.05}
            ctx.profile_enable(False)
            same = np.array_equal(ctx.download_array(out_val, min(n, 4_000_000), np.uint32),
                                  ctx.download_array(values, min(n, 4_000_000), np.uint32))
            print(f"{series} x {points} / {chunk}, parallel decoder {setting or 'default'}: {len(dev)} segments, grid {dt*1e3:.2f} ms "
                  f"({n/dt/1e9:.1f} Gvalues/s) lossless={same} {kernels}", flush=True)
        for pointer in (values, off_dev, out_ts, out_val):
            ctx.dev_free(pointer)
        dev.free()
main()
