#!/usr/bin/env python3
"""SUM over lossless (MacaqueV) segments on the device: the serial value decoder of
mdb_segment_dev.hpp is the whole cost. Development tool."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import modelardb_rs_amd as mdb  # noqa: E402
mdb._abi.RELOAD_OPTIONS_BEFORE_EVERY_CALL = True  # (this script changes MDB_* switches between calls: the library reads them once otherwise)

def config_1(ctx, eb):
    """BASELINE configs[0] as an aggregate query: one series, 16 streams of 65 536 values."""
    import datagen
    n = 1_000_000
    ts, v = datagen.sine_series(0, n)
    offsets = np.arange(0, n + 65536, 65536, dtype=np.uint64)
    offsets[-1] = n
    dev = ctx.upload_segments(ctx.compress_chunks(ts, v, offsets, eb))
    for setting in ("off", None):
        if setting is None:
            os.environ.pop("MDB_GRID_MV_MIN_VALUES", None)
        else:
            os.environ["MDB_GRID_MV_MIN_VALUES"] = setting
        ctx.agg_batch_dev(dev, 15)
        ctx.profile_enable(True); ctx.profile_reset(); ctx.sync(); t0 = time.perf_counter()
        for _ in range(5):
            state = ctx.agg_batch_dev(dev, 15)
        ctx.sync(); dt = (time.perf_counter() - t0) / 5
        kernels = {k: round(x[1] / x[0], 3) for k, x in ctx.profile().items() if x[1] / x[0] > 0.05}
        ctx.profile_enable(False)
        print(f"config 1 SUM, parallel decoder {setting or 'default'}: {dt*1e3:.2f} ms, sum {state.sum!r} {kernels}", flush=True)
    dev.free()


def main():
    ctx = mdb.Context(0)
    eb = mdb.error_bound("lossless")
    config_1(ctx, eb)
    for series, points, chunk in ((20000, 20000, 2000), (2000, 200_000, 65536)):
        total = series * points
        values = ctx.dev_alloc(4 * total)
        ctx.synth_values_dev(values, 0, series, points)
        cps = (points + chunk - 1) // chunk
        offsets = np.array([s * points + c * chunk for s in range(series) for c in range(cps)] + [total], dtype=np.uint64)
        off_dev = ctx.upload_array(offsets)
        dev = ctx.compress_chunks_dev(0, values, off_dev, len(offsets) - 1, eb, 0, 1000, 0)
        mask = mdb.MDB_AGG_SUM | mdb.MDB_AGG_COUNT
        state = ctx.agg_batch_dev(dev, mask)
        ctx.profile_enable(True); ctx.profile_reset(); ctx.sync(); t0 = time.perf_counter()
        for _ in range(3):
            state = ctx.agg_batch_dev(dev, mask)
        ctx.sync(); whole = (time.perf_counter() - t0) / 3
        kernels = {k: round(v[1] / v[0], 3) for k, v in ctx.profile().items() if v[1] / v[0] > 0.05}
        ctx.profile_enable(False)
        lo, hi = chunk // 3 * 1000, chunk // 3 * 2000
        ctx.agg_batch_range_dev(dev, lo, hi, mask)
        ctx.sync(); t0 = time.perf_counter()
        ranged = ctx.agg_batch_range_dev(dev, lo, hi, mask)
        ctx.sync(); part = time.perf_counter() - t0
        print(f"{series} x {points} / {chunk}: {len(dev)} segments, SUM {whole*1e3:.2f} ms ({total/whole/1e9:.1f} Gpts/s) "
              f"range SUM {part*1e3:.2f} ms; sum {state.sum!r} count {state.count}; range sum {ranged.sum!r} count {ranged.count} {kernels}", flush=True)
        ctx.dev_free(values); ctx.dev_free(off_dev); dev.free()
main()
