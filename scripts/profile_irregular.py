#!/usr/bin/env python3
"""Fit and grid of series with irregular (materialised) timestamps on the device: how far are the
delta-of-delta paths from the regular ones? Development tool."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import modelardb_rs_amd as mdb  # noqa: E402

def main():
    series, points, chunk = 100, 10_000_000, 65536
    ctx = mdb.Context(0)
    eb = mdb.error_bound("relative", 1.0)
    total = series * points
    values = ctx.dev_alloc(4 * total)
    ctx.synth_values_dev(values, 0, series, points)
    cps = (points + chunk - 1) // chunk
    offsets = np.array([s * points + c * chunk for s in range(series) for c in range(cps)] + [total], dtype=np.uint64)
    off_dev = ctx.upload_array(offsets)
    rng = np.random.default_rng(5)
    one = np.cumsum(rng.integers(900, 1100, points).astype(np.int64))
    # What irregular usually means in practice: a fixed sampling rate with a sample missing now and then.
    gaps = np.cumsum(np.where(rng.random(points) < 0.01, 2000, 1000).astype(np.int64))
    only = sys.argv[1] if len(sys.argv) > 1 else None
    for label, ts in (("regular", np.tile(np.arange(points, dtype=np.int64) * 1000, series)), ("irregular", np.tile(one, series)),
                      ("1 % gaps", np.tile(gaps, series))):
        if only and only != label.split()[0]:
            continue
        ts_dev = ctx.upload_array(ts)
        ctx.compress_chunks_dev(ts_dev, values, off_dev, len(offsets) - 1, eb, 0, 0, 0).free()
        ctx.sync(); t0 = time.perf_counter()
        dev = ctx.compress_chunks_dev(ts_dev, values, off_dev, len(offsets) - 1, eb, 0, 0, 0)
        ctx.sync(); fit = time.perf_counter() - t0
        n = ctx.grid_count_dev(dev)
        out_ts, out_val = ctx.dev_alloc(8 * n), ctx.dev_alloc(4 * n)
        ctx.grid_batch_dev(dev, out_ts, out_val, n)
        ctx.profile_enable(True); ctx.profile_reset(); ctx.sync(); t0 = time.perf_counter()
        for _ in range(3):
            ctx.grid_batch_dev(dev, out_ts, out_val, n)
        ctx.sync(); grid = (time.perf_counter() - t0) / 3
        kernels = {k: round(v[1] / v[0], 3) for k, v in ctx.profile().items() if v[1] / v[0] > 0.05}
        ctx.profile_enable(False)
        mask = mdb.MDB_AGG_COUNT | mdb.MDB_AGG_MIN | mdb.MDB_AGG_MAX | mdb.MDB_AGG_SUM
        ctx.agg_batch_dev(dev, mask)
        ctx.profile_enable(True); ctx.profile_reset(); ctx.sync(); t0 = time.perf_counter()
        for _ in range(3):
            state = ctx.agg_batch_dev(dev, mask)
        ctx.sync(); agg = (time.perf_counter() - t0) / 3
        agg_kernels = {k: round(v[1] / v[0], 3) for k, v in ctx.profile().items() if v[1] / v[0] > 0.05}
        ctx.profile_enable(False)
        print(f"{label}: COUNT/MIN/MAX/SUM of the segments {agg*1e3:.2f} ms (count {state.count}) {agg_kernels}", flush=True)
        # ... and of the points in the middle half of the time axis (WHERE timestamp BETWEEN)
        t_lo, t_hi = int(ts[points // 4]), int(ts[3 * points // 4])
        ctx.agg_batch_range_dev(dev, t_lo, t_hi, mask)
        ctx.profile_enable(True); ctx.profile_reset(); ctx.sync(); t0 = time.perf_counter()
        for _ in range(3):
            state = ctx.agg_batch_range_dev(dev, t_lo, t_hi, mask)
        ctx.sync(); agg = (time.perf_counter() - t0) / 3
        agg_kernels = {k: round(v[1] / v[0], 3) for k, v in ctx.profile().items() if v[1] / v[0] > 0.05}
        ctx.profile_enable(False)
        print(f"{label}: the same BETWEEN the quartiles of the time axis {agg*1e3:.2f} ms (count {state.count}) {agg_kernels}", flush=True)
        got = ctx.download_array(out_ts, 1_000_000, np.int64)
        assert os.environ.get("MDB_HIP_LIBRARY") or np.array_equal(got, ts[:1_000_000])
        print(f"{label}: fit {fit*1e3:.1f} ms ({total/fit/1e9:.1f} Gpts/s), {len(dev)} segments; grid {grid*1e3:.2f} ms "
              f"({n/grid/1e9:.1f} Gvalues/s) {kernels}", flush=True)
        for pointer in (ts_dev, out_ts, out_val):
            ctx.dev_free(pointer)
        dev.free()
main()
