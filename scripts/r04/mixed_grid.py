#!/usr/bin/env python3
"""grid() and the segment aggregates of the reference's acceptance recipe (all three model types) at the bench's size,
kernel by kernel, checked against the fitted values (lossless) / the counts."""
import os, sys, time, statistics
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import modelardb_rs_amd as mdb  # noqa: E402
import datagen  # noqa: E402

def main():
    total_wanted = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
    bounds = sys.argv[2].split(",") if len(sys.argv) > 2 else ["lossless", "rel1"]
    ctx = mdb.Context(0)
    points, chunk = 1_000_000, 65536
    distinct = max(2, min(64, total_wanted // points // 2 * 2))
    copies = max(1, total_wanted // (distinct * points))
    series, total = distinct * copies, distinct * copies * points
    host_values = np.concatenate([datagen.mixed_series(points, 1000 + s, (1.0, 1.05) if s % 2 else None)[1] for s in range(distinct)])
    values = ctx.dev_alloc(4 * total)
    for copy in range(copies):
        ctx.lib.mdb_dev_upload(ctx.handle, values + 4 * copy * distinct * points, host_values.ctypes.data, host_values.nbytes)
    starts = np.arange(0, points, chunk, dtype=np.uint64)
    offsets = np.concatenate([(s * points + starts) for s in range(series)] + [np.array([total], dtype=np.uint64)]).astype(np.uint64)
    offsets_dev, first_dev = ctx.upload_array(offsets), ctx.upload_array(np.tile(starts, series))
    n_chunks = len(offsets) - 1
    mask = mdb.MDB_AGG_COUNT | mdb.MDB_AGG_MIN | mdb.MDB_AGG_MAX | mdb.MDB_AGG_SUM
    t_lo, t_hi = (points // 4) * 100, (3 * points // 4) * 100
    for name in bounds:
        eb = mdb.error_bound("lossless") if name == "lossless" else mdb.error_bound("relative", float(name[3:]))
        seg = ctx.compress_chunks_dev(0, values, offsets_dev, n_chunks, eb, 0, 100, first_dev)
        n = ctx.grid_count_dev(seg)
        out_ts, out_val = ctx.dev_alloc(8 * n), ctx.dev_alloc(4 * n)
        for label, call in (("grid", lambda: ctx.grid_batch_dev(seg, out_ts, out_val, n)),
                            ("aggregates", lambda: ctx.agg_batch_dev(seg, mask)),
                            ("aggregates between", lambda: ctx.agg_batch_range_dev(seg, t_lo, t_hi, mask))):
            result = call()
            ctx.profile_enable(True); ctx.profile_reset()
            seconds = []
            for _ in range(5):
                ctx.sync(); t0 = time.perf_counter(); result = call(); ctx.sync(); seconds.append(time.perf_counter() - t0)
            kernels = {k: round(v[1] / v[0], 3) for k, v in ctx.profile().items() if v[1] / v[0] > 0.05}
            ctx.profile_enable(False)
            extra = ""
            if label == "grid" and name == "lossless":
                got = ctx.download_array(out_val, 2 * points, np.uint32)
                extra = " bit-exact " + str(bool(np.array_equal(got, host_values[:2 * points].view(np.uint32))))
            if label == "aggregates":
                extra = f" count {result.count} sum {result.sum!r}"
            print(f"{name} {label}: {1e3 * statistics.median(seconds):.3f} ms {kernels}{extra}", flush=True)
        for pointer in (out_ts, out_val):
            ctx.dev_free(pointer)
        seg.free()
main()
