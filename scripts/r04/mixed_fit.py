#!/usr/bin/env python3
"""The fit of the reference's acceptance recipe (runs of Constant / Linear / Random data, compression.rs:733-863) at
the bench's size, kernel by kernel, under the environment it is started with (MDB_FIT_*)."""
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
    for name in bounds:
        eb = mdb.error_bound("lossless") if name == "lossless" else mdb.error_bound("relative", float(name[3:]))
        ctx.compress_chunks_dev(0, values, offsets_dev, n_chunks, eb, 0, 100, first_dev).free()
        ctx.sync(); ctx.profile_enable(True); ctx.profile_reset()
        seconds = []
        for _ in range(3):
            ctx.sync(); t0 = time.perf_counter()
            seg = ctx.compress_chunks_dev(0, values, offsets_dev, n_chunks, eb, 0, 100, first_dev)
            ctx.sync(); seconds.append(time.perf_counter() - t0)
            n_seg = len(seg); seg.free()
        kernels = {k: round(v[1] / 3, 2) for k, v in ctx.profile().items() if v[1] / 3 > 0.2}
        ctx.profile_enable(False)
        print(f"{name}: {total} points, {n_chunks} chunks, {n_seg} segments: {1e3 * statistics.median(seconds):.2f} ms  {kernels}", flush=True)
main()
