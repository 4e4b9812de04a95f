#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry points at the reference's batch granularity
(DataFusion hands GridExec 8 192 segment rows per batch). Development / documentation tool."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import datagen, oracle_lib as ora, modelardb_rs_amd as mdb  # noqa: E402
mdb._abi.RELOAD_OPTIONS_BEFORE_EVERY_CALL = True  # (this script changes MDB_* switches between calls: the library reads them once otherwise)

def main():
    ctx = mdb.Context(0)
    eb = mdb.error_bound("relative", 1.0)
    n = 3_000_000
    ts = np.arange(n, dtype=np.int64) * 1000
    offs = np.arange(0, n + 65536, 65536, dtype=np.uint64); offs[-1] = n
    parts = [ora.compress_chunks(ts, datagen.sine_series(s, n)[1], offs, eb, n_threads=8) for s in range(4)]
    batch = mdb.SegmentBatch.concat(parts)
    for rows in (8192, len(batch)):
        part = batch.slice(0, min(rows, len(batch)))
        cap = ctx.grid_count(part)
        ctx.grid_batch(part, cap)
        t0 = time.perf_counter(); reps = 5
        for _ in range(reps):
            out = ctx.grid_batch(part, cap)
        dt = (time.perf_counter() - t0) / reps
        print(f"grid host path: {len(part)} segments -> {cap} points: {dt*1e3:.2f} ms/batch, {cap/dt/1e9:.2f} Gpoints/s, {12*cap/dt/1e9:.1f} GB/s over PCIe")
        ctx.grid_batch_owned(part, copy=False)[4]()
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.grid_batch_owned(part, copy=False)[4]()
        dt = (time.perf_counter() - t0) / reps
        print(f"grid owned/pinned: {len(part)} segments -> {cap} points: {dt*1e3:.2f} ms/batch, {cap/dt/1e9:.2f} Gpoints/s, {12*cap/dt/1e9:.1f} GB/s over PCIe")
        mask = mdb.MDB_AGG_COUNT | mdb.MDB_AGG_MIN | mdb.MDB_AGG_MAX | mdb.MDB_AGG_SUM
        ctx.agg_batch(part, mask)
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.agg_batch(part, mask)
        dt = (time.perf_counter() - t0) / reps
        print(f"agg  host path: {len(part)} segments: {dt*1e3:.3f} ms/batch, {len(part)/dt/1e6:.1f} Msegments/s")
    values = np.concatenate([datagen.sine_series(s, n)[1] for s in range(4)]); tss = np.tile(ts, 4)
    o = np.concatenate([offs[:-1] + s * n for s in range(4)] + [[4 * n]]).astype(np.uint64)
    # The C++ GridStream (leftovers, tag replication, slicing into DataFusion-sized batches) on top.
    from modelardb_rs_amd import host
    arrow = host.segments_with_tags(batch.slice(0, 8192).to_arrow(), {"tag": "turbine-0001"})
    for tags in ((), ("tag",)):
        source = arrow if tags else batch.slice(0, 8192).to_arrow()
        stream = host.GridStream(ctx, tag_names=tags, batch_size=8192)
        t0 = time.perf_counter(); rows = 0
        for _ in range(5):
            stream.push(source)
        stream.finish_input()
        batches, _ = stream.collect()
        rows = sum(b.num_rows for b in batches)
        dt = (time.perf_counter() - t0) / 5
        print(f"GridStream ({len(tags)} tag columns): 8192 segments -> {rows // 5} points per input batch: {dt*1e3:.2f} ms, "
              f"{rows / 5 / dt / 1e9:.2f} Gpoints/s, {len(batches) // 5} output batches")
    one_ts, one_values = datagen.sine_series(9, 1_000_000)
    lossless = mdb.error_bound("lossless")
    for mode, label in (("1", "one lane per chunk"), (None, "split mode (auto)")):
        if mode is None:
            os.environ.pop("MDB_FIT_PIECE_POINTS", None)
        else:
            os.environ["MDB_FIT_PIECE_POINTS"] = mode
        ctx.compress_chunks(tss, values, o, eb)
        ctx.profile_enable(True); ctx.profile_reset()
        t0 = time.perf_counter(); got = ctx.compress_chunks(tss, values, o, eb); dt = time.perf_counter() - t0
        kernels = {k: round(v[1], 2) for k, v in ctx.profile().items() if v[1] >= 0.05}
        ctx.profile_enable(False)
        print(f"fit  host path, {label}: {4*n} points in {len(o)-1} chunks: {dt*1e3:.1f} ms, "
              f"{4*n/dt/1e6:.0f} Mpoints/s, {len(got)} segments; kernels ms {kernels}")
        for bound, name in ((eb, "rel 1 %"), (lossless, "lossless")):
            ctx.try_compress_univariate_time_series(one_ts, one_values, bound)
            t0 = time.perf_counter()
            got = ctx.try_compress_univariate_time_series(one_ts, one_values, bound)
            dt = time.perf_counter() - t0
            print(f"fit  one series x 1M points in ONE call ({name}), {label}: {dt*1e3:.1f} ms, {len(got)} segments")

main()
