#!/usr/bin/env python3
"""The parallel MacaqueV decoder on differently shaped lossless data: does every stream get through
it (k_grid_serial ~ 0 ms) and how long does a grid of 16 streams of 65 536 values take?
Development tool (A/B of MDB_MV_PIECE_BITS builds via MDB_HIP_LIBRARY)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import modelardb_rs_amd as mdb  # noqa: E402

def shapes(n, rng):
    i = np.arange(n)
    yield "sine + noise", (100 + 10 * np.sin(i / 300.0) + rng.uniform(-0.05, 0.05, n)).astype(np.float32)
    yield "random walk", np.cumsum(rng.normal(0, 1, n)).astype(np.float32)
    yield "random bits", rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    yield "steps (a new level every ~50 points, tiny jitter)", (np.repeat(rng.normal(0, 100, n // 50 + 1), 50)[:n] + rng.integers(0, 2, n) * 1e-3).astype(np.float32)
    yield "integers 0..9", rng.integers(0, 10, n).astype(np.float32)
    yield "mostly repeats", np.where(rng.random(n) < 0.02, rng.normal(0, 1, n), 0).cumsum().astype(np.float32)
    yield "two regimes", np.where((i // 4000) % 2 == 0, rng.normal(0, 1e-3, n), rng.normal(1e6, 1e5, n)).astype(np.float32)

def main():
    ctx = mdb.Context(0)
    eb = mdb.error_bound("lossless")
    n = 1 << 20
    rng = np.random.default_rng(9)
    ts = np.arange(n, dtype=np.int64) * 1000
    offsets = np.arange(0, n + 1, 65536, dtype=np.uint64)
    for name, values in shapes(n, rng):
        segments = ctx.compress_chunks(ts, values, offsets, eb)
        dev = ctx.upload_segments(segments)
        count = ctx.grid_count_dev(dev)
        out_ts, out_val = ctx.dev_alloc(8 * count), ctx.dev_alloc(4 * count)
        ctx.grid_batch_dev(dev, out_ts, out_val, count)
        ctx.profile_enable(True); ctx.profile_reset(); ctx.sync(); t0 = time.perf_counter()
        for _ in range(3):
            ctx.grid_batch_dev(dev, out_ts, out_val, count)
        ctx.sync(); dt = (time.perf_counter() - t0) / 3
        profile = ctx.profile()
        ctx.profile_enable(False)
        serial = profile.get("k_grid_serial", (1, 0.0))
        same = np.array_equal(ctx.download_array(out_val, count, np.uint32), values.view(np.uint32)[:count])
        kinds = np.bincount(np.asarray(segments.model_type_id), minlength=3).tolist()
        print(f"{name:50s} segments by type {kinds} grid {dt*1e3:6.2f} ms, k_grid_serial {serial[1]/max(serial[0],1):6.3f} ms, bit-exact {same}", flush=True)
        for pointer in (out_ts, out_val):
            ctx.dev_free(pointer)
        dev.free()
main()
