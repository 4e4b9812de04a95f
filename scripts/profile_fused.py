#!/usr/bin/env python3
"""k_grid_fused against the prepass / offsets / tiles pipeline on batches of SIMPLE segments only (runs of exactly
L equal values with a jump between them: every segment a PMC-Mean model of L points, no residuals)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import modelardb_rs_amd as mdb
mdb._abi.RELOAD_OPTIONS_BEFORE_EVERY_CALL = True  # (this script changes MDB_* switches between calls: the library reads them once otherwise)
ctx = mdb.Context(0)
n = 1 << 28
eb = mdb.error_bound("absolute", 0.01)
lengths = [int(x) for x in sys.argv[1:]] or [8, 16, 64]
for length in lengths:
    rng = np.random.default_rng(length)
    n_runs = n // length
    kind = os.environ.get("KIND", "constant")
    level = np.repeat(rng.normal(0.0, 50.0, n_runs), length)
    if kind == "lines":
        level = level + np.repeat(rng.normal(0.0, 0.5, n_runs), length) * (np.arange(n) % length)
    values_dev = ctx.upload_array(level.astype(np.float32))
    offsets = np.arange(0, n + 65536, 65536, dtype=np.uint64)
    offsets_dev = ctx.upload_array(offsets)
    dev = ctx.compress_chunks_dev(0, values_dev, offsets_dev, len(offsets) - 1, eb, 0, 1000, 0)
    ctx.dev_free(values_dev)
    count = ctx.grid_count_dev(dev)
    out_ts, out_val = ctx.dev_alloc(8 * count), ctx.dev_alloc(4 * count)
    for setting in ("0", None):
        if setting is None:
            os.environ.pop("MDB_GRID_FUSED", None)
        else:
            os.environ["MDB_GRID_FUSED"] = setting
        ctx.grid_batch_dev(dev, out_ts, out_val, count)
        ctx.profile_enable(True); ctx.profile_reset(); ctx.sync()
        started = time.perf_counter()
        for _ in range(5):
            produced, metrics = ctx.grid_batch_dev(dev, out_ts, out_val, count)
        ctx.sync()
        wall = (time.perf_counter() - started) / 5
        kernels = {k: round(v[1] / v[0], 3) for k, v in ctx.profile().items() if v[1] / v[0] > 0.01}
        ctx.profile_enable(False)
        algorithmic = 73 * len(dev) + 12 * count
        print(f"L={length} {kind} segments={len(dev)} fused={'on' if setting is None else 'off'}: {1e3 * wall:.3f} ms "
              f"{algorithmic / wall / 1e9 / 8000:.3f} of HBM {kernels}", flush=True)
    ctx.dev_free(out_ts); ctx.dev_free(out_val); ctx.dev_free(offsets_dev); dev.free()
