#!/usr/bin/env python3
"""Aggregates under a time range over the mixed series of bench.py's mixed_models block (resident segments): the
call's milliseconds and kernels for ranges that take nothing, a sliver, the middle half and everything, per kind of
segment (all, no MacaqueV, only PMC-Mean / Swing without residuals).
Usage (on the GPU box): python3 scripts/profile_range_aggregates.py [--series N]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import modelardb_rs_amd as mdb  # noqa: E402
import datagen  # noqa: E402


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--series", type=int, default=64)
    parser.add_argument("--bound", default="lossless")
    a = parser.parse_args()
    ctx = mdb.Context(0)
    points = 1_000_000
    eb = mdb.error_bound("lossless") if a.bound == "lossless" else mdb.error_bound("relative", float(a.bound))
    host_values = np.concatenate([datagen.mixed_series(points, 1000 + s, (1.0, 1.05) if s % 2 else None)[1] for s in range(a.series)])
    values = ctx.upload_array(host_values)
    starts = np.arange(0, points, 65536, dtype=np.uint64)
    offsets = np.concatenate([s * points + starts for s in range(a.series)] + [np.array([a.series * points], dtype=np.uint64)]).astype(np.uint64)
    first_index = np.tile(starts, a.series)
    offsets_dev, first_index_dev = ctx.upload_array(offsets), ctx.upload_array(first_index)
    fitted = ctx.compress_chunks_dev(0, values, offsets_dev, len(offsets) - 1, eb, 0, 100, first_index_dev)
    everything = fitted.download()
    fitted.free()
    mask = mdb.MDB_AGG_COUNT | mdb.MDB_AGG_MIN | mdb.MDB_AGG_MAX | mdb.MDB_AGG_SUM
    ts = np.arange(points, dtype=np.int64) * 100
    kinds = {
        "all": np.arange(len(everything)),
        "no MacaqueV": np.nonzero(everything.model_type_id != mdb.MDB_MACAQUE_V_ID)[0],
        "PMC-Mean / Swing without residuals": np.nonzero((everything.model_type_id != mdb.MDB_MACAQUE_V_ID) &
                                                          (everything.residuals.lengths() == 0))[0],
        "with residuals": np.nonzero(everything.residuals.lengths() > 0)[0],
    }
    for kind, rows in kinds.items():
        batch = everything.take(rows)
        resident = ctx.upload_segments(batch)
        for label, t_lo, t_hi in (("nothing", int(ts[-1]) + 1, int(ts[-1]) + 100), ("a sliver", int(ts[500_000]), int(ts[500_100])),
                                  ("the middle half", int(ts[points // 4]), int(ts[3 * points // 4])), ("everything", -1, int(ts[-1]) + 1)):
            ctx.agg_batch_range_dev(resident, t_lo, t_hi, mask)
            ctx.profile_enable(True); ctx.profile_reset(); ctx.sync()
            started = time.perf_counter()
            for _ in range(3):
                state = ctx.agg_batch_range_dev(resident, t_lo, t_hi, mask)
            ctx.sync()
            ms = (time.perf_counter() - started) / 3 * 1e3
            kernels = {k: round(v[1] / v[0], 3) for k, v in ctx.profile().items() if v[1] / v[0] > 0.02}
            ctx.profile_enable(False)
            print(f"{kind:36s} {len(batch):9d} segments  {label:16s} {ms:7.3f} ms  count {state.count:12d}  {kernels}", flush=True)
        resident.free()
    ctx.close()


if __name__ == "__main__":
    main()
