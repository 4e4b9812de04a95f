#!/usr/bin/env python3
"""The fit under a LOSSLESS bound with and without the straight-line fitter (k_fit_models_lean<..., lossless>;
MDB_FIT_LEAN=0: k_fit_models), with the one-wave-per-chunk kernel switched off (MDB_FIT_WAVE=0) so that the two are
what runs, and the library's own choice beside them. Series: the reference's acceptance recipe (bench.py's
mixed_models: Constant / Linear / Random runs of 50..500 points, every second series with noise) and plain noise
(every start point rejected: all MacaqueV). Chunk shapes: few long chunks (speculative pieces) and many short ones
(one lane per chunk). Checks that every mode returns the same segments, byte for byte.

Usage (on the GPU box): python3 scripts/profile_fit_lossless.py [--points N] [--out file.csv]
"""
import argparse
import csv
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import modelardb_rs_amd as mdb  # noqa: E402
mdb._abi.RELOAD_OPTIONS_BEFORE_EVERY_CALL = True  # (this script changes MDB_* switches between calls: the library reads them once otherwise)
import datagen  # noqa: E402


def fit(ctx, values_dev, offsets_dev, n_chunks, eb, first_index_dev):
    best, kernels, dev = None, {}, None
    for repetition in range(3):
        if dev is not None:
            dev.free()
        ctx.profile_enable(True); ctx.profile_reset(); ctx.sync()
        started = time.perf_counter()
        dev = ctx.compress_chunks_dev(0, values_dev, offsets_dev, n_chunks, eb, 0, 100, first_index_dev)
        ctx.sync()
        seconds = time.perf_counter() - started
        if repetition > 0 and (best is None or seconds < best):
            best = seconds
            kernels = {name: total for name, (calls, total) in ctx.profile().items()}
        ctx.profile_enable(False)
    return dev, best, kernels


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--points", type=int, default=512_000_000)
    parser.add_argument("--out", default=None)
    a = parser.parse_args()
    ctx = mdb.Context(0)
    eb = mdb.error_bound("lossless")
    per_series = 1_000_000
    distinct = 32
    copies = max(1, a.points // (distinct * per_series))
    total = distinct * copies * per_series
    rng = np.random.default_rng(5)
    series = {
        "mixed": np.concatenate([datagen.mixed_series(per_series, 1000 + s, (1.0, 1.05) if s % 2 else None)[1]
                                 for s in range(distinct)]),
        "noise": rng.uniform(100.0, 200.0, distinct * per_series).astype(np.float32),
    }
    rows = []
    for name, host_values in series.items():
        values_dev = ctx.dev_alloc(4 * total)
        for copy in range(copies):
            ctx.lib.mdb_dev_upload(ctx.handle, values_dev + 4 * copy * host_values.size, host_values.ctypes.data,
                                   host_values.nbytes)
        for chunk_points in (1_000_000, 65_536, 4_000):
            starts = np.arange(0, per_series, chunk_points, dtype=np.uint64)
            offsets = np.concatenate([s * per_series + starts for s in range(distinct * copies)] +
                                     [np.array([total], dtype=np.uint64)]).astype(np.uint64)
            first_index = np.tile(starts, distinct * copies)
            offsets_dev, first_index_dev = ctx.upload_array(offsets), ctx.upload_array(first_index)
            n_chunks = len(offsets) - 1
            row = {"series": name, "points": total, "chunks": n_chunks, "chunk_points": chunk_points}
            batches = {}
            for mode, env in (("default", {}), ("wave", {"MDB_FIT_WAVE": "1"}), ("lean", {"MDB_FIT_WAVE": "0"}),
                              ("plain", {"MDB_FIT_WAVE": "0", "MDB_FIT_LEAN": "0"})):
                for key in ("MDB_FIT_WAVE", "MDB_FIT_LEAN"):
                    os.environ.pop(key, None)
                os.environ.update(env)
                dev, seconds, kernels = fit(ctx, values_dev, offsets_dev, n_chunks, eb, first_index_dev)
                row["segments"] = len(dev)
                row[f"{mode}_ms"] = f"{seconds * 1e3:.2f}"
                row[f"{mode}_kernels_ms"] = " ".join(f"{k}={ms:.2f}" for k, ms in sorted(kernels.items(), key=lambda i: -i[1])[:3])
                batches[mode] = dev.download()
                dev.free()
            row["identical"] = all(batches[m].identical(batches["plain"]) for m in ("default", "wave", "lean"))
            print(row, flush=True)
            rows.append(row)
            ctx.dev_free(offsets_dev); ctx.dev_free(first_index_dev)
        ctx.dev_free(values_dev)
    if a.out:
        with open(a.out, "w", newline="") as f:
            writer = csv.DictWriter(f, fieldnames=list(rows[0]))
            writer.writeheader()
            writer.writerows(rows)
    ctx.close()
    if not all(row["identical"] for row in rows):
        raise SystemExit("the modes disagree")


if __name__ == "__main__":
    main()
