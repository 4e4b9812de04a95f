#!/usr/bin/env python3
"""The fit of calls with few chunks (4 096 chunks of 65 536 points of sine + noise; other shapes by argument)
under a relative bound of 50 % ... 0.1 %: milliseconds per call with the kernels behind it - the library's own
choice (one wave per chunk, chunks with short models left to speculative pieces), one wave per chunk to the end
(k_fit_models_wave, MDB_FIT_WAVE=1) and without that kernel (MDB_FIT_WAVE=0: speculative pieces or one lane per
chunk) - and whether all three return the same segments, column by column, byte for byte.

Usage (on the GPU box): python3 scripts/profile_fit_sweep.py [--chunks N] [--chunk-points N] [--out file.csv]
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


def noisy(n, seed, noise):
    rng = np.random.default_rng(seed)
    i = np.arange(n, dtype=np.float64)
    return (100.0 + 10.0 * np.sin(i / 2000.0) + rng.uniform(-noise, noise, n)).astype(np.float32)


def fit(ctx, values_dev, offsets_dev, n_chunks, eb, repetitions=2):
    best, kernels, dev = None, {}, None
    for repetition in range(repetitions + 1):
        if dev is not None:
            dev.free()
        ctx.profile_enable(True); ctx.profile_reset(); ctx.sync()
        started = time.perf_counter()
        dev = ctx.compress_chunks_dev(0, values_dev, offsets_dev, n_chunks, eb, 0, 1000, 0)
        ctx.sync()
        seconds = time.perf_counter() - started
        if repetition > 0 and (best is None or seconds < best):
            best = seconds
            kernels = {name: total for name, (calls, total) in ctx.profile().items()}
        ctx.profile_enable(False)
    return dev, best, kernels


def top(kernels, count=4):
    return " ".join(f"{name}={ms:.2f}" for name, ms in sorted(kernels.items(), key=lambda item: -item[1])[:count])


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--chunks", type=int, default=4096)
    parser.add_argument("--chunk-points", type=int, default=65536)
    parser.add_argument("--noise", type=float, default=0.5)
    parser.add_argument("--bounds", default="50,20,10,5,2,1,0.7,0.5,0.3,0.1")
    parser.add_argument("--absolute", action="store_true")
    parser.add_argument("--no-compare", action="store_true")
    parser.add_argument("--out", default=None)
    a = parser.parse_args()
    n = a.chunks * a.chunk_points
    ctx = mdb.Context(0)
    values_dev = ctx.upload_array(noisy(n, 11, a.noise))
    offsets = np.arange(0, n + a.chunk_points, a.chunk_points, dtype=np.uint64)
    offsets_dev = ctx.upload_array(offsets)
    rows = []
    for bound in (float(text) for text in a.bounds.split(",")):
        eb = mdb.error_bound("absolute" if a.absolute else "relative", bound)
        row = {"error_bound": ("absolute " if a.absolute else "relative % ") + str(bound), "chunks": a.chunks, "points": n}
        batches = {}
        for mode, setting in (("default", None), ("wave", "1"), ("without", "0")):
            os.environ.pop("MDB_FIT_WAVE", None)
            if setting is not None:
                os.environ["MDB_FIT_WAVE"] = setting
            dev, seconds, kernels = fit(ctx, values_dev, offsets_dev, a.chunks, eb)
            row["segments"] = len(dev)
            row[f"{mode}_ms"] = f"{seconds * 1e3:.2f}"
            row[f"{mode}_kernels_ms"] = top(kernels)
            if not a.no_compare:
                batches[mode] = dev.download()
            dev.free()
        if not a.no_compare:
            row["identical"] = batches["wave"].identical(batches["without"]) and batches["default"].identical(batches["without"])
        print(row, flush=True)
        rows.append(row)
    if a.out:
        with open(a.out, "w", newline="") as f:
            writer = csv.DictWriter(f, fieldnames=list(rows[0]))
            writer.writeheader()
            writer.writerows(rows)
    ctx.dev_free(values_dev); ctx.dev_free(offsets_dev)
    ctx.close()


if __name__ == "__main__":
    main()
