#!/usr/bin/env python3
"""k_fit_models_lean against the number of waves per SIMD: N chunks of P points (one lane per chunk), the kernel's
time per wave-step. Tells issue-bound (time grows with the waves on the busiest SIMD) from latency-bound (it does not)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import modelardb_rs_amd as mdb  # noqa: E402

def main():
    points = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    counts = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [65536, 98304, 131072, 152588, 163840, 196608, 229376, 262144, 327680]
    ctx = mdb.Context(0)
    info = ctx.device_info()
    simds = info.get("compute_units", 256) * 4
    eb = mdb.error_bound("relative", 1.0)
    for n_chunks in counts:
        total = n_chunks * points
        values = ctx.dev_alloc(4 * total)
        # series of 10 M points cut into chunks, as the bench has them
        series = max(1, total // 10_000_000)
        ctx.synth_values_dev(values, 0, series, total // series)
        total = series * (total // series)
        n_chunks = total // points
        offsets = np.arange(0, n_chunks * points + 1, points, dtype=np.uint64)
        first = (offsets[:-1] % np.uint64(total // series)).astype(np.uint64)
        off_dev, first_dev = ctx.upload_array(offsets), ctx.upload_array(first)
        best = None
        for rep in range(3):
            ctx.profile_enable(True); ctx.profile_reset(); ctx.sync()
            dev = ctx.compress_chunks_dev(0, values, off_dev, n_chunks, eb, 0, 1000, first_dev)
            ctx.sync()
            prof = ctx.profile()
            ms = sum(v[1] for k, v in prof.items() if k.startswith("k_fit_models"))
            names = [k for k in prof if k.startswith("k_fit_models")]
            best = ms if best is None else min(best, ms)
            n_segments = len(dev)
            dev.free()
        waves = (n_chunks + 63) // 64
        print(f"chunks {n_chunks:7d} x {points}: waves {waves:5d} = {waves / simds:.2f} per SIMD, {names} {best:8.3f} ms, "
              f"{best * 1e6 / points:.1f} ns per wave-step, {best / (n_chunks * points) * 1e10:.1f} ms per 1e10 points, {n_segments} segments", flush=True)
        ctx.dev_free(values)
main()
