#!/usr/bin/env python3
"""Host-to-host latency of mdb_compress_chunk_list by launch size (chunks of 65 536 points of the bench's series),
the path of a handful of chunks against the general driver (MDB_FIT_SMALL=0), with the kernels' HIP-event times."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import modelardb_rs_amd as mdb  # noqa: E402
from modelardb_rs_amd import _abi  # noqa: E402
_abi.RELOAD_OPTIONS_BEFORE_EVERY_CALL = True  # (the switches change between calls)
import datagen  # noqa: E402
import oracle_lib as ora  # noqa: E402

def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "relative"
    bound = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    eb = mdb.error_bound(kind, bound) if kind != "lossless" else mdb.error_bound("lossless")
    ctx = mdb.Context(0)
    points = 65536
    series = [datagen.bench_series(s, 4 * points, 0x4D44425F52454631) for s in range(16)]
    ts = np.arange(4 * points, dtype=np.int64) * 1000
    chunks = [(ts[k * points:(k + 1) * points], v[k * points:(k + 1) * points]) for v in series for k in range(4)]
    for n in (1, 4, 16, 64):
        launch = chunks[:n]
        expected = ora.compress_chunks(np.concatenate([c[0] for c in launch]), np.concatenate([c[1] for c in launch]),
                                       np.arange(0, (n + 1) * points, points, dtype=np.uint64), eb)
        row = {}
        for label, small in (("small", None), ("general", "0")):
            if small is None: os.environ.pop("MDB_FIT_SMALL", None)
            else: os.environ["MDB_FIT_SMALL"] = small
            got = ctx.compress_chunk_list(launch, eb)
            assert got.identical(expected), (label, n)
            timings = []
            for _ in range(7):
                ctx.compress_chunk_list(launch, eb)
                timings.append(ctx.last_call_seconds)
            ctx.profile_enable(True); ctx.profile_reset()
            ctx.compress_chunk_list(launch, eb)
            kernels = {k: round(v[1], 3) for k, v in ctx.profile().items() if v[1] >= 0.003}
            ctx.profile_enable(False)
            row[label] = (1e3 * min(timings), 1e3 * float(np.median(timings)), kernels)
        os.environ.pop("MDB_FIT_SMALL", None)
        print(f"{n:3d} chunks ({len(expected)} segments): small {row['small'][0]:.3f} ms (median {row['small'][1]:.3f}), "
              f"general {row['general'][0]:.3f} ms (median {row['general'][1]:.3f})")
        print("      small kernels ms:", row["small"][2])
main()
