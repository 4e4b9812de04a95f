#!/usr/bin/env python3
"""SortedJoinExec over one GridExec per field column (three fields sharing timestamps and a tag), host batches
in, joined batches of 8 192 rows out, polled to the end inside the library. Development tool."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import modelardb_rs_amd as mdb  # noqa: E402
from modelardb_rs_amd import host  # noqa: E402


def main():
    series, points, chunk = 10, 10_000_000, 65536
    ctx = mdb.Context(0)
    total = series * points
    fields = []
    for f in range(3):
        values = ctx.dev_alloc(4 * total)
        ctx.synth_values_dev(values, 100 * f, series, points, 20260101)
        starts = np.arange(0, points, chunk, dtype=np.uint64)
        offsets = (np.arange(series, dtype=np.uint64)[:, None] * np.uint64(points) + starts[None, :]).reshape(-1)
        offsets = np.concatenate([offsets, np.array([total], dtype=np.uint64)])
        dev = ctx.compress_chunks_dev(0, values, ctx.upload_array(offsets), len(offsets) - 1,
                                      mdb.error_bound("relative", 1.0), 0, 1000, ctx.upload_array(np.tile(starts, series)))
        ctx.dev_free(values)
        fields.append(host.segments_with_tags(dev.download().to_arrow(), {"tag": "wind-turbine-0042"}))
        dev.free()
    for tags in ((), ("tag",)):
        for rep in range(2):
            order = ["timestamp", "field", "field", "field"] + ([("tag", "tag")] if tags else [])
            join = host.SortedJoinStream(ctx, 3, order, tag_names=tags, batch_size=8192)
            for index, arrow in enumerate(fields):
                source = arrow if tags else arrow.drop_columns(["tag"])
                for first in range(0, source.num_rows, 8192):
                    join.push(index, source.slice(first, 8192))
            join.finish_input()
            started = time.perf_counter()
            rows, _ = join.drain()
            seconds = time.perf_counter() - started
            print(f"{len(tags)} tag columns: {rows} joined rows of 3 fields in {seconds * 1e3:.1f} ms: "
                  f"{rows / seconds / 1e9:.2f} G rows/s, {3 * rows / seconds / 1e9:.2f} G values/s", flush=True)


main()
