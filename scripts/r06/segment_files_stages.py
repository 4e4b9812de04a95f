#!/usr/bin/env python3
"""Where the pipelined file leg's time goes: the consumer's stages per row group (main thread) next to the decoders'."""
import os, sys, tempfile, time
import numpy as np, pyarrow as pa, pyarrow.parquet as pq
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import modelardb_rs_amd as mdb
from modelardb_rs_amd import segment_files
from modelardb_rs_amd.segments import SEGMENT_COLUMN_NAMES, SegmentBatch
series, points, chunk = 64, 10_000_000, 65536
ctx = mdb.Context(0)
eb = mdb.error_bound("relative", 1.0)
total = series * points
values = ctx.dev_alloc(4 * total)
ctx.synth_values_dev(values, 0, series, points)
starts = np.arange(0, points, chunk, dtype=np.uint64)
offsets = np.concatenate([(s * points + starts) for s in range(series)] + [np.array([total], dtype=np.uint64)]).astype(np.uint64)
dev = ctx.compress_chunks_dev(0, values, ctx.upload_array(offsets), len(offsets) - 1, eb, 0, 1000, ctx.upload_array(np.tile(starts, series)))
segments = dev.download(); dev.free(); ctx.dev_free(values)
cps = len(starts)
folder = tempfile.mkdtemp(prefix="mdb_segment_files_", dir="/tmp")
paths = []
for first in range(0, series, 8):
    rows = np.nonzero((segments.chunk_index >= first * cps) & (segments.chunk_index < (first + 8) * cps))[0]
    part = segments.take(rows).to_arrow()
    tags = pa.array([f"series-{first + int(c) // cps:05d}" for c in segments.chunk_index[rows]], type=pa.string_view())
    part = pa.RecordBatch.from_arrays(list(part.columns) + [tags], names=list(part.schema.names) + ["tag"])
    paths.append(segment_files.write_segment_file(os.path.join(folder, f"part-{first:05d}.parquet"), part))
out_ts, out_val = ctx.dev_alloc(8 * (total + 1024)), ctx.dev_alloc(4 * (total + 1024))
import pyarrow.dataset as ds
def best(f, n=4):
    times = []
    for _ in range(n):
        t0 = time.perf_counter(); r = f(); times.append(time.perf_counter() - t0)
    return r, min(times)
print("usable cpus", segment_files.usable_cpus(), "arrow cpu_count", pa.cpu_count(), flush=True)
for workers in (8, 16):
    pa.set_cpu_count(workers)
    _, t = best(lambda: ds.dataset(paths, format="parquet").to_table()); print(workers, "dataset.to_table", round(1e3 * t, 1))
    _, t = best(lambda: [pq.read_table(p) for p in paths]); print(workers, "read_table loop", round(1e3 * t, 1))
    for files_ahead in (1, 2, 4, 8):
        def drain():
            first = None; t0 = time.perf_counter(); n = 0
            for b in segment_files.iter_segment_batches(paths, workers=workers, files_ahead=files_ahead):
                if first is None: first = time.perf_counter() - t0
                n += b.num_rows
            return first
        first, t = best(drain); print(workers, "iter_segment_batches drained, files ahead", files_ahead, ":", round(1e3 * t, 1), "first after", round(1e3 * first, 1))
    def drain_upload():
        groups = [g for g, _ in segment_files.load_segments_pipelined(ctx, paths, workers=workers)]
        for g in groups: g.free()
    _, t = best(drain_upload); print(workers, "load_segments_pipelined drained (+ frees)", round(1e3 * t, 1))
    def e2e():
        at = 0; groups = []
        for group, _tags in segment_files.load_segments_pipelined(ctx, paths, workers=workers):
            m, _ = ctx.grid_batch_dev(group, out_ts + 8 * at, out_val + 4 * at, total + 1024 - at)
            at = (at + m + 3) & ~3; groups.append(group)
        ctx.sync(); t1 = time.perf_counter()
        for g in groups: g.free()
        return time.perf_counter() - t1
    frees, t = best(e2e); print(workers, "e2e", round(1e3 * t, 1), "of which frees", round(1e3 * frees, 1), flush=True)
