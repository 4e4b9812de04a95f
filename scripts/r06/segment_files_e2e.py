#!/usr/bin/env python3
"""(round 6: + the pipelined form) N2 on the record: segment files (the reference's Parquet properties: ZSTD, PLAIN, 65 536-row groups,
crates/modelardb_storage/src/lib.rs:248-261) -> Arrow columns -> device -> grid(), end to end, with the pyarrow decode
timed by itself. The segments are those of the bench's first series (relative bound 1 %); one file per 8 series."""
import json, os, sys, tempfile, time
import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import modelardb_rs_amd as mdb  # noqa: E402
from modelardb_rs_amd import host, segment_files  # noqa: E402

def main():
    series, points, chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 10_000_000, 65536
    out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "segment_files_e2e.json")
    ctx = mdb.Context(0)
    eb = mdb.error_bound("relative", 1.0)
    total = series * points
    values = ctx.dev_alloc(4 * total)
    ctx.synth_values_dev(values, 0, series, points)
    starts = np.arange(0, points, chunk, dtype=np.uint64)
    offsets = np.concatenate([(s * points + starts) for s in range(series)] + [np.array([total], dtype=np.uint64)]).astype(np.uint64)
    dev = ctx.compress_chunks_dev(0, values, ctx.upload_array(offsets), len(offsets) - 1, eb, 0, 1000, ctx.upload_array(np.tile(starts, series)))
    segments = dev.download()
    dev.free(); ctx.dev_free(values)
    chunks_per_series = len(starts)
    folder = tempfile.mkdtemp(prefix="mdb_segment_files_", dir="/tmp")
    paths, file_bytes = [], 0
    for first in range(0, series, 8):
        rows = np.nonzero((segments.chunk_index >= first * chunks_per_series) & (segments.chunk_index < (first + 8) * chunks_per_series))[0]
        part = segments.take(rows).to_arrow()
        tags = pa.array([f"series-{first + int(c) // chunks_per_series:05d}" for c in segments.chunk_index[rows]], type=pa.string_view())
        part = pa.RecordBatch.from_arrays(list(part.columns) + [tags], names=list(part.schema.names) + ["tag"])
        path = segment_files.write_segment_file(os.path.join(segment_files.partition_directory(folder, 1), f"part-{first:05d}.parquet"), part)
        paths.append(path); file_bytes += os.path.getsize(path)
    def best(f, n=3):
        times = []
        for _ in range(n):
            t0 = time.perf_counter(); result = f(); times.append(time.perf_counter() - t0)
        return result, min(times)
    tables, decode_threads = best(lambda: [pq.read_table(p) for p in paths])
    _, decode_one = best(lambda: [pq.read_table(p, use_threads=False) for p in paths])
    batch, to_views = best(lambda: segment_files.read_segment_files(paths))
    host_segments = mdb.SegmentBatch.from_arrow(batch)
    resident, upload = best(lambda: ctx.upload_segments(host_segments))
    n = ctx.grid_count_dev(resident)
    out_ts, out_val = ctx.dev_alloc(8 * (n + 1024)), ctx.dev_alloc(4 * (n + 1024))
    ctx.grid_batch_dev(resident, out_ts, out_val, n); ctx.sync()
    def grid():
        ctx.grid_batch_dev(resident, out_ts, out_val, n); ctx.sync()
    _, grid_seconds = best(grid)
    def end_to_end():
        loaded, _tags = segment_files.load_segments(ctx, paths)
        ctx.grid_batch_dev(loaded, out_ts, out_val, n); ctx.sync()
        loaded.free()
    _, e2e = best(end_to_end)
    def end_to_end_pipelined(workers):
        done = at = 0
        groups = []
        for group, _tags in segment_files.load_segments_pipelined(ctx, paths, workers=workers):
            m, _ = ctx.grid_batch_dev(group, out_ts + 8 * at, out_val + 4 * at, total + 1024 - at)
            done += m
            at = (at + m + 3) & ~3  # (output columns begin on 16-byte boundaries)
            groups.append(group)
        ctx.sync()
        for group in groups:
            group.free()
        assert done == total
    pipelined = {}
    for workers in (4, 8, 16):
        _, pipelined[str(workers)] = best(lambda: end_to_end_pipelined(workers), 5)
    check_ts = ctx.download_array(out_ts, 1_000_000, np.int64)
    assert np.array_equal(check_ts, np.arange(1_000_000, dtype=np.int64) * 1000)  # (the first row group's first points)
    assert n == total
    row = {"series": series, "points": total, "segments": len(segments), "files": len(paths), "file_bytes": file_bytes,
           "bytes_per_segment_on_disk": file_bytes / len(segments),
           "pyarrow_decode_s": {"threads": decode_threads, "one_thread": decode_one},
           "read_segment_files_s": to_views, "upload_s": upload, "grid_s": grid_seconds, "end_to_end_s": e2e,
           "end_to_end_pipelined_s_by_decoder_threads": pipelined, "row_groups": sum(pq.ParquetFile(p).metadata.num_row_groups for p in paths),
           "end_to_end_values_per_s": total / e2e, "end_to_end_segments_per_s": len(segments) / e2e,
           "decode_MB_per_s": {"threads": file_bytes / decode_threads / 1e6, "one_thread": file_bytes / decode_one / 1e6},
           "note": "files written with the reference's writer properties; pq.read_table alone (best of 3, page cache warm), "
                   "read_segment_files = decode + concat + cast to view types, upload = one mdb_segments_upload, grid = one "
                   "launch over all segments into resident columns, end to end = load_segments + grid; pipelined = load_segments_pipelined: row groups decoded by a pool of threads, each uploaded and reconstructed (behind the last one's points) while the next ones are decoded"}
    json.dump(row, open(out_path, "w"), indent=1)
    print(json.dumps(row))
main()
