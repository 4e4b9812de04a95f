"""Segment files: written with the reference's Parquet properties and read back as
one batch of Arrow view columns."""

import os

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import pytest

import cases
import oracle_lib as ora
import modelardb_rs_amd as mdb
from modelardb_rs_amd import host, segment_files


def _segments_with_tags():
    eb = cases.error_bounds()["rel5"]
    parts = []
    for seed, tag in ((501, "A"), (502, "a-long-tag-value-beyond-12-bytes")):
        ts, values = cases.synthetic_series(20_000, seed % 2 == 0, (1.0, 1.05), seed)
        batch = ora.try_compress_univariate_time_series(ts, values, eb)
        arrow = host.segments_with_tags(batch.to_arrow(), {"tag": tag})
        parts.append(arrow)
    table = pa.Table.from_batches(parts).combine_chunks()
    return pa.RecordBatch.from_arrays([c.chunk(0) for c in table.columns], names=table.schema.names)


def test_written_file_uses_the_reference_writer_properties(tmp_path):
    segments = _segments_with_tags()
    path = segment_files.write_segment_file(
        os.path.join(segment_files.partition_directory(tmp_path, 1), "part-0.parquet"), segments)
    assert "field_column=1" in path
    metadata = pq.ParquetFile(path).metadata
    assert metadata.num_rows == segments.num_rows
    row_group = metadata.row_group(0)
    assert row_group.num_rows <= 65536
    sorting = [(s.column_index, s.descending) for s in row_group.sorting_columns]
    names = metadata.schema.names
    assert sorting == [(names.index("tag"), False), (names.index("start_time"), False)]
    for c in range(row_group.num_columns):
        column = row_group.column(c)
        assert column.compression == "ZSTD"
        assert "PLAIN" in column.encodings and "RLE_DICTIONARY" not in column.encodings
        assert not column.is_stats_set
    assert "field_column" not in names          # partition column, not stored (schemas.rs:38-40)


def test_file_round_trip_preserves_every_segment(tmp_path):
    segments = _segments_with_tags()
    paths = []
    half = segments.num_rows // 2
    for k, part in enumerate((segments.slice(0, half), segments.slice(half))):
        paths.append(segment_files.write_segment_file(str(tmp_path / f"part-{k}.parquet"), part))
    loaded = segment_files.read_segment_files(paths)
    assert str(loaded.schema.field("timestamps").type) == "binary_view"
    assert str(loaded.schema.field("tag").type) == "string_view"
    assert loaded.num_rows == segments.num_rows
    original = mdb.SegmentBatch.from_arrow(segments)
    back = mdb.SegmentBatch.from_arrow(loaded)
    for got, expected in zip(back.rows(), original.rows()):
        assert got[:4] == expected[:4] and got[6:] == expected[6:]
        assert np.float32(got[4]) == np.float32(expected[4]) and np.float32(got[5]) == np.float32(expected[5])
    assert loaded.column("tag").to_pylist() == segments.column("tag").to_pylist()
    # and the oracle reconstructs the same points from the loaded segments
    a, b = ora.grid_batch(back), ora.grid_batch(original)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))


@pytest.mark.parametrize("workers", [1, 3])
def test_row_groups_streamed_in_order_are_the_files_rows(tmp_path, workers, monkeypatch):
    """iter_segment_batches: the row groups decoded by a pool of threads, handed on in file order - the same rows as
    the whole-file read, group after group (small groups here: four per file)."""
    import pyarrow.parquet as pq
    segments = _segments_with_tags()
    half = segments.num_rows // 2
    real_write = pq.write_table
    monkeypatch.setattr(pq, "write_table", lambda table, path, **kw: real_write(table, path, **{**kw, "row_group_size": max(half // 4, 1)}))
    paths = [segment_files.write_segment_file(str(tmp_path / f"part-{k}.parquet"), part)
             for k, part in enumerate((segments.slice(0, half), segments.slice(half)))]
    whole = mdb.SegmentBatch.from_arrow(segment_files.read_segment_files(paths))
    groups = list(segment_files.iter_segment_batches(paths, workers=workers, ahead=2))
    assert len(groups) >= 8 and all(str(g.schema.field("values").type) == "binary_view" for g in groups)
    streamed = mdb.SegmentBatch.concat([mdb.SegmentBatch.from_arrow(g) for g in groups])
    assert len(streamed) == len(whole)
    for got, expected in zip(streamed.rows(), whole.rows()):
        assert got[:4] == expected[:4] and got[6:8] == expected[6:8]
    assert sum((g.column("tag").to_pylist() for g in groups), []) == segments.column("tag").to_pylist()
