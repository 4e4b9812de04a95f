"""Segment files <-> device columns: the step before grid().

The reference stores compressed segments in Delta Lake tables partitioned by ``field_column``, as
Apache Parquet files written with the properties of
``crates/modelardb_storage/src/lib.rs:248-261`` (16 KiB data pages, 65 536-row row groups, PLAIN
encoding, ZSTD, no dictionary, no statistics, no bloom filter) and sorted by (tags..., start_time)
(``crates/modelardb_storage/src/data_folder/delta_table_writer.rs:64-91``); ``field_column`` is the
partition column and is not stored in the files (``crates/modelardb_types/src/schemas.rs:38-40``).
On the query side ``DataSourceExec`` decodes those files on the host and hands ``GridExec`` batches
of 8 192 rows. Here a whole file (or many) becomes ONE batch of Arrow columns that is uploaded in
one copy, so the GPU sees millions of segments per launch instead of 8 192.

Only the Parquet layer is implemented (pyarrow does the ZSTD/PLAIN decoding); the Delta log is out
of scope.
"""

import os

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq

from .segments import SEGMENT_COLUMN_NAMES, SegmentBatch

FIELD_COLUMN = "field_column"


def partition_directory(table_folder, field_column_index):
    """Hive-style partition directory Delta Lake uses for `field_column`."""
    return os.path.join(table_folder, f"{FIELD_COLUMN}={int(field_column_index)}")


def _storable(batch):
    """Parquet has no view types: BinaryView -> Binary, Utf8View -> Utf8 (the bytes are identical)."""
    arrays, names = [], []
    for name, column in zip(batch.schema.names, batch.columns):
        if name == FIELD_COLUMN:
            continue
        if pa.types.is_binary_view(column.type):
            column = column.cast(pa.binary())
        elif pa.types.is_string_view(column.type):
            column = column.cast(pa.string())
        arrays.append(column)
        names.append(name)
    return pa.table(arrays, names=names)


def write_segment_file(path, segments):
    """Write a RecordBatch with COMPRESSED_SCHEMA (+ tag columns) the way the reference does.
    The rows must already be sorted by (tags..., start_time), as the reference's writer requires."""
    table = _storable(segments)
    tag_names = [n for n in table.schema.names if n not in SEGMENT_COLUMN_NAMES]
    sorting = [pq.SortingColumn(table.schema.get_field_index(n)) for n in tag_names]
    sorting.append(pq.SortingColumn(table.schema.get_field_index("start_time")))
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    pq.write_table(table, path, row_group_size=65536, data_page_size=16384, use_dictionary=False,
                   compression="zstd", write_statistics=False, column_encoding="PLAIN",
                   sorting_columns=sorting)
    return path


def read_segment_files(paths, columns=None):
    """Read whole segment files into ONE Arrow table with BinaryView / Utf8View columns."""
    if isinstance(paths, (str, os.PathLike)):
        paths = [paths]
    tables = [pq.read_table(p, columns=columns) for p in paths]
    table = pa.concat_tables(tables).combine_chunks()
    arrays = []
    for column in table.columns:
        array = column.chunk(0) if column.num_chunks else pa.array([], type=column.type)
        if pa.types.is_binary(array.type) or pa.types.is_large_binary(array.type):
            array = array.cast(pa.binary_view())
        elif pa.types.is_string(array.type) or pa.types.is_large_string(array.type):
            array = array.cast(pa.string_view())
        arrays.append(array)
    return pa.RecordBatch.from_arrays(arrays, names=table.schema.names)


def load_segments(context, paths):
    """Files -> (DeviceSegments, tag columns as a RecordBatch): one upload for everything."""
    batch = read_segment_files(paths)
    segments = SegmentBatch.from_arrow(batch)
    tag_names = [n for n in batch.schema.names if n not in SEGMENT_COLUMN_NAMES]
    tags = batch.select(tag_names)
    return context.upload_segments(segments), tags
