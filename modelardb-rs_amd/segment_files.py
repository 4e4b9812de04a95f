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

Decoding is the host's work and the slowest leg by far (26 of 31 ms for 857 000 segments, profiles/r05/
segment_files_e2e.txt), so the streaming form overlaps it with everything behind it: row groups - the
unit the reference's scan hands on (65 536 rows, lib.rs:248-261) - are decoded by a pool of threads
(pyarrow releases the GIL) and yielded IN FILE ORDER, so that the upload and the grid() of group k run
while groups k + 1 .. are being decoded (``iter_segment_batches``, ``load_segments_pipelined``).
"""

import concurrent.futures
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
    # (Arrow's dataset reader: files, row groups and columns decoded at once, by as many threads as the process may
    # use - its default is every core of the machine, which a container with a CPU quota pays for in throttling)
    import pyarrow.dataset as ds
    workers = min(usable_cpus(), 16)
    if pa.cpu_count() != workers:
        pa.set_cpu_count(workers)
    return _with_view_columns(ds.dataset([os.fspath(p) for p in paths], format="parquet").to_table(columns=columns))


def _with_view_columns(table):
    """One RecordBatch with BinaryView / Utf8View columns from a decoded table (the bytes are identical)."""
    table = table.combine_chunks()
    arrays = []
    for column in table.columns:
        array = column.chunk(0) if column.num_chunks else pa.array([], type=column.type)
        if pa.types.is_binary(array.type) or pa.types.is_large_binary(array.type):
            array = array.cast(pa.binary_view())
        elif pa.types.is_string(array.type) or pa.types.is_large_string(array.type):
            array = array.cast(pa.string_view())
        arrays.append(array)
    return pa.RecordBatch.from_arrays(arrays, names=table.schema.names)


def usable_cpus():
    """CPUs this process may use at once: its affinity mask, cut down to the cgroup's quota (cpu.max) - a container
    sees every core of the machine and may run on a few of them."""
    cpus = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cpus = min(cpus, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(cpus, 1)


def iter_segment_batches(paths, columns=None, workers=None, ahead=None, then=None, files_ahead=None):
    """The files' row groups as RecordBatches with view columns, in file order (the order GridExec needs: the files
    are sorted by (tags..., start_time)), decoded ahead of the consumer: what the consumer does with group k - upload,
    grid() - overlaps the decoding of the groups behind it. The decoding is Arrow's dataset scanner (C++ threads over
    files, row groups and columns at once - a pool of Python threads calling the reader got 2.3x out of 8 cores);
    `workers` of them, by default as many as the process may use (Arrow's own default is every core of the MACHINE,
    which a container with a CPU quota pays for in throttling). The cast to view types and `then(batch)`, if given, run
    in up to four more threads, and the results come out in order."""
    import pyarrow.dataset as ds
    import queue
    import threading
    if isinstance(paths, (str, os.PathLike)):
        paths = [paths]
    paths = [os.fspath(p) for p in paths]
    workers = workers or min(usable_cpus(), 16)
    ahead = ahead or 2 * workers
    if pa.cpu_count() != workers:
        pa.set_cpu_count(workers)
    scanner = ds.dataset(paths, format="parquet").scanner(columns=columns, batch_size=65536, use_threads=True,
                                                          fragment_readahead=files_ahead or 2, batch_readahead=ahead)

    def finish(batch):
        batch = _with_view_columns(pa.Table.from_batches([batch]))
        return then(batch) if then else batch

    handed_on = queue.Queue(maxsize=ahead)
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(workers, 4)) as pool:
        def produce():
            try:
                for batch in scanner.to_batches():
                    if batch.num_rows:
                        handed_on.put(pool.submit(finish, batch))
                handed_on.put(None)
            except BaseException as error:  # noqa: BLE001 - handed to the consumer, which raises it
                handed_on.put(error)

        producer = threading.Thread(target=produce, daemon=True)
        producer.start()
        try:
            while True:
                item = handed_on.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                yield item.result()
        finally:
            # (a consumer that stops early: let the producer run out, its results are dropped)
            while producer.is_alive():
                try:
                    handed_on.get(timeout=0.05)
                except queue.Empty:
                    pass
            producer.join()


def load_segments_pipelined(context, paths, workers=None):
    """Files -> (DeviceSegments, tag columns) per row group, in order. A group is uploaded by the thread that decoded
    it, through a context of its own (mdb_clone: its own stream; one upload at a time), so the caller's launches on
    group k overlap both the decoding and the upload of the groups behind it. The caller frees each batch when it is
    through with it (hipFree waits for the device: best left until the last launch has been issued)."""
    import threading
    uploader, one_at_a_time = context.clone(), threading.Lock()

    def upload(batch):
        tag_names = [n for n in batch.schema.names if n not in SEGMENT_COLUMN_NAMES]
        host_segments = SegmentBatch.from_arrow(batch)
        with one_at_a_time:
            uploaded = uploader.upload_segments(host_segments)
        # (the batch is the CALLER's from here on: its context outlives the uploader's, which is closed below)
        handed_over = type(uploaded)(context, uploaded.pointer)
        uploaded.pointer = None
        return handed_over, batch.select(tag_names)

    try:
        yield from iter_segment_batches(paths, workers=workers, then=upload)
    finally:
        uploader.close()


def load_segments(context, paths):
    """Files -> (DeviceSegments, tag columns as a RecordBatch): one upload for everything."""
    batch = read_segment_files(paths)
    segments = SegmentBatch.from_arrow(batch)
    tag_names = [n for n in batch.schema.names if n not in SEGMENT_COLUMN_NAMES]
    tags = batch.select(tag_names)
    return context.upload_segments(segments), tags
