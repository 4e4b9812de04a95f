"""Python face of the C++ host operators (modelardb-rs_amd/csrc/host, libmdb_host.so): GridExec /
GridStream, the five Model*Accumulators and try_compress_*. RecordBatches cross as pyarrow objects
through the Arrow C Data Interface; the operators themselves are C++ and call libmdb_hip.so."""

import ctypes as C
import os

import pyarrow as pa

from . import _abi

HOST_LIBRARY_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "host",
                                 "libmdb_host.so")
# Set by tests/test_host_sanitizers_cpu.py only: a sanitizer build of mdb_host.cpp with the test stand-in
# for libmdb_hip compiled in (tests/stub). It answers from canned fixtures and computes nothing.
LIBRARY_UNDER_TEST = os.environ.get("MDB_HOST_LIBRARY_UNDER_TEST")


class ArrowSchemaC(C.Structure):
    _fields_ = [("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_char_p),
                ("flags", C.c_int64), ("n_children", C.c_int64), ("children", C.c_void_p),
                ("dictionary", C.c_void_p), ("release", C.c_void_p), ("private_data", C.c_void_p)]


class ArrowArrayC(C.Structure):
    _fields_ = [("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64),
                ("n_buffers", C.c_int64), ("n_children", C.c_int64), ("buffers", C.c_void_p),
                ("children", C.c_void_p), ("dictionary", C.c_void_p), ("release", C.c_void_p),
                ("private_data", C.c_void_p)]


class HostError(RuntimeError):
    """Err(...) of the Rust operator this call stands in for."""


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = LIBRARY_UNDER_TEST or HOST_LIBRARY_PATH
        if not LIBRARY_UNDER_TEST:
            _abi.load_hip_library()  # RTLD_GLOBAL: resolves libmdb_host's dependency
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run __graft_entry__.build().")
        _lib = C.CDLL(path)
        _lib.mdbh_last_error.restype = C.c_char_p
        _lib.mdbh_accumulator_size.restype = C.c_uint64
        _lib.mdbh_grid_stream_free.restype = None
        _lib.mdbh_sorted_join_free.restype = None
        _lib.mdbh_accumulator_free.restype = None
        _lib.mdbh_batches_free.restype = None
        _lib.mdbh_udm_free.restype = None
        _lib.mdbh_query_free.restype = None
    return _lib


def _check(code):
    if code != 0:
        raise HostError(lib().mdbh_last_error().decode())


def _export(obj):
    """pyarrow Array / RecordBatch -> (ArrowArrayC, ArrowSchemaC) owned by the caller."""
    array, schema = ArrowArrayC(), ArrowSchemaC()
    obj._export_to_c(C.addressof(array), C.addressof(schema))
    return array, schema


def _import_batch(array, schema):
    return pa.RecordBatch._import_from_c(C.addressof(array), C.addressof(schema))


def _strings(values):
    encoded = [v.encode() for v in values]
    return (C.c_char_p * max(len(encoded), 1))(*encoded), encoded


MODEL_SEGMENT_COLUMNS = ("model_type_id", "start_time", "end_time", "timestamps", "min_value",
                         "max_value", "values", "residuals", "error")


def segments_with_tags(batch, tags):
    """Append Utf8View tag columns to a segment RecordBatch: what DataSourceExec hands GridExec."""
    arrays = [batch.column(i) for i in range(batch.num_columns)]
    names = list(batch.schema.names)
    for name, value in tags.items():
        arrays.append(pa.array([value] * batch.num_rows, type=pa.string_view()))
        names.append(name)
    return pa.RecordBatch.from_arrays(arrays, names=names)


class GridStream:
    """GridExec::execute(...) of crates/modelardb_storage/src/query/grid_exec.rs, fed by hand."""

    READY_SOME, READY_NONE, PENDING = 0, 1, 2

    def __init__(self, context, tag_names=(), limit=None, predicate=(None, None), batch_size=8192):
        """predicate: (lower, upper) bounds on the timestamp, either may be None - or the expression GridExec is
        handed (`maybe_predicate`, grid_exec.rs:60) as text, e.g. "(and (>= timestamp ts:100) (< timestamp ts:900))"
        (mdbhost::parse_expr): the stream works the time range out of it and pushes it into the library."""
        self._context = context
        self.handle = C.c_void_p()
        names, self._keep = _strings(tag_names)
        if isinstance(predicate, str):
            _check(lib().mdbh_grid_exec_create_expr(
                context.handle, names, C.c_int32(len(tag_names)), C.c_int64(-1 if limit is None else limit),
                predicate.encode(), C.c_uint64(batch_size), C.byref(self.handle)))
            return
        lower, upper = predicate
        _check(lib().mdbh_grid_exec_create(
            context.handle, names, C.c_int32(len(tag_names)), C.c_int64(-1 if limit is None else limit),
            C.c_int32(lower is not None), C.c_int64(lower or 0), C.c_int32(upper is not None),
            C.c_int64(upper or 0), C.c_uint64(batch_size), C.byref(self.handle)))

    def push(self, batch):
        array, schema = _export(batch)
        _check(lib().mdbh_grid_stream_push(self.handle, C.byref(array), C.byref(schema)))

    def finish_input(self):
        _check(lib().mdbh_grid_stream_finish_input(self.handle))

    def poll_next(self):
        """Returns (state, RecordBatch or None)."""
        array, schema, state = ArrowArrayC(), ArrowSchemaC(), C.c_int32()
        _check(lib().mdbh_grid_stream_poll_next(self.handle, C.byref(array), C.byref(schema),
                                                C.byref(state)))
        if state.value == self.READY_SOME:
            return state.value, _import_batch(array, schema)
        return state.value, None

    def collect(self):
        batches = []
        while True:
            state, batch = self.poll_next()
            if state != self.READY_SOME:
                return batches, state
            batches.append(batch)

    def drain(self):
        """Poll the stream to its end inside the library (no batch crosses into Python). Returns
        (rows, batches, checksum of the batches' first timestamps)."""
        rows, batches, checksum = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(lib().mdbh_grid_stream_drain(self.handle, C.byref(rows), C.byref(batches), C.byref(checksum)))
        return rows.value, batches.value, checksum.value

    def metrics(self):
        out = (C.c_uint64 * 12)()
        _check(lib().mdbh_grid_stream_metrics(self.handle, out))
        names = ["rows_created"] + [f"rows_created_by_{n}" for n in _abi.MODEL_TYPE_NAMES]
        names += ["segments_with_residuals"] + [f"segments_with_{n}" for n in _abi.MODEL_TYPE_NAMES]
        names += ["regular_segments", "irregular_segments", "output_rows", "elapsed_compute_ns"]
        return dict(zip(names, out))

    def describe(self):
        out = C.create_string_buffer(1024)
        _check(lib().mdbh_grid_exec_describe(self.handle, out, C.c_uint64(1024)))
        return out.value.decode()

    def close(self):
        if self.handle:
            lib().mdbh_grid_stream_free(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def measure_grid_stream(context, segments, batch_size, segments_per_batch=8192, tags=None, predicate=(None, None)):
    """The host operator end to end, for bench.py's host_path: `segments` (a SegmentBatch in host
    memory) handed to a GridStream `segments_per_batch` rows at a time - what the Parquet scan below a
    GridExec delivers - and the stream polled to its end in slices of `batch_size` data points. Every
    byte crosses PCIe: the segment columns up, 12 bytes per data point down into page-locked memory.
    `predicate`: what GridExec is handed for a query with a range on the timestamp (see GridStream).
    Returns (data points, seconds, bytes copied down)."""
    import time
    arrow = segments.to_arrow()
    if tags:  # (tag columns: their value is repeated for every data point of a segment, grid_exec.rs:341-346)
        arrow = segments_with_tags(arrow, tags)
    stream = GridStream(context, tag_names=tuple(tags or ()), batch_size=batch_size, predicate=predicate)
    for first in range(0, arrow.num_rows, segments_per_batch):
        stream.push(arrow.slice(first, min(segments_per_batch, arrow.num_rows - first)))
    stream.finish_input()
    started = time.perf_counter()
    rows, _, _ = stream.drain()
    seconds = time.perf_counter() - started
    stream.close()
    return rows, seconds, 12 * rows


class SortedJoinStream:
    """SortedJoinExec::execute(...) of crates/modelardb_storage/src/query/sorted_join_exec.rs over
    one hand-fed input per field column. With use_grid=True every input is a GridExec (segment
    batches are pushed, the inputs after the first only reconstruct values); with use_grid=False
    the inputs are plain streams of (timestamp, value, tags...) batches.

    return_order: sequence of "timestamp", "field" or ("tag", name), the order of the columns
    returned (SortedJoinColumnType, sorted_join_exec.rs:46-52)."""

    READY_SOME, READY_NONE, PENDING = 0, 1, 2

    def __init__(self, context, n_fields, return_order, tag_names=(), use_grid=True, limit=None,
                 predicate=(None, None), batch_size=8192):
        self._context = context
        self.handle = C.c_void_p()
        names, self._keep = _strings(tag_names)
        kinds, tag_of = [], []
        for element in return_order:
            if element == "timestamp":
                kinds.append(0), tag_of.append("")
            elif element == "field":
                kinds.append(1), tag_of.append("")
            else:
                kinds.append(2), tag_of.append(element[1])
        return_names, self._keep_return = _strings(tag_of)
        lower, upper = predicate
        _check(lib().mdbh_sorted_join_create(
            context.handle if context is not None else None, C.c_int32(n_fields), names,
            C.c_int32(len(tag_names)), (C.c_int32 * max(len(kinds), 1))(*kinds), return_names,
            C.c_int32(len(kinds)), C.c_int32(bool(use_grid)), C.c_int64(-1 if limit is None else limit),
            C.c_int32(lower is not None), C.c_int64(lower or 0), C.c_int32(upper is not None),
            C.c_int64(upper or 0), C.c_uint64(batch_size), C.byref(self.handle)))

    def push(self, input_index, batch):
        array, schema = _export(batch)
        _check(lib().mdbh_sorted_join_push(self.handle, C.c_int32(input_index), C.byref(array),
                                           C.byref(schema)))

    def finish_input(self, input_index=None):
        indices = range(self.n_inputs()) if input_index is None else [input_index]
        for index in indices:
            _check(lib().mdbh_sorted_join_finish_input(self.handle, C.c_int32(index)))

    def n_inputs(self):
        return int(self.describe().split("|children=")[1].split("|")[0])

    def poll_next(self):
        array, schema, state = ArrowArrayC(), ArrowSchemaC(), C.c_int32()
        _check(lib().mdbh_sorted_join_poll_next(self.handle, C.byref(array), C.byref(schema),
                                                C.byref(state)))
        if state.value == self.READY_SOME:
            return state.value, _import_batch(array, schema)
        return state.value, None

    def collect(self):
        batches = []
        while True:
            state, batch = self.poll_next()
            if state != self.READY_SOME:
                return batches, state
            batches.append(batch)

    def drain(self):
        """Poll the join to its end inside the library; returns (rows, batches)."""
        rows, batches = C.c_uint64(), C.c_uint64()
        _check(lib().mdbh_sorted_join_drain(self.handle, C.byref(rows), C.byref(batches)))
        return rows.value, batches.value

    def describe(self):
        out = C.create_string_buffer(1024)
        _check(lib().mdbh_sorted_join_describe(self.handle, out, C.c_uint64(1024)))
        return out.value.decode()

    def close(self):
        if self.handle:
            lib().mdbh_sorted_join_free(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _ModelAccumulator:
    KIND = None

    def __init__(self, context, time_range=None):
        """time_range: (t_lo, t_hi) for the accumulators of a query with a range on the timestamp (what the patched
        ModelSimpleAggregates rule hands them, rust/patches/0002): only the data points inside it count."""
        self.handle = C.c_void_p()
        lo, hi = time_range if time_range is not None else (0, 0)
        _check(lib().mdbh_accumulator_create_range(context.handle, C.c_int32(self.KIND), C.c_int32(time_range is not None),
                                                   C.c_int64(lo), C.c_int64(hi), C.byref(self.handle)))

    def update_batch(self, batch):
        array, schema = _export(batch)
        _check(lib().mdbh_accumulator_update_batch(self.handle, C.byref(array), C.byref(schema)))

    def state(self):
        kinds, f64, i64, nulls = (C.c_int32 * 2)(), (C.c_double * 2)(), (C.c_int64 * 2)(), (C.c_int32 * 2)()
        n = C.c_int32()
        _check(lib().mdbh_accumulator_state(self.handle, kinds, f64, i64, C.byref(n), nulls))
        return [None if nulls[k] else (i64[k] if kinds[k] in (0, 1) else f64[k]) for k in range(n.value)]

    def size(self):
        return lib().mdbh_accumulator_size(self.handle)

    def merge_batch(self):
        _check(lib().mdbh_accumulator_unreachable(self.handle, C.c_int32(0)))

    def evaluate(self):
        _check(lib().mdbh_accumulator_unreachable(self.handle, C.c_int32(1)))

    def __del__(self):
        try:
            if self.handle:
                lib().mdbh_accumulator_free(self.handle)
        except Exception:
            pass


def measure_accumulator(context, segments, accumulator_class, segments_per_batch=8192, time_range=None):
    """One of the five accumulators fed `segments` (a SegmentBatch in host memory) `segments_per_batch` rows at a
    time - the batches DataFusion hands update_batch; the accumulator keeps them until 262 144 segments are pending or
    its state is read and folds them with ONE mdb_agg_batch_list (mdb_agg_batch_range_list under `time_range`), as
    rust/patches/0002 makes it - for bench.py. Returns (state, seconds)."""
    import time
    arrow = segments.to_arrow()
    batches = [arrow.slice(first, min(segments_per_batch, arrow.num_rows - first))
               for first in range(0, arrow.num_rows, segments_per_batch)]
    accumulator = accumulator_class(context, time_range)
    started = time.perf_counter()
    for batch in batches:
        accumulator.update_batch(batch)
    state = accumulator.state()   # (folds what is still pending: part of the work)
    seconds = time.perf_counter() - started
    return state, seconds


class ModelCountAccumulator(_ModelAccumulator):
    KIND = 0


class ModelMinAccumulator(_ModelAccumulator):
    KIND = 1


class ModelMaxAccumulator(_ModelAccumulator):
    KIND = 2


class ModelSumAccumulator(_ModelAccumulator):
    KIND = 3


class ModelAvgAccumulator(_ModelAccumulator):
    KIND = 4


def rewrite_filters(filters, n_fields=2):
    """rewrite_and_combine_filters (query/time_series_table.rs:269-288) over the query schema (timestamp, field_1..,
    tag): the filters as text (see GridStream) -> (parquet filter, grid filter) as text, None where the reference
    returns None."""
    parquet, grid = C.create_string_buffer(2048), C.create_string_buffer(2048)
    _check(lib().mdbh_rewrite_filters(C.c_int32(n_fields), ";".join(filters).encode(), parquet, grid, C.c_uint64(2048)))
    return parquet.value.decode() or None, grid.value.decode() or None


def time_range_of_predicate(predicate):
    """What the patched GridStream::new and the patched rule take from a predicate over `timestamp`: None, or
    (t_lo, t_hi, exact) - exact: the predicate IS that range."""
    found, exact, lo, hi = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int64()
    _check(lib().mdbh_time_range_of_predicate(predicate.encode(), C.byref(found), C.byref(exact), C.byref(lo), C.byref(hi)))
    return (lo.value, hi.value, bool(exact.value)) if found.value else None


class AggregateQuery:
    """SELECT agg(field), ... FROM table [WHERE ...] over a TimeSeriesTable fed by hand: the physical plan DataFusion
    and TimeSeriesTable::scan make of it (query/time_series_table.rs:494-671), optionally rewritten by the
    ModelSimpleAggregates rule (optimizer/model_simple_aggregates.rs:176-302 as rust/patches/0002 extends it), and run.
    aggregates: [("sum", 0), ...] (function, field column); filters: expressions as text (see GridStream)."""

    def __init__(self, context, n_fields=1, tag_names=()):
        self._context = context
        self.handle = C.c_void_p()
        names, self._keep = _strings(tag_names)
        _check(lib().mdbh_query_table_create(context.handle if context is not None else None, C.c_int32(n_fields),
                                             names, C.c_int32(len(tag_names)), C.byref(self.handle)))

    def push_segments(self, field, batch):
        array, schema = _export(batch)
        _check(lib().mdbh_query_table_push(self.handle, C.c_int32(field), C.byref(array), C.byref(schema)))

    def plan(self, aggregates, filters=(), optimize=True):
        text = ",".join(f"{function}:{field}" for function, field in aggregates)
        context = self._context.handle if self._context is not None else None
        _check(lib().mdbh_query_plan(self.handle, context, text.encode(), ";".join(filters).encode(),
                                     C.c_int32(bool(optimize))))
        return self

    def describe(self):
        out = C.create_string_buffer(4096)
        _check(lib().mdbh_query_describe(self.handle, out, C.c_uint64(4096)))
        return out.value.decode()

    def levels(self):
        """The plan level by level, like assert_eq_physical_plan_expected (model_simple_aggregates.rs:765-790)."""
        return [line.split(",") for line in self.describe().splitlines() if not line.startswith("aggregate ")]

    def aggregates(self):
        return [line.split(" ", 1)[1] for line in self.describe().splitlines() if line.startswith("aggregate ")]

    def execute(self, batch_size=8192):
        values, nulls, n = (C.c_double * 16)(), (C.c_int32 * 16)(), C.c_int32()
        _check(lib().mdbh_query_execute(self.handle, C.c_uint64(batch_size), values, nulls, C.byref(n)))
        return [None if nulls[k] else values[k] for k in range(n.value)]

    def close(self):
        if self.handle:
            lib().mdbh_query_free(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def try_compress_univariate_time_series(context, uncompressed_timestamps, uncompressed_values,
                                        error_bound, tags, field_column_index):
    """compression.rs:191-275. `tags` is an ordered {tag column name: tag value} mapping."""
    ts = pa.array(uncompressed_timestamps, type=pa.int64()).cast(pa.timestamp("us"))
    values = pa.array(uncompressed_values, type=pa.float32())
    ts_c, values_c = _export(ts), _export(values)
    names, keep_names = _strings(list(tags.keys()))
    vals, keep_values = _strings(list(tags.values()))
    out_array, out_schema = ArrowArrayC(), ArrowSchemaC()
    _check(lib().mdbh_try_compress_univariate_time_series(
        context.handle, C.byref(ts_c[0]), C.byref(ts_c[1]), C.byref(values_c[0]), C.byref(values_c[1]),
        error_bound, names, vals, C.c_int32(len(tags)), C.c_int16(field_column_index),
        C.byref(out_array), C.byref(out_schema)))
    return _import_batch(out_array, out_schema)


def try_compress_multivariate_time_series(context, batch, timestamp_column, field_columns,
                                          tag_columns, error_bounds):
    """compression.rs:42-179. `error_bounds` maps field column index -> ErrorBound."""
    array, schema = _export(batch)
    bounds = (_abi.ErrorBoundC * batch.num_columns)()
    for index in range(batch.num_columns):
        bounds[index] = error_bounds.get(index, _abi.ErrorBoundC(_abi.MDB_EB_LOSSLESS, 0.0))
    fields = (C.c_int32 * len(field_columns))(*field_columns)
    tags = (C.c_int32 * max(len(tag_columns), 1))(*tag_columns)
    handle, n = C.c_void_p(), C.c_int32()
    _check(lib().mdbh_try_compress_multivariate_time_series(
        context.handle, C.byref(array), C.byref(schema), C.c_int32(timestamp_column), fields,
        C.c_int32(len(field_columns)), tags, C.c_int32(len(tag_columns)), bounds, C.byref(handle),
        C.byref(n)))
    try:
        out = []
        for index in range(n.value):
            out_array, out_schema = ArrowArrayC(), ArrowSchemaC()
            _check(lib().mdbh_batches_get(handle, C.c_int32(index), C.byref(out_array),
                                          C.byref(out_schema)))
            out.append(_import_batch(out_array, out_schema))
        return out
    finally:
        lib().mdbh_batches_free(handle)


class UncompressedDataManager:
    """The compress side of crates/modelardb_server/src/storage/uncompressed_data_manager.rs with
    finished buffers compressed together in one GPU launch per error bound."""

    def __init__(self, context, schema, timestamp_column, field_columns, tag_columns, error_bounds,
                 buffer_capacity=65536):
        self.handle = C.c_void_p()
        bounds = (_abi.ErrorBoundC * len(schema.names))()
        for index in range(len(schema.names)):
            bounds[index] = error_bounds.get(index, _abi.ErrorBoundC(_abi.MDB_EB_LOSSLESS, 0.0))
        fields = (C.c_int32 * len(field_columns))(*field_columns)
        tags = (C.c_int32 * max(len(tag_columns), 1))(*tag_columns)
        names, self._keep = _strings([schema.names[i] for i in tag_columns])
        _check(lib().mdbh_udm_create(context.handle, C.c_int32(timestamp_column), fields,
                                     C.c_int32(len(field_columns)), tags, names,
                                     C.c_int32(len(tag_columns)), bounds, C.c_int32(len(schema.names)),
                                     C.c_uint64(buffer_capacity), C.byref(self.handle)))

    def insert_data_points(self, batch):
        array, schema = _export(batch)
        _check(lib().mdbh_udm_insert_data_points(self.handle, C.byref(array), C.byref(schema)))

    def flush(self):
        _check(lib().mdbh_udm_flush(self.handle))

    def counts(self):
        active, finished = C.c_uint64(), C.c_uint64()
        _check(lib().mdbh_udm_counts(self.handle, C.byref(active), C.byref(finished)))
        return active.value, finished.value

    def compress_finished_buffers(self):
        handle, n = C.c_void_p(), C.c_int32()
        _check(lib().mdbh_udm_compress_finished_buffers(self.handle, C.byref(handle), C.byref(n)))
        try:
            out = []
            for index in range(n.value):
                out_array, out_schema = ArrowArrayC(), ArrowSchemaC()
                _check(lib().mdbh_batches_get(handle, C.c_int32(index), C.byref(out_array),
                                              C.byref(out_schema)))
                out.append(_import_batch(out_array, out_schema))
            return out
        finally:
            lib().mdbh_batches_free(handle)

    def __del__(self):
        try:
            if self.handle:
                lib().mdbh_udm_free(self.handle)
        except Exception:
            pass
