"""Thin Python binding of the C ABI in include/mdb.h (ctypes; no torch types cross the boundary).

``Context`` owns one ``mdb_ctx`` (one HIP stream + scratch). The host methods take numpy-backed
``SegmentBatch`` objects whose buffers are handed to the library as plain pointers, exactly like
the Rust shim in INTEGRATION.md would hand over Arrow buffers. There is no CPU fallback: if the HIP
library cannot be loaded or a call fails, ``HipError`` is raised.
"""

import ctypes as C
import time

import numpy as np

from . import _abi
from .segments import SegmentBatch


class HipError(RuntimeError):
    pass


class DeviceSegments:
    """A batch of segments resident in HBM (``mdb_segments_owned`` with on_device = 1)."""

    def __init__(self, context, pointer):
        self._context = context
        self.pointer = pointer

    def __len__(self):
        return int(self.pointer.contents.seg.n)

    @property
    def seg(self):
        return self.pointer.contents.seg

    def free(self):
        if self.pointer:
            self._context.lib.mdb_segments_free(self.pointer)
            self.pointer = None

    def download(self):
        """Copy back to the host as a ``SegmentBatch``."""
        out = C.POINTER(_abi.SegmentsOwnedC)()
        self._context._check(self._context.lib.mdb_segments_download(
            self._context.handle, self.pointer, C.byref(out)))
        try:
            return SegmentBatch.from_owned(out)
        finally:
            self._context.lib.mdb_segments_free(out)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class GridTicket:
    """An outstanding mdb_grid_submit. wait() returns (timestamps, values, rows_per_segment, metrics,
    tag_views) as copies ((n, 16) uint8 arrays for the tag views) and consumes the ticket."""

    def __init__(self, context, pointer, keep, n_tags, values_only):
        self._context, self._pointer, self._keep = context, pointer, keep
        self._n_tags, self._values_only = n_tags, values_only

    def wait(self):
        context, pointer = self._context, self._pointer
        self._pointer = None
        out = C.POINTER(_abi.GridResultC)()
        context._check(context.lib.mdb_grid_wait(pointer, C.byref(out)))
        self._keep = None
        result = out.contents
        n, n_segments = int(result.n), int(result.n_segments)

        def copy_of(pointer, count, dtype, width=1):
            if count == 0:
                return np.zeros((0, width) if width > 1 else 0, dtype=dtype)
            nbytes = count * width * np.dtype(dtype).itemsize
            array = np.frombuffer((C.c_char * nbytes).from_address(pointer), dtype=dtype).copy()
            return array.reshape(count, width) if width > 1 else array

        try:
            ts = None if self._values_only else copy_of(result.timestamps, n, np.int64)
            values = copy_of(result.values, n, np.float32)
            rows = copy_of(result.rows_per_segment, n_segments, np.uint32)
            tags = [copy_of(context.lib.mdb_grid_result_tag_views(out, t), n, np.uint8, 16)
                    for t in range(self._n_tags)]
            return ts, values, rows, result.metrics.as_dict(), tags
        finally:
            context.lib.mdb_grid_result_free(out)

    def cancel(self):
        if self._pointer is not None:
            self._context.lib.mdb_grid_cancel(self._pointer)
            self._pointer = None

    def __del__(self):
        try:
            self.cancel()
        except Exception:
            pass


def comm_unique_id():
    """ncclGetUniqueId through the C ABI: 128 bytes, made by one rank."""
    lib = _abi.load_hip_library()
    buffer = C.create_string_buffer(_abi.MDB_COMM_ID_BYTES)
    if lib.mdb_comm_unique_id(buffer) != 0:
        raise HipError(lib.mdb_last_error().decode())
    return buffer.raw


def is_value_within_error_bound(eb, real_value, approximate_value):
    """models/mod.rs:53-77 through the C ABI (host arithmetic, no GPU)."""
    lib = _abi.load_hip_library()
    within = C.c_int32()
    if lib.mdb_is_value_within_error_bound(eb, real_value, approximate_value, C.byref(within)) != 0:
        raise HipError(lib.mdb_last_error().decode())
    return bool(within.value)


def are_compressed_timestamps_regular(data):
    """models/timestamps.rs:199-202 through the C ABI (host arithmetic, no GPU)."""
    lib = _abi.load_hip_library()
    data = bytes(data)
    regular = C.c_int32()
    if lib.mdb_are_compressed_timestamps_regular(data, len(data), C.byref(regular)) != 0:
        raise HipError(lib.mdb_last_error().decode())
    return bool(regular.value)


class Context:
    def __init__(self, device=0):
        self.lib = _abi.load_hip_library()
        self.handle = C.c_void_p()
        if self.lib.mdb_init(int(device), C.byref(self.handle)) != 0:
            raise HipError(self.lib.mdb_last_error().decode())
        self.device = device
        self.last_call_seconds = 0.0

    def _check(self, code):
        if code != 0:
            raise HipError(self.lib.mdb_last_error().decode())

    def close(self):
        if self.handle:
            self.lib.mdb_close(self.handle)
            self.handle = C.c_void_p()

    def clone(self):
        """Another context on the same device (mdb_clone): its own stream and scratch."""
        other = Context.__new__(Context)
        other.lib, other.device, other.handle = self.lib, self.device, C.c_void_p()
        self._check(self.lib.mdb_clone(self.handle, C.byref(other.handle)))
        return other

    def device_info(self):
        name = C.create_string_buffer(256)
        cus, hbm = C.c_int32(), C.c_uint64()
        self._check(self.lib.mdb_device_info(self.handle, name, 256, C.byref(cus), C.byref(hbm)))
        return {"name": name.value.decode(), "compute_units": cus.value, "hbm_bytes": hbm.value}

    def set_stream(self, hip_stream):
        self._check(self.lib.mdb_set_stream(self.handle, C.c_void_p(hip_stream)))

    def trim(self):
        """Give back the scratch / staging memory the context has grown; returns the device bytes."""
        released = C.c_uint64()
        self._check(self.lib.mdb_trim(self.handle, C.byref(released)))
        return released.value

    def set_scratch_limit(self, nbytes):
        """Device scratch beyond `nbytes` is given back after every call (0: keep everything)."""
        self._check(self.lib.mdb_set_scratch_limit(self.handle, C.c_uint64(nbytes)))

    # ---- device memory -----------------------------------------------------------------------

    def dev_alloc(self, nbytes):
        pointer = C.c_void_p()
        self._check(self.lib.mdb_dev_alloc(self.handle, int(nbytes), C.byref(pointer)))
        return pointer.value

    def dev_free(self, pointer):
        self._check(self.lib.mdb_dev_free(self.handle, C.c_void_p(pointer)))

    def upload_array(self, array):
        array = np.ascontiguousarray(array)
        pointer = self.dev_alloc(max(array.nbytes, 1))
        self._check(self.lib.mdb_dev_upload(self.handle, C.c_void_p(pointer),
                                            array.ctypes.data_as(C.c_void_p), array.nbytes))
        return pointer

    def download_array(self, pointer, count, dtype, offset_elements=0):
        out = np.empty(count, dtype=dtype)
        source = pointer + offset_elements * out.itemsize
        self._check(self.lib.mdb_dev_download(self.handle, out.ctypes.data_as(C.c_void_p),
                                              C.c_void_p(source), out.nbytes))
        return out

    def sync(self):
        self._check(self.lib.mdb_dev_sync(self.handle))

    def upload_segments(self, batch):
        seg = batch.as_c()
        out = C.POINTER(_abi.SegmentsOwnedC)()
        self._check(self.lib.mdb_segments_upload(self.handle, C.byref(seg), C.byref(out)))
        return DeviceSegments(self, out)

    # ---- grid ------------------------------------------------------------------------------------

    def grid_count(self, batch):
        seg = batch.as_c()
        n_out = C.c_uint64()
        self._check(self.lib.mdb_grid_count(self.handle, C.byref(seg), C.byref(n_out)))
        return n_out.value

    def grid_batch(self, batch, cap=None):
        """Returns (timestamps i64[], values f32[], rows_per_segment u32[], metrics dict)."""
        seg = batch.as_c()
        if cap is None:
            cap = self.grid_count(batch)
        out_ts = np.empty(cap, dtype=np.int64)
        out_val = np.empty(cap, dtype=np.float32)
        rows = np.empty(len(batch), dtype=np.uint32)
        n_out = C.c_uint64()
        metrics = _abi.GridMetricsC()
        self._check(self.lib.mdb_grid_batch(
            self.handle, C.byref(seg), out_ts.ctypes.data_as(C.c_void_p),
            out_val.ctypes.data_as(C.c_void_p), rows.ctypes.data_as(C.c_void_p), cap,
            C.byref(n_out), C.byref(metrics)))
        return out_ts[: n_out.value], out_val[: n_out.value], rows, metrics.as_dict()

    def grid_batch_owned(self, batch, time_range=None, copy=True, values_only=False):
        """One-call grid into page-locked memory owned by the library (mdb_grid_batch_owned).
        Returns (timestamps, values, rows_per_segment, metrics); the arrays are copies unless
        copy=False, in which case they alias the library's buffer and `release()` (5th item) must
        be called when done. values_only=True skips the timestamps (None is returned for them)."""
        seg = batch.as_c()
        out = C.POINTER(_abi.GridResultC)()
        has_range = time_range is not None
        t_lo, t_hi = time_range if has_range else (0, 0)
        flags = (1 if has_range else 0) | (2 if values_only else 0)
        self._check(self.lib.mdb_grid_batch_owned(self.handle, C.byref(seg), flags, t_lo, t_hi, 0,
                                                  C.byref(out)))
        result = out.contents
        n, n_segments = int(result.n), int(result.n_segments)

        def view(pointer, count, dtype):
            if count == 0:
                return np.zeros(0, dtype=dtype)
            buffer = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(pointer)
            return np.frombuffer(buffer, dtype=dtype)

        ts = None if values_only else view(result.timestamps, n, np.int64)
        values = view(result.values, n, np.float32)
        rows = view(result.rows_per_segment, n_segments, np.uint32)
        metrics = result.metrics.as_dict()
        release = lambda: self.lib.mdb_grid_result_free(out)
        if copy:
            ts = None if ts is None else ts.copy()
            values, rows = values.copy(), rows.copy()
            release()
            return ts, values, rows, metrics
        return ts, values, rows, metrics, release

    def grid_submit(self, batches, tag_views=None, tag_buffer_shifts=None, time_range=None,
                    values_only=False, reserve_front=0):
        """mdb_grid_submit: one or several SegmentBatches through one launch, on a worker thread of the
        library. tag_views: per batch, a list of (n, 16)-byte uint8 arrays (one per tag column) with one
        view per segment; tag_buffer_shifts: per batch, one int per tag column. Returns a GridTicket."""
        batches = list(batches)
        n_tags = len(tag_views[0]) if tag_views else 0
        inputs = (_abi.GridInputC * len(batches))()
        keep = [batches]
        for b, batch in enumerate(batches):
            inputs[b].segments = batch.as_c()
            if n_tags:
                views = [np.ascontiguousarray(v, dtype=np.uint8) for v in tag_views[b]]
                pointers = (C.c_void_p * n_tags)(*[v.ctypes.data for v in views])
                shifts = (C.c_int32 * n_tags)(*(tag_buffer_shifts[b] if tag_buffer_shifts else [0] * n_tags))
                inputs[b].tag_views = pointers
                inputs[b].tag_buffer_shift = shifts
                keep += [views, pointers, shifts]
        has_range = time_range is not None
        t_lo, t_hi = time_range if has_range else (0, 0)
        request = _abi.GridRequestC((1 if has_range else 0) | (2 if values_only else 0), n_tags, t_lo, t_hi,
                                    reserve_front)
        ticket = C.c_void_p()
        self._check(self.lib.mdb_grid_submit(self.handle, inputs, len(batches), C.byref(request),
                                             C.byref(ticket)))
        return GridTicket(self, ticket, keep, n_tags, values_only)

    def replicate_views(self, views, rows_per_segment, buffer_shift=0):
        """mdb_replicate_views: views (n, 16) uint8 -> (sum(rows), 16) uint8."""
        views = np.ascontiguousarray(views, dtype=np.uint8).reshape(-1, 16)
        rows = np.ascontiguousarray(rows_per_segment, dtype=np.uint32)
        out = np.empty((int(rows.sum()), 16), dtype=np.uint8)
        self._check(self.lib.mdb_replicate_views(views.ctypes.data_as(C.c_void_p), rows.ctypes.data_as(C.c_void_p),
                                                 len(rows), buffer_shift, out.ctypes.data_as(C.c_void_p), len(out)))
        return out

    def grid_batch_range(self, batch, t_lo, t_hi):
        """grid() with the predicate t_lo <= timestamp <= t_hi pushed down."""
        seg = batch.as_c()
        n_out = C.c_uint64()
        self._check(self.lib.mdb_grid_count_range(self.handle, C.byref(seg), t_lo, t_hi,
                                                  C.byref(n_out)))
        cap = n_out.value
        out_ts = np.empty(cap, dtype=np.int64)
        out_val = np.empty(cap, dtype=np.float32)
        rows = np.empty(len(batch), dtype=np.uint32)
        metrics = _abi.GridMetricsC()
        self._check(self.lib.mdb_grid_batch_range(
            self.handle, C.byref(seg), t_lo, t_hi, out_ts.ctypes.data_as(C.c_void_p),
            out_val.ctypes.data_as(C.c_void_p), rows.ctypes.data_as(C.c_void_p), cap,
            C.byref(n_out), C.byref(metrics)))
        return out_ts[: n_out.value], out_val[: n_out.value], rows, metrics.as_dict()

    def grid_batch_range_dev(self, dev_segments, t_lo, t_hi, out_ts_ptr, out_val_ptr, cap,
                             rows_ptr=None):
        n_out = C.c_uint64()
        metrics = _abi.GridMetricsC()
        self._check(self.lib.mdb_grid_batch_range_dev(
            self.handle, C.byref(dev_segments.seg), t_lo, t_hi, C.c_void_p(out_ts_ptr),
            C.c_void_p(out_val_ptr), C.c_void_p(rows_ptr), cap, C.byref(n_out), C.byref(metrics)))
        return n_out.value, metrics.as_dict()

    def grid_count_range_dev(self, dev_segments, t_lo, t_hi):
        n_out = C.c_uint64()
        self._check(self.lib.mdb_grid_count_range_dev(self.handle, C.byref(dev_segments.seg), t_lo,
                                                      t_hi, C.byref(n_out)))
        return n_out.value

    def grid_count_dev(self, dev_segments):
        n_out = C.c_uint64()
        self._check(self.lib.mdb_grid_count_dev(self.handle, C.byref(dev_segments.seg),
                                                C.byref(n_out)))
        return n_out.value

    def grid_batch_dev(self, dev_segments, out_ts_ptr, out_val_ptr, cap, rows_ptr=None):
        n_out = C.c_uint64()
        metrics = _abi.GridMetricsC()
        self._check(self.lib.mdb_grid_batch_dev(
            self.handle, C.byref(dev_segments.seg), C.c_void_p(out_ts_ptr), C.c_void_p(out_val_ptr),
            C.c_void_p(rows_ptr), cap, C.byref(n_out), C.byref(metrics)))
        return n_out.value, metrics.as_dict()

    def grid_resident(self, dev_segments, time_range=None):
        """grid() of a batch that is resident on the device into device columns, downloaded for the tests:
        (timestamps, values). The path a server with its segments in HBM takes (mdb_grid_batch[_range]_dev)."""
        if time_range is None:
            n = self.grid_count_dev(dev_segments)
        else:
            n = self.grid_count_range_dev(dev_segments, *time_range)
        out_ts, out_val = self.dev_alloc(8 * max(n, 1)), self.dev_alloc(4 * max(n, 1))
        try:
            if time_range is None:
                produced, _ = self.grid_batch_dev(dev_segments, out_ts, out_val, n)
            else:
                produced, _ = self.grid_batch_range_dev(dev_segments, time_range[0], time_range[1], out_ts, out_val, n)
            assert produced == n
            return self.download_array(out_ts, n, np.int64), self.download_array(out_val, n, np.float32)
        finally:
            self.dev_free(out_ts)
            self.dev_free(out_val)

    # ---- aggregates ------------------------------------------------------------------------------

    def agg_batch(self, batch, which_mask, state=None):
        seg = batch.as_c()
        state = state or _abi.AggStateC.fresh()
        self._check(self.lib.mdb_agg_batch(self.handle, C.byref(seg), which_mask, C.byref(state)))
        return state

    def agg_batch_list(self, batches, which_mask, state=None):
        """Several host batches folded as one (mdb_agg_batch_list): what an accumulator that has gathered the batches
        of a run of update_batch calls passes."""
        views = [batch.as_c() for batch in batches]
        pointers = (C.POINTER(_abi.SegmentsC) * len(views))(*[C.pointer(view) for view in views])
        state = state or _abi.AggStateC.fresh()
        self._check(self.lib.mdb_agg_batch_list(self.handle, pointers, len(views), which_mask, C.byref(state)))
        return state

    def agg_batch_range(self, batch, t_lo, t_hi, which_mask, state=None):
        seg = batch.as_c()
        state = state or _abi.AggStateC.fresh()
        self._check(self.lib.mdb_agg_batch_range(self.handle, C.byref(seg), t_lo, t_hi, which_mask,
                                                 C.byref(state)))
        return state

    def agg_batch_range_list(self, batches, t_lo, t_hi, which_mask, state=None):
        """Several host batches folded as one under a time range (mdb_agg_batch_range_list): what the accumulators
        of a ranged query pass (rust/patches/0002)."""
        views = [batch.as_c() for batch in batches]
        pointers = (C.POINTER(_abi.SegmentsC) * len(views))(*[C.pointer(view) for view in views])
        state = state or _abi.AggStateC.fresh()
        self._check(self.lib.mdb_agg_batch_range_list(self.handle, pointers, len(views), t_lo, t_hi, which_mask,
                                                      C.byref(state)))
        return state

    def agg_batch_dev(self, dev_segments, which_mask, state=None):
        state = state or _abi.AggStateC.fresh()
        self._check(self.lib.mdb_agg_batch_dev(self.handle, C.byref(dev_segments.seg), which_mask,
                                               C.byref(state)))
        return state

    def agg_batch_range_dev(self, dev_segments, t_lo, t_hi, which_mask, state=None):
        state = state or _abi.AggStateC.fresh()
        self._check(self.lib.mdb_agg_batch_range_dev(self.handle, C.byref(dev_segments.seg), t_lo,
                                                     t_hi, which_mask, C.byref(state)))
        return state

    # ---- fit -------------------------------------------------------------------------------------

    def compress_chunks(self, timestamps, values, chunk_offsets, eb):
        ts = np.ascontiguousarray(timestamps, dtype=np.int64)
        v = np.ascontiguousarray(values, dtype=np.float32)
        if len(ts) != len(v):
            # compression.rs:202-206
            raise HipError(
                "Uncompressed timestamps and uncompressed values have different lengths.")
        offsets = np.ascontiguousarray(chunk_offsets, dtype=np.uint64)
        out = C.POINTER(_abi.SegmentsOwnedC)()
        started = time.perf_counter()
        code = self.lib.mdb_compress_chunks(
            self.handle, ts.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p),
            offsets.ctypes.data_as(C.c_void_p), len(offsets) - 1, eb, C.byref(out))
        self.last_call_seconds = time.perf_counter() - started
        self._check(code)
        try:
            return SegmentBatch.from_owned(out)
        finally:
            self.lib.mdb_segments_free(out)

    def compress_chunk_list(self, chunks, eb):
        """mdb_compress_chunk_list: chunks = [(timestamps, values), ...] lying wherever they lie."""
        arrays = [(np.ascontiguousarray(ts, dtype=np.int64), np.ascontiguousarray(v, dtype=np.float32))
                  for ts, v in chunks]
        for ts, v in arrays:
            if len(ts) != len(v):
                raise HipError("Uncompressed timestamps and uncompressed values have different lengths.")
        # (an array that was contiguous already is passed as it is: chunks that share a timestamp array
        # keep sharing it)
        table = (_abi.ChunkC * max(len(arrays), 1))(*[_abi.ChunkC(ts.ctypes.data, v.ctypes.data, len(v))
                                                      for ts, v in arrays])
        out = C.POINTER(_abi.SegmentsOwnedC)()
        started = time.perf_counter()
        code = self.lib.mdb_compress_chunk_list(self.handle, table, len(arrays), eb, C.byref(out))
        self.last_call_seconds = time.perf_counter() - started  # (the library call alone, for bench.py)
        self._check(code)
        try:
            return SegmentBatch.from_owned(out)
        finally:
            self.lib.mdb_segments_free(out)

    def try_compress_univariate_time_series(self, timestamps, values, eb):
        """compression.rs:191-275 for one sorted series."""
        return self.compress_chunks(timestamps, values, [0, len(values)], eb)

    def compress_chunks_dev(self, ts_ptr, values_ptr, chunk_offsets_ptr, n_chunks, eb,
                            regular_start=0, regular_interval=0, series_first_index_ptr=None):
        out = C.POINTER(_abi.SegmentsOwnedC)()
        self._check(self.lib.mdb_compress_chunks_dev(
            self.handle, C.c_void_p(ts_ptr), C.c_void_p(values_ptr), C.c_void_p(chunk_offsets_ptr),
            n_chunks, eb, regular_start, regular_interval, C.c_void_p(series_first_index_ptr),
            C.byref(out)))
        return DeviceSegments(self, out)

    def try_split_and_compress_univariate_time_series(self, timestamps, field_values, error_bounds):
        """compression.rs:147-179: one sorted series, several field columns sharing its timestamps,
        one error bound per field. Returns one SegmentBatch per field."""
        ts = np.ascontiguousarray(timestamps, dtype=np.int64)
        fields = [np.ascontiguousarray(v, dtype=np.float32) for v in field_values]
        for v in fields:
            if len(v) != len(ts):
                raise HipError(
                    "Uncompressed timestamps and uncompressed values have different lengths.")
        n_fields = len(fields)
        pointers = (C.c_void_p * max(n_fields, 1))(*[v.ctypes.data for v in fields])
        bounds = (_abi.ErrorBoundC * max(n_fields, 1))(*error_bounds)
        out = (C.POINTER(_abi.SegmentsOwnedC) * max(n_fields, 1))()
        self._check(self.lib.mdb_split_and_compress_univariate(
            self.handle, ts.ctypes.data_as(C.c_void_p), pointers, bounds, n_fields, len(ts), out))
        batches = []
        for f in range(n_fields):
            try:
                batches.append(SegmentBatch.from_owned(out[f]))
            finally:
                self.lib.mdb_segments_free(out[f])
        return batches

    def validate_segments_dev(self, dev_segments):
        """Raises HipError if an out-of-line view of a device batch points outside its buffers."""
        seg = dev_segments.seg if hasattr(dev_segments, "seg") else dev_segments
        self._check(self.lib.mdb_segments_validate_dev(self.handle, C.byref(seg)))

    # ---- multi-GPU: the final aggregate merge over RCCL ------------------------------------------

    def comm_init(self, rank, world, unique_id):
        """ncclCommInitRank on this context's device (collective). `unique_id`: the 128 bytes of
        `comm_unique_id()` made by ONE rank and handed to the others."""
        buffer = C.create_string_buffer(bytes(unique_id), _abi.MDB_COMM_ID_BYTES)
        self._check(self.lib.mdb_comm_init(self.handle, rank, world, buffer))

    def comm_close(self):
        self._check(self.lib.mdb_comm_close(self.handle))

    def agg_all_reduce(self, state):
        """Merge the partial aggregate states of all ranks (one 32-byte all-gather over RCCL + a
        rank-ordered fold). Returns (merged state, ranks seen)."""
        merged = _abi.AggStateC(state.sum, state.count, state.min, state.max)
        seen = C.c_int32()
        self._check(self.lib.mdb_agg_all_reduce(self.handle, C.byref(merged), C.byref(seen)))
        return merged, seen.value

    def synth_values_dev(self, out_ptr, first_series, n_series, n_per_series,
                         seed=0x4D44425F52454631):
        self._check(self.lib.mdb_synth_values_dev(self.handle, C.c_void_p(out_ptr), first_series,
                                                  n_series, n_per_series, seed))

    # ---- measurement -----------------------------------------------------------------------------

    def profile_enable(self, enabled=True):
        self._check(self.lib.mdb_profile_enable(self.handle, int(enabled)))

    def profile_reset(self):
        self._check(self.lib.mdb_profile_reset(self.handle))

    def profile(self):
        """{kernel name: (launches, total_ms)} since the last reset."""
        names = C.create_string_buffer(4096)
        self._check(self.lib.mdb_profile_names(self.handle, names, 4096))
        out = {}
        for name in filter(None, names.value.decode().split("\n")):
            launches, total_ms = C.c_uint64(), C.c_double()
            self._check(self.lib.mdb_profile_get(self.handle, name.encode(), C.byref(launches),
                                                 C.byref(total_ms)))
            out[name] = (launches.value, total_ms.value)
        return out
