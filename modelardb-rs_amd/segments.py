"""Host-side batches of compressed segments in Arrow layout.

A ``SegmentBatch`` holds the columns of QUERY_COMPRESSED_SCHEMA
(crates/modelardb_types/src/schemas.rs:40-52 in the reference) as numpy arrays: three primitive
columns plus min/max and three BinaryView columns (16-byte views + variadic data buffers). It
converts to and from ``pyarrow.RecordBatch`` without copying and exposes the ``mdb_segments`` C
struct whose pointers go straight into those buffers - which is exactly what the Rust shim of
INTEGRATION.md passes.
"""

import ctypes as C

import numpy as np

from . import _abi

SEGMENT_COLUMN_NAMES = (
    "model_type_id", "start_time", "end_time", "timestamps", "min_value", "max_value",
    "values", "residuals", "error",
)


class BinaryViewColumn:
    """An Arrow BinaryViewArray: ``views`` is uint8[n, 16], ``buffers`` a list of uint8 arrays."""

    def __init__(self, views, buffers):
        self.views = np.ascontiguousarray(views, dtype=np.uint8).reshape(-1, 16)
        self.buffers = [np.ascontiguousarray(b, dtype=np.uint8) for b in buffers]

    def __len__(self):
        return self.views.shape[0]

    @classmethod
    def from_bytes_list(cls, items):
        """Build views the way arrow's BinaryViewBuilder does: <= 12 bytes inline, else buffer 0."""
        views = np.zeros((len(items), 16), dtype=np.uint8)
        data = bytearray()
        for i, item in enumerate(items):
            item = bytes(item)
            views[i, 0:4] = np.frombuffer(np.int32(len(item)).tobytes(), dtype=np.uint8)
            if len(item) <= 12:
                views[i, 4:4 + len(item)] = np.frombuffer(item, dtype=np.uint8)
            else:
                views[i, 4:8] = np.frombuffer(item[:4], dtype=np.uint8)
                views[i, 8:12] = np.frombuffer(np.int32(0).tobytes(), dtype=np.uint8)
                views[i, 12:16] = np.frombuffer(np.int32(len(data)).tobytes(), dtype=np.uint8)
                data += item
        buffers = [np.frombuffer(bytes(data), dtype=np.uint8)] if data else []
        return cls(views, buffers)

    def lengths(self):
        return self.views[:, 0:4].copy().view(np.int32).reshape(-1)

    def value(self, i):
        length, _, index, offset = (int(word) for word in self.views[i].view(np.int32))
        if length <= 12:
            return self.views[i, 4:4 + length].tobytes()
        return self.buffers[index][offset:offset + length].tobytes()

    def to_bytes_list(self):
        words = self.views.view(np.int32).reshape(-1, 4).tolist()
        inline = self.views[:, 4:16].tobytes()
        buffers = [memoryview(b) for b in self.buffers]
        return [inline[12 * i:12 * i + length] if length <= 12 else bytes(buffers[index][offset:offset + length])
                for i, (length, _, index, offset) in enumerate(words)]

    def take(self, indices):
        return BinaryViewColumn(self.views[indices], self.buffers)

    def to_arrow(self):
        import pyarrow as pa
        buffers = [None, pa.py_buffer(self.views)] + [pa.py_buffer(b) for b in self.buffers]
        return pa.Array.from_buffers(pa.binary_view(), len(self), buffers)

    @classmethod
    def from_arrow(cls, array):
        if array.offset != 0:
            array = array.take(np.arange(len(array)))
        buffers = array.buffers()
        views = np.frombuffer(buffers[1], dtype=np.uint8)[: 16 * len(array)].reshape(-1, 16)
        data = [np.frombuffer(b, dtype=np.uint8) for b in buffers[2:]]
        return cls(views, data)

    def as_c(self, keep_alive):
        col = _abi.BinViewColC()
        col.views = self.views.ctypes.data
        n = len(self.buffers)
        pointers = (C.c_void_p * max(n, 1))(*[b.ctypes.data for b in self.buffers])
        sizes = (C.c_int64 * max(n, 1))(*[b.size for b in self.buffers])
        keep_alive.extend([pointers, sizes, self.views, self.buffers])
        col.buffers = C.cast(pointers, C.POINTER(C.c_void_p))
        col.buffer_sizes = C.cast(sizes, C.POINTER(C.c_int64))
        col.n_buffers = n
        return col


class SegmentBatch:
    """The nine segment columns of one RecordBatch (tags and field_column are the caller's)."""

    def __init__(self, model_type_id, start_time, end_time, timestamps, min_value, max_value,
                 values, residuals, error=None, chunk_index=None):
        self.model_type_id = np.ascontiguousarray(model_type_id, dtype=np.int8)
        self.start_time = np.ascontiguousarray(start_time, dtype=np.int64)
        self.end_time = np.ascontiguousarray(end_time, dtype=np.int64)
        self.timestamps = timestamps
        self.min_value = np.ascontiguousarray(min_value, dtype=np.float32)
        self.max_value = np.ascontiguousarray(max_value, dtype=np.float32)
        self.values = values
        self.residuals = residuals
        n = len(self.model_type_id)
        self.error = (np.full(n, np.nan, dtype=np.float32) if error is None
                      else np.ascontiguousarray(error, dtype=np.float32))
        self.chunk_index = (None if chunk_index is None
                            else np.ascontiguousarray(chunk_index, dtype=np.uint32))
        self._keep_alive = []

    def __len__(self):
        return len(self.model_type_id)

    @classmethod
    def from_rows(cls, rows):
        """rows: iterable of (model_type_id, start, end, timestamps, min, max, values, residuals)."""
        rows = list(rows)
        col = lambda i: [r[i] for r in rows]
        return cls(col(0), col(1), col(2), BinaryViewColumn.from_bytes_list(col(3)), col(4), col(5),
                   BinaryViewColumn.from_bytes_list(col(6)), BinaryViewColumn.from_bytes_list(col(7)))

    def rows(self):
        ts, vals, res = (c.to_bytes_list() for c in (self.timestamps, self.values, self.residuals))
        return [(int(self.model_type_id[i]), int(self.start_time[i]), int(self.end_time[i]), ts[i],
                 float(self.min_value[i]), float(self.max_value[i]), vals[i], res[i])
                for i in range(len(self))]

    def identical(self, other):
        """Every column of the two batches bit for bit: ids, times, payload bytes, and min / max as BIT PATTERNS
        (as Python floats -0.0 equals 0.0 and a NaN never equals itself)."""
        return (len(self) == len(other)
                and np.array_equal(self.model_type_id, other.model_type_id)
                and np.array_equal(self.start_time, other.start_time)
                and np.array_equal(self.end_time, other.end_time)
                and np.array_equal(self.min_value.view(np.uint32), other.min_value.view(np.uint32))
                and np.array_equal(self.max_value.view(np.uint32), other.max_value.view(np.uint32))
                and self.timestamps.to_bytes_list() == other.timestamps.to_bytes_list()
                and self.values.to_bytes_list() == other.values.to_bytes_list()
                and self.residuals.to_bytes_list() == other.residuals.to_bytes_list())

    def take(self, indices):
        indices = np.asarray(indices)
        return SegmentBatch(self.model_type_id[indices], self.start_time[indices],
                            self.end_time[indices], self.timestamps.take(indices),
                            self.min_value[indices], self.max_value[indices],
                            self.values.take(indices), self.residuals.take(indices),
                            self.error[indices],
                            None if self.chunk_index is None else self.chunk_index[indices])

    def slice(self, start, stop):
        return self.take(np.arange(start, stop))

    def as_c(self):
        """The ``mdb_segments`` struct borrowing this batch's buffers (kept alive by ``self``)."""
        keep = []
        seg = _abi.SegmentsC()
        seg.n = len(self)
        seg.model_type_id = self.model_type_id.ctypes.data
        seg.start_time = self.start_time.ctypes.data
        seg.end_time = self.end_time.ctypes.data
        seg.timestamps = self.timestamps.as_c(keep)
        seg.min_value = self.min_value.ctypes.data
        seg.max_value = self.max_value.ctypes.data
        seg.values = self.values.as_c(keep)
        seg.residuals = self.residuals.as_c(keep)
        self._keep_alive = keep
        return seg

    def to_arrow(self):
        """RecordBatch with QUERY_COMPRESSED_SCHEMA (schemas.rs:40-52)."""
        import pyarrow as pa
        ts_type = pa.timestamp("us")
        arrays = [
            pa.array(self.model_type_id, type=pa.int8()),
            pa.array(self.start_time, type=pa.int64()).cast(ts_type),
            pa.array(self.end_time, type=pa.int64()).cast(ts_type),
            self.timestamps.to_arrow(),
            pa.array(self.min_value, type=pa.float32()),
            pa.array(self.max_value, type=pa.float32()),
            self.values.to_arrow(),
            self.residuals.to_arrow(),
            pa.array(self.error, type=pa.float32()),
        ]
        return pa.RecordBatch.from_arrays(arrays, names=list(SEGMENT_COLUMN_NAMES))

    @classmethod
    def from_arrow(cls, batch):
        import pyarrow as pa
        column = lambda name: batch.column(batch.schema.get_field_index(name))
        i64 = lambda name: column(name).cast(pa.int64()).to_numpy(zero_copy_only=False)
        return cls(column("model_type_id").to_numpy(zero_copy_only=False), i64("start_time"),
                   i64("end_time"), BinaryViewColumn.from_arrow(column("timestamps")),
                   column("min_value").to_numpy(zero_copy_only=False),
                   column("max_value").to_numpy(zero_copy_only=False),
                   BinaryViewColumn.from_arrow(column("values")),
                   BinaryViewColumn.from_arrow(column("residuals")),
                   column("error").to_numpy(zero_copy_only=False))

    @classmethod
    def from_owned(cls, owned_ptr):
        """Copy a host-resident ``mdb_segments_owned`` into numpy arrays."""
        owned = owned_ptr.contents
        if owned.on_device:
            raise ValueError("from_owned() needs a host batch; download it first")
        seg = owned.seg
        n = int(seg.n)

        def array(pointer, dtype, count):
            if count == 0 or not pointer:
                return np.zeros(0, dtype=dtype)
            size = count * np.dtype(dtype).itemsize
            return np.frombuffer(C.string_at(pointer, size), dtype=dtype).copy()

        def column(col):
            views = array(col.views, np.uint8, 16 * n).reshape(-1, 16)
            buffers = [array(col.buffers[i], np.uint8, int(col.buffer_sizes[i]))
                       for i in range(col.n_buffers)]
            return BinaryViewColumn(views, buffers)

        return cls(array(seg.model_type_id, np.int8, n), array(seg.start_time, np.int64, n),
                   array(seg.end_time, np.int64, n), column(seg.timestamps),
                   array(seg.min_value, np.float32, n), array(seg.max_value, np.float32, n),
                   column(seg.values), column(seg.residuals), array(owned.error, np.float32, n),
                   array(owned.chunk_index, np.uint32, n) if owned.chunk_index else None)

    @classmethod
    def concat(cls, batches):
        batches = list(batches)
        rows = [row for b in batches for row in b.rows()]
        out = cls.from_rows(rows)
        if batches and all(b.chunk_index is not None for b in batches):
            out.chunk_index = np.concatenate([b.chunk_index for b in batches])
        return out


def error_bound(kind, value=0.0):
    """ErrorBound::{Lossless, Absolute, Relative} (crates/modelardb_types/src/types.rs:299-335)."""
    kinds = {"lossless": _abi.MDB_EB_LOSSLESS, "absolute": _abi.MDB_EB_ABSOLUTE,
             "relative": _abi.MDB_EB_RELATIVE}
    kind = kinds[kind] if isinstance(kind, str) else kind
    if kind == _abi.MDB_EB_ABSOLUTE and not (np.isfinite(value) and value > 0.0):
        raise ValueError("An absolute error bound must be a positive finite value.")
    if kind == _abi.MDB_EB_RELATIVE and not (0.0 < value <= 100.0):
        raise ValueError("A relative error bound must be a positive value that is at most 100.0%.")
    return _abi.ErrorBoundC(kind, value)
