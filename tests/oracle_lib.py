"""ctypes wrapper of the CPU oracle (oracle/libmdb_oracle.so). Test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""

import ctypes as C
import os
import subprocess

import numpy as np

import modelardb_rs_amd as mdb
from modelardb_rs_amd import _abi

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(REPO_ROOT, "oracle")
# MDB_ORACLE_LIB selects another build of the oracle (e.g. the ASan/UBSan one: `make -C oracle asan`).
ORACLE_PATH = os.environ.get("MDB_ORACLE_LIB", os.path.join(ORACLE_DIR, "libmdb_oracle.so"))


class OracleError(RuntimeError):
    pass


class OraModelC(C.Structure):
    _fields_ = [
        ("model_type_id", C.c_int8),
        ("start_index", C.c_uint64),
        ("end_index", C.c_uint64),
        ("min_value", C.c_float),
        ("max_value", C.c_float),
        ("values", C.c_uint8 * 8),
        ("values_len", C.c_uint32),
        ("model_last_value", C.c_float),
        ("bytes_per_value", C.c_float),
    ]


_lib = None


def build():
    sources = [os.path.join(ORACLE_DIR, f) for f in ("mdb_oracle.cpp", "mdb_oracle.h")]
    sources.append(os.path.join(REPO_ROOT, "include", "mdb_format.h"))
    if os.path.exists(ORACLE_PATH) and all(
            os.path.getmtime(ORACLE_PATH) >= os.path.getmtime(s) for s in sources):
        return
    subprocess.run(["make", "-C", ORACLE_DIR, "libmdb_oracle.so"], check=True,
                   stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(ORACLE_PATH)
        _lib.ora_last_error.restype = C.c_char_p
        _lib.ora_maximum_allowed_deviation.restype = C.c_double
        _lib.ora_maximum_allowed_deviation.argtypes = [_abi.ErrorBoundC, C.c_double]
        _lib.ora_is_value_within_error_bound.argtypes = [_abi.ErrorBoundC, C.c_float, C.c_float]
        _lib.ora_swing_sum.restype = C.c_float
        _lib.ora_swing_sum.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_uint64, C.c_float,
                                       C.c_float, C.c_uint64]
        _lib.ora_segments_free.argtypes = [C.POINTER(_abi.SegmentsOwnedC)]
        _lib.ora_segments_free.restype = None
        # Prototypes for everything that takes the error bound struct by value.
        EB, P, U64 = _abi.ErrorBoundC, C.c_void_p, C.c_uint64
        _lib.ora_pmc_mean_fit.argtypes = [EB, P, U64, P, P, P]
        _lib.ora_swing_fit.argtypes = [EB, P, P, U64, P, P, P, P, P]
        _lib.ora_macaque_v_compress.argtypes = [EB, P, U64, C.c_int, C.c_float, P, U64, P, P, P, P, P, P]
        _lib.ora_fit_next_model.argtypes = [U64, EB, P, P, U64, P]
        _lib.ora_model_finish.argtypes = [P, EB, U64, P, P, U64, P]
        _lib.ora_compress_chunks.argtypes = [P, P, P, U64, EB, C.c_int, P]
    return _lib


def _check(code):
    if code != 0:
        raise OracleError(lib().ora_last_error().decode())


def _u8(data):
    array = np.frombuffer(bytes(data), dtype=np.uint8).copy()
    return array, array.ctypes.data_as(C.c_void_p), C.c_uint64(array.size)


def _f32(values):
    return np.ascontiguousarray(values, dtype=np.float32)


def _i64(values):
    return np.ascontiguousarray(values, dtype=np.int64)


def _ptr(array):
    return array.ctypes.data_as(C.c_void_p)


# ---- scalar helpers -------------------------------------------------------------------------

def is_value_within_error_bound(eb, real_value, approximate_value):
    return bool(lib().ora_is_value_within_error_bound(eb, C.c_float(real_value),
                                                      C.c_float(approximate_value)))


def maximum_allowed_deviation(eb, value):
    return lib().ora_maximum_allowed_deviation(eb, C.c_double(value))


# ---- bits -----------------------------------------------------------------------------------

def bits_write(items, finish_with_ones=False):
    """items: list of (bits, nbits)."""
    bits = np.array([b for b, _ in items], dtype=np.uint64)
    nbits = np.array([n for _, n in items], dtype=np.uint8)
    out = np.zeros(8 * len(items) + 8, dtype=np.uint8)
    out_len = C.c_uint64()
    _check(lib().ora_bits_write(_ptr(bits), _ptr(nbits), C.c_uint64(len(items)),
                                C.c_int(int(finish_with_ones)), _ptr(out), C.c_uint64(out.size),
                                C.byref(out_len)))
    return out[: out_len.value].tobytes()


def bits_read(data, widths):
    _, pointer, size = keep = _u8(data)
    nbits = np.array(widths, dtype=np.uint8)
    out = np.zeros(len(widths), dtype=np.uint64)
    remaining = C.c_uint64()
    _check(lib().ora_bits_read(pointer, size, _ptr(nbits), C.c_uint64(len(widths)), _ptr(out),
                               C.byref(remaining)))
    del keep
    return [int(v) for v in out], remaining.value


# ---- timestamps -----------------------------------------------------------------------------

def compress_residual_timestamps(timestamps):
    ts = _i64(timestamps)
    out = np.zeros(16 * len(ts) + 16, dtype=np.uint8)
    out_len = C.c_uint64()
    _check(lib().ora_compress_residual_timestamps(_ptr(ts), C.c_uint64(len(ts)), _ptr(out),
                                                  C.c_uint64(out.size), C.byref(out_len)))
    return out[: out_len.value].tobytes()


def decompress_all_timestamps(start_time, end_time, data, cap=1 << 20):
    keep = _u8(data)
    out = np.zeros(cap, dtype=np.int64)
    n_out = C.c_uint64()
    _check(lib().ora_decompress_all_timestamps(C.c_int64(start_time), C.c_int64(end_time), keep[1],
                                               keep[2], _ptr(out), C.c_uint64(cap),
                                               C.byref(n_out)))
    return out[: n_out.value].copy()


def are_compressed_timestamps_regular(data):
    keep = _u8(data)
    return bool(lib().ora_are_compressed_timestamps_regular(keep[1], keep[2]))


# ---- per-segment len / sum / grid -------------------------------------------------------------

def seg_len(start_time, end_time, timestamps):
    keep = _u8(timestamps)
    out = C.c_uint64()
    _check(lib().ora_len(C.c_int64(start_time), C.c_int64(end_time), keep[1], keep[2],
                         C.byref(out)))
    return out.value


def seg_sum(model_type_id, start_time, end_time, timestamps, min_value, max_value, values,
            residuals):
    t, v, r = _u8(timestamps), _u8(values), _u8(residuals)
    out = C.c_float()
    _check(lib().ora_sum(C.c_int8(model_type_id), C.c_int64(start_time), C.c_int64(end_time), t[1],
                         t[2], C.c_float(min_value), C.c_float(max_value), v[1], v[2], r[1], r[2],
                         C.byref(out)))
    return np.float32(out.value)


def seg_grid(model_type_id, start_time, end_time, timestamps, min_value, max_value, values,
             residuals, cap=1 << 20):
    t, v, r = _u8(timestamps), _u8(values), _u8(residuals)
    out_ts = np.zeros(cap, dtype=np.int64)
    out_val = np.zeros(cap, dtype=np.float32)
    n_out = C.c_uint64()
    _check(lib().ora_grid(C.c_int8(model_type_id), C.c_int64(start_time), C.c_int64(end_time), t[1],
                          t[2], C.c_float(min_value), C.c_float(max_value), v[1], v[2], r[1], r[2],
                          _ptr(out_ts), _ptr(out_val), C.c_uint64(cap), C.byref(n_out)))
    return out_ts[: n_out.value].copy(), out_val[: n_out.value].copy()


# ---- model types ----------------------------------------------------------------------------

def pmc_mean_fit(eb, values):
    v = _f32(values)
    n_fit, model, bpv = C.c_uint64(), C.c_float(), C.c_float()
    _check(lib().ora_pmc_mean_fit(eb, _ptr(v), C.c_uint64(len(v)), C.byref(n_fit), C.byref(model),
                                  C.byref(bpv)))
    return n_fit.value, np.float32(model.value), np.float32(bpv.value)


def swing_fit(eb, timestamps, values):
    ts, v = _i64(timestamps), _f32(values)
    n_fit, first, last, bpv = C.c_uint64(), C.c_float(), C.c_float(), C.c_float()
    bounds = (C.c_double * 4)()
    _check(lib().ora_swing_fit(eb, _ptr(ts), _ptr(v), C.c_uint64(len(v)), C.byref(n_fit),
                               C.byref(first), C.byref(last), C.byref(bpv), bounds))
    return (n_fit.value, np.float32(first.value), np.float32(last.value), np.float32(bpv.value),
            list(bounds))


def swing_sum(start_time, end_time, timestamps, first_value, last_value, residuals_length):
    keep = _u8(timestamps)
    return np.float32(lib().ora_swing_sum(C.c_int64(start_time), C.c_int64(end_time), keep[1],
                                          keep[2], C.c_float(first_value), C.c_float(last_value),
                                          C.c_uint64(residuals_length)))


def macaque_v_compress(eb, values, seed=None):
    """Returns (bytes, min, max, last_leading_zero_bits, last_trailing_zero_bits, last_value)."""
    v = _f32(values)
    out = np.zeros(8 * len(v) + 16, dtype=np.uint8)
    out_len = C.c_uint64()
    mn, mx, last = C.c_float(), C.c_float(), C.c_float()
    lz, tz = C.c_uint8(), C.c_uint8()
    _check(lib().ora_macaque_v_compress(eb, _ptr(v), C.c_uint64(len(v)),
                                        C.c_int(int(seed is not None)),
                                        C.c_float(0.0 if seed is None else seed), _ptr(out),
                                        C.c_uint64(out.size), C.byref(out_len), C.byref(mn),
                                        C.byref(mx), C.byref(lz), C.byref(tz), C.byref(last)))
    return (out[: out_len.value].tobytes(), np.float32(mn.value), np.float32(mx.value), lz.value,
            tz.value, np.float32(last.value))


def macaque_v_grid(data, n, seed=None):
    keep = _u8(data)
    out = np.zeros(n, dtype=np.float32)
    _check(lib().ora_macaque_v_grid(keep[1], keep[2], C.c_uint64(n), C.c_int(int(seed is not None)),
                                    C.c_float(0.0 if seed is None else seed), _ptr(out)))
    return out


def macaque_v_sum(data, n, seed=None):
    keep = _u8(data)
    out = C.c_float()
    _check(lib().ora_macaque_v_sum(keep[1], keep[2], C.c_uint64(n), C.c_int(int(seed is not None)),
                                   C.c_float(0.0 if seed is None else seed), C.byref(out)))
    return np.float32(out.value)


# ---- values column encodings -------------------------------------------------------------------

def encode_values_for_pmc_mean(min_value, max_value, rmin, rmax):
    out = (C.c_uint8 * 8)()
    out_len = C.c_uint64()
    _check(lib().ora_encode_values_for_pmc_mean(C.c_float(min_value), C.c_float(max_value),
                                                C.c_float(rmin), C.c_float(rmax), out,
                                                C.byref(out_len)))
    return bytes(out[: out_len.value])


def decode_values_for_pmc_mean(min_value, max_value, values):
    keep = _u8(values)
    out = C.c_float()
    _check(lib().ora_decode_values_for_pmc_mean(C.c_float(min_value), C.c_float(max_value), keep[1],
                                                keep[2], C.byref(out)))
    return np.float32(out.value)


def encode_values_for_swing(min_value, max_value, min_value_is_first, rmin, rmax):
    out = (C.c_uint8 * 8)()
    out_len = C.c_uint64()
    _check(lib().ora_encode_values_for_swing(C.c_float(min_value), C.c_float(max_value),
                                             C.c_int(int(min_value_is_first)), C.c_float(rmin),
                                             C.c_float(rmax), out, C.byref(out_len)))
    return bytes(out[: out_len.value])


def decode_values_for_swing(min_value, max_value, values):
    keep = _u8(values)
    first, last = C.c_float(), C.c_float()
    _check(lib().ora_decode_values_for_swing(C.c_float(min_value), C.c_float(max_value), keep[1],
                                             keep[2], C.byref(first), C.byref(last)))
    return np.float32(first.value), np.float32(last.value)


# ---- compression driver ----------------------------------------------------------------------

def fit_next_model(start_index, eb, timestamps, values):
    ts, v = _i64(timestamps), _f32(values)
    model = OraModelC()
    _check(lib().ora_fit_next_model(C.c_uint64(start_index), eb, _ptr(ts), _ptr(v),
                                    C.c_uint64(len(v)), C.byref(model)))
    return model


def _take_owned(pointer):
    try:
        return mdb.SegmentBatch.from_owned(pointer)
    finally:
        lib().ora_segments_free(pointer)


def model_finish(model, eb, residuals_end_index, timestamps, values):
    ts, v = _i64(timestamps), _f32(values)
    out = C.POINTER(_abi.SegmentsOwnedC)()
    _check(lib().ora_model_finish(C.byref(model), eb, C.c_uint64(residuals_end_index), _ptr(ts),
                                  _ptr(v), C.c_uint64(len(v)), C.byref(out)))
    return _take_owned(out)


def compress_chunks(timestamps, values, chunk_offsets, eb, n_threads=1):
    ts, v = _i64(timestamps), _f32(values)
    offsets = np.ascontiguousarray(chunk_offsets, dtype=np.uint64)
    if len(ts) != len(v):
        raise OracleError(
            "Uncompressed timestamps and uncompressed values have different lengths.")
    out = C.POINTER(_abi.SegmentsOwnedC)()
    _check(lib().ora_compress_chunks(_ptr(ts), _ptr(v), _ptr(offsets),
                                     C.c_uint64(len(offsets) - 1), eb, C.c_int(n_threads),
                                     C.byref(out)))
    return _take_owned(out)


def try_compress_univariate_time_series(timestamps, values, eb):
    """compression.rs:191-275 for one series."""
    return compress_chunks(timestamps, values, [0, len(values)], eb)


# ---- batch operators --------------------------------------------------------------------------

def grid_count(batch):
    seg = batch.as_c()
    n_out = C.c_uint64()
    _check(lib().ora_grid_count(C.byref(seg), C.byref(n_out)))
    return n_out.value


def grid_batch(batch, n_threads=1, timing=None):
    """Returns (timestamps, values, rows_per_segment, metrics dict). With `timing` (a dict) the
    output buffers are allocated and touched first and only the oracle call itself is timed."""
    import time
    seg = batch.as_c()
    cap = grid_count(batch)
    out_ts = np.zeros(cap, dtype=np.int64)
    out_val = np.zeros(cap, dtype=np.float32)
    rows = np.zeros(len(batch), dtype=np.uint32)
    if timing is not None:
        out_ts.fill(1)
        out_val.fill(1)
    n_out = C.c_uint64()
    started = time.perf_counter()
    if n_threads > 1:
        _check(lib().ora_grid_batch_mt(C.byref(seg), _ptr(out_ts), _ptr(out_val), C.c_uint64(cap),
                                       C.byref(n_out), C.c_int(n_threads)))
        if timing is not None:
            timing["seconds"] = time.perf_counter() - started
        return out_ts[: n_out.value], out_val[: n_out.value], None, None
    metrics = _abi.GridMetricsC()
    _check(lib().ora_grid_batch(C.byref(seg), _ptr(out_ts), _ptr(out_val), _ptr(rows),
                                C.c_uint64(cap), C.byref(n_out), C.byref(metrics)))
    if timing is not None:
        timing["seconds"] = time.perf_counter() - started
    return out_ts[: n_out.value], out_val[: n_out.value], rows, metrics.as_dict()


def grid_batch_timed(batch, n_threads, repetitions=3, pin=True):
    """The timed CPU-baseline leg: the per-row GridStream loop sharded over `n_threads` workers, each
    pinned to a CPU of its own (ora_set_thread_pinning). The outputs are allocated untouched; one
    untimed pass lets every worker first-touch the pages it writes (NUMA-local under Linux's default
    first-touch policy), then `repetitions` passes are timed. Returns (timestamps, values, seconds[])."""
    import time
    seg = batch.as_c()
    cap = grid_count(batch)
    out_ts = np.empty(cap, dtype=np.int64)
    out_val = np.empty(cap, dtype=np.float32)
    n_out = C.c_uint64()
    lib().ora_set_thread_pinning(1 if pin else 0)
    try:
        seconds = []
        for repetition in range(repetitions + 1):
            started = time.perf_counter()
            _check(lib().ora_grid_batch_mt(C.byref(seg), _ptr(out_ts), _ptr(out_val), C.c_uint64(cap),
                                           C.byref(n_out), C.c_int(max(n_threads, 1))))
            if repetition > 0:
                seconds.append(time.perf_counter() - started)
    finally:
        lib().ora_set_thread_pinning(0)
    return out_ts[: n_out.value], out_val[: n_out.value], seconds


def compress_chunks_timed(timestamps, values, chunk_offsets, eb, n_threads, repetitions=3, pin=True):
    """The timed CPU-baseline leg of the fitter: returns (segments of the last pass, seconds[])."""
    import time
    lib().ora_set_thread_pinning(1 if pin else 0)
    try:
        seconds, fitted = [], None
        for _ in range(repetitions):
            started = time.perf_counter()
            fitted = compress_chunks(timestamps, values, chunk_offsets, eb, n_threads=n_threads)
            seconds.append(time.perf_counter() - started)
    finally:
        lib().ora_set_thread_pinning(0)
    return fitted, seconds


def agg_batch(batch, which_mask, state=None):
    seg = batch.as_c()
    state = state or _abi.AggStateC.fresh()
    _check(lib().ora_agg_batch(C.byref(seg), C.c_uint32(which_mask), C.byref(state)))
    return state


def agg_batch_range(batch, t_lo, t_hi, which_mask, state=None):
    seg = batch.as_c()
    state = state or _abi.AggStateC.fresh()
    _check(lib().ora_agg_batch_range(C.byref(seg), C.c_int64(t_lo), C.c_int64(t_hi),
                                     C.c_uint32(which_mask), C.byref(state)))
    return state
