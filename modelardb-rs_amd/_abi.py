"""ctypes mirror of include/mdb_format.h and include/mdb.h, plus the loader of libmdb_hip.so.

The product path has no CPU fallback: if the HIP library is missing or cannot be loaded,
``load_hip_library`` raises. The structs here are shared with the oracle wrapper in ``tests/`` so
both sides consume exactly the same in-memory segment batches.
"""

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_HERE)
# MDB_HIP_LIBRARY selects another build of the same library (A/B timing of kernel variants).
HIP_LIBRARY_PATH = os.environ.get("MDB_HIP_LIBRARY", os.path.join(_HERE, "csrc", "libmdb_hip.so"))

MDB_PMC_MEAN_ID = 0
MDB_SWING_ID = 1
MDB_MACAQUE_V_ID = 2
MODEL_TYPE_NAMES = ("pmc_mean", "swing", "macaque_v")  # models/mod.rs:44

MDB_EB_LOSSLESS = 0
MDB_EB_ABSOLUTE = 1
MDB_EB_RELATIVE = 2

MDB_AGG_COUNT = 1
MDB_AGG_MIN = 2
MDB_AGG_MAX = 4
MDB_AGG_SUM = 8
MDB_AGG_AVG = 16

MDB_COMM_ID_BYTES = 128

F32_MAX = 3.4028234663852886e38


class ErrorBoundC(C.Structure):
    _fields_ = [("kind", C.c_int32), ("value", C.c_float)]


class BinViewColC(C.Structure):
    _fields_ = [
        ("views", C.c_void_p),
        ("buffers", C.POINTER(C.c_void_p)),
        ("buffer_sizes", C.POINTER(C.c_int64)),
        ("n_buffers", C.c_int32),
    ]


class SegmentsC(C.Structure):
    _fields_ = [
        ("n", C.c_uint64),
        ("model_type_id", C.c_void_p),
        ("start_time", C.c_void_p),
        ("end_time", C.c_void_p),
        ("timestamps", BinViewColC),
        ("min_value", C.c_void_p),
        ("max_value", C.c_void_p),
        ("values", BinViewColC),
        ("residuals", BinViewColC),
    ]


class GridMetricsC(C.Structure):
    _fields_ = [
        ("rows_created", C.c_uint64),
        ("rows_created_by_model_type", C.c_uint64 * 3),
        ("segments_with_residuals", C.c_uint64),
        ("segments_with_model_type", C.c_uint64 * 3),
        ("segments_regular", C.c_uint64),
        ("segments_irregular", C.c_uint64),
    ]

    def as_dict(self):
        out = {"rows_created": self.rows_created}
        for i, name in enumerate(MODEL_TYPE_NAMES):
            out[f"rows_created_by_{name}"] = self.rows_created_by_model_type[i]
        out["segments_with_residuals"] = self.segments_with_residuals
        for i, name in enumerate(MODEL_TYPE_NAMES):
            out[f"segments_with_{name}"] = self.segments_with_model_type[i]
        out["regular_segments"] = self.segments_regular
        out["irregular_segments"] = self.segments_irregular
        return out


class AggStateC(C.Structure):
    _fields_ = [
        ("sum", C.c_double),
        ("count", C.c_int64),
        ("min", C.c_float),
        ("max", C.c_float),
    ]

    @classmethod
    def fresh(cls):
        # model_simple_aggregates.rs:413,456: min starts at f32::MAX, max at f32::MIN.
        return cls(0.0, 0, F32_MAX, -F32_MAX)


class SegmentsOwnedC(C.Structure):
    _fields_ = [
        ("seg", SegmentsC),
        ("error", C.c_void_p),
        ("chunk_index", C.c_void_p),
        ("on_device", C.c_int32),
        ("priv_", C.c_void_p),
    ]


class GridResultC(C.Structure):
    _fields_ = [
        ("timestamps", C.c_void_p),
        ("values", C.c_void_p),
        ("rows_per_segment", C.c_void_p),
        ("n", C.c_uint64),
        ("n_segments", C.c_uint64),
        ("reserved_front", C.c_uint64),
        ("metrics", GridMetricsC),
        ("priv_", C.c_void_p),
    ]


class GridInputC(C.Structure):
    _fields_ = [
        ("segments", SegmentsC),
        ("tag_views", C.POINTER(C.c_void_p)),
        ("tag_buffer_shift", C.POINTER(C.c_int32)),
    ]


class GridRequestC(C.Structure):
    _fields_ = [
        ("flags", C.c_uint32),
        ("n_tag_columns", C.c_uint32),
        ("t_lo", C.c_int64),
        ("t_hi", C.c_int64),
        ("reserve_front", C.c_uint64),
    ]


class ChunkC(C.Structure):
    _fields_ = [("ts", C.c_void_p), ("values", C.c_void_p), ("n", C.c_uint64)]


_HIP_SYMBOLS = {
    # name: (restype, argtypes)
    "mdb_init": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "mdb_close": (C.c_int, [C.c_void_p]),
    "mdb_clone": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "mdb_last_error": (C.c_char_p, []),
    "mdb_version": (C.c_char_p, []),
    "mdb_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mdb_trim": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "mdb_set_scratch_limit": (C.c_int, [C.c_void_p, C.c_uint64]),
    "mdb_device_info": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint64, C.POINTER(C.c_int32),
                                  C.POINTER(C.c_uint64)]),
    "mdb_dev_alloc": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "mdb_dev_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mdb_dev_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "mdb_dev_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "mdb_dev_sync": (C.c_int, [C.c_void_p]),
    "mdb_segments_upload": (C.c_int, [C.c_void_p, C.POINTER(SegmentsC),
                                      C.POINTER(C.POINTER(SegmentsOwnedC))]),
    "mdb_segments_download": (C.c_int, [C.c_void_p, C.POINTER(SegmentsOwnedC),
                                        C.POINTER(C.POINTER(SegmentsOwnedC))]),
    "mdb_segments_free": (None, [C.POINTER(SegmentsOwnedC)]),
    "mdb_grid_count": (C.c_int, [C.c_void_p, C.POINTER(SegmentsC), C.POINTER(C.c_uint64)]),
    "mdb_grid_batch": (C.c_int, [C.c_void_p, C.POINTER(SegmentsC), C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                                 C.POINTER(GridMetricsC)]),
    "mdb_grid_count_dev": (C.c_int, [C.c_void_p, C.POINTER(SegmentsC), C.POINTER(C.c_uint64)]),
    "mdb_grid_batch_dev": (C.c_int, [C.c_void_p, C.POINTER(SegmentsC), C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                                     C.POINTER(GridMetricsC)]),
    "mdb_grid_count_range": (C.c_int, [C.c_void_p, C.POINTER(SegmentsC), C.c_int64, C.c_int64,
                                       C.POINTER(C.c_uint64)]),
    "mdb_grid_batch_range": (C.c_int, [C.c_void_p, C.POINTER(SegmentsC), C.c_int64, C.c_int64,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                       C.POINTER(C.c_uint64), C.POINTER(GridMetricsC)]),
    "mdb_grid_count_range_dev": (C.c_int, [C.c_void_p, C.POINTER(SegmentsC), C.c_int64, C.c_int64,
                                           C.POINTER(C.c_uint64)]),
    "mdb_grid_batch_range_dev": (C.c_int, [C.c_void_p, C.POINTER(SegmentsC), C.c_int64, C.c_int64,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                           C.POINTER(C.c_uint64), C.POINTER(GridMetricsC)]),
    "mdb_grid_batch_owned": (C.c_int, [C.c_void_p, C.POINTER(SegmentsC), C.c_uint32, C.c_int64,
                                       C.c_int64, C.c_uint64, C.POINTER(C.POINTER(GridResultC))]),
    "mdb_grid_result_free": (None, [C.POINTER(GridResultC)]),
    "mdb_grid_submit": (C.c_int, [C.c_void_p, C.POINTER(GridInputC), C.c_uint32, C.POINTER(GridRequestC),
                                  C.POINTER(C.c_void_p)]),
    "mdb_grid_wait": (C.c_int, [C.c_void_p, C.POINTER(C.POINTER(GridResultC))]),
    "mdb_grid_cancel": (None, [C.c_void_p]),
    "mdb_grid_result_tag_views": (C.c_void_p, [C.POINTER(GridResultC), C.c_uint32]),
    "mdb_replicate_views": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.c_uint64]),
    "mdb_agg_batch": (C.c_int, [C.c_void_p, C.POINTER(SegmentsC), C.c_uint32,
                                C.POINTER(AggStateC)]),
    "mdb_agg_batch_dev": (C.c_int, [C.c_void_p, C.POINTER(SegmentsC), C.c_uint32,
                                    C.POINTER(AggStateC)]),
    "mdb_agg_batch_list": (C.c_int, [C.c_void_p, C.POINTER(C.POINTER(SegmentsC)), C.c_uint32, C.c_uint32,
                                     C.POINTER(AggStateC)]),
    "mdb_agg_batch_range": (C.c_int, [C.c_void_p, C.POINTER(SegmentsC), C.c_int64, C.c_int64,
                                      C.c_uint32, C.POINTER(AggStateC)]),
    "mdb_agg_batch_range_dev": (C.c_int, [C.c_void_p, C.POINTER(SegmentsC), C.c_int64, C.c_int64,
                                          C.c_uint32, C.POINTER(AggStateC)]),
    "mdb_agg_batch_range_list": (C.c_int, [C.c_void_p, C.POINTER(C.POINTER(SegmentsC)), C.c_uint32, C.c_int64,
                                           C.c_int64, C.c_uint32, C.POINTER(AggStateC)]),
    "mdb_compress_series": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, ErrorBoundC,
                                      C.POINTER(C.POINTER(SegmentsOwnedC))]),
    "mdb_compress_chunks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                      ErrorBoundC, C.POINTER(C.POINTER(SegmentsOwnedC))]),
    "mdb_compress_chunk_list": (C.c_int, [C.c_void_p, C.POINTER(ChunkC), C.c_uint64, ErrorBoundC,
                                          C.POINTER(C.POINTER(SegmentsOwnedC))]),
    "mdb_compress_chunks_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_uint64, ErrorBoundC, C.c_int64, C.c_int64,
                                          C.c_void_p, C.POINTER(C.POINTER(SegmentsOwnedC))]),
    "mdb_segments_validate_dev": (C.c_int, [C.c_void_p, C.POINTER(SegmentsC)]),
    "mdb_split_and_compress_univariate": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p),
                                                    C.POINTER(ErrorBoundC), C.c_uint32, C.c_uint64,
                                                    C.POINTER(C.POINTER(SegmentsOwnedC))]),
    "mdb_is_value_within_error_bound": (C.c_int, [ErrorBoundC, C.c_float, C.c_float,
                                                  C.POINTER(C.c_int32)]),
    "mdb_are_compressed_timestamps_regular": (C.c_int, [C.c_void_p, C.c_uint64,
                                                        C.POINTER(C.c_int32)]),
    "mdb_comm_unique_id": (C.c_int, [C.c_void_p]),
    "mdb_comm_init": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "mdb_comm_close": (C.c_int, [C.c_void_p]),
    "mdb_agg_all_reduce": (C.c_int, [C.c_void_p, C.POINTER(AggStateC), C.POINTER(C.c_int32)]),
    "mdb_agg_merge": (C.c_int, [C.POINTER(AggStateC), C.POINTER(AggStateC)]),
    "mdb_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "mdb_profile_reset": (C.c_int, [C.c_void_p]),
    "mdb_profile_get": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64),
                                  C.POINTER(C.c_double)]),
    "mdb_profile_names": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint64]),
    "mdb_synth_values_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64,
                                       C.c_uint64]),
    "mdb_set_option": (C.c_int, [C.c_char_p, C.c_char_p]),
    "mdb_reload_options": (C.c_int, []),
    "mdb_option": (C.c_char_p, [C.c_char_p]),
}

_hip_library = None

# The library reads its switches (the MDB_* environment variables) once per process. Tests change os.environ from one
# call to the next: with this set (tests/conftest.py) every call through the binding is preceded by
# mdb_reload_options(), so that a test's monkeypatch.setenv() is seen by the call behind it.
RELOAD_OPTIONS_BEFORE_EVERY_CALL = False


class _ReloadingLibrary:
    """The ctypes library with mdb_reload_options() in front of every call while RELOAD_OPTIONS_BEFORE_EVERY_CALL."""

    def __init__(self, library):
        self._library = library

    def __getattr__(self, name):
        function = getattr(self._library, name)
        if not RELOAD_OPTIONS_BEFORE_EVERY_CALL or name in ("mdb_reload_options", "mdb_set_option", "mdb_option", "mdb_last_error"):
            return function
        reload_options = self._library.mdb_reload_options

        def call(*args):
            reload_options()
            return function(*args)

        return call


def hip_symbol_names():
    """Every symbol include/mdb.h declares (checked against the built library by the tests)."""
    return sorted(_HIP_SYMBOLS)


def load_hip_library():
    """Load libmdb_hip.so and declare its prototypes. Raises if it is missing: no fallback."""
    global _hip_library
    if _hip_library is not None:
        return _hip_library
    if not os.path.exists(HIP_LIBRARY_PATH):
        raise RuntimeError(
            f"{HIP_LIBRARY_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback."
        )
    library = C.CDLL(HIP_LIBRARY_PATH, mode=C.RTLD_GLOBAL)
    for name, (restype, argtypes) in _HIP_SYMBOLS.items():
        if "MDB_HIP_LIBRARY" in os.environ and not hasattr(library, name):
            continue  # A/B timing against an older build that predates a newer entry point
        function = getattr(library, name)  # AttributeError if the library lacks a declared symbol
        function.restype = restype
        function.argtypes = argtypes
    _hip_library = _ReloadingLibrary(library) if hasattr(library, "mdb_reload_options") else library
    return _hip_library
