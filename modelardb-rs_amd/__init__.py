"""MI355X-native implementation of ModelarDB's model-compression / grid / segment-aggregate path.

The directory name carries the reference's repository name, so it is imported either through the
``modelardb_rs_amd`` alias module at the repository root or with
``importlib.import_module("modelardb-rs_amd")``.
"""

from . import _abi, api, host, segment_files, segments, sharding  # noqa: F401
from ._abi import (  # noqa: F401
    MDB_AGG_AVG, MDB_AGG_COUNT, MDB_AGG_MAX, MDB_AGG_MIN, MDB_AGG_SUM, MDB_MACAQUE_V_ID,
    MDB_PMC_MEAN_ID, MDB_SWING_ID, MODEL_TYPE_NAMES, load_hip_library,
)
from .segments import BinaryViewColumn, SegmentBatch, error_bound  # noqa: F401
from .api import (  # noqa: F401
    Context, DeviceSegments, HipError, are_compressed_timestamps_regular, comm_unique_id,
    is_value_within_error_bound,
)
