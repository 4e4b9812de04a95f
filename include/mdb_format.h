/*
 * mdb_format.h - constants and memory layouts of ModelarDB compressed segments.
 *
 * Pure C header shared by the HIP product library (modelardb-rs_amd/csrc), the host operators and
 * the CPU oracle (oracle/). Nothing here is executable; it restates the *format* the reference
 * defines so both sides agree on it.
 *
 * Reference sources restated (paths relative to the reference repository root):
 *   crates/modelardb_compression/src/models/mod.rs:36-50     model type ids, value size
 *   crates/modelardb_compression/src/compression.rs:38       RESIDUAL_VALUES_MAX_LENGTH
 *   crates/modelardb_types/src/types.rs:37-50,299-335        Timestamp=i64 us, Value=f32, ErrorBound
 *   crates/modelardb_types/src/schemas.rs:31-72              segment schema, metadata size (29)
 *   crates/modelardb_server/src/storage/mod.rs:58            65 536 point ingest buffers
 */
#ifndef MDB_FORMAT_H
#define MDB_FORMAT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Model type ids (models/mod.rs:36-44). */
#define MDB_PMC_MEAN_ID 0
#define MDB_SWING_ID 1
#define MDB_MACAQUE_V_ID 2
#define MDB_MODEL_TYPE_COUNT 3

/* sizeof(Value) in bytes / bits (models/mod.rs:46-50). */
#define MDB_VALUE_SIZE_IN_BYTES 4
#define MDB_VALUE_SIZE_IN_BITS 32

/* Fixed-width bytes of one segment row: Int8 + 2*Timestamp + 3*Float32, BinaryView columns count
 * as 0 (schemas.rs:57-64 via arrow's primitive_width()). It only enters the bytes-per-value accept
 * test: PMC 29/len, Swing 30/len, accepted iff <= 4.0 (compression.rs:238). */
#define MDB_COMPRESSED_METADATA_SIZE_IN_BYTES 29

/* At most 255 residual values ride inside a model segment (compression.rs:38). */
#define MDB_RESIDUAL_VALUES_MAX_LENGTH 255

/* Points per ingest buffer handed to the compressor by the server (storage/mod.rs:58). */
#define MDB_UNCOMPRESSED_DATA_BUFFER_CAPACITY 65536

/* ErrorBound (types.rs:299-335). */
#define MDB_EB_LOSSLESS 0
#define MDB_EB_ABSOLUTE 1
#define MDB_EB_RELATIVE 2

typedef struct mdb_error_bound {
    int32_t kind; /* MDB_EB_* */
    float value;  /* absolute bound, or relative bound in percent; ignored for lossless */
} mdb_error_bound;

/* One Arrow BinaryView "view" (16 bytes). length <= 12: the bytes are inline (zero padded);
 * otherwise 4 prefix bytes, the index of the variadic data buffer and the offset into it. */
typedef struct mdb_view16 {
    int32_t length;
    union {
        uint8_t inlined[12];
        struct {
            uint8_t prefix[4];
            int32_t buffer_index;
            int32_t offset;
        } ref;
    } u;
} mdb_view16;

/* One Arrow BinaryViewArray column: views + its variadic data buffers, borrowed from Arrow. */
typedef struct mdb_binview_col {
    const mdb_view16 *views;       /* n views */
    const uint8_t *const *buffers; /* n_buffers data buffer base pointers (may be NULL if 0) */
    const int64_t *buffer_sizes;   /* n_buffers sizes in bytes (needed to stage to the device) */
    int32_t n_buffers;
} mdb_binview_col;

/* Struct-of-arrays view of a RecordBatch with QUERY_COMPRESSED_SCHEMA (schemas.rs:40-52):
 * 0 model_type_id | 1 start_time | 2 end_time | 3 timestamps | 4 min_value | 5 max_value |
 * 6 values | 7 residuals | (8 error: never read). Pointers go straight into Arrow buffers. */
typedef struct mdb_segments {
    uint64_t n;
    const int8_t *model_type_id;
    const int64_t *start_time;
    const int64_t *end_time;
    mdb_binview_col timestamps;
    const float *min_value;
    const float *max_value;
    mdb_binview_col values;
    mdb_binview_col residuals;
} mdb_segments;

/* Counters of GridStreamMetrics (query/grid_exec.rs:441-518). */
typedef struct mdb_grid_metrics {
    uint64_t rows_created;
    uint64_t rows_created_by_model_type[MDB_MODEL_TYPE_COUNT];
    uint64_t segments_with_residuals;
    uint64_t segments_with_model_type[MDB_MODEL_TYPE_COUNT];
    uint64_t segments_regular;
    uint64_t segments_irregular;
} mdb_grid_metrics;

/* Partial state of the five Model*Accumulators (optimizer/model_simple_aggregates.rs:336-618).
 * A fresh state is {0.0, 0, FLT_MAX, -FLT_MAX} (f32::MAX / f32::MIN, :413, :456). */
typedef struct mdb_agg_state {
    double sum;
    int64_t count;
    float min;
    float max;
} mdb_agg_state;

#define MDB_AGG_COUNT 1u
#define MDB_AGG_MIN 2u
#define MDB_AGG_MAX 4u
#define MDB_AGG_SUM 8u
#define MDB_AGG_AVG 16u

/* Segments produced by the compressor, owned by the library that made them: the nine columns of
 * COMPRESSED_SCHEMA minus the constant field_column/tag columns, BinaryView columns with one data
 * buffer each. `seg` aliases the owned memory so it can be passed straight back to grid/agg. */
typedef struct mdb_segments_owned {
    mdb_segments seg;
    const float *error;          /* always NaN (types.rs:265, compression.rs:398) */
    const uint32_t *chunk_index; /* which input chunk each segment came from */
    int32_t on_device;           /* 0: host pointers, 1: device pointers */
    void *priv_;                 /* owner's bookkeeping */
} mdb_segments_owned;

/* Reconstructed data points owned by the library: page-locked host memory (so the copy from the
 * device runs at the full PCIe rate) that the caller wraps without copying (e.g. arrow-rs
 * Buffer::from_custom_allocation) and returns with mdb_grid_result_free(). */
typedef struct mdb_grid_result {
    int64_t *timestamps;        /* n */
    float *values;              /* n */
    uint32_t *rows_per_segment; /* n_segments */
    uint64_t n;
    uint64_t n_segments;
    uint64_t reserved_front;    /* writable rows BEFORE timestamps[0] / values[0] (for leftovers) */
    mdb_grid_metrics metrics;
    void *priv_;
} mdb_grid_result;

/* One input RecordBatch of a pipelined grid call (mdb_grid_submit): its segment columns and, when the
 * stream's table has tag columns, the views of the batch's tag arrays (Utf8View: the same 16-byte
 * views, one per segment row), which the library repeats once per reconstructed data point. */
typedef struct mdb_grid_input {
    mdb_segments segments;
    const mdb_view16 *const *tag_views; /* n_tag_columns arrays of segments.n views; NULL without tags */
    /* n_tag_columns numbers added to buffer_index of every view longer than 12 bytes: the output tag
     * column's data buffers are [the caller's own buffers..., this batch's buffers...], and this is
     * where this batch's buffers start in that list. NULL: 0. */
    const int32_t *tag_buffer_shift;
} mdb_grid_input;

/* What to do with the inputs of one mdb_grid_submit. */
typedef struct mdb_grid_request {
    uint32_t flags;          /* MDB_GRID_HAS_RANGE | MDB_GRID_VALUES_ONLY */
    uint32_t n_tag_columns;  /* tag arrays per input */
    int64_t t_lo, t_hi;      /* with MDB_GRID_HAS_RANGE */
    uint64_t reserve_front;  /* writable rows in front of every output column (the leftovers) */
} mdb_grid_request;

/* One series chunk of mdb_compress_chunk_list: n sorted data points in two arrays of the caller. */
typedef struct mdb_chunk {
    const int64_t *ts;
    const float *values;
    uint64_t n;
} mdb_chunk;

#ifdef __cplusplus
}
#endif

/* The layouts above are the ABI: pinned here for C and C++ and, with the same numbers, as `const`
 * assertions in rust/modelardb_hip/src/sys.rs and in tests/test_abi_cpu.py (ctypes). */
#include <stddef.h>
#ifdef __cplusplus
#define MDB_LAYOUT_ASSERT(condition) static_assert(condition, #condition)
#else
#define MDB_LAYOUT_ASSERT(condition) _Static_assert(condition, #condition)
#endif
MDB_LAYOUT_ASSERT(sizeof(mdb_error_bound) == 8);
MDB_LAYOUT_ASSERT(offsetof(mdb_error_bound, value) == 4);
MDB_LAYOUT_ASSERT(sizeof(mdb_view16) == 16);
MDB_LAYOUT_ASSERT(offsetof(mdb_view16, u) == 4);
MDB_LAYOUT_ASSERT(sizeof(mdb_binview_col) == 32);
MDB_LAYOUT_ASSERT(offsetof(mdb_binview_col, buffers) == 8);
MDB_LAYOUT_ASSERT(offsetof(mdb_binview_col, buffer_sizes) == 16);
MDB_LAYOUT_ASSERT(offsetof(mdb_binview_col, n_buffers) == 24);
MDB_LAYOUT_ASSERT(sizeof(mdb_segments) == 144);
MDB_LAYOUT_ASSERT(offsetof(mdb_segments, model_type_id) == 8);
MDB_LAYOUT_ASSERT(offsetof(mdb_segments, start_time) == 16);
MDB_LAYOUT_ASSERT(offsetof(mdb_segments, end_time) == 24);
MDB_LAYOUT_ASSERT(offsetof(mdb_segments, timestamps) == 32);
MDB_LAYOUT_ASSERT(offsetof(mdb_segments, min_value) == 64);
MDB_LAYOUT_ASSERT(offsetof(mdb_segments, max_value) == 72);
MDB_LAYOUT_ASSERT(offsetof(mdb_segments, values) == 80);
MDB_LAYOUT_ASSERT(offsetof(mdb_segments, residuals) == 112);
MDB_LAYOUT_ASSERT(sizeof(mdb_grid_metrics) == 80);
MDB_LAYOUT_ASSERT(offsetof(mdb_grid_metrics, rows_created_by_model_type) == 8);
MDB_LAYOUT_ASSERT(offsetof(mdb_grid_metrics, segments_with_residuals) == 32);
MDB_LAYOUT_ASSERT(offsetof(mdb_grid_metrics, segments_with_model_type) == 40);
MDB_LAYOUT_ASSERT(offsetof(mdb_grid_metrics, segments_regular) == 64);
MDB_LAYOUT_ASSERT(offsetof(mdb_grid_metrics, segments_irregular) == 72);
MDB_LAYOUT_ASSERT(sizeof(mdb_agg_state) == 24);
MDB_LAYOUT_ASSERT(offsetof(mdb_agg_state, count) == 8);
MDB_LAYOUT_ASSERT(offsetof(mdb_agg_state, min) == 16);
MDB_LAYOUT_ASSERT(offsetof(mdb_agg_state, max) == 20);
MDB_LAYOUT_ASSERT(sizeof(mdb_segments_owned) == 176);
MDB_LAYOUT_ASSERT(offsetof(mdb_segments_owned, error) == 144);
MDB_LAYOUT_ASSERT(offsetof(mdb_segments_owned, chunk_index) == 152);
MDB_LAYOUT_ASSERT(offsetof(mdb_segments_owned, on_device) == 160);
MDB_LAYOUT_ASSERT(offsetof(mdb_segments_owned, priv_) == 168);
MDB_LAYOUT_ASSERT(sizeof(mdb_grid_result) == 136);
MDB_LAYOUT_ASSERT(offsetof(mdb_grid_result, values) == 8);
MDB_LAYOUT_ASSERT(offsetof(mdb_grid_result, rows_per_segment) == 16);
MDB_LAYOUT_ASSERT(offsetof(mdb_grid_result, n) == 24);
MDB_LAYOUT_ASSERT(offsetof(mdb_grid_result, n_segments) == 32);
MDB_LAYOUT_ASSERT(offsetof(mdb_grid_result, reserved_front) == 40);
MDB_LAYOUT_ASSERT(offsetof(mdb_grid_result, metrics) == 48);
MDB_LAYOUT_ASSERT(offsetof(mdb_grid_result, priv_) == 128);
MDB_LAYOUT_ASSERT(sizeof(mdb_grid_input) == 160);
MDB_LAYOUT_ASSERT(offsetof(mdb_grid_input, tag_views) == 144);
MDB_LAYOUT_ASSERT(offsetof(mdb_grid_input, tag_buffer_shift) == 152);
MDB_LAYOUT_ASSERT(sizeof(mdb_grid_request) == 32);
MDB_LAYOUT_ASSERT(offsetof(mdb_grid_request, n_tag_columns) == 4);
MDB_LAYOUT_ASSERT(offsetof(mdb_grid_request, t_lo) == 8);
MDB_LAYOUT_ASSERT(offsetof(mdb_grid_request, t_hi) == 16);
MDB_LAYOUT_ASSERT(offsetof(mdb_grid_request, reserve_front) == 24);
MDB_LAYOUT_ASSERT(sizeof(mdb_chunk) == 24);
MDB_LAYOUT_ASSERT(offsetof(mdb_chunk, values) == 8);
MDB_LAYOUT_ASSERT(offsetof(mdb_chunk, n) == 16);

#endif /* MDB_FORMAT_H */
