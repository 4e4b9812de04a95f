/*
 * mdb_oracle.h - C surface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE. The oracle is a from-scratch CPU restatement of the
 * reference's model-compression / grid / segment-aggregate algorithms. Only tests/, the smoke
 * check in __graft_entry__.py and bench.py's cpu_baseline leg may load it, and only as the checker
 * or the reported CPU baseline. The HIP product library never links or calls it.
 *
 * Parity status: the reference (Rust) cannot be compiled in the authoring container (no cargo or
 * rustc), so the oracle is pinned against the known-answer vectors of the reference's own unit
 * tests (tests/test_oracle_kat.py lists each with its reference file:line). One constant is
 * "parity unpinned": COMPRESSED_METADATA_SIZE_IN_BYTES = 29 is derived from arrow 58.3.0's
 * DataType::primitive_width(), which is not in the reference tree and no reference test pins it.
 *
 * Every function cites the reference file:line it follows, paths relative to
 * crates/modelardb_compression/src/ unless stated otherwise.
 */
#ifndef MDB_ORACLE_H
#define MDB_ORACLE_H

#include "../include/mdb_format.h"

#ifdef __cplusplus
extern "C" {
#endif

/* 0 on success, 1 on failure; message of the last failure on this thread. */
const char *ora_last_error(void);

/* models/mod.rs:53-95 */
int ora_is_value_within_error_bound(mdb_error_bound eb, float real_value, float approximate_value);
double ora_maximum_allowed_deviation(mdb_error_bound eb, double value);

/* models/bits.rs:25-174. The writer appends `nbits[i]` low bits of `bits[i]` MSB-first. */
int ora_bits_write(const uint64_t *bits, const uint8_t *nbits, uint64_t n, int finish_with_ones,
                   uint8_t *out, uint64_t cap, uint64_t *out_len);
int ora_bits_read(const uint8_t *bytes, uint64_t nbytes, const uint8_t *nbits, uint64_t n,
                  uint64_t *out_values, uint64_t *remaining_bits);

/* models/timestamps.rs:56-292 */
int ora_compress_residual_timestamps(const int64_t *ts, uint64_t n, uint8_t *out, uint64_t cap,
                                     uint64_t *out_len);
int ora_decompress_all_timestamps(int64_t start_time, int64_t end_time, const uint8_t *bytes,
                                  uint64_t nbytes, int64_t *out, uint64_t cap, uint64_t *n_out);
int ora_are_compressed_timestamps_regular(const uint8_t *bytes, uint64_t nbytes);

/* models/mod.rs:98-284 (per-segment entry points) */
int ora_len(int64_t start_time, int64_t end_time, const uint8_t *ts, uint64_t ts_len,
            uint64_t *out_len);
int ora_sum(int8_t model_type_id, int64_t start_time, int64_t end_time, const uint8_t *ts,
            uint64_t ts_len, float min_value, float max_value, const uint8_t *values,
            uint64_t values_len, const uint8_t *residuals, uint64_t residuals_len, float *out_sum);
int ora_grid(int8_t model_type_id, int64_t start_time, int64_t end_time, const uint8_t *ts,
             uint64_t ts_len, float min_value, float max_value, const uint8_t *values,
             uint64_t values_len, const uint8_t *residuals, uint64_t residuals_len, int64_t *out_ts,
             float *out_val, uint64_t cap, uint64_t *n_out);

/* models/pmc_mean.rs:31-108: feeds values until one is rejected; returns how many were accepted
 * and the model value. */
int ora_pmc_mean_fit(mdb_error_bound eb, const float *v, uint64_t n, uint64_t *n_fit,
                     float *model_value, float *bytes_per_value);
/* models/swing.rs:34-340 */
int ora_swing_fit(mdb_error_bound eb, const int64_t *ts, const float *v, uint64_t n,
                  uint64_t *n_fit, float *first_value, float *last_value, float *bytes_per_value,
                  double *bounds4 /* upper slope, upper intercept, lower slope, lower intercept */);
float ora_swing_sum(int64_t start_time, int64_t end_time, const uint8_t *ts, uint64_t ts_len,
                    float first_value, float last_value, uint64_t residuals_length);
/* models/macaque_v.rs:39-336. seeded != 0: compress_values_without_first(values, seed). */
int ora_macaque_v_compress(mdb_error_bound eb, const float *v, uint64_t n, int seeded, float seed,
                           uint8_t *out, uint64_t cap, uint64_t *out_len, float *min_value,
                           float *max_value, uint8_t *last_leading_zero_bits,
                           uint8_t *last_trailing_zero_bits, float *last_value);
int ora_macaque_v_grid(const uint8_t *bytes, uint64_t nbytes, uint64_t n, int seeded, float seed,
                       float *out);
int ora_macaque_v_sum(const uint8_t *bytes, uint64_t nbytes, uint64_t n, int seeded, float seed,
                      float *out);

/* types.rs:283-407 */
int ora_encode_values_for_pmc_mean(float min_value, float max_value, float residuals_min_value,
                                   float residuals_max_value, uint8_t *out8, uint64_t *out_len);
int ora_decode_values_for_pmc_mean(float min_value, float max_value, const uint8_t *values,
                                   uint64_t values_len, float *out);
int ora_encode_values_for_swing(float min_value, float max_value, int min_value_is_first,
                                float residuals_min_value, float residuals_max_value,
                                uint8_t *out8, uint64_t *out_len);
int ora_decode_values_for_swing(float min_value, float max_value, const uint8_t *values,
                                uint64_t values_len, float *first_value, float *last_value);

/* compression.rs:280-301 + types.rs:61-145: the model selected when fitting from `start_index`. */
typedef struct ora_model {
    int8_t model_type_id;
    uint64_t start_index;
    uint64_t end_index;
    float min_value;
    float max_value;
    uint8_t values[8];
    uint32_t values_len;
    float model_last_value;
    float bytes_per_value;
} ora_model;
int ora_fit_next_model(uint64_t start_index, mdb_error_bound eb, const int64_t *ts, const float *v,
                       uint64_t n, ora_model *out);
/* types.rs:197-267: one segment for `model` plus residuals up to residuals_end_index. */
int ora_model_finish(const ora_model *model, mdb_error_bound eb, uint64_t residuals_end_index,
                     const int64_t *ts, const float *v, uint64_t n, mdb_segments_owned **out);

/* compression.rs:191-275 for every chunk [chunk_offsets[c], chunk_offsets[c+1]) of (ts, v).
 * n_threads > 1 shards chunks over host threads (used only by the CPU baseline). */
int ora_compress_chunks(const int64_t *ts, const float *v, const uint64_t *chunk_offsets,
                        uint64_t n_chunks, mdb_error_bound eb, int n_threads,
                        mdb_segments_owned **out);
void ora_segments_free(mdb_segments_owned *s);

/* query/grid_exec.rs:323-356 per-row loop over a batch (crates/modelardb_storage/src). */
int ora_grid_count(const mdb_segments *in, uint64_t *n_out);
int ora_grid_batch(const mdb_segments *in, int64_t *out_ts, float *out_val,
                   uint32_t *out_rows_per_segment, uint64_t cap, uint64_t *n_out,
                   mdb_grid_metrics *metrics);
/* Same loop sharded over n_threads host threads by contiguous segment ranges (CPU baseline). */
int ora_grid_batch_mt(const mdb_segments *in, int64_t *out_ts, float *out_val, uint64_t cap,
                      uint64_t *n_out, int n_threads);
/* Timed legs only (bench.py cpu_baseline): pin worker w of the threaded entry points to the w-th CPU
 * the process may run on, so first-touched pages stay local to their worker. Off by default. */
void ora_set_thread_pinning(int enabled);
/* optimizer/model_simple_aggregates.rs:345-358,395-401,438-444,481-513,553-587 */
int ora_agg_batch(const mdb_segments *in, uint32_t which_mask, mdb_agg_state *inout);
/* Oracle of the time-range extension: grid + filter t_lo <= ts <= t_hi + aggregate, which is what
 * the reference executes for any WHERE on the timestamp (model_simple_aggregates.rs:284-302). The
 * SUM is accumulated in f64 over the f32 values like DataFusion's SUM(Float32). */
int ora_agg_batch_range(const mdb_segments *in, int64_t t_lo, int64_t t_hi, uint32_t which_mask,
                        mdb_agg_state *inout);

#ifdef __cplusplus
}
#endif

#endif /* MDB_ORACLE_H */
