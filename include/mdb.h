/*
 * mdb.h - C ABI of libmdb_hip.so: ModelarDB's model-compression / grid / segment-aggregate hot
 * path as hand-written HIP kernels for gfx950 (MI355X).
 *
 * This is the drop-in boundary. The reference has no plugin API: modelardb_storage and
 * modelardb_server call the modelardb_compression crate directly. Each entry point below replaces
 * one of those call sites at BATCH granularity (one call per Arrow RecordBatch instead of one per
 * row); INTEGRATION.md shows the Rust `extern "C"` block and the patched call sites.
 *
 * Conventions (mirroring the reference's own C API, crates/modelardb_embedded/src/capi.rs:58-80
 * and bindings/c/modelardb_embedded.h:74-78,202-203):
 *   - every function returns 0 on success and 1 on failure;
 *   - mdb_last_error() returns the message of the last failure on the calling thread, valid until
 *     the next failing call on that thread;
 *   - malformed segments (the reference panics: models/mod.rs:170,237, macaque_v.rs:224,279-280,
 *     types.rs:316-318,391,405) are error returns, never aborts;
 *   - inputs are borrowed for the duration of the call; outputs are written into caller-allocated
 *     buffers, or returned as mdb_segments_owned that the caller frees with mdb_segments_free().
 *   - a context (mdb_ctx) owns one HIP stream and scratch memory. Calls on one context are
 *     serialised by an internal mutex; use one context per thread for concurrency.
 *
 * "host" entry points take host pointers (straight into Arrow buffers) and return after the
 * results are in host memory. "_dev" entry points take device pointers, enqueue on the context's
 * stream and return after the stream has been synchronised unless stated otherwise.
 *
 * Paths in the citations are relative to the reference repository root.
 */
#ifndef MDB_H
#define MDB_H

#include "mdb_format.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mdb_ctx mdb_ctx;

/* ---- lifetime ------------------------------------------------------------------------------- */

/* Create a context on HIP device `device`. Constructor convention of capi.rs:218-229. */
int mdb_init(int device, mdb_ctx **ctx);
int mdb_close(mdb_ctx *ctx);
/* Another context on the device of `ctx` (its own stream and scratch memory): what an operator that
 * wants two batches in flight - the copy of one overlapping the kernels of the next - asks for.
 * (All contexts of a device share one pool of page-locked result blocks, so a context made per query does
 * not pay for pinning memory again; and a clone that is closed is kept - up to four of them - for the next
 * mdb_clone of the same context, stream and scratch included, until that context is closed itself.) */
int mdb_clone(mdb_ctx *ctx, mdb_ctx **out);
const char *mdb_last_error(void);
/* "libmdb_hip <version> gfx950"; never fails. */
const char *mdb_version(void);
/* Use an externally created hipStream_t (e.g. torch's current stream) instead of the context's. */
int mdb_set_stream(mdb_ctx *ctx, void *hip_stream);
/* Give back the working memory the context has grown for the batches seen so far (device scratch,
 * page-locked staging and the recycled result blocks); the next call grows what it needs again.
 * A context keeps this memory between calls because allocating is slow - a fit of 10^10 points
 * leaves tens of GB behind - so a long-lived owner calls this after an unusually large batch.
 * released_bytes (may be NULL): device bytes given back. Results and segments already handed out
 * stay valid. */
int mdb_trim(mdb_ctx *ctx, uint64_t *released_bytes);
/* The same, automatically: after every call on this context, device scratch beyond `bytes` is given back
 * (largest allocations first; 0, the default, keeps everything). For owners of many contexts - one
 * GridStream per field column - each of which would otherwise keep what its largest batch needed. A clone
 * (mdb_clone) starts with the limit of the context it was made from. */
int mdb_set_scratch_limit(mdb_ctx *ctx, uint64_t bytes);
/* Name, CU count, HBM bytes of the context's device. */
int mdb_device_info(mdb_ctx *ctx, char *name, uint64_t name_cap, int32_t *compute_units,
                    uint64_t *hbm_bytes);

/* ---- device memory owned by the library (so the bench needs no other allocator) -------------- */

int mdb_dev_alloc(mdb_ctx *ctx, uint64_t bytes, void **dev_ptr);
int mdb_dev_free(mdb_ctx *ctx, void *dev_ptr);
int mdb_dev_upload(mdb_ctx *ctx, void *dev_dst, const void *host_src, uint64_t bytes);
int mdb_dev_download(mdb_ctx *ctx, void *host_dst, const void *dev_src, uint64_t bytes);
int mdb_dev_sync(mdb_ctx *ctx);
/* Copy a host batch of segments (Arrow buffers) to the device; the result has on_device = 1. */
int mdb_segments_upload(mdb_ctx *ctx, const mdb_segments *host, mdb_segments_owned **dev);
/* Copy a device batch back; the result has on_device = 0 and one data buffer per column (several if the
 * column's payloads exceed 2 GiB). */
int mdb_segments_download(mdb_ctx *ctx, const mdb_segments_owned *dev, mdb_segments_owned **host);
void mdb_segments_free(mdb_segments_owned *segments);
/* mdb_segments_upload checks every out-of-line view of a HOST batch against the column's data buffers
 * (a view that points outside them is an error, never a wild device read). A batch that is ALREADY on
 * the device and was not made by this library (mdb_segments_upload / mdb_compress_chunks*) must be
 * well formed: the "_dev" entry points follow buffer_index and offset without looking. This runs the
 * same check on such a batch (views in device memory, buffer_sizes a host array as everywhere). */
int mdb_segments_validate_dev(mdb_ctx *ctx, const mdb_segments *dev);

/* ---- grid: replaces the per-row loop of GridStream::grid_and_append_to_leftovers_in_current_batch
 *      (crates/modelardb_storage/src/query/grid_exec.rs:323-356) which calls
 *      modelardb_compression::grid (crates/modelardb_compression/src/models/mod.rs:190-251) ------ */

/* Number of data points the batch reconstructs to (sum over rows of the timestamps a row
 * decompresses to), so the caller can allocate the outputs. */
int mdb_grid_count(mdb_ctx *ctx, const mdb_segments *in, uint64_t *n_out);

/* Reconstruct every data point of every segment, in segment order. out_ts/out_val need `cap`
 * elements (cap >= mdb_grid_count). out_rows_per_segment (optional, n entries) is what the caller
 * uses to replicate tag values (grid_exec.rs:339-346). metrics is optional (grid_exec.rs:511-518). */
int mdb_grid_batch(mdb_ctx *ctx, const mdb_segments *in, int64_t *out_ts, float *out_val,
                   uint32_t *out_rows_per_segment, uint64_t cap, uint64_t *n_out,
                   mdb_grid_metrics *metrics);

/* Device resident variant: `in` holds device pointers (e.g. from mdb_segments_upload or
 * mdb_compress_chunks_dev), outputs are device buffers. out_ts may be NULL: then only the values
 * are reconstructed, which is what a join of several field columns of the same series needs for
 * every field after the first (SortedJoinExec zips per-field GridExec outputs that share their
 * timestamps, crates/modelardb_storage/src/query/sorted_join_exec.rs:252-310). */
int mdb_grid_count_dev(mdb_ctx *ctx, const mdb_segments *in, uint64_t *n_out);
int mdb_grid_batch_dev(mdb_ctx *ctx, const mdb_segments *in, int64_t *out_ts, float *out_val,
                       uint32_t *out_rows_per_segment, uint64_t cap, uint64_t *n_out,
                       mdb_grid_metrics *metrics);

/* Predicate pushdown: only the data points with t_lo <= timestamp <= t_hi are
 * reconstructed. The reference's GridStream reconstructs every point of every segment the Parquet
 * filter let through and prunes afterwards (grid_exec.rs:366-387); the result is the same rows in
 * the same order. rows_per_segment and the row counters of `metrics` count the rows produced. */
int mdb_grid_count_range(mdb_ctx *ctx, const mdb_segments *in, int64_t t_lo, int64_t t_hi,
                         uint64_t *n_out);
int mdb_grid_batch_range(mdb_ctx *ctx, const mdb_segments *in, int64_t t_lo, int64_t t_hi,
                         int64_t *out_ts, float *out_val, uint32_t *out_rows_per_segment, uint64_t cap,
                         uint64_t *n_out, mdb_grid_metrics *metrics);
int mdb_grid_count_range_dev(mdb_ctx *ctx, const mdb_segments *in, int64_t t_lo, int64_t t_hi,
                             uint64_t *n_out);
int mdb_grid_batch_range_dev(mdb_ctx *ctx, const mdb_segments *in, int64_t t_lo, int64_t t_hi,
                             int64_t *out_ts, float *out_val, uint32_t *out_rows_per_segment,
                             uint64_t cap, uint64_t *n_out, mdb_grid_metrics *metrics);

/* One-call form for host callers: sizes, reconstructs and copies back in one go (one upload of the
 * segments, one prepass, one copy from the device) into page-locked memory owned by the library,
 * which the caller wraps without copying and returns with mdb_grid_result_free() (allowed after
 * mdb_close()). flags: MDB_GRID_HAS_RANGE applies the predicate t_lo <= timestamp <= t_hi;
 * MDB_GRID_VALUES_ONLY skips the timestamps (result->timestamps is NULL): the second and later
 * field columns of a SortedJoinExec only contribute their values (sorted_join_exec.rs:268-275), so
 * their timestamps need not cross PCIe. Two to three times
 * faster end to end than mdb_grid_count + mdb_grid_batch into pageable memory.
 * reserve_front asks for that many writable rows in front of the reconstructed points, so a
 * GridStream can put the leftovers of its previous batch there (grid_exec.rs:302-320) and hand
 * out slices of the block without copying the new points at all. */
#define MDB_GRID_HAS_RANGE 1u
#define MDB_GRID_VALUES_ONLY 2u
int mdb_grid_batch_owned(mdb_ctx *ctx, const mdb_segments *in, uint32_t flags, int64_t t_lo,
                         int64_t t_hi, uint64_t reserve_front, mdb_grid_result **out);
void mdb_grid_result_free(mdb_grid_result *result);

/* The library's switches (the MDB_* names of INTEGRATION.md 2.6: A/B timings and the scheduling modes the tests force;
 * the defaults are what a deployment runs). The process's environment is read ONCE, by the first call that asks for a
 * switch: no getenv() inside a call. mdb_set_option sets one switch for the whole process without touching the
 * environment (value NULL: back to "not set"); mdb_reload_options reads the environment again (what a test that has
 * changed it calls). No counterpart in the reference (its settings are
 * crates/modelardb_server/src/configuration.rs, none of which reaches this path). */
int mdb_set_option(const char *name, const char *value);
int mdb_reload_options(void);
/* What a switch is set to (NULL: not set). The text stays valid for the life of the process (setting the switch again
 * or reloading the table makes later look-ups return another text and leaves this one as it is), so mdb_set_option and
 * mdb_reload_options may be called while other threads are inside calls: a call sees a switch as it was when it asked.
 * The price of that: every DISTINCT value a switch has ever had is kept (a few bytes each, never freed) - switches are
 * for deployments and tests, not a per-query channel; a caller that sets one to ever-changing values grows the table. */
const char *mdb_option(const char *name);

/* Pipelined form, for an operator that is polled (GridStream::poll_next, grid_exec.rs:402-429): submit
 * returns at once with a ticket, a worker thread of the library reconstructs the batch on the context or
 * on a clone of it that the library keeps (submits alternate between the two), so the kernels of one
 * batch run while the previous batch's points still cross PCIe; mdb_grid_wait blocks until the result
 * is in host memory. Two submits may be outstanding per context before the third one queues behind them.
 *   - inputs: one OR SEVERAL RecordBatches of segments, reconstructed by one launch as if they were one
 *     batch (rows in the order of the list): the batches DataSourceExec hands a GridStream hold 8 192
 *     segments, and a launch pays off from 10^5 (SURVEY 8(f) N2 - the caller keeps polling its input and
 *     submits what it has got, no concat_batches on the host). rows_per_segment and n_segments of the
 *     result run over all inputs.
 *   - tags (grid_exec.rs:339-346): with request->n_tag_columns > 0 every input carries the views of its
 *     tag arrays; the result then also holds, per tag column, the views repeated once per reconstructed
 *     row (mdb_grid_result_tag_views), written by the library's host threads past the cache while the
 *     next batch is on the GPU. The strings themselves are not copied: an output view points into the
 *     input's data buffers, which the caller lists behind its own (tag_buffer_shift).
 *   - the mdb_grid_input / mdb_grid_request structs and the small tables they point at are copied by
 *     submit; the Arrow buffers behind them must stay alive until mdb_grid_wait / mdb_grid_cancel returns.
 *   - mdb_grid_wait consumes the ticket whether it succeeds or not; a stream that is dropped with a ticket
 *     outstanding calls mdb_grid_cancel (waits for the job and frees its result).
 * Replaces, together with mdb_grid_result_*: grid_exec.rs:261-391 for several input batches at once. */
typedef struct mdb_grid_ticket mdb_grid_ticket;
int mdb_grid_submit(mdb_ctx *ctx, const mdb_grid_input *inputs, uint32_t n_inputs,
                    const mdb_grid_request *request, mdb_grid_ticket **ticket);
int mdb_grid_wait(mdb_grid_ticket *ticket, mdb_grid_result **out);
void mdb_grid_cancel(mdb_grid_ticket *ticket);
/* The replicated views of tag column `column` of a result of mdb_grid_submit: result->n views, the view of
 * the first reconstructed row first, with result->reserved_front writable views in front of it (the
 * leftovers' tags). NULL if the request had fewer tag columns. Freed with the result. */
mdb_view16 *mdb_grid_result_tag_views(const mdb_grid_result *result, uint32_t column);
/* The replication by itself, for callers that keep their own output buffers (host arithmetic, no
 * context): out[k] = views[i] for the rows_per_segment[i] rows of segment i, buffer_index of views longer
 * than 12 bytes moved by buffer_shift. out needs sum(rows_per_segment) views (checked against out_cap).
 * Large fills are split over the library's host threads and written with streaming stores. */
int mdb_replicate_views(const mdb_view16 *views, const uint32_t *rows_per_segment, uint64_t n_segments,
                        int32_t buffer_shift, mdb_view16 *out, uint64_t out_cap);

/* ---- aggregates: replaces Model{Count,Min,Max,Sum,Avg}Accumulator::update_batch
 *      (crates/modelardb_storage/src/optimizer/model_simple_aggregates.rs:345-358, 395-401,
 *      438-444, 481-513, 553-587) which call modelardb_compression::{len,sum}
 *      (crates/modelardb_compression/src/models/mod.rs:98-184) ---------------------------------- */

/* Fold the batch into *inout for the aggregates in which_mask (MDB_AGG_*). COUNT/MIN/MAX are exact;
 * SUM adds the f32 per-segment sums in f64 with a fixed (deterministic) tree order, so it can
 * differ from the reference's sequential f64 accumulation in the last bits (the reference's own
 * tests allow 0.001 %: crates/modelardb_server/tests/integration_test.rs:1155-1171). */
int mdb_agg_batch(mdb_ctx *ctx, const mdb_segments *in, uint32_t which_mask, mdb_agg_state *inout);
int mdb_agg_batch_dev(mdb_ctx *ctx, const mdb_segments *in, uint32_t which_mask,
                      mdb_agg_state *inout);
/* The same for SEVERAL RecordBatches of segments (rows in the order of the list), uploaded and folded as one
 * batch: an accumulator is handed 8 192 segments per update_batch (model_simple_aggregates.rs:345, 481, 553) and
 * nobody sees its state before state() (:367, 523, 590), while a call costs the same for 8 192 and for 262 144
 * segments (the batches are gathered by the caller, without copying, and passed here). */
int mdb_agg_batch_list(mdb_ctx *ctx, const mdb_segments *const *inputs, uint32_t n_inputs, uint32_t which_mask,
                       mdb_agg_state *inout);

/* Extension (BASELINE config 3): aggregates over the data points with
 * t_lo <= timestamp <= t_hi without materialising them. The reference has no such operator: any
 * WHERE on the timestamp falls back to GridExec + filter + AggregateExec
 * (model_simple_aggregates.rs:284-302), and that result is the parity oracle: COUNT / MIN / MAX of the
 * reconstructed points inside the range, SUM their f64 sum. */
int mdb_agg_batch_range(mdb_ctx *ctx, const mdb_segments *in, int64_t t_lo, int64_t t_hi,
                        uint32_t which_mask, mdb_agg_state *inout);
int mdb_agg_batch_range_dev(mdb_ctx *ctx, const mdb_segments *in, int64_t t_lo, int64_t t_hi,
                            uint32_t which_mask, mdb_agg_state *inout);
/* The list form (mdb_agg_batch_list under a time range): what the accumulators of a ranged query fold their
 * gathered batches with. The patched optimizer rule (rust/patches/0002) accepts AggregateExec <- FilterExec(a
 * conjunction of comparisons of the timestamp column with literals) <- SortedJoinExec <- GridExec <-
 * DataSourceExec(the start_time / end_time filter TimeSeriesTable::scan derived from the same comparisons,
 * query/time_series_table.rs:290-373) and hands [t_lo, t_hi] to the accumulators it creates. */
int mdb_agg_batch_range_list(mdb_ctx *ctx, const mdb_segments *const *inputs, uint32_t n_inputs, int64_t t_lo,
                             int64_t t_hi, uint32_t which_mask, mdb_agg_state *inout);

/* ---- fit: replaces try_compress_univariate_time_series
 *      (crates/modelardb_compression/src/compression.rs:191-275), called per field column by
 *      crates/modelardb_server/src/storage/uncompressed_data_manager.rs:563-581 and, through
 *      try_compress_multivariate_time_series (compression.rs:42-179), by
 *      crates/modelardb_embedded/src/operations/data_folder.rs:214-217 and
 *      crates/modelardb_bulkloader/src/main.rs:429-432 ------------------------------------------- */

/* Compress one sorted univariate series. n == 0 gives an empty batch (compression.rs:208-211). */
int mdb_compress_series(mdb_ctx *ctx, const int64_t *ts, const float *values, uint64_t n,
                        mdb_error_bound error_bound, mdb_segments_owned **out);

/* Compress many independent series chunks in one launch: chunk c is
 * [chunk_offsets[c], chunk_offsets[c + 1]) of ts/values. Segments come out grouped by chunk in
 * chunk order, each chunk's segments in time order; out->chunk_index names the chunk.
 * A BinaryView column whose payloads (irregular timestamps, MacaqueV values, residuals) add up to more
 * than 1 GiB comes back with several data buffers, as arrow's builders produce them (types.rs:444-516):
 * the views carry the buffer index, mdb_grid_* / mdb_agg_* read such columns, and mdb_segments_download
 * keeps the buffers apart when they do not fit into one. */
int mdb_compress_chunks(mdb_ctx *ctx, const int64_t *ts, const float *values,
                        const uint64_t *chunk_offsets, uint64_t n_chunks,
                        mdb_error_bound error_bound, mdb_segments_owned **out);

/* The same for chunks that lie wherever the caller has them (one slice of a sorted RecordBatch per series
 * and field, compression.rs:42-107; one finished ingest buffer per series and field,
 * uncompressed_data_manager.rs:530-596): the library gathers them into its page-locked staging block with
 * several threads while earlier parts cross PCIe, so the caller concatenates nothing. Chunks that share a
 * timestamp array (the fields of one series) are checked for regular spacing once. Output as above. */
int mdb_compress_chunk_list(mdb_ctx *ctx, const mdb_chunk *chunks, uint64_t n_chunks,
                            mdb_error_bound error_bound, mdb_segments_owned **out);

/* Device resident variant. ts may be NULL: then chunk c has the regular timestamps
 * regular_start + i * regular_interval (i counted from the start of the chunk's series, given by
 * series_first_index[c], or from 0 if that is NULL), synthesised on the fly. */
int mdb_compress_chunks_dev(mdb_ctx *ctx, const int64_t *ts, const float *values,
                            const uint64_t *chunk_offsets, uint64_t n_chunks,
                            mdb_error_bound error_bound, int64_t regular_start,
                            int64_t regular_interval, const uint64_t *series_first_index,
                            mdb_segments_owned **out);

/* try_split_and_compress_univariate_time_series (compression.rs:147-179): ONE sorted series with
 * n_fields field columns that share its timestamps, field f compressed within error_bounds[f].
 * out[f] receives field f's segments (host memory, as mdb_compress_series); on failure nothing is
 * returned. The timestamps cross PCIe once. */
int mdb_split_and_compress_univariate(mdb_ctx *ctx, const int64_t *ts, const float *const *field_values,
                                      const mdb_error_bound *error_bounds, uint32_t n_fields, uint64_t n,
                                      mdb_segments_owned **out);

/* ---- the crate's two remaining public helpers (crates/modelardb_compression/src/lib.rs:30-33):
 *      plain host arithmetic, no context, no GPU ------------------------------------------------- */

/* is_value_within_error_bound (models/mod.rs:53-77; used by tests of the callers, e.g.
 * crates/modelardb_server/tests/integration_test.rs:1232): *within = 1 or 0. */
int mdb_is_value_within_error_bound(mdb_error_bound error_bound, float real_value, float approximate_value,
                                    int32_t *within);
/* are_compressed_timestamps_regular (models/timestamps.rs:199-202; grid_exec.rs:352 feeds the
 * GridStreamMetrics with it): empty, or the top bit of the first byte is 0. */
int mdb_are_compressed_timestamps_regular(const uint8_t *compressed_timestamps, uint64_t n_bytes,
                                          int32_t *regular);

/* ---- multi-GPU: the final aggregate merge over RCCL / xGMI -----------------------
 * One process and one context per GPU; series are sharded over the GPUs and fit / grid / the
 * per-segment aggregates never exchange anything. The single exchange step is the merge of the
 * accumulators' partial states, which the reference hands to DataFusion's final aggregate
 * (crates/modelardb_storage/src/optimizer/model_simple_aggregates.rs:362-378, 517-534, 591-612). */

#define MDB_COMM_ID_BYTES 128
/* ncclGetUniqueId: called by ONE rank, which hands the 128 bytes to the others by whatever channel
 * the host has (the reference's cluster already talks Arrow Flight; the bench uses torch's store). */
int mdb_comm_unique_id(void *id_out);
/* ncclCommInitRank on the context's device; collective over all `world` ranks. */
int mdb_comm_init(mdb_ctx *ctx, int32_t rank, int32_t world, const void *unique_id);
int mdb_comm_close(mdb_ctx *ctx); /* also done by mdb_close */
/* Merge the partial states of all ranks: one ncclAllGather of 32 bytes per rank on the context's
 * stream, then a fold in RANK ORDER with the accumulators' own rules, so that the f64 SUM is
 * reproducible run to run and identical on every rank (an all-reduce leaves the order of the
 * additions to the ring). Collective. ranks_seen (may be NULL): how many states arrived. */
int mdb_agg_all_reduce(mdb_ctx *ctx, mdb_agg_state *inout, int32_t *ranks_seen);
/* The fold itself, for hosts that move the states themselves: into = merge(into, from). */
int mdb_agg_merge(mdb_agg_state *into, const mdb_agg_state *from);

/* ---- measurement ---------------------------------------------------------------------------- */

/* When enabled every kernel launch is bracketed with hipEvents on the context's stream. */
int mdb_profile_enable(mdb_ctx *ctx, int enabled);
int mdb_profile_reset(mdb_ctx *ctx);
/* Accumulated launches and milliseconds of kernel `name` since the last reset. mdb_compress_chunk_list adds the host
 * side of its calls under names that begin with "host:" (calls and wall-clock milliseconds): host:chunk_list_gather
 * (the host threads' copies into page-locked memory, the copies to the device running behind them),
 * host:chunk_list_upload_tail (what is left of those copies when the last slice is gathered), host:chunk_list_fit,
 * host:chunk_list_download. The jobs of mdb_grid_submit add theirs: host:grid_cursors_by_host_threads,
 * host:grid_wait_for_the_context, host:grid_upload_segments, host:grid_plan, host:grid_launches,
 * host:grid_kernels_and_copy_down, host:grid_free_segments. The profile of a context covers the clones its
 * mdb_grid_submit workers run jobs on: they are switched and reset with it, their launches are counted as its own. */
int mdb_profile_get(mdb_ctx *ctx, const char *name, uint64_t *launches, double *total_ms);
/* Names of all profiled kernels, '\n' separated. */
int mdb_profile_names(mdb_ctx *ctx, char *out, uint64_t cap);

/* Fill out[i] = synthetic series value: series s = first_series + i / n_per_series,
 * point j = i % n_per_series: 100 + 10 sin(2 pi j / P_s + phi_s) + U(-0.05, 0.05). Device buffer. */
int mdb_synth_values_dev(mdb_ctx *ctx, float *out, uint64_t first_series, uint64_t n_series,
                         uint64_t n_per_series, uint64_t seed);

#ifdef __cplusplus
}
#endif

#endif /* MDB_H */
