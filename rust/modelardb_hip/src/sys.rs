//! Raw declarations of the C ABI in `include/mdb_format.h` and `include/mdb.h`.
//!
//! The struct names are the C names on purpose (`mdb_segments`, not `MdbSegments`): the layout
//! assertions at the end of this file carry the same numbers as the `MDB_LAYOUT_ASSERT` lines of
//! `include/mdb_format.h`, and `tests/test_abi_cpu.py` checks that the two lists agree.

#![allow(non_camel_case_types)]

use std::mem::{offset_of, size_of};
use std::os::raw::{c_char, c_int, c_void};

/// Opaque context: one HIP stream + scratch memory. Calls on one context are serialised inside the
/// library; use one context per `GridStream` / accumulator / compression thread.
#[repr(C)]
pub struct mdb_ctx {
    _private: [u8; 0],
}

pub const MDB_PMC_MEAN_ID: i8 = 0;
pub const MDB_SWING_ID: i8 = 1;
pub const MDB_MACAQUE_V_ID: i8 = 2;
pub const MDB_MODEL_TYPE_COUNT: usize = 3;

pub const MDB_EB_LOSSLESS: i32 = 0;
pub const MDB_EB_ABSOLUTE: i32 = 1;
pub const MDB_EB_RELATIVE: i32 = 2;

pub const MDB_AGG_COUNT: u32 = 1;
pub const MDB_AGG_MIN: u32 = 2;
pub const MDB_AGG_MAX: u32 = 4;
pub const MDB_AGG_SUM: u32 = 8;
pub const MDB_AGG_AVG: u32 = 16;

pub const MDB_GRID_HAS_RANGE: u32 = 1;
pub const MDB_GRID_VALUES_ONLY: u32 = 2;

pub const MDB_COMM_ID_BYTES: usize = 128;

/// `ErrorBound` (crates/modelardb_types/src/types.rs:299-335).
#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq)]
pub struct mdb_error_bound {
    pub kind: i32,
    pub value: f32,
}

/// One Arrow BinaryView view: `length`, then 12 inline bytes or {prefix, buffer_index, offset}.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct mdb_view16 {
    pub length: i32,
    pub u: [u8; 12],
}

/// One `BinaryViewArray`: `views()`, `data_buffers()` base pointers and lengths.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct mdb_binview_col {
    pub views: *const mdb_view16,
    pub buffers: *const *const u8,
    pub buffer_sizes: *const i64,
    pub n_buffers: i32,
}

/// Struct-of-arrays view of a RecordBatch with `QUERY_COMPRESSED_SCHEMA`
/// (crates/modelardb_types/src/schemas.rs:40-52); pointers go straight into Arrow buffers.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct mdb_segments {
    pub n: u64,
    pub model_type_id: *const i8,
    pub start_time: *const i64,
    pub end_time: *const i64,
    pub timestamps: mdb_binview_col,
    pub min_value: *const f32,
    pub max_value: *const f32,
    pub values: mdb_binview_col,
    pub residuals: mdb_binview_col,
}

/// The counters of `GridStreamMetrics` (query/grid_exec.rs:441-518), per batch.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct mdb_grid_metrics {
    pub rows_created: u64,
    pub rows_created_by_model_type: [u64; MDB_MODEL_TYPE_COUNT],
    pub segments_with_residuals: u64,
    pub segments_with_model_type: [u64; MDB_MODEL_TYPE_COUNT],
    pub segments_regular: u64,
    pub segments_irregular: u64,
}

/// Partial state of the five `Model*Accumulator`s (optimizer/model_simple_aggregates.rs:336-618).
#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq)]
pub struct mdb_agg_state {
    pub sum: f64,
    pub count: i64,
    pub min: f32,
    pub max: f32,
}

impl mdb_agg_state {
    /// `min: f32::MAX`, `max: f32::MIN` as in model_simple_aggregates.rs:413, 456.
    pub const FRESH: Self = Self { sum: 0.0, count: 0, min: f32::MAX, max: f32::MIN };
}

#[repr(C)]
pub struct mdb_segments_owned {
    pub seg: mdb_segments,
    pub error: *const f32,
    pub chunk_index: *const u32,
    pub on_device: i32,
    pub priv_: *mut c_void,
}

#[repr(C)]
pub struct mdb_grid_result {
    pub timestamps: *mut i64,
    pub values: *mut f32,
    pub rows_per_segment: *mut u32,
    pub n: u64,
    pub n_segments: u64,
    pub reserved_front: u64,
    pub metrics: mdb_grid_metrics,
    pub priv_: *mut c_void,
}

/// One input `RecordBatch` of a pipelined grid call: its segment columns and the views of its tag arrays.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct mdb_grid_input {
    pub segments: mdb_segments,
    /// `n_tag_columns` arrays of `segments.n` views (`StringViewArray::views()`), or null without tags.
    pub tag_views: *const *const mdb_view16,
    /// Per tag column: added to `buffer_index` of every view longer than 12 bytes; null means 0.
    pub tag_buffer_shift: *const i32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct mdb_grid_request {
    pub flags: u32,
    pub n_tag_columns: u32,
    pub t_lo: i64,
    pub t_hi: i64,
    pub reserve_front: u64,
}

/// An outstanding `mdb_grid_submit`.
#[repr(C)]
pub struct mdb_grid_ticket {
    _private: [u8; 0],
}

/// One series chunk of `mdb_compress_chunk_list`: `n` sorted data points in two arrays of the caller.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct mdb_chunk {
    pub ts: *const i64,
    pub values: *const f32,
    pub n: u64,
}

#[link(name = "mdb_hip")]
unsafe extern "C" {
    // ---- lifetime ----------------------------------------------------------------------------
    pub fn mdb_init(device: c_int, ctx: *mut *mut mdb_ctx) -> c_int;
    pub fn mdb_close(ctx: *mut mdb_ctx) -> c_int;
    pub fn mdb_clone(ctx: *mut mdb_ctx, out: *mut *mut mdb_ctx) -> c_int;
    pub fn mdb_last_error() -> *const c_char;
    pub fn mdb_version() -> *const c_char;
    pub fn mdb_set_stream(ctx: *mut mdb_ctx, hip_stream: *mut c_void) -> c_int;
    pub fn mdb_trim(ctx: *mut mdb_ctx, released_bytes: *mut u64) -> c_int;
    pub fn mdb_set_scratch_limit(ctx: *mut mdb_ctx, bytes: u64) -> c_int;
    pub fn mdb_device_info(ctx: *mut mdb_ctx, name: *mut c_char, name_cap: u64, compute_units: *mut i32,
                           hbm_bytes: *mut u64) -> c_int;

    // ---- device memory ---------------------------------------------------------------------------
    pub fn mdb_dev_alloc(ctx: *mut mdb_ctx, bytes: u64, dev_ptr: *mut *mut c_void) -> c_int;
    pub fn mdb_dev_free(ctx: *mut mdb_ctx, dev_ptr: *mut c_void) -> c_int;
    pub fn mdb_dev_upload(ctx: *mut mdb_ctx, dev_dst: *mut c_void, host_src: *const c_void, bytes: u64) -> c_int;
    pub fn mdb_dev_download(ctx: *mut mdb_ctx, host_dst: *mut c_void, dev_src: *const c_void, bytes: u64) -> c_int;
    pub fn mdb_dev_sync(ctx: *mut mdb_ctx) -> c_int;
    pub fn mdb_segments_upload(ctx: *mut mdb_ctx, host: *const mdb_segments, dev: *mut *mut mdb_segments_owned) -> c_int;
    pub fn mdb_segments_download(ctx: *mut mdb_ctx, dev: *const mdb_segments_owned,
                                 host: *mut *mut mdb_segments_owned) -> c_int;
    pub fn mdb_segments_free(segments: *mut mdb_segments_owned);
    pub fn mdb_segments_validate_dev(ctx: *mut mdb_ctx, dev: *const mdb_segments) -> c_int;

    // ---- grid (replaces the per-row loop of grid_exec.rs:323-356) ----------------------------------
    pub fn mdb_grid_count(ctx: *mut mdb_ctx, input: *const mdb_segments, n_out: *mut u64) -> c_int;
    pub fn mdb_grid_batch(ctx: *mut mdb_ctx, input: *const mdb_segments, out_ts: *mut i64, out_val: *mut f32,
                          out_rows_per_segment: *mut u32, cap: u64, n_out: *mut u64,
                          metrics: *mut mdb_grid_metrics) -> c_int;
    pub fn mdb_grid_count_dev(ctx: *mut mdb_ctx, input: *const mdb_segments, n_out: *mut u64) -> c_int;
    pub fn mdb_grid_batch_dev(ctx: *mut mdb_ctx, input: *const mdb_segments, out_ts: *mut i64, out_val: *mut f32,
                              out_rows_per_segment: *mut u32, cap: u64, n_out: *mut u64,
                              metrics: *mut mdb_grid_metrics) -> c_int;
    pub fn mdb_grid_count_range(ctx: *mut mdb_ctx, input: *const mdb_segments, t_lo: i64, t_hi: i64,
                                n_out: *mut u64) -> c_int;
    pub fn mdb_grid_batch_range(ctx: *mut mdb_ctx, input: *const mdb_segments, t_lo: i64, t_hi: i64,
                                out_ts: *mut i64, out_val: *mut f32, out_rows_per_segment: *mut u32, cap: u64,
                                n_out: *mut u64, metrics: *mut mdb_grid_metrics) -> c_int;
    pub fn mdb_grid_count_range_dev(ctx: *mut mdb_ctx, input: *const mdb_segments, t_lo: i64, t_hi: i64,
                                    n_out: *mut u64) -> c_int;
    pub fn mdb_grid_batch_range_dev(ctx: *mut mdb_ctx, input: *const mdb_segments, t_lo: i64, t_hi: i64,
                                    out_ts: *mut i64, out_val: *mut f32, out_rows_per_segment: *mut u32,
                                    cap: u64, n_out: *mut u64, metrics: *mut mdb_grid_metrics) -> c_int;
    pub fn mdb_grid_batch_owned(ctx: *mut mdb_ctx, input: *const mdb_segments, flags: u32, t_lo: i64, t_hi: i64,
                                reserve_front: u64, out: *mut *mut mdb_grid_result) -> c_int;
    pub fn mdb_grid_result_free(result: *mut mdb_grid_result);
    // (pipelined: several input batches per launch, two launches in flight, tag views replicated by the library)
    pub fn mdb_grid_submit(ctx: *mut mdb_ctx, inputs: *const mdb_grid_input, n_inputs: u32,
                           request: *const mdb_grid_request, ticket: *mut *mut mdb_grid_ticket) -> c_int;
    pub fn mdb_grid_wait(ticket: *mut mdb_grid_ticket, out: *mut *mut mdb_grid_result) -> c_int;
    pub fn mdb_grid_cancel(ticket: *mut mdb_grid_ticket);
    pub fn mdb_grid_result_tag_views(result: *const mdb_grid_result, column: u32) -> *mut mdb_view16;
    pub fn mdb_replicate_views(views: *const mdb_view16, rows_per_segment: *const u32, n_segments: u64,
                               buffer_shift: i32, out: *mut mdb_view16, out_cap: u64) -> c_int;

    // ---- the library's switches (read from the environment once per process; include/mdb.h) ----
    pub fn mdb_set_option(name: *const c_char, value: *const c_char) -> c_int;
    pub fn mdb_reload_options() -> c_int;
    pub fn mdb_option(name: *const c_char) -> *const c_char;

    // ---- aggregates (replace Model*Accumulator::update_batch, model_simple_aggregates.rs:345-587) ----
    pub fn mdb_agg_batch(ctx: *mut mdb_ctx, input: *const mdb_segments, which_mask: u32,
                         inout: *mut mdb_agg_state) -> c_int;
    pub fn mdb_agg_batch_dev(ctx: *mut mdb_ctx, input: *const mdb_segments, which_mask: u32,
                             inout: *mut mdb_agg_state) -> c_int;
    pub fn mdb_agg_batch_list(ctx: *mut mdb_ctx, inputs: *const *const mdb_segments, n_inputs: u32,
                              which_mask: u32, inout: *mut mdb_agg_state) -> c_int;
    pub fn mdb_agg_batch_range(ctx: *mut mdb_ctx, input: *const mdb_segments, t_lo: i64, t_hi: i64,
                               which_mask: u32, inout: *mut mdb_agg_state) -> c_int;
    pub fn mdb_agg_batch_range_dev(ctx: *mut mdb_ctx, input: *const mdb_segments, t_lo: i64, t_hi: i64,
                                   which_mask: u32, inout: *mut mdb_agg_state) -> c_int;
    pub fn mdb_agg_batch_range_list(ctx: *mut mdb_ctx, inputs: *const *const mdb_segments, n_inputs: u32,
                                    t_lo: i64, t_hi: i64, which_mask: u32, inout: *mut mdb_agg_state) -> c_int;

    // ---- fit (replaces try_compress_univariate_time_series, compression.rs:191-275) -----------------
    pub fn mdb_compress_series(ctx: *mut mdb_ctx, ts: *const i64, values: *const f32, n: u64,
                               error_bound: mdb_error_bound, out: *mut *mut mdb_segments_owned) -> c_int;
    pub fn mdb_compress_chunks(ctx: *mut mdb_ctx, ts: *const i64, values: *const f32, chunk_offsets: *const u64,
                               n_chunks: u64, error_bound: mdb_error_bound,
                               out: *mut *mut mdb_segments_owned) -> c_int;
    pub fn mdb_compress_chunk_list(ctx: *mut mdb_ctx, chunks: *const mdb_chunk, n_chunks: u64,
                                   error_bound: mdb_error_bound, out: *mut *mut mdb_segments_owned) -> c_int;
    pub fn mdb_compress_chunks_dev(ctx: *mut mdb_ctx, ts: *const i64, values: *const f32,
                                   chunk_offsets: *const u64, n_chunks: u64, error_bound: mdb_error_bound,
                                   regular_start: i64, regular_interval: i64, series_first_index: *const u64,
                                   out: *mut *mut mdb_segments_owned) -> c_int;
    pub fn mdb_split_and_compress_univariate(ctx: *mut mdb_ctx, ts: *const i64, field_values: *const *const f32,
                                             error_bounds: *const mdb_error_bound, n_fields: u32, n: u64,
                                             out: *mut *mut mdb_segments_owned) -> c_int;

    // ---- the crate's remaining public helpers (lib.rs:30-33), host arithmetic ------------------------
    pub fn mdb_is_value_within_error_bound(error_bound: mdb_error_bound, real_value: f32, approximate_value: f32,
                                           within: *mut i32) -> c_int;
    pub fn mdb_are_compressed_timestamps_regular(compressed_timestamps: *const u8, n_bytes: u64,
                                                 regular: *mut i32) -> c_int;

    // ---- multi-GPU: the final aggregate merge over RCCL / xGMI ---------------------------------------
    pub fn mdb_comm_unique_id(id_out: *mut c_void) -> c_int;
    pub fn mdb_comm_init(ctx: *mut mdb_ctx, rank: i32, world: i32, unique_id: *const c_void) -> c_int;
    pub fn mdb_comm_close(ctx: *mut mdb_ctx) -> c_int;
    pub fn mdb_agg_all_reduce(ctx: *mut mdb_ctx, inout: *mut mdb_agg_state, ranks_seen: *mut i32) -> c_int;
    pub fn mdb_agg_merge(into: *mut mdb_agg_state, from: *const mdb_agg_state) -> c_int;

    // ---- measurement ---------------------------------------------------------------------------------
    pub fn mdb_profile_enable(ctx: *mut mdb_ctx, enabled: c_int) -> c_int;
    pub fn mdb_profile_reset(ctx: *mut mdb_ctx) -> c_int;
    pub fn mdb_profile_get(ctx: *mut mdb_ctx, name: *const c_char, launches: *mut u64, total_ms: *mut f64) -> c_int;
    pub fn mdb_profile_names(ctx: *mut mdb_ctx, out: *mut c_char, cap: u64) -> c_int;
    pub fn mdb_synth_values_dev(ctx: *mut mdb_ctx, out: *mut f32, first_series: u64, n_series: u64,
                                n_per_series: u64, seed: u64) -> c_int;
}

// ---- layouts: the same numbers as the MDB_LAYOUT_ASSERT lines of include/mdb_format.h -----------------
const _: () = assert!(size_of::<mdb_error_bound>() == 8);
const _: () = assert!(offset_of!(mdb_error_bound, value) == 4);
const _: () = assert!(size_of::<mdb_view16>() == 16);
const _: () = assert!(offset_of!(mdb_view16, u) == 4);
const _: () = assert!(size_of::<mdb_binview_col>() == 32);
const _: () = assert!(offset_of!(mdb_binview_col, buffers) == 8);
const _: () = assert!(offset_of!(mdb_binview_col, buffer_sizes) == 16);
const _: () = assert!(offset_of!(mdb_binview_col, n_buffers) == 24);
const _: () = assert!(size_of::<mdb_segments>() == 144);
const _: () = assert!(offset_of!(mdb_segments, model_type_id) == 8);
const _: () = assert!(offset_of!(mdb_segments, start_time) == 16);
const _: () = assert!(offset_of!(mdb_segments, end_time) == 24);
const _: () = assert!(offset_of!(mdb_segments, timestamps) == 32);
const _: () = assert!(offset_of!(mdb_segments, min_value) == 64);
const _: () = assert!(offset_of!(mdb_segments, max_value) == 72);
const _: () = assert!(offset_of!(mdb_segments, values) == 80);
const _: () = assert!(offset_of!(mdb_segments, residuals) == 112);
const _: () = assert!(size_of::<mdb_grid_metrics>() == 80);
const _: () = assert!(offset_of!(mdb_grid_metrics, rows_created_by_model_type) == 8);
const _: () = assert!(offset_of!(mdb_grid_metrics, segments_with_residuals) == 32);
const _: () = assert!(offset_of!(mdb_grid_metrics, segments_with_model_type) == 40);
const _: () = assert!(offset_of!(mdb_grid_metrics, segments_regular) == 64);
const _: () = assert!(offset_of!(mdb_grid_metrics, segments_irregular) == 72);
const _: () = assert!(size_of::<mdb_agg_state>() == 24);
const _: () = assert!(offset_of!(mdb_agg_state, count) == 8);
const _: () = assert!(offset_of!(mdb_agg_state, min) == 16);
const _: () = assert!(offset_of!(mdb_agg_state, max) == 20);
const _: () = assert!(size_of::<mdb_segments_owned>() == 176);
const _: () = assert!(offset_of!(mdb_segments_owned, error) == 144);
const _: () = assert!(offset_of!(mdb_segments_owned, chunk_index) == 152);
const _: () = assert!(offset_of!(mdb_segments_owned, on_device) == 160);
const _: () = assert!(offset_of!(mdb_segments_owned, priv_) == 168);
const _: () = assert!(size_of::<mdb_grid_result>() == 136);
const _: () = assert!(offset_of!(mdb_grid_result, values) == 8);
const _: () = assert!(offset_of!(mdb_grid_result, rows_per_segment) == 16);
const _: () = assert!(offset_of!(mdb_grid_result, n) == 24);
const _: () = assert!(offset_of!(mdb_grid_result, n_segments) == 32);
const _: () = assert!(offset_of!(mdb_grid_result, reserved_front) == 40);
const _: () = assert!(offset_of!(mdb_grid_result, metrics) == 48);
const _: () = assert!(offset_of!(mdb_grid_result, priv_) == 128);
const _: () = assert!(size_of::<mdb_grid_input>() == 160);
const _: () = assert!(offset_of!(mdb_grid_input, tag_views) == 144);
const _: () = assert!(offset_of!(mdb_grid_input, tag_buffer_shift) == 152);
const _: () = assert!(size_of::<mdb_grid_request>() == 32);
const _: () = assert!(offset_of!(mdb_grid_request, n_tag_columns) == 4);
const _: () = assert!(offset_of!(mdb_grid_request, t_lo) == 8);
const _: () = assert!(offset_of!(mdb_grid_request, t_hi) == 16);
const _: () = assert!(offset_of!(mdb_grid_request, reserve_front) == 24);
const _: () = assert!(size_of::<mdb_chunk>() == 24);
const _: () = assert!(offset_of!(mdb_chunk, values) == 8);
const _: () = assert!(offset_of!(mdb_chunk, n) == 16);
