//! Safe wrappers over `libmdb_hip.so` for the call sites of ModelarDB-RS that this library replaces
//! (see `rust/patches/`): `GridStream::poll_next` / `grid_and_append_to_leftovers_in_current_batch`
//! (crates/modelardb_storage/src/query/grid_exec.rs:261-429) through [`Context::grid_submit`] and
//! [`GridTicket::wait`]; the `Model*Accumulator::update_batch` methods
//! (crates/modelardb_storage/src/optimizer/model_simple_aggregates.rs:345-587) through
//! [`Context::aggregate`]; `try_compress_multivariate_time_series` and
//! `try_split_and_compress_univariate_time_series` (crates/modelardb_compression/src/compression.rs:42-179)
//! and `UncompressedDataManager::process_compressor_messages`
//! (crates/modelardb_server/src/storage/uncompressed_data_manager.rs:505-596) through
//! [`Context::compress_chunks`]: every series and field of a batch, or every finished ingest buffer of a
//! drained channel, in ONE launch per error bound (the fitter needs more than 10^4 chunks per launch to
//! beat one CPU thread; one launch per series does not).
//!
//! Every decision lives behind the C ABI; this crate only converts between Arrow arrays and the
//! pointer structs of `include/mdb_format.h`, owns the handles, and turns the `0 / 1 + last error`
//! convention into `Result`s. It copies no payload bytes on the way in.

pub mod sys;

use std::ffi::CStr;
use std::fmt::{Display, Formatter};
use std::iter;
use std::ptr::{self, NonNull};
use std::sync::Arc;

use arrow::array::{
    Array, ArrayRef, BinaryViewArray, Float32Array, Int8Array, Int16Array, StringViewArray,
};
use arrow::buffer::{Buffer, ScalarBuffer};
use arrow::datatypes::Schema;
use arrow::record_batch::RecordBatch;
use modelardb_types::types::{ErrorBound, TimestampArray, ValueArray};

pub use sys::{mdb_agg_state as AggState, mdb_grid_metrics as GridMetrics};
pub use sys::{MDB_AGG_AVG, MDB_AGG_COUNT, MDB_AGG_MAX, MDB_AGG_MIN, MDB_AGG_SUM};

/// Failure reported by the library (the text of `mdb_last_error()`).
#[derive(Debug, Clone)]
pub struct HipError(pub String);

impl Display for HipError {
    fn fmt(&self, f: &mut Formatter) -> std::fmt::Result {
        write!(f, "HIP Error: {}", self.0)
    }
}

impl std::error::Error for HipError {}

pub type Result<T> = std::result::Result<T, HipError>;

fn check(code: i32) -> Result<()> {
    if code == 0 {
        return Ok(());
    }
    // Valid until the next failing call on this thread (capi.rs:58-80 convention).
    let message = unsafe { CStr::from_ptr(sys::mdb_last_error()) };
    Err(HipError(message.to_string_lossy().into_owned()))
}

/// One `mdb_ctx`: a HIP stream plus the scratch memory the batches seen so far needed. Calls on one
/// context are serialised inside the library, so give every `GridStream`, accumulator and
/// compression thread its own.
pub struct Context(NonNull<sys::mdb_ctx>);

// The library locks the context's mutex in every entry point.
unsafe impl Send for Context {}
unsafe impl Sync for Context {}

impl Context {
    pub fn new(device: i32) -> Result<Self> {
        let mut raw = ptr::null_mut();
        check(unsafe { sys::mdb_init(device, &mut raw) })?;
        Ok(Self(NonNull::new(raw).expect("mdb_init returned success and a null context")))
    }

    fn raw(&self) -> *mut sys::mdb_ctx {
        self.0.as_ptr()
    }

    /// Give the grown scratch and staging memory back (call after an unusually large batch).
    pub fn trim(&self) -> Result<u64> {
        let mut released = 0u64;
        check(unsafe { sys::mdb_trim(self.raw(), &mut released) })?;
        Ok(released)
    }

    /// Device scratch beyond `bytes` is given back after every call on this context (0 keeps everything):
    /// for owners of many contexts, e.g. one `GridStream` per field column.
    pub fn set_scratch_limit(&self, bytes: u64) -> Result<()> {
        check(unsafe { sys::mdb_set_scratch_limit(self.raw(), bytes) })
    }

    /// `ncclCommInitRank` for the final aggregate merge; `unique_id` comes from
    /// [`comm_unique_id`] on one rank.
    pub fn comm_init(&self, rank: i32, world: i32, unique_id: &[u8; sys::MDB_COMM_ID_BYTES]) -> Result<()> {
        check(unsafe { sys::mdb_comm_init(self.raw(), rank, world, unique_id.as_ptr().cast()) })
    }

    /// Merge the partial aggregate states of all ranks (one 32-byte all-gather over RCCL and a fold
    /// in rank order, so every rank gets the same f64 sum, run after run).
    pub fn agg_all_reduce(&self, state: &mut AggState) -> Result<i32> {
        let mut ranks_seen = 0i32;
        check(unsafe { sys::mdb_agg_all_reduce(self.raw(), state, &mut ranks_seen) })?;
        Ok(ranks_seen)
    }
}

impl Drop for Context {
    fn drop(&mut self) {
        unsafe { sys::mdb_close(self.raw()) };
    }
}

pub fn comm_unique_id() -> Result<[u8; sys::MDB_COMM_ID_BYTES]> {
    let mut id = [0u8; sys::MDB_COMM_ID_BYTES];
    check(unsafe { sys::mdb_comm_unique_id(id.as_mut_ptr().cast()) })?;
    Ok(id)
}

impl From<ErrorBound> for sys::mdb_error_bound {
    fn from(error_bound: ErrorBound) -> Self {
        match error_bound {
            ErrorBound::Lossless => Self { kind: sys::MDB_EB_LOSSLESS, value: 0.0 },
            ErrorBound::Absolute(value) => Self { kind: sys::MDB_EB_ABSOLUTE, value },
            ErrorBound::Relative(value) => Self { kind: sys::MDB_EB_RELATIVE, value },
        }
    }
}

/// `modelardb_compression::is_value_within_error_bound` (models/mod.rs:53-77).
pub fn is_value_within_error_bound(error_bound: ErrorBound, real_value: f32, approximate_value: f32) -> bool {
    let mut within = 0i32;
    check(unsafe {
        sys::mdb_is_value_within_error_bound(error_bound.into(), real_value, approximate_value, &mut within)
    })
    .expect("a valid ErrorBound cannot be rejected");
    within != 0
}

/// `modelardb_compression::are_compressed_timestamps_regular` (models/timestamps.rs:199-202).
pub fn are_compressed_timestamps_regular(compressed_timestamps: &[u8]) -> bool {
    let mut regular = 0i32;
    check(unsafe {
        sys::mdb_are_compressed_timestamps_regular(
            compressed_timestamps.as_ptr(),
            compressed_timestamps.len() as u64,
            &mut regular,
        )
    })
    .expect("cannot fail for a slice");
    regular != 0
}

/// Borrowed `mdb_segments` view of the eight segment columns. Keeps the small pointer / size tables
/// the view refers to; the Arrow buffers themselves stay owned by the arrays.
pub struct SegmentsView<'a> {
    raw: sys::mdb_segments,
    _tables: Box<[(Vec<*const u8>, Vec<i64>); 3]>,
    _arrays: std::marker::PhantomData<&'a ()>,
}

impl<'a> SegmentsView<'a> {
    /// From the typed arrays `modelardb_types::arrays!` / `value!` yield (columns 0..=7 of
    /// `QUERY_COMPRESSED_SCHEMA`, crates/modelardb_types/src/schemas.rs:40-52).
    #[allow(clippy::too_many_arguments)]
    pub fn new(
        model_type_ids: &'a Int8Array,
        start_times: &'a TimestampArray,
        end_times: &'a TimestampArray,
        timestamps: &'a BinaryViewArray,
        min_values: &'a ValueArray,
        max_values: &'a ValueArray,
        values: &'a BinaryViewArray,
        residuals: &'a BinaryViewArray,
    ) -> Self {
        let mut tables: Box<[(Vec<*const u8>, Vec<i64>); 3]> = Box::default();
        let mut column = |index: usize, array: &BinaryViewArray| {
            let (pointers, sizes) = &mut tables[index];
            for buffer in array.data_buffers() {
                pointers.push(buffer.as_ptr());
                sizes.push(buffer.len() as i64);
            }
            sys::mdb_binview_col {
                views: array.views().as_ptr().cast(),
                buffers: pointers.as_ptr(),
                buffer_sizes: sizes.as_ptr(),
                n_buffers: pointers.len() as i32,
            }
        };
        let raw = sys::mdb_segments {
            n: model_type_ids.len() as u64,
            model_type_id: model_type_ids.values().as_ptr(),
            start_time: start_times.values().as_ptr(),
            end_time: end_times.values().as_ptr(),
            timestamps: column(0, timestamps),
            min_value: min_values.values().as_ptr(),
            max_value: max_values.values().as_ptr(),
            values: column(1, values),
            residuals: column(2, residuals),
        };
        Self { raw, _tables: tables, _arrays: std::marker::PhantomData }
    }

    /// From a batch whose first eight columns follow `QUERY_COMPRESSED_SCHEMA` (what `GridStream` and
    /// the accumulators are handed; tag columns may follow).
    pub fn from_record_batch(batch: &'a RecordBatch) -> Self {
        fn column<'b, T: 'static>(batch: &'b RecordBatch, index: usize) -> &'b T {
            batch
                .column(index)
                .as_any()
                .downcast_ref::<T>()
                .expect("The batch should follow QUERY_COMPRESSED_SCHEMA.")
        }
        Self::new(
            column::<Int8Array>(batch, 0),
            column::<TimestampArray>(batch, 1),
            column::<TimestampArray>(batch, 2),
            column::<BinaryViewArray>(batch, 3),
            column::<ValueArray>(batch, 4),
            column::<ValueArray>(batch, 5),
            column::<BinaryViewArray>(batch, 6),
            column::<BinaryViewArray>(batch, 7),
        )
    }
}

/// The block of page-locked memory `mdb_grid_batch_owned` reconstructed a batch into. Arrow buffers
/// made from it keep it alive; the last one to go returns it to the library's pool.
struct GridBlock(NonNull<sys::mdb_grid_result>);

unsafe impl Send for GridBlock {}
unsafe impl Sync for GridBlock {}
impl std::panic::RefUnwindSafe for GridBlock {}

impl Drop for GridBlock {
    fn drop(&mut self) {
        unsafe { sys::mdb_grid_result_free(self.0.as_ptr()) };
    }
}

/// What one call of [`Context::grid`] produced.
pub struct GridOutput {
    /// `leftovers.len() + rows created` timestamps: the leftovers first (grid_exec.rs:302-320).
    pub timestamps: TimestampArray,
    pub values: ValueArray,
    /// Data points each segment row reconstructed to, for the tag replication of grid_exec.rs:341-346.
    pub rows_per_segment: Vec<u32>,
    pub metrics: GridMetrics,
}

impl Context {
    /// Replaces the per-row `modelardb_compression::grid` loop of grid_exec.rs:323-356 for a whole
    /// batch: one upload of the segment columns, the kernels, one copy of the reconstructed columns
    /// into page-locked memory, which the returned arrays wrap without copying. `leftover_*` are the
    /// rows of the current batch that have not been handed out yet; they are placed in front.
    /// `time_range`: `Some((lo, hi))` reconstructs only `lo <= timestamp <= hi`.
    pub fn grid(
        &self,
        segments: &SegmentsView,
        leftover_timestamps: &[i64],
        leftover_values: &[f32],
        time_range: Option<(i64, i64)>,
    ) -> Result<GridOutput> {
        assert_eq!(leftover_timestamps.len(), leftover_values.len());
        let leftovers = leftover_timestamps.len();
        let (flags, t_lo, t_hi) = match time_range {
            Some((lo, hi)) => (sys::MDB_GRID_HAS_RANGE, lo, hi),
            None => (0, 0, 0),
        };
        let mut raw = ptr::null_mut();
        check(unsafe {
            sys::mdb_grid_batch_owned(self.raw(), &segments.raw, flags, t_lo, t_hi, leftovers as u64, &mut raw)
        })?;
        let block = Arc::new(GridBlock(NonNull::new(raw).expect("success with a null result")));
        let result = unsafe { block.0.as_ref() };
        let total = leftovers + result.n as usize;
        let rows_per_segment =
            unsafe { std::slice::from_raw_parts(result.rows_per_segment, result.n_segments as usize) }.to_vec();
        let (timestamps, values) = unsafe {
            // `reserve_front` rows of writable room sit in front of the new points.
            let first_timestamp = result.timestamps.sub(leftovers);
            let first_value = result.values.sub(leftovers);
            ptr::copy_nonoverlapping(leftover_timestamps.as_ptr(), first_timestamp, leftovers);
            ptr::copy_nonoverlapping(leftover_values.as_ptr(), first_value, leftovers);
            let timestamps = Buffer::from_custom_allocation(
                NonNull::new_unchecked(first_timestamp.cast::<u8>()),
                8 * total,
                block.clone(),
            );
            let values =
                Buffer::from_custom_allocation(NonNull::new_unchecked(first_value.cast::<u8>()), 4 * total, block.clone());
            (timestamps, values)
        };
        Ok(GridOutput {
            timestamps: TimestampArray::new(ScalarBuffer::new(timestamps, 0, total), None),
            values: ValueArray::new(ScalarBuffer::new(values, 0, total), None),
            rows_per_segment,
            metrics: result.metrics,
        })
    }

    /// Replaces the per-row `len` / `sum` loops of the accumulators: folds the batch into `state`
    /// for the aggregates in `which_mask` (`MDB_AGG_*`).
    pub fn aggregate(&self, segments: &SegmentsView, which_mask: u32, state: &mut AggState) -> Result<()> {
        check(unsafe { sys::mdb_agg_batch(self.raw(), &segments.raw, which_mask, state) })
    }

    /// The same for several batches at once (rows in the order of the slice): uploaded and reduced as ONE
    /// batch, so that accumulators which are handed 8192 segments at a time can fold hundreds of
    /// thousands with one call (patches/0002-model_simple_aggregates.patch: `PendingSegments`).
    pub fn aggregate_list(&self, segments: &[SegmentsView], which_mask: u32, state: &mut AggState) -> Result<()> {
        let inputs: Vec<*const sys::mdb_segments> = segments.iter().map(|view| &view.raw as *const _).collect();
        check(unsafe { sys::mdb_agg_batch_list(self.raw(), inputs.as_ptr(), inputs.len() as u32, which_mask, state) })
    }

    /// The same restricted to `t_lo <= timestamp <= t_hi`, without materialising a data point.
    pub fn aggregate_range(
        &self,
        segments: &SegmentsView,
        t_lo: i64,
        t_hi: i64,
        which_mask: u32,
        state: &mut AggState,
    ) -> Result<()> {
        check(unsafe { sys::mdb_agg_batch_range(self.raw(), &segments.raw, t_lo, t_hi, which_mask, state) })
    }

    /// [`Context::aggregate_list`] restricted to `t_lo <= timestamp <= t_hi`: what the accumulators of a query
    /// with a range on the timestamp fold their pending batches with
    /// (patches/0002-model_simple_aggregates.patch: `PendingSegments::fold_into`).
    pub fn aggregate_range_list(
        &self,
        segments: &[SegmentsView],
        t_lo: i64,
        t_hi: i64,
        which_mask: u32,
        state: &mut AggState,
    ) -> Result<()> {
        let inputs: Vec<*const sys::mdb_segments> = segments.iter().map(|view| &view.raw as *const _).collect();
        check(unsafe {
            sys::mdb_agg_batch_range_list(self.raw(), inputs.as_ptr(), inputs.len() as u32, t_lo, t_hi, which_mask, state)
        })
    }

    /// Replaces the body of `try_compress_univariate_time_series` after its two argument checks
    /// (compression.rs:202-211): fits PMC-Mean / Swing / MacaqueV on the GPU and builds the batch
    /// `CompressedSegmentBatchBuilder::finish` builds (types.rs:492-516).
    pub fn compress_univariate(
        &self,
        uncompressed_timestamps: &TimestampArray,
        uncompressed_values: &ValueArray,
        error_bound: ErrorBound,
        compressed_schema: Arc<Schema>,
        tag_values: &[String],
        field_column_index: i16,
    ) -> Result<RecordBatch> {
        // compression.rs:202-206: the library reads `len` timestamps and `len` values.
        if uncompressed_timestamps.len() != uncompressed_values.len() {
            return Err(HipError(
                "Uncompressed timestamps and uncompressed values have different lengths.".to_owned(),
            ));
        }
        let mut raw = ptr::null_mut();
        check(unsafe {
            sys::mdb_compress_series(
                self.raw(),
                uncompressed_timestamps.values().as_ptr(),
                uncompressed_values.values().as_ptr(),
                uncompressed_values.len() as u64,
                error_bound.into(),
                &mut raw,
            )
        })?;
        let owned = Arc::new(OwnedSegments(NonNull::new(raw).expect("success with a null result")));
        let rows = unsafe { owned.0.as_ref() }.seg.n as usize;
        record_batch_of_rows(&owned.whole_columns()?, 0, rows, compressed_schema, tag_values, field_column_index)
    }

    /// Compresses MANY univariate time series with one launch per distinct error bound and returns one
    /// batch of segments per chunk, in the order of `chunks`. This is the form the fitter is fast in: a
    /// launch fits one chunk per lane, so 10^4 chunks take about as long as one. The chunks are read where
    /// they lie (slices of the caller's arrays); chunks that share a timestamp array - the fields of one
    /// series - have it looked at once.
    pub fn compress_chunks(&self, chunks: &[SeriesChunk]) -> Result<Vec<RecordBatch>> {
        let mut compressed: Vec<Option<RecordBatch>> = vec![None; chunks.len()];
        let mut done = vec![false; chunks.len()];
        for first in 0..chunks.len() {
            if done[first] {
                continue;
            }
            // Every chunk with this error bound (an error bound is an argument of the launch).
            let bound: sys::mdb_error_bound = chunks[first].error_bound.into();
            let mut group = Vec::new();
            let mut list = Vec::new();
            for index in first..chunks.len() {
                let chunk = &chunks[index];
                let chunk_bound: sys::mdb_error_bound = chunk.error_bound.into();
                if done[index] || chunk_bound != bound {
                    continue;
                }
                if chunk.timestamps.len() != chunk.values.len() {
                    return Err(HipError(
                        "Uncompressed timestamps and uncompressed values have different lengths.".to_owned(),
                    ));
                }
                done[index] = true;
                group.push(index);
                list.push(sys::mdb_chunk {
                    ts: chunk.timestamps.as_ptr(),
                    values: chunk.values.as_ptr(),
                    n: chunk.values.len() as u64,
                });
            }
            let mut raw = ptr::null_mut();
            check(unsafe { sys::mdb_compress_chunk_list(self.raw(), list.as_ptr(), list.len() as u64, bound, &mut raw) })?;
            let owned = Arc::new(OwnedSegments(NonNull::new(raw).expect("success with a null result")));
            // Segments come back grouped by chunk, in chunk order.
            let (rows, chunk_index) = unsafe {
                let c = owned.0.as_ref();
                if c.seg.n > 0 && c.chunk_index.is_null() {
                    return Err(HipError("The library returned segments without their chunk index.".to_owned()));
                }
                (c.seg.n as usize, std::slice::from_raw_parts(c.chunk_index, c.seg.n as usize))
            };
            let whole = owned.whole_columns()?; // (validated once; every chunk's batch is a slice of it)
            let mut row = 0;
            for (position, &index) in group.iter().enumerate() {
                let begin = row;
                while row < rows && chunk_index[row] as usize == position {
                    row += 1;
                }
                let chunk = &chunks[index];
                compressed[index] = Some(record_batch_of_rows(
                    &whole,
                    begin,
                    row - begin,
                    chunk.compressed_schema.clone(),
                    chunk.tag_values,
                    chunk.field_column_index,
                )?);
            }
        }
        Ok(compressed.into_iter().map(|batch| batch.expect("every chunk is in one group")).collect())
    }

    /// `try_split_and_compress_univariate_time_series` (compression.rs:147-179): the field columns of ONE
    /// time series, which share its timestamps; `fields` = (values, error bound, field column index).
    pub fn split_and_compress(
        &self,
        uncompressed_timestamps: &TimestampArray,
        fields: &[(&ValueArray, ErrorBound, i16)],
        compressed_schema: Arc<Schema>,
        tag_values: &[String],
    ) -> Result<Vec<RecordBatch>> {
        let chunks: Vec<SeriesChunk> = fields
            .iter()
            .map(|(values, error_bound, field_column_index)| SeriesChunk {
                timestamps: uncompressed_timestamps.values(),
                values: values.values(),
                error_bound: *error_bound,
                compressed_schema: compressed_schema.clone(),
                tag_values,
                field_column_index: *field_column_index,
            })
            .collect();
        self.compress_chunks(&chunks)
    }
}

/// One univariate time series to compress: its sorted data points where they lie, and what the resulting
/// batch of segments is labelled with (`CompressedSegmentBatchBuilder::new`, types.rs:444-470).
pub struct SeriesChunk<'a> {
    pub timestamps: &'a [i64],
    pub values: &'a [f32],
    pub error_bound: ErrorBound,
    pub compressed_schema: Arc<Schema>,
    pub tag_values: &'a [String],
    pub field_column_index: i16,
}

/// Number of columns of `QUERY_COMPRESSED_SCHEMA` (crates/modelardb_types/src/schemas.rs:40-52); the tag
/// columns of a batch of segments follow them.
const QUERY_COMPRESSED_COLUMNS: usize = 9;

/// An outstanding [`Context::grid_submit`]: the library reconstructs the batches on one of its worker
/// threads while the caller goes on (polls its input for the next batches, hands out slices of the
/// previous result). Keeps the input batches alive, since the library reads their Arrow buffers until
/// [`GridTicket::wait`] returns. Dropping a ticket waits for the job and frees what it made.
pub struct GridTicket {
    raw: Option<NonNull<sys::mdb_grid_ticket>>,
    batches: Vec<RecordBatch>,
    n_tag_columns: usize,
}

// The ticket is only a handle; the library synchronises the job behind it.
unsafe impl Send for GridTicket {}

impl Drop for GridTicket {
    fn drop(&mut self) {
        if let Some(raw) = self.raw.take() {
            unsafe { sys::mdb_grid_cancel(raw.as_ptr()) };
        }
    }
}

/// What [`GridTicket::wait`] returns: the columns of the new current batch of a `GridStream`.
pub struct PipelinedGridOutput {
    /// Leftovers first, then the reconstructed data points (grid_exec.rs:302-320).
    pub timestamps: TimestampArray,
    pub values: ValueArray,
    /// One array per tag column: every segment's tag once per data point reconstructed from it
    /// (grid_exec.rs:339-346), the leftovers' tags in front. The strings are not copied: long ones stay in
    /// the data buffers of the input batches, which the arrays share.
    pub tags: Vec<StringViewArray>,
    pub metrics: GridMetrics,
    /// Segments and data points of this submit (a stream sizes its next submit by them).
    pub segments: u64,
    pub rows_created: u64,
}

impl Context {
    /// Starts the reconstruction of ALL segments of `batches` (each with the columns of
    /// `QUERY_COMPRESSED_SCHEMA` followed by the table's tag columns) as one launch and returns at once.
    /// `reserve_front`: room in front of the result for the rows the stream has left (`batch_size` is
    /// enough: a stream only asks for more when it has fewer than that). Replaces, together with
    /// [`GridTicket::wait`], grid_exec.rs:261-364 for several input batches at once.
    pub fn grid_submit(
        &self,
        batches: Vec<RecordBatch>,
        reserve_front: usize,
        time_range: Option<(i64, i64)>,
    ) -> Result<GridTicket> {
        assert!(!batches.is_empty(), "grid_submit needs at least one batch.");
        let n_tag_columns = batches[0].num_columns() - QUERY_COMPRESSED_COLUMNS;
        let views: Vec<SegmentsView> = batches.iter().map(SegmentsView::from_record_batch).collect();
        // Per batch: the views of its tag arrays, and where its data buffers start in the output tag
        // column's list of buffers (buffer 0 is the ticket's own, for long leftover strings).
        let mut tag_views: Vec<Vec<*const sys::mdb_view16>> = Vec::with_capacity(batches.len());
        let mut tag_shifts: Vec<Vec<i32>> = Vec::with_capacity(batches.len());
        let mut next_buffer = vec![1i32; n_tag_columns];
        for batch in &batches {
            assert_eq!(batch.num_columns(), QUERY_COMPRESSED_COLUMNS + n_tag_columns);
            let mut views_of_batch = Vec::with_capacity(n_tag_columns);
            let mut shifts_of_batch = Vec::with_capacity(n_tag_columns);
            for tag_column in 0..n_tag_columns {
                let tags = tag_array(batch, tag_column);
                views_of_batch.push(tags.views().as_ptr().cast::<sys::mdb_view16>());
                shifts_of_batch.push(next_buffer[tag_column]);
                next_buffer[tag_column] += tags.data_buffers().len() as i32;
            }
            tag_views.push(views_of_batch);
            tag_shifts.push(shifts_of_batch);
        }
        let inputs: Vec<sys::mdb_grid_input> = (0..batches.len())
            .map(|index| sys::mdb_grid_input {
                segments: views[index].raw,
                tag_views: if n_tag_columns > 0 { tag_views[index].as_ptr() } else { ptr::null() },
                tag_buffer_shift: if n_tag_columns > 0 { tag_shifts[index].as_ptr() } else { ptr::null() },
            })
            .collect();
        let (flags, t_lo, t_hi) = match time_range {
            Some((lo, hi)) => (sys::MDB_GRID_HAS_RANGE, lo, hi),
            None => (0, 0, 0),
        };
        let request = sys::mdb_grid_request {
            flags,
            n_tag_columns: n_tag_columns as u32,
            t_lo,
            t_hi,
            reserve_front: reserve_front as u64,
        };
        let mut raw = ptr::null_mut();
        // The structs and the small tables above are copied by the library before this returns; the
        // Arrow buffers behind them are kept alive by `batches` in the ticket.
        check(unsafe {
            sys::mdb_grid_submit(self.raw(), inputs.as_ptr(), inputs.len() as u32, &request, &mut raw)
        })?;
        drop(views);
        Ok(GridTicket {
            raw: Some(NonNull::new(raw).expect("success with a null ticket")),
            batches,
            n_tag_columns,
        })
    }
}

fn tag_array(batch: &RecordBatch, tag_column: usize) -> &StringViewArray {
    batch
        .column(QUERY_COMPRESSED_COLUMNS + tag_column)
        .as_any()
        .downcast_ref::<StringViewArray>()
        .expect("The tag columns of a batch of segments should be StringViewArrays.")
}

impl GridTicket {
    /// Segment rows this ticket was made from.
    pub fn segments(&self) -> usize {
        self.batches.iter().map(RecordBatch::num_rows).sum()
    }

    /// Blocks until the batches are reconstructed and returns the columns of the stream's new current
    /// batch: `leftover_*` (the rows of the current batch that have not been handed out, fewer than the
    /// `reserve_front` of the submit) in front, the new data points behind them. The arrays wrap memory of
    /// the library (page-locked for the data points) without copying; the last of them to be dropped
    /// returns it to the library's pools.
    pub fn wait(
        mut self,
        leftover_timestamps: &[i64],
        leftover_values: &[f32],
        leftover_tags: &[StringViewArray],
    ) -> Result<PipelinedGridOutput> {
        assert_eq!(leftover_timestamps.len(), leftover_values.len());
        assert_eq!(leftover_tags.len(), self.n_tag_columns);
        let leftovers = leftover_timestamps.len();
        let ticket = self.raw.take().expect("a ticket is waited for once");
        let mut raw = ptr::null_mut();
        check(unsafe { sys::mdb_grid_wait(ticket.as_ptr(), &mut raw) })?;
        let block = Arc::new(GridBlock(NonNull::new(raw).expect("success with a null result")));
        let result = unsafe { block.0.as_ref() };
        assert!(leftovers as u64 <= result.reserved_front, "more leftovers than room was reserved for");
        let total = leftovers + result.n as usize;
        let (timestamps, values) = unsafe {
            let first_timestamp = result.timestamps.sub(leftovers);
            let first_value = result.values.sub(leftovers);
            ptr::copy_nonoverlapping(leftover_timestamps.as_ptr(), first_timestamp, leftovers);
            ptr::copy_nonoverlapping(leftover_values.as_ptr(), first_value, leftovers);
            (
                Buffer::from_custom_allocation(
                    NonNull::new_unchecked(first_timestamp.cast::<u8>()),
                    8 * total,
                    block.clone(),
                ),
                Buffer::from_custom_allocation(NonNull::new_unchecked(first_value.cast::<u8>()), 4 * total, block.clone()),
            )
        };
        let mut tags = Vec::with_capacity(self.n_tag_columns);
        for (tag_column, leftover) in leftover_tags.iter().enumerate() {
            assert_eq!(leftover.len(), leftovers);
            // The leftovers' views in front of the replicated ones. Their long strings are copied into a
            // buffer of the new array (buffer 0), so that old inputs can be dropped.
            let mut leftover_payload: Vec<u8> = Vec::new();
            let first_view = unsafe {
                let replicated = sys::mdb_grid_result_tag_views(result, tag_column as u32);
                assert!(!replicated.is_null(), "the result has fewer tag columns than the submit asked for");
                let first_view = replicated.sub(leftovers);
                for row in 0..leftovers {
                    let value = leftover.value(row).as_bytes();
                    let mut view = sys::mdb_view16 { length: value.len() as i32, u: [0; 12] };
                    if value.len() <= 12 {
                        view.u[..value.len()].copy_from_slice(value);
                    } else {
                        view.u[..4].copy_from_slice(&value[..4]);
                        view.u[4..8].copy_from_slice(&0i32.to_le_bytes());
                        view.u[8..12].copy_from_slice(&(leftover_payload.len() as i32).to_le_bytes());
                        leftover_payload.extend_from_slice(value);
                    }
                    first_view.add(row).write(view);
                }
                first_view
            };
            let views = unsafe {
                Buffer::from_custom_allocation(NonNull::new_unchecked(first_view.cast::<u8>()), 16 * total, block.clone())
            };
            // The same order `grid_submit` numbered them in (`tag_buffer_shift`).
            let mut buffers = vec![Buffer::from_vec(leftover_payload)];
            for batch in &self.batches {
                buffers.extend(tag_array(batch, tag_column).data_buffers().iter().cloned());
            }
            // Every view is a copy of a view of a valid StringViewArray with its buffer index moved to where
            // that buffer is in `buffers`, or was written above from a &str: valid by construction, and
            // validating 16 bytes per data point again would cost more than reconstructing the point.
            tags.push(unsafe { StringViewArray::new_unchecked(ScalarBuffer::new(views, 0, total), buffers, None) });
        }
        Ok(PipelinedGridOutput {
            timestamps: TimestampArray::new(ScalarBuffer::new(timestamps, 0, total), None),
            values: ValueArray::new(ScalarBuffer::new(values, 0, total), None),
            tags,
            metrics: result.metrics,
            segments: result.n_segments,
            rows_created: result.n,
        })
    }
}

/// Segments returned by the compressor (host memory), freed when the last array made from them is dropped.
struct OwnedSegments(NonNull<sys::mdb_segments_owned>);

unsafe impl Send for OwnedSegments {}
unsafe impl Sync for OwnedSegments {}
impl std::panic::RefUnwindSafe for OwnedSegments {}

impl Drop for OwnedSegments {
    fn drop(&mut self) {
        unsafe { sys::mdb_segments_free(self.0.as_ptr()) };
    }
}

impl OwnedSegments {
    /// The nine segment columns over ALL rows of the result, wrapping the library's memory (kept alive by
    /// `self`). Built - and the three BinaryView columns validated - ONCE per result: a result of one launch
    /// is cut into a batch per chunk, and validating every view of the joint result for every chunk would
    /// cost (chunks x segments) view checks where (segments) suffice.
    fn whole_columns(self: &Arc<Self>) -> Result<Vec<ArrayRef>> {
        let owned = unsafe { self.0.as_ref() };
        let segments = &owned.seg;
        let n = segments.n as usize;
        if n > 0 && owned.error.is_null() {
            return Err(HipError("The library returned segments without their error column.".to_owned()));
        }
        let wrap = |pointer: *const u8, bytes: usize| -> Buffer {
            match NonNull::new(pointer.cast_mut()) {
                Some(pointer) if bytes > 0 => unsafe { Buffer::from_custom_allocation(pointer, bytes, self.clone()) },
                _ => Buffer::from(Vec::<u8>::new()),
            }
        };
        let binary_view = |column: &sys::mdb_binview_col| -> Result<ArrayRef> {
            if column.n_buffers < 0 || (column.n_buffers > 0 && (column.buffers.is_null() || column.buffer_sizes.is_null())) {
                return Err(HipError("The library returned a malformed BinaryView column.".to_owned()));
            }
            let mut buffers = Vec::with_capacity(column.n_buffers as usize);
            for index in 0..column.n_buffers as usize {
                let (pointer, size) = unsafe { (*column.buffers.add(index), *column.buffer_sizes.add(index)) };
                buffers.push(wrap(pointer, size as usize));
            }
            // A ScalarBuffer<u128> must be 16-byte aligned. The library's views are (every column of a result
            // begins on a 16-byte boundary of an allocation of the C++ runtime or at a multiple of 64 bytes in
            // one); a pointer that is not is copied instead of trusted.
            let view_bytes = if (column.views as usize) % std::mem::align_of::<u128>() == 0 {
                wrap(column.views.cast(), 16 * n)
            } else {
                let copied: Vec<u128> = (0..n)
                    .map(|row| unsafe { ptr::read_unaligned(column.views.cast::<u128>().add(row)) })
                    .collect();
                Buffer::from_vec(copied)
            };
            let views = ScalarBuffer::<u128>::new(view_bytes, 0, n);
            // try_new checks every view against `buffers` (length, buffer index, offset, prefix).
            let array = BinaryViewArray::try_new(views, buffers, None).map_err(|error| HipError(error.to_string()))?;
            Ok(Arc::new(array))
        };
        Ok(vec![
            Arc::new(Int8Array::new(ScalarBuffer::new(wrap(segments.model_type_id.cast(), n), 0, n), None)) as ArrayRef,
            Arc::new(TimestampArray::new(ScalarBuffer::new(wrap(segments.start_time.cast(), 8 * n), 0, n), None)) as ArrayRef,
            Arc::new(TimestampArray::new(ScalarBuffer::new(wrap(segments.end_time.cast(), 8 * n), 0, n), None)) as ArrayRef,
            binary_view(&segments.timestamps)?,
            Arc::new(ValueArray::new(ScalarBuffer::new(wrap(segments.min_value.cast(), 4 * n), 0, n), None)) as ArrayRef,
            Arc::new(ValueArray::new(ScalarBuffer::new(wrap(segments.max_value.cast(), 4 * n), 0, n), None)) as ArrayRef,
            binary_view(&segments.values)?,
            binary_view(&segments.residuals)?,
            Arc::new(Float32Array::new(ScalarBuffer::new(wrap(owned.error.cast(), 4 * n), 0, n), None)) as ArrayRef,
        ])
    }
}

/// Rows `[first, first + length)` of `whole` (see [`OwnedSegments::whole_columns`]) as a batch with
/// `compressed_schema` (`CompressedSegmentBatchBuilder::finish`, types.rs:492-516): slices of the arrays over
/// the whole result, so cutting one result into a batch per chunk copies and checks nothing.
fn record_batch_of_rows(
    whole: &[ArrayRef],
    first: usize,
    length: usize,
    compressed_schema: Arc<Schema>,
    tag_values: &[String],
    field_column_index: i16,
) -> Result<RecordBatch> {
    assert!(whole.iter().all(|column| first + length <= column.len()));
    let mut columns: Vec<ArrayRef> = Vec::with_capacity(compressed_schema.fields().len());
    columns.extend(whole.iter().map(|column| column.slice(first, length)));
    columns.push(Arc::new(iter::repeat_n(field_column_index, length).collect::<Int16Array>()));
    for tag_value in tag_values {
        columns.push(Arc::new(iter::repeat_n(Some(tag_value), length).collect::<StringViewArray>()));
    }
    RecordBatch::try_new(compressed_schema, columns).map_err(|error| HipError(error.to_string()))
}

/// A process-wide context for call sites that have no natural owner for one (the accumulators are
/// created per partition by a closure, `try_compress_univariate_time_series` is a free function).
/// Device from `MODELARDB_HIP_DEVICE` (default 0). Calls through it are serialised; a `GridStream`
/// takes a context of its own from [`pooled_context`] instead.
pub fn shared_context() -> &'static Context {
    static SHARED: std::sync::LazyLock<Context> = std::sync::LazyLock::new(|| {
        let device = default_device();
        Context::new(device).unwrap_or_else(|error| panic!("libmdb_hip cannot use device {device}: {error}"))
    });
    &SHARED
}

/// A context of its own for an operator that lives as long as a query (`GridStream`), taken from a pool the
/// process keeps: a fresh context costs a stream and, with its first batches, device scratch (≈ 15 ms added to
/// the first query), a pooled one nothing. Goes back to the pool when dropped.
pub fn pooled_context() -> Result<PooledContext> {
    let recycled = CONTEXT_POOL.lock().unwrap_or_else(|poisoned| poisoned.into_inner()).pop();
    let context = match recycled {
        Some(context) => context,
        None => Context::new(default_device())?,
    };
    Ok(PooledContext(Some(context)))
}

static CONTEXT_POOL: std::sync::Mutex<Vec<Context>> = std::sync::Mutex::new(Vec::new());

/// At most this many idle contexts are kept; each keeps what `mdb_set_scratch_limit` allows it.
const CONTEXT_POOL_CAPACITY: usize = 64;
/// Device scratch an idle pooled context may keep (256 MiB: the scratch of an 8 192-segment batch is ≈ 100 MB).
const POOLED_SCRATCH_LIMIT: u64 = 256 << 20;

pub struct PooledContext(Option<Context>);

impl std::ops::Deref for PooledContext {
    type Target = Context;
    fn deref(&self) -> &Context {
        self.0.as_ref().expect("the context is only taken out when the guard is dropped")
    }
}

impl Drop for PooledContext {
    fn drop(&mut self) {
        if let Some(context) = self.0.take() {
            let _ = context.set_scratch_limit(POOLED_SCRATCH_LIMIT);
            let mut pool = CONTEXT_POOL.lock().unwrap_or_else(|poisoned| poisoned.into_inner());
            if pool.len() < CONTEXT_POOL_CAPACITY {
                pool.push(context);
            }
        }
    }
}

/// The HIP device this process computes on: `MODELARDB_HIP_DEVICE`, default 0 (one process per GPU).
pub fn default_device() -> i32 {
    std::env::var("MODELARDB_HIP_DEVICE").ok().and_then(|text| text.parse().ok()).unwrap_or(0)
}
