//! Safe wrappers over `libmdb_hip.so` for the three call sites of ModelarDB-RS that this library
//! replaces (see `rust/patches/`): `GridStream::grid_and_append_to_leftovers_in_current_batch`
//! (crates/modelardb_storage/src/query/grid_exec.rs:261-391), the `Model*Accumulator::update_batch`
//! methods (crates/modelardb_storage/src/optimizer/model_simple_aggregates.rs:345-587) and
//! `try_compress_univariate_time_series` (crates/modelardb_compression/src/compression.rs:191-275).
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
    Array, ArrayRef, BinaryViewArray, BinaryViewBuilder, Float32Array, Int8Array, Int16Array,
    StringViewArray,
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
        let owned = OwnedSegments(NonNull::new(raw).expect("success with a null result"));
        Ok(owned.to_record_batch(compressed_schema, tag_values, field_column_index))
    }
}

/// Segments returned by the compressor (host memory), freed on drop.
struct OwnedSegments(NonNull<sys::mdb_segments_owned>);

impl Drop for OwnedSegments {
    fn drop(&mut self) {
        unsafe { sys::mdb_segments_free(self.0.as_ptr()) };
    }
}

impl OwnedSegments {
    fn to_record_batch(&self, compressed_schema: Arc<Schema>, tag_values: &[String], field_column_index: i16) -> RecordBatch {
        let owned = unsafe { self.0.as_ref() };
        let segments = &owned.seg;
        let n = segments.n as usize;
        let primitive = |pointer: *const u8, bytes: usize| -> Buffer {
            // Copied: the batch outlives the library's block (9 small columns per call).
            Buffer::from(unsafe { std::slice::from_raw_parts(pointer, bytes) })
        };
        let binary_view = |column: &sys::mdb_binview_col| -> BinaryViewArray {
            let mut builder = BinaryViewBuilder::with_capacity(n);
            for row in 0..n {
                let view = unsafe { &*column.views.add(row) };
                let length = view.length as usize;
                let bytes = if length <= 12 {
                    &view.u[..length]
                } else {
                    let buffer_index = i32::from_le_bytes(view.u[4..8].try_into().unwrap()) as usize;
                    let offset = i32::from_le_bytes(view.u[8..12].try_into().unwrap()) as usize;
                    unsafe { std::slice::from_raw_parts((*column.buffers.add(buffer_index)).add(offset), length) }
                };
                builder.append_value(bytes);
            }
            builder.finish()
        };
        let mut columns: Vec<ArrayRef> = Vec::with_capacity(compressed_schema.fields().len());
        columns.push(Arc::new(Int8Array::new(ScalarBuffer::new(primitive(segments.model_type_id.cast(), n), 0, n), None)));
        columns.push(Arc::new(TimestampArray::new(ScalarBuffer::new(primitive(segments.start_time.cast(), 8 * n), 0, n), None)));
        columns.push(Arc::new(TimestampArray::new(ScalarBuffer::new(primitive(segments.end_time.cast(), 8 * n), 0, n), None)));
        columns.push(Arc::new(binary_view(&segments.timestamps)));
        columns.push(Arc::new(ValueArray::new(ScalarBuffer::new(primitive(segments.min_value.cast(), 4 * n), 0, n), None)));
        columns.push(Arc::new(ValueArray::new(ScalarBuffer::new(primitive(segments.max_value.cast(), 4 * n), 0, n), None)));
        columns.push(Arc::new(binary_view(&segments.values)));
        columns.push(Arc::new(binary_view(&segments.residuals)));
        columns.push(Arc::new(Float32Array::new(ScalarBuffer::new(primitive(owned.error.cast(), 4 * n), 0, n), None)));
        columns.push(Arc::new(iter::repeat_n(field_column_index, n).collect::<Int16Array>()));
        for tag_value in tag_values {
            columns.push(Arc::new(iter::repeat_n(Some(tag_value), n).collect::<StringViewArray>()));
        }
        RecordBatch::try_new(compressed_schema, columns).expect("the columns follow COMPRESSED_SCHEMA")
    }
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
