// mdb_agg.hip - COUNT / MIN / MAX / SUM / AVG computed directly on segments.
//
// Replaces Model{Count,Min,Max,Sum,Avg}Accumulator::update_batch
// (crates/modelardb_storage/src/optimizer/model_simple_aggregates.rs:345-358, 395-401, 438-444,
// 481-513, 553-587) and the functions they call, modelardb_compression::{len,sum}
// (crates/modelardb_compression/src/models/mod.rs:98-184, pmc_mean.rs:98-100, swing.rs:264-300,
// macaque_v.rs:220-265). The time-range variant is the SURVEY 8(f) N1 extension: it produces what
// the reference computes with GridExec + filter + AggregateExec, without materialising points.
//
// k_agg_segments: 1 thread / segment -> per-segment {f32 sum widened to f64, count, min, max},
// reduced with a fixed tree (wave shuffles, LDS across waves) to one partial per workgroup;
// k_agg_finish reduces the partials with one workgroup. The fixed tree makes SUM run-to-run
// deterministic; it differs from the reference's sequential f64 accumulation only in rounding order.
// Algorithmic bytes: COUNT 32 B/segment, MIN 4, MAX 4, SUM/AVG 73 B/segment + payloads.
#include "mdb_segment_dev.hpp"

#include <cfloat>
#include <thread>

namespace mdb {

constexpr int AGG_THREADS = 256;

struct AggPartial {
    double sum;
    long long count;
    float min;
    float max;
    unsigned int error;
    unsigned int deferred;                // MacaqueV streams left to the parallel decoder ...
    unsigned long long deferred_values;   // ... the values in them ...
    unsigned long long deferred_bytes;    // ... and their bytes
};

__device__ __forceinline__ double shfl_down_f64(double v, int delta) {
    unsigned long long bits = __double_as_longlong(v);
    uint32_t lo = __shfl_down((uint32_t)bits, delta, MDB_WAVE);
    uint32_t hi = __shfl_down((uint32_t)(bits >> 32), delta, MDB_WAVE);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ long long shfl_down_i64(long long v, int delta) {
    uint32_t lo = __shfl_down((uint32_t)v, delta, MDB_WAVE);
    uint32_t hi = __shfl_down((uint32_t)((unsigned long long)v >> 32), delta, MDB_WAVE);
    return (long long)(((unsigned long long)hi << 32) | lo);
}

// Fixed-order reduction of one value per thread to thread 0 of the block.
__device__ __forceinline__ void block_reduce(AggPartial &p, AggPartial *lds) {
#pragma unroll
    for (int delta = MDB_WAVE / 2; delta > 0; delta >>= 1) {
        p.sum += shfl_down_f64(p.sum, delta);
        p.count += shfl_down_i64(p.count, delta);
        p.min = min_num(p.min, __shfl_down(p.min, delta, MDB_WAVE));
        p.max = max_num(p.max, __shfl_down(p.max, delta, MDB_WAVE));
        p.error |= __shfl_down(p.error, delta, MDB_WAVE);
        p.deferred += __shfl_down(p.deferred, delta, MDB_WAVE);
        p.deferred_values += (unsigned long long)shfl_down_i64((long long)p.deferred_values, delta);
        p.deferred_bytes += (unsigned long long)shfl_down_i64((long long)p.deferred_bytes, delta);
    }
    const int lane = threadIdx.x & (MDB_WAVE - 1);
    const int wave = threadIdx.x / MDB_WAVE;
    if (lane == 0) lds[wave] = p;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int n_waves = blockDim.x / MDB_WAVE;
        for (int w = 1; w < n_waves; w++) {
            p.sum += lds[w].sum;
            p.count += lds[w].count;
            p.min = min_num(p.min, lds[w].min);
            p.max = max_num(p.max, lds[w].max);
            p.error |= lds[w].error;
            p.deferred += lds[w].deferred;
            p.deferred_values += lds[w].deferred_values;
            p.deferred_bytes += lds[w].deferred_bytes;
        }
    }
}

__device__ __forceinline__ AggPartial empty_partial() {
    AggPartial p;
    p.sum = 0.0;
    p.count = 0;
    p.min = FLT_MAX;   // f32::MAX (model_simple_aggregates.rs:413)
    p.max = -FLT_MAX;  // f32::MIN (model_simple_aggregates.rs:456)
    p.error = 0;
    p.deferred = 0;
    p.deferred_values = 0;
    p.deferred_bytes = 0;
    return p;
}

// models/mod.rs:129-184: the f32 sum of one segment.
// `walked_sums` (may be nullptr): what the walk of the irregular timestamp streams has added up for the Swing
// segments it adds up (ts_walk_adds): the same terms in the same order as below.
// `stream_sums` (may be nullptr): the sums of the batch's MacaqueV streams, added up by k_agg_mv_chains from the
// values the batch's cursor index let k_agg_mv_pieces decode piece by piece: 2 i the model's, 2 i + 1 the tail's.
__device__ __forceinline__ float segment_sum(const DevSegments &s, uint64_t i, const SegInfo &info,
                                             uint32_t length, uint32_t *error, const double *walked_sums = nullptr,
                                             const float *stream_sums = nullptr,
                                             const unsigned long long *only_with_pieces = nullptr) {
    // (only_with_pieces: the index behind stream_sums is one call's, of the long streams: piece_base of the segments)
    if (stream_sums && only_with_pieces && only_with_pieces[i + 1] == only_with_pieces[i]) stream_sums = nullptr;
    const SegDesc &d = info.desc;
    const uint32_t type = d.flags & FLAG_TYPE_MASK;
    const uint32_t n_res = d.n_total - d.n_model;
    if (length < n_res) {
        *error |= ERR_RESIDUALS;
        return 0.0f;
    }
    const uint32_t model_length = length - n_res;
    float model_last_value = 0.0f;
    float model_sum = 0.0f;
    if (type == MDB_PMC_MEAN_ID) {
        model_last_value = d.value;
        model_sum = (float)model_length * d.value; // pmc_mean.rs:98-100
    } else if (type == MDB_SWING_ID) {
        model_last_value = info.swing_last;
        // swing.rs:264-300, called with the SEGMENT's end_time.
        const int64_t end = s.end_time[i];
        LineDev line = line_through(d.start, (double)info.swing_first, end, (double)info.swing_last);
        if (d.flags & FLAG_REGULAR) {
            double first = line.slope * (double)d.start + line.intercept;
            double last = line.slope * (double)end + line.intercept;
            double average = (first + last) / 2.0;
            model_sum = (float)(average * (double)model_length);
        } else if (walked_sums && ts_walk_adds(s, i)) {
            model_sum = (float)walked_sums[i];
        } else {
            const uint4 vt = s.timestamps.views[i];
            double sum = 0.0;
            decode_irregular_timestamps(view_data(s.timestamps, i, vt), vt.x, d.start, end,
                                        0xffffffffu, error, [&](uint32_t k, int64_t t) {
                                            if (k < d.n_model)
                                                sum += line.slope * (double)t + line.intercept;
                                        });
            model_sum = (float)sum;
        }
    } else if (stream_sums && model_length == d.n_model) {
        model_sum = stream_sums[2 * i];
    } else {
        model_last_value = __uint_as_float(0x7fc00000u); // f32::NAN (models/mod.rs:167)
        const uint4 vv = s.values.views[i];
        float sum = 0.0f;
        decode_macaque_v(view_data(s.values, i, vv), vv.x, model_length, false, 0, error,
                         [&](uint32_t k, uint32_t bits) {
                             // macaque_v.rs:228-235: the sum starts AS the first value.
                             if (k == 0) sum = __uint_as_float(bits);
                             else sum += __uint_as_float(bits);
                         });
        model_sum = sum;
    }
    if (!(d.flags & FLAG_HAS_RESIDUALS)) return model_sum;
    if (stream_sums) return model_sum + stream_sums[2 * i + 1];
    const uint4 vr = s.residuals.views[i];
    float residuals_sum = 0.0f;
    decode_macaque_v(view_data(s.residuals, i, vr), vr.x - 1, n_res, true,
                     __float_as_uint(model_last_value), error,
                     [&](uint32_t, uint32_t bits) { residuals_sum += __uint_as_float(bits); });
    return model_sum + residuals_sum;
}

// How k_agg_segments treats the long MacaqueV streams (mv_qualifies_for_sum): add them up like any
// other (ALL), leave them aside and count them for macaque_deferred_sum (DEFER), or add up nothing
// but them, and only their sums (ONLY_DEFERRED - the way out when macaque_deferred_sum declines,
// which it only does beyond 2^40 values).
enum : uint32_t { AGG_SUM_ALL = 0, AGG_SUM_DEFER = 1, AGG_SUM_ONLY_DEFERRED = 2 };

// `walked_totals`, `walked_sums`, `walked_error` (all may be nullptr): len() of the segments with irregular
// timestamps, the sums of the Swing segments among them, and the error word of the walk that found them
// (ts_walk_for_aggregates) - without them every such segment's stream is decoded here, by one lane, once to be
// counted and once more to be summed.
__global__ __launch_bounds__(AGG_THREADS) void k_agg_segments(DevSegments s, uint32_t which_mask, uint32_t mode,
                                                              uint32_t mv_min_values,
                                                              AggPartial *__restrict__ partials,
                                                              const uint32_t *__restrict__ walked_totals,
                                                              const double *__restrict__ walked_sums,
                                                              const unsigned int *__restrict__ walked_error,
                                                              const float *__restrict__ stream_sums,
                                                              const unsigned long long *__restrict__ only_with_pieces) {
    __shared__ AggPartial lds[AGG_THREADS / MDB_WAVE];
    AggPartial p = empty_partial();
    if (walked_error && blockIdx.x == 0 && threadIdx.x == 0) p.error |= *walked_error;
    const bool need_len = which_mask & (MDB_AGG_COUNT | MDB_AGG_AVG | MDB_AGG_SUM);
    for (uint64_t i = (uint64_t)blockIdx.x * AGG_THREADS + threadIdx.x; i < s.n;
         i += (uint64_t)gridDim.x * AGG_THREADS) {
        if (mode == AGG_SUM_ONLY_DEFERRED) {
            if (s.model_type_id[i] != MDB_MACAQUE_V_ID) continue;
            SegInfo info = analyse_segment(s, i, walked_totals);
            if (mv_deferred_values(s, i, info, mv_min_values, TimeRange{0, 0, 0}) == 0) continue;
            uint32_t error = 0;
            p.sum += (double)segment_sum(s, i, info, info.desc.n_model, &error);
            p.error |= error;
            continue;
        }
        if (which_mask & MDB_AGG_MIN) p.min = min_num(p.min, s.min_value[i]);
        if (which_mask & MDB_AGG_MAX) p.max = max_num(p.max, s.max_value[i]);
        if (!need_len) continue;
        SegInfo info = analyse_segment(s, i, walked_totals, nullptr, false);
        // Only what len()/sum() themselves would trip over is an error here.
        uint32_t error = info.error & (ERR_TIMESTAMPS | ERR_TOO_LONG);
        // len() (models/mod.rs:98-124): a regular stream reports its stored length.
        const uint32_t length = (info.desc.flags & FLAG_REGULAR) ? info.regular_length
                                                                 : info.desc.n_total;
        if (which_mask & (MDB_AGG_COUNT | MDB_AGG_AVG)) p.count += length;
        if (which_mask & (MDB_AGG_SUM | MDB_AGG_AVG)) {
            error |= info.error;
            if (mode == AGG_SUM_DEFER && !error && mv_deferred_values(s, i, info, mv_min_values, TimeRange{0, 0, 0}) != 0) {
                p.deferred += 1;
                p.deferred_values += length;
                p.deferred_bytes += s.values.views[i].x;
            } else if (!error) {
                p.sum += (double)segment_sum(s, i, info, length, &error, walked_sums, stream_sums, only_with_pieces);
            }
        }
        p.error |= error;
    }
    block_reduce(p, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = p;
}

__global__ __launch_bounds__(AGG_THREADS) void k_agg_finish(const AggPartial *__restrict__ partials,
                                                            uint32_t n_partials,
                                                            AggPartial *__restrict__ result) {
    __shared__ AggPartial lds[AGG_THREADS / MDB_WAVE];
    AggPartial p = empty_partial();
    for (uint32_t i = threadIdx.x; i < n_partials; i += AGG_THREADS) {
        const AggPartial q = partials[i];
        p.sum += q.sum;
        p.count += q.count;
        p.min = min_num(p.min, q.min);
        p.max = max_num(p.max, q.max);
        p.error |= q.error;
        p.deferred += q.deferred;
        p.deferred_values += q.deferred_values;
        p.deferred_bytes += q.deferred_bytes;
    }
    block_reduce(p, lds);
    if (threadIdx.x == 0) *result = p;
}

// ---- time-range extension ---------------------------------------------------------------------------

struct RangeAcc {
    double sum = 0.0;
    long long count = 0;
    float min = FLT_MAX;
    float max = -FLT_MAX;
    __device__ __forceinline__ void point(float v) {
        sum += (double)v;
        count += 1;
        min = min_num(min, v);
        max = max_num(max, v);
    }
};

__device__ __forceinline__ float model_value_at(const SegDesc &d, uint32_t type, int64_t t) {
    return type == MDB_PMC_MEAN_ID ? d.value : (float)(d.slope * (double)t + d.intercept);
}

// Aggregate the points of segment i whose timestamp lies in [t_lo, t_hi].
// tail_by_pieces: the residual tail's points are k_agg_mv_range's (regular timestamps only).
__device__ __forceinline__ void segment_range(const DevSegments &s, uint64_t i, const SegInfo &info,
                                              int64_t t_lo, int64_t t_hi, RangeAcc &acc,
                                              uint32_t *error, bool tail_by_pieces = false) {
    const SegDesc &d = info.desc;
    const uint32_t type = d.flags & FLAG_TYPE_MASK;
    const int64_t end = s.end_time[i];
    const uint32_t n_res = d.n_total - d.n_model;
    if (!(d.flags & FLAG_REGULAR)) {
        // Irregular timestamps: one serial pass, every point tested.
        if (end < t_lo || d.start > t_hi) return;
        const uint4 vt = s.timestamps.views[i];
        const uint8_t *ts_bytes = view_data(s.timestamps, i, vt);
        if (type != MDB_MACAQUE_V_ID && n_res == 0) {
            decode_irregular_timestamps(ts_bytes, vt.x, d.start, end, 0xffffffffu, error,
                                        [&](uint32_t, int64_t t) {
                                            if (t >= t_lo && t <= t_hi)
                                                acc.point(model_value_at(d, type, t));
                                        });
            return;
        }
        // Values are a bitstream too (MacaqueV model or residual tail): first find the index
        // interval of the in-range timestamps, then decode the values once. Rare combination.
        uint32_t k_lo = 0xffffffffu, k_hi = 0;
        decode_irregular_timestamps(ts_bytes, vt.x, d.start, end, 0xffffffffu, error,
                                    [&](uint32_t k, int64_t t) {
                                        if (t >= t_lo && t <= t_hi) {
                                            if (k < k_lo) k_lo = k;
                                            if (k > k_hi) k_hi = k;
                                        }
                                    });
        if (k_lo == 0xffffffffu) return;
        // Timestamps are sorted, so the in-range points are exactly the indices k_lo..k_hi.
        float seed = d.value;
        if (type == MDB_MACAQUE_V_ID) {
            const uint4 vv = s.values.views[i];
            uint32_t last_bits = 0;
            decode_macaque_v(view_data(s.values, i, vv), vv.x, d.n_model, false, 0, error,
                             [&](uint32_t k, uint32_t bits) {
                                 if (k >= k_lo && k <= k_hi) acc.point(__uint_as_float(bits));
                                 last_bits = bits;
                             });
            seed = __uint_as_float(last_bits);
        } else {
            decode_irregular_timestamps(ts_bytes, vt.x, d.start, end, d.n_model, error,
                                        [&](uint32_t k, int64_t t) {
                                            if (k >= k_lo && k <= k_hi)
                                                acc.point(model_value_at(d, type, t));
                                        });
        }
        if (n_res > 0) {
            const uint4 vr = s.residuals.views[i];
            decode_macaque_v(view_data(s.residuals, i, vr), vr.x - 1, n_res, true,
                             __float_as_uint(seed), error, [&](uint32_t k, uint32_t bits) {
                                 uint32_t index = d.n_model + k;
                                 if (index >= k_lo && index <= k_hi) acc.point(__uint_as_float(bits));
                             });
        }
        return;
    }

    // Regular timestamps start + k * delta: the in-range indices are an interval [k_lo, k_hi].
    uint32_t k_lo = 0, k_hi = 0;
    if (!regular_index_interval(d.start, d.delta, d.n_total, t_lo, t_hi, &k_lo, &k_hi)) return;

    // Model part [a, b] of the interval.
    if (type != MDB_MACAQUE_V_ID && k_lo < d.n_model) {
        const uint32_t a = k_lo;
        const uint32_t b = min(k_hi, d.n_model - 1);
        const uint32_t n = b - a + 1;
        const int64_t ta = d.start + (int64_t)((uint64_t)a * (uint64_t)d.delta);
        const int64_t tb = d.start + (int64_t)((uint64_t)b * (uint64_t)d.delta);
        const float va = model_value_at(d, type, ta);
        const float vb = model_value_at(d, type, tb);
        // (float)(slope * t + intercept) is monotone in t, so the extremes sit at the ends.
        acc.min = min_num(acc.min, min_num(va, vb));
        acc.max = max_num(acc.max, max_num(va, vb));
        acc.count += n;
        if (type == MDB_PMC_MEAN_ID) {
            acc.sum += (double)d.value * (double)n;
        } else {
            // Sum of the line over n equally spaced points: the f64 closed form of the f32 values
            // grid() would produce - unless it could miss their sum by more than a tenth of the
            // 0.001 % the reference allows (integration_test.rs:1155-1171). That happens when
            // slope * t + intercept cancels almost completely (epoch timestamps, a model that lasts
            // microseconds, values near zero): every reconstructed point then carries rounding noise
            // of ulp(slope * t), which averages out over the points but not over the two end points
            // the closed form uses. The bound below is the worst case of that noise plus the f32
            // rounding of the points; beyond it the points are summed one by one, which is exactly
            // what the reference's plan (GridExec + filter + SUM) computes.
            const double fa = d.slope * (double)ta + d.intercept;
            const double fb = d.slope * (double)tb + d.intercept;
            const double closed = (fa + fb) / 2.0 * (double)n;
            const double magnitude = fmax(fabs(fa), fabs(fb));
            const double cancelled = fmax(fmax(fabs(d.slope * (double)ta), fabs(d.slope * (double)tb)),
                                          fabs(d.intercept));
            const double worst = (double)n * (6.0e-8 * magnitude + 7.0e-46 + 2.3e-16 * cancelled);
            if (worst <= 1.0e-6 * fabs(closed)) {
                acc.sum += closed;
            } else {
                double pointwise = 0.0;
                for (uint32_t k = a; k <= b; k++) {
                    const int64_t t = d.start + (int64_t)((uint64_t)k * (uint64_t)d.delta);
                    pointwise += (double)model_value_at(d, type, t);
                }
                acc.sum += pointwise;
            }
        }
    }
    float seed = d.value;
    if (type == MDB_MACAQUE_V_ID) {
        const uint4 vv = s.values.views[i];
        uint32_t last_bits = 0;
        // Decode only as far as needed unless the residual seed (last value) is needed too.
        const bool residuals_in_range = n_res > 0 && k_hi >= d.n_model;
        const uint32_t upto = residuals_in_range ? d.n_model : min(d.n_model, k_hi + 1);
        if (k_lo < d.n_model || residuals_in_range) {
            decode_macaque_v(view_data(s.values, i, vv), vv.x, upto, false, 0, error,
                             [&](uint32_t k, uint32_t bits) {
                                 if (k >= k_lo && k <= k_hi) acc.point(__uint_as_float(bits));
                                 last_bits = bits;
                             });
        }
        seed = __uint_as_float(last_bits);
    }
    if (n_res > 0 && k_hi >= d.n_model && !tail_by_pieces) {
        const uint4 vr = s.residuals.views[i];
        const uint32_t upto = k_hi - d.n_model + 1;
        decode_macaque_v(view_data(s.residuals, i, vr), vr.x - 1, upto, true, __float_as_uint(seed),
                         error, [&](uint32_t k, uint32_t bits) {
                             uint32_t index = d.n_model + k;
                             if (index >= k_lo) acc.point(__uint_as_float(bits));
                         });
    }
}

// `walked_totals`, `walked_ranges`, `walked_error` (all may be nullptr): of the segments with irregular timestamps
// that reach into the range, len() and - PMC-Mean / Swing without residuals - the aggregates of their points inside
// it, as the walk of their streams has found them (ts_walk_for_aggregates; the same values in the same order as
// segment_range's own pass over such a segment).
__global__ __launch_bounds__(AGG_THREADS) void k_agg_range(DevSegments s, int64_t t_lo, int64_t t_hi, uint32_t mode,
                                                           uint32_t mv_min_values,
                                                           AggPartial *__restrict__ partials,
                                                           const uint32_t *__restrict__ walked_totals,
                                                           const TsWalkRange *__restrict__ walked_ranges,
                                                           const unsigned int *__restrict__ walked_error,
                                                           const unsigned long long *__restrict__ indexed_piece_base,
                                                           const TsWalkRange *__restrict__ whole_in,
                                                           TsWalkRange *__restrict__ whole_out) {
    __shared__ AggPartial lds[AGG_THREADS / MDB_WAVE];
    AggPartial p = empty_partial();
    if (walked_error && blockIdx.x == 0 && threadIdx.x == 0) p.error |= *walked_error;
    const TimeRange range = {t_lo, t_hi, 1};
    for (uint64_t i = (uint64_t)blockIdx.x * AGG_THREADS + threadIdx.x; i < s.n;
         i += (uint64_t)gridDim.x * AGG_THREADS) {
        // Cheap rejection on the two columns the reference prunes on (start_time / end_time).
        const int64_t start_time = s.start_time[i], end_time = s.end_time[i];
        if (end_time < t_lo || start_time > t_hi) continue;
        // `whole_in` (a batch that stays on the device): what this loop made of every point of the segment when it
        // ran over the whole time axis (count < 0: nothing - left to the decoders then, as it is now). A range that
        // contains the segment asks for the same points in the same order.
        if (whole_in && start_time >= t_lo && end_time <= t_hi) {
            const TsWalkRange whole = whole_in[i];
            if (whole.count >= 0) {
                if (mode != AGG_SUM_ONLY_DEFERRED) {
                    p.sum += whole.sum;
                    p.count += whole.count;
                    p.min = min_num(p.min, whole.min);
                    p.max = max_num(p.max, whole.max);
                }
                continue;
            }
        }
        if (mode == AGG_SUM_ONLY_DEFERRED && s.model_type_id[i] != MDB_MACAQUE_V_ID) continue;
        SegInfo info = analyse_segment(s, i, walked_totals);
        uint32_t error = info.error;
        // MacaqueV segments with cursors into their stream: piece by piece (k_agg_mv_range, mdb_grid.hip).
        const bool has_pieces = indexed_piece_base && indexed_piece_base[i + 1] > indexed_piece_base[i];
        if (has_pieces && mv_range_by_pieces(s, i, info)) continue;
        const bool tail_by_pieces = has_pieces && mv_range_tail_by_pieces(s, i, info);
        // Long MacaqueV streams are left to the decoders of mdb_grid.hip (see AGG_SUM_DEFER).
        const uint32_t deferred_values =
            (mode != AGG_SUM_ALL && !error && s.model_type_id[i] == MDB_MACAQUE_V_ID)
                ? mv_deferred_values(s, i, info, mv_min_values, range) : 0u;
        if (mode == AGG_SUM_ONLY_DEFERRED && deferred_values == 0) continue;
        if (mode == AGG_SUM_DEFER && deferred_values != 0) {
            p.deferred += 1;
            p.deferred_values += deferred_values;
            p.deferred_bytes += s.values.views[i].x;
            continue;
        }
        if (!error) {
            RangeAcc acc;
            if (walked_ranges && !(info.desc.flags & FLAG_REGULAR) && ts_walk_aggregates_range(s, i)) {
                const TsWalkRange walked = walked_ranges[i];
                acc.sum = walked.sum;
                acc.count = walked.count;
                acc.min = walked.min;
                acc.max = walked.max;
            } else {
                segment_range(s, i, info, t_lo, t_hi, acc, &error, tail_by_pieces);
            }
            p.sum += acc.sum;
            p.count += acc.count;
            p.min = min_num(p.min, acc.min);
            p.max = max_num(p.max, acc.max);
            if (whole_out && !error) whole_out[i] = TsWalkRange{acc.sum, (long long)acc.count, acc.min, acc.max};
        }
        p.error |= error;
    }
    block_reduce(p, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = p;
}

int agg_run(mdb_ctx *ctx, const mdb_segments *in, bool range, int64_t t_lo, int64_t t_hi,
            uint32_t which_mask, mdb_agg_state *inout) {
    if (in->n == 0) return 0;
    const uint32_t max_blocks = 256 * 8; // grid-stride beyond 8 workgroups per CU
    const uint32_t n_blocks =
        (uint32_t)std::min<uint64_t>((in->n + AGG_THREADS - 1) / AGG_THREADS, max_blocks);
    void *p;
    if (scratch_reserve(ctx, SCRATCH_AGG_PARTIALS, (uint64_t)(n_blocks + 1) * sizeof(AggPartial), &p))
        return 1;
    AggPartial *partials = static_cast<AggPartial *>(p);
    AggPartial *result = partials + n_blocks;
    DevSegments s = to_dev(in);
    const bool sums_wanted = !range && (which_mask & (MDB_AGG_SUM | MDB_AGG_AVG));
    bool mv_forced = false;
    const uint32_t mv_min_values = macaque_parallel_min_values(&mv_forced);
    // Irregular timestamps: len() is the number of codes of the stream and swing::sum a sum over its timestamps;
    // the wave-synchronous walk of the grid path finds both (MDB_AGG_TS_WALK=0: every lane for itself).
    const uint32_t *walked_totals = nullptr;
    const double *walked_sums = nullptr;
    const TsWalkRange *walked_ranges = nullptr;
    const unsigned int *walked_error = nullptr; // (what the walk found wrong with a stream)
    // A batch that stays on the device has (from its first grid or aggregate call on) cursors into its MacaqueV
    // streams: SUM then decodes them piece by piece (mdb_grid.hip, k_agg_mv_pieces) instead of one lane per stream.
    if (sums_wanted && mv_index_ensure(ctx, in)) return 1;
    // Under a time range the same cursors let the points inside it be decoded piece by piece (MDB_AGG_RANGE_PIECES=0:
    // one lane per stream, or the parallel decoder for the long ones).
    std::shared_ptr<MvIndex> range_index;
    const unsigned long long *indexed_piece_base = nullptr;
    const char *range_pieces_setting = option_text("MDB_AGG_RANGE_PIECES");
    if (range && !(range_pieces_setting && std::strcmp(range_pieces_setting, "0") == 0) &&
        mv_index_for_range(ctx, in, &range_index, &indexed_piece_base))
        return 1;
    // What that walk finds is the same for every call without a time range: a batch that stays on the device keeps it
    // (MvIndex::agg_walk_*, as the grid path keeps its cursors; MDB_GRID_TS_CACHE=0: walk every time).
    const bool walk_wanted = range || (which_mask & (MDB_AGG_COUNT | MDB_AGG_AVG | MDB_AGG_SUM));
    const char *cache_setting = option_text("MDB_GRID_TS_CACHE");
    std::shared_ptr<MvIndex> resident;
    const char *walk_setting = option_text("MDB_AGG_TS_WALK"); // (0: no walk, every lane for itself - nothing to keep)
    if (walk_wanted && !range && !(cache_setting && std::strcmp(cache_setting, "0") == 0) &&
        !(walk_setting && std::strcmp(walk_setting, "0") == 0))
        resident = owned_segments_index(in);
    bool walk_kept = false, keep_walk = false;
    if (resident) {
        std::lock_guard<std::mutex> lock(resident->mutex);
        if (resident->agg_walk_built && (!sums_wanted || resident->agg_walk_with_sums)) {
            walk_kept = true;
            walked_totals = static_cast<const uint32_t *>(resident->agg_walk_totals); // (nullptr: no such streams)
            walked_sums = sums_wanted ? static_cast<const double *>(resident->agg_walk_sums) : nullptr;
        }
    }
    // Under a time range a batch that stays on the device keeps what a walk over the WHOLE time axis found: the
    // segments a query's range contains whole take that, only the ones it cuts are walked (MDB_GRID_TS_CACHE=0: all
    // of them that reach into the range, every time).
    bool range_from_kept = false;
    if (range && !(cache_setting && std::strcmp(cache_setting, "0") == 0) && !(walk_setting && std::strcmp(walk_setting, "0") == 0)) {
        if (std::shared_ptr<MvIndex> kept = owned_segments_index(in)) {
            if (ts_range_from_kept(ctx, in, s, TimeRange{t_lo, t_hi, 1}, *kept, &walked_totals, &walked_ranges, &walked_error,
                                   &range_from_kept))
                return 1;
        }
    }
    // ... and what k_agg_range itself makes of every point of a segment (the ones it leaves to the decoders aside):
    // the same for every range that contains the segment (MvIndex::range_acc; made by a pass over the whole time
    // axis when the batch is first asked about a range, again if a switch has moved the line to the decoders).
    const TsWalkRange *whole_acc = nullptr;
    std::shared_ptr<void> whole_acc_held; // (the array stays this call's while its kernels read it)
    std::shared_ptr<void> retired;        // (an array a switch has made stale: let go of after the lock, hipFree waits for the device)
    if (range && !(cache_setting && std::strcmp(cache_setting, "0") == 0)) {
        if (std::shared_ptr<MvIndex> kept = owned_segments_index(in)) {
            // One array per kind of call - with cursors made for this call (host batches) and without - so that calls of
            // both kinds on one batch do not make each other's array again and again; within a kind the key is the line to
            // the decoders (a switch that tests move).
            const int slot = indexed_piece_base ? 1 : 0;
            const uint64_t key = ((uint64_t)mv_min_values << 2) | (indexed_piece_base ? 2u : 0u) | 1u;
            std::lock_guard<std::mutex> lock(kept->mutex);
            // (needs the walk over the whole axis where there are streams to walk: range_from_kept or none at all)
            uint64_t ts_payload = 0;
            for (int32_t b = 0; b < in->timestamps.n_buffers && in->timestamps.buffer_sizes; b++)
                ts_payload += (uint64_t)std::max<int64_t>(in->timestamps.buffer_sizes[b], 0);
            const bool walkable = range_from_kept || ts_payload == 0;
            if (walkable && kept->range_acc_key[slot] != key && !kept->range_acc_failed[slot]) {
                // A new array per key (another call may still be reading the one made under the last key); without the
                // memory for it the query works every segment out itself, now and from now on.
                void *fresh = nullptr;
                if (hipMalloc(&fresh, in->n * sizeof(TsWalkRange)) != hipSuccess) {
                    (void)hipGetLastError();
                    kept->range_acc_failed[slot] = true;
                }
                if (fresh) {
                const int device = ctx->device;
                std::shared_ptr<void> made_array(fresh, [device](void *allocation) {
                    (void)hipSetDevice(device);
                    (void)hipFree(allocation);
                });
                MDB_HIP_CHECK(hipMemsetAsync(fresh, 0xff, in->n * sizeof(TsWalkRange), ctx->stream)); // (count -1)
                {
                    LaunchTimer timer(ctx, "k_agg_range");
                    hipLaunchKernelGGL(k_agg_range, dim3(n_blocks), dim3(AGG_THREADS), 0, ctx->stream, s, INT64_MIN, INT64_MAX,
                                       AGG_SUM_DEFER, mv_min_values, partials, static_cast<const uint32_t *>(kept->range_whole_totals),
                                       static_cast<const TsWalkRange *>(kept->range_whole), static_cast<const unsigned int *>(nullptr),
                                       indexed_piece_base, static_cast<const TsWalkRange *>(nullptr),
                                       static_cast<TsWalkRange *>(fresh));
                    hipLaunchKernelGGL(k_agg_finish, dim3(1), dim3(AGG_THREADS), 0, ctx->stream, partials, n_blocks, result);
                }
                AggPartial made;
                MDB_HIP_CHECK(hipMemcpyAsync(&made, result, sizeof(AggPartial), hipMemcpyDeviceToHost, ctx->stream));
                MDB_HIP_CHECK(hipStreamSynchronize(ctx->stream));
                if (made.error) { // (a fault in the streams: every call finds and reports it itself; not tried again for this kind)
                    kept->range_acc_failed[slot] = true;
                } else {
                    retired = std::move(kept->range_acc[slot]);
                    kept->range_acc[slot] = made_array;
                    kept->range_acc_key[slot] = key;
                }
                }
            }
            if (walkable && kept->range_acc_key[slot] == key && kept->range_acc[slot]) {
                whole_acc_held = kept->range_acc[slot];
                whole_acc = static_cast<const TsWalkRange *>(whole_acc_held.get());
            }
        }
    }
    if (walk_wanted && !walk_kept && !range_from_kept) {
        if (ts_walk_for_aggregates(ctx, in, s, sums_wanted, TimeRange{t_lo, t_hi, range ? 1 : 0}, &walked_totals, &walked_sums,
                                   &walked_ranges, &walked_error))
            return 1;
        if (resident) { // copied now (the scratch is reused), valid once this call has found no fault in the streams
            std::lock_guard<std::mutex> lock(resident->mutex);
            keep_walk = true;
            if (walked_totals) {
                if (!resident->agg_walk_totals) MDB_HIP_CHECK(hipMalloc(&resident->agg_walk_totals, in->n * 4));
                MDB_HIP_CHECK(hipMemcpyAsync(resident->agg_walk_totals, walked_totals, in->n * 4, hipMemcpyDeviceToDevice, ctx->stream));
                if (walked_sums) {
                    if (!resident->agg_walk_sums) MDB_HIP_CHECK(hipMalloc(&resident->agg_walk_sums, in->n * 8));
                    MDB_HIP_CHECK(hipMemcpyAsync(resident->agg_walk_sums, walked_sums, in->n * 8, hipMemcpyDeviceToDevice, ctx->stream));
                }
            }
        }
    }
    if (range) {
        LaunchTimer timer(ctx, "k_agg_range");
        hipLaunchKernelGGL(k_agg_range, dim3(n_blocks), dim3(AGG_THREADS), 0, ctx->stream, s, t_lo,
                           t_hi, AGG_SUM_DEFER, mv_min_values, partials, walked_totals, walked_ranges, walked_error,
                           indexed_piece_base, whole_acc, static_cast<TsWalkRange *>(nullptr));
    } else {
        const float *stream_sums = nullptr;
        const unsigned long long *only_with_pieces = nullptr;
        if (sums_wanted && mv_index_stream_sums(ctx, in, s, walked_totals, &stream_sums, &only_with_pieces)) return 1;
        LaunchTimer timer(ctx, "k_agg_segments");
        hipLaunchKernelGGL(k_agg_segments, dim3(n_blocks), dim3(AGG_THREADS), 0, ctx->stream, s,
                           which_mask, sums_wanted && !stream_sums ? AGG_SUM_DEFER : AGG_SUM_ALL, mv_min_values, partials,
                           walked_totals, walked_sums, walked_error, stream_sums, only_with_pieces);
    }
    {
        LaunchTimer timer(ctx, "k_agg_finish");
        hipLaunchKernelGGL(k_agg_finish, dim3(1), dim3(AGG_THREADS), 0, ctx->stream, partials,
                           n_blocks, result);
    }
    AggPartial host;
    MDB_HIP_CHECK(hipMemcpyAsync(&host, result, sizeof(AggPartial), hipMemcpyDeviceToHost,
                                 ctx->stream));
    MDB_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    MDB_HIP_CHECK(hipGetLastError());
    if (keep_walk && !host.error) {
        std::lock_guard<std::mutex> lock(resident->mutex);
        resident->agg_walk_built = true;
        if (walked_sums || !walked_totals) resident->agg_walk_with_sums = true;
    }
    if (host.error) return fail(describe_error(host.error));
    if (host.deferred > 0) {
        // Long MacaqueV streams were left aside for the decoders of mdb_grid.hip ...
        bool handled = false;
        DeferredTotals totals;
        const TimeRange time_range = {t_lo, t_hi, range ? 1 : 0};
        if (macaque_deferred(ctx, s, time_range, mv_min_values, mv_forced, host.deferred, host.deferred_values,
                             host.deferred_bytes, &handled, &totals, indexed_piece_base))
            return 1;
        if (!handled) { // ... or, if those decline, one lane per stream here after all
            if (range) {
                LaunchTimer timer(ctx, "k_agg_range");
                hipLaunchKernelGGL(k_agg_range, dim3(n_blocks), dim3(AGG_THREADS), 0, ctx->stream, s, t_lo,
                                   t_hi, AGG_SUM_ONLY_DEFERRED, mv_min_values, partials,
                                   static_cast<const uint32_t *>(nullptr), static_cast<const TsWalkRange *>(nullptr),
                                   static_cast<const unsigned int *>(nullptr), indexed_piece_base, whole_acc,
                                   static_cast<TsWalkRange *>(nullptr));
            } else {
                LaunchTimer timer(ctx, "k_agg_segments");
                hipLaunchKernelGGL(k_agg_segments, dim3(n_blocks), dim3(AGG_THREADS), 0, ctx->stream, s,
                                   which_mask, AGG_SUM_ONLY_DEFERRED, mv_min_values, partials,
                                   static_cast<const uint32_t *>(nullptr), static_cast<const double *>(nullptr),
                                   static_cast<const unsigned int *>(nullptr), static_cast<const float *>(nullptr),
                                   static_cast<const unsigned long long *>(nullptr));
            }
            {
                LaunchTimer timer(ctx, "k_agg_finish");
                hipLaunchKernelGGL(k_agg_finish, dim3(1), dim3(AGG_THREADS), 0, ctx->stream, partials,
                                   n_blocks, result);
            }
            AggPartial late;
            MDB_HIP_CHECK(hipMemcpyAsync(&late, result, sizeof(AggPartial), hipMemcpyDeviceToHost,
                                         ctx->stream));
            MDB_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            MDB_HIP_CHECK(hipGetLastError());
            if (late.error) return fail(describe_error(late.error));
            totals.sum = late.sum;
            totals.count = late.count;
            totals.min = late.min;
            totals.max = late.max;
        }
        host.sum += totals.sum;
        if (range) { // without one, COUNT / MIN / MAX of these segments came from their metadata
            host.count += totals.count;
            host.min = totals.min < host.min ? totals.min : host.min;
            host.max = totals.max > host.max ? totals.max : host.max;
        }
    }
    if (indexed_piece_base) {
        DeferredTotals by_pieces;
        if (mv_index_range_totals(ctx, s, TimeRange{t_lo, t_hi, 1}, *range_index, &by_pieces)) return 1;
        host.sum += by_pieces.sum;
        host.count += by_pieces.count;
        host.min = by_pieces.min < host.min ? by_pieces.min : host.min;
        host.max = by_pieces.max > host.max ? by_pieces.max : host.max;
    }
    // Fold into the caller's running state exactly as update_batch would continue it.
    if (which_mask & (MDB_AGG_COUNT | MDB_AGG_AVG)) inout->count += host.count;
    if (which_mask & MDB_AGG_MIN) inout->min = (inout->min != inout->min) ? host.min
                                               : (host.min < inout->min ? host.min : inout->min);
    if (which_mask & MDB_AGG_MAX) inout->max = (inout->max != inout->max) ? host.max
                                               : (host.max > inout->max ? host.max : inout->max);
    if (which_mask & (MDB_AGG_SUM | MDB_AGG_AVG)) inout->sum += host.sum;
    return 0;
}

} // namespace mdb

using namespace mdb;

// The host threads' walk of a call's long MacaqueV streams (mv_host_index) on a thread of its own, so that it runs
// while the calling thread stages the batch and sends it across PCIe: a fold of 262 144 segments of the mixed series is
// 9 ms of walking and 4 ms of upload. Joined before the index is used and on every way out.
namespace {
struct HostWalk {
    std::thread thread;
    template <typename Work> void start(Work work) { thread = std::thread(work); }
    void finish() {
        if (thread.joinable()) thread.join();
    }
    ~HostWalk() { finish(); }
};
} // namespace

extern "C" {

int mdb_agg_batch_dev(mdb_ctx *ctx, const mdb_segments *in, uint32_t which_mask,
                      mdb_agg_state *inout) {
    if (!ctx || !in || !inout) return fail("ctx, in and inout must not be NULL.");
    mdb::CallGuard lock(ctx);
    MDB_HIP_CHECK(hipSetDevice(ctx->device));
    return agg_run(ctx, in, false, 0, 0, which_mask, inout);
}

int mdb_agg_batch_range_dev(mdb_ctx *ctx, const mdb_segments *in, int64_t t_lo, int64_t t_hi,
                            uint32_t which_mask, mdb_agg_state *inout) {
    if (!ctx || !in || !inout) return fail("ctx, in and inout must not be NULL.");
    mdb::CallGuard lock(ctx);
    MDB_HIP_CHECK(hipSetDevice(ctx->device));
    return agg_run(ctx, in, true, t_lo, t_hi, which_mask, inout);
}

int mdb_agg_batch(mdb_ctx *ctx, const mdb_segments *in, uint32_t which_mask, mdb_agg_state *inout) {
    if (!ctx || !in || !inout) return fail("ctx, in and inout must not be NULL.");
    // SUM decodes every value of a MacaqueV stream: the long ones piece by piece from cursors that host threads
    // leave while the batch is on its way (mdb_grid.hip, mv_host_index).
    // (they walk while the batch is staged and crosses PCIe: HostWalk)
    MvCallIndex index;
    HostWalk walk;
    if ((which_mask & (MDB_AGG_SUM | MDB_AGG_AVG)) && mv_host_index_worthwhile(&in, 1))
        walk.start([&index, in] { mv_call_index_build(in, &index); });
    mdb::CallGuard lock(ctx);
    MDB_HIP_CHECK(hipSetDevice(ctx->device));
    mdb_segments_owned *dev = nullptr;
    if (upload_segments_locked(ctx, in, true, &dev)) return 1;
    walk.finish();
    int rc = mv_call_index_use(ctx, dev->seg, index);
    if (!rc) rc = agg_run(ctx, &dev->seg, false, 0, 0, which_mask, inout);
    mv_call_index_done();
    mdb_segments_free(dev);
    return rc;
}

int mdb_agg_batch_list(mdb_ctx *ctx, const mdb_segments *const *inputs, uint32_t n_inputs, uint32_t which_mask,
                       mdb_agg_state *inout) {
    if (!ctx || !inputs || !inout) return fail("ctx, inputs and inout must not be NULL.");
    if (n_inputs == 0) return 0;
    for (uint32_t k = 0; k < n_inputs; k++)
        if (!inputs[k]) return fail("A batch of the list is NULL.");
    MvCallIndex index;
    HostWalk walk;
    if ((which_mask & (MDB_AGG_SUM | MDB_AGG_AVG)) && mv_host_index_worthwhile(inputs, n_inputs))
        walk.start([&index, inputs, n_inputs] { mv_host_index(inputs, n_inputs, &index.piece_base, &index.cursors); });
    mdb::CallGuard lock(ctx);
    MDB_HIP_CHECK(hipSetDevice(ctx->device));
    mdb_segments_owned *dev = nullptr;
    if (upload_segment_list_locked(ctx, inputs, n_inputs, true, &dev)) return 1;
    walk.finish();
    int rc = mv_call_index_use(ctx, dev->seg, index);
    if (!rc) rc = agg_run(ctx, &dev->seg, false, 0, 0, which_mask, inout);
    mv_call_index_done();
    mdb_segments_free(dev);
    return rc;
}

int mdb_agg_batch_range(mdb_ctx *ctx, const mdb_segments *in, int64_t t_lo, int64_t t_hi,
                        uint32_t which_mask, mdb_agg_state *inout) {
    if (!ctx || !in || !inout) return fail("ctx, in and inout must not be NULL.");
    // (cursors into the long MacaqueV streams that reach into the range, by host threads: see mdb_agg_batch)
    MvCallIndex index;
    const MvHostRange host_range{t_lo, t_hi};
    HostWalk walk;
    // (COUNT alone reads no value; MIN / MAX / SUM of a segment the range cuts do)
    if ((which_mask & ~(uint32_t)MDB_AGG_COUNT) != 0 && mv_host_index_worthwhile(&in, 1))
        walk.start([&index, in, &host_range] { mv_call_index_build(in, &index, &host_range); });
    mdb::CallGuard lock(ctx);
    MDB_HIP_CHECK(hipSetDevice(ctx->device));
    mdb_segments_owned *dev = nullptr;
    if (upload_segments_locked(ctx, in, true, &dev)) return 1;
    walk.finish();
    int rc = mv_call_index_use(ctx, dev->seg, index);
    if (!rc) rc = agg_run(ctx, &dev->seg, true, t_lo, t_hi, which_mask, inout);
    mv_call_index_done();
    mdb_segments_free(dev);
    return rc;
}

int mdb_agg_batch_range_list(mdb_ctx *ctx, const mdb_segments *const *inputs, uint32_t n_inputs, int64_t t_lo,
                             int64_t t_hi, uint32_t which_mask, mdb_agg_state *inout) {
    if (!ctx || !inputs || !inout) return fail("ctx, inputs and inout must not be NULL.");
    if (n_inputs == 0) return 0;
    for (uint32_t k = 0; k < n_inputs; k++)
        if (!inputs[k]) return fail("A batch of the list is NULL.");
    // (the list form of mdb_agg_batch_range: cursors into the long MacaqueV streams that reach into the range)
    MvCallIndex index;
    const MvHostRange host_range{t_lo, t_hi};
    HostWalk walk;
    if ((which_mask & ~(uint32_t)MDB_AGG_COUNT) != 0 && mv_host_index_worthwhile(inputs, n_inputs))
        walk.start([&index, inputs, n_inputs, &host_range] {
            mv_host_index(inputs, n_inputs, &index.piece_base, &index.cursors, &host_range);
        });
    mdb::CallGuard lock(ctx);
    MDB_HIP_CHECK(hipSetDevice(ctx->device));
    mdb_segments_owned *dev = nullptr;
    if (upload_segment_list_locked(ctx, inputs, n_inputs, true, &dev)) return 1;
    walk.finish();
    int rc = mv_call_index_use(ctx, dev->seg, index);
    if (!rc) rc = agg_run(ctx, &dev->seg, true, t_lo, t_hi, which_mask, inout);
    mv_call_index_done();
    mdb_segments_free(dev);
    return rc;
}

} // extern "C"
