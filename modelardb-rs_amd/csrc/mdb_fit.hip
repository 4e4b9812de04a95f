// mdb_fit.hip - model-based compression (PMC-Mean / Swing / MacaqueV fitting) on gfx950.
//
// Replaces try_compress_univariate_time_series and everything under it
// (crates/modelardb_compression/src/compression.rs:191-400, types.rs:40-278, models/pmc_mean.rs,
// models/swing.rs, models/macaque_v.rs, models/timestamps.rs:56-155). Citations below are relative to
// crates/modelardb_compression/src/.
//
// The greedy segmentation is strictly sequential inside a series chunk (each accepted model's end
// decides the next start), so the parallel unit is the chunk: C4 has 1.6 M of them.
//   k_fit_models   1 lane / chunk: the greedy loop (compression.rs:224-263) as a per-lane state
//                  machine; every point goes through PMC-Mean and Swing exactly as in the reference.
//                  The wave prefetches each lane's next 24 points into an LDS ring with one
//                  synchronous burst of independent loads, so the sequential loop waits on memory
//                  once per ~24 points instead of once per point. Emits 16-byte model records and
//                  the chunk's segment count.
//   (scan)         chunk -> first segment.
//   k_fit_plan     1 wave / chunk: model records -> per-segment work items (model + residual tail,
//                  or a MacaqueV-only run), the rules of compression.rs:310-362.
//   k_fit_size     1 lane / segment: exact byte length of the timestamps / values / residuals payloads
//                  (encoders run with a counting sink).
//   (scans)        out-of-line payload offsets per BinaryView column.
//   k_fit_encode   1 lane / segment: writes the nine columns in Arrow layout (BinaryView views with
//                  <= 12 bytes inline, larger payloads in one data buffer per column).
// Output is bit-identical to the CPU oracle: same segment boundaries, same bytes.
// Algorithmic bytes: 4 B/point read (12 B/point when timestamps are materialised) + segments written.
#include <type_traits>

#include "mdb_floor_log2.hpp"
#include "mdb_scan.hpp"
#include "mdb_segment_dev.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <unordered_map>
#include <vector>

namespace mdb {

__global__ __launch_bounds__(1024) void k_scan_block_sums(unsigned long long *__restrict__ block_sums,
                                                          uint32_t n_blocks,
                                                          unsigned long long *__restrict__ total_out) {
    __shared__ uint64_t lds[17];
    uint64_t carry = 0;
    for (uint32_t base = 0; base < n_blocks; base += 1024) {
        uint32_t i = base + threadIdx.x;
        uint64_t v = i < n_blocks ? block_sums[i] : 0;
        uint64_t total;
        uint64_t e = block_exclusive_scan_u64(v, lds, &total);
        if (i < n_blocks) block_sums[i] = carry + e;
        carry += total;
    }
    if (threadIdx.x == 0) *total_out = carry;
}

// ---- error bound math (models/mod.rs:53-95) --------------------------------------------------------

__device__ __forceinline__ bool within_error_bound(mdb_error_bound eb, float real_value,
                                                   float approximate_value) {
    if (equal_or_nan((double)real_value, (double)approximate_value)) return true;
    if (eb.kind == MDB_EB_ABSOLUTE) return fabsf(real_value - approximate_value) <= eb.value;
    if (eb.kind == MDB_EB_RELATIVE) {
        float difference = real_value - approximate_value;
        float result = fabsf(difference / real_value);
        return (result * 100.0f) <= eb.value;
    }
    return false;
}

__device__ __forceinline__ double max_allowed_deviation(mdb_error_bound eb, double value) {
    if (eb.kind == MDB_EB_ABSOLUTE) return (double)eb.value * 0.99;
    if (eb.kind == MDB_EB_RELATIVE) return fabs(value * ((double)eb.value / 100.1));
    return 0.0;
}

// The same function with its loop-invariant part evaluated once: `factor` is (double)eps * 0.99
// for an absolute bound and (double)eps / 100.1 for a relative one - the identical IEEE operations
// the per-point form performs, just not repeated per point (the division is ~30 f64 instructions).
struct DeviationFactor {
    int32_t kind;
    double factor;
    __device__ __forceinline__ double of(double value) const {
        if (kind == MDB_EB_ABSOLUTE) return factor;
        if (kind == MDB_EB_RELATIVE) return fabs(value * factor);
        return 0.0;
    }
};

__device__ __forceinline__ DeviationFactor deviation_factor(mdb_error_bound eb) {
    DeviationFactor d;
    d.kind = eb.kind;
    d.factor = eb.kind == MDB_EB_ABSOLUTE ? (double)eb.value * 0.99
                                          : (eb.kind == MDB_EB_RELATIVE ? (double)eb.value / 100.1 : 0.0);
    return d;
}

// ---- timestamps of a chunk: materialised or synthesised regular ----------------------------------------

struct TimestampSource {
    const int64_t *ts; // nullptr: regular
    int64_t regular_start;
    int64_t regular_interval;
    const unsigned long long *series_first_index; // per chunk, may be nullptr
    // Per chunk (first timestamp, interval), or nullptr. Set instead of `ts` when k_fit_regular has
    // found the materialised timestamps of EVERY chunk equally spaced: nothing downstream has to
    // load a timestamp then.
    const long long *chunk_first;
    const long long *chunk_interval;
    // With `ts` set: per chunk, 1 if k_fit_regular found an irregularity in it (nullptr: not known).
    // Segments of the other chunks are regular without looking.
    const unsigned int *chunk_irregular;
};

struct ChunkTimestamps {
    const int64_t *ts;
    int64_t first; // timestamp of point 0 of the chunk when regular
    int64_t interval;
    __device__ __forceinline__ int64_t at(uint32_t j) const {
        return ts ? ts[j] : first + (int64_t)((uint64_t)j * (uint64_t)interval);
    }
    // Regular timestamps only: no pointer test, hence no load the compiler has to wait for.
    __device__ __forceinline__ int64_t regular_at(uint32_t j) const {
        return first + (int64_t)((uint64_t)j * (uint64_t)interval);
    }
};

__device__ __forceinline__ ChunkTimestamps chunk_timestamps(const TimestampSource &src, uint64_t chunk,
                                                            uint64_t chunk_base) {
    ChunkTimestamps t;
    t.ts = src.ts ? src.ts + chunk_base : nullptr;
    if (!src.ts && src.chunk_first) {
        t.first = src.chunk_first[chunk];
        t.interval = src.chunk_interval[chunk];
        return t;
    }
    uint64_t first_index = src.series_first_index ? src.series_first_index[chunk] : 0;
    t.first = src.regular_start + (int64_t)(first_index * (uint64_t)src.regular_interval);
    t.interval = src.regular_interval;
    return t;
}

// One workgroup per chunk: are its timestamps equally spaced (timestamps.rs:56-97 asks that of every
// segment)? Real ingest nearly always hands over timestamp arrays of regular series; knowing that the
// whole call is regular lets every later kernel compute timestamps instead of loading them.
__global__ __launch_bounds__(256) void k_fit_regular(const int64_t *__restrict__ ts,
                                                     const unsigned long long *__restrict__ chunk_offsets,
                                                     uint64_t n_chunks, long long *__restrict__ chunk_first,
                                                     long long *__restrict__ chunk_interval,
                                                     unsigned int *__restrict__ chunk_irregular,
                                                     unsigned int *__restrict__ n_irregular) {
    // n_irregular[1] counts the chunks with a timestamp beyond +-2^52: below that every timestamp and every
    // difference of two is exactly an f64, which the straight-line fitter relies on (SwingFast).
    __shared__ int irregular, beyond;
    const uint64_t chunk = blockIdx.x;
    if (chunk >= n_chunks) return;
    if (threadIdx.x == 0) {
        irregular = 0;
        beyond = 0;
    }
    __syncthreads();
    const uint64_t base = chunk_offsets[chunk];
    const uint64_t n = chunk_offsets[chunk + 1] - base;
    const int64_t *__restrict__ t = ts + base;
    const int64_t first = n > 0 ? t[0] : 0;
    const int64_t interval = n > 1 ? t[1] - t[0] : 0;
    const int64_t exact_limit = 1ll << 52;
    bool mine = false, far = false;
    // Two timestamps per lane and load (16 bytes, a kilobyte per wave instruction) from the first 16-byte boundary
    // on; the one in front of a pair comes from the lane below (from memory for a wave's first lane).
    const uint64_t head = n > 0 ? min((uint64_t)((reinterpret_cast<uintptr_t>(t) >> 3) & 1u), n) : 0; // timestamps in front of the boundary
    const uint64_t pairs = (n - head) / 2;
    const longlong2 *__restrict__ t2 = reinterpret_cast<const longlong2 *>(t + head);
    const int lane = threadIdx.x & (MDB_WAVE - 1);
    for (uint64_t first_pair = 0; first_pair < pairs; first_pair += blockDim.x) { // (every lane takes every round)
        const uint64_t pair = first_pair + threadIdx.x;
        const bool have = pair < pairs;
        const longlong2 here = have ? t2[pair] : make_longlong2(0, 0);
        const uint64_t j = head + 2 * pair; // index of here.x
        int64_t before = (int64_t)(((uint64_t)(uint32_t)__shfl_up((int)(uint32_t)((uint64_t)here.y >> 32), 1, MDB_WAVE) << 32) |
                                   (uint32_t)__shfl_up((int)(uint32_t)(uint64_t)here.y, 1, MDB_WAVE));
        if (lane == 0 && have && j >= 1) before = t[j - 1];
        if (have) {
            far = far || here.x > exact_limit || here.x < -exact_limit || here.y > exact_limit || here.y < -exact_limit;
            if (j >= 2) mine = mine || (here.x - before != interval);
            if (j + 1 >= 2) mine = mine || (here.y - here.x != interval);
        }
    }
    // (the timestamp in front of the boundary, and an odd one at the end)
    if (threadIdx.x == 0 && head == 1) far = far || t[0] > exact_limit || t[0] < -exact_limit;
    if (threadIdx.x == 0 && head + 2 * pairs < n) {
        const uint64_t j = n - 1;
        far = far || t[j] > exact_limit || t[j] < -exact_limit;
        if (j >= 2) mine = mine || (t[j] - t[j - 1] != interval);
    }
    if (mine) irregular = 1;
    if (far) beyond = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        chunk_first[chunk] = first;
        chunk_interval[chunk] = interval;
        chunk_irregular[chunk] = irregular ? 1u : 0u;
        if (irregular) atomicAdd(n_irregular, 1u);
        if (beyond) atomicAdd(n_irregular + 1, 1u);
    }
}

// ---- PMC-Mean (models/pmc_mean.rs:31-93) ---------------------------------------------------------------

struct PmcDev {
    float min_value, max_value;
    double sum;
    uint32_t length;
    __device__ __forceinline__ void reset() {
        min_value = __uint_as_float(0x7fc00000u);
        max_value = __uint_as_float(0x7fc00000u);
        sum = 0.0;
        length = 0;
    }
    __device__ __forceinline__ bool fit(mdb_error_bound eb, float value) {
        float next_min = min_num(min_value, value);
        float next_max = max_num(max_value, value);
        double next_sum = sum + (double)value;
        uint32_t next_length = length + 1;
        float average = (float)(next_sum / (double)next_length);
        if (within_error_bound(eb, next_min, average) && within_error_bound(eb, next_max, average)) {
            min_value = next_min;
            max_value = next_max;
            sum = next_sum;
            length = next_length;
            return true;
        }
        return false;
    }
};

// ---- Swing (models/swing.rs:34-259) -----------------------------------------------------------------------

struct SwingDev {
    int64_t start_time, end_time;
    double first_value;
    LineDev upper, lower;
    double numerator, denominator;
    uint32_t length;
    bool all_finite;
    __device__ __forceinline__ void reset() {
        const double nan = __longlong_as_double(0x7ff8000000000000ll);
        all_finite = false;
        start_time = 0;
        end_time = 0;
        first_value = nan;
        upper = {nan, nan};
        lower = {nan, nan};
        numerator = 0.0;
        denominator = 0.0;
        length = 0;
    }
    __device__ __forceinline__ bool fit(const DeviationFactor &dev, int64_t timestamp, float value32) {
        const double value = (double)value32;
        const double deviation = dev.of(value);
        // Line 6 onwards of Algorithm 1 (swing.rs:144-197) is what runs for almost every point, so
        // it is tested first; `all_finite` caches !first_value.is_finite() of swing.rs:113.
        if (length >= 2 && all_finite) {
            if (!isfinite(value)) return false; // equal_or_nan(finite, non-finite) is false
            const double t = (double)timestamp;
            const double upper_approximation = upper.slope * t + upper.intercept;
            const double lower_approximation = lower.slope * t + lower.intercept;
            if (upper_approximation + deviation < value || lower_approximation - deviation > value)
                return false;
            end_time = timestamp;
            if (upper_approximation - deviation > value)
                upper = line_through(start_time, first_value, timestamp, value + deviation);
            if (lower_approximation + deviation < value)
                lower = line_through(start_time, first_value, timestamp, value - deviation);
            if (first_value != value) { // swing.rs:212-228 (both finite here)
                const double dt = (double)(timestamp - start_time);
                numerator += (value - first_value) * dt;
                denominator += dt * dt;
            } else {
                numerator += 0.0;
                denominator += 0.0;
            }
            length += 1;
            return true;
        }
        if (length == 0) {
            start_time = timestamp;
            end_time = timestamp;
            first_value = value;
            all_finite = isfinite(value);
            length = 1;
            return true;
        }
        if (!all_finite || !isfinite(value)) {
            if (!equal_or_nan(first_value, value)) return false;
            end_time = timestamp;
            upper = {value, value};
            lower = {value, value};
            length += 1;
            return true;
        }
        // length == 1 (swing.rs:126-143)
        end_time = timestamp;
        upper = line_through(start_time, first_value, timestamp, value + deviation);
        lower = line_through(start_time, first_value, timestamp, value - deviation);
        length += 1;
        return true;
    }
    __device__ __forceinline__ void model(float *first, float *last) const { // swing.rs:246-259
        double projected = numerator / denominator;
        double slope = max_num(lower.slope, min_num(projected, upper.slope));
        double last_value = slope * (double)(end_time - start_time) + first_value;
        *first = (float)first_value;
        *last = (float)last_value;
    }
};

// ---- the same two fitters, cheaper, for the kernel's main path -------------------------------------------
//
// k_fit_models runs 64 greedy loops in lockstep, each at a different place of its model, so whatever
// ANY lane needs the wave executes: PMC-Mean's acceptance test (a correctly rounded f64 division for
// the average and two f32 divisions for the relative bound) runs at nearly every step although each
// lane's PMC-Mean model is alive for a few percent of its points, and Swing converts 64-bit
// timestamps to f64 several times per point. Both are replaced by forms that decide the same thing
// with the same bits:
//
// PmcDev::fit_fast   The reference accepts the point iff min and max are within the bound of
//                    avg = (f32)(sum / len). An approximate average (f32 reciprocal, error below 16
//                    ulp) decides that with margins that cover its error and every rounding of the
//                    reference's test; only a point that falls between the margins (about one in 10^5
//                    for a 1 % bound), NaNs and magnitudes near the f32 limits take the exact test. The
//                    model's value itself is computed exactly when the model is finished.
// SwingFast          With regular timestamps t(j) = first + j * interval that stay below 2^53 in
//                    magnitude (checked per call by k_fit_exact_double_timestamps) every timestamp and
//                    every difference of two timestamps IS an f64, so (f64)t, (f64)(t1 - t0) of
//                    swing.rs:150, 215, 335, 338 are computed as such: the same numbers, no 64-bit
//                    integer arithmetic or conversions on the way.

struct PmcFast {
    int32_t kind;    // MDB_EB_*; lossless: always the exact test
    float pass_bound; // relative: eps (1 - 2^-17) / 100; absolute: eps (1 - 2^-17)
    float fail_bound; // relative: eps (1 + 2^-17) / 100; absolute: eps (1 + 2^-17)
    bool enabled;
};

__device__ __forceinline__ PmcFast pmc_fast_constants(mdb_error_bound eb) {
    PmcFast f;
    f.kind = eb.kind;
    const double scale = eb.kind == MDB_EB_RELATIVE ? 0.01 : 1.0;
    f.pass_bound = (float)((double)eb.value * (1.0 - 0x1p-17) * scale);
    f.fail_bound = (float)((double)eb.value * (1.0 + 0x1p-17) * scale);
    // (an absolute bound in the subnormal range leaves no room for the margins)
    f.enabled = eb.kind == MDB_EB_RELATIVE || (eb.kind == MDB_EB_ABSOLUTE && eb.value >= 0x1p-60f);
    return f;
}

// 1: certainly within the bound, -1: certainly not, 0: too close to call (or not a case for the
// margins). `approximate_average` is within `average_error` of the f32 the reference compares with.
__device__ __forceinline__ int pmc_fast_within(const PmcFast &f, float real_value, float approximate_average,
                                               float average_error) {
    const float difference = fabsf(real_value - approximate_average);
    float pass_bound = f.pass_bound, fail_bound = f.fail_bound;
    bool usable = true;
    if (f.kind == MDB_EB_RELATIVE) {
        const float magnitude = fabsf(real_value);
        pass_bound *= magnitude;
        fail_bound *= magnitude;
        usable = magnitude >= 0x1p-60f; // (false for NaN) products far from the subnormal range
    }
    if (usable && difference <= pass_bound - average_error) return 1;
    if (usable && difference > fail_bound + average_error) return -1;
    return 0;
}

// PMCMean::fit_value (pmc_mean.rs:58-76) with the decision taken as described above.
__device__ __forceinline__ bool pmc_fit_fast(PmcDev &pmc, const PmcFast &fast, mdb_error_bound eb, float value) {
    const float next_min = min_num(pmc.min_value, value);
    const float next_max = max_num(pmc.max_value, value);
    const double next_sum = pmc.sum + (double)value;
    const uint32_t next_length = pmc.length + 1;
    int verdict = 0;
    if (fast.enabled) {
        const float approximate = (float)next_sum * __builtin_amdgcn_rcpf((float)next_length);
        const float error = fmaxf(fabsf(approximate) * 0x1p-20f, 0x1p-120f);
        const int low = pmc_fast_within(fast, next_min, approximate, error);
        const int high = pmc_fast_within(fast, next_max, approximate, error);
        verdict = (low < 0 || high < 0) ? -1 : ((low > 0 && high > 0) ? 1 : 0);
    }
    bool accepted = verdict > 0;
    if (verdict == 0) {
        const float average = (float)(next_sum / (double)next_length);
        accepted = within_error_bound(eb, next_min, average) && within_error_bound(eb, next_max, average);
    }
    if (accepted) {
        pmc.min_value = next_min;
        pmc.max_value = next_max;
        pmc.sum = next_sum;
        pmc.length = next_length;
    }
    return accepted;
}

__device__ __forceinline__ LineDev line_through_exact(double t0, double v0, double t1, double v1) {
    if (equal_or_nan(v0, v1)) return {0.0, v0};
    const double slope = (v1 - v0) / (t1 - t0);
    const double intercept = v0 - slope * t0;
    return {slope, intercept};
}

struct SwingFast {
    double start_time; // (f64)start_time, exact
    double first_value;
    LineDev upper, lower;
    double numerator, denominator;
    uint32_t length;
    bool all_finite;
    __device__ __forceinline__ void reset() {
        const double nan = __longlong_as_double(0x7ff8000000000000ll);
        all_finite = false;
        start_time = 0.0;
        first_value = nan;
        upper = {nan, nan};
        lower = {nan, nan};
        numerator = 0.0;
        denominator = 0.0;
        length = 0;
    }
    // SwingDev::fit with `t` = (f64)timestamp. The model's end is not kept: points are fed one after
    // the other, so it is the start plus length - 1 intervals.
    __device__ __forceinline__ bool fit(const DeviationFactor &dev, double t, float value32) {
        const double value = (double)value32;
        const double deviation = dev.of(value);
        if (length >= 2 && all_finite) {
            if (!isfinite(value)) return false;
            const double upper_approximation = upper.slope * t + upper.intercept;
            const double lower_approximation = lower.slope * t + lower.intercept;
            if (upper_approximation + deviation < value || lower_approximation - deviation > value)
                return false;
            // One line per step for the two bounds: the corridor narrows from above or from below,
            // both at the same point only when it is about as wide as the deviation.
            const bool lowers_upper = upper_approximation - deviation > value;
            const bool raises_lower = lower_approximation + deviation < value;
            if (lowers_upper || raises_lower) {
                const LineDev line = line_through_exact(start_time, first_value, t,
                                                        lowers_upper ? value + deviation : value - deviation);
                if (lowers_upper) upper = line;
                else lower = line;
                if (lowers_upper && raises_lower)
                    lower = line_through_exact(start_time, first_value, t, value - deviation);
            }
            if (first_value != value) {
                const double dt = t - start_time;
                numerator += (value - first_value) * dt;
                denominator += dt * dt;
            } else {
                numerator += 0.0;
                denominator += 0.0;
            }
            length += 1;
            return true;
        }
        if (length == 0) {
            start_time = t;
            first_value = value;
            all_finite = isfinite(value);
            length = 1;
            return true;
        }
        if (!all_finite || !isfinite(value)) {
            if (!equal_or_nan(first_value, value)) return false;
            upper = {value, value};
            lower = {value, value};
            length += 1;
            return true;
        }
        upper = line_through_exact(start_time, first_value, t, value + deviation);
        lower = line_through_exact(start_time, first_value, t, value - deviation);
        length += 1;
        return true;
    }
    // `interval`: (f64) of the sampling interval; end_time - start_time is length - 1 of them.
    __device__ __forceinline__ void model(double interval, float *first, float *last) const {
        double projected = numerator / denominator;
        double slope = max_num(lower.slope, min_num(projected, upper.slope));
        double last_value = slope * ((double)(length - 1) * interval) + first_value;
        *first = (float)first_value;
        *last = (float)last_value;
    }
};

// ---- bit sinks (models/bits.rs:86-174, MSB first) -------------------------------------------------------

struct CountSink {
    uint64_t bits = 0;
    __device__ __forceinline__ void put(uint32_t, uint32_t count) { bits += count; }
    __device__ __forceinline__ uint64_t bytes() const { return (bits + 7) >> 3; }
};

// Writes whole 4-byte words once the destination is 4-byte aligned (a payload starts at any byte):
// every lane writes a stream of its own, so each store instruction of a wave touches 64 cache lines
// whatever its width, and a byte at a time is four times as many of them.
struct ByteSink {
    uint8_t *dst;
    uint64_t acc = 0;
    uint32_t pending = 0; // < 32 between calls
    uint64_t written = 0;
    __device__ __forceinline__ explicit ByteSink(uint8_t *d) : dst(d) {}
    // count in [0, 32]
    __device__ __forceinline__ void put(uint32_t value, uint32_t count) {
        if (count == 0) return;
        uint64_t masked = count == 32 ? (uint64_t)value : ((uint64_t)value & ((1ull << count) - 1ull));
        acc = (acc << count) | masked;
        pending += count;
        while (pending >= 32) {
            uint8_t *at = dst + written;
            if ((reinterpret_cast<uintptr_t>(at) & 3u) == 0) {
                *reinterpret_cast<uint32_t *>(at) = __builtin_bswap32((uint32_t)(acc >> (pending - 32)));
                written += 4;
                pending -= 32;
            } else {
                *at = (uint8_t)(acc >> (pending - 8));
                written += 1;
                pending -= 8;
            }
        }
    }
    __device__ __forceinline__ void finish(bool pad_with_ones) {
        while (pending >= 8) {
            dst[written++] = (uint8_t)(acc >> (pending - 8));
            pending -= 8;
        }
        if (pending == 0) return;
        uint32_t pad = 8 - pending;
        uint32_t byte = (uint32_t)(acc << pad) & 0xffu;
        if (pad_with_ones) byte |= (1u << pad) - 1u;
        dst[written++] = (uint8_t)byte;
        pending = 0;
    }
};

template <typename Sink> __device__ __forceinline__ void put64(Sink &sink, uint64_t value, uint32_t count) {
    if (count > 32) {
        sink.put((uint32_t)(value >> 32), count - 32);
        sink.put((uint32_t)value, 32);
    } else {
        sink.put((uint32_t)value, count);
    }
}

// ---- MacaqueV encoder (models/macaque_v.rs:76-214) -----------------------------------------------------

__device__ __forceinline__ int saturating_f32_to_i32(float v) { // Rust `as i32`
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}

struct MacaqueState {
    float min_value = __uint_as_float(0x7fc00000u);
    float max_value = __uint_as_float(0x7fc00000u);
    float last_value = 0.0f;
    uint32_t last_leading = 255;
    uint32_t last_trailing = 0;
};

// (`deviation`: deviation_factor(eb) - max_allowed_deviation's division once per stream, not per value, as in the
// fitters; the division by 2^exponent as a scaling by 2^-exponent: the same real number, rounded the same way)
__device__ __forceinline__ float rewrite_least_mantissa_bits(mdb_error_bound eb, const DeviationFactor &deviation, float value) {
    if (fabsf(value) == 0.0f || value != value || isinf(value)) return value;
    uint32_t bits = __float_as_uint(value);
    float abs_error_bound = (float)deviation.of((double)value);
    int exponent = (int)((bits >> 23) & 0xffu) - 127;
    float factorized_epsilon = ldexpf(abs_error_bound, -exponent);
    // (float)log2((double)x): correctly rounded log2f, the same definition the oracle uses - and a hundred instructions
    // per value of a noisy stream. All that is kept of it is floor(|.|), which mdb_floor_log2.hpp reads off the float's
    // exponent wherever that is certain (all but one value in 16 000; every such value checked on the CPU:
    // tests/test_log2_shortcut_cpu.py).
    float magnitude;
    if (!mdb::floor_abs_log2_from_exponent(__float_as_uint(factorized_epsilon), &magnitude))
        magnitude = floorf(fabsf((float)log2((double)factorized_epsilon)));
    long long wide_position = 23ll - (long long)saturating_f32_to_i32(magnitude);
    int position = wide_position < -2147483647ll ? -2147483647 : (int)wide_position;
    auto rewrite = [](uint32_t b, int pos) -> uint32_t {
        if (pos < 0) pos = 0; // clamp instead of wrapping the shift
        if (pos > 31) return 0u;
        return b & (0xFFFFFFFFu << pos);
    };
    float rewritten = __uint_as_float(rewrite(bits, position));
    if (!within_error_bound(eb, value, rewritten)) {
        position -= 1;
        rewritten = __uint_as_float(rewrite(bits, position));
    }
    return rewritten;
}

template <typename Sink>
__device__ __forceinline__ void macaque_update(MacaqueState &m, float value) {
    m.min_value = min_num(m.min_value, value);
    m.max_value = max_num(m.max_value, value);
    m.last_value = value;
}

template <typename Sink>
__device__ __forceinline__ void macaque_xor_value(MacaqueState &m, Sink &sink, mdb_error_bound eb,
                                                  const DeviationFactor &deviation, float value) {
    if (eb.kind != MDB_EB_LOSSLESS) {
        if (within_error_bound(eb, value, m.last_value)) value = m.last_value;
        else value = rewrite_least_mantissa_bits(eb, deviation, value);
    }
    uint32_t x = __float_as_uint(value) ^ __float_as_uint(m.last_value);
    if (x == 0) {
        sink.put(0b10u, 2);
    } else {
        uint32_t leading = (uint32_t)__clz((int)x);
        uint32_t trailing = (uint32_t)__ffs((int)x) - 1u;
        if (leading >= m.last_leading && trailing >= m.last_trailing) {
            sink.put(0, 1);
            uint32_t meaningful = 32u - m.last_leading - m.last_trailing;
            sink.put(x >> m.last_trailing, meaningful);
        } else {
            uint32_t meaningful = 32u - leading - trailing;
            sink.put((0b11u << 11) | (leading << 6) | meaningful, 13);
            sink.put(x >> trailing, meaningful);
            m.last_leading = leading;
            m.last_trailing = trailing;
        }
    }
    macaque_update<Sink>(m, value);
}

// compress_values (first value raw) or compress_values_without_first (seeded).
template <typename Sink>
__device__ __forceinline__ void macaque_encode(MacaqueState &m, Sink &sink, mdb_error_bound eb,
                                               const float *__restrict__ values, uint32_t n, bool seeded,
                                               float seed) {
    uint32_t i = 0;
    if (seeded) {
        m.last_value = seed;
    } else if (n > 0) {
        sink.put(__float_as_uint(values[0]), 32);
        macaque_update<Sink>(m, values[0]);
        i = 1;
    }
    const DeviationFactor deviation = deviation_factor(eb);
    for (; i < n; i++) macaque_xor_value(m, sink, eb, deviation, values[i]);
}

// ---- MacaqueTS encoder (models/timestamps.rs:56-155) ----------------------------------------------------

__device__ __forceinline__ uint32_t regular_length_bytes(uint64_t length) { // timestamps.rs:99-108
    uint32_t significant = 64u - (uint32_t)__clzll((long long)length);
    return (significant + 1u + 7u) / 8u;
}

__device__ __forceinline__ bool chunk_range_regular(const ChunkTimestamps &t, uint32_t a, uint32_t b) {
    if (!t.ts) return true;
    if (b - a + 1 < 2) return true;
    int64_t expected = t.ts[a + 1] - t.ts[a];
    for (uint32_t j = a + 1; j <= b; j++)
        if (t.ts[j] - t.ts[j - 1] != expected) return false;
    return true;
}

template <typename Sink>
__device__ __forceinline__ void encode_irregular_timestamps(Sink &sink, const ChunkTimestamps &t, uint32_t a,
                                                            uint32_t b) {
    sink.put(1, 1);
    uint64_t last_timestamp = (uint64_t)t.at(a);
    uint64_t last_delta = 0;
    auto encode = [&](uint64_t current) {
        uint64_t delta = current - last_timestamp;
        int64_t dod = (int64_t)(delta - last_delta);
        if (dod == 0) {
            sink.put(0, 1);
        } else if (dod >= -63 && dod <= 64) {
            sink.put((0b10u << 7) | ((uint32_t)dod & 0x7fu), 9);
        } else if (dod >= -255 && dod <= 256) {
            sink.put((0b110u << 9) | ((uint32_t)dod & 0x1ffu), 12);
        } else if (dod >= -2047 && dod <= 2048) {
            sink.put((0b1110u << 12) | ((uint32_t)dod & 0xfffu), 16);
        } else if (dod >= -2147483647ll && dod <= 2147483648ll) {
            sink.put(0b11110u, 5);
            sink.put((uint32_t)dod, 32);
        } else {
            sink.put(0b11111u, 5);
            put64(sink, (uint64_t)dod, 64);
        }
        last_delta = delta;
        last_timestamp = current;
    };
    // Every lane walks a segment of its own, so a load is a trip to memory for the whole wave: AHEAD
    // timestamps are fetched before any of them is encoded.
    constexpr uint32_t AHEAD = 8;
    uint32_t j = a + 1;
    for (; j + AHEAD <= b; j += AHEAD) {
        uint64_t fetched[AHEAD];
#pragma unroll
        for (uint32_t k = 0; k < AHEAD; k++) fetched[k] = (uint64_t)t.at(j + k);
#pragma unroll
        for (uint32_t k = 0; k < AHEAD; k++) encode(fetched[k]);
    }
    for (; j < b; j++) encode((uint64_t)t.at(j));
}

// Length in bytes of compress_residual_timestamps(ts[a..=b]) and whether it is the regular form.
// `known_regular`: 1 / 0 if another kernel has already looked (k_fit_gap), -1 to find out here.
__device__ __forceinline__ uint32_t timestamps_payload_length(const ChunkTimestamps &t, uint32_t a,
                                                              uint32_t b, bool *regular, int known_regular = -1) {
    uint32_t count = b - a + 1;
    *regular = true;
    if (count <= 2) return 0;
    if (known_regular < 0 ? chunk_range_regular(t, a, b) : known_regular != 0) return regular_length_bytes(count);
    *regular = false;
    CountSink sink;
    encode_irregular_timestamps(sink, t, a, b);
    return (uint32_t)sink.bytes();
}

// ---- values column codecs (types.rs:283-370) ------------------------------------------------------------

__device__ __forceinline__ uint32_t encode_values_for_pmc_mean(float mn, float mx, float rmin, float rmax,
                                                               uint8_t *out) {
    if (mn > rmin) {
        if (mx >= rmax) {
            out[0] = 1;
            return 1;
        }
        uint32_t bits = __float_as_uint(mn);
        for (int k = 0; k < 4; k++) out[k] = (uint8_t)(bits >> (8 * k));
        return 4;
    }
    return 0;
}

__device__ __forceinline__ uint32_t encode_values_for_swing(float mn, float mx, bool min_is_first, float rmin,
                                                            float rmax, uint8_t *out) {
    auto put = [&](uint32_t at, float v) {
        uint32_t bits = __float_as_uint(v);
        for (int k = 0; k < 4; k++) out[at + k] = (uint8_t)(bits >> (8 * k));
    };
    if (rmin < mn && mx < rmax) {
        put(0, min_is_first ? mn : mx);
        put(4, min_is_first ? mx : mn);
        return 8;
    }
    if (rmin < mn) {
        out[0] = min_is_first ? 0 : 1;
        put(1, mn);
        return 5;
    }
    if (mx < rmax) {
        out[0] = min_is_first ? 2 : 3;
        put(1, mx);
        return 5;
    }
    if (!min_is_first) {
        out[0] = 0;
        return 1;
    }
    return 0;
}

// ---- k_fit_models -----------------------------------------------------------------------------------------

struct ModelRec { // 16 bytes
    uint32_t start_and_type; // bit 31: 1 = Swing, 0 = PMC-Mean; low 31 bits: first point in the chunk
    uint32_t end;            // last point in the chunk the model represents
    float p0;                // PMC-Mean: value. Swing: first value
    float p1;                // Swing: last value
};

struct ChunkPlan { // per chunk
    uint32_t n_models;
    uint32_t n_segments;
};

struct FitArgs {
    const float *values;
    TimestampSource timestamps;
    const unsigned long long *chunk_offsets;
    uint64_t n_chunks;
    mdb_error_bound eb;
    // Lossless MacaqueV-only segments of at least this many values are encoded by one WAVE each
    // (k_fit_gap) instead of one lane; 0xffffffff: never. gap_results[segment] then holds what
    // process_segment needs to know about them.
    uint32_t gap_min_values;
    const struct GapResult *gap_results;
    // ... and those of at least this many values are cut into blocks with a wave each (k_fit_long*); 0xffffffff: never.
    uint32_t gap_long_min_values = 0xffffffffu;
    // Timestamps of segments inside irregular chunks are sized and written by one WAVE each
    // (k_fit_timestamps); ts_results[segment].bytes == 0xffffffff: not this one. May be nullptr.
    const struct TsResult *ts_results;
    // The call of a handful of chunks (fit_few_chunks) never asks the device how far it is: the number of segments
    // and where the encoders write are read from device memory by the kernels behind k_fit_walk, which are launched
    // over upper bounds. nullptr: the values the kernels are passed.
    const unsigned long long *n_segments_dev = nullptr;
    const struct EncodeTargets *targets_dev = nullptr;
};

__device__ __forceinline__ uint64_t chunk_record_capacity(uint64_t length) { return length / 8 + 1; }

struct RecordCapacity {
    const unsigned long long *chunk_offsets;
    __device__ uint64_t operator()(uint64_t c) const {
        return chunk_record_capacity(chunk_offsets[c + 1] - chunk_offsets[c]);
    }
};

// Segments implied by the gaps between accepted models (compression.rs:240-249, 310-362): a gap of
// more than 255 points, or any gap in front of the first model, is a MacaqueV segment of its own.
struct GapCounter {
    bool have_previous = false;
    uint32_t previous_end = 0;
    uint32_t n_segments = 0;
    __device__ __forceinline__ void on_model(uint32_t start, uint32_t end) {
        if (have_previous) {
            if (start - 1 - previous_end > MDB_RESIDUAL_VALUES_MAX_LENGTH) n_segments += 1;
        } else if (start > 0) {
            n_segments += 1;
        }
        n_segments += 1;
        have_previous = true;
        previous_end = end;
    }
    __device__ __forceinline__ uint32_t finish(uint32_t n) {
        if (have_previous) {
            if (n - 1 - previous_end > MDB_RESIDUAL_VALUES_MAX_LENGTH) n_segments += 1;
        } else {
            n_segments += 1;
        }
        return n_segments;
    }
};

// Split mode (few chunks per call: an embedded-API series, a handful of ingest buffers). The greedy
// loop is a pure function "model fitted from point c -> (accepted, end) or rejected", and the chain
// of models of a chunk is that function iterated from point 0. With one lane per chunk a call with
// few chunks leaves the GPU idle for chunk_length x ~1 us. Instead every chunk is cut into pieces of
// `piece_points` and one lane starts the same greedy loop at the start of each piece, speculatively:
// its first models are in general NOT models of the real chain, but greedy chains that start at
// different points meet after a few models (as soon as two of them end a model at the same point)
// and are identical from there on. Every lane records what it fitted from each start point it
// visits in a per-point table and stops as soon as it reaches a point some lane has already been
// at (normally the lane of the next piece, a few models into that piece): from there the recorded
// chain is the one it would produce itself. k_fit_walk then follows the table from point 0 of every
// chunk and copies the models on the real chain into the per-chunk record lists the non-split path
// produces, so everything downstream is unchanged and the result is bit-identical.
// Table entry per input point: 0 = not visited, 1 = no model accepted from here (the point becomes a
// residual, compression.rs:258-262), otherwise (end + 2) | type << 31.
struct SplitArgs {
    const unsigned long long *piece_base; // exclusive scan of the pieces per chunk, n_chunks + 1
    uint64_t n_pieces;
    uint32_t piece_points;
    unsigned int *entry; // indexed by chunk_offsets[chunk] + point
    float *p0;
    float *p1;
    // Per chunk, or nullptr: 0 = this chunk has its models already (k_fit_models_wave), it gets no pieces and
    // is not walked.
    const unsigned int *chunk_left;
    // Or nullptr: a bit per point, set where k_fit_reject_flags has found that NO model can begin (see there), laid
    // out by piece - piece u of the call owns words [u * reject_words_per_piece, ...), bit k of word w is point
    // 64 w + k of the piece - so that the words of a chunk follow each other.
    const unsigned long long *reject_words;
    uint32_t reject_words_per_piece;
};

constexpr uint32_t ENTRY_REJECTED = 1u;
// (no model accepted from here either, but said by k_fit_reject_flags before any lane came by: NOT a lane's track. A
// model's entry is its end + 2 and a model has eight points or more, so 2 is nobody's end.)
constexpr uint32_t ENTRY_FLAGGED = 2u;
__device__ __forceinline__ bool entry_is_a_track(uint32_t entry) { return entry == ENTRY_REJECTED || entry > ENTRY_FLAGGED; }
__device__ __forceinline__ uint32_t entry_for_the_walk(uint32_t entry) { return entry == ENTRY_FLAGGED ? ENTRY_REJECTED : entry; }
constexpr uint32_t ENTRY_END_BIAS = 2u;

struct PieceCount {
    const unsigned long long *chunk_offsets;
    uint32_t piece_points;
    const unsigned int *chunk_left;
    __device__ uint64_t operator()(uint64_t c) const {
        if (chunk_left && chunk_left[c] == 0u) return 0;
        const uint64_t length = chunk_offsets[c + 1] - chunk_offsets[c];
        return length > COUNT_MASK - ENTRY_END_BIAS ? 0 : (length + piece_points - 1) / piece_points;
    }
};

// One lane per chunk, one wave per workgroup. The greedy loop is flattened into a per-lane state
// machine ("feed one point" or "finish the model") so that all 64 lanes can share wave-synchronous
// prefetching: every lane keeps a ring of its next FIT_RING values (and timestamps, when they are
// materialised) in LDS, laid out [slot][lane] so a slot is one bank-conflict-free row. When ANY lane
// runs dry the WHOLE wave tops its rings up with independent, predicated loads issued back to back
// and waits once - one HBM/L2 latency per ~FIT_RING-8 points instead of one per point. The ring
// always keeps the 8 most recent points, because a rejected model restarts at most 7 points back.
#ifndef MDB_FIT_RING
#define MDB_FIT_RING 32
#endif
constexpr int FIT_RING = MDB_FIT_RING;
#ifndef MDB_FIT_RING_WITH_TIMESTAMPS
#define MDB_FIT_RING_WITH_TIMESTAMPS 16
#endif
constexpr int FIT_RING_WITH_TIMESTAMPS = MDB_FIT_RING_WITH_TIMESTAMPS;
constexpr int FIT_HISTORY = 8;
constexpr int FIT_THREADS = MDB_WAVE;
constexpr int FIT_QUICK_REJECTS = 16; // rejected points skipped per trip (lossless bound only)

// Can every timestamp of the call be held exactly in an f64, differences included (SwingFast)? Regular
// timestamps only: |first| <= 2^52 and |interval| * length <= 2^52 for every chunk.
__global__ __launch_bounds__(256) void k_fit_exact_double_timestamps(TimestampSource timestamps,
                                                                     const unsigned long long *__restrict__ chunk_offsets,
                                                                     uint64_t n_chunks, unsigned int *__restrict__ inexact) {

    const uint64_t chunk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (chunk >= n_chunks) return;
    const uint64_t base = chunk_offsets[chunk];
    const uint64_t n = chunk_offsets[chunk + 1] - base;
    const ChunkTimestamps t = chunk_timestamps(timestamps, chunk, base);
    const uint64_t limit = 1ull << 52;
    const uint64_t first = t.first < 0 ? 0ull - (uint64_t)t.first : (uint64_t)t.first;
    const uint64_t interval = t.interval < 0 ? 0ull - (uint64_t)t.interval : (uint64_t)t.interval;
    const bool fits = first <= limit && n <= limit && (interval == 0 || n <= limit / interval);
    if (!fits) atomicOr(inexact, 1u);
}

// FAST (regular timestamps that are exact in f64 only): PmcDev::fit_fast and SwingFast above.
template <bool HAS_TS, bool SPLIT, bool FAST>
__global__ __launch_bounds__(FIT_THREADS) void k_fit_models(FitArgs args, SplitArgs split,
                                                           const unsigned long long *__restrict__ record_base,
                                                           ModelRec *__restrict__ records,
                                                           ChunkPlan *__restrict__ plans,
                                                           unsigned int *__restrict__ error) {
    // With timestamps a ring slot is 12 bytes per lane instead of 4: fewer slots, so that as many
    // waves fit into a CU's LDS as without (what the kernel lives on is waves in flight).
    constexpr int RING = HAS_TS ? FIT_RING_WITH_TIMESTAMPS : FIT_RING;
    __shared__ float ring_values[RING][MDB_WAVE];
    __shared__ long long ring_ts[HAS_TS ? RING : 1][MDB_WAVE];
    const int lane = threadIdx.x;
    const uint64_t unit = (uint64_t)blockIdx.x * FIT_THREADS + lane;
    uint64_t chunk = unit;
    bool active = unit < (SPLIT ? split.n_pieces : args.n_chunks);
    uint32_t first_point = 0;
    if (SPLIT && active) {
        // The chunk of this piece: the last c with piece_base[c] <= unit.
        uint64_t lo = 0, hi = args.n_chunks;
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) / 2;
            if (split.piece_base[mid] <= unit) lo = mid;
            else hi = mid;
        }
        chunk = lo;
        first_point = (uint32_t)(unit - split.piece_base[chunk]) * split.piece_points;
    }
    uint64_t base = 0;
    uint32_t n = 0;
    if (active) {
        base = args.chunk_offsets[chunk];
        const uint64_t length64 = args.chunk_offsets[chunk + 1] - base;
        if (length64 > COUNT_MASK - ENTRY_END_BIAS) { // (split mode gives such a chunk no pieces)
            atomicOr(error, ERR_TOO_LONG);
            plans[chunk] = {0, 0};
            active = false;
        } else {
            n = (uint32_t)length64;
        }
    }
    const float *__restrict__ values = args.values + base;
    const int64_t *__restrict__ timestamps = HAS_TS ? args.timestamps.ts + base : nullptr;
    const ChunkTimestamps regular_ts = chunk_timestamps(args.timestamps, active ? chunk : 0, base);
    const mdb_error_bound eb = args.eb;
    const DeviationFactor dev = deviation_factor(eb);
    ModelRec *__restrict__ out = records + ((active && !SPLIT) ? record_base[chunk] : 0);

    uint32_t n_models = 0;
    GapCounter gaps;
    const uint32_t piece_end = SPLIT ? first_point + split.piece_points : 0u; // (split mode) of this lane's piece
    uint32_t current = first_point; // first point of the model being fitted
    uint32_t j = first_point;       // next point to feed
    uint32_t loaded = first_point;  // the ring holds points [low, loaded) of the chunk, loaded - low <= RING
    uint32_t low = first_point;
    static_assert(!(FAST && HAS_TS), "the fast path computes its timestamps");
    PmcDev pmc;
    typename std::conditional<FAST, SwingFast, SwingDev>::type swing;
    pmc.reset();
    swing.reset();
    const PmcFast pmc_fast = pmc_fast_constants(eb);
    const double first_time = (double)regular_ts.first, interval_time = (double)regular_ts.interval; // exact if FAST
    bool pmc_fits = true, swing_fits = true;
    if (!SPLIT && active && n == 0) {
        plans[chunk] = {0, 0};
        active = false;
    }

    while (__any(active)) {
        // Lossless bound, noisy data: nearly every point is rejected (PMC-Mean ends at the second
        // point because the values differ, Swing at the third because the points are not exactly
        // collinear, and neither reaches the 8 points a model needs). Feeding three points and
        // finishing costs four trips through this loop per point; the same verdict follows from the
        // first three points directly (pmc_mean.rs:58-76 and swing.rs:101-198 with a zero deviation),
        // so runs of such points are skipped here, a few per trip.
        if (args.eb.kind == MDB_EB_LOSSLESS) {
            for (int skipped = 0; skipped < FIT_QUICK_REJECTS; skipped++) {
                bool reject = false;
                // (only on points that are in the lane's LDS ring already: a load per point would cost
                // more than the trips it saves)
                if (active && j == current && current + 2 < n && current >= low && current + 2 < loaded) {
                    const float v0 = ring_values[current % RING][lane];
                    const float v1 = ring_values[(current + 1) % RING][lane];
                    const float v2 = ring_values[(current + 2) % RING][lane];
                    if (isfinite(v0) && isfinite(v1) && isfinite(v2) && v0 != v1) {
                        const int64_t t0 = HAS_TS ? (int64_t)ring_ts[current % RING][lane] : regular_ts.regular_at(current);
                        const int64_t t1 = HAS_TS ? (int64_t)ring_ts[(current + 1) % RING][lane]
                                                  : regular_ts.regular_at(current + 1);
                        const int64_t t2 = HAS_TS ? (int64_t)ring_ts[(current + 2) % RING][lane]
                                                  : regular_ts.regular_at(current + 2);
                        const LineDev line = line_through(t0, (double)v0, t1, (double)v1);
                        const double approximation = line.slope * (double)t2 + line.intercept;
                        reject = approximation < (double)v2 || approximation > (double)v2;
                    }
                }
                if (!__any(reject)) break;
                if (reject) {
                    if (SPLIT)
                        __hip_atomic_store(&split.entry[base + current], ENTRY_REJECTED, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                    current += 1;
                    j = current;
                    // Inside its own piece a lane is normally the first one there: it only has to look
                    // for another lane's tracks once it is past the end of its piece.
                    if (SPLIT && current >= piece_end &&
                        __hip_atomic_load(&split.entry[base + current], __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_AGENT) != 0u)
                        active = false; // some lane has been here already
                }
            }
        }
        const bool feeding = active && j < n && (pmc_fits || swing_fits);
        // Wave-synchronous top-up: triggered by any lane whose next point is not in its ring.
        if (__any(feeding && (j >= loaded || j < low))) {
            float fetched_values[RING - FIT_HISTORY];
            long long fetched_ts[HAS_TS ? RING - FIT_HISTORY : 1];
            // Normally the ring is extended at `loaded`. A lane whose next point fell out of the
            // back of its ring (PMC-Mean chosen although Swing had reached > 8 points further,
            // types.rs:84-101) restarts its ring at j; points before j are never needed again.
            if (active && (j < low || j > loaded)) { // (j > loaded: after a run of quick rejects)
                loaded = j;
                low = j;
            }
            const uint32_t first = loaded;
            const uint32_t room = active ? (uint32_t)max(0, (int)(j + RING - FIT_HISTORY) - (int)first) : 0u;
#pragma unroll
            for (int k = 0; k < RING - FIT_HISTORY; k++) {
                const uint32_t index = first + k;
                const bool wanted = (uint32_t)k < room && index < n;
                fetched_values[k] = wanted ? values[index] : 0.0f;
                if (HAS_TS) fetched_ts[k] = wanted ? timestamps[index] : 0;
            }
#pragma unroll
            for (int k = 0; k < RING - FIT_HISTORY; k++) {
                const uint32_t index = first + k;
                if ((uint32_t)k < room && index < n) {
                    ring_values[index % RING][lane] = fetched_values[k];
                    if (HAS_TS) ring_ts[index % RING][lane] = fetched_ts[k];
                }
            }
            loaded = min(n, first + min(room, (uint32_t)(RING - FIT_HISTORY)));
            if (loaded > low + RING) low = loaded - RING;
            // Only this lane reads its own column: no cross-lane hazard, just LDS program order.
        }
        if (feeding) {
            const float v = ring_values[j % RING][lane];
            // try_to_update_models (types.rs:74-81): a model that failed once is never fed again.
            if constexpr (FAST) {
                // first + j * interval: the product and the sum are integers below 2^53, hence exact
                // whether fused or not.
                const double t = __builtin_fma((double)j, interval_time, first_time);
                if (pmc_fits) pmc_fits = pmc_fit_fast(pmc, pmc_fast, eb, v);
                if (swing_fits) swing_fits = swing.fit(dev, t, v);
            } else {
                const int64_t t = HAS_TS ? (int64_t)ring_ts[j % RING][lane] : regular_ts.regular_at(j);
                if (pmc_fits) pmc_fits = pmc.fit(eb, v);
                if (swing_fits) swing_fits = swing.fit(dev, t, v);
            }
            j += 1;
        } else if (active) {
            // ModelBuilder::finish (types.rs:84-101): fewest bytes per value, PMC-Mean wins ties.
            const float pmc_bpv = (float)MDB_COMPRESSED_METADATA_SIZE_IN_BYTES / (float)pmc.length;
            const float swing_bpv = ((float)MDB_COMPRESSED_METADATA_SIZE_IN_BYTES + 1.0f) / (float)swing.length;
            const bool choose_pmc = pmc_bpv <= swing_bpv;
            const float bpv = choose_pmc ? pmc_bpv : swing_bpv;
            if (bpv <= (float)MDB_VALUE_SIZE_IN_BYTES) { // compression.rs:238
                ModelRec rec;
                if (choose_pmc) {
                    rec.start_and_type = current;
                    rec.end = current + pmc.length - 1;
                    rec.p0 = (float)(pmc.sum / (double)pmc.length); // pmc_mean.rs:91-93
                    rec.p1 = rec.p0;
                } else {
                    rec.start_and_type = current | 0x80000000u;
                    rec.end = current + swing.length - 1;
                    if constexpr (FAST) swing.model(interval_time, &rec.p0, &rec.p1);
                    else swing.model(&rec.p0, &rec.p1);
                }
                if (SPLIT) {
                    split.p0[base + current] = rec.p0;
                    split.p1[base + current] = rec.p1;
                    __hip_atomic_store(&split.entry[base + current],
                                       (rec.end + ENTRY_END_BIAS) | (rec.start_and_type & 0x80000000u),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    gaps.on_model(current, rec.end);
                    out[n_models++] = rec;
                }
                current = rec.end + 1;
            } else {
                if (SPLIT)
                    __hip_atomic_store(&split.entry[base + current], ENTRY_REJECTED, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                current += 1; // the point becomes a residual (compression.rs:258-262)
            }
            if (current >= n) {
                if (!SPLIT) plans[chunk] = {n_models, gaps.finish(n)};
                active = false;
            } else if (SPLIT && current >= piece_end &&
                       __hip_atomic_load(&split.entry[base + current], __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                active = false; // some lane has been here: the chain from this point on is recorded
            } else {
                pmc.reset();
                swing.reset();
                pmc_fits = true;
                swing_fits = true;
                j = current;
            }
        }
    }
}

// ---- k_fit_models_lean --------------------------------------------------------------------------------------
//
// The same greedy loop for the case that matters most - regular timestamps that are exact in f64
// (k_fit_exact_double_timestamps), any kind of error bound - written for how a wave
// executes it. Counters of k_fit_models on the benchmark workload (scripts/pmc_fit_modes.sh): per
// point 161 vector, 131 SCALAR and 28 branch instructions - every `if` on a per-lane condition costs
// the wave a handful of scalar instructions to split and rejoin its lanes whether or not a lane
// takes it, and with 64 models at 64 different places nearly every `if` of the two fitters is taken
// by somebody. Here one step of PMC-Mean and one step of Swing are straight-line code: every lane
// computes the step, conditions select what is kept. What stays behind a branch is what is rare for
// the whole wave: the exact PMC-Mean test (pmc_fit_fast), the second line of a Swing step that
// moves both bounds, non-finite values, finishing a model, and topping up the ring.
// The ring holds 16-byte groups (one global_load_dwordx4 per 4 points instead of 4 loads; a chunk
// may start at any float, so groups are counted from the 16-byte boundary below its first value).
// Same records, same bytes as k_fit_models: every fit test runs both (MDB_FIT_LEAN=0 selects the
// other).
#ifndef MDB_LEAN_GROUPS
#define MDB_LEAN_GROUPS 8
#endif
#ifndef MDB_LEAN_LOADS
#define MDB_LEAN_LOADS 6
#endif
constexpr int LEAN_GROUPS = MDB_LEAN_GROUPS; // 16-byte groups per lane in the ring (32 points)
constexpr int LEAN_LOADS = MDB_LEAN_LOADS;   // groups fetched per top-up at most; 2 groups of history stay
static_assert(LEAN_LOADS + 2 <= LEAN_GROUPS, "a top-up must leave two groups of history in the ring");

// One bit per lane of the wave, the same value in every lane (so: scalar registers).
using LaneMask = unsigned long long;
__device__ __forceinline__ LaneMask lanes_where(bool condition) { return __builtin_amdgcn_ballot_w64(condition); }
__device__ __forceinline__ bool in_lanes(LaneMask mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }
// First statement of an `if (in_lanes(m)) { a = b; ... }` whose assignments should run under the lanes' mask
// (one move per register, one per DOUBLE register pair) rather than be turned into a select per 32 bits.
__device__ __forceinline__ void keep_under_mask() { asm volatile(""); }

template <int KIND> __device__ __forceinline__ double lean_deviation(double factor, double value) {
    if (KIND == MDB_EB_LOSSLESS) return 0.0;
    return KIND == MDB_EB_RELATIVE ? fabs(value * factor) : factor; // DeviationFactor::of
}

// The lanes whose value is certainly within the bound (the first half of pmc_fast_within, without branches).
// (lanes_where of ONE comparison is the comparison's own result; of an expression it is a detour through a
// vector register.)
template <int KIND>
__device__ __forceinline__ LaneMask lean_passes(const PmcFast &f, float real_value, float approximate_average,
                                                float average_error) {
    const float difference = fabsf(real_value - approximate_average);
    float pass_bound = f.pass_bound;
    LaneMask usable = ~0ull;
    if (KIND == MDB_EB_RELATIVE) {
        const float magnitude = fabsf(real_value);
        pass_bound *= magnitude;
        usable = lanes_where(magnitude >= 0x1p-60f);
    }
    return usable & lanes_where(difference <= pass_bound - average_error);
}

// MDB_FIT_TIMING=N (a build of its own, scripts/r04/build_timing.sh; never defined in the product): shader clock
// cycles (s_memtime) a wave of k_fit_models_lean<false, RELATIVE, false> spends in region N of a step, summed over
// all waves into g_fit_timing[0] (cycles), [1] (passes through the region), [2] (whole loop, cycles), [3] (steps),
// [4] (s_memrealtime ticks of the whole loop, 100 MHz). Regions: 0 nothing (two clock reads back to back), 1 the
// ring's top-up, 2 the LDS read of the step's value, 3 PMC-Mean, 4 Swing, 5 finishing models.
#ifdef MDB_FIT_TIMING
__device__ unsigned long long g_fit_timing[8];
__device__ unsigned long long g_fit_waves[4 * 65536]; // MDB_FIT_TIMING=6: per wave {loop ticks, HW_ID, XCC_ID, start tick}
#define FIT_TIMING_BEGIN(N) unsigned long long timing_t0_##N = 0; if (TIMED && MDB_FIT_TIMING == N) timing_t0_##N = __builtin_amdgcn_s_memtime()
#define FIT_TIMING_END(N) if (TIMED && MDB_FIT_TIMING == N) { timing_cycles += __builtin_amdgcn_s_memtime() - timing_t0_##N; timing_passes += 1; }
#else
#define FIT_TIMING_BEGIN(N)
#define FIT_TIMING_END(N)
#endif

template <int GROUPS>
__device__ __forceinline__ float ring_value(const float4 (*ring)[MDB_WAVE], int lane, uint32_t position) {
    return reinterpret_cast<const float *>(&ring[(position >> 2) % GROUPS][lane])[position & 3u];
}

// HAS_TS: timestamps are loaded (irregular series, k_fit_regular has found all of them within +-2^52, so that
// (f64)t and differences of such are exact: SwingFast's arithmetic on SwingDev's inputs). Their ring holds
// pairs; with 24 bytes per point and lane instead of 4 the rings are half as long, so that as many waves
// fit into a CU.
//
// ROTATE: a lane is a chunk, a wave 64 chunks that it steps through from their first point to their last - so a call's
// waves are indivisible, and 2 391 of them (the bench's 153 000 chunks) on 1 024 SIMDs mean that 343 SIMDs hold three
// and 681 two: a wave that shares its SIMD with two others steps at 0.85 of the rate of one that shares it with one
// (measured: 2 048 waves take the 2 391 groups through in 62.9 ms, 2 391 waves in 63.7), the call lasts as long as
// the slow ones, and the fast ones' SIMDs idle at its end. With ROTATE a wave works on a group for a STRETCH of steps
// only: it then writes the lanes' fitters to memory (LeanSaved: 120 bytes a lane), puts the group at the tail of a
// queue and takes the one at its head - the one that has waited longest. There are a few waves fewer than groups, so
// no wave ever waits and a handful of groups do: every group is on a slow SIMD for its share of stretches and on a
// fast one for the rest, and they all end together (63.9 -> 54.4 ms for the bench's call, stretches of 512 steps;
// docs/NOTES_r05.md has the sweep). The ring needs no saving: a lane whose next point lies outside what its ring
// holds starts the ring again at that point.
struct LeanSaved { // one lane's state between two stretches
    uint32_t n_models, current, j, pmc_length, swing_length;
    uint32_t gaps_have_previous, gaps_previous_end, gaps_segments;
    float pmc_min, pmc_max;
    double pmc_sum, swing_start, swing_first, upper_slope, upper_intercept, lower_slope, lower_intercept, numerator,
        denominator, swing_end;
};
static_assert(sizeof(LeanSaved) == 120, "LeanSaved is sized by hand");
constexpr int LEAN_SAVED_WORDS = sizeof(LeanSaved) / 8;

// A group's saved state goes from one wave to another - on another XCD with an L2 of its own, as likely as not - many
// times a kernel. It is written and read as 64-bit words with device-scope atomic stores and loads (written through the
// L2, read past what the L2 holds), so that a handover needs no write-back and no invalidation of a whole L2 (that is
// what a device-scope fence is on this part: with one on either side of every handover the L2s were emptied every few
// microseconds, and the kernel's other waves paid for it). Laid out [group][word][lane]: a wave's store of one word is
// 512 consecutive bytes.
__device__ __forceinline__ void lean_saved_store(unsigned long long *saved, uint32_t group, int lane, const LeanSaved &state) {
    unsigned long long words[LEAN_SAVED_WORDS];
    __builtin_memcpy(words, &state, sizeof(LeanSaved));
    unsigned long long *at = saved + (uint64_t)group * (LEAN_SAVED_WORDS * MDB_WAVE) + lane;
#pragma unroll
    for (int w = 0; w < LEAN_SAVED_WORDS; w++) __hip_atomic_store(at + w * MDB_WAVE, words[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ LeanSaved lean_saved_load(const unsigned long long *saved, uint32_t group, int lane) {
    unsigned long long words[LEAN_SAVED_WORDS];
    const unsigned long long *at = saved + (uint64_t)group * (LEAN_SAVED_WORDS * MDB_WAVE) + lane;
#pragma unroll
    for (int w = 0; w < LEAN_SAVED_WORDS; w++) words[w] = __hip_atomic_load(at + w * MDB_WAVE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    LeanSaved state;
    __builtin_memcpy(&state, words, sizeof(LeanSaved));
    return state;
}

struct LeanRotation {
    unsigned int *ready = nullptr;          // the queue's slots: a group's number + 1, or 0
    uint32_t slot_mask = 0;                 // slots - 1 (a power of two, more than groups + waves)
    unsigned int *counters = nullptr;       // [0] places taken at the queue's head, [1] places given at its tail, [2] groups through
    uint32_t n_groups = 0;
    uint32_t stretch_steps = 0;
    unsigned long long *saved = nullptr;    // [group][word of LeanSaved][lane]
    unsigned long long *masks = nullptr;    // [group][4]: active, PMC-Mean fits, Swing fits, Swing's first value finite
    unsigned int *error = nullptr;          // the call's error word (ERR_ROTATION_STALL)
    uint32_t max_naps = 0;                  // naps a wave takes at an empty place before it gives the call up
};
constexpr int LEAN_MASK_WORDS = 4;
constexpr uint32_t LEAN_STRETCH_STEPS = 512; // (256 to 1 024 are within a percent of each other; 128 and 4 096 cost 6 %)

// The queue begins with every group in it, in order.
__global__ void k_fit_rotation_begin(LeanRotation rotation) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot <= rotation.slot_mask) rotation.ready[slot] = slot < rotation.n_groups ? slot + 1u : 0u;
    if (slot < rotation.n_groups) {
        unsigned long long *group_masks = rotation.masks + (uint64_t)slot * LEAN_MASK_WORDS;
        group_masks[0] = ~0ull; // (of the lanes that have points: the kernel knows which)
        group_masks[1] = ~0ull;
        group_masks[2] = ~0ull;
        group_masks[3] = 0ull;
    }
    // Every lane's fitters as they are before a chunk's first point: a group is "restored" the first time as well
    // (one way into the kernel's loop, not two).
    if (slot < (uint64_t)rotation.n_groups * MDB_WAVE) {
        const double nan64 = __longlong_as_double(0x7ff8000000000000ll);
        const float nan32 = __uint_as_float(0x7fc00000u);
        LeanSaved fresh;
        fresh.n_models = fresh.current = fresh.j = fresh.pmc_length = fresh.swing_length = 0u;
        fresh.gaps_have_previous = fresh.gaps_previous_end = fresh.gaps_segments = 0u;
        fresh.pmc_min = fresh.pmc_max = nan32;
        fresh.pmc_sum = 0.0;
        fresh.swing_start = 0.0;
        fresh.swing_first = nan64;
        fresh.upper_slope = fresh.upper_intercept = fresh.lower_slope = fresh.lower_intercept = nan64;
        fresh.numerator = fresh.denominator = 0.0;
        fresh.swing_end = 0.0;
        lean_saved_store(rotation.saved, slot / MDB_WAVE, (int)(slot % MDB_WAVE), fresh);
    }

    if (slot == 0) {
        rotation.counters[0] = 0u;
        rotation.counters[1] = rotation.n_groups;
        rotation.counters[2] = 0u;
    }
}

// The queue, a place at a time: a wave takes the next place at the head and waits for the group that is - or will be -
// put there (0xffffffff: every group is through); a wave that has worked on a group for a stretch puts it at the tail.
//
// What orders a handover. The state of a group (LeanSaved, the four masks) is written with RELAXED device-scope stores
// and read with relaxed device-scope loads; the handover itself is a relaxed store to / load of the group's place in
// the queue. Nothing here is a release or an acquire (at device scope those are a write-back and an invalidation of an
// XCD's whole L2, see lean_saved_store), so the order rests on two things that have to be stated:
//   * the HARDWARE: on gfx942 / gfx950 a wave's vector memory operations are counted by ONE counter (vmcnt) that
//     covers the sc1 write-through stores, so `s_waitcnt vmcnt(0)` in rotation_give means "my stores have reached the
//     memory side" (targets with a separate store counter - gfx10 and later - would need s_waitcnt_vscnt: refused
//     below), and loads are issued in program order, so the taker's loads of the state leave after its look at the
//     place has come back;
//   * the COMPILER, which must leave the queue's operations where they are written. With BOTH functions inlined into
//     k_fit_models_lean (ROCm 7.2's clang) the kernel is wrong - deterministically: the first rotation test fails in half a
//     second, the headline fit "takes" 42 ms instead of 54 because groups are dropped - and its code has a loop level
//     the source does not have (the queue's "nothing taken" path threaded into lean_group's loop). Measured in round 6
//     (scripts/r06/rotation_inline.sh, docs/rotation_inline/): either function inlined alone is right; both inlined are
//     right again with ANY `asm volatile(... ::: "memory")` at rotation_take's entry - an s_waitcnt of either counter,
//     or no instruction at all - and wrong with one at its exit or at rotation_give's entry only. So it is a
//     transformation across the top of rotation_take, not the hardware's order, and a barrier for the compiler there is
//     what keeps it out; a function call did the same by accident. Both functions carry the barriers at entry and exit
//     now and stay calls (the loop they are called from is long enough as it is; same speed either way, 54.0-54.9 ms).
// A wave that waits at a place for longer than any kernel runs (max_naps naps of some 14 us) sets ERR_ROTATION_STALL
// and reports "every group is through": the call fails instead of hanging.
#if !defined(__HIP_DEVICE_COMPILE__) || defined(__gfx942__) || defined(__gfx950__)
#define MDB_ROTATION_ORDERED_BY_VMCNT 1
#else
#error "rotation_give orders its stores with s_waitcnt vmcnt(0): valid where one counter covers loads and stores (gfx942, gfx950)"
#endif
#ifdef MDB_ROTATION_NO_BARRIER
__device__ __forceinline__ void compiler_barrier() {}
#else
__device__ __forceinline__ void compiler_barrier() { asm volatile("" ::: "memory"); }
#endif
// (scripts/r06/rotation_inline.sh: the experiments behind the paragraph above, never the product. MDB_ROTATION_INLINE: 1 = take
// inlined, 2 = give inlined, 3 = both; MDB_ROTATION_ENTRY_WAIT: the s_waitcnt a function's entry has, at the inlined places)
#ifndef MDB_ROTATION_INLINE
#define MDB_ROTATION_INLINE 0
#endif
#if MDB_ROTATION_INLINE & 1
#define MDB_ROTATION_TAKE_CALL __forceinline__
#else
#define MDB_ROTATION_TAKE_CALL __noinline__
#endif
#if MDB_ROTATION_INLINE & 2
#define MDB_ROTATION_GIVE_CALL __forceinline__
#else
#define MDB_ROTATION_GIVE_CALL __noinline__
#endif
// MDB_ROTATION_ENTRY_WAIT: bit 0 = at take's entry, bit 1 = at give's; MDB_ROTATION_ENTRY_WAIT_KIND: 0 all counters, 1 vmcnt, 2 lgkmcnt, 3 none (a barrier for the compiler only)
#ifndef MDB_ROTATION_ENTRY_WAIT
#define MDB_ROTATION_ENTRY_WAIT 0
#endif
#ifndef MDB_ROTATION_ENTRY_WAIT_KIND
#define MDB_ROTATION_ENTRY_WAIT_KIND 0
#endif
template <int PLACE> __device__ __forceinline__ void entry_wait() {
    if ((MDB_ROTATION_ENTRY_WAIT & PLACE) == 0) return;
    if (MDB_ROTATION_ENTRY_WAIT_KIND == 0) asm volatile("s_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0)" ::: "memory");
    if (MDB_ROTATION_ENTRY_WAIT_KIND == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MDB_ROTATION_ENTRY_WAIT_KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (MDB_ROTATION_ENTRY_WAIT_KIND == 3) asm volatile("" ::: "memory"); // (no instruction: the compiler alone)
}

__device__ MDB_ROTATION_TAKE_CALL uint32_t rotation_take(const LeanRotation rotation) {
    entry_wait<1>();
#if MDB_ROTATION_INLINE == 0
    compiler_barrier();
#endif
    unsigned int taken = 0;
    if (threadIdx.x % MDB_WAVE == 0) {
        const unsigned int place = atomicAdd(&rotation.counters[0], 1u);
        unsigned int *slot = rotation.ready + (place & rotation.slot_mask);
        for (uint32_t naps = 0;; naps++) {
            // (a look, not an atomic, while there is nothing: hundreds of waves wait at any time, and a stretch takes
            // milliseconds - a look every few microseconds is early enough and leaves the memory system alone)
            if (__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                taken = atomicExch(slot, 0u); // (nobody else waits at this place)
                break;
            }
            if (__hip_atomic_load(&rotation.counters[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= rotation.n_groups) break;
            if (naps >= rotation.max_naps) {
                atomicOr(rotation.error, ERR_ROTATION_STALL);
                break;
            }
            for (int nap = 0; nap < 4; nap++) __builtin_amdgcn_s_sleep(127);
        }
    }
    taken = (unsigned int)__builtin_amdgcn_readfirstlane((int)taken);
    // (No fence: what the wave that had the group before wrote about it is read with device-scope loads, and those
    // are issued after the look at the slot has come back - the barrier keeps the compiler to that.)
    compiler_barrier();
    return taken - 1u;
}

__device__ MDB_ROTATION_GIVE_CALL void rotation_give(const LeanRotation rotation, uint32_t group) {
    entry_wait<2>();
    // Everything written about the group (device-scope stores, written through) has arrived before it can be taken
    // again: the wave waits for its stores, no more (see lean_saved_store for the fence this is instead of).
    compiler_barrier();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (threadIdx.x % MDB_WAVE == 0) {
        const unsigned int place = atomicAdd(&rotation.counters[1], 1u);
        __hip_atomic_store(rotation.ready + (place & rotation.slot_mask), group + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    compiler_barrier();
}

__device__ __forceinline__ LaneMask uniform_mask(unsigned long long loaded) {
    const uint32_t low = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)loaded);
    const uint32_t high = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(loaded >> 32));
    return ((LaneMask)high << 32) | low;
}

// One group of 64 chunks (ROTATE: for a stretch): the kernel's body; the plain kernel is this function and nothing
// else, the rotating one calls it for group after group. Its loop over the steps has to come out of the compiler as
// in the plain kernel, and three things see to that (each of them cost a fifth to a half of the kernel's time while it
// was missing): the group's number and the lane masks read from memory go through readfirstlane (they are in scalar
// registers as blockIdx.x and constants are), the state is loaded whether the group has been begun or not (one way
// into the loop: k_fit_rotation_begin writes a fresh state for every group), and the loop has ONE exit, its condition
// (a `break` at the stretch's end made the compiler copy the lanes' whole state from register to register every step).
template <bool SPLIT, int KIND, bool HAS_TS, bool ROTATE>
__device__ __forceinline__ void lean_group(const FitArgs &args, const SplitArgs &split,
                                           const unsigned long long *__restrict__ record_base, ModelRec *__restrict__ records,
                                           ChunkPlan *__restrict__ plans, unsigned int *__restrict__ error,
                                           const LeanRotation &rotation, float4 (*ring)[MDB_WAVE], longlong2 (*ring_ts)[MDB_WAVE],
                                           const int lane, const uint32_t group_of_wave) {
    static_assert(KIND == MDB_EB_RELATIVE || KIND == MDB_EB_ABSOLUTE || KIND == MDB_EB_LOSSLESS, "an error bound's kind");
    constexpr int LEAN_GROUPS = HAS_TS ? 4 : mdb::LEAN_GROUPS; // (shadow the constants of the values-only form)
    constexpr int LEAN_LOADS = HAS_TS ? 2 : mdb::LEAN_LOADS;
    constexpr int LEAN_TRASH = LEAN_GROUPS;
    static_assert(!(ROTATE && SPLIT), "split mode's pieces are not rotated");
    const uint64_t unit = (uint64_t)group_of_wave * FIT_THREADS + lane;
    uint64_t chunk = unit;
    bool active = unit < (SPLIT ? split.n_pieces : args.n_chunks);
    uint32_t first_point = 0;
    if (SPLIT && active) {
        uint64_t lo = 0, hi = args.n_chunks;
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) / 2;
            if (split.piece_base[mid] <= unit) lo = mid;
            else hi = mid;
        }
        chunk = lo;
        first_point = (uint32_t)(unit - split.piece_base[chunk]) * split.piece_points;
    }
    uint64_t base = 0;
    uint32_t n = 0;
    if (active) {
        base = args.chunk_offsets[chunk];
        const uint64_t length64 = args.chunk_offsets[chunk + 1] - base;
        if (length64 > COUNT_MASK - ENTRY_END_BIAS) {
            atomicOr(error, ERR_TOO_LONG);
            plans[chunk] = {0, 0};
            active = false;
        } else {
            n = (uint32_t)length64;
        }
    }
    if (!SPLIT && active && n == 0) {
        plans[chunk] = {0, 0};
        active = false;
    }
    // Lanes without points read (and ignore) the first group of the call's values.
    const float *chunk_values = args.values + (active ? base : 0);
    const uint32_t misalign = (uint32_t)((reinterpret_cast<uintptr_t>(chunk_values) >> 2) & 3u);
    const float4 *__restrict__ groups = reinterpret_cast<const float4 *>(chunk_values - misalign);
    const uint32_t last_group = active ? (n - 1 + misalign) >> 2 : 0u;
    const ChunkTimestamps regular_ts = chunk_timestamps(args.timestamps, active ? chunk : 0, base);
    const double first_time = (double)regular_ts.first, interval_time = (double)regular_ts.interval; // exact
    const int64_t *__restrict__ chunk_ts = HAS_TS ? args.timestamps.ts + (active ? base : 0) : nullptr;
    const mdb_error_bound eb = args.eb;
    const PmcFast pmc_fast = pmc_fast_constants(eb);
    const double deviation_factor_value = deviation_factor(eb).factor;
    const double nan64 = __longlong_as_double(0x7ff8000000000000ll);
    const float nan32 = __uint_as_float(0x7fc00000u);
    ModelRec *__restrict__ out = records + ((active && !SPLIT) ? record_base[chunk] : 0);

    uint32_t n_models = 0;
    GapCounter gaps;
    const uint32_t piece_end = SPLIT ? first_point + split.piece_points : 0u;
    uint32_t current = first_point; // first point of the model being fitted
    uint32_t j = first_point;       // next point to feed
    // The ring holds groups [low_group, loaded_group) of the chunk, at most LEAN_GROUPS of them.
    uint32_t loaded_group = (first_point + misalign) >> 2, low_group = loaded_group;
    // PMC-Mean (PmcDev) and Swing (SwingFast) of the model being fitted, in registers.
    float pmc_min = nan32, pmc_max = nan32;
    double pmc_sum = 0.0;
    uint32_t pmc_length = 0;
    double swing_start = 0.0, swing_first = nan64;
    double upper_slope = nan64, upper_intercept = nan64, lower_slope = nan64, lower_intercept = nan64;
    double numerator = 0.0, denominator = 0.0;
    double swing_end = 0.0; // (HAS_TS) time of the last point the model has accepted
    uint32_t swing_length = 0;
    // What is true of which lane is kept as 64-bit lane masks: they are the same for the whole wave, so
    // they live in scalar registers and `and`, `or`, `not` of conditions are scalar instructions. (As
    // `bool` variables that survive an iteration the compiler keeps them as 0 / 1 in vector registers
    // and converts to and fro: some forty vector instructions of a step, a fifth of it.)
    LaneMask active_m = lanes_where(active);
    LaneMask pmc_fits_m = ~0ull, swing_fits_m = ~0ull, swing_finite_m = 0;
    const LaneMask pmc_fast_m = pmc_fast.enabled ? ~0ull : 0ull;
    // (split mode under a lossy bound) k_fit_reject_flags' bits of 128 points from point `flags_first` of the chunk on -
    // the two words that every top-up of the ring fetches with the values - and the words of this lane's chunk.
    constexpr bool FLAGGED = SPLIT && KIND != MDB_EB_LOSSLESS && !HAS_TS;
    const bool flagged = FLAGGED && split.reject_words != nullptr;
    unsigned long long flags_near = 0, flags_far = 0;
    uint32_t flags_first = 0;
    const unsigned long long *__restrict__ chunk_flag_words =
        flagged ? split.reject_words + (active ? split.piece_base[chunk] * split.reject_words_per_piece : 0ull) : nullptr;
    const uint32_t last_flag_word = n > 0 ? (n - 1u) >> 6 : 0u;
    uint32_t steps_left = ROTATE ? rotation.stretch_steps : 0u;
    if (ROTATE) {
        // The group goes on where it was left (the first time: where k_fit_rotation_begin says; the ring starts at the lane's next point).
        const unsigned long long *group_masks = rotation.masks + (uint64_t)group_of_wave * LEAN_MASK_WORDS;
        {
            const LeanSaved from = lean_saved_load(rotation.saved, group_of_wave, lane);
            n_models = from.n_models;
            current = from.current;
            j = from.j;
            pmc_length = from.pmc_length;
            swing_length = from.swing_length;
            gaps.have_previous = from.gaps_have_previous != 0u;
            gaps.previous_end = from.gaps_previous_end;
            gaps.n_segments = from.gaps_segments;
            pmc_min = from.pmc_min;
            pmc_max = from.pmc_max;
            pmc_sum = from.pmc_sum;
            swing_start = from.swing_start;
            swing_first = from.swing_first;
            upper_slope = from.upper_slope;
            upper_intercept = from.upper_intercept;
            lower_slope = from.lower_slope;
            lower_intercept = from.lower_intercept;
            numerator = from.numerator;
            denominator = from.denominator;
            swing_end = from.swing_end;
            // (the same for every lane, and the compiler is to know it)
            active_m &= uniform_mask(__hip_atomic_load(group_masks + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            pmc_fits_m = uniform_mask(__hip_atomic_load(group_masks + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            swing_fits_m = uniform_mask(__hip_atomic_load(group_masks + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            swing_finite_m = uniform_mask(__hip_atomic_load(group_masks + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            loaded_group = (j + misalign) >> 2;
            low_group = loaded_group;
        }
    }
#ifdef MDB_FIT_TIMING
#ifdef MDB_FIT_TIMING_SPLIT // (the same regions in split mode: scripts/r04/build_timing.sh with -DMDB_FIT_TIMING_SPLIT)
    constexpr bool TIMED = SPLIT && KIND == MDB_EB_RELATIVE && !HAS_TS;
#else
    constexpr bool TIMED = !SPLIT && KIND == MDB_EB_RELATIVE && !HAS_TS;
#endif
    unsigned long long timing_cycles = 0, timing_passes = 0, timing_steps = 0;
    const unsigned long long timing_loop_t0 = __builtin_amdgcn_s_memtime(), timing_real_t0 = __builtin_amdgcn_s_memrealtime();
#endif

    while (active_m != 0 && (!ROTATE || steps_left != 0u)) { // (ROTATE: or the stretch is over)
#ifdef MDB_FIT_TIMING
        timing_steps += 1;
        { FIT_TIMING_BEGIN(0); FIT_TIMING_END(0) }
#endif
        if (KIND == MDB_EB_LOSSLESS) {
            // Noise under a lossless bound: a start point whose next value differs (PMC-Mean ends at one point) and
            // whose third one is off the line through the first two (Swing ends at two) is rejected, which the three
            // points say at once (k_fit_models has the same shortcut and the reasoning): runs of such start points
            // are skipped here while their points are in the ring.
            for (int skipped = 0; skipped < FIT_QUICK_REJECTS; skipped++) {
                const uint32_t start = current + misalign;
                const LaneMask candidate_m = active_m & lanes_where(j == current) & lanes_where(current + 2 < n) &
                                             lanes_where((start >> 2) >= low_group) &
                                             lanes_where(((start + 2) >> 2) < loaded_group);
                if (candidate_m == 0) break;
                const float v0 = ring_value<LEAN_GROUPS>(ring, lane, start);
                const float v1 = ring_value<LEAN_GROUPS>(ring, lane, start + 1);
                const float v2 = ring_value<LEAN_GROUPS>(ring, lane, start + 2);
                double t0, t1, t2;
                if (HAS_TS) {
                    const longlong2 a = ring_ts[(start >> 1) % (2 * LEAN_GROUPS)][lane];
                    const longlong2 b = ring_ts[((start + 1) >> 1) % (2 * LEAN_GROUPS)][lane];
                    const longlong2 c = ring_ts[((start + 2) >> 1) % (2 * LEAN_GROUPS)][lane];
                    t0 = (double)((start & 1u) ? a.y : a.x);
                    t1 = (double)((start & 1u) ? b.x : b.y);
                    t2 = (double)((start & 1u) ? c.y : c.x);
                } else {
                    t0 = __builtin_fma((double)current, interval_time, first_time);
                    t1 = __builtin_fma((double)(current + 1), interval_time, first_time);
                    t2 = __builtin_fma((double)(current + 2), interval_time, first_time);
                }
                const double slope = ((double)v1 - (double)v0) / (t1 - t0); // line_through_exact of two different values
                const double approximation = slope * t2 + ((double)v0 - slope * t0);
                const LaneMask rejected_m = candidate_m & lanes_where(__builtin_amdgcn_classf(v0, 0x1f8)) &
                                            lanes_where(__builtin_amdgcn_classf(v1, 0x1f8)) &
                                            lanes_where(__builtin_amdgcn_classf(v2, 0x1f8)) & lanes_where(v0 != v1) &
                                            (lanes_where(approximation < (double)v2) | lanes_where(approximation > (double)v2));
                if (rejected_m == 0) break;
                bool met = false;
                if (in_lanes(rejected_m)) {
                    if (SPLIT)
                        __hip_atomic_store(&split.entry[base + current], ENTRY_REJECTED, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                    current += 1;
                    j = current;
                    // (past the end of its piece a lane looks for another lane's tracks)
                    if (SPLIT && current >= piece_end &&
                        __hip_atomic_load(&split.entry[base + current], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
                        met = true;
                }
                if (SPLIT) active_m &= ~lanes_where(met);
            }
            if (active_m == 0) break;
        }
        if (FLAGGED && flagged) {
            // Start points at which no model can begin (k_fit_reject_flags: both fitters certainly end before their eighth
            // point; their entries say so already) are passed over without a point being fed: a whole run of them at once,
            // as far as the lane's two words of bits reach - which is further than its ring of values does, so a lane in
            // a long run of them asks for memory once per hundred points, not once per ring.
            for (int more = 0; more < 2; more++) { // (a run of more than 64: the second word's worth)
                const uint32_t offset = current - flags_first; // (unsigned: a start point in front of the words is out of them too)
                const unsigned long long window = offset < 64u ? (flags_near >> offset) | (offset ? flags_far << (64u - offset) : 0ull)
                                                               : flags_far >> (offset & 63u);
                const LaneMask candidate_m = active_m & lanes_where(j == current) & lanes_where(offset < 128u) & lanes_where((window & 1ull) != 0ull);
                if (candidate_m == 0) break;
                bool met = false;
                if (in_lanes(candidate_m)) {
                    const uint32_t ones = ~window ? (uint32_t)__builtin_ctzll(~window) : 64u;
                    current += min(ones, 128u - offset); // (a set bit says that seven more points of the chunk follow: current < n)
                    j = current;
                    // (past the end of its piece a lane looks for another lane's tracks)
                    if (current >= piece_end &&
                        entry_is_a_track(__hip_atomic_load(&split.entry[base + current], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
                        met = true;
                }
                active_m &= ~lanes_where(met);
            }
            if (active_m == 0) break;
        }
        const LaneMask feeding_m = active_m & lanes_where(j < n) & (pmc_fits_m | swing_fits_m);
        const uint32_t position = j + misalign; // of point j, counted from the 16-byte boundary
        const uint32_t group = position >> 2;
        if (feeding_m & (lanes_where(group >= loaded_group) | lanes_where(group < low_group))) {
            FIT_TIMING_BEGIN(1);
            // Normally the ring is extended at loaded_group. A lane whose next point fell out of the back
            // of its ring (PMC-Mean chosen although Swing had run far ahead, types.rs:84-101) or lies
            // beyond it restarts the ring at that point's group.
            const bool lane_active = in_lanes(active_m);
            if (lane_active & ((group < low_group) | (group > loaded_group))) {
                loaded_group = group;
                low_group = group;
            }
            const uint32_t first_group = loaded_group;
            // Two groups behind the current one stay (a rejected model restarts at most 7 points back).
            const uint32_t end_group = lane_active ? min(group + (uint32_t)LEAN_LOADS, last_group + 1u) : first_group;
            float4 fetched[LEAN_LOADS];
            long long fetched_ts[HAS_TS ? LEAN_LOADS : 1][4];
#pragma unroll
            for (int k = 0; k < LEAN_LOADS; k++) {
                const uint32_t g = min(first_group + (uint32_t)k, last_group);
                fetched[k] = groups[g];
                if (HAS_TS) {
                    // The four points of the group, the chunk's first or last one where the group reaches
                    // beyond it (a chunk may start and end anywhere in its first and last group).
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const uint32_t slot = 4u * g + (uint32_t)q;
                        const uint32_t index = slot < misalign ? 0u : min(slot - misalign, n - 1u);
                        fetched_ts[k][q] = (active && n > 0) ? chunk_ts[index] : 0ll;
                    }
                }
            }
            unsigned long long flags_low = 0, flags_high = 0;
            // (the words from the one with the point the lane feeds next)
            const uint32_t flags_word = min(j >> 6, last_flag_word);
            if (FLAGGED && flagged) { // (with the values' loads: one wait for all)
                flags_low = chunk_flag_words[flags_word];
                flags_high = chunk_flag_words[min(flags_word + 1u, last_flag_word)];
            }
#pragma unroll
            for (int k = 0; k < LEAN_LOADS; k++) {
                const uint32_t g = first_group + (uint32_t)k;
                ring[g < end_group ? g % LEAN_GROUPS : (uint32_t)LEAN_TRASH][lane] = fetched[k];
                if (HAS_TS) {
                    const uint32_t row = g < end_group ? 2u * (g % LEAN_GROUPS) : 2u * (uint32_t)LEAN_GROUPS;
                    ring_ts[row][lane] = make_longlong2(fetched_ts[k][0], fetched_ts[k][1]);
                    ring_ts[row + 1][lane] = make_longlong2(fetched_ts[k][2], fetched_ts[k][3]);
                }
            }
            if (FLAGGED && flagged && lane_active) {
                flags_near = flags_low;
                flags_far = flags_word < last_flag_word ? flags_high : 0ull; // (the chunk's last word has no successor)
                flags_first = flags_word << 6;
            }
            loaded_group = max(first_group, end_group);
            if (loaded_group > low_group + LEAN_GROUPS) low_group = loaded_group - LEAN_GROUPS;
#ifdef MDB_FIT_TIMING
            if (TIMED && MDB_FIT_TIMING == 1) __builtin_amdgcn_s_waitcnt(0); // (the ring's stores have left)
#endif
            FIT_TIMING_END(1)
        }
        FIT_TIMING_BEGIN(2);
        const float value32 = ring_value<LEAN_GROUPS>(ring, lane, position);
#ifdef MDB_FIT_TIMING
        if (TIMED && MDB_FIT_TIMING == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::"v"(value32));
#endif
        FIT_TIMING_END(2)
        const double value = (double)value32;
        FIT_TIMING_BEGIN(3);

        // ---- PMC-Mean: PMCMean::fit_value (pmc_mean.rs:58-76), decided as in pmc_fit_fast ----
        // (What the two fitters compute for a step does not depend on each other: PMC-Mean's certain test and Swing's
        // lines stand next to each other in one stretch of straight-line code, so that one's instructions fill the
        // waits of the other's; what is rare for the whole wave - the exact PMC-Mean test, Swing's second line - and
        // what is kept of the step comes behind both.)
        const LaneMask pmc_steps_m = feeding_m & pmc_fits_m;
        const float next_min = min_num(pmc_min, value32);
        const float next_max = max_num(pmc_max, value32);
        const double next_sum = pmc_sum + value;
        const uint32_t next_length = pmc_length + 1;
        LaneMask pmc_accepts_m, doubtful_m;
        float approximate = 0.0f, average_error = 0.0f;
        if (KIND == MDB_EB_LOSSLESS) {
            // Within a lossless bound of the average are a minimum and a maximum that both equal it. Certain: every
            // value so far and this one are the same finite number and few enough for every partial sum to be
            // exact (a 24-bit significand times a count below 2^24), so the average is that number. Certainly not:
            // a minimum below the maximum. Anything else (NaNs, infinities, very long models): the exact test.
            pmc_accepts_m = lanes_where(__builtin_amdgcn_classf(value32, 0x1f8)) & lanes_where(next_length < (1u << 24)) &
                            (lanes_where(pmc_length == 0) | (lanes_where(pmc_min == value32) & lanes_where(pmc_max == value32)));
            doubtful_m = pmc_steps_m & ~pmc_accepts_m & ~lanes_where(next_min < next_max);
        } else {
            approximate = (float)next_sum * __builtin_amdgcn_rcpf((float)next_length);
            average_error = fmaxf(fabsf(approximate) * 0x1p-20f, 0x1p-120f);
            pmc_accepts_m = pmc_fast_m & lean_passes<KIND>(pmc_fast, next_min, approximate, average_error) &
                            lean_passes<KIND>(pmc_fast, next_max, approximate, average_error);
            // Not certainly within the bound: certainly outside it (the step that ends a PMC-Mean model,
            // once per model), or too close to call and decided by the exact test.
            doubtful_m = pmc_steps_m & ~pmc_accepts_m;
        }

        // ---- Swing: Swing::fit_data_point (swing.rs:101-198) on exact f64 timestamps (SwingFast) ----
        const LaneMask swing_steps_m = feeding_m & swing_fits_m;
        // first + j * interval: integers below 2^53, exact whether fused or not. (HAS_TS: the timestamp
        // itself, exact for the same reason.)
        double time;
        if (HAS_TS) {
            const longlong2 pair = ring_ts[(position >> 1) % (2 * LEAN_GROUPS)][lane];
            time = (double)((position & 1u) ? pair.y : pair.x);
        } else {
            time = __builtin_fma((double)j, interval_time, first_time);
        }
        const double deviation = lean_deviation<KIND>(deviation_factor_value, value);
        const LaneMask first_m = lanes_where(swing_length == 0), second_m = lanes_where(swing_length == 1);
        const LaneMask later_m = ~(first_m | second_m);
        const LaneMask value_finite_m = lanes_where(__builtin_amdgcn_classf(value32, 0x1f8)); // isfinite: +-normal, +-subnormal, +-0
        // A non-finite first value only accepts copies of itself, a finite one no non-finite value
        // (swing.rs:113-125): rare, behind a branch below.
        const LaneMask special_m = ~first_m & ~(swing_finite_m & value_finite_m);
        const double upper_approximation = upper_slope * time + upper_intercept;
        const double lower_approximation = lower_slope * time + lower_intercept;
        const LaneMask outside_m = later_m & (lanes_where(upper_approximation + deviation < value) |
                                              lanes_where(lower_approximation - deviation > value));
        const LaneMask lowers_upper_m = second_m | (later_m & lanes_where(upper_approximation - deviation > value));
        const LaneMask raises_lower_m = second_m | (later_m & lanes_where(lower_approximation + deviation < value));
        const bool lowers_upper = in_lanes(lowers_upper_m);
        // One line per step serves whichever bound moves (line_through_exact, start and first value of
        // the model); a step that moves both gets its second line behind the branch.
        const double target = lowers_upper ? value + deviation : value - deviation;
        const double elapsed = time - swing_start;
        double line_slope = (target - swing_first) / elapsed;
        double line_intercept = swing_first - line_slope * swing_start;
        LaneMask swing_accepts_m = first_m | ~outside_m;
        double second_slope = lower_slope, second_intercept = lower_intercept;
        const LaneMask moves_both_m = swing_steps_m & ~special_m & lowers_upper_m & raises_lower_m & ~outside_m;
        // (a line to a value equal to the model's first one is flat, swing.rs:331-333; and such a value
        // adds nothing to the sums of swing.rs:212-228)
        const LaneMask level_m = swing_steps_m & (lanes_where(swing_first == target) | lanes_where(swing_first == value));
        const LaneMask special_steps_m = swing_steps_m & special_m;
        double weighted = (value - swing_first) * elapsed, squared = elapsed * elapsed;

        // ---- PMC-Mean, the rest: the exact test where the certain one does not say, and what is kept ----
        if (doubtful_m) {
            bool exactly_within = false;
            if (in_lanes(doubtful_m)) {
                const int low = (KIND != MDB_EB_LOSSLESS && pmc_fast.enabled) ? pmc_fast_within(pmc_fast, next_min, approximate, average_error) : 0;
                const int high = (KIND != MDB_EB_LOSSLESS && pmc_fast.enabled) ? pmc_fast_within(pmc_fast, next_max, approximate, average_error) : 0;
                if (low >= 0 && high >= 0) {
                    const float average = (float)(next_sum / (double)next_length);
                    exactly_within = within_error_bound(eb, next_min, average) && within_error_bound(eb, next_max, average);
                }
            }
            pmc_accepts_m |= lanes_where(exactly_within);
        }
        if (in_lanes(pmc_steps_m & pmc_accepts_m)) { // (kept under the lanes' mask: no select per register)
            keep_under_mask();
            pmc_min = next_min;
            pmc_max = next_max;
            pmc_sum = next_sum;
            pmc_length = next_length;
        }
        pmc_fits_m &= ~pmc_steps_m | pmc_accepts_m;
#ifdef MDB_FIT_TIMING
        if (TIMED && MDB_FIT_TIMING == 3) asm volatile("" ::"v"(pmc_min), "v"(pmc_max), "v"(pmc_sum), "v"(pmc_length), "s"(pmc_fits_m));
#endif
        FIT_TIMING_END(3)
        FIT_TIMING_BEGIN(4);

        // ---- Swing, the rest ----
        if (moves_both_m | level_m | special_steps_m) {
            if (in_lanes(moves_both_m)) {
                const LineDev line = line_through_exact(swing_start, swing_first, time, value - deviation);
                second_slope = line.slope;
                second_intercept = line.intercept;
            }
            if (in_lanes(level_m)) {
                if (swing_first == target) {
                    line_slope = 0.0;
                    line_intercept = swing_first;
                }
                if (swing_first == value) {
                    weighted = 0.0;
                    squared = 0.0;
                }
            }
            if (special_steps_m)
                swing_accepts_m = (swing_accepts_m & ~special_steps_m) | (special_steps_m & lanes_where(equal_or_nan(swing_first, value)));
        }
        const LaneMask swing_keeps_m = swing_steps_m & swing_accepts_m;
        const LaneMask plain_m = swing_keeps_m & ~special_m & ~first_m;
        if (in_lanes(plain_m & lowers_upper_m)) {
            keep_under_mask();
            upper_slope = line_slope;
            upper_intercept = line_intercept;
        }
        if (in_lanes(plain_m & raises_lower_m)) {
            keep_under_mask();
            lower_slope = lowers_upper ? second_slope : line_slope;
            lower_intercept = lowers_upper ? second_intercept : line_intercept;
        }
        if (in_lanes(plain_m & later_m)) { // swing.rs:212-228: only from the third point on
            keep_under_mask();
            numerator += weighted;
            denominator += squared;
        }
        const LaneMask opens_m = swing_keeps_m & (special_m | first_m); // (once per model, or never finite)
        if (opens_m) {
            if (in_lanes(opens_m & special_m)) // upper = lower = {value, value} (swing.rs:121-123)
                upper_slope = upper_intercept = lower_slope = lower_intercept = value;
            const LaneMask starts_m = opens_m & ~special_m;
            if (in_lanes(starts_m)) {
                swing_start = time;
                swing_first = value;
            }
            swing_finite_m = (swing_finite_m & ~starts_m) | (starts_m & value_finite_m);
        }
        if (HAS_TS) { // SwingDev::end_time: the model's last point is no longer (length - 1) intervals from its first
            if (in_lanes(swing_keeps_m)) {
                keep_under_mask();
                swing_end = time;
            }
        }
        swing_length += in_lanes(swing_keeps_m) ? 1u : 0u;
        swing_fits_m &= ~swing_steps_m | swing_accepts_m;
        j += in_lanes(feeding_m) ? 1u : 0u;
#ifdef MDB_FIT_TIMING
        if (TIMED && MDB_FIT_TIMING == 4) asm volatile("" ::"v"(upper_slope), "v"(upper_intercept), "v"(lower_slope), "v"(lower_intercept), "v"(numerator), "v"(denominator), "v"(swing_length), "s"(swing_fits_m));
#endif
        FIT_TIMING_END(4)

        const LaneMask finishing_m = active_m & ~feeding_m;
        if (finishing_m) {
            FIT_TIMING_BEGIN(5);
            bool ends = false; // this lane has no model left to fit
            if (in_lanes(finishing_m)) {
                // ModelBuilder::finish (types.rs:84-101): fewest bytes per value, PMC-Mean wins ties.
                const float pmc_bpv = (float)MDB_COMPRESSED_METADATA_SIZE_IN_BYTES / (float)pmc_length;
                const float swing_bpv = ((float)MDB_COMPRESSED_METADATA_SIZE_IN_BYTES + 1.0f) / (float)swing_length;
                const bool choose_pmc = pmc_bpv <= swing_bpv;
                const float bpv = choose_pmc ? pmc_bpv : swing_bpv;
                if (bpv <= (float)MDB_VALUE_SIZE_IN_BYTES) { // compression.rs:238
                    ModelRec rec;
                    if (choose_pmc) {
                        rec.start_and_type = current;
                        rec.end = current + pmc_length - 1;
                        rec.p0 = (float)(pmc_sum / (double)pmc_length); // pmc_mean.rs:91-93
                        rec.p1 = rec.p0;
                    } else { // swing.rs:246-259
                        rec.start_and_type = current | 0x80000000u;
                        rec.end = current + swing_length - 1;
                        const double projected = numerator / denominator;
                        const double slope = max_num(lower_slope, min_num(projected, upper_slope));
                        const double span = HAS_TS ? swing_end - swing_start : (double)(swing_length - 1) * interval_time;
                        const double last_value = slope * span + swing_first;
                        rec.p0 = (float)swing_first;
                        rec.p1 = (float)last_value;
                    }
                    if (SPLIT) {
                        split.p0[base + current] = rec.p0;
                        split.p1[base + current] = rec.p1;
                        __hip_atomic_store(&split.entry[base + current],
                                           (rec.end + ENTRY_END_BIAS) | (rec.start_and_type & 0x80000000u),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        gaps.on_model(current, rec.end);
                        out[n_models++] = rec;
                    }
                    current = rec.end + 1;
                } else {
                    if (SPLIT)
                        __hip_atomic_store(&split.entry[base + current], ENTRY_REJECTED, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                    current += 1; // the point becomes a residual (compression.rs:258-262)
                }
                if (current >= n) {
                    if (!SPLIT) plans[chunk] = {n_models, gaps.finish(n)};
                    ends = true;
                } else if (SPLIT && current >= piece_end &&
                           entry_is_a_track(__hip_atomic_load(&split.entry[base + current], __ATOMIC_RELAXED,
                                                              __HIP_MEMORY_SCOPE_AGENT))) {
                    ends = true; // some lane has been here: the chain from this point on is recorded
                } else {
                    pmc_min = nan32;
                    pmc_max = nan32;
                    pmc_sum = 0.0;
                    pmc_length = 0;
                    swing_start = 0.0;
                    swing_first = nan64;
                    upper_slope = upper_intercept = lower_slope = lower_intercept = nan64;
                    numerator = 0.0;
                    denominator = 0.0;
                    swing_length = 0;
                    j = current;
                }
            }
            const LaneMask ended_m = lanes_where(ends);
            const LaneMask next_model_m = finishing_m & ~ended_m;
            active_m &= ~ended_m;
            pmc_fits_m |= next_model_m;
            swing_fits_m |= next_model_m;
            swing_finite_m &= ~next_model_m;
            FIT_TIMING_END(5)
        }
        if (ROTATE) steps_left -= 1u;
    }
    if (ROTATE) {
        if (active_m != 0) {
            // Not through: the lanes' fitters go to memory, the group to the end of the queue.
            LeanSaved to;
            to.n_models = n_models;
            to.current = current;
            to.j = j;
            to.pmc_length = pmc_length;
            to.swing_length = swing_length;
            to.gaps_have_previous = gaps.have_previous ? 1u : 0u;
            to.gaps_previous_end = gaps.previous_end;
            to.gaps_segments = gaps.n_segments;
            to.pmc_min = pmc_min;
            to.pmc_max = pmc_max;
            to.pmc_sum = pmc_sum;
            to.swing_start = swing_start;
            to.swing_first = swing_first;
            to.upper_slope = upper_slope;
            to.upper_intercept = upper_intercept;
            to.lower_slope = lower_slope;
            to.lower_intercept = lower_intercept;
            to.numerator = numerator;
            to.denominator = denominator;
            to.swing_end = swing_end;
            lean_saved_store(rotation.saved, group_of_wave, lane, to);
            unsigned long long *group_masks = rotation.masks + (uint64_t)group_of_wave * LEAN_MASK_WORDS;
            if (lane == 0) {
                __hip_atomic_store(group_masks + 0, active_m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(group_masks + 1, pmc_fits_m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(group_masks + 2, swing_fits_m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(group_masks + 3, swing_finite_m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            rotation_give(rotation, group_of_wave);
        } else if (lane == 0) {
            atomicAdd(&rotation.counters[2], 1u);
        }
    }
#ifdef MDB_FIT_TIMING
    if (TIMED && lane == 0) {
        atomicAdd(&g_fit_timing[0], timing_cycles);
        atomicAdd(&g_fit_timing[1], timing_passes);
        atomicAdd(&g_fit_timing[2], __builtin_amdgcn_s_memtime() - timing_loop_t0);
        atomicAdd(&g_fit_timing[3], timing_steps);
        atomicAdd(&g_fit_timing[4], __builtin_amdgcn_s_memrealtime() - timing_real_t0);
        atomicAdd(&g_fit_timing[5], 1ull);
        if (MDB_FIT_TIMING == 6 && blockIdx.x < 65536) {
            g_fit_waves[4 * blockIdx.x] = __builtin_amdgcn_s_memrealtime() - timing_real_t0;
            g_fit_waves[4 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_REG_HW_ID
            g_fit_waves[4 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); // HW_REG_XCC_ID
            g_fit_waves[4 * blockIdx.x + 3] = timing_real_t0;
        }
    }
#endif
}

// ---- k_fit_reject_flags (split mode under a lossy bound) ------------------------------------------------------
//
// On data that no model fits (the Random runs of the reference's acceptance recipe, compression.rs:733-863, under a
// bound of 1 %) the greedy loop spends its time on start points it rejects: both fitters are fed until PMC-Mean has
// failed (by the second or third point) and Swing has (by the third or fourth), the builder finds that neither model
// pays for its 29 bytes (fewer than eight points: types.rs:84-101, compression.rs:238), the point becomes a residual
// and the same happens one point later - four to five steps of 107 cycles per point, and a lane of such a piece keeps
// its wave for four times as long as a lane whose piece is all models. Whether a start point is rejected is a function
// of the eight values from it on, so it is asked of all points at once here, one lane per point, conservatively:
//   * PMC-Mean certainly ends before its eighth point if for some m <= 8 the first m values' maximum and minimum are
//     further apart than any average can be from both (relative: |r - a| <= |r| e for r = min and r = max needs
//     max - min <= (|min| + |max|) e; absolute: max - min <= 2 e; pmc_mean.rs:58-76 with models/mod.rs's test, whose f32
//     roundings - three of 2^-24 - are covered by the factor 1 + 2^-16);
//   * Swing certainly ends before its eighth point if some point j in 2..7 lies outside the cone its first two points
//     open - the lines from (t0, v0) through (t1, v1 +- deviation(v1)), swing.rs:130-140 - by more than its own deviation:
//     every later pair of bounds goes through (t0, v0) with slopes between those two (a bound only moves inwards,
//     swing.rs:141-198). The fitter evaluates its lines as slope * t + intercept in f64, so what it sees differs from
//     the exact cone by rounding: the margin 2^-46 (|slope| (|t0| + |tj|) + |v0| + |v1| + |vj| + the cone's half width)
//     is 128 ulps of the largest term any of those evaluations has.
// Only where both are certain, all eight values are finite and all eight lie in the chunk is the point's bit set: the
// fitter (lean_group) takes a set bit as "rejected" without feeding a point, an unset one says nothing. Regular
// timestamps only (the lean fitter's values-only form).
constexpr uint32_t FLAG_PIECES = 8; // pieces per workgroup of k_fit_reject_flags
template <int KIND>
__global__ __launch_bounds__(256) void k_fit_reject_flags(FitArgs args, SplitArgs split, unsigned long long *__restrict__ words,
                                                          unsigned long long *__restrict__ bits_set) {
    // (a workgroup takes FLAG_PIECES pieces one after the other: finding the first one's chunk is a few trips to memory
    // one after the other, which a piece of 512 points is too little work to hide)
    const uint64_t first_unit = (uint64_t)blockIdx.x * FLAG_PIECES;
    if (first_unit >= split.n_pieces) return;
    const uint64_t unit0 = first_unit;
    // The piece's chunk: the last c with piece_base[c] <= unit (the same for the whole workgroup: scalar loads). Chunks
    // of one length have as many pieces each, so the search starts where that would put it and widens from there - a
    // bisection of 15 000 chunks is fourteen trips to memory one after the other, and there are millions of pieces.
    uint64_t lo = 0, hi = args.n_chunks;
    {
        const uint64_t guess = min(unit0 * args.n_chunks / split.n_pieces, args.n_chunks - 1);
        uint64_t reach = 1;
        if (split.piece_base[guess] <= unit0) {
            lo = guess;
            while (lo + reach < args.n_chunks && split.piece_base[lo + reach] <= unit0) {
                lo += reach;
                reach *= 2;
            }
            hi = min(lo + reach, args.n_chunks);
        } else {
            hi = guess;
            while (hi > reach && split.piece_base[hi - reach] > unit0) {
                hi -= reach;
                reach *= 2;
            }
            lo = hi > reach ? hi - reach : 0;
        }
    }
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) / 2;
        if (split.piece_base[mid] <= unit0) lo = mid;
        else hi = mid;
    }
    uint64_t chunk = lo;
    const int lane = threadIdx.x % MDB_WAVE, wave = threadIdx.x / MDB_WAVE;
    uint32_t set_here = 0;
    for (uint64_t unit = first_unit; unit < min(first_unit + FLAG_PIECES, split.n_pieces); unit++) {
    while (chunk + 1 < args.n_chunks && split.piece_base[chunk + 1] <= unit) chunk++; // (the next piece's chunk: this one or one close behind)
    const uint32_t first_point = (uint32_t)(unit - split.piece_base[chunk]) * split.piece_points;
    const uint64_t base = args.chunk_offsets[chunk];
    const uint64_t length64 = args.chunk_offsets[chunk + 1] - base;
    const uint32_t n = length64 > COUNT_MASK - ENTRY_END_BIAS ? 0u : (uint32_t)length64; // (too long: the fitter reports it)
    const ChunkTimestamps regular_ts = chunk_timestamps(args.timestamps, chunk, base);
    // Everything below is an upper bound of what the fitters allow, in f32 with room for its own roundings (a few of 2^-24
    // each against factors of 1 + 2^-12 and terms of 2^-15 of the magnitudes): a bit that is not set costs time, never a byte.
    const float roomy = 1.0f + 0x1p-12f;
    const float factor = (float)deviation_factor(args.eb).factor * roomy; // (of the deviation Swing allows: a bound from above)
    const float pmc_bound = (KIND == MDB_EB_RELATIVE ? args.eb.value / 100.0f : 2.0f * args.eb.value) * roomy;
    const float interval = fabsf((float)regular_ts.interval); // (a magnitude: a bound from above whatever its sign)
    // 2^-46 (|t0| + |tk|) / interval of any window of the piece, from above
    const float times = 0x1p-45f * (fabsf((float)regular_ts.first) + ((float)first_point + (float)split.piece_points + 8.0f) * interval) / interval * roomy;
    const float *__restrict__ values = args.values + base;
    for (uint32_t row = (uint32_t)wave; row < split.reject_words_per_piece; row += 256 / MDB_WAVE) {
        const uint32_t point = first_point + row * (uint32_t)MDB_WAVE + (uint32_t)lane;
        bool rejected = false;
        if (point < n && n - point > 7u) { // (all eight in the chunk)
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = values[point + (uint32_t)k];
            // (all eight finite: a product with zero of anything else is a NaN)
            float zero = 0.0f;
            float lowest = v[0], highest = v[0];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                zero = __builtin_fmaf(v[k], 0.0f, zero);
                lowest = fminf(lowest, v[k]);
                highest = fmaxf(highest, v[k]);
            }
            // PMC-Mean: had it accepted all eight, the eighth average would be within the bound of their minimum and of
            // their maximum
            const float magnitudes = fabsf(lowest) + fabsf(highest);
            const float allowed = KIND == MDB_EB_RELATIVE ? magnitudes * pmc_bound : pmc_bound;
            // (relative: magnitudes whose differences could be subnormal floats are left to the fitter)
            const bool pmc_ends = (KIND != MDB_EB_RELATIVE || magnitudes >= 0x1p-60f) && highest - lowest > allowed;
            // Swing: a point outside the first cone by more than the largest deviation among the eight and the room for
            // the roundings on either side
            const float rise = v[1] - v[0];
            const float deviation1 = (KIND == MDB_EB_RELATIVE ? fabsf(v[1]) * factor : factor) * roomy;
            const float largest = fmaxf(fabsf(lowest), fabsf(highest));
            const float deviation_any = KIND == MDB_EB_RELATIVE ? largest * factor : factor;
            const float constant = deviation_any * roomy + 0x1p-15f * largest + (fabsf(rise) + deviation1) * times;
            bool swing_ends = false;
#pragma unroll
            for (int k = 2; k < 8; k++) {
                const float middle = __builtin_fmaf((float)k, rise, v[0]);
                swing_ends = swing_ends || fabsf(v[k] - middle) > __builtin_fmaf((float)k, deviation1, constant);
            }
            rejected = zero == 0.0f && pmc_ends && swing_ends;
        }
        if (point < n) split.entry[base + point] = rejected ? ENTRY_FLAGGED : 0u;
        const unsigned long long word = __ballot(rejected);
        if (lane == 0) words[unit * split.reject_words_per_piece + row] = word;
        set_here += (uint32_t)__popcll(word);
    }
    // (how many there are decides whether the fitter looks at the bits at all: estimated from the first wave's rows of
    // every 8th workgroup - an addition to one address takes its turn behind all the others, some 7 ns each: one per
    // wave was 40 ms of them, one per wave of every 16th piece still 3)
    }
    if (lane == 0 && wave == 0 && set_here && (blockIdx.x & 7u) == 0u) atomicAdd(bits_set, (unsigned long long)set_here);
}

template <bool SPLIT, int KIND, bool HAS_TS = false, bool ROTATE = false>
__global__ __launch_bounds__(FIT_THREADS) void k_fit_models_lean(
    FitArgs args, SplitArgs split, const unsigned long long *__restrict__ record_base, ModelRec *__restrict__ records,
    ChunkPlan *__restrict__ plans, unsigned int *__restrict__ error, LeanRotation rotation) {
    constexpr int LEAN_GROUPS = HAS_TS ? 4 : mdb::LEAN_GROUPS;
    __shared__ float4 ring[LEAN_GROUPS + 1][MDB_WAVE];
    // Timestamps of the points of value group g: pairs 2g and 2g + 1 (row 2 * LEAN_GROUPS and the one behind it: trash).
    __shared__ longlong2 ring_ts[HAS_TS ? 2 * LEAN_GROUPS + 2 : 1][MDB_WAVE];
    const int lane = threadIdx.x;
    if (!ROTATE) {
        lean_group<SPLIT, KIND, HAS_TS, false>(args, split, record_base, records, plans, error, rotation, ring, ring_ts, lane, blockIdx.x);
        return;
    }
    for (;;) { // (group after group from the queue)
        const uint32_t group_of_wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)rotation_take(rotation)); // (in a scalar register, as blockIdx.x is)
        if (group_of_wave == 0xffffffffu) return; // (every group is through)
        lean_group<SPLIT, KIND, HAS_TS, ROTATE>(args, split, record_base, records, plans, error, rotation, ring, ring_ts, lane, group_of_wave);
    }
}

// ---- k_fit_models_wave: one WAVE per chunk ---------------------------------------------------------------------
//
// For calls with few chunks (an embedded-API series, some thousand ingest buffers) under an absolute or relative
// bound with regular timestamps that are exact in f64 (the regime of k_fit_models_lean). One lane per chunk
// leaves most of the GPU idle there, and speculative pieces (split mode) only help while models are shorter than
// a piece: a model that spans its chunk (a generous bound: PMC-Mean over all 65 536 points) is walked by every
// piece's lane to its end. Here the 64 lanes of a wave take 64 CONSECUTIVE points of ONE model per step:
//
// PMC-Mean (pmc_mean.rs:58-75)  The model accepts point i iff the minimum and maximum of points 0..i are within
//     the bound of (f32)(sum(0..i) / (i + 1)): three prefix scans and one test per lane, the first lane that fails
//     ends the model. min_num / max_num keep their first operand on ties, which is associative. The f64 sum is the
//     one place where the order of the additions could show - it cannot where every partial sum is exact: the
//     summands are f32 (24 bits) whose exponents span `spread` binades, so `length` of them add up below
//     2^(24 + spread + bits(length)) quanta; while that stays within 53 bits no addition rounds in ANY order. The
//     spread is tracked per model (a block that breaks the condition hands the model to one lane, below).
// Swing (swing.rs:101-198)      Both bounds are lines through the model's first point; a step replaces the upper
//     one by the line through (t, v + deviation) when that line is lower, the lower one likewise. Which line is
//     current at point i is a sequential chain - but a predictable one: in exact arithmetic the upper bound after
//     points 2..i-1 is the candidate with the smallest slope. So every lane computes its two candidate lines
//     (functions of the first point and its own), a scan finds the running extremes, and every lane evaluates the
//     reference's own three comparisons against the state the scan predicts for it. The prediction is only a
//     guess; what is exact is the check: the first lane that fails, or whose comparisons do not come out as the
//     scan assumed (a rounding-level disagreement), has - by induction over the lanes in front of it - seen the
//     true state, so its own outcome is the reference's; the wave takes it and scans again behind that lane.
//     The sums of model() (swing.rs:212-228, 246-259) are f64 sums of rounded products: sequential by nature, and
//     only needed once a Swing model has been chosen. They are queued and computed by one lane per model, 64
//     models at a time.
// Rejected start points (neither model reaches the 8 points it needs, compression.rs:238, 258-262) would cost a
// step each; after two in a row the start points that follow are tried by one lane each - on three points first,
// which is where noise is rejected, the survivors of up to 1 024 start points then on the eight that decide - and
// the run of rejected ones is skipped at once.
// A model that meets a non-finite value or an inexact sum is fitted by lane 0 with the plain fitters.
// Same records as the other kernels: every fit test runs with MDB_FIT_WAVE=1 (always), 0 (never) and the default.
struct PendingSwing {
    uint32_t record;
    uint32_t start;
    uint32_t length;
    double lower_slope, upper_slope;
};

// Data-parallel primitives: values move between the lanes of a wave inside the vector ALU (no trip through LDS).
template <int CONTROL, typename T> __device__ __forceinline__ T dpp_move(T x) {
    static_assert(sizeof(T) % 4 == 0, "made of 32-bit words");
    int words[sizeof(T) / 4];
    __builtin_memcpy(words, &x, sizeof(T));
#pragma unroll
    for (size_t k = 0; k < sizeof(T) / 4; k++)
        words[k] = __builtin_amdgcn_update_dpp(words[k], words[k], CONTROL, 0xf, 0xf, false);
    __builtin_memcpy(&x, words, sizeof(T));
    return x;
}

// Inclusive scan over the 64 lanes; op(earlier, later) need not commute. Within rows of 16 lanes by shifts of
// 1, 2, 4, 8, then the last lane of a row to the row above, then lane 31 to the upper half.
template <typename T, typename Op> __device__ __forceinline__ T wave_inclusive_scan(T x, int lane, Op op) {
    const int row_lane = lane & 15;
    { const T t = dpp_move<0x111>(x); if (row_lane >= 1) x = op(t, x); }          // row_shr:1
    { const T t = dpp_move<0x112>(x); if (row_lane >= 2) x = op(t, x); }          // row_shr:2
    { const T t = dpp_move<0x114>(x); if (row_lane >= 4) x = op(t, x); }          // row_shr:4
    { const T t = dpp_move<0x118>(x); if (row_lane >= 8) x = op(t, x); }          // row_shr:8
    { const T t = dpp_move<0x142>(x); if ((lane & 31) >= 16) x = op(t, x); }      // row_bcast:15
    { const T t = dpp_move<0x143>(x); if (lane >= 32) x = op(t, x); }             // row_bcast:31
    return x;
}

// The same scan without the lane tests: a lane that has no source lane for a step (the first k lanes of a row, the
// rows a broadcast does not reach) receives `otherwise(x)` instead - its own value where op(x, x) = x (minimum,
// maximum), the neutral element where there is one (0 for a sum) - so that op can be applied by every lane.
template <int CONTROL, int ROW_MASK, typename T> __device__ __forceinline__ T dpp_move_or(T otherwise, T x) {
    static_assert(sizeof(T) % 4 == 0, "made of 32-bit words");
    int words[sizeof(T) / 4], kept[sizeof(T) / 4];
    __builtin_memcpy(words, &x, sizeof(T));
    __builtin_memcpy(kept, &otherwise, sizeof(T));
#pragma unroll
    for (size_t k = 0; k < sizeof(T) / 4; k++)
        words[k] = __builtin_amdgcn_update_dpp(kept[k], words[k], CONTROL, ROW_MASK, 0xf, false);
    __builtin_memcpy(&x, words, sizeof(T));
    return x;
}
template <typename T, typename Op, typename Otherwise>
__device__ __forceinline__ T wave_inclusive_scan_all_lanes(T x, Op op, Otherwise otherwise) {
    x = op(dpp_move_or<0x111, 0xf>(otherwise(x), x), x); // row_shr:1
    x = op(dpp_move_or<0x112, 0xf>(otherwise(x), x), x); // row_shr:2
    x = op(dpp_move_or<0x114, 0xf>(otherwise(x), x), x); // row_shr:4
    x = op(dpp_move_or<0x118, 0xf>(otherwise(x), x), x); // row_shr:8
    x = op(dpp_move_or<0x142, 0xa>(otherwise(x), x), x); // row_bcast:15 into rows 1 and 3
    x = op(dpp_move_or<0x143, 0xc>(otherwise(x), x), x); // row_bcast:31 into rows 2 and 3
    return x;
}

// The value of one lane, the same lane for the whole wave, in every lane.
template <typename T> __device__ __forceinline__ T read_lane(T x, int lane_of_all) {
    static_assert(sizeof(T) % 4 == 0, "made of 32-bit words");
    const int from = __builtin_amdgcn_readfirstlane(lane_of_all);
    int words[sizeof(T) / 4];
    __builtin_memcpy(words, &x, sizeof(T));
#pragma unroll
    for (size_t k = 0; k < sizeof(T) / 4; k++) words[k] = __builtin_amdgcn_readlane(words[k], from);
    __builtin_memcpy(&x, words, sizeof(T));
    return x;
}

struct PmcScan {
    float min_value, max_value;
    double sum;
};

struct ExponentRange {
    int low, high;
};

struct ValueRange {
    float low, high;
};

// What folding min_num / max_num over the wave's values in lane order gives (macaque_v.rs:199-204: the first operand is kept
// on ties, a NaN is never taken over a number; lanes that are not `active` count as NaN): the FIRST of the least values
// and the first of the greatest - or, where no lane has a number, whatever the last lane holds (a fold through nothing
// but NaNs ends on the last of them). Not a scan of two floats with the two functions by hand (some 75 vector
// instructions a batch in a kernel that is bound by them: k_fit_gap<SIZE>), but two unsigned reductions over keys in
// the values' order - both zeros one key, as they compare equal -, a ballot for the first lane that has the extreme
// and its value read from there.
__device__ __forceinline__ ValueRange wave_value_range(bool active, float stored) {
    const uint32_t bits = __float_as_uint(stored);
    const bool counts = active && stored == stored;
    const uint32_t plain = (bits << 1) == 0u ? 0u : bits;
    const uint32_t key = (plain & 0x80000000u) ? ~plain : (plain | 0x80000000u); // 0x007fffff (-inf) .. 0xff800000 (+inf)
    const uint32_t least = read_lane(wave_inclusive_scan_all_lanes(counts ? key : 0xffffffffu, [](uint32_t a, uint32_t b) { return min(a, b); },
                                                                   [](uint32_t x) { return x; }), MDB_WAVE - 1);
    const uint32_t most = read_lane(wave_inclusive_scan_all_lanes(counts ? key : 0u, [](uint32_t a, uint32_t b) { return max(a, b); },
                                                                  [](uint32_t x) { return x; }), MDB_WAVE - 1);
    const uint32_t mine = active ? bits : 0x7fc00000u;
    if (least == 0xffffffffu) { // (no lane has a number)
        const float last = __uint_as_float(read_lane(mine, MDB_WAVE - 1));
        return ValueRange{last, last};
    }
    const int first_least = __builtin_ctzll(__ballot(counts && key == least));
    const int first_most = __builtin_ctzll(__ballot(counts && key == most));
    return ValueRange{__uint_as_float(read_lane(bits, first_least)), __uint_as_float(read_lane(bits, first_most))};
}

// A chunk whose models turn out short is not this kernel's: a model costs at least one block of 64 points however
// short it is, a pass over rejected start points half a block per 64 start points it looks at and four for its
// second stage. Split mode takes about 100 cycles per point on such data, a block 2 300. Every `window_points` the
// wave looks at the steps it has taken since the last look: if the rest of the chunk at more than one step per
// `points_per_step` points would cost more than the whole chunk in split mode, it leaves the chunk to split mode
// (chunk_left[chunk] = 1, counted in *n_left; nothing the wave has written for the chunk is used then).
struct WaveLeave {
    unsigned int *chunk_left; // nullptr: never leave
    unsigned int *n_left; // [0] chunks left, [1] how many of them for their length alone
    uint32_t window_points;
    uint32_t points_per_step;
    // A chunk longer than this is left at once: one wave walks 65 536 points in a millisecond, a million (a whole
    // series handed over by the embedded API) in 15 to 30 - speculative pieces take 10 for it.
    uint32_t max_chunk_points;
    // MDB_FIT_DEBUG: models, rejected start points, passes over 64 start points, blocks of 64 points, scans of a
    // Swing block, models fitted by one lane, chunks left (nullptr: not counted).
    unsigned long long *counts;
    // The probe (probe_stride != 0): the launch's wave b takes chunk b * probe_stride, starts somewhere inside it, fits
    // one window's worth of points and says in n_left whether the pace there is one to leave the chunk at ([0]) and that
    // it has looked ([1]); what it writes besides is written again by the launch that fits the chunks.
    uint32_t probe_stride;
};

enum WaveCount { WAVE_MODELS, WAVE_REJECTED, WAVE_START_PASSES, WAVE_BLOCKS, WAVE_SWING_SCANS, WAVE_BY_ONE_LANE,
                 WAVE_QUIET_BLOCKS, WAVE_PMC_BLOCKS,
#ifdef MDB_WAVE_TIMING // (a build of its own, never the product's: shader clock cycles by region, summed over the waves)
                 WAVE_T_FIRST_STAGE, WAVE_T_SECOND_STAGE, WAVE_T_BLOCK_LOAD, WAVE_T_PMC, WAVE_T_SWING, WAVE_T_FINISH,
                 WAVE_T_FLUSH, WAVE_T_TOTAL, WAVE_N_FIRST_STAGE,
#endif
                 WAVE_COUNTS };
#ifdef MDB_WAVE_TIMING
#define WAVE_TIMED_BEGIN() const uint64_t timed_from = __builtin_amdgcn_s_memtime()
#define WAVE_TIMED_END(REGION) timed[(REGION) - WAVE_T_FIRST_STAGE] += __builtin_amdgcn_s_memtime() - timed_from
#else
#define WAVE_TIMED_BEGIN() do {} while (0)
#define WAVE_TIMED_END(REGION) do {} while (0)
#endif

constexpr uint32_t WAVE_PASS_POINTS = 1024; // start points a pass over rejected start points looks at, at most
constexpr uint32_t WAVE_PASS_STEPS = 4;     // what its second stage costs, in blocks of a model

// HAS_TS: the timestamps are loaded (some chunk is irregular; all of them within +-2^52, so exact as f64).
// PIECES: one wave per PIECE of a chunk (split mode's speculation, SplitArgs, with a wave instead of a lane): the wave
// starts its greedy chain at the piece's first point, records what it fits in the per-point table instead of the
// chunk's record list and stops at the first start point past its piece that some wave has been at; k_fit_walk
// then collects the chunk's real chain. For calls of a handful of chunks (a server's finished buffer): a chunk's
// latency becomes a piece's plus the stretch the chains need to meet.
template <int KIND, bool HAS_TS = false, bool PIECES = false>
__global__ __launch_bounds__(MDB_WAVE) void k_fit_models_wave(FitArgs args, WaveLeave leave, SplitArgs split,
                                                             const unsigned long long *__restrict__ record_base,
                                                             ModelRec *__restrict__ records,
                                                             ChunkPlan *__restrict__ plans,
                                                             unsigned int *__restrict__ error) {
    __shared__ PendingSwing pending[MDB_WAVE];
    __shared__ uint32_t survivors[2 * MDB_WAVE];
    const int lane = threadIdx.x;
    uint64_t chunk = blockIdx.x;
    const bool probe = !PIECES && leave.probe_stride != 0u;
    if (probe) {
        chunk = (uint64_t)blockIdx.x * leave.probe_stride;
        if (chunk >= args.n_chunks) return;
    }
    uint32_t first_point = 0;
    if (PIECES) {
        // The chunk of this piece: the last c with piece_base[c] <= the piece's number (the same for all lanes).
        const uint64_t unit = blockIdx.x;
        if (unit >= split.n_pieces) return;
        uint64_t lo = 0, hi = args.n_chunks;
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) / 2;
            if (split.piece_base[mid] <= unit) lo = mid;
            else hi = mid;
        }
        chunk = lo;
        first_point = (uint32_t)(unit - split.piece_base[chunk]) * split.piece_points;
    }
    const uint64_t base = args.chunk_offsets[chunk];
    const uint64_t length64 = args.chunk_offsets[chunk + 1] - base;
    if (length64 > COUNT_MASK - ENTRY_END_BIAS) { // (pieces: such a chunk has none, PieceCount)
        if (lane == 0) {
            atomicOr(error, ERR_TOO_LONG);
            plans[chunk] = {0, 0};
        }
        return;
    }
    const uint32_t n = (uint32_t)length64;
    if (n == 0) {
        if (lane == 0 && !PIECES) plans[chunk] = {0, 0};
        return;
    }
    const uint32_t piece_end = PIECES ? first_point + split.piece_points : 0u;
    // (pieces) Has some wave been at start point `at` already? Asked once the wave is past its own piece.
    auto visited = [&](uint32_t at) -> bool {
        return PIECES && at >= piece_end && at < n &&
               __hip_atomic_load(&split.entry[base + at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    };
    // (pieces) Start points [from, from + count) are rejected: no model stands on them.
    auto mark_rejected = [&](uint32_t from, uint32_t count) {
        for (uint32_t k = lane; k < count; k += MDB_WAVE)
            __hip_atomic_store(&split.entry[base + from + k], ENTRY_REJECTED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    const float *__restrict__ values = args.values + base;
    const ChunkTimestamps regular_ts = chunk_timestamps(args.timestamps, chunk, base);
    const double first_time = (double)regular_ts.first, interval = (double)regular_ts.interval; // exact
    const int64_t *__restrict__ chunk_ts = HAS_TS ? args.timestamps.ts + base : nullptr;
    // (f64)timestamp of point j: computed (the product and the sum are integers below 2^53) or loaded, exact
    auto time_at = [&](uint32_t j) -> double {
        return HAS_TS ? (double)chunk_ts[j] : __builtin_fma((double)j, interval, first_time);
    };
    const mdb_error_bound eb = args.eb;
    const DeviationFactor dev = deviation_factor(eb);
    const PmcFast pmc_fast = pmc_fast_constants(eb);
    ModelRec *__restrict__ out = records + record_base[chunk];
    const double nan = __longlong_as_double(0x7ff8000000000000ll);

    uint32_t n_models = 0, n_pending = 0;
    GapCounter gaps;
    uint32_t current = first_point;
    bool after_rejection = false;
    uint32_t half_steps = 0, next_look = leave.window_points, looked_at = 0; // (WaveLeave)
    uint32_t counted[WAVE_COUNTS] = {};
    uint32_t rejections_in_a_row = 0;
#ifdef MDB_WAVE_TIMING
    uint64_t timed[WAVE_COUNTS - WAVE_T_FIRST_STAGE] = {};
    const uint64_t timed_start = __builtin_amdgcn_s_memtime();
#endif
    auto report_counts = [&]() {
        if (leave.counts && lane == 0) {
#ifdef MDB_WAVE_TIMING
            timed[WAVE_T_TOTAL - WAVE_T_FIRST_STAGE] = __builtin_amdgcn_s_memtime() - timed_start;
            for (int k = 0; k < WAVE_T_FIRST_STAGE; k++) atomicAdd(leave.counts + k, (unsigned long long)counted[k]);
            for (int k = WAVE_T_FIRST_STAGE; k < WAVE_COUNTS; k++) atomicAdd(leave.counts + k, (unsigned long long)timed[k - WAVE_T_FIRST_STAGE]);
#else
            for (int k = 0; k < WAVE_COUNTS; k++) atomicAdd(leave.counts + k, (unsigned long long)counted[k]);
#endif
        }
    };
    if (probe) { // (a window somewhere in the chunk, not the same place in every chunk)
        if (n < 2u * leave.window_points) return;
        current = (uint32_t)(((chunk * 2654435761ull) >> 7) % (n / leave.window_points - 1u)) * leave.window_points;
        looked_at = current;
        next_look = current + leave.window_points;
    } else if (!PIECES && leave.chunk_left && n > leave.max_chunk_points) {
        if (lane == 0) {
            leave.chunk_left[chunk] = 1u;
            atomicAdd(leave.n_left, 1u);
            atomicAdd(leave.n_left + 1, 1u); // (... for its length, whatever its models are like)
        }
        return;
    }
    if (!probe && !PIECES && leave.chunk_left && lane == 0) leave.chunk_left[chunk] = 0u;

    // The sums of the queued Swing models, one lane per model, and with them the models' last values.
    auto flush_pending = [&]() {
        WAVE_TIMED_BEGIN();
        if (lane < (int)n_pending) {
            const PendingSwing item = pending[lane];
            const float *__restrict__ v = values + item.start;
            const double first_value = (double)v[0];
            const double start_time = time_at(item.start);
            double numerator = 0.0, denominator = 0.0;
            auto add_point = [&](uint32_t r, float v32) {
                const double value = (double)v32;
                if (first_value != value) { // (adding 0.0 otherwise: swing.rs:222-227, no change)
                    // (f64)(t - start_time), exact
                    const double dt = HAS_TS ? time_at(item.start + r) - start_time : (double)r * interval;
                    numerator += (value - first_value) * dt;
                    denominator += dt * dt;
                }
            };
            uint32_t r = 2;
            for (; r + 8 <= item.length; r += 8) { // (eight loads in flight, then the two chains of additions)
                float group[8];
#pragma unroll
                for (int k = 0; k < 8; k++) group[k] = v[r + k];
#pragma unroll
                for (int k = 0; k < 8; k++) add_point(r + k, group[k]);
            }
            for (; r < item.length; r++) add_point(r, v[r]);
            const double projected = numerator / denominator; // swing.rs:246-259
            const double slope = max_num(item.lower_slope, min_num(projected, item.upper_slope));
            const double span = HAS_TS ? time_at(item.start + item.length - 1) - start_time
                                       : (double)(item.length - 1) * interval;
            const double last_value = slope * span + first_value;
            if (PIECES) split.p1[base + item.start] = (float)last_value;
            else out[item.record].p1 = (float)last_value;
        }
        n_pending = 0;
        WAVE_TIMED_END(WAVE_T_FLUSH);
    };

    // One model from `current` by the plain fitters, lane 0 alone (non-finite values, sums that may round). Writes the
    // record (or the table's parameters); true: accepted.
    auto fit_by_one_lane = [&](ModelRec &rec) -> bool {
            counted[WAVE_BY_ONE_LANE] += 1;
            int accepted_flag = 0;
            if (lane == 0) {
                PmcDev pmc;
                SwingFast swing;
                pmc.reset();
                swing.reset();
                bool pmc_fits = true, swing_fits = true;
                for (uint32_t j = current; j < n && (pmc_fits || swing_fits); j++) {
                    const float v = values[j];
                    const double t = time_at(j);
                    if (pmc_fits) pmc_fits = pmc.fit(eb, v);
                    if (swing_fits) swing_fits = swing.fit(dev, t, v);
                }
                const float pmc_bpv = (float)MDB_COMPRESSED_METADATA_SIZE_IN_BYTES / (float)pmc.length;
                const float swing_bpv = ((float)MDB_COMPRESSED_METADATA_SIZE_IN_BYTES + 1.0f) / (float)swing.length;
                const bool choose_pmc = pmc_bpv <= swing_bpv;
                if ((choose_pmc ? pmc_bpv : swing_bpv) <= (float)MDB_VALUE_SIZE_IN_BYTES) {
                    accepted_flag = 1;
                    if (choose_pmc) {
                        rec.start_and_type = current;
                        rec.end = current + pmc.length - 1;
                        rec.p0 = (float)(pmc.sum / (double)pmc.length);
                        rec.p1 = rec.p0;
                    } else {
                        rec.start_and_type = current | 0x80000000u;
                        rec.end = current + swing.length - 1;
                        if (HAS_TS) { // SwingFast::model with the span from the timestamps (swing.rs:246-259)
                            const double projected = swing.numerator / swing.denominator;
                            const double slope = max_num(swing.lower.slope, min_num(projected, swing.upper.slope));
                            const double span = time_at(rec.end) - swing.start_time;
                            rec.p0 = (float)swing.first_value;
                            rec.p1 = (float)(slope * span + swing.first_value);
                        } else {
                            swing.model(interval, &rec.p0, &rec.p1);
                        }
                    }
                }
            }
            const bool accepted_model = __shfl(accepted_flag, 0) != 0;
            rec.start_and_type = __shfl(rec.start_and_type, 0);
            rec.end = __shfl(rec.end, 0);
            if (accepted_model && lane == 0) {
                if (PIECES) {
                    split.p0[base + current] = rec.p0;
                    split.p1[base + current] = rec.p1;
                } else {
                    out[n_models] = rec;
                }
            }
            return accepted_model;
    };

    // Does a model stand on start point `start` (PMC-Mean or Swing gets to 8 points: ModelBuilder::finish accepts it,
    // types.rs:84-101 with compression.rs:238)? The two fitters step by step, this lane alone; the eight values are
    // asked for up front (a model of fewer than 8 points is never accepted, so a start within 7 points of the end falls).
    auto stands_by_steps = [&](uint32_t start) -> bool {
        if (start + 7 >= n) return false;
        float window[8];
#pragma unroll
        for (int k = 0; k < 8; k++) window[k] = values[start + k];
        PmcDev pmc;
        SwingFast swing;
        pmc.reset();
        swing.reset();
        bool pmc_fits = true, swing_fits = true;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (pmc_fits || swing_fits) {
                const double t = time_at(start + k);
                if (pmc_fits) pmc_fits = pmc_fit_fast(pmc, pmc_fast, eb, window[k]);
                if (swing_fits) swing_fits = swing.fit(dev, t, window[k]);
            }
        }
        return pmc_fits || swing_fits;
    };

    // (for the margins within which a third point is near a line; an interval of 0: infinite - every margin is)
    const double unit = 0x1p-53;
    const double per_interval = HAS_TS ? 0.0 : 1.0 / interval;

    // ---- a lossless bound: both fitters decide by equality, and nothing of them moves ------------------------------
    // PMC-Mean accepts a value iff it equals the values before it (minimum and maximum must both equal the average,
    // pmc_mean.rs:58-75; the f64 sum of up to 2^29 equal f32 values is exact, so the average IS the value). Swing's
    // two bounds are one line - the one through the model's first two points, swing.rs:126-143 with a deviation of 0 -
    // which accepts a point iff slope * t + intercept == value and never moves (swing.rs:144-197: neither bound is
    // ever beside the value). So a model stands on a start point iff its first 8 values are equal (PMC-Mean; Swing
    // then draws the line of slope 0 through them, accepts exactly the same points and loses the tie, types.rs:84-101)
    // or lie on that line (Swing; PMC-Mean has ended at the second point), and it runs as far as that holds. 64 start
    // points are decided per round, one lane each; a round in which no lane's third point is anywhere near its line
    // does not divide. Windows with a NaN or an infinity in them (runs of which both fitters accept) go the long way.
    if (KIND == MDB_EB_LOSSLESS) {
        while (current < n) {
            if (visited(current)) break; // (pieces) the chain from here on is in the table
            counted[WAVE_START_PASSES] += 1;
            const uint32_t start = current + (uint32_t)lane;
            bool stands = false, by_pmc = false, long_way = false;
            double slope = 0.0, intercept = 0.0;
            float window[8] = {};
            if (start + 7 < n) {
#pragma unroll
                for (int k = 0; k < 8; k++) window[k] = values[start + k];
                uint32_t largest = 0;
                bool all_equal = true;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    largest = max(largest, __float_as_uint(window[k]) & 0x7fffffffu);
                    all_equal = all_equal && window[k] == window[0];
                }
                if (largest >= 0x7f800000u) {
                    long_way = true;
                } else if (window[0] == window[1]) {
                    by_pmc = all_equal;
                    stands = all_equal;
                } else {
                    // Is the third point within the rounding of the reference's arithmetic of the line through the first
                    // two? |slope * t2 + (v0 - slope * t0) - (v0 + (v1 - v0) k)| <= 2^-53 (4 |slope| (|t0| + |t2|) + |v0| +
                    // |v2|) (1 + ...), k = (t2 - t0) / (t1 - t0): four times that, and this test's own roundings, are
                    // covered; with loaded timestamps k comes from a 32-bit reciprocal, which costs another 2^-20 of it.
                    const double v0 = (double)window[0], v1 = (double)window[1], v2 = (double)window[2];
                    const double t0 = time_at(start);
                    const double t1 = HAS_TS ? time_at(start + 1) : t0 + interval; // (exact, as every timestamp here)
                    const double t2 = HAS_TS ? time_at(start + 2) : t1 + interval;
                    const double rise = v1 - v0;
                    double k = 2.0, per_step = per_interval, slack = 0.0;
                    if (HAS_TS) {
                        const float reciprocal = __builtin_amdgcn_rcpf((float)(t1 - t0));
                        k = (double)((float)(t2 - t0) * reciprocal);
                        per_step = (double)reciprocal;
                        slack = fabs(rise) * k * 0x1p-19;
                    }
                    const double expected = v0 + rise * k;
                    const double margin = 16.0 * unit * (fabs(rise) * ((fabs(t0) + fabs(t2)) * per_step) * 1.001 + fabs(v0) + fabs(v1) + fabs(v2)) + slack;
                    if (!(fabs(expected - v2) > margin)) { // (also when the margin is not a number)
                        slope = rise / (t1 - t0); // line_through_exact (swing.rs:323-340)
                        intercept = v0 - slope * t0;
                        // (the reference's own two comparisons, swing.rs:161-166 with a deviation of 0: a point is
                        // rejected when the line's value is below or above it - which a line that is not a number, two
                        // points with ONE timestamp, never is)
                        auto off_the_line = [](double approximation, double value) { return approximation < value || approximation > value; };
                        bool on_the_line = !off_the_line(slope * t2 + intercept, v2);
#pragma unroll
                        for (int j = 3; j < 8; j++) {
                            const double t = HAS_TS ? time_at(start + j) : __builtin_fma((double)(start + j), interval, first_time);
                            on_the_line = on_the_line && !off_the_line(slope * t + intercept, (double)window[j]);
                        }
                        stands = on_the_line;
                    }
                }
            }
            if (__ballot(long_way)) {
                if (long_way) stands = stands_by_steps(start);
            }
            const unsigned long long standing = __ballot(stands);
            if (!standing) { // nothing stands on these start points: they are residuals (compression.rs:258-262)
                const uint32_t covered = min((uint32_t)MDB_WAVE, n - current);
                if (PIECES) mark_rejected(current, covered);
                current += covered;
                counted[WAVE_REJECTED] += covered;
                continue;
            }
            const int first = __builtin_ctzll(standing);
            if (first > 0) {
                if (PIECES) mark_rejected(current, (uint32_t)first);
                current += (uint32_t)first;
                counted[WAVE_REJECTED] += (uint32_t)first;
            }
            ModelRec rec{};
            if (read_lane((int)long_way, first) != 0) {
                if (!fit_by_one_lane(rec)) { // (it stands, so this is not expected: the start point is a residual then)
                    if (PIECES) mark_rejected(current, 1);
                    current += 1;
                    counted[WAVE_REJECTED] += 1;
                    continue;
                }
            } else {
                const bool pmc_model = read_lane((int)by_pmc, first) != 0;
                const float first_value32 = read_lane(window[0], first);
                const double model_slope = read_lane(slope, first), model_intercept = read_lane(intercept, first);
                uint32_t length = 8;
                bool alive = true;
                for (uint32_t position = current + 8; alive && position < n; position += MDB_WAVE) {
                    counted[WAVE_BLOCKS] += 1;
                    const uint32_t index = position + (uint32_t)lane;
                    const bool valid = index < n;
                    const float v = valid ? values[index] : 0.0f;
                    bool differs; // (a NaN or an infinity differs from everything finite, as in both fitters)
                    if (pmc_model) {
                        differs = v != first_value32;
                    } else { // (swing.rs:152-166: not finite, or the line's value below or above it)
                        const double approximation = model_slope * time_at(valid ? index : position) + model_intercept;
                        differs = !isfinite(v) || approximation < (double)v || approximation > (double)v;
                    }
                    const unsigned long long stops = __ballot(valid && differs);
                    const uint32_t n_valid = min((uint32_t)MDB_WAVE, n - position);
                    length += stops ? (uint32_t)__builtin_ctzll(stops) : n_valid;
                    alive = !stops;
                }
                rec.end = current + length - 1;
                if (pmc_model) {
                    // (the sum starts at +0.0: of zeros of either sign it is +0.0, of anything else length * value)
                    rec.start_and_type = current;
                    rec.p0 = first_value32 == 0.0f ? 0.0f : first_value32;
                    rec.p1 = rec.p0;
                } else {
                    // SwingFast::model (swing.rs:246-259): the slope between two bounds that are the same line
                    const double start_time = time_at(current);
                    const double span = HAS_TS ? time_at(rec.end) - start_time : (double)(length - 1) * interval;
                    rec.start_and_type = current | 0x80000000u;
                    rec.p0 = first_value32;
                    rec.p1 = (float)(model_slope * span + (double)first_value32);
                }
                if (lane == 0) {
                    if (PIECES) {
                        split.p0[base + current] = rec.p0;
                        split.p1[base + current] = rec.p1;
                    } else {
                        out[n_models] = rec;
                    }
                }
            }
            if (PIECES) {
                if (lane == 0)
                    __hip_atomic_store(&split.entry[base + current], (rec.end + ENTRY_END_BIAS) | (rec.start_and_type & 0x80000000u),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                gaps.on_model(current, rec.end);
            }
            n_models += 1;
            current = rec.end + 1;
        }
        if (lane == 0 && !PIECES) plans[chunk] = {n_models, gaps.finish(n)};
        counted[WAVE_MODELS] = n_models;
        report_counts();
        return;
    }

    while (current < n) {
        if (visited(current)) break; // (pieces) the chain from here on is in the table
        if (probe && current >= next_look) { // (the pace of this window, judged as a chunk's first one is)
            if (lane == 0) {
                if ((uint64_t)half_steps * leave.points_per_step > 2ull * (current - looked_at)) atomicAdd(leave.n_left, 1u);
                atomicAdd(leave.n_left + 1, 1u);
                // (the model that ran out of the window went on for another window's length or more: long ones)
                if (current - looked_at >= 2u * leave.window_points) atomicAdd(leave.n_left + 2, 1u);
            }
            return;
        }
        if (!probe && !PIECES && leave.chunk_left && current >= next_look) {
            // (what is left of the chunk at the pace of this window against the whole chunk in split mode)
            if ((uint64_t)half_steps * leave.points_per_step * (n - current) > 2ull * (current - looked_at) * n) {
                if (lane == 0) {
                    leave.chunk_left[chunk] = 1u;
                    atomicAdd(leave.n_left, 1u);
                }
                report_counts();
                return;
            }
            half_steps = 0;
            looked_at = current;
            next_look = current + leave.window_points;
        }
        if (after_rejection) {
            counted[WAVE_START_PASSES] += 1;
            // Start points after a rejected one, one lane each. Most rejected start points are rejected early:
            // PMC-Mean by the second or third point, Swing by the third (the first two only draw its bounds). So
            // the next start points are first tried on three points, 64 at a time, and those that survive - up to
            // 64 of them, from up to 1 024 start points - are collected ...
            uint32_t n_survivors = 0, scanned = 0;
            WAVE_TIMED_BEGIN();
            while (n_survivors < (uint32_t)MDB_WAVE && scanned < WAVE_PASS_POINTS && current + scanned < n) {
                const uint32_t start = current + scanned + lane;
                // (a start within 7 points of the end never stands: nothing of fewer than 8 points is accepted)
                bool early = start < n && start + 7 >= n;
                if (start + 7 < n) {
                    const float v0 = values[start], v1 = values[start + 1], v2 = values[start + 2];
                    if (isfinite(v0) && isfinite(v1) && isfinite(v2)) {
                        PmcDev pmc;
                        pmc.reset();
                        const bool pmc_fits = pmc_fit_fast(pmc, pmc_fast, eb, v0) && pmc_fit_fast(pmc, pmc_fast, eb, v1) &&
                                              pmc_fit_fast(pmc, pmc_fast, eb, v2);
                        // Swing (swing.rs:126-143 for the second point, :144-160 for the third) is CERTAINLY out at the
                        // third point when the value lies beyond both bounds by more than the reference's own arithmetic
                        // can be off: its bounds at t2 are within 2^-53 (4 |slope| (|t0| + |t2|) + |v0| + |bound|) of
                        // v0 + (v1 +- deviation1 - v0) k, k = (t2 - t0) / (t1 - t0) - no division here, the survivors get
                        // the fitters themselves. (Loaded timestamps: k from a 32-bit reciprocal, another 2^-20 of it.)
                        const double t0 = time_at(start);
                        const double t2 = HAS_TS ? time_at(start + 2) : t0 + 2.0 * interval; // (exact, as every timestamp here)
                        const double w0 = (double)v0, w1 = (double)v1, w2 = (double)v2;
                        const double deviation1 = dev.of(w1), deviation2 = dev.of(w2);
                        const double rise_above = (w1 + deviation1) - w0, rise_below = (w1 - deviation1) - w0;
                        double k = 2.0, per_step = per_interval, slack = 0.0;
                        if (HAS_TS) {
                            const float reciprocal = __builtin_amdgcn_rcpf((float)(time_at(start + 1) - t0));
                            k = (double)((float)(t2 - t0) * reciprocal);
                            per_step = (double)reciprocal;
                            slack = (fabs(rise_above) + fabs(rise_below)) * k * 0x1p-19;
                        }
                        const double margin = 16.0 * unit * ((fabs(rise_above) + fabs(rise_below)) * ((fabs(t0) + fabs(t2)) * per_step) * 1.001 +
                                                             fabs(w0) + fabs(w1) + fabs(w2) + deviation1 + deviation2) + slack;
                        const bool swing_out = (w0 + rise_above * k) + deviation2 < w2 - margin ||
                                               (w0 + rise_below * k) - deviation2 > w2 + margin; // (false when anything is not a number)
                        early = !pmc_fits && swing_out;
                    }
                }
                const bool survives = start < n && !early;
                const unsigned long long survivors_here = __ballot(survives);
                if (survives)
                    survivors[n_survivors + (uint32_t)__popcll(survivors_here & ((1ull << lane) - 1ull))] = scanned + lane;
                n_survivors += (uint32_t)__popcll(survivors_here);
                scanned += MDB_WAVE;
                half_steps += 1; // (a look at 64 start points costs about half of what a block of a model costs)
#ifdef MDB_WAVE_TIMING
                timed[WAVE_N_FIRST_STAGE - WAVE_T_FIRST_STAGE] += 1;
#endif
            }
            WAVE_TIMED_END(WAVE_T_FIRST_STAGE);
#ifdef MDB_WAVE_TIMING
            const uint64_t second_from = __builtin_amdgcn_s_memtime();
#endif
            uint32_t covered = min(scanned, n - current);
            if (n_survivors == 0) { // nobody: all of them are residuals (compression.rs:258-262)
                if (PIECES) mark_rejected(current, covered);
                current += covered;
                counted[WAVE_REJECTED] += covered;
                continue;
            }
            half_steps += 2 * WAVE_PASS_STEPS;
            __syncthreads();
            // ... and taken through the 8 points that decide it (neither fitter gets to 8: rejected), one lane each.
            // What lies behind the 64th survivor is the next pass's.
            if (n_survivors > (uint32_t)MDB_WAVE) covered = survivors[MDB_WAVE];
            bool stands = false;
            uint32_t offset = 0;
            if ((uint32_t)lane < n_survivors) { // (nobody: the branch is not taken)
                offset = survivors[lane];
                stands = stands_by_steps(current + offset);
            }
            const unsigned long long standing = __ballot(stands);
            __syncthreads();
            if (standing) { // the first start point a model stands on: everything in front of it is rejected
                const uint32_t skipped = read_lane(offset, __builtin_ctzll(standing));
                if (PIECES) mark_rejected(current, skipped);
                current += skipped;
                counted[WAVE_REJECTED] += skipped;
                after_rejection = false;
            } else {
                if (PIECES) mark_rejected(current, covered);
                current += covered;
                counted[WAVE_REJECTED] += covered;
            }
#ifdef MDB_WAVE_TIMING
            timed[WAVE_T_SECOND_STAGE - WAVE_T_FIRST_STAGE] += __builtin_amdgcn_s_memtime() - second_from;
#endif
            continue;
        }

        // ---- one model from `current` ----
        uint32_t pmc_length = 0, swing_length = 0;
        double pmc_sum = 0.0;
        float pmc_min = __uint_as_float(0x7fc00000u), pmc_max = __uint_as_float(0x7fc00000u);
        int exponent_low = 255, exponent_high = 0; // of the non-zero values PMC-Mean has been offered
        bool pmc_alive = true, swing_alive = true, by_one_lane = false;
        double first_value = nan;
        const double start_time = time_at(current);
        double upper_slope = nan, lower_slope = nan;
        uint32_t position = current;
        while ((pmc_alive || swing_alive) && position < n) {
            if (probe && position >= next_look + leave.window_points) {
                // (the probe: this model has gone on for a window's length past the window - a long one; no need to see its end)
                if (lane == 0) {
                    atomicAdd(leave.n_left + 1, 1u);
                    atomicAdd(leave.n_left + 2, 1u);
                }
                return;
            }
            counted[WAVE_BLOCKS] += 1;
            half_steps += 2;
            const uint32_t index = position + lane;
            const int n_valid = (int)min((uint32_t)MDB_WAVE, n - position);
            const bool valid = lane < n_valid;
#ifdef MDB_WAVE_TIMING
            const uint64_t load_from = __builtin_amdgcn_s_memtime();
#endif
            const float v = valid ? values[index] : 0.0f;
            const bool not_finite = __ballot(valid && !isfinite(v)) != 0;
#ifdef MDB_WAVE_TIMING
            timed[WAVE_T_BLOCK_LOAD - WAVE_T_FIRST_STAGE] += __builtin_amdgcn_s_memtime() - load_from;
#endif
            if (not_finite) {
                by_one_lane = true;
                break;
            }
            const double value = (double)v;
            if (position == current) first_value = read_lane(value, 0);

            if (pmc_alive) {
                WAVE_TIMED_BEGIN();
                counted[WAVE_PMC_BLOCKS] += 1;
                // Up to which lane is every partial sum exact (file comment)? The exponents met so far, per lane.
                const uint32_t bits = __float_as_uint(v);
                const bool non_zero = valid && (bits << 1) != 0u;
                const int exponent = max((int)((bits >> 23) & 0xffu), 1);
                ExponentRange range = {non_zero ? exponent : 255, non_zero ? exponent : 0};
                range = wave_inclusive_scan_all_lanes(
                    range, [](ExponentRange a, ExponentRange b) { return ExponentRange{min(a.low, b.low), max(a.high, b.high)}; },
                    [](ExponentRange x) { return x; });
                range.low = min(range.low, exponent_low);
                range.high = max(range.high, exponent_high);
                const uint32_t next_length = pmc_length + (uint32_t)lane + 1u;
                const int length_bits = 32 - __clz((int)next_length);
                const bool exact_to_here = range.high < range.low || range.high - range.low + 24 + length_bits + 1 <= 53;
                // (the block's values are finite: minimum and maximum without min_num's care for NaN, and the sign of a
                // zero among them is nothing within_error_bound can see)
                const PmcScan scan = wave_inclusive_scan_all_lanes(
                    PmcScan{v, v, value},
                    [](PmcScan a, PmcScan b) { return PmcScan{fminf(a.min_value, b.min_value), fmaxf(a.max_value, b.max_value), a.sum + b.sum}; },
                    [](PmcScan x) { return PmcScan{x.min_value, x.max_value, 0.0}; });
                const float next_min = min_num(pmc_min, scan.min_value);
                const float next_max = max_num(pmc_max, scan.max_value);
                // The reference's sum at this lane is (sum of everything in front) + value: the former in any order
                // while it is exact, the latter one addition, rounded as the reference rounds it.
                double in_front = dpp_move<0x138>(scan.sum); // wave_shr:1
                bool exact_in_front = dpp_move<0x138>((int)exact_to_here) != 0;
                if (lane == 0) {
                    in_front = 0.0;
                    exact_in_front = true; // (the model's state is, or the model would be lane 0's already)
                }
                const double next_sum = (pmc_sum + in_front) + value;
                const float average = (float)(next_sum / (double)next_length);
                const bool within = within_error_bound(eb, next_min, average) && within_error_bound(eb, next_max, average);
                const unsigned long long known = __ballot(valid && exact_in_front); // lanes 0..e, e the first inexact one
                const unsigned long long failing = __ballot(valid && !within) & known;
                const int reach = ~known ? __builtin_ctzll(~known) : MDB_WAVE;     // (lane 0 is always known)
                const int accepted = failing ? __builtin_ctzll(failing) : reach;
                if (!failing && reach < n_valid) { // a sum has rounded and the model goes on: one lane, in order
                    by_one_lane = true;
                    break;
                }
                WAVE_TIMED_END(WAVE_T_PMC);
                if (accepted > 0) {
                    pmc_min = read_lane(next_min, accepted - 1);
                    pmc_max = read_lane(next_max, accepted - 1);
                    pmc_sum = read_lane(next_sum, accepted - 1);
                    const ExponentRange met = read_lane(range, accepted - 1);
                    exponent_low = met.low;
                    exponent_high = met.high;
                }
                pmc_length += (uint32_t)accepted;
                if (failing) pmc_alive = false;
            }

            if (swing_alive) {
                WAVE_TIMED_BEGIN();
                const double t = valid ? time_at(index) : 0.0;
                const double deviation = lean_deviation<KIND>(dev.factor, value);
                // Every bound is a line through the model's first point, so its slope says it all: the intercept of
                // line_through_exact is first_value - slope * start_time (its special case, slope 0 through equal
                // values, gives the same up to the sign of a zero, which no comparison sees).
                const double upper_candidate = line_through_exact(start_time, first_value, t, value + deviation).slope;
                const double lower_candidate = line_through_exact(start_time, first_value, t, value - deviation).slope;
                int first_lane = 0;
                if (position == current) { // the model's first two points are accepted as they come (swing.rs:107-143)
                    swing_length = (uint32_t)min(n_valid, 2);
                    first_lane = 2;
                    if (n_valid >= 2) {
                        upper_slope = read_lane(upper_candidate, 1);
                        lower_slope = read_lane(lower_candidate, 1);
                    }
                }
                while (swing_alive && first_lane < n_valid) {
                    counted[WAVE_SWING_SCANS] += 1;
                    // The bounds this lane would meet if every lane in front of it moved them as lines of smaller /
                    // larger slope do: the extreme of the state's slope and the candidates of [first_lane, lane).
                    double above = dpp_move<0x138>(upper_candidate); // wave_shr:1
                    double below = dpp_move<0x138>(lower_candidate);
                    if (lane <= first_lane) {
                        above = upper_slope;
                        below = lower_slope;
                    }
                    // (slopes are finite here; which of two equal ones stays does not matter: they are the same line)
                    above = wave_inclusive_scan_all_lanes(above, [](double a, double b) { return fmin(a, b); }, [](double x) { return x; });
                    below = wave_inclusive_scan_all_lanes(below, [](double a, double b) { return fmax(a, b); }, [](double x) { return x; });
                    // The step itself (swing.rs:144-197) against that state.
                    const double upper_approximation = above * t + (first_value - above * start_time);
                    const double lower_approximation = below * t + (first_value - below * start_time);
                    const bool fails = upper_approximation + deviation < value || lower_approximation - deviation > value;
                    const bool lowers_upper = upper_approximation - deviation > value;
                    const bool raises_lower = lower_approximation + deviation < value;
                    const bool as_assumed = lowers_upper == (upper_candidate < above) && raises_lower == (lower_candidate > below);
                    const bool mine = lane >= first_lane && lane < n_valid;
                    if (leave.counts && first_lane == 0 && !__ballot(mine && (lowers_upper || raises_lower || fails)))
                        counted[WAVE_QUIET_BLOCKS] += 1; // (a whole block in which neither bound moves)
                    const unsigned long long stops = __ballot(mine && (fails || !as_assumed));
                    const int at = stops ? __builtin_ctzll(stops) : n_valid - 1; // its incoming state is the true one
                    const bool ends = stops && ((__ballot(fails) >> at) & 1ull) != 0;
                    // The state behind lane `at` (in front of it when the model ends there).
                    upper_slope = read_lane((lowers_upper && !ends) ? upper_candidate : above, at);
                    lower_slope = read_lane((raises_lower && !ends) ? lower_candidate : below, at);
                    if (ends) {
                        swing_length += (uint32_t)(at - first_lane);
                        swing_alive = false;
                    } else {
                        swing_length += (uint32_t)(at + 1 - first_lane);
                        first_lane = at + 1;
                    }
                }
                WAVE_TIMED_END(WAVE_T_SWING);
            }
            position += MDB_WAVE;
        }

#ifdef MDB_WAVE_TIMING
        const uint64_t finish_from = __builtin_amdgcn_s_memtime();
#endif
        bool accepted_model = false;
        ModelRec rec{};
        if (by_one_lane) {
            accepted_model = fit_by_one_lane(rec);
        } else {
            // ModelBuilder::finish (types.rs:84-101): fewest bytes per value, PMC-Mean wins ties.
            const float pmc_bpv = (float)MDB_COMPRESSED_METADATA_SIZE_IN_BYTES / (float)pmc_length;
            const float swing_bpv = ((float)MDB_COMPRESSED_METADATA_SIZE_IN_BYTES + 1.0f) / (float)swing_length;
            const bool choose_pmc = pmc_bpv <= swing_bpv;
            accepted_model = (choose_pmc ? pmc_bpv : swing_bpv) <= (float)MDB_VALUE_SIZE_IN_BYTES; // compression.rs:238
            if (accepted_model) {
                if (choose_pmc) {
                    rec.start_and_type = current;
                    rec.end = current + pmc_length - 1;
                    rec.p0 = (float)(pmc_sum / (double)pmc_length); // pmc_mean.rs:91-93
                    rec.p1 = rec.p0;
                } else {
                    rec.start_and_type = current | 0x80000000u;
                    rec.end = current + swing_length - 1;
                    rec.p0 = (float)first_value;
                    if (lane == 0) pending[n_pending] = {n_models, current, swing_length, lower_slope, upper_slope};
                    n_pending += 1;
                }
                if (lane == 0 && PIECES) {
                    split.p0[base + current] = rec.p0;
                    if (choose_pmc) split.p1[base + current] = rec.p1;
                } else if (lane == 0) {
                    // (p1 of a queued Swing model is flush_pending's to write: some lane's, some time later)
                    out[n_models].start_and_type = rec.start_and_type;
                    out[n_models].end = rec.end;
                    out[n_models].p0 = rec.p0;
                    if (choose_pmc) out[n_models].p1 = rec.p1;
                }
            }
        }
        if (accepted_model) {
            rejections_in_a_row = 0;
            if (PIECES) {
                // (the entry says "a wave has been here" and where its model ends; p0 / p1 are read by k_fit_walk,
                // a kernel later)
                if (lane == 0)
                    __hip_atomic_store(&split.entry[base + current], (rec.end + ENTRY_END_BIAS) | (rec.start_and_type & 0x80000000u),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                gaps.on_model(current, rec.end);
            }
            n_models += 1;
            current = rec.end + 1;
            if (n_pending == (uint32_t)MDB_WAVE) {
                __syncthreads();
                flush_pending();
                __syncthreads();
            }
        } else {
            if (PIECES) mark_rejected(current, 1);
            current += 1; // the point becomes a residual (compression.rs:258-262)
            counted[WAVE_REJECTED] += 1;
            // (a lone rejected point between two models is cheaper found by the next model's step than by a pass
            // over 64 start points)
            rejections_in_a_row += 1;
            after_rejection = rejections_in_a_row >= 2;
        }
#ifdef MDB_WAVE_TIMING
        timed[WAVE_T_FINISH - WAVE_T_FIRST_STAGE] += __builtin_amdgcn_s_memtime() - finish_from;
#endif
    }
    __syncthreads();
    flush_pending();
    if (lane == 0 && !PIECES) plans[chunk] = {n_models, gaps.finish(n)};
    counted[WAVE_MODELS] = n_models;
    report_counts();
}

// ---- k_fit_walk (split mode) ---------------------------------------------------------------------------------
//
// One wave per chunk follows the table from point 0. What it reads decides where it reads next, so every look at the
// table is a trip to memory the wave waits for - and that, not bandwidth, was the kernel: a chunk of nothing but
// rejected points cost 1 024 trips of 64 entries, a chunk of 26-point models a trip for the entries and another for the
// two parameters of EVERY model. Now:
//   * the entries are staged in LDS a stretch at a time, the stretch as long as the walk is dense: 64 entries behind a
//     jump far past what was staged (a model much longer than the stretch: sine data pays one trip per model as
//     before, not kilobytes per model), twice as many every time the walk leaves its stretch by less than four times
//     its length, up to WALK_STRETCH;
//   * a model costs no trip: its start, end and type go to a list in LDS, and when 64 are listed (or the chunk is
//     through) every lane fetches one model's parameters and writes its record;
//   * a walk that has reached the longest stretch asks for the stretch behind it while it walks this one (the entries
//     wait in registers), and knows of every 64 staged entries whether any is something else than rejected: a chunk of
//     10^6 points of noise - one MacaqueV segment - is 512 stretches that are looked at once each, not 16 384 groups.
constexpr uint32_t WALK_STRETCH = 2048; // entries staged at most (8 KB of LDS per wave)
#ifdef MDB_WALK_DEBUG
__device__ unsigned long long g_walk_counts[8];
#endif

__global__ __launch_bounds__(MDB_WAVE) void k_fit_walk(const unsigned long long *__restrict__ chunk_offsets,
                                                       uint64_t n_chunks, SplitArgs split,
                                                       const unsigned long long *__restrict__ record_base,
                                                       ModelRec *__restrict__ records,
                                                       ChunkPlan *__restrict__ plans,
                                                       unsigned int *__restrict__ error) {
    __shared__ uint32_t staged[WALK_STRETCH];
    __shared__ uint32_t listed_start[MDB_WAVE], listed_end[MDB_WAVE]; // (start with the type in its top bit)
    const uint64_t chunk = blockIdx.x;
    if (chunk >= n_chunks) return;
    if (split.chunk_left && split.chunk_left[chunk] == 0u) return;
    const int lane = threadIdx.x;
    const uint64_t base = chunk_offsets[chunk];
    const uint64_t length64 = chunk_offsets[chunk + 1] - base;
    if (length64 > COUNT_MASK - ENTRY_END_BIAS) {
        if (lane == 0) {
            atomicOr(error, ERR_TOO_LONG);
            plans[chunk] = {0, 0};
        }
        return;
    }
    const uint32_t n = (uint32_t)length64;
    ModelRec *__restrict__ out = records + record_base[chunk];
    uint32_t n_models = 0, n_listed = 0;
    GapCounter gaps;
    uint32_t position = 0;
    uint32_t stretch_first = 0, stretch_size = 0, next_size = MDB_WAVE;
    constexpr uint32_t GROUPS = WALK_STRETCH / MDB_WAVE;
    uint32_t ahead[GROUPS];                 // the stretch behind the staged one, asked for while this one is walked
    uint32_t ahead_first = 0xffffffffu, ahead_size = 0;
    uint32_t live_groups = 0;               // bit g: entries [64 g, 64 g + 64) of the stretch hold something not rejected
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // The listed models' records: lane k fetches the parameters of the k-th and writes record n_models - n_listed + k.
    auto write_listed = [&]() {
        wave_sync();
        if ((uint32_t)lane < n_listed) {
            const uint32_t start = listed_start[lane];
            ModelRec rec;
            rec.start_and_type = start;
            rec.end = listed_end[lane];
            rec.p0 = split.p0[base + (start & COUNT_MASK)];
            rec.p1 = split.p1[base + (start & COUNT_MASK)];
            out[n_models - n_listed + (uint32_t)lane] = rec;
        }
        wave_sync();
        n_listed = 0;
    };
#ifdef MDB_WALK_DEBUG
    unsigned long long dbg_iter = 0, dbg_reload = 0, dbg_flush = 0, dbg_rej = 0;
    const unsigned long long dbg_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    while (position < n) {
#ifdef MDB_WALK_DEBUG
        dbg_iter++;
        if (position < stretch_first || position - stretch_first >= stretch_size) dbg_reload++;
#endif
        if (position < stretch_first || position - stretch_first >= stretch_size) {
            // (dense: the walk has come no further than four stretches from where the last one began - rejected points,
            // or models that are short against what is staged)
            const bool asked_for = ahead_first != 0xffffffffu && position >= ahead_first && position - ahead_first < ahead_size;
            if (asked_for) { // (the stretch behind the last one, asked for while that one was walked)
                stretch_first = ahead_first;
                stretch_size = ahead_size;
            } else {
                const bool dense = stretch_size > 0 && position >= stretch_first && position - stretch_first <= 4u * stretch_size;
                next_size = dense ? min(2u * next_size, WALK_STRETCH) : (uint32_t)MDB_WAVE;
                stretch_first = position;
                stretch_size = min(next_size, n - position);
            }
            wave_sync(); // (nobody reads the old stretch any more)
            live_groups = 0;
            if (stretch_size <= (uint32_t)MDB_WAVE) {
                // (the stretch behind a long model - sine data pays this once per model: one load, not the unrolled
                // thirty-two of the long stretch with all but one masked off; a lone wave issues an instruction every
                // five cycles or so, and a 65 536-point buffer's walk was 0.18 ms instead of 0.08)
                const uint32_t one = (uint32_t)lane < stretch_size ? entry_for_the_walk(split.entry[base + stretch_first + lane]) : ENTRY_REJECTED;
                if ((uint32_t)lane < stretch_size) staged[lane] = one;
                live_groups = __ballot(one != ENTRY_REJECTED) ? 1u : 0u;
            } else {
                if (!asked_for) { // (all loads first, then what depends on them)
#pragma unroll
                    for (uint32_t g = 0; g < GROUPS; g++) {
                        const uint32_t k = g * MDB_WAVE + lane;
                        ahead[g] = k < stretch_size ? entry_for_the_walk(split.entry[base + stretch_first + k]) : ENTRY_REJECTED;
                    }
                }
#pragma unroll
                for (uint32_t g = 0; g < GROUPS; g++) {
                    const uint32_t k = g * MDB_WAVE + lane;
                    if (g * MDB_WAVE < stretch_size) { // (uniform)
                        if (k < stretch_size) staged[k] = ahead[g];
                        if (__ballot(k < stretch_size && ahead[g] != ENTRY_REJECTED)) live_groups |= 1u << g;
                    }
                }
            }
            ahead_first = 0xffffffffu;
            if (stretch_size == WALK_STRETCH && n - stretch_first >= 2u * WALK_STRETCH) {
                ahead_first = stretch_first + WALK_STRETCH;
                ahead_size = WALK_STRETCH;
#pragma unroll
                for (uint32_t g = 0; g < GROUPS; g++) ahead[g] = entry_for_the_walk(split.entry[base + ahead_first + g * MDB_WAVE + lane]);
            }
            wave_sync();
        }
        const uint32_t at = position - stretch_first;
        const uint32_t entry = staged[at];
        if (entry == ENTRY_REJECTED) {
            // A run of rejected points (noise under a lossless bound rejects every point) is skipped 64 at a
            // time: to the first entry of the next 64 staged ones that is something else.
            // (whole groups of nothing but rejected points first: one look at the stretch's mask)
            if ((at & (MDB_WAVE - 1)) == 0u) {
                const uint32_t from_here = live_groups >> (at / MDB_WAVE);
                const uint32_t skipped = from_here ? (uint32_t)__ffs((int)from_here) - 1u : (stretch_size - at + MDB_WAVE - 1) / MDB_WAVE;
                if (skipped > 0) {
                    position = min(stretch_first + stretch_size, position + skipped * (uint32_t)MDB_WAVE);
                    continue;
                }
            }
            const uint32_t mine = at + (uint32_t)lane;
            const unsigned long long others = __ballot(mine < stretch_size && staged[mine] != ENTRY_REJECTED);
            position = others ? position + (uint32_t)__ffsll((long long)others) - 1u
                              : min(stretch_first + stretch_size, position + (uint32_t)MDB_WAVE);
        } else if (entry == 0u) { // cannot happen: every point on the chain was visited by some lane
            if (lane == 0) atomicOr(error, ERR_SPLIT_CHAIN);
            break;
        } else {
            const uint32_t end = (entry & COUNT_MASK) - ENTRY_END_BIAS;
            if (lane == 0) {
                listed_start[n_listed] = position | (entry & 0x80000000u);
                listed_end[n_listed] = end;
            }
            gaps.on_model(position, end);
            n_models += 1;
            n_listed += 1;
            if (n_listed == (uint32_t)MDB_WAVE) write_listed();
            position = end + 1;
        }
    }
    if (n_listed) write_listed();
#ifdef MDB_WALK_DEBUG
    if (lane == 0) {
        atomicAdd(&g_walk_counts[0], dbg_iter); atomicAdd(&g_walk_counts[1], dbg_reload); atomicAdd(&g_walk_counts[2], (unsigned long long)n_models);
        atomicAdd(&g_walk_counts[3], 1ull); atomicAdd(&g_walk_counts[4], __builtin_amdgcn_s_memrealtime() - dbg_t0);
        atomicMax(&g_walk_counts[5], __builtin_amdgcn_s_memrealtime() - dbg_t0);
    }
#endif
    if (lane == 0) plans[chunk] = n == 0 ? ChunkPlan{0, 0} : ChunkPlan{n_models, gaps.finish(n)};
}

// ---- k_fit_plan --------------------------------------------------------------------------------------------

struct SegItem { // 16 bytes
    uint32_t chunk;
    uint32_t first;  // first point of the segment in the chunk
    uint32_t last;   // last point (residuals included)
    uint32_t record; // index into the chunk's model records, 0xffffffff: MacaqueV-only segment
};

struct SegmentCount {
    const ChunkPlan *plans;
    __device__ uint64_t operator()(uint64_t c) const { return plans[c].n_segments; }
};

// One WAVE per chunk (one lane per chunk walked its models one after the other: 2.4 ms for 4 096 chunks of 2 500 models
// each). Producer j of a chunk with M models: j = 0 the points in front of the first model (a MacaqueV-only segment, if
// there are any), j = 1 .. M - 1 the segment of model j - 1 with the points between it and model j as its residuals - or,
// if they are more than a residual tail holds (compression.rs:310-400), without them and a MacaqueV-only segment for
// them -, j = M the same for the last model and the end of the chunk. Every lane takes a producer, a scan over the
// wave says where its items go.
__global__ __launch_bounds__(MDB_WAVE) void k_fit_plan(const unsigned long long *__restrict__ chunk_offsets,
                                                       uint64_t n_chunks,
                                                       const unsigned long long *__restrict__ record_base,
                                                       const ModelRec *__restrict__ records,
                                                       const ChunkPlan *__restrict__ plans,
                                                       const unsigned long long *__restrict__ segment_base,
                                                       SegItem *__restrict__ items) {
    const uint64_t chunk = blockIdx.x;
    if (chunk >= n_chunks) return;
    const int lane = threadIdx.x;
    const uint32_t n = (uint32_t)(chunk_offsets[chunk + 1] - chunk_offsets[chunk]);
    const ModelRec *__restrict__ recs = records + record_base[chunk];
    const uint32_t n_models = plans[chunk].n_models;
    SegItem *__restrict__ out = items + segment_base[chunk];
    if (n == 0) return;
    if (n_models == 0) {
        if (lane == 0) out[0] = {(uint32_t)chunk, 0, n - 1, 0xffffffffu};
        return;
    }
    uint32_t placed = 0; // items of the producers in front of this round's
    for (uint32_t first = 0; first <= n_models; first += MDB_WAVE) {
        const uint32_t j = first + (uint32_t)lane;
        uint32_t count = 0;
        SegItem one = {}, two = {};
        if (j == 0) {
            const uint32_t start = recs[0].start_and_type & COUNT_MASK;
            if (start > 0) {
                one = {(uint32_t)chunk, 0, start - 1, 0xffffffffu};
                count = 1;
            }
        } else if (j <= n_models) {
            const uint32_t previous_start = recs[j - 1].start_and_type & COUNT_MASK;
            const uint32_t previous_end = recs[j - 1].end;
            const uint32_t residuals_end = j < n_models ? (recs[j].start_and_type & COUNT_MASK) - 1 : n - 1;
            if (residuals_end - previous_end <= MDB_RESIDUAL_VALUES_MAX_LENGTH) {
                one = {(uint32_t)chunk, previous_start, residuals_end, j - 1};
                count = 1;
            } else {
                one = {(uint32_t)chunk, previous_start, previous_end, j - 1};
                two = {(uint32_t)chunk, previous_end + 1, residuals_end, 0xffffffffu};
                count = 2;
            }
        }
        uint32_t before = count;
#pragma unroll
        for (int delta = 1; delta < MDB_WAVE; delta <<= 1) {
            const uint32_t up = __shfl_up(before, delta, MDB_WAVE);
            if (lane >= delta) before += up;
        }
        const uint32_t round_total = __shfl(before, MDB_WAVE - 1, MDB_WAVE);
        before -= count;
        if (count >= 1) out[placed + before] = one;
        if (count == 2) out[placed + before + 1] = two;
        placed += round_total;
    }
}

// ---- k_fit_size / k_fit_encode -------------------------------------------------------------------------------

struct SegSizes { // payload bytes per BinaryView column
    uint32_t timestamps;
    uint32_t values;
    uint32_t residuals;
    uint32_t pad;
};

struct OutOfLineBytes {
    const SegSizes *sizes;
    int column;
    __device__ uint64_t operator()(uint64_t i) const {
        uint32_t n = column == 0 ? sizes[i].timestamps : (column == 1 ? sizes[i].values : sizes[i].residuals);
        return n > 12 ? n : 0;
    }
};

struct EncodeTargets {
    int8_t *model_type_id;
    int64_t *start_time;
    int64_t *end_time;
    float *min_value;
    float *max_value;
    float *error;
    uint32_t *chunk_index;
    uint4 *views[3];
    uint8_t *data[3]; // the column's payloads, one after the other
    const unsigned long long *data_offsets[3];
    // The payloads of a column are ONE run of device memory presented as several Arrow data buffers (a
    // view addresses at most 2 GiB of one buffer): buffer k begins at the payload that begins at or
    // behind k x MDB_FIT_DATA_BUFFER_BYTES, data_bases[c][k] is where.
    const unsigned long long *data_bases[3];
    uint32_t n_data_buffers[3];
};

// Which data buffer the payload at `offset` of a column lies in, and where inside it (the view's
// buffer_index and offset, types.rs:444-516 leaves the same to arrow's BinaryViewBuilder).
__device__ __forceinline__ void data_buffer_of(const unsigned long long *bases, uint32_t n_buffers, uint64_t offset,
                                               uint32_t *buffer, uint32_t *offset_in_buffer) {
    uint32_t lo = 0, hi = n_buffers; // the last buffer that begins at or in front of the payload
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) / 2;
        if (bases[mid] <= offset) lo = mid;
        else hi = mid;
    }
    *buffer = lo;
    *offset_in_buffer = (uint32_t)(offset - bases[lo]);
}

// data_bases of one column: buffer k begins at the first payload at or behind k x buffer_bytes.
// data_offsets is the exclusive scan of the out-of-line payload sizes over the segments (n + 1 entries).
__global__ __launch_bounds__(64) void k_fit_data_bases(const unsigned long long *__restrict__ data_offsets,
                                                       uint64_t n_segments, uint64_t buffer_bytes, uint32_t n_buffers,
                                                       unsigned long long *__restrict__ bases) {
    for (uint32_t k = threadIdx.x; k < n_buffers; k += blockDim.x) {
        const uint64_t wanted = (uint64_t)k * buffer_bytes;
        uint64_t lo = 0, hi = n_segments; // the first segment whose payload begins at or behind `wanted`
        while (lo < hi) {
            const uint64_t mid = (lo + hi) / 2;
            if (data_offsets[mid] >= wanted) hi = mid;
            else lo = mid + 1;
        }
        bases[k] = data_offsets[lo];
    }
}

struct TsResult { // of k_fit_timestamps<false>, indexed by segment
    uint32_t bytes; // length of compress_residual_timestamps(); 0xffffffff: the one-lane path does it
    uint32_t regular;
};

struct GapResult { // of k_fit_gap<GAP_SIZE>, indexed by segment
    uint32_t values_bytes;
    float min_value;
    float max_value;
    uint32_t regular; // are the timestamps of the segment equally spaced? (timestamps.rs:56-97)
};

__device__ __forceinline__ bool gap_goes_to_a_wave(const FitArgs &args, const SegItem &item) {
    return item.record == 0xffffffffu && item.last - item.first + 1 >= args.gap_min_values;
}

// Runs every encoder of one segment against `Sink`-typed sinks created by `make_sink(column, bytes)`.
// With CountSink it sizes the payloads; with ByteSink it writes them.
template <bool WRITE>
__device__ __forceinline__ void process_segment(const FitArgs &args, const unsigned long long *record_base,
                                                const ModelRec *records, const SegItem &item,
                                                uint64_t segment, SegSizes *sizes_io,
                                                const EncodeTargets *targets) {
    const uint64_t base = args.chunk_offsets[item.chunk];
    const float *__restrict__ values = args.values + base;
    const ChunkTimestamps ts = chunk_timestamps(args.timestamps, item.chunk, base);
    const mdb_error_bound eb = args.eb;

    // Destination of a payload: inline in the view (<= 12 bytes) or in the column's data buffer.
    auto payload_destination = [&](int column, uint32_t bytes) -> uint8_t * {
        if (!WRITE) return nullptr;
        if (bytes <= 12) return reinterpret_cast<uint8_t *>(targets->views[column] + segment) + 4;
        return targets->data[column] + targets->data_offsets[column][segment];
    };
    auto finish_view = [&](int column, uint32_t bytes) {
        if (!WRITE || bytes <= 12) return;
        // Out-of-line view: length, 4-byte prefix, buffer index, offset in that buffer (Arrow BinaryView).
        const uint64_t offset = targets->data_offsets[column][segment];
        const uint8_t *p = targets->data[column] + offset;
        uint4 view;
        view.x = bytes;
        view.y = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
        data_buffer_of(targets->data_bases[column], targets->n_data_buffers[column], offset, &view.z, &view.w);
        targets->views[column][segment] = view;
    };
    if (WRITE) {
        // Inline views start as {length, 0, 0, 0}; payload bytes are then stored into the zeros.
        targets->views[0][segment] = make_uint4(sizes_io->timestamps, 0, 0, 0);
        targets->views[1][segment] = make_uint4(sizes_io->values, 0, 0, 0);
        targets->views[2][segment] = make_uint4(sizes_io->residuals, 0, 0, 0);
    }

    // -- timestamps (timestamps.rs:56-155) over [first, last]
    const uint32_t count = item.last - item.first + 1;
    uint32_t ts_bytes;
    const bool ts_by_wave = args.ts_results != nullptr && args.ts_results[segment].bytes != 0xffffffffu;
    if (!WRITE && ts_by_wave) {
        ts_bytes = args.ts_results[segment].bytes;
        sizes_io->pad = args.ts_results[segment].regular;
    } else if (!WRITE) {
        bool regular;
        int known_regular = gap_goes_to_a_wave(args, item) ? (int)args.gap_results[segment].regular : -1;
        if (known_regular < 0 && args.timestamps.ts && args.timestamps.chunk_irregular &&
            !args.timestamps.chunk_irregular[item.chunk])
            known_regular = 1; // the whole chunk is equally spaced
        ts_bytes = timestamps_payload_length(ts, item.first, item.last, &regular, known_regular);
        sizes_io->pad = regular ? 1u : 0u;
    } else {
        ts_bytes = sizes_io->timestamps;
        uint8_t *dst = payload_destination(0, ts_bytes);
        if (ts_bytes > 0) {
            if (sizes_io->pad) { // regular: the length, big endian
                for (uint32_t k = 0; k < ts_bytes; k++)
                    dst[k] = (uint8_t)((uint64_t)count >> (8 * (ts_bytes - 1 - k)));
            } else if (ts_by_wave && ts_bytes > 12) {
                // k_fit_timestamps<true> has written it (before this kernel: the view reads its prefix)
            } else {
                ByteSink sink(dst);
                encode_irregular_timestamps(sink, ts, item.first, item.last);
                sink.finish(true);
            }
        }
        finish_view(0, ts_bytes);
    }

    int8_t type;
    float min_value, max_value;
    uint32_t values_bytes = 0, residual_bytes = 0;
    if (item.record == 0xffffffffu) {
        // compress_and_store_residuals_in_a_separate_segment (compression.rs:367-400)
        type = MDB_MACAQUE_V_ID;
        MacaqueState m;
        if (gap_goes_to_a_wave(args, item) && (!WRITE || sizes_io->values > 12)) {
            // k_fit_gap sized it before k_fit_size and wrote it before k_fit_encode (payloads of up to
            // 12 bytes live inside the view, which this lane writes itself below).
            const GapResult gap = args.gap_results[segment];
            values_bytes = gap.values_bytes;
            m.min_value = gap.min_value;
            m.max_value = gap.max_value;
            if (WRITE) finish_view(1, values_bytes);
        } else if (!WRITE) {
            CountSink sink;
            macaque_encode(m, sink, eb, values + item.first, count, false, 0.0f);
            values_bytes = (uint32_t)sink.bytes();
        } else {
            values_bytes = sizes_io->values;
            ByteSink sink(payload_destination(1, values_bytes));
            macaque_encode(m, sink, eb, values + item.first, count, false, 0.0f);
            sink.finish(false);
            finish_view(1, values_bytes);
        }
        min_value = m.min_value;
        max_value = m.max_value;
    } else {
        // CompressedSegmentBuilder::finish (types.rs:197-267)
        const ModelRec rec = records[record_base[item.chunk] + item.record];
        const bool is_swing = (rec.start_and_type & 0x80000000u) != 0;
        type = is_swing ? MDB_SWING_ID : MDB_PMC_MEAN_ID;
        float model_last_value;
        uint8_t encoded[8];
        if (is_swing) { // types.rs:122-144
            min_value = min_num(rec.p0, rec.p1);
            max_value = max_num(rec.p0, rec.p1);
            model_last_value = rec.p1;
            if (!(rec.p0 < rec.p1)) {
                encoded[0] = 0;
                values_bytes = 1;
            }
        } else { // types.rs:104-119
            min_value = rec.p0;
            max_value = rec.p0;
            model_last_value = rec.p0;
        }
        if (rec.end < item.last) {
            const uint32_t n_residuals = item.last - rec.end;
            MacaqueState m;
            if (!WRITE) {
                CountSink sink;
                macaque_encode(m, sink, eb, values + rec.end + 1, n_residuals, true, model_last_value);
                residual_bytes = (uint32_t)sink.bytes() + 1;
            } else {
                residual_bytes = sizes_io->residuals;
                uint8_t *dst = payload_destination(2, residual_bytes);
                ByteSink sink(dst);
                macaque_encode(m, sink, eb, values + rec.end + 1, n_residuals, true, model_last_value);
                sink.finish(false);
                dst[residual_bytes - 1] = (uint8_t)n_residuals; // types.rs:249-250
                finish_view(2, residual_bytes);
            }
            const float rmin = m.min_value, rmax = m.max_value;
            if (is_swing)
                values_bytes = encode_values_for_swing(min_value, max_value, values_bytes == 0, rmin, rmax, encoded);
            else
                values_bytes = encode_values_for_pmc_mean(min_value, max_value, rmin, rmax, encoded);
            min_value = min_num(min_value, rmin);
            max_value = max_num(max_value, rmax);
        }
        if (WRITE) {
            uint8_t *dst = payload_destination(1, values_bytes); // always <= 8 bytes: inline
            for (uint32_t k = 0; k < values_bytes; k++) dst[k] = encoded[k];
        }
    }

    if (!WRITE) {
        sizes_io->timestamps = ts_bytes;
        sizes_io->values = values_bytes;
        sizes_io->residuals = residual_bytes;
    } else {
        targets->model_type_id[segment] = type;
        targets->start_time[segment] = ts.at(item.first);
        targets->end_time[segment] = ts.at(item.last);
        targets->min_value[segment] = min_value;
        targets->max_value[segment] = max_value;
        targets->error[segment] = __uint_as_float(0x7fc00000u); // f32::NAN (types.rs:265)
        targets->chunk_index[segment] = item.chunk;
    }
}

// ---- k_fit_gap: one wave per long MacaqueV-only segment -----------------------------------------------------
//
// macaque_v.rs:76-164: what is stored for value i is XORed with what was stored for value i-1, and
// the only other thing carried from code to code is the window (leading / trailing zeros of the last
// `11` code). A wave takes 64 values at a time: every lane computes its XOR, then the lanes whose
// XOR does not fit the carried window are found with a ballot, one after the other - each of them
// opens a new window for the lanes behind it. Code lengths are prefix-summed into bit offsets; in
// write mode the codes are ORed into an LDS bit buffer and whole bytes flushed. Bit-identical to
// the one-lane encoder, which for a 65 536-value chunk of noise needs 2 x 30 ms at the latency of a
// single lane.

constexpr uint32_t GAP_DEFAULT_MIN_VALUES = 256;
constexpr int GAP_BUFFER_WORDS = 96; // 64 codes x 45 bits + a carried partial byte

__global__ __launch_bounds__(256) void k_fit_gap_select(FitArgs args, const SegItem *__restrict__ items,
                                                        uint64_t n_segments, uint32_t *__restrict__ gap_ids,
                                                        uint32_t *__restrict__ n_gaps) {
    // (one atomic on the list's counter per workgroup: one per listed segment was 1.4 ms for the mixed series' 8.5 M
    // segments, a third of them MacaqueV segments of their own)
    __shared__ uint32_t listed_by_wave[256 / MDB_WAVE + 1];
    if (args.n_segments_dev) n_segments = *args.n_segments_dev;
    const uint64_t segment = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x % MDB_WAVE, wave = threadIdx.x / MDB_WAVE;
    // (the long ones are k_fit_long_select's)
    const bool listed = segment < n_segments && gap_goes_to_a_wave(args, items[segment]) &&
                        items[segment].last - items[segment].first + 1 < args.gap_long_min_values;
    const unsigned long long lanes = __ballot(listed);
    if (lane == 0) listed_by_wave[wave] = (uint32_t)__popcll(lanes);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int w = 0; w < 256 / MDB_WAVE; w++) {
            const uint32_t of_wave = listed_by_wave[w];
            listed_by_wave[w] = total;
            total += of_wave;
        }
        listed_by_wave[256 / MDB_WAVE] = total ? atomicAdd(n_gaps, total) : 0u;
    }
    __syncthreads();
    if (listed)
        gap_ids[listed_by_wave[256 / MDB_WAVE] + listed_by_wave[wave] + (uint32_t)__popcll(lanes & ((1ull << lane) - 1ull))] = (uint32_t)segment;
}

// value (count <= 32 bits, right aligned) -> bits [at, at + count) of the big-endian bit buffer.
__device__ __forceinline__ void gap_put(uint32_t *buffer, uint32_t at, uint32_t value, uint32_t count) {
    if (count == 0) return;
    const uint32_t word = at >> 5, offset = at & 31u;
    if (offset + count <= 32u) {
        atomicOr(&buffer[word], value << (32u - offset - count));
    } else {
        const uint32_t low_bits = offset + count - 32u; // bits that go to the next word
        atomicOr(&buffer[word], value >> low_bits);
        atomicOr(&buffer[word + 1], value << (32u - low_bits));
    }
}

// The LDS bit buffer of a wave holds `buffered` bits (the codes of one batch behind a carried partial
// byte, all lanes' atomicOr's done): its whole bytes go to dst + *written, the partial byte moves to the
// front of the cleared buffer. Returns how many bits that partial byte has.
__device__ __forceinline__ uint32_t wave_flush_bits(uint32_t *buffer, int buffer_words, uint8_t *__restrict__ dst,
                                                    uint64_t *written, uint32_t buffered, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t whole_bytes = buffered >> 3;
    for (uint32_t b = lane; b < whole_bytes; b += MDB_WAVE)
        dst[*written + b] = (uint8_t)(buffer[b >> 2] >> (24u - 8u * (b & 3u)));
    const uint32_t partial = (buffer[whole_bytes >> 2] >> (24u - 8u * (whole_bytes & 3u))) & 0xffu;
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < buffer_words; k += MDB_WAVE) buffer[k] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t carry_bits = buffered & 7u;
    if (lane == 0 && carry_bits) buffer[0] = partial << 24;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    *written += whole_bytes;
    return carry_bits;
}

// GAP_SIZE: the stream is measured (its bytes, min / max, whether the timestamps are equally spaced); GAP_WRITE: it is
// written where the scan over the sizes has put it; GAP_STAGE: both at once - measured, and written to a staging place
// found from an upper bound (45 bits a value), from where k_fit_gap_place copies it once the scan has run: the codes are
// worked out once, not twice (the copy is a quarter of what the second encoding was).
enum GapMode { GAP_SIZE = 0, GAP_WRITE = 1, GAP_STAGE = 2 };
struct GapStage {
    uint8_t *bytes = nullptr;                    // 16-byte aligned places, in the order of the list of segments
    const unsigned long long *offsets = nullptr; // [position in the list]
};
// Staging bytes of a MacaqueV-only segment of n values: 32 raw bits, at most 2 + 5 + 6 + 32 bits for every other value
// (macaque_v.rs:120-164), up to the next multiple of 16 and 16 more (the copy reads whole words past the end).
__host__ __device__ inline unsigned long long gap_stage_bytes(uint32_t n) {
    return ((4ull + ((unsigned long long)(n - 1) * 45u + 7u) / 8u + 15u) & ~15ull) + 16u;
}
struct GapStageBytes {
    const SegItem *items;
    const uint32_t *gap_ids;
    __device__ uint64_t operator()(uint64_t g) const {
        const SegItem item = items[gap_ids[g]];
        return gap_stage_bytes(item.last - item.first + 1);
    }
};

template <int MODE>
__global__ __launch_bounds__(MDB_WAVE) void k_fit_gap(FitArgs args, const SegItem *__restrict__ items,
                                                      const uint32_t *__restrict__ gap_ids,
                                                      const uint32_t *__restrict__ n_gaps,
                                                      GapResult *__restrict__ results, EncodeTargets targets,
                                                      GapStage stage = GapStage{}) {
    constexpr bool WRITE = MODE != GAP_SIZE;    // the codes go to memory
    constexpr bool MEASURE = MODE != GAP_WRITE; // the segment's GapResult is made here
    __shared__ uint32_t buffer[GAP_BUFFER_WORDS];
    if (blockIdx.x >= *n_gaps) return;
    if (MODE == GAP_WRITE && args.targets_dev) targets = *args.targets_dev;
    const uint32_t segment = gap_ids[blockIdx.x];
    const SegItem item = items[segment];
    const int lane = threadIdx.x;
    const uint32_t n = item.last - item.first + 1;
    const float *__restrict__ values = args.values + args.chunk_offsets[item.chunk] + item.first;
    uint8_t *__restrict__ dst = nullptr;
    if (MODE == GAP_WRITE) {
        if (results[segment].values_bytes <= 12) return; // lives inside the view: k_fit_encode writes it
        dst = targets.data[1] + targets.data_offsets[1][segment];
    } else if (MODE == GAP_STAGE) {
        dst = stage.bytes + stage.offsets[blockIdx.x];
    }

    // The first value: 32 raw bits.
    const uint32_t first_bits = __float_as_uint(values[0]);
    float min_value = values[0], max_value = values[0]; // min_num / max_num of NaN and the first value
    uint64_t total_bits = 32;
    uint64_t written = 0; // bytes stored so far
    uint32_t carry_bits = 0; // bits of a partial byte waiting at the front of the buffer
    if (WRITE) {
        if (lane < 4) dst[lane] = (uint8_t)(first_bits >> (24 - 8 * lane));
        written = 4;
        for (int k = lane; k < GAP_BUFFER_WORDS; k += MDB_WAVE) buffer[k] = 0;
    }
    uint32_t window_leading = 255, window_trailing = 0; // uniform: macaque_v.rs:63-64
    const mdb_error_bound eb = args.eb;
    const DeviationFactor deviation = deviation_factor(eb);
    float carried_in = values[0];  // uniform: the value stored last (the first one is stored as it is)
    float last_stored = values[0];
    // While sizing, the wave also checks whether the timestamps are equally spaced, which the one-lane
    // path (chunk_range_regular) would otherwise do point by point.
    const ChunkTimestamps ts = chunk_timestamps(args.timestamps, item.chunk, args.chunk_offsets[item.chunk]);
    bool regular = true;
    const int64_t expected_delta = (MEASURE && ts.ts && n >= 2) ? ts.ts[item.first + 1] - ts.ts[item.first] : 0;
    // (asking for a batch's values while the batch before it is worked on: 2.57 -> 2.54 / 3.40 -> 3.45 ms, nothing)
    for (uint32_t base = 1; base < n; base += MDB_WAVE) {
        const uint32_t i = base + lane;
        const bool active = i < n;
        if (MEASURE && ts.ts && active && ts.ts[item.first + i] - ts.ts[item.first + i - 1] != expected_delta)
            regular = false;
        // What is stored for value i (macaque_v.rs:100-118). Lossless: the value. Otherwise the value
        // stored before it if that is within the bound of value i, else value i with its least
        // mantissa bits rewritten - which every lane can work out for its own value up front; who
        // keeps the carried value and who starts a new one is again settled with ballots.
        float stored = active ? values[i] : 0.0f;
        if (eb.kind != MDB_EB_LOSSLESS) {
            const float raw = stored;
            const float rewritten = active ? rewrite_least_mantissa_bits(eb, deviation, raw) : 0.0f;
            // Where nearly every value breaks away from the one stored before it (noise under a tight bound), one
            // trip per breaker is 64 trips per batch. If lane i - 1 stores its own value, lane i is compared with
            // that: whether it then breaks too is known up front for all lanes at once, so a run of breakers
            // behind a breaker is taken in one trip.
            const float rewritten_before = dpp_move<0x138>(rewritten); // wave_shr:1
            const unsigned long long breaks_behind_a_breaker =
                __ballot(active && lane > 0 && !within_error_bound(eb, raw, rewritten_before));
            int cursor = 0;
            while (true) {
                const bool breaks = active && lane >= cursor && !within_error_bound(eb, raw, last_stored);
                const unsigned long long mask = __ballot(breaks);
                if (mask == 0) {
                    if (lane >= cursor) stored = last_stored;
                    break;
                }
                const int breaker = __ffsll((long long)mask) - 1;
                if (lane >= cursor && lane < breaker) stored = last_stored;
                const unsigned long long behind = breaker + 1 < MDB_WAVE ? breaks_behind_a_breaker >> (breaker + 1) : 0ull;
                const int last_breaker = breaker + (~behind ? __builtin_ctzll(~behind) : MDB_WAVE - 1 - breaker);
                if (lane >= breaker && lane <= last_breaker) stored = rewritten;
                last_stored = read_lane(rewritten, last_breaker);
                cursor = last_breaker + 1;
            }
        }
        const float before = dpp_move<0x138>(stored); // wave_shr:1 (lane 0: not used)
        const uint32_t current = active ? __float_as_uint(stored) : 0u;
        const uint32_t previous = active ? __float_as_uint(lane == 0 ? carried_in : before) : 0u;
        // The value stored last in this batch is what the next batch starts from.
        const int last_lane = (int)min((uint32_t)MDB_WAVE, n - base) - 1;
        carried_in = read_lane(stored, last_lane);
        last_stored = carried_in;
        const uint32_t x = current ^ previous;
        const bool repeat = x == 0;
        const uint32_t leading = repeat ? 32u : (uint32_t)__clz((int)x);
        const uint32_t trailing = repeat ? 32u : (uint32_t)__ffs((int)x) - 1u;
        // The window each lane's code is written with; lanes that open one get their own.
        uint32_t my_leading = window_leading, my_trailing = window_trailing;
        bool opens = false;
        // (runs of lanes that each open a window of their own in one trip, as with the stored values above: a lane
        // behind an opener is written with that lane's window unless it does not fit it)
        const uint32_t leading_before = dpp_move<0x138>(leading), trailing_before = dpp_move<0x138>(trailing);
        const unsigned long long opens_behind_an_opener =
            __ballot(active && lane > 0 && !repeat && !(leading >= leading_before && trailing >= trailing_before));
        int cursor = 0;
        while (true) {
            const bool misfit = active && lane >= cursor && !repeat &&
                                !(leading >= window_leading && trailing >= window_trailing);
            const unsigned long long mask = __ballot(misfit);
            if (mask == 0) {
                if (lane >= cursor) {
                    my_leading = window_leading;
                    my_trailing = window_trailing;
                }
                break;
            }
            const int opener = __ffsll((long long)mask) - 1;
            if (lane >= cursor && lane < opener) {
                my_leading = window_leading;
                my_trailing = window_trailing;
            }
            const unsigned long long behind = opener + 1 < MDB_WAVE ? opens_behind_an_opener >> (opener + 1) : 0ull;
            const int last_opener = opener + (~behind ? __builtin_ctzll(~behind) : MDB_WAVE - 1 - opener);
            if (lane >= opener && lane <= last_opener) {
                opens = true;
                my_leading = leading;
                my_trailing = trailing;
            }
            window_leading = read_lane(leading, last_opener);
            window_trailing = read_lane(trailing, last_opener);
            cursor = last_opener + 1;
        }
        const uint32_t meaningful = 32u - my_leading - my_trailing;
        uint32_t code_bits = 0;
        if (active) code_bits = repeat ? 2u : (opens ? 13u + meaningful : 1u + meaningful);
        // Exclusive prefix sum of the code lengths: where each code goes. (A pass that only measures needs their sum,
        // and a batch in which no window opens - nearly every batch of noise - has it in closed form: two bits a repeat,
        // one and the window's a code.)
        uint32_t inclusive = 0, batch_bits;
        const unsigned long long opening = WRITE ? 1ull : __ballot(opens);
        if (!WRITE && opening == 0) {
            const uint32_t repeats = (uint32_t)__popcll(__ballot(active && repeat)), codes = (uint32_t)__popcll(__ballot(active));
            batch_bits = 2u * repeats + (codes - repeats) * (33u - window_leading - window_trailing);
        } else {
            inclusive = wave_inclusive_scan(code_bits, lane, [](uint32_t a, uint32_t b) { return a + b; });
            batch_bits = read_lane(inclusive, MDB_WAVE - 1);
        }
        // min / max with the first operand kept on ties, NaN as the neutral element (macaque_v.rs:199-204); earlier
        // batches are the first operand, so that e.g. the sign of a zero minimum is the one the sequential encoder
        // would report.
        if (MEASURE) {
            const ValueRange range = wave_value_range(active, stored);
            min_value = min_num(min_value, range.low);
            max_value = max_num(max_value, range.high);
        }
        if (WRITE) {
            if (active) {
                const uint32_t at = carry_bits + inclusive - code_bits;
                if (repeat) {
                    gap_put(buffer, at, 0b10u, 2);
                } else if (opens) {
                    gap_put(buffer, at, (0b11u << 11) | (my_leading << 6) | meaningful, 13);
                    gap_put(buffer, at + 13, x >> my_trailing, meaningful);
                } else {
                    gap_put(buffer, at, 0u, 1);
                    gap_put(buffer, at + 1, x >> my_trailing, meaningful);
                }
            }
            carry_bits = wave_flush_bits(buffer, GAP_BUFFER_WORDS, dst, &written, carry_bits + batch_bits, lane);
        }
        total_bits += batch_bits;
    }
    const bool all_regular = __all(regular);
    if (WRITE) {
        if (carry_bits && lane == 0) dst[written] = (uint8_t)(buffer[0] >> 24); // padded with zero bits
    }
    if (MEASURE && lane == 0) {
        GapResult result;
        result.values_bytes = (uint32_t)((total_bits + 7) >> 3);
        result.min_value = min_value;
        result.max_value = max_value;
        result.regular = all_regular ? 1u : 0u;
        results[segment] = result;
    }
}

// The staged streams to their places (GAP_STAGE): a wave per segment copies values_bytes bytes from a 16-byte aligned
// place to one of any alignment - aligned words read, shifted by the difference, aligned words written; the few bytes
// in front of the first and behind the last whole word of the destination one by one. (Streams of up to 12 bytes live in
// their views: k_fit_encode writes those.)
__global__ __launch_bounds__(MDB_WAVE) void k_fit_gap_place(const uint32_t *__restrict__ gap_ids, const uint32_t *__restrict__ n_gaps,
                                                            const GapResult *__restrict__ results, EncodeTargets targets,
                                                            GapStage stage) {
    if (blockIdx.x >= *n_gaps) return;
    const uint32_t segment = gap_ids[blockIdx.x];
    const uint32_t n_bytes = results[segment].values_bytes;
    if (n_bytes <= 12) return;
    const int lane = threadIdx.x;
    const uint8_t *__restrict__ from = stage.bytes + stage.offsets[blockIdx.x];
    uint8_t *__restrict__ to = targets.data[1] + targets.data_offsets[1][segment];
    const uint32_t head = (uint32_t)((4u - (reinterpret_cast<uintptr_t>(to) & 3u)) & 3u); // (n_bytes > 12 > head)
    if ((uint32_t)lane < head) to[lane] = from[lane];
    const uint32_t n_words = (n_bytes - head) >> 2;
    const uint32_t *__restrict__ from_words = reinterpret_cast<const uint32_t *>(from);
    uint32_t *__restrict__ to_words = reinterpret_cast<uint32_t *>(to + head);
    const uint32_t shift = head & 3u, first_word = head >> 2; // (head < 4: first_word is 0; kept for the form's sake)
    for (uint32_t w = lane; w < n_words; w += MDB_WAVE) {
        const uint32_t low = from_words[first_word + w], high = from_words[first_word + w + 1];
        to_words[w] = __builtin_amdgcn_alignbyte(high, low, shift);
    }
    const uint32_t done = head + 4u * n_words;
    if ((uint32_t)lane < n_bytes - done) to[done + lane] = from[done + lane];
}

// ---- long MacaqueV-only segments: cut into blocks, a wave per block ----------------------------------------
//
// One wave encodes 64 values in about a microsecond, so a stream of 10^6 values - BASELINE's configuration 1, noise
// under a lossless bound handed over as one series - took 2 x 17 ms behind a single wave (and a CPU core 25). What
// runs from code to code is little: the value stored last (lossy bounds: macaque_v.rs:100-118) and the window of the
// last `11` code (:120-164). Both are CHAINS OF RESETS - a value that is not within the bound of the one stored before
// it stores its own rewritten self whatever was stored; a code whose bits do not fit the window opens its own whatever
// the window was - so a wave that starts in the middle of a stream without knowing either falls in with the true chain:
//   * windows: started with the empty window (255, 0) a wave opens one at its first non-zero XOR; from then on its
//     window is never narrower than the true one (if the true encoder did not open where the wave did, the bits fit
//     its window, so the wave's new one contains it; where the true encoder opens, the bits do not fit its window and
//     so not the wave's wider one either), and at the first code the TRUE encoder opens in the block both hold the
//     same window: from there to the end of the block the wave's codes are the stream's.
//   * stored values: two chains hold the same value from the first value on that both of them store anew.
// So every block of a long segment (GapBlock; 1 024 values or more, at most 256 blocks per segment) is sized by a wave
// of its own from such a start (k_fit_long<LONG_SPEC_*>), keeping per batch of 64 values what a later look needs
// (GapBatchNote: the bits before it, the wave's window in front of it, the widest code in it, how many codes are not
// repeats); ONE wave per segment then walks the blocks in order with the true state (k_fit_long_stitch). Noise settles
// on the widest window it needs and never opens another, so the true chain may not reset for a whole block: the walk
// therefore goes over the NOTES, 64 batches per step - a batch in which no code opens a window costs
// 2 x repeats + (33 - leading - trailing) x others bits whatever its values are - up to the first batch in front of
// which the block's wave holds the true window (from there on its codes are the stream's) or in which the true chain
// opens one (that batch alone is encoded again from its values). It leaves every block its true state and bit offset;
// with those the blocks are written by a wave each (k_fit_long<LONG_WRITE>), the byte two blocks share ORed in by
// both. The stream is the one-wave encoder's bit for bit.

constexpr uint32_t GAP_LONG_DEFAULT_MIN_VALUES = 8192;  // (MDB_FIT_GAP_LONG_MIN_VALUES)
constexpr uint32_t GAP_BLOCK_DEFAULT_MIN_VALUES = 1024; // (MDB_FIT_GAP_BLOCK_VALUES; a multiple of 64)
constexpr uint32_t GAP_BLOCKS_PER_SEGMENT = 256;        // the stitch of a segment walks at most this many blocks

// Values per block of a segment with n_codes values behind its first (raw) one.
__host__ __device__ inline uint32_t gap_block_values(uint32_t n_codes, uint32_t block_min) {
    uint64_t block = block_min;
    while (block * GAP_BLOCKS_PER_SEGMENT < n_codes) block <<= 1;
    return (uint32_t)block;
}

struct GapBlock { // 64 bytes, one per block of a long segment
    uint32_t segment;    // index into the items
    uint32_t first;      // index within the segment of the block's first value (>= 1: value 0 is stored raw)
    uint32_t count;      // values of the block
    uint32_t batch_base; // the block's first slot in the per-batch arrays
    // k_fit_long<LONG_SPEC_VALUES>: what is stored last if nothing was stored before the block
    float spec_stored_end;
    // k_fit_long_stitch<STITCH_VALUES>: the value really stored before the block's first
    float stored_in;
    // k_fit_long<LONG_SPEC_SIZE>: the block's codes if the window is empty before it
    uint32_t spec_bits;
    uint32_t spec_end_window; // leading | trailing << 8 | (1 << 16 if a window was opened in the block)
    float min_value, max_value;
    uint32_t regular;
    // k_fit_long_stitch<STITCH_WINDOWS>: the true window before the block, where its codes begin, how many bits
    uint32_t in_window;
    unsigned long long bit_offset;
    uint32_t bits;
    uint32_t reserved;
};
static_assert(sizeof(GapBlock) == 64, "GapBlock is laid out by hand");

struct GapBatchNote { // 16 bytes, one per batch of 64 values of a block; by k_fit_long<LONG_SPEC_SIZE>
    uint32_t prefix_bits;   // bits of the block's codes in front of the batch (from an empty window)
    uint32_t windows;       // the wave's window in front of the batch: leading | trailing << 8; the least leading /
                            // trailing zeros of the batch's non-zero XORs (32: none): << 16 / << 24
    uint32_t others;        // codes of the batch that are not repeats (non-zero XORs)
    float stored_before;    // the value stored in front of the batch's first
};

struct LongCounters { // (lives behind the gap list's counter: [0] ordinary gaps, [1] long segments, [2] blocks, [3] batches)
    uint32_t n_gaps, n_long, n_blocks, n_batches;
};

struct LongArgs {
    const LongCounters *counters;
    uint32_t *long_ids;                    // segment of long segment g
    uint32_t *long_block_base;             // its first block
    uint32_t *long_batch_base;             // its first batch slot
    GapBlock *blocks;
    GapBatchNote *batch_notes;             // what the stitch needs to know of a batch
    unsigned long long *batch_breakers;    // lanes that store themselves anew (from nothing stored); lossy bounds only
    uint32_t long_min_values = 0xffffffffu;
    uint32_t block_min_values = GAP_BLOCK_DEFAULT_MIN_VALUES;
};

__device__ __forceinline__ bool gap_is_long(const LongArgs &la, const SegItem &item) {
    return item.last - item.first + 1 >= la.long_min_values;
}

// The long segments of the call, each with room for its blocks and their batches. (There are few: a lane per segment
// looks, the ones that find one take their slots with atomics.)
__global__ __launch_bounds__(256) void k_fit_long_select(FitArgs args, const SegItem *__restrict__ items, uint64_t n_segments,
                                                         LongArgs la, LongCounters *__restrict__ counters) {
    if (args.n_segments_dev) n_segments = *args.n_segments_dev;
    const uint64_t segment = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (segment >= n_segments) return;
    const SegItem item = items[segment];
    if (!gap_goes_to_a_wave(args, item) || !gap_is_long(la, item)) return;
    const uint32_t n_codes = item.last - item.first;
    const uint32_t block_values = gap_block_values(n_codes, la.block_min_values);
    const uint32_t n_blocks = (n_codes + block_values - 1) / block_values;
    const uint32_t g = atomicAdd(&counters->n_long, 1u);
    la.long_ids[g] = (uint32_t)segment;
    la.long_block_base[g] = atomicAdd(&counters->n_blocks, n_blocks);
    la.long_batch_base[g] = atomicAdd(&counters->n_batches, n_blocks * (block_values / MDB_WAVE));
}

// The headers of a long segment's blocks: one wave per long segment.
__global__ __launch_bounds__(MDB_WAVE) void k_fit_long_blocks(const SegItem *__restrict__ items, LongArgs la) {
    if (blockIdx.x >= la.counters->n_long) return;
    const uint32_t segment = la.long_ids[blockIdx.x];
    const SegItem item = items[segment];
    const uint32_t n_codes = item.last - item.first; // values behind the first
    const uint32_t block_values = gap_block_values(n_codes, la.block_min_values);
    const uint32_t n_blocks = (n_codes + block_values - 1) / block_values;
    const uint32_t block_base = la.long_block_base[blockIdx.x], batch_base = la.long_batch_base[blockIdx.x];
    for (uint32_t b = threadIdx.x; b < n_blocks; b += MDB_WAVE) {
        GapBlock block{};
        block.segment = segment;
        block.first = 1 + b * block_values;
        block.count = min(block_values, n_codes - b * block_values);
        block.batch_base = batch_base + b * (block_values / MDB_WAVE);
        la.blocks[block_base + b] = block;
    }
}

// What the lanes of a batch store (macaque_v.rs:100-118) given the value stored before the batch's first: the value
// stored before it where that is within the bound, otherwise the lane's own value with its least mantissa bits
// rewritten. `last_stored` (uniform) comes back as what the batch's last lane stores; *own: the lanes that store
// themselves anew. (The loop of k_fit_gap.)
__device__ __forceinline__ float gap_value_chain(const mdb_error_bound &eb, const DeviationFactor &deviation, bool active,
                                                 float raw, int lane, int last_lane, float &last_stored,
                                                 unsigned long long *own) {
    float stored = raw;
    unsigned long long anew = 0;
    const float rewritten = active ? rewrite_least_mantissa_bits(eb, deviation, raw) : 0.0f;
    const float rewritten_before = dpp_move<0x138>(rewritten); // wave_shr:1
    const unsigned long long breaks_behind_a_breaker =
        __ballot(active && lane > 0 && !within_error_bound(eb, raw, rewritten_before));
    int cursor = 0;
    while (true) {
        const bool breaks = active && lane >= cursor && !within_error_bound(eb, raw, last_stored);
        const unsigned long long mask = __ballot(breaks);
        if (mask == 0) {
            if (lane >= cursor) stored = last_stored;
            break;
        }
        const int breaker = __ffsll((long long)mask) - 1;
        if (lane >= cursor && lane < breaker) stored = last_stored;
        const unsigned long long behind = breaker + 1 < MDB_WAVE ? breaks_behind_a_breaker >> (breaker + 1) : 0ull;
        const int last_breaker = breaker + (~behind ? __builtin_ctzll(~behind) : MDB_WAVE - 1 - breaker);
        if (lane >= breaker && lane <= last_breaker) stored = rewritten;
        const unsigned long long up_to_last = last_breaker == MDB_WAVE - 1 ? ~0ull : (1ull << (last_breaker + 1)) - 1ull;
        anew |= up_to_last & ~((1ull << breaker) - 1ull);
        last_stored = read_lane(rewritten, last_breaker);
        cursor = last_breaker + 1;
    }
    last_stored = read_lane(stored, last_lane);
    *own = anew;
    return stored;
}

// The window every lane's code is written with (macaque_v.rs:120-164) given the window before the batch's first
// code; the window (uniform) comes back as the one behind the batch's last. Returns whether the lane opens one.
// (The loop of k_fit_gap.)
__device__ __forceinline__ bool gap_window_chain(bool active, bool repeat, uint32_t leading, uint32_t trailing, int lane,
                                                 uint32_t &window_leading, uint32_t &window_trailing, uint32_t *my_leading,
                                                 uint32_t *my_trailing) {
    *my_leading = window_leading;
    *my_trailing = window_trailing;
    bool opens = false;
    const uint32_t leading_before = dpp_move<0x138>(leading), trailing_before = dpp_move<0x138>(trailing);
    const unsigned long long opens_behind_an_opener =
        __ballot(active && lane > 0 && !repeat && !(leading >= leading_before && trailing >= trailing_before));
    int cursor = 0;
    while (true) {
        const bool misfit = active && lane >= cursor && !repeat && !(leading >= window_leading && trailing >= window_trailing);
        const unsigned long long mask = __ballot(misfit);
        if (mask == 0) {
            if (lane >= cursor) {
                *my_leading = window_leading;
                *my_trailing = window_trailing;
            }
            break;
        }
        const int opener = __ffsll((long long)mask) - 1;
        if (lane >= cursor && lane < opener) {
            *my_leading = window_leading;
            *my_trailing = window_trailing;
        }
        const unsigned long long behind = opener + 1 < MDB_WAVE ? opens_behind_an_opener >> (opener + 1) : 0ull;
        const int last_opener = opener + (~behind ? __builtin_ctzll(~behind) : MDB_WAVE - 1 - opener);
        if (lane >= opener && lane <= last_opener) {
            opens = true;
            *my_leading = leading;
            *my_trailing = trailing;
        }
        window_leading = read_lane(leading, last_opener);
        window_trailing = read_lane(trailing, last_opener);
        cursor = last_opener + 1;
    }
    return opens;
}

// One batch of 64 values of a stream: what each lane stores, its code, where the code lies in the batch.
struct GapBatch {
    float stored;
    uint32_t x, code_bits, before_bits; // the XOR with the value stored before; the code's length; bits of the batch in front
    uint32_t my_leading, my_trailing;
    bool repeat, opens;
    uint32_t batch_bits;
    unsigned long long anew; // lanes that store themselves anew (lossy bounds)
    bool any_opens;
    uint32_t leading, trailing; // of the lane's XOR (32: a repeat)
};

// (the body of k_fit_gap's loop: `carried_in` / the window are the stream's state in front of the batch and behind it)
__device__ __forceinline__ GapBatch gap_batch(const mdb_error_bound &eb, const DeviationFactor &deviation, bool active, float raw,
                                              int lane, int last_lane, float &carried_in, uint32_t &window_leading,
                                              uint32_t &window_trailing) {
    GapBatch batch;
    batch.anew = 0;
    batch.stored = active ? raw : 0.0f;
    const float stored_before_batch = carried_in;
    if (eb.kind != MDB_EB_LOSSLESS) {
        batch.stored = gap_value_chain(eb, deviation, active, raw, lane, last_lane, carried_in, &batch.anew);
    } else {
        carried_in = read_lane(batch.stored, last_lane);
    }
    const float before = dpp_move<0x138>(batch.stored); // wave_shr:1 (lane 0: not used)
    const uint32_t current = active ? __float_as_uint(batch.stored) : 0u;
    const uint32_t previous = active ? __float_as_uint(lane == 0 ? stored_before_batch : before) : 0u;
    batch.x = current ^ previous;
    batch.repeat = batch.x == 0;
    const uint32_t leading = batch.repeat ? 32u : (uint32_t)__clz((int)batch.x);
    const uint32_t trailing = batch.repeat ? 32u : (uint32_t)__ffs((int)batch.x) - 1u;
    batch.leading = leading;
    batch.trailing = trailing;
    batch.opens = gap_window_chain(active, batch.repeat, leading, trailing, lane, window_leading, window_trailing,
                                   &batch.my_leading, &batch.my_trailing);
    batch.any_opens = __ballot(batch.opens) != 0;
    const uint32_t meaningful = 32u - batch.my_leading - batch.my_trailing;
    batch.code_bits = 0;
    if (active) batch.code_bits = batch.repeat ? 2u : (batch.opens ? 13u + meaningful : 1u + meaningful);
    const uint32_t inclusive = wave_inclusive_scan(batch.code_bits, lane, [](uint32_t a, uint32_t b) { return a + b; });
    batch.batch_bits = read_lane(inclusive, MDB_WAVE - 1);
    batch.before_bits = inclusive - batch.code_bits;
    return batch;
}

// wave_flush_bits for a block whose first and last byte it shares with its neighbours: those are ORed into memory
// that k_fit_long_clear has zeroed (a byte at a time through the aligned word around it), the bytes in between stored.
__device__ __forceinline__ void or_byte(uint8_t *address, uint32_t byte) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(address);
    atomicOr(reinterpret_cast<unsigned int *>(a & ~(uintptr_t)3), byte << (8u * (uint32_t)(a & 3u)));
}

__device__ __forceinline__ uint32_t wave_flush_bits_shared_first(uint32_t *buffer, int buffer_words, uint8_t *__restrict__ dst,
                                                                 uint64_t *written, uint32_t buffered, int lane,
                                                                 bool *first_byte_shared) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t whole_bytes = buffered >> 3;
    for (uint32_t b = lane; b < whole_bytes; b += MDB_WAVE) {
        const uint32_t byte = (buffer[b >> 2] >> (24u - 8u * (b & 3u))) & 0xffu;
        if (b == 0 && *first_byte_shared) or_byte(dst + *written, byte);
        else dst[*written + b] = (uint8_t)byte;
    }
    if (whole_bytes > 0) *first_byte_shared = false;
    const uint32_t partial = (buffer[whole_bytes >> 2] >> (24u - 8u * (whole_bytes & 3u))) & 0xffu;
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < buffer_words; k += MDB_WAVE) buffer[k] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t carry_bits = buffered & 7u;
    if (lane == 0 && carry_bits) buffer[0] = partial << 24;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    *written += whole_bytes;
    return carry_bits;
}

enum { LONG_SPEC_VALUES = 0, LONG_SPEC_SIZE = 1, LONG_WRITE = 2 };

// One wave per block. LONG_SPEC_VALUES (lossy bounds): which lanes store themselves anew if nothing was stored
// before the block. LONG_SPEC_SIZE: the codes' lengths from an empty window (stored values: the true ones, from
// stored_in). LONG_WRITE: the block's codes, from its true state, where they belong.
template <int MODE>
__global__ __launch_bounds__(MDB_WAVE) void k_fit_long(FitArgs args, const SegItem *__restrict__ items, LongArgs la,
                                                       EncodeTargets targets) {
    __shared__ uint32_t buffer[GAP_BUFFER_WORDS];
    if (blockIdx.x >= la.counters->n_blocks) return;
    if (MODE == LONG_WRITE && args.targets_dev) targets = *args.targets_dev;
    GapBlock &block = la.blocks[blockIdx.x];
    const uint32_t segment = block.segment, first = block.first, count = block.count;
    const SegItem item = items[segment];
    const int lane = threadIdx.x;
    const float *__restrict__ values = args.values + args.chunk_offsets[item.chunk] + item.first;
    const mdb_error_bound eb = args.eb;
    const DeviationFactor deviation = deviation_factor(eb);
    const float nan32 = __uint_as_float(0x7fc00000u);
    const bool lossless = eb.kind == MDB_EB_LOSSLESS;

    if (MODE == LONG_SPEC_VALUES) {
        // (the segment's first block begins behind the first value, which is stored as it is)
        float last_stored = first == 1 ? values[0] : nan32;
        for (uint32_t base = first, k = 0; base < first + count; base += MDB_WAVE, k++) {
            const uint32_t i = base + lane;
            const bool active = i < first + count;
            const int last_lane = (int)min((uint32_t)MDB_WAVE, first + count - base) - 1;
            unsigned long long anew = 0;
            (void)gap_value_chain(eb, deviation, active, active ? values[i] : 0.0f, lane, last_lane, last_stored, &anew);
            if (lane == 0) la.batch_breakers[block.batch_base + k] = anew;
        }
        if (lane == 0) block.spec_stored_end = last_stored;
        return;
    }

    float carried_in = lossless ? values[first - 1] : block.stored_in;
    uint32_t window_leading = 255, window_trailing = 0; // macaque_v.rs:63-64
    uint8_t *__restrict__ dst = nullptr;
    uint64_t written = 0;
    uint32_t carry_bits = 0;
    bool first_byte_shared = false;
    if (MODE == LONG_WRITE) {
        window_leading = block.in_window & 0xffu;
        window_trailing = (block.in_window >> 8) & 0xffu;
        dst = targets.data[1] + targets.data_offsets[1][segment];
        if (first == 1 && lane < 4) dst[lane] = (uint8_t)(__float_as_uint(values[0]) >> (24 - 8 * lane)); // the first value: raw
        written = block.bit_offset >> 3;
        carry_bits = (uint32_t)(block.bit_offset & 7u);
        first_byte_shared = carry_bits != 0;
        for (int k = lane; k < GAP_BUFFER_WORDS; k += MDB_WAVE) buffer[k] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // (as k_fit_gap: equally spaced timestamps are checked while sizing; min / max in the order of the values)
    const ChunkTimestamps ts = chunk_timestamps(args.timestamps, item.chunk, args.chunk_offsets[item.chunk]);
    const uint32_t n = item.last - item.first + 1;
    bool regular = true;
    const int64_t expected_delta = (MODE == LONG_SPEC_SIZE && ts.ts && n >= 2) ? ts.ts[item.first + 1] - ts.ts[item.first] : 0;
    float min_value = nan32, max_value = nan32;
    uint32_t bits = 0;
    bool opened = false;
    for (uint32_t base = first, k = 0; base < first + count; base += MDB_WAVE, k++) {
        const uint32_t i = base + lane;
        const bool active = i < first + count;
        const int last_lane = (int)min((uint32_t)MDB_WAVE, first + count - base) - 1;
        GapBatchNote note;
        if (MODE == LONG_SPEC_SIZE) {
            note.prefix_bits = bits;
            note.windows = window_leading | (window_trailing << 8);
            note.stored_before = carried_in;
            if (ts.ts && active && ts.ts[item.first + i] - ts.ts[item.first + i - 1] != expected_delta) regular = false;
        }
        const GapBatch batch = gap_batch(eb, deviation, active, active ? values[i] : 0.0f, lane, last_lane, carried_in,
                                         window_leading, window_trailing);
        if (MODE == LONG_SPEC_SIZE) {
            const bool other = active && !batch.repeat;
            const uint32_t least_leading = wave_inclusive_scan(other ? batch.leading : 32u, lane, [](uint32_t a, uint32_t b) { return min(a, b); });
            const uint32_t least_trailing = wave_inclusive_scan(other ? batch.trailing : 32u, lane, [](uint32_t a, uint32_t b) { return min(a, b); });
            note.windows |= (read_lane(least_leading, MDB_WAVE - 1) << 16) | (read_lane(least_trailing, MDB_WAVE - 1) << 24);
            note.others = (uint32_t)__popcll(__ballot(other));
            if (lane == 0) la.batch_notes[block.batch_base + k] = note;
        }
        bits += batch.batch_bits;
        opened = opened || batch.any_opens;
        if (MODE == LONG_SPEC_SIZE) {
            const ValueRange range = wave_value_range(active, batch.stored);
            min_value = min_num(min_value, range.low);
            max_value = max_num(max_value, range.high);
        } else {
            if (active) {
                const uint32_t at = carry_bits + batch.before_bits;
                const uint32_t meaningful = 32u - batch.my_leading - batch.my_trailing;
                if (batch.repeat) {
                    gap_put(buffer, at, 0b10u, 2);
                } else if (batch.opens) {
                    gap_put(buffer, at, (0b11u << 11) | (batch.my_leading << 6) | meaningful, 13);
                    gap_put(buffer, at + 13, batch.x >> batch.my_trailing, meaningful);
                } else {
                    gap_put(buffer, at, 0u, 1);
                    gap_put(buffer, at + 1, batch.x >> batch.my_trailing, meaningful);
                }
            }
            carry_bits = wave_flush_bits_shared_first(buffer, GAP_BUFFER_WORDS, dst, &written, carry_bits + batch.batch_bits, lane,
                                                      &first_byte_shared);
        }
    }
    if (MODE == LONG_SPEC_SIZE) {
        const bool all_regular = __all(regular);
        if (lane == 0) {
            block.spec_bits = bits;
            block.spec_end_window = window_leading | (window_trailing << 8) | (opened ? 1u << 16 : 0u);
            block.min_value = min_value;
            block.max_value = max_value;
            block.regular = all_regular ? 1u : 0u;
        }
    } else if (carry_bits && lane == 0) {
        or_byte(dst + written, buffer[0] >> 24); // the byte the next block's codes begin in (the last block: zero bits behind)
    }
}

enum { STITCH_VALUES = 0, STITCH_WINDOWS = 1 };

// One wave per long segment, its blocks in order. STITCH_VALUES (lossy bounds): the value really stored in front of
// every block. STITCH_WINDOWS: every block's true window, bit offset and bit count; the segment's GapResult.
template <int STAGE>
__global__ __launch_bounds__(MDB_WAVE) void k_fit_long_stitch(FitArgs args, const SegItem *__restrict__ items, LongArgs la,
                                                              GapResult *__restrict__ results) {
    if (blockIdx.x >= la.counters->n_long) return;
    const uint32_t segment = la.long_ids[blockIdx.x];
    const SegItem item = items[segment];
    const int lane = threadIdx.x;
    const float *__restrict__ values = args.values + args.chunk_offsets[item.chunk] + item.first;
    const uint32_t n_codes = item.last - item.first;
    const uint32_t block_values = gap_block_values(n_codes, la.block_min_values);
    const uint32_t n_blocks = (n_codes + block_values - 1) / block_values;
    GapBlock *__restrict__ blocks = la.blocks + la.long_block_base[blockIdx.x];
    const mdb_error_bound eb = args.eb;
    const DeviationFactor deviation = deviation_factor(eb);
    if (STAGE == STITCH_VALUES) {
        float stored = values[0];
        for (uint32_t b = 0; b < n_blocks; b++) {
            const uint32_t first = blocks[b].first, count = blocks[b].count, batch_base = blocks[b].batch_base;
            const float spec_end = blocks[b].spec_stored_end;
            if (lane == 0) blocks[b].stored_in = stored;
            if (b == 0) { // (its wave began with the segment's first value: what it found is the stream's)
                stored = spec_end;
                continue;
            }
            // The true chain, a batch at a time, until it stores a value anew where the block's wave did too.
            for (uint32_t base = first, k = 0; base < first + count; base += MDB_WAVE, k++) {
                const uint32_t i = base + lane;
                const bool active = i < first + count;
                const int last_lane = (int)min((uint32_t)MDB_WAVE, first + count - base) - 1;
                unsigned long long anew = 0;
                (void)gap_value_chain(eb, deviation, active, active ? values[i] : 0.0f, lane, last_lane, stored, &anew);
                if (anew & la.batch_breakers[batch_base + k]) {
                    stored = spec_end;
                    break;
                }
            }
        }
        return;
    }

    uint32_t window_leading = 255, window_trailing = 0;
    unsigned long long offset = 32; // the first value: 32 raw bits
    float min_value = values[0], max_value = values[0];
    bool regular = true;
    for (uint32_t b = 0; b < n_blocks; b++) {
        const GapBlock block = blocks[b];
        const uint32_t in_window = window_leading | (window_trailing << 8);
        uint32_t bits = 0;
        bool joined = b == 0; // (the first block's wave began with the stream's own state)
        if (b == 0) {
            bits = block.spec_bits;
        } else {
            // Over the notes of the block's batches, 64 of them per step: the first batch in front of which the block's
            // wave held the true window, or in which a code does not fit the true window.
            const uint32_t n_batches = (block.count + MDB_WAVE - 1) / MDB_WAVE;
            const GapBatchNote *__restrict__ notes = la.batch_notes + block.batch_base;
            const uint32_t in_front = 33u - window_leading - window_trailing; // bits of a code that keeps the window (unused while it is empty)
            for (uint32_t k0 = 0; k0 < n_batches && !joined; k0 += MDB_WAVE) {
                const uint32_t k = k0 + lane;
                const bool have = k < n_batches;
                GapBatchNote note{};
                if (have) note = notes[k];
                const uint32_t in_batch = have ? min((uint32_t)MDB_WAVE, block.count - k * MDB_WAVE) : 0u;
                const bool same_window = have && (note.windows & 0xffffu) == in_window;
                const bool opens = have && note.others > 0 &&
                                   (((note.windows >> 16) & 0xffu) < window_leading || (note.windows >> 24) < window_trailing);
                const unsigned long long events = __ballot(same_window || opens);
                const int event = events ? __ffsll((long long)events) - 1 : MDB_WAVE;
                // (a batch in front of the event: its codes keep the window - two bits for a repeat, the window's for the others)
                const uint32_t kept = lane < event ? 2u * (in_batch - note.others) + (note.others ? in_front * note.others : 0u) : 0u;
                bits += read_lane(wave_inclusive_scan(kept, lane, [](uint32_t a, uint32_t b) { return a + b; }), MDB_WAVE - 1);
                if (event == MDB_WAVE) continue;
                joined = true;
                const GapBatchNote at = read_lane(note, event);
                const bool held_already = read_lane((uint32_t)same_window, event) != 0;
                if (held_already) {
                    bits += block.spec_bits - at.prefix_bits;
                } else {
                    // The batch in which the true chain opens a window, from its values.
                    const uint32_t base = block.first + (k0 + (uint32_t)event) * MDB_WAVE;
                    const uint32_t i = base + lane;
                    const bool active = i < block.first + block.count;
                    const int last_lane = (int)min((uint32_t)MDB_WAVE, block.first + block.count - base) - 1;
                    float carried_in = at.stored_before;
                    uint32_t leading_here = window_leading, trailing_here = window_trailing;
                    const GapBatch batch = gap_batch(eb, deviation, active, active ? values[i] : 0.0f, lane, last_lane, carried_in,
                                                     leading_here, trailing_here);
                    const uint32_t next = k0 + (uint32_t)event + 1;
                    const uint32_t behind = next < n_batches ? notes[next].prefix_bits : block.spec_bits;
                    bits += batch.batch_bits + (block.spec_bits - behind);
                }
            }
        }
        // Behind the block: once the two chains agree, the window the block's wave ended with; if they never did - no code
        // of the block opened a window for the true chain - the window in front.
        if (joined) {
            window_leading = block.spec_end_window & 0xffu;
            window_trailing = (block.spec_end_window >> 8) & 0xffu;
        }
        if (lane == 0) {
            blocks[b].in_window = in_window;
            blocks[b].bit_offset = offset;
            blocks[b].bits = bits;
        }
        offset += bits;
        min_value = min_num(min_value, block.min_value);
        max_value = max_num(max_value, block.max_value);
        regular = regular && block.regular != 0;
    }
    if (lane == 0) {
        GapResult result;
        result.values_bytes = (uint32_t)((offset + 7) >> 3);
        result.min_value = min_value;
        result.max_value = max_value;
        result.regular = regular ? 1u : 0u;
        results[segment] = result;
    }
}

// The bytes two blocks share (and the last byte of the stream, padded with zero bits): zeroed before the blocks OR
// their codes into them. One thread per block.
__global__ __launch_bounds__(256) void k_fit_long_clear(FitArgs args, LongArgs la, EncodeTargets targets) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= la.counters->n_blocks) return;
    if (args.targets_dev) targets = *args.targets_dev;
    const GapBlock &block = la.blocks[slot];
    const unsigned long long end = block.bit_offset + block.bits;
    if (end & 7ull) (targets.data[1] + targets.data_offsets[1][block.segment])[end >> 3] = 0;
}

// ---- k_fit_timestamps: one wave per segment of an irregular chunk -----------------------------------------
//
// compress_residual_timestamps (timestamps.rs:56-155) has no state that runs from code to code: the
// delta-of-delta of point j needs timestamps j-2, j-1 and j, nothing else. So the lanes of a wave take
// 64 points at a time - coalesced loads instead of one trip to memory per lane and point -, the code
// lengths add up to the size (and to bit offsets, in write mode, where the codes are ORed into an LDS
// bit buffer whose whole bytes are flushed, as k_fit_gap does for values). The same pass finds out
// whether the segment is equally spaced after all. Bit-identical to encode_irregular_timestamps.

constexpr uint32_t TS_WAVE_MIN_POINTS = 32;
constexpr int TS_BUFFER_WORDS = 144; // 64 codes x 69 bits + a carried partial byte

// Bits of the code for one delta-of-delta: `prefix` (prefix_bits <= 5... 16 for the short codes, which
// carry their payload with them) followed by `payload_bits` in {0, 32, 64} bits of payload.
__device__ __forceinline__ uint32_t delta_of_delta_code(int64_t dod, uint32_t *prefix, uint32_t *prefix_bits) {
    if (dod == 0) { *prefix = 0; *prefix_bits = 1; return 0; }
    if (dod >= -63 && dod <= 64) { *prefix = (0b10u << 7) | ((uint32_t)dod & 0x7fu); *prefix_bits = 9; return 0; }
    if (dod >= -255 && dod <= 256) { *prefix = (0b110u << 9) | ((uint32_t)dod & 0x1ffu); *prefix_bits = 12; return 0; }
    if (dod >= -2047 && dod <= 2048) { *prefix = (0b1110u << 12) | ((uint32_t)dod & 0xfffu); *prefix_bits = 16; return 0; }
    *prefix_bits = 5;
    if (dod >= -2147483647ll && dod <= 2147483648ll) { *prefix = 0b11110u; return 32; }
    *prefix = 0b11111u;
    return 64;
}

template <bool WRITE>
__global__ __launch_bounds__(MDB_WAVE) void k_fit_timestamps(FitArgs args, const SegItem *__restrict__ items,
                                                             TsResult *__restrict__ results, EncodeTargets targets) {
    __shared__ uint32_t buffer[TS_BUFFER_WORDS];
    const uint32_t segment = blockIdx.x;
    const SegItem item = items[segment];
    const uint32_t count = item.last - item.first + 1;
    if (count < TS_WAVE_MIN_POINTS) return;
    if (args.timestamps.chunk_irregular && !args.timestamps.chunk_irregular[item.chunk]) return; // O(1) elsewhere
    const int lane = threadIdx.x;
    const int64_t *__restrict__ t = args.timestamps.ts + args.chunk_offsets[item.chunk];
    const uint32_t a = item.first, b = item.last;
    // Point j (a < j < b) is stored as the code of (t[j] - t[j-1]) - (t[j-1] - t[j-2]); the first delta is
    // compared with zero (timestamps.rs:118-137).
    auto delta_of_delta = [&](uint32_t j) -> int64_t {
        const uint64_t current = (uint64_t)t[j], before = (uint64_t)t[j - 1];
        const uint64_t last_delta = j == a + 1 ? 0ull : before - (uint64_t)t[j - 2];
        return (int64_t)((current - before) - last_delta);
    };
    if (!WRITE) {
        const int64_t expected = t[a + 1] - t[a];
        bool regular = true;
        unsigned long long bits = 0;
        for (uint32_t j = a + 1 + lane; j <= b; j += MDB_WAVE) {
            if (t[j] - t[j - 1] != expected) regular = false;
            if (j < b) {
                uint32_t prefix, prefix_bits;
                bits += delta_of_delta_code(delta_of_delta(j), &prefix, &prefix_bits) + prefix_bits;
            }
        }
#pragma unroll
        for (int delta = MDB_WAVE / 2; delta > 0; delta >>= 1) {
            const uint32_t low = __shfl_down((uint32_t)bits, delta, MDB_WAVE);
            const uint32_t high = __shfl_down((uint32_t)(bits >> 32), delta, MDB_WAVE);
            bits += ((unsigned long long)high << 32) | low;
        }
        const bool all_regular = __all(regular);
        if (lane == 0) {
            TsResult result;
            result.regular = all_regular ? 1u : 0u;
            result.bytes = all_regular ? regular_length_bytes(count) : (uint32_t)((bits + 1 + 7) >> 3);
            results[segment] = result;
        }
        return;
    }
    const TsResult mine = results[segment];
    if (mine.bytes == 0xffffffffu || mine.regular || mine.bytes <= 12) return; // k_fit_encode writes those
    uint8_t *__restrict__ dst = targets.data[0] + targets.data_offsets[0][segment];
    for (int k = lane; k < TS_BUFFER_WORDS; k += MDB_WAVE) buffer[k] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) buffer[0] = 0x80000000u; // the flag "irregular" (timestamps.rs:116)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint32_t carry_bits = 1;
    uint64_t written = 0;
    for (uint32_t base = a + 1; base < b; base += MDB_WAVE) {
        const uint32_t j = base + lane;
        const bool active = j < b;
        uint32_t prefix = 0, prefix_bits = 0, payload_bits = 0;
        int64_t dod = 0;
        if (active) {
            dod = delta_of_delta(j);
            payload_bits = delta_of_delta_code(dod, &prefix, &prefix_bits);
        }
        const uint32_t code_bits = prefix_bits + payload_bits;
        const uint32_t inclusive = wave_inclusive_scan(code_bits, lane, [](uint32_t x, uint32_t y) { return x + y; });
        const uint32_t batch_bits = read_lane(inclusive, MDB_WAVE - 1);
        if (active) {
            uint32_t at = carry_bits + inclusive - code_bits;
            gap_put(buffer, at, prefix, prefix_bits);
            at += prefix_bits;
            if (payload_bits == 64) {
                gap_put(buffer, at, (uint32_t)((uint64_t)dod >> 32), 32);
                at += 32;
            }
            if (payload_bits != 0) gap_put(buffer, at, (uint32_t)dod, 32);
        }
        carry_bits = wave_flush_bits(buffer, TS_BUFFER_WORDS, dst, &written, carry_bits + batch_bits, lane);
    }
    // finish_with_one_bits (bits.rs:159-166): the last byte is filled up with ones.
    if (carry_bits && lane == 0) dst[written] = (uint8_t)((buffer[0] >> 24) | ((1u << (8u - carry_bits)) - 1u));
}

// A lane of the two kernels below encodes what its segment holds of MacaqueV codes one value after the other: a
// residual tail of up to 255 values (types.rs:224-262), or a gap shorter than the wave encoder's threshold. A wave is
// done when its longest is - and in noisy data the tails run from nothing to 255 values (the reference's acceptance
// series under 1 %: two thirds of them under 16 values, one in a hundred over 190), so that nearly every wave of 64
// consecutive segments waited for a long one. The workgroup therefore deals its 1 024 segments to its lanes by that
// length (a counting sort in LDS, classes of 16 values, the longest first): one wave takes the long ones, most take
// segments with nothing of the kind. Every segment's results go to its own places: the order changes nothing else.
constexpr int FIT_SEGMENT_THREADS = 1024;
constexpr int FIT_WORK_CLASSES = 17; // 16 x 16 values (longest first), then: nothing

__device__ __forceinline__ uint32_t serial_values_of(const FitArgs &args, const unsigned long long *record_base,
                                                     const ModelRec *records, const SegItem &item) {
    if (item.record == 0xffffffffu) return gap_goes_to_a_wave(args, item) ? 0u : item.last - item.first + 1;
    const uint32_t model_end = records[record_base[item.chunk] + item.record].end;
    return model_end < item.last ? item.last - model_end : 0u;
}

// The segment this lane takes (>= n_segments: none).
__device__ __forceinline__ uint64_t segment_by_serial_work(const FitArgs &args, const unsigned long long *record_base,
                                                           const ModelRec *records, const SegItem *items, uint64_t n_segments) {
    __shared__ uint32_t class_count[FIT_WORK_CLASSES], class_next[FIT_WORK_CLASSES];
    __shared__ uint16_t order[FIT_SEGMENT_THREADS];
    const uint64_t block_first = (uint64_t)blockIdx.x * FIT_SEGMENT_THREADS;
    const uint64_t mine = block_first + threadIdx.x;
    const int lane = threadIdx.x & (MDB_WAVE - 1);
    if (threadIdx.x < FIT_WORK_CLASSES) class_count[threadIdx.x] = 0;
    __syncthreads();
    int my_class = FIT_WORK_CLASSES - 1;
    if (mine < n_segments) {
        const uint32_t work = serial_values_of(args, record_base, records, items[mine]);
        if (work > 0) my_class = 15 - (int)min(work >> 4, 15u);
    }
    // (a place per wave and class with one LDS atomic: the lanes of a class are counted with a ballot)
    uint32_t rank_in_wave = 0, place = 0;
#pragma unroll 1
    for (int c = 0; c < FIT_WORK_CLASSES; c++) {
        const unsigned long long lanes = __ballot(my_class == c);
        if (lanes == 0) continue;
        uint32_t first = 0;
        if (lane == 0) first = atomicAdd(&class_count[c], (uint32_t)__popcll(lanes));
        first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
        if (my_class == c) {
            place = first;
            rank_in_wave = (uint32_t)__popcll(lanes & ((1ull << lane) - 1ull));
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t running = 0;
        for (int c = 0; c < FIT_WORK_CLASSES; c++) {
            class_next[c] = running;
            running += class_count[c];
        }
    }
    __syncthreads();
    order[class_next[my_class] + place + rank_in_wave] = (uint16_t)threadIdx.x;
    __syncthreads();
    return block_first + order[threadIdx.x];
}

__global__ __launch_bounds__(FIT_SEGMENT_THREADS) void k_fit_size(FitArgs args, const unsigned long long *__restrict__ record_base,
                                                  const ModelRec *__restrict__ records,
                                                  const SegItem *__restrict__ items, uint64_t n_segments,
                                                  SegSizes *__restrict__ sizes) {
    if (args.n_segments_dev) n_segments = *args.n_segments_dev;
    if ((uint64_t)blockIdx.x * FIT_SEGMENT_THREADS >= n_segments) return; // (a launch over an upper bound)
    const uint64_t segment = segment_by_serial_work(args, record_base, records, items, n_segments);
    if (segment >= n_segments) return;
    SegSizes s = {0, 0, 0, 0};
    process_segment<false>(args, record_base, records, items[segment], segment, &s, nullptr);
    sizes[segment] = s;
}

__global__ __launch_bounds__(FIT_SEGMENT_THREADS) void k_fit_encode(FitArgs args, const unsigned long long *__restrict__ record_base,
                                                    const ModelRec *__restrict__ records,
                                                    const SegItem *__restrict__ items, uint64_t n_segments,
                                                    const SegSizes *__restrict__ sizes, EncodeTargets targets) {
    if (args.n_segments_dev) n_segments = *args.n_segments_dev;
    if ((uint64_t)blockIdx.x * FIT_SEGMENT_THREADS >= n_segments) return;
    if (args.targets_dev) targets = *args.targets_dev;
    const uint64_t segment = segment_by_serial_work(args, record_base, records, items, n_segments);
    if (segment >= n_segments) return;
    SegSizes s = sizes[segment];
    process_segment<true>(args, record_base, records, items[segment], segment, &s, &targets);
}

// ---- a handful of chunks: no question to the device between the upload and the download -----------------------
//
// The reference's server compresses ONE finished buffer of 65 536 points per call (uncompressed_data_manager.rs:
// 530-596, storage/mod.rs:58). The general driver below asks the device eight times how much it has made so far
// (records, pieces, segments, gaps, three payload sizes, the tables) to size what comes next, allocates the batch
// with hipMalloc and downloads it column by column: 2.5 ms for a buffer one CPU thread fits in 0.8. Here everything
// is sized by upper bounds the host can know (a model has at least 8 points: at most n / 8 + n / 256 + 2 segments per
// chunk; a MacaqueV code at most 45 bits), the kernels read the counts the kernels before them left in device memory,
// the last ones write all columns packed behind a header into one block, and the host copies that block once.
// Models: one wave per PIECE of a chunk (k_fit_models_wave<.., PIECES>), so that a chunk's latency is a piece's.
struct SmallHeader {
    unsigned long long n_segments;
    unsigned long long blob_bytes;   // header included
    unsigned long long offsets[13];  // into the block: type, start, end, min, max, error, chunk, views x 3, data x 3
    unsigned long long data_bytes[3];
    unsigned int error;              // ERR_* of the fit, or SMALL_OVERFLOW
    unsigned int pad;
};
constexpr unsigned int SMALL_OVERFLOW = 0x40000000u;
constexpr uint64_t SMALL_HEADER_BYTES = 256;
static_assert(sizeof(SmallHeader) <= SMALL_HEADER_BYTES, "the header's place in the block");

// Segments per chunk -> first segment of every chunk, and the call's number of segments.
__global__ __launch_bounds__(64) void k_small_segments(const ChunkPlan *__restrict__ plans, uint64_t n_chunks,
                                                       unsigned long long *__restrict__ segment_base,
                                                       unsigned long long *__restrict__ n_segments) {
    if (threadIdx.x != 0) return;
    unsigned long long running = 0;
    for (uint64_t c = 0; c < n_chunks; c++) {
        segment_base[c] = running;
        running += plans[c].n_segments;
    }
    segment_base[n_chunks] = running;
    *n_segments = running;
}

// Where every out-of-line payload goes (three exclusive scans over the segments), where every column goes in the
// block, and the header the host reads.
__global__ __launch_bounds__(1024) void k_small_layout(const SegSizes *__restrict__ sizes,
                                                       unsigned long long *__restrict__ n_segments_dev,
                                                       unsigned long long *__restrict__ data_offsets, uint64_t offsets_stride,
                                                       const unsigned long long *__restrict__ zero_base, uint8_t *__restrict__ blob,
                                                       uint64_t blob_capacity, const unsigned int *__restrict__ fit_error,
                                                       EncodeTargets *__restrict__ targets_out) {
    __shared__ uint64_t lds[17];
    __shared__ unsigned long long totals[3];
    const uint64_t n = *n_segments_dev;
    for (int c = 0; c < 3; c++) {
        unsigned long long *out = data_offsets + c * offsets_stride;
        const OutOfLineBytes bytes{sizes, c};
        uint64_t carry = 0;
        for (uint64_t first = 0; first < n; first += 1024) {
            const uint64_t i = first + threadIdx.x;
            const uint64_t mine = i < n ? bytes(i) : 0;
            uint64_t total;
            const uint64_t before = block_exclusive_scan_u64(mine, lds, &total);
            if (i < n) out[i] = carry + before;
            carry += total;
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            out[n] = carry;
            totals[c] = carry;
        }
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    SmallHeader header{};
    uint64_t cursor = SMALL_HEADER_BYTES;
    auto carve = [&](int k, uint64_t bytes) {
        header.offsets[k] = cursor;
        cursor = (cursor + bytes + 63) / 64 * 64;
    };
    carve(0, n); carve(1, 8 * n); carve(2, 8 * n); carve(3, 4 * n); carve(4, 4 * n); carve(5, 4 * n); carve(6, 4 * n);
    for (int c = 0; c < 3; c++) carve(7 + c, 16 * n);
    for (int c = 0; c < 3; c++) carve(10 + c, totals[c]);
    header.n_segments = n;
    header.blob_bytes = cursor;
    for (int c = 0; c < 3; c++) header.data_bytes[c] = totals[c];
    header.error = *fit_error | (cursor > blob_capacity ? SMALL_OVERFLOW : 0u);
    EncodeTargets t{};
    if (cursor <= blob_capacity) {
        t.model_type_id = reinterpret_cast<int8_t *>(blob + header.offsets[0]);
        t.start_time = reinterpret_cast<int64_t *>(blob + header.offsets[1]);
        t.end_time = reinterpret_cast<int64_t *>(blob + header.offsets[2]);
        t.min_value = reinterpret_cast<float *>(blob + header.offsets[3]);
        t.max_value = reinterpret_cast<float *>(blob + header.offsets[4]);
        t.error = reinterpret_cast<float *>(blob + header.offsets[5]);
        t.chunk_index = reinterpret_cast<uint32_t *>(blob + header.offsets[6]);
        for (int c = 0; c < 3; c++) {
            t.views[c] = reinterpret_cast<uint4 *>(blob + header.offsets[7 + c]);
            t.data[c] = blob + header.offsets[10 + c];
            t.data_offsets[c] = data_offsets + c * offsets_stride;
            t.data_bases[c] = zero_base; // one data buffer per column
            t.n_data_buffers[c] = 1;
        }
    } else {
        header.n_segments = 0;
        *n_segments_dev = 0; // (the encoders behind this kernel find nothing to do)
    }
    *targets_out = t;
    *reinterpret_cast<SmallHeader *>(blob) = header;
}

// ---- host driver ----------------------------------------------------------------------------------------------

static bool valid_error_bound(mdb_error_bound eb) { // crates/modelardb_types/src/types.rs:312-334
    if (eb.kind == MDB_EB_LOSSLESS) return true;
    if (eb.kind == MDB_EB_ABSOLUTE) return std::isfinite(eb.value) && eb.value > 0.0f;
    if (eb.kind == MDB_EB_RELATIVE) return 0.0f < eb.value && eb.value <= 100.0f;
    return false;
}

// MDB_FIT_GAP_MIN_VALUES: "off" keeps one lane per MacaqueV-only segment, a number sets the length
// from which a lossless one gets a wave of its own.
static uint32_t gap_min_values_setting() {
    if (const char *text = option_text("MDB_FIT_GAP_MIN_VALUES")) {
        if (std::strcmp(text, "off") == 0) return 0xffffffffu;
        const long long value = std::atoll(text);
        if (value >= 2) return (uint32_t)std::min<long long>(value, 0x7fffffff);
    }
    return GAP_DEFAULT_MIN_VALUES;
}

// MDB_FIT_GAP_ONCE: 0 - the waves' MacaqueV-only segments are sized by one kernel and encoded again by another; 1 - they
// are encoded once into staging places and copied (GAP_STAGE); not set: once under a lossy bound, twice under a lossless
// one (see where it is asked).
static bool gap_once_setting(const mdb_error_bound &eb) {
    const char *text = option_text("MDB_FIT_GAP_ONCE");
    if (text && std::strcmp(text, "0") == 0) return false;
    if (text && std::strcmp(text, "1") == 0) return true;
    return eb.kind != MDB_EB_LOSSLESS;
}

// MDB_FIT_GAP_LONG_MIN_VALUES: "off" keeps one wave per MacaqueV-only segment however long it is, a number sets the
// length from which one is cut into blocks with a wave each (k_fit_long*). MDB_FIT_GAP_BLOCK_VALUES: the least number
// of values per block (rounded up to whole batches of 64; the tests cut short streams into many blocks with it).
static uint32_t gap_long_min_values_setting() {
    if (const char *text = option_text("MDB_FIT_GAP_LONG_MIN_VALUES")) {
        if (std::strcmp(text, "off") == 0) return 0xffffffffu;
        const long long value = std::atoll(text);
        if (value >= 2) return (uint32_t)std::min<long long>(value, 0x7fffffff);
    }
    return GAP_LONG_DEFAULT_MIN_VALUES;
}
static uint32_t gap_block_values_setting() {
    if (const char *text = option_text("MDB_FIT_GAP_BLOCK_VALUES")) {
        const long long value = std::atoll(text);
        if (value >= 1) return (uint32_t)align_up((uint64_t)std::min<long long>(value, 1ll << 24), MDB_WAVE);
    }
    return GAP_BLOCK_DEFAULT_MIN_VALUES;
}

// The launches that size the long segments (behind k_fit_long_select and a look at its counters, or over upper bounds):
// block headers, [lossy bounds: what every block stores from nothing, then the true value in front of every block,]
// every block's codes from an empty window, then the true window, offset and bits of every block and the segments'
// GapResults.
static void launch_long_sizing(mdb_ctx *ctx, hipStream_t stream, const FitArgs &args, const SegItem *items, const LongArgs &la,
                               GapResult *gap_results, uint32_t n_long, uint32_t n_blocks) {
    LaunchTimer timer(ctx, "k_fit_long_size");
    hipLaunchKernelGGL(k_fit_long_blocks, dim3(n_long), dim3(MDB_WAVE), 0, stream, items, la);
    if (args.eb.kind != MDB_EB_LOSSLESS) {
        hipLaunchKernelGGL(k_fit_long<LONG_SPEC_VALUES>, dim3(n_blocks), dim3(MDB_WAVE), 0, stream, args, items, la, EncodeTargets{});
        hipLaunchKernelGGL(k_fit_long_stitch<STITCH_VALUES>, dim3(n_long), dim3(MDB_WAVE), 0, stream, args, items, la, gap_results);
    }
    hipLaunchKernelGGL(k_fit_long<LONG_SPEC_SIZE>, dim3(n_blocks), dim3(MDB_WAVE), 0, stream, args, items, la, EncodeTargets{});
    hipLaunchKernelGGL(k_fit_long_stitch<STITCH_WINDOWS>, dim3(n_long), dim3(MDB_WAVE), 0, stream, args, items, la, gap_results);
}

static void launch_long_encode(mdb_ctx *ctx, hipStream_t stream, const FitArgs &args, const SegItem *items, const LongArgs &la,
                               const EncodeTargets &targets, uint32_t n_blocks) {
    LaunchTimer timer(ctx, "k_fit_long_encode");
    hipLaunchKernelGGL(k_fit_long_clear, dim3((n_blocks + 255) / 256), dim3(256), 0, stream, args, la, targets);
    hipLaunchKernelGGL(k_fit_long<LONG_WRITE>, dim3(n_blocks), dim3(MDB_WAVE), 0, stream, args, items, la, targets);
}

// MDB_FIT_FAST=0: the plain forms of PMC-Mean and Swing in k_fit_models even where the fast ones apply.
static bool fit_fast_setting() {
    const char *text = option_text("MDB_FIT_FAST");
    return !(text && std::strcmp(text, "0") == 0);
}

// From how many bytes of payloads on a BinaryView column begins another data buffer (at most 1 GiB, so
// that a buffer with the payload that ends it stays below the 2 GiB a view can address).
constexpr uint64_t MAX_DATA_BUFFERS = 4096;
static uint64_t data_buffer_bytes_setting() {
    if (const char *text = option_text("MDB_FIT_DATA_BUFFER_BYTES")) {
        const long long value = std::atoll(text);
        if (value >= 16) return (uint64_t)std::min<long long>(value, 1ll << 30);
    }
    return 1ull << 30;
}

// MDB_FIT_LEAN=0: k_fit_models (its fast form) even where k_fit_models_lean applies.
static bool fit_lean_setting() {
    const char *text = option_text("MDB_FIT_LEAN");
    return !(text && std::strcmp(text, "0") == 0);
}

// Points per piece for split mode, 0 = one lane per chunk. Split when the call has too few chunks
// to give every SIMD a couple of waves and the chunks are long enough to be worth cutting.
// MDB_FIT_PIECE_POINTS overrides: 1 = never split, N >= 64 = always split into pieces of N points.
// MDB_FIT_WAVE: 0 = never k_fit_models_wave; 1 = wherever it applies, every chunk to its end; 2 = wherever split
// mode is possible (also with a forced piece size), chunks with short models left to split mode; otherwise 2's
// behaviour for the calls the library would have given to split mode by itself.
static int fit_wave_setting() {
    const char *setting = option_text("MDB_FIT_WAVE");
    if (!setting || !*setting) return -1;
    const int value = std::atoi(setting);
    return value == 1 || value == 2 ? value : 0;
}

// One wave per chunk takes about 2 200 cycles per 64 points with every SIMD busy, one lane per chunk about 800 per
// point with one wave per 64 chunks: beyond some 24 000 chunks the latter has enough waves to be the faster one.
// (24 576 on the MI355X's 256 compute units: 96 chunks - waves - per compute unit)
static uint64_t fit_wave_max_chunks(const mdb_ctx *ctx) { return 96ull * (uint64_t)std::max(ctx->compute_units, 1); }

static uint32_t fit_wave_number(const char *name, uint32_t otherwise) {
    const char *setting = option_text(name);
    if (!setting || !*setting) return otherwise;
    return (uint32_t)std::min<long long>(std::max<long long>(std::atoll(setting), 1), 1ll << 30);
}

constexpr uint32_t FIT_LEFT_PIECE_POINTS = 512; // pieces of the chunks k_fit_models_wave leaves to split mode
// ... and of all chunks when the probe sends them there untried (the mixed series at 1 % with k_fit_reject_flags, whole call:
// 39.1 ms in pieces of 256 points, 33.3 / 32.3 / 31.9 / 32.8 / 33.0 / 34.4 in pieces of 512 / 768 / 1 024 / 1 536 / 2 048 / 3 072)
constexpr uint32_t FIT_UNTRIED_PIECE_POINTS = 1024;
constexpr unsigned long long FIT_FLAGS_WORTH_ONE_IN = 8; // the fitter looks at k_fit_reject_flags' bits if one point in so many has its bit set

static uint32_t split_piece_points(const mdb_ctx *ctx, uint64_t n_chunks, uint64_t total_points) {
    if (const char *forced = option_text("MDB_FIT_PIECE_POINTS")) {
        const long long value = std::atoll(forced);
        if (value == 1) return 0;
        if (value >= 64) return (uint32_t)std::min<long long>(value, 1 << 30);
    }
    // (MDB_FIT_SPLIT_WAVES_PER_SIMD: the waves of pieces asked for per SIMD, 2 unless set)
    const uint64_t waves_per_simd = fit_wave_number("MDB_FIT_SPLIT_WAVES_PER_SIMD", 2);
    const uint64_t target_lanes = (uint64_t)std::max(ctx->compute_units, 1) * 4 * MDB_WAVE * std::max<uint64_t>(waves_per_simd, 1);
    if (n_chunks == 0 || n_chunks >= target_lanes / 2) return 0;
    if (total_points * 12 > (48ull << 30)) return 0; // the per-point table would be too large
    const uint64_t piece = std::max<uint64_t>(512, align_up(total_points / target_lanes + 1, 64));
    if (total_points / n_chunks < 2 * piece) return 0; // chunks too short to gain anything
    return (uint32_t)piece;
}

// chunk_first / chunk_interval (device arrays, may be nullptr): the caller has looked at the timestamps of every
// chunk itself, found them equally spaced and passes what k_fit_regular would have found (ts is nullptr then).
int compress_chunks_dev_locked(mdb_ctx *ctx, const int64_t *ts, const float *values,
                               const uint64_t *chunk_offsets, uint64_t n_chunks, mdb_error_bound eb,
                               int64_t regular_start, int64_t regular_interval,
                               const uint64_t *series_first_index, mdb_segments_owned **out,
                               const long long *chunk_first = nullptr, const long long *chunk_interval = nullptr) {
    if (!valid_error_bound(eb)) return fail("Invalid error bound.");
    if (n_chunks > 0xfffffff0ull) return fail("Too many chunks in one call.");
    if (!ts && !chunk_first && regular_interval <= 0)
        return fail("Either timestamps or a positive regular_interval must be given.");
    FitArgs args;
    args.values = values;
    args.timestamps = {ts, regular_start, regular_interval,
                       reinterpret_cast<const unsigned long long *>(series_first_index), chunk_first, chunk_interval,
                       nullptr};
    args.chunk_offsets = reinterpret_cast<const unsigned long long *>(chunk_offsets);
    args.n_chunks = n_chunks;
    args.eb = eb;
    args.gap_min_values = 0xffffffffu;
    args.gap_results = nullptr;
    args.ts_results = nullptr;

    OwnedSegments *owned = new OwnedSegments();
    owned->device = ctx->device;
    owned->host_allocs.resize(3);
    auto release = [&]() {
        mail_drop(ctx); // (small reads that were under way: their destinations are this frame's)
        for (void *p : owned->device_allocs) (void)hipFree(p);
        delete owned;
    };
#define FIT_CHECK(expr)                                                                            \
    do {                                                                                           \
        const hipError_t fit_err_ = (expr);                                                        \
        if (fit_err_ != hipSuccess) {                                                              \
            release();                                                                             \
            return fail(std::string(#expr) + " failed: " + hipGetErrorString(fit_err_));           \
        }                                                                                          \
    } while (0)
#define FIT_TRY(expr)                                                                              \
    do {                                                                                           \
        if (expr) {                                                                                \
            release();                                                                             \
            return 1;                                                                              \
        }                                                                                          \
    } while (0)

    void *p = nullptr;
    // Scratch: A record_base (n_chunks+1 u64) + block sums, B model records, C chunk plans,
    // D segment_base (n_chunks+1 u64), E items, F sizes + 3 x offsets.
    const uint64_t sums_bytes = scan_block_sums_bytes(std::max<uint64_t>(n_chunks, 1));
    FIT_TRY(scratch_reserve(ctx, SCRATCH_FIT_A, (n_chunks + 1) * 8 + sums_bytes + 64, &p));
    unsigned long long *record_base = static_cast<unsigned long long *>(p);
    unsigned long long *block_sums = record_base + n_chunks + 1;
    FIT_TRY(scratch_reserve(ctx, SCRATCH_HEADER, sizeof(unsigned int) * 64, &p));
    unsigned int *error_flag = static_cast<unsigned int *>(p);
    FIT_CHECK(hipMemsetAsync(error_flag, 0, 4, ctx->stream));

    unsigned long long total_records = 0;
    unsigned long long n_segments = 0;
    if (n_chunks > 0) {
        unsigned long long points_end = 0;
        // Materialised timestamps: are they all equally spaced? (answer read with the sync below)
        unsigned int regular_verdict[2] = {1u, 1u}; // irregular chunks, chunks with a timestamp beyond +-2^52
        long long *chunk_first = nullptr, *chunk_interval = nullptr;
        unsigned int *chunk_irregular = nullptr;
        if (ts && n_chunks <= 0x7fffffffull) {
            FIT_TRY(scratch_reserve(ctx, SCRATCH_FIT_REGULAR, n_chunks * 20 + 128, &p));
            chunk_first = static_cast<long long *>(p);
            chunk_interval = chunk_first + n_chunks;
            chunk_irregular = reinterpret_cast<unsigned int *>(chunk_interval + n_chunks);
            unsigned int *counter = chunk_irregular + n_chunks; // [0] irregular chunks, [1] chunks beyond +-2^52
            FIT_CHECK(hipMemsetAsync(counter, 0, 8, ctx->stream));
            {
                LaunchTimer timer(ctx, "k_fit_regular");
                hipLaunchKernelGGL(k_fit_regular, dim3((uint32_t)n_chunks), dim3(256), 0, ctx->stream, ts,
                                   args.chunk_offsets, n_chunks, chunk_first, chunk_interval, chunk_irregular,
                                   counter);
            }
            FIT_CHECK(mail_read(ctx, regular_verdict, counter, 8));
        }
        FIT_TRY(device_exclusive_scan(ctx, RecordCapacity{args.chunk_offsets}, n_chunks, record_base,
                                      block_sums, "k_fit_scan"));
        FIT_CHECK(mail_read(ctx, &total_records, record_base + n_chunks, 8));
        FIT_CHECK(mail_read(ctx, &points_end, args.chunk_offsets + n_chunks, 8));
        FIT_CHECK(mail_sync(ctx));
        const unsigned int n_irregular_chunks = regular_verdict[0];
        if (ts && n_irregular_chunks == 0) {
            // Every chunk is regular: from here on timestamps are computed, not loaded.
            args.timestamps.ts = nullptr;
            args.timestamps.chunk_first = chunk_first;
            args.timestamps.chunk_interval = chunk_interval;
            ts = nullptr;
        } else if (ts) {
            args.timestamps.chunk_irregular = chunk_irregular;
        }
        // Regular timestamps (given as such, or found to be): are they all exact as f64? Then the
        // fast forms of the two fitters apply (MDB_FIT_FAST=0 keeps the plain ones: A/B and tests).
        bool fast = false;
        if (!ts && fit_fast_setting()) {
            unsigned int inexact = 0;
            FIT_CHECK(hipMemsetAsync(error_flag + 1, 0, 4, ctx->stream));
            hipLaunchKernelGGL(k_fit_exact_double_timestamps, dim3((uint32_t)((n_chunks + 255) / 256)), dim3(256), 0,
                               ctx->stream, args.timestamps, args.chunk_offsets, n_chunks, error_flag + 1);
            FIT_CHECK(mail_read(ctx, &inexact, error_flag + 1, 4));
            FIT_CHECK(mail_sync(ctx));
            fast = inexact == 0;
        }
        const bool lean = fast && fit_lean_setting();
        // Timestamps that have to be loaded (some chunk is irregular), all of them exact as f64: the
        // straight-line fitter with a ring of timestamps.
        const bool lean_ts = ts && chunk_irregular && regular_verdict[1] == 0 && fit_fast_setting() && fit_lean_setting();
        uint32_t piece_points = split_piece_points(ctx, n_chunks, points_end);
        FIT_TRY(scratch_reserve(ctx, SCRATCH_FIT_B, total_records * sizeof(ModelRec), &p));
        ModelRec *records = static_cast<ModelRec *>(p);
        FIT_TRY(scratch_reserve(ctx, SCRATCH_FIT_C, n_chunks * sizeof(ChunkPlan), &p));
        ChunkPlan *plans = static_cast<ChunkPlan *>(p);
        FIT_TRY(scratch_reserve(ctx, SCRATCH_FIT_D, (n_chunks + 1) * 8, &p));
        unsigned long long *segment_base = static_cast<unsigned long long *>(p);
        // Few chunks: one wave per chunk, which leaves the chunks whose models turn out short to split mode
        // (forced piece sizes are the tests' way to ask for split mode alone; MDB_FIT_WAVE=1: every chunk, to the end).
        const int wave_setting = fit_wave_setting();
        const bool exact_loaded_timestamps = ts && chunk_irregular && regular_verdict[1] == 0 && fit_fast_setting();
        // (Under a LOSSLESS bound the wave kernel decides by equality - 64 start points a round, a comparison per block of a
        // model - and is the fastest fitter for any number of chunks: 64 000 chunks of 4 000 points 3.9 ms against split
        // mode's 20.2 on the mixed series, 2.7 against 7.9 on noise (profiles/r06/fit_lossless.csv); only chunks longer than
        // MDB_FIT_WAVE_MAX_CHUNK_POINTS are still left to split mode, where there is one.)
        const bool wave = ((fast && !ts) || exact_loaded_timestamps) && fit_lean_setting() && wave_setting != 0 &&
                          (wave_setting == 1 ||
                           (eb.kind == MDB_EB_LOSSLESS && !option_text("MDB_FIT_PIECE_POINTS")) ||
                           (piece_points != 0 &&
                            (wave_setting == 2 || (n_chunks <= fit_wave_max_chunks(ctx) && !option_text("MDB_FIT_PIECE_POINTS")))));
        bool split_mode = !wave && piece_points != 0;
        const unsigned int *split_only = nullptr; // (per chunk: 1 = left to split mode by k_fit_models_wave)
        if (wave) {
            WaveLeave leave{};
            if (wave_setting != 1 && piece_points != 0) {
                FIT_TRY(scratch_reserve(ctx, SCRATCH_FIT_WAVE, (n_chunks + 4) * 4, &p));
                leave.chunk_left = static_cast<unsigned int *>(p);
                leave.n_left = leave.chunk_left + n_chunks;
                leave.window_points = fit_wave_number("MDB_FIT_WAVE_WINDOW_POINTS", 1024);
                leave.max_chunk_points = fit_wave_number("MDB_FIT_WAVE_MAX_CHUNK_POINTS", 262144);
                // (2 300 cycles per step against split mode's 107 per point under a relative or absolute bound:
                // k_fit_models_lean. Under a lossless one, on the bench's mixed series, every threshold above 3 only
                // moved chunks to the slower side: 41 ms at 3, 54 at 12, 81 at 45 - measured with k_fit_models as
                // split mode's fitter; the lean one that has replaced it is a fifth faster, which does not turn that.)
                leave.points_per_step = fit_wave_number("MDB_FIT_WAVE_POINTS_PER_STEP", eb.kind == MDB_EB_LOSSLESS ? 3 : 20);
                FIT_CHECK(hipMemsetAsync(leave.n_left, 0, 16, ctx->stream));
            }
            const bool count_steps = option_text("MDB_FIT_DEBUG") != nullptr;
            if (count_steps) {
                FIT_CHECK(hipMalloc(reinterpret_cast<void **>(&leave.counts), WAVE_COUNTS * 8));
                FIT_CHECK(hipMemsetAsync(leave.counts, 0, WAVE_COUNTS * 8, ctx->stream));
            }
            // The probe (lossy bounds, calls whose chunks may leave): a wave for every 17th chunk (every chunk of a call of a
            // few hundred) fits one window of 1 024 points somewhere in it and says whether that pace is one to leave a chunk
            // at. The call then goes ONE way. A quarter of the windows slow - models short in many places, the mixed series
            // at 1 %: 36 % - and every chunk goes to split mode untried: what waves spend on chunks before they leave them
            // was a third of that call (14.4 of 44 ms). Fewer, and every chunk keeps its wave to its end: a chunk that leaves
            // for ONE rough window hands split mode its long models too, which speculative pieces fit once each - smooth
            // series with a tenth of their windows rough took 79 ms that way, 18.8 with a wave per chunk to the end, and split
            // mode for all of it 62 (profiles/r06/probe_rough_smooth.txt: the two ways cross where a quarter of the windows
            // are slow; leaving chunk by chunk was the slowest or near it in every row). A tenth of a millisecond and one
            // wait for the host (MDB_FIT_WAVE_PROBE=0: no probe, chunks leave one by one as before).
            const char *probe_setting = option_text("MDB_FIT_WAVE_PROBE");
            const uint32_t probe_stride = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(n_chunks / 256, 1), 17);
            const bool probing = leave.chunk_left && eb.kind != MDB_EB_LOSSLESS && n_chunks >= 64u &&
                                 !(probe_setting && std::strcmp(probe_setting, "0") == 0);
            bool untried = false; // every chunk is left to split mode without a wave having fitted it
            auto launch_waves = [&](uint32_t stride, uint64_t grid) {
                leave.probe_stride = stride;
                LaunchTimer timer(ctx, stride ? "k_fit_models_wave_probe" : "k_fit_models_wave");
#define MDB_LAUNCH_WAVE(KIND, HAS_TS)                                                                                    \
    hipLaunchKernelGGL((k_fit_models_wave<KIND, HAS_TS>), dim3((uint32_t)grid), dim3(MDB_WAVE), 0, ctx->stream, args, \
                       leave, SplitArgs{}, record_base, records, plans, error_flag)
                if (eb.kind == MDB_EB_RELATIVE) {
                    if (ts) MDB_LAUNCH_WAVE(MDB_EB_RELATIVE, true);
                    else MDB_LAUNCH_WAVE(MDB_EB_RELATIVE, false);
                } else if (eb.kind == MDB_EB_ABSOLUTE) {
                    if (ts) MDB_LAUNCH_WAVE(MDB_EB_ABSOLUTE, true);
                    else MDB_LAUNCH_WAVE(MDB_EB_ABSOLUTE, false);
                } else {
                    if (ts) MDB_LAUNCH_WAVE(MDB_EB_LOSSLESS, true);
                    else MDB_LAUNCH_WAVE(MDB_EB_LOSSLESS, false);
                }
#undef MDB_LAUNCH_WAVE
            };
            if (probing) {
                launch_waves(probe_stride, (n_chunks + probe_stride - 1) / probe_stride);
                unsigned int seen[4] = {0, 0, 0, 0}; // windows at a pace to leave at, windows looked at, windows out of which a model ran for another window's length
                FIT_CHECK(mail_read(ctx, seen, leave.n_left, 16));
                FIT_CHECK(mail_sync(ctx));
                if (option_text("MDB_FIT_DEBUG"))
                    std::fprintf(stderr, "[fit] k_fit_models_wave_probe: %u of %u windows at a pace to leave the chunk at, %u in models of more than a window's length\n",
                                 seen[0], seen[1], seen[2]);
                if (seen[1] >= 32u) {
                    untried = 4ull * seen[0] >= seen[1];
                    if (!untried) leave.points_per_step = 0; // (no pace is one to leave at: only a chunk's length is)
                }
                FIT_CHECK(hipMemsetAsync(leave.n_left, 0, 16, ctx->stream));
            }
            if (untried) {
                // (every chunk counts as left: any non-zero word says so)
                FIT_CHECK(hipMemsetAsync(leave.chunk_left, 1, n_chunks * 4, ctx->stream));
            } else {
                launch_waves(0, n_chunks);
            }
            if (leave.counts) {
                unsigned long long counts[WAVE_COUNTS] = {};
                FIT_CHECK(mail_read(ctx, counts, leave.counts, sizeof(counts)));
                FIT_CHECK(mail_sync(ctx));
                FIT_CHECK(hipFree(leave.counts));
                std::fprintf(stderr, "[fit] k_fit_models_wave: %llu chunks, %llu points: %llu models, %llu rejected start points, "
                             "%llu passes over 64 start points, %llu blocks (%llu with PMC-Mean alive, %llu in which no bound of Swing moves), %llu Swing scans, %llu models by one lane\n",
                             (unsigned long long)n_chunks, (unsigned long long)points_end, counts[WAVE_MODELS], counts[WAVE_REJECTED],
                             counts[WAVE_START_PASSES], counts[WAVE_BLOCKS], counts[WAVE_PMC_BLOCKS], counts[WAVE_QUIET_BLOCKS],
                             counts[WAVE_SWING_SCANS], counts[WAVE_BY_ONE_LANE]);
#ifdef MDB_WAVE_TIMING
                std::fprintf(stderr, "[fit] k_fit_models_wave cycles: total %llu; first stage %llu (%llu rounds), second stage %llu, block load %llu, "
                             "PMC-Mean %llu, Swing %llu, finish %llu, flush %llu\n", counts[WAVE_T_TOTAL], counts[WAVE_T_FIRST_STAGE],
                             counts[WAVE_N_FIRST_STAGE], counts[WAVE_T_SECOND_STAGE], counts[WAVE_T_BLOCK_LOAD], counts[WAVE_T_PMC],
                             counts[WAVE_T_SWING], counts[WAVE_T_FINISH], counts[WAVE_T_FLUSH]);
#endif
            }
            if (leave.chunk_left) {
                unsigned int left[2] = {0, 0};
                FIT_CHECK(mail_read(ctx, left, leave.n_left, 8));
                FIT_CHECK(mail_sync(ctx));
                if (untried) { // (all of them; "some for their length alone" if the average chunk is anywhere near that length)
                    left[0] = (unsigned int)n_chunks;
                    left[1] = points_end / n_chunks > leave.max_chunk_points / 2u ? 1u : 0u;
                }
                const unsigned int n_left = left[0];
                if (n_left > 0) {
                    split_mode = true;
                    split_only = leave.chunk_left;
                    // (the chunks that are left are the ones whose models are short: chains that start at different
                    // points of such a chunk meet within a few models, so short pieces cost little twice-fitted ground and
                    // give every SIMD many waves - the mixed series' left half in pieces of 3 712 / 1 280 / 512 / 256 / 128
                    // points: 21.0 / 17.0 / 15.5 / 15.5 / 17.8 ms, 26-point models 21.9 / - / 19.0 / 18.1 / - ms)
                    // (a chunk left for its length alone - a whole series handed over as one chunk - may have models of any
                    // length, and chains through long models meet late: pieces sized for the chunks that are left, taken to
                    // be as long as the others, as for a call that goes to split mode by itself)
                    if (!option_text("MDB_FIT_PIECE_POINTS")) {
                        if (left[1] == 0) {
                            piece_points = std::min<uint32_t>(piece_points, untried ? FIT_UNTRIED_PIECE_POINTS : FIT_LEFT_PIECE_POINTS);
                        } else {
                            const uint32_t for_these = split_piece_points(ctx, n_left, points_end / n_chunks * n_left);
                            if (for_these != 0) piece_points = for_these;
                        }
                    }
                }
            }
        }
        if (!wave && piece_points == 0) {
            LaunchTimer timer(ctx, (lean_ts || (!ts && lean)) ? "k_fit_models_lean" : "k_fit_models");
            const uint32_t fit_blocks = (uint32_t)((n_chunks + FIT_THREADS - 1) / FIT_THREADS);
#define MDB_LAUNCH_LEAN(SPLIT, SPLIT_ARGS, HAS_TS)                                                                          \
    do {                                                                                                                   \
        if (eb.kind == MDB_EB_RELATIVE)                                                                                    \
            hipLaunchKernelGGL((k_fit_models_lean<SPLIT, MDB_EB_RELATIVE, HAS_TS>), dim3(fit_blocks), dim3(FIT_THREADS), 0, \
                               ctx->stream, args, SPLIT_ARGS, record_base, records, plans, error_flag, LeanRotation{});    \
        else if (eb.kind == MDB_EB_ABSOLUTE)                                                                               \
            hipLaunchKernelGGL((k_fit_models_lean<SPLIT, MDB_EB_ABSOLUTE, HAS_TS>), dim3(fit_blocks), dim3(FIT_THREADS), 0, \
                               ctx->stream, args, SPLIT_ARGS, record_base, records, plans, error_flag, LeanRotation{});    \
        else                                                                                                               \
            hipLaunchKernelGGL((k_fit_models_lean<SPLIT, MDB_EB_LOSSLESS, HAS_TS>), dim3(fit_blocks), dim3(FIT_THREADS), 0, \
                               ctx->stream, args, SPLIT_ARGS, record_base, records, plans, error_flag, LeanRotation{});    \
    } while (0)
            if (lean_ts)
                MDB_LAUNCH_LEAN(false, SplitArgs{}, true);
            else if (ts)
                hipLaunchKernelGGL((k_fit_models<true, false, false>), dim3(fit_blocks), dim3(FIT_THREADS), 0,
                                   ctx->stream, args, SplitArgs{}, record_base, records, plans, error_flag);
            else if (lean) {
                // Rotation (see k_fit_models_lean): when the groups of 64 chunks are not a whole number per SIMD, they
                // are taken by a few waves fewer, in stretches. MDB_FIT_ROTATE: 0 never, 1 whenever it can be done;
                // MDB_FIT_ROTATE_STEPS: steps per stretch.
                const uint64_t simds = (uint64_t)std::max(ctx->compute_units, 1) * 4u, groups = fit_blocks;
                const uint64_t per_simd = (groups + simds - 1) / simds; // waves the fullest SIMDs hold
                const char *rotate_setting = option_text("MDB_FIT_ROTATE");
                const bool forced = rotate_setting && std::strcmp(rotate_setting, "1") == 0;
                const bool rotate = !(rotate_setting && std::strcmp(rotate_setting, "0") == 0) && groups >= 2 &&
                                    (forced || (groups > simds && per_simd <= 4 && per_simd * simds - groups >= simds / 8));
                LeanRotation rotation;
                // A few waves fewer than groups: every wave has a group at all times, and the few groups that wait at
                // any moment are what makes the groups change places.
                const uint32_t rotating_blocks = (uint32_t)(groups - std::max<uint64_t>(1, groups / 128));
                if (rotate) {
                    rotation.n_groups = (uint32_t)groups;
                    rotation.error = error_flag;
                    // (a nap is 4 x s_sleep(127), some 14 us: a minute by default; MDB_FIT_ROTATE_MAX_NAPS for the test of it)
                    rotation.max_naps = fit_wave_number("MDB_FIT_ROTATE_MAX_NAPS", 1u << 22);
                    rotation.stretch_steps = LEAN_STRETCH_STEPS;
                    if (const char *text = option_text("MDB_FIT_ROTATE_STEPS")) rotation.stretch_steps = (uint32_t)std::max(1ll, std::atoll(text));
                    uint64_t slots = 64;
                    while (slots < 2 * (groups + rotating_blocks)) slots <<= 1;
                    rotation.slot_mask = (uint32_t)(slots - 1);
                    const uint64_t ready_bytes = align_up(slots * 4, 256), saved_bytes = align_up(groups * MDB_WAVE * sizeof(LeanSaved), 256);
                    const uint64_t mask_bytes = align_up(groups * LEAN_MASK_WORDS * 8, 256);
                    void *q = nullptr;
                    FIT_TRY(scratch_reserve(ctx, SCRATCH_FIT_ROTATION, 256 + ready_bytes + saved_bytes + mask_bytes, &q));
                    uint8_t *at = static_cast<uint8_t *>(q);
                    rotation.counters = reinterpret_cast<unsigned int *>(at);
                    rotation.ready = reinterpret_cast<unsigned int *>(at + 256);
                    rotation.saved = reinterpret_cast<unsigned long long *>(at + 256 + ready_bytes);
                    rotation.masks = reinterpret_cast<unsigned long long *>(at + 256 + ready_bytes + saved_bytes);

                    const uint64_t begin_threads = std::max<uint64_t>((uint64_t)rotation.slot_mask + 1, groups * MDB_WAVE);
                    hipLaunchKernelGGL(k_fit_rotation_begin, dim3((uint32_t)((begin_threads + 255) / 256)), dim3(256), 0, ctx->stream, rotation);
#define MDB_LAUNCH_ROTATING(KIND)                                                                                              \
    hipLaunchKernelGGL((k_fit_models_lean<false, KIND, false, true>), dim3(rotating_blocks), dim3(FIT_THREADS), 0,             \
                       ctx->stream, args, SplitArgs{}, record_base, records, plans, error_flag, rotation)
                    if (eb.kind == MDB_EB_RELATIVE) MDB_LAUNCH_ROTATING(MDB_EB_RELATIVE);
                    else if (eb.kind == MDB_EB_ABSOLUTE) MDB_LAUNCH_ROTATING(MDB_EB_ABSOLUTE);
                    else MDB_LAUNCH_ROTATING(MDB_EB_LOSSLESS);
#undef MDB_LAUNCH_ROTATING
                } else {
                    MDB_LAUNCH_LEAN(false, SplitArgs{}, false);
                }
            } else if (fast)
                hipLaunchKernelGGL((k_fit_models<false, false, true>), dim3(fit_blocks), dim3(FIT_THREADS), 0,
                                   ctx->stream, args, SplitArgs{}, record_base, records, plans, error_flag);
            else
                hipLaunchKernelGGL((k_fit_models<false, false, false>), dim3(fit_blocks), dim3(FIT_THREADS), 0,
                                   ctx->stream, args, SplitArgs{}, record_base, records, plans, error_flag);
        }
#ifdef MDB_FIT_TIMING
        if (!wave && piece_points == 0 && lean && !ts && eb.kind == MDB_EB_RELATIVE) {
            unsigned long long t[8] = {};
            FIT_CHECK(mail_sync(ctx));
            FIT_CHECK(hipMemcpyFromSymbol(t, HIP_SYMBOL(g_fit_timing), sizeof(t)));
            const unsigned long long zero[8] = {};
            FIT_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_fit_timing), zero, sizeof(zero)));
            const double waves = (double)t[5], steps = (double)t[3];
            std::fprintf(stderr, "[fit timing] region %d: %.1f cycles per pass, %.4f passes per step = %.1f cycles per step; loop %.1f cycles per step "
                         "(%.0f waves, %.0f steps per wave, shader clock %.0f MHz)\n", (int)MDB_FIT_TIMING,
                         t[1] ? (double)t[0] / (double)t[1] : 0.0, (double)t[1] / steps, (double)t[0] / steps, (double)t[2] / steps, waves,
                         steps / waves, t[4] ? (double)t[2] / ((double)t[4] / 100.0) : 0.0);
            if (MDB_FIT_TIMING == 6) {
                const size_t n_waves = std::min<size_t>((size_t)t[5], 65536);
                std::vector<unsigned long long> w(4 * n_waves);
                FIT_CHECK(hipMemcpyFromSymbol(w.data(), HIP_SYMBOL(g_fit_waves), 32 * n_waves));
                if (FILE *f = std::fopen(std::getenv("MDB_FIT_TIMING_FILE") ? std::getenv("MDB_FIT_TIMING_FILE") : "/tmp/fit_waves.csv", "w")) {
                    std::fprintf(f, "wave,ticks,hw_id,xcc_id,start\n");
                    for (size_t k = 0; k < n_waves; k++)
                        std::fprintf(f, "%zu,%llu,%llu,%llu,%llu\n", k, w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
                    std::fclose(f);
                }
            }
        }
#endif
        if (split_mode) {
            // Split mode: pieces of every chunk fitted speculatively, then the real chain is walked.
            SplitArgs split{};
            split.piece_points = piece_points;
            split.chunk_left = split_only;
            const uint64_t table_points = points_end;
            FIT_TRY(scratch_reserve(ctx, SCRATCH_FIT_SPLIT, (n_chunks + 1) * 8 + 64 + table_points * 12, &p));
            unsigned long long *piece_base = static_cast<unsigned long long *>(p);
            split.piece_base = piece_base;
            split.entry = reinterpret_cast<unsigned int *>(reinterpret_cast<uint8_t *>(p) +
                                                           align_up((n_chunks + 1) * 8, 64));
            split.p0 = reinterpret_cast<float *>(split.entry + table_points);
            split.p1 = split.p0 + table_points;
            FIT_TRY(device_exclusive_scan(ctx, PieceCount{args.chunk_offsets, piece_points, split_only}, n_chunks, piece_base,
                                          block_sums, "k_fit_scan"));
            unsigned long long n_pieces = 0;
            FIT_CHECK(mail_read(ctx, &n_pieces, piece_base + n_chunks, 8));
            FIT_CHECK(mail_sync(ctx));
            split.n_pieces = n_pieces;
            // The start points at which no model can begin, found for all points at once (k_fit_reject_flags; the lean
            // fitter's values-only form under a lossy bound; MDB_FIT_REJECT_FLAGS=0: every start point fed to the fitters).
            // (the kernel writes the entry of every point that has a piece - 0 or "no model from here": no clearing then)
            const char *flags_setting = option_text("MDB_FIT_REJECT_FLAGS");
            const bool with_flags = n_pieces > 0 && lean && !ts && !lean_ts && eb.kind != MDB_EB_LOSSLESS && piece_points % MDB_WAVE == 0 &&
                                    !(flags_setting && std::strcmp(flags_setting, "0") == 0);
            if (!with_flags) FIT_CHECK(hipMemsetAsync(split.entry, 0, table_points * 4, ctx->stream));
            if (with_flags) {
                split.reject_words_per_piece = piece_points / MDB_WAVE;
                const uint64_t n_words = n_pieces * split.reject_words_per_piece;
                FIT_TRY(scratch_reserve(ctx, SCRATCH_FIT_REJECTS, (n_words + 1) * 8, &p));
                unsigned long long *words = static_cast<unsigned long long *>(p);
                unsigned long long *bits_set = words + n_words;
                FIT_CHECK(hipMemsetAsync(bits_set, 0, 8, ctx->stream));
                split.reject_words = words;
                {
                    LaunchTimer timer(ctx, "k_fit_reject_flags");
                    if (eb.kind == MDB_EB_RELATIVE)
                        hipLaunchKernelGGL(k_fit_reject_flags<MDB_EB_RELATIVE>, dim3((uint32_t)((n_pieces + FLAG_PIECES - 1) / FLAG_PIECES)), dim3(256), 0, ctx->stream, args, split, words, bits_set);
                    else
                        hipLaunchKernelGGL(k_fit_reject_flags<MDB_EB_ABSOLUTE>, dim3((uint32_t)((n_pieces + FLAG_PIECES - 1) / FLAG_PIECES)), dim3(256), 0, ctx->stream, args, split, words, bits_set);
                }
                // Few of them (sine + noise under a bound of the noise's size: 26-point models, hardly a start point that
                // certainly fails): the fitter's look at the bits every step would cost more than it saves (7.5 -> 8.4 ms).
                // The entries that say "no model from here" stay: a lane that comes by overwrites them like any other.
                unsigned long long set = 0;
                FIT_CHECK(mail_read(ctx, &set, bits_set, 8));
                FIT_CHECK(mail_sync(ctx));
                set *= 8 * (256 / MDB_WAVE); // (counted in one wave's rows of every 8th workgroup)
                if (option_text("MDB_FIT_DEBUG"))
                    std::fprintf(stderr, "[fit] k_fit_reject_flags: %llu pieces of %u points, about %llu start points at which no model can begin\n",
                                 n_pieces, piece_points, set);
                if (set * FIT_FLAGS_WORTH_ONE_IN < n_pieces * (unsigned long long)piece_points &&
                    !(flags_setting && std::strcmp(flags_setting, "1") == 0))
                    split.reject_words = nullptr;
            }
            if (n_pieces > 0) {
                LaunchTimer timer(ctx, "k_fit_models_split");
                const uint32_t fit_blocks = (uint32_t)((n_pieces + FIT_THREADS - 1) / FIT_THREADS);
                if (lean_ts)
                    MDB_LAUNCH_LEAN(true, split, true);
                else if (ts)
                    hipLaunchKernelGGL((k_fit_models<true, true, false>), dim3(fit_blocks), dim3(FIT_THREADS), 0,
                                       ctx->stream, args, split, record_base, records, plans, error_flag);
                else if (lean)
                    MDB_LAUNCH_LEAN(true, split, false);
                else if (fast)
                    hipLaunchKernelGGL((k_fit_models<false, true, true>), dim3(fit_blocks), dim3(FIT_THREADS), 0,
                                       ctx->stream, args, split, record_base, records, plans, error_flag);
                else
                    hipLaunchKernelGGL((k_fit_models<false, true, false>), dim3(fit_blocks), dim3(FIT_THREADS), 0,
                                       ctx->stream, args, split, record_base, records, plans, error_flag);
            }
#if defined(MDB_FIT_TIMING) && defined(MDB_FIT_TIMING_SPLIT)
            if (n_pieces > 0 && lean && !ts && eb.kind == MDB_EB_RELATIVE) {
                unsigned long long t[8] = {};
                FIT_CHECK(mail_sync(ctx));
                FIT_CHECK(hipMemcpyFromSymbol(t, HIP_SYMBOL(g_fit_timing), sizeof(t)));
                const unsigned long long zero[8] = {};
                FIT_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_fit_timing), zero, sizeof(zero)));
                const double waves = (double)t[5], steps = (double)t[3];
                std::fprintf(stderr, "[fit timing, split] region %d: %.1f cycles per pass, %.4f passes per step = %.1f cycles per step; loop %.1f cycles per step "
                             "(%.0f waves, %.0f steps per wave, %llu pieces of %u points)\n", (int)MDB_FIT_TIMING,
                             t[1] ? (double)t[0] / (double)t[1] : 0.0, (double)t[1] / steps, (double)t[0] / steps, (double)t[2] / steps, waves,
                             steps / waves, n_pieces, piece_points);
            }
#endif
            LaunchTimer timer(ctx, "k_fit_walk");
            hipLaunchKernelGGL(k_fit_walk, dim3((uint32_t)n_chunks), dim3(MDB_WAVE), 0, ctx->stream,
                               args.chunk_offsets, n_chunks, split, record_base, records, plans, error_flag);
#ifdef MDB_WALK_DEBUG
            {
                unsigned long long c[8] = {};
                (void)mail_sync(ctx);
                (void)hipMemcpyFromSymbol(c, HIP_SYMBOL(g_walk_counts), sizeof(c));
                const unsigned long long zero[8] = {};
                (void)hipMemcpyToSymbol(HIP_SYMBOL(g_walk_counts), zero, sizeof(zero));
                std::fprintf(stderr, "[walk] %llu chunks: iterations %llu, reloads %llu, models %llu; per chunk %.0f iterations %.0f reloads; mean %.1f us max %.1f us\n",
                             c[3], c[0], c[1], c[2], c[3] ? (double)c[0] / c[3] : 0.0, c[3] ? (double)c[1] / c[3] : 0.0, c[3] ? c[4] / 100.0 / c[3] : 0.0, c[5] / 100.0);
            }
#endif
        }
        FIT_TRY(device_exclusive_scan(ctx, SegmentCount{plans}, n_chunks, segment_base, block_sums,
                                      "k_fit_scan"));
        unsigned int error = 0;
        FIT_CHECK(mail_read(ctx, &n_segments, segment_base + n_chunks, 8));
        FIT_CHECK(mail_read(ctx, &error, error_flag, 4));
        FIT_CHECK(mail_sync(ctx));
        FIT_CHECK(hipGetLastError());
        if (error & ERR_SPLIT_CHAIN) {
            release();
            return fail("Internal error: the split fit left a gap in a chunk's model chain.");
        }
        if (error & ERR_ROTATION_STALL) {
            release();
            return fail("Internal error: a wave of the rotating fit waited for a group of chunks that was never handed over.");
        }
        if (error) {
            release();
            return fail("A chunk holds more than 2^31-3 data points.");
        }
        if (n_segments > 0x7ffffff0ull) {
            release();
            return fail("Too many segments for one batch.");
        }

        FIT_TRY(scratch_reserve(ctx, SCRATCH_FIT_E, n_segments * sizeof(SegItem), &p));
        SegItem *items = static_cast<SegItem *>(p);
        const uint64_t seg_sums = scan_block_sums_bytes(std::max<uint64_t>(n_segments, 1));
        FIT_TRY(scratch_reserve(ctx, SCRATCH_FIT_F,
                                n_segments * sizeof(SegSizes) + 3 * (n_segments + 1) * 8 + seg_sums + 64, &p));
        SegSizes *sizes = static_cast<SegSizes *>(p);
        unsigned long long *data_offsets[3];
        data_offsets[0] = reinterpret_cast<unsigned long long *>(sizes + n_segments);
        data_offsets[1] = data_offsets[0] + n_segments + 1;
        data_offsets[2] = data_offsets[1] + n_segments + 1;
        unsigned long long *seg_block_sums = data_offsets[2] + n_segments + 1;
        {
            LaunchTimer timer(ctx, "k_fit_plan");
            hipLaunchKernelGGL(k_fit_plan, dim3((uint32_t)n_chunks), dim3(MDB_WAVE), 0, ctx->stream,
                               args.chunk_offsets, n_chunks, record_base, records, plans, segment_base,
                               items);
        }
        const uint32_t segment_blocks = (uint32_t)((n_segments + 255) / 256);
        // Long lossless MacaqueV-only segments: one wave each (k_fit_gap), sized here, written below.
        uint32_t *gap_ids = nullptr, *n_gaps = nullptr;
        uint32_t gap_waves = 0;
        GapStage gap_stage; // (bytes: the listed segments are encoded once, into staging places)
        LongArgs long_args;
        uint32_t long_segments = 0, long_blocks = 0;
        bool long_in_blocks = false; // (false: the long ones take a wave each like the others, listed apart)
        const uint32_t gap_min_values = gap_min_values_setting();
        if (n_segments > 0 && gap_min_values != 0xffffffffu) {
            const uint64_t most = std::min<uint64_t>(n_segments, points_end / gap_min_values + 1);
            FIT_TRY(scratch_reserve(ctx, SCRATCH_FIT_GAP, n_segments * sizeof(GapResult) + most * 4 + 512, &p));
            GapResult *gap_results = static_cast<GapResult *>(p);
            gap_ids = reinterpret_cast<uint32_t *>(gap_results + n_segments);
            n_gaps = gap_ids + align_up(most, 64);
            gap_waves = (uint32_t)most;
            args.gap_min_values = gap_min_values;
            args.gap_results = gap_results;
            // The longest of them are cut into blocks (k_fit_long*): listed apart, with room for their blocks.
            LongCounters *counters = reinterpret_cast<LongCounters *>(n_gaps); // ([0] is the gap list's counter)
            long_args.long_min_values = std::max(gap_long_min_values_setting(), gap_min_values);
            long_args.block_min_values = gap_block_values_setting();
            args.gap_long_min_values = long_args.long_min_values;
            const uint64_t most_long = long_args.long_min_values == 0xffffffffu
                                           ? 0
                                           : std::min<uint64_t>(n_segments, points_end / long_args.long_min_values + 1);
            if (most_long > 0) {
                FIT_TRY(scratch_reserve(ctx, SCRATCH_FIT_LONG_IDS, 3 * 4 * most_long, &p));
                long_args.long_ids = static_cast<uint32_t *>(p);
                long_args.long_block_base = long_args.long_ids + most_long;
                long_args.long_batch_base = long_args.long_block_base + most_long;
            }
            long_args.counters = counters;
            FIT_CHECK(hipMemsetAsync(counters, 0, sizeof(LongCounters), ctx->stream));
            {
                LaunchTimer timer(ctx, "k_fit_gap_select");
                hipLaunchKernelGGL(k_fit_gap_select, dim3(segment_blocks), dim3(256), 0, ctx->stream, args, items,
                                   (uint64_t)n_segments, gap_ids, n_gaps);
                if (most_long > 0)
                    hipLaunchKernelGGL(k_fit_long_select, dim3(segment_blocks), dim3(256), 0, ctx->stream, args, items,
                                       (uint64_t)n_segments, long_args, counters);
            }
            // How many there are decides the launches (usually none, and then nothing is launched).
            LongCounters found{};
            FIT_CHECK(mail_read(ctx, &found, counters, sizeof(LongCounters)));
            FIT_CHECK(mail_sync(ctx));
            gap_waves = found.n_gaps;
            long_segments = found.n_long;
            long_blocks = found.n_blocks;
            // Encoded once (GAP_STAGE; MDB_FIT_GAP_ONCE=0: sized, then encoded again; =1: under a lossless bound too):
            // every listed segment gets a staging place from an upper bound of its bytes, unless those add up to more
            // than the device has to spare. Under a lossy bound, where measuring a stream is all of encoding it but the
            // stores (the mixed series at 1 %: 2.54 + 2.68 ms -> 3.46 + 0.43 for the copy); under a lossless bound the
            // measuring pass is the cheaper one and staging gains nothing (2.79 + 3.46 -> 4.47 + 1.03).
            if (gap_waves > 0 && gap_once_setting(eb)) {
                FIT_TRY(scratch_reserve(ctx, SCRATCH_FIT_GAP_STAGE_OFFSETS, ((uint64_t)gap_waves + 1) * 8, &p));
                unsigned long long *stage_offsets = static_cast<unsigned long long *>(p);
                FIT_TRY(device_exclusive_scan(ctx, GapStageBytes{items, gap_ids}, gap_waves, stage_offsets, seg_block_sums, "k_fit_scan"));
                // What the places add up to at most is known here (gap_stage_bytes: every listed segment's values are among
                // the call's points): while the slot holds that much the call asks neither the device for the sum nor the
                // driver for its free memory - the steady state of a server's calls.
                const unsigned long long at_most = points_end * 45u / 8u + 48ull * gap_waves;
                if (at_most <= ctx->scratch_bytes[SCRATCH_FIT_GAP_STAGE] && ctx->scratch[SCRATCH_FIT_GAP_STAGE]) {
                    gap_stage.bytes = static_cast<uint8_t *>(ctx->scratch[SCRATCH_FIT_GAP_STAGE]);
                    gap_stage.offsets = stage_offsets;
                } else {
                    unsigned long long stage_bytes = 0;
                    FIT_CHECK(mail_read(ctx, &stage_bytes, stage_offsets + gap_waves, 8));
                    FIT_CHECK(mail_sync(ctx));
                    size_t device_free = 0, device_total = 0;
                    if (stage_bytes > ctx->scratch_bytes[SCRATCH_FIT_GAP_STAGE] && hipMemGetInfo(&device_free, &device_total) != hipSuccess) {
                        (void)hipGetLastError();
                        device_free = 0;
                    }
                    // (what is reserved already counts as free for this: the slot grows, it is not added to)
                    if (stage_bytes <= ctx->scratch_bytes[SCRATCH_FIT_GAP_STAGE] || stage_bytes <= device_free / 4) {
                        FIT_TRY(scratch_reserve(ctx, SCRATCH_FIT_GAP_STAGE, stage_bytes, &p));
                        gap_stage.bytes = static_cast<uint8_t *>(p);
                        gap_stage.offsets = stage_offsets;
                    }
                }
            }
            if (gap_waves > 0) {
                LaunchTimer timer(ctx, "k_fit_gap_size");
                if (gap_stage.bytes)
                    hipLaunchKernelGGL(k_fit_gap<GAP_STAGE>, dim3(gap_waves), dim3(MDB_WAVE), 0, ctx->stream, args, items,
                                       gap_ids, n_gaps, gap_results, EncodeTargets{}, gap_stage);
                else
                    hipLaunchKernelGGL(k_fit_gap<GAP_SIZE>, dim3(gap_waves), dim3(MDB_WAVE), 0, ctx->stream, args, items,
                                       gap_ids, n_gaps, gap_results, EncodeTargets{}, GapStage{});
            }
            // Blocks are for calls with too few streams to occupy the device with a wave each: 10^3 streams of 50 000
            // values are fitted in 1.7 ms cut into blocks and 2.7 ms with a wave each, 10^4 in 7.2 against 6.7 (the
            // blocks' notes and the second look at every block's first batch are work the one wave does not do).
            // (MDB_FIT_GAP_LONG_BELOW_WAVES: the number of streams from which a wave each is enough; default 16 per compute unit)
            uint64_t enough_waves = (uint64_t)std::max(ctx->compute_units, 1) * 16u;
            if (const char *text = option_text("MDB_FIT_GAP_LONG_BELOW_WAVES")) enough_waves = (uint64_t)std::max(0ll, std::atoll(text));
            long_in_blocks = long_segments > 0 && (uint64_t)gap_waves + long_segments < enough_waves;
            if (long_segments > 0 && !long_in_blocks) {
                LaunchTimer timer(ctx, "k_fit_gap_size");
                hipLaunchKernelGGL(k_fit_gap<GAP_SIZE>, dim3(long_segments), dim3(MDB_WAVE), 0, ctx->stream, args, items,
                                   long_args.long_ids, &counters->n_long, gap_results, EncodeTargets{});
            }
            if (long_in_blocks) {
                const uint64_t block_bytes = align_up((uint64_t)long_blocks * sizeof(GapBlock), 256);
                const uint64_t prefix_bytes = align_up((uint64_t)found.n_batches * sizeof(GapBatchNote), 256);
                FIT_TRY(scratch_reserve(ctx, SCRATCH_FIT_LONG, block_bytes + prefix_bytes + (uint64_t)found.n_batches * 8, &p));
                long_args.blocks = static_cast<GapBlock *>(p);
                long_args.batch_notes = reinterpret_cast<GapBatchNote *>(static_cast<uint8_t *>(p) + block_bytes);
                long_args.batch_breakers = reinterpret_cast<unsigned long long *>(static_cast<uint8_t *>(p) + block_bytes + prefix_bytes);
                launch_long_sizing(ctx, ctx->stream, args, items, long_args, gap_results, long_segments, long_blocks);
            }
        }
        // Segments of irregular chunks: their timestamps are sized (and below, written) by a wave each.
        const bool ts_by_wave = ts != nullptr && n_segments > 0 && n_segments <= 0x7fffffffull;
        if (ts_by_wave) {
            FIT_TRY(scratch_reserve(ctx, SCRATCH_FIT_TS, n_segments * sizeof(TsResult), &p));
            TsResult *ts_results = static_cast<TsResult *>(p);
            FIT_CHECK(hipMemsetAsync(ts_results, 0xff, n_segments * sizeof(TsResult), ctx->stream));
            args.ts_results = ts_results;
            LaunchTimer timer(ctx, "k_fit_timestamps_size");
            hipLaunchKernelGGL(k_fit_timestamps<false>, dim3((uint32_t)n_segments), dim3(MDB_WAVE), 0, ctx->stream, args,
                               items, ts_results, EncodeTargets{});
        }
        if (n_segments > 0) {
            LaunchTimer timer(ctx, "k_fit_size");
            hipLaunchKernelGGL(k_fit_size, dim3((uint32_t)((n_segments + FIT_SEGMENT_THREADS - 1) / FIT_SEGMENT_THREADS)),
                               dim3(FIT_SEGMENT_THREADS), 0, ctx->stream, args,
                               record_base, records, items, (uint64_t)n_segments, sizes);
        }
        unsigned long long data_bytes[3] = {0, 0, 0};
        for (int c = 0; c < 3; c++) {
            FIT_TRY(device_exclusive_scan(ctx, OutOfLineBytes{sizes, c}, n_segments, data_offsets[c],
                                          seg_block_sums, "k_fit_scan"));
            FIT_CHECK(mail_read(ctx, &data_bytes[c], data_offsets[c] + n_segments, 8));
            FIT_CHECK(mail_sync(ctx)); // seg_block_sums is reused by the next scan
        }
        // A column's payloads become several data buffers when they add up to more than one buffer may
        // hold (arrow's builders roll over the same way, types.rs:444-516). MDB_FIT_DATA_BUFFER_BYTES: the
        // size from which a new buffer is begun (tests force it down); a buffer ends with the payload that
        // begins in it, so it stays below 2 GiB as long as no single payload is larger than the rest.
        const uint64_t buffer_bytes = data_buffer_bytes_setting();
        uint32_t n_data_buffers[3];
        std::vector<unsigned long long> data_bases[3];
        FIT_TRY(scratch_reserve(ctx, SCRATCH_FIT_BASES, 3 * 8 * (uint64_t)MAX_DATA_BUFFERS + 64, &p));
        unsigned long long *bases_dev = static_cast<unsigned long long *>(p);
        for (int c = 0; c < 3; c++) {
            const uint64_t wanted = data_bytes[c] <= buffer_bytes ? 1 : (data_bytes[c] + buffer_bytes - 1) / buffer_bytes;
            if (wanted > MAX_DATA_BUFFERS) {
                release();
                return fail("Too many BinaryView data buffers for one batch; compress fewer chunks per call.");
            }
            n_data_buffers[c] = (uint32_t)wanted;
            data_bases[c].assign(wanted + 1, 0);
            if (wanted > 1) {
                hipLaunchKernelGGL(k_fit_data_bases, dim3(1), dim3(64), 0, ctx->stream, data_offsets[c],
                                   (uint64_t)n_segments, buffer_bytes, n_data_buffers[c], bases_dev + c * MAX_DATA_BUFFERS);
                FIT_CHECK(mail_read(ctx, data_bases[c].data(), bases_dev + c * MAX_DATA_BUFFERS, 8 * wanted));
                FIT_CHECK(mail_sync(ctx));
            } else {
                FIT_CHECK(hipMemsetAsync(bases_dev + c * MAX_DATA_BUFFERS, 0, 8, ctx->stream));
            }
            data_bases[c][wanted] = data_bytes[c];
            for (uint64_t k = 0; k < wanted; k++)
                if (data_bases[c][k + 1] - data_bases[c][k] > 0x7fffffffull) {
                    release();
                    return fail("A BinaryView data buffer would exceed 2 GiB: one payload is too large.");
                }
        }

        // One device blob for the output batch.
        const uint64_t n = n_segments;
        uint64_t cursor = 0;
        auto carve = [&](uint64_t bytes) {
            uint64_t at = cursor;
            cursor = align_up(cursor + bytes, 256);
            return at;
        };
        const uint64_t off_type = carve(n), off_start = carve(8 * n), off_end = carve(8 * n);
        const uint64_t off_min = carve(4 * n), off_max = carve(4 * n), off_error = carve(4 * n);
        const uint64_t off_chunk = carve(4 * n);
        uint64_t off_views[3], off_data[3], off_table[3];
        for (int c = 0; c < 3; c++) off_views[c] = carve(16 * n);
        for (int c = 0; c < 3; c++) off_data[c] = carve(data_bytes[c]);
        for (int c = 0; c < 3; c++) off_table[c] = carve(8 * (uint64_t)(n_data_buffers[c] + 1));
        void *blob = nullptr;
        FIT_CHECK(hipMalloc(&blob, cursor ? cursor : 256));
        owned->device_allocs.push_back(blob);
        uint8_t *dev = static_cast<uint8_t *>(blob);

        EncodeTargets targets;
        targets.model_type_id = reinterpret_cast<int8_t *>(dev + off_type);
        targets.start_time = reinterpret_cast<int64_t *>(dev + off_start);
        targets.end_time = reinterpret_cast<int64_t *>(dev + off_end);
        targets.min_value = reinterpret_cast<float *>(dev + off_min);
        targets.max_value = reinterpret_cast<float *>(dev + off_max);
        targets.error = reinterpret_cast<float *>(dev + off_error);
        targets.chunk_index = reinterpret_cast<uint32_t *>(dev + off_chunk);
        std::vector<uint64_t> tables[3];
        for (int c = 0; c < 3; c++) {
            targets.views[c] = reinterpret_cast<uint4 *>(dev + off_views[c]);
            targets.data[c] = dev + off_data[c];
            targets.data_offsets[c] = data_offsets[c];
            targets.data_bases[c] = bases_dev + c * MAX_DATA_BUFFERS;
            targets.n_data_buffers[c] = n_data_buffers[c];
            tables[c].assign(n_data_buffers[c] + 1, 0);
            for (uint32_t k = 0; k < n_data_buffers[c]; k++)
                tables[c][k] = reinterpret_cast<uint64_t>(dev + off_data[c] + data_bases[c][k]);
            FIT_CHECK(mail_write(ctx, dev + off_table[c], tables[c].data(), 8 * tables[c].size()));
        }
        FIT_CHECK(mail_sync(ctx)); // (`tables` is pageable memory of this frame)
        if (gap_waves > 0 && gap_stage.bytes) { // before k_fit_encode, which reads the first payload bytes for the views
            LaunchTimer timer(ctx, "k_fit_gap_place");
            hipLaunchKernelGGL(k_fit_gap_place, dim3(gap_waves), dim3(MDB_WAVE), 0, ctx->stream, gap_ids, n_gaps,
                               args.gap_results, targets, gap_stage);
        } else if (gap_waves > 0) {
            LaunchTimer timer(ctx, "k_fit_gap_encode");
            hipLaunchKernelGGL(k_fit_gap<GAP_WRITE>, dim3(gap_waves), dim3(MDB_WAVE), 0, ctx->stream, args, items,
                               gap_ids, n_gaps, const_cast<GapResult *>(args.gap_results), targets, GapStage{});
        }
        if (long_in_blocks) {
            launch_long_encode(ctx, ctx->stream, args, items, long_args, targets, long_blocks);
        } else if (long_segments > 0) {
            LaunchTimer timer(ctx, "k_fit_gap_encode");
            hipLaunchKernelGGL(k_fit_gap<GAP_WRITE>, dim3(long_segments), dim3(MDB_WAVE), 0, ctx->stream, args, items,
                               long_args.long_ids, &long_args.counters->n_long, const_cast<GapResult *>(args.gap_results), targets);
        }
        if (ts_by_wave) { // before k_fit_encode as well
            LaunchTimer timer(ctx, "k_fit_timestamps_encode");
            hipLaunchKernelGGL(k_fit_timestamps<true>, dim3((uint32_t)n_segments), dim3(MDB_WAVE), 0, ctx->stream, args,
                               items, const_cast<TsResult *>(args.ts_results), targets);
        }
        if (n_segments > 0) {
            LaunchTimer timer(ctx, "k_fit_encode");
            hipLaunchKernelGGL(k_fit_encode, dim3((uint32_t)((n_segments + FIT_SEGMENT_THREADS - 1) / FIT_SEGMENT_THREADS)),
                               dim3(FIT_SEGMENT_THREADS), 0, ctx->stream, args,
                               record_base, records, items, (uint64_t)n_segments, sizes, targets);
        }
        FIT_CHECK(mail_sync(ctx));
        FIT_CHECK(hipGetLastError());

        mdb_segments &s = owned->c.seg;
        s.n = n;
        s.model_type_id = targets.model_type_id;
        s.start_time = targets.start_time;
        s.end_time = targets.end_time;
        s.min_value = targets.min_value;
        s.max_value = targets.max_value;
        mdb_binview_col *cols[3] = {&s.timestamps, &s.values, &s.residuals};
        for (int c = 0; c < 3; c++) {
            owned->host_allocs[c].resize(8 * (size_t)n_data_buffers[c]);
            int64_t *size_slots = reinterpret_cast<int64_t *>(owned->host_allocs[c].data());
            for (uint32_t k = 0; k < n_data_buffers[c]; k++)
                size_slots[k] = (int64_t)(data_bases[c][k + 1] - data_bases[c][k]);
            cols[c]->views = reinterpret_cast<const mdb_view16 *>(targets.views[c]);
            cols[c]->buffers = reinterpret_cast<const uint8_t *const *>(dev + off_table[c]);
            cols[c]->buffer_sizes = size_slots;
            cols[c]->n_buffers = (int32_t)n_data_buffers[c];
        }
        owned->c.error = targets.error;
        owned->c.chunk_index = targets.chunk_index;
    } else {
        std::memset(&owned->c.seg, 0, sizeof(owned->c.seg));
        owned->c.error = nullptr;
        owned->c.chunk_index = nullptr;
    }
    owned->c.on_device = 1;
    owned->c.priv_ = owned;
    owned_segments_register(owned);
    *out = &owned->c;
    return 0;
#undef FIT_CHECK
#undef FIT_TRY
}

// MDB_FIT_SMALL=0: never the path of a handful of chunks; N >= 1: for calls of up to N chunks (default 64). Calls
// with any other MDB_FIT_* switch set take the general driver (the switches select among ITS kernels).
static uint64_t fit_small_max_chunks() {
    for (const char *name : {"MDB_FIT_WAVE", "MDB_FIT_PIECE_POINTS", "MDB_FIT_LEAN", "MDB_FIT_FAST", "MDB_FIT_GAP_MIN_VALUES",
                             "MDB_FIT_DATA_BUFFER_BYTES", "MDB_FIT_DEBUG", "MDB_FIT_ROTATE", "MDB_FIT_GAP_ONCE"})
        if (option_text(name)) return 0;
    if (const char *text = option_text("MDB_FIT_SMALL")) return (uint64_t)std::max(0ll, std::atoll(text));
    return 64;
}
constexpr uint64_t FIT_SMALL_MAX_POINTS = 1ull << 22;

// The fit of a handful of chunks whose values lie in HOST memory, to segments in host memory (file comment above
// k_small_segments). Returns 0: done, *out is the batch; 1: failed (fail() has the text); 2: not a call for this path
// (too large, timestamps that are not equally spaced or not exact as f64): the general driver takes it.
int fit_few_chunks(mdb_ctx *ctx, const mdb_chunk *chunks, uint64_t n_chunks, mdb_error_bound eb, mdb_segments_owned **out) {
    const uint64_t max_chunks = fit_small_max_chunks();
    if (n_chunks == 0 || n_chunks > max_chunks) return 2;
    if (!valid_error_bound(eb)) return 2; // (the general driver has the message)
    uint64_t total = 0;
    for (uint64_t c = 0; c < n_chunks; c++) {
        if (chunks[c].n > COUNT_MASK - ENTRY_END_BIAS) return 2;
        total += chunks[c].n;
    }
    if (total == 0 || total > FIT_SMALL_MAX_POINTS) return 2;
    // Equally spaced, and exact as f64 (k_fit_exact_double_timestamps)? One pass over the timestamps of every chunk.
    std::vector<long long> first(n_chunks, 0), interval(n_chunks, 0);
    const uint64_t limit = 1ull << 52;
    for (uint64_t c = 0; c < n_chunks; c++) {
        const uint64_t n = chunks[c].n;
        if (n == 0) continue;
        const int64_t *t = chunks[c].ts;
        const int64_t step = n > 1 ? (int64_t)((uint64_t)t[1] - (uint64_t)t[0]) : 0;
        bool differs = false;
        for (uint64_t j = 2; j < n; j++) differs |= (int64_t)((uint64_t)t[j] - (uint64_t)t[j - 1]) != step;
        if (differs) return 2;
        first[c] = t[0];
        interval[c] = step;
        const uint64_t magnitude = t[0] < 0 ? 0ull - (uint64_t)t[0] : (uint64_t)t[0];
        const uint64_t stride = step < 0 ? 0ull - (uint64_t)step : (uint64_t)step;
        if (!(magnitude <= limit && (stride == 0 || n <= limit / stride))) return 2;
        // (and no timestamp beyond +-2^52 at all: the line k_fit_regular draws for the general driver - a chunk that
        // begins below 2^52 and ends above it is "beyond" there and takes the careful fitter, so it does here)
        const uint64_t last_magnitude = t[n - 1] < 0 ? 0ull - (uint64_t)t[n - 1] : (uint64_t)t[n - 1];
        if (last_magnitude > limit) return 2;
    }

    // ---- sizes the host can know ----
    const uint32_t piece_points = (uint32_t)std::max<uint64_t>(2048, align_up(total / 2048 + 1, 64));
    std::vector<unsigned long long> offsets(n_chunks + 1, 0), piece_base(n_chunks + 1, 0), record_base(n_chunks + 1, 0);
    uint64_t segment_bound = 0;
    for (uint64_t c = 0; c < n_chunks; c++) {
        const uint64_t n = chunks[c].n;
        offsets[c + 1] = offsets[c] + n;
        piece_base[c + 1] = piece_base[c] + (n + piece_points - 1) / piece_points;
        record_base[c + 1] = record_base[c] + (n / 8 + 1);
        segment_bound += n / 8 + n / 256 + 2;
    }
    const uint64_t n_pieces = piece_base[n_chunks], total_records = record_base[n_chunks];
    const uint64_t gap_bound = total / GAP_DEFAULT_MIN_VALUES + n_chunks;
    const uint64_t blob_capacity = align_up(SMALL_HEADER_BYTES + 128 * segment_bound + 6 * total + 16 * segment_bound + 4096, 4096);

    // ---- the page-locked block: [meta | values] go up in one copy, [header | columns] come down into the rest ----
    uint64_t cursor = 0;
    auto carve = [&](uint64_t bytes) {
        const uint64_t at = cursor;
        cursor = align_up(cursor + bytes, 256);
        return at;
    };
    const uint64_t m_offsets = carve(8 * (n_chunks + 1)), m_first = carve(8 * n_chunks), m_interval = carve(8 * n_chunks);
    const uint64_t m_piece_base = carve(8 * (n_chunks + 1)), m_record_base = carve(8 * (n_chunks + 1));
    const uint64_t m_values = carve(4 * total);
    const uint64_t upload_bytes = cursor;
    // (device only, behind the upload; the zeroed part first)
    const uint64_t d_zero = cursor;
    const uint64_t d_error = carve(64), d_n_gaps = carve(64), d_n_segments = carve(64), d_zero_base = carve(64);
    const uint64_t d_entry = carve(4 * total);
    const uint64_t zero_bytes = cursor - d_zero;
    const uint64_t d_p0 = carve(4 * total), d_p1 = carve(4 * total);
    const uint64_t d_records = carve(sizeof(ModelRec) * total_records), d_plans = carve(sizeof(ChunkPlan) * n_chunks);
    const uint64_t d_segment_base = carve(8 * (n_chunks + 1)), d_items = carve(sizeof(SegItem) * segment_bound);
    const uint64_t d_sizes = carve(sizeof(SegSizes) * segment_bound), d_data_offsets = carve(3 * 8 * (segment_bound + 1));
    const uint64_t d_gap_results = carve(sizeof(GapResult) * segment_bound), d_gap_ids = carve(4 * gap_bound);
    // (the long MacaqueV-only segments, cut into blocks: k_fit_long*; upper bounds again)
    const uint32_t long_min_values = std::max(gap_long_min_values_setting(), GAP_DEFAULT_MIN_VALUES);
    const uint32_t block_min_values = gap_block_values_setting();
    const uint64_t long_bound = long_min_values == 0xffffffffu ? 0 : total / long_min_values + 1;
    const uint64_t long_blocks_bound = long_bound ? total / block_min_values + long_bound : 0;
    const uint64_t long_batches_bound = long_bound ? total / MDB_WAVE + total / 4096 + long_bound * (block_min_values / MDB_WAVE) + 64 : 0;
    const uint64_t d_long_ids = carve(3 * 4 * long_bound), d_long_blocks = carve(sizeof(GapBlock) * long_blocks_bound);
    const uint64_t d_long_prefix = carve(sizeof(GapBatchNote) * long_batches_bound);
    const uint64_t d_long_breakers = carve(eb.kind != MDB_EB_LOSSLESS ? 8 * long_batches_bound : 0);
    const uint64_t d_targets = carve(sizeof(EncodeTargets));
    const uint64_t d_blob = carve(blob_capacity);
    const uint64_t device_bytes = cursor;

    mdb::CallGuard lock(ctx);
    MDB_HIP_CHECK(hipSetDevice(ctx->device));
    void *pinned = nullptr, *device = nullptr;
    const uint64_t pinned_down = align_up(upload_bytes, 4096);
    if (pinned_reserve(ctx, pinned_down + blob_capacity, &pinned)) return 1;
    if (scratch_reserve(ctx, SCRATCH_FIT_SMALL, device_bytes, &device)) return 1;
    uint8_t *host = static_cast<uint8_t *>(pinned), *dev = static_cast<uint8_t *>(device);
    std::memcpy(host + m_offsets, offsets.data(), 8 * (n_chunks + 1));
    std::memcpy(host + m_first, first.data(), 8 * n_chunks);
    std::memcpy(host + m_interval, interval.data(), 8 * n_chunks);
    std::memcpy(host + m_piece_base, piece_base.data(), 8 * (n_chunks + 1));
    std::memcpy(host + m_record_base, record_base.data(), 8 * (n_chunks + 1));
    {
        // (the values: by the host's threads when there is more than a buffer or two of them)
        struct Gather {
            const mdb_chunk *chunks;
            const unsigned long long *offsets;
            float *to;
        } gather{chunks, offsets.data(), reinterpret_cast<float *>(host + m_values)};
        auto one = [](unsigned c, void *arg) {
            const Gather &g = *static_cast<Gather *>(arg);
            if (g.chunks[c].n) std::memcpy(g.to + g.offsets[c], g.chunks[c].values, 4 * g.chunks[c].n);
        };
        if (n_chunks >= 4 && total >= (1u << 18)) host_parallel((unsigned)n_chunks, one, &gather);
        else
            for (unsigned c = 0; c < (unsigned)n_chunks; c++) one(c, &gather);
    }
    hipStream_t stream = ctx->stream;
    MDB_HIP_CHECK(hipMemcpyAsync(dev, host, upload_bytes, hipMemcpyHostToDevice, stream));
    MDB_HIP_CHECK(hipMemsetAsync(dev + d_zero, 0, zero_bytes, stream));

    FitArgs args;
    args.values = reinterpret_cast<const float *>(dev + m_values);
    args.timestamps = {nullptr, 0, 0, nullptr, reinterpret_cast<const long long *>(dev + m_first),
                       reinterpret_cast<const long long *>(dev + m_interval), nullptr};
    args.chunk_offsets = reinterpret_cast<const unsigned long long *>(dev + m_offsets);
    args.n_chunks = n_chunks;
    args.eb = eb;
    args.gap_min_values = GAP_DEFAULT_MIN_VALUES;
    args.gap_results = reinterpret_cast<const GapResult *>(dev + d_gap_results);
    args.gap_long_min_values = long_min_values;
    args.ts_results = nullptr;
    args.n_segments_dev = reinterpret_cast<const unsigned long long *>(dev + d_n_segments);
    args.targets_dev = reinterpret_cast<const EncodeTargets *>(dev + d_targets);
    SplitArgs split{};
    split.piece_base = reinterpret_cast<const unsigned long long *>(dev + m_piece_base);
    split.n_pieces = n_pieces;
    split.piece_points = piece_points;
    split.entry = reinterpret_cast<unsigned int *>(dev + d_entry);
    split.p0 = reinterpret_cast<float *>(dev + d_p0);
    split.p1 = reinterpret_cast<float *>(dev + d_p1);
    split.chunk_left = nullptr;
    const unsigned long long *record_base_dev = reinterpret_cast<const unsigned long long *>(dev + m_record_base);
    ModelRec *records = reinterpret_cast<ModelRec *>(dev + d_records);
    ChunkPlan *plans = reinterpret_cast<ChunkPlan *>(dev + d_plans);
    unsigned long long *segment_base = reinterpret_cast<unsigned long long *>(dev + d_segment_base);
    SegItem *items = reinterpret_cast<SegItem *>(dev + d_items);
    SegSizes *sizes = reinterpret_cast<SegSizes *>(dev + d_sizes);
    unsigned int *error_flag = reinterpret_cast<unsigned int *>(dev + d_error);
    uint32_t *n_gaps = reinterpret_cast<uint32_t *>(dev + d_n_gaps), *gap_ids = reinterpret_cast<uint32_t *>(dev + d_gap_ids);
    unsigned long long *n_segments_dev = reinterpret_cast<unsigned long long *>(dev + d_n_segments);
    GapResult *gap_results = reinterpret_cast<GapResult *>(dev + d_gap_results);
    {
        LaunchTimer timer(ctx, "k_fit_models_wave_pieces");
#define MDB_LAUNCH_PIECES(KIND)                                                                                              \
    hipLaunchKernelGGL((k_fit_models_wave<KIND, false, true>), dim3((uint32_t)n_pieces), dim3(MDB_WAVE), 0, stream, args,    \
                       WaveLeave{}, split, record_base_dev, records, plans, error_flag)
        if (eb.kind == MDB_EB_RELATIVE) MDB_LAUNCH_PIECES(MDB_EB_RELATIVE);
        else if (eb.kind == MDB_EB_ABSOLUTE) MDB_LAUNCH_PIECES(MDB_EB_ABSOLUTE);
        else MDB_LAUNCH_PIECES(MDB_EB_LOSSLESS);
#undef MDB_LAUNCH_PIECES
    }
    {
        LaunchTimer timer(ctx, "k_fit_walk");
        hipLaunchKernelGGL(k_fit_walk, dim3((uint32_t)n_chunks), dim3(MDB_WAVE), 0, stream, args.chunk_offsets, n_chunks, split,
                           record_base_dev, records, plans, error_flag);
    }
    {
        LaunchTimer timer(ctx, "k_fit_plan");
        hipLaunchKernelGGL(k_small_segments, dim3(1), dim3(64), 0, stream, plans, n_chunks, segment_base, n_segments_dev);
        hipLaunchKernelGGL(k_fit_plan, dim3((uint32_t)n_chunks), dim3(MDB_WAVE), 0, stream, args.chunk_offsets,
                           n_chunks, record_base_dev, records, plans, segment_base, items);
    }
    {
        LaunchTimer timer(ctx, "k_fit_gap_size");
        hipLaunchKernelGGL(k_fit_gap_select, dim3((uint32_t)((segment_bound + 255) / 256)), dim3(256), 0, stream, args, items,
                           (uint64_t)0, gap_ids, n_gaps);
        hipLaunchKernelGGL(k_fit_gap<GAP_SIZE>, dim3((uint32_t)gap_bound), dim3(MDB_WAVE), 0, stream, args, items, gap_ids, n_gaps,
                           gap_results, EncodeTargets{});
    }
    LongArgs long_args;
    if (long_bound > 0) {
        long_args.counters = reinterpret_cast<const LongCounters *>(n_gaps); // ([0] is the gap list's counter; zeroed)
        long_args.long_ids = reinterpret_cast<uint32_t *>(dev + d_long_ids);
        long_args.long_block_base = long_args.long_ids + long_bound;
        long_args.long_batch_base = long_args.long_block_base + long_bound;
        long_args.blocks = reinterpret_cast<GapBlock *>(dev + d_long_blocks);
        long_args.batch_notes = reinterpret_cast<GapBatchNote *>(dev + d_long_prefix);
        long_args.batch_breakers = reinterpret_cast<unsigned long long *>(dev + d_long_breakers);
        long_args.long_min_values = long_min_values;
        long_args.block_min_values = block_min_values;
        hipLaunchKernelGGL(k_fit_long_select, dim3((uint32_t)((segment_bound + 255) / 256)), dim3(256), 0, stream, args, items,
                           (uint64_t)0, long_args, reinterpret_cast<LongCounters *>(n_gaps));
        launch_long_sizing(ctx, stream, args, items, long_args, gap_results, (uint32_t)long_bound, (uint32_t)long_blocks_bound);
    }
    const uint32_t segment_blocks = (uint32_t)((segment_bound + FIT_SEGMENT_THREADS - 1) / FIT_SEGMENT_THREADS);
    {
        LaunchTimer timer(ctx, "k_fit_size");
        hipLaunchKernelGGL(k_fit_size, dim3(segment_blocks), dim3(FIT_SEGMENT_THREADS), 0, stream, args, record_base_dev, records,
                           items, (uint64_t)0, sizes);
    }
    {
        LaunchTimer timer(ctx, "k_small_layout");
        hipLaunchKernelGGL(k_small_layout, dim3(1), dim3(1024), 0, stream, sizes, n_segments_dev,
                           reinterpret_cast<unsigned long long *>(dev + d_data_offsets), segment_bound + 1,
                           reinterpret_cast<const unsigned long long *>(dev + d_zero_base), dev + d_blob, blob_capacity, error_flag,
                           reinterpret_cast<EncodeTargets *>(dev + d_targets));
    }
    {
        LaunchTimer timer(ctx, "k_fit_encode");
        hipLaunchKernelGGL(k_fit_gap<GAP_WRITE>, dim3((uint32_t)gap_bound), dim3(MDB_WAVE), 0, stream, args, items, gap_ids, n_gaps,
                           gap_results, EncodeTargets{});
        if (long_bound > 0) launch_long_encode(ctx, stream, args, items, long_args, EncodeTargets{}, (uint32_t)long_blocks_bound);
        hipLaunchKernelGGL(k_fit_encode, dim3(segment_blocks), dim3(FIT_SEGMENT_THREADS), 0, stream, args, record_base_dev, records,
                           items, (uint64_t)0, sizes, EncodeTargets{});
    }
    // ---- one copy down: the header and what usually is the whole block; the rest if there is more ----
    uint8_t *down = host + pinned_down;
    const uint64_t first_copy = std::min<uint64_t>(blob_capacity, 192u << 10);
    MDB_HIP_CHECK(hipMemcpyAsync(down, dev + d_blob, first_copy, hipMemcpyDeviceToHost, stream));
    MDB_HIP_CHECK(hipStreamSynchronize(stream));
    MDB_HIP_CHECK(hipGetLastError());
    const SmallHeader header = *reinterpret_cast<const SmallHeader *>(down);
    if (header.error & ERR_SPLIT_CHAIN) return fail("Internal error: the split fit left a gap in a chunk's model chain.");
    if (header.error & SMALL_OVERFLOW) return 2; // (cannot happen: the block is sized for the worst case)
    if (header.error) return fail("A chunk holds more than 2^31-3 data points.");
    if (header.blob_bytes > first_copy) {
        MDB_HIP_CHECK(hipMemcpyAsync(down + first_copy, dev + d_blob + first_copy, header.blob_bytes - first_copy,
                                     hipMemcpyDeviceToHost, stream));
        MDB_HIP_CHECK(hipStreamSynchronize(stream));
    }
    // ---- the batch in host memory: one allocation, the columns where the kernels packed them ----
    OwnedSegments *owned = new OwnedSegments();
    owned->host_allocs.resize(1);
    owned->host_allocs[0].assign(down, down + header.blob_bytes);
    const uint8_t *blob = owned->host_allocs[0].data();
    const uint64_t n = header.n_segments;
    mdb_segments &seg = owned->c.seg;
    seg.n = n;
    seg.model_type_id = reinterpret_cast<const int8_t *>(blob + header.offsets[0]);
    seg.start_time = reinterpret_cast<const int64_t *>(blob + header.offsets[1]);
    seg.end_time = reinterpret_cast<const int64_t *>(blob + header.offsets[2]);
    seg.min_value = reinterpret_cast<const float *>(blob + header.offsets[3]);
    seg.max_value = reinterpret_cast<const float *>(blob + header.offsets[4]);
    mdb_binview_col *cols[3] = {&seg.timestamps, &seg.values, &seg.residuals};
    for (int c = 0; c < 3; c++) {
        owned->buffer_ptrs[c].push_back(blob + header.offsets[10 + c]);
        owned->buffer_sizes[c].push_back((int64_t)header.data_bytes[c]);
        cols[c]->views = reinterpret_cast<const mdb_view16 *>(blob + header.offsets[7 + c]);
        cols[c]->buffers = owned->buffer_ptrs[c].data();
        cols[c]->buffer_sizes = owned->buffer_sizes[c].data();
        cols[c]->n_buffers = 1;
    }
    owned->c.error = reinterpret_cast<const float *>(blob + header.offsets[5]);
    owned->c.chunk_index = reinterpret_cast<const uint32_t *>(blob + header.offsets[6]);
    owned->c.on_device = 0;
    owned->c.priv_ = owned;
    *out = &owned->c;
    return 0;
}

} // namespace mdb

using namespace mdb;

extern "C" {

int mdb_compress_chunks_dev(mdb_ctx *ctx, const int64_t *ts, const float *values,
                            const uint64_t *chunk_offsets, uint64_t n_chunks,
                            mdb_error_bound error_bound, int64_t regular_start,
                            int64_t regular_interval, const uint64_t *series_first_index,
                            mdb_segments_owned **out) {
    if (!ctx || !out) return fail("ctx and out must not be NULL.");
    if (n_chunks > 0 && (!values || !chunk_offsets)) return fail("values and chunk_offsets must not be NULL.");
    mdb::CallGuard lock(ctx);
    MDB_HIP_CHECK(hipSetDevice(ctx->device));
    return compress_chunks_dev_locked(ctx, ts, values, chunk_offsets, n_chunks, error_bound,
                                      regular_start, regular_interval, series_first_index, out);
}

int mdb_compress_chunks(mdb_ctx *ctx, const int64_t *ts, const float *values,
                        const uint64_t *chunk_offsets, uint64_t n_chunks, mdb_error_bound error_bound,
                        mdb_segments_owned **out) {
    if (!ctx || !out) return fail("ctx and out must not be NULL.");
    if (n_chunks > 0 && !chunk_offsets) return fail("chunk_offsets must not be NULL.");
    const uint64_t total = n_chunks ? chunk_offsets[n_chunks] : 0;
    if (total > 0 && (!ts || !values)) return fail("ts and values must not be NULL.");
    for (uint64_t c = 0; c < n_chunks; c++)
        if (chunk_offsets[c] > chunk_offsets[c + 1]) return fail("chunk_offsets must be non-decreasing.");
    if (n_chunks > 0 && n_chunks <= 4096 && total <= FIT_SMALL_MAX_POINTS) { // (a handful of chunks: fit_few_chunks)
        std::vector<mdb_chunk> list(n_chunks);
        for (uint64_t c = 0; c < n_chunks; c++)
            list[c] = {ts + chunk_offsets[c], values + chunk_offsets[c], chunk_offsets[c + 1] - chunk_offsets[c]};
        const int small = fit_few_chunks(ctx, list.data(), n_chunks, error_bound, out);
        if (small != 2) return small;
    }
    mdb_segments_owned *dev = nullptr;
    int rc = 0;
    {
        mdb::CallGuard lock(ctx);
        MDB_HIP_CHECK(hipSetDevice(ctx->device));
        // Two thirds of what a caller hands over are timestamps, and nearly always they are equally spaced
        // within every chunk: then nothing downstream loads one (k_fit_regular finds that out on the device,
        // for timestamps that are there already). Here they are in host memory and PCIe is what the call
        // waits for, so host threads look while the values cross: per chunk the first timestamp and the
        // interval, or the news that some chunk is not regular - only then do the timestamps cross too.
        std::vector<long long> first(n_chunks), interval(n_chunks);
        std::atomic<bool> irregular{false};
        const unsigned n_workers = (unsigned)std::min<uint64_t>(
            std::max(1u, std::min(16u, std::thread::hardware_concurrency())), std::max<uint64_t>(1, total >> 20));
        std::vector<std::thread> workers;
        if (total > 0) {
            // (contiguous ranges of chunks with about the same number of points)
            uint64_t next_chunk = 0;
            for (unsigned w = 0; w < n_workers; w++) {
                const uint64_t target = chunk_offsets[0] + (total * (w + 1)) / n_workers;
                uint64_t end_chunk = next_chunk;
                while (end_chunk < n_chunks && (chunk_offsets[end_chunk + 1] <= target || w + 1 == n_workers)) end_chunk++;
                const uint64_t begin = next_chunk;
                next_chunk = end_chunk;
                workers.emplace_back([&, begin, end_chunk] {
                    for (uint64_t c = begin; c < end_chunk && !irregular.load(std::memory_order_relaxed); c++) {
                        const int64_t *t = ts + chunk_offsets[c];
                        const uint64_t n = chunk_offsets[c + 1] - chunk_offsets[c];
                        first[c] = n > 0 ? t[0] : 0;
                        const int64_t step = n > 1 ? (int64_t)((uint64_t)t[1] - (uint64_t)t[0]) : 0;
                        interval[c] = step;
                        bool differs = false;
                        for (uint64_t j = 2; j < n; j++) differs |= (int64_t)((uint64_t)t[j] - (uint64_t)t[j - 1]) != step;
                        if (differs) irregular.store(true, std::memory_order_relaxed);
                    }
                });
            }
        }
        auto join = [&] {
            for (auto &worker : workers)
                if (worker.joinable()) worker.join();
        };
        void *dev_values = nullptr, *dev_offsets = nullptr, *dev_ts = nullptr;
        auto upload = [&](ScratchSlot slot, void **dst, const void *src, uint64_t bytes) {
            if (rc) return;
            if (scratch_reserve(ctx, slot, bytes ? bytes : 256, dst)) {
                rc = 1;
                return;
            }
            if (bytes && hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
                rc = fail("hipMemcpy host to device failed.");
        };
        upload(SCRATCH_FIT_IN_VALUES, &dev_values, values, total * 4);
        upload(SCRATCH_FIT_IN_OFFSETS, &dev_offsets, chunk_offsets, (n_chunks + 1) * 8);
        join();
        const bool regular = total > 0 && !irregular.load();
        if (regular) {
            void *dev_first = nullptr;
            std::vector<long long> both(2 * n_chunks);
            std::copy(first.begin(), first.end(), both.begin());
            std::copy(interval.begin(), interval.end(), both.begin() + n_chunks);
            upload(SCRATCH_FIT_IN_TS, &dev_first, both.data(), 16 * n_chunks);
            if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail("stream sync failed.");
            if (!rc)
                rc = compress_chunks_dev_locked(ctx, nullptr, static_cast<const float *>(dev_values),
                                                static_cast<const uint64_t *>(dev_offsets), n_chunks, error_bound, 0, 0,
                                                nullptr, &dev, static_cast<const long long *>(dev_first),
                                                static_cast<const long long *>(dev_first) + n_chunks);
        } else {
            upload(SCRATCH_FIT_IN_TS, &dev_ts, ts, total * 8);
            if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail("stream sync failed.");
            if (!rc)
                rc = compress_chunks_dev_locked(ctx, static_cast<const int64_t *>(dev_ts),
                                                static_cast<const float *>(dev_values),
                                                static_cast<const uint64_t *>(dev_offsets), n_chunks,
                                                error_bound, 0, 0, nullptr, &dev);
        }
        (void)hipStreamSynchronize(ctx->stream);
    }
    if (rc) return 1;
    rc = mdb_segments_download(ctx, dev, out);
    mdb_segments_free(dev);
    return rc;
}

namespace {

// One share of the gather of mdb_compress_chunk_list: a run of chunks whose values are copied into the staging
// block and whose timestamp arrays (those seen for the first time) are looked at.
struct GatherJob {
    const mdb_chunk *chunks;
    const uint64_t *offsets;       // n_chunks + 1
    const uint64_t *same_ts_as;    // per chunk: the earlier chunk with the same timestamp array, or itself
    float *stage_values;           // total points
    int64_t *stage_ts;             // total points, or nullptr while only the values are gathered
    long long *first, *interval;   // per chunk
    unsigned char *irregular;      // per chunk
    std::vector<std::pair<uint64_t, uint64_t>> shares; // [first chunk, last chunk)
};

void gather_share(unsigned index, void *arg) {
    GatherJob &job = *static_cast<GatherJob *>(arg);
    for (uint64_t c = job.shares[index].first; c < job.shares[index].second; c++) {
        const mdb_chunk &chunk = job.chunks[c];
        if (chunk.n == 0) continue;
        if (job.stage_ts) {
            host_copy_streaming(job.stage_ts + job.offsets[c], chunk.ts, 8 * chunk.n);
            continue;
        }
        host_copy_streaming(job.stage_values + job.offsets[c], chunk.values, 4 * chunk.n);
        if (job.same_ts_as[c] != c) continue; // (checked with the chunk that had this array first)
        const int64_t *t = chunk.ts;
        job.first[c] = t[0];
        const int64_t step = chunk.n > 1 ? (int64_t)((uint64_t)t[1] - (uint64_t)t[0]) : 0;
        job.interval[c] = step;
        bool differs = false;
        for (uint64_t j = 2; j < chunk.n; j++) differs |= (int64_t)((uint64_t)t[j] - (uint64_t)t[j - 1]) != step;
        job.irregular[c] = differs ? 1 : 0;
    }
}

} // namespace

int mdb_compress_chunk_list(mdb_ctx *ctx, const mdb_chunk *chunks, uint64_t n_chunks, mdb_error_bound error_bound,
                            mdb_segments_owned **out) {
    if (!ctx || !out) return fail("ctx and out must not be NULL.");
    if (n_chunks > 0 && !chunks) return fail("chunks must not be NULL.");
    std::vector<uint64_t> offsets(n_chunks + 1, 0), same_ts_as(n_chunks);
    {
        std::unordered_map<const int64_t *, uint64_t> seen;
        for (uint64_t c = 0; c < n_chunks; c++) {
            if (chunks[c].n > 0 && (!chunks[c].ts || !chunks[c].values)) return fail("ts and values of a chunk must not be NULL.");
            offsets[c + 1] = offsets[c] + chunks[c].n;
            same_ts_as[c] = c;
            if (chunks[c].n == 0) continue;
            auto found = seen.emplace(chunks[c].ts, c);
            if (!found.second && chunks[found.first->second].n == chunks[c].n) same_ts_as[c] = found.first->second;
        }
    }
    const uint64_t total = offsets[n_chunks];
    {
        // A handful of chunks (the server's call: one finished buffer): upload, kernels, download without a question
        // to the device in between.
        const int small = fit_few_chunks(ctx, chunks, n_chunks, error_bound, out);
        if (small != 2) return small;
    }
    mdb_segments_owned *dev = nullptr;
    int rc = 0;
    // MDB_FIT_DEBUG: where the call's time goes, on stderr
    const bool debug = option_text("MDB_FIT_DEBUG") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto since_start = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
    double t_gathered = 0.0, t_uploaded = 0.0, t_fitted = 0.0, t_in_gather = 0.0, t_in_copy_calls = 0.0;
    {
        mdb::CallGuard lock(ctx);
        MDB_HIP_CHECK(hipSetDevice(ctx->device));
        void *stage = nullptr, *dev_values = nullptr, *dev_offsets = nullptr, *dev_ts = nullptr;
        if (pinned_reserve(ctx, total * 4, &stage)) return 1;
        if (scratch_reserve(ctx, SCRATCH_FIT_IN_VALUES, total * 4, &dev_values)) return 1;
        if (scratch_reserve(ctx, SCRATCH_FIT_IN_OFFSETS, (n_chunks + 1) * 8, &dev_offsets)) return 1;
        std::vector<long long> both(2 * n_chunks, 0);
        std::vector<unsigned char> irregular(n_chunks, 0);
        GatherJob job{chunks, offsets.data(), same_ts_as.data(), static_cast<float *>(stage), nullptr,
                      both.data(), both.data() + n_chunks, irregular.data(), {}};
        MDB_HIP_CHECK(mail_write(ctx, dev_offsets, offsets.data(), (n_chunks + 1) * 8));
        // The gather in slices of about 64 MB of values, each slice by all host threads, and behind every slice its
        // copy to the device: the copy of one slice runs while the threads gather the next.
        const uint64_t slice_points = 16u << 20;
        const unsigned width = host_parallel_width();
        auto gather = [&](auto copy_slice) {
            uint64_t c = 0;
            while (c < n_chunks && !rc) {
                const uint64_t slice_first = c, slice_begin = offsets[c];
                while (c < n_chunks && offsets[c + 1] - slice_begin <= slice_points) c++;
                if (c == slice_first) c++; // (one chunk longer than a slice)
                const uint64_t slice_end = offsets[c], slice_chunks = c - slice_first;
                job.shares.clear();
                // (Shares of about 128 K points, many more than threads: they are handed out one by one, and a thread
                // that shares its core with something else for a while takes fewer.)
                const unsigned n_shares = width <= 1 ? 1u : (unsigned)std::min<uint64_t>(
                    slice_chunks, std::max<uint64_t>(1, (slice_end - slice_begin) >> 17));
                uint64_t next = slice_first;
                for (unsigned w = 0; w < n_shares; w++) {
                    const uint64_t target = slice_begin + (slice_end - slice_begin) * (w + 1) / n_shares;
                    uint64_t last = next;
                    while (last < c && (offsets[last + 1] <= target || w + 1 == n_shares)) last++;
                    job.shares.push_back({next, last});
                    next = last;
                }
                const double t0 = since_start();
                host_parallel((unsigned)job.shares.size(), gather_share, &job);
                const double t1 = since_start();
                copy_slice(slice_begin, slice_end);
                t_in_gather += t1 - t0;
                t_in_copy_calls += since_start() - t1;
            }
        };
        gather([&](uint64_t begin, uint64_t end) {
            if (end > begin && hipMemcpyAsync(static_cast<float *>(dev_values) + begin, job.stage_values + begin, 4 * (end - begin),
                                              hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
                rc = fail("hipMemcpy host to device failed.");
        });
        t_gathered = since_start();
        bool regular = total > 0;
        for (uint64_t c = 0; c < n_chunks; c++) {
            const uint64_t source = same_ts_as[c];
            if (source != c) {
                both[c] = both[source];
                both[n_chunks + c] = both[n_chunks + source];
            }
            regular = regular && !irregular[source];
        }
        if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail("stream sync failed.");
        t_uploaded = since_start();
        if (!rc && regular) {
            void *dev_first = nullptr;
            if (scratch_reserve(ctx, SCRATCH_FIT_IN_TS, 16 * n_chunks, &dev_first)) return 1;
            if (mail_write(ctx, dev_first, both.data(), 16 * n_chunks) != hipSuccess)
                rc = fail("hipMemcpy host to device failed.");
            if (!rc)
                rc = compress_chunks_dev_locked(ctx, nullptr, static_cast<const float *>(dev_values),
                                                static_cast<const uint64_t *>(dev_offsets), n_chunks, error_bound, 0, 0,
                                                nullptr, &dev, static_cast<const long long *>(dev_first),
                                                static_cast<const long long *>(dev_first) + n_chunks);
        } else if (!rc) {
            // Some chunk is not equally spaced: the timestamps cross too (the staging block is free again).
            if (pinned_reserve(ctx, total * 8, &stage)) return 1;
            if (scratch_reserve(ctx, SCRATCH_FIT_IN_TS, total * 8, &dev_ts)) return 1;
            job.stage_ts = static_cast<int64_t *>(stage);
            gather([&](uint64_t begin, uint64_t end) {
                if (end > begin && hipMemcpyAsync(static_cast<int64_t *>(dev_ts) + begin, job.stage_ts + begin, 8 * (end - begin),
                                                  hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
                    rc = fail("hipMemcpy host to device failed.");
            });
            if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail("stream sync failed.");
            if (!rc)
                rc = compress_chunks_dev_locked(ctx, static_cast<const int64_t *>(dev_ts), static_cast<const float *>(dev_values),
                                                static_cast<const uint64_t *>(dev_offsets), n_chunks, error_bound, 0, 0,
                                                nullptr, &dev);
        }
        (void)hipStreamSynchronize(ctx->stream);
        t_fitted = since_start();
    }
    if (rc) return 1;
    rc = mdb_segments_download(ctx, dev, out);
    mdb_segments_free(dev);
    {
        // The call's host-side phases next to the kernels' times (mdb_profile_get, names with "host:").
        mdb::CallGuard lock(ctx);
        if (ctx->profiling) {
            const double t_done = since_start();
            const std::pair<const char *, double> phases[] = {
                {"host:chunk_list_gather", t_in_gather},                  // the threads' copies into page-locked memory (the slices' copies to the device run behind them)
                {"host:chunk_list_upload_tail", t_uploaded - t_gathered}, // what of the copies to the device is left when the last slice is gathered
                {"host:chunk_list_fit", t_fitted - t_uploaded},
                {"host:chunk_list_download", t_done - t_fitted}};
            for (const auto &phase : phases) {
                auto &entry = ctx->kernel_times[phase.first];
                entry.launches += 1;
                entry.total_ms += phase.second;
            }
        }
    }
    if (debug)
        std::fprintf(stderr, "mdb_compress_chunk_list: %llu chunks, %llu points: gathered %.2f ms (%.2f in the threads' copies, %.2f in "
                             "the calls that start the slices' copies to the device), on the device %.2f, fitted %.2f, downloaded %.2f\n",
                     (unsigned long long)n_chunks, (unsigned long long)total, t_gathered, t_in_gather, t_in_copy_calls, t_uploaded,
                     t_fitted, since_start());
    return rc;
}

int mdb_compress_series(mdb_ctx *ctx, const int64_t *ts, const float *values, uint64_t n,
                        mdb_error_bound error_bound, mdb_segments_owned **out) {
    const uint64_t offsets[2] = {0, n};
    return mdb_compress_chunks(ctx, ts, values, offsets, 1, error_bound, out);
}

int mdb_split_and_compress_univariate(mdb_ctx *ctx, const int64_t *ts, const float *const *field_values,
                                      const mdb_error_bound *error_bounds, uint32_t n_fields, uint64_t n,
                                      mdb_segments_owned **out) {
    if (!ctx || !out) return fail("ctx and out must not be NULL.");
    if (n_fields > 0 && (!field_values || !error_bounds)) return fail("field_values and error_bounds must not be NULL.");
    if (n > 0 && !ts) return fail("ts must not be NULL.");
    for (uint32_t f = 0; f < n_fields; f++) {
        out[f] = nullptr;
        if (n > 0 && !field_values[f]) return fail("field_values[f] must not be NULL.");
    }
    // The fields share the timestamps (compression.rs:153-157): uploaded once, every field fitted
    // against the same device array.
    std::vector<mdb_segments_owned *> on_device(n_fields, nullptr);
    void *dev_ts = nullptr, *dev_values = nullptr, *dev_offsets = nullptr;
    int rc = 0;
    {
        mdb::CallGuard lock(ctx);
        MDB_HIP_CHECK(hipSetDevice(ctx->device));
        const uint64_t offsets[2] = {0, n};
        if (hipMalloc(&dev_ts, n ? 8 * n : 256) != hipSuccess || hipMalloc(&dev_values, n ? 4 * n : 256) != hipSuccess ||
            hipMalloc(&dev_offsets, 16) != hipSuccess)
            rc = fail("hipMalloc failed.");
        if (!rc && n && hipMemcpyAsync(dev_ts, ts, 8 * n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
            rc = fail("hipMemcpy host to device failed.");
        if (!rc && hipMemcpyAsync(dev_offsets, offsets, 16, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
            rc = fail("hipMemcpy host to device failed.");
        if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail("stream sync failed.");
        for (uint32_t f = 0; f < n_fields && !rc; f++) {
            if (n && hipMemcpyAsync(dev_values, field_values[f], 4 * n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
                rc = fail("hipMemcpy host to device failed.");
            if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail("stream sync failed.");
            if (!rc)
                rc = compress_chunks_dev_locked(ctx, static_cast<const int64_t *>(dev_ts),
                                                static_cast<const float *>(dev_values),
                                                static_cast<const uint64_t *>(dev_offsets), n ? 1 : 0, error_bounds[f],
                                                0, 0, nullptr, &on_device[f]);
        }
        (void)hipStreamSynchronize(ctx->stream);
        if (dev_ts) (void)hipFree(dev_ts);
        if (dev_values) (void)hipFree(dev_values);
        if (dev_offsets) (void)hipFree(dev_offsets);
    }
    for (uint32_t f = 0; f < n_fields && !rc; f++) rc = mdb_segments_download(ctx, on_device[f], &out[f]);
    for (uint32_t f = 0; f < n_fields; f++) {
        if (on_device[f]) mdb_segments_free(on_device[f]);
        if (rc && out[f]) {
            mdb_segments_free(out[f]);
            out[f] = nullptr;
        }
    }
    return rc ? 1 : 0;
}

} // extern "C"
