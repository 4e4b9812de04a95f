/*
 * mdb_oracle.cpp - CPU oracle for the ModelarDB model-compression / grid / segment-aggregate path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE (see mdb_oracle.h). A from-scratch C++ restatement of the
 * algorithms in the reference's crates/modelardb_compression (paths below are relative to
 * crates/modelardb_compression/src/). Build: see oracle/Makefile; always with -ffp-contract=off
 * because the reference (Rust) never fuses a*b+c.
 *
 * Deliberate, documented choices where the reference is platform dependent or undefined:
 *  - f32/f64 min/max follow IEEE minNum/maxNum and keep the FIRST operand on ties (what LLVM emits
 *    for Rust's f32::min on x86-64), so min(-0,+0) is -0. Only the sign of a stored zero depends
 *    on this.
 *  - log2 in rewrite_least_mantissa_bits (models/macaque_v.rs:185) is evaluated as
 *    (float)log2((double)x), i.e. a correctly rounded log2f. Rust calls the platform's log2f.
 *  - a negative rewrite position (shift overflow, models/macaque_v.rs:333-336) is
 *    clamped to 0 instead of wrapping; unreachable for relative bounds.
 *  - malformed segments make the reference panic; here they become error returns.
 */
#include "mdb_oracle.h"

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <limits>
#include <mutex>
#include <string>
#include <thread>
#include <pthread.h>
#include <sched.h>
#include <vector>

namespace {

thread_local std::string g_last_error;

int fail(const char *message) {
    g_last_error = message;
    return 1;
}

/* ---------------------------------------------------------------------------------------------
 * Scalar helpers.
 * ------------------------------------------------------------------------------------------- */

inline uint32_t f32_bits(float v) {
    uint32_t u;
    std::memcpy(&u, &v, 4);
    return u;
}

inline float f32_from_bits(uint32_t u) {
    float v;
    std::memcpy(&v, &u, 4);
    return v;
}

/* Rust f32::min / f32::max (minNum / maxNum), first operand kept on ties. */
template <typename T> inline T min_num(T a, T b) {
    if (std::isnan(a)) return b;
    return (b < a) ? b : a;
}

template <typename T> inline T max_num(T a, T b) {
    if (std::isnan(a)) return b;
    return (b > a) ? b : a;
}

/* models/mod.rs:92-95 */
inline bool equal_or_nan(double a, double b) { return a == b || (std::isnan(a) && std::isnan(b)); }

/* models/mod.rs:53-80 */
inline bool within_error_bound(mdb_error_bound eb, float real_value, float approximate_value) {
    if (equal_or_nan((double)real_value, (double)approximate_value)) return true;
    switch (eb.kind) {
    case MDB_EB_ABSOLUTE:
        return std::fabs(real_value - approximate_value) <= eb.value;
    case MDB_EB_RELATIVE: {
        float difference = real_value - approximate_value;
        float result = std::fabs(difference / real_value);
        return (result * 100.0f) <= eb.value;
    }
    default:
        return false;
    }
}

/* models/mod.rs:83-90 */
inline double max_allowed_deviation(mdb_error_bound eb, double value) {
    switch (eb.kind) {
    case MDB_EB_ABSOLUTE:
        return (double)eb.value * 0.99;
    case MDB_EB_RELATIVE:
        return std::fabs(value * ((double)eb.value / 100.1));
    default:
        return 0.0;
    }
}

/* ---------------------------------------------------------------------------------------------
 * Bit packing, MSB first (models/bits.rs:25-174).
 * ------------------------------------------------------------------------------------------- */

struct BitWriter {
    std::vector<uint8_t> bytes;
    uint8_t current = 0;
    unsigned free_bits = 8;

    void put(uint64_t bits, unsigned count) {
        while (count > 0) {
            unsigned take = count < free_bits ? count : free_bits;
            uint64_t chunk = (bits >> (count - take)) & ((1ull << take) - 1ull);
            current |= (uint8_t)(chunk << (free_bits - take));
            free_bits -= take;
            count -= take;
            if (free_bits == 0) {
                bytes.push_back(current);
                current = 0;
                free_bits = 8;
            }
        }
    }

    /* bits.rs:145-147: "empty" means no complete byte has been stored yet. */
    bool no_full_byte() const { return bytes.empty(); }

    std::vector<uint8_t> finish() {
        if (free_bits != 8) bytes.push_back(current);
        return std::move(bytes);
    }

    std::vector<uint8_t> finish_with_one_bits() {
        if (free_bits != 8) put((1ull << free_bits) - 1ull, free_bits);
        return finish();
    }
};

struct BitReader {
    const uint8_t *bytes;
    uint64_t nbytes;
    uint64_t next_bit = 0;
    bool overrun = false;

    BitReader(const uint8_t *b, uint64_t n) : bytes(b), nbytes(n) {}

    bool exhausted() const { return (next_bit / 8) == nbytes; }
    uint64_t remaining() const { return 8 * nbytes - next_bit; }

    uint64_t get(unsigned count) {
        if (next_bit + count > 8 * nbytes) { /* the reference would panic on the slice index */
            overrun = true;
            next_bit = 8 * nbytes;
            return 0;
        }
        uint64_t value = 0;
        while (count > 0) {
            unsigned offset = (unsigned)(next_bit & 7);
            unsigned available = 8 - offset;
            unsigned take = count < available ? count : available;
            uint8_t byte = bytes[next_bit >> 3];
            value = (value << take) | ((uint64_t)(byte >> (available - take)) & ((1u << take) - 1u));
            next_bit += take;
            count -= take;
        }
        return value;
    }

    bool bit() { return get(1) == 1; }
};

/* ---------------------------------------------------------------------------------------------
 * MacaqueTS (models/timestamps.rs:56-292).
 * ------------------------------------------------------------------------------------------- */

/* timestamps.rs:77-95 */
bool uncompressed_timestamps_regular(const int64_t *ts, uint64_t n) {
    if (n < 2) return true;
    int64_t expected = ts[1] - ts[0];
    for (uint64_t i = 1; i < n; i++)
        if (ts[i] - ts[i - 1] != expected) return false;
    return true;
}

/* timestamps.rs:56-155 */
std::vector<uint8_t> compress_residual_timestamps(const int64_t *ts, uint64_t n) {
    if (n <= 2) return {};
    if (uncompressed_timestamps_regular(ts, n)) {
        /* timestamps.rs:99-108: length, big endian, leading zero BYTES stripped, top bit zero. */
        unsigned significant = 64 - (unsigned)__builtin_clzll(n);
        unsigned bits_to_write = significant + 1;
        unsigned bytes_to_write = (bits_to_write + 7) / 8;
        std::vector<uint8_t> out(bytes_to_write);
        for (unsigned i = 0; i < bytes_to_write; i++)
            out[i] = (uint8_t)(n >> (8 * (bytes_to_write - 1 - i)));
        return out;
    }
    /* timestamps.rs:113-155 */
    BitWriter w;
    w.put(1, 1);
    uint64_t last_timestamp = (uint64_t)ts[0];
    uint64_t last_delta = 0;
    for (uint64_t i = 1; i + 1 < n; i++) {
        uint64_t delta = (uint64_t)ts[i] - last_timestamp;
        int64_t dod = (int64_t)(delta - last_delta);
        if (dod == 0) {
            w.put(0, 1);
        } else if (dod >= -63 && dod <= 64) {
            w.put(0b10, 2);
            w.put((uint64_t)dod, 7);
        } else if (dod >= -255 && dod <= 256) {
            w.put(0b110, 3);
            w.put((uint64_t)dod, 9);
        } else if (dod >= -2047 && dod <= 2048) {
            w.put(0b1110, 4);
            w.put((uint64_t)dod, 12);
        } else if (dod >= -2147483647LL && dod <= 2147483648LL) {
            w.put(0b11110, 5);
            w.put((uint64_t)dod, 32);
        } else {
            w.put(0b11111, 5);
            w.put((uint64_t)dod, 64);
        }
        last_delta = delta;
        last_timestamp = (uint64_t)ts[i];
    }
    return w.finish_with_one_bits();
}

/* timestamps.rs:199-202 */
inline bool compressed_timestamps_regular(const uint8_t *bytes, uint64_t n) {
    return n == 0 || (bytes[0] & 128) == 0;
}

/* Big endian integer of up to 8 bytes (models/mod.rs:105-111, timestamps.rs:213-218). */
bool regular_length(const uint8_t *bytes, uint64_t n, uint64_t *length) {
    if (n > 8) return false; /* copy_from_slice would panic */
    uint64_t value = 0;
    for (uint64_t i = 0; i < n; i++) value = (value << 8) | bytes[i];
    *length = value;
    return true;
}

/* Visitor based decoder so len()/sum()/grid() share one loop (timestamps.rs:163-292). `emit` is
 * called once per timestamp in order. Returns false for streams the reference would panic on. */
template <typename Emit>
bool decompress_all_timestamps(int64_t start_time, int64_t end_time, const uint8_t *bytes,
                               uint64_t nbytes, Emit emit) {
    if (nbytes == 0 && start_time == end_time) {
        emit(start_time);
        return true;
    }
    if (nbytes == 0) {
        emit(start_time);
        emit(end_time);
        return true;
    }
    if (compressed_timestamps_regular(bytes, nbytes)) {
        /* timestamps.rs:207-223 */
        uint64_t length;
        if (!regular_length(bytes, nbytes, &length)) return false;
        if (length < 2) return false;    /* division by zero in the reference */
        uint64_t span = (uint64_t)(end_time - start_time); /* `as usize` */
        uint64_t interval = span / (length - 1);
        if (interval == 0) return false; /* step_by(0) panics */
        if (end_time < start_time) return true; /* empty range */
        for (int64_t t = start_time;; t += (int64_t)interval) {
            emit(t);
            if ((uint64_t)(end_time - t) < interval) break;
        }
        return true;
    }
    /* timestamps.rs:228-292 */
    emit(start_time);
    BitReader r(bytes, nbytes);
    r.bit();
    uint64_t last_delta = 0;
    int64_t timestamp = start_time;
    while (!r.exhausted()) {
        unsigned ones = 0;
        while (ones < 5 && !r.exhausted() && r.bit()) ones++;
        if (ones != 0 && r.remaining() < 7) break;
        if (ones != 0) {
            static const unsigned widths[6] = {0, 7, 9, 12, 32, 64};
            unsigned width = widths[ones];
            if (r.remaining() < width) return false;
            uint64_t encoded = r.get(width);
            uint64_t dod = encoded;
            if (width < 64 && encoded > (1ull << (width - 1))) dod = encoded | (~0ull << width);
            last_delta += dod; /* wrapping_add */
        }
        timestamp = (int64_t)((uint64_t)timestamp + last_delta);
        emit(timestamp);
    }
    emit(end_time);
    return true;
}

/* Number of timestamps decompress_all_timestamps() produces. */
bool decompressed_timestamp_count(int64_t start_time, int64_t end_time, const uint8_t *bytes,
                                  uint64_t nbytes, uint64_t *count) {
    uint64_t c = 0;
    bool ok = decompress_all_timestamps(start_time, end_time, bytes, nbytes, [&](int64_t) { c++; });
    *count = c;
    return ok;
}

/* models/mod.rs:98-124 */
bool segment_len(int64_t start_time, int64_t end_time, const uint8_t *bytes, uint64_t nbytes,
                 uint64_t *length) {
    if (nbytes == 0) {
        *length = (start_time == end_time) ? 1 : 2;
        return true;
    }
    if (compressed_timestamps_regular(bytes, nbytes)) return regular_length(bytes, nbytes, length);
    return decompressed_timestamp_count(start_time, end_time, bytes, nbytes, length);
}

/* ---------------------------------------------------------------------------------------------
 * PMC-Mean (models/pmc_mean.rs:31-108).
 * ------------------------------------------------------------------------------------------- */

struct PmcMean {
    mdb_error_bound eb;
    float min_value = std::numeric_limits<float>::quiet_NaN();
    float max_value = std::numeric_limits<float>::quiet_NaN();
    double sum_of_values = 0.0;
    uint64_t length = 0;

    explicit PmcMean(mdb_error_bound e) : eb(e) {}

    /* pmc_mean.rs:58-75 */
    bool fit_value(float value) {
        float next_min = min_num(min_value, value);
        float next_max = max_num(max_value, value);
        double next_sum = sum_of_values + (double)value;
        uint64_t next_length = length + 1;
        float average = (float)(next_sum / (double)next_length);
        if (within_error_bound(eb, next_min, average) && within_error_bound(eb, next_max, average)) {
            min_value = next_min;
            max_value = next_max;
            sum_of_values = next_sum;
            length = next_length;
            return true;
        }
        return false;
    }

    /* pmc_mean.rs:83-87 */
    float bytes_per_value() const {
        return (float)MDB_COMPRESSED_METADATA_SIZE_IN_BYTES / (float)length;
    }

    /* pmc_mean.rs:91-93 */
    float model() const { return (float)(sum_of_values / (double)length); }
};

/* ---------------------------------------------------------------------------------------------
 * Swing (models/swing.rs:34-340).
 * ------------------------------------------------------------------------------------------- */

struct Line {
    double slope;
    double intercept;
};

/* swing.rs:323-340 */
inline Line line_through(int64_t t0, double v0, int64_t t1, double v1) {
    if (equal_or_nan(v0, v1)) return {0.0, v0};
    double slope = (v1 - v0) / (double)(t1 - t0);
    double intercept = v0 - slope * (double)t0;
    return {slope, intercept};
}

struct Swing {
    mdb_error_bound eb;
    int64_t start_time = 0;
    int64_t end_time = 0;
    double first_value = std::numeric_limits<double>::quiet_NaN();
    Line upper{std::numeric_limits<double>::quiet_NaN(), std::numeric_limits<double>::quiet_NaN()};
    Line lower{std::numeric_limits<double>::quiet_NaN(), std::numeric_limits<double>::quiet_NaN()};
    double mse_numerator = 0.0;
    double mse_denominator = 0.0;
    uint64_t length = 0;

    explicit Swing(mdb_error_bound e) : eb(e) {}

    /* swing.rs:101-198 */
    bool fit_data_point(int64_t timestamp, float value32) {
        double value = (double)value32;
        double deviation = max_allowed_deviation(eb, value);
        if (length == 0) {
            start_time = timestamp;
            end_time = timestamp;
            first_value = value;
            length = 1;
            return true;
        }
        if (!std::isfinite(first_value) || !std::isfinite(value)) {
            if (!equal_or_nan(first_value, value)) return false;
            end_time = timestamp;
            upper = {value, value};
            lower = {value, value};
            length += 1;
            return true;
        }
        if (length == 1) {
            end_time = timestamp;
            upper = line_through(start_time, first_value, timestamp, value + deviation);
            lower = line_through(start_time, first_value, timestamp, value - deviation);
            length += 1;
            return true;
        }
        double upper_approximation = upper.slope * (double)timestamp + upper.intercept;
        double lower_approximation = lower.slope * (double)timestamp + lower.intercept;
        if (upper_approximation + deviation < value || lower_approximation - deviation > value)
            return false;
        end_time = timestamp;
        if (upper_approximation - deviation > value)
            upper = line_through(start_time, first_value, timestamp, value + deviation);
        if (lower_approximation + deviation < value)
            lower = line_through(start_time, first_value, timestamp, value - deviation);
        /* swing.rs:212-228 */
        if (!equal_or_nan(first_value, value)) {
            double dt = (double)(timestamp - start_time);
            mse_numerator += (value - first_value) * dt;
            mse_denominator += dt * dt;
        } else {
            mse_numerator += 0.0;
            mse_denominator += 0.0;
        }
        length += 1;
        return true;
    }

    /* swing.rs:236-239 */
    float bytes_per_value() const {
        return ((float)MDB_COMPRESSED_METADATA_SIZE_IN_BYTES + 1.0f) / (float)length;
    }

    /* swing.rs:246-259 */
    void model(float *first, float *last) const {
        double projected = mse_numerator / mse_denominator;
        double slope = max_num(lower.slope, min_num(projected, upper.slope));
        double last_value = slope * (double)(end_time - start_time) + first_value;
        *first = (float)first_value;
        *last = (float)last_value;
    }
};

/* ---------------------------------------------------------------------------------------------
 * MacaqueV (models/macaque_v.rs:39-336).
 * ------------------------------------------------------------------------------------------- */

/* (float)log2((double)x): the correctly rounded f32 base-2 logarithm (see file header). */
inline float log2_f32(float x) { return (float)std::log2((double)x); }

struct MacaqueV {
    mdb_error_bound eb;
    float min_value = std::numeric_limits<float>::quiet_NaN();
    float max_value = std::numeric_limits<float>::quiet_NaN();
    float last_value = 0.0f;
    uint8_t last_leading_zero_bits = 255;
    uint8_t last_trailing_zero_bits = 0;
    BitWriter out;
    uint64_t length = 0;

    explicit MacaqueV(mdb_error_bound e) : eb(e) {}

    /* macaque_v.rs:199-204 */
    void update_min_max_and_last_value(float value) {
        min_value = min_num(min_value, value);
        max_value = max_num(max_value, value);
        last_value = value;
        length += 1;
    }

    /* macaque_v.rs:168-196, 326-336 */
    float rewrite_least_mantissa_bits(float value) const {
        if (std::fabs(value) == 0.0f || std::isnan(value) || std::isinf(value)) return value;
        uint32_t bits = f32_bits(value);
        float abs_error_bound = (float)max_allowed_deviation(eb, (double)value);
        int exponent = (int)((bits >> 23) & 0xff) - 127;
        float factorized_epsilon = abs_error_bound / std::ldexp(1.0f, exponent); /* 2f32.powi(e) */
        /* Rust's `as i32` saturates (and maps NaN to 0); a plain C cast of inf would be undefined.
         * A non-finite magnitude arises when the deviation underflows to 0 for subnormal values. */
        const float magnitude = std::floor(std::fabs(log2_f32(factorized_epsilon)));
        long long saturated = magnitude != magnitude ? 0
                              : magnitude >= 2147483648.0f ? 2147483647LL
                              : magnitude <= -2147483648.0f ? -2147483648LL : (long long)magnitude;
        long long wide_position = 23 - saturated;
        int position = wide_position < -2147483647LL ? -2147483647 : (int)wide_position;
        auto rewrite = [](uint32_t b, int pos) {
            if (pos < 0) pos = 0; /* SURVEY A.6 Q4 */
            if (pos > 31) return 0u;
            return b & (0xFFFFFFFFu << pos);
        };
        float rewritten = f32_from_bits(rewrite(bits, position));
        if (!within_error_bound(eb, value, rewritten)) {
            position -= 1;
            rewritten = f32_from_bits(rewrite(bits, position));
        }
        return rewritten;
    }

    /* macaque_v.rs:100-164 */
    void compress_value_xor_last_value(float value) {
        if (eb.kind != MDB_EB_LOSSLESS) {
            if (within_error_bound(eb, value, last_value))
                value = last_value;
            else
                value = rewrite_least_mantissa_bits(value);
        }
        uint32_t x = f32_bits(value) ^ f32_bits(last_value);
        if (x == 0) {
            out.put(1, 1);
            out.put(0, 1);
        } else {
            uint8_t leading = (uint8_t)__builtin_clz(x);
            uint8_t trailing = (uint8_t)__builtin_ctz(x);
            if (leading >= last_leading_zero_bits && trailing >= last_trailing_zero_bits) {
                out.put(0, 1);
                unsigned meaningful = 32u - last_leading_zero_bits - last_trailing_zero_bits;
                out.put((uint64_t)(x >> last_trailing_zero_bits), meaningful);
            } else {
                out.put(1, 1);
                out.put(1, 1);
                out.put(leading, 5);
                unsigned meaningful = 32u - leading - trailing;
                out.put(meaningful, 6);
                out.put((uint64_t)(x >> trailing), meaningful);
                last_leading_zero_bits = leading;
                last_trailing_zero_bits = trailing;
            }
        }
        update_min_max_and_last_value(value);
    }

    /* macaque_v.rs:76-88 */
    void compress_values(const float *values, uint64_t n) {
        for (uint64_t i = 0; i < n; i++) {
            if (out.no_full_byte()) {
                out.put(f32_bits(values[i]), 32);
                update_min_max_and_last_value(values[i]);
            } else {
                compress_value_xor_last_value(values[i]);
            }
        }
    }

    /* macaque_v.rs:92-97 */
    void compress_values_without_first(const float *values, uint64_t n, float model_last_value) {
        last_value = model_last_value;
        for (uint64_t i = 0; i < n; i++) compress_value_xor_last_value(values[i]);
    }
};

/* Decoder shared by sum() and grid() (macaque_v.rs:220-323). */
template <typename Emit>
bool macaque_v_decode(const uint8_t *bytes, uint64_t nbytes, uint64_t count, bool seeded,
                      float seed, Emit emit) {
    if (nbytes == 0) return false; /* BitReader::try_new(..).unwrap() */
    BitReader r(bytes, nbytes);
    uint8_t leading = 255;
    uint8_t trailing = 0;
    uint32_t last;
    uint64_t remaining_values = count;
    if (seeded) {
        last = f32_bits(seed);
    } else {
        last = (uint32_t)r.get(32);
        emit(f32_from_bits(last));
        if (remaining_values == 0) return false; /* `length - 1` underflows in the reference */
        remaining_values -= 1;
    }
    for (uint64_t i = 0; i < remaining_values; i++) {
        if (r.overrun) return false; /* the reference panics at the first bit past the end (bits.rs:61-82) */
        bool decode_value = true;
        if (r.bit()) {
            if (r.bit()) {
                leading = (uint8_t)r.get(5);
                uint8_t meaningful = (uint8_t)r.get(6);
                trailing = (uint8_t)(32 - meaningful - leading);
            } else {
                decode_value = false; /* control bits `10`: the value repeats */
            }
        }
        if (decode_value) {
            uint8_t meaningful = (uint8_t)(32 - leading - trailing);
            if (meaningful > 32 || trailing > 31) return false;
            uint32_t value = (uint32_t)r.get(meaningful);
            value <<= trailing;
            value ^= last;
            last = value;
        }
        emit(f32_from_bits(last));
    }
    return !r.overrun;
}

/* ---------------------------------------------------------------------------------------------
 * The values column of PMC-Mean / Swing segments (types.rs:283-407).
 * ------------------------------------------------------------------------------------------- */

void put_le_f32(std::vector<uint8_t> &out, float v) {
    uint32_t u = f32_bits(v);
    for (int i = 0; i < 4; i++) out.push_back((uint8_t)(u >> (8 * i)));
}

float get_le_f32(const uint8_t *p) {
    uint32_t u = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    return f32_from_bits(u);
}

/* types.rs:283-303 */
std::vector<uint8_t> encode_values_for_pmc_mean(float min_value, float max_value, float rmin,
                                                float rmax) {
    std::vector<uint8_t> values;
    if (min_value > rmin) {
        if (max_value >= rmax)
            values.push_back(1);
        else
            put_le_f32(values, min_value);
    }
    return values;
}

/* types.rs:307-321 */
bool decode_values_for_pmc_mean(float min_value, float max_value, const uint8_t *values,
                                uint64_t n, float *out) {
    if (n == 0) *out = min_value;
    else if (n == 1) *out = max_value;
    else if (n == 4) *out = get_le_f32(values);
    else return false;
    return true;
}

/* types.rs:325-370 */
std::vector<uint8_t> encode_values_for_swing(float min_value, float max_value,
                                             bool min_value_is_first, float rmin, float rmax) {
    std::vector<uint8_t> values;
    if (rmin < min_value && max_value < rmax) {
        put_le_f32(values, min_value_is_first ? min_value : max_value);
        put_le_f32(values, min_value_is_first ? max_value : min_value);
    } else if (rmin < min_value) {
        values.push_back(min_value_is_first ? 0 : 1);
        put_le_f32(values, min_value);
    } else if (max_value < rmax) {
        values.push_back(min_value_is_first ? 2 : 3);
        put_le_f32(values, max_value);
    } else if (!min_value_is_first) {
        values.push_back(0);
    }
    return values;
}

/* types.rs:374-407 */
bool decode_values_for_swing(float min_value, float max_value, const uint8_t *values, uint64_t n,
                             float *first, float *last) {
    if (n == 0) {
        *first = min_value;
        *last = max_value;
    } else if (n == 1) {
        *first = max_value;
        *last = min_value;
    } else if (n == 5) {
        float value = get_le_f32(values + 1);
        switch (values[0]) {
        case 0: *first = value; *last = max_value; break;
        case 1: *first = max_value; *last = value; break;
        case 2: *first = min_value; *last = value; break;
        case 3: *first = value; *last = min_value; break;
        default: return false;
        }
    } else if (n == 8) {
        *first = get_le_f32(values);
        *last = get_le_f32(values + 4);
    } else {
        return false;
    }
    return true;
}

/* ---------------------------------------------------------------------------------------------
 * sum() and grid() of one segment (models/mod.rs:129-284, pmc_mean.rs:98-108, swing.rs:264-319).
 * ------------------------------------------------------------------------------------------- */

inline uint64_t residuals_length(const uint8_t *residuals, uint64_t n) {
    return n == 0 ? 0 : residuals[n - 1];
}

/* swing.rs:264-300. NOTE: called with the SEGMENT's end_time. */
bool swing_sum(int64_t start_time, int64_t end_time, const uint8_t *ts, uint64_t ts_len,
               float first_value, float last_value, uint64_t n_residuals, float *out) {
    Line line = line_through(start_time, (double)first_value, end_time, (double)last_value);
    if (compressed_timestamps_regular(ts, ts_len)) {
        double first = line.slope * (double)start_time + line.intercept;
        double last = line.slope * (double)end_time + line.intercept;
        double average = (first + last) / 2.0;
        uint64_t total;
        if (!segment_len(start_time, end_time, ts, ts_len, &total) || total < n_residuals)
            return false;
        uint64_t length = total - n_residuals;
        *out = (float)(average * (double)length);
        return true;
    }
    std::vector<int64_t> timestamps;
    if (!decompress_all_timestamps(start_time, end_time, ts, ts_len,
                                   [&](int64_t t) { timestamps.push_back(t); }))
        return false;
    if (timestamps.size() < n_residuals) return false;
    double sum = 0.0;
    for (uint64_t i = 0; i < timestamps.size() - n_residuals; i++)
        sum += line.slope * (double)timestamps[i] + line.intercept;
    *out = (float)sum;
    return true;
}

/* models/mod.rs:129-184 */
bool segment_sum(int8_t model_type_id, int64_t start_time, int64_t end_time, const uint8_t *ts,
                 uint64_t ts_len, float min_value, float max_value, const uint8_t *values,
                 uint64_t values_len, const uint8_t *residuals, uint64_t residuals_len, float *out) {
    uint64_t n_residuals = residuals_length(residuals, residuals_len);
    uint64_t total;
    if (!segment_len(start_time, end_time, ts, ts_len, &total) || total < n_residuals) return false;
    uint64_t model_length = total - n_residuals;
    float model_last_value;
    float model_sum;
    switch (model_type_id) {
    case MDB_PMC_MEAN_ID: {
        float value;
        if (!decode_values_for_pmc_mean(min_value, max_value, values, values_len, &value))
            return false;
        model_last_value = value;
        model_sum = (float)model_length * value; /* pmc_mean.rs:98-100 */
        break;
    }
    case MDB_SWING_ID: {
        float first, last;
        if (!decode_values_for_swing(min_value, max_value, values, values_len, &first, &last))
            return false;
        model_last_value = last;
        if (!swing_sum(start_time, end_time, ts, ts_len, first, last, n_residuals, &model_sum))
            return false;
        break;
    }
    case MDB_MACAQUE_V_ID: {
        model_last_value = std::numeric_limits<float>::quiet_NaN();
        float sum = 0.0f;
        bool first_value = true;
        if (!macaque_v_decode(values, values_len, model_length, false, 0.0f, [&](float v) {
                /* macaque_v.rs:228-235: the sum starts AS the first value, not 0 + first. */
                if (first_value) { sum = v; first_value = false; } else { sum += v; }
            }))
            return false;
        model_sum = sum;
        break;
    }
    default:
        return false;
    }
    if (residuals_len == 0) {
        *out = model_sum;
        return true;
    }
    float residuals_sum = 0.0f;
    if (!macaque_v_decode(residuals, residuals_len - 1, n_residuals, true, model_last_value,
                          [&](float v) { residuals_sum += v; }))
        return false;
    *out = model_sum + residuals_sum;
    return true;
}

/* models/mod.rs:190-251. Appends to ts_out / val_out. */
bool segment_grid(int8_t model_type_id, int64_t start_time, int64_t end_time, const uint8_t *ts,
                  uint64_t ts_len, float min_value, float max_value, const uint8_t *values,
                  uint64_t values_len, const uint8_t *residuals, uint64_t residuals_len,
                  std::vector<int64_t> &ts_out, std::vector<float> &val_out) {
    uint64_t n_residuals = residuals_length(residuals, residuals_len);
    size_t first_index = ts_out.size();
    if (!decompress_all_timestamps(start_time, end_time, ts, ts_len,
                                   [&](int64_t t) { ts_out.push_back(t); }))
        return false;
    uint64_t total = ts_out.size() - first_index;
    if (total < n_residuals) return false;
    uint64_t n_model = total - n_residuals;
    const int64_t *model_ts = ts_out.data() + first_index;
    switch (model_type_id) {
    case MDB_PMC_MEAN_ID: {
        float value;
        if (!decode_values_for_pmc_mean(min_value, max_value, values, values_len, &value))
            return false;
        for (uint64_t i = 0; i < n_model; i++) val_out.push_back(value); /* pmc_mean.rs:104-108 */
        break;
    }
    case MDB_SWING_ID: {
        float first, last;
        if (!decode_values_for_swing(min_value, max_value, values, values_len, &first, &last))
            return false;
        if (n_model == 0) return false; /* expect("Model should represent at least one value.") */
        int64_t model_end_time = model_ts[n_model - 1];
        Line line = line_through(start_time, (double)first, model_end_time, (double)last);
        for (uint64_t i = 0; i < n_model; i++) /* swing.rs:315-318 */
            val_out.push_back((float)(line.slope * (double)model_ts[i] + line.intercept));
        break;
    }
    case MDB_MACAQUE_V_ID:
        if (!macaque_v_decode(values, values_len, n_model, false, 0.0f,
                              [&](float v) { val_out.push_back(v); }))
            return false;
        break;
    default:
        return false;
    }
    if (residuals_len != 0) {
        if (val_out.empty()) return false;
        float model_last_value = val_out.back(); /* last RECONSTRUCTED value */
        if (!macaque_v_decode(residuals, residuals_len - 1, n_residuals, true, model_last_value,
                              [&](float v) { val_out.push_back(v); }))
            return false;
    }
    return true;
}

/* ---------------------------------------------------------------------------------------------
 * Owned segment batches with Arrow BinaryView columns.
 * ------------------------------------------------------------------------------------------- */

struct BinaryViewBuilder {
    std::vector<mdb_view16> views;
    std::vector<uint8_t> data;

    void append(const uint8_t *bytes, uint64_t n) {
        mdb_view16 view;
        std::memset(&view, 0, sizeof(view));
        view.length = (int32_t)n;
        if (n <= 12) {
            if (n) std::memcpy(view.u.inlined, bytes, n);
        } else {
            std::memcpy(view.u.ref.prefix, bytes, 4);
            view.u.ref.buffer_index = 0;
            view.u.ref.offset = (int32_t)data.size();
            data.insert(data.end(), bytes, bytes + n);
        }
        views.push_back(view);
    }
};

struct OwnedBatch {
    std::vector<int8_t> model_type_id;
    std::vector<int64_t> start_time, end_time;
    std::vector<float> min_value, max_value, error;
    std::vector<uint32_t> chunk_index;
    BinaryViewBuilder timestamps, values, residuals;
    const uint8_t *buffer_ptrs[3];
    int64_t buffer_sizes[3];
    mdb_segments_owned c;

    /* types.rs:468-489 */
    void append(int8_t id, int64_t start, int64_t end, const std::vector<uint8_t> &ts, float mn,
                float mx, const std::vector<uint8_t> &vals, const std::vector<uint8_t> &res,
                uint32_t chunk) {
        model_type_id.push_back(id);
        start_time.push_back(start);
        end_time.push_back(end);
        timestamps.append(ts.data(), ts.size());
        min_value.push_back(mn);
        max_value.push_back(mx);
        values.append(vals.data(), vals.size());
        residuals.append(res.data(), res.size());
        error.push_back(std::numeric_limits<float>::quiet_NaN());
        chunk_index.push_back(chunk);
    }

    void append_all(const OwnedBatch &other) {
        for (size_t i = 0; i < other.model_type_id.size(); i++) {
            auto bytes = [&](const BinaryViewBuilder &b, size_t row) {
                const mdb_view16 &v = b.views[row];
                const uint8_t *p = v.length <= 12 ? v.u.inlined : b.data.data() + v.u.ref.offset;
                return std::vector<uint8_t>(p, p + v.length);
            };
            append(other.model_type_id[i], other.start_time[i], other.end_time[i],
                   bytes(other.timestamps, i), other.min_value[i], other.max_value[i],
                   bytes(other.values, i), bytes(other.residuals, i), other.chunk_index[i]);
        }
    }

    mdb_segments_owned *seal() {
        auto column = [&](BinaryViewBuilder &b, int slot) {
            buffer_ptrs[slot] = b.data.data();
            buffer_sizes[slot] = (int64_t)b.data.size();
            mdb_binview_col col;
            col.views = b.views.data();
            col.buffers = &buffer_ptrs[slot];
            col.buffer_sizes = &buffer_sizes[slot];
            col.n_buffers = 1;
            return col;
        };
        c.seg.n = model_type_id.size();
        c.seg.model_type_id = model_type_id.data();
        c.seg.start_time = start_time.data();
        c.seg.end_time = end_time.data();
        c.seg.timestamps = column(timestamps, 0);
        c.seg.min_value = min_value.data();
        c.seg.max_value = max_value.data();
        c.seg.values = column(values, 1);
        c.seg.residuals = column(residuals, 2);
        c.error = error.data();
        c.chunk_index = chunk_index.data();
        c.on_device = 0;
        c.priv_ = this;
        return &c;
    }
};

/* The bytes of row `i` of a borrowed BinaryView column. */
inline bool view_bytes(const mdb_binview_col &col, uint64_t i, const uint8_t **p, uint64_t *n) {
    const mdb_view16 &v = col.views[i];
    if (v.length < 0) return false;
    *n = (uint64_t)v.length;
    if (v.length <= 12) {
        *p = v.u.inlined;
        return true;
    }
    if (v.u.ref.buffer_index < 0 || v.u.ref.buffer_index >= col.n_buffers) return false;
    *p = col.buffers[v.u.ref.buffer_index] + v.u.ref.offset;
    return true;
}

/* ---------------------------------------------------------------------------------------------
 * The compression driver (compression.rs:191-400, types.rs:40-278).
 * ------------------------------------------------------------------------------------------- */

struct SelectedModel {
    int8_t model_type_id;
    uint64_t start_index;
    uint64_t end_index;
    float min_value;
    float max_value;
    std::vector<uint8_t> values;
    float model_last_value;
    float bytes_per_value;
};

/* compression.rs:280-301 + types.rs:61-145 */
SelectedModel fit_next_model(uint64_t start_index, mdb_error_bound eb, const int64_t *ts,
                             const float *v, uint64_t n) {
    PmcMean pmc(eb);
    Swing swing(eb);
    bool pmc_fits = true;
    bool swing_fits = true;
    bool can_fit_more = true;
    for (uint64_t i = start_index; can_fit_more && i < n; i++) {
        /* types.rs:74-81: a model that failed once is never fed again. */
        pmc_fits = pmc_fits && pmc.fit_value(v[i]);
        swing_fits = swing_fits && swing.fit_data_point(ts[i], v[i]);
        can_fit_more = pmc_fits || swing_fits;
    }
    /* types.rs:84-101: min_by keeps the first minimum, so PMC-Mean wins ties. */
    SelectedModel m;
    m.start_index = start_index;
    if (pmc.bytes_per_value() <= swing.bytes_per_value()) {
        float value = pmc.model(); /* types.rs:104-119 */
        m.model_type_id = MDB_PMC_MEAN_ID;
        m.end_index = start_index + pmc.length - 1;
        m.min_value = value;
        m.max_value = value;
        m.model_last_value = value;
        m.bytes_per_value = pmc.bytes_per_value();
    } else {
        float first, last; /* types.rs:122-144 */
        swing.model(&first, &last);
        m.model_type_id = MDB_SWING_ID;
        m.end_index = start_index + swing.length - 1;
        m.min_value = min_num(first, last);
        m.max_value = max_num(first, last);
        if (!(first < last)) m.values.push_back(0);
        m.model_last_value = last;
        m.bytes_per_value = swing.bytes_per_value();
    }
    return m;
}

/* types.rs:197-267 */
void finish_model(SelectedModel m, mdb_error_bound eb, uint64_t residuals_end_index,
                  const int64_t *ts, const float *v, OwnedBatch &out, uint32_t chunk) {
    int64_t start_time = ts[m.start_index];
    int64_t end_time = ts[residuals_end_index];
    std::vector<uint8_t> timestamps = compress_residual_timestamps(
        ts + m.start_index, residuals_end_index - m.start_index + 1);
    std::vector<uint8_t> residuals;
    if (m.end_index < residuals_end_index) {
        uint64_t first_residual = m.end_index + 1;
        uint64_t n_residuals = residuals_end_index - first_residual + 1;
        MacaqueV macaque(eb); /* types.rs:270-278 */
        macaque.compress_values_without_first(v + first_residual, n_residuals, m.model_last_value);
        float rmin = macaque.min_value;
        float rmax = macaque.max_value;
        residuals = macaque.out.finish();
        if (m.model_type_id == MDB_PMC_MEAN_ID)
            m.values = encode_values_for_pmc_mean(m.min_value, m.max_value, rmin, rmax);
        else
            m.values = encode_values_for_swing(m.min_value, m.max_value, m.values.empty(), rmin, rmax);
        m.min_value = min_num(m.min_value, rmin);
        m.max_value = max_num(m.max_value, rmax);
        residuals.push_back((uint8_t)n_residuals);
    }
    out.append(m.model_type_id, start_time, end_time, timestamps, m.min_value, m.max_value,
               m.values, residuals, chunk);
}

/* compression.rs:367-400 */
void store_macaque_v_segment(mdb_error_bound eb, uint64_t start_index, uint64_t end_index,
                             const int64_t *ts, const float *v, OwnedBatch &out, uint32_t chunk) {
    uint64_t n = end_index - start_index + 1;
    std::vector<uint8_t> timestamps = compress_residual_timestamps(ts + start_index, n);
    MacaqueV macaque(eb);
    macaque.compress_values(v + start_index, n);
    float mn = macaque.min_value;
    float mx = macaque.max_value;
    std::vector<uint8_t> values = macaque.out.finish();
    out.append(MDB_MACAQUE_V_ID, ts[start_index], ts[end_index], timestamps, mn, mx, values, {},
               chunk);
}

/* compression.rs:310-362 */
void store_model_and_or_residuals(mdb_error_bound eb, const SelectedModel *model,
                                  uint64_t residuals_end_index, const int64_t *ts, const float *v,
                                  OwnedBatch &out, uint32_t chunk) {
    if (model == nullptr) {
        store_macaque_v_segment(eb, 0, residuals_end_index, ts, v, out, chunk);
    } else if (residuals_end_index - model->end_index <= MDB_RESIDUAL_VALUES_MAX_LENGTH) {
        finish_model(*model, eb, residuals_end_index, ts, v, out, chunk);
    } else {
        finish_model(*model, eb, model->end_index, ts, v, out, chunk);
        store_macaque_v_segment(eb, model->end_index + 1, residuals_end_index, ts, v, out, chunk);
    }
}

/* compression.rs:191-275 for one chunk. */
void compress_univariate(const int64_t *ts, const float *v, uint64_t n, mdb_error_bound eb,
                         OwnedBatch &out, uint32_t chunk) {
    if (n == 0) return;
    uint64_t current = 0;
    bool have_previous = false;
    SelectedModel previous;
    while (current < n) {
        SelectedModel model = fit_next_model(current, eb, ts, v, n);
        if (model.bytes_per_value <= (float)MDB_VALUE_SIZE_IN_BYTES) {
            if (current > 0)
                store_model_and_or_residuals(eb, have_previous ? &previous : nullptr, current - 1,
                                             ts, v, out, chunk);
            current = model.end_index + 1;
            previous = std::move(model);
            have_previous = true;
        } else {
            current += 1;
        }
    }
    store_model_and_or_residuals(eb, have_previous ? &previous : nullptr, n - 1, ts, v, out, chunk);
}

bool valid_error_bound(mdb_error_bound eb) {
    /* crates/modelardb_types/src/types.rs:312-334 */
    if (eb.kind == MDB_EB_LOSSLESS) return true;
    if (eb.kind == MDB_EB_ABSOLUTE) return std::isfinite(eb.value) && eb.value > 0.0f;
    if (eb.kind == MDB_EB_RELATIVE) return 0.0f < eb.value && eb.value <= 100.0f;
    return false;
}

/* One row of GridStream's loop (crates/modelardb_storage/src/query/grid_exec.rs:323-356). */
bool grid_row(const mdb_segments *in, uint64_t row, std::vector<int64_t> &ts_out,
              std::vector<float> &val_out) {
    const uint8_t *ts, *values, *residuals;
    uint64_t ts_len, values_len, residuals_len;
    if (!view_bytes(in->timestamps, row, &ts, &ts_len) ||
        !view_bytes(in->values, row, &values, &values_len) ||
        !view_bytes(in->residuals, row, &residuals, &residuals_len))
        return false;
    return segment_grid(in->model_type_id[row], in->start_time[row], in->end_time[row], ts, ts_len,
                        in->min_value[row], in->max_value[row], values, values_len, residuals,
                        residuals_len, ts_out, val_out);
}

} // namespace

/* =============================================================================================
 * C surface.
 * =========================================================================================== */

extern "C" {

const char *ora_last_error(void) { return g_last_error.c_str(); }

int ora_is_value_within_error_bound(mdb_error_bound eb, float real_value, float approximate_value) {
    return within_error_bound(eb, real_value, approximate_value) ? 1 : 0;
}

double ora_maximum_allowed_deviation(mdb_error_bound eb, double value) {
    return max_allowed_deviation(eb, value);
}

int ora_bits_write(const uint64_t *bits, const uint8_t *nbits, uint64_t n, int finish_with_ones,
                   uint8_t *out, uint64_t cap, uint64_t *out_len) {
    BitWriter w;
    for (uint64_t i = 0; i < n; i++) {
        if (nbits[i] > 64) return fail("The number of bits to write must be at most 64.");
        w.put(bits[i], nbits[i]);
    }
    std::vector<uint8_t> bytes = finish_with_ones ? w.finish_with_one_bits() : w.finish();
    *out_len = bytes.size();
    if (bytes.size() > cap) return fail("Output buffer too small.");
    if (!bytes.empty()) std::memcpy(out, bytes.data(), bytes.size());
    return 0;
}

int ora_bits_read(const uint8_t *bytes, uint64_t nbytes, const uint8_t *nbits, uint64_t n,
                  uint64_t *out_values, uint64_t *remaining_bits) {
    if (nbytes == 0) return fail("The bytes slice must not be empty."); /* bits.rs:34-42 */
    BitReader r(bytes, nbytes);
    for (uint64_t i = 0; i < n; i++) out_values[i] = r.get(nbits[i]);
    if (remaining_bits) *remaining_bits = r.remaining();
    return r.overrun ? fail("Read past the end of the bytes.") : 0;
}

int ora_compress_residual_timestamps(const int64_t *ts, uint64_t n, uint8_t *out, uint64_t cap,
                                     uint64_t *out_len) {
    std::vector<uint8_t> bytes = compress_residual_timestamps(ts, n);
    *out_len = bytes.size();
    if (bytes.size() > cap) return fail("Output buffer too small.");
    if (!bytes.empty()) std::memcpy(out, bytes.data(), bytes.size());
    return 0;
}

int ora_decompress_all_timestamps(int64_t start_time, int64_t end_time, const uint8_t *bytes,
                                  uint64_t nbytes, int64_t *out, uint64_t cap, uint64_t *n_out) {
    uint64_t n = 0;
    bool ok = decompress_all_timestamps(start_time, end_time, bytes, nbytes, [&](int64_t t) {
        if (n < cap) out[n] = t;
        n++;
    });
    *n_out = n;
    if (!ok) return fail("Malformed compressed timestamps.");
    return n > cap ? fail("Output buffer too small.") : 0;
}

int ora_are_compressed_timestamps_regular(const uint8_t *bytes, uint64_t nbytes) {
    return compressed_timestamps_regular(bytes, nbytes) ? 1 : 0;
}

int ora_len(int64_t start_time, int64_t end_time, const uint8_t *ts, uint64_t ts_len,
            uint64_t *out_len) {
    return segment_len(start_time, end_time, ts, ts_len, out_len)
               ? 0
               : fail("Malformed compressed timestamps.");
}

int ora_sum(int8_t model_type_id, int64_t start_time, int64_t end_time, const uint8_t *ts,
            uint64_t ts_len, float min_value, float max_value, const uint8_t *values,
            uint64_t values_len, const uint8_t *residuals, uint64_t residuals_len, float *out_sum) {
    return segment_sum(model_type_id, start_time, end_time, ts, ts_len, min_value, max_value,
                       values, values_len, residuals, residuals_len, out_sum)
               ? 0
               : fail("Malformed segment or unknown model type.");
}

int ora_grid(int8_t model_type_id, int64_t start_time, int64_t end_time, const uint8_t *ts,
             uint64_t ts_len, float min_value, float max_value, const uint8_t *values,
             uint64_t values_len, const uint8_t *residuals, uint64_t residuals_len, int64_t *out_ts,
             float *out_val, uint64_t cap, uint64_t *n_out) {
    std::vector<int64_t> t;
    std::vector<float> v;
    if (!segment_grid(model_type_id, start_time, end_time, ts, ts_len, min_value, max_value, values,
                      values_len, residuals, residuals_len, t, v))
        return fail("Malformed segment or unknown model type.");
    *n_out = t.size();
    if (t.size() > cap) return fail("Output buffer too small.");
    if (!t.empty()) {
        std::memcpy(out_ts, t.data(), t.size() * sizeof(int64_t));
        std::memcpy(out_val, v.data(), v.size() * sizeof(float));
    }
    return 0;
}

int ora_pmc_mean_fit(mdb_error_bound eb, const float *v, uint64_t n, uint64_t *n_fit,
                     float *model_value, float *bytes_per_value) {
    PmcMean pmc(eb);
    for (uint64_t i = 0; i < n; i++)
        if (!pmc.fit_value(v[i])) break;
    *n_fit = pmc.length;
    *model_value = pmc.model();
    *bytes_per_value = pmc.bytes_per_value();
    return 0;
}

int ora_swing_fit(mdb_error_bound eb, const int64_t *ts, const float *v, uint64_t n,
                  uint64_t *n_fit, float *first_value, float *last_value, float *bytes_per_value,
                  double *bounds4) {
    Swing swing(eb);
    for (uint64_t i = 0; i < n; i++)
        if (!swing.fit_data_point(ts[i], v[i])) break;
    *n_fit = swing.length;
    swing.model(first_value, last_value);
    *bytes_per_value = swing.bytes_per_value();
    if (bounds4) {
        bounds4[0] = swing.upper.slope;
        bounds4[1] = swing.upper.intercept;
        bounds4[2] = swing.lower.slope;
        bounds4[3] = swing.lower.intercept;
    }
    return 0;
}

float ora_swing_sum(int64_t start_time, int64_t end_time, const uint8_t *ts, uint64_t ts_len,
                    float first_value, float last_value, uint64_t residuals_length) {
    float out = std::numeric_limits<float>::quiet_NaN();
    swing_sum(start_time, end_time, ts, ts_len, first_value, last_value, residuals_length, &out);
    return out;
}

int ora_macaque_v_compress(mdb_error_bound eb, const float *v, uint64_t n, int seeded, float seed,
                           uint8_t *out, uint64_t cap, uint64_t *out_len, float *min_value,
                           float *max_value, uint8_t *last_leading_zero_bits,
                           uint8_t *last_trailing_zero_bits, float *last_value) {
    MacaqueV m(eb);
    if (seeded)
        m.compress_values_without_first(v, n, seed);
    else
        m.compress_values(v, n);
    if (min_value) *min_value = m.min_value;
    if (max_value) *max_value = m.max_value;
    if (last_leading_zero_bits) *last_leading_zero_bits = m.last_leading_zero_bits;
    if (last_trailing_zero_bits) *last_trailing_zero_bits = m.last_trailing_zero_bits;
    if (last_value) *last_value = m.last_value;
    std::vector<uint8_t> bytes = m.out.finish();
    *out_len = bytes.size();
    if (bytes.size() > cap) return fail("Output buffer too small.");
    if (!bytes.empty()) std::memcpy(out, bytes.data(), bytes.size());
    return 0;
}

int ora_macaque_v_grid(const uint8_t *bytes, uint64_t nbytes, uint64_t n, int seeded, float seed,
                       float *out) {
    uint64_t i = 0;
    bool ok = macaque_v_decode(bytes, nbytes, n, seeded != 0, seed, [&](float v) { out[i++] = v; });
    return ok ? 0 : fail("Malformed MacaqueV values.");
}

int ora_macaque_v_sum(const uint8_t *bytes, uint64_t nbytes, uint64_t n, int seeded, float seed,
                      float *out) {
    float sum = 0.0f;
    bool first = !seeded;
    bool ok = macaque_v_decode(bytes, nbytes, n, seeded != 0, seed, [&](float v) {
        if (first) { sum = v; first = false; } else { sum += v; }
    });
    *out = sum;
    return ok ? 0 : fail("Malformed MacaqueV values.");
}

int ora_encode_values_for_pmc_mean(float min_value, float max_value, float residuals_min_value,
                                   float residuals_max_value, uint8_t *out8, uint64_t *out_len) {
    std::vector<uint8_t> b =
        encode_values_for_pmc_mean(min_value, max_value, residuals_min_value, residuals_max_value);
    *out_len = b.size();
    if (!b.empty()) std::memcpy(out8, b.data(), b.size());
    return 0;
}

int ora_decode_values_for_pmc_mean(float min_value, float max_value, const uint8_t *values,
                                   uint64_t values_len, float *out) {
    return decode_values_for_pmc_mean(min_value, max_value, values, values_len, out)
               ? 0
               : fail("Values should be encoded by encode_values_for_pmc_mean().");
}

int ora_encode_values_for_swing(float min_value, float max_value, int min_value_is_first,
                                float residuals_min_value, float residuals_max_value,
                                uint8_t *out8, uint64_t *out_len) {
    std::vector<uint8_t> b = encode_values_for_swing(min_value, max_value, min_value_is_first != 0,
                                                     residuals_min_value, residuals_max_value);
    *out_len = b.size();
    if (!b.empty()) std::memcpy(out8, b.data(), b.size());
    return 0;
}

int ora_decode_values_for_swing(float min_value, float max_value, const uint8_t *values,
                                uint64_t values_len, float *first_value, float *last_value) {
    return decode_values_for_swing(min_value, max_value, values, values_len, first_value, last_value)
               ? 0
               : fail("Unknown encoding of swing.");
}

int ora_fit_next_model(uint64_t start_index, mdb_error_bound eb, const int64_t *ts, const float *v,
                       uint64_t n, ora_model *out) {
    if (start_index >= n) return fail("start_index must be the index of a data point.");
    SelectedModel m = fit_next_model(start_index, eb, ts, v, n);
    out->model_type_id = m.model_type_id;
    out->start_index = m.start_index;
    out->end_index = m.end_index;
    out->min_value = m.min_value;
    out->max_value = m.max_value;
    out->values_len = (uint32_t)m.values.size();
    std::memset(out->values, 0, sizeof(out->values));
    if (!m.values.empty()) std::memcpy(out->values, m.values.data(), m.values.size());
    out->model_last_value = m.model_last_value;
    out->bytes_per_value = m.bytes_per_value;
    return 0;
}

int ora_model_finish(const ora_model *model, mdb_error_bound eb, uint64_t residuals_end_index,
                     const int64_t *ts, const float *v, uint64_t n, mdb_segments_owned **out) {
    if (residuals_end_index >= n || model->end_index > residuals_end_index)
        return fail("residuals_end_index must follow the model and be a valid index.");
    SelectedModel m;
    m.model_type_id = model->model_type_id;
    m.start_index = model->start_index;
    m.end_index = model->end_index;
    m.min_value = model->min_value;
    m.max_value = model->max_value;
    m.values.assign(model->values, model->values + model->values_len);
    m.model_last_value = model->model_last_value;
    m.bytes_per_value = model->bytes_per_value;
    OwnedBatch *batch = new OwnedBatch();
    finish_model(m, eb, residuals_end_index, ts, v, *batch, 0);
    *out = batch->seal();
    return 0;
}

/* Baseline hygiene for the timed legs (bench.py's cpu_baseline): the threaded entry points run on a STANDING
 * pool of workers - made (and, with pinning on, pinned: worker w to the w-th CPU the process is allowed on, so
 * that a page a worker touches first is and stays local to it under Linux's first-touch policy) by the first
 * call that needs them, i.e. by the untimed warm-up pass, and kept for the later ones. Starting and joining 256
 * threads twice per call cost ten times what the workers then computed. */
static int g_pin_threads = 0;
void ora_set_thread_pinning(int enabled) { g_pin_threads = enabled != 0; }

extern "C++" {
namespace {
class StandingPool {
  public:
    static StandingPool &instance() {
        static StandingPool *pool = new StandingPool(); /* never destroyed: the workers outlive main() */
        return *pool;
    }
    /* job(w) for w in [0, workers) on workers of the pool, the caller waits. One run at a time. */
    void run(uint64_t workers, const std::function<void(uint64_t)> &call) {
        if (workers == 0) return;
        std::lock_guard<std::mutex> one_run(run_mutex_);
        grow(workers);
        {
            std::lock_guard<std::mutex> lock(mutex_);
            job_ = &call;
            active_ = workers;
            unfinished_ = workers;
            generation_ += 1;
        }
        wake_.notify_all();
        std::unique_lock<std::mutex> lock(mutex_);
        done_.wait(lock, [&] { return unfinished_ == 0; });
        job_ = nullptr;
    }

  private:
    void grow(uint64_t workers) {
        if (threads_.empty()) { /* the CPUs of the process, read before any worker narrows its own mask */
            cpu_set_t allowed;
            CPU_ZERO(&allowed);
            if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0)
                for (int cpu = 0; cpu < CPU_SETSIZE; cpu++)
                    if (CPU_ISSET(cpu, &allowed)) cpus_.push_back(cpu);
        }
        while (threads_.size() < workers) {
            const uint64_t w = threads_.size();
            const bool pin = g_pin_threads && !cpus_.empty();
            const int cpu = pin ? cpus_[w % cpus_.size()] : -1;
            uint64_t seen;
            {
                std::lock_guard<std::mutex> lock(mutex_);
                seen = generation_;
            }
            threads_.emplace_back([this, w, cpu, seen]() mutable {
                if (cpu >= 0) {
                    cpu_set_t one;
                    CPU_ZERO(&one);
                    CPU_SET(cpu, &one);
                    (void)pthread_setaffinity_np(pthread_self(), sizeof(one), &one);
                }
                for (;;) {
                    const std::function<void(uint64_t)> *job = nullptr;
                    {
                        std::unique_lock<std::mutex> lock(mutex_);
                        wake_.wait(lock, [&] { return generation_ != seen; });
                        seen = generation_;
                        if (w < active_) job = job_;
                    }
                    if (!job) continue;
                    (*job)(w);
                    std::lock_guard<std::mutex> lock(mutex_);
                    if (--unfinished_ == 0) done_.notify_all();
                }
            });
            threads_.back().detach();
        }
    }
    std::mutex run_mutex_, mutex_;
    std::condition_variable wake_, done_;
    std::vector<std::thread> threads_;
    std::vector<int> cpus_;
    const std::function<void(uint64_t)> *job_ = nullptr;
    uint64_t active_ = 0, unfinished_ = 0, generation_ = 0;
};
} /* namespace */
} /* extern "C++" */

int ora_compress_chunks(const int64_t *ts, const float *v, const uint64_t *chunk_offsets,
                        uint64_t n_chunks, mdb_error_bound eb, int n_threads,
                        mdb_segments_owned **out) {
    if (!valid_error_bound(eb)) return fail("Invalid error bound.");
    OwnedBatch *batch = new OwnedBatch();
    if (n_threads <= 1 || n_chunks < 2) {
        for (uint64_t c = 0; c < n_chunks; c++)
            compress_univariate(ts + chunk_offsets[c], v + chunk_offsets[c],
                                chunk_offsets[c + 1] - chunk_offsets[c], eb, *batch, (uint32_t)c);
    } else {
        uint64_t workers = std::min<uint64_t>((uint64_t)n_threads, n_chunks);
        std::vector<OwnedBatch> parts(workers);
        StandingPool::instance().run(workers, [&](uint64_t w) {
            uint64_t begin = n_chunks * w / workers;
            uint64_t end = n_chunks * (w + 1) / workers;
            for (uint64_t c = begin; c < end; c++)
                compress_univariate(ts + chunk_offsets[c], v + chunk_offsets[c],
                                    chunk_offsets[c + 1] - chunk_offsets[c], eb, parts[w], (uint32_t)c);
        });
        for (auto &p : parts) batch->append_all(p);
    }
    *out = batch->seal();
    return 0;
}

void ora_segments_free(mdb_segments_owned *s) {
    if (s) delete static_cast<OwnedBatch *>(s->priv_);
}

int ora_grid_count(const mdb_segments *in, uint64_t *n_out) {
    uint64_t total = 0;
    for (uint64_t row = 0; row < in->n; row++) {
        const uint8_t *ts;
        uint64_t ts_len, count;
        if (!view_bytes(in->timestamps, row, &ts, &ts_len) ||
            !decompressed_timestamp_count(in->start_time[row], in->end_time[row], ts, ts_len, &count))
            return fail("Malformed compressed timestamps.");
        total += count;
    }
    *n_out = total;
    return 0;
}

int ora_grid_batch(const mdb_segments *in, int64_t *out_ts, float *out_val,
                   uint32_t *out_rows_per_segment, uint64_t cap, uint64_t *n_out,
                   mdb_grid_metrics *metrics) {
    std::vector<int64_t> t;
    std::vector<float> v;
    mdb_grid_metrics m;
    std::memset(&m, 0, sizeof(m));
    for (uint64_t row = 0; row < in->n; row++) {
        size_t before = v.size();
        if (!grid_row(in, row, t, v)) return fail("Malformed segment or unknown model type.");
        uint64_t created = v.size() - before;
        if (out_rows_per_segment) out_rows_per_segment[row] = (uint32_t)created;
        /* grid_exec.rs:511-518 */
        int8_t id = in->model_type_id[row];
        const mdb_view16 &ts_view = in->timestamps.views[row];
        bool regular = ts_view.length == 0 ||
                       ((ts_view.length <= 12 ? ts_view.u.inlined[0] : ts_view.u.ref.prefix[0]) & 128) == 0;
        m.rows_created += created;
        m.rows_created_by_model_type[id] += created;
        m.segments_with_residuals += in->residuals.views[row].length != 0;
        m.segments_with_model_type[id] += 1;
        m.segments_regular += regular;
        m.segments_irregular += !regular;
    }
    *n_out = t.size();
    if (metrics) *metrics = m;
    if (t.size() > cap) return fail("Output buffer too small.");
    if (!t.empty()) {
        std::memcpy(out_ts, t.data(), t.size() * sizeof(int64_t));
        std::memcpy(out_val, v.data(), v.size() * sizeof(float));
    }
    return 0;
}

int ora_grid_batch_mt(const mdb_segments *in, int64_t *out_ts, float *out_val, uint64_t cap,
                      uint64_t *n_out, int n_threads) {
    /* Pass 1: how many points each worker's contiguous range of rows produces, so that pass 2 can
     * write straight into the output at the right offset. Pass 2 is GridStream's per-row loop
     * (grid_exec.rs:323-356): grid one row into a builder, append it to the output. */
    const uint64_t workers = std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)n_threads, in->n));
    struct alignas(128) PerWorker { /* (a cache line of its own: 256 workers' counters side by side would share 32) */
        uint64_t count = 0;
        uint64_t offset = 0;
        int failed = 0;
    };
    std::vector<PerWorker> per_worker(workers);
    auto range = [&](uint64_t w, uint64_t *begin, uint64_t *end) {
        *begin = in->n * w / workers;
        *end = in->n * (w + 1) / workers;
    };
    StandingPool::instance().run(workers, [&](uint64_t w) {
        uint64_t begin, end, sum = 0;
        range(w, &begin, &end);
        for (uint64_t row = begin; row < end; row++) {
            const uint8_t *ts;
            uint64_t ts_len, count;
            if (!view_bytes(in->timestamps, row, &ts, &ts_len) ||
                !decompressed_timestamp_count(in->start_time[row], in->end_time[row], ts, ts_len, &count)) {
                per_worker[w].failed = 1;
                return;
            }
            sum += count;
        }
        per_worker[w].count = sum;
    });
    uint64_t total = 0;
    for (uint64_t w = 0; w < workers; w++) {
        if (per_worker[w].failed) return fail("Malformed compressed timestamps.");
        per_worker[w].offset = total;
        total += per_worker[w].count;
    }
    *n_out = total;
    if (total > cap) return fail("Output buffer too small.");
    StandingPool::instance().run(workers, [&](uint64_t w) {
        uint64_t begin, end;
        range(w, &begin, &end);
        std::vector<int64_t> ts_builder;
        std::vector<float> value_builder;
        uint64_t at = per_worker[w].offset;
        for (uint64_t row = begin; row < end; row++) {
            ts_builder.clear();
            value_builder.clear();
            if (!grid_row(in, row, ts_builder, value_builder)) {
                per_worker[w].failed = 1;
                return;
            }
            std::memcpy(out_ts + at, ts_builder.data(), ts_builder.size() * sizeof(int64_t));
            std::memcpy(out_val + at, value_builder.data(), value_builder.size() * sizeof(float));
            at += ts_builder.size();
        }
    });
    for (uint64_t w = 0; w < workers; w++)
        if (per_worker[w].failed) return fail("Malformed segment or unknown model type.");
    return 0;
}

int ora_agg_batch(const mdb_segments *in, uint32_t which_mask, mdb_agg_state *inout) {
    for (uint64_t row = 0; row < in->n; row++) {
        const uint8_t *ts, *values, *residuals;
        uint64_t ts_len, values_len, residuals_len;
        if (!view_bytes(in->timestamps, row, &ts, &ts_len) ||
            !view_bytes(in->values, row, &values, &values_len) ||
            !view_bytes(in->residuals, row, &residuals, &residuals_len))
            return fail("Malformed BinaryView.");
        if (which_mask & (MDB_AGG_COUNT | MDB_AGG_AVG)) { /* :345-358, :584 */
            uint64_t length;
            if (!segment_len(in->start_time[row], in->end_time[row], ts, ts_len, &length))
                return fail("Malformed compressed timestamps.");
            inout->count += (int64_t)length;
        }
        if (which_mask & MDB_AGG_MIN) inout->min = min_num(inout->min, in->min_value[row]); /* :395-401 */
        if (which_mask & MDB_AGG_MAX) inout->max = max_num(inout->max, in->max_value[row]); /* :438-444 */
        if (which_mask & (MDB_AGG_SUM | MDB_AGG_AVG)) { /* :481-513, :553-587 */
            float sum;
            if (!segment_sum(in->model_type_id[row], in->start_time[row], in->end_time[row], ts,
                             ts_len, in->min_value[row], in->max_value[row], values, values_len,
                             residuals, residuals_len, &sum))
                return fail("Malformed segment or unknown model type.");
            inout->sum += (double)sum;
        }
    }
    return 0;
}

int ora_agg_batch_range(const mdb_segments *in, int64_t t_lo, int64_t t_hi, uint32_t which_mask,
                        mdb_agg_state *inout) {
    std::vector<int64_t> t;
    std::vector<float> v;
    for (uint64_t row = 0; row < in->n; row++) {
        t.clear();
        v.clear();
        if (!grid_row(in, row, t, v)) return fail("Malformed segment or unknown model type.");
        for (size_t i = 0; i < t.size(); i++) {
            if (t[i] < t_lo || t[i] > t_hi) continue;
            if (which_mask & (MDB_AGG_COUNT | MDB_AGG_AVG)) inout->count += 1;
            if (which_mask & MDB_AGG_MIN) inout->min = min_num(inout->min, v[i]);
            if (which_mask & MDB_AGG_MAX) inout->max = max_num(inout->max, v[i]);
            if (which_mask & (MDB_AGG_SUM | MDB_AGG_AVG)) inout->sum += (double)v[i];
        }
    }
    return 0;
}

} /* extern "C" */
