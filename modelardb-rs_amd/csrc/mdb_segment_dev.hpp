// mdb_segment_dev.hpp - device-side understanding of one compressed segment, shared by the grid
// and aggregate kernels: descriptor, MacaqueTS / MacaqueV decoders and the values-column codecs.
// Citations are relative to crates/modelardb_compression/src/ in the reference.
#pragma once

#include "mdb_common.hpp"

namespace mdb {

constexpr uint32_t COUNT_MASK = 0x7fffffffu;
constexpr uint32_t SERIAL_BIT = 0x80000000u;

enum : uint32_t {
    FLAG_TYPE_MASK = 3u,
    FLAG_REGULAR = 1u << 2,
    FLAG_HAS_RESIDUALS = 1u << 3,
    FLAG_SERIAL = 1u << 4,
    FLAG_CHECKPOINTS = 1u << 5, // irregular timestamps with checkpoints: k_grid_timestamps writes them
    FLAG_JUMPS = 1u << 6,       // (TileDesc only) irregular timestamps that are a fixed rate with a few jumps:
                                // k_grid_tiles writes them from the segment's jump list (TsJump)
};
constexpr uint32_t FLAG_JUMP_COUNT_SHIFT = 8; // (TileDesc only) how many jumps: the bits above

enum : uint32_t {
    ERR_TIMESTAMPS = 1u << 0,
    ERR_VALUES = 1u << 1,
    ERR_MODEL_TYPE = 1u << 2,
    ERR_RESIDUALS = 1u << 3,
    ERR_BITSTREAM = 1u << 4,
    ERR_TOO_LONG = 1u << 5,
    ERR_SPLIT_CHAIN = 1u << 6, // fit, split mode: the walk reached a point no lane visited (a bug)
    ERR_HOST_INDEX = 1u << 7,  // grid: the cursors host threads left disagree with the kernels' analysis (a bug)
    ERR_ROTATION_STALL = 1u << 8, // fit, rotating groups: a wave waited for a group longer than any kernel runs (a bug)
};

struct SegDesc { // what one lane knows about a segment (registers only)
    int64_t start;
    int64_t delta;
    double slope;
    double intercept;
    uint32_t n_total; // points of the whole segment
    uint32_t n_model; // of which the model represents the first n_model; the rest are residuals
    float value;      // PMC-Mean: the model value. Swing: the last reconstructed value (residual seed).
    uint32_t flags;
    uint32_t first;     // index of the first point inside the requested time range (0 without one)
    uint32_t n_visible; // points this segment contributes to the output (n_total without a range)
};

// The 48 bytes k_grid_tiles reads per segment: the VISIBLE part of the segment, i.e. `start` is
// already advanced to the first point inside the requested time range.
struct TileDesc {
    int64_t start;
    int64_t delta;
    double slope;
    double intercept;
    uint32_t n_points; // visible points
    uint32_t n_model;  // how many of them the model represents
    float value;
    uint32_t flags;
};

__device__ __forceinline__ TileDesc make_tile_desc(const SegDesc &d) {
    TileDesc t;
    t.start = d.start + (int64_t)((uint64_t)d.first * (uint64_t)d.delta);
    t.delta = d.delta;
    t.slope = d.slope;
    t.intercept = d.intercept;
    t.n_points = d.n_visible;
    t.n_model = d.first < d.n_model ? min(d.n_model - d.first, d.n_visible) : 0u;
    t.value = d.value;
    t.flags = d.flags;
    return t;
}

// Optional predicate lo <= timestamp <= hi pushed down into grid (grid_exec.rs:366-387 evaluates it
// after reconstruction; here out-of-range points are never materialised).
struct TimeRange {
    int64_t lo;
    int64_t hi;
    int32_t enabled;
};

// Index interval [k_lo, k_hi] of the points start + k * delta, k in [0, n), inside [lo, hi].
__device__ __forceinline__ bool regular_index_interval(int64_t start, int64_t delta, uint32_t n, int64_t lo,
                                                       int64_t hi, uint32_t *k_lo, uint32_t *k_hi) {
    if (n == 0) return false;
    if (n <= 2 || delta <= 0) { // one or two points, or a degenerate interval: test them one by one
        bool any = false;
        for (uint32_t k = 0; k < n; k++) {
            int64_t t = start + (int64_t)((uint64_t)k * (uint64_t)delta);
            if (t < lo || t > hi) continue;
            if (!any) { *k_lo = k; any = true; }
            *k_hi = k;
        }
        return any;
    }
    const int64_t last_t = start + (int64_t)((uint64_t)(n - 1) * (uint64_t)delta);
    if (last_t < lo || start > hi) return false;
    if (lo <= start && hi >= last_t) { // the whole segment (most of the segments a range reaches into): no division
        *k_lo = 0;
        *k_hi = n - 1;
        return true;
    }
    uint32_t a = 0, b;
    if (lo > start) {
        uint64_t k = ((uint64_t)lo - (uint64_t)start + (uint64_t)delta - 1) / (uint64_t)delta;
        if (k > n - 1) return false;
        a = (uint32_t)k;
    }
    b = hi >= last_t ? n - 1 : (uint32_t)(((uint64_t)hi - (uint64_t)start) / (uint64_t)delta);
    if (b < a) return false;
    *k_lo = a;
    *k_hi = b;
    return true;
}

// ---- MacaqueTS irregular decode (models/timestamps.rs:228-292) ---------------------------------

// Where a delta-of-delta stream stands in front of one of its codes: everything the decoder carries
// from code to code. The stream of a long segment is cut into pieces of TS_PIECE_BITS bits; the one
// sequential parse every such stream needs anyway (len() is the number of codes) leaves the cursor of
// the first code that starts in each piece behind (a "checkpoint"), and from there on the pieces
// are independent: k_grid_timestamps decodes them one lane per piece, and whoever needs the
// timestamp of one point, or the points inside a time range, decodes one piece instead of the stream.
struct TsCursor {
    uint32_t bit;        // of the code, from the start of the payload (bit 0 is the flag "irregular")
    uint32_t count;      // index of the point the code produces
    int64_t timestamp;   // of the point before
    uint64_t last_delta; // delta of the point before
    const uint8_t *stream; // (checkpoints only) the payload, so that a piece can be fetched without its segment
};
static_assert(sizeof(TsCursor) == 32, "32 bytes per piece of a stream");

constexpr uint32_t TS_PIECE_BITS = 256;
constexpr uint32_t TS_NO_CODE = 0xffffffffu; // TsCursor::count of a piece in which no code starts (the last one)
__device__ __forceinline__ uint32_t ts_pieces(uint32_t nbytes) { return (nbytes * 8u + TS_PIECE_BITS - 1) / TS_PIECE_BITS; }
// Streams that live inside their view (at most 12 bytes, a handful of points) have no checkpoints.
__device__ __forceinline__ bool ts_has_checkpoints(int32_t nbytes) { return nbytes > 12 && nbytes < (1 << 28); }

// Decodes the codes that start at `at` and before bit `stop_bit`: emit(index, timestamp) for every
// point, at most up to index limit - 1; cross(cursor) in front of the first code of every piece after
// the one `at` lies in. Returns the cursor it stopped at; *finished: the stream is exhausted and its
// last point (end_time, which is not stored, timestamps.rs:108-113) has been emitted.
template <typename Emit, typename Cross>
__device__ __forceinline__ TsCursor decode_irregular_span(const uint8_t *bytes, uint32_t nbytes, int64_t end_time,
                                                        uint32_t limit, uint32_t *error, TsCursor at,
                                                        uint32_t stop_bit, bool *finished, Emit emit, Cross cross) {
    *finished = false;
    uint32_t count = at.count;
    uint64_t last_delta = at.last_delta;
    int64_t timestamp = at.timestamp;
    uint32_t piece = at.bit / TS_PIECE_BITS;
    if (count >= limit) return at;
    auto cursor = [&](uint64_t bit) { return TsCursor{(uint32_t)bit, count, timestamp, last_delta, bytes}; };
    // Far from the end of the stream (the longest code is 5 + 64 bits) nothing can go wrong, and the
    // codes are taken off the top of a 64-bit buffer with as few branches as possible: 64 lanes decode
    // 64 different codes per step, and every branch they disagree on is executed by all of them.
    LeanReaderDev w;
    w.open(bytes, nbytes, at.bit);
    while (w.far_from_end(80)) {
        if (w.position >= stop_bit) return cursor(w.position);
        if ((uint32_t)(w.position / TS_PIECE_BITS) != piece) {
            piece = (uint32_t)(w.position / TS_PIECE_BITS);
            cross(cursor(w.position));
        }
        w.refill();
        const uint32_t top = w.top();
        if ((top >> 31) == 0u) {
            // A run of `0` codes - the delta repeats: what a series sampled at a fixed rate with the
            // odd gap or jitter consists of almost entirely (one irregularity makes the whole segment
            // "irregular", timestamps.rs:77-96). The whole run is taken off the stream at once, as far
            // as it stays inside the piece.
            uint32_t run = top == 0u ? 32u : (uint32_t)__clz((int)top);
            run = min(run, min(limit - count, COUNT_MASK - count));
            run = min(run, (piece + 1u) * TS_PIECE_BITS - (uint32_t)w.position);
            run = min(run, stop_bit - (uint32_t)w.position);
            if (run == 0) { // count == COUNT_MASK: as below
                *error |= ERR_TOO_LONG;
                return cursor(w.position);
            }
            w.consume(run);
            for (uint32_t j = 0; j < run; j++) {
                timestamp = (int64_t)((uint64_t)timestamp + last_delta);
                emit(count++, timestamp);
            }
            if (count >= limit) return cursor(w.position);
            continue;
        }
        const uint32_t ones = min((uint32_t)__clz((int)~top), 5u);
        if (ones <= 3) {
            // `10` + 7, `110` + 9 or `1110` + 12 bits: two's complement of `width` bits, except that
            // 2^(width-1) itself is positive (timestamps.rs:139-155).
            const uint32_t width = (0x0c090700u >> (8u * ones)) & 0xffu; // (0,) 7, 9, 12
            const uint32_t length = ones + 1u + width;
            const uint32_t encoded = __builtin_amdgcn_ubfe(top, 32u - length, width);
            w.consume(length);
            const uint32_t half = (1u << width) >> 1;
            const int32_t delta_of_delta = encoded > half ? (int32_t)(encoded - (half << 1)) : (int32_t)encoded;
            last_delta += (uint64_t)(int64_t)delta_of_delta;
        } else {
            w.consume(5); // `11110` or `11111`
            w.refill();
            uint64_t encoded = w.top();
            w.consume(32);
            if (ones == 5) {
                w.refill();
                encoded = (encoded << 32) | w.top();
                w.consume(32);
                last_delta += encoded;
            } else {
                last_delta += encoded > (1ull << 31) ? (encoded | (~0ull << 32)) : encoded;
            }
        }
        timestamp = (int64_t)((uint64_t)timestamp + last_delta);
        if (count == COUNT_MASK) {
            *error |= ERR_TOO_LONG;
            return cursor(w.position);
        }
        emit(count++, timestamp);
        if (count >= limit) return cursor(w.position);
    }
    // The last few codes, where running out of bits has a meaning, with the careful reader.
    BitReaderDev r;
    r.seek(bytes, nbytes, w.position);
    while (!r.exhausted()) {
        if (r.used_bits >= stop_bit) return cursor(r.used_bits);
        if ((uint32_t)(r.used_bits / TS_PIECE_BITS) != piece) {
            piece = (uint32_t)(r.used_bits / TS_PIECE_BITS);
            cross(cursor(r.used_bits));
        }
        uint32_t ones = 0;
        while (ones < 5 && !r.exhausted() && r.get(1)) ones++;
        if (ones != 0 && r.remaining() < 7) break;
        if (ones != 0) {
            const uint32_t width = ones == 1 ? 7u : ones == 2 ? 9u : ones == 3 ? 12u : ones == 4 ? 32u : 64u;
            if (r.remaining() < width) {
                *error |= ERR_TIMESTAMPS;
                return cursor(r.used_bits);
            }
            uint64_t encoded = r.get64(width);
            uint64_t dod = encoded;
            if (width < 64 && encoded > (1ull << (width - 1))) dod = encoded | (~0ull << width);
            last_delta += dod;
        }
        timestamp = (int64_t)((uint64_t)timestamp + last_delta);
        if (count == COUNT_MASK) {
            *error |= ERR_TOO_LONG;
            return cursor(r.used_bits);
        }
        emit(count++, timestamp);
        if (count >= limit) return cursor(r.used_bits);
    }
    emit(count++, end_time);
    *finished = true;
    return cursor(nbytes * 8ull);
}

// The cursor in front of the first code of a stream: point 0 is start_time, bit 0 the flag.
__device__ __forceinline__ TsCursor ts_stream_start(int64_t start_time, const uint8_t *stream = nullptr) {
    return TsCursor{1u, 1u, start_time, 0ull, stream};
}

// One code of at most 16 bits off the top of `top` (the next 32 bits of the stream): `0`, `10` + 7,
// `110` + 9 or `1110` + 12 bits, two's complement of `width` bits except that 2^(width-1) itself is
// positive (timestamps.rs:139-155). Returns the length of the code; `ones` = 4 or 5 (a 32- or 64-bit
// payload) is the caller's. No branch: lanes that stand at different kinds of code stay together.
__device__ __forceinline__ uint32_t ts_short_code(uint32_t top, uint32_t ones, int32_t *delta_of_delta) {
    const uint32_t width = (0x0c090700u >> (8u * ones)) & 0xffu; // 0, 7, 9, 12
    const uint32_t length = ones + 1u + width;
    const uint32_t encoded = __builtin_amdgcn_ubfe(top, 32u - length, width);
    const uint32_t half = (1u << width) >> 1;
    *delta_of_delta = encoded > half ? (int32_t)(encoded - (half << 1)) : (int32_t)encoded;
    return length;
}

// Calls emit(index, timestamp) for every timestamp; returns the count. `limit` stops early.
template <typename Emit>
__device__ __forceinline__ uint32_t decode_irregular_timestamps(const uint8_t *bytes, uint32_t nbytes,
                                                              int64_t start_time, int64_t end_time,
                                                              uint32_t limit, uint32_t *error,
                                                              Emit emit) {
    emit(0u, start_time);
    if (limit <= 1) return 1;
    bool finished;
    return decode_irregular_span(bytes, nbytes, end_time, limit, error, ts_stream_start(start_time), 0xffffffffu,
                                 &finished, emit, [](const TsCursor &) {})
        .count;
}

// A series sampled at a fixed rate with the odd sample missing (or late) has "irregular" timestamps
// (timestamps.rs:77-96), but nearly all of its deltas are one and the same: point k lies at
//   start_time + k * base + (sum of (delta_j - base) over the points j <= k whose delta is not base),
// and the few places where that sum changes - the jumps - are all there is to know about the stream. The walk
// that counts the stream's codes (k_grid_ts_count) writes them down: the list of segment i is the
// TS_JUMPS_PER_PIECE entries per piece of its stream from jumps[piece_base[i] * TS_JUMPS_PER_PIECE] on, entry 0 the
// header, the jumps behind it in the order of their points. Such a segment is reconstructed by k_grid_tiles like
// one with regular timestamps, plus a search in its list. A stream with more jumps than its list has room
// for (one per 64 bits of the stream), or more than one per sixteen points, has none (TS_NO_JUMPS) and is
// decoded piece by piece (k_grid_timestamps).
struct TsJump {
    uint32_t position; // index of the point (header: unused)
    uint32_t count;    // header: how many jumps follow, or TS_NO_JUMPS
    int64_t value;     // what the jumps up to and including this one add up to (header: the base delta)
};
static_assert(sizeof(TsJump) == 16, "16 bytes per jump");
constexpr uint32_t TS_JUMPS_PER_PIECE = 4;
constexpr uint32_t TS_NO_JUMPS = 0xffffffffu;
constexpr uint32_t TS_MAX_JUMPS = (1u << (32 - FLAG_JUMP_COUNT_SHIFT)) - 1u; // (the count travels in TileDesc::flags)
constexpr uint32_t TS_PIECE_LISTED = 0x80000000u; // TsCheckpoints::piece_segment[].y: the piece's segment has a jump list

// Where the list of a segment with FLAG_JUMPS begins (its first piece, i.e. entry * TS_JUMPS_PER_PIECE of
// TsCheckpoints::jumps), in the one member of its descriptor k_grid_tiles does not read for its model type:
// a Swing segment's `value` (the seed of its residuals, which k_grid_serial works out again), the low half of
// the `intercept` of any other. Knowing it from the descriptor saves the tile kernel a trip to memory per
// segment and wave.
__device__ __forceinline__ void set_jump_list(TileDesc &t, uint32_t first_piece) {
    if ((t.flags & FLAG_TYPE_MASK) == MDB_SWING_ID) t.value = __uint_as_float(first_piece);
    else t.intercept = __longlong_as_double((long long)first_piece);
}
__device__ __forceinline__ uint32_t jump_list_of(const TileDesc &t) {
    return (t.flags & FLAG_TYPE_MASK) == MDB_SWING_ID ? __float_as_uint(t.value)
                                                       : (uint32_t)__double_as_longlong(t.intercept);
}

// The checkpoints of a batch (or none: piece_base == nullptr). Segment i owns the slots
// [piece_base[i], piece_base[i + 1]): ts_pieces(length of its stream) of them if the stream has
// checkpoints (irregular, out of line), none otherwise.
struct TsCheckpoints {
    const unsigned long long *piece_base;
    TsCursor *slots;
    uint2 *piece_segment; // slot -> {segment, bytes of its stream}
    TsJump *jumps;        // TS_JUMPS_PER_PIECE entries per slot, or nullptr: no jump lists are kept
    uint32_t *live;       // (with jump lists) the slots of segments that have none, where those are few; or nullptr
};

// How many of a stream's pieces have a code starting in them: all, or all but the last.
__device__ __forceinline__ uint32_t ts_valid_pieces(const TsCursor *slots, uint32_t n_pieces) {
    while (n_pieces > 1 && slots[n_pieces - 1].count == TS_NO_CODE) n_pieces -= 1;
    return n_pieces;
}

// Timestamp of point `index` (>= 1) of a stream with checkpoints: one piece is decoded.
__device__ __forceinline__ int64_t ts_point_at(const TsCursor *slots, uint32_t n_pieces, const uint8_t *bytes,
                                               uint32_t nbytes, int64_t end_time, uint32_t index, uint32_t *error) {
    n_pieces = ts_valid_pieces(slots, n_pieces);
    uint32_t lo = 0, hi = n_pieces; // the last piece whose first code produces a point <= index
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) / 2;
        if (slots[mid].count <= index) lo = mid;
        else hi = mid;
    }
    int64_t found = slots[lo].timestamp;
    bool finished;
    decode_irregular_span(bytes, nbytes, end_time, index + 1, error, slots[lo], 0xffffffffu, &finished,
                          [&](uint32_t, int64_t t) { found = t; }, [](const TsCursor &) {});
    return found;
}

// Index of the first point with timestamp >= bound (or > bound if `beyond`), n_total if there is none;
// timestamps ascend (the compressor requires it). Point 0 is start_time. One piece is decoded.
__device__ __forceinline__ uint32_t ts_first_index(const TsCursor *slots, uint32_t n_pieces, const uint8_t *bytes,
                                                   uint32_t nbytes, int64_t start_time, int64_t end_time,
                                                   uint32_t n_total, int64_t bound, bool beyond, uint32_t *error) {
    auto reached = [&](int64_t t) { return beyond ? t > bound : t >= bound; };
    if (reached(start_time)) return 0;
    n_pieces = ts_valid_pieces(slots, n_pieces);
    uint32_t lo = 0, hi = n_pieces; // the last piece that begins behind a point not yet at the bound
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) / 2;
        if (!reached(slots[mid].timestamp)) lo = mid;
        else hi = mid;
    }
    uint32_t found = n_total;
    bool finished;
    decode_irregular_span(bytes, nbytes, end_time, 0xffffffffu, error, slots[lo], (lo + 1) * TS_PIECE_BITS, &finished,
                          [&](uint32_t k, int64_t t) {
                              if (found == n_total && reached(t)) found = k;
                          },
                          [](const TsCursor &) {});
    return found;
}

// ---- prepass -----------------------------------------------------------------------------------

__device__ __forceinline__ float f32_from_inline(const uint4 &view, uint32_t first_byte) {
    uint32_t bits = view_inline_byte(view, first_byte) | (view_inline_byte(view, first_byte + 1) << 8) |
                    (view_inline_byte(view, first_byte + 2) << 16) |
                    (view_inline_byte(view, first_byte + 3) << 24);
    return __uint_as_float(bits);
}

// types.rs:307-321
__device__ __forceinline__ bool decode_pmc_value(const uint4 &view, float mn, float mx, float *value) {
    switch ((int32_t)view.x) {
    case 0: *value = mn; return true;
    case 1: *value = mx; return true;
    case 4: *value = __uint_as_float(view.y); return true;
    default: return false;
    }
}

// types.rs:374-407
__device__ __forceinline__ bool decode_swing_values(const uint4 &view, float mn, float mx, float *first,
                                                    float *last) {
    switch ((int32_t)view.x) {
    case 0: *first = mn; *last = mx; return true;
    case 1: *first = mx; *last = mn; return true;
    case 5: {
        float value = f32_from_inline(view, 1);
        switch (view_inline_byte(view, 0)) {
        case 0: *first = value; *last = mx; return true;
        case 1: *first = mx; *last = value; return true;
        case 2: *first = mn; *last = value; return true;
        case 3: *first = value; *last = mn; return true;
        default: return false;
        }
    }
    case 8: *first = __uint_as_float(view.y); *last = __uint_as_float(view.z); return true;
    default: return false;
    }
}

// Everything grid()/sum()/len() need to know about one segment before touching its points.
struct SegInfo {
    SegDesc desc;
    uint32_t error;
    float swing_first; // Swing only: decoded (first, last) model values
    float swing_last;
    uint32_t regular_length; // len() of a regular stream = the stored big-endian integer
};

// The timestamps of segment i are regular (timestamps.rs:199-202), or the empty stream of one or two points.
__device__ __forceinline__ bool segment_has_regular_timestamps(const DevSegments &s, uint64_t i) {
    const uint4 vt = s.timestamps.views[i];
    const int32_t ts_len = (int32_t)vt.x;
    return ts_len == 0 || (ts_len > 0 && (view_inline_byte(vt, 0) & 0x80u) == 0);
}

// PMC-Mean or Swing, timestamps regular (or the empty stream of one or two points), no residuals: nothing
// about such a segment needs a bit stream walked.
__device__ __forceinline__ bool segment_is_simple(const DevSegments &s, uint64_t i) {
    const int32_t type = s.model_type_id[i];
    const uint4 vt = s.timestamps.views[i];
    const int32_t ts_len = (int32_t)vt.x;
    const int32_t res_len = (int32_t)s.residuals.views[i].x;
    const bool regular = ts_len == 0 || (ts_len > 0 && (view_inline_byte(vt, 0) & 0x80u) == 0);
    return (type == MDB_PMC_MEAN_ID || type == MDB_SWING_ID) && regular && res_len == 0;
}

// `known_totals` (may be nullptr): per segment, the number of points of a segment with irregular
// timestamps as an earlier analyse_segment of the same batch counted it - counting means parsing
// the whole delta-of-delta stream, which the prepass has already done once.
// `checkpoints` (may be nullptr): the batch's timestamp checkpoints, made by k_grid_ts_count (which has
// also left the streams' lengths in known_totals).
//
// KIND = 1 (ANALYSE_SIMPLE) is the same analysis for a segment the caller has found to be "simple"
// (segment_is_simple(): PMC-Mean or Swing, regular timestamps, no residuals): the branches such a
// segment cannot take are not compiled, which is what lets the prepass run them at three times the
// occupancy (the delta-of-delta decoders are most of the generic version's 132 registers).
// `line_wanted` = false: the caller has no use for the Swing line through the MODEL's end (grid()'s line; the
// aggregates' goes through the segment's end, swing.rs:273-274) - which, for a segment with irregular timestamps
// and residuals, costs a decode of its stream up to that point.
// KIND = 2 (ANALYSE_REGULAR): any model type, residuals allowed, but timestamps the caller has found to be regular
// (segment_has_regular_timestamps): the delta-of-delta decoders are not compiled, the rest is the generic analysis.
constexpr int ANALYSE_GENERIC = 0, ANALYSE_SIMPLE = 1, ANALYSE_REGULAR = 2;
template <int KIND = ANALYSE_GENERIC>
__device__ __forceinline__ SegInfo analyse_segment(const DevSegments &s, uint64_t i,
                                                   const uint32_t *known_totals = nullptr,
                                                   const TsCheckpoints *checkpoints = nullptr,
                                                   bool line_wanted = true) {
    constexpr bool SIMPLE = KIND == ANALYSE_SIMPLE;         // no residuals, not a MacaqueV model
    constexpr bool NO_STREAMS = KIND != ANALYSE_GENERIC;    // no delta-of-delta timestamps
    SegInfo info;
    info.error = 0;
    info.swing_first = 0.0f;
    info.swing_last = 0.0f;
    SegDesc &d = info.desc;
    const int32_t type = s.model_type_id[i];
    const int64_t start = s.start_time[i];
    const int64_t end = s.end_time[i];
    const uint4 vt = s.timestamps.views[i];
    const uint4 vv = s.values.views[i];
    const uint4 vr = s.residuals.views[i];
    const float mn = s.min_value[i];
    const float mx = s.max_value[i];
    d.start = start;
    d.delta = 0;
    d.slope = 0.0;
    d.intercept = 0.0;
    d.value = 0.0f;
    uint32_t flags = (uint32_t)type & FLAG_TYPE_MASK;
    if (type < 0 || type >= MDB_MODEL_TYPE_COUNT) info.error |= ERR_MODEL_TYPE;

    // len() / decompress_all_timestamps() count (models/mod.rs:98-124, timestamps.rs:163-223).
    const int32_t ts_len = (int32_t)vt.x;
    uint32_t n_total = 0;
    bool regular = true;
    const uint8_t *ts_bytes = nullptr;
    TsCursor *slots = nullptr; // the stream's checkpoints, if it has any
    uint32_t n_slots = 0;
    info.regular_length = 0;
    if (ts_len < 0) {
        info.error |= ERR_TIMESTAMPS;
    } else if (ts_len == 0) {
        n_total = (start == end) ? 1u : 2u;
        d.delta = end - start;
        info.regular_length = n_total;
    } else {
        // timestamps.rs:199-202. Byte 0 of an out-of-line view is byte 0 of its 4-byte prefix, which
        // sits where the inline bytes start, so one accessor serves both.
        regular = (view_inline_byte(vt, 0) & 0x80u) == 0;
    }
    if (ts_len > 0 && regular) {
        if (ts_len > 8) {
            info.error |= ERR_TIMESTAMPS;
        } else {
            uint64_t length = 0;
            for (int32_t k = 0; k < ts_len; k++) length = (length << 8) | view_inline_byte(vt, k);
            info.regular_length = (uint32_t)(length > COUNT_MASK ? COUNT_MASK : length);
            uint64_t span = (uint64_t)(end - start);
            if (length < 2) {
                info.error |= ERR_TIMESTAMPS; // the interval is a division by length - 1 (timestamps.rs:218-219)
            } else if (end < start) {
                // (start..=end).step_by(..) is an empty range: grid() produces no point for such a
                // segment while len() still reports the stored length.
                n_total = 0;
            } else {
                uint64_t interval = span / (length - 1);
                if (interval == 0) {
                    info.error |= ERR_TIMESTAMPS;
                } else {
                    uint64_t produced = span / interval + 1; // (start..=end).step_by(interval)
                    if (produced > COUNT_MASK) info.error |= ERR_TOO_LONG;
                    n_total = (uint32_t)produced;
                    d.delta = (int64_t)interval;
                }
            }
        }
    } else if (NO_STREAMS) {
        if (ts_len > 0) info.error |= ERR_TIMESTAMPS; // (not a segment with regular timestamps: the caller's mistake)
    } else if (ts_len > 0) {
        regular = false;
        ts_bytes = view_data(s.timestamps, i, vt);
        if (checkpoints && checkpoints->piece_base && ts_has_checkpoints(ts_len)) {
            flags |= FLAG_CHECKPOINTS;
            slots = checkpoints->slots + checkpoints->piece_base[i];
            n_slots = ts_pieces((uint32_t)ts_len);
        }
        if (known_totals) {
            n_total = known_totals[i];
        } else {
            n_total = decode_irregular_timestamps(ts_bytes, (uint32_t)ts_len, start, end, 0xffffffffu,
                                                  &info.error, [](uint32_t, int64_t) {});
        }
    }
    if (regular) flags |= FLAG_REGULAR;

    // residuals_length() (models/mod.rs:277-284)
    const int32_t res_len = (int32_t)vr.x;
    uint32_t n_res = 0;
    if (SIMPLE) {
        if (res_len != 0) info.error |= ERR_RESIDUALS; // (not a simple segment)
    } else if (res_len < 0) {
        info.error |= ERR_RESIDUALS;
    } else if (res_len > 0) {
        flags |= FLAG_HAS_RESIDUALS;
        n_res = (res_len <= 12) ? view_inline_byte(vr, (uint32_t)res_len - 1)
                                : (uint32_t)view_data(s.residuals, i, vr)[res_len - 1];
        if (res_len < 2) info.error |= ERR_RESIDUALS; // BitReader::try_new(&[]) fails (macaque_v.rs:279)
    }
    if (n_res > n_total) {
        info.error |= ERR_RESIDUALS;
        n_res = n_total;
    }
    const uint32_t n_model = n_total - n_res;
    d.n_total = n_total;
    d.n_model = n_model;

    if (type == MDB_PMC_MEAN_ID) {
        if (!decode_pmc_value(vv, mn, mx, &d.value)) info.error |= ERR_VALUES;
    } else if (type == MDB_SWING_ID) {
        float first = 0.0f, last = 0.0f;
        if (!decode_swing_values(vv, mn, mx, &first, &last)) info.error |= ERR_VALUES;
        if (n_model == 0) {
            info.error |= ERR_VALUES; // expect("Model should represent at least one value.")
        } else if (!info.error) {
            int64_t model_end = start;
            if (regular || NO_STREAMS) {
                model_end = start + (int64_t)((uint64_t)(n_model - 1) * (uint64_t)d.delta);
            } else if (n_res == 0 || !line_wanted) {
                model_end = end; // the last timestamp is not stored in the stream: it is end_time
            } else if (slots) {
                model_end = n_model == 1 ? start
                                         : ts_point_at(slots, n_slots, ts_bytes, (uint32_t)ts_len, end, n_model - 1, &info.error);
            } else {
                decode_irregular_timestamps(ts_bytes, (uint32_t)ts_len, start, end, n_model,
                                            &info.error,
                                            [&](uint32_t, int64_t t) { model_end = t; });
            }
            // models/mod.rs:219-234 + swing.rs:304-319: the line goes through the MODEL's end.
            LineDev line = line_through(start, (double)first, model_end, (double)last);
            d.slope = line.slope;
            d.intercept = line.intercept;
            d.value = (float)(line.slope * (double)model_end + line.intercept);
        }
        info.swing_first = first;
        info.swing_last = last;
    } else if (type == MDB_MACAQUE_V_ID) {
        if ((int32_t)vv.x <= 0 || n_model == 0) info.error |= ERR_VALUES;
    }
    // (irregular timestamps with checkpoints are k_grid_timestamps' work, not the serial kernel's)
    if (!SIMPLE && ((!regular && !(flags & FLAG_CHECKPOINTS)) || type == MDB_MACAQUE_V_ID || n_res > 0))
        flags |= FLAG_SERIAL;
    d.flags = flags;
    d.first = 0;
    d.n_visible = n_total;
    return info;
}

// Restrict a segment to the points with lo <= timestamp <= hi: sets first / n_visible and drops
// the serial flag when no serially decoded point remains visible.
// `known_first` / `known_visible` (may be nullptr): what an earlier apply_time_range of the same batch
// found for a segment with irregular timestamps (it has to decode them all to find out).
__device__ __forceinline__ void apply_time_range(const DevSegments &s, uint64_t i, SegInfo &info,
                                                 const TimeRange &range, const uint32_t *known_first = nullptr,
                                                 const uint32_t *known_visible = nullptr,
                                                 const TsCheckpoints *checkpoints = nullptr) {
    SegDesc &d = info.desc;
    if (info.error) return;
    uint32_t k_lo = 0, k_hi = 0;
    bool any;
    if (d.flags & FLAG_REGULAR) {
        any = regular_index_interval(d.start, d.delta, d.n_total, range.lo, range.hi, &k_lo, &k_hi);
    } else if (known_first && known_visible) {
        const uint32_t visible = known_visible[i] & COUNT_MASK;
        any = visible > 0;
        k_lo = known_first[i];
        k_hi = k_lo + visible - 1;
    } else if ((d.flags & FLAG_CHECKPOINTS) && checkpoints && checkpoints->piece_base) {
        // Timestamps are sorted (the compressor requires it): the first point at or behind lo and the
        // first one behind hi, each found in one piece of the stream.
        const uint4 vt = s.timestamps.views[i];
        const uint8_t *bytes = view_data(s.timestamps, i, vt);
        const TsCursor *slots = checkpoints->slots + checkpoints->piece_base[i];
        const uint32_t n_slots = ts_pieces(vt.x);
        const int64_t end = s.end_time[i];
        k_lo = ts_first_index(slots, n_slots, bytes, vt.x, d.start, end, d.n_total, range.lo, false, &info.error);
        const uint32_t behind = ts_first_index(slots, n_slots, bytes, vt.x, d.start, end, d.n_total, range.hi, true,
                                               &info.error);
        any = k_lo < behind;
        k_hi = any ? behind - 1 : 0;
    } else {
        // Timestamps are sorted (the compressor requires it), so the in-range ones are an interval.
        any = false;
        const uint4 vt = s.timestamps.views[i];
        decode_irregular_timestamps(view_data(s.timestamps, i, vt), vt.x, d.start, s.end_time[i],
                                    0xffffffffu, &info.error, [&](uint32_t k, int64_t t) {
                                        if (t >= range.lo && t <= range.hi) {
                                            if (!any) { k_lo = k; any = true; }
                                            k_hi = k;
                                        }
                                    });
    }
    d.first = any ? k_lo : 0;
    d.n_visible = any ? k_hi - k_lo + 1 : 0;
    const uint32_t type = d.flags & FLAG_TYPE_MASK;
    const bool model_visible = any && d.first < d.n_model;
    const bool residuals_visible = any && d.first + d.n_visible > d.n_model;
    const bool serial = any && (!(d.flags & (FLAG_REGULAR | FLAG_CHECKPOINTS)) ||
                                (type == MDB_MACAQUE_V_ID && model_visible) || residuals_visible);
    d.flags = (d.flags & ~FLAG_SERIAL) | (serial ? FLAG_SERIAL : 0u);
}

// MacaqueV decoder (models/macaque_v.rs:272-323). emit(i, bits) for i in [0, count).
template <typename Emit>
__device__ __forceinline__ void decode_macaque_v(const uint8_t *bytes, uint32_t nbytes, uint32_t count,
                                                 bool seeded, uint32_t seed_bits, uint32_t *error,
                                                 Emit emit) {
    if (nbytes == 0) {
        if (count > 0 || !seeded) *error |= ERR_BITSTREAM;
        return;
    }
    uint32_t leading = 255, trailing = 0;
    uint32_t last = seed_bits;
    uint32_t emitted = 0;
    if (!seeded && count == 0) {
        *error |= ERR_BITSTREAM;
        return;
    }
    // Far from the end of the stream the codes are taken off a 128-bit window (see WindowReaderDev):
    // one lane per stream, so every branch the lanes disagree on costs all of them.
    WindowReaderDev w;
    w.open(bytes, nbytes, 0);
    bool first_pending = !seeded;
    while (emitted < count && w.far_from_end(96)) {
        if (first_pending) { // the first value: 32 raw bits
            last = w.top();
            w.consume(32);
            first_pending = false;
            emit(emitted++, last);
            continue;
        }
        const uint32_t top = w.top(); // c0 c1 leading[5] meaningful[6] ...
        if ((top >> 30) == 2u) {      // `10`: the value repeats
            w.consume(2);
        } else {
            uint32_t meaningful;
            if ((top >> 31) == 0u) { // `0`: the previous window
                meaningful = 32u - leading - trailing;
                if (meaningful > 32u || trailing > 31u) {
                    *error |= ERR_BITSTREAM;
                    return;
                }
                w.consume(1);
            } else { // `11` + 5 bits leading zeros + 6 bits length
                leading = (top >> 25) & 31u;
                meaningful = (top >> 19) & 63u;
                trailing = 32u - meaningful - leading;
                if (meaningful > 32u || trailing > 31u) {
                    *error |= ERR_BITSTREAM;
                    return;
                }
                w.consume(13);
            }
            if (meaningful > 0) {
                const uint32_t value = w.top() >> (32u - meaningful);
                w.consume(meaningful);
                last ^= value << trailing;
            }
        }
        emit(emitted++, last);
    }
    // The rest (or all of a short stream) with the careful reader, where running out of bits counts.
    BitReaderDev r;
    r.seek(bytes, nbytes, w.position);
    if (first_pending) {
        last = r.get(32);
        emit(emitted++, last);
    }
    while (emitted < count) {
        // A stream shorter than its segment claims (a corrupted length can claim 2^31 values): every
        // value takes at least one bit, so stopping here bounds the loop by the size of the stream.
        if (r.overrun()) break;
        bool decode_value = true;
        if (r.get(1)) {
            if (r.get(1)) {
                leading = r.get(5);
                uint32_t meaningful = r.get(6);
                trailing = 32u - meaningful - leading; // wraps for malformed streams, checked below
            } else {
                decode_value = false;
            }
        }
        if (decode_value) {
            uint32_t meaningful = 32u - leading - trailing;
            if (meaningful > 32u || trailing > 31u) {
                *error |= ERR_BITSTREAM;
                return;
            }
            uint32_t value = r.get(meaningful);
            value <<= trailing;
            value ^= last;
            last = value;
        }
        emit(emitted++, last);
    }
    if (r.overrun()) *error |= ERR_BITSTREAM;
}

inline std::string describe_error(uint32_t error) {
    std::string message = "Malformed segment:";
    if (error & ERR_TIMESTAMPS) message += " compressed timestamps;";
    if (error & ERR_VALUES) message += " values are not encoded for the model type;";
    if (error & ERR_MODEL_TYPE) message += " unknown model type;";
    if (error & ERR_RESIDUALS) message += " residuals;";
    if (error & ERR_BITSTREAM) message += " MacaqueV bitstream;";
    if (error & ERR_TOO_LONG) message += " more than 2^31-1 data points in one segment;";
    if (error & ERR_HOST_INDEX) message += " (internal) the host threads' cursors and the kernels disagree about a segment's length;";
    return message;
}

// ---- which MacaqueV streams go through the parallel decoder (mdb_macaque_parallel.hpp) ----------------

#ifndef MDB_MV_PIECE_BITS
#define MDB_MV_PIECE_BITS 4096
#endif
constexpr uint32_t MV_PIECE_BITS = MDB_MV_PIECE_BITS; // a stream is cut into pieces of this many bits
constexpr uint32_t MV_MAX_STREAM_BYTES = 1u << 27; // bit positions stay below 2^30
constexpr uint32_t MV_DEFAULT_MIN_VALUES = 1024;
// More pieces than this in one batch: there are enough streams to keep the GPU busy with one lane
// per stream, which does a third of the work per value.
constexpr uint64_t MV_MAX_PIECES = (1ull << 29) / MV_PIECE_BITS; // 2^29 bits of streams

// Should this segment's values go through the parallel decoder? Evaluated identically by whoever
// bounds the scratch memory (k_grid_prepass, k_agg_segments) and by the kernels that select streams.
__device__ __forceinline__ bool mv_qualifies(const SegInfo &info, uint32_t values_bytes, uint32_t min_values) {
    const SegDesc &d = info.desc;
    return min_values != 0xffffffffu && !info.error && (d.flags & FLAG_TYPE_MASK) == MDB_MACAQUE_V_ID &&
           !(d.flags & FLAG_HAS_RESIDUALS) && d.n_model >= min_values && d.n_visible > 0 &&
           values_bytes > 12 && values_bytes < MV_MAX_STREAM_BYTES;
}

// The same for SUM, whose length is len()'s (models/mod.rs:98-124: a regular stream reports its STORED
// length): only streams for which that is also the number of values grid() would produce.
__device__ __forceinline__ bool mv_qualifies_for_sum(const SegInfo &info, uint32_t values_bytes, uint32_t min_values) {
    const uint32_t length = (info.desc.flags & FLAG_REGULAR) ? info.regular_length : info.desc.n_total;
    return mv_qualifies(info, values_bytes, min_values) && length == info.desc.n_model;
}

// Aggregates leave long MacaqueV streams to the decoders of mdb_grid.hip (macaque_deferred_sum): how
// many values of segment i that would be - none if it does not qualify. `info` is what
// analyse_segment() said about it. With a time range: the values up to the last one inside it
// (regular timestamps only; what is summed there is what grid() would produce, so grid()'s length
// counts); without: the whole stream, if len() and grid() agree on its length.
__device__ __forceinline__ uint32_t mv_deferred_values(const DevSegments &s, uint64_t i, SegInfo info,
                                                       uint32_t min_values, const TimeRange &range) {
    const uint32_t bytes = s.values.views[i].x;
    if (!range.enabled) return mv_qualifies_for_sum(info, bytes, min_values) ? info.desc.n_model : 0u;
    if (!(info.desc.flags & FLAG_REGULAR)) return 0u;
    apply_time_range(s, i, info, range);
    return mv_qualifies(info, bytes, min_values) ? info.desc.n_visible : 0u;
}

// Under a time range, with cursors into the batch's MacaqueV streams: is segment i (analysed: `info`) aggregated piece
// by piece (k_agg_mv_range, mdb_grid.hip) rather than by k_agg_range's lane? Evaluated identically by both; the
// caller adds that the segment has pieces in the index.
__device__ __forceinline__ bool mv_range_by_pieces(const DevSegments &s, uint64_t i, const SegInfo &info) {
    const SegDesc &d = info.desc;
    return !info.error && (d.flags & FLAG_TYPE_MASK) == MDB_MACAQUE_V_ID && (d.flags & FLAG_REGULAR) &&
           !(d.flags & FLAG_HAS_RESIDUALS) && d.n_model == d.n_total;
}

// ... and the residual tail of a PMC-Mean or Swing segment (k_agg_range then takes the model's points only)?
__device__ __forceinline__ bool mv_range_tail_by_pieces(const DevSegments &s, uint64_t i, const SegInfo &info) {
    const SegDesc &d = info.desc;
    const uint32_t type = d.flags & FLAG_TYPE_MASK;
    return !info.error && (type == MDB_PMC_MEAN_ID || type == MDB_SWING_ID) && (d.flags & FLAG_REGULAR) &&
           (d.flags & FLAG_HAS_RESIDUALS) && d.n_model < d.n_total;
}

// Does the walk of the irregular timestamp streams (k_grid_ts_count<SUMS>, mdb_grid.hip) add up segment i's
// values for the aggregates? A Swing segment without residuals: swing::sum needs every timestamp of such a
// segment (swing.rs:283-299) and nothing else. Evaluated identically by the walk and by k_agg_segments.
__device__ __forceinline__ bool ts_walk_adds(const DevSegments &s, uint64_t i) {
    return s.model_type_id[i] == MDB_SWING_ID && (int32_t)s.residuals.views[i].x == 0;
}

// ... and, under a time range, aggregate its points inside the range (k_grid_ts_count<WALK_RANGE>)? PMC-Mean and
// Swing segments without residuals: their values follow from the timestamps alone.
__device__ __forceinline__ bool ts_walk_aggregates_range(const DevSegments &s, uint64_t i) {
    const int32_t type = s.model_type_id[i];
    return (type == MDB_PMC_MEAN_ID || type == MDB_SWING_ID) && (int32_t)s.residuals.views[i].x == 0;
}
struct TsWalkRange { // the points of a segment inside a time range, as GridExec + filter + aggregate see them
    double sum;      // of the f32 values, added up in f64 in the order of the points
    long long count;
    float min, max;
};

// (with_sums, !range.enabled): *sums; (range.enabled): *ranges, and only the segments that reach into the range
// are walked; *totals in every case, for the segments that were walked.
int ts_walk_for_aggregates(mdb_ctx *ctx, const mdb_segments *in, const DevSegments &s, bool with_sums, TimeRange range,
                           const uint32_t **totals, const double **sums, const TsWalkRange **ranges,
                           const unsigned int **error_word);
// The same under a time range for a batch that stays on the device (`kept`: its sidecar): the segments the range
// contains whole take what a walk over the whole time axis found once, only the ones it cuts are walked.
// *available = false: nothing was kept and nothing can be (a malformed stream: the caller walks as for any batch).
int ts_range_from_kept(mdb_ctx *ctx, const mdb_segments *in, const DevSegments &s, TimeRange range, MvIndex &kept,
                       const uint32_t **totals, const TsWalkRange **ranges, const unsigned int **error_word_out, bool *available);

int mv_index_range_totals(mdb_ctx *ctx, const DevSegments &s, TimeRange range, const MvIndex &index, DeferredTotals *totals);
int macaque_deferred(mdb_ctx *ctx, const DevSegments &s, TimeRange range, uint32_t min_values, bool forced,
                     uint64_t n_streams, uint64_t n_values, uint64_t n_bytes, bool *handled,
                     DeferredTotals *totals, const unsigned long long *by_pieces = nullptr);

} // namespace mdb
