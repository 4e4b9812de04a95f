// The cursor index of ONE call over host batches, made by host threads (see the comment below): plain C++ without a
// line of device code, compiled into libmdb_hip.so and, with sanitizers, into the CPU test of tests/
// test_mv_host_index_cpu.py.
#include "mdb_host_side.hpp"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

namespace mdb {

namespace {
constexpr uint32_t COUNT_LIMIT = 0x7fffffffu; // COUNT_MASK of the device code: points of a segment
// ---- the cursor index of ONE call over host batches, made by host threads -------------------------------------
//
// A batch that comes from the host is decoded once: an index walk on the GPU (one lane per stream, k_mv_index_walk)
// would cost what the serial decode costs. But the long streams are what hurts - a 65 536-value MacaqueV segment
// (a whole ingest buffer of noise under a lossless bound) keeps one lane busy for 10 ms however idle the GPU is - and
// a CPU core walks such a stream in a third of a millisecond. So the call's host threads walk the long streams of the
// batches (MacaqueV segments whose model values number at least MV_HOST_MIN_VALUES; their residual tails with them;
// where the timestamps are irregular the points are counted from the delta-of-delta stream first) while nothing else
// needs them, the cursors travel with the batch (0.5 bytes per value), and
// k_grid_mv_pieces decodes those streams piece by piece; everything else is k_grid_serial's as before. Same cursors
// as k_mv_index_walk leaves (macaque_v.rs:272-323 walked, not decoded to floats). A malformed stream or segment:
// no index at all, the kernels that report it run as before.
constexpr uint32_t MV_HOST_MIN_VALUES = 512;

struct HostStreamBits { // MSB-first bits of a byte string; zeros behind its end
    const uint8_t *bytes;
    uint64_t n_bytes;
    uint64_t used = 0;
    uint64_t peek64() const {
        const uint64_t byte = used >> 3, shift = used & 7u;
        uint64_t window = 0;
        uint8_t next = 0;
        if (byte + 9 <= n_bytes) {
            std::memcpy(&window, bytes + byte, 8);
            window = __builtin_bswap64(window);
            next = bytes[byte + 8];
        } else {
            for (uint64_t k = 0; k < 8; k++) window = (window << 8) | (byte + k < n_bytes ? bytes[byte + k] : 0u);
        }
        return shift ? (window << shift) | ((uint64_t)next >> (8u - shift)) : window;
    }
    // The next 57 bits at least, on top of the word (the bits below them are not the stream's): ONE unaligned load
    // where eight bytes are left, which is enough for a whole MacaqueV code (13 bits of header, 32 of payload).
    uint64_t peek57() const {
        const uint64_t byte = used >> 3;
        if (byte + 8 <= n_bytes) {
            uint64_t window;
            std::memcpy(&window, bytes + byte, 8);
            return __builtin_bswap64(window) << (used & 7u);
        }
        return peek64();
    }
};

struct HostViewBytes {
    const uint8_t *data;
    uint64_t length;
};
HostViewBytes host_view_bytes(const mdb_binview_col &col, uint64_t i) {
    const mdb_view16 &view = col.views[i];
    if (view.length <= 12) return {view.u.inlined, (uint64_t)std::max(view.length, 0)};
    return {col.buffers[view.u.ref.buffer_index] + view.u.ref.offset, (uint64_t)view.length};
}

// Every view inside its column's data buffers (validate_views_host's rules, without the error text: the upload
// that follows makes that check again and reports what it finds).
bool host_views_are_valid(const mdb_binview_col &col, uint64_t n) {
    if (n > 0 && !col.views) return false;
    if (col.n_buffers < 0 || (col.n_buffers > 0 && (!col.buffers || !col.buffer_sizes))) return false;
    for (uint64_t i = 0; i < n; i++) {
        const mdb_view16 &view = col.views[i];
        if (view.length < 0) return false;
        if (view.length <= 12) continue;
        const int32_t buffer = view.u.ref.buffer_index;
        const int64_t offset = view.u.ref.offset;
        if (buffer < 0 || buffer >= col.n_buffers || offset < 0 || col.buffer_sizes[buffer] < 0 ||
            offset + (int64_t)view.length > col.buffer_sizes[buffer] || !col.buffers[buffer])
            return false;
    }
    return true;
}

// Points of a segment with regular timestamps (analyse_segment's rules); false: not regular, or malformed.
bool host_regular_points(const mdb_segments &seg, uint64_t i, uint32_t *n_total) {
    const HostViewBytes ts = host_view_bytes(seg.timestamps, i);
    const int64_t start = seg.start_time[i], end = seg.end_time[i];
    if (ts.length == 0) {
        *n_total = start == end ? 1u : 2u;
        return true;
    }
    if ((ts.data[0] & 0x80u) != 0 || ts.length > 8) return false;
    uint64_t length = 0;
    for (uint64_t k = 0; k < ts.length; k++) length = (length << 8) | ts.data[k];
    if (length < 2) return false;
    if (end < start) {
        *n_total = 0;
        return true;
    }
    const uint64_t span = (uint64_t)(end - start), interval = span / (length - 1);
    if (interval == 0) return false;
    const uint64_t produced = span / interval + 1;
    if (produced > COUNT_LIMIT) return false;
    *n_total = (uint32_t)produced;
    return true;
}

// Points of a segment with irregular timestamps: the codes of its delta-of-delta stream (timestamps.rs:228-292;
// decode_irregular_timestamps of the device code) counted, plus its first and its last point. false: malformed.
bool host_irregular_points(const HostViewBytes &ts, uint32_t *n_total) {
    HostStreamBits bits{ts.data, ts.length};
    const uint64_t total_bits = ts.length * 8;
    bits.used = 1; // (bit 0: the flag "irregular")
    uint64_t count = 1;
    while (bits.used < total_bits) {
        const uint64_t window = bits.peek64(), left = total_bits - bits.used;
        if ((window >> 63) == 0) { // a run of `0` codes: the delta repeats
            const uint64_t run = std::min<uint64_t>(window ? (uint64_t)__builtin_clzll(window) : 64u, left);
            bits.used += run;
            count += run;
            if (count > COUNT_LIMIT) return false;
            continue;
        }
        // `10`, `110`, `1110`, `11110` or `11111` in front of 7, 9, 12, 32 or 64 bits
        const uint64_t leading_ones = ~window ? (uint64_t)__builtin_clzll(~window) : 64u;
        const uint64_t ones = std::min<uint64_t>(std::min<uint64_t>(leading_ones, 5u), left);
        bits.used += ones < 5 && ones < left ? ones + 1 : ones; // (the `0` behind fewer than five ones, if there is one)
        const uint64_t remaining = total_bits - bits.used;
        if (remaining < 7) break; // (the ones that pad the last byte)
        static const uint32_t widths[6] = {0, 7, 9, 12, 32, 64};
        if (remaining < widths[ones]) return false;
        bits.used += widths[ones];
        count += 1;
        if (count > COUNT_LIMIT) return false;
    }
    count += 1; // (the last point is end_time, which is not stored)
    if (count > COUNT_LIMIT) return false;
    *n_total = (uint32_t)count;
    return true;
}

struct HostIndexJob {
    const mdb_segments *const *ins;
    std::vector<uint64_t> first_row;     // of every input batch in the joint batch
    std::vector<uint64_t> chosen;        // joint row of every segment that gets cursors
    std::vector<uint32_t> chosen_input;  // its input batch
    std::vector<uint32_t> n_values, n_residuals, n_model;
    std::vector<unsigned long long> *piece_base;
    std::vector<MvCursor> *cursors;
    std::atomic<uint64_t> next{0};
    std::atomic<int> malformed{0};
};

void host_index_share(unsigned, void *arg) {
    HostIndexJob &job = *static_cast<HostIndexJob *>(arg);
    while (true) {
        const uint64_t k = job.next.fetch_add(1);
        if (k >= job.chosen.size() || job.malformed.load()) return;
        const uint64_t joint = job.chosen[k];
        const mdb_segments &seg = *job.ins[job.chosen_input[k]];
        const uint64_t i = joint - job.first_row[job.chosen_input[k]];
        MvCursor *out = job.cursors->data() + (*job.piece_base)[joint];
        uint32_t last = 0, chain_seed = 0;
        for (int which = 0; which < 2; which++) { // the model's values, then the residual tail
            const bool residual = which == 1;
            uint32_t remaining = residual ? job.n_residuals[k] : job.n_values[k];
            if (remaining == 0) continue;
            HostViewBytes bytes = host_view_bytes(residual ? seg.residuals : seg.values, i);
            if (residual) {
                if (bytes.length < 2) { job.malformed = 1; return; }
                bytes.length -= 1; // (its last byte is the number of residuals)
                chain_seed = last; // (0 unless a MacaqueV model's values have just been walked)
                last = 0;
            } else if (bytes.length == 0) {
                job.malformed = 1;
                return;
            }
            HostStreamBits bits{bytes.data, bytes.length};
            uint32_t leading = 255, trailing = 0, position = residual ? job.n_model[k] : 0u, in_stream = 0;
            bool raw = !residual;
            while (remaining > 0) {
                if (in_stream % MV_PIECE_VALUES == 0) {
                    MvCursor cursor;
                    cursor.bit_position = (uint32_t)bits.used;
                    cursor.xor_bits = last;
                    cursor.segment = (uint32_t)joint;
                    cursor.point_index = position;
                    cursor.n_values = std::min(remaining, MV_PIECE_VALUES);
                    cursor.window = (leading & 255u) | ((trailing & 255u) << 8) | (residual ? MV_WINDOW_RESIDUAL : 0u) |
                                    (raw ? MV_WINDOW_RAW : 0u);
                    cursor.chain_seed = residual ? chain_seed : 0u;
                    cursor.pad = remaining <= MV_PIECE_VALUES ? MV_CURSOR_LAST_OF_STREAM : 0u; // (checked by k_grid_mv_pieces)
                    *out++ = cursor;
                }
                // one code (ring_decode_value): header and payload out of one look at the stream (45 bits at most)
                const uint64_t window = bits.peek57();
                const uint32_t top = (uint32_t)(window >> 51);
                const bool c0 = (top >> 12) != 0u, c1 = ((top >> 11) & 1u) != 0u;
                const bool opens = !raw && c0 && c1, repeats = !raw && c0 && !c1;
                const uint32_t header_bits = raw ? 0u : (c0 ? (c1 ? 13u : 2u) : 1u);
                if (opens) {
                    leading = (top >> 6) & 31u;
                    trailing = 32u - (top & 63u) - leading;
                }
                uint32_t meaningful = 32u - leading - trailing;
                if (!raw && !repeats && (meaningful > 32u || trailing > 31u)) { job.malformed = 1; return; }
                meaningful = raw ? 32u : (repeats ? 0u : meaningful);
                const uint32_t payload = meaningful ? (uint32_t)((window << header_bits) >> (64u - meaningful)) : 0u;
                bits.used += header_bits + meaningful;
                last = raw ? payload : (last ^ (payload << (trailing & 31u)));
                raw = false;
                position += 1;
                remaining -= 1;
                in_stream += 1;
                if (bits.used > bytes.length * 8) { job.malformed = 1; return; } // (past the end: not another code)
            }
            if (bits.used > bytes.length * 8 || bits.used > 0xffffffffull) { job.malformed = 1; return; }
        }
    }
}

} // namespace

// piece_base (rows + 1) and cursors of the call's long streams; both stay empty when there is nothing to index
// (MDB_GRID_MV_INDEX=0 included).
static void mv_host_index_or_throw(const mdb_segments *const *ins, uint32_t n_ins, std::vector<unsigned long long> *piece_base,
                                   std::vector<MvCursor> *cursors, const MvHostRange *range);

static uint32_t mv_host_min_values() {
    // (MDB_GRID_MV_HOST_MIN_VALUES: streams from that many values on: tests index short ones too)
    const char *text = option_text("MDB_GRID_MV_HOST_MIN_VALUES");
    const long long wanted = text ? std::atoll(text) : 0;
    return wanted > 0 ? (uint32_t)std::min<long long>(wanted, 1 << 30) : MV_HOST_MIN_VALUES;
}

bool mv_host_index_worthwhile(const mdb_segments *const *ins, uint32_t n_ins) {
    const char *setting = option_text("MDB_GRID_MV_INDEX");
    if (setting && std::strcmp(setting, "0") == 0) return false;
    // (a stream of k values: the first raw, every further one at least two bits)
    const uint64_t least_bytes = (32ull + 2ull * (mv_host_min_values() - 1) + 7) / 8;
    for (uint32_t h = 0; h < n_ins; h++) {
        const mdb_segments &seg = *ins[h];
        if (seg.n == 0) continue;
        if (!seg.model_type_id || !seg.values.views) return false;
        for (uint64_t i = 0; i < seg.n; i++)
            if (seg.model_type_id[i] == MDB_MACAQUE_V_ID && (uint64_t)(uint32_t)seg.values.views[i].length >= least_bytes) return true;
    }
    return false;
}

// (an allocation that fails - the cursors of a huge batch - means "no index", never an exception through the C ABI or
// a terminated process when the walk runs on a thread of its own)
void mv_host_index(const mdb_segments *const *ins, uint32_t n_ins, std::vector<unsigned long long> *piece_base,
                   std::vector<MvCursor> *cursors, const MvHostRange *range) {
    try {
        mv_host_index_or_throw(ins, n_ins, piece_base, cursors, range);
    } catch (...) {
        piece_base->clear();
        cursors->clear();
    }
}

static void mv_host_index_or_throw(const mdb_segments *const *ins, uint32_t n_ins, std::vector<unsigned long long> *piece_base,
                                   std::vector<MvCursor> *cursors, const MvHostRange *range) {
    piece_base->clear();
    cursors->clear();
    const char *setting = option_text("MDB_GRID_MV_INDEX");
    if (setting && std::strcmp(setting, "0") == 0) return;
    const uint32_t min_values = mv_host_min_values();
    HostIndexJob job;
    job.ins = ins;
    uint64_t rows = 0;
    for (uint32_t h = 0; h < n_ins; h++) {
        job.first_row.push_back(rows);
        rows += ins[h]->n;
    }
    if (rows == 0 || rows > 0xfffffff0ull) return;
    for (uint32_t h = 0; h < n_ins; h++) { // (nothing below looks at a byte before the views are known to be sound)
        const mdb_segments &seg = *ins[h];
        if (seg.n > 0 && (!seg.model_type_id || !seg.start_time || !seg.end_time)) return;
        if (!host_views_are_valid(seg.timestamps, seg.n) || !host_views_are_valid(seg.values, seg.n) ||
            !host_views_are_valid(seg.residuals, seg.n))
            return;
    }
    std::vector<unsigned long long> pieces(rows + 1, 0);
    // Segments with irregular timestamps: their number of points is the number of codes of a delta-of-delta
    // stream. The streams that can be long enough (a code is at least a bit) are counted by the host threads first.
    struct CountJob {
        const mdb_segments *const *ins;
        std::vector<std::pair<uint32_t, uint64_t>> segments; // (input, row)
        std::vector<uint32_t> points;                         // 0: malformed
        std::atomic<uint64_t> next{0};
    } counting;
    counting.ins = ins;
    for (uint32_t h = 0; h < n_ins; h++) {
        const mdb_segments &seg = *ins[h];
        for (uint64_t i = 0; i < seg.n; i++) {
            if (seg.model_type_id[i] != MDB_MACAQUE_V_ID) continue;
            if (range && (seg.end_time[i] < range->lo || seg.start_time[i] > range->hi)) continue;
            const HostViewBytes ts = host_view_bytes(seg.timestamps, i);
            if (ts.length > 0 && (ts.data[0] & 0x80u) != 0 && ts.length * 8 + 1 >= min_values) counting.segments.push_back({h, i});
        }
    }
    counting.points.assign(counting.segments.size(), 0);
    if (!counting.segments.empty())
        host_parallel((unsigned)std::min<uint64_t>(host_parallel_width(), counting.segments.size()),
                      [](unsigned, void *arg) {
                          CountJob &job = *static_cast<CountJob *>(arg);
                          for (uint64_t k = job.next.fetch_add(1); k < job.segments.size(); k = job.next.fetch_add(1)) {
                              const mdb_segments &seg = *job.ins[job.segments[k].first];
                              uint32_t n_total = 0;
                              if (host_irregular_points(host_view_bytes(seg.timestamps, job.segments[k].second), &n_total))
                                  job.points[k] = n_total;
                          }
                      },
                      &counting);
    size_t next_counted = 0;
    for (uint32_t h = 0; h < n_ins; h++) {
        const mdb_segments &seg = *ins[h];
        for (uint64_t i = 0; i < seg.n; i++) {
            if (seg.model_type_id[i] != MDB_MACAQUE_V_ID) continue; // (residual tails alone are short: at most 255 values)
            // (a call under a time range, the aggregates': the segments that do not reach into it are nobody's)
            if (range && (seg.end_time[i] < range->lo || seg.start_time[i] > range->hi)) continue;
            uint32_t n_total = 0;
            if (next_counted < counting.segments.size() && counting.segments[next_counted] == std::make_pair(h, i)) {
                n_total = counting.points[next_counted++];
                if (n_total == 0) continue; // (malformed: the kernels that report it take the segment)
            } else if (!host_regular_points(seg, i, &n_total)) {
                continue;
            }
            const HostViewBytes residuals = host_view_bytes(seg.residuals, i);
            const uint32_t n_res = residuals.length > 0 ? residuals.data[residuals.length - 1] : 0u;
            if (n_res > n_total || residuals.length == 1) continue;
            const uint32_t n_model = n_total - n_res;
            if (n_model < min_values) continue;
            // The point count comes from the metadata (start, end, the length field): a row whose payload cannot hold
            // that many codes - the first value raw, every further one at least two bits - is malformed and not worth a
            // walk of 2^31 steps and a cursor array to match; the kernels that report it take the segment.
            const HostViewBytes stream = host_view_bytes(seg.values, i);
            if (32ull + 2ull * (n_model - 1) > 8ull * stream.length) continue;
            if (n_res > 0 && 2ull * n_res > 8ull * (residuals.length - 1)) continue;
            job.chosen.push_back(job.first_row[h] + i);
            job.chosen_input.push_back(h);
            job.n_values.push_back(n_model);
            job.n_residuals.push_back(n_res);
            job.n_model.push_back(n_model);
            pieces[job.first_row[h] + i] = (n_model + MV_PIECE_VALUES - 1) / MV_PIECE_VALUES + (n_res + MV_PIECE_VALUES - 1) / MV_PIECE_VALUES;
        }
    }
    if (job.chosen.empty()) return;
    unsigned long long total = 0;
    for (uint64_t r = 0; r < rows; r++) { // exclusive scan
        const unsigned long long here = pieces[r];
        pieces[r] = total;
        total += here;
    }
    pieces[rows] = total;
    std::vector<MvCursor> walked((size_t)total);
    job.piece_base = &pieces;
    job.cursors = &walked;
    host_parallel((unsigned)std::min<uint64_t>(host_parallel_width(), job.chosen.size()), host_index_share, &job);
    if (job.malformed.load()) return;
    piece_base->swap(pieces);
    cursors->swap(walked);
}

} // namespace mdb
