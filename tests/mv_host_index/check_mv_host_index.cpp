// TEST INFRASTRUCTURE (CPU, no GPU): the cursors that a call's host threads leave in the MacaqueV streams of a host
// batch (modelardb-rs_amd/csrc/mdb_mv_host_index.cpp) against the oracle. Streams are made by the oracle's encoder
// (oracle/mdb_oracle.cpp, macaque_v.rs:76-214) from several kinds of data and wrapped into segments the way the
// fitter stores them; from every cursor the values of its piece are decoded again by the few lines below - a
// restatement of what k_grid_mv_pieces does with a cursor - and have to be the oracle's own decode of the stream,
// bit for bit. Built by tests/test_mv_host_index_cpu.py, plain and under AddressSanitizer / UBSan / ThreadSanitizer.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../modelardb-rs_amd/csrc/mdb_host_side.hpp"
#include "../../oracle/mdb_oracle.h"

// What the rest of the library provides to mdb_pipeline.cpp (whose thread pool is what this test needs of it): the
// grid pipeline itself is never started here.
namespace mdb {
thread_local std::string g_last_error;
GridPipeline *ctx_pipeline(mdb_ctx *) { return nullptr; }
GridPipeline *ctx_pipeline_detach(mdb_ctx *) { return nullptr; }
GridPipeline *ctx_pipeline_install(mdb_ctx *, GridPipeline *pipeline) { return pipeline; }
int grid_batch_owned_list(mdb_ctx *, const mdb_segments *const *, uint32_t, TimeRangeArg, bool, uint64_t, mdb_grid_result **) {
    return fail("no kernels in this test");
}
} // namespace mdb
extern "C" {
int mdb_clone(mdb_ctx *, mdb_ctx **) { return 1; }
int mdb_close(mdb_ctx *) { return 0; }
void mdb_grid_result_free(mdb_grid_result *) {}
}

namespace {

struct Bits {
    const uint8_t *bytes;
    uint64_t n_bytes, used;
    uint32_t get(uint32_t count) { // MSB first, zeros behind the end
        uint32_t value = 0;
        for (uint32_t k = 0; k < count; k++, used++) {
            const uint64_t byte = used >> 3;
            const uint32_t bit = byte < n_bytes ? (bytes[byte] >> (7 - (used & 7))) & 1u : 0u;
            value = (value << 1) | bit;
        }
        return value;
    }
};

struct Stream {
    std::vector<uint8_t> bytes;
    std::vector<float> decoded;
    uint32_t n;
};

int failures = 0;
void expect(bool ok, const char *what, uint64_t a = 0, uint64_t b = 0) {
    if (ok) return;
    if (failures++ < 20) std::fprintf(stderr, "MISMATCH: %s (%llu, %llu)\n", what, (unsigned long long)a, (unsigned long long)b);
}

std::vector<float> make_values(std::mt19937 &rng, uint32_t n, int kind) {
    std::vector<float> v(n);
    std::uniform_real_distribution<float> noise(-1.0f, 1.0f);
    for (uint32_t i = 0; i < n; i++) {
        switch (kind) {
        case 0: v[i] = 100.0f + 10.0f * std::sin(i / 200.0f) + 0.05f * noise(rng); break;   // the benchmark's shape
        case 1: v[i] = 42.0f; break;                                                          // repeats only
        case 2: v[i] = 1000.0f * noise(rng); break;                                           // every window new
        case 3: { uint32_t bits = rng(); std::memcpy(&v[i], &bits, 4); break; }               // any bit pattern
        default: v[i] = (i % 97 == 0) ? 1e30f * noise(rng) : (float)(i / 50);                 // steps with spikes
        }
    }
    return v;
}

} // namespace

// host_copy_streaming (mdb_pipeline.cpp: the gather of mdb_compress_chunk_list): every length up to a few hundred bytes
// from and to every alignment, guard bytes on either side untouched (and the sanitizers' eyes on the loads).
static int check_streaming_copy() {
    std::vector<uint8_t> from(1024), to(1024);
    for (size_t i = 0; i < from.size(); i++) from[i] = (uint8_t)(i * 131 + 7);
    for (size_t n : {size_t(0), size_t(1), size_t(3), size_t(15), size_t(16), size_t(17), size_t(63), size_t(64), size_t(65), size_t(127),
                     size_t(200), size_t(333)})
        for (size_t a = 0; a < 32; a += 3)
            for (size_t b = 0; b < 32; b += 5) {
                std::fill(to.begin(), to.end(), (uint8_t)0x5c);
                mdb::host_copy_streaming(to.data() + 64 + b, from.data() + a, n);
                for (size_t i = 0; i < to.size(); i++) {
                    const bool inside = i >= 64 + b && i < 64 + b + n;
                    if (to[i] != (inside ? from[a + i - 64 - b] : (uint8_t)0x5c)) {
                        std::fprintf(stderr, "host_copy_streaming: %zu bytes from +%zu to +%zu: byte %zu\n", n, a, b, i);
                        return 1;
                    }
                }
            }
    return 0;
}

int main() {
    if (check_streaming_copy()) return 3;
    setenv("MDB_GRID_MV_HOST_MIN_VALUES", "1", 1); // every stream, not only the long ones
    std::mt19937 rng(20261003);
    const mdb_error_bound lossless{MDB_EB_LOSSLESS, 0.0f};
    const uint32_t lengths[] = {1, 2, 63, 64, 65, 128, 129, 700, 4096, 65536, 70001};
    // Segments: MacaqueV models (values stream, first value raw), some with a residual tail (seeded stream).
    std::vector<Stream> models, tails;
    for (int kind = 0; kind < 5; kind++)
        for (uint32_t n : lengths) {
            if (n > 5000 && kind > 2) continue;
            Stream model;
            model.n = n;
            const std::vector<float> values = make_values(rng, n, kind);
            model.bytes.resize(8 + 6 * (size_t)n);
            uint64_t length = 0;
            if (ora_macaque_v_compress(lossless, values.data(), n, 0, 0.0f, model.bytes.data(), model.bytes.size(), &length,
                                       nullptr, nullptr, nullptr, nullptr, nullptr)) return 2;
            model.bytes.resize(length);
            model.decoded.resize(n);
            if (ora_macaque_v_grid(model.bytes.data(), length, n, 0, 0.0f, model.decoded.data())) return 2;
            Stream tail;
            tail.n = (kind + n) % 3 == 0 ? 0 : 1 + (uint32_t)(rng() % 255);
            if (tail.n) {
                const std::vector<float> residuals = make_values(rng, tail.n, (kind + 1) % 5);
                tail.bytes.resize(8 + 6 * (size_t)tail.n);
                const float seed = model.decoded[n - 1];
                if (ora_macaque_v_compress(lossless, residuals.data(), tail.n, 1, seed, tail.bytes.data(), tail.bytes.size(),
                                           &length, nullptr, nullptr, nullptr, nullptr, nullptr)) return 2;
                tail.bytes.resize(length);
                tail.decoded.resize(tail.n);
                if (ora_macaque_v_grid(tail.bytes.data(), length, tail.n, 1, seed, tail.decoded.data())) return 2;
                tail.bytes.push_back((uint8_t)tail.n); // (types.rs:266: the number of residuals ends the column)
            }
            models.push_back(std::move(model));
            tails.push_back(std::move(tail));
        }
    // Two batches that share their data buffers with padding in front (the views point into the middle of them).
    const size_t n_segments = models.size(), split = n_segments / 3;
    std::vector<uint8_t> value_buffer(37, 0xAA), residual_buffer(5, 0xBB), timestamp_buffer;
    std::vector<mdb_view16> value_views(n_segments), residual_views(n_segments), timestamp_views(n_segments);
    std::vector<int8_t> types(n_segments, (int8_t)MDB_MACAQUE_V_ID);
    std::vector<int64_t> starts(n_segments), ends(n_segments);
    std::vector<float> mins(n_segments, 0.0f), maxs(n_segments, 0.0f);
    auto make_view = [](std::vector<uint8_t> &buffer, const std::vector<uint8_t> &payload) {
        mdb_view16 view;
        std::memset(&view, 0, sizeof(view));
        view.length = (int32_t)payload.size();
        if (payload.size() <= 12) {
            if (!payload.empty()) std::memcpy(view.u.inlined, payload.data(), payload.size());
        } else {
            std::memcpy(view.u.ref.prefix, payload.data(), 4);
            view.u.ref.buffer_index = 0;
            view.u.ref.offset = (int32_t)buffer.size();
            buffer.insert(buffer.end(), payload.begin(), payload.end());
        }
        return view;
    };
    int irregular_segments = 0;
    for (size_t i = 0; i < n_segments; i++) {
        value_views[i] = make_view(value_buffer, models[i].bytes);
        residual_views[i] = make_view(residual_buffer, tails[i].bytes);
        const uint64_t total = models[i].n + tails[i].n;
        starts[i] = 1000 * (int64_t)i;
        if (i % 2 == 1 && total >= 3) {
            // irregular timestamps: the oracle's delta-of-delta stream (timestamps.rs:56-155), whose codes the host
            // threads count to know how many points - and so how many model values - the segment has
            std::vector<int64_t> timestamps(total);
            int64_t t = starts[i];
            for (uint64_t k = 0; k < total; k++) {
                timestamps[k] = t;
                const uint32_t kind = rng() % 100;
                t += kind < 70 ? 10 : (kind < 90 ? 1 + (int64_t)(rng() % 300) : (kind < 99 ? 1 + (int64_t)(rng() % 100000) : 5000000000ll));
            }
            ends[i] = timestamps[total - 1];
            std::vector<uint8_t> stream(16 + 10 * total);
            uint64_t length = 0;
            if (ora_compress_residual_timestamps(timestamps.data(), total, stream.data(), stream.size(), &length)) return 2;
            stream.resize(length);
            if (stream.empty() || (stream[0] & 0x80u) == 0) { // (came out regular after all)
                irregular_segments -= 1;
            }
            irregular_segments += 1;
            timestamp_views[i] = make_view(timestamp_buffer, stream);
            continue;
        }
        ends[i] = starts[i] + 10 * (int64_t)(total - 1);
        std::vector<uint8_t> length_bytes; // timestamps.rs:99-108: the length, big endian, as few bytes as it needs
        if (total > 2)
            for (int shift = 24; shift >= 0; shift -= 8)
                if ((total >> shift) != 0 || shift == 0) length_bytes.push_back((uint8_t)(total >> shift));
        if (!length_bytes.empty() && (length_bytes[0] & 0x80u)) length_bytes.insert(length_bytes.begin(), 0);
        timestamp_views[i] = make_view(timestamp_buffer, length_bytes);
        if (total == 2 && ends[i] == starts[i]) ends[i] += 10;
    }
    const uint8_t *value_buffers[1] = {value_buffer.data()}, *residual_buffers[1] = {residual_buffer.data()},
                  *timestamp_buffers[1] = {timestamp_buffer.data()};
    const int64_t value_sizes[1] = {(int64_t)value_buffer.size()}, residual_sizes[1] = {(int64_t)residual_buffer.size()},
                  timestamp_sizes[1] = {(int64_t)timestamp_buffer.size()};
    mdb_segments batches[2];
    for (int h = 0; h < 2; h++) {
        const size_t first = h == 0 ? 0 : split, count = h == 0 ? split : n_segments - split;
        mdb_segments &seg = batches[h];
        std::memset(&seg, 0, sizeof(seg));
        seg.n = count;
        seg.model_type_id = types.data() + first;
        seg.start_time = starts.data() + first;
        seg.end_time = ends.data() + first;
        seg.min_value = mins.data() + first;
        seg.max_value = maxs.data() + first;
        seg.timestamps = {timestamp_views.data() + first, timestamp_buffers, timestamp_sizes, 1};
        seg.values = {value_views.data() + first, value_buffers, value_sizes, 1};
        seg.residuals = {residual_views.data() + first, residual_buffers, residual_sizes, 1};
    }
    const mdb_segments *list[2] = {&batches[0], &batches[1]};
    std::vector<unsigned long long> piece_base;
    std::vector<mdb::MvCursor> cursors;
    for (int repetition = 0; repetition < 3; repetition++) mdb::mv_host_index(list, 2, &piece_base, &cursors);
    expect(piece_base.size() == n_segments + 1, "piece_base has rows + 1 entries", piece_base.size(), n_segments + 1);
    if (failures) return 1;
    uint64_t checked_values = 0;
    for (size_t i = 0; i < n_segments; i++) {
        const uint64_t value_pieces = (models[i].n + 63) / 64, tail_pieces = (tails[i].n + 63) / 64;
        expect(piece_base[i + 1] - piece_base[i] == value_pieces + tail_pieces, "pieces of a segment", i,
               piece_base[i + 1] - piece_base[i]);
        for (uint64_t p = 0; p < value_pieces + tail_pieces; p++) {
            const mdb::MvCursor &c = cursors[piece_base[i] + p];
            const bool residual = p >= value_pieces;
            const Stream &stream = residual ? tails[i] : models[i];
            const uint32_t first = (uint32_t)((residual ? p - value_pieces : p) * 64);
            expect(c.segment == i, "cursor.segment", c.segment, i);
            expect(c.point_index == (residual ? models[i].n : 0u) + first, "cursor.point_index", c.point_index, first);
            expect(c.n_values == std::min<uint32_t>(64, stream.n - first), "cursor.n_values", c.n_values, stream.n - first);
            expect(((c.window & mdb::MV_WINDOW_RESIDUAL) != 0) == residual, "cursor is of the tail", i, p);
            expect(((c.pad & mdb::MV_CURSOR_LAST_OF_STREAM) != 0) == (first + 64 >= stream.n),
                   "only the last piece of a stream says so", i, p);
            uint32_t seed_bits = 0; // (k_grid_mv_pieces: a MacaqueV model's tail starts from chain_seed, its values from 0)
            if (residual) seed_bits = c.chain_seed;
            if (residual) {
                uint32_t expected_seed;
                std::memcpy(&expected_seed, &models[i].decoded[models[i].n - 1], 4);
                expect(c.chain_seed == expected_seed, "cursor.chain_seed is the model's last value", c.chain_seed, expected_seed);
            }
            // decode the piece from the cursor (lean_decode_value)
            Bits bits{stream.bytes.data(), stream.bytes.size() - (residual ? 1u : 0u), c.bit_position};
            uint32_t last = seed_bits ^ c.xor_bits, leading = c.window & 255u, trailing = (c.window >> 8) & 255u;
            bool raw = (c.window & mdb::MV_WINDOW_RAW) != 0;
            for (uint32_t k = 0; k < c.n_values; k++) {
                uint32_t value_bits;
                if (raw) {
                    value_bits = bits.get(32);
                    raw = false;
                } else if (bits.get(1) == 0) {
                    value_bits = last ^ (bits.get(32 - leading - trailing) << trailing);
                } else if (bits.get(1) == 0) {
                    value_bits = last;
                } else {
                    leading = bits.get(5);
                    const uint32_t meaningful = bits.get(6);
                    trailing = 32 - meaningful - leading;
                    value_bits = last ^ (bits.get(meaningful) << trailing);
                }
                last = value_bits;
                uint32_t expected;
                std::memcpy(&expected, &stream.decoded[first + k], 4);
                expect(value_bits == expected, "a value decoded from its piece's cursor", i, first + k);
                checked_values++;
            }
        }
    }
    const size_t n_cursors = cursors.size();
    // Under a time range (the range aggregates' call): the segments that do not reach into it get no pieces, the
    // others exactly the ones they have without it.
    {
        std::vector<int64_t> sorted_starts(starts.begin(), starts.end());
        std::sort(sorted_starts.begin(), sorted_starts.end());
        const mdb::MvHostRange range{sorted_starts[n_segments / 3], sorted_starts[2 * n_segments / 3]};
        std::vector<unsigned long long> ranged_base;
        std::vector<mdb::MvCursor> ranged_cursors;
        mdb::mv_host_index(list, 2, &ranged_base, &ranged_cursors, &range);
        expect(ranged_base.size() == n_segments + 1, "piece_base under a range has rows + 1 entries", ranged_base.size(), n_segments + 1);
        size_t outside = 0;
        for (size_t i = 0; i < n_segments && ranged_base.size() == n_segments + 1; i++) {
            const bool reaches = !(ends[i] < range.lo || starts[i] > range.hi);
            const uint64_t expected = reaches ? piece_base[i + 1] - piece_base[i] : 0;
            expect(ranged_base[i + 1] - ranged_base[i] == expected, "pieces of a segment under a time range", i, expected);
            outside += reaches ? 0 : 1;
            for (uint64_t q = 0; q < expected && ranged_base[i + 1] - ranged_base[i] == expected; q++)
                expect(std::memcmp(&ranged_cursors[ranged_base[i] + q], &cursors[piece_base[i] + q], sizeof(mdb::MvCursor)) == 0,
                       "a cursor under a time range", i, q);
        }
        expect(outside > 0 && outside < n_segments, "the range leaves some segments out", outside, n_segments);
    }
    // A malformed stream (a window that cannot be: `11`, 31 leading zeros, 63 meaningful bits): no index at all.
    size_t victim = 0;
    while (models[victim].n < 2) victim++;
    std::vector<uint8_t> bad(4, 0);
    bad.insert(bad.end(), 16, 0xFF);
    value_views[victim] = make_view(value_buffer, bad);
    const uint8_t *grown_buffers[1] = {value_buffer.data()};
    const int64_t grown_sizes[1] = {(int64_t)value_buffer.size()};
    batches[0].values = {value_views.data(), grown_buffers, grown_sizes, 1};
    batches[1].values = {value_views.data() + split, grown_buffers, grown_sizes, 1};
    mdb::mv_host_index(list, 2, &piece_base, &cursors);
    expect(piece_base.empty() && cursors.empty(), "a malformed stream leaves no index", piece_base.size(), cursors.size());
    // A view that points outside its data buffers: nothing is dereferenced, no index (the upload reports it).
    value_views[victim] = value_views[victim + 1];
    value_views[victim].length = 1 << 20;
    value_views[victim].u.ref.offset = (int32_t)value_buffer.size() - 8;
    mdb::mv_host_index(list, 2, &piece_base, &cursors);
    expect(piece_base.empty() && cursors.empty(), "a view outside its buffers leaves no index", piece_base.size(), cursors.size());
    timestamp_views[victim + 2].length = 4000;
    timestamp_views[victim + 2].u.ref.buffer_index = 7;
    value_views[victim] = value_views[victim + 1];
    mdb::mv_host_index(list, 2, &piece_base, &cursors);
    expect(piece_base.empty() && cursors.empty(), "a view into a buffer that does not exist leaves no index", piece_base.size(), cursors.size());
    expect(irregular_segments >= 10, "segments with irregular timestamps among them", irregular_segments, 10);
    std::printf("%s: %zu segments (%d with irregular timestamps), %llu values decoded from %zu cursors\n",
                failures ? "FAILED" : "ok", n_segments, irregular_segments, (unsigned long long)checked_values, n_cursors);
    return failures ? 1 : 0;
}
