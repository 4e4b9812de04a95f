// Measuring tool (CPU, no GPU): the host threads' walk of long MacaqueV streams (modelardb-rs_amd/csrc/
// mdb_mv_host_index.cpp) timed on streams made by the oracle's encoder: nanoseconds per value per thread and in all.
// Built by scripts/r06/host_walk/Makefile against the library's own sources; prints a checksum of the cursors so that
// two builds of the walk can be compared.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../../modelardb-rs_amd/csrc/mdb_host_side.hpp"
#include "../../../oracle/mdb_oracle.h"

namespace mdb {
thread_local std::string g_last_error;
GridPipeline *ctx_pipeline(mdb_ctx *) { return nullptr; }
GridPipeline *ctx_pipeline_detach(mdb_ctx *) { return nullptr; }
GridPipeline *ctx_pipeline_install(mdb_ctx *, GridPipeline *pipeline) { return pipeline; }
int grid_batch_owned_list(mdb_ctx *, const mdb_segments *const *, uint32_t, TimeRangeArg, bool, uint64_t, mdb_grid_result **) {
    return fail("no kernels in this tool");
}
} // namespace mdb
extern "C" {
int mdb_clone(mdb_ctx *, mdb_ctx **) { return 1; }
int mdb_close(mdb_ctx *) { return 0; }
void mdb_grid_result_free(mdb_grid_result *) {}
}

int main(int argc, char **argv) {
    const uint32_t n_streams = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 256, n = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 65536;
    const int kind = argc > 3 ? std::atoi(argv[3]) : 0;
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> noise(-1.0f, 1.0f);
    const mdb_error_bound lossless{MDB_EB_LOSSLESS, 0.0f};
    std::vector<uint8_t> buffer(64, 0);
    std::vector<mdb_view16> value_views(n_streams), residual_views(n_streams), timestamp_views(n_streams);
    std::vector<uint8_t> timestamp_buffer(16, 0);
    for (uint32_t s = 0; s < n_streams; s++) {
        std::vector<float> v(n);
        for (uint32_t i = 0; i < n; i++)
            v[i] = kind == 0 ? 100.0f + 10.0f * std::sin(i / 200.0f) + 0.05f * noise(rng)
                             : (kind == 1 ? (float)(int)(20.0f * noise(rng)) : 1000.0f * noise(rng));
        std::vector<uint8_t> bytes(8 + 6 * (size_t)n);
        uint64_t length = 0;
        if (ora_macaque_v_compress(lossless, v.data(), n, 0, 0.0f, bytes.data(), bytes.size(), &length, nullptr, nullptr, nullptr,
                                   nullptr, nullptr)) return 2;
        mdb_view16 view;
        std::memset(&view, 0, sizeof(view));
        view.length = (int32_t)length;
        std::memcpy(view.u.ref.prefix, bytes.data(), 4);
        view.u.ref.buffer_index = 0;
        view.u.ref.offset = (int32_t)buffer.size();
        buffer.insert(buffer.end(), bytes.begin(), bytes.begin() + length);
        value_views[s] = view;
        std::memset(&residual_views[s], 0, sizeof(mdb_view16));
        std::vector<uint8_t> length_bytes;
        for (int shift = 24; shift >= 0; shift -= 8)
            if ((n >> shift) != 0 || shift == 0) length_bytes.push_back((uint8_t)(n >> shift));
        if (length_bytes[0] & 0x80u) length_bytes.insert(length_bytes.begin(), 0);
        std::memset(&timestamp_views[s], 0, sizeof(mdb_view16));
        timestamp_views[s].length = (int32_t)length_bytes.size();
        std::memcpy(timestamp_views[s].u.inlined, length_bytes.data(), length_bytes.size());
    }
    std::vector<int8_t> types(n_streams, (int8_t)MDB_MACAQUE_V_ID);
    std::vector<int64_t> starts(n_streams), ends(n_streams);
    std::vector<float> mins(n_streams, 0.0f), maxs(n_streams, 0.0f);
    for (uint32_t s = 0; s < n_streams; s++) {
        starts[s] = 0;
        ends[s] = 10 * (int64_t)(n - 1);
    }
    const uint8_t *value_buffers[1] = {buffer.data()}, *timestamp_buffers[1] = {timestamp_buffer.data()};
    const int64_t value_sizes[1] = {(int64_t)buffer.size()}, timestamp_sizes[1] = {(int64_t)timestamp_buffer.size()};
    mdb_segments seg;
    std::memset(&seg, 0, sizeof(seg));
    seg.n = n_streams;
    seg.model_type_id = types.data();
    seg.start_time = starts.data();
    seg.end_time = ends.data();
    seg.min_value = mins.data();
    seg.max_value = maxs.data();
    seg.timestamps = {timestamp_views.data(), timestamp_buffers, timestamp_sizes, 1};
    seg.values = {value_views.data(), value_buffers, value_sizes, 1};
    seg.residuals = {residual_views.data(), value_buffers, value_sizes, 1};
    const mdb_segments *list[1] = {&seg};
    std::vector<unsigned long long> piece_base;
    std::vector<mdb::MvCursor> cursors;
    mdb::mv_host_index(list, 1, &piece_base, &cursors);
    double best = 1e30;
    for (int repetition = 0; repetition < 7; repetition++) {
        const auto from = std::chrono::steady_clock::now();
        mdb::mv_host_index(list, 1, &piece_base, &cursors);
        best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - from).count());
    }
    uint64_t checksum = 1469598103934665603ull;
    const uint8_t *raw = reinterpret_cast<const uint8_t *>(cursors.data());
    for (size_t k = 0; k < cursors.size() * sizeof(mdb::MvCursor); k++) checksum = (checksum ^ raw[k]) * 1099511628211ull;
    const double values = (double)n_streams * n;
    std::printf("kind %d: %u streams of %u values, %.2f bits a value: %.2f ms, %.2f ns a value in all; %zu cursors, checksum %016llx\n",
                kind, n_streams, n, 8.0 * (buffer.size() - 64) / values, best * 1e3, best * 1e9 / values, cursors.size(),
                (unsigned long long)checksum);
    return cursors.empty() ? 1 : 0;
}
