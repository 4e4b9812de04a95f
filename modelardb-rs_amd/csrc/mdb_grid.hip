// mdb_grid.hip - grid(): segment -> data point reconstruction on gfx950.
//
// Replaces the per-row loop of GridStream (crates/modelardb_storage/src/query/grid_exec.rs:323-356)
// and modelardb_compression::grid (crates/modelardb_compression/src/models/mod.rs:190-251).
//
// Pipeline (all on the context's stream):
//   k_ts_sort, k_grid_ts_count  (batches with delta-of-delta timestamps) one lane per stream, longest streams
//                    first: counts the codes (len()), leaves a cursor in front of every 256-bit piece, and
//                    lists the jumps of a stream that is a fixed rate with the odd gap.
//   k_grid_prepass   1 thread / segment: len(), residual count, values-column decode, Swing
//                    slope/intercept (f64 divide once per segment, not per point) -> 48 B descriptor
//                    + per-segment point count; per-block totals.
//   k_scan_blocks    one block scans the per-block totals.
//   k_grid_offsets   block-wide wavefront prefix sums -> 64-bit output offsets, the compacted list of
//                    segments with serial work, rows-per-segment, and the tile -> first segment map.
//   k_grid_tiles     the HBM-store-bound kernel: one 4096-point output tile per workgroup, segment
//                    offsets and descriptors staged in LDS, each lane reconstructs 4 consecutive
//                    points; every wave-level store instruction writes 1 KiB contiguous (timestamps
//                    are transposed through LDS for that). Writes placeholders for points it cannot
//                    reconstruct. k_grid_tiles_jumps is the same for a batch with listed segments: a table
//                    of rows per tile, one per segment and per jump.
//   k_grid_timestamps 1 lane / 256-bit piece of a delta-of-delta timestamp stream, from the cursor the
//                    prepass left in front of the piece's first code: timestamps and the Swing values
//                    that follow from them overwrite the placeholders, each lane a contiguous run.
//   k_grid_serial    1 lane / segment with a serial dependency, always after k_grid_tiles: MacaqueV
//                    value streams, residual tails (<= 255 values) and the few irregular timestamp
//                    streams short enough to live inside their view overwrite the placeholders.
// Algorithmic bytes: 73 B/segment read + 12 B/point written (8 B timestamp + 4 B value).
#include "mdb_segment_dev.hpp"
#include "mdb_scan.hpp"
#include "mdb_macaque_parallel.hpp"

#include <atomic>
#include <cfloat>

namespace mdb {

constexpr int PREPASS_THREADS = 256;
constexpr int PREPASS_ITEMS = 4; // segments per thread in the scan kernels
constexpr int SEGS_PER_BLOCK = PREPASS_THREADS * PREPASS_ITEMS;
#ifndef MDB_TILE_POINTS
#define MDB_TILE_POINTS 4096
#endif
constexpr uint32_t TILE_POINTS = MDB_TILE_POINTS;
constexpr int TILE_THREADS = 256;
constexpr int TILE_LDS_SEGMENTS = 1024; // more segments than this in one tile -> global search
struct GridHeader {
    unsigned long long total_points;
    unsigned long long n_serial;
    unsigned int error;
    unsigned int pad;
    unsigned long long metrics[10]; // [0..8] mdb_grid_metrics minus rows_created (= total_points);
                                    // [9] bytes of the MacaqueV streams the parallel decoder takes
    unsigned long long checkpointed_points; // visible points of the segments k_grid_timestamps decodes
    unsigned long long checkpointed_pieces; // pieces of their streams
    unsigned long long jump_segments;       // segments whose timestamps k_grid_tiles takes from a jump list
    unsigned long long live_pieces;         // pieces listed in TsCheckpoints::live (k_grid_ts_count)
};

// MODE 0: every segment through the generic analysis (a time range is given).
// MODE 1: the simple segments (segment_is_simple) through the trimmed analysis; the others are left to a
//         MODE 2 launch behind it: one bit per segment in `pending`, one flag per block in `block_pending`.
// MODE 2: the segments MODE 1 left, through the generic analysis; a block without any returns at once.
template <int MODE>
__global__ __launch_bounds__(PREPASS_THREADS) void k_grid_prepass(
    DevSegments s, TimeRange range, uint32_t mv_min_values, TileDesc *__restrict__ desc,
    uint32_t *__restrict__ counts, uint32_t *__restrict__ irregular_totals,
    uint32_t *__restrict__ irregular_first, unsigned long long *__restrict__ block_points,
    unsigned long long *__restrict__ block_serial, GridHeader *__restrict__ header, TsCheckpoints checkpoints,
    const uint32_t *__restrict__ known_totals, unsigned long long *__restrict__ pending,
    uint32_t *__restrict__ block_pending, uint32_t n_blocks) {
    __shared__ unsigned long long lds_metrics[16];
    // MODE 2 is launched with few workgroups that walk the blocks of the MODE 1 launch: a batch of simple
    // segments only costs a look at its block flags (32 k workgroups that return at once cost 0.16 ms).
    for (uint32_t block = blockIdx.x; block < n_blocks; block += gridDim.x) {
    if (MODE == 2 && block_pending[block] == 0) continue;
    if (MODE == 2) __syncthreads();
    if (threadIdx.x < 16) lds_metrics[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)block * SEGS_PER_BLOCK;
    unsigned long long points = 0, serial = 0, checkpointed = 0, checkpointed_pieces = 0, with_jumps = 0;
    unsigned long long m[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t error = 0;
    bool left_any = false;
#pragma unroll 1
    for (int k = 0; k < PREPASS_ITEMS; k++) {
        uint64_t i = base + (uint64_t)k * PREPASS_THREADS + threadIdx.x;
        if (MODE == 1) {
            // The 64 segments of a wave's iteration are one word of `pending` (base and k * 256 are
            // multiples of 64).
            const bool simple = i < s.n && segment_is_simple(s, i);
            const unsigned long long left = __ballot(i < s.n && !simple);
            if ((threadIdx.x & (MDB_WAVE - 1)) == 0 && (base + (uint64_t)k * PREPASS_THREADS + (threadIdx.x & ~(MDB_WAVE - 1))) < s.n)
                pending[i >> 6] = left;
            left_any |= left != 0;
            if (!simple) continue;
        } else {
            if (i >= s.n) break;
            if (MODE == 2 && !((pending[i >> 6] >> (i & 63)) & 1ull)) continue;
        }
        SegInfo info = MODE == 1 ? analyse_segment<ANALYSE_SIMPLE>(s, i) : analyse_segment<ANALYSE_GENERIC>(s, i, known_totals, &checkpoints);
        if (MODE != 1 && range.enabled) apply_time_range(s, i, info, range, nullptr, nullptr, &checkpoints);
        uint32_t jump_base = 0;
        if (MODE != 1 && !range.enabled && (info.desc.flags & FLAG_CHECKPOINTS) && checkpoints.jumps) {
            // A fixed rate with a few jumps (k_grid_ts_count has listed them): k_grid_tiles' work.
            const unsigned long long first_piece = checkpoints.piece_base[i];
            const TsJump list = checkpoints.jumps[first_piece * TS_JUMPS_PER_PIECE];
            if (list.count != TS_NO_JUMPS) {
                info.desc.flags = (info.desc.flags & ~FLAG_CHECKPOINTS) | FLAG_JUMPS | (list.count << FLAG_JUMP_COUNT_SHIFT);
                info.desc.delta = list.value;
                jump_base = (uint32_t)first_piece;
                with_jumps += 1;
            }
        }
        error |= info.error;
        const SegDesc &d = info.desc;
        TileDesc tile_desc = make_tile_desc(d);
        if (d.flags & FLAG_JUMPS) set_jump_list(tile_desc, jump_base);
        desc[i] = tile_desc;
        bool is_serial = (d.flags & FLAG_SERIAL) != 0;
        counts[i] = d.n_visible | (is_serial ? SERIAL_BIT : 0u);
        if (MODE != 1 && !(d.flags & FLAG_REGULAR)) { // k_grid_serial need not parse the timestamps to count again
            irregular_totals[i] = d.n_total;
            irregular_first[i] = d.first;
        }
        points += d.n_visible;
        serial += is_serial ? 1 : 0;
        uint32_t type = d.flags & FLAG_TYPE_MASK;
        if (type < 3) {
            m[type] += d.n_visible; // rows_created_by_model_type
            m[4 + type] += 1;       // segments_with_model_type
        }
        m[3] += (d.flags & FLAG_HAS_RESIDUALS) ? 1 : 0;
        m[7] += (d.flags & FLAG_REGULAR) ? 1 : 0;
        m[8] += (d.flags & FLAG_REGULAR) ? 0 : 1;
        if (MODE != 1 && (d.flags & FLAG_CHECKPOINTS)) {
            checkpointed += d.n_visible;
            if (checkpoints.piece_base) checkpointed_pieces += checkpoints.piece_base[i + 1] - checkpoints.piece_base[i];
        }
        // Bytes of the MacaqueV streams long enough for the parallel decoder (bounds its scratch).
        if (MODE != 1 && type == MDB_MACAQUE_V_ID && mv_min_values != 0xffffffffu) {
            const uint32_t bytes = s.values.views[i].x;
            if (mv_qualifies(info, bytes, mv_min_values)) m[9] += bytes;
        }
    }
    // Block totals through LDS atomics (few per thread, once per block).
    atomicAdd(&lds_metrics[10], points);
    atomicAdd(&lds_metrics[11], serial);
    if (checkpointed) atomicAdd(&lds_metrics[13], checkpointed);
    if (checkpointed_pieces) atomicAdd(&lds_metrics[15], checkpointed_pieces);
    if (with_jumps) atomicAdd(&lds_metrics[14], with_jumps);
    if (MODE == 1 && left_any && (threadIdx.x & (MDB_WAVE - 1)) == 0) atomicAdd(&lds_metrics[12], 1ull);
#pragma unroll
    for (int k = 0; k < 10; k++)
        if (m[k]) atomicAdd(&lds_metrics[k], m[k]);
    if (error) atomicOr(&header->error, error);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (MODE == 2) {
            block_points[block] += lds_metrics[10];
            block_serial[block] += lds_metrics[11];
        } else {
            block_points[block] = lds_metrics[10];
            block_serial[block] = lds_metrics[11];
        }
        if (MODE == 1) block_pending[block] = (uint32_t)lds_metrics[12];
    }
    if (threadIdx.x < 10 && lds_metrics[threadIdx.x])
        atomicAdd(&header->metrics[threadIdx.x], lds_metrics[threadIdx.x]);
    if (threadIdx.x == 13 && lds_metrics[13]) atomicAdd(&header->checkpointed_points, lds_metrics[13]);
    if (threadIdx.x == 14 && lds_metrics[14]) atomicAdd(&header->jump_segments, lds_metrics[14]);
    if (threadIdx.x == 15 && lds_metrics[15]) atomicAdd(&header->checkpointed_pieces, lds_metrics[15]);
    }
}

// One block: exclusive scan of the per-block totals (in place), totals into the header.
__global__ __launch_bounds__(1024) void k_scan_blocks(unsigned long long *__restrict__ block_points,
                                                      unsigned long long *__restrict__ block_serial,
                                                      uint32_t n_blocks, GridHeader *__restrict__ header) {
    __shared__ uint64_t lds[17];
    uint64_t carry_points = 0, carry_serial = 0;
    for (uint32_t base = 0; base < n_blocks; base += 1024) {
        uint32_t i = base + threadIdx.x;
        uint64_t p = i < n_blocks ? block_points[i] : 0;
        uint64_t q = i < n_blocks ? block_serial[i] : 0;
        uint64_t total_p, total_q;
        uint64_t ep = block_exclusive_scan_u64(p, lds, &total_p);
        uint64_t eq = block_exclusive_scan_u64(q, lds, &total_q);
        if (i < n_blocks) {
            block_points[i] = carry_points + ep;
            block_serial[i] = carry_serial + eq;
        }
        carry_points += total_p;
        carry_serial += total_q;
    }
    if (threadIdx.x == 0) {
        header->total_points = carry_points;
        header->n_serial = carry_serial;
    }
}

__global__ __launch_bounds__(PREPASS_THREADS) void k_grid_offsets(
    const uint32_t *__restrict__ counts, uint64_t n, const unsigned long long *__restrict__ block_points,
    const unsigned long long *__restrict__ block_serial, unsigned long long *__restrict__ offsets,
    uint32_t *__restrict__ serial_ids, uint32_t *__restrict__ tile_first,
    uint32_t *__restrict__ rows_per_segment) {
    __shared__ uint64_t lds[17];
    const uint64_t first = (uint64_t)blockIdx.x * SEGS_PER_BLOCK + (uint64_t)threadIdx.x * PREPASS_ITEMS;
    uint32_t c[PREPASS_ITEMS];
    if (first + PREPASS_ITEMS <= n) {
        uint4 v = *reinterpret_cast<const uint4 *>(counts + first);
        c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w;
    } else {
#pragma unroll
        for (int k = 0; k < PREPASS_ITEMS; k++) c[k] = (first + k < n) ? counts[first + k] : 0u;
    }
    uint64_t local_points = 0, local_serial = 0;
#pragma unroll
    for (int k = 0; k < PREPASS_ITEMS; k++) {
        local_points += c[k] & COUNT_MASK;
        local_serial += c[k] >> 31;
    }
    uint64_t total;
    uint64_t point_offset = block_points[blockIdx.x] + block_exclusive_scan_u64(local_points, lds, &total);
    uint64_t serial_offset = block_serial[blockIdx.x] + block_exclusive_scan_u64(local_serial, lds, &total);
#pragma unroll
    for (int k = 0; k < PREPASS_ITEMS; k++) {
        uint64_t i = first + k;
        if (i >= n) break;
        uint32_t count = c[k] & COUNT_MASK;
        offsets[i] = point_offset;
        if (rows_per_segment) rows_per_segment[i] = count;
        if (c[k] >> 31) serial_ids[serial_offset++] = (uint32_t)i;
        // This segment owns every tile whose first point lies inside it.
        uint64_t end = point_offset + count;
        for (uint64_t t = (point_offset + TILE_POINTS - 1) / TILE_POINTS; t * TILE_POINTS < end; t++)
            tile_first[t] = (uint32_t)i;
        point_offset = end;
        if (i == n - 1) offsets[n] = end;
    }
}

// ---- the tile kernel -------------------------------------------------------------------------------
//
// One 4096-point output tile per workgroup. Each lane computes 4 consecutive points per iteration.
// Values leave as one 16-byte store per lane (1 KiB contiguous per wave instruction). Timestamps
// (32 bytes per lane) are transposed through a wave-private LDS slab so that each of the two
// timestamp store instructions also writes 1 KiB contiguous instead of 16 bytes at a 32-byte
// stride (measured +9 % at 740 points/segment, +18 % at 100; scripts/micro/tiles_ablate.hip).
//
// The kernel writes EVERY point of its tile. Points it cannot reconstruct (MacaqueV values,
// residual values, everything of an irregular segment) get placeholders that k_grid_serial, which
// always runs after it on the same stream, overwrites. That keeps every store a full-width vector
// store; the price is 4 (or 12) extra bytes for exactly those points.

struct PointValue {
    int64_t t;
    float v;
};

// `index` counts from the first VISIBLE point of the segment (TileDesc::start is that point).
__device__ __forceinline__ PointValue reconstruct_point(const TileDesc &d, uint32_t index) {
    PointValue out;
    out.t = d.start + (int64_t)((uint64_t)index * (uint64_t)d.delta);
    const uint32_t type = d.flags & FLAG_TYPE_MASK;
    out.v = type == MDB_SWING_ID ? (float)(d.slope * (double)out.t + d.intercept) : d.value;
    return out;
}

constexpr int TILE_LDS_DESCS = 128; // descriptors of the first segments of a tile staged in LDS
constexpr uint32_t NO_JUMP_LIST = 0xffffffffu;
constexpr uint32_t TILE_ROWS = 256; // (k_grid_tiles_jumps) segments plus jumps of a tile that is done by its table of rows

// JUMPS: some segment of the batch has a jump list (GridHeader::jump_segments; TsJump says what that is); the
// flavour without is what a batch of regular timestamps runs.
template <bool JUMPS>
__device__ __forceinline__ void grid_tile(
    const TileDesc *__restrict__ desc, const unsigned long long *__restrict__ offsets,
    const uint32_t *__restrict__ tile_first, uint64_t n_segments, uint64_t total_points,
    uint64_t n_tiles, int64_t *__restrict__ out_ts, float *__restrict__ out_val, const TsJump *__restrict__ jumps) {
    __shared__ __attribute__((aligned(16))) uint32_t rel[TILE_LDS_SEGMENTS + 1]; // rel[k] = offsets[s0 + k] - tile_start, k >= 1
    __shared__ __attribute__((aligned(16))) TileDesc lds_desc[TILE_LDS_DESCS];
    __shared__ __attribute__((aligned(16))) longlong2 ts_slab[TILE_THREADS / MDB_WAVE][2 * MDB_WAVE];
    const uint64_t tile = blockIdx.x;
    const uint64_t tile_start = tile * TILE_POINTS;
    const uint64_t tile_end = min(total_points, tile_start + TILE_POINTS);
    const uint32_t s0 = tile_first[tile];
    const uint32_t s1 = (tile + 1 < n_tiles) ? tile_first[tile + 1] : (uint32_t)(n_segments - 1);
    uint32_t n_in_tile = s1 - s0 + 1;
    bool use_lds = n_in_tile <= TILE_LDS_SEGMENTS;
    const uint64_t s0_offset = offsets[s0];
    if (use_lds)
        for (uint32_t k = 1 + threadIdx.x; k < n_in_tile; k += TILE_THREADS)
            rel[k] = (uint32_t)(offsets[s0 + k] - tile_start);
    // A tile with a run of segments WITHOUT a visible point in its middle (a query over a time range: the segments
    // behind the range of one series and in front of the range of the next, hundreds of them between the two that
    // the tile's points belong to): the segments that do have points in the tile, a handful, are put in front of
    // the table, and the tile is an ordinary one - its descriptors in LDS, a few steps per search - instead of one
    // that reads descriptors from memory point by point. (thin_segment[k]: which segment of the tile entry k is.)
    // (both live where the timestamps are transposed later: the barrier behind the descriptors lies in between)
    uint32_t *kept_rel = reinterpret_cast<uint32_t *>(&ts_slab[0][0]);
    uint32_t *thin_segment = kept_rel + TILE_LDS_DESCS;
    __shared__ uint32_t n_kept;
    bool thinned = false;
    if (!JUMPS && n_in_tile > (uint32_t)TILE_LDS_DESCS) {
        if (threadIdx.x < MDB_WAVE) {
            // (the offsets from memory, 64 at a time: the run may be longer than `rel`)
            const uint32_t tile_points = (uint32_t)(tile_end - tile_start);
            uint32_t running = 0;
            for (uint32_t base = 0; base < n_in_tile && running <= (uint32_t)TILE_LDS_DESCS; base += MDB_WAVE) {
                const uint32_t k = base + threadIdx.x;
                bool keep = false;
                uint32_t begins = 0;
                if (k < n_in_tile) {
                    begins = k == 0 ? 0u : (uint32_t)min(offsets[s0 + k] - tile_start, (unsigned long long)tile_points);
                    const uint32_t ends = k + 1 < n_in_tile ? (uint32_t)min(offsets[s0 + k + 1] - tile_start, (unsigned long long)tile_points)
                                                            : tile_points; // (the last one reaches the tile's end)
                    keep = k == 0 || (begins < tile_points && ends > begins);
                }
                const unsigned long long keepers = __ballot(keep);
                const uint32_t at = running + (uint32_t)__popcll(keepers & ((1ull << threadIdx.x) - 1ull));
                if (keep && at < (uint32_t)TILE_LDS_DESCS) {
                    thin_segment[at] = k;
                    kept_rel[at] = begins;
                }
                running += (uint32_t)__popcll(keepers);
            }
            if (threadIdx.x == 0) n_kept = running;
        }
        __syncthreads();
        thinned = n_kept <= (uint32_t)TILE_LDS_DESCS;
        if (thinned) {
            n_in_tile = n_kept;
            use_lds = true;
            for (uint32_t k = 1 + threadIdx.x; k < n_in_tile; k += TILE_THREADS) rel[k] = kept_rel[k];
        }
    }
    // A tile all of whose segments have their timestamps decoded by k_grid_timestamps is written there,
    // values included: nothing to do here.
    bool mine_are_left = n_in_tile <= (uint32_t)TILE_LDS_DESCS;
    for (uint32_t k = threadIdx.x; k < min(n_in_tile, (uint32_t)TILE_LDS_DESCS); k += TILE_THREADS) {
        const TileDesc d = desc[s0 + (thinned ? thin_segment[k] : k)];
        lds_desc[k] = d;
        mine_are_left = mine_are_left && (d.flags & FLAG_CHECKPOINTS) != 0;
    }
    if (__syncthreads_and(mine_are_left)) return;
    auto descriptor = [&](uint32_t k) -> TileDesc { // k relative to s0
        return k < TILE_LDS_DESCS ? lds_desc[k] : desc[s0 + k];
    };
    const int lane = threadIdx.x & (MDB_WAVE - 1);
    const int wave = threadIdx.x / MDB_WAVE;

    // (JUMPS) The rows of the tile: a segment with a jump list is, between two jumps, a segment with regular
    // timestamps that are all late by what the jumps so far add up to. So the table the points are looked up in
    // gets a row per segment AND per jump - where it begins in the tile, which segment it belongs to, how late
    // it is - built once per tile, the jumps read from memory by all threads at once, and a point costs what
    // it costs in a batch of regular timestamps plus a few steps of the search. (Looking the jumps up where the
    // points are computed, even by the whole wave together as below, doubles the instructions per point: 3.5
    // instead of 2.0 ms per 10^9 points.) A tile with more rows than the table holds (a list of hundreds of
    // jumps: the whole list is taken, not only the jumps inside the tile), or with more segments than have their
    // descriptors in LDS, is done the other way.
    // (The table takes no LDS of its own, which would cost the kernel an eighth of its waves: a tile done this
    // way has at most TILE_LDS_DESCS segments, and the rest of `rel` is room enough; what is only needed while
    // the table is built lies where the timestamps are transposed later.)
    constexpr uint32_t ROWS_AT = (TILE_LDS_DESCS + 4) & ~3u;
    static_assert(ROWS_AT + TILE_ROWS + 2 * TILE_ROWS + TILE_ROWS / 4 <= TILE_LDS_SEGMENTS + 1, "the rows fit behind rel[TILE_LDS_DESCS]");
    uint32_t *row_rel = rel + ROWS_AT;
    long long *row_late = reinterpret_cast<long long *>(rel + ROWS_AT + TILE_ROWS);
    uint8_t *row_seg = reinterpret_cast<uint8_t *>(rel + ROWS_AT + 3 * TILE_ROWS);
    uint32_t *row_start = reinterpret_cast<uint32_t *>(&ts_slab[0][0]); // the first row of segment k
    uint32_t *row_totals = row_start + TILE_LDS_DESCS + 1;
    uint32_t n_rows = 0; // (0: no table)
    if (JUMPS && n_in_tile <= (uint32_t)TILE_LDS_DESCS) {
        uint32_t mine = 0; // rows of segment threadIdx.x
        if (threadIdx.x < n_in_tile) {
            const uint32_t flags = lds_desc[threadIdx.x].flags;
            mine = 1u + ((flags & FLAG_JUMPS) ? flags >> FLAG_JUMP_COUNT_SHIFT : 0u);
        }
        uint32_t inclusive = mine;
#pragma unroll
        for (int delta = 1; delta < MDB_WAVE; delta <<= 1) {
            const uint32_t up = __shfl_up(inclusive, delta, MDB_WAVE);
            if (lane >= delta) inclusive += up;
        }
        if (lane == MDB_WAVE - 1) row_totals[wave] = inclusive;
        __syncthreads();
        uint32_t in_front = 0, total = 0;
#pragma unroll
        for (int w = 0; w < TILE_THREADS / MDB_WAVE; w++) {
            in_front += w < wave ? row_totals[w] : 0u;
            total += row_totals[w];
        }
        if (threadIdx.x < n_in_tile) row_start[threadIdx.x] = in_front + inclusive - mine;
        __syncthreads();
        if (total <= TILE_ROWS) {
            n_rows = total;
            const uint32_t first_index = (uint32_t)(tile_start - s0_offset); // of segment 0's points, the first in the tile
            for (uint32_t row = threadIdx.x; row < total; row += TILE_THREADS) {
                uint32_t k = 0, hi = n_in_tile; // the last segment whose first row is at or before `row`
                while (hi - k > 1) {
                    const uint32_t mid = (k + hi) >> 1;
                    if (row_start[mid] <= row) k = mid; else hi = mid;
                }
                const uint32_t e = row - row_start[k];
                uint32_t begins = k == 0 ? 0u : rel[k];
                long long late = 0;
                if (e > 0) {
                    const TsJump jump = jumps[(uint64_t)jump_list_of(lds_desc[k]) * TS_JUMPS_PER_PIECE + e];
                    // (a jump of segment 0 in front of the tile: at the tile's first point, where the search
                    // takes the last of the rows that begin at the same point)
                    const uint32_t from = k == 0 ? first_index : 0u;
                    begins += jump.position > from ? jump.position - from : 0u;
                    late = jump.value;
                }
                row_rel[row] = begins;
                row_seg[row] = (uint8_t)k;
                row_late[row] = late;
            }
        }
        __syncthreads();
    }

    // (JUMPS) the entries of a jump list this lane has read last: entry kept_first + lane of segment kept_segment
    TsJump kept = TsJump{0xffffffffu, 0u, 0};
    uint32_t kept_segment = NO_JUMP_LIST, kept_first = 0;
#pragma unroll 1
    for (uint32_t j = 0; j < TILE_POINTS / (TILE_THREADS * 4); j++) {
        // The wave's 256 points of this iteration start at wave_base (wave-uniform).
        // (JUMPS, a tile without the table of rows: the four rounds of a wave follow each other in the output, so
        // that they mostly stay inside one segment and the part of its jump list the wave has read serves the
        // next round too)
        const uint64_t wave_base = JUMPS && n_rows == 0 ? tile_start + (uint64_t)wave * (TILE_POINTS / (TILE_THREADS / MDB_WAVE)) + (uint64_t)j * (MDB_WAVE * 4)
                                         : tile_start + (uint64_t)j * (TILE_THREADS * 4) + (uint64_t)wave * (MDB_WAVE * 4);
        if (wave_base >= tile_end) {
            if (JUMPS) continue;
            break;
        }
        const uint64_t p = wave_base + (uint64_t)lane * 4;
        int64_t t[4] = {0, 0, 0, 0};
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        // Points of a segment whose timestamps k_grid_timestamps decodes are written by it, values included:
        // where all 256 points of the wave's iteration are such points nothing is stored here.
        bool left_to_timestamps = p >= tile_end;
        // (JUMPS) for each of the lane's points: its segment (relative to s0) if that has a jump list, and its
        // index in the segment.
        uint32_t list_of[4] = {NO_JUMP_LIST, NO_JUMP_LIST, NO_JUMP_LIST, NO_JUMP_LIST};
        uint32_t index_of[4] = {0, 0, 0, 0};
        if (JUMPS && n_rows > 0) {
            if (p < tile_end) {
                const uint32_t local = (uint32_t)(p - tile_start);
                uint32_t r = 0, hi = n_rows; // the last row that begins at or before the point
                while (hi - r > 1) {
                    const uint32_t mid = (r + hi) >> 1;
                    if (row_rel[mid] <= local) r = mid; else hi = mid;
                }
                uint32_t segment = row_seg[r];
                int64_t late = row_late[r];
                uint64_t segment_offset = segment == 0 ? s0_offset : tile_start + rel[segment];
                TileDesc d = lds_desc[segment];
                const uint32_t index = (uint32_t)(p - segment_offset);
                const uint32_t next_begins = r + 1 < n_rows ? row_rel[r + 1] : 0xffffffffu;
                // All four points in one row, the common case - or in two rows of one segment, which is what a
                // jump looks like to the lane it falls into (with a jump per hundred points every other round of
                // a wave has such a lane, and the way out below costs the whole wave more than the round).
                bool together = index + 4 <= d.n_points;
                uint32_t split = 4; // the first of the four points that lies in the second row
                int64_t late_then = late;
                if (together && local + 4 > next_begins) {
                    const uint32_t after_begins = r + 2 < n_rows ? row_rel[r + 2] : 0xffffffffu;
                    together = row_seg[r + 1] == segment && local + 4 <= after_begins;
                    split = next_begins - local;
                    late_then = row_late[r + 1];
                }
                if (together) {
                    left_to_timestamps = (d.flags & FLAG_CHECKPOINTS) != 0;
                    const int64_t regular = reconstruct_point(d, index).t;
                    t[0] = regular + late;
                    t[1] = regular + d.delta + (split <= 1 ? late_then : late);
                    t[2] = regular + 2 * d.delta + (split <= 2 ? late_then : late);
                    t[3] = regular + 3 * d.delta + (split <= 3 ? late_then : late);
                    if ((d.flags & FLAG_TYPE_MASK) == MDB_SWING_ID) {
                        v.x = (float)(d.slope * (double)t[0] + d.intercept);
                        v.y = (float)(d.slope * (double)t[1] + d.intercept);
                        v.z = (float)(d.slope * (double)t[2] + d.intercept);
                        v.w = (float)(d.slope * (double)t[3] + d.intercept);
                    } else {
                        v = make_float4(d.value, d.value, d.value, d.value);
                    }
                } else {
                    // The group straddles a segment boundary (or two jumps): row by row.
                    float values[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 1
                    for (uint32_t k = 0; k < 4; k++) {
                        const uint64_t q = p + k;
                        if (q >= tile_end) break;
                        bool moved = false;
                        while (r + 1 < n_rows && row_rel[r + 1] <= local + k) {
                            r += 1;
                            moved = true;
                        }
                        if (moved) {
                            late = row_late[r];
                            if (row_seg[r] != segment) {
                                segment = row_seg[r];
                                segment_offset = tile_start + rel[segment]; // (not segment 0: that one comes first)
                                d = lds_desc[segment];
                            }
                        }
                        const int64_t timestamp = reconstruct_point(d, (uint32_t)(q - segment_offset)).t + late;
                        t[k] = timestamp;
                        values[k] = (d.flags & FLAG_TYPE_MASK) == MDB_SWING_ID ? (float)(d.slope * (double)timestamp + d.intercept)
                                                                               : d.value;
                    }
                    v = make_float4(values[0], values[1], values[2], values[3]);
                }
            }
        } else if (p < tile_end) {
            const uint32_t local = (uint32_t)(p - tile_start);
            // Largest k in [0, n_in_tile) with offsets[s0 + k] <= p.
            uint32_t lo = 0, hi = n_in_tile;
            if (use_lds) {
                while (hi - lo > 1) {
                    uint32_t mid = (lo + hi) >> 1;
                    if (rel[mid] <= local) lo = mid; else hi = mid;
                }
            } else {
                while (hi - lo > 1) {
                    uint32_t mid = (lo + hi) >> 1;
                    if (offsets[s0 + mid] <= p) lo = mid; else hi = mid;
                }
            }
            uint64_t segment_offset =
                lo == 0 ? s0_offset : (use_lds ? tile_start + rel[lo] : (uint64_t)offsets[s0 + lo]);
            TileDesc d = descriptor(lo);
            const uint32_t index = (uint32_t)(p - segment_offset);
            if (index + 4 <= d.n_points) {
                // All four points in one segment: the common case.
                left_to_timestamps = (d.flags & FLAG_CHECKPOINTS) != 0;
                PointValue q0 = reconstruct_point(d, index);
                t[0] = q0.t; t[1] = q0.t + d.delta; t[2] = t[1] + d.delta; t[3] = t[2] + d.delta;
                if (JUMPS && (d.flags & FLAG_JUMPS)) {
                    // (the timestamps are finished, and the values computed from them, below)
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) {
                        list_of[k] = lo;
                        index_of[k] = index + k;
                    }
                } else if ((d.flags & FLAG_TYPE_MASK) == MDB_SWING_ID) {
                    v.x = q0.v;
                    v.y = (float)(d.slope * (double)t[1] + d.intercept);
                    v.z = (float)(d.slope * (double)t[2] + d.intercept);
                    v.w = (float)(d.slope * (double)t[3] + d.intercept);
                } else {
                    v = make_float4(d.value, d.value, d.value, d.value);
                }
            } else {
                // The group straddles a segment boundary: walk point by point.
                uint64_t next_offset = segment_offset + d.n_points;
                float values[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 1
                for (uint32_t k = 0; k < 4; k++) {
                    const uint64_t q = p + k;
                    if (q >= tile_end) break;
                    while (q >= next_offset) {
                        lo += 1;
                        segment_offset = next_offset;
                        d = descriptor(lo);
                        next_offset = segment_offset + d.n_points;
                    }
                    PointValue point = reconstruct_point(d, (uint32_t)(q - segment_offset));
                    t[k] = point.t;
                    values[k] = point.v;
                    if (JUMPS && (d.flags & FLAG_JUMPS)) {
                        // (k is not a compile-time index here: the arrays stay in registers this way)
                        const uint32_t index = (uint32_t)(q - segment_offset);
                        if (k == 0) { list_of[0] = lo; index_of[0] = index; }
                        if (k == 1) { list_of[1] = lo; index_of[1] = index; }
                        if (k == 2) { list_of[2] = lo; index_of[2] = index; }
                        if (k == 3) { list_of[3] = lo; index_of[3] = index; }
                    }
                }
                v = make_float4(values[0], values[1], values[2], values[3]);
            }
        }
        if (JUMPS && n_rows == 0) {
            // (A tile without the table of rows.)
            // The points of the wave that lie in segments with a jump list, segment by segment (one or two per
            // wave, unless the segments are short): they follow each other in the segment as they do in the
            // wave, so the wave reads the part of the list it needs together, 64 entries per load, and every
            // jump that lies among its points is applied by all lanes at once. (Every lane searching the list
            // of its segment by itself is seven dependent loads per lane and four times that for a lane whose
            // points lie in two segments: 4.8 ms per 10^9 points against 2.0 for regular timestamps.)
            uint32_t next_segment = 0; // lists of segments below this one have been applied
            for (;;) {
                uint32_t lowest = NO_JUMP_LIST; // the lane's first segment with a list that is still to do
#pragma unroll
                for (int k = 3; k >= 0; k--)
                    if (list_of[k] != NO_JUMP_LIST && list_of[k] >= next_segment) lowest = list_of[k];
                const unsigned long long waiting = __ballot(lowest != NO_JUMP_LIST);
                if (!waiting) break;
                // (segments ascend with the points: the first waiting lane has the lowest one)
                const uint32_t segment = (uint32_t)__builtin_amdgcn_readlane((int)lowest, __ffsll((long long)waiting) - 1);
                next_segment = segment + 1;
                bool in_it[4];
                uint32_t my_first = 0xffffffffu, my_last = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    in_it[k] = list_of[k] == segment;
                    if (in_it[k]) {
                        my_first = min(my_first, index_of[k]);
                        my_last = max(my_last, index_of[k]);
                    }
                }
                const unsigned long long group = __ballot(my_first != 0xffffffffu);
                const uint32_t index_first = (uint32_t)__builtin_amdgcn_readlane((int)my_first, __ffsll((long long)group) - 1);
                const uint32_t index_end = (uint32_t)__builtin_amdgcn_readlane((int)my_last, 63 - __clzll((long long)group)) + 1u;
                const TileDesc listed = descriptor(segment); // (the same for every lane)
                const uint32_t count = listed.flags >> FLAG_JUMP_COUNT_SHIFT;
                const TsJump *list = jumps + (uint64_t)jump_list_of(listed) * TS_JUMPS_PER_PIECE + 1;
                // How many jumps lie before index_first: narrowed down to a range of at most 64 entries by
                // probes of the whole wave (none for a list of up to 64 entries).
                uint32_t e_lo = 0, e_hi = count;
                while (e_hi - e_lo > (uint32_t)MDB_WAVE) {
                    const uint32_t stride = (e_hi - e_lo + MDB_WAVE - 1) / MDB_WAVE;
                    const uint32_t e = e_lo + (uint32_t)lane * stride;
                    const bool before = e < e_hi && list[e].position < index_first;
                    const uint32_t c = (uint32_t)__popcll(__ballot(before)); // (the probes ascend: a prefix of the lanes)
                    const uint32_t probe_end = e_lo + c * stride;
                    e_lo = c ? probe_end - stride + 1u : e_lo;
                    e_hi = min(e_hi, probe_end);
                }
                e_lo -= e_lo ? 1u : 0u; // (the last jump before index_first may be the entry in front of that range)
                int64_t applied = 0;    // what the jumps applied so far add up to (the last one's value)
                for (;;) {
                    const uint32_t e = e_lo + (uint32_t)lane;
                    TsJump entry = kept;
                    if (kept_segment != segment || kept_first != e_lo) {
                        entry = TsJump{0xffffffffu, 0u, 0};
                        if (e < count) entry = list[e];
                        kept = entry;
                        kept_segment = segment;
                        kept_first = e_lo;
                    }
                    const unsigned long long before = __ballot(entry.position < index_first);
                    unsigned long long inside = __ballot(entry.position >= index_first && entry.position < index_end);
                    auto value_of = [&](int from) {
                        const uint32_t high = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)entry.value >> 32), from);
                        const uint32_t low = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)entry.value, from);
                        return (int64_t)(((uint64_t)high << 32) | low);
                    };
                    if (before) {
                        const int64_t value = value_of(63 - __clzll((long long)before));
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            if (in_it[k]) t[k] += value - applied;
                        applied = value;
                    }
                    while (inside) {
                        const int from = __ffsll((long long)inside) - 1;
                        inside &= inside - 1;
                        const uint32_t position = (uint32_t)__builtin_amdgcn_readlane((int)entry.position, from);
                        const int64_t value = value_of(from);
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            if (in_it[k] && index_of[k] >= position) t[k] += value - applied;
                        applied = value;
                    }
                    // (more of the list may lie among the wave's points: the entries behind this load's last one)
                    const uint32_t last_position = (uint32_t)__builtin_amdgcn_readlane((int)entry.position, MDB_WAVE - 1);
                    if (e_lo + MDB_WAVE >= count || last_position >= index_end) break;
                    e_lo += MDB_WAVE;
                }
            }
            // Swing values of these points are values of their timestamps (swing.rs:304-319).
            if (list_of[0] != NO_JUMP_LIST || list_of[1] != NO_JUMP_LIST || list_of[2] != NO_JUMP_LIST || list_of[3] != NO_JUMP_LIST) {
                uint32_t have = NO_JUMP_LIST;
                TileDesc listed = TileDesc{};
                float values[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (list_of[k] == NO_JUMP_LIST) continue;
                    if (list_of[k] != have) {
                        have = list_of[k];
                        listed = descriptor(have);
                    }
                    values[k] = (listed.flags & FLAG_TYPE_MASK) == MDB_SWING_ID
                                    ? (float)(listed.slope * (double)t[k] + listed.intercept)
                                    : ((listed.flags & FLAG_TYPE_MASK) == MDB_PMC_MEAN_ID ? listed.value : 0.0f);
                }
                v = make_float4(values[0], values[1], values[2], values[3]);
            }
        }
        if (__all(left_to_timestamps)) continue; // (the same for the whole wave)
        if (p < tile_end) {
            if (p + 4 <= tile_end) {
                *reinterpret_cast<float4 *>(out_val + p) = v;
            } else {
                if (p + 0 < tile_end) out_val[p + 0] = v.x;
                if (p + 1 < tile_end) out_val[p + 1] = v.y;
                if (p + 2 < tile_end) out_val[p + 2] = v.z;
            }
        }
        // A join of several field columns reconstructs the shared timestamps only once (N3).
        if (out_ts == nullptr) continue;
        // Transpose the timestamps through LDS: lane l holds points 4l..4l+3 (chunks 2l, 2l+1 of
        // 16 bytes); store instruction A writes chunks 0..63, B writes chunks 64..127.
        longlong2 *slab = ts_slab[wave];
        slab[2 * lane] = make_longlong2(t[0], t[1]);
        slab[2 * lane + 1] = make_longlong2(t[2], t[3]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const longlong2 chunk_a = slab[lane];
        const longlong2 chunk_b = slab[MDB_WAVE + lane];
        __builtin_amdgcn_wave_barrier();
        const uint64_t point_a = wave_base + 2 * (uint64_t)lane;
        const uint64_t point_b = point_a + 2 * MDB_WAVE;
        longlong2 *ts_out = reinterpret_cast<longlong2 *>(out_ts + wave_base);
        if (point_a + 2 <= tile_end) ts_out[lane] = chunk_a;
        else if (point_a < tile_end) out_ts[point_a] = chunk_a.x;
        if (point_b + 2 <= tile_end) ts_out[MDB_WAVE + lane] = chunk_b;
        else if (point_b < tile_end) out_ts[point_b] = chunk_b.x;
    }
}

__global__ __launch_bounds__(TILE_THREADS) void k_grid_tiles(
    const TileDesc *__restrict__ desc, const unsigned long long *__restrict__ offsets,
    const uint32_t *__restrict__ tile_first, uint64_t n_segments, uint64_t total_points,
    uint64_t n_tiles, int64_t *__restrict__ out_ts, float *__restrict__ out_val) {
    grid_tile<false>(desc, offsets, tile_first, n_segments, total_points, n_tiles, out_ts, out_val, nullptr);
}

__global__ __launch_bounds__(TILE_THREADS, 8) void k_grid_tiles_jumps(
    const TileDesc *__restrict__ desc, const unsigned long long *__restrict__ offsets,
    const uint32_t *__restrict__ tile_first, uint64_t n_segments, uint64_t total_points,
    uint64_t n_tiles, int64_t *__restrict__ out_ts, float *__restrict__ out_val, const TsJump *__restrict__ jumps) {
    grid_tile<true>(desc, offsets, tile_first, n_segments, total_points, n_tiles, out_ts, out_val, jumps);
}

// ---- short segments: one pass over the raw rows ------------------------------------------------------------
//
// With 8 points per segment the pipeline above moves 193 bytes per segment around its 96 bytes of output - the
// prepass reads the 73-byte row and writes a 48-byte descriptor and a count, the offsets pass turns the count into
// an offset, the tile kernel reads descriptor and offset - and runs at a third of the HBM peak against the
// algorithmic bytes. A batch of SIMPLE segments (PMC-Mean / Swing, regular timestamps, no residuals: nothing any
// other kernel needs a descriptor for) can do without all of that: a workgroup takes 256 consecutive segments,
// analyses them from the raw rows into LDS, learns where its points begin from the workgroups before it
// (a decoupled look-back over one 64-bit word per workgroup: 2 flag bits and a running total) and writes them,
// 73 + 12 L bytes per segment in all. The first segment that is not simple raises a flag and the call falls back
// to the general pipeline, which overwrites whatever this pass wrote.

constexpr int FUSED_THREADS = 256;
// (a workgroup takes FUSED_ROUNDS x 256 segments: fewer, larger links in the look-back)
constexpr unsigned long long FUSED_FLAG_AGGREGATE = 1ull << 62, FUSED_FLAG_PREFIX = 2ull << 62, FUSED_VALUE_MASK = (1ull << 62) - 1;

// The points of a segment with regular timestamps (decompress_all_timestamps' count, timestamps.rs:163-223), from
// its timestamps view, start and end time alone; 0 where analyse_segment finds an error.
__device__ __forceinline__ uint32_t regular_segment_points(const DevSegments &s, uint64_t i) {
    const uint4 vt = s.timestamps.views[i];
    const int32_t ts_len = (int32_t)vt.x;
    const int64_t start = s.start_time[i], end = s.end_time[i];
    if (ts_len == 0) return start == end ? 1u : 2u;
    if (ts_len < 0 || ts_len > 8) return 0;
    uint64_t length = 0;
    for (int32_t k = 0; k < ts_len; k++) length = (length << 8) | view_inline_byte(vt, k);
    if (length < 2 || end < start) return 0;
    const uint64_t span = (uint64_t)(end - start), interval = span / (length - 1);
    if (interval == 0) return 0;
    const uint64_t produced = span / interval + 1;
    return produced > COUNT_MASK ? 0u : (uint32_t)produced;
}

// desc / offsets / serial_ids: written for the segments with serial work only (MacaqueV values, residual tails),
// which the kernels behind this one decode (k_grid_mv_pieces or k_grid_serial); header->n_serial counts them.
template <int FUSED_ROUNDS>
__global__ __launch_bounds__(FUSED_THREADS) __attribute__((amdgpu_waves_per_eu(7))) void k_grid_fused(DevSegments s, unsigned long long *__restrict__ lookback,
                                                              GridHeader *__restrict__ header, int64_t *__restrict__ out_ts,
                                                              float *__restrict__ out_val, uint32_t *__restrict__ rows_per_segment,
                                                              uint64_t cap, TileDesc *__restrict__ desc,
                                                              unsigned long long *__restrict__ offsets,
                                                              uint32_t *__restrict__ serial_ids, bool list_serial) {
    __shared__ __attribute__((aligned(16))) TileDesc lds_desc[FUSED_THREADS];
    __shared__ uint32_t rel[FUSED_THREADS + 1]; // rel[k]: where segment k's points begin among the round's
    __shared__ uint64_t scan_lds[17];
    __shared__ unsigned long long lds_metrics[12];
    __shared__ unsigned long long block_base;
    __shared__ uint32_t serial_in_round;
    __shared__ uint32_t serial_of_round[FUSED_THREADS];
    __shared__ __attribute__((aligned(16))) longlong2 ts_slab[FUSED_THREADS / MDB_WAVE][2 * MDB_WAVE];
    const int lane = threadIdx.x & (MDB_WAVE - 1);
    const int wave = threadIdx.x / MDB_WAVE;
    constexpr int FUSED_SEGMENTS = FUSED_THREADS * FUSED_ROUNDS;
    if (threadIdx.x < 12) lds_metrics[threadIdx.x] = 0;
    if (threadIdx.x == 12) serial_in_round = 0;
    const uint64_t block_first = (uint64_t)blockIdx.x * FUSED_SEGMENTS;

    // ---- how many points the workgroup's segments have, and where they begin in the output ---------------------
    // (the counts are worked out again in the rounds below: kept here they would be an array indexed by the round,
    // which lives in scratch memory - a store and a load per segment)
    uint64_t mine_total = 0;
    bool irregular = false;
#pragma unroll
    for (int k = 0; k < FUSED_ROUNDS; k++) {
        const uint64_t i = block_first + (uint64_t)k * FUSED_THREADS + threadIdx.x;
        if (i < s.n) {
            if (segment_has_regular_timestamps(s, i)) mine_total += regular_segment_points(s, i);
            else irregular = true;
        }
    }
    if (irregular) header->pad = 1u; // (a delta-of-delta stream: the general pipeline takes the batch)
    uint64_t total;
    (void)block_exclusive_scan_u64(mine_total, scan_lds, &total);
    if (wave == 0) {
        // The look-back, by the whole wave: publish this workgroup's total, read the states of the 64 workgroups in
        // front at once, add up their totals down to the nearest one that has its running total already, publish
        // ours. (Workgroups start in the order of their indices, so the ones looked back at are running or done.)
        unsigned long long in_front = 0;
        if (blockIdx.x > 0) {
            if (lane == 0)
                __hip_atomic_store(lookback + blockIdx.x, FUSED_FLAG_AGGREGATE | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            long long top = (long long)blockIdx.x - 1;
            for (;;) {
                const long long b = top - lane;
                const unsigned long long state = b >= 0 ? __hip_atomic_load(lookback + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                        : FUSED_FLAG_PREFIX; // (in front of the first workgroup: nothing)
                const unsigned long long has_prefix = __ballot((state & FUSED_FLAG_PREFIX) != 0);
                const unsigned long long missing = __ballot((state >> 62) == 0);
                const int nearest_prefix = has_prefix ? __ffsll((long long)has_prefix) - 1 : MDB_WAVE - 1;
                const unsigned long long needed = nearest_prefix >= MDB_WAVE - 1 ? ~0ull : ((1ull << (nearest_prefix + 1)) - 1ull);
                if (missing & needed) {
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                unsigned long long value = lane <= nearest_prefix ? (state & FUSED_VALUE_MASK) : 0ull;
#pragma unroll
                for (int delta = 1; delta < MDB_WAVE; delta <<= 1)
                    value += ((unsigned long long)__shfl_xor((uint32_t)(value >> 32), delta, MDB_WAVE) << 32) |
                             __shfl_xor((uint32_t)value, delta, MDB_WAVE);
                in_front += value;
                if (has_prefix) break;
                top -= MDB_WAVE;
            }
        }
        if (lane == 0) {
            __hip_atomic_store(lookback + blockIdx.x, FUSED_FLAG_PREFIX | (in_front + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            block_base = in_front;
            if ((uint64_t)blockIdx.x == (s.n - 1) / FUSED_SEGMENTS) header->total_points = in_front + total;
        }
    }
    __syncthreads();
    unsigned long long running = block_base;

    // ---- 256 segments at a time: analysed from their rows into LDS, their points written -----------------------
#pragma unroll 1
    for (int k = 0; k < FUSED_ROUNDS; k++) {
        const uint64_t round_first = block_first + (uint64_t)k * FUSED_THREADS;
        if (round_first >= s.n) break;
        const uint64_t i = round_first + threadIdx.x;
        const uint32_t n_in_round = (uint32_t)min((uint64_t)FUSED_THREADS, s.n - round_first);
        uint32_t count = 0;
        TileDesc mine = TileDesc{};
        uint32_t error = 0;
        bool serial = false;
        if (i < s.n && segment_has_regular_timestamps(s, i)) {
            count = regular_segment_points(s, i);
            const SegInfo info = analyse_segment<ANALYSE_REGULAR>(s, i);
            error = info.error;
            mine = make_tile_desc(info.desc);
            mine.n_points = count; // (what the layout was made with; differs from the analysis only where that fails)
            mine.n_model = min(mine.n_model, count);
            serial = !error && (info.desc.flags & FLAG_SERIAL) != 0;
            const uint32_t type = mine.flags & FLAG_TYPE_MASK;
            if (type < 3) {
                atomicAdd(&lds_metrics[type], (unsigned long long)count);
                atomicAdd(&lds_metrics[4 + type], 1ull);
            }
            if (info.desc.flags & FLAG_HAS_RESIDUALS) atomicAdd(&lds_metrics[3], 1ull);
            atomicAdd(&lds_metrics[7], 1ull);
        }
        lds_desc[threadIdx.x] = mine;
        uint64_t round_total;
        const uint64_t exclusive = block_exclusive_scan_u64(count, scan_lds, &round_total);
        rel[threadIdx.x] = (uint32_t)exclusive;
        if (error) atomicOr(&header->error, error);
        if (rows_per_segment && i < s.n) rows_per_segment[i] = count;
        // (the kernels behind this one look the segments with serial work up: a place in their list for each of the
        // round's with ONE atomic - one per segment on the one counter took 2 ms for a million of them, one per wave
        // 1 ms for the 160 000 waves of ten million short segments with residual tails; a plain store of a flag to
        // one address by every wave, tried for the case below, 4.6 ms)
        // With cursors into every stream of the batch (a resident batch's index) nobody reads the list: k_grid_mv_pieces
        // goes by the cursors and needs the descriptors and offsets only. Otherwise the round's segments with serial
        // work are gathered in LDS and the first wave moves them to the list with one atomic on its counter.
        const unsigned long long serial_lanes = __ballot(serial);
        if (serial_lanes) {
            const int first_lane = __ffsll((long long)serial_lanes) - 1;
            uint32_t place = 0;
            if (list_serial) {
                if (lane == first_lane) place = atomicAdd(&serial_in_round, (uint32_t)__popcll(serial_lanes));
                place = (uint32_t)__builtin_amdgcn_readlane((int)place, first_lane) + (uint32_t)__popcll(serial_lanes & ((1ull << lane) - 1ull));
            }
            if (serial) {
                desc[i] = mine;
                offsets[i] = running + exclusive;
                if (list_serial) serial_of_round[place] = (uint32_t)i;
            }
        }
        __syncthreads();
        if (list_serial && wave == 0) {
            const uint32_t listed = serial_in_round;
            if (listed) {
                unsigned long long first_place = 0;
                if (lane == 0) first_place = atomicAdd(&header->n_serial, (unsigned long long)listed);
                first_place = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(first_place >> 32), 0) << 32) |
                              (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)first_place, 0);
                for (uint32_t k = lane; k < listed; k += MDB_WAVE) serial_ids[first_place + k] = serial_of_round[k];
                if (lane == 0) serial_in_round = 0; // (for the next round, which starts behind a barrier)
            }
        }

        // The round's points [base, end) of the output, four consecutive ones per lane and step; the groups of four
        // are aligned in the OUTPUT (16-byte stores), so the first and the last group may be shared with the
        // neighbouring round or workgroup: each stores its own points of them one by one.
        const unsigned long long base = min(running, (unsigned long long)cap), end = min(running + round_total, (unsigned long long)cap);
        running += round_total;
        if (base < end) {
            const unsigned long long first_group = base >> 2, groups = ((end + 3) >> 2) - first_group;
            for (unsigned long long g0 = 0; g0 < groups; g0 += FUSED_THREADS) {
                const unsigned long long wave_base = ((first_group + g0) << 2) + (unsigned long long)wave * (MDB_WAVE * 4);
                if (wave_base >= end) break; // (wave-uniform: the wave's points lie behind the round's)
                const unsigned long long p = wave_base + (unsigned long long)lane * 4;
                int64_t t[4] = {0, 0, 0, 0};
                float values[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                if (p + 4 > base && p < end) {
                    // The segment of the group's first point that belongs to this round.
                    const unsigned long long q0 = max(p, base);
                    const uint32_t local = (uint32_t)(q0 - base);
                    // Largest k with rel[k] <= local: segments of a round are about equally long more often than not, so
                    // the search starts where equally long ones would put the point and looks at most four segments
                    // to either side before it halves what is left (eight dependent LDS reads per group otherwise).
                    uint32_t lo = 0, hi = n_in_round;
                    {
                        uint32_t guess = min(n_in_round - 1, (uint32_t)(((unsigned long long)local * n_in_round) / round_total));
#pragma unroll 1
                        for (int step = 0; step < 4 && rel[guess] > local; step++) guess -= 1; // (rel[0] = 0 <= local)
                        if (rel[guess] <= local) {
                            lo = guess;
#pragma unroll 1
                            for (int step = 0; step < 4 && lo + 1 < n_in_round && rel[lo + 1] <= local; step++) lo += 1;
                            hi = (lo + 1 < n_in_round && rel[lo + 1] <= local) ? n_in_round : lo + 1;
                        } else {
                            hi = guess;
                        }
                    }
                    while (hi - lo > 1) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (rel[mid] <= local) lo = mid; else hi = mid;
                    }
                    TileDesc d = lds_desc[lo];
                    unsigned long long segment_offset = base + rel[lo];
                    const uint32_t index = (uint32_t)(q0 - segment_offset);
                    if (p >= base && index + 4 <= d.n_model) { // all four points of one segment's model
                        const PointValue first_point = reconstruct_point(d, index);
                        t[0] = first_point.t; t[1] = t[0] + d.delta; t[2] = t[1] + d.delta; t[3] = t[2] + d.delta;
                        if ((d.flags & FLAG_TYPE_MASK) == MDB_SWING_ID) {
                            values[0] = first_point.v;
                            values[1] = (float)(d.slope * (double)t[1] + d.intercept);
                            values[2] = (float)(d.slope * (double)t[2] + d.intercept);
                            values[3] = (float)(d.slope * (double)t[3] + d.intercept);
                        } else {
                            values[0] = values[1] = values[2] = values[3] = d.value;
                        }
                    } else {
                        unsigned long long next_offset = segment_offset + d.n_points;
#pragma unroll
                        for (uint32_t m = 0; m < 4; m++) {
                            const unsigned long long q = p + m;
                            if (q < base) continue;
                            if (q >= end) break;
                            while (q >= next_offset) {
                                lo += 1;
                                segment_offset = next_offset;
                                d = lds_desc[lo];
                                next_offset = segment_offset + d.n_points;
                            }
                            const uint32_t at = (uint32_t)(q - segment_offset);
                            PointValue point = reconstruct_point(d, at);
                            // (a residual or MacaqueV value: a placeholder, the kernels behind this one write it)
                            if (at >= d.n_model || (d.flags & FLAG_TYPE_MASK) == MDB_MACAQUE_V_ID) point.v = 0.0f;
                            if (m == 0) { t[0] = point.t; values[0] = point.v; }
                            if (m == 1) { t[1] = point.t; values[1] = point.v; }
                            if (m == 2) { t[2] = point.t; values[2] = point.v; }
                            if (m == 3) { t[3] = point.t; values[3] = point.v; }
                        }
                    }
                    if (p >= base && p + 4 <= end) {
                        *reinterpret_cast<float4 *>(out_val + p) = make_float4(values[0], values[1], values[2], values[3]);
                    } else {
#pragma unroll
                        for (uint32_t m = 0; m < 4; m++)
                            if (p + m >= base && p + m < end) out_val[p + m] = values[m];
                    }
                }
                // Timestamps through the wave's LDS slab, as k_grid_tiles does: 1 KiB per store instruction.
                longlong2 *slab = ts_slab[wave];
                slab[2 * lane] = make_longlong2(t[0], t[1]);
                slab[2 * lane + 1] = make_longlong2(t[2], t[3]);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const longlong2 chunk_a = slab[lane];
                const longlong2 chunk_b = slab[MDB_WAVE + lane];
                __builtin_amdgcn_wave_barrier();
                const unsigned long long point_a = wave_base + 2 * (unsigned long long)lane;
                const unsigned long long point_b = point_a + 2 * MDB_WAVE;
                longlong2 *ts_out = reinterpret_cast<longlong2 *>(out_ts + wave_base);
                if (point_a >= base && point_a + 2 <= end) ts_out[lane] = chunk_a;
                else {
                    if (point_a >= base && point_a < end) out_ts[point_a] = chunk_a.x;
                    if (point_a + 1 >= base && point_a + 1 < end) out_ts[point_a + 1] = chunk_a.y;
                }
                if (point_b >= base && point_b + 2 <= end) ts_out[MDB_WAVE + lane] = chunk_b;
                else {
                    if (point_b >= base && point_b < end) out_ts[point_b] = chunk_b.x;
                    if (point_b + 1 >= base && point_b + 1 < end) out_ts[point_b + 1] = chunk_b.y;
                }
            }
        }
        __syncthreads(); // (the next round reuses lds_desc and rel)
    }
    if (threadIdx.x < 10 && lds_metrics[threadIdx.x]) atomicAdd(&header->metrics[threadIdx.x], lds_metrics[threadIdx.x]);
}

// ---- irregular timestamps, one lane per piece of a stream ---------------------------------------------------
//
// timestamps.rs:228-292 decodes a delta-of-delta stream code by code; what it carries from one code to
// the next is a TsCursor, and the prepass - which has to walk every such stream once anyway, because
// the number of points of a segment is the number of its codes - has left the cursor in front of the
// first code of every 256-bit piece. So the second walk is not one: every piece is decoded by a lane
// of its own (about 22 codes of a randomly sampled series, up to 256 of a fixed-rate series with gaps),
// lanes next to each other decode pieces next to each other and write runs of points next to each
// other. Swing values are (slope * t + intercept) of the timestamps just decoded (swing.rs:304-319, the
// line is in the descriptor); PMC-Mean values do not depend on the timestamp (k_grid_tiles has written
// them); MacaqueV values and residual tails are k_grid_serial's.

struct TsPieceCount { // pieces of the stream of segment i, 0 if it has no checkpoints
    DevSegments s;
    __device__ uint64_t operator()(uint64_t i) const {
        const uint4 view = s.timestamps.views[i];
        const int32_t length = (int32_t)view.x;
        const bool irregular = length > 0 && (view_inline_byte(view, 0) & 0x80u) != 0;
        return irregular && ts_has_checkpoints(length) ? ts_pieces((uint32_t)length) : 0u;
    }
};

// ---- the serial kernel -------------------------------------------------------------------------------
//
// One lane per segment that carries a serial dependency, one wave per workgroup. MacaqueV streams
// (a model's values and/or the residual tail) are decoded from an LDS ring: every lane keeps the
// next SERIAL_RING_WORDS 32-bit words of ITS bitstream in LDS ([slot][lane], bank-conflict free),
// and when any lane runs low the whole wave tops all rings up with independent predicated loads
// issued back to back - one memory latency per ~24 words instead of one per word. The decode itself
// stays sequential per stream: value i's position depends on every earlier value.
// Irregular (delta-of-delta) timestamps are rare and keep the direct reader.

constexpr int SERIAL_RING_WORDS = 32;
constexpr int SERIAL_TOPUP_WORDS = 24;
constexpr int SERIAL_THREADS = MDB_WAVE;

struct RingBitReader {
    const uint32_t *words; // aligned base in global memory
    uint32_t n_words;
    uint32_t next_word; // next word to move from the ring into the bit buffer
    uint32_t loaded;    // words [next_word, loaded) are in the ring
    uint64_t buffer;    // MSB aligned
    int32_t available;
    uint32_t skip_bits; // slack bits in front of the first payload byte
    uint64_t used_bits;
    uint64_t total_bits;
    uint32_t ahead;     // refill(): ring word next_word, byte swapped

    __device__ __forceinline__ void begin(const uint8_t *bytes, uint64_t nbytes) {
        uintptr_t address = reinterpret_cast<uintptr_t>(bytes);
        uint32_t misalign = (uint32_t)(address & 3u);
        words = reinterpret_cast<const uint32_t *>(address - misalign);
        n_words = (uint32_t)((nbytes + misalign + 3u) >> 2);
        next_word = 0;
        loaded = 0;
        buffer = 0;
        available = 0;
        skip_bits = 8u * misalign;
        used_bits = 0;
        total_bits = nbytes * 8u;
        ahead = 0;
    }
    __device__ __forceinline__ bool hungry() const { return loaded < n_words && loaded - next_word < 3; }
    __device__ __forceinline__ void pull(const uint32_t (*ring)[MDB_WAVE], int lane) {
        while (available <= 32 && next_word < loaded) {
            uint32_t w = __builtin_bswap32(ring[next_word % SERIAL_RING_WORDS][lane]);
            next_word += 1;
            if (skip_bits) { // only the very first word can carry slack bytes
                buffer |= ((uint64_t)w << 32) << skip_bits;
                available += 32 - (int32_t)skip_bits;
                skip_bits = 0;
            } else {
                buffer |= (uint64_t)w << (32 - available);
                available += 32;
            }
        }
    }
    __device__ __forceinline__ uint32_t get(uint32_t count, const uint32_t (*ring)[MDB_WAVE], int lane) {
        if (count == 0) return 0;
        pull(ring, lane);
        uint32_t value = (uint32_t)(buffer >> (64u - count));
        buffer <<= count;
        available -= (int32_t)count;
        used_bits += count;
        return value;
    }
    // The same without branches (64 lanes that each stand somewhere else in a code execute both sides
    // of every branch anyway): at most one word, so a caller that needs up to 45 bits refills before
    // the control bits and again before the payload. Needs 0 <= available while words remain, which
    // holds from the second refill of a stream on (the first word may bring as few as 8 bits).
    // The word comes out of a register (`ahead` = ring word next_word, see look_ahead) so that no LDS
    // latency sits on the chain from one code to the next.
    __device__ __forceinline__ void refill(const uint32_t (*ring)[MDB_WAVE], int lane) {
        const bool want = available <= 32 && next_word < loaded;
        const uint64_t placed = (((uint64_t)ahead << 32) << skip_bits) >> (available & 63);
        buffer |= want ? placed : 0ull;
        available += want ? 32 - (int32_t)skip_bits : 0;
        skip_bits = want ? 0u : skip_bits;
        next_word += want ? 1u : 0u;
        look_ahead(ring, lane);
    }
    // Ring slot of next_word, read whether or not it has been filled yet: it is not used before it has
    // (refill() tests next_word < loaded), and a stream that starts calls this again after its first
    // top-up. A top-up never overwrites the slots [next_word, loaded).
    __device__ __forceinline__ void look_ahead(const uint32_t (*ring)[MDB_WAVE], int lane) {
        ahead = __builtin_bswap32(ring[next_word % SERIAL_RING_WORDS][lane]);
    }
    __device__ __forceinline__ void consume(uint32_t count) {
        buffer <<= count;
        available -= (int32_t)count;
        used_bits += count;
    }
    __device__ __forceinline__ bool overrun() const { return used_bits > total_bits; }
};

struct MacaqueStream {
    uint32_t remaining;  // values still to decode
    uint32_t position;   // index of the next value inside the whole segment
    uint32_t last;       // bits of the previous value
    uint32_t leading, trailing;
    bool first_is_raw;   // the next value is stored as 32 raw bits (macaque_v.rs:289-293)
    bool fresh;          // nothing has been read from the stream yet
};

// Tops the ring of every lane of the wave up at once (the caller has found a hungry lane): independent
// predicated loads first, then the LDS writes.
__device__ __forceinline__ void ring_top_up(RingBitReader &reader, uint32_t (*ring)[MDB_WAVE], int lane, bool active) {
    uint32_t fetched[SERIAL_TOPUP_WORDS];
    const uint32_t first = reader.loaded;
    const uint32_t room = active ? SERIAL_RING_WORDS - (reader.loaded - reader.next_word) : 0u;
#pragma unroll
    for (int k = 0; k < SERIAL_TOPUP_WORDS; k++) {
        const uint32_t index = first + k;
        fetched[k] = ((uint32_t)k < room && index < reader.n_words) ? load_global(reader.words + index) : 0u;
    }
#pragma unroll
    for (int k = 0; k < SERIAL_TOPUP_WORDS; k++) {
        const uint32_t index = first + k;
        if ((uint32_t)k < room && index < reader.n_words) ring[index % SERIAL_RING_WORDS][lane] = fetched[k];
    }
    reader.loaded = min(reader.n_words, first + min(room, (uint32_t)SERIAL_TOPUP_WORDS));
}

// One value (macaque_v.rs:297-322): `10` it repeats, `0` + the bits of the previous window, `11` + 5
// bits leading zeros + 6 bits length + the bits - at most 45 bits, of which the control bits come off
// the top of the 64-bit buffer after one refill and the payload after another. Every lane of the wave
// stands at a different kind of code, so the three cases are computed with selects rather than
// branched to. Returns the bits of the value (stream.last is updated); *malformed: the window is
// impossible, the value returned is the previous one and the caller stops.
__device__ __forceinline__ uint32_t ring_decode_value(RingBitReader &reader, MacaqueStream &stream,
                                                      const uint32_t (*ring)[MDB_WAVE], int lane, bool *malformed) {
    if (stream.fresh) {
        reader.look_ahead(ring, lane);
        reader.refill(ring, lane);
        stream.fresh = false;
    }
    reader.refill(ring, lane);
    const uint32_t top = (uint32_t)(reader.buffer >> 51); // 13 bits: c0 c1 lz[5] len[6]
    const bool raw = stream.first_is_raw;                 // 32 raw bits, no control bits
    const bool c0 = (top >> 12) != 0u, c1 = ((top >> 11) & 1u) != 0u;
    const bool opens = !raw && c0 && c1;
    const bool repeats = !raw && c0 && !c1;
    const uint32_t header_bits = raw ? 0u : (c0 ? (c1 ? 13u : 2u) : 1u);
    const uint32_t leading = opens ? ((top >> 6) & 31u) : stream.leading;
    const uint32_t trailing = opens ? 32u - (top & 63u) - leading : stream.trailing;
    uint32_t meaningful = 32u - leading - trailing;
    *malformed = !raw && !repeats && (meaningful > 32u || trailing > 31u);
    const bool silent = repeats || *malformed; // no payload: the value is the previous one
    stream.leading = leading;
    stream.trailing = trailing;
    stream.first_is_raw = false;
    meaningful = raw ? 32u : (silent ? 0u : meaningful);
    reader.consume(header_bits);
    reader.refill(ring, lane);
    const uint32_t payload = (uint32_t)((reader.buffer >> 1) >> (63u - meaningful)); // 0 bits: 0
    reader.consume(meaningful);
    const uint32_t bits = raw ? payload : (stream.last ^ (payload << (trailing & 31u)));
    stream.last = bits;
    return bits;
}

// ---- random access into MacaqueV streams: the cursor index of a resident batch -----------------------------
//
// macaque_v.rs:272-323 decodes a stream from its first bit: where code i begins depends on every code before it,
// and its value on the XOR of all of them. One lane per stream therefore needs 20 ms for 50 000 values however idle
// the GPU is (MacaqueV streams). But a batch that STAYS on the device (mdb_segments_upload,
// mdb_compress_chunks_dev) is decoded again and again, and what a lane needs to start in the middle of a stream is
// little: the bit position of the code, the window (leading / trailing zeros of the last `11` code) and the XOR of
// all deltas so far. k_mv_index_walk leaves such a cursor in front of every 64th value of every stream - the
// model's values of a MacaqueV segment, and the residual tail of any segment - once per batch; from then on
// k_grid_mv_pieces decodes every piece with a lane of its own, 64 pieces per wave, the wave's 4 096 values staged
// in LDS and written in rows of 64 consecutive values.
//
// The XOR is kept relative to the stream's seed: value = seed ^ cursor.xor_bits ^ (deltas of the piece so far). A
// residual tail is XOR-seeded with the last RECONSTRUCTED value of its model by grid() (models/mod.rs:241-249:
// for a Swing model with irregular timestamps only the grid call's own analysis knows it) and with the last
// DECODED value, or NaN for a MacaqueV model, by sum() (models/mod.rs:145-181) - the caller brings the seed. A
// MacaqueV segment's values start from 0 with the raw first value as their first delta; the last of them is what
// grid() seeds such a segment's residual tail with (the reference's compressor never makes one, a foreign batch
// may), and the walk leaves it in the tail's cursors (chain_seed).

// (MvCursor, MV_PIECE_VALUES, MV_WINDOW_*: mdb_common.hpp)

// Values of the two streams of segment i: the model's (MacaqueV segments only) and the residual tail's.
__device__ __forceinline__ void mv_stream_lengths(const DevSegments &s, uint64_t i, const uint32_t *known_totals,
                                                  uint32_t *n_model_values, uint32_t *n_residuals, uint32_t *n_model_points,
                                                  uint32_t *error) {
    const SegInfo info = analyse_segment(s, i, known_totals, nullptr, false);
    *error = info.error;
    const uint32_t n_res = info.desc.n_total - info.desc.n_model;
    *n_model_points = info.desc.n_model;
    *n_model_values = (info.desc.flags & FLAG_TYPE_MASK) == MDB_MACAQUE_V_ID ? info.desc.n_model : 0u;
    *n_residuals = n_res;
}

struct MvIndexPieces { // pieces of segment i (0 for a malformed one: the batch is then not indexed at all)
    DevSegments s;
    const uint32_t *known_totals;
    __device__ uint64_t operator()(uint64_t i) const {
        uint32_t n_values, n_res, n_model, error;
        mv_stream_lengths(s, i, known_totals, &n_values, &n_res, &n_model, &error);
        if (error) return 0;
        return (uint64_t)((n_values + MV_PIECE_VALUES - 1) / MV_PIECE_VALUES) + (n_res + MV_PIECE_VALUES - 1) / MV_PIECE_VALUES;
    }
};

// One lane per segment, one wave per workgroup, the streams read through the LDS ring of k_grid_serial: the walk
// that k_grid_serial makes on every call, made once, leaving cursors instead of values. verdict[0] |= what is
// wrong with a stream (the batch is then left to the kernels that report it), verdict[1] += values walked.
__global__ __launch_bounds__(SERIAL_THREADS) void k_mv_index_walk(DevSegments s, const uint32_t *__restrict__ known_totals,
                                                                  const unsigned long long *__restrict__ piece_base,
                                                                  MvCursor *__restrict__ cursors,
                                                                  unsigned long long *__restrict__ verdict) {
    __shared__ uint32_t ring[SERIAL_RING_WORDS][MDB_WAVE];
    const int lane = threadIdx.x;
    const uint64_t i = (uint64_t)blockIdx.x * SERIAL_THREADS + lane;
    const bool present = i < s.n;
    uint32_t n_values = 0, n_res = 0, n_model = 0, error = 0;
    if (present) mv_stream_lengths(s, i, known_totals, &n_values, &n_res, &n_model, &error);
    if (error) n_values = n_res = 0;
    unsigned long long piece = present ? piece_base[i] : 0;
    bool values_pending = n_values > 0, residuals_pending = n_res > 0;
    RingBitReader reader;
    reader.begin(nullptr, 0);
    MacaqueStream stream;
    stream.remaining = 0; stream.position = 0; stream.last = 0;
    stream.leading = 255; stream.trailing = 0; stream.first_is_raw = false; stream.fresh = false;
    bool active = false, residual = false;
    uint32_t in_stream = 0;  // values of the open stream walked so far
    uint32_t chain_seed = 0; // (residual tail: the last value of the MacaqueV model in front of it)
    unsigned long long walked = 0;
    auto open_next_stream = [&]() {
        in_stream = 0;
        if (values_pending) {
            const uint4 vv = s.values.views[i];
            reader.begin(view_data(s.values, i, vv), vv.x);
            if (vv.x == 0) error |= ERR_BITSTREAM;
            stream.remaining = n_values; stream.position = 0; stream.leading = 255; stream.trailing = 0;
            stream.first_is_raw = true; stream.fresh = true;
            values_pending = false; residual = false;
            active = vv.x != 0;
        } else if (residuals_pending) {
            const uint4 vr = s.residuals.views[i];
            reader.begin(view_data(s.residuals, i, vr), vr.x - 1);
            if (vr.x < 2) error |= ERR_BITSTREAM;
            stream.remaining = n_res; stream.position = n_model; stream.leading = 255; stream.trailing = 0;
            stream.first_is_raw = false; stream.fresh = true;
            chain_seed = stream.last; // (0 unless a MacaqueV model's values have just been walked)
            stream.last = 0;
            residuals_pending = false; residual = true;
            active = vr.x >= 2;
        } else {
            active = false;
        }
    };
    open_next_stream();
    while (__any(active)) {
        if (__any(active && reader.hungry())) ring_top_up(reader, ring, lane, active);
        if (active) {
            if (in_stream % MV_PIECE_VALUES == 0) {
                MvCursor cursor;
                cursor.bit_position = (uint32_t)reader.used_bits;
                cursor.xor_bits = stream.last;
                cursor.segment = (uint32_t)i;
                cursor.point_index = stream.position;
                cursor.n_values = min(stream.remaining, MV_PIECE_VALUES);
                cursor.window = (stream.leading & 255u) | ((stream.trailing & 255u) << 8) | (residual ? MV_WINDOW_RESIDUAL : 0u) |
                                (stream.first_is_raw ? MV_WINDOW_RAW : 0u);
                cursor.chain_seed = residual ? chain_seed : 0u;
                cursor.pad = 0;
                uint4 *to = reinterpret_cast<uint4 *>(cursors + piece);
                const uint4 *from = reinterpret_cast<const uint4 *>(&cursor);
                to[0] = from[0];
                to[1] = from[1];
                piece += 1;
            }
            bool malformed;
            (void)ring_decode_value(reader, stream, ring, lane, &malformed);
            if (malformed) {
                error |= ERR_BITSTREAM;
                stream.remaining = 1;
                values_pending = residuals_pending = false;
            }
            stream.position += 1;
            stream.remaining -= 1;
            in_stream += 1;
            walked += 1;
            if (stream.remaining == 0) {
                if (reader.overrun() || reader.used_bits > 0xffffffffull) error |= ERR_BITSTREAM;
                open_next_stream();
            }
        }
    }
    if (error) atomicOr(verdict, (unsigned long long)error);
    if (walked) atomicAdd(verdict + 1, walked);
}

// ---- the same step for the kernels that decode one PIECE per lane, on 32-bit words -------------------------------
//
// ring_decode_value's step over a 64-bit buffer with a reader of its own per lane cost the wave 104 vector instructions
// per value (every shift, add and compare on 64 bits is two or three instructions, two refills per code each rotated
// four words through registers, the bit position was 64 bits wide). Here a lane keeps the three big-endian words its next code can reach into (a code is at most 45 bits:
// 13 of header, 32 of payload) and the bit offset into the first; header and payload come out of them with one funnel
// shift each, and the words behind them come from a ring of the lane's stream in LDS ([word][lane]: the lanes of a
// wave read different rows of their own column, two lanes per bank at worst), read at the top of the step and needed
// at its end. The ring is topped up for the whole wave from 16-byte chunks that were loaded one top-up earlier.
constexpr int PIECE_RING_WORDS = 16;
constexpr int PIECE_RING_ROWS = PIECE_RING_WORDS + 4; // (and four rows nobody reads: where a lane without room puts its chunk)
struct PieceReader {
    const uint4 *chunks; // 16-byte aligned; chunk k holds words [4 k, 4 k + 4) of the stream as this reader counts them
    uint32_t last_chunk; // the last one that holds payload (loads never go behind it)
    uint32_t loaded;     // words [.., loaded) have been put into the ring; a multiple of 4
    uint32_t word;       // index of w0
    uint32_t w0, w1, w2; // words word, word + 1, word + 2 (big endian: the stream's first bit on top)
    uint32_t shift;      // bits of w0 already consumed (0..31)
    uint4 ahead, further, beyond, last; // chunks loaded / 4 .. loaded / 4 + 3, on their way from memory

    __device__ __forceinline__ uint4 load(uint32_t index) const { return load_global(chunks + min(index, last_chunk)); }
    // nbytes > 0
    __device__ __forceinline__ void open(const uint8_t *bytes, uint64_t nbytes, uint64_t start_bit) {
        const uintptr_t address = reinterpret_cast<uintptr_t>(bytes);
        const uint32_t misalign = (uint32_t)(address & 15u);
        const uint64_t first_bit = 8ull * misalign + start_bit; // counted from the aligned base
        const uint64_t skipped = first_bit >> 7;                // whole chunks in front of it
        const uint64_t all_chunks = (nbytes + misalign + 15u) >> 4;
        chunks = reinterpret_cast<const uint4 *>(address - misalign) + skipped;
        last_chunk = (uint32_t)(all_chunks > skipped ? all_chunks - skipped - 1 : 0u);
        word = (uint32_t)((first_bit >> 5) & 3u);
        shift = (uint32_t)(first_bit & 31u);
    }
    // A lane without a piece runs through the same straight-line code as the others (nothing it makes is looked at):
    // its reader reads `anywhere`, 16 bytes that may be read.
    __device__ __forceinline__ void idle(const void *anywhere) {
        chunks = reinterpret_cast<const uint4 *>(reinterpret_cast<uintptr_t>(anywhere) & ~(uintptr_t)15u);
        last_chunk = 0;
        word = shift = 0;
    }
    __device__ __forceinline__ void begin() { // (every lane, after open() or idle())
        loaded = 0;
        w0 = w1 = w2 = 0;
        ahead = load(0);
        further = load(1);
        beyond = load(2);
        last = load(3);
    }
    // Can the next TWO values be decoded without another look? (a value moves up at most two words; the second one's
    // words behind w2 are rows word + 5 and word + 6 then)
    __device__ __forceinline__ bool hungry() const { return loaded < word + 7u; }
    // The whole wave, without a branch: every lane puts the chunk it has waited for into its column - behind what it
    // has there, or into rows nobody reads when there is no room for it (a lane far ahead of the hungry one) - and
    // asks for another one (the same one again when it could not place this one). Four chunks are under way: the one
    // placed now was asked for four top-ups - the time of a dozen values - ago.
    __device__ __forceinline__ void top_up(uint32_t (*ring)[MDB_WAVE], int lane) {
        const bool room = loaded <= word + ((uint32_t)PIECE_RING_WORDS - 4u);
        const uint32_t row = room ? loaded & (uint32_t)(PIECE_RING_WORDS - 1) : (uint32_t)PIECE_RING_WORDS;
        ring[row + 0][lane] = ahead.x;
        ring[row + 1][lane] = ahead.y;
        ring[row + 2][lane] = ahead.z;
        ring[row + 3][lane] = ahead.w;
        loaded += room ? 4u : 0u;
        auto move_up = [room](uint4 &to, const uint4 &from) {
            to.x = room ? from.x : to.x;
            to.y = room ? from.y : to.y;
            to.z = room ? from.z : to.z;
            to.w = room ? from.w : to.w;
        };
        move_up(ahead, further);
        move_up(further, beyond);
        move_up(beyond, last);
        last = load((loaded >> 2) + 3u);
    }
    // After the first top-ups: the three words the first code can reach into.
    __device__ __forceinline__ void start(const uint32_t (*ring)[MDB_WAVE], int lane) {
        w0 = __builtin_bswap32(ring[word][lane]);
        w1 = __builtin_bswap32(ring[word + 1u][lane]);
        w2 = __builtin_bswap32(ring[word + 2u][lane]);
    }
};

struct PieceState {
    uint32_t last;        // bits of the previous value
    uint32_t trailing;    // of the window the last `11` code opened
    uint32_t window_bits; // 32 - leading - trailing of that window (0: none yet - the index has seen every code: there is)
    bool raw;             // the next value is 32 raw bits (macaque_v.rs:289-293)
};

// The 32 bits that begin `shift` (0..31) bits into the 64 bits high:low.
__device__ __forceinline__ uint32_t bits_at(uint32_t high, uint32_t low, uint32_t shift) {
    return (uint32_t)(((((uint64_t)high) << 32) | low) << shift >> 32);
}

// One value (macaque_v.rs:297-322; the cursor index has seen every window of the stream: they are possible ones).
__device__ __forceinline__ uint32_t piece_decode_value(PieceReader &reader, PieceState &state, const uint32_t (*ring)[MDB_WAVE], int lane) {
    // (the two words that may move up, asked for now, needed last)
    const uint32_t behind0 = ring[(reader.word + 3u) & (uint32_t)(PIECE_RING_WORDS - 1)][lane];
    const uint32_t behind1 = ring[(reader.word + 4u) & (uint32_t)(PIECE_RING_WORDS - 1)][lane];
    const uint32_t head = bits_at(reader.w0, reader.w1, reader.shift); // the next 32 bits of the stream
    const uint32_t code = head >> 30;                                  // 0x: `0`, 2: `10`, 3: `11`
    const bool raw = state.raw;
    const bool opens = !raw && code == 3u, repeats = !raw && code == 2u;
    const uint32_t header_bits = raw ? 0u : ((0x0d020101u >> (code << 3)) & 15u);
    const uint32_t leading = (head >> 25) & 31u, length = (head >> 19) & 63u;
    state.trailing = opens ? (32u - length - leading) & 31u : state.trailing;
    state.window_bits = opens ? min(length, 32u) : state.window_bits;
    const uint32_t meaningful = raw ? 32u : (repeats ? 0u : state.window_bits);
    const uint32_t at = reader.shift + header_bits; // 0..44: where the payload begins, in bits from the top of w0
    const bool in_first = at < 32u;
    const uint32_t body = bits_at(in_first ? reader.w0 : reader.w1, in_first ? reader.w1 : reader.w2, at & 31u);
    const uint32_t payload = meaningful ? body >> ((32u - meaningful) & 31u) : 0u;
    const uint32_t bits = raw ? payload : (state.last ^ (payload << state.trailing));
    state.last = bits;
    state.raw = false;
    const uint32_t end = at + meaningful; // 0..76
    const uint32_t taken = end >> 5;      // whole words consumed: 0, 1 or 2
    reader.shift = end & 31u;
    reader.word += taken;
    const uint32_t up0 = __builtin_bswap32(behind0), up1 = __builtin_bswap32(behind1);
    const uint32_t n0 = taken == 0u ? reader.w0 : (taken == 1u ? reader.w1 : reader.w2);
    const uint32_t n1 = taken == 0u ? reader.w1 : (taken == 1u ? reader.w2 : up0);
    const uint32_t n2 = taken == 0u ? reader.w2 : (taken == 1u ? up0 : up1);
    reader.w0 = n0;
    reader.w1 = n1;
    reader.w2 = n2;
    return bits;
}

// The largest of the lanes' values, in every lane.
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x) {
#pragma unroll
    for (int step = 1; step < MDB_WAVE; step <<= 1) x = max(x, (uint32_t)__shfl_xor((int)x, step, MDB_WAVE));
    return x;
}

// The one kind of stream of an indexed batch that k_grid_serial keeps (see k_grid_mv_pieces).
__device__ __forceinline__ bool mv_left_to_serial(uint32_t tile_flags) {
    return (tile_flags & FLAG_JUMPS) && (tile_flags & FLAG_TYPE_MASK) == MDB_SWING_ID;
}

// One lane per piece, one wave per workgroup. The piece is decoded ROUND values at a time into the wave's LDS
// (a row per lane) and written out between the rounds, ROUND consecutive values per lane-row. (Measured: the
// kernel is bound by vector instructions - 104 per value step of the wave at four cycles each, rocprofv3
// SQ_INSTS_VALU x 4 = SQ_WAVE_CYCLES - so neither more waves per SIMD (16 / 32 / 64 values per round: 15.3 / 13.8
// / 16.3 ms for 5 x 10^9 values) nor staging the streams in LDS (25.7 ms: more instructions, and pieces 48 words
// apart meet in two banks) changes it; 32 values per round are whole 128-byte lines per row.)
// MDB_MVP_TIMING (a build of its own, never the product's): shader clock cycles of k_grid_mv_pieces' waves in [0] the
// whole kernel, [1] the decode loops with [2] their top-ups (and [3] how many), [4] the rows written out; [5] waves.
#ifdef MDB_MVP_TIMING
__device__ unsigned long long g_mvp_timing[8];
#define MVP_CLOCK() __builtin_amdgcn_s_memtime()
#else
#define MVP_CLOCK() 0ull
#endif

template <int ROUND>
__global__ __launch_bounds__(MDB_WAVE) void k_grid_mv_pieces(DevSegments s, TimeRange range, const TileDesc *__restrict__ desc,
                                                             const unsigned long long *__restrict__ offsets,
                                                             const uint32_t *__restrict__ irregular_first,
                                                             const MvCursor *__restrict__ cursors, unsigned long long n_pieces,
                                                             float *__restrict__ out_val, GridHeader *__restrict__ header) {
    constexpr int STRIDE = ROUND + 1; // (a row per lane: an odd stride keeps the banks apart)
    __shared__ uint32_t stage[MDB_WAVE * STRIDE];
    __shared__ uint32_t ring[PIECE_RING_ROWS][MDB_WAVE];
    __shared__ uint32_t row_count[MDB_WAVE];
    __shared__ unsigned long long row_out[MDB_WAVE];
    const int lane = threadIdx.x;
    [[maybe_unused]] const unsigned long long t_kernel = MVP_CLOCK();
    [[maybe_unused]] unsigned long long t_decode = 0, t_top_up = 0, n_top_up = 0, t_out = 0;
    const unsigned long long piece = (unsigned long long)blockIdx.x * MDB_WAVE + lane;
    // (the columns' first data buffers, asked for before anything else: see view_data())
    const uint8_t *values_first = first_buffer(s.values), *residuals_first = first_buffer(s.residuals);
    uint32_t count = 0, skip = 0; // values of this lane's piece to write, and how many in front of them are not wanted
    unsigned long long out_at = 0;
    PieceReader reader;
    PieceState state;
    reader.idle(cursors);
    state.last = 0; state.trailing = 0; state.window_bits = 0; state.raw = false;
    if (piece < n_pieces) {
        const uint4 c0 = load_global(reinterpret_cast<const uint4 *>(cursors + piece));
        const uint4 c1 = load_global(reinterpret_cast<const uint4 *>(cursors + piece) + 1);
        const uint32_t i = c0.z, point_index = c0.w, n_values = c1.x, window = c1.y;
        const TileDesc t = desc[i];
        // Cursors that host threads have left (mv_host_index) mark the last piece of every stream: it has to end where
        // this call's own analysis of the segment ends the stream, or the two disagree about the segment's length.
        if ((c1.w & MV_CURSOR_LAST_OF_STREAM) && !range.enabled &&
            point_index + n_values != ((window & MV_WINDOW_RESIDUAL) ? t.n_points : t.n_model))
            atomicOr(&header->error, ERR_HOST_INDEX);
        uint32_t first = 0;
        if (range.enabled) {
            if (t.flags & FLAG_REGULAR) first = t.delta > 0 ? (uint32_t)((uint64_t)(t.start - s.start_time[i]) / (uint64_t)t.delta) : 0u;
            else first = irregular_first[i];
        }
        const uint32_t lo = max(point_index, first), hi = min(point_index + n_values, first + t.n_points);
        // (a Swing segment whose timestamps the tile kernel writes from a jump list keeps the list's place where the
        // seed of its residual tail would be, set_jump_list(): such a tail is left to k_grid_serial, which works the
        // seed out again - the analysis that takes would cost this kernel 40 registers)
        if (lo < hi && !mv_left_to_serial(t.flags)) {
            skip = lo - point_index;
            count = hi - lo;
            out_at = offsets[i] + (lo - first);
            const bool residual = (window & MV_WINDOW_RESIDUAL) != 0;
            const DevCol &column = residual ? s.residuals : s.values;
            const uint4 view = column.views[i];
            const uint64_t nbytes = residual ? (uint64_t)view.x - 1u : (uint64_t)view.x;
            reader.open(view_data(column, i, view, residual ? residuals_first : values_first), nbytes, c0.x);
            const bool macaque = (t.flags & FLAG_TYPE_MASK) == MDB_MACAQUE_V_ID;
            state.last = (macaque ? c1.z : __float_as_uint(t.value)) ^ c0.y;
            // (no window yet - leading 255 - is a window of no bits: the stream's first code opens one)
            const uint32_t leading = window & 255u, trailing = (window >> 8) & 255u;
            state.trailing = trailing & 31u;
            state.window_bits = leading + trailing <= 32u ? 32u - leading - trailing : 0u;
            state.raw = (window & MV_WINDOW_RAW) != 0;
        }
    }
    // (every lane decodes in every step, wanted or not: straight-line code; a lane that is through with its piece
    // makes values nobody looks at from bytes it may read)
    reader.begin();
    reader.top_up(ring, lane);
    reader.top_up(ring, lane);
    reader.start(ring, lane);
    // The values in front of the wanted ones are decoded (the chain runs through them) and dropped.
    for (uint32_t k = 0, most = wave_max_u32(skip); k < most; k++) {
        if (__any(reader.hungry())) reader.top_up(ring, lane);
        const PieceReader reader_before = reader;
        const PieceState state_before = state;
        (void)piece_decode_value(reader, state, ring, lane);
        if (k >= skip) { // (not one of this lane's: it stays where it was)
            reader = reader_before;
            state = state_before;
        }
    }
    for (uint32_t done = 0; __any(done < count); done += ROUND) {
        const uint32_t mine = done < count ? min(count - done, (uint32_t)ROUND) : 0u;
        static_assert(ROUND % 2 == 0, "values are decoded in pairs");
        const unsigned long long t_loop = MVP_CLOCK();
        for (uint32_t k = 0, most = wave_max_u32(mine); k < most; k += 2) {
            if (__any(reader.hungry())) {
                const unsigned long long t0 = MVP_CLOCK();
                reader.top_up(ring, lane);
#ifdef MDB_MVP_TIMING
                __builtin_amdgcn_s_waitcnt(0);
#endif
                t_top_up += MVP_CLOCK() - t0;
                n_top_up += 1;
            }
            stage[lane * STRIDE + k] = piece_decode_value(reader, state, ring, lane);
            stage[lane * STRIDE + k + 1] = piece_decode_value(reader, state, ring, lane); // (k + 1 == ROUND: the row's spare word)
        }
        t_decode += MVP_CLOCK() - t_loop;
        const unsigned long long t_rows = MVP_CLOCK();
        // Row r = this round's values of lane r's piece: consecutive floats, (64 / ROUND) rows per store instruction.
        // (how many, and where to: through LDS, read by the lanes of a store without waiting for each other)
        row_count[lane] = mine;
        row_out[lane] = out_at + done;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        constexpr int ROWS_PER_STORE = MDB_WAVE / ROUND;
        const int sub_row = lane / ROUND, column_of_lane = lane % ROUND;
        // (eight stores' rows read from LDS together, then the stores: one wait for eight, not two for each)
        constexpr int BATCH = 8;
        static_assert(MDB_WAVE % (BATCH * ROWS_PER_STORE) == 0, "whole batches of stores");
        for (int r0 = 0; r0 < MDB_WAVE; r0 += BATCH * ROWS_PER_STORE) {
            uint32_t counts[BATCH], staged[BATCH];
            unsigned long long to[BATCH];
#pragma unroll
            for (int q = 0; q < BATCH; q++) {
                const int r = r0 + q * ROWS_PER_STORE + sub_row;
                counts[q] = row_count[r];
                to[q] = row_out[r];
                staged[q] = stage[r * STRIDE + column_of_lane];
            }
#pragma unroll
            for (int q = 0; q < BATCH; q++)
                if ((uint32_t)column_of_lane < counts[q]) out_val[to[q] + column_of_lane] = __uint_as_float(staged[q]);
        }
        __builtin_amdgcn_wave_barrier();
        t_out += MVP_CLOCK() - t_rows;
    }
#ifdef MDB_MVP_TIMING
    if (lane == 0) {
        atomicAdd(&g_mvp_timing[0], MVP_CLOCK() - t_kernel);
        atomicAdd(&g_mvp_timing[1], t_decode);
        atomicAdd(&g_mvp_timing[2], t_top_up);
        atomicAdd(&g_mvp_timing[3], n_top_up);
        atomicAdd(&g_mvp_timing[4], t_out);
        atomicAdd(&g_mvp_timing[5], 1ull);
    }
#endif
}

// ---- the same cursors for SUM: macaque_v::sum adds a stream's values one after the other in f32 --------------
//
// f32 addition does not associate, so the additions of one stream stay a chain - but 64 chains fit into a wave, and
// the decoding, a hundred times the work, is again one lane per piece: k_agg_mv_pieces decodes every piece of the
// batch into scratch (piece p at values[64 p ..]: a stream's values follow each other), with the seeds sum() uses
// (models/mod.rs:145-181: the model's last DECODED value, NaN behind a MacaqueV model), and k_agg_mv_chains adds
// every stream up with one lane. stream_sums[2 i] = the sum of MacaqueV segment i's values (macaque_v.rs:228-235:
// it starts AS the first value), [2 i + 1] = the sum of segment i's residual tail.
struct ChainItem { // 16 bytes: a stream of two pieces or more, listed by the wave of k_agg_mv_pieces it begins in
    uint32_t segment_and_kind; // segment << 1 | 1 for its residual tail
    uint32_t n;                // its values, if it ends in the wave it begins in (0: it goes on - its segment's analysis knows)
    unsigned long long first_piece;
};

// Is piece `piece` (of segment i's values or of its tail) the first / the last of its stream? (The pieces of a stream
// follow each other in the cursor index.)
__device__ __forceinline__ void piece_neighbours(const MvCursor *__restrict__ cursors, unsigned long long piece, unsigned long long n_pieces,
                                                 uint32_t i, bool residual, bool *is_head, bool *is_tail) {
    if (piece == 0) {
        *is_head = true;
    } else {
        const MvCursor *before = cursors + piece - 1;
        *is_head = load_global(&before->segment) != i || ((load_global(&before->window) & MV_WINDOW_RESIDUAL) != 0) != residual;
    }
    if (piece + 1 == n_pieces) {
        *is_tail = true;
    } else {
        const MvCursor *behind = cursors + piece + 1;
        *is_tail = load_global(&behind->segment) != i || ((load_global(&behind->window) & MV_WINDOW_RESIDUAL) != 0) != residual;
    }
}

// Which lanes of a wave of pieces list a stream - the first pieces of streams of two pieces or more -, of which kind
// (short: up to CHAIN_SHORT_VALUES values; long: more, or going on behind the wave, where only the segment's analysis
// knows how many), with how many values, and the lane's place among the wave's listed streams of its kind.
constexpr uint32_t CHAIN_SHORT_VALUES = 1024;
struct ChainListing {
    bool lists, is_long;
    uint32_t n;                // values of the stream; 0: it goes on behind the wave
    uint32_t rank;             // among the wave's listed streams of the same kind
    uint32_t n_short, n_long;  // of the wave
};
__device__ __forceinline__ ChainListing chain_listing(bool present, bool is_head, bool is_tail, uint32_t to_decode, int lane) {
    ChainListing out;
    out.lists = present && is_head && !is_tail;
    const unsigned long long tails = __ballot(present && is_tail);
    const unsigned long long tails_behind = lane == 63 ? 0ull : (tails >> (lane + 1));
    const int my_tail = tails_behind ? lane + 1 + __builtin_ctzll(tails_behind) : -1; // (none: the stream goes on behind the wave)
    const uint32_t last_count = (uint32_t)__shfl((int)to_decode, my_tail >= 0 ? my_tail : lane, MDB_WAVE);
    out.n = my_tail >= 0 ? (uint32_t)(my_tail - lane) * MV_PIECE_VALUES + last_count : 0u; // (64 a piece but the last)
    out.is_long = out.n == 0u || out.n > CHAIN_SHORT_VALUES;
    const unsigned long long shorts = __ballot(out.lists && !out.is_long), longs = __ballot(out.lists && out.is_long);
    const unsigned long long below = (1ull << lane) - 1ull;
    out.rank = (uint32_t)__popcll((out.is_long ? longs : shorts) & below);
    out.n_short = (uint32_t)__popcll(shorts);
    out.n_long = (uint32_t)__popcll(longs);
    return out;
}

// How many streams of either kind every wave of pieces lists: long << 32 | short (the scan over these says where).
__global__ __launch_bounds__(MDB_WAVE) void k_agg_mv_chain_count(const MvCursor *__restrict__ cursors, unsigned long long n_pieces,
                                                                 unsigned long long *__restrict__ counts) {
    const int lane = threadIdx.x;
    const unsigned long long piece = (unsigned long long)blockIdx.x * MDB_WAVE + lane;
    bool is_head = false, is_tail = false;
    uint32_t to_decode = 0;
    if (piece < n_pieces) {
        const uint4 c0 = load_global(reinterpret_cast<const uint4 *>(cursors + piece));
        const uint4 c1 = load_global(reinterpret_cast<const uint4 *>(cursors + piece) + 1);
        to_decode = c1.x;
        piece_neighbours(cursors, piece, n_pieces, c0.z, (c1.y & MV_WINDOW_RESIDUAL) != 0, &is_head, &is_tail);
    }
    const ChainListing listing = chain_listing(piece < n_pieces, is_head, is_tail, to_decode, lane);
    if (lane == 0) counts[blockIdx.x] = ((unsigned long long)listing.n_long << 32) | listing.n_short;
}
struct ChainCount {
    const unsigned long long *counts;
    __device__ uint64_t operator()(uint64_t wave) const { return counts[wave]; }
};

// One lane per piece, as k_grid_mv_pieces (the same decoder, rounds of ROUND values staged in LDS): a piece that is a
// whole stream - most residual tails are - is added up by its own lane while it is decoded (its values ARE the stream,
// in order); the pieces of longer streams go to `values` (piece p at 64 p, rows of ROUND consecutive values per store)
// and the stream is listed for k_agg_mv_chain_groups by the lane of its first piece. (Round 5's kernel staged all 64
// values of every piece - 22 KB of LDS a wave, 1.75 waves a SIMD - so that the lane of a stream's first piece could add
// up the stream inside the wave while the other 63 waited: 2.7 / 2.3 ms where this decoder needs 2.0 / 1.45 for grid().)
template <int ROUND>
__global__ __launch_bounds__(MDB_WAVE) void k_agg_mv_pieces(DevSegments s, const MvCursor *__restrict__ cursors,
                                                            unsigned long long n_pieces, uint32_t *__restrict__ values,
                                                            float *__restrict__ stream_sums, ChainItem *__restrict__ short_items,
                                                            ChainItem *__restrict__ long_items,
                                                            const unsigned long long *__restrict__ chain_offsets) {
    constexpr int STRIDE = ROUND + 1; // (a row per lane: an odd stride keeps the banks apart)
    __shared__ uint32_t stage[MDB_WAVE * STRIDE];
    __shared__ uint32_t ring[PIECE_RING_ROWS][MDB_WAVE];
    __shared__ uint32_t row_count[MDB_WAVE];
    const int lane = threadIdx.x;
    const unsigned long long first_piece = (unsigned long long)blockIdx.x * MDB_WAVE;
    const unsigned long long piece = first_piece + lane;
    const uint8_t *values_first = first_buffer(s.values), *residuals_first = first_buffer(s.residuals); // (see view_data())
    uint32_t to_decode = 0, segment = 0xffffffffu;
    bool residual = false, is_head = false, is_tail = false;
    PieceReader reader;
    PieceState state;
    reader.idle(cursors);
    state.last = 0; state.trailing = 0; state.window_bits = 0; state.raw = false;
    if (piece < n_pieces) {
        const uint4 c0 = load_global(reinterpret_cast<const uint4 *>(cursors + piece));
        const uint4 c1 = load_global(reinterpret_cast<const uint4 *>(cursors + piece) + 1);
        const uint32_t i = c0.z, window = c1.y;
        segment = i;
        to_decode = c1.x;
        residual = (window & MV_WINDOW_RESIDUAL) != 0;
        piece_neighbours(cursors, piece, n_pieces, i, residual, &is_head, &is_tail);
        const DevCol &column = residual ? s.residuals : s.values;
        const uint4 view = column.views[i];
        reader.open(view_data(column, i, view, residual ? residuals_first : values_first),
                    residual ? (uint64_t)view.x - 1u : (uint64_t)view.x, c0.x);
        uint32_t seed = 0;
        if (residual) {
            const int32_t type = s.model_type_id[i];
            if (type == MDB_PMC_MEAN_ID) {
                float value = 0.0f;
                (void)decode_pmc_value(s.values.views[i], s.min_value[i], s.max_value[i], &value);
                seed = __float_as_uint(value);
            } else if (type == MDB_SWING_ID) {
                float first = 0.0f, last = 0.0f;
                (void)decode_swing_values(s.values.views[i], s.min_value[i], s.max_value[i], &first, &last);
                seed = __float_as_uint(last);
            } else {
                seed = 0x7fc00000u; // f32::NAN (models/mod.rs:167)
            }
        }
        state.last = seed ^ c0.y;
        const uint32_t leading = window & 255u, trailing = (window >> 8) & 255u;
        state.trailing = trailing & 31u;
        state.window_bits = leading + trailing <= 32u ? 32u - leading - trailing : 0u;
        state.raw = (window & MV_WINDOW_RAW) != 0;
    }
    const bool present = piece < n_pieces;
    const bool whole = present && is_head && is_tail; // the piece is its stream: summed here
    const bool spilled = present && !whole;           // a piece of a longer stream: its values go to memory
    // The streams of two pieces or more that begin in this wave, listed where the scan says (k_agg_mv_chain_count made
    // the same tests): the short ones in one list, the long ones in another.
    {
        const ChainListing listing = chain_listing(present, is_head, is_tail, to_decode, lane);
        if (listing.lists) {
            const unsigned long long where = chain_offsets[blockIdx.x];
            ChainItem *to = listing.is_long ? long_items + (where >> 32) : short_items + (where & 0xffffffffull);
            to[listing.rank] = {(segment << 1) | (residual ? 1u : 0u), listing.n, piece};
        }
    }
    reader.begin();
    reader.top_up(ring, lane);
    reader.top_up(ring, lane);
    reader.start(ring, lane);
    // macaque_v.rs:220-265: a segment's values are added to the first one, a tail's to 0, one after the other.
    float own = 0.0f;
    const bool starts_as_first = !residual;
    for (uint32_t done = 0; __any(done < to_decode); done += ROUND) {
        const uint32_t mine = done < to_decode ? min(to_decode - done, (uint32_t)ROUND) : 0u;
        static_assert(ROUND % 2 == 0, "values are decoded in pairs");
        for (uint32_t k = 0, most = wave_max_u32(mine); k < most; k += 2) {
            if (__any(reader.hungry())) reader.top_up(ring, lane);
            const uint32_t even = piece_decode_value(reader, state, ring, lane);
            const uint32_t odd = piece_decode_value(reader, state, ring, lane);
            stage[lane * STRIDE + k] = even;
            stage[lane * STRIDE + k + 1] = odd; // (k + 1 == ROUND: the row's spare word)
            // (the running sum of the lane's own piece: what a piece that is a whole stream reports)
            const float with_even = (starts_as_first && done + k == 0u) ? __uint_as_float(even) : own + __uint_as_float(even);
            own = k < mine ? with_even : own;
            own = k + 1 < mine ? own + __uint_as_float(odd) : own;
        }
        // Row r = this round's values of lane r's piece, for the pieces that go to memory: consecutive values,
        // (64 / ROUND) rows per store instruction.
        row_count[lane] = spilled ? mine : 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        constexpr int ROWS_PER_STORE = MDB_WAVE / ROUND;
        const int sub_row = lane / ROUND, column_of_lane = lane % ROUND;
        constexpr int BATCH = 8;
        static_assert(MDB_WAVE % (BATCH * ROWS_PER_STORE) == 0, "whole batches of stores");
        for (int r0 = 0; r0 < MDB_WAVE; r0 += BATCH * ROWS_PER_STORE) {
            uint32_t counts[BATCH], staged[BATCH];
#pragma unroll
            for (int q = 0; q < BATCH; q++) {
                const int r = r0 + q * ROWS_PER_STORE + sub_row;
                counts[q] = row_count[r];
                staged[q] = stage[r * STRIDE + column_of_lane];
            }
#pragma unroll
            for (int q = 0; q < BATCH; q++) {
                const int r = r0 + q * ROWS_PER_STORE + sub_row;
                if ((uint32_t)column_of_lane < counts[q])
                    values[(first_piece + (unsigned long long)r) * MV_PIECE_VALUES + done + (uint32_t)column_of_lane] = staged[q];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (whole && to_decode > 0) stream_sums[2ull * segment + (residual ? 1u : 0u)] = own;
}

// A stream of two pieces or more: its values lie in `values` (piece p at 64 p), and its f32 additions are one chain,
// in stream order (macaque_v.rs:228-235). A chain is 4 cycles an addition; what it waited for was memory - a lane of
// its own kept 128 bytes of its stream in flight, a round trip per 32 additions (round 6's counters: 65 % of the
// kernel's wave-cycles waiting, the vector ALU busy 5 %; a 65 536-value stream 0.7 ms) - and, in front of that, a thread
// per SEGMENT that looked for the streams not summed yet (0.75 ms for 8.5 M segments). Now the waves of k_agg_mv_pieces
// list the streams that begin in them (k_agg_mv_chain_count + a scan say where), and k_agg_mv_chain_groups gives every
// listed stream EIGHT lanes: together they keep 2 KB of it in flight - a round of 512 values, 16 chunks of 16 bytes per
// lane, the eight lanes' chunks side by side in memory - park a round in LDS, ask for the next one, and the first of
// the eight adds the parked round up, value after value.
// Two sizes of round: most listed streams are a few pieces long (a stretch of rejected points between two models) and
// want many waves in flight more than bytes - 4 loads a lane, 128 values a round, 4 KB of LDS a wave; the long ones
// (a whole chunk of noise is one stream of 65 536 values) want the bytes - 16 loads a lane, 512 values a round. A list
// of its own for each kind (a stream that goes on behind its wave counts as long): the long chains begin at once
// instead of behind the dispatch of the short ones' hundred thousand waves.
constexpr int CHAIN_GROUP_LANES = 8;
constexpr int CHAIN_GROUPS_PER_WAVE = MDB_WAVE / CHAIN_GROUP_LANES;

// Cursors left by host threads (the index of one call): should they ever disagree with the kernels' own analysis about
// a segment's streams, its sums are made unusable rather than a little wrong. (A resident batch's cursors come from
// that same analysis, k_mv_index_walk: nothing to compare.)
__global__ __launch_bounds__(256) void k_agg_mv_check_cursors(DevSegments s, const uint32_t *__restrict__ known_totals,
                                                              const unsigned long long *__restrict__ piece_base,
                                                              float *__restrict__ stream_sums) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.n) return;
    const unsigned long long first_piece = piece_base[i];
    if (piece_base[i + 1] == first_piece) return; // (no stream: nobody reads this segment's sums)
    uint32_t n_values, n_res, n_model, error;
    mv_stream_lengths(s, i, known_totals, &n_values, &n_res, &n_model, &error);
    if ((unsigned long long)((n_values + MV_PIECE_VALUES - 1) / MV_PIECE_VALUES + (n_res + MV_PIECE_VALUES - 1) / MV_PIECE_VALUES) !=
        piece_base[i + 1] - first_piece)
        stream_sums[2 * i] = stream_sums[2 * i + 1] = __uint_as_float(0x7fc00000u);
}

template <int CHAIN_LOADS> // 16-byte loads a lane has in flight: 4 for the list of short streams, 16 for the long ones
__global__ __launch_bounds__(MDB_WAVE) void k_agg_mv_chain_groups(DevSegments s, const uint32_t *__restrict__ known_totals,
                                                                  const uint32_t *__restrict__ values, float *__restrict__ stream_sums,
                                                                  const ChainItem *__restrict__ items, unsigned int n_items) {
    constexpr int CHAIN_ROUND_CHUNKS = CHAIN_LOADS * CHAIN_GROUP_LANES; // chunks of 16 bytes = 4 values a round
    constexpr int CHAIN_GROUP_STRIDE = CHAIN_ROUND_CHUNKS + 1;          // (in chunks: the eight adding lanes read eight banks)
    __shared__ uint4 parked[CHAIN_GROUPS_PER_WAVE * CHAIN_GROUP_STRIDE];
    const int lane = threadIdx.x, group = lane / CHAIN_GROUP_LANES, member = lane % CHAIN_GROUP_LANES;
    const unsigned int mine = blockIdx.x * CHAIN_GROUPS_PER_WAVE + (unsigned int)group;
    const bool listed = mine < n_items;
    ChainItem item{0u, 1u, 0ull}; // (a group without a stream: one value of the scratch's first piece, nobody's sum)
    if (listed) item = items[mine];
    // How many values the stream has: its pieces said so if it ended in the wave it began in, else the analysis of
    // its segment, by the group's first lane.
    uint32_t n = item.n;
    const bool tail = (item.segment_and_kind & 1u) != 0u;
    if (listed && member == 0 && n == 0) {
        uint32_t n_values, n_res, n_model, error;
        mv_stream_lengths(s, item.segment_and_kind >> 1, known_totals, &n_values, &n_res, &n_model, &error);
        n = tail ? n_res : n_values;
    }
    n = (uint32_t)__shfl((int)n, group * CHAIN_GROUP_LANES, MDB_WAVE);
    const uint32_t n_chunks = (n + 3u) >> 2;
    const uint4 *__restrict__ from = reinterpret_cast<const uint4 *>(values + item.first_piece * MV_PIECE_VALUES);
    uint4 *mine_parked = parked + group * CHAIN_GROUP_STRIDE;
    auto ask = [&](uint32_t round, uint4 (&into)[CHAIN_LOADS]) { // chunk j * 8 + member of the round: the group's lanes side by side
#pragma unroll
        for (int j = 0; j < CHAIN_LOADS; j++) {
            const uint32_t chunk = round * CHAIN_ROUND_CHUNKS + (uint32_t)(j * CHAIN_GROUP_LANES + member);
            into[j] = chunk < n_chunks ? load_global(from + chunk) : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    const uint32_t rounds = (n_chunks + CHAIN_ROUND_CHUNKS - 1) / CHAIN_ROUND_CHUNKS;
    const uint32_t most_rounds = wave_max_u32(rounds);
    // macaque_v.rs:228-235: the sum of a MacaqueV segment's values starts AS the first of them, a tail's at zero.
    const bool starts_as_first = !tail;
    float sum = 0.0f;
    uint4 asked[CHAIN_LOADS];
    ask(0, asked);
    for (uint32_t round = 0; round < most_rounds; round++) {
#pragma unroll
        for (int j = 0; j < CHAIN_LOADS; j++) mine_parked[j * CHAIN_GROUP_LANES + member] = asked[j];
        wave_sync();
        if (round + 1 < most_rounds) ask(round + 1, asked); // (under way while the round that is parked is added up)
        if (member == 0 && round < rounds) {
            const uint32_t first_value = round * (uint32_t)(4 * CHAIN_ROUND_CHUNKS);
            const uint32_t here = min(n - first_value, (uint32_t)(4 * CHAIN_ROUND_CHUNKS)); // values of this round
            uint32_t k = 0;
            if (round == 0 && starts_as_first) {
                const uint4 q = mine_parked[0];
                sum = __uint_as_float(q.x);
                if (here > 1) sum += __uint_as_float(q.y);
                if (here > 2) sum += __uint_as_float(q.z);
                if (here > 3) sum += __uint_as_float(q.w);
                k = 4;
            }
            for (; k + 32 <= here; k += 32) { // (eight reads of the parked round under way, then the chain of additions;
                                              // the next batch's reads under way during the additions: slower, 0.77 -> 0.92 ms)
                uint4 q[8];
#pragma unroll
                for (int j = 0; j < 8; j++) q[j] = mine_parked[(k >> 2) + (uint32_t)j];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    sum += __uint_as_float(q[j].x);
                    sum += __uint_as_float(q[j].y);
                    sum += __uint_as_float(q[j].z);
                    sum += __uint_as_float(q[j].w);
                }
            }
            for (; k + 4 <= here; k += 4) {
                const uint4 q = mine_parked[k >> 2];
                sum += __uint_as_float(q.x);
                sum += __uint_as_float(q.y);
                sum += __uint_as_float(q.z);
                sum += __uint_as_float(q.w);
            }
            if (k < here) { // (the stream's last, partial chunk)
                const uint4 q = mine_parked[k >> 2];
                sum += __uint_as_float(q.x);
                if (k + 1 < here) sum += __uint_as_float(q.y);
                if (k + 2 < here) sum += __uint_as_float(q.z);
            }
        }
        wave_sync(); // (the next round overwrites what was parked)
    }
    if (member == 0 && listed && n > 0) stream_sums[2ull * (item.segment_and_kind >> 1) + (tail ? 1u : 0u)] = sum;
}

// ---- the order in which k_grid_ts_count takes the streams ----------------------------------------------------
//
// A wave of k_grid_ts_count walks 64 streams in lockstep and is done when the longest of them is, and the
// segments of a batch are not equally long - a model lasts as long as the signal lets it. So the streams are
// dealt to the waves by length, the longest first: a counting sort of the segments with irregular timestamps
// by the bytes of their streams, in 16-byte classes (10^9 randomly spaced points in 1.4 M streams: the walk
// 3.9 -> 3.4 ms, the sort 0.05 ms; segments with regular timestamps take no lane at all).

constexpr int TS_SORT_CLASSES = 1024;
constexpr int TS_SORT_ITEMS = 4;

__device__ __forceinline__ bool ts_sort_class(const DevSegments &s, uint64_t i, const TimeRange &range, uint32_t *sort_class) {
    const uint4 view = s.timestamps.views[i];
    const int32_t length = (int32_t)view.x;
    if (!(length > 0 && (view_inline_byte(view, 0) & 0x80u) != 0)) return false;
    // (with a time range: only the segments that reach into it - start_time / end_time say which)
    if (range.enabled && (s.end_time[i] < range.lo || s.start_time[i] > range.hi)) return false;
    *sort_class = (uint32_t)(TS_SORT_CLASSES - 1) - min((uint32_t)length >> 4, (uint32_t)(TS_SORT_CLASSES - 1));
    return true;
}

// SCATTER = false: how many streams there are of every class (counts += ...). SCATTER = true: `counts` holds
// where each class begins in `order` (k_ts_sort_scan) and is advanced by what the block places.
template <bool SCATTER>
__global__ __launch_bounds__(PREPASS_THREADS) void k_ts_sort(DevSegments s, TimeRange range, uint32_t *__restrict__ counts,
                                                             uint32_t *__restrict__ order) {
    __shared__ uint32_t local[TS_SORT_CLASSES]; // of this block's streams; then where they begin in `order`
    for (int c = threadIdx.x; c < TS_SORT_CLASSES; c += PREPASS_THREADS) local[c] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * (PREPASS_THREADS * TS_SORT_ITEMS);
    uint32_t sort_class[TS_SORT_ITEMS], rank[TS_SORT_ITEMS];
    bool irregular[TS_SORT_ITEMS];
#pragma unroll
    for (int k = 0; k < TS_SORT_ITEMS; k++) {
        const uint64_t i = base + (uint64_t)k * PREPASS_THREADS + threadIdx.x;
        irregular[k] = i < s.n && ts_sort_class(s, i, range, &sort_class[k]);
        if (irregular[k]) rank[k] = atomicAdd(&local[sort_class[k]], 1u);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < TS_SORT_CLASSES; c += PREPASS_THREADS) {
        const uint32_t mine = local[c];
        if (mine) local[c] = atomicAdd(&counts[c], mine);
    }
    if (!SCATTER) return;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TS_SORT_ITEMS; k++)
        if (irregular[k]) order[local[sort_class[k]] + rank[k]] = (uint32_t)(base + (uint64_t)k * PREPASS_THREADS + threadIdx.x);
}

// counts -> where each class begins; counts[TS_SORT_CLASSES] = how many streams there are.
__global__ __launch_bounds__(TS_SORT_CLASSES) void k_ts_sort_scan(uint32_t *__restrict__ counts) {
    __shared__ uint64_t lds[17];
    uint64_t total;
    const uint64_t begins = block_exclusive_scan_u64(counts[threadIdx.x], lds, &total);
    counts[threadIdx.x] = (uint32_t)begins;
    if (threadIdx.x == 0) counts[TS_SORT_CLASSES] = (uint32_t)total;
}

// ---- k_grid_ts_count: the one sequential walk over every delta-of-delta stream ---------------------------
//
// len() of a segment with irregular timestamps is the number of codes of its stream (models/mod.rs:
// 98-124), so every such stream has to be walked once before anything can be placed. One lane per
// segment, one wave per workgroup, the streams read through the LDS ring of k_grid_serial (a wave
// that walks 64 streams needs a word of some stream in nearly every step: fetched one by one that is a
// trip to memory per step, fetched 24 words per lane at a time it is one per seventy). On the way the
// lane leaves the cursor in front of the first code of every 256-bit piece (TsCursor): from there on
// the pieces are independent. The last 80 bits of a stream, where running out of bits has a meaning,
// are left to the careful decoder. The step carries nothing it does not need: no per-point callback,
// one branch for the rare 32- and 64-bit codes, one for a run of `0` codes (counted and skipped at
// once: a series sampled at a fixed rate with the odd gap is such runs almost entirely), one for the
// checkpoint.
// `order` (or nullptr: the segments as they come): the segments with irregular timestamps, n_order of them.
// WALK (for the aggregates, mdb_agg.hip) - WALK_SUMS: the walk also adds up what swing::sum adds up for a Swing
// segment without residuals - (slope * t + intercept) of every timestamp, in f64, in the order of the points
// (swing.rs:283-299; the line through the SEGMENT's end points) - and leaves it in sums[i].
// WALK_RANGE: it aggregates, for a PMC-Mean or Swing segment without residuals, the values grid() would produce
// for the points with range.lo <= timestamp <= range.hi, the way GridExec + filter + aggregate would (f32 values
// added up in f64 in the order of the points, their count, minimum and maximum) into ranges[i]. In both, `header`
// is a plain error word.
enum : int { WALK_GRID = 0, WALK_SUMS = 1, WALK_RANGE = 2 };

template <int WALK>
__global__ __launch_bounds__(SERIAL_THREADS) void k_grid_ts_count(DevSegments s, TsCheckpoints checkpoints,
                                                                 uint32_t *__restrict__ totals,
                                                                 GridHeader *__restrict__ header,
                                                                 const uint32_t *__restrict__ order, uint64_t n_order,
                                                                 double *__restrict__ sums, TimeRange range,
                                                                 TsWalkRange *__restrict__ ranges) {
    constexpr bool SUMS = WALK == WALK_SUMS;
    constexpr bool RANGE = WALK == WALK_RANGE;
    __shared__ uint32_t ring[SERIAL_RING_WORDS][MDB_WAVE];
    const int lane = threadIdx.x;
    const uint64_t at_order = (uint64_t)blockIdx.x * SERIAL_THREADS + lane;
    const uint64_t i = order ? (at_order < n_order ? (uint64_t)order[at_order] : s.n) : at_order;
    uint4 view = make_uint4(0u, 0u, 0u, 0u);
    if (i < s.n) view = s.timestamps.views[i];
    const int32_t length = (int32_t)view.x;
    const bool irregular = i < s.n && length > 0 && (view_inline_byte(view, 0) & 0x80u) != 0;
    if (!__any(irregular)) return;
    const uint8_t *bytes = irregular ? view_data(s.timestamps, i, view) : nullptr;
    const uint32_t nbytes = irregular ? (uint32_t)length : 0u;
    const bool keeps = irregular && checkpoints.piece_base != nullptr && ts_has_checkpoints(length);
    TsCursor *slots = keeps ? checkpoints.slots + checkpoints.piece_base[i] : nullptr;
    const uint32_t n_slots = keeps ? ts_pieces(nbytes) : 0u;
    TsCursor at = ts_stream_start(irregular ? s.start_time[i] : 0, bytes);
    // Cursors leave in pairs, 64 bytes that lie in one 64-byte block of memory: the first of a pair waits in
    // LDS for the second. (One by one they are 32-byte writes scattered over the 5 000 waves' 64 streams each -
    // more lines under way than the caches hold until their other halves arrive: the walk executed fewer
    // instructions than the one without cursors and took twice as long, 296 000 of a wave's 430 000 cycles
    // waiting to issue.)
    __shared__ __attribute__((aligned(16))) uint4 waiting[WALK == WALK_GRID ? 2 : 1][WALK == WALK_GRID ? MDB_WAVE : 1];
    bool one_waits = false;
    uint32_t waiting_piece = 0;
    const bool slots_begin_odd = keeps && (checkpoints.piece_base[i] & 1ull) != 0; // (in units of cursors)
    auto leave = [&](uint32_t at_piece, const TsCursor &cursor) {
        if (WALK != WALK_GRID) return;
        const bool second_of_pair = ((at_piece & 1u) != 0) != slots_begin_odd;
        const uint4 *halves = reinterpret_cast<const uint4 *>(&cursor);
        if (!second_of_pair) { // waits for its neighbour (or for the end of the walk)
            waiting[0][lane] = halves[0];
            waiting[1][lane] = halves[1];
            one_waits = true;
            waiting_piece = at_piece;
            return;
        }
        uint4 *out = reinterpret_cast<uint4 *>(slots + at_piece);
        if (one_waits) {
            out[-2] = waiting[0][lane];
            out[-1] = waiting[1][lane];
            one_waits = false;
        }
        out[0] = halves[0];
        out[1] = halves[1];
    };
    auto leave_the_waiting_one = [&]() {
        if (WALK != WALK_GRID || !one_waits) return;
        uint4 *out = reinterpret_cast<uint4 *>(slots + waiting_piece);
        out[0] = waiting[0][lane];
        out[1] = waiting[1][lane];
        one_waits = false;
    };
    if (keeps) leave(0u, at);
    uint32_t piece = 0;
    uint32_t error = 0;
    // (SUMS) the line of a Swing segment without residuals, and the sum of its values so far
    bool adds = false;
    LineDev line = LineDev{0.0, 0.0};
    double added = 0.0;
    if (SUMS && irregular && ts_walk_adds(s, i)) {
        float first = 0.0f, last = 0.0f;
        adds = decode_swing_values(s.values.views[i], s.min_value[i], s.max_value[i], &first, &last);
        line = line_through(at.timestamp, (double)first, s.end_time[i], (double)last);
        added = line.slope * (double)at.timestamp + line.intercept; // point 0 is the start time
    }
    // (RANGE) what the points inside the range add up to, and the value of a PMC-Mean segment
    TsWalkRange inside = TsWalkRange{0.0, 0ll, FLT_MAX, -FLT_MAX};
    bool is_swing = false;
    float pmc_value = 0.0f;
    auto visit = [&](int64_t t) { // (RANGE) one point of a segment the walk aggregates
        if (t < range.lo || t > range.hi) return;
        const float v = is_swing ? (float)(line.slope * (double)t + line.intercept) : pmc_value;
        inside.sum += (double)v;
        inside.count += 1;
        inside.min = min_num(inside.min, v);
        inside.max = max_num(inside.max, v);
    };
    if (RANGE && irregular && ts_walk_aggregates_range(s, i)) {
        is_swing = s.model_type_id[i] == MDB_SWING_ID;
        if (is_swing) {
            float first = 0.0f, last = 0.0f;
            adds = decode_swing_values(s.values.views[i], s.min_value[i], s.max_value[i], &first, &last);
            line = line_through(at.timestamp, (double)first, s.end_time[i], (double)last);
        } else {
            adds = decode_pmc_value(s.values.views[i], s.min_value[i], s.max_value[i], &pmc_value);
        }
        if (adds) visit(at.timestamp); // point 0 is the start time
    }
    // The stream's jumps (TsJump), as long as they are few: every point whose delta is not `base`.
    TsJump *jumps = keeps && checkpoints.jumps ? checkpoints.jumps + checkpoints.piece_base[i] * TS_JUMPS_PER_PIECE : nullptr;
    bool tracking = jumps != nullptr && checkpoints.piece_base[i] <= 0xffffffffull; // (TileDesc carries 32 bits of it)
    const uint32_t jump_room = min(n_slots * TS_JUMPS_PER_PIECE - 1u, TS_MAX_JUMPS);
    uint32_t n_jumps = 0;
    uint64_t base = 0, jumped = 0; // the delta nearly every point has; what the jumps so far add up to
    auto jump = [&](uint32_t position) { // (`jumped` has just changed, at this point)
        // More than the list holds, or than one point in sixteen: not a fixed rate with the odd gap.
        if (n_jumps >= jump_room || n_jumps * 16u > position + 64u) {
            tracking = false;
            return;
        }
        n_jumps += 1;
        jumps[n_jumps] = TsJump{position, 0u, (int64_t)jumped};
    };
    // (streams of 2^28 bytes and more have no checkpoints and go through the careful decoder whole)
    // The walk goes on as long as a short code (16 bits at most) lies inside the stream wherever it begins: a
    // code that lies inside the stream means what it means anywhere else (timestamps.rs:228-292: only a code the
    // stream ends in the middle of, or the ones that pad its last byte, are something else), a `0` inside it
    // is a point. A 37- or 69-bit code that does not fit any more is left to the careful decoder, with whatever
    // follows. (Stopping 80 bits in front of the end, the longest code and some, left some sixty codes of a
    // stream of `0` runs to the careful decoder: a third of what a wave executed.)
    const uint32_t total_bits = nbytes * 8u;
    const uint32_t fast_end = nbytes < (1u << 28) && total_bits >= 16u ? total_bits - 16u : 0u;
    RingBitReader reader;
    reader.begin(bytes, nbytes);
    bool active = irregular && at.bit <= fast_end;
    bool fresh = true;
    // Everything loaded so far has to have arrived before the loop, not in it: a value whose load is still on
    // its way when the loop is entered gets its wait (s_waitcnt vmcnt(0)) at the top of the loop, where it is met
    // again in every round - and there it also waits for the round before's stores of cursors to be acknowledged
    // by memory. (No measurable difference by itself - what the cursor stores cost is their scatter, see `waiting`
    // above - but a wait on stores in every round is nothing to leave lying around.)
    __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0)
    while (__any(active)) {
        if (__any(active && reader.hungry())) {
            ring_top_up(reader, ring, lane, active);
            // (a step that took three words off the ring - a 64-bit code - has looked ahead at a slot
            // that was not filled yet)
            reader.look_ahead(ring, lane);
        }
        if (active) {
            if (fresh) { // the flag "irregular" (timestamps.rs:116) is bit 0
                reader.look_ahead(ring, lane);
                reader.refill(ring, lane);
                reader.refill(ring, lane);
                reader.consume(1);
                fresh = false;
            }
            if ((at.bit >> 8) != piece) { // (every lane keeps count of its pieces: a run of `0` codes ends with one)
                piece = at.bit >> 8;
                if (keeps) leave(piece, at);
            }
            reader.refill(ring, lane);
            const uint32_t top = (uint32_t)(reader.buffer >> 32);
            uint32_t length_of_code;
            bool beyond = false; // a long code that does not lie inside the stream
            if (top < 0x08000000u) { // five `0` codes or more
                uint32_t run = top == 0u ? 32u : (uint32_t)__clz((int)top);
                run = min(run, ((piece + 1u) << 8) - at.bit); // (the next piece's cursor is met)
                run = min(run, total_bits - at.bit);          // (behind the stream's last bit the buffer is zeros)
                tracking = tracking && at.last_delta == base;  // (five jumps in a row are not the odd gap)
                if ((SUMS || RANGE) && adds) {
                    int64_t t = at.timestamp;
                    for (uint32_t k = 0; k < run; k++) {
                        t = (int64_t)((uint64_t)t + at.last_delta);
                        if (SUMS) added += line.slope * (double)t + line.intercept;
                        else visit(t);
                    }
                }
                at.timestamp = (int64_t)((uint64_t)at.timestamp + (uint64_t)run * at.last_delta);
                at.count += run;
                length_of_code = run;
            } else if (const uint32_t ones = (uint32_t)__clz((int)~top);
                       ones >= 4 && at.bit + (ones >= 5 ? 69u : 37u) > total_bits) {
                beyond = true;
                length_of_code = 0;
            } else {
                if (ones >= 4) {
                    reader.consume(5); // `11110` + 32 bits or `11111` + 64 bits
                    reader.refill(ring, lane);
                    uint64_t encoded = (uint32_t)(reader.buffer >> 32);
                    reader.consume(32);
                    at.bit += 37;
                    if (ones >= 5) {
                        reader.refill(ring, lane);
                        encoded = (encoded << 32) | (uint32_t)(reader.buffer >> 32);
                        reader.consume(32);
                        at.bit += 32;
                        at.last_delta += encoded;
                    } else {
                        at.last_delta += encoded > (1ull << 31) ? (encoded | (~0ull << 32)) : encoded;
                    }
                    length_of_code = 0;
                } else {
                    int32_t delta_of_delta;
                    length_of_code = ts_short_code(top, ones, &delta_of_delta);
                    at.last_delta += (uint64_t)(int64_t)delta_of_delta;
                }
                at.timestamp = (int64_t)((uint64_t)at.timestamp + at.last_delta);
                if (SUMS && adds) added += line.slope * (double)at.timestamp + line.intercept;
                if (RANGE && adds) visit(at.timestamp);
                if (tracking) {
                    if (at.count == 1u) {
                        base = at.last_delta; // the delta of the first code
                    } else if (at.last_delta != base) {
                        jumped += at.last_delta - base;
                        jump(at.count);
                    }
                }
                at.count += 1;
                // A second short code out of the same 32 bits, if it begins in the same piece (no cursor is due
                // in front of it) and nobody is noting jumps: what a step costs around the code itself - the
                // look at the ring, the refill, the checks - is then paid once for two codes. (Without a branch:
                // the lanes of a wave do not agree on it.)
                {
                    const uint32_t rest = top << (length_of_code & 31u);
                    const uint32_t ones_then = (uint32_t)__clz((int)~rest);
                    int32_t delta_of_delta_then;
                    const uint32_t both = length_of_code + ts_short_code(rest, min(ones_then, 3u), &delta_of_delta_then);
                    const uint32_t bit_then = at.bit + length_of_code;
                    const bool second = !tracking && length_of_code != 0u && ones_then < 4u && both <= 32u &&
                                        (bit_then >> 8) == (at.bit >> 8) && bit_then <= fast_end;
                    at.last_delta += second ? (uint64_t)(int64_t)delta_of_delta_then : 0ull;
                    at.timestamp = (int64_t)((uint64_t)at.timestamp + (second ? at.last_delta : 0ull));
                    if (SUMS && adds && second) added += line.slope * (double)at.timestamp + line.intercept;
                    if (RANGE && adds && second) visit(at.timestamp);
                    at.count += second ? 1u : 0u;
                    length_of_code = second ? both : length_of_code;
                }
            }
            reader.consume(length_of_code);
            at.bit += length_of_code;
            active = !beyond && at.bit <= fast_end;
        }
    }
    bool listed = false; // the segment has a jump list
    if (irregular) {
        leave_the_waiting_one();
        if (keeps && (at.bit >> 8) != piece && at.bit < nbytes * 8u) {
            piece = at.bit >> 8;
            slots[piece] = at;
        }
        // The last codes with the careful decoder (it meets the cursors of the pieces it enters itself).
        uint32_t last_piece = piece;
        bool finished = false;
        const int64_t start_time = s.start_time[i];
        at = decode_irregular_span(bytes, nbytes, s.end_time[i], 0xffffffffu, &error, at, 0xffffffffu, &finished,
                                   [&](uint32_t k, int64_t t) {
                                       // (the last point, which is end_time whatever the deltas say, included)
                                       if (SUMS && adds) added += line.slope * (double)t + line.intercept;
                                       if (RANGE && adds) visit(t);
                                       if (!tracking) return;
                                       const uint64_t now = (uint64_t)t - (uint64_t)start_time - (uint64_t)k * base;
                                       if (now == jumped) return;
                                       jumped = now;
                                       jump(k);
                                   },
                                   [&](const TsCursor &c) {
                                       if (!keeps) return;
                                       last_piece = c.bit / TS_PIECE_BITS;
                                       slots[last_piece] = c;
                                   });
        for (uint32_t k = last_piece + 1; k < n_slots; k++)
            slots[k] = TsCursor{nbytes * 8u, TS_NO_CODE, s.end_time[i], 0ull, bytes};
        totals[i] = at.count;
        if (SUMS && adds) sums[i] = added;
        if (RANGE && adds) ranges[i] = inside;
        // With a list the segment is k_grid_tiles' work: its pieces are marked as not to be decoded.
        listed = tracking && finished && !error;
        if (jumps) jumps[0] = TsJump{0u, listed ? n_jumps : TS_NO_JUMPS, (int64_t)base};
        if (keeps) {
            uint2 *owner = checkpoints.piece_segment + checkpoints.piece_base[i];
            for (uint32_t k = 0; k < n_slots; k++) owner[k] = make_uint2((uint32_t)i, nbytes | (listed ? TS_PIECE_LISTED : 0u));
        }
    }
    // Where most segments have jump lists, the pieces that are still to be decoded are few and far between:
    // k_grid_timestamps, one lane per piece, would find a lane or two of every wave with work. They are written
    // down, so that it can take 64 of them per wave. (Only by waves that have listed a segment: a batch of
    // randomly spaced timestamps does not pay for a list it has no use for. Whether the list is complete the
    // host sees by comparing GridHeader::live_pieces with the number of pieces the prepass counts.)
    if (WALK == WALK_GRID && checkpoints.live && __any(listed) && keeps && !listed) {
        const unsigned long long first_piece = checkpoints.piece_base[i];
        const unsigned long long at_list = atomicAdd(&header->live_pieces, (unsigned long long)n_slots);
        for (uint32_t k = 0; k < n_slots; k++) checkpoints.live[at_list + k] = (uint32_t)(first_piece + k);
    }
    if (error) atomicOr(WALK != WALK_GRID ? reinterpret_cast<unsigned int *>(header) : &header->error, error);
}

// ---- k_grid_timestamps: the second walk is not one -----------------------------------------------------------
//
// One lane per piece of a stream, from the cursor k_grid_ts_count left in front of the piece's first
// code: about 22 codes of a randomly sampled series, up to 256 of a fixed-rate series with gaps. How
// many codes it is the next piece's cursor says, so they are simply taken off the stream (the careful
// decoder has been over the last bits of every stream already). The bytes of the piece - at most 80,
// counted from the 16-byte boundary below its first one - are fetched at once and parked in LDS; the
// points go to LDS too, as 32-bit offsets from the first timestamp of the wave, because the 64 pieces
// of a wave are usually pieces of one stream that follow each other, and so do the points they decode
// to: the wave then writes them out together, every store instruction a contiguous run (a lane storing
// its own points one by one is a request per point to the memory system, and that is what bounds the
// kernel then). A wave whose pieces hold more points than the buffer has room for (long runs of `0`
// codes: hundreds of points per piece), whose points do not follow each other in the output or span
// more than 2^32 microseconds stores directly, two timestamps per store. (The buffer holds 2 048 points: at 1 536,
// half the waves of a series sampled at random intervals - 24.4 points per piece on average - did not fit,
// decoded their pieces into runs in vain and then stored directly: 7.9 instead of 5.0 ms per 10^9 points.)
// Swing values are (slope * t + intercept) of the timestamps just decoded (swing.rs:304-319, the line
// is in the descriptor), a PMC-Mean value does not depend on the timestamp, MacaqueV values and
// residuals are written later by k_grid_serial.
constexpr uint32_t TS_STAGE_POINTS = 2048;
constexpr uint32_t TS_MAX_RUNS = 2 * TS_STAGE_POINTS / (3 * MDB_WAVE); // arithmetic runs of a piece that fit into the wave's buffer (16)
constexpr int TS_THREADS = 128;
constexpr int TS_PIECE_CHUNKS = 4; // 16-byte chunks parked per piece: 127 + 255 + 69 + 32 bits at most

struct TsWaveArgs {
    DevSegments s;
    const TileDesc *desc;
    const unsigned long long *offsets;
    const uint32_t *irregular_totals, *irregular_first, *counts;
    TsCheckpoints checkpoints;
    uint64_t n_pieces;
    int64_t *out_ts;
    float *out_val;
    // (sparse flavour) waves it leaves to the general one: their numbers, and how many there are
    uint32_t *left_waves;
    unsigned int *n_left_waves;
    // the pieces to decode, if they are few (TsCheckpoints::live), or nullptr: all n_pieces
    const uint32_t *live;
    uint64_t n_live;
};

// The points a wave has staged in LDS (timestamps as 32-bit distances from base_time, and values) to their place in
// the output, wave_first onwards. Every store instruction of the wave a kilobyte in one piece, as in k_grid_tiles:
// two timestamps per lane, four values per lane; in front of and behind the aligned middle, one point per lane
// (one point per lane and instruction throughout: 4.1 instead of 3.9 ms per 10^9 randomly spaced points).
__device__ __forceinline__ void write_staged_points(int lane, uint64_t wave_first, uint32_t wave_total, int64_t base_time,
                                                    const uint32_t *my_ts, const float *my_val,
                                                    int64_t *__restrict__ out_ts, float *__restrict__ out_val) {
    auto timestamp_of = [&](uint32_t k) { return base_time + (int64_t)(uint64_t)my_ts[k]; };
    if (out_ts) {
        const uint32_t head = (uint32_t)(wave_first & 1ull) & (wave_total > 0 ? 1u : 0u); // points in front of the 16-byte boundary
        const uint32_t pairs = (wave_total - head) / 2u;
        if (lane == 0 && head) out_ts[wave_first] = timestamp_of(0u);
        longlong2 *out_pairs = reinterpret_cast<longlong2 *>(out_ts + wave_first + head);
        for (uint32_t pair = lane; pair < pairs; pair += MDB_WAVE)
            out_pairs[pair] = make_longlong2(timestamp_of(head + 2u * pair), timestamp_of(head + 2u * pair + 1u));
        if (lane == 0 && head + 2u * pairs < wave_total) out_ts[wave_first + wave_total - 1u] = timestamp_of(wave_total - 1u);
    }
    const uint32_t head = min((uint32_t)((0ull - wave_first) & 3ull), wave_total);
    const uint32_t quads = (wave_total - head) / 4u;
    if ((uint32_t)lane < head) out_val[wave_first + lane] = my_val[lane];
    float4 *out_quads = reinterpret_cast<float4 *>(out_val + wave_first + head);
    for (uint32_t quad = lane; quad < quads; quad += MDB_WAVE) {
        const uint32_t k = head + 4u * quad;
        out_quads[quad] = make_float4(my_val[k], my_val[k + 1u], my_val[k + 2u], my_val[k + 3u]);
    }
    const uint32_t done = head + 4u * quads;
    if (done + (uint32_t)lane < wave_total) out_val[wave_first + done + lane] = my_val[done + lane];
}

// The 64 pieces `wave_index * 64 ...` of the batch, by one wave. SPARSE: only the first of the three ways
// out below, as a loop without the other two's branches and stores (they are most of what a wave executes
// per code otherwise); a wave that needs another way is noted in args.left_waves for the general flavour.
template <bool SPARSE>
__device__ __forceinline__ void ts_wave(const TsWaveArgs &args, uint64_t wave_index, uint32_t *stage_words,
                                        uint4 (*parked_chunks)[MDB_WAVE]) {
    const DevSegments &s = args.s;
    const TileDesc *__restrict__ desc = args.desc;
    const unsigned long long *__restrict__ offsets = args.offsets;
    const uint32_t *__restrict__ irregular_totals = args.irregular_totals;
    const uint32_t *__restrict__ irregular_first = args.irregular_first;
    const uint32_t *__restrict__ counts = args.counts;
    const TsCheckpoints &checkpoints = args.checkpoints;
    const uint64_t n_pieces = args.n_pieces;
    int64_t *__restrict__ out_ts = args.out_ts;
    float *__restrict__ out_val = args.out_val;
    const int lane = threadIdx.x % MDB_WAVE;
    // (with a list of the pieces that are to be decoded: the wave's 64 of those)
    uint64_t slot = wave_index * MDB_WAVE + (uint64_t)lane;
    if (args.live) slot = slot < args.n_live ? (uint64_t)args.live[slot] : n_pieces;
    // What the lane needs to know arrives in two rounds of loads: the piece's cursor (with the address
    // of the stream), its segment and the next piece's; then everything about the segment, together
    // with the bytes of the piece.
    TsCursor from = TsCursor{0u, TS_NO_CODE, 0, 0ull, nullptr};
    uint32_t i = 0, stream_bytes = 0, next_count = TS_NO_CODE;
    bool next_is_mine = false; // the next piece belongs to the same stream
    // (the pieces of a segment with a jump list are not decoded: k_grid_tiles writes its points)
    // (all four loads at once, whether the piece is to be decoded or not: one trip to memory, not two)
    uint2 owner = make_uint2(0u, TS_PIECE_LISTED);
    uint32_t next_owner = 0xffffffffu;
    if (slot < n_pieces) {
        owner = checkpoints.piece_segment[slot];
        from = checkpoints.slots[slot];
        if (slot + 1 < n_pieces) {
            next_owner = checkpoints.piece_segment[slot + 1].x;
            next_count = checkpoints.slots[slot + 1].count;
        }
    }
    const bool present = !(owner.y & TS_PIECE_LISTED);
    if (!__any(present)) return;
    if (present) {
        i = owner.x;
        stream_bytes = owner.y;
        next_is_mine = next_owner == i;
    } else {
        from = TsCursor{0u, TS_NO_CODE, 0, 0ull, nullptr};
        next_count = TS_NO_CODE;
    }
    const bool has_code = from.count != TS_NO_CODE;
    const uint32_t piece = from.bit >> 8; // (a piece without a code is never the first of its stream)
    const uint32_t visible = present ? counts[i] & COUNT_MASK : 0u;
    const uint32_t first = present ? irregular_first[i] : 0u; // index of the first visible point
    const uint32_t visible_end = first + visible;              // (and behind the last)
    const uint32_t n_total = present ? irregular_totals[i] : 0u;
    const int64_t end_time = present ? s.end_time[i] : 0;
    // The bytes of the piece: at most 16-byte chunks 0..3 from the boundary below its first code.
    uint32_t first_bit = 0; // of the piece's first code, counted from the first parked chunk
    uint4 fetched[TS_PIECE_CHUNKS];
    if (present && has_code) {
        const uintptr_t address = reinterpret_cast<uintptr_t>(from.stream) + (from.bit >> 3);
        const uint4 *chunk = reinterpret_cast<const uint4 *>(address & ~(uintptr_t)15u);
        // (a chunk that holds a byte of the stream never crosses a page; behind the last one it repeats)
        const uintptr_t last_byte = reinterpret_cast<uintptr_t>(from.stream) + stream_bytes - 1;
        const uint4 *last_chunk = reinterpret_cast<const uint4 *>(last_byte & ~(uintptr_t)15u);
        first_bit = (uint32_t)(address & 15u) * 8u + (from.bit & 7u);
#pragma unroll
        for (int c = 0; c < TS_PIECE_CHUNKS; c++) fetched[c] = load_global(min(chunk + c, last_chunk));
    }
    // The points this lane decodes: from the one its first code produces (piece 0: from point 0, the
    // start time) up to where the next piece takes over, or to the end of the segment.
    uint32_t run_first = 0, run_end = 0;
    const bool last_of_stream = !next_is_mine || next_count == TS_NO_CODE; // no code starts in a later piece
    if (present && visible > 0 && has_code) {
        run_first = piece == 0 ? 0u : from.count;
        run_end = last_of_stream ? n_total : next_count;
        run_first = max(run_first, first);
        run_end = min(run_end, visible_end);
        if (run_end < run_first) run_end = run_first;
    }
    const uint32_t mine = run_end - run_first;
    const uint64_t o = present ? (uint64_t)offsets[i] : 0;
    const uint64_t out_first = o + (run_first - first);
    // Where the lane's points go inside the wave's: an exclusive prefix sum over the lanes.
    uint32_t before = mine;
#pragma unroll
    for (int delta = 1; delta < MDB_WAVE; delta <<= 1) {
        const uint32_t up = __shfl_up(before, delta, MDB_WAVE);
        if (lane >= delta) before += up;
    }
    const uint32_t wave_total = __shfl(before, MDB_WAVE - 1, MDB_WAVE);
    before -= mine;
    if (wave_total == 0) return;

    // The bytes of the piece, parked in LDS: chunk c of the lane is parked[wave][c][lane].
    if (present && has_code) {
#pragma unroll
        for (int c = 0; c < TS_PIECE_CHUNKS; c++) parked_chunks[c][lane] = fetched[c];
    }
    const uint32_t *my_words = reinterpret_cast<const uint32_t *>(&parked_chunks[0][0]);
    auto word = [&](uint32_t k) { // word k of the lane's parked bytes, most significant byte first
        k = min(k, (uint32_t)(4 * TS_PIECE_CHUNKS - 1));
        return __builtin_bswap32(my_words[((k >> 2) * MDB_WAVE + lane) * 4 + (k & 3u)]);
    };

    // All the lanes' runs one after the other in the output, and close enough in time?
    const unsigned long long producing = __ballot(mine > 0);
    const int leader = __ffsll((long long)producing) - 1;
    const uint64_t wave_first = __shfl((uint32_t)(out_first >> 32), leader, MDB_WAVE) * 0x100000000ull +
                                __shfl((uint32_t)out_first, leader, MDB_WAVE);
    const int64_t lane_base = from.timestamp; // (piece 0: the start time)
    const int64_t base_time = (int64_t)(__shfl((uint32_t)((uint64_t)lane_base >> 32), leader, MDB_WAVE) * 0x100000000ull +
                                        __shfl((uint32_t)lane_base, leader, MDB_WAVE));
    const bool consecutive = !__any(mine > 0 && out_first != wave_first + before);
    bool staged = wave_total <= TS_STAGE_POINTS && consecutive;

    const TileDesc d = present ? desc[i] : TileDesc{};
    const uint32_t type = d.flags & FLAG_TYPE_MASK;
    // d.n_model counts the VISIBLE points the model stands for; they are the first ones.
    const uint32_t model_end = first + d.n_model;
    uint32_t *my_ts = stage_words;
    float *my_val = reinterpret_cast<float *>(stage_words + TS_STAGE_POINTS);
    if constexpr (SPARSE) {
        // The wave's points as one contiguous run, or not at all here.
        bool leaves = !staged;
        if (staged) {
            bool too_far = false;
            auto emit = [&](uint32_t k, int64_t t) {
                if (k < run_first || k >= run_end) return;
                float value = 0.0f;
                if (k < model_end) value = type == MDB_SWING_ID ? (float)(d.slope * (double)t + d.intercept) : d.value;
                const uint64_t distance = (uint64_t)(t - base_time);
                too_far |= distance > 0xffffffffull;
                my_ts[before + (k - run_first)] = (uint32_t)distance;
                my_val[before + (k - run_first)] = value;
            };
            if (mine > 0) {
                if (piece == 0) emit(0u, from.timestamp); // point 0 is the start time
                if (has_code && from.count < run_end) {
                    const uint32_t codes_end = min(run_end, last_of_stream ? n_total - 1u : next_count);
                    uint32_t next_word = first_bit >> 5;
                    uint64_t buffer = (((uint64_t)word(next_word) << 32) | word(next_word + 1)) << (first_bit & 31u);
                    int32_t available = 64 - (int32_t)(first_bit & 31u);
                    next_word += 2;
                    auto refill = [&]() { // at least 33 bits afterwards, without a branch
                        const bool want = available <= 32;
                        const uint64_t placed = (uint64_t)word(next_word) << ((32 - available) & 63);
                        buffer |= want ? placed : 0ull;
                        available += want ? 32 : 0;
                        next_word += want ? 1u : 0u;
                    };
                    auto consume = [&](uint32_t bits) {
                        buffer <<= bits;
                        available -= (int32_t)bits;
                    };
                    int64_t timestamp = from.timestamp;
                    uint64_t last_delta = from.last_delta;
                    uint32_t k = from.count;
                    while (k < codes_end) {
                        refill();
                        const uint32_t top = (uint32_t)(buffer >> 32);
                        const uint32_t ones = (uint32_t)__clz((int)~top);
                        uint32_t length = 0;
                        if (ones >= 4) { // `11110` + 32 bits or `11111` + 64 bits: rare
                            consume(5);
                            refill();
                            uint64_t encoded = (uint32_t)(buffer >> 32);
                            consume(32);
                            if (ones >= 5) {
                                refill();
                                encoded = (encoded << 32) | (uint32_t)(buffer >> 32);
                                consume(32);
                                last_delta += encoded;
                            } else {
                                last_delta += encoded > (1ull << 31) ? (encoded | (~0ull << 32)) : encoded;
                            }
                        } else { // (a `0` code is a short code with a delta of delta of zero)
                            int32_t delta_of_delta;
                            length = ts_short_code(top, ones, &delta_of_delta);
                            last_delta += (uint64_t)(int64_t)delta_of_delta;
                        }
                        timestamp = (int64_t)((uint64_t)timestamp + last_delta);
                        emit(k++, timestamp);
                        // A second short code out of the same 32 bits: the refill and the loop are paid once for two.
                        const uint32_t rest = top << (length & 31u);
                        const uint32_t ones_then = (uint32_t)__clz((int)~rest);
                        int32_t delta_of_delta_then;
                        const uint32_t both = length + ts_short_code(rest, min(ones_then, 3u), &delta_of_delta_then);
                        if (length != 0u && ones_then < 4u && both <= 32u && k < codes_end) {
                            last_delta += (uint64_t)(int64_t)delta_of_delta_then;
                            timestamp = (int64_t)((uint64_t)timestamp + last_delta);
                            emit(k++, timestamp);
                            length = both;
                        }
                        consume(length);
                    }
                    if (last_of_stream && run_end == n_total) emit(n_total - 1u, end_time);
                }
            }
            leaves = __any(too_far);
        }
        if (leaves) {
            if (lane == 0) args.left_waves[atomicAdd(args.n_left_waves, 1u)] = (uint32_t)wave_index;
            return;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        write_staged_points(lane, wave_first, wave_total, base_time, my_ts, my_val, out_ts, out_val);
        return;
    }
    // A wave with more points than its buffer holds has long runs of `0` codes in its pieces (hundreds of
    // points per piece). A `0` code repeats the delta, so the points between two other codes are an
    // arithmetic run: the lanes decode their pieces into (first index, first timestamp, delta) triples, a
    // handful per piece, and then the whole wave writes the points of all the runs, 64 consecutive ones
    // per store instruction, each lane computing its point from the run it lies in. (A timestamp or a
    // delta that does not fit into 32 bits, or more than TS_MAX_RUNS runs in a piece: direct stores.)
    if (wave_total > TS_STAGE_POINTS) {
        // run r of the lane is runs[(lane * TS_MAX_RUNS + r) * 3 ..]: the index of its first point in the
        // stream, that point's distance from the lane's base time, the delta.
        uint32_t *runs = stage_words;
        uint32_t n_runs = 0;
        bool unfit = false;
        auto open_run = [&](uint32_t k_first, int64_t t, uint64_t delta) {
            const uint64_t distance = (uint64_t)(t - lane_base);
            unfit |= distance > 0xffffffffull || delta > 0xffffffffull || n_runs >= TS_MAX_RUNS;
            if (unfit) return;
            uint32_t *run = runs + ((uint32_t)lane * TS_MAX_RUNS + n_runs) * 3u;
            run[0] = k_first;
            run[1] = (uint32_t)distance;
            run[2] = (uint32_t)delta;
            n_runs += 1;
        };
        if (mine > 0) {
            if (piece == 0) open_run(0u, from.timestamp, 0ull); // point 0 is the start time
            if (has_code && from.count < run_end) {
                const uint32_t codes_end = min(run_end, last_of_stream ? n_total - 1u : next_count);
                uint32_t next_word = first_bit >> 5;
                uint64_t buffer = (((uint64_t)word(next_word) << 32) | word(next_word + 1)) << (first_bit & 31u);
                int32_t available = 64 - (int32_t)(first_bit & 31u);
                next_word += 2;
                auto refill = [&]() { // at least 33 bits afterwards, without a branch
                    const bool want = available <= 32;
                    const uint64_t placed = (uint64_t)word(next_word) << ((32 - available) & 63);
                    buffer |= want ? placed : 0ull;
                    available += want ? 32 : 0;
                    next_word += want ? 1u : 0u;
                };
                auto consume = [&](uint32_t bits) {
                    buffer <<= bits;
                    available -= (int32_t)bits;
                };
                int64_t timestamp = from.timestamp;
                uint64_t last_delta = from.last_delta;
                uint32_t k = from.count;
                bool in_run = false; // the points being produced belong to the run opened last
                while (k < codes_end) {
                    refill();
                    const uint32_t top = (uint32_t)(buffer >> 32);
                    if (top < 0x00800000u) { // nine `0` codes or more: the delta repeats
                        uint32_t run = top == 0u ? 32u : (uint32_t)__clz((int)top);
                        run = min(run, codes_end - k);
                        consume(run);
                        if (!in_run) open_run(k, (int64_t)((uint64_t)timestamp + last_delta), last_delta);
                        in_run = true;
                        timestamp = (int64_t)((uint64_t)timestamp + (uint64_t)run * last_delta);
                        k += run;
                        continue;
                    }
                    const uint32_t ones = (uint32_t)__clz((int)~top);
                    uint32_t length = 0;
                    bool repeats = false; // a single `0` code
                    if (ones >= 4) {
                        consume(5); // `11110` + 32 bits or `11111` + 64 bits
                        refill();
                        uint64_t encoded = (uint32_t)(buffer >> 32);
                        consume(32);
                        if (ones >= 5) {
                            refill();
                            encoded = (encoded << 32) | (uint32_t)(buffer >> 32);
                            consume(32);
                            last_delta += encoded;
                        } else {
                            last_delta += encoded > (1ull << 31) ? (encoded | (~0ull << 32)) : encoded;
                        }
                    } else {
                        int32_t delta_of_delta;
                        length = ts_short_code(top, ones, &delta_of_delta);
                        last_delta += (uint64_t)(int64_t)delta_of_delta;
                        repeats = ones == 0;
                    }
                    consume(length);
                    timestamp = (int64_t)((uint64_t)timestamp + last_delta);
                    if (!(repeats && in_run)) open_run(k, timestamp, last_delta);
                    in_run = true;
                    k += 1;
                }
                if (last_of_stream && run_end == n_total) open_run(n_total - 1u, end_time, 0ull);
            }
        }
        if (!__any(unfit)) {
            // What the other lanes have to know about a piece takes the place of the parked bytes (16 words
            // per piece, word w of piece q at meta[w * 64 + q]); then every lane walks the wave's points 64
            // apart.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            uint32_t *meta = reinterpret_cast<uint32_t *>(&parked_chunks[0][0]);
            auto put = [&](int w, uint32_t value) { meta[w * MDB_WAVE + lane] = value; };
            auto put64 = [&](int w, uint64_t value) {
                put(w, (uint32_t)value);
                put(w + 1, (uint32_t)(value >> 32));
            };
            put(0, before);
            put(1, n_runs);
            put(2, run_first);
            put(3, type == MDB_SWING_ID ? model_end : 0u); // (up to where a Swing value is computed)
            put(4, model_end);                             // (and up to where the value is d.value otherwise)
            put(5, __float_as_uint(d.value));
            put64(6, out_first);
            put64(8, (uint64_t)lane_base);
            put64(10, (uint64_t)__double_as_longlong(d.slope));
            put64(12, (uint64_t)__double_as_longlong(d.intercept));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            auto get = [&](int w, uint32_t q) { return meta[w * MDB_WAVE + q]; };
            auto get64 = [&](int w, uint32_t q) { return (uint64_t)get(w, q) | ((uint64_t)get(w + 1, q) << 32); };
            // (Four consecutive points per lane share the search for their run, but a lane storing its own
            // four writes a quarter of a cache line per instruction: 7.4 ms instead of 5.0 for 10^9 points;
            // through a slab in LDS the stores are as here, but the slab's room leaves 12 runs per piece
            // instead of 16, and a wave with a 13-run piece - nearly every wave at one gap per 100 points -
            // stores directly.)
            uint32_t q = 0; // the piece point p lies in: the last one that starts at or before it
            for (uint32_t p = (uint32_t)lane; p < wave_total; p += MDB_WAVE) {
                while (q + 1 < MDB_WAVE && get(0, q + 1) <= p) q += 1;
                const uint32_t k = get(2, q) + (p - get(0, q)); // index of the point in its stream
                // The last run of the piece that starts at or before k.
                const uint32_t *piece_runs = runs + q * TS_MAX_RUNS * 3u;
                uint32_t lo = 0, hi = get(1, q);
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (piece_runs[mid * 3u] <= k) lo = mid;
                    else hi = mid;
                }
                const uint32_t *run = piece_runs + lo * 3u;
                const int64_t t = (int64_t)(get64(8, q) + run[1] + (uint64_t)(k - run[0]) * run[2]);
                float value = 0.0f;
                if (k < get(3, q))
                    value = (float)(__longlong_as_double((long long)get64(10, q)) * (double)t +
                                    __longlong_as_double((long long)get64(12, q)));
                else if (k < get(4, q))
                    value = __uint_as_float(get(5, q));
                const uint64_t at = get64(6, q) + (p - get(0, q));
                if (out_ts) out_ts[at] = t;
                out_val[at] = value;
            }
            return;
        }
    }
    bool too_far = false;  // a timestamp does not fit into 32 bits from base_time: the wave stores directly
    int64_t held = 0;      // (direct stores) timestamp of an even output position waiting for its neighbour
    bool holding = false;
    for (int attempt = 0; attempt < 2; attempt++) {
        auto emit = [&](uint32_t k, int64_t t) {
            if (k < run_first || k >= run_end) return;
            // swing.rs:304-319 on the timestamp just decoded; a PMC-Mean value is what k_grid_tiles has
            // written already; MacaqueV values and residuals are written later (placeholder, as there).
            float value = 0.0f;
            if (k < model_end) value = type == MDB_SWING_ID ? (float)(d.slope * (double)t + d.intercept) : d.value;
            if (staged) {
                const uint64_t distance = (uint64_t)(t - base_time);
                too_far |= distance > 0xffffffffull;
                my_ts[before + (k - run_first)] = (uint32_t)distance;
                my_val[before + (k - run_first)] = value;
                return;
            }
            const uint64_t at = out_first + (k - run_first);
            out_val[at] = value; // (k_grid_tiles leaves these points alone)
            if (!out_ts) return;
            if (at & 1ull) {
                if (holding) {
                    *reinterpret_cast<longlong2 *>(out_ts + at - 1) = make_longlong2(held, t);
                    holding = false;
                } else {
                    out_ts[at] = t;
                }
            } else {
                held = t;
                holding = true;
            }
        };
        if (mine > 0) {
            if (piece == 0) emit(0u, from.timestamp); // point 0 is the start time
            if (has_code && from.count < run_end) {
                // The codes of this piece produce the points from.count .. codes_end - 1. The last point
                // of a segment is not a code: it is end_time (timestamps.rs:108-113).
                const uint32_t codes_end = min(run_end, last_of_stream ? n_total - 1u : next_count);
                uint32_t next_word = first_bit >> 5;
                uint64_t buffer = (((uint64_t)word(next_word) << 32) | word(next_word + 1)) << (first_bit & 31u);
                int32_t available = 64 - (int32_t)(first_bit & 31u);
                next_word += 2;
                auto refill = [&]() { // at least 33 bits afterwards, without a branch
                    const bool want = available <= 32;
                    const uint64_t placed = (uint64_t)word(next_word) << ((32 - available) & 63);
                    buffer |= want ? placed : 0ull;
                    available += want ? 32 : 0;
                    next_word += want ? 1u : 0u;
                };
                auto consume = [&](uint32_t bits) {
                    buffer <<= bits;
                    available -= (int32_t)bits;
                };
                int64_t timestamp = from.timestamp;
                uint64_t last_delta = from.last_delta;
                uint32_t k = from.count;
                while (k < codes_end) {
                    refill();
                    const uint32_t top = (uint32_t)(buffer >> 32);
                    if (top < 0x00800000u) { // nine `0` codes or more: the delta repeats
                        uint32_t run = top == 0u ? 32u : (uint32_t)__clz((int)top);
                        run = min(run, codes_end - k);
                        consume(run);
                        for (uint32_t j = 0; j < run; j++) {
                            timestamp = (int64_t)((uint64_t)timestamp + last_delta);
                            emit(k++, timestamp);
                        }
                        continue;
                    }
                    const uint32_t ones = (uint32_t)__clz((int)~top);
                    uint32_t length = 0;
                    if (ones >= 4) {
                        consume(5); // `11110` + 32 bits or `11111` + 64 bits
                        refill();
                        uint64_t encoded = (uint32_t)(buffer >> 32);
                        consume(32);
                        if (ones >= 5) {
                            refill();
                            encoded = (encoded << 32) | (uint32_t)(buffer >> 32);
                            consume(32);
                            last_delta += encoded;
                        } else {
                            last_delta += encoded > (1ull << 31) ? (encoded | (~0ull << 32)) : encoded;
                        }
                    } else {
                        int32_t delta_of_delta;
                        length = ts_short_code(top, ones, &delta_of_delta);
                        last_delta += (uint64_t)(int64_t)delta_of_delta;
                    }
                    consume(length);
                    timestamp = (int64_t)((uint64_t)timestamp + last_delta);
                    emit(k++, timestamp);
                }
                if (last_of_stream && run_end == n_total) emit(n_total - 1u, end_time);
            }
            if (holding) {
                out_ts[out_first + (run_end - 1 - run_first)] = held;
                holding = false;
            }
        }
        if (!staged || !__any(too_far)) break;
        staged = false; // once more, with direct stores
    }
    if (!staged) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    write_staged_points(lane, wave_first, wave_total, base_time, my_ts, my_val, out_ts, out_val);
}

// Every wave of the batch, each in whichever way it needs.
__global__ __launch_bounds__(TS_THREADS) void k_grid_timestamps(TsWaveArgs args) {
    __shared__ __attribute__((aligned(16))) uint32_t stage[TS_THREADS / MDB_WAVE][2 * TS_STAGE_POINTS]; // timestamps, then values; or the runs
    __shared__ uint4 parked[TS_THREADS / MDB_WAVE][TS_PIECE_CHUNKS][MDB_WAVE];
    const int wave = threadIdx.x / MDB_WAVE;
    ts_wave<false>(args, (uint64_t)blockIdx.x * (TS_THREADS / MDB_WAVE) + wave, stage[wave], parked[wave]);
}

// Every wave of the batch the sparse way (randomly sampled series: some twenty codes per piece, all different);
// the few that cannot go that way are listed for k_grid_timestamps_left.
__global__ __launch_bounds__(TS_THREADS) void k_grid_timestamps_sparse(TsWaveArgs args) {
    __shared__ __attribute__((aligned(16))) uint32_t stage[TS_THREADS / MDB_WAVE][2 * TS_STAGE_POINTS];
    __shared__ uint4 parked[TS_THREADS / MDB_WAVE][TS_PIECE_CHUNKS][MDB_WAVE];
    const int wave = threadIdx.x / MDB_WAVE;
    ts_wave<true>(args, (uint64_t)blockIdx.x * (TS_THREADS / MDB_WAVE) + wave, stage[wave], parked[wave]);
}

// The waves k_grid_timestamps_sparse has listed, by a fixed number of workgroups (usually there are none).
__global__ __launch_bounds__(TS_THREADS) void k_grid_timestamps_left(TsWaveArgs args) {
    __shared__ __attribute__((aligned(16))) uint32_t stage[TS_THREADS / MDB_WAVE][2 * TS_STAGE_POINTS];
    __shared__ uint4 parked[TS_THREADS / MDB_WAVE][TS_PIECE_CHUNKS][MDB_WAVE];
    const int wave = threadIdx.x / MDB_WAVE;
    const uint32_t n_left = *args.n_left_waves;
    const uint32_t stride = gridDim.x * (TS_THREADS / MDB_WAVE);
    for (uint32_t k = blockIdx.x * (TS_THREADS / MDB_WAVE) + wave; k < n_left; k += stride) {
        ts_wave<false>(args, args.left_waves[k], stage[wave], parked[wave]);
        __builtin_amdgcn_wave_barrier(); // (the next round reuses the wave's LDS)
    }
}

__global__ __launch_bounds__(SERIAL_THREADS) void k_grid_serial(
    DevSegments s, TimeRange range, const unsigned long long *__restrict__ offsets,
    const uint32_t *__restrict__ serial_ids, uint64_t n_serial, const MvSeg *__restrict__ mv_segs,
    const uint32_t *__restrict__ counts, const uint32_t *__restrict__ irregular_totals,
    const uint32_t *__restrict__ irregular_first, int64_t *__restrict__ out_ts, float *__restrict__ out_val,
    GridHeader *__restrict__ header, TsCheckpoints checkpoints, const TileDesc *__restrict__ indexed_desc,
    const unsigned long long *__restrict__ indexed_piece_base) {
    __shared__ uint32_t ring[SERIAL_RING_WORDS][MDB_WAVE];
    const int lane = threadIdx.x;
    const uint64_t slot = (uint64_t)blockIdx.x * SERIAL_THREADS + lane;
    const bool present = slot < n_serial;
    const uint32_t i = present ? serial_ids[slot] : 0;
    SegDesc d;
    d.flags = 0; d.n_total = 0; d.n_model = 0; d.value = 0.0f; d.start = 0; d.delta = 0; d.slope = 0.0; d.intercept = 0.0;
    d.first = 0; d.n_visible = 0;
    uint32_t error = 0;
    if (present) {
        // The few segments with serial work are analysed again here rather than carrying the full
        // descriptor (first visible index, whole-segment counts) through memory for all of them.
        SegInfo info = analyse_segment(s, i, irregular_totals, &checkpoints);
        if (range.enabled) apply_time_range(s, i, info, range, irregular_first, counts, &checkpoints);
        d = info.desc;
        error = info.error;
    }
    // Points [d.first, visible_end) of the segment are wanted; point f goes to o + (f - d.first).
    const uint32_t visible_end = d.first + d.n_visible;
    const uint64_t o = present ? (uint64_t)offsets[i] : 0;
    const uint32_t type = d.flags & FLAG_TYPE_MASK;

    if (present && !(d.flags & (FLAG_REGULAR | FLAG_CHECKPOINTS))) {
        // Irregular timestamps short enough to live inside their view (k_grid_timestamps has the
        // others). Every lane writes into a region of its own, so a store instruction
        // of the wave touches 64 different cache lines: two timestamps are paired into one aligned
        // 16-byte store, and the values are not written here at all when the timestamps are (a
        // PMC-Mean value does not depend on the timestamp and k_grid_tiles has written it already;
        // Swing values are filled in from the stored timestamps by k_grid_swing_irregular, coalesced).
        const uint4 vt = s.timestamps.views[i];
        const uint8_t *bytes = view_data(s.timestamps, i, vt);
        int64_t held = 0;      // timestamp of an even output position waiting for its neighbour
        bool holding = false;
        uint64_t held_at = 0;
        decode_irregular_timestamps(bytes, vt.x, d.start, s.end_time[i], visible_end, &error,
                                    [&](uint32_t k, int64_t t) {
                                        if (k < d.first || k >= visible_end) return;
                                        const uint64_t at = o + (k - d.first);
#ifdef MDB_EXPERIMENT_NO_IRREGULAR_STORES
                                        if (t == 0x7fffffffffffffffll) out_ts[at] = t;
                                        return;
#endif
                                        if (out_ts) {
                                            if (at & 1ull) {
                                                if (holding) {
                                                    *reinterpret_cast<longlong2 *>(out_ts + at - 1) = make_longlong2(held, t);
                                                    holding = false;
                                                } else {
                                                    out_ts[at] = t;
                                                }
                                            } else {
                                                held = t;
                                                held_at = at;
                                                holding = true;
                                            }
                                        } else if (k < d.n_model && type == MDB_SWING_ID) {
                                            // values only (a joined field column): nothing to read t from later
                                            out_val[at] = (float)(d.slope * (double)t + d.intercept);
                                        }
                                    });
        if (holding) out_ts[held_at] = held;
    }

    // Up to two MacaqueV streams per segment: the model's values (type 2) and the residual tail.
    // A stream is decoded from its beginning (the format has no random access) but only as far as
    // the last wanted value; the model's values are also needed in full when residuals are wanted,
    // because the residual stream is seeded with the model's last value.
    // (indexed_desc: the batch has a cursor index and k_grid_mv_pieces decodes every stream piece by piece, but for
    // the residual tails it leaves here, mv_left_to_serial)
    // (indexed_piece_base: the index covers the long streams only - a segment without pieces is decoded here)
    const bool skip_macaque = present && indexed_desc != nullptr && !mv_left_to_serial(indexed_desc[i].flags) &&
                              (indexed_piece_base == nullptr || indexed_piece_base[i + 1] > indexed_piece_base[i]);
    const uint32_t n_res = d.n_total - d.n_model;
    const bool residuals_wanted = present && !skip_macaque && n_res > 0 && visible_end > d.n_model;
    const uint32_t values_to_decode = residuals_wanted ? d.n_model : min(d.n_model, visible_end);
    bool values_pending = present && !skip_macaque && type == MDB_MACAQUE_V_ID && d.n_model > 0 && d.n_visible > 0 &&
                          (residuals_wanted || d.first < d.n_model);
    // Long streams may already have been decoded by the parallel decoder (mdb_macaque_parallel.hpp).
    if (values_pending && mv_segs != nullptr && mv_segs[slot].done) values_pending = false;
    bool residuals_pending = residuals_wanted;
    RingBitReader reader;
    reader.begin(nullptr, 0);
    MacaqueStream stream;
    stream.remaining = 0; stream.position = 0; stream.last = __float_as_uint(d.value);
    stream.leading = 255; stream.trailing = 0; stream.first_is_raw = false; stream.fresh = false;
    bool active = false;
    auto open_next_stream = [&]() {
        if (values_pending) {
            const uint4 vv = s.values.views[i];
            reader.begin(view_data(s.values, i, vv), vv.x);
            if (vv.x == 0) error |= ERR_BITSTREAM;
            stream.remaining = values_to_decode; stream.position = 0; stream.leading = 255; stream.trailing = 0;
            stream.first_is_raw = true;
            stream.fresh = true;
            values_pending = false;
            active = vv.x != 0;
        } else if (residuals_pending) {
            // models/mod.rs:241-249: XOR-seeded with the last RECONSTRUCTED value;
            // stream.last already holds it (PMC / Swing: d.value, MacaqueV: its last decoded value).
            const uint4 vr = s.residuals.views[i];
            reader.begin(view_data(s.residuals, i, vr), vr.x - 1);
            if (vr.x < 2) error |= ERR_BITSTREAM;
            stream.remaining = visible_end - d.n_model; stream.position = d.n_model;
            stream.leading = 255; stream.trailing = 0;
            stream.first_is_raw = false;
            stream.fresh = true;
            residuals_pending = false;
            active = vr.x >= 2;
        } else {
            active = false;
        }
    };
    open_next_stream();

    while (__any(active)) {
        if (__any(active && reader.hungry())) ring_top_up(reader, ring, lane, active);
        if (active) {
            bool malformed;
            const uint32_t bits = ring_decode_value(reader, stream, ring, lane, &malformed);
            if (malformed) {
                error |= ERR_BITSTREAM; // stop after this value
                stream.remaining = 1;
                values_pending = false;
                residuals_pending = false;
            }
            if (stream.position >= d.first && stream.position < visible_end)
                out_val[o + (stream.position - d.first)] = __uint_as_float(bits);
            stream.position += 1;
            stream.remaining -= 1;
            if (stream.remaining == 0) {
                if (reader.overrun()) error |= ERR_BITSTREAM;
                open_next_stream();
            }
        }
    }
    if (error) atomicOr(&header->error, error);
}

// Swing values of segments with irregular timestamps, from the timestamps k_grid_serial has just
// stored: one wave per such segment, coalesced reads of out_ts and writes of out_val
// (swing.rs:304-319: (slope * t + intercept) as f32, in f64).
__global__ __launch_bounds__(256) void k_grid_swing_irregular(
    const TileDesc *__restrict__ desc, const unsigned long long *__restrict__ offsets,
    const uint32_t *__restrict__ serial_ids, uint64_t n_serial, const int64_t *__restrict__ out_ts,
    float *__restrict__ out_val) {
    const uint64_t slot = (uint64_t)blockIdx.x * (blockDim.x / MDB_WAVE) + threadIdx.x / MDB_WAVE;
    if (slot >= n_serial) return;
    const int lane = threadIdx.x % MDB_WAVE;
    const uint32_t i = serial_ids[slot];
    // The prepass has left what is needed in the segment's descriptor: the line, and how many of the
    // visible points the model stands for.
    const TileDesc t = desc[i];
    if ((t.flags & (FLAG_REGULAR | FLAG_CHECKPOINTS | FLAG_JUMPS)) || (t.flags & FLAG_TYPE_MASK) != MDB_SWING_ID) return;
    const uint64_t o = offsets[i];
    for (uint32_t k = lane; k < t.n_model; k += MDB_WAVE)
        out_val[o + k] = (float)(t.slope * (double)out_ts[o + k] + t.intercept);
}

// ---- host side ---------------------------------------------------------------------------------

struct GridPlan {
    TileDesc *desc;
    uint32_t *counts;
    uint32_t *irregular_totals;
    uint32_t *irregular_first;
    unsigned long long *offsets;
    unsigned long long *block_points;
    unsigned long long *block_serial;
    uint32_t *serial_ids;
    uint32_t *tile_first;
    GridHeader *header;
    GridHeader host_header;
    uint32_t n_blocks;
    uint32_t mv_min_values; // MacaqueV streams at least this long go to the parallel decoder
    bool mv_forced;         // MDB_GRID_MV_MIN_VALUES is set: no upper limit on the number of pieces
    TsCheckpoints checkpoints; // of the batch's irregular timestamp streams (piece_base == nullptr: none)
    uint64_t n_ts_pieces;
    std::shared_ptr<MvIndex> mv_index; // cursors into the batch's MacaqueV streams, if it is a resident batch
    std::shared_ptr<MvIndex> ts_cache_pending; // (grid_plan: the batch's timestamp cursors were copied for later calls)
};

// MDB_GRID_MV_MIN_VALUES: "off" disables the parallel MacaqueV decoder, a number sets the stream
// length from which it is used (tests force it down so that short streams exercise it).
static uint32_t mv_min_values_setting() {
    if (const char *text = option_text("MDB_GRID_MV_MIN_VALUES")) {
        if (std::strcmp(text, "off") == 0) return 0xffffffffu;
        const long long value = std::atoll(text);
        if (value >= 2) return (uint32_t)std::min<long long>(value, 0x7fffffff);
    }
    return MV_DEFAULT_MIN_VALUES;
}

// The segments with irregular timestamps in the order k_grid_ts_count takes them (by the length of their
// streams, the longest first): counts them (k_ts_sort<false>, k_ts_sort_scan; *n_streams arrives with the
// caller's next synchronisation of the stream) ...
static int ts_sort_count(mdb_ctx *ctx, const DevSegments &s, uint64_t n, const TimeRange &range, uint32_t *sort_counts,
                         uint32_t *n_streams) {
    const uint32_t sort_blocks = (uint32_t)((n + PREPASS_THREADS * TS_SORT_ITEMS - 1) / (PREPASS_THREADS * TS_SORT_ITEMS));
    MDB_HIP_CHECK(hipMemsetAsync(sort_counts, 0, (TS_SORT_CLASSES + 1) * 4, ctx->stream));
    LaunchTimer timer(ctx, "k_ts_sort");
    hipLaunchKernelGGL(k_ts_sort<false>, dim3(sort_blocks), dim3(PREPASS_THREADS), 0, ctx->stream, s, range, sort_counts,
                       static_cast<uint32_t *>(nullptr));
    hipLaunchKernelGGL(k_ts_sort_scan, dim3(1), dim3(TS_SORT_CLASSES), 0, ctx->stream, sort_counts);
    MDB_HIP_CHECK(hipMemcpyAsync(n_streams, sort_counts + TS_SORT_CLASSES, 4, hipMemcpyDeviceToHost, ctx->stream));
    return 0;
}
// ... and places them.
static void ts_sort_place(mdb_ctx *ctx, const DevSegments &s, uint64_t n, const TimeRange &range, uint32_t *sort_counts,
                          uint32_t *order) {
    const uint32_t sort_blocks = (uint32_t)((n + PREPASS_THREADS * TS_SORT_ITEMS - 1) / (PREPASS_THREADS * TS_SORT_ITEMS));
    LaunchTimer timer(ctx, "k_ts_sort");
    hipLaunchKernelGGL(k_ts_sort<true>, dim3(sort_blocks), dim3(PREPASS_THREADS), 0, ctx->stream, s, range, sort_counts, order);
}

// For the aggregates (mdb_agg.hip): len() of every segment with irregular timestamps (models/mod.rs:98-124: the
// number of codes of its stream) and the sum of every Swing segment among them that has no residuals
// (ts_walk_adds), by the wave-synchronous walk of the grid path instead of one lane per segment decoding its
// stream by itself, twice for a Swing segment. *totals stays nullptr if the batch has no out-of-line timestamps
// (streams inside their views are a handful of points; k_agg_segments counts those itself).
int ts_walk_for_aggregates(mdb_ctx *ctx, const mdb_segments *in, const DevSegments &s, bool with_sums, TimeRange range,
                           const uint32_t **totals, const double **sums, const TsWalkRange **ranges,
                           const unsigned int **error_word_out) {
    *totals = nullptr;
    *sums = nullptr;
    *ranges = nullptr;
    *error_word_out = nullptr;
    const uint64_t n = in->n;
    uint64_t ts_payload = 0;
    for (int32_t b = 0; b < in->timestamps.n_buffers && in->timestamps.buffer_sizes; b++)
        ts_payload += (uint64_t)std::max<int64_t>(in->timestamps.buffer_sizes[b], 0);
    const char *setting = option_text("MDB_AGG_TS_WALK");
    if (ts_payload == 0 || n == 0 || n > 0xfffffff0ull || (setting && std::strcmp(setting, "0") == 0)) return 0;
    void *p;
    // sums or range aggregates, totals, the order, the classes of the sort, an error word
    const uint64_t n_padded = (n + 15) & ~15ull;
    if (scratch_reserve(ctx, SCRATCH_COUNTS, n_padded * (sizeof(TsWalkRange) + 4 + 4) + (TS_SORT_CLASSES + 32) * 4, &p)) return 1;
    TsWalkRange *walk_ranges = static_cast<TsWalkRange *>(p);
    double *walk_sums = static_cast<double *>(p); // (one or the other)
    uint32_t *walk_totals = reinterpret_cast<uint32_t *>(walk_ranges + n_padded);
    uint32_t *order = walk_totals + n_padded;
    uint32_t *sort_counts = order + n_padded;
    unsigned int *error_word = sort_counts + TS_SORT_CLASSES + 16;
    uint32_t n_streams = 0;
    MDB_HIP_CHECK(hipMemsetAsync(error_word, 0, 4, ctx->stream));
    if (ts_sort_count(ctx, s, n, range, sort_counts, &n_streams)) return 1;
    MDB_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (n_streams == 0) return 0;
    ts_sort_place(ctx, s, n, range, sort_counts, order);
    {
        LaunchTimer timer(ctx, "k_grid_ts_count");
        const dim3 blocks((uint32_t)(((uint64_t)n_streams + SERIAL_THREADS - 1) / SERIAL_THREADS));
        const TsCheckpoints none{nullptr, nullptr, nullptr, nullptr, nullptr};
        GridHeader *errors = reinterpret_cast<GridHeader *>(error_word);
        if (range.enabled)
            hipLaunchKernelGGL(k_grid_ts_count<WALK_RANGE>, blocks, dim3(SERIAL_THREADS), 0, ctx->stream, s, none, walk_totals,
                               errors, order, (uint64_t)n_streams, static_cast<double *>(nullptr), range, walk_ranges);
        else if (with_sums)
            hipLaunchKernelGGL(k_grid_ts_count<WALK_SUMS>, blocks, dim3(SERIAL_THREADS), 0, ctx->stream, s, none, walk_totals,
                               errors, order, (uint64_t)n_streams, walk_sums, range, static_cast<TsWalkRange *>(nullptr));
        else
            hipLaunchKernelGGL(k_grid_ts_count<WALK_GRID>, blocks, dim3(SERIAL_THREADS), 0, ctx->stream, s, none, walk_totals,
                               errors, order, (uint64_t)n_streams, static_cast<double *>(nullptr), range,
                               static_cast<TsWalkRange *>(nullptr));
    }
    // (what the walk finds wrong with a stream the aggregate kernels report with their own findings)
    *error_word_out = error_word;
    *totals = walk_totals;
    *sums = !range.enabled && with_sums ? walk_sums : nullptr;
    *ranges = range.enabled ? walk_ranges : nullptr;
    return 0;
}

// One lane per segment: what `whole` (the walk over the whole time axis) says of the segments the range contains, an
// empty answer for the ones it misses, and the list of the ones it cuts (for the walk behind this kernel).
__global__ __launch_bounds__(256) void k_ts_range_select(DevSegments s, TimeRange range, const TsWalkRange *__restrict__ whole,
                                                         const uint32_t *__restrict__ whole_totals, TsWalkRange *__restrict__ ranges,
                                                         uint32_t *__restrict__ totals, uint32_t *__restrict__ cut,
                                                         uint32_t *__restrict__ n_cut) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.n) return;
    totals[i] = whole_totals[i];
    const uint4 view = s.timestamps.views[i];
    const bool irregular = (int32_t)view.x > 0 && (view_inline_byte(view, 0) & 0x80u) != 0;
    if (!irregular) return;
    const int64_t start = s.start_time[i], end = s.end_time[i];
    if (end < range.lo || start > range.hi) return; // (nobody asks about it)
    if (start >= range.lo && end <= range.hi) {
        ranges[i] = whole[i]; // (every point: what the walk with this range would add up, term by term)
        return;
    }
    // (cut by the range: walked - every irregular segment, as the walk of all of them does: the others' number of
    // points is there already, and walking them again changes nothing)
    cut[atomicAdd(n_cut, 1u)] = (uint32_t)i;
}

int ts_range_from_kept(mdb_ctx *ctx, const mdb_segments *in, const DevSegments &s, TimeRange range, MvIndex &kept,
                       const uint32_t **totals, const TsWalkRange **ranges, const unsigned int **error_word_out, bool *available) {
    *available = false;
    const uint64_t n = in->n;
    if (n == 0 || n > 0xfffffff0ull) return 0;
    {
        std::lock_guard<std::mutex> lock(kept.mutex);
        if (kept.range_whole_failed) return 0; // (the walk of every call finds and reports what is wrong with a stream)
        if (!kept.range_whole_built) {
            // Once: every irregular stream walked with the whole time axis as the range.
            const uint32_t *walk_totals = nullptr;
            const double *walk_sums = nullptr;
            const TsWalkRange *walk_ranges = nullptr;
            const unsigned int *walk_error = nullptr;
            if (ts_walk_for_aggregates(ctx, in, s, false, TimeRange{INT64_MIN, INT64_MAX, 1}, &walk_totals, &walk_sums, &walk_ranges,
                                       &walk_error))
                return 1;
            if (!walk_totals || !walk_ranges) return 0; // (no out-of-line timestamp streams: nothing to keep, nothing to walk)
            unsigned int found = 0;
            MDB_HIP_CHECK(hipMemcpyAsync(&found, walk_error, 4, hipMemcpyDeviceToHost, ctx->stream));
            // (no memory to keep it in: the query goes on without, walking what it needs - now and from now on)
            if ((!kept.range_whole && hipMalloc(&kept.range_whole, n * sizeof(TsWalkRange)) != hipSuccess) ||
                (!kept.range_whole_totals && hipMalloc(&kept.range_whole_totals, n * 4) != hipSuccess)) {
                (void)hipGetLastError();
                kept.range_whole_failed = true;
                return 0;
            }
            MDB_HIP_CHECK(hipMemcpyAsync(kept.range_whole, walk_ranges, n * sizeof(TsWalkRange), hipMemcpyDeviceToDevice, ctx->stream));
            MDB_HIP_CHECK(hipMemcpyAsync(kept.range_whole_totals, walk_totals, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
            MDB_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            if (found) { // (a malformed stream: the walk of every call reports it; this pass is not made again)
                kept.range_whole_failed = true;
                return 0;
            }
            kept.range_whole_built = true;
        }
    }
    void *p;
    const uint64_t n_padded = (n + 15) & ~15ull;
    if (scratch_reserve(ctx, SCRATCH_COUNTS, n_padded * (sizeof(TsWalkRange) + 4 + 4) + (TS_SORT_CLASSES + 32) * 4, &p)) return 1;
    TsWalkRange *walk_ranges = static_cast<TsWalkRange *>(p);
    uint32_t *walk_totals = reinterpret_cast<uint32_t *>(walk_ranges + n_padded);
    uint32_t *cut = walk_totals + n_padded;
    uint32_t *counters = cut + n_padded;
    unsigned int *error_word = counters + TS_SORT_CLASSES + 16;
    MDB_HIP_CHECK(hipMemsetAsync(counters, 0, 4, ctx->stream));
    MDB_HIP_CHECK(hipMemsetAsync(error_word, 0, 4, ctx->stream));
    uint32_t n_cut = 0;
    {
        LaunchTimer timer(ctx, "k_ts_range_select");
        hipLaunchKernelGGL(k_ts_range_select, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, s, range,
                           static_cast<const TsWalkRange *>(kept.range_whole), static_cast<const uint32_t *>(kept.range_whole_totals),
                           walk_ranges, walk_totals, cut, counters);
    }
    MDB_HIP_CHECK(hipMemcpyAsync(&n_cut, counters, 4, hipMemcpyDeviceToHost, ctx->stream));
    MDB_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (n_cut > 0) {
        LaunchTimer timer(ctx, "k_grid_ts_count");
        const TsCheckpoints none{nullptr, nullptr, nullptr, nullptr, nullptr};
        hipLaunchKernelGGL(k_grid_ts_count<WALK_RANGE>, dim3((n_cut + SERIAL_THREADS - 1) / SERIAL_THREADS), dim3(SERIAL_THREADS), 0,
                           ctx->stream, s, none, walk_totals, reinterpret_cast<GridHeader *>(error_word), cut, (uint64_t)n_cut,
                           static_cast<double *>(nullptr), range, walk_ranges);
    }
    *error_word_out = error_word;
    *totals = walk_totals;
    *ranges = walk_ranges;
    *available = true;
    return 0;
}

// The cursor index of a batch the library owns on the device (see k_mv_index_walk), built by the first call that
// asks for it. *out stays empty for a foreign or transient batch, for one without MacaqueV streams, for one with a
// malformed stream (the serial kernel reports it) and with MDB_GRID_MV_INDEX=0 (A/B, tests). Uses the counting
// walk's scratch: call it BEFORE grid_plan / the aggregate kernels lay theirs out.
// (the index of a call's own upload, mv_host_index, for the entry points that reach the kernels through
// grid_batch_dev_locked: set around that call by the thread that makes it)
thread_local std::shared_ptr<MvIndex> t_call_index;
thread_local const void *t_call_index_views = nullptr;

int mv_index_prepare(mdb_ctx *ctx, const mdb_segments *in, std::shared_ptr<MvIndex> *out) {
    out->reset();
    const char *setting = option_text("MDB_GRID_MV_INDEX");
    if (setting && std::strcmp(setting, "0") == 0) return 0;
    if (t_call_index && t_call_index_views == in->values.views) {
        *out = t_call_index;
        return 0;
    }
    std::shared_ptr<MvIndex> index = owned_segments_index(in);
    if (!index) return 0;
    std::lock_guard<std::mutex> lock(index->mutex);
    if (!index->built) {
        const uint64_t n = in->n;
        if (n > 0xfffffff0ull) return 0;
        const DevSegments s = to_dev(in);
        const uint32_t *totals = nullptr;
        const double *sums = nullptr;
        const TsWalkRange *ranges = nullptr;
        const unsigned int *walk_error = nullptr;
        if (ts_walk_for_aggregates(ctx, in, s, false, TimeRange{0, 0, 0}, &totals, &sums, &ranges, &walk_error)) return 1;
        void *p = nullptr;
        if (scratch_reserve(ctx, SCRATCH_MV, scan_block_sums_bytes(n) + 64, &p)) return 1;
        unsigned long long *block_sums = static_cast<unsigned long long *>(p);
        unsigned long long *verdict = block_sums + scan_block_sums_bytes(n) / 8; // [0] errors, [1] values walked
        MDB_HIP_CHECK(hipMalloc(&index->piece_base, (n + 1) * 8));
        unsigned long long *piece_base = static_cast<unsigned long long *>(index->piece_base);
        if (device_exclusive_scan(ctx, MvIndexPieces{s, totals}, n, piece_base, block_sums, "k_mv_index_scan")) return 1;
        unsigned long long n_pieces = 0;
        MDB_HIP_CHECK(hipMemcpyAsync(&n_pieces, piece_base + n, 8, hipMemcpyDeviceToHost, ctx->stream));
        MDB_HIP_CHECK(hipMemsetAsync(verdict, 0, 16, ctx->stream));
        MDB_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        unsigned long long host_verdict[2] = {0, 0};
        if (n_pieces > 0) {
            MDB_HIP_CHECK(hipMalloc(&index->cursors, n_pieces * sizeof(MvCursor)));
            {
                LaunchTimer timer(ctx, "k_mv_index_walk");
                hipLaunchKernelGGL(k_mv_index_walk, dim3((uint32_t)((n + SERIAL_THREADS - 1) / SERIAL_THREADS)),
                                   dim3(SERIAL_THREADS), 0, ctx->stream, s, totals, piece_base,
                                   static_cast<MvCursor *>(index->cursors), verdict);
            }
            MDB_HIP_CHECK(hipMemcpyAsync(host_verdict, verdict, 16, hipMemcpyDeviceToHost, ctx->stream));
            MDB_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            MDB_HIP_CHECK(hipGetLastError());
        }
        index->n_pieces = n_pieces;
        index->stream_values = host_verdict[1];
        index->usable = n_pieces > 0 && host_verdict[0] == 0;
        index->built = true;
    }
    if (index->usable) *out = index;
    return 0;
}

// For mdb_agg.hip: the f32 sums of all MacaqueV streams of a batch that has a cursor index (see k_agg_mv_pieces);
// *stream_sums stays nullptr when it has none. known_totals: of the caller's own counting walk (may be nullptr).
int mv_index_stream_sums(mdb_ctx *ctx, const mdb_segments *in, const DevSegments &s, const uint32_t *known_totals,
                         const float **stream_sums, const unsigned long long **only_with_pieces) {
    *stream_sums = nullptr;
    *only_with_pieces = nullptr;
    const char *setting = option_text("MDB_GRID_MV_INDEX");
    if (setting && std::strcmp(setting, "0") == 0) return 0;
    const bool of_this_call = t_call_index && t_call_index_views == in->values.views;
    std::shared_ptr<MvIndex> index = of_this_call ? t_call_index : owned_segments_index(in);
    if (!index) return 0;
    {
        std::lock_guard<std::mutex> lock(index->mutex);
        if (!index->built || !index->usable) return 0; // (built by the first grid call, or by agg_run before its own walk)
    }
    // (the index of one call covers its long streams only: the sums of a segment without pieces are nobody's)
    if (index->of_one_call) *only_with_pieces = static_cast<const unsigned long long *>(index->piece_base);
    const uint64_t piece_waves = (index->n_pieces + MDB_WAVE - 1) / MDB_WAVE;
    if (piece_waves > 0x7ffffff0ull) return fail("Too many MacaqueV streams for one batch.");
    // (every listed stream has two pieces or more)
    const uint64_t most_items = index->n_pieces / 2 + 1;
    const uint64_t counts_bytes = align_up(piece_waves * 8, 256), offsets_bytes = align_up((piece_waves + 1) * 8, 256);
    const uint64_t block_sums_bytes = align_up(scan_block_sums_bytes(piece_waves), 256);
    void *p = nullptr;
    if (scratch_reserve(ctx, SCRATCH_AGG_MV, index->n_pieces * MV_PIECE_VALUES * 4 + in->n * 8 + 256, &p)) return 1;
    uint32_t *values = static_cast<uint32_t *>(p);
    float *sums = reinterpret_cast<float *>(values + index->n_pieces * MV_PIECE_VALUES);
    // (a stream without values - the tail of a segment that has none - sums to 0: the kernels write the others)
    MDB_HIP_CHECK(hipMemsetAsync(sums, 0, 8 * in->n, ctx->stream));
    void *q = nullptr;
    if (scratch_reserve(ctx, SCRATCH_AGG_CHAIN_LIST, counts_bytes + offsets_bytes + block_sums_bytes + 2 * most_items * sizeof(ChainItem) + 64, &q)) return 1;
    uint8_t *at = static_cast<uint8_t *>(q);
    unsigned long long *counts = reinterpret_cast<unsigned long long *>(at);
    unsigned long long *offsets = reinterpret_cast<unsigned long long *>(at + counts_bytes);
    unsigned long long *block_sums = reinterpret_cast<unsigned long long *>(at + counts_bytes + offsets_bytes);
    ChainItem *short_items = reinterpret_cast<ChainItem *>(at + counts_bytes + offsets_bytes + block_sums_bytes);
    ChainItem *long_items = short_items + most_items;
    const MvCursor *cursors = static_cast<const MvCursor *>(index->cursors);
    // Where every wave of pieces lists the streams of two pieces or more that begin in it: a function of the cursors
    // alone, so the index of a resident batch keeps it from the first call that asks (MDB_AGG_KEEP_CHAIN_OFFSETS=0:
    // counted by every call, as the index of one call over host batches is).
    unsigned long long listed = 0;
    bool counted_before = false;
    const char *keep_setting = option_text("MDB_AGG_KEEP_CHAIN_OFFSETS");
    const bool keep = !index->of_one_call && !(keep_setting && std::strcmp(keep_setting, "0") == 0);
    if (keep) {
        std::lock_guard<std::mutex> lock(index->mutex);
        if (index->chains_built) {
            offsets = static_cast<unsigned long long *>(index->chain_offsets);
            listed = index->chains_listed;
            counted_before = true;
        }
    }
    if (!counted_before) {
        void *kept = nullptr;
        if (keep && hipMalloc(&kept, (piece_waves + 1) * 8) != hipSuccess) { // (no memory to keep it in: counted every time)
            (void)hipGetLastError();
            kept = nullptr;
        }
        if (kept) offsets = static_cast<unsigned long long *>(kept);
        {
            LaunchTimer timer(ctx, "k_agg_mv_chain_count");
            hipLaunchKernelGGL(k_agg_mv_chain_count, dim3((uint32_t)piece_waves), dim3(MDB_WAVE), 0, ctx->stream, cursors, index->n_pieces, counts);
        }
        int failed = device_exclusive_scan(ctx, ChainCount{counts}, piece_waves, offsets, block_sums, "k_agg_mv_chain_scan");
        // (how many are listed sizes the launches behind the piece kernel: a wave that finds nothing to do still costs a
        // third of a microsecond, and the most there can be is 5.4 M groups for the mixed series' 10.8 M pieces)
        if (!failed && (mail_read(ctx, &listed, offsets + piece_waves, 8) != hipSuccess || mail_sync(ctx) != hipSuccess))
            failed = fail("Could not read how many MacaqueV streams the pieces list.");
        if (failed) {
            if (kept) (void)hipFree(kept);
            return 1;
        }
        if (kept) { // (complete: the stream has been waited for) - unless another context's call has left its own meanwhile
            std::lock_guard<std::mutex> lock(index->mutex);
            if (!index->chains_built) {
                index->chain_offsets = kept;
                index->chains_listed = listed;
                index->chains_built = true;
                kept = nullptr;
            } else {
                offsets = static_cast<unsigned long long *>(index->chain_offsets);
            }
        }
        if (kept) MDB_HIP_CHECK(hipFree(kept));
    }
    const uint64_t n_short = listed & 0xffffffffull, n_long = listed >> 32;
    if (n_short > most_items || n_long > most_items) return fail("Internal error: more MacaqueV streams listed than there are pieces for.");
    {
        LaunchTimer timer(ctx, "k_agg_mv_pieces");
        hipLaunchKernelGGL(k_agg_mv_pieces<32>, dim3((uint32_t)piece_waves), dim3(MDB_WAVE), 0, ctx->stream, s, cursors, index->n_pieces,
                           values, sums, short_items, long_items, static_cast<const unsigned long long *>(offsets));
    }
    {
        // The listed streams, eight lanes each: the long kind first (they are what takes longest), the short kind behind.
        LaunchTimer timer(ctx, "k_agg_mv_chains");
        if (n_long > 0)
            hipLaunchKernelGGL((k_agg_mv_chain_groups<16>), dim3((uint32_t)((n_long + CHAIN_GROUPS_PER_WAVE - 1) / CHAIN_GROUPS_PER_WAVE)),
                               dim3(MDB_WAVE), 0, ctx->stream, s, known_totals, values, sums, long_items, (unsigned int)n_long);
        if (n_short > 0)
            hipLaunchKernelGGL((k_agg_mv_chain_groups<4>), dim3((uint32_t)((n_short + CHAIN_GROUPS_PER_WAVE - 1) / CHAIN_GROUPS_PER_WAVE)),
                               dim3(MDB_WAVE), 0, ctx->stream, s, known_totals, values, sums, short_items, (unsigned int)n_short);
        if (index->of_one_call)
            hipLaunchKernelGGL(k_agg_mv_check_cursors, dim3((uint32_t)((in->n + 255) / 256)), dim3(256), 0, ctx->stream, s, known_totals,
                               static_cast<const unsigned long long *>(index->piece_base), sums);
    }
    *stream_sums = sums;
    return 0;
}

// For mdb_agg.hip: make sure the batch's index exists before the aggregates lay out their scratch.
int mv_index_ensure(mdb_ctx *ctx, const mdb_segments *in) {
    std::shared_ptr<MvIndex> index;
    return mv_index_prepare(ctx, in, &index);
}

// Runs prepass + scans; leaves descriptors/offsets in scratch and the header on the host.
// capacity_tiles bounds the tile map: if the batch needs more the caller gets an error before any
// out-of-bounds write can happen (tile map writes are guarded by the allocation made here).
// Prepass + scans; leaves descriptors / counts in scratch and the header (totals, error flags,
// metrics) on the host.
int grid_plan(mdb_ctx *ctx, const mdb_segments *in, TimeRange range, GridPlan *plan) {
    const uint64_t n = in->n;
    if (n > 0xfffffff0ull) return fail("Too many segments in one batch.");
    const uint32_t n_blocks = (uint32_t)((n + SEGS_PER_BLOCK - 1) / SEGS_PER_BLOCK);
    plan->n_blocks = n_blocks;
    void *p;
    if (scratch_reserve(ctx, SCRATCH_DESC, n * sizeof(TileDesc), &p)) return 1;
    plan->desc = static_cast<TileDesc *>(p);
    // counts, then (for segments with irregular timestamps only) their totals and first visible index
    if (scratch_reserve(ctx, SCRATCH_COUNTS, 3 * (n + 4) * 4, &p)) return 1;
    plan->counts = static_cast<uint32_t *>(p);
    plan->irregular_totals = plan->counts + n + 4;
    plan->irregular_first = plan->irregular_totals + n + 4;
    if (scratch_reserve(ctx, SCRATCH_OFFSETS, (n + 1) * 8, &p)) return 1;
    plan->offsets = static_cast<unsigned long long *>(p);
    if (scratch_reserve(ctx, SCRATCH_BLOCK_SUMS, (uint64_t)(n_blocks + 1) * 16, &p)) return 1;
    plan->block_points = static_cast<unsigned long long *>(p);
    plan->block_serial = plan->block_points + n_blocks + 1;
    if (scratch_reserve(ctx, SCRATCH_SERIAL_IDS, (n + 1) * 4, &p)) return 1;
    plan->serial_ids = static_cast<uint32_t *>(p);
    plan->tile_first = nullptr;
    if (scratch_reserve(ctx, SCRATCH_HEADER, sizeof(GridHeader), &p)) return 1;
    plan->header = static_cast<GridHeader *>(p);

    MDB_HIP_CHECK(hipMemsetAsync(plan->header, 0, sizeof(GridHeader), ctx->stream));
    std::memset(&plan->host_header, 0, sizeof(GridHeader));
    plan->mv_min_values = mv_min_values_setting();
    plan->mv_forced = option_text("MDB_GRID_MV_MIN_VALUES") != nullptr;
    plan->checkpoints = TsCheckpoints{nullptr, nullptr, nullptr, nullptr, nullptr};
    plan->n_ts_pieces = 0;
    if (n == 0) return 0;
    DevSegments s = to_dev(in);
    // Timestamp streams that do not fit into their views (irregular timestamps of more than a handful
    // of points): a slot per 256-bit piece for the cursors the prepass leaves behind. The column's data
    // buffers say whether there are any (MDB_GRID_TS_PIECES=off: decode them one lane per segment).
    uint64_t ts_payload = 0;
    uint32_t n_ts_streams = 0;          // segments with irregular timestamps, if they have been sorted
    uint32_t *ts_order = nullptr;       // by the length of their streams
    for (int32_t b = 0; b < in->timestamps.n_buffers && in->timestamps.buffer_sizes; b++)
        ts_payload += (uint64_t)std::max<int64_t>(in->timestamps.buffer_sizes[b], 0);
    const char *pieces_setting = option_text("MDB_GRID_TS_PIECES");
    if (pieces_setting && std::strcmp(pieces_setting, "off") == 0) ts_payload = 0;
    // (MDB_GRID_TS_JUMPS=0: no jump lists, every such stream is decoded piece by piece)
    const char *jumps_setting = option_text("MDB_GRID_TS_JUMPS");
    const bool jumps = !range.enabled && !(jumps_setting && std::strcmp(jumps_setting, "0") == 0);
    auto point_checkpoints_into = [&](void *block, unsigned long long *piece_base, unsigned long long n_pieces) {
        plan->checkpoints.piece_base = piece_base;
        plan->checkpoints.slots = static_cast<TsCursor *>(block);
        plan->checkpoints.piece_segment = reinterpret_cast<uint2 *>(plan->checkpoints.slots + n_pieces);
        if (jumps) // (n_pieces * 40 bytes in front: 8-byte aligned, which is all a TsJump's members need)
            plan->checkpoints.jumps = reinterpret_cast<TsJump *>(plan->checkpoints.piece_segment + n_pieces);
        if (jumps && n_pieces < 0xffffffffull)
            plan->checkpoints.live = reinterpret_cast<uint32_t *>(plan->checkpoints.jumps + n_pieces * TS_JUMPS_PER_PIECE);
        plan->n_ts_pieces = n_pieces;
    };
    const uint64_t per_piece = sizeof(TsCursor) + sizeof(uint2) + (jumps ? TS_JUMPS_PER_PIECE * sizeof(TsJump) + 4 : 0);
    // A batch that stays on the device keeps what the walk below leaves (MvIndex::ts_*): the walk is the same
    // every time, and for randomly spaced timestamps it is a third of the call (MDB_GRID_TS_CACHE=0: walk every time).
    const char *cache_setting = option_text("MDB_GRID_TS_CACHE");
    std::shared_ptr<MvIndex> resident;
    if (ts_payload > 0 && !range.enabled && !(cache_setting && std::strcmp(cache_setting, "0") == 0)) resident = owned_segments_index(in);
    bool from_cache = false;
    if (resident) {
        std::lock_guard<std::mutex> lock(resident->mutex);
        from_cache = resident->ts_built && resident->ts_jumps == jumps;
    }
    const uint32_t *known_totals = nullptr;
    if (from_cache) {
        if (resident->ts_n_pieces > 0)
            point_checkpoints_into(resident->ts_slots, static_cast<unsigned long long *>(resident->ts_piece_base), resident->ts_n_pieces);
        MDB_HIP_CHECK(hipMemcpyAsync(plan->irregular_totals, resident->ts_totals, (n + 4) * 4, hipMemcpyDeviceToDevice, ctx->stream));
        MDB_HIP_CHECK(hipMemcpyAsync(&plan->header->live_pieces, &resident->ts_live_pieces, 8, hipMemcpyHostToDevice, ctx->stream));
        known_totals = plan->irregular_totals;
    } else if (ts_payload > 0) {
        // (behind the pieces' scan: the classes of the sort, then the order it puts the streams in)
        const uint64_t scan_bytes = ((n + 1) * 8 + scan_block_sums_bytes(n) + 63) & ~63ull;
        if (scratch_reserve(ctx, SCRATCH_TS_BASE, scan_bytes + (TS_SORT_CLASSES + 16) * 4 + n * 4 + 64, &p)) return 1;
        unsigned long long *piece_base = static_cast<unsigned long long *>(p);
        if (device_exclusive_scan(ctx, TsPieceCount{s}, n, piece_base, piece_base + n + 1, "k_grid_ts_scan")) return 1;
        // The streams by length (MDB_GRID_TS_SORT=0: as they come).
        const char *sort_setting = option_text("MDB_GRID_TS_SORT");
        const bool sorted = !(sort_setting && std::strcmp(sort_setting, "0") == 0);
        uint32_t *sort_counts = reinterpret_cast<uint32_t *>(static_cast<char *>(p) + scan_bytes);
        if (sorted) {
            if (ts_sort_count(ctx, s, n, TimeRange{0, 0, 0}, sort_counts, &n_ts_streams)) return 1;
            ts_order = sort_counts + TS_SORT_CLASSES + 16;
        }
        unsigned long long n_pieces = 0;
        MDB_HIP_CHECK(hipMemcpyAsync(&n_pieces, piece_base + n, 8, hipMemcpyDeviceToHost, ctx->stream));
        MDB_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (sorted && n_ts_streams > 0) ts_sort_place(ctx, s, n, TimeRange{0, 0, 0}, sort_counts, ts_order);
        if (n_pieces > 0) {
            if (scratch_reserve(ctx, SCRATCH_TS_SLOTS, n_pieces * per_piece + 64, &p)) return 1;
            point_checkpoints_into(p, piece_base, n_pieces);
        }
        // The lengths of the irregular timestamp streams (and the cursors of their pieces) first: a walk of
        // its own, so that it can be a wave-synchronous one.
        const uint64_t lanes = ts_order ? n_ts_streams : n;
        if (lanes > 0) {
            LaunchTimer timer(ctx, "k_grid_ts_count");
            hipLaunchKernelGGL(k_grid_ts_count<WALK_GRID>, dim3((uint32_t)((lanes + SERIAL_THREADS - 1) / SERIAL_THREADS)),
                               dim3(SERIAL_THREADS), 0, ctx->stream, s, plan->checkpoints, plan->irregular_totals,
                               plan->header, ts_order, (uint64_t)n_ts_streams, static_cast<double *>(nullptr),
                               TimeRange{0, 0, 0}, static_cast<TsWalkRange *>(nullptr));
        }
        known_totals = plan->irregular_totals;
        if (resident) {
            // Kept for the batch's later calls: copies of what lies in scratch now (enqueued behind the walk).
            std::lock_guard<std::mutex> lock(resident->mutex);
            if (!resident->ts_built && !resident->ts_totals) {
                bool kept = hipMalloc(&resident->ts_totals, (n + 4) * 4) == hipSuccess &&
                            hipMemcpyAsync(resident->ts_totals, plan->irregular_totals, (n + 4) * 4, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess;
                if (kept && n_pieces > 0)
                    kept = hipMalloc(&resident->ts_piece_base, (n + 1) * 8) == hipSuccess &&
                           hipMalloc(&resident->ts_slots, n_pieces * per_piece + 64) == hipSuccess &&
                           hipMemcpyAsync(resident->ts_piece_base, piece_base, (n + 1) * 8, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess &&
                           hipMemcpyAsync(resident->ts_slots, plan->checkpoints.slots, n_pieces * per_piece + 64, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess;
                if (kept) {
                    resident->ts_n_pieces = n_pieces;
                    resident->ts_jumps = jumps;
                    plan->ts_cache_pending = resident; // (usable once this plan's header says the walk found no fault)
                }
            }
        }
    }
    // Simple segments (PMC-Mean / Swing, regular timestamps, no residuals) first, through the trimmed
    // analysis; what that leaves, through the generic one (MDB_GRID_PREPASS_SPLIT=0: everything generic).
    const char *split_setting = option_text("MDB_GRID_PREPASS_SPLIT");
    const bool split = !range.enabled && !(split_setting && std::strcmp(split_setting, "0") == 0);
    unsigned long long *pending = nullptr;
    uint32_t *block_pending = nullptr;
    if (split) {
        if (scratch_reserve(ctx, SCRATCH_PENDING, (n / 64 + 2) * 8 + (uint64_t)n_blocks * 4, &p)) return 1;
        pending = static_cast<unsigned long long *>(p);
        block_pending = reinterpret_cast<uint32_t *>(pending + n / 64 + 2);
    }
    auto prepass = [&](auto kernel, const char *name, uint32_t workgroups) {
        LaunchTimer timer(ctx, name);
        hipLaunchKernelGGL(kernel, dim3(workgroups), dim3(PREPASS_THREADS), 0, ctx->stream, s, range,
                           plan->mv_min_values, plan->desc, plan->counts, plan->irregular_totals,
                           plan->irregular_first, plan->block_points, plan->block_serial, plan->header,
                           plan->checkpoints, known_totals, pending, block_pending, n_blocks);
    };
    if (split) {
        prepass(k_grid_prepass<1>, "k_grid_prepass_simple", n_blocks);
        prepass(k_grid_prepass<2>, "k_grid_prepass", std::min<uint32_t>(n_blocks, 3 * 256 * 4));
    } else {
        prepass(k_grid_prepass<0>, "k_grid_prepass", n_blocks);
    }
    {
        LaunchTimer timer(ctx, "k_scan_blocks");
        hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, ctx->stream, plan->block_points,
                           plan->block_serial, n_blocks, plan->header);
    }
    MDB_HIP_CHECK(hipMemcpyAsync(&plan->host_header, plan->header, sizeof(GridHeader),
                                 hipMemcpyDeviceToHost, ctx->stream));
    MDB_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (plan->ts_cache_pending) {
        std::lock_guard<std::mutex> lock(plan->ts_cache_pending->mutex);
        MvIndex &kept = *plan->ts_cache_pending;
        if (!plan->host_header.error) {
            kept.ts_live_pieces = plan->host_header.live_pieces;
            kept.ts_built = true;
        } else { // (a malformed stream: nothing is kept)
            for (void **allocation : {&kept.ts_totals, &kept.ts_piece_base, &kept.ts_slots}) {
                if (*allocation) (void)hipFree(*allocation);
                *allocation = nullptr;
            }
        }
        plan->ts_cache_pending.reset();
    }
    if (plan->host_header.error) return fail(describe_error(plan->host_header.error));
    if (option_text("MDB_GRID_DEBUG"))
        std::fprintf(stderr, "grid: %llu points, %llu segments with jump lists, %llu points in %llu of %llu pieces left to checkpoints (%llu listed)\n",
                     plan->host_header.total_points, plan->host_header.jump_segments, plan->host_header.checkpointed_points,
                     plan->host_header.checkpointed_pieces, (unsigned long long)plan->n_ts_pieces, plan->host_header.live_pieces);
    return 0;
}

// Output offsets, serial work list, rows-per-segment and the tile map for the planned batch.
int grid_offsets(mdb_ctx *ctx, const mdb_segments *in, uint32_t *rows_per_segment, GridPlan *plan) {
    const uint64_t total = plan->host_header.total_points;
    void *p;
    if (scratch_reserve(ctx, SCRATCH_TILE_MAP, ((total + TILE_POINTS - 1) / TILE_POINTS + 2) * 4, &p)) return 1;
    plan->tile_first = static_cast<uint32_t *>(p);
    if (in->n == 0) return 0;
    LaunchTimer timer(ctx, "k_grid_offsets");
    hipLaunchKernelGGL(k_grid_offsets, dim3(plan->n_blocks), dim3(PREPASS_THREADS), 0, ctx->stream,
                       plan->counts, in->n, plan->block_points, plan->block_serial, plan->offsets,
                       plan->serial_ids, plan->tile_first, rows_per_segment);
    return 0;
}

void fill_metrics(const GridHeader &h, mdb_grid_metrics *m) {
    if (!m) return;
    m->rows_created = h.total_points;
    for (int k = 0; k < 3; k++) {
        m->rows_created_by_model_type[k] = h.metrics[k];
        m->segments_with_model_type[k] = h.metrics[4 + k];
    }
    m->segments_with_residuals = h.metrics[3];
    m->segments_regular = h.metrics[7];
    m->segments_irregular = h.metrics[8];
}

// The parallel decoder (mdb_macaque_parallel.hpp) over n_serial candidate streams holding stream_bytes
// bytes between them: `select` launches the kernel that fills segs[0..n_serial), the decoded values go
// to out_val + MvSeg::out_offset and MvSeg::done says which streams were decoded. *segs_out stays
// nullptr when the batch has so many streams that one lane per stream is the better plan.
template <typename Select>
int mv_pipeline(mdb_ctx *ctx, uint64_t n_serial, uint64_t stream_bytes, bool forced, Select select, float *out_val,
                unsigned int *error, MvSeg **segs_out) {
    // Every qualifying stream has ceil(bits / MV_PIECE_BITS) pieces.
    const uint64_t max_pieces = stream_bytes * 8 / MV_PIECE_BITS + n_serial + 1;
    if (max_pieces > MV_MAX_PIECES && !forced) return 0; // enough streams for one lane per stream
    if (max_pieces * MV_CHAINS > 0x7fffff00ull) return 0;
    const uint64_t sums_bytes = scan_block_sums_bytes(n_serial);
    const uint64_t segs_bytes = align_up(n_serial * sizeof(MvSeg), 256);
    const uint64_t base_bytes = align_up((n_serial + 1) * 8, 256) + align_up(sums_bytes, 256);
    const uint64_t heads_bytes = align_up(max_pieces * MV_CHAINS * MV_HEAD * sizeof(MvRec), 256);
    const uint64_t chains_bytes = align_up(max_pieces * MV_CHAINS * sizeof(MvChain), 256);
    const uint64_t links_bytes = align_up(max_pieces * MV_CHAINS * sizeof(MvLink), 256);
    const uint64_t starts_bytes = align_up(max_pieces * sizeof(MvStart), 256);
    const uint64_t guesses_bytes = 2 * align_up(max_pieces * 4, 256) + 256; // guesses + tried + pending
    void *p = nullptr;
    if (scratch_reserve(ctx, SCRATCH_MV, segs_bytes + base_bytes + heads_bytes + chains_bytes + links_bytes +
                                            starts_bytes + guesses_bytes, &p))
        return 1;
    uint8_t *at = static_cast<uint8_t *>(p);
    MvSeg *segs = reinterpret_cast<MvSeg *>(at);
    at += segs_bytes;
    unsigned long long *piece_base = reinterpret_cast<unsigned long long *>(at);
    unsigned long long *block_sums = reinterpret_cast<unsigned long long *>(at + align_up((n_serial + 1) * 8, 256));
    at += base_bytes;
    MvRec *heads = reinterpret_cast<MvRec *>(at);
    at += heads_bytes;
    MvChain *chains = reinterpret_cast<MvChain *>(at);
    at += chains_bytes;
    MvLink *links = reinterpret_cast<MvLink *>(at);
    at += links_bytes;
    MvStart *starts = reinterpret_cast<MvStart *>(at);
    at += starts_bytes;
    uint32_t *guesses = reinterpret_cast<uint32_t *>(at);
    uint32_t *tried = reinterpret_cast<uint32_t *>(at + align_up(max_pieces * 4, 256));
    uint32_t *pending = reinterpret_cast<uint32_t *>(at + 2 * align_up(max_pieces * 4, 256));
    MDB_HIP_CHECK(hipMemsetAsync(pending, 0, 256, ctx->stream));
    MDB_HIP_CHECK(hipMemsetAsync(starts, 0, starts_bytes, ctx->stream));
    select(segs);
    if (device_exclusive_scan(ctx, MvPieceCount{segs}, n_serial, piece_base, block_sums, "k_mv_scan")) return 1;
    const uint32_t piece_blocks = (uint32_t)((max_pieces + MDB_WAVE - 1) / MDB_WAVE);
    // A wave that stages whole pieces takes 38 KB of LDS, so four of them fill a CU: beyond the
    // 1 024 waves that are resident at once, less staging and more waves is the better trade
    // (16 streams of 65 536 values: 5.8 / 5.95 / 6.0 ms with whole / half / quarter pieces staged;
    // 64 streams: 7.3 / 6.85 / 6.95 ms; 256 streams: 17.0 / 14.7 / 13.8 ms).
    const int staging = max_pieces <= 12288 ? 0 : (max_pieces <= 49152 ? 1 : 2);
    for (int round = 0; round < MV_ROUNDS; round++) {
        if (mv_round_kind(round) == MV_ROUND_GUESS) {
            LaunchTimer timer(ctx, "k_mv_guess");
            hipLaunchKernelGGL(k_mv_guess, dim3((uint32_t)n_serial), dim3(MDB_WAVE), 0, ctx->stream, segs,
                               piece_base, chains, guesses);
        }
        LaunchTimer timer(ctx, round == 0 ? "k_mv_chains_start" : (round == 1 ? "k_mv_chains_first" : "k_mv_chains_more"));
        const dim3 chain_grid((uint32_t)((max_pieces * MV_CHAINS + MDB_WAVE - 1) / MDB_WAVE));
        if (staging == 0)
            hipLaunchKernelGGL(k_mv_chains<MV_STAGE_WORDS>, chain_grid, dim3(MDB_WAVE), 0, ctx->stream, segs, piece_base,
                               n_serial, round, guesses, tried, pending, heads, chains);
        else if (staging == 1)
            hipLaunchKernelGGL(k_mv_chains<MV_STAGE_WORDS_HALF>, chain_grid, dim3(MDB_WAVE), 0, ctx->stream, segs,
                               piece_base, n_serial, round, guesses, tried, pending, heads, chains);
        else
            hipLaunchKernelGGL(k_mv_chains<MV_STAGE_WORDS_QUARTER>, chain_grid, dim3(MDB_WAVE), 0, ctx->stream, segs,
                               piece_base, n_serial, round, guesses, tried, pending, heads, chains);
    }
    {
        LaunchTimer timer(ctx, "k_mv_links");
        const uint32_t chain_blocks = (uint32_t)((max_pieces * MV_CHAINS + MDB_WAVE - 1) / MDB_WAVE);
        hipLaunchKernelGGL(k_mv_links, dim3(chain_blocks), dim3(MDB_WAVE), 0, ctx->stream, segs, piece_base,
                           n_serial, heads, chains, links);
    }
    {
        LaunchTimer timer(ctx, "k_mv_walk");
        hipLaunchKernelGGL(k_mv_walk, dim3((uint32_t)n_serial), dim3(MDB_WAVE), 0, ctx->stream, segs, piece_base,
                           chains, links, starts);
    }
    {
        LaunchTimer timer(ctx, "k_mv_decode");
        if (staging == 0)
            hipLaunchKernelGGL(k_mv_decode<MV_STAGE_WORDS>, dim3(piece_blocks), dim3(MDB_WAVE), 0, ctx->stream, segs,
                               piece_base, n_serial, starts, out_val, error);
        else if (staging == 1)
            hipLaunchKernelGGL(k_mv_decode<MV_STAGE_WORDS_HALF>, dim3(piece_blocks), dim3(MDB_WAVE), 0, ctx->stream, segs,
                               piece_base, n_serial, starts, out_val, error);
        else
            hipLaunchKernelGGL(k_mv_decode<MV_STAGE_WORDS_QUARTER>, dim3(piece_blocks), dim3(MDB_WAVE), 0, ctx->stream,
                               segs, piece_base, n_serial, starts, out_val, error);
    }
    *segs_out = segs;
    return 0;
}

// The long MacaqueV streams of a grid batch; runs after k_grid_tiles and before k_grid_serial, which
// skips the streams marked done.
int grid_parallel_macaque(mdb_ctx *ctx, const DevSegments &s, TimeRange range, GridPlan &plan, float *out_val,
                          MvSeg **segs_out) {
    const uint64_t n_serial = plan.host_header.n_serial;
    auto select = [&](MvSeg *segs) {
        LaunchTimer timer(ctx, "k_mv_select");
        const MvIndex *index = plan.mv_index.get();
        hipLaunchKernelGGL(k_mv_select, dim3((uint32_t)((n_serial + 255) / 256)), dim3(256), 0, ctx->stream, s,
                           range, plan.offsets, plan.serial_ids, n_serial, plan.mv_min_values, segs,
                           index && index->of_one_call ? static_cast<const unsigned long long *>(index->piece_base)
                                                       : static_cast<const unsigned long long *>(nullptr));
    };
    return mv_pipeline(ctx, n_serial, plan.host_header.metrics[9], plan.mv_forced, select, out_val,
                       &plan.header->error, segs_out);
}

// ---- SUM over long MacaqueV streams (for mdb_agg.hip) ------------------------------------------------
//
// macaque_v::sum (macaque_v.rs:220-265) adds the values of a stream one after the other in f32, and
// f32 addition does not associate, so the additions stay with one lane per stream. What need not stay
// there is the decoding, which is a hundred times the work: k_agg_segments leaves the streams that
// qualify for the parallel decoder aside, they are decoded into scratch memory here, and k_mv_sums
// then only has to add floats.

constexpr unsigned long long DEFERRED_ONE = 1ull << 40; // scan item: streams above bit 40, their values below

struct DeferredItem {
    DevSegments s;
    uint32_t min_values;
    TimeRange range;
    const unsigned long long *by_pieces; // (under a time range: the index whose segments k_agg_mv_range takes, or nullptr)
    __device__ uint64_t operator()(uint64_t i) const {
        if (s.model_type_id[i] != MDB_MACAQUE_V_ID) return 0;
        const SegInfo info = analyse_segment(s, i);
        if (by_pieces && by_pieces[i + 1] > by_pieces[i] && mv_range_by_pieces(s, i, info)) return 0;
        const uint32_t values = mv_deferred_values(s, i, info, min_values, range);
        return values ? (DEFERRED_ONE | values) : 0;
    }
};

__global__ __launch_bounds__(256) void k_mv_select_scanned(DevSegments s, TimeRange range,
                                                           const unsigned long long *__restrict__ scan,
                                                           uint32_t min_values, MvSeg *__restrict__ segs) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.n) return;
    const unsigned long long mine = scan[i];
    if ((scan[i + 1] >> 40) == (mine >> 40)) return; // not one of the streams left aside
    SegInfo info = analyse_segment(s, i);
    if (range.enabled) apply_time_range(s, i, info, range);
    segs[mine >> 40] = mv_describe(s, i, info, min_values, mine & (DEFERRED_ONE - 1));
}

struct DeferredResult {
    double sum;
    long long count; // the three below: time-range aggregates only
    float min;
    float max;
    unsigned int error;
    unsigned int pad;
};

// What one stream contributes to a time-range aggregate.
struct RangePartial {
    double sum;
    long long count;
    float min;
    float max;
    __device__ __forceinline__ void clear() {
        sum = 0.0;
        count = 0;
        min = FLT_MAX;
        max = -FLT_MAX;
    }
    __device__ __forceinline__ void point(float v) {
        sum += (double)v;
        count += 1;
        min = min_num(min, v);
        max = max_num(max, v);
    }
    __device__ __forceinline__ void merge(const RangePartial &other) {
        sum += other.sum;
        count += other.count;
        min = min_num(min, other.min);
        max = max_num(max, other.max);
    }
};

// One wave per stream. The additions are a dependent chain that only one lane can walk, so the wave's
// job is to keep that lane fed: all lanes fetch the next MV_SUM_CHUNK values (coalesced, in flight
// while lane 0 adds up the chunk before) and hand them over through LDS.
constexpr uint32_t MV_SUM_CHUNK = 1024;

__global__ __launch_bounds__(MDB_WAVE) void k_mv_sums(const MvSeg *__restrict__ segs, uint64_t n_slots,
                                                      const float *__restrict__ values, float *__restrict__ sums,
                                                      DeferredResult *__restrict__ result) {
    __shared__ float4 chunk_lds[2][MV_SUM_CHUNK / 4];
    const uint32_t slot = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const MvSeg seg = segs[slot];
    if (!seg.done) {
        // The parallel decoder gave this stream up: decode it here, one lane.
        if (lane != 0) return;
        float sum = 0.0f;
        uint32_t error = 0;
        decode_macaque_v(reinterpret_cast<const uint8_t *>(seg.words) + seg.bias_bits / 8, seg.total_bits / 8,
                         seg.n_model, false, 0, &error, [&](uint32_t k, uint32_t bits) {
                             if (k == 0) sum = __uint_as_float(bits);
                             else sum += __uint_as_float(bits);
                         });
        if (error) atomicOr(&result->error, error);
        sums[slot] = sum;
        return;
    }
    constexpr uint32_t PER_LANE = MV_SUM_CHUNK / MDB_WAVE;
    const float *__restrict__ v = values + seg.out_offset;
    const uint32_t n = seg.n_model;
    float fetched[PER_LANE];
    auto fetch = [&](uint32_t base) {
#pragma unroll
        for (uint32_t j = 0; j < PER_LANE; j++) {
            const uint32_t k = base + j * MDB_WAVE + lane;
            fetched[j] = k < n ? v[k] : 0.0f;
        }
    };
    auto hand_over = [&](uint32_t buffer) {
        float *to = reinterpret_cast<float *>(chunk_lds[buffer]);
#pragma unroll
        for (uint32_t j = 0; j < PER_LANE; j++) to[j * MDB_WAVE + lane] = fetched[j];
    };
    fetch(0);
    hand_over(0);
    __syncthreads();
    float sum = 0.0f;
    uint32_t buffer = 0;
    for (uint32_t base = 0; base < n; base += MV_SUM_CHUNK, buffer ^= 1u) {
        const bool more = base + MV_SUM_CHUNK < n;
        if (more) fetch(base + MV_SUM_CHUNK);
        if (lane == 0) {
            const uint32_t count = min(MV_SUM_CHUNK, n - base);
            const float4 *from = chunk_lds[buffer];
            uint32_t k = 0;
            if (base == 0) { // the sum starts AS the first value (macaque_v.rs:228-235)
                const float *first = reinterpret_cast<const float *>(from);
                sum = first[0];
                for (k = 1; k < 4 && k < count; k++) sum += first[k];
            }
#pragma unroll 4
            for (; k + 4 <= count; k += 4) {
                const float4 q = from[k / 4];
                sum += q.x;
                sum += q.y;
                sum += q.z;
                sum += q.w;
            }
            const float *rest = reinterpret_cast<const float *>(from);
            for (; k < count; k++) sum += rest[k];
        }
        if (more) hand_over(buffer ^ 1u);
        __syncthreads();
    }
    if (lane == 0) sums[slot] = sum;
}

// The same sums when there are too many streams for the parallel decoder to pay off: one lane per
// stream decodes (LDS ring, as k_grid_serial) and adds as it goes.
__global__ __launch_bounds__(SERIAL_THREADS) void k_mv_serial_sums(const MvSeg *__restrict__ segs, uint64_t n_slots,
                                                                  float *__restrict__ sums,
                                                                  DeferredResult *__restrict__ result) {
    __shared__ uint32_t ring[SERIAL_RING_WORDS][MDB_WAVE];
    const int lane = threadIdx.x;
    const uint64_t slot = (uint64_t)blockIdx.x * SERIAL_THREADS + lane;
    bool active = slot < n_slots;
    RingBitReader reader;
    reader.begin(nullptr, 0);
    MacaqueStream stream;
    stream.remaining = 0; stream.position = 0; stream.last = 0;
    stream.leading = 255; stream.trailing = 0; stream.first_is_raw = true; stream.fresh = true;
    if (active) {
        const MvSeg seg = segs[slot];
        reader.begin(reinterpret_cast<const uint8_t *>(seg.words) + seg.bias_bits / 8, seg.total_bits / 8);
        stream.remaining = seg.n_model;
        active = seg.n_model > 0 && seg.total_bits > 0;
    }
    float sum = 0.0f;
    uint32_t error = 0;
    while (__any(active)) {
        if (__any(active && reader.hungry())) ring_top_up(reader, ring, lane, active);
        if (active) {
            const bool first = stream.first_is_raw;
            bool malformed;
            const float value = __uint_as_float(ring_decode_value(reader, stream, ring, lane, &malformed));
            sum = first ? value : sum + value; // the sum starts AS the first value (macaque_v.rs:228-235)
            stream.remaining -= 1;
            // A stream shorter than its segment claims ends the loop too: it is bounded by the bits
            // there are, not by a (possibly corrupted) count.
            if (malformed || reader.overrun()) error |= ERR_BITSTREAM;
            if (malformed || reader.overrun() || stream.remaining == 0) active = false;
        }
    }
    if (slot < n_slots) sums[slot] = sum;
    if (error) atomicOr(&result->error, error);
}

constexpr int MV_FINISH_THREADS = 1024;

// ---- the same for aggregates under a time range: SUM (f64), COUNT, MIN, MAX of the visible values ----------

__device__ __forceinline__ RangePartial shfl_down_partial(const RangePartial &p, int delta) {
    RangePartial q;
    const unsigned long long sum_bits = (unsigned long long)__double_as_longlong(p.sum);
    q.sum = __longlong_as_double((long long)(((unsigned long long)__shfl_down((uint32_t)(sum_bits >> 32), delta, MDB_WAVE) << 32) |
                                             __shfl_down((uint32_t)sum_bits, delta, MDB_WAVE)));
    q.count = (long long)(((unsigned long long)__shfl_down((uint32_t)((unsigned long long)p.count >> 32), delta, MDB_WAVE) << 32) |
                          __shfl_down((uint32_t)p.count, delta, MDB_WAVE));
    q.min = __shfl_down(p.min, delta, MDB_WAVE);
    q.max = __shfl_down(p.max, delta, MDB_WAVE);
    return q;
}

// One wave per stream the parallel decoder has put into `values` (its visible values only).
__global__ __launch_bounds__(MDB_WAVE) void k_mv_range_partials(const MvSeg *__restrict__ segs, uint64_t n_slots,
                                                                const float *__restrict__ values,
                                                                RangePartial *__restrict__ partials,
                                                                DeferredResult *__restrict__ result) {
    const uint32_t slot = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const MvSeg seg = segs[slot];
    RangePartial mine;
    mine.clear();
    if (!seg.done) {
        // The parallel decoder gave this stream up: decode it here, one lane.
        if (lane != 0) return;
        uint32_t error = 0;
        decode_macaque_v(reinterpret_cast<const uint8_t *>(seg.words) + seg.bias_bits / 8, seg.total_bits / 8,
                         seg.visible_end, false, 0, &error, [&](uint32_t k, uint32_t bits) {
                             if (k >= seg.first) mine.point(__uint_as_float(bits));
                         });
        if (error) atomicOr(&result->error, error);
        partials[slot] = mine;
        return;
    }
    const float *__restrict__ v = values + seg.out_offset;
    const uint32_t n = seg.visible_end - seg.first;
    for (uint32_t k = lane; k < n; k += MDB_WAVE) mine.point(v[k]);
#pragma unroll
    for (int delta = MDB_WAVE / 2; delta > 0; delta >>= 1) mine.merge(shfl_down_partial(mine, delta));
    if (lane == 0) partials[slot] = mine;
}

// One lane per stream, out of the LDS ring (too many streams for the parallel decoder to pay off).
__global__ __launch_bounds__(SERIAL_THREADS) void k_mv_serial_range(const MvSeg *__restrict__ segs, uint64_t n_slots,
                                                                   RangePartial *__restrict__ partials,
                                                                   DeferredResult *__restrict__ result) {
    __shared__ uint32_t ring[SERIAL_RING_WORDS][MDB_WAVE];
    const int lane = threadIdx.x;
    const uint64_t slot = (uint64_t)blockIdx.x * SERIAL_THREADS + lane;
    bool active = slot < n_slots;
    RingBitReader reader;
    reader.begin(nullptr, 0);
    MacaqueStream stream;
    stream.remaining = 0; stream.position = 0; stream.last = 0;
    stream.leading = 255; stream.trailing = 0; stream.first_is_raw = true; stream.fresh = true;
    uint32_t first = 0;
    if (active) {
        const MvSeg seg = segs[slot];
        reader.begin(reinterpret_cast<const uint8_t *>(seg.words) + seg.bias_bits / 8, seg.total_bits / 8);
        stream.remaining = seg.visible_end; // the format has no random access: from the beginning
        first = seg.first;
        active = seg.visible_end > 0 && seg.total_bits > 0;
    }
    RangePartial mine;
    mine.clear();
    uint32_t error = 0;
    while (__any(active)) {
        if (__any(active && reader.hungry())) ring_top_up(reader, ring, lane, active);
        if (active) {
            bool malformed;
            const float value = __uint_as_float(ring_decode_value(reader, stream, ring, lane, &malformed));
            if (stream.position >= first) mine.point(value);
            stream.position += 1;
            stream.remaining -= 1;
            if (malformed || reader.overrun()) error |= ERR_BITSTREAM;
            if (malformed || reader.overrun() || stream.remaining == 0) active = false;
        }
    }
    if (slot < n_slots) partials[slot] = mine;
    if (error) atomicOr(&result->error, error);
}

__global__ __launch_bounds__(MV_FINISH_THREADS) void k_mv_range_finish(const RangePartial *__restrict__ partials,
                                                                       uint64_t n_slots,
                                                                       DeferredResult *__restrict__ result) {
    __shared__ RangePartial lds[MV_FINISH_THREADS];
    RangePartial mine;
    mine.clear();
    for (uint64_t slot = threadIdx.x; slot < n_slots; slot += MV_FINISH_THREADS) mine.merge(partials[slot]);
    lds[threadIdx.x] = mine;
    __syncthreads();
    for (int width = MV_FINISH_THREADS / 2; width > 0; width >>= 1) {
        if ((int)threadIdx.x < width) {
            RangePartial a = lds[threadIdx.x];
            a.merge(lds[threadIdx.x + width]);
            lds[threadIdx.x] = a;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        result->sum = lds[0].sum;
        result->count = lds[0].count;
        result->min = lds[0].min;
        result->max = lds[0].max;
    }
}

// Aggregates under a time range over a batch with cursors into its MacaqueV streams (a resident batch's sidecar, or
// what the call's host threads left): one lane per piece of 64 values, as k_grid_mv_pieces - but only the pieces that
// reach into the range are decoded, and only as far as it goes; what the points inside it contribute (GridExec +
// filter + aggregate: f64 sum of the f32 values, count, extremes) is reduced per wave. Taken are the MacaqueV
// segments with regular timestamps, no residuals and pieces in the index - the ones k_agg_range leaves out by the
// same test (mv_range_by_pieces) - and the residual tails of PMC-Mean and Swing segments with regular timestamps
// (mv_range_tail_by_pieces; a resident batch's index has their cursors: in k_agg_range a tenth of the lanes of a
// wave would each decode one while the others wait).
__global__ __launch_bounds__(MDB_WAVE) void k_agg_mv_range(DevSegments s, TimeRange range, const MvCursor *__restrict__ cursors,
                                                           unsigned long long n_pieces, RangePartial *__restrict__ partials) {
    __shared__ uint32_t ring[PIECE_RING_ROWS][MDB_WAVE];
    const int lane = threadIdx.x;
    const unsigned long long piece = (unsigned long long)blockIdx.x * MDB_WAVE + lane;
    const uint8_t *values_first = first_buffer(s.values), *residuals_first = first_buffer(s.residuals); // (see view_data())
    RangePartial mine;
    mine.clear();
    uint32_t to_decode = 0, to_skip = 0;
    PieceReader reader;
    PieceState state;
    reader.idle(cursors);
    state.last = 0; state.trailing = 0; state.window_bits = 0; state.raw = false;
    if (piece < n_pieces) {
        const uint4 c0 = load_global(reinterpret_cast<const uint4 *>(cursors + piece));
        const uint4 c1 = load_global(reinterpret_cast<const uint4 *>(cursors + piece) + 1);
        const uint32_t i = c0.z, point_index = c0.w, n_values = c1.x, window = c1.y;
        // (a segment with irregular timestamps is not this kernel's, mv_range_by_pieces: it is left before the analysis,
        // which would walk its timestamp stream to count its points - 0.9 ms per 10^9 points of such series)
        const uint4 ts_view = s.timestamps.views[i];
        const bool irregular = (int32_t)ts_view.x > 0 && (view_inline_byte(ts_view, 0) & 0x80u) != 0;
        if (!irregular && !(s.end_time[i] < range.lo || s.start_time[i] > range.hi)) {
            SegInfo info = analyse_segment(s, i);
            const bool residual = (window & MV_WINDOW_RESIDUAL) != 0;
            if (residual ? mv_range_tail_by_pieces(s, i, info) : mv_range_by_pieces(s, i, info)) {
                // (a tail is XOR-seeded with the model's last RECONSTRUCTED value, models/mod.rs:241-249: what grid() sees)
                const uint32_t seed = residual ? __float_as_uint(info.desc.value) : 0u;
                apply_time_range(s, i, info, range);
                const uint32_t from = max(point_index, info.desc.first);
                const uint32_t upto = min(point_index + n_values, info.desc.first + info.desc.n_visible);
                if (info.desc.n_visible > 0 && from < upto) {
                    to_decode = upto - point_index;
                    to_skip = from - point_index;
                    const DevCol &column = residual ? s.residuals : s.values;
                    const uint4 view = column.views[i];
                    reader.open(view_data(column, i, view, residual ? residuals_first : values_first),
                                residual ? (uint64_t)view.x - 1u : (uint64_t)view.x, c0.x);
                    state.last = seed ^ c0.y;
                    const uint32_t leading = window & 255u, trailing = (window >> 8) & 255u;
                    state.trailing = trailing & 31u;
                    state.window_bits = leading + trailing <= 32u ? 32u - leading - trailing : 0u;
                    state.raw = (window & MV_WINDOW_RAW) != 0;
                }
            }
        }
    }
    if (!__any(to_decode > 0)) { // (no piece of the wave reaches into the range)
        if (lane == 0) partials[blockIdx.x] = mine;
        return;
    }
    reader.begin();
    reader.top_up(ring, lane);
    reader.top_up(ring, lane);
    reader.start(ring, lane);
    // (every lane decodes in every step - straight-line code -, the values wanted are taken)
    for (uint32_t k = 0, most = wave_max_u32(to_decode); k < most; k += 2) {
        if (__any(reader.hungry())) reader.top_up(ring, lane);
        const uint32_t even = piece_decode_value(reader, state, ring, lane);
        const uint32_t odd = piece_decode_value(reader, state, ring, lane);
        if (k >= to_skip && k < to_decode) mine.point(__uint_as_float(even));
        if (k + 1 >= to_skip && k + 1 < to_decode) mine.point(__uint_as_float(odd));
    }
#pragma unroll
    for (int delta = MDB_WAVE / 2; delta > 0; delta >>= 1) mine.merge(shfl_down_partial(mine, delta));
    if (lane == 0) partials[blockIdx.x] = mine;
}

// One workgroup, fixed order (strided partial sums, then a fixed tree): the result does not depend
// on how the work was scheduled.
__global__ __launch_bounds__(MV_FINISH_THREADS) void k_mv_sums_finish(const float *__restrict__ sums, uint64_t n_slots,
                                                                      DeferredResult *__restrict__ result) {
    __shared__ double partial[MV_FINISH_THREADS];
    double sum = 0.0;
    for (uint64_t slot = threadIdx.x; slot < n_slots; slot += MV_FINISH_THREADS) sum += (double)sums[slot];
    partial[threadIdx.x] = sum;
    __syncthreads();
    for (int width = MV_FINISH_THREADS / 2; width > 0; width >>= 1) {
        if ((int)threadIdx.x < width) partial[threadIdx.x] += partial[threadIdx.x + width];
        __syncthreads();
    }
    if (threadIdx.x == 0) result->sum = partial[0];
}

uint32_t macaque_parallel_min_values(bool *forced) {
    *forced = option_text("MDB_GRID_MV_MIN_VALUES") != nullptr;
    return mv_min_values_setting();
}

// The long MacaqueV streams k_agg_segments / k_agg_range left aside (mv_deferred_values; the caller has
// counted them: n_streams streams, n_values values to decode into scratch memory at most, n_bytes
// bytes). Without a range: each stream added up in f32 in stream order, the streams in f64
// (totals->sum). With one: SUM (f64), COUNT, MIN and MAX of the values inside it. *handled stays
// false only when the counts are beyond what the scan item can carry.
int macaque_deferred(mdb_ctx *ctx, const DevSegments &s, TimeRange range, uint32_t min_values, bool forced,
                     uint64_t n_streams, uint64_t n_values, uint64_t n_bytes, bool *handled,
                     DeferredTotals *totals, const unsigned long long *by_pieces) {
    *handled = false;
    if (n_streams == 0 || n_values >= DEFERRED_ONE) return 0;
    // Few enough streams for the parallel decoder (the gate of mv_pipeline)? Then their values go to
    // scratch memory first; otherwise a lane per stream decodes and accumulates in one go.
    const bool parallel = n_bytes * 8 / MV_PIECE_BITS + n_streams + 1 <= MV_MAX_PIECES || forced;
    const uint64_t scan_bytes = align_up((s.n + 1) * 8, 256);
    const uint64_t block_sums_bytes = align_up(scan_block_sums_bytes(s.n), 256);
    const uint64_t values_bytes = parallel ? align_up(n_values * 4, 256) : 256;
    const uint64_t per_stream_bytes = align_up(n_streams * sizeof(RangePartial), 256); // or one float each
    void *p = nullptr;
    if (scratch_reserve(ctx, SCRATCH_AGG_MV, scan_bytes + block_sums_bytes + values_bytes + per_stream_bytes + 256, &p))
        return 1;
    uint8_t *at = static_cast<uint8_t *>(p);
    unsigned long long *scan = reinterpret_cast<unsigned long long *>(at);
    at += scan_bytes;
    unsigned long long *block_sums = reinterpret_cast<unsigned long long *>(at);
    at += block_sums_bytes;
    float *values = reinterpret_cast<float *>(at);
    at += values_bytes;
    float *sums = reinterpret_cast<float *>(at);
    RangePartial *partials = reinterpret_cast<RangePartial *>(at);
    at += per_stream_bytes;
    DeferredResult *result = reinterpret_cast<DeferredResult *>(at);
    MDB_HIP_CHECK(hipMemsetAsync(result, 0, sizeof(DeferredResult), ctx->stream));
    if (device_exclusive_scan(ctx, DeferredItem{s, min_values, range, by_pieces}, s.n, scan, block_sums, "k_mv_deferred_scan"))
        return 1;
    auto select = [&](MvSeg *segs) {
        LaunchTimer timer(ctx, "k_mv_select");
        hipLaunchKernelGGL(k_mv_select_scanned, dim3((uint32_t)((s.n + 255) / 256)), dim3(256), 0, ctx->stream, s,
                           range, scan, min_values, segs);
    };
    MvSeg *segs = nullptr;
    if (parallel && mv_pipeline(ctx, n_streams, n_bytes, forced, select, values, &result->error, &segs)) return 1;
    const bool decoded = segs != nullptr;
    if (!decoded) {
        void *q = nullptr;
        if (scratch_reserve(ctx, SCRATCH_MV, n_streams * sizeof(MvSeg), &q)) return 1;
        segs = static_cast<MvSeg *>(q);
        select(segs);
    }
    const uint32_t lane_blocks = (uint32_t)((n_streams + SERIAL_THREADS - 1) / SERIAL_THREADS);
    if (range.enabled) {
        if (decoded) {
            LaunchTimer timer(ctx, "k_mv_range_partials");
            hipLaunchKernelGGL(k_mv_range_partials, dim3((uint32_t)n_streams), dim3(MDB_WAVE), 0, ctx->stream, segs,
                               n_streams, values, partials, result);
        } else {
            LaunchTimer timer(ctx, "k_mv_serial_range");
            hipLaunchKernelGGL(k_mv_serial_range, dim3(lane_blocks), dim3(SERIAL_THREADS), 0, ctx->stream, segs,
                               n_streams, partials, result);
        }
        LaunchTimer timer(ctx, "k_mv_range_finish");
        hipLaunchKernelGGL(k_mv_range_finish, dim3(1), dim3(MV_FINISH_THREADS), 0, ctx->stream, partials, n_streams,
                           result);
    } else {
        if (decoded) {
            LaunchTimer timer(ctx, "k_mv_sums");
            hipLaunchKernelGGL(k_mv_sums, dim3((uint32_t)n_streams), dim3(MDB_WAVE), 0, ctx->stream, segs, n_streams,
                               values, sums, result);
        } else {
            LaunchTimer timer(ctx, "k_mv_serial_sums");
            hipLaunchKernelGGL(k_mv_serial_sums, dim3(lane_blocks), dim3(SERIAL_THREADS), 0, ctx->stream, segs,
                               n_streams, sums, result);
        }
        LaunchTimer timer(ctx, "k_mv_sums_finish");
        hipLaunchKernelGGL(k_mv_sums_finish, dim3(1), dim3(MV_FINISH_THREADS), 0, ctx->stream, sums, n_streams, result);
    }
    DeferredResult host;
    MDB_HIP_CHECK(hipMemcpyAsync(&host, result, sizeof(DeferredResult), hipMemcpyDeviceToHost, ctx->stream));
    MDB_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    MDB_HIP_CHECK(hipGetLastError());
    if (host.error) return fail(describe_error(host.error));
    *handled = true;
    totals->sum = host.sum;
    totals->count = host.count;
    totals->min = host.min;
    totals->max = host.max;
    return 0;
}

// For agg_run under a time range: the batch's cursor index if it has a usable one (*piece_base stays nullptr
// otherwise) - call it before the aggregates lay out their scratch, a resident batch's index is built here - ...
int mv_index_for_range(mdb_ctx *ctx, const mdb_segments *in, std::shared_ptr<MvIndex> *index, const unsigned long long **piece_base) {
    *piece_base = nullptr;
    if (mv_index_prepare(ctx, in, index)) return 1;
    if (!*index) return 0;
    std::lock_guard<std::mutex> lock((*index)->mutex);
    if (!(*index)->built || !(*index)->usable || (*index)->n_pieces == 0) {
        index->reset();
        return 0;
    }
    *piece_base = static_cast<const unsigned long long *>((*index)->piece_base);
    return 0;
}
// ... and what the indexed MacaqueV segments' points inside the range add up to (k_agg_mv_range).
int mv_index_range_totals(mdb_ctx *ctx, const DevSegments &s, TimeRange range, const MvIndex &index, DeferredTotals *totals) {
    const uint32_t n_blocks = (uint32_t)((index.n_pieces + MDB_WAVE - 1) / MDB_WAVE);
    void *p = nullptr;
    if (scratch_reserve(ctx, SCRATCH_AGG_MV, (uint64_t)n_blocks * sizeof(RangePartial) + 256, &p)) return 1;
    RangePartial *partials = static_cast<RangePartial *>(p);
    DeferredResult *result = reinterpret_cast<DeferredResult *>(reinterpret_cast<uint8_t *>(p) + align_up((uint64_t)n_blocks * sizeof(RangePartial), 256));
    MDB_HIP_CHECK(hipMemsetAsync(result, 0, sizeof(DeferredResult), ctx->stream));
    {
        LaunchTimer timer(ctx, "k_agg_mv_range");
        hipLaunchKernelGGL(k_agg_mv_range, dim3(n_blocks), dim3(MDB_WAVE), 0, ctx->stream, s, range,
                           static_cast<const MvCursor *>(index.cursors), index.n_pieces, partials);
    }
    {
        LaunchTimer timer(ctx, "k_mv_range_finish");
        hipLaunchKernelGGL(k_mv_range_finish, dim3(1), dim3(MV_FINISH_THREADS), 0, ctx->stream, partials, (uint64_t)n_blocks, result);
    }
    DeferredResult host;
    MDB_HIP_CHECK(hipMemcpyAsync(&host, result, sizeof(DeferredResult), hipMemcpyDeviceToHost, ctx->stream));
    MDB_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    MDB_HIP_CHECK(hipGetLastError());
    totals->sum = host.sum;
    totals->count = host.count;
    totals->min = host.min;
    totals->max = host.max;
    return 0;
}

// What k_grid_tiles (or k_grid_fused) leaves: MacaqueV values and residual tails - piece by piece when the batch has
// a cursor index, else by the speculative decoder and one lane per stream - and the irregular timestamps that live
// inside their views.
// A resident batch with cursors into its MacaqueV streams: every piece of 64 values by a lane of its own.
static void launch_mv_pieces(mdb_ctx *ctx, const DevSegments &s, TimeRange range, GridPlan &plan, float *out_val, hipStream_t stream) {
    const MvIndex *index = plan.mv_index.get();
    LaunchTimer timer(ctx, "k_grid_mv_pieces");
    // (MDB_GRID_MV_ROUND: values a lane stages per round, 16 / 32 / 64: A/B)
    const int round = [] {
        const char *text = option_text("MDB_GRID_MV_ROUND");
        const int value = text ? std::atoi(text) : 0;
        return value == 64 || value == 16 ? value : 32;
    }();
    const dim3 blocks((uint32_t)((index->n_pieces + MDB_WAVE - 1) / MDB_WAVE));
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, blocks, dim3(MDB_WAVE), 0, stream, s, range, plan.desc, plan.offsets,
                           plan.irregular_first, static_cast<const MvCursor *>(index->cursors), index->n_pieces, out_val,
                           plan.header);
    };
    if (round == 64) launch(k_grid_mv_pieces<64>);
    else if (round == 16) launch(k_grid_mv_pieces<16>);
    else launch(k_grid_mv_pieces<32>);
#ifdef MDB_MVP_TIMING
    {
        unsigned long long t[8] = {};
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_mvp_timing), sizeof(t));
        const unsigned long long zero[8] = {};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mvp_timing), zero, sizeof(zero));
        const double waves = (double)t[5];
        std::fprintf(stderr, "[mv pieces timing] per wave: kernel %.0f cycles, decode loops %.0f of which top-ups %.0f (%.1f of them, %.0f each), rows out %.0f; %.0f waves\n",
                     t[0] / waves, t[1] / waves, t[2] / waves, t[3] / waves, t[3] ? (double)t[2] / (double)t[3] : 0.0, t[4] / waves, waves);
    }
#endif
}

int grid_launch_streams(mdb_ctx *ctx, const DevSegments &s, TimeRange range, GridPlan &plan, int64_t *out_ts, float *out_val) {
    const uint64_t n_serial = plan.host_header.n_serial;
    const MvSeg *mv_segs = nullptr;
    const MvIndex *index = plan.mv_index.get();
    if (index) launch_mv_pieces(ctx, s, range, plan, out_val, ctx->stream);
    // (the index of one call covers its long streams with regular timestamps: the others are still this decoder's)
    if ((!index || index->of_one_call) && n_serial > 0 && plan.host_header.metrics[9] > 0) {
        MvSeg *segs = nullptr;
        if (grid_parallel_macaque(ctx, s, range, plan, out_val, &segs)) return 1;
        mv_segs = segs;
    }
    // (with the index the serial kernel is left with the irregular timestamps that live inside their views)
    if (n_serial > 0 && (!index || index->of_one_call || plan.host_header.metrics[8] > 0)) {
        LaunchTimer timer(ctx, "k_grid_serial");
        hipLaunchKernelGGL(k_grid_serial,
                           dim3((uint32_t)((n_serial + SERIAL_THREADS - 1) / SERIAL_THREADS)),
                           dim3(SERIAL_THREADS), 0, ctx->stream, s, range, plan.offsets,
                           plan.serial_ids, n_serial, mv_segs, plan.counts, plan.irregular_totals,
                           plan.irregular_first, out_ts, out_val, plan.header, plan.checkpoints,
                           index ? plan.desc : static_cast<const TileDesc *>(nullptr),
                           index && index->of_one_call ? static_cast<const unsigned long long *>(index->piece_base)
                                                       : static_cast<const unsigned long long *>(nullptr));
    }
    if (n_serial > 0 && out_ts != nullptr && plan.host_header.metrics[8] > 0) { // irregular segments exist
        LaunchTimer timer(ctx, "k_grid_swing_irregular");
        hipLaunchKernelGGL(k_grid_swing_irregular, dim3((uint32_t)((n_serial + 3) / 4)), dim3(256), 0, ctx->stream,
                           plan.desc, plan.offsets, plan.serial_ids, n_serial, out_ts, out_val);
    }
    return 0;
}

// Launch the reconstruction of a planned batch into device buffers (enqueue + final error check).
int grid_launch(mdb_ctx *ctx, const mdb_segments *in, TimeRange range, GridPlan &plan, int64_t *out_ts,
                float *out_val, uint32_t *out_rows) {
    const uint64_t total = plan.host_header.total_points;
    if (grid_offsets(ctx, in, out_rows, &plan)) return 1;
    if (total == 0) return 0;
    if (!out_val) return fail("out_val must not be NULL.");
    if ((reinterpret_cast<uintptr_t>(out_ts) & 15u) || (reinterpret_cast<uintptr_t>(out_val) & 15u))
        return fail("Output buffers must be 16-byte aligned.");
    DevSegments s = to_dev(in);
    const uint64_t n_tiles = (total + TILE_POINTS - 1) / TILE_POINTS;
    if (n_tiles > 0x7fffffffull) return fail("Too many output tiles for one launch.");
    {
        // When no segment of the batch has regular timestamps, every timestamp comes from
        // k_grid_serial and the tiles need not store 8 placeholder bytes per point.
        const bool jumps = plan.host_header.jump_segments > 0;
        int64_t *tile_ts = plan.host_header.metrics[7] == 0 && !jumps ? nullptr : out_ts;
        LaunchTimer timer(ctx, jumps ? "k_grid_tiles_jumps" : "k_grid_tiles");
        if (jumps)
            hipLaunchKernelGGL(k_grid_tiles_jumps, dim3((uint32_t)n_tiles), dim3(TILE_THREADS), 0, ctx->stream,
                               plan.desc, plan.offsets, plan.tile_first, s.n, total, n_tiles, tile_ts, out_val,
                               plan.checkpoints.jumps);
        else
            hipLaunchKernelGGL(k_grid_tiles, dim3((uint32_t)n_tiles), dim3(TILE_THREADS), 0, ctx->stream,
                               plan.desc, plan.offsets, plan.tile_first, s.n, total, n_tiles, tile_ts, out_val);
    }
    // (no checkpointed point: every stream with checkpoints has a jump list, or lies outside the time range)
    if (plan.n_ts_pieces > 0 && plan.host_header.checkpointed_points > 0) {
        TsWaveArgs ts_args{s, plan.desc, plan.offsets, plan.irregular_totals, plan.irregular_first, plan.counts,
                           plan.checkpoints, plan.n_ts_pieces, out_ts, out_val, nullptr, nullptr, nullptr, 0};
        // Few of the pieces are to be decoded (the others' segments have jump lists) and the list of those few
        // is complete: one lane per entry of the list instead of one per piece.
        uint64_t n_lanes = plan.n_ts_pieces;
        if (plan.checkpoints.live && plan.host_header.live_pieces == plan.host_header.checkpointed_pieces &&
            plan.host_header.live_pieces * 4 <= plan.n_ts_pieces) {
            ts_args.live = plan.checkpoints.live;
            ts_args.n_live = n_lanes = plan.host_header.live_pieces;
        }
        const uint64_t n_waves = (n_lanes + MDB_WAVE - 1) / MDB_WAVE;
        const uint32_t ts_blocks = (uint32_t)((n_lanes + TS_THREADS - 1) / TS_THREADS);
        // Few points per piece (randomly sampled series): the sparse flavour, then the general one for the
        // waves it has listed. Many (a fixed rate with gaps), or MDB_GRID_TS_SPARSE=0: the general one.
        const char *sparse_setting = option_text("MDB_GRID_TS_SPARSE");
        const bool sparse = !(sparse_setting && std::strcmp(sparse_setting, "0") == 0) &&
                            plan.host_header.checkpointed_points <= (uint64_t)(TS_STAGE_POINTS / MDB_WAVE) * plan.host_header.checkpointed_pieces &&
                            n_waves < 0xffffffffull;
        if (sparse) {
            void *p;
            if (scratch_reserve(ctx, SCRATCH_TS_LEFT, (n_waves + 16) * 4, &p)) return 1;
            ts_args.n_left_waves = static_cast<unsigned int *>(p);
            ts_args.left_waves = ts_args.n_left_waves + 16;
            MDB_HIP_CHECK(hipMemsetAsync(ts_args.n_left_waves, 0, 4, ctx->stream));
            {
                LaunchTimer timer(ctx, "k_grid_timestamps_sparse");
                hipLaunchKernelGGL(k_grid_timestamps_sparse, dim3(ts_blocks), dim3(TS_THREADS), 0, ctx->stream, ts_args);
            }
            LaunchTimer timer(ctx, "k_grid_timestamps");
            hipLaunchKernelGGL(k_grid_timestamps_left, dim3(std::min<uint32_t>(ts_blocks, 2048u)), dim3(TS_THREADS), 0,
                               ctx->stream, ts_args);
        } else {
            LaunchTimer timer(ctx, "k_grid_timestamps");
            hipLaunchKernelGGL(k_grid_timestamps, dim3(ts_blocks), dim3(TS_THREADS), 0, ctx->stream, ts_args);
        }
    }
    return grid_launch_streams(ctx, s, range, plan, out_ts, out_val);
}

// The serial kernel can still find a malformed bitstream; read its verdict (syncs the stream).
int grid_late_error(mdb_ctx *ctx, GridPlan &plan) {
    uint32_t late_error = 0;
    MDB_HIP_CHECK(hipMemcpyAsync(&late_error, &plan.header->error, 4, hipMemcpyDeviceToHost,
                                 ctx->stream));
    MDB_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    MDB_HIP_CHECK(hipGetLastError());
    if (late_error) return fail(describe_error(late_error));
    return 0;
}

// The one-pass path for a batch of short segments with regular timestamps (k_grid_fused). *done = false: the batch
// does not qualify (or turned out to hold a delta-of-delta timestamp stream) and the general pipeline has to run.
int grid_fused(mdb_ctx *ctx, const mdb_segments *in, int64_t *out_ts, float *out_val, uint32_t *out_rows, uint64_t cap,
               uint64_t *n_out, mdb_grid_metrics *metrics, bool *done) {
    *done = false;
    const uint64_t n = in->n;
    // (MDB_GRID_FUSED=0: never; =1: whenever the batch has no out-of-line timestamps, whatever its segments' lengths)
    const char *setting = option_text("MDB_GRID_FUSED");
    const bool forced = setting && std::strcmp(setting, "1") == 0;
    if (setting && std::strcmp(setting, "0") == 0) return 0;
    if (!out_ts || !out_val || n == 0 || n > 0xfffffff0ull) return 0;
    if (!forced && (n < 65536 || cap / n > 96)) return 0; // (long segments: the tile kernel's ground)
    for (int32_t b = 0; b < in->timestamps.n_buffers && in->timestamps.buffer_sizes; b++)
        if (in->timestamps.buffer_sizes[b] > 0) return 0; // (out-of-line timestamps: delta-of-delta streams)
    if ((reinterpret_cast<uintptr_t>(out_ts) & 15u) || (reinterpret_cast<uintptr_t>(out_val) & 15u)) return 0;
    GridPlan plan;
    if (mv_index_prepare(ctx, in, &plan.mv_index)) return 1; // (before the scratch below is laid out)
    const int rounds = [] { // (MDB_GRID_FUSED_ROUNDS: 256-segment rounds per workgroup, 2 / 4 / 8 / 16: A/B; 2.0 / 1.8 / 1.6 / 2.4 ms at 8 points per segment)
        const char *text = option_text("MDB_GRID_FUSED_ROUNDS");
        const int value = text ? std::atoi(text) : 0;
        return value == 2 || value == 4 || value == 16 ? value : 8;
    }();
    const uint64_t n_blocks = (n + (uint64_t)FUSED_THREADS * rounds - 1) / ((uint64_t)FUSED_THREADS * rounds);
    void *p = nullptr;
    if (scratch_reserve(ctx, SCRATCH_BLOCK_SUMS, n_blocks * 8 + 64, &p)) return 1;
    unsigned long long *lookback = static_cast<unsigned long long *>(p);
    if (scratch_reserve(ctx, SCRATCH_HEADER, sizeof(GridHeader), &p)) return 1;
    plan.header = static_cast<GridHeader *>(p);
    // (for the segments with serial work only: the three arrays are sparse)
    if (scratch_reserve(ctx, SCRATCH_DESC, n * sizeof(TileDesc), &p)) return 1;
    plan.desc = static_cast<TileDesc *>(p);
    if (scratch_reserve(ctx, SCRATCH_OFFSETS, (n + 1) * 8, &p)) return 1;
    plan.offsets = static_cast<unsigned long long *>(p);
    if (scratch_reserve(ctx, SCRATCH_SERIAL_IDS, (n + 1) * 4, &p)) return 1;
    plan.serial_ids = static_cast<uint32_t *>(p);
    plan.counts = plan.irregular_totals = plan.irregular_first = nullptr;
    plan.checkpoints = TsCheckpoints{nullptr, nullptr, nullptr, nullptr, nullptr};
    plan.n_ts_pieces = 0;
    MDB_HIP_CHECK(hipMemsetAsync(lookback, 0, n_blocks * 8, ctx->stream));
    MDB_HIP_CHECK(hipMemsetAsync(plan.header, 0, sizeof(GridHeader), ctx->stream));
    const DevSegments s = to_dev(in);
    // (the list of segments with serial work is k_grid_serial's: not needed where cursors cover every stream)
    const bool list_serial = !(plan.mv_index && !plan.mv_index->of_one_call);
    {
        LaunchTimer timer(ctx, "k_grid_fused");
        auto launch = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3((uint32_t)n_blocks), dim3(FUSED_THREADS), 0, ctx->stream, s, lookback, plan.header,
                               out_ts, out_val, out_rows, cap, plan.desc, plan.offsets, plan.serial_ids, list_serial);
        };
        if (rounds == 2) launch(k_grid_fused<2>);
        else if (rounds == 4) launch(k_grid_fused<4>);
        else if (rounds == 16) launch(k_grid_fused<16>);
        else launch(k_grid_fused<8>);
    }
    MDB_HIP_CHECK(hipMemcpyAsync(&plan.host_header, plan.header, sizeof(GridHeader), hipMemcpyDeviceToHost, ctx->stream));
    MDB_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    MDB_HIP_CHECK(hipGetLastError());
    const GridHeader &host = plan.host_header;
    if (host.pad) return 0; // a delta-of-delta timestamp stream: the general pipeline
    if (host.error) return fail(describe_error(host.error));
    if (n_out) *n_out = host.total_points;
    fill_metrics(host, metrics);
    if (host.total_points > cap)
        return fail("Output buffers too small: " + std::to_string(host.total_points) + " data points but capacity " +
                    std::to_string(cap) + ".");
    if (host.n_serial > 0 || (!list_serial && plan.mv_index->n_pieces > 0)) { // (without the list: the cursors say what there is)
        plan.mv_min_values = 0xffffffffu; // (the speculative decoder is for a few long streams: not this path's batches)
        plan.mv_forced = false;
        if (grid_launch_streams(ctx, s, TimeRange{0, 0, 0}, plan, out_ts, out_val)) return 1;
        if (grid_late_error(ctx, plan)) return 1;
    }
    *done = true;
    return 0;
}

int grid_batch_dev_locked(mdb_ctx *ctx, const mdb_segments *in, TimeRange range, int64_t *out_ts,
                          float *out_val, uint32_t *out_rows, uint64_t cap, uint64_t *n_out,
                          mdb_grid_metrics *metrics) {
    if (!range.enabled) {
        bool done = false;
        if (grid_fused(ctx, in, out_ts, out_val, out_rows, cap, n_out, metrics, &done)) return 1;
        if (done) return 0;
    }
    GridPlan plan;
    std::shared_ptr<MvIndex> index;
    if (mv_index_prepare(ctx, in, &index)) return 1;
    if (grid_plan(ctx, in, range, &plan)) return 1;
    plan.mv_index = index;
    const uint64_t total = plan.host_header.total_points;
    if (n_out) *n_out = total;
    fill_metrics(plan.host_header, metrics);
    if (total > cap)
        return fail("Output buffers too small: " + std::to_string(total) + " data points but capacity " +
                    std::to_string(cap) + ".");
    if (grid_launch(ctx, in, range, plan, out_ts, out_val, out_rows)) return 1;
    return grid_late_error(ctx, plan);
}

} // namespace mdb

using namespace mdb;

// (mv_host_index: mdb_mv_host_index.cpp - plain C++, no device code, also built for the CPU tests)

// The walked cursors onto the device (the context's scratch) as the index of the uploaded batch `seg`.
static int mv_host_index_attach(mdb_ctx *ctx, const mdb_segments &seg, const std::vector<unsigned long long> &piece_base,
                                const std::vector<MvCursor> &cursors, std::shared_ptr<MvIndex> *out) {
    out->reset();
    if (cursors.empty() || piece_base.size() != seg.n + 1) return 0;
    const uint64_t base_bytes = align_up(piece_base.size() * 8, 256);
    void *p = nullptr;
    if (scratch_reserve(ctx, SCRATCH_MV_HOST_INDEX, base_bytes + cursors.size() * sizeof(MvCursor), &p)) return 1;
    MDB_HIP_CHECK(hipMemcpyAsync(p, piece_base.data(), piece_base.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    MDB_HIP_CHECK(hipMemcpyAsync(static_cast<uint8_t *>(p) + base_bytes, cursors.data(), cursors.size() * sizeof(MvCursor),
                                 hipMemcpyHostToDevice, ctx->stream));
    auto index = std::make_shared<MvIndex>();
    index->of_one_call = true;
    index->built = index->usable = true;
    index->device = ctx->device;
    index->n_pieces = cursors.size();
    index->piece_base = p;
    index->cursors = static_cast<uint8_t *>(p) + base_bytes;
    *out = index;
    return 0;
}

namespace {

const TimeRange NO_RANGE = {0, 0, 0};

int grid_count_dev_locked(mdb_ctx *ctx, const mdb_segments *in, TimeRange range, uint64_t *n_out) {
    GridPlan plan;
    if (grid_plan(ctx, in, range, &plan)) return 1;
    *n_out = plan.host_header.total_points;
    return 0;
}

int grid_count_dev_impl(mdb_ctx *ctx, const mdb_segments *in, TimeRange range, uint64_t *n_out) {
    if (!ctx || !in || !n_out) return fail("ctx, in and n_out must not be NULL.");
    mdb::CallGuard lock(ctx);
    MDB_HIP_CHECK(hipSetDevice(ctx->device));
    return grid_count_dev_locked(ctx, in, range, n_out);
}

int grid_batch_dev_impl(mdb_ctx *ctx, const mdb_segments *in, TimeRange range, int64_t *out_ts,
                        float *out_val, uint32_t *out_rows_per_segment, uint64_t cap, uint64_t *n_out,
                        mdb_grid_metrics *metrics) {
    if (!ctx || !in) return fail("ctx and in must not be NULL.");
    if (cap > 0 && !out_val) return fail("out_val must not be NULL.");
    mdb::CallGuard lock(ctx);
    MDB_HIP_CHECK(hipSetDevice(ctx->device));
    return grid_batch_dev_locked(ctx, in, range, out_ts, out_val, out_rows_per_segment, cap, n_out,
                                 metrics);
}

int grid_count_host_impl(mdb_ctx *ctx, const mdb_segments *in, TimeRange range, uint64_t *n_out) {
    if (!ctx || !in || !n_out) return fail("ctx, in and n_out must not be NULL.");
    mdb::CallGuard lock(ctx);
    MDB_HIP_CHECK(hipSetDevice(ctx->device));
    mdb_segments_owned *dev = nullptr;
    if (upload_segments_locked(ctx, in, true, &dev)) return 1;
    int rc = grid_count_dev_locked(ctx, &dev->seg, range, n_out);
    mdb_segments_free(dev);
    return rc;
}

int grid_batch_host_impl(mdb_ctx *ctx, const mdb_segments *in, TimeRange range, int64_t *out_ts,
                         float *out_val, uint32_t *out_rows_per_segment, uint64_t cap, uint64_t *n_out,
                         mdb_grid_metrics *metrics) {
    if (!ctx || !in) return fail("ctx and in must not be NULL.");
    if (cap > 0 && (!out_ts || !out_val)) return fail("out_ts and out_val must not be NULL.");
    std::vector<unsigned long long> index_piece_base;
    std::vector<MvCursor> index_cursors;
    // (cursors into the long MacaqueV streams, by host threads; under a time range only into the streams of segments
    // that reach into it: the others have no point to decode)
    const MvHostRange host_range{range.lo, range.hi};
    mv_host_index(&in, 1, &index_piece_base, &index_cursors, range.enabled ? &host_range : nullptr);
    mdb::CallGuard lock(ctx);
    MDB_HIP_CHECK(hipSetDevice(ctx->device));
    mdb_segments_owned *dev = nullptr;
    if (upload_segments_locked(ctx, in, true, &dev)) return 1;
    int rc = mv_host_index_attach(ctx, dev->seg, index_piece_base, index_cursors, &t_call_index);
    t_call_index_views = dev->seg.values.views;
    struct ForgetCallIndex {
        ~ForgetCallIndex() {
            t_call_index.reset();
            t_call_index_views = nullptr;
        }
    } forget_call_index;
    {
        void *stage = nullptr;
        // Device staging for the outputs: timestamps, values, rows-per-segment (256 B aligned).
        const uint64_t ts_bytes = align_up(cap * 8, 256), val_bytes = align_up(cap * 4, 256);
        const uint64_t rows_bytes = align_up(in->n * 4, 256);
        if (!rc) rc = scratch_reserve(ctx, SCRATCH_STAGE_DEV, ts_bytes + val_bytes + rows_bytes, &stage);
        uint8_t *base = static_cast<uint8_t *>(stage);
        int64_t *dev_ts = reinterpret_cast<int64_t *>(base);
        float *dev_val = reinterpret_cast<float *>(base + ts_bytes);
        uint32_t *dev_rows = reinterpret_cast<uint32_t *>(base + ts_bytes + val_bytes);
        uint64_t total = 0;
        if (!rc)
            rc = grid_batch_dev_locked(ctx, &dev->seg, range, dev_ts, dev_val,
                                       out_rows_per_segment ? dev_rows : nullptr, cap, &total, metrics);
        if (n_out) *n_out = total;
        auto copy_back = [&](void *dst, const void *src, uint64_t bytes) {
            if (rc || bytes == 0) return;
            if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
                rc = fail("hipMemcpy device to host failed.");
        };
        copy_back(out_ts, dev_ts, total * 8);
        copy_back(out_val, dev_val, total * 4);
        if (out_rows_per_segment) copy_back(out_rows_per_segment, dev_rows, in->n * 4);
        if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail("stream sync failed.");
    }
    mdb_segments_free(dev);
    return rc;
}

} // namespace

// For mdb_agg.hip: the same around an aggregate call over one host batch.
void mdb::mv_call_index_build(const mdb_segments *in, MvCallIndex *out, const MvHostRange *range) {
    mv_host_index(&in, 1, &out->piece_base, &out->cursors, range);
}
int mdb::mv_call_index_use(mdb_ctx *ctx, const mdb_segments &uploaded, const MvCallIndex &index) {
    if (mv_host_index_attach(ctx, uploaded, index.piece_base, index.cursors, &t_call_index)) return 1;
    t_call_index_views = uploaded.values.views;
    return 0;
}
void mdb::mv_call_index_done() {
    t_call_index.reset();
    t_call_index_views = nullptr;
}

// One or several host batches (rows in the order of the list) reconstructed by one launch into a page-locked
// block of the device's pool: the body of mdb_grid_batch_owned and of the jobs behind mdb_grid_submit.
int mdb::grid_batch_owned_list(mdb_ctx *ctx, const mdb_segments *const *ins, uint32_t n_ins, TimeRangeArg range_arg,
                               bool values_only, uint64_t reserve_front, mdb_grid_result **out) {
    if (!ctx || !ins || !out || n_ins == 0) return fail("ctx, in and out must not be NULL.");
    for (uint32_t k = 0; k < n_ins; k++)
        if (!ins[k]) return fail("ctx, in and out must not be NULL.");
    const TimeRange range{range_arg.lo, range_arg.hi, range_arg.enabled};
    // Cursors into the call's long MacaqueV streams, by host threads (before the context is taken: the other worker
    // of a pipelined stream has it while this one walks).
    std::vector<unsigned long long> index_piece_base;
    std::vector<MvCursor> index_cursors;
    const MvHostRange host_range{range.lo, range.hi}; // (under a time range: only the segments that reach into it)
    // (the call's host-side phases next to the kernels' times when the context is profiled: mdb_profile_get, "host:" names)
    const auto t_begin = std::chrono::steady_clock::now();
    auto since_begin = [&t_begin]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    mv_host_index(ins, n_ins, &index_piece_base, &index_cursors, range.enabled ? &host_range : nullptr);
    const double t_indexed = since_begin();
    mdb::CallGuard lock(ctx);
    const double t_locked = since_begin();
    MDB_HIP_CHECK(hipSetDevice(ctx->device));
    mdb_segments_owned *dev = nullptr;
    if (upload_segment_list_locked(ctx, ins, n_ins, true, &dev)) return 1;
    const double t_uploaded = since_begin();
    double t_planned = t_uploaded, t_launched = t_uploaded, t_down = t_uploaded;
    const uint64_t n_segments = dev->seg.n;
    int rc = 0;
    OwnedGridResult *result = nullptr;
    {
        GridPlan plan;
        rc = grid_plan(ctx, &dev->seg, range, &plan);
        if (!rc) rc = mv_host_index_attach(ctx, dev->seg, index_piece_base, index_cursors, &plan.mv_index);
        t_planned = since_begin();
        const uint64_t total = plan.host_header.total_points;
        // The device staging area mirrors the host block (same gaps), so one copy moves it all.
        const uint64_t front = align_up(reserve_front, 4); // keeps the 16-byte store alignment
        const uint64_t ts_bytes = values_only ? 0 : align_up((front + total) * 8, 256);
        const uint64_t val_bytes = align_up((front + total) * 4, 256);
        const uint64_t rows_bytes = align_up(n_segments * 4, 256);
        void *stage = nullptr;
        if (!rc) rc = scratch_reserve(ctx, SCRATCH_STAGE_DEV, ts_bytes + val_bytes + rows_bytes, &stage);
        uint8_t *base = static_cast<uint8_t *>(stage);
        if (!rc)
            rc = grid_launch(ctx, &dev->seg, range, plan,
                             values_only ? nullptr : reinterpret_cast<int64_t *>(base) + front,
                             reinterpret_cast<float *>(base + ts_bytes) + front,
                             reinterpret_cast<uint32_t *>(base + ts_bytes + val_bytes));
        t_launched = since_begin();
        void *block = nullptr;
        uint64_t capacity = 0;
        if (!rc) rc = ctx->pinned_pool->take(ts_bytes + val_bytes + rows_bytes, &block, &capacity);
        if (!rc) {
            // One copy of the three columns (they are contiguous in the staging area) into the
            // page-locked block; then the serial kernel's verdict.
            if (hipMemcpyAsync(block, stage, ts_bytes + val_bytes + rows_bytes, hipMemcpyDeviceToHost,
                               ctx->stream) != hipSuccess)
                rc = fail("hipMemcpy device to host failed.");
            if (!rc) rc = grid_late_error(ctx, plan);
            t_down = since_begin();
            if (rc) {
                ctx->pinned_pool->give(block, capacity);
            } else {
                result = new OwnedGridResult();
                uint8_t *host = static_cast<uint8_t *>(block);
                result->c.timestamps = values_only ? nullptr : reinterpret_cast<int64_t *>(host) + front;
                result->c.values = reinterpret_cast<float *>(host + ts_bytes) + front;
                result->c.rows_per_segment = reinterpret_cast<uint32_t *>(host + ts_bytes + val_bytes);
                result->c.n = total;
                result->c.n_segments = n_segments;
                result->c.reserved_front = front;
                std::memset(&result->c.metrics, 0, sizeof(result->c.metrics));
                fill_metrics(plan.host_header, &result->c.metrics);
                result->c.priv_ = result;
                result->pool = ctx->pinned_pool;
                result->block = block;
                result->capacity = capacity;
            }
        }
    }
    mdb_segments_free(dev);
    if (ctx->profiling) {
        const std::pair<const char *, double> phases[] = {
            {"host:grid_cursors_by_host_threads", t_indexed},          // the walk of the call's long MacaqueV streams
            {"host:grid_wait_for_the_context", t_locked - t_indexed},  // (the other worker of a pipelined stream has it)
            {"host:grid_upload_segments", t_uploaded - t_locked},      // staging copy by host threads + the copy to the device
            {"host:grid_plan", t_planned - t_uploaded},                // prepass, scans, the header's read-back
            {"host:grid_launches", t_launched - t_planned},            // (asynchronous: the kernels run into the next phase)
            {"host:grid_kernels_and_copy_down", t_down - t_launched},  // ... which ends when the points are in the page-locked block
            {"host:grid_free_segments", since_begin() - t_down}};
        for (const auto &phase : phases) {
            auto &entry = ctx->kernel_times[phase.first];
            entry.launches += 1;
            entry.total_ms += phase.second;
        }
    }
    if (rc) return 1;
    *out = &result->c;
    return 0;
}

namespace {

int grid_batch_owned_impl(mdb_ctx *ctx, const mdb_segments *in, TimeRange range, bool values_only,
                          uint64_t reserve_front, mdb_grid_result **out) {
    return mdb::grid_batch_owned_list(ctx, &in, 1, TimeRangeArg{range.lo, range.hi, range.enabled}, values_only,
                                      reserve_front, out);
}

} // namespace

extern "C" {

int mdb_grid_batch_owned(mdb_ctx *ctx, const mdb_segments *in, uint32_t flags, int64_t t_lo,
                         int64_t t_hi, uint64_t reserve_front, mdb_grid_result **out) {
    return grid_batch_owned_impl(ctx, in, TimeRange{t_lo, t_hi, (flags & MDB_GRID_HAS_RANGE) ? 1 : 0},
                                 (flags & MDB_GRID_VALUES_ONLY) != 0, reserve_front, out);
}

void mdb_grid_result_free(mdb_grid_result *result) {
    if (!result) return;
    OwnedGridResult *owned = static_cast<OwnedGridResult *>(result->priv_);
    if (!owned) return;
    owned->pool->give(owned->block, owned->capacity);
    for (auto &tags : owned->tag_blocks) host_block_give(tags.first, tags.second);
    delete owned;
}

int mdb_grid_count_dev(mdb_ctx *ctx, const mdb_segments *in, uint64_t *n_out) {
    return grid_count_dev_impl(ctx, in, NO_RANGE, n_out);
}

int mdb_grid_batch_dev(mdb_ctx *ctx, const mdb_segments *in, int64_t *out_ts, float *out_val,
                       uint32_t *out_rows_per_segment, uint64_t cap, uint64_t *n_out,
                       mdb_grid_metrics *metrics) {
    return grid_batch_dev_impl(ctx, in, NO_RANGE, out_ts, out_val, out_rows_per_segment, cap, n_out, metrics);
}

int mdb_grid_count(mdb_ctx *ctx, const mdb_segments *in, uint64_t *n_out) {
    return grid_count_host_impl(ctx, in, NO_RANGE, n_out);
}

int mdb_grid_batch(mdb_ctx *ctx, const mdb_segments *in, int64_t *out_ts, float *out_val,
                   uint32_t *out_rows_per_segment, uint64_t cap, uint64_t *n_out,
                   mdb_grid_metrics *metrics) {
    return grid_batch_host_impl(ctx, in, NO_RANGE, out_ts, out_val, out_rows_per_segment, cap, n_out, metrics);
}

int mdb_grid_count_range_dev(mdb_ctx *ctx, const mdb_segments *in, int64_t t_lo, int64_t t_hi,
                             uint64_t *n_out) {
    return grid_count_dev_impl(ctx, in, TimeRange{t_lo, t_hi, 1}, n_out);
}

int mdb_grid_batch_range_dev(mdb_ctx *ctx, const mdb_segments *in, int64_t t_lo, int64_t t_hi,
                             int64_t *out_ts, float *out_val, uint32_t *out_rows_per_segment,
                             uint64_t cap, uint64_t *n_out, mdb_grid_metrics *metrics) {
    return grid_batch_dev_impl(ctx, in, TimeRange{t_lo, t_hi, 1}, out_ts, out_val, out_rows_per_segment, cap,
                               n_out, metrics);
}

int mdb_grid_count_range(mdb_ctx *ctx, const mdb_segments *in, int64_t t_lo, int64_t t_hi,
                         uint64_t *n_out) {
    return grid_count_host_impl(ctx, in, TimeRange{t_lo, t_hi, 1}, n_out);
}

int mdb_grid_batch_range(mdb_ctx *ctx, const mdb_segments *in, int64_t t_lo, int64_t t_hi, int64_t *out_ts,
                         float *out_val, uint32_t *out_rows_per_segment, uint64_t cap, uint64_t *n_out,
                         mdb_grid_metrics *metrics) {
    return grid_batch_host_impl(ctx, in, TimeRange{t_lo, t_hi, 1}, out_ts, out_val, out_rows_per_segment, cap,
                                n_out, metrics);
}

} // extern "C"
