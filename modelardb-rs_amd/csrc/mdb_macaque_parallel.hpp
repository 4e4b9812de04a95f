// mdb_macaque_parallel.hpp - decoding ONE long MacaqueV value stream with many lanes.
//
// macaque_v::grid (crates/modelardb_compression/src/models/macaque_v.rs:272-323) is sequential by
// construction: value i's position in the bit stream depends on every earlier code, and its bits
// are XORed onto value i-1. One lane per stream (k_grid_serial) therefore runs at the latency of a
// single wave, and lossless data (BASELINE config 1: every chunk becomes one 65 536-value MacaqueV
// segment) leaves most of the GPU idle. This file cuts a stream into pieces of MV_PIECE_BITS bits and
// decodes the pieces in parallel, still bit for bit the same values.
//
// What a parse needs from the past is small. Codes are `10` (repeat), `0` + n bits (n = "meaningful
// bits" of the current window) and `11` + 5 bits leading zeros + 6 bits n + n bits (a new window).
// WHERE the codes are therefore depends on (bit position, n) only; the leading zeros of the window
// matter for the values (how far the n bits are shifted), not for the positions, and a `11` code
// sets both from the stream itself.
//
//  * k_mv_chains: the lanes of a piece look for places in it from which the stream parses cleanly to
//    the end of the piece: speculative "chains". Round 0 starts piece 0 at the real beginning. Then,
//    since real streams settle on one window and consist of `0` and `10` codes only from there on,
//    the candidates are the first 45 bits of the piece (the longest code) combined with a guess for
//    n: the n at the end of the nearest earlier piece that has chains (k_mv_guess). Pieces that are
//    still without a chain try the `11` patterns near their start (those need nothing from the
//    past), and further rounds of guesses carry the windows found so far past one more change of n
//    each, until no piece is without a chain. Wrong candidates die within a few codes, with one
//    exception: a parse that is a few bits late keeps reading the (almost always zero) top bits of
//    the window as control bits; such late copies are recognised by their distance to another
//    chain on the grid of code boundaries. Per chain a few boundaries are recorded where the real
//    parse may join it, and per such boundary what only the real parse can interpret (MvTrack).
//  * k_mv_links: every chain is parsed on from the end of its piece until it stands on a boundary
//    that a chain of a later piece recorded with the same n: from there on the two parses visit
//    the same positions. The link notes that target and both chains' accumulators at the boundary.
//  * k_mv_walk: one wave per stream follows the links from piece 0, whose chain starts at the real
//    beginning. Every chain it visits is thereby proven to be on the real parse from the linked
//    boundary on. Along the way it carries the real window (from the last real `11` code) and turns
//    the accumulators into the index and the predecessor value each confirmed chain starts with.
//    Chains it never visits are ignored. A broken link (no partner within MV_MAX_TAIL_BITS, a
//    malformed stream) hands the whole stream back to k_grid_serial.
//  * k_mv_decode: one lane per confirmed piece decodes its values from its confirmed start (position,
//    real window, index, predecessor value) straight to their final positions.
// Nothing is assumed about the data: a speculative chain is only ever used from a boundary the real
// parse has been shown to pass with the same n, and every other situation falls back to the
// sequential decoder. Included by mdb_grid.hip only.
#pragma once

#include "mdb_segment_dev.hpp"

namespace mdb {

constexpr int MV_CHAINS = 4; // speculative chains kept per piece
constexpr int MV_HEAD = 4;   // boundaries recorded per chain
constexpr uint32_t MV_SCAN_BITS = 256;    // `11` patterns are looked for this far into a piece
constexpr uint32_t MV_MAX_CODE_BITS = 45; // 2 + 5 + 6 + 32
constexpr int MV_ROUNDS = 12;             // k_mv_chains launches at most (a round without work costs ~nothing)
enum : int { MV_ROUND_START = 0, MV_ROUND_GUESS = 1, MV_ROUND_SCAN = 2 };
// Piece 0 from the real start; guessed windows for everyone; `11` patterns for pieces that still have
// no chain; then rounds of guesses, each of which carries the windows found so far past one more
// change of n, until no piece is without a chain.
__device__ __host__ inline int mv_round_kind(int round) {
    return round == 0 ? MV_ROUND_START : (round == 2 ? MV_ROUND_SCAN : MV_ROUND_GUESS);
}
constexpr uint32_t MV_SHIFT_BITS = 8;     // how late a parse can be and still live on zero top bits
constexpr uint32_t MV_SETTLE_CODES = 8;   // codes after which a guessed chain is recorded and compared
constexpr uint32_t MV_NO_WINDOW = 0xffffu;
constexpr uint32_t MV_NO_LENGTH = 0xffu;
#ifndef MDB_MV_MAX_TAIL_BITS
#define MDB_MV_MAX_TAIL_BITS (1u << 16)
#endif
constexpr uint32_t MV_MAX_TAIL_BITS = MDB_MV_MAX_TAIL_BITS;
constexpr uint32_t MV_NONE = 0xffffffffu; // link: no partner found
constexpr uint32_t MV_END = 0xfffffffeu;  // link: parsed to the end of the stream

// One stream that qualifies (indexed like serial_ids).
struct MvSeg {
    const uint32_t *words;         // aligned base of the payload
    unsigned long long out_offset; // of the segment's first visible point in out_val
    uint32_t bias_bits;            // slack bits in front of the payload in words[0]
    uint32_t total_bits;
    uint32_t n_words;
    uint32_t n_model;     // values in the stream
    uint32_t first;       // first wanted value index
    uint32_t visible_end; // one past the last wanted value index
    uint32_t n_pieces;    // 0: the stream does not qualify
    uint32_t done;        // set by k_mv_walk: k_mv_decode handles it, k_grid_serial skips it
};

// A recorded code boundary of a chain: a place where the real parse may join it.
struct MvRec {
    uint32_t pos;   // bit position of the next code
    uint32_t state; // window there: leading | meaningful << 8, or MV_NO_WINDOW
    uint32_t count; // values the chain has decoded before it
};

// A chain while it is being followed, and what it has accumulated. A chain cannot know from where on
// it coincides with the real parse, nor whether the leading zeros of its window are real before it
// has read a `11` code AFTER that point. So for each of its recorded boundaries h it keeps apart:
// raw[h], the XOR of the unshifted bits of the `0` codes between boundary h and the next `11` code
// (only the real parse knows how far those are shifted), and snap[h], the value of x at that `11`
// code; x is the XOR of everything the chain decoded, shifted with its own windows, which are real
// from that `11` code on if the real parse joined at boundary h.
struct MvTrack {
    uint32_t pos;
    uint32_t state;
    uint32_t count;
    uint32_t x;
    uint32_t seen; // bit h: a `11` code has been read since boundary h
    uint32_t raw[MV_HEAD];
    uint32_t snap[MV_HEAD];
};

struct MvChain { // chains[piece * MV_CHAINS + c]
    uint32_t n_head; // recorded boundaries, 0: unused
    MvTrack end;     // the chain at its first boundary at or beyond the end of its piece
};

struct MvLink { // links[piece * MV_CHAINS + c]
    uint32_t target;     // id (piece * MV_CHAINS + c) of the chain this parse joins, MV_END or MV_NONE
    uint32_t into_head;  // which recorded boundary of the target it joins at
    uint32_t into_count; // values the target had decoded there
    MvTrack from;        // this parse at the shared boundary
};

struct MvStart {
    uint32_t valid;
    uint32_t pos, state;  // real window
    uint32_t first_index; // index of the first value this piece decodes
    uint32_t value_bits;  // the value before it
    uint32_t n_values;
};

struct MvPieceCount {
    const MvSeg *segs;
    __device__ uint64_t operator()(uint64_t slot) const { return segs[slot].n_pieces; }
};

// Random-access reader: bits [pos, pos + count) of the stream, MSB first, zeros past the end.
// A lane's 64 neighbours read pieces that lie MV_PIECE_BITS apart, so a word load per code would
// touch 64 cache lines per instruction, over and over. Each lane therefore copies the words it is
// going to walk over into LDS once, 16 bytes at a time (`stage`), laid out [word][lane] so that a
// row is conflict free whatever word each lane is at; words outside that window (a parse that runs
// on for several pieces) still come from global memory.
// How much of a piece a lane copies: all of it when the batch is small (few waves, each as fast as it
// can be), half or a quarter when there are more waves than fit next to each other with 38 KB of LDS
// each (what does not get staged comes from global memory as before).
constexpr uint32_t MV_STAGE_WORDS = MV_PIECE_BITS / 32 + 20; // a piece, 8 bits before it, ~600 bits after it
constexpr uint32_t MV_STAGE_WORDS_HALF = MV_PIECE_BITS / 64 + 20;
constexpr uint32_t MV_STAGE_WORDS_QUARTER = MV_PIECE_BITS / 128 + 20;

struct MvReader {
    const uint32_t *words;
    uint32_t n_words;
    uint32_t bias_bits;
    uint32_t total_bits;
    uint32_t cached_word;
    uint64_t cache;
    const uint32_t *staged; // LDS, this lane's column, already byte swapped; nullptr: nothing staged
    uint32_t staged_first;  // first staged word
    uint32_t staged_words;  // how many
    __device__ __forceinline__ void open(const MvSeg &seg) {
        words = seg.words;
        n_words = seg.n_words;
        bias_bits = seg.bias_bits;
        total_bits = seg.total_bits;
        cached_word = 0xfffffffeu; // never index - 1 of a real word
        cache = 0;
        staged = nullptr;
        staged_first = 0;
        staged_words = 0;
    }
    // Copies words [first, first + WORDS) around bit position `from_pos` into `column` (this lane's
    // column of a [WORDS][MDB_WAVE] LDS array). Only this lane reads it back.
    template <uint32_t WORDS> __device__ __forceinline__ void stage(uint32_t *column, uint32_t from_pos) {
        // First staged word: at or before the word of from_pos, on a 16-byte boundary of the ADDRESS
        // (the payload itself is only byte aligned) so that the copy can use 16-byte loads.
        const uint32_t skew = (uint32_t)((reinterpret_cast<uintptr_t>(words) >> 2) & 3u);
        const uint32_t wanted = (bias_bits + from_pos) >> 5;
        const uint32_t rounded = (wanted + skew) & ~3u;
        const bool aligned = rounded >= skew;
        const uint32_t first = aligned ? rounded - skew : 0u;
#pragma unroll 4
        for (uint32_t k = 0; k < WORDS / 4; k++) {
            const uint32_t index = first + 4 * k;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (aligned && index + 3 < n_words) {
                v = *reinterpret_cast<const uint4 *>(words + index);
            } else {
                if (index < n_words) v.x = words[index];
                if (index + 1 < n_words) v.y = words[index + 1];
                if (index + 2 < n_words) v.z = words[index + 2];
                if (index + 3 < n_words) v.w = words[index + 3];
            }
            column[(4 * k + 0) * MDB_WAVE] = __builtin_bswap32(v.x);
            column[(4 * k + 1) * MDB_WAVE] = __builtin_bswap32(v.y);
            column[(4 * k + 2) * MDB_WAVE] = __builtin_bswap32(v.z);
            column[(4 * k + 3) * MDB_WAVE] = __builtin_bswap32(v.w);
        }
        staged = column;
        staged_first = first;
        staged_words = WORDS / 4 * 4;
    }
    __device__ __forceinline__ uint32_t word(uint32_t index) const {
        const uint32_t k = index - staged_first;
        if (k < staged_words) return staged[k * MDB_WAVE];
        return index < n_words ? __builtin_bswap32(words[index]) : 0u;
    }
    // count in [0, 32]
    __device__ __forceinline__ uint32_t peek(uint32_t pos, uint32_t count) {
        if (count == 0) return 0;
        const uint32_t at = bias_bits + pos;
        const uint32_t index = at >> 5;
        if (index != cached_word) {
            // Moving one word forward is the common case: reuse the low half.
            const uint32_t high = index == cached_word + 1 ? (uint32_t)cache : word(index);
            cache = ((uint64_t)high << 32) | word(index + 1);
            cached_word = index;
        }
        return (uint32_t)((cache << (at & 31u)) >> (64u - count));
    }
};

enum : int { MV_OK = 0, MV_MALFORMED = 1, MV_OVERRUN = 2 };
enum : uint32_t { MV_CODE_BITS = 0, MV_CODE_REPEAT = 1, MV_CODE_WINDOW = 2 };

__device__ __forceinline__ bool mv_valid_window(uint32_t leading, uint32_t meaningful) {
    // macaque_v.rs:305-313 as decode_macaque_v checks it: meaningful <= 32 and trailing <= 31.
    return meaningful <= 32u && leading + meaningful <= 32u && leading + meaningful >= 1u;
}

__device__ __forceinline__ uint32_t mv_length(uint32_t state) { return state >> 8; }

__device__ __forceinline__ uint32_t mv_shifted(uint32_t bits, uint32_t state) {
    const uint32_t trailing = 32u - (state >> 8) - (state & 0xffu);
    return bits << (trailing & 31u);
}

// Decodes the code at `pos` under window `state`; on MV_OK pos / state are advanced, `kind` says
// which code it was and `bits` holds its unshifted payload. Nothing is changed otherwise.
__device__ __forceinline__ int mv_step(MvReader &r, uint32_t &pos, uint32_t &state, uint32_t &kind,
                                       uint32_t &bits) {
    const uint32_t top = r.peek(pos, 13); // c0 c1 leading[5] meaningful[6]
    uint32_t header, meaningful, next_state = state;
    if ((top >> 12) == 0) { // `0`: the previous window again
        if (state == MV_NO_WINDOW) return MV_MALFORMED;
        header = 1;
        meaningful = state >> 8;
        kind = MV_CODE_BITS;
    } else if ((top >> 11) == 2) { // `10`: the value repeats
        if (pos + 2 > r.total_bits) return MV_OVERRUN;
        pos += 2;
        kind = MV_CODE_REPEAT;
        bits = 0;
        return MV_OK;
    } else { // `11` + window
        const uint32_t leading = (top >> 6) & 31u;
        meaningful = top & 63u;
        if (!mv_valid_window(leading, meaningful)) return MV_MALFORMED;
        header = 13;
        next_state = leading | (meaningful << 8);
        kind = MV_CODE_WINDOW;
    }
    if (pos + header + meaningful > r.total_bits) return MV_OVERRUN;
    bits = r.peek(pos + header, meaningful);
    pos += header + meaningful;
    state = next_state;
    return MV_OK;
}

// One code of a chain that has n_head recorded boundaries: like mv_step, plus the accumulators.
__device__ __forceinline__ int mv_track_step(MvReader &r, MvTrack &t, uint32_t n_head) {
    uint32_t kind = 0, bits = 0;
    const int rc = mv_step(r, t.pos, t.state, kind, bits);
    if (rc != MV_OK) return rc;
    t.count += 1;
    if (kind == MV_CODE_WINDOW) {
#pragma unroll
        for (uint32_t h = 0; h < MV_HEAD; h++)
            if (h < n_head && !((t.seen >> h) & 1u)) {
                t.snap[h] = t.x;
                t.seen |= 1u << h;
            }
    } else if (kind == MV_CODE_BITS) {
#pragma unroll
        for (uint32_t h = 0; h < MV_HEAD; h++)
            if (h < n_head && !((t.seen >> h) & 1u)) t.raw[h] ^= bits;
    }
    if (kind != MV_CODE_REPEAT) t.x ^= mv_shifted(bits, t.state);
    return MV_OK;
}

__device__ __forceinline__ MvTrack mv_track_at(uint32_t pos, uint32_t state) {
    MvTrack t;
    t.pos = pos;
    t.state = state;
    t.count = 0;
    t.x = 0;
    t.seen = 0;
#pragma unroll
    for (int h = 0; h < MV_HEAD; h++) t.raw[h] = t.snap[h] = 0;
    return t;
}

// Last slot whose first piece is <= piece (slots without pieces share the base of the next one).
__device__ __forceinline__ uint32_t mv_slot_of(const unsigned long long *piece_base, uint64_t n_slots,
                                               uint64_t piece) {
    uint64_t lo = 0, hi = n_slots;
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) / 2;
        if (piece_base[mid] <= piece) lo = mid;
        else hi = mid;
    }
    return (uint32_t)lo;
}

// What the decoder needs to know about the stream of segment i: nothing (n_pieces 0) unless it
// qualifies. out_offset: where the value of its first visible point goes.
__device__ __forceinline__ MvSeg mv_describe(const DevSegments &s, uint64_t i, const SegInfo &info, uint32_t min_values,
                                             unsigned long long out_offset) {
    const uint4 view = s.values.views[i];
    MvSeg seg;
    seg.words = nullptr;
    seg.out_offset = 0;
    seg.bias_bits = seg.total_bits = seg.n_words = seg.n_model = seg.first = seg.visible_end = 0;
    seg.n_pieces = 0;
    seg.done = 0;
    if (mv_qualifies(info, view.x, min_values)) {
        const uint8_t *bytes = view_data(s.values, i, view);
        const uintptr_t address = reinterpret_cast<uintptr_t>(bytes);
        const uint32_t misalign = (uint32_t)(address & 3u);
        seg.words = reinterpret_cast<const uint32_t *>(address - misalign);
        seg.bias_bits = 8u * misalign;
        seg.total_bits = 8u * view.x;
        seg.n_words = (view.x + misalign + 3u) >> 2;
        seg.n_model = info.desc.n_model;
        seg.first = info.desc.first;
        seg.visible_end = info.desc.first + info.desc.n_visible;
        seg.out_offset = out_offset;
        seg.n_pieces = (seg.total_bits + MV_PIECE_BITS - 1) / MV_PIECE_BITS;
    }
    return seg;
}

// ---- k_mv_select: one lane per entry of the serial list -----------------------------------------------

__global__ __launch_bounds__(256) void k_mv_select(DevSegments s, TimeRange range,
                                                   const unsigned long long *__restrict__ offsets,
                                                   const uint32_t *__restrict__ serial_ids, uint64_t n_serial,
                                                   uint32_t min_values, MvSeg *__restrict__ segs,
                                                   const unsigned long long *__restrict__ indexed_piece_base) {
    const uint64_t slot = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_serial) return;
    const uint32_t i = serial_ids[slot];
    SegInfo info = analyse_segment(s, i);
    if (range.enabled) apply_time_range(s, i, info, range);
    // (a segment the call's cursor index has pieces for is k_grid_mv_pieces': never long enough for this decoder)
    const bool indexed = indexed_piece_base != nullptr && indexed_piece_base[i + 1] > indexed_piece_base[i];
    segs[slot] = mv_describe(s, i, info, indexed ? 0xffffffffu : min_values, offsets[i]);
}

#ifdef MDB_MV_DEBUG
__device__ unsigned long long mv_debug_counters[4]; // iterations, steps, max iterations of a lane, max ticks of a lane
#endif

// ---- k_mv_chains: MV_CHAINS lanes per piece -----------------------------------------------------------------
//
// guesses[piece]: up to two candidate n (one per byte, MV_NO_LENGTH = none); tried[piece]: the n this
// piece has already been searched with (one per byte, low three bytes) and, in the top byte, 0x01
// once a chain found with a guessed n is kept; pending[r]: pieces without any chain after round r.
//
// Several chains can survive a piece: in a stream of `0` codes of one length a parse that is a few
// bits late keeps reading the (almost always zero) top bits of the window as control bits and never
// notices, and parses that start early can hop onto such a late grid through a `10` code. They
// cannot be told apart for certain locally, so each of the MV_CHAINS lanes of a piece tries every
// MV_CHAINS-th entry point and keeps the first chain that survives; the real parse is usually among
// them, and k_mv_links / k_mv_walk find out which. A lane tries its candidates as ONE loop in which
// it either picks the next candidate or advances the current one by a code.

__device__ __forceinline__ bool mv_byte_listed(uint32_t list, uint32_t value) {
    for (int k = 0; k < 3; k++)
        if (((list >> (8 * k)) & 0xffu) == value) return true;
    return false;
}

template <uint32_t STAGE_WORDS>
__global__ __launch_bounds__(MDB_WAVE) void k_mv_chains(const MvSeg *__restrict__ segs,
                                                        const unsigned long long *__restrict__ piece_base,
                                                        uint64_t n_slots, int round,
                                                        const uint32_t *__restrict__ guesses,
                                                        uint32_t *__restrict__ tried, uint32_t *__restrict__ pending,
                                                        MvRec *__restrict__ heads, MvChain *__restrict__ chains) {
    __shared__ uint32_t stage_lds[STAGE_WORDS][MDB_WAVE];
    const uint64_t lane_id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t piece = lane_id / MV_CHAINS;
    const uint32_t sub = (uint32_t)(lane_id % MV_CHAINS);
    const int kind = mv_round_kind(round);
    // No lane leaves before the ballot at the end; `present` lanes have a piece.
    const bool present = piece < piece_base[n_slots] && !(round > 1 && pending[round - 1] == 0);
    MvChain *__restrict__ mine = chains + piece * MV_CHAINS; // the piece's chains; this lane owns [sub]
    MvRec *__restrict__ head = heads + (piece * MV_CHAINS + sub) * MV_HEAD;
    bool work = false, slot_used = false, piece_has_chain = false;
    uint32_t p = 0, piece_begin = 0, piece_end = 0;
    MvSeg seg;
    seg.total_bits = 0;
    MvReader reader;
    uint32_t tried_here = 0xffffffffu;
    if (present) {
        const uint32_t slot = mv_slot_of(piece_base, n_slots, piece);
        seg = segs[slot];
        p = (uint32_t)(piece - piece_base[slot]);
        reader.open(seg);
        piece_begin = p * MV_PIECE_BITS;
        piece_end = min(piece_begin + MV_PIECE_BITS, seg.total_bits);
        if (kind == MV_ROUND_START) {
            mine[sub].n_head = 0;
            if (sub == 0) tried[piece] = 0xffffffffu;
            work = p == 0 && sub == 0 && seg.total_bits >= 32;
        } else {
            for (uint32_t c = 0; c < MV_CHAINS; c++) piece_has_chain = piece_has_chain || mine[c].n_head > 0;
            slot_used = mine[sub].n_head > 0;
            tried_here = tried[piece];
            if (kind == MV_ROUND_SCAN) {
                work = p > 0 && !piece_has_chain;
            } else {
                // A guess this piece has not been searched with yet; once a chain found with a guessed
                // window is kept the piece stops searching.
                bool untried = false;
                for (int g = 0; g < 2; g++) {
                    const uint32_t candidate = (guesses[piece] >> (8 * g)) & 0xffu;
                    untried = untried || (candidate <= 32u && !mv_byte_listed(tried_here, candidate));
                }
                work = p > 0 && untried && (tried_here >> 24) != 0x01u;
            }
        }
    }
    // Every lane of a piece walks over the same bits: each keeps its own copy (a column of the array).
    if (work) reader.template stage<STAGE_WORDS>(&stage_lds[0][threadIdx.x], piece_begin);

    int guess_index = -1;           // which guess is being searched (guess rounds)
    uint32_t length = MV_NO_LENGTH; // its n, MV_NO_LENGTH: candidates are `11` patterns / the real start
    uint32_t o = 0, o_end = 0;      // next and last+1 entry point of this lane
    bool searching = work, running = false, found_with_guess = false;
    MvTrack at = mv_track_at(0u, 0u);
    uint32_t steps = 0, n_head = 0;
    if (work && kind == MV_ROUND_START) {
        // The real start: 32 raw bits of the first value, then codes, no window yet.
        at = mv_track_at(32u, MV_NO_WINDOW);
        head[0] = {32u, MV_NO_WINDOW, 0u};
        n_head = 1;
        running = true;
    } else if (work && kind == MV_ROUND_SCAN) {
        o = piece_begin + sub;
        o_end = min(piece_begin + MV_SCAN_BITS, piece_end);
    }
#ifdef MDB_MV_DEBUG
    unsigned long long debug_iterations = 0, debug_steps = 0;
    const unsigned long long debug_t0 = wall_clock64();
#endif
    while (searching) {
#ifdef MDB_MV_DEBUG
        debug_iterations += 1;
        debug_steps += running ? 1 : 0;
#endif
        if (!running) {
            if (slot_used) {
                searching = false; // this lane's slot is taken: one chain per lane
            } else if (o >= o_end) {
                // The next guess, if any.
                length = MV_NO_LENGTH;
                while (kind == MV_ROUND_GUESS && ++guess_index < 2) {
                    const uint32_t candidate = (guesses[piece] >> (8 * guess_index)) & 0xffu;
                    if (candidate > 32u || mv_byte_listed(tried_here, candidate)) continue;
                    length = candidate;
                    break;
                }
                if (length == MV_NO_LENGTH) {
                    searching = false;
                } else {
                    tried_here = (tried_here & 0xff000000u) | ((tried_here << 8) & 0x00ffff00u) | length;
                    // The real parse enters the piece within its first 45 bits. The leading zeros of
                    // the guessed window are unknown (and not needed: see MvTrack).
                    o = piece_begin + sub;
                    o_end = min(piece_begin + MV_MAX_CODE_BITS, piece_end);
                }
            } else if (length == MV_NO_LENGTH) { // a scan: is there a plausible `11` code at o?
                if (o + 13 <= seg.total_bits) {
                    const uint32_t top = reader.peek(o, 13);
                    if ((top >> 11) == 3u && mv_valid_window((top >> 6) & 31u, top & 63u)) {
                        at = mv_track_at(o, MV_NO_WINDOW);
                        steps = 0;
                        n_head = 0;
                        running = true;
                    }
                }
                o += MV_CHAINS;
            } else {
                at = mv_track_at(o, length << 8);
                steps = 0;
                n_head = 0;
                running = true;
                o += MV_CHAINS;
            }
        } else {
            // A candidate is followed to the first boundary at or beyond the end of the piece and kept
            // unless it is malformed. A guessed chain may only fall in step with the real parse after
            // a few codes, so its boundaries are recorded once it has settled.
            bool finished = at.pos >= piece_end;
            if (!finished) {
                const int rc = mv_track_step(reader, at, n_head);
                if (rc == MV_MALFORMED) {
                    running = false;
                } else if (rc == MV_OVERRUN) {
                    finished = true; // only padding is left: the chain reaches the end
                } else {
                    steps += 1;
                    const bool guessed = length != MV_NO_LENGTH;
                    if ((!guessed || steps >= MV_SETTLE_CODES) && n_head < MV_HEAD)
                        head[n_head++] = {at.pos, at.state, at.count};
                    finished = at.pos >= piece_end;
                }
            }
            if (finished) {
                if (n_head > 0) {
                    mine[sub].n_head = n_head;
                    mine[sub].end = at;
                    slot_used = true;
                    found_with_guess = length != MV_NO_LENGTH;
                }
                running = false;
            }
        }
    }
#ifdef MDB_MV_DEBUG
    atomicAdd(&mv_debug_counters[0], debug_iterations);
    atomicAdd(&mv_debug_counters[1], debug_steps);
    atomicMax(&mv_debug_counters[2], debug_iterations);
    atomicMax(&mv_debug_counters[3], wall_clock64() - debug_t0);
#endif
    // Bookkeeping per piece: its MV_CHAINS lanes sit next to each other in the wave.
    const uint64_t group = 0xfull << (4u * (threadIdx.x / MV_CHAINS));
    static_assert(MV_CHAINS == 4, "the group mask above assumes four lanes per piece");
    // A chain that ends 1..MV_SHIFT_BITS bits behind another chain of the piece on the same grid
    // (same n, a whole number of `0` codes apart) is a late copy of it: it lives on the zero top bits
    // of the window, never meets the real parse and would only cost k_mv_links a long walk.
    {
        const bool kept_now = present && work && slot_used && n_head > 0 && found_with_guess;
        const uint32_t my_pos = at.pos, my_state = at.state;
        bool late_copy = false;
        for (int other = 0; other < MV_CHAINS; other++) {
            const int source = (int)(threadIdx.x / MV_CHAINS) * MV_CHAINS + other;
            const uint32_t their_pos = __shfl(my_pos, source, MDB_WAVE);
            const uint32_t their_state = __shfl(my_state, source, MDB_WAVE);
            const bool their_kept = __shfl((int)kept_now, source, MDB_WAVE) != 0;
            if (!kept_now || !their_kept || other == (int)sub) continue;
            if (mv_length(their_state) != mv_length(my_state) || mv_length(my_state) > 32u) continue;
            const uint32_t code_bits = 1u + mv_length(my_state);
            const uint32_t lag = (my_pos + code_bits * 256u - their_pos) % code_bits;
            late_copy = late_copy || (lag >= 1 && lag <= MV_SHIFT_BITS);
        }
        if (late_copy) {
            mine[sub].n_head = 0;
            slot_used = false;
            found_with_guess = false;
        }
    }
    const bool any_chain = (__ballot(present && slot_used) & group) != 0;
    const bool any_guessed = (__ballot(found_with_guess) & group) != 0;
    if (present && sub == 0) {
        if (work && kind == MV_ROUND_GUESS)
            tried[piece] = (tried_here & 0x00ffffffu) | (any_guessed || (tried_here >> 24) == 0x01u ? 0x01000000u : 0xff000000u);
        if (!any_chain) atomicAdd(&pending[round], 1u);
    }
}

// ---- k_mv_guess: one wave per entry of the serial list ----------------------------------------------------
//
// guesses[piece] = the n at the end of the chains of the nearest earlier piece of the stream that
// has chains (two of them, if its chains disagree).
__global__ __launch_bounds__(MDB_WAVE) void k_mv_guess(const MvSeg *__restrict__ segs,
                                                       const unsigned long long *__restrict__ piece_base,
                                                       const MvChain *__restrict__ chains,
                                                       uint32_t *__restrict__ guesses) {
    const uint64_t slot = blockIdx.x;
    const uint32_t n_pieces = segs[slot].n_pieces;
    if (n_pieces == 0) return;
    const int lane = threadIdx.x;
    const uint64_t first_piece = piece_base[slot];
    uint32_t carry = MV_NONE; // uniform: the answer of the last piece with chains in earlier groups of 64
    for (uint32_t base = 0; base < n_pieces; base += MDB_WAVE) {
        const uint32_t q = base + lane;
        uint32_t seen = MV_NONE;
        if (q < n_pieces) {
            const MvChain *__restrict__ theirs = chains + (first_piece + q) * MV_CHAINS;
            bool has_chain = false;
            for (int c = 0; c < MV_CHAINS; c++) has_chain = has_chain || theirs[c].n_head > 0;
            if (has_chain) {
                // The chains found last first: those come from guessed windows, which are right more
                // often than the survivors of a scan for `11` patterns.
                uint32_t a = MV_NO_LENGTH, b = MV_NO_LENGTH;
                for (int c = MV_CHAINS - 1; c >= 0; c--) {
                    if (theirs[c].n_head == 0) continue;
                    const uint32_t length = mv_length(theirs[c].end.state) & 0xffu;
                    if (a == MV_NO_LENGTH) a = length;
                    else if (b == MV_NO_LENGTH && length != a) b = length;
                }
                seen = 0xffff0000u | (b << 8) | a;
            }
        }
        // Inclusive "last one seen at or before this lane".
        uint32_t inclusive = seen;
#pragma unroll
        for (int delta = 1; delta < MDB_WAVE; delta <<= 1) {
            const uint32_t up = __shfl_up(inclusive, delta, MDB_WAVE);
            if (lane >= delta && inclusive == MV_NONE) inclusive = up;
        }
        uint32_t before = __shfl_up(inclusive, 1, MDB_WAVE);
        if (lane == 0 || before == MV_NONE) before = carry; // nothing earlier in this group of pieces
        if (q < n_pieces) guesses[first_piece + q] = before;
        const uint32_t last = __shfl(inclusive, MDB_WAVE - 1, MDB_WAVE);
        if (last != MV_NONE) carry = last;
    }
}

// ---- k_mv_links: one lane per chain ---------------------------------------------------------------------

__global__ __launch_bounds__(MDB_WAVE) void k_mv_links(const MvSeg *__restrict__ segs,
                                                       const unsigned long long *__restrict__ piece_base,
                                                       uint64_t n_slots, const MvRec *__restrict__ heads,
                                                       const MvChain *__restrict__ chains,
                                                       MvLink *__restrict__ links) {
    const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t piece = id / MV_CHAINS;
    if (piece >= piece_base[n_slots]) return;
    MvLink link;
    link.target = MV_NONE;
    link.into_head = 0;
    link.into_count = 0;
    link.from = mv_track_at(0u, 0u);
    const uint32_t n_head = chains[id].n_head;
    if (n_head > 0) {
        const uint32_t slot = mv_slot_of(piece_base, n_slots, piece);
        const MvSeg seg = segs[slot];
        const uint64_t first_piece = piece_base[slot];
        const uint32_t p = (uint32_t)(piece - first_piece);
        MvReader reader;
        reader.open(seg);
        MvTrack at = chains[id].end;
        const uint32_t tail_begin = at.pos;
        uint32_t partner = 0xffffffffu, partner_last_pos = 0;
        while (true) {
            const uint32_t r = at.pos / MV_PIECE_BITS;
            if (at.pos >= seg.total_bits || r >= seg.n_pieces) {
                link.target = MV_END;
                break;
            }
            if (r != partner) {
                // Only later pieces can be joined (a chain that ends in the padding of the last piece
                // still stands inside its own piece). Their recorded boundaries all lie near the start
                // of the piece: note where they end to stop looking early.
                partner = r;
                partner_last_pos = 0;
                if (r > p) {
                    for (uint32_t k = 0; k < MV_CHAINS; k++) {
                        const uint32_t n = chains[(first_piece + r) * MV_CHAINS + k].n_head;
                        if (n == 0) continue;
                        const uint32_t last = heads[((first_piece + r) * MV_CHAINS + k) * MV_HEAD + n - 1].pos;
                        partner_last_pos = max(partner_last_pos, last + 1);
                    }
                }
            }
            if (at.pos < partner_last_pos) {
                bool joined = false;
                for (uint32_t k = 0; k < MV_CHAINS && !joined; k++) {
                    const uint64_t other = (first_piece + r) * MV_CHAINS + k;
                    const uint32_t n = chains[other].n_head;
                    if (n == 0) continue;
                    for (uint32_t h = 0; h < n; h++) {
                        const MvRec rec = heads[other * MV_HEAD + h];
                        if (rec.pos != at.pos || mv_length(rec.state) != mv_length(at.state)) continue;
                        link.target = (uint32_t)other;
                        link.into_head = h;
                        link.into_count = rec.count;
                        joined = true;
                        break;
                    }
                }
                if (joined) break;
            }
            if (at.pos - tail_begin > MV_MAX_TAIL_BITS) break; // MV_NONE: the sequential decoder takes over
            const int rc = mv_track_step(reader, at, n_head);
            if (rc == MV_MALFORMED) break;
            if (rc == MV_OVERRUN) {
                link.target = MV_END;
                break;
            }
        }
        link.from = at;
    }
    links[id] = link;
}

// ---- k_mv_walk: one wave per entry of the serial list -------------------------------------------------------

__global__ __launch_bounds__(MDB_WAVE) void k_mv_walk(MvSeg *__restrict__ segs,
                                                      const unsigned long long *__restrict__ piece_base,
                                                      const MvChain *__restrict__ chains,
                                                      const MvLink *__restrict__ links,
                                                      MvStart *__restrict__ starts) {
    const uint64_t slot = blockIdx.x;
    const MvSeg seg = segs[slot];
    if (seg.n_pieces == 0) return;
    const int lane = threadIdx.x;
    const uint64_t first_piece = piece_base[slot];
    const uint64_t first_chain = first_piece * MV_CHAINS;
    const uint32_t n_ids = seg.n_pieces * MV_CHAINS;
    MvReader reader;
    reader.open(seg);
    bool ok = seg.total_bits >= 32 && seg.n_model >= 1 && chains[first_chain].n_head > 0;
    uint32_t q = 0;           // chain on the real parse, relative to first_chain (uniform)
    uint32_t first_index = 1; // value 0 is the raw first value
    uint32_t value_bits = reader.peek(0, 32);
    // Where the real parse entered chain q (position, real window, which recorded boundary of the
    // chain that is, how many values the chain had decoded there).
    uint32_t pos = 32, state = MV_NO_WINDOW, entered_head = 0, entered_count = 0;
    uint32_t window_first = 0;
    bool window_valid = false;
    MvLink window;
    window.target = MV_NONE;
    window.into_head = window.into_count = 0;
    window.from = mv_track_at(0u, 0u);
    while (ok) {
        if (!window_valid || q - window_first >= MDB_WAVE) {
            window_first = q;
            if (q + lane < n_ids) window = links[first_chain + q + lane];
            window_valid = true;
        }
        const int source = (int)(q - window_first);
        const uint32_t target = __shfl(window.target, source, MDB_WAVE);
        const uint32_t into_head = __shfl(window.into_head, source, MDB_WAVE);
        const uint32_t into_count = __shfl(window.into_count, source, MDB_WAVE);
        const uint32_t from_pos = __shfl(window.from.pos, source, MDB_WAVE);
        const uint32_t from_state = __shfl(window.from.state, source, MDB_WAVE);
        const uint32_t from_count = __shfl(window.from.count, source, MDB_WAVE);
        const uint32_t from_x = __shfl(window.from.x, source, MDB_WAVE);
        const uint32_t from_seen = __shfl(window.from.seen, source, MDB_WAVE);
        uint32_t from_raw = 0, from_snap = 0;
#pragma unroll
        for (uint32_t h = 0; h < MV_HEAD; h++) {
            const uint32_t raw_h = __shfl(window.from.raw[h], source, MDB_WAVE);
            const uint32_t snap_h = __shfl(window.from.snap[h], source, MDB_WAVE);
            if (h == entered_head) {
                from_raw = raw_h;
                from_snap = snap_h;
            }
        }
        if (target == MV_NONE) {
            ok = false;
            break;
        }
        const uint32_t left = seg.n_model - first_index;
        uint32_t n_values = target == MV_END ? left : from_count - entered_count;
        const bool last = target == MV_END || n_values >= left;
        if (n_values > left) n_values = left;
        if (lane == 0) starts[first_piece + q / MV_CHAINS] = {1u, pos, state, first_index, value_bits, n_values};
        if (last) break;
        const uint32_t next = target - (uint32_t)first_chain;
        if (next / MV_CHAINS <= q / MV_CHAINS || next >= n_ids) { // links only ever point to later pieces
            ok = false;
            break;
        }
        // The XOR of the values decoded between the two boundaries (see MvTrack): the `0` codes up
        // to the first `11` code carry bits that the REAL window at the entry shifts; from that `11`
        // on the chain's own shifts were real.
        const bool seen = (from_seen >> entered_head) & 1u;
        uint32_t delta = 0;
        if (from_raw != 0) {
            if (state == MV_NO_WINDOW) { // cannot happen: `0` codes need a window
                ok = false;
                break;
            }
            delta = mv_shifted(from_raw, state);
        }
        if (seen) {
            delta ^= from_x ^ from_snap;
            state = from_state; // set by a `11` code the real parse has read too
        }
        if (mv_length(state) != mv_length(from_state)) { // cannot happen: same positions, same n
            ok = false;
            break;
        }
        value_bits ^= delta;
        first_index += n_values;
        pos = from_pos;
        entered_head = into_head;
        entered_count = into_count;
        q = next;
    }
    if (lane == 0) segs[slot].done = ok ? 1u : 0u;
}

// ---- k_mv_decode: one lane per piece -----------------------------------------------------------------------

template <uint32_t STAGE_WORDS>
__global__ __launch_bounds__(MDB_WAVE) void k_mv_decode(const MvSeg *__restrict__ segs,
                                                        const unsigned long long *__restrict__ piece_base,
                                                        uint64_t n_slots, const MvStart *__restrict__ starts,
                                                        float *__restrict__ out_val,
                                                        unsigned int *__restrict__ error) {
    __shared__ uint32_t stage_lds[STAGE_WORDS][MDB_WAVE];
    const uint64_t piece = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (piece >= piece_base[n_slots]) return;
    const MvStart start = starts[piece];
    if (!start.valid) return;
    const uint32_t slot = mv_slot_of(piece_base, n_slots, piece);
    const MvSeg seg = segs[slot];
    if (!seg.done) return; // the sequential decoder handles this stream
    MvReader reader;
    reader.open(seg);
    reader.template stage<STAGE_WORDS>(&stage_lds[0][threadIdx.x], start.pos);
    float *__restrict__ out = out_val + seg.out_offset;
    uint32_t value = start.value_bits;
    if (piece == piece_base[slot] && seg.first == 0) out[0] = __uint_as_float(value); // the raw first value
    uint32_t pos = start.pos, state = start.state, index = start.first_index;
    for (uint32_t k = 0; k < start.n_values; k++, index++) {
        uint32_t kind = 0, bits = 0;
        if (mv_step(reader, pos, state, kind, bits) != MV_OK) {
            atomicOr(error, ERR_BITSTREAM);
            return;
        }
        if (kind != MV_CODE_REPEAT) value ^= mv_shifted(bits, state);
        if (index >= seg.first && index < seg.visible_end) out[index - seg.first] = __uint_as_float(value);
    }
}

} // namespace mdb
