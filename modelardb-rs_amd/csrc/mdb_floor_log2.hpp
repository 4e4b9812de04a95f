// mdb_floor_log2.hpp - floor(|log2 x|) of a float from its exponent field, where that is certain.
// Used by rewrite_least_mantissa_bits (mdb_fit.hip; the reference: crates/modelardb_compression/src/models/
// macaque_v.rs:168-196, `23 - (int)floorf(fabsf(log2f(factorized_epsilon)))`) and checked exhaustively on the CPU by
// tests/log2_shortcut/check_log2_shortcut.cpp against (float)log2((double)x), the oracle's definition of log2f.
#pragma once

#include <cstdint>

#ifdef __HIPCC__
#define MDB_FLOOR_LOG2_FN __host__ __device__ inline __attribute__((always_inline))
#else
#define MDB_FLOOR_LOG2_FN inline
#endif

namespace mdb {

// Steps of 2^-23 a fraction has to keep from 1 and from 2 for the shortcut to be taken.
constexpr uint32_t FLOOR_LOG2_EDGE_STEPS = 256u;

// For x = 2^E m (1 <= m < 2, positive, normal) floor(|log2 x|) is E (E >= 0) or -E - 1 (E < 0) unless the logarithm lies
// so close to a whole number that its rounding to f32 reaches it: farther than 256 of m's steps of 2^-23 from 1 and from
// 2 the logarithm is more than 2 x 10^-5 from E and E + 1, a float of its size (below 150) less than 8 x 10^-6 from
// it. Returns false for the values in between (one in 16 000), zeros, subnormals, infinities, NaNs and negative
// values: they take the function itself.
MDB_FLOOR_LOG2_FN bool floor_abs_log2_from_exponent(uint32_t bits, float *magnitude) {
    const uint32_t exponent = (bits >> 23) & 0xffu, fraction = bits & 0x7fffffu;
    if ((bits >> 31) != 0u || exponent == 0u || exponent == 255u || fraction < FLOOR_LOG2_EDGE_STEPS ||
        fraction >= (1u << 23) - FLOOR_LOG2_EDGE_STEPS)
        return false;
    const int e = (int)exponent - 127;
    *magnitude = (float)(e >= 0 ? e : -e - 1);
    return true;
}

} // namespace mdb
