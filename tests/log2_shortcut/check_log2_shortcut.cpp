// TEST INFRASTRUCTURE. The shortcut rewrite_least_mantissa_bits takes for floor(|log2 x|)
// (modelardb-rs_amd/csrc/mdb_floor_log2.hpp, included by the kernels of mdb_fit.hip) against the definition the oracle
// and the kernels' slow path use, floorf(fabsf((float)log2((double)x))) (the reference: `f32::log2`, macaque_v.rs:185):
//   1. EVERY float the shortcut accepts: all 254 normal exponents x the 2^23 - 512 fractions it takes (2.13 x 10^9);
//   2. that it declines everything else: the 512 fractions around each power of two, zeros, subnormals, infinities,
//      NaNs, every negative value (sampled) - those take the function itself in the kernel, so nothing is to compare.
// Prints "ok: <n> values" or the first mismatches.
#include "../../modelardb-rs_amd/csrc/mdb_floor_log2.hpp"

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

static float from_bits(uint32_t bits) {
    float x;
    std::memcpy(&x, &bits, 4);
    return x;
}

int main() {
    const unsigned n_threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::atomic<unsigned long long> checked{0}, mismatches{0}, wrongly_taken{0};
    std::vector<std::thread> threads;
    for (unsigned t = 0; t < n_threads; t++) {
        threads.emplace_back([&, t] {
            unsigned long long mine = 0;
            for (uint32_t exponent = 1 + t; exponent <= 254; exponent += n_threads) {
                for (uint32_t fraction = 0; fraction < (1u << 23); fraction++) {
                    const uint32_t bits = (exponent << 23) | fraction;
                    float magnitude = -1.0f;
                    const bool taken = mdb::floor_abs_log2_from_exponent(bits, &magnitude);
                    const bool edge = fraction < mdb::FLOOR_LOG2_EDGE_STEPS || fraction >= (1u << 23) - mdb::FLOOR_LOG2_EDGE_STEPS;
                    if (taken == edge) wrongly_taken++;
                    if (!taken) continue;
                    const float expected = std::floor(std::fabs((float)std::log2((double)from_bits(bits))));
                    mine++;
                    if (magnitude != expected && mismatches++ < 10)
                        std::printf("MISMATCH bits %08x: shortcut %g, floor|log2| %g\n", bits, magnitude, expected);
                }
            }
            checked += mine;
        });
    }
    for (auto &thread : threads) thread.join();
    // what must be declined: zeros, subnormals, infinities, NaNs, and anything negative
    unsigned long long declined = 0;
    float unused;
    for (uint32_t fraction = 0; fraction < (1u << 23); fraction += 4099) {
        for (uint32_t exponent : {0u, 255u}) {
            if (mdb::floor_abs_log2_from_exponent((exponent << 23) | fraction, &unused)) wrongly_taken++;
            declined++;
        }
        for (uint32_t exponent = 0; exponent <= 255; exponent += 5) {
            if (mdb::floor_abs_log2_from_exponent(0x80000000u | (exponent << 23) | fraction, &unused)) wrongly_taken++;
            declined++;
        }
    }
    if (mismatches.load() || wrongly_taken.load()) {
        std::printf("FAILED: %llu mismatches, %llu values taken or declined wrongly\n", mismatches.load(), wrongly_taken.load());
        return 1;
    }
    std::printf("ok: %llu values the shortcut takes equal floor|log2|, %llu it must decline declined\n", checked.load(), declined);
    return 0;
}
