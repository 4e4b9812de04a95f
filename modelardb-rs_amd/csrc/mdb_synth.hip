// mdb_synth.hip - on-device generator of the benchmark's synthetic series:
//   v(s, i) = 100 + 10 sin(2 pi (i / P_s + f_s)) + u,  P_s = 2000 + 37 (s mod 64),
//   f_s = frac(0.61803 s),  u ~ U(-0.05, 0.05) from splitmix64(seed ^ s << 40 ^ i),
// evaluated in f64 and rounded to f32. The sine is a FIXED polynomial over IEEE +, *, /, floor only
// (no libm call, no fused multiply-add: the library is built with -ffp-contract=off), so that
// tests/datagen.bench_series - the same operations in numpy - produces the same bits on the host:
// the oracle can be run on exactly the bytes the bench fits (tests/test_gpu_fit.py checks this bit
// for bit). Timestamps are regular (T0 = 0, 1000 us) and are never materialised for the fit
// benchmark. Not on the reference's path: it only feeds bench.py.
#include "mdb_common.hpp"

namespace mdb {

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}


// sin(2 pi turns) for turns in [0, 1): the quarter turn it falls into picks +-sin / +-cos of an
// angle in [0, pi/2), both as Taylor polynomials in Horner form (error < 1e-11, far below f32).
__device__ __forceinline__ double synth_sine_of_turns(double turns) {
    const double quarters = turns * 4.0;
    const double quadrant = floor(quarters);
    const double x = (quarters - quadrant) * 1.5707963267948966;
    const double x2 = x * x;
    double s = -7.6471637318198164759e-13;          // -1/15!
    s = s * x2 + 1.6059043836821614599e-10;         //  1/13!
    s = s * x2 + -2.5052108385441718775e-08;        // -1/11!
    s = s * x2 + 2.7557319223985890653e-06;         //  1/9!
    s = s * x2 + -1.9841269841269841270e-04;        // -1/7!
    s = s * x2 + 8.3333333333333333333e-03;         //  1/5!
    s = s * x2 + -1.6666666666666666667e-01;        // -1/3!
    s = s * x2 + 1.0;
    s = s * x;
    double c = 4.7794773323873852974e-14;           //  1/16!
    c = c * x2 + -1.1470745597729724714e-11;        // -1/14!
    c = c * x2 + 2.0876756987868098979e-09;         //  1/12!
    c = c * x2 + -2.7557319223985890653e-07;        // -1/10!
    c = c * x2 + 2.4801587301587301587e-05;         //  1/8!
    c = c * x2 + -1.3888888888888888889e-03;        // -1/6!
    c = c * x2 + 4.1666666666666666667e-02;         //  1/4!
    c = c * x2 + -0.5;
    c = c * x2 + 1.0;
    if (quadrant == 0.0) return s;
    if (quadrant == 1.0) return c;
    if (quadrant == 2.0) return -s;
    return -c;
}

__device__ __forceinline__ float synth_value(uint64_t series, uint64_t i, uint64_t seed) {
    const double period = 2000.0 + 37.0 * (double)(series % 64);
    double fraction = (double)series * 0.61803;
    fraction -= floor(fraction);
    double turns = (double)i / period + fraction;
    turns -= floor(turns);
    const uint64_t h = splitmix64(seed ^ (series << 40) ^ i);
    const double u = ((double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 0.1;
    return (float)(100.0 + 10.0 * synth_sine_of_turns(turns) + u);
}

__global__ __launch_bounds__(256) void k_synth_values(float *__restrict__ out, uint64_t first_series,
                                                      uint64_t n_per_series, uint64_t total,
                                                      uint64_t seed) {
    for (uint64_t e = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4; e < total;
         e += (uint64_t)gridDim.x * 256 * 4) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint64_t idx = e + k;
            const uint64_t series = first_series + idx / n_per_series;
            const uint64_t i = idx % n_per_series;
            v[k] = synth_value(series, i, seed);
        }
        if (e + 4 <= total) {
            *reinterpret_cast<float4 *>(out + e) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            for (int k = 0; k < 4 && e + k < total; k++) out[e + k] = v[k];
        }
    }
}

} // namespace mdb

using namespace mdb;

extern "C" int mdb_synth_values_dev(mdb_ctx *ctx, float *out, uint64_t first_series,
                                    uint64_t n_series, uint64_t n_per_series, uint64_t seed) {
    if (!ctx || !out) return fail("ctx and out must not be NULL.");
    mdb::CallGuard lock(ctx);
    MDB_HIP_CHECK(hipSetDevice(ctx->device));
    const uint64_t total = n_series * n_per_series;
    if (total == 0) return 0;
    if (reinterpret_cast<uintptr_t>(out) & 15u) return fail("out must be 16-byte aligned.");
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((total / 4 + 255) / 256 + 1, 256 * 16);
    {
        LaunchTimer timer(ctx, "k_synth_values");
        hipLaunchKernelGGL(k_synth_values, dim3(blocks), dim3(256), 0, ctx->stream, out, first_series,
                           n_per_series, total, seed);
    }
    MDB_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    MDB_HIP_CHECK(hipGetLastError());
    return 0;
}
