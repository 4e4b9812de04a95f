// mdb_synth.hip - on-device generator of the benchmark's synthetic series:
//   v(s, i) = 100 + 10 sin(2 pi i / P_s + phi_s) + u,  P_s = 2000 + 37 (s mod 64),
//   phi_s = 2 pi frac(0.61803 s),  u ~ U(-0.05, 0.05) from splitmix64(seed ^ s << 40 ^ i),
// evaluated in f64 and rounded to f32. Timestamps are regular (T0 = 0, 1000 us) and are never
// materialised for the fit benchmark. Not on the reference's path: it only feeds bench.py.
#include "mdb_common.hpp"

namespace mdb {

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void k_synth_values(float *__restrict__ out, uint64_t first_series,
                                                      uint64_t n_per_series, uint64_t total,
                                                      uint64_t seed) {
    const double two_pi = 6.283185307179586476925286766559;
    for (uint64_t e = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4; e < total;
         e += (uint64_t)gridDim.x * 256 * 4) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint64_t idx = e + k;
            uint64_t series = first_series + idx / n_per_series;
            uint64_t i = idx % n_per_series;
            double period = 2000.0 + 37.0 * (double)(series % 64);
            double frac = (double)series * 0.61803;
            frac -= floor(frac);
            double phase = two_pi * frac;
            uint64_t h = splitmix64(seed ^ (series << 40) ^ i);
            double u = ((double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 0.1;
            v[k] = (float)(100.0 + 10.0 * sin(two_pi * (double)i / period + phase) + u);
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
    std::lock_guard<std::mutex> lock(ctx->mutex);
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
