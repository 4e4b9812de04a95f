// Microbenchmark: how fast can MI355X stream-store the grid output pattern (8 B timestamp + 4 B
// value per point) with no computation at all? Gives the ceiling k_grid_tiles is measured against.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// V0: plain float4 fill of one array (12 B/point worth of bytes in total).
__global__ __launch_bounds__(256) void k_fill16(float4 *out, uint64_t n16) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256)
        out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
// V1: tile kernel shape: per lane 4 points: 2 x 16 B ts at 32 B stride + 1 x 16 B value; tile per block.
__global__ __launch_bounds__(256) void k_tiles_strided(int64_t *ts, float *val, uint64_t n) {
    uint64_t tile = (uint64_t)blockIdx.x * 4096;
    for (int j = 0; j < 4; j++) {
        uint64_t p = tile + j * 1024 + threadIdx.x * 4;
        if (p + 4 > n) return;
        longlong2 *t = reinterpret_cast<longlong2 *>(ts + p);
        t[0] = make_longlong2(p, p + 1);
        t[1] = make_longlong2(p + 2, p + 3);
        *reinterpret_cast<float4 *>(val + p) = make_float4(1.f, 2.f, 3.f, 4.f);
    }
}
// V2: ts stores contiguous per wave instruction (lane l writes 16 B at l*16, then +1 KiB).
__global__ __launch_bounds__(256) void k_tiles_contig(int64_t *ts, float *val, uint64_t n) {
    uint64_t tile = (uint64_t)blockIdx.x * 4096;
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = 0; j < 4; j++) {
        uint64_t wave_base = tile + j * 1024 + wave * 256; // 256 points per wave
        if (wave_base + 256 > n) return;
        longlong2 *t = reinterpret_cast<longlong2 *>(ts + wave_base);
        uint64_t p0 = wave_base + lane * 2, p1 = wave_base + 128 + lane * 2;
        t[lane] = make_longlong2(p0, p0 + 1);
        t[64 + lane] = make_longlong2(p1, p1 + 1);
        *reinterpret_cast<float4 *>(val + wave_base + lane * 4) = make_float4(1.f, 2.f, 3.f, 4.f);
    }
}
// V3: like V1 but 8192-point tiles, 2 x more work per block.
__global__ __launch_bounds__(256) void k_tiles_big(int64_t *ts, float *val, uint64_t n) {
    uint64_t tile = (uint64_t)blockIdx.x * 16384;
    for (int j = 0; j < 16; j++) {
        uint64_t p = tile + j * 1024 + threadIdx.x * 4;
        if (p + 4 > n) return;
        longlong2 *t = reinterpret_cast<longlong2 *>(ts + p);
        t[0] = make_longlong2(p, p + 1);
        t[1] = make_longlong2(p + 2, p + 3);
        *reinterpret_cast<float4 *>(val + p) = make_float4(1.f, 2.f, 3.f, 4.f);
    }
}
// V4: nontemporal stores
__global__ __launch_bounds__(256) void k_tiles_nt(int64_t *ts, float *val, uint64_t n) {
    uint64_t tile = (uint64_t)blockIdx.x * 4096;
    for (int j = 0; j < 4; j++) {
        uint64_t p = tile + j * 1024 + threadIdx.x * 4;
        if (p + 4 > n) return;
        long long *t = reinterpret_cast<long long *>(ts + p);
        __builtin_nontemporal_store((long long)p, t);
        __builtin_nontemporal_store((long long)p + 1, t + 1);
        __builtin_nontemporal_store((long long)p + 2, t + 2);
        __builtin_nontemporal_store((long long)p + 3, t + 3);
        float *v = val + p;
        __builtin_nontemporal_store(1.f, v);
        __builtin_nontemporal_store(2.f, v + 1);
        __builtin_nontemporal_store(3.f, v + 2);
        __builtin_nontemporal_store(4.f, v + 3);
    }
}

int main() {
    const uint64_t n = 2000000000ull; // points
    int64_t *ts; float *val;
    CHECK(hipMalloc(&ts, n * 8));
    CHECK(hipMalloc(&val, n * 4));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    auto time = [&](const char *name, auto launch) {
        launch(); hipDeviceSynchronize();
        float best = 1e9f, sum = 0;
        for (int r = 0; r < 5; r++) {
            hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best; sum += ms;
        }
        printf("%-18s best %.3f ms  avg %.3f ms -> %.1f GB/s (best) %.1f GB/s (avg)\n", name, best, sum / 5,
               12.0 * n / best / 1e6, 12.0 * n / (sum / 5) / 1e6);
    };
    time("fill16 (1 array)", [&] { hipLaunchKernelGGL(k_fill16, dim3(256 * 16), dim3(256), 0, 0, reinterpret_cast<float4 *>(ts), n * 8 / 16); hipLaunchKernelGGL(k_fill16, dim3(256 * 16), dim3(256), 0, 0, reinterpret_cast<float4 *>(val), n * 4 / 16); });
    time("tiles strided", [&] { hipLaunchKernelGGL(k_tiles_strided, dim3((n + 4095) / 4096), dim3(256), 0, 0, ts, val, n); });
    time("tiles contiguous", [&] { hipLaunchKernelGGL(k_tiles_contig, dim3((n + 4095) / 4096), dim3(256), 0, 0, ts, val, n); });
    time("tiles 16384", [&] { hipLaunchKernelGGL(k_tiles_big, dim3((n + 16383) / 16384), dim3(256), 0, 0, ts, val, n); });
    time("tiles nontemporal", [&] { hipLaunchKernelGGL(k_tiles_nt, dim3((n + 4095) / 4096), dim3(256), 0, 0, ts, val, n); });
    return 0;
}
