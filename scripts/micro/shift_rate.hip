// How fast are 64-bit variable shifts next to 32-bit ones and funnel shifts on gfx950? (The bit
// readers of the delta-of-delta and MacaqueV decoders live on them.) Throughput with all SIMDs full.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int ITER = 4096;

__global__ void k_shift64(uint64_t *out, uint32_t s) {
    uint64_t a = threadIdx.x + 1, b = a * 3, c = a * 5, d = a * 7;
    for (int i = 0; i < ITER; i++) {
        a = (a << (s & 31)) | 1; b = (b >> (s & 15)) + a; c = (c << ((s + 1) & 31)) | 3; d = (d >> ((s + 2) & 7)) + c;
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
}
__global__ void k_shift32(uint32_t *out, uint32_t s) {
    uint32_t a = threadIdx.x + 1, b = a * 3, c = a * 5, d = a * 7;
    for (int i = 0; i < ITER; i++) {
        a = (a << (s & 31)) | 1; b = (b >> (s & 15)) + a; c = (c << ((s + 1) & 31)) | 3; d = (d >> ((s + 2) & 7)) + c;
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
}
__global__ void k_alignbit(uint32_t *out, uint32_t s) {
    uint32_t a = threadIdx.x + 1, b = a * 3, c = a * 5, d = a * 7;
    for (int i = 0; i < ITER; i++) {
        a = __builtin_amdgcn_alignbit(a, b, s & 31) | 1; b = __builtin_amdgcn_alignbit(b, c, (s + 1) & 31) + a;
        c = __builtin_amdgcn_alignbit(c, d, (s + 2) & 31) | 3; d = __builtin_amdgcn_alignbit(d, a, (s + 3) & 31) + c;
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
}

template <typename F> float time_ms(F launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

int main() {
    const int blocks = 256 * 32, threads = 256;
    void *out; hipMalloc(&out, (size_t)blocks * threads * 8);
    const double waves = (double)blocks * threads / 64;
    auto report = [&](const char *name, float ms, double ops_per_iter) {
        // wave-instructions per SIMD per cycle, assuming 2.4 GHz and 1024 SIMDs
        double wave_ops = waves * ITER * ops_per_iter;
        printf("%-10s %.3f ms  %.2f cycles per wave-op per SIMD\n", name, ms, ms * 1e-3 * 2.4e9 * 1024 / wave_ops);
    };
    report("shift64", time_ms([&] { hipLaunchKernelGGL(k_shift64, dim3(blocks), dim3(threads), 0, 0, (uint64_t *)out, 5u); }), 8);
    report("shift32", time_ms([&] { hipLaunchKernelGGL(k_shift32, dim3(blocks), dim3(threads), 0, 0, (uint32_t *)out, 5u); }), 8);
    report("alignbit", time_ms([&] { hipLaunchKernelGGL(k_alignbit, dim3(blocks), dim3(threads), 0, 0, (uint32_t *)out, 5u); }), 8);
    return 0;
}
