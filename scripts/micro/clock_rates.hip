// clock_rates: the device's reported clock rates, and the shader clock measured against s_memrealtime and the host's
// clock while a kernel spins (development tool).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(unsigned long long *out, unsigned long long cycles) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    double x = threadIdx.x;
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) x = x * 1.0000001 + 0.5;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = __builtin_amdgcn_s_memtime() - t0;
        out[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    if (x == 12345.678) out[2] = 1;
}
int main() {
    int clock_khz = 0, wall_khz = 0, mem_khz = 0;
    hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeClockRate, 0);
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    hipDeviceGetAttribute(&mem_khz, hipDeviceAttributeMemoryClockRate, 0);
    std::printf("clockRate %d kHz, wallClockRate %d kHz, memoryClockRate %d kHz\n", clock_khz, wall_khz, mem_khz);
    unsigned long long *out;
    hipMalloc(&out, 64);
    for (int blocks : {1, 256, 2048, 8192}) {
        hipMemset(out, 0, 64);
        hipDeviceSynchronize();
        const auto h0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, 0, out, 400000000ull);
        hipDeviceSynchronize();
        const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - h0).count();
        unsigned long long host[2];
        hipMemcpy(host, out, 16, hipMemcpyDeviceToHost);
        std::printf("%5d blocks: %.1f ms on the host; s_memtime %llu ticks = %.1f MHz; s_memrealtime %llu ticks = %.2f MHz\n", blocks,
                    seconds * 1e3, host[0], host[0] / seconds / 1e6, host[1], host[1] / seconds / 1e6);
    }
    return 0;
}
