// d2h_rate.hip - how fast does a 70 MB block (the reconstructed points of one 8 192-segment batch) reach
// page-locked host memory? hipHostMalloc flavours, one stream and two streams at once, and a kernel that
// writes straight into mapped host memory. Development microbenchmark behind DESIGN.md's host-path numbers.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_copy(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = in[i];
}

static double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    const size_t bytes = 70u << 20;
    const int reps = 20;
    void *dev[2];
    for (auto &d : dev) { CHECK(hipMalloc(&d, bytes)); CHECK(hipMemset(d, 1, bytes)); }
    hipStream_t streams[2];
    for (auto &s : streams) CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    struct Flavour { const char *name; unsigned flags; };
    const Flavour flavours[] = {{"default", hipHostMallocDefault}, {"non-coherent", hipHostMallocNonCoherent},
                                {"coherent", hipHostMallocCoherent}, {"portable|mapped", hipHostMallocPortable | hipHostMallocMapped},
                                {"write-combined", hipHostMallocWriteCombined}, {"numa-user", hipHostMallocNumaUser}};
    for (const Flavour &f : flavours) {
        void *host[2];
        bool ok = true;
        for (auto &h : host) ok = ok && hipHostMalloc(&h, bytes, f.flags) == hipSuccess;
        if (!ok) { printf("%-16s: hipHostMalloc refused\n", f.name); (void)hipGetLastError(); continue; }
        for (auto &h : host) memset(h, 0, bytes);
        CHECK(hipMemcpyAsync(host[0], dev[0], bytes, hipMemcpyDeviceToHost, streams[0]));
        CHECK(hipStreamSynchronize(streams[0]));
        double t0 = now();
        for (int r = 0; r < reps; r++) CHECK(hipMemcpyAsync(host[0], dev[0], bytes, hipMemcpyDeviceToHost, streams[0]));
        CHECK(hipStreamSynchronize(streams[0]));
        double one = (now() - t0) / reps;
        t0 = now();
        for (int r = 0; r < reps; r++)
            for (int s = 0; s < 2; s++) CHECK(hipMemcpyAsync(host[s], dev[s], bytes, hipMemcpyDeviceToHost, streams[s]));
        for (auto &s : streams) CHECK(hipStreamSynchronize(s));
        double two = (now() - t0) / (2 * reps);
        // a kernel storing into the mapped block
        void *mapped = nullptr;
        double kernel = 0;
        if (hipHostGetDevicePointer(&mapped, host[0], 0) == hipSuccess) {
            k_copy<<<1024, 256, 0, streams[0]>>>((const uint4 *)dev[0], (uint4 *)mapped, bytes / 16);
            CHECK(hipStreamSynchronize(streams[0]));
            t0 = now();
            for (int r = 0; r < reps; r++) k_copy<<<1024, 256, 0, streams[0]>>>((const uint4 *)dev[0], (uint4 *)mapped, bytes / 16);
            CHECK(hipStreamSynchronize(streams[0]));
            kernel = (now() - t0) / reps;
        } else (void)hipGetLastError();
        // host read rate of the block (what the consumer of the points pays)
        t0 = now();
        unsigned long long sum = 0;
        for (size_t i = 0; i < bytes / 8; i++) sum += ((const unsigned long long *)host[0])[i];
        double read = now() - t0;
        printf("%-16s: 1 stream %.1f GB/s, 2 streams %.1f GB/s (aggregate), kernel store %.1f GB/s, host read %.1f GB/s (%llu)\n",
               f.name, bytes / one / 1e9, bytes / two / 1e9, kernel > 0 ? bytes / kernel / 1e9 : 0.0, bytes / read / 1e9, sum & 1);
        for (auto &h : host) CHECK(hipHostFree(h));
    }
    // malloc + hipHostRegister
    void *plain = aligned_alloc(1 << 21, bytes);
    memset(plain, 0, bytes);
    if (hipHostRegister(plain, bytes, hipHostRegisterDefault) == hipSuccess) {
        CHECK(hipMemcpyAsync(plain, dev[0], bytes, hipMemcpyDeviceToHost, streams[0]));
        CHECK(hipStreamSynchronize(streams[0]));
        double t0 = now();
        for (int r = 0; r < reps; r++) CHECK(hipMemcpyAsync(plain, dev[0], bytes, hipMemcpyDeviceToHost, streams[0]));
        CHECK(hipStreamSynchronize(streams[0]));
        printf("registered malloc : 1 stream %.1f GB/s\n", bytes / ((now() - t0) / reps) / 1e9);
    }
    return 0;
}
