// What the host threads of mdb_compress_chunk_list can move (the gather: values copied into a staging block, timestamps
// scanned for equal spacing), by where the threads sit and how they store. g++ -O3 -pthread (baseline x86-64: SSE2 streaming stores, as the library is built).
// usage: host_gather_bench [points=160000000] [threads=16]
#include <immintrin.h>
#include <sched.h>
#include <pthread.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static void pin(int cpu) {
    if (cpu < 0) return;
    cpu_set_t set;
    CPU_ZERO(&set);
    CPU_SET(cpu, &set);
    pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
}

static void copy_nt(float *to, const float *from, size_t n) {
    size_t i = 0;
    while (i < n && (reinterpret_cast<uintptr_t>(to + i) & 63)) { to[i] = from[i]; i++; }
    for (; i + 16 <= n; i += 16) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(from + i));
        const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i *>(from + i + 4));
        const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i *>(from + i + 8));
        const __m128i d = _mm_loadu_si128(reinterpret_cast<const __m128i *>(from + i + 12));
        _mm_stream_si128(reinterpret_cast<__m128i *>(to + i), a);
        _mm_stream_si128(reinterpret_cast<__m128i *>(to + i + 4), b);
        _mm_stream_si128(reinterpret_cast<__m128i *>(to + i + 8), c);
        _mm_stream_si128(reinterpret_cast<__m128i *>(to + i + 12), d);
    }
    for (; i < n; i++) to[i] = from[i];
    _mm_sfence();
}

static bool regular(const int64_t *t, size_t n) {
    if (n < 3) return true;
    const int64_t step = t[1] - t[0];
    bool differs = false;
    for (size_t j = 2; j < n; j++) differs |= (t[j] - t[j - 1]) != step;
    return !differs;
}

int main(int argc, char **argv) {
    const size_t points = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 160000000ull;
    const int threads = argc > 2 ? std::atoi(argv[2]) : 16;
    const size_t chunk = 65536, n_chunks = points / chunk;
    float *values = static_cast<float *>(std::aligned_alloc(4096, points * 4));
    float *stage = static_cast<float *>(std::aligned_alloc(4096, points * 4));
    int64_t *ts = static_cast<int64_t *>(std::aligned_alloc(4096, points * 8));
    {
        std::vector<std::thread> init; // (first touch spread the way the python process would NOT: one thread)
    }
    for (size_t i = 0; i < points; i++) { values[i] = (float)i; ts[i] = (int64_t)i * 1000; }
    std::memset(stage, 0, points * 4);
    struct Placement { const char *name; std::vector<int> cpus; };
    std::vector<Placement> placements;
    placements.push_back({"unpinned", std::vector<int>(threads, -1)});
    { std::vector<int> c; for (int w = 0; w < threads; w++) c.push_back(w); placements.push_back({"cpus 0..n-1", c}); }
    { std::vector<int> c; for (int w = 0; w < threads; w++) c.push_back((w % 8) * 8 + w / 8); placements.push_back({"one per L3 of node 0, then the second cores", c}); }
    { std::vector<int> c; for (int w = 0; w < threads; w++) c.push_back((w * 8) % 128 + (w * 8) / 128); placements.push_back({"one per L3 of both nodes", c}); }
    { std::vector<int> c; for (int w = 0; w < threads; w++) c.push_back(64 + (w % 8) * 8 + w / 8); placements.push_back({"one per L3 of node 1, then the second cores", c}); }
    for (int what = 0; what < 4; what++) {
        const char *names[4] = {"memcpy + scan", "nt copy + scan", "nt copy only", "scan only"};
        for (const Placement &placement : placements) {
            double best = 1e9;
            for (int rep = 0; rep < 4; rep++) {
                std::atomic<size_t> next{0};
                std::atomic<int> irregular{0};
                const auto t0 = std::chrono::steady_clock::now();
                std::vector<std::thread> pool;
                for (int w = 0; w < threads; w++)
                    pool.emplace_back([&, w] {
                        pin(placement.cpus[w]);
                        for (;;) {
                            const size_t c = next.fetch_add(1);
                            if (c >= n_chunks) break;
                            if (what == 0) std::memcpy(stage + c * chunk, values + c * chunk, chunk * 4);
                            if (what == 1 || what == 2) copy_nt(stage + c * chunk, values + c * chunk, chunk);
                            if (what != 2 && !regular(ts + c * chunk, chunk)) irregular++;
                        }
                    });
                for (auto &t : pool) t.join();
                const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                if (rep > 0 && ms < best) best = ms;
                if (irregular.load()) std::printf("?");
            }
            std::printf("%-16s %-48s %7.2f ms  %.3g points/s\n", names[what], placement.name, best, points / (best * 1e-3));
            std::fflush(stdout);
        }
    }
    return 0;
}
