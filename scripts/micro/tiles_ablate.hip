// Ablation microbenchmark of the grid tile kernel: synthetic Swing segments of fixed length, the
// real kernel's structure with pieces switched off by template flags, to see which piece keeps it
// below the pure-store ceiling. Timing tool only (the flags produce wrong values on purpose).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
struct SegDesc { int64_t start, delta; double slope, intercept; uint32_t n_total, n_model; float value; uint32_t flags; };
constexpr uint32_t TILE = 4096; constexpr int THREADS = 256; constexpr int LDS_SEGS = 1024;

template <bool SEARCH, bool DESC, bool MATH, bool CONTIG>
__global__ __launch_bounds__(THREADS) void k_tiles(const SegDesc *__restrict__ desc, const unsigned long long *__restrict__ offsets,
        const uint32_t *__restrict__ tile_first, uint64_t n_segments, uint64_t total, uint64_t n_tiles,
        int64_t *__restrict__ out_ts, float *__restrict__ out_val) {
    __shared__ uint32_t rel[LDS_SEGS + 1];
    const uint64_t tile = blockIdx.x, tile_start = tile * TILE;
    const uint64_t tile_end = min(total, tile_start + TILE);
    const uint32_t s0 = tile_first[tile];
    const uint32_t s1 = (tile + 1 < n_tiles) ? tile_first[tile + 1] : (uint32_t)(n_segments - 1);
    const uint32_t n_in_tile = s1 - s0 + 1;
    const uint64_t s0_offset = offsets[s0];
    for (uint32_t k = 1 + threadIdx.x; k < n_in_tile; k += THREADS) rel[k] = (uint32_t)(offsets[s0 + k] - tile_start);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll 1
    for (uint32_t j = 0; j < TILE / (THREADS * 4); j++) {
        const uint64_t p = tile_start + (uint64_t)j * (THREADS * 4) + (uint64_t)threadIdx.x * 4;
        if (p >= tile_end) break;
        const uint32_t local = (uint32_t)(p - tile_start);
        uint32_t lo = 0, hi = n_in_tile;
        if (SEARCH) while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (rel[mid] <= local) lo = mid; else hi = mid; }
        uint32_t segment = s0 + lo;
        uint64_t segment_offset = lo == 0 ? s0_offset : tile_start + rel[lo];
        SegDesc d;
        if (DESC) d = desc[segment]; else { d.start = 0; d.delta = 1000; d.slope = 0.5; d.intercept = 3.0; d.n_total = 1u << 30; d.n_model = 1u << 30; d.value = 1.f; d.flags = 5; }
        uint32_t index = (uint32_t)(p - segment_offset);
        if ((p + 4 <= tile_end) && (index + 4 <= d.n_total)) {
            int64_t t0 = d.start + (int64_t)((uint64_t)index * (uint64_t)d.delta);
            int64_t t1 = t0 + d.delta, t2 = t1 + d.delta, t3 = t2 + d.delta;
            float4 v;
            if (MATH) {
                v.x = (float)(d.slope * (double)t0 + d.intercept); v.y = (float)(d.slope * (double)t1 + d.intercept);
                v.z = (float)(d.slope * (double)t2 + d.intercept); v.w = (float)(d.slope * (double)t3 + d.intercept);
            } else { v = make_float4(d.value, d.value, d.value, d.value); }
            if (!CONTIG) {
                longlong2 *ts_out = reinterpret_cast<longlong2 *>(out_ts + p);
                ts_out[0] = make_longlong2(t0, t1); ts_out[1] = make_longlong2(t2, t3);
            } else {
                // exchange so each wave instruction stores 1 KiB contiguous: lane l needs chunk l
                // (points 2l, 2l+1 of the wave's 256) which lane l/2 owns.
                int src = lane >> 1; bool odd = lane & 1;
                long long a0 = __shfl(odd ? t2 : t0, src, 64) , a1 = 0; (void)a1;
                // emulate: 4 shuffles of 64-bit = 8 dword shuffles
                long long x0 = __shfl(t0, src, 64), x1 = __shfl(t1, src, 64), x2 = __shfl(t2, src, 64), x3 = __shfl(t3, src, 64);
                int src2 = 32 + (lane >> 1);
                long long y0 = __shfl(t0, src2, 64), y1 = __shfl(t1, src2, 64), y2 = __shfl(t2, src2, 64), y3 = __shfl(t3, src2, 64);
                longlong2 *wave_out = reinterpret_cast<longlong2 *>(out_ts + (p - (uint64_t)lane * 4));
                wave_out[lane] = odd ? make_longlong2(x2, x3) : make_longlong2(x0, x1);
                wave_out[64 + lane] = odd ? make_longlong2(y2, y3) : make_longlong2(y0, y1);
                (void)a0;
            }
            *reinterpret_cast<float4 *>(out_val + p) = v;
        } else {
            for (uint32_t k = 0; k < 4 && p + k < tile_end; k++) { out_ts[p + k] = (int64_t)p; out_val[p + k] = 1.f; }
        }
    }
    (void)wave;
}

int main(int argc, char **argv) {
    const uint64_t L = argc > 1 ? atoll(argv[1]) : 740;
    const uint64_t total = 2000000000ull / L * L, n_seg = total / L, n_tiles = (total + TILE - 1) / TILE;
    std::vector<SegDesc> desc(n_seg); std::vector<unsigned long long> off(n_seg + 1); std::vector<uint32_t> tf(n_tiles + 1);
    for (uint64_t i = 0; i < n_seg; i++) { desc[i] = {(int64_t)(i * L * 1000), 1000, 1e-6 * (i % 7), 100.0 + i % 13, (uint32_t)L, (uint32_t)L, 1.f, 5u}; off[i] = i * L; }
    off[n_seg] = total;
    for (uint64_t t = 0; t < n_tiles; t++) tf[t] = (uint32_t)((t * TILE) / L);
    SegDesc *d_desc; unsigned long long *d_off; uint32_t *d_tf; int64_t *ts; float *val;
    hipMalloc(&d_desc, n_seg * sizeof(SegDesc)); hipMalloc(&d_off, (n_seg + 1) * 8); hipMalloc(&d_tf, (n_tiles + 1) * 4);
    hipMalloc(&ts, total * 8); hipMalloc(&val, total * 4);
    hipMemcpy(d_desc, desc.data(), n_seg * sizeof(SegDesc), hipMemcpyHostToDevice);
    hipMemcpy(d_off, off.data(), (n_seg + 1) * 8, hipMemcpyHostToDevice); hipMemcpy(d_tf, tf.data(), (n_tiles + 1) * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char *name, auto kernel) {
        float best = 1e9f;
        for (int r = 0; r < 6; r++) { hipEventRecord(a); hipLaunchKernelGGL(kernel, dim3((uint32_t)n_tiles), dim3(THREADS), 0, 0, d_desc, d_off, d_tf, n_seg, total, n_tiles, ts, val); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (r && ms < best) best = ms; }
        printf("L=%llu %-28s %.3f ms  %.1f GB/s\n", (unsigned long long)L, name, best, 12.0 * total / best / 1e6);
    };
    run("full", k_tiles<true, true, true, false>);
    run("no search", k_tiles<false, true, true, false>);
    run("no desc load", k_tiles<true, false, true, false>);
    run("no f64 math", k_tiles<true, true, false, false>);
    run("no search/desc/math", k_tiles<false, false, false, false>);
    run("full + contiguous ts", k_tiles<true, true, true, true>);
    return 0;
}
