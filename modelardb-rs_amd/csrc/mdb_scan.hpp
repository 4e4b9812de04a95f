// mdb_scan.hpp - device-wide exclusive prefix sum of f(i), i in [0, n), into 64-bit offsets:
// reduce per 1024-item block, scan the block sums with one workgroup, then a block-wide wavefront
// scan adds the block base. out[n] receives the total. F is a device functor uint64_t(uint64_t).
#pragma once

#include "mdb_common.hpp"

namespace mdb {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 4;
constexpr int SCAN_BLOCK_ITEMS = SCAN_THREADS * SCAN_ITEMS;

template <typename F>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_reduce(F f, uint64_t n,
                                                             unsigned long long *__restrict__ block_sums) {
    __shared__ uint64_t lds[17];
    const uint64_t first = (uint64_t)blockIdx.x * SCAN_BLOCK_ITEMS + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t local = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++)
        if (first + k < n) local += f(first + k);
    uint64_t total;
    (void)block_exclusive_scan_u64(local, lds, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void k_scan_block_sums(unsigned long long *__restrict__ block_sums,
                                                          uint32_t n_blocks,
                                                          unsigned long long *__restrict__ total_out);

template <typename F>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_write(F f, uint64_t n,
                                                            const unsigned long long *__restrict__ block_sums,
                                                            unsigned long long *__restrict__ out) {
    __shared__ uint64_t lds[17];
    const uint64_t first = (uint64_t)blockIdx.x * SCAN_BLOCK_ITEMS + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t item[SCAN_ITEMS];
    uint64_t local = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        item[k] = (first + k < n) ? f(first + k) : 0;
        local += item[k];
    }
    uint64_t total;
    uint64_t offset = block_sums[blockIdx.x] + block_exclusive_scan_u64(local, lds, &total);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (first + k < n) {
            out[first + k] = offset;
            offset += item[k];
            if (first + k == n - 1) out[n] = offset;
        }
    }
}

// Enqueues the three kernels on the context's stream. `block_sums` needs n/1024 + 2 entries; the
// total lands in block_sums[n_blocks] as well as out[n]. n == 0 writes out[0] = 0.
template <typename F>
int device_exclusive_scan(mdb_ctx *ctx, F f, uint64_t n, unsigned long long *out,
                          unsigned long long *block_sums, const char *name) {
    if (n == 0) {
        MDB_HIP_CHECK(hipMemsetAsync(out, 0, 8, ctx->stream));
        return 0;
    }
    const uint64_t n_blocks = (n + SCAN_BLOCK_ITEMS - 1) / SCAN_BLOCK_ITEMS;
    if (n_blocks > 0x7fffffffull) return fail("Too many items for one scan.");
    LaunchTimer timer(ctx, name);
    hipLaunchKernelGGL(k_scan_reduce<F>, dim3((uint32_t)n_blocks), dim3(SCAN_THREADS), 0, ctx->stream,
                       f, n, block_sums);
    hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(1024), 0, ctx->stream, block_sums,
                       (uint32_t)n_blocks, block_sums + n_blocks);
    hipLaunchKernelGGL(k_scan_write<F>, dim3((uint32_t)n_blocks), dim3(SCAN_THREADS), 0, ctx->stream, f,
                       n, block_sums, out);
    return 0;
}

inline uint64_t scan_block_sums_bytes(uint64_t n) {
    return ((n + SCAN_BLOCK_ITEMS - 1) / SCAN_BLOCK_ITEMS + 2) * 8;
}

} // namespace mdb
