// Exclusive scan of uint32 arrays on the device (three small kernels), shared by the liftover and the column engine.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace hgx {

// ---- exclusive scan of uint32 (three small kernels; n up to 2^32-1) ----
static constexpr int SCAN_BLOCK = 1024; // elements per block (256 threads x 4)
static __global__ void __launch_bounds__(256) k_scan_block_sums(const uint32_t *__restrict__ in, uint32_t n, uint32_t *__restrict__ blockSums) {
    __shared__ uint32_t red[256];
    const uint32_t base = blockIdx.x * SCAN_BLOCK;
    uint32_t s = 0;
    for (int k = 0; k < 4; ++k) {
        const uint32_t i = base + threadIdx.x * 4 + k;
        if (i < n)
            s += in[i];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        blockSums[blockIdx.x] = red[0];
}
// exclusive scan of the per-block sums by one 1024-thread block (each thread owns a contiguous run)
static __global__ void __launch_bounds__(1024) k_scan_sums(uint32_t *blockSums, uint32_t nb, uint32_t *total) {
    __shared__ uint32_t part[1024];
    const uint32_t per = (nb + 1023) / 1024;
    const uint32_t lo = threadIdx.x * per, hi = min(nb, lo + per);
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; ++i)
        s += blockSums[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        uint32_t t = 0;
        if ((int)threadIdx.x >= o)
            t = part[threadIdx.x - o];
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t acc = part[threadIdx.x] - s;
    for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t v = blockSums[i];
        blockSums[i] = acc;
        acc += v;
    }
    if (threadIdx.x == 1023)
        *total = part[1023];
}
static __global__ void __launch_bounds__(256) k_scan_apply(const uint32_t *__restrict__ in, uint32_t n, const uint32_t *__restrict__ blockSums,
                                                    uint32_t *__restrict__ out) {
    __shared__ uint32_t part[256];
    const uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * 4;
    uint32_t v[4];
    uint32_t s = 0;
    for (int k = 0; k < 4; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        s += v[k];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    // Hillis-Steele inclusive scan over 256 partials
    for (int o = 1; o < 256; o <<= 1) {
        uint32_t t = 0;
        if ((int)threadIdx.x >= o)
            t = part[threadIdx.x - o];
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t acc = blockSums[blockIdx.x] + part[threadIdx.x] - s;
    for (int k = 0; k < 4; ++k) {
        if (base + k < n)
            out[base + k] = acc;
        acc += v[k];
    }
}

} // namespace hgx
