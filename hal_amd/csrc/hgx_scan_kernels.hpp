// Exclusive scan of uint32 arrays on the device (three small kernels), shared by the liftover and the column engine.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#ifndef HGX_SCAN_DEV
#define HGX_SCAN_DEV __device__
#endif

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


// ---- one pass: exclusive scan of (count, weight) pairs across the tiles of a launch (decoupled look-back) ----
// The three kernels above read their input twice and hand the total to the host; a compaction wants neither.  Here every workgroup
// takes a tile in the order the workgroups start (a ticket: the tiles before it are running or done, whatever the dispatcher's order),
// scans its own pairs, publishes the tile's sums and looks back over the tiles before it until it meets one whose running total is
// known (Merrill & Garland's decoupled look-back).  A tile's state is ONE 64-bit word — status (2 bits: 0 nothing yet, 1 the tile's
// own sums, 2 the sums up to and including the tile), count (24 bits), weight (38 bits) — so a reader sees a state and its sums
// together without a fence.  Launches of up to 2^24 - 1 counted items and 2^38 - 1 of weight (the callers check both).
// ctl: {ticket, unused} cleared before the launch; tiles: one word a tile, cleared before the launch.
static constexpr unsigned long long LB_COUNT_MAX = (1ull << 24) - 1, LB_WEIGHT_MAX = (1ull << 38) - 1;
HGX_SCAN_DEV __forceinline__ unsigned long long lb_pack(unsigned status, unsigned long long count, unsigned long long weight) {
    return ((unsigned long long)status << 62) | ((count < LB_COUNT_MAX ? count : LB_COUNT_MAX) << 38) | (weight < LB_WEIGHT_MAX ? weight : LB_WEIGHT_MAX);
}
HGX_SCAN_DEV __forceinline__ unsigned long long lb_load(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
HGX_SCAN_DEV __forceinline__ void lb_store(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the tile of this workgroup (every thread gets it); call once, first thing in the kernel
HGX_SCAN_DEV __forceinline__ unsigned lb_take_tile(unsigned int *ticket) {
    __shared__ unsigned sTile;
    __syncthreads(); // (the last call's readers are done with the word; and the host-side emulation, which runs a kernel again from
                     // its first barrier on, meets one before the ticket is taken)
    if (threadIdx.x == 0)
        sTile = atomicAdd(ticket, 1u);
    __syncthreads();
    return sTile;
}
// the same, and a word every thread of the workgroup must see the same value of (an error flag other workgroups may set while this
// one looks at it: threads that read it for themselves could part ways in front of a barrier)
HGX_SCAN_DEV __forceinline__ unsigned lb_take_tile(unsigned int *ticket, const unsigned int *word, unsigned &value) {
    __shared__ unsigned sTile2, sWord;
    __syncthreads();
    if (threadIdx.x == 0) {
        sTile2 = atomicAdd(ticket, 1u);
        sWord = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    value = sWord;
    return sTile2;
}
// A workgroup of 256 threads: thread t brings the sums (c, w) of its own items; it gets back the sums of everything in front of its
// items in the whole launch (exC, exW), and every thread the tile's own totals and the totals in front of the tile.
struct LbResult {
    unsigned long long exC, exW;     // in front of this thread's items
    unsigned long long tileC, tileW; // of this tile
    unsigned long long baseC, baseW; // in front of this tile
};
HGX_SCAN_DEV __forceinline__ LbResult lb_scan_tile(unsigned tile, unsigned long long c, unsigned long long w, unsigned long long *tiles) {
    __shared__ unsigned long long sWaveC[4], sWaveW[4], sBaseC, sBaseW;
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // inclusive scan inside the wavefront
    unsigned long long ic = c, iw = w;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long tc = __shfl_up(ic, o), tw = __shfl_up(iw, o);
        if ((int)lane >= o) {
            ic += tc;
            iw += tw;
        }
    }
    if (lane == 63) {
        sWaveC[wave] = ic;
        sWaveW[wave] = iw;
    }
    __syncthreads();
    unsigned long long wc = 0, ww = 0, tileC = 0, tileW = 0;
    for (unsigned k = 0; k < 4; ++k) {
        if (k < wave) {
            wc += sWaveC[k];
            ww += sWaveW[k];
        }
        tileC += sWaveC[k];
        tileW += sWaveW[k];
    }
    if (wave == 0) {
        // the first wavefront looks back: lane l reads the state of tile (base - l), a window of 64 tiles at a time
        if (lane == 0)
            lb_store(tiles + tile, lb_pack(tile == 0 ? 2u : 1u, tileC, tileW));
        unsigned long long baseC = 0, baseW = 0;
        long long base = (long long)tile - 1;
        while (base >= 0) {
            const long long idx = base - (long long)lane;
            unsigned long long word = 2ull << 62; // (in front of the first tile: a running total of nothing)
            if (idx >= 0) {
                do {
                    word = lb_load(tiles + idx);
                } while ((word >> 62) == 0);
            }
            const unsigned long long full = __ballot((word >> 62) == 2);
            const unsigned first = full ? (unsigned)__builtin_ctzll(full) : 64u; // the nearest tile whose running total is known
            unsigned long long pc = lane <= first ? (word >> 38) & LB_COUNT_MAX : 0, pw = lane <= first ? word & LB_WEIGHT_MAX : 0;
            for (int o = 32; o > 0; o >>= 1) {
                pc += __shfl_down(pc, o);
                pw += __shfl_down(pw, o);
            }
            pc = __shfl(pc, 0);
            pw = __shfl(pw, 0);
            baseC += pc;
            baseW += pw;
            if (full)
                break;
            base -= 64;
        }
        if (lane == 0) {
            if (tile != 0)
                lb_store(tiles + tile, lb_pack(2u, baseC + tileC, baseW + tileW));
            sBaseC = baseC;
            sBaseW = baseW;
        }
    }
    __syncthreads();
    LbResult r;
    r.baseC = sBaseC;
    r.baseW = sBaseW;
    r.tileC = tileC;
    r.tileW = tileW;
    r.exC = sBaseC + wc + ic - c;
    r.exW = sBaseW + ww + iw - w;
    __syncthreads(); // (the shared words are the next call's too)
    return r;
}

} // namespace hgx
