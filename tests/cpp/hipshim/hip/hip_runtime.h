// A host-side stand-in for the part of the HIP runtime the COLUMN ENGINE uses (hal_amd/csrc/hgx_columns.hip, hgx_device_image.hip
// and their kernel headers), so that the engine's own launch sequences, buffer sizes and kernels run on a machine without a GPU:
// test infrastructure only (tests/test_cpu_emulation.py builds a library of its own with it — libhgx.so is never built this way,
// and the liftover engine, whose kernels speak to the wavefront through DPP and ballots, is not part of it).
//   * device memory is host memory (hipMalloc = malloc, exact sizes: an address sanitizer sees a kernel's stray access);
//   * a launch runs the kernel function once per thread of the grid, block after block, one thread after the other; a kernel that
//     meets __syncthreads() or a shuffle is run again with the block's threads as fibers (ucontext), which hand over at every
//     barrier — __shared__ variables are statics (one block runs at a time);
//   * __shfl_down goes through a per-block array between two barriers; atomics are plain operations (one OS thread);
//   * launches of the fixed grid-stride grids (1024 / 2048 / 4096 blocks) are cut to 8 blocks: their kernels loop over the work.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <ucontext.h>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorNotInitialized = 3, hipErrorInvalidDevice = 101, hipErrorNoDevice = 100, hipErrorInsufficientDriver = 35 };
typedef void *hipStream_t;
struct hipshim_event {
    std::chrono::steady_clock::time_point t;
};
typedef hipshim_event *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipHostMallocPortable = 1 };
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace hipshim {
struct State {
    dim3 tIdx, bIdx, bDim, gDim;
    bool fibers = false;
    // the running block's fibers
    ucontext_t scheduler;
    std::vector<ucontext_t> ctx;
    std::vector<char> done;
    std::vector<std::unique_ptr<char[]>> stacks;
    unsigned current = 0;
    std::vector<unsigned long long> slots;
    const std::function<void()> *body = nullptr;
    unsigned long long launches = 0, fiberLaunches = 0;
    unsigned long long syncGeneration = 0;
    unsigned syncArrived = 0;
};
inline State &state() {
    static thread_local State s;
    return s;
}
struct NeedFibers {};
inline void barrier() {
    State &S = state();
    if (!S.fibers)
        throw NeedFibers();
    swapcontext(&S.ctx[S.current], &S.scheduler); // (the scheduler resumes every fiber of the block once per round: a barrier)
}
inline void fiberEntry() {
    State &S = state();
    (*S.body)();
    S.done[S.current] = 1;
    swapcontext(&S.ctx[S.current], &S.scheduler);
}
inline void runBlockWithFibers(unsigned threads) {
    State &S = state();
    const size_t STACK = 256 << 10;
    if (S.ctx.size() < threads) {
        S.ctx.resize(threads);
        S.done.resize(threads);
        while (S.stacks.size() < threads)
            S.stacks.emplace_back(new char[STACK]);
    }
    S.slots.assign(threads, 0);
    S.syncArrived = 0;
    for (unsigned t = 0; t < threads; ++t) {
        getcontext(&S.ctx[t]);
        S.ctx[t].uc_stack.ss_sp = S.stacks[t].get();
        S.ctx[t].uc_stack.ss_size = STACK;
        S.ctx[t].uc_link = nullptr;
        makecontext(&S.ctx[t], fiberEntry, 0);
        S.done[t] = 0;
    }
    for (;;) {
        unsigned alive = 0;
        for (unsigned t = 0; t < threads; ++t) {
            if (S.done[t])
                continue;
            ++alive;
            S.current = t;
            S.tIdx = dim3(t);
            swapcontext(&S.scheduler, &S.ctx[t]);
        }
        if (!alive)
            break;
    }
}
template <class F> void launch(dim3 grid, dim3 block, F &&f) {
    State &S = state();
    if ((grid.x == 1024 || grid.x == 2048 || grid.x == 4096) && (unsigned long long)grid.x * block.x >= 65536)
        grid.x = 8; // (the fixed grids of the grid-stride kernels)
    if (grid.y != 1 || grid.z != 1 || block.y != 1 || block.z != 1)
        throw std::runtime_error("hipshim: one-dimensional launches only");
    ++S.launches;
    const std::function<void()> body = f;
    S.body = &body;
    S.gDim = grid;
    S.bDim = block;
    bool fibers = false;
    for (;;) {
        S.fibers = fibers;
        try {
            for (unsigned b = 0; b < grid.x; ++b) {
                S.bIdx = dim3(b);
                if (fibers) {
                    runBlockWithFibers(block.x);
                } else {
                    for (unsigned t = 0; t < block.x; ++t) {
                        S.tIdx = dim3(t);
                        body();
                    }
                }
            }
            break;
        } catch (const NeedFibers &) { // (thrown by the launch's first thread at its first barrier: nothing was written that the rerun does not write again)
            if (fibers)
                throw std::runtime_error("hipshim: barrier outside a fiber");
            fibers = true;
            ++S.fiberLaunches;
        }
    }
    S.fibers = false;
    S.body = nullptr;
}
} // namespace hipshim

#define threadIdx (hipshim::state().tIdx)
#define blockIdx (hipshim::state().bIdx)
#define blockDim (hipshim::state().bDim)
#define gridDim (hipshim::state().gDim)
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) hipshim::launch((grid), (block), [&]() { kernel(__VA_ARGS__); })

// (a barrier of the whole block: a fiber stays in it until every fiber of the block that is still running has come to one — the
// wavefront operations below hand over too, a round each, and must not let the other wavefronts' threads through their barrier)
inline void __syncthreads() {
    hipshim::State &S = hipshim::state();
    if (!S.fibers)
        throw hipshim::NeedFibers();
    const unsigned long long gen = S.syncGeneration;
    ++S.syncArrived;
    for (;;) {
        unsigned alive = 0;
        for (unsigned t = 0; t < S.bDim.x; ++t)
            alive += S.done[t] ? 0u : 1u;
        if (S.syncGeneration != gen)
            break;
        if (S.syncArrived >= alive) {
            S.syncArrived = 0;
            ++S.syncGeneration;
            break;
        }
        hipshim::barrier();
    }
}
template <typename T> inline T __shfl_down(T v, int delta, int width = 64) {
    hipshim::State &S = hipshim::state();
    if (!S.fibers)
        throw hipshim::NeedFibers();
    static_assert(sizeof(T) <= 8, "hipshim: shuffles of up to eight bytes");
    unsigned long long raw = 0;
    memcpy(&raw, &v, sizeof(T));
    const unsigned me = S.current;
    S.slots[me] = raw;
    hipshim::barrier();
    const unsigned lane = me & 63u, from = me + (unsigned)delta;
    T out = v;
    if ((int)(lane % (unsigned)width) + delta < width && from < S.bDim.x && (from >> 6) == (me >> 6))
        memcpy(&out, &S.slots[from], sizeof(T));
    hipshim::barrier();
    return out;
}
// (the other wavefront operations the column engine uses, the same way: through the block's slots between two barriers; a
// wavefront is 64 consecutive threads of the block)
template <typename T, typename Pick> inline T hipshim_exchange(T v, Pick pick) {
    hipshim::State &S = hipshim::state();
    if (!S.fibers)
        throw hipshim::NeedFibers();
    static_assert(sizeof(T) <= 8, "hipshim: shuffles of up to eight bytes");
    unsigned long long raw = 0;
    memcpy(&raw, &v, sizeof(T));
    const unsigned me = S.current;
    S.slots[me] = raw;
    hipshim::barrier();
    const long long from = pick(me);
    T out = v;
    if (from >= 0 && (unsigned long long)from < S.bDim.x && ((unsigned)from >> 6) == (me >> 6))
        memcpy(&out, &S.slots[(size_t)from], sizeof(T));
    hipshim::barrier();
    return out;
}
template <typename T> inline T __shfl_up(T v, int delta, int width = 64) {
    return hipshim_exchange(v, [&](unsigned me) { return (int)(me & 63u) % width >= delta ? (long long)me - delta : -1ll; });
}
template <typename T> inline T __shfl(T v, int lane, int width = 64) {
    return hipshim_exchange(v, [&](unsigned me) { return (long long)((me & ~63u) + ((me & 63u) / (unsigned)width) * (unsigned)width + (unsigned)lane % (unsigned)width); });
}
inline unsigned long long __ballot(int predicate) {
    hipshim::State &S = hipshim::state();
    if (!S.fibers)
        throw hipshim::NeedFibers();
    const unsigned me = S.current;
    S.slots[me] = predicate ? 1ull : 0ull;
    hipshim::barrier();
    unsigned long long mask = 0;
    const unsigned w0 = me & ~63u;
    for (unsigned l = 0; l < 64 && w0 + l < S.bDim.x; ++l)
        if (S.slots[w0 + l])
            mask |= 1ull << l;
    hipshim::barrier();
    return mask;
}
#define __builtin_amdgcn_fence(order, scope) ((void)0)
inline void __builtin_amdgcn_wave_barrier() { // (a round of the block's fibers: the wavefront's other lanes have come as far)
    hipshim::barrier();
}
#define __HIP_MEMORY_SCOPE_AGENT 0
template <typename T> inline T __hip_atomic_load(const T *p, int, int) {
    return *p;
}
template <typename T> inline void __hip_atomic_store(T *p, T v, int, int) {
    *p = v;
}
inline int __popc(unsigned v) {
    return __builtin_popcount(v);
}
inline int __popcll(unsigned long long v) {
    return __builtin_popcountll(v);
}
template <typename T, typename U> inline T atomicAdd(T *p, U v) {
    const T old = *p;
    *p = (T)(old + (T)v);
    return old;
}
using std::max;
using std::min;

// ---- runtime ----
inline hipError_t hipGetDeviceCount(int *n) {
    const char *e = getenv("HGX_EMULATED_DEVICES"); // (the dry run of bench.py with several ranks: every ordinal is the host)
    *n = e ? std::max(1, atoi(e)) : 1;
    return hipSuccess;
}
inline hipError_t hipSetDevice(int) {
    return hipSuccess;
}
inline hipError_t hipGetDevice(int *d) {
    *d = 0;
    return hipSuccess;
}
inline hipError_t hipMalloc(void **p, size_t bytes) {
    *p = malloc(bytes ? bytes : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
template <typename T> inline hipError_t hipMalloc(T **p, size_t bytes) {
    return hipMalloc((void **)p, bytes);
}
inline hipError_t hipFree(void *p) {
    free(p);
    return hipSuccess;
}
inline hipError_t hipHostMalloc(void **p, size_t bytes, unsigned = 0) {
    return hipMalloc(p, bytes);
}
template <typename T> inline hipError_t hipHostMalloc(T **p, size_t bytes, unsigned = 0) {
    return hipMalloc((void **)p, bytes);
}
inline hipError_t hipHostFree(void *p) {
    free(p);
    return hipSuccess;
}
inline hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind) {
    if (bytes)
        memmove(dst, src, bytes);
    return hipSuccess;
}
inline hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind k, hipStream_t) {
    return hipMemcpy(dst, src, bytes, k);
}
inline hipError_t hipMemset(void *p, int v, size_t bytes) {
    if (bytes)
        memset(p, v, bytes);
    return hipSuccess;
}
inline hipError_t hipMemsetAsync(void *p, int v, size_t bytes, hipStream_t) {
    return hipMemset(p, v, bytes);
}
inline hipError_t hipMemGetInfo(size_t *freeB, size_t *totalB) {
    *freeB = (size_t)48 << 30;
    *totalB = (size_t)64 << 30;
    return hipSuccess;
}
enum { hipStreamDefault = 0, hipStreamNonBlocking = 1 };
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { // (every stream is the one thread of the emulation)
    *s = nullptr;
    return hipSuccess;
}
inline hipError_t hipStreamCreate(hipStream_t *s) {
    *s = nullptr;
    return hipSuccess;
}
inline hipError_t hipStreamDestroy(hipStream_t) {
    return hipSuccess;
}
inline hipError_t hipStreamSynchronize(hipStream_t) {
    return hipSuccess;
}
inline hipError_t hipDeviceSynchronize() {
    return hipSuccess;
}
inline hipError_t hipGetLastError() {
    return hipSuccess;
}
inline const char *hipGetErrorString(hipError_t e) {
    return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "out of memory (host-side emulation)" : "error (host-side emulation)";
}
inline hipError_t hipEventCreate(hipEvent_t *e) {
    *e = new hipshim_event;
    return hipSuccess;
}
inline hipError_t hipEventDestroy(hipEvent_t e) {
    delete e;
    return hipSuccess;
}
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
    e->t = std::chrono::steady_clock::now();
    return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t) {
    return hipSuccess;
}
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
