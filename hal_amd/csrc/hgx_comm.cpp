// The exchange step of the multi-GPU path from inside the library: RCCL (librccl.so, loaded at run time) called directly, one
// process per GPU, ONE collective per batch.  Every rank writes the records of its plan's last run as a self-describing wire
// blob (hgx_liftover_wire_blob) into its own slot of a buffer of n_ranks equal slots and a single in-place ncclAllGather fills
// in the others: no exchange of sizes first — a blob carries its size in its header — and no host wait (the collective is
// ordered on the caller's stream).  The reference has no counterpart (its parallelism is a pool of processes writing files:
// maf/hal2mafMP.py:176-190); the semantics pinned by it are those of Liftover::visitLine's per-line independence
// (liftover/impl/halLiftover.cpp:46-92): rank-major concatenation of the shards' records is the unsharded output.
#include "../../include/hgx.h"
#include "hgx_liftover_engine.hpp"
#include <cstring>
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <mutex>
#include <stdexcept>
#include <algorithm>
#include <string>

namespace {

struct UniqueId { // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
    char internal[128];
};
typedef int (*GetUniqueIdFn)(UniqueId *);
typedef int (*CommInitRankFn)(void **, int, UniqueId, int);
typedef int (*AllGatherFn)(const void *, void *, size_t, int, void *, hipStream_t);
typedef int (*CommDestroyFn)(void *);
typedef int (*SendFn)(const void *, size_t, int, int, void *, hipStream_t);
typedef int (*RecvFn)(void *, size_t, int, int, void *, hipStream_t);
typedef int (*GroupFn)(void);
typedef const char *(*GetErrorStringFn)(int);

struct Rccl {
    void *lib = nullptr;
    GetUniqueIdFn getUniqueId = nullptr;
    CommInitRankFn commInitRank = nullptr;
    AllGatherFn allGather = nullptr;
    CommDestroyFn commDestroy = nullptr;
    GetErrorStringFn getErrorString = nullptr;
    SendFn send = nullptr;
    RecvFn recv = nullptr;
    GroupFn groupStart = nullptr, groupEnd = nullptr;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    static std::string failure;
    std::call_once(once, [&]() {
        // (a process that has loaded a librccl already — PyTorch's — gets that one back under the same soname)
        const char *names[] = {getenv("HGX_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            if (!n || !*n)
                continue;
            r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.lib)
                break;
        }
        if (!r.lib) {
            failure = "the RCCL library (librccl.so) could not be loaded; set HGX_RCCL_LIB to its path";
            return;
        }
        r.getUniqueId = (GetUniqueIdFn)dlsym(r.lib, "ncclGetUniqueId");
        r.commInitRank = (CommInitRankFn)dlsym(r.lib, "ncclCommInitRank");
        r.allGather = (AllGatherFn)dlsym(r.lib, "ncclAllGather");
        r.commDestroy = (CommDestroyFn)dlsym(r.lib, "ncclCommDestroy");
        r.getErrorString = (GetErrorStringFn)dlsym(r.lib, "ncclGetErrorString");
        r.send = (SendFn)dlsym(r.lib, "ncclSend"); // (hgx_liftover_gather; checked there)
        r.recv = (RecvFn)dlsym(r.lib, "ncclRecv");
        r.groupStart = (GroupFn)dlsym(r.lib, "ncclGroupStart");
        r.groupEnd = (GroupFn)dlsym(r.lib, "ncclGroupEnd");
        if (!r.getUniqueId || !r.commInitRank || !r.allGather || !r.commDestroy)
            failure = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy";
    });
    if (!failure.empty())
        throw std::runtime_error(failure);
    return r;
}

void check(int rc, const char *what) {
    if (rc != 0) {
        Rccl &r = rccl();
        throw std::runtime_error(std::string(what) + ": " + (r.getErrorString ? r.getErrorString(rc) : "RCCL error") + " (" + std::to_string(rc) + ")");
    }
}

void setErr(char **err, const std::string &m) {
    if (err) {
        *err = (char *)malloc(m.size() + 1);
        if (*err)
            memcpy(*err, m.c_str(), m.size() + 1);
    }
}

} // namespace

struct hgx_comm {
    void *comm = nullptr;
    int rank = 0, nRanks = 1, device = 0;
};

extern "C" {

int hgx_comm_unique_id(unsigned char *id128, char **err) {
    try {
        if (!id128)
            throw std::runtime_error("hgx_comm_unique_id: null argument");
        UniqueId id;
        memset(&id, 0, sizeof id);
        check(rccl().getUniqueId(&id), "ncclGetUniqueId");
        memcpy(id128, id.internal, 128);
        return HGX_OK;
    } catch (std::exception &e) {
        setErr(err, e.what());
        return HGX_ERR;
    }
}

int hgx_comm_create(const unsigned char *id128, int rank, int n_ranks, int device, hgx_comm **out, char **err) {
    try {
        if (!id128 || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks)
            throw std::runtime_error("hgx_comm_create: bad argument");
        if (hipSetDevice(device) != hipSuccess)
            throw std::runtime_error("hgx_comm_create: invalid device ordinal " + std::to_string(device));
        UniqueId id;
        memcpy(id.internal, id128, 128);
        hgx_comm *c = new hgx_comm;
        c->rank = rank;
        c->nRanks = n_ranks;
        c->device = device;
        const int rc = rccl().commInitRank(&c->comm, n_ranks, id, rank);
        if (rc != 0) {
            delete c;
            check(rc, "ncclCommInitRank");
        }
        *out = c;
        return HGX_OK;
    } catch (std::exception &e) {
        setErr(err, e.what());
        return HGX_ERR;
    }
}

void hgx_comm_destroy(hgx_comm *c) {
    if (!c)
        return;
    if (c->comm) {
        (void)hipSetDevice(c->device);
        (void)rccl().commDestroy(c->comm);
    }
    delete c;
}

// this rank's blob of the plan's last run into `mine` (a slot of slot_bytes); on failure the slot's header says so and `failure`
// why — the caller still takes part in its collective
static void buildSlot(hgx_liftover_plan *p, unsigned char *mine, size_t slot_bytes, int64_t first_query, int bed_only, void *hip_stream,
                      size_t &wrote, std::string &failure) {
    size_t need = 0;
    wrote = 0;
    // (the buffers rotate: the slot may hold a complete blob of an earlier batch.  Its header is cleared first, so that whatever
    // goes wrong below the other ranks cannot take stale records for this batch's: a slot without the magic is not a blob)
    (void)hipMemsetAsync(mine, 0, 32, (hipStream_t)hip_stream);
    bool tooSmall = false;
    try {
        need = hgx::liftoverPlanWireBlob(p, nullptr, 0, first_query, nullptr, hip_stream);
        if (need <= slot_bytes) {
            int fmt = bed_only ? 8 : 0;
            wrote = hgx::liftoverPlanWireBlob(p, mine, slot_bytes, first_query, &fmt, hip_stream);
        } else {
            tooSmall = true;
            failure = "this rank's records need " + std::to_string(need) + " bytes, the slot has " + std::to_string(slot_bytes);
        }
    } catch (std::exception &e) { // a plan with a batch in flight, a HIP error while the blob was made, no memory for its staging
        failure = std::string("no blob from this rank: ") + e.what();
    }
    if (!failure.empty()) {
        // the slot's header says so to the other ranks: format 0; n_queries 0 = the slot was too small, and the bytes the blob
        // would have needed stand in the record count; n_queries 1 = no blob for another reason (a batch in flight, a HIP error)
        struct {
            char magic[4];
            uint32_t format;
            int64_t firstQuery;
            uint64_t nq, nrec;
        } h = {{'H', 'G', 'X', 'W'}, 0u, first_query, tooSmall ? 0ull : 1ull, (uint64_t)need};
        // (best effort: when even this copy fails the slot keeps its cleared header — the collective is still posted)
        if (hipMemcpyAsync(mine, &h, sizeof h, hipMemcpyHostToDevice, (hipStream_t)hip_stream) == hipSuccess)
            (void)hipStreamSynchronize((hipStream_t)hip_stream); // (h lives on this stack frame)
    }
}

int hgx_liftover_exchange(hgx_liftover_plan *p, hgx_comm *c, int64_t first_query, void *d_gathered, size_t slot_bytes, void *hip_stream,
                          size_t *my_bytes, char **err) {
    try {
        if (!p || !c || !d_gathered)
            throw std::runtime_error("hgx_liftover_exchange: null argument");
        if (slot_bytes < 64 || slot_bytes % 8)
            throw std::runtime_error("hgx_liftover_exchange: the slot size must be a multiple of 8 and at least 64 bytes");
        // (argument errors above are the caller's on every rank alike; from here on this rank takes part in the collective
        // whatever happens to its own blob — the other ranks are waiting in theirs)
        unsigned char *mine = (unsigned char *)d_gathered + (size_t)c->rank * slot_bytes;
        size_t wrote = 0;
        std::string failure;
        buildSlot(p, mine, slot_bytes, first_query, 0, hip_stream, wrote, failure);
        check(rccl().allGather(mine, d_gathered, slot_bytes, /*ncclUint8*/ 1, c->comm, (hipStream_t)hip_stream), "ncclAllGather");
        if (my_bytes)
            *my_bytes = wrote;
        if (!failure.empty())
            throw std::runtime_error("hgx_liftover_exchange: " + failure + " (the exchange was carried out; the slot's header says so to the other ranks)");
        return HGX_OK;
    } catch (std::exception &e) {
        setErr(err, e.what());
        return HGX_ERR;
    }
}

// every rank of [first, last) sends its slot to `root` (one of them), whose buffer holds them at their distance from `first`
static int gatherSlots(const char *who, hgx_liftover_plan *p, hgx_comm *c, int root, int first, int last, int64_t first_query, void *d_gathered,
                       size_t slot_bytes, int bed_only, void *hip_stream, size_t *my_bytes, char **err) {
    try {
        if (!p || !c || !d_gathered)
            throw std::runtime_error(std::string(who) + ": null argument");
        if (slot_bytes < 64 || slot_bytes % 8)
            throw std::runtime_error(std::string(who) + ": the slot size must be a multiple of 8 and at least 64 bytes");
        if (root < first || root >= last || first < 0 || last > c->nRanks)
            throw std::runtime_error(std::string(who) + ": no such root rank");
        Rccl &R = rccl();
        if (last - first > 1 && (!R.send || !R.recv || !R.groupStart || !R.groupEnd))
            throw std::runtime_error("librccl.so lacks ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd");
        const bool isRoot = c->rank == root;
        unsigned char *mine = (unsigned char *)d_gathered + (isRoot ? (size_t)(c->rank - first) * slot_bytes : 0);
        size_t wrote = 0;
        std::string failure;
        buildSlot(p, mine, slot_bytes, first_query, bed_only, hip_stream, wrote, failure);
        if (last - first > 1) {
            check(R.groupStart(), "ncclGroupStart");
            int rc = 0;
            if (isRoot) {
                for (int r = first; r < last && rc == 0; ++r)
                    if (r != root)
                        rc = R.recv((unsigned char *)d_gathered + (size_t)(r - first) * slot_bytes, slot_bytes, /*ncclUint8*/ 1, r, c->comm,
                                    (hipStream_t)hip_stream);
            } else {
                rc = R.send(mine, slot_bytes, /*ncclUint8*/ 1, root, c->comm, (hipStream_t)hip_stream);
            }
            const int rcEnd = R.groupEnd();
            check(rc, isRoot ? "ncclRecv" : "ncclSend");
            check(rcEnd, "ncclGroupEnd");
        }
        if (my_bytes)
            *my_bytes = wrote;
        if (!failure.empty())
            throw std::runtime_error(std::string(who) + ": " + failure + " (the transfer was carried out; the slot's header says so to the root)");
        return HGX_OK;
    } catch (std::exception &e) {
        setErr(err, e.what());
        return HGX_ERR;
    }
}

int hgx_liftover_gather(hgx_liftover_plan *p, hgx_comm *c, int root, int64_t first_query, void *d_gathered, size_t slot_bytes, int bed_only,
                        void *hip_stream, size_t *my_bytes, char **err) {
    return gatherSlots("hgx_liftover_gather", p, c, root, 0, c ? c->nRanks : 0, first_query, d_gathered, slot_bytes, bed_only, hip_stream, my_bytes,
                       err);
}

int hgx_liftover_gather_writers(hgx_liftover_plan *p, hgx_comm *c, int group_size, int64_t first_query, void *d_gathered, size_t slot_bytes,
                                int bed_only, void *hip_stream, size_t *my_bytes, char **err) {
    if (!c || group_size < 1) {
        setErr(err, "hgx_liftover_gather_writers: null communicator or a group of less than one rank");
        return HGX_ERR;
    }
    const int writer = c->rank - c->rank % group_size;
    return gatherSlots("hgx_liftover_gather_writers", p, c, writer, writer, std::min(writer + group_size, c->nRanks), first_query, d_gathered,
                       slot_bytes, bed_only, hip_stream, my_bytes, err);
}

int hgx_comm_all_sizes(hgx_comm *c, uint64_t mine, uint64_t *sizes, char **err) {
    try {
        if (!c || !sizes)
            throw std::runtime_error("hgx_comm_all_sizes: null argument");
        if (hipSetDevice(c->device) != hipSuccess)
            throw std::runtime_error("hgx_comm_all_sizes: hipSetDevice failed");
        void *d = nullptr;
        if (hipMalloc(&d, (size_t)c->nRanks * 8) != hipSuccess)
            throw std::runtime_error("hgx_comm_all_sizes: hipMalloc failed");
        std::string failure;
        try {
            if (hipMemcpy((unsigned char *)d + (size_t)c->rank * 8, &mine, 8, hipMemcpyHostToDevice) != hipSuccess)
                failure = "hipMemcpy failed"; // (the collective is still posted: the other ranks wait in theirs)
            check(rccl().allGather((unsigned char *)d + (size_t)c->rank * 8, d, 8, /*ncclUint8*/ 1, c->comm, nullptr), "ncclAllGather");
            if (hipStreamSynchronize(nullptr) != hipSuccess || hipMemcpy(sizes, d, (size_t)c->nRanks * 8, hipMemcpyDeviceToHost) != hipSuccess)
                failure = "hipMemcpy failed";
        } catch (...) {
            (void)hipFree(d);
            throw;
        }
        (void)hipFree(d);
        if (!failure.empty())
            throw std::runtime_error("hgx_comm_all_sizes: " + failure);
        return HGX_OK;
    } catch (std::exception &e) {
        setErr(err, e.what());
        return HGX_ERR;
    }
}

} // extern "C"
