// shard.cpp -- the multi-GPU form of the hot path behind the C ABI: ONE process drives ndev devices; the batch is split into
// contiguous shards (problems are independent, model + task table replicated), device r solves its shard into a packed slot
//     [ q rows: nq x b float64 | iterations: b int32 | success: b uint8 ]
// and ONE RCCL all-gather (ncclAllGather inside ncclGroupStart/End, one communicator + one stream per device) leaves every device
// with every slot: [ndev][slot_bytes].  Same shard rule and slot layout as the one-process-per-GPU Python path
// (ik_amd/distributed.py: shard_range, _layout), which calls these entry points for both.
//
// The reference's caller is a single C++ process (ik_ros/src/cassie.cpp:95-112 calls ik::dls once per tick); this is what a C++
// caller with a batch of targets and several GPUs links against.  librccl is opened with dlopen (libikgpu.so stays loadable
// without it); a group of ONE device without RCCL falls back to a device-to-device copy of its own slot.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "ikgpu.h"

int ikgpu_set_last_error(int code, const std::string &msg);   // capi.cpp

namespace {

// the subset of rccl.h this file uses (ABI of RCCL 2.x / NCCL 2.x)
typedef struct ncclComm *ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1 } ncclDataType_t;

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

const Rccl &rccl() {
    static const Rccl r = [] {
        Rccl a;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"}) {
            a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (a.lib) break;
        }
        if (!a.lib) return a;
        auto sym = [&](const char *n) { return dlsym(a.lib, n); };
        a.CommInitAll = reinterpret_cast<decltype(a.CommInitAll)>(sym("ncclCommInitAll"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
        a.AllGather = reinterpret_cast<decltype(a.AllGather)>(sym("ncclAllGather"));
        a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
        a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
        a.ok = a.CommInitAll && a.CommDestroy && a.AllGather && a.GroupStart && a.GroupEnd;
        return a;
    }();
    return r;
}

struct Rank {
    int device = 0;
    ikgpu_problem *problem = nullptr;
    hipStream_t stream = nullptr;
    ncclComm_t comm = nullptr;
    void *slot = nullptr;      // send buffer: this rank's packed slot
    size_t slot_cap = 0;
};

}  // namespace

struct ikgpu_shard_group {
    std::vector<Rank> ranks;
    int nq = 0, ntasks = 0;
    bool use_rccl = false;
};

namespace {

int fail(int code, const std::string &msg) { return ikgpu_set_last_error(code, msg); }

struct DeviceScope {
    int prev = -1;
    explicit DeviceScope(int dev) {
        (void)hipGetDevice(&prev);
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DeviceScope() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace

extern "C" {

void ikgpu_shard_range(int64_t total, int32_t rank, int32_t nranks, int64_t *lo, int64_t *hi) {
    const int64_t base = total / nranks, rem = total % nranks;
    const int64_t l = rank * base + (rank < rem ? rank : rem);
    if (lo) *lo = l;
    if (hi) *hi = l + base + (rank < rem ? 1 : 0);
}

size_t ikgpu_shard_slot_layout(int32_t rows, int64_t b, size_t *off_q, size_t *off_iters, size_t *off_success) {
    const size_t q_bytes = static_cast<size_t>(rows) * static_cast<size_t>(b) * 8, it_bytes = static_cast<size_t>(b) * 4;
    if (off_q) *off_q = 0;
    if (off_iters) *off_iters = q_bytes;
    if (off_success) *off_success = q_bytes + it_bytes;
    return q_bytes + it_bytes + static_cast<size_t>(b);
}

size_t ikgpu_shard_slot_bytes(int32_t rows, int64_t total, int32_t nranks) {
    int64_t lo = 0, hi = 0;
    ikgpu_shard_range(total, 0, nranks, &lo, &hi);   // rank 0 always holds a largest shard
    return (ikgpu_shard_slot_layout(rows, hi - lo, nullptr, nullptr, nullptr) + 15) / 16 * 16;
}

int ikgpu_shard_group_create(const ikgpu_model *m, const ikgpu_task *tasks, int32_t ntasks, const ikgpu_task *constraints,
                             int32_t nconstraints, const int32_t *devices, int32_t ndev, ikgpu_shard_group **out) {
    if (!m || !tasks || !devices || !out) return fail(IKGPU_ERR_INVALID, "null argument");
    if (ndev < 1 || ndev > 64) return fail(IKGPU_ERR_INVALID, "ndev must be in 1..64");
    *out = nullptr;
    for (int i = 0; i < ndev; ++i)
        for (int j = 0; j < i; ++j)
            if (devices[i] == devices[j]) return fail(IKGPU_ERR_INVALID, "a device appears twice in the group");
    std::unique_ptr<ikgpu_shard_group> g(new ikgpu_shard_group);
    g->ranks.resize(static_cast<size_t>(ndev));
    g->ntasks = ntasks;
    int rc = IKGPU_OK;
    for (int r = 0; r < ndev && rc == IKGPU_OK; ++r) {
        Rank &k = g->ranks[static_cast<size_t>(r)];
        k.device = devices[r];
        rc = ikgpu_problem_create_constrained(m, tasks, ntasks, constraints, nconstraints, devices[r], &k.problem);
        if (rc != IKGPU_OK) break;
        DeviceScope scope(k.device);
        if (hipStreamCreateWithFlags(&k.stream, hipStreamNonBlocking) != hipSuccess) rc = fail(IKGPU_ERR_DEVICE, "hipStreamCreate failed");
    }
    if (rc == IKGPU_OK) {
        ikgpu_flat_model flat;
        (void)ikgpu_model_get_flat(m, &flat);
        g->nq = flat.nq;
        const Rccl &R = rccl();
        if (R.ok) {
            std::vector<ncclComm_t> comms(static_cast<size_t>(ndev));
            const ncclResult_t nr = R.CommInitAll(comms.data(), ndev, devices);
            if (nr != ncclSuccess) rc = fail(IKGPU_ERR_DEVICE, std::string("ncclCommInitAll: ") + (R.GetErrorString ? R.GetErrorString(nr) : "error"));
            else {
                for (int r = 0; r < ndev; ++r) g->ranks[static_cast<size_t>(r)].comm = comms[static_cast<size_t>(r)];
                g->use_rccl = true;
            }
        } else if (ndev > 1) {
            rc = fail(IKGPU_ERR_UNSUPPORTED, "librccl could not be loaded: a group of more than one device needs it");
        }
    }
    if (rc != IKGPU_OK) {
        const std::string keep = ikgpu_last_error();
        ikgpu_shard_group_destroy(g.release());
        return fail(rc, keep);
    }
    *out = g.release();
    return IKGPU_OK;
}

void ikgpu_shard_group_destroy(ikgpu_shard_group *g) {
    if (!g) return;
    for (Rank &k : g->ranks) {
        DeviceScope scope(k.device);
        if (k.stream) (void)hipStreamSynchronize(k.stream);
        if (k.comm && rccl().ok) (void)rccl().CommDestroy(k.comm);
        if (k.slot) (void)hipFree(k.slot);
        if (k.stream) (void)hipStreamDestroy(k.stream);
        if (k.problem) ikgpu_problem_destroy(k.problem);
    }
    delete g;
}

int32_t ikgpu_shard_group_size(const ikgpu_shard_group *g) { return g ? static_cast<int32_t>(g->ranks.size()) : 0; }

const ikgpu_problem *ikgpu_shard_group_problem(const ikgpu_shard_group *g, int32_t rank) {
    return g && rank >= 0 && rank < static_cast<int32_t>(g->ranks.size()) ? g->ranks[static_cast<size_t>(rank)].problem : nullptr;
}

int32_t ikgpu_shard_group_uses_rccl(const ikgpu_shard_group *g) { return g && g->use_rccl ? 1 : 0; }

int ikgpu_dls_solve_batch_sharded(ikgpu_shard_group *g, int64_t total, const double *const *q0, const double *const *targets,
                                  const ikgpu_dls_params *params, void *const *gathered) {
    if (!g || !q0 || !targets || !params || !gathered) return fail(IKGPU_ERR_INVALID, "null argument");
    const int ndev = static_cast<int>(g->ranks.size());
    if (total < ndev) return fail(IKGPU_ERR_INVALID, "fewer problems than devices in the group");
    const size_t slot = ikgpu_shard_slot_bytes(g->nq, total, ndev);
    // 1. every device solves its shard straight into typed views of its send slot
    for (int r = 0; r < ndev; ++r) {
        Rank &k = g->ranks[static_cast<size_t>(r)];
        if (!q0[r] || !targets[r] || !gathered[r]) return fail(IKGPU_ERR_INVALID, "null per-device pointer");
        DeviceScope scope(k.device);
        if (k.slot_cap < slot) {
            if (k.slot) { (void)hipStreamSynchronize(k.stream); (void)hipFree(k.slot); k.slot = nullptr; k.slot_cap = 0; }
            if (hipMalloc(&k.slot, slot) != hipSuccess) return fail(IKGPU_ERR_DEVICE, "hipMalloc of the send slot failed");
            k.slot_cap = slot;
        }
        int64_t lo = 0, hi = 0;
        ikgpu_shard_range(total, r, ndev, &lo, &hi);
        size_t oq = 0, oi = 0, os = 0;
        (void)ikgpu_shard_slot_layout(g->nq, hi - lo, &oq, &oi, &os);
        char *base = static_cast<char *>(k.slot);
        const int rc = ikgpu_dls_solve_batch(k.problem, hi - lo, q0[r], targets[r], params, reinterpret_cast<double *>(base + oq),
                                             reinterpret_cast<uint8_t *>(base + os), reinterpret_cast<int32_t *>(base + oi), IKGPU_SOA, k.stream);
        if (rc != IKGPU_OK) return rc;
    }
    // 2. one all-gather: every device ends with [ndev][slot]
    if (g->use_rccl) {
        const Rccl &R = rccl();
        ncclResult_t nr = R.GroupStart();
        for (int r = 0; r < ndev && nr == ncclSuccess; ++r) {
            Rank &k = g->ranks[static_cast<size_t>(r)];
            nr = R.AllGather(k.slot, gathered[r], slot, ncclUint8, k.comm, k.stream);
        }
        const ncclResult_t ne = R.GroupEnd();
        if (nr == ncclSuccess) nr = ne;
        if (nr != ncclSuccess) return fail(IKGPU_ERR_DEVICE, std::string("ncclAllGather: ") + (R.GetErrorString ? R.GetErrorString(nr) : "error"));
    } else {   // one device, no RCCL: its own slot is the whole result
        Rank &k = g->ranks[0];
        DeviceScope scope(k.device);
        if (hipMemcpyAsync(gathered[0], k.slot, slot, hipMemcpyDeviceToDevice, k.stream) != hipSuccess)
            return fail(IKGPU_ERR_DEVICE, "copy of the slot failed");
    }
    return IKGPU_OK;
}

int ikgpu_shard_group_synchronize(ikgpu_shard_group *g) {
    if (!g) return fail(IKGPU_ERR_INVALID, "null group");
    for (Rank &k : g->ranks) {
        DeviceScope scope(k.device);
        const hipError_t e = hipStreamSynchronize(k.stream);
        if (e != hipSuccess) return fail(IKGPU_ERR_DEVICE, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    }
    return IKGPU_OK;
}

void *ikgpu_shard_group_stream(const ikgpu_shard_group *g, int32_t rank) {
    return g && rank >= 0 && rank < static_cast<int32_t>(g->ranks.size()) ? static_cast<void *>(g->ranks[static_cast<size_t>(rank)].stream) : nullptr;
}

}  // extern "C"
