// shard.cpp -- the multi-GPU form of the hot path behind the C ABI: ONE process drives ndev devices; the batch is split into
// contiguous shards (problems are independent, model + task table replicated), device r solves its shard into a packed slot
//     [ q rows: nq x b float64 | iterations: b int32 | success: b uint8 ]
// and ONE RCCL all-gather (ncclAllGather inside ncclGroupStart/End, one communicator + one stream per device) leaves every device
// with every slot: [ndev][slot_bytes].  Same shard rule and slot layout as the one-process-per-GPU Python path
// (ik_amd/distributed.py: shard_range, _layout), which calls these entry points for both.
//
// Issue is PARALLEL: every rank has a worker thread bound to its device (created with the group); a solve hands each worker its
// shard's launch and waits until all have enqueued -- at 8 GPUs and a 0.07-0.14 ms kernel, eight launches issued one after the other
// from one thread cost as much as the solve itself.  The collective is then issued by the calling thread in one group call.
// IKGPU_SHARD_LOOPBACK=1 (tests, rehearsals on a one-GPU box): ranks may share a device and the all-gather is done with
// device-to-device copies ordered by events -- the whole group code path (threads, shard rule, slot layout, decode) without RCCL.
//
// The reference's caller is a single C++ process (ik_ros/src/cassie.cpp:95-112 calls ik::dls once per tick); this is what a C++
// caller with a batch of targets and several GPUs links against.  librccl is opened with dlopen (libikgpu.so stays loadable
// without it); a group of ONE device without RCCL falls back to a device-to-device copy of its own slot.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <time.h>

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
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

double now_us() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return 1e6 * static_cast<double>(ts.tv_sec) + 1e-3 * static_cast<double>(ts.tv_nsec);
}

// One worker thread per rank, bound to the rank's device: runs the job it is handed (the shard's launch) and reports its status.
struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = false, quit = false;
    int rc = IKGPU_OK;
    std::string err;
    double issue_us = 0.0;

    void start(int device) {
        th = std::thread([this, device] {
            (void)hipSetDevice(device);
            std::unique_lock<std::mutex> lock(mu);
            for (;;) {
                cv.wait(lock, [this] { return has_job || quit; });
                if (quit) return;
                std::function<int()> f = std::move(job);
                has_job = false;
                lock.unlock();
                const double t0 = now_us();
                const int r = f();
                const double t1 = now_us();
                const std::string msg = r == IKGPU_OK ? std::string() : std::string(ikgpu_last_error());   // (thread-local: carried over)
                lock.lock();
                rc = r; err = msg; issue_us = t1 - t0; done = true;
                cv.notify_all();
            }
        });
    }
    void submit(std::function<int()> f) {
        std::lock_guard<std::mutex> lock(mu);
        job = std::move(f); has_job = true; done = false;
        cv.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> lock(mu);
        cv.wait(lock, [this] { return done; });
        return rc;
    }
    void stop() {
        if (!th.joinable()) return;
        { std::lock_guard<std::mutex> lock(mu); quit = true; cv.notify_all(); }
        th.join();
    }
};

struct Rank {
    int device = 0;
    ikgpu_problem *problem = nullptr;
    hipStream_t stream = nullptr;
    ncclComm_t comm = nullptr;
    void *slot = nullptr;      // send buffer: this rank's packed slot
    size_t slot_cap = 0;
    hipEvent_t solved = nullptr;   // (loopback collective) recorded after the rank's solve
    std::unique_ptr<Worker> worker;
};

}  // namespace

struct ikgpu_shard_group {
    std::vector<Rank> ranks;
    int nq = 0, ntasks = 0;
    bool use_rccl = false, loopback = false;
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
    const char *lb = std::getenv("IKGPU_SHARD_LOOPBACK");
    const bool loopback = lb && lb[0] == '1';
    for (int i = 0; i < ndev; ++i)
        for (int j = 0; j < i; ++j)
            if (devices[i] == devices[j] && !loopback) return fail(IKGPU_ERR_INVALID, "a device appears twice in the group");
    std::unique_ptr<ikgpu_shard_group> g(new ikgpu_shard_group);
    g->loopback = loopback;
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
        else if (hipEventCreateWithFlags(&k.solved, hipEventDisableTiming) != hipSuccess) rc = fail(IKGPU_ERR_DEVICE, "hipEventCreate failed");
        if (rc == IKGPU_OK) {
            k.worker.reset(new Worker);
            k.worker->start(k.device);
        }
    }
    if (rc == IKGPU_OK) {
        ikgpu_flat_model flat;
        (void)ikgpu_model_get_flat(m, &flat);
        g->nq = flat.nq;
        const Rccl &R = rccl();
        if (loopback) {
            // (no communicator: the gather below is device-to-device copies)
        } else if (R.ok) {
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
        if (k.worker) k.worker->stop();
        DeviceScope scope(k.device);
        if (k.stream) (void)hipStreamSynchronize(k.stream);
        if (k.solved) (void)hipEventDestroy(k.solved);
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
    // 1. every device solves its shard straight into typed views of its send slot -- issued by the ranks' own threads, in parallel
    for (int r = 0; r < ndev; ++r)
        if (!q0[r] || !targets[r] || !gathered[r]) return fail(IKGPU_ERR_INVALID, "null per-device pointer");
    for (int r = 0; r < ndev; ++r) {
        Rank *k = &g->ranks[static_cast<size_t>(r)];
        const double *q0r = q0[r], *tr = targets[r];
        const int nq = g->nq;
        const bool loopback = g->loopback;
        k->worker->submit([k, r, ndev, total, slot, q0r, tr, params, nq, loopback]() -> int {
            if (k->slot_cap < slot) {
                if (k->slot) { (void)hipStreamSynchronize(k->stream); (void)hipFree(k->slot); k->slot = nullptr; k->slot_cap = 0; }
                if (hipMalloc(&k->slot, slot) != hipSuccess) return fail(IKGPU_ERR_DEVICE, "hipMalloc of the send slot failed");
                k->slot_cap = slot;
            }
            int64_t lo = 0, hi = 0;
            ikgpu_shard_range(total, r, ndev, &lo, &hi);
            size_t oq = 0, oi = 0, os = 0;
            (void)ikgpu_shard_slot_layout(nq, hi - lo, &oq, &oi, &os);
            char *base = static_cast<char *>(k->slot);
            const int rc = ikgpu_dls_solve_batch(k->problem, hi - lo, q0r, tr, params, reinterpret_cast<double *>(base + oq),
                                                 reinterpret_cast<uint8_t *>(base + os), reinterpret_cast<int32_t *>(base + oi), IKGPU_SOA, k->stream);
            if (rc == IKGPU_OK && loopback && hipEventRecord(k->solved, k->stream) != hipSuccess) return fail(IKGPU_ERR_DEVICE, "hipEventRecord failed");
            return rc;
        });
    }
    int first_rc = IKGPU_OK;
    std::string first_err;
    for (int r = 0; r < ndev; ++r) {   // (every worker is waited for, also after a failure: none may still be touching its arguments)
        Worker &w = *g->ranks[static_cast<size_t>(r)].worker;
        const int rc = w.wait();
        if (rc != IKGPU_OK && first_rc == IKGPU_OK) { first_rc = rc; first_err = w.err; }
    }
    if (first_rc != IKGPU_OK) return fail(first_rc, first_err);
    // 2. one all-gather: every device ends with [ndev][slot]
    if (g->loopback) {   // every rank's stream waits for every solve, then copies every slot into its own gathered buffer
        for (int r = 0; r < ndev; ++r) {
            Rank &k = g->ranks[static_cast<size_t>(r)];
            DeviceScope scope(k.device);
            for (int src = 0; src < ndev; ++src) {
                Rank &from = g->ranks[static_cast<size_t>(src)];
                if (src != r && hipStreamWaitEvent(k.stream, from.solved, 0) != hipSuccess) return fail(IKGPU_ERR_DEVICE, "hipStreamWaitEvent failed");
                if (hipMemcpyAsync(static_cast<char *>(gathered[r]) + static_cast<size_t>(src) * slot, from.slot, slot, hipMemcpyDeviceToDevice, k.stream) != hipSuccess)
                    return fail(IKGPU_ERR_DEVICE, "loopback gather: copy of a slot failed");
            }
        }
    } else if (g->use_rccl) {
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

double ikgpu_shard_group_last_issue_us(const ikgpu_shard_group *g, int32_t rank) {
    if (!g || rank < 0 || rank >= static_cast<int32_t>(g->ranks.size()) || !g->ranks[static_cast<size_t>(rank)].worker) return -1.0;
    Worker &w = *g->ranks[static_cast<size_t>(rank)].worker;
    std::lock_guard<std::mutex> lock(w.mu);
    return w.issue_us;
}

void *ikgpu_shard_group_stream(const ikgpu_shard_group *g, int32_t rank) {
    return g && rank >= 0 && rank < static_cast<int32_t>(g->ranks.size()) ? static_cast<void *>(g->ranks[static_cast<size_t>(rank)].stream) : nullptr;
}

}  // extern "C"
