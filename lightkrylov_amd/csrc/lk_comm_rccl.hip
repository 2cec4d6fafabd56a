// lk_comm_rccl.hip -- native RCCL sum all-reduce for the row-sharded engine (one process per GPU).
//
// The reference has no collective (a distributed user does the reduction "inside dot",
// paper/paper.md:35,97,101); here the <= 129 (258 complex) reduction scalars of each panel sweep are
// summed over the ranks by ncclAllReduce(ncclDouble, ncclSum) on the context's own stream, in place in
// device memory -- no host round trip, no interpreter in the path.  Layered on the public ABI:
// lk_comm_init_rank installs an lk_allreduce_fn through lk_set_allreduce.
//
// librccl is opened lazily (dlopen) the first time a communicator is requested, re-using an already
// loaded copy if the process has one (PyTorch ships its own), so single-GPU users of the library never
// load it.
#include "lk_internal.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <map>
#include <mutex>
#include <stddef.h>

namespace {

// the RCCL entry points used, typed from rccl.h but resolved with dlsym (no DT_NEEDED on librccl)
struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    decltype(&ncclSend) send = nullptr;
    decltype(&ncclRecv) recv = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
};
static_assert(sizeof(ncclUniqueId) == LK_COMM_ID_BYTES, "LK_COMM_ID_BYTES must equal NCCL_UNIQUE_ID_BYTES");
Rccl g_rccl;
std::mutex g_mu;
struct CommState { ncclComm_t comm; int nranks, rank; };
std::map<lk_context_t, CommState *> g_comms;

int load_rccl() {
    if (g_rccl.handle) return LK_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *nm : names) {            // a copy already mapped into the process wins
        h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        if (h) break;
    }
    for (size_t i = 0; !h && i < sizeof(names) / sizeof(names[0]); ++i) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!h) return lk_fail_(LK_ERR_COMM, "lk_comm: cannot open librccl (%s)", dlerror());
    g_rccl.get_unique_id = (decltype(g_rccl.get_unique_id))dlsym(h, "ncclGetUniqueId");
    g_rccl.comm_init_rank = (decltype(g_rccl.comm_init_rank))dlsym(h, "ncclCommInitRank");
    g_rccl.comm_destroy = (decltype(g_rccl.comm_destroy))dlsym(h, "ncclCommDestroy");
    g_rccl.all_reduce = (decltype(g_rccl.all_reduce))dlsym(h, "ncclAllReduce");
    g_rccl.error_string = (decltype(g_rccl.error_string))dlsym(h, "ncclGetErrorString");
    g_rccl.send = (decltype(g_rccl.send))dlsym(h, "ncclSend");
    g_rccl.recv = (decltype(g_rccl.recv))dlsym(h, "ncclRecv");
    g_rccl.group_start = (decltype(g_rccl.group_start))dlsym(h, "ncclGroupStart");
    g_rccl.group_end = (decltype(g_rccl.group_end))dlsym(h, "ncclGroupEnd");
    g_rccl.all_gather = (decltype(g_rccl.all_gather))dlsym(h, "ncclAllGather");
    if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.comm_destroy || !g_rccl.all_reduce || !g_rccl.send ||
        !g_rccl.recv || !g_rccl.group_start || !g_rccl.group_end || !g_rccl.all_gather)
        return lk_fail_(LK_ERR_COMM, "lk_comm: librccl lacks a required symbol");
    g_rccl.handle = h;
    return LK_OK;
}

const char *errstr(ncclResult_t rc) { return g_rccl.error_string ? g_rccl.error_string(rc) : "?"; }

int rccl_sum(void *user, void *dev_buf, int64_t count, void *stream) {
    CommState *st = (CommState *)user;
    return g_rccl.all_reduce(dev_buf, dev_buf, (size_t)count, ncclDouble, ncclSum, st->comm, (hipStream_t)stream) == ncclSuccess ? 0 : 1;
}

// one grid line / one point to and from the ranks that own the neighbouring row blocks
int rccl_halo(void *user, const void *send_lo, const void *send_hi, void *recv_lo, void *recv_hi, int64_t count, void *stream) {
    CommState *st = (CommState *)user;
    hipStream_t s = (hipStream_t)stream;
    bool ok = g_rccl.group_start() == ncclSuccess;
    if (send_lo && st->rank > 0) ok = ok && g_rccl.send(send_lo, (size_t)count, ncclDouble, st->rank - 1, st->comm, s) == ncclSuccess;
    if (recv_lo && st->rank > 0) ok = ok && g_rccl.recv(recv_lo, (size_t)count, ncclDouble, st->rank - 1, st->comm, s) == ncclSuccess;
    if (send_hi && st->rank + 1 < st->nranks) ok = ok && g_rccl.send(send_hi, (size_t)count, ncclDouble, st->rank + 1, st->comm, s) == ncclSuccess;
    if (recv_hi && st->rank + 1 < st->nranks) ok = ok && g_rccl.recv(recv_hi, (size_t)count, ncclDouble, st->rank + 1, st->comm, s) == ncclSuccess;
    ok = (g_rccl.group_end() == ncclSuccess) && ok;
    return ok ? 0 : 1;
}

// every rank's row block of a vector to every rank (the input of a row-sharded dense / CSR matvec).  Equal blocks laid out in
// rank order are one ncclAllGather; otherwise (the last rank of row_partition holds the remainder) every pair exchanges its
// blocks by ncclSend / ncclRecv in one group and the own block is a device copy.
int rccl_allgatherv(void *user, const void *send, void *recv, const int64_t *counts, const int64_t *displs, int nranks, void *stream) {
    CommState *st = (CommState *)user;
    hipStream_t s = (hipStream_t)stream;
    if (nranks != st->nranks) return 1;
    bool uniform = true;
    for (int r = 0; r < nranks; ++r) uniform = uniform && counts[r] == counts[0] && displs[r] == (int64_t)r * counts[0];
    if (uniform)
        return g_rccl.all_gather(send, recv, (size_t)counts[0], ncclDouble, st->comm, s) == ncclSuccess ? 0 : 1;
    double *out = (double *)recv;
    if (counts[st->rank] > 0 &&
        hipMemcpyAsync(out + displs[st->rank], send, (size_t)counts[st->rank] * sizeof(double), hipMemcpyDeviceToDevice, s) != hipSuccess)
        return 1;
    bool ok = g_rccl.group_start() == ncclSuccess;
    for (int r = 0; r < nranks; ++r) {
        if (r == st->rank) continue;
        if (counts[st->rank] > 0) ok = ok && g_rccl.send(send, (size_t)counts[st->rank], ncclDouble, r, st->comm, s) == ncclSuccess;
        if (counts[r] > 0) ok = ok && g_rccl.recv(out + displs[r], (size_t)counts[r], ncclDouble, r, st->comm, s) == ncclSuccess;
    }
    ok = (g_rccl.group_end() == ncclSuccess) && ok;
    return ok ? 0 : 1;
}

}  // namespace

extern "C" {

int lk_comm_available(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    return load_rccl();
}

int lk_comm_get_unique_id(void *id_out) {
    if (!id_out) return lk_fail_(LK_ERR_INVALID, "lk_comm_get_unique_id: null id");
    std::lock_guard<std::mutex> lock(g_mu);
    int rc = load_rccl();
    if (rc != LK_OK) return rc;
    ncclUniqueId id;
    const ncclResult_t nrc = g_rccl.get_unique_id(&id);
    if (nrc != ncclSuccess) return lk_fail_(LK_ERR_COMM, "ncclGetUniqueId failed: %s", errstr(nrc));
    __builtin_memcpy(id_out, id.internal, LK_COMM_ID_BYTES);
    return LK_OK;
}

int lk_comm_init_rank(lk_context_t ctx, int nranks, int rank, const void *id) {
    if (!ctx || !id) return lk_fail_(LK_ERR_INVALID, "lk_comm_init_rank: null argument");
    if (nranks < 1 || rank < 0 || rank >= nranks) return lk_fail_(LK_ERR_INVALID, "lk_comm_init_rank: bad rank %d/%d", rank, nranks);
    std::lock_guard<std::mutex> lock(g_mu);
    int rc = load_rccl();
    if (rc != LK_OK) return rc;
    if (g_comms.count(ctx)) return lk_fail_(LK_ERR_INVALID, "lk_comm_init_rank: context already has a communicator");
    int device = 0;
    rc = lk_context_info(ctx, &device, nullptr);
    if (rc != LK_OK) return rc;
    if (hipSetDevice(device) != hipSuccess) return lk_fail_(LK_ERR_HIP, "lk_comm_init_rank: hipSetDevice(%d) failed", device);
    ncclUniqueId uid;
    __builtin_memcpy(uid.internal, id, LK_COMM_ID_BYTES);
    ncclComm_t comm = nullptr;
    const ncclResult_t nrc = g_rccl.comm_init_rank(&comm, nranks, uid, rank);
    if (nrc != ncclSuccess) return lk_fail_(LK_ERR_COMM, "ncclCommInitRank(%d/%d) failed: %s", rank, nranks, errstr(nrc));
    CommState *st = new CommState{comm, nranks, rank};
    rc = lk_set_allreduce(ctx, rccl_sum, st, nranks, rank);
    if (rc == LK_OK) rc = lk_set_halo_exchange(ctx, rccl_halo, st);
    if (rc == LK_OK) rc = lk_set_allgather(ctx, rccl_allgatherv, st);
    if (rc != LK_OK) {
        (void)g_rccl.comm_destroy(comm);
        delete st;
        return rc;
    }
    g_comms[ctx] = st;
    return LK_OK;
}

int lk_comm_destroy(lk_context_t ctx) {
    if (!ctx) return LK_OK;
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_comms.find(ctx);
    if (it == g_comms.end()) return LK_OK;
    (void)lk_sync(ctx);
    (void)lk_set_allreduce(ctx, nullptr, nullptr, 1, 0);
    (void)lk_set_halo_exchange(ctx, nullptr, nullptr);
    (void)lk_set_allgather(ctx, nullptr, nullptr);
    const ncclResult_t nrc = g_rccl.comm_destroy(it->second->comm);
    delete it->second;
    g_comms.erase(it);
    if (nrc != ncclSuccess) return lk_fail_(LK_ERR_COMM, "ncclCommDestroy failed: %s", errstr(nrc));
    return LK_OK;
}

}  // extern "C"
