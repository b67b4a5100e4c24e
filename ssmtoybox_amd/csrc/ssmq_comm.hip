// The path's only collective: sums of the Monte-Carlo error statistics over the ranks of one job, one
// process per GPU, RCCL over xGMI - behind the C ABI, no PyTorch.  A few KB per call: latency-bound, so the buffer goes
// host -> device -> ncclAllReduce -> host on the library's stream and the call returns the reduced values.
//
// librccl is opened at run time (dlopen) by the first ssmq_comm_* call: the compute entry points of libssmq carry no
// dependency on it, and single-process users never load it.
#include <dlfcn.h>
#include <atomic>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include "ssmq_host.h"

namespace {

constexpr int kIdBytes = 128;   // NCCL_UNIQUE_ID_BYTES
struct UniqueId { char internal[kIdBytes]; };
typedef void *comm_t;
enum { kSum = 0, kMax = 2, kFloat64 = 8 };   // ncclSum / ncclMax / ncclFloat64 of rccl.h

struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(comm_t *, int, UniqueId, int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
bool g_stub_marker = false;   // the loaded librccl is the stand-in of tests/stub_rccl (see host_staging)
comm_t g_comm = nullptr;
int g_rank = 0, g_world = 1;
double *g_dbuf = nullptr;
size_t g_dbuf_n = 0;
// stdout is parked on this descriptor while ncclCommInitRank runs (see ssmq_comm_init); -1 when stdout is in place
std::atomic<int> g_saved_stdout{-1};

void restore_stdout() {
    const int saved = g_saved_stdout.exchange(-1);
    if (saved >= 0) {
        fflush(stdout);
        dup2(saved, STDOUT_FILENO);
        close(saved);
    }
}

int load_rccl() {
    if (g_rccl.lib) return SSMQ_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) {
        ssmq::set_error(std::string("ssmq_comm: cannot open librccl: ") + dlerror());
        return SSMQ_E_UNSUPPORTED;
    }
    g_rccl.GetUniqueId = (int (*)(UniqueId *))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(comm_t *, int, UniqueId, int))dlsym(h, "ncclCommInitRank");
    g_rccl.AllReduce = (int (*)(const void *, void *, size_t, int, int, comm_t, hipStream_t))dlsym(h, "ncclAllReduce");
    g_rccl.CommDestroy = (int (*)(comm_t))dlsym(h, "ncclCommDestroy");
    g_rccl.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy) {
        ssmq::set_error("ssmq_comm: librccl lacks a required symbol");
        dlclose(h);
        return SSMQ_E_UNSUPPORTED;
    }
    g_rccl.lib = h;
    g_stub_marker = dlsym(h, "ssmq_stub_rccl_marker") != nullptr;
    if (ssmq::sw("SSMQ_COMM_HOST_STAGING") && ssmq::sw("SSMQ_COMM_HOST_STAGING")[0] == '1' && !g_stub_marker) {
        ssmq::set_error("ssmq_comm: SSMQ_COMM_HOST_STAGING=1 is a test hook for the stand-in librccl (tests/stub_rccl); the library "
                        "that was loaded is not the stand-in");
        return SSMQ_E_UNSUPPORTED;
    }
    return SSMQ_OK;
}

int nccl_fail(int rc, const char *what) {
    if (rc == 0) return SSMQ_OK;
    ssmq::set_error(std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error"));
    return SSMQ_E_HIP;
}

// SSMQ_COMM_HOST_STAGING=1: the buffers handed to ncclAllReduce are the caller's HOST arrays and no HIP call is made - for
// the stand-in librccl of tests/test_rccl_stub.py (which reduces through files), so that everything around the collective -
// id publication, ncclCommInitRank on every rank, op codes, barrier, destroy - runs on a machine without a GPU.  Never set it
// with the real RCCL: the switch is honoured only if the loaded library exports `ssmq_stub_rccl_marker`, which only the
// stand-in does - with the real library host pointers would otherwise reach ncclAllReduce.
bool host_staging() {
    static const bool asked = ssmq::sw("SSMQ_COMM_HOST_STAGING") && ssmq::sw("SSMQ_COMM_HOST_STAGING")[0] == '1';
    return asked && g_stub_marker;
}

int allreduce(double *buf, int64_t n, int op) {
    if (!buf || n < 0) {
        ssmq::set_error("ssmq_allreduce: bad argument");
        return SSMQ_E_ARG;
    }
    if (n == 0) return SSMQ_OK;
    if (!g_comm) {   // no communicator: fine for a single process, an error for a rank that never joined
        if (g_world == 1) return SSMQ_OK;
        ssmq::set_error("ssmq_allreduce: ssmq_comm_init has not been called");
        return SSMQ_E_ARG;
    }
    if (host_staging()) return nccl_fail(g_rccl.AllReduce(buf, buf, (size_t)n, kFloat64, op, g_comm, nullptr), "ncclAllReduce");
    int rc = ssmq::ensure_device();
    if (rc) return rc;
    hipStream_t s = ssmq::stream();
    if (g_dbuf_n < (size_t)n) {
        if (g_dbuf) hipFree(g_dbuf);
        g_dbuf = nullptr;
        g_dbuf_n = 0;
        SSMQ_HIP(hipMalloc(&g_dbuf, sizeof(double) * (size_t)n));
        g_dbuf_n = (size_t)n;
    }
    SSMQ_HIP(hipMemcpyAsync(g_dbuf, buf, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, s));
    if ((rc = nccl_fail(g_rccl.AllReduce(g_dbuf, g_dbuf, (size_t)n, kFloat64, op, g_comm, s), "ncclAllReduce"))) return rc;
    SSMQ_HIP(hipMemcpyAsync(buf, g_dbuf, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s));
    SSMQ_HIP(hipStreamSynchronize(s));
    return SSMQ_OK;
}

}  // namespace

extern "C" {

int ssmq_comm_unique_id(char *id, int len) {
    if (!id || len < kIdBytes) {
        ssmq::set_error("ssmq_comm_unique_id: buffer of at least 128 bytes required");
        return SSMQ_E_ARG;
    }
    int rc = load_rccl();
    if (rc) return rc;
    UniqueId u;
    memset(&u, 0, sizeof(u));
    if ((rc = nccl_fail(g_rccl.GetUniqueId(&u), "ncclGetUniqueId"))) return rc;
    memcpy(id, u.internal, kIdBytes);
    return SSMQ_OK;
}

int ssmq_comm_init(int rank, int world, const char *id, int len) {
    if (world < 1 || rank < 0 || rank >= world || (world > 1 && (!id || len < kIdBytes))) {
        ssmq::set_error("ssmq_comm_init: bad rank / world / id");
        return SSMQ_E_ARG;
    }
    if (g_comm) {
        ssmq::set_error("ssmq_comm_init: a communicator already exists (ssmq_comm_destroy first)");
        return SSMQ_E_ARG;
    }
    g_rank = rank;
    g_world = world;
    if (world == 1 && !id) return SSMQ_OK;   // single process: every reduction is the identity, RCCL is not loaded
    int rc = load_rccl();                                        // (first: host_staging() depends on what was loaded)
    if (rc) return rc;
    rc = host_staging() ? SSMQ_OK : ssmq::ensure_device();       // the communicator binds to the calling thread's current device
    if (rc) return rc;
    UniqueId u;
    memcpy(u.internal, id, kIdBytes);
    // RCCL prints a version banner on stdout while the communicator is created; callers own stdout (bench.py prints one
    // JSON line there), so it is sent to stderr for the duration of the call.  The saved descriptor is global: a caller
    // that runs this function on a helper thread and gives up on it (a peer never arrived) gets its stdout back with
    // ssmq_comm_abandon_init().
    fflush(stdout);
    const int saved = dup(STDOUT_FILENO);
    if (saved >= 0) {
        dup2(STDERR_FILENO, STDOUT_FILENO);
        g_saved_stdout.store(saved);
    }
    const int nrc = g_rccl.CommInitRank(&g_comm, world, u, rank);
    restore_stdout();
    return nccl_fail(nrc, "ncclCommInitRank");
}

int ssmq_comm_abandon_init(void) {
    restore_stdout();
    return SSMQ_OK;
}

int ssmq_comm_rank(void) { return g_rank; }
int ssmq_comm_world(void) { return g_world; }

int ssmq_allreduce_sum(double *buf, int64_t n) { return allreduce(buf, n, kSum); }
int ssmq_allreduce_max(double *buf, int64_t n) { return allreduce(buf, n, kMax); }

int ssmq_comm_barrier(void) {
    double one = 1.0;
    if (!host_staging()) {
        int rc = ssmq::ensure_device();
        if (rc) return rc;
        SSMQ_HIP(hipStreamSynchronize(ssmq::stream()));
    }
    return allreduce(&one, 1, kSum);
}

int ssmq_comm_destroy(void) {
    int rc = SSMQ_OK;
    if (g_comm) {
        if (!host_staging()) hipStreamSynchronize(ssmq::stream());
        rc = nccl_fail(g_rccl.CommDestroy(g_comm), "ncclCommDestroy");
        g_comm = nullptr;
    }
    if (g_dbuf) hipFree(g_dbuf);
    g_dbuf = nullptr;
    g_dbuf_n = 0;
    g_rank = 0;
    g_world = 1;
    return rc;
}

}  // extern "C"
