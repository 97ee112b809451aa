// libvxm_comm.so: thin RCCL wrapper behind include/vxm_comm.h (one communicator per process, xGMI only on one node).
#include "../../include/vxm_comm.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>

namespace {

ncclComm_t g_comm = nullptr;
int g_world = 0;
thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define COMM_CHECK(expr, what)                                                                          \
    do {                                                                                                \
        const ncclResult_t r_ = (expr);                                                                 \
        if (r_ != ncclSuccess) return fail(100 + (int)r_, "%s: %s", what, ncclGetErrorString(r_));     \
    } while (0)

}  // namespace

extern "C" {

const char* vxm_comm_last_error_string(void) { return g_err; }

int vxm_comm_unique_id(void* out) {
    static_assert(sizeof(ncclUniqueId) == VXM_COMM_UNIQUE_ID_BYTES, "ncclUniqueId size");
    if (!out) return fail(1, "vxm_comm_unique_id: null pointer");
    ncclUniqueId id;
    COMM_CHECK(ncclGetUniqueId(&id), "ncclGetUniqueId");
    memcpy(out, &id, sizeof(id));
    return 0;
}

int vxm_comm_init(int rank, int world, const void* unique_id) {
    if (g_comm) return fail(2, "vxm_comm_init: communicator already initialised (one per process)");
    if (!unique_id || world < 1 || rank < 0 || rank >= world) return fail(1, "vxm_comm_init: bad arguments rank=%d world=%d", rank, world);
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    // ncclCommInitRank blocks until EVERY rank has joined: a rank that died before it (or never calls it) would leave the others
    // waiting for ever.  It runs on a helper thread bound to the caller's device; the caller waits at most
    // VXM_COMM_INIT_TIMEOUT_S seconds (default 180) and then reports an error instead of hanging (the helper is abandoned).
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(4, "vxm_comm_init: no current HIP device");
    struct Shared { std::mutex m; std::condition_variable cv; bool done = false, abandoned = false; ncclResult_t res = ncclSuccess; ncclComm_t comm = nullptr; };
    auto sh = std::make_shared<Shared>();
    std::thread([sh, dev, world, id, rank] {
        ncclComm_t c = nullptr;
        ncclResult_t r = hipSetDevice(dev) == hipSuccess ? ncclCommInitRank(&c, world, id, rank) : ncclUnhandledCudaError;
        bool late = false;
        {
            std::lock_guard<std::mutex> lk(sh->m);
            sh->res = r; sh->comm = c; sh->done = true;
            late = sh->abandoned;
            sh->cv.notify_all();
        }
        // the caller gave up on this rank (timeout below) and told its peers so: a communicator that completes afterwards belongs to
        // nobody -- abort it here, so that it neither leaks nor leaves peers that did finish with a member that reported failure
        if (late && r == ncclSuccess && c) (void)ncclCommAbort(c);
    }).detach();
    const char* e = getenv("VXM_COMM_INIT_TIMEOUT_S");
    const long secs = e && atol(e) > 0 ? atol(e) : 180;
    std::unique_lock<std::mutex> lk(sh->m);
    if (!sh->cv.wait_for(lk, std::chrono::seconds(secs), [&] { return sh->done; })) {
        sh->abandoned = true;
        return fail(5, "vxm_comm_init: ncclCommInitRank did not complete within %ld s (rank %d of %d): a rank is missing", secs, rank, world);
    }
    if (sh->res != ncclSuccess) return fail(100 + (int)sh->res, "ncclCommInitRank: %s", ncclGetErrorString(sh->res));
    g_comm = sh->comm;
    int n = 0;
    g_world = ncclCommCount(g_comm, &n) == ncclSuccess ? n : world;     // what RCCL itself says the communicator spans
    return 0;
}

int vxm_comm_rccl_version(void) {
    int v = 0;
    return ncclGetVersion(&v) == ncclSuccess ? v : 0;
}

int vxm_comm_world(void) { return g_world; }

int vxm_allreduce_sum_f32(float* buf, int64_t n, void* stream) {
    if (!g_comm) return fail(3, "vxm_allreduce_sum_f32: communicator not initialised");
    if (!buf || n <= 0) return fail(1, "vxm_allreduce_sum_f32: bad buffer");
    COMM_CHECK(ncclAllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, g_comm, reinterpret_cast<hipStream_t>(stream)), "ncclAllReduce");
    return 0;
}

int vxm_broadcast_f32(float* buf, int64_t n, int root, void* stream) {
    if (!g_comm) return fail(3, "vxm_broadcast_f32: communicator not initialised");
    if (!buf || n <= 0 || root < 0 || root >= g_world) return fail(1, "vxm_broadcast_f32: bad arguments");
    COMM_CHECK(ncclBroadcast(buf, buf, (size_t)n, ncclFloat32, root, g_comm, reinterpret_cast<hipStream_t>(stream)), "ncclBroadcast");
    return 0;
}

int vxm_comm_abort(void) {
    if (!g_comm) return 0;
    const ncclResult_t r = ncclCommAbort(g_comm);       // frees the communicator and abandons collectives that can no longer complete
    g_comm = nullptr;
    g_world = 0;
    if (r != ncclSuccess) return fail(100 + (int)r, "ncclCommAbort: %s", ncclGetErrorString(r));
    return 0;
}

int vxm_comm_destroy(void) {
    if (!g_comm) return 0;
    const ncclResult_t r = ncclCommDestroy(g_comm);
    g_comm = nullptr;
    g_world = 0;
    if (r != ncclSuccess) return fail(100 + (int)r, "ncclCommDestroy: %s", ncclGetErrorString(r));
    return 0;
}

}  // extern "C"
