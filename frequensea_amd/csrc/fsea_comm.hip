// fsea_comm.hip -- libfsea_rccl.so: the gather of output tiles to the root GPU (include/fsea_comm.h).
// RCCL grouped send/recv over xGMI between distinct devices; an event-ordered peer-copy backend when
// members share a device.
#include "../../include/fsea_comm.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return -1;
}

#define COMM_HIP(call)                                                                              \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) return fail("%s failed: %s", #call, hipGetErrorString(e_));           \
    } while (0)
#define COMM_NCCL(call)                                                                             \
    do {                                                                                            \
        ncclResult_t r_ = (call);                                                                   \
        if (r_ != ncclSuccess) return fail("%s failed: %s", #call, ncclGetErrorString(r_));         \
    } while (0)

struct Post {  // what a non-root member publishes for one gather of the "copy" backend
    const void *src = nullptr;
    hipEvent_t ready = nullptr;  // recorded on the member's stream: its tile is complete
    unsigned long long seq = 0;  // gather number this post belongs to
};

}  // namespace

struct fsea_comm {
    int n = 0;
    std::vector<int> devices;
    bool rccl = false;
    std::vector<ncclComm_t> comms;
    // "copy" backend and barrier: host rendezvous
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Post> posts;
    std::vector<unsigned long long> gather_seq;  // per member: gathers issued so far
    unsigned long long root_done = 0;            // gathers the root has consumed
    int barrier_count = 0;
    unsigned long long barrier_gen = 0;
};

extern "C" {

const char *fsea_comm_last_error(void) { return g_err.c_str(); }

int fsea_comm_create(fsea_comm **out, int n_members, const int *devices) {
    if (!out || n_members <= 0 || !devices) return fail("fsea_comm_create: bad arguments");
    *out = nullptr;
    int count = 0;
    COMM_HIP(hipGetDeviceCount(&count));
    for (int m = 0; m < n_members; ++m) {
        if (devices[m] < 0 || devices[m] >= count) return fail("member %d: device %d out of range [0,%d)", m, devices[m], count);
    }
    fsea_comm *c = new fsea_comm();
    c->n = n_members;
    c->devices.assign(devices, devices + n_members);
    c->posts.resize(n_members);
    c->gather_seq.assign(n_members, 0);
    const std::set<int> distinct(c->devices.begin(), c->devices.end());
    const char *env = std::getenv("FSEA_COMM_BACKEND");
    c->rccl = (int)distinct.size() == n_members && n_members > 1 && !(env && std::strcmp(env, "copy") == 0);
    if (c->rccl) {
        c->comms.resize(n_members);
        ncclResult_t r = ncclCommInitAll(c->comms.data(), n_members, c->devices.data());
        if (r != ncclSuccess) {
            // FSEA_COMM_BACKEND=rccl insists; otherwise the gather still works over event-ordered peer copies, and says so
            if (env && std::strcmp(env, "rccl") == 0) {
                delete c;
                return fail("ncclCommInitAll failed: %s", ncclGetErrorString(r));
            }
            std::fprintf(stderr, "fsea_comm: *** ncclCommInitAll over %d devices FAILED (%s): falling back to the \"copy\" backend "
                                 "(hipMemcpyPeerAsync); set FSEA_COMM_BACKEND=rccl to make this fatal ***\n",
                         n_members, ncclGetErrorString(r));
            c->comms.clear();
            c->rccl = false;
        }
    }
    if (!c->rccl) {
        for (int m = 0; m < n_members; ++m) {
            if (hipSetDevice(devices[m]) != hipSuccess ||
                hipEventCreateWithFlags(&c->posts[m].ready, hipEventDisableTiming) != hipSuccess) {
                delete c;
                return fail("event setup failed on device %d", devices[m]);
            }
        }
        // peer access for hipMemcpyPeerAsync between distinct devices (ignored when already enabled)
        for (int a : distinct) {
            for (int b : distinct) {
                if (a == b) continue;
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can && hipSetDevice(a) == hipSuccess) {
                    (void)hipDeviceEnablePeerAccess(b, 0);
                    (void)hipGetLastError();
                }
            }
        }
    }
    *out = c;
    return 0;
}

int fsea_comm_destroy(fsea_comm *c) {
    if (!c) return 0;
    for (auto &k : c->comms) {
        if (k) (void)ncclCommDestroy(k);
    }
    for (int m = 0; m < c->n; ++m) {
        if (c->posts[m].ready) {
            (void)hipSetDevice(c->devices[m]);
            (void)hipEventDestroy(c->posts[m].ready);
        }
    }
    delete c;
    return 0;
}

int fsea_comm_size(const fsea_comm *c) { return c ? c->n : 0; }
const char *fsea_comm_backend(const fsea_comm *c) { return (c && c->rccl) ? "rccl" : "copy"; }

int fsea_comm_stream_create(int device, void **stream) {
    if (!stream) return fail("NULL out-pointer");
    COMM_HIP(hipSetDevice(device));
    hipStream_t s = nullptr;
    COMM_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return 0;
}

int fsea_comm_stream_destroy(int device, void *stream) {
    if (!stream) return 0;
    COMM_HIP(hipSetDevice(device));
    COMM_HIP(hipStreamDestroy(static_cast<hipStream_t>(stream)));
    return 0;
}

int fsea_comm_gather(fsea_comm *c, int member, const void *d_src, const size_t *bytes, const size_t *offsets,
                     void *d_dst_root, void *stream) {
    if (!c || member < 0 || member >= c->n || !bytes || !offsets) return fail("fsea_comm_gather: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    COMM_HIP(hipSetDevice(c->devices[member]));
    if (member == 0 && bytes[0] > 0) {
        if (!d_dst_root || !d_src) return fail("fsea_comm_gather: NULL buffer on the root");
        COMM_HIP(hipMemcpyAsync(static_cast<char *>(d_dst_root) + offsets[0], d_src, bytes[0], hipMemcpyDeviceToDevice, s));
    }
    if (c->rccl) {
        if (member != 0) {
            if (bytes[member] > 0) COMM_NCCL(ncclSend(d_src, bytes[member], ncclUint8, 0, c->comms[member], s));
            return 0;
        }
        COMM_NCCL(ncclGroupStart());
        for (int m = 1; m < c->n; ++m) {
            if (bytes[m] > 0) {
                COMM_NCCL(ncclRecv(static_cast<char *>(d_dst_root) + offsets[m], bytes[m], ncclUint8, m, c->comms[0], s));
            }
        }
        COMM_NCCL(ncclGroupEnd());
        return 0;
    }
    // "copy" backend: members publish (pointer, ready event); the root pulls
    const unsigned long long seq = ++c->gather_seq[member];
    if (member != 0) {
        {
            // the previous post of this member must have been consumed before it is overwritten
            std::unique_lock<std::mutex> lock(c->mu);
            c->cv.wait(lock, [&] { return c->root_done >= seq - 1; });
        }
        if (bytes[member] > 0) COMM_HIP(hipEventRecord(c->posts[member].ready, s));
        {
            std::lock_guard<std::mutex> lock(c->mu);
            c->posts[member].src = d_src;
            c->posts[member].seq = seq;
        }
        c->cv.notify_all();
        return 0;
    }
    for (int m = 1; m < c->n; ++m) {
        const void *src = nullptr;
        {
            std::unique_lock<std::mutex> lock(c->mu);
            c->cv.wait(lock, [&] { return c->posts[m].seq >= seq; });
            src = c->posts[m].src;
        }
        if (bytes[m] == 0) continue;
        COMM_HIP(hipStreamWaitEvent(s, c->posts[m].ready, 0));
        COMM_HIP(hipMemcpyPeerAsync(static_cast<char *>(d_dst_root) + offsets[m], c->devices[0], src, c->devices[m],
                                    bytes[m], s));
    }
    {
        std::lock_guard<std::mutex> lock(c->mu);
        c->root_done = seq;
    }
    c->cv.notify_all();
    return 0;
}

int fsea_comm_selftest_rccl(int device, size_t bytes) {
    if (bytes == 0) return fail("fsea_comm_selftest_rccl: bytes must be positive");
    COMM_HIP(hipSetDevice(device));
    std::vector<unsigned char> host(bytes), back(bytes);
    for (size_t i = 0; i < bytes; ++i) host[i] = (unsigned char)((i * 2654435761u) >> 13);
    unsigned char *d_src = nullptr, *d_dst = nullptr;
    hipStream_t s = nullptr;
    ncclComm_t comm = nullptr;
    int rc = 0;
    auto body = [&]() -> int {
        COMM_HIP(hipMalloc(reinterpret_cast<void **>(&d_src), bytes));
        COMM_HIP(hipMalloc(reinterpret_cast<void **>(&d_dst), bytes));
        COMM_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        COMM_HIP(hipMemcpy(d_src, host.data(), bytes, hipMemcpyHostToDevice));
        COMM_HIP(hipMemset(d_dst, 0, bytes));
        COMM_NCCL(ncclCommInitAll(&comm, 1, &device));
        COMM_NCCL(ncclGroupStart());
        COMM_NCCL(ncclSend(d_src, bytes, ncclUint8, 0, comm, s));
        COMM_NCCL(ncclRecv(d_dst, bytes, ncclUint8, 0, comm, s));
        COMM_NCCL(ncclGroupEnd());
        COMM_HIP(hipStreamSynchronize(s));
        COMM_HIP(hipMemcpy(back.data(), d_dst, bytes, hipMemcpyDeviceToHost));
        if (std::memcmp(host.data(), back.data(), bytes) != 0) return fail("RCCL self send/recv of %zu bytes arrived altered", bytes);
        return 0;
    };
    rc = body();
    if (comm) (void)ncclCommDestroy(comm);
    if (s) (void)hipStreamDestroy(s);
    if (d_src) (void)hipFree(d_src);
    if (d_dst) (void)hipFree(d_dst);
    return rc;
}

int fsea_comm_barrier(fsea_comm *c, int member, void *stream) {
    if (!c || member < 0 || member >= c->n) return fail("fsea_comm_barrier: bad arguments");
    COMM_HIP(hipSetDevice(c->devices[member]));
    // two phases: everyone has issued its work, then the root's stream (which holds the receives /
    // pulls) and the member's own stream are drained, then everyone leaves together
    auto rendezvous = [&] {
        std::unique_lock<std::mutex> lock(c->mu);
        const unsigned long long gen = c->barrier_gen;
        if (++c->barrier_count == c->n) {
            c->barrier_count = 0;
            ++c->barrier_gen;
            c->cv.notify_all();
        } else {
            c->cv.wait(lock, [&] { return c->barrier_gen != gen; });
        }
    };
    rendezvous();
    COMM_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    rendezvous();
    return 0;
}

}  // extern "C"
