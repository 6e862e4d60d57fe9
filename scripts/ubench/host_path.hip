// host_path.hip -- what a host-memory entry point can be built from on this box: cost of pinning the caller's pages in
// place (hipHostRegister), copy rates from pageable / registered / hipHostMalloc'd memory, one direction and both at
// once, and the CPU's own memcpy into pinned staging.  (Round 3)
// Build: hipcc --offload-arch=gfx950 -O2 -o bin/host_path host_path.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t IN = 64u << 20, OUT = 128u << 20;
    void *d_in, *d_out;
    CK(hipMalloc(&d_in, IN)); CK(hipMalloc(&d_out, OUT));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    char *pg_in = (char *)aligned_alloc(4096, IN), *pg_out = (char *)aligned_alloc(4096, OUT);
    memset(pg_in, 1, IN); memset(pg_out, 2, OUT);
    void *pin_in, *pin_out;
    CK(hipHostMalloc(&pin_in, IN, hipHostMallocDefault)); CK(hipHostMalloc(&pin_out, OUT, hipHostMallocDefault));
    memset(pin_in, 1, IN); memset(pin_out, 2, OUT);
    auto both = [&](const char *what, void *hin, void *hout, int reps) {
        for (int w = 0; w < 2; ++w) { CK(hipMemcpyAsync(d_in, hin, IN, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(hout, d_out, OUT, hipMemcpyDeviceToHost, s2)); CK(hipDeviceSynchronize()); }
        double t0 = now();
        for (int i = 0; i < reps; ++i) { CK(hipMemcpyAsync(d_in, hin, IN, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); }
        double t1 = now();
        for (int i = 0; i < reps; ++i) { CK(hipMemcpyAsync(hout, d_out, OUT, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s2)); }
        double t2 = now();
        for (int i = 0; i < reps; ++i) { CK(hipMemcpyAsync(d_in, hin, IN, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(hout, d_out, OUT, hipMemcpyDeviceToHost, s2)); CK(hipDeviceSynchronize()); }
        double t3 = now();
        printf("%-34s H2D 64 MiB %6.2f ms (%5.1f GB/s)   D2H 128 MiB %6.2f ms (%5.1f GB/s)   both at once %6.2f ms\n", what,
               (t1 - t0) / reps * 1e3, IN / ((t1 - t0) / reps) / 1e9, (t2 - t1) / reps * 1e3, OUT / ((t2 - t1) / reps) / 1e9, (t3 - t2) / reps * 1e3);
    };
    both("pageable (hipMemcpyAsync stages)", pg_in, pg_out, 5);
    both("hipHostMalloc", pin_in, pin_out, 10);
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        CK(hipHostRegister(pg_in, IN, hipHostRegisterDefault));
        double t1 = now();
        CK(hipHostRegister(pg_out, OUT, hipHostRegisterDefault));
        double t2 = now();
        if (rep == 0) both("hipHostRegister'ed in place", pg_in, pg_out, 10);
        double t3 = now();
        CK(hipHostUnregister(pg_in)); CK(hipHostUnregister(pg_out));
        double t4 = now();
        printf("hipHostRegister 64 MiB %6.2f ms, 128 MiB %6.2f ms; unregister both %6.2f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t4 - t3) * 1e3);
    }
    // a fresh allocation, never touched (page faults inside the registration)
    { char *fresh = (char *)aligned_alloc(4096, OUT); double t0 = now(); CK(hipHostRegister(fresh, OUT, hipHostRegisterDefault)); double t1 = now();
      printf("hipHostRegister 128 MiB of untouched memory %6.2f ms\n", (t1 - t0) * 1e3); CK(hipHostUnregister(fresh)); free(fresh); }
    // CPU memcpy pageable -> pinned, 1..16 threads
    for (int nt : {1, 2, 4, 8, 16}) {
        double best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            double t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { size_t a = OUT / nt * t, b = (t == nt - 1) ? OUT : OUT / nt * (t + 1); memcpy(pg_out + a, (char *)pin_out + a, b - a); });
            for (auto &x : th) x.join();
            double dt = now() - t0; if (dt < best) best = dt;
        }
        printf("CPU memcpy pinned -> pageable 128 MiB, %2d threads: %6.2f ms (%5.1f GB/s)\n", nt, best * 1e3, OUT / best / 1e9);
    }
    // chunked both directions on two streams (8 chunks), registered-in-place equivalent = pinned
    for (int chunks : {2, 4, 8, 16}) {
        double t0 = now();
        for (int rep = 0; rep < 5; ++rep) {
            for (int c = 0; c < chunks; ++c) {
                CK(hipMemcpyAsync((char *)d_in + IN / chunks * c, (char *)pin_in + IN / chunks * c, IN / chunks, hipMemcpyHostToDevice, s1));
                CK(hipMemcpyAsync((char *)pin_out + OUT / chunks * c, (char *)d_out + OUT / chunks * c, OUT / chunks, hipMemcpyDeviceToHost, s2));
            }
            CK(hipDeviceSynchronize());
        }
        printf("pinned, %2d chunks per direction, two streams: %6.2f ms per 64 MiB in + 128 MiB out\n", chunks, (now() - t0) / 5 * 1e3);
    }
    return 0;
}
