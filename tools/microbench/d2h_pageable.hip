// d2h_pageable.hip -- how fast does a device buffer reach PAGEABLE host memory (what a ModPlugin's Buffer is), as a
// function of the size of one hipMemcpyAsync and of the way a large copy is cut into pieces?  Decides the piece size of
// HostIO::out (dabgpu_ctx.h).   hipcc --offload-arch=gfx950 -O2 d2h_pageable.hip -o d2h_pageable
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t total = (size_t)192 << 20;
    char *d = nullptr, *pin[2] = {nullptr, nullptr};
    CK(hipMalloc(&d, total));
    CK(hipMemset(d, 1, total));
    char *h = (char *)aligned_alloc(4096, total);
    memset(h, 0, total);                                  // touched: no first-touch faults in the timed copies
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t pin_cap = (size_t)32 << 20;
    CK(hipHostMalloc((void **)&pin[0], pin_cap, hipHostMallocDefault));
    CK(hipHostMalloc((void **)&pin[1], pin_cap, hipHostMallocDefault));
    printf("%-44s %10s %10s\n", "method", "MB", "GB/s");
    for (size_t mb : {1, 4, 12, 25, 50, 100, 192}) {
        const size_t n = mb << 20;
        // (a) one hipMemcpyAsync into the pageable buffer
        for (int rep = 0; rep < 2; ++rep) {
            const double t0 = now();
            CK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            if (rep) printf("%-44s %10zu %10.1f\n", "one hipMemcpyAsync -> pageable", mb, n / (now() - t0) / 1e9);
        }
        // (b) pieces of P MB straight into the pageable buffer
        for (size_t pmb : {1, 2, 4, 8, 16}) {
            const size_t p = pmb << 20;
            if (p >= n) continue;
            for (int rep = 0; rep < 2; ++rep) {
                const double t0 = now();
                for (size_t o = 0; o < n; o += p) CK(hipMemcpyAsync(h + o, d + o, std::min(p, n - o), hipMemcpyDeviceToHost, s));
                CK(hipStreamSynchronize(s));
                if (rep) {
                    char name[64];
                    snprintf(name, sizeof name, "pieces of %zu MB -> pageable", pmb);
                    printf("%-44s %10zu %10.1f\n", name, mb, n / (now() - t0) / 1e9);
                }
            }
        }
        // (c) pieces through two pinned buffers, memcpy by this thread while the next piece is in flight
        for (size_t pmb : {2, 4, 8, 16}) {
            const size_t p = pmb << 20;
            if (p >= n) continue;
            hipEvent_t ev[2];
            CK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
            CK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
            for (int rep = 0; rep < 2; ++rep) {
                const double t0 = now();
                const size_t np = (n + p - 1) / p;
                for (size_t i = 0; i < np + 1; ++i) {
                    if (i < np) {
                        CK(hipMemcpyAsync(pin[i & 1], d + i * p, std::min(p, n - i * p), hipMemcpyDeviceToHost, s));
                        CK(hipEventRecord(ev[i & 1], s));
                    }
                    if (i > 0) {
                        CK(hipEventSynchronize(ev[(i - 1) & 1]));
                        memcpy(h + (i - 1) * p, pin[(i - 1) & 1], std::min(p, n - (i - 1) * p));
                    }
                }
                if (rep) {
                    char name[64];
                    snprintf(name, sizeof name, "pinned ring, pieces of %zu MB + memcpy", pmb);
                    printf("%-44s %10zu %10.1f\n", name, mb, n / (now() - t0) / 1e9);
                }
            }
        }
    }
    // (d) the pinned buffer itself (what the asynchronous path hands out)
    for (size_t mb : {8, 32}) {
        const size_t n = mb << 20;
        const double t0 = now();
        CK(hipMemcpyAsync(pin[0], d, n, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        printf("%-44s %10zu %10.1f\n", "one hipMemcpyAsync -> pinned", mb, n / (now() - t0) / 1e9);
    }
    return 0;
}
