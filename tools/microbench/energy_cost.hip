// energy_cost.hip -- what does one wave-instruction COST IN ENERGY on gfx950?  (tuning aid, not product code)
// cfg 3 and cfg 4 run at the board's 1400 W power limit (bench.py measures it), so under the limit throughput is 1400 W
// divided by the energy of a frame.  This is the price list for that budget, the counterpart of issue_cost.hip's cycles:
// every kind runs ~2.5 s on every CU at 4 waves per SIMD (1024 workgroups of 256 lanes) while the host samples the socket's
// power (hwmon power1_input) and shader clock; the dynamic energy per wave-instruction is (W - W of an s_nop loop at the same
// occupancy) / rate.            hipcc --offload-arch=gfx950 -O2 energy_cost.hip -o energy_cost
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <glob.h>
#include <unistd.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum Kind { K_NOP, K_FMA, K_PKFMA, K_ADD_DPP, K_DSR64, K_DSW64, K_DSR32, K_DSR128, K_DSW128, K_GSTORE8, K_GLOAD8_L2, K_GSTORE8_NT, K_GSTORE16, K_GSTORE16_NT, K_GSTORE16_PAIR32, K_COUNT };
static const char *kNames[K_COUNT] = {"s_nop 0 (baseline: waves resident, clocks running)", "v_fma_f32", "v_pk_fma_f32", "v_add_f32 row_ror dpp",
                                      "ds_read_b64", "ds_write_b64", "ds_read_b32", "ds_read_b128", "ds_write_b128",
                                      "global_store_dwordx2 (streaming to HBM)", "global_load_dwordx2 (L2-resident 8 MB)",
                                      "global_store_dwordx2 nt (streaming to HBM)", "global_store_dwordx4 (streaming to HBM)",
                                      "global_store_dwordx4 nt (streaming to HBM)",
                                      "2 x global_store_dwordx4, 32 B per lane (the resampler's pattern)"};

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// 32 instructions of the kind per loop trip
template <int KIND> __global__ __launch_bounds__(256) void k(long iters, float *sink, float2 *gbuf, size_t gelems)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x;
    float a0 = 1.f + t, a1 = 2.f, a2 = 3.f, a3 = 4.f, a4 = 5.f, a5 = 6.f, a6 = 7.f, a7 = 8.f;
    v2f p0 = {1.f + t, 2.f}, p1 = {3.f, 4.f}, p2 = {5.f, 6.f}, p3 = {7.f, 8.f};
    const float m = 0.999f, c = 0.001f;
    const v2f m2 = {0.999f, 1.001f}, c2 = {0.001f, -0.001f};
    const unsigned a8 = (unsigned)(uintptr_t)smem + (unsigned)t * 8u, a4b = (unsigned)(uintptr_t)smem + (unsigned)t * 4u,
                   a16 = (unsigned)(uintptr_t)smem + (unsigned)t * 16u;
    v2f r0, r1, r2, r3;
    v4f q0, q1;
    float s0, s1, s2, s3;
    size_t gi = ((size_t)blockIdx.x * 256 + t) % gelems;
    const size_t gstride = (size_t)gridDim.x * 256;
    for (long i = 0; i < iters; ++i) {
        if (KIND == K_NOP) {
            asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\t"
                         "s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\t"
                         "s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\t"
                         "s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0" ::: "memory");
        } else if (KIND == K_FMA) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                             "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if (KIND == K_PKFMA) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(m2), "v"(c2));
        } else if (KIND == K_ADD_DPP) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                             "v_add_f32_dpp %2, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %3, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                             "v_add_f32_dpp %4, %4, %4 row_ror:4 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %5, %5, %5 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                             "v_add_f32_dpp %6, %6, %6 row_ror:4 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %7, %7, %7 row_ror:4 row_mask:0xf bank_mask:0xf"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == K_DSR64) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:2048\n\tds_read_b64 %2, %4 offset:4096\n\tds_read_b64 %3, %4 offset:6144\n\t"
                             "s_waitcnt lgkmcnt(0)" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a8) : "memory");
        } else if (KIND == K_DSW64) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:2048\n\tds_write_b64 %0, %3 offset:4096\n\tds_write_b64 %0, %4 offset:6144\n\t"
                             "s_waitcnt lgkmcnt(0)" :: "v"(a8), "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
        } else if (KIND == K_DSR32) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:1024\n\tds_read_b32 %2, %4 offset:2048\n\tds_read_b32 %3, %4 offset:3072\n\t"
                             "s_waitcnt lgkmcnt(0)" : "=v"(s0), "=v"(s1), "=v"(s2), "=v"(s3) : "v"(a4b) : "memory");
        } else if (KIND == K_DSR128) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:4096\n\ts_waitcnt lgkmcnt(0)" : "=v"(q0), "=v"(q1) : "v"(a16) : "memory");
        } else if (KIND == K_DSW128) {
            const v4f w = {p0.x, p0.y, p1.x, p1.y};
#pragma unroll
            for (int u = 0; u < 16; ++u)
                asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %1 offset:4096\n\ts_waitcnt lgkmcnt(0)" :: "v"(a16), "v"(w) : "memory");
        } else if (KIND == K_GSTORE8) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                gbuf[gi] = make_float2(a0, a1);
                gi += gstride;
                if (gi >= gelems) gi -= gelems;
            }
        } else if (KIND == K_GSTORE8_NT) {
            typedef float v2f_ __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                __builtin_nontemporal_store((v2f_){a0, a1}, reinterpret_cast<v2f_ *>(gbuf + gi));
                gi += gstride;
                if (gi >= gelems) gi -= gelems;
            }
        } else if (KIND == K_GSTORE16 || KIND == K_GSTORE16_NT) {
            typedef float v4f_ __attribute__((ext_vector_type(4)));
            v4f_ *g4 = reinterpret_cast<v4f_ *>(gbuf);
            size_t g = gi % (gelems >> 1);                       // 16-byte elements, one per lane
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                if (KIND == K_GSTORE16_NT) __builtin_nontemporal_store((v4f_){a0, a1, a2, a3}, g4 + g);
                else g4[g] = (v4f_){a0, a1, a2, a3};
                g += gstride;
                if (g >= (gelems >> 1)) g -= (gelems >> 1);
            }
        } else if (KIND == K_GSTORE16_PAIR32) {
            typedef float v4f_ __attribute__((ext_vector_type(4)));
            v4f_ *g4 = reinterpret_cast<v4f_ *>(gbuf);
            size_t g = (2 * gi) % (gelems >> 1);                 // 16-byte elements, two per lane, side by side
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                g4[g] = (v4f_){a0, a1, a2, a3};
                g4[g + 1] = (v4f_){a4, a5, a6, a7};
                g += 2 * gstride;
                if (g >= (gelems >> 1)) g -= (gelems >> 1);
            }
        } else if (KIND == K_GLOAD8_L2) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const float2 v = gbuf[(gi + (size_t)u * 256) & ((1u << 20) - 1)];
                a0 += v.x; a1 += v.y;
            }
        }
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x == 12345.678f) sink[t] = a0 + r0.x + r1.x + r2.x + r3.x + q0.x + q1.x + s0 + s1 + s2 + s3;
}

static std::vector<std::string> hwmons()
{
    std::vector<std::string> v;
    glob_t g;
    if (glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input", 0, nullptr, &g) == 0)
        for (size_t i = 0; i < g.gl_pathc; ++i) { std::string p = g.gl_pathv[i]; v.push_back(p.substr(0, p.rfind('/'))); }
    globfree(&g);
    return v;
}
static double rd(const std::string &p) { FILE *f = fopen(p.c_str(), "r"); double x = 0; if (f) { if (fscanf(f, "%lf", &x) != 1) x = 0; fclose(f); } return x; }

template <int KIND> static void launch(long iters, float *sink, float2 *g, size_t ge, hipStream_t s)
{
    hipLaunchKernelGGL(k<KIND>, dim3(1024), dim3(256), 16384, s, iters, sink, g, ge);
}
typedef void (*Fn)(long, float *, float2 *, size_t, hipStream_t);

int main(int argc, char **argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 2.5;
    float *sink; float2 *gbuf;
    const size_t ge = (size_t)1 << 29;                       // 4 GiB of float2: a streaming store target
    CK(hipMalloc(&sink, 4096)); CK(hipMalloc(&gbuf, ge * sizeof(float2)));
    CK(hipMemset(gbuf, 0, ge * sizeof(float2)));
    hipStream_t s; CK(hipStreamCreate(&s));
    const Fn fns[K_COUNT] = {launch<K_NOP>, launch<K_FMA>, launch<K_PKFMA>, launch<K_ADD_DPP>, launch<K_DSR64>, launch<K_DSW64>, launch<K_DSR32>,
                             launch<K_DSR128>, launch<K_DSW128>, launch<K_GSTORE8>, launch<K_GLOAD8_L2>, launch<K_GSTORE8_NT>, launch<K_GSTORE16>, launch<K_GSTORE16_NT>, launch<K_GSTORE16_PAIR32>};
    const auto mons = hwmons();
    std::vector<double> idle;
    for (auto &m : mons) idle.push_back(rd(m + "/power1_input"));
    printf("%-52s %9s %8s %8s %14s %12s\n", "32 x this per loop trip, 4 waves per SIMD", "G w-i/s", "W", "sclk MHz", "nJ per wave-i", "pJ per lane");
    double w_base = 0, base_rate = 0;
    for (int kind = 0; kind < K_COUNT; ++kind) {
        // calibrate, then one launch of about `seconds`
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        long it = 2000;
        fns[kind](it, sink, gbuf, ge, s); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s)); fns[kind](it, sink, gbuf, ge, s); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        const long iters = (long)(it * seconds * 1e3 / (ms > 0.01f ? ms : 0.01f));
        std::atomic<bool> done{false};
        std::vector<std::vector<double>> pw, fq;
        std::thread sampler([&] {
            while (!done.load()) {
                std::vector<double> p, f;
                for (auto &m : mons) { p.push_back(rd(m + "/power1_input")); f.push_back(rd(m + "/freq1_input")); }
                pw.push_back(p); fq.push_back(f);
                usleep(20000);
            }
        });
        CK(hipEventRecord(e0, s)); fns[kind](iters, sink, gbuf, ge, s); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        done = true; sampler.join();
        CK(hipEventElapsedTime(&ms, e0, e1));
        // the card whose power rose; the settled last two fifths
        size_t n = pw.size(), c = 0; double best = -1e30, W = 0, F = 0;
        for (size_t m = 0; m < mons.size(); ++m) {
            double a = 0; size_t cnt = 0;
            for (size_t i = 3 * n / 5; i < n; ++i) { a += pw[i][m]; ++cnt; }
            a = cnt ? a / cnt : 0;
            if (a - idle[m] > best) { best = a - idle[m]; c = m; }
        }
        { size_t cnt = 0; for (size_t i = 3 * n / 5; i < n; ++i) { W += pw[i][c]; F += fq[i][c]; ++cnt; } if (cnt) { W /= cnt; F /= cnt; } }
        W /= 1e6; F /= 1e6;
        const double rate = 1024.0 * 4 * 32.0 * (double)iters / (ms * 1e-3);      // wave-instructions per second
        if (kind == K_NOP) { w_base = W; base_rate = rate; }
        const double nj = kind == K_NOP ? 0.0 : (W - w_base) / rate * 1e9;
        printf("%-52s %9.1f %8.0f %8.0f %14.3f %12.2f\n", kNames[kind], rate / 1e9, W, F, nj, nj * 1e3 / 64.0);
        fflush(stdout);
        sleep(1);
    }
    (void)base_rate;
    return 0;
}
