// Store-pattern microbenchmark (tuning aid): one 256-lane workgroup per frame writes the frame's 196608 complex
// samples symbol by symbol like tf_kernel does -- 8 bytes per lane and store (A), or 16 bytes per lane and store (B).
// usage: store_bw [frames]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int kSym = 2552, kNull = 2656, kFrame = 196608;
// W = 7: one workgroup per SEGMENT (grid = frames x 77): workgroups that run together write adjacent memory
template <int CH> __global__ __launch_bounds__(256) void kseg(float2 *out, float v)
{
    const size_t frame = blockIdx.x / 77;
    const int s = (int)(blockIdx.x % 77);
    float2 *f = out + frame * kFrame + (s == 0 ? 0 : (size_t)kNull + (size_t)(s - 1) * kSym);
    const int len = s == 0 ? kNull : kSym;
    for (int i = threadIdx.x; i < len; i += 256) f[i] = make_float2(v + s, v);
}
// The frame kernel's stores exactly: lane t holds samples n = t + 256 m; the cyclic prefix (n >= 1544) goes to seg + (n - 1544),
// the body to seg + 504 + n.  A symbol is 20416 bytes = 159.5 cache lines, so every other symbol's wave stores (512 bytes) start
// in the middle of a 128-byte line.  ROT: those symbols use lane (t + 8) mod 256 for sample index t -- every wave store aligned.
template <bool ROT> __global__ __launch_bounds__(256) void kframe(float2 *out, float v, int work)
{
    float2 *f = out + (size_t)blockIdx.x * kFrame;
    const int t = threadIdx.x;
    for (int i = t; i < kNull; i += 256) f[i] = make_float2(v, v);
    size_t seg = kNull;
    for (int s = 1; s < 77; ++s) {
        const int tn = (ROT && (s & 1)) ? ((t + 8) & 255) : t;
        // `work` dependent FMAs per symbol and wave between the store bursts (the frame kernel computes ~1500 cycles' worth)
        for (int i = 0; i < work; ++i) v = __builtin_fmaf(v, 1.0000001f, 1e-9f);
        if (work > 0) __syncthreads();
#pragma unroll
        for (int m = 0; m < 8; ++m) f[seg + 504 + tn + 256 * m] = make_float2(v + s, v);
        if (tn >= 8) f[seg + tn - 8] = make_float2(v + s, v);
        f[seg + tn + 248] = make_float2(v + s, v);
        seg += kSym;
    }
}
// W = 3: as 1, but frame b starts at symbol (b * 29) % 77 and wraps (do the resident workgroups' equal offsets within
// their frames -- frames are 3 * 2^19 bytes apart -- cost bandwidth?)
template <int W> __global__ __launch_bounds__(256) void k(float2 *out, float v)
{
    float2 *f = out + (size_t)blockIdx.x * kFrame;
    const int t = threadIdx.x;
    size_t pos = 0;
    const int rot = W == 3 ? (int)((blockIdx.x * 29u) % 77u) : 0;
    for (int s0 = 0; s0 < 77; ++s0) {
        const int s = (s0 + rot) % 77;
        if (W == 3) pos = s == 0 ? 0 : (size_t)kNull + (size_t)(s - 1) * kSym;
        const int len = s == 0 ? kNull : kSym;
        if (W == 4) {           // non-temporal 8-byte stores
            for (int i = t; i < len; i += 256) {
                const float2 x = make_float2(v + s, v);
                __builtin_nontemporal_store(x.x, &f[pos + i].x);
                __builtin_nontemporal_store(x.y, &f[pos + i].y);
            }
        } else if (W == 5) {    // non-temporal 8-byte stores as one instruction
            for (int i = t; i < len; i += 256) {
                typedef float v2f __attribute__((ext_vector_type(2)));
                v2f x = {v + s, v};
                __builtin_nontemporal_store(x, reinterpret_cast<v2f *>(&f[pos + i]));
            }
        } else if (W == 6) {    // non-temporal 16-byte stores
            typedef float v4f __attribute__((ext_vector_type(4)));
            v4f *f4 = reinterpret_cast<v4f *>(f + pos);
            for (int i = t; i < len / 2; i += 256) { v4f x = {v + s, v, v, v}; __builtin_nontemporal_store(x, f4 + i); }
        } else if (W != 2) {
            for (int i = t; i < len; i += 256) f[pos + i] = make_float2(v + s, v);
        } else {
            float4 *f4 = reinterpret_cast<float4 *>(f + pos);
            for (int i = t; i < len / 2; i += 256) f4[i] = make_float4(v + s, v, v, v);
        }
        pos += len;
    }
}
int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 8192;
    float2 *d;
    CK(hipMalloc(&d, (size_t)B * kFrame * sizeof(float2)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 1; w <= 6; ++w) {
        for (int rep = 0; rep < 2; ++rep) {
            if (w == 1) hipLaunchKernelGGL(k<1>, dim3(B), dim3(256), 0, 0, d, 1.0f); else if (w == 2) hipLaunchKernelGGL(k<2>, dim3(B), dim3(256), 0, 0, d, 1.0f); else if (w == 3) hipLaunchKernelGGL(k<3>, dim3(B), dim3(256), 0, 0, d, 1.0f); else if (w == 4) hipLaunchKernelGGL(k<4>, dim3(B), dim3(256), 0, 0, d, 1.0f); else if (w == 5) hipLaunchKernelGGL(k<5>, dim3(B), dim3(256), 0, 0, d, 1.0f); else hipLaunchKernelGGL(k<6>, dim3(B), dim3(256), 0, 0, d, 1.0f);
        }
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int rep = 0; rep < 5; ++rep) {
            if (w == 1) hipLaunchKernelGGL(k<1>, dim3(B), dim3(256), 0, 0, d, 1.0f); else if (w == 2) hipLaunchKernelGGL(k<2>, dim3(B), dim3(256), 0, 0, d, 1.0f); else if (w == 3) hipLaunchKernelGGL(k<3>, dim3(B), dim3(256), 0, 0, d, 1.0f); else if (w == 4) hipLaunchKernelGGL(k<4>, dim3(B), dim3(256), 0, 0, d, 1.0f); else if (w == 5) hipLaunchKernelGGL(k<5>, dim3(B), dim3(256), 0, 0, d, 1.0f); else hipLaunchKernelGGL(k<6>, dim3(B), dim3(256), 0, 0, d, 1.0f);
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: %.0f GB/s (%.0f frames/s)\n", w == 1 ? " 8 bytes per lane and store" : w == 2 ? "16 bytes per lane and store" : w == 3 ? " 8 bytes, rotated start symbol" : w == 4 ? " 2 x 4 bytes non-temporal" : w == 5 ? " 8 bytes non-temporal" : "16 bytes non-temporal", 5.0 * B * kFrame * 8 / (ms * 1e-3) / 1e9, 5.0 * B / (ms * 1e-3));
    }
    for (int rot = 0; rot < 6; ++rot) {
        const int work = rot < 2 ? 0 : (rot - 1) * 100;      // 0, 0, 100 ... 400 dependent FMAs (~4-5 cycles each per wave)
        auto go = [&]() { if (rot == 1) hipLaunchKernelGGL(kframe<true>, dim3(B), dim3(256), 0, 0, d, 1.0f, 0); else hipLaunchKernelGGL(kframe<false>, dim3(B), dim3(256), 0, 0, d, 1.0f, work); };
        go(); go();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int rep = 0; rep < 5; ++rep) go();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms2;
        CK(hipEventElapsedTime(&ms2, e0, e1));
        char what[96];
        if (rot == 1) snprintf(what, sizeof what, ", odd symbols on rotated lanes (all wave stores line-aligned)");
        else if (rot) snprintf(what, sizeof what, ", %d dependent FMAs + a barrier per symbol", work);
        else what[0] = 0;
        printf("frame kernel's stores (prefix + body)%s: %.0f GB/s (%.0f frames/s)\n", what,
               5.0 * B * kFrame * 8 / (ms2 * 1e-3) / 1e9, 5.0 * B / (ms2 * 1e-3));
    }
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kseg<1>, dim3(B * 77), dim3(256), 0, 0, d, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(kseg<1>, dim3(B * 77), dim3(256), 0, 0, d, 1.0f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("one workgroup per segment, 8 bytes per lane: %.0f GB/s (%.0f frames/s)\n", 5.0 * B * kFrame * 8 / (ms * 1e-3) / 1e9, 5.0 * B / (ms * 1e-3));
    return 0;
}
