// Store-pattern microbenchmark (tuning aid): one 256-lane workgroup per frame writes the frame's 196608 complex
// samples symbol by symbol like tf_kernel does -- 8 bytes per lane and store (A), or 16 bytes per lane and store (B).
// usage: store_bw [frames]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int kSym = 2552, kNull = 2656, kFrame = 196608;
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
        if (W != 2) {
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
    for (int w = 1; w <= 3; ++w) {
        for (int rep = 0; rep < 2; ++rep) {
            if (w == 1) hipLaunchKernelGGL(k<1>, dim3(B), dim3(256), 0, 0, d, 1.0f); else if (w == 2) hipLaunchKernelGGL(k<2>, dim3(B), dim3(256), 0, 0, d, 1.0f); else hipLaunchKernelGGL(k<3>, dim3(B), dim3(256), 0, 0, d, 1.0f);
        }
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int rep = 0; rep < 5; ++rep) {
            if (w == 1) hipLaunchKernelGGL(k<1>, dim3(B), dim3(256), 0, 0, d, 1.0f); else if (w == 2) hipLaunchKernelGGL(k<2>, dim3(B), dim3(256), 0, 0, d, 1.0f); else hipLaunchKernelGGL(k<3>, dim3(B), dim3(256), 0, 0, d, 1.0f);
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: %.0f GB/s (%.0f frames/s)\n", w == 1 ? " 8 bytes per lane and store" : w == 2 ? "16 bytes per lane and store" : " 8 bytes, rotated start symbol", 5.0 * B * kFrame * 8 / (ms * 1e-3) / 1e9, 5.0 * B / (ms * 1e-3));
    }
    return 0;
}
