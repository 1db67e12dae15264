// Store-bandwidth microbenchmark 2 (tuning aid): what separates a plain fill (7.0 TB/s on the faster boxes of the pool) from the
// frame kernel's store pattern (6.0 there, 5.3 on the slower ones; store_bw.hip)?  Fills of one 12 GiB buffer, varying the bytes a
// workgroup writes in a row, their alignment, which workgroup writes which piece, and how a workgroup's waves divide a piece.
// usage: fill_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// Workgroup b writes `pieces` pieces of `piece` bytes: piece j is number (j * gridDim + b) * run_len ... (run_len = 1: grid-stride;
// run_len = pieces: all of them in a row).  PAT 0: lane t writes 8 bytes at t * 8 + 2048 k (the frame kernel's pattern: a wave's
// stores 2 KB apart); 1: every wave writes a contiguous quarter of the piece; 2: 16 bytes per lane at t * 16 + 4096 k.
template <int PAT> __global__ __launch_bounds__(256) void fill(char *out, size_t piece, int pieces, int run_len, size_t skew, float v)
{
    const int t = threadIdx.x;
    for (int j = 0; j < pieces; ++j) {
        const size_t idx = ((size_t)(j / run_len) * gridDim.x + blockIdx.x) * run_len + j % run_len;
        char *p = out + idx * piece + skew;
        if (PAT == 0) {
            for (size_t o = (size_t)t * 8; o < piece; o += 2048) *reinterpret_cast<float2 *>(p + o) = make_float2(v, v);
        } else if (PAT == 1) {
            const size_t q = piece / 4;
            for (size_t o = (size_t)(t & 63) * 8; o < q; o += 512) *reinterpret_cast<float2 *>(p + (t >> 6) * q + o) = make_float2(v, v);
        } else {
            for (size_t o = (size_t)t * 16; o < piece; o += 4096) *reinterpret_cast<float4 *>(p + o) = make_float4(v, v, v, v);
        }
    }
}
template <int PAT> void run(char *d, size_t total, size_t piece, int pieces, int run_len, size_t skew, const char *what)
{
    const unsigned grid = (unsigned)(total / piece / pieces);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(fill<PAT>, dim3(grid), dim3(256), 0, 0, d, piece, pieces, run_len, skew, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(fill<PAT>, dim3(grid), dim3(256), 0, 0, d, piece, pieces, run_len, skew, 1.0f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-86s %5.0f GB/s\n", what, 4.0 * grid * piece * pieces / (ms * 1e-3) / 1e9);
}
int main()
{
    const size_t total = (size_t)12 << 30;
    char *d;
    CK(hipMalloc(&d, total + (1 << 20)));
    run<0>(d, total, 4096, 1, 1, 0, "4 KB per workgroup");
    run<2>(d, total, 4096, 1, 1, 0, "4 KB per workgroup, 16 B per lane");
    run<0>(d, total, 4096, 100, 1, 0, "100 x 4 KB per workgroup, grid-stride");
    run<0>(d, total, 4096, 5, 5, 0, "5 x 4 KB in a row per workgroup (= 20 KB)");
    run<0>(d, total, 20480, 1, 1, 0, "20 KB per workgroup");
    run<1>(d, total, 20480, 1, 1, 0, "20 KB per workgroup, each wave a contiguous quarter");
    run<2>(d, total, 20480, 1, 1, 0, "20 KB per workgroup, 16 B per lane");
    run<0>(d, total, 20416, 1, 1, 0, "20416 B per workgroup (a symbol: 64-byte aligned)");
    run<0>(d, total, 20416, 77, 77, 0, "77 x 20416 B in a row per workgroup (a frame)");
    run<1>(d, total, 20416, 77, 77, 0, "77 x 20416 B in a row, each wave a contiguous quarter");
    run<2>(d, total, 20416, 77, 77, 0, "77 x 20416 B in a row, 16 B per lane");
    run<0>(d, total, 20416, 77, 1, 0, "77 x 20416 B per workgroup, grid-stride");
    run<0>(d, total, 8192, 1, 1, 0, "8 KB per workgroup");
    run<0>(d, total, 16384, 1, 1, 0, "16 KB per workgroup");
    run<0>(d, total, 4096, 1, 1, 64, "4 KB per workgroup, skewed 64 B");
    run<0>(d, total, 4096, 1, 1, 2048, "4 KB per workgroup, skewed 2 KB");
    return 0;
}
