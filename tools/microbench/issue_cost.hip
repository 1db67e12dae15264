// Instruction-cost microbenchmark for gfx950 (tuning aid, not product code).
// Every kernel runs `iters` iterations of one block of 32 identical-shape instructions per wave and
// reports shader cycles (s_memtime) per wave-instruction at 1, 2 and 3 waves per SIMD (256-lane
// workgroups, one per CU slot), i.e. the issue cost the SIMD charges for it.  Mixed blocks answer
// whether LDS traffic and VALU work of one SIMD overlap or add up.
// usage: issue_cost [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define REP8(X) X X X X X X X X
#define REP4(X) X X X X

enum Kind {
    K_FMA, K_PKFMA, K_PKADD, K_PKMUL, K_ADD, K_MOV_DPP, K_PERMLANE32, K_BPERMUTE,
    K_DSW128, K_DSW64, K_DSW32, K_DSWADDTID, K_DSR128, K_DSR64, K_DSR32,
    K_MIX_PK_DSW128, K_MIX_PK_DSR128, K_MIX_PK_DSW64, K_MIX_FMA_DSW128, K_MIX_PK_DSWADDTID, K_PKFMA_HALF,
    K_MIX_PK_DSW32, K_DSW2ST64, K_DSR2ST64, K_GSTORE8, K_GSTORE16, K_GSTORE8_L2, K_GSTORE16_L2, K_MIX_FMA_GSTORE8_L2, K_FMA_AFTER_GSTORE8_L2, K_MIX_FMA_GSTORE16_L2, K_MIX_FMA_GSTORE8x4_L2, K_SALU, K_MIX_FMA_SALU, K_MIX_PK_SALU, K_WAITCNT, K_MIX_FMA_WAITCNT, K_COUNT
};
static const char *kNames[K_COUNT] = {
    "v_fma_f32 x32", "v_pk_fma_f32 x32", "v_pk_add_f32 x32", "v_pk_mul_f32 x32", "v_add_f32 x32", "v_mov_b32_dpp x32",
    "v_permlane32_swap x32", "ds_bpermute_b32 x32",
    "ds_write_b128 x32", "ds_write_b64 x32", "ds_write_b32 x32", "ds_write_addtid_b32 x32", "ds_read_b128 x32", "ds_read_b64 x32",
    "ds_read_b32 x32",
    "32 v_pk_fma + 8 ds_write_b128", "32 v_pk_fma + 8 ds_read_b128", "32 v_pk_fma + 16 ds_write_b64", "32 v_fma + 8 ds_write_b128",
    "32 v_pk_fma + 32 ds_write_addtid_b32", "v_pk_fma_f32 x16 (half block)", "32 v_pk_fma + 32 ds_write_b32",
    "ds_write2st64_b32 x32", "ds_read2st64_b32 x32", "global_store_dwordx2 x8 (8 KB/wave iter)", "global_store_dwordx4 x8",
    "global_store_dwordx2 x8, the same 4 KB per wave (L2)", "global_store_dwordx4 x8, the same 8 KB per wave (L2)",
    "8 global_store_dwordx2 (L2) + 32 v_fma interleaved", "8 global_store_dwordx2 (L2), then 32 v_fma",
    "4 global_store_dwordx4 (L2) + 32 v_fma interleaved", "4 global_store_dwordx2 (L2) + 32 v_fma interleaved",
    "s_add_u32 x32", "32 v_fma + 32 s_add interleaved", "32 v_pk_fma + 32 s_add interleaved", "s_waitcnt lgkmcnt(0) x32", "32 v_fma + 32 s_waitcnt interleaved"};

template <int KIND> __global__ __launch_bounds__(256) void k(int iters, unsigned long long *cyc, float *sink, float4 *gout)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x;
    v2f a0 = {1.f + t, 2.f}, a1 = {3.f, 4.f}, a2 = {5.f, 6.f}, a3 = {7.f, 8.f}, a4 = {1.5f, 2.5f}, a5 = {3.5f, 4.5f},
        a6 = {5.5f, 6.5f}, a7 = {7.5f, 8.5f};
    const v2f m = {0.999f, 1.001f}, c = {0.001f, -0.001f};
    v4f w = {1.f, 2.f, 3.f, 4.f};
    // every lane its own 16-byte slot, conflict-free, wave-private region
    const unsigned addr = (unsigned)(uintptr_t)smem + (unsigned)t * 16u;
    const unsigned addr8 = (unsigned)(uintptr_t)smem + (unsigned)t * 8u;
    const unsigned addr4 = (unsigned)(uintptr_t)smem + (unsigned)t * 4u;
    const unsigned bperm = (unsigned)((t * 4 + 4) & 255);
    float4 *gp = gout + ((size_t)blockIdx.x * 256 + t);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#define PK8 \
        "v_pk_fma_f32 %0, %0, %8, %9\n\tv_pk_fma_f32 %1, %1, %8, %9\n\tv_pk_fma_f32 %2, %2, %8, %9\n\tv_pk_fma_f32 %3, %3, %8, %9\n\t" \
        "v_pk_fma_f32 %4, %4, %8, %9\n\tv_pk_fma_f32 %5, %5, %8, %9\n\tv_pk_fma_f32 %6, %6, %8, %9\n\tv_pk_fma_f32 %7, %7, %8, %9\n\t"
#define OPS8(OP) \
        OP " %0, %0, %8, %9\n\t" OP " %1, %1, %8, %9\n\t" OP " %2, %2, %8, %9\n\t" OP " %3, %3, %8, %9\n\t" \
        OP " %4, %4, %8, %9\n\t" OP " %5, %5, %8, %9\n\t" OP " %6, %6, %8, %9\n\t" OP " %7, %7, %8, %9\n\t"
#define OPS8_2(OP) \
        OP " %0, %0, %8\n\t" OP " %1, %1, %8\n\t" OP " %2, %2, %8\n\t" OP " %3, %3, %8\n\t" \
        OP " %4, %4, %8\n\t" OP " %5, %5, %8\n\t" OP " %6, %6, %8\n\t" OP " %7, %7, %8\n\t"
#define REGS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c)
        if (KIND == K_PKFMA) asm volatile(REP4(PK8) REGS);
        if (KIND == K_PKFMA_HALF) asm volatile(PK8 PK8 REGS);
        if (KIND == K_PKADD) asm volatile(REP4(OPS8_2("v_pk_add_f32")) REGS);
        if (KIND == K_PKMUL) asm volatile(REP4(OPS8_2("v_pk_mul_f32")) REGS);
#define F8 \
        "v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t" \
        "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9\n\t"
#define FREGS : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(m.x), "v"(c.x)
        if (KIND == K_FMA) asm volatile(REP4(F8) FREGS);
        if (KIND == K_ADD) asm volatile(REP4(OPS8_2("v_add_f32")) FREGS);
        if (KIND == K_MOV_DPP)
            asm volatile(REP4("v_mov_b32_dpp %0, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %2 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                              "v_mov_b32_dpp %2, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %4 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                              "v_mov_b32_dpp %4, %5 row_ror:4 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %5, %6 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                              "v_mov_b32_dpp %6, %7 row_ror:4 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %7, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t") FREGS);
        if (KIND == K_PERMLANE32)
            asm volatile(REP8("v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\t"
                              "v_permlane32_swap_b32 %6, %7\n\t") FREGS);
        if (KIND == K_BPERMUTE)
            asm volatile(REP4("ds_bpermute_b32 %0, %10, %0\n\tds_bpermute_b32 %1, %10, %1\n\tds_bpermute_b32 %2, %10, %2\n\t"
                              "ds_bpermute_b32 %3, %10, %3\n\tds_bpermute_b32 %4, %10, %4\n\tds_bpermute_b32 %5, %10, %5\n\t"
                              "ds_bpermute_b32 %6, %10, %6\n\tds_bpermute_b32 %7, %10, %7\n\t") "s_waitcnt lgkmcnt(0)"
                         : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x)
                         : "v"(m.x), "v"(c.x), "v"(bperm));
#define W128x8 \
        "ds_write_b128 %10, %11\n\tds_write_b128 %10, %11 offset:4096\n\tds_write_b128 %10, %11 offset:8192\n\tds_write_b128 %10, %11 offset:12288\n\t" \
        "ds_write_b128 %10, %11 offset:16384\n\tds_write_b128 %10, %11 offset:20480\n\tds_write_b128 %10, %11 offset:24576\n\tds_write_b128 %10, %11 offset:28672\n\t"
#define W64x8 \
        "ds_write_b64 %10, %11\n\tds_write_b64 %10, %11 offset:2048\n\tds_write_b64 %10, %11 offset:4096\n\tds_write_b64 %10, %11 offset:6144\n\t" \
        "ds_write_b64 %10, %11 offset:8192\n\tds_write_b64 %10, %11 offset:10240\n\tds_write_b64 %10, %11 offset:12288\n\tds_write_b64 %10, %11 offset:14336\n\t"
#define W32x8 \
        "ds_write_b32 %10, %11\n\tds_write_b32 %10, %11 offset:1024\n\tds_write_b32 %10, %11 offset:2048\n\tds_write_b32 %10, %11 offset:3072\n\t" \
        "ds_write_b32 %10, %11 offset:4096\n\tds_write_b32 %10, %11 offset:5120\n\tds_write_b32 %10, %11 offset:6144\n\tds_write_b32 %10, %11 offset:7168\n\t"
#define WTID8 \
        "ds_write_addtid_b32 %11\n\tds_write_addtid_b32 %11 offset:1024\n\tds_write_addtid_b32 %11 offset:2048\n\tds_write_addtid_b32 %11 offset:3072\n\t" \
        "ds_write_addtid_b32 %11 offset:4096\n\tds_write_addtid_b32 %11 offset:5120\n\tds_write_addtid_b32 %11 offset:6144\n\tds_write_addtid_b32 %11 offset:7168\n\t"
#define MREGS(A, D) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c), "v"(A), "v"(D) : "memory"
        if (KIND == K_DSW128) asm volatile(REP4(W128x8) "s_waitcnt lgkmcnt(0)" MREGS(addr, w));
        if (KIND == K_DSW64) asm volatile(REP4(W64x8) "s_waitcnt lgkmcnt(0)" MREGS(addr8, m));
        if (KIND == K_DSW32) asm volatile(REP4(W32x8) "s_waitcnt lgkmcnt(0)" MREGS(addr4, m.x));
        if (KIND == K_DSWADDTID)
            asm volatile("s_mov_b32 m0, 0\n\t" REP4(WTID8) "s_waitcnt lgkmcnt(0)" MREGS(addr4, m.x));
        if (KIND == K_DSW2ST64)
            asm volatile(REP8("ds_write2st64_b32 %10, %11, %12 offset0:0 offset1:16\n\tds_write2st64_b32 %10, %11, %12 offset0:4 offset1:20\n\t"
                              "ds_write2st64_b32 %10, %11, %12 offset0:8 offset1:24\n\tds_write2st64_b32 %10, %11, %12 offset0:12 offset1:28\n\t")
                         "s_waitcnt lgkmcnt(0)"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(m), "v"(c), "v"(addr4), "v"(m.x), "v"(m.y) : "memory");
        if (KIND == K_DSR2ST64) {
            v2f r0, r1, r2, r3;
            asm volatile(REP8("ds_read2st64_b32 %0, %4 offset0:0 offset1:16\n\tds_read2st64_b32 %1, %4 offset0:4 offset1:20\n\t"
                              "ds_read2st64_b32 %2, %4 offset0:8 offset1:24\n\tds_read2st64_b32 %3, %4 offset0:12 offset1:28\n\t")
                         "s_waitcnt lgkmcnt(0)" : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(addr4) : "memory");
            a0 += r0 + r1 + r2 + r3;
        }
        if (KIND == K_DSR128) {
            v4f r0, r1, r2, r3;
            asm volatile(REP8("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:4096\n\tds_read_b128 %2, %4 offset:8192\n\t"
                              "ds_read_b128 %3, %4 offset:12288\n\t") "s_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(addr) : "memory");
            a0.x += r0.x + r1.x + r2.x + r3.x;
        }
        if (KIND == K_DSR64) {
            v2f r0, r1, r2, r3;
            asm volatile(REP8("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:2048\n\tds_read_b64 %2, %4 offset:4096\n\t"
                              "ds_read_b64 %3, %4 offset:6144\n\t") "s_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(addr8) : "memory");
            a0 += r0 + r1 + r2 + r3;
        }
        if (KIND == K_DSR32) {
            float r0, r1, r2, r3;
            asm volatile(REP8("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:1024\n\tds_read_b32 %2, %4 offset:2048\n\t"
                              "ds_read_b32 %3, %4 offset:3072\n\t") "s_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(addr4) : "memory");
            a0.x += r0 + r1 + r2 + r3;
        }
        // mixed: the LDS operations first (in flight), the VALU block behind them, one wait at the end
        if (KIND == K_MIX_PK_DSW128) asm volatile(W128x8 REP4(PK8) "s_waitcnt lgkmcnt(0)" MREGS(addr, w));
        if (KIND == K_MIX_FMA_DSW128)
            asm volatile(W128x8 REP4(F8) "s_waitcnt lgkmcnt(0)"
                         : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x)
                         : "v"(m.x), "v"(c.x), "v"(addr), "v"(w) : "memory");
        if (KIND == K_MIX_PK_DSW64) asm volatile(W64x8 W64x8 REP4(PK8) "s_waitcnt lgkmcnt(0)" MREGS(addr8, m));
        if (KIND == K_MIX_PK_DSW32) asm volatile(REP4(W32x8) REP4(PK8) "s_waitcnt lgkmcnt(0)" MREGS(addr4, m.x));
        if (KIND == K_MIX_PK_DSWADDTID)
            asm volatile("s_mov_b32 m0, 0\n\t" REP4(WTID8) REP4(PK8) "s_waitcnt lgkmcnt(0)" MREGS(addr4, m.x));
        if (KIND == K_MIX_PK_DSR128) {
            v4f r0, r1, r2, r3, r4, r5, r6, r7;
            asm volatile("ds_read_b128 %8, %18\n\tds_read_b128 %9, %18 offset:4096\n\tds_read_b128 %10, %18 offset:8192\n\t"
                         "ds_read_b128 %11, %18 offset:12288\n\tds_read_b128 %12, %18 offset:16384\n\tds_read_b128 %13, %18 offset:20480\n\t"
                         "ds_read_b128 %14, %18 offset:24576\n\tds_read_b128 %15, %18 offset:28672\n\t"
                         "v_pk_fma_f32 %0, %0, %16, %17\n\tv_pk_fma_f32 %1, %1, %16, %17\n\tv_pk_fma_f32 %2, %2, %16, %17\n\tv_pk_fma_f32 %3, %3, %16, %17\n\t"
                         "v_pk_fma_f32 %4, %4, %16, %17\n\tv_pk_fma_f32 %5, %5, %16, %17\n\tv_pk_fma_f32 %6, %6, %16, %17\n\tv_pk_fma_f32 %7, %7, %16, %17\n\t"
                         "v_pk_fma_f32 %0, %0, %16, %17\n\tv_pk_fma_f32 %1, %1, %16, %17\n\tv_pk_fma_f32 %2, %2, %16, %17\n\tv_pk_fma_f32 %3, %3, %16, %17\n\t"
                         "v_pk_fma_f32 %4, %4, %16, %17\n\tv_pk_fma_f32 %5, %5, %16, %17\n\tv_pk_fma_f32 %6, %6, %16, %17\n\tv_pk_fma_f32 %7, %7, %16, %17\n\t"
                         "v_pk_fma_f32 %0, %0, %16, %17\n\tv_pk_fma_f32 %1, %1, %16, %17\n\tv_pk_fma_f32 %2, %2, %16, %17\n\tv_pk_fma_f32 %3, %3, %16, %17\n\t"
                         "v_pk_fma_f32 %4, %4, %16, %17\n\tv_pk_fma_f32 %5, %5, %16, %17\n\tv_pk_fma_f32 %6, %6, %16, %17\n\tv_pk_fma_f32 %7, %7, %16, %17\n\t"
                         "v_pk_fma_f32 %0, %0, %16, %17\n\tv_pk_fma_f32 %1, %1, %16, %17\n\tv_pk_fma_f32 %2, %2, %16, %17\n\tv_pk_fma_f32 %3, %3, %16, %17\n\t"
                         "v_pk_fma_f32 %4, %4, %16, %17\n\tv_pk_fma_f32 %5, %5, %16, %17\n\tv_pk_fma_f32 %6, %6, %16, %17\n\tv_pk_fma_f32 %7, %7, %16, %17\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(r0), "=&v"(r1),
                           "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
                         : "v"(m), "v"(c), "v"(addr) : "memory");
            a0.x += r0.x + r1.x + r2.x + r3.x + r4.x + r5.x + r6.x + r7.x;
        }
#define SA8 "s_add_u32 s20, s20, 1\n\ts_add_u32 s21, s21, 1\n\ts_add_u32 s22, s22, 1\n\ts_add_u32 s23, s23, 1\n\ts_add_u32 s24, s24, 1\n\ts_add_u32 s25, s25, 1\n\ts_add_u32 s26, s26, 1\n\ts_add_u32 s27, s27, 1\n\t"
#define SCLOB : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc"
        if (KIND == K_SALU) asm volatile(REP4(SA8) ::SCLOB);
#define FS(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n\ts_add_u32 s2" #i ", s2" #i ", 1\n\t"
#define PS(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n\ts_add_u32 s2" #i ", s2" #i ", 1\n\t"
#define FW(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n\ts_waitcnt lgkmcnt(0)\n\t"
        if (KIND == K_MIX_FMA_SALU)
            asm volatile(REP4(FS(0) FS(1) FS(2) FS(3) FS(4) FS(5) FS(6) FS(7))
                         : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(m.x), "v"(c.x) SCLOB);
        if (KIND == K_MIX_PK_SALU)
            asm volatile(REP4(PS(0) PS(1) PS(2) PS(3) PS(4) PS(5) PS(6) PS(7))
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c) SCLOB);
        if (KIND == K_WAITCNT) asm volatile(REP4(REP8("s_waitcnt lgkmcnt(0)\n\t")));
        if (KIND == K_MIX_FMA_WAITCNT)
            asm volatile(REP4(FW(0) FW(1) FW(2) FW(3) FW(4) FW(5) FW(6) FW(7))
                         : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(m.x), "v"(c.x));
        if (KIND == K_GSTORE8) {
            float2 *g2 = reinterpret_cast<float2 *>(gp);
#pragma unroll
            for (int j = 0; j < 8; ++j) g2[(size_t)j * 65536 * 16 + (size_t)(i & 15) * 65536] = make_float2(a0.x, a1.x);
        }
        // stores that stay in L2 (every wave rewrites its own lines): the cost of ISSUING a store, not of HBM
        if (KIND == K_GSTORE8_L2) {
            float2 *g2 = reinterpret_cast<float2 *>(gout) + ((size_t)blockIdx.x * 256 + t);
#pragma unroll
            for (int j = 0; j < 8; ++j) g2[(size_t)j * 65536 * 4] = make_float2(a0.x, a1.x);
        }
        if (KIND == K_GSTORE16_L2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) gp[(size_t)j * 65536 * 4] = make_float4(a0.x, a1.x, a2.x, a3.x);
        }
        if (KIND == K_MIX_FMA_GSTORE8_L2 || KIND == K_FMA_AFTER_GSTORE8_L2) {
            float2 *g2 = reinterpret_cast<float2 *>(gout) + ((size_t)blockIdx.x * 256 + t);
            if (KIND == K_FMA_AFTER_GSTORE8_L2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) g2[(size_t)j * 65536 * 4] = make_float2(a0.x, a1.x);
                asm volatile(REP4(F8) FREGS);
            } else {
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    g2[(size_t)j * 65536 * 4] = make_float2(a0.x, a1.x);
                    g2[(size_t)(j + 1) * 65536 * 4] = make_float2(a0.x, a1.x);
                    asm volatile(F8 FREGS);
                }
            }
        }
        if (KIND == K_MIX_FMA_GSTORE16_L2) {
            float4 *g4 = gout + ((size_t)blockIdx.x * 256 + t);
#pragma unroll
            for (int j = 0; j < 4; ++j) {           // (four stores of 1 KB: the bytes of eight dwordx2 stores)
                g4[(size_t)j * 65536 * 2] = make_float4(a0.x, a1.x, a2.x, a3.x);
                asm volatile(F8 FREGS);
            }
        }
        if (KIND == K_MIX_FMA_GSTORE8x4_L2) {
            float2 *g2 = reinterpret_cast<float2 *>(gout) + ((size_t)blockIdx.x * 256 + t);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                g2[(size_t)j * 65536 * 4] = make_float2(a0.x, a1.x);
                asm volatile(F8 FREGS);
            }
        }
        if (KIND == K_GSTORE16) {
#pragma unroll
            for (int j = 0; j < 8; ++j) gp[(size_t)j * 65536 * 16 + (size_t)(i & 15) * 65536] = make_float4(a0.x, a1.x, a2.x, a3.x);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (t == 0) cyc[blockIdx.x] = t1 - t0;
    if (a0.x + a1.x + a2.x + a3.x + a4.x + a5.x + a6.x + a7.x + a0.y + a1.y + a2.y + a3.y == 12345.678f) sink[0] = 1.f;
}

template <int KIND> void run(int iters, unsigned long long *dcyc, float *dsink, float4 *gout)
{
    // 256 CUs x occ workgroups of 4 waves (one per SIMD): occ = waves per SIMD.  LDS sized so that exactly occ fit.
    for (int occ = 1; occ <= 3; ++occ) {
        const size_t lds = occ == 1 ? 96 * 1024 : (occ == 2 ? 64 * 1024 : 36 * 1024);
        const int blocks = 256 * occ;
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), lds, 0, iters / 4 + 1, dcyc, dsink, gout);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), lds, 0, iters, dcyc, dsink, gout);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(blocks);
        CK(hipMemcpy(h.data(), dcyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double avg = 0;
        for (auto v : h) avg += (double)v;
        avg /= blocks;
        // cycles the SIMD spends per iteration of ONE wave's block (all waves on the SIMD progress together)
        printf("  %-42s occ %d: %8.1f counter ticks per iteration per wave, %8.1f per SIMD-iteration (wall %.3f ms)\n",
               kNames[KIND], occ, avg / iters, avg / iters / occ, ms);
    }
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    unsigned long long *dcyc;
    float *dsink;
    float4 *gout;
    CK(hipMalloc(&dcyc, 4096 * sizeof(unsigned long long)));
    CK(hipMalloc(&dsink, 16));
    CK(hipMalloc(&gout, (size_t)65536 * 16 * 8 * sizeof(float4) + (1 << 20)));
    int clk = 0;
    CK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
    printf("device clock attribute %d kHz; counter = s_memrealtime/s_memtime via __builtin_readcyclecounter\n", clk);
    run<K_FMA>(iters, dcyc, dsink, gout);
    run<K_ADD>(iters, dcyc, dsink, gout);
    run<K_PKFMA>(iters, dcyc, dsink, gout);
    run<K_PKFMA_HALF>(iters, dcyc, dsink, gout);
    run<K_PKADD>(iters, dcyc, dsink, gout);
    run<K_PKMUL>(iters, dcyc, dsink, gout);
    run<K_MOV_DPP>(iters, dcyc, dsink, gout);
    run<K_PERMLANE32>(iters, dcyc, dsink, gout);
    run<K_BPERMUTE>(iters, dcyc, dsink, gout);
    run<K_DSW128>(iters, dcyc, dsink, gout);
    run<K_DSW64>(iters, dcyc, dsink, gout);
    run<K_DSW32>(iters, dcyc, dsink, gout);
    run<K_DSWADDTID>(iters, dcyc, dsink, gout);
    run<K_DSW2ST64>(iters, dcyc, dsink, gout);
    run<K_DSR128>(iters, dcyc, dsink, gout);
    run<K_DSR64>(iters, dcyc, dsink, gout);
    run<K_DSR32>(iters, dcyc, dsink, gout);
    run<K_DSR2ST64>(iters, dcyc, dsink, gout);
    run<K_MIX_PK_DSW128>(iters, dcyc, dsink, gout);
    run<K_MIX_FMA_DSW128>(iters, dcyc, dsink, gout);
    run<K_MIX_PK_DSW64>(iters, dcyc, dsink, gout);
    run<K_MIX_PK_DSW32>(iters, dcyc, dsink, gout);
    run<K_MIX_PK_DSWADDTID>(iters, dcyc, dsink, gout);
    run<K_MIX_PK_DSR128>(iters, dcyc, dsink, gout);
    run<K_SALU>(iters, dcyc, dsink, gout);
    run<K_MIX_FMA_SALU>(iters, dcyc, dsink, gout);
    run<K_MIX_PK_SALU>(iters, dcyc, dsink, gout);
    run<K_WAITCNT>(iters, dcyc, dsink, gout);
    run<K_MIX_FMA_WAITCNT>(iters, dcyc, dsink, gout);
    run<K_GSTORE8>(iters / 4, dcyc, dsink, gout);
    run<K_GSTORE16>(iters / 4, dcyc, dsink, gout);
    run<K_GSTORE8_L2>(iters, dcyc, dsink, gout);
    run<K_GSTORE16_L2>(iters, dcyc, dsink, gout);
    run<K_MIX_FMA_GSTORE8_L2>(iters, dcyc, dsink, gout);
    run<K_FMA_AFTER_GSTORE8_L2>(iters, dcyc, dsink, gout);
    run<K_MIX_FMA_GSTORE16_L2>(iters, dcyc, dsink, gout);
    run<K_MIX_FMA_GSTORE8x4_L2>(iters, dcyc, dsink, gout);
    return 0;
}
