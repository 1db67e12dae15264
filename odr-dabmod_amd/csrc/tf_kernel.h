// tf_kernel.h -- the frame kernel of the DAB COFDM hot path and its launcher template.
//
// tf_kernel: ONE launch takes the coded bits of a batch
// of transmission frames and produces the finished I/Q stream, i.e. the
// reference's QpskSymbolMapper -> FrequencyInterleaver -> DifferentialModulator
// -> SignalMultiplexer -> OfdmGenerator -> GainControl -> GuardIntervalInserter
// -> FIRFilter sub-graph (src/DabModulator.cpp:385-419) with no intermediate in
// HBM.  A workgroup owns a run of consecutive OFDM symbols of one frame:
//   * the differential-modulation state lives in registers as integer phases (six 4-bit fields of one
//     register per lane); each lane's six carriers are exactly its inputs of the first FFT stage, so
//     nothing is scattered through LDS;
//   * the N-point backward FFT is a Stockham radix-8 autosort (8 . 8 . 8 . 4 for N = 2048), 8 points
//     per lane, three exchanges through one padded LDS buffer (row layout for the first, additive
//     padding for the stride-8 one; every access is base + immediate), twiddles in registers / a small
//     LDS table;
//   * gain statistics come from the SPECTRUM (population variance through Parseval on carrier pairs),
//     counted with ballots; modes max / fix reduce over the FFT output with DPP;
//   * the cyclic prefix is a second store of the same registers;
//   * the FIR is spectral: inside a symbol it is the factor H[k] on the carriers.  Mode I coded-bits chain with the
//     45-tap filter (EQ): ONE transform per symbol, of X H; the 44 outputs per symbol boundary that the cyclic
//     filtering gets wrong are corrected from the filtered symbols alone through a host-designed inverse of the
//     taps on the occupied carriers.  Every other FIR variant: the unfiltered and the filtered IFFT of a symbol as
//     ONE packed dual transform (struct c2), the unfiltered half pruned to the 88 samples the boundary FIR reads
//     (Fft::run_dual_zonly), those boundary outputs a direct FIR;
//   * OFDM windowing (WIN): the raised-cosine seams between symbols through LDS; with FIR as well, the windowed stream
//     around every seam is built in LDS and the outputs that look into it are a direct FIR (packed dual transform) -- except
//     on the EQ chain with overlaps up to kEqWinMax, where the seam rides on the equalised boundary outputs (WIN && EQ).
// HBM traffic is therefore the compulsory 28.8 kB in + 1.57 MB out per frame.
//
// No MFMA (no dense contraction in this path), wave64 throughout.
#pragma once
#include <cstdio>
#include "device_common.h"
#include "tf_layout.h"

namespace dabgpu {
namespace {

// ---------------------------------------------------------------------------
// cos/sin of p*45deg as {-1,0,+1} codes: (CX >> 2p) & 3 = value + 1
constexpr unsigned kCX = 0x901Au;

// FIR inside the fused kernel ("spectral FIR").
// The stream is a chain of cyclically extended symbols, and the FIR looks AHEAD
// (out[n] = sum_j taps[j] in[n+j]), so every output whose ntaps-1 look-ahead
// samples stay inside its own segment is a CIRCULAR convolution of the symbol:
//     out[p] = g_s * IDFT_N( X_s[k] * H[k] )[(p - cp) mod N],  H[k] = sum_j taps[j] e^{+2 pi i jk/N}
// i.e. a second IFFT of the same carriers under a per-bin factor.  Only the last
// C = ntaps-1 samples of a segment see the next symbol; those 44 outputs are
// computed directly from the C-sample tail of this symbol and the C-sample head
// of the next one (unfiltered, gain applied), kept in LDS.  Cost per symbol:
// 2 FFTs + C*ntaps MACs instead of 1 FFT + N*ntaps MACs.

// The transmission-mode geometry is a function of the FFT size (reference
// src/DabModulator.cpp:84-122), so it is compile-time here; NT is the number of FIR taps
// when known at compile time (the default 45-tap filter) or 0 for "read it from the args".
template <int LOGN> struct ModeGeom;
template <> struct ModeGeom<11> { static constexpr int nb_symbols = 76, K = 1536, null_size = 2656, sym_size = 2552; };
template <> struct ModeGeom<9> { static constexpr int nb_symbols = 76, K = 384, null_size = 664, sym_size = 638; };
template <> struct ModeGeom<8> { static constexpr int nb_symbols = 153, K = 192, null_size = 345, sym_size = 319; };
template <> struct ModeGeom<10> { static constexpr int nb_symbols = 76, K = 768, null_size = 1328, sym_size = 1276; };

// CFR (f-3; either with the whole fused epilogue GUARD + FIR or with neither: then the chain continues with the stand-alone guard and FIR
// kernels): crest-factor reduction of every symbol right after its IFFT, in registers -- clip, forward
// FFT, error clip against the lane's own input bins, IFFT again -- plus the reference's statistics.
// GVAR (carriers path with GAIN only): the gain mode is known to be "var" -- the statistics come from
// the spectrum and the time-domain reduction (which keeps both transforms of a symbol live and costs
// the third workgroup per CU) is compiled out.
// ZONLY (Mode I with the fused FIR and no gain statistics over the time domain: the coded-bits path with gain fix / var,
// the carriers path with gain var or none): the unfiltered transform is formed only where
// the boundary FIR reads it (Fft::run_dual_zonly).
// OFMT = 1: the output is s16 (4 bytes per sample, FormatConverter semantics) instead of cf32; OFMT = 2 / 3 (round 5): u8 / s8, 2
// bytes per sample, on the equalised-boundary and the no-FIRFilter variants -- instantiated for the
// production variants only (Mode I coded-bits chain, default filter); everything else converts in format_kernel.
// WIN (coded-bits chain with guard interval, no FIR): the guard interval is windowed (ofdmwindowing > 0, f-4,
// src/GuardIntervalInserter.cpp:149-300).  Every sample outside the 2W-wide seams is the copy it is without a window;
// seam sample j between symbols s-1 and s is  prev[j] * w[2W-1-j] + rise[j] * w[j], with prev = the last W samples of
// symbol s-1 followed by its first W (the suffix written past its end) and rise = samples [N-cp-W, N-cp+W) of symbol
// s.  The 2W + 2W samples go through LDS; the seam before a run's first symbol is written by the run before it,
// which transforms that symbol too (look-ahead, as with the FIR).
// EQ (Mode I coded-bits chain with the 45-tap FIR, gain none / fix / var -- the cfg 3 chain): ONE transform per symbol,
// of the FILTERED spectrum X H, instead of the packed (unfiltered, filtered) pair.  The 44 outputs between two symbols
// that the cyclic filtering gets wrong are corrected from the filtered symbols alone:
//     y[N-44+i] = z_prev[N-44+i] + sum_{j >= 44-i} taps[j] d[i+j-44],   d[m] = x_cur[N-cp+m] - x_prev[m]
// (the cyclic result looked into x_prev's own start where the stream continues with x_cur's prefix), and the
// unfiltered difference d comes out of a short inverse filter g of the taps (G H = 1 on the occupied bins -- the only
// ones a symbol has energy in; designed on the host, api_context.hip design_inverse_filter):
//     d[m] = sum_j g[j] w[m - (j - c)],   w[q] = z_cur[N-cp+q] - z_prev[q mod N],   q in [-103, 99].
// 44 x 160 + 990 real-by-complex multiply-adds per symbol replace half of a packed 2048-point transform, its 16-byte
// exchanges and the pack / unpack around it.
// EQ with WIN (round 5; overlap W <= kEqWinMax = 10): around a seam the windowed stream is x_prev + omega d with omega the rising
// raised-cosine factor (0 before the seam, 1 behind it; the reference's two factors of a seam sample add up to one), so the
// same correction serves with d taken on [-W, W + 44) and weighted: 2W + 44 <= 64 outputs per seam, (2W + 44) x 160 for d
// (all 256 lanes where the plain form keeps 176 busy) and up to 45 terms per output.  TII and the integer formats as in the
// plain form.  Measured: 2.51 M frames/s against 1.91 M for the packed dual transform that served these settings before.
template <int LOGN, bool FROM_BITS, bool GAIN, bool GUARD, bool FIR, int NT, bool CFR = false, bool GVAR = false,
          bool ZONLY = false, int OFMT = 0, bool WIN = false, bool EQ = false>
// (waves per SIMD, buffer scheme: tf_layout.h, tf_variant -- the one table the host's LDS size reads too)
__global__ __launch_bounds__((1 << LOGN) / 8 < 64 ? 64 : (1 << LOGN) / 8,
                             tf_variant(LOGN, FROM_BITS, GAIN, GUARD, FIR, NT, CFR, GVAR, OFMT, WIN, EQ).waves_per_simd)
void tf_kernel(const TfArgs a)
{
    static_assert(!GVAR || (GAIN && !FROM_BITS && !CFR), "GVAR is a specialisation of the carriers path with gain");
    static_assert(!CFR || GUARD == FIR || (FROM_BITS && GUARD),
                  "CFR variants: the full fused epilogue or none of it; the coded-bits chain also with the guard interval alone");
    static_assert(!ZONLY || (LOGN == 11 && GUARD && FIR && NT > 0 && !CFR),
                  "ZONLY: the dual transform of the Mode I chain with the fused FIR");
    static_assert(!ZONLY || FROM_BITS || GVAR || !GAIN, "ZONLY: no gain statistics over the time domain");
    static_assert(!WIN || (FROM_BITS && GUARD && (OFMT == 0 || EQ || (OFMT == 1 && !FIR && !CFR))),
                  "WIN: coded-bits chain with guard interval (integer store: every format with EQ, s16 without FIRFilter)");
    static_assert(!(WIN && FIR) || (!ZONLY && (NT == 0 || (NT == 45 && !CFR)) && !GVAR),
                  "WIN with FIR: the generic packed dual transform (all unfiltered samples at hand), run-time tap count -- or EQ");
    static_assert(!EQ || (FROM_BITS && GUARD && FIR && NT == 45 && !CFR && !GVAR && !ZONLY),
                  "EQ: the coded-bits chain with the 45-tap filter");
    static_assert(!EQ || LOGN == 11 || (LOGN == 10 && !WIN && OFMT == 0),
                  "EQ in transmission mode IV (round 6): complexf output, no windowing (Mode I: WIN with overlap <= kEqWinMax, every format)");
    typedef ModeGeom<LOGN> G;
    typedef Fft<LOGN> F;
    constexpr int N = F::N, T = F::T;
    // exchange buffers: the variants without FIR alternate between two (one barrier per exchange); the FIR variants
    // keep one (two barriers per exchange) -- 36 KB of LDS per workgroup and three workgroups per CU, EQ 29 KB and four
    // CFR_LEAN (round 5: the Mode I coded-bits CFR chains with the guard interval -- no FIRFilter, or the default-length one): built
    // for FOUR waves per SIMD.  What kept these kernels above 128 registers were loop invariants, not the transforms; each went
    // where it costs an instruction or two per symbol instead of a register (see advance, fetch_block, cfr_symbol, boundary), and
    // one exchange buffer serves (two barriers per exchange; 24 kB of LDS per workgroup).
    // NOFIR_1BUF (round 5: the reference's default chain, Mode I from coded bits): ONE exchange buffer (two barriers per exchange)
    // and 24 kB of LDS instead of 42 -- five workgroups per CU (<= 96 registers) instead of three.  Measured: complexf output
    // unchanged (3.3 M frames/s: the board's power limit), s16 4.07 -> 4.29 M, u8 3.86 -> 4.13 M; four waves: 4.12 / 3.96 M.
    // (the s8 store without GainControl spills 8 bytes at five waves: that one is built for four)
    // (windowed: with the s16 store, 3.57 -> 3.82 M; the complexf form loses 2 % and keeps two buffers)
    constexpr TfVariant VAR = tf_variant(LOGN, FROM_BITS, GAIN, GUARD, FIR, NT, CFR, GVAR, OFMT, WIN, EQ);
    constexpr bool CFR_LEAN = VAR.cfr_lean, NOFIR_1BUF = VAR.nofir_1buf, DBUF = VAR.dbuf;
    (void)NOFIR_1BUF;
    // HALVES (round 6; Mode III, the plain coded-bits chain with the default-length filter): a 256-point symbol is 32 lanes of
    // eight points, half a wave -- so the two halves of the workgroup's one wave work on TWO FRAMES (2 p and 2 p + 1, the same run
    // of symbols in each): every lane quantity is the frame's own as it is, every LDS buffer that belongs to a frame exists
    // twice, and the few wave-uniform quantities that differ between the frames (the frame's address, the gain statistic, the
    // store offset) become lane quantities.  `t` below is the lane's index INSIDE ITS FRAME (0 .. 31), t_wg the lane of the
    // workgroup.  Every other variant: one frame per workgroup, t == t_wg, NH == 1 -- the same instructions as before.
    constexpr bool HALVES = VAR.halves;
    constexpr int NH = HALVES ? 2 : 1;
    const int t_wg = threadIdx.x;
    const int half = HALVES ? (t_wg >> 5) : 0;
    const int t = HALVES ? (t_wg & 31) : t_wg;
    const bool lane_on = (HALVES || T >= 64) ? true : t < T;  // only N=256 (T=32) without HALVES runs with idle lanes (the block is max(T, 64) lanes)
    const unsigned long long on_mask = (HALVES || T >= 64) ? ~0ull : ((1ull << (T & 63)) - 1ull);   // the same as a wave mask
    const int tt = lane_on ? t : 0;
    auto hoff = [&](int n) __attribute__((always_inline)) -> int { return HALVES ? half * n : 0; };   // the frame's copy of a buffer

    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf *fbuf = reinterpret_cast<cf *>(smem);                            // 2 x (N + N/8) complex
    int fpar = 0;                                                       // which half the next exchange uses
    // packed dual transforms (FIR variants other than EQ) exchange 16-byte elements
    // CFR_SEQ (Mode I coded-bits chain, CFR with the fused FIRFilter): the corrected spectrum's two inverse transforms run one
    // after the other as plain transforms instead of one packed pair -- the same instruction count (a packed fp32 instruction
    // occupies the SIMD twice as long), 8-byte exchanges, and registers for a fourth wave per SIMD
    // (the default filter length only: the run-time tap count's boundary loop does not fit the 128 registers)
    constexpr bool CFR_SEQ = VAR.cfr_seq;
    constexpr int kXElems = VAR.dual ? 2 * F::LDS_ELEMS : (DBUF ? 2 : 1) * F::LDS_ELEMS;
    double *red = reinterpret_cast<double *>(fbuf + NH * kXElems);  // 16 doubles
    fbuf += hoff(kXElems);
    // FIR boundary samples: two buffers [tail of symbol s (C) | head of symbol s+1 (C)], contiguous so
    // that the boundary outputs read in[i + j] without a tail/head case split
    // frequency-domain gain statistics (coded-bits path): one packed word of phases per lane
    uint32_t *phw = reinterpret_cast<uint32_t *>(red + NH * 16);          // [T]
    red += hoff(16);
    // (carriers path: three complex bins per lane instead -- the general form of the same statistic)
    cf *bnd = reinterpret_cast<cf *>(phw + NH * (GAIN ? (FROM_BITS ? T : 6 * T) : 0));
    phw += hoff(T);
    // coded bits of one OFDM symbol (K/4 bytes), double buffered, behind the FIR buffers
    constexpr int KB = NT ? NT - 1 : kBnd;      // slots per half buffer: the look-ahead C when it is a compile-time constant
    // WIN: two seam buffers [last W | first W samples of a symbol], the rising 2W samples of the next one, the window
    cf *wbuf = bnd;
    float *win_l = reinterpret_cast<float *>(wbuf + 6 * kWinMax);     // (WIN with FIR: moved behind the other tables below)
    // EQ: two windows of the previous filtered symbol (index q + kEqQL, q in [-kEqQL, kEqQH]), the difference w, the
    // 44 unfiltered differences d, the inverse filter
    // (kEqW: the tail past kEqQL + kEqQH stays zero)
    // EQ with WIN (overlap W <= kEqWinMax): the seam of 2W samples widens everything by W on either side -- differences d[m] for
    // m in [-W, W + 44), windows q in [-kEqQL, kEqQH] with kEqWm more on both ends, 2W + 44 <= 64 boundary outputs
    constexpr int kEqWm = (EQ && WIN) ? kEqWinMax : 0;
    constexpr int kEqQL = kEqTaps - 1 - kEqCentre + kEqWm, kEqQH = 43 + kEqCentre + kEqWm, kEqW = kEqWLen;
    static_assert(kEqQL + kEqQH + 1 + kEqWm <= kEqW && kEqTaps == 160, "EQ window (+ what the unused outputs of a narrower overlap read)");
    static_assert(kEqElems == 3 * kEqW + kEqDLen + (kEqTaps + 8) / 2 + 16 && 2 * kEqWinMax <= 32 && 2 * kEqWinMax + 44 <= 64,
                  "LDS share of the EQ variants (tf_lds_bytes)");
    cf *eq_zp = bnd, *eq_w = bnd + 2 * kEqW, *eq_d = eq_w + kEqW;
    float *g_l = reinterpret_cast<float *>(eq_d + kEqDLen);
    // BWIN (round 6; modes II - IV, packed dual transform, 45 taps): the boundary filter reads a register window of fifteen samples
    // per lane; the last tap group's window runs up to three slots past the buffers (under zero taps): four slots of zeros there
    constexpr bool BWIN = VAR.bwin;
    constexpr int kBndElems = 4 * KB + (BWIN ? 4 : 0);
    uint32_t *bitbuf = reinterpret_cast<uint32_t *>(bnd + NH * (EQ ? kEqElems : (FIR && !WIN) ? kBndElems : ((WIN && !FIR) ? 7 * kWinMax : 0)));
    bnd += hoff(kBndElems);
    constexpr int kBitWords = (3 * N / 4) / 16;  // K/4 bytes = K/16 dwords, K = 3N/4
    constexpr int kBitStride = kBitWords + 2;     // + one dummy slot per half (and one more: the halves stay 8-byte aligned)
    // small read-only tables copied to LDS once: read through global memory they compile to
    // vector loads (the output stores may alias them), and every such load drags an
    // s_waitcnt vmcnt(0) -- i.e. a wait for the previous symbol's stores -- into the loop
    // PAIR (Mode I equalised kernels, round 6): four block slots, block d in slot d & 3, fetched two blocks at a time (below)
    constexpr bool PAIR = VAR.pair;
    constexpr int kBitBlocks = PAIR ? 4 : 2;
    float *taps_l = reinterpret_cast<float *>(bitbuf + NH * (FROM_BITS ? kBitBlocks * kBitStride : 0));
    bitbuf += hoff(kBitBlocks * kBitStride);
    // (BWIN, HALVES: the boundary filter holds its taps in registers -- no tap table; the 512 bytes are what lets an eleventh
    // Mode III workgroup onto a CU)
    constexpr int kTapsL = VAR.bwin ? 0 : kMaxTaps, kMagL = 160;
    float *mag_l = taps_l + kTapsL;
    // exp(i p pi/4) with exact 0 / +-1 entries, in 8 rotated copies: entry [rot * 8 + p] = exp(i (p + rot) pi/4).
    // The coded-bits path keeps its differential phases without the common "+1 eighth per symbol" term and
    // unreduced (see advance); the rotation is the symbol's share, picked through the table's base address.
    cf *unit8 = reinterpret_cast<cf *>(mag_l + kMagL);
    cf *tw8_l = unit8 + 64;                             // 7 x 8 twiddles of the stride-8 stage (Fft::fill_tw8)
    F::fill_tw8(a.t.twiddle, tw8_l, t_wg);
    // CFR statistics: per-wave partials (2 + 4 floats per wave), behind everything else
    float *cfr_red = reinterpret_cast<float *>(tw8_l + 56);
    uint2 *bsh_l = reinterpret_cast<uint2 *>(cfr_red + 6 * ((T + 63) / 64));     // CFR_LEAN: [3][T] bit positions of the lane's six carriers
    if (t_wg < 64) {
        const unsigned p = ((unsigned)t_wg + ((unsigned)t_wg >> 3)) & 7u;
        const float cx = (float)((int)((kCX >> (2u * p)) & 3u) - 1);
        const float cy = (float)((int)((kCX >> (2u * ((p + 6u) & 7u))) & 3u) - 1);
        unit8[t_wg] = mk(cx, cy);
    }
    for (int i = t_wg; i < kTapsL; i += blockDim.x) taps_l[i] = FIR ? a.t.taps[i] : 0.f;
    if (EQ)
        for (int i = t; i < kEqTaps + 8; i += blockDim.x) g_l[i] = i < kEqTaps ? a.t.eq_g[i] : 0.f;
    if (EQ) {
        // (both windows, w: the tails stay zero; d: what the last boundary outputs read past its end)
        for (int i = t; i < 3 * kEqW + kEqDLen; i += blockDim.x) eq_zp[i] = mk(0.f, 0.f);
    }
    const int W = WIN ? a.overlap : 0;
    // (skipping the lane tests of the slots that cannot touch a seam: measured +10 % on the windowed default chain with s16 output,
    // +0.8 % on the equalised windowed kernel -- and -5 % on the windowed default chain with complexf output, which keeps them)
    constexpr bool kSkipTests = WIN && (FIR || OFMT != 0);
    constexpr int kWm = WIN ? (EQ ? kEqWinMax : (kWinMax < G::sym_size - N ? kWinMax : G::sym_size - N)) : 0;   // W <= kWm (tf_has_window / tf_has_eq)
    // WIN with FIR: behind everything else, sized at run time (C = ntaps - 1): two stashes of a symbol's
    // [x[N-W-C .. N) | x[0 .. W)] (C + 2W each), the next symbol's x[N-cp-W .. N-cp+W+C) (2W + C), the windowed stream
    // U around the seam (2W + 2C), the window
    const int wfC = (WIN && FIR && !EQ) ? a.ntaps - 1 : 0, wfLP = wfC + 2 * W;
    cf *wfb = tw8_l + 56 + (CFR ? 3 * ((T + 63) / 64) : 0);        // (behind cfr_red: 6 floats per wave)
    cf *wf_cur = wfb + 2 * wfLP, *wf_U = wf_cur + (2 * W + wfC);
    if (WIN && FIR) win_l = EQ ? g_l + (kEqTaps + 8) : reinterpret_cast<float *>(wf_U + (2 * W + 2 * wfC));
    if (WIN)
        for (int i = t; i < 2 * W; i += blockDim.x) win_l[i] = a.t.window[i];
    if (FROM_BITS)
        for (int i = t_wg; i < G::nb_symbols; i += blockDim.x) mag_l[i] = a.t.mag[i];
    if (BWIN && t < 4) bnd[4 * KB + t] = mk(0.f, 0.f);
    lds_barrier();
    // BWIN: the lane's twelve taps (tap group t & 3: taps 12 q ... 12 q + 11; the table is zero behind tap 44), held in registers
    // (three float4 variables, not an array: an array captured by the boundary lambda stays in scratch memory)
    float4 tapa = make_float4(0.f, 0.f, 0.f, 0.f), tapb = tapa, tapc = tapa;
    if constexpr (BWIN) {
        const float4 *tp = reinterpret_cast<const float4 *>(a.t.taps + 12 * (t & 3));      // (kMaxTaps floats, zero behind tap 44)
        tapa = tp[0]; tapb = tp[1]; tapc = tp[2];
    }
    const float4 tapm = make_float4((t & 3) == 0 ? 1.f : 0.f, (t & 3) == 1 ? 1.f : 0.f, (t & 3) == 2 ? 1.f : 0.f, (t & 3) == 3 ? 1.f : 0.f);

    constexpr int K = G::K, nsym = G::nb_symbols + 1;
    // (HALVES: blockIdx.x counts PAIRS of frames; frame0 = the pair's first frame, wave-uniform)
    const int frame0 = NH * (int)(blockIdx.x / a.chunks_per_frame);
    const int chunk = blockIdx.x - (frame0 / NH) * a.chunks_per_frame;
    // the lane's frame; a pair's second frame may lie behind the batch's last one: it is computed like the first (its input read
    // from the last frame again) and none of its stores leaves (put)
    const bool frame_ok = !HALVES || frame0 + half < a.n_frames;
    const int frame = HALVES ? min(frame0 + half, a.n_frames - 1) : frame0;
    const int s_begin = chunk * a.syms_per_chunk;
    // (the frame's last run takes whatever is left: run_symbols, api_chain.hip)
    const int s_end = chunk == a.chunks_per_frame - 1 ? nsym : min(nsym, s_begin + a.syms_per_chunk);
    // TII (f-4) inside the kernel: everything after the IFFT is linear and the null symbol takes the multiplier of symbol 1,
    // so on a frame that carries TII the null symbol's segment is g_1 times a constant segment (a.tii_seg, computed once per
    // setting) instead of zeros: stored by the workgroup that owns symbols 0 and 1 when its run is over, its last C
    // samples added to the boundary outputs that symbol 1 completes.
    constexpr bool TII_IN = FROM_BITS && GUARD && (EQ || !WIN) && !HALVES;    // (Mode III has no TII: src/TII.cpp:144-149)
    const bool tii_on = TII_IN && a.tii_seg != nullptr && s_begin == 0 && (((frame & 1) == 0) == (a.tii_insert0 != 0));
    float g1s = 1.0f;            // the multiplier of symbol 1 (wave-uniform: a scalar register)
    if (frame0 >= a.n_frames || s_begin >= nsym) return;

    const int ntaps = NT ? NT : a.ntaps;
    const int C = FIR ? ntaps - 1 : 0;  // FIR look-ahead
    constexpr int cp0 = GUARD ? G::null_size - N : 0, cp = GUARD ? G::sym_size - N : 0;
    constexpr int len0 = N + cp0, len = N + cp;

    // ---- per-lane constants ------------------------------------------------
    cf tw[F::NTW > 0 ? F::NTW : 1];
    F::template load_twiddles<true>(a.t.twiddle, tt, tw);

    // the lane's 6 active first-stage inputs: r = {0|3,1,2,5,6,7}; bin = t + T*r
    // interleaved position k: bins 1..K/2 -> k = bin-1 ; bins N-K/2.. -> k = bin-N+K
    const int r0 = (tt == 0) ? 3 : 0;
    int kpos[6];
    cf hk[6];
    cf hk8[CFR && FIR ? 8 : 1];      // CFR: the corrected spectrum is dense, all eight bins of the lane are filtered
    {
        const int rr[6] = {r0, 1, 2, 5, 6, 7};
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int bin = tt + T * rr[c];
            kpos[c] = (bin <= K / 2) ? bin - 1 : bin - N + K;
            if (FIR) hk[c] = a.t.fir_h[bin];
        }
    }
    if (CFR && FIR && !CFR_SEQ) {          // (CFR_SEQ reads them where it uses them: sixteen registers less across the transforms)
#pragma unroll
        for (int m = 0; m < 8; ++m) hk8[m] = a.t.fir_h[tt + T * m];
    }
    int bitpos[6];
    unsigned bsh[6] = {0u, 0u, 0u, 0u, 0u, 0u};   // bitpos ^ 7: the symbol loop's form of it (see advance)
    // differential state of the lane's carriers, without the
    // "+1 eighth" every data block adds to every carrier: phase of symbol s = 2 q_c + s - 1 eighths.
    // Kept as six 4-bit fields of ONE register, in quarter turns (every increment is an even number of eighths): the
    // block update and the pair sums of the gain statistic work on all fields at once.  Field of carrier c at bit
    // fpos[c]: the positive carriers 0, 1, 2 at bits 0, 4, 8; the negative ones so that bits 12.. read (-k0, -k1, -k2)
    // in the lane that holds them -- carriers (5, 4, 3) at bits (12, 16, 20), lane 0 (which pairs with itself and has
    // bin 3T in slot 0): carriers (3, 5, 4).
    unsigned P = 0u;
    unsigned fpos[6] = {0u, 4u, 8u, tt == 0 ? 12u : 20u, tt == 0 ? 20u : 16u, tt == 0 ? 16u : 12u};
    const uint8_t *fbits = nullptr;
    if (FROM_BITS) {
        fbits = a.bits + (size_t)frame * (size_t)(G::nb_symbols - 1) * (size_t)(K / 4);
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            bitpos[c] = a.t.src_carrier[kpos[c]];
            bsh[c] = (unsigned)bitpos[c] ^ 7u;
            P |= ((unsigned)a.t.phase_q[kpos[c]] & 3u) << fpos[c];
        }
        if constexpr (CFR_LEAN) {
#pragma unroll
            for (int j = 0; j < 3; ++j) bsh_l[j * T + t] = make_uint2(bsh[2 * j], bsh[2 * j + 1]);
        }
    }
    const cf *fcar = FROM_BITS ? nullptr
                               : a.carriers + (size_t)frame * (size_t)nsym * (size_t)K;
    // The frame's output through a buffer resource: every store is "scalar base + scalar offset + 32-bit lane offset"
    // (buffer_store ... offen).  Flat 64-bit addresses cost a register pair and a 64-bit add per store, and were
    // what the register allocator spilled first.  soff: wave-uniform sample index inside the frame (>= 0), voff: the
    // lane's; out-of-range lanes are exec-masked by the callers (the hardware would drop them as well).
    constexpr int kOutBytes = OFMT == 0 ? 8 : (OFMT == 1 ? 4 : 2);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char *>(a.out) + (size_t)frame0 * a.out_stride * kOutBytes, 0, (int)(NH * a.out_stride * kOutBytes), 0x00020000);
    // HALVES: the lane's frame inside the pair's two-frame window; a frame behind the batch's end gets an offset no buffer has
    // (the hardware drops what lies outside the resource: nothing of it is stored)
    const int hvoff = HALVES ? (frame_ok ? half * (int)a.out_stride : 0x07ffffff) : 0;
    unsigned nclip = 0;
    typedef unsigned v2u_ __attribute__((ext_vector_type(2)));
    // NON-TEMPORAL output stores on the coded-bits chain (round 4; aux bit 1 = nt).  In a pure bandwidth test a non-temporal
    // store is the SLOWER one (-10 %), but these kernels do not run at a bandwidth limit: they run at the
    // board's 1400 W power limit, where the stores are a third of a frame's energy (tools/microbench/energy_cost.hip) and the
    // cheaper path wins -- cfg 3 +2.5 ... 3.8 %, 16 frames per launch +10 %, 256 +4 %, cfg 4 +0.5 %, the default chain, CFR and
    // windowing unchanged (tools/experiments/exp_store_policy.sh, exp_store_policy2.sh; same-box A/B).  (16-byte stores, which the same
    // microbenchmark prices a third cheaper per byte, were tried too: neighbouring lanes swap half of their samples by DPP and
    // store pairs -- parity green, -2 % with either policy: the 40 instructions of the swap and two 512-byte halves per store.)  The chains from carriers (cfg 2, the
    // IFFT + FIR stage) ARE bandwidth-bound, at the nominal clock with power to spare, and lose 0 ... 1.5 %: they keep plain stores.
    // (tool builds only, tools/experiments/exp_store_policy.sh: -DDABGPU_STORE_AUX=n forces the policy bits of every variant's stores --
    // 0 plain, 1 sc0, 2 nt, 16 sc1 and their sums -- so that the A/B behind the figures above can be re-run from the tree)
#ifdef DABGPU_STORE_AUX
    constexpr int kStoreAux = DABGPU_STORE_AUX;
#else
    // (Mode III, round 6: its symbols are 319 samples and a store instruction covers 256 bytes per frame -- one whole line and two
    // halves; streamed out non-temporally the halves reach memory unmerged.  Plain stores: 0.370 -> 0.404 of the roofline, same box;
    // Mode II, 512 bytes per store, is indifferent, Mode IV keeps nt: profiles/r06_store_policy_modes.txt.  Regrouping the two
    // frames' slots by v_permlane32_swap so that a store instruction writes 512 contiguous bytes of ONE frame: -1.5 %, not kept.)
    constexpr int kStoreAux = (FROM_BITS && LOGN != 8) ? 2 : 0;
#endif
    auto put = [&](int soff, int voff, cf y) __attribute__((always_inline)) {
        if constexpr (OFMT == 1) {
            __builtin_amdgcn_raw_buffer_store_b32(s16_pack(y, nclip), orsrc, voff * 4, soff * 4, kStoreAux);
        } else if constexpr (OFMT == 2 || OFMT == 3) {
            __builtin_amdgcn_raw_buffer_store_b16(b8_pack<(OFMT == 3 ? 3 : 2)>(y, nclip), orsrc, voff * 2, soff * 2, kStoreAux);
        } else {
            const v2u_ d = {__builtin_bit_cast(unsigned, y.x), __builtin_bit_cast(unsigned, y.y)};
            __builtin_amdgcn_raw_buffer_store_b64(d, orsrc, (voff + hvoff) * 8, soff * 8, kStoreAux);
        }
    };

    // advance the differential state over one data block (K/4 bytes: I bits, then Q bits) staged in LDS.
    // The block is staged DWORD-INTERLEAVED (round 5): dword j of the I half at slot 2 j, dword j of the Q half at slot
    // 2 j + 1 (the lane that fetched a dword simply parks it at another address: bit_slot), so the I and the Q bit of a
    // carrier arrive in ONE 8-byte read -- six ds_read_b64 per lane and symbol where the byte layout took twelve
    // ds_read_u8, whose 2 x 32-lane groups hit two dwords of one bank in almost every instruction (the 48 dwords of a
    // half block over 32 banks; 11 % of the LDS cycles of the cfg 3 kernel were bank conflicts, all of them here).  Bit of
    // carrier n inside its dword: byte (n >> 3) & 3, bit 7 - (n & 7) of it = (n & 31) ^ 7 -- v_bfe_u32 reads the low five
    // bits of its offset operand, so n ^ 7 (bsh) serves as it is, and bits 5 ... of it are the dword index.
    auto advance = [&](const uint32_t *blk) __attribute__((always_inline)) {
        uint2 w[6];
        unsigned bs[6];
        if constexpr (CFR_LEAN) {
            // (at the register limit: the six bit positions live in LDS -- three 8-byte reads per symbol -- instead of six lane
            // registers and six more for the dword offsets hoisted out of the loop; spilled, they came back from scratch behind a
            // wait for the previous symbol's stores)
            int tl = t;
            asm volatile("" : "+v"(tl));
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const uint2 b2 = bsh_l[j * T + tl];
                bs[2 * j] = b2.x;
                bs[2 * j + 1] = b2.y;
            }
        } else {
#pragma unroll
            for (int c = 0; c < 6; ++c) bs[c] = bsh[c];
        }
#pragma unroll
        for (int c = 0; c < 6; ++c)
            w[c] = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(blk) + (__builtin_amdgcn_ubfe(bs[c], 5u, 6u) << 3));
        unsigned I = 0u, Q = 0u;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            I |= __builtin_amdgcn_ubfe(w[c].x, bs[c], 1u) << fpos[c];
            Q |= __builtin_amdgcn_ubfe(w[c].y, bs[c], 1u) << fpos[c];
        }
        // (I, Q) = 00 -> 0, 10 -> 1, 11 -> 2, 01 -> 3 quarter turns, in every field at once; the guard bits absorb the carry
        P = (P + ((I ^ Q) | (Q << 1))) & 0x333333u;
    };
    // global -> register half of the staging of block d: lanes 0 .. K/16-1 fetch one dword
    // each.  Kept free of divergent control flow on purpose (the other lanes re-read word 0
    // and later park it in a dummy LDS slot): a load or its wait inside an exec-masked
    // branch makes the compiler re-wait vmcnt(0) -- i.e. for the previous symbol's stores --
    // at the top of the next iteration.
    auto fetch_block = [&](int d) __attribute__((always_inline)) -> uint32_t {
        const int dd = min(max(d, 0), G::nb_symbols - 2);
        int tf = t < kBitWords ? t : 0;
        if constexpr (CFR_LEAN) asm volatile("" : "+v"(tf));    // (the address formed here, not held as a lane register pair)
        return reinterpret_cast<const uint32_t *>(fbits + (size_t)dd * (size_t)(K / 4))[tf];
    };
    // (I dword j -> slot 2 j, Q dword j -> slot 2 j + 1; kBitWords = dummy slot)
    const int bit_slot = t < kBitWords / 2 ? 2 * t : (t < kBitWords ? 2 * (t - kBitWords / 2) + 1 : kBitWords);

    // PAIR: the vector load of the prefetch shares its in-order counter (vmcnt) with the stores, so waiting for it is waiting for every
    // store of the previous iteration (profiles/r06_prefetch_wait_bound.txt: +4.7 % with the wait out of the loop).  Two consecutive
    // blocks -- 192 contiguous dwords, one per lane of the first three waves -- are fetched on every EVEN symbol and parked behind its
    // transform; the odd symbol in between neither loads nor waits.  Block d lives in slot d & 3: pair p = blocks 2p, 2p + 1.
    static_assert(!PAIR || (T >= 2 * kBitWords && FROM_BITS), "PAIR: a lane per dword of two blocks");
    // (the lane's block, dword and LDS slot inside a pair are re-derived from its index where they are used -- a handful of integer
    // instructions every other symbol -- instead of living in three lane registers across the loop: the kernel sits at 128)
    auto pair_lane = [&](int &b, int &j, int &slot) __attribute__((always_inline)) {
        int tl = t;
        asm volatile("" : "+v"(tl));
        b = (tl >= kBitWords && tl < 2 * kBitWords) ? 1 : 0;
        j = tl - b * kBitWords;
        slot = tl < 2 * kBitWords ? b * kBitStride + (j < kBitWords / 2 ? 2 * j : 2 * (j - kBitWords / 2) + 1) : kBitWords;
        if (tl >= 2 * kBitWords) j = 0;
    };
    // The load and its wait are written out (inline assembly, tied through the loaded register): left to the compiler, the symbol
    // that does NOT fetch inherits "this register may have a load in flight" where the two paths meet and waits there all the same
    // -- for nothing of its own, i.e. for the previous iteration's stores, which is the wait this exists to halve.
    auto fetch_pair = [&](int pr) __attribute__((always_inline)) -> uint32_t {
        int b, j, slot;
        pair_lane(b, j, slot);
        const int blk = min(max(2 * pr + b, 0), G::nb_symbols - 2);
        const uint32_t *src = reinterpret_cast<const uint32_t *>(fbits + (size_t)blk * (size_t)(K / 4)) + j;
        uint32_t w;
        asm volatile("global_load_dword %0, %1, off" : "=v"(w) : "v"(src));
        return w;
    };
    auto park_pair = [&](int pr, uint32_t w) __attribute__((always_inline)) {
        int b, j, slot;
        pair_lane(b, j, slot);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(w));
        bitbuf[2 * (pr & 1) * kBitStride + slot] = w;
    };

    // the lane's 6 active carriers of symbol s
    // MAG_IN_GAIN (the equalised-boundary variant with GainControl: every sample of the symbol is scaled by g after the
    // transform anyway): the symbol's common carrier magnitude |y_s| is not applied to the twelve carrier components here but
    // to the gain, g |y_s| -- one scalar product instead of twelve (everything between the carriers and the scaling is linear)
    constexpr bool MAG_IN_GAIN = EQ && GAIN;
    auto load_active = [&](int s, cf *val) __attribute__((always_inline)) {
        if (FROM_BITS) {
            const float mg = s >= 1 ? mag_l[s - 1] : 0.f;
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const unsigned rot64 = ((unsigned)(s - 1) & 7u) << 6;        // (byte offset of the rotated copy)
                const cf u = *reinterpret_cast<const cf *>(reinterpret_cast<const char *>(unit8) +
                                                          ((__builtin_amdgcn_ubfe(P, fpos[c], 2u) << 4) | rot64));
                if (MAG_IN_GAIN) val[c] = u;                                 // (the loop never sees the null symbol here)
                else val[c] = s >= 1 ? mk(u.x * mg, u.y * mg) : mk(0.f, 0.f);   // blank NULL symbol: +0
            }
        } else {
            // position of bin tt + T r: r <= 3 -> bin - 1 (positive carriers first), r >= 5 -> bin - N + K.
            // Spelled out as lane + constant so that the six loads share one address register.
            const cf *sym = fcar + (size_t)min(s, nsym - 1) * (size_t)K + tt;
            val[0] = sym[(tt == 0 ? 3 * T : 0) - 1];
            val[1] = sym[T - 1];
            val[2] = sym[2 * T - 1];
            val[3] = sym[5 * T - N + K];
            val[4] = sym[6 * T - N + K];
            val[5] = sym[7 * T - N + K];
        }
    };
    // scatter them into the first-stage register layout
    const float m_r0 = r0 == 0 ? 1.0f : 0.0f, m_r3 = 1.0f - m_r0;       // (two multiplies are two packed instructions
    auto place = [&](const cf *val, cf *v) __attribute__((always_inline)) {   //  per pair; two selects are four)
        v[0] = cscale(val[0], m_r0);
        v[3] = cscale(val[0], m_r3);
        v[4] = mk(0.f, 0.f);
        v[1] = val[1]; v[2] = val[2]; v[5] = val[3]; v[6] = val[4]; v[7] = val[5];
    };

    if (FROM_BITS) {
        // the loop below applies block s-2 on entering symbol s; bring the state to
        // "blocks 0 .. s_begin-3 applied"
        // A chunk that starts deep inside the frame needs the sum (mod 4 quarter turns) of up to 74 blocks' increments for
        // each of its carriers.  The sum is formed BIT-SLICED, 32 carriers per register: dword j of a block's I half and dword
        // j of its Q half hold the bits of carriers 32 j .. 32 j + 31, the increment of a carrier is the 2-bit number
        // (I ^ Q) + 2 Q, and a 2-bit counter per carrier kept as two bit planes (s0, s1) takes it with five bitwise operations
        // for all 32 at once:  c = s0 & l;  s0 ^= l;  s1 ^= h ^ c.  K / 32 lanes cover a block, the workgroup's lanes form
        // as many groups of them as fit, group g walks blocks g, g + NG, ... straight from global memory (coalesced dwords,
        // all loads independent), the groups' counters are added through LDS, and every lane then picks its six carriers'
        // two bits out of the planes.  (Round 2 replayed the blocks one by one through the symbol loop's gather -- 12 LDS byte
        // reads and ~50 instructions per lane and block, up to 74 times: most of a small batch's time.)
        {
            const int nblk = s_begin - 2;                        // blocks 0 .. s_begin - 3
            constexpr int W = K / 32;                            // dwords per half block
            constexpr int kThreadsTf = HALVES ? 32 : (T < 64 ? 64 : T);          // == blockDim.x (HALVES: the lanes of one frame)
            constexpr int NG = kThreadsTf / W;                   // block-parallel groups of W lanes
            static_assert(K % 32 == 0 && NG >= 1 && (2 * NG + 2) * W * 4 <= kXElems * (int)sizeof(cf), "bit-sliced prefix");
            if (nblk > 0) {                                      // (workgroup-uniform)
                uint32_t *planes = reinterpret_cast<uint32_t *>(fbuf);      // [NG][2][W], then the total [2][W]
                const int j = t % W, g = t / W;
                uint32_t s0 = 0u, s1 = 0u;
                if (g < NG) {
                    const uint32_t *col = reinterpret_cast<const uint32_t *>(fbits) + j;
                    for (int b = g; b < nblk; b += NG) {
                        const uint32_t iw = col[(size_t)b * (K / 16)], qw = col[(size_t)b * (K / 16) + W];
                        const uint32_t l = iw ^ qw, c = s0 & l;
                        s0 ^= l;
                        s1 ^= qw ^ c;
                    }
                    planes[(2 * g) * W + j] = s0;
                    planes[(2 * g + 1) * W + j] = s1;
                }
                lds_barrier_vm();
                if (t < W) {
                    uint32_t a0 = planes[t], a1 = planes[W + t];
#pragma unroll
                    for (int gg = 1; gg < NG; ++gg) {
                        const uint32_t b0 = planes[(2 * gg) * W + t], b1 = planes[(2 * gg + 1) * W + t], c = a0 & b0;
                        a0 ^= b0;
                        a1 ^= b1 ^ c;
                    }
                    planes[2 * NG * W + t] = a0;
                    planes[(2 * NG + 1) * W + t] = a1;
                }
                lds_barrier();
                unsigned inc = 0u;
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    // carrier n: bit 7 - (n & 7) of byte n >> 3 of the half block, i.e. of byte (n >> 3) & 3 of dword n >> 5
                    const unsigned n = (unsigned)bitpos[c], bit = 8u * ((n >> 3) & 3u) + 7u - (n & 7u);
                    const uint32_t lo = planes[2 * NG * W + (n >> 5)], hi = planes[(2 * NG + 1) * W + (n >> 5)];
                    inc |= (__builtin_amdgcn_ubfe(lo, bit, 1u) | (__builtin_amdgcn_ubfe(hi, bit, 1u) << 1)) << fpos[c];
                }
                P = (P + inc) & 0x333333u;
                lds_barrier();                                   // the planes are consumed (first exchange)
            }
        }
        // stage the block of the first symbol (block s_begin-2) into bitbuf[0]
        if constexpr (PAIR) {
            // the pair that holds the first symbol's block -- and the next pair as well when that block is the pair's second one
            // (the even symbol that would have fetched it is not part of this run)
            const int d_first = max(max(s_begin, 1) - 2, 0), p0 = d_first >> 1;
            park_pair(p0, fetch_pair(p0));
            if (d_first & 1) park_pair(p0 + 1, fetch_pair(p0 + 1));
        } else {
            bitbuf[bit_slot] = fetch_block(s_begin - 2);   // (clamped; unused when the loop starts at s <= 1)
        }
    }

    // f-3 crest-factor reduction of one symbol held 8 samples per lane (reference
    // src/OfdmGenerator.cpp:222-277 and cfr_one_iteration :310-373).  v: IFFT output in, CFR output
    // out; refv: the lane's 8 input bins (a forward transform returns every bin to the lane it came
    // from, so the error is formed in place).  stats: also the side statistics of symbol s.
    auto cfr_symbol = [&](cf *v, cf *zf, const cf *refv, int s, bool stats) __attribute__((always_inline)) {
        const float clip2 = a.cfr_clip * a.cfr_clip, eclip2 = a.cfr_errclip * a.cfr_errclip;   // :315, :339
        const bool mer_sym = stats && s > 0 && s == (a.cfr_mer_base + frame) % nsym;             // :198, :250
        constexpr int NW = (T + 63) / 64;
        // (CFR_LEAN: the symbol before CFR is not held across the three transforms for the one MER symbol of a frame -- that symbol
        // is transformed once more, when the sums are due)
        constexpr bool KEEP_BEFORE = !CFR_LEAN;
        cf before[KEEP_BEFORE ? 8 : 1];
        float pk = 0.f, sm = 0.f;
        unsigned nclip = 0, neclip = 0;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const float mag2 = v[m].x * v[m].x + v[m].y * v[m].y;
            pk = fmaxf(pk, mag2);
            sm += mag2;
            if (KEEP_BEFORE) before[m] = v[m];
            // :320-330, x * sqrt(clip^2 / |x|^2) as x * clip * rsq(|x|^2): one v_rsq_f32 and a select, no branch, no
            // correctly rounded division and square root (twenty-odd instructions each; the factor is good to 1 ulp and
            // the transforms that follow round more than that)
            const bool over = mag2 > clip2;
            v[m] = cscale(v[m], over ? fabsf(a.cfr_clip) * fast_rsq(mag2) : 1.0f);
            nclip += over ? 1u : 0u;
        }
        if (stats) {
            // PAPRStats::process_block before CFR (src/PAPRStats.cpp:41-60): per-wave partials now,
            // combined by lane 0 behind the forward transform's barriers
            pk = lane_on ? pk : 0.f; sm = lane_on ? sm : 0.f;
            wave_max_sum_dpp(pk, sm);
            if ((t & 63) == 0) { cfr_red[2 * (t >> 6)] = pk; cfr_red[2 * (t >> 6) + 1] = sm; }
        }
        F::template run<-1, DBUF, cf, true>(v, fbuf, fpar, tw, tt, tw8_l);
        if (stats && t == 0) {
            float p = 0.f;
            double q = 0.;
#pragma unroll
            for (int w = 0; w < NW; ++w) { p = fmaxf(p, cfr_red[2 * w]); q += (double)cfr_red[2 * w + 1]; }
            double *pp = a.cfr_papr + ((size_t)frame * nsym + s) * 4;
            pp[0] = (double)p;
            pp[1] = q / (double)N;
        }
        // (CFR_LEAN: the lane's reference bins are formed again here -- six reads of the unit-vector table -- rather than held across
        // two transforms)
        cf rv[CFR_LEAN ? 8 : 1];
        if constexpr (CFR_LEAN) {
            cf val2[6];
            load_active(s, val2);
            place(val2, rv);
            refv = rv;
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const cf c = cscale(v[m], 1.0f / (float)N);         // :349-350 (a power of two: exact)
            cf e = csub(refv[m], c);
            const float mag2 = e.x * e.x + e.y * e.y;
            const bool over = mag2 > eclip2;                      // :357-360
            const float f = over ? fabsf(a.cfr_errclip) * fast_rsq(mag2) : 1.0f;
            neclip += over ? 1u : 0u;
            v[m] = mk(fmaf(e.x, f, c.x), fmaf(e.y, f, c.y));
        }
        if constexpr (CFR_SEQ) {
            // the filtered copy first (zf), then the corrected spectrum itself.  The filter's response at the lane's eight bins comes
            // from memory here (a 16 kB table, cache-resident; the previous symbol's stores, which the wait for these loads also
            // covers, are two transforms old)
            cf hh[8];
            int tl = tt;
            asm volatile("" : "+v"(tl));          // (the table address formed here, not held as a lane register pair)
#pragma unroll
            for (int m = 0; m < 8; ++m) hh[m] = a.t.fir_h[tl + T * m];
#pragma unroll
            for (int m = 0; m < 8; ++m) zf[m] = cmul(v[m], hh[m]);
            F::template run<+1, DBUF, cf, true>(zf, fbuf, fpar, tw, tt, tw8_l);
            F::template run<+1, DBUF, cf, true>(v, fbuf, fpar, tw, tt, tw8_l);
        } else if (FIR) {
            // the corrected spectrum and its filtered copy go back to the time domain as one packed transform;
            // zf receives the filtered symbol
            c2 v2[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const cf f = cmul(v[m], hk8[CFR && FIR ? m : 0]);
                v2[m] = c2{make_float2(v[m].x, f.x), make_float2(v[m].y, f.y)};
            }
            F::template run<+1, DBUF, c2, true>(v2, reinterpret_cast<c2 *>(fbuf), fpar, tw, tt, tw8_l);
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                v[m] = mk(v2[m].re.x, v2[m].im.x);
                zf[m] = mk(v2[m].re.y, v2[m].im.y);
            }
        } else {
            F::template run<+1, DBUF, cf, true>(v, fbuf, fpar, tw, tt, tw8_l);
        }
        if (stats) {
            // (a lane counts at most 8: the wave's sums are exact as floats, and the DPP adds cost no LDS round trips)
            float c1 = lane_on ? (float)nclip : 0.f, c2_ = lane_on ? (float)neclip : 0.f;
            wave_sum2_dpp(c1, c2_);
            const unsigned n1 = (unsigned)c1, n2 = (unsigned)c2_;
            if ((t & 63) == 0) {
                if (n1) atomicAdd(a.cfr_counts + 2 * (size_t)frame, n1);
                if (n2) atomicAdd(a.cfr_counts + 2 * (size_t)frame + 1, n2);
            }
            if (s > 0) {                                          // :246-248: symbol 0 is skipped
                float pk2 = 0.f, sm2 = 0.f, siq = 0.f, sdl = 0.f;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const float mag2 = v[m].x * v[m].x + v[m].y * v[m].y;
                    pk2 = fmaxf(pk2, mag2);
                    sm2 += mag2;
                    if (KEEP_BEFORE) {
                        const cf d = csub(v[m], before[m]);
                        siq += before[m].x * before[m].x + before[m].y * before[m].y;
                        sdl += d.x * d.x + d.y * d.y;
                    }
                }
                if (!KEEP_BEFORE && mer_sym) {
                    cf val2[6], b[8];
                    load_active(s, val2);
                    place(val2, b);
                    F::template run<+1, DBUF, cf, true>(b, fbuf, fpar, tw, tt, tw8_l);
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        const cf d = csub(v[m], b[m]);
                        siq += b[m].x * b[m].x + b[m].y * b[m].y;
                        sdl += d.x * d.x + d.y * d.y;
                    }
                }
                pk2 = lane_on ? pk2 : 0.f; sm2 = lane_on ? sm2 : 0.f; siq = lane_on ? siq : 0.f; sdl = lane_on ? sdl : 0.f;
                wave_max_sum3_dpp(pk2, sm2, siq, sdl);
                float *r2 = cfr_red + 2 * NW;
                if ((t & 63) == 0) {
                    r2[4 * (t >> 6)] = pk2; r2[4 * (t >> 6) + 1] = sm2;
                    r2[4 * (t >> 6) + 2] = siq; r2[4 * (t >> 6) + 3] = sdl;
                }
                lds_barrier();
                if (t == 0) {
                    float p = 0.f;
                    double q = 0., iq = 0., dl = 0.;
#pragma unroll
                    for (int w = 0; w < NW; ++w) {
                        p = fmaxf(p, r2[4 * w]);
                        q += (double)r2[4 * w + 1];
                        iq += (double)r2[4 * w + 2];
                        dl += (double)r2[4 * w + 3];
                    }
                    double *pp = a.cfr_papr + ((size_t)frame * nsym + s) * 4;
                    pp[2] = (double)p;
                    pp[3] = q / (double)N;
                    if (mer_sym) {                                // :250-273
                        a.cfr_mer[2 * (size_t)frame] = iq;
                        a.cfr_mer[2 * (size_t)frame + 1] = dl;
                    }
                }
            }
        }
    };

    // Carriers path, gain mode var: the statistic of the coded-bits path for arbitrary carriers
    // (zero DC bin, so zero mean):
    //   var(re) = (P + Re Q) / 2,  var(im) = (P - Re Q) / 2,
    //   P = sum_k |X[k]|^2,  Q = sum_k X[k] X[-k] = 2 sum over pairs {k, -k}.
    // Bin -k of the lane's three positive bins lives in lane T - t: exchange three values, leave the
    // per-wave partial sums in redf (combined by spectral_gain after at least one more barrier).
    auto spectral_partial = [&](const cf *val, float *redf) __attribute__((always_inline)) {
        cf *pw = reinterpret_cast<cf *>(phw);
        lds_barrier();                         // the previous symbol's partner reads are done
        pw[tt] = (tt == 0) ? val[3] : val[5];
        pw[T + tt] = (tt == 0) ? val[5] : val[4];
        pw[2 * T + tt] = (tt == 0) ? val[4] : val[3];
        lds_barrier();
        const int o = (T - tt) & (T - 1);
        const cf oa = pw[o], ob = pw[T + o], oc = pw[2 * T + o];
        float q = (val[0].x * oa.x - val[0].y * oa.y) + (val[1].x * ob.x - val[1].y * ob.y) +
                  (val[2].x * oc.x - val[2].y * oc.y);
        float pwr = 0.f;
#pragma unroll
        for (int c = 0; c < 6; ++c) pwr += val[c].x * val[c].x + val[c].y * val[c].y;
        q = lane_on ? 2.0f * q : 0.f; pwr = lane_on ? pwr : 0.f;
        wave_sum2_dpp(q, pwr);
        if ((t & 63) == 0) { redf[2 * (t >> 6)] = pwr; redf[2 * (t >> 6) + 1] = q; }
    };
    auto spectral_gain = [&](const float *redf) __attribute__((always_inline)) -> float {
        float P = 0.f, Q = 0.f;
#pragma unroll
        for (int w = 0; w < (T + 63) / 64; ++w) { P += redf[2 * w]; Q += redf[2 * w + 1]; }
        const float vr = fast_sqrt(fmaxf(0.5f * (P + Q), 0.f)) * a.gain.var_variance;
        const float vi = fast_sqrt(fmaxf(0.5f * (P - Q), 0.f)) * a.gain.var_variance;
        return ((int)vr == 0) ? 1.0f : 32767.0f * fast_rcp(fmaxf(vr, vi));
    };

    // gain of the NULL symbol = gain computed on symbol 1 (reference
    // src/GainControl.cpp:139-144); only matters when symbol 0 is not blank.
    float g_null = 1.0f;
    if (GVAR && s_begin == 0) {
        cf val[6];
        load_active(1, val);
        float *redf = reinterpret_cast<float *>(red + 8);
        spectral_partial(val, redf);
        lds_barrier();
        g_null = spectral_gain(redf);
    } else if (GAIN && !FROM_BITS && s_begin == 0) {
        cf val[6], v[8];
        load_active(1, val);
        place(val, v);
        F::template run<+1, DBUF, cf, true>(v, fbuf, fpar, tw, tt, tw8_l);
        if (CFR) {
            cf refv[8], zdummy[8];
            place(val, refv);
            cfr_symbol(v, zdummy, refv, 1, false);
        }
        g_null = symbol_gain_fused<T>(v, a.gain, red + 8, tt, lane_on);
    }

    // With FIR the symbol after the chunk is transformed too (first IFFT only) to
    // obtain the head that the chunk's last boundary outputs look into.
    const int s_stop = ((FIR || WIN) && s_end < nsym) ? s_end + 1 : s_end;
    int cur = 0;               // which tail buffer holds the previous symbol's tail
    int prev_pos = 0;          // stream position of the previous segment
    int prev_seg = 0;
    bool have_prev = false;

    // Run-time tap count: lane q of an output's quad takes the taps q, q + 4, ... -- FOUR of them per trip, their eight LDS
    // reads issued together and waited for once (one tap per trip is one LDS round trip per tap: twelve in a row for the
    // default filter).  Taps past the filter count as zero and read the lane's first sample again.  Same order of
    // accumulation as one tap per trip.  (Fully unrolled, the NT > 0 way, it pushes these variants into spilling.)
    auto fir_quad_lane = [&](const cf *sp0, int q) __attribute__((always_inline)) -> cf {
        cf acc = mk(0.f, 0.f);
        if (NT > 0) {
            // tap count known: all reads of a lane at base + immediate, issued together and waited for once.  The last group of
            // four runs past the filter for q > 0: the zero padding of the tap table cancels it, and its sample read is
            // redirected to an address inside the buffer.
            constexpr int KT = NT > 0 ? (NT + 3) / 4 : 1, REM = NT - 4 * (KT - 1);    // lanes q < REM own a tap in the last group
            const cf *sp = sp0 + q;
            const float *tq = taps_l + q;
            cf x[KT];
            float tp[KT];
#pragma unroll
            for (int k = 0; k < KT - 1; ++k) { x[k] = sp[4 * k]; tp[k] = tq[4 * k]; }
            x[KT - 1] = (REM == 4 || q < REM) ? sp[4 * (KT - 1)] : sp[0];
            tp[KT - 1] = tq[4 * (KT - 1)];
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                acc.x = fmaf(x[k].x, tp[k], acc.x);
                acc.y = fmaf(x[k].y, tp[k], acc.y);
            }
            return acc;
        }
        const cf *sp = sp0;
#pragma unroll 1
        for (int j = q; j < ntaps; j += 16) {
            cf x[4];
            float tp[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int jj = j + 4 * k;
                const bool in = jj < ntaps;
                x[k] = sp[in ? jj : q];
                tp[k] = in ? taps_l[jj] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc.x = fmaf(x[k].x, tp[k], acc.x);
                acc.y = fmaf(x[k].y, tp[k], acc.y);
            }
        }
        return acc;
    };

    // boundary outputs of the previous segment: 4 lanes per output, shuffle-reduced
    constexpr int kThreads = HALVES ? 32 : (T < 64 ? 64 : T);      // == blockDim.x (a compile-time constant keeps it out of the loop; HALVES: the lanes of one frame)
    auto boundary = [&](const cf *src) __attribute__((always_inline)) {
        // src = [tail (C) | head (C)]; output i of the C boundary outputs = sum_j taps[j] src[i + j].
        // Four lanes (one DPP quad) share an output, lane q taking taps q, q+4, ...
        if constexpr (BWIN) {
            // Register-window form: a quad of lanes owns FOUR consecutive outputs (11 quads = 44 lanes), lane q of it the twelve taps
            // 12 q ... 12 q + 11 -- it reads the fifteen samples src[4 g + 12 q ... + 14], forms its share of all four outputs
            // from them (48 multiply-adds for 21 LDS reads; the one-output-per-quad form below: 12 for 24, taps included) and the
            // quad's shares are added by DPP.  In the small transmission modes the boundary filter's reads were as much LDS traffic
            // as the transform's exchanges (Mode III) or a third of it (Mode II).
            static_assert(!BWIN || NT == 45, "eleven quads of four outputs, four groups of twelve taps");
            for (int g0 = 0; g0 < 11; g0 += kThreads / 4) {
                const int g = g0 + (t >> 2), q = t & 3;
                const bool on = g < 11;
                const cf *wp = src + (4 * (on ? g : 0) + 12 * q);
                cf acc[4] = {mk(0.f, 0.f), mk(0.f, 0.f), mk(0.f, 0.f), mk(0.f, 0.f)};
                // (in three steps of four taps: a window of seven samples is live at a time, not fifteen -- the filtered symbol's
                // sixteen registers are still waiting for their stores here)
#pragma unroll
                for (int c4 = 0; c4 < 3; ++c4) {
                    cf w[7];
#pragma unroll
                    for (int k = 0; k < 7; ++k) w[k] = wp[4 * c4 + k];
                    const float4 tp = c4 == 0 ? tapa : (c4 == 1 ? tapb : tapc);
                    const float tk[4] = {tp.x, tp.y, tp.z, tp.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[r] = axpy(acc[r], tk[k], w[r + k]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) quad_sum2_dpp(acc[r].x, acc[r].y);
                // lane q stores output 4 g + q: picked with 0 / 1 factors (a select chain over the array index makes the compiler park
                // the four sums in scratch memory and load one back by address)
                cf y = cscale(acc[0], tapm.x);
                y = axpy(y, tapm.y, acc[1]);
                y = axpy(y, tapm.z, acc[2]);
                y = axpy(y, tapm.w, acc[3]);
                if (TII_IN && tii_on && prev_pos == 0 && on) {          // (the null symbol's boundary outputs: plus the TII segment's)
                    const cf ts = a.tii_seg[len0 - C + 4 * g + q];
                    y = mk(fmaf(g1s, ts.x, y.x), fmaf(g1s, ts.y, y.y));
                }
                if (on) put(prev_pos + prev_seg - C + 4 * g0, t, y);
            }
            return;
        }
        int tb = t;
        if constexpr (CFR_SEQ) asm volatile("" : "+v"(tb));    // (at the register limit: lane indices re-derived, not held)
        for (int i0 = 0; i0 < C; i0 += kThreads / 4) {
            const int i = i0 + (tb >> 2), q = tb & 3;
            const int ii = i < C ? i : 0;
            cf acc = fir_quad_lane(src + ii, q);
            quad_sum2_dpp(acc.x, acc.y);                                    // the 4 lanes of an output are one DPP quad
            if (TII_IN && tii_on && prev_pos == 0 && i < C) {               // (the null symbol's boundary outputs: plus the TII segment's)
                const cf ts = a.tii_seg[len0 - C + i];
                acc = mk(fmaf(g1s, ts.x, acc.x), fmaf(g1s, ts.y, acc.y));
            }
            if (i < C && q == 0) put(prev_pos + prev_seg - C + i0, tb >> 2, acc);
        }
    };

    // the same for n_out outputs at stream position out_pos .. (WIN with FIR): output i = sum_j taps[j] src[i + j]
    auto boundary_n = [&](const cf *src, int n_out, int out_pos) __attribute__((always_inline)) {
        for (int i0 = 0; i0 < n_out; i0 += kThreads / 4) {
            const int i = i0 + (t >> 2), q = t & 3;
            const int ii = i < n_out ? i : 0;
            cf acc = fir_quad_lane(src + ii, q);
            quad_sum2_dpp(acc.x, acc.y);
            if (i < n_out && q == 0) put(out_pos + i0, t >> 2, acc);
        }
    };

    // EQ: the 44 boundary outputs of the previous segment from the filtered symbols (see the template's comment).
    // zp = the previous symbol's windows; eq_w holds w (written by the lanes that own those samples, a barrier ago).
    // WIN (overlap W <= kEqWinMax): the stream around the seam is x_prev + omega d, omega = 0 before the seam, the rising
    // raised-cosine factor on its 2W samples, 1 behind it (the reference forms prev w[2W-1-j] + rise w[j]; the factors of a
    // pair add up to one), d[m] = x_cur[N-cp+m] - x_prev[m mod N] now for m in [-W, W + 44).  The 2W + 44 outputs whose
    // look-ahead reaches the seam, stream positions pos - W - 44 ... pos + W - 1:
    //     y[i] = z_prev[(i - W - 44) mod N] + sum_j taps[j] (omega d)[i - 44 + j - W]        (terms before the seam: zero)
    // -- the form above is the case W = 0, omega a step.  final: the frame ends behind the previous symbol (a zero symbol
    // follows, no window: omega is the step again), only the 44 outputs in front of the end exist.
    auto eq_boundary = [&](const cf *zp, bool final = false) __attribute__((always_inline)) {
        // d = g (*) w: 11 blocks of four outputs x 16 groups of ten taps = 176 lanes (WIN: 16 blocks, 256 lanes), the 16 groups
        // of a block being one DPP row.  Output m = m0 + r, tap jj = j0 + u reads w[q] at index q + kEqQL = m + (kEqTaps - 1 - jj)
        // (WIN: output m' = m + W, index m' + (kEqTaps - 1 - jj) + kEqWm - W).
        // (four outputs per lane over 176 lanes measured 2 % faster than three over 240)
        constexpr int kEqOut = 44 + 2 * kEqWm;
        constexpr int kEqR = 4, kEqLanes = 16 * ((kEqOut + kEqR - 1) / kEqR);
        static_assert(!EQ || LOGN != 11 || kEqLanes <= T, "EQ: outputs per lane");
        cf acc[4] = {mk(0.f, 0.f), mk(0.f, 0.f), mk(0.f, 0.f), mk(0.f, 0.f)};
        if (t < kEqLanes) {
            const int m0 = kEqR * (t >> 4), j0 = 10 * (t & 15);
            const cf *wp = eq_w + (m0 + (kEqTaps - 1 - 9) - j0) + (kEqWm - W);
            const float2 *g2 = reinterpret_cast<const float2 *>(g_l + j0);
            cf wv[kEqR + 9];
            float gg[10];
#pragma unroll
            for (int i = 0; i < kEqR + 9; ++i) wv[i] = wp[i];
#pragma unroll
            for (int u = 0; u < 5; ++u) { const float2 g = g2[u]; gg[2 * u] = g.x; gg[2 * u + 1] = g.y; }
#pragma unroll
            for (int u = 0; u < 10; ++u)
#pragma unroll
                for (int r = 0; r < kEqR; ++r) acc[r] = axpy(acc[r], gg[u], wv[r + 9 - u]);
        }
        // sum over the 16 lanes of a row: x += x(lane ^ 1), x += x(lane ^ 2), x += x(ror 4), x += x(ror 8) as v_add_f32 with
        // the DPP operand in place -- four instructions per float.  (Through update_dpp the compiler spends a
        // v_mov_b32_dpp plus a zeroing v_mov_b32 per term, and an addition.)  Hazard: a DPP read needs two wait states
        // after the VALU write of its source -- the s_nop covers the first step, the independent additions of a step
        // the following ones.
#define DABGPU_DPP6(CTRL)                                                                       \
        "v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                      \
        "v_add_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                      \
        "v_add_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                      \
        "v_add_f32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                      \
        "v_add_f32_dpp %4, %4, %4 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                      \
        "v_add_f32_dpp %5, %5, %5 " CTRL " row_mask:0xf bank_mask:0xf\n\t"
#define DABGPU_DPP2(CTRL)                                                                       \
        "v_add_f32_dpp %6, %6, %6 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                      \
        "v_add_f32_dpp %7, %7, %7 " CTRL " row_mask:0xf bank_mask:0xf\n\t"
        asm volatile("s_nop 1\n\t"
                     DABGPU_DPP6("quad_perm:[1,0,3,2]") DABGPU_DPP2("quad_perm:[1,0,3,2]")
                     DABGPU_DPP6("quad_perm:[2,3,0,1]") DABGPU_DPP2("quad_perm:[2,3,0,1]")
                     DABGPU_DPP6("row_ror:4") DABGPU_DPP2("row_ror:4") DABGPU_DPP6("row_ror:8") DABGPU_DPP2("row_ror:8")
                     : "+v"(acc[0].x), "+v"(acc[0].y), "+v"(acc[1].x), "+v"(acc[1].y), "+v"(acc[2].x), "+v"(acc[2].y),
                       "+v"(acc[3].x), "+v"(acc[3].y));
#undef DABGPU_DPP6
#undef DABGPU_DPP2
        if (t < kEqLanes && (t & 15) == 0) {
#pragma unroll
            for (int r = 0; r < kEqR; ++r) {
                const int mp = kEqR * (t >> 4) + r;
                if constexpr (WIN) {
                    const float om = final ? (mp >= W ? 1.0f : 0.0f) : (mp < 2 * W ? win_l[mp] : 1.0f);
                    eq_d[mp] = cscale(acc[r], om);
                } else {
                    eq_d[mp] = acc[r];
                }
            }
        }
        lds_barrier();
        if constexpr (WIN) {
            // output i of 64 (2W + 44 of them wanted): terms jd = max(i - 44, 0) ... i of (omega d), tap 44 - i + jd; four lanes
            // (one DPP quad) per output, twelve terms each (the tap table is zero past tap 44, eq_d past its last entry)
            // (the lane's output index re-derived here, behind an opaque move: as a loop invariant it is a lane register held --
            // or spilled, and reloaded behind a wait for the symbol's stores -- across the transform)
            int tb = t;
            asm volatile("" : "+v"(tb));
            const int i = tb >> 2, q = tb & 3;
            const int jd0 = max(i - C, 0);
            const float *tq = taps_l + max(C - i, 0) + q;
            const cf *dq = eq_d + jd0 + q;
            cf y = mk(0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 12; ++k) y = axpy(y, tq[4 * k], dq[4 * k]);
            quad_sum2_dpp(y.x, y.y);
            y = cadd(y, zp[kEqQL - W - C + i]);
            if (TII_IN && tii_on && prev_pos == 0 && i < 2 * W + C) {   // (the null symbol's outputs and its spill into symbol 1: plus the TII segment's)
                const cf ts = a.tii_seg[len0 - W - C + i];
                y = mk(fmaf(g1s, ts.x, y.x), fmaf(g1s, ts.y, y.y));
            }
            const int lo = final ? W : 0, hi = final ? W + C : 2 * W + C;
            if (q == 0 && i >= lo && i < hi) put(prev_pos + prev_seg - W - C, i, y);
            return;
        }
        // y[N-44+i] = z_prev[N-44+i] + sum_{jd <= i} taps[44-i+jd] d[jd]: four lanes (one DPP quad) per output, lane q
        // taking jd = q, q+4, ...; past jd = i the tap index runs into the table's zero padding
        {
            const int i = min(t >> 2, C - 1), q = t & 3;
            const float *tq = taps_l + (C - i) + q;
            const cf *dq = eq_d + q;
            cf y = mk(0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 11; ++k) y = axpy(y, tq[4 * k], dq[4 * k]);
            quad_sum2_dpp(y.x, y.y);
            y = cadd(y, zp[kEqQL - C + i]);
            if (TII_IN && tii_on && prev_pos == 0) {          // (the null symbol's boundary outputs: plus the TII segment's)
                const cf ts = a.tii_seg[len0 - C + i];
                y = mk(fmaf(g1s, ts.x, y.x), fmaf(g1s, ts.y, y.y));
            }
            if (t < 4 * C && q == 0) put(prev_pos + prev_seg - C, t >> 2, y);
        }
    };

    // PIPELINED boundary (round 6; Mode I, no windowing): the two steps of a boundary -- d = g (*) w, then y = z_prev + taps (*) d --
    // were two dependent LDS round trips with a barrier between them, on three of the four waves, per symbol (a quarter of a wave's
    // iteration by the phase stamps).  Step 2 of boundary b needs nothing of symbol b + 1, but it can WAIT for it: iteration s now
    // runs step 1 of its own boundary and step 2 of the PREVIOUS one side by side -- independent, their LDS reads in flight together,
    // no barrier between them (d of the previous boundary was written an iteration ago).  d and the 44 tail samples of z_prev
    // that step 2 adds are double-buffered (eq_d: [d0 | d1 | tails0 | tails1], 44 each); the last boundary of a run is finished
    // behind the loop (eq_boundary_flush).  Same arithmetic, same order: the samples do not change.
    constexpr bool kPipeB = EQ && !WIN && LOGN == 11 && OFMT == 0;     // (the integer-store forms spill with it)
    [[maybe_unused]] int pb_pos = -1;         // stream position of the pending boundary's outputs (none: -1); its buffers: eq_d half cur ^ 1
    auto eq_step2_pending = [&](int par) __attribute__((always_inline)) -> cf {
        const int i = min(t >> 2, C - 1), q = t & 3;
        const float *tq = taps_l + (C - i) + q;
        const cf *dq = eq_d + 44 * par + q;
        cf y = mk(0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 11; ++k) y = axpy(y, tq[4 * k], dq[4 * k]);
        quad_sum2_dpp(y.x, y.y);
        return cadd(y, eq_d[88 + 44 * par + i]);
    };
    auto eq_store_pending = [&](cf y) __attribute__((always_inline)) {
        const int i = min(t >> 2, C - 1);
        if (TII_IN && tii_on && pb_pos == len0 - C) {         // (the null symbol's boundary outputs: plus the TII segment's)
            const cf ts = a.tii_seg[len0 - C + i];
            y = mk(fmaf(g1s, ts.x, y.x), fmaf(g1s, ts.y, y.y));
        }
        if (t < 4 * C && (t & 3) == 0) put(pb_pos, t >> 2, y);
    };
    auto eq_boundary_pipelined = [&](const cf *zp) __attribute__((always_inline)) {
        constexpr int kEqR = 4, kEqLanes = 16 * (44 / kEqR);                   // 176
        cf acc[4] = {mk(0.f, 0.f), mk(0.f, 0.f), mk(0.f, 0.f), mk(0.f, 0.f)};
        // (the pending boundary's step 2 first -- two registers to carry -- then this boundary's step 1: no barrier, and the reads of
        // the one behind the arithmetic of the other)
        cf ypend = mk(0.f, 0.f);
        if (t < kEqLanes) {
            ypend = eq_step2_pending(cur ^ 1);
            const int m0 = kEqR * (t >> 4), j0 = 10 * (t & 15);
            const cf *wp = eq_w + (m0 + (kEqTaps - 1 - 9) - j0);
            const float2 *g2 = reinterpret_cast<const float2 *>(g_l + j0);
            cf wv[kEqR + 9];
            float gg[10];
#pragma unroll
            for (int i = 0; i < kEqR + 9; ++i) wv[i] = wp[i];
#pragma unroll
            for (int u = 0; u < 5; ++u) { const float2 g = g2[u]; gg[2 * u] = g.x; gg[2 * u + 1] = g.y; }
#pragma unroll
            for (int u = 0; u < 10; ++u)
#pragma unroll
                for (int r = 0; r < kEqR; ++r) acc[r] = axpy(acc[r], gg[u], wv[r + 9 - u]);
        }
#define DABGPU_DPP6(CTRL)                                                                       \
        "v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                      \
        "v_add_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                      \
        "v_add_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                      \
        "v_add_f32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                      \
        "v_add_f32_dpp %4, %4, %4 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                      \
        "v_add_f32_dpp %5, %5, %5 " CTRL " row_mask:0xf bank_mask:0xf\n\t"
#define DABGPU_DPP2(CTRL)                                                                       \
        "v_add_f32_dpp %6, %6, %6 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                      \
        "v_add_f32_dpp %7, %7, %7 " CTRL " row_mask:0xf bank_mask:0xf\n\t"
        asm volatile("s_nop 1\n\t"
                     DABGPU_DPP6("quad_perm:[1,0,3,2]") DABGPU_DPP2("quad_perm:[1,0,3,2]")
                     DABGPU_DPP6("quad_perm:[2,3,0,1]") DABGPU_DPP2("quad_perm:[2,3,0,1]")
                     DABGPU_DPP6("row_ror:4") DABGPU_DPP2("row_ror:4") DABGPU_DPP6("row_ror:8") DABGPU_DPP2("row_ror:8")
                     : "+v"(acc[0].x), "+v"(acc[0].y), "+v"(acc[1].x), "+v"(acc[1].y), "+v"(acc[2].x), "+v"(acc[2].y),
                       "+v"(acc[3].x), "+v"(acc[3].y));
#undef DABGPU_DPP6
#undef DABGPU_DPP2
        if (t < kEqLanes && (t & 15) == 0) {
#pragma unroll
            for (int r = 0; r < kEqR; ++r) eq_d[44 * cur + kEqR * (t >> 4) + r] = acc[r];
        }
        // the 44 samples of z_prev this boundary's step 2 will add -- kept, the window they come from is rewritten by then
        // (the lanes behind the 176: otherwise idle here)
        if (t >= kEqLanes && t < kEqLanes + 44) eq_d[88 + 44 * cur + (t - kEqLanes)] = zp[kEqQL - C + (t - kEqLanes)];
        if (pb_pos >= 0 && t < kEqLanes) eq_store_pending(ypend);
        pb_pos = prev_pos + prev_seg - C;                      // this boundary becomes the pending one (the caller flips `cur`)
    };
    // the pending boundary's step 2 behind the loop (a barrier first: its d was written in the last iteration)
    auto eq_boundary_flush = [&]() __attribute__((always_inline)) {
        if (pb_pos < 0) return;
        lds_barrier();
        eq_store_pending(eq_step2_pending(cur ^ 1));
        pb_pos = -1;
    };

    // The same for transmission mode IV (round 6; no windowing, no TII -- modes III and IV have none --, complexf output): the
    // workgroup has 128 lanes, the 176 lane-jobs of either step are done in two passes.  (Kept apart from the Mode I form above so
    // that the headline kernel's instruction stream is exactly what it was.)
    auto eq_boundary_small = [&](const cf *zp) __attribute__((always_inline)) {
        constexpr int kEqR = 4, kEqLanes = 16 * ((44 + kEqR - 1) / kEqR);      // 176
        constexpr int kEqThreads = T < 64 ? 64 : T;
#pragma unroll
        for (int eq_base = 0; eq_base < kEqLanes; eq_base += kEqThreads) {
            const int tj = eq_base + t;
            cf acc[4] = {mk(0.f, 0.f), mk(0.f, 0.f), mk(0.f, 0.f), mk(0.f, 0.f)};
            if (tj < kEqLanes) {
                const int m0 = kEqR * (tj >> 4), j0 = 10 * (tj & 15);
                const cf *wp = eq_w + (m0 + (kEqTaps - 1 - 9) - j0);
                const float2 *g2 = reinterpret_cast<const float2 *>(g_l + j0);
                cf wv[kEqR + 9];
                float gg[10];
#pragma unroll
                for (int i = 0; i < kEqR + 9; ++i) wv[i] = wp[i];
#pragma unroll
                for (int u = 0; u < 5; ++u) { const float2 g = g2[u]; gg[2 * u] = g.x; gg[2 * u + 1] = g.y; }
#pragma unroll
                for (int u = 0; u < 10; ++u)
#pragma unroll
                    for (int r = 0; r < kEqR; ++r) acc[r] = axpy(acc[r], gg[u], wv[r + 9 - u]);
            }
#pragma unroll
            for (int r = 0; r < kEqR; ++r) row16_sum2_dpp(acc[r].x, acc[r].y);
            if (tj < kEqLanes && (tj & 15) == 0) {
#pragma unroll
                for (int r = 0; r < kEqR; ++r) eq_d[kEqR * (tj >> 4) + r] = acc[r];
            }
        }
        lds_barrier();
#pragma unroll
        for (int eq_base = 0; eq_base < 4 * 44; eq_base += kEqThreads) {
            const int tj = eq_base + t;
            const int i = min(tj >> 2, C - 1), q = tj & 3;
            const float *tq = taps_l + (C - i) + q;
            const cf *dq = eq_d + q;
            cf y = mk(0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 11; ++k) y = axpy(y, tq[4 * k], dq[4 * k]);
            quad_sum2_dpp(y.x, y.y);
            y = cadd(y, zp[kEqQL - C + i]);
            if (tj < 4 * C && q == 0) put(prev_pos + prev_seg - C, tj >> 2, y);
        }
    };

    // Input of symbol s+1 is requested while symbol s is being transformed and BEFORE
    // symbol s is stored: vmcnt retires in order, so a load issued after the stores would
    // make every symbol wait for the previous symbol's HBM writes.
    int bb = 0;                 // which bitbuf half holds the block of the current symbol
    cf nval[6];                 // carriers path: the next symbol's active carriers
    if (!FROM_BITS) load_active(s_begin, nval);

    // Coded-bits path: the NULL symbol is blank (no TII), its segment is exact zeros and its
    // tail is a zero tail.  Peeling it off makes the guard length a loop constant, so the
    // lane predicates of the prefix copy and of the boundary samples hoist out of the loop.
    int s_loop = s_begin;
    if (FROM_BITS && s_begin == 0) {
        const int nz = len0 - C - W;                  // the last C outputs belong to `boundary` (W: to the seam)
        if (!tii_on)
            for (int i0 = 0; i0 < nz; i0 += kThreads)
                if (i0 + t < nz) put(i0, t, mk(0.f, 0.f));
        if (EQ) {
            for (int i = t; i < kEqW; i += (int)blockDim.x) eq_zp[cur * kEqW + i] = mk(0.f, 0.f);
        } else if (FIR && !WIN) {
            for (int i = t; i < KB; i += (HALVES ? 32 : (int)blockDim.x)) bnd[cur * 2 * KB + i] = mk(0.f, 0.f);
        }
        if (FIR) {
            have_prev = true;
            prev_pos = 0;
            prev_seg = len0;
        }
        if (WIN && FIR && !EQ) {
            for (int i = t; i < wfLP; i += (int)blockDim.x) wfb[cur * wfLP + i] = mk(0.f, 0.f);
        } else if (WIN && !FIR) {
            for (int i = t; i < 2 * W; i += (int)blockDim.x) wbuf[cur * 2 * kWinMax + i] = mk(0.f, 0.f);
            have_prev = true;
        }
        // bring the staged block to the state the loop expects at s = 1 (block index -1: none)
        s_loop = 1;
    }

    // (per-phase cycle stamps: an empty type unless the library is a -DDABGPU_PHASE_TIMING tool build, see device_common.h)
    PhaseTimer pt;
    unsigned pt_iterations = 0;
    pt.begin();
    for (int s = s_loop; s < s_stop; ++s) {
        if (FROM_BITS) __builtin_assume(s >= 1);    // (the blank null symbol was peeled off above)
        pt.stamp(PH_LOOP);
        ++pt_iterations;
        const bool lookahead = s >= s_end;      // FIR / WIN only: no output for this symbol
        cf val[6], v[8];
        uint32_t pf = 0u;
        if (FROM_BITS) {
            lds_barrier();                    // bitbuf[bb] written (prologue / previous iteration)
            if constexpr (PAIR) {
                if (s >= 2) advance(bitbuf + ((s - 2) & 3) * kBitStride);
                if ((s & 1) == 0) pf = fetch_pair(s >> 1);     // blocks s, s + 1: of symbols s + 2, s + 3 (clamped past the end)
            } else {
                if (s >= 2) advance(bitbuf + bb * kBitStride);
                pf = fetch_block(s - 1);            // block of symbol s+1 (clamped; unused past the end)
            }
            load_active(s, val);
            if (GAIN && !CFR && a.gain.mode == 2) {
                // Gain statistics without touching the time domain.  With every carrier on the
                // unit circle (times |y_s|) and a zero DC bin:
                //   var(re) = |X|^2 (K/2 + S),  var(im) = |X|^2 (K/2 - S),
                //   S = sum over carrier pairs {k, -k} of cos(pi/4 (p_k + p_-k)),
                // because sum_n x[n]^2 = N sum_k X[k] X[-k].  Carrier -k of the lane's three positive
                // carriers lives in lane T - t (lane 0 pairs with itself): exchange one packed word.
                phw[tt] = P >> 12;                        // fields (-k0, -k1, -k2) of this lane
                lds_barrier();
                const unsigned o = phw[(T - tt) & (T - 1)];
                // Every carrier of a symbol has the same phase parity (each block adds an odd number of eighths
                // to all of them), so a pair's phase sum is an even number of eighths and its cosine is +1, 0 or -1:
                // S = #(sum = 0 mod 4 quarter turns) - #(sum = 2 mod 4).  Both phases of a pair carry the symbol's
                // rotation, s - 1 quarter turns in all.  The three sums in one addition (fields cannot carry into each
                // other: 3 + 3 + 3 < 16); counted per wave with ballots -- the additions run on the scalar unit.
                const unsigned sums = P + o + (((unsigned)(s - 1) & 3u) * 0x111u);
                int cnt = 0, cnt_hi = 0;          // (HALVES: the counts of lanes 0 .. 31 and of lanes 32 .. 63 -- two frames)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const unsigned f = sums & (3u << (4 * j));
                    // (v_cmp_eq_u32 straight into an SGPR pair: 32 = ICMP_EQ)
                    const unsigned long long b0 = __builtin_amdgcn_uicmp(f, 0u, 32), b2 = __builtin_amdgcn_uicmp(f, 2u << (4 * j), 32);
                    if constexpr (HALVES) {
                        cnt += __builtin_popcount((unsigned)b0) - __builtin_popcount((unsigned)b2);
                        cnt_hi += __builtin_popcount((unsigned)(b0 >> 32)) - __builtin_popcount((unsigned)(b2 >> 32));
                    } else {
                        cnt += __builtin_popcountll(b0 & on_mask);
                        cnt -= __builtin_popcountll(b2 & on_mask);
                    }
                }
                const float part = (float)((HALVES && half) ? cnt_hi : cnt);
                float *redf = reinterpret_cast<float *>(red + 8 * (s & 1));
                // (WIN, at the register limit: the wave's index as a scalar -- the slot's address is scalar arithmetic and a
                // move, not a lane register that stays live across the transform and gets spilled)
                if ((t & 63) == 0) redf[WIN ? __builtin_amdgcn_readfirstlane(t >> 6) : t >> 6] = part;      // combined after the transform's barriers (HALVES: lane 0 of each frame, into the frame's own slot)
            }
        } else {
#pragma unroll
            for (int c = 0; c < 6; ++c) val[c] = nval[c];
            if (s + 1 < s_stop) load_active(s + 1, nval);
            if (GAIN && !CFR && (GVAR || a.gain.mode == 2) && s > 0)
                spectral_partial(val, reinterpret_cast<float *>(red + 8 * (s & 1)));   // combined after the transform
        }
        constexpr bool DUAL = FIR && !EQ;         // unfiltered and filtered IFFT of a symbol as ONE packed transform
        cf z[8];                                  // DUAL: the filtered symbol
        cf uedge = mk(0.f, 0.f);                  // ZONLY: the lane's boundary sample of the unfiltered symbol
        if (DUAL && CFR) {
            // IFFT alone, crest-factor reduction on it, and back through the packed pair (inside cfr_symbol)
            place(val, v);
            F::template run<+1, DBUF, cf, true>(v, fbuf, fpar, tw, tt, tw8_l);
            if constexpr (CFR_SEQ) {
                cfr_symbol(v, z, nullptr, s, !lookahead);         // (forms the reference bins again itself, behind the forward transform)
            } else {
                cf refv[8];
                place(val, refv);
                cfr_symbol(v, z, refv, s, !lookahead);
            }
        } else if (DUAL) {
            // unfiltered and filtered transform of the symbol in lockstep (see struct c2)
            cf valf[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) valf[c] = cmul(val[c], hk[c]);
            place(val, v);
            place(valf, z);
            c2 v2[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) v2[r] = c2{make_float2(v[r].x, z[r].x), make_float2(v[r].y, z[r].y)};
            if constexpr (ZONLY) {
                F::template run_dual_zonly<+1>(v2, reinterpret_cast<c2 *>(fbuf), tw, tt, tw8_l, z, uedge);
            } else {
                F::template run<+1, DBUF, c2, true>(v2, reinterpret_cast<c2 *>(fbuf), fpar, tw, tt, tw8_l);
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    v[m] = mk(v2[m].re.x, v2[m].im.x);
                    z[m] = mk(v2[m].re.y, v2[m].im.y);
                }
            }
        } else {
            if (EQ) {
                // the filtered spectrum alone
#pragma unroll
                for (int c = 0; c < 6; ++c) val[c] = cmul(val[c], hk[c]);
            }
            place(val, v);
            pt.stamp(PH_INPUT);
            F::template run<+1, DBUF, cf, true>(v, fbuf, fpar, tw, tt, tw8_l, &pt);
            if constexpr (CFR_LEAN) {
                cfr_symbol(v, z, nullptr, s, !lookahead);
            } else if (CFR) {
                cf refv[8];
                place(val, refv);
                cfr_symbol(v, z, refv, s, !lookahead);       // (WIN without FIR looks one symbol ahead: no statistics for it)
            }
        }

        if (FROM_BITS) {
            // Park the prefetched block of the next symbol in LDS now, BEFORE anything of this iteration is
            // stored: vmcnt retires in order and also counts stores, so a wait for this load placed after
            // the boundary outputs' (conditional) store has to be vmcnt(0) -- every wave would sit out the
            // full HBM write latency of that store once per symbol.  (Half bb^1 was last read an iteration ago.)
            if constexpr (PAIR) {
                // (the slots of pair (s >> 1) - 2, last read an iteration ago)
                if ((s & 1) == 0) park_pair(s >> 1, pf);
            } else {
                bitbuf[(bb ^ 1) * kBitStride + bit_slot] = pf;
            }
        }

        float g = 1.0f;
        if (GAIN) {
            bool g_final = false;                             // g already carries `constant` (and the carrier magnitude)
            if (FROM_BITS && !CFR && a.gain.mode == 2) {
                const float *redf = reinterpret_cast<const float *>(red + 8 * (s & 1));
                float S = 0.f;
#pragma unroll
                for (int w = 0; w < (T + 63) / 64; ++w) S += redf[w];
                // |X| of the symbol: the table holds the COMPONENT magnitude; diagonal states
                // (odd phase, the same parity on every carrier) have modulus sqrt(2) times that
                const float mg = mag_l[s - 1];                               // the loop never sees s = 0 here
                const float par1 = (float)(1u + ((unsigned)(s - 1) & 1u));   // (the carriers' own parts are even)
                const float m2 = mg * mg * par1;
                // The multiplier in as few roundings as it takes (round 4: the scalar is held to 2e-7 of the EXACT value):
                //     g constant = [32767 constant / var_variance] / sqrt(|X|^2 (K/2 + |S|)),
                // the bracket formed in float64 on the host and handed over as a float pair (GainParams::var_c1 + var_c1_lo: its
                // own rounding alone was a constant -4e-8 at normalise = 1/50000), K/2 + |S| an exact integer, the inverse root
                // as v_rsq_f32 (1 ulp) plus one Newton step.  |X| = (1 - dlt) with dlt ~ 1e-7 the drift of the reference's
                // fp32 recurrence (the table holds the COMPONENT magnitude: |X| = mg, or mg sqrt 2 on the diagonal states):
                // 1 / |X| = 1 + dlt goes into the low word.  Where the carrier magnitude rides on the multiplier (MAG_IN_GAIN)
                // it cancels against |X| up to the factor 1 or sqrt 2, and (1 + parity)(K/2 + |S|) is an exact integer too.
                const float A = (float)(K / 2) + fabsf(S);
                const float u = MAG_IN_GAIN ? par1 * A : A;
                float r = fast_rsq(u);
                r = fmaf(0.5f * r, fmaf(-u * r, r, 1.0f), r);
                const float dlt = MAG_IN_GAIN ? 0.f
                                  : (par1 == 1.0f ? 1.0f - mg : fmaf(-mg, 2.4203e-8f, fmaf(-mg, 1.41421354f, 1.0f)));
                // "(int)(var_variance sigma_re) == 0 -> gain 1" (src/GainControl.cpp:324-331): sigma_re^2 var^2 < 1
                const bool blank = m2 * fmaxf((float)(K / 2) + S, 0.f) * a.gain.var_sq < 1.0f;
                // (MAG_IN_GAIN: dlt = 0 and the low word is the kernel argument itself.  Spelled out for WIN, at the register
                // limit: the fused form fma(c1, 0, lo) is hoisted out of the symbol loop into a lane register, spilled there)
                const float lo = (MAG_IN_GAIN && WIN) ? a.gain.var_c1_lo : fmaf(a.gain.var_c1, dlt, a.gain.var_c1_lo);
                g = blank ? a.gain.constant * (MAG_IN_GAIN ? mg : 1.0f) : fmaf(a.gain.var_c1, r, lo * r);
                g_final = true;
            } else if (!FROM_BITS && !CFR && (GVAR || a.gain.mode == 2) && s > 0) {
                g = spectral_gain(reinterpret_cast<const float *>(red + 8 * (s & 1)));
            } else if (GVAR) {
                g = g_null;                                   // s == 0
            } else if (ZONLY || EQ) {
                g = 512.0f;                                   // mode fix (the launcher keeps mode max off this variant)
            } else {
                g = (s == 0) ? g_null : symbol_gain_fused<T>(v, a.gain, red + 8 * (s & 1), tt, lane_on);
            }
            if (!g_final) g = g * a.gain.constant;
            // TII (f-4): the null symbol of the coded-bits path is added afterwards, scaled by the
            // multiplier of symbol 1 (src/GainControl.cpp:139-144)
            if (FROM_BITS && a.gain1 != nullptr && s == 1 && t == 0) a.gain1[frame] = g;
            if (TII_IN && s == 1) g1s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, g)));
            if (MAG_IN_GAIN && !g_final) g *= mag_l[s - 1];   // (symbol 1's |y| is exactly 1: the exported g1 is the plain multiplier)
        }

        // FIR variants: both transforms of the symbol take the gain here, as packed multiplies on the
        // (unfiltered, filtered) pairs the dual transform left side by side; everything below uses v and z as is
        constexpr bool PRESCALED = (DUAL || WIN || EQ) && GAIN;
        if (((WIN && !FIR) || EQ) && GAIN) {
#pragma unroll
            for (int m = 0; m < 8; ++m) v[m] = cscale(v[m], g);
        } else if (PRESCALED && ZONLY) {
            uedge = cscale(uedge, g);
#pragma unroll
            for (int m = 0; m < 8; ++m) z[m] = cscale(z[m], g);
        } else if (PRESCALED) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const float2 re = make_float2(v[m].x, z[m].x) * g, im = make_float2(v[m].y, z[m].y) * g;
                v[m] = mk(re.x, im.x);
                z[m] = mk(re.y, im.y);
            }
        }
        auto scaled = [&](cf x) __attribute__((always_inline)) -> cf {
            return (PRESCALED || !GAIN) ? x : cscale(x, g);
        };

        const int cpl = (!FROM_BITS && s == 0) ? cp0 : cp;
        const int seg = N + cpl;
        // position of this segment in the frame's output stream
        const int pos = GUARD ? (s == 0 ? 0 : len0 + (s - 1) * len) : s * N;
        if constexpr (EQ) {
            // ---- the windows of the filtered, gain-scaled symbol that the boundary outputs need ----
            // w[q] = z_cur[N - cp + q] - z_prev[q mod N] for q in [-kEqQL, kEqQH], written by the lanes that hold
            // z_cur[N - cp + q] (slots 5 and 6); the symbol's own windows around its start (slots 7 and 0) are parked
            // for the next symbol.  Index of q everywhere: q + kEqQL.
            cf *zp_prev = eq_zp + cur * kEqW, *zp_new = eq_zp + (cur ^ 1) * kEqW;
            [[maybe_unused]] constexpr int n0 = (N - cp) - kEqQL;              // first sample of the window in z_cur (1441)
            if constexpr (LOGN == 11) {
                static_assert(LOGN != 11 || (n0 >= 5 * T && n0 + kEqQL + kEqQH < 7 * T && kEqQL < T && kEqQH < T), "EQ windows: slots 5, 6, 7, 0");
                if (t >= n0 - 5 * T) { const int iw = t - (n0 - 5 * T); eq_w[iw] = csub(v[5], zp_prev[iw]); }
                if (t <= n0 + kEqQL + kEqQH - 6 * T) { const int iw = t + (6 * T - n0); eq_w[iw] = csub(v[6], zp_prev[iw]); }
                if (t >= T - kEqQL) zp_new[t - (T - kEqQL)] = v[7];
                if (t <= kEqQH) zp_new[kEqQL + t] = v[0];
            } else if (lane_on) {
                // Transmission modes II - IV: the windows are wider than a register slot (T = 128, 64, 32 lanes), and in Mode III
                // wider than the cyclic prefix -- q runs past it into the symbol's body, sample (N - cp + q) mod N: the two symbols
                // are compared as the N-periodic sequences the cyclic filtering makes of them, which is all the derivation uses.
                // Which slots can hold a sample of either window is known at compile time; the lane tests are vector work.
                static_assert(kEqQL + kEqQH + 1 <= N && kEqQH < N - kEqQL, "EQ windows: one period holds them");
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const int n = t + T * m;
                    // q of sample n in the window around N - cp: the representative of n - (N - cp) mod N in [-kEqQL, kEqQH]
                    int q = n - (N - cp);
                    if (q > kEqQH) q -= N;
                    if (q < -kEqQL) q += N;
                    if (q >= -kEqQL && q <= kEqQH) eq_w[q + kEqQL] = csub(v[m], zp_prev[q + kEqQL]);
                    if (m * T <= kEqQH && n <= kEqQH) zp_new[kEqQL + n] = v[m];
                    if ((m + 1) * T > N - kEqQL && n >= N - kEqQL) zp_new[n - (N - kEqQL)] = v[m];
                }
            }
            lds_barrier();
            pt.stamp(PH_GAIN_WINDOWS);
            // (the boundary outputs follow the symbol's own stores, below: its samples are dead registers by then)
        } else if constexpr (WIN && FIR) {
            // ---- windowed seam AND look-ahead filter: the C + 2W outputs whose 45 samples touch the seam ----
            // U = [x_prev[N-W-C .. N-W) | the 2W seam samples (as without FIR) | x_cur[N-cp+W .. N-cp+W+C)] is the stream as the
            // reference's FIRFilter sees it; output i of them, at stream position pos - W - C + i, is sum_j taps[j] U[i + j].
            cf *pprev = wfb + cur * wfLP, *pnew = wfb + (cur ^ 1) * wfLP;
            if (lane_on) {
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const int n = t + T * m, ta = n - (N - W - C), r = n - (N - cpl - W);
                    if (ta >= 0) pnew[ta] = v[m];
                    if (n < W) pnew[C + W + n] = v[m];
                    if (r >= 0 && r < 2 * W + C) wf_cur[r] = v[m];
                }
            }
            lds_barrier();
            if (have_prev) {
                for (int i = t; i < 2 * W + 2 * C; i += kThreads) {
#pragma clang fp contract(off)  // seam: products and sum rounded separately, like the reference (see guard_window_at)
                    cf u;
                    if (i < C) {
                        u = pprev[i];
                    } else if (i < C + 2 * W) {
                        const int j = i - C;
                        const cf xp = pprev[i], xr = wf_cur[j];
                        const float fp = win_l[2 * W - 1 - j], fr = win_l[j];
                        const float ar = xp.x * fp, ai = xp.y * fp, br = xr.x * fr, bi = xr.y * fr;
                        u = mk(ar + br, ai + bi);
                    } else {
                        u = wf_cur[i - C];
                    }
                    wf_U[i] = u;
                }
                lds_barrier();
                boundary_n(wf_U, C + 2 * W, pos - W - C);
            }
            cur ^= 1;
#pragma unroll
            for (int m = 0; m < 8; ++m) v[m] = z[m];
        } else if (FIR) {
            // ---- boundary samples of the unfiltered, gain-scaled symbol ---------------
            cf *tail_new = bnd + (cur ^ 1) * 2 * KB, *tail_prev = bnd + cur * 2 * KB, *head = tail_prev + C;
            if (lane_on) {
                // The last C samples sit in the top register slot(s); the head of the segment (the
                // first C samples of the cyclic prefix) in slot m_h0 and maybe the following ones.
                // Slot tests are wave-uniform, only the lane tests are vector work.
                const int m_h0 = (N - cpl) / T;
                if (ZONLY) {
                    // Tail = slot 7 of the last C lanes.  Head (cpl == cp: the head of symbol 0, whose prefix is
                    // longer in the carriers path, is read by nobody -- there is no segment before it) = samples
                    // [N - cp, N - cp + C) = slot 6 of lanes [h0, h0 + C), all in the first wave
                    constexpr int h0 = (N - cp) - 6 * T;
                    static_assert(!ZONLY || (h0 >= 0 && h0 + (NT - 1) <= 64 && NT - 1 <= 64), "boundary lanes");
                    if (t >= T - C) tail_new[t - (T - C)] = uedge;
                    if (t >= h0 && t < h0 + C) head[t - h0] = uedge;
                } else if (C <= T) {      // the usual case (45 taps, T = 256): one tail slot, at most two head slots
                    if (t >= T - C) tail_new[t - (T - C)] = scaled(v[7]);
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        if (m == m_h0 || m == m_h0 + 1) {
                            const int hn = t + T * m - (N - cpl);
                            if (hn >= 0 && hn < C) head[hn] = scaled(v[m]);
                        }
                    }
                } else {           // short FFTs (T = 32, 64) or long filters
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        const int tn = t + T * m - (N - C), hn = t + T * m - (N - cpl);
                        if (tn >= 0) tail_new[tn] = scaled(v[m]);
                        if (hn >= 0 && hn < C) head[hn] = scaled(v[m]);
                    }
                }
            }
            lds_barrier();
            if (have_prev) boundary(tail_prev);
            cur ^= 1;
#pragma unroll
            for (int m = 0; m < 8; ++m) v[m] = z[m];        // the rest of the iteration stores the filtered symbol
        }
        if (WIN && !FIR) {
            // ---- seam between the previous symbol and this one -------------------------
            cf *pprev = wbuf + cur * 2 * kWinMax, *pnew = wbuf + (cur ^ 1) * 2 * kWinMax, *rise = wbuf + 4 * kWinMax;
            if (lane_on) {
                // (W <= kWm: which register slots can hold a seam sample at all is known at compile time -- the lane tests of
                // the other slots fold away)
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const int n = t + T * m, r = n - (N - cpl - W);
                    if ((!kSkipTests || (m + 1) * T > N - kWm) && n >= N - W) pnew[n - (N - W)] = v[m];
                    if ((!kSkipTests || m * T < kWm) && n < W) pnew[W + n] = v[m];
                    if ((!kSkipTests || ((m + 1) * T > N - cp - kWm && m * T < N - cp + kWm)) && r >= 0 && r < 2 * W) rise[r] = v[m];
                }
            }
            lds_barrier();
            if (have_prev) {
                for (int j = t; j < 2 * W; j += kThreads) {
#pragma clang fp contract(off)  // products and sum rounded separately, like the reference (see guard_window_at)
                    const cf xp = pprev[j], xr = rise[j];
                    const float fp = win_l[2 * W - 1 - j], fr = win_l[j];
                    const float ar = xp.x * fp, ai = xp.y * fp, br = xr.x * fr, bi = xr.y * fr;
                    put(pos - W, j, mk(ar + br, ai + bi));
                }
            }
            cur ^= 1;
        }
        if (FROM_BITS) bb ^= 1;
        if (lookahead && !EQ) break;
        if (lane_on && !(EQ && lookahead)) {
            const int m_cp = (N - cpl) / T;   // first register slot that is also copied into the prefix
            const bool keep_tail = !WIN || s == nsym - 1;    // WIN: the last W samples belong to the next seam
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int n = t + T * m;
                const cf y = scaled(v[m]);
                // FIR: the last C belong to `boundary`; WIN: the last W to the seam (with both: the last C + W)
                // (WIN: slots that lie clear of the seam whatever the overlap -- known at compile time -- skip the lane tests)
                const bool body_clear = kSkipTests && (m + 1) * T <= N - kWm - (FIR ? (NT ? NT - 1 : kBnd - 1) : 0);
                const bool prefix_clear = !WIN || (kSkipTests && m * T - (N - cp) >= kWm);
                if (body_clear || (FIR ? n < N - C - (keep_tail ? 0 : W) : (keep_tail || n < N - W))) put(pos + cpl + T * m, t, y);
                if ((m > m_cp || (m == m_cp && n >= N - cpl)) && (prefix_clear || n - (N - cpl) >= W)) put(pos, n - (N - cpl), y);
            }
        }
        pt.stamp(PH_STORES);
        if constexpr (EQ) {
            if (have_prev) {
                if constexpr (kPipeB) eq_boundary_pipelined(eq_zp + cur * kEqW);
                else if constexpr (LOGN == 11) eq_boundary(eq_zp + cur * kEqW, false);
                else eq_boundary_small(eq_zp + cur * kEqW);
            }
            cur ^= 1;
            pt.stamp(PH_BOUNDARY);
            if (lookahead) break;
        }
        have_prev = true;
        prev_pos = pos;
        prev_seg = seg;
    }
    if (TII_IN && tii_on) {
        // the TII null symbol (all of it without FIRFilter; up to the boundary outputs with it)
        const int nz = len0 - C - W;
        for (int i0 = 0; i0 < nz; i0 += kThreads) {
            if (i0 + t < nz) {
                const cf ts = a.tii_seg[i0 + t];
                put(i0, t, mk(fmaf(g1s, ts.x, 0.f), fmaf(g1s, ts.y, 0.f)));
            }
        }
    }
    if constexpr (kPipeB) eq_boundary_flush();
    if (EQ && s_end == nsym && have_prev) {
        // end of the frame: nothing follows (a zero symbol: w = -z_prev), missing terms are dropped
        const cf *zp = eq_zp + cur * kEqW;
        lds_barrier();
        for (int i = t; i < kEqW; i += (int)blockDim.x) eq_w[i] = mk(-zp[i].x, -zp[i].y);
        lds_barrier();
        if constexpr (LOGN == 11) eq_boundary(zp, true); else eq_boundary_small(zp);
    } else if (WIN && FIR && s_end == nsym && have_prev) {
        // end of the frame: the last symbol keeps its (unwindowed) tail, nothing follows it
        const cf *stash = wfb + cur * wfLP;
        lds_barrier();
        for (int i = t; i < 2 * C; i += (int)blockDim.x) wf_U[i] = i < C ? stash[W + i] : mk(0.f, 0.f);
        lds_barrier();
        boundary_n(wf_U, C, prev_pos + prev_seg - C);
    } else if (FIR && s_end == nsym && have_prev) {
        // end of the frame: the look-ahead runs off the buffer, missing terms are
        // dropped (reference src/FIRFilter.cpp:186-191)
        lds_barrier();
        for (int i = t; i < C; i += (HALVES ? 32 : (int)blockDim.x)) bnd[cur * 2 * KB + C + i] = mk(0.f, 0.f);   // zero head
        lds_barrier();
        boundary(bnd + cur * 2 * KB);
    }
    if (OFMT != 0) s16_flush_count(nclip, a.clipped);
    pt.flush(a.phase_cycles, pt_iterations);
}

// one launch of one instantiation, named to the launch trace by its template arguments' VALUES
template <int LOGN, bool FROM_BITS, bool GAIN, bool GUARD, bool FIR, int NT, bool CFR = false, bool GVAR = false,
          bool ZONLY = false, int OFMT = 0, bool WIN = false, bool EQ = false>
inline void tf_go(dim3 grid, dim3 block, size_t lds, hipStream_t s, const TfArgs &a)
{
    if (trace_on()) {
        char name[200];
        snprintf(name, sizeof name, "tf_kernel<logn=%d bits=%d gain=%d guard=%d fir=%d nt=%d cfr=%d gvar=%d zonly=%d ofmt=%d win=%d eq=%d>",
                 LOGN, (int)FROM_BITS, (int)GAIN, (int)GUARD, (int)FIR, NT, (int)CFR, (int)GVAR, (int)ZONLY, OFMT, (int)WIN, (int)EQ);
        trace_launch(name);
    }
    hipLaunchKernelGGL((tf_kernel<LOGN, FROM_BITS, GAIN, GUARD, FIR, NT, CFR, GVAR, ZONLY, OFMT, WIN, EQ>), grid, block, lds, s, a);
}

template <int LOGN, int NT> hipError_t launch_tf_n(const TfArgs &a, unsigned flags, hipStream_t s)
{
    constexpr int T = (1 << LOGN) / 8;
    typedef ModeGeom<LOGN> G;
    if (a.g.K != G::K || a.g.nb_symbols != G::nb_symbols || a.g.null_size != G::null_size ||
        a.g.sym_size != G::sym_size)
        return hipErrorInvalidValue;
    const dim3 block(T < 64 ? 64 : T);
    const dim3 grid((unsigned)(a.n_frames * a.chunks_per_frame));
    const bool gvar = !(flags & TF_FROM_BITS) && (flags & TF_GAIN) && !(flags & TF_CFR) && a.gain.mode == 2;
    const size_t lds = tf_lds_bytes(LOGN, flags | (gvar ? TF_GVAR : 0), (flags & TF_FIR) ? NT : 0, a.overlap, a.ntaps);
#define TF_LAUNCH(FB, GN, GD, FR)                                                              \
    tf_go<LOGN, FB, GN, GD, FR, (FR ? NT : 0)>(grid, block, lds, s, a)
#define TF_LAUNCH_CFR(FB, GN, EPI)                                                             \
    tf_go<LOGN, FB, GN, EPI, EPI, (EPI ? NT : 0), true>(grid, block, lds, s, a)
#define TF_LAUNCH_CFR_GUARD(GN)                                                                \
    tf_go<LOGN, true, GN, true, false, 0, true>(grid, block, lds, s, a)
    const bool fb = flags & TF_FROM_BITS, gn = flags & TF_GAIN, gd = flags & TF_GUARD,
               fr = flags & TF_FIR;
    if (fr && !gd) return hipErrorInvalidValue;
    const int of = tf_ofmt(flags);                           // 0 complexf, else the DABGPU_FMT_* the kernel stores itself
    if (of && (!tf_has_fmt(a, flags) || !a.clipped)) return hipErrorInvalidValue;
    if ((flags & TF_WINDOW) && !tf_has_window(a, flags)) return hipErrorInvalidValue;
    if ((flags & TF_EQ) && !tf_has_eq(a, flags)) return hipErrorInvalidValue;
    if (flags & TF_CFR) {
        // with the whole fused epilogue (guard + FIR) or with none of it; from coded bits also with the guard interval alone
        // (firfilter is off by default in the reference's configuration: src/ConfigParser.cpp:198)
        if ((gd != fr && !(fb && gd)) || (NT != 0 && (!fr || (flags & TF_WINDOW))) || !a.cfr_counts || !a.cfr_mer || !a.cfr_papr)
            return hipErrorInvalidValue;
        if (flags & TF_WINDOW) {
            // OFDM windowing with crest-factor reduction (coded-bits chain): the windowed variants with the CFR'd symbol
            if (!tf_has_window(a, flags)) return hipErrorInvalidValue;
#define TF_LAUNCH_CFR_WIN(GN, FR) \
            tf_go<LOGN, true, GN, true, FR, 0, true, false, false, 0, true>(grid, block, lds, s, a)
            if (fr) { if (gn) TF_LAUNCH_CFR_WIN(true, true); else TF_LAUNCH_CFR_WIN(false, true); }
            else    { if (gn) TF_LAUNCH_CFR_WIN(true, false); else TF_LAUNCH_CFR_WIN(false, false); }
#undef TF_LAUNCH_CFR_WIN
        } else if (of) {
            // s16 stored by the CFR kernel itself (round 5): the Mode I coded-bits chain with the guard interval, with or without FIRFilter
            if (of != 1 || !fb || !gd) return hipErrorInvalidValue;
            if constexpr (LOGN == 11) {
                if (fr) {
                    if (gn) tf_go<11, true, true, true, true, NT, true, false, false, 1>(grid, block, lds, s, a);
                    else tf_go<11, true, false, true, true, NT, true, false, false, 1>(grid, block, lds, s, a);
                } else if constexpr (NT == 0) {
                    if (gn) tf_go<11, true, true, true, false, 0, true, false, false, 1>(grid, block, lds, s, a);
                    else tf_go<11, true, false, true, false, 0, true, false, false, 1>(grid, block, lds, s, a);
                }
            }
        } else if (gd && !fr) {
            if (gn) TF_LAUNCH_CFR_GUARD(true); else TF_LAUNCH_CFR_GUARD(false);
        } else if (fr) {
            if (fb) { if (gn) TF_LAUNCH_CFR(true, true, true); else TF_LAUNCH_CFR(true, false, true); }
            else    { if (gn) TF_LAUNCH_CFR(false, true, true); else TF_LAUNCH_CFR(false, false, true); }
        } else {
            if (fb) { if (gn) TF_LAUNCH_CFR(true, true, false); else TF_LAUNCH_CFR(true, false, false); }
            else    { if (gn) TF_LAUNCH_CFR(false, true, false); else TF_LAUNCH_CFR(false, false, false); }
        }
        return hipGetLastError();
    }
#define TF_LAUNCH_GVAR(GD, FR)                                                                 \
    tf_go<LOGN, false, true, GD, FR, (FR ? NT : 0), false, true>(grid, block, lds, s, a)
    if (gvar) {
        if constexpr (LOGN == 11 && NT == 45) if (fr && gd) {
            tf_go<11, false, true, true, true, 45, false, true, true>(grid, block, lds, s, a);
            return hipGetLastError();
        }
        if (fr) TF_LAUNCH_GVAR(true, true); else if (gd) TF_LAUNCH_GVAR(true, false); else TF_LAUNCH_GVAR(false, false);
        return hipGetLastError();
    }
#undef TF_LAUNCH_GVAR
    if (flags & TF_WINDOW) {
        if (!tf_has_window(a, flags) || (NT != 0 && !fr)) return hipErrorInvalidValue;
        if constexpr (LOGN == 11 && NT == 45) if (flags & TF_EQ) {
            // the equalised-boundary variant with the seam inside its boundary outputs (overlap <= kEqWinMax; tf_has_eq was asked above)
#define TF_LAUNCH_EQW(GN, OF) \
            tf_go<11, true, GN, true, true, 45, false, false, false, OF, true, true>(grid, block, lds, s, a)
            switch (of) {
                case 1: if (gn) TF_LAUNCH_EQW(true, 1); else TF_LAUNCH_EQW(false, 1); break;
                case 2: if (gn) TF_LAUNCH_EQW(true, 2); else TF_LAUNCH_EQW(false, 2); break;
                case 3: if (gn) TF_LAUNCH_EQW(true, 3); else TF_LAUNCH_EQW(false, 3); break;
                default: if (gn) TF_LAUNCH_EQW(true, 0); else TF_LAUNCH_EQW(false, 0);
            }
#undef TF_LAUNCH_EQW
            return hipGetLastError();
        }
        if (flags & TF_EQ) return hipErrorInvalidValue;
        if (fr) {
            if (gn) tf_go<LOGN, true, true, true, true, NT, false, false, false, 0, true>(grid, block, lds, s, a);
            else tf_go<LOGN, true, false, true, true, NT, false, false, false, 0, true>(grid, block, lds, s, a);
        } else if constexpr (NT == 0) {
            if (of) {
                // the reference's default chain (no FIRFilter) with a windowed guard interval and s16 output, Mode I
                if (of != 1 || LOGN != 11) return hipErrorInvalidValue;
                if constexpr (LOGN == 11) {
                    if (gn) tf_go<11, true, true, true, false, 0, false, false, false, 1, true>(grid, block, lds, s, a);
                    else tf_go<11, true, false, true, false, 0, false, false, false, 1, true>(grid, block, lds, s, a);
                }
                return hipGetLastError();
            }
            if (gn) tf_go<LOGN, true, true, true, false, 0, false, false, false, 0, true>(grid, block, lds, s, a);
            else tf_go<LOGN, true, false, true, false, 0, false, false, false, 0, true>(grid, block, lds, s, a);
        }
        return hipGetLastError();
    }
    if constexpr (LOGN == 11 && NT == 45) if (!fb && !gn && fr && gd) {
        tf_go<11, false, false, true, true, 45, false, false, true>(grid, block, lds, s, a);
        return hipGetLastError();
    }
    if constexpr (LOGN == 11 && NT == 45) if (fb && fr && gd && (!gn || a.gain.mode != 1)) {
        if (flags & TF_EQ) {
            // ... or the one that runs the filtered transform alone and equalises the boundary (needs the taps' inverse)
            if (!a.t.eq_g) return hipErrorInvalidValue;
#define TF_LAUNCH_EQ(GN, OF) \
            tf_go<11, true, GN, true, true, 45, false, false, false, OF, false, true>(grid, block, lds, s, a)
            switch (of) {
                case 1: if (gn) TF_LAUNCH_EQ(true, 1); else TF_LAUNCH_EQ(false, 1); break;
                case 2: if (gn) TF_LAUNCH_EQ(true, 2); else TF_LAUNCH_EQ(false, 2); break;
                case 3: if (gn) TF_LAUNCH_EQ(true, 3); else TF_LAUNCH_EQ(false, 3); break;
                default: if (gn) TF_LAUNCH_EQ(true, 0); else TF_LAUNCH_EQ(false, 0);
            }
#undef TF_LAUNCH_EQ
            return hipGetLastError();
        }
        // Mode I, default filter length, gain fix / var (or none): the variant that prunes the unfiltered transform
        if (of > 1) return hipErrorInvalidValue;           // (u8 / s8 are stored by the equalised and the no-FIRFilter variants)
        if (of == 1) {
            if (gn) tf_go<11, true, true, true, true, 45, false, false, true, 1>(grid, block, lds, s, a);
            else tf_go<11, true, false, true, true, 45, false, false, true, 1>(grid, block, lds, s, a);
            return hipGetLastError();
        }
        if (gn) tf_go<11, true, true, true, true, 45, false, false, true>(grid, block, lds, s, a);
        else tf_go<11, true, false, true, true, 45, false, false, true>(grid, block, lds, s, a);
        return hipGetLastError();
    }
    if (of == 1 && fr) {
        // s16 behind the generic FIR forms of Mode I (round 5): gain mode max with the default-length filter (NT = 45: the packed
        // dual transform), any other tap count (NT = 0)
        if (LOGN != 11 || !fb || !gd) return hipErrorInvalidValue;
        if constexpr (LOGN == 11) {
            if (gn) tf_go<11, true, true, true, true, NT, false, false, false, 1>(grid, block, lds, s, a);
            else tf_go<11, true, false, true, true, NT, false, false, false, 1>(grid, block, lds, s, a);
        }
        return hipGetLastError();
    }
    if (of) {
        // the reference's default chain (no FIRFilter) with integer output, Mode I: any gain mode
        if (LOGN != 11 || NT != 0 || !fb || !gd || fr) return hipErrorInvalidValue;   // (callers ask tf_has_fmt first)
        if constexpr (LOGN == 11 && NT == 0) {
#define TF_LAUNCH_NOFIR(GN, OF) tf_go<11, true, GN, true, false, 0, false, false, false, OF>(grid, block, lds, s, a)
            switch (of) {
                case 1: if (gn) TF_LAUNCH_NOFIR(true, 1); else TF_LAUNCH_NOFIR(false, 1); break;
                case 2: if (gn) TF_LAUNCH_NOFIR(true, 2); else TF_LAUNCH_NOFIR(false, 2); break;
                default: if (gn) TF_LAUNCH_NOFIR(true, 3); else TF_LAUNCH_NOFIR(false, 3);
            }
#undef TF_LAUNCH_NOFIR
        }
        return hipGetLastError();
    }
    if (fb) {
        if (gn) { if (fr) TF_LAUNCH(true, true, true, true); else if (gd) TF_LAUNCH(true, true, true, false); else TF_LAUNCH(true, true, false, false); }
        else    { if (fr) TF_LAUNCH(true, false, true, true); else if (gd) TF_LAUNCH(true, false, true, false); else TF_LAUNCH(true, false, false, false); }
    } else {
        if (gn) { if (fr) TF_LAUNCH(false, true, true, true); else if (gd) TF_LAUNCH(false, true, true, false); else TF_LAUNCH(false, true, false, false); }
        else    { if (fr) TF_LAUNCH(false, false, true, true); else if (gd) TF_LAUNCH(false, false, true, false); else TF_LAUNCH(false, false, false, false); }
    }
#undef TF_LAUNCH
#undef TF_LAUNCH_CFR
#undef TF_LAUNCH_CFR_GUARD
    return hipGetLastError();
}


// Transmission modes II - IV with the default-length filter (round 6): the two plain coded-bits chains with the compile-time tap
// count (tf_inst_<8|9|10>_45.o) -- the equalised-boundary variant (Mode IV: its 160-tap inverse and 44 x 45 correction cost less
// than the second half of a packed 1024-point transform; at 512 and 256 points they cost MORE, tf_has_eq) and the packed dual
// transform with the boundary loops unrolled.  Every other chain of those modes runs the generic kernels (NT = 0).
template <int LOGN> hipError_t launch_tf_small45(const TfArgs &a, unsigned flags, hipStream_t s)
{
    constexpr int T = (1 << LOGN) / 8;
    typedef ModeGeom<LOGN> G;
    if (a.g.K != G::K || a.g.nb_symbols != G::nb_symbols || a.g.null_size != G::null_size || a.g.sym_size != G::sym_size)
        return hipErrorInvalidValue;
    if (!tf_small45(a, flags)) return hipErrorInvalidValue;
    const dim3 block(T < 64 ? 64 : T);
    // (Mode III: two frames per workgroup -- tf_variant(...).halves --, the grid counts pairs of frames)
    constexpr bool halves = tf_variant(LOGN, true, true, true, true, 45, false, false, 0, false, false).halves;
    const dim3 grid((unsigned)((halves ? (a.n_frames + 1) / 2 : a.n_frames) * a.chunks_per_frame));
    const size_t lds = tf_lds_bytes(LOGN, flags, 45, a.overlap, a.ntaps);
    if (flags & TF_EQ) {
        if (!tf_has_eq(a, flags) || !a.t.eq_g) return hipErrorInvalidValue;
        if constexpr (LOGN == 10) {
            if (flags & TF_GAIN) tf_go<10, true, true, true, true, 45, false, false, false, 0, false, true>(grid, block, lds, s, a);
            else tf_go<10, true, false, true, true, 45, false, false, false, 0, false, true>(grid, block, lds, s, a);
        } else {
            return hipErrorInvalidValue;          // (tf_has_eq: Modes I and IV)
        }
    } else {
        if (flags & TF_GAIN) tf_go<LOGN, true, true, true, true, 45>(grid, block, lds, s, a);
        else tf_go<LOGN, true, false, true, true, 45>(grid, block, lds, s, a);
    }
    return hipGetLastError();
}

}  // namespace
}  // namespace dabgpu
