// dabgpu_internal.h -- shared between the kernel file and the C-ABI file.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace dabgpu {

constexpr int kMaxTaps = 128;        // fused (spectral) FIR of the frame kernel
constexpr int kEqTaps = 160;         // length of the inverse filter of the equalised-boundary variant (TF_EQ), centre at kEqCentre:
constexpr int kEqCentre = 56;        //   x[n] = sum_j eq_g[j] z[n - (j - kEqCentre)]
constexpr int kMaxTapsUnfused = 512; // direct FIR kernels (the chain falls back to them for longer filters)

// Transmission-mode geometry (reference src/DabModulator.cpp:84-122).
struct Geometry {
    int mode;
    int nb_symbols;  // 76 / 153 (phase reference + data symbols)
    int K;           // carriers
    int N;           // FFT size ("spacing")
    int logN;
    int null_size;
    int sym_size;
};

// Device-resident constant tables, built once per context.
struct Tables {
    const float2 *twiddle;        // N entries, exp(+2 pi i m / N)
    const uint16_t *src_carrier;  // K: interleaved position k -> carrier index n before interleaving
    const uint16_t *dst_pos;      // K: carrier n -> interleaved position (reference's m_indices)
    const uint8_t *phase_q;       // K: quarter-turn index (0..3) of the phase reference at position k
    const float *mag;             // nb_symbols: |y| after s differential multiplications (fp32 recurrence)
    const float *taps;            // kMaxTaps floats, zero padded
    const float *window;          // 2*overlap floats (guard-interval raised cosine)
    const float2 *fir_h;          // N: frequency response of the taps, sum_j taps[j] e^{+2 pi i jk/N}
    const float *eq_g;            // TF_EQ: kEqTaps + 1 floats, the inverse of the taps on the occupied bins (see tf_kernel<..., EQ>)
};

// Launch trace (dabgpu_debug_last_variant): every kernel launch of the library goes through DABGPU_LAUNCH, which names the
// kernel to the calling thread's trace sink when one is set -- the chain entry points set it for the duration of the call.
// Off the traced path this is one thread-local pointer test per launch.
bool trace_on();
void trace_launch(const char *what);
#define DABGPU_LAUNCH(kernel, ...)                                      \
    do {                                                                \
        if (::dabgpu::trace_on()) ::dabgpu::trace_launch(#kernel);      \
        hipLaunchKernelGGL(kernel, __VA_ARGS__);                        \
    } while (0)

// stream_probe.hip: streams on hardware queues of their own (the lanes of a context; the async host path's copy stream)
bool streams_overlap(hipStream_t a, hipStream_t b);
hipError_t create_stream_apart(const hipStream_t *others, int n, hipStream_t *out, bool *own);

struct GainParams {
    int mode;            // 0 fix, 1 max, 2 var
    float constant;      // normalise * digital  (reference src/GainControl.cpp:118)
    float var_variance;
    float var_c1;        // mode var: 32767 * constant / var_variance, formed in float64 (the fused kernel's multiplier is
                         // var_c1 / sqrt(exact variance): two roundings instead of seven)
    float var_c1_lo;     // var_c1 + var_c1_lo = the float64 value to 48 bits
    float var_sq;        // var_variance^2 (the "(int)(var_variance sigma) == 0" test on sigma^2)
};

// Arguments of the fused per-transmission-frame kernel.
struct TfArgs {
    Geometry g;
    Tables t;
    GainParams gain;
    int ntaps;
    int n_frames;
    int chunks_per_frame;
    int syms_per_chunk;
    const uint8_t *bits;      // FROM_BITS: n_frames * (nb_symbols-1)*K/4 bytes
    const float2 *carriers;   // else: n_frames * (nb_symbols+1)*K samples
    float2 *out;
    size_t out_stride;        // samples per frame in `out`
    float *gain1;             // FROM_BITS, optional: per frame, the multiplier applied to symbol 1 (TII)
    // TII inside the frame kernel (f-4; variants tf_has_tii names): the unit-gain response of the TII null symbol (its segment,
    // g.null_size samples), added -- times the multiplier of symbol 1 -- on the frames whose index parity says so
    // (TII::m_insert, src/TII.cpp:241-242).  nullptr: blank null symbol (launch_tii_add adds it afterwards, or TII is off).
    const float2 *tii_seg;
    int tii_insert0;          // frame 0 of this launch carries TII (then every other one)
    // TF_CFR: crest-factor reduction inside OfdmGenerator (f-3) and its statistics
    float cfr_clip, cfr_errclip;
    int cfr_mer_base;         // the MER symbol of frame f is (cfr_mer_base + f) % (nb_symbols + 1)
    unsigned *cfr_counts;     // [frame][2]: clipped samples, clipped errors (pre-zeroed)
    double *cfr_mer;          // [frame][2]: sum |before|^2, sum |after - before|^2 of the MER symbol (pre-zeroed)
    double *cfr_papr;         // [frame][nb_symbols+1][4]: peak, mean of |x|^2 before / after CFR (pre-zeroed)
    // TF_OUT_S16 / _U8 / _S8: `out` holds 4-byte s16 pairs / 2-byte u8 or s8 pairs; the number of clipped components is ADDED to *clipped
    unsigned long long *clipped;
    // TF_WINDOW: raised-cosine overlap of the guard interval (t.window holds the 2 * overlap factors)
    int overlap;
    // tool builds only (-DDABGPU_PHASE_TIMING, tools/phase_timing.py): 16 counters the symbol loop adds its per-phase
    // shader cycles to; nullptr (and ignored) in the product
    unsigned long long *phase_cycles;
};

enum TfFlags { TF_FROM_BITS = 1, TF_GAIN = 2, TF_GUARD = 4, TF_FIR = 8, TF_CFR = 16, TF_GVAR = 32 /* internal */,
               TF_OUT_S16 = 64, TF_WINDOW = 256, TF_EQ = 512, TF_OUT_U8 = 1024, TF_OUT_S8 = 2048 };
// the integer format the frame kernel is asked to store itself: 0 = none (complexf), else DABGPU_FMT_S16 / _U8 / _S8 (1 / 2 / 3)
inline int tf_ofmt(unsigned flags) { return (flags & TF_OUT_S16) ? 1 : (flags & TF_OUT_U8) ? 2 : (flags & TF_OUT_S8) ? 3 : 0; }
inline unsigned tf_ofmt_flag(int fmt) { return fmt == 1 ? TF_OUT_S16 : fmt == 2 ? TF_OUT_U8 : fmt == 3 ? TF_OUT_S8 : 0u; }

hipError_t launch_tf(const TfArgs &a, unsigned flags, hipStream_t s);
size_t tf_lds_bytes(int logN, unsigned flags, int nt = 0, int overlap = 0, int ntaps = 0);
int tf_max_fused_taps();   // longest FIR the fused kernel handles (longer ones take the unfused path)
bool tf_has_eq(const TfArgs &a, unsigned flags);     // the equalised-boundary variant exists for this chain (TF_EQ; needs t.eq_g)
bool tf_small45(const TfArgs &a, unsigned flags);    // modes II - IV: the chain is one of those built with the compile-time tap count
bool tf_has_window(const TfArgs &a, unsigned flags); // a frame-kernel variant windows the guard interval itself (TF_WINDOW)
bool tf_has_fmt(const TfArgs &a, unsigned flags);   // a frame-kernel variant stores the format of flags' TF_OUT_* bit itself
bool tf_has_tii(const TfArgs &a, unsigned flags);   // the variant these flags select adds the TII null symbol itself (a.tii_seg)

// Stand-alone stage kernels (per-stage drop-ins and the non-fused fallbacks).
hipError_t launch_qpsk(const uint8_t *in, size_t nbytes, int K, float2 *out, hipStream_t s);
hipError_t launch_freq_interleave(const float2 *in, size_t nsamples, int K,
                                  const uint16_t *src_carrier, float2 *out, hipStream_t s);
hipError_t launch_phase_reference(const uint8_t *phase_q, int K, float2 *out, hipStream_t s);
hipError_t launch_diff_mod(const float2 *phase, const float2 *data, size_t nsym_data, int K,
                           float2 *out, hipStream_t s);
hipError_t launch_gain(const float2 *in, size_t nsym, int N, GainParams gp, float2 *out,
                       hipStream_t s);
// gain mode var with the reference's running recurrence (chain calls under dabgpu_set_gain_rounding(ctx, 1)): x0 holds
// n_frames x nsym unscaled symbols of N samples; the multipliers are left in gains[n_frames * nsym] (symbol 1's also in gain1[frame]
// when given); apply: scale the symbols in place -- or leave that to the guard kernels below (their `gains` argument)
hipError_t launch_gain_replay(float2 *x0, size_t n_frames, int nsym, int N, GainParams gp, float *gains, float *gain1,
                              bool apply, hipStream_t s);
// (gains: optional, n_frames x (nb_symbols + 1) multipliers applied to the symbols as they are gathered -- symbol 0 with symbol 1's)
hipError_t launch_guard_copy(const float2 *in, size_t n_frames, Geometry g, float2 *out,
                             hipStream_t s, const float *gains = nullptr);
hipError_t launch_guard_window(const float2 *in, size_t n_frames, Geometry g, int overlap,
                               const float *window, float2 *out, hipStream_t s, const float *gains = nullptr);
// guard interval (copy or windowed) + FIR in one pass over the IFFT output ((nb_symbols+1) x N per frame).
// For this and launch_fir, `taps` is a HOST pointer to ntaps floats: they travel as a kernel argument.
hipError_t launch_guard_fir(const float2 *in, size_t n_frames, Geometry g, int overlap, const float *window,
                            const float *taps, int ntaps, float2 *out, hipStream_t s, const float *gains = nullptr);
hipError_t launch_fir(const float2 *in, size_t frame_samples, size_t n_frames, const float *taps,
                      int ntaps, float2 *out, hipStream_t s);
hipError_t launch_poly(const float2 *in, size_t nsamples, const float *am, const float *pm,
                       float2 *out, hipStream_t s);
// a12 CicEqualizer: out[i] = in[i] * filter[i % K]
hipError_t launch_cic(const float2 *in, size_t nsamples, int K, const float *filter, float2 *out, hipStream_t s);
// f-4 TII: the sparse symbol (stand-alone stage), and its addition to a stream whose null symbol is blank
hipError_t launch_tii(const float2 *in, const uint8_t *acp, int K, int old_variant, int insert, float2 *out,
                      hipStream_t s);
hipError_t launch_tii_add(float2 *out, size_t stride, const float2 *seg, int seg_len, const float *gain1,
                          int insert0, size_t n_frames, hipStream_t s);
// f-2 FormatConverter: fmt 1 = s16, 2 = u8, 3 = s8; *clipped (device) is incremented
hipError_t launch_format(const float *in, size_t nfloats, int fmt, void *out, unsigned long long *clipped,
                         hipStream_t s);
hipError_t launch_lut(const float2 *in, size_t nsamples, float scale, const float *lut,
                      float2 *out, hipStream_t s);

// Resampler (reference src/Resampler.cpp:131-195), power-of-two FFT sizes.
struct ResamplerArgs {
    int nin, nout;          // FFT sizes (e.g. 4096 -> 16384)
    float factor;
    const float *window;    // nin
    const float2 *tw_in;    // nin entries exp(+2 pi i m / nin)
    const float2 *tw_out;   // nout entries
    const float2 *in;       // nhops * nin/2 samples (one stream)
    const float2 *halo;     // nin samples: the two hops before `in` (zeros at stream start)
    float2 *halo_out;       // non-null (the x2 / x4 kernel at nin = 4096 only, see resampler_writes_halo): the kernel leaves the
                            // last two hops of [halo | in] there -- the next call's halo -- instead of a copy launched behind it
    float2 *out;            // nhops * nout/2
    const float *poly;      // nullptr, or am[5] at [0..4] and pm[5] at [8..12]: MemlessPoly fused into the store
    unsigned long long *clipped;   // non-null: store s16 pairs (FormatConverter fused, x2 / x4 kernels with nin = 4096)
    size_t nhops;
    // rational ratios L/M (the general kernel): nout = (nin / M) * L
    int L, M;
    const float2 *tw_s;     // nin / M entries exp(+2 pi i m / (nin / M))
    const float2 *tw_l;     // L entries exp(+2 pi i m / L)
};
hipError_t launch_resampler(const ResamplerArgs &a, hipStream_t s);
bool resampler_has_s16(const ResamplerArgs &a);
bool resampler_writes_halo(const ResamplerArgs &a);   // the kernel this geometry runs honours halo_out

}  // namespace dabgpu
