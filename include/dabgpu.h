/*
 * dabgpu.h -- C-ABI of the MI355X-native DAB COFDM hot path (libdabgpu.so).
 *
 * This is the drop-in boundary: plain C, no exceptions, no C++/torch types.
 * Every stage entry point replaces the `process()` of one ODR-DabMod flowgraph
 * plugin (reference file:line cited at each declaration); the C++ adapters in
 * odr-dabmod_amd/host/ keep the reference's ModPlugin class names and
 * constructor signatures and call these functions (INTEGRATION.md shows the
 * binding a maintainer adds to src/DabModulator.cpp).
 *
 * Conventions
 *   - return 0 on success, <0 on error (DABGPU_E_*); dabgpu_last_error(ctx)
 *     gives the message the adapter throws as std::runtime_error, mirroring
 *     the reference's own size-check throws.
 *   - samples are interleaved (re,im) float32 == std::complex<float>
 *     (reference src/Buffer.h:40).
 *   - `*_process` take HOST pointers (what a ModPlugin's Buffer holds) and
 *     stage through pinned memory; `*_dev` take DEVICE pointers plus a HIP
 *     stream handle (hipStream_t cast to void*, NULL = the context's stream)
 *     and are asynchronous on that stream.
 *   - out_cap is the capacity of the output buffer in bytes; *out_bytes
 *     receives the produced length (the producer sizes its output, like
 *     Buffer::setLength at the top of every reference process()).
 *   - setters may be called from another thread (remote-control thread in the
 *     reference); they take effect at the next *_process call.
 *   - There is NO CPU fallback: every entry point fails with
 *     DABGPU_E_DEVICE when no gfx950 device/kernel is available.
 */
#ifndef DABGPU_H
#define DABGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DABGPU_API __attribute__((visibility("default")))

enum {
    DABGPU_OK = 0,
    DABGPU_E_INVALID = -1, /* bad argument / size check failed (reference throws std::runtime_error) */
    DABGPU_E_DEVICE = -2,  /* HIP error, no device, kernel image missing */
    DABGPU_E_NOMEM = -3,
    DABGPU_E_CAPACITY = -4 /* out_cap too small / n_frames > max_frames */
};

/* GainMode, reference src/GainControl.h:45 */
enum { DABGPU_GAIN_FIX = 0, DABGPU_GAIN_MAX = 1, DABGPU_GAIN_VAR = 2 };

/* stage mask for the fused chain (order of src/DabModulator.cpp:385-419).
 * QPSK map, frequency interleave, differential modulation, signal mux, OFDM
 * IFFT and guard-interval insertion always run. */
enum {
    DABGPU_STAGE_GAIN = 1u << 0,     /* GainControl   */
    DABGPU_STAGE_FIR = 1u << 1,      /* FIRFilter     */
    DABGPU_STAGE_RESAMPLE = 1u << 2, /* Resampler     */
    DABGPU_STAGE_POLY = 1u << 3,     /* MemlessPoly   */
    DABGPU_STAGE_NOGUARD = 1u << 8   /* stop after OfdmGenerator(+GainControl): no guard interval */
};

typedef struct dabgpu_ctx dabgpu_ctx;

typedef struct {
    int mode;          /* transmission mode 1..4 (src/DabModulator.cpp:84-122); 0 = IV, as the stage classes
                        * read it (src/PhaseReference.cpp:72-76, src/FrequencyInterleaver.cpp:57-58) */
    int device;        /* HIP device ordinal */
    int max_frames;    /* largest n_frames of a *_process call (scratch sizing); 0 -> 1 */
    int chunks_per_frame; /* workgroups per transmission frame for the fused kernel; 0 = auto */
} dabgpu_config;

DABGPU_API int dabgpu_create(const dabgpu_config *cfg, dabgpu_ctx **out);
DABGPU_API void dabgpu_destroy(dabgpu_ctx *ctx);
DABGPU_API const char *dabgpu_last_error(const dabgpu_ctx *ctx); /* ctx may be NULL: last create error */
DABGPU_API const char *dabgpu_version(void);

/* geometry of the configured mode (src/DabModulator.cpp:84-122) */
typedef struct {
    int mode, nb_symbols, carriers, spacing, null_size, sym_size;
    size_t tf_input_bytes; /* hot-path input per transmission frame (BlockPartitioner output) */
    size_t tf_samples;     /* complex samples per transmission frame at the native rate */
} dabgpu_geometry;
DABGPU_API int dabgpu_get_geometry(const dabgpu_ctx *ctx, dabgpu_geometry *g);

/* ---- runtime parameters (remote-control setters of the reference) -------- */

/* GainControl::set_parameter digital/mode/var, src/GainControl.cpp:505-554; normalise is the
 * constructor argument derived from the sink, src/DabMod.cpp:259-347 */
DABGPU_API int dabgpu_set_gain(dabgpu_ctx *ctx, int gain_mode, float digital, float normalise,
                               float var_variance);
/* How a CHAIN call forms the multiplier of gain mode var (computeGainVar, src/GainControl.cpp:251-340).  The reference walks a
 * symbol with four fp32 running means and four running variances (mean += (x - mean) / count), 2 x N/2 dependent divisions
 * whose rounding error (up to 5.8e-7 relative, tests/test_oracle_golden.py) is part of its output.
 *   DABGPU_GAIN_ROUNDING_EXACT (default): the exact population variance inside the frame kernel -- closer to the arithmetic
 *     the reference approximates, 5.8 ... 6.2e-7 from ITS scalar, one kernel per chain call.
 *   DABGPU_GAIN_ROUNDING_REFERENCE: the reference's recurrence operation for operation, as the stand-alone stage
 *     (dabgpu_gain_process) does: the frame kernel stops after OfdmGenerator, a kernel of four lanes per symbol replays the
 *     recurrence, and the guard interval / FIRFilter run as kernels of their own that scale the symbols as they read them.  The gain
 *     scalars then equal the reference's bit for bit on the same symbols (along a chain: within 2.3e-7, the recurrence's own
 *     sensitivity to the last bits of its input; chain total 2.5e-7 instead of 6.3e-7); cfg 3 runs at about 30 % of its rate.
 * No effect on gain modes fix and max (their scalars are exact either way).  Takes effect at the next *_process call. */
enum { DABGPU_GAIN_ROUNDING_EXACT = 0, DABGPU_GAIN_ROUNDING_REFERENCE = 1 };
DABGPU_API int dabgpu_set_gain_rounding(dabgpu_ctx *ctx, int rounding);
/* FIRFilter::load_filter_taps, src/FIRFilter.cpp:95-141 (n <= 512; up to 128 taps run fused) */
DABGPU_API int dabgpu_set_fir_taps(dabgpu_ctx *ctx, const float *taps, size_t n);
/* FIRFilter("default"): the built-in 45 taps, src/FIRFilter.cpp:59-71 */
DABGPU_API int dabgpu_set_fir_default_taps(dabgpu_ctx *ctx);
/* GuardIntervalInserter::update_window, src/GuardIntervalInserter.cpp:96-113.  (The fused chain windows overlaps up to 128
 * samples inside the frame kernel; up to 10 on the Mode I chain with a filter of up to the default length it stays the
 * one-transform-per-symbol kernel of that chain.) */
DABGPU_API int dabgpu_set_window_overlap(dabgpu_ctx *ctx, size_t overlap);
/* Resampler(inputRate, outputRate, resolution = spacing), src/Resampler.cpp:51-112;
 * resets the stream state (prev-input halo and overlap tail).  Built: every ratio L / M (the
 * rates reduced by their gcd) with M a power of two up to the FFT size -- up- AND down-sampling
 * (1.024, 1.536, 2.4, 3.072, 4.096, 6.144, 8.192 ... Msps in Mode I; x2 and x4 have their own
 * faster kernel).  Any other ratio is refused HERE with DABGPU_E_INVALID (the reference's own
 * hop loop cannot run those on whole transmission frames). */
DABGPU_API int dabgpu_set_resampler(dabgpu_ctx *ctx, size_t in_rate, size_t out_rate);
/* MemlessPoly::load_coefficients format 1, src/MemlessPoly.cpp:154-202 */
DABGPU_API int dabgpu_set_poly(dabgpu_ctx *ctx, const float am[5], const float pm[5]);
/* MemlessPoly::load_coefficients format 2, src/MemlessPoly.cpp:203-226 */
DABGPU_API int dabgpu_set_lut(dabgpu_ctx *ctx, float scalefactor, const float lut[32]);

/* ---- per-stage entry points (HOST buffers; one per reference plugin) ----- */

/* QpskSymbolMapper::process, src/QpskSymbolMapper.cpp:39-213 */
DABGPU_API int dabgpu_qpsk_process(dabgpu_ctx *ctx, const void *in, size_t in_bytes, void *out,
                                   size_t out_cap, size_t *out_bytes);
/* FrequencyInterleaver::process, src/FrequencyInterleaver.cpp:128-145 */
DABGPU_API int dabgpu_freq_interleave_process(dabgpu_ctx *ctx, const void *in, size_t in_bytes,
                                              void *out, size_t out_cap, size_t *out_bytes);
/* PhaseReference::process, src/PhaseReference.cpp:174-190 */
DABGPU_API int dabgpu_phase_reference_process(dabgpu_ctx *ctx, void *out, size_t out_cap,
                                              size_t *out_bytes);
/* DifferentialModulator::process, src/DifferentialModulator.cpp:80-108 (in0 phase ref, in1 data) */
DABGPU_API int dabgpu_diff_mod_process(dabgpu_ctx *ctx, const void *phase, size_t phase_bytes,
                                       const void *data, size_t data_bytes, void *out,
                                       size_t out_cap, size_t *out_bytes);
/* NullSymbol::process src/NullSymbol.cpp:49-57 */
DABGPU_API int dabgpu_null_symbol_process(dabgpu_ctx *ctx, void *out, size_t out_cap,
                                          size_t *out_bytes);
/* SignalMultiplexer::process, src/SignalMultiplexer.cpp:45-71: out = first ++ rest */
DABGPU_API int dabgpu_signal_mux_process(dabgpu_ctx *ctx, const void *first, size_t first_bytes,
                                         const void *rest, size_t rest_bytes, void *out,
                                         size_t out_cap, size_t *out_bytes);
/* OfdmGeneratorCF32::process, src/OfdmGenerator.cpp:157-308 (CFR off) */
DABGPU_API int dabgpu_ofdm_process(dabgpu_ctx *ctx, const void *in, size_t in_bytes, void *out,
                                   size_t out_cap, size_t *out_bytes);
/* GainControl::internal_process, src/GainControl.cpp:82-192 */
DABGPU_API int dabgpu_gain_process(dabgpu_ctx *ctx, const void *in, size_t in_bytes, void *out,
                                   size_t out_cap, size_t *out_bytes);
/* GuardIntervalInserter::process, src/GuardIntervalInserter.cpp:325-336 */
DABGPU_API int dabgpu_guard_process(dabgpu_ctx *ctx, const void *in, size_t in_bytes, void *out,
                                    size_t out_cap, size_t *out_bytes);
/* FIRFilter::internal_process, src/FIRFilter.cpp:144-309 (any length, one frame per call) */
DABGPU_API int dabgpu_fir_process(dabgpu_ctx *ctx, const void *in, size_t in_bytes, void *out,
                                  size_t out_cap, size_t *out_bytes);
/* Resampler::process, src/Resampler.cpp:131-195 (stateful across calls) */
DABGPU_API int dabgpu_resampler_process(dabgpu_ctx *ctx, const void *in, size_t in_bytes,
                                        void *out, size_t out_cap, size_t *out_bytes);
/* MemlessPoly::internal_process, src/MemlessPoly.cpp:342-411 */
DABGPU_API int dabgpu_poly_process(dabgpu_ctx *ctx, const void *in, size_t in_bytes, void *out,
                                   size_t out_cap, size_t *out_bytes);

/* Crest-factor reduction inside OfdmGenerator: the constructor arguments / RC
 * parameters cfr, clip, errorclip of OfdmGeneratorCF32 (src/OfdmGenerator.h:50-56, .cpp:376-404).
 * When enabled every symbol is clipped, transformed forward, its error against the input
 * constellation clipped, and transformed back (cfr_one_iteration, src/OfdmGenerator.cpp:310-373),
 * in dabgpu_ofdm_process and in the chain entry points alike. */
DABGPU_API int dabgpu_set_cfr(dabgpu_ctx *ctx, int enable, float clip, float error_clip);
/* Per-frame raw statistics of the most recent call that ran with CFR on (frame = index inside that
 * call); the reference's running averages (clip_stats, papr: src/OfdmGenerator.cpp:285-306,419-451,
 * src/PAPRStats.cpp) are formed from these by the caller.  Waits for the call to finish. */
typedef struct dabgpu_cfr_stats {
    uint64_t num_clip, num_error_clip; /* samples / errors clipped in the frame (:275-276) */
    uint64_t num_samples;              /* nbSymbols * spacing (:286) */
    int mer_symbol;                    /* myMERCalcIndex of this frame (:198); 0 = no MER pushed (:250) */
    double mer_sum_iq, mer_sum_delta;  /* the two sums of :262-266 for that symbol */
    int nb_symbols;                    /* entries used below (transmission-frame symbols incl. null) */
    double papr_before[154][2];        /* per symbol {peak, mean} of |x|^2 before CFR (PAPRStats::process_block) */
    double papr_after[154][2];         /* after CFR; symbol 0 is not measured (:246-248) */
} dabgpu_cfr_stats;
DABGPU_API int dabgpu_get_cfr_stats(dabgpu_ctx *ctx, size_t frame, dabgpu_cfr_stats *out);

/* TII.  dabgpu_set_tii = tii_config_t + the RC parameters enable / comb / pattern /
 * old_variant (src/TII.h:42-69, src/TII.cpp:339-372); invalid mode (only I and II carry TII), comb
 * outside [0,23] or pattern outside [0,69] is DABGPU_E_INVALID with the reference's TIIError text
 * (src/TII.cpp:119-150).  In the fused chain (dabgpu_chain_process*) an enabled TII replaces the
 * null symbol on every other frame of the stream, starting with the first (TII::m_insert,
 * src/TII.h:112, src/TII.cpp:226-242); the frame parity is per context and advances whether or
 * not TII is enabled, like the reference's.
 * dabgpu_tii_process = TII::process, src/TII.cpp:213-245: `in` is the PhaseReference symbol
 * (carriers x cf32), out the TII symbol or zeros; each call toggles the insert flag. */
DABGPU_API int dabgpu_set_tii(dabgpu_ctx *ctx, int enable, int comb, int pattern, int old_variant);
DABGPU_API int dabgpu_tii_process(dabgpu_ctx *ctx, const void *in, size_t in_bytes, void *out,
                                  size_t out_cap, size_t *out_bytes);

/* CicEqualizer(nbCarriers, spacing, R)::process, src/CicEqualizer.cpp:29-91: every
 * symbol of `carriers` samples times the per-carrier compensation gain of an R-fold, 4-stage CIC
 * interpolator.  Stage drop-in only: the fused chain does not apply it (the reference wires it in only
 * when an FPGA clockRate is configured, src/DabModulator.cpp:154-176). */
DABGPU_API int dabgpu_cic_equalizer_process(dabgpu_ctx *ctx, size_t spacing, int R, const void *in,
                                            size_t in_bytes, void *out, size_t out_cap,
                                            size_t *out_bytes);

/* FormatConverter::process, float input path, src/FormatConverter.cpp:111-178:
 * cf32 -> interleaved s16 / u8 / s8 with the reference's range test, truncation toward zero and
 * count of clipped components (FormatConverter::get_num_clipped_samples, :186-189).
 * An unknown format is DABGPU_E_INVALID ("FormatConverter: Invalid format", :171-173). */
enum {
    DABGPU_FMT_S16 = 1,
    DABGPU_FMT_U8 = 2,
    DABGPU_FMT_S8 = 3
};
/* bytes per I/Q pair, FormatConverter::get_format_size, src/FormatConverter.cpp:192-208 (0 = unknown) */
DABGPU_API size_t dabgpu_format_size(int format);
DABGPU_API int dabgpu_format_process(dabgpu_ctx *ctx, const void *in, size_t in_bytes, int format,
                                     void *out, size_t out_cap, size_t *out_bytes,
                                     size_t *num_clipped);
/* same, device-resident and asynchronous on `stream`; *d_num_clipped (device, 8 bytes, may be
 * NULL) is INCREMENTED by the number of clipped components */
DABGPU_API int dabgpu_format_process_dev(dabgpu_ctx *ctx, const void *d_in, size_t n_floats,
                                         int format, void *d_out, size_t out_cap,
                                         size_t *out_bytes, unsigned long long *d_num_clipped,
                                         void *stream);

/* Host-side helper of the fused chain, no device involved: the inverse filter the Mode I frame kernel uses to
 * correct the FIR outputs at symbol boundaries from the FILTERED symbols alone ("equalised
 * boundary").  For `taps` (the reference's FIRFilter taps, src/FIRFilter.cpp:59-133) it returns in g[0 .. 160) the
 * real filter with  x[n] = sum_j g[j] z[n - (j - 56)]  wherever z[n] = sum_j taps[j] x[n + j] (cyclically) and x
 * has energy on the 1536 occupied carriers of a 2048-point symbol only; *fit = max |G H - 1| over those carriers.
 * Returns DABGPU_OK when such a filter exists to 1e-7 (the chain then uses it), DABGPU_E_INVALID when the taps
 * have no well-conditioned inverse there or ntaps is outside 1 ... 45 (the chain keeps the packed dual transform; a
 * filter of fewer than 45 taps runs as the 45-tap filter with zero taps behind it). */
DABGPU_API int dabgpu_fir_inverse_design(const float *taps, size_t ntaps, float *g, double *fit);

/* How the fused Mode I chain forms the FIRFilter outputs whose look-ahead crosses a symbol boundary (the last
 * ntaps - 1 of every symbol; reference loop src/FIRFilter.cpp:168-191).  AUTO (the default): from the filtered
 * symbols alone through the inverse above whenever the taps have one, otherwise as DIRECT.  DIRECT: always by the
 * direct sum over the unfiltered samples (a second, pruned transform per symbol).  Same results within the FIRFilter
 * bar either way; exists so that both kernels can be selected from a test or a measurement -- it is not read from
 * the environment.  Takes effect at the next *_process call. */
enum { DABGPU_FIR_BOUNDARY_AUTO = 0, DABGPU_FIR_BOUNDARY_DIRECT = 1 };
DABGPU_API int dabgpu_set_fir_boundary_mode(dabgpu_ctx *ctx, int mode);

/* ---- the fused chain ----------------------------------------------------- */

/* FormatConverter as the last step of the chain (the reference wires it after cifPoly when the output is not
 * complexf, src/DabModulator.cpp:270-276, :407): format = 0 (complexf, the default) or DABGPU_FMT_*.  The chain's last
 * kernel stores the integers itself where it has a variant for it -- s16: every Mode I coded-bits chain that is one frame kernel
 * (any filter the fused FIRFilter takes, any gain mode, with or without crest-factor reduction; a windowed guard interval
 * without FIRFilter, or with it up to 10 samples of overlap), and the x2 / x4 resampler with or without the polynomial
 * predistorter; u8 / s8: Mode I coded-bits chain ending in the guard interval or in a filter of up to the default length whose
 * boundary outputs come through the taps' inverse (the default), also with up to 10 samples of OFDM windowing -- half / a
 * quarter of the bytes written and copied to the host; every other combination converts in a kernel of its own.  Output sizes of
 * dabgpu_chain_out_bytes_per_frame / _process / _submit follow the format.  dabgpu_get_num_clipped: the number of
 * clipped components of the most recent chain call (FormatConverter::get_num_clipped_samples, :56-59), after
 * waiting for that call -- or, on the asynchronous path, of the batch dabgpu_chain_collect returned last. */
DABGPU_API int dabgpu_set_output_format(dabgpu_ctx *ctx, int format);
DABGPU_API int dabgpu_get_num_clipped(dabgpu_ctx *ctx, size_t *num_clipped);

/* bytes of IQ produced per transmission frame for a stage mask */
DABGPU_API size_t dabgpu_chain_out_bytes_per_frame(const dabgpu_ctx *ctx, unsigned stage_mask);

/* cifPart output (n_frames x tf_input_bytes) -> IQ.  Replaces the sub-graph
 * cifMap .. cifGuard/cifFilter/cifRes/cifPoly of src/DabModulator.cpp:385-419
 * with ONE plugin.  Frames are consecutive frames of one stream (the resampler
 * state carries from frame to frame and from call to call). */
DABGPU_API int dabgpu_chain_process(dabgpu_ctx *ctx, const uint8_t *bits, size_t n_frames,
                                    unsigned stage_mask, void *iq_out, size_t out_cap,
                                    size_t *out_bytes);
/* same, device-resident input and output, asynchronous on `stream` */
DABGPU_API int dabgpu_chain_process_dev(dabgpu_ctx *ctx, const void *d_bits, size_t n_frames,
                                        unsigned stage_mask, void *d_iq, size_t out_cap,
                                        size_t *out_bytes, void *stream);
/* SignalMultiplexer output ((nb_symbols+1) x carriers cf32 per frame) -> IQ:
 * the OfdmGenerator[+GainControl][+Guard][+FIR..] part of the chain. */
DABGPU_API int dabgpu_symbols_process_dev(dabgpu_ctx *ctx, const void *d_carriers,
                                          size_t n_frames, unsigned stage_mask, void *d_iq,
                                          size_t out_cap, size_t *out_bytes, void *stream);

/* Resampler::process -> MemlessPoly::internal_process (cifRes -> cifPoly, src/DabModulator.cpp:403-406) on a native-rate
 * stream that is already in device memory: the tail of the chain by itself.  stage_mask = DABGPU_STAGE_RESAMPLE and / or
 * DABGPU_STAGE_POLY; n_samples complex samples in (a whole number of resampler hops), n_samples * L / M out.  Stateful like
 * the Resampler (the context's halo), asynchronous on `stream`. */
DABGPU_API int dabgpu_post_process_dev(dabgpu_ctx *ctx, const void *d_native, size_t n_samples, unsigned stage_mask,
                                       void *d_iq, size_t out_cap, size_t *out_bytes, void *stream);

/* ---- batches in flight inside one context --------------------------------- *
 * The reference overlaps its stages by handing frame i + 1 to a stage while frame i is still inside it
 * (PipelinedModCodec, src/ModPlugin.cpp:90-154).  The counterpart here: a chain call on the context's OWN stream
 * (dabgpu_chain_process_dev / dabgpu_symbols_process_dev with stream == NULL, and the two batches of dabgpu_chain_submit)
 * goes to one of `lanes` internal HIP streams in turn, each with its own scratch, so that the kernels of consecutive
 * calls overlap where one launch alone cannot fill the chip (frames are independent units).  Consequences for the caller:
 *   - outputs of calls on the context's own stream are complete after dabgpu_synchronize (all lanes), or, in stream
 *     order, for work queued on `stream` after dabgpu_stream_wait_for(ctx, stream);
 *   - inputs produced on a stream of the caller's are ordered in front with dabgpu_wait_for_stream(ctx, stream);
 *   - calls that carry stream state (DABGPU_STAGE_RESAMPLE at a ratio other than 1) and batches of more than 2048 frames
 *     stay on lane 0, in call order;
 *   - a call with an explicit stream argument is what it always was: asynchronous on that stream, the context's scratch.
 *     Do not mix the two on one context without a dabgpu_synchronize in between.
 * dabgpu_set_lanes: 1 ... 4 (default 3: measured best at 1 ... 64 frames per call, tools/experiments/exp_r05.py lanes; 1 = every call on the one context stream, in order).  Waits for the context. */
DABGPU_API int dabgpu_set_lanes(dabgpu_ctx *ctx, int lanes);
/* Diagnostic: how many lanes exist so far (lane 0 = the context's stream, the others are created on first use), and in
 * *own_queue_mask, bit i: lane i was found a hardware queue of its own.  (The HIP runtime multiplexes streams onto a few
 * hardware queues, four by default -- GPU_MAX_HW_QUEUES --, and streams that share one run in order; which queue a new
 * stream joins depends on the process's history, so a lane's stream is probed against the lanes before it and replaced
 * until it overlaps with all of them.  A process that keeps more busy streams of its own than the device has hardware
 * queues left cannot be given that.) */
DABGPU_API int dabgpu_debug_lanes(dabgpu_ctx *ctx, int *own_queue_mask);
/* everything the context queues from now on starts after what `stream` holds now */
DABGPU_API int dabgpu_wait_for_stream(dabgpu_ctx *ctx, void *stream);
/* everything queued on `stream` from now on starts after what the context has queued so far, on every lane */
DABGPU_API int dabgpu_stream_wait_for(dabgpu_ctx *ctx, void *stream);

/* The hand-over of the native-rate stream from FIRFilter to Resampler (src/DabModulator.cpp:403-406) inside the fused
 * chain: in pieces of `frames` transmission frames through a two-piece ring (2 x frames x 1.57 MB: sized to stay in
 * the 256 MiB last-level cache), the producer of piece i + 1 on a second internal stream while the x2 / x4 resampler works on
 * piece i.  0 = one piece: the whole batch goes through memory between the two kernels.  `frames` is even (TII frame
 * parity).  Same samples either way (the resampler's state runs through the pieces).  Waits for the context. */
DABGPU_API int dabgpu_set_handover_frames(dabgpu_ctx *ctx, int frames);

/* Asynchronous host path: the streaming shape of dabgpu_chain_process.  submit() stages the coded
 * bits in pinned memory and queues upload, kernels and -- on a second HIP stream -- the copy back
 * into a pinned buffer owned by the context; up to TWO batches may be in flight, so the copy of
 * batch i overlaps the kernels of batch i+1 (a third submit is DABGPU_E_CAPACITY).  collect() waits
 * for the OLDEST batch and hands out its buffer: *iq stays valid until the second next submit.
 * Same settings snapshot, stream state (resampler halo, TII parity) and ordering as the synchronous
 * call; do not mix the two on one context while batches are in flight. */
DABGPU_API int dabgpu_chain_submit(dabgpu_ctx *ctx, const uint8_t *bits, size_t n_frames,
                                   unsigned stage_mask);
DABGPU_API int dabgpu_chain_collect(dabgpu_ctx *ctx, const void **iq, size_t *out_bytes);

/* Which kernels did the most recent chain call (dabgpu_chain_process / _process_dev / _submit, dabgpu_symbols_process_dev)
 * launch?  A "; "-separated list of kernel names in launch order, the frame kernel with the VALUES of its template arguments
 * ("tf_kernel<logn=11 bits=1 gain=1 guard=1 fir=1 nt=45 cfr=0 gvar=0 zonly=0 ofmt=0 win=0 eq=1>").  The fused chain picks
 * one of ~120 instantiations from the settings (mode, gain mode, filter length, windowing, CFR, TII, output format): this
 * makes the choice observable, so that a test can walk the whole matrix (tests/test_dispatch_matrix.py) and a user can see
 * what a configuration costs.  Diagnostic: not part of the reference's interface. */
DABGPU_API int dabgpu_debug_last_variant(dabgpu_ctx *ctx, char *buf, size_t cap);
/* The trace is OFF by default (a launch then costs one pointer test); dabgpu_debug_trace(ctx, 1) turns it on for the chain
 * calls that follow.  Both calls belong to the thread that issues the chain calls (the trace is not guarded by the settings
 * mutex). */
DABGPU_API int dabgpu_debug_trace(dabgpu_ctx *ctx, int enable);

/* wait for everything queued on the context's own stream(s): every lane */
DABGPU_API int dabgpu_synchronize(dabgpu_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
