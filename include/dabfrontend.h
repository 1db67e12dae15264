/* dabfrontend.h -- C view of the CPU front-end (odr-dabmod_amd/host/Frontend.h) for
 * tests and non-C++ callers.  The product interface is the C++ one (reference class names); these
 * entry points run one class each, or the whole sub-graph of src/DabModulator.cpp:131-139,281-385.
 * All return >= 0 on success (a byte / block / rule count) and -1 when the class throws.
 */
#ifndef DABFRONTEND_H
#define DABFRONTEND_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DABFE_API __attribute__((visibility("default")))

/* PrbsGenerator(framesize, 0x110), src/PrbsGenerator.cpp:130-188; in may be NULL (padding source) */
DABFE_API int dabfe_prbs(size_t framesize, const uint8_t *in, uint8_t *out);
/* ConvEncoder(framesize), src/ConvEncoder.cpp:59-150; out holds 4 * framesize + 3 bytes */
DABFE_API int dabfe_conv_encode(const uint8_t *in, size_t framesize, uint8_t *out);
/* SubchannelSource(0, stl, tpl): rules[2i] = length, rules[2i+1] = pattern (up to 8 rules),
 * src/SubchannelSource.cpp:70-643; *framesize_cu, *bitrate as :657-1017 */
DABFE_API int dabfe_subchannel_profile(unsigned stl, unsigned tpl, uint32_t *rules, size_t *framesize_cu,
                                       size_t *bitrate);
/* PuncturingEncoder configured as DabModulator does for the FIC (is_fic, mid) or for a sub-channel
 * (stl, tpl), tail rule (3, 0xcccccc) appended; src/PuncturingEncoder.cpp:102-210 */
DABFE_API int dabfe_puncture(const uint8_t *in, size_t in_len, unsigned stl, unsigned tpl, int is_fic, unsigned mid,
                             uint8_t *out);
/* nframes frames through ONE TimeInterleaver(framesize), src/TimeInterleaver.cpp:51-96 */
DABFE_API int dabfe_time_interleave(const uint8_t *in, size_t framesize, size_t nframes, uint8_t *out);
/* nframes raw ETI(NI) frames of 6144 bytes -> one BlockPartitioner block per completed transmission
 * frame (Mode I: 28 800 bytes = the hot path's input); returns the number of blocks */
DABFE_API int dabfe_eti_frontend(const uint8_t *eti, size_t nframes, unsigned mode, uint8_t *out, size_t out_cap);

/* A raw ETI(NI) byte stream through ONE EtiReader in pieces of `piece` bytes (the reader accepts input cut anywhere,
 * src/EtiReader.cpp:93-284): the frame counters (FCT) of the headers it parsed, in order, into fct_out; *n_errors = calls that
 * threw (a refused header), *n_short = calls that consumed less than they were given.  Returns the number of headers. */
DABFE_API int dabfe_eti_reader_stream(const uint8_t *bytes, size_t n, size_t piece, unsigned *fct_out, size_t fct_cap,
                                      size_t *n_errors, size_t *n_short);

#ifdef __cplusplus
}
#endif
#endif
