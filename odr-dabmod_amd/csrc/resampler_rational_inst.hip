// resampler_rational_inst.hip -- instantiates the general-ratio resampler kernels for ONE input FFT size (RS_LOGNIN);
// compiled once per size by the Makefile.
#include "device_common.h"
#include "resampler_rational.h"

#if !defined(RS_LOGNIN)
#error "compile with -DRS_LOGNIN=<9..12>"
#endif
#define RS_CAT2(a) launch_resampler_rational_##a
#define RS_CAT(a) RS_CAT2(a)

namespace dabgpu {
hipError_t RS_CAT(RS_LOGNIN)(const ResamplerArgs &a, hipStream_t s) { return launch_resampler_rational_n<RS_LOGNIN>(a, s); }
}  // namespace dabgpu
