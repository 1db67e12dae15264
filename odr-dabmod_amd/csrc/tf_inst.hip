// tf_inst.hip -- instantiates the frame kernel's variants for ONE transmission mode (TF_LOGN: FFT size 2^TF_LOGN) and ONE
// compile-time FIR length (TF_NT: 45 = the default filter, 0 = read it from the arguments).  The Makefile compiles this
// file once per pair, so that the ~120 instantiations of tf_kernel build in parallel.
#include "tf_kernel.h"

#if !defined(TF_LOGN) || !defined(TF_NT)
#error "compile with -DTF_LOGN=<8..11> -DTF_NT=<0|45>"
#endif
#define TF_CAT2(a, b, c) launch_tf_##a##_##b
#define TF_CAT(a, b) TF_CAT2(a, b, )

namespace dabgpu {
#if TF_LOGN != 11 && TF_NT == 45
hipError_t TF_CAT(TF_LOGN, TF_NT)(const TfArgs &a, unsigned flags, hipStream_t s) { return launch_tf_small45<TF_LOGN>(a, flags, s); }
#else
hipError_t TF_CAT(TF_LOGN, TF_NT)(const TfArgs &a, unsigned flags, hipStream_t s) { return launch_tf_n<TF_LOGN, TF_NT>(a, flags, s); }
#endif
}  // namespace dabgpu
