// tf_layout.h -- LDS layout constants of the frame kernel (tf_kernel.h) that its host-side size computation
// (tf_launch.hip: tf_lds_bytes) has to agree with.
#pragma once
#include "dabgpu_internal.h"

namespace dabgpu {
constexpr int kEqElems = 3 * 208 + 48 + (kEqTaps + 8) / 2;   // cf slots of LDS the EQ variant keeps (see tf_kernel)
constexpr int kWinMax = 128;       // widest raised-cosine overlap the frame kernel applies itself (TF_WINDOW)
constexpr int kBnd = 128;          // LDS slots per boundary buffer; the fused FIR handles ntaps <= kBnd
}  // namespace dabgpu
