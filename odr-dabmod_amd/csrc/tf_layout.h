// tf_layout.h -- LDS layout constants of the frame kernel (tf_kernel.h) that its host-side size computation
// (tf_launch.hip: tf_lds_bytes) has to agree with.
#pragma once
#include "dabgpu_internal.h"

namespace dabgpu {
// LDS of the equalised-boundary variants (see tf_kernel), in complex slots: two windows of the previous filtered symbol and
// their difference against the current one (kEqWLen each), the unfiltered differences d (kEqDLen), the inverse filter
// (kEqTaps + 8 floats), the raised-cosine factors of the windowed form (2 kEqWinMax floats)
constexpr int kEqWinMax = 10;      // widest overlap the equalised-boundary variant windows itself: 2 x 10 + 44 = 64 boundary outputs
constexpr int kEqWLen = 240, kEqDLen = 80;
constexpr int kEqElems = 3 * kEqWLen + kEqDLen + (kEqTaps + 8) / 2 + 16;
constexpr int kWinMax = 128;       // widest raised-cosine overlap the frame kernel applies itself (TF_WINDOW)
constexpr int kBnd = 128;          // LDS slots per boundary buffer; the fused FIR handles ntaps <= kBnd
}  // namespace dabgpu
