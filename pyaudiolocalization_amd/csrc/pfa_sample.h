// pfa_sample.h - tile constants of the prime-factor column pass, shared by pfa_kernels.h (k_pfa_cols), pfa_cols_stats.h
// (the fused column pass + peak statistics) and pfa_forward.h.
#pragma once
#include "fft_core.h"

namespace pal {

constexpr int kPfaTC = 11;   // output indices t per wavefront of the column pass (x 4 accumulators each)
constexpr int kPfaUnr = 4;   // steps j per loop iteration of the column pass (the table is padded with kPfaUnr zero rows)

}  // namespace pal
