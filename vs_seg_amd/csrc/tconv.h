// Transition kernel: all eight output-parity classes of a 3x3x3 stride-(2,2,2) transposed convolution / strided data gradient between the 96x32x128 and 48x16x64 levels
// as ONE launch (tconv.hip, launch plans with depth -8).
#pragma once
#include "common.h"
// LDS bytes of the launch, or VSSEG_EINVAL (with vsseg_last_error set to the reason) when the descriptor is outside the kernel's domain.
int vsseg_tconv_lds_bytes(const vsseg_igemm_desc* d);
int vsseg_tconv_launch(const vsseg_igemm_desc* d, const void* zeros, hipStream_t s);
