// Gathering marching variant of vsseg_igemm for the stride-(2,2,1) 3x3x1 launches that read a fine tensor and write a coarse one (strided convolutions, data gradients of the
// transposed convolutions between levels 0-2): gconv.hip (launch plans with depth -9).
#pragma once
#include "common.h"
// LDS bytes of the launch, or VSSEG_EINVAL (with vsseg_last_error set to the reason) when the descriptor is outside the kernel's domain.
int vsseg_gconv_lds_bytes(const vsseg_igemm_desc* d);
int vsseg_gconv_launch(const vsseg_igemm_desc* d, const void* zeros, hipStream_t s);
