// Deep-level variant of vsseg_igemm for the SMALL launches of levels 3-5 of the 2.5D U-Net and the stride-2 transitions around them: dconv.hip (launch plans with depth -7).
#pragma once
#include "common.h"
// LDS bytes of the launch, or VSSEG_EINVAL (with vsseg_last_error set to the reason) when the descriptor is outside the kernel's domain.
int vsseg_dconv_lds_bytes(const vsseg_igemm_desc* d);
int vsseg_dconv_launch(const vsseg_igemm_desc* d, const void* zeros, hipStream_t s);
