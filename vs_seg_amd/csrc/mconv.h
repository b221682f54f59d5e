// Marching (x-ring) variant of vsseg_igemm for the HBM-bound stride-1 3x3x1 bf16 launches: mconv.hip (launch plans with depth -5).
#pragma once
#include "common.h"
// LDS bytes of the launch, or VSSEG_EINVAL (with vsseg_last_error set to the reason) when the descriptor is outside the kernel's domain.
int vsseg_mconv_lds_bytes(const vsseg_igemm_desc* d);
int vsseg_mconv_launch(const vsseg_igemm_desc* d, const void* zeros, hipStream_t s);
