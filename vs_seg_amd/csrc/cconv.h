// Compute-bound (compile-time geometry) variant of vsseg_igemm for the MFMA-bound stride-1 3x3x3 bf16 launches: cconv.hip.
#pragma once
#include "common.h"
// LDS bytes of the launch, or VSSEG_EINVAL (with vsseg_last_error set to the reason) when the descriptor is outside the kernel's domain.
int vsseg_cconv_lds_bytes(const vsseg_igemm_desc* d);
int vsseg_cconv_launch(const vsseg_igemm_desc* d, const void* zeros, hipStream_t s);
