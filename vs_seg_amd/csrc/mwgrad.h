// Marching (x-ring) variant of vsseg_wgrad for the HBM-bound stride-1 3x3x1 bf16 weight gradients: mwgrad.hip (descriptors with march = 1).
#pragma once
#include "common.h"
int vsseg_mwgrad_launch(const vsseg_wgrad_desc* d, const void* zeros, hipStream_t s);
