// Fused BatchNorm-backward-on-load + data gradient + weight gradient of a stride-1 3x3x1 Convolution block: mbwd.hip (vsseg_conv_bwd_fused).
#pragma once
#include "common.h"
