// Chained marching convolution (two stride-1 3x3x1 convolutions, the tensor between them in LDS; eval only): chain.hip, entry point vsseg_conv_chain.
#pragma once
#include "common.h"
const void* vsseg_zero_page();  // igemm.hip: 256 zero bytes on the device (source of the padding pieces of an LDS-DMA fetch), or nullptr
