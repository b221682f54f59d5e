// One translation unit per (element type, NT): the 9 (MTW, MODE) instantiations of igemm_kernel for that pair, so that the
// Makefile can build the 108 kernel variants in parallel.  Built with -DIG_T=<float|bf16_t> -DIG_TNAME=<f32|bf16> -DIG_NT=<1..6>.
#include "igemm_kernel.h"
#define IG_CAT2(a, b, c) a##b##_##c
#define IG_CAT(a, b, c) IG_CAT2(a, b, c)
int IG_CAT(vsseg_igemm_launch_, IG_TNAME, IG_NT)(const IgemmK& k, dim3 grid, int lds, hipStream_t s) { return launch_mtw<IG_T, IG_NT>(k, grid, lds, s); }
