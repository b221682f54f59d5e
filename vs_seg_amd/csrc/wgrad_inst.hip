// One translation unit per element type: the (MAXT, NTP, HG) instantiations of wgrad_kernel for that type (the Makefile builds the
// two in parallel).  Built with -DWG_T=<float|bf16_t> -DWG_TNAME=<f32|bf16>.
#define WG_INST
#include "wgrad.hip"
#define WG_CAT2(a, b) a##b
#define WG_CAT(a, b) WG_CAT2(a, b)
int WG_CAT(vsseg_wgrad_launch_, WG_TNAME)(WgradK& k, int maxt, int hg, dim3& grid, int lds, hipStream_t s) { return wg_maxt<WG_T>(k, maxt, hg, grid, lds, s); }
