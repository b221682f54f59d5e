// Gaussian-weighted sliding-window blend (MONAI 0.4.0 sliding_window_inference steps 6-7, SURVEY.md App. B;
// call site ref:params/VSparams.py:568-574).  Windows are blended one after the other in the reference's window
// order, each voxel doing the same `out += map*seg; cnt += map` fp32 sequence as the reference, so the result does
// not depend on how windows were distributed over GPUs.
#include "common.h"

__global__ void swi_accumulate_kernel(const float* __restrict__ seg, const float* __restrict__ imap, int rx, int ry, int rz, int sx, int sy, int sz, int c,
                                      float* __restrict__ out, float* __restrict__ cnt, int px, int py, int pz) {
  const int64_t total = (int64_t)rx * ry * rz;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int z = (int)(i % rz);
    int64_t r = i / rz;
    int y = (int)(r % ry), x = (int)(r / ry);
    const int64_t o = (((int64_t)(sx + x)) * py + (sy + y)) * pz + (sz + z);
    const float w = imap[i];
    if (seg != nullptr)
      for (int k = 0; k < c; ++k) out[o * c + k] += w * seg[i * c + k];
    if (cnt != nullptr) cnt[o] += w;
  }
}
extern "C" int vsseg_swi_accumulate(const float* seg, const float* imap, const int32_t roi[3], const int32_t start[3], int32_t c, float* out, float* cnt, const int32_t pdims[3], void* stream) {
  VSSEG_CHECK(imap && ((seg && out) || cnt) && (!seg || out) && c >= 1, "vsseg_swi_accumulate: bad arguments");  // seg == NULL: only the weight map; cnt == NULL: only the blend
  for (int a = 0; a < 3; ++a) VSSEG_CHECK(start[a] >= 0 && start[a] + roi[a] <= pdims[a], "vsseg_swi_accumulate: window outside the padded volume (dim %d)", a);
  int64_t total = (int64_t)roi[0] * roi[1] * roi[2];
  hipLaunchKernelGGL(swi_accumulate_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), seg, imap, roi[0], roi[1], roi[2], start[0], start[1], start[2], c, out, cnt, pdims[0], pdims[1], pdims[2]);
  VSSEG_LAUNCH_CHECK("vsseg_swi_accumulate");
  return VSSEG_OK;
}

__global__ void swi_finalize_kernel(const float* __restrict__ out, const float* __restrict__ cnt, int px, int py, int pz, int bx, int by, int bz, int dx, int dy, int dz, int c, float* __restrict__ dst) {
  const int64_t total = (int64_t)dx * dy * dz;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int z = (int)(i % dz);
    int64_t r = i / dz;
    int y = (int)(r % dy), x = (int)(r / dy);
    const int64_t o = (((int64_t)(bx + x)) * py + (by + y)) * pz + (bz + z);
    const float w = cnt[o];
    for (int k = 0; k < c; ++k) dst[i * c + k] = out[o * c + k] / w;
  }
}
extern "C" int vsseg_swi_finalize(const float* out, const float* cnt, const int32_t pdims[3], const int32_t pad_before[3], const int32_t dims[3], int32_t c, float* dst, void* stream) {
  VSSEG_CHECK(out && cnt && dst && c >= 1, "vsseg_swi_finalize: bad arguments");
  int64_t total = (int64_t)dims[0] * dims[1] * dims[2];
  hipLaunchKernelGGL(swi_finalize_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), out, cnt, pdims[0], pdims[1], pdims[2], pad_before[0], pad_before[1], pad_before[2], dims[0], dims[1], dims[2], c, dst);
  VSSEG_LAUNCH_CHECK("vsseg_swi_finalize");
  return VSSEG_OK;
}
