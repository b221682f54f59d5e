// Data-side kernels: the random tail of the reference's MONAI transform chain and the intensity
// normalisation, on volumes cached in HBM, so that no CPU DataLoader worker touches the 25 MB/sample tensors
// (ref:params/VSparams.py:205-245: NormalizeIntensityd, SpatialPadd, RandFlipd(spatial_axis=0), RandSpatialCropd).
#include "common.h"

// dst[b][x][y][z] = vol_b[flip ? X-1-(sx+x) : sx+x][sy+y][sz+z], 0 outside the volume (= SpatialPadd's constant padding).
// One launch crops image and label of a whole batch: `srcs` holds 2*n device pointers (image_0, label_0, image_1, ...).
__global__ void crop_flip_kernel(const vsseg_crop_job* __restrict__ jobs, float* __restrict__ dst, int rx, int ry, int rz) {
  const vsseg_crop_job j = jobs[blockIdx.y];
  const int64_t per = (int64_t)rx * ry * rz;
  float* out = dst + (int64_t)blockIdx.y * per;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < per; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i;
    const int z = (int)(r % rz); r /= rz;
    const int y = (int)(r % ry);
    const int x = (int)(r / ry);
    int gx = x + j.origin[0];
    const int gy = y + j.origin[1], gz = z + j.origin[2];
    if (j.flip_x) gx = j.sdims[0] - 1 - gx;  // RandFlipd acts on the padded volume, before the crop
    float v = 0.f;
    if ((unsigned)gx < (unsigned)j.sdims[0] && (unsigned)gy < (unsigned)j.sdims[1] && (unsigned)gz < (unsigned)j.sdims[2]) v = j.src[((int64_t)gx * j.sdims[1] + gy) * j.sdims[2] + gz];
    out[i] = v;
  }
}
extern "C" int vsseg_crop_flip(const void* jobs, int32_t njobs, float* dst, const int32_t roi[3], void* stream) {
  VSSEG_CHECK(jobs && dst && njobs >= 1 && roi[0] > 0 && roi[1] > 0 && roi[2] > 0, "vsseg_crop_flip: bad arguments");
  const int64_t per = (int64_t)roi[0] * roi[1] * roi[2];
  dim3 grid(grid_for(per, 256, 2048), njobs);
  hipLaunchKernelGGL(crop_flip_kernel, grid, dim3(256), 0, as_stream(stream), (const vsseg_crop_job*)jobs, dst, roi[0], roi[1], roi[2]);
  VSSEG_LAUNCH_CHECK("vsseg_crop_flip");
  return VSSEG_OK;
}

// NormalizeIntensityd: (x - mean) / std over the whole image, population std, no division when std == 0.
// Pass 1: fp64 sum / sum of squares (sharded atomics); pass 2 applies.  acc = 2 doubles, zeroed by the caller.
__global__ void intensity_sums_kernel(const float* __restrict__ x, int64_t n, double* __restrict__ acc) {
  double s = 0.0, q = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double v = x[i];
    s += v;
    q += v * v;
  }
  s = wave_sum_d(s);
  q = wave_sum_d(q);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&acc[0], s);
    atomicAdd(&acc[1], q);
  }
}
__global__ void intensity_apply_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, const double* __restrict__ acc) {
  const double mean = acc[0] / (double)n;
  double var = acc[1] / (double)n - mean * mean;
  if (var < 0.0) var = 0.0;
  const double sd = sqrt(var);
  const float m = (float)mean, inv = sd == 0.0 ? 1.f : (float)(1.0 / sd);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = (x[i] - m) * inv;
}
extern "C" int vsseg_normalize_intensity(const float* x, float* y, int64_t n, double* acc2, void* stream) {
  VSSEG_CHECK(x && y && acc2 && n > 0, "vsseg_normalize_intensity: bad arguments");
  hipStream_t s = as_stream(stream);
  if (hipMemsetAsync(acc2, 0, 2 * sizeof(double), s) != hipSuccess) { vsseg_set_error("vsseg_normalize_intensity: memset failed"); return VSSEG_ELAUNCH; }
  hipLaunchKernelGGL(intensity_sums_kernel, dim3(grid_for(n, 256, 1024)), dim3(256), 0, s, x, n, acc2);
  hipLaunchKernelGGL(intensity_apply_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, s, x, y, n, acc2);
  VSSEG_LAUNCH_CHECK("vsseg_normalize_intensity");
  return VSSEG_OK;
}
