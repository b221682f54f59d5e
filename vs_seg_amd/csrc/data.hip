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

// ------------------------------------------------------------------------------------------------------------
// Convolutions of the ONE-channel network input (first encoder block: model.0.conv.unit0 3x3x1 and model.0.residual 1x1x1,
// ref:params/networks/blocks/convolutions.py:114-146, 241-250 with in_channels = 1).  On the MFMA path they need the input
// zero-extended to an 8-channel K-group (7/8 of the input bytes and of the K work are zeros); as a direct stencil they are a
// streaming kernel: each thread reads the (kx x ky) neighbourhood of 4 z-consecutive voxels (8-byte / 16-byte loads of the
// compact 1-channel tensor) and writes 8 output channels for each of them.
//   v = bias[c] + sum_t w[c][t] * x[voxel + off_t]          (zero padding, "same")
//   STATS: per-channel sum / sum of squares of v into the sharded fp64 statistics (training BatchNorm), out = v
//   else : out = act(v * scale[c] + shift[c])               (eval: BatchNorm folded; PReLU when alpha != NULL)
// ------------------------------------------------------------------------------------------------------------
template <typename T, bool STATS, int TAPS>
__global__ __launch_bounds__(256, 3) void conv1ch_fwd_kernel(const T* __restrict__ x, int N, int X, int Y, int Z, const float* __restrict__ w, const float* __restrict__ bias,
                                                             const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ alpha_p, T* __restrict__ out, int op, int cgs,
                                                             double* __restrict__ stats, int stats_stride) {
  // thread = (4 z-consecutive voxels, 4 output channels): 36 weights + 16 accumulators in registers (8 channels per thread needed
  // ~230 VGPRs); TAPS = 9 (3x3x1) or 1 (1x1x1) is a compile-time constant so that the weight array is never indexed dynamically
  extern __shared__ float red[];  // STATS: [2][C]
  const int C = cgs * 4;
  if (STATS) {
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) red[i] = 0.f;
    __syncthreads();
  }
  const int64_t gt = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, nthreads = (int64_t)gridDim.x * blockDim.x;
  const int cg = (int)(gt % cgs), c = cg * 4;
  float wr[4][TAPS], b4[4], sc[4], sh[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int t = 0; t < TAPS; ++t) wr[j][t] = w[(c + j) * TAPS + t];
    b4[j] = bias ? bias[c + j] : 0.f;
    sc[j] = scale ? scale[c + j] : 1.f;
    sh[j] = scale ? shift[c + j] : 0.f;
  }
  const bool prelu = alpha_p != nullptr;
  const float alpha = prelu ? *alpha_p : 1.f;
  const int Z4 = Z >> 2;
  const int64_t nquads = (int64_t)N * X * Y * Z4;
  float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  // 32-bit index arithmetic (the host checks that the quad count fits): six 64-bit divisions per iteration cost more than the stencil
  const unsigned qstep = (unsigned)(nthreads / cgs);
  for (unsigned q = (unsigned)(gt / cgs); q < (unsigned)nquads; q += qstep) {
    unsigned r = q;
    const int z4 = (int)(r % (unsigned)Z4); r /= (unsigned)Z4;
    const int y = (int)(r % (unsigned)Y); r /= (unsigned)Y;
    const int xx = (int)(r % (unsigned)X);
    const int n = (int)(r / (unsigned)X);
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = b4[j];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      const int dx = t / 3 - 1, dy = t % 3 - 1;  // 3x3 layout; a 1x1 kernel has its only tap at the centre
      const int gx = TAPS == 1 ? xx : xx + dx, gy = TAPS == 1 ? y : y + dy;
      float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((unsigned)gx < (unsigned)X && (unsigned)gy < (unsigned)Y) v4 = ld4(x + (((int64_t)n * X + gx) * Y + gy) * Z + z4 * 4);
      const float xv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += wr[j][t] * xv[i];
    }
    const int64_t v0 = (((int64_t)n * X + xx) * Y + y) * Z + z4 * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a = acc[i][j];
        if (STATS) { s1[j] += a; s2[j] += a * a; }
        else {
          a = a * sc[j] + sh[j];
          if (prelu) a = a > 0.f ? a : alpha * a;
        }
        o[j] = a;
      }
      st4(out + (v0 + i) * op + c, make_float4(o[0], o[1], o[2], o[3]));
    }
  }
  if (STATS) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { atomicAdd(&red[c + j], s1[j]); atomicAdd(&red[C + c + j], s2[j]); }
    __syncthreads();
    double* st = stats + (int64_t)(blockIdx.x % VSSEG_STAT_SHARDS) * 2 * stats_stride;
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) atomicAdd(&st[(i / C) * stats_stride + (i % C)], (double)red[i]);
  }
}
static inline int conv1ch_block(int cgs) {  // block size = multiple of the channel-group count (fixed group per thread)
  int blk = 256;
  while (blk % cgs) blk -= 64;
  return blk > 0 ? blk : 0;
}
extern "C" int vsseg_conv1ch_fwd(const void* x1, int32_t dtype, int32_t n, const int32_t dims[3], const float* w, const float* bias, const int32_t kernel[3], const float* scale, const float* shift,
                                 const float* alpha, vsseg_tensor out, double* stats, int32_t stats_stride, void* stream) {
  VSSEG_CHECK(x1 && w && out.ptr && !out.ptr2 && out.dtype == dtype && out.c % 8 == 0 && out.pitch % 8 == 0 && out.n == n && out.x == dims[0] && out.y == dims[1] && out.z == dims[2],
              "vsseg_conv1ch_fwd: bad arguments");
  VSSEG_CHECK(kernel[2] == 1 && ((kernel[0] == 3 && kernel[1] == 3) || (kernel[0] == 1 && kernel[1] == 1)) && dims[2] % 4 == 0, "vsseg_conv1ch_fwd: kernel must be 3x3x1 or 1x1x1 and Z a multiple of 4");
  VSSEG_CHECK(!stats || (!scale && !alpha && stats_stride >= out.c), "vsseg_conv1ch_fwd: statistics mode takes no affine / activation");
  const int cgs = out.c / 4, blk = conv1ch_block(cgs);  // 4 output channels per thread
  VSSEG_CHECK((int64_t)n * dims[0] * dims[1] * (dims[2] / 4) < (1ll << 31) - (1ll << 24), "vsseg_conv1ch_fwd: tensor too large for 32-bit voxel indices");
  VSSEG_CHECK(blk > 0, "vsseg_conv1ch_fwd: unsupported channel count %d", out.c);
  const int64_t work = (int64_t)n * dims[0] * dims[1] * (dims[2] / 4) * cgs;
  dim3 g(grid_for(work, blk, 256 * 16)), b(blk);
  hipStream_t s = as_stream(stream);
#define VSSEG_C1_LAUNCH(TP) \
  if (dtype == VSSEG_F32) { \
    if (stats) hipLaunchKernelGGL((conv1ch_fwd_kernel<float, true, TP>), g, b, 2 * out.c * sizeof(float), s, (const float*)x1, n, dims[0], dims[1], dims[2], w, bias, scale, shift, alpha, (float*)out.ptr, out.pitch, cgs, stats, stats_stride); \
    else hipLaunchKernelGGL((conv1ch_fwd_kernel<float, false, TP>), g, b, 0, s, (const float*)x1, n, dims[0], dims[1], dims[2], w, bias, scale, shift, alpha, (float*)out.ptr, out.pitch, cgs, stats, stats_stride); \
  } else { \
    if (stats) hipLaunchKernelGGL((conv1ch_fwd_kernel<bf16_t, true, TP>), g, b, 2 * out.c * sizeof(float), s, (const bf16_t*)x1, n, dims[0], dims[1], dims[2], w, bias, scale, shift, alpha, (bf16_t*)out.ptr, out.pitch, cgs, stats, stats_stride); \
    else hipLaunchKernelGGL((conv1ch_fwd_kernel<bf16_t, false, TP>), g, b, 0, s, (const bf16_t*)x1, n, dims[0], dims[1], dims[2], w, bias, scale, shift, alpha, (bf16_t*)out.ptr, out.pitch, cgs, stats, stats_stride); \
  }
  if (kernel[0] == 3) { VSSEG_C1_LAUNCH(9) } else { VSSEG_C1_LAUNCH(1) }
#undef VSSEG_C1_LAUNCH
  VSSEG_LAUNCH_CHECK("vsseg_conv1ch_fwd");
  return VSSEG_OK;
}
