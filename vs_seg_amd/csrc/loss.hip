// Dice_spvPA loss (ref:params/losses/dice_spvPA.py:90-167, 238-297) as fused reductions + one elementwise backward,
// and the hard-Dice metric (ref:params/VSparams.py:393-408).  Everything here is HBM-bound: logits are read once
// forward and once backward; the ~25 ATen temporaries of the reference never exist.
#include "common.h"

#define SMOOTH 1e-5

// label pyramid for the attention supervision: G_{l+1} = MaxPool3d(ratio)(G_l)   (ref :268-277)
__global__ void maxpool_label_kernel(const float* __restrict__ src, int n, int sx, int sy, int sz, int rx, int ry, int rz, float* __restrict__ dst) {
  const int dx = sx / rx, dy = sy / ry, dz = sz / rz;
  const int64_t total = (int64_t)n * dx * dy * dz;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i;
    int z = (int)(r % dz); r /= dz;
    int y = (int)(r % dy); r /= dy;
    int x = (int)(r % dx);
    int64_t b = r / dx;
    float m = -INFINITY;
    for (int a = 0; a < rx; ++a)
      for (int c = 0; c < ry; ++c)
        for (int e = 0; e < rz; ++e) m = fmaxf(m, src[((b * sx + x * rx + a) * sy + y * ry + c) * (int64_t)sz + z * rz + e]);
    dst[i] = m;
  }
}
extern "C" int vsseg_maxpool_label(const float* src, int32_t n, const int32_t sdims[3], const int32_t ratio[3], float* dst, void* stream) {
  VSSEG_CHECK(src && dst && ratio[0] >= 1 && ratio[1] >= 1 && ratio[2] >= 1, "vsseg_maxpool_label: bad arguments");
  VSSEG_CHECK(sdims[0] % ratio[0] == 0 && sdims[1] % ratio[1] == 0 && sdims[2] % ratio[2] == 0, "vsseg_maxpool_label: attention pyramid shapes must divide (ref dice_spvPA.py:273)");
  int64_t total = (int64_t)n * (sdims[0] / ratio[0]) * (sdims[1] / ratio[1]) * (sdims[2] / ratio[2]);
  hipLaunchKernelGGL(maxpool_label_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), src, n, sdims[0], sdims[1], sdims[2], ratio[0], ratio[1], ratio[2], dst);
  VSSEG_LAUNCH_CHECK("vsseg_maxpool_label");
  return VSSEG_OK;
}

// `fx` > 0: the block's sums are added as fixed-point integers of that scale (vsseg_fx_add: order-independent); 0: fp64 atomics (the hard-Dice voxel
// counts: integers, exact in fp64 whatever the order)
constexpr double VSSEG_FX_DICE = 4294967296.0;  // 2^32: sums of at most 2^31 voxel weights in [0, 1], 2.3e-10 resolution
__device__ __forceinline__ void block_reduce_add(double* vals, int nvals, double* dst, double fx, unsigned* fxflag) {
  __shared__ double sh[16][8];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int k = 0; k < nvals; ++k) {
    double v = wave_sum_d(vals[k]);
    if (lane == 0) sh[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < nvals) {
    double t = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w][threadIdx.x];
    if (fx > 0.0) vsseg_fx_add(&dst[threadIdx.x], t, fx, fxflag);
    else atomicAdd(&dst[threadIdx.x], t);
  }
}

// sums[b][c][0..2] = (I, G, P) of the (optionally hardness-weighted) 2-class soft Dice
__global__ void dice_pred_sums_kernel(const float* __restrict__ logits, int pitch, const float* __restrict__ label, int64_t nvox, int hardness, double* __restrict__ sums, unsigned* fxflag) {
  const int b = blockIdx.y;
  const float* lg = logits + (int64_t)b * nvox * pitch;
  const float* lb = label + (int64_t)b * nvox;
  float I0 = 0, G0 = 0, P0 = 0, I1 = 0, G1 = 0, P1 = 0;
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < nvox; v += (int64_t)gridDim.x * blockDim.x) {
    float2 l = *reinterpret_cast<const float2*>(lg + v * pitch);
    float m = fmaxf(l.x, l.y), e0 = __expf(l.x - m), e1 = __expf(l.y - m), inv = 1.f / (e0 + e1);
    float p0 = e0 * inv, p1 = e1 * inv;
    int cls = (int)(long long)lb[v];
    float g1 = cls == 1 ? 1.f : 0.f, g0 = cls == 0 ? 1.f : 0.f;
    float w0 = 1.f, w1 = 1.f;
    if (hardness) { w0 = 0.6f * fabsf(p0 - g0) + 0.4f; w1 = 0.6f * fabsf(p1 - g1) + 0.4f; }
    I0 += w0 * g0 * p0; G0 += w0 * g0; P0 += w0 * p0;
    I1 += w1 * g1 * p1; G1 += w1 * g1; P1 += w1 * p1;
  }
  double vals[6] = {I0, G0, P0, I1, G1, P1};
  block_reduce_add(vals, 6, sums + (int64_t)b * 6, VSSEG_FX_DICE, fxflag);
}
extern "C" int vsseg_dice_pred_sums(const float* logits, int32_t pitch, const float* label, int32_t n, int64_t nvox, int32_t hardness, double* sums, void* stream) {
  VSSEG_CHECK(logits && label && sums && pitch >= 2 && pitch % 2 == 0 && n >= 1, "vsseg_dice_pred_sums: bad arguments");
  dim3 g(grid_for(nvox, 256, 1024), n);
  VSSEG_FX_FLAG(fxflag, "vsseg_dice_pred_sums");
  hipLaunchKernelGGL(dice_pred_sums_kernel, g, dim3(256), 0, as_stream(stream), logits, pitch, label, nvox, hardness, sums, fxflag);
  VSSEG_LAUNCH_CHECK("vsseg_dice_pred_sums");
  return VSSEG_OK;
}

// sums[b][0..2] = (I, G, P) of the single-channel Dice between an attention map and the pooled label
__global__ void dice_att_sums_kernel(const float* __restrict__ att, const float* __restrict__ label, int64_t nvox, double* __restrict__ sums, unsigned* fxflag) {
  const int b = blockIdx.y;
  const float* a = att + (int64_t)b * nvox;
  const float* lb = label + (int64_t)b * nvox;
  float I = 0, G = 0, P = 0;
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < nvox; v += (int64_t)gridDim.x * blockDim.x) {
    float p = a[v], g = lb[v];
    I += g * p; G += g; P += p;
  }
  double vals[3] = {I, G, P};
  block_reduce_add(vals, 3, sums + (int64_t)b * 3, VSSEG_FX_DICE, fxflag);
}
extern "C" int vsseg_dice_att_sums(const float* att, const float* label, int32_t n, int64_t nvox, double* sums, void* stream) {
  VSSEG_CHECK(att && label && sums && n >= 1, "vsseg_dice_att_sums: bad arguments");
  dim3 g(grid_for(nvox, 256, 1024), n);
  VSSEG_FX_FLAG(fxflag, "vsseg_dice_att_sums");
  hipLaunchKernelGGL(dice_att_sums_kernel, g, dim3(256), 0, as_stream(stream), att, label, nvox, sums, fxflag);
  VSSEG_LAUNCH_CHECK("vsseg_dice_att_sums");
  return VSSEG_OK;
}

// loss = mean_{b,c} f_pred + sum_l (1/L) mean_b f_att,  f = 1 - (2I+eps)/(G+P+eps).  coef holds d(loss)/dI and d(loss)/dG(=dP).
__global__ void dice_finalize_kernel(const double* pred_sums, const double* att_sums, int n, int nlevels, float* loss, float* coef, const unsigned* fxflag) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double total = vsseg_fx_poison(fxflag);  // NaN while a partial sum of this process was non-finite / out of range (common.h)
  for (int b = 0; b < n; ++b)
    for (int c = 0; c < 2; ++c) {
      const double* sp = pred_sums + (b * 2 + c) * 3;
      const double s[3] = {vsseg_fx_get(sp, VSSEG_FX_DICE), vsseg_fx_get(sp + 1, VSSEG_FX_DICE), vsseg_fx_get(sp + 2, VSSEG_FX_DICE)};
      double D = s[1] + s[2] + SMOOTH + total * 0.0, num = 2.0 * s[0] + SMOOTH, wgt = 1.0 / (2.0 * n);  // (+ NaN when poisoned: the coefficients too)
      total += wgt * (1.0 - num / D);
      coef[(b * 2 + c) * 2 + 0] = (float)(-2.0 / D * wgt);
      coef[(b * 2 + c) * 2 + 1] = (float)(num / (D * D) * wgt);
    }
  for (int l = 0; l < nlevels; ++l)
    for (int b = 0; b < n; ++b) {
      const double* sp = att_sums + (l * n + b) * 3;
      const double s[3] = {vsseg_fx_get(sp, VSSEG_FX_DICE), vsseg_fx_get(sp + 1, VSSEG_FX_DICE), vsseg_fx_get(sp + 2, VSSEG_FX_DICE)};
      double D = s[1] + s[2] + SMOOTH + total * 0.0, num = 2.0 * s[0] + SMOOTH, wgt = 1.0 / ((double)nlevels * n);
      total += wgt * (1.0 - num / D);
      coef[n * 4 + (l * n + b) * 2 + 0] = (float)(-2.0 / D * wgt);
      coef[n * 4 + (l * n + b) * 2 + 1] = (float)(num / (D * D) * wgt);
    }
  *loss = (float)total;
}
extern "C" int vsseg_dice_finalize(const double* pred_sums, const double* att_sums, int32_t n, int32_t nlevels, float* loss, float* coef, void* stream) {
  VSSEG_CHECK(pred_sums && loss && coef && (nlevels == 0 || att_sums), "vsseg_dice_finalize: bad arguments");
  VSSEG_FX_FLAG(fxflag, "vsseg_dice_finalize");
  hipLaunchKernelGGL(dice_finalize_kernel, dim3(1), dim3(64), 0, as_stream(stream), pred_sums, att_sums, n, nlevels, loss, coef, fxflag);
  VSSEG_LAUNCH_CHECK("vsseg_dice_finalize");
  return VSSEG_OK;
}

// d(loss)/d(logits): through the Dice sums, the (non-detached) hardness weight (ref :279-283) and the softmax
__global__ void dice_pred_bwd_kernel(const float* __restrict__ logits, int pitch, const float* __restrict__ label, int64_t nvox, int hardness, const float* __restrict__ coef, const float* __restrict__ gscale, float* __restrict__ dlogits) {
  const int b = blockIdx.y;
  const float* lg = logits + (int64_t)b * nvox * pitch;
  const float* lb = label + (int64_t)b * nvox;
  float* dl = dlogits + (int64_t)b * nvox * 2;
  const float gs = gscale ? *gscale : 1.f;
  const float A0 = coef[(b * 2 + 0) * 2] * gs, B0 = coef[(b * 2 + 0) * 2 + 1] * gs, A1 = coef[(b * 2 + 1) * 2] * gs, B1 = coef[(b * 2 + 1) * 2 + 1] * gs;
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < nvox; v += (int64_t)gridDim.x * blockDim.x) {
    float2 l = *reinterpret_cast<const float2*>(lg + v * pitch);
    float m = fmaxf(l.x, l.y), e0 = __expf(l.x - m), e1 = __expf(l.y - m), inv = 1.f / (e0 + e1);
    float p0 = e0 * inv, p1 = e1 * inv;
    int cls = (int)(long long)lb[v];
    float g1 = cls == 1 ? 1.f : 0.f, g0 = cls == 0 ? 1.f : 0.f;
    float w0 = 1.f, w1 = 1.f, s0 = 0.f, s1 = 0.f;
    if (hardness) {
      float d0 = p0 - g0, d1 = p1 - g1;
      w0 = 0.6f * fabsf(d0) + 0.4f; w1 = 0.6f * fabsf(d1) + 0.4f;
      s0 = d0 > 0.f ? 0.6f : (d0 < 0.f ? -0.6f : 0.f);
      s1 = d1 > 0.f ? 0.6f : (d1 < 0.f ? -0.6f : 0.f);
    }
    float t0 = w0 + p0 * s0, t1 = w1 + p1 * s1;  // d(w*p)/dp
    float dp0 = A0 * g0 * t0 + B0 * (g0 * s0 + t0);
    float dp1 = A1 * g1 * t1 + B1 * (g1 * s1 + t1);
    float dot = dp0 * p0 + dp1 * p1;
    *reinterpret_cast<float2*>(dl + v * 2) = make_float2(p0 * (dp0 - dot), p1 * (dp1 - dot));
  }
}
extern "C" int vsseg_dice_pred_bwd(const float* logits, int32_t pitch, const float* label, int32_t n, int64_t nvox, int32_t hardness, const float* coef, const float* gscale, float* dlogits, void* stream) {
  VSSEG_CHECK(logits && label && coef && dlogits && pitch >= 2 && pitch % 2 == 0, "vsseg_dice_pred_bwd: bad arguments");
  dim3 g(grid_for(nvox, 256, 2048), n);
  hipLaunchKernelGGL(dice_pred_bwd_kernel, g, dim3(256), 0, as_stream(stream), logits, pitch, label, nvox, hardness, coef, gscale, dlogits);
  VSSEG_LAUNCH_CHECK("vsseg_dice_pred_bwd");
  return VSSEG_OK;
}

__global__ void dice_att_bwd_kernel(const float* __restrict__ label, int64_t nvox, const float* __restrict__ coef, const float* __restrict__ gscale, float* __restrict__ datt) {
  const int b = blockIdx.y;
  const float gs = gscale ? *gscale : 1.f;
  const float A = coef[b * 2] * gs, B = coef[b * 2 + 1] * gs;
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < nvox; v += (int64_t)gridDim.x * blockDim.x) datt[(int64_t)b * nvox + v] = A * label[(int64_t)b * nvox + v] + B;
}
extern "C" int vsseg_dice_att_bwd(const float* label, int32_t n, int64_t nvox, const float* coef, float inv_levels, const float* gscale, float* datt, void* stream) {
  (void)inv_levels;  // the 1/L factor is already folded into coef by vsseg_dice_finalize
  VSSEG_CHECK(label && coef && datt, "vsseg_dice_att_bwd: bad arguments");
  dim3 g(grid_for(nvox, 256, 2048), n);
  hipLaunchKernelGGL(dice_att_bwd_kernel, g, dim3(256), 0, as_stream(stream), label, nvox, coef, gscale, datt);
  VSSEG_LAUNCH_CHECK("vsseg_dice_att_bwd");
  return VSSEG_OK;
}

// hard Dice: argmax over the 2 channels vs label (ties -> class 0, like torch.argmax)
__global__ void hard_dice_kernel(const float* __restrict__ logits, int pitch, const float* __restrict__ label, int64_t nvox, double* __restrict__ counts) {
  float pg = 0, p = 0, g = 0;
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < nvox; v += (int64_t)gridDim.x * blockDim.x) {
    float2 l = *reinterpret_cast<const float2*>(logits + v * pitch);
    float pr = l.y > l.x ? 1.f : 0.f;
    float gg = ((int)(long long)label[v]) == 1 ? 1.f : 0.f;
    pg += pr * gg; p += pr; g += gg;
  }
  double vals[3] = {pg, p, g};
  block_reduce_add(vals, 3, counts, 0.0, nullptr);
}
extern "C" int vsseg_hard_dice_counts(const float* logits, int32_t pitch, const float* label, int64_t nvox, double* counts, void* stream) {
  VSSEG_CHECK(logits && label && counts && pitch >= 2 && pitch % 2 == 0, "vsseg_hard_dice_counts: bad arguments");
  hipLaunchKernelGGL(hard_dice_kernel, dim3(grid_for(nvox, 256, 1024)), dim3(256), 0, as_stream(stream), logits, pitch, label, nvox, counts);
  VSSEG_LAUNCH_CHECK("vsseg_hard_dice_counts");
  return VSSEG_OK;
}
__global__ void argmax2_kernel(const float* __restrict__ logits, int pitch, int64_t nvox, uint8_t* __restrict__ dst) {
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < nvox; v += (int64_t)gridDim.x * blockDim.x) {
    float2 l = *reinterpret_cast<const float2*>(logits + v * pitch);
    dst[v] = l.y > l.x ? 1 : 0;
  }
}
extern "C" int vsseg_argmax2(const float* logits, int32_t pitch, int64_t nvox, uint8_t* dst, void* stream) {
  VSSEG_CHECK(logits && dst && pitch >= 2 && pitch % 2 == 0, "vsseg_argmax2: bad arguments");
  hipLaunchKernelGGL(argmax2_kernel, dim3(grid_for(nvox, 256)), dim3(256), 0, as_stream(stream), logits, pitch, nvox, dst);
  VSSEG_LAUNCH_CHECK("vsseg_argmax2");
  return VSSEG_OK;
}
