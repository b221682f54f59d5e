// The per-element arithmetic of the BatchNorm -> Dropout -> PReLU backward (second pass), shared by vsseg_bn_act_bwd_apply (elementwise.hip) and by the
// fused data + weight gradient kernel that applies it ON LOAD (mbwd.hip): both produce bit-identical d(conv output) values.
//
//   forward (ref:params/networks/blocks/convolutions.py:148-156):  z = y*scale + shift;  d = keep ? z/(1-p) : 0;  a = d > 0 ? d : alpha*d
//   backward:  dz = keep ? (z > 0 ? g : alpha*g)/(1-p) : 0           (g = d(loss)/d(a); the PReLU branch is re-decided on the forward's own fp32 z)
//              dy = gamma*invstd * (dz - mean(dz) - xhat*mean(dz*xhat)),   xhat = (y - mean)*invstd
//            = k1i * t + (kc * (y - mean) + kd)       t = keep ? (z > 0 ? g : alpha*g) : 0,   k1i = gamma*invstd/(1-p),
//                                                      kc = -gamma*invstd^2*mean(dz*xhat),     kd = -gamma*invstd*mean(dz)
// (y - mean) is formed per element, so nothing cancels when |mean| >> std.
#pragma once
#include "common.h"

struct BnBwdC8 {  // constants of one 8-channel group, in registers
  float sc[8], sh[8], mu[8], k1i[8], kc[8], kd[8];
};

__device__ __forceinline__ void bn_bwd_consts(BnBwdC8& c, const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ scale,
                                              const float* __restrict__ shift, const float* __restrict__ mean_dz, const float* __restrict__ mean_dzx, int ch0, float inv_keep) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float is = invstd[ch0 + j], k1 = gamma[ch0 + j] * is;
    c.sc[j] = scale[ch0 + j];
    c.sh[j] = shift[ch0 + j];
    c.mu[j] = mean[ch0 + j];
    c.k1i[j] = k1 * inv_keep;
    c.kc[j] = -(k1 * mean_dzx[ch0 + j]) * is;
    c.kd[j] = -(k1 * mean_dz[ch0 + j]);
  }
}

// The forward of the same block on 8 channels, shared by vsseg_bn_act_fwd (elementwise.hip) and by the kernels that apply it ON LOAD to a convolution input that
// was never materialised (mconv.hip BIN, mbwd.hip XBN): bit-identical activations.   z = y*scale + shift;  d = keep ? z/(1-p) : 0;  a = d > 0 ? d : alpha*d
__device__ __forceinline__ void bn_fwd_act8(const f8& y, unsigned keep, float alpha, float inv_keep, const float (&sc)[8], const float (&sh)[8], f8& o) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float z = y.v[j] * sc[j] + sh[j];
    z = ((keep >> j) & 1u) ? z * inv_keep : 0.f;  // p_drop == 0: keep == 0xff and inv_keep == 1
    o.v[j] = z > 0.f ? z : alpha * z;
  }
}

// 8 channels: y (conv output), g (gradient of the block output), keep bits -> dy
__device__ __forceinline__ void bn_bwd_dy8(const f8& y, const f8& g, unsigned keep, float alpha, const BnBwdC8& c, f8& o) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float z = y.v[j] * c.sc[j] + c.sh[j];  // bit-identical to the forward's value: same side of the PReLU kink
    const float dd = z > 0.f ? g.v[j] : alpha * g.v[j];
    const float t = ((keep >> j) & 1u) ? dd : 0.f;
    o.v[j] = __builtin_fmaf(c.k1i[j], t, __builtin_fmaf(c.kc[j], y.v[j] - c.mu[j], c.kd[j]));
  }
}

__device__ __forceinline__ f8 bf16x8_to_f8(const uint4 u) {
  return f8{{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u),
             __uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u), __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u)}};
}
__device__ __forceinline__ uint4 f8_to_bf16x8(const f8& a) {
  uint4 u;
  u.x = f2bf2(a.v[0], a.v[1]);
  u.y = f2bf2(a.v[2], a.v[3]);
  u.z = f2bf2(a.v[4], a.v[5]);
  u.w = f2bf2(a.v[6], a.v[7]);
  return u;
}
