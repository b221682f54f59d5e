// HBM-bound kernels of the hot path: BatchNorm/Dropout/PReLU forward+backward, attention gate, staging, Adam.
// All are one-pass streaming kernels with 16-byte (8 x bf16) accesses along the channel axis of the channels-last layout.
// The streaming kernels of this file read and write their bf16 tensors with non-temporal loads / stores (ld8 / st4 / st8 in common.h, Raw8
// below): stores +3-6 % on each kernel and -0.2 ms per step, loads another +5-12 % (bn_act_bwd_apply 5.2 -> 5.8 TB/s, att_apply_fwd 5.1 -> 5.7)
// and -0.3 ms.  NOT the convolution kernels: with their outputs stored non-temporally the step got 1.5 ms slower.
#define VSSEG_NT_STORES
#define VSSEG_NT_LOADS
#include "common.h"
#include "bn_bwd.h"
#include <algorithm>
#include <type_traits>

#define DISPATCH_T(dtype, ...)                         \
  do {                                                 \
    if ((dtype) == VSSEG_F32) { typedef float T; __VA_ARGS__; } \
    else { typedef bf16_t T; __VA_ARGS__; }            \
  } while (0)

// two-part tensors (vsseg_hip.h): channels >= csplit live at ptr2
static inline int split_of(const vsseg_tensor& t) { return t.ptr2 ? t.csplit : 0x7fffffff; }
static inline bool two_part_ok(const vsseg_tensor& t) {
  return !t.ptr2 || (t.csplit > 0 && t.csplit < t.c && t.csplit % 16 == 0 && t.pitch >= t.csplit && t.pitch >= t.c - t.csplit);
}
#define VSSEG_ONE_PART(name, ...)                                                                     \
  do {                                                                                                \
    const vsseg_tensor* ts_[] = {__VA_ARGS__};                                                        \
    for (const vsseg_tensor* t_ : ts_) VSSEG_CHECK(!t_->ptr2, name ": two-part tensors are not supported here"); \
  } while (0)

// block size such that every thread keeps one fixed 8-channel group while striding over voxels
static inline int block_for_cgs(int cgs) {
  int a = cgs, b = 64;
  while (b) { int t = a % b; a = b; b = t; }
  int blk = 64 * (cgs / a);
  while (blk < 256 && (blk * 2) % cgs == 0) blk *= 2;
  return blk > 1024 ? 0 : blk;
}

// ------------------------------------------------------------------------------------------------------------
// gather-cast (weight packing), input staging, copies
// ------------------------------------------------------------------------------------------------------------
template <typename T> __global__ void gather_cast_kernel(const float* __restrict__ src, const int32_t* __restrict__ map, const int32_t* __restrict__ map2, T* __restrict__ dst, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int m = map[i];
    float v = m >= 0 ? src[m] : 0.f;
    if (map2) {
      int m2 = map2[i];
      if (m2 >= 0) v += src[m2];
    }
    Elem<T>::st(dst + i, v);
  }
}
extern "C" int vsseg_gather_cast(const float* src, const int32_t* map, const int32_t* map2, void* dst, int64_t n, int32_t dst_dtype, void* stream) {
  VSSEG_CHECK(src && map && dst && n >= 0, "vsseg_gather_cast: bad arguments");
  if (n == 0) return VSSEG_OK;
  DISPATCH_T(dst_dtype, hipLaunchKernelGGL(gather_cast_kernel<T>, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), src, map, map2, (T*)dst, n));
  VSSEG_LAUNCH_CHECK("vsseg_gather_cast");
  return VSSEG_OK;
}

__global__ void store_u64_kernel(uint64_t* dst, uint64_t value) { *dst = value; }
extern "C" int vsseg_store_u64(uint64_t* dst, uint64_t value, void* stream) {
  VSSEG_CHECK(dst, "vsseg_store_u64: null pointer");
  hipLaunchKernelGGL(store_u64_kernel, dim3(1), dim3(1), 0, as_stream(stream), dst, value);
  VSSEG_LAUNCH_CHECK("vsseg_store_u64");
  return VSSEG_OK;
}

template <typename T> __global__ void stage_input_kernel(const float* __restrict__ src, int n, int sx, int sy, int sz, int ox, int oy, int oz, vsseg_tensor dst) {
  const int64_t per = (int64_t)dst.x * dst.y * dst.z, total = per * n;
  T* out = reinterpret_cast<T*>(dst.ptr);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t b = i / per, r = i - b * per;
    int z = (int)(r % dst.z); r /= dst.z;
    int y = (int)(r % dst.y);
    int x = (int)(r / dst.y);
    int gx = x + ox, gy = y + oy, gz = z + oz;
    float v = 0.f;
    if ((unsigned)gx < (unsigned)sx && (unsigned)gy < (unsigned)sy && (unsigned)gz < (unsigned)sz) v = src[((b * sx + gx) * sy + gy) * (int64_t)sz + gz];
    T* o = out + i * dst.pitch;
    if ((dst.c & 7) == 0) {
      f8 e{{v, 0, 0, 0, 0, 0, 0, 0}};
      for (int c = 0; c < dst.c; c += 8) { st8(o + c, e); e.v[0] = 0.f; }
    } else {  // plain crop (c == 1) or an odd channel count: scalar stores
      Elem<T>::st(o, v);
      for (int c = 1; c < dst.c; ++c) Elem<T>::st(o + c, 0.f);
    }
  }
}
// the same for dst.z % 4 == 0 and fewer than 2^31 voxels: a thread owns four z-consecutive voxels of one row — one 32-bit coordinate decode and (inside the volume, 16-byte
// aligned) one 16-byte load per four voxels, one 8- / 16-byte store for the compact (one-channel) layouts.  The scalar kernel above spends ~150 instructions of 64-bit
// division per voxel: 40 us for one 384x128x128 window (0.95 TB/s), 85 us for the batch of four; this one is bandwidth-bound
template <typename T> __global__ void stage_input_z4_kernel(const float* __restrict__ src, int n, int sx, int sy, int sz, int ox, int oy, int oz, vsseg_tensor dst) {
  const unsigned zq = (unsigned)dst.z >> 2, items = (unsigned)n * dst.x * dst.y * zq;
  T* out = reinterpret_cast<T*>(dst.ptr);
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < items; i += gridDim.x * blockDim.x) {
    const unsigned row = i / zq, z = (i - row * zq) * 4u;
    const unsigned t = row / (unsigned)dst.y, y = row - t * dst.y, b = t / (unsigned)dst.x, x = t - b * dst.x;
    const int gx = (int)x + ox, gy = (int)y + oy, gz = (int)z + oz;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)gx < (unsigned)sx && (unsigned)gy < (unsigned)sy) {
      const float* sp = src + (((int64_t)b * sx + gx) * sy + gy) * (int64_t)sz + gz;
      if (gz >= 0 && gz + 3 < sz && ((uintptr_t)sp & 15) == 0) {
        const float4 q = *reinterpret_cast<const float4*>(sp);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if ((unsigned)(gz + j) < (unsigned)sz) v[j] = sp[j];
      }
    }
    T* o = out + ((int64_t)row * dst.z + z) * dst.pitch;
    if (dst.c == 1 && dst.pitch == 1) {
      st4(o, make_float4(v[0], v[1], v[2], v[3]));
    } else if ((dst.c & 7) == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f8 e{{v[j], 0, 0, 0, 0, 0, 0, 0}};
        for (int c = 0; c < dst.c; c += 8) { st8(o + (int64_t)j * dst.pitch + c, e); e.v[0] = 0.f; }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        Elem<T>::st(o + (int64_t)j * dst.pitch, v[j]);
        for (int c = 1; c < dst.c; ++c) Elem<T>::st(o + (int64_t)j * dst.pitch + c, 0.f);
      }
    }
  }
}
extern "C" int vsseg_stage_input(const float* src, int32_t n, const int32_t sdims[3], const int32_t origin[3], vsseg_tensor dst, void* stream) {
  VSSEG_ONE_PART("vsseg_stage_input", &dst);
  VSSEG_CHECK(src && dst.ptr && dst.c >= 1 && dst.pitch >= dst.c && dst.n == n, "vsseg_stage_input: bad arguments");
  VSSEG_CHECK(dst.c % 8 != 0 || dst.pitch % 8 == 0, "vsseg_stage_input: vectorised path needs pitch %% 8 == 0");
  int64_t total = tensor_voxels(dst);
  const int es = dst.dtype == VSSEG_F32 ? 4 : 2;
  if (dst.z % 4 == 0 && total < (1ll << 31) && total > 0 && ((uintptr_t)dst.ptr % (4 * es)) == 0 && (dst.c != 1 || dst.pitch == 1)) {
    DISPATCH_T(dst.dtype, hipLaunchKernelGGL(stage_input_z4_kernel<T>, dim3(grid_for(total / 4, 256)), dim3(256), 0, as_stream(stream), src, n, sdims[0], sdims[1], sdims[2], origin[0], origin[1], origin[2], dst));
  } else {
    DISPATCH_T(dst.dtype, hipLaunchKernelGGL(stage_input_kernel<T>, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), src, n, sdims[0], sdims[1], sdims[2], origin[0], origin[1], origin[2], dst));
  }
  VSSEG_LAUNCH_CHECK("vsseg_stage_input");
  return VSSEG_OK;
}

template <typename S, typename D, bool ADD> __global__ void copy_kernel(const S* __restrict__ src, int sp, D* __restrict__ dst, int dp, int c, int64_t nvox) {
  const int64_t total = nvox * c;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t v = i / c;
    int ch = (int)(i - v * c);
    float x = Elem<S>::ld(src + v * sp + ch);
    D* o = dst + v * dp + ch;
    Elem<D>::st(o, ADD ? Elem<D>::ld(o) + x : x);
  }
}
// fp32 [voxel][2] -> bf16 rows of 8 channels (the loss' logits gradient staged for the backward convolutions): one 8-byte load and one full 16-byte
// row store per voxel (channels 2..7 = 0) instead of two 2-byte stores into every 16-byte row (0.22 -> 0.13 ms on 4 x 384x128x128)
__global__ void copy_f32x2_to_bf16_row8_kernel(const float2* __restrict__ src, bf16_t* __restrict__ dst, int64_t nvox) {
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < nvox; v += (int64_t)gridDim.x * blockDim.x) {
    const float2 x = src[v];
    st8(dst + v * 8, f8{{x.x, x.y, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}});
  }
}
template <bool ADD> static int copy_impl(vsseg_tensor src, vsseg_tensor dst, void* stream, const char* name) {
  VSSEG_ONE_PART("vsseg_copy_cast/add_inplace", &src, &dst);
  VSSEG_CHECK(src.ptr && dst.ptr && src.c == dst.c && tensor_voxels(src) == tensor_voxels(dst), "%s: shape mismatch", name);
  int64_t nv = tensor_voxels(src), total = nv * src.c;
  dim3 g(grid_for(total, 256)), b(256);
  hipStream_t s = as_stream(stream);
  if (!ADD && src.dtype == VSSEG_F32 && dst.dtype == VSSEG_BF16 && src.c == 2 && src.pitch == 2 && dst.pitch == 8 && (dst.reserved & VSSEG_ZERO_PADDED) && ((uintptr_t)src.ptr & 7) == 0 && ((uintptr_t)dst.ptr & 15) == 0) {
    hipLaunchKernelGGL(copy_f32x2_to_bf16_row8_kernel, dim3(grid_for(nv, 256)), b, 0, s, (const float2*)src.ptr, (bf16_t*)dst.ptr, nv);
    VSSEG_LAUNCH_CHECK(name);
    return VSSEG_OK;
  }
  if (src.dtype == VSSEG_F32 && dst.dtype == VSSEG_F32) hipLaunchKernelGGL((copy_kernel<float, float, ADD>), g, b, 0, s, (const float*)src.ptr, src.pitch, (float*)dst.ptr, dst.pitch, src.c, nv);
  else if (src.dtype == VSSEG_F32) hipLaunchKernelGGL((copy_kernel<float, bf16_t, ADD>), g, b, 0, s, (const float*)src.ptr, src.pitch, (bf16_t*)dst.ptr, dst.pitch, src.c, nv);
  else if (dst.dtype == VSSEG_F32) hipLaunchKernelGGL((copy_kernel<bf16_t, float, ADD>), g, b, 0, s, (const bf16_t*)src.ptr, src.pitch, (float*)dst.ptr, dst.pitch, src.c, nv);
  else hipLaunchKernelGGL((copy_kernel<bf16_t, bf16_t, ADD>), g, b, 0, s, (const bf16_t*)src.ptr, src.pitch, (bf16_t*)dst.ptr, dst.pitch, src.c, nv);
  VSSEG_LAUNCH_CHECK(name);
  return VSSEG_OK;
}
extern "C" int vsseg_copy_cast(vsseg_tensor src, vsseg_tensor dst, void* stream) { return copy_impl<false>(src, dst, stream, "vsseg_copy_cast"); }
extern "C" int vsseg_add_inplace(vsseg_tensor dst, vsseg_tensor src, void* stream) { return copy_impl<true>(src, dst, stream, "vsseg_add_inplace"); }

// ------------------------------------------------------------------------------------------------------------
// BatchNorm statistics
// ------------------------------------------------------------------------------------------------------------
// sum of one value per shard (thread = shard) over the 256 threads of a block; result valid in thread 0
__device__ __forceinline__ double block256_sum_d(double v, double* part) {
  v = wave_sum_d(v);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  const double r = part[0] + part[1] + part[2] + part[3];
  __syncthreads();
  return r;
}
// one 256-thread block per channel: thread = statistics shard (a 64-thread block walking the 256 shards serially cost ~30 us
// per layer, 1.6 ms per step over the 52 forward/backward finalisations)
__global__ void bn_finalize_kernel(const double* __restrict__ stats, int stride, int c, double count, const float* gamma, const float* beta, float eps, float momentum,
                                   float* running_mean, float* running_var, int64_t* num_batches, float* mean, float* invstd, float* scale, float* shift, const unsigned* fxflag) {
  static_assert(VSSEG_STAT_SHARDS == 256, "thread = shard");
  __shared__ double part[4];
  const int ch = blockIdx.x, sh = threadIdx.x;
  const double s = block256_sum_d(vsseg_fx_get(&stats[(int64_t)sh * 2 * stride + ch], VSSEG_FX_STAT), part);  // fixed-point shards (vsseg_fx_add), fixed summation order
  const double q = block256_sum_d(vsseg_fx_get(&stats[(int64_t)sh * 2 * stride + stride + ch], VSSEG_FX_STAT), part);
  if (threadIdx.x != 0) return;
  if (ch == 0 && num_batches) *num_batches += 1;
  double m = s / count + vsseg_fx_poison(fxflag);  // NaN while a partial sum was non-finite / out of range (common.h)
  double var = q / count - m * m;
  if (var < 0) var = 0;
  float is = (float)(1.0 / sqrt(var + (double)eps));
  mean[ch] = (float)m;
  invstd[ch] = is;
  float sc = gamma[ch] * is;
  scale[ch] = sc;
  shift[ch] = beta[ch] - (float)m * sc;
  if (running_mean) {
    double unbiased = count > 1 ? var * count / (count - 1) : var;
    running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * (float)m;
    running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * (float)unbiased;
  }
}
extern "C" int vsseg_bn_finalize(const double* stats, int32_t stride, int32_t c, double count, const float* gamma, const float* beta, float eps, float momentum,
                                 float* running_mean, float* running_var, int64_t* num_batches, float* mean, float* invstd, float* scale, float* shift, void* stream) {
  VSSEG_CHECK(stats && gamma && beta && mean && invstd && scale && shift && c > 0 && c <= stride, "vsseg_bn_finalize: bad arguments");
  VSSEG_FX_FLAG(fxflag, "vsseg_bn_finalize");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(c), dim3(256), 0, as_stream(stream), stats, stride, c, count, gamma, beta, eps, momentum, running_mean, running_var, num_batches, mean, invstd, scale, shift, fxflag);
  VSSEG_LAUNCH_CHECK("vsseg_bn_finalize");
  return VSSEG_OK;
}
__global__ void bn_fold_eval_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, float* scale, float* shift, int c) {
  int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  float sc = gamma[ch] / sqrtf(rv[ch] + eps);
  scale[ch] = sc;
  shift[ch] = beta[ch] - rm[ch] * sc;
}
extern "C" int vsseg_bn_fold_eval(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, float* scale, float* shift, int32_t c, void* stream) {
  VSSEG_CHECK(gamma && beta && rm && rv && scale && shift && c > 0, "vsseg_bn_fold_eval: bad arguments");
  hipLaunchKernelGGL(bn_fold_eval_kernel, dim3((c + 63) / 64), dim3(64), 0, as_stream(stream), gamma, beta, rm, rv, eps, scale, shift, c);
  VSSEG_LAUNCH_CHECK("vsseg_bn_fold_eval");
  return VSSEG_OK;
}

// ------------------------------------------------------------------------------------------------------------
// BN -> Dropout -> PReLU (+ residual) forward            ref:params/networks/blocks/convolutions.py:148-156, 252-255
// ------------------------------------------------------------------------------------------------------------
// RES: 0 no residual, 1 residual tensor, 2 residual = 1x1x1 convolution of a ONE-channel input computed on the fly
// (res[v][c] = x1[v] * rw[c] + rb[c]: the ResidualUnit residual of the first encoder block, ref:.../convolutions.py:241-250 with
// in_channels = 1 — its 16-channel output tensor is never written or read)
template <typename T, int RES>
__global__ void bn_act_fwd_kernel(const T* __restrict__ y, int yp, const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ alpha_p,
                                  float p_drop, uint64_t seed, uint32_t salt, const T* __restrict__ res, int rp, T* __restrict__ out, int op, int cgs, int64_t nvox,
                                  const float* __restrict__ rw, const float* __restrict__ rb, uint8_t* __restrict__ keep_out) {
  // The grid is a multiple of the group count (bn_fwd_grid), so a thread keeps ONE 8-channel group over all its voxels: the folded affine (and the
  // residual convolution's weights, RES 2) live in registers instead of being re-loaded per element (the res1 variant ran at 3.6 TB/s with the
  // 32 per-element constant loads, the plain one at 4.6-5.2), and the voxel index advances by an add instead of a 64-bit division.
  const int64_t gt = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, nthreads = (int64_t)gridDim.x * blockDim.x;
  const int cg = (int)(gt % cgs), c = cg * 8;
  const int64_t vstep = nthreads / cgs;
  const float alpha = *alpha_p, inv_keep = 1.f / (1.f - p_drop);
  float sc[8], sh[8], w1[RES == 2 ? 8 : 1], b1[RES == 2 ? 8 : 1];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = scale[c + j]; sh[j] = shift[c + j];
    if constexpr (RES == 2) { w1[j] = rw[c + j]; b1[j] = rb[c + j]; }
  }
  if (p_drop > 0.f) dropout_resolve_seed(seed, salt);
  for (int64_t v = gt / cgs; v < nvox; v += vstep) {
    const int64_t i = v * cgs + cg;  // index of the (voxel, 8-channel group) item: the dropout counter and the keep-mask byte
    f8 x = ld8(y + v * yp + c);
    unsigned keep = p_drop > 0.f ? dropout_keep8(seed, salt, (uint64_t)i, p_drop) : 0xffu;
    if (keep_out) keep_out[i] = (uint8_t)keep;  // one byte per (voxel, 8-channel group): the backward passes read it instead of re-running Philox twice
    f8 r;
    if (RES == 1) r = ld8(res + v * rp + c);
    if constexpr (RES == 2) {
      const float x1 = Elem<T>::ld(res + v);
#pragma unroll
      for (int j = 0; j < 8; ++j) r.v[j] = x1 * w1[j] + b1[j];
    }
    bn_fwd_act8(x, keep, alpha, inv_keep, sc, sh, x);  // (bn_bwd.h)
    if constexpr (RES != 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) x.v[j] += r.v[j];
    }
    st8(out + v * op + c, x);
  }
}
// <= 16 K workgroups of 256 threads (measured over the step's 26 launches: 2.49 ms with 4 K, 2.38 with 16-32 K, 2.59 with 256 K), a multiple of the
// 8-channel group count (every thread then owns one group)
static inline int bn_fwd_grid(int64_t items, int cgs) {
  int g = grid_for(items, 256, 16384);
  g = (g + cgs - 1) / cgs * cgs;
  return g;
}
extern "C" int vsseg_bn_act_fwd(vsseg_tensor y, const float* scale, const float* shift, const float* alpha, float p_drop, uint64_t seed, uint32_t salt,
                                vsseg_tensor res, int32_t has_res, vsseg_tensor out, uint8_t* keep_out, void* stream) {
  VSSEG_ONE_PART("vsseg_bn_act_fwd", &y, &res, &out);
  VSSEG_CHECK(y.ptr && out.ptr && scale && shift && alpha && y.c % 8 == 0 && y.pitch % 8 == 0 && out.pitch % 8 == 0 && out.c == y.c && out.dtype == y.dtype, "vsseg_bn_act_fwd: bad arguments");
  VSSEG_CHECK(!has_res || (res.ptr && res.dtype == y.dtype && res.c == y.c && res.pitch % 8 == 0), "vsseg_bn_act_fwd: bad residual");
  VSSEG_CHECK(p_drop >= 0.f && p_drop < 1.f, "vsseg_bn_act_fwd: dropout p out of range");
  int64_t nv = tensor_voxels(y);
  int cgs = y.c / 8;
  dim3 g(bn_fwd_grid(nv * cgs, cgs)), b(256);
  DISPATCH_T(y.dtype, if (has_res) hipLaunchKernelGGL((bn_act_fwd_kernel<T, 1>), g, b, 0, as_stream(stream), (const T*)y.ptr, y.pitch, scale, shift, alpha, p_drop, seed, salt, (const T*)res.ptr, res.pitch, (T*)out.ptr, out.pitch, cgs, nv, (const float*)nullptr, (const float*)nullptr, keep_out);
             else hipLaunchKernelGGL((bn_act_fwd_kernel<T, 0>), g, b, 0, as_stream(stream), (const T*)y.ptr, y.pitch, scale, shift, alpha, p_drop, seed, salt, (const T*)nullptr, 0, (T*)out.ptr, out.pitch, cgs, nv, (const float*)nullptr, (const float*)nullptr, keep_out));
  VSSEG_LAUNCH_CHECK("vsseg_bn_act_fwd");
  return VSSEG_OK;
}
extern "C" int vsseg_bn_act_fwd_res1(vsseg_tensor y, const float* scale, const float* shift, const float* alpha, float p_drop, uint64_t seed, uint32_t salt,
                                     const void* x1, const float* res_w, const float* res_b, vsseg_tensor out, uint8_t* keep_out, void* stream) {
  VSSEG_ONE_PART("vsseg_bn_act_fwd_res1", &y, &out);
  VSSEG_CHECK(y.ptr && out.ptr && scale && shift && alpha && x1 && res_w && res_b && y.c % 8 == 0 && y.pitch % 8 == 0 && out.pitch % 8 == 0 && out.c == y.c && out.dtype == y.dtype, "vsseg_bn_act_fwd_res1: bad arguments");
  VSSEG_CHECK(p_drop >= 0.f && p_drop < 1.f, "vsseg_bn_act_fwd_res1: dropout p out of range");
  int64_t nv = tensor_voxels(y);
  int cgs = y.c / 8;
  dim3 g(bn_fwd_grid(nv * cgs, cgs)), b(256);
  DISPATCH_T(y.dtype, hipLaunchKernelGGL((bn_act_fwd_kernel<T, 2>), g, b, 0, as_stream(stream), (const T*)y.ptr, y.pitch, scale, shift, alpha, p_drop, seed, salt, (const T*)x1, 0, (T*)out.ptr, out.pitch, cgs, nv, res_w, res_b, keep_out));
  VSSEG_LAUNCH_CHECK("vsseg_bn_act_fwd_res1");
  return VSSEG_OK;
}

__global__ void dropout_mask_kernel(float* mask, int64_t n8, float p, uint64_t seed, uint32_t salt) {
  if (p > 0.f) dropout_resolve_seed(seed, salt);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    unsigned keep = p > 0.f ? dropout_keep8(seed, salt, (uint64_t)i, p) : 0xffu;
    for (int j = 0; j < 8; ++j) mask[i * 8 + j] = (float)((keep >> j) & 1u);
  }
}
extern "C" int vsseg_dropout_mask(float* mask, int64_t nvox, int32_t c, float p_drop, uint64_t seed, uint32_t salt, void* stream) {
  VSSEG_CHECK(mask && c % 8 == 0, "vsseg_dropout_mask: bad arguments");
  int64_t n8 = nvox * (c / 8);
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n8, 256)), dim3(256), 0, as_stream(stream), mask, n8, p_drop, seed, salt);
  VSSEG_LAUNCH_CHECK("vsseg_dropout_mask");
  return VSSEG_OK;
}

// ------------------------------------------------------------------------------------------------------------
// backward of BN -> Dropout -> PReLU.  dz = d(loss)/d(BN output); pass 1 reduces, pass 2 applies.
// ------------------------------------------------------------------------------------------------------------
struct BnBwdArgs {
  const float *mean, *invstd, *gamma, *beta, *scale, *shift, *alpha;
  float p_drop;
  uint64_t seed;
  uint32_t salt;
};
// recompute the forward for 8 channels and return dz (in dz) and xhat (in xh); returns the dalpha contribution
__device__ __forceinline__ float bn_bwd_elem8(const f8& y, const f8& da, int c, int64_t i, const BnBwdArgs& a, float alpha, f8& dz, f8& xh) {
  const float inv_keep = 1.f / (1.f - a.p_drop);
  unsigned keep = a.p_drop > 0.f ? dropout_keep8(a.seed, a.salt, (uint64_t)i, a.p_drop) : 0xffu;
  float dal = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float xhat = (y.v[j] - a.mean[c + j]) * a.invstd[c + j];
    float z = y.v[j] * a.scale[c + j] + a.shift[c + j];  // bit-identical to the forward's value: same side of the PReLU kink
    bool k = (keep >> j) & 1u;
    float d = k ? z * inv_keep : 0.f;
    float g = da.v[j];
    float dd = d > 0.f ? g : alpha * g;
    if (d < 0.f) dal += g * d;
    dz.v[j] = k ? dd * inv_keep : 0.f;
    xh.v[j] = xhat;
  }
  return dal;
}

// Pass 1 of the BatchNorm + dropout + PReLU backward.  Per thread: one 8-channel group, U voxels in flight (2*U independent 16-byte loads).
// The loop keeps the forward's folded affine (scale, shift: the PReLU branch is re-decided on the SAME fp32 value) and the channel means in
// registers and accumulates sum(dz), sum(dz*(y - mean)), sum(dout); sum(dz*xhat) = invstd * sum(dz*(y - mean)) is formed once per workgroup.
// (Centred per element: sum(dz*y) - mean*sum(dz) from fp32 partial sums cancels when |mean| >> std; the extra subtract is free — A/B 32.1 ms
// per step either way, tests/test_gpu_ops.py::test_bn_dropout_prelu_forward_backward[(10, 0.05)].)
// 8 channels as loaded (bf16: 4 registers) — converted to fp32 only when consumed, so that U voxels in flight cost 8*U registers, not 16*U
template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> {
  uint4 u;
  static __device__ __forceinline__ Raw8 ld(const bf16_t* p) {
    Raw8 r;
#ifdef VSSEG_NT_LOADS
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    r.u = make_uint4(t[0], t[1], t[2], t[3]);
#else
    r.u = *reinterpret_cast<const uint4*>(p);
#endif
    return r;
  }
  __device__ __forceinline__ f8 cvt() const {
    return f8{{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u),
               __uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u), __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u)}};
  }
};
template <> struct Raw8<float> {
  f8 v;
  static __device__ __forceinline__ Raw8 ld(const float* p) { Raw8 r; r.v = ld8(p); return r; }
  __device__ __forceinline__ f8 cvt() const { return v; }
};

template <typename T, int U>
__global__ __launch_bounds__(1024) void bn_act_bwd_reduce_kernel(const T* __restrict__ y, int yp, const T* __restrict__ dout, int dp, BnBwdArgs a, int cgs, int64_t nvox, double* __restrict__ sums, int stride, double* __restrict__ alpha_acc,
                                                               const uint8_t* __restrict__ keep_in, unsigned* fxflag) {
  // one [3][c] + [1] region per WAVE: with a single region every thread of the workgroup added its 24 partial sums to the same 3*C
  // addresses (64 .. 128 lanes per address, serialised by the LDS): on a 48-channel 151 MB tensor that tail cost as much as the pass itself
  extern __shared__ float red[];
  const int C = cgs * 8, RW = 3 * C + 1, nw = (blockDim.x + 63) >> 6;
  for (int i = threadIdx.x; i < nw * RW; i += blockDim.x) red[i] = 0.f;
  __syncthreads();
  const float alpha = *a.alpha;
  if (a.p_drop > 0.f && keep_in == nullptr) dropout_resolve_seed(a.seed, a.salt);
  const int64_t gt = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, nthreads = (int64_t)gridDim.x * blockDim.x;
  const int cg = (int)(gt % cgs);
  const int c = cg * 8;
  float sc[8], sh[8], mu[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = a.scale[c + j]; sh[j] = a.shift[c + j]; mu[j] = a.mean[c + j]; }
  const bool drop = a.p_drop > 0.f;
  const float inv_keep = 1.f / (1.f - a.p_drop);
  float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s3[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dal = 0.f;
  const int64_t vstep = nthreads / cgs;
  // keep-mask: the byte the forward stored for this (voxel, 8-channel group) when the caller kept one (keep_in), else Philox again
  auto one = [&](const f8& yy, const f8& da, unsigned keep) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float z = yy.v[j] * sc[j] + sh[j];
      const bool k = (keep >> j) & 1u;
      const float d = k ? z * inv_keep : 0.f;
      const float g = da.v[j];
      const float dd = d > 0.f ? g : alpha * g;
      if (d < 0.f) dal += g * d;
      const float dz = k ? dd * inv_keep : 0.f;
      s1[j] += dz; s2[j] += dz * (yy.v[j] - mu[j]); s3[j] += g;  // centred per element: sum(dz*y) - mean*sum(dz) in fp32 cancels when |mean| >> std
    }
  };
  // The source of the keep-mask is decided ONCE, outside the loop: 0 no dropout, 1 the stored bytes, 2 Philox.  A per-element
  // `stored ? load : philox()` makes hipcc branch around the byte load and wait for it inside the batch of independent loads.
  auto sweep = [&](auto src_c) {
    constexpr int SRC = decltype(src_c)::value;
    auto mask_of = [&](int64_t vv) -> unsigned {
      if constexpr (SRC == 0) return 0xffu;
      else if constexpr (SRC == 1) return (unsigned)keep_in[vv * cgs + cg];
      else return dropout_keep8(a.seed, a.salt, (uint64_t)(vv * cgs + cg), a.p_drop);
    };
    int64_t v = gt / cgs;
    for (; v + (U - 1) * vstep < nvox; v += U * vstep) {
      Raw8<T> yv[U], dv[U];
      unsigned kp[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { yv[u] = Raw8<T>::ld(y + (v + u * vstep) * yp + c); dv[u] = Raw8<T>::ld(dout + (v + u * vstep) * dp + c); kp[u] = mask_of(v + u * vstep); }
#pragma unroll
      for (int u = 0; u < U; ++u) one(yv[u].cvt(), dv[u].cvt(), kp[u]);
    }
    for (; v < nvox; v += vstep) one(ld8(y + v * yp + c), ld8(dout + v * dp + c), mask_of(v));
  };
  if (!drop) sweep(std::integral_constant<int, 0>{});
  else if (keep_in != nullptr) sweep(std::integral_constant<int, 1>{});
  else sweep(std::integral_constant<int, 2>{});
  float* rw = red + (threadIdx.x >> 6) * RW;
#pragma unroll
  for (int j = 0; j < 8; ++j) { atomicAdd(&rw[c + j], s1[j]); atomicAdd(&rw[C + c + j], s2[j]); atomicAdd(&rw[2 * C + c + j], s3[j]); }
  dal = wave_sum(dal);
  if ((threadIdx.x & 63) == 0) rw[3 * C] = dal;
  __syncthreads();
  auto wsum = [&](int i) { float v = 0.f; for (int w = 0; w < nw; ++w) v += red[w * RW + i]; return v; };
  const int shard = blockIdx.x % VSSEG_STAT_SHARDS;
  for (int i = threadIdx.x; i < 3 * C; i += blockDim.x) {
    const int which = i / C, ch = i % C;
    double val = (double)wsum(i);
    if (which == 1) val *= (double)a.invstd[ch];  // sum(dz * xhat) of this workgroup
    vsseg_fx_add(&sums[(int64_t)shard * 3 * stride + which * stride + ch], val, VSSEG_FX_GRAD, fxflag);  // order-independent (fixed-point integer atomics)
  }
  if (threadIdx.x == 0) vsseg_fx_add(&alpha_acc[shard], (double)wsum(3 * C), VSSEG_FX_GRAD, fxflag);
}
extern "C" int vsseg_bn_act_bwd_reduce(vsseg_tensor y, vsseg_tensor dout, const float* mean, const float* invstd, const float* gamma, const float* beta, const float* scale, const float* shift, const float* alpha,
                                       float p_drop, uint64_t seed, uint32_t salt, double* sums, int32_t stride, double* alpha_acc, const uint8_t* keep_in, void* stream) {
  VSSEG_ONE_PART("vsseg_bn_act_bwd_reduce", &y, &dout);
  VSSEG_CHECK(y.ptr && dout.ptr && y.dtype == dout.dtype && y.c == dout.c && y.c % 8 == 0 && y.pitch % 8 == 0 && dout.pitch % 8 == 0 && sums && alpha_acc && stride >= y.c, "vsseg_bn_act_bwd_reduce: bad arguments");
  int cgs = y.c / 8, blk = block_for_cgs(cgs);
  VSSEG_CHECK(blk > 0, "vsseg_bn_act_bwd_reduce: unsupported channel count %d", y.c);
  int64_t nv = tensor_voxels(y);
  BnBwdArgs a{mean, invstd, gamma, beta, scale, shift, alpha, p_drop, seed, salt};
  size_t lds = (size_t)((blk + 63) / 64) * (3 * y.c + 1) * sizeof(float);
  // Every workgroup ends with a flush (LDS reduction, then 3*C dependent fp64 adds into the sharded sums), and on everything but the largest
  // tensors that tail — paid once per wave of workgroups on a CU — costs as much as the streaming itself (measured: a 403 MB 32-channel tensor
  // 0.185 ms with 1536 workgroups, 0.151 ms = 5.3 TB/s with 682; a 151 MB 48-channel tensor 0.133 ms with 1536, 0.070 ms with 455).  Giving
  // every workgroup its own row and plain read-modify-writes instead of atomics measured no better: it is the number of flushes, not the
  // atomics.  So the grid is sized by a flush budget (~64 K adds per launch beyond 16 channels: 682 workgroups at 32 channels, 455 at 48),
  // 768 workgroups on the 16-channel full-resolution tensors (the step's 26 launches: 2.36 ms with 2048 there, 2.25 with 512-1024),
  // never below 192 workgroups (the 6 MB tensors of level 4: 27 us with a floor of 384, 19 us with 192, no better below) and never with fewer
  // than 16 voxel groups per thread.
  const int64_t items = nv * cgs;
  const int budget = 65536;  // fp64 flush atomics per launch the grid is sized for (measured)
  const int cap = (int)std::min<int64_t>(256 * 8, std::max<int64_t>(192, std::min<int64_t>(items / ((int64_t)blk * 16), y.c <= 16 ? 256 * 3 : budget / (3 * y.c))));
  int grid = grid_for((nv * cgs + 1) / 2, blk, cap);  // 2 voxels in flight per thread (measured: 2 beats 4)
  VSSEG_FX_FLAG(fxflag, "vsseg_bn_act_bwd_reduce");
  DISPATCH_T(y.dtype, hipLaunchKernelGGL((bn_act_bwd_reduce_kernel<T, 2>), dim3(grid), dim3(blk), lds, as_stream(stream), (const T*)y.ptr, y.pitch, (const T*)dout.ptr, dout.pitch, a, cgs, nv, sums, stride, alpha_acc, keep_in, fxflag));
  VSSEG_LAUNCH_CHECK("vsseg_bn_act_bwd_reduce");
  return VSSEG_OK;
}

__global__ void bn_act_bwd_finalize_kernel(const double* __restrict__ sums, int stride, const double* __restrict__ alpha_acc, int c, double count, float* dgamma, float* dbeta, float* dalpha, float* mean_dz, float* mean_dzx, float* dres_bias, const unsigned* fxflag) {
  __shared__ double part[4];
  const int ch = blockIdx.x, sh = threadIdx.x;  // one block per channel, thread = shard
  const double* base = sums + (int64_t)sh * 3 * stride + ch;
  const double s = block256_sum_d(vsseg_fx_get(&base[0], VSSEG_FX_GRAD), part) + vsseg_fx_poison(fxflag);  // fixed-point shards (vsseg_fx_add); NaN while poisoned
  const double q = block256_sum_d(vsseg_fx_get(&base[stride], VSSEG_FX_GRAD), part) + vsseg_fx_poison(fxflag);
  const double r = block256_sum_d(vsseg_fx_get(&base[2 * stride], VSSEG_FX_GRAD), part) + vsseg_fx_poison(fxflag);
  double a = 0.0;
  if (ch == 0) a = block256_sum_d(vsseg_fx_get(&alpha_acc[sh], VSSEG_FX_GRAD), part);  // block-uniform branch
  if (threadIdx.x != 0) return;
  if (ch == 0) *dalpha += (float)(a + vsseg_fx_poison(fxflag));
  if (dres_bias) dres_bias[ch] += (float)r;  // d(out)/d(residual) = 1: the residual convolution's bias gradient is sum(dout)
  dbeta[ch] += (float)s;
  dgamma[ch] += (float)q;
  mean_dz[ch] = (float)(s / count);
  mean_dzx[ch] = (float)(q / count);
}
extern "C" int vsseg_bn_act_bwd_finalize(const double* sums, int32_t stride, const double* alpha_acc, int32_t c, double count, float* dgamma, float* dbeta, float* dalpha,
                                         float* mean_dz, float* mean_dzx, float* dres_bias, void* stream) {
  VSSEG_CHECK(sums && alpha_acc && dgamma && dbeta && dalpha && mean_dz && mean_dzx && c > 0, "vsseg_bn_act_bwd_finalize: bad arguments");
  VSSEG_FX_FLAG(fxflag, "vsseg_bn_act_bwd_finalize");
  hipLaunchKernelGGL(bn_act_bwd_finalize_kernel, dim3(c), dim3(256), 0, as_stream(stream), sums, stride, alpha_acc, c, count, dgamma, dbeta, dalpha, mean_dz, mean_dzx, dres_bias, fxflag);
  VSSEG_LAUNCH_CHECK("vsseg_bn_act_bwd_finalize");
  return VSSEG_OK;
}

// Every thread keeps one 8-channel group (block size is a multiple of the group count), so the 7 per-channel constants live in
// registers and the loop body is address adds + two independent 16-byte load pairs (U = 2 voxels in flight per thread).
template <typename T>
__global__ void bn_act_bwd_apply_kernel(const T* __restrict__ y, int yp, const T* __restrict__ dout, int dp, BnBwdArgs a, const float* __restrict__ mean_dz, const float* __restrict__ mean_dzx,
                                        T* __restrict__ dy, int dyp, int cgs, int64_t nvox, const uint8_t* __restrict__ keep_in) {
  const int64_t gt = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, nthreads = (int64_t)gridDim.x * blockDim.x;
  const int cg = (int)(gt % cgs), c = cg * 8;
  const int64_t vstep = nthreads / cgs;
  const float alpha = *a.alpha, inv_keep = 1.f / (1.f - a.p_drop);
  BnBwdC8 k;  // the arithmetic is bn_bwd.h's: the fused data + weight gradient kernel (mbwd.hip) forms the same values on load
  bn_bwd_consts(k, a.mean, a.invstd, a.gamma, a.scale, a.shift, mean_dz, mean_dzx, c, inv_keep);
  const bool drop = a.p_drop > 0.f;
  if (drop && keep_in == nullptr) dropout_resolve_seed(a.seed, a.salt);
  auto one = [&](const f8& yy, const f8& da, unsigned keep, int64_t v) {
    f8 o;
    bn_bwd_dy8(yy, da, keep, alpha, k, o);
    st8(dy + v * dyp + c, o);
  };
  // mask source decided once, outside the loop (0 no dropout, 1 stored bytes, 2 Philox): see bn_act_bwd_reduce_kernel
  auto sweep = [&](auto src_c) {
    constexpr int SRC = decltype(src_c)::value;
    auto mask_of = [&](int64_t vv) -> unsigned {
      if constexpr (SRC == 0) return 0xffu;
      else if constexpr (SRC == 1) return (unsigned)keep_in[vv * cgs + cg];
      else return dropout_keep8(a.seed, a.salt, (uint64_t)(vv * cgs + cg), a.p_drop);
    };
    int64_t v = gt / cgs;
    for (; v + vstep < nvox; v += 2 * vstep) {
      const f8 y0 = ld8(y + v * yp + c), d0 = ld8(dout + v * dp + c);
      const f8 y1 = ld8(y + (v + vstep) * yp + c), d1 = ld8(dout + (v + vstep) * dp + c);
      const unsigned m0 = mask_of(v), m1 = mask_of(v + vstep);
      one(y0, d0, m0, v);
      one(y1, d1, m1, v + vstep);
    }
    if (v < nvox) one(ld8(y + v * yp + c), ld8(dout + v * dp + c), mask_of(v), v);
  };
  if (!drop) sweep(std::integral_constant<int, 0>{});
  else if (keep_in != nullptr) sweep(std::integral_constant<int, 1>{});
  else sweep(std::integral_constant<int, 2>{});
}

extern "C" int vsseg_bn_act_bwd_apply(vsseg_tensor y, vsseg_tensor dout, const float* mean, const float* invstd, const float* gamma, const float* beta, const float* scale, const float* shift, const float* alpha,
                                      float p_drop, uint64_t seed, uint32_t salt, const float* mean_dz, const float* mean_dzx, vsseg_tensor dy, const uint8_t* keep_in, void* stream) {
  VSSEG_ONE_PART("vsseg_bn_act_bwd_apply", &y, &dout, &dy);
  VSSEG_CHECK(y.ptr && dout.ptr && dy.ptr && y.dtype == dout.dtype && y.dtype == dy.dtype && y.c == dout.c && y.c == dy.c && y.c % 8 == 0 && y.pitch % 8 == 0 && dout.pitch % 8 == 0 && dy.pitch % 8 == 0,
              "vsseg_bn_act_bwd_apply: bad arguments");
  int cgs = y.c / 8, blk = block_for_cgs(cgs);
  VSSEG_CHECK(blk > 0, "vsseg_bn_act_bwd_apply: unsupported channel count %d", y.c);
  int64_t nv = tensor_voxels(y);
  BnBwdArgs a{mean, invstd, gamma, beta, scale, shift, alpha, p_drop, seed, salt};
  DISPATCH_T(y.dtype, hipLaunchKernelGGL(bn_act_bwd_apply_kernel<T>, dim3(grid_for((nv * cgs + 1) / 2, blk, 256 * 8)), dim3(blk), 0, as_stream(stream), (const T*)y.ptr, y.pitch, (const T*)dout.ptr, dout.pitch, a, mean_dz, mean_dzx, (T*)dy.ptr, dy.pitch, cgs, nv, keep_in));
  VSSEG_LAUNCH_CHECK("vsseg_bn_act_bwd_apply");
  return VSSEG_OK;
}

// ------------------------------------------------------------------------------------------------------------
// attention gate  out = x*(1+att)                       ref:params/networks/blocks/attentionblock.py:43-47
// ------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void att_apply_fwd_kernel(const T* __restrict__ x, const T* __restrict__ x2, int xsplit, int xp, const float* __restrict__ att, T* __restrict__ out, int op, int cgs, int64_t nvox) {
  const int64_t total = nvox * cgs;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t v = i / cgs;
    int c = (int)(i - v * cgs) * 8;
    const float g = 1.f + att[v];
    f8 a = c >= xsplit ? ld8(x2 + v * xp + (c - xsplit)) : ld8(x + v * xp + c);  // two-part x: the concat of the skip and upsample tensors
#pragma unroll
    for (int j = 0; j < 8; ++j) a.v[j] *= g;
    st8(out + v * op + c, a);
  }
}
extern "C" int vsseg_att_apply_fwd(vsseg_tensor x, const float* att, vsseg_tensor out, void* stream) {
  VSSEG_CHECK(x.ptr && att && out.ptr && x.dtype == out.dtype && x.c == out.c && x.c % 8 == 0 && x.pitch % 8 == 0 && out.pitch % 8 == 0 && !out.ptr2, "vsseg_att_apply_fwd: bad arguments");
  VSSEG_CHECK(two_part_ok(x), "vsseg_att_apply_fwd: bad two-part x");
  int cgs = x.c / 8;
  int64_t nv = tensor_voxels(x);
  DISPATCH_T(x.dtype, hipLaunchKernelGGL(att_apply_fwd_kernel<T>, dim3(grid_for(nv * cgs, 256)), dim3(256), 0, as_stream(stream), (const T*)x.ptr, (const T*)x.ptr2, split_of(x), x.pitch, att, (T*)out.ptr, out.pitch, cgs, nv));
  VSSEG_LAUNCH_CHECK("vsseg_att_apply_fwd");
  return VSSEG_OK;
}

// G lanes per voxel (G = next power of two >= channel groups): dx = dout*(1+att) (optionally +=),
// dpre = (sum_c dout*x + datt_ext) * att*(1-att) in channel 0 of an 8-wide row; sum(dpre) -> bias gradient of attention conv2
template <typename T, bool ACC, int G>
__global__ void att_apply_bwd_kernel(const T* __restrict__ x0, const T* __restrict__ x1, int xsplit, int xp, const float* __restrict__ att, const T* __restrict__ dout, int dp, const float* __restrict__ datt_ext,
                                     T* __restrict__ dx0, T* __restrict__ dx1, int dxsplit, int dxp, T* __restrict__ dpre, int dprep, int cgs, int64_t nvox, float* __restrict__ dbias, T* __restrict__ dpre1, int skip_dx) {
  const int sub = threadIdx.x % G;
  // two-part x / dx (skip-connection concat and its gradient): this lane's 8-channel group lies in one part
  const T* x = sub * 8 >= xsplit ? x1 - xsplit : x0;
  T* dx = sub * 8 >= dxsplit ? dx1 - dxsplit : dx0;
  const int64_t vstep = (int64_t)gridDim.x * (blockDim.x / G);
  float bsum = 0.f;
  for (int64_t v0 = blockIdx.x * (int64_t)(blockDim.x / G); v0 < nvox; v0 += vstep) {  // all lanes iterate together (shuffles need full groups)
    const int64_t v = v0 + threadIdx.x / G;
    const bool live = v < nvox;
    float dot = 0.f, a = 0.f;
    if (live) {
      a = att[v];
      if (sub < cgs) {
        const float g = 1.f + a;
        f8 xx = ld8(x + v * xp + sub * 8), d = ld8(dout + v * dp + sub * 8), o;
        if (ACC) o = ld8(dx + v * dxp + sub * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          dot += d.v[j] * xx.v[j];
          o.v[j] = ACC ? o.v[j] + d.v[j] * g : d.v[j] * g;
        }
        if (!skip_dx) st8(dx + v * dxp + sub * 8, o);  // skip_dx: d(x) is produced by the attention conv's data gradient (VSSEG_RES_GATE)
      }
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
    if (live && sub == 0) {
      if (datt_ext) dot += datt_ext[v];
      const float r = dot * a * (1.f - a);
      if (dpre) st8(dpre + v * dprep, f8{{r, 0, 0, 0, 0, 0, 0, 0}});  // (nullptr: every consumer reads the compact copy)
      if (dpre1) Elem<T>::st(dpre1 + v, r);  // compact 1-channel copy for the z-folded data gradient of the sigmoid convolution
      bsum += r;
    }
  }
  if (dbias) {
    bsum = wave_sum(bsum);
    __shared__ float part[16];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = bsum;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += part[w];
      atomicAdd(dbias, t);
    }
  }
}
template <typename T, bool ACC> static void att_bwd_launch(int G, dim3 g, dim3 b, hipStream_t s, vsseg_tensor x, const float* att, const T* dout, int dp, const float* de, vsseg_tensor dx, T* dpre, int dprep, int cgs, int64_t nv, float* dbias, T* dpre1, int skip_dx) {
#define VSSEG_ATT_BWD(GG) hipLaunchKernelGGL((att_apply_bwd_kernel<T, ACC, GG>), g, b, 0, s, (const T*)x.ptr, (const T*)x.ptr2, split_of(x), x.pitch, att, dout, dp, de, (T*)dx.ptr, (T*)dx.ptr2, split_of(dx), dx.pitch, dpre, dprep, cgs, nv, dbias, dpre1, skip_dx)
  switch (G) {
    case 1: VSSEG_ATT_BWD(1); break;
    case 2: VSSEG_ATT_BWD(2); break;
    case 4: VSSEG_ATT_BWD(4); break;
    case 8: VSSEG_ATT_BWD(8); break;
    case 16: VSSEG_ATT_BWD(16); break;
    default: VSSEG_ATT_BWD(32); break;
  }
#undef VSSEG_ATT_BWD
}
extern "C" int vsseg_att_apply_bwd(vsseg_tensor x, const float* att, vsseg_tensor dout, const float* datt_ext, vsseg_tensor dx, int32_t accumulate_dx, vsseg_tensor dpre, float* dbias, void* dpre1, void* stream) {
  VSSEG_CHECK(x.ptr && att && dout.ptr && dx.ptr && (dpre.ptr || dpre1) && x.dtype == dout.dtype && x.dtype == dx.dtype && (!dpre.ptr || (x.dtype == dpre.dtype && dpre.c == 8 && dpre.pitch % 8 == 0)) && x.c == dout.c && x.c == dx.c && x.c % 8 == 0 &&
                  x.pitch % 8 == 0 && dout.pitch % 8 == 0 && dx.pitch % 8 == 0 && x.c <= 256,
              "vsseg_att_apply_bwd: bad arguments");
  int cgs = x.c / 8;
  int G = 1;
  while (G < cgs) G *= 2;
  int64_t nv = tensor_voxels(x);
  dim3 g(grid_for(nv * G, 256)), b(256);
  hipStream_t s = as_stream(stream);
  VSSEG_CHECK(two_part_ok(x) && two_part_ok(dx) && !dout.ptr2 && !dpre.ptr2, "vsseg_att_apply_bwd: bad two-part tensor");
  const int skip_dx = accumulate_dx == 2;  // 2: do not produce d(x) here at all
  DISPATCH_T(x.dtype, if (accumulate_dx == 1) att_bwd_launch<T, true>(G, g, b, s, x, att, (const T*)dout.ptr, dout.pitch, datt_ext, dx, (T*)dpre.ptr, dpre.pitch, cgs, nv, dbias, (T*)dpre1, 0);
             else att_bwd_launch<T, false>(G, g, b, s, x, att, (const T*)dout.ptr, dout.pitch, datt_ext, dx, (T*)dpre.ptr, dpre.pitch, cgs, nv, dbias, (T*)dpre1, skip_dx));
  VSSEG_LAUNCH_CHECK("vsseg_att_apply_bwd");
  return VSSEG_OK;
}

// ------------------------------------------------------------------------------------------------------------
// per-channel sum (bias gradients of convolutions that are not followed by BatchNorm): 8 channels per thread
// ------------------------------------------------------------------------------------------------------------
template <typename T> __global__ void channel_sum_kernel(const T* __restrict__ t, int pitch, int c, int cgs, int64_t nvox, float* __restrict__ out) {
  extern __shared__ float red[];
  for (int i = threadIdx.x; i < cgs * 8; i += blockDim.x) red[i] = 0.f;
  __syncthreads();
  const int64_t gt = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, nthreads = (int64_t)gridDim.x * blockDim.x;
  const int cg = (int)(gt % cgs);
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t v = gt / cgs; v < nvox; v += nthreads / cgs) {
    f8 a = ld8(t + v * pitch + cg * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] += a.v[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) atomicAdd(&red[cg * 8 + j], s[j]);
  __syncthreads();
  for (int i = threadIdx.x; i < c; i += blockDim.x) atomicAdd(&out[i], red[i]);
}
extern "C" int vsseg_channel_sum(vsseg_tensor t, float* out, void* stream) {
  VSSEG_ONE_PART("vsseg_channel_sum", &t);
  VSSEG_CHECK(t.ptr && out && t.c >= 1 && t.c <= 256 && t.pitch % 8 == 0 && (t.c + 7) / 8 * 8 <= t.pitch, "vsseg_channel_sum: bad arguments (the row must hold the channels rounded up to 8)");
  int cgs = (t.c + 7) / 8, blk = block_for_cgs(cgs);
  VSSEG_CHECK(blk > 0, "vsseg_channel_sum: unsupported channel count");
  int64_t nv = tensor_voxels(t);
  int grid = grid_for(nv * cgs, blk, 2048);
  DISPATCH_T(t.dtype, hipLaunchKernelGGL(channel_sum_kernel<T>, dim3(grid), dim3(blk), cgs * 8 * sizeof(float), as_stream(stream), (const T*)t.ptr, t.pitch, t.c, cgs, nv, out));
  VSSEG_LAUNCH_CHECK("vsseg_channel_sum");
  return VSSEG_OK;
}

__global__ void merge_residual_grads_kernel(const float* dw, const float* db, float* dwr, float* dbr, int cout, int cin, int ktaps, int centre) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cout * cin) dwr[i] += dw[(int64_t)i * ktaps + centre];
  if (i < cout) dbr[i] += db[i];
}
extern "C" int vsseg_merge_residual_grads(const float* dw, const float* db, float* dwr, float* dbr, int32_t cout, int32_t cin, int32_t ktaps, int32_t centre, void* stream) {
  VSSEG_CHECK(dw && db && dwr && dbr && cout > 0 && cin > 0 && centre >= 0 && centre < ktaps, "vsseg_merge_residual_grads: bad arguments");
  hipLaunchKernelGGL(merge_residual_grads_kernel, dim3((cout * cin + 255) / 256), dim3(256), 0, as_stream(stream), dw, db, dwr, dbr, cout, cin, ktaps, centre);
  VSSEG_LAUNCH_CHECK("vsseg_merge_residual_grads");
  return VSSEG_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Adam over the flat fp32 parameter buffer           torch.optim.Adam(lr, weight_decay) — ref:params/VSparams.py:388-391
// ------------------------------------------------------------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2, float gscale) {
  const float step_size = lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float pp = p[i];
    float gg = g[i] * gscale + wd * pp;
    float mm = b1 * m[i] + (1.f - b1) * gg;
    float vv = b2 * v[i] + (1.f - b2) * gg * gg;
    m[i] = mm;
    v[i] = vv;
    p[i] = pp - step_size * mm / (sqrtf(vv) * inv_sqrt_bc2 + eps);
  }
}
extern "C" int vsseg_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float wd, float bc1, float bc2, float gscale, void* stream) {
  VSSEG_CHECK(p && g && m && v && n >= 0, "vsseg_adam: bad arguments");
  if (n == 0) return VSSEG_OK;
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), p, g, m, v, n, lr, beta1, beta2, eps, wd, bc1, bc2, gscale);
  VSSEG_LAUNCH_CHECK("vsseg_adam");
  return VSSEG_OK;
}
