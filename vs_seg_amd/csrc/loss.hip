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
  __shared__ double sh[16][12];
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

// Launch shape of the three reductions below: every workgroup ends with one fixed-point atomic per sum, all of a launch's sums share one or two cache lines, and the
// atomics of one LINE retire one after the other (~4 ns each: with the 1024 workgroups per sample of the first version — 12288 atomics into 96 bytes — the
// attention-map sums of the 96x32x128 level took 49 us for 6 MB, profiles/r06_kernel_stats.txt).  ~512 workgroups of 512 threads in total (2 per CU, 16 waves per CU),
// 16-byte loads.
constexpr int DICE_THREADS = 512;
static inline int dice_blocks(int64_t items, int n) {
  int64_t want = (items + DICE_THREADS * 2 - 1) / (DICE_THREADS * 2);
  const int cap = n >= 512 ? 1 : 512 / n;
  return (int)(want < 1 ? 1 : (want > cap ? cap : want));
}

// the six terms of one voxel of the (optionally hardness-weighted) 2-class soft Dice (ref :279-283)
__device__ __forceinline__ void dice_pred_terms(float l0, float l1, float lab, int hardness, float* t) {
  const float m = fmaxf(l0, l1), e0 = __expf(l0 - m), e1 = __expf(l1 - m), inv = 1.f / (e0 + e1);
  const float p0 = e0 * inv, p1 = e1 * inv;
  const int cls = (int)(long long)lab;
  const float g1 = cls == 1 ? 1.f : 0.f, g0 = cls == 0 ? 1.f : 0.f;
  float w0 = 1.f, w1 = 1.f;
  if (hardness) { w0 = 0.6f * fabsf(p0 - g0) + 0.4f; w1 = 0.6f * fabsf(p1 - g1) + 0.4f; }
  t[0] += w0 * g0 * p0; t[1] += w0 * g0; t[2] += w0 * p0;
  t[3] += w1 * g1 * p1; t[4] += w1 * g1; t[5] += w1 * p1;
}

// sums[b][c][0..2] = (I, G, P) of the (optionally hardness-weighted) 2-class soft Dice.  A thread adds four voxels in fp32 and keeps its running sums in fp64.
__global__ __launch_bounds__(DICE_THREADS) void dice_pred_sums_kernel(const float* __restrict__ logits, int pitch, const float* __restrict__ label, int64_t nvox, int hardness, double* __restrict__ sums, unsigned* fxflag) {
  const int b = blockIdx.y;
  const float* lg = logits + (int64_t)b * nvox * pitch;
  const float* lb = label + (int64_t)b * nvox;
  double acc[6] = {0, 0, 0, 0, 0, 0};
  const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, nthr = (int64_t)gridDim.x * blockDim.x;
  if (pitch == 2 && (nvox & 3) == 0 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0 && (reinterpret_cast<uintptr_t>(label) & 15) == 0) {
    const float4* lg4 = reinterpret_cast<const float4*>(lg);
    const float4* lb4 = reinterpret_cast<const float4*>(lb);
    for (int64_t i = tid; i < (nvox >> 2); i += nthr) {
      const float4 a = lg4[2 * i], c = lg4[2 * i + 1], g = lb4[i];
      float t[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      dice_pred_terms(a.x, a.y, g.x, hardness, t);
      dice_pred_terms(a.z, a.w, g.y, hardness, t);
      dice_pred_terms(c.x, c.y, g.z, hardness, t);
      dice_pred_terms(c.z, c.w, g.w, hardness, t);
#pragma unroll
      for (int k = 0; k < 6; ++k) acc[k] += (double)t[k];
    }
  } else {
    for (int64_t v = tid; v < nvox; v += nthr) {
      const float2 l = *reinterpret_cast<const float2*>(lg + v * pitch);
      float t[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      dice_pred_terms(l.x, l.y, lb[v], hardness, t);
#pragma unroll
      for (int k = 0; k < 6; ++k) acc[k] += (double)t[k];
    }
  }
  block_reduce_add(acc, 6, sums + (int64_t)b * 6, VSSEG_FX_DICE, fxflag);
}
extern "C" int vsseg_dice_pred_sums(const float* logits, int32_t pitch, const float* label, int32_t n, int64_t nvox, int32_t hardness, double* sums, void* stream) {
  VSSEG_CHECK(logits && label && sums && pitch >= 2 && pitch % 2 == 0 && n >= 1, "vsseg_dice_pred_sums: bad arguments");
  dim3 g(dice_blocks(nvox / 4, n), n);
  VSSEG_FX_FLAG(fxflag, "vsseg_dice_pred_sums");
  hipLaunchKernelGGL(dice_pred_sums_kernel, g, dim3(DICE_THREADS), 0, as_stream(stream), logits, pitch, label, nvox, hardness, sums, fxflag);
  VSSEG_LAUNCH_CHECK("vsseg_dice_pred_sums");
  return VSSEG_OK;
}

// sums[b][0..2] = (I, G, P) of the single-channel Dice between an attention map and the pooled label
__global__ __launch_bounds__(DICE_THREADS) void dice_att_sums_kernel(const float* __restrict__ att, const float* __restrict__ label, int64_t nvox, double* __restrict__ sums, unsigned* fxflag) {
  const int b = blockIdx.y;
  const float* a = att + (int64_t)b * nvox;
  const float* lb = label + (int64_t)b * nvox;
  double acc[3] = {0, 0, 0};
  const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, nthr = (int64_t)gridDim.x * blockDim.x;
  if ((nvox & 3) == 0 && (reinterpret_cast<uintptr_t>(att) & 15) == 0 && (reinterpret_cast<uintptr_t>(label) & 15) == 0) {
    const float4* a4 = reinterpret_cast<const float4*>(a);
    const float4* l4 = reinterpret_cast<const float4*>(lb);
    for (int64_t i = tid; i < (nvox >> 2); i += nthr) {
      const float4 p = a4[i], g = l4[i];
      acc[0] += (double)((g.x * p.x + g.y * p.y) + (g.z * p.z + g.w * p.w));
      acc[1] += (double)((g.x + g.y) + (g.z + g.w));
      acc[2] += (double)((p.x + p.y) + (p.z + p.w));
    }
  } else {
    for (int64_t v = tid; v < nvox; v += nthr) {
      const float p = a[v], g = lb[v];
      acc[0] += (double)(g * p); acc[1] += (double)g; acc[2] += (double)p;
    }
  }
  block_reduce_add(acc, 3, sums + (int64_t)b * 3, VSSEG_FX_DICE, fxflag);
}
extern "C" int vsseg_dice_att_sums(const float* att, const float* label, int32_t n, int64_t nvox, double* sums, void* stream) {
  VSSEG_CHECK(att && label && sums && n >= 1, "vsseg_dice_att_sums: bad arguments");
  dim3 g(dice_blocks(nvox / 4, n), n);
  VSSEG_FX_FLAG(fxflag, "vsseg_dice_att_sums");
  hipLaunchKernelGGL(dice_att_sums_kernel, g, dim3(DICE_THREADS), 0, as_stream(stream), att, label, nvox, sums, fxflag);
  VSSEG_LAUNCH_CHECK("vsseg_dice_att_sums");
  return VSSEG_OK;
}

// One pass over one level of the supervision pyramid whose next level pools (2, 2, 1) (the two finest levels of the 2.5D network): the attention-map sums against the
// level's label, the next level's label G' = MaxPool3d((2,2,1))(G) (ref :268-277) and, at the finest level (PRED), the sums of the softmax Dice of the logits — what
// vsseg_dice_pred_sums + vsseg_dice_att_sums + vsseg_maxpool_label do in three passes that read the label three times (0.63 GB instead of 0.43 at 4 x 384x128x128).
// A thread owns a 2 x 2 x 4 block of voxels = four z-quads of one pooling window row: sixteen 16-byte loads in flight, one 16-byte store of pooled labels.
template <bool PRED>
__global__ __launch_bounds__(DICE_THREADS) void dice_level_kernel(const float* __restrict__ logits, const float* __restrict__ att, const float* __restrict__ label, int X, int Y, int Z, int hardness,
                                                                    double* __restrict__ pred_sums, double* __restrict__ att_sums, float* __restrict__ pooled, unsigned* fxflag) {
  const int b = blockIdx.y;
  const int64_t nvox = (int64_t)X * Y * Z;
  const int X2 = X >> 1, Y2 = Y >> 1, Zq = Z >> 2;
  const float4* lg4 = PRED ? reinterpret_cast<const float4*>(logits + (int64_t)b * nvox * 2) : nullptr;
  const float4* at4 = reinterpret_cast<const float4*>(att + (int64_t)b * nvox);
  const float4* lb4 = reinterpret_cast<const float4*>(label + (int64_t)b * nvox);
  float4* po4 = pooled ? reinterpret_cast<float4*>(pooled + (int64_t)b * (nvox >> 2)) : nullptr;
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  const int64_t items = (int64_t)X2 * Y2 * Zq, nthr = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < items; i += nthr) {
    const int zq = (int)(i % Zq);
    const int64_t r = i / Zq;
    const int y2 = (int)(r % Y2), x2 = (int)(r / Y2);
    float4 g[4], p[4], la[4], lc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t q = ((int64_t)(2 * x2 + (j >> 1)) * Y + 2 * y2 + (j & 1)) * Zq + zq;  // float4 index of the quad
      g[j] = lb4[q];
      p[j] = at4[q];
      if constexpr (PRED) { la[j] = lg4[2 * q]; lc[j] = lg4[2 * q + 1]; }
    }
    float t[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ai = 0.f, ag = 0.f, ap = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (PRED) {
        dice_pred_terms(la[j].x, la[j].y, g[j].x, hardness, t);
        dice_pred_terms(la[j].z, la[j].w, g[j].y, hardness, t);
        dice_pred_terms(lc[j].x, lc[j].y, g[j].z, hardness, t);
        dice_pred_terms(lc[j].z, lc[j].w, g[j].w, hardness, t);
      }
      ai += (g[j].x * p[j].x + g[j].y * p[j].y) + (g[j].z * p[j].z + g[j].w * p[j].w);
      ag += (g[j].x + g[j].y) + (g[j].z + g[j].w);
      ap += (p[j].x + p[j].y) + (p[j].z + p[j].w);
    }
    if constexpr (PRED) {
#pragma unroll
      for (int k = 0; k < 6; ++k) acc[k] += (double)t[k];
    }
    acc[6] += (double)ai; acc[7] += (double)ag; acc[8] += (double)ap;
    if (po4) {
      float4 m;
      m.x = fmaxf(fmaxf(g[0].x, g[1].x), fmaxf(g[2].x, g[3].x));
      m.y = fmaxf(fmaxf(g[0].y, g[1].y), fmaxf(g[2].y, g[3].y));
      m.z = fmaxf(fmaxf(g[0].z, g[1].z), fmaxf(g[2].z, g[3].z));
      m.w = fmaxf(fmaxf(g[0].w, g[1].w), fmaxf(g[2].w, g[3].w));
      po4[((int64_t)x2 * Y2 + y2) * Zq + zq] = m;
    }
  }
  // (block_reduce_add: threads 0..nvals-1 add one value each — the two groups go to their own buffers in two calls)
  if constexpr (PRED) block_reduce_add(acc, 6, pred_sums + (int64_t)b * 6, VSSEG_FX_DICE, fxflag);
  if constexpr (PRED) __syncthreads();
  block_reduce_add(acc + 6, 3, att_sums + (int64_t)b * 3, VSSEG_FX_DICE, fxflag);
}
extern "C" int vsseg_dice_level_sums(const float* logits, const float* att, const float* label, int32_t n, const int32_t dims[3], int32_t hardness, double* pred_sums, double* att_sums, float* pooled, void* stream) {
  VSSEG_CHECK(att && label && att_sums && n >= 1 && (!logits || pred_sums), "vsseg_dice_level_sums: bad arguments");
  VSSEG_CHECK(dims[0] >= 2 && dims[1] >= 2 && dims[2] >= 4 && dims[0] % 2 == 0 && dims[1] % 2 == 0 && dims[2] % 4 == 0, "vsseg_dice_level_sums: %d x %d x %d is not made of 2 x 2 x 4 blocks", dims[0], dims[1], dims[2]);
  VSSEG_CHECK(((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(att) | reinterpret_cast<uintptr_t>(label) | reinterpret_cast<uintptr_t>(pooled)) & 15) == 0, "vsseg_dice_level_sums: operands must be 16-byte aligned");
  const int64_t items = (int64_t)(dims[0] / 2) * (dims[1] / 2) * (dims[2] / 4);
  dim3 g(dice_blocks(items * 2, n), n);  // (an item is sixteen voxels: twice the loads of the other reductions' eight per thread and round)
  VSSEG_FX_FLAG(fxflag, "vsseg_dice_level_sums");
  if (logits) hipLaunchKernelGGL(dice_level_kernel<true>, g, dim3(DICE_THREADS), 0, as_stream(stream), logits, att, label, dims[0], dims[1], dims[2], hardness, pred_sums, att_sums, pooled, fxflag);
  else hipLaunchKernelGGL(dice_level_kernel<false>, g, dim3(DICE_THREADS), 0, as_stream(stream), logits, att, label, dims[0], dims[1], dims[2], hardness, pred_sums, att_sums, pooled, fxflag);
  VSSEG_LAUNCH_CHECK("vsseg_dice_level_sums");
  return VSSEG_OK;
}

// The coarse levels of the supervision pyramid in ONE launch: level i's label = MaxPool3d(sdims / dims_i)(src) straight from the finest of them (a max-pool of a max-pool
// is the max-pool over the product window) and its attention-map sums.  Level by level this was two launches of a few thousand voxels each, ~6 us apiece whatever their size.
// 1024-thread workgroups: the sums of all levels and samples of this launch share three cache lines and the fixed-point atomics of one LINE retire one after the other:
// a quarter of the workgroups of a 256-thread launch, four times the threads (measured the same 32 us either way: the launch is bound by its dependent loads, not by these)
constexpr int DICE_TAIL_THREADS = 1024;
struct DiceTailK {
  vsseg_dice_tail_desc d;
  int first[VSSEG_DICE_MAX_LEVELS + 1];  // workgroup range of level i: [first[i], first[i+1])
};
__global__ __launch_bounds__(DICE_TAIL_THREADS) void dice_tail_kernel(const DiceTailK k, unsigned* fxflag) {
  const int b = blockIdx.y;
  int l = 0;
  while (l + 1 < k.d.nlevels && (int)blockIdx.x >= k.first[l + 1]) ++l;
  const int nb = k.first[l + 1] - k.first[l], lb = (int)blockIdx.x - k.first[l];
  const int dx = k.d.dims[l][0], dy = k.d.dims[l][1], dz = k.d.dims[l][2];
  const int sx = k.d.sdims[0], sy = k.d.sdims[1], sz = k.d.sdims[2];
  const int rx = sx / dx, ry = sy / dy, rz = sz / dz;
  const int nvox = dx * dy * dz;
  const float* __restrict__ src = k.d.src + (int64_t)b * sx * sy * sz;
  const float* __restrict__ att = k.d.att[l] + (int64_t)b * nvox;
  float* __restrict__ out = k.d.label[l] ? k.d.label[l] + (int64_t)b * nvox : nullptr;
  double acc[3] = {0, 0, 0};
  // T lanes (a power of two <= 64) share the pooling window of one voxel — the coarsest level of the benchmark network pools 8 x 8 x 8 = 512 values per voxel for 768 voxels:
  // one thread per voxel walked them one dependent load after the other (170 us for the launch) — and meet in a butterfly of lane exchanges (max: exact in any order)
  const int win = rx * ry * rz;
  int T = 1;
  while (T < 64 && T * 2 <= win) T *= 2;
  const int vpb = DICE_TAIL_THREADS / T, sub = (int)threadIdx.x % T, slot = (int)threadIdx.x / T;
  const float idz = 1.0f / (float)dz, idy = 1.0f / (float)dy, irz = 1.0f / (float)rz, iry = 1.0f / (float)ry;
  if (win == 1 && (nvox & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(att) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
    // the level IS the source level: dice_att_sums_kernel's loop (16-byte loads, no coordinates)
    for (int i = lb * DICE_TAIL_THREADS + (int)threadIdx.x; i < (nvox >> 2); i += nb * DICE_TAIL_THREADS) {
      const float4 g = reinterpret_cast<const float4*>(src)[i], p = reinterpret_cast<const float4*>(att)[i];
      if (out) reinterpret_cast<float4*>(out)[i] = g;
      acc[0] += (double)((g.x * p.x + g.y * p.y) + (g.z * p.z + g.w * p.w));
      acc[1] += (double)((g.x + g.y) + (g.z + g.w));
      acc[2] += (double)((p.x + p.y) + (p.z + p.w));
    }
  } else
  for (int base = lb * vpb * 4; base < nvox; base += nb * vpb * 4) {  // (block-uniform trip count: every lane takes part in the exchanges)
    // four voxels per lane group and round: their loads are issued together (one voxel per round was a chain of dependent ~1 us loads: 8 rounds, 30 us)
    int iv[4];
    float m[4], p[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + u * vpb + slot;
      iv[u] = i < nvox ? i : -1;
      const int ic = i < nvox ? i : 0, r = vsseg_fdiv(ic, dz, idz), z = ic - r * dz, x = vsseg_fdiv(r, dy, idy), y = r - x * dy;  // (values < 2^22: common.h)
      m[u] = -INFINITY;
      for (int w = sub; w < win; w += T) {
        const int ac = vsseg_fdiv(w, rz, irz), e = w - ac * rz, a = vsseg_fdiv(ac, ry, iry), c = ac - a * ry;
        m[u] = fmaxf(m[u], src[((int64_t)(x * rx + a) * sy + y * ry + c) * sz + z * rz + e]);
      }
      p[u] = att[ic];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      for (int off = T >> 1; off > 0; off >>= 1) m[u] = fmaxf(m[u], __shfl_xor(m[u], off));
      if (iv[u] >= 0 && sub == 0) {
        if (out) out[iv[u]] = m[u];
        acc[0] += (double)(m[u] * p[u]); acc[1] += (double)m[u]; acc[2] += (double)p[u];
      }
    }
  }
  block_reduce_add(acc, 3, k.d.sums + ((int64_t)l * k.d.n + b) * 3, VSSEG_FX_DICE, fxflag);
}
extern "C" int vsseg_dice_tail_sums(const vsseg_dice_tail_desc* d, void* stream) {
  VSSEG_CHECK(d && d->src && d->sums && d->n >= 1 && d->nlevels >= 1 && d->nlevels <= VSSEG_DICE_MAX_LEVELS, "vsseg_dice_tail_sums: bad arguments");
  DiceTailK k;
  k.d = *d;
  int nb = 0;
  for (int l = 0; l < d->nlevels; ++l) {
    VSSEG_CHECK(d->att[l], "vsseg_dice_tail_sums: level %d has no attention map", l);
    bool same = true;
    for (int a = 0; a < 3; ++a) {
      VSSEG_CHECK(d->dims[l][a] >= 1 && d->sdims[a] % d->dims[l][a] == 0, "vsseg_dice_tail_sums: attention pyramid shapes must divide (ref dice_spvPA.py:273)");
      same = same && d->dims[l][a] == d->sdims[a];
    }
    VSSEG_CHECK(d->label[l] || same, "vsseg_dice_tail_sums: level %d pools but has no label buffer", l);
    const int64_t nvox = (int64_t)d->dims[l][0] * d->dims[l][1] * d->dims[l][2];
    VSSEG_CHECK(nvox < (1ll << 22), "vsseg_dice_tail_sums: level %d is too large for the tail launch (>= 2^22 voxels per sample)", l);
    k.first[l] = nb;
    int64_t win = 1, T = 1;
    for (int a = 0; a < 3; ++a) win *= d->sdims[a] / d->dims[l][a];
    while (T < 64 && T * 2 <= win) T *= 2;
    nb += (int)std::min<int64_t>(48, win == 1 ? (nvox + 8191) / 8192 : (nvox * T + 2047) / 2048);  // T lanes per voxel (dice_tail_kernel)
  }
  for (int l = d->nlevels; l <= VSSEG_DICE_MAX_LEVELS; ++l) k.first[l] = nb;
  VSSEG_FX_FLAG(fxflag, "vsseg_dice_tail_sums");
  hipLaunchKernelGGL(dice_tail_kernel, dim3(nb, d->n), dim3(DICE_TAIL_THREADS), 0, as_stream(stream), k, fxflag);
  VSSEG_LAUNCH_CHECK("vsseg_dice_tail_sums");
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

// d(loss)/d(logits) of one voxel: through the Dice sums, the (non-detached) hardness weight (ref :279-283) and the softmax
__device__ __forceinline__ float2 dice_pred_grad(float l0, float l1, float lab, int hardness, float A0, float B0, float A1, float B1) {
  const float m = fmaxf(l0, l1), e0 = __expf(l0 - m), e1 = __expf(l1 - m), inv = 1.f / (e0 + e1);
  const float p0 = e0 * inv, p1 = e1 * inv;
  const int cls = (int)(long long)lab;
  const float g1 = cls == 1 ? 1.f : 0.f, g0 = cls == 0 ? 1.f : 0.f;
  float w0 = 1.f, w1 = 1.f, s0 = 0.f, s1 = 0.f;
  if (hardness) {
    const float d0 = p0 - g0, d1 = p1 - g1;
    w0 = 0.6f * fabsf(d0) + 0.4f; w1 = 0.6f * fabsf(d1) + 0.4f;
    s0 = d0 > 0.f ? 0.6f : (d0 < 0.f ? -0.6f : 0.f);
    s1 = d1 > 0.f ? 0.6f : (d1 < 0.f ? -0.6f : 0.f);
  }
  const float t0 = w0 + p0 * s0, t1 = w1 + p1 * s1;  // d(w*p)/dp
  const float dp0 = A0 * g0 * t0 + B0 * (g0 * s0 + t0);
  const float dp1 = A1 * g1 * t1 + B1 * (g1 * s1 + t1);
  const float dot = dp0 * p0 + dp1 * p1;
  return make_float2(p0 * (dp0 - dot), p1 * (dp1 - dot));
}
// OUT 0: fp32 [voxel][2];  1: bf16 [voxel][2] (the compact operand of the marching backward kernels);  2: bf16 rows of 8 channels, channels 2..7 written as zeros;
// 3: fp32, any pitch (channels >= 2 are not touched).  1-3 are the layouts the training plan stages the gradient in (vsseg_dice_pred_bwd_to): written here, the
// fp32 tensor and the cast pass over it (0.3 GB read + written per step at 4 x 384x128x128) do not exist.  Same values: the cast was the same round-to-nearest-even.
template <int OUT> __global__ __launch_bounds__(256) void dice_pred_bwd_kernel(const float* __restrict__ logits, int pitch, const float* __restrict__ label, int64_t nvox, int hardness, const float* __restrict__ coef,
                                                                                 const float* __restrict__ gscale, void* __restrict__ dst, int dpitch) {
  const int b = blockIdx.y;
  const float* lg = logits + (int64_t)b * nvox * pitch;
  const float* lb = label + (int64_t)b * nvox;
  const float gs = gscale ? *gscale : 1.f;
  const float A0 = coef[(b * 2 + 0) * 2] * gs, B0 = coef[(b * 2 + 0) * 2 + 1] * gs, A1 = coef[(b * 2 + 1) * 2] * gs, B1 = coef[(b * 2 + 1) * 2 + 1] * gs;
  const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, nthr = (int64_t)gridDim.x * blockDim.x;
  auto store = [&](int64_t v, float2 g) {
    const int64_t e = ((int64_t)b * nvox + v) * dpitch;
    if constexpr (OUT == 1) *reinterpret_cast<unsigned*>(reinterpret_cast<bf16_t*>(dst) + e) = f2bf2(g.x, g.y);
    else if constexpr (OUT == 2) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(dst) + e) = make_uint4(f2bf2(g.x, g.y), 0u, 0u, 0u);
    else *reinterpret_cast<float2*>(reinterpret_cast<float*>(dst) + e) = g;
  };
  if (pitch == 2 && (nvox & 3) == 0 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0 && (reinterpret_cast<uintptr_t>(label) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
    const float4* lg4 = reinterpret_cast<const float4*>(lg);
    const float4* lb4 = reinterpret_cast<const float4*>(lb);
    for (int64_t i = tid; i < (nvox >> 2); i += nthr) {
      const float4 a = lg4[2 * i], c = lg4[2 * i + 1], g = lb4[i];
      const float2 r0 = dice_pred_grad(a.x, a.y, g.x, hardness, A0, B0, A1, B1), r1 = dice_pred_grad(a.z, a.w, g.y, hardness, A0, B0, A1, B1);
      const float2 r2 = dice_pred_grad(c.x, c.y, g.z, hardness, A0, B0, A1, B1), r3 = dice_pred_grad(c.z, c.w, g.w, hardness, A0, B0, A1, B1);
      if constexpr (OUT == 0) {
        float4* o = reinterpret_cast<float4*>(reinterpret_cast<float*>(dst) + ((int64_t)b * nvox + 4 * i) * 2);
        o[0] = make_float4(r0.x, r0.y, r1.x, r1.y);
        o[1] = make_float4(r2.x, r2.y, r3.x, r3.y);
      } else if constexpr (OUT == 1) {
        *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(dst) + ((int64_t)b * nvox + 4 * i) * 2) = make_uint4(f2bf2(r0.x, r0.y), f2bf2(r1.x, r1.y), f2bf2(r2.x, r2.y), f2bf2(r3.x, r3.y));
      } else {
        store(4 * i, r0); store(4 * i + 1, r1); store(4 * i + 2, r2); store(4 * i + 3, r3);
      }
    }
  } else {
    for (int64_t v = tid; v < nvox; v += nthr) {
      const float2 l = *reinterpret_cast<const float2*>(lg + v * pitch);
      store(v, dice_pred_grad(l.x, l.y, lb[v], hardness, A0, B0, A1, B1));
    }
  }
}
static int dice_pred_bwd_launch(const float* logits, int32_t pitch, const float* label, int32_t n, int64_t nvox, int32_t hardness, const float* coef, const float* gscale, void* dst, int out, int dpitch, void* stream,
                                const char* who) {
  VSSEG_CHECK(logits && label && coef && dst && pitch >= 2 && pitch % 2 == 0 && n >= 1, "%s: bad arguments", who);
  dim3 g(grid_for((nvox + 3) / 4, 256, 2048), n), t(256);
  hipStream_t s = as_stream(stream);
  if (out == 0) hipLaunchKernelGGL(dice_pred_bwd_kernel<0>, g, t, 0, s, logits, pitch, label, nvox, hardness, coef, gscale, dst, 2);
  else if (out == 1) hipLaunchKernelGGL(dice_pred_bwd_kernel<1>, g, t, 0, s, logits, pitch, label, nvox, hardness, coef, gscale, dst, 2);
  else if (out == 2) hipLaunchKernelGGL(dice_pred_bwd_kernel<2>, g, t, 0, s, logits, pitch, label, nvox, hardness, coef, gscale, dst, 8);
  else hipLaunchKernelGGL(dice_pred_bwd_kernel<3>, g, t, 0, s, logits, pitch, label, nvox, hardness, coef, gscale, dst, dpitch);
  VSSEG_LAUNCH_CHECK(who);
  return VSSEG_OK;
}
extern "C" int vsseg_dice_pred_bwd(const float* logits, int32_t pitch, const float* label, int32_t n, int64_t nvox, int32_t hardness, const float* coef, const float* gscale, float* dlogits, void* stream) {
  return dice_pred_bwd_launch(logits, pitch, label, n, nvox, hardness, coef, gscale, dlogits, 0, 2, stream, "vsseg_dice_pred_bwd");
}
extern "C" int vsseg_dice_pred_bwd_to(const float* logits, int32_t pitch, const float* label, int32_t n, int64_t nvox, int32_t hardness, const float* coef, const float* gscale, vsseg_tensor dst, void* stream) {
  VSSEG_CHECK(dst.ptr && !dst.ptr2 && dst.c == 2 && dst.pitch >= 2 && tensor_voxels(dst) == (int64_t)n * nvox, "vsseg_dice_pred_bwd_to: destination is not a 2-channel tensor of %lld voxels", (long long)n * nvox);
  int out;
  if (dst.dtype == VSSEG_BF16 && dst.pitch == 2) out = 1;
  else if (dst.dtype == VSSEG_BF16 && dst.pitch == 8 && (reinterpret_cast<uintptr_t>(dst.ptr) & 15) == 0) out = 2;
  else if (dst.dtype == VSSEG_F32 && dst.pitch % 2 == 0) out = dst.pitch == 2 ? 0 : 3;
  else { vsseg_set_error("vsseg_dice_pred_bwd_to: destination layout (dtype %d, pitch %d) is not one the training plan stages the gradient in", dst.dtype, dst.pitch); return VSSEG_EINVAL; }
  return dice_pred_bwd_launch(logits, pitch, label, n, nvox, hardness, coef, gscale, dst.ptr, out, dst.pitch, stream, "vsseg_dice_pred_bwd_to");
}

__global__ void dice_att_bwd_kernel(const float* __restrict__ label, int64_t nvox, const float* __restrict__ coef, const float* __restrict__ gscale, float* __restrict__ datt) {
  const int b = blockIdx.y;
  const float gs = gscale ? *gscale : 1.f;
  const float A = coef[b * 2] * gs, B = coef[b * 2 + 1] * gs;
  const float* lb = label + (int64_t)b * nvox;
  float* o = datt + (int64_t)b * nvox;
  const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, nthr = (int64_t)gridDim.x * blockDim.x;
  if ((nvox & 3) == 0 && (reinterpret_cast<uintptr_t>(label) & 15) == 0 && (reinterpret_cast<uintptr_t>(datt) & 15) == 0) {
    for (int64_t i = tid; i < (nvox >> 2); i += nthr) {
      const float4 g = reinterpret_cast<const float4*>(lb)[i];
      reinterpret_cast<float4*>(o)[i] = make_float4(A * g.x + B, A * g.y + B, A * g.z + B, A * g.w + B);
    }
  } else {
    for (int64_t v = tid; v < nvox; v += nthr) o[v] = A * lb[v] + B;
  }
}
extern "C" int vsseg_dice_att_bwd(const float* label, int32_t n, int64_t nvox, const float* coef, float inv_levels, const float* gscale, float* datt, void* stream) {
  (void)inv_levels;  // the 1/L factor is already folded into coef by vsseg_dice_finalize
  VSSEG_CHECK(label && coef && datt, "vsseg_dice_att_bwd: bad arguments");
  dim3 g(grid_for((nvox + 3) / 4, 256, 2048), n);
  hipLaunchKernelGGL(dice_att_bwd_kernel, g, dim3(256), 0, as_stream(stream), label, nvox, coef, gscale, datt);
  VSSEG_LAUNCH_CHECK("vsseg_dice_att_bwd");
  return VSSEG_OK;
}

// d(loss)/d(att) of several levels in one launch (the coarse levels: a few thousand voxels each)
struct DiceBwdLevelsK {
  vsseg_dice_bwd_levels_desc d;
  int first[VSSEG_DICE_MAX_LEVELS + 1];
};
__global__ __launch_bounds__(256) void dice_att_bwd_levels_kernel(const DiceBwdLevelsK k) {
  const int b = blockIdx.y;
  int l = 0;
  while (l + 1 < k.d.nlevels && (int)blockIdx.x >= k.first[l + 1]) ++l;
  const int nb = k.first[l + 1] - k.first[l], lb = (int)blockIdx.x - k.first[l];
  const float gs = k.d.gscale ? *k.d.gscale : 1.f;
  const float A = k.d.coef[l][b * 2] * gs, B = k.d.coef[l][b * 2 + 1] * gs;
  const int64_t nvox = k.d.nvox[l];
  const float* lb_ = k.d.label[l] + (int64_t)b * nvox;
  float* o = k.d.datt[l] + (int64_t)b * nvox;
  for (int64_t v = (int64_t)lb * 256 + threadIdx.x; v < nvox; v += (int64_t)nb * 256) o[v] = A * lb_[v] + B;
}
extern "C" int vsseg_dice_att_bwd_levels(const vsseg_dice_bwd_levels_desc* d, void* stream) {
  VSSEG_CHECK(d && d->n >= 1 && d->nlevels >= 1 && d->nlevels <= VSSEG_DICE_MAX_LEVELS, "vsseg_dice_att_bwd_levels: bad arguments");
  DiceBwdLevelsK k;
  k.d = *d;
  int nb = 0;
  for (int l = 0; l < d->nlevels; ++l) {
    VSSEG_CHECK(d->label[l] && d->datt[l] && d->coef[l] && d->nvox[l] >= 1, "vsseg_dice_att_bwd_levels: level %d is incomplete", l);
    k.first[l] = nb;
    nb += (int)std::min<int64_t>(256, (d->nvox[l] + 2047) / 2048);
  }
  for (int l = d->nlevels; l <= VSSEG_DICE_MAX_LEVELS; ++l) k.first[l] = nb;
  hipLaunchKernelGGL(dice_att_bwd_levels_kernel, dim3(nb, d->n), dim3(256), 0, as_stream(stream), k);
  VSSEG_LAUNCH_CHECK("vsseg_dice_att_bwd_levels");
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
