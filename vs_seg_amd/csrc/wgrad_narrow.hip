// Weight gradient of the convolutions with ONE input or ONE output channel on the two finest levels (network input 1 -> 16, 3x3x1 and 1x1x1;
// attention sigmoid convolutions 16 -> 1 and 32 -> 1, 3x3x1; ref:params/networks/nets/unet2d5_spvPA.py:56-93, attentionblock.py:10-30;
// SURVEY §8a rows 0, 1, 42, 47) as a bandwidth reduction instead of an MFMA launch:
//
//     dW[c][tap (dx, dy)] = sum_v T[v][c] * s[v + sign * (dx, dy)]         T: the C-channel tensor, s: the one-channel field
//
//   1 -> C:  T = dY (C output channels), s = x (the network input),   sign = +1       (dW[c][0][tap])
//   C -> 1:  T = x  (C input channels),  s = dY (one real channel),   sign = -1       (dW[0][c][tap]; substitution u = v + off)
//
// The MFMA kernel (wgrad.hip) zero-extends the one-channel operand to an 8-channel K group and builds both operands with transpose reads:
// 0.41-0.43 ms per launch at full resolution, 2.0-2.1 TB/s of its algorithmic bytes, the matrix cores multiplying zeros.  Here a thread owns
// one 8-channel group of 4 y-consecutive voxels: 4 coalesced 16-byte loads of T and the 3 x 6 values of s its 9 taps touch (the y taps of
// neighbouring voxels overlap), 288 FMAs, 72 accumulators kept in registers over all its voxels; per workgroup one LDS reduction and one
// slab of partial sums (plain stores), summed into dw by a second tiny kernel — 1024 workgroups x 144 fp32 atomics on the same 144
// addresses measured 0.72 ms for the whole launch, slower than the MFMA kernel.  T is read exactly once, s nine times out of L1/L2.
#define VSSEG_NT_LOADS  // the C-channel tensor is read exactly once (ld8): non-temporal; the one-channel field is re-read nine times and stays cached
#include "common.h"
#include "bn_bwd.h"

struct WnK {
  const void* t;
  const void* s;
  float* dw;
  float* slab;  // [gridDim.x][NT * C (+ 1)] partial sums (+ the bias gradient sum_v s[v] of the C -> 1 case)
  int bias;     // 1: slab rows carry one more element, sum of the one-channel field over this workgroup's voxels
  int tpitch, C, cgs;
  int n, X, Y, Z, yq;  // yq = Y / 4
  int sign;
  int64_t stride_c;
  int64_t items;  // n * X * yq * Z work items per channel group
  // BN (vsseg_wgrad_narrow_bn): T = d(conv output) is formed ON LOAD from the convolution output `t`, the gradient of the block output `da` and the forward's
  // keep-mask bytes — the second pass of the BatchNorm -> Dropout -> PReLU backward (bn_bwd.h, bit-identical to vsseg_bn_act_bwd_apply, which is then not launched)
  const void* da;
  const unsigned char* keep;
  const float *mean, *invstd, *gamma, *scale, *shift, *alpha, *mean_dz, *mean_dzx;
  float inv_keep;
  int dapitch;
};

template <typename T> __device__ __forceinline__ float wn_ld(const T* p);
template <> __device__ __forceinline__ float wn_ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float wn_ld<bf16_t>(const bf16_t* p) { return bf2f(*p); }

template <typename T, int K3, bool BN = false>  // K3: 3 = 3x3x1 taps, 1 = 1x1x1
__global__ __launch_bounds__(256) void wgrad_narrow_kernel(const WnK k) {
  constexpr int NT = K3 * K3, ROWS = K3 == 3 ? 6 : 4, R = K3 / 2;
  __shared__ float red[4][NT * 64 + 1];  // one row per wave, summed in wave order (run-to-run bit-identical)
  const int tid = threadIdx.x, C = k.C, cgs = k.cgs;
  float bsum = 0.f;
  const int64_t gt = blockIdx.x * 256ll + tid, nthreads = (int64_t)gridDim.x * 256;
  const int cg = (int)(gt % cgs);
  const int64_t step = nthreads / cgs;
  const T* tp = reinterpret_cast<const T*>(k.t) + cg * 8;
  const T* sp = reinterpret_cast<const T*>(k.s);
  const T* dap = BN ? reinterpret_cast<const T*>(k.da) + cg * 8 : nullptr;
  BnBwdC8 bc;
  float alpha = 0.f;
  if constexpr (BN) {
    bn_bwd_consts(bc, k.mean, k.invstd, k.gamma, k.scale, k.shift, k.mean_dz, k.mean_dzx, cg * 8, k.inv_keep);
    alpha = *k.alpha;
  }
  const int X = k.X, Y = k.Y, Z = k.Z;
  float acc[NT][8];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[t][c] = 0.f;
  // Work item = (n, x, y quad, z), z fastest: the lanes of a wave read consecutive voxels.  A thread's items are `step` apart; its coordinates
  // advance by the mixed-radix digits of `step` with carries (the first version decoded a 64-bit item index with six divisions per item and
  // formed eighteen 64-bit voxel addresses: ~900 instructions around 288 FMAs, 1.1 TB/s)
  const unsigned first = (unsigned)(gt / cgs), ustep = (unsigned)step, uitems = (unsigned)k.items;
  const int YQ = k.yq, ZY = Z * Y;
  int z = (int)(first % (unsigned)Z), yq = (int)((first / (unsigned)Z) % (unsigned)YQ), x = (int)((first / (unsigned)(Z * YQ)) % (unsigned)X), n = (int)(first / (unsigned)(Z * YQ * X));
  const int sz = (int)(ustep % (unsigned)Z), syq = (int)((ustep / (unsigned)Z) % (unsigned)YQ), sx = (int)((ustep / (unsigned)(Z * YQ)) % (unsigned)X), sn = (int)(ustep / (unsigned)(Z * YQ * X));
  const int my_items = first < uitems ? (int)((uitems - 1u - first) / ustep) + 1 : 0;
  const int64_t trow = (int64_t)Z * k.tpitch;  // elements between y-consecutive voxels of T
  for (int it = 0; it < my_items; ++it) {
    const int y0 = yq * 4;
    const int v0 = ((n * X + x) * Y + y0) * Z + z;  // voxel index (< 2^31: checked on the host)
    const T* t0 = tp + (int64_t)v0 * k.tpitch;
    f8 tv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) tv[i] = ld8(t0 + i * trow);
    if constexpr (BN) {  // y, dA, keep-mask -> dy, rounded to the storage type as the materialised tensor would have been
      const T* d0 = dap + (int64_t)v0 * k.dapitch;
      const int64_t darow = (int64_t)Z * k.dapitch;
      f8 dv[4];
      unsigned kp[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        dv[i] = ld8(d0 + i * darow);
        kp[i] = k.keep ? (unsigned)k.keep[(int64_t)(v0 + i * Z) * cgs + cg] : 0xffu;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f8 o;
        bn_bwd_dy8(tv[i], dv[i], kp[i], alpha, bc, o);
        if constexpr (sizeof(T) == 2) tv[i] = bf16x8_to_f8(f8_to_bf16x8(o));
        else tv[i] = o;
      }
    }
#pragma unroll
    for (int dxi = 0; dxi < K3; ++dxi) {
      const int xs = x + dxi - R;
      const bool xok = (unsigned)xs < (unsigned)X;
      // every load is unconditional, from a clamped (always valid) row, and the zero padding of the convolution is a select on the loaded
      // value: with `ok ? load : 0` hipcc branched around each of the 18 loads and waited for each one in turn (0.70 ms per launch)
      const T* s0 = sp + (v0 + (xok ? (dxi - R) * ZY : 0));
      float sr[ROWS];
#pragma unroll
      for (int j = 0; j < ROWS; ++j) {
        const int ys = y0 - R + j;
        const bool yok = (unsigned)ys < (unsigned)Y;
        const float v = wn_ld<T>(s0 + (yok ? (j - R) * Z : 0));
        sr[j] = (xok && yok) ? v : 0.f;
      }
      if (dxi == R) bsum += (sr[R] + sr[R + 1]) + (sr[R + 2] + sr[R + 3]);  // the four centre voxels of this item
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int dyi = 0; dyi < K3; ++dyi)
#pragma unroll
          for (int c = 0; c < 8; ++c) acc[dxi * K3 + dyi][c] += tv[i].v[c] * sr[i + dyi];
    }
    z += sz; if (z >= Z) { z -= Z; ++yq; }
    yq += syq; if (yq >= YQ) { yq -= YQ; ++x; }
    x += sx; if (x >= X) { x -= X; ++n; }
    n += sn;
  }
  const int wave = tid >> 6;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float v = acc[t][c];
      for (int o = 32; o >= cgs; o >>= 1) v += __shfl_xor(v, o, 64);  // lanes with the same channel group (lane % cgs): cgs divides 64
      if ((tid & 63) < cgs) red[wave][t * C + cg * 8 + c] = v;
    }
  bsum = wave_sum(cg == 0 ? bsum : 0.f);  // every voxel is visited once per channel group: group 0 carries the bias sum
  if ((tid & 63) == 0) red[wave][NT * C] = bsum;
  __syncthreads();
  const int row = NT * C + (k.bias ? 1 : 0);
  float* slab = k.slab + (int64_t)blockIdx.x * row;
  for (int i = tid; i < row; i += 256) slab[i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

// dw[c*stride_c + weight tap] += sum over workgroups of slab[b][t*C + c] (vsseg_slab_sum, as wgrad.hip) — one thread per element walking 1024
// slabs cost 224 us per launch, as much as the reduction over the tensor itself
__global__ __launch_bounds__(VSSEG_SLAB_THREADS) void wgrad_narrow_reduce_kernel(const float* __restrict__ slab, int nblk, int C, int K3, int sign, float* __restrict__ dw, int64_t stride_c, float* __restrict__ dbias) {
  __shared__ float lds[VSSEG_SLAB_THREADS];
  const int NT = K3 * K3, R = K3 / 2;
  const int64_t total = (int64_t)NT * C + (dbias ? 1 : 0);
  const int64_t i = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const float s = vsseg_slab_sum(slab, total, i, nblk, lds);
  if (threadIdx.x >= 64 || i >= total) return;
  if (i == (int64_t)NT * C) { *dbias += s; return; }
  const int t = (int)i / C, c = (int)i - t * C;
  const int dx = t / K3 - R, dy = t % K3 - R;  // s-offset (dx, dy) = sign * (weight tap offset)
  const int widx = (sign * dx + R) * K3 + (sign * dy + R);
  dw[(int64_t)c * stride_c + widx] += s;
}

static int wgrad_narrow_impl(vsseg_tensor t, const void* s, int32_t k3, int32_t sign, float* dw, int64_t stride_c, float* dbias, float* scratch, int64_t scratch_elems, void* stream, const WnK* bn);
extern "C" int vsseg_wgrad_narrow(vsseg_tensor t, const void* s, int32_t k3, int32_t sign, float* dw, int64_t stride_c, float* dbias, float* scratch, int64_t scratch_elems, void* stream) {
  return wgrad_narrow_impl(t, s, k3, sign, dw, stride_c, dbias, scratch, scratch_elems, stream, nullptr);
}
// 1 -> C, 3x3x1, bf16: the same reduction with T = d(conv output) formed on load (see WnK): y = convolution output, dout = gradient of the block output
extern "C" int vsseg_wgrad_narrow_bn(vsseg_tensor y, vsseg_tensor dout, const uint8_t* keep, const float* mean, const float* invstd, const float* gamma, const float* scale, const float* shift, const float* alpha,
                                     const float* mean_dz, const float* mean_dzx, float p_drop, const void* s, float* dw, int64_t stride_c, float* scratch, int64_t scratch_elems, void* stream) {
  VSSEG_CHECK(y.ptr && dout.ptr && !dout.ptr2 && mean && invstd && gamma && scale && shift && alpha && mean_dz && mean_dzx, "vsseg_wgrad_narrow_bn: null pointer");
  VSSEG_CHECK(y.dtype == VSSEG_BF16 && dout.dtype == VSSEG_BF16 && dout.c == y.c && dout.pitch % 8 == 0 && dout.n == y.n && dout.x == y.x && dout.y == y.y && dout.z == y.z, "vsseg_wgrad_narrow_bn: y / dout must be bf16 tensors of one shape");
  VSSEG_CHECK(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || keep), "vsseg_wgrad_narrow_bn: dropout needs the keep-mask bytes of the forward");
  WnK bn;
  bn.da = dout.ptr; bn.dapitch = dout.pitch; bn.keep = p_drop > 0.f ? keep : nullptr;
  bn.mean = mean; bn.invstd = invstd; bn.gamma = gamma; bn.scale = scale; bn.shift = shift; bn.alpha = alpha; bn.mean_dz = mean_dz; bn.mean_dzx = mean_dzx;
  bn.inv_keep = 1.f / (1.f - p_drop);
  return wgrad_narrow_impl(y, s, 3, 1, dw, stride_c, nullptr, scratch, scratch_elems, stream, &bn);
}
static int wgrad_narrow_impl(vsseg_tensor t, const void* s, int32_t k3, int32_t sign, float* dw, int64_t stride_c, float* dbias, float* scratch, int64_t scratch_elems, void* stream, const WnK* bn) {
  VSSEG_CHECK(!dbias || sign == -1, "vsseg_wgrad_narrow: the bias gradient (sum of the one-channel dY) belongs to the C -> 1 case (sign = -1)");
  VSSEG_CHECK(t.ptr && s && dw && scratch && !t.ptr2, "vsseg_wgrad_narrow: bad pointers (two-part tensors are not supported)");
  VSSEG_CHECK(k3 == 1 || k3 == 3, "vsseg_wgrad_narrow: 3x3x1 or 1x1x1 kernels only (k3 = %d)", k3);
  VSSEG_CHECK(sign == 1 || sign == -1, "vsseg_wgrad_narrow: sign must be +1 or -1");
  VSSEG_CHECK(t.c % 8 == 0 && t.c >= 8 && t.c <= 64 && 64 % (t.c / 8) == 0 && t.pitch % 8 == 0, "vsseg_wgrad_narrow: channel count %d must be 8, 16, 32 or 64", t.c);
  VSSEG_CHECK(t.y % 4 == 0, "vsseg_wgrad_narrow: y extent %d must be a multiple of 4", t.y);
  WnK k;
  k.t = t.ptr; k.s = s; k.dw = dw;
  k.tpitch = t.pitch; k.C = t.c; k.cgs = t.c / 8;
  k.n = t.n; k.X = t.x; k.Y = t.y; k.Z = t.z; k.yq = t.y / 4;
  k.sign = sign; k.stride_c = stride_c;
  k.items = (int64_t)t.n * t.x * k.yq * t.z;
  VSSEG_CHECK((int64_t)t.n * t.x * t.y * t.z < (1ll << 31), "vsseg_wgrad_narrow: more than 2^31 voxels");
  // >= 16 items per thread before another workgroup is worth its flush; at most 3 workgroups per CU (the step's 4 launches: 0.68 ms with
  // 1024 workgroups, 0.62 with 512-768, 0.72 with 384, 0.74 with 4096)
  int grid = grid_for(k.items * k.cgs / 16, 256, 256 * 3);
  k.bias = dbias ? 1 : 0;
  const int64_t cap = scratch_elems / (k3 * k3 * t.c + 1);
  VSSEG_CHECK(cap >= 1, "vsseg_wgrad_narrow: scratch too small");
  if (grid > cap) grid = (int)cap;
  k.slab = scratch;
  k.da = nullptr; k.keep = nullptr;
  if (bn) {
    k.da = bn->da; k.dapitch = bn->dapitch; k.keep = bn->keep; k.inv_keep = bn->inv_keep;
    k.mean = bn->mean; k.invstd = bn->invstd; k.gamma = bn->gamma; k.scale = bn->scale; k.shift = bn->shift; k.alpha = bn->alpha; k.mean_dz = bn->mean_dz; k.mean_dzx = bn->mean_dzx;
    hipLaunchKernelGGL((wgrad_narrow_kernel<bf16_t, 3, true>), dim3(grid), dim3(256), 0, as_stream(stream), k);
  } else if (t.dtype == VSSEG_F32) {
    if (k3 == 3) hipLaunchKernelGGL((wgrad_narrow_kernel<float, 3>), dim3(grid), dim3(256), 0, as_stream(stream), k);
    else hipLaunchKernelGGL((wgrad_narrow_kernel<float, 1>), dim3(grid), dim3(256), 0, as_stream(stream), k);
  } else {
    if (k3 == 3) hipLaunchKernelGGL((wgrad_narrow_kernel<bf16_t, 3>), dim3(grid), dim3(256), 0, as_stream(stream), k);
    else hipLaunchKernelGGL((wgrad_narrow_kernel<bf16_t, 1>), dim3(grid), dim3(256), 0, as_stream(stream), k);
  }
  VSSEG_LAUNCH_CHECK("vsseg_wgrad_narrow");
  const int total = k3 * k3 * t.c + (dbias ? 1 : 0);
  hipLaunchKernelGGL(wgrad_narrow_reduce_kernel, dim3((total + 63) / 64), dim3(VSSEG_SLAB_THREADS), 0, as_stream(stream), (const float*)scratch, grid, (int)t.c, (int)k3, (int)sign, dw, stride_c, dbias);
  VSSEG_LAUNCH_CHECK("vsseg_wgrad_narrow(reduce)");
  return VSSEG_OK;
}
