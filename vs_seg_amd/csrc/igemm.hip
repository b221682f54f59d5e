// Implicit-GEMM 3-D convolution on the gfx950 matrix cores.
//
// One kernel serves Conv3d forward, every parity class of ConvTranspose3d forward and both data gradients
// (reference ops: ref:params/networks/blocks/convolutions.py:114-146; autograd of them at ref:params/VSparams.py:461).
//
//   out[q*os+oo][n] = epilogue( sum_{tap t} sum_{c} in[q*is + off_t][c] * W[t][c][n] )
//
// Mapping to MFMA (16x16 output tiles, wave64):
//   * D rows   <- 16 output channels, D cols <- 16 lattice voxels ("swapped" GEMM), so after the MFMA a lane owns
//     4 consecutive channels of one voxel and the epilogue stores 8/16 contiguous bytes per lane.
//   * K        <- (tap, input channel).  K is cut in groups of 8 channels; lane-group g = lane>>4 of a K-step owns
//     group ks*4+g, i.e. one 16-byte (bf16) / 32-byte (f32) LDS read per lane per K-step.  bf16: one
//     v_mfma_f32_16x16x32_bf16 per (voxel tile, channel tile); f32: eight v_mfma_f32_16x16x4_f32 (exact fp32).
//   * the workgroup's input halo tile [hx][hy][hz][ck] and the chunk's packed weights are staged in LDS once per
//     channel chunk; the 9/27 tap re-reads hit LDS, not L2/HBM.
#include "common.h"

struct IgemmK {
  vsseg_igemm_desc d;
  int halo[3];
  int off_min[3];
  int ntile[3];
  int cgs;  // 8-channel groups per chunk
  int lds_ktab, lds_vbase, lds_b, lds_halo;  // byte offsets
};

template <typename T> struct Frag;
template <> struct Frag<bf16_t> {
  bf16x8 v;
  static __device__ __forceinline__ Frag ld(const char* p) { Frag f; f.v = *reinterpret_cast<const bf16x8*>(p); return f; }
};
template <> struct Frag<float> {
  float4 lo, hi;
  static __device__ __forceinline__ Frag ld(const char* p) {
    Frag f;
    f.lo = *reinterpret_cast<const float4*>(p);
    f.hi = *reinterpret_cast<const float4*>(p + 16);
    return f;
  }
};
__device__ __forceinline__ void mma(f32x4& acc, const Frag<bf16_t>& w, const Frag<bf16_t>& a) { acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.v, a.v, acc, 0, 0, 0); }
__device__ __forceinline__ void mma(f32x4& acc, const Frag<float>& w, const Frag<float>& a) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.lo.x, a.lo.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.lo.y, a.lo.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.lo.z, a.lo.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.lo.w, a.lo.w, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.hi.x, a.hi.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.hi.y, a.hi.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.hi.z, a.hi.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.hi.w, a.hi.w, acc, 0, 0, 0);
}

template <typename T, int NT, int MTW>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmK k) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ES = sizeof(T);
  constexpr int GB = 8 * ES;  // bytes of one 8-channel group
  const vsseg_igemm_desc& d = k.d;
  int* ktab = reinterpret_cast<int*>(smem + k.lds_ktab);
  int* vbase = reinterpret_cast<int*>(smem + k.lds_vbase);
  char* Bl = smem + k.lds_b;
  char* Hl = smem + k.lds_halo;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
  // ---- decode workgroup -> (batch n, tile origin q0) ----
  int b = blockIdx.x;
  const int tz = b % k.ntile[2]; b /= k.ntile[2];
  const int ty = b % k.ntile[1]; b /= k.ntile[1];
  const int tx = b % k.ntile[0];
  const int n = b / k.ntile[0];
  const int q0[3] = {tx * d.tile[0], ty * d.tile[1], tz * d.tile[2]};
  const int split = blockIdx.y;
  const int HY = k.halo[1], HZ = k.halo[2], CK = d.ck;
  const int hvox = k.halo[0] * HY * HZ;

  // ---- per-workgroup lookup tables ----
  for (int i = tid; i < d.ksteps * 4; i += 256) {
    int off = 0;
    if (i < d.ntaps * k.cgs) {
      int t = i / k.cgs, cg = i - t * k.cgs;
      int hx = d.tap_off[t][0] - k.off_min[0], hy = d.tap_off[t][1] - k.off_min[1], hz = d.tap_off[t][2] - k.off_min[2];
      off = ((hx * HY + hy) * HZ + hz) * CK * ES + cg * GB;
    }
    ktab[i] = off;
  }
  for (int v = tid; v < 64 * MTW; v += 256) {
    int vz = v % d.tile[2], r = v / d.tile[2];
    int vy = r % d.tile[1], vx = r / d.tile[1];
    vbase[v] = (((vx * d.is[0]) * HY + vy * d.is[1]) * HZ + vz * d.is[2]) * CK * ES;
  }

  f32x4 acc[MTW][NT];
#pragma unroll
  for (int m = 0; m < MTW; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const T* inp = reinterpret_cast<const T*>(d.in.ptr);
  const int X = d.in.x, Y = d.in.y, Z = d.in.z;
  const int gx0 = q0[0] * d.is[0] + k.off_min[0], gy0 = q0[1] * d.is[1] + k.off_min[1], gz0 = q0[2] * d.is[2] + k.off_min[2];
  const int64_t chunk_w_bytes = (int64_t)d.ksteps * NT * 64 * GB;

  for (int ch = 0; ch < d.nchunks; ++ch) {
    __syncthreads();  // previous chunk's fragment reads are done (also publishes ktab/vbase on the first pass)
    {  // packed weights of this chunk: linear copy, 16 B per lane
      const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(d.wpack) + ((int64_t)split * d.nchunks + ch) * chunk_w_bytes);
      uint4* dst = reinterpret_cast<uint4*>(Bl);
      const int n16 = (int)(chunk_w_bytes >> 4);
      for (int i = tid; i < n16; i += 256) dst[i] = src[i];
    }
    {  // input halo tile, zero-filled outside the tensor
      const int c0 = ch * CK;
      const int items = hvox * k.cgs;
      for (int i = tid; i < items; i += 256) {
        int hv = i / k.cgs, cg = i - hv * k.cgs;
        int hz = hv % HZ, r = hv / HZ;
        int hy = r % HY, hx = r / HY;
        int gx = gx0 + hx, gy = gy0 + hy, gz = gz0 + hz;
        char* dst = Hl + (int64_t)hv * CK * ES + cg * GB;
        const bool ok = (unsigned)gx < (unsigned)X && (unsigned)gy < (unsigned)Y && (unsigned)gz < (unsigned)Z && (c0 + cg * 8 + 8 <= d.in.c);
        if (ES == 2) {
          uint4 v = make_uint4(0, 0, 0, 0);
          if (ok) v = *reinterpret_cast<const uint4*>(inp + ((((int64_t)n * X + gx) * Y + gy) * Z + gz) * d.in.pitch + c0 + cg * 8);
          *reinterpret_cast<uint4*>(dst) = v;
        } else {
          uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0;
          if (ok) {
            const uint4* s = reinterpret_cast<const uint4*>(inp + ((((int64_t)n * X + gx) * Y + gy) * Z + gz) * d.in.pitch + c0 + cg * 8);
            v0 = s[0];
            v1 = s[1];
          }
          reinterpret_cast<uint4*>(dst)[0] = v0;
          reinterpret_cast<uint4*>(dst)[1] = v1;
        }
      }
    }
    __syncthreads();

    int vb[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) vb[m] = vbase[(wave * MTW + m) * 16 + l15];
    for (int ks = 0; ks < d.ksteps; ++ks) {
      const int koff = ktab[ks * 4 + g];
      Frag<T> w[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) w[t] = Frag<T>::ld(Bl + ((int64_t)(ks * NT + t) * 64 + lane) * GB);
#pragma unroll
      for (int m = 0; m < MTW; ++m) {
        Frag<T> a = Frag<T>::ld(Hl + vb[m] + koff);
#pragma unroll
        for (int t = 0; t < NT; ++t) mma(acc[m][t], w[t], a);
      }
    }
  }

  // ---- epilogue ----
  const int cout = d.out.c;
  const float alpha = (d.act == VSSEG_ACT_PRELU && d.alpha) ? *d.alpha : 0.f;
  float ssum[NT][4], ssq[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) ssum[t][r] = ssq[t][r] = 0.f;

#pragma unroll
  for (int m = 0; m < MTW; ++m) {
    const int v = (wave * MTW + m) * 16 + l15;
    int vz = v % d.tile[2], rr = v / d.tile[2];
    int vy = rr % d.tile[1], vx = rr / d.tile[1];
    const int qx = q0[0] + vx, qy = q0[1] + vy, qz = q0[2] + vz;
    const int ox = qx * d.os[0] + d.oo[0], oy = qy * d.os[1] + d.oo[1], oz = qz * d.os[2] + d.oo[2];
    const bool vok = qx < d.q[0] && qy < d.q[1] && qz < d.q[2] && ox < d.out.x && oy < d.out.y && oz < d.out.z;
    const int64_t ovox = (((int64_t)n * d.out.x + ox) * d.out.y + oy) * d.out.z + oz;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int c = (split * NT + t) * 16 + g * 4;
      if (!vok || c >= cout) continue;
      float val[4] = {acc[m][t][0], acc[m][t][1], acc[m][t][2], acc[m][t][3]};
      const int nc = min(4, cout - c);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (r >= nc) break;
        float x = val[r];
        if (d.bias) x += d.bias[c + r];
        if (d.stats) { ssum[t][r] += x; ssq[t][r] += x * x; }
        if (d.scale) x = x * d.scale[c + r] + d.shift[c + r];
        if (d.act == VSSEG_ACT_PRELU) x = x > 0.f ? x : alpha * x;
        else if (d.act == VSSEG_ACT_RELU) x = fmaxf(x, 0.f);
        else if (d.act == VSSEG_ACT_SIGMOID) x = 1.f / (1.f + __expf(-x));
        if (d.res_mode != VSSEG_RES_NONE) {
          const int64_t ro = ovox * d.res.pitch + c + r;
          float rv = d.res.dtype == VSSEG_F32 ? reinterpret_cast<const float*>(d.res.ptr)[ro] : bf2f(reinterpret_cast<const bf16_t*>(d.res.ptr)[ro]);
          x = d.res_mode == VSSEG_RES_ADD ? x + rv : (rv > 0.f ? x : 0.f);
        }
        val[r] = x;
      }
      const int64_t oo = ovox * d.out.pitch + c;
      if (d.out.dtype == VSSEG_F32) {
        float* op = reinterpret_cast<float*>(d.out.ptr) + oo;
        if (nc == 4 && (d.out.pitch & 3) == 0 && !d.accumulate) st4(op, make_float4(val[0], val[1], val[2], val[3]));
        else
          for (int r = 0; r < nc; ++r) op[r] = d.accumulate ? op[r] + val[r] : val[r];
      } else {
        bf16_t* op = reinterpret_cast<bf16_t*>(d.out.ptr) + oo;
        if (nc == 4 && (d.out.pitch & 3) == 0) {
          if (d.accumulate) {
            float4 o = ld4(op);
            val[0] += o.x; val[1] += o.y; val[2] += o.z; val[3] += o.w;
          }
          st4(op, make_float4(val[0], val[1], val[2], val[3]));
        } else
          for (int r = 0; r < nc; ++r) op[r] = f2bf(d.accumulate ? bf2f(op[r]) + val[r] : val[r]);
      }
    }
  }

  if (d.stats) {  // per-channel sum / sum-of-squares: 16-lane shuffle tree -> LDS across waves -> sharded fp64 atomics
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);  // [2][NT*16]
    for (int i = tid; i < 2 * NT * 16; i += 256) red[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = ssum[t][r], q = ssq[t][r];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
        if (l15 == 0) {
          atomicAdd(&red[t * 16 + g * 4 + r], s);
          atomicAdd(&red[NT * 16 + t * 16 + g * 4 + r], q);
        }
      }
    __syncthreads();
    double* st = d.stats + (int64_t)(blockIdx.x % VSSEG_STAT_SHARDS) * 2 * d.stats_stride;
    for (int i = tid; i < 2 * NT * 16; i += 256) {
      int which = i / (NT * 16), cc = i - which * NT * 16;
      int c = split * NT * 16 + cc;
      if (c < cout) atomicAdd(&st[which * d.stats_stride + c], (double)red[i]);
    }
  }
}

template <typename T, int NT, int MTW> static int launch(const IgemmK& k, dim3 grid, int lds, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<T, NT, MTW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((igemm_kernel<T, NT, MTW>), grid, dim3(256), lds, s, k);
  VSSEG_LAUNCH_CHECK("vsseg_igemm");
  return VSSEG_OK;
}
template <typename T, int NT> static int launch_mtw(const IgemmK& k, dim3 grid, int lds, hipStream_t s) {
  switch (k.d.mtw) {
    case 1: return launch<T, NT, 1>(k, grid, lds, s);
    case 2: return launch<T, NT, 2>(k, grid, lds, s);
    case 4: return launch<T, NT, 4>(k, grid, lds, s);
  }
  vsseg_set_error("vsseg_igemm: mtw must be 1, 2 or 4 (got %d)", k.d.mtw);
  return VSSEG_EINVAL;
}
template <typename T> static int launch_nt(const IgemmK& k, dim3 grid, int lds, hipStream_t s) {
  switch (k.d.nt) {
    case 1: return launch_mtw<T, 1>(k, grid, lds, s);
    case 2: return launch_mtw<T, 2>(k, grid, lds, s);
    case 3: return launch_mtw<T, 3>(k, grid, lds, s);
    case 4: return launch_mtw<T, 4>(k, grid, lds, s);
    case 5: return launch_mtw<T, 5>(k, grid, lds, s);
    case 6: return launch_mtw<T, 6>(k, grid, lds, s);
  }
  vsseg_set_error("vsseg_igemm: nt must be 1..6 (got %d)", k.d.nt);
  return VSSEG_EINVAL;
}

static int igemm_prepare(const vsseg_igemm_desc* d, IgemmK& k) {
  VSSEG_CHECK(d && d->in.ptr && d->out.ptr && d->wpack, "vsseg_igemm: null pointer");
  VSSEG_CHECK(d->in.dtype == VSSEG_F32 || d->in.dtype == VSSEG_BF16, "vsseg_igemm: bad input dtype");
  VSSEG_CHECK(d->ntaps >= 1 && d->ntaps <= VSSEG_MAX_TAPS, "vsseg_igemm: ntaps %d out of range", d->ntaps);
  VSSEG_CHECK(d->ck >= 8 && d->ck % 8 == 0 && d->nchunks >= 1, "vsseg_igemm: bad channel chunking ck=%d nchunks=%d", d->ck, d->nchunks);
  VSSEG_CHECK(d->in.c % 8 == 0 && d->in.pitch % 8 == 0, "vsseg_igemm: input channels/pitch must be multiples of 8 (c=%d pitch=%d)", d->in.c, d->in.pitch);
  VSSEG_CHECK(d->tile[0] * d->tile[1] * d->tile[2] == 64 * d->mtw, "vsseg_igemm: tile %dx%dx%d != 64*mtw", d->tile[0], d->tile[1], d->tile[2]);
  VSSEG_CHECK(d->nsplit >= 1 && d->nsplit * d->nt * 16 >= d->out.c, "vsseg_igemm: nsplit*nt*16 < cout");
  k.d = *d;
  const int es = d->in.dtype == VSSEG_F32 ? 4 : 2;
  k.cgs = d->ck / 8;
  VSSEG_CHECK(d->ksteps * 4 >= d->ntaps * k.cgs, "vsseg_igemm: ksteps too small");
  for (int a = 0; a < 3; ++a) {
    int lo = d->tap_off[0][a], hi = lo;
    for (int t = 1; t < d->ntaps; ++t) { lo = min(lo, d->tap_off[t][a]); hi = max(hi, d->tap_off[t][a]); }
    k.off_min[a] = lo;
    k.halo[a] = (d->tile[a] - 1) * d->is[a] + (hi - lo + 1);
    k.ntile[a] = (d->q[a] + d->tile[a] - 1) / d->tile[a];
  }
  int off = 0;
  k.lds_ktab = off; off += ((d->ksteps * 4 * 4 + 15) / 16) * 16;
  k.lds_vbase = off; off += 64 * d->mtw * 4;
  k.lds_b = off; off += d->ksteps * d->nt * 64 * 8 * es;
  k.lds_halo = off; off += k.halo[0] * k.halo[1] * k.halo[2] * d->ck * es;
  VSSEG_CHECK(off <= 160 * 1024, "vsseg_igemm: needs %d bytes of LDS (> 160 KiB); reduce ck or the tile", off);
  return off;
}

extern "C" int vsseg_igemm_lds_bytes(const vsseg_igemm_desc* d) {
  IgemmK k;
  return igemm_prepare(d, k);
}

extern "C" int vsseg_igemm(const vsseg_igemm_desc* d, void* stream) {
  IgemmK k;
  int lds = igemm_prepare(d, k);
  if (lds < 0) return lds;
  int64_t blocks = (int64_t)d->in.n * k.ntile[0] * k.ntile[1] * k.ntile[2];
  VSSEG_CHECK(blocks > 0 && blocks < (1ll << 31), "vsseg_igemm: bad grid");
  dim3 grid((unsigned)blocks, (unsigned)d->nsplit);
  if (d->in.dtype == VSSEG_F32) return launch_nt<float>(k, grid, lds, as_stream(stream));
  return launch_nt<bf16_t>(k, grid, lds, as_stream(stream));
}
