// Weight gradients of Conv3d / ConvTranspose3d on the gfx950 matrix cores (autograd of
// ref:params/networks/blocks/convolutions.py:114-146 as run by `loss.backward()`, ref:params/VSparams.py:461).
//
//   dW[tap][cP][cH] += sum_q P[q][cP] * H[q*hs + off_tap][cH]        Conv3d: P = dY, H = X;  ConvTranspose3d: P = X, H = dY
//
// GEMM view per tap: rows = P channels, cols = H channels, reduction = lattice voxels.  In the channels-last layout the
// reduction axis is the *strided* one, so the MFMA operands (8 consecutive voxels of one channel per lane) are built with
// the LDS transpose read ds_read_b64_tr_b16 from [voxel][channel] tiles (bf16); the f32 path uses v_mfma_f32_16x16x4_f32
// whose operands are single elements and need no transpose.
//
// Work split: blockIdx.y = 16-channel chunk of H (only those channels of the halo tile are staged), blockIdx.x = persistent
// workgroup striding over lattice tiles; accumulators live in registers across all tiles of a workgroup and are flushed
// once as a partial-sum slab (plain coalesced stores; fp32 atomics measured ~18 G/s and dominated the kernel); a tiny
// second kernel sums the slabs of all workgroups into the weight gradient.  The 4 waves of a workgroup split the taps (wave w owns taps t = w mod WT) and, for 1x1x1
// kernels, the K-steps.
#include "common.h"

struct WgradK {
  vsseg_wgrad_desc d;
  int halo[3];
  int off_min[3];
  int ntile[3];
  int wt, wv;     // wave split: taps x K-steps (wt*wv == 4)
  int tvox;       // voxels per tile
  int p_row;      // bytes per P row in LDS
  int lds_hbase, lds_p, lds_h;
  int64_t total_tiles;
  float* slab;     // [gridDim.x][hchunks][ntaps][ntp*16][16] partial sums
  int slab_chunk;  // ntaps * ntp*16 * 16
};

template <typename T, int MAXT, int NTP>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradK k) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ES = sizeof(T);
  constexpr int HROW = 16 * ES;  // bytes per halo row (16 channels)
  const vsseg_wgrad_desc& d = k.d;
  int* hbase = reinterpret_cast<int*>(smem + k.lds_hbase);
  char* Pl = smem + k.lds_p;
  char* Hl = smem + k.lds_h;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, l15 = lane & 15;
  const int wt = wave % k.wt, wv = wave / k.wt;
  const int HY = k.halo[1], HZ = k.halo[2];
  const int hvox = k.halo[0] * HY * HZ;
  const int chunk = blockIdx.y;

  for (int v = tid; v < k.tvox; v += 256) {
    int vz = v % d.tile[2], r = v / d.tile[2];
    int vy = r % d.tile[1], vx = r / d.tile[1];
    hbase[v] = ((vx * d.hs[0]) * HY + vy * d.hs[1]) * HZ + vz * d.hs[2];
  }
  int toff[MAXT];  // LDS byte offset of each owned tap inside the halo tile (-1: not owned)
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    int t = wt + i * k.wt;
    toff[i] = t < d.ntaps ? (((d.tap_off[t][0] - k.off_min[0]) * HY + (d.tap_off[t][1] - k.off_min[1])) * HZ + (d.tap_off[t][2] - k.off_min[2])) * HROW : -1;
  }
  f32x4 acc[MAXT][NTP];
#pragma unroll
  for (int i = 0; i < MAXT; ++i)
#pragma unroll
    for (int p = 0; p < NTP; ++p) acc[i][p] = f32x4{0.f, 0.f, 0.f, 0.f};

  const T* Pg = reinterpret_cast<const T*>(d.p.ptr);
  const T* Hg = reinterpret_cast<const T*>(d.h.ptr);
  const int PC = NTP * 16;
  const int ksteps = k.tvox / 32;

  for (int64_t tile = blockIdx.x; tile < k.total_tiles; tile += gridDim.x) {
    int64_t b = tile;
    const int tz = (int)(b % k.ntile[2]); b /= k.ntile[2];
    const int ty = (int)(b % k.ntile[1]); b /= k.ntile[1];
    const int tx = (int)(b % k.ntile[0]);
    const int n = (int)(b / k.ntile[0]);
    const int q0x = tx * d.tile[0], q0y = ty * d.tile[1], q0z = tz * d.tile[2];
    __syncthreads();
    {  // P tile [tvox][NTP*16], zero outside the lattice / beyond the valid channels.  Loads are issued in batches of U
       // before any LDS store so that U global loads per thread are in flight (the staging is latency-, not bandwidth-bound).
      constexpr int U = 4;
      constexpr int V16 = ES == 2 ? 1 : 2;  // 16-byte pieces per 8-channel group
      const int cgs = PC / 8, items = k.tvox * cgs;
      for (int base = 0; base < items; base += 256 * U) {
        uint4 val[U][V16];
        int dsto[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = base + u * 256 + tid;
          const int ii = min(i, items - 1);
          int v = ii / cgs, cg = ii - v * cgs;
          int vz = v % d.tile[2], r = v / d.tile[2];
          int vy = r % d.tile[1], vx = r / d.tile[1];
          int qx = q0x + vx, qy = q0y + vy, qz = q0z + vz;
          const bool ok = qx < d.q[0] && qy < d.q[1] && qz < d.q[2] && cg * 8 + 8 <= d.p.c;
          dsto[u] = i < items ? v * k.p_row + cg * 8 * ES : -1;
          const T* src = ok ? Pg + ((((int64_t)n * d.p.x + qx) * d.p.y + qy) * d.p.z + qz) * d.p.pitch + cg * 8 : Pg;
#pragma unroll
          for (int w = 0; w < V16; ++w) {
            uint4 x = reinterpret_cast<const uint4*>(src)[w];
            val[u][w] = ok ? x : make_uint4(0, 0, 0, 0);
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (dsto[u] >= 0) {
#pragma unroll
            for (int w = 0; w < V16; ++w) reinterpret_cast<uint4*>(Pl + dsto[u])[w] = val[u][w];
          }
      }
    }
    {  // H halo tile [hvox][16] for this workgroup's channel chunk
      constexpr int U = 4;
      constexpr int V16 = ES == 2 ? 1 : 2;
      const int gx0 = q0x * d.hs[0] + k.off_min[0], gy0 = q0y * d.hs[1] + k.off_min[1], gz0 = q0z * d.hs[2] + k.off_min[2];
      const int items = hvox * 2;
      for (int base = 0; base < items; base += 256 * U) {
        uint4 val[U][V16];
        int dsto[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = base + u * 256 + tid;
          const int ii = min(i, items - 1);
          int hv = ii >> 1, cg = ii & 1;
          int hz = hv % HZ, r = hv / HZ;
          int hy = r % HY, hx = r / HY;
          int gx = gx0 + hx, gy = gy0 + hy, gz = gz0 + hz;
          const int c = chunk * 16 + cg * 8;
          const bool ok = (unsigned)gx < (unsigned)d.h.x && (unsigned)gy < (unsigned)d.h.y && (unsigned)gz < (unsigned)d.h.z && c + 8 <= d.h.c;
          dsto[u] = i < items ? hv * HROW + cg * 8 * ES : -1;
          const T* src = ok ? Hg + ((((int64_t)n * d.h.x + gx) * d.h.y + gy) * d.h.z + gz) * d.h.pitch + c : Hg;
#pragma unroll
          for (int w = 0; w < V16; ++w) {
            uint4 x = reinterpret_cast<const uint4*>(src)[w];
            val[u][w] = ok ? x : make_uint4(0, 0, 0, 0);
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (dsto[u] >= 0) {
#pragma unroll
            for (int w = 0; w < V16; ++w) reinterpret_cast<uint4*>(Hl + dsto[u])[w] = val[u][w];
          }
      }
    }
    __syncthreads();

    for (int ks = wv; ks < ksteps; ks += k.wv) {
      if constexpr (ES == 2) {
        // lane (g, i=l15): rows r=i>>2 of two 4-voxel blocks, 4-channel column chunk q=i&3
        const int r = l15 >> 2, qc = (l15 & 3) * 8;  // byte offset of the 4-channel chunk
        const int v0 = ks * 32 + g * 8 + r, v1 = v0 + 4;
        const int h0 = hbase[v0] * HROW + qc, h1 = hbase[v1] * HROW + qc;
        bf16x8 pa[NTP];
#pragma unroll
        for (int p = 0; p < NTP; ++p) {
          typedef __attribute__((address_space(3))) bf16x4 lds_b4;
          bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b4*)(Pl + v0 * k.p_row + p * 32 + qc));
          bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b4*)(Pl + v1 * k.p_row + p * 32 + qc));
          pa[p] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int i = 0; i < MAXT; ++i) {
          if (toff[i] < 0) continue;
          typedef __attribute__((address_space(3))) bf16x4 lds_b4;
          bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b4*)(Hl + h0 + toff[i]));
          bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b4*)(Hl + h1 + toff[i]));
          bf16x8 hb = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
          for (int p = 0; p < NTP; ++p) acc[i][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[p], hb, acc[i][p], 0, 0, 0);
        }
      } else {
#pragma unroll 2
        for (int s = 0; s < 8; ++s) {
          const int v = ks * 32 + s * 4 + g;
          const int hb0 = hbase[v] * HROW + l15 * 4;
          float pa[NTP];
#pragma unroll
          for (int p = 0; p < NTP; ++p) pa[p] = *reinterpret_cast<const float*>(Pl + v * k.p_row + (p * 16 + l15) * 4);
#pragma unroll
          for (int i = 0; i < MAXT; ++i) {
            if (toff[i] < 0) continue;
            const float hb = *reinterpret_cast<const float*>(Hl + hb0 + toff[i]);
#pragma unroll
            for (int p = 0; p < NTP; ++p) acc[i][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[p], hb, acc[i][p], 0, 0, 0);
          }
        }
      }
    }
  }

  // flush: lane holds rows g*4+r (P channel) x col l15 (H channel) -> this workgroup's slab [tap][cP][16]
  float* slab = k.slab + ((int64_t)blockIdx.x * gridDim.y + chunk) * k.slab_chunk;
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    const int t = wt + i * k.wt;
    if (t >= d.ntaps) continue;
#pragma unroll
    for (int p = 0; p < NTP; ++p)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cp = p * 16 + g * 4 + r;
        float* dst = slab + ((int64_t)t * (NTP * 16) + cp) * 16 + l15;
        if (k.wv == 1) *dst = acc[i][p][r];
        else atomicAdd(dst, acc[i][p][r]);  // K-steps split over waves (1x1x1 kernels): slab pre-zeroed by the host wrapper
      }
  }
}

// dw[cp][ch][tap] += sum over workgroups of their slabs
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, int nblk, int hchunks, int ntaps, int cpad, int slab_chunk, vsseg_wgrad_desc d) {
  const int total = hchunks * slab_chunk;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int chunk = i / slab_chunk, r = i - chunk * slab_chunk;
    const int l15 = r & 15, cp = (r >> 4) % cpad, t = (r >> 4) / cpad;
    const int ch = chunk * 16 + l15;
    if (cp >= d.cp_valid || ch >= d.ch_valid) continue;
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += slab[((int64_t)b * hchunks + chunk) * slab_chunk + r];
    d.dw[cp * d.stride_p + ch * d.stride_h + d.tap_widx[t] * d.stride_tap] += s;
  }
}

template <typename T, int MAXT, int NTP> static int wg_launch(const WgradK& k, dim3 grid, int lds, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<T, MAXT, NTP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((wgrad_kernel<T, MAXT, NTP>), grid, dim3(256), lds, s, k);
  VSSEG_LAUNCH_CHECK("vsseg_wgrad");
  return VSSEG_OK;
}
template <typename T, int MAXT> static int wg_ntp(const WgradK& k, dim3 grid, int lds, hipStream_t s) {
  switch (k.d.ntp) {
    case 1: return wg_launch<T, MAXT, 1>(k, grid, lds, s);
    case 2: return wg_launch<T, MAXT, 2>(k, grid, lds, s);
    case 3: return wg_launch<T, MAXT, 3>(k, grid, lds, s);
    case 4: return wg_launch<T, MAXT, 4>(k, grid, lds, s);
    case 5: return wg_launch<T, MAXT, 5>(k, grid, lds, s);
    case 6: return wg_launch<T, MAXT, 6>(k, grid, lds, s);
  }
  vsseg_set_error("vsseg_wgrad: ntp must be 1..6 (got %d)", k.d.ntp);
  return VSSEG_EINVAL;
}
template <typename T> static int wg_maxt(const WgradK& k, int maxt, dim3 grid, int lds, hipStream_t s) {
  if (maxt <= 1) return wg_ntp<T, 1>(k, grid, lds, s);
  if (maxt <= 3) return wg_ntp<T, 3>(k, grid, lds, s);
  if (maxt <= 7) return wg_ntp<T, 7>(k, grid, lds, s);
  vsseg_set_error("vsseg_wgrad: too many taps per wave (%d)", maxt);
  return VSSEG_EINVAL;
}

extern "C" int vsseg_wgrad(const vsseg_wgrad_desc* d, void* stream) {
  VSSEG_CHECK(d && d->p.ptr && d->h.ptr && d->dw, "vsseg_wgrad: null pointer");
  VSSEG_CHECK(d->p.dtype == d->h.dtype, "vsseg_wgrad: dtype mismatch");
  VSSEG_CHECK(d->ntaps >= 1 && d->ntaps <= VSSEG_MAX_TAPS, "vsseg_wgrad: ntaps out of range");
  VSSEG_CHECK(d->p.c % 8 == 0 && d->p.pitch % 8 == 0 && d->h.c % 8 == 0 && d->h.pitch % 8 == 0, "vsseg_wgrad: channels/pitch must be multiples of 8");
  VSSEG_CHECK(d->ntp >= 1 && d->ntp * 16 >= d->cp_valid && d->cp_valid <= d->p.c, "vsseg_wgrad: ntp too small for %d P channels", d->cp_valid);
  WgradK k;
  k.d = *d;
  k.tvox = d->tile[0] * d->tile[1] * d->tile[2];
  VSSEG_CHECK(k.tvox % 32 == 0 && k.tvox >= 32, "vsseg_wgrad: tile voxel count must be a multiple of 32");
  const int es = d->p.dtype == VSSEG_F32 ? 4 : 2;
  k.total_tiles = d->p.n;
  for (int a = 0; a < 3; ++a) {
    int lo = d->tap_off[0][a], hi = lo;
    for (int t = 1; t < d->ntaps; ++t) { lo = min(lo, d->tap_off[t][a]); hi = max(hi, d->tap_off[t][a]); }
    k.off_min[a] = lo;
    k.halo[a] = (d->tile[a] - 1) * d->hs[a] + (hi - lo + 1);
    k.ntile[a] = (d->q[a] + d->tile[a] - 1) / d->tile[a];
    k.total_tiles *= k.ntile[a];
  }
  k.wt = d->ntaps >= 4 ? 4 : (d->ntaps >= 2 ? 2 : 1);
  k.wv = 4 / k.wt;
  const int maxt = (d->ntaps + k.wt - 1) / k.wt;
  k.p_row = d->ntp * 16 * es;
  int off = 0;
  k.lds_hbase = off; off += ((k.tvox * 4 + 15) / 16) * 16;
  k.lds_p = off; off += k.tvox * k.p_row;
  k.lds_h = off; off += k.halo[0] * k.halo[1] * k.halo[2] * 16 * es;
  VSSEG_CHECK(off <= 160 * 1024, "vsseg_wgrad: needs %d bytes of LDS (> 160 KiB); reduce the tile", off);
  const int hchunks = (d->ch_valid + 15) / 16;
  k.slab_chunk = d->ntaps * d->ntp * 16 * 16;
  VSSEG_CHECK(d->scratch && d->scratch_elems >= (int64_t)hchunks * k.slab_chunk, "vsseg_wgrad: scratch too small (%lld < %lld floats)", (long long)d->scratch_elems, (long long)hchunks * k.slab_chunk);
  int64_t gx = d->persistent_blocks > 0 ? d->persistent_blocks : 256;
  if (gx > k.total_tiles) gx = k.total_tiles;
  const int64_t cap = d->scratch_elems / ((int64_t)hchunks * k.slab_chunk);
  if (gx > cap) gx = cap;
  k.slab = d->scratch;
  if (k.wv != 1) hipMemsetAsync(d->scratch, 0, sizeof(float) * gx * hchunks * k.slab_chunk, as_stream(stream));
  dim3 grid((unsigned)gx, (unsigned)hchunks);
  int rc = d->p.dtype == VSSEG_F32 ? wg_maxt<float>(k, maxt, grid, off, as_stream(stream)) : wg_maxt<bf16_t>(k, maxt, grid, off, as_stream(stream));
  if (rc) return rc;
  const int total = hchunks * k.slab_chunk;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), (const float*)d->scratch, (int)gx, hchunks, d->ntaps, d->ntp * 16, k.slab_chunk, *d);
  VSSEG_LAUNCH_CHECK("vsseg_wgrad(reduce)");
  return VSSEG_OK;
}
