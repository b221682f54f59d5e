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
// Work split: blockIdx.y = group of HG 16-channel chunks of H (only those channels of the halo tile are staged; one workgroup
// multiplies the P tile it fetched with all HG chunks, so P is read ceil(chunks / HG) times instead of once per chunk — the
// per-chunk split of round 1 moved 1.7x the algorithmic bytes on the 32/64-channel layers), blockIdx.x = persistent
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
  int lds_hbase, lds_p, lds_h, lds_tab;
  int npp, nph;   // rows of the per-thread coordinate tables (P pieces, H pieces) kept in LDS for the boundary-tile path
  int nbuf;       // 2: tile s+1 streams in while tile s is multiplied; 1: no prefetch, half the LDS, more resident workgroups
  int walk;       // 1: XCD-contiguous tile walk (see the kernel)
  int hchunks;    // 16-channel chunks of H in total (slab layout); a workgroup owns HG consecutive ones
  int64_t total_tiles;
  const void* zeros;  // >= 16 zero bytes in global memory (source of out-of-bounds pieces)
  float* slab;     // [gridDim.x * wv][hchunks][ntaps][ntp*16][16] partial sums (one slab per workgroup and K-step share: plain stores, no atomics)
  float* bias_slab;  // [gridDim.x * wv][ntp*16] partial bias gradients (workgroups of H-chunk group 0), or nullptr
  int slab_chunk;  // ntaps * ntp*16 * 16
};

typedef __attribute__((address_space(1))) const void wg_gvoid_t;
typedef __attribute__((address_space(3))) void wg_lvoid_t;
__device__ __forceinline__ void wg_dma16(const void* gsrc, char* lds_wave_base) {  // LDS address = wave-uniform base + lane*16
  vsseg_dma16(gsrc, lds_wave_base);  // inline assembly: see common.h (the builtin made hipcc wait for tile s+1 before multiplying tile s)
}
constexpr int WPP = 12;  // 16-byte pieces of the P tile per thread  (P tile <= 48 KiB)
constexpr int WPH = 16;  // ... of the H halo tile per thread         (halo    <= 64 KiB)

template <typename T, int MAXT, int NTP, int HG>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradK k) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ES = sizeof(T);
  constexpr int EPP = 16 / ES;
  constexpr int HROW = HG * 16 * ES;  // bytes per halo row (HG chunks of 16 channels)
  constexpr int HPP = HROW / 16;      // 16-byte pieces per halo voxel
  const vsseg_wgrad_desc& d = k.d;
  int* hbase = reinterpret_cast<int*>(smem + k.lds_hbase);
  char* Pl = smem + k.lds_p;
  char* Hl = smem + k.lds_h;
  unsigned* pinfo_l = reinterpret_cast<unsigned*>(smem + k.lds_tab);  // [npp][256] packed tile coordinates of this thread's P pieces
  unsigned* hinfo_l = pinfo_l + k.npp * 256;                          // [nph][256] ... of its halo pieces
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, l15 = lane & 15;
  const int wt = wave % k.wt, wv = wave / k.wt;
  const int HX = k.halo[0], HY = k.halo[1], HZ = k.halo[2];
  const int hvox = HX * HY * HZ;
  const int chunk0 = blockIdx.y * HG;  // first 16-channel chunk of H this workgroup owns
  const int p_bytes = k.tvox * k.p_row, h_bytes = hvox * HROW;

  // (run-time divisors, values < 2^16: vsseg_fdiv.  With plain `/` and `%` the coordinate tables below took 6.3 us per workgroup, more than its first tile)
  const int T1 = d.tile[1], T2 = d.tile[2];
  const float iT1 = 1.0f / (float)T1, iT2 = 1.0f / (float)T2, iHY = 1.0f / (float)HY, iHZ = 1.0f / (float)HZ;
  for (int v = tid; v < k.tvox; v += 256) {
    const int r = vsseg_fdiv(v, T2, iT2), vz = v - r * T2;
    const int vx = vsseg_fdiv(r, T1, iT1), vy = r - vx * T1;
    hbase[v] = ((vx * d.hs[0]) * HY + vy * d.hs[1]) * HZ + vz * d.hs[2];
  }
  int toff[MAXT];  // LDS byte offset of each owned tap inside the halo tile (-1: not owned)
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    int t = wt + i * k.wt;
    // a wave with fewer taps than MAXT multiplies tap slot 0 again in its spare slots (never flushed): a `continue` on the spare slots was a branch inside
    // the unrolled tap loop that kept hipcc from hoisting the next taps' fragment reads over the MFMAs of the current one
    toff[i] = t < d.ntaps ? (((d.tap_off[t][0] - k.off_min[0]) * HY + (d.tap_off[t][1] - k.off_min[1])) * HZ + (d.tap_off[t][2] - k.off_min[2])) * HROW : 0;
  }
  f32x4 acc[MAXT][HG][NTP];
#pragma unroll
  for (int i = 0; i < MAXT; ++i)
#pragma unroll
    for (int h = 0; h < HG; ++h)
#pragma unroll
      for (int p = 0; p < NTP; ++p) acc[i][h][p] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- per-thread constants of the two DMA tiles (same scheme as igemm.hip: tile-independent part precomputed once) ----
  const int PX = d.p.x, PY = d.p.y, PZ = d.p.z, QX = d.h.x, QY = d.h.y, QZ = d.h.z;
  const unsigned p_vox_bytes = (unsigned)d.p.pitch * ES, h_vox_bytes = (unsigned)d.h.pitch * ES;
  constexpr int ppp = NTP * ES;   // 16-byte pieces per P voxel row (NTP*16 channels; = k.p_row >> 4)
  const int ppieces = k.tvox * ppp;
  const int hpieces = hvox * HPP;
  unsigned prel[WPP], hrel[WPH];  // 0xffffffff: no piece
  unsigned h2mask = 0;            // bit u: halo piece u lies in part 1 of a two-part H (static per thread)
#pragma unroll
  for (int u = 0; u < WPP; ++u) {
    const int j = (u * 4 + wave) * 64 + lane;
    unsigned info = 0xffffffffu, rel = 0xffffffffu;
    if (j < ppieces) {
      const int v = j / ppp, c16 = j - v * ppp;
      const int r = vsseg_fdiv(v, T2, iT2), vz = v - r * T2;
      const int vx = vsseg_fdiv(r, T1, iT1), vy = r - vx * T1;
      const bool cok = c16 * EPP + EPP <= d.p.c;  // channels beyond the tensor are zero-filled
      info = (unsigned)vx | ((unsigned)vy << 8) | ((unsigned)vz << 16) | ((unsigned)(cok ? c16 : 255) << 24);
      rel = (unsigned)((vx * PY + vy) * PZ + vz) * p_vox_bytes + (unsigned)c16 * 16u;
    }
    if (u < k.npp) pinfo_l[u * 256 + tid] = info;
    prel[u] = rel;
  }
#pragma unroll
  for (int u = 0; u < WPH; ++u) {
    const int j = (u * 4 + wave) * 64 + lane;
    unsigned info = 0xffffffffu, rel = 0xffffffffu;
    if (j < hpieces) {
      const int hv = j / HPP, c16 = j - hv * HPP;
      const int r = vsseg_fdiv(hv, HZ, iHZ), hz = hv - r * HZ;
      const int hx = vsseg_fdiv(r, HY, iHY), hy = r - hx * HY;
      const int ch = chunk0 * 16 + c16 * EPP;  // first channel of the piece inside H
      const bool cok = ch + EPP <= d.h.c;
      info = (unsigned)hx | ((unsigned)hy << 8) | ((unsigned)hz << 16) | ((unsigned)(cok ? c16 : 255) << 24);
      rel = (unsigned)((hx * QY + hy) * QZ + hz) * h_vox_bytes + (unsigned)c16 * 16u;
      if (d.h.ptr2 != nullptr && ch >= d.h.csplit) h2mask |= 1u << u;
    }
    if (u < k.nph) hinfo_l[u * 256 + tid] = info;
    hrel[u] = rel;
  }
  const bool p_chan_ok = NTP * 16 <= d.p.c, h_chan_ok = chunk0 * 16 + HG * 16 <= d.h.c;  // every piece has real channels (fast path)
  const char* p_base = reinterpret_cast<const char*>(d.p.ptr);
  // two-part H (the skip-connection concat as the convolution input): a 16-byte piece lies in one part (csplit % 16 == 0); part 1's
  // base is biased by -csplit channels so that the same per-piece offsets address it
  const char* h_base = reinterpret_cast<const char*>(d.h.ptr) + (int64_t)chunk0 * 16 * ES;
  const char* h_base1 = d.h.ptr2 != nullptr ? reinterpret_cast<const char*>(d.h.ptr2) + ((int64_t)chunk0 * 16 - d.h.csplit) * ES : h_base;
  const int64_t p_sample = (int64_t)PX * PY * PZ * p_vox_bytes, h_sample = (int64_t)QX * QY * QZ * h_vox_bytes;

  struct TileIdx { int tz, ty, tx, n; };
  auto tile_decode = [&](int b) {
    TileIdx t;
    t.tz = b % k.ntile[2]; b /= k.ntile[2];
    t.ty = b % k.ntile[1]; b /= k.ntile[1];
    t.tx = b % k.ntile[0];
    t.n = b / k.ntile[0];
    return t;
  };
  // Tile walk.  Workgroups are dealt to the 8 XCDs round-robin (workgroup b runs on XCD b % 8), so with the plain walk "tile = b + s*G" the
  // y- and x-neighbours of a tile — which share two halo rows with it — are in flight on OTHER XCDs and every XCD's L2 fetches those rows from
  // HBM again (PMC: 1.46x the algorithmic bytes on the 16-channel full-resolution layers).  With k.walk, XCD x owns the contiguous range
  // [x*tpx, (x+1)*tpx) of the (n, x, y, z)-ordered tile list and its workgroups stride through it: the tiles in flight on one XCD form a
  // compact brick whose shared rows are L2 hits.
  const bool xwalk = k.walk && (gridDim.x % 8) == 0 && k.total_tiles >= 8 * (int64_t)(gridDim.x / 8);
  const int G = xwalk ? (int)gridDim.x / 8 : (int)gridDim.x;
  const int tpx = xwalk ? (int)((k.total_tiles + 7) / 8) : (int)k.total_tiles;
  const int t_first = xwalk ? (int)(blockIdx.x & 7) * tpx : 0;
  const int slot = xwalk ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  int t_cnt = (int)k.total_tiles - t_first;
  t_cnt = t_cnt > tpx ? tpx : (t_cnt < 0 ? 0 : t_cnt);
  const TileIdx step = tile_decode(G);
  auto tile_advance = [&](TileIdx& t) {
    t.tz += step.tz; if (t.tz >= k.ntile[2]) { t.tz -= k.ntile[2]; ++t.ty; }
    t.ty += step.ty; if (t.ty >= k.ntile[1]) { t.ty -= k.ntile[1]; ++t.tx; }
    t.tx += step.tx; if (t.tx >= k.ntile[0]) { t.tx -= k.ntile[0]; ++t.n; }
    t.n += step.n;
  };
  TileIdx t_issue = tile_decode(t_first + slot);
  const int my_tiles = slot < t_cnt ? (t_cnt - 1 - slot) / G + 1 : 0;

  const int bufmask = k.nbuf - 1;
  auto issue = [&](int s) {  // LDS-DMA of tile s into buffer s & bufmask
    const int n = t_issue.n, q0x = t_issue.tx * d.tile[0], q0y = t_issue.ty * d.tile[1], q0z = t_issue.tz * d.tile[2];
    tile_advance(t_issue);
    char* Pdst = Pl + (s & bufmask) * p_bytes;
    char* Hdst = Hl + (s & bufmask) * h_bytes;
    {
      const char* sample = p_base + (int64_t)n * p_sample;
      const bool interior = q0x + d.tile[0] <= d.q[0] && q0y + d.tile[1] <= d.q[1] && q0z + d.tile[2] <= d.q[2] && p_chan_ok;
      if (interior) {
        const char* origin = sample + (int64_t)((q0x * PY + q0y) * PZ + q0z) * p_vox_bytes;
#pragma unroll
        for (int u = 0; u < WPP; ++u) {
          if ((u * 4 + wave) * 64 >= ppieces) break;
          if (prel[u] != 0xffffffffu) wg_dma16(origin + prel[u], Pdst + (u * 4 + wave) * 1024);
        }
      } else {
#pragma unroll
        for (int u = 0; u < WPP; ++u) {
          if ((u * 4 + wave) * 64 >= ppieces) break;
          const unsigned info = pinfo_l[u * 256 + tid];
          if (info != 0xffffffffu) {
            const int qx = q0x + (int)(info & 255u), qy = q0y + (int)((info >> 8) & 255u), qz = q0z + (int)((info >> 16) & 255u);
            const unsigned c16 = info >> 24;
            const bool ok = qx < d.q[0] && qy < d.q[1] && qz < d.q[2] && c16 != 255u;
            const void* src = ok ? (const void*)(sample + (int64_t)((qx * PY + qy) * PZ + qz) * p_vox_bytes + c16 * 16u) : k.zeros;
            wg_dma16(src, Pdst + (u * 4 + wave) * 1024);
          }
        }
      }
    }
    {
      const char* sample = h_base + (int64_t)n * h_sample;
      const char* sample1 = h_base1 + (int64_t)n * h_sample;
      const int gx0 = q0x * d.hs[0] + k.off_min[0], gy0 = q0y * d.hs[1] + k.off_min[1], gz0 = q0z * d.hs[2] + k.off_min[2];
      const bool interior = gx0 >= 0 && gy0 >= 0 && gz0 >= 0 && gx0 + HX <= QX && gy0 + HY <= QY && gz0 + HZ <= QZ && h_chan_ok;
      if (interior) {
        const int64_t ooff = (int64_t)((gx0 * QY + gy0) * QZ + gz0) * h_vox_bytes;
        const char* origin = sample + ooff;
        const char* origin1 = sample1 + ooff;
#pragma unroll
        for (int u = 0; u < WPH; ++u) {
          if ((u * 4 + wave) * 64 >= hpieces) break;
          if (hrel[u] != 0xffffffffu) wg_dma16(((h2mask >> u) & 1u ? origin1 : origin) + hrel[u], Hdst + (u * 4 + wave) * 1024);
        }
      } else {
#pragma unroll
        for (int u = 0; u < WPH; ++u) {
          if ((u * 4 + wave) * 64 >= hpieces) break;
          const unsigned info = hinfo_l[u * 256 + tid];
          if (info != 0xffffffffu) {
            const int gx = gx0 + (int)(info & 255u), gy = gy0 + (int)((info >> 8) & 255u), gz = gz0 + (int)((info >> 16) & 255u);
            const unsigned c16 = info >> 24;
            const bool ok = (unsigned)gx < (unsigned)QX && (unsigned)gy < (unsigned)QY && (unsigned)gz < (unsigned)QZ && c16 != 255u;
            const void* src = ok ? (const void*)(((h2mask >> u) & 1u ? sample1 : sample) + (int64_t)((gx * QY + gy) * QZ + gz) * h_vox_bytes + c16 * 16u) : k.zeros;
            wg_dma16(src, Hdst + (u * 4 + wave) * 1024);
          }
        }
      }
    }
  };

  // optional bias gradient dbias[cP] += sum_q P[q][cP] (P = dY of a convolution without BatchNorm): one extra MFMA per K-step
  // against an all-ones operand in the workgroups of H-chunk 0 — replaces a separate pass over dY (vsseg_channel_sum)
  const bool do_bias = d.dbias_p != nullptr && chunk0 == 0 && wt == 0;
  f32x4 accb[NTP];
#pragma unroll
  for (int p = 0; p < NTP; ++p) accb[p] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ksteps = k.tvox / 32;
  __syncthreads();  // coordinate tables visible
  if (bufmask && my_tiles > 0) issue(0);
  for (int s = 0; s < my_tiles; ++s) {
    if (!bufmask) {  // single buffer: refill only after every wave finished the previous tile
      if (s > 0) __syncthreads();
      issue(s);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile s landed for every wave; every wave is done reading tile s-1's buffers
    if (bufmask && s + 1 < my_tiles) issue(s + 1);
    const char* Ps = Pl + (s & bufmask) * p_bytes;
    const char* Hs = Hl + (s & bufmask) * h_bytes;
    for (int ks = wv; ks < ksteps; ks += k.wv) {
      if constexpr (ES == 2) {
        // lane (g, i=l15): rows r=i>>2 of two 4-voxel blocks, 4-channel column chunk q=i&3
        const int r = l15 >> 2, qc = (l15 & 3) * 8;  // byte offset of the 4-channel chunk
        // K-slot g*8 + j of the MFMA holds tile voxel 4g + j (j < 4) / 16 + 4g + (j - 4) of the K-step — the same permutation for both operands, so
        // the product is unchanged — instead of 8g + j: the two 4-voxel blocks a half-wave reads in one ds_read_b64_tr_b16 are then NEIGHBOURS
        // (voxels 0..7 / 8..15 of the K-step: one contiguous 256-byte z-row of the 16-channel halo tile, 8 rows of 96 bytes of a 48-channel P tile =
        // 8 different 32-byte bank groups) where 8g + j put them 8 voxels = a multiple of 256 bytes apart (47 % bank-conflict cycles, r02_pmc_sq.txt)
        const int v0 = ks * 32 + g * 4 + r, v1 = v0 + 16;
        const int h0 = hbase[v0] * HROW + qc, h1 = hbase[v1] * HROW + qc;
        bf16x8 pa[NTP];
#pragma unroll
        for (int p = 0; p < NTP; ++p) {
          typedef __attribute__((address_space(3))) bf16x4 lds_b4;
          bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b4*)(Ps + v0 * k.p_row + p * 32 + qc));
          bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b4*)(Ps + v1 * k.p_row + p * 32 + qc));
          pa[p] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
        if (do_bias) {
          const short one = 0x3F80;  // bf16 1.0
          const bf16x8 ones = bf16x8{one, one, one, one, one, one, one, one};
#pragma unroll
          for (int p = 0; p < NTP; ++p) accb[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[p], ones, accb[p], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < MAXT; ++i) {
          typedef __attribute__((address_space(3))) bf16x4 lds_b4;
#pragma unroll
          for (int h = 0; h < HG; ++h) {
            bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b4*)(Hs + h0 + toff[i] + h * 32));
            bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b4*)(Hs + h1 + toff[i] + h * 32));
            bf16x8 hb = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
            for (int p = 0; p < NTP; ++p) acc[i][h][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[p], hb, acc[i][h][p], 0, 0, 0);
          }
        }
      } else {
#pragma unroll 2
        for (int ss = 0; ss < 8; ++ss) {
          const int v = ks * 32 + ss * 4 + g;
          const int hb0 = hbase[v] * HROW + l15 * 4;
          float pa[NTP];
#pragma unroll
          for (int p = 0; p < NTP; ++p) pa[p] = *reinterpret_cast<const float*>(Ps + v * k.p_row + (p * 16 + l15) * 4);
          if (do_bias) {
#pragma unroll
            for (int p = 0; p < NTP; ++p) accb[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[p], 1.0f, accb[p], 0, 0, 0);
          }
#pragma unroll
          for (int i = 0; i < MAXT; ++i) {
  #pragma unroll
            for (int h = 0; h < HG; ++h) {
              const float hb = *reinterpret_cast<const float*>(Hs + hb0 + toff[i] + h * 64);
#pragma unroll
              for (int p = 0; p < NTP; ++p) acc[i][h][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[p], hb, acc[i][h][p], 0, 0, 0);
            }
          }
        }
      }
    }
  }

  if (do_bias && l15 == 0) {  // every column of accb holds the same row sums: this wave's row of the bias slab (summed in a fixed order by vsseg_slab_add_kernel)
    float* brow = k.bias_slab + ((int64_t)blockIdx.x * k.wv + wv) * (NTP * 16);
#pragma unroll
    for (int p = 0; p < NTP; ++p)
#pragma unroll
      for (int r = 0; r < 4; ++r) brow[p * 16 + g * 4 + r] = accb[p][r];
  }
  // flush: lane holds rows g*4+r (P channel) x col l15 (H channel) of every owned tile -> this workgroup's slab [chunk][tap][P tile][64 lanes][4]: a tile
  // leaves as ONE coalesced 1 KiB store per wave (the first layout, [tap][cP][16], took four 4-byte stores of four 64-byte segments each: on the levels
  // with few voxels and many channels, where a workgroup's slab is as large as the weight gradient itself, the flush was the longest phase of the launch)
#pragma unroll
  for (int h = 0; h < HG; ++h) {
    if (chunk0 + h >= k.hchunks) break;  // the last group may be short
    float* slab = k.slab + (((int64_t)blockIdx.x * k.wv + wv) * k.hchunks + chunk0 + h) * k.slab_chunk + lane * 4;  // K-steps split over waves (1x1x1 kernels): one slab per share
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      const int t = wt + i * k.wt;
      if (t >= d.ntaps) continue;
#pragma unroll
      for (int p = 0; p < NTP; ++p) *reinterpret_cast<f32x4*>(slab + (t * NTP + p) * 256) = acc[i][h][p];
    }
  }
}

#ifndef WG_INST
// dw[cp][ch][tap] += sum over workgroups of their slabs (vsseg_slab_sum4: 64 x 4 elements x 16 slab lanes per block, fixed summation order); element e of a
// slab = (chunk, tap, P tile, lane, r) as the flushes of wgrad_kernel, mwgrad_kernel and cwgrad_kernel write it: value (cP = tile*16 + (lane>>4)*4 + r, cH = chunk*16 + (lane&15)).
// (One thread per element walking all <= 1024 slabs was a 41 us latency chain per layer: 1.9 ms per step in 45 launches, profiles/r02_kernel_stats.txt.)
__global__ __launch_bounds__(VSSEG_SLAB_THREADS) void wgrad_reduce_kernel(const float* __restrict__ slab, int nblk, int hchunks, int ntp, int slab_chunk, vsseg_wgrad_desc d) {
  __shared__ f32x4 lds[VSSEG_SLAB_THREADS];
  const int64_t total = (int64_t)hchunks * slab_chunk;
  const int64_t i = ((int64_t)blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
  const f32x4 s = vsseg_slab_sum4(slab, total, i, nblk, lds);
  if (threadIdx.x >= 64 || i >= total) return;
  const int chunk = (int)(i / slab_chunk), e = (int)(i - (int64_t)chunk * slab_chunk);
  const int lane = (e >> 2) & 63, tile = e >> 8, ptile = tile % ntp, tap = tile / ntp;
  const int cp0 = ptile * 16 + (lane >> 4) * 4, ch = chunk * 16 + (lane & 15);
  if (ch >= d.ch_valid) return;
  float* dst = d.dw + (int64_t)ch * d.stride_h + (int64_t)d.tap_widx[tap] * d.stride_tap;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (cp0 + r < d.cp_valid) dst[(int64_t)(cp0 + r) * d.stride_p] += s[r];
}

#endif  // WG_INST

template <typename T, int MAXT, int NTP, int HG> static int wg_launch(WgradK& k, dim3& grid, int lds, hipStream_t s) {
  if constexpr (MAXT * NTP * HG > 42) {  // accumulators beyond ~170 VGPRs are never planned (vsseg_wgrad clamps HG): keep the instantiation count down
    vsseg_set_error("vsseg_wgrad: maxt %d x ntp %d x hgroup %d accumulator tiles exceed the register budget", MAXT, NTP, HG);
    return VSSEG_EINVAL;
  } else {
    static bool attr_set_dev[16] = {}; bool& attr_set = vsseg_dev_once(attr_set_dev);  // per device: the LDS opt-in is a per-device function attribute
    if (!attr_set) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<T, MAXT, NTP, HG>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_set = true;
    }
    // persistent grid: never more workgroups than are resident at once (a late workgroup would be a serial tail)
    static int cached_lds = -1, cached_per_cu = 1;
    if (cached_lds != lds) {
      int n = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wgrad_kernel<T, MAXT, NTP, HG>, 256, lds) != hipSuccess || n < 1) n = 1;
      cached_per_cu = n > 4 ? 4 : n;
      cached_lds = lds;
    }
    unsigned cap = (unsigned)(256 * cached_per_cu) / grid.y;
    if (cap < 1) cap = 1;
    if (grid.x > cap) grid.x = cap;
    if (k.walk && grid.x >= 8) grid.x &= ~7u;  // whole rounds of the 8 XCDs (the XCD-contiguous walk needs gridDim.x % 8 == 0)
    hipLaunchKernelGGL((wgrad_kernel<T, MAXT, NTP, HG>), grid, dim3(256), lds, s, k);
    VSSEG_LAUNCH_CHECK("vsseg_wgrad");
    return VSSEG_OK;
  }
}
template <typename T, int MAXT, int NTP> static int wg_hg(WgradK& k, int hg, dim3& grid, int lds, hipStream_t s) {
  switch (hg) {
    case 1: return wg_launch<T, MAXT, NTP, 1>(k, grid, lds, s);
    case 2: return wg_launch<T, MAXT, NTP, 2>(k, grid, lds, s);
    case 3: return wg_launch<T, MAXT, NTP, 3>(k, grid, lds, s);
    case 4: return wg_launch<T, MAXT, NTP, 4>(k, grid, lds, s);
  }
  vsseg_set_error("vsseg_wgrad: hgroup must be 1..4 (got %d)", hg);
  return VSSEG_EINVAL;
}
template <typename T, int MAXT> static int wg_ntp(WgradK& k, int hg, dim3& grid, int lds, hipStream_t s) {
  switch (k.d.ntp) {
    case 1: return wg_hg<T, MAXT, 1>(k, hg, grid, lds, s);
    case 2: return wg_hg<T, MAXT, 2>(k, hg, grid, lds, s);
    case 3: return wg_hg<T, MAXT, 3>(k, hg, grid, lds, s);
    case 4: return wg_hg<T, MAXT, 4>(k, hg, grid, lds, s);
    case 5: return wg_hg<T, MAXT, 5>(k, hg, grid, lds, s);
    case 6: return wg_hg<T, MAXT, 6>(k, hg, grid, lds, s);
  }
  vsseg_set_error("vsseg_wgrad: ntp must be 1..6 (got %d)", k.d.ntp);
  return VSSEG_EINVAL;
}
template <typename T> static int wg_maxt(WgradK& k, int maxt, int hg, dim3& grid, int lds, hipStream_t s) {
  if (maxt <= 1) return wg_ntp<T, 1>(k, hg, grid, lds, s);
  if (maxt <= 3) return wg_ntp<T, 3>(k, hg, grid, lds, s);
  if (maxt <= 7) return wg_ntp<T, 7>(k, hg, grid, lds, s);
  vsseg_set_error("vsseg_wgrad: too many taps per wave (%d)", maxt);
  return VSSEG_EINVAL;
}
// instantiated per element type in wgrad_inst.hip (two translation units that build in parallel)
int vsseg_wgrad_launch_f32(WgradK& k, int maxt, int hg, dim3& grid, int lds, hipStream_t s);
int vsseg_wgrad_launch_bf16(WgradK& k, int maxt, int hg, dim3& grid, int lds, hipStream_t s);

#ifndef WG_INST
// sums `nblk` partial-sum slabs [nblk][hchunks][slab_chunk] into d->dw
int vsseg_wgrad_reduce_launch(const vsseg_wgrad_desc* d, float* slab, int nblk, int hchunks, int slab_chunk, hipStream_t s) {
  const int total = hchunks * slab_chunk;  // a multiple of 256
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total / 4 + 63) / 64), dim3(VSSEG_SLAB_THREADS), 0, s, (const float*)slab, nblk, hchunks, d->ntp, slab_chunk, *d);
  VSSEG_LAUNCH_CHECK("vsseg_wgrad(reduce)");
  return VSSEG_OK;
}

int vsseg_mwgrad_launch(const vsseg_wgrad_desc* d, const void* zeros, hipStream_t s);  // mwgrad.hip
int vsseg_cwgrad_launch(const vsseg_wgrad_desc* d, const void* zeros, hipStream_t s);  // cwgrad.hip

extern "C" int vsseg_wgrad(const vsseg_wgrad_desc* d, void* stream) {
  VSSEG_CHECK(d && d->p.ptr && d->h.ptr && d->dw, "vsseg_wgrad: null pointer");
  if (d->march) {  // 1: marching kernel (mwgrad.hip), 2: compute kernel (cwgrad.hip): they fail loudly outside their domains, never fall back
    static void* z = nullptr;
    if (!z && (hipMalloc(&z, 256) != hipSuccess || hipMemset(z, 0, 256) != hipSuccess)) z = nullptr;
    VSSEG_CHECK(z, "vsseg_wgrad: could not allocate the zero page");
    return d->march == 2 ? vsseg_cwgrad_launch(d, z, as_stream(stream)) : vsseg_mwgrad_launch(d, z, as_stream(stream));
  }
  VSSEG_CHECK(!d->h_gate, "vsseg_wgrad: the gated H operand (h_gate) needs the marching kernel (march = 1)");
  VSSEG_CHECK(d->p.dtype == d->h.dtype, "vsseg_wgrad: dtype mismatch");
  VSSEG_CHECK(d->ntaps >= 1 && d->ntaps <= VSSEG_MAX_TAPS, "vsseg_wgrad: ntaps out of range");
  VSSEG_CHECK(d->p.c % 8 == 0 && d->p.pitch % 8 == 0 && d->h.c % 8 == 0 && d->h.pitch % 8 == 0, "vsseg_wgrad: channels/pitch must be multiples of 8");
  VSSEG_CHECK(!d->p.ptr2, "vsseg_wgrad: P may not be a two-part tensor");
  VSSEG_CHECK(!d->h.ptr2 || (d->h.csplit > 0 && d->h.csplit < d->h.c && d->h.csplit % 16 == 0 && d->h.pitch >= d->h.csplit && d->h.pitch >= d->h.c - d->h.csplit),
              "vsseg_wgrad: bad two-part H (c=%d csplit=%d pitch=%d)", d->h.c, d->h.csplit, d->h.pitch);
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
  k.nbuf = d->single_buffer ? 1 : 2;
  const int hchunks = (d->ch_valid + 15) / 16;
  const int hvox = k.halo[0] * k.halo[1] * k.halo[2];
  // H-chunk group per workgroup: the requested one (0 = 1), clamped to what the accumulators (maxt*ntp*hg <= 42 tiles), the DMA piece
  // budget and the LDS allow; a divisor of the chunk count so that every group is whole (partial groups take the slow boundary path)
  int hg = d->hgroup < 1 ? 1 : (d->hgroup > 4 ? 4 : d->hgroup);
  const int maxt_t = maxt <= 1 ? 1 : (maxt <= 3 ? 3 : 7);
  auto lds_need = [&](int g) { return ((k.tvox * 4 + 15) / 16) * 16 + k.nbuf * (k.tvox * k.p_row + hvox * g * 16 * es) + ((k.tvox * (k.p_row >> 4) + 255) / 256 + (hvox * g * es + 255) / 256) * 1024; };
  while (hg > 1 && (hchunks % hg != 0 || maxt_t * d->ntp * hg > 42 || hvox * hg * 16 * es > WPH * 256 * 16 || lds_need(hg) > 160 * 1024)) --hg;
  k.hchunks = hchunks;
  k.lds_p = off; off += k.nbuf * k.tvox * k.p_row;                     // double-buffered: tile s+1 streams in by LDS-DMA while tile s is multiplied
  k.lds_h = off; off += k.nbuf * hvox * hg * 16 * es;
  k.npp = (k.tvox * (k.p_row >> 4) + 255) / 256;
  {  // every tile whole and every P piece made of real channels: the boundary path of the P tile is never taken, its coordinate table (1 KiB per
     // piece row: 6 KiB on the 48-channel 3x3x3 layers) is not allocated — there that is the difference between two and three resident workgroups
    bool p_whole = d->ntp * 16 <= d->p.c;
    for (int a = 0; a < 3; ++a) p_whole = p_whole && d->q[a] % d->tile[a] == 0;
    if (p_whole) k.npp = 0;
  }
  k.nph = (hvox * hg * es + 255) / 256;
  k.lds_tab = off; off += (k.npp + k.nph) * 1024;
  VSSEG_CHECK(k.tvox * k.p_row <= WPP * 256 * 16 && hvox * hg * 16 * es <= WPH * 256 * 16, "vsseg_wgrad: tile too large for the DMA piece budget");
  for (int a = 0; a < 3; ++a) VSSEG_CHECK(k.halo[a] <= 255 && d->tile[a] <= 255, "vsseg_wgrad: tile/halo extent > 255");
  {
    static void* z = nullptr;
    if (!z && (hipMalloc(&z, 256) != hipSuccess || hipMemset(z, 0, 256) != hipSuccess)) z = nullptr;
    VSSEG_CHECK(z, "vsseg_wgrad: could not allocate the zero page");
    k.zeros = z;
  }
  VSSEG_CHECK(k.total_tiles < (1ll << 31), "vsseg_wgrad: too many tiles");
  VSSEG_CHECK(off <= 160 * 1024, "vsseg_wgrad: needs %d bytes of LDS (> 160 KiB); reduce the tile", off);
  k.slab_chunk = d->ntaps * d->ntp * 16 * 16;
  VSSEG_CHECK(d->scratch && d->scratch_elems >= (int64_t)hchunks * k.slab_chunk, "vsseg_wgrad: scratch too small (%lld < %lld floats)", (long long)d->scratch_elems, (long long)hchunks * k.slab_chunk);
  int64_t gx = d->persistent_blocks > 0 ? d->persistent_blocks : 256;
  if (gx > k.total_tiles) gx = k.total_tiles;
  const int64_t per_blk = (int64_t)k.wv * ((int64_t)hchunks * k.slab_chunk + (d->dbias_p ? d->ntp * 16 : 0));  // slabs (+ bias rows) of one workgroup
  const int64_t cap = d->scratch_elems / per_blk;
  if (gx > cap) gx = cap;
  k.slab = d->scratch;
  k.bias_slab = d->dbias_p ? d->scratch + gx * k.wv * (int64_t)hchunks * k.slab_chunk : nullptr;
  k.walk = 1;
  dim3 grid((unsigned)gx, (unsigned)(hchunks / hg));
  int rc = d->p.dtype == VSSEG_F32 ? vsseg_wgrad_launch_f32(k, maxt, hg, grid, off, as_stream(stream)) : vsseg_wgrad_launch_bf16(k, maxt, hg, grid, off, as_stream(stream));
  if (rc) return rc;
  rc = vsseg_wgrad_reduce_launch(d, d->scratch, (int)grid.x * k.wv, hchunks, k.slab_chunk, as_stream(stream));
  if (rc || !d->dbias_p) return rc;
  hipLaunchKernelGGL(vsseg_slab_add_kernel, dim3((d->ntp * 16 + 63) / 64), dim3(VSSEG_SLAB_THREADS), 0, as_stream(stream), (const float*)k.bias_slab, (int)grid.x * k.wv, d->ntp * 16, d->cp_valid, d->dbias_p);
  VSSEG_LAUNCH_CHECK("vsseg_wgrad(bias)");
  return VSSEG_OK;
}
#endif  // WG_INST
