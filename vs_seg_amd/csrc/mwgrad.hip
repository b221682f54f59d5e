// Marching weight gradient (vsseg_wgrad descriptors with march = 1): the HBM-bound weight gradients of the stride-1 3x3x1 bf16 convolutions of
// the two finest levels (autograd of ref:params/networks/blocks/convolutions.py:114-146 as run by `loss.backward()`, ref:params/VSparams.py:461;
// SURVEY §8a rows 2, 5, 6, 41, 44, 46, 49) with both operands fetched from HBM exactly once.
//
//   dW[tap][cP][cH] += sum_q P[q][cP] * H[q + off_tap][cH]        Conv3d: P = dY, H = X
//
// The tile kernel (wgrad.hip) fetches an (8+2)x(8+2)x4 halo of H per 8x8x4 tile and the P tile once per H-chunk group: 1.16-1.33x the
// algorithmic bytes (profiles/r02_pmc_hbm.txt), the slowest kernel family per byte.  Here, as in mconv.hip, a workgroup owns a column (sample,
// TYB rows, TZ slices) and marches along x: planes of H go through a ring of four LDS slots (the stencil reads x-1, x, x+1 while x+2 is in
// flight), planes of P through two; nothing is fetched twice except two H planes per x segment.
//
//   * same LDS plane layout as mconv.hip ([row][piece'][z], tools/lds_conflicts.py); the reduction axis (voxels) is the strided one, so both MFMA
//     operands are built with ds_read_b64_tr_b16 from 4 voxel x 16 channel blocks.  A K-step is 32 consecutive voxels of the plane; K-slot
//     g*8 + j holds voxel 4g + j (j < 4) / 16 + 4g + (j - 4): the two 4-voxel blocks a half-wave reads together lie in different halves of a
//     256-byte bank row (they are row neighbours, and the layout's swizzle alternates with the row) — no bank conflicts
//   * the accumulators (9 taps x cH tiles x cP tiles) live in registers over the whole march; waves split the K-steps (every wave all taps:
//     no operand is read twice) or, where those accumulators do not fit, the (tap, cH tile) units (the P fragments are then read by every wave)
//   * flush: cross-wave sum in a fixed order through LDS, one partial-sum slab per workgroup, wgrad.hip's fixed-order reduction sums the slabs
#include "common.h"
#include "mwgrad.h"

constexpr int MW_NR = 4;  // ring slots of H: planes x-1, x, x+1 + the one in flight

struct MwgradK {
  const char* h0; const char* h1;  // H (ring operand): channels [0, csplit) / [csplit, c) biased by -csplit channels
  const char* p;                   // P (centre operand), one part
  float* slab;                     // [gridDim.x][NTH][9][NTP][64 lanes][4] (wgrad.hip's slab layout: a tile leaves as one 1 KiB store)
  float* dbias;                    // optional: [gridDim.x][NTP*16] rows of partial bias gradients sum_q P[q][cP] (summed in a fixed order by vsseg_slab_add_kernel)
  const float* h_gate;             // GIN: fp32 attention map of H: voxel v of H is multiplied by (1 + h_gate[v]) on load (mconv.hip, MODE 3)
  const void* zeros;
  int h_csplit_pc, h_vox_bytes, p_vox_bytes, cp_valid;
  int X, Y, Z;
  int lx, nxs, nyb, nzb;
};

typedef __attribute__((address_space(3))) bf16x4 mw_lds_b4;
__device__ __forceinline__ bf16x8 mw_tr(const char* lo, const char* hi) {
  const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((mw_lds_b4*)lo);
  const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((mw_lds_b4*)hi);
  return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

// CH: channels of H (16, 32, 64); CP: channels of P as stored (8, 16, 32: 8 = a 1/2-channel gradient zero-extended to one channel group);
// UNITSPLIT: waves split the (tap, cH tile) units instead of the K-steps
// PC2 (CP 8 only): P is a COMPACT two-channel tensor [N][X][Y][Z][2] (4 bytes per voxel: the gradient of the two logits) instead of its zero-extension to one
// 8-channel group: the thread that owns a plane slot loads the voxel's pair one step ahead and writes the zero-extended 16-byte piece into the P buffer itself.
template <int CH, int CP, int TZ, int MT, bool UNITSPLIT, bool GIN, bool PC2 = false>
__global__ __launch_bounds__(256, 2) void mwgrad_kernel(const MwgradK k) {
  static_assert(!PC2 || CP == 8, "a compact P operand is one zero-extended channel group");
  constexpr int GH = CH / 8, GP = CP / 8, RSH = TZ * GH, RSP = TZ * GP, RPM = 16 / TZ, TYB = MT * 4 * RPM, ROWS = TYB + 2;
  constexpr int NTH = CH / 16, NTP = CP >= 16 ? CP / 16 : 1;
  constexpr int HSLOTS = ROWS * RSH, PSLOTS = TYB * RSP;
  constexpr int HPLANE = (HSLOTS * 16 + 255) / 256 * 256, PPLANE = (PSLOTS * 16 + 255) / 256 * 256;
  constexpr int HINST = (HSLOTS + 255) / 256, PINST = (PSLOTS + 255) / 256;
  constexpr int KS = 2 * MT;            // K-steps (32 voxels) per plane
  constexpr int UNITS = 9 * NTH;        // (tap, cH tile) pairs
  constexpr int MYU = UNITSPLIT ? (UNITS + 3) / 4 : UNITS;
  constexpr int KSTEP_BYTES_H = 32 * GH * 16, KSTEP_BYTES_P = 32 * GP * 16;  // 32 voxels further down the plane (whole rows: the swizzle term is unchanged)
  static_assert(UNITSPLIT || MT % 2 == 0, "the K-step split needs a multiple of 4 K-steps per plane");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Hl = smem;
  char* Pl = smem + MW_NR * HPLANE;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, l15 = lane & 15;
  const int X = k.X, Y = k.Y, Z = k.Z;

  int b = vsseg_xcd_contiguous(blockIdx.x, gridDim.x);
  const int zb = b % k.nzb; b /= k.nzb;
  const int yb = b % k.nyb; b /= k.nyb;
  const int xs = b % k.nxs; const int n = b / k.nxs;
  const int y0 = yb * TYB, z0 = zb * TZ, xb = xs * k.lx, steps = min(k.lx, X - xb);

  for (int i = tid; i < (MW_NR * HPLANE + 2 * PPLANE) / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0u, 0u, 0u, 0u);  // H rows outside the image stay zero

  // ---- DMA pieces (mconv.hip's scheme): LDS slot j of a plane holds (row j / RS, piece' (j % RS) / TZ, z j % TZ)
  int hrel[HINST], prel[PINST], grel[GIN ? HINST : 1];
  unsigned hok = 0, h1m = 0, pok = 0;
#pragma unroll
  for (int u = 0; u < HINST; ++u) {
    const int j = (u * 4 + wave) * 64 + lane;
    const int r = j / RSH, within = j % RSH, pp = within / TZ, z = within % TZ;
    const int pc = (pp - 2 * (r * RSH / 16)) & (GH - 1);
    const int gy = y0 + r - 1;
    const bool ok = j < HSLOTS && (unsigned)gy < (unsigned)Y;
    hrel[u] = ok ? ((r - 1) * Z + z) * k.h_vox_bytes + pc * 16 : 0;
    if constexpr (GIN) grel[u] = ok ? (r - 1) * Z + z : 0;
    if (ok) hok |= 1u << u;
    if (ok && pc >= k.h_csplit_pc) h1m |= 1u << u;
  }
#pragma unroll
  for (int u = 0; u < PINST; ++u) {
    const int j = (u * 4 + wave) * 64 + lane;
    const int r = j / RSP, within = j % RSP, pp = within / TZ, z = within % TZ;
    const int pc = (pp - 2 * (r * RSP / 16)) & (GP - 1);
    const bool ok = j < PSLOTS;
    prel[u] = ok ? (r * Z + z) * k.p_vox_bytes + (PC2 ? 0 : pc * 16) : 0;
    if (ok) pok |= 1u << u;
  }
  const int64_t col0 = (((int64_t)n * X) * Y + y0) * Z + z0;
  const int64_t hstride = (int64_t)Y * Z * k.h_vox_bytes, pstride = (int64_t)Y * Z * k.p_vox_bytes;
  const char* horg0 = k.h0 + col0 * k.h_vox_bytes;
  const char* horg1 = k.h1 + col0 * k.h_vox_bytes;
  const char* porg = k.p + col0 * k.p_vox_bytes;
  auto issue_h = [&](int i) {  // plane i (x = xb - 1 + i) of H into ring slot i & 3; planes outside the image are zero
    const int x = xb - 1 + i;
    char* dst = Hl + (i & (MW_NR - 1)) * HPLANE;
    const bool inside = (unsigned)x < (unsigned)X;
    const char* q0 = horg0 + (int64_t)x * hstride;
    const char* q1 = horg1 + (int64_t)x * hstride;
#pragma unroll
    for (int u = 0; u < HINST; ++u)
      if ((hok >> u) & 1u) vsseg_dma16(inside ? (const void*)(((h1m >> u) & 1u ? q1 : q0) + hrel[u]) : k.zeros, dst + (u * 4 + wave) * 1024);
  };
  const float* gcol = GIN ? k.h_gate + col0 : nullptr;
  auto load_gate = [&](int i, float (&gv)[GIN ? HINST : 1]) {  // gate values of this thread's pieces of H plane i: ordinary loads in front of the plane's DMAs
    if constexpr (GIN) {
      const int x = xb - 1 + i;
      const float* gp = gcol + (int64_t)((unsigned)x < (unsigned)X ? x : 0) * Y * Z;
#pragma unroll
      for (int u = 0; u < HINST; ++u) gv[u] = gp[grel[u]];
    }
  };
  auto apply_gate = [&](int i, const float (&gv)[GIN ? HINST : 1]) {  // every thread gates the pieces IT fetched, in LDS, in front of the step's barrier
    if constexpr (GIN) {
      char* dst = Hl + (i & (MW_NR - 1)) * HPLANE + lane * 16;
#pragma unroll
      for (int u = 0; u < HINST; ++u) {
        if (!((hok >> u) & 1u)) continue;
        uint4* p = reinterpret_cast<uint4*>(dst + (u * 4 + wave) * 1024);
        const uint4 qv = *p;
        const float gg = 1.f + gv[u];
        uint4 o;
        o.x = f2bf2(vsseg_mul_unpacked(__uint_as_float(qv.x << 16), gg), vsseg_mul_unpacked(__uint_as_float(qv.x & 0xffff0000u), gg));
        o.y = f2bf2(vsseg_mul_unpacked(__uint_as_float(qv.y << 16), gg), vsseg_mul_unpacked(__uint_as_float(qv.y & 0xffff0000u), gg));
        o.z = f2bf2(vsseg_mul_unpacked(__uint_as_float(qv.z << 16), gg), vsseg_mul_unpacked(__uint_as_float(qv.z & 0xffff0000u), gg));
        o.w = f2bf2(vsseg_mul_unpacked(__uint_as_float(qv.w << 16), gg), vsseg_mul_unpacked(__uint_as_float(qv.w & 0xffff0000u), gg));
        *p = o;
      }
    }
  };
  unsigned pcv[PC2 ? PINST : 1];
  auto issue_p = [&](int i) {  // plane i of P (always inside the image) into buffer i & 1 (PC2: its values into registers; written by store_p in front of the next barrier)
    const char* q = porg + (int64_t)(xb - 1 + i) * pstride;
    char* dst = Pl + (i & 1) * PPLANE;
#pragma unroll
    for (int u = 0; u < PINST; ++u) {
      if constexpr (PC2) pcv[u] = ((pok >> u) & 1u) ? *reinterpret_cast<const unsigned*>(q + prel[u]) : 0u;
      else if ((pok >> u) & 1u) vsseg_dma16(q + prel[u], dst + (u * 4 + wave) * 1024);
    }
  };
  auto store_p = [&](int i) {  // PC2: buffer i & 1 held plane i-2, read last in step i-2 by waves that have all passed the barrier of step i-1
    if constexpr (PC2) {
      char* dst = Pl + (i & 1) * PPLANE + lane * 16;
#pragma unroll
      for (int u = 0; u < PINST; ++u)
        if ((pok >> u) & 1u) *reinterpret_cast<uint4*>(dst + (u * 4 + wave) * 1024) = make_uint4(pcv[u], 0u, 0u, 0u);
    }
  };

  // ---- transpose-read addressing.  Lane (g, i = l15): voxel r4 = i >> 2 of a 4-voxel block, 4-channel chunk q = i & 3 of a 16-channel tile.
  //      Block `lo` = voxels 4g .. 4g+3 of the K-step, `hi` = 16 + 4g .. ; voxel v -> (row v / TZ, z v % TZ)
  const int r4 = l15 >> 2, q = l15 & 3;
  auto h_off = [&](int v, int dy, int th) {  // byte offset inside a ring slot of (voxel v shifted by dy rows, tile th, this lane's chunk)
    const int row = v / TZ + 1 + dy, z = v % TZ, pc = 2 * th + (q >> 1);
    return (row * RSH + ((pc + 2 * (row * RSH / 16)) & (GH - 1)) * TZ + z) * 16 + (q & 1) * 8;
  };
  auto p_off = [&](int v, int tp) {
    const int row = v / TZ, z = v % TZ;
    const int pc = CP >= 16 ? 2 * tp + (q >> 1) : 0;  // CP == 8: chunks 2, 3 re-read the one real channel group (their rows of the result are never stored)
    return (row * RSP + ((pc + 2 * (row * RSP / 16)) & (GP - 1)) * TZ + z) * 16 + (q & 1) * 8;
  };
  const int vlo = 4 * g + r4, vhi = 16 + 4 * g + r4;
  int hlo[3][NTH], hhi[3][NTH], plo[NTP], phi[NTP];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int th = 0; th < NTH; ++th) { hlo[d][th] = h_off(vlo, d - 1, th); hhi[d][th] = h_off(vhi, d - 1, th); }
#pragma unroll
  for (int tp = 0; tp < NTP; ++tp) { plo[tp] = p_off(vlo, tp); phi[tp] = p_off(vhi, tp); }

  f32x4 acc[MYU][NTP];
#pragma unroll
  for (int u = 0; u < MYU; ++u)
#pragma unroll
    for (int tp = 0; tp < NTP; ++tp) acc[u][tp] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 accb[NTP];
#pragma unroll
  for (int tp = 0; tp < NTP; ++tp) accb[tp] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_bias = k.dbias != nullptr && (!UNITSPLIT || wave == 0);

  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();  // the buffers are zeroed before any DMA writes them
  float gin0[GIN ? HINST : 1], gin1[GIN ? HINST : 1], gin[GIN ? HINST : 1];
  load_gate(0, gin0);
  load_gate(1, gin1);
  load_gate(2, gin);
  issue_h(0);
  issue_h(1);
  issue_h(2);
  issue_p(1);
  if constexpr (GIN) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    apply_gate(0, gin0);
    apply_gate(1, gin1);
    apply_gate(2, gin);
  }

  for (int i = 1; i <= steps; ++i) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (GIN) {
      if (i > 1) apply_gate(i + 1, gin);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if constexpr (PC2) {
      store_p(i);  // the values of P plane i, loaded one step ago
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();  // H plane i+1 and P plane i have landed for every wave; every wave has finished step i-1
    if (i + 2 <= steps + 1) { load_gate(i + 2, gin); issue_h(i + 2); }
    if (i + 1 <= steps) issue_p(i + 1);
    const char* Ps = Pl + (i & 1) * PPLANE;
    const char* Hs[3] = {Hl + ((i - 1) & (MW_NR - 1)) * HPLANE, Hl + (i & (MW_NR - 1)) * HPLANE, Hl + ((i + 1) & (MW_NR - 1)) * HPLANE};
#pragma unroll
    for (int kk = 0; kk < (UNITSPLIT ? KS : KS / 4); ++kk) {
      const int ks = UNITSPLIT ? kk : kk * 4 + wave;
      bf16x8 pa[NTP];
#pragma unroll
      for (int tp = 0; tp < NTP; ++tp) pa[tp] = mw_tr(Ps + plo[tp] + ks * KSTEP_BYTES_P, Ps + phi[tp] + ks * KSTEP_BYTES_P);
      if (do_bias) {
        const short one = 0x3F80;  // bf16 1.0
        const bf16x8 ones = bf16x8{one, one, one, one, one, one, one, one};
#pragma unroll
        for (int tp = 0; tp < NTP; ++tp) accb[tp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[tp], ones, accb[tp], 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < MYU; ++u) {
        const int unit = UNITSPLIT ? u * 4 + wave : u;  // (tap, cH tile) = (unit / NTH, unit % NTH)
        if (UNITSPLIT && unit >= UNITS) break;
        const int tap = unit / NTH, th = unit % NTH, dx = tap / 3, dy = tap % 3;
        int olo, ohi;
        if constexpr (UNITSPLIT) {  // the unit is wave-dependent: select the precomputed offsets (wave-uniform selects)
          olo = hlo[0][0]; ohi = hhi[0][0];
#pragma unroll
          for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int t2 = 0; t2 < NTH; ++t2)
              if (d == dy && t2 == th) { olo = hlo[d][t2]; ohi = hhi[d][t2]; }
        } else {
          olo = hlo[dy][th]; ohi = hhi[dy][th];
        }
        const char* hb = (dx == 0 ? Hs[0] : (dx == 1 ? Hs[1] : Hs[2])) + ks * KSTEP_BYTES_H;
        const bf16x8 hv = mw_tr(hb + olo, hb + ohi);
#pragma unroll
        for (int tp = 0; tp < NTP; ++tp) acc[u][tp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[tp], hv, acc[u][tp], 0, 0, 0);
      }
    }
  }

  // ---- flush.  Lane holds rows g*4 + r (P channel) x column l15 (H channel) of every owned (tap, cH tile, cP tile): its four values are 16 bytes of the slab.
  __syncthreads();  // the ring is free: it becomes the cross-wave reduction buffer
  float* red = reinterpret_cast<float*>(smem);
  float* slab = k.slab + (int64_t)blockIdx.x * (NTH * 9 * NTP * 256);
  if constexpr (UNITSPLIT) {  // every wave owns its outputs outright
#pragma unroll
    for (int u = 0; u < MYU; ++u) {
      const int unit = u * 4 + wave;
      if (unit >= UNITS) break;
      const int tap = unit / NTH, th = unit % NTH;
#pragma unroll
      for (int tp = 0; tp < NTP; ++tp) *reinterpret_cast<f32x4*>(slab + ((th * 9 + tap) * NTP + tp) * 256 + lane * 4) = acc[u][tp];
    }
  } else {  // the four waves hold partial sums over their K-steps: added in wave order through LDS (run-to-run bit-identical)
    constexpr int NACC = UNITS * NTP * 256;  // floats
    static_assert(NACC * 4 <= MW_NR * HPLANE + 2 * PPLANE, "reduction buffer does not fit the ring");
    for (int w = 0; w < 4; ++w) {
      if (wave == w) {
#pragma unroll
        for (int u = 0; u < MYU; ++u) {
          const int tap = u / NTH, th = u % NTH;
#pragma unroll
          for (int tp = 0; tp < NTP; ++tp) {
            f32x4* dst = reinterpret_cast<f32x4*>(red + ((th * 9 + tap) * NTP + tp) * 256 + lane * 4);
            if (w == 0) *dst = acc[u][tp];
            else if (w < 3) *dst += acc[u][tp];
            else *reinterpret_cast<f32x4*>(slab + (reinterpret_cast<float*>(dst) - red)) = *dst + acc[u][tp];
          }
        }
      }
      __syncthreads();
    }
  }
  if (k.dbias != nullptr) {  // every column of accb holds the same row sums -> this workgroup's row of the bias slab
    float* bred = red + UNITS * NTP * 256;  // behind the accumulator image
    float* brow = k.dbias + (int64_t)blockIdx.x * (NTP * 16);
    if (UNITSPLIT) {
      if (wave == 0 && l15 == 0) {
#pragma unroll
        for (int tp = 0; tp < NTP; ++tp)
#pragma unroll
          for (int r = 0; r < 4; ++r) brow[tp * 16 + g * 4 + r] = accb[tp][r];
      }
    } else {
      if (l15 == 0) {
#pragma unroll
        for (int tp = 0; tp < NTP; ++tp)
#pragma unroll
          for (int r = 0; r < 4; ++r) bred[wave * (NTP * 16) + tp * 16 + g * 4 + r] = accb[tp][r];
      }
      __syncthreads();
      if (tid < NTP * 16) brow[tid] = (bred[tid] + bred[NTP * 16 + tid]) + (bred[2 * NTP * 16 + tid] + bred[3 * NTP * 16 + tid]);  // the four waves' K-step shares, fixed order
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------
template <int CH, int CP, int TZ, int MT> static int mw_lds() {
  constexpr int RPM = 16 / TZ, TYB = MT * 4 * RPM, ROWS = TYB + 2;
  return MW_NR * ((ROWS * TZ * (CH / 8) * 16 + 255) / 256 * 256) + 2 * ((TYB * TZ * (CP / 8) * 16 + 255) / 256 * 256);
}
template <int CH, int CP, int TZ, int MT, bool US, bool GIN> static int mw_launch_g(const MwgradK& k, int grid, hipStream_t s) {
  static bool init_dev[16] = {}; bool& init = vsseg_dev_once(init_dev);  // per device: the LDS opt-in is a per-device function attribute
  const int lds = mw_lds<CH, CP, TZ, MT>();
  if (!init) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&mwgrad_kernel<CH, CP, TZ, MT, US, GIN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    init = true;
  }
  hipLaunchKernelGGL((mwgrad_kernel<CH, CP, TZ, MT, US, GIN>), dim3((unsigned)grid), dim3(256), lds, s, k);
  VSSEG_LAUNCH_CHECK("vsseg_wgrad (marching)");
  return VSSEG_OK;
}
template <int CH, int CP, int TZ, int MT, bool US> static int mw_launch_pc2(const MwgradK& k, int grid, hipStream_t s) {  // gated H + compact two-channel P: the logits convolution
  static bool init_dev[16] = {}; bool& init = vsseg_dev_once(init_dev);  // per device: the LDS opt-in is a per-device function attribute
  const int lds = mw_lds<CH, CP, TZ, MT>();
  if (!init) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&mwgrad_kernel<CH, CP, TZ, MT, US, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&mwgrad_kernel<CH, CP, TZ, MT, US, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    init = true;
  }
  if (k.h_gate) hipLaunchKernelGGL((mwgrad_kernel<CH, CP, TZ, MT, US, true, true>), dim3((unsigned)grid), dim3(256), lds, s, k);
  else hipLaunchKernelGGL((mwgrad_kernel<CH, CP, TZ, MT, US, false, true>), dim3((unsigned)grid), dim3(256), lds, s, k);
  VSSEG_LAUNCH_CHECK("vsseg_wgrad (marching, compact P)");
  return VSSEG_OK;
}
template <int CH, int CP, int TZ, int MT, bool US> static int mw_launch(const MwgradK& k, int grid, hipStream_t s) {
  if (k.p_vox_bytes == 4) {
    if constexpr (CH == 32 && CP == 8) return mw_launch_pc2<CH, CP, TZ, MT, US>(k, grid, s);
    else { vsseg_set_error("vsseg_wgrad: no marching-kernel instantiation with a compact two-channel P for this shape"); return VSSEG_EINVAL; }
  }
  if (k.h_gate) {
    if constexpr (CH == 32 && CP == 8) return mw_launch_g<CH, CP, TZ, MT, US, true>(k, grid, s);  // the level-0 decoder convolution behind the attention gate
    else { vsseg_set_error("vsseg_wgrad: no marching-kernel instantiation with the gated H operand for this shape"); return VSSEG_EINVAL; }
  }
  return mw_launch_g<CH, CP, TZ, MT, US, false>(k, grid, s);
}

typedef int (*mw_fn_t)(const MwgradK&, int, hipStream_t);
struct MwEntry { int ch, cp, tz, mt; mw_fn_t fn; int (*lds)(); };
#define MW_E(H, P, Z, M, US) {H, P, Z, M, mw_launch<H, P, Z, M, US>, mw_lds<H, P, Z, M>}
// (H channels, P channels as stored, TZ, M-tiles per wave) — rows per workgroup TYB = 64 * MT / TZ
static const MwEntry mw_table[] = {
    MW_E(16, 16, 4, 4, false), MW_E(16, 16, 4, 2, false), MW_E(16, 16, 8, 4, false), MW_E(16, 16, 8, 8, false),  // 16 -> 16 (level 0)
    MW_E(32, 16, 4, 4, false), MW_E(32, 16, 4, 2, false), MW_E(32, 16, 2, 2, false),                             // 32 -> 16 (level 0 attention conv, two-part H)
    MW_E(32, 8, 4, 4, false), MW_E(32, 8, 4, 2, false),                                                         // 32 -> 2 (logits)
    MW_E(16, 32, 4, 4, false), MW_E(16, 32, 8, 4, false), MW_E(16, 32, 4, 2, false),                             // 16 -> 32 (level 1)
    MW_E(32, 32, 4, 4, true), MW_E(32, 32, 4, 2, true), MW_E(32, 32, 2, 2, true),                                // 32 -> 32
    MW_E(64, 32, 2, 2, true), MW_E(64, 32, 2, 1, true), MW_E(64, 32, 4, 2, true)};                               // 64 -> 32 (level 1 attention conv / decoder unit)

int vsseg_wgrad_reduce_launch(const vsseg_wgrad_desc* d, float* slab, int nblk, int hchunks, int slab_chunk, hipStream_t s);  // wgrad.hip

static const MwEntry* mw_find(const vsseg_wgrad_desc* d, const char** why) {
  *why = nullptr;
  auto no = [&](const char* w) { *why = w; return (const MwEntry*)nullptr; };
  if (d->p.dtype != VSSEG_BF16 || d->h.dtype != VSSEG_BF16) return no("operands are not bf16");
  if (d->ntaps != 9) return no("needs the 9 taps of a 3x3x1 stencil");
  for (int a = 0; a < 3; ++a)
    if (d->hs[a] != 1) return no("stride-1 only");
  for (int t = 0; t < 9; ++t)
    if (d->tap_off[t][0] != t / 3 - 1 || d->tap_off[t][1] != t % 3 - 1 || d->tap_off[t][2] != 0) return no("taps are not the 3x3x1 stencil in (x, y) order");
  if (d->q[0] != d->p.x || d->q[1] != d->p.y || d->q[2] != d->p.z || d->q[0] != d->h.x || d->q[1] != d->h.y || d->q[2] != d->h.z || d->p.n != d->h.n) return no("lattice, P and H extents differ");
  if (d->p.ptr2) return no("P may not be a two-part tensor");
  const int tz = d->tile[2], tyb = d->tile[1];
  if ((tz != 2 && tz != 4 && tz != 8) || tyb < 1 || (tyb * tz) % 64 || d->tile[0] < 1) return no("tile must be (x steps per workgroup, rows, tz in {2, 4, 8}) with rows * tz a multiple of 64");
  const int mt = tyb * tz / 64;
  if (d->q[1] % tyb || d->q[2] % tz) return no("extent is not a multiple of the column block");
  const bool pc2 = d->p.c == 2 && d->p.pitch == 2;  // P = a compact two-channel tensor standing for one zero-extended channel group
  if (d->h.c % 16 || d->h.pitch % 8 || ((uintptr_t)d->h.ptr & 15) || ((uintptr_t)d->h.ptr2 & 15) || ((uintptr_t)d->p.ptr & (pc2 ? 3 : 15)) || (!pc2 && d->p.pitch % 8)) return no("operand alignment");
  if (d->ch_valid != d->h.c || d->cp_valid > d->p.c || d->ntp != (d->p.c >= 16 ? d->p.c / 16 : 1)) return no("channel counts");
  for (const MwEntry& e : mw_table)
    if (e.ch == d->h.c && e.cp == (pc2 ? 8 : d->p.c) && e.tz == tz && e.mt == mt) return &e;
  return no("no instantiation for this (H channels, P channels, tz, rows)");
}

int vsseg_mwgrad_launch(const vsseg_wgrad_desc* d, const void* zeros, hipStream_t s) {
  const char* why;
  const MwEntry* e = mw_find(d, &why);
  if (!e) { vsseg_set_error("vsseg_wgrad: march = 1 (marching kernel) not applicable: %s", why); return VSSEG_EINVAL; }
  MwgradK k;
  k.h0 = reinterpret_cast<const char*>(d->h.ptr);
  k.h1 = d->h.ptr2 ? reinterpret_cast<const char*>(d->h.ptr2) - (int64_t)d->h.csplit * 2 : k.h0;
  k.h_csplit_pc = d->h.ptr2 ? d->h.csplit / 8 : 1 << 20;
  k.h_vox_bytes = d->h.pitch * 2;
  k.p = reinterpret_cast<const char*>(d->p.ptr);
  k.p_vox_bytes = d->p.pitch * 2;
  k.cp_valid = d->cp_valid;
  k.h_gate = d->h_gate;
  k.zeros = zeros;
  k.X = d->q[0]; k.Y = d->q[1]; k.Z = d->q[2];
  k.lx = d->tile[0] > k.X ? k.X : d->tile[0];
  k.nxs = (k.X + k.lx - 1) / k.lx; k.nyb = k.Y / d->tile[1]; k.nzb = k.Z / d->tile[2];
  const int nth = d->h.c / 16, slab_chunk = 9 * d->ntp * 16 * 16;
  const int64_t per_blk = (int64_t)nth * slab_chunk;
  int64_t grid = (int64_t)d->p.n * k.nxs * k.nyb * k.nzb;
  VSSEG_CHECK(grid > 0 && grid < (1ll << 24), "vsseg_wgrad: bad marching grid");
  const int brow = d->ntp * 16;
  VSSEG_CHECK(d->scratch && d->scratch_elems >= grid * (per_blk + (d->dbias_p ? brow : 0)), "vsseg_wgrad: scratch too small for %lld marching workgroups (%lld < %lld floats); use longer x segments", (long long)grid,
              (long long)d->scratch_elems, (long long)(grid * per_blk));
  k.slab = d->scratch;
  k.dbias = d->dbias_p ? d->scratch + grid * per_blk : nullptr;  // bias rows behind the slabs
  int rc = e->fn(k, (int)grid, s);
  if (rc) return rc;
  rc = vsseg_wgrad_reduce_launch(d, d->scratch, (int)grid, nth, slab_chunk, s);
  if (rc || !d->dbias_p) return rc;
  hipLaunchKernelGGL(vsseg_slab_add_kernel, dim3((brow + 63) / 64), dim3(VSSEG_SLAB_THREADS), 0, s, (const float*)k.dbias, (int)grid, brow, d->cp_valid, d->dbias_p);
  VSSEG_LAUNCH_CHECK("vsseg_wgrad (marching, bias)");
  return VSSEG_OK;
}
