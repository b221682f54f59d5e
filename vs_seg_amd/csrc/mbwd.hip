// Fused backward of a stride-1 3x3x1 Convolution block (vsseg_conv_bwd_fused): the second pass of the BatchNorm -> Dropout -> PReLU backward APPLIED ON LOAD,
// the data gradient and the weight gradient of the convolution in ONE marching launch.
//
//   reference: Conv3d -> BatchNorm3d -> Dropout -> PReLU (ref:params/networks/blocks/convolutions.py:114-156), differentiated by `loss.backward()`
//   (ref:params/VSparams.py:461).  SURVEY §8a rows 2, 5, 6, 44 (the 16/32-channel layers of levels 0-1).
//
// Unfused, a layer's backward moves (T = one tensor of the layer's output size): vsseg_bn_act_bwd_apply reads y, dA and writes dy (3T), the data gradient
// reads dy (T), the weight gradient reads dy and x (T + Tx).  Here dy never exists in HBM: a workgroup owns a column (sample, TYB rows, TZ slices) and marches
// along x exactly as mconv.hip / mwgrad.hip do; the ring of four LDS planes holds dy, formed by the thread that fetched the (y, dA, keep-mask) pieces of a
// plane — ordinary loads into registers one step ahead, bn_bwd.h's arithmetic (bit-identical to vsseg_bn_act_bwd_apply), one ds_write per piece.  From the
// SAME planes
//   * the data gradient is mconv.hip's stencil:  dX[q] = sum_t W'[t] dy[q + off_t]     (same packed weights, K order and MFMA order: bit-identical to the
//     marching data-gradient launch on a materialised dy), stored as plain bf16 rows;
//   * the weight gradient is mwgrad.hip's, with the operands' roles swapped: the ring operand is dy, the centre operand is x (planes through two LDS buffers
//     by LDS-DMA):  acc[t][ci][co] = sum_q x[q][ci] * dy[q + off_t][co] = dW[co][ci][mirror(t)] — transpose reads (ds_read_b64_tr_b16), accumulators in
//     registers over the whole march, fixed-order flush into per-workgroup slabs, wgrad.hip's fixed-order slab reduction.
// HBM traffic: y + dA + x in, dX out (4T for a 16 -> 16 layer) against 3T + 2T + 2T + T.
//
// RES: the ResidualUnit's 1x1x1 residual convolution of the SAME input x (ref:params/networks/blocks/convolutions.py:241-255: out = conv-path(x) + residual(x)) rides
// along: its output gradient dR (the unit's dA — RES 1: the very tensor this launch already loads; RES 2: another tensor, the encoder units whose residual is
// added behind the second convolution) goes raw through two more LDS planes, its data gradient Wr' dR[q] is one more centre-tap K-step of the stencil (dX gets
// both terms in one store instead of a second read-modify-write launch) and its weight gradient sum_q x[q] dR[q] one more "tap" of the accumulators.
#include "common.h"
#include "bn_bwd.h"
#include "mbwd.h"

constexpr int MB_NR = 4;  // ring slots of dy: planes x-1, x, x+1 + the one being formed

struct MbwdK {
  const char* y; const char* da;       // conv output (before BatchNorm), gradient of the block's output: CY channels each
  const unsigned char* keep;           // keep-mask bytes [voxel][CY / 8] of the forward, or nullptr (no dropout)
  const char* x;                       // conv input, CX channels: channels [0, csplit) ...
  const char* x1;                      // ... and [csplit, CX), biased by -csplit channels (== x for a one-part tensor)
  int x_csplit_pc;                     // first 16-byte piece of a voxel row that lives in part 1
  const float* x_gate;                 // XG: fp32 attention map of x: voxel v of x is multiplied by (1 + x_gate[v]) on load (AttentionBlock2 in front of the unit,
                                       //     ref:params/networks/blocks/attentionblock.py:43-47: the gated tensor is never materialised)
  const float *mean, *invstd, *gamma, *scale, *shift, *alpha, *mean_dz, *mean_dzx;
  float inv_keep;
  const char* wpack;                   // packed weights of the data gradient (K = 9 * CY -> N = CX)
  const char* dr;                      // RES: gradient of the residual convolution's output (CY channels); == da for RES 1
  const char* wpack_r;                 // RES: packed weights of the residual convolution's data gradient (K = CY -> N = CX, one tap)
  int dr_vox_bytes;
  char* dx;
  float* slab;                         // [gridDim.x][NTH][9 (+1 with RES)][NTP*16][16]
  int y_vox_bytes, da_vox_bytes, x_vox_bytes, dx_vox_bytes;
  int X, Y, Z;
  int lx, nxs, nyb, nzb;
};

typedef __attribute__((address_space(3))) bf16x4 mb_lds_b4;
__device__ __forceinline__ bf16x8 mb_tr(const char* lo, const char* hi) {
  const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((mb_lds_b4*)lo);
  const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((mb_lds_b4*)hi);
  return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

// CY: channels of y / dA / dy (the convolution's outputs); CX: channels of x (its inputs); UNITSPLIT: the waves split the weight gradient's (tap, dy tile)
// units instead of its K-steps (mwgrad.hip)
constexpr int mb_lds_bytes(int CY, int CX, int TZ, int MT, int RES) {
  const int RPM = 16 / TZ, TYB = MT * 4 * RPM, ROWS = TYB + 2, HPLANE = (ROWS * TZ * (CY / 8) * 16 + 255) / 256 * 256;
  return (MB_NR + (RES ? 2 : 0)) * HPLANE + 2 * ((TYB * TZ * (CX / 8) * 16 + 255) / 256 * 256) + ((9 * (CY / 8) + 3) / 4) * (CX / 16) * 1024;
}

// (one workgroup per CU where the LDS footprint allows no second one anyway: the register budget is then 512 per lane instead of 256)
template <int CY, int CX, int TZ, int MT, bool UNITSPLIT, int RES, bool XG = false>
__global__ __launch_bounds__(256, mb_lds_bytes(CY, CX, TZ, MT, RES) > 80 * 1024 ? 1 : 2) void mbwd_kernel(const MbwdK k) {
  constexpr int GH = CY / 8, GP = CX / 8, RSH = TZ * GH, RSP = TZ * GP, RPM = 16 / TZ, TYB = MT * 4 * RPM, ROWS = TYB + 2;
  constexpr int NTH = CY / 16, NTP = CX / 16;
  constexpr int HSLOTS = ROWS * RSH, PSLOTS = TYB * RSP;
  constexpr int HPLANE = (HSLOTS * 16 + 255) / 256 * 256, PPLANE = (PSLOTS * 16 + 255) / 256 * 256;
  constexpr int HINST = (HSLOTS + 255) / 256, PINST = (PSLOTS + 255) / 256;
  constexpr int KS = 2 * MT;            // weight gradient: K-steps (32 voxels) per plane
  constexpr int NTAPS = RES ? 10 : 9;   // tap 9 = the residual convolution (centre voxel of the raw dR planes)
  constexpr int UNITS = NTAPS * NTH;    // (tap, dy tile) pairs
  constexpr int MYU = UNITSPLIT ? (UNITS + 3) / 4 : UNITS;
  constexpr int KSTEP_BYTES_H = 32 * GH * 16, KSTEP_BYTES_P = 32 * GP * 16;
  constexpr int KSD = (9 * GH + 3) / 4;  // data gradient: K-steps (4 groups of 8 dy channels) per voxel
  constexpr int KSR = RES ? (GH + 3) / 4 : 0;  // ... + the residual convolution's K-steps (one tap)
  constexpr int W_BYTES = KSD * NTP * 1024;  // (the residual convolution's KSR * NTP weight fragments live in registers)
  constexpr int MT_BYTES = RPM * RSH * 16;  // LDS bytes between consecutive M-tiles of the data gradient (RPM rows of a ring plane)
  static_assert(UNITSPLIT || MT % 2 == 0, "the K-step split needs a multiple of 4 K-steps per plane");
  static_assert(256 % RSH == 0, "a thread's pieces of a plane must share one channel group");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Hl = smem;
  char* Pl = smem + MB_NR * HPLANE;
  char* Wl = smem + MB_NR * HPLANE + 2 * PPLANE;
  char* Al = Wl + W_BYTES;  // RES: two planes (x, and the one being written) of raw dR in the ring's layout; only the centre rows are read
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, l15 = lane & 15;
  const int X = k.X, Y = k.Y, Z = k.Z;

  int b = vsseg_xcd_contiguous(blockIdx.x, gridDim.x);
  const int zb = b % k.nzb; b /= k.nzb;
  const int yb = b % k.nyb; b /= k.nyb;
  const int xs = b % k.nxs; const int n = b / k.nxs;
  const int y0 = yb * TYB, z0 = zb * TZ, xb = xs * k.lx, steps = min(k.lx, X - xb);

  for (int i = tid; i < (MB_NR * HPLANE + 2 * PPLANE) / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0u, 0u, 0u, 0u);  // dy rows outside the image stay zero
  for (int i = tid; i < W_BYTES / 16; i += 256) reinterpret_cast<uint4*>(Wl)[i] = reinterpret_cast<const uint4*>(k.wpack)[i];
  bf16x8 wres[RES ? KSR : 1][RES ? NTP : 1];
  if constexpr (RES != 0) {
    for (int i = tid; i < 2 * HPLANE / 16; i += 256) reinterpret_cast<uint4*>(Al)[i] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int ks = 0; ks < KSR; ++ks)
#pragma unroll
      for (int t = 0; t < NTP; ++t) wres[ks][t] = *reinterpret_cast<const bf16x8*>(k.wpack_r + ((ks * NTP + t) * 64 + lane) * 16);
  }

  // ---- this thread's pieces of a dy plane: LDS slot j = (u*4 + wave)*64 + lane holds (row j / RSH, piece' (j % RSH) / TZ, z j % TZ) — mconv.hip's layout.
  //      256 % RSH == 0 and the swizzle term advances by a multiple of GH per u, so ALL pieces of a thread carry the same 8-channel group `pc`:
  //      its BatchNorm constants live in registers.
  int vrel[HINST];
  unsigned hok = 0;
  int pc = 0;
#pragma unroll
  for (int u = 0; u < HINST; ++u) {
    const int j = (u * 4 + wave) * 64 + lane;
    const int r = j / RSH, within = j % RSH, pp = within / TZ, z = within % TZ;
    const int gy = y0 + r - 1;
    const bool ok = j < HSLOTS && (unsigned)gy < (unsigned)Y;
    if (u == 0) pc = (pp - 2 * (r * RSH / 16)) & (GH - 1);
    vrel[u] = ok ? (r - 1) * Z + z : 0;
    if (ok) hok |= 1u << u;
  }
  int prel[PINST], pgrel[XG ? PINST : 1];
  unsigned pok = 0, p1m = 0;
#pragma unroll
  for (int u = 0; u < PINST; ++u) {
    const int j = (u * 4 + wave) * 64 + lane;
    const int r = j / RSP, within = j % RSP, pp = within / TZ, z = within % TZ;
    const int pcx = (pp - 2 * (r * RSP / 16)) & (GP - 1);
    const bool ok = j < PSLOTS;
    prel[u] = ok ? (r * Z + z) * k.x_vox_bytes + pcx * 16 : 0;
    if constexpr (XG) pgrel[u] = ok ? r * Z + z : 0;
    if (ok) pok |= 1u << u;
    if (ok && pcx >= k.x_csplit_pc) p1m |= 1u << u;
  }
  const float alpha = *k.alpha;
  BnBwdC8 bc;
  bn_bwd_consts(bc, k.mean, k.invstd, k.gamma, k.scale, k.shift, k.mean_dz, k.mean_dzx, pc * 8, k.inv_keep);

  const int64_t col0 = (((int64_t)n * X) * Y + y0) * Z + z0;  // voxel (n, 0, y0, z0)
  const int64_t plane_vox = (int64_t)Y * Z;
  const char* ycol = k.y + col0 * k.y_vox_bytes + pc * 16;
  const char* dcol = k.da + col0 * k.da_vox_bytes + pc * 16;
  const unsigned char* kcol = k.keep ? k.keep + col0 * GH + pc : nullptr;
  const char* rcol = RES == 2 ? k.dr + col0 * k.dr_vox_bytes + pc * 16 : nullptr;
  const char* porg = k.x + col0 * k.x_vox_bytes;
  const char* porg1 = k.x1 + col0 * k.x_vox_bytes;
  const float* pgcol = XG ? k.x_gate + col0 : nullptr;
  float pg[XG ? PINST : 1];

  // raw pieces of one dy plane in flight (registers): y, dA (16 bytes each) and the keep-mask byte
  uint4 ry[HINST], rd[HINST], rr2[RES == 2 ? HINST : 1];
  unsigned rk[HINST];
  auto load_h = [&](int i) {  // plane i (x = xb - 1 + i); planes outside the image are zero and are not loaded
    const int x = xb - 1 + i;
    if ((unsigned)x >= (unsigned)X) return;
    const int64_t pv = (int64_t)x * plane_vox;
#pragma unroll
    for (int u = 0; u < HINST; ++u) {  // unconditional loads from clamped addresses (a branch around a load makes hipcc wait for it: DESIGN §3.3)
      const int64_t v = pv + vrel[u];
      ry[u] = *reinterpret_cast<const uint4*>(ycol + v * k.y_vox_bytes);
      rd[u] = *reinterpret_cast<const uint4*>(dcol + v * k.da_vox_bytes);
      rk[u] = kcol ? (unsigned)kcol[v * GH] : 0xffu;
      if constexpr (RES == 2) rr2[u] = *reinterpret_cast<const uint4*>(rcol + v * k.dr_vox_bytes);
    }
  };
  auto store_h = [&](int i) {  // dy = BatchNorm/dropout/PReLU backward of the pieces this thread loaded -> ring slot i & 3
    const int x = xb - 1 + i;
    const bool inside = (unsigned)x < (unsigned)X;
    char* dst = Hl + (i & (MB_NR - 1)) * HPLANE + lane * 16;
#pragma unroll
    for (int u = 0; u < HINST; ++u) {
      if (!((hok >> u) & 1u)) continue;
      uint4 o = make_uint4(0u, 0u, 0u, 0u);
      if (inside) {
        f8 dy;
        bn_bwd_dy8(bf16x8_to_f8(ry[u]), bf16x8_to_f8(rd[u]), rk[u], alpha, bc, dy);
        o = f8_to_bf16x8(dy);
      }
      *reinterpret_cast<uint4*>(dst + (u * 4 + wave) * 1024) = o;
    }
  };
  auto store_a = [&](int i) {  // RES: the raw dR pieces of plane i (still in registers) -> buffer i & 1.  Called BEHIND the step's barrier: the buffer held plane i-2,
    if constexpr (RES != 0) {    // read last in step i-2... by waves that have all passed that barrier
      const bool inside = (unsigned)(xb - 1 + i) < (unsigned)X;
      char* adst = Al + (i & 1) * HPLANE + lane * 16;
#pragma unroll
      for (int u = 0; u < HINST; ++u) {
        if (!((hok >> u) & 1u)) continue;
        uint4 ro = make_uint4(0u, 0u, 0u, 0u);
        if (inside) ro = RES == 1 ? rd[u] : rr2[u];
        *reinterpret_cast<uint4*>(adst + (u * 4 + wave) * 1024) = ro;
      }
    }
  };
  auto issue_p = [&](int i) {  // plane i of x (always inside the image) into buffer i & 1, by LDS-DMA (XG: its gate values by ordinary loads, issued in front)
    const int64_t pv = (int64_t)(xb - 1 + i) * plane_vox;
    if constexpr (XG) {
#pragma unroll
      for (int u = 0; u < PINST; ++u) pg[u] = pgcol[pv + pgrel[u]];
    }
    const char* q = porg + pv * k.x_vox_bytes;
    const char* q1 = porg1 + pv * k.x_vox_bytes;
    char* dst = Pl + (i & 1) * PPLANE;
#pragma unroll
    for (int u = 0; u < PINST; ++u)
      if ((pok >> u) & 1u) vsseg_dma16(((p1m >> u) & 1u ? q1 : q) + prel[u], dst + (u * 4 + wave) * 1024);
  };
  auto gate_p = [&](int i) {  // XG: every thread gates the pieces of x plane i IT fetched, in LDS, in front of the step's barrier (fp32 product rounded to bf16:
    if constexpr (XG) {          // bit-identical to vsseg_att_apply_fwd)
      char* dst = Pl + (i & 1) * PPLANE + lane * 16;
#pragma unroll
      for (int u = 0; u < PINST; ++u) {
        if (!((pok >> u) & 1u)) continue;
        uint4* p = reinterpret_cast<uint4*>(dst + (u * 4 + wave) * 1024);
        const uint4 qv = *p;
        const float gg = 1.f + pg[u];
        uint4 o;
        o.x = f2bf2(vsseg_mul_unpacked(__uint_as_float(qv.x << 16), gg), vsseg_mul_unpacked(__uint_as_float(qv.x & 0xffff0000u), gg));
        o.y = f2bf2(vsseg_mul_unpacked(__uint_as_float(qv.y << 16), gg), vsseg_mul_unpacked(__uint_as_float(qv.y & 0xffff0000u), gg));
        o.z = f2bf2(vsseg_mul_unpacked(__uint_as_float(qv.z << 16), gg), vsseg_mul_unpacked(__uint_as_float(qv.z & 0xffff0000u), gg));
        o.w = f2bf2(vsseg_mul_unpacked(__uint_as_float(qv.w << 16), gg), vsseg_mul_unpacked(__uint_as_float(qv.w & 0xffff0000u), gg));
        *p = o;
      }
    }
  };

  // ---- weight gradient: transpose-read addressing (mwgrad.hip).  Lane (g, i = l15): voxel r4 = i >> 2 of a 4-voxel block, 4-channel chunk q = i & 3.
  const int r4 = l15 >> 2, q4 = l15 & 3;
  auto h_off = [&](int v, int dy, int th) {
    const int row = v / TZ + 1 + dy, z = v % TZ, pcc = 2 * th + (q4 >> 1);
    return (row * RSH + ((pcc + 2 * (row * RSH / 16)) & (GH - 1)) * TZ + z) * 16 + (q4 & 1) * 8;
  };
  auto p_off = [&](int v, int tp) {
    const int row = v / TZ, z = v % TZ, pcc = 2 * tp + (q4 >> 1);
    return (row * RSP + ((pcc + 2 * (row * RSP / 16)) & (GP - 1)) * TZ + z) * 16 + (q4 & 1) * 8;
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

  // ---- data gradient: MFMA operand addressing (mconv.hip).  K-group p = ks*4 + g -> (tap p / GH, piece p % GH); lane column l15 -> voxel (row l15 / TZ, z l15 % TZ)
  const int rr = l15 / TZ, zz = l15 % TZ;
  int koff[KSD], dxk[KSD];
#pragma unroll
  for (int ks = 0; ks < KSD; ++ks) {
    int p = ks * 4 + g;
    if (p >= 9 * GH) p -= 9 * GH;  // padded K-groups: zero weights times a genuine tap of the same voxel
    const int tap = p / GH, pcc = p % GH, dy = tap % 3 - 1;
    const int row = 1 + rr + dy;
    koff[ks] = (row * RSH + ((pcc + 2 * (row * RSH / 16)) & (GH - 1)) * TZ + zz) * 16 + wave * (MT * MT_BYTES);
    dxk[ks] = tap / 3;
  }
  int koffr[RES ? KSR : 1];
#pragma unroll
  for (int ks = 0; ks < KSR; ++ks) {  // residual K-groups: piece ks*4 + g of the centre voxel (groups beyond GH: zero weights times a genuine piece)
    const int pcc = (ks * 4 + g) & (GH - 1), row = 1 + rr;
    koffr[ks] = (row * RSH + ((pcc + 2 * (row * RSH / 16)) & (GH - 1)) * TZ + zz) * 16 + wave * (MT * MT_BYTES);
  }
  const char* Wlane = Wl + lane * 16;
  const int64_t ocol = col0 + (int64_t)((wave * MT) * RPM + rr) * Z + zz;

  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();  // the buffers are zeroed (and the weights staged) before anything else writes / reads them
  load_h(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  store_h(0);
  load_h(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  store_h(1);
  store_a(1);
  load_h(2);
  issue_p(1);

  for (int i = 1; i <= steps; ++i) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the raw pieces of plane i+1 and this wave's DMA pieces of x plane i have landed (and the previous step's stores have left)
    store_h(i + 1);                                    // slot (i+1) & 3 held plane i-3: every wave finished reading it before the barrier of step i-1
    gate_p(i);                                         // (XG) this thread's pieces of x plane i, gate values loaded with the plane's DMAs
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // dy plane i+1 and x plane i are complete for every wave; every wave has finished step i-1
    if (i + 1 <= steps) store_a(i + 1);                // (before the registers are reloaded)
    if (i + 2 <= steps + 1) load_h(i + 2);
    if (i + 1 <= steps) issue_p(i + 1);
    const char* Ps = Pl + (i & 1) * PPLANE;
    const int sm1 = ((i - 1) & (MB_NR - 1)) * HPLANE, s0 = (i & (MB_NR - 1)) * HPLANE, sp1 = ((i + 1) & (MB_NR - 1)) * HPLANE;
    const char* Hs[3] = {Hl + sm1, Hl + s0, Hl + sp1};
    const char* As = Al + (i & 1) * HPLANE;

    // ---- weight gradient of plane i
#pragma unroll
    for (int kk = 0; kk < (UNITSPLIT ? KS : KS / 4); ++kk) {
      const int ks = UNITSPLIT ? kk : kk * 4 + wave;
      bf16x8 pa[NTP];
#pragma unroll
      for (int tp = 0; tp < NTP; ++tp) pa[tp] = mb_tr(Ps + plo[tp] + ks * KSTEP_BYTES_P, Ps + phi[tp] + ks * KSTEP_BYTES_P);
#pragma unroll
      for (int u = 0; u < MYU; ++u) {
        const int unit = UNITSPLIT ? u * 4 + wave : u;  // (tap, dy tile) = (unit / NTH, unit % NTH)
        if (UNITSPLIT && unit >= UNITS) break;
        const int tap = unit / NTH, th = unit % NTH, dx = tap == 9 ? 3 : tap / 3, dy = tap == 9 ? 1 : tap % 3;  // tap 9: the centre voxel of the raw dR plane
        int olo, ohi;
        if constexpr (UNITSPLIT) {
          olo = hlo[0][0]; ohi = hhi[0][0];
#pragma unroll
          for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int t2 = 0; t2 < NTH; ++t2)
              if (d == dy && t2 == th) { olo = hlo[d][t2]; ohi = hhi[d][t2]; }
        } else {
          olo = hlo[dy][th]; ohi = hhi[dy][th];
        }
        const char* hb = (dx == 0 ? Hs[0] : (dx == 1 ? Hs[1] : (dx == 2 ? Hs[2] : As))) + ks * KSTEP_BYTES_H;
        const bf16x8 hv = mb_tr(hb + olo, hb + ohi);
#pragma unroll
        for (int tp = 0; tp < NTP; ++tp) acc[u][tp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[tp], hv, acc[u][tp], 0, 0, 0);
      }
    }

    // ---- data gradient of plane i: dX = sum over (tap, dy channel group) W' * dy, rows [wave*MT*RPM, (wave+1)*MT*RPM) of the column
    f32x4 dacc[MT][NTP];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < NTP; ++t) dacc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KSD; ++ks) {
      bf16x8 w[NTP];
#pragma unroll
      for (int t = 0; t < NTP; ++t) w[t] = *reinterpret_cast<const bf16x8*>(Wlane + (ks * NTP + t) * 1024);
      const char* hb = Hl + koff[ks] + (dxk[ks] == 0 ? sm1 : (dxk[ks] == 1 ? s0 : sp1));
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const bf16x8 av = *reinterpret_cast<const bf16x8*>(hb + m * MT_BYTES);
#pragma unroll
        for (int t = 0; t < NTP; ++t) dacc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[t], av, dacc[m][t], 0, 0, 0);
      }
    }
    if constexpr (RES != 0) {
#pragma unroll
      for (int ks = 0; ks < KSR; ++ks) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const bf16x8 av = *reinterpret_cast<const bf16x8*>(As + koffr[ks] + m * MT_BYTES);
#pragma unroll
          for (int t = 0; t < NTP; ++t) dacc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wres[ks][t], av, dacc[m][t], 0, 0, 0);
        }
      }
    }
    const int64_t ovox0 = ocol + (int64_t)(xb - 1 + i) * plane_vox;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      char* op = k.dx + (ovox0 + (int64_t)m * RPM * Z) * k.dx_vox_bytes + g * 8;
#pragma unroll
      for (int t = 0; t < NTP; ++t) st4(reinterpret_cast<bf16_t*>(op + t * 32), make_float4(dacc[m][t][0], dacc[m][t][1], dacc[m][t][2], dacc[m][t][3]));
    }
  }

  // ---- flush of the weight gradient (mwgrad.hip).  Lane holds rows g*4 + r (x channel) x column l15 (dy channel) of every owned (tap, dy tile, x tile).
  __syncthreads();  // the ring is free: it becomes the cross-wave reduction buffer
  float* red = reinterpret_cast<float*>(smem);
  float* slab = k.slab + (int64_t)blockIdx.x * (NTH * NTAPS * NTP * 256);
  if constexpr (UNITSPLIT) {
#pragma unroll
    for (int u = 0; u < MYU; ++u) {
      const int unit = u * 4 + wave;
      if (unit >= UNITS) break;
      const int tap = unit / NTH, th = unit % NTH;
#pragma unroll
      for (int tp = 0; tp < NTP; ++tp)
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[((int64_t)(th * NTAPS + tap) * (NTP * 16) + tp * 16 + g * 4 + r) * 16 + l15] = acc[u][tp][r];
    }
  } else {  // the four waves hold partial sums over their K-steps: added in wave order through LDS (run-to-run bit-identical)
    constexpr int NACC = UNITS * NTP * 256;
    static_assert(NACC * 4 <= MB_NR * HPLANE + 2 * PPLANE + W_BYTES + (RES ? 2 * HPLANE : 0), "reduction buffer does not fit the workgroup's LDS");
    for (int w = 0; w < 4; ++w) {
      if (wave == w) {
#pragma unroll
        for (int u = 0; u < MYU; ++u) {
          const int tap = u / NTH, th = u % NTH;
#pragma unroll
          for (int tp = 0; tp < NTP; ++tp)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float* dst = red + ((th * NTAPS + tap) * (NTP * 16) + tp * 16 + g * 4 + r) * 16 + l15;
              if (w == 0) *dst = acc[u][tp][r];
              else if (w < 3) *dst += acc[u][tp][r];
              else slab[dst - red] = *dst + acc[u][tp][r];
            }
        }
      }
      __syncthreads();
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------
template <int CY, int CX, int TZ, int MT, int RES> static int mb_lds() { return mb_lds_bytes(CY, CX, TZ, MT, RES); }
template <int CY, int CX, int TZ, int MT, bool US, int RES> static int mb_launch_r(const MbwdK& k, int grid, hipStream_t s) {
  static bool init_dev[16] = {}; bool& init = vsseg_dev_once(init_dev);  // per device: the LDS opt-in is a per-device function attribute
  const int lds = mb_lds<CY, CX, TZ, MT, RES>();
  if (lds > 160 * 1024) { vsseg_set_error("vsseg_conv_bwd_fused: %d bytes of LDS", lds); return VSSEG_EINVAL; }
  if (!init) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&mbwd_kernel<CY, CX, TZ, MT, US, RES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    init = true;
  }
  hipLaunchKernelGGL((mbwd_kernel<CY, CX, TZ, MT, US, RES>), dim3((unsigned)grid), dim3(256), lds, s, k);
  VSSEG_LAUNCH_CHECK("vsseg_conv_bwd_fused");
  return VSSEG_OK;
}
template <int CY, int CX, int TZ, int MT, bool US> static int mb_launch_xg(const MbwdK& k, int grid, hipStream_t s) {  // x gated on load: the level-1 decoder unit (64 -> 32 + residual)
  static bool init_dev[16] = {}; bool& init = vsseg_dev_once(init_dev);  // per device: the LDS opt-in is a per-device function attribute
  const int lds = mb_lds<CY, CX, TZ, MT, 1>();
  if (lds > 160 * 1024) { vsseg_set_error("vsseg_conv_bwd_fused: %d bytes of LDS", lds); return VSSEG_EINVAL; }
  if (!init) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&mbwd_kernel<CY, CX, TZ, MT, US, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    init = true;
  }
  hipLaunchKernelGGL((mbwd_kernel<CY, CX, TZ, MT, US, 1, true>), dim3((unsigned)grid), dim3(256), lds, s, k);
  VSSEG_LAUNCH_CHECK("vsseg_conv_bwd_fused");
  return VSSEG_OK;
}
template <int CY, int CX, int TZ, int MT, bool US> static int mb_launch(const MbwdK& k, int grid, hipStream_t s) {
  if (k.x_gate) {
    if constexpr (CX == 64) {
      if (k.dr == k.da && k.dr_vox_bytes == k.da_vox_bytes) return mb_launch_xg<CY, CX, TZ, MT, US>(k, grid, s);
    }
    vsseg_set_error("vsseg_conv_bwd_fused: the gated input (x_gate) is instantiated for 64 input channels with the residual convolution on dout only");
    return VSSEG_EINVAL;
  }
  if (!k.dr) return mb_launch_r<CY, CX, TZ, MT, US, 0>(k, grid, s);
  if constexpr (CX > CY || CY == 32) {  // the units with a 1x1x1 residual convolution: 16 -> 32 (encoder, level 1) and 64 -> 32 (decoder, level 1)
    if (k.dr == k.da && k.dr_vox_bytes == k.da_vox_bytes) return mb_launch_r<CY, CX, TZ, MT, US, 1>(k, grid, s);
    return mb_launch_r<CY, CX, TZ, MT, US, 2>(k, grid, s);
  } else {
    vsseg_set_error("vsseg_conv_bwd_fused: no instantiation with a residual convolution for %d -> %d channels", CX, CY);
    return VSSEG_EINVAL;
  }
}

typedef int (*mb_fn_t)(const MbwdK&, int, hipStream_t);
struct MbEntry { int cy, cx, tz, mt; mb_fn_t fn; };
#define MB_E(Y, X, Z, M, US) {Y, X, Z, M, mb_launch<Y, X, Z, M, US>}
// (output channels = channels of y / dA, input channels = channels of x, TZ, M-tiles per wave) — rows per workgroup TYB = 64 * MT / TZ
static const MbEntry mb_table[] = {
    MB_E(16, 16, 8, 4, false), MB_E(16, 16, 4, 4, false), MB_E(16, 16, 4, 2, false), MB_E(16, 16, 8, 8, false),  // 16 -> 16 (level 0)
    MB_E(32, 16, 4, 4, false), MB_E(32, 16, 4, 2, false), MB_E(32, 16, 8, 4, false),                             // 16 -> 32 (level 1)
    MB_E(32, 32, 4, 4, true), MB_E(32, 32, 4, 2, true), MB_E(32, 32, 2, 2, true),                                // 32 -> 32
    MB_E(32, 64, 4, 2, true), MB_E(32, 64, 2, 2, true), MB_E(32, 64, 2, 1, true)};                               // 64 -> 32 (level 1 decoder unit)

static const MbEntry* mb_find(const vsseg_conv_bwd_desc* d, const char** why) {
  *why = nullptr;
  auto no = [&](const char* w) { *why = w; return (const MbEntry*)nullptr; };
  const vsseg_tensor* ts[] = {&d->y, &d->dout, &d->x, &d->dx, &d->dres};
  for (const vsseg_tensor* t : ts) {
    if (t == &d->dres && !t->ptr) continue;
    if (!t->ptr || t->dtype != VSSEG_BF16 || (t->ptr2 && t != &d->x)) return no("tensors must be one-part bf16 (x may be a two-part tensor)");
    if (t->ptr2 && (t->csplit <= 0 || t->csplit >= t->c || t->csplit % 16 || ((uintptr_t)t->ptr2 & 15))) return no("bad two-part x");
    if (t->pitch % 8 || ((uintptr_t)t->ptr & 15)) return no("tensor rows must be 16-byte aligned");
    if (t->n != d->y.n || t->x != d->y.x || t->y != d->y.y || t->z != d->y.z) return no("tensor extents differ");
  }
  if (d->dout.c != d->y.c || d->dx.c != d->x.c) return no("channel counts of y / dout or x / dx differ");
  if (d->dres.ptr && (d->dres.c != d->y.c || !d->wpack_res || !d->dw_res)) return no("the residual convolution needs dres with the channels of y, its packed weights and its weight-gradient destination");
  const int tz = d->tile[2], tyb = d->tile[1];
  if ((tz != 2 && tz != 4 && tz != 8) || tyb < 1 || (tyb * tz) % 64 || d->tile[0] < 1) return no("tile must be (x steps per workgroup, rows, tz in {2, 4, 8}) with rows * tz a multiple of 64");
  if (d->y.y % tyb || d->y.z % tz) return no("extent is not a multiple of the column block");
  const int mt = tyb * tz / 64;
  for (const MbEntry& e : mb_table)
    if (e.cy == d->y.c && e.cx == d->x.c && e.tz == tz && e.mt == mt) return &e;
  return no("no instantiation for this (output channels, input channels, tz, rows)");
}

// dw[co][ci][mirror(t)] += sum over workgroups of slab[b][th][t][ci][co16]  (t = 9: the residual convolution's dw_res[co][ci]); fixed summation order (vsseg_slab_sum)
__global__ __launch_bounds__(VSSEG_SLAB_THREADS) void mbwd_reduce_kernel(const float* __restrict__ slab, int nblk, int nth, int ntaps, int cx, float* __restrict__ dw, float* __restrict__ dwr) {
  __shared__ float lds[VSSEG_SLAB_THREADS];
  const int64_t total = (int64_t)nth * ntaps * cx * 16;
  const int64_t i = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const float s = vsseg_slab_sum(slab, total, i, nblk, lds);
  if (threadIdx.x >= 64 || i >= total) return;
  const int l15 = (int)(i & 15), ci = (int)((i >> 4) % cx), t = (int)(((i >> 4) / cx) % ntaps), th = (int)((i >> 4) / cx / ntaps);
  const int co = th * 16 + l15;
  if (t < 9) dw[((int64_t)co * cx + ci) * 9 + (8 - t)] += s;
  else dwr[(int64_t)co * cx + ci] += s;
}

extern "C" int vsseg_conv_bwd_fused(const vsseg_conv_bwd_desc* d, void* stream) {
  VSSEG_CHECK(d && d->mean && d->invstd && d->gamma && d->scale && d->shift && d->alpha && d->mean_dz && d->mean_dzx && d->wpack && d->dw && d->scratch, "vsseg_conv_bwd_fused: null pointer");
  VSSEG_CHECK(d->p_drop >= 0.f && d->p_drop < 1.f && (d->p_drop == 0.f || d->keep), "vsseg_conv_bwd_fused: dropout needs the keep-mask bytes of the forward (p_drop %g)", (double)d->p_drop);
  const char* why;
  const MbEntry* e = mb_find(d, &why);
  if (!e) { vsseg_set_error("vsseg_conv_bwd_fused: not applicable: %s", why); return VSSEG_EINVAL; }
  hipStream_t s = as_stream(stream);
  MbwdK k;
  k.y = reinterpret_cast<const char*>(d->y.ptr); k.da = reinterpret_cast<const char*>(d->dout.ptr); k.keep = d->p_drop > 0.f ? d->keep : nullptr;
  k.x = reinterpret_cast<const char*>(d->x.ptr); k.dx = reinterpret_cast<char*>(d->dx.ptr);
  k.x1 = d->x.ptr2 ? reinterpret_cast<const char*>(d->x.ptr2) - (int64_t)d->x.csplit * 2 : k.x;
  k.x_csplit_pc = d->x.ptr2 ? d->x.csplit / 8 : 1 << 20;
  k.x_gate = d->x_gate;
  k.mean = d->mean; k.invstd = d->invstd; k.gamma = d->gamma; k.scale = d->scale; k.shift = d->shift; k.alpha = d->alpha; k.mean_dz = d->mean_dz; k.mean_dzx = d->mean_dzx;
  k.inv_keep = 1.f / (1.f - d->p_drop);
  k.wpack = reinterpret_cast<const char*>(d->wpack);
  k.dr = reinterpret_cast<const char*>(d->dres.ptr); k.dr_vox_bytes = d->dres.pitch * 2; k.wpack_r = reinterpret_cast<const char*>(d->wpack_res);
  k.y_vox_bytes = d->y.pitch * 2; k.da_vox_bytes = d->dout.pitch * 2; k.x_vox_bytes = d->x.pitch * 2; k.dx_vox_bytes = d->dx.pitch * 2;
  k.X = d->y.x; k.Y = d->y.y; k.Z = d->y.z;
  k.lx = d->tile[0] > k.X ? k.X : d->tile[0];
  k.nxs = (k.X + k.lx - 1) / k.lx; k.nyb = k.Y / d->tile[1]; k.nzb = k.Z / d->tile[2];
  const int nth = d->y.c / 16, ntaps = k.dr ? 10 : 9;
  const int64_t per_blk = (int64_t)nth * ntaps * d->x.c * 16;
  const int64_t grid = (int64_t)d->y.n * k.nxs * k.nyb * k.nzb;
  VSSEG_CHECK(grid > 0 && grid < (1ll << 24), "vsseg_conv_bwd_fused: bad grid");
  VSSEG_CHECK(d->scratch_elems >= grid * per_blk, "vsseg_conv_bwd_fused: scratch too small for %lld workgroups (%lld < %lld floats); use longer x segments", (long long)grid, (long long)d->scratch_elems,
              (long long)(grid * per_blk));
  k.slab = d->scratch;
  int rc = e->fn(k, (int)grid, s);
  if (rc) return rc;
  hipLaunchKernelGGL(mbwd_reduce_kernel, dim3((unsigned)((per_blk + 63) / 64)), dim3(VSSEG_SLAB_THREADS), 0, s, (const float*)d->scratch, (int)grid, nth, ntaps, d->x.c, d->dw, d->dw_res);
  VSSEG_LAUNCH_CHECK("vsseg_conv_bwd_fused (reduce)");
  return VSSEG_OK;
}
