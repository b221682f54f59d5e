// Marching streaming convolution (launch plans with depth -5): the HBM-bound stride-1 3x3x1 bf16 convolutions and data gradients of the two
// finest levels of the 2.5D U-Net (ref:params/networks/blocks/convolutions.py:114-146; the 16/32/64-channel layers at 384x128x128 and
// 192x64x128, SURVEY §8a rows 2, 5, 6, 41, 44, 46, 49 and their data gradients) with every input voxel fetched from HBM exactly once.
//
// The tile kernel (sconv.hip) fetches a (8+2)x(8+2)x4 halo per 8x8x4 output tile: 1.56x the input bytes, and the PMC counters show the
// overlap is NOT served by L2 (profiles/r02_pmc_hbm.txt: 1.28x the algorithmic bytes on the 16->16 layers = exactly in*1.56 + out).  Here a
// workgroup owns a column (sample n, rows [y0, y0+TYB), slices [z0, z0+TZ)) and MARCHES along x: a plane = one x position of the column
// ((TYB+2) rows x TZ voxels x CIN channels, <= 17 KB) is DMA'd into a ring of four LDS slots, the 3x3x1 stencil of plane x reads the slots
// of x-1, x, x+1 while the DMA of plane x+2 is in flight (issued one whole step ahead: the load latency is hidden inside the workgroup, not
// only by the neighbours on the CU).  With TYB = Y (the benchmark shapes) there is no halo at all: rows -1 and Y are the convolution's zero
// padding and are never fetched; x segments re-fetch 2 planes per `lx` steps (2-4 %).  A 3x3x1 stencil has no z halo, so TZ is free: it is
// chosen so that a plane row (TZ voxels) is a 128- or 256-byte run in HBM.
//
//   * LDS plane layout [row][piece'][z] (16-byte pieces of 8 channels, piece' = (piece + 2 * (row * RS / 16)) mod G, RS = TZ*G slots per
//     row): the 16-lane groups in which the LDS serves a ds_read_b128 hit 16 different slots of a 256-byte bank row for every tap
//     (tools/lds_conflicts.py checks it exhaustively: 0 conflict cycles for 16/32/64 channels); the permutation is free — the DMA writes LDS in
//     lane order, so it is applied to the GLOBAL address each lane fetches, and each run of RS lanes still reads one contiguous 128/256-byte
//     row segment
//   * same packed weights, K order (tap, 8-channel group), MFMA operand order, output-channel ownership and epilogue semantics as the
//     general kernel and sconv.hip: results agree bit for bit
//   * the auxiliary operand of an accumulating / residual / gated launch is loaded one step ahead into registers, in FRONT of the next
//     plane's DMAs (loads return in order: behind them it would wait for the whole plane)
#include "common.h"
#include "bn_bwd.h"
#include "mconv.h"
#include <type_traits>

constexpr int MC_NR = 4;  // ring slots: planes x-1, x, x+1 + the one in flight

struct MconvK {
  const char* in0;  // channels [0, csplit) ...
  const char* in1;  // ... and [csplit, c), biased by -csplit channels (== in0 for an ordinary tensor)
  char* out0; char* out1;
  const char* aux0; const char* aux1;
  const float* gate;
  const float* in_gate;  // GIN: fp32 attention map of the INPUT tensor: voxel v is multiplied by (1 + in_gate[v]) on load
  const bf16_t* x1c;     // aux_mode 5: one-channel tensor [N][X][Y][Z]; out += x1c[voxel] * in1_w[c] + in1_b[c] behind the activation (the first ResidualUnit's 1x1x1
  const float *in1_w, *in1_b;  // residual convolution of the ONE-channel network input, ref:params/networks/blocks/convolutions.py:241-255 with in_channels = 1)
  const char* wpack;
  const float *bias, *bias2, *scale, *shift, *alpha;
  double* stats;
  unsigned* fxflag;  // sticky range / non-finite flag of the fixed-point statistics (common.h)
  const void* zeros;
  int in_csplit_pc;  // first 16-byte piece of a voxel row that lives in part 1
  int in_vox_bytes, out_vox_bytes, aux_vox_bytes;
  int out_csplit, aux_csplit;
  int out_f32, aux_mode, act, cout, cout_mod, stats_stride;
  int X, Y, Z;
  int lx, nxs, nyb, nzb;  // x steps per workgroup; segments in x, blocks in y and z
  int ps;                 // host side only: a pixel-shuffle launch (template PS)
  // NR > 0: NR more 16-channel tiles of a 1x1x1 convolution of the SAME input ride along (the ResidualUnit's residual convolution,
  // ref:params/networks/blocks/convolutions.py:241-255): only the K-steps of the centre tap, weights in registers, the operand reads shared with the main tiles
  const char* wpack_r;   // [KHI - KLO][NR][64 lanes][8]
  const float* bias_r;
  char* res_out;         // its output tensor (bf16), or nullptr: added to the main tiles behind their activation (eval: out = act(bn(conv(x))) + residual(x))
  int res_vox_bytes, res_cout;
};

// MODE: 0 plain, 1 + BatchNorm statistics, 2 + auxiliary operand (bf16), 3 plain + attention gate applied to the input on load (GIN):
// AttentionBlock2's `att.repeat(C) * x + x` (ref:params/networks/blocks/attentionblock.py:43-47) is a per-voxel scalar on the convolution's input.
// A thread multiplies the pieces IT fetched by (1 + att[voxel]) in LDS (fp32 product rounded to bf16: bit-identical to vsseg_att_apply_fwd) right
// after its own DMA wait and in front of the step's barrier — no extra barrier, and the gated tensor is never written to HBM or read back.
// WREG (launch plans with depth -6): the packed weights live in REGISTERS (KSTEPS x NT fragments of 4 VGPRs, loaded once per workgroup) instead of in LDS.
// With few M-tiles per wave the K loop re-reads every weight fragment from LDS at every x step — on the 64 -> 32 layers (MT = 1, 36 fragments per
// step and wave against 18 operand fragments) two thirds of the LDS bandwidth the launch is bound by; in registers they cost nothing per step and
// the 9-36 KB of LDS they occupied go back to the ring.
// CC = 1 / 2 (CIN 8 only): the input is a COMPACT one- / two-channel tensor [N][X][Y][Z][CC] (2 / 4 bytes per voxel: the network input, the pre-sigmoid gradient of an
// attention map; the gradient of the two logits) instead of its zero-extension to one 8-channel group (16 bytes per voxel, most of them zeros read from HBM): the thread
// that owns a plane slot loads the voxel's value(s) one step ahead (ordinary load into a register) and writes the zero-extended 16-byte piece into the ring itself.
// PS = TPC > 0 (NT = 4 * TPC): ALL four output-parity classes of a stride-(2,2,1) 3x3x1 transposed convolution as ONE marching launch on the COARSE lattice ("pixel
// shuffle", as sconv.hip's TAPS 4): output channel tile t is tile t % TPC of class (px, py) = ((t / TPC) >> 1, (t / TPC) & 1), stored at fine voxel (2x + px, 2y + py, z).
// The K loop runs the 4 taps of the 2x2x1 neighbourhood (+0 / +1: the lower-right corner of this kernel's 3x3 stencil, the other five taps are never multiplied), K-group
// p = ks*4 + g -> (tap p / G = dx*2 + dy, piece p % G); packed weights [(4G + 3) / 4][NT][64][8].  Every fine output row is written as TZ consecutive voxels.
// G need not be a power of two (48 input channels: mc_mod).
template <int G> __device__ __forceinline__ int mc_mod(int v) {  // v mod G, v may be negative
  if constexpr ((G & (G - 1)) == 0) return v & (G - 1);
  else return ((v % G) + G) % G;
}
template <int CIN, int NT, int TZ, int MT, int MODE, bool WREG, int NR = 0, int CC = 0, int TPC = 0>
__global__ __launch_bounds__(256, 2) void mconv_kernel(const MconvK k) {
  constexpr bool PS = TPC > 0;
  static_assert(CC == 0 || CIN == 8, "compact inputs are one zero-extended channel group");
  static_assert(!PS || (NT == 4 * TPC && NR == 0 && CC == 0 && (MODE == 0 || MODE == 1 || MODE == 2)), "pixel-shuffle launches: four classes of TPC tiles; plain / statistics / accumulating epilogue");
  constexpr bool C1 = CC != 0;
  constexpr bool STATS = MODE == 1 || MODE == 4, AUXM = MODE == 2, GIN = MODE == 3 || MODE == 4;  // (4: statistics + input gate: the level-1 decoder unit's first convolution)
  constexpr int KLO = CIN / 8, KHI = (5 * (CIN / 8) + 3) / 4, KR = NR ? KHI - KLO : 0;  // K-steps that hold the centre tap's channel groups [4G, 5G)
  constexpr int G = CIN / 8, CINB = CIN * 2, RS = TZ * G, RPM = 16 / TZ, TYB = MT * 4 * RPM, ROWS = TYB + 2;
  constexpr int PLANE_SLOTS = ROWS * RS, PLANE_BYTES = (PLANE_SLOTS * 16 + 255) / 256 * 256, NINST = (PLANE_SLOTS + 255) / 256;  // ring slots start on a 256-byte bank row
  constexpr int KSTEPS = PS ? (4 * G + 3) / 4 : (9 * G + 3) / 4, KSW = KSTEPS, W_BYTES = WREG ? 0 : KSW * NT * 1024;
  // the 64-channel launches (18 K-steps per plane, 36+ MFMAs per wave): the next plane's LDS-DMA instructions are issued one at a time BETWEEN the K-steps — an LDS-DMA holds its
  // wave ~85 cycles of a per-CU serial resource, and all of a wave's pieces in front of its K loop kept the matrix pipe idle that long: 64 -> 32 at 192x64x128 x 4
  // 0.33 -> 0.29 ms.  The shorter K loops (16 / 32 channels: HBM-bound, a plane's MFMAs do not cover the issue) measured 3-15 % SLOWER that way (tools/bench_mconv.py)
  constexpr bool SPREAD = KSTEPS >= 18 && CC == 0;
  constexpr int MT_BYTES = RPM * RS * 16;  // LDS bytes between consecutive M-tiles (RPM rows)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Wl = smem;
  char* Rl = smem + W_BYTES;
  float* epi = reinterpret_cast<float*>(smem + W_BYTES + MC_NR * PLANE_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, l15 = lane & 15;
  const int X = k.X, Y = k.Y, Z = k.Z, cout = k.cout;

  // ---- workgroup -> column segment
  int b = vsseg_xcd_contiguous(blockIdx.x, gridDim.x);
  const int zb = b % k.nzb; b /= k.nzb;
  const int yb = b % k.nyb; b /= k.nyb;
  const int xs = b % k.nxs; const int n = b / k.nxs;
  const int y0 = yb * TYB, z0 = zb * TZ, xb = xs * k.lx, steps = min(k.lx, X - xb);

  if constexpr (!WREG)
    for (int i = tid; i < W_BYTES / 16; i += 256) reinterpret_cast<uint4*>(Wl)[i] = reinterpret_cast<const uint4*>(k.wpack)[i];
  bf16x8 wreg[WREG ? KSW : 1][WREG ? NT : 1];
  if constexpr (WREG) {
#pragma unroll
    for (int ks = 0; ks < KSW; ++ks)
#pragma unroll
      for (int t = 0; t < NT; ++t) wreg[ks][t] = *reinterpret_cast<const bf16x8*>(k.wpack + ((ks * NT + t) * 64 + lane) * 16);
  }
  for (int i = tid; i < MC_NR * PLANE_BYTES / 16; i += 256) reinterpret_cast<uint4*>(Rl)[i] = make_uint4(0u, 0u, 0u, 0u);  // rows outside the image stay zero: they are never fetched
  for (int i = tid; i < NT * 16; i += 256) {
    const bool ok = PS || i < cout;  // (PS: every class tile carries the real channels 0 .. cout_mod - 1)
    const int cv = k.cout_mod > 0 ? i % k.cout_mod : i;
    epi[i] = ((ok && k.bias) ? k.bias[cv] : 0.f) + ((ok && k.bias2) ? k.bias2[cv] : 0.f);
    epi[NT * 16 + i] = (ok && k.scale) ? k.scale[cv] : 1.f;
    epi[2 * NT * 16 + i] = (ok && k.scale) ? k.shift[cv] : 0.f;
    epi[3 * NT * 16 + i] = (ok && k.in1_w) ? k.in1_w[cv] : 0.f;
    epi[4 * NT * 16 + i] = (ok && k.in1_b) ? k.in1_b[cv] : 0.f;
  }
  const float alpha = (k.act == VSSEG_ACT_PRELU && k.alpha) ? *k.alpha : 0.f;
  bf16x8 wres[NR ? KR : 1][NR ? NR : 1];
  float rb[NR ? NR : 1][4];
  if constexpr (NR > 0) {
#pragma unroll
    for (int ks = 0; ks < KR; ++ks)
#pragma unroll
      for (int t = 0; t < NR; ++t) wres[ks][t] = *reinterpret_cast<const bf16x8*>(k.wpack_r + ((ks * NR + t) * 64 + (threadIdx.x & 63)) * 16);
#pragma unroll
    for (int t = 0; t < NR; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = t * 16 + ((threadIdx.x & 63) >> 4) * 4 + r;
        rb[t][r] = (k.bias_r && c < k.res_cout) ? k.bias_r[c] : 0.f;
      }
  }

  // ---- this thread's DMA pieces: LDS slot j = (u*4 + wave)*64 + lane of a plane holds (row j / RS, piece' (j % RS) / TZ, z j % TZ)
  int rel[NINST], grel[GIN ? NINST : 1];
  unsigned okmask = 0, p1mask = 0;
#pragma unroll
  for (int u = 0; u < NINST; ++u) {
    const int j = (u * 4 + wave) * 64 + lane;
    const int r = j / RS, within = j % RS, pp = within / TZ, z = within % TZ;
    const int pc = mc_mod<G>(pp - 2 * (r * RS / 16));
    const int gy = y0 + r - 1;
    const bool ok = j < PLANE_SLOTS && (unsigned)gy < (unsigned)Y;
    rel[u] = ok ? ((r - 1) * Z + z) * k.in_vox_bytes + (C1 ? 0 : pc * 16) : 0;
    if constexpr (GIN) grel[u] = ok ? (r - 1) * Z + z : 0;
    if (ok) okmask |= 1u << u;
    if (ok && pc >= k.in_csplit_pc) p1mask |= 1u << u;
  }
  const int64_t plane_stride = (int64_t)Y * Z * k.in_vox_bytes;
  const int64_t col0 = (((int64_t)n * X) * Y + y0) * Z + z0;  // voxel (n, 0, y0, z0)
  const char* org0 = k.in0 + col0 * k.in_vox_bytes;
  const char* org1 = k.in1 + col0 * k.in_vox_bytes;
  auto issue_piece = [&](int i, int u) __attribute__((always_inline)) {
    const int x = xb - 1 + i;
    char* dst = Rl + (i & (MC_NR - 1)) * PLANE_BYTES;
    const bool inside = (unsigned)x < (unsigned)X;
    const char* p0 = org0 + (int64_t)x * plane_stride;
    const char* p1 = org1 + (int64_t)x * plane_stride;
    if ((okmask >> u) & 1u) vsseg_dma16(inside ? (const void*)(((p1mask >> u) & 1u ? p1 : p0) + rel[u]) : k.zeros, dst + (u * 4 + wave) * 1024);
  };
  auto issue = [&](int i) {  // DMA plane i (x = xb - 1 + i) into ring slot i & 3; planes outside the image are zero
    const int x = xb - 1 + i;
    char* dst = Rl + (i & (MC_NR - 1)) * PLANE_BYTES;
    const bool inside = (unsigned)x < (unsigned)X;
    const char* p0 = org0 + (int64_t)x * plane_stride;
    const char* p1 = org1 + (int64_t)x * plane_stride;
#pragma unroll
    for (int u = 0; u < NINST; ++u)
      if ((okmask >> u) & 1u) vsseg_dma16(inside ? (const void*)(((p1mask >> u) & 1u ? p1 : p0) + rel[u]) : k.zeros, dst + (u * 4 + wave) * 1024);
  };

  // C1: the one-channel values of this thread's slots of plane i -> registers; -> zero-extended 16-byte pieces in ring slot i & 3 (planes / rows outside the image: zero)
  auto loadc = [&](int i, unsigned (&cv)[C1 ? NINST : 1]) {
    if constexpr (C1) {
      const int x = xb - 1 + i;
      const bool inside = (unsigned)x < (unsigned)X;
      const char* p0 = org0 + (int64_t)(inside ? x : 0) * plane_stride;
#pragma unroll
      for (int u = 0; u < NINST; ++u) {  // unconditional loads from valid addresses, the zero padding is a select on the loaded value
        unsigned v;
        if constexpr (CC == 2) v = *reinterpret_cast<const unsigned*>(p0 + rel[u]);
        else v = *reinterpret_cast<const unsigned short*>(p0 + rel[u]);
        cv[u] = (inside && ((okmask >> u) & 1u)) ? v : 0u;
      }
    }
  };
  auto storec = [&](int i, const unsigned (&cv)[C1 ? NINST : 1]) {
    if constexpr (C1) {
      char* dst = Rl + (i & (MC_NR - 1)) * PLANE_BYTES + lane * 16;
#pragma unroll
      for (int u = 0; u < NINST; ++u)
        if ((u * 4 + wave) * 64 + lane < PLANE_SLOTS) *reinterpret_cast<uint4*>(dst + (u * 4 + wave) * 1024) = make_uint4(cv[u], 0u, 0u, 0u);
    }
  };

  // GIN: the gate values of this thread's pieces of plane i (ordinary loads, issued in FRONT of the plane's DMAs), and the in-place product
  const float* gcol = GIN ? k.in_gate + col0 : nullptr;
  auto load_gate = [&](int i, float (&gv)[GIN ? NINST : 1]) {
    if constexpr (GIN) {
      const int x = xb - 1 + i;
      const bool inside = (unsigned)x < (unsigned)X;
      const float* gp = gcol + (int64_t)(inside ? x : 0) * Y * Z;
#pragma unroll
      for (int u = 0; u < NINST; ++u) gv[u] = gp[grel[u]];  // pieces outside the image read a valid voxel of the column and are never used
    }
  };
  auto apply_gate = [&](int i, const float (&gv)[GIN ? NINST : 1]) {
    if constexpr (GIN) {
      char* dst = Rl + (i & (MC_NR - 1)) * PLANE_BYTES + lane * 16;
#pragma unroll
      for (int u = 0; u < NINST; ++u) {
        if (!((okmask >> u) & 1u)) continue;
        uint4* p = reinterpret_cast<uint4*>(dst + (u * 4 + wave) * 1024);
        const uint4 q = *p;
        const float gg = 1.f + gv[u];
        uint4 o;
        o.x = f2bf2(vsseg_mul_unpacked(__uint_as_float(q.x << 16), gg), vsseg_mul_unpacked(__uint_as_float(q.x & 0xffff0000u), gg));
        o.y = f2bf2(vsseg_mul_unpacked(__uint_as_float(q.y << 16), gg), vsseg_mul_unpacked(__uint_as_float(q.y & 0xffff0000u), gg));
        o.z = f2bf2(vsseg_mul_unpacked(__uint_as_float(q.z << 16), gg), vsseg_mul_unpacked(__uint_as_float(q.z & 0xffff0000u), gg));
        o.w = f2bf2(vsseg_mul_unpacked(__uint_as_float(q.w << 16), gg), vsseg_mul_unpacked(__uint_as_float(q.w & 0xffff0000u), gg));
        *p = o;
      }
    }
  };

  // ---- MFMA operand addressing: K-group p = ks*4 + g -> (tap p / G, piece p % G); lane column l15 -> voxel (row l15 / TZ, z l15 % TZ) of the M-tile
  const int rr = l15 / TZ, zz = l15 % TZ;
  int koff[KSTEPS], dxk[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    int p = ks * 4 + g;
    constexpr int NTAPG = (PS ? 4 : 9) * G;
    if (p >= NTAPG) p -= NTAPG;  // padded K-groups: zero weights times a genuine tap of the same voxel (conflict-free like the real ones)
    const int tap = p / G, pc = p % G, dy = PS ? (tap & 1) : tap % 3 - 1;  // PS: tap = dx*2 + dy with dx, dy in {0, +1}
    const int row = 1 + rr + dy;  // + the M-tile's first row (a multiple of RPM: it does not change the swizzle term)
    koff[ks] = (row * RS + mc_mod<G>(pc + 2 * (row * RS / 16)) * TZ + zz) * 16 + wave * (MT * MT_BYTES);
    dxk[ks] = PS ? 1 + (tap >> 1) : tap / 3;
  }
  const unsigned out_es = k.out_f32 ? 4u : 2u;
  const bool vec_store = (cout & 3) == 0;
  const bool simple = vec_store && !k.out_f32 && !k.scale && (k.act == VSSEG_ACT_NONE || k.act == VSSEG_ACT_PRELU);
  const int ekind = !simple ? 2 : (k.aux_mode == 3 ? 1 : 0);
  const float alpha_eff = k.act == VSSEG_ACT_PRELU ? alpha : 1.f;
  const char* Wlane = Wl + lane * 16;
  float ssum[STATS ? NT : 1][4], ssq[STATS ? NT : 1][4];
#pragma unroll
  for (int t = 0; t < (STATS ? NT : 1); ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) { ssum[t][r] = 0.f; ssq[t][r] = 0.f; }

  // output voxel of M-tile m (this lane's column) at x: ((n*X + x)*Y + y0 + (wave*MT + m)*RPM + rr)*Z + z0 + zz
  // PS: fine voxel (n, 2x + px, 2(y0 + row) + py, z0 + zz) of the (2X, 2Y, Z) output; ocol = its (px, py) = (0, 0) corner at x = 0, oplane = two fine planes
  const int64_t ocol = PS ? (((int64_t)n * 2 * X) * (2 * Y) + 2 * (y0 + (wave * MT) * RPM + rr)) * Z + z0 + zz : col0 + (int64_t)((wave * MT) * RPM + rr) * Z + zz;
  const int64_t oplane = PS ? (int64_t)4 * Y * Z : (int64_t)Y * Z;
  auto out_ch = [&](int t) -> int { return t * 16 + g * 4; };

  uint2 auxv[AUXM ? MT : 1][AUXM ? NT : 1], auxn[AUXM ? MT : 1][AUXM ? NT : 1];
  float gatev[AUXM ? MT : 1], gaten[AUXM ? MT : 1];
  auto load_aux = [&](int i, uint2 (&av)[AUXM ? MT : 1][AUXM ? NT : 1], float (&gv)[AUXM ? MT : 1]) {
    if constexpr (AUXM) {
      const int64_t vox0 = ocol + (int64_t)(xb - 1 + i) * oplane;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int64_t vox = vox0 + (int64_t)m * RPM * Z * (PS ? 2 : 1);
        if constexpr (PS) {  // accumulate: the previous gradient at the fine voxels this step stores to (same addresses as the store)
          gv[m] = 0.f;
#pragma unroll
          for (int t = 0; t < NT; ++t)
            av[m][t] = *reinterpret_cast<const uint2*>(k.aux0 + (vox + (int64_t)((t / (PS ? TPC : 1)) >> 1) * (2 * Y) * Z + ((t / (PS ? TPC : 1)) & 1) * Z) * k.aux_vox_bytes + (t % (PS ? TPC : 1)) * 32 + g * 8);
          continue;
        }
        gv[m] = k.aux_mode == 4 ? k.gate[vox] : (k.aux_mode == 5 ? bf2f(k.x1c[vox]) : 0.f);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int c = t * 16 + g * 4;
          const char* ap = (t * 16 >= k.aux_csplit ? k.aux1 : k.aux0) + vox * k.aux_vox_bytes + c * 2;
          av[m][t] = (c < cout && k.aux_mode != 5) ? *reinterpret_cast<const uint2*>(ap) : make_uint2(0u, 0u);
        }
      }
    }
  };

  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();  // the ring is zeroed before any DMA writes it
  load_aux(1, auxv, gatev);
  float gin0[GIN ? NINST : 1], gin1[GIN ? NINST : 1], gin[GIN ? NINST : 1];
  load_gate(0, gin0);
  load_gate(1, gin1);
  load_gate(2, gin);
  unsigned cv0[C1 ? NINST : 1], cv1[C1 ? NINST : 1], cvn[C1 ? NINST : 1];
  if constexpr (C1) {
    loadc(0, cv0);
    loadc(1, cv1);
    loadc(2, cvn);
    storec(0, cv0);
    storec(1, cv1);
    storec(2, cvn);
  } else {
    issue(0);
    issue(1);
    issue(2);
  }
  if constexpr (GIN) {  // the three prologue planes are gated once they have landed
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    apply_gate(0, gin0);
    apply_gate(1, gin1);
    apply_gate(2, gin);
  }

  for (int i = 1; i <= steps; ++i) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of plane i+1 have landed (and the previous step's stores have left)
    if constexpr (GIN) {
      if (i > 1) apply_gate(i + 1, gin);                 // ... and are gated by the thread that fetched them
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if constexpr (C1) {
      if (i > 1) storec(i + 1, cvn);                     // ... (C1) or are written as zero-extended pieces by the thread that loaded the values
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                        // ... everybody's; and every wave has finished reading plane i-2
    if (i < steps) load_aux(i + 1, auxn, gaten);         // next step's auxiliary operand, in front of the DMAs
    if (i + 2 <= steps + 1) {
      if constexpr (C1) loadc(i + 2, cvn);
      else { load_gate(i + 2, gin); if constexpr (!SPREAD) issue(i + 2); }
    }
    const bool spread_on = SPREAD && i + 2 <= steps + 1;
    const int sm1 = ((i - 1) & (MC_NR - 1)) * PLANE_BYTES, s0 = (i & (MC_NR - 1)) * PLANE_BYTES, sp1 = ((i + 1) & (MC_NR - 1)) * PLANE_BYTES;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 racc[NR ? MT : 1][NR ? NR : 1];
    if constexpr (NR > 0) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NR; ++t) racc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int wks = ks;
      bf16x8 w[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if constexpr (WREG) w[t] = wreg[wks][t];
        else w[t] = *reinterpret_cast<const bf16x8*>(Wlane + (wks * NT + t) * 1024);
      }
      const char* hb = Rl + koff[ks] + (dxk[ks] == 0 ? sm1 : (dxk[ks] == 1 ? s0 : sp1));
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const bf16x8 av = *reinterpret_cast<const bf16x8*>(hb + m * MT_BYTES);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[t], av, acc[m][t], 0, 0, 0);
        if constexpr (NR > 0) {
          if (ks >= KLO && ks < KHI) {
#pragma unroll
            for (int t = 0; t < NR; ++t) racc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wres[ks >= KLO ? ks - KLO : 0][t], av, racc[m][t], 0, 0, 0);
          }
        }
      }
      if constexpr (SPREAD) {  // the next plane's DMA pieces between the K-steps instead of all in front of them
        if (spread_on) {
#pragma unroll
          for (int u = 0; u < NINST; ++u)
            if (ks == (KSTEPS > NINST ? (u * KSTEPS) / NINST : (u < KSTEPS ? u : KSTEPS - 1))) issue_piece(i + 2, u);
        }
      }
    }

    // ---- epilogue (sconv.hip's, on the marching voxel mapping): bias (+ statistics) (+ eval affine) + activation (+ auxiliary operand)
    const int64_t ovox0 = ocol + (int64_t)(xb - 1 + i) * oplane;
    auto epilogue = [&](auto kind_c) {
      constexpr int KIND = decltype(kind_c)::value;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int64_t ovox = ovox0 + (int64_t)m * RPM * Z * (PS ? 2 : 1);
        float gt = 1.f;
        if constexpr (AUXM) gt = k.aux_mode == 4 ? 1.f + gatev[m] : 1.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int c = t * 16 + g * 4;
          if (!PS && c >= cout) continue;
          const float4 bi = *reinterpret_cast<const float4*>(epi + c);
          float val[4] = {acc[m][t][0] + bi.x, acc[m][t][1] + bi.y, acc[m][t][2] + bi.z, acc[m][t][3] + bi.w};
          if constexpr (STATS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { ssum[t][r] += val[r]; ssq[t][r] += val[r] * val[r]; }
          }
          if constexpr (KIND == 2) {
            if (k.scale) {
              const float4 sc = *reinterpret_cast<const float4*>(epi + NT * 16 + c), sh = *reinterpret_cast<const float4*>(epi + 2 * NT * 16 + c);
              val[0] = val[0] * sc.x + sh.x; val[1] = val[1] * sc.y + sh.y; val[2] = val[2] * sc.z + sh.z; val[3] = val[3] * sc.w + sh.w;
            }
            if (k.act == VSSEG_ACT_PRELU) {
#pragma unroll
              for (int r = 0; r < 4; ++r) val[r] = val[r] > 0.f ? val[r] : alpha * val[r];
            } else if (k.act == VSSEG_ACT_RELU) {
#pragma unroll
              for (int r = 0; r < 4; ++r) val[r] = fmaxf(val[r], 0.f);
            } else if (k.act == VSSEG_ACT_SIGMOID) {
#pragma unroll
              for (int r = 0; r < 4; ++r) val[r] = 1.f / (1.f + __expf(-val[r]));
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) val[r] = val[r] > 0.f ? val[r] : alpha_eff * val[r];
          }
          if constexpr (NR > 0) {
            if (k.res_out == nullptr && t < NR) {  // eval: + residual(x), behind the activation
#pragma unroll
              for (int r = 0; r < 4; ++r) val[r] += racc[m][t][r] + rb[t][r];
            }
          }
          if constexpr (AUXM) {
            const uint2 a = auxv[m][t];
            const float4 av = make_float4(__uint_as_float(a.x << 16), __uint_as_float(a.x & 0xffff0000u), __uint_as_float(a.y << 16), __uint_as_float(a.y & 0xffff0000u));
            if (KIND == 1 || (KIND == 2 && k.aux_mode == 3)) {
              val[0] = av.x > 0.f ? val[0] : 0.f; val[1] = av.y > 0.f ? val[1] : 0.f; val[2] = av.z > 0.f ? val[2] : 0.f; val[3] = av.w > 0.f ? val[3] : 0.f;
            } else if (KIND == 2 && k.aux_mode == 5) {  // + x1 * w[c] + b[c]: the 1 -> C residual convolution, never materialised
              const float4 w1 = *reinterpret_cast<const float4*>(epi + 3 * NT * 16 + c), b1 = *reinterpret_cast<const float4*>(epi + 4 * NT * 16 + c);
              const float x1 = gatev[m];
              val[0] += x1 * w1.x + b1.x; val[1] += x1 * w1.y + b1.y; val[2] += x1 * w1.z + b1.z; val[3] += x1 * w1.w + b1.w;
            } else if (KIND == 2 && k.aux_mode != 4) {
              val[0] += av.x; val[1] += av.y; val[2] += av.z; val[3] += av.w;
            } else {
              val[0] = vsseg_fma_unpacked(av.x, gt, val[0]); val[1] = vsseg_fma_unpacked(av.y, gt, val[1]);  // (not v_pk_fma_f32 op_sel: common.h)
              val[2] = vsseg_fma_unpacked(av.z, gt, val[2]); val[3] = vsseg_fma_unpacked(av.w, gt, val[3]);
            }
          }
          char* op;
          if constexpr (PS) op = k.out0 + (ovox + (int64_t)((t / TPC) >> 1) * (2 * Y) * Z + ((t / TPC) & 1) * Z) * k.out_vox_bytes + (t % TPC) * 32 + g * 8;  // class (px, py): + px fine planes, + py fine rows
          else op = (t * 16 >= k.out_csplit ? k.out1 : k.out0) + ovox * k.out_vox_bytes + out_ch(t) * (int)out_es;
          if constexpr (KIND != 2) {
            st4(reinterpret_cast<bf16_t*>(op), make_float4(val[0], val[1], val[2], val[3]));
          } else if (vec_store) {
            if (k.out_f32) st4(reinterpret_cast<float*>(op), make_float4(val[0], val[1], val[2], val[3]));
            else st4(reinterpret_cast<bf16_t*>(op), make_float4(val[0], val[1], val[2], val[3]));
          } else if (cout == 2 && k.out_f32) {  // the logits: one 8-byte store per voxel
            *reinterpret_cast<float2*>(op) = make_float2(val[0], val[1]);
          } else {  // 1-channel outputs (attention map)
            const int nc = min(4, cout - c);
            for (int r = 0; r < nc; ++r) {
              if (k.out_f32) reinterpret_cast<float*>(op)[r] = val[r];
              else reinterpret_cast<bf16_t*>(op)[r] = f2bf(val[r]);
            }
          }
        }
      }
    };
    if (ekind == 0) epilogue(std::integral_constant<int, 0>{});
    else if (AUXM && ekind == 1) epilogue(std::integral_constant<int, AUXM ? 1 : 0>{});
    else epilogue(std::integral_constant<int, 2>{});
    if constexpr (NR > 0) {
      if (k.res_out != nullptr) {  // the residual convolution's own output tensor (training: added behind the BatchNorm pass)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          char* rp = k.res_out + (ovox0 + (int64_t)m * RPM * Z) * k.res_vox_bytes + g * 8;
#pragma unroll
          for (int t = 0; t < NR; ++t)
            if (t * 16 + g * 4 < k.res_cout) st4(reinterpret_cast<bf16_t*>(rp + t * 32), make_float4(racc[m][t][0] + rb[t][0], racc[m][t][1] + rb[t][1], racc[m][t][2] + rb[t][2], racc[m][t][3] + rb[t][3]));
        }
      }
    }
    if constexpr (AUXM) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        gatev[m] = gaten[m];
#pragma unroll
        for (int t = 0; t < NT; ++t) auxv[m][t] = auxn[m][t];
      }
    }
  }

  if constexpr (STATS) {  // per-channel sum / sum of squares of this workgroup's voxels: shuffle tree -> one LDS row per wave, summed in wave order -> the layer's
                          // sharded statistics as fixed-point integer atomics (order-independent: vsseg_fx_add; layout of vsseg_igemm_desc.stats)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);  // [4 waves][2][NT*16]: the weights / the ring are no longer needed
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = ssum[t][r], q = ssq[t][r];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
        if (l15 == 0) {
          red[wave * (2 * NT * 16) + t * 16 + g * 4 + r] = s;
          red[wave * (2 * NT * 16) + NT * 16 + t * 16 + g * 4 + r] = q;
        }
      }
    __syncthreads();
    double* st = k.stats + (int64_t)(blockIdx.x % VSSEG_STAT_SHARDS) * 2 * k.stats_stride;
    for (int i = tid; i < 2 * NT * 16; i += 256) {
      const int which = i / (NT * 16), c = i - which * NT * 16;
      const float v = (red[i] + red[2 * NT * 16 + i]) + (red[4 * NT * 16 + i] + red[6 * NT * 16 + i]);
      if (PS || c < cout) vsseg_fx_add(&st[which * k.stats_stride + (k.cout_mod > 0 ? c % k.cout_mod : c)], (double)v, VSSEG_FX_STAT, k.fxflag);
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------
template <int CIN, int NT, int TZ, int MT, bool WREG> static int mc_lds() {
  constexpr int G = CIN / 8, RS = TZ * G, RPM = 16 / TZ, ROWS = MT * 4 * RPM + 2, KSTEPS = (9 * G + 3) / 4;
  constexpr int red = 4 * 2 * NT * 16 * 4;  // statistics reduction buffer (aliases the weights / the ring)
  const int lds = (WREG ? 0 : KSTEPS * NT * 1024) + MC_NR * ((ROWS * RS * 16 + 255) / 256 * 256) + 5 * NT * 16 * 4 + 16;
  return lds > red ? lds : red;
}
template <int CIN, int NT, int TZ, int MT, int MODE, bool WREG> static int mc_launch_mode(const MconvK& k, int grid, hipStream_t s) {
  static bool init_dev[16] = {}; bool& init = vsseg_dev_once(init_dev);  // per device: the LDS opt-in is a per-device function attribute
  const int lds = mc_lds<CIN, NT, TZ, MT, WREG>();
  if (!init) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&mconv_kernel<CIN, NT, TZ, MT, MODE, WREG>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    init = true;
  }
  hipLaunchKernelGGL((mconv_kernel<CIN, NT, TZ, MT, MODE, WREG>), dim3((unsigned)grid), dim3(256), lds, s, k);
  VSSEG_LAUNCH_CHECK("vsseg_igemm (marching)");
  return VSSEG_OK;
}
template <int CIN, int NT, int TZ, int MT, bool WREG, int NR> static int mc_launch_res(const MconvK& k, int grid, hipStream_t s) {  // + NR residual tiles: plain / statistics epilogues
  if (k.aux_mode) { vsseg_set_error("vsseg_igemm: residual tiles combine with the plain and the statistics epilogue only"); return VSSEG_EINVAL; }
  static bool init_dev[16] = {}; bool& init = vsseg_dev_once(init_dev);  // per device: the LDS opt-in is a per-device function attribute
  const int lds = mc_lds<CIN, NT, TZ, MT, WREG>();
  if (!init) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&mconv_kernel<CIN, NT, TZ, MT, 0, WREG, NR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&mconv_kernel<CIN, NT, TZ, MT, 1, WREG, NR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if constexpr (CIN == 64) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(&mconv_kernel<CIN, NT, TZ, MT, 3, WREG, NR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipFuncSetAttribute(reinterpret_cast<const void*>(&mconv_kernel<CIN, NT, TZ, MT, 4, WREG, NR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    init = true;
  }
  if (k.in_gate) {  // the attention gate in front of the unit applied on load (the level-1 decoder unit: 64 input channels)
    if constexpr (CIN == 64) {
      if (k.stats) hipLaunchKernelGGL((mconv_kernel<CIN, NT, TZ, MT, 4, WREG, NR>), dim3((unsigned)grid), dim3(256), lds, s, k);
      else hipLaunchKernelGGL((mconv_kernel<CIN, NT, TZ, MT, 3, WREG, NR>), dim3((unsigned)grid), dim3(256), lds, s, k);
    } else {
      vsseg_set_error("vsseg_igemm: no marching-kernel instantiation with residual tiles and the input gate for this shape");
      return VSSEG_EINVAL;
    }
  } else if (k.stats) hipLaunchKernelGGL((mconv_kernel<CIN, NT, TZ, MT, 1, WREG, NR>), dim3((unsigned)grid), dim3(256), lds, s, k);
  else hipLaunchKernelGGL((mconv_kernel<CIN, NT, TZ, MT, 0, WREG, NR>), dim3((unsigned)grid), dim3(256), lds, s, k);
  VSSEG_LAUNCH_CHECK("vsseg_igemm (marching, residual tiles)");
  return VSSEG_OK;
}
template <int CIN, int NT, int TZ, int MT, int MODE, int CC> static int mc_launch_c1_mode(const MconvK& k, int grid, hipStream_t s) {
  static bool init_dev[16] = {}; bool& init = vsseg_dev_once(init_dev);  // per device: the LDS opt-in is a per-device function attribute
  const int lds = mc_lds<CIN, NT, TZ, MT, false>();
  if (!init) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&mconv_kernel<CIN, NT, TZ, MT, MODE, false, 0, CC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    init = true;
  }
  hipLaunchKernelGGL((mconv_kernel<CIN, NT, TZ, MT, MODE, false, 0, CC>), dim3((unsigned)grid), dim3(256), lds, s, k);
  VSSEG_LAUNCH_CHECK("vsseg_igemm (marching, compact input)");
  return VSSEG_OK;
}
template <int CIN, int NT, int TZ, int MT> static int mc_launch_c1(const MconvK& k, int grid, hipStream_t s) {  // compact one- / two-channel input (CIN 8 entries)
  if (k.in_gate) { vsseg_set_error("vsseg_igemm: the input gate does not combine with a compact input"); return VSSEG_EINVAL; }
  if (k.in_vox_bytes == 4) {  // two channels: the logits gradient (plain epilogue only)
    if (k.stats || k.aux_mode) { vsseg_set_error("vsseg_igemm: a compact two-channel input combines with the plain epilogue only"); return VSSEG_EINVAL; }
    return mc_launch_c1_mode<CIN, NT, TZ, MT, 0, 2>(k, grid, s);
  }
  if (k.stats) return mc_launch_c1_mode<CIN, NT, TZ, MT, 1, 1>(k, grid, s);
  if (k.aux_mode) return mc_launch_c1_mode<CIN, NT, TZ, MT, 2, 1>(k, grid, s);
  return mc_launch_c1_mode<CIN, NT, TZ, MT, 0, 1>(k, grid, s);
}
template <int CIN, int NT, int TZ, int MT, bool WREG> static int mc_lds_ps() {  // pixel-shuffle launches: (4G + 3) / 4 K-steps of weights
  constexpr int G = CIN / 8, RS = TZ * G, RPM = 16 / TZ, ROWS = MT * 4 * RPM + 2;
  constexpr int red = 4 * 2 * NT * 16 * 4;
  const int lds = (WREG ? 0 : ((4 * G + 3) / 4) * NT * 1024) + MC_NR * ((ROWS * RS * 16 + 255) / 256 * 256) + 5 * NT * 16 * 4 + 16;
  return lds > red ? lds : red;
}
template <int CIN, int NT, int TZ, int MT, bool WREG> static int mc_launch_ps(const MconvK& k, int grid, hipStream_t s) {
  if constexpr ((CIN == 32 && NT == 4) || (CIN == 48 && NT == 8 && !WREG)) {  // the transposed convolutions: plain (eval) / statistics epilogue
    constexpr int TPC = NT / 4;
    static bool init_dev[16] = {}; bool& init = vsseg_dev_once(init_dev);  // per device: the LDS opt-in is a per-device function attribute
    const int lds = mc_lds_ps<CIN, NT, TZ, MT, WREG>();
    if (!init) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(&mconv_kernel<CIN, NT, TZ, MT, 0, WREG, 0, 0, TPC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipFuncSetAttribute(reinterpret_cast<const void*>(&mconv_kernel<CIN, NT, TZ, MT, 1, WREG, 0, 0, TPC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      init = true;
    }
    if (k.aux_mode) { vsseg_set_error("vsseg_igemm: this pixel-shuffle shape has no accumulating instantiation"); return VSSEG_EINVAL; }
    if (k.stats) hipLaunchKernelGGL((mconv_kernel<CIN, NT, TZ, MT, 1, WREG, 0, 0, TPC>), dim3((unsigned)grid), dim3(256), lds, s, k);
    else hipLaunchKernelGGL((mconv_kernel<CIN, NT, TZ, MT, 0, WREG, 0, 0, TPC>), dim3((unsigned)grid), dim3(256), lds, s, k);
    VSSEG_LAUNCH_CHECK("vsseg_igemm (marching, fused output-parity classes)");
    return VSSEG_OK;
  } else if constexpr (((CIN == 16 && NT == 4) || (CIN == 32 && NT == 8)) && !WREG) {  // the data gradients of the strided convolutions: plain / accumulating epilogue
    constexpr int TPC = NT / 4;
    static bool init_dev[16] = {}; bool& init = vsseg_dev_once(init_dev);  // per device: the LDS opt-in is a per-device function attribute
    const int lds = mc_lds_ps<CIN, NT, TZ, MT, WREG>();
    if (!init) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(&mconv_kernel<CIN, NT, TZ, MT, 0, WREG, 0, 0, TPC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipFuncSetAttribute(reinterpret_cast<const void*>(&mconv_kernel<CIN, NT, TZ, MT, 2, WREG, 0, 0, TPC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      init = true;
    }
    if (k.stats) { vsseg_set_error("vsseg_igemm: this pixel-shuffle shape has no statistics instantiation"); return VSSEG_EINVAL; }
    if (k.aux_mode) hipLaunchKernelGGL((mconv_kernel<CIN, NT, TZ, MT, 2, WREG, 0, 0, TPC>), dim3((unsigned)grid), dim3(256), lds, s, k);
    else hipLaunchKernelGGL((mconv_kernel<CIN, NT, TZ, MT, 0, WREG, 0, 0, TPC>), dim3((unsigned)grid), dim3(256), lds, s, k);
    VSSEG_LAUNCH_CHECK("vsseg_igemm (marching, fused output-parity classes)");
    return VSSEG_OK;
  } else {
    vsseg_set_error("vsseg_igemm: the marching kernel runs fused output-parity classes for 16 / 32 -> 4 x 16, 32 / 48 -> 4 x 32 channels only");
    return VSSEG_EINVAL;
  }
}
template <int CIN, int NT, int TZ, int MT> static int mc_launch_psonly(const MconvK& k, int grid, hipStream_t s) {  // table entries that exist for pixel-shuffle launches only
  if (!k.ps) { vsseg_set_error("vsseg_igemm: this marching-kernel shape is instantiated for fused output-parity classes only"); return VSSEG_EINVAL; }
  return mc_launch_ps<CIN, NT, TZ, MT, false>(k, grid, s);
}
template <int CIN, int NT, int TZ, int MT, bool WREG> static int mc_launch(const MconvK& k, int grid, hipStream_t s) {
  if (k.ps) return mc_launch_ps<CIN, NT, TZ, MT, WREG>(k, grid, s);
  if (k.in_gate) {
    if constexpr (CIN == 32 && NT == 1) return mc_launch_mode<CIN, NT, TZ, MT, 3, WREG>(k, grid, s);  // the level-0 decoder convolution behind the attention gate
    else { vsseg_set_error("vsseg_igemm: no marching-kernel instantiation with the input gate for this shape"); return VSSEG_EINVAL; }
  }
  if (k.stats) return mc_launch_mode<CIN, NT, TZ, MT, 1, WREG>(k, grid, s);
  if (k.aux_mode) return mc_launch_mode<CIN, NT, TZ, MT, 2, WREG>(k, grid, s);
  return mc_launch_mode<CIN, NT, TZ, MT, 0, WREG>(k, grid, s);
}

typedef int (*mc_fn_t)(const MconvK&, int, hipStream_t);
struct McEntry { int cin, nt, tz, mt; mc_fn_t fn; int (*lds)(); mc_fn_t fn_wreg; int (*lds_wreg)(); int nr; mc_fn_t fn_c1; };
#define MC_E(C, N, Z, M) {C, N, Z, M, mc_launch<C, N, Z, M, false>, mc_lds<C, N, Z, M, false>, nullptr, nullptr}
#define MC_1(N, Z, M) {8, N, Z, M, mc_launch<8, N, Z, M, false>, mc_lds<8, N, Z, M, false>, nullptr, nullptr, 0, mc_launch_c1<8, N, Z, M>}  // one zero-extended channel group, also from a COMPACT one-channel tensor
#define MC_W(C, N, Z, M) {C, N, Z, M, mc_launch<C, N, Z, M, false>, mc_lds<C, N, Z, M, false>, mc_launch<C, N, Z, M, true>, mc_lds<C, N, Z, M, true>}  // + the depth -6 twin (weights in registers)
#define MC_P(C, N, Z, M) {C, N, Z, M, mc_launch_psonly<C, N, Z, M>, mc_lds_ps<C, N, Z, M, false>, nullptr, nullptr}  // fused output-parity classes only (48 -> 4 x 32 channels)
#define MC_R(C, N, R, Z, M) {C, N, Z, M, mc_launch_res<C, N, Z, M, false, R>, mc_lds<C, N, Z, M, false>, mc_launch_res<C, N, Z, M, true, R>, mc_lds<C, N, Z, M, true>, R}  // + R residual tiles (res_tiles)
// (input channels, 16-channel output tiles, TZ, M-tiles per wave): rows per workgroup TYB = 64 * MT / TZ
static const McEntry mc_table[] = {
    MC_1(1, 8, 8), MC_1(2, 8, 8), MC_1(1, 4, 8), MC_1(2, 4, 8), MC_1(1, 4, 4), MC_1(2, 4, 4),  // 1 / 2 real channels zero-extended to one 8-channel group -> 16 / 32
    MC_E(16, 1, 4, 8), MC_E(16, 1, 4, 4), MC_E(16, 2, 4, 8), MC_E(16, 2, 4, 4), MC_E(16, 2, 8, 8), MC_E(16, 1, 8, 8),  // 16 -> 16 / 32 (levels 0, 1)
    MC_E(16, 1, 4, 2), MC_E(16, 1, 8, 4), MC_W(16, 2, 4, 2), MC_W(16, 2, 8, 4), MC_W(32, 1, 4, 2), MC_W(32, 2, 4, 2), MC_1(1, 8, 4), MC_1(2, 8, 4),  // 32-row columns: more, longer marches at batch 1 (sliding-window predictor)
    MC_W(32, 1, 2, 4), MC_W(32, 1, 4, 4), MC_W(32, 1, 2, 2), MC_W(32, 2, 4, 4), MC_W(32, 2, 2, 4), MC_W(32, 2, 2, 2), MC_E(32, 4, 4, 4), MC_W(32, 4, 2, 2), MC_W(32, 4, 4, 2),  // 32 -> 2 / 16 / 32 / 64
    MC_W(64, 2, 2, 2), MC_W(64, 2, 2, 1), MC_W(64, 1, 2, 2), MC_W(64, 1, 2, 1),                                                    // 64 -> 32 / 16
    MC_P(48, 8, 4, 2), MC_P(48, 8, 8, 4), MC_P(48, 8, 2, 1), MC_P(48, 8, 4, 4),                                                    // level-2 -> level-1 transposed convolution: 48 -> 4 classes x 32
    MC_P(16, 4, 8, 4), MC_P(16, 4, 4, 2), MC_P(16, 4, 4, 4), MC_P(16, 4, 8, 8), MC_P(32, 8, 4, 2), MC_P(32, 8, 8, 4), MC_P(32, 8, 2, 1), MC_P(32, 8, 4, 4),  // data gradients of the stride-(2,2,1) convolutions: 16 -> 4 x 16, 32 -> 4 x 32
    MC_R(16, 2, 2, 8, 4), MC_R(16, 2, 2, 8, 8), MC_R(16, 2, 2, 4, 4), MC_R(16, 2, 2, 4, 2), MC_R(64, 2, 2, 2, 1), MC_R(64, 2, 2, 2, 2)};   // ResidualUnit first convolutions of level 1 with their 1x1x1 residual convolution: 16 -> 32 + 32, 64 -> 32 + 32

static const McEntry* mc_find(const vsseg_igemm_desc* d, const char** why) {
  *why = nullptr;
  auto no = [&](const char* w) { *why = w; return (const McEntry*)nullptr; };
  if (d->in.dtype != VSSEG_BF16) return no("input is not bf16");
  const bool ps = d->os[0] == 2 && d->os[1] == 2 && d->os[2] == 1;  // fused output-parity classes of a stride-(2,2,1) transposed convolution: coarse lattice in, (2x, 2y, z) out, 4 taps
  if (ps) {
    const int tpc = d->nt / 4;
    if (d->nchunks != 1 || d->nsplit != 1 || d->ntaps != 4 || (d->ck != 16 && d->ck != 32 && d->ck != 48) || (d->nt != 4 && d->nt != 8) || d->ksteps != (4 * (d->ck / 8) + 3) / 4) return no("pixel-shuffle launches need one chunk of 16 / 32 / 48 input channels, 4 taps, 4 / 8 class tiles");
    if (d->is[0] != 1 || d->is[1] != 1 || d->is[2] != 1 || d->oo[0] || d->oo[1] || d->oo[2]) return no("pixel-shuffle launches need is = 1, os = (2, 2, 1), oo = 0");
    if (d->q[0] != d->in.x || d->q[1] != d->in.y || d->q[2] != d->in.z || 2 * d->q[0] != d->out.x || 2 * d->q[1] != d->out.y || d->q[2] != d->out.z) return no("pixel-shuffle output must be (2x, 2y, z) of the lattice");
    for (int t = 0; t < 4; ++t)
      if (d->tap_off[t][0] != (t >> 1) || d->tap_off[t][1] != (t & 1) || d->tap_off[t][2] != 0) return no("taps are not the 2x2x1 neighbourhood in (x, y) order");
    if (d->out.c != 16 * tpc || d->cout_mod != 16 * tpc || d->out.ptr2 || d->out.dtype != VSSEG_BF16 || (d->out.pitch & 3) || d->in.ptr2 || d->in.c != d->ck || d->in.pitch % 8 || ((uintptr_t)d->in.ptr & 15))
      return no("pixel-shuffle launches need a one-part 32 / 48-channel input and a one-part 16 / 32-channel bf16 output (cout_mod = channels)");
    if (d->res_mode != VSSEG_RES_NONE || d->in_gate || d->res_tiles || (d->accumulate && d->stats)) return no("pixel-shuffle launches support the plain, the statistics and the accumulating epilogue only");
    const int tz = d->tile[2], tyb = d->tile[1], mt = d->mtw;
    if ((tz != 2 && tz != 4 && tz != 8) || tyb != 64 * mt / tz || d->tile[0] < 1) return no("tile must be (x steps per workgroup, 64 * mtw / tz rows, tz in {2, 4, 8})");
    if (d->q[1] % tyb || d->q[2] % tz) return no("extent is not a multiple of the column block");
    for (const McEntry& e : mc_table)
      if (e.cin == d->ck && e.nt == d->nt && e.tz == tz && e.mt == mt && e.nr == 0) return (d->depth == -6 && !e.fn_wreg) ? no("no weights-in-registers instantiation (depth -6) for this shape") : &e;
    return no("no instantiation for this (channels, nt, tz, mtw)");
  }
  if (d->nchunks != 1 || d->nsplit != 1 || d->ntaps != 9) return no("needs nchunks = nsplit = 1 and the 9 taps of a 3x3x1 stencil");
  for (int a = 0; a < 3; ++a)
    if (d->is[a] != 1 || d->os[a] != 1 || d->oo[a] != 0) return no("stride-1 lattices only");
  if (d->q[0] != d->in.x || d->q[1] != d->in.y || d->q[2] != d->in.z || d->q[0] != d->out.x || d->q[1] != d->out.y || d->q[2] != d->out.z) return no("lattice, input and output extents differ");
  for (int t = 0; t < 9; ++t)
    if (d->tap_off[t][0] != t / 3 - 1 || d->tap_off[t][1] != t % 3 - 1 || d->tap_off[t][2] != 0) return no("taps are not the 3x3x1 stencil in (x, y) order");
  const int tz = d->tile[2], tyb = d->tile[1], mt = d->mtw;
  if ((tz != 2 && tz != 4 && tz != 8) || tyb != 64 * mt / tz || d->tile[0] < 1) return no("tile must be (x steps per workgroup, 64 * mtw / tz rows, tz in {2, 4, 8})");
  if (d->q[1] % tyb || d->q[2] % tz) return no("extent is not a multiple of the column block");
  const bool c1 = (d->in.c == 1 || d->in.c == 2) && d->in.pitch == d->in.c && d->ck == 8 && !d->in.ptr2;  // a compact one- / two-channel tensor standing for one zero-extended channel group
  if (!c1 && (d->in.c != d->ck || d->in.pitch % 8 || ((uintptr_t)d->in.ptr & 15) || ((uintptr_t)d->in.ptr2 & 15))) return no("input must be one channel chunk of 16-byte aligned voxel rows (or a compact one-channel tensor with ck = 8)");
  if (c1 && (d->depth != -5 || d->in_gate || d->res_tiles || ((uintptr_t)d->in.ptr & (2 * d->in.c - 1)))) return no("a compact input needs a depth -5 plan without input gate / residual tiles");
  if (d->ksteps != (9 * (d->ck / 8) + 3) / 4) return no("ksteps");
  if (d->out.c > d->nt * 16 || (d->out.dtype != VSSEG_BF16 && d->out.dtype != VSSEG_F32)) return no("output channels / dtype");
  if ((d->out.c & 3) == 0 && (d->out.pitch & 3)) return no("output pitch");
  if (d->stats && (d->accumulate || d->res_mode != VSSEG_RES_NONE)) return no("statistics combined with a residual");
  if (d->accumulate && d->res_mode != VSSEG_RES_NONE) return no("accumulate combined with a residual");
  if (d->res_mode == VSSEG_RES_IN1) {
    if (!d->in1 || !d->in1_w || !d->in1_b || (d->out.c & 3) || d->out.dtype != VSSEG_BF16 || d->accumulate || d->stats) return no("RES_IN1 needs in1 / in1_w / in1_b and a bf16 output");
  } else if (d->accumulate || d->res_mode != VSSEG_RES_NONE) {
    const vsseg_tensor& a = d->accumulate ? d->out : d->res;
    if ((d->out.c & 3) || (a.pitch & 3) || a.c < d->out.c || a.dtype != VSSEG_BF16) return no("auxiliary tensor layout / dtype");
  }
  if (d->res_tiles && (!d->wpack_res || d->res_tiles < 1 || d->res_tiles > d->nt || d->accumulate || d->res_mode != VSSEG_RES_NONE || d->out.ptr2 ||
                       (d->res_out.ptr && (d->res_out.dtype != VSSEG_BF16 || d->res_out.ptr2 || (d->res_out.c & 3) || d->res_out.c > d->res_tiles * 16 || (d->res_out.pitch & 3)))))
    return no("residual tiles need their packed weights, a plain or statistics epilogue and a one-part bf16 output");
  if (d->res_tiles && !d->res_out.ptr && (d->out.dtype != VSSEG_BF16 || (d->out.c & 3))) return no("residual tiles added in the epilogue need a bf16 output");
  for (const McEntry& e : mc_table)
    if (e.cin == d->ck && e.nt == d->nt && e.tz == tz && e.mt == mt && e.nr == d->res_tiles) {
      if (c1 && !e.fn_c1) return no("no compact-input instantiation for this shape");
      return (d->depth == -6 && !e.fn_wreg) ? no("no weights-in-registers instantiation (depth -6) for this shape") : &e;
    }
  return no("no instantiation for this (channels, nt, tz, mtw)");
}

int vsseg_mconv_lds_bytes(const vsseg_igemm_desc* d) {
  const char* why;
  const McEntry* e = mc_find(d, &why);
  if (!e) { vsseg_set_error("vsseg_igemm: depth -5 / -6 (marching kernel) not applicable: %s", why); return VSSEG_EINVAL; }
  if (d->os[0] == 2 && d->ck == 32 && d->nt == 4) return (d->depth == -6 ? e->lds_wreg() : e->lds()) - 5 * d->nt * 1024 * (d->depth == -6 ? 0 : 1);  // pixel shuffle on a general entry: 4 of the 9 K-steps of weights in LDS (the MC_P entries' lds() is already the pixel-shuffle figure)
  return d->depth == -6 ? e->lds_wreg() : e->lds();
}

int vsseg_mconv_launch(const vsseg_igemm_desc* d, const void* zeros, hipStream_t s) {
  const char* why;
  const McEntry* e = mc_find(d, &why);
  if (!e) { vsseg_set_error("vsseg_igemm: depth -5 / -6 (marching kernel) not applicable: %s", why); return VSSEG_EINVAL; }
  MconvK k;
  k.in0 = reinterpret_cast<const char*>(d->in.ptr);
  k.in1 = d->in.ptr2 ? reinterpret_cast<const char*>(d->in.ptr2) - (int64_t)d->in.csplit * 2 : k.in0;
  k.in_csplit_pc = d->in.ptr2 ? d->in.csplit / 8 : 1 << 20;
  k.in_vox_bytes = d->in.pitch * 2;
  const int oes = d->out.dtype == VSSEG_F32 ? 4 : 2;
  k.out0 = reinterpret_cast<char*>(d->out.ptr);
  k.out1 = d->out.ptr2 ? reinterpret_cast<char*>(d->out.ptr2) - (int64_t)d->out.csplit * oes : k.out0;
  k.out_csplit = d->out.ptr2 ? d->out.csplit : 0x7fffffff;
  k.out_vox_bytes = d->out.pitch * oes;
  k.out_f32 = d->out.dtype == VSSEG_F32;
  k.aux_mode = 0;
  k.aux0 = k.aux1 = nullptr; k.aux_csplit = 0x7fffffff; k.aux_vox_bytes = 0;
  if (d->accumulate) k.aux_mode = 1;
  else if (d->res_mode == VSSEG_RES_ADD) k.aux_mode = 2;
  else if (d->res_mode == VSSEG_RES_RELUMASK) k.aux_mode = 3;
  else if (d->res_mode == VSSEG_RES_GATE) k.aux_mode = 4;
  else if (d->res_mode == VSSEG_RES_IN1) k.aux_mode = 5;
  k.x1c = reinterpret_cast<const bf16_t*>(d->in1); k.in1_w = d->in1_w; k.in1_b = d->in1_b;
  if (k.aux_mode && k.aux_mode != 5) {
    const vsseg_tensor& a = d->accumulate ? d->out : d->res;
    k.aux0 = reinterpret_cast<const char*>(a.ptr);
    k.aux1 = a.ptr2 ? reinterpret_cast<const char*>(a.ptr2) - (int64_t)a.csplit * 2 : k.aux0;
    k.aux_csplit = a.ptr2 ? a.csplit : 0x7fffffff;
    k.aux_vox_bytes = a.pitch * 2;
  }
  VSSEG_CHECK(k.aux_mode != 4 || d->gate, "vsseg_igemm: RES_GATE needs the gate map");
  k.gate = d->gate;
  k.in_gate = d->in_gate;
  VSSEG_CHECK(!d->in_gate || !k.aux_mode, "vsseg_igemm: the input gate does not combine with an auxiliary operand");
  VSSEG_CHECK(!d->in_gate || !d->stats || d->res_tiles, "vsseg_igemm: the input gate combines with statistics only in the residual-tile instantiations");
  k.wpack = reinterpret_cast<const char*>(d->wpack);
  k.bias = d->bias; k.bias2 = d->bias2; k.scale = d->scale; k.shift = d->shift; k.alpha = d->alpha;
  k.stats = d->stats; k.stats_stride = d->stats_stride;
  VSSEG_FX_FLAG(fxflag_, "vsseg_igemm (marching kernel)");
  k.fxflag = fxflag_;
  k.zeros = zeros;
  k.act = d->act; k.cout = d->out.c; k.cout_mod = d->cout_mod;
  k.ps = d->os[0] == 2;
  k.X = d->q[0]; k.Y = d->q[1]; k.Z = d->q[2];
  k.lx = d->tile[0] > k.X ? k.X : d->tile[0];
  k.nxs = (k.X + k.lx - 1) / k.lx; k.nyb = k.Y / d->tile[1]; k.nzb = k.Z / d->tile[2];
  k.wpack_r = reinterpret_cast<const char*>(d->wpack_res); k.bias_r = d->bias_res;
  k.res_out = reinterpret_cast<char*>(d->res_out.ptr); k.res_vox_bytes = d->res_out.pitch * 2;
  k.res_cout = d->res_tiles ? (d->res_out.ptr ? d->res_out.c : d->out.c) : 0;
  const int64_t grid = (int64_t)d->in.n * k.nxs * k.nyb * k.nzb;
  VSSEG_CHECK(grid > 0 && grid < (1ll << 30), "vsseg_igemm: bad marching grid");
  if (d->in.c <= 2 && d->ck == 8) return e->fn_c1(k, (int)grid, s);
  return d->depth == -6 ? e->fn_wreg(k, (int)grid, s) : e->fn(k, (int)grid, s);
}
