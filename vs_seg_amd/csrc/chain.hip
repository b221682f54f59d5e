// Chained marching convolution (vsseg_conv_chain; the inference forward): TWO consecutive stride-1 3x3x1 bf16 convolutions of the two finest levels of the 2.5D U-Net as ONE launch,
//   h = act_a(scale_a * (conv_a(x) + bias_a) + shift_a)          (conv + eval-mode BatchNorm + PReLU: ref:params/networks/blocks/convolutions.py:114-146; or conv + ReLU)
//   y = act_b(scale_b * (conv_b(h) + bias_b) + shift_b) [+ x * in1_w + in1_b] [+ bf16(residual(x) + bias_r)]
// with the tensor between them (16 channels at 384x128x128: 201 MB written and read back per sliding-window patch) kept in LDS.  Users (engine.py, Plan._lower): the first
// ResidualUnit of the encoder (1 -> 16 -> 16 with its 1x1x1 residual convolution of the network input, ref:params/networks/nets/unet2d5_spvPA.py:52-77,
// blocks/convolutions.py:241-255), the attention block of the level-0 decoder (32 -> 16 -> 1 + sigmoid, ref:params/networks/blocks/attentionblock.py:20-41) and the
// two-sub-unit ResidualUnit of level 1 (16 -> 32 -> 32 with its residual convolution as residual tiles of stage B).  In TRAINING the BatchNorm between two sub-units needs the
// statistics of the whole tensor before any element of h exists, so the units are inference-only launches (the sliding-window predictor, ref:params/VSparams.py:553-567); the
// attention block has no BatchNorm; running it in the training forward too (h stored as well) measured no faster than its two launches and was deleted in round 6.
//
// Structure = mconv.hip's (a workgroup owns a column: sample n, ALL rows, slices [z0, z0 + TZ), and marches along x; a plane = one x position of the column, LDS layout
// [row][piece'][z] with the same bank swizzle) with a second ring: iteration s
//   * waits for input plane s (LDS-DMA, fetched LEAD = 1 or 3 iterations ahead; the compact one-channel input travels through registers, three planes ahead),
//   * issues the MFMAs of BOTH stages — stage A: input planes s-2, s-1, s -> h plane s-1; stage B: h planes s-4, s-3, s-2, all written in EARLIER iterations -> output plane
//     s-3 — before either epilogue: one barrier per iteration, the accumulator chains of the two stages are independent,
//   * stage A's epilogue writes the bf16 result into the H ring in stage B's operand layout (a lane of the 16x16 MFMA result holds 4 consecutive channels of one voxel = half
//     a 16-byte piece: one ds_write_b64), zeros where the plane lies outside the image (stage B's zero padding is a padding of h, not h of a padded x),
//   * stage B's epilogue stores output plane s-3; residual tiles (NR): the centre-tap K-steps of input plane s-3, which the ring keeps one iteration longer.
// Rows -1 and Y of both rings are the zero padding (all rows in one workgroup: no halo in y); an x segment re-fetches 4 input planes and recomputes 2 h planes.  One large
// workgroup (8 or 16 waves) per CU: its own waves are its latency cover, and the packed weights of both stages stay in registers (tools/bench_chain.py: 16 waves beat 8 beat 4).
// Same packed weights ([K-steps][tiles][64 lanes][8], K order (tap, channel group)), MFMA operand order and epilogue arithmetic as the mconv launches it replaces: the results
// are bit-identical to them (tests/test_gpu_ops.py::test_chained_marching_convolution_*).
#include "common.h"
#include "chain.h"

constexpr int CH_HR = 4;  // H ring: planes s-4 .. s-2 read, s-1 written

struct ChainK {
  const char* in0; const char* in1;  // two-part input as in mconv.hip (in1 biased by -csplit channels)
  int in_csplit_pc, in_vox_bytes;
  char* out;
  int out_vox_bytes, out_f32, cout;
  const char *wa, *wb;
  const float *bias_a, *scale_a, *shift_a, *alpha_a;
  const float *bias_b, *scale_b, *shift_b, *alpha_b;
  int act_a, act_b;
  const float *in1_w, *in1_b;
  const char* wres;      // NR > 0: packed weights [K-steps of the centre tap][NR][64][8] of a 1x1x1 convolution of the INPUT (the ResidualUnit's residual convolution) ...
  const float* bias_r;   // ... and its bias: out += bf16(residual(in) + bias_r) behind stage B's activation
  const void* zeros;
  int X, Y, Z, lx, nxs, nzb;
};

template <int G> __device__ __forceinline__ int ch_mod(int v) {
  if constexpr ((G & (G - 1)) == 0) return v & (G - 1);
  else return ((v % G) + G) % G;
}

// CIN: input channels of stage A (8 = one zero-extended channel group, read COMPACT when CC = 1), CM: channels between the stages, NTB: 16-channel output tiles of stage B,
// TZ: z voxels per column, CH_NW waves per workgroup with MT 16-voxel M-tiles each: Y = CH_NW * MT * 16 / TZ rows; LEAD: iterations between the fetch of an input plane and
// its first use (1: the next plane is in flight while this one is multiplied, as in mconv.hip — several small workgroups per CU cover the latency for each other;
// 3: one large workgroup per CU covers it with the depth of its own ring)
template <int CIN, int CM, int NTB, int TZ, int MT, int CC, int CH_NW, int LEAD, int NR>
__global__ __launch_bounds__(CH_NW * 64) void chain_kernel(const ChainK k) {
  constexpr bool RES = NR > 0;
  static_assert(!RES || (CC == 0 && NR == NTB), "residual tiles: one per output tile of stage B, ordinary input");
  constexpr int CH_THREADS = CH_NW * 64;
  constexpr bool C1 = CC != 0;
  static_assert(LEAD == 1 || LEAD == 3, "fetch distance");
  static_assert(!C1 || LEAD == 3, "the compact input keeps three planes in registers");
  static_assert(!C1 || CIN == 8, "a compact input is one zero-extended channel group");
  constexpr int GA = CIN / 8, GB = CM / 8, NTA = CM / 16;
  constexpr int RSA = TZ * GA, RSB = TZ * GB, RPM = 16 / TZ, TY = CH_NW * MT * RPM, ROWS = TY + 2;
  constexpr int SLOTS_A = ROWS * RSA, SLOTS_B = ROWS * RSB;
  constexpr int WIA = (SLOTS_A + 63) / 64;        // wave-instructions (64 slots of 16 bytes) per input plane ...
  constexpr int NIA = (WIA + CH_NW - 1) / CH_NW;  // ... per wave: NIA for waves < WFULL, NIA - 1 for the others
  constexpr int WFULL = WIA - (NIA - 1) * CH_NW;
  constexpr int PA_BYTES = WIA * 1024, PB_BYTES = (SLOTS_B * 16 + 255) / 256 * 256;
  constexpr int RING = C1 ? 5 : 3 + LEAD + (RES ? 1 : 0);         // input planes s-3 (compact: the residual operand of output plane s-3) / s-2 .. s in LDS, (DMA) s+1 .. s+LEAD in flight / being issued
  constexpr int KSA = (9 * GA + 3) / 4, KSB = (9 * GB + 3) / 4;
  constexpr int KLO = GA, KHI = (5 * GA + 3) / 4, KR = RES ? KHI - KLO : 0;  // K-steps of stage A's K order that hold the centre tap's channel groups [4G, 5G)
  constexpr int MTA_BYTES = RPM * RSA * 16, MTB_BYTES = RPM * RSB * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Rin = smem;
  char* Hl = smem + RING * PA_BYTES;
  float* epi = reinterpret_cast<float*>(smem + RING * PA_BYTES + CH_HR * PB_BYTES);  // A: bias, scale, shift [CM each]; B: bias, scale, shift, in1_w, in1_b [NTB*16 each]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, l15 = lane & 15;
  const int X = k.X, Y = k.Y, Z = k.Z, cout = k.cout;

  int b = vsseg_xcd_contiguous(blockIdx.x, gridDim.x);
  const int zb = b % k.nzb; b /= k.nzb;
  const int xs = b % k.nxs; const int n = b / k.nxs;
  const int z0 = zb * TZ, xb = xs * k.lx, steps = min(k.lx, X - xb);

  bf16x8 wa[KSA][NTA], wb[KSB][NTB];
#pragma unroll
  for (int ks = 0; ks < KSA; ++ks)
#pragma unroll
    for (int t = 0; t < NTA; ++t) wa[ks][t] = *reinterpret_cast<const bf16x8*>(k.wa + ((ks * NTA + t) * 64 + lane) * 16);
#pragma unroll
  for (int ks = 0; ks < KSB; ++ks)
#pragma unroll
    for (int t = 0; t < NTB; ++t) wb[ks][t] = *reinterpret_cast<const bf16x8*>(k.wb + ((ks * NTB + t) * 64 + lane) * 16);
  bf16x8 wres[RES ? KR : 1][RES ? NR : 1];
  float rb[RES ? NR : 1][4];
  if constexpr (RES) {
#pragma unroll
    for (int ks = 0; ks < KR; ++ks)
#pragma unroll
      for (int t = 0; t < NR; ++t) wres[ks][t] = *reinterpret_cast<const bf16x8*>(k.wres + ((ks * NR + t) * 64 + lane) * 16);
#pragma unroll
    for (int t = 0; t < NR; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = t * 16 + g * 4 + r;
        rb[t][r] = (k.bias_r && c < cout) ? k.bias_r[c] : 0.f;
      }
  }
  for (int i = tid; i < CH_HR * PB_BYTES / 16; i += CH_THREADS) reinterpret_cast<uint4*>(Hl)[i] = make_uint4(0u, 0u, 0u, 0u);  // rows -1 and Y of h stay zero
  for (int i = tid; i < CM; i += CH_THREADS) {
    epi[i] = k.bias_a ? k.bias_a[i] : 0.f;
    epi[CM + i] = k.scale_a ? k.scale_a[i] : 1.f;
    epi[2 * CM + i] = k.scale_a ? k.shift_a[i] : 0.f;
  }
  float* epb = epi + 3 * CM;
  for (int i = tid; i < NTB * 16; i += CH_THREADS) {
    const bool ok = i < cout;
    epb[i] = (ok && k.bias_b) ? k.bias_b[i] : 0.f;
    epb[NTB * 16 + i] = (ok && k.scale_b) ? k.scale_b[i] : 1.f;
    epb[2 * NTB * 16 + i] = (ok && k.scale_b) ? k.shift_b[i] : 0.f;
    epb[3 * NTB * 16 + i] = (ok && k.in1_w) ? k.in1_w[i] : 0.f;
    epb[4 * NTB * 16 + i] = (ok && k.in1_b) ? k.in1_b[i] : 0.f;
  }
  const float alpha_a = (k.act_a == VSSEG_ACT_PRELU && k.alpha_a) ? *k.alpha_a : 0.f;
  const float alpha_b = (k.act_b == VSSEG_ACT_PRELU && k.alpha_b) ? *k.alpha_b : 0.f;

  // ---- this thread's pieces of an input plane: slot j = (u*8 + wave)*64 + lane holds (row j / RSA, piece' (j % RSA) / TZ, z j % TZ); rows 0 and TY + 1 are the zero padding
  int rel[NIA];
  unsigned okmask = 0, p1mask = 0;
#pragma unroll
  for (int u = 0; u < NIA; ++u) {
    const int j = (u * CH_NW + wave) * 64 + lane;
    const int r = j / RSA, within = j % RSA, pp = within / TZ, z = within % TZ;
    const int pc = ch_mod<GA>(pp - 2 * (r * RSA / 16));
    const bool ok = j < SLOTS_A && (unsigned)(r - 1) < (unsigned)TY;
    rel[u] = ok ? ((r - 1) * Z + z) * k.in_vox_bytes + (C1 ? 0 : pc * 16) : 0;
    if (ok) okmask |= 1u << u;
    if (ok && pc >= k.in_csplit_pc) p1mask |= 1u << u;
  }
  const int64_t plane_stride = (int64_t)Y * Z * k.in_vox_bytes;
  const int64_t col0 = ((int64_t)n * X * Y) * Z + z0;  // voxel (n, 0, 0, z0)
  const char* org0 = k.in0 + col0 * k.in_vox_bytes;
  const char* org1 = k.in1 + col0 * k.in_vox_bytes;
  // input plane p has x = xb + p (p = -2 .. steps + 1) and lives in ring slot (p + 2) % RING; planes outside the image (and the planes behind the segment that only keep the
  // number of loads in flight constant) are zero
  auto issue = [&](int p) __attribute__((always_inline)) {
    const int x = xb + p;
    char* dst = Rin + ((p + 2) % RING) * PA_BYTES;
    const bool inside = (unsigned)x < (unsigned)X && p <= steps + 1;
    const char* p0 = org0 + (int64_t)x * plane_stride;
    const char* p1 = org1 + (int64_t)x * plane_stride;
#pragma unroll
    for (int u = 0; u < NIA; ++u)
      if (u < NIA - 1 || wave < WFULL)  // (wave-uniform: every wave issues a fixed number of loads per plane, the counted wait below relies on it)
        vsseg_dma16((inside && ((okmask >> u) & 1u)) ? (const void*)(((p1mask >> u) & 1u ? p1 : p0) + rel[u]) : k.zeros, dst + (u * CH_NW + wave) * 1024);
  };
  // compact input: this thread's values of plane p -> registers (unconditional loads from valid addresses; the zero padding is applied when the values are written to LDS,
  // three iterations later: a select right behind the load would wait for it)
  auto loadc = [&](int p, unsigned (&cv)[NIA]) __attribute__((always_inline)) {
    const int x = xb + p;
    const bool inside = (unsigned)x < (unsigned)X && p <= steps + 1;
    const char* p0 = org0 + (int64_t)(inside ? x : 0) * plane_stride;
#pragma unroll
    for (int u = 0; u < NIA; ++u) cv[u] = *reinterpret_cast<const unsigned short*>(p0 + rel[u]);
  };
  auto storec = [&](int p, const unsigned (&cv)[NIA]) __attribute__((always_inline)) {  // ... -> zero-extended 16-byte pieces of ring slot (p + 2) % RING
    const bool inside = (unsigned)(xb + p) < (unsigned)X && p <= steps + 1;
    char* dst = Rin + ((p + 2) % RING) * PA_BYTES + lane * 16;
#pragma unroll
    for (int u = 0; u < NIA; ++u)
      if ((u * CH_NW + wave) * 64 + lane < SLOTS_A) *reinterpret_cast<uint4*>(dst + (u * CH_NW + wave) * 1024) = make_uint4((inside && ((okmask >> u) & 1u)) ? cv[u] : 0u, 0u, 0u, 0u);
  };

  // ---- MFMA operand addressing (mconv.hip): K-group p = ks*4 + g -> (tap p / G, piece p % G); lane column l15 -> voxel (row l15 / TZ, z l15 % TZ) of the M-tile
  const int rr = l15 / TZ, zz = l15 % TZ;
  int koffA[KSA], dxA[KSA], koffB[KSB], dxB[KSB];
#pragma unroll
  for (int ks = 0; ks < KSA; ++ks) {
    int p = ks * 4 + g;
    if (p >= 9 * GA) p -= 9 * GA;  // padded K-groups: zero weights times a genuine tap of the same voxel
    const int tap = p / GA, pc = p % GA, row = 1 + rr + tap % 3 - 1;
    koffA[ks] = (row * RSA + ch_mod<GA>(pc + 2 * (row * RSA / 16)) * TZ + zz) * 16 + wave * (MT * MTA_BYTES);
    dxA[ks] = tap / 3;
  }
#pragma unroll
  for (int ks = 0; ks < KSB; ++ks) {
    int p = ks * 4 + g;
    if (p >= 9 * GB) p -= 9 * GB;
    const int tap = p / GB, pc = p % GB, row = 1 + rr + tap % 3 - 1;
    koffB[ks] = (row * RSB + ch_mod<GB>(pc + 2 * (row * RSB / 16)) * TZ + zz) * 16 + wave * (MT * MTB_BYTES);
    dxB[ks] = tap / 3;
  }
  // where stage A's result goes: tile t, lane group g = channels t*16 + g*4 .. + 3 = piece t*2 + g/2, half g & 1 of voxel (row 1 + rr, z zz) of the M-tile
  int hw[NTA];
#pragma unroll
  for (int t = 0; t < NTA; ++t) {
    const int row = 1 + rr, pcb = t * 2 + (g >> 1);
    hw[t] = (row * RSB + ch_mod<GB>(pcb + 2 * (row * RSB / 16)) * TZ + zz) * 16 + (g & 1) * 8 + wave * (MT * MTB_BYTES);
  }
  const int x1off = ((1 + rr) * RSA + zz) * 16 + wave * (MT * MTA_BYTES);  // (compact input) channel 0 of the voxel in an input plane
  const int64_t ocol = col0 + (int64_t)((wave * MT) * RPM + rr) * Z + zz;  // output voxel of M-tile m at x: ocol + x*Y*Z + m*RPM*Z
  const int64_t oplane = (int64_t)Y * Z;
  const unsigned out_es = k.out_f32 ? 4u : 2u;
  const bool vec_store = (cout & 3) == 0 && !k.out_f32;
  const bool out1 = cout == 1 && k.out_f32 && !k.in1_w;

  // ---- prologue: input planes -2 .. LEAD - 1 (DMA) / -2 .. -1 into LDS and 0 .. 2 into the three register sets (compact)
  unsigned c0[NIA], c1[NIA], c2[NIA];
  if constexpr (C1) {
    loadc(-2, c0);
    loadc(-1, c1);
    storec(-2, c0);
    storec(-1, c1);
    loadc(0, c0);
    loadc(1, c1);
    loadc(2, c2);
  } else {
#pragma unroll
    for (int p = -2; p < LEAD; ++p) issue(p);
  }
  // the packed weights are in their registers before the loop (the compiler would otherwise keep its counted waits for them INSIDE the loop, where they wait for stores)
#pragma unroll
  for (int ks = 0; ks < KSA; ++ks)
#pragma unroll
    for (int t = 0; t < NTA; ++t) asm volatile("" : "+v"(wa[ks][t]));
#pragma unroll
  for (int ks = 0; ks < KSB; ++ks)
#pragma unroll
    for (int t = 0; t < NTB; ++t) asm volatile("" : "+v"(wb[ks][t]));
  if constexpr (RES) {
#pragma unroll
    for (int ks = 0; ks < KR; ++ks)
#pragma unroll
      for (int t = 0; t < NR; ++t) asm volatile("" : "+v"(wres[ks][t]));
  }

  auto iter = [&](int s, unsigned (&cv)[NIA]) __attribute__((always_inline)) {
    if constexpr (C1) {
      storec(s, cv);  // input plane s, loaded three iterations ago (the compiler's own counted wait)
    } else if constexpr (LEAD == 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // input plane s has landed (and the previous iteration's stores have left)
    } else {  // input plane s has landed: the loads of planes s+1 and s+2 are the only younger LOADS (loads return in order; stores in flight only make the wait longer)
      if (wave < WFULL) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NIA) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NIA - 1)) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's LDS writes (stage A's h plane of the previous iteration, the compact pieces) are done
    __builtin_amdgcn_s_barrier();
    if constexpr (C1) loadc(s + 3, cv);
    else issue(s + LEAD);

    const bool a_on = s <= steps + 1, a_in = (unsigned)(xb + s - 1) < (unsigned)X, b_on = s >= 3;
    // ---- the MFMAs of both stages first (A: h plane s-1 from input planes s-2, s-1, s; B: output plane s-3 from h planes s-4, s-3, s-2 = H slots (s+1) & 3, (s+2) & 3,
    //      (s+3) & 3, all written in earlier iterations): eight independent accumulator chains, no LDS write between their operand reads
    f32x4 acca[MT][NTA], accb[MT][NTB], racc[RES ? MT : 1][RES ? NR : 1];
    if constexpr (RES) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NR; ++t) racc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int t = 0; t < NTA; ++t) acca[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < NTB; ++t) accb[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (a_on && a_in) {
      const int sm1 = (s % RING) * PA_BYTES, s0 = ((s + 1) % RING) * PA_BYTES, sp1 = ((s + 2) % RING) * PA_BYTES;
#pragma unroll
      for (int ks = 0; ks < KSA; ++ks) {
        const char* hb = Rin + koffA[ks] + (dxA[ks] == 0 ? sm1 : (dxA[ks] == 1 ? s0 : sp1));
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const bf16x8 av = *reinterpret_cast<const bf16x8*>(hb + m * MTA_BYTES);
#pragma unroll
          for (int t = 0; t < NTA; ++t) acca[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ks][t], av, acca[m][t], 0, 0, 0);
        }
      }
    }
    if (b_on) {
      const int hm1 = ((s + 1) & (CH_HR - 1)) * PB_BYTES, h0 = ((s + 2) & (CH_HR - 1)) * PB_BYTES, hp1 = ((s + 3) & (CH_HR - 1)) * PB_BYTES;
#pragma unroll
      for (int ks = 0; ks < KSB; ++ks) {
        const char* hb = Hl + koffB[ks] + (dxB[ks] == 0 ? hm1 : (dxB[ks] == 1 ? h0 : hp1));
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const bf16x8 av = *reinterpret_cast<const bf16x8*>(hb + m * MTB_BYTES);
#pragma unroll
          for (int t = 0; t < NTB; ++t) accb[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[ks][t], av, accb[m][t], 0, 0, 0);
        }
      }
      if constexpr (RES) {  // the 1x1x1 residual convolution of the input at output plane s-3: the centre-tap K-steps on input plane s-3 (ring slot (s-1) % RING)
        const int sr = ((s - 1) % RING) * PA_BYTES;
#pragma unroll
        for (int ks = 0; ks < KR; ++ks) {
          const char* hb = Rin + koffA[KLO + ks] + sr;
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const bf16x8 av = *reinterpret_cast<const bf16x8*>(hb + m * MTA_BYTES);
#pragma unroll
            for (int t = 0; t < NR; ++t) racc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wres[ks][t], av, racc[m][t], 0, 0, 0);
          }
        }
      }
    }

    // ---- stage A's epilogue -> H slot s & 3 (zeros where the plane lies outside the image: stage B's zero padding)
    if (a_on) {
      char* hdst = Hl + (s & (CH_HR - 1)) * PB_BYTES;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NTA; ++t) {
          const int c = t * 16 + g * 4;
          const float4 bi = *reinterpret_cast<const float4*>(epi + c), sc = *reinterpret_cast<const float4*>(epi + CM + c), sh = *reinterpret_cast<const float4*>(epi + 2 * CM + c);
          float val[4] = {acca[m][t][0] + bi.x, acca[m][t][1] + bi.y, acca[m][t][2] + bi.z, acca[m][t][3] + bi.w};
          if (k.scale_a) { val[0] = val[0] * sc.x + sh.x; val[1] = val[1] * sc.y + sh.y; val[2] = val[2] * sc.z + sh.z; val[3] = val[3] * sc.w + sh.w; }
          if (k.act_a == VSSEG_ACT_PRELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) val[r] = val[r] > 0.f ? val[r] : alpha_a * val[r];
          } else if (k.act_a == VSSEG_ACT_RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) val[r] = fmaxf(val[r], 0.f);
          }
          const uint2 hv = a_in ? make_uint2(f2bf2(val[0], val[1]), f2bf2(val[2], val[3])) : make_uint2(0u, 0u);
          *reinterpret_cast<uint2*>(hdst + hw[t] + m * MTB_BYTES) = hv;
        }
    }

    // ---- stage B's epilogue -> output plane s-3 (x = xb + s - 3)
    if (b_on) {
      const int64_t ovox0 = ocol + (int64_t)(xb + s - 3) * oplane;
      const char* x1p = Rin + ((s - 1) % RING) * PA_BYTES + x1off;  // (compact input) input plane s-3
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int64_t ovox = ovox0 + (int64_t)m * RPM * Z;
        if (out1) {  // one fp32 channel (attention map): lane group 0 holds it
          float v = accb[m][0][0] + epb[0];
          if (k.scale_b) v = v * epb[NTB * 16] + epb[2 * NTB * 16];
          if (k.act_b == VSSEG_ACT_PRELU) v = v > 0.f ? v : alpha_b * v;
          else if (k.act_b == VSSEG_ACT_RELU) v = fmaxf(v, 0.f);
          else if (k.act_b == VSSEG_ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
          if (g == 0) *reinterpret_cast<float*>(k.out + ovox * k.out_vox_bytes) = v;
          continue;
        }
        float x1 = 0.f;
        if constexpr (C1) {
          if (k.in1_w) x1 = __uint_as_float((unsigned)*reinterpret_cast<const unsigned short*>(x1p + m * MTA_BYTES) << 16);
        }
#pragma unroll
        for (int t = 0; t < NTB; ++t) {
          const int c = t * 16 + g * 4;
          if (c >= cout) continue;
          const float4 bi = *reinterpret_cast<const float4*>(epb + c);
          float val[4] = {accb[m][t][0] + bi.x, accb[m][t][1] + bi.y, accb[m][t][2] + bi.z, accb[m][t][3] + bi.w};
          if (k.scale_b) {
            const float4 sc = *reinterpret_cast<const float4*>(epb + NTB * 16 + c), sh = *reinterpret_cast<const float4*>(epb + 2 * NTB * 16 + c);
            val[0] = val[0] * sc.x + sh.x; val[1] = val[1] * sc.y + sh.y; val[2] = val[2] * sc.z + sh.z; val[3] = val[3] * sc.w + sh.w;
          }
          if (k.act_b == VSSEG_ACT_PRELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) val[r] = val[r] > 0.f ? val[r] : alpha_b * val[r];
          } else if (k.act_b == VSSEG_ACT_RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) val[r] = fmaxf(val[r], 0.f);
          } else if (k.act_b == VSSEG_ACT_SIGMOID) {
#pragma unroll
            for (int r = 0; r < 4; ++r) val[r] = 1.f / (1.f + __expf(-val[r]));
          }
          if constexpr (C1) {
            if (k.in1_w) {  // + x * w[c] + b[c]: the 1 -> C residual convolution of the network input, behind the activation
              const float4 w1 = *reinterpret_cast<const float4*>(epb + 3 * NTB * 16 + c), b1 = *reinterpret_cast<const float4*>(epb + 4 * NTB * 16 + c);
              val[0] += x1 * w1.x + b1.x; val[1] += x1 * w1.y + b1.y; val[2] += x1 * w1.z + b1.z; val[3] += x1 * w1.w + b1.w;
            }
          }
          if constexpr (RES) {  // + the residual convolution, rounded to bf16 like the tensor the separate launch stores
            const unsigned r01 = f2bf2(racc[m][t][0] + rb[t][0], racc[m][t][1] + rb[t][1]), r23 = f2bf2(racc[m][t][2] + rb[t][2], racc[m][t][3] + rb[t][3]);
            val[0] += __uint_as_float(r01 << 16); val[1] += __uint_as_float(r01 & 0xffff0000u); val[2] += __uint_as_float(r23 << 16); val[3] += __uint_as_float(r23 & 0xffff0000u);
          }
          char* op = k.out + ovox * k.out_vox_bytes + c * (int)out_es;
          if (vec_store) {
            st4(reinterpret_cast<bf16_t*>(op), make_float4(val[0], val[1], val[2], val[3]));
          } else {  // 1 .. 3 channels
            const int nc = min(4, cout - c);
            for (int r = 0; r < nc; ++r) {
              if (k.out_f32) reinterpret_cast<float*>(op)[r] = val[r];
              else reinterpret_cast<bf16_t*>(op)[r] = f2bf(val[r]);
            }
          }
        }
      }
    }
  };

  const int last = steps + 2;
  for (int s = 0; s <= last; s += 3) {
    iter(s, c0);
    if (s + 1 <= last) iter(s + 1, c1);
    if (s + 2 <= last) iter(s + 2, c2);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the planes fetched behind the segment land before the workgroup's LDS is released
}

// ---- host side ------------------------------------------------------------------------------------------------------
template <int CIN, int CM, int NTB, int TZ, int MT, int CC, int NW, int LEAD, int NR> static int ch_lds() {
  constexpr int GA = CIN / 8, GB = CM / 8, RPM = 16 / TZ, ROWS = NW * MT * RPM + 2;
  constexpr int WIA = (ROWS * TZ * GA + 63) / 64, PB = (ROWS * TZ * GB * 16 + 255) / 256 * 256;
  return (CC ? 5 : 3 + LEAD + (NR ? 1 : 0)) * WIA * 1024 + CH_HR * PB + (3 * CM + 5 * NTB * 16) * 4 + 16;
}
template <int CIN, int CM, int NTB, int TZ, int MT, int CC, int NW, int LEAD, int NR> static int ch_launch(const ChainK& k, int grid, hipStream_t s) {
  static bool init_dev[16] = {}; bool& init = vsseg_dev_once(init_dev);  // per device: the LDS opt-in is a per-device function attribute
  const int lds = ch_lds<CIN, CM, NTB, TZ, MT, CC, NW, LEAD, NR>();
  if (!init) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_kernel<CIN, CM, NTB, TZ, MT, CC, NW, LEAD, NR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    init = true;
  }
  hipLaunchKernelGGL((chain_kernel<CIN, CM, NTB, TZ, MT, CC, NW, LEAD, NR>), dim3((unsigned)grid), dim3(NW * 64), lds, s, k);
  VSSEG_LAUNCH_CHECK("vsseg_conv_chain");
  return VSSEG_OK;
}
typedef int (*ch_fn_t)(const ChainK&, int, hipStream_t);
struct ChEntry { int cin, cm, ntb, tz, mt, cc, nw, lead, nr; ch_fn_t fn; int (*lds)(); };
#define CH_E(CI, CMID, NB, Z, M, C, W, LD) {CI, CMID, NB, Z, M, C, W, LD, 0, ch_launch<CI, CMID, NB, Z, M, C, W, LD, 0>, ch_lds<CI, CMID, NB, Z, M, C, W, LD, 0>}
#define CH_R(CI, CMID, NB, Z, M, W, LD) {CI, CMID, NB, Z, M, 0, W, LD, NB, ch_launch<CI, CMID, NB, Z, M, 0, W, LD, NB>, ch_lds<CI, CMID, NB, Z, M, 0, W, LD, NB>}  // + the residual convolution of the input
// (input channels of A, channels between the stages, output tiles of B, tz, M-tiles per wave, compact input, waves, fetch distance): rows Y = waves * mt * 16 / tz
static const ChEntry ch_table[] = {
    // 1 -> 16 -> 16 (+ residual of the input)
    CH_E(8, 16, 1, 4, 4, 1, 8, 3), CH_E(8, 16, 1, 2, 2, 1, 8, 3), CH_E(8, 16, 1, 4, 2, 1, 8, 3), CH_E(8, 16, 1, 2, 1, 1, 8, 3),  // Y = 128 / 128 / 64 / 64
    CH_E(8, 16, 1, 2, 4, 1, 4, 3), CH_E(8, 16, 1, 4, 8, 1, 4, 3), CH_E(8, 16, 1, 2, 2, 1, 4, 3), CH_E(8, 16, 1, 4, 4, 1, 4, 3),  // ... four waves: Y = 128 / 128 / 64 / 64
    // 32 -> 16 -> 1 (attention block)
    CH_E(32, 16, 1, 2, 2, 0, 8, 3), CH_E(32, 16, 1, 4, 2, 0, 8, 3), CH_E(32, 16, 1, 2, 1, 0, 8, 3),                               // one workgroup per CU: Y = 128 / 64 / 64
    CH_E(32, 16, 1, 1, 1, 0, 8, 1), CH_E(32, 16, 1, 1, 2, 0, 4, 1), CH_E(32, 16, 1, 2, 4, 0, 4, 1), CH_E(32, 16, 1, 2, 2, 0, 8, 1),  // several per CU: Y = 128
    CH_E(32, 16, 1, 1, 1, 0, 4, 1), CH_E(32, 16, 1, 2, 2, 0, 4, 1), CH_E(32, 16, 1, 2, 1, 0, 8, 1),                               // ... Y = 64
    CH_E(32, 16, 1, 2, 1, 0, 16, 1), CH_E(32, 16, 1, 2, 1, 0, 16, 3), CH_E(32, 16, 1, 4, 2, 0, 16, 1), CH_E(8, 16, 1, 2, 1, 1, 16, 3), CH_E(8, 16, 1, 4, 2, 1, 16, 3),  // sixteen waves: Y = 128
    // 16 -> 32 -> 32 + the 1x1x1 residual convolution 16 -> 32 of the input (the two-sub-unit ResidualUnit of level 1)
    CH_R(16, 32, 2, 2, 1, 8, 1), CH_R(16, 32, 2, 4, 2, 8, 1), CH_R(16, 32, 2, 2, 2, 8, 1)};  // Y = 64 / 64 / 128

static const ChEntry* ch_find(const vsseg_chain_desc* d, const char** why) {
  *why = nullptr;
  auto no = [&](const char* w) { *why = w; return (const ChEntry*)nullptr; };
  if (!d->in.ptr || !d->out.ptr || !d->wpack_a || !d->wpack_b) return no("null pointer");
  if (d->in.dtype != VSSEG_BF16) return no("input is not bf16");
  const bool c1 = d->in.c == 1 && d->in.pitch == 1 && !d->in.ptr2;
  const int cin = c1 ? 8 : d->in.c;
  if (!c1 && (d->in.c % 8 || d->in.pitch % 8 || ((uintptr_t)d->in.ptr & 15) || ((uintptr_t)d->in.ptr2 & 15) || (d->in.ptr2 && d->in.csplit % 8))) return no("input must be 16-byte aligned voxel rows of 8-channel groups (or a compact one-channel tensor)");
  if (c1 && ((uintptr_t)d->in.ptr & 1)) return no("unaligned compact input");
  if (d->in.n != d->out.n || d->in.x != d->out.x || d->in.y != d->out.y || d->in.z != d->out.z || d->out.ptr2) return no("input and output extents differ / two-part output");
  if (d->out.dtype != VSSEG_BF16 && d->out.dtype != VSSEG_F32) return no("output dtype");
  if (d->out.c < 1 || d->out.c > 32 || d->out.pitch < d->out.c) return no("output channels");
  const int ntb = (d->out.c + 15) / 16;
  if (d->res_tiles && (d->res_tiles != ntb || c1 || !d->wpack_res || (d->out.c & 3) || d->in1_w)) return no("residual tiles: one per output tile, packed weights, an ordinary input, a bf16 output");
  if ((d->out.c & 3) == 0 && (d->out.dtype != VSSEG_BF16 || (d->out.pitch & 3) || ((uintptr_t)d->out.ptr & 7))) return no("a 4k-channel output is bf16 with 8-byte aligned rows");
  if ((d->in1_w || d->in1_b) && (!c1 || !d->in1_w || !d->in1_b || (d->out.c & 3))) return no("the residual of the input needs a compact one-channel input, weights and bias, and a bf16 output");
  if ((d->scale_a == nullptr) != (d->shift_a == nullptr) || (d->scale_b == nullptr) != (d->shift_b == nullptr)) return no("scale without shift");
  if (d->act_b < VSSEG_ACT_NONE || d->act_b > VSSEG_ACT_SIGMOID) return no("act_b");
  if (d->act_a != VSSEG_ACT_NONE && d->act_a != VSSEG_ACT_PRELU && d->act_a != VSSEG_ACT_RELU) return no("act_a");
  if ((d->tz != 1 && d->tz != 2 && d->tz != 4 && d->tz != 8) || d->mtw < 1 || (d->waves != 4 && d->waves != 8 && d->waves != 16) || d->in.y != d->waves * d->mtw * 16 / d->tz || d->in.z % d->tz || d->lx < 1)
    return no("plan: 4, 8 or 16 waves, y must be waves * mtw * 16 / tz rows, z a multiple of tz, lx >= 1");
  for (const ChEntry& e : ch_table)
    if (e.cin == cin && e.cm == d->cmid && e.ntb == ntb && e.nr == d->res_tiles && e.tz == d->tz && e.mt == d->mtw && e.cc == (c1 ? 1 : 0) && e.nw == d->waves && e.lead == d->lead) return e.lds() <= 160 * 1024 ? &e : no("more than 160 KB of LDS");
  return no("no instantiation for this (channels, tz, mtw, waves, lead)");
}

extern "C" int vsseg_conv_chain_lds_bytes(const vsseg_chain_desc* d) {
  VSSEG_CHECK(d, "vsseg_conv_chain: null descriptor");
  const char* why;
  const ChEntry* e = ch_find(d, &why);
  if (!e) { vsseg_set_error("vsseg_conv_chain: not applicable: %s", why); return VSSEG_EINVAL; }
  return e->lds();
}

extern "C" int vsseg_conv_chain(const vsseg_chain_desc* d, void* stream) {
  VSSEG_CHECK(d, "vsseg_conv_chain: null descriptor");
  const char* why;
  const ChEntry* e = ch_find(d, &why);
  if (!e) { vsseg_set_error("vsseg_conv_chain: not applicable: %s", why); return VSSEG_EINVAL; }
  ChainK k;
  k.in0 = reinterpret_cast<const char*>(d->in.ptr);
  k.in1 = d->in.ptr2 ? reinterpret_cast<const char*>(d->in.ptr2) - (int64_t)d->in.csplit * 2 : k.in0;
  k.in_csplit_pc = d->in.ptr2 ? d->in.csplit / 8 : 1 << 20;
  k.in_vox_bytes = d->in.pitch * 2;
  const int oes = d->out.dtype == VSSEG_F32 ? 4 : 2;
  k.out = reinterpret_cast<char*>(d->out.ptr);
  k.out_vox_bytes = d->out.pitch * oes;
  k.out_f32 = d->out.dtype == VSSEG_F32;
  k.cout = d->out.c;
  k.wa = reinterpret_cast<const char*>(d->wpack_a); k.wb = reinterpret_cast<const char*>(d->wpack_b);
  k.bias_a = d->bias_a; k.scale_a = d->scale_a; k.shift_a = d->shift_a; k.alpha_a = d->alpha_a;
  k.bias_b = d->bias_b; k.scale_b = d->scale_b; k.shift_b = d->shift_b; k.alpha_b = d->alpha_b;
  k.act_a = d->act_a; k.act_b = d->act_b;
  k.in1_w = d->in1_w; k.in1_b = d->in1_b;
  k.wres = reinterpret_cast<const char*>(d->wpack_res); k.bias_r = d->bias_res;
  k.zeros = vsseg_zero_page();
  VSSEG_CHECK(k.zeros, "vsseg_conv_chain: could not allocate the zero page");
  k.X = d->in.x; k.Y = d->in.y; k.Z = d->in.z;
  k.lx = d->lx > k.X ? k.X : d->lx;
  k.nxs = (k.X + k.lx - 1) / k.lx; k.nzb = k.Z / d->tz;
  const int64_t grid = (int64_t)d->in.n * k.nxs * k.nzb;
  VSSEG_CHECK(grid > 0 && grid < (1ll << 30), "vsseg_conv_chain: bad grid");
  return e->fn(k, (int)grid, as_stream(stream));
}
