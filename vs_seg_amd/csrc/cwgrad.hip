// Weight gradient of the MFMA-bound stride-1 3x3x3 convolutions (levels 2-3 of the 2.5D U-Net: 96x32x128 and 48x16x64 voxels; SURVEY §8a rows 9, 10, 13, 14,
// 31, 34, 36, 39 — autograd of ref:params/networks/blocks/convolutions.py:137-146 as run by loss.backward(), ref:params/VSparams.py:461) with a compile-time
// geometry: vsseg_wgrad with march = 2.
//
//   dW[(dx, dy, dz)][cP][cH] = sum_q P[q][cP] * H[q + (dx, dy, dz)][cH]          P = dY (48 or 64 channels), H = X (a multiple of 16 channels; may be two-part)
//
// What held wgrad_kernel (wgrad.hip) at 0.19-0.28 of the bf16 MFMA peak on these layers in the training step: every H fragment it read from LDS fed only NTP
// MFMAs (10 fragment reads per 21 MFMAs), 256-voxel tiles with a load -> wait -> multiply round each, a (4+2)x(8+2)x(8+2) halo per 4x8x8 tile (2.3x the
// tile's voxels through the LDS-DMA path, which delivers ~5 TB/s chip-wide whatever the L2 hit rate).  Here:
//
//   * THE THREE z-TAPS OF A VOXEL ROW SHARE ONE H FRAGMENT.  The reduction axis (voxels) runs along z, 32 voxels per MFMA K-step.  Shifting the P operand
//     instead of the H operand —  dW[dz] = sum_z' P[z' - dz] * H[z'] — one H fragment (32 voxels x 16 channels at column (x + dx, y + dy)) is multiplied
//     with the three z-shifted P fragments of every P tile, and the shifted P fragments (3 x NTP, held in registers) serve all nine (dx, dy) taps of the
//     K-step: 9 + 9 fragment reads for 81 MFMAs (NTP = 3).  Work items therefore partition H along z (no z halo in H) and P along x, y; P carries a z
//     halo of one voxel (zero outside the volume)
//   * one wave per SIMD owns ALL 27 taps of its 16-channel chunk of H: 27 x NPW accumulator tiles (324 / 216 registers) live in registers over the whole
//     launch, the fragment reads of tap t + 2 are issued in front of the MFMAs of tap t, the P fragments of the next K-step during the last three taps of
//     the current one; nothing inside a step waits on memory
//   * MARCH ALONG x: a workgroup owns a strip (8 rows of y x 32 voxels of z) and walks x.  A plane of H — (8 + 2) columns of 32 voxels x 16 channels =
//     10 KiB per chunk, each column exactly one 1 KiB LDS-DMA row ([z][32 bytes]: the transpose reads ds_read_b64_tr_b16 of 8 consecutive voxels hit 8
//     different 32-byte bank groups) — is fetched ONCE into a ring of four planes and multiplied in three consecutive steps (dx = +1, 0, -1); the plane
//     of P (8 columns x 34 rows of 96 bytes, or 160 for 64 channels: an odd number of 32-byte groups, conflict-free for every z shift) is double
//     buffered.  Per step 46 KiB arrive (1.25x the H bytes, 1.06x the P bytes of the step) against 74 KiB for a 2x4x32 tile with its halo (the first
//     version of this kernel: LDS-DMA bound at 0.44 ms on the 96 -> 48 layer, as fast as wgrad_kernel).  The DMAs of the next step are issued one per
//     tap inside the MFMA stream; one `s_waitcnt vmcnt(0)` + one s_barrier per step
//   * the four waves of a workgroup are (P half) x (H chunk) x (K-step share); workgroups of chunk class c own the chunks [c*CG, (c+1)*CG); the classes of
//     a strip run on the same XCD at about the same time (P comes from that L2); every workgroup walks a contiguous run of (strip, x) steps
//   * partial sums leave as slabs in wgrad_kernel's layout and are summed in a fixed order by its reduce launch (run-to-run bit-identical)
#include "common.h"
#include <type_traits>
#include <utility>

constexpr int CW_TY = 8, CW_TZ = 32;
constexpr int CW_HY = CW_TY + 2;         // halo columns of a plane, 1 KiB each per 16-channel chunk
constexpr int CW_RING = 4;               // planes x - 1, x, x + 1 of the current step + plane x + 2 in flight
constexpr int CW_PROWS = CW_TZ + 2;      // P column with its z halo
constexpr int cw_prow(int ntp) { return (ntp & 1) ? ntp * 32 : ntp * 32 + 32; }  // bytes per P voxel row in LDS: an odd number of 32-byte groups
constexpr int cw_nhd(int cg) { return (cg * CW_HY + 3) / 4; }  // H columns per wave and plane
constexpr int cw_lds_bytes(int ntp, int cg) { return CW_RING * cw_nhd(cg) * 4 * 1024 + 2 * CW_TY * CW_PROWS * cw_prow(ntp); }

struct CwK {
  const char* p;
  const char* h0;
  const char* h1;       // part 1 of a two-part H, biased by -csplit channels (== h0 for an ordinary tensor)
  int h_split_chunk;    // first 16-channel chunk that lives in part 1
  int p_vox_bytes, h_vox_bytes;
  int X, Y, Z, nty, ntz;
  unsigned mg_x, mg_ty, mg_tz;
  int ncls, pow2;       // chunk classes; class map (see cw_class)
  int hchunks, slab_chunk;
  const char* zeros;    // >= 16 KiB of zeros
  float* slab;
  float* bias_slab;
  int fstart[8][8];     // first (strip, x) step of the (XCD, class) group
  int fcnt[8][8];       // its step count
  short gsz[8][8];      // its workgroups
};

__device__ __forceinline__ unsigned cw_div(unsigned n, unsigned magic) { return magic ? __umulhi(n, magic) : n; }
// chunk class of workgroup L (runs on XCD L % 8): with a power-of-two class count the classes alternate inside an XCD's workgroups; otherwise
// L % ncls, which spreads every class over all XCDs when ncls is odd
__host__ __device__ inline int cw_class(int L, int ncls, int pow2) { return pow2 ? (L >> 3) % ncls : L % ncls; }
__host__ __device__ inline int cw_widx(int L, int ncls, int pow2) { return pow2 ? ((L >> 3) / ncls) * 8 + (L & 7) : L / ncls; }

typedef __attribute__((address_space(3))) bf16x4 cw_lds_b4;
// LDS addresses are carried as 32-bit integers: a generic pointer that went through a select makes hipcc (ROCm 7.2) emit the generic -> LDS null check as
// `v_cmp_ne_u32 0, src_shared_base`, which its own verifier rejects ("Operand has incorrect register class")
__device__ __forceinline__ bf16x8 cw_frag(unsigned a, int second) {  // two transpose reads: voxels 4g .. 4g+3 and 16 + 4g .. of the lane's channel
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((cw_lds_b4*)(uintptr_t)(a));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((cw_lds_b4*)(uintptr_t)(a + (unsigned)second));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// Compile-time loops: the accumulator index decides the register class of an MFMA below, so it must be a constant expression (not just unrollable).
template <typename F, int... I> __device__ __forceinline__ void cw_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void cw_for(F&& f) { cw_for_impl(f, std::make_integer_sequence<int, N>{}); }
// One MFMA with its accumulator pinned to the accumulation registers (AG) or to the ordinary vector registers.  81 accumulator tiles (27 taps x 48 P
// channels) are 324 registers: more than either class holds (256 + 256 in one 512-entry file).  Left to itself hipcc keeps all of them in the accumulation
// class and, out of room, multiplies into a temporary and copies every result out through v_accvgpr_read behind an `s_nop 7` (seen in the ISA: the matrix pipe
// idles 4 of 5 cycles).  With the class chosen per tile — the first 64 tiles "a", the rest "v" — nothing moves.  hipcc pads no hazards around inline assembly:
// the operands come from LDS reads (ordered by the compiler's s_waitcnt), MFMAs on the same accumulator are 27 x NPW instructions apart, and the flush waits
// (cw_mfma_drain) before it reads the results.
// PAD: two wait states in front (VALU write of an operand register -> MFMA read needs two; hipcc may rematerialise or copy an operand with a VALU move
// right in front of the statement — it did for the all-ones operand of the bias MFMAs, whose first one then read stale registers); used on the first
// MFMA behind every scheduling barrier, where such a move can only sit.  Hidden behind the previous MFMA's 16 cycles.
template <bool AG, bool PAD> __device__ __forceinline__ void cw_mfma(f32x4& c, const bf16x8& a, const bf16x8& b) {
  if constexpr (AG && PAD) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else if constexpr (AG) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else if constexpr (PAD) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void cw_mfma_drain() { asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory"); }

// NPW: 16-channel P tiles per wave; PS: waves a K-step's P tiles are split over; CG: 16-channel H chunks per workgroup (one per wave); the K-steps of a
// plane are shared by KS = 4 / (PS * CG) waves
template <int NPW, int PS, int CG, bool BIAS>
__global__ __launch_bounds__(256, 1) void cwgrad_kernel(const CwK k) {
  constexpr int KS = 4 / (PS * CG), NTP = NPW * PS, PROW = cw_prow(NTP), PREAL = NTP * 2;
  constexpr int PCOL = CW_PROWS * PROW, NHD = cw_nhd(CG), HPLANE = NHD * 4 * 1024, HBYTES = CW_RING * HPLANE, PBUF = CW_TY * PCOL;
  constexpr int NPI = CW_TY * CW_TZ * PREAL / 256, NPD = NPI + 1, NK = CW_TY / KS, NDMA = NHD + NPD, NTAP = NK * 9, PHALO = CW_TY * PREAL;
  static_assert(KS >= 1 && KS * PS * CG == 4 && CW_TY % KS == 0 && NDMA <= NTAP && NHD <= 6 && (CW_TY * CW_TZ * PREAL) % 256 == 0 && PHALO <= 64, "wave roles");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned sm0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, l15 = lane & 15;
  const int ps = wave % PS, cg = (wave / PS) % CG, ks = wave / (PS * CG);
  const int X = k.X, Y = k.Y, Z = k.Z;

  // ---- which steps: workgroup L runs on XCD L % 8; group (XCD, class) owns a contiguous range of the (strip, x) list, its workgroups contiguous parts ----
  const int L = blockIdx.x, xcd = L & 7, cls = cw_class(L, k.ncls, k.pow2), widx = cw_widx(L, k.ncls, k.pow2);
  int rank = 0;
  for (int l2 = xcd; l2 < L; l2 += 8) rank += cw_class(l2, k.ncls, k.pow2) == cls;
  const int gsz = k.gsz[xcd][cls];
  const int64_t gcnt = k.fcnt[xcd][cls];
  const int f_lo = k.fstart[xcd][cls] + (int)(gcnt * rank / gsz), f_hi = k.fstart[xcd][cls] + (int)(gcnt * (rank + 1) / gsz);

  // ---- operand pieces (16 bytes per lane), NDMA per wave and step.  REGISTER PATH: a global load now, a ds_write_b128 a whole step later.  The first
  //      versions used LDS-DMA (global_load_lds): each such instruction held its wave for ~85 cycles of a CU-wide serial resource — 48 per step, issued
  //      from the MFMA stream, cost their full 0.12 ms on top of the 0.31 ms of the MFMAs (measured: H only + 0.06, P only + 0.06, neither 0.31): the MFMA
  //      wave IS the loader here, there is no second wave per SIMD to take the stall.  And every instruction between two MFMA groups is a bubble for the
  //      one wave of the SIMD: a piece costs a scalar select + one load, and one vector add + one store, placed between the MFMAs of a tap.
  //   H piece u < NHD: column q = u*4 + wave of the plane's CG x 10 columns; lane -> (voxel z = lane >> 1, 16-byte half of the chunk): scalar base + lane offset
  const unsigned h_lane = (unsigned)(lane >> 1) * (unsigned)k.h_vox_bytes + (unsigned)(lane & 1) * 16u;
  const unsigned h_st = sm0 + (unsigned)(wave * 1024 + lane * 16);  // + slot * HPLANE + u * 4096
  int hconst[NHD];      // byte offset of this wave's column u inside a plane of its H part: hy * Z * vox + chunk * 32
  unsigned hbits = 0;   // per u: bits [4u, 4u+4) = hy (0..9); bit 24 + u: the column's chunk lives in part 1
  const int hys = Z * k.h_vox_bytes;
#pragma unroll
  for (int u = 0; u < NHD; ++u) {
    const int q = u * 4 + wave, c = q / CW_HY, hy = q - c * CW_HY, chunk = cls * CG + c;
    hconst[u] = __builtin_amdgcn_readfirstlane(hy * hys + chunk * 32);
    hbits |= (unsigned)hy << (4 * u);
    hbits |= (chunk >= k.h_split_chunk ? 1u : 0u) << (24 + u);
  }
  hbits = __builtin_amdgcn_readfirstlane(hbits);
  //   P piece v < NPI: piece (v*4 + wave)*64 + lane of the plane's interior [column][row 1..32][real 16-byte pieces]: no condition at all;
  //   v == NPI: the z halo — wave 0: row 0 (voxel z0 - 1), wave 1: row 33 (z0 + 32) of every column, zeros at the volume's z borders (wave-uniform)
  unsigned prel[NPD], plds[NPD];  // byte offset inside the P plane in memory (from voxel (y0, z0 - 1)) / inside the LDS buffer
#pragma unroll
  for (int v = 0; v < NPD; ++v) {
    int c, row, pc;
    if (v < NPI) {
      const int j = (v * 4 + wave) * 64 + lane;
      c = j / (CW_TZ * PREAL);
      const int rem = j - c * (CW_TZ * PREAL);
      row = 1 + rem / PREAL;
      pc = rem % PREAL;
    } else {
      const int j = lane < PHALO ? lane : PHALO - 1;
      c = j / PREAL;
      pc = j % PREAL;
      row = wave == 0 ? 0 : CW_PROWS - 1;
    }
    prel[v] = (unsigned)((c * Z + row) * k.p_vox_bytes + pc * 16);
    plds[v] = sm0 + (unsigned)(HBYTES + c * PCOL + row * PROW + pc * 16);
  }
  const int64_t hxs = (int64_t)Y * hys, pxs = (int64_t)Y * Z * k.p_vox_bytes;  // bytes per x plane of H / P
  // run state (wave-uniform): plane 0 of the strip, biased to row y0 - 1 (H) / row y0 (P) and voxel z0 (H) / z0 - 1 (P)
  const char *hrun0 = nullptr, *hrun1 = nullptr, *prun = nullptr;
  unsigned colmask = 0, s_edge = 0;  // valid rows of the halo; z borders of the volume
  typedef unsigned cw_u32x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) cw_u32x4 cw_lds_u32x4;
  // piece u of H plane xh (u < NHD) / of P plane xp (clamped into the volume by the caller: a plane nobody will read is still loaded, from valid memory)
  auto ld = [&](auto uc, int xh, int xp, cw_u32x4& dst) {
    constexpr int u = decltype(uc)::value;
    if constexpr (u < NHD) {
      if (CG * CW_HY % 4 != 0 && u * 4 + wave >= CG * CW_HY) return;  // wave-uniform: this wave has no column u
      const bool ok = ((unsigned)xh < (unsigned)X) & (((colmask >> ((hbits >> (4 * u)) & 15u)) & 1u) != 0);  // the zero padding of the convolution comes from the zero page
      const char* base = (((hbits >> (24 + u)) & 1u) ? hrun1 : hrun0) + xh * hxs + hconst[u];
      dst = *reinterpret_cast<const cw_u32x4*>((ok ? base : k.zeros) + h_lane);
    } else if constexpr (u < NHD + NPI) {
      dst = *reinterpret_cast<const cw_u32x4*>(prun + xp * pxs + prel[u - NHD]);
    } else {
      if (wave >= 2) return;  // wave-uniform
      const bool zero = (s_edge >> wave) & 1u;
      const char* base = zero ? k.zeros : prun + xp * pxs;
      if (lane < PHALO) dst = *reinterpret_cast<const cw_u32x4*>(base + (zero ? (unsigned)lane * 16u : prel[NPI]));
    }
  };
  auto st = [&](auto uc, int xh, int xp, const cw_u32x4& val) {
    constexpr int u = decltype(uc)::value;
    if constexpr (u < NHD) {
      if (CG * CW_HY % 4 != 0 && u * 4 + wave >= CG * CW_HY) return;
      *(cw_lds_u32x4*)(uintptr_t)(h_st + (unsigned)((xh & 3) * HPLANE + u * 4096)) = val;
    } else if constexpr (u < NHD + NPI) {
      *(cw_lds_u32x4*)(uintptr_t)(plds[u - NHD] + (unsigned)((xp & 1) * PBUF)) = val;
    } else {
      if (wave >= 2) return;
      if (lane < PHALO) *(cw_lds_u32x4*)(uintptr_t)(plds[NPI] + (unsigned)((xp & 1) * PBUF)) = val;
    }
  };

  constexpr int NACC = 27 * NPW, NAG = 64;  // accumulator tiles; the first NAG live in accumulation registers (cw_mfma)
  constexpr bool ASM = NACC > NAG;
  f32x4 acc[NACC];  // [tap t = dx*3 + dy][P shift s][P tile p]
  cw_for<NACC>([&](auto ic) { acc[decltype(ic)::value] = f32x4{0.f, 0.f, 0.f, 0.f}; });
  f32x4 accb[BIAS ? NPW : 1];
#pragma unroll
  for (int p = 0; p < (BIAS ? NPW : 1); ++p) accb[p] = f32x4{0.f, 0.f, 0.f, 0.f};
  // bias gradient dbias[cP] = sum_q P[q][cP]: one more MFMA per K-step and P tile against an all-ones operand (zeros in the waves that do not own it: no branch)
  const short one = (BIAS && cls == 0 && cg == 0) ? (short)0x3F80 : (short)0;
  const bf16x8 ones = bf16x8{one, one, one, one, one, one, one, one};

  // operand addressing: lane (g, l15) reads row 4g + (l15 >> 2) of a 4-voxel block, 4-channel group l15 & 3 (ds_read_b64_tr_b16 returns the lane's channel of
  // the four voxels); K-slot g*8 + j holds voxel 4g + j (j < 4) / 16 + 4g + (j - 4) of the K-step for both operands
  const int r4 = g * 4 + (l15 >> 2), qc = (l15 & 3) * 8;
  const int h_lane_off = (cg * CW_HY + ks) * 1024 + r4 * 32 + qc;
  const int p_lane_off = HBYTES + ks * PCOL + r4 * PROW + ps * NPW * 32 + qc;
  cw_u32x4 stage[NDMA];  // pieces on their way through the registers
  cw_for<NDMA>([&](auto uc) { stage[decltype(uc)::value] = cw_u32x4{0u, 0u, 0u, 0u}; });

  int f = f_lo;
  while (f < f_hi) {
    // ---- a run: consecutive x of one strip ----
    unsigned b = (unsigned)f, qd = cw_div(b, k.mg_x);
    const int x0 = (int)(b - qd * (unsigned)X);
    b = qd;
    qd = cw_div(b, k.mg_ty);
    const int y0 = (int)(b - qd * k.nty) * CW_TY;
    b = qd;
    qd = cw_div(b, k.mg_tz);
    const int z0 = (int)(b - qd * k.ntz) * CW_TZ, n = (int)qd;
    s_edge = (z0 == 0 ? 1u : 0u) | (z0 + CW_TZ == Z ? 2u : 0u);
    colmask = 0;
#pragma unroll
    for (int hy = 0; hy < CW_HY; ++hy) colmask |= ((unsigned)(y0 - 1 + hy) < (unsigned)Y ? 1u : 0u) << hy;
    hrun0 = k.h0 + ((((int64_t)n * X) * Y + (y0 - 1)) * Z + z0) * k.h_vox_bytes;
    hrun1 = k.h1 + ((((int64_t)n * X) * Y + (y0 - 1)) * Z + z0) * k.h_vox_bytes;
    prun = k.p + ((((int64_t)n * X) * Y + y0) * Z + (z0 - 1)) * k.p_vox_bytes;
    int nsteps = X - x0;
    if (nsteps > f_hi - f) nsteps = f_hi - f;
    f += nsteps;
    const int xlast = X - 1;
    __syncthreads();  // every wave is done with the previous run's planes
    // prime through the same registers, two round trips: P plane x0 + H plane x0 - 1, then H planes x0 and x0 + 1 (the second in the P pieces' registers)
    cw_for<NDMA>([&](auto uc) { ld(uc, x0 - 1, x0, stage[decltype(uc)::value]); });
    cw_for<NDMA>([&](auto uc) { st(uc, x0 - 1, x0, stage[decltype(uc)::value]); });
    static_assert(NPD >= NHD, "prime");
    cw_for<NHD>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      ld(uc, x0, 0, stage[u]);
      ld(uc, x0 + 1, 0, stage[NHD + u]);
    });
    cw_for<NHD>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      st(uc, x0, 0, stage[u]);
      st(uc, x0 + 1, 0, stage[NHD + u]);
    });
    // the pieces of step x0 + 1 (H plane x0 + 2, P plane x0 + 1) start their way through the registers: a load has a whole step to land
    cw_for<NDMA>([&](auto uc) { ld(uc, x0 + 2, x0 + 1 < xlast ? x0 + 1 : xlast, stage[decltype(uc)::value]); });
    for (int sti = 0; sti < nsteps; ++sti) {
      const int x = x0 + sti;
      const int xp2 = x + 2 < xlast ? x + 2 : xlast;  // P plane of the step after the next, clamped into the volume (scalar)
      __syncthreads();  // planes x - 1 .. x + 1 of H and plane x of P are in LDS for every wave; every wave is done with step x - 1
      unsigned Hs[3] = {sm0 + (unsigned)(((x - 1) & 3) * HPLANE + h_lane_off), sm0 + (unsigned)((x & 3) * HPLANE + h_lane_off), sm0 + (unsigned)(((x + 1) & 3) * HPLANE + h_lane_off)};
      asm volatile("" : "+v"(Hs[0]), "+v"(Hs[1]), "+v"(Hs[2]));  // three address registers per step, immediates per read (hipcc otherwise adds a scalar slot offset in front of every read)
      const unsigned Ps = sm0 + (unsigned)((x & 1) * PBUF + p_lane_off);
      bf16x8 pa[3][NPW], hb[3];  // P fragments of the K-step (refilled for the next one behind their last MFMA); H fragments: a ring, read two taps ahead
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int p = 0; p < NPW; ++p) pa[s][p] = cw_frag(Ps + s * PROW + p * 32, 16 * PROW);
      // H fragment of the j-th (K-step i, tap t) pair of the step: plane dx = t / 3, column i*KS (+ ks) + dy
      auto h_frag = [&](auto jc) { constexpr int j = decltype(jc)::value, i = j / 9, t = j % 9; return cw_frag(Hs[t / 3] + (i * KS + t % 3) * 1024, 512); };
      hb[0] = h_frag(std::integral_constant<int, 0>{});
      hb[1] = h_frag(std::integral_constant<int, 1>{});
      __builtin_amdgcn_sched_barrier(0);
      cw_for<NTAP>([&](auto jc) {
        constexpr int j = decltype(jc)::value, i = j / 9, t = j % 9;
        constexpr int NQ = 3 * NPW, Q_ST = NQ / 3, Q_LD = 2 * NQ / 3;  // MFMAs of the tap; the piece's store / load sit behind MFMA Q_ST - 1 / Q_LD - 1
        if constexpr (j + 2 < NTAP) hb[(j + 2) % 3] = h_frag(std::integral_constant<int, (j + 2 < NTAP ? j + 2 : 0)>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (BIAS && t == 4) {  // the centre tap's iteration carries the bias MFMAs (shift 0 = the plane's own voxels, each exactly once)
#pragma unroll
          for (int p = 0; p < NPW; ++p) {
            if constexpr (ASM) cw_mfma<false, true>(accb[p], pa[1][p], ones);
            else accb[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[1][p], ones, accb[p], 0, 0, 0);
          }
        }
        cw_for<NQ>([&](auto qc_) {
          constexpr int q = decltype(qc_)::value, a = t * NQ + q;
          bf16x8& pa_q = pa[q / NPW][q % NPW];
          const bf16x8& hb_j = hb[j % 3];
          // operand pieces, one per tap in the first taps: the piece loaded during the previous step (H plane x + 2 / P plane x + 1, read from the next step
          // on) goes to LDS, the same registers then take the piece of the step after (H plane x + 3 / P plane x + 2)
          if constexpr (j < NDMA && (q == Q_ST || q == Q_LD)) {
            constexpr auto uc = std::integral_constant<int, (j < NDMA ? j : 0)>{};
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (q == Q_ST) st(uc, x + 2, x + 1, stage[uc.value]);
            else ld(uc, x + 3, xp2, stage[uc.value]);
            __builtin_amdgcn_sched_barrier(0);
          }
          if constexpr (ASM) cw_mfma<(a < NAG), (q == 0 || (j < NDMA && (q == Q_ST || q == Q_LD)))>(acc[a], pa_q, hb_j);
          else acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa_q, hb_j, acc[a], 0, 0, 0);
          if constexpr (t == 8 && i + 1 < NK) {  // last tap: this P fragment is dead — refill it for the next K-step (used 3 * NPW MFMAs from now)
            __builtin_amdgcn_sched_barrier(0);
            pa_q = cw_frag(Ps + (i + 1) * KS * PCOL + (q / NPW) * PROW + (q % NPW) * 32, 16 * PROW);
            __builtin_amdgcn_sched_barrier(0);
          }
        });
        __builtin_amdgcn_sched_barrier(0);
      });
    }
  }
  if constexpr (ASM) cw_mfma_drain();

  // ---- flush: every tile leaves as one coalesced 1 KiB store (a lane's four values = 16 bytes), one slab per (workgroup, K share).  The first version stored 4-byte
  //      values into wgrad_kernel's [tap][cP][16] layout: 64-byte fragments, 0.06 ms of a 0.42 ms launch.  (Adding the K shares of a workgroup up through LDS first
  //      halves the slab bytes but made hipcc spill accumulators to scratch memory around the exchange — and a kernel with a scratch segment pays for it at EVERY
  //      launch: in bench.py's process the four launches of a step took 6.8 ms instead of 1.2.  No scratch, twice the slabs.)
  // slab of (workgroup, K share): [chunk][tap][P tile][64 lanes][4]; lane (g, l15), value r = element (cP = tile*16 + g*4 + r, cH = chunk*16 + l15)
  const int chunk = cls * CG + cg;
  float* slab = k.slab + (((int64_t)widx * KS + ks) * k.hchunks + chunk) * k.slab_chunk + lane * 4;
  cw_for<NACC>([&](auto ic) {
    constexpr int a = decltype(ic)::value, t = a / (3 * NPW), sft = (a / NPW) % 3, p = a % NPW;
    constexpr int tap = t * 3 + (2 - sft);  // P shifted by sft - 1 pairs P[z' - dz] with H[z']: dz = 1 - sft
    *reinterpret_cast<f32x4*>(slab + (tap * NTP + ps * NPW + p) * 256) = acc[a];
  });
  if constexpr (BIAS) {
    if (cls == 0 && cg == 0 && l15 == 0) {
      float* brow = k.bias_slab + ((int64_t)widx * KS + ks) * (NTP * 16);
#pragma unroll
      for (int p = 0; p < NPW; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) brow[(ps * NPW + p) * 16 + g * 4 + r] = accb[p][r];
    }
  }
}

int vsseg_wgrad_reduce_launch(const vsseg_wgrad_desc* d, float* slab, int nblk, int hchunks, int slab_chunk, hipStream_t s);  // wgrad.hip: sums slabs of this layout

// ---- host side ------------------------------------------------------------------------------------------------------
constexpr int CW_ZERO_BYTES = 16384;
static const char* cw_check(const vsseg_wgrad_desc* d) {
  if (d->p.dtype != VSSEG_BF16 || d->h.dtype != VSSEG_BF16) return "operands are not bf16";
  if (d->ntaps != 27) return "3x3x3 taps only";
  for (int t = 0; t < 27; ++t)
    if (d->tap_off[t][0] != t / 9 - 1 || d->tap_off[t][1] != (t / 3) % 3 - 1 || d->tap_off[t][2] != t % 3 - 1) return "taps are not the 3x3x3 stencil in (x, y, z) order";
  for (int a = 0; a < 3; ++a)
    if (d->hs[a] != 1) return "stride-1 lattices only";
  if (d->q[0] != d->p.x || d->q[1] != d->p.y || d->q[2] != d->p.z || d->q[0] != d->h.x || d->q[1] != d->h.y || d->q[2] != d->h.z || d->p.n != d->h.n) return "lattice, P and H extents differ";
  if (d->q[1] % CW_TY || d->q[2] % CW_TZ) return "y / z extent is not a multiple of the 8 x 32 strip";
  if (d->ntp != 3 && d->ntp != 4) return "P must have 48 or 64 channels";
  if (d->p.c != d->ntp * 16 || d->cp_valid != d->p.c || d->p.ptr2) return "P channels must be ntp x 16, one part";
  if (d->p.pitch % 8 || ((uintptr_t)d->p.ptr & 15)) return "P must be 16-byte aligned voxel rows";
  if (d->h.c % 16 || d->ch_valid != d->h.c) return "H channels must be a multiple of 16";
  if (d->h.pitch % 8 || ((uintptr_t)d->h.ptr & 15) || ((uintptr_t)d->h.ptr2 & 15)) return "H must be 16-byte aligned voxel rows";
  if (d->h.ptr2 && (d->h.csplit % 16 || d->h.csplit <= 0 || d->h.csplit >= d->h.c)) return "H split must be a multiple of 16 channels";
  if (d->h_gate) return "a gated H operand needs the marching kernel";
  const int cg = d->hgroup < 1 ? 1 : d->hgroup;
  if (cg != 1 && cg != 2) return "hgroup (H chunks per workgroup) must be 1 or 2";
  if (d->ntp == 4 && cg != 1) return "64 P channels: hgroup must be 1";
  if ((d->h.c / 16) % cg) return "hgroup must divide the number of 16-channel H chunks";
  if ((d->h.c / 16) / cg > 8) return "more than 8 chunk classes";
  if ((int64_t)d->p.n * d->q[0] * d->q[1] * d->q[2] / (CW_TY * CW_TZ) >= (1ll << 24)) return "too many steps";
  if (d->h.pitch * 2 * 31 + 16 > CW_ZERO_BYTES) return "H pitch too large for the zero page";
  return nullptr;
}

template <int NPW, int PS, int CG, bool BIAS> static int cw_launch_inst(const CwK& k, int grid, hipStream_t s) {
  static bool attr_set[16] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 16 && !attr_set[dev]) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&cwgrad_kernel<NPW, PS, CG, BIAS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL((cwgrad_kernel<NPW, PS, CG, BIAS>), dim3((unsigned)grid), dim3(256), cw_lds_bytes(NPW * PS, CG), s, k);
  VSSEG_LAUNCH_CHECK("vsseg_wgrad (compute kernel)");
  return VSSEG_OK;
}

int vsseg_cwgrad_launch(const vsseg_wgrad_desc* d, const void* /*zeros*/, hipStream_t s) {
  const char* why = cw_check(d);
  if (why) { vsseg_set_error("vsseg_wgrad: march = 2 (compute kernel) not applicable: %s", why); return VSSEG_EINVAL; }
  VSSEG_CHECK(d->scratch, "vsseg_wgrad: no scratch");
  const int cg = d->hgroup < 1 ? 1 : d->hgroup, ps = d->ntp == 4 ? 2 : 1, ksh = 4 / (ps * cg);
  CwK k;
  auto magic = [](int dv) { return dv <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)dv - 1) / (unsigned)dv); };
  k.p = reinterpret_cast<const char*>(d->p.ptr);
  k.h0 = reinterpret_cast<const char*>(d->h.ptr);
  k.h1 = d->h.ptr2 ? reinterpret_cast<const char*>(d->h.ptr2) - (int64_t)d->h.csplit * 2 : k.h0;
  k.h_split_chunk = d->h.ptr2 ? d->h.csplit / 16 : 1 << 20;
  k.p_vox_bytes = d->p.pitch * 2;
  k.h_vox_bytes = d->h.pitch * 2;
  k.X = d->q[0]; k.Y = d->q[1]; k.Z = d->q[2];
  k.nty = k.Y / CW_TY; k.ntz = k.Z / CW_TZ;
  k.mg_x = magic(k.X); k.mg_ty = magic(k.nty); k.mg_tz = magic(k.ntz);
  const int steps = d->p.n * k.nty * k.ntz * k.X;  // (strip, x) list, x fastest
  k.hchunks = d->h.c / 16;
  k.ncls = k.hchunks / cg;
  k.pow2 = (k.ncls & (k.ncls - 1)) == 0;
  k.slab_chunk = 27 * d->ntp * 16 * 16;
  // workgroups per class: one workgroup per CU over all classes, and what the scratch holds (ksh slabs per workgroup, + a bias row)
  int gpc = 256 / k.ncls;
  if (k.pow2) gpc &= ~7;  // whole rounds of the 8 XCDs (cw_widx)
  const int64_t per_wg = (int64_t)ksh * ((int64_t)k.hchunks * k.slab_chunk + (d->dbias_p ? d->ntp * 16 : 0));  // one slab (+ a bias row) per workgroup and K share
  const int64_t cap = d->scratch_elems / per_wg;
  if (gpc > cap) gpc = k.pow2 ? (int)(cap & ~7ll) : (int)cap;
  if (d->persistent_blocks > 0 && gpc > d->persistent_blocks) gpc = k.pow2 ? (d->persistent_blocks & ~7) : d->persistent_blocks;
  VSSEG_CHECK(gpc >= (k.pow2 ? 8 : 1), "vsseg_wgrad: compute kernel: scratch (%lld floats per workgroup) or persistent_blocks allow fewer than %d workgroups per class", (long long)per_wg, k.pow2 ? 8 : 1);
  const int G = gpc * k.ncls;
  // (XCD, class) groups: sizes by enumeration, step ranges proportional to the sizes (every step exactly once per class)
  int gs[8][8] = {};
  for (int L = 0; L < G; ++L) ++gs[L & 7][cw_class(L, k.ncls, k.pow2)];
  for (int c = 0; c < k.ncls; ++c) {
    int64_t seen = 0;
    for (int x = 0; x < 8; ++x) {
      const int64_t lo = (int64_t)steps * seen / gpc;
      seen += gs[x][c];
      const int64_t hi = (int64_t)steps * seen / gpc;
      k.fstart[x][c] = (int)lo;
      k.fcnt[x][c] = (int)(hi - lo);
      k.gsz[x][c] = (short)(gs[x][c] > 0 ? gs[x][c] : 1);
    }
  }
  {
    static void* zpage[16] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    VSSEG_CHECK(dev >= 0 && dev < 16, "vsseg_wgrad: device index");
    if (!zpage[dev] && (hipMalloc(&zpage[dev], CW_ZERO_BYTES) != hipSuccess || hipMemset(zpage[dev], 0, CW_ZERO_BYTES) != hipSuccess)) zpage[dev] = nullptr;
    VSSEG_CHECK(zpage[dev], "vsseg_wgrad: could not allocate the zero page");
    k.zeros = reinterpret_cast<const char*>(zpage[dev]);
  }
  k.slab = d->scratch;
  k.bias_slab = d->dbias_p ? d->scratch + (int64_t)gpc * ksh * k.hchunks * k.slab_chunk : nullptr;
  int rc;
  const bool b = d->dbias_p != nullptr;
  if (d->ntp == 3 && cg == 1) rc = b ? cw_launch_inst<3, 1, 1, true>(k, G, s) : cw_launch_inst<3, 1, 1, false>(k, G, s);
  else if (d->ntp == 3) rc = b ? cw_launch_inst<3, 1, 2, true>(k, G, s) : cw_launch_inst<3, 1, 2, false>(k, G, s);
  else rc = b ? cw_launch_inst<2, 2, 1, true>(k, G, s) : cw_launch_inst<2, 2, 1, false>(k, G, s);
  if (rc) return rc;
  rc = vsseg_wgrad_reduce_launch(d, d->scratch, gpc * ksh, k.hchunks, k.slab_chunk, s);
  if (rc) return rc;
  if (!d->dbias_p) return VSSEG_OK;
  hipLaunchKernelGGL(vsseg_slab_add_kernel, dim3((d->ntp * 16 + 63) / 64), dim3(VSSEG_SLAB_THREADS), 0, s, (const float*)k.bias_slab, gpc * ksh, d->ntp * 16, d->cp_valid, d->dbias_p);
  VSSEG_LAUNCH_CHECK("vsseg_wgrad (compute kernel, bias)");
  return VSSEG_OK;
}
