// Weight gradient of the MFMA-bound stride-1 3x3x3 convolutions (levels 2-3 of the 2.5D U-Net: 96x32x128 and 48x16x64 voxels; SURVEY §8a rows 9, 10, 13, 14,
// 31, 34, 36, 39 — autograd of ref:params/networks/blocks/convolutions.py:137-146 as run by loss.backward(), ref:params/VSparams.py:461) with a compile-time
// geometry: vsseg_wgrad with march = 2.
//
//   dW[(dx, dy, dz)][cP][cH] = sum_q P[q][cP] * H[q + (dx, dy, dz)][cH]          P = dY (48 or 64 channels), H = X (a multiple of 16 channels; may be two-part)
//
// What held wgrad_kernel (wgrad.hip) at 0.19-0.28 of the bf16 MFMA peak on these layers: every H fragment it read from LDS fed only NTP MFMAs (10 fragment
// reads per 21 MFMAs), 256-voxel tiles with a load -> wait -> multiply round each and 2-3 small workgroups per CU taking turns.  Here:
//
//   * THE THREE z-TAPS OF A VOXEL ROW SHARE ONE H FRAGMENT.  The reduction axis (voxels) runs along z, 32 voxels per MFMA K-step.  Shifting the P operand
//     instead of the H operand —  dW[dz] = sum_z' P[z' - dz] * H[z'] — one H fragment (32 voxels x 16 channels at column (x + dx, y + dy)) is multiplied
//     with the three z-shifted P fragments of every P tile, and the shifted P fragments (3 x NTP, held in registers) serve all nine (dx, dy) taps of the
//     K-step: 9 + 9 fragment reads for 81 MFMAs (NTP = 3).  Tiles therefore partition H along z (no z halo in H) and P along x, y; P carries a z halo of
//     one voxel (zero outside the volume), H an (x, y) halo
//   * one wave per SIMD owns ALL 27 taps of its 16-channel chunk of H: 27 x NPW accumulator tiles (324 / 216 registers) live in registers over every
//     tile of the persistent workgroup, the fragment reads of tap t + 1 are issued in front of the MFMAs of tap t, the P fragments of the next K-step
//     during the last three taps of the current one; nothing inside a stage waits on memory
//   * tile = 2 x 4 columns x 32 z-voxels: each H column of a 16-channel chunk is exactly one 1 KiB LDS-DMA row ([z][32 bytes]: the transpose reads
//     ds_read_b64_tr_b16 of 8 consecutive voxels hit 8 different 32-byte bank groups), P rows are 96 bytes (48 channels) or padded to 160 (64 channels):
//     an odd number of 32-byte groups, conflict-free for every z shift.  Both operands of a tile arrive by LDS-DMA in one of two buffers while the other
//     is multiplied (one `s_waitcnt vmcnt(0)` + one s_barrier per tile)
//   * the four waves of a workgroup are (P half) x (H chunk) x (K-step share); workgroups of chunk class c own the chunks [c*CG, (c+1)*CG); the classes of
//     a tile run on the same XCD at about the same time (P comes from that L2), every (XCD, class) group walks a contiguous tile range
//   * partial sums leave as slabs in wgrad_kernel's layout and are summed in a fixed order by its reduce launch (run-to-run bit-identical)
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

constexpr int CW_TX = 2, CW_TY = 4, CW_TZ = 32;
constexpr int CW_HX = CW_TX + 2, CW_HY = CW_TY + 2;
constexpr int CW_HCOLS = CW_HX * CW_HY;  // 24 halo columns, 1 KiB each per 16-channel chunk
constexpr int CW_NCOL = CW_TX * CW_TY;   // 8 columns = 8 K-steps per tile
constexpr int CW_PROWS = CW_TZ + 2;      // P column with its z halo
constexpr int cw_prow(int ntp) { return (ntp & 1) ? ntp * 32 : ntp * 32 + 32; }  // bytes per P voxel row in LDS: an odd number of 32-byte groups
constexpr int cw_pdrows(int ntp) { return (CW_NCOL * CW_PROWS * (cw_prow(ntp) / 16) + 63) / 64; }
constexpr int cw_buf_bytes(int ntp, int cg) { return cg * CW_HCOLS * 1024 + cw_pdrows(ntp) * 1024; }

struct CwK {
  const char* p;
  const char* h0;
  const char* h1;       // part 1 of a two-part H, biased by -csplit channels (== h0 for an ordinary tensor)
  int h_split_chunk;    // first 16-channel chunk that lives in part 1
  int p_vox_bytes, h_vox_bytes;
  int X, Y, Z, ntx, nty, ntz;
  unsigned mg_ty, mg_tx, mg_tz;
  int ncls, pow2, gpc;  // chunk classes; class map (see cw_class); workgroups per class
  int hchunks, slab_chunk;
  const void* zeros;
  float* slab;
  float* bias_slab;
  int tstart[8][8];     // first tile of the (XCD, class) group
  int tcnt[8][8];       // its tile count
  short gsz[8][8];      // its workgroups
};

__device__ __forceinline__ unsigned cw_div(unsigned n, unsigned magic) { return magic ? __umulhi(n, magic) : n; }
// chunk class of workgroup L (runs on XCD L % 8): with a power-of-two class count the classes alternate inside an XCD's workgroups; otherwise
// L % ncls, which spreads every class over all XCDs when ncls is odd
__host__ __device__ inline int cw_class(int L, int ncls, int pow2) { return pow2 ? (L >> 3) % ncls : L % ncls; }
__host__ __device__ inline int cw_widx(int L, int ncls, int pow2) { return pow2 ? ((L >> 3) / ncls) * 8 + (L & 7) : L / ncls; }

typedef __attribute__((address_space(3))) bf16x4 cw_lds_b4;
__device__ __forceinline__ bf16x8 cw_frag(const char* a, int second) {  // two transpose reads: voxels 4g .. 4g+3 and 16 + 4g .. of the lane's channel
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((cw_lds_b4*)(a));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((cw_lds_b4*)(a + second));
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
// tile are shared by KS = 4 / (PS * CG) waves
template <int NPW, int PS, int CG, bool BIAS, int EXP = 0>
__global__ __launch_bounds__(256, 1) void cwgrad_kernel(const CwK k) {
  constexpr int KS = 4 / (PS * CG), NTP = NPW * PS, PROW = cw_prow(NTP), PPR = PROW / 16, PREAL = NTP * 2;
  constexpr int PCOL = CW_PROWS * PROW, HBYTES = CG * CW_HCOLS * 1024, PSLOTS = CW_NCOL * CW_PROWS * PPR, PDROWS = cw_pdrows(NTP);
  constexpr int BUF = cw_buf_bytes(NTP, CG), NPD = (PDROWS + 3) / 4, NHD = CG * CW_HCOLS / 4, NK = CW_NCOL / KS;
  static_assert(KS >= 1 && KS * PS * CG == 4 && CW_TY % KS == 0, "wave roles");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, l15 = lane & 15;
  const int ps = wave % PS, cg = (wave / PS) % CG, ks = wave / (PS * CG);
  const int X = k.X, Y = k.Y, Z = k.Z;

  // ---- which tiles: workgroup L runs on XCD L % 8; group (XCD, class) owns a contiguous tile range and its workgroups stride through it ----
  const int L = blockIdx.x, xcd = L & 7, cls = cw_class(L, k.ncls, k.pow2), widx = cw_widx(L, k.ncls, k.pow2);
  int rank = 0;
  for (int l2 = xcd; l2 < L; l2 += 8) rank += cw_class(l2, k.ncls, k.pow2) == cls;
  const int gsz = k.gsz[xcd][cls], t_first = k.tstart[xcd][cls], t_cnt = k.tcnt[xcd][cls];
  const int my_tiles = t_cnt > rank ? (t_cnt - 1 - rank) / gsz + 1 : 0;
  struct Tile { int n, x0, y0, z0; };
  auto tile_of = [&](int i) {  // tile order (n, z, x, y), y fastest: the tiles in flight on an XCD are (x, y) neighbours, which share halo columns
    unsigned b = (unsigned)(t_first + rank + i * gsz);
    Tile t;
    unsigned q = cw_div(b, k.mg_ty); t.y0 = (int)(b - q * k.nty) * CW_TY; b = q;
    q = cw_div(b, k.mg_tx); t.x0 = (int)(b - q * k.ntx) * CW_TX; b = q;
    q = cw_div(b, k.mg_tz); t.z0 = (int)(b - q * k.ntz) * CW_TZ; t.n = (int)q;
    return t;
  };

  // ---- LDS-DMA tables.  H: column j = i*4 + wave of the CG x 24 halo columns; lane -> (z = lane >> 1, 16-byte half of the chunk) ----
  const unsigned h_lane = (unsigned)(lane >> 1) * (unsigned)k.h_vox_bytes + (unsigned)(lane & 1) * 16u;
  int hcol[NHD];  // hx | hy << 8 | chunk-in-group << 16 (wave-uniform)
#pragma unroll
  for (int i = 0; i < NHD; ++i) {
    const int j = i * 4 + wave, c = j / CW_HCOLS, hc = j - c * CW_HCOLS;
    hcol[i] = __builtin_amdgcn_readfirstlane((hc / CW_HY) | ((hc % CW_HY) << 8) | (c << 16));
  }
  // P: slot j = (u*4 + wave)*64 + lane of [column][row 0..33][PPR 16-byte slots]; padding slots are never written
  int prel[NPD];
  unsigned pflags = 0;  // 2 bits per u: 0 interior row, 1 row z0 - 1, 2 row z0 + 32, 3 no piece
#pragma unroll
  for (int u = 0; u < NPD; ++u) {
    const int j = (u * 4 + wave) * 64 + lane;
    const int c = j / (CW_PROWS * PPR), rem = j - c * (CW_PROWS * PPR), row = rem / PPR, pc = rem - row * PPR;
    const bool real = j < PSLOTS && pc < PREAL;
    prel[u] = real ? (((c / CW_TY) * Y + (c % CW_TY)) * Z + row - 1) * k.p_vox_bytes + pc * 16 : 0;
    pflags |= (unsigned)(!real ? 3 : (row == 0 ? 1 : (row == CW_PROWS - 1 ? 2 : 0))) << (2 * u);
  }

  auto issue = [&](const Tile& t, int buf) {
    char* Hdst = smem + buf * BUF;
    char* Pdst = Hdst + HBYTES;
#pragma unroll
    for (int i = 0; i < NHD; ++i) {
      const int hx = hcol[i] & 255, hy = (hcol[i] >> 8) & 255, c = hcol[i] >> 16;
      const int gx = t.x0 - 1 + hx, gy = t.y0 - 1 + hy, chunk = cls * CG + c;
      const bool ok = ((unsigned)gx < (unsigned)X) & ((unsigned)gy < (unsigned)Y);  // wave-uniform: the zero padding of the convolution comes from the zero page
      const char* base = (chunk >= k.h_split_chunk ? k.h1 : k.h0) + ((((int64_t)t.n * X + gx) * Y + gy) * Z + t.z0) * k.h_vox_bytes + chunk * 32;
      vsseg_dma16(ok ? (const void*)(base + h_lane) : k.zeros, Hdst + (i * 4 + wave) * 1024);
    }
    const char* origin = k.p + ((((int64_t)t.n * X + t.x0) * Y + t.y0) * Z + t.z0) * k.p_vox_bytes;
    const unsigned edge = (t.z0 == 0 ? 1u : 0u) | (t.z0 + CW_TZ == Z ? 2u : 0u);
#pragma unroll
    for (int u = 0; u < NPD; ++u) {
      const int row = u * 4 + wave;
      if (row >= PDROWS) break;  // wave-uniform
      const unsigned f = (pflags >> (2 * u)) & 3u;
      if (f != 3u) vsseg_dma16((f & edge) ? k.zeros : (const void*)(origin + prel[u]), Pdst + row * 1024);
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
  const int h_lane_off = (cg * CW_HCOLS + ks) * 1024 + r4 * 32 + qc;
  const int p_lane_off = HBYTES + ks * PCOL + r4 * PROW + ps * NPW * 32 + qc;

  int exp_dummy = 0;
  if (my_tiles > 0) issue(tile_of(0), 0);
  for (int st = 0; st < my_tiles; ++st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // tile st has landed for every wave; every wave is done with the other buffer
    const char* Hs = smem + (st & 1) * BUF + h_lane_off;
    const char* Ps = smem + (st & 1) * BUF + p_lane_off;
    bf16x8 pa[2][3][NPW], hb[3];  // P fragments of this / the next K-step; H fragments: a ring, read two taps ahead
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int p = 0; p < NPW; ++p) pa[0][s][p] = cw_frag(Ps + s * PROW + p * 32, 16 * PROW);
    // H fragment of the j-th (K-step, tap) pair of the stage
    auto h_off = [](int j) { const int i = j / 9, t = j % 9; return ((((i * KS) / CW_TY) + t / 3) * CW_HY + ((i * KS) % CW_TY) + t % 3) * 1024; };
    hb[0] = cw_frag(Hs + h_off(0), 512);
    hb[1] = cw_frag(Hs + h_off(1), 512);
    __builtin_amdgcn_sched_barrier(0);
    if (st + 1 < my_tiles && (EXP != 1 || st < 1)) issue(tile_of(st + 1), (st + 1) & 1);  // behind the first fragment reads: its address arithmetic covers their latency
    __builtin_amdgcn_sched_barrier(0);
    cw_for<NK * 9>([&](auto jc) {
      constexpr int j = decltype(jc)::value, i = j / 9, t = j % 9;
      if constexpr (j + 2 < NK * 9 && EXP != 3) hb[(j + 2) % 3] = cw_frag(Hs + h_off(j + 2), 512);
      if constexpr (i + 1 < NK && t >= 6 && EXP != 3) {  // the P fragments of the next K-step during the last three taps of this one
#pragma unroll
        for (int p = 0; p < NPW; ++p) pa[(i + 1) & 1][t - 6][p] = cw_frag(Ps + (i + 1) * KS * PCOL + (t - 6) * PROW + p * 32, 16 * PROW);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (BIAS && t == 4) {  // the centre tap's iteration carries the bias MFMAs (shift 0 = the tile's own voxels, each exactly once)
#pragma unroll
        for (int p = 0; p < NPW; ++p) {
          if constexpr (ASM) cw_mfma<false, true>(accb[p], pa[i & 1][1][p], ones);
          else accb[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[i & 1][1][p], ones, accb[p], 0, 0, 0);
        }
      }
      cw_for<3 * NPW>([&](auto qc_) {
        constexpr int q = decltype(qc_)::value, a = t * 3 * NPW + q;
        const bf16x8& pa_q = pa[i & 1][q / NPW][q % NPW];
        const bf16x8& hb_j = hb[j % 3];
        if constexpr (EXP == 2) exp_dummy ^= (int)pa_q[0] ^ (int)hb_j[q & 7];
        else if constexpr (ASM) cw_mfma<(a < NAG), q == 0>(acc[a], pa_q, hb_j);
        else acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa_q, hb_j, acc[a], 0, 0, 0);
      });
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  if constexpr (ASM) cw_mfma_drain();
  if (EXP == 2 && exp_dummy == 0x12345) k.slab[0] = 1.f;

  // ---- flush: this wave's accumulators -> its slab [tap][cP][16 channels of its chunk] (wgrad_kernel's layout: slab index = workgroup-of-class * KS + K share) ----
  const int chunk = cls * CG + cg;
  float* slab = k.slab + (((int64_t)widx * KS + ks) * k.hchunks + chunk) * k.slab_chunk;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int tap = t * 3 + (2 - s);  // P shifted by s - 1 pairs P[z' - dz] with H[z']: dz = 1 - s
#pragma unroll
      for (int p = 0; p < NPW; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[((int64_t)tap * (NTP * 16) + (ps * NPW + p) * 16 + g * 4 + r) * 16 + l15] = acc[(t * 3 + s) * NPW + p][r];
    }
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

// ---- host side ------------------------------------------------------------------------------------------------------
static const char* cw_check(const vsseg_wgrad_desc* d) {
  if (d->p.dtype != VSSEG_BF16 || d->h.dtype != VSSEG_BF16) return "operands are not bf16";
  if (d->ntaps != 27) return "3x3x3 taps only";
  for (int t = 0; t < 27; ++t)
    if (d->tap_off[t][0] != t / 9 - 1 || d->tap_off[t][1] != (t / 3) % 3 - 1 || d->tap_off[t][2] != t % 3 - 1) return "taps are not the 3x3x3 stencil in (x, y, z) order";
  for (int a = 0; a < 3; ++a)
    if (d->hs[a] != 1) return "stride-1 lattices only";
  if (d->q[0] != d->p.x || d->q[1] != d->p.y || d->q[2] != d->p.z || d->q[0] != d->h.x || d->q[1] != d->h.y || d->q[2] != d->h.z || d->p.n != d->h.n) return "lattice, P and H extents differ";
  if (d->q[0] % CW_TX || d->q[1] % CW_TY || d->q[2] % CW_TZ) return "extent is not a multiple of the 2x4x32 tile";
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
  if ((int64_t)d->p.n * d->q[0] * d->q[1] * d->q[2] / (CW_TX * CW_TY * CW_TZ) >= (1ll << 24)) return "too many tiles";
  return nullptr;
}

template <int NPW, int PS, int CG, bool BIAS, int EXP = 0> static int cw_launch_inst(const CwK& k, int grid, hipStream_t s) {
  static bool attr_set[16] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 16 && !attr_set[dev]) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&cwgrad_kernel<NPW, PS, CG, BIAS, EXP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL((cwgrad_kernel<NPW, PS, CG, BIAS, EXP>), dim3((unsigned)grid), dim3(256), 2 * cw_buf_bytes(NPW * PS, CG), s, k);
  VSSEG_LAUNCH_CHECK("vsseg_wgrad (compute kernel)");
  return VSSEG_OK;
}

int vsseg_wgrad_reduce_launch(const vsseg_wgrad_desc* d, float* slab, int nblk, int hchunks, int slab_chunk, hipStream_t s);  // wgrad.hip

int vsseg_cwgrad_launch(const vsseg_wgrad_desc* d, const void* zeros, hipStream_t s) {
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
  k.ntx = k.X / CW_TX; k.nty = k.Y / CW_TY; k.ntz = k.Z / CW_TZ;
  k.mg_tx = magic(k.ntx); k.mg_ty = magic(k.nty); k.mg_tz = magic(k.ntz);
  const int tiles = d->p.n * k.ntx * k.nty * k.ntz;
  k.hchunks = d->h.c / 16;
  k.ncls = k.hchunks / cg;
  k.pow2 = (k.ncls & (k.ncls - 1)) == 0;
  k.slab_chunk = 27 * d->ntp * 16 * 16;
  // workgroups per class: one workgroup per CU over all classes, never more than tiles, and what the scratch holds (ksh slabs per workgroup, + a bias row)
  int gpc = 256 / k.ncls;
  if (k.pow2) gpc &= ~7;  // whole rounds of the 8 XCDs (cw_widx)
  const int64_t per_wg = (int64_t)ksh * ((int64_t)k.hchunks * k.slab_chunk + (d->dbias_p ? d->ntp * 16 : 0));
  const int64_t cap = d->scratch_elems / per_wg;
  if (gpc > cap) gpc = k.pow2 ? (int)(cap & ~7ll) : (int)cap;
  if (d->persistent_blocks > 0 && gpc > d->persistent_blocks) gpc = k.pow2 ? (d->persistent_blocks & ~7) : d->persistent_blocks;
  VSSEG_CHECK(gpc >= (k.pow2 ? 8 : 1), "vsseg_wgrad: scratch too small for the compute kernel (%lld floats per workgroup)", (long long)per_wg);
  k.gpc = gpc;
  const int G = gpc * k.ncls;
  // (XCD, class) groups: sizes by enumeration, tile ranges proportional to the sizes (every tile exactly once per class)
  int gs[8][8] = {};
  for (int L = 0; L < G; ++L) ++gs[L & 7][cw_class(L, k.ncls, k.pow2)];
  for (int c = 0; c < k.ncls; ++c) {
    int64_t seen = 0;
    for (int x = 0; x < 8; ++x) {
      const int64_t lo = (int64_t)tiles * seen / gpc;
      seen += gs[x][c];
      const int64_t hi = (int64_t)tiles * seen / gpc;
      k.tstart[x][c] = (int)lo;
      k.tcnt[x][c] = (int)(hi - lo);
      k.gsz[x][c] = (short)(gs[x][c] > 0 ? gs[x][c] : 1);
    }
  }
  k.zeros = zeros;
  k.slab = d->scratch;
  k.bias_slab = d->dbias_p ? d->scratch + (int64_t)gpc * ksh * k.hchunks * k.slab_chunk : nullptr;
  int rc;
  const bool b = d->dbias_p != nullptr;
  if (d->ntp == 3 && cg == 1) rc = b ? cw_launch_inst<3, 1, 1, true>(k, G, s) : cw_launch_inst<3, 1, 1, false>(k, G, s);
  else if (d->ntp == 3 && !b && getenv("VSSEG_CW_EXP")) {
    const int e = atoi(getenv("VSSEG_CW_EXP"));
    rc = e == 1 ? cw_launch_inst<3, 1, 2, false, 1>(k, G, s) : e == 2 ? cw_launch_inst<3, 1, 2, false, 2>(k, G, s) : e == 3 ? cw_launch_inst<3, 1, 2, false, 3>(k, G, s) : cw_launch_inst<3, 1, 2, false>(k, G, s);
  } else if (d->ntp == 3) rc = b ? cw_launch_inst<3, 1, 2, true>(k, G, s) : cw_launch_inst<3, 1, 2, false>(k, G, s);
  else rc = b ? cw_launch_inst<2, 2, 1, true>(k, G, s) : cw_launch_inst<2, 2, 1, false>(k, G, s);
  if (rc) return rc;
  rc = vsseg_wgrad_reduce_launch(d, d->scratch, gpc * ksh, k.hchunks, k.slab_chunk, s);
  if (rc || !d->dbias_p) return rc;
  hipLaunchKernelGGL(vsseg_slab_add_kernel, dim3((d->ntp * 16 + 63) / 64), dim3(VSSEG_SLAB_THREADS), 0, s, (const float*)k.bias_slab, gpc * ksh, d->ntp * 16, d->cp_valid, d->dbias_p);
  VSSEG_LAUNCH_CHECK("vsseg_wgrad (compute kernel, bias)");
  return VSSEG_OK;
}
