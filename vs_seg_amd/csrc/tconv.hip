// Transition kernel (launch plans with depth -8): the 3x3x3 stride-(2,2,2) transposed convolution 64 -> 48 from the 48x16x64 level up to 96x32x128 and the data gradient of the
// strided convolution 48 -> 48 between the same two levels (ref:params/networks/nets/unet2d5_spvPA.py:56-93, blocks/convolutions.py:114-156; SURVEY §8a rows 10 / 36 and their
// autograd) — ALL EIGHT output-parity classes in one launch.
//
// out[2q + p] = sum over the taps of class p of W_t * in[q + off_t]: class p = (px, py, pz) has 1, 2, 4 or 8 taps (one per axis where p is even, two where it is odd: 27 in all),
// its operand is the COARSE tensor at stride 1.  The general kernel runs the classes as eight launches (each reads the whole input, 31 us apiece at batch 4) or as one class-split
// launch (every workgroup row fetches the union halo and keeps the LDS of the 8-tap class); the deep-level kernel (dconv.hip) runs them in one workgroup with whole classes per
// wave, weights streamed to registers from L2 — all three land at 170-280 TFLOP/s and ~1 TB/s on these two layers (profiles/r06_dconv_bench.txt: 183-282 us for 0.15-0.18 GB and
// 25-33 GFLOP).  Here:
//   * a workgroup (4 waves) owns a 4x8x8 tile of the COARSE lattice = 16 M-tiles, 4 per wave, all 3 output-channel tiles: the waves split M, nothing is reduced across waves
//   * the tile's halo (5x9x9 coarse voxels, all channels, voxel stride padded to an odd number of 16-byte units as in dconv.hip) is fetched ONCE per tile by LDS-DMA
//   * a stage = one parity class: its packed weights (2 K-steps of 3 KiB per tap at 64 channels: 6-48 KiB) arrive by LDS-DMA in one of two buffers while the previous class is
//     multiplied — one counted wait and one barrier per class — and are shared by the four waves (a weight fragment feeds 4 M-tiles per read)
//   * the K loop reads the next K-step's fragments while the current one multiplies; epilogue per class: bias (+ statistics) (+ eval affine / activation) (+ accumulate), one
//     8-byte store per lane and (M-tile, channel tile) at the fine voxel 2q + p
// Same packed weights as a class-split plan of the general kernel ([class][K-steps][tiles][64 lanes][8], K order (tap, 8-channel group): planner.pack_map) and the same
// accumulation order per output value: results are bit-identical to its launches (tests/test_gpu_ops.py::test_transition_kernel_*).
#include "tconv.h"
#include <type_traits>

constexpr int TC_THREADS = 256, TC_WAVES = 4, TC_MT = 4, TC_NT = 3;
constexpr int TC_TX = 4, TC_TY = 8, TC_TZ = 8;
constexpr int TC_HX = TC_TX + 1, TC_HY = TC_TY + 1, TC_HZ = TC_TZ + 1, TC_HVOX = TC_HX * TC_HY * TC_HZ;  // 405
constexpr int tc_vs(int cg) { return (cg | 1) * 16; }
constexpr int tc_ksmax(int cg) { return (8 * cg + 3) / 4; }
constexpr int tc_hrows(int cg) { return (TC_HVOX * (cg | 1) + 63) / 64; }  // 1 KiB DMA rows of the halo (slot = voxel * (cg | 1) + piece)
constexpr int tc_wbuf(int cg) { return tc_ksmax(cg) * TC_NT * 1024; }
constexpr int tc_off_epi(int cg) { return 8 * tc_ksmax(cg) * 4 * 4; }
constexpr int tc_off_stat(int cg) { return tc_off_epi(cg) + 3 * TC_NT * 16 * 4; }
constexpr int tc_off_halo(int cg) { return (tc_off_stat(cg) + TC_WAVES * 2 * TC_NT * 16 * 4 + 1023) / 1024 * 1024; }
constexpr int tc_off_w(int cg) { return tc_off_halo(cg) + tc_hrows(cg) * 1024; }
constexpr int tc_lds_bytes(int cg) { return tc_off_w(cg) + 2 * tc_wbuf(cg); }

struct TconvK {
  const char* in;
  char* out;
  const char* wpack;
  const float *bias, *bias2, *scale, *shift, *alpha;
  double* stats;
  unsigned* fxflag;
  const void* zeros;
  int in_vox_bytes, out_vox_bytes, accumulate, act, cout, stats_stride;
  int X, Y, Z;        // coarse (input) extents
  int omin[3];        // halo origin relative to the tile origin (-1 or 0 per axis)
  int ntx, nty, ntz;
  unsigned mg_tz, mg_ty, mg_tx;
  int tiles, per_xcd;
  int class_ntaps[8], class_oo[8][3];
  int order[8];       // the classes in the order a tile runs them (most taps first)
  int class_hoff[8][8];  // halo-voxel offset of tap i of class c relative to the voxel's own halo position
};

__device__ __forceinline__ unsigned tc_div(unsigned n, unsigned magic) { return magic ? __umulhi(n, magic) : n; }
// The MFMAs are inline assembly with the accumulator tied in place ("+a"): with the builtin hipcc rotated five of the twelve accumulator tiles of the K loop through a[0:3]
// (four v_accvgpr_mov and two s_nop behind every one of those MFMAs: the K loop ran at a third of its MFMA time).  hipcc pads no hazards around inline assembly (cwgrad.hip):
// PAD = two wait states in front of the first MFMA of a block (a VALU move of an operand register may sit right in front of it), tc_mfma_drain before the results are read.
template <bool PAD> __device__ __forceinline__ void tc_mfma(f32x4& c, const bf16x8& a, const bf16x8& b) {
  if constexpr (PAD) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void tc_mfma_drain() { asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory"); }

// MODE: 0 plain, 1 + BatchNorm statistics, 2 accumulate (out += ...), 3 eval affine + activation
template <int CG, int MODE>
__global__ __launch_bounds__(TC_THREADS, 1) void tconv_kernel(const TconvK k) {
  constexpr int VS = tc_vs(CG), CGP = CG | 1, KSMAX = tc_ksmax(CG), HROWS = tc_hrows(CG), WBUF = tc_wbuf(CG), NH = (HROWS + TC_WAVES - 1) / TC_WAVES;
  constexpr int MT = TC_MT, NT = TC_NT;
  constexpr bool STATS = MODE == 1, ACC = MODE == 2, EVAL = MODE == 3;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* ktab = reinterpret_cast<int*>(smem);  // [class][K-step * 4 + K-group] -> byte offset of the group's 16 bytes relative to the voxel's own halo position
  float* epi = reinterpret_cast<float*>(smem + tc_off_epi(CG));
  float* sred = reinterpret_cast<float*>(smem + tc_off_stat(CG));
  char* halo = smem + tc_off_halo(CG);
  char* wbuf = smem + tc_off_w(CG);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, l15 = lane & 15;
  const TconvK* __restrict__ kp = (const TconvK*)__builtin_amdgcn_kernarg_segment_ptr();  // run-time indexed tables are read through the kernarg segment (dconv.hip)
  const int X = k.X, Y = k.Y, Z = k.Z, OY = 2 * Y, OZ = 2 * Z;

  for (int e = tid; e < 8 * KSMAX * 4; e += TC_THREADS) {
    const int cl = e / (KSMAX * 4), p = e - cl * (KSMAX * 4), tap = p / CG, cg = p - tap * CG;
    ktab[e] = tap < kp->class_ntaps[cl] ? kp->class_hoff[cl][tap & 7] * VS + cg * 16 : 0;  // padded K-groups: zero weights times the voxel's own (finite) data
  }
  for (int i = tid; i < NT * 16; i += TC_THREADS) {
    const bool ok = i < k.cout;
    epi[i] = ((ok && k.bias) ? k.bias[i] : 0.f) + ((ok && k.bias2) ? k.bias2[i] : 0.f);
    epi[NT * 16 + i] = (ok && k.scale) ? k.scale[i] : 1.f;
    epi[2 * NT * 16 + i] = (ok && k.scale) ? k.shift[i] : 0.f;
  }
  const float slope = !EVAL ? 1.f : (k.act == VSSEG_ACT_PRELU ? (k.alpha ? *k.alpha : 0.f) : (k.act == VSSEG_ACT_RELU ? 0.f : 1.f));

  // ---- this thread's halo pieces: LDS slot sl = (u*4 + wave)*64 + lane holds piece sl % CGP of halo voxel sl / CGP (piece CG = the padding unit: zero page) ----
  int hrel[NH];
  unsigned hxyz[NH];
#pragma unroll
  for (int u = 0; u < NH; ++u) {
    const int sl = (u * TC_WAVES + wave) * 64 + lane;
    const int hv = sl / CGP, pc = sl - hv * CGP, hz = hv % TC_HZ, r = hv / TC_HZ, hy = r % TC_HY, hx = r / TC_HY;
    const bool ok = sl < TC_HVOX * CGP && pc < CG;
    hrel[u] = ok ? ((hx * Y + hy) * Z + hz) * k.in_vox_bytes + pc * 16 : 0;
    hxyz[u] = ok ? (unsigned)(hx | (hy << 8) | (hz << 16)) : 0xffffffffu;
  }
  // ---- this lane's voxel of each of its wave's M-tiles: v = (wave*MT + m)*16 + l15 -> (vx, vy, vz) = (v >> 6, (v >> 3) & 7, v & 7): two y-rows of eight z.  (The operand reads of
  //      this mapping are not conflict-free — SQ_LDS_BANK_CONFLICT is half of SQ_LDS_IDX_ACTIVE; the lane order that the LDS model of tools/lds_conflicts.py prefers, even z
  //      on lanes 0-3 / 12-15 and odd z on lanes 4-11, measured 5-12 % SLOWER: the K loop is not what bounds this kernel) ----
  int abase[MT], ofine[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int v = (wave * MT + m) * 16 + l15, vz = v & 7, vy = (v >> 3) & 7, vx = v >> 6;
    abase[m] = ((vx * TC_HY + vy) * TC_HZ + vz) * VS;
    ofine[m] = ((2 * vx) * OY + 2 * vy) * OZ + 2 * vz;  // fine-lattice voxel offset relative to the tile's fine origin (class offsets are added per stage)
  }
  float ssum[STATS ? NT : 1][4], ssq[STATS ? NT : 1][4];
#pragma unroll
  for (int t = 0; t < (STATS ? NT : 1); ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) { ssum[t][r] = 0.f; ssq[t][r] = 0.f; }

  // ---- tile schedule (cconv.hip): XCD x owns tiles [x*per_xcd, (x+1)*per_xcd), its workgroups stride through them ----
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, wgs = gridDim.x >> 3;
  int cnt = kp->tiles - xcd * kp->per_xcd;
  cnt = cnt > kp->per_xcd ? kp->per_xcd : cnt;
  const int my_tiles = cnt > slot ? (cnt - 1 - slot) / wgs + 1 : 0;

  auto issue_w = [&](int cl, int buf) __attribute__((always_inline)) {  // the packed weights of class cl: K-steps [0, nks) x NT rows of 1 KiB
    const int rows = ((kp->class_ntaps[cl] * CG + 3) >> 2) * NT;
    const char* src = k.wpack + (int64_t)cl * (KSMAX * NT * 1024) + lane * 16;
    char* dst = wbuf + buf * WBUF;
    for (int row = wave; row < rows; row += TC_WAVES) vsseg_dma16(src + row * 1024, dst + row * 1024);
  };

  struct Tile { int smp, q0x, q0y, q0z; };
  auto tile_of = [&](int ti) {
    unsigned b = (unsigned)(xcd * kp->per_xcd + slot + ti * wgs);
    Tile t;
    unsigned qd = tc_div(b, kp->mg_tz); t.q0z = (int)(b - qd * kp->ntz) * TC_TZ; b = qd;
    qd = tc_div(b, kp->mg_ty); t.q0y = (int)(b - qd * kp->nty) * TC_TY; b = qd;
    qd = tc_div(b, kp->mg_tx); t.q0x = (int)(b - qd * kp->ntx) * TC_TX; t.smp = (int)qd;
    return t;
  };
  auto issue_halo = [&](const Tile& t) __attribute__((always_inline)) {
    const int ix = t.q0x + kp->omin[0], iy = t.q0y + kp->omin[1], iz = t.q0z + kp->omin[2];
    const char* org = k.in + ((((int64_t)t.smp * X + ix) * Y + iy) * Z + iz) * k.in_vox_bytes;
#pragma unroll
    for (int u = 0; u < NH; ++u) {
      const int row = u * TC_WAVES + wave;
      if (row >= HROWS) break;  // wave-uniform
      const unsigned h = hxyz[u];
      const int gx = ix + (int)(h & 255u), gy = iy + (int)((h >> 8) & 255u), gz = iz + (int)((h >> 16) & 255u);
      const bool ok = (h != 0xffffffffu) & ((unsigned)gx < (unsigned)X) & ((unsigned)gy < (unsigned)Y) & ((unsigned)gz < (unsigned)Z);
      vsseg_dma16(ok ? (const void*)(org + hrel[u]) : k.zeros, halo + row * 1024);
    }
  };

  // The classes of a tile run in the order kp->order (most taps first: the weights of the next, smaller class arrive while the current one multiplies); stage s = 8 * tile + i uses
  // weight buffer s & 1.  Inside a stage the next stage's weight rows are issued between the K-steps (an LDS-DMA instruction holds its wave ~85 cycles and this kernel has one wave per
  // SIMD: DESIGN.md 3.12 / 3.15); behind the last class's K loop, once every wave has left it, the NEXT tile's halo is issued in front of the epilogue.
  const int nstages = my_tiles * 8;
  Tile t_cur{0, 0, 0, 0};
  if (nstages > 0) {
    t_cur = tile_of(0);
    __syncthreads();  // the tables are visible
    issue_halo(t_cur);
    issue_w(kp->order[0], 0);
  }
  for (int s = 0; s < nstages; ++s) {
    const int ci = s & 7, cl = kp->order[ci];
    // the halo and this class's weights have landed once only the previous stage's stores are still in flight (VMEM operations of a wave retire in issue order)
    if (s == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MT * NT) : "memory");
    __builtin_amdgcn_s_barrier();  // ... for every wave; and every wave has finished reading the other weight buffer
    const bool more = s + 1 < nstages;
    const int ncl = kp->order[(ci + 1) & 7];
    const int nrows = more ? ((kp->class_ntaps[ncl] * CG + 3) >> 2) * NT : 0;   // rows of 1 KiB of the next stage's weights; this wave issues rows wave, wave + 4, ...
    const char* nsrc = k.wpack + (int64_t)ncl * (KSMAX * NT * 1024) + lane * 16;
    char* ndst = wbuf + ((s + 1) & 1) * WBUF;
    int nrow = wave;
    const int nks = (kp->class_ntaps[cl] * CG + 3) >> 2, last = nks - 1;
    const int per_ks = ((nrows + TC_WAVES - 1) / TC_WAVES + nks - 1) / nks;  // pieces per K-step so that all are issued by the last one
    const char* Ws = wbuf + (s & 1) * WBUF + lane * 16;
    const int* kt = ktab + cl * (KSMAX * 4) + g;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // K loop: a ring of three operand sets, the fragments of K-step i + 2 are requested while K-step i multiplies (one wave per SIMD: with the next step only, the wave stood in
    // s_waitcnt lgkmcnt(0) behind every block of twelve MFMAs — four waves' 28 KiB of fragment reads per K-step queue in the LDS — and with two sets and register copies it also
    // paid 28 v_mov per step: 640 cycles per K-step against 192 of MFMA time).  The loop is unrolled by three so that the sets are named statically; clamped indices, no branch
    // around a load (the last steps re-read the final fragments).
    bf16x8 wr[3][NT], ar[3][MT];
    auto ld = [&](auto sc, int kidx, int koff) __attribute__((always_inline)) {
      constexpr int S = decltype(sc)::value;
#pragma unroll
      for (int t = 0; t < NT; ++t) wr[S][t] = *reinterpret_cast<const bf16x8*>(Ws + (kidx * NT + t) * 1024);
#pragma unroll
      for (int m = 0; m < MT; ++m) ar[S][m] = *reinterpret_cast<const bf16x8*>(halo + abase[m] + koff);
    };
    int koff_q;  // the K-group offset of the K-step whose fragments are requested next
    auto step = [&](auto sc, int ks) __attribute__((always_inline)) {
      constexpr int S = decltype(sc)::value, S2 = (S + 2) % 3;
      const int k2 = ks + 2 < last ? ks + 2 : last, k3 = ks + 3 < last ? ks + 3 : last;
      ld(std::integral_constant<int, S2>{}, k2, koff_q);
      koff_q = kt[k3 * 4];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (m == 0 && t == 0) tc_mfma<true>(acc[m][t], wr[S][t], ar[S][m]);
          else tc_mfma<false>(acc[m][t], wr[S][t], ar[S][m]);
        }
      __builtin_amdgcn_sched_barrier(0);
      // this K-step's share of the next stage's weight rows, behind the step's MFMAs (they are in the matrix pipe while the wave sits in the DMA instructions)
      for (int j = 0; j < per_ks; ++j) {
        if (nrow < nrows) vsseg_dma16(nsrc + nrow * 1024, ndst + nrow * 1024);
        nrow += TC_WAVES;
      }
    };
    {
      const int k1 = 1 < last ? 1 : last, k2 = 2 < last ? 2 : last;
      const int koff0 = kt[0], koff1 = kt[k1 * 4];
      koff_q = kt[k2 * 4];
      ld(std::integral_constant<int, 0>{}, 0, koff0);
      ld(std::integral_constant<int, 1>{}, k1, koff1);
    }
    int ks = 0;
    for (; ks + 3 <= nks; ks += 3) {
      step(std::integral_constant<int, 0>{}, ks);
      step(std::integral_constant<int, 1>{}, ks + 1);
      step(std::integral_constant<int, 2>{}, ks + 2);
    }
    if (ks < nks) step(std::integral_constant<int, 0>{}, ks);
    if (ks + 1 < nks) step(std::integral_constant<int, 1>{}, ks + 1);
    tc_mfma_drain();
    const Tile t_this = t_cur;
    if (ci == 7 && more) {  // the next tile's halo, as soon as every wave has left this tile's last K loop: its latency lies behind this class's epilogue
      t_cur = tile_of((s >> 3) + 1);
      __syncthreads();
      issue_halo(t_cur);
    }

    // ---- epilogue of the class: fine voxel 2q + p ----
    const int64_t ofine0 = (((int64_t)t_this.smp * (2 * X) + 2 * t_this.q0x) * OY + 2 * t_this.q0y) * OZ + 2 * t_this.q0z;
    const int64_t ocl = ofine0 + ((int64_t)kp->class_oo[cl][0] * OY + kp->class_oo[cl][1]) * OZ + kp->class_oo[cl][2];
    uint2 auxv[ACC ? MT : 1][ACC ? NT : 1];
    if constexpr (ACC) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) auxv[m][t] = *reinterpret_cast<const uint2*>(k.out + (ocl + ofine[m]) * k.out_vox_bytes + (t * 16 + g * 4) * 2);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      char* orow = k.out + (ocl + ofine[m]) * k.out_vox_bytes;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int c = t * 16 + g * 4;
        const float4 bi = *reinterpret_cast<const float4*>(epi + c);
        float val[4] = {acc[m][t][0] + bi.x, acc[m][t][1] + bi.y, acc[m][t][2] + bi.z, acc[m][t][3] + bi.w};
        if constexpr (STATS) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { ssum[t][r] += val[r]; ssq[t][r] += val[r] * val[r]; }
        }
        if constexpr (EVAL) {  // scale / shift (1 / 0 without them) and the activation as a slope for negative values: PReLU alpha, ReLU 0, none 1
          const float4 sc = *reinterpret_cast<const float4*>(epi + NT * 16 + c), sh = *reinterpret_cast<const float4*>(epi + 2 * NT * 16 + c);
          val[0] = val[0] * sc.x + sh.x; val[1] = val[1] * sc.y + sh.y; val[2] = val[2] * sc.z + sh.z; val[3] = val[3] * sc.w + sh.w;
#pragma unroll
          for (int r = 0; r < 4; ++r) val[r] = val[r] > 0.f ? val[r] : slope * val[r];
        }
        if constexpr (ACC) {
          const uint2 a = auxv[m][t];
          val[0] += __uint_as_float(a.x << 16); val[1] += __uint_as_float(a.x & 0xffff0000u); val[2] += __uint_as_float(a.y << 16); val[3] += __uint_as_float(a.y & 0xffff0000u);
        }
        st4(reinterpret_cast<bf16_t*>(orow + c * 2), make_float4(val[0], val[1], val[2], val[3]));
      }
    }
  }

  if constexpr (STATS) {  // per-channel sum / sum of squares of this workgroup's outputs -> the layer's sharded statistics (fixed-point atomics: order-independent; cconv.hip)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = ssum[t][r], q2 = ssq[t][r];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q2 += __shfl_xor(q2, o, 64); }
        if (l15 == 0) {
          sred[wave * (2 * NT * 16) + t * 16 + g * 4 + r] = s;
          sred[wave * (2 * NT * 16) + NT * 16 + t * 16 + g * 4 + r] = q2;
        }
      }
    __syncthreads();
    double* st = kp->stats + (int64_t)(blockIdx.x % VSSEG_STAT_SHARDS) * 2 * kp->stats_stride;
    for (int i = tid; i < 2 * NT * 16; i += TC_THREADS) {
      const int which = i / (NT * 16), c = i - which * NT * 16;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < TC_WAVES; ++w) v += sred[w * (2 * NT * 16) + i];
      if (c < kp->cout) vsseg_fx_add(&st[which * kp->stats_stride + c], (double)v, VSSEG_FX_STAT, kp->fxflag);
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------
static const char* tc_check(const vsseg_igemm_desc* d, int omin[3]) {
  if (d->in.dtype != VSSEG_BF16 || d->out.dtype != VSSEG_BF16) return "input and output must be bf16";
  if (d->in.ptr2 || d->out.ptr2) return "one-part tensors only";
  if (d->ck != 48 && d->ck != 64) return "48 or 64 input channels";
  if (d->nchunks != 1 || d->in.c != d->ck || d->in.pitch % 8 || ((uintptr_t)d->in.ptr & 15)) return "input must be one chunk of 16-byte aligned voxel rows";
  if (d->nt != TC_NT || d->out.c != 48 || (d->out.pitch & 3) || ((uintptr_t)d->out.ptr & 7)) return "output must be 48 channels (nt = 3) with 8-byte aligned rows";
  if (d->mtw != 16 || d->tile[0] != TC_TX || d->tile[1] != TC_TY || d->tile[2] != TC_TZ) return "tile must be 4x8x8 (mtw = 16)";
  if (d->class_split != 8 || d->nsplit != 8) return "needs the eight parity classes of a stride-(2,2,2) transition (class_split = nsplit = 8)";
  if (d->ksteps != tc_ksmax(d->ck / 8)) return "ksteps must be ceil(8 * (ck / 8) / 4)";
  for (int a = 0; a < 3; ++a) {
    if (d->is[a] != 1 || d->os[a] != 2 || d->oo[a] != 0) return "is = 1, os = 2, oo = 0 on every axis";
    if (d->q[a] % d->tile[a]) return "lattice extent is not a multiple of the tile";
  }
  if (d->q[0] != d->in.x || d->q[1] != d->in.y || d->q[2] != d->in.z || d->out.x != 2 * d->q[0] || d->out.y != 2 * d->q[1] || d->out.z != 2 * d->q[2] || d->out.n != d->in.n) return "output must be (2x, 2y, 2z) of the input lattice";
  if (d->ntaps < 1 || d->ntaps > VSSEG_MAX_TAPS) return "ntaps out of range";
  for (int a = 0; a < 3; ++a) {
    int lo = d->tap_off[0][a], hi = lo;
    for (int t = 1; t < d->ntaps; ++t) { lo = min(lo, d->tap_off[t][a]); hi = max(hi, d->tap_off[t][a]); }
    if (hi - lo != 1 || (lo != 0 && lo != -1)) return "the union of the class taps must span two neighbouring coarse voxels per axis";
    omin[a] = lo;
  }
  for (int c = 0; c < 8; ++c) {
    if (d->class_ntaps[c] < 1 || d->class_ntaps[c] > 8) return "class tap count";
    for (int t = 0; t < d->class_ntaps[c]; ++t)
      if (d->class_tap[c][t] < 0 || d->class_tap[c][t] >= d->ntaps) return "class tap index";
    for (int a = 0; a < 3; ++a)
      if (d->class_oo[c][a] < 0 || d->class_oo[c][a] > 1) return "class offset outside the output stride";
  }
  if (d->cout_mod > 0 || d->in_gate || d->res_tiles || d->in1) return "z-folded / marching-kernel-only features";
  if (d->res_mode != VSSEG_RES_NONE) return "residual epilogues are not supported (plain, statistics, accumulate)";
  if (d->stats && d->accumulate) return "statistics combined with accumulate";
  if ((d->stats || d->accumulate) && (d->scale || d->act != VSSEG_ACT_NONE)) return "the eval affine / activation combined with statistics or accumulate";
  if (d->act == VSSEG_ACT_SIGMOID) return "sigmoid epilogue";
  if ((d->scale == nullptr) != (d->shift == nullptr)) return "scale without shift";
  if ((int64_t)d->in.n * d->q[0] * d->q[1] * d->q[2] / 256 >= (1ll << 24)) return "too many tiles";
  return nullptr;
}

int vsseg_tconv_lds_bytes(const vsseg_igemm_desc* d) {
  int omin[3];
  const char* why = tc_check(d, omin);
  if (why) { vsseg_set_error("vsseg_igemm: depth -8 (transition kernel) not applicable: %s", why); return VSSEG_EINVAL; }
  return tc_lds_bytes(d->ck / 8);
}

template <int CG, int MODE> static int tc_launch_mode(const TconvK& k, hipStream_t s) {
  static bool attr_set_dev[16] = {}; bool& attr_set = vsseg_dev_once(attr_set_dev);  // per device: the LDS opt-in is a per-device function attribute
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&tconv_kernel<CG, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  int gx = 256;  // one workgroup per CU, whole rounds of the 8 XCDs
  const int need = (k.tiles + 7) / 8 * 8;
  if (gx > need) gx = need;
  hipLaunchKernelGGL((tconv_kernel<CG, MODE>), dim3((unsigned)gx), dim3(TC_THREADS), tc_lds_bytes(CG), s, k);
  VSSEG_LAUNCH_CHECK("vsseg_igemm (transition kernel)");
  return VSSEG_OK;
}
template <int CG> static int tc_launch(const TconvK& k, hipStream_t s) {
  if (k.stats) return tc_launch_mode<CG, 1>(k, s);
  if (k.accumulate) return tc_launch_mode<CG, 2>(k, s);
  if (k.scale || k.act != VSSEG_ACT_NONE) return tc_launch_mode<CG, 3>(k, s);
  return tc_launch_mode<CG, 0>(k, s);
}

int vsseg_tconv_launch(const vsseg_igemm_desc* d, const void* zeros, hipStream_t s) {
  int omin[3];
  const char* why = tc_check(d, omin);
  if (why) { vsseg_set_error("vsseg_igemm: depth -8 (transition kernel) not applicable: %s", why); return VSSEG_EINVAL; }
  TconvK k{};
  auto magic = [](int dv) { return dv <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)dv - 1) / (unsigned)dv); };
  k.in = reinterpret_cast<const char*>(d->in.ptr);
  k.out = reinterpret_cast<char*>(d->out.ptr);
  k.wpack = reinterpret_cast<const char*>(d->wpack);
  k.bias = d->bias; k.bias2 = d->bias2; k.scale = d->scale; k.shift = d->shift; k.alpha = d->alpha;
  k.stats = d->stats; k.stats_stride = d->stats_stride;
  VSSEG_FX_FLAG(fxflag_, "vsseg_igemm (transition kernel)");
  k.fxflag = fxflag_;
  k.zeros = zeros;
  k.in_vox_bytes = d->in.pitch * 2; k.out_vox_bytes = d->out.pitch * 2;
  k.accumulate = d->accumulate ? 1 : 0; k.act = d->act; k.cout = d->out.c;
  k.X = d->q[0]; k.Y = d->q[1]; k.Z = d->q[2];
  for (int a = 0; a < 3; ++a) k.omin[a] = omin[a];
  k.ntx = k.X / TC_TX; k.nty = k.Y / TC_TY; k.ntz = k.Z / TC_TZ;
  k.mg_tx = magic(k.ntx); k.mg_ty = magic(k.nty); k.mg_tz = magic(k.ntz);
  k.tiles = d->in.n * k.ntx * k.nty * k.ntz;
  k.per_xcd = (k.tiles + 7) / 8;
  for (int c = 0; c < 8; ++c) {
    k.class_ntaps[c] = d->class_ntaps[c];
    for (int a = 0; a < 3; ++a) k.class_oo[c][a] = d->class_oo[c][a];
    for (int t = 0; t < 8; ++t) {
      const int ti = t < d->class_ntaps[c] ? d->class_tap[c][t] : d->class_tap[c][0];
      k.class_hoff[c][t] = ((d->tap_off[ti][0] - omin[0]) * TC_HY + (d->tap_off[ti][1] - omin[1])) * TC_HZ + (d->tap_off[ti][2] - omin[2]);
    }
  }
  {  // most taps first (stable): 8, 4, 4, 4, 2, 2, 2, 1
    bool used[8] = {false, false, false, false, false, false, false, false};
    for (int i = 0; i < 8; ++i) {
      int best = -1;
      for (int c = 0; c < 8; ++c)
        if (!used[c] && (best < 0 || d->class_ntaps[c] > d->class_ntaps[best])) best = c;
      used[best] = true;
      k.order[i] = best;
    }
  }
  return d->ck == 64 ? tc_launch<8>(k, s) : tc_launch<6>(k, s);
}
