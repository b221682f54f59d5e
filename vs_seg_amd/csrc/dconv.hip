// Deep-level convolution: the SMALL launches of vsseg_igemm — every convolution, transposed convolution and data gradient of levels 3-5 of the 2.5D U-Net
// (48x16x64 ... 12x4x16 voxels per sample, 40-160 channels: ref:params/networks/nets/unet2d5_spvPA.py:56-89, SURVEY §8a rows 11-35 and their autograd) and the
// stride-(2,2,2) transitions around them — as launch plans with depth = -7.
//
// Why another kernel.  On these tensors the general kernel (igemm_kernel.h) and the compute kernel (cconv.hip) are neither HBM- nor MFMA-bound but LATENCY-bound
// (profiles/r04_*: 15-180 us per launch against 1-30 us of roofline, 4.7 ms of a 28 ms step in 154 launches): a launch is a few dozen workgroups, and each walks a serial
// chain of (channel chunk -> LDS-DMA of halo + packed weights -> barrier -> K loop) stages of ~2-3 us.  A level-5 convolution has 0.35-0.5 MB of weights for 0.15-0.6 MB
// of activations, so staging the WEIGHTS through LDS chunk by chunk is what serialises.  Here:
//
//   * a workgroup (4 waves) owns 16*MT output-lattice voxels x NT 16-channel tiles; the halo of its input with ALL channels of a chunk (usually the only one) is
//     loaded ONCE through registers into LDS (every load in flight at once; the voxel stride is padded to an odd number of 16-byte units: conflict-free operand reads)
//   * the K loop is split ACROSS the waves by K-step (tap, 32-channel group): wave w multiplies K-steps w, w+4, ...  against all MT x NT accumulator tiles.  A weight
//     fragment is therefore read by exactly one wave of the workgroup, straight from L2 into registers (coalesced 1 KiB rows of the packed weights, a ring of 3 K-steps
//     in flight) — no LDS staging of weights, no barrier inside the K loop, and the activations of a K-step are read from LDS once per workgroup
//   * the four partial accumulators are summed through LDS in wave order (fixed order: run-to-run bit-identical), then the usual epilogue (bias, BatchNorm statistics as
//     fixed-point atomics, eval affine, activation, residual / accumulate / ReLU mask / gated add) runs on the sums
//   * all output-parity classes of a transposed convolution / strided data gradient run in ONE workgroup, one after the other, from the same halo (class_split)
//
// Same packed weights and K order (tap, 8-channel group) as the general kernel (planner.pack_map); the sum over K is split four ways, so results agree with it to fp32
// rounding, not bit for bit (tests/test_gpu_ops.py::test_deep_kernel_*).
#include "dconv.h"
#include <type_traits>

constexpr int DC_THREADS = 256, DC_WAVES = 4;
constexpr int dc_red_tiles(int mt, int nt) { return mt * nt <= 24 ? mt * nt : (mt * nt + 1) / 2; }  // accumulator tiles per slab round (4 KiB of LDS each)
// K-steps of weight fragments in flight per wave.  A fragment comes from L2 (~0.7-1 us under load) and a K-step is MT*NT MFMAs of 16 cycles (80 ns at 4 x 3 tiles): with a
// ring of 3 the level-5 launches waited ~0.5 us per K-step (first version: 13 us per launch, 10 of them in this wait).  One wave per SIMD: 512 registers per lane.
constexpr int dc_ring(int mt, int nt) {
  // fragments and operands must sit in the 256 architectural VGPRs (the accumulators go to the AGPRs): ring + two operand sets + ~70 for addressing
  int r = (230 - 70 - mt * 8 - (mt * nt >= 32 ? 32 : 0)) / (nt * 4);
  if (r * nt > 60) r = 60 / nt;  // vmcnt counts at most 63 outstanding operations
  return r < 3 ? 3 : (r > 16 ? 16 : r);
}

struct DconvK {
  const char *in0, *in1;  // in1: channels >= in_csplit, biased by -in_csplit channels
  char *out0, *out1;
  const char *aux0, *aux1;
  const float* gate;
  const char* wpack;
  const float *bias, *bias2, *scale, *shift, *alpha;
  double* stats;
  unsigned* fxflag;
  unsigned long long* prof;  // tuning build (-DVSSEG_DC_PROF): cycle counters of workgroup 0 / wave 0 per phase
  const void* zeros;
  int in_csplit, in_vox_bytes;
  int out_csplit, out_vox_bytes, out_f32, cout;
  int aux_csplit, aux_vox_bytes, aux_mode;  // 0 none, 1 accumulate, 2 residual add, 3 ReLU mask, 4 gated add
  int act, stats_stride;
  int X, Y, Z, OX, OY, OZ;
  int q[3], is[3], os[3], oo[3];
  int tl[3];      // log2 of the tile extents
  int ntile[3];
  unsigned mg_t2, mg_t1, mg_t0;
  int halo[3], omin[3];
  unsigned mg_h2, mg_h1, mg_cgs, mg_cgp;
  int ntaps;
  int tapofs[VSSEG_MAX_TAPS];  // halo-voxel offset of each tap relative to the voxel's own halo position
  int ck, cgs, nchunks, ksteps, vs;  // vs: bytes per halo voxel in LDS
  int nclass;                  // 0: one lattice class (the descriptor's taps, grid.y = output-channel split; the waves split the K-steps); 2..8: every class in this workgroup, whole classes per wave
  int class_ntaps[8], class_tap[8][8], class_oo[8][3];
  int wave_ncls[DC_WAVES], wave_cls[DC_WAVES][8];  // class mode: the classes each wave computes (balanced by tap count on the host)
  int lds_epi, lds_stat, lds_halo, lds_red;
};

__device__ __forceinline__ unsigned dc_div(unsigned n, unsigned magic) { return magic ? __umulhi(n, magic) : n; }  // magic 0: divisor 1

template <int MT, int NT>
__global__ __launch_bounds__(DC_THREADS) void dconv_kernel(const DconvK k) {
  constexpr int TILES = MT * NT, RT = dc_red_tiles(MT, NT), ROUNDS = (TILES + RT - 1) / RT, RING = dc_ring(MT, NT);
  constexpr int MR = RT / NT;  // M-tiles per slab round
  static_assert(ROUNDS <= 2 && RT % NT == 0, "slab rounds hold whole M-tiles");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* ktab = reinterpret_cast<int*>(smem);                        // [class][K-steps x 4 K-groups] -> LDS byte offset of the group's 16 bytes relative to the voxel
  float* epi = reinterpret_cast<float*>(smem + k.lds_epi);         // bias | scale | shift, NT*16 each
  float* sred = reinterpret_cast<float*>(smem + k.lds_stat);       // [4 waves][2][NT*16]
  char* halo = smem + k.lds_halo;
  char* red = smem + k.lds_red;                                    // [4 waves][RT][64 lanes][16 bytes]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, l15 = lane & 15;
  // the tables of the argument block that are indexed at run time are read THROUGH the kernarg segment (constant memory): indexing the by-value copy
  // dynamically made hipcc spill those arrays to scratch (272 bytes per lane)
  const DconvK* __restrict__ kp = (const DconvK*)__builtin_amdgcn_kernarg_segment_ptr();
  const bool classes = k.nclass > 0;
  const int row0 = classes ? 0 : (int)blockIdx.y;
  const int c_base = row0 * NT * 16;
#ifdef VSSEG_DC_PROF
  unsigned long long pt[6];
  pt[0] = __builtin_readcyclecounter();
#define DC_MARK(i) pt[i] = __builtin_readcyclecounter()
#else
#define DC_MARK(i)
#endif
  // Small tables, one element per lane (read with shuffles: no dependent memory round trip per K-group when the K-group table is built):
  // tap -> halo-voxel offset, (class, tap) -> tap of the union table, class -> tap count.  Issued first: they are back long before they are used.
  const int tapofs_l = kp->tapofs[lane < VSSEG_MAX_TAPS ? lane : 0];
  const int ctap_l = kp->class_tap[lane >> 3][lane & 7];
  const int cnt_l = kp->class_ntaps[lane & 7];
  // per-channel epilogue constants of this workgroup's channel tiles (class mode: all channels): loads issued here, stored to LDS behind the halo DMAs
  const float alpha = (k.act == VSSEG_ACT_PRELU && k.alpha) ? *k.alpha : 0.f;
  float e_bias = 0.f, e_scale = 1.f, e_shift = 0.f;
  if (tid < NT * 16) {
    const int c = c_base + tid;
    const bool ok = c < k.cout;
    e_bias = ((ok && k.bias) ? k.bias[c] : 0.f) + ((ok && k.bias2) ? k.bias2[c] : 0.f);
    e_scale = (ok && k.scale) ? k.scale[c] : 1.f;
    e_shift = (ok && k.scale) ? k.shift[c] : 0.f;
  }

  // ---- this workgroup's tile of the output lattice (neighbouring tiles -> the same XCD: they share halo lines in its L2) ----
  int q0[3], smp;
  {
    unsigned b = (unsigned)vsseg_xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
    unsigned t = dc_div(b, k.mg_t2); q0[2] = (int)(b - t * k.ntile[2]) << k.tl[2]; b = t;
    t = dc_div(b, k.mg_t1); q0[1] = (int)(b - t * k.ntile[1]) << k.tl[1]; b = t;
    t = dc_div(b, k.mg_t0); q0[0] = (int)(b - t * k.ntile[0]) << k.tl[0]; smp = (int)t;
  }
  const int g0x = q0[0] * k.is[0] + k.omin[0], g0y = q0[1] * k.is[1] + k.omin[1], g0z = q0[2] * k.is[2] + k.omin[2];
  const int hvox = k.halo[0] * k.halo[1] * k.halo[2];
  const int64_t in_base = (int64_t)smp * k.X * k.Y * k.Z;

  // ---- halo of one channel chunk: HBM / L2 -> LDS by LDS-DMA (16 bytes per lane, 1 KiB per instruction, every instruction of the chunk in flight at once; no registers).
  //      LDS slot s = voxel * cgp + piece, cgp = cgs | 1: the DMA writes LDS in lane order, so the padding slot of a voxel (and everything outside the tensor: the
  //      convolution's zero padding) is simply fetched from the zero page.  ISSUE only: the caller waits (counted) after it has issued more work ----
  auto issue_halo = [&](int ch) __attribute__((always_inline)) {
    const int c0 = ch * k.ck, cgp = k.cgs | 1, slots = hvox * cgp, rows = (slots + 63) >> 6;
    for (int row = wave; row < rows; row += DC_WAVES) {
      const unsigned sl = (unsigned)(row * 64 + lane);
      const unsigned hv = dc_div(sl, k.mg_cgp);
      const int pc = (int)(sl - hv * cgp);
      unsigned t = dc_div(hv, k.mg_h2); const int hz = (int)(hv - t * k.halo[2]);
      const unsigned hx = dc_div(t, k.mg_h1); const int hy = (int)(t - hx * k.halo[1]);
      const int gx = g0x + (int)hx, gy = g0y + hy, gz = g0z + hz;
      const bool ok = ((int)sl < slots) & (pc < k.cgs) & ((unsigned)gx < (unsigned)k.X) & ((unsigned)gy < (unsigned)k.Y) & ((unsigned)gz < (unsigned)k.Z);
      const int c = c0 + pc * 8;
      const char* src = (c >= k.in_csplit ? k.in1 : k.in0) + (in_base + ((int64_t)gx * k.Y + gy) * k.Z + gz) * k.in_vox_bytes + c * 2;
      vsseg_dma16(ok ? (const void*)src : k.zeros, halo + row * 1024);
    }
  };

  // ---- K loop over `cnt` K-steps of this wave: weight fragments straight from L2 (RING steps in flight, `wstep` bytes apart), operand fragments from the halo.
  //      k_preload issues the first RING steps' fragments (always RING * NT loads, from clamped steps) — BEFORE the halo has landed, so that both latencies overlap;
  //      k_run multiplies; the operand fragments of step i+1 and the K-group offset of step i+2 are read while step i multiplies ----
  f32x4 acc[MT][NT];
  bf16x8 wb[RING][NT];
  auto k_preload = [&](const char* wsrc, int64_t wstep, int cnt) __attribute__((always_inline)) {
    if (cnt <= 0) wsrc = k.wpack + lane * 16;  // a wave without K-steps (fewer than 4 in the pass) still issues its RING * NT loads (wait_halo counts them): from the first fragments
    const int last = cnt > 0 ? cnt - 1 : 0;
#pragma unroll
    for (int dd = 0; dd < RING; ++dd) {
      const int sc = dd < last ? dd : last;
#pragma unroll
      for (int t = 0; t < NT; ++t) wb[dd][t] = *reinterpret_cast<const bf16x8*>(wsrc + sc * wstep + t * 1024);
    }
  };
  int abase[MT];
  auto k_run = [&](const char* wsrc, int64_t wstep, const int* kt, int ktstep, int cnt) __attribute__((always_inline)) {
    if (cnt <= 0) return;
    // The steady-state loop has NO branch inside: every load is unconditional from a clamped K-step (the last groups re-fetch the final fragment, harmlessly).  With
    // `if (i + RING < cnt)` around the refills hipcc's wait-count pass lost the order of the pending loads at the control-flow merges and put vmcnt(4..0) in front of
    // every K-step — i.e. it waited for the refill it had just issued.
    // APF: the operand fragments of step i+1 are read while step i multiplies (a second register set); with 8 M-tiles a step is >= 8 * NT MFMAs and hipcc interleaves the
    // step's own reads with them — the second set would only cost 32 registers there.
    constexpr bool APF = MT <= 4;
    const int last = cnt - 1;
    bf16x8 a[MT], an[APF ? MT : 1];
    int koff1;
    if constexpr (APF) {
      const int koff0 = kt[0];
#pragma unroll
      for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const bf16x8*>(halo + abase[m] + koff0);
      koff1 = kt[(1 < last ? 1 : last) * ktstep];
    } else {
      koff1 = kt[0];
    }
    int i = 0;
    for (; i + RING <= cnt; i += RING) {
#pragma unroll
      for (int dd = 0; dd < RING; ++dd) {
        const int ii = i + dd;
        const int nx = APF ? ii + 2 : ii + 1;
        const int koff2 = kt[(nx < last ? nx : last) * ktstep];
        if constexpr (APF) {
#pragma unroll
          for (int m = 0; m < MT; ++m) an[m] = *reinterpret_cast<const bf16x8*>(halo + abase[m] + koff1);
        } else {
#pragma unroll
          for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const bf16x8*>(halo + abase[m] + koff1);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[dd][t], a[m], acc[m][t], 0, 0, 0);
        const int rs = ii + RING < last ? ii + RING : last;
#pragma unroll
        for (int t = 0; t < NT; ++t) wb[dd][t] = *reinterpret_cast<const bf16x8*>(wsrc + (int64_t)rs * wstep + t * 1024);
        if constexpr (APF) {
#pragma unroll
          for (int m = 0; m < MT; ++m) a[m] = an[m];
        }
        koff1 = koff2;
      }
    }
    const int rem = cnt - i;  // < RING: the fragments of these K-steps are already in the ring
#pragma unroll
    for (int dd = 0; dd < RING - 1; ++dd) {
      if (dd < rem) {
        const int ii = i + dd;
        const int nx = APF ? ii + 2 : ii + 1;
        const int koff2 = kt[(nx < last ? nx : last) * ktstep];
        if constexpr (APF) {
#pragma unroll
          for (int m = 0; m < MT; ++m) an[m] = *reinterpret_cast<const bf16x8*>(halo + abase[m] + koff1);
        } else {
#pragma unroll
          for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const bf16x8*>(halo + abase[m] + koff1);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[dd][t], a[m], acc[m][t], 0, 0, 0);
        if constexpr (APF) {
#pragma unroll
          for (int m = 0; m < MT; ++m) a[m] = an[m];
        }
        koff1 = koff2;
      }
    }
  };
  // the DMAs of issue_halo are older than the RING * NT fragment loads of k_preload (VMEM operations of a wave retire in issue order): this wave's halo pieces have
  // landed once at most that many operations are outstanding.  (hipcc does not count the inline-assembly DMAs: its own waits can only come out longer)
  auto wait_halo = [&]() __attribute__((always_inline)) {
    constexpr int N = RING * NT;
    static_assert(N <= 63, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
  };
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // ================= prologue: everything that touches memory is issued first =================
  const int nks0 = (k.ntaps * k.cgs + 3) >> 2;
  const int cnt0 = nks0 > wave ? (nks0 - 1 - wave) / DC_WAVES + 1 : 0;   // one class: K-steps wave, wave + 4, ...
  const int ncls_w = classes ? kp->wave_ncls[wave] : 0;
  const int cl_first = ncls_w ? kp->wave_cls[wave][0] : 0;  // (a wave without a class — 2 or 3 classes over 4 waves — preloads nothing: count 0 below)
  DC_MARK(1);
  issue_halo(0);
  if (!classes) k_preload(k.wpack + ((int64_t)(row0 * k.nchunks) * k.ksteps + wave) * (NT * 1024) + lane * 16, (int64_t)DC_WAVES * NT * 1024, cnt0);
  else k_preload(k.wpack + (int64_t)cl_first * k.ksteps * (NT * 1024) + lane * 16, (int64_t)NT * 1024, ncls_w ? (__shfl(cnt_l, cl_first) * k.cgs + 3) >> 2 : 0);
  if (tid < NT * 16) { epi[tid] = e_bias; epi[NT * 16 + tid] = e_scale; epi[2 * NT * 16 + tid] = e_shift; }
  // K-group tables: K-group p = ks*4 + g of a class -> (tap p / cgs, channel group p % cgs) -> byte offset inside the halo relative to the voxel
  {
    const int per = k.ksteps * 4, ncl = classes ? k.nclass : 1;
    for (int e0 = 0; e0 < per * ncl; e0 += DC_THREADS) {  // (uniform trip count: the shuffles below need every lane)
      const int e = e0 + tid;
      const int ec = e < per * ncl ? e : 0;
      const int cl = ec / per, p = ec - cl * per;
      const int ntaps_c = classes ? __shfl(cnt_l, cl) : k.ntaps;
      const unsigned tap = dc_div((unsigned)p, k.mg_cgs);
      const int cg = p - (int)tap * k.cgs;
      const int tc = (int)tap < ntaps_c ? (int)tap : 0;
      const int ti = classes ? __shfl(ctap_l, cl * 8 + (tc & 7)) : tc;
      const int to = __shfl(tapofs_l, ti & 31);
      if (e < per * ncl) ktab[e] = (int)tap < ntaps_c ? to * k.vs + cg * 16 : 0;  // padded K-groups: zero weights times the voxel's own (finite) data
    }
  }
  // this lane's voxel of every M-tile: v = m*16 + l15 -> (vx, vy, vz) inside the tile
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int v = m * 16 + l15;
    const int vz = v & ((1 << k.tl[2]) - 1), vy = (v >> k.tl[2]) & ((1 << k.tl[1]) - 1), vx = v >> (k.tl[2] + k.tl[1]);
    abase[m] = ((vx * k.is[0] * k.halo[1] + vy * k.is[1]) * k.halo[2] + vz * k.is[2]) * k.vs;
  }
  float ssum[NT][4], ssq[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) { ssum[t][r] = 0.f; ssq[t][r] = 0.f; }
  zero_acc();
  DC_MARK(2);

  // ---- epilogue of M-tile m, channel tile T: its `parts` partial sums sit in the slabs at position j of waves w0, w0+1, ... ----
  auto tile_epilogue = [&](auto tc, int m, int j, int w0, int parts, int oo0, int oo1, int oo2) __attribute__((always_inline)) {
    constexpr int T = decltype(tc)::value;
    f32x4 s = *reinterpret_cast<const f32x4*>(red + ((w0 * RT + j) * 64 + lane) * 16);
    for (int w = 1; w < parts; ++w) {  // fixed order: run-to-run bit-identical
      const f32x4 p = *reinterpret_cast<const f32x4*>(red + (((w0 + w) * RT + j) * 64 + lane) * 16);
      s[0] += p[0]; s[1] += p[1]; s[2] += p[2]; s[3] += p[3];
    }
    const int v = m * 16 + l15;
    const int vz = v & ((1 << k.tl[2]) - 1), vy = (v >> k.tl[2]) & ((1 << k.tl[1]) - 1), vx = v >> (k.tl[2] + k.tl[1]);
    const int qx = q0[0] + vx, qy = q0[1] + vy, qz = q0[2] + vz;
    const int ox = qx * k.os[0] + oo0, oy = qy * k.os[1] + oo1, oz = qz * k.os[2] + oo2;
    const bool vok = (qx < k.q[0]) & (qy < k.q[1]) & (qz < k.q[2]) & (ox < k.OX) & (oy < k.OY) & (oz < k.OZ);
    const int cl16 = T * 16 + g * 4, c = c_base + cl16;
    const float4 bi = *reinterpret_cast<const float4*>(epi + cl16);
    float val[4] = {s[0] + bi.x, s[1] + bi.y, s[2] + bi.z, s[3] + bi.w};
    if (!vok | (c >= k.cout)) return;  // (channels past the last real one of a padded channel tile: nothing to add, nothing to load — the auxiliary row ends at cout)
    if (k.stats) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { ssum[T][e] += val[e]; ssq[T][e] += val[e] * val[e]; }
    }
    if (k.scale) {
      const float4 sc = *reinterpret_cast<const float4*>(epi + NT * 16 + cl16), sh = *reinterpret_cast<const float4*>(epi + 2 * NT * 16 + cl16);
      val[0] = val[0] * sc.x + sh.x; val[1] = val[1] * sc.y + sh.y; val[2] = val[2] * sc.z + sh.z; val[3] = val[3] * sc.w + sh.w;
    }
    if (k.act == VSSEG_ACT_PRELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) val[e] = val[e] > 0.f ? val[e] : alpha * val[e];
    } else if (k.act == VSSEG_ACT_RELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) val[e] = fmaxf(val[e], 0.f);
    } else if (k.act == VSSEG_ACT_SIGMOID) {
#pragma unroll
      for (int e = 0; e < 4; ++e) val[e] = 1.f / (1.f + __expf(-val[e]));
    }
    const int64_t ovox = (((int64_t)smp * k.OX + ox) * k.OY + oy) * k.OZ + oz;
    if (k.aux_mode) {
      const float4 av = ld4(reinterpret_cast<const bf16_t*>((c >= k.aux_csplit ? k.aux1 : k.aux0) + ovox * k.aux_vox_bytes + c * 2));
      if (k.aux_mode == 3) {
        val[0] = av.x > 0.f ? val[0] : 0.f; val[1] = av.y > 0.f ? val[1] : 0.f; val[2] = av.z > 0.f ? val[2] : 0.f; val[3] = av.w > 0.f ? val[3] : 0.f;
      } else if (k.aux_mode == 4) {
        const float gt = 1.f + k.gate[ovox];
        val[0] = vsseg_fma_unpacked(av.x, gt, val[0]); val[1] = vsseg_fma_unpacked(av.y, gt, val[1]);  // (not v_pk_fma_f32 op_sel: common.h)
        val[2] = vsseg_fma_unpacked(av.z, gt, val[2]); val[3] = vsseg_fma_unpacked(av.w, gt, val[3]);
      } else {
        val[0] += av.x; val[1] += av.y; val[2] += av.z; val[3] += av.w;
      }
    }
    if (k.out_f32) {
      float* op = reinterpret_cast<float*>((c >= k.out_csplit ? k.out1 : k.out0) + ovox * k.out_vox_bytes) + c;
      if (c + 3 < k.cout) st4(op, make_float4(val[0], val[1], val[2], val[3]));
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c + e < k.cout) op[e] = val[e];
      }
    } else {
      bf16_t* op = reinterpret_cast<bf16_t*>((c >= k.out_csplit ? k.out1 : k.out0) + ovox * k.out_vox_bytes) + c;
      if (c + 3 < k.cout) st4(op, make_float4(val[0], val[1], val[2], val[3]));
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c + e < k.cout) op[e] = f2bf(val[e]);
      }
    }
  };
  // all NT channel tiles of M-tile m (independent: their slab reads, auxiliary loads and stores overlap)
  auto mtile_epilogue = [&](int m, int r, int w0, int parts, int oo0, int oo1, int oo2) __attribute__((always_inline)) {
    const int j0 = (m - r * MR) * NT;
    tile_epilogue(std::integral_constant<int, 0>{}, m, j0, w0, parts, oo0, oo1, oo2);
    if constexpr (NT > 1) tile_epilogue(std::integral_constant<int, 1>{}, m, j0 + 1, w0, parts, oo0, oo1, oo2);
    if constexpr (NT > 2) tile_epilogue(std::integral_constant<int, 2>{}, m, j0 + 2, w0, parts, oo0, oo1, oo2);
    if constexpr (NT > 3) tile_epilogue(std::integral_constant<int, 3>{}, m, j0 + 3, w0, parts, oo0, oo1, oo2);
    if constexpr (NT > 4) tile_epilogue(std::integral_constant<int, 4>{}, m, j0 + 4, w0, parts, oo0, oo1, oo2);
    if constexpr (NT > 5) tile_epilogue(std::integral_constant<int, 5>{}, m, j0 + 5, w0, parts, oo0, oo1, oo2);
  };
  // this wave's accumulators of slab round R -> its slab (registers are indexed statically: this part is unrolled; the epilogues are loops)
  auto write_round = [&](auto rc) __attribute__((always_inline)) {
    constexpr int R = decltype(rc)::value;
#pragma unroll
    for (int i = R * RT; i < (R + 1) * RT && i < TILES; ++i)
      *reinterpret_cast<f32x4*>(red + ((wave * RT + (i - R * RT)) * 64 + lane) * 16) = acc[i / NT][i % NT];
  };

  if (!classes) {
    // ================= one lattice class: the waves split the K-steps (wave w: K-steps w, w+4, ...), partial sums meet in the slabs =================
    for (int ch = 0; ch < k.nchunks; ++ch) {
      const char* wsrc = k.wpack + ((int64_t)(row0 * k.nchunks + ch) * k.ksteps + wave) * (NT * 1024) + lane * 16;
      if (ch > 0) {
        __syncthreads();  // every wave finished the previous chunk's K loop
        issue_halo(ch);
        k_preload(wsrc, (int64_t)DC_WAVES * NT * 1024, cnt0);
      }
      wait_halo();
      __syncthreads();    // halo (and the tables) visible
      DC_MARK(3);
      k_run(wsrc, (int64_t)DC_WAVES * NT * 1024, ktab + wave * 4 + g, DC_WAVES * 4, cnt0);
    }
    DC_MARK(4);
    for (int r = 0; r < ROUNDS; ++r) {
      __syncthreads();  // K loops done with the halo (the slabs overlap it) / previous round's slabs consumed
      if (r == 0) write_round(std::integral_constant<int, 0>{});
      else write_round(std::integral_constant<int, ROUNDS - 1>{});
      __syncthreads();
      for (int m = r * MR + wave; m < (r + 1) * MR && m < MT; m += DC_WAVES)  // M-tile m belongs to wave (m - r * MR) % 4
        mtile_epilogue(m, r, 0, DC_WAVES, k.oo[0], k.oo[1], k.oo[2]);
    }
  } else {
    // ================= all parity classes: the halo is loaded once, every wave computes WHOLE classes (all K-steps, all tiles): no partial sums, no barriers =================
    wait_halo();
    __syncthreads();
    DC_MARK(3);
    for (int ci = 0; ci < ncls_w; ++ci) {
      const int cl = kp->wave_cls[wave][ci];
      const int nks = (__shfl(cnt_l, cl) * k.cgs + 3) >> 2;
      const char* wsrc = k.wpack + (int64_t)cl * k.ksteps * (NT * 1024) + lane * 16;
      if (ci > 0) { zero_acc(); k_preload(wsrc, (int64_t)NT * 1024, nks); }
      k_run(wsrc, (int64_t)NT * 1024, ktab + cl * k.ksteps * 4 + g, 4, nks);
      const int oo0 = kp->class_oo[cl][0], oo1 = kp->class_oo[cl][1], oo2 = kp->class_oo[cl][2];
      for (int r = 0; r < ROUNDS; ++r) {  // through this wave's own slab (every lane reads back what it wrote): the epilogue stays a loop
        if (r == 0) write_round(std::integral_constant<int, 0>{});
        else write_round(std::integral_constant<int, ROUNDS - 1>{});
        for (int m = r * MR; m < (r + 1) * MR && m < MT; ++m) mtile_epilogue(m, r, wave, 1, oo0, oo1, oo2);
      }
    }
    DC_MARK(4);
  }
  DC_MARK(5);
#ifdef VSSEG_DC_PROF
  if (k.prof && blockIdx.x == 8 && blockIdx.y == 0 && tid == 0)
    for (int i = 0; i < 5; ++i) atomicAdd(&k.prof[i], pt[i + 1] - pt[i]);
#endif

  if (k.stats) {  // per-channel sum / sum of squares of this workgroup's voxels -> the layer's sharded statistics (fixed-point atomics: order-independent)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float s = ssum[t][e], q2 = ssq[t][e];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q2 += __shfl_xor(q2, o, 64); }
        if (l15 == 0) {
          sred[wave * (2 * NT * 16) + t * 16 + g * 4 + e] = s;
          sred[wave * (2 * NT * 16) + NT * 16 + t * 16 + g * 4 + e] = q2;
        }
      }
    __syncthreads();
    double* st = k.stats + (int64_t)((blockIdx.x + blockIdx.y * gridDim.x) % VSSEG_STAT_SHARDS) * 2 * k.stats_stride;
    for (int i = tid; i < 2 * NT * 16; i += DC_THREADS) {
      const int which = i / (NT * 16), c = c_base + i - which * NT * 16;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < DC_WAVES; ++w) v += sred[w * (2 * NT * 16) + i];
      if (c < k.cout) vsseg_fx_add(&st[which * k.stats_stride + c], (double)v, VSSEG_FX_STAT, k.fxflag);
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------
static int dc_log2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return (1 << l) == v ? l : -1;
}

struct DcGeom {
  int halo[3], omin[3], vs, cgs, lds_epi, lds_stat, lds_halo, lds_red, lds_total, rt;
};

static const char* dc_check(const vsseg_igemm_desc* d, DcGeom& gm) {
  if (d->in.dtype != VSSEG_BF16) return "input is not bf16";
  if (d->out.dtype != VSSEG_BF16 && d->out.dtype != VSSEG_F32) return "output dtype";
  if (d->mtw != 2 && d->mtw != 4 && d->mtw != 8) return "mtw (16-voxel tiles per workgroup) must be 2, 4 or 8";
  if (d->nt < 1 || d->nt > 6) return "nt must be 1..6";
  if (d->tile[0] * d->tile[1] * d->tile[2] != 16 * d->mtw) return "tile must hold 16 * mtw voxels";
  if (d->mtw * d->nt > 40) return "mtw * nt must not exceed 40 accumulator tiles";
  for (int a = 0; a < 3; ++a)
    if (dc_log2(d->tile[a]) < 0 || d->tile[a] > 128) return "tile extents must be powers of two <= 128";
  if (d->ntaps < 1 || d->ntaps > VSSEG_MAX_TAPS) return "ntaps out of range";
  if (d->ck < 8 || d->ck % 8 || d->nchunks < 1 || d->in.c != d->ck * d->nchunks) return "input channels must be nchunks x ck (ck a multiple of 8)";
  if (d->in.pitch % 8 || ((uintptr_t)d->in.ptr & 15) || ((uintptr_t)d->in.ptr2 & 15)) return "input must be 16-byte aligned voxel rows";
  if (d->in.ptr2 && (d->in.csplit % 8 || d->in.csplit <= 0 || d->in.csplit >= d->in.c)) return "input split must be a multiple of 8 channels";
  if (d->out.ptr2 && (d->out.csplit % 16 || d->out.csplit <= 0 || d->out.csplit >= d->out.c)) return "output split must be a multiple of 16 channels";
  if (d->out.c % 4 && d->out.ptr2) return "two-part output needs whole 4-channel groups";
  if (d->out.dtype == VSSEG_BF16 && (d->out.pitch & 3) && d->out.c >= 4) return "output pitch";
  if (d->cout_mod > 0) return "z-folded launches are not supported";
  if (d->in_gate || d->res_tiles || d->in1 || d->res_mode == 5) return "marching-kernel-only features (in_gate / res_tiles / in1)";
  if (d->stats && (d->accumulate || d->res_mode != VSSEG_RES_NONE)) return "statistics combined with a residual";
  if (d->accumulate && d->res_mode != VSSEG_RES_NONE) return "accumulate combined with a residual";
  if (d->accumulate || d->res_mode != VSSEG_RES_NONE) {
    const vsseg_tensor& a = d->accumulate ? d->out : d->res;
    if ((a.pitch & 3) || (d->out.c & 3) || a.c < d->out.c || a.dtype != VSSEG_BF16 || (a.ptr2 && a.csplit % 16)) return "auxiliary tensor layout / dtype";
  }
  if (d->res_mode == VSSEG_RES_GATE && !d->gate) return "RES_GATE needs the gate map";
  if (d->class_split) {
    if (d->class_split < 2 || d->class_split > 8 || d->nsplit != d->class_split) return "class_split must be 2..8 and equal nsplit";
    if (d->nchunks != 1) return "class_split needs the whole input in one channel chunk";
    if (d->oo[0] || d->oo[1] || d->oo[2] || d->out.ptr2) return "class_split needs oo = 0 and a one-part output";
    if (d->nt * 16 < d->out.c) return "class_split: nt*16 < cout";
    for (int c = 0; c < d->class_split; ++c) {
      if (d->class_ntaps[c] < 1 || d->class_ntaps[c] > 8 || d->class_ntaps[c] > d->ntaps) return "class tap count";
      for (int t = 0; t < d->class_ntaps[c]; ++t)
        if (d->class_tap[c][t] < 0 || d->class_tap[c][t] >= d->ntaps) return "class tap index";
      for (int a = 0; a < 3; ++a)
        if (d->class_oo[c][a] < 0 || d->class_oo[c][a] >= d->os[a]) return "class offset outside the output stride";
    }
  } else if (d->nsplit < 1 || d->nsplit * d->nt * 16 < d->out.c) return "nsplit*nt*16 < cout";
  gm.cgs = d->ck / 8;
  if (d->ksteps * 4 < d->ntaps * gm.cgs && !d->class_split) return "ksteps too small";
  int64_t hv = 1;
  for (int a = 0; a < 3; ++a) {
    int lo = d->tap_off[0][a], hi = lo;
    for (int t = 1; t < d->ntaps; ++t) { lo = min(lo, d->tap_off[t][a]); hi = max(hi, d->tap_off[t][a]); }
    gm.omin[a] = lo;
    gm.halo[a] = (d->tile[a] - 1) * d->is[a] + (hi - lo + 1);
    if (gm.halo[a] > 255 || d->is[a] < 1 || d->os[a] < 1) return "halo extent > 255";
    hv *= gm.halo[a];
  }
  gm.vs = (gm.cgs | 1) * 16;  // odd number of 16-byte units per voxel: the 16 voxels x 4 K-groups of an operand read spread over the banks
  if (hv * gm.cgs >= (1 << 22)) return "halo too large";
  gm.rt = dc_red_tiles(d->mtw, d->nt);
  int off = ((d->ksteps * 4 * 4 * (d->class_split ? d->class_split : 1) + 15) / 16) * 16;
  gm.lds_epi = off; off += 3 * d->nt * 16 * 4;
  gm.lds_stat = off; off += DC_WAVES * 2 * d->nt * 16 * 4;
  gm.lds_halo = off;
  const int hb = (int)((hv * gm.vs + 1023) / 1024 * 1024), rb = DC_WAVES * gm.rt * 1024;  // whole 1 KiB DMA rows
  if (d->class_split) { gm.lds_red = off + hb; off += hb + rb; }  // every class reads the halo: the slabs have their own space
  else { gm.lds_red = off; off += hb > rb ? hb : rb; }              // the slabs reuse the halo's space once the K loops are done
  gm.lds_total = off;
  if (off > 160 * 1024) return "needs more than 160 KiB of LDS; reduce ck or the tile";
  return nullptr;
}

int vsseg_dconv_lds_bytes(const vsseg_igemm_desc* d) {
  DcGeom gm;
  const char* why = dc_check(d, gm);
  if (why) { vsseg_set_error("vsseg_igemm: depth -7 (deep-level kernel) not applicable: %s", why); return VSSEG_EINVAL; }
  return gm.lds_total;
}

template <int MT, int NT> static int dc_launch(const DconvK& k, dim3 grid, int lds, hipStream_t s) {
  static bool attr_set_dev[16] = {}; bool& attr_set = vsseg_dev_once(attr_set_dev);  // per device: the LDS opt-in is a per-device function attribute
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dconv_kernel<MT, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((dconv_kernel<MT, NT>), grid, dim3(DC_THREADS), lds, s, k);
  VSSEG_LAUNCH_CHECK("vsseg_igemm (deep-level kernel)");
  return VSSEG_OK;
}
template <int MT> static int dc_launch_nt(int nt, const DconvK& k, dim3 grid, int lds, hipStream_t s) {
  switch (nt) {
    case 1: return dc_launch<MT, 1>(k, grid, lds, s);
    case 2: return dc_launch<MT, 2>(k, grid, lds, s);
    case 3: return dc_launch<MT, 3>(k, grid, lds, s);
    case 4: return dc_launch<MT, 4>(k, grid, lds, s);
    case 5: return dc_launch<MT, 5>(k, grid, lds, s);
    default: return dc_launch<MT, 6>(k, grid, lds, s);
  }
}

int vsseg_dconv_launch(const vsseg_igemm_desc* d, const void* zeros, hipStream_t s) {
  DcGeom gm;
  const char* why = dc_check(d, gm);
  if (why) { vsseg_set_error("vsseg_igemm: depth -7 (deep-level kernel) not applicable: %s", why); return VSSEG_EINVAL; }
  DconvK k{};
  auto magic = [](int dv) { return dv <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)dv - 1) / (unsigned)dv); };
  k.in0 = reinterpret_cast<const char*>(d->in.ptr);
  k.in1 = d->in.ptr2 ? reinterpret_cast<const char*>(d->in.ptr2) - (int64_t)d->in.csplit * 2 : k.in0;
  k.in_csplit = d->in.ptr2 ? d->in.csplit : 0x7fffffff;
  k.in_vox_bytes = d->in.pitch * 2;
  const int oes = d->out.dtype == VSSEG_F32 ? 4 : 2;
  k.out0 = reinterpret_cast<char*>(d->out.ptr);
  k.out1 = d->out.ptr2 ? reinterpret_cast<char*>(d->out.ptr2) - (int64_t)d->out.csplit * oes : k.out0;
  k.out_csplit = d->out.ptr2 ? d->out.csplit : 0x7fffffff;
  k.out_vox_bytes = d->out.pitch * oes;
  k.out_f32 = d->out.dtype == VSSEG_F32;
  k.cout = d->out.c;
  k.aux_mode = 0;
  k.aux0 = k.aux1 = nullptr; k.aux_csplit = 0x7fffffff; k.aux_vox_bytes = 0;
  if (d->accumulate) k.aux_mode = 1;
  else if (d->res_mode == VSSEG_RES_ADD) k.aux_mode = 2;
  else if (d->res_mode == VSSEG_RES_RELUMASK) k.aux_mode = 3;
  else if (d->res_mode == VSSEG_RES_GATE) k.aux_mode = 4;
  if (k.aux_mode) {
    const vsseg_tensor& a = d->accumulate ? d->out : d->res;
    k.aux0 = reinterpret_cast<const char*>(a.ptr);
    k.aux1 = a.ptr2 ? reinterpret_cast<const char*>(a.ptr2) - (int64_t)a.csplit * 2 : k.aux0;
    k.aux_csplit = a.ptr2 ? a.csplit : 0x7fffffff;
    k.aux_vox_bytes = a.pitch * 2;
  }
  k.gate = d->gate;
  k.wpack = reinterpret_cast<const char*>(d->wpack);
  k.bias = d->bias; k.bias2 = d->bias2; k.scale = d->scale; k.shift = d->shift; k.alpha = d->alpha;
  k.stats = d->stats; k.stats_stride = d->stats_stride;
  k.fxflag = vsseg_fx_flag();
  VSSEG_CHECK(k.fxflag, "vsseg_igemm: could not allocate the flag word");
  k.prof = nullptr;
#ifdef VSSEG_DC_PROF
  static unsigned long long* prof = nullptr;
  if (!prof) (void)hipMalloc(&prof, 8 * 8);
  (void)hipMemsetAsync(prof, 0, 8 * 8, s);
  k.prof = prof;
  struct ProfPrint { unsigned long long* p; hipStream_t s; ~ProfPrint() {
    if (!getenv("VSSEG_DC_PROF_PRINT")) return;
    unsigned long long h[8];
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h, p, sizeof(h), hipMemcpyDeviceToHost);
    fprintf(stderr, "dconv prof (workgroup 8, wave 0; cycles): setup %llu | issue halo + preload + tables %llu | halo wait + barrier %llu | k-loop %llu | reduce+epilogue %llu\n", h[0], h[1], h[2], h[3], h[4]);
  } } pp{prof, s};
#endif
  k.zeros = zeros;
  k.act = d->act;
  k.X = d->in.x; k.Y = d->in.y; k.Z = d->in.z; k.OX = d->out.x; k.OY = d->out.y; k.OZ = d->out.z;
  int64_t tiles = d->in.n;
  for (int a = 0; a < 3; ++a) {
    k.q[a] = d->q[a]; k.is[a] = d->is[a]; k.os[a] = d->os[a]; k.oo[a] = d->oo[a];
    k.tl[a] = dc_log2(d->tile[a]);
    k.ntile[a] = (d->q[a] + d->tile[a] - 1) / d->tile[a];
    k.halo[a] = gm.halo[a]; k.omin[a] = gm.omin[a];
    tiles *= k.ntile[a];
  }
  VSSEG_CHECK(tiles > 0 && tiles < (1 << 24), "vsseg_igemm: bad tile count");
  k.mg_t2 = magic(k.ntile[2]); k.mg_t1 = magic(k.ntile[1]); k.mg_t0 = magic(k.ntile[0]);
  k.mg_h2 = magic(k.halo[2]); k.mg_h1 = magic(k.halo[1]); k.mg_cgs = magic(gm.cgs); k.mg_cgp = magic(gm.cgs | 1);
  k.ntaps = d->ntaps;
  for (int t = 0; t < VSSEG_MAX_TAPS; ++t)
    k.tapofs[t] = t < d->ntaps ? ((d->tap_off[t][0] - gm.omin[0]) * gm.halo[1] + (d->tap_off[t][1] - gm.omin[1])) * gm.halo[2] + (d->tap_off[t][2] - gm.omin[2]) : 0;
  k.ck = d->ck; k.cgs = gm.cgs; k.nchunks = d->nchunks; k.ksteps = d->ksteps; k.vs = gm.vs;
  k.nclass = d->class_split;
  for (int c = 0; c < 8; ++c) {
    k.class_ntaps[c] = c < d->class_split ? d->class_ntaps[c] : 0;
    for (int t = 0; t < 8; ++t) k.class_tap[c][t] = c < d->class_split ? d->class_tap[c][t] : 0;
    for (int a = 0; a < 3; ++a) k.class_oo[c][a] = c < d->class_split ? d->class_oo[c][a] : 0;
  }
  for (int w = 0; w < DC_WAVES; ++w) k.wave_ncls[w] = 0;
  {  // classes -> waves, longest first onto the least loaded wave (3x3x3 stride 2: taps 8,4,4,4,2,2,2,1 -> loads 9,6,6,6)
    int load[DC_WAVES] = {0, 0, 0, 0};
    bool used[8] = {false, false, false, false, false, false, false, false};
    for (int n = 0; n < d->class_split; ++n) {
      int best = -1;
      for (int c = 0; c < d->class_split; ++c)
        if (!used[c] && (best < 0 || d->class_ntaps[c] > d->class_ntaps[best])) best = c;
      used[best] = true;
      int w = 0;
      for (int u = 1; u < DC_WAVES; ++u)
        if (load[u] < load[w]) w = u;
      k.wave_cls[w][k.wave_ncls[w]++] = best;
      load[w] += d->class_ntaps[best];
    }
  }
  k.lds_epi = gm.lds_epi; k.lds_stat = gm.lds_stat; k.lds_halo = gm.lds_halo; k.lds_red = gm.lds_red;
  dim3 grid((unsigned)tiles, (unsigned)(d->class_split ? 1 : d->nsplit));
  switch (d->mtw) {
    case 2: return dc_launch_nt<2>(d->nt, k, grid, gm.lds_total, s);
    case 4: return dc_launch_nt<4>(d->nt, k, grid, gm.lds_total, s);
    default: return dc_launch_nt<8>(d->nt, k, grid, gm.lds_total, s);
  }
}
