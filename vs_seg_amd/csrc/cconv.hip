// Compute-bound convolution: the MFMA-bound launches of vsseg_igemm — stride-1 3x3x3 bf16 convolutions and data gradients with 32..160 input
// channels and 32 / 48 output channels per workgroup (1, 2 or 4 workgroups per voxel tile) on the 96x32x128 and 48x16x64 levels of the 2.5D U-Net
// (ref:params/networks/blocks/convolutions.py:114-146; SURVEY §8a rows 9, 10, 13, 14, 31, 34, 36, 39 and their data gradients: 54 % of the
// network's MACs) — as a kernel whose geometry is a compile-time constant (launch plans with depth = -3).
//
// What held the general kernel (igemm_kernel.h, NT >= 3: producer / consumer waves) at 0.20-0.32 of the bf16 MFMA peak (profiles/r02_pmc_sq.txt):
// one consumer wave per SIMD, so every ds_read -> MFMA dependency it could not software-pipeline left the matrix pipe idle (waves parked 54-58 %
// of their cycles), ~4 VALU + ~3 SALU per MFMA of run-time addressing, and 26-34 % of the LDS cycles lost to bank conflicts.  Here:
//
//   * tile 4x8x16 = 512 output voxels per workgroup, 8 waves x 4 M-tiles; an M-tile is 16 z-CONSECUTIVE voxels, so with 16-channel chunks
//     (32 bytes per halo voxel) the 16 lanes x 4 K-groups of an MFMA operand read are 16 different 16-byte bank groups: conflict-free with
//     no swizzle.  Two waves per SIMD: one wave's fragment reads hide behind the other's MFMAs
//   * K = (tap, 8-channel group) in chunks of 16 input channels: 14 fully unrolled K-steps per chunk (27 taps x 2 groups, one padded half
//     step), every LDS offset an instruction immediate; per K-step and wave 4 + NT ds_read_b128 feed 4 x NT MFMAs (LDS pipe <= 60 % busy)
//   * both operands of a chunk — the halo (6x10x18 voxels x 16 channels = 34 KiB) and the packed weights (14 x NT KiB) — arrive by LDS-DMA
//     into one of two LDS buffers while the other is multiplied: one counted `s_waitcnt vmcnt` + one s_barrier per stage, nothing inside the K
//     loop waits on memory.  The DMAs are issued from inline assembly (common.h) so hipcc keeps its `vmcnt(0)` out of the K loop — ONE per
//     K-step, between the MFMAs of the step (round 6): an LDS-DMA instruction holds its wave ~85 cycles of a per-CU serial resource, and the
//     9-11 of them a wave issues per stage, back to back at the top of the stage as rounds 2-5 had it, stopped both waves of every SIMD at once
//     (96 -> 48 at 96x32x128 x 4: 380 us that way, 322 us this way, 283 us with the fetch removed; the register path global_load -> ds_write_b128
//     measured 354 us: DESIGN.md 3.15)
//   * same packed weights, K order, fp32 accumulation order and epilogue semantics as the general kernel: results agree bit for bit
//     (tests/test_gpu_ops.py::test_compute_kernel_equals_general_kernel); XCD-contiguous persistent tile walk, one workgroup per CU
#include "common.h"
#include "cconv.h"
#include <type_traits>

constexpr int CC_TX = 4, CC_TY = 8, CC_TZ = 16;
constexpr int CC_HX = CC_TX + 2, CC_HY = CC_TY + 2, CC_HZ = CC_TZ + 2;
constexpr int CC_HVOX = CC_HX * CC_HY * CC_HZ;            // 1080
constexpr int CC_VB = 32;                                 // bytes per halo voxel: one 16-channel chunk
constexpr int CC_PIECES = CC_HVOX * 2;                    // 16-byte pieces of a halo chunk
constexpr int CC_HROWS = (CC_PIECES + 63) / 64;           // 1 KiB DMA rows (34; the last one is 3/4 full)
constexpr int CC_HBYTES = CC_HROWS * 1024;
constexpr int CC_KS = 14;                                 // K-steps per chunk: 54 K-groups of 8 channels -> 13.5 steps of 4
constexpr int CC_NH = (CC_HROWS + 7) / 8;                 // halo DMA instructions per thread and stage (rows u*8 + wave)
constexpr int cc_buf_bytes(int nt) { return CC_HBYTES + CC_KS * nt * 1024; }
constexpr int cc_lds_bytes(int nt) { return 2 * cc_buf_bytes(nt) + 5 * nt * 16 * 4 + 16; }

struct CconvK {
  const char* in0;   // channels [0, csplit) ...
  const char* in1;   // ... and [csplit, c), biased by -csplit channels (== in0 for an ordinary tensor)
  char* out0; char* out1;
  const char* aux0; const char* aux1;
  const float* gate;
  const char* wpack;
  const float *bias, *bias2, *scale, *shift, *alpha;
  double* stats;
  unsigned* fxflag;  // sticky range / non-finite flag of the fixed-point statistics (common.h)
  const void* zeros;
  int in_csplit_ch;  // first 16-channel chunk that lives in part 1 (>= nch for an ordinary tensor)
  int in_vox_bytes, out_vox_bytes, aux_vox_bytes;
  int out_csplit, aux_csplit;  // channels (0x7fffffff: ordinary tensor)
  int out_f32;
  int aux_mode;  // 0 none, 1 accumulate (aux = out), 2 residual add, 3 ReLU mask, 4 gated add
  int act, cout, stats_stride;
  int nch;       // 16-channel input chunks
  int X, Y, Z, ntx, nty, ntz;
  unsigned mg_tz, mg_ty, mg_tx;
  int tiles, per_xcd;
};

__device__ __forceinline__ unsigned cc_div(unsigned n, unsigned magic) { return magic ? __umulhi(n, magic) : n; }  // magic 0: divisor 1
__device__ __forceinline__ void cc_wait_vm(int n) {  // s_waitcnt vmcnt(n) for the counts this kernel uses (n is wave-uniform)
  if (n >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// MODE: 0 plain, 1 + BatchNorm statistics, 2 + auxiliary operand (bf16)
template <int NT, int MODE>
__global__ __launch_bounds__(512, 2) void cconv_kernel(const CconvK k) {
  constexpr bool STATS = MODE == 1, AUXM = MODE == 2;
  constexpr int MT = 4;
  constexpr int BUF = cc_buf_bytes(NT);
  constexpr int WROWS = CC_KS * NT, NW = (WROWS + 7) / 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* epi = reinterpret_cast<float*>(smem + 2 * BUF);  // bias | scale | shift, NT*16 each (reused by the statistics reduction)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, l15 = lane & 15;
  const int X = k.X, Y = k.Y, Z = k.Z, cout = k.cout, nch = k.nch;
  const int split = blockIdx.y, c_base = split * NT * 16;

  for (int i = tid; i < NT * 16; i += 512) {
    const int c = c_base + i;
    const bool ok = c < cout;
    epi[i] = ((ok && k.bias) ? k.bias[c] : 0.f) + ((ok && k.bias2) ? k.bias2[c] : 0.f);
    epi[NT * 16 + i] = (ok && k.scale) ? k.scale[c] : 1.f;
    epi[2 * NT * 16 + i] = (ok && k.scale) ? k.shift[c] : 0.f;
  }
  const float alpha = (k.act == VSSEG_ACT_PRELU && k.alpha) ? *k.alpha : 0.f;

  // ---- this thread's halo pieces: LDS slot j = (u*8 + wave)*64 + lane holds the 16-byte group j & 1 of halo voxel j >> 1 ----
  unsigned rel[CC_NH], hxyz[CC_NH];
#pragma unroll
  for (int u = 0; u < CC_NH; ++u) {
    const int j = (u * 8 + wave) * 64 + lane;
    const int hv = j >> 1, c16 = j & 1, hz = hv % CC_HZ, r = hv / CC_HZ, hy = r % CC_HY, hx = r / CC_HY;
    const bool ok = j < CC_PIECES;
    rel[u] = ok ? (unsigned)((hx * Y + hy) * Z + hz) * (unsigned)k.in_vox_bytes + (unsigned)c16 * 16u : 0u;
    hxyz[u] = ok ? (unsigned)(hx | (hy << 8) | (hz << 16)) : 0xffffffffu;  // padding lanes of the last row fetch from the zero page
  }
  // ---- MFMA operand addressing.  K-group p = ks*4 + g -> tap p >> 1, channel group p & 1; lane column l15 -> voxel z = l15 of the M-tile ----
  int koff[CC_KS];
#pragma unroll
  for (int ks = 0; ks < CC_KS; ++ks) {
    const int p = ks * 4 + g, tap = p >> 1;
    const int dx = tap / 9, dy = (tap / 3) % 3, dz = tap % 3;
    koff[ks] = tap < 27 ? ((dx * CC_HY + dy) * CC_HZ + dz) * CC_VB : 0;  // padded K-groups: zero weights times valid data
  }
  // M-tile m of wave w: x = w >> 1, y = (w & 1) * 4 + m  (m advances one halo row of y: + CC_HZ * CC_VB bytes, an immediate)
  const int vb0 = ((((wave >> 1) * CC_HY + (wave & 1) * 4) * CC_HZ + l15) * CC_VB) + (g & 1) * 16;
  const unsigned ov0 = (unsigned)(((wave >> 1) * Y + (wave & 1) * 4) * Z + l15);
  const unsigned out_es = k.out_f32 ? 4u : 2u;
  const bool simple = !k.out_f32 && !k.scale && (k.act == VSSEG_ACT_NONE || k.act == VSSEG_ACT_PRELU);
  const int ekind = !simple ? 2 : (k.aux_mode == 3 ? 1 : 0);
  const float alpha_eff = k.act == VSSEG_ACT_PRELU ? alpha : 1.f;
  float ssum[STATS ? NT : 1][4], ssq[STATS ? NT : 1][4];  // sum / sum of squares of the output
#pragma unroll
  for (int t = 0; t < (STATS ? NT : 1); ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) { ssum[t][r] = 0.f; ssq[t][r] = 0.f; }

  // ---- tile schedule: XCD x (= blockIdx.x % 8) owns tiles [x*per_xcd, (x+1)*per_xcd); its workgroups stride through them ----
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, wgs = gridDim.x >> 3;
  int cnt = k.tiles - xcd * k.per_xcd;
  cnt = cnt > k.per_xcd ? k.per_xcd : cnt;
  const int my_tiles = cnt > slot ? (cnt - 1 - slot) / wgs + 1 : 0;
  const int nstages = my_tiles * nch;
  struct Tile { int n, x0, y0, z0; };
  auto tile_of = [&](int i) {  // i-th tile of this workgroup
    unsigned b = (unsigned)(xcd * k.per_xcd + slot + i * wgs);
    Tile t;
    unsigned q = cc_div(b, k.mg_tz); t.z0 = (int)(b - q * k.ntz) * CC_TZ; b = q;
    q = cc_div(b, k.mg_ty); t.y0 = (int)(b - q * k.nty) * CC_TY; b = q;
    q = cc_div(b, k.mg_tx); t.x0 = (int)(b - q * k.ntx) * CC_TX; t.n = (int)q;
    return t;
  };
  // ---- LDS-DMA of one stage = (tile, 16-channel chunk): halo chunk + packed weights of the chunk into buffer `buf` ----
  constexpr int NPIECES = CC_NH + NW;
  static_assert(NPIECES <= CC_KS, "one DMA piece per K-step");
  // piece p of a stage: p < CC_NH -> halo row p*8 + wave, else packed-weight row (p - CC_NH)*8 + wave (1 KiB each: 16 bytes per lane); rows past the end: no piece (wave-uniform)
  auto piece_row = [&](int p) __attribute__((always_inline)) { return (p < CC_NH ? p : p - CC_NH) * 8 + wave; };
  auto piece_on = [&](int p) __attribute__((always_inline)) { return p < CC_NH ? piece_row(p) < CC_HROWS : (p < NPIECES && piece_row(p) < WROWS); };
  auto piece_src = [&](const Tile& t, int ch, int p) __attribute__((always_inline)) -> const void* {
    if (p < CC_NH) {
      const int u = p;
      const int64_t ivox = (((int64_t)t.n * X + (t.x0 - 1)) * Y + (t.y0 - 1)) * Z + (t.z0 - 1);
      const char* org = (ch >= k.in_csplit_ch ? k.in1 : k.in0) + ivox * k.in_vox_bytes + ch * CC_VB;
      const int ix = t.x0 - 1, iy = t.y0 - 1, iz = t.z0 - 1;
      // branch-free bounds test (boundary tiles are 2/3 of a 96x32x128 level: the zero padding of the convolution comes from the zero page)
      const unsigned h = hxyz[u];
      const int gx = ix + (int)(h & 255u), gy = iy + (int)((h >> 8) & 255u), gz = iz + (int)((h >> 16) & 255u);
      const bool ok = (h != 0xffffffffu) & ((unsigned)gx < (unsigned)X) & ((unsigned)gy < (unsigned)Y) & ((unsigned)gz < (unsigned)Z);
      return ok ? (const void*)(org + rel[u]) : k.zeros;
    }
    return k.wpack + ((int64_t)(split * nch + ch) * WROWS + piece_row(p)) * 1024 + lane * 16;
  };
  auto piece_dst = [&](int buf, int p) __attribute__((always_inline)) { return smem + buf * BUF + (p < CC_NH ? 0 : CC_HBYTES) + piece_row(p) * 1024; };  // (wave base: + lane*16 for a register store)
  auto issue_piece = [&](const Tile& t, int ch, int buf, int p) __attribute__((always_inline)) {
    if (piece_on(p)) vsseg_dma16(piece_src(t, ch, p), piece_dst(buf, p));
  };
  auto issue = [&](const Tile& t, int ch, int buf) {
#pragma unroll
    for (int p = 0; p < NPIECES; ++p) issue_piece(t, ch, buf, p);
  };

  __syncthreads();  // epilogue constants visible
  Tile t_issue{0, 0, 0, 0}, t_cur{0, 0, 0, 0};
  int ti_issue = 0, ch_issue = 0, ti_cur = 0, ch_cur = 0;
  if (nstages > 0) {
    t_issue = tile_of(0);
    issue(t_issue, 0, 0);
    if (++ch_issue == nch) { ch_issue = 0; ++ti_issue; }
  }
  f32x4 acc[MT][NT];
  int st_prev = 0;  // store instructions issued after the DMAs of the stage about to be consumed (the previous stage's epilogue)
  for (int s = 0; s < nstages; ++s) {
    // stage s has landed once only the younger stores remain in flight (VMEM operations of a wave retire in issue order)
    cc_wait_vm(st_prev);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave's pieces of stage s are in LDS; every wave finished reading the other buffer (stage s-1)
    st_prev = 0;
    if (ch_cur == 0) {
      t_cur = tile_of(ti_cur);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const bool last = ch_cur + 1 == nch;
    const int64_t ovox = (((int64_t)t_cur.n * X + t_cur.x0) * Y + t_cur.y0) * Z + t_cur.z0;
    // auxiliary operands of the epilogue: ordinary loads issued in front of the tile's last K loop — and in front of the next stage's DMAs:
    // hipcc guards the reuse of their registers with a counted wait that knows nothing of the inline-assembly DMAs, so behind them it
    // waited for the first DMA of the stage to land before it even issued these loads
    uint2 auxv[AUXM ? MT : 1][AUXM ? NT : 1];
    float gatev[AUXM ? MT : 1];
    if constexpr (AUXM) {
      if (last) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int64_t vox = ovox + ov0 + (unsigned)(m * Z);
          if (k.aux_mode == 4) gatev[m] = k.gate[vox];
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int c = c_base + t * 16 + g * 4;
            auxv[m][t] = *reinterpret_cast<const uint2*>((c_base + t * 16 >= k.aux_csplit ? k.aux1 : k.aux0) + vox * k.aux_vox_bytes + c * 2);
          }
        }
      }
    }
    const bool more = s + 1 < nstages;
    if (more) {
      if (ch_issue == 0) t_issue = tile_of(ti_issue);
    }
    {
      // K loop: the fragments of step ks+1 are read before the MFMAs of step ks are issued (two register sets), so that a wave's LDS latency
      // sits behind its own 4 x NT MFMAs as well as behind the SIMD's other wave
      const char* Hs = smem + (s & 1) * BUF + vb0;
      const char* Ws = smem + (s & 1) * BUF + CC_HBYTES + lane * 16;
      bf16x8 wc[NT], ac[MT], wn[NT], an[MT];
#pragma unroll
      for (int t = 0; t < NT; ++t) wc[t] = *reinterpret_cast<const bf16x8*>(Ws + t * 1024);
#pragma unroll
      for (int m = 0; m < MT; ++m) ac[m] = *reinterpret_cast<const bf16x8*>(Hs + koff[0] + m * (CC_HZ * CC_VB));
#pragma unroll
      for (int ks = 0; ks < CC_KS; ++ks) {
        if (ks + 1 < CC_KS) {
#pragma unroll
          for (int t = 0; t < NT; ++t) wn[t] = *reinterpret_cast<const bf16x8*>(Ws + ((ks + 1) * NT + t) * 1024);
          const char* Hk = Hs + koff[ks + 1 < CC_KS ? ks + 1 : ks];
#pragma unroll
          for (int m = 0; m < MT; ++m) an[m] = *reinterpret_cast<const bf16x8*>(Hk + m * (CC_HZ * CC_VB));
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the reads of step ks+1 in front of the MFMAs of step ks (+2-6 % over hipcc's own interleaving)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[t], ac[m], acc[m][t], 0, 0, 0);
            // piece ks of the NEXT stage, in the middle of this K-step's MFMAs (header: the SIMD's other wave multiplies while this one sits in the DMA instruction)
            if (m == 1 && t == NT - 1 && ks < NPIECES && more) issue_piece(t_issue, ch_issue, (s + 1) & 1, ks);
          }
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 1 < CC_KS) {
#pragma unroll
          for (int t = 0; t < NT; ++t) wc[t] = wn[t];
#pragma unroll
          for (int m = 0; m < MT; ++m) ac[m] = an[m];
        }
      }
    }
    if (more && ++ch_issue == nch) { ch_issue = 0; ++ti_issue; }
    if (!last) { ++ch_cur; continue; }
    ch_cur = 0;
    ++ti_cur;
    st_prev = MT * NT;

    // ---- epilogue: bias (+ statistics) (+ eval affine) + activation (+ auxiliary operand), 4 channels per lane; one 8-byte (bf16) store per
    //      (M-tile, 16-channel block): MT*NT store instructions per wave, which the next stage's counted wait leaves in flight.  The epilogue
    //      kind is wave-uniform and dispatched once per tile (kinds as in sconv.hip).
    auto epilogue = [&](auto kind_c) {
      constexpr int KIND = decltype(kind_c)::value;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int64_t vox = ovox + ov0 + (unsigned)(m * Z);
        float gt = 1.f;
        if constexpr (AUXM) gt = k.aux_mode == 4 ? 1.f + gatev[m] : 1.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int cl = t * 16 + g * 4, c = c_base + cl;
          const float4 bi = *reinterpret_cast<const float4*>(epi + cl);
          float val[4] = {acc[m][t][0] + bi.x, acc[m][t][1] + bi.y, acc[m][t][2] + bi.z, acc[m][t][3] + bi.w};
          if constexpr (STATS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { ssum[t][r] += val[r]; ssq[t][r] += val[r] * val[r]; }
          }
          if constexpr (KIND == 2) {
            if (k.scale) {
              const float4 sc = *reinterpret_cast<const float4*>(epi + NT * 16 + cl), sh = *reinterpret_cast<const float4*>(epi + 2 * NT * 16 + cl);
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
          if constexpr (AUXM) {
            const uint2 a = auxv[m][t];
            const float4 av = make_float4(__uint_as_float(a.x << 16), __uint_as_float(a.x & 0xffff0000u), __uint_as_float(a.y << 16), __uint_as_float(a.y & 0xffff0000u));
            if (KIND == 1 || (KIND == 2 && k.aux_mode == 3)) {
              val[0] = av.x > 0.f ? val[0] : 0.f; val[1] = av.y > 0.f ? val[1] : 0.f; val[2] = av.z > 0.f ? val[2] : 0.f; val[3] = av.w > 0.f ? val[3] : 0.f;
            } else if (KIND == 2 && k.aux_mode != 4) {
              val[0] += av.x; val[1] += av.y; val[2] += av.z; val[3] += av.w;
            } else {
              val[0] = vsseg_fma_unpacked(av.x, gt, val[0]); val[1] = vsseg_fma_unpacked(av.y, gt, val[1]);  // (not v_pk_fma_f32 op_sel: common.h)
              val[2] = vsseg_fma_unpacked(av.z, gt, val[2]); val[3] = vsseg_fma_unpacked(av.w, gt, val[3]);
            }
          }
          char* op = (c_base + t * 16 >= k.out_csplit ? k.out1 : k.out0) + vox * k.out_vox_bytes + c * (int)out_es;
          if (KIND != 2 || !k.out_f32) st4(reinterpret_cast<bf16_t*>(op), make_float4(val[0], val[1], val[2], val[3]));
          else st4(reinterpret_cast<float*>(op), make_float4(val[0], val[1], val[2], val[3]));
        }
      }
    };
    if (ekind == 0) epilogue(std::integral_constant<int, 0>{});
    else if (AUXM && ekind == 1) epilogue(std::integral_constant<int, AUXM ? 1 : 0>{});
    else epilogue(std::integral_constant<int, 2>{});
  }

  if constexpr (STATS) {  // per-channel sum / sum of squares of this workgroup's tiles: shuffle tree -> one LDS row per wave, summed in wave order -> the layer's
                          // sharded statistics as fixed-point integer atomics (order-independent: vsseg_fx_add; layout of vsseg_igemm_desc.stats)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);  // [8 waves][2][NT*16]: the staging buffers are no longer needed
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
    double* st = k.stats + (int64_t)((blockIdx.x + blockIdx.y * gridDim.x) % VSSEG_STAT_SHARDS) * 2 * k.stats_stride;
    for (int i = tid; i < 2 * NT * 16; i += 512) {
      const int which = i / (NT * 16), c = c_base + i - which * NT * 16;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v += red[w * (2 * NT * 16) + i];
      if (c < cout) vsseg_fx_add(&st[which * k.stats_stride + c], (double)v, VSSEG_FX_STAT, k.fxflag);
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------
template <int NT, int MODE> static int cc_launch_mode(const CconvK& k, int nsplit, hipStream_t s) {
  static bool attr_set_dev[16] = {}; bool& attr_set = vsseg_dev_once(attr_set_dev);  // per device: the LDS opt-in is a per-device function attribute
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&cconv_kernel<NT, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  int gx = (256 / nsplit) & ~7;  // one workgroup per CU, whole rounds of the 8 XCDs
  const int need = (k.tiles + 7) / 8 * 8;
  if (gx > need) gx = need;
  hipLaunchKernelGGL((cconv_kernel<NT, MODE>), dim3((unsigned)gx, (unsigned)nsplit), dim3(512), cc_lds_bytes(NT), s, k);
  VSSEG_LAUNCH_CHECK("vsseg_igemm (compute kernel)");
  return VSSEG_OK;
}
template <int NT> static int cc_launch(const CconvK& k, int nsplit, hipStream_t s) {
  if (k.stats) return cc_launch_mode<NT, 1>(k, nsplit, s);
  if (k.aux_mode) return cc_launch_mode<NT, 2>(k, nsplit, s);
  return cc_launch_mode<NT, 0>(k, nsplit, s);
}

static const char* cc_check(const vsseg_igemm_desc* d) {
  if (d->in.dtype != VSSEG_BF16) return "input is not bf16";
  if (d->nt != 2 && d->nt != 3) return "nt must be 2 or 3";
  if (d->ck != 16 || d->ksteps != CC_KS || d->mtw != 8) return "needs ck = 16, ksteps = 14, mtw = 8";
  if (d->nchunks < 1 || d->in.c != d->nchunks * 16) return "input channels must be nchunks x 16";
  if (d->tile[0] != CC_TX || d->tile[1] != CC_TY || d->tile[2] != CC_TZ) return "tile must be 4x8x16";
  for (int a = 0; a < 3; ++a)
    if (d->is[a] != 1 || d->os[a] != 1 || d->oo[a] != 0) return "stride-1 lattices only";
  if (d->q[0] != d->in.x || d->q[1] != d->in.y || d->q[2] != d->in.z || d->q[0] != d->out.x || d->q[1] != d->out.y || d->q[2] != d->out.z) return "lattice, input and output extents differ";
  if (d->q[0] % CC_TX || d->q[1] % CC_TY || d->q[2] % CC_TZ) return "extent is not a multiple of the tile";
  if (d->ntaps != 27) return "3x3x3 taps only";
  for (int t = 0; t < 27; ++t)
    if (d->tap_off[t][0] != t / 9 - 1 || d->tap_off[t][1] != (t / 3) % 3 - 1 || d->tap_off[t][2] != t % 3 - 1) return "taps are not the 3x3x3 stencil in (x, y, z) order";
  if (d->in.pitch % 8 || ((uintptr_t)d->in.ptr & 15) || ((uintptr_t)d->in.ptr2 & 15)) return "input must be 16-byte aligned voxel rows";
  if (d->in.ptr2 && d->in.csplit % 16) return "input split must be a multiple of 16 channels";
  if (d->nsplit < 1 || d->nsplit > 4 || d->nsplit == 3 || d->out.c != d->nsplit * d->nt * 16) return "output channels must be nsplit (1, 2, 4) x nt x 16";
  if (d->out.dtype != VSSEG_BF16 && d->out.dtype != VSSEG_F32) return "output dtype";
  if (d->out.pitch & 3) return "output pitch";
  if (d->out.ptr2 && d->out.csplit % 16) return "output split must be a multiple of 16 channels";
  if (d->cout_mod > 0) return "z-folded launches are not supported";
  if (d->stats && (d->accumulate || d->res_mode != VSSEG_RES_NONE)) return "statistics combined with a residual";
  if (d->accumulate && d->res_mode != VSSEG_RES_NONE) return "accumulate combined with a residual";
  if (d->accumulate || d->res_mode != VSSEG_RES_NONE) {
    const vsseg_tensor& a = d->accumulate ? d->out : d->res;
    if ((a.pitch & 3) || a.c < d->out.c || a.dtype != VSSEG_BF16 || (a.ptr2 && a.csplit % 16)) return "auxiliary tensor layout / dtype";
  }
  if (d->res_mode == VSSEG_RES_GATE && !d->gate) return "RES_GATE needs the gate map";
  if ((int64_t)d->in.n * d->q[0] * d->q[1] * d->q[2] / 512 >= (1ll << 24)) return "too many tiles";
  return nullptr;
}

int vsseg_cconv_lds_bytes(const vsseg_igemm_desc* d) {
  const char* why = cc_check(d);
  if (why) { vsseg_set_error("vsseg_igemm: depth -3 (compute kernel) not applicable: %s", why); return VSSEG_EINVAL; }
  return d->nt == 3 ? cc_lds_bytes(3) : cc_lds_bytes(2);
}

int vsseg_cconv_launch(const vsseg_igemm_desc* d, const void* zeros, hipStream_t s) {
  const char* why = cc_check(d);
  if (why) { vsseg_set_error("vsseg_igemm: depth -3 (compute kernel) not applicable: %s", why); return VSSEG_EINVAL; }
  CconvK k;
  auto magic = [](int dv) { return dv <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)dv - 1) / (unsigned)dv); };
  k.in0 = reinterpret_cast<const char*>(d->in.ptr);
  k.in1 = d->in.ptr2 ? reinterpret_cast<const char*>(d->in.ptr2) - (int64_t)d->in.csplit * 2 : k.in0;
  k.in_csplit_ch = d->in.ptr2 ? d->in.csplit / 16 : 1 << 20;
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
  VSSEG_FX_FLAG(fxflag_, "vsseg_igemm (compute kernel)");
  k.fxflag = fxflag_;
  k.zeros = zeros;
  k.act = d->act; k.cout = d->out.c;
  k.nch = d->nchunks;
  k.X = d->q[0]; k.Y = d->q[1]; k.Z = d->q[2];
  k.ntx = k.X / CC_TX; k.nty = k.Y / CC_TY; k.ntz = k.Z / CC_TZ;
  k.mg_tx = magic(k.ntx); k.mg_ty = magic(k.nty); k.mg_tz = magic(k.ntz);
  k.tiles = d->in.n * k.ntx * k.nty * k.ntz;
  k.per_xcd = (k.tiles + 7) / 8;
  return d->nt == 3 ? cc_launch<3>(k, d->nsplit, s) : cc_launch<2>(k, d->nsplit, s);
}
