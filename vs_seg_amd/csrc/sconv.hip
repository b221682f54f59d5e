// Streaming convolution: the HBM-bound launches of vsseg_igemm — stride-1 3x3x1 / 1x1x1 bf16 convolutions and data gradients with at most 64
// input and output channels on the two finest levels of the 2.5D U-Net (ref:params/networks/blocks/convolutions.py:114-146; the 16/32-channel
// layers at 384x128x128 and 192x64x128; also the 1x1x1 residual convolutions of the level-2 units, 32 -> 48 and 96 -> 48) — as a kernel whose geometry is a compile-time constant.
//
// vsseg_igemm's general kernel (igemm_kernel.h) serves every lattice class, tile shape, channel chunking and prefetch depth from run-time
// tables; on these layers it issued ~1150 instructions per tile and wave around 20-72 MFMAs and ran at 2.1-3.5 TB/s with 55-65 % of its
// cycles in s_waitcnt (profiles/r02_pmc_sq.txt).  Here the tile (8x8x4 voxels), the taps, the channel counts and therefore every LDS offset
// of the K loop are template parameters: ~200 instructions per tile and wave, no tables, no tile descriptors, no spills.
//
//   * tile 8x8x4 output voxels per workgroup (4 waves x 4 M-tiles of 16 voxels), halo (8+2)x(8+2)x4 fetched by LDS-DMA (16-byte pieces,
//     whole voxel rows: every global request is a full 32..128-byte run), one LDS buffer: the latency of a tile's fetch is hidden by the
//     other 2-3 workgroups resident on the CU, not by a software pipeline (measured optimum, tools/probes/lean_conv_probe.hip)
//   * the halo is stored voxel-major ([halo voxel][CIN]); the 16-byte piece slots inside a voxel are XOR-swizzled with (hy, hz) bits so that
//     the 16 lanes of an MFMA operand read (16 z/y-consecutive voxels, one 8-channel group) hit 16 different 16-byte bank groups.  The
//     swizzle costs nothing: the DMA writes LDS in lane order, so it is applied to the GLOBAL address each lane fetches
//   * same packed weights, K order (tap, 8-channel group), MFMA operand order, output-channel ownership and epilogue semantics as the general
//     kernel (bias, BatchNorm statistics in sharded fp64 atomics, eval affine, activation, accumulate / residual add / ReLU mask / gated add) —
//     a launch plan only has to say depth = -2; results agree with the general kernel bit for bit (same K order, same fp32 accumulation)
//   * XCD-aware persistent walk: workgroup b works for XCD b % 8, which owns a contiguous 1/8 of the (n, x, y, z)-ordered tile list
#include "common.h"
#include "sconv.h"
#include <type_traits>

typedef __attribute__((address_space(1))) const void sc_gvoid_t;
typedef __attribute__((address_space(3))) void sc_lvoid_t;
__device__ __forceinline__ void sc_dma16(const void* gsrc, char* lds_wave_base) { __builtin_amdgcn_global_load_lds((sc_gvoid_t*)gsrc, (sc_lvoid_t*)lds_wave_base, 16, 0, 0); }

constexpr int SC_TY = 8, SC_TZ = 4;
// M-tiles (16 voxels) per wave: 4 = tile 8x8x4; 2 = tile 4x8x4 for the 64-channel 3x3x1 inputs, whose 8x8x4 halo + weights would leave one workgroup per CU
constexpr int sc_mt(int cin, int taps) { return (cin >= 64 && taps == 9) ? 2 : 4; }

struct SconvK {
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
  int in_csplit_pc;  // first 16-byte piece of a voxel row that lives in part 1 (>= pieces per voxel for an ordinary tensor)
  int in_vox_bytes, out_vox_bytes, aux_vox_bytes;
  int out_csplit, aux_csplit;  // channels (0x7fffffff: ordinary tensor)
  int out_f32;
  int aux_mode;  // 0 none, 1 accumulate (aux = out), 2 residual add, 3 ReLU mask, 4 gated add
  int act, cout, cout_mod, stats_stride;
  int X, Y, Z, ntx, nty, ntz;
  unsigned mg_tz, mg_ty, mg_tx;  // ceil(2^32 / d): exact quotients by one s_mul_hi_u32
  int tiles, per_xcd, walk;
  int ps_cls0, ps_tpc_shift;  // fused output-parity classes (taps 4): first class of the launch, log2(channel tiles per class)
};

template <int G> __device__ __forceinline__ int sc_swz(int hy, int hz) {
  if constexpr (G == 2) return (hy >> 1) & 1;
  else if constexpr (G == 4) return hy & 3;
  else if constexpr (G == 8) return ((hy & 3) << 1) | ((hz >> 1) & 1);
  else return 0;
}
__device__ __forceinline__ unsigned sc_div(unsigned n, unsigned magic, unsigned d) { return magic ? __umulhi(n, magic) : n; }  // magic 0: d == 1

// MODE: 0 plain, 1 + BatchNorm statistics, 2 + auxiliary operand (bf16) — registers are spent only on what a launch uses
template <int CIN, int NT, int TAPS, int MODE>
__global__ __launch_bounds__(256, NT >= 4 ? 2 : 3) void sconv_kernel(const SconvK k) {
  constexpr bool STATS = MODE == 1, AUXM = MODE == 2;
  constexpr int G = CIN / 8, CINB = CIN * 2;
  // TAPS 9: 3x3x1 stencil (halo 1 on both sides); 1: 1x1x1; 4: the 2x2x1 neighbourhood (+0 / +1) of the FUSED output-parity classes of a
  // stride-2 transposed convolution / stride-2 data gradient ("pixel shuffle", PS): the launch runs on the coarse lattice, output channel tile
  // t is parity class (px, py) = (t >> 1, t & 1) and is stored at fine voxel (2x + px, 2y + py, z) — ONE read of the input instead of one per class
  constexpr int R = TAPS == 9 ? 1 : 0, RH = (TAPS == 9 || TAPS == 4) ? 1 : 0;
  constexpr bool PS = TAPS == 4;
  const int tpc_sh = PS ? k.ps_tpc_shift : 0;  // PS: log2 of the 16-channel tiles per parity class (1 tile: all 4 classes in this launch; 2 tiles: classes (px, 0) and (px, 1))
  constexpr int MT = sc_mt(CIN, TAPS), SC_TX = 2 * MT;
  constexpr int HX = SC_TX + R + RH, HY = SC_TY + R + RH, HZ = SC_TZ;
  constexpr int PIECES = HX * HY * HZ * G, NINST = (PIECES + 255) / 256;
  constexpr int KSTEPS = (TAPS * G + 3) / 4;
  constexpr int W_BYTES = KSTEPS * NT * 1024, H_BYTES = NINST * 4096;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Wl = smem;
  char* Hl = smem + W_BYTES;
  float* epi = reinterpret_cast<float*>(smem + W_BYTES + H_BYTES);  // bias | scale | shift, NT*16 each (reused by the statistics reduction)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, l15 = lane & 15;
  const int X = k.X, Y = k.Y, Z = k.Z, cout = k.cout;

  for (int i = tid; i < W_BYTES / 16; i += 256) reinterpret_cast<uint4*>(Wl)[i] = reinterpret_cast<const uint4*>(k.wpack)[i];
  for (int i = tid; i < NT * 16; i += 256) {
    const bool ok = i < cout;
    const int cv = k.cout_mod > 0 ? i % k.cout_mod : i;
    epi[i] = ((ok && k.bias) ? k.bias[cv] : 0.f) + ((ok && k.bias2) ? k.bias2[cv] : 0.f);
    epi[NT * 16 + i] = (ok && k.scale) ? k.scale[cv] : 1.f;
    epi[2 * NT * 16 + i] = (ok && k.scale) ? k.shift[cv] : 0.f;
  }
  const float alpha = (k.act == VSSEG_ACT_PRELU && k.alpha) ? *k.alpha : 0.f;

  // ---- this thread's DMA pieces: LDS slot j = (u*4 + wave)*64 + lane holds piece (halo voxel j / G, 16-byte group (j % G) ^ swizzle) ----
  unsigned rel[NINST], hxy[NINST], p1mask = 0;
#pragma unroll
  for (int u = 0; u < NINST; ++u) {
    const int j = (u * 4 + wave) * 64 + lane;
    const int hv = j / G, cs = j % G, hz = hv % HZ, r = hv / HZ, hy = r % HY, hx = r / HY;
    const int c16 = cs ^ sc_swz<G>(hy, hz);
    const bool ok = j < PIECES;
    rel[u] = ok ? (unsigned)((hx * Y + hy) * Z + hz) * (unsigned)k.in_vox_bytes + (unsigned)c16 * 16u : 0u;  // padding lanes copy 16 harmless bytes into the buffer's padding
    hxy[u] = ok ? (unsigned)(hx | (hy << 8)) : 0xffffu;
    if (ok && c16 >= k.in_csplit_pc) p1mask |= 1u << u;
  }
  const bool in_two = k.in1 != k.in0;
  // ---- MFMA operand addressing: K-group p = ks*4 + g -> (tap p / G, channel group p % G); lane column l15 -> voxel (vy = l15 >> 2 (+4 for odd M-tiles), vz = l15 & 3)
  const int vy0 = l15 >> 2, vz = l15 & 3;
  int koff[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    const int p = ks * 4 + g, tap = p / G, cg = p % G;
    const int dx = TAPS == 9 ? tap / 3 : (TAPS == 4 ? tap >> 1 : 0), dy = TAPS == 9 ? tap % 3 : (TAPS == 4 ? tap & 1 : 0);
    koff[ks] = tap < TAPS ? ((dx * HY + dy) * HZ) * CINB + ((cg ^ sc_swz<G>(vy0 + dy, vz)) * 16) : 0;  // padded K-groups: zero weights times valid data
  }
  const int vb0 = (((wave * (MT / 2)) * HY + vy0) * HZ + vz) * CINB;  // M-tile m: + (m & 1) * 4 rows of y, + (m >> 1) rows of x (immediates; neither changes the swizzle)
  const unsigned ov0 = (unsigned)(((wave * (MT / 2)) * Y + vy0) * Z + vz);
  const unsigned out_es = k.out_f32 ? 4u : 2u;
  const bool vec_store = (cout & 3) == 0;
  const bool simple = vec_store && !k.out_f32 && !k.scale && (k.act == VSSEG_ACT_NONE || k.act == VSSEG_ACT_PRELU);
  const int ekind = !simple ? 2 : (k.aux_mode == 3 ? 1 : 0);
  const float alpha_eff = k.act == VSSEG_ACT_PRELU ? alpha : 1.f;
  const char* Wlane = Wl + lane * 16;
  float ssum[STATS ? NT : 1][4], ssq[STATS ? NT : 1][4];  // sum / sum of squares of the output
#pragma unroll
  for (int t = 0; t < (STATS ? NT : 1); ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) { ssum[t][r] = 0.f; ssq[t][r] = 0.f; }
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, wgs = gridDim.x >> 3;
  __syncthreads();

  for (int tl = k.walk ? slot : (int)blockIdx.x; tl < (k.walk ? k.per_xcd : k.tiles); tl += (k.walk ? wgs : (int)gridDim.x)) {
    const int ti = k.walk ? xcd * k.per_xcd + tl : tl;
    if (ti >= k.tiles) break;
    unsigned b = (unsigned)ti;
    unsigned qq = sc_div(b, k.mg_tz, k.ntz); const int tz = (int)(b - qq * k.ntz); b = qq;
    qq = sc_div(b, k.mg_ty, k.nty); const int ty = (int)(b - qq * k.nty); b = qq;
    qq = sc_div(b, k.mg_tx, k.ntx); const int tx = (int)(b - qq * k.ntx); const int n = (int)qq;
    const int x0 = tx * SC_TX, y0 = ty * SC_TY, z0 = tz * SC_TZ;
    const int64_t ivox = (((int64_t)n * X + (x0 - R)) * Y + (y0 - R)) * Z + z0;
    const int64_t ovox = (((int64_t)n * X + x0) * Y + y0) * Z + z0;
    const char* org0 = k.in0 + ivox * k.in_vox_bytes;
    const char* org1 = k.in1 + ivox * k.in_vox_bytes;
    // output voxel of (M-tile m, channel tile t) and the channel of this lane's 4 values inside the OUTPUT tensor
    auto out_vox = [&](int m, int t) -> int64_t {
      if constexpr (PS) {
        const int OY = 2 * Y;
        const int64_t base = (((int64_t)n * (2 * X) + 2 * x0) * OY + 2 * y0) * Z + z0;
        const int cls = k.ps_cls0 + (t >> tpc_sh);  // parity class (px, py) = (cls >> 1, cls & 1)
        return base + (int64_t)((2 * (wave * (MT / 2) + (m >> 1)) + (cls >> 1)) * OY + 2 * (vy0 + (m & 1) * 4) + (cls & 1)) * Z + vz;
      } else {
        return ovox + ov0 + (unsigned)((m & 1) * 4 * Z + (m >> 1) * Y * Z);
      }
    };
    auto out_ch = [&](int t) -> int { return PS ? (t & ((1 << tpc_sh) - 1)) * 16 + g * 4 : t * 16 + g * 4; };
    const bool interior = (R == 0 || (x0 > 0 && y0 > 0)) && (RH == 0 || (x0 + SC_TX < X && y0 + SC_TY < Y));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave has read the previous tile's halo
    if (interior) {
      if (!in_two) {
#pragma unroll
        for (int u = 0; u < NINST; ++u) sc_dma16(org0 + rel[u], Hl + (u * 4 + wave) * 1024);
      } else {
#pragma unroll
        for (int u = 0; u < NINST; ++u) sc_dma16(((p1mask >> u) & 1u ? org1 : org0) + rel[u], Hl + (u * 4 + wave) * 1024);
      }
    } else {
#pragma unroll
      for (int u = 0; u < NINST; ++u) {
        const int gx = x0 - R + (int)(hxy[u] & 255u), gy = y0 - R + (int)(hxy[u] >> 8);
        const bool ok = hxy[u] != 0xffffu && (unsigned)gx < (unsigned)X && (unsigned)gy < (unsigned)Y;
        sc_dma16(ok ? (const void*)(((p1mask >> u) & 1u ? org1 : org0) + rel[u]) : k.zeros, Hl + (u * 4 + wave) * 1024);
      }
    }
    // auxiliary operands of the epilogue: ordinary loads issued behind the DMA, in flight with it
    uint2 auxv[AUXM ? MT : 1][AUXM ? NT : 1];
    float gatev[AUXM ? MT : 1];
    if constexpr (AUXM) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        if (!PS && k.aux_mode == 4) gatev[m] = k.gate[out_vox(m, 0)];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int c = t * 16 + g * 4;
          const char* ap = (t * 16 >= k.aux_csplit ? k.aux1 : k.aux0) + out_vox(m, t) * k.aux_vox_bytes + out_ch(t) * 2;
          if (c < cout) auxv[m][t] = *reinterpret_cast<const uint2*>(ap);
        }
      }
    }
    // fused BatchNorm-backward reduction: the layer's pre-activation and keep-mask byte for this lane's output elements
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      bf16x8 w[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) w[t] = *reinterpret_cast<const bf16x8*>(Wlane + (ks * NT + t) * 1024);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const bf16x8 av = *reinterpret_cast<const bf16x8*>(Hl + vb0 + koff[ks] + (m & 1) * (4 * HZ * CINB) + (m >> 1) * (HY * HZ * CINB));
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[t], av, acc[m][t], 0, 0, 0);
      }
    }

    // ---- epilogue: bias (+ statistics) (+ eval affine) + activation (+ auxiliary operand), 4 channels per lane.  The launch's epilogue kind is
    //      wave-uniform: it is dispatched ONCE per tile and each kind is straight-line code (per-element uniform branches made the epilogue a
    //      forest of ~150 s_cbranch per tile).  Kind 0: bf16 vector stores, no affine, identity / PReLU (alpha_eff = 1 for the identity), additive
    //      auxiliary operand scaled by gt (1, or 1 + gate); kind 1: the same with the ReLU-mask operand; kind 2: everything else.
    auto epilogue = [&](auto kind_c) {
      constexpr int KIND = decltype(kind_c)::value;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        float gt = 1.f;
        if constexpr (AUXM) gt = (!PS && k.aux_mode == 4) ? 1.f + gatev[m] : 1.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int c = t * 16 + g * 4;
          if (c >= cout) continue;
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
          char* op = (t * 16 >= k.out_csplit ? k.out1 : k.out0) + out_vox(m, t) * k.out_vox_bytes + out_ch(t) * (int)out_es;
          if constexpr (KIND != 2) {
            st4(reinterpret_cast<bf16_t*>(op), make_float4(val[0], val[1], val[2], val[3]));
          } else if (vec_store) {
            if (k.out_f32) st4(reinterpret_cast<float*>(op), make_float4(val[0], val[1], val[2], val[3]));
            else st4(reinterpret_cast<bf16_t*>(op), make_float4(val[0], val[1], val[2], val[3]));
          } else {  // 1- and 2-channel outputs (attention map, logits)
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
  }

  if constexpr (STATS) {  // per-channel sum / sum of squares of this workgroup's voxels: shuffle tree -> one LDS row per wave, summed in wave order -> the layer's
                          // sharded statistics as fixed-point integer atomics (order-independent: vsseg_fx_add; layout of vsseg_igemm_desc.stats)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);  // [4 waves][2][NT*16]: the weights are no longer needed
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
      if (c < cout) vsseg_fx_add(&st[which * k.stats_stride + (k.cout_mod > 0 ? c % k.cout_mod : c)], (double)v, VSSEG_FX_STAT, k.fxflag);
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------
template <int CIN, int NT, int TAPS> static int sc_lds() {
  constexpr int G = CIN / 8, R = (TAPS == 9 ? 2 : (TAPS == 4 ? 1 : 0));  // halo voxels added per axis
  constexpr int PIECES = (2 * sc_mt(CIN, TAPS) + R) * (SC_TY + R) * SC_TZ * G, NINST = (PIECES + 255) / 256, KSTEPS = (TAPS * G + 3) / 4;
  return KSTEPS * NT * 1024 + NINST * 4096 + 5 * NT * 16 * 4 + 16;
}
template <int CIN, int NT, int TAPS, int MODE> static int sc_launch_mode(const SconvK& k, hipStream_t s) {
  static int per_cu = 0;
  const int lds = sc_lds<CIN, NT, TAPS>();
  if (!per_cu) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&sconv_kernel<CIN, NT, TAPS, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sconv_kernel<CIN, NT, TAPS, MODE>, 256, lds) != hipSuccess || n < 1) n = 1;
    per_cu = n > 3 ? 3 : n;  // measured: 3 resident workgroups per CU stream fastest; more only spread the DRAM pages in flight
  }
  int grid = 256 * per_cu;
  const int need = (k.tiles + 7) / 8 * 8;
  if (grid > need) grid = need;
  hipLaunchKernelGGL((sconv_kernel<CIN, NT, TAPS, MODE>), dim3((unsigned)grid), dim3(256), lds, s, k);
  VSSEG_LAUNCH_CHECK("vsseg_igemm (streaming)");
  return VSSEG_OK;
}
template <int CIN, int NT, int TAPS> static int sc_launch(const SconvK& k, hipStream_t s) {
  if (k.stats) return sc_launch_mode<CIN, NT, TAPS, 1>(k, s);
  if (k.aux_mode) return sc_launch_mode<CIN, NT, TAPS, 2>(k, s);
  return sc_launch_mode<CIN, NT, TAPS, 0>(k, s);
}

typedef int (*sc_fn_t)(const SconvK&, hipStream_t);
struct ScEntry { int cin, nt, taps; sc_fn_t fn; int (*lds)(); };
#define SC_E(C, N, T) {C, N, T, sc_launch<C, N, T>, sc_lds<C, N, T>}
static const ScEntry sc_table[] = {SC_E(8, 1, 9),  SC_E(8, 2, 9),  SC_E(16, 1, 9), SC_E(16, 2, 9), SC_E(16, 4, 9), SC_E(32, 1, 9), SC_E(32, 2, 9), SC_E(32, 4, 9),
                                   SC_E(16, 1, 1), SC_E(16, 2, 1), SC_E(32, 1, 1), SC_E(32, 2, 1), SC_E(32, 4, 1), SC_E(64, 2, 1), SC_E(64, 4, 1), SC_E(64, 2, 9),
                                   SC_E(32, 3, 1), SC_E(48, 2, 1), SC_E(48, 3, 1), SC_E(48, 6, 1), SC_E(96, 3, 1), SC_E(96, 6, 1),  // the 1x1x1 residual convolutions of the level-2 units (32 -> 48, 96 -> 48) and their data gradients
                                   SC_E(16, 4, 4), SC_E(32, 4, 4), SC_E(48, 4, 4)};  // taps 4: fused output-parity classes (pixel shuffle): 4 classes x 16 channels, or 2 classes x 32 channels per launch

static const ScEntry* sc_find(const vsseg_igemm_desc* d, const char** why) {
  *why = nullptr;
  auto no = [&](const char* w) { *why = w; return (const ScEntry*)nullptr; };
  if (d->in.dtype != VSSEG_BF16) return no("input is not bf16");
  const int mt = sc_mt(d->ck, d->ntaps), SC_TX = 2 * mt;
  if (d->nchunks != 1 || d->nsplit != 1 || d->mtw != mt) return no("needs nchunks = nsplit = 1 and mtw = 4 (2 for 64 input channels x 9 taps)");
  if (d->tile[0] != SC_TX || d->tile[1] != SC_TY || d->tile[2] != SC_TZ) return no("tile must be 8x8x4 (4x8x4 for 64 input channels x 9 taps)");
  const bool ps = d->depth == -4;  // fused output-parity classes: coarse lattice in, fine tensor (2x, 2y, z) out, 4 taps (+0 / +1 in x and y)
  if (ps) {
    if (d->is[0] != 1 || d->is[1] != 1 || d->is[2] != 1 || d->os[0] != 2 || d->os[1] != 2 || d->os[2] != 1 || d->oo[1] || d->oo[2]) return no("pixel-shuffle launches need is = 1, os = (2, 2, 1), oo = (px, 0, 0)");
    if (d->q[0] != d->in.x || d->q[1] != d->in.y || d->q[2] != d->in.z || 2 * d->q[0] != d->out.x || 2 * d->q[1] != d->out.y || d->q[2] != d->out.z) return no("pixel-shuffle output must be (2x, 2y, z) of the lattice");
    // nt = 4 channel tiles: all four classes of 16 channels (oo = 0), or the classes (px, 0), (px, 1) of 32 channels (oo = (px, 0, 0))
    if (d->ntaps != 4 || d->nt != 4 || (d->out.c != 16 && d->out.c != 32) || d->cout_mod != d->out.c || d->out.ptr2 || d->out.dtype != VSSEG_BF16) return no("pixel-shuffle launches need 4 taps, 4 channel tiles, 16 or 32 bf16 output channels (cout_mod = channels), a one-part output");
    if ((d->out.c == 16 && d->oo[0] != 0) || (unsigned)d->oo[0] > 1u) return no("pixel-shuffle class offset");
    for (int t = 0; t < 4; ++t)
      if (d->tap_off[t][0] != (t >> 1) || d->tap_off[t][1] != (t & 1) || d->tap_off[t][2] != 0) return no("taps are not the 2x2x1 neighbourhood in (x, y) order");
    if (d->res_mode != VSSEG_RES_NONE) return no("pixel-shuffle launches support statistics or accumulate only");
  } else {
    for (int a = 0; a < 3; ++a)
      if (d->is[a] != 1 || d->os[a] != 1 || d->oo[a] != 0) return no("stride-1 lattices only");
    if (d->q[0] != d->in.x || d->q[1] != d->in.y || d->q[2] != d->in.z || d->q[0] != d->out.x || d->q[1] != d->out.y || d->q[2] != d->out.z) return no("lattice, input and output extents differ");
  }
  if (d->q[0] % SC_TX || d->q[1] % SC_TY || d->q[2] % SC_TZ) return no("extent is not a multiple of the tile");
  if (ps) {
  } else if (d->ntaps == 9) {
    for (int t = 0; t < 9; ++t)
      if (d->tap_off[t][0] != t / 3 - 1 || d->tap_off[t][1] != t % 3 - 1 || d->tap_off[t][2] != 0) return no("taps are not the 3x3x1 stencil in (x, y) order");
  } else if (d->ntaps == 1) {
    if (d->tap_off[0][0] || d->tap_off[0][1] || d->tap_off[0][2]) return no("single tap with an offset");
  } else return no("3x3x1 or 1x1x1 taps only");
  if (d->in.c != d->ck || d->in.pitch % 8 || ((uintptr_t)d->in.ptr & 15) || ((uintptr_t)d->in.ptr2 & 15)) return no("input must be one channel chunk of 16-byte aligned voxel rows");
  if (d->ksteps != (d->ntaps * (d->ck / 8) + 3) / 4) return no("ksteps");
  if ((!ps && d->out.c > d->nt * 16) || (d->out.dtype != VSSEG_BF16 && d->out.dtype != VSSEG_F32)) return no("output channels / dtype");
  if ((d->out.c & 3) == 0 && (d->out.pitch & 3)) return no("output pitch");
  if (d->stats && (d->accumulate || d->res_mode != VSSEG_RES_NONE)) return no("statistics combined with a residual");
  if (d->accumulate && d->res_mode != VSSEG_RES_NONE) return no("accumulate combined with a residual");
  if (d->accumulate || d->res_mode != VSSEG_RES_NONE) {
    const vsseg_tensor& a = d->accumulate ? d->out : d->res;
    if ((d->out.c & 3) || (a.pitch & 3) || a.c < d->out.c || a.dtype != VSSEG_BF16) return no("auxiliary tensor layout / dtype");
  }
  if ((int64_t)d->in.n * d->q[0] * d->q[1] * d->q[2] / 256 >= (1ll << 24)) return no("too many tiles");  // tile index * tiles-per-axis < 2^32: exact multiply-high quotients
  for (const ScEntry& e : sc_table)
    if (e.cin == d->ck && e.nt == d->nt && e.taps == d->ntaps) return &e;
  return no("no instantiation for this (channels, nt, taps)");
}

int vsseg_sconv_lds_bytes(const vsseg_igemm_desc* d) {
  const char* why;
  const ScEntry* e = sc_find(d, &why);
  if (!e) { vsseg_set_error("vsseg_igemm: depth -2 / -4 (streaming kernel) not applicable: %s", why); return VSSEG_EINVAL; }
  return e->lds();
}

int vsseg_sconv_launch(const vsseg_igemm_desc* d, const void* zeros, hipStream_t s) {
  const char* why;
  const ScEntry* e = sc_find(d, &why);
  if (!e) { vsseg_set_error("vsseg_igemm: depth -2 / -4 (streaming kernel) not applicable: %s", why); return VSSEG_EINVAL; }
  SconvK k;
  auto magic = [](int dv) { return dv <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)dv - 1) / (unsigned)dv); };
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
  if (k.aux_mode) {
    const vsseg_tensor& a = d->accumulate ? d->out : d->res;
    const int aes = a.dtype == VSSEG_F32 ? 4 : 2;
    k.aux0 = reinterpret_cast<const char*>(a.ptr);
    k.aux1 = a.ptr2 ? reinterpret_cast<const char*>(a.ptr2) - (int64_t)a.csplit * aes : k.aux0;
    k.aux_csplit = a.ptr2 ? a.csplit : 0x7fffffff;
    k.aux_vox_bytes = a.pitch * aes;
  }
  VSSEG_CHECK(k.aux_mode != 4 || d->gate, "vsseg_igemm: RES_GATE needs the gate map");
  k.gate = d->gate;
  k.wpack = reinterpret_cast<const char*>(d->wpack);
  k.bias = d->bias; k.bias2 = d->bias2; k.scale = d->scale; k.shift = d->shift; k.alpha = d->alpha;
  k.stats = d->stats; k.stats_stride = d->stats_stride;
  VSSEG_FX_FLAG(fxflag_, "vsseg_igemm (streaming kernel)");
  k.fxflag = fxflag_;
  k.zeros = zeros;
  k.act = d->act; k.cout = d->depth == -4 ? d->nt * 16 : d->out.c; k.cout_mod = d->cout_mod;
  k.ps_cls0 = d->depth == -4 ? 2 * d->oo[0] : 0;
  k.ps_tpc_shift = (d->depth == -4 && d->out.c == 32) ? 1 : 0;
  k.X = d->q[0]; k.Y = d->q[1]; k.Z = d->q[2];
  k.ntx = k.X / (2 * sc_mt(d->ck, d->ntaps)); k.nty = k.Y / SC_TY; k.ntz = k.Z / SC_TZ;
  k.mg_tx = magic(k.ntx); k.mg_ty = magic(k.nty); k.mg_tz = magic(k.ntz);
  k.tiles = d->in.n * k.ntx * k.nty * k.ntz;
  k.per_xcd = (k.tiles + 7) / 8;
  k.walk = 1;
  return e->fn(k, s);
}
