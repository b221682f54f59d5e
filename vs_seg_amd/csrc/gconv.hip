// Gathering marching convolution (launch plans with depth -9): the stride-(2,2,1) 3x3x1 launches between the levels of the 2.5D U-Net that read a FINE tensor and write a coarse one
// — the strided convolutions 16 -> 16 (level 0 -> 1) and 32 -> 32 (level 1 -> 2) and the data gradients of the transposed convolutions 32 -> 16 / 48 -> 32 on the way back up
// (ref:params/networks/nets/unet2d5_spvPA.py:56-93, blocks/convolutions.py:114-156; SURVEY §8a rows 3, 7, 40, 45 and their autograd):
//     out[x][y][z] = sum over (dx, dy) of W[dx][dy] * in[2x + dx - 1][2y + dy - 1][z]
// They ran on the general kernel (tile 4x8x4: the (2*4+1) x (2*8+1) x 4 halo of every tile fetched by its workgroup, 2.5-4.2 TB/s).  Here, as in mconv.hip, a workgroup owns a
// column (sample n, coarse rows [y0, y0 + TYB), slices [z0, z0 + TZ)) and MARCHES along x with a ring of FINE planes in LDS: output plane x reads the fine planes 2x-1, 2x, 2x+1,
// two new planes are fetched per step (one whole step ahead), every fine voxel of the column travels from HBM once.  A fine plane is stored as two HALF planes — its odd rows
// 2(y0 + j) - 1 and its even rows 2(y0 + j) — each in mconv.hip's layout [row j][piece'][z] (piece' = (piece + 2 * (j * RS / 16)) mod G): the three row taps of an M-tile's
// sixteen voxels then read CONSECUTIVE rows of one half (dy = 0: odd half row j, dy = 1: even half row j, dy = 2: odd half row j + 1), which is the access mconv.hip's layout is
// conflict-free for (tools/lds_conflicts.py); the de-interleaving costs nothing — the DMA writes LDS in lane order, the permutation is in the global address each lane fetches.
// Same packed weights ([K-steps][tiles][64 lanes][8], K order (tap, 8-channel group)), MFMA operand order and accumulation order as the general kernel's plan with the whole input
// in one chunk: bit-identical results (tests/test_gpu_ops.py::test_gathering_marching_kernel_*).  Epilogues: plain, + BatchNorm statistics, accumulate, eval affine + activation.
#include "gconv.h"
#include <type_traits>

constexpr int GC_NR = 6;  // ring slots: fine planes 2x-1, 2x, 2x+1 + the two in flight (+ one: the slot index is i % 6)

struct GconvK {
  const char* in;
  char* out;
  const char* wpack;
  const float *bias, *bias2, *scale, *shift, *alpha;
  double* stats;
  unsigned* fxflag;
  const void* zeros;
  int in_vox_bytes, out_vox_bytes, accumulate, act, cout, stats_stride;
  int X, Y, Z;     // coarse (output) extents; the input is (XF, YF, Z)
  int XF, YF;
  int lx, nxs, nyb, nzb;
};

// MODE: 0 plain, 1 + BatchNorm statistics, 2 accumulate (out += ...), 3 eval affine + activation
template <int CIN, int NT, int TZ, int MT, int MODE>
__global__ __launch_bounds__(256, 2) void gconv_kernel(const GconvK k) {
  constexpr bool STATS = MODE == 1, ACC = MODE == 2, EVAL = MODE == 3;
  constexpr int G = CIN / 8, RS = TZ * G, RPM = 16 / TZ, TYB = MT * 4 * RPM, HROWS = TYB + 1;
  constexpr int HALF_SLOTS = HROWS * RS, HS16 = (HALF_SLOTS + 15) / 16 * 16, PLANE_SLOTS = 2 * HS16, PLANE_BYTES = (PLANE_SLOTS * 16 + 1023) / 1024 * 1024, NINST = (PLANE_SLOTS + 255) / 256;  // (whole 1 KiB DMA rows: the tail of the last row is written too, with zeros)
  constexpr int KSTEPS = (9 * G + 3) / 4, W_BYTES = KSTEPS * NT * 1024, MT_BYTES = RPM * RS * 16;
  static_assert((G & (G - 1)) == 0 && 16 % TZ == 0, "channel groups and z slices are powers of two");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Wl = smem;
  char* Rl = smem + W_BYTES;
  float* epi = reinterpret_cast<float*>(smem + W_BYTES + GC_NR * PLANE_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, l15 = lane & 15;
  const int X = k.X, Y = k.Y, Z = k.Z, XF = k.XF, YF = k.YF, cout = k.cout;

  int b = vsseg_xcd_contiguous(blockIdx.x, gridDim.x);
  const int zb = b % k.nzb; b /= k.nzb;
  const int yb = b % k.nyb; b /= k.nyb;
  const int xs = b % k.nxs; const int n = b / k.nxs;
  const int y0 = yb * TYB, z0 = zb * TZ, xb = xs * k.lx, steps = min(k.lx, X - xb);

  for (int i = tid; i < W_BYTES / 16; i += 256) reinterpret_cast<uint4*>(Wl)[i] = reinterpret_cast<const uint4*>(k.wpack)[i];
  for (int i = tid; i < GC_NR * PLANE_BYTES / 16; i += 256) reinterpret_cast<uint4*>(Rl)[i] = make_uint4(0u, 0u, 0u, 0u);  // rows outside the image are never fetched: zero
  for (int i = tid; i < NT * 16; i += 256) {
    const bool ok = i < cout;
    epi[i] = ((ok && k.bias) ? k.bias[i] : 0.f) + ((ok && k.bias2) ? k.bias2[i] : 0.f);
    epi[NT * 16 + i] = (ok && k.scale) ? k.scale[i] : 1.f;
    epi[2 * NT * 16 + i] = (ok && k.scale) ? k.shift[i] : 0.f;
  }
  const float slope = !EVAL ? 1.f : (k.act == VSSEG_ACT_PRELU ? (k.alpha ? *k.alpha : 0.f) : (k.act == VSSEG_ACT_RELU ? 0.f : 1.f));

  // ---- this thread's DMA pieces of a fine plane: LDS slot j = (u*4 + wave)*64 + lane = half * HS16 + (row jr) * RS + piece' * TZ + z; half 0 = the odd fine rows 2(y0 + jr) - 1,
  //      half 1 = the even fine rows 2(y0 + jr)
  int rel[NINST];
  unsigned okmask = 0;
#pragma unroll
  for (int u = 0; u < NINST; ++u) {
    const int j = (u * 4 + wave) * 64 + lane;
    const int half = j / HS16, within = j - half * HS16, jr = within / RS, rem = within % RS, pp = rem / TZ, z = rem % TZ;
    const int pc = (pp - 2 * (jr * RS / 16)) & (G - 1);
    const int fy = 2 * (y0 + jr) - 1 + half;
    const bool ok = j < PLANE_SLOTS && within < HALF_SLOTS && (half == 0 || jr < TYB) && (unsigned)fy < (unsigned)YF;
    rel[u] = ok ? (fy * Z + z) * k.in_vox_bytes + pc * 16 : 0;
    if (ok) okmask |= 1u << u;
  }
  const int64_t fplane = (int64_t)YF * Z * k.in_vox_bytes;
  const char* org = k.in + (((int64_t)n * XF) * YF * Z + z0) * k.in_vox_bytes;  // fine voxel (n, 0, 0, z0)
  auto issue = [&](int i) __attribute__((always_inline)) {  // fine plane i of the segment (fx = 2*xb - 1 + i) into ring slot i % 6; planes outside the image are zero
    const int fx = 2 * xb - 1 + i;
    char* dst = Rl + (i % GC_NR) * PLANE_BYTES;
    const bool inside = (unsigned)fx < (unsigned)XF;
    const char* p0 = org + (int64_t)fx * fplane;
#pragma unroll
    for (int u = 0; u < NINST; ++u)
      if ((u * 4 + wave) * 64 < PLANE_SLOTS)  // (wave-uniform: whole 1 KiB rows; the slots of a row that hold nothing fetch the zero page)
        vsseg_dma16((inside && ((okmask >> u) & 1u)) ? (const void*)(p0 + rel[u]) : k.zeros, dst + (u * 4 + wave) * 1024);
  };

  // ---- MFMA operand addressing: K-group p = ks*4 + g -> (tap p / G = dx*3 + dy, piece p % G); lane column l15 -> voxel (row l15 / TZ, z l15 % TZ) of the M-tile
  const int rr = l15 / TZ, zz = l15 % TZ;
  int koff[KSTEPS], dxk[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    int p = ks * 4 + g;
    if (p >= 9 * G) p -= 9 * G;  // padded K-groups: zero weights times a genuine tap of the same voxel
    const int tap = p / G, pc = p % G, dy = tap % 3;
    const int half = dy == 1 ? 1 : 0, jr = rr + (dy == 2 ? 1 : 0);  // + the M-tile's first row (a multiple of RPM: it does not change the swizzle term)
    koff[ks] = (half * HS16 + jr * RS + ((pc + 2 * (jr * RS / 16)) & (G - 1)) * TZ + zz) * 16 + wave * (MT * MT_BYTES);
    dxk[ks] = tap / 3;
  }
  const char* Wlane = Wl + lane * 16;
  float ssum[STATS ? NT : 1][4], ssq[STATS ? NT : 1][4];
#pragma unroll
  for (int t = 0; t < (STATS ? NT : 1); ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) { ssum[t][r] = 0.f; ssq[t][r] = 0.f; }
  // output voxel of M-tile m (this lane's column) at x: ((n*X + x)*Y + y0 + (wave*MT + m)*RPM + rr)*Z + z0 + zz
  const int64_t ocol = (((int64_t)n * X) * Y + y0 + (wave * MT) * RPM + rr) * Z + z0 + zz;
  const int64_t oplane = (int64_t)Y * Z;

  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();  // the ring is zeroed before any DMA writes it
  // accumulate: the previous gradient at the voxels a step stores to is loaded ONE STEP AHEAD, in front of the next planes' DMAs (loads return in order: behind them the epilogue
  // would wait for the planes; hipcc's own wait for these registers knows nothing of the inline-assembly DMAs)
  uint2 auxv[ACC ? MT : 1][ACC ? NT : 1], auxn[ACC ? MT : 1][ACC ? NT : 1];
  auto load_aux = [&](int s, uint2 (&av)[ACC ? MT : 1][ACC ? NT : 1]) __attribute__((always_inline)) {
    if constexpr (ACC) {
      const int64_t v0 = ocol + (int64_t)(xb + s) * oplane;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int c = t * 16 + g * 4;
          av[m][t] = c < cout ? *reinterpret_cast<const uint2*>(k.out + (v0 + (int64_t)m * RPM * Z) * k.out_vox_bytes + c * 2) : make_uint2(0u, 0u);
        }
    }
  };
  load_aux(0, auxn);
  issue(0);
  issue(1);
  issue(2);
  for (int s = 0; s < steps; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the fine planes 2s+1, 2s+2 have landed (and the previous step's stores have left)
    __builtin_amdgcn_s_barrier();                      // ... everybody's; and every wave has finished reading the planes of step s-1
    const int64_t vox0 = ocol + (int64_t)(xb + s) * oplane;
    if constexpr (ACC) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) auxv[m][t] = auxn[m][t];
      if (s + 1 < steps) load_aux(s + 1, auxn);
    }
    if (s + 1 < steps) { issue(2 * s + 3); issue(2 * s + 4); }
    const int sl0 = ((2 * s) % GC_NR) * PLANE_BYTES, sl1 = ((2 * s + 1) % GC_NR) * PLANE_BYTES, sl2 = ((2 * s + 2) % GC_NR) * PLANE_BYTES;
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
      const char* hb = Rl + koff[ks] + (dxk[ks] == 0 ? sl0 : (dxk[ks] == 1 ? sl1 : sl2));
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const bf16x8 av = *reinterpret_cast<const bf16x8*>(hb + m * MT_BYTES);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[t], av, acc[m][t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      char* orow = k.out + (vox0 + (int64_t)m * RPM * Z) * k.out_vox_bytes;
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
        if constexpr (EVAL) {
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
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  if constexpr (STATS) {  // per-channel sum / sum of squares of this workgroup's outputs -> the layer's sharded statistics (fixed-point atomics: order-independent; mconv.hip)
    __syncthreads();
    float* red = reinterpret_cast<float*>(Rl);  // [4 waves][2][NT*16]: the ring is no longer needed
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float su = ssum[t][r], q2 = ssq[t][r];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { su += __shfl_xor(su, o, 64); q2 += __shfl_xor(q2, o, 64); }
        if (l15 == 0) {
          red[wave * (2 * NT * 16) + t * 16 + g * 4 + r] = su;
          red[wave * (2 * NT * 16) + NT * 16 + t * 16 + g * 4 + r] = q2;
        }
      }
    __syncthreads();
    double* st = k.stats + (int64_t)(blockIdx.x % VSSEG_STAT_SHARDS) * 2 * k.stats_stride;
    for (int i = tid; i < 2 * NT * 16; i += 256) {
      const int which = i / (NT * 16), c = i - which * NT * 16;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) v += red[w * (2 * NT * 16) + i];
      if (c < cout) vsseg_fx_add(&st[which * k.stats_stride + c], (double)v, VSSEG_FX_STAT, k.fxflag);
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------
template <int CIN, int NT, int TZ, int MT> static int gc_lds() {
  constexpr int G = CIN / 8, RS = TZ * G, RPM = 16 / TZ, TYB = MT * 4 * RPM, HS16 = ((TYB + 1) * RS + 15) / 16 * 16;
  return ((9 * G + 3) / 4) * NT * 1024 + GC_NR * ((2 * HS16 * 16 + 1023) / 1024 * 1024) + 3 * NT * 16 * 4 + 16;
}
template <int CIN, int NT, int TZ, int MT, int MODE> static int gc_launch_mode(const GconvK& k, int grid, hipStream_t s) {
  static bool attr_set_dev[16] = {}; bool& attr_set = vsseg_dev_once(attr_set_dev);  // per device: the LDS opt-in is a per-device function attribute
  const int lds = gc_lds<CIN, NT, TZ, MT>();
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gconv_kernel<CIN, NT, TZ, MT, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((gconv_kernel<CIN, NT, TZ, MT, MODE>), dim3((unsigned)grid), dim3(256), lds, s, k);
  VSSEG_LAUNCH_CHECK("vsseg_igemm (gathering marching kernel)");
  return VSSEG_OK;
}
template <int CIN, int NT, int TZ, int MT> static int gc_launch(const GconvK& k, int grid, hipStream_t s) {
  if (k.stats) return gc_launch_mode<CIN, NT, TZ, MT, 1>(k, grid, s);
  if (k.accumulate) return gc_launch_mode<CIN, NT, TZ, MT, 2>(k, grid, s);
  if (k.scale || k.act != VSSEG_ACT_NONE) return gc_launch_mode<CIN, NT, TZ, MT, 3>(k, grid, s);
  return gc_launch_mode<CIN, NT, TZ, MT, 0>(k, grid, s);
}

typedef int (*gc_fn_t)(const GconvK&, int, hipStream_t);
struct GcEntry { int cin, nt, tz, mt; gc_fn_t fn; int (*lds)(); };
#define GC_E(C, N, Z, M) {C, N, Z, M, gc_launch<C, N, Z, M>, gc_lds<C, N, Z, M>}
// (input channels, 16-channel output tiles, TZ, M-tiles per wave): coarse rows per workgroup TYB = 64 * MT / TZ
static const GcEntry gc_table[] = {
    GC_E(16, 1, 4, 2), GC_E(16, 1, 8, 4), GC_E(16, 1, 4, 4),   // strided convolution 16 -> 16 (level 0 -> 1): rows 32 / 32 / 64
    GC_E(16, 2, 4, 2), GC_E(16, 2, 8, 4), GC_E(16, 2, 4, 4),   // data gradient of the transposed convolution 32 -> 16: 16 -> 32
    GC_E(32, 2, 2, 1), GC_E(32, 2, 4, 2),                      // strided convolution 32 -> 32 (level 1 -> 2): rows 32
    GC_E(32, 3, 2, 1), GC_E(32, 3, 4, 2)};                     // data gradient of the transposed convolution 48 -> 32: 32 -> 48

static const GcEntry* gc_find(const vsseg_igemm_desc* d, const char** why) {
  *why = nullptr;
  auto no = [&](const char* w) { *why = w; return (const GcEntry*)nullptr; };
  if (d->in.dtype != VSSEG_BF16 || d->out.dtype != VSSEG_BF16) return no("input and output must be bf16");
  if (d->in.ptr2 || d->out.ptr2) return no("one-part tensors only");
  if (d->nchunks != 1 || d->nsplit != 1 || d->ntaps != 9 || d->class_split) return no("needs nchunks = nsplit = 1 and the 9 taps of a 3x3x1 stencil");
  if (d->is[0] != 2 || d->is[1] != 2 || d->is[2] != 1 || d->os[0] != 1 || d->os[1] != 1 || d->os[2] != 1 || d->oo[0] || d->oo[1] || d->oo[2]) return no("is = (2, 2, 1), os = 1, oo = 0");
  for (int t = 0; t < 9; ++t)
    if (d->tap_off[t][0] != t / 3 - 1 || d->tap_off[t][1] != t % 3 - 1 || d->tap_off[t][2] != 0) return no("taps are not the 3x3x1 stencil in (x, y) order");
  if (d->q[0] != d->out.x || d->q[1] != d->out.y || d->q[2] != d->out.z || d->in.z != d->out.z || d->in.n != d->out.n) return no("lattice and output extents differ");
  if (d->in.x != 2 * d->q[0] || d->in.y != 2 * d->q[1]) return no("the input must be (2x, 2y, z) of the lattice");
  if (d->in.c != d->ck || d->in.pitch % 8 || ((uintptr_t)d->in.ptr & 15)) return no("input must be one channel chunk of 16-byte aligned voxel rows");
  if (d->ksteps != (9 * (d->ck / 8) + 3) / 4) return no("ksteps");
  if (d->out.c > d->nt * 16 || (d->out.c & 3) || (d->out.pitch & 3) || ((uintptr_t)d->out.ptr & 7)) return no("output channels / pitch");
  if (d->cout_mod > 0 || d->in_gate || d->res_tiles || d->in1 || d->res_mode != VSSEG_RES_NONE) return no("plain, statistics, accumulate and eval epilogues only");
  if (d->stats && d->accumulate) return no("statistics combined with accumulate");
  if ((d->stats || d->accumulate) && (d->scale || d->act != VSSEG_ACT_NONE)) return no("the eval affine / activation combined with statistics or accumulate");
  if (d->act == VSSEG_ACT_SIGMOID || (d->scale == nullptr) != (d->shift == nullptr)) return no("sigmoid epilogue / scale without shift");
  const int tz = d->tile[2], tyb = d->tile[1], mt = d->mtw;
  if ((tz != 2 && tz != 4 && tz != 8) || tyb != 64 * mt / tz || d->tile[0] < 1) return no("tile must be (x steps per workgroup, 64 * mtw / tz rows, tz in {2, 4, 8})");
  if (d->q[1] % tyb || d->q[2] % tz) return no("extent is not a multiple of the column block");
  for (const GcEntry& e : gc_table)
    if (e.cin == d->ck && e.nt == d->nt && e.tz == tz && e.mt == mt) return e.lds() <= 160 * 1024 ? &e : no("needs more than 160 KiB of LDS");
  return no("no instantiation for this (channels, nt, tz, mtw)");
}

int vsseg_gconv_lds_bytes(const vsseg_igemm_desc* d) {
  const char* why;
  const GcEntry* e = gc_find(d, &why);
  if (!e) { vsseg_set_error("vsseg_igemm: depth -9 (gathering marching kernel) not applicable: %s", why); return VSSEG_EINVAL; }
  return e->lds();
}

int vsseg_gconv_launch(const vsseg_igemm_desc* d, const void* zeros, hipStream_t s) {
  const char* why;
  const GcEntry* e = gc_find(d, &why);
  if (!e) { vsseg_set_error("vsseg_igemm: depth -9 (gathering marching kernel) not applicable: %s", why); return VSSEG_EINVAL; }
  GconvK k{};
  k.in = reinterpret_cast<const char*>(d->in.ptr);
  k.out = reinterpret_cast<char*>(d->out.ptr);
  k.wpack = reinterpret_cast<const char*>(d->wpack);
  k.bias = d->bias; k.bias2 = d->bias2; k.scale = d->scale; k.shift = d->shift; k.alpha = d->alpha;
  k.stats = d->stats; k.stats_stride = d->stats_stride;
  VSSEG_FX_FLAG(fxflag_, "vsseg_igemm (gathering marching kernel)");
  k.fxflag = fxflag_;
  k.zeros = zeros;
  k.in_vox_bytes = d->in.pitch * 2; k.out_vox_bytes = d->out.pitch * 2;
  k.accumulate = d->accumulate ? 1 : 0; k.act = d->act; k.cout = d->out.c;
  k.X = d->q[0]; k.Y = d->q[1]; k.Z = d->q[2]; k.XF = d->in.x; k.YF = d->in.y;
  k.lx = d->tile[0] > k.X ? k.X : d->tile[0];
  k.nxs = (k.X + k.lx - 1) / k.lx; k.nyb = k.Y / d->tile[1]; k.nzb = k.Z / d->tile[2];
  const int64_t grid = (int64_t)d->in.n * k.nxs * k.nyb * k.nzb;
  VSSEG_CHECK(grid > 0 && grid < (1ll << 24), "vsseg_igemm: bad grid");
  return e->fn(k, (int)grid, s);
}
