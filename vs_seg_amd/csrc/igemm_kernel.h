// Implicit-GEMM 3-D convolution on the gfx950 matrix cores.
//
// One kernel serves Conv3d forward, every parity class of ConvTranspose3d forward and both data gradients
// (reference ops: ref:params/networks/blocks/convolutions.py:114-146; autograd of them at ref:params/VSparams.py:461).
//
//   out[q*os+oo][n] = epilogue( sum_{tap t} sum_{c} in[q*is + off_t][c] * W[t][c][n] )
//
// Mapping to MFMA (16x16 output tiles, wave64):
//   * D rows   <- 16 output channels, D cols <- 16 lattice voxels ("swapped" GEMM), so after the MFMA a lane owns
//     4 consecutive channels of one voxel and the epilogue stores 8/16 contiguous bytes per lane.
//   * K        <- (tap, input channel).  K is cut in groups of 8 channels; lane-group g = lane>>4 of a K-step owns
//     group ks*4+g, i.e. one 16-byte (bf16) / 32-byte (f32) LDS read per lane per K-step.  bf16: one
//     v_mfma_f32_16x16x32_bf16 per (voxel tile, channel tile); f32: eight v_mfma_f32_16x16x4_f32 (exact fp32).
//
// Structure: persistent workgroups walk (tile, channel-chunk) stages.  The input halo tile [hx][hy][hz][ck] of stage s+1
// is fetched by LDS-DMA (global_load_lds, 16 B per lane, no VGPR round trip) into the second LDS buffer while stage s is
// multiplied and stored, so every CU always has a whole tile of HBM reads in flight — most layers of this network are
// HBM-bound, and a load->barrier->compute->store workgroup measured only ~1.6 TB/s.  Tiles are assigned
// so that each XCD walks one contiguous eighth of the lattice (neighbouring tiles share halo rows through that XCD's L2).
// Packed weights stay resident in LDS for single-chunk layers; per-channel epilogue constants live in LDS, so the hot
// loop issues no ordinary global load that would make hipcc drain the DMA queue (cdna_hip_programming.md §5).
#pragma once
#include "common.h"
#include <stdlib.h>

constexpr int PMAX = 8;  // 16-byte halo pieces per thread per stage (halo chunk <= 32 KiB)
constexpr int AMAX = 8;  // 16-byte pieces per thread of the auxiliary (residual / accumulate) output tile (12 for the 512-voxel tiles, MTW = 8)
constexpr int amax_for(int mtw) { return mtw == 8 ? 12 : AMAX; }

struct IgemmK {
  vsseg_igemm_desc d;
  int halo[3];
  int off_min[3];
  int ntile[3];
  int cgs;      // 8-channel groups per chunk
  int w_bytes;  // packed weights per chunk
  int h_bytes;  // halo chunk
  unsigned mg_ppv, mg_hyz, mg_hz;  // ceil(2^32 / d) for d = pieces per halo voxel, HY*HZ, HZ: exact quotients by one v_mul_hi_u32 (piece index -> halo coordinates)
  int h_stride, aux_stride;  // LDS bytes between consecutive ring buffers: h_bytes / aux_bytes rounded up to 1 KiB, so that the lanes of the last (partial)
                             // DMA instruction of a buffer land in its own padding — every lane of every DMA instruction is then issued unconditionally
  int lds_ktab, lds_epi, lds_w, lds_h, lds_aux, lds_pinfo;
  int npu;        // 16-byte halo pieces per thread and stage = rows of the per-thread coordinate table in LDS
  int aux_mode;   // 0 none, 1 accumulate (aux = out), 2 residual add, 3 ReLU mask, 4 gated add (aux * (1 + gate[voxel])): the aux tile is prefetched by DMA like the halo
  int aux_bytes;  // per buffer: tile voxels * NT*16 * aux element size (+ the gate floats) (0: aux handled by the slow path)
  int aux_gate_off;  // aux_mode 4 with the gate map DMA-prefetched: byte offset of the tile's 64*MTW gate floats inside an aux buffer; 0: ordinary loads
  int depth;      // prefetch distance in stages (1..3); the LDS rings hold depth+1 buffers
  int class_vox[8];  // class_split: output voxel offset (oo_x*OY + oo_y)*OZ + oo_z of each class (workgroup row)
  vsseg_tensor aux;
  int64_t total_tiles;
  const void* zeros;  // >= 16 bytes of zeros in global memory (source of out-of-bounds halo pieces)
  unsigned* fxflag;   // sticky range / non-finite flag of the fixed-point statistics (common.h)
  const struct TileDesc* tiles;
#ifdef VSSEG_IG_PROF
  unsigned long long* prof;  // tuning build only: per-phase shader-cycle sums of workgroup 0 / wave 0
#endif
};

// Per-tile descriptor, computed once per (geometry) by a setup kernel and cached: the main kernel fetches it with one scalar
// load per stage instead of re-deriving tile coordinates, origins and boundary flags from ~40 uniform values (which made
// hipcc spill ~500 SGPRs per kernel and left the HBM-bound layers instruction-issue-bound).
struct __attribute__((aligned(64))) TileDesc {
  int64_t in_vox;   // voxel index (incl. batch) of the halo origin inside the input tensor (meaningful when interior)
  int64_t out_vox;  // voxel index (incl. batch) of the tile's first output voxel
  int32_t g0[3];    // halo origin coordinates (may be negative)
  int32_t q0[3];    // lattice origin of the tile
  int32_t n;
  int32_t flags;    // bit 0: halo entirely inside the input; bit 1: tile entirely inside lattice and output
};

// Table order = execution order (an XCD's workgroups walk a contiguous range of the table, ~64 tiles at a time).  Tiles are
// ordered (sample, x-band of `xb` tiles, y, x inside the band, z): the tiles in flight on one XCD then form a compact
// xb x 1..2 x nz brick whose x- and y-neighbours were fetched at most a few steps earlier, so the halo overlap between
// neighbouring tiles is served by that XCD's L2 instead of being fetched from HBM again.
template <typename T> struct Frag;
template <> struct Frag<bf16_t> {
  bf16x8 v;
  static __device__ __forceinline__ Frag ld(const char* p) { Frag f; f.v = *reinterpret_cast<const bf16x8*>(p); return f; }
};
template <> struct Frag<float> {
  float4 lo, hi;
  static __device__ __forceinline__ Frag ld(const char* p) {
    Frag f;
    f.lo = *reinterpret_cast<const float4*>(p);
    f.hi = *reinterpret_cast<const float4*>(p + 16);
    return f;
  }
};
__device__ __forceinline__ void mma(f32x4& acc, const Frag<bf16_t>& w, const Frag<bf16_t>& a) { acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.v, a.v, acc, 0, 0, 0); }
__device__ __forceinline__ void mma(f32x4& acc, const Frag<float>& w, const Frag<float>& a) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.lo.x, a.lo.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.lo.y, a.lo.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.lo.z, a.lo.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.lo.w, a.lo.w, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.hi.x, a.hi.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.hi.y, a.hi.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.hi.z, a.hi.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.hi.w, a.hi.w, acc, 0, 0, 0);
}

// Tile descriptors are read through the constant address space: with a wave-uniform address that is a scalar load (s_load_dwordx16)
// into SGPRs.  Through a plain global pointer hipcc emitted VECTOR loads + `s_waitcnt vmcnt(0)` + v_readfirstlane for every
// descriptor (it cannot prove that the kernel's own stores leave the table alone), and that wait drained the whole LDS-DMA /
// store queue twice per tile (tools/prof_phases.sh: ~2000 of ~7800 cycles per stage on the HBM-bound layers).
__device__ __forceinline__ TileDesc load_tile(const TileDesc* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef __attribute__((address_space(4))) const long long c64_t;
  typedef __attribute__((address_space(4))) const int c32_t;
  c64_t* q = reinterpret_cast<c64_t*>(reinterpret_cast<unsigned long long>(p));
  c32_t* r = reinterpret_cast<c32_t*>(reinterpret_cast<unsigned long long>(p) + 16);
  TileDesc t;
  t.in_vox = q[0]; t.out_vox = q[1];
  t.g0[0] = r[0]; t.g0[1] = r[1]; t.g0[2] = r[2];
  t.q0[0] = r[3]; t.q0[1] = r[4]; t.q0[2] = r[5];
  t.n = r[6]; t.flags = r[7];
  return t;
#else
  return *p;
#endif
}

typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;
// 16-byte LDS-DMA: LDS address = wave-uniform `lds_wave_base` + lane*16, global address per lane
__device__ __forceinline__ void dma16(const void* gsrc, char* lds_wave_base) {
  vsseg_dma16(gsrc, lds_wave_base);  // inline assembly: see common.h (the builtin made hipcc drain the DMA queue in front of every K loop)
}

// s_waitcnt vmcnt(n) with a run-time (wave-uniform) n: the immediate must be a literal.  Waiting for a SMALLER count than
// necessary is always safe (it only waits longer), so n is clamped to the largest literal provided.
__device__ __forceinline__ void wait_vmcnt(int n) {
#define VSSEG_VM_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
  switch (n < 0 ? 0 : (n > 40 ? 40 : n)) {
    VSSEG_VM_CASE(0) VSSEG_VM_CASE(1) VSSEG_VM_CASE(2) VSSEG_VM_CASE(3) VSSEG_VM_CASE(4) VSSEG_VM_CASE(5) VSSEG_VM_CASE(6) VSSEG_VM_CASE(7) VSSEG_VM_CASE(8) VSSEG_VM_CASE(9)
    VSSEG_VM_CASE(10) VSSEG_VM_CASE(11) VSSEG_VM_CASE(12) VSSEG_VM_CASE(13) VSSEG_VM_CASE(14) VSSEG_VM_CASE(15) VSSEG_VM_CASE(16) VSSEG_VM_CASE(17) VSSEG_VM_CASE(18) VSSEG_VM_CASE(19)
    VSSEG_VM_CASE(20) VSSEG_VM_CASE(21) VSSEG_VM_CASE(22) VSSEG_VM_CASE(23) VSSEG_VM_CASE(24) VSSEG_VM_CASE(25) VSSEG_VM_CASE(26) VSSEG_VM_CASE(27) VSSEG_VM_CASE(28) VSSEG_VM_CASE(29)
    VSSEG_VM_CASE(30) VSSEG_VM_CASE(31) VSSEG_VM_CASE(32) VSSEG_VM_CASE(33) VSSEG_VM_CASE(34) VSSEG_VM_CASE(35) VSSEG_VM_CASE(36) VSSEG_VM_CASE(37) VSSEG_VM_CASE(38) VSSEG_VM_CASE(39)
    VSSEG_VM_CASE(40)
  }
#undef VSSEG_VM_CASE
}

// MODE specialises the epilogue so that registers are spent only on what a launch uses (the generic kernel held the BatchNorm
// statistics accumulators, the auxiliary-tile DMA table and the slow-path coordinate tables of every variant live at once:
// 189-256 VGPRs, spills in <2,4>): 0 plain, 1 +BatchNorm statistics (training forward), 2 +auxiliary tile (residual add /
// gradient accumulation / ReLU mask fetched by DMA).
enum { IG_PLAIN = 0, IG_STATS = 1, IG_AUX = 2 };
#ifdef VSSEG_IG_PROF
#define IG_TICK(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); prof_acc[i] += t_ - prof_t; prof_t = t_; }
#else
#define IG_TICK(i)
#endif
// KS > 0: the number of K-steps per chunk is a compile-time constant (the MFMA-bound configurations: 27 taps x 8 / 16 channels): the K loop
// is fully unrolled, every LDS offset of the weight fragments and of the tap table becomes an instruction immediate and the address /
// loop-counter arithmetic that cost ~6 VALU + SALU issues per MFMA in the run-time loop disappears.
// NT >= 3 (the MFMA-bound configurations) run with 8 waves: waves 0-3 are the consumers (fragments + MFMAs + epilogue, exactly the 4-wave
// tile mapping of the other configurations), waves 4-7 are producers that only issue the LDS-DMA of the stages ahead and wait for it.
// One consumer and one producer share each SIMD, so the ~100-200 issue cycles of every DMA instruction, its address arithmetic and
// the DMA latency never sit in the consumer's instruction stream, and the consumer's K loop is one straight block of ds_read / MFMA.
constexpr bool ig_spec(int nt) { return nt >= 3; }
template <typename T, int NT, int MTW, int MODE, int KS = 0>
__global__ __launch_bounds__(ig_spec(NT) ? 512 : 256, NT == 1 ? 4 : (NT == 2 ? (MODE == IG_PLAIN ? 3 : 2) : 2)) void igemm_kernel(const IgemmK k) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool STATS = MODE == IG_STATS, AUXM = MODE == IG_AUX;
  constexpr int ES = sizeof(T);
  constexpr int GB = 8 * ES;   // bytes of one 8-channel group
  constexpr int EPP = 16 / ES; // elements per 16-byte piece
  const vsseg_igemm_desc& d = k.d;
  int* ktab = reinterpret_cast<int*>(smem + k.lds_ktab);
  float* epi = reinterpret_cast<float*>(smem + k.lds_epi);  // bias | scale | shift, NT*16 each
  char* Wl = smem + k.lds_w;
  char* Hl = smem + k.lds_h;
  char* Al = smem + k.lds_aux;
  unsigned* pinfo_l = reinterpret_cast<unsigned*>(smem + k.lds_pinfo);  // [u][256] packed halo coordinates (hx | hy<<8 | hz<<16 | 16-byte piece of the voxel row <<24) of this thread's DMA pieces
  unsigned* vxyz_l = pinfo_l + k.npu * 256;                             // [64*MTW] packed tile coordinates of the tile's voxels (partial tiles only): VROWS rows of 256
  constexpr int VROWS = (64 * MTW + 255) / 256;
  constexpr int AM = amax_for(MTW);
  // MFMA-bound configurations: the halo coordinates of a DMA piece are recomputed from its index (three multiply-high divisions, ~12 VALU)
  // instead of being read from a per-thread LDS table: no table (8-16 KiB of LDS that the second weight / halo buffer needs), and no LDS
  // read — hence no `s_waitcnt lgkmcnt(0)` that would drain the fragment prefetch — when pieces are issued between the MFMAs of the K loop.
  constexpr bool ARITH = NT >= 3;

  constexpr bool SPEC = ig_spec(NT);
  // `wave` is the index inside the role: consumers 0-3 own voxel tiles (and, without specialisation, DMA pieces), producers 0-3 own DMA pieces
  const int tid = threadIdx.x & 255, lane = tid & 63, raw_wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), wave = raw_wave & 3, g = lane >> 4, l15 = lane & 15;
  const bool producer = SPEC && raw_wave >= 4;
  const int split = blockIdx.y;
  const int HY = k.halo[1], HZ = k.halo[2], CK = d.ck;
  const int hvox = k.halo[0] * HY * HZ;
  const int ppv = CK / EPP;  // 16-byte pieces per halo voxel
  const int pieces = hvox * ppv;
  const int wpieces = k.w_bytes >> 4;
  const int cout = d.out.c;
  const int nch = d.nchunks;
  // class_split: workgroup row `split` is an output-parity class — all output channels (channel base 0), its own taps and K-step count, its own
  // output voxel offset; otherwise `split` is a slice of the output channels
  const bool csplit = d.class_split > 0;
  const int cbase = csplit ? 0 : split * NT * 16;
  const int64_t cvox = csplit ? k.class_vox[split] : 0;
  const int my_ntaps = csplit ? d.class_ntaps[split] : d.ntaps;
  const int my_ks = csplit ? (my_ntaps * k.cgs + 3) / 4 : d.ksteps;

  // ---- per-workgroup tables ----
  for (int i = tid; i < my_ks * 4; i += 256) {
    int off = 0;
    if (i < my_ntaps * k.cgs) {
      int t = i / k.cgs, cg = i - t * k.cgs;
      if (csplit) t = d.class_tap[split][t];
      int hx = d.tap_off[t][0] - k.off_min[0], hy = d.tap_off[t][1] - k.off_min[1], hz = d.tap_off[t][2] - k.off_min[2];
      off = ((hx * HY + hy) * HZ + hz) * CK * ES + cg * GB;
    }
    ktab[i] = off;
  }
  for (int i = tid; i < NT * 16; i += 256) {
    const int c = cbase + i;
    const bool ok = c < cout;
    const int cv = d.cout_mod > 0 ? c % d.cout_mod : c;  // z-folded launches: 8 z-neighbours x cout real channels share the per-channel vectors
    epi[i] = ((ok && d.bias) ? d.bias[cv] : 0.f) + ((ok && d.bias2) ? d.bias2[cv] : 0.f);
    epi[NT * 16 + i] = (ok && d.scale) ? d.scale[cv] : 1.f;
    epi[2 * NT * 16 + i] = (ok && d.scale) ? d.shift[cv] : 0.f;
  }
  if (nch == 1) {  // packed weights stay resident for the whole kernel
    const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(d.wpack) + (int64_t)split * k.w_bytes);
    uint4* dst = reinterpret_cast<uint4*>(Wl);
    for (int i = tid; i < wpieces; i += 256) dst[i] = src[i];
  }
  const float alpha = (d.act == VSSEG_ACT_PRELU && d.alpha) ? *d.alpha : 0.f;

  // ---- per-thread constants: the halo pieces this thread fetches every stage, and the voxels it stores.
  //      Everything that does not depend on the tile is computed once here: the per-tile work is then a handful of
  //      32-bit adds (the unoptimised version spent ~1100 VALU+SALU instructions per tile and wave on index arithmetic
  //      and was issue-bound at 1.7 TB/s on the HBM-bound layers).
  const int X = d.in.x, Y = d.in.y, Z = d.in.z;
  const unsigned in_vox_bytes = (unsigned)d.in.pitch * ES;
  const bool in_two = d.in.ptr2 != nullptr;
  const int in_csplit = in_two ? d.in.csplit : 0x7fffffff;
  unsigned* prel_l = pinfo_l + (k.npu + VROWS) * 256;  // [npu][256] interior-tile byte offsets of the same pieces (not ARITH)
  if (ARITH) vxyz_l = pinfo_l;  // no piece tables: the voxel coordinates are the only rows
#pragma unroll
  for (int u = 0; u < PMAX; ++u) {
    const int j = (u * 4 + wave) * 64 + lane;
    unsigned info = 0xffffffffu;  // no piece: the lane copies 16 harmless bytes into the padding of the LDS buffer
    if (j < pieces) {
      int hv = j / ppv, c16 = j - hv * ppv;
      int hz = hv % HZ, r = hv / HZ;
      int hy = r % HY, hx = r / HY;
      info = (unsigned)hx | ((unsigned)hy << 8) | ((unsigned)hz << 16) | ((unsigned)c16 << 24);
    }
    if (!ARITH && u < k.npu) {
      pinfo_l[u * 256 + tid] = info;
      // interior tiles: byte offset of the piece relative to the halo origin voxel (0 for padding lanes: a valid, harmless source)
      prel_l[u * 256 + tid] = info == 0xffffffffu ? 0u : (unsigned)((int)(info & 255u) * Y + (int)((info >> 8) & 255u)) * (unsigned)Z * in_vox_bytes + ((info >> 16) & 255u) * in_vox_bytes + (info >> 24) * 16u;
    }
  }
  // Two-part input (skip-connection concat): channels >= csplit come from in.ptr2.  pinfo keeps the channel offset of the
  // virtual concatenated row; part 1's base is biased by -csplit channels so the same offsets address it.  A chunk never
  // straddles the split unless it is the only chunk (checked on the host): then the part is chosen per piece.
  const int OX = d.out.x, OY = d.out.y, OZ = d.out.z;
  int vb[MTW];
  unsigned ovrel[MTW];  // output voxel index relative to the tile's first output voxel
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
    const int v = (wave * MTW + m) * 16 + l15;
    int vz = v % d.tile[2], r = v / d.tile[2];
    int vy = r % d.tile[1], vx = r / d.tile[1];
    vb[m] = (((vx * d.is[0]) * HY + vy * d.is[1]) * HZ + vz * d.is[2]) * CK * ES;
    if (g == 0) vxyz_l[v] = (unsigned)vx | ((unsigned)vy << 8) | ((unsigned)vz << 16);
    ovrel[m] = (unsigned)((vx * d.os[0] * OY + vy * d.os[1]) * OZ + vz * d.os[2]);
  }
  const unsigned out_es = d.out.dtype == VSSEG_F32 ? 4u : 2u;
  const unsigned out_vox_bytes = (unsigned)d.out.pitch * out_es;
  // auxiliary tile (old output for accumulate / residual / ReLU mask): [tile voxel][NT*16] fetched by DMA one tile ahead
  constexpr bool aux_on = AUXM;  // the host picks MODE == IG_AUX exactly when the auxiliary tile is DMA-eligible (aux_bytes > 0)
  const unsigned aux_es = k.aux.dtype == VSSEG_F32 ? 4u : 2u;
  const unsigned aux_vox_bytes = (unsigned)k.aux.pitch * aux_es;
  const int aux_row = NT * 16 * (int)aux_es;  // LDS bytes per voxel
  const int ppa = aux_row >> 4;
  const int apieces = aux_on ? 64 * MTW * ppa : 0;
  unsigned arel[AUXM ? AM : 1];
#pragma unroll
  for (int u = 0; u < (AUXM ? AM : 1); ++u) {
    const int j = (u * 4 + wave) * 64 + lane;
    unsigned rel = 0xffffffffu;
    if (j < apieces) {
      const int v = j / ppa, c16 = j - v * ppa;
      int vz = v % d.tile[2], r = v / d.tile[2];
      int vy = r % d.tile[1], vx = r / d.tile[1];
      rel = (unsigned)((vx * d.os[0] * OY + vy * d.os[1]) * OZ + vz * d.os[2]) * aux_vox_bytes + (unsigned)cbase * aux_es + (unsigned)c16 * 16u;
    }
    arel[u] = rel;
  }
  // two-part auxiliary / output tensors: same scheme (static per-piece part choice, part-1 base biased by -csplit channels)
  const bool aux_two = aux_on && k.aux.ptr2 != nullptr;
  unsigned a2mask = 0;
  if (aux_two) {
#pragma unroll
    for (int u = 0; u < (AUXM ? AM : 1); ++u) {
      const int j = (u * 4 + wave) * 64 + lane;
      if (j < apieces && cbase + (j % ppa) * (16 / (int)aux_es) >= k.aux.csplit) a2mask |= 1u << u;
    }
  }
  const char* aux_base = reinterpret_cast<const char*>(k.aux.ptr);
  const char* aux_base1 = aux_two ? reinterpret_cast<const char*>(k.aux.ptr2) - (int64_t)k.aux.csplit * aux_es : aux_base;
  const bool out_two = d.out.ptr2 != nullptr;
  const int out_csplit = out_two ? d.out.csplit : 0x7fffffff;
  char* out_base = reinterpret_cast<char*>(d.out.ptr);
  char* out_base1 = out_two ? reinterpret_cast<char*>(d.out.ptr2) - (int64_t)out_csplit * out_es : out_base;
  const bool res_two = d.res_mode != VSSEG_RES_NONE && d.res.ptr2 != nullptr;
  const int res_csplit = res_two ? d.res.csplit : 0x7fffffff;
  // gated add (aux_mode 4): the tile's attention values ride in the aux buffer too — one 16-byte piece per 4 z-consecutive voxels,
  // fetched by wave 0 (16*MTW pieces); an ordinary load in the epilogue would make hipcc drain the DMA queue
  const bool gate_dma = AUXM && k.aux_gate_off > 0;
  unsigned grel = 0;  // voxel offset (relative to the tile's first output voxel) of this lane's gate piece
  if (gate_dma && wave == 0 && lane < 16 * MTW) {
    const int v = lane * 4;
    int vz = v % d.tile[2], r = v / d.tile[2];
    int vy = r % d.tile[1], vx = r / d.tile[1];
    grel = (unsigned)((vx * OY + vy) * OZ + vz);
  }
  const bool fast_store = (d.res_mode == VSSEG_RES_NONE && !d.accumulate) || aux_on;
  const bool vec_store = (d.out.pitch & 3) == 0 && (cout & 3) == 0;

  // ---- tile schedule: XCD x (= blockIdx % 8) owns tiles [x*tpx, (x+1)*tpx); its workgroups stride through them ----
  const int G = gridDim.x;
  const bool xcd_map = (G % 8) == 0;
  const int tpx = xcd_map ? (int)((k.total_tiles + 7) / 8) : (int)k.total_tiles;
  const int xcd = xcd_map ? (int)(blockIdx.x & 7) : 0;
  const int slot = xcd_map ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int S = xcd_map ? G / 8 : G;
  const int t_first = xcd * tpx;
  int cnt = (int)k.total_tiles - t_first;
  if (cnt > tpx) cnt = tpx;
  const int my_tiles = cnt > slot ? (cnt - 1 - slot) / S + 1 : 0;
  const int nstages = my_tiles * nch;

  const char* in_base = reinterpret_cast<const char*>(d.in.ptr);
  const char* in_base1 = in_two ? reinterpret_cast<const char*>(d.in.ptr2) - (int64_t)in_csplit * ES : in_base;
  const int64_t in_sample_bytes = (int64_t)X * Y * Z * in_vox_bytes;

  const TileDesc* tiles = k.tiles + (t_first + slot);  // this workgroup's tiles: tiles[i * S]
  // descriptors are fetched (scalar loads) one tile ahead of their first use, so their latency never sits in front of a DMA issue
  TileDesc td_next = load_tile(tiles);  // descriptor of the next tile to be issued
  int ti_issue = 0, ti_cur = 0;
  int ch_issue = 0, ch_cur = 0;
  const int D = k.depth, nbuf = D + 1;
  int buf_issue = 0, buf_cur = 0, abuf_issue = 0, abuf_cur = 0;  // ring positions (stage ring / per-tile auxiliary ring)
  // VMEM instructions this wave issues per stage: the counted waits below rely on these being exact and wave-uniform
  int nh = 0, na = 0, nw = 0;
#pragma unroll
  for (int u = 0; u < PMAX; ++u) nh += ((u * 4 + wave) * 64 < pieces) ? 1 : 0;
#pragma unroll
  for (int u = 0; u < (AUXM ? AM : 0); ++u) na += ((u * 4 + wave) * 64 < apieces) ? 1 : 0;
  if (gate_dma && wave == 0) ++na;
  if (nch > 1)
    for (int j0 = wave * 64; j0 < wpieces; j0 += 256) ++nw;
  if (SPEC && !producer) nh = na = nw = 0;  // consumers issue no DMA
  // store instructions a wave issues in the fast epilogue of one tile (exactly one 8/16-byte store per valid 16-channel
  // block and M tile; the scalar-store variant and the slow epilogue count as 0 = their stores are simply waited for)
  int nst_fast = 0;
  if (vec_store && !producer) {
#pragma unroll
    for (int t = 0; t < NT; ++t) nst_fast += (cbase + t * 16 < cout) ? MTW : 0;
  }
  int st_h0 = 0, st_h1 = 0, st_h2 = 0;
  // ---- LDS-DMA of one stage (halo chunk, auxiliary tile, weight chunk when the weights are not resident) ----
  // issue_setup() fixes the stage's wave-uniform addresses; halo_piece() / weight_piece() issue one 1 KiB instruction each; issue() =
  // all of a stage.  In the specialised kernels only the producer waves call it (tools/prof_phases.sh measured 2 800 of 7 900 cycles per
  // stage spent issuing DMAs in front of the K loop when every wave did both jobs).
  const char *pi_org = nullptr, *pi_org1 = nullptr, *pi_wsrc = nullptr;
  char *pi_hdst = nullptr, *pi_wdst = nullptr;
  int pi_g0x = 0, pi_g0y = 0, pi_g0z = 0, pi_c0 = 0;
  bool pi_interior = false;
  const bool piece_select = in_two && nch == 1;  // the only chunk spans both parts of a two-part input: the part is chosen per piece
  auto issue_setup = [&]() {
    const int ch = ch_issue;
    const TileDesc td = td_next;
    if (++ch_issue == nch) {
      ch_issue = 0;
      ++ti_issue;
      if (ti_issue < my_tiles) td_next = load_tile(tiles + (int64_t)ti_issue * S);  // uniform address: scalar loads, consumed a whole stage later
    }
    const int c0 = ch * CK;
    pi_c0 = c0;
    pi_hdst = Hl + buf_issue * k.h_stride;
    pi_interior = (td.flags & 1) && c0 + CK <= d.in.c;
    if (pi_interior) {  // every source is origin + a per-thread constant derived from the packed halo coordinates
      const int64_t ooff = td.in_vox * in_vox_bytes + (int64_t)c0 * ES;
      pi_org = in_base + ooff;
      pi_org1 = in_base1 + ooff;
      if (in_two && nch > 1 && c0 >= in_csplit) pi_org = pi_org1;  // the whole chunk lies in part 1
    } else {
      pi_g0x = td.g0[0]; pi_g0y = td.g0[1]; pi_g0z = td.g0[2];
      const int64_t soff = (int64_t)td.n * in_sample_bytes + (int64_t)c0 * ES;
      pi_org = in_base + soff;
      pi_org1 = in_base1 + soff;
    }
    if constexpr (AUXM) if (ch == 0) {
      // partial tiles use the slow epilogue (ordinary loads) but still issue the same number of DMAs (from the zero page),
      // so that the per-stage instruction count the vmcnt arithmetic relies on stays exact
      const bool whole = (td.flags & 2) != 0;
      const char* aorigin = aux_base + (td.out_vox + cvox) * aux_vox_bytes;
      const char* aorigin1 = aux_base1 + (td.out_vox + cvox) * aux_vox_bytes;
      char* Adst = Al + abuf_issue * k.aux_stride;
#pragma unroll
      for (int u = 0; u < (AUXM ? AM : 0); ++u) {
        if ((u * 4 + wave) * 64 >= apieces) break;
        dma16(whole && arel[u] != 0xffffffffu ? (const void*)(((a2mask >> u) & 1u ? aorigin1 : aorigin) + arel[u]) : k.zeros, Adst + (u * 4 + wave) * 1024);  // padding lanes: zeros into the padding
      }
      if (gate_dma && wave == 0) {
        if (lane < 16 * MTW) dma16(whole ? (const void*)(d.gate + td.out_vox + cvox + grel) : k.zeros, Adst + k.aux_gate_off);
      }
      if (++abuf_issue == nbuf) abuf_issue = 0;
    }
    if (nch > 1) {
      pi_wsrc = reinterpret_cast<const char*>(d.wpack) + ((int64_t)split * nch + ch) * k.w_bytes;
      pi_wdst = Wl + buf_issue * k.w_bytes;
    }
    if (++buf_issue == nbuf) buf_issue = 0;
  };
  // `word` = the piece's entry of the table the stage uses: the byte offset (interior tiles) or the packed coordinates (boundary tiles)
  auto halo_piece = [&](int u, unsigned word) {  // every lane issues: lanes beyond the last piece copy 16 harmless bytes into the buffer's padding
    char* dst = pi_hdst + (u * 4 + wave) * 1024;
    if constexpr (ARITH) {  // word is ignored: packed coordinates from the piece index
      const unsigned j = (unsigned)((u * 4 + wave) * 64 + lane);
      auto divm = [](unsigned n, unsigned mg) { return mg ? __umulhi(n, mg) : n; };  // mg == 0: divisor 1
      const unsigned hv = divm(j, k.mg_ppv), c16 = j - hv * (unsigned)ppv;
      const unsigned hx = divm(hv, k.mg_hyz), r = hv - hx * (unsigned)(HY * HZ);
      const unsigned hy = divm(r, k.mg_hz), hz = r - hy * (unsigned)HZ;
      word = (int)j < pieces ? (hx | (hy << 8) | (hz << 16) | (c16 << 24)) : 0xffffffffu;
      if (pi_interior && !piece_select) {
        dma16(pi_org + (word == 0xffffffffu ? 0u : ((hx * (unsigned)Y + hy) * (unsigned)Z + hz) * in_vox_bytes + c16 * 16u), dst);
        return;
      }
    }
    if (pi_interior && !piece_select) {
      dma16(pi_org + word, dst);  // scalar base + 32-bit per-lane byte offset
    } else if (pi_interior) {  // the only chunk spans both parts of a two-part input: the part is chosen per piece from its packed coordinates
      const unsigned c16 = word >> 24;
      const unsigned rel = word == 0xffffffffu ? 0u : (((word & 255u) * (unsigned)Y + ((word >> 8) & 255u)) * (unsigned)Z + ((word >> 16) & 255u)) * in_vox_bytes + c16 * 16u;
      dma16(((int)(c16 * EPP) >= in_csplit && word != 0xffffffffu ? pi_org1 : pi_org) + rel, dst);
    } else {
      const unsigned hx = word & 255u, hy = (word >> 8) & 255u, hz = (word >> 16) & 255u, c16 = word >> 24;
      const int gx = pi_g0x + (int)hx, gy = pi_g0y + (int)hy, gz = pi_g0z + (int)hz;
      const int c = pi_c0 + (int)c16 * EPP;
      const bool ok = word != 0xffffffffu && (unsigned)gx < (unsigned)X && (unsigned)gy < (unsigned)Y && (unsigned)gz < (unsigned)Z && c + EPP <= d.in.c;
      const void* src = ok ? (const void*)((c >= in_csplit ? pi_org1 : pi_org) + (int64_t)((gx * Y + gy) * Z + gz) * in_vox_bytes + c16 * 16u) : k.zeros;
      dma16(src, dst);
    }
  };
  auto weight_piece = [&](int i) {  // w_bytes is a multiple of 1 KiB: no partial instruction
    const int j0 = wave * 64 + i * 256;
    dma16(pi_wsrc + (int64_t)(j0 + lane) * 16, pi_wdst + j0 * 16);
  };
  auto issue = [&]() {  // the whole stage at once; the packed coordinates of piece u+1 are read from the LDS table while piece u is issued
    issue_setup();
    if constexpr (ARITH) {
      for (int u = 0; u < nh; ++u) halo_piece(u, 0u);
    } else {
      const unsigned* tab = (pi_interior && !piece_select) ? prel_l : pinfo_l;
      unsigned word = nh > 0 ? tab[tid] : 0u;
      for (int u = 0; u < nh; ++u) {  // wave-uniform trip count
        const unsigned nxt = tab[(u + 1 < nh ? u + 1 : u) * 256 + tid];
        halo_piece(u, word);
        word = nxt;
      }
    }
    for (int i = 0; i < nw; ++i) weight_piece(i);
  };
  f32x4 acc[MTW][NT];
  float ssum[STATS ? NT : 1][4], ssq[STATS ? NT : 1][4];
#pragma unroll
  for (int t = 0; t < (STATS ? NT : 1); ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) ssum[t][r] = ssq[t][r] = 0.f;

  __syncthreads();  // tables / resident weights / epilogue constants written above are visible before the pipeline starts
#ifdef VSSEG_IG_PROF
  unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prof_t = __builtin_amdgcn_s_memtime();
#endif
  if (!SPEC || producer)
    for (int j = 0; j < D && j < nstages; ++j) issue();  // stages 0..D-1 in flight
  for (int s = 0; s < nstages; ++s) {
    IG_TICK(5)
    // Stage s has landed once at most the DMAs of the younger stages s+1..s+D-1 remain (VMEM ops complete in issue order).  The
    // epilogue stores issued after those DMAs are ignored in the count, which only makes the wait conservative.
    // Also younger than stage s's DMAs, and still allowed to be in flight: the output stores of the last D stages (vmcnt retires
    // loads and stores of a wave in issue order).  Waiting for them too would put a write-acknowledge latency into every stage.
    if (SPEC && !producer) {
      // consumer: nothing of its own to wait for — the producers wait for the stage's DMA before the barrier below
    } else if (D == 0) {
      // No prefetch, one LDS buffer: the workgroup is half the LDS size, so twice as many share a CU and overlap each other's
      // load and compute phases instead (what the streaming kernels do with plain occupancy).
      if (s > 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // every wave is done reading the previous stage's tile
      }
      IG_TICK(0)
      issue();
      IG_TICK(1)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      IG_TICK(2)
    } else if (D == 1) {
      wait_vmcnt(st_h0);
    } else {
      int younger = st_h0 + st_h1 + (D >= 3 ? st_h2 : 0), chj = ch_cur;
      for (int j = 1; j < D && s + j < nstages; ++j) {
        if (++chj == nch) chj = 0;
        younger += nh + nw + (chj == 0 ? na : 0);
      }
      wait_vmcnt(younger);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // raw barrier: __syncthreads() would drain the DMA queue (vmcnt(0)) and serialise the ring
    IG_TICK(3)
    if ((!SPEC || producer) && D > 0 && s + D < nstages) issue();  // refill the ring slot everybody finished reading before this barrier (stage s-1's)
    if (producer) {  // on to the next stage's wait (one barrier per stage in both roles)
      if (++ch_cur == nch) ch_cur = 0;
      continue;
    }
    const TileDesc tc = load_tile(tiles + (int64_t)ti_cur * S);  // this stage's tile (scalar load, consumed in the epilogue)
    const int ch = ch_cur;
    if (ch == 0) {
#pragma unroll
      for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const char* Hs = Hl + buf_cur * k.h_stride;
    const char* Ws = nch > 1 ? Wl + buf_cur * k.w_bytes : Wl;
    if (++buf_cur == nbuf) buf_cur = 0;
    if constexpr (NT <= 2) {
      // HBM-bound configurations (<= 32 output channels per workgroup): plain K loop — two resident workgroups per CU hide the
      // LDS latency, and the registers of a second fragment set would cost that second workgroup
      const char* Wlane = Ws + lane * GB;
      const int nks = my_ks;
      int koff_next = ktab[g];
      for (int ks = 0; ks < nks; ++ks) {
        const int koff = koff_next;
        koff_next = ktab[(ks + 1 < nks ? ks + 1 : ks) * 4 + g];  // the next step's tap offset is read while this step's fragments / MFMAs are in flight
        Frag<T> w[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) w[t] = Frag<T>::ld(Wlane + (ks * NT + t) * 64 * GB);
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
          Frag<T> a = Frag<T>::ld(Hs + vb[m] + koff);
#pragma unroll
          for (int t = 0; t < NT; ++t) mma(acc[m][t], w[t], a);
        }
      }
    } else
    {  // K loop of the MFMA-bound configurations.  ONE set of voxel (A) fragments, rotated in place: right after the NT MFMAs that use
       // a[m] its registers are reloaded with the next K-step's fragment, whose LDS latency is then covered by the MFMAs of the other
       // MTW-1 voxel tiles; the weight (W) fragments are used by every voxel tile of a step, so those alternate between two sets.
       // (Two full A sets cost 4*MTW more VGPRs — MTW 8 spilled — and hipcc collapsed them into one quad anyway.)
      Frag<T> wa[NT], wb[NT], a[MTW];
      const char* Wlane = Ws + lane * GB;
      const int nks = KS > 0 ? KS : my_ks;
      int ko1 = ktab[(nks > 1 ? 1 : 0) * 4 + g], ko2 = ktab[(nks > 2 ? 2 : 0) * 4 + g];  // tap offsets a whole step ahead of the reads that need them
      {
        const int koff = ktab[g];
#pragma unroll
        for (int t = 0; t < NT; ++t) wa[t] = Frag<T>::ld(Wlane + t * 64 * GB);
#pragma unroll
        for (int m = 0; m < MTW; ++m) a[m] = Frag<T>::ld(Hs + vb[m] + koff);
      }
      // one K-step: MFMAs with the current W set `wc`, reloading a[m] (offset `kon`) and filling the other W set `wn` for step ks+1
      auto kstep = [&](const Frag<T>(&wc)[NT], Frag<T>(&wn)[NT], int ks, int kon) {
        const bool more = ks + 1 < nks;
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
          if (more && m < NT) wn[m] = Frag<T>::ld(Wlane + ((ks + 1) * NT + m) * 64 * GB);  // the next step's weights, one fragment per voxel tile (MTW >= NT is not required: see below)
#pragma unroll
          for (int t = 0; t < NT; ++t) mma(acc[m][t], wc[t], a[m]);
          if (more) a[m] = Frag<T>::ld(Hs + vb[m] + kon);
#ifdef VSSEG_IG_SGB
          if constexpr (sizeof(T) == 2) {  // pin the interleave
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
#endif
        }
        if (more && MTW < NT) {
#pragma unroll
          for (int t = MTW; t < NT; ++t) wn[t] = Frag<T>::ld(Wlane + ((ks + 1) * NT + t) * 64 * GB);
        }
      };
#pragma unroll(KS > 0 ? 16 : 1)
      for (int ks = 0; ks < nks; ks += 2) {
        const int ko1n = ktab[(ks + 3 < nks ? ks + 3 : 0) * 4 + g], ko2n = ktab[(ks + 4 < nks ? ks + 4 : 0) * 4 + g];
        kstep(wa, wb, ks, ko1);
        if (ks + 1 < nks) kstep(wb, wa, ks + 1, ko2);
        ko1 = ko1n; ko2 = ko2n;
      }
    }
    st_h2 = st_h1; st_h1 = st_h0; st_h0 = 0;  // store instructions of the last three stages (this stage's are added below)
    IG_TICK(4)
    if (++ch_cur != nch) continue;
    ch_cur = 0;

    // ---- epilogue of the current tile ----
    ++ti_cur;
    const bool whole = (tc.flags & 2) != 0;
    char* out_tile = out_base + (tc.out_vox + cvox) * out_vox_bytes;
    const int64_t out_delta = out_base1 - out_base;  // 0 for an ordinary tensor
    const char* Aux = Al + abuf_cur * k.aux_stride;
    if (++abuf_cur == nbuf) abuf_cur = 0;
    if (whole && fast_store) {  // interior tile, plain store: bias (+stats) (+affine) + activation, 4 channels per lane
      st_h0 = nst_fast;
      float gate[MTW];  // aux_mode 4: 1 + attention value of this lane's voxels (one ordinary load each, issued together)
      if constexpr (AUXM) {
        if (k.aux_mode == 4) {
#pragma unroll
          for (int m = 0; m < MTW; ++m)
            gate[m] = 1.f + (gate_dma ? reinterpret_cast<const float*>(Aux + k.aux_gate_off)[(wave * MTW + m) * 16 + l15] : d.gate[tc.out_vox + cvox + ovrel[m]]);
        }
      }
#pragma unroll
      for (int m = 0; m < MTW; ++m) {
        char* op = out_tile + ovrel[m] * out_vox_bytes;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int cl = t * 16 + g * 4;
          const int c = cbase + cl;
          if (c >= cout) continue;
          const float4 bi = *reinterpret_cast<const float4*>(epi + cl);
          float val[4] = {acc[m][t][0] + bi.x, acc[m][t][1] + bi.y, acc[m][t][2] + bi.z, acc[m][t][3] + bi.w};
          if constexpr (STATS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { ssum[t][r] += val[r]; ssq[t][r] += val[r] * val[r]; }
          }
          if (d.scale) {
            const float4 sc = *reinterpret_cast<const float4*>(epi + NT * 16 + cl), sh = *reinterpret_cast<const float4*>(epi + 2 * NT * 16 + cl);
            val[0] = val[0] * sc.x + sh.x; val[1] = val[1] * sc.y + sh.y; val[2] = val[2] * sc.z + sh.z; val[3] = val[3] * sc.w + sh.w;
          }
          if (d.act == VSSEG_ACT_PRELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) val[r] = val[r] > 0.f ? val[r] : alpha * val[r];
          } else if (d.act == VSSEG_ACT_RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) val[r] = fmaxf(val[r], 0.f);
          } else if (d.act == VSSEG_ACT_SIGMOID) {
#pragma unroll
            for (int r = 0; r < 4; ++r) val[r] = 1.f / (1.f + __expf(-val[r]));
          }
          if constexpr (AUXM) {
            const char* ap = Aux + ((wave * MTW + m) * 16 + l15) * aux_row + cl * (int)aux_es;
            const float4 av = aux_es == 4 ? *reinterpret_cast<const float4*>(ap) : ld4(reinterpret_cast<const bf16_t*>(ap));
            if (k.aux_mode == 3) {
              val[0] = av.x > 0.f ? val[0] : 0.f; val[1] = av.y > 0.f ? val[1] : 0.f; val[2] = av.z > 0.f ? val[2] : 0.f; val[3] = av.w > 0.f ? val[3] : 0.f;
            } else if (k.aux_mode == 4) {
              val[0] = vsseg_fma_unpacked(av.x, gate[m], val[0]); val[1] = vsseg_fma_unpacked(av.y, gate[m], val[1]);  // (not v_pk_fma_f32 op_sel: common.h)
              val[2] = vsseg_fma_unpacked(av.z, gate[m], val[2]); val[3] = vsseg_fma_unpacked(av.w, gate[m], val[3]);
            } else {
              val[0] += av.x; val[1] += av.y; val[2] += av.z; val[3] += av.w;
            }
          }
          char* opt = op + (cbase + t * 16 >= out_csplit ? out_delta : 0);  // uniform per 16-channel block
          if (vec_store) {
            if (out_es == 4) st4(reinterpret_cast<float*>(opt) + c, make_float4(val[0], val[1], val[2], val[3]));
            else st4(reinterpret_cast<bf16_t*>(opt) + c, make_float4(val[0], val[1], val[2], val[3]));
          } else {  // 1- and 2-channel outputs (attention map, logits): scalar stores of the valid channels
            const int nc = min(4, cout - c);
            for (int r = 0; r < nc; ++r) {
              if (out_es == 4) reinterpret_cast<float*>(opt)[c + r] = val[r];
              else reinterpret_cast<bf16_t*>(opt)[c + r] = f2bf(val[r]);
            }
          }
        }
      }
    } else {
    const int n = tc.n, q0x = tc.q0[0], q0y = tc.q0[1], q0z = tc.q0[2];
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
      const unsigned vx_ = vxyz_l[(wave * MTW + m) * 16 + l15];
      const int qx = q0x + (int)(vx_ & 255u), qy = q0y + (int)((vx_ >> 8) & 255u), qz = q0z + (int)(vx_ >> 16);
      const int ox = qx * d.os[0] + (csplit ? d.class_oo[split][0] : d.oo[0]), oy = qy * d.os[1] + (csplit ? d.class_oo[split][1] : d.oo[1]), oz = qz * d.os[2] + (csplit ? d.class_oo[split][2] : d.oo[2]);
      const bool vok = qx < d.q[0] && qy < d.q[1] && qz < d.q[2] && ox < OX && oy < OY && oz < OZ;
      const int64_t ovox = (((int64_t)n * OX + ox) * OY + oy) * OZ + oz;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int cl = t * 16 + g * 4;  // channel inside this workgroup's NT*16 slice
        const int c = cbase + cl;
        if (!vok || c >= cout) continue;
        float val[4] = {acc[m][t][0], acc[m][t][1], acc[m][t][2], acc[m][t][3]};
        const int nc = min(4, cout - c);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (r >= nc) break;
          float x = val[r] + epi[cl + r];
          if constexpr (STATS) { ssum[t][r] += x; ssq[t][r] += x * x; }
          x = x * epi[NT * 16 + cl + r] + epi[2 * NT * 16 + cl + r];
          if (d.act == VSSEG_ACT_PRELU) x = x > 0.f ? x : alpha * x;
          else if (d.act == VSSEG_ACT_RELU) x = fmaxf(x, 0.f);
          else if (d.act == VSSEG_ACT_SIGMOID) x = 1.f / (1.f + __expf(-x));
          if (d.res_mode != VSSEG_RES_NONE) {
            const bool r2 = c >= res_csplit;
            const void* rbase = r2 ? d.res.ptr2 : d.res.ptr;
            const int64_t ro = ovox * d.res.pitch + c + r - (r2 ? res_csplit : 0);
            float rv = d.res.dtype == VSSEG_F32 ? reinterpret_cast<const float*>(rbase)[ro] : bf2f(reinterpret_cast<const bf16_t*>(rbase)[ro]);
            x = d.res_mode == VSSEG_RES_ADD ? x + rv : (d.res_mode == VSSEG_RES_GATE ? x + rv * (1.f + d.gate[ovox]) : (rv > 0.f ? x : 0.f));
          }
          val[r] = x;
        }
        const int64_t oo = ovox * d.out.pitch + c;
        char* obase = c >= out_csplit ? out_base1 : out_base;
        if (d.out.dtype == VSSEG_F32) {
          float* op = reinterpret_cast<float*>(obase) + oo;
          if (nc == 4 && (d.out.pitch & 3) == 0 && !d.accumulate) st4(op, make_float4(val[0], val[1], val[2], val[3]));
          else
            for (int r = 0; r < nc; ++r) op[r] = d.accumulate ? op[r] + val[r] : val[r];
        } else {
          bf16_t* op = reinterpret_cast<bf16_t*>(obase) + oo;
          if (nc == 4 && (d.out.pitch & 3) == 0) {
            if (d.accumulate) {
              float4 o = ld4(op);
              val[0] += o.x; val[1] += o.y; val[2] += o.z; val[3] += o.w;
            }
            st4(op, make_float4(val[0], val[1], val[2], val[3]));
          } else
            for (int r = 0; r < nc; ++r) op[r] = f2bf(d.accumulate ? bf2f(op[r]) + val[r] : val[r]);
        }
      }
    }
    // The slow path's ordinary loads (residual / previous gradient / gate of boundary tiles) leave hipcc's scoreboard with "maybe pending"
    // VGPRs at the loop header; it then put `s_waitcnt vmcnt(0)` in front of the K loop of EVERY stage — right behind the next stage's DMA
    // issue, which serialised the ring.  An explicit (modelled) wait here, on the rare path, leaves the merged state clean.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) only
    }  // slow epilogue
  }

#ifdef VSSEG_IG_PROF
  if (k.prof && blockIdx.x == 8 && blockIdx.y == 0 && tid == 0) {
    for (int i = 0; i < 8; ++i) k.prof[i] = prof_acc[i];
    k.prof[8] = (unsigned long long)nstages;
  }
#endif
  if constexpr (STATS) {  // per-channel sum / sum-of-squares of all this workgroup's tiles: shuffle tree -> one LDS row per consumer wave, summed in wave
                          // order -> sharded statistics as fixed-point integer atomics (order-independent: vsseg_fx_add)
    __syncthreads();
    float* red = reinterpret_cast<float*>(Wl);  // [4 consumer waves][2][NT*16]: the packed weights are no longer needed (>= 1 KiB per channel tile)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = ssum[t][r], q = ssq[t][r];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
        if (l15 == 0 && !producer) {
          red[wave * (2 * NT * 16) + t * 16 + g * 4 + r] = s;
          red[wave * (2 * NT * 16) + NT * 16 + t * 16 + g * 4 + r] = q;
        }
      }
    __syncthreads();
    double* st = d.stats + (int64_t)(blockIdx.x % VSSEG_STAT_SHARDS) * 2 * d.stats_stride;
    for (int i = tid + (producer ? 1 << 20 : 0); i < 2 * NT * 16; i += 256) {  // consumers only (the producer half would add the sums a second time)
      int which = i / (NT * 16), cc = i - which * NT * 16;
      int c = cbase + cc;
      const float v = (red[i] + red[2 * NT * 16 + i]) + (red[4 * NT * 16 + i] + red[6 * NT * 16 + i]);
      if (c < cout) vsseg_fx_add(&st[which * d.stats_stride + (d.cout_mod > 0 ? c % d.cout_mod : c)], (double)v, VSSEG_FX_STAT, k.fxflag);
    }
  }
}

template <typename T, int NT, int MTW, int MODE, int KS = 0> static int launch_mode(const IgemmK& k, dim3 grid, int lds, hipStream_t s) {
  if constexpr (KS == 0 && NT >= 3 && NT <= 4 && MODE != IG_AUX) {  // unrolled K loops of the hot MFMA-bound shapes (3x3x3 taps, 8 / 16-channel chunks)
    if (k.d.ksteps == 14 && !k.d.class_split) return launch_mode<T, NT, MTW, MODE, 14>(k, grid, lds, s);  // (class_split: the K-step count differs per workgroup row)
    if (k.d.ksteps == 7 && !k.d.class_split) return launch_mode<T, NT, MTW, MODE, 7>(k, grid, lds, s);
  }
  static bool attr_set_dev[16] = {}; bool& attr_set = vsseg_dev_once(attr_set_dev);  // per device: the LDS opt-in is a per-device function attribute
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<T, NT, MTW, MODE, KS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  // persistent grid = resident workgroups only (registers AND LDS), otherwise the late workgroups form a tail
  static int cached_lds = -1, cached_per_cu = 1;
  if (cached_lds != lds) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, igemm_kernel<T, NT, MTW, MODE, KS>, ig_spec(NT) ? 512 : 256, lds) != hipSuccess || n < 1) n = 1;
    cached_per_cu = n > 6 ? 6 : n;
    cached_lds = lds;
  }
  int64_t gx = 256ll * cached_per_cu;
  // class_split: the rows of all classes resident together (each row its share of the CUs, a multiple of 8 so that the rows agree on the
  // tile -> XCD map): the classes of a tile then run at about the same time on the same XCD — one halo fetch from HBM instead of one per
  // class, and their interleaved output voxels meet in that XCD's L2 before they are written back
  if (k.d.class_split) gx = std::max<int64_t>(8, (gx / k.d.class_split) & ~7ll);
  if (gx > k.total_tiles) gx = k.total_tiles;
  grid.x = (unsigned)gx;
  hipLaunchKernelGGL((igemm_kernel<T, NT, MTW, MODE, KS>), grid, dim3(ig_spec(NT) ? 512 : 256), lds, s, k);
  VSSEG_LAUNCH_CHECK("vsseg_igemm");
  return VSSEG_OK;
}
template <typename T, int NT, int MTW> static int launch(const IgemmK& k, dim3 grid, int lds, hipStream_t s) {
  if (k.d.stats) return launch_mode<T, NT, MTW, IG_STATS>(k, grid, lds, s);
  if (k.aux_bytes > 0) return launch_mode<T, NT, MTW, IG_AUX>(k, grid, lds, s);
  return launch_mode<T, NT, MTW, IG_PLAIN>(k, grid, lds, s);
}
template <typename T, int NT> static int launch_mtw(const IgemmK& k, dim3 grid, int lds, hipStream_t s) {
  switch (k.d.mtw) {
    case 1: return launch<T, NT, 1>(k, grid, lds, s);
    case 2: return launch<T, NT, 2>(k, grid, lds, s);
    case 4: return launch<T, NT, 4>(k, grid, lds, s);
    case 8:  // 512-voxel tiles: only the MFMA-bound configurations (>= 48 output channels per workgroup) — the packed weights of a channel
             // chunk are then streamed once per 512 voxels instead of once per 128-256
      if constexpr (NT >= 3) return launch<T, NT, 8>(k, grid, lds, s);
      break;
  }
  vsseg_set_error("vsseg_igemm: mtw must be 1, 2, 4 (or 8 with nt >= 3), got mtw %d nt %d", k.d.mtw, k.d.nt);
  return VSSEG_EINVAL;
}
