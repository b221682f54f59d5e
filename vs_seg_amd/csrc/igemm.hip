// Host side of the implicit-GEMM convolution: argument checks, LDS layout, tile-descriptor table, dispatch.
// The kernel itself is in igemm_kernel.h (instantiated by igemm_inst.hip).
#include "igemm_kernel.h"
#include "sconv.h"
#include "cconv.h"
#include "mconv.h"
#include "dconv.h"
#include "tconv.h"
#include "gconv.h"
#include "chain.h"

__global__ void igemm_tile_setup_kernel(const IgemmK k, TileDesc* __restrict__ tab, int xb) {
  const vsseg_igemm_desc& d = k.d;
  const int64_t per_sample = (int64_t)k.ntile[0] * k.ntile[1] * k.ntile[2];
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < k.total_tiles; t += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(t / per_sample);
    int64_t b = t - n * per_sample;
    const int64_t band_tiles = (int64_t)xb * k.ntile[1] * k.ntile[2];
    const int band = (int)(b / band_tiles);
    b -= band * band_tiles;
    const int bw = min(xb, k.ntile[0] - band * xb);  // the last band may be narrower
    const int tz = (int)(b % k.ntile[2]); b /= k.ntile[2];
    const int txl = (int)(b % bw);
    const int ty = (int)(b / bw);
    const int tx = band * xb + txl;
    TileDesc td;
    td.n = n;
    td.q0[0] = tx * d.tile[0]; td.q0[1] = ty * d.tile[1]; td.q0[2] = tz * d.tile[2];
    bool interior = true, whole = true;
    const int dims_in[3] = {d.in.x, d.in.y, d.in.z}, dims_out[3] = {d.out.x, d.out.y, d.out.z};
    int o0[3];
    for (int a = 0; a < 3; ++a) {
      td.g0[a] = td.q0[a] * d.is[a] + k.off_min[a];
      interior = interior && td.g0[a] >= 0 && td.g0[a] + k.halo[a] <= dims_in[a];
      o0[a] = td.q0[a] * d.os[a] + d.oo[a];
      int oo_hi = d.oo[a];  // class_split: the tile must be whole for every class (the largest class offset)
      for (int c = 0; c < d.class_split; ++c) oo_hi = max(oo_hi, d.class_oo[c][a]);
      whole = whole && td.q0[a] + d.tile[a] <= d.q[a] && (td.q0[a] + d.tile[a] - 1) * d.os[a] + oo_hi < dims_out[a];
    }
    td.in_vox = (((int64_t)n * d.in.x + td.g0[0]) * d.in.y + td.g0[1]) * d.in.z + td.g0[2];
    td.out_vox = (((int64_t)n * d.out.x + o0[0]) * d.out.y + o0[1]) * d.out.z + o0[2];
    td.flags = (interior ? 1 : 0) | (whole ? 2 : 0);
    tab[t] = td;
  }
}


// igemm_inst.hip, compiled once per (element type, NT)
int vsseg_igemm_launch_f32_1(const IgemmK& k, dim3 grid, int lds, hipStream_t s);
int vsseg_igemm_launch_f32_2(const IgemmK& k, dim3 grid, int lds, hipStream_t s);
int vsseg_igemm_launch_f32_3(const IgemmK& k, dim3 grid, int lds, hipStream_t s);
int vsseg_igemm_launch_f32_4(const IgemmK& k, dim3 grid, int lds, hipStream_t s);
int vsseg_igemm_launch_f32_5(const IgemmK& k, dim3 grid, int lds, hipStream_t s);
int vsseg_igemm_launch_f32_6(const IgemmK& k, dim3 grid, int lds, hipStream_t s);
int vsseg_igemm_launch_bf16_1(const IgemmK& k, dim3 grid, int lds, hipStream_t s);
int vsseg_igemm_launch_bf16_2(const IgemmK& k, dim3 grid, int lds, hipStream_t s);
int vsseg_igemm_launch_bf16_3(const IgemmK& k, dim3 grid, int lds, hipStream_t s);
int vsseg_igemm_launch_bf16_4(const IgemmK& k, dim3 grid, int lds, hipStream_t s);
int vsseg_igemm_launch_bf16_5(const IgemmK& k, dim3 grid, int lds, hipStream_t s);
int vsseg_igemm_launch_bf16_6(const IgemmK& k, dim3 grid, int lds, hipStream_t s);
static int launch_nt(bool f32, const IgemmK& k, dim3 grid, int lds, hipStream_t s) {
  typedef int (*fn_t)(const IgemmK&, dim3, int, hipStream_t);
  static const fn_t tab[2][6] = {{vsseg_igemm_launch_bf16_1, vsseg_igemm_launch_bf16_2, vsseg_igemm_launch_bf16_3, vsseg_igemm_launch_bf16_4, vsseg_igemm_launch_bf16_5, vsseg_igemm_launch_bf16_6},
                                 {vsseg_igemm_launch_f32_1, vsseg_igemm_launch_f32_2, vsseg_igemm_launch_f32_3, vsseg_igemm_launch_f32_4, vsseg_igemm_launch_f32_5, vsseg_igemm_launch_f32_6}};
  if (k.d.nt < 1 || k.d.nt > 6) {
    vsseg_set_error("vsseg_igemm: nt must be 1..6 (got %d)", k.d.nt);
    return VSSEG_EINVAL;
  }
  return tab[f32 ? 1 : 0][k.d.nt - 1](k, grid, lds, s);
}

#include <map>
#include <vector>
static const TileDesc* tile_table(const IgemmK& k, hipStream_t stream) {
  static std::map<std::vector<int64_t>, TileDesc*> cache;
  const vsseg_igemm_desc& d = k.d;
  int xb = 128 / k.ntile[2];  // ~2 scheduling steps of one XCD per (band, y) row
  xb = xb < 1 ? 1 : (xb > k.ntile[0] ? k.ntile[0] : xb);
  std::vector<int64_t> key = {d.in.n, d.in.x, d.in.y, d.in.z, d.out.x, d.out.y, d.out.z, k.total_tiles, xb, d.class_split};
  for (int c = 0; c < d.class_split; ++c) for (int a = 0; a < 3; ++a) key.push_back(d.class_oo[c][a]);
  for (int a = 0; a < 3; ++a) { key.push_back(d.q[a]); key.push_back(d.tile[a]); key.push_back(d.is[a]); key.push_back(d.os[a]); key.push_back(d.oo[a]); key.push_back(k.off_min[a]); key.push_back(k.halo[a]); }
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  TileDesc* tab = nullptr;
  if (hipMalloc(&tab, sizeof(TileDesc) * k.total_tiles) != hipSuccess) return nullptr;
  hipLaunchKernelGGL(igemm_tile_setup_kernel, dim3((unsigned)((k.total_tiles + 255) / 256 > 1024 ? 1024 : (k.total_tiles + 255) / 256)), dim3(256), 0, stream, k, tab, xb);
  cache[key] = tab;
  return tab;
}

static const void* zero_page() {
  static void* z = nullptr;
  if (!z) {
    if (hipMalloc(&z, 256) != hipSuccess || hipMemset(z, 0, 256) != hipSuccess) z = nullptr;
  }
  return z;
}

const void* vsseg_zero_page() { return zero_page(); }  // (chain.hip)

static int igemm_prepare(const vsseg_igemm_desc* d, IgemmK& k) {
  VSSEG_CHECK(d && d->in.ptr && d->out.ptr && d->wpack, "vsseg_igemm: null pointer");
  VSSEG_CHECK(d->in.dtype == VSSEG_F32 || d->in.dtype == VSSEG_BF16, "vsseg_igemm: bad input dtype");
  VSSEG_CHECK(d->ntaps >= 1 && d->ntaps <= VSSEG_MAX_TAPS, "vsseg_igemm: ntaps %d out of range", d->ntaps);
  VSSEG_CHECK(d->ck >= 8 && d->ck % 8 == 0 && d->nchunks >= 1, "vsseg_igemm: bad channel chunking ck=%d nchunks=%d", d->ck, d->nchunks);
  VSSEG_CHECK(d->in.c % 8 == 0 && d->in.pitch % 8 == 0, "vsseg_igemm: input channels/pitch must be multiples of 8 (c=%d pitch=%d)", d->in.c, d->in.pitch);
  VSSEG_CHECK(d->tile[0] * d->tile[1] * d->tile[2] == 64 * d->mtw, "vsseg_igemm: tile %dx%dx%d != 64*mtw", d->tile[0], d->tile[1], d->tile[2]);
  VSSEG_CHECK(d->nsplit >= 1 && (d->class_split ? d->nt : d->nsplit * d->nt) * 16 >= d->out.c, "vsseg_igemm: nsplit*nt*16 < cout");
  if (d->class_split) {  // all output-parity classes in one launch: workgroup row = class
    VSSEG_CHECK(d->class_split >= 2 && d->class_split <= 8 && d->nsplit == d->class_split, "vsseg_igemm: class_split must be 2..8 and equal nsplit");
    VSSEG_CHECK(d->oo[0] == 0 && d->oo[1] == 0 && d->oo[2] == 0 && d->cout_mod == 0 && d->depth >= -1 && !d->out.ptr2 && d->res_mode != VSSEG_RES_GATE,
                "vsseg_igemm: class_split needs oo = 0, the general kernel (depth >= -1), a one-part output and no gated residual");
    for (int c = 0; c < d->class_split; ++c) {
      VSSEG_CHECK(d->class_ntaps[c] >= 1 && d->class_ntaps[c] <= 8 && d->class_ntaps[c] <= d->ntaps, "vsseg_igemm: class %d has %d taps", c, d->class_ntaps[c]);
      for (int t = 0; t < d->class_ntaps[c]; ++t) VSSEG_CHECK(d->class_tap[c][t] >= 0 && d->class_tap[c][t] < d->ntaps, "vsseg_igemm: class %d tap %d out of range", c, t);
      for (int a = 0; a < 3; ++a) VSSEG_CHECK(d->class_oo[c][a] >= 0 && d->class_oo[c][a] < d->os[a], "vsseg_igemm: class offset outside the output stride");
    }
  }
  for (const vsseg_tensor* t : {&d->in, &d->out, &d->res}) {
    if (t == &d->res && d->res_mode == VSSEG_RES_NONE) continue;
    if (t->ptr2) VSSEG_CHECK(t->csplit > 0 && t->csplit < t->c && t->csplit % 16 == 0 && t->pitch >= t->csplit && t->pitch >= t->c - t->csplit,
                             "vsseg_igemm: bad two-part tensor (c=%d csplit=%d pitch=%d)", t->c, t->csplit, t->pitch);
  }
  VSSEG_CHECK(!d->in.ptr2 || d->nchunks == 1 || d->in.csplit % d->ck == 0, "vsseg_igemm: a channel chunk (ck=%d) straddles the input split at %d", d->ck, d->in.csplit);
  VSSEG_CHECK(d->res_mode != VSSEG_RES_GATE || (d->gate && !d->accumulate), "vsseg_igemm: RES_GATE needs the gate map and does not combine with accumulate");
  k.d = *d;
  const int es = d->in.dtype == VSSEG_F32 ? 4 : 2;
  k.cgs = d->ck / 8;
  VSSEG_CHECK(d->ksteps * 4 >= d->ntaps * k.cgs, "vsseg_igemm: ksteps too small");
  k.total_tiles = d->in.n;
  for (int a = 0; a < 3; ++a) {
    int lo = d->tap_off[0][a], hi = lo;
    for (int t = 1; t < d->ntaps; ++t) { lo = min(lo, d->tap_off[t][a]); hi = max(hi, d->tap_off[t][a]); }
    k.off_min[a] = lo;
    k.halo[a] = (d->tile[a] - 1) * d->is[a] + (hi - lo + 1);
    k.ntile[a] = (d->q[a] + d->tile[a] - 1) / d->tile[a];
    k.total_tiles *= k.ntile[a];
    VSSEG_CHECK(k.halo[a] <= 255 && d->tile[a] <= 255, "vsseg_igemm: tile/halo extent > 255");
  }
  for (int c = 0; c < 8; ++c) k.class_vox[c] = c < d->class_split ? (d->class_oo[c][0] * d->out.y + d->class_oo[c][1]) * d->out.z + d->class_oo[c][2] : 0;
  k.w_bytes = d->ksteps * d->nt * 64 * 8 * es;
  k.h_bytes = k.halo[0] * k.halo[1] * k.halo[2] * d->ck * es;
  k.aux_mode = 0;
  k.aux_gate_off = 0;
  k.aux_bytes = 0;
  if (d->accumulate && d->res_mode == VSSEG_RES_NONE) { k.aux_mode = 1; k.aux = d->out; }
  else if (!d->accumulate && d->res_mode == VSSEG_RES_ADD) { k.aux_mode = 2; k.aux = d->res; }
  else if (!d->accumulate && d->res_mode == VSSEG_RES_RELUMASK) { k.aux_mode = 3; k.aux = d->res; }
  else if (!d->accumulate && d->res_mode == VSSEG_RES_GATE) { k.aux_mode = 4; k.aux = d->res; }
  if (d->stats) k.aux_mode = 0;  // statistics + residual in one launch does not occur in this network: generic epilogue
  if (k.aux_mode) {
    const int aes = k.aux.dtype == VSSEG_F32 ? 4 : 2;
    const int row = d->nt * 16 * aes;
    const int beyond = d->nsplit * d->nt * 16 - (k.aux.ptr2 ? k.aux.csplit : 0);  // channels the DMA rows touch in the last part
    const bool ok = (k.aux.pitch % 8) == 0 && (d->out.c % 4) == 0 && beyond <= k.aux.pitch && 64 * d->mtw * (row / 16) <= amax_for(d->mtw) * 256 && ((uintptr_t)k.aux.ptr2 % 16) == 0 &&
                    ((uintptr_t)k.aux.ptr % 16) == 0 && k.aux.c >= d->out.c;
    if (ok) k.aux_bytes = 64 * d->mtw * row; else k.aux_mode = 0;
    // the gate map of a gated add is DMA-prefetched with the tile when 4 z-consecutive tile voxels are 16 contiguous, aligned bytes
    if (ok && k.aux_mode == 4 && d->tile[2] % 4 == 0 && d->out.z % 4 == 0 && d->os[0] == 1 && d->os[1] == 1 && d->os[2] == 1 && d->oo[0] == 0 && d->oo[1] == 0 && d->oo[2] == 0 &&
        ((uintptr_t)d->gate % 16) == 0) {
      k.aux_gate_off = k.aux_bytes;
      k.aux_bytes += 64 * d->mtw * 4;
    }
  }
  VSSEG_CHECK(k.h_bytes <= PMAX * 256 * 16, "vsseg_igemm: halo chunk of %d bytes exceeds %d; reduce ck or the tile", k.h_bytes, PMAX * 256 * 16);
  VSSEG_CHECK(d->ck * es / 16 <= 255, "vsseg_igemm: ck too large");
  int off = 0;
  k.lds_ktab = off; off += ((d->ksteps * 4 * 4 + 15) / 16) * 16;
  k.lds_epi = off; off += 3 * d->nt * 16 * 4;
  k.depth = d->depth < 0 ? 0 : (d->depth == 0 ? 1 : (d->depth > 3 ? 3 : d->depth));  // -1: no prefetch (single buffer); 0: default = 1
  if (d->nt >= 3 && k.depth < 1) k.depth = 1;  // producer / consumer wave specialisation (igemm_kernel.h) always prefetches
  const int nbuf = k.depth + 1;
  k.lds_w = off; off += k.w_bytes * (d->nchunks > 1 ? nbuf : 1);
  k.h_stride = (k.h_bytes + 1023) / 1024 * 1024;      // ring buffers padded to whole 1 KiB DMA instructions (igemm_kernel.h: unconditional lanes)
  k.aux_stride = (k.aux_bytes + 1023) / 1024 * 1024;
  k.lds_h = off; off += nbuf * k.h_stride;
  k.lds_aux = off; off += nbuf * k.aux_stride;
  k.npu = (k.h_bytes / 16 + 255) / 256;
  {
    auto magic = [](int dv) { return dv <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)dv - 1) / (unsigned)dv); };  // exact floor(n / dv) = umulhi(n, magic) for n * dv < 2^32; 0 = divisor 1
    k.mg_ppv = magic(d->ck * es / 16);
    k.mg_hyz = magic(k.halo[1] * k.halo[2]);
    k.mg_hz = magic(k.halo[2]);
  }
  k.lds_pinfo = off; off += ((d->nt >= 3 ? 0 : 2 * k.npu) + (64 * d->mtw + 255) / 256) * 1024;  // nt >= 3: the DMA pieces are decoded arithmetically, no per-thread tables  // per-thread tables of the DMA pieces (packed halo coordinates, interior byte offsets) + tile-voxel coordinates (partial tiles)
  VSSEG_CHECK(off <= 160 * 1024, "vsseg_igemm: needs %d bytes of LDS (> 160 KiB); reduce ck or the tile", off);
  return off;
}

extern "C" int vsseg_igemm_lds_bytes(const vsseg_igemm_desc* d) {
  if (d && (d->depth == -2 || d->depth == -4)) return vsseg_sconv_lds_bytes(d);
  if (d && d->depth == -3) return vsseg_cconv_lds_bytes(d);
  if (d && (d->depth == -5 || d->depth == -6)) return vsseg_mconv_lds_bytes(d);
  if (d && d->depth == -7) return vsseg_dconv_lds_bytes(d);
  if (d && d->depth == -8) return vsseg_tconv_lds_bytes(d);
  if (d && d->depth == -9) return vsseg_gconv_lds_bytes(d);
  IgemmK k;
  return igemm_prepare(d, k);
}

extern "C" int vsseg_igemm(const vsseg_igemm_desc* d, void* stream) {
  VSSEG_CHECK(!d || !d->in_gate || d->depth == -5 || d->depth == -6, "vsseg_igemm: the input gate (in_gate) needs a marching-kernel plan (depth -5 / -6)");
  VSSEG_CHECK(!d || d->res_mode != VSSEG_RES_IN1 || d->depth == -5 || d->depth == -6, "vsseg_igemm: VSSEG_RES_IN1 needs a marching-kernel plan (depth -5 / -6)");
  VSSEG_CHECK(!d || !d->res_tiles || d->depth == -5 || d->depth == -6, "vsseg_igemm: residual tiles (res_tiles) need a marching-kernel plan (depth -5 / -6)");
  if (d && (d->depth == -2 || d->depth == -4)) {  // streaming kernel (sconv.hip; -4: fused output-parity classes): fails loudly when the launch is outside its domain, never falls back
    VSSEG_CHECK(d->in.ptr && d->out.ptr && d->wpack, "vsseg_igemm: null pointer");
    const void* z = zero_page();
    VSSEG_CHECK(z, "vsseg_igemm: could not allocate the zero page");
    return vsseg_sconv_launch(d, z, as_stream(stream));
  }
  if (d && d->depth == -3) {  // compute-bound kernel (cconv.hip): same contract — outside its domain is an error
    VSSEG_CHECK(d->in.ptr && d->out.ptr && d->wpack, "vsseg_igemm: null pointer");
    const void* z = zero_page();
    VSSEG_CHECK(z, "vsseg_igemm: could not allocate the zero page");
    return vsseg_cconv_launch(d, z, as_stream(stream));
  }
  if (d && (d->depth == -5 || d->depth == -6)) {  // marching streaming kernel (mconv.hip; -6: packed weights in registers): same contract
    VSSEG_CHECK(d->in.ptr && d->out.ptr && d->wpack, "vsseg_igemm: null pointer");
    const void* z = zero_page();
    VSSEG_CHECK(z, "vsseg_igemm: could not allocate the zero page");
    return vsseg_mconv_launch(d, z, as_stream(stream));
  }
  if (d && d->depth == -7) {  // deep-level kernel (dconv.hip: the small launches of levels 3-5 and the stride-2 transitions around them): same contract
    VSSEG_CHECK(d->in.ptr && d->out.ptr && d->wpack, "vsseg_igemm: null pointer");
    const void* z = zero_page();
    VSSEG_CHECK(z, "vsseg_igemm: could not allocate the zero page");
    return vsseg_dconv_launch(d, z, as_stream(stream));
  }
  if (d && d->depth == -8) {  // transition kernel (tconv.hip: the eight parity classes of a 3x3x3 stride-(2,2,2) transition between levels 2 and 3 in one launch): same contract
    VSSEG_CHECK(d->in.ptr && d->out.ptr && d->wpack, "vsseg_igemm: null pointer");
    const void* z = zero_page();
    VSSEG_CHECK(z, "vsseg_igemm: could not allocate the zero page");
    return vsseg_tconv_launch(d, z, as_stream(stream));
  }
  if (d && d->depth == -9) {  // gathering marching kernel (gconv.hip: the stride-(2,2,1) 3x3x1 launches that read the fine level and write the coarse one): same contract
    VSSEG_CHECK(d->in.ptr && d->out.ptr && d->wpack, "vsseg_igemm: null pointer");
    const void* z = zero_page();
    VSSEG_CHECK(z, "vsseg_igemm: could not allocate the zero page");
    return vsseg_gconv_launch(d, z, as_stream(stream));
  }
  IgemmK k;
  int lds = igemm_prepare(d, k);
  if (lds < 0) return lds;
  k.zeros = zero_page();
  k.fxflag = vsseg_fx_flag();
  VSSEG_CHECK(k.zeros && k.fxflag, "vsseg_igemm: could not allocate the zero page / flag word");
  VSSEG_CHECK(k.total_tiles > 0 && k.total_tiles < (1ll << 31), "vsseg_igemm: bad tile count");
  k.tiles = tile_table(k, as_stream(stream));
  VSSEG_CHECK(k.tiles, "vsseg_igemm: could not allocate the tile table");
  dim3 grid(1, (unsigned)d->nsplit);  // grid.x is set to the resident workgroup count by launch<>()
#ifdef VSSEG_IG_PROF
  static unsigned long long* prof = nullptr;
  if (!prof) hipMalloc(&prof, 16 * 8);
  hipMemset(prof, 0, 16 * 8);
  k.prof = prof;
  int rc = launch_nt(d->in.dtype == VSSEG_F32, k, grid, lds, as_stream(stream));
  if (getenv("VSSEG_IG_PROF_PRINT")) {
    unsigned long long h[16];
    hipStreamSynchronize(as_stream(stream));
    hipMemcpy(h, prof, sizeof(h), hipMemcpyDeviceToHost);
    const double n = h[8] ? (double)h[8] : 1.0;
    fprintf(stderr, "igemm prof (workgroup 8, wave 0; cycles per stage over %llu stages): wait-others %.0f | dma-issue %.0f | dma-wait %.0f | barrier %.0f | k-loop %.0f | epilogue %.0f\n", h[8], h[0] / n, h[1] / n, h[2] / n,
            h[3] / n, h[4] / n, h[5] / n);
  }
  return rc;
#else
  return launch_nt(d->in.dtype == VSSEG_F32, k, grid, lds, as_stream(stream));
#endif
}
