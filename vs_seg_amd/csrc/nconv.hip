// Narrow-output convolution: the stride-1 3x3x1 convolution with ONE output channel that closes an attention block of the two finest levels
// (AttentionBlock1.conv2 + sigmoid, ref:params/networks/blocks/attentionblock.py:20-35; SURVEY §8a rows 42, 47: 32 -> 1 at 192x64x128, 16 -> 1 at 384x128x128),
// forward, as a bandwidth kernel on the vector ALUs: vsseg_conv_to1.
//
// One output channel is 1/16 of an MFMA tile.  The launches it replaces ran on the matrix cores anyway — z-folded onto the general kernel (8 z-neighbours as the
// channel group, block-diagonal weights: 0.35 ms for the 0.9 GB of the 16 -> 1 layer at batch 4 = 2.6 TB/s, 1.7x its bytes in HBM traffic, 40 % of its LDS cycles
// bank conflicts) or on the marching kernel with a one-value epilogue (0.39 ms), and round 5's vector-ALU attempt staged an 8x8x16 tile with its (8+2)x(8+2) halo
// through LDS-DMA and waited for it (0.46 ms).  Here NO input voxel is fetched twice and nothing is staged:
//
//   * a workgroup owns all Y rows x TZ voxels of z (Y * TZ = 512 threads: no halo in y) and marches along x; thread (y, z) loads ITS voxel of the plane —
//     C channels = C/8 16-byte loads, issued one plane ahead — and multiplies it with all nine taps:  p[dx][dy] = sum_c in[x'][y][z][c] * w[dx][dy][c], C/2
//     v_dot2c_f32_bf16 per tap on the packed channel pairs as they come from memory (the weights, rounded to bf16 as the packed weights of the MFMA path are,
//     sit in registers)
//   * what the output voxel (x, y) needs from its neighbours are their PARTIAL SUMS, not their inputs:  out[x][y] = sum_dx sum_dy p_{(x + dx - 1, y + dy - 1)}[dx][dy].
//     Along y the partials of the rows above and below come through LDS (6 floats written and 6 read per thread and plane, one barrier per plane, two buffers);
//     along x the same thread sees the planes one after the other: two running sums in registers
//   * bias + sigmoid, one fp32 (or bf16) value stored per thread and plane.
// HBM traffic = the input once (+ 2 planes per x segment) + the one-channel output.  Same products as the MFMA launches (bf16 x bf16 in fp32), another summation order.
#include "common.h"

struct NconvK {
  const char* in;
  char* out;
  const float* w;     // [1][C][3][3][1] fp32 master weights (element (c, dx, dy) at c*9 + dx*3 + dy)
  const float* bias;  // [1] or nullptr
  int in_vox_bytes, out_f32, act;
  int X, Y, Z, lx, nxs, nzb;
};

typedef __bf16 nc_bf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float nc_dot2(unsigned a, unsigned b, float c) {  // c + a.lo * b.lo + a.hi * b.hi (v_dot2c_f32_bf16)
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(nc_bf2, a), __builtin_bit_cast(nc_bf2, b), c, false);
}

template <int C, int TZ>
__global__ __launch_bounds__(512) void nconv_kernel(const NconvK k) {
  constexpr int CP = C / 2, NL = C / 8, ROWS = 512 / TZ, SLOTS = (ROWS + 2) * TZ;  // channel pairs; 16-byte loads per voxel; rows of the workgroup (= Y); exchange slots per (buffer, dx, side)
  __shared__ float xch[2][2][3][SLOTS];  // [buffer][side: 0 = for the row below (dy = 0 partials), 1 = for the row above (dy = 2)][dx][slot = (row + 1) * TZ + z]; rows 0 and ROWS + 1 stay zero
  __shared__ unsigned wl[9 * CP];
  const int tid = threadIdx.x, z = tid % TZ, y = tid / TZ;
  const int X = k.X, Y = k.Y, Z = k.Z;
  int b = vsseg_xcd_contiguous(blockIdx.x, gridDim.x);  // z blocks of one (sample, x segment) are neighbours on one XCD: they share the 128-byte lines of the output rows
  const int zb = b % k.nzb; b /= k.nzb;
  const int xs = b % k.nxs; const int n = b / k.nxs;
  const int z0 = zb * TZ, xa = xs * k.lx, xb = min(X, xa + k.lx);

  for (int i = tid; i < 2 * 2 * 3 * SLOTS; i += 512) (&xch[0][0][0][0])[i] = 0.f;
  for (int i = tid; i < 9 * CP; i += 512) {  // tap-major packed pairs: wl[tap * CP + j] = (w[2j], w[2j + 1]) of tap = dx*3 + dy, rounded to bf16
    const int tap = i / CP, j = i - tap * CP;
    wl[i] = f2bf2(k.w[(2 * j) * 9 + tap], k.w[(2 * j + 1) * 9 + tap]);
  }
  __syncthreads();
  unsigned w[9][CP];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < CP; ++j) w[t][j] = wl[t * CP + j];
  const float bias = k.bias ? k.bias[0] : 0.f;

  const int64_t col = (((int64_t)n * X) * Y + y) * Z + z0 + z;  // voxel (n, 0, y, z0 + z)
  const int64_t plane = (int64_t)Y * Z;
  const char* ip = k.in + col * k.in_vox_bytes;
  uint4 nxt[NL];
  auto load = [&](int xx) {  // plane xx into nxt (zeros outside the image: unconditional loads from a clamped plane, select on the value)
    const bool inside = (unsigned)xx < (unsigned)X;
    const char* p = ip + (int64_t)(inside ? xx : 0) * plane * k.in_vox_bytes;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const uint4 v = *reinterpret_cast<const uint4*>(p + i * 16);
      nxt[i] = inside ? v : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  load(xa - 1);
  float acc0 = 0.f, acc1 = 0.f;  // acc1: out[x' - 1] without plane x'; acc0: out[x'] without planes x', x' + 1
  const int slot = (y + 1) * TZ + z;
  for (int xx = xa - 1; xx <= xb; ++xx) {
    unsigned cur[CP];
#pragma unroll
    for (int i = 0; i < NL; ++i) { cur[4 * i] = nxt[i].x; cur[4 * i + 1] = nxt[i].y; cur[4 * i + 2] = nxt[i].z; cur[4 * i + 3] = nxt[i].w; }
    if (xx < xb) load(xx + 1);
    float p[3][3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < CP; ++j) s = nc_dot2(w[dx * 3 + dy][j], cur[j], s);
        p[dx][dy] = s;
      }
    const int bf = (xx - xa + 1) & 1;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) { xch[bf][0][dx][slot] = p[dx][0]; xch[bf][1][dx][slot] = p[dx][2]; }
    __syncthreads();  // (two buffers: a thread can be at most one barrier ahead of the slowest reader of the other buffer)
    float r[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) r[dx] = p[dx][1] + xch[bf][0][dx][slot - TZ] + xch[bf][1][dx][slot + TZ];  // row y - 1's dy = 0 partial, row y + 1's dy = 2 partial
    // plane xx feeds out[xx + 1] with its dx = 0 taps, out[xx] with dx = 1, out[xx - 1] with dx = 2 (out[x] = sum_dx in[x + dx - 1] ...)
    const float done = acc1 + r[2];
    acc1 = acc0 + r[1];
    acc0 = r[0];
    const int xo = xx - 1;
    if (xo >= xa) {  // (xo < xb always: xx <= xb)
      float v = done + bias;
      if (k.act == VSSEG_ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
      const int64_t o = col + (int64_t)xo * plane;
      if (k.out_f32) reinterpret_cast<float*>(k.out)[o] = v;
      else reinterpret_cast<bf16_t*>(k.out)[o] = f2bf(v);
    }
  }
}

template <int C, int TZ> static int nc_launch(const NconvK& k, int grid, hipStream_t s) {
  hipLaunchKernelGGL((nconv_kernel<C, TZ>), dim3((unsigned)grid), dim3(512), 0, s, k);
  VSSEG_LAUNCH_CHECK("vsseg_conv_to1");
  return VSSEG_OK;
}
template <int C> static int nc_tz(const NconvK& k, int tz, int grid, hipStream_t s) {
  switch (tz) {
    case 2: return nc_launch<C, 2>(k, grid, s);
    case 4: return nc_launch<C, 4>(k, grid, s);
    case 8: return nc_launch<C, 8>(k, grid, s);
    case 16: return nc_launch<C, 16>(k, grid, s);
    case 32: return nc_launch<C, 32>(k, grid, s);
  }
  vsseg_set_error("vsseg_conv_to1: the y extent must be 16, 32, 64, 128 or 256 (512 threads = y rows x z voxels)");
  return VSSEG_EINVAL;
}

extern "C" int vsseg_conv_to1(vsseg_tensor in, const float* w, const float* bias, int32_t act, vsseg_tensor out, int32_t lx, void* stream) {
  VSSEG_CHECK(in.ptr && out.ptr && w, "vsseg_conv_to1: null pointer");
  VSSEG_CHECK(in.dtype == VSSEG_BF16 && !in.ptr2 && (in.c == 16 || in.c == 32) && in.pitch % 8 == 0 && !((uintptr_t)in.ptr & 15), "vsseg_conv_to1: the input must be a one-part bf16 tensor of 16 or 32 channels, 16-byte aligned voxel rows");
  VSSEG_CHECK(!out.ptr2 && out.c == 1 && out.pitch == 1 && (out.dtype == VSSEG_F32 || out.dtype == VSSEG_BF16), "vsseg_conv_to1: the output must be a dense one-channel fp32 / bf16 tensor");
  VSSEG_CHECK(out.n == in.n && out.x == in.x && out.y == in.y && out.z == in.z, "vsseg_conv_to1: extents differ");
  VSSEG_CHECK(act == VSSEG_ACT_NONE || act == VSSEG_ACT_SIGMOID, "vsseg_conv_to1: activation must be none or sigmoid");
  VSSEG_CHECK(in.y >= 16 && in.y <= 256 && 512 % in.y == 0, "vsseg_conv_to1: the y extent must be 16, 32, 64, 128 or 256 (got %d)", in.y);
  const int tz = 512 / in.y;
  VSSEG_CHECK(in.z % tz == 0, "vsseg_conv_to1: the z extent must be a multiple of %d", tz);
  NconvK k;
  k.in = reinterpret_cast<const char*>(in.ptr); k.out = reinterpret_cast<char*>(out.ptr); k.w = w; k.bias = bias;
  k.in_vox_bytes = in.pitch * 2; k.out_f32 = out.dtype == VSSEG_F32; k.act = act;
  k.X = in.x; k.Y = in.y; k.Z = in.z;
  k.lx = lx < 1 ? in.x : (lx > in.x ? in.x : lx);
  k.nxs = (k.X + k.lx - 1) / k.lx; k.nzb = k.Z / tz;
  const int64_t grid = (int64_t)in.n * k.nxs * k.nzb;
  VSSEG_CHECK(grid > 0 && grid < (1ll << 24), "vsseg_conv_to1: bad grid");
  return in.c == 16 ? nc_tz<16>(k, tz, (int)grid, as_stream(stream)) : nc_tz<32>(k, tz, (int)grid, as_stream(stream));
}
