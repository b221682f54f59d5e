// Probe: how fast can a SPECIALISED (compile-time geometry) streaming 3x3x1 convolution run on the full-resolution 16-channel layer?
// bf16 channels-last [N][X][Y][Z][16] -> [N][X][Y][Z][16], stride 1, zero padding, bias; tile 8x8x4 voxels per workgroup, halo 10x10x4 via
// LDS-DMA, packed weights resident in LDS, 20 MFMAs (16x16x32 bf16) per wave and tile.  Everything the generic vsseg_igemm kernel keeps in
// (spilled) SGPRs or LDS tables is a compile-time constant or a per-thread register here.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/lean_conv_probe.hip -o tools/probes/lean_conv_probe && tools/probes/lean_conv_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>
#include <string.h>

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;
__device__ __forceinline__ void dma16(const void* g, char* l) { __builtin_amdgcn_global_load_lds((gvoid_t*)g, (lvoid_t*)l, 16, 0, 0); }
typedef __bf16 hbf2 __attribute__((ext_vector_type(2)));
typedef float hf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(hf2{a, b}, hbf2)); }

constexpr int C = 16, TX = 8, TY = 8, TZ = 4, HX = TX + 2, HY = TY + 2, HZ = TZ;
constexpr int HVOX = HX * HY * HZ, PIECES = HVOX * 2, NINST = (PIECES + 255) / 256;  // 16-byte pieces; DMA instructions per wave
constexpr int KS = 5;                                                                 // ceil(9 taps * 2 groups / 4)
constexpr int W_BYTES = KS * 64 * 16, H_BYTES = NINST * 256 * 16;

struct Args {
  const bf16_t* in; bf16_t* out; const bf16_t* wpack; const float* bias; const void* zeros;
  int N, X, Y, Z, ntx, nty, ntz; int64_t tiles;
};

template <int NBUF>
__global__ __launch_bounds__(256, 4) void lean_conv(const Args a) {
  __shared__ __attribute__((aligned(16))) char smem[W_BYTES + NBUF * H_BYTES];
  char* Wl = smem; char* Hl = smem + W_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, l15 = lane & 15;
  const int Y = a.Y, Z = a.Z, X = a.X;
  for (int i = tid; i < W_BYTES / 16; i += 256) reinterpret_cast<uint4*>(Wl)[i] = reinterpret_cast<const uint4*>(a.wpack)[i];
  // per-thread DMA pieces: u-th instruction of this wave covers pieces (u*4+wave)*64+lane
  unsigned rel[NINST]; unsigned hxy[NINST];
#pragma unroll
  for (int u = 0; u < NINST; ++u) {
    const int j = (u * 4 + wave) * 64 + lane;
    const int hv = j >> 1, c16 = j & 1, hz = hv % HZ, r = hv / HZ, hy = r % HY, hx = r / HY;
    const bool ok = j < PIECES;
    rel[u] = ok ? (unsigned)(((hx * Y + hy) * Z + hz) * (C * 2) + c16 * 16) : 0u;
    hxy[u] = ok ? (unsigned)(hx | (hy << 8)) : 0xffffu;
  }
  // fragment addressing: K-group p = ks*4+g -> tap p>>1 (dx = tap/3, dy = tap%3), channel group p&1
  int koff[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int p = ks * 4 + g, tap = p >> 1, cg = p & 1;
    koff[ks] = tap < 9 ? (((tap / 3) * HY + (tap % 3)) * HZ) * (C * 2) + cg * 16 : 0;  // padded K-groups multiply zero weights
  }
  const int v0 = wave * 64 + l15;  // voxel of M-tile 0; M-tile m adds 16 voxels
  const int vb0 = ((((v0 >> 5) * HY) + ((v0 >> 2) & 7)) * HZ + (v0 & 3)) * (C * 2);
  const int ob0 = ((((v0 >> 5) * Y) + ((v0 >> 2) & 7)) * Z + (v0 & 3)) * (C * 2) + g * 8;  // output byte offset inside the tile (relative to its first voxel)
  float bias[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bias[r] = a.bias[g * 4 + r];
  const char* Wlane = Wl + lane * 16;
  const int64_t sample_bytes = (int64_t)X * Y * Z * C * 2;
  __syncthreads();
  auto issue = [&](int64_t t, char* Hb) {
    int b = (int)t;
    const int tz = b % a.ntz; b /= a.ntz;
    const int ty = b % a.nty; b /= a.nty;
    const int tx = b % a.ntx; const int n = b / a.ntx;
    const int x0 = tx * TX, y0 = ty * TY, z0 = tz * TZ;
    const bool interior = x0 > 0 && y0 > 0 && x0 + TX < X && y0 + TY < Y;
    const char* origin = reinterpret_cast<const char*>(a.in) + n * sample_bytes + ((int64_t)((x0 - 1) * Y + (y0 - 1)) * Z + z0) * (C * 2);
    if (interior) {
#pragma unroll
      for (int u = 0; u < NINST; ++u) dma16(origin + rel[u], Hb + (u * 4 + wave) * 1024);
    } else {
#pragma unroll
      for (int u = 0; u < NINST; ++u) {
        const int gx = x0 - 1 + (int)(hxy[u] & 255u), gy = y0 - 1 + (int)(hxy[u] >> 8);
        const bool ok = hxy[u] != 0xffffu && (unsigned)gx < (unsigned)X && (unsigned)gy < (unsigned)Y;
        dma16(ok ? (const void*)(origin + rel[u]) : a.zeros, Hb + (u * 4 + wave) * 1024);
      }
    }
  };
  int cur = 0;
  if (NBUF == 2 && (int64_t)blockIdx.x < a.tiles) issue(blockIdx.x, Hl);
  for (int64_t t = blockIdx.x; t < a.tiles; t += gridDim.x) {
    int b = (int)t;
    const int tz = b % a.ntz; b /= a.ntz;
    const int ty = b % a.nty; b /= a.nty;
    const int tx = b % a.ntx; const int n = b / a.ntx;
    const int x0 = tx * TX, y0 = ty * TY, z0 = tz * TZ;
    char* Hc = Hl + cur * H_BYTES;
    if (NBUF == 1) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // every wave finished reading the previous tile
      issue(t, Hc);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (NBUF == 2: also drains the previous tile's output stores)
    __builtin_amdgcn_s_barrier();
    if (NBUF == 2) {  // the other buffer was last read during the previous iteration, which every wave has left by now
      if (t + gridDim.x < a.tiles) issue(t + gridDim.x, Hl + (cur ^ 1) * H_BYTES);
      cur ^= 1;
    }
    f32x4 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 w = *reinterpret_cast<const bf16x8*>(Wlane + ks * 1024);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        // M-tile m: 16 voxels further = vy + 4 (m odd), vx + 1 (m >= 2)
        const bf16x8 av = *reinterpret_cast<const bf16x8*>(Hc + vb0 + koff[ks] + (m & 1) * (4 * HZ * C * 2) + (m >> 1) * (HY * HZ * C * 2));
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, av, acc[m], 0, 0, 0);
      }
    }
    char* otile = reinterpret_cast<char*>(a.out) + n * sample_bytes + ((int64_t)(x0 * Y + y0) * Z + z0) * (C * 2);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      uint2 o;
      o.x = pk(acc[m][0] + bias[0], acc[m][1] + bias[1]);
      o.y = pk(acc[m][2] + bias[2], acc[m][3] + bias[3]);
      *reinterpret_cast<uint2*>(otile + ob0 + ((m & 1) * 4 * Z + (m >> 1) * Y * Z) * (C * 2)) = o;
    }
  }
}

static float bf2f(bf16_t h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
static bf16_t f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (bf16_t)(u >> 16); }

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 4, X = 384, Y = 128, Z = 128;
  const int64_t vox = (int64_t)N * X * Y * Z;
  std::vector<bf16_t> hin(vox * C), hout(vox * C), hw(9 * C * C);
  srand(1);
  for (auto& v : hin) v = f2bf((rand() % 2001 - 1000) / 1000.f);
  for (auto& v : hw) v = f2bf((rand() % 2001 - 1000) / 4000.f);
  std::vector<float> hb(C);
  for (auto& v : hb) v = (rand() % 200 - 100) / 100.f;
  // pack W[tap][cin][cout] -> [ks][lane][8]: lane = g*16 + n, K-group p = ks*4+g -> (tap = p>>1, cin = (p&1)*8 + j)
  std::vector<bf16_t> hp(KS * 64 * 8, 0);
  for (int ks = 0; ks < KS; ++ks)
    for (int lane = 0; lane < 64; ++lane)
      for (int j = 0; j < 8; ++j) {
        const int p = ks * 4 + (lane >> 4), tap = p >> 1, ci = (p & 1) * 8 + j, co = lane & 15;
        if (tap < 9) hp[(ks * 64 + lane) * 8 + j] = hw[(tap * C + ci) * C + co];
      }
  bf16_t *din, *dout, *dw; float* db; void* dz;
  hipMalloc(&din, vox * C * 2); hipMalloc(&dout, vox * C * 2); hipMalloc(&dw, hp.size() * 2); hipMalloc(&db, C * 4); hipMalloc(&dz, 256);
  hipMemcpy(din, hin.data(), vox * C * 2, hipMemcpyHostToDevice); hipMemcpy(dw, hp.data(), hp.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(db, hb.data(), C * 4, hipMemcpyHostToDevice); hipMemset(dz, 0, 256);
  Args a{din, dout, dw, db, dz, N, X, Y, Z, X / TX, Y / TY, Z / TZ, (int64_t)N * (X / TX) * (Y / TY) * (Z / TZ)};
  for (int nbuf = 1; nbuf <= 2; ++nbuf) {
    auto kern = nbuf == 1 ? lean_conv<1> : lean_conv<2>;
    int per_cu = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, 0);
    printf("NBUF %d: occupancy %d workgroups/CU, LDS %d B, %lld tiles\n", nbuf, per_cu, W_BYTES + nbuf * H_BYTES, (long long)a.tiles);
    for (int wpc = 1; wpc <= per_cu; ++wpc) {
      const int grid = 256 * wpc;
      hipMemset(dout, 0, vox * C * 2);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, a);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, a);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
      printf("  %d WG/CU: %.3f ms  %.0f GB/s (in + out)\n", wpc, ms, 2.0 * vox * C * 2 / ms / 1e6);
    }
  }
  hipMemcpy(hout.data(), dout, vox * C * 2, hipMemcpyDeviceToHost);
  // spot check against the definition (zero padding), 2000 random outputs incl. borders
  double worst = 0;
  for (int s = 0; s < 2000; ++s) {
    const int n = rand() % N, x = s < 200 ? (s & 1 ? 0 : X - 1) : rand() % X, y = s < 400 ? (s & 2 ? 0 : Y - 1) : rand() % Y, z = rand() % Z, co = rand() % C;
    double ref = hb[co];
    for (int dx = 0; dx < 3; ++dx)
      for (int dy = 0; dy < 3; ++dy) {
        const int gx = x + dx - 1, gy = y + dy - 1;
        if (gx < 0 || gx >= X || gy < 0 || gy >= Y) continue;
        for (int ci = 0; ci < C; ++ci) ref += (double)bf2f(hin[((((int64_t)n * X + gx) * Y + gy) * Z + z) * C + ci]) * bf2f(hw[((dx * 3 + dy) * C + ci) * C + co]);
      }
    const double got = bf2f(hout[((((int64_t)n * X + x) * Y + y) * Z + z) * C + co]);
    worst = fmax(worst, fabs(got - ref) / (fabs(ref) + 0.5));
  }
  printf("max relative error of 2000 samples: %.4f (%s)\n", worst, worst < 0.02 ? "OK" : "WRONG");
  return worst < 0.02 ? 0 : 1;
}
