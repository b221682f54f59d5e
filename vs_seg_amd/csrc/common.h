// Device/host helpers shared by the gfx950 kernels of libvsseg_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/vsseg_hip.h"

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

extern "C" void vsseg_set_error(const char* fmt, ...);

// Every kernel launch of the library goes through vsseg_launch_kernel: while the calling thread has an event armed (vsseg_fork_arm, api.cpp) the kernel is launched with
// that event BOUND TO ITS OWN COMPLETION SIGNAL (hipExtLaunchKernelGGL's stopEvent) — how the training backward forks its side stream without a marker packet on the main
// stream: hipEventRecord costs the main stream's next kernel ~4.7 us (5-8 in the step), the bound event ~1.3 (tools/probes/fork_probe.hip).
#include <hip/hip_ext.h>
hipEvent_t vsseg_fork_event();  // the armed event of the calling thread (counted as used), or nullptr
template <typename F, typename... Args> static inline void vsseg_launch_kernel(F kernel, dim3 grid, dim3 block, size_t lds, hipStream_t s, Args... args) {
  if (hipEvent_t ev = vsseg_fork_event()) hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, s, nullptr, ev, 0, args...);
  else hipLaunchKernelGGL(kernel, grid, block, lds, s, args...);
}
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, lds, s, ...) vsseg_launch_kernel(kernel, dim3(grid), dim3(block), lds, s, ##__VA_ARGS__)

#define VSSEG_CHECK(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      vsseg_set_error(__VA_ARGS__);       \
      return VSSEG_EINVAL;                \
    }                                     \
  } while (0)

#define VSSEG_LAUNCH_CHECK(name)                                                   \
  do {                                                                             \
    hipError_t e_ = hipGetLastError();                                             \
    if (e_ != hipSuccess) {                                                        \
      vsseg_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));       \
      return VSSEG_ELAUNCH;                                                        \
    }                                                                              \
  } while (0)

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
typedef __bf16 hbf16x2 __attribute__((ext_vector_type(2)));
typedef float hf32x2 __attribute__((ext_vector_type(2)));
// two fp32 -> packed bf16 pair, round-to-nearest-even: one v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ unsigned f2bf2(float lo, float hi) {
  hbf16x2 h = __builtin_convertvector(hf32x2{lo, hi}, hbf16x2);
  return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(f2bf2(f, 0.f) & 0xffffu); }

template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// load / store 4 consecutive channels as floats (p 8-byte aligned for bf16, 16-byte for f32)
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
  uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(bf16_t* p, float4 v) {
  uint2 u;
  u.x = f2bf2(v.x, v.y);
  u.y = f2bf2(v.z, v.w);
#ifdef VSSEG_NT_STORES
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  __builtin_nontemporal_store(u32x2{u.x, u.y}, reinterpret_cast<u32x2*>(p));
#else
  *reinterpret_cast<uint2*>(p) = u;
#endif
}
// 8 consecutive channels
struct f8 { float v[8]; };
__device__ __forceinline__ f8 ld8(const float* p) {
  float4 a = ld4(p), b = ld4(p + 4);
  return f8{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
}
__device__ __forceinline__ f8 ld8(const bf16_t* p) {
#ifdef VSSEG_NT_LOADS
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  uint4 u = make_uint4(t[0], t[1], t[2], t[3]);
#else
  uint4 u = *reinterpret_cast<const uint4*>(p);
#endif
  return f8{{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u),
             __uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u), __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u)}};
}
__device__ __forceinline__ void st8(float* p, const f8& a) {
  st4(p, make_float4(a.v[0], a.v[1], a.v[2], a.v[3]));
  st4(p + 4, make_float4(a.v[4], a.v[5], a.v[6], a.v[7]));
}
__device__ __forceinline__ void st8(bf16_t* p, const f8& a) {
  uint4 u;
  u.x = f2bf2(a.v[0], a.v[1]);
  u.y = f2bf2(a.v[2], a.v[3]);
  u.z = f2bf2(a.v[4], a.v[5]);
  u.w = f2bf2(a.v[6], a.v[7]);
#ifdef VSSEG_NT_STORES
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(u32x4{u.x, u.y, u.z, u.w}, reinterpret_cast<u32x4*>(p));
#else
  *reinterpret_cast<uint4*>(p) = u;
#endif
}

// ---- Philox4x32-10 counter-based RNG: dropout keep-masks are regenerated in backward instead of stored ----
__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key) {
  const unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    // one 32x32 -> 64-bit product per multiplier (v_mad_u64_u32) instead of a v_mul_hi_u32 + v_mul_lo_u32 pair: the generator is what bounds
    // bn_act_fwd, and these are quarter-rate instructions
    const unsigned long long p0 = (unsigned long long)M0 * ctr.x, p1 = (unsigned long long)M1 * ctr.z;
    const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0, hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
    ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
    key.x += W0;
    key.y += W1;
  }
  return ctr;
}
// keep-mask for the 8 elements [8*i8, 8*i8+8) of a layer's activation tensor (logical [voxel][c] order, pitch-free):
// one Philox call = 128 random bits = eight 16-bit draws, keep iff draw >= round(p * 2^16)
// salt | VSSEG_SEED_INDIRECT: `seed` is the device address of the 64-bit seed instead of its value (a launch list with fixed kernel
// arguments — a captured hipGraph — then draws new masks every step from a seed the host stores with vsseg_store_u64)
__device__ __forceinline__ unsigned dropout_keep8(uint64_t seed, uint32_t salt, uint64_t i8, float p) {
  if (salt & VSSEG_SEED_INDIRECT) {
    seed = *reinterpret_cast<const uint64_t*>(seed);
    salt &= ~VSSEG_SEED_INDIRECT;
  }
  uint4 r = philox4x32_10(make_uint4((unsigned)i8, (unsigned)(i8 >> 32), salt, 0u), make_uint2((unsigned)seed, (unsigned)(seed >> 32)));
  const unsigned thr = (unsigned)(p * 65536.0f + 0.5f);
  unsigned m = 0;
  m |= ((r.x & 0xffffu) >= thr) << 0; m |= ((r.x >> 16) >= thr) << 1;
  m |= ((r.y & 0xffffu) >= thr) << 2; m |= ((r.y >> 16) >= thr) << 3;
  m |= ((r.z & 0xffffu) >= thr) << 4; m |= ((r.z >> 16) >= thr) << 5;
  m |= ((r.w & 0xffffu) >= thr) << 6; m |= ((r.w >> 16) >= thr) << 7;
  return m;
}

// Resolve an indirect seed ONCE per kernel, in front of the element loop: inside dropout_keep8 the read sits behind a branch hipcc cannot
// speculate, i.e. one scalar load + wait per 8 elements.
__device__ __forceinline__ void dropout_resolve_seed(uint64_t& seed, uint32_t& salt) {
  if (salt & VSSEG_SEED_INDIRECT) {
    seed = *reinterpret_cast<const uint64_t*>(seed);
    salt &= ~VSSEG_SEED_INDIRECT;
  }
}

// a*b + c that hipcc may NOT fuse with its neighbours into v_pk_fma_f32 with an op_sel broadcast of one half of a register pair.  That form — four
// products of 4 channels with ONE per-voxel scalar (the attention gate) — returned the addend alone in the low halves of lanes 48-63 in a
// timing-dependent ~1e-4 of the elements (gfx950, ROCm 7.2; found by the run-to-run bit-identity test, reproduced in isolation: the same source
// with unfused v_fma_f32 is exact and deterministic).  The empty asm makes the multiplier opaque per element, so no broadcast pair is formed.
__device__ __forceinline__ float vsseg_fma_unpacked(float a, float b, float c) {
  asm volatile("" : "+v"(b));
  return __builtin_fmaf(a, b, c);
}

__device__ __forceinline__ float vsseg_mul_unpacked(float a, float b) {  // same for a plain product with a broadcast scalar
  asm volatile("" : "+v"(b));
  return a * b;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// 16-byte LDS-DMA (global_load_lds_dwordx4: LDS address = wave-uniform base in M0 + lane*16, global address per lane), issued from inline
// assembly.  Through __builtin_amdgcn_global_load_lds hipcc (ROCm 7.2) tracks the DMA as a pending LDS write and puts `s_waitcnt vmcnt(0)` in
// front of the next ds_read it cannot prove disjoint — in the double-buffered kernels that is the first fragment read of the K loop, right
// after the NEXT stage's DMAs were issued, which serialised "issue stage s+1, wait for it, multiply stage s" (seen in the ISA of
// igemm_kernel<bf16,2,4,0>; the measured ring depths 1..3 were all equal for that reason).  The kernels order DMA against LDS reads
// themselves (counted `s_waitcnt vmcnt(N)` + s_barrier), so nothing is lost by hiding the instruction from the compiler's scoreboard.
// M0 is declared clobbered: hipcc must not keep a live value in it across the DMA (s_set_gpr_idx / v_movrel / its own LDS-DMA builtins use M0).
typedef __attribute__((address_space(3))) const void vsseg_lds_cvoid_t;
__device__ __forceinline__ void vsseg_dma16(const void* gsrc, const void* lds_wave_base) {
  const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(vsseg_lds_cvoid_t*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(lds), "v"(gsrc) : "memory", "m0");
}

// Order-independent accumulation across workgroups: a partial sum is added as a 64-bit FIXED-POINT integer (integer addition is associative, so the
// total does not depend on the order in which workgroups finish — fp64 atomics differ in the last bits from run to run, and training-mode BatchNorm
// turns that into different bf16 roundings downstream).  The accumulator slots are the zeroed `double` buffers of the C ABI reinterpreted as int64.
//   statistics (sum, sum of squares of a convolution output, Dice sums): 2^-20 resolution per addend — an absolute error of <= 1e-6 * addends / count in a
//     mean / variance, far below BatchNorm's eps = 1e-5 and below the fp32 rounding of the partial sums themselves; range 2^43 = 8.8e12 per slot
//   gradient sums (BatchNorm backward, PReLU slope, bias gradients): 2^-44 resolution, range 2^19 = 5e5 per slot (the gradients of an O(1) loss)
// RANGE AND NON-FINITE VALUES.  An addend (one workgroup's partial sum) must satisfy |v * scale| < 2^52 — then up to 2048 addends per slot cannot wrap the
// 64-bit accumulator: statistics |v| < 4.3e9, gradient sums |v| < 256, Dice sums |v| < 1.0e6 per workgroup.  An addend outside that range, or a NaN / Inf
// (a diverging run), is CLAMPED (NaN -> 0) and sets the library's sticky device flag (vsseg_fx_flag(), api.cpp); every kernel that DECODES fixed-point
// sums (vsseg_bn_finalize, vsseg_bn_act_bwd_finalize, vsseg_dice_loss_fwd / _bwd) returns NaN while the flag is set, so that divergence or an
// out-of-range loss scale shows up as NaN in the loss, the statistics and dgamma / dbeta / dalpha exactly as it did with floating-point atomics,
// instead of as silently wrapped integers.  The host reads / clears the flag with vsseg_fx_status().
constexpr double VSSEG_FX_STAT = 1048576.0;         // 2^20
constexpr double VSSEG_FX_GRAD = 17592186044416.0;  // 2^44
constexpr double VSSEG_FX_LIMIT = 4503599627370496.0;  // 2^52, in fixed-point units
unsigned* vsseg_fx_flag();  // api.cpp: the sticky range / non-finite flag word in device memory (nullptr if it could not be allocated)
// the flag word of the current device for a launcher, or an error return: a kernel is never handed a null flag pointer
#define VSSEG_FX_FLAG(var, who)                                                                    \
  unsigned* var = vsseg_fx_flag();                                                                 \
  VSSEG_CHECK(var, who ": could not allocate the fixed-point range flag word on the current device")
__device__ __forceinline__ void vsseg_fx_add(double* slot, double v, double scale, unsigned* flag) {
  double s = v * scale;
  if (!(fabs(s) < VSSEG_FX_LIMIT)) {  // out of range or not finite
    atomicOr(flag, 1u);
    s = s >= VSSEG_FX_LIMIT ? VSSEG_FX_LIMIT : (s <= -VSSEG_FX_LIMIT ? -VSSEG_FX_LIMIT : 0.0);
  }
  atomicAdd(reinterpret_cast<unsigned long long*>(slot), (unsigned long long)__double2ll_rn(s));
}
__device__ __forceinline__ double vsseg_fx_get(const double* slot, double scale) { return (double)*reinterpret_cast<const long long*>(slot) / scale; }
// value a decoding kernel adds to its results: 0, or NaN while the flag is set
__device__ __forceinline__ double vsseg_fx_poison(const unsigned* flag) { return *flag ? __longlong_as_double(0x7ff8000000000000ll) : 0.0; }

// Partial-sum slabs slab[b][i] (b = workgroup of the producing kernel, i = output element) summed over all b by a 1024-thread block that owns 64
// consecutive elements: thread (il = tid & 63, bl = tid >> 6) adds the slabs bl, bl + 16, ... with 8 independent loads in flight (<= 2048 slabs: 16
// dependent rounds), the sixteen block-lanes are combined through LDS in a FIXED order (the result is run-to-run bit-identical).  Threads of
// block-lane 0 return the sum of element i.  One launch per layer: a two-stage tree cost a second ~15 us launch on the eager side stream.
constexpr int VSSEG_SLAB_THREADS = 1024;
__device__ __forceinline__ float vsseg_slab_sum(const float* __restrict__ slab, int64_t total, int64_t i, int nblk, float* lds1024) {
  const int il = threadIdx.x & 63, bl = threadIdx.x >> 6;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (i < total) {
    const float* p = slab + i;
    int b = bl;
    for (; b + 112 < nblk; b += 128) {
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += p[(int64_t)(b + 16 * j) * total];
    }
    for (; b < nblk; b += 16) s[0] += p[(int64_t)b * total];
  }
  lds1024[threadIdx.x] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  float r = 0.f;
  if (bl == 0) {
#pragma unroll
    for (int j = 0; j < 16; ++j) r += lds1024[j * 64 + il];
  }
  return r;
}

// The same sum for FOUR consecutive elements i .. i+3 per thread (i % 4 == 0, total % 4 == 0, 16-byte loads: a wave reads 1 KiB of a slab per instruction
// instead of 256 bytes).  Per element the additions are the ones of vsseg_slab_sum in the same order: the two give bit-identical sums.  lds4096: 1024 float4.
__device__ __forceinline__ f32x4 vsseg_slab_sum4(const float* __restrict__ slab, int64_t total, int64_t i, int nblk, f32x4* lds4096) {
  const int il = threadIdx.x & 63, bl = threadIdx.x >> 6;
  const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 s[8] = {z, z, z, z, z, z, z, z};
  if (i < total) {
    const float* p = slab + i;
    int b = bl;
    for (; b + 112 < nblk; b += 128) {
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += *reinterpret_cast<const f32x4*>(p + (int64_t)(b + 16 * j) * total);
    }
    for (; b < nblk; b += 16) s[0] += *reinterpret_cast<const f32x4*>(p + (int64_t)b * total);
  }
  lds4096[threadIdx.x] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  f32x4 r = z;
  if (bl == 0) {
#pragma unroll
    for (int j = 0; j < 16; ++j) r += lds4096[j * 64 + il];
  }
  return r;
}

// dst[i] += sum over rows b of slab[b][i], i < nvalid <= total (bias-gradient rows of the weight-gradient kernels): the same fixed-order sum
static __global__ __launch_bounds__(VSSEG_SLAB_THREADS) void vsseg_slab_add_kernel(const float* __restrict__ slab, int nrows, int total, int nvalid, float* __restrict__ dst) {
  __shared__ float lds[VSSEG_SLAB_THREADS];
  const int64_t i = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const float s = vsseg_slab_sum(slab, total, i, nrows, lds);
  if (threadIdx.x < 64 && i < nvalid) dst[i] += s;
}

// n / d for 0 <= n < 2^22, d >= 1 with inv = 1.0f / d: a float multiply and one correction step instead of the ~40-instruction sequence hipcc emits for a
// 32-bit division by a run-time value (the coordinate tables of a kernel prologue are made of such divisions: 6.3 us of a 17 us wgrad launch)
__device__ __forceinline__ int vsseg_fdiv(int n, int d, float inv) {
  const int q = (int)((float)n * inv), r = n - q * d;
  return q + (r >= d) - (r < 0);
}

// Workgroup b runs on XCD b % 8 (round-robin dispatch).  Work item of workgroup b such that XCD x owns the contiguous item range
// [x*G/8, (x+1)*G/8) and walks it in dispatch order: neighbouring items (z-columns of the marching kernels: they share the 128-byte lines of
// every tensor with < 128 bytes per z-row — the fp32 attention map, the 2-channel logits) then run at about the same time on the SAME L2
// instead of on eight different ones (PMC: the gate map was fetched 8x, r03_pmc_hbm.txt).  Identity when the grid is not a multiple of 8.
__device__ __forceinline__ int vsseg_xcd_contiguous(int b, int grid) { return (grid & 7) == 0 ? (b & 7) * (grid >> 3) + (b >> 3) : b; }

// The once-flag of a launcher for the CURRENT device (hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device function attribute: a process-wide flag
// left the opt-in for more than 64 KiB of LDS unapplied on a second device of the same process)
static inline bool& vsseg_dev_once(bool (&flags)[16]) {
  static bool never;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { never = false; return never; }
  return flags[dev];
}
static inline int64_t tensor_voxels(const vsseg_tensor& t) { return (int64_t)t.n * t.x * t.y * t.z; }
static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int grid_for(int64_t work_items, int block, int cap = 256 * 16) {
  int64_t g = (work_items + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}
