// Probe: does `global_load_lds_dwordx4 voff, s[base:base+1]` (SGPR base + 32-bit VGPR offset) behave like the builtin's
// 64-bit-VGPR-address form?  Each lane copies 16 B from src + lane-dependent offset into LDS at M0 + lane*16, then LDS is dumped.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void lvoid_t;
__global__ void probe(const char* src, unsigned* out, int big) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned voff = (unsigned)(lane * 32 + wave * 4096) + (big ? 0x90000000u : 0u);  // big: offset >= 2^31 (signed/unsigned test)
  const char* base = big ? src - 0x90000000ll : src;
  unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lvoid_t*)(smem + wave * 1024));
  unsigned long long b = (unsigned long long)base;
  const char* sb = (const char*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(b >> 32)) << 32) | (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)b));
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds), "v"(voff), "s"(sb) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned* l = reinterpret_cast<const unsigned*>(smem);
  for (int i = threadIdx.x; i < 4 * 256; i += 256) out[i] = l[i];
}
int main() {
  const int N = 1 << 16;
  unsigned* h = (unsigned*)malloc(N * 4);
  for (int i = 0; i < N; ++i) h[i] = i;
  char* d; unsigned* o;
  hipMalloc(&d, N * 4); hipMalloc(&o, 4096 * 4);
  hipMemcpy(d, h, N * 4, hipMemcpyHostToDevice);
  for (int big = 0; big < 2; ++big) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 4096, 0, d, o, big);
    unsigned r[1024];
    if (hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost) != hipSuccess) { printf("big=%d: launch failed: %s\n", big, hipGetErrorString(hipGetLastError())); return 1; }
    int bad = 0;
    for (int w = 0; w < 4; ++w) for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 4; ++j) {
      unsigned want = (lane * 32 + w * 4096) / 4 + j;
      if (r[w * 256 + lane * 4 + j] != want) ++bad;
    }
    printf("big=%d: %d mismatches (first words %u %u %u %u %u)\n", big, bad, r[0], r[1], r[4], r[5], r[256]);
  }
  return 0;
}
