// Probe: does the 256 MiB Infinity Cache (memory-side L3) pay for the SECOND of two consecutive streaming passes over one tensor, and does the ORDER of the second pass
// matter?  A tensor larger than the cache read twice in the same order is the LRU worst case (every line evicted before its re-use); read the second time in REVERSE order
// the last ~cache-size bytes of the first pass are the first bytes of the second.  Variants per size S:
//   first pass : read (plain loads) | write (plain stores) | write (non-temporal stores)
//   second pass: read forward | read reverse, each with plain and with non-temporal loads
// Prints the second pass's time and bytes/s.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mall_probe tools/probes/mall_probe.hip && tools/probes/mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned v4u __attribute__((ext_vector_type(4)));

// each workgroup owns one contiguous chunk of `per` 16-byte pieces; chunk index = blockIdx (forward) or gridDim-1-blockIdx (reverse): workgroups are dispatched in blockIdx order
template <bool NT>
__global__ __launch_bounds__(256) void read_pass(const v4u* __restrict__ buf, size_t per, int reverse, unsigned* __restrict__ out) {
  const size_t chunk = reverse ? (size_t)(gridDim.x - 1 - blockIdx.x) : (size_t)blockIdx.x;
  const v4u* p = buf + chunk * per;
  v4u acc = {0, 0, 0, 0};
  if (reverse) {
    for (size_t i = per - 256 + threadIdx.x + 256; i >= 256; i -= 256) {
      const v4u v = NT ? __builtin_nontemporal_load(p + i - 256) : p[i - 256];
      acc += v;
    }
  } else {
    for (size_t i = threadIdx.x; i < per; i += 256) {
      const v4u v = NT ? __builtin_nontemporal_load(p + i) : p[i];
      acc += v;
    }
  }
  const unsigned s = acc.x + acc.y + acc.z + acc.w;
  if (s == 0x12345678u) out[blockIdx.x] = s;  // never true for the fill used: keeps the loads alive without a store per thread
}

template <bool NT>
__global__ __launch_bounds__(256) void write_pass(v4u* __restrict__ buf, size_t per, unsigned val) {
  v4u* p = buf + (size_t)blockIdx.x * per;
  const v4u v = {val, val + 1, val + 2, val + 3};
  for (size_t i = threadIdx.x; i < per; i += 256) {
    if (NT) __builtin_nontemporal_store(v, p + i);
    else p[i] = v;
  }
}

// read-modify-write pass (what an element-wise kernel does): reads a, writes b
__global__ __launch_bounds__(256) void copy_pass(const v4u* __restrict__ a, v4u* __restrict__ b, size_t per, int reverse) {
  const size_t chunk = reverse ? (size_t)(gridDim.x - 1 - blockIdx.x) : (size_t)blockIdx.x;
  const v4u* p = a + chunk * per;
  v4u* q = b + chunk * per;
  for (size_t i = threadIdx.x; i < per; i += 256) __builtin_nontemporal_store(__builtin_nontemporal_load(p + i) + 1u, q + i);
}

int main() {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t max_bytes = (size_t)1024 << 20;
  v4u *buf, *buf2;
  unsigned* out;
  CK(hipMalloc(&buf, max_bytes));
  CK(hipMalloc(&buf2, max_bytes));
  CK(hipMalloc(&out, 1 << 20));
  CK(hipMemset(buf, 1, max_bytes));
  CK(hipMemset(buf2, 1, max_bytes));
  const int sizes_mb[] = {64, 128, 192, 256, 384, 512, 768, 1024};
  printf("second pass over the same S bytes, time in us (TB/s); first pass in forward order\n");
  printf("%8s | %-22s | %14s %14s %14s %14s\n", "S MB", "first pass", "fwd plain", "rev plain", "fwd nt", "rev nt");
  for (int mb : sizes_mb) {
    const size_t bytes = (size_t)mb << 20, pieces = bytes / 16;
    const size_t per = 2048;  // 32 KB per workgroup: many short workgroups dispatched in blockIdx order, like the element-wise kernels
    const int grid = (int)(pieces / per);
    for (int first = 0; first < 4; ++first) {
      float t[4];
      for (int second = 0; second < 4; ++second) {
        const int rev = second & 1, nt = second >> 1;
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
          // flush: stream the OTHER buffer through the caches so every repetition starts from the same state
          hipLaunchKernelGGL(read_pass<false>, dim3(4096), dim3(256), 0, st, buf2, (max_bytes / 16) / 4096, 0, out);
          if (first == 0) hipLaunchKernelGGL(read_pass<false>, dim3(grid), dim3(256), 0, st, buf, per, 0, out);
          else if (first == 1) hipLaunchKernelGGL(read_pass<true>, dim3(grid), dim3(256), 0, st, buf, per, 0, out);
          else if (first == 2) hipLaunchKernelGGL(write_pass<false>, dim3(grid), dim3(256), 0, st, buf, per, 7u);
          else hipLaunchKernelGGL(write_pass<true>, dim3(grid), dim3(256), 0, st, buf, per, 7u);
          CK(hipEventRecord(e0, st));
          if (nt) hipLaunchKernelGGL(read_pass<true>, dim3(grid), dim3(256), 0, st, buf, per, rev, out);
          else hipLaunchKernelGGL(read_pass<false>, dim3(grid), dim3(256), 0, st, buf, per, rev, out);
          CK(hipEventRecord(e1, st));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best) best = ms;
        }
        t[second] = best;
      }
      static const char* names[] = {"read plain", "read nt", "write plain", "write nt"};
      printf("%8d | %-22s |", mb, names[first]);
      for (int k = 0; k < 4; ++k) printf(" %7.1f (%4.2f)", t[k] * 1e3, bytes / (t[k] * 1e-3) / 1e12);
      printf("\n");
    }
  }
  // element-wise chain: copy a -> b forward, then copy b -> a forward / reverse (reads what was just written, writes what was just read)
  printf("\nelement-wise chain: pass 1 copies a->b forward (nt loads and stores); pass 2 copies b->a; time of pass 2 in us (TB/s of read+write bytes)\n");
  for (int mb : sizes_mb) {
    const size_t bytes = (size_t)mb << 20, pieces = bytes / 16;
    const size_t per = 2048;
    const int grid = (int)(pieces / per);
    float t[2];
    for (int rev = 0; rev < 2; ++rev) {
      float best = 1e30f;
      for (int rep = 0; rep < 5; ++rep) {
        hipLaunchKernelGGL(copy_pass, dim3(grid), dim3(256), 0, st, buf, buf2, per, 0);
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(copy_pass, dim3(grid), dim3(256), 0, st, buf2, buf, per, rev);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      t[rev] = best;
    }
    printf("%8d MB per tensor | fwd %7.1f (%4.2f) | rev %7.1f (%4.2f)\n", mb, t[0] * 1e3, 2.0 * bytes / (t[0] * 1e-3) / 1e12, t[1] * 1e3, 2.0 * bytes / (t[1] * 1e-3) / 1e12);
  }
  return 0;
}
