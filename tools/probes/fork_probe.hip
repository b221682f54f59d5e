// Probe: what a FORK of a second stream costs the first one.  The training backward forks a weight-gradient launch off the main stream ~42 times per step (hipEventRecord on the
// main stream + hipStreamWaitEvent on the side stream), and the main stream's next kernel starts 5-8 us late behind each.  A chain of N dependent ~busy kernels on
// the main stream, a small side kernel forked after every one:
//   (a) no forks                                                     -> us per main kernel
//   (b) hipEventRecord(main) + hipStreamWaitEvent(side)              -> what the launch lists do
//   (c) the event bound to the kernel's own completion signal (hipExtLaunchKernelGGL stopEvent) + hipStreamWaitEvent(side): no marker packet on the main stream
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o tools/probes/fork_probe tools/probes/fork_probe.hip && tools/probes/fork_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void busy_kernel(float* __restrict__ p, int iters) {
  float v = p[blockIdx.x * blockDim.x + threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  p[blockIdx.x * blockDim.x + threadIdx.x] = v;
}

int main() {
  const int N = 200, G = 1024;
  float *a, *b;
  CK(hipMalloc(&a, G * 256 * 4)); CK(hipMalloc(&b, G * 256 * 4));
  CK(hipMemset(a, 0, G * 256 * 4)); CK(hipMemset(b, 0, G * 256 * 4));
  hipStream_t m, s;
  CK(hipStreamCreateWithFlags(&m, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  std::vector<hipEvent_t> ev(N);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (int iters : {2000, 20000}) {
    for (int mode = 0; mode < 3; ++mode) {
      double best = 1e30;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) {
          if (mode == 2) hipExtLaunchKernelGGL(busy_kernel, dim3(G), dim3(256), 0, m, nullptr, ev[i], 0, a, iters);
          else hipLaunchKernelGGL(busy_kernel, dim3(G), dim3(256), 0, m, a, iters);
          if (mode == 1) CK(hipEventRecord(ev[i], m));
          if (mode) {
            CK(hipStreamWaitEvent(s, ev[i], 0));
            hipLaunchKernelGGL(busy_kernel, dim3(8), dim3(256), 0, s, b, 100);
          }
        }
        CK(hipStreamSynchronize(m));
        auto t1 = std::chrono::steady_clock::now();
        CK(hipDeviceSynchronize());
        best = std::min(best, std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
      }
      printf("iters %6d  %-62s %7.2f us per main kernel\n", iters, mode == 0 ? "(a) no forks" : mode == 1 ? "(b) hipEventRecord + hipStreamWaitEvent" : "(c) stopEvent of the kernel (hipExtLaunchKernelGGL) + hipStreamWaitEvent", best);
    }
  }
  return 0;
}
