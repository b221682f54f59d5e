// Probe: what a DEPENDENT STAGE BOUNDARY costs on this GPU, three ways — the number that decides whether the ~125 small launches of levels 3-5 are better off as one
// persistent launch with grid barriers (VERDICT rounds 2-4) or as launches.  Each variant runs the same chain of N trivial dependent stages (every workgroup adds 1 to its
// own word; stage s+1 reads what stage s wrote) on G workgroups of 256 threads:
//   (a) N plain launches back to back on one stream                       -> us per launch boundary (what the eager / hipGraph-replayed launch lists pay)
//   (b) the same N launches captured in a hipGraph and replayed            -> us per graph node
//   (c) ONE launch of G co-resident workgroups, N grid barriers inside     -> us per barrier: a monotonic counter (release fence + atomic arrive + acquire-load poll), and the
//       XCD-hierarchical form (per-XCD counter, leader -> top counter -> per-XCD generation word), /opt/skills/guides/MI355X_MICROARCH.md "barrier-counter" / "barrier-xcd"
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o tools/probes/chain_probe tools/probes/chain_probe.hip && tools/probes/chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void stage_kernel(unsigned* __restrict__ words, int stage) {
  if (threadIdx.x == 0) {
    const unsigned nb = gridDim.x, me = blockIdx.x;
    const unsigned prev = words[(me + 1) % nb];  // the neighbour's word of the previous stage: a real cross-workgroup dependency
    words[nb + me] = prev + (unsigned)stage;
    words[me] += 1;
  }
}

__device__ __forceinline__ void barrier_counter(unsigned* counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

// per-XCD arrival counters [8], top counter, per-XCD generation words [8] (each on its own 128-byte line)
__device__ __forceinline__ void barrier_xcd(unsigned* st, unsigned epoch, unsigned per_xcd) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned xcd = blockIdx.x & 7u;
    unsigned* arrive = st + xcd * 32, *top = st + 8 * 32, *gen = st + (9 + xcd) * 32;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned old = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == epoch * per_xcd) {  // last of this XCD: tell the top, wait for all 8 XCDs, release this XCD
      __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * 8u) __builtin_amdgcn_s_sleep(1);
      __hip_atomic_store(gen, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

template <int KIND>
__global__ void chain_kernel(unsigned* __restrict__ words, unsigned* __restrict__ sync, int stages, unsigned base_epoch) {
  const unsigned nb = gridDim.x, me = blockIdx.x;
  for (int s = 0; s < stages; ++s) {
    if (threadIdx.x == 0) {
      const unsigned prev = __hip_atomic_load(&words[(me + 1) % nb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      words[nb + me] = prev + (unsigned)s;
      __hip_atomic_store(&words[me], __hip_atomic_load(&words[me], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (KIND == 0) barrier_counter(sync, (base_epoch + (unsigned)s + 1) * nb);
    else barrier_xcd(sync, base_epoch + (unsigned)s + 1, nb / 8);
  }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const int N = 200;
  for (int G : {64, 128, 256, 512}) {
    unsigned *words, *sync;
    CK(hipMalloc(&words, 2 * G * sizeof(unsigned)));
    CK(hipMalloc(&sync, 4096 * sizeof(unsigned)));
    CK(hipMemset(words, 0, 2 * G * sizeof(unsigned)));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    // (a) plain launches
    for (int w = 0; w < 2; ++w) {
      CK(hipEventRecord(e0, st));
      for (int s = 0; s < N; ++s) hipLaunchKernelGGL(stage_kernel, dim3(G), dim3(256), 0, st, words, s);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
    }
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double t_launch = ms * 1e3 / N;
    // (b) graph replay
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int s = 0; s < N; ++s) hipLaunchKernelGGL(stage_kernel, dim3(G), dim3(256), 0, st, words, s);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 2; ++w) {
      CK(hipEventRecord(e0, st));
      CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
    }
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double t_graph = ms * 1e3 / N;
    // (c) one launch, grid barriers inside
    double t_bar[2];
    for (int kind = 0; kind < 2; ++kind) {
      CK(hipMemset(sync, 0, 4096 * sizeof(unsigned)));
      unsigned epoch = 0;
      for (int w = 0; w < 2; ++w) {
        CK(hipEventRecord(e0, st));
        if (kind == 0) hipLaunchKernelGGL(chain_kernel<0>, dim3(G), dim3(256), 0, st, words, sync, N, epoch);
        else hipLaunchKernelGGL(chain_kernel<1>, dim3(G), dim3(256), 0, st, words, sync, N, epoch);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        epoch += N;
      }
      CK(hipEventElapsedTime(&ms, e0, e1));
      t_bar[kind] = ms * 1e3 / N;
    }
    // host-side cost of issuing one launch (no GPU wait)
    const double h0 = now_us();
    for (int s = 0; s < N; ++s) hipLaunchKernelGGL(stage_kernel, dim3(G), dim3(256), 0, st, words, s);
    const double h1 = now_us();
    CK(hipStreamSynchronize(st));
    printf("G=%4d workgroups: plain launches %.2f us/stage | hipGraph replay %.2f us/stage | in-kernel barrier: counter %.2f, XCD-hierarchical %.2f us/stage | host issue %.2f us/launch\n", G, t_launch, t_graph, t_bar[0], t_bar[1],
           (h1 - h0) / N);
    CK(hipFree(words)); CK(hipFree(sync));
  }
  return 0;
}
