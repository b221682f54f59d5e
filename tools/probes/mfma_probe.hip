// Probe: verify MFMA operand/accumulator lane layouts and ds_read_tr16_b64 semantics on gfx950.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ inline short f2bf(float f) { unsigned u = __float_as_uint(f); u += 0x7fff + ((u >> 16) & 1); return (short)(u >> 16); }

__global__ void k_bf16(const float* A, const float* B, float* D) {  // A[16][32], B[32][16], D[16][16]
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = f2bf(A[(l & 15) * 32 + (l >> 4) * 8 + j]); b[j] = f2bf(B[((l >> 4) * 8 + j) * 16 + (l & 15)]); }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
__global__ void k_f32(const float* A, const float* B, float* D) {  // A[16][4], B[4][16]
  int l = threadIdx.x;
  float a = A[(l & 15) * 4 + (l >> 4)], b = B[(l >> 4) * 16 + (l & 15)];
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
__global__ void k_tr(short* out) {  // each lane reads 8 bytes at lane*8 from an LDS region holding values = element index
  __shared__ __attribute__((aligned(16))) short lds[64 * 4];
  int l = threadIdx.x;
  for (int i = l; i < 256; i += 64) lds[i] = (short)i;
  __syncthreads();
  bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(lds + l * 4));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  std::vector<float> A(16 * 32), B(32 * 16), D(256), R(256);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) A[i * 32 + k] = (float)((i * 7 + k * 3) % 11 - 5);
  for (int k = 0; k < 32; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (float)((k * 5 + j * 13) % 7 - 3);
  float *dA, *dB, *dD; hipMalloc(&dA, 4 * 512); hipMalloc(&dB, 4 * 512); hipMalloc(&dD, 4 * 256);
  hipMemcpy(dA, A.data(), 4 * 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 4 * 512, hipMemcpyHostToDevice);
  k_bf16<<<1, 64>>>(dA, dB, dD); hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
  double e = 0; for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 32; ++k) s += A[i * 32 + k] * B[k * 16 + j]; e = fmax(e, fabs(s - D[i * 16 + j])); }
  printf("bf16 16x16x32 max err %g\n", e);
  std::vector<float> A4(64), B4(64);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A4[i * 4 + k] = (float)((i * 7 + k * 3) % 11 - 5) * 0.37f;
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B4[k * 16 + j] = (float)((k * 5 + j * 13) % 7 - 3) * 1.13f;
  hipMemcpy(dA, A4.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dB, B4.data(), 256, hipMemcpyHostToDevice);
  k_f32<<<1, 64>>>(dA, dB, dD); hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
  e = 0; for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 4; ++k) s = fmaf(A4[i * 4 + k], B4[k * 16 + j], s); e = fmax(e, fabs(s - D[i * 16 + j])); }
  printf("f32 16x16x4 max err %g\n", e);
  short* dO; hipMalloc(&dO, 512); std::vector<short> O(256);
  k_tr<<<1, 64>>>(dO); hipMemcpy(O.data(), dO, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("tr lane %2d: %3d %3d %3d %3d\n", l, O[l * 4], O[l * 4 + 1], O[l * 4 + 2], O[l * 4 + 3]);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); printf("dev %s CUs %d clock %d kHz mem %zu\n", p.gcnArchName, p.multiProcessorCount, p.clockRate, p.totalGlobalMem);
  return 0;
}
