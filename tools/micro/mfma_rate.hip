// What one wave gets out of the matrix pipe and the LDS on this machine: N dependent / independent fp32 MFMAs per wave, with
// 1 or 2 waves per SIMD, timed with s_memrealtime (100 MHz) inside the kernel.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void k(unsigned long long *out, float *sink, int n) {
  __shared__ float lds[64 * 68];
  for (int i = threadIdx.x; i < 64 * 68; i += blockDim.x) lds[i] = (float)i;
  __syncthreads();
  f32x16 acc = {0}, acc2 = {0};
  f32x4 c4 = {0}, c4b = {0};
  float a = threadIdx.x, b = 1.f;
  const int lane = threadIdx.x & 63;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (MODE == 0) {  // dependent 32x32x2
    for (int i = 0; i < n; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  } else if (MODE == 1) {  // two independent chains 32x32x2
    for (int i = 0; i < n; i += 2) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc2, 0, 0, 0);
    }
  } else if (MODE == 2) {  // dependent 16x16x4
    for (int i = 0; i < n; ++i) c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c4, 0, 0, 0);
  } else if (MODE == 3) {  // two independent 16x16x4
    for (int i = 0; i < n; i += 2) {
      c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c4, 0, 0, 0);
      c4b = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c4b, 0, 0, 0);
    }
  } else if (MODE == 4) {  // the GEMM's inner pattern: 2 ds_read_b32 per MFMA, operands read 8 MFMAs ahead
    const float *A = lds + (lane & 31) + (lane >> 5) * 68;
    float av[2][8], bv[2][8];
    for (int j = 0; j < 8; ++j) { av[0][j] = A[2 * j * 68]; bv[0][j] = A[2 * j * 68 + 32]; }
    for (int i = 0; i < n; i += 16) {
      for (int j = 0; j < 8; ++j) { av[1][j] = A[(16 + 2 * j) * 68]; bv[1][j] = A[(16 + 2 * j) * 68 + 32]; }
      for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][j], bv[0][j], acc, 0, 0, 0);
      for (int j = 0; j < 8; ++j) { av[0][j] = A[(32 + 2 * j) * 68]; bv[0][j] = A[(32 + 2 * j) * 68 + 32]; }
      for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][j], bv[1][j], acc, 0, 0, 0);
      asm volatile("" ::: "memory");
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r] + acc2[r];
  s += c4[0] + c4[1] + c4b[2] + c4b[3];
  if (s == 12345.678f) *sink = s;
  if (lane == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
void run(const char *what, int waves_per_cu, int n, double flop_per_mfma) {
  const int blocks = 256, threads = 64 * waves_per_cu;
  unsigned long long *d;
  float *sink;
  hipMalloc(&d, blocks * waves_per_cu * 8);
  hipMalloc(&sink, 4);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, sink, n);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * waves_per_cu);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double us = h[h.size() / 2] / 100.0;
  const double per = us * 1e3 / n;  // ns per MFMA per wave
  const double tf = 256.0 * waves_per_cu * n * flop_per_mfma / (us * 1e-6) / 1e12;
  printf("%-52s %2d waves/CU: %7.2f us for %d MFMAs per wave = %6.2f ns each; all 256 CUs %6.1f TFLOP/s\n", what, waves_per_cu, us, n, per, tf);
  hipFree(d);
  hipFree(sink);
}

int main() {
  const int n = 4096;
  for (int w : {4, 8, 16}) {
    run<0>("32x32x2 f32, one dependent chain", w, n, 4096);
    run<1>("32x32x2 f32, two independent chains", w, n, 4096);
    run<2>("16x16x4 f32, one dependent chain", w, n, 2048);
    run<3>("16x16x4 f32, two independent chains", w, n, 2048);
    run<4>("32x32x2 f32 + 2 ds_read_b32 per MFMA (GEMM pattern)", w, n, 4096);
  }
  return 0;
}
