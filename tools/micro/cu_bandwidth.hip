// What ONE compute unit can pull through its vector memory path: every workgroup (one per CU) streams the same / its own
// region with 16-byte loads, U loads in flight per thread, W waves.  Regions: 1 MB shared by all (L2-resident after the
// first pass), 1 MB per workgroup (256 MB in total: Infinity Cache / HBM).  s_memrealtime inside the kernel.
// Build: hipcc --offload-arch=gfx950 -O3 -o cu_bandwidth cu_bandwidth.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int U>
__global__ void k(const float4 *buf, size_t region_f4, int shared, int passes, unsigned long long *out, float *sink) {
  const float4 *p = buf + (shared ? 0 : (size_t)blockIdx.x * region_f4);
  float acc = 0.f;
  const int nt = blockDim.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int pass = 0; pass < passes; ++pass) {
    for (size_t i = threadIdx.x; i + (size_t)(U - 1) * nt < region_f4; i += (size_t)U * nt) {
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = p[i + (size_t)u * nt];
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[u].x + v[u].w;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (acc == 12345.f) *sink = acc;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int U>
void run(const float4 *buf, int waves, int shared, int blocks) {
  const size_t region = (1 << 20) / 16;
  const int passes = 8;
  unsigned long long *d;
  float *sink;
  (void)hipMalloc(&d, blocks * 8);
  (void)hipMalloc(&sink, 4);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<U>, dim3(blocks), dim3(64 * waves), 0, 0, buf, region, shared, passes, d, sink);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks);
  (void)hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double us = h[blocks / 2] / 100.0, gb = passes * 1048576.0 / (us * 1e-6) / 1e9;
  printf("%-26s %3d workgroups x %2d waves, %2d loads in flight per thread: %7.1f GB/s per CU, %6.2f TB/s over the launch\n",
         shared ? "1 MB shared (L2)" : "1 MB per workgroup (memory)", blocks, waves, U, gb, gb * blocks / 1e3);
  (void)hipFree(d);
  (void)hipFree(sink);
}

int main() {
  float4 *buf;
  (void)hipMalloc(&buf, (size_t)256 << 20);
  (void)hipMemset(buf, 0, (size_t)256 << 20);
  for (int shared : {1, 0})
    for (int blocks : {1, 32, 256})
      for (int waves : {4, 8, 16}) {
        run<1>(buf, waves, shared, blocks);
        run<4>(buf, waves, shared, blocks);
        run<8>(buf, waves, shared, blocks);
      }
  return 0;
}
