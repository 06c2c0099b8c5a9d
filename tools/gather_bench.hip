// gather_bench.hip -- microbenchmarks that price the access patterns of the FM step on one MI355X:
//   (1) launch floor: an empty kernel, back to back
//   (2) random row gathers: rows of 64 / 128 / 136 bytes at strides 128 / 192 / 256 B from tables of 1M and 8M rows,
//       160K rows per launch (one B=4096 x 39 step) and 4M rows per launch (steady state)
//   (3) random row read-modify-write of the same shapes
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_bench tools/gather_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_empty() {}

// LPR lanes per row, each lane loads NV float4 (16 B) at offsets (v * LPR + q) * 16 (+ optional 8-byte tail by lane 0)
template <int LPR, int NV, bool TAIL, bool RMW>
__global__ __launch_bounds__(256) void k_gather(float* table, const int* rows, int n_rows, int stride_f, float* sink) {
  const int g = (blockIdx.x * blockDim.x + threadIdx.x) / LPR;
  const int q = threadIdx.x % LPR;
  if (g >= n_rows) return;
  float* rp = table + (size_t)rows[g] * stride_f;
  float4 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const float4*>(rp + (i * LPR + q) * 4);
  float2 t = {0.f, 0.f};
  if (TAIL && q == 0) t = *reinterpret_cast<const float2*>(rp + NV * LPR * 4);
  float acc = t.x + t.y;
#pragma unroll
  for (int i = 0; i < NV; ++i) acc += v[i].x + v[i].y + v[i].z + v[i].w;
  if (RMW) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      v[i].x += 1.f; v[i].y += 1.f; v[i].z += 1.f; v[i].w += 1.f;
      *reinterpret_cast<float4*>(rp + (i * LPR + q) * 4) = v[i];
    }
    if (TAIL && q == 0) { t.x += 1.f; *reinterpret_cast<float2*>(rp + NV * LPR * 4) = t; }
  }
  if (acc == 123.456f) *sink = acc;
}

template <int LPR, int NV, bool TAIL, bool RMW>
double time_gather(float* table, const int* rows, int n, int stride_f, float* sink, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int threads = 256;
  const int blocks = (int)(((size_t)n * LPR + threads - 1) / threads);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_gather<LPR, NV, TAIL, RMW>), dim3(blocks), dim3(threads), 0, 0, table, rows, n, stride_f, sink);
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_gather<LPR, NV, TAIL, RMW>), dim3(blocks), dim3(threads), 0, 0, table, rows, n, stride_f, sink);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3 / iters;  // us per launch
}

int main() {
  // launch floor
  {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, 0);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, 0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("empty kernel (1024 x 256), back to back: %.2f us per launch\n", ms);
  }
  float* sink; CK(hipMalloc(&sink, 4));
  std::mt19937 rng(1);
  for (long R : {1L << 20, 8L << 20}) {
    for (int stride_f : {32, 36, 48, 64}) {
      float* table; CK(hipMalloc(&table, (size_t)R * stride_f * 4)); CK(hipMemset(table, 0, (size_t)R * stride_f * 4));
      for (int n : {159744, 4 << 20}) {
        std::vector<int> h(n);
        for (auto& x : h) x = (int)(rng() % R);
        int* rows; CK(hipMalloc(&rows, (size_t)n * 4)); CK(hipMemcpy(rows, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        const int it = n > 1000000 ? 20 : 200;
        double a = time_gather<4, 1, false, false>(table, rows, n, stride_f, sink, it);
        double b = time_gather<4, 2, false, false>(table, rows, n, stride_f, sink, it);
        double c = stride_f >= 36 ? time_gather<4, 2, true, false>(table, rows, n, stride_f, sink, it) : 0;
        double d = time_gather<8, 1, false, false>(table, rows, n, stride_f, sink, it);
        double e = time_gather<4, 1, false, true>(table, rows, n, stride_f, sink, it);
        double f = time_gather<4, 2, false, true>(table, rows, n, stride_f, sink, it);
        double g = stride_f >= 36 ? time_gather<4, 2, true, true>(table, rows, n, stride_f, sink, it) : 0;
        printf("R=%ldM stride=%dB n=%d | read 64B %.2f us (%.0f Mrows/s) | 128B(4x2) %.2f us (%.0f) | 136B %.2f us (%.0f) | 128B(8x1) %.2f us (%.0f) || rmw 64B %.2f us (%.0f) | 128B %.2f us (%.0f) | 136B %.2f us (%.0f)\n",
               R >> 20, stride_f * 4, n, a, n / a, b, n / b, c, c > 0 ? n / c : 0, d, n / d, e, n / e, f, n / f, g, g > 0 ? n / g : 0);
        CK(hipFree(rows));
      }
      CK(hipFree(table));
    }
  }
  return 0;
}
