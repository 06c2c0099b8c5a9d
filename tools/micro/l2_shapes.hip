// What request SHAPE does to a compute unit's read rate from L2: every workgroup (one per CU, 4 waves) reads the same
// 256 KB matrix (256 rows of 1 KB) over and over, 16-byte loads, DEPTH groups of 8 loads in flight per wave, and only the
// assignment of addresses to lanes and instructions differs:
//   0 contiguous   an instruction = 1 KB contiguous
//   1 dgrad        an instruction = 4 rows x 256 contiguous bytes (lane (m, kq): row 4 c + kq, bytes 256 w + 16 m)
//   2 fragment     an instruction = 16 rows x 64 bytes (lane (m, kq): row m of the tile, bytes 16 kq), the two halves of a
//                  line in consecutive instructions
//   3 line8        an instruction = 8 rows x 128 bytes, lane L: row L % 8, piece L / 8 (a 16-lane group touches 8 lines)
//   4 line16g      the same addresses, lane L: row L / 8, piece L % 8 (a 16-lane group = 2 whole lines)
//   5 line16g2k    as 4 with the rows of an instruction 2 KB apart (rows 0, 2, 4 .. then 1, 3, ..)
//   6 line4        an instruction = 16 rows x 64 bytes, lane L: row L / 4, piece L % 4 (the fragment's addresses, four
//                  consecutive lanes contiguous)
//   7 line2        an instruction = 32 rows x 32 bytes, lane L: row L / 2, piece L % 2
// Build: hipcc --offload-arch=gfx950 -O3 -o l2_shapes l2_shapes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int STEPS = 8;   // steps of 8 loads per wave and pass: 4 waves x 8 steps x 8 KB = 256 KB
constexpr int DEPTH = 3;   // steps in flight

template <int SHAPE>
__device__ __forceinline__ unsigned offset_of(int w, int L, int step, int j) {
  const int m = L & 15, kq = L >> 4;
  if (SHAPE == 0) return (unsigned)((((step * 4 + w) * 8 + j) * 1024) + L * 16);
  if (SHAPE == 1) {  // chunk c = 2 step + j / 4, s = j % 4
    const int c = 2 * step + (j >> 2), s = j & 3;
    return (unsigned)((16 * c + 4 * kq + s) * 1024 + 256 * w + 16 * m);
  }
  if (SHAPE == 2) {  // tile i = j / 2, half h = j % 2, unit = step
    const int i = j >> 1, h = j & 1;
    return (unsigned)((64 * w + 16 * i + m) * 1024 + step * 128 + h * 64 + 16 * kq);
  }
  if (SHAPE == 3) {
    const int i = j >> 1, h = j & 1;
    return (unsigned)((64 * w + 16 * i + 8 * h + (L & 7)) * 1024 + step * 128 + (L >> 3) * 16);
  }
  if (SHAPE == 4) {
    const int i = j >> 1, h = j & 1;
    return (unsigned)((64 * w + 16 * i + 8 * h + (L >> 3)) * 1024 + step * 128 + (L & 7) * 16);
  }
  if (SHAPE == 5) {
    const int i = j >> 1, h = j & 1;
    return (unsigned)((64 * w + 16 * i + h + 2 * (L >> 3)) * 1024 + step * 128 + (L & 7) * 16);
  }
  if (SHAPE == 6) {  // 16 rows x 64 bytes, four consecutive lanes contiguous
    const int i = j >> 1, h = j & 1;
    return (unsigned)((64 * w + 16 * i + (L >> 2)) * 1024 + step * 128 + h * 64 + (L & 3) * 16);
  }
  {  // 7: 32 rows x 32 bytes, two consecutive lanes contiguous
    const int i = j >> 2, h = j & 3;
    return (unsigned)((64 * w + 32 * i + (L >> 1)) * 1024 + step * 128 + h * 32 + (L & 1) * 16);
  }
}

template <int SHAPE>
__global__ __launch_bounds__(256) void k(const float *buf, int passes, unsigned long long *out, float *sink) {
  const int w = threadIdx.x >> 6, L = threadIdx.x & 63;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(buf), 0, 256 * 1024, 0x00020000);
  unsigned voff[STEPS][8];
#pragma unroll
  for (int s = 0; s < STEPS; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) voff[s][j] = offset_of<SHAPE>(w, L, s, j);
  float acc = 0.f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int pass = 0; pass < passes; ++pass) {
    u32x4 r[STEPS][8];
#pragma unroll
    for (int s = 0; s < DEPTH; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) r[s][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[s][j], 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      if (s + DEPTH < STEPS) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[s + DEPTH][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[s + DEPTH][j], 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += __uint_as_float(r[s][j].x) + __uint_as_float(r[s][j].w);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (acc == 12345.f) *sink = acc;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int SHAPE>
void run(const float *buf, const char *name, int blocks) {
  const int passes = 64;
  unsigned long long *d;
  float *sink;
  (void)hipMalloc(&d, blocks * 8);
  (void)hipMalloc(&sink, 4);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, buf, passes, d, sink);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks);
  (void)hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double us = h[blocks / 2] / 100.0 / passes;
  printf("%-12s %3d workgroups: %6.2f us per 256 KB pass per CU = %6.1f GB/s per CU, %6.2f TB/s over the launch\n", name, blocks, us,
         262144.0 / us / 1e3, 262144.0 / us / 1e3 * blocks / 1e3);
  (void)hipFree(d);
  (void)hipFree(sink);
}

int main() {
  float *buf;
  (void)hipMalloc(&buf, 1 << 20);
  (void)hipMemset(buf, 0, 1 << 20);
  for (int blocks : {1, 256}) {
    run<0>(buf, "contiguous", blocks);
    run<1>(buf, "dgrad", blocks);
    run<2>(buf, "fragment", blocks);
    run<3>(buf, "line8", blocks);
    run<4>(buf, "line16g", blocks);
    run<5>(buf, "line16g2k", blocks);
    run<6>(buf, "line4", blocks);
    run<7>(buf, "line2", blocks);
  }
  return 0;
}
