// sort_bench.hip -- the occurrence sort kernels of fmx_sort.inc on their own: identical lists and time per launch, one batch
// and eight batches per launch, Criteo vocabulary, B = 4096 (and a floor: the index loads + list stores alone).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -o tools/micro/sort_bench tools/micro/sort_bench.hip
#include "../../fm-for-online-recommendation_amd/csrc/fmx_common.h"
#include <algorithm>
#include <random>
#include <vector>
namespace {
#include "../../fm-for-online-recommendation_amd/csrc/fmx_sort.inc"

// floor: what the loads of one field's column and the stores of its list cost with nothing between
__global__ __launch_bounds__(1024) void k_floor(SortArgs a) {
  int j, f;
  if (a.n_batches >= 8) { if (!xcd_unit(a.n_batches, a.F, j, f)) return; }
  else { j = blockIdx.x / a.F; f = blockIdx.x - j * a.F; }
  a.idx += (size_t)((a.pool_first + j) % a.n_pool) * a.pool_stride;
  a.sorted += (size_t)j * a.sorted_stride;
  uint32_t v[4];
  load_composites<4>(a, f, threadIdx.x * 4, v);
  uint32_t *dst = a.sorted + (size_t)f * a.Bp;
  for (int r = 0; r < 4; ++r) dst[threadIdx.x * 4 + r] = v[r];
}
}  // namespace
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char **argv) {
  const int sizes[39] = {1457, 555, 175446, 129683, 305, 19, 11887, 632, 3, 41738, 5170, 172761, 3170, 27, 11356, 165602, 10, 4641, 2030, 4, 176373, 18, 15, 57903, 86, 44549,
                         9, 63, 79, 82, 100, 104, 113, 126, 148, 224, 32, 51, 57};
  const int F = 39, B = argc > 1 ? atoi(argv[1]) : 4096, NB = 8;
  int Bp = 64; while (Bp < B) Bp <<= 1;
  int bbits = 0; while ((1 << bbits) < Bp) ++bbits;
  std::vector<int64_t> foff(F + 1, 0);
  for (int f = 0; f < F; ++f) foff[f + 1] = foff[f] + sizes[f];
  std::mt19937 rng(7);
  std::vector<int32_t> idx((size_t)NB * B * F);
  for (int j = 0; j < NB; ++j)
    for (int b = 0; b < B; ++b)
      for (int f = 0; f < F; ++f) idx[((size_t)j * B + b) * F + f] = (int32_t)(rng() % (uint32_t)sizes[f]);
  int32_t *d_idx; int64_t *d_foff; uint32_t *d_a, *d_b, *d_starts; int32_t *d_err;
  const size_t per = (size_t)F * Bp;
  CK(hipMalloc(&d_idx, idx.size() * 4)); CK(hipMalloc(&d_foff, (F + 1) * 8)); CK(hipMalloc(&d_a, NB * per * 4)); CK(hipMalloc(&d_b, NB * per * 4));
  CK(hipMalloc(&d_starts, (size_t)NB * F * 2048 * 4)); CK(hipMalloc(&d_err, 4)); CK(hipMemset(d_err, 0, 4));
  CK(hipMemcpy(d_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_foff, foff.data(), (F + 1) * 8, hipMemcpyHostToDevice));
  SortArgs a{};
  a.n_pool = NB; a.pool_first = 0; a.pool_stride = (int64_t)B * F; a.sorted_stride = (int64_t)per;
  a.idx = d_idx; a.foff = d_foff; a.soff = d_foff; a.cols = nullptr; a.error = d_err; a.B = B; a.F = F; a.Bp = Bp; a.bbits = bbits; a.Fi = F;
  const size_t rlds = radix_lds_bytes(Bp, RADIX_THREADS);
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sort_radix), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rlds));
  auto grid = [&](int nb) { return nb >= 8 ? 8 * F * ((nb + 7) / 8) : F * nb; };
  auto run_bitonic = [&](int nb, uint32_t *dst) { SortArgs x = a; x.n_batches = nb; x.sorted = dst; hipLaunchKernelGGL((k_sort_occ<4>), dim3(grid(nb)), dim3(Bp / 4), Bp * 4, 0, x); };
  auto run_radix = [&](int nb, uint32_t *dst) { SortArgs x = a; x.n_batches = nb; x.sorted = dst; hipLaunchKernelGGL(k_sort_radix, dim3(grid(nb)), dim3(RADIX_THREADS < Bp ? RADIX_THREADS : Bp), rlds, 0, x, d_starts, 2048); };
  const size_t hlds = radix_small_lds_bytes(Bp, Bp / 4) > (size_t)Bp * 4 ? radix_small_lds_bytes(Bp, Bp / 4) : (size_t)Bp * 4;
  auto run_hybrid = [&](int nb, uint32_t *dst) { SortArgs x = a; x.n_batches = nb; x.sorted = dst; x.small_bits = RADIX_SMALL_BITS; hipLaunchKernelGGL((k_sort_occ<4>), dim3(grid(nb)), dim3(Bp / 4), hlds, 0, x); };
  auto run_floor = [&](int nb, uint32_t *dst) { SortArgs x = a; x.n_batches = nb; x.sorted = dst; hipLaunchKernelGGL(k_floor, dim3(grid(nb)), dim3(Bp / 4), 0, 0, x); };
  // ---- identical lists ----
  run_bitonic(NB, d_a); run_hybrid(NB, d_b); CK(hipDeviceSynchronize());
  {
    std::vector<uint32_t> ha(NB * per), hb(NB * per);
    CK(hipMemcpy(ha.data(), d_a, ha.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), d_b, hb.size() * 4, hipMemcpyDeviceToHost));
    size_t diff = 0; for (size_t i = 0; i < ha.size(); ++i) diff += ha[i] != hb[i];
    printf("hybrid (k_sort_occ, small fields by one counting pass) vs bitonic: %zu words differ\n", diff);
  }
  run_bitonic(NB, d_a); run_radix(NB, d_b); CK(hipDeviceSynchronize());
  std::vector<uint32_t> ha(NB * per), hb(NB * per);
  CK(hipMemcpy(ha.data(), d_a, ha.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), d_b, hb.size() * 4, hipMemcpyDeviceToHost));
  size_t diff = 0; for (size_t i = 0; i < ha.size(); ++i) diff += ha[i] != hb[i];
  // host check of batch 3
  size_t hdiff = 0;
  for (int f = 0; f < F; ++f) {
    std::vector<uint32_t> ref(Bp, 0xFFFFFFFFu);
    for (int b = 0; b < B; ++b) ref[b] = ((uint32_t)idx[((size_t)3 * B + b) * F + f] << bbits) | (uint32_t)b;
    std::sort(ref.begin(), ref.end());
    for (int i = 0; i < Bp; ++i) hdiff += ref[i] != hb[3 * per + (size_t)f * Bp + i];
  }
  std::vector<uint32_t> hs((size_t)NB * F * 2048);
  CK(hipMemcpy(hs.data(), d_starts, hs.size() * 4, hipMemcpyDeviceToHost));
  size_t sdiff = 0;
  for (int f = 0; f < F; ++f) {
    if (sizes[f] >= 2048) continue;
    const uint32_t *st = &hs[((size_t)3 * F + f) * 2048], *lst = &hb[3 * per + (size_t)f * Bp];
    for (int r = 0; r <= sizes[f]; ++r) {  // starts[r] = first position whose key >= r
      uint32_t want = 0; while (want < (uint32_t)Bp && (lst[want] >> bbits) < (uint32_t)r && lst[want] != 0xFFFFFFFFu) ++want;
      if (r == sizes[f]) { want = 0; while (want < (uint32_t)Bp && lst[want] != 0xFFFFFFFFu) ++want; }
      sdiff += st[r] != want;
    }
  }
  printf("B=%d: radix vs bitonic: %zu words differ; radix vs std::sort (batch 3): %zu; run table: %zu wrong; lds %zu B, %d threads\n", B, diff, hdiff, sdiff, rlds, RADIX_THREADS);
  // ---- times ----
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time_it = [&](const char *what, auto fn, int nb) {
    for (int i = 0; i < 5; ++i) fn(nb, d_a);
    CK(hipDeviceSynchronize());
    const int reps = 50;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) fn(nb, d_a);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  %-10s %d batch(es) per launch: %7.2f us per launch\n", what, nb, ms * 1e3 / reps);
  };
  for (int nb : {1, 2, 4, 8}) { time_it("bitonic", run_bitonic, nb); time_it("radix", run_radix, nb); time_it("hybrid", run_hybrid, nb); time_it("floor", run_floor, nb); }
  return 0;
}
