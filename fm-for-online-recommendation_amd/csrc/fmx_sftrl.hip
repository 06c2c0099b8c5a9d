// fmx_sftrl.hip -- the sketched-FTRL family on the device and its C ABI: fmx_sftrl_run, fmx_sftrl_grid.
// Kernel: fmx_sftrl.inc.
#include "fmx_common.h"

namespace {
#include "fmx_sftrl.inc"
}  // namespace

extern "C" {

int fmx_sftrl_run(const double *X, const double *y, int32_t N, int32_t D, int32_t d, int32_t m, double eta, double thres,
                  int32_t task, double *BP, double *BN, int32_t *counts, double *w, double *g_w, double *pred_out,
                  int32_t *status, fmx_stream_t stream) {
  if (!X || !y || !BP || !BN || !counts || !pred_out || !status) return fail(FMX_ERR_ARG, "fmx_sftrl_run: null argument");
  if ((w == nullptr) != (g_w == nullptr)) return fail(FMX_ERR_ARG, "fmx_sftrl_run: w and g_w go together");
  if (N < 0 || D < 1 || d < 1 || d > D || m < 1) return fail(FMX_ERR_ARG, "fmx_sftrl_run: bad sizes");
  if (task != 0 && task != 1) return fail(FMX_ERR_ARG, "fmx_sftrl_run: task must be 0 (cls) or 1 (reg)");
  if (d > SF_MAX_D || 2 * m > SF_MAX_C || D > 64)
    return fail(FMX_ERR_UNSUPPORTED, "fmx_sftrl_run: needs sketch dim <= %d, 2 m <= %d, features <= 64 (got %d, %d, %d)", SF_MAX_D,
                SF_MAX_C, d, 2 * m, D);
  if (N == 0) return FMX_OK;
  SftrlArgs a;
  a.X = X;
  a.y = y;
  a.BP = BP;
  a.BN = BN;
  a.counts = counts;
  a.w = w;
  a.g_w = g_w;
  a.pred = pred_out;
  a.status = status;
  a.eta = eta;
  a.thres = thres;
  a.N = N;
  a.D = D;
  a.d = d;
  a.m = m;
  a.cls = task == 0;
  a.ms = nullptr;
  a.etas = nullptr;
  a.B_stride = a.w_stride = a.pred_stride = 0;
  const size_t lds = ((size_t)2 * d * 2 * m + 2 * (size_t)d * d + d + 48 + D) * sizeof(double) + (size_t)d * sizeof(int) + 16;
  static bool raised = false;
  if (!raised) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_sftrl_online), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    raised = true;
  }
  hipLaunchKernelGGL(k_sftrl_online, dim3(1), dim3(64), lds, static_cast<hipStream_t>(stream), a);
  return check_launch("k_sftrl_online");
}

int fmx_sftrl_grid(const double *X, const double *y, int32_t N, int32_t D, int32_t d, int32_t n_settings, const int32_t *ms,
                   const double *etas, int32_t m_max, double thres, int32_t task, double *BP, double *BN, int32_t *counts, double *w,
                   double *g_w, double *pred_out, int32_t *status, fmx_stream_t stream) {
  if (!X || !y || !ms || !etas || !BP || !BN || !counts || !pred_out || !status) return fail(FMX_ERR_ARG, "fmx_sftrl_grid: null argument");
  if ((w == nullptr) != (g_w == nullptr)) return fail(FMX_ERR_ARG, "fmx_sftrl_grid: w and g_w go together");
  if (N < 0 || D < 1 || d < 1 || d > D || m_max < 1 || n_settings < 0) return fail(FMX_ERR_ARG, "fmx_sftrl_grid: bad sizes");
  if (task != 0 && task != 1) return fail(FMX_ERR_ARG, "fmx_sftrl_grid: task must be 0 (cls) or 1 (reg)");
  if (d > SF_MAX_D || 2 * m_max > SF_MAX_C || D > 64)
    return fail(FMX_ERR_UNSUPPORTED, "fmx_sftrl_grid: needs sketch dim <= %d, 2 m <= %d, features <= 64 (got %d, %d, %d)", SF_MAX_D,
                SF_MAX_C, d, 2 * m_max, D);
  if (N == 0 || n_settings == 0) return FMX_OK;
  SftrlArgs a;
  a.X = X;
  a.y = y;
  a.BP = BP;
  a.BN = BN;
  a.counts = counts;
  a.w = w;
  a.g_w = g_w;
  a.pred = pred_out;
  a.status = status;
  a.eta = 0.0;
  a.thres = thres;
  a.N = N;
  a.D = D;
  a.d = d;
  a.m = m_max;
  a.cls = task == 0;
  a.ms = ms;
  a.etas = etas;
  a.B_stride = (long long)d * 2 * m_max;
  a.w_stride = D;
  a.pred_stride = N;
  const size_t lds = ((size_t)2 * d * 2 * m_max + 2 * (size_t)d * d + d + 48 + D) * sizeof(double) + (size_t)d * sizeof(int) + 16;
  static bool raised = false;
  if (!raised) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_sftrl_online), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    raised = true;
  }
  hipLaunchKernelGGL(k_sftrl_online, dim3(n_settings), dim3(64), lds, static_cast<hipStream_t>(stream), a);
  return check_launch("k_sftrl_online (grid)");
}

}  // extern "C"
