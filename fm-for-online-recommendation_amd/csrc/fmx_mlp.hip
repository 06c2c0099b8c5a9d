// fmx_mlp.hip -- the relu MLP of DeepFM / NFM / the ONN classes at mini-batch sizes (fp32 MFMA) and its C ABI:
// fmx_mlp_section, fmx_mlp_forward_batch, fmx_mlp_hedge_section, fmx_mlp_section_workspace_bytes.  Kernels: fmx_mlp_gemm.inc.
#include "fmx_common.h"

namespace {
#include "fmx_mlp_gemm.inc"
}  // namespace

extern "C" {

// the forward GEMM chain of the mini-batch MLP: acts[l] = relu(acts[l-1] . W_l^T + b_l)
static void mlp_big_forward(const fmx_mlp_t *mlp, const MlpBigWs &w, const float *bi, int32_t ld_bi, int32_t B, hipStream_t st) {
  const int L = mlp->n_layers, H = mlp->hidden, k = mlp->k;
  const size_t act = align_up((size_t)B * H * 4, 256) / 4;
  long long o = 0;
  for (int l = 0; l < L; ++l) {
    const int in = l == 0 ? k : H;
    GemmArgs g;
    g.mask = nullptr;
    g.rowadd = nullptr;
    g.ldmask = 0;
    g.c_split_stride = 0;
    g.zero_cols_to = 0;
    g.A = l == 0 ? bi : w.acts + (size_t)(l - 1) * act;
    g.lda = l == 0 ? ld_bi : H;
    g.Bm = mlp->params + o;
    g.ldb = in;
    g.C = w.acts + (size_t)l * act;
    g.ldc = H;
    g.bias = mlp->params + o + (long long)H * in;
    g.M = B;
    g.N = H;
    g.K = in;
    g.k_chunk = in;
    g.a_bytes = (unsigned)((size_t)B * g.lda * 4);
    g.b_bytes = (unsigned)((size_t)H * g.ldb * 4);
    g.vec = (size_t)B * g.lda * 4 < 0xFFFFFF00ull && gemm_vec_ok(g, 0, 1);
    launch_gemm<0, 1, EPI_BIAS_RELU>(g, 1, st);
    o += (long long)H * in + H;
  }
}

int fmx_mlp_forward_batch(const fmx_mlp_t *mlp, const float *bi, int32_t ld_bi, const float *base, int32_t B, void *workspace,
                          float *logit_out, float *layers_out, fmx_stream_t stream) {
  if (!mlp || !mlp->params || !bi || !base || !workspace || (!logit_out && !layers_out))
    return fail(FMX_ERR_ARG, "fmx_mlp_forward_batch: null argument");
  if (mlp->n_layers < 1 || mlp->n_layers > MLP_BIG_MAX_L || mlp->k < 1 || mlp->hidden < 1 || B < 1)
    return fail(FMX_ERR_UNSUPPORTED, "fmx_mlp_forward_batch: needs 1 <= layers <= %d, k >= 1, hidden >= 1, B >= 1", MLP_BIG_MAX_L);
  if (ld_bi < mlp->k) return fail(FMX_ERR_SHAPE, "fmx_mlp_forward_batch: ld_bi smaller than k");
  if (!aligned16(workspace)) return fail(FMX_ERR_ALIGN, "workspace must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const MlpBigWs w = mlp_big_carve(mlp, B, workspace);
  mlp_big_forward(mlp, w, bi, ld_bi, B, st);
  MlpOutArgs a;
  a.acts = w.acts;
  a.act_stride = align_up((size_t)B * mlp->hidden * 4, 256) / 4;
  a.base = base;
  a.logit = logit_out;
  a.layers = layers_out;
  a.B = B;
  a.hidden = mlp->hidden;
  a.n_layers = mlp->n_layers;
  hipLaunchKernelGGL(k_mlp_outputs, dim3((B + 3) / 4), dim3(256), 0, st, a);
  return check_launch("fmx_mlp_forward_batch");
}

int64_t fmx_mlp_section_workspace_bytes(const fmx_mlp_t *mlp, int32_t B) {
  if (!mlp || mlp->n_layers < 1 || mlp->n_layers > MLP_BIG_MAX_L || mlp->k < 1 || mlp->hidden < 1 || B < 1)
    return fail(FMX_ERR_ARG, "fmx_mlp_section_workspace_bytes: bad mlp / B");
  return (int64_t)mlp_big_carve(mlp, B, nullptr).bytes;
}

// blocks per layer of the reduction for workgroups of `threads` threads: one 16-byte group per thread, at most 256 blocks
extern "C++" int fmxd::mlp_reduce_blocks_per_layer(const MlpReduceArgs &a, int threads) {
  long long biggest = 0;
  for (int l = 0; l < a.n_layers; ++l) {
    const long long n = (long long)a.out_dim[l] * (a.ldp[l] >> 2);
    if (n > biggest) biggest = n;
  }
  const long long bx = (biggest + threads - 1) / threads;
  return (int)(bx > 256 ? 256 : bx < 1 ? 1 : bx);
}
extern "C++" void fmxd::mlp_launch_reduce(const MlpReduceArgs &a, hipStream_t st) {
  hipLaunchKernelGGL(k_mlp_reduce, dim3(mlp_reduce_blocks_per_layer(a, 256), a.n_layers), dim3(256), 0, st, a);
}

// the backward of the mini-batch MLP from dH_{L-1} (already in w.dH): the dgrad chain (with `rowadd_l` [L, B] added to
// layer l's dH before its mask when given: Hedge), dL/dbi into gbi_out when given, every layer's dW | db in one launch,
// then the fixed-order reduction into `grads` (+ optional SGD, + the mean of w.loss_b into loss_out when given)
// skip_dgrad: k_mlp_chain has already produced every dH_l and gbi; only the weight gradients and their reduction remain
static void mlp_big_backward(const fmx_mlp_t *mlp, const MlpBigWs &w, const float *bi, int32_t ld_bi, int32_t B,
                             const float *rowadd_l, float *gbi_out, int32_t ld_gbi, float *grads, float lr_apply,
                             float *loss_out, float inv_b, hipStream_t st, bool skip_dgrad = false, MlpReduceArgs *deferred = nullptr) {
  const int L = mlp->n_layers, H = mlp->hidden, k = mlp->k;
  const size_t act = align_up((size_t)B * H * 4, 256) / 4;
  const float *Wl[MLP_BIG_MAX_L];
  long long off[MLP_BIG_MAX_L];
  {
    long long o = 0;
    for (int l = 0; l < L; ++l) {
      const int in = l == 0 ? k : H;
      off[l] = o;
      Wl[l] = mlp->params + o;
      o += (long long)H * in + H;
    }
  }
  auto base_args = [] {
    GemmArgs g;
    g.bias = nullptr;
    g.mask = nullptr;
    g.rowadd = nullptr;
    g.ldmask = 0;
    g.c_split_stride = 0;
    g.zero_cols_to = 0;
    g.stamps = nullptr;
    return g;
  };
  int splits[MLP_BIG_MAX_L] = {0};
  WgradBatch wb;
  wb.n = 0;
  int wgx = 1;
  // the streaming weight-gradient kernel (k_mlp_wgrad_stream): 16-byte loads of dH_l (and of H_{l-1} where the layer is at
  // least 64 wide), one split count for every layer chosen so that the 64 x 64 blocks of the wide layers fill the chip once
  WgradStream sb;
  sb.n = 0;
  sb.per = 1;
  bool stream = H % 4 == 0 && aligned16(w.dH) && aligned16(w.acts) && (size_t)B * H * 4 < 0x7FFFFF00ull;
  int stream_splits = 1;
  {
    int heavy = 0;
    for (int l = 0; l < L; ++l) {
      const int in = l == 0 ? k : H;
      if (in % 4 != 0 || (in < 64 && in > 48)) stream = false;
      if (l == 0 && in >= 64 && (ld_bi % 4 != 0 || !aligned16(bi))) stream = false;
      if (in >= 64) heavy += ((H + 63) / 64) * ((in + 63) / 64);
    }
    if (heavy < 1) heavy = 1;
    stream_splits = 256 / heavy;
    if (stream_splits > w.n_split) stream_splits = w.n_split;
    if (stream_splits < 1) stream_splits = 1;
    sb.B = B;
  }
  auto stream_rows = [&](int in) {  // rows per split: the narrow layers (a quarter of the MFMAs per step) in four times the splits
    int ns = in >= 64 ? stream_splits : 2 * stream_splits;
    if (ns > w.n_split) ns = w.n_split;
    return ((B + ns - 1) / ns + 15) / 16 * 16;
  };
  for (int l = L - 1; l >= 0; --l) {
    const int in = l == 0 ? k : H;
    float *cur = w.dH + (size_t)l * act;
    const float *prev = l == 0 ? bi : w.acts + (size_t)(l - 1) * act;
    const int ldprev = l == 0 ? ld_bi : H;
    {  // dW_l = dH_l^T . H_{l-1}, split over the batch
      GemmArgs g = base_args();
      g.A = cur;
      g.lda = H;
      g.Bm = prev;
      g.ldb = ldprev;
      g.C = w.parts[l];
      g.ldc = w.ldp[l];
      g.M = H;
      g.N = in;
      g.K = B;
      // splits of the batch: 16 for a 256 x 256 layer (528 workgroups over the launch), at most B / 128
      const int tiles = ((in + G_BN - 1) / G_BN) * ((H + G_BM - 1) / G_BM);
      int n_split = 256 / tiles;  // (192 .. 384: the same within 0.5 us; 128: +6 us, 512 / 1024: +4 us -- tools/mlp_section_times.py)
      if (n_split > w.n_split) n_split = w.n_split;
      if (n_split < 1) n_split = 1;
      g.k_chunk = ((B + n_split - 1) / n_split + G_BK_WGRAD - 1) / G_BK_WGRAD * G_BK_WGRAD;
      n_split = (B + g.k_chunk - 1) / g.k_chunk;
      splits[l] = n_split;
      g.c_split_stride = (long long)H * w.ldp[l];
      g.a_bytes = (unsigned)((size_t)B * g.lda * 4);
      g.b_bytes = (unsigned)((size_t)B * g.ldb * 4);
      g.vec = (size_t)B * (g.lda > g.ldb ? g.lda : g.ldb) * 4 < 0xFFFFFF00ull && gemm_vec_ok(g, 1, 0);
      if (stream) {
        WgradLayer &sl = sb.g[sb.n];
        sl.dH = cur;
        sl.Hp = prev;
        sl.P = w.parts[l];
        sl.split_stride = (long long)H * w.ldp[l];
        sl.M = H;
        sl.N = in;
        sl.lda = H;
        sl.ldb = ldprev;
        sl.ldp = w.ldp[l];
        sl.tiles_m = (H + 63) / 64;
        sl.tiles_n = in >= 64 ? (in + 63) / 64 : 1;
        sl.rows_per_split = stream_rows(in);
        splits[l] = (B + sl.rows_per_split - 1) / sl.rows_per_split;
        sb.z_end[sb.n] = (sb.n ? sb.z_end[sb.n - 1] : 0) + splits[l];
        if (sl.tiles_m * sl.tiles_n > sb.per) sb.per = sl.tiles_m * sl.tiles_n;
        ++sb.n;
      }
      wb.g[wb.n] = g;                                  // launched together with the other layers' after the dgrad chain
      wb.z_end[wb.n] = (wb.n ? wb.z_end[wb.n - 1] : 0) + n_split;
      wb.bias_col[wb.n] = in;
      if ((in + G_BN - 1) / G_BN > wgx) wgx = (in + G_BN - 1) / G_BN;
      ++wb.n;
    }
    if (skip_dgrad) continue;
    if (l == 0 && !gbi_out) continue;  // nothing below the first layer wants a gradient (Hedge leaves the tables alone)
    GemmArgs g = base_args();
    g.A = cur;
    g.lda = H;
    g.Bm = Wl[l];
    g.ldb = in;
    g.M = B;
    g.N = in;
    g.K = H;
    g.k_chunk = H;
    g.a_bytes = (unsigned)((size_t)B * g.lda * 4);
    g.b_bytes = (unsigned)((size_t)H * g.ldb * 4);
    g.vec = (size_t)B * g.lda * 4 < 0xFFFFFF00ull && gemm_vec_ok(g, 0, 0);
    if (l > 0) {  // dH_{l-1} = (dH_l . W_l [+ the layer's own output gradient]) * (H_{l-1} > 0)
      g.C = w.dH + (size_t)(l - 1) * act;
      g.ldc = H;
      g.mask = prev;
      g.ldmask = H;
      g.rowadd = rowadd_l ? rowadd_l + (size_t)(l - 1) * B : nullptr;
      launch_gemm<0, 0, EPI_MASK>(g, 1, st);
    } else {  // dL/dbi through the MLP, padding columns zeroed
      g.C = gbi_out;
      g.ldc = ld_gbi;
      g.zero_cols_to = ld_gbi;
      launch_gemm<0, 0, EPI_NONE>(g, 1, st);
    }
  }
  if (stream) {  // ---- every layer's dW_l | db_l in one launch, operands streamed into registers ----
    static bool raised_s = false;
    if (!raised_s) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_mlp_wgrad_stream), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WS_LDS_BYTES);
      raised_s = true;
    }
    const int zs = sb.z_end[sb.n - 1];
    sb.stamps = tune().mlp_chain == 3 ? reinterpret_cast<unsigned long long *>(w.dzl) : nullptr;  // debug: tools/mlp_wgrad_stamps.py
    hipLaunchKernelGGL(k_mlp_wgrad_stream, dim3(8 * ((zs + 7) / 8) * sb.per), dim3(256), WS_LDS_BYTES, st, sb);
  } else {  // ---- the same as 64 x 64 x 32 tiles staged through LDS (any shape) ----
    static bool raised = false;
    if (!raised) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_mlp_wgrad), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)g_lds_bytes(G_BK_WGRAD));
      raised = true;
    }
    wb.tx = wgx;
    wb.stamps = tune().mlp_chain == 3 ? reinterpret_cast<unsigned long long *>(w.dzl) : nullptr;  // debug: tools/mlp_wgrad_stamps.py
    wb.ty = (H + G_BM - 1) / G_BM;
    const int per = wb.tx * wb.ty + wb.ty, zs = wb.z_end[wb.n - 1];
    hipLaunchKernelGGL(k_mlp_wgrad, dim3(8 * ((zs + 7) / 8) * per), dim3(256), g_lds_bytes(G_BK_WGRAD), st, wb);
  }
  MlpReduceArgs a;
  for (int l = 0; l < MLP_BIG_MAX_L; ++l) {
    const int in = l == 0 ? k : H;
    a.parts[l] = l < L ? w.parts[l] : nullptr;
    a.out_dim[l] = H;
    a.in_dim[l] = in;
    a.ldp[l] = l < L ? w.ldp[l] : 0;
    a.grad_off[l] = l < L ? off[l] : 0;
    a.n_split[l] = splits[l];
  }
  a.grads = grads;
  a.params = mlp->params;
  a.lr = lr_apply;
  a.n_layers = L;
  a.loss_b = w.loss_b;
  a.loss_out = loss_out;
  a.B = B;
  a.inv_b = inv_b;
  if (deferred) {  // the caller carries the reduction's blocks in another launch (or calls mlp_launch_reduce)
    *deferred = a;
    return;
  }
  mlp_launch_reduce(a, st);
}

static int mlp_big_check(const fmx_mlp_t *mlp, int32_t B, const void *workspace, const char *who) {
  if (!mlp || !mlp->params || !workspace) return fail(FMX_ERR_ARG, "%s: null argument", who);
  if (mlp->n_layers < 1 || mlp->n_layers > MLP_BIG_MAX_L || mlp->k < 1 || mlp->hidden < 1 || B < 1)
    return fail(FMX_ERR_UNSUPPORTED, "%s: needs 1 <= layers <= %d, k >= 1, hidden >= 1, B >= 1", who, MLP_BIG_MAX_L);
  if (!aligned16(workspace)) return fail(FMX_ERR_ALIGN, "workspace must be 16-byte aligned");
  return FMX_OK;
}

int fmx_mlp_section(const fmx_mlp_t *mlp, int32_t loss_kind, const float *bi, int32_t ld_bi, const float *base,
                    const float *y, int32_t B, float inv_b, void *workspace, float *logit_out, float *dz_out,
                    float *gbi_out, int32_t ld_gbi, float *grads, float lr_apply, float *loss_out, fmx_stream_t stream) {
  return mlp_section_deferred_reduce(mlp, loss_kind, bi, ld_bi, base, y, B, inv_b, workspace, logit_out, dz_out, gbi_out, ld_gbi, grads, lr_apply,
                                     loss_out, static_cast<hipStream_t>(stream), nullptr);
}

extern "C++" int fmxd::mlp_section_deferred_reduce(const fmx_mlp_t *mlp, int32_t loss_kind, const float *bi, int32_t ld_bi, const float *base,
                                                   const float *y, int32_t B, float inv_b, void *workspace, float *logit_out, float *dz_out,
                                                   float *gbi_out, int32_t ld_gbi, float *grads, float lr_apply, float *loss_out, hipStream_t st,
                                                   MlpReduceArgs *deferred) {
  if (int rc = mlp_big_check(mlp, B, workspace, "fmx_mlp_section")) return rc;
  if (!bi || !base || !y || !dz_out || !gbi_out || !grads) return fail(FMX_ERR_ARG, "fmx_mlp_section: null argument");
  if (ld_bi < mlp->k || ld_gbi < mlp->k) return fail(FMX_ERR_SHAPE, "fmx_mlp_section: ld_bi / ld_gbi smaller than k");
  if (loss_kind != FMX_LOSS_BCE_LOGITS && loss_kind != FMX_LOSS_BCE_SIGMOID) return fail(FMX_ERR_ARG, "fmx_mlp_section needs a loss");
  const MlpBigWs w = mlp_big_carve(mlp, B, workspace);
  const int L = mlp->n_layers, H = mlp->hidden;
  const size_t act = align_up((size_t)B * H * 4, 256) / 4;
  if (tune().mlp_chain && chain_eligible(mlp)) {
    // forward, loss and the dgrad chain in ONE launch (k_mlp_chain), then the weight gradients and their reduction
    ChainArgs c;
    c.params = mlp->params;
    c.bi = bi;
    c.base = base;
    c.y = y;
    c.acts = w.acts;
    c.dH = w.dH;
    c.act_stride = act;
    c.logit_out = logit_out;
    c.dz_out = dz_out;
    c.loss_b = w.loss_b;
    c.gbi_out = gbi_out;
    long long o = 0;
    for (int l = 0; l < L; ++l) {
      const int in = l == 0 ? mlp->k : H;
      c.w_off[l] = o;
      c.b_off[l] = o + (long long)H * in;
      o += (long long)H * in + H;
    }
    c.B = B;
    c.k = mlp->k;
    c.H = H;
    c.L = L;
    c.ld_bi = ld_bi;
    c.ld_gbi = ld_gbi;
    c.loss_kind = loss_kind;
    c.inv_b = inv_b;
    c.stamps = tune().mlp_chain == 2 ? reinterpret_cast<unsigned long long *>(w.loss_lb) : nullptr;  // debug: tools/mlp_chain_stamps.py
#ifdef FMX_MLP_EXPERIMENTS  // diagnostic build only (tools/corun_experiment.py): FMX_EXP_ONLY_WGRAD=1 launches the gradient kernel alone
    static int only_wgrad = getenv("FMX_EXP_ONLY_WGRAD") ? atoi(getenv("FMX_EXP_ONLY_WGRAD")) : 0;
    if (only_wgrad) {
      MlpReduceArgs unused;
      mlp_big_backward(mlp, w, bi, ld_bi, B, nullptr, gbi_out, ld_gbi, grads, lr_apply, loss_out, inv_b, st, true, &unused);
      return check_launch("fmx_mlp_section (gradient kernel only)");
    }
#endif
    hipLaunchKernelGGL(k_mlp_chain, dim3((B + CH_R - 1) / CH_R), dim3(256), chain_lds_bytes(H), st, c);
    mlp_big_backward(mlp, w, bi, ld_bi, B, nullptr, gbi_out, ld_gbi, grads, lr_apply, loss_out, inv_b, st, true, deferred);
    return check_launch("fmx_mlp_section (k_mlp_chain)");
  }
  mlp_big_forward(mlp, w, bi, ld_bi, B, st);
  {  // ---- loss, dL/dlogit, dH_L ----
    MlpLossArgs a;
    a.H = w.acts + (size_t)(L - 1) * act;
    a.dH = w.dH + (size_t)(L - 1) * act;
    a.base = base;
    a.y = y;
    a.out = logit_out;
    a.dz = dz_out;
    a.loss_b = w.loss_b;
    a.B = B;
    a.hidden = H;
    a.ldh = H;
    a.loss_kind = loss_kind;
    a.inv_b = inv_b;
    hipLaunchKernelGGL(k_mlp_loss, dim3((B + 3) / 4), dim3(256), 0, st, a);
  }
  mlp_big_backward(mlp, w, bi, ld_bi, B, nullptr, gbi_out, ld_gbi, grads, lr_apply, loss_out, inv_b, st, false, deferred);
  return check_launch("fmx_mlp_section");
}

int fmx_mlp_hedge_section(const fmx_mlp_t *mlp, float lr, float hedge_b, float hedge_s, float *alpha, const float *bi,
                          int32_t ld_bi, const float *base, const float *y, int32_t B, void *workspace, float *grads,
                          float *losses_out, fmx_stream_t stream) {
  if (int rc = mlp_big_check(mlp, B, workspace, "fmx_mlp_hedge_section")) return rc;
  if (!alpha || !bi || !base || !y || !grads) return fail(FMX_ERR_ARG, "fmx_mlp_hedge_section: null argument");
  if (ld_bi < mlp->k) return fail(FMX_ERR_SHAPE, "fmx_mlp_hedge_section: ld_bi smaller than k");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const MlpBigWs w = mlp_big_carve(mlp, B, workspace);
  const int L = mlp->n_layers, H = mlp->hidden;
  const size_t act = align_up((size_t)B * H * 4, 256) / 4;
  const float inv_b = 1.0f / (float)B;  // nn.BCELoss: the mean over the batch
  mlp_big_forward(mlp, w, bi, ld_bi, B, st);
  {
    MlpHedgeLossArgs a;
    a.acts = w.acts;
    a.act_stride = act;
    a.dH_top = w.dH + (size_t)(L - 1) * act;
    a.base = base;
    a.y = y;
    a.alpha = alpha;
    a.dzl = w.dzl;
    a.loss_lb = w.loss_lb;
    a.B = B;
    a.hidden = H;
    a.n_layers = L;
    a.inv_b = inv_b;
    hipLaunchKernelGGL(k_mlp_hedge_loss, dim3((B + 3) / 4), dim3(256), 0, st, a);
  }
  // hidden layers: theta -= lr * sum_i alpha_i d loss_i / d theta (one backward pass: d loss_i / d layer_j = 0 for j > i)
  mlp_big_backward(mlp, w, bi, ld_bi, B, w.dzl, nullptr, 0, grads, lr, nullptr, inv_b, st);
  {
    MlpHedgeAlphaArgs a;
    a.loss_lb = w.loss_lb;
    a.alpha = alpha;
    a.losses_out = losses_out;
    a.B = B;
    a.n_layers = L;
    a.inv_b = inv_b;
    a.hedge_b = hedge_b;
    a.hedge_s = hedge_s;
    hipLaunchKernelGGL(k_mlp_hedge_alpha, dim3(1), dim3(256), 0, st, a);
  }
  return check_launch("fmx_mlp_hedge_section");
}

}  // extern "C"
