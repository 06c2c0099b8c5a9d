// fmx_common.h -- what the translation units of libfmx.so share: the error plumbing and the tuning switches (one copy, defined
// in fmx_kernels.hip), and the device helpers every kernel file uses (per-unit copies in an anonymous namespace).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>

#include "fmx.h"

// No floating-point contraction: every kernel evaluates the expressions as written (one rounding per operation), so two
// kernels that state the same arithmetic -- k_fm_forward + k_fm_update at B = 1 and k_fm_online, the update with and
// without the in-launch hand-off, ... -- give the same bits whatever the surrounding code looks like.  The kernels are
// bound by memory round trips, not by VALU issue; fused multiply-adds are written explicitly where they are wanted.
#pragma clang fp contract(off)

// ------------------------------------------------------------------------------------------------------------
// host-side error plumbing
// ------------------------------------------------------------------------------------------------------------
namespace fmxd {  // defined in fmx_kernels.hip
extern thread_local char g_err[512];
int fail(int code, const char *fmt, ...);
int check_launch(const char *what);

// ---- launch geometry knobs (waves per workgroup), overridable from the environment for experiments ----
struct Tune {
  int wpb_fwd = 2, wpb_upd = 2;  // waves per workgroup (FMX_WPB_FWD / FMX_WPB_UPD: 1, 2 or 4; a 3 x 3 sweep is flat within 1 %)
  int sort_e = 0;     // FMX_SORT_E: elements per thread of the bitonic sort (0 = default)
  int inline_fixup = 1;  // FMX_INLINE_FIXUP=0 / fmx_set_option("inline_fixup", 0): partial records are combined by a second
                         // launch (k_fm_fixup) instead of the in-launch hand-off; both give identical bits
  int online_persistent = 1;  // FMX_ONLINE_PERSISTENT=0 / fmx_set_option("online_persistent", 0): fmx_online_run_mlp as per-sample launches
  int sort_ahead = 16;  // FMX_SORT_AHEAD: batches per side-stream sort launch in fmx_fm_stream (1..16)
  int mlp_chain = 1;          // FMX_MLP_CHAIN=0 / fmx_set_option("mlp_chain", 0): fmx_mlp_section as separate GEMM launches
                              // (forward x L, loss, dgrad x L) instead of k_mlp_chain; same results up to summation order
  int sort_chunked = 1;  // FMX_SORT_CHUNKED / fmx_set_option("sort_chunked", v): 0: one workgroup per field (k_sort_occ) at
                         // every width; 1: k_sort_chunk + k_sort_merge from 8,192 composites per field on; 2: from 2,048 on.
                         // Identical lists either way
};
Tune &tune();

// ---- the library's RCCL communicator for the field-owner step (fmx_comm.hip) ----
constexpr int FMX_COMM_SLOTS = 4;  // batches whose indices may be gathered + sorted ahead of their step
struct Comm {
  int rank = 0, world = 1, n_blocks = 1;
  bool force = false;            // issue the collectives even with one rank (tests: the RCCL calls of a step on a one-GPU box)
  int block_count[FMX_COMM_MAX_WORLD] = {0}, block_first[FMX_COMM_MAX_WORLD] = {0};  // tree blocks per rank (fmx/plan.py)
  void *main = nullptr, *pf = nullptr;  // ncclComm_t of the step's stream / of the prefetch stream; null: one rank, nothing to exchange
  hipStream_t pf_stream = nullptr;      // the prefetch stream (library-owned)
  hipEvent_t fork = nullptr, ready[FMX_COMM_SLOTS] = {nullptr}, free_[FMX_COMM_SLOTS] = {nullptr};
  bool used[FMX_COMM_SLOTS] = {false};  // free_[s] has been recorded at least once
};
int comm_all_gather(Comm *c, int which, const void *send, void *recv, size_t count, hipStream_t st);
int comm_exchange_blocks(Comm *c, const float *send, float *recv, size_t per, hipStream_t st);

// ---- the fixed-order reduction of the MLP's partial weight gradients (k_mlp_reduce, or carried by k_fm_update_rider) ----
constexpr int MLP_BIG_MAX_L = 8;
struct MlpReduceArgs {
  const float *parts[MLP_BIG_MAX_L];  // [n_split, out_l, ldp_l]
  int out_dim[MLP_BIG_MAX_L], in_dim[MLP_BIG_MAX_L], ldp[MLP_BIG_MAX_L];
  long long grad_off[MLP_BIG_MAX_L];  // offset of W_l in the flat buffer; b_l follows W_l
  float *grads;
  float *params;  // with lr != 0: params -= lr * grad in the same pass (single-rank SGD)
  float lr;
  int n_split[MLP_BIG_MAX_L], n_layers;
  const float *loss_b;
  float *loss_out;
  int B;
  float inv_b;
};
// fmx_mlp_section without its last launch: `deferred` receives the arguments of the reduction instead, for a caller that carries
// its blocks in another launch of the same stream (fmx_deepfm_stream: inside the table update's) or launches it itself
// (mlp_launch_reduce).  Defined in fmx_mlp.hip.
int mlp_section_deferred_reduce(const fmx_mlp_t *mlp, int32_t loss_kind, const float *bi, int32_t ld_bi, const float *base, const float *y, int32_t B,
                                float inv_b, void *workspace, float *logit_out, float *dz_out, float *gbi_out, int32_t ld_gbi, float *grads,
                                float lr_apply, float *loss_out, hipStream_t st, MlpReduceArgs *deferred);
void mlp_launch_reduce(const MlpReduceArgs &a, hipStream_t st);
int mlp_reduce_blocks_per_layer(const MlpReduceArgs &a, int threads);
}  // namespace fmxd
using namespace fmxd;

namespace {

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

constexpr uint32_t SENT = 0xFFFFFFFFu;
constexpr int WAVE = 64;
constexpr int MAX_SORT_WIDTH = 32768;  // 128 KiB of the 160 KiB LDS

// ------------------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------------------

__device__ __forceinline__ float4 operator+(float4 a, float4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
__device__ __forceinline__ float4 operator-(float4 a, float4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
__device__ __forceinline__ float4 operator*(float4 a, float4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
__device__ __forceinline__ float4 operator*(float a, float4 b) { return {a * b.x, a * b.y, a * b.z, a * b.w}; }
__device__ __forceinline__ float4 splat(float a) { return {a, a, a, a}; }

__device__ __forceinline__ float4 shfl_xor4(float4 v, int m) {
  return {__shfl_xor(v.x, m), __shfl_xor(v.y, m), __shfl_xor(v.z, m), __shfl_xor(v.w, m)};
}
__device__ __forceinline__ float4 shfl_up4(float4 v, int d) {
  return {__shfl_up(v.x, d), __shfl_up(v.y, d), __shfl_up(v.z, d), __shfl_up(v.w, d)};
}
__device__ __forceinline__ float4 shfl4(float4 v, int src) {
  return {__shfl(v.x, src), __shfl(v.y, src), __shfl(v.z, src), __shfl(v.w, src)};
}

// v_rcp_f32 / v_sqrt_f32 are 1-ulp instructions; the IEEE-exact expansions hipcc emits for `/` and sqrtf cost 10-14
// VALU instructions each and made the FTRL kernels VALU-bound (profiles/r01_*).  1 ulp is ~1e-7 relative, two orders
// below the 1e-5 parity tolerance.
__device__ __forceinline__ float rcp_(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }

// FTRL-proximal weight from (z, n)  (McMahan et al. 2013, Algorithm 1); h.alpha holds 1/alpha on the device
__device__ __forceinline__ float ftrl_w(float z, float n, const fmx_hyper_t &h) {
  const float denom = fmaf(h.beta + sqrt_(n), h.alpha, h.l2);
  const float w = -(z - copysignf(h.l1, z)) * rcp_(denom);
  return fabsf(z) <= h.l1 ? 0.f : w;
}
__device__ __forceinline__ float4 ftrl_w4(float4 z, float4 n, const fmx_hyper_t &h) {
  return {ftrl_w(z.x, n.x, h), ftrl_w(z.y, n.y, h), ftrl_w(z.z, n.z, h), ftrl_w(z.w, n.w, h)};
}
// one FTRL-proximal update of (z, n) by gradient g; w is the weight derived from the OLD (z, n)
__device__ __forceinline__ void ftrl_upd(float &z, float &n, float w, float g, const fmx_hyper_t &h) {
  const float n2 = fmaf(g, g, n);
  const float sigma = (sqrt_(n2) - sqrt_(n)) * h.alpha;
  z = fmaf(-sigma, w, z + g);
  n = n2;
}

template <int RULE>
__device__ __forceinline__ float apply_rule(float p, float g, const fmx_hyper_t &h) {
  if (RULE == FMX_RULE_SIGNADAM) return p - h.lr * g * rcp_(fabsf(g) + h.eps);
  return p - h.lr * g;  // FMX_RULE_SGD
}
template <int RULE>
__device__ __forceinline__ float4 apply_rule4(float4 p, float4 g, const fmx_hyper_t &h) {
  return {apply_rule<RULE>(p.x, g.x, h), apply_rule<RULE>(p.y, g.y, h), apply_rule<RULE>(p.z, g.z, h),
          apply_rule<RULE>(p.w, g.w, h)};
}

__device__ __forceinline__ float sigmoidf_(float z) { return 1.f / (1.f + expf(-z)); }
// F.binary_cross_entropy_with_logits per element
__device__ __forceinline__ float bcewl(float z, float y) {
  return (1.f - y) * z + log1pf(expf(-fabsf(z))) + fmaxf(-z, 0.f);
}

// lane ^ M exchanges without the LDS crossbar (ds_bpermute made the sort LDS-pipe bound): DPP for M = 1, 2, 4, 8,
// v_permlane16/32_swap for M = 16, 32.
template <int M>
__device__ __forceinline__ uint32_t xor_lane(uint32_t v, int lane) {
  if constexpr (M == 1) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
  } else if constexpr (M == 2) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
  } else if constexpr (M == 4) {
    const int t = __builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true);     // row_half_mirror: i -> 7 - i
    return (uint32_t)__builtin_amdgcn_mov_dpp(t, 0x1B, 0xF, 0xF, true);         // quad_perm [3,2,1,0]: together i ^ 4
  } else if constexpr (M == 8) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xF, 0xF, true);  // row_ror:8
  } else if constexpr (M == 16) {
    const auto sw = __builtin_amdgcn_permlane16_swap(v, v, false, false);  // {[r0,r0,r2,r2], [r1,r1,r3,r3]}
    return (lane & 16) ? sw[0] : sw[1];
  } else {
    const auto sw = __builtin_amdgcn_permlane32_swap(v, v, false, false);  // {[lo,lo], [hi,hi]}
    return (lane & 32) ? sw[0] : sw[1];
  }
}

template <int M>
__device__ __forceinline__ float xor_lane_f(float v, int lane) {
  return __uint_as_float(xor_lane<M>(__float_as_uint(v), lane));
}
template <int M>
__device__ __forceinline__ float4 xor_lane_f4(float4 v, int lane) {
  return {xor_lane_f<M>(v.x, lane), xor_lane_f<M>(v.y, lane), xor_lane_f<M>(v.z, lane), xor_lane_f<M>(v.w, lane)};
}

// deterministic block reduction of src[0..n): every thread sums a strided set of elements (16-byte groups when dense),
// 16 independent loads in flight per round -- at B = 16,384 with 128 threads a 4-deep unroll left 32 dependent rounds of
// HBM latency per sum and the one workgroup that owns the bias became the longest path of the launch -- then an LDS tree.
// The order of the additions depends only on (n, ld == 1, min(blockDim, 128)): at most 128 threads take part, so that
// workgroups of any width give identical bits.
__device__ float block_sum(const float *src, int n, int ld, float *sm) {
  constexpr int U = 16;
  const int tid = threadIdx.x, nt = blockDim.x < 128 ? blockDim.x : 128;
  float acc = 0.f;
  if (tid >= nt) {  // bystanders of a wider workgroup: only the barriers
    __syncthreads();
    for (int w = nt >> 1; w > 0; w >>= 1) __syncthreads();
    const float r = sm[0];
    __syncthreads();
    return r;
  }
  if (ld == 1) {
    const int n4 = n >> 2;
    const float4 *src4 = reinterpret_cast<const float4 *>(src);
    for (int i0 = tid; i0 < n4; i0 += U * nt) {
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {  // branch-free: a group beyond the end reads group 0 and counts as zeros (a branch per
        const int i = i0 + u * nt;   // load made the compiler wait for every load before it issued the next)
        v[u] = src4[i < n4 ? i : 0];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool in = i0 + u * nt < n4;
        acc += in ? (v[u].x + v[u].y) + (v[u].z + v[u].w) : 0.f;
      }
    }
    for (int i = (n4 << 2) + tid; i < n; i += nt) acc += src[i];
  } else {  // strided sample records: the SAME order of additions as the dense form (groups of four elements, then the
            // tail), so a step over gathered records and a step over dense arrays give identical bits
    constexpr int V = 4;
    const int n4 = n >> 2;
    for (int i0 = tid; i0 < n4; i0 += V * nt) {
      float4 v[V];
#pragma unroll
      for (int u = 0; u < V; ++u) {
        const int i = i0 + u * nt;
        const float *p = src + (size_t)(4 * (i < n4 ? i : 0)) * ld;
        v[u] = float4{p[0], p[ld], p[2 * (size_t)ld], p[3 * (size_t)ld]};
      }
#pragma unroll
      for (int u = 0; u < V; ++u) {
        const bool in = i0 + u * nt < n4;
        acc += in ? (v[u].x + v[u].y) + (v[u].z + v[u].w) : 0.f;
      }
    }
    for (int i = (n4 << 2) + tid; i < n; i += nt) acc += src[(size_t)i * ld];
  }
  sm[tid] = acc;
  __syncthreads();
  for (int w = nt >> 1; w > 0; w >>= 1) {
    if (tid < w) sm[tid] += sm[tid + w];
    __syncthreads();
  }
  const float r = sm[0];
  __syncthreads();
  return r;
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Block `block` of `n_blocks` of layer l: one thread per 4 consecutive columns of one row of the layer's partial [out, ldp] (ldp a
// multiple of 4: the weight columns, the bias column `in`, padding): 16-byte loads of up to 16 splits in flight, summed in the
// order z = 0, 1, ...; block 0 of layer 0 also reduces the loss.  Any workgroup width (the table update's is 64 or 128 or 256).
__device__ __forceinline__ void mlp_reduce_block(const MlpReduceArgs &a, int l, int block, int n_blocks) {
  const int out = a.out_dim[l], in = a.in_dim[l], ldp = a.ldp[l], ns = a.n_split[l];
  const int groups = ldp >> 2;
  const long long n = (long long)out * groups;
  const float *p = a.parts[l];
  const size_t zs = (size_t)out * ldp;
  for (long long i = (long long)block * blockDim.x + threadIdx.x; i < n; i += (long long)n_blocks * blockDim.x) {
    const int m = (int)(i / groups), c = (int)(i - (long long)m * groups) * 4;
    const float *q = p + (size_t)m * ldp + c;
    // the parameters this thread updates, requested with the partials instead of behind their sum (a dependent round trip less);
    // a group of four weight columns is 16 contiguous, 16-byte aligned bytes of W_l when in % 4 == 0 (else the scalar path below)
    const bool vec4 = a.lr != 0.f && c + 3 < in && (in & 3) == 0 && (a.grad_off[l] & 3) == 0;
    float4 pv = {0.f, 0.f, 0.f, 0.f};
    if (vec4) pv = *reinterpret_cast<const float4 *>(a.params + a.grad_off[l] + (long long)m * in + c);
    float4 s = {0.f, 0.f, 0.f, 0.f};
    for (int z0 = 0; z0 < ns; z0 += 16) {
      float4 r[16];
#pragma unroll
      for (int z = 0; z < 16; ++z)
        if (z0 + z < ns) r[z] = *reinterpret_cast<const float4 *>(q + (size_t)(z0 + z) * zs);
#pragma unroll
      for (int z = 0; z < 16; ++z)
        if (z0 + z < ns) {
          s.x += r[z].x;
          s.y += r[z].y;
          s.z += r[z].z;
          s.w += r[z].w;
        }
    }
    if (vec4) {
      const long long o = a.grad_off[l] + (long long)m * in + c;
      *reinterpret_cast<float4 *>(a.grads + o) = s;
      *reinterpret_cast<float4 *>(a.params + o) = float4{pv.x - a.lr * s.x, pv.y - a.lr * s.y, pv.z - a.lr * s.z, pv.w - a.lr * s.w};
      continue;
    }
    const float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (c + j > in) break;  // padding
      const long long o = c + j < in ? a.grad_off[l] + (long long)m * in + c + j : a.grad_off[l] + (long long)out * in + m;
      a.grads[o] = v[j];
      if (a.lr != 0.f) a.params[o] -= a.lr * v[j];
    }
  }
  if (block == 0 && l == 0 && a.loss_out) {
    __shared__ float sm[256];
    const float ls = block_sum(a.loss_b, a.B, 1, sm);
    if (threadIdx.x == 0) a.loss_out[0] = ls * a.inv_b;
  }
}

}  // namespace
