// fmx_kernels.hip -- gfx950 (MI355X, CDNA4) kernels for the FM / DeepFM / NFM online hot path and their C ABI.
//
// Four kernels make one mini-batch step (DESIGN.md has the data layout and the byte accounting):
//
//   k_sort_occ    one workgroup per field: the batch's (local index, sample) pairs are packed into 32-bit
//                 composites and bitonic-sorted (registers + wave shuffles, LDS only for the cross-wave stages).
//                 Equal rows become adjacent runs ordered by sample, so the duplicate-row reduction the reference
//                 gets from embedding_dense_backward (reference fm_adam.py:67) is deterministic.  Independent of
//                 the weights.
//   k_fm_forward  one 64-lane wavefront per sample, LPR lanes per gathered row (16-byte loads, one request per
//                 64-byte row), every index / row load of the sample issued before the first use, butterfly
//                 shuffles for the field sums (reference fm_adam.py:35-53), fused loss / dlogit epilogue
//                 (fm_adam.py:61,66 / :76,80).
//   k_fm_update   one wavefront per tile of 64 sorted occurrences: a segmented scan over the runs in the tile; a run
//                 that lies inside the tile gets ONE fused read-modify-write of its row under the chosen rule
//                 (reference loss.backward() + optimizer.step(), fm_adam.py:67-68 / :81-82); a run that crosses a tile
//                 boundary leaves a partial sum.  The last workgroup reduces the bias gradient and the loss.
//   k_fm_fixup    one wavefront per run that crosses tile boundaries (rows hit > 64 times, i.e. the small-vocabulary
//                 fields): adds the partial sums in tile order and applies the row update.
//
// Everything is HBM / cache-line bound integer+fp32 work; there is no GEMM here and no MFMA.

#include "fmx_common.h"

// ------------------------------------------------------------------------------------------------------------
// host-side error plumbing (shared by the translation units: fmx_common.h)
// ------------------------------------------------------------------------------------------------------------
namespace fmxd {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(FMX_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return FMX_OK;
}

}  // namespace fmxd

namespace {

#include "fmx_sort.inc"

// ------------------------------------------------------------------------------------------------------------
// k_fm_forward
// ------------------------------------------------------------------------------------------------------------
// Agent-scope relaxed atomic accesses compile to `global_store/load ... sc1` (write-through / L1-bypassing): the form
// the in-launch hand-off of partial records uses on BOTH sides (MI355X_MICROARCH.md, "Valid forms": every store and every
// load of the handed-off bytes sc1, the storing wave's s_waitcnt vmcnt(0) before its flag store).
__device__ __forceinline__ void st_sc1(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_sc1(const float *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4 ld_sc1_4(const float *p) { return {ld_sc1(p), ld_sc1(p + 1), ld_sc1(p + 2), ld_sc1(p + 3)}; }
// 16-byte store: one instruction per lane (`global_store_dwordx4 ... sc1`; scalar sc1 stores are one fabric write each).
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_sc1_4(float *p, float4 v) {
  const v4f x = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ void store_part_sc1(float *rec, int q, int kp, float4 cV, float4 cA, float cw) {
  st_sc1_4(rec + 4 * q, cV);
  st_sc1_4(rec + kp + 4 * q, cA);
  if (q == 0) st_sc1(rec + 2 * kp, cw);
}

struct FwdArgs {
  const float *rows;
  const int64_t *foff;
  const float *bias;
  const int32_t *idx;
  const float *xv;
  const float *y;
  fmx_fwd_out_t out;
  fmx_hyper_t h;
  int32_t B, F, kp, stride, zoff, loss_kind;
  int32_t ldS, ld1;  // floats between consecutive samples in out.S and in out.dz / out.loss (kp and 1 when dense)
  float inv_b;
  // fields as pieces of index columns (fmx_table_t.field_cols / field_base; both null on ordinary tables): Fc = columns of idx
  const int32_t *fcols, *fbase;
  int32_t Fc;
};

// The part of the forward pass behind the row gather: field sums (butterfly over the lane groups), bi-interaction, logit,
// loss and dlogit, stores.  s / ss / fo: this lane's partial sums of e, e*e and of the first-order terms over ITS fields.
template <int LPR, int LAYOUT>
__device__ __forceinline__ void forward_finish(const FwdArgs &a, const int b, const int lane, float4 s, float4 ss, float fo, bool bad,
                                               const float y_early, const float bias0_early, const float bias1_early) {
  const int q = lane % LPR;
  const int kp = LPR * 4;
  if (bad && a.out.error) *a.out.error = 1;

  // field sums: butterfly over the slots (lanes with equal q), DPP / permlane exchanges (no LDS crossbar)
#define FMX_BFLY(M)                               \
  if (LPR <= M) {                                 \
    s = s + xor_lane_f4<M>(s, lane);              \
    ss = ss + xor_lane_f4<M>(ss, lane);           \
    fo += xor_lane_f<M>(fo, lane);                \
  }
  FMX_BFLY(1) FMX_BFLY(2) FMX_BFLY(4) FMX_BFLY(8) FMX_BFLY(16) FMX_BFLY(32)
#undef FMX_BFLY
  const float4 bi = 0.5f * (s * s - ss);
  float sbi = (bi.x + bi.y) + (bi.z + bi.w);
#pragma unroll
  for (int m = 1; m < LPR; m <<= 1) sbi += __shfl_xor(sbi, m);
  // fo: lanes with q != 0 hold the sum of zeros; take the q == 0 value
  fo = __shfl(fo, 0);

  if (lane < LPR) {
    if (a.out.S) *reinterpret_cast<float4 *>(a.out.S + (size_t)b * a.ldS + 4 * q) = s;
    if (a.out.bi) *reinterpret_cast<float4 *>(a.out.bi + (size_t)b * kp + 4 * q) = bi;
  }
  if (lane == 0) {
    float bias;
    if (LAYOUT == FMX_LAYOUT_WEIGHTS) bias = bias0_early;
    else bias = ftrl_w(bias0_early, bias1_early, a.h);
    const float z = fo + sbi + bias;
    if (a.out.sfirst) a.out.sfirst[b] = fo;
    if (a.out.sbi) a.out.sbi[b] = sbi;
    if (a.out.logit) a.out.logit[b] = z;
    if (a.loss_kind != FMX_LOSS_NONE) {
      const float y = y_early;
      float loss, dz;
      if (a.loss_kind == FMX_LOSS_BCE_LOGITS) {
        loss = bcewl(z, y);
        dz = (sigmoidf_(z) - y) * a.inv_b;
      } else {
        const float p = sigmoidf_(z);
        loss = bcewl(p, y);
        dz = (sigmoidf_(p) - y) * p * (1.f - p) * a.inv_b;
      }
      if (a.out.loss) a.out.loss[(size_t)b * a.ld1] = loss;
      if (a.out.dz) a.out.dz[(size_t)b * a.ld1] = dz;
    }
  }
}

// NPASS > 0: the field loop is fully unrolled (F <= NPASS * SLOTS) and every index, value, offset and row load of the
// sample is issued before the first use, so one wave keeps up to 3 * NPASS row requests in flight.  NPASS == 0: generic.
// MAPPED (tables whose fields are pieces of index columns; the generic field loop only): field f reads column fcols[f], holds
// the indices [fbase[f], fbase[f] + rows) of it, and an index outside belongs to another piece: no contribution, no error.
template <int LPR, int LAYOUT, int NPASS, bool MAPPED = false>
__device__ __forceinline__ void forward_sample(const FwdArgs &a, const int b, const int lane) {
  constexpr int SLOTS = WAVE / LPR;
  constexpr int NP = NPASS > 0 ? NPASS : 1;
  const int slot = lane / LPR, q = lane % LPR;
  const int kp = LPR * 4;

  float4 s = splat(0.f), ss = splat(0.f);
  float fo = 0.f;
  bool bad = false;
  // the label and the bias are only needed by lane 0's epilogue, but a load issued there is one more dependent round trip
  // at the end of every wave: request them now, with the indices (every lane the same address: one request each)
  const float y_early = a.loss_kind != FMX_LOSS_NONE ? a.y[b] : 0.f;
  const float bias0_early = a.bias[0];
  const float bias1_early = LAYOUT == FMX_LAYOUT_WEIGHTS ? 0.f : a.bias[1];
  __builtin_amdgcn_sched_barrier(0);  // (left to itself the scheduler sinks the bias load behind the first waits of the gather:
                                      //  in-order vmcnt then makes the row requests wait for it -- one more round trip)
  const int n_outer = NPASS > 0 ? 1 : (a.F + SLOTS - 1) / SLOTS;
  for (int it = 0; it < n_outer; ++it) {
    uint32_t li[NP];
    float x[NP];
    int64_t lo[NP];
    uint32_t vocab[NP];
    bool live[NP];
    // Branch-free: a lane group beyond the last field reads field F - 1's index and offsets, an index beyond the field's
    // vocabulary reads the field's first row, and the results are dropped by selects.  With `if (live) { loads; arithmetic on
    // them }` per pass the compiler put an s_waitcnt vmcnt(0) at the end of every pass's region: three dependent round trips
    // for the indices of a 39-field sample instead of one.
    const float *xsrc = a.xv ? a.xv : reinterpret_cast<const float *>(a.idx);  // something loadable when there are no values
    const bool has_x = a.xv != nullptr;
    float xl[NP];
    int64_t hi[NP];
    int fc[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int f = (it * NP + p) * SLOTS + slot;
      live[p] = f < a.F;
      fc[p] = live[p] ? f : a.F - 1;
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {  // the offsets do not depend on the sample: first
      lo[p] = a.foff[fc[p]];
      hi[p] = a.foff[fc[p] + 1];
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      size_t o = (size_t)b * a.F + fc[p];
      uint32_t fb = 0u;
      if (MAPPED) {
        o = (size_t)b * a.Fc + (a.fcols ? a.fcols[fc[p]] : fc[p]);
        fb = a.fbase ? (uint32_t)a.fbase[fc[p]] : 0u;
      }
      li[p] = (uint32_t)a.idx[o] - fb;  // (wraps to a huge value below the piece)
      xl[p] = xsrc[o];
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      x[p] = has_x ? xl[p] : 1.f;
      vocab[p] = (uint32_t)(hi[p] - lo[p]);
    }
    // both layouts keep [ V | w ] at the head of the row: the forward never touches the FTRL (z, n) half
    float4 r0[NP];
    float rw[NP];
    bool ok[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      ok[p] = live[p] && li[p] < vocab[p];
      const float *rp = a.rows + (size_t)(lo[p] + (ok[p] ? li[p] : 0u)) * a.stride;
      r0[p] = *reinterpret_cast<const float4 *>(rp + 4 * q);
      rw[p] = rp[kp];  // (every lane of the group: the same address, one request)
    }
    __builtin_amdgcn_sched_barrier(0);  // ... and every row request before the first sum
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const float4 e = x[p] * r0[p];
      const float f1 = ok[p] ? rw[p] * x[p] : 0.f;
      if (ok[p]) {  // selects
        s = s + e;
        ss = ss + e * e;
        fo += f1;
      }
      bad = bad || (!MAPPED && live[p] && !ok[p]);
      if (live[p] && a.out.first && q == 0) a.out.first[(size_t)b * a.F + (it * NP + p) * SLOTS + slot] = f1;
    }
  }
  forward_finish<LPR, LAYOUT>(a, b, lane, s, ss, fo, bad, y_early, bias0_early, bias1_early);
}

template <int LPR, int LAYOUT, int NPASS, bool MAPPED = false>
__global__ __launch_bounds__(256) void k_fm_forward(FwdArgs a) {
  __builtin_amdgcn_s_setprio(3);  // ahead of the side-stream sort's waves at the CU's instruction arbiter
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= a.B) return;  // wave-uniform
  forward_sample<LPR, LAYOUT, NPASS, MAPPED>(a, b, threadIdx.x & 63);
}

// ------------------------------------------------------------------------------------------------------------
// k_fm_forward_part / k_fm_forward_finish: the forward pass split over FIELD OWNERS (model-parallel multi-GPU mode)
// ------------------------------------------------------------------------------------------------------------
// k_fm_forward sums a sample's rows in a fixed tree: lane group `slot` (of SLOTS = 64 / LPR) adds the fields
// slot, SLOTS + slot, 2 SLOTS + slot, ... in order, then a butterfly over the lane groups (slot ^ 1, ^ 2, ^ 4, ...).  The tree is
// cut into NB BLOCKS of SL = SLOTS / NB consecutive lane groups (NB a power of two); an owner holds one or more blocks -- the
// fields at their positions, as a table of its own: local field (lb NP + p) SL + s is position p SLOTS + (first block + lb) SL + s
// of the whole tree -- and k_fm_forward_part evaluates exactly those sub-trees for EVERY sample of the global batch: NB samples
// per wave, SL lane groups each, the butterfly levels below SL -- one record (S_part[kp], ss_part[kp], fo_part) per sample
// and block.  The records of a sample meet on the rank that holds its label (an all-to-all), where k_fm_forward_finish adds
// them in the order of the remaining butterfly levels, ((b0 + b1) + (b2 + b3)) + ..., and runs k_fm_forward's epilogue.
// The result is bit-identical to k_fm_forward on one GPU whose table has the same fields at the same positions.
struct PartArgs {
  const float *rows;
  const int64_t *foff;   // the owner's LOCAL table (blockIdx.y = local block lb): fields [lb NP SL, (lb + 1) NP SL) are the block's
  const int32_t *fcols, *fbase;  // fmx_table_t.field_cols / field_base, or null
  const int32_t *idx;    // [B, Fc], B = global batch
  const float *xv;       // [B, Fc] or null
  float *rec;            // [B / group][n_blocks][group, 2 kp + 4]: S | ss | fo, 0, 0, 0 -- the records of the `group` samples one rank
                         // holds the labels of lie together, block after block: one contiguous message per destination
  int32_t *error;
  int32_t B, F, Fc, stride, sl_log2, group;  // F: fields of the table (all blocks)
};

template <int LPR, int NPASS>
__global__ __launch_bounds__(256) void k_fm_forward_part(PartArgs a) {
  constexpr int SLOTS = WAVE / LPR;
  constexpr int kp = LPR * 4, REC = 2 * kp + 4;
  const int lane = threadIdx.x & 63;
  const int slot = lane / LPR, q = lane % LPR;
  const int SL = 1 << a.sl_log2;
  const int slot_l = slot & (SL - 1);
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int b = wave * (SLOTS >> a.sl_log2) + (slot >> a.sl_log2);
  const bool valid = b < a.B;
  const int f0 = (int)blockIdx.y * NPASS * SL;  // first field of this block
  const bool pieces = a.fcols || a.fbase;
  // branch-free gather, as in forward_sample: every offset and index load, then every row load, then the sums by selects
  uint32_t li[NPASS], vocab[NPASS];
  float x[NPASS], xl[NPASS];
  int64_t lo[NPASS], hi[NPASS];
  bool live[NPASS], ok[NPASS];
  int fc[NPASS];
  const float *xsrc = a.xv ? a.xv : reinterpret_cast<const float *>(a.idx);
  const bool has_x = a.xv != nullptr;
  const int bc = valid ? b : 0;
#pragma unroll
  for (int p = 0; p < NPASS; ++p) {
    const int f = f0 + p * SL + slot_l;
    live[p] = valid && f < a.F;
    fc[p] = f < a.F ? f : a.F - 1;
  }
  int colp[NPASS];
  uint32_t fb[NPASS];
#pragma unroll
  for (int p = 0; p < NPASS; ++p) {
    lo[p] = a.foff[fc[p]];
    hi[p] = a.foff[fc[p] + 1];
    colp[p] = a.fcols ? a.fcols[fc[p]] : fc[p];
    fb[p] = a.fbase ? (uint32_t)a.fbase[fc[p]] : 0u;
  }
#pragma unroll
  for (int p = 0; p < NPASS; ++p) {
    const size_t o = (size_t)bc * a.Fc + colp[p];
    li[p] = (uint32_t)a.idx[o] - fb[p];  // (wraps to a huge value below the piece)
    xl[p] = xsrc[o];
  }
#pragma unroll
  for (int p = 0; p < NPASS; ++p) {
    x[p] = has_x ? xl[p] : 1.f;
    vocab[p] = (uint32_t)(hi[p] - lo[p]);
  }
  float4 r0[NPASS];
  float rw[NPASS];
#pragma unroll
  for (int p = 0; p < NPASS; ++p) {
    ok[p] = live[p] && li[p] < vocab[p];
    const float *rp = a.rows + (size_t)(lo[p] + (ok[p] ? li[p] : 0u)) * a.stride;
    r0[p] = *reinterpret_cast<const float4 *>(rp + 4 * q);
    rw[p] = rp[kp];
  }
  __builtin_amdgcn_sched_barrier(0);
  float4 s = splat(0.f), ss = splat(0.f);
  float fo = 0.f;
  bool bad = false;
#pragma unroll
  for (int p = 0; p < NPASS; ++p) {
    const float4 e = x[p] * r0[p];
    if (ok[p]) {  // selects
      s = s + e;
      ss = ss + e * e;
      fo += rw[p] * x[p];
    }
    bad = bad || (!pieces && live[p] && !ok[p]);
  }
  if (bad && a.error) *a.error = 1;
  // the butterfly levels inside the block's lane groups (wave-uniform conditions)
#define FMX_BFLY_L(M)                              \
  if (LPR <= M && (M / LPR) < SL) {                \
    s = s + xor_lane_f4<M>(s, lane);               \
    ss = ss + xor_lane_f4<M>(ss, lane);            \
    fo += xor_lane_f<M>(fo, lane);                 \
  }
  FMX_BFLY_L(1) FMX_BFLY_L(2) FMX_BFLY_L(4) FMX_BFLY_L(8) FMX_BFLY_L(16) FMX_BFLY_L(32)
#undef FMX_BFLY_L
  if (valid && slot_l == 0) {
    const int dst = b / a.group;
    float *r = a.rec + (((size_t)dst * gridDim.y + blockIdx.y) * a.group + (b - dst * a.group)) * REC;
    *reinterpret_cast<float4 *>(r + 4 * q) = s;
    *reinterpret_cast<float4 *>(r + kp + 4 * q) = ss;
    if (q == 0) *reinterpret_cast<float4 *>(r + 2 * kp) = float4{fo, 0.f, 0.f, 0.f};
  }
}

struct FinishArgs {
  const float *parts;    // [G][B, 2 kp + 4]: block r holds rank r's records of THIS rank's B samples
  int64_t rank_stride;   // floats between the blocks
  const float *bias;
  const float *y;
  fmx_fwd_out_t out;
  fmx_hyper_t h;
  int32_t B, loss_kind, ldS, ld1;
  float inv_b;
};

template <int LPR, int LAYOUT, int G>
__global__ __launch_bounds__(256) void k_fm_forward_finish(FinishArgs a) {
  constexpr int SLOTS = WAVE / LPR;
  constexpr int kp = LPR * 4, REC = 2 * kp + 4;
  const int lane = threadIdx.x & 63;
  const int slot = lane / LPR, q = lane % LPR;
  const int b = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * SLOTS + slot;
  const bool valid = b < a.B;
  const float y_early = (valid && a.loss_kind != FMX_LOSS_NONE) ? a.y[b] : 0.f;
  const float bias0 = a.bias[0];
  const float bias1 = LAYOUT == FMX_LAYOUT_WEIGHTS ? 0.f : a.bias[1];
  float4 s[G], ss[G];
  float fo[G];
#pragma unroll
  for (int r = 0; r < G; ++r) {
    s[r] = ss[r] = splat(0.f);
    fo[r] = 0.f;
    if (valid) {
      const float *p = a.parts + (size_t)r * a.rank_stride + (size_t)b * REC;
      s[r] = *reinterpret_cast<const float4 *>(p + 4 * q);
      ss[r] = *reinterpret_cast<const float4 *>(p + kp + 4 * q);
      if (q == 0) fo[r] = p[2 * kp];
    }
  }
  // the butterfly levels ABOVE the owners' lane groups: rank pairs, then pairs of pairs, ...
#pragma unroll
  for (int st = 1; st < G; st <<= 1) {
#pragma unroll
    for (int r = 0; r < G; r += 2 * st) {
      s[r] = s[r] + s[r + st];
      ss[r] = ss[r] + ss[r + st];
      fo[r] += fo[r + st];
    }
  }
  const float4 bi = 0.5f * (s[0] * s[0] - ss[0]);
  float sbi = (bi.x + bi.y) + (bi.z + bi.w);
#pragma unroll
  for (int m = 1; m < LPR; m <<= 1) sbi += __shfl_xor(sbi, m);
  if (!valid) return;
  if (a.out.S) *reinterpret_cast<float4 *>(a.out.S + (size_t)b * a.ldS + 4 * q) = s[0];
  if (a.out.bi) *reinterpret_cast<float4 *>(a.out.bi + (size_t)b * kp + 4 * q) = bi;
  if (q == 0) {
    float bias;
    if (LAYOUT == FMX_LAYOUT_WEIGHTS) bias = bias0;
    else bias = ftrl_w(bias0, bias1, a.h);
    const float z = fo[0] + sbi + bias;
    if (a.out.sfirst) a.out.sfirst[b] = fo[0];
    if (a.out.sbi) a.out.sbi[b] = sbi;
    if (a.out.logit) a.out.logit[b] = z;
    if (a.loss_kind != FMX_LOSS_NONE) {
      const float y = y_early;
      float loss, dz;
      if (a.loss_kind == FMX_LOSS_BCE_LOGITS) {
        loss = bcewl(z, y);
        dz = (sigmoidf_(z) - y) * a.inv_b;
      } else {
        const float pp = sigmoidf_(z);
        loss = bcewl(pp, y);
        dz = (sigmoidf_(pp) - y) * pp * (1.f - pp) * a.inv_b;
      }
      if (a.out.loss) a.out.loss[(size_t)b * a.ld1] = loss;
      if (a.out.dz) a.out.dz[(size_t)b * a.ld1] = dz;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// k_fm_update
// ------------------------------------------------------------------------------------------------------------
struct UpdArgs {
  float *rows;
  const int64_t *foff;
  float *bias;
  const uint32_t *sorted;
  float *parts;   // [F * tiles, 2 (lead, trail), REC] partial sums of runs that cross a tile boundary
  int32_t *meta;  // [F * tiles, 2] (lead_state, trail_state)
  const float *xv;
  const float *S;
  const float *dz_first;
  const float *dz_bi;
  const float *gbi;
  const float *loss_b;
  float *loss_out;
  int32_t *step_counter;  // null: loss_out[0]; else loss_out[*step_counter], then *step_counter += 1
  fmx_hyper_t h;
  int32_t B, F, Bp, bbits, kp, stride, zoff;
  int32_t ldS, ld1;  // floats between consecutive samples in S and in dz_first / dz_bi / loss_b (kp and 1 when dense)
  int32_t ldG;       // ... and in gbi (kp when dense)
  const int32_t *cols;  // sort field -> field, or null; fcols: field -> column of xv, or null; Fx: columns of xv
  const int32_t *fcols;
  int32_t Fx;
  float *red;        // [16][4] partial (sum dlogit, sum loss, launch sequence, -) of the batch slices (B > RED_SLICE)
  uint32_t seq;      // launch sequence number tagging the tile meta words (INL) and the slice partials of this launch
  int32_t *error;    // INL: set to 2 if a hand-off wait ran into its bound
  float inv_b;
};

// tile meta states
constexpr int LEAD_NONE = 0, LEAD_CLOSES = 1, LEAD_THROUGH = 2;

// Row stores are nontemporal: the lines go out while the launch runs instead of staying dirty in L2 for the write-back at its end
// (same-box A/B of the online loop, tools/ab_old_new.sh: 21.51 / 21.60 against 21.90 / 21.83 us per step).
__device__ __forceinline__ void st16(float *p, float4 v) {
  typedef float f4v __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(f4v{v.x, v.y, v.z, v.w}, reinterpret_cast<f4v *>(p));
}
__device__ __forceinline__ void st4(float *p, float v) { __builtin_nontemporal_store(v, p); }

// The bias gradient (sum of dlogit over the batch) and the mean loss.  The batch is cut into slices of RED_SLICE samples, one
// workgroup each (the first workgroups of the launch): with the samples' (S, dlogit, loss) records gathered from G ranks the
// values lie 80 bytes apart, one 64-byte request each, and ONE workgroup walking 2 x 32,768 of them was the longest path of
// the launch by far (47-53 us of the update at 8 x 4,096 samples against 13 at 4,096).  One slice (B <= RED_SLICE): the sum
// and the update in place, as before.  Several: every slice's workgroup leaves (sum dlogit, sum loss, launch sequence) as ONE
// 16-byte write-through granule; the first workgroup polls the others' granules until they carry this launch's sequence number
// (data and tag in one granule: no ordering needed; they belong to workgroups dispatched right behind it, which wait on
// nothing), adds the partials in slice order and applies the update -- or, without the in-launch hand-off, k_fm_fixup's
// last workgroup does that.  The order of the additions depends on the batch size alone: every mode gives the same bits.
constexpr int RED_SLICE = 4096;
__host__ __device__ inline int red_slices(int B) { return (B + RED_SLICE - 1) / RED_SLICE; }

template <int LAYOUT, int RULE>
__device__ __forceinline__ void apply_bias_and_loss(const UpdArgs &a, float db, float ls) {
  if (LAYOUT == FMX_LAYOUT_WEIGHTS) {
    st4(a.bias, apply_rule<RULE>(a.bias[0], db, a.h));
  } else {
    float z = a.bias[0], n = a.bias[1];
    const float w = ftrl_w(z, n, a.h);
    ftrl_upd(z, n, w, db, a.h);
    st4(a.bias, z);
    st4(a.bias + 1, n);
  }
  if (a.loss_b && a.loss_out) {
    int i = 0;
    if (a.step_counter) {
      i = *a.step_counter;
      *a.step_counter = i + 1;
    }
    a.loss_out[i] = ls * a.inv_b;
  }
}

// the partials of all R slices added in slice order (thread 0 of the calling workgroup applies them)
template <int LAYOUT, int RULE>
__device__ void finish_bias_and_loss(const UpdArgs &a, int R, bool poll) {
  __shared__ float part[2 * 16];
  const int t = threadIdx.x;
  bool failed = false;
  if (t < R) {
    const v4f *src = reinterpret_cast<const v4f *>(a.red) + t;
    v4f g = {0.f, 0.f, 0.f, 0.f};
    if (poll) {
      for (int spin = 0;; ++spin) {
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(g) : "v"(src) : "memory");
        if (__float_as_uint(g.z) == a.seq) break;
        if (spin >= (1 << 20)) {
          failed = true;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    } else {
      g = *src;
    }
    part[2 * t] = g.x;
    part[2 * t + 1] = g.y;
  }
  if (failed && a.error) *a.error = 2;
  __syncthreads();
  if (t == 0) {
    float db = 0.f, ls = 0.f;
    for (int r = 0; r < R; ++r) {
      db += part[2 * r];
      ls += part[2 * r + 1];
    }
    apply_bias_and_loss<LAYOUT, RULE>(a, db, ls);
  }
}

template <int LAYOUT, int RULE, bool INL>
__device__ void bias_and_loss(const UpdArgs &a, int r) {
  __shared__ float sm[256];
  const int R = red_slices(a.B);
  const int first = r * RED_SLICE, n = (a.B - first) < RED_SLICE ? (a.B - first) : RED_SLICE;
  const float db = block_sum(a.dz_first + (size_t)first * a.ld1, n, a.ld1, sm);
  float ls = 0.f;
  if (a.loss_b && a.loss_out) ls = block_sum(a.loss_b + (size_t)first * a.ld1, n, a.ld1, sm);
  if (R == 1) {
    if (threadIdx.x == 0) apply_bias_and_loss<LAYOUT, RULE>(a, db, ls);
    return;
  }
  if (threadIdx.x == 0) st_sc1_4(a.red + 4 * r, float4{db, ls, __uint_as_float(a.seq), 0.f});
  if (INL && r == 0) finish_bias_and_loss<LAYOUT, RULE>(a, R, true);
}

// The state of one row as LPR lanes hold it: lane q owns coordinates 4q..4q+3; lane 0 also the first-order part.
//   WEIGHTS row  [ V(kp) | w | pad ]
//   FTRL row     [ V(kp) | w, zw, nw, pad | ... | zV(kp) | nV(kp) ]   zV starts at float `zoff`;
//                V and w are the weights derived from (z, n), re-derived and stored by every update
struct RowRegs {
  float4 v;       // V
  float4 z, n;    // FTRL only
  float4 fo;      // lane 0: (w, zw, nw, -)
};

template <int LAYOUT>
__device__ __forceinline__ RowRegs load_row(const float *rp, int q, int kp, int zoff) {
  RowRegs r;
  r.v = *reinterpret_cast<const float4 *>(rp + 4 * q);
  r.z = splat(0.f);
  r.n = splat(0.f);
  r.fo = splat(0.f);
  // the first-order part is used by lane 0 of the group only, but every lane requests it (same address: one request): a
  // load under `if (q == 0)` is a branch, and the compiler's wait counts fall back to vmcnt(0) around it
  if (LAYOUT == FMX_LAYOUT_WEIGHTS) {
    r.fo.x = rp[kp];
  } else {
    r.z = *reinterpret_cast<const float4 *>(rp + zoff + 4 * q);
    r.n = *reinterpret_cast<const float4 *>(rp + zoff + kp + 4 * q);
    r.fo = *reinterpret_cast<const float4 *>(rp + kp);
  }
  return r;
}

// gradient of the row from the run sums (dV = cV - V * cA, dw = cw), one application of the rule, store
template <int LAYOUT, int RULE>
__device__ __forceinline__ void update_row(float *rp, int q, int kp, int zoff, RowRegs r, float4 cV, float4 cA, float cw,
                                           const fmx_hyper_t &h) {
  // two roundings on purpose (no fma): where a sample is the row's only contribution to S (x = 1), cV = dz * V and
  // V * cA = V * dz round alike and the gradient is exactly 0, as in the reference's dz * x * (S - e)
  const float4 gr = cV - r.v * cA;
  if (LAYOUT == FMX_LAYOUT_WEIGHTS) {
    st16(rp + 4 * q, apply_rule4<RULE>(r.v, gr, h));
    if (q == 0) st4(rp + kp, apply_rule<RULE>(r.fo.x, cw, h));
  } else {
    float4 z4 = r.z, n4 = r.n;
    ftrl_upd(z4.x, n4.x, r.v.x, gr.x, h);
    ftrl_upd(z4.y, n4.y, r.v.y, gr.y, h);
    ftrl_upd(z4.z, n4.z, r.v.z, gr.z, h);
    ftrl_upd(z4.w, n4.w, r.v.w, gr.w, h);
    // the (z, n) half is read by nobody but the next update of this row (a later launch): plain stores
    *reinterpret_cast<float4 *>(rp + zoff + 4 * q) = z4;
    *reinterpret_cast<float4 *>(rp + zoff + kp + 4 * q) = n4;
    st16(rp + 4 * q, ftrl_w4(z4, n4, h));
    if (q == 0) {
      float4 fo = r.fo;
      ftrl_upd(fo.y, fo.z, fo.x, cw, h);
      fo.x = ftrl_w(fo.y, fo.z, h);
      st16(rp + kp, fo);
    }
  }
}

// partial-sum record: [cV (kp) | cA (kp) | cw, pad3]
__device__ __forceinline__ void store_part(float *rec, int q, int kp, float4 cV, float4 cA, float cw) {
  *reinterpret_cast<float4 *>(rec + 4 * q) = cV;
  *reinterpret_cast<float4 *>(rec + kp + 4 * q) = cA;
  if (q == 0) rec[2 * kp] = cw;
}

// run sums carried per occurrence: cV = sum x G S (vector), cA = sum x^2 G (a scalar when G is one: pure FM), cw
template <bool VEC> struct CoefA;
template <> struct CoefA<true> {
  float4 v;
  __device__ __forceinline__ void zero() { v = splat(0.f); }
  __device__ __forceinline__ void add(const CoefA &o) { v = v + o.v; }
  __device__ __forceinline__ float4 vec() const { return v; }
  __device__ __forceinline__ CoefA up(int d) const { return {shfl_up4(v, d)}; }
};
template <> struct CoefA<false> {
  float v;
  __device__ __forceinline__ void zero() { v = 0.f; }
  __device__ __forceinline__ void add(const CoefA &o) { v += o.v; }
  __device__ __forceinline__ float4 vec() const { return splat(v); }
  __device__ __forceinline__ CoefA up(int d) const { return {__shfl_up(v, d)}; }
};

#ifdef FMX_STAMPS  // diagnostic build (tools/update_stamps.sh): s_memrealtime (100 MHz) of every tile wave of the LAST k_fm_update launch
__device__ unsigned long long g_upd_stamps[8192 * 6];
#define FMX_STAMP(slot_, dep_)                                                                                      \
  do {                                                                                                              \
    unsigned long long t_;                                                                                          \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "v"(dep_) : "memory");                      \
    if (lane == 0 && gt < 8192) g_upd_stamps[(size_t)gt * 6 + (slot_)] = t_;                                         \
  } while (0)
#else
#define FMX_STAMP(slot_, dep_) do {} while (0)
#endif

// One wave per tile of 64 sorted occurrences of one field.  Lane group s (LPR lanes; lane q owns coordinates 4q..4q+3)
// walks EPG = 64 / SLOTS CONSECUTIVE occurrences sequentially, so duplicates inside a group are summed in registers;
// one segmented scan over the SLOTS groups (log2(SLOTS) steps of wave shuffles) carries the sums of runs that span
// groups.  At the tail of a run: the row update when the run began in this tile, a partial record otherwise.
template <int LPR, int LAYOUT, int RULE, bool HAS_GBI, bool INL>
__device__ __forceinline__ void update_body(const UpdArgs &a, const int blk) {
  constexpr int SLOTS = WAVE / LPR;  // lane groups
  constexpr int EPG = LPR;           // consecutive occurrences per group
  constexpr int REC = 2 * LPR * 4 + 4;
  constexpr bool PREFETCH_ROWS = EPG <= 4;
  using CA = CoefA<HAS_GBI>;
  const int n_red = red_slices(a.B);
  if (blk < n_red) {  // the first blocks (dispatched first) own the bias and the loss reduction, one slice of the batch each
    bias_and_loss<LAYOUT, RULE, INL>(a, blk);
    return;
  }
  const int lane = threadIdx.x & 63;
  const int slot = lane / LPR, q = lane % LPR;
  const int kp = LPR * 4;
  const int tiles_per_field = a.Bp >> 6;
  const int gt = (blk - n_red) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (gt >= a.F * tiles_per_field) return;
  const int f = gt / tiles_per_field;
  const int base = (gt - f * tiles_per_field) << 6;
  const uint32_t *sf = a.sorted + (size_t)f * a.Bp;
  const int bbits = a.bbits;
  const uint32_t bmask = (1u << bbits) - 1u;
  const int e0 = base + slot * EPG;
  FMX_STAMP(0, lane);

  uint32_t c[EPG];
  if constexpr (EPG % 4 == 0) {  // 16-byte loads (e0 is a multiple of EPG)
#pragma unroll
    for (int j = 0; j < EPG; j += 4) {
      const uint4 t = *reinterpret_cast<const uint4 *>(sf + e0 + j);
      c[j] = t.x;
      c[j + 1] = t.y;
      c[j + 2] = t.z;
      c[j + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < EPG; ++j) c[j] = sf[e0 + j];
  }
  const uint32_t kprev = (e0 == 0 ? SENT : sf[e0 - 1]) >> bbits;
  const uint32_t knext = (e0 + EPG < a.Bp ? sf[e0 + EPG] : SENT) >> bbits;
  const uint32_t tile_prevkey = (base == 0 ? SENT : sf[base - 1]) >> bbits;
  const size_t row0 = (size_t)a.foff[f];
  float *part = a.parts + (size_t)gt * 2 * REC;

  uint32_t k[EPG];
  bool val[EPG], tail[EPG];
#pragma unroll
  for (int j = 0; j < EPG; ++j) {
    k[j] = c[j] >> bbits;
    val[j] = c[j] != SENT;
  }
#pragma unroll
  for (int j = 0; j < EPG; ++j) tail[j] = val[j] && (k[j] != (j + 1 < EPG ? k[j + 1] : knext));
  FMX_STAMP(1, k[0] + knext + kprev + tile_prevkey);  // the sorted list has arrived

  // ---- issue the loads: rows of the runs that end here (HBM / MALL), then S of every occurrence (L2) ----
  // Branch-free: an occurrence that is not the tail of a run starting in this tile requests the field's FIRST row instead
  // (one address for all such lanes: one request per instruction) and never looks at the result.  With
  // `if (tail) row[j] = load_row(...)` the compiler closed every j's region with s_waitcnt vmcnt(0): the four row requests
  // of a lane group -- HBM / Infinity Cache round trips -- went out one after the other.
  RowRegs row[PREFETCH_ROWS ? EPG : 1];
  // INL: the row of the run that comes in from the previous tile (updated by THIS wave if the run ends here, by nobody
  // else in this launch) is requested with the other rows, instead of behind this wave's own stores
  RowRegs row_in;
  row_in.v = row_in.z = row_in.n = row_in.fo = splat(0.f);
  const bool run_comes_in = INL && base > 0 && val[0] && k[0] == tile_prevkey;  // meaningful in lane group 0
  auto request_rows = [&]() {
    if (PREFETCH_ROWS) {
#pragma unroll
      for (int j = 0; j < EPG; ++j) {
        const bool need = tail[j] && k[j] != tile_prevkey;
        row[j] = load_row<LAYOUT>(a.rows + (row0 + (need ? k[j] : 0u)) * (size_t)a.stride, q, kp, a.zoff);
      }
    }
    if (INL)  // (branch-free like the rows above; used by lane group 0 of a closing tile only)
      row_in = load_row<LAYOUT>(a.rows + (row0 + ((slot == 0 && run_comes_in) ? tile_prevkey : 0u)) * (size_t)a.stride, q, kp, a.zoff);
  };
  // A tile whose last run goes on into the next tile PUBLISHES its partial sums (below) for the tile that closes the run,
  // and the publication waits for everything this wave has in flight (s_waitcnt vmcnt(0) before the flag): such a tile
  // requests its rows only AFTER it has published -- its sums need S and dlogit (L2), not the rows (HBM / Infinity Cache).
  const bool tile_open_early = INL && __shfl((int)(val[EPG - 1] && k[EPG - 1] == knext), WAVE - 1) != 0;  // wave-uniform
  // Branch-free like the rows: a padding entry reads sample 0 and its contribution is dropped by a select.  (`if (val[j])
  // { loads; products }` closed every j's region with s_waitcnt vmcnt(0): four dependent L2 round trips per lane group.)
  float4 cV[EPG];
  CA cA[EPG];
  float cw[EPG];
  {
    const bool has_x = a.xv != nullptr;
    const float *xsrc = has_x ? a.xv : a.dz_first;  // something loadable
    const int fld = (has_x && a.cols) ? a.cols[f] : f;
    const int col = (has_x && a.fcols) ? a.fcols[fld] : fld;
    float4 S4[EPG], G4[EPG];
    float xl[EPG], dzf[EPG], dzbl[EPG];
    uint32_t bj[EPG];
#pragma unroll
    for (int j = 0; j < EPG; ++j) bj[j] = val[j] ? (c[j] & bmask) : 0u;
    // What the common callers do not need is not requested: feature values when there are none, and the bi-interaction's
    // coefficient when it is the first-order one (pure FM, DeepFM: dz_bi == dz_first; NFM: none).  Wave-uniform branches AHEAD
    // of the other requests: the wait the compiler puts at their joins covers nothing else.  (No measurable change of the
    // launch: the 2.5 us between the list's arrival and the arrival of S / dlogit / rows -- in-kernel stamps,
    // tools/update_stamps.sh -- are the ~110 K distinct row lines of a step at the chip's ~54 G random lines per second.)
    const bool sep_dzbi = a.dz_bi != nullptr && a.dz_bi != a.dz_first;
#pragma unroll
    for (int j = 0; j < EPG; ++j) {
      xl[j] = 1.f;
      dzbl[j] = 0.f;
    }
    if (has_x) {
#pragma unroll
      for (int j = 0; j < EPG; ++j) xl[j] = xsrc[(size_t)bj[j] * a.Fx + col];
    }
    if (sep_dzbi) {
#pragma unroll
      for (int j = 0; j < EPG; ++j) dzbl[j] = a.dz_bi[(size_t)bj[j] * a.ld1];
    }
#pragma unroll
    for (int j = 0; j < EPG; ++j) {
      const uint32_t b = bj[j];
      S4[j] = *reinterpret_cast<const float4 *>(a.S + (size_t)b * a.ldS + 4 * q);
      dzf[j] = a.dz_first[(size_t)b * a.ld1];
      if constexpr (HAS_GBI) G4[j] = *reinterpret_cast<const float4 *>(a.gbi + (size_t)b * a.ldG + 4 * q);
      else G4[j] = splat(0.f);
    }
    if (!tile_open_early) request_rows();
#pragma unroll
    for (int j = 0; j < EPG; ++j) {
      const float x = xl[j];
      const float dzb = a.dz_bi ? (sep_dzbi ? dzbl[j] : dzf[j]) : 0.f;
      const float w1 = x * dzf[j];
      float4 v;
      CA ca;
      if constexpr (HAS_GBI) {
        const float4 G = splat(dzb) + G4[j];
        const float4 xG = x * G;
        v = xG * S4[j];
        ca.v = x * xG;
      } else {
        const float xG = x * dzb;
        v = xG * S4[j];
        ca.v = x * xG;
      }
      cV[j] = splat(0.f);
      cA[j].zero();
      cw[j] = 0.f;
      if (val[j]) {  // selects
        cV[j] = v;
        cA[j] = ca;
        cw[j] = w1;
      }
    }
  }

  // ---- pass 1: the sum of the group's last run, and whether the group lies inside one longer run ----
  float4 tV = splat(0.f);
  CA tA;
  tA.zero();
  float tw = 0.f;
  bool uniform = true;
#pragma unroll
  for (int j = 0; j < EPG; ++j) {
    if (j > 0 && k[j] != k[j - 1]) {
      tV = splat(0.f);
      tA.zero();
      tw = 0.f;
      uniform = false;
    }
    tV = tV + cV[j];
    tA.add(cA[j]);
    tw += cw[j];
  }
  const bool lead_open = val[0] && k[0] == kprev;
  const bool trail_open = val[EPG - 1] && k[EPG - 1] == knext;
  bool pass = uniform && lead_open && trail_open;
  if (!trail_open) {
    tV = splat(0.f);
    tA.zero();
    tw = 0.f;
  }
  // ---- segmented scan over the groups: carry_out(s) = v(s) + (pass(s) ? carry_out(s-1) : 0) ----
#pragma unroll
  for (int off = 1; off < SLOTS; off <<= 1) {
    const float4 uV = shfl_up4(tV, off * LPR);
    const CA uA = tA.up(off * LPR);
    const float uw = __shfl_up(tw, off * LPR);
    const bool up = __shfl_up((int)pass, off * LPR) != 0;
    if (slot >= off) {
      if (pass) {
        tV = tV + uV;
        tA.add(uA);
        tw += uw;
      }
      pass = pass && up;
    }
  }
  // carry into this group = carry out of the previous one
  float4 accV = shfl_up4(tV, LPR);
  CA accA = tA.up(LPR);
  float accw = __shfl_up(tw, LPR);
  if (slot == 0 || !lead_open) {
    accV = splat(0.f);
    accA.zero();
    accw = 0.f;
  }

  // ---- the tile's hand-off state is known before any row is touched: does the run that came in end here (this tile
  //      CLOSES it), does the tile lie inside one run (THROUGH), does its last run go on (trail)?  The sum of that last
  //      run so far is the last group's carry-out.  With the in-launch hand-off (INL) the record and the flag word
  //      (launch sequence << 4 | lead_state << 2 | trail_state) are published NOW, before the row updates of pass 2, so
  //      that closing tiles further on never wait for this tile's FTRL arithmetic and store acknowledgements: records
  //      write-through, s_waitcnt vmcnt(0), then the flag as an agent-scope atomic; records and flags sc1 on both sides.
  bool closes_here = false;
#pragma unroll
  for (int j = 0; j < EPG; ++j) closes_here = closes_here || (tail[j] && k[j] == tile_prevkey);
  int lead_state = __ballot(closes_here) != 0ull ? LEAD_CLOSES : LEAD_NONE;
  int trail_state = 0;
  const bool tile_open = __shfl((int)trail_open, WAVE - 1) != 0;
  if (tile_open) {
    const bool through = __shfl((int)(k[EPG - 1] == tile_prevkey), WAVE - 1) != 0;
    if (through) lead_state = LEAD_THROUGH;  // the whole tile is one run, open at both ends
    else trail_state = 1;
    if (slot == SLOTS - 1) {
      if (INL) store_part_sc1(part + (through ? 0 : REC), q, kp, tV, tA.vec(), tw);
      else store_part(part + (through ? 0 : REC), q, kp, tV, tA.vec(), tw);
    }
  }
  if (INL) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FMX_STAMP(2, lane);  // every load issued so far (S / dlogit, the rows unless the tile publishes first) has arrived
    if (lane == 0)
      __hip_atomic_store(a.meta + (size_t)gt * 2, (int32_t)((a.seq << 4) | ((uint32_t)lead_state << 2) | (uint32_t)trail_state),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if (lane == 0) {
    a.meta[(size_t)gt * 2] = lead_state;
    a.meta[(size_t)gt * 2 + 1] = trail_state;
  }

  if (tile_open_early) request_rows();
  // ---- pass 2: walk the occurrences again; at the tail of a run apply the update or leave a partial ----
  float4 leadV = splat(0.f), leadA = splat(0.f);  // INL: this tile's part of the run that came in and ends here
  float leadw = 0.f;
  bool have_lead = false;
#pragma unroll
  for (int j = 0; j < EPG; ++j) {
    if (j > 0 && k[j] != k[j - 1]) {
      accV = splat(0.f);
      accA.zero();
      accw = 0.f;
    }
    accV = accV + cV[j];
    accA.add(cA[j]);
    accw += cw[j];
    if (tail[j]) {
      if (k[j] != tile_prevkey) {
        float *rp = a.rows + (row0 + k[j]) * (size_t)a.stride;
        const RowRegs r = PREFETCH_ROWS ? row[PREFETCH_ROWS ? j : 0] : load_row<LAYOUT>(rp, q, kp, a.zoff);
        update_row<LAYOUT, RULE>(rp, q, kp, a.zoff, r, accV, accA.vec(), accw, a.h);
      } else {
        // the run that came in from the previous tile ends here: its part inside this tile stays in registers for this
        // wave's combine below (INL), or goes to memory for k_fm_fixup
        if (INL) {
          leadV = accV;
          leadA = accA.vec();
          leadw = accw;
          have_lead = true;
        } else {
          store_part(part, q, kp, accV, accA.vec(), accw);
        }
      }
    }
  }
  FMX_STAMP(3, lane);  // pass 2 done: the row updates of the runs inside the tile are issued
  if (!INL) return;
  // ---- in-launch hand-off (INL): the CLOSING tile of a run sums the records of the tiles before it (they were
  //      dispatched earlier and wait on nothing) and applies the row update -- no second launch ----
  if (lead_state != LEAD_CLOSES) return;  // wave-uniform
  const int t = gt - f * tiles_per_field;
  float *rp = a.rows + (row0 + tile_prevkey) * (size_t)a.stride;
  const RowRegs r = row_in;  // lanes < LPR: requested at the top (the row is final until this wave writes it)
  // this tile's own part of the run: from the registers of the lane group that closed it to every lane group's lane q
  const int src0 = __ffsll((long long)__ballot(have_lead)) - 1;  // first lane of that group (q == 0)
  const float4 ownV = shfl4(leadV, src0 + q), ownA = shfl4(leadA, src0 + q);
  const float ownw = __shfl(leadw, src0);
  // distance m to the head tile: tiles t-1, t-2, ... are THROUGH until the head (trail_state == 1)
  int m = 0;
  bool failed = false;
  for (int j0 = 1; j0 <= t && m == 0 && !failed; j0 += 64) {
    const int tj = t - j0 - lane;  // lane i looks at tile t - j0 - i
    int w = 0;
    bool ready = false;
    for (int spin = 0;; ++spin) {
      if (tj >= 0 && !ready) {
        w = __hip_atomic_load(a.meta + ((size_t)f * tiles_per_field + tj) * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ready = ((uint32_t)w >> 4) == (a.seq & 0x0FFFFFFFu);
      }
      const bool is_through = tj >= 0 && ready && ((w >> 2) & 3) == LEAD_THROUGH;
      const unsigned long long stop = __ballot(!is_through);  // not published yet, or not THROUGH, or before the field
      if (stop == 0ull) break;                                // 64 THROUGH tiles: look further back
      const int pos = __ffsll((long long)stop) - 1;
      const bool resolved = __shfl((int)(ready || tj < 0), pos) != 0;
      if (resolved) {  // the chain ends at a published tile: it must be the head (its last run goes on)
        if (__shfl((int)(tj >= 0 && (w & 3) == 1), pos) != 0) m = j0 + pos;
        else failed = true;
        break;
      }
      if (spin >= (1 << 20)) {
        failed = true;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  if (failed || m == 0 || m > t) {
    if (lane == 0 && a.error) *a.error = 2;
    return;
  }
  // the same record order and lane-group assignment as k_fm_fixup, so both modes give identical bits
  const size_t gh = (size_t)gt - m;
  float4 aV = splat(0.f), aA = splat(0.f);
  float aw = 0.f;
  for (int j = slot; j <= m; j += SLOTS) {
    if (j == m) {  // this tile's own record: the last one of its lane group, as in k_fm_fixup's order
      aV = aV + ownV;
      aA = aA + ownA;
      aw += ownw;
    } else {
      const float *rec = j == 0 ? a.parts + (gh * 2 + 1) * REC : a.parts + (gh + j) * 2 * REC;
      aV = aV + ld_sc1_4(rec + 4 * q);
      aA = aA + ld_sc1_4(rec + kp + 4 * q);
      aw += ld_sc1(rec + 2 * kp);
    }
  }
#pragma unroll
  for (int mm = LPR; mm < WAVE; mm <<= 1) {
    aV = aV + shfl_xor4(aV, mm);
    aA = aA + shfl_xor4(aA, mm);
    aw += __shfl_xor(aw, mm);
  }
  if (lane < LPR) update_row<LAYOUT, RULE>(rp, q, kp, a.zoff, r, aV, aA, aw, a.h);
  FMX_STAMP(4, lane);  // a closing tile: the crossing run's row is updated
}

template <int LPR, int LAYOUT, int RULE, bool HAS_GBI, bool INL>
__global__ __launch_bounds__(256) void k_fm_update(UpdArgs a) {
  __builtin_amdgcn_s_setprio(3);  // ahead of the side-stream sort's waves at the CU's instruction arbiter
  update_body<LPR, LAYOUT, RULE, HAS_GBI, INL>(a, blockIdx.x);
}

// The same launch with a RIDER: the workgroups behind the update's own carry the fixed-order reduction of the MLP's partial weight
// gradients (mlp_reduce_block: a few hundred latency-bound waves that the table update neither feeds nor needs -- both wait only for
// the launches in front).  fmx_deepfm_stream: one launch and 6 - 7 us less per step than k_mlp_reduce as a launch of its own in front
// of the update; on a second stream the same overlap lost to the cross-stream hand-off.  Identical results.
template <int LPR, int LAYOUT, int RULE, bool HAS_GBI>
__global__ __launch_bounds__(256) void k_fm_update_rider(UpdArgs a, MlpReduceArgs r, int n_update_blocks, int rider_blocks_per_layer) {
  __builtin_amdgcn_s_setprio(3);
  if ((int)blockIdx.x >= n_update_blocks) {
    const int rb = (int)blockIdx.x - n_update_blocks;
    mlp_reduce_block(r, rb / rider_blocks_per_layer, rb % rider_blocks_per_layer, rider_blocks_per_layer);
    return;
  }
  update_body<LPR, LAYOUT, RULE, HAS_GBI, true>(a, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------------------
// k_fm_online: the reference's online protocol on a device-resident stream (pure FM)
// ------------------------------------------------------------------------------------------------------------
// run_experiment (reference fm_adam.py:90-119): for every sample, predict (sigmoid(forward) > 0.5), then fit on that
// one sample.  Steps of one sample are inherently sequential (step i+1 reads the rows step i wrote), so ONE wavefront
// walks the stream: per sample it gathers the F rows (sc1 loads: a row may have been written by the previous sample),
// evaluates the logit, stores the prediction, and applies the rule to the same rows from registers -- the arithmetic of
// k_fm_forward + k_fm_update at B = 1 (each row of a sample is a run of one occurrence), so the tables end
// bit-identical to N calls of fmx_fm_step with B = 1.  The next sample's indices are fetched while the current one is
// processed; the bias lives in registers.  Per sample: one dependent gather + the store acknowledgement (~3 us).
struct OnlineArgs {
  float *rows;
  const int64_t *foff;
  float *bias;
  const int32_t *idx;  // [N, F]
  const float *xv;     // [N, F] or null
  const float *y;      // [N]
  uint8_t *pred;       // [N] sigmoid(logit) > 0.5 BEFORE the sample's update
  float *loss;         // [N] or null
  int32_t *error;
  fmx_hyper_t h;
  int32_t N, F, stride, zoff, loss_kind;
};

template <int LAYOUT>
__device__ __forceinline__ RowRegs load_row_sc1(const float *rp, int q, int kp, int zoff) {
  RowRegs r;
  r.v = ld_sc1_4(rp + 4 * q);
  r.z = splat(0.f);
  r.n = splat(0.f);
  r.fo = splat(0.f);
  if (LAYOUT == FMX_LAYOUT_WEIGHTS) {  // (every lane, as in load_row: no branch around a load)
    r.fo.x = ld_sc1(rp + kp);
  } else {
    r.z = ld_sc1_4(rp + zoff + 4 * q);
    r.n = ld_sc1_4(rp + zoff + kp + 4 * q);
    r.fo = ld_sc1_4(rp + kp);
  }
  return r;
}

template <int LPR, int LAYOUT, int RULE, int NP>
__global__ __launch_bounds__(64) void k_fm_online(OnlineArgs a) {
  constexpr int SLOTS = WAVE / LPR;
  const int lane = threadIdx.x & 63;
  const int slot = lane / LPR, q = lane % LPR;
  const int kp = LPR * 4;
  float b0 = a.bias[0], b1 = LAYOUT == FMX_LAYOUT_FTRL ? a.bias[1] : 0.f;  // the bias (or its (z, n)) stays in registers
  int64_t lo[NP];
  uint32_t vocab[NP];
  bool live[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int f = p * SLOTS + slot;
    live[p] = f < a.F;
    lo[p] = live[p] ? a.foff[f] : 0;
    vocab[p] = live[p] ? (uint32_t)(a.foff[f + 1] - lo[p]) : 0u;
  }
  uint32_t li_n[NP];
  float x_n[NP], y_n = 0.f;
  // branch-free (see forward_sample): beyond the stream or the last field the loads read element 0 and are dropped
  const float *xsrc = a.xv ? a.xv : reinterpret_cast<const float *>(a.idx);
  const bool has_x = a.xv != nullptr;
  auto fetch_inputs = [&](int i) {
    const bool in = i < a.N;
    uint32_t l_[NP];
    float x_[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const size_t o = (live[p] && in) ? (size_t)i * a.F + p * SLOTS + slot : (size_t)0;
      l_[p] = (uint32_t)a.idx[o];
      x_[p] = xsrc[o];
    }
    const float yy = a.y[in ? i : 0];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      li_n[p] = (live[p] && in) ? l_[p] : 0u;
      x_n[p] = (has_x && live[p] && in) ? x_[p] : 1.f;
    }
    y_n = in ? yy : 0.f;
  };
  fetch_inputs(0);
  bool bad = false;
  for (int i = 0; i < a.N; ++i) {
    uint32_t li[NP];
    float x[NP];
    const float y = y_n;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      li[p] = li_n[p];
      x[p] = x_n[p];
    }
    RowRegs row[NP];
    bool ok[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      ok[p] = live[p] && li[p] < vocab[p];
      // branch-free: a dead lane group or a bad index requests the table's first row and drops it (with a branch per
      // pass the rows of a sample went out in NP dependent round trips)
      row[p] = load_row_sc1<LAYOUT>(a.rows + (size_t)(ok[p] ? lo[p] + li[p] : 0) * a.stride, q, kp, a.zoff);
      bad = bad || (live[p] && !ok[p]);
    }
    fetch_inputs(i + 1);  // independent of the weights: in flight while this sample is processed
    // ---- forward: the arithmetic of k_fm_forward ----
    float4 s = splat(0.f), ss = splat(0.f);
    float fo = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (ok[p]) {
        const float4 e = x[p] * row[p].v;
        s = s + e;
        ss = ss + e * e;
        fo += row[p].fo.x * x[p];
      }
    }
#define FMX_BFLY(M)                               \
  if (LPR <= M) {                                 \
    s = s + xor_lane_f4<M>(s, lane);              \
    ss = ss + xor_lane_f4<M>(ss, lane);           \
    fo += xor_lane_f<M>(fo, lane);                \
  }
    FMX_BFLY(1) FMX_BFLY(2) FMX_BFLY(4) FMX_BFLY(8) FMX_BFLY(16) FMX_BFLY(32)
#undef FMX_BFLY
    const float4 bi = 0.5f * (s * s - ss);
    float sbi = (bi.x + bi.y) + (bi.z + bi.w);
#pragma unroll
    for (int m = 1; m < LPR; m <<= 1) sbi += __shfl_xor(sbi, m);
    fo = __shfl(fo, 0);
    const float bias_w = LAYOUT == FMX_LAYOUT_WEIGHTS ? b0 : ftrl_w(b0, b1, a.h);
    const float z = fo + sbi + bias_w;
    float loss, dz;
    if (a.loss_kind == FMX_LOSS_BCE_LOGITS) {
      loss = bcewl(z, y);
      dz = sigmoidf_(z) - y;
    } else {
      const float pz = sigmoidf_(z);
      loss = bcewl(pz, y);
      dz = (sigmoidf_(pz) - y) * pz * (1.f - pz);
    }
    if (lane == 0) {
      a.pred[i] = sigmoidf_(z) > 0.5f ? 1 : 0;
      if (a.loss) a.loss[i] = loss;
    }
    // ---- fit: every row of the sample is a run of one occurrence (k_fm_update's sums with B = 1, inv_b = 1) ----
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (ok[p]) {
        const float xG = x[p] * dz;
        update_row<LAYOUT, RULE>(a.rows + (size_t)(lo[p] + li[p]) * a.stride, q, kp, a.zoff, row[p], xG * s, splat(x[p] * xG),
                                 xG, a.h);
      }
    }
    if (LAYOUT == FMX_LAYOUT_WEIGHTS) {
      b0 = apply_rule<RULE>(b0, dz, a.h);
    } else {
      const float w = ftrl_w(b0, b1, a.h);
      ftrl_upd(b0, b1, w, dz, a.h);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the row stores are acknowledged before the next sample's loads
  }
  const bool any_bad = __ballot(bad) != 0ull;  // an out-of-range index seen by any lane group
  if (lane == 0) {
    a.bias[0] = b0;
    if (LAYOUT == FMX_LAYOUT_FTRL) a.bias[1] = b1;
    if (any_bad && a.error) *a.error = 1;
  }
}

// Runs that cross tile boundaries: the wave of the tile holding the run's head adds the partial sums in tile order
// (trail of the head tile, then the lead partial of every following tile up to the one where the run ends) and
// applies the row update.
template <int LPR, int LAYOUT, int RULE>
__global__ __launch_bounds__(256) void k_fm_fixup(UpdArgs a) {
  constexpr int SLOTS = WAVE / LPR;
  constexpr int REC = 2 * LPR * 4 + 4;
  const int lane = threadIdx.x & 63;
  const int slot = lane / LPR, q = lane % LPR;
  const int kp = LPR * 4;
  const int tiles_per_field = a.Bp >> 6;
  if (blockIdx.x == gridDim.x - 1 && red_slices(a.B) > 1) {  // (an extra workgroup: the slices' bias / loss partials, in slice order)
    finish_bias_and_loss<LAYOUT, RULE>(a, red_slices(a.B), false);
    return;
  }
  const int gt = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (gt >= a.F * tiles_per_field) return;
  const int f = gt / tiles_per_field;
  const int t = gt - f * tiles_per_field;
  // three dependent round trips instead of five: the head flag, the run's key (last entry of the tile) and the meta of
  // the following tiles are loaded together; the row is requested as soon as the key is known, before the partial
  // records are summed
  const int my_trail = a.meta[(size_t)gt * 2 + 1];
  const uint32_t key = a.sorted[(size_t)f * a.Bp + ((size_t)t << 6) + 63] >> a.bbits;
  const int tj0 = t + 1 + lane;
  const int st0 = tj0 < tiles_per_field ? a.meta[((size_t)f * tiles_per_field + tj0) * 2] : LEAD_NONE;
  if (my_trail != 1) return;  // wave-uniform
  float *rp = a.rows + ((size_t)a.foff[f] + key) * (size_t)a.stride;
  RowRegs r;
  if (lane < LPR) r = load_row<LAYOUT>(rp, q, kp, a.zoff);
  // m = number of following tiles that hold a piece of the run
  int m = 0;
  for (int j0 = 1; t + j0 < tiles_per_field; j0 += 64) {
    const int tj = t + j0 + lane;
    const int st = j0 == 1 ? st0 : (tj < tiles_per_field ? a.meta[((size_t)f * tiles_per_field + tj) * 2] : LEAD_NONE);
    const unsigned long long stop = __ballot(st != LEAD_THROUGH);
    if (stop != 0ull) {
      const int pos = __ffsll((long long)stop) - 1;
      const int st_pos = __shfl(st, pos);
      m = j0 + pos - (st_pos == LEAD_CLOSES ? 0 : 1);
      break;
    }
    m = j0 + 63;
  }
  if (t + m >= tiles_per_field) m = tiles_per_field - 1 - t;
  float4 aV = splat(0.f), aA = splat(0.f);
  float aw = 0.f;
  for (int j = slot; j <= m; j += SLOTS) {
    const float *rec = j == 0 ? a.parts + ((size_t)gt * 2 + 1) * REC : a.parts + (size_t)(gt + j) * 2 * REC;
    aV = aV + *reinterpret_cast<const float4 *>(rec + 4 * q);
    aA = aA + *reinterpret_cast<const float4 *>(rec + kp + 4 * q);
    aw += rec[2 * kp];
  }
#pragma unroll
  for (int mm = LPR; mm < WAVE; mm <<= 1) {
    aV = aV + shfl_xor4(aV, mm);
    aA = aA + shfl_xor4(aA, mm);
    aw += __shfl_xor(aw, mm);
  }
  if (lane < LPR) update_row<LAYOUT, RULE>(rp, q, kp, a.zoff, r, aV, aA, aw, a.h);
}

// ------------------------------------------------------------------------------------------------------------
// k_mlp_small: the relu MLP on top of the bi-interaction vector for the online (small batch) steps of DeepFM / NFM
// (reference deepfm_adam.py:82-88,106-119) and the ONN classes' Hedge backprop (deepfm_onn.py:88-154).
// One workgroup does forward, loss, backward and the parameter update of every layer: at B = 1 the reference's autograd
// graph is ~40 ATen launches and a fresh Adam over the hidden layers; here it is one launch.  Limits (host-checked):
// B <= 16, k <= 64, hidden <= 64, layers <= 8; larger shapes stay on the caller's PyTorch path.
// ------------------------------------------------------------------------------------------------------------
constexpr int MLP_MAX_B = 16, MLP_MAX_W = 64, MLP_MAX_L = 8;

enum { MLP_MODE_FORWARD = 0, MLP_MODE_FIT = 1, MLP_MODE_HEDGE = 2 };

struct MlpArgs {
  float *params;  // packed: per layer W [out, in] row-major, then b [out]
  const float *bi;     // [B, kp]
  const float *base;   // [B] logit without the MLP term
  const float *y;      // [B]
  float *alpha;        // HEDGE: [L] in/out
  float *dz_out;       // FIT: [B]
  float *gbi_out;      // FIT: [B, kp]
  float *out;          // FORWARD: [B] logit (adam classes) ; FIT: [1] mean loss or null
  float *layers_out;   // FORWARD: [L, B] sigmoid(base + sum x_l) or null ; HEDGE: [L] losses or null
  float *pred_out;     // FIT: [B] the logit, HEDGE: [B] sigmoid(last layer's logit) -- what forward() returns, before the update; or null
  const float *base_bias;  // null, or the table's bias words: base[b] + bias weight is the logit without the MLP term (NFM)
  int32_t base_bias_ftrl;  // base_bias holds (z, n) of an FTRL table instead of the weight
  fmx_hyper_t h;       // lr / eps (FIT: rule) ; HEDGE: lr = n
  fmx_hyper_t h_table; // base_bias_ftrl: the table's FTRL hyper-parameters (alpha already inverted)
  float hedge_b, hedge_s;
  int32_t B, k, kp, hidden, n_layers, mode, rule, loss_kind;
  float inv_b;
};

__device__ __forceinline__ int mlp_in(const MlpArgs &a, int l) { return l == 0 ? a.k : a.hidden; }
__device__ __forceinline__ float *mlp_w(const MlpArgs &a, int l) {
  size_t off = 0;
  for (int i = 0; i < l; ++i) off += (size_t)a.hidden * mlp_in(a, i) + a.hidden;
  return a.params + off;
}

// the body of k_mlp_small; also called once per sample by k_online_mlp (every pointer may then point into LDS)
__device__ void mlp_small_body(const MlpArgs &a) {
  __shared__ float acts[(MLP_MAX_L + 1) * MLP_MAX_B * MLP_MAX_W];  // x_0 .. x_L, [l][b][j]
  __shared__ float dA[MLP_MAX_B * MLP_MAX_W], dB[MLP_MAX_B * MLP_MAX_W];  // d x_l (ping-pong); dA is reused as d pre
  __shared__ float dout[MLP_MAX_L * MLP_MAX_B];  // d loss / d (out_l[b]) per layer (HEDGE) or for the last layer (FIT)
  __shared__ float lsum[MLP_MAX_L];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int B = a.B, H = a.hidden, L = a.n_layers;
  auto X = [&](int l, int b, int j) -> float & { return acts[((size_t)l * MLP_MAX_B + b) * MLP_MAX_W + j]; };

  for (int i = tid; i < B * a.k; i += nt) X(0, i / a.k, i % a.k) = a.bi[(size_t)(i / a.k) * a.kp + (i % a.k)];
  __syncthreads();
  // ---- forward ----
  for (int l = 0; l < L; ++l) {
    const int in = mlp_in(a, l);
    const float *W = mlp_w(a, l), *bias = W + (size_t)H * in;
    for (int i = tid; i < B * H; i += nt) {
      const int b = i / H, j = i % H;
      float s = bias[j];
      for (int c = 0; c < in; ++c) s += W[(size_t)j * in + c] * X(l, b, c);
      X(l + 1, b, j) = fmaxf(s, 0.f);
    }
    __syncthreads();
  }
  // ---- per-layer outputs, losses and d loss / d out ----
  if (tid < L) lsum[tid] = 0.f;
  __syncthreads();
  if (tid < L * B) {
    const int l = tid / B, b = tid % B;  // layer l+1's output for sample b
    const bool need = a.mode == MLP_MODE_HEDGE || l == L - 1 || (a.mode == MLP_MODE_FORWARD && a.layers_out);
    float g = 0.f;
    if (need) {
      float s = 0.f;
      for (int j = 0; j < H; ++j) s += X(l + 1, b, j);
      float base_b = a.base[b];
      if (a.base_bias) base_b += a.base_bias_ftrl ? ftrl_w(a.base_bias[0], a.base_bias[1], a.h_table) : a.base_bias[0];
      const float z = base_b + s;
      if (a.pred_out && l == L - 1 && a.mode != MLP_MODE_FORWARD) a.pred_out[b] = a.mode == MLP_MODE_HEDGE ? sigmoidf_(z) : z;
      if (a.mode == MLP_MODE_FORWARD) {
        if (a.layers_out) a.layers_out[(size_t)l * B + b] = sigmoidf_(z);
        if (l == L - 1 && a.out) a.out[b] = z;
      } else if (a.mode == MLP_MODE_FIT) {
        const float yy = a.y[b];
        float loss;
        if (a.loss_kind == FMX_LOSS_BCE_LOGITS) {
          loss = bcewl(z, yy);
          g = (sigmoidf_(z) - yy) * a.inv_b;
        } else {
          const float p = sigmoidf_(z);
          loss = bcewl(p, yy);
          g = (sigmoidf_(p) - yy) * p * (1.f - p) * a.inv_b;
        }
        a.dz_out[b] = g;
        X(0, b, MLP_MAX_W - 1) = loss;  // parked for the ordered sum below (k <= 63 is host-checked in FIT mode)
      } else {  // HEDGE: BCELoss(sigmoid(z), y), mean over the batch; d/dz = (p - y) / B
        const float yy = a.y[b];
        const float p = sigmoidf_(z);
        const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(log1pf(-p), -100.f);
        X(0, b, MLP_MAX_W - 1 - l) = -(yy * lp + (1.f - yy) * l1p);  // parked per layer
        // autograd through nn.BCELoss then sigmoid: (p - y) / max(p (1 - p), 1e-12) * p (1 - p) -- NOT (p - y) once p
        // saturates (p == 1.0f gives exactly 0, as in the reference)
        const float pq = p * (1.f - p);
        g = a.alpha[l] * ((p - yy) / fmaxf(pq, 1e-12f)) * pq * a.inv_b;
      }
    }
    dout[l * MLP_MAX_B + b] = g;
  }
  __syncthreads();
  if (a.mode == MLP_MODE_FORWARD) return;
  if (a.mode == MLP_MODE_FIT) {
    if (tid == 0 && a.out) {
      float s = 0.f;
      for (int b = 0; b < B; ++b) s += X(0, b, MLP_MAX_W - 1);
      a.out[0] = s * a.inv_b;
    }
  } else if (tid < L) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += X(0, b, MLP_MAX_W - 1 - tid);
    lsum[tid] = s * a.inv_b;
  }
  // ---- backward + update, top layer first ----
  float *dcur = dA, *dnext = dB;
  for (int i = tid; i < B * H; i += nt) dcur[(i / H) * MLP_MAX_W + (i % H)] = dout[(L - 1) * MLP_MAX_B + i / H];
  __syncthreads();
  for (int l = L - 1; l >= 0; --l) {
    const int in = mlp_in(a, l);
    float *W = mlp_w(a, l), *bias = W + (size_t)H * in;
    // d pre = d x_{l+1} * (x_{l+1} > 0), in place
    for (int i = tid; i < B * H; i += nt) {
      const int b = i / H, j = i % H;
      if (!(X(l + 1, b, j) > 0.f)) dcur[b * MLP_MAX_W + j] = 0.f;
    }
    __syncthreads();
    // d x_l = W^T d pre (+ this layer's own output gradient in HEDGE mode), with the OLD weights
    for (int i = tid; i < B * in; i += nt) {
      const int b = i / in, c = i % in;
      float s = 0.f;
      for (int j = 0; j < H; ++j) s += W[(size_t)j * in + c] * dcur[b * MLP_MAX_W + j];
      if (a.mode == MLP_MODE_HEDGE && l >= 1) s += dout[(l - 1) * MLP_MAX_B + b];
      dnext[b * MLP_MAX_W + c] = s;
    }
    __syncthreads();
    // parameter gradients (batch summed in sample order) and the update
    for (int i = tid; i < H * in + H; i += nt) {
      float g = 0.f;
      float *p;
      if (i < H * in) {
        const int j = i / in, c = i % in;
        for (int b = 0; b < B; ++b) g += dcur[b * MLP_MAX_W + j] * X(l, b, c);
        p = W + i;
      } else {
        const int j = i - H * in;
        for (int b = 0; b < B; ++b) g += dcur[b * MLP_MAX_W + j];
        p = bias + j;
      }
      if (a.mode == MLP_MODE_HEDGE || a.rule == FMX_RULE_SGD) *p = *p - a.h.lr * g;
      else *p = *p - a.h.lr * g * rcp_(fabsf(g) + a.h.eps);
    }
    __syncthreads();
    float *t = dcur;
    dcur = dnext;
    dnext = t;
  }
  if (a.mode == MLP_MODE_FIT) {
    for (int i = tid; i < B * a.kp; i += nt) {
      const int b = i / a.kp, c = i % a.kp;
      a.gbi_out[i] = c < a.k ? dcur[b * MLP_MAX_W + c] : 0.f;
    }
  } else if (tid == 0) {  // Hedge: alpha_i *= b^loss_i, floor s / L, normalise (deepfm_onn.py:147-154)
    float al[MLP_MAX_L], z = 0.f;
    for (int i = 0; i < L; ++i) {
      al[i] = fmaxf(a.alpha[i] * powf(a.hedge_b, lsum[i]), a.hedge_s / (float)L);
      z += al[i];
    }
    for (int i = 0; i < L; ++i) {
      a.alpha[i] = al[i] / z;
      if (a.layers_out) a.layers_out[i] = lsum[i];
    }
  }
}

__global__ __launch_bounds__(256) void k_mlp_small(MlpArgs a) { mlp_small_body(a); }

// ------------------------------------------------------------------------------------------------------------
// k_online_mlp: the online predict-then-fit loop of the classes with an MLP, one workgroup walking the stream
// ------------------------------------------------------------------------------------------------------------
// Per sample: wave 0 gathers the sample's rows (sc1 loads: the previous sample may have written them) and evaluates the
// FM part exactly as k_fm_forward does; the whole workgroup runs the MLP step of k_mlp_small (fit or Hedge) on parameters
// that live in LDS for the length of the stream; wave 0 then applies the table update of k_fm_update at B = 1 from the
// rows it still holds (not with Hedge, which leaves the tables alone).  Same arithmetic as the per-sample launches
// (forward, k_mlp_small, sort, update), so the parameters end bit-identical; no launch gaps, no host in the loop.
struct OnlineMlpArgs {
  float *rows;
  const int64_t *foff;
  float *bias;
  const int32_t *idx;
  const float *xv;
  const float *y;
  float *pred;      // [N] what forward() returns for the sample, before its update
  int32_t *error;
  float *params;    // global: copied into LDS, written back at the end
  float *alpha;     // Hedge: global [L], same treatment
  fmx_hyper_t h;    // alpha already inverted (table rule); lr / eps also drive the MLP rule
  float hedge_b, hedge_s;
  int32_t N, F, stride, zoff, n_params;
  int32_t k, hidden, n_layers, hedge, fm_term, rule, loss_kind;
};

template <int LPR, int LAYOUT, int RULE>
__global__ __launch_bounds__(256) void k_online_mlp(OnlineMlpArgs a) {
  constexpr int SLOTS = WAVE / LPR, NP = 4;
  extern __shared__ float p_lds[];  // [n_params] the MLP's parameters
  __shared__ float bi_lds[MLP_MAX_W], gbi_lds[MLP_MAX_W], alpha_lds[MLP_MAX_L];
  __shared__ float base_lds, dz_lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int slot = lane / LPR, q = lane % LPR;
  const int kp = LPR * 4;
  for (int i = tid; i < a.n_params; i += blockDim.x) p_lds[i] = a.params[i];
  if (a.hedge && tid < a.n_layers) alpha_lds[tid] = a.alpha[tid];
  float b0 = a.bias[0], b1 = LAYOUT == FMX_LAYOUT_FTRL ? a.bias[1] : 0.f;
  int64_t lo[NP];
  uint32_t vocab[NP];
  bool live[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int f = p * SLOTS + slot;
    live[p] = f < a.F;
    lo[p] = live[p] ? a.foff[f] : 0;
    vocab[p] = live[p] ? (uint32_t)(a.foff[f + 1] - lo[p]) : 0u;
  }
  bool bad = false;
  // wave 0: the NEXT sample's indices (and values) are requested while this one is processed -- they do not depend on the
  // weights (as in k_fm_online); branch-free loads (see forward_sample)
  const float *xsrc = a.xv ? a.xv : reinterpret_cast<const float *>(a.idx);
  const bool has_x = a.xv != nullptr;
  uint32_t l_n[NP];
  float x_n[NP], y_n = 0.f;
  __shared__ float y_lds;  // the sample's label, for the MLP step (its own load of y would be one more exposed round trip)
  auto fetch_inputs = [&](int i) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const size_t o = (live[p] && i < a.N) ? (size_t)i * a.F + p * SLOTS + slot : (size_t)0;
      l_n[p] = (uint32_t)a.idx[o];
      x_n[p] = xsrc[o];
    }
    y_n = a.y[i < a.N ? i : 0];
  };
  if (wave == 0) fetch_inputs(0);
  __syncthreads();
  for (int i = 0; i < a.N; ++i) {
    uint32_t li[NP];
    float x[NP];
    RowRegs row[NP];
    bool ok[NP];
    float4 S = splat(0.f);
    if (wave == 0) {
      // ---- the FM part: the arithmetic of k_fm_forward ----
      // branch-free, all row loads together (see forward_sample / k_fm_online): with the loads of a pass under
      // `if (live[p])` a sample's rows went out in 2 NP dependent round trips
      {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          li[p] = live[p] ? l_n[p] : 0u;
          x[p] = (has_x && live[p]) ? x_n[p] : 1.f;
          ok[p] = live[p] && li[p] < vocab[p];
        }
        if (lane == 0) y_lds = y_n;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          row[p] = load_row_sc1<LAYOUT>(a.rows + (size_t)(ok[p] ? lo[p] + li[p] : 0) * a.stride, q, kp, a.zoff);
          bad = bad || (live[p] && !ok[p]);
        }
        fetch_inputs(i + 1);
      }
      float4 s = splat(0.f), ss = splat(0.f);
      float fo = 0.f;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        if (ok[p]) {
          const float4 e = x[p] * row[p].v;
          s = s + e;
          ss = ss + e * e;
          fo += row[p].fo.x * x[p];
        }
      }
#define FMX_BFLY(M)                               \
  if (LPR <= M) {                                 \
    s = s + xor_lane_f4<M>(s, lane);              \
    ss = ss + xor_lane_f4<M>(ss, lane);           \
    fo += xor_lane_f<M>(fo, lane);                \
  }
      FMX_BFLY(1) FMX_BFLY(2) FMX_BFLY(4) FMX_BFLY(8) FMX_BFLY(16) FMX_BFLY(32)
#undef FMX_BFLY
      S = s;
      const float4 bi = 0.5f * (s * s - ss);
      float sbi = (bi.x + bi.y) + (bi.z + bi.w);
#pragma unroll
      for (int m = 1; m < LPR; m <<= 1) sbi += __shfl_xor(sbi, m);
      fo = __shfl(fo, 0);
      const float bias_w = LAYOUT == FMX_LAYOUT_WEIGHTS ? b0 : ftrl_w(b0, b1, a.h);
      if (lane < LPR) {
        bi_lds[4 * q] = bi.x;
        bi_lds[4 * q + 1] = bi.y;
        bi_lds[4 * q + 2] = bi.z;
        bi_lds[4 * q + 3] = bi.w;
      }
      if (lane == 0) base_lds = a.fm_term ? fo + sbi + bias_w : fo + bias_w;
    }
    __syncthreads();
    // ---- the MLP step of k_mlp_small on LDS-resident parameters ----
    MlpArgs m{};
    m.params = p_lds;
    m.bi = bi_lds;
    m.base = &base_lds;
    m.y = &y_lds;
    m.pred_out = a.pred + i;
    m.h = a.h;
    m.inv_b = 1.0f;
    m.B = 1;
    m.k = a.k;
    m.kp = kp;
    m.hidden = a.hidden;
    m.n_layers = a.n_layers;
    if (a.hedge) {
      m.alpha = alpha_lds;
      m.hedge_b = a.hedge_b;
      m.hedge_s = a.hedge_s;
      m.mode = MLP_MODE_HEDGE;
    } else {
      m.dz_out = &dz_lds;
      m.gbi_out = gbi_lds;
      m.mode = MLP_MODE_FIT;
      m.rule = a.rule;
      m.loss_kind = a.loss_kind;
    }
    mlp_small_body(m);
    __syncthreads();
    if (wave == 0 && !a.hedge) {
      // ---- the table update of k_fm_update at B = 1: every row is a run of one occurrence, G = dz [+ dL/dbi] ----
      const float dz = dz_lds;
      const float4 g4 = {gbi_lds[4 * q], gbi_lds[4 * q + 1], gbi_lds[4 * q + 2], gbi_lds[4 * q + 3]};
      const float4 G = splat(a.fm_term ? dz : 0.f) + g4;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        if (ok[p]) {
          const float4 xG = x[p] * G;
          update_row<LAYOUT, RULE>(a.rows + (size_t)(lo[p] + li[p]) * a.stride, q, kp, a.zoff, row[p], xG * S, x[p] * xG, x[p] * dz,
                                   a.h);
        }
      }
      if (LAYOUT == FMX_LAYOUT_WEIGHTS) {
        b0 = apply_rule<RULE>(b0, dz, a.h);
      } else {
        const float w = ftrl_w(b0, b1, a.h);
        ftrl_upd(b0, b1, w, dz, a.h);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the row stores are acknowledged before the next sample's loads
    }
    __syncthreads();
  }
  for (int i = tid; i < a.n_params; i += blockDim.x) a.params[i] = p_lds[i];
  if (a.hedge && tid < a.n_layers) a.alpha[tid] = alpha_lds[tid];
  if (wave == 0) {
    const bool any_bad = __ballot(bad) != 0ull;
    if (lane == 0) {
      if (!a.hedge) {
        a.bias[0] = b0;
        if (LAYOUT == FMX_LAYOUT_FTRL) a.bias[1] = b1;
      }
      if (any_bad && a.error) *a.error = 1;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// k_stream_read: HBM-read ceiling probe
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stream_read(const float4 *buf, int64_t n16, float *sink) {
  float acc = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
    const float4 v = buf[i];
    acc += (v.x + v.y) + (v.z + v.w);
  }
  if (acc == 123456.789f) *sink = acc;  // keeps the loads live; practically never true
}

// k_gather_read: random-row read ceiling probe -- n rows of 64 or 128 bytes at pseudo-random (hashed) positions of a buffer,
// 16 bytes per lane, LPR lanes per row: the access pattern of the forward gather with nothing else around it
template <int LPR>
__global__ __launch_bounds__(256) void k_gather_read(const float4 *buf, uint64_t n_rows, int64_t n, uint32_t seed, float *sink) {
  const int64_t g = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LPR;
  const int q = threadIdx.x % LPR;
  if (g >= n) return;
  uint64_t h = ((uint64_t)g + seed) * 0x9E3779B97F4A7C15ull;
  h ^= h >> 29;
  h *= 0xBF58476D1CE4E5B9ull;
  h ^= h >> 32;
  const float4 v = buf[(h % n_rows) * LPR + q];
  if ((v.x + v.y) + (v.z + v.w) == 123456.789f) *sink = v.x;  // keeps the load live; practically never true
}

// ------------------------------------------------------------------------------------------------------------
// host-side validation and dispatch
// ------------------------------------------------------------------------------------------------------------

int lpr_of(int kp) {
  switch (kp) {
    case 4: return 1;
    case 8: return 2;
    case 16: return 4;
    case 32: return 8;
    case 64: return 16;
    default: return 0;
  }
}

int check_table(const fmx_table_t *t) {
  if (!t) return fail(FMX_ERR_ARG, "table is null");
  if (!t->rows || !t->field_offsets || !t->bias) return fail(FMX_ERR_ARG, "table has a null pointer");
  if (t->n_fields < 1 || t->n_rows < 1 || t->k < 1) return fail(FMX_ERR_ARG, "table sizes must be positive");
  if (!lpr_of(t->kp) || t->k > t->kp) return fail(FMX_ERR_SHAPE, "kp=%d must be 4/8/16/32/64 and >= k=%d", t->kp, t->k);
  if (t->layout != FMX_LAYOUT_WEIGHTS && t->layout != FMX_LAYOUT_FTRL) return fail(FMX_ERR_ARG, "unknown layout %d", t->layout);
  int need = t->kp + 4;
  if (t->layout == FMX_LAYOUT_FTRL) {
    if (t->z_offset % 4 || t->z_offset < t->kp + 4)
      return fail(FMX_ERR_SHAPE, "z_offset=%d must be a multiple of 4 and >= kp + 4 = %d", t->z_offset, t->kp + 4);
    need = t->z_offset + 2 * t->kp;
  }
  if (t->row_stride % 4 || t->row_stride < need)
    return fail(FMX_ERR_SHAPE, "row_stride=%d must be a multiple of 4 and >= %d", t->row_stride, need);
  if (!aligned16(t->rows)) return fail(FMX_ERR_ALIGN, "table rows must be 16-byte aligned");
  if (t->max_field_rows < 1 || t->max_field_rows > t->n_rows) return fail(FMX_ERR_SHAPE, "max_field_rows out of range");
  if (t->n_sort_fields < 0 || (t->n_sort_fields > 0 && (t->n_sort_fields < t->n_fields || !t->sort_offsets || !t->sort_cols ||
                                                        t->max_sort_field_rows < 1 || t->max_sort_field_rows > t->max_field_rows)))
    return fail(FMX_ERR_SHAPE, "sort fields: n_sort_fields >= n_fields with sort_offsets, sort_cols and max_sort_field_rows, or 0");
  if (t->n_cols < 0 || (t->field_cols && t->n_cols < 1)) return fail(FMX_ERR_SHAPE, "field_cols needs n_cols >= 1");
  return FMX_OK;
}

int check_rule(const fmx_table_t *t, int rule) {
  if (rule == FMX_RULE_FTRL) {
    if (t->layout != FMX_LAYOUT_FTRL) return fail(FMX_ERR_ARG, "FMX_RULE_FTRL needs FMX_LAYOUT_FTRL");
  } else if (rule == FMX_RULE_SIGNADAM || rule == FMX_RULE_SGD) {
    if (t->layout != FMX_LAYOUT_WEIGHTS) return fail(FMX_ERR_ARG, "rule %d needs FMX_LAYOUT_WEIGHTS", rule);
  } else {
    return fail(FMX_ERR_ARG, "unknown rule %d", rule);
  }
  return FMX_OK;
}

// the SORT fields of a table: its fields, or the finer partition fmx_table_t.sort_offsets describes
inline bool mapped(const fmx_table_t *t) { return t->field_cols || t->field_base; }  // fields are pieces of index columns
inline int n_cols(const fmx_table_t *t) { return t->n_cols > 0 ? t->n_cols : t->n_fields; }
inline int n_sort_fields(const fmx_table_t *t) { return t->n_sort_fields > 0 ? t->n_sort_fields : t->n_fields; }
inline const int64_t *sort_offsets(const fmx_table_t *t) { return t->n_sort_fields > 0 ? t->sort_offsets : t->field_offsets; }
inline const int32_t *sort_cols(const fmx_table_t *t) { return t->n_sort_fields > 0 ? t->sort_cols : nullptr; }
inline int64_t max_sort_rows(const fmx_table_t *t) { return t->n_sort_fields > 0 ? t->max_sort_field_rows : t->max_field_rows; }

int check_sort_geometry(const fmx_table_t *t, int B) {
  if (B < 1) return fail(FMX_ERR_ARG, "B must be >= 1");
  const int Bp = fmx_sorted_width(B);
  if (Bp > MAX_SORT_WIDTH)
    return fail(FMX_ERR_UNSUPPORTED, "batch %d exceeds the LDS sort width %d", B, MAX_SORT_WIDTH);
  const int bbits = fmx_sorted_bbits(B);
  if ((uint64_t)(max_sort_rows(t) - 1) >= (uint64_t)(SENT >> bbits))
    return fail(FMX_ERR_UNSUPPORTED, "largest sort field (%lld rows) and batch %d do not fit a 32-bit (index, sample) composite: "
                "split the large fields (fmx_table_t.sort_offsets)", (long long)max_sort_rows(t), B);
  return FMX_OK;
}

}  // namespace
namespace fmxd {
Tune &tune() {
  static Tune t = [] {
    Tune x;
    if (const char *e = getenv("FMX_WPB_FWD")) x.wpb_fwd = atoi(e);
    if (const char *e = getenv("FMX_WPB_UPD")) x.wpb_upd = atoi(e);
    if (const char *e = getenv("FMX_SORT_E")) x.sort_e = atoi(e);
    if (const char *e = getenv("FMX_SORT_AHEAD")) x.sort_ahead = atoi(e);
    if (const char *e = getenv("FMX_ONLINE_PERSISTENT")) x.online_persistent = atoi(e);
    if (const char *e = getenv("FMX_INLINE_FIXUP")) x.inline_fixup = atoi(e);
    if (const char *e = getenv("FMX_SORT_CHUNKED")) x.sort_chunked = atoi(e);
    if (const char *e = getenv("FMX_MLP_CHAIN")) x.mlp_chain = atoi(e);
    auto ok = [](int v) { return v == 1 || v == 2 || v == 4; };
    if (!ok(x.wpb_fwd)) x.wpb_fwd = 2;
    if (!ok(x.wpb_upd)) x.wpb_upd = 2;
    return x;
  }();
  return t;
}
}  // namespace fmxd
namespace {

constexpr int SORT_AHEAD_MAX = 16;  // batches sorted per side-stream launch in fmx_fm_stream (r3: 16, was 8 -- every group boundary puts a
                                    // cross-stream wait of 5 - 6 us on the step's stream: 21.33 - 21.39 against 21.53 - 21.79 us per step)

// ---- workspace carving: [ sorted u32 F*Bp (x 2*SORT_AHEAD_MAX: the online loop sorts a group of batches ahead) |
//                          meta i32 F*tiles*2 | counter | parts f32 F*tiles*2*REC ], each 256-byte aligned ----
struct Workspace {
  uint32_t *sorted;       // buffer 0 of a ring of 2 * SORT_AHEAD_MAX buffers, `sorted_stride` elements apart
  size_t sorted_stride;
  uint32_t *runs;         // SORT_AHEAD_MAX buffers of the same shape: the chunk-sorted intermediate of k_sort_chunk / k_sort_merge
  int32_t *meta;
  int32_t *counter;  // step counter (one int32 in its own 256-byte slot; unused by the current loop)
  float *parts;
  size_t bytes;
};



Workspace carve(const fmx_table_t *t, int B, void *base) {
  const size_t F = (size_t)n_sort_fields(t), Bp = (size_t)fmx_sorted_width(B), tiles = Bp >> 6;
  const size_t rec = 2 * (size_t)t->kp + 4;
  const size_t o_sorted1 = align_up(F * Bp * 4, 256);
  const size_t o_runs = 2 * SORT_AHEAD_MAX * o_sorted1;
  const size_t o_meta = o_runs + (Bp >= 2 * SORT_CHUNK ? SORT_AHEAD_MAX * o_sorted1 : 0);
  const size_t o_counter = o_meta + align_up(F * tiles * 2 * 4, 256);
  const size_t o_parts = o_counter + 256;
  Workspace w;
  char *p = static_cast<char *>(base);
  w.sorted = reinterpret_cast<uint32_t *>(p);
  w.sorted_stride = o_sorted1 / 4;
  w.runs = reinterpret_cast<uint32_t *>(p + o_runs);
  w.meta = reinterpret_cast<int32_t *>(p + o_meta);
  w.counter = reinterpret_cast<int32_t *>(p + o_counter);
  w.parts = reinterpret_cast<float *>(p + o_parts);
  w.bytes = o_parts + align_up(F * tiles * 2 * rec * 4, 256);
  return w;
}

// the caller's workspace against what a step of B samples on this table needs NOW (the table's sort fields may have been
// split since the buffer was sized: fmx_workspace_bytes grows with them)
int check_workspace(const fmx_table_t *t, int B, const void *workspace, int64_t workspace_bytes, const char *who) {
  if (!workspace) return fail(FMX_ERR_ARG, "%s: null workspace", who);
  if (!aligned16(workspace)) return fail(FMX_ERR_ALIGN, "%s: workspace must be 16-byte aligned", who);
  const int64_t need = (int64_t)carve(t, B, nullptr).bytes;
  if (workspace_bytes < need)
    return fail(FMX_ERR_SHAPE, "%s: workspace of %lld bytes, %lld needed for B = %d and %d sort fields (fmx_workspace_bytes)", who,
                (long long)workspace_bytes, (long long)need, B, n_sort_fields(t));
  return FMX_OK;
}

// ---- a library-owned side stream per device: the occurrence sort does not depend on the weights, so it runs beside
//      the forward pass (fmx_fm_step) or one batch ahead (fmx_fm_stream).  Created on first use, never destroyed. ----
struct Side {
  hipStream_t stream = nullptr;  // the sort runs here
  hipStream_t main = nullptr;    // stands in for the caller's stream when that is the legacy default stream, which
                                 // cannot be captured into a hipGraph
  hipEvent_t fork = nullptr, sorted[2] = {nullptr, nullptr}, consumed[2] = {nullptr, nullptr};
  hipEvent_t user_fork = nullptr, user_join = nullptr;
};

Side *side_for_current_device() {
  static std::mutex mu;
  static Side sides[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  Side &sd = sides[dev];
  if (!sd.stream) {
    // the sort is background work: lowest stream priority for it, highest for the stand-in main stream
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (hipStreamCreateWithPriority(&sd.stream, hipStreamNonBlocking, lo) != hipSuccess) return nullptr;
    if (hipStreamCreateWithPriority(&sd.main, hipStreamNonBlocking, hi) != hipSuccess) return nullptr;
    bool ok = hipEventCreateWithFlags(&sd.fork, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&sd.user_fork, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&sd.user_join, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < 2 && ok; ++i) {
      ok = hipEventCreateWithFlags(&sd.sorted[i], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&sd.consumed[i], hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) return nullptr;
  }
  return &sd;
}

constexpr int OVERLAP_MIN_BATCH = 512;  // below this the extra event traffic costs more than the sort

template <int LPR, int NPASS>
void launch_forward_np(const FwdArgs &a, int layout, hipStream_t st) {
  const int wpb = tune().wpb_fwd;
  const dim3 grid((a.B + wpb - 1) / wpb), block(64 * wpb);
  if (layout == FMX_LAYOUT_WEIGHTS) hipLaunchKernelGGL((k_fm_forward<LPR, FMX_LAYOUT_WEIGHTS, NPASS>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((k_fm_forward<LPR, FMX_LAYOUT_FTRL, NPASS>), grid, block, 0, st, a);
}

template <int LPR>
void launch_forward(const FwdArgs &a, int layout, hipStream_t st) {
  const int slots = WAVE / LPR;
  const int np = (a.F + slots - 1) / slots;
  if (a.fcols || a.fbase) {  // fields are pieces of index columns: the generic field loop (the additions and their order are the same)
    const int wpb = tune().wpb_fwd;
    const dim3 grid((a.B + wpb - 1) / wpb), block(64 * wpb);
    if (layout == FMX_LAYOUT_WEIGHTS) hipLaunchKernelGGL((k_fm_forward<LPR, FMX_LAYOUT_WEIGHTS, 0, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_fm_forward<LPR, FMX_LAYOUT_FTRL, 0, true>), grid, block, 0, st, a);
    return;
  }
  switch (np) {
    case 1: launch_forward_np<LPR, 1>(a, layout, st); break;
    case 2: launch_forward_np<LPR, 2>(a, layout, st); break;
    case 3: launch_forward_np<LPR, 3>(a, layout, st); break;
    case 4: launch_forward_np<LPR, 4>(a, layout, st); break;
    default: launch_forward_np<LPR, 0>(a, layout, st); break;
  }
}

// The in-launch hand-offs tag their flag words with a per-launch sequence number passed as a kernel argument; a captured
// launch would replay a frozen number, so captures take the paths without hand-offs.
bool is_capturing(hipStream_t st) {
  if (!st) return false;  // the legacy default stream cannot be captured
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
}

template <int LPR, bool HAS_GBI, bool INL>
void launch_update(const UpdArgs &a, int rule, hipStream_t st) {
  const int tiles = a.F * (a.Bp >> 6);
  const int wpb = tune().wpb_upd;
  const dim3 grid((tiles + wpb - 1) / wpb + red_slices(a.B)), block(64 * wpb);
  switch (rule) {
    case FMX_RULE_SIGNADAM:
      hipLaunchKernelGGL((k_fm_update<LPR, FMX_LAYOUT_WEIGHTS, FMX_RULE_SIGNADAM, HAS_GBI, INL>), grid, block, 0, st, a);
      break;
    case FMX_RULE_SGD:
      hipLaunchKernelGGL((k_fm_update<LPR, FMX_LAYOUT_WEIGHTS, FMX_RULE_SGD, HAS_GBI, INL>), grid, block, 0, st, a);
      break;
    default:
      hipLaunchKernelGGL((k_fm_update<LPR, FMX_LAYOUT_FTRL, FMX_RULE_FTRL, HAS_GBI, INL>), grid, block, 0, st, a);
      break;
  }
}

template <int LPR>
void launch_update_rider(const UpdArgs &a, int rule, const MlpReduceArgs &r, hipStream_t st) {  // HAS_GBI, in-launch hand-off
  const int tiles = a.F * (a.Bp >> 6);
  const int wpb = tune().wpb_upd;
  const int n_upd = (tiles + wpb - 1) / wpb + red_slices(a.B), per = mlp_reduce_blocks_per_layer(r, 64 * wpb);
  const dim3 grid(n_upd + per * r.n_layers), block(64 * wpb);
  switch (rule) {
    case FMX_RULE_SIGNADAM:
      hipLaunchKernelGGL((k_fm_update_rider<LPR, FMX_LAYOUT_WEIGHTS, FMX_RULE_SIGNADAM, true>), grid, block, 0, st, a, r, n_upd, per);
      break;
    case FMX_RULE_SGD:
      hipLaunchKernelGGL((k_fm_update_rider<LPR, FMX_LAYOUT_WEIGHTS, FMX_RULE_SGD, true>), grid, block, 0, st, a, r, n_upd, per);
      break;
    default:
      hipLaunchKernelGGL((k_fm_update_rider<LPR, FMX_LAYOUT_FTRL, FMX_RULE_FTRL, true>), grid, block, 0, st, a, r, n_upd, per);
      break;
  }
}

template <int LPR>
void launch_fixup(const UpdArgs &a, int rule, hipStream_t st) {
  const int tiles = a.F * (a.Bp >> 6);
  const int wpb = tune().wpb_upd;
  const dim3 grid((tiles + wpb - 1) / wpb + (red_slices(a.B) > 1 ? 1 : 0)), block(64 * wpb);
  switch (rule) {
    case FMX_RULE_SIGNADAM:
      hipLaunchKernelGGL((k_fm_fixup<LPR, FMX_LAYOUT_WEIGHTS, FMX_RULE_SIGNADAM>), grid, block, 0, st, a);
      break;
    case FMX_RULE_SGD: hipLaunchKernelGGL((k_fm_fixup<LPR, FMX_LAYOUT_WEIGHTS, FMX_RULE_SGD>), grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL((k_fm_fixup<LPR, FMX_LAYOUT_FTRL, FMX_RULE_FTRL>), grid, block, 0, st, a); break;
  }
}

template <int LPR>
void launch_update_pair(const UpdArgs &a, int rule, bool has_gbi, hipStream_t st) {
  if (tune().inline_fixup && !is_capturing(st)) {  // one launch: the closing tile of a crossing run sums the records itself
    if (has_gbi) launch_update<LPR, true, true>(a, rule, st);
    else launch_update<LPR, false, true>(a, rule, st);
    return;
  }
  if (has_gbi) launch_update<LPR, true, false>(a, rule, st);
  else launch_update<LPR, false, false>(a, rule, st);
  launch_fixup<LPR>(a, rule, st);
}

template <int E>
void launch_sort(SortArgs a, hipStream_t st) {
  const int threads = a.Bp / E;
  uint32_t lds = (uint32_t)(a.Bp * sizeof(uint32_t));
  // fields of at most 511 rows take one counting pass inside the same launch (fmx_sort.inc, radix_field): on the Criteo list 24
  // of the 39 fields -- the launch's length is still the network's (the 15 larger fields), its chip time is not
  if (a.Bp <= RADIX_SMALL_WIDTH && threads >= 64) {
    a.small_bits = RADIX_SMALL_BITS;
    const uint32_t need = (uint32_t)radix_small_lds_bytes(a.Bp, threads);
    if (need > lds) lds = need;
  }
  const int grid = a.n_batches >= 8 ? 8 * a.F * ((a.n_batches + 7) / 8) : a.F * a.n_batches;
  hipLaunchKernelGGL((k_sort_occ<E>), dim3(grid), dim3(threads), lds, st, a);
}

int prepare_sort(int B) {
  if ((size_t)fmx_sorted_width(B) * 4 <= 64 * 1024) return FMX_OK;
  static std::mutex mu;
  static bool raised = false;
  std::lock_guard<std::mutex> lock(mu);
  if (!raised) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_sort_occ<8>), hipFuncAttributeMaxDynamicSharedMemorySize, MAX_SORT_WIDTH * 4);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_sort_occ<16>), hipFuncAttributeMaxDynamicSharedMemorySize, MAX_SORT_WIDTH * 4);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_sort_occ<32>), hipFuncAttributeMaxDynamicSharedMemorySize, MAX_SORT_WIDTH * 4);
    if (e != hipSuccess) return fail(FMX_ERR_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    raised = true;
  }
  return FMX_OK;
}

struct SortBatch {  // several batches of a pool in one launch
  int n_pool = 1, first = 0, n_batches = 1;
  int64_t pool_stride = 0, sorted_stride = 0;
};

int prepare_merge(int Bp) {
  if ((size_t)(Bp + Bp / 32) * 4 <= 64 * 1024) return FMX_OK;
  static std::mutex mu;
  static bool raised = false;
  std::lock_guard<std::mutex> lock(mu);
  if (!raised) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_sort_merge), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (MAX_SORT_WIDTH + MAX_SORT_WIDTH / 32) * 4);
    if (e != hipSuccess) return fail(FMX_ERR_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    raised = true;
  }
  return FMX_OK;
}

// `runs`: the workspace's chunk-sort intermediate (Workspace::runs; used when Bp >= 2 * SORT_CHUNK)
int sort_impl(const fmx_table_t *table, const int32_t *idx, int32_t B, uint32_t *sorted, uint32_t *runs, int32_t *error,
              hipStream_t st, const SortBatch *mb = nullptr) {
  SortArgs a;
  a.n_pool = mb ? mb->n_pool : 1;
  a.pool_first = mb ? mb->first : 0;
  a.n_batches = mb ? mb->n_batches : 1;
  a.pool_stride = mb ? mb->pool_stride : 0;
  a.sorted_stride = mb ? mb->sorted_stride : 0;
  a.idx = idx;
  a.foff = table->field_offsets;
  a.soff = sort_offsets(table);
  a.cols = sort_cols(table);
  a.sorted = sorted;
  a.error = error;
  a.B = B;
  a.F = n_sort_fields(table);
  a.Fi = n_cols(table);
  a.small_bits = 0;
  a.fcols = table->field_cols;
  a.fbase = table->field_base;
  a.Bp = fmx_sorted_width(B);
  a.bbits = fmx_sorted_bbits(B);
  // the chunked form wins on LATENCY (one or two batches per launch: the prefetched global sorts of the multi-GPU modes, a
  // single step); a launch of many batches fills the chip either way and the rank merge then costs more work than the
  // bitonic stages it replaces (8 batches of 16,384: 294 vs 118 us)
  if (a.Bp >= (tune().sort_chunked > 1 ? 2 * SORT_CHUNK : SORT_CHUNKED_MIN_WIDTH) && tune().sort_chunked && runs &&
      (a.n_batches <= 2 || tune().sort_chunked > 1)) {
    // chunk sort + rank merge, spread over the chip (k_sort_chunk / k_sort_merge)
    if (int rc = prepare_merge(a.Bp)) return rc;
    ChunkArgs c;
    c.s = a;
    c.runs = runs;
    c.runs_stride = (int64_t)align_up((size_t)a.F * a.Bp * 4, 256) / 4;
    c.C = a.Bp / SORT_CHUNK;
    const int units1 = a.n_batches * c.C, grid1 = 8 * a.F * ((units1 + 7) / 8);
    hipLaunchKernelGGL(k_sort_chunk, dim3(grid1), dim3(SORT_CHUNK_THREADS), 0, st, c);
    const int units2 = a.n_batches * a.F, grid2 = 8 * c.C * ((units2 + 7) / 8);
    hipLaunchKernelGGL(k_sort_merge, dim3(grid2), dim3(SORT_CHUNK_THREADS), (uint32_t)((a.Bp + a.Bp / 32) * sizeof(uint32_t)), st, c);
    return check_launch("k_sort_chunk / k_sort_merge");
  }
  if (int rc = prepare_sort(B)) return rc;
  // E elements per thread, Bp / E threads (a multiple of 64, at most 1024)
  int E = a.Bp <= 64 ? 1 : a.Bp <= 128 ? 2 : a.Bp <= 4096 ? 4 : a.Bp <= 8192 ? 8 : a.Bp <= 16384 ? 16 : 32;
  const int want = tune().sort_e;
  if (want > E && want <= 32 && (want & (want - 1)) == 0 && a.Bp / want >= 64) E = want;
  switch (E) {
    case 1: launch_sort<1>(a, st); break;
    case 2: launch_sort<2>(a, st); break;
    case 4: launch_sort<4>(a, st); break;
    case 8: launch_sort<8>(a, st); break;
    case 16: launch_sort<16>(a, st); break;
    default: launch_sort<32>(a, st); break;
  }
  return check_launch("k_sort_occ");
}

FwdArgs fill_fwd(const fmx_table_t *table, const fmx_hyper_t *hyper, const int32_t *idx, const float *xv, const float *y,
                 int32_t B, int32_t loss_kind, float inv_b, const fmx_fwd_out_t *out) {
  FwdArgs a;
  a.ldS = out->sample_ld > 0 ? out->sample_ld : table->kp;
  a.ld1 = out->sample_ld > 0 ? out->sample_ld : 1;
  a.rows = table->rows;
  a.foff = table->field_offsets;
  a.bias = table->bias;
  a.idx = idx;
  a.xv = xv;
  a.y = y;
  a.out = *out;
  a.h = *hyper;
  a.B = B;
  a.F = table->n_fields;
  a.Fc = n_cols(table);
  a.fcols = table->field_cols;
  a.fbase = table->field_base;
  a.kp = table->kp;
  a.stride = table->row_stride;
  a.zoff = table->z_offset;
  a.loss_kind = loss_kind;
  a.h.alpha = 1.0f / hyper->alpha;  // the kernels multiply by 1/alpha
  a.inv_b = inv_b;
  return a;
}

int forward_impl(const fmx_table_t *table, const fmx_hyper_t *hyper, const int32_t *idx, const float *xv, const float *y,
                 int32_t B, int32_t loss_kind, float inv_b, const fmx_fwd_out_t *out, hipStream_t st) {
  const FwdArgs a = fill_fwd(table, hyper, idx, xv, y, B, loss_kind, inv_b, out);
  switch (lpr_of(table->kp)) {
    case 1: launch_forward<1>(a, table->layout, st); break;
    case 2: launch_forward<2>(a, table->layout, st); break;
    case 4: launch_forward<4>(a, table->layout, st); break;
    case 8: launch_forward<8>(a, table->layout, st); break;
    default: launch_forward<16>(a, table->layout, st); break;
  }
  return check_launch("k_fm_forward");
}

UpdArgs fill_upd(const fmx_table_t *table, const fmx_hyper_t *hyper, const Workspace &w, const uint32_t *sorted,
                 const float *xv, const float *S, const float *dz_first, const float *dz_bi, const float *gbi, int32_t B,
                 const float *loss_b, float inv_b, float *loss_out, int32_t *step_counter, int32_t sample_ld,
                 int32_t *err_flag) {
  UpdArgs a;
  a.ldS = sample_ld > 0 ? sample_ld : table->kp;
  a.ld1 = sample_ld > 0 ? sample_ld : 1;
  a.ldG = sample_ld > 0 ? sample_ld : table->kp;
  static std::atomic<uint32_t> launch_seq{1};
  a.seq = launch_seq.fetch_add(1) & 0x0FFFFFFFu;
  if (a.seq == 0) a.seq = launch_seq.fetch_add(1) & 0x0FFFFFFFu;  // 0 is what a zeroed workspace holds
  a.error = err_flag;
  a.rows = table->rows;
  a.foff = sort_offsets(table);  // the update walks the SORT fields' lists; a sort field's rows start at its own offset
  a.cols = sort_cols(table);
  a.Fx = n_cols(table);
  a.fcols = table->field_cols;
  a.bias = table->bias;
  a.sorted = sorted;
  a.parts = w.parts;
  a.meta = w.meta;
  a.red = reinterpret_cast<float *>(w.counter);
  a.xv = xv;
  a.S = S;
  a.dz_first = dz_first;
  a.dz_bi = dz_bi;
  a.gbi = gbi;
  a.loss_b = loss_b;
  a.loss_out = loss_out;
  a.step_counter = step_counter;
  a.h = *hyper;
  a.B = B;
  a.F = n_sort_fields(table);
  a.Bp = fmx_sorted_width(B);
  a.bbits = fmx_sorted_bbits(B);
  a.kp = table->kp;
  a.stride = table->row_stride;
  a.zoff = table->z_offset;
  a.h.alpha = 1.0f / hyper->alpha;  // the kernels multiply by 1/alpha
  a.inv_b = inv_b;
  return a;
}

int update_impl(const fmx_table_t *table, const fmx_hyper_t *hyper, int32_t rule, const Workspace &w,
                const uint32_t *sorted, const float *xv,
                const float *S, const float *dz_first, const float *dz_bi, const float *gbi, int32_t B,
                const float *loss_b, float inv_b, float *loss_out, hipStream_t st,
                int32_t *step_counter = nullptr, int32_t sample_ld = 0,
                int32_t *err_flag = nullptr, const MlpReduceArgs *rider = nullptr) {
  const UpdArgs a = fill_upd(table, hyper, w, sorted, xv, S, dz_first, dz_bi, gbi, B, loss_b, inv_b, loss_out, step_counter,
                             sample_ld, err_flag);
  if (rider) {
    if (gbi != nullptr && tune().inline_fixup && !is_capturing(st)) {  // the one-launch form of the update: the rider goes with it
      switch (lpr_of(table->kp)) {
        case 1: launch_update_rider<1>(a, rule, *rider, st); break;
        case 2: launch_update_rider<2>(a, rule, *rider, st); break;
        case 4: launch_update_rider<4>(a, rule, *rider, st); break;
        case 8: launch_update_rider<8>(a, rule, *rider, st); break;
        default: launch_update_rider<16>(a, rule, *rider, st); break;
      }
      return check_launch("k_fm_update_rider");
    }
    mlp_launch_reduce(*rider, st);  // otherwise the reduction as a launch of its own, in front
  }
  switch (lpr_of(table->kp)) {
    case 1: launch_update_pair<1>(a, rule, gbi != nullptr, st); break;
    case 2: launch_update_pair<2>(a, rule, gbi != nullptr, st); break;
    case 4: launch_update_pair<4>(a, rule, gbi != nullptr, st); break;
    case 8: launch_update_pair<8>(a, rule, gbi != nullptr, st); break;
    default: launch_update_pair<16>(a, rule, gbi != nullptr, st); break;
  }
  return check_launch("k_fm_update / k_fm_fixup");
}

template <int LPR, int LAYOUT, int RULE>
void launch_online_np(const OnlineArgs &a, int np, hipStream_t st) {
  switch (np) {
    case 1: hipLaunchKernelGGL((k_fm_online<LPR, LAYOUT, RULE, 1>), dim3(1), dim3(64), 0, st, a); break;
    case 2: hipLaunchKernelGGL((k_fm_online<LPR, LAYOUT, RULE, 2>), dim3(1), dim3(64), 0, st, a); break;
    case 3: hipLaunchKernelGGL((k_fm_online<LPR, LAYOUT, RULE, 3>), dim3(1), dim3(64), 0, st, a); break;
    default: hipLaunchKernelGGL((k_fm_online<LPR, LAYOUT, RULE, 4>), dim3(1), dim3(64), 0, st, a); break;
  }
}

template <int LPR>
void launch_online(const OnlineArgs &a, int rule, int np, hipStream_t st) {
  switch (rule) {
    case FMX_RULE_SIGNADAM: launch_online_np<LPR, FMX_LAYOUT_WEIGHTS, FMX_RULE_SIGNADAM>(a, np, st); break;
    case FMX_RULE_SGD: launch_online_np<LPR, FMX_LAYOUT_WEIGHTS, FMX_RULE_SGD>(a, np, st); break;
    default: launch_online_np<LPR, FMX_LAYOUT_FTRL, FMX_RULE_FTRL>(a, np, st); break;
  }
}

constexpr int ONLINE_MLP_MAX_PARAMS = 8192;  // floats of MLP parameters kept in LDS by k_online_mlp

template <int LPR, int LAYOUT, int RULE>
void launch_online_mlp_k(const OnlineMlpArgs &a, hipStream_t st) {
  static bool raised = false;
  if (!raised) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_online_mlp<LPR, LAYOUT, RULE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, ONLINE_MLP_MAX_PARAMS * 4);
    raised = true;
  }
  hipLaunchKernelGGL((k_online_mlp<LPR, LAYOUT, RULE>), dim3(1), dim3(256), (size_t)a.n_params * 4, st, a);
}

template <int LPR>
void launch_online_mlp(const OnlineMlpArgs &a, int layout, int rule, hipStream_t st) {
  if (layout == FMX_LAYOUT_FTRL) launch_online_mlp_k<LPR, FMX_LAYOUT_FTRL, FMX_RULE_FTRL>(a, st);  // Hedge only: tables are read
  else if (rule == FMX_RULE_SGD) launch_online_mlp_k<LPR, FMX_LAYOUT_WEIGHTS, FMX_RULE_SGD>(a, st);
  else launch_online_mlp_k<LPR, FMX_LAYOUT_WEIGHTS, FMX_RULE_SIGNADAM>(a, st);
}

template <int LPR>
int launch_forward_part(const PartArgs &a, int np, int n_local_blocks, hipStream_t st) {
  const int per_wave = (WAVE / LPR) >> a.sl_log2, waves = (a.B + per_wave - 1) / per_wave;
  const dim3 grid((waves + 3) / 4, n_local_blocks), block(256);
  switch (np) {
    case 1: hipLaunchKernelGGL((k_fm_forward_part<LPR, 1>), grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL((k_fm_forward_part<LPR, 2>), grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL((k_fm_forward_part<LPR, 3>), grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL((k_fm_forward_part<LPR, 4>), grid, block, 0, st, a); break;
    default: return fail(FMX_ERR_UNSUPPORTED, "fmx_fm_forward_partial: more than 4 fields per lane group");
  }
  return check_launch("k_fm_forward_part");
}

template <int LPR, int LAYOUT>
int launch_forward_finish_g(const FinishArgs &a, int G, hipStream_t st) {
  const int waves = (a.B + (WAVE / LPR) - 1) / (WAVE / LPR);
  const dim3 grid((waves + 3) / 4), block(256);
  switch (G) {
    case 1: hipLaunchKernelGGL((k_fm_forward_finish<LPR, LAYOUT, 1>), grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL((k_fm_forward_finish<LPR, LAYOUT, 2>), grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL((k_fm_forward_finish<LPR, LAYOUT, 4>), grid, block, 0, st, a); break;
    case 8: hipLaunchKernelGGL((k_fm_forward_finish<LPR, LAYOUT, 8>), grid, block, 0, st, a); break;
    case 16: hipLaunchKernelGGL((k_fm_forward_finish<LPR, LAYOUT, 16>), grid, block, 0, st, a); break;
    default: return fail(FMX_ERR_ARG, "fmx_fm_forward_finish: n_owners must be 1, 2, 4, 8 or 16");
  }
  return check_launch("k_fm_forward_finish");
}

template <int LPR>
int launch_forward_finish(const FinishArgs &a, int layout, int G, hipStream_t st) {
  if (G > WAVE / LPR) return fail(FMX_ERR_ARG, "fmx_fm_forward_finish: more owners than lane groups");
  return layout == FMX_LAYOUT_WEIGHTS ? launch_forward_finish_g<LPR, FMX_LAYOUT_WEIGHTS>(a, G, st)
                                      : launch_forward_finish_g<LPR, FMX_LAYOUT_FTRL>(a, G, st);
}

// the tree cut into n_blocks blocks, the table holding n_local_blocks of them: fields [lb NP SL, (lb + 1) NP SL) are block lb's
int part_impl(const fmx_table_t *table, const int32_t *idx, const float *xv, int32_t B, int32_t n_blocks, int32_t n_local_blocks,
              int32_t group, float *parts_out, int32_t *error, hipStream_t st) {
  if (group <= 0) group = B;
  if (B % group) return fail(FMX_ERR_SHAPE, "fmx_fm_forward_partial: B = %d is not a multiple of group = %d", B, group);
  const int lpr = lpr_of(table->kp), slots = WAVE / lpr;
  if (n_blocks < 1 || n_blocks > slots || (n_blocks & (n_blocks - 1))) return fail(FMX_ERR_ARG, "n_blocks must be a power of two <= %d", slots);
  if (n_local_blocks < 1 || n_local_blocks > n_blocks) return fail(FMX_ERR_ARG, "n_local_blocks must be in [1, n_blocks]");
  const int sl = slots / n_blocks;
  if (n_local_blocks > 1 && table->n_fields % (n_local_blocks * sl))  // (one block: a ragged last pass is fine)
    return fail(FMX_ERR_SHAPE, "a table of %d blocks of %d lane groups holds a multiple of %d fields (empty fields fill the holes), not %d",
                n_local_blocks, sl, n_local_blocks * sl, table->n_fields);
  int sl_log2 = 0;
  while ((1 << sl_log2) < sl) ++sl_log2;
  PartArgs a;
  a.rows = table->rows;
  a.foff = table->field_offsets;
  a.fcols = table->field_cols;
  a.fbase = table->field_base;
  a.idx = idx;
  a.xv = xv;
  a.rec = parts_out;
  a.error = error;
  a.B = B;
  a.F = table->n_fields;
  a.Fc = n_cols(table);
  a.stride = table->row_stride;
  a.sl_log2 = sl_log2;
  a.group = group;
  const int np = (table->n_fields + n_local_blocks * sl - 1) / (n_local_blocks * sl);
  switch (lpr) {
    case 1: return launch_forward_part<1>(a, np, n_local_blocks, st);
    case 2: return launch_forward_part<2>(a, np, n_local_blocks, st);
    case 4: return launch_forward_part<4>(a, np, n_local_blocks, st);
    case 8: return launch_forward_part<8>(a, np, n_local_blocks, st);
    default: return launch_forward_part<16>(a, np, n_local_blocks, st);
  }
}

int check_forward_args(const fmx_table_t *table, const fmx_hyper_t *hyper, const int32_t *idx, const float *y, int32_t B,
                       int32_t loss_kind, const fmx_fwd_out_t *out) {
  if (int rc = check_table(table)) return rc;
  if (!hyper || !idx || !out) return fail(FMX_ERR_ARG, "fmx_fm_forward: null argument");
  if (B < 1) return fail(FMX_ERR_ARG, "B must be >= 1");
  if (loss_kind < FMX_LOSS_NONE || loss_kind > FMX_LOSS_BCE_SIGMOID) return fail(FMX_ERR_ARG, "unknown loss %d", loss_kind);
  if (loss_kind != FMX_LOSS_NONE && !y) return fail(FMX_ERR_ARG, "a loss needs labels y");
  if ((out->S && !aligned16(out->S)) || (out->bi && !aligned16(out->bi)))
    return fail(FMX_ERR_ALIGN, "S and bi must be 16-byte aligned");
  if (out->sample_ld != 0 && (out->sample_ld < table->kp || out->sample_ld % 4))
    return fail(FMX_ERR_SHAPE, "sample_ld=%d must be 0 or a multiple of 4 that is >= kp", out->sample_ld);
  return FMX_OK;
}

int check_step_args(const fmx_table_t *table, const fmx_hyper_t *hyper, int32_t rule, int32_t loss_kind, int32_t B,
                    const void *workspace, const fmx_fwd_out_t *fwd) {
  if (int rc = check_table(table)) return rc;
  if (int rc = check_rule(table, rule)) return rc;
  if (int rc = check_sort_geometry(table, B)) return rc;
  if (!hyper || !workspace) return fail(FMX_ERR_ARG, "null hyper / workspace");
  if (!aligned16(workspace)) return fail(FMX_ERR_ALIGN, "workspace must be 16-byte aligned");
  if (!fwd || !fwd->S || !fwd->dz || !fwd->loss) return fail(FMX_ERR_ARG, "fwd->S, fwd->loss, fwd->dz are required");
  if (!aligned16(fwd->S) || !aligned16(fwd->dz) || !aligned16(fwd->loss))
    return fail(FMX_ERR_ALIGN, "fwd->S, fwd->dz and fwd->loss must be 16-byte aligned");
  if (loss_kind != FMX_LOSS_BCE_LOGITS && loss_kind != FMX_LOSS_BCE_SIGMOID) return fail(FMX_ERR_ARG, "a step needs a loss");
  return FMX_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------
extern "C" {

int fmx_version(void) { return FMX_VERSION; }

int fmx_set_option(const char *name, int value) {
  if (!name) return fail(FMX_ERR_ARG, "fmx_set_option: null name");
  Tune &t = tune();
  int *slot = nullptr;
  if (!strcmp(name, "inline_fixup")) slot = &t.inline_fixup;
  else if (!strcmp(name, "sort_ahead")) slot = &t.sort_ahead;
  else if (!strcmp(name, "online_persistent")) slot = &t.online_persistent;
  else if (!strcmp(name, "sort_chunked")) slot = &t.sort_chunked;
  else if (!strcmp(name, "mlp_chain")) slot = &t.mlp_chain;
  else return fail(FMX_ERR_ARG, "fmx_set_option: unknown option '%s'", name);
  const int old = *slot;
  *slot = value;
  return old;
}

const char *fmx_last_error_string(void) { return g_err; }

int fmx_sorted_width(int B) {
  int w = 64;
  while (w < B && w < (1 << 30)) w <<= 1;
  return w;
}

int fmx_sorted_bbits(int B) {
  int w = fmx_sorted_width(B), bits = 0;
  while ((1 << bits) < w) ++bits;
  return bits;
}

int64_t fmx_workspace_bytes(const fmx_table_t *table, int32_t B) {
  if (check_table(table) != FMX_OK) return FMX_ERR_ARG;
  if (B < 1) return fail(FMX_ERR_ARG, "B must be >= 1");
  return (int64_t)carve(table, B, nullptr).bytes;
}

int fmx_fm_forward(const fmx_table_t *table, const fmx_hyper_t *hyper, const int32_t *idx, const float *xv,
                   const float *y, int32_t B, int32_t loss_kind, float inv_b, const fmx_fwd_out_t *out,
                   fmx_stream_t stream) {
  if (int rc = check_forward_args(table, hyper, idx, y, B, loss_kind, out)) return rc;
  return forward_impl(table, hyper, idx, xv, y, B, loss_kind, inv_b, out, static_cast<hipStream_t>(stream));
}

int fmx_fm_forward_partial(const fmx_table_t *table, const int32_t *idx, const float *xv, int32_t B, int32_t n_blocks,
                           int32_t n_local_blocks, int32_t group, float *parts_out, int32_t *error, fmx_stream_t stream) {
  if (int rc = check_table(table)) return rc;
  if (!idx || !parts_out) return fail(FMX_ERR_ARG, "fmx_fm_forward_partial: null argument");
  if (B < 1) return fail(FMX_ERR_ARG, "B must be >= 1");
  if (!aligned16(parts_out)) return fail(FMX_ERR_ALIGN, "parts_out must be 16-byte aligned");
  return part_impl(table, idx, xv, B, n_blocks, n_local_blocks, group, parts_out, error, static_cast<hipStream_t>(stream));
}

int fmx_fm_forward_finish(const fmx_hyper_t *hyper, const float *bias, int32_t layout, int32_t kp, const float *parts,
                          int64_t owner_stride, int32_t n_owners, const float *y, int32_t B, int32_t loss_kind, float inv_b,
                          const fmx_fwd_out_t *out, fmx_stream_t stream) {
  if (!hyper || !bias || !parts || !out) return fail(FMX_ERR_ARG, "fmx_fm_forward_finish: null argument");
  if (layout != FMX_LAYOUT_WEIGHTS && layout != FMX_LAYOUT_FTRL) return fail(FMX_ERR_ARG, "unknown layout %d", layout);
  if (!lpr_of(kp)) return fail(FMX_ERR_SHAPE, "kp=%d must be 4/8/16/32/64", kp);
  if (B < 1) return fail(FMX_ERR_ARG, "B must be >= 1");
  if (loss_kind < FMX_LOSS_NONE || loss_kind > FMX_LOSS_BCE_SIGMOID) return fail(FMX_ERR_ARG, "unknown loss %d", loss_kind);
  if (loss_kind != FMX_LOSS_NONE && !y) return fail(FMX_ERR_ARG, "a loss needs labels y");
  if (!aligned16(parts) || owner_stride % 4 || (out->S && !aligned16(out->S)) || (out->bi && !aligned16(out->bi)))
    return fail(FMX_ERR_ALIGN, "parts, S and bi must be 16-byte aligned, owner_stride a multiple of 4");
  if (out->sample_ld != 0 && (out->sample_ld < kp || out->sample_ld % 4))
    return fail(FMX_ERR_SHAPE, "sample_ld=%d must be 0 or a multiple of 4 that is >= kp", out->sample_ld);
  FinishArgs a;
  a.parts = parts;
  a.rank_stride = owner_stride;
  a.bias = bias;
  a.y = y;
  a.out = *out;
  a.h = *hyper;
  a.h.alpha = 1.0f / hyper->alpha;  // the kernels multiply by 1/alpha
  a.B = B;
  a.loss_kind = loss_kind;
  a.ldS = out->sample_ld > 0 ? out->sample_ld : kp;
  a.ld1 = out->sample_ld > 0 ? out->sample_ld : 1;
  a.inv_b = inv_b;
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (lpr_of(kp)) {
    case 1: return launch_forward_finish<1>(a, layout, n_owners, st);
    case 2: return launch_forward_finish<2>(a, layout, n_owners, st);
    case 4: return launch_forward_finish<4>(a, layout, n_owners, st);
    case 8: return launch_forward_finish<8>(a, layout, n_owners, st);
    default: return launch_forward_finish<16>(a, layout, n_owners, st);
  }
}

int fmx_sort_occurrences(const fmx_table_t *table, const int32_t *idx, int32_t B, void *workspace, int64_t workspace_bytes, int32_t *error,
                         fmx_stream_t stream) {
  if (int rc = check_table(table)) return rc;
  if (!idx || !workspace) return fail(FMX_ERR_ARG, "fmx_sort_occurrences: null argument");
  if (!aligned16(workspace)) return fail(FMX_ERR_ALIGN, "workspace must be 16-byte aligned");
  if (int rc = check_sort_geometry(table, B)) return rc;
  if (int rc = check_workspace(table, B, workspace, workspace_bytes, "fmx_sort_occurrences")) return rc;
  const Workspace w = carve(table, B, workspace);
  return sort_impl(table, idx, B, w.sorted, w.runs, error, static_cast<hipStream_t>(stream));
}

int fmx_fm_update(const fmx_table_t *table, const fmx_hyper_t *hyper, int32_t rule, void *workspace, int64_t workspace_bytes, const float *xv,
                  const float *S, const float *dz_first, const float *dz_bi, const float *gbi, int32_t B,
                  int32_t sample_ld, const float *loss_b, float inv_b, float *loss_out, fmx_stream_t stream) {
  if (int rc = check_table(table)) return rc;
  if (sample_ld != 0 && (sample_ld < table->kp || sample_ld % 4))
    return fail(FMX_ERR_SHAPE, "sample_ld=%d must be 0 or a multiple of 4 that is >= kp", sample_ld);
  if (int rc = check_rule(table, rule)) return rc;
  if (!hyper || !workspace || !S || !dz_first) return fail(FMX_ERR_ARG, "fmx_fm_update: null argument");
  if (!dz_bi && !gbi) return fail(FMX_ERR_ARG, "fmx_fm_update: one of dz_bi / gbi is required");
  if (int rc = check_sort_geometry(table, B)) return rc;
  if (!aligned16(workspace) || !aligned16(S) || (gbi && !aligned16(gbi)) ||
      (sample_ld == 0 && (!aligned16(dz_first) || (loss_b && !aligned16(loss_b)))))
    return fail(FMX_ERR_ALIGN, "workspace, S, gbi (and dense dz_first / loss_b) must be 16-byte aligned");
  if (int rc = check_workspace(table, B, workspace, workspace_bytes, "fmx_fm_update")) return rc;
  const Workspace w = carve(table, B, workspace);
  return update_impl(table, hyper, rule, w, w.sorted, xv, S, dz_first, dz_bi, gbi, B, loss_b, inv_b, loss_out,
                     static_cast<hipStream_t>(stream), nullptr, sample_ld);
}

// NFM's input logit: the first-order sum plus the bias (reference nfm_adam.py:78-88), one fp32 add per sample as the trainer does it
__global__ void k_first_plus_bias(float *out, const float *sfirst, const float *bias, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) out[b] = sfirst[b] + bias[0];
}

// The mini-batch DeepFM loop over a device-resident pool (BASELINE configs[3]): per step the forward of the tables, the MLP
// section on bi (fmx_mlp_section: k_mlp_chain, k_mlp_wgrad_stream, k_mlp_reduce with the SGD of the MLP applied in it) and the
// table update with dL/dbi, all issued from here; the occurrence sorts run in groups on the side stream as in fmx_fm_stream.
// Through the Python trainer the same step is bound by its host side (84 us of calls per step for 67 us of kernels).
int fmx_deepfm_stream(const fmx_table_t *table, const fmx_hyper_t *hyper, int32_t rule, const fmx_mlp_t *mlp, int32_t loss_kind, int32_t fm_term,
                      const int32_t *idx_pool, const float *y_pool, int32_t n_pool, int32_t B, float inv_b, int32_t n_steps,
                      void *workspace, int64_t workspace_bytes, void *mlp_workspace, const fmx_fwd_out_t *fwd, float *dz, float *gbi,
                      float *grads, float lr_mlp, float *loss_out, fmx_stream_t stream) {
  if (int rc = check_table(table)) return rc;
  if (int rc = check_rule(table, rule)) return rc;
  if (!hyper || !mlp || !workspace || !mlp_workspace || !fwd || !fwd->S || !fwd->bi || !fwd->logit || !dz || !gbi || !grads)
    return fail(FMX_ERR_ARG, "fmx_deepfm_stream: null argument (fwd needs S, bi and logit)");
  if (fwd->sample_ld != 0) return fail(FMX_ERR_ARG, "fmx_deepfm_stream: dense forward outputs only (sample_ld = 0)");
  if (!fm_term && (!fwd->sfirst || table->layout != FMX_LAYOUT_WEIGHTS))
    return fail(FMX_ERR_UNSUPPORTED, "fmx_deepfm_stream: fm_term = 0 (NFM) needs fwd->sfirst and a table in the weights layout");
  if (!idx_pool || !y_pool || n_pool < 1 || n_steps < 0 || B < 1) return fail(FMX_ERR_ARG, "fmx_deepfm_stream: bad pool / step count");
  if (mlp->k > table->kp) return fail(FMX_ERR_SHAPE, "fmx_deepfm_stream: the MLP reads k=%d columns of a bi of kp=%d", mlp->k, table->kp);
  if (!aligned16(gbi) || !aligned16(dz)) return fail(FMX_ERR_ALIGN, "dz and gbi must be 16-byte aligned");
  if (int rc = check_sort_geometry(table, B)) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (int rc = check_workspace(table, B, workspace, workspace_bytes, "fmx_deepfm_stream")) return rc;
  const Workspace w = carve(table, B, workspace);
  const size_t F = (size_t)n_cols(table);
  int rc = FMX_OK;
  Side *sd = (B >= OVERLAP_MIN_BATCH && n_steps > 0) ? side_for_current_device() : nullptr;
  hipStream_t user = st;
  const bool detour = sd && st == nullptr;  // off the legacy default stream (see fmx_fm_stream)
  if (detour) {
    (void)hipEventRecord(sd->user_fork, user);
    st = sd->main;
    (void)hipStreamWaitEvent(st, sd->user_fork, 0);
  }
  int ahead = tune().sort_ahead;
  if (ahead < 1) ahead = 1;
  if (ahead > SORT_AHEAD_MAX) ahead = SORT_AHEAD_MAX;
  auto group_size = [&](int g, int first_step) {
    int n = g == 0 ? 4 : ahead;
    if (n > ahead) n = ahead;
    if (n > n_steps - first_step) n = n_steps - first_step;
    return n;
  };
  auto sort_group = [&](int g, int first_step, int n, hipStream_t where) -> int {
    SortBatch mb;
    mb.n_pool = n_pool;
    mb.first = first_step % n_pool;
    mb.n_batches = n;
    mb.pool_stride = (int64_t)B * (int64_t)F;
    mb.sorted_stride = (int64_t)w.sorted_stride;
    return sort_impl(table, idx_pool, B, w.sorted + (size_t)(g & 1) * ahead * w.sorted_stride, w.runs, fwd->error, where, &mb);
  };
  if (sd && n_steps > 0) {
    (void)hipEventRecord(sd->fork, st);
    (void)hipStreamWaitEvent(sd->stream, sd->fork, 0);
    rc = sort_group(0, 0, group_size(0, 0), sd->stream);
    (void)hipEventRecord(sd->sorted[0], sd->stream);
  }
  int first_step = 0;
  for (int g = 0; first_step < n_steps && rc == FMX_OK; ++g) {
    const int n = group_size(g, first_step);
    const int next_first = first_step + n;
    if (!sd) rc = sort_group(g, first_step, n, st);
    for (int i = 0; i < n && rc == FMX_OK; ++i) {
      const int s = first_step + i, j = s % n_pool;
      const int32_t *idx = idx_pool + (size_t)j * B * F;
      const float *y = y_pool + (size_t)j * B;
      const uint32_t *sorted = w.sorted + ((size_t)(g & 1) * ahead + i) * w.sorted_stride;
      rc = forward_impl(table, hyper, idx, nullptr, nullptr, B, FMX_LOSS_NONE, inv_b, fwd, st);
      if (rc == FMX_OK && !fm_term) {  // NFM: the network's input logit is first-order + bias; the FM logit's buffer holds it
        hipLaunchKernelGGL(k_first_plus_bias, dim3((B + 255) / 256), dim3(256), 0, st, fwd->logit, fwd->sfirst, table->bias, B);
        rc = check_launch("fmx_deepfm_stream (k_first_plus_bias)");
      }
      MlpReduceArgs red;  // the section's last launch rides inside the table update's (k_fm_update_rider)
      if (rc == FMX_OK)
        rc = mlp_section_deferred_reduce(mlp, loss_kind, fwd->bi, table->kp, fwd->logit, y, B, inv_b, mlp_workspace, nullptr, dz, gbi, table->kp,
                                         grads, lr_mlp, loss_out ? loss_out + s : nullptr, st, &red);
      if (sd && i == 0) (void)hipStreamWaitEvent(st, sd->sorted[g & 1], 0);
      if (rc == FMX_OK)
        rc = update_impl(table, hyper, rule, w, sorted, nullptr, fwd->S, dz, fm_term ? dz : nullptr, gbi, B, nullptr, inv_b, nullptr, st, nullptr, 0,
                         fwd->error, &red);
      if (sd && i == 0 && next_first < n_steps && rc == FMX_OK) {
        if (g >= 1) (void)hipStreamWaitEvent(sd->stream, sd->consumed[(g + 1) & 1], 0);
        rc = sort_group(g + 1, next_first, group_size(g + 1, next_first), sd->stream);
        (void)hipEventRecord(sd->sorted[(g + 1) & 1], sd->stream);
      }
    }
    if (sd) (void)hipEventRecord(sd->consumed[g & 1], st);
    first_step = next_first;
  }
  if (detour) {
    (void)hipEventRecord(sd->user_join, st);
    (void)hipStreamWaitEvent(user, sd->user_join, 0);
  }
  return rc;
}

// ---- the field-owner step with the library's own communicator (fmx_comm.hip) ----
static int owner_geometry(const Comm *c, const fmx_table_t *table, int32_t B, int32_t slot, int &GB, const char *who) {
  if (!c) return fail(FMX_ERR_ARG, "%s: null communicator", who);
  if (int rc = check_table(table)) return rc;
  if (B < 1) return fail(FMX_ERR_ARG, "%s: B must be >= 1", who);
  if (slot < 0 || slot >= FMX_COMM_SLOTS) return fail(FMX_ERR_ARG, "%s: slot %d outside [0, %d)", who, slot, FMX_COMM_SLOTS);
  if ((int64_t)B * c->world > MAX_SORT_WIDTH) return fail(FMX_ERR_UNSUPPORTED, "%s: %d ranks x %d samples exceed one exact step (%d)", who, c->world, B, MAX_SORT_WIDTH);
  GB = B * c->world;
  return check_sort_geometry(table, GB);
}

int fmx_owner_prefetch(fmx_comm_t *comm, const fmx_table_t *table, const int32_t *idx_local, int32_t B, int32_t slot, int32_t *idx_all,
                       void *workspace, int64_t workspace_bytes, int32_t *error, fmx_stream_t stream) {
  Comm *c = reinterpret_cast<Comm *>(comm);
  int GB = 0;
  if (int rc = owner_geometry(c, table, B, slot, GB, "fmx_owner_prefetch")) return rc;
  if (!idx_local || (!idx_all && !(c->world == 1 && !c->force))) return fail(FMX_ERR_ARG, "fmx_owner_prefetch: null argument");
  if (int rc = check_workspace(table, GB, workspace, workspace_bytes, "fmx_owner_prefetch")) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream), pf = c->pf_stream;
  // behind whatever wrote idx_local on the caller's stream, and behind the step that last used this slot's buffers
  (void)hipEventRecord(c->fork, st);
  (void)hipStreamWaitEvent(pf, c->fork, 0);
  if (c->used[slot]) (void)hipStreamWaitEvent(pf, c->free_[slot], 0);
  const size_t words = (size_t)B * n_cols(table);
  // one rank, nothing forced: no copy -- the lists are sorted from idx_local itself, which the caller then also hands to
  // fmx_owner_step as idx_all (and keeps unchanged until that step has run)
  const bool alone = c->world == 1 && !c->force;
  if (!alone)
    if (int rc = comm_all_gather(c, 1, idx_local, idx_all, words, pf)) return rc;
  const Workspace w = carve(table, GB, workspace);
  if (int rc = sort_impl(table, alone ? idx_local : idx_all, GB, w.sorted, w.runs, error, pf)) return rc;
  (void)hipEventRecord(c->ready[slot], pf);
  return FMX_OK;
}

int fmx_owner_step(fmx_comm_t *comm, const fmx_table_t *table, const fmx_hyper_t *hyper, int32_t rule, int32_t loss_kind,
                   const int32_t *idx_all, const float *y_local, int32_t B, int32_t slot, void *workspace, int64_t workspace_bytes,
                   const fmx_owner_bufs_t *bufs, float *loss_out, int32_t *error, fmx_stream_t stream) {
  Comm *c = reinterpret_cast<Comm *>(comm);
  int GB = 0;
  if (int rc = owner_geometry(c, table, B, slot, GB, "fmx_owner_step")) return rc;
  if (int rc = check_rule(table, rule)) return rc;
  if (!hyper || !idx_all || !y_local || !bufs || !bufs->parts_send || !bufs->parts_recv || !bufs->rec_local || !bufs->rec_all)
    return fail(FMX_ERR_ARG, "fmx_owner_step: null argument");
  if (loss_kind != FMX_LOSS_BCE_LOGITS && loss_kind != FMX_LOSS_BCE_SIGMOID) return fail(FMX_ERR_ARG, "a step needs a loss");
  if (!aligned16(bufs->parts_send) || !aligned16(bufs->parts_recv) || !aligned16(bufs->rec_local) || !aligned16(bufs->rec_all))
    return fail(FMX_ERR_ALIGN, "fmx_owner_step: the record buffers must be 16-byte aligned");
  if (int rc = check_workspace(table, GB, workspace, workspace_bytes, "fmx_owner_step")) return rc;
  const bool alone = c->world == 1 && !c->force;
  if (alone ? false : (bufs->parts_recv == bufs->parts_send || bufs->rec_all == bufs->rec_local))
    return fail(FMX_ERR_ARG, "fmx_owner_step: with an exchange the receive buffers must be buffers of their own");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int kp = table->kp, rec_in = 2 * kp + 4, rec_out = kp + 4;
  const float inv_b = 1.0f / (float)GB;
  (void)hipStreamWaitEvent(st, c->ready[slot], 0);  // the slot's gather + sort
  // 1. every owned block's sub-tree for every sample of the global batch, destination-major
  if (int rc = part_impl(table, idx_all, nullptr, GB, c->n_blocks, c->block_count[c->rank], B, bufs->parts_send, error, st)) return rc;
  // 2. the records of this rank's samples from every block
  if (int rc = comm_exchange_blocks(c, bufs->parts_send, bufs->parts_recv, (size_t)B * rec_in, st)) return rc;
  // 3. the rest of the tree, bias, loss, dlogit -> one (S, dlogit, loss) record per local sample
  fmx_fwd_out_t out;
  memset(&out, 0, sizeof(out));
  out.S = bufs->rec_local;
  out.dz = bufs->rec_local + kp;
  out.loss = bufs->rec_local + kp + 1;
  out.sample_ld = rec_out;
  out.error = error;
  if (int rc = fmx_fm_forward_finish(hyper, table->bias, table->layout, kp, bufs->parts_recv, (int64_t)B * rec_in, c->n_blocks, y_local, B,
                                     loss_kind, inv_b, &out, stream))
    return rc;
  // 4. everybody's records
  if (int rc = comm_all_gather(c, 0, bufs->rec_local, bufs->rec_all, (size_t)B * rec_out, st)) return rc;
  // 5. the owned rows (and the replicated bias, identically everywhere)
  const Workspace w = carve(table, GB, workspace);
  const float *rec = bufs->rec_all;
  if (int rc = update_impl(table, hyper, rule, w, w.sorted, nullptr, rec, rec + kp, rec + kp, nullptr, GB, rec + kp + 1, inv_b, loss_out, st,
                           nullptr, rec_out, error))
    return rc;
  (void)hipEventRecord(c->free_[slot], st);
  c->used[slot] = true;
  return FMX_OK;
}

int fmx_fm_step(const fmx_table_t *table, const fmx_hyper_t *hyper, int32_t rule, int32_t loss_kind,
                const int32_t *idx, const float *xv, const float *y, int32_t B, float inv_b, void *workspace, int64_t workspace_bytes,
                const fmx_fwd_out_t *fwd, float *loss_out, fmx_stream_t stream) {
  if (int rc = check_step_args(table, hyper, rule, loss_kind, B, workspace, fwd)) return rc;
  if (!idx || !y) return fail(FMX_ERR_ARG, "fmx_fm_step: idx and y are required");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (int rc = check_workspace(table, B, workspace, workspace_bytes, "fmx_fm_step")) return rc;
  const Workspace w = carve(table, B, workspace);
  // One stream: sort -> forward -> update.  (Until r3 the sort ran on the side stream beside the forward pass; a dependency that
  // crosses streams costs 5 - 6 us on the device and four more runtime calls on the host, as much as the overlap of a 7 us forward
  // with an 18 us sort saves -- and a caller of single steps is host-bound: FMAdam.update_embedding 65 - 73 against 90 - 112 us
  // per batch, tools/class_surface_profile.py.  Loops over many batches: fmx_fm_stream, where a sort launch covers 16 of them.)
  if (int rc = sort_impl(table, idx, B, w.sorted, w.runs, fwd->error, st)) return rc;
  if (int rc = forward_impl(table, hyper, idx, xv, y, B, loss_kind, inv_b, fwd, st)) return rc;
  return update_impl(table, hyper, rule, w, w.sorted, xv, fwd->S, fwd->dz, fwd->dz, nullptr, B, fwd->loss, inv_b, loss_out, st,
                     nullptr, fwd->sample_ld, fwd->error);
}

int fmx_fm_stream(const fmx_table_t *table, const fmx_hyper_t *hyper, int32_t rule, int32_t loss_kind,
                  const int32_t *idx_pool, const float *y_pool, int32_t n_pool, int32_t B, float inv_b,
                  int32_t n_steps, void *workspace, int64_t workspace_bytes, const fmx_fwd_out_t *fwd, float *loss_out, float *kernel_ms,
                  fmx_stream_t stream) {
  if (int rc = check_step_args(table, hyper, rule, loss_kind, B, workspace, fwd)) return rc;
  if (!idx_pool || !y_pool || n_pool < 1 || n_steps < 0) return fail(FMX_ERR_ARG, "fmx_fm_stream: bad pool / step count");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (int rc = check_workspace(table, B, workspace, workspace_bytes, "fmx_fm_stream")) return rc;
  const Workspace w = carve(table, B, workspace);
  const size_t F = (size_t)n_cols(table);
  int rc = FMX_OK;

  if (!kernel_ms) {
    // production path.  The occurrence sort does not depend on the weights: groups of `ahead` batches are sorted by ONE
    // launch on the side stream while the previous group runs forward / update / fixup on `stream`.  Ring of
    // 2 * ahead sorted buffers; per group one sort launch and four event operations, so the host issues ~3.6 runtime
    // calls per step instead of 8 (at ~4 us each the per-batch version was host-bound).
    Side *sd = (B >= OVERLAP_MIN_BATCH && n_steps > 0) ? side_for_current_device() : nullptr;
    hipStream_t user = st;
    const bool detour = sd && st == nullptr;  // the legacy default stream cannot be captured / is slow to enqueue on
    if (detour) {
      (void)hipEventRecord(sd->user_fork, user);
      st = sd->main;
      (void)hipStreamWaitEvent(st, sd->user_fork, 0);
    }
    auto rejoin = [&](int r) -> int {
      if (detour) {
        (void)hipEventRecord(sd->user_join, st);
        (void)hipStreamWaitEvent(user, sd->user_join, 0);
      }
      return r;
    };
    int ahead = tune().sort_ahead;
    if (ahead < 1) ahead = 1;
    if (ahead > SORT_AHEAD_MAX) ahead = SORT_AHEAD_MAX;
    // the first group holds up to 4 batches, the others `ahead` (8): one launch sorts four batches in about the time of one (one
    // workgroup per field and batch: 18.4 against 17.1 us, tools/micro/sort_bench.hip), so the first update waits no longer than
    // behind a one-batch group, and a short call issues three sort launches and their events instead of five while the device
    // is still waiting for the host.  Group g uses half (g & 1) of the ring of 2 * ahead sorted buffers.
    auto group_size = [&](int g, int first_step) {
      int n = g == 0 ? 4 : ahead;
      if (n > ahead) n = ahead;
      if (n > n_steps - first_step) n = n_steps - first_step;
      return n;
    };
    auto sort_group = [&](int g, int first_step, int n, hipStream_t where) -> int {  // steps [first_step, first_step + n)
      SortBatch mb;
      mb.n_pool = n_pool;
      mb.first = first_step % n_pool;
      mb.n_batches = n;
      mb.pool_stride = (int64_t)B * (int64_t)F;
      mb.sorted_stride = (int64_t)w.sorted_stride;
      return sort_impl(table, idx_pool, B, w.sorted + (size_t)(g & 1) * ahead * w.sorted_stride, w.runs, fwd->error, where, &mb);
    };
    if (sd && n_steps > 0) {
      (void)hipEventRecord(sd->fork, st);
      (void)hipStreamWaitEvent(sd->stream, sd->fork, 0);
      rc = sort_group(0, 0, group_size(0, 0), sd->stream);
      (void)hipEventRecord(sd->sorted[0], sd->stream);
    }
    int first_step = 0;
    for (int g = 0; first_step < n_steps && rc == FMX_OK; ++g) {
      const int n = group_size(g, first_step);
      const int next_first = first_step + n;
      if (!sd) rc = sort_group(g, first_step, n, st);
      for (int i = 0; i < n && rc == FMX_OK; ++i) {
        const int s = first_step + i, j = s % n_pool;
        const int32_t *idx = idx_pool + (size_t)j * B * F;
        const float *y = y_pool + (size_t)j * B;
        const uint32_t *sorted = w.sorted + ((size_t)(g & 1) * ahead + i) * w.sorted_stride;
        rc = forward_impl(table, hyper, idx, nullptr, y, B, loss_kind, inv_b, fwd, st);
        if (sd && i == 0) (void)hipStreamWaitEvent(st, sd->sorted[g & 1], 0);
        if (rc == FMX_OK)
          rc = update_impl(table, hyper, rule, w, sorted, nullptr, fwd->S, fwd->dz, fwd->dz, nullptr, B, fwd->loss, inv_b,
                           loss_out ? loss_out + s : nullptr, st, nullptr, fwd->sample_ld, fwd->error);
        // the next group is sorted while this one runs; its launch is issued BEHIND the group's first step, so that at the start
        // of a call -- the device idle, every launch waiting for the host -- the first forward and update are not held up by it
        if (sd && i == 0 && next_first < n_steps && rc == FMX_OK) {
          if (g >= 1) (void)hipStreamWaitEvent(sd->stream, sd->consumed[(g + 1) & 1], 0);  // group g-1 is done with that half
          rc = sort_group(g + 1, next_first, group_size(g + 1, next_first), sd->stream);
          (void)hipEventRecord(sd->sorted[(g + 1) & 1], sd->stream);
        }
      }
      if (sd) (void)hipEventRecord(sd->consumed[g & 1], st);
      first_step = next_first;
    }
    return rejoin(rc);
  }

  // measuring mode: everything on `stream`.  An event pair costs several microseconds of its own on this stack (slot 3
  // measures exactly that: two records with nothing between), so launches are timed in groups of up to 8 steps between
  // two events and the pair's own cost is subtracted: ONE sort launch for the group's batches (as in the production
  // loop), then the group's forwards back to back, then its updates back to back -- every launch on a DIFFERENT batch of
  // the pool (its own sorted list, its own S / dz / loss), so the rows come from HBM / MALL as they do in production
  // instead of from an L2 warmed by the previous launch of the same batch.  The forwards of a group all read the table
  // before the group's updates, so this pass is a measurement, not the online algorithm.
  const int REP = 8, n_ev = 5;
  hipStream_t user_t = st;
  Side *sdt = (st == nullptr) ? side_for_current_device() : nullptr;  // same detour off the legacy stream as above
  if (sdt) {
    (void)hipEventRecord(sdt->user_fork, user_t);
    st = sdt->main;
    (void)hipStreamWaitEvent(st, sdt->user_fork, 0);
  }
  const int n_groups = (n_steps + REP - 1) / REP;
  const size_t region = align_up((size_t)B * table->kp + 2 * (size_t)B, 64);  // floats: S | dz | loss of one batch
  float *tmp = nullptr;
  if (n_groups > 0 && hipMalloc(&tmp, REP * region * sizeof(float)) != hipSuccess) return fail(FMX_ERR_LAUNCH, "hipMalloc (measuring mode)");
  hipEvent_t *ev = new hipEvent_t[(size_t)n_groups * n_ev];
  for (int i = 0; i < n_groups * n_ev; ++i) (void)hipEventCreate(&ev[i]);
  for (int g = 0; g < n_groups && rc == FMX_OK; ++g) {
    const int first = g * REP, n = (n_steps - first) < REP ? (n_steps - first) : REP;
    hipEvent_t *e = ev + (size_t)g * n_ev;
    (void)hipEventRecord(e[0], st);
    {
      SortBatch mb;
      mb.n_pool = n_pool;
      mb.first = first % n_pool;
      mb.n_batches = n;
      mb.pool_stride = (int64_t)B * (int64_t)F;
      mb.sorted_stride = (int64_t)w.sorted_stride;
      rc = sort_impl(table, idx_pool, B, w.sorted, w.runs, fwd->error, st, &mb);
    }
    (void)hipEventRecord(e[1], st);
    for (int r = 0; r < n && rc == FMX_OK; ++r) {
      const int j = (first + r) % n_pool;
      fmx_fwd_out_t fr = *fwd;
      fr.S = tmp + (size_t)r * region;
      fr.dz = fr.S + (size_t)B * table->kp;
      fr.loss = fr.dz + B;
      fr.sample_ld = 0;
      rc = forward_impl(table, hyper, idx_pool + (size_t)j * B * F, nullptr, y_pool + (size_t)j * B, B, loss_kind, inv_b, &fr, st);
    }
    (void)hipEventRecord(e[2], st);
    for (int r = 0; r < n && rc == FMX_OK; ++r) {
      const float *S = tmp + (size_t)r * region, *dz = S + (size_t)B * table->kp;
      rc = update_impl(table, hyper, rule, w, w.sorted + (size_t)r * w.sorted_stride, nullptr, S, dz, dz, nullptr, B, dz + B, inv_b,
                       loss_out ? loss_out + first + r : nullptr, st, nullptr, 0, fwd->error);
    }
    (void)hipEventRecord(e[3], st);
    (void)hipEventRecord(e[4], st);
  }
  (void)hipStreamSynchronize(st);
  if (sdt) {
    (void)hipEventRecord(sdt->user_join, st);
    (void)hipStreamWaitEvent(user_t, sdt->user_join, 0);
  }
  for (int k = 0; k < 4; ++k) kernel_ms[k] = 0.f;
  if (rc == FMX_OK && n_groups > 0) {
    double t_sort = 0, t_fwd = 0, t_upd = 0, t_pair = 0;
    for (int g = 0; g < n_groups; ++g) {
      hipEvent_t *e = ev + (size_t)g * n_ev;
      float pair = 0.f, ms[3] = {0.f, 0.f, 0.f};
      (void)hipEventElapsedTime(&pair, e[3], e[4]);
      for (int k = 0; k < 3; ++k) (void)hipEventElapsedTime(&ms[k], e[k], e[k + 1]);
      t_pair += pair;
      t_sort += fmaxf(ms[0] - pair, 0.f);
      t_fwd += fmaxf(ms[1] - pair, 0.f);
      t_upd += fmaxf(ms[2] - pair, 0.f);
    }
    // returned as (average per launch) x n_steps, so that dividing by n_steps gives the per-launch averages
    kernel_ms[0] = (float)(t_sort / n_groups * n_steps);  // one launch per group of up to 8 batches, as in production
    kernel_ms[1] = (float)t_fwd;                           // n_steps forward launches in all
    kernel_ms[2] = (float)t_upd;
    kernel_ms[3] = (float)(t_pair / n_groups * n_steps);
  }
  for (int i = 0; i < n_groups * n_ev; ++i) (void)hipEventDestroy(ev[i]);
  delete[] ev;
  if (tmp) (void)hipFree(tmp);
  return rc;
}

int fmx_fm_online_run(const fmx_table_t *table, const fmx_hyper_t *hyper, int32_t rule, int32_t loss_kind,
                      const int32_t *idx, const float *xv, const float *y, int32_t N, uint8_t *pred_out, float *loss_out,
                      int32_t *error, fmx_stream_t stream) {
  if (int rc = check_table(table)) return rc;
  if (int rc = check_rule(table, rule)) return rc;
  if (mapped(table)) return fail(FMX_ERR_UNSUPPORTED, "fmx_fm_online_run: tables whose fields are pieces of index columns are not taken");
  if (N < 0) return fail(FMX_ERR_ARG, "fmx_fm_online_run: N must be >= 0");
  if (N == 0) return FMX_OK;  // an empty stream (its buffers may be null)
  if (!hyper || !idx || !y || !pred_out) return fail(FMX_ERR_ARG, "fmx_fm_online_run: null argument");
  if (loss_kind != FMX_LOSS_BCE_LOGITS && loss_kind != FMX_LOSS_BCE_SIGMOID) return fail(FMX_ERR_ARG, "fit needs a loss");
  const int lpr = lpr_of(table->kp), slots = WAVE / lpr;
  const int np = (table->n_fields + slots - 1) / slots;
  if (np > 4)
    return fail(FMX_ERR_UNSUPPORTED, "fmx_fm_online_run: %d fields at kp = %d exceed the %d rows one wavefront holds", table->n_fields,
                table->kp, 4 * slots);
  OnlineArgs a;
  a.rows = table->rows;
  a.foff = table->field_offsets;
  a.bias = table->bias;
  a.idx = idx;
  a.xv = xv;
  a.y = y;
  a.pred = pred_out;
  a.loss = loss_out;
  a.error = error;
  a.h = *hyper;
  a.h.alpha = 1.0f / hyper->alpha;  // the kernels multiply by 1/alpha
  a.N = N;
  a.F = table->n_fields;
  a.stride = table->row_stride;
  a.zoff = table->z_offset;
  a.loss_kind = loss_kind;
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (lpr) {
    case 1: launch_online<1>(a, rule, np, st); break;
    case 2: launch_online<2>(a, rule, np, st); break;
    case 4: launch_online<4>(a, rule, np, st); break;
    case 8: launch_online<8>(a, rule, np, st); break;
    default: launch_online<16>(a, rule, np, st); break;
  }
  return check_launch("k_fm_online");
}

static int mlp_launch(const fmx_mlp_t *mlp, MlpArgs &a, int32_t B, int32_t kp, fmx_stream_t stream, const char *who) {
  if (!mlp || !mlp->params) return fail(FMX_ERR_ARG, "%s: null mlp", who);
  if (mlp->n_layers < 1 || mlp->n_layers > MLP_MAX_L || mlp->hidden < 1 || mlp->hidden > MLP_MAX_W || mlp->k < 1 ||
      mlp->k > MLP_MAX_W || B < 1 || B > MLP_MAX_B || kp < mlp->k)
    return fail(FMX_ERR_UNSUPPORTED, "%s: needs B <= %d, k <= %d, hidden <= %d, layers <= %d (got B=%d k=%d hidden=%d layers=%d)", who,
                MLP_MAX_B, MLP_MAX_W, MLP_MAX_W, MLP_MAX_L, B, mlp->k, mlp->hidden, mlp->n_layers);
  a.params = mlp->params;
  a.B = B;
  a.k = mlp->k;
  a.kp = kp;
  a.hidden = mlp->hidden;
  a.n_layers = mlp->n_layers;
  hipLaunchKernelGGL(k_mlp_small, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return check_launch("k_mlp_small");
}

int fmx_online_run_mlp(const fmx_table_t *table, const fmx_hyper_t *hyper, int32_t rule, int32_t loss_kind,
                       const fmx_mlp_t *mlp, int32_t hedge, int32_t fm_term, float hedge_b, float hedge_s, float *alpha,
                       const int32_t *idx, const float *xv, const float *y, int32_t N, void *workspace, int64_t workspace_bytes,
                       const fmx_fwd_out_t *fwd, float *scratch, float *pred_out, fmx_stream_t stream) {
  if (int rc = check_table(table)) return rc;
  if (!hyper || !mlp || !idx || !y || !workspace || !fwd || !scratch || !pred_out)
    return fail(FMX_ERR_ARG, "fmx_online_run_mlp: null argument");
  if (mapped(table)) return fail(FMX_ERR_UNSUPPORTED, "fmx_online_run_mlp: tables whose fields are pieces of index columns are not taken");
  if (!fwd->S || !fwd->bi || !fwd->sfirst || !fwd->logit) return fail(FMX_ERR_ARG, "fmx_online_run_mlp: fwd needs S, bi, sfirst, logit");
  if (!aligned16(workspace) || !aligned16(scratch)) return fail(FMX_ERR_ALIGN, "workspace and scratch must be 16-byte aligned");
  if (hedge && !alpha) return fail(FMX_ERR_ARG, "fmx_online_run_mlp: Hedge needs alpha");
  if (!hedge) {
    if (int rc = check_rule(table, rule)) return rc;
    if (rule != FMX_RULE_SIGNADAM && rule != FMX_RULE_SGD) return fail(FMX_ERR_ARG, "fmx_online_run_mlp: rule must be SIGNADAM or SGD");
    if (loss_kind != FMX_LOSS_BCE_LOGITS && loss_kind != FMX_LOSS_BCE_SIGMOID) return fail(FMX_ERR_ARG, "fit needs a loss");
    if (mlp->k > MLP_MAX_W - 1) return fail(FMX_ERR_UNSUPPORTED, "fmx_online_run_mlp: k <= %d", MLP_MAX_W - 1);
    if (int rc = check_sort_geometry(table, 1)) return rc;
  } else if (mlp->k + mlp->n_layers > MLP_MAX_W) {
    return fail(FMX_ERR_UNSUPPORTED, "fmx_online_run_mlp: k + layers <= %d", MLP_MAX_W);
  }
  if (N < 0) return fail(FMX_ERR_ARG, "N must be >= 0");
  hipStream_t st = static_cast<hipStream_t>(stream);
  {  // one workgroup walks the stream when the network fits in LDS and the fields fit one wavefront (k_online_mlp)
    long long n_params = 0;
    for (int l = 0; l < mlp->n_layers; ++l) n_params += (long long)mlp->hidden * (l == 0 ? mlp->k : mlp->hidden) + mlp->hidden;
    const int lpr = lpr_of(table->kp), slots = WAVE / lpr;
    const bool tables_ok = hedge || table->layout == FMX_LAYOUT_WEIGHTS;
    if (tune().online_persistent && n_params <= ONLINE_MLP_MAX_PARAMS && table->n_fields <= 4 * slots && tables_ok &&
        mlp->hidden <= MLP_MAX_W && mlp->n_layers <= MLP_MAX_L && mlp->k <= MLP_MAX_W - 1 && N > 0) {
      OnlineMlpArgs a;
      a.rows = table->rows;
      a.foff = table->field_offsets;
      a.bias = table->bias;
      a.idx = idx;
      a.xv = xv;
      a.y = y;
      a.pred = pred_out;
      a.error = fwd->error;
      a.params = mlp->params;
      a.alpha = alpha;
      a.h = *hyper;
      a.h.alpha = 1.0f / hyper->alpha;
      a.hedge_b = hedge_b;
      a.hedge_s = hedge_s;
      a.N = N;
      a.F = table->n_fields;
      a.stride = table->row_stride;
      a.zoff = table->z_offset;
      a.n_params = (int32_t)n_params;
      a.k = mlp->k;
      a.hidden = mlp->hidden;
      a.n_layers = mlp->n_layers;
      a.hedge = hedge;
      a.fm_term = fm_term;
      a.rule = rule;
      a.loss_kind = loss_kind;
      switch (lpr) {
        case 1: launch_online_mlp<1>(a, table->layout, rule, st); break;
        case 2: launch_online_mlp<2>(a, table->layout, rule, st); break;
        case 4: launch_online_mlp<4>(a, table->layout, rule, st); break;
        case 8: launch_online_mlp<8>(a, table->layout, rule, st); break;
        default: launch_online_mlp<16>(a, table->layout, rule, st); break;
      }
      return check_launch("k_online_mlp");
    }
  }
  if (int rc = check_workspace(table, 1, workspace, workspace_bytes, "fmx_online_run_mlp")) return rc;
  const Workspace w = carve(table, 1, workspace);
  const size_t F = (size_t)table->n_fields;
  fmx_fwd_out_t f1 = *fwd;  // one sample: dense outputs
  f1.sample_ld = 0;
  float *dz = scratch, *gbi = scratch + 8;
  for (int i = 0; i < N; ++i) {
    const int32_t *idx_i = idx + (size_t)i * F;
    const float *xv_i = xv ? xv + (size_t)i * F : nullptr;
    if (int rc = forward_impl(table, hyper, idx_i, xv_i, nullptr, 1, FMX_LOSS_NONE, 1.0f, &f1, st)) return rc;
    MlpArgs a{};
    a.bi = fwd->bi;
    a.base = fm_term ? fwd->logit : fwd->sfirst;
    if (!fm_term) {  // NFM: the logit without the MLP term is the first-order sum plus the bias weight
      a.base_bias = table->bias;
      a.base_bias_ftrl = table->layout == FMX_LAYOUT_FTRL;
      a.h_table = *hyper;
      a.h_table.alpha = 1.0f / hyper->alpha;
    }
    a.y = y + i;
    a.pred_out = pred_out + i;
    a.h = *hyper;
    a.inv_b = 1.0f;
    if (hedge) {
      a.alpha = alpha;
      a.hedge_b = hedge_b;
      a.hedge_s = hedge_s;
      a.mode = MLP_MODE_HEDGE;
    } else {
      a.dz_out = dz;
      a.gbi_out = gbi;
      a.mode = MLP_MODE_FIT;
      a.rule = rule;
      a.loss_kind = loss_kind;
    }
    if (int rc = mlp_launch(mlp, a, 1, table->kp, stream, "fmx_online_run_mlp")) return rc;
    if (hedge) continue;  // Hedge trains the hidden layers and alpha only (reference deepfm_onn.py:109-154)
    if (int rc = sort_impl(table, idx_i, 1, w.sorted, w.runs, fwd->error, st)) return rc;
    if (int rc = update_impl(table, hyper, rule, w, w.sorted, xv_i, fwd->S, dz, fm_term ? dz : nullptr, gbi, 1, nullptr, 1.0f, nullptr,
                             st, nullptr, 0, fwd->error))
      return rc;
  }
  return FMX_OK;
}

int fmx_mlp_forward(const fmx_mlp_t *mlp, const float *bi, int32_t kp, const float *base, int32_t B, float *out,
                    float *layers_out, fmx_stream_t stream) {
  if (!bi || !base || (!out && !layers_out)) return fail(FMX_ERR_ARG, "fmx_mlp_forward: null argument");
  MlpArgs a{};
  a.bi = bi;
  a.base = base;
  a.out = out;
  a.layers_out = layers_out;
  a.mode = MLP_MODE_FORWARD;
  return mlp_launch(mlp, a, B, kp, stream, "fmx_mlp_forward");
}

int fmx_mlp_fit(const fmx_mlp_t *mlp, const fmx_hyper_t *hyper, int32_t rule, int32_t loss_kind, const float *bi, int32_t kp,
                const float *base, const float *y, int32_t B, float inv_b, float *dz_out, float *gbi_out, float *loss_out,
                fmx_stream_t stream) {
  if (!hyper || !bi || !base || !y || !dz_out || !gbi_out) return fail(FMX_ERR_ARG, "fmx_mlp_fit: null argument");
  if (rule != FMX_RULE_SIGNADAM && rule != FMX_RULE_SGD) return fail(FMX_ERR_ARG, "fmx_mlp_fit: rule must be SIGNADAM or SGD");
  if (loss_kind != FMX_LOSS_BCE_LOGITS && loss_kind != FMX_LOSS_BCE_SIGMOID) return fail(FMX_ERR_ARG, "fmx_mlp_fit needs a loss");
  if (mlp && mlp->k > MLP_MAX_W - 1) return fail(FMX_ERR_UNSUPPORTED, "fmx_mlp_fit: k <= %d", MLP_MAX_W - 1);
  MlpArgs a{};
  a.bi = bi;
  a.base = base;
  a.y = y;
  a.dz_out = dz_out;
  a.gbi_out = gbi_out;
  a.out = loss_out;
  a.h = *hyper;
  a.mode = MLP_MODE_FIT;
  a.rule = rule;
  a.loss_kind = loss_kind;
  a.inv_b = inv_b;
  return mlp_launch(mlp, a, B, kp, stream, "fmx_mlp_fit");
}

int fmx_mlp_hedge_fit(const fmx_mlp_t *mlp, float lr, float hedge_b, float hedge_s, float *alpha, const float *bi, int32_t kp,
                      const float *base, const float *y, int32_t B, float *losses_out, fmx_stream_t stream) {
  if (!alpha || !bi || !base || !y) return fail(FMX_ERR_ARG, "fmx_mlp_hedge_fit: null argument");
  if (mlp && mlp->k + mlp->n_layers > MLP_MAX_W) return fail(FMX_ERR_UNSUPPORTED, "fmx_mlp_hedge_fit: k + layers <= %d", MLP_MAX_W);
  MlpArgs a{};
  a.bi = bi;
  a.base = base;
  a.y = y;
  a.alpha = alpha;
  a.layers_out = losses_out;
  a.h.lr = lr;
  a.hedge_b = hedge_b;
  a.hedge_s = hedge_s;
  a.mode = MLP_MODE_HEDGE;
  a.inv_b = 1.0f / (float)B;
  return mlp_launch(mlp, a, B, kp, stream, "fmx_mlp_hedge_fit");
}

int fmx_gather_read(const void *buf, int64_t bytes, int32_t row_bytes, int64_t n_rows_read, uint32_t seed, float *sink, fmx_stream_t stream) {
  if (!buf || !sink || bytes < 128 || n_rows_read < 1) return fail(FMX_ERR_ARG, "fmx_gather_read: bad buffer / count");
  if (row_bytes != 64 && row_bytes != 128) return fail(FMX_ERR_ARG, "fmx_gather_read: rows of 64 or 128 bytes");
  if (!aligned16(buf)) return fail(FMX_ERR_ALIGN, "fmx_gather_read: buffer must be 16-byte aligned");
  const int lpr = row_bytes / 16;
  const uint64_t n_rows = (uint64_t)bytes / (uint64_t)row_bytes;
  const int64_t threads = n_rows_read * lpr;
  const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (lpr == 4) hipLaunchKernelGGL((k_gather_read<4>), grid, block, 0, st, static_cast<const float4 *>(buf), n_rows, n_rows_read, seed, sink);
  else hipLaunchKernelGGL((k_gather_read<8>), grid, block, 0, st, static_cast<const float4 *>(buf), n_rows, n_rows_read, seed, sink);
  return check_launch("k_gather_read");
}

#ifdef FMX_STAMPS
int fmx_debug_update_stamps(unsigned long long *host_out) {  // [8192][6]; diagnostic build only (not in include/fmx.h)
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_upd_stamps), sizeof(unsigned long long) * 8192 * 6) == hipSuccess ? FMX_OK : FMX_ERR_LAUNCH;
}
#endif

int fmx_stream_read(const void *buf, int64_t bytes, float *sink, fmx_stream_t stream) {
  if (!buf || !sink || bytes < 16 || bytes % 16) return fail(FMX_ERR_ARG, "fmx_stream_read: bad buffer");
  if (!aligned16(buf)) return fail(FMX_ERR_ALIGN, "fmx_stream_read: buffer must be 16-byte aligned");
  hipLaunchKernelGGL(k_stream_read, dim3(256 * 8), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const float4 *>(buf), bytes / 16, sink);
  return check_launch("k_stream_read");
}

}  // extern "C"
