/* fmx.h -- C ABI of libfmx.so: the MI355X (gfx950) kernels behind the FM / DeepFM / NFM online hot path.
 *
 * The reference (haan6/fm-for-online-recommendation) has no FFI layer: its hot path is sequences of ATen ops
 * inside Python classes.  Each entry point below therefore cites the reference *op sequence* it replaces
 * (paths relative to the reference repository).  The Python classes in
 * fm-for-online-recommendation_amd/models/ bind these with ctypes (fm-for-online-recommendation_amd/fmx/_lib.py);
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned memory unless the comment says "host".  The data path allocates
 *     nothing: tables, batches, outputs and the per-step workspace (fmx_workspace_bytes) are the caller's.  What the library
 *     DOES own, per process: (a) per device, created on first use and never destroyed, two non-blocking HIP streams (one for
 *     the occurrence sorts that run beside the steps, one that stands in for the legacy default stream) and a handful of
 *     timing-disabled events ordering them against `stream`; (b) the tuning switches of fmx_set_option (process-wide, read
 *     by every call); (c) a launch sequence counter tagging the in-launch hand-offs; (d) fmx_fm_stream's MEASURING mode
 *     (kernel_ms != null) alone creates its HIP events and one temporary device buffer per call and frees them before it
 *     returns -- it synchronises the stream and is a benchmark facility, not part of the data path;
 *   - every other call is asynchronous on `stream` (a hipStream_t passed as void*) and performs no implicit
 *     synchronisation; work the library puts on its own streams is ordered behind what `stream` held at the call and
 *     `stream` is ordered behind it before the call returns.  Calls are safe to capture into a hipGraph except fmx_fm_stream
 *     (cross-stream events, optional timing); a capturing stream takes the paths without in-launch hand-offs;
 *   - return value: 0 on success, a negative fmx_status otherwise; the message for the calling thread is
 *     available from fmx_last_error_string();
 *   - streams: fmx_fm_stream / fmx_deepfm_stream sort on a library-owned low-priority side stream beside the caller's stream.  HIP maps
 *     the streams of a process onto its hardware queues: a caller's stream that shares a queue with the side stream runs BEHIND the
 *     sorts (same results, 1.5 - 4 x the time per step).  Measured (tools/queue_alias.py): never with the runtime's default
 *     GPU_MAX_HW_QUEUES = 4; with 8 for the 4th and 11th stream a process creates, with 16 for every fourth.  Reuse one stream per
 *     loop.  The legacy default stream (NULL) is detoured through a library-owned one.
 *   - the library never throws.  Thread safety: calls on different tables / workspaces / streams may run concurrently;
 *     the mutable process state is (a)-(c) above plus the thread-local error string;
 *   - device-side conditions are reported through the caller's int32 error word (fmx_fwd_out_t.error and the `error`
 *     arguments): 1 = an index outside its field (that row is treated as absent), 2 = an in-launch hand-off of the update ran
 *     into its spin bound (the row update of that run was SKIPPED: the table is no longer the exact result).  2 means
 *     a dispatch-order assumption was violated; it has never been observed and is there so that a fault ends in a
 *     flag, not in a hang.
 *
 * Table layout in HBM (one flat buffer for all fields; field f owns rows [field_offsets[f], field_offsets[f+1])):
 *   FMX_LAYOUT_WEIGHTS  row = [ V[0..kp) | w | pad ]                                  row_stride >= kp + 4
 *   FMX_LAYOUT_FTRL     row = [ V[0..kp) | w, zw, nw, 0 | pad | zV[0..kp) | nV[0..kp) ]    zV at float z_offset,
 *                                                                                     row_stride >= z_offset + 2*kp
 * kp is k rounded up to 4, 8, 16, 32 or 64; the pad components must be zero (they then stay zero under every
 * rule).  row_stride and z_offset are in floats and multiples of 4 (16-byte pieces).  Both layouts start with the
 * weights the forward pass reads, so a forward gather is ONE 64-byte request per row (k = 16) inside one 128-byte
 * line.  In the FTRL layout the state is (z, n); V and w are the weights derived from it,
 *     w = 0 if |z| <= l1 else -(z - sgn(z) l1) / ((beta + sqrt(n)) / alpha + l2)      (McMahan et al. 2013),
 * re-derived and stored by every update -- a cache, never an independent parameter: whoever writes (z, n) or changes
 * the hyper-parameters must rewrite V and w with the same formula.
 */
#ifndef FMX_H
#define FMX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FMX_VERSION 104 /* 0.1.4: fmx_deepfm_stream (0.1.3: fields as row-range pieces of index columns -- field_cols / field_base /
                           n_cols --, workspace_bytes arguments, fmx_owner_*) */

typedef void *fmx_stream_t; /* hipStream_t */

enum fmx_status {
  FMX_OK = 0,
  FMX_ERR_ARG = -1,         /* null pointer / negative size / unknown enum */
  FMX_ERR_SHAPE = -2,       /* sizes inconsistent with each other (kp, row_stride, Bp, bbits ...) */
  FMX_ERR_ALIGN = -3,       /* a pointer that must be 16-byte aligned is not */
  FMX_ERR_LAUNCH = -4,      /* hipGetLastError() after a launch, or another HIP runtime error */
  FMX_ERR_UNSUPPORTED = -5  /* valid request the kernels do not cover (batch too large for the LDS sort, ...) */
};

enum fmx_layout { FMX_LAYOUT_WEIGHTS = 0, FMX_LAYOUT_FTRL = 1 };

/* per-coordinate update rules (g = gradient summed over every occurrence of the row in the mini-batch) */
enum fmx_rule {
  FMX_RULE_SIGNADAM = 0, /* p -= lr * g / (|g| + eps): what a fresh torch.optim.Adam per call reduces to
                            (reference fm_adam.py:60,68 / :75,82).  FMX_LAYOUT_WEIGHTS */
  FMX_RULE_SGD = 1,      /* p -= lr * g (stale notebook prototypes only; parity unpinned).  FMX_LAYOUT_WEIGHTS */
  FMX_RULE_FTRL = 2      /* FTRL-proximal on (z, n) (not in the reference; parity unpinned).  FMX_LAYOUT_FTRL */
};

/* loss applied to the FM logit z in the fused epilogue of fmx_fm_forward */
enum fmx_loss {
  FMX_LOSS_NONE = 0,
  FMX_LOSS_BCE_LOGITS = 1, /* BCEwl(z, y)            reference fm_adam.py:61,66 */
  FMX_LOSS_BCE_SIGMOID = 2 /* BCEwl(sigmoid(z), y)   reference fm_adam.py:76,80 (the "double sigmoid") */
};

typedef struct fmx_table {
  float *rows;                  /* [n_rows, row_stride] */
  const int64_t *field_offsets; /* [n_fields + 1] prefix sums of the per-field vocabulary sizes */
  float *bias;                  /* WEIGHTS: [1] = bias;  FTRL: [2] = (z, n) of the bias */
  int64_t n_rows;
  int32_t n_fields;
  int32_t k;          /* embedding size */
  int32_t kp;         /* k padded to 4 / 8 / 16 / 32 / 64 */
  int32_t row_stride; /* floats */
  int32_t layout;     /* enum fmx_layout */
  int32_t z_offset;   /* FTRL: float offset of zV inside the row (multiple of 4, >= kp + 4); WEIGHTS: ignored */
  int64_t max_field_rows; /* largest per-field vocabulary (host copy; bounds the sort's composite keys) */
  /* SORT FIELDS (optional; n_sort_fields = 0: the fields themselves).  The occurrence lists and the update work per "sort
   * field".  An occurrence list packs (index, sample) into 32 bits, so a field of 176,373 rows (18 bits) leaves 14 bits for
   * the sample: 16,384 samples per exact step.  A field may therefore be cut into consecutive PIECES of at most
   * max_sort_field_rows rows, each sorted and updated as a field of its own (a sample appears in the list of the one piece
   * its index falls into): sort_offsets [n_sort_fields + 1] (host-built, device-resident) refines field_offsets,
   * sort_cols [n_sort_fields] names the field every piece belongs to.  The forward pass is unaffected. */
  const int64_t *sort_offsets;
  const int32_t *sort_cols;
  int32_t n_sort_fields;
  int32_t reserved;
  int64_t max_sort_field_rows;
  /* FIELDS AS PIECES OF INDEX COLUMNS (optional; all null / 0: field f holds every index of column f).  For the multi-GPU
   * field-owner mode a "field" of a table is a consecutive row range of one column of idx: field f reads column
   * field_cols[f] and holds its indices [field_base[f], field_base[f] + rows_f) -- a sample whose index lies outside belongs
   * to another piece (another field of this table, or a field of another owner's table) and contributes nothing here.  A
   * field may be EMPTY (0 rows): a hole in the forward tree.  The forward pass adds the fields in position order (lane
   * group f % SLOTS, pass f / SLOTS, SLOTS = 64 / (kp / 4)), so which piece sits at which field number decides the order
   * of the floating-point additions: tables that place the same pieces at the same positions give identical bits however
   * the positions are dealt over owners.  With pieces the kernels cannot tell a bad index from one of another piece:
   * device-side range errors (error word 1) are reported for unmapped tables only. */
  const int32_t *field_cols; /* [n_fields] or null */
  const int32_t *field_base; /* [n_fields] or null */
  int32_t n_cols;            /* columns of idx / xv; 0: n_fields */
  int32_t reserved2;
} fmx_table_t;

typedef struct fmx_hyper {
  float lr, eps;             /* SIGNADAM / SGD */
  float alpha, beta, l1, l2; /* FTRL */
} fmx_hyper_t;

/* Outputs of the forward pass.  Any pointer may be null except S when an update follows. */
typedef struct fmx_fwd_out {
  float *S;       /* [B, kp]  S_b = sum_f V[row_bf] * x_bf            (kept for the update)        */
  float *bi;      /* [B, kp]  0.5 * (S*S - sum_f e*e)                  reference second_order()     */
  float *first;   /* [B, F]   w[row_bf] * x_bf                         reference first_order()      */
  float *sfirst;  /* [B]      sum_f first                                                             */
  float *sbi;     /* [B]      sum_d bi                                                                */
  float *logit;   /* [B]      sfirst + sbi + bias                      reference forward_fm()       */
  float *loss;    /* [B]      per-sample loss (unscaled), needs y                                     */
  float *dz;      /* [B]      d(mean loss)/d logit = (...) * inv_b, needs y                           */
  int32_t *error; /* [1]      set to 1 when an index is outside its field (that row is treated as 0)  */
  int32_t sample_ld; /* 0: S, loss, dz are dense arrays as above.  > 0 (multiple of 4, >= kp): element b of S, of loss and
                        of dz lies sample_ld floats after element b-1 -- the three pointers then address fields of one
                        per-sample record, which a data-parallel caller all-gathers with ONE collective */
  int32_t reserved;
} fmx_fwd_out_t;

int fmx_version(void);
const char *fmx_last_error_string(void);

/* Process-wide tuning switches (mutable global state: see Conventions); returns the previous value (>= 0) or a negative
 * status for an unknown name.  Every switch changes HOW the same result is computed; results are identical bits.
 *   "inline_fixup" (default 1)  1: runs that cross 64-occurrence tiles are finished inside k_fm_update by an in-launch
 *                                hand-off; 0: by a second launch (k_fm_fixup).
 *   "sort_ahead"   (default 16) most batches sorted per side-stream launch in fmx_fm_stream / fmx_deepfm_stream (1..16).
 *   "sort_chunked" (default 1)  0: one workgroup per field at every width; 1: k_sort_chunk + k_sort_merge (1,024-composite
 *                                chunks spread over the chip, stable rank merge) from 8,192 composites per field on; 2: from 2,048 on.
 *   "mlp_chain"    (default 1)  0: fmx_mlp_section as separate GEMM launches instead of k_mlp_chain (forward + loss + dgrad chain
 *                                in one launch); same results up to summation order.
 *   "online_persistent" (default 1)  0: fmx_online_run_mlp as per-sample launches instead of one workgroup walking the stream. */
int fmx_set_option(const char *name, int value);

/* Smallest sort width for a batch: max(64, next power of two >= B); and log2 of it. */
int fmx_sorted_width(int B);
int fmx_sorted_bbits(int B);

/* Bytes of caller-owned device workspace a step of batch size B needs (16-byte aligned).  Layout:
 *   (F below: the number of SORT fields of the table, = n_fields unless large fields are split)
 *   sorted  uint32 [32][F, Bp]       occurrence lists: (local index << bbits) | sample, padded with 0xFFFFFFFF
 *                                    (a ring of 32: fmx_fm_stream sorts up to 16 batches ahead; single steps use the first)
 *   runs    uint32 [16][F, Bp]       Bp >= 2048 only: the chunk-sorted intermediate of the wide sort
 *   meta    int32  [F, Bp/64, 2]     per 64-entry tile: does a run come in from / go out to the neighbouring tile; with the
 *                                    in-launch hand-off word 0 is (launch sequence << 4 | states) and is polled by later tiles
 *   parts   float  [F, Bp/64, 2, 2*kp+4]  partial sums of the runs that cross a tile boundary
 * Every entry point that takes a workspace also takes its size in bytes and returns FMX_ERR_SHAPE when that is less than
 * fmx_workspace_bytes(table, B) NOW: the size depends on the table's sort fields, and a table whose fields were split for a
 * larger batch needs a larger workspace at every batch size.
 * The workspace must be ZERO-FILLED once before its first use (the hand-off's flag words are compared with a
 * non-zero launch sequence number; never-written words must not match one by accident).
 * Negative on a bad table. */
int64_t fmx_workspace_bytes(const fmx_table_t *table, int32_t B);

/* Gather + bi-interaction forward.
 * Replaces: first_order / second_order / forward_fm (reference deepfm_adam.py:46-77, fm_adam.py:35-53),
 * i.e. 2 x 39 nn.Embedding gathers, the two 39-term Python sums and the bi-interaction, plus (loss_kind != NONE)
 * the BCE-with-logits loss and its derivative (fm_adam.py:61,66 / :76,80).
 *   idx  [B, F] int32, per-field LOCAL indices exactly as the reference passes them (bit-exact; range-checked)
 *   xv   [B, F] fp32 feature values or null (== 1.0, the Criteo case)
 *   y    [B] fp32 labels or null (required when loss_kind != FMX_LOSS_NONE)
 *   inv_b  1 / (global batch size) folded into dz
 */
int fmx_fm_forward(const fmx_table_t *table, const fmx_hyper_t *hyper, const int32_t *idx, const float *xv,
                   const float *y, int32_t B, int32_t loss_kind, float inv_b, const fmx_fwd_out_t *out,
                   fmx_stream_t stream);

/* ---- the forward pass split over field owners (the model-parallel multi-GPU mode; fmx/owner.py, fmx/plan.py) ----
 * fmx_fm_forward sums a sample's rows in a fixed tree: lane group `slot` of SLOTS = 64 / (kp / 4) adds the fields slot,
 * SLOTS + slot, ... in order, then a butterfly over the lane groups.  The tree is cut into n_blocks BLOCKS of SL = SLOTS /
 * n_blocks consecutive lane groups (n_blocks a power of two <= SLOTS); an owner holds n_local_blocks of them as a table of its
 * own: with NP = n_fields / (n_local_blocks SL) passes, local field (lb NP + p) SL + s sits at position p SLOTS + (first + lb) SL + s
 * of the whole tree (fields are pieces of index columns: fmx_table_t.field_cols / field_base; empty fields fill the holes).
 * fmx_fm_forward_partial evaluates those sub-trees for every sample of the GLOBAL batch: parts_out [B / group][n_local_blocks][group,
 * 2 kp + 4] = per block and sample (S_part[kp], sum e*e part[kp], first-order part, 0, 0, 0); group (0: B) = the samples one rank
 * holds the labels of: their records lie together, block after block -- one contiguous message per destination.  The records of a sample meet on the rank
 * that holds its label; fmx_fm_forward_finish adds the n_blocks records in the order of the remaining butterfly levels --
 * block pairs, pairs of pairs, ... -- and applies fmx_fm_forward's epilogue (bias, loss, dlogit).  The outputs are
 * bit-identical to fmx_fm_forward on one device whose table has the same fields at the same positions.
 *   idx [B, n_cols] int32 (all columns; the owner's fields pick theirs), xv likewise or null; error as in fmx_fwd_out_t
 *   parts [n_blocks][B, 2 kp + 4], block r's records owner_stride floats after block r-1's; bias [1] or [2] as in fmx_table_t
 * Replaces: the same reference sites as fmx_fm_forward (deepfm_adam.py:46-77, fm_adam.py:35-53, :61,66 / :76,80). */
int fmx_fm_forward_partial(const fmx_table_t *table, const int32_t *idx, const float *xv, int32_t B, int32_t n_blocks,
                           int32_t n_local_blocks, int32_t group, float *parts_out, int32_t *error, fmx_stream_t stream);
int fmx_fm_forward_finish(const fmx_hyper_t *hyper, const float *bias, int32_t layout, int32_t kp, const float *parts,
                          int64_t owner_stride, int32_t n_owners, const float *y, int32_t B, int32_t loss_kind, float inv_b,
                          const fmx_fwd_out_t *out, fmx_stream_t stream);

/* ---- the field-owner step as ONE call per step, with the library's own RCCL communicator (one process per GPU) ----
 * What fmx/owner.py's FieldOwnerFM does with torch.distributed collectives between three ctypes calls -- 70 us of host time per
 * step -- as two entry points: fmx_owner_prefetch (weights-free, ahead of time, on the communicator's own stream: all-gather of
 * the ranks' index batches, occurrence sort of the owned pieces over the global batch) and fmx_owner_step (partial forward ->
 * all-to-all of the per-block records -> finish -> all-gather of (S, dlogit, loss) -> update of the owned rows), every launch
 * and both exchanges issued from C on `stream`.  RCCL is loaded at run time (librccl.so.1); with one rank nothing is exchanged
 * (flags bit 0 forces the calls: the RCCL path on a one-GPU box) and RCCL is not needed.
 *   fmx_comm_unique_id   rank 0 fills 2 x FMX_COMM_ID_BYTES bytes (two ncclUniqueId: the step's communicator and the prefetch
 *                        stream's -- operations of ONE RCCL communicator are serialised in issue order whatever their stream);
 *                        the host side broadcasts them (torch.distributed) and every rank calls fmx_comm_create
 *   block_count [world]  tree blocks per rank (fmx.plan.OwnerPlan.block_count; null: one each)
 * A slot (0 .. FMX_COMM_SLOTS-1) names one batch in flight: fmx_owner_prefetch(slot) orders itself behind `stream` as it is at
 * the call and behind the last fmx_owner_step that used the slot; fmx_owner_step(slot) waits for that prefetch.  The caller owns
 * every buffer: idx_all [world B, n_cols] and the workspace of the slot, parts_send [world][blocks of this rank][B, 2 kp + 4],
 * parts_recv [n_blocks][B, 2 kp + 4], rec_local [B, kp + 4], rec_all [world B, kp + 4] (with one rank and no forced collectives
 * parts_recv may be parts_send and rec_all rec_local, and nothing is copied: fmx_owner_prefetch sorts from idx_local, idx_all may be
 * null there, and the caller hands the same idx_local to fmx_owner_step as idx_all).  Same results, bit for bit, as the separate calls.
 * Replaces: reference fm_adam.py:56-69 (update_embedding: forward_fm, loss, backward, optimizer) on a batch sharded over ranks. */
#define FMX_COMM_ID_BYTES 128
#define FMX_COMM_MAX_WORLD 16
typedef struct fmx_comm fmx_comm_t;
typedef struct fmx_owner_bufs {
  float *parts_send, *parts_recv, *rec_local, *rec_all;
} fmx_owner_bufs_t;
int fmx_comm_unique_id(void *ids_out);
int fmx_comm_create(const void *ids, int32_t rank, int32_t world, const int32_t *block_count, int32_t flags, fmx_comm_t **out);
int fmx_comm_destroy(fmx_comm_t *comm);
int fmx_owner_prefetch(fmx_comm_t *comm, const fmx_table_t *table, const int32_t *idx_local, int32_t B, int32_t slot, int32_t *idx_all,
                       void *workspace, int64_t workspace_bytes, int32_t *error, fmx_stream_t stream);
int fmx_owner_step(fmx_comm_t *comm, const fmx_table_t *table, const fmx_hyper_t *hyper, int32_t rule, int32_t loss_kind,
                   const int32_t *idx_all, const float *y_local, int32_t B, int32_t slot, void *workspace, int64_t workspace_bytes,
                   const fmx_owner_bufs_t *bufs, float *loss_out, int32_t *error, fmx_stream_t stream);

/* Occurrence lists: for every field, the batch's (local index, sample) pairs sorted by index then sample.
 * Replaces: the duplicate-row summation embedding_dense_backward performs inside loss.backward()
 * (reference fm_adam.py:67,81) -- sorting is what makes "reduce per unique row, then
 * update once" deterministic.
 *   workspace: fmx_workspace_bytes(table, B) bytes; the lists land at its start as uint32 [F, Bp],
 *   entry = (local index << bbits) | sample, padded with 0xFFFFFFFF;
 *   Bp = fmx_sorted_width(B), bbits = fmx_sorted_bbits(B); requires (largest sort field - 1) < (0xFFFFFFFF >> bbits)
 *   and Bp <= 32768 (a field's composites are merged in one workgroup's LDS).
 */
int fmx_sort_occurrences(const fmx_table_t *table, const int32_t *idx, int32_t B, void *workspace, int64_t workspace_bytes, int32_t *error,
                         fmx_stream_t stream);

/* Row-reduced backward + fused per-row update.
 * Replaces: loss.backward() into 78 dense table gradients + optimizer.step() over all parameters
 * (reference fm_adam.py:67-68 / :81-82).  For every unique row of the batch:
 *     G[b,d]  = dz_bi[b] + gbi[b,d]                       (either term may be absent)
 *     dV[row] = sum_b x (S_b - x V_row) * G[b,:]          dw[row] = sum_b x dz_first[b]
 * summed in sample order, then ONE application of `rule` per coordinate.  The bias gets sum_b dz_first[b].
 *   workspace the one fmx_sort_occurrences filled for the same idx
 *   S         [B, kp] from fmx_fm_forward
 *   dz_first  [B] coefficient of the first-order weights and the bias
 *   dz_bi     [B] or null: scalar coefficient on every bi component (the FM term sum_d bi_d)
 *   gbi       [B, kp] or null: dL/dbi from a network on top of bi (DeepFM / NFM)
 *   loss_b    [B] or null with loss_out [1] or null: loss_out = inv_b * sum_b loss_b (deterministic order)
 *   sample_ld 0, or the record stride of fmx_fwd_out_t.sample_ld: applies to S, dz_first, dz_bi, loss_b and gbi (fields of one
 *             per-sample record, which a data-parallel caller all-gathers with ONE collective)
 */
int fmx_fm_update(const fmx_table_t *table, const fmx_hyper_t *hyper, int32_t rule, void *workspace, int64_t workspace_bytes,
                  const float *xv, const float *S, const float *dz_first, const float *dz_bi, const float *gbi,
                  int32_t B, int32_t sample_ld, const float *loss_b, float inv_b, float *loss_out, fmx_stream_t stream);

/* One pure-FM mini-batch step = sort + forward(+loss) + update on one stream.
 * Replaces: FMAdam.update_embedding / FMAdam.fit (reference fm_adam.py:56-82) and every class's
 * update_embedding (deepfm_adam.py:91-104 etc.), which all train on forward_fm only.
 * workspace: fmx_workspace_bytes(table, B) bytes; fwd->S, fwd->loss, fwd->dz must be non-null. */
int fmx_fm_step(const fmx_table_t *table, const fmx_hyper_t *hyper, int32_t rule, int32_t loss_kind,
                const int32_t *idx, const float *xv, const float *y, int32_t B, float inv_b, void *workspace, int64_t workspace_bytes,
                const fmx_fwd_out_t *fwd, float *loss_out, fmx_stream_t stream);

/* The online loop over a device-resident stream of mini-batches: step s uses batch (s mod n_pool).
 * Replaces: the driver loops reference main_experiment.py:92-105 (pre-training) and fm_adam.py:97-99
 * (run_experiment), batched.  idx_pool [n_pool, B, F], y_pool [n_pool, B]; loss_out [n_steps] or null.
 * kernel_ms (HOST pointer, [4]) or null: the measuring mode.  Everything runs on `stream` in groups of up to 8 steps: ONE
 * sort launch for the group's batches (as in production), the group's forwards back to back, then its updates back to
 * back, each block between two HIP events, every launch on a different batch of the pool (own sorted list, own
 * S / dz / loss in a temporary buffer) so that rows come from HBM / MALL as in production.  The forwards of a group all
 * read the table before the group's updates: the table is NOT the production run's.  After a stream synchronise
 * kernel_ms[0..2] = n_steps x the average per-launch milliseconds of {sort (up to 8 batches per launch), forward,
 * update (both launches of it when inline_fixup = 0)}, each with the cost of an empty event pair -- [3], same scaling --
 * subtracted. */
int fmx_fm_stream(const fmx_table_t *table, const fmx_hyper_t *hyper, int32_t rule, int32_t loss_kind,
                  const int32_t *idx_pool, const float *y_pool, int32_t n_pool, int32_t B, float inv_b,
                  int32_t n_steps, void *workspace, int64_t workspace_bytes, const fmx_fwd_out_t *fwd, float *loss_out, float *kernel_ms,
                  fmx_stream_t stream);

/* The reference's online protocol for the pure-FM class on a device-resident stream of N samples: for every sample,
 * predict (pred_out[i] = sigmoid(logit) > 0.5 with the weights BEFORE the sample's update), then one fit step on that
 * sample alone (B = 1, inv_b = 1) under `rule` and `loss_kind`.  One wavefront walks the stream (the steps are sequential by
 * definition); the table and bias end bit-identical to N calls of fmx_fm_step with B = 1.  loss_out [N] may be null.
 * Needs n_fields <= 4 * (64 / (kp / 4)) (64 fields at kp = 16), else FMX_ERR_UNSUPPORTED.
 * Replaces: FMAdam.run_experiment's loop body (reference fm_adam.py:97-99: predict at :84-88, fit at :71-82). */
int fmx_fm_online_run(const fmx_table_t *table, const fmx_hyper_t *hyper, int32_t rule, int32_t loss_kind,
                      const int32_t *idx, const float *xv, const float *y, int32_t N, uint8_t *pred_out, float *loss_out,
                      int32_t *error, fmx_stream_t stream);

/* ---- the small relu MLP on top of the bi-interaction vector (online steps of DeepFM / NFM and the ONN classes) ----
 * params: per layer W [out, in] row-major then b [out]; layer 0 maps k -> hidden, the others hidden -> hidden; the network's
 * contribution to the logit is the sum of the last activation (reference deepfm_adam.py:82-88: there is no output layer).
 * One workgroup per call; limits B <= 16, k <= 64 (63 for fit), hidden <= 64, layers <= 8, else FMX_ERR_UNSUPPORTED (larger
 * shapes stay on the caller's PyTorch path). */
typedef struct fmx_mlp {
  float *params;
  int32_t n_layers, k, hidden, reserved;
} fmx_mlp_t;

/* out [B] = base + sum_j x_L[j]  (may be null) ; layers_out [L, B] = sigmoid(base + sum_j x_l[j]) (may be null).
 * Replaces: the MLP part of forward() (reference deepfm_adam.py:79-89, deepfm_onn.py:88-102). */
int fmx_mlp_forward(const fmx_mlp_t *mlp, const float *bi, int32_t kp, const float *base, int32_t B, float *out,
                    float *layers_out, fmx_stream_t stream);

/* Forward, loss (fmx_loss on base + MLP), backward and the update of every hidden layer under `rule`
 * (FMX_RULE_SIGNADAM = the reference's fresh Adam, or FMX_RULE_SGD); emits what fmx_fm_update needs for the tables:
 * dz_out [B] = dL/dlogit and gbi_out [B, kp] = dL/dbi through the MLP.  loss_out [1] = mean loss (may be null).
 * Replaces: DeepFMAdam.fit / NFMAdam.fit minus the table part (reference deepfm_adam.py:106-119, nfm_adam.py:105-118). */
int fmx_mlp_fit(const fmx_mlp_t *mlp, const fmx_hyper_t *hyper, int32_t rule, int32_t loss_kind, const float *bi, int32_t kp,
                const float *base, const float *y, int32_t B, float inv_b, float *dz_out, float *gbi_out, float *loss_out,
                fmx_stream_t stream);

/* Hedge backprop (reference deepfm_onn.py:109-154): per-layer BCELoss(sigmoid(base + sum x_l), y), hidden layers updated by
 * lr * sum_{i >= j} alpha_i dloss_i/dlayer_j, then alpha_i <- max(alpha_i * hedge_b^loss_i, hedge_s / L) normalised.
 * alpha [L] is updated in place; losses_out [L] may be null.  The tables are not touched (as in the reference). */
int fmx_mlp_hedge_fit(const fmx_mlp_t *mlp, float lr, float hedge_b, float hedge_s, float *alpha, const float *bi, int32_t kp,
                      const float *base, const float *y, int32_t B, float *losses_out, fmx_stream_t stream);

/* The reference's online protocol for the classes with an MLP on a device-resident stream of N samples: per sample the
 * forward (pred_out[i] = what forward() returns: the logit for the Adam classes, sigmoid of the last layer's logit for
 * the ONN classes), then fit on that sample -- hedge = 0: fmx_mlp_fit + the table update (DeepFMAdam / NFMAdam.fit),
 * hedge = 1: fmx_mlp_hedge_fit (the ONN classes: hidden layers and alpha only).  fm_term: the FM logit is part of the
 * network's input logit (DeepFM) or only the first-order sum and the bias (NFM).  One workgroup walks the stream with the
 * network's parameters in LDS (k_online_mlp) when they are at most 8,192 floats, the fields fit one wavefront and the tables
 * are not FTRL tables under a fit step; otherwise the launches of all samples are queued without any host synchronisation
 * (2 per sample with Hedge, 4 otherwise).  Either way the parameters end bit-identical to per-sample calls.  workspace: fmx_workspace_bytes(table, 1);
 * fwd: S, bi, sfirst, logit of at least one sample; scratch: >= kp + 8 floats, 16-byte aligned.
 * Replaces: the loop body of run_experiment (reference deepfm_adam.py:128-130, deepfm_onn.py:178-180, ...). */
int fmx_online_run_mlp(const fmx_table_t *table, const fmx_hyper_t *hyper, int32_t rule, int32_t loss_kind,
                       const fmx_mlp_t *mlp, int32_t hedge, int32_t fm_term, float hedge_b, float hedge_s, float *alpha,
                       const int32_t *idx, const float *xv, const float *y, int32_t N, void *workspace, int64_t workspace_bytes,
                       const fmx_fwd_out_t *fwd, float *scratch, float *pred_out, fmx_stream_t stream);

/* The same network at mini-batch sizes (BASELINE configs[3]: 3 x 256, B = 4096), any B / hidden / k, up to 8 layers:
 * forward, loss on (base + sum_j x_L[j]), backward -- fp32 MFMA GEMMs (v_mfma_f32_32x32x2_f32: exact f32 products and
 * accumulation), deterministic (split-K partials summed in a fixed order, no atomics).
 *   bi [B, ld_bi] (first k columns used), base [B], y [B];  logit_out [B] may be null
 *   dz_out [B] = dL/dlogit (inv_b folded in), gbi_out [B, ld_gbi] = dL/dbi through the MLP (columns k..ld_gbi-1 zeroed)
 *   grads: flat, the layout of mlp->params (W_l [out, in] then b_l [out] per layer); lr_apply != 0 also applies
 *   params -= lr_apply * grads in the same pass (single-rank SGD); loss_out [1] = inv_b * sum of the per-sample losses
 *   workspace: fmx_mlp_section_workspace_bytes(mlp, B) bytes, 16-byte aligned (activations, dH ping-pong, split partials)
 * Replaces: the MLP part of DeepFMAdam.fit / NFMAdam.fit at batch sizes the one-workgroup kernel does not take
 * (reference deepfm_adam.py:79-89,106-119; nfm_adam.py:78-88,105-118), i.e. nn.Linear + relu + autograd. */
/* Forward only at mini-batch sizes (predict / forward() of the MLP classes on whole batches, e.g. the accuracy print of
 * the pre-training loop, reference main_experiment.py:98): logit_out [B] = base + sum_j x_L[j] (may be null),
 * layers_out [L, B] = sigmoid(base + sum_j x_l[j]) per layer (may be null; what the ONN classes' forward() returns).
 * Same GEMMs and workspace as fmx_mlp_section. */
int fmx_mlp_forward_batch(const fmx_mlp_t *mlp, const float *bi, int32_t ld_bi, const float *base, int32_t B, void *workspace,
                          float *logit_out, float *layers_out, fmx_stream_t stream);
int64_t fmx_mlp_section_workspace_bytes(const fmx_mlp_t *mlp, int32_t B);
int fmx_mlp_section(const fmx_mlp_t *mlp, int32_t loss_kind, const float *bi, int32_t ld_bi, const float *base,
                    const float *y, int32_t B, float inv_b, void *workspace, float *logit_out, float *dz_out,
                    float *gbi_out, int32_t ld_gbi, float *grads, float lr_apply, float *loss_out, fmx_stream_t stream);

/* The mini-batch DeepFM loop over a device-resident stream of mini-batches (BASELINE configs[3]): step s uses batch (s mod n_pool);
 * per step the forward of the tables (S, bi, FM logit), fmx_mlp_section on bi with the FM logit as base (the SGD of the MLP applied in
 * its reduction: lr_mlp), then the row-reduced table update with dz_first = dz_bi = dL/dlogit and gbi = dL/dbi -- the launches of a
 * step issued back to back from one call, the occurrence sorts in groups on the library's side stream as in fmx_fm_stream.
 * idx_pool [n_pool, B, F], y_pool [n_pool, B]; fwd: S, bi, logit (dense, sample_ld = 0); dz [B], gbi [B, kp], grads (layout of
 * mlp->params) are scratch the call fills; loss_out [n_steps] or null; workspace: fmx_workspace_bytes(table, B), mlp_workspace:
 * fmx_mlp_section_workspace_bytes(mlp, B).  The result is the one of calling fmx_fm_forward, fmx_mlp_section, fmx_sort_occurrences
 * and fmx_fm_update per step.
 * fm_term: 1 = DeepFM (the FM logit is the network's base and dz also drives the FM term of the rows' gradient); 0 = NFM (base =
 * first-order sum + bias, written into fwd->logit's buffer; fwd->sfirst required; tables in the weights layout; the rows' gradient
 * comes through dL/dbi only -- reference nfm_adam.py:78-88,105-118).
 * Replaces: the mini-batch driver loop over DeepFMAdam.fit / NFMAdam.fit (reference main_experiment.py:92-105 with
 * deepfm_adam.py:106-119, nfm_adam.py:105-118). */
int fmx_deepfm_stream(const fmx_table_t *table, const fmx_hyper_t *hyper, int32_t rule, const fmx_mlp_t *mlp, int32_t loss_kind, int32_t fm_term,
                      const int32_t *idx_pool, const float *y_pool, int32_t n_pool, int32_t B, float inv_b, int32_t n_steps,
                      void *workspace, int64_t workspace_bytes, void *mlp_workspace, const fmx_fwd_out_t *fwd, float *dz, float *gbi,
                      float *grads, float lr_mlp, float *loss_out, fmx_stream_t stream);

/* Hedge backprop at mini-batch sizes (the ONN classes' fit() beyond 16 samples; reference deepfm_onn.py:109-154): per
 * layer BCELoss(sigmoid(base + sum_j x_l[j]), y), hidden layers updated by lr * sum_{i >= j} alpha_i dloss_i/dlayer_j (one
 * backward pass on the same GEMMs; `grads` receives that gradient), then alpha_i <- max(alpha_i * hedge_b^loss_i,
 * hedge_s / L) normalised, in place.  losses_out [L] may be null.  The tables are not touched, as in the reference. */
int fmx_mlp_hedge_section(const fmx_mlp_t *mlp, float lr, float hedge_b, float hedge_s, float *alpha, const float *bi,
                          int32_t ld_bi, const float *base, const float *y, int32_t B, void *workspace, float *grads,
                          float *losses_out, fmx_stream_t stream);

/* The sketched-FTRL family on a device-resident stream (hot path B's sketch classes): for every
 * sample predict y_hat = ||BP^T x||^2 - ||BN^T x||^2 (+ w^T x), then append sqrt(eta |s|) x to the sketch the gradient sign s
 * picks and shrink a full sketch (frequent directions).  fp64 like the reference, strictly sequential: ONE wavefront walks the
 * stream with both sketches in LDS; the shrink is a Jacobi eigen-decomposition of the d x d Gram matrix (same B B^T as the
 * reference's SVD of B^T B; the columns of B may differ by sign).
 *   X [N, D], y [N] fp64; the first d features enter the sketches (d = D: SFTRL_CCFM; d = D - 1 with w, g_w [D]: SFTRL_Vanila)
 *   BP, BN [d, 2 m] row-major and counts [2] (columns in use) are read and written back; task 0 = cls (+-1 predictions), 1 = reg
 *   status [2]: status[0] = 1 when the prediction of sample status[1] was NaN (the run stops there)
 * Limits: d <= 32, 2 m <= 128, D <= 64, else FMX_ERR_UNSUPPORTED (the caller's host path).
 * Replaces: SFTRL_CCFM.online_learning / _GFD (reference models/models_online/SFTRL_CCFM.py:30-121), SFTRL_Vanila.py:30-130. */
int fmx_sftrl_run(const double *X, const double *y, int32_t N, int32_t D, int32_t d, int32_t m, double eta, double thres,
                  int32_t task, double *BP, double *BN, int32_t *counts, double *w, double *g_w, double *pred_out,
                  int32_t *status, fmx_stream_t stream);

/* A GRID of sketched-FTRL settings over the same stream in one launch: workgroup s runs (ms[s], etas[s]) -- one wavefront
 * per setting is latency-bound, 256 CUs run 256 settings in the time of one (the reference's notebooks try (eta, m) pairs one
 * run_experiment at a time).  ms [S] int32 (each <= m_max), etas [S] fp64 are device arrays.  Per setting s:
 *   BP, BN  [S][d * 2 * m_max]: setting s's sketch as [d, 2 ms[s]] row-major at the start of its slot;  counts [S, 2];
 *   w, g_w  [S, D] or both null;  pred_out [S, N];  status [S, 2]  -- everything else as fmx_sftrl_run, same limits;
 *   status[s][0] = 2: ms[s] = status[s][1] lies outside [1, m_max] (the launch is sized for m_max): setting s was not run.
 * Every setting's result is bit-identical to its own fmx_sftrl_run. */
int fmx_sftrl_grid(const double *X, const double *y, int32_t N, int32_t D, int32_t d, int32_t n_settings, const int32_t *ms,
                   const double *etas, int32_t m_max, double thres, int32_t task, double *BP, double *BN, int32_t *counts, double *w,
                   double *g_w, double *pred_out, int32_t *status, fmx_stream_t stream);

/* Streaming read of `bytes` (multiple of 16) with 16-byte loads; sink [1] receives a checksum so the loads stay
 * live.  Used by bench.py to measure the HBM-read ceiling on the same GPU in the same run. */
int fmx_stream_read(const void *buf, int64_t bytes, float *sink, fmx_stream_t stream);

/* Random-row read probe (measurement aid, SURVEY.md section 8(d) "informational gather ceilings"): n_rows_read rows of row_bytes
 * (64 or 128) at hashed positions of buf, 16 bytes per lane -- the forward gather's access pattern alone. */
int fmx_gather_read(const void *buf, int64_t bytes, int32_t row_bytes, int64_t n_rows_read, uint32_t seed, float *sink, fmx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FMX_H */
