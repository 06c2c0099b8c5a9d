/* fm_oracle.c -- CPU restatement (plain C; one thread, plus an OpenMP form of the same step) of the flat-table FM mini-batch step.
 * TEST INFRASTRUCTURE ONLY: built by oracle/Makefile into oracle/_build/liboracle.so and loaded by
 * oracle/c_oracle.py; used by tests/ to cross-check the numpy oracle and by bench.py's cpu_baseline leg.
 * Never linked into, loaded by or shipped with the product library (libfmx.so).
 *
 * It restates the same reference sites as oracle/fm_oracle.py (paths relative to the reference repository):
 *   forward      models/models_online_deep/fm_adam.py:35-53  (fields summed in order f = 0..F-1)
 *   loss / dz    fm_adam.py:61,66 (BCEwl(z)) and :76,80 (BCEwl(sigmoid(z)))
 *   backward     loss.backward() -> duplicate rows of the batch summed before the update (fm_adam.py:67)
 *   update       fresh Adam == p -= lr g / (|g| + eps) (fm_adam.py:60,68); SGD and FTRL-proximal are not in the
 *                reference (McMahan et al. 2013, Algorithm 1): parity unpinned, as in fm_oracle.py.
 * Pinned through tests/test_oracle_golden.py::test_c_oracle_matches_numpy_oracle (numpy oracle is pinned by the
 * golden fixtures generated from the reference).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { RULE_SIGNADAM = 0, RULE_SGD = 1, RULE_FTRL = 2 };
enum { LOSS_LOGITS = 1, LOSS_SIGMOID = 2 };

typedef struct {
  float lr, eps, alpha, beta, l1, l2;
} fmo_hyper;

static float sigmoidf_(float z) { return 1.0f / (1.0f + expf(-z)); }
static float bcewl(float z, float y) { return (1.0f - y) * z + log1pf(expf(-fabsf(z))) + fmaxf(-z, 0.0f); }

static float ftrl_w(float z, float n, const fmo_hyper *h) {
  if (fabsf(z) <= h->l1) return 0.0f;
  float sgn = z > 0 ? 1.0f : -1.0f;
  return -(z - sgn * h->l1) / ((h->beta + sqrtf(n)) / h->alpha + h->l2);
}
static void ftrl_upd(float *z, float *n, float g, const fmo_hyper *h) {
  float w = ftrl_w(*z, *n, h);
  float n2 = *n + g * g;
  float sigma = (sqrtf(n2) - sqrtf(*n)) / h->alpha;
  *z = *z + g - sigma * w;
  *n = n2;
}
static float apply_rule(int rule, float p, float g, const fmo_hyper *h) {
  if (rule == RULE_SIGNADAM) return p - h->lr * g / (fabsf(g) + h->eps);
  return p - h->lr * g;
}

typedef struct {
  int64_t row;
  int32_t b, f;
} occ_t;

static int occ_cmp(const void *pa, const void *pb) {
  const occ_t *a = (const occ_t *)pa, *b = (const occ_t *)pb;
  if (a->row != b->row) return a->row < b->row ? -1 : 1;
  return a->b < b->b ? -1 : (a->b > b->b);
}

/* One pure-FM mini-batch step on a flat table.
 *   rule SIGNADAM / SGD: P0 = V [R,k], P1 = w [R], bias[0]
 *   rule FTRL          : P0 = zV, P2 = nV [R,k]; P1 = zw, P3 = nw [R]; bias[0] = z, bias[1] = n (weights derived)
 *   rows [B,F] global row ids; x [B,F] or NULL (== 1); y [B]
 * Returns the mean loss; logit_out [B] and dz_out [B] may be NULL. */
double fmo_fm_step(int rule, int loss_kind, float *P0, float *P1, float *P2, float *P3, float *bias, int64_t R, int k,
                   const int64_t *rows, const float *x, const float *y, int B, int F, const fmo_hyper *h, float inv_b,
                   float *logit_out, float *dz_out) {
  (void)R;
  float *S = (float *)malloc((size_t)B * k * sizeof(float));
  float *dz = (float *)malloc((size_t)B * sizeof(float));
  float *v = (float *)malloc((size_t)k * sizeof(float));
  occ_t *occ = (occ_t *)malloc((size_t)B * F * sizeof(occ_t));
  float *gV = (float *)malloc((size_t)k * sizeof(float));
  float bias_w = rule == RULE_FTRL ? ftrl_w(bias[0], bias[1], h) : bias[0];
  float loss_sum = 0.0f, db = 0.0f;
  /* ---- forward + loss ---- */
  for (int b = 0; b < B; ++b) {
    float *Sb = S + (size_t)b * k;
    float sfirst = 0.0f, sbi = 0.0f;
    float *SS = v; /* sum of squares, reuse */
    for (int d = 0; d < k; ++d) { Sb[d] = 0.0f; SS[d] = 0.0f; }
    for (int f = 0; f < F; ++f) {
      int64_t r = rows[(size_t)b * F + f];
      float xv = x ? x[(size_t)b * F + f] : 1.0f;
      for (int d = 0; d < k; ++d) {
        float w = rule == RULE_FTRL ? ftrl_w(P0[r * k + d], P2[r * k + d], h) : P0[r * k + d];
        float e = w * xv;
        Sb[d] += e;
        SS[d] += e * e;
      }
      float w1 = rule == RULE_FTRL ? ftrl_w(P1[r], P3[r], h) : P1[r];
      sfirst += w1 * xv;
      occ[(size_t)b * F + f].row = r;
      occ[(size_t)b * F + f].b = b;
      occ[(size_t)b * F + f].f = f;
    }
    for (int d = 0; d < k; ++d) sbi += (Sb[d] * Sb[d] - SS[d]) * 0.5f;
    float z = sfirst + sbi + bias_w;
    float yy = y[b], l, g;
    if (loss_kind == LOSS_LOGITS) {
      l = bcewl(z, yy);
      g = (sigmoidf_(z) - yy) * inv_b;
    } else {
      float p = sigmoidf_(z);
      l = bcewl(p, yy);
      g = (sigmoidf_(p) - yy) * p * (1.0f - p) * inv_b;
    }
    loss_sum += l;
    dz[b] = g;
    db += g;
    if (logit_out) logit_out[b] = z;
    if (dz_out) dz_out[b] = g;
  }
  /* ---- duplicates summed per unique row (sample order), one update per row ---- */
  qsort(occ, (size_t)B * F, sizeof(occ_t), occ_cmp);
  size_t n = (size_t)B * F, i = 0;
  while (i < n) {
    int64_t r = occ[i].row;
    float gw = 0.0f;
    for (int d = 0; d < k; ++d) {
      gV[d] = 0.0f;
      v[d] = rule == RULE_FTRL ? ftrl_w(P0[r * k + d], P2[r * k + d], h) : P0[r * k + d];
    }
    size_t j = i;
    for (; j < n && occ[j].row == r; ++j) {
      int b = occ[j].b;
      float xv = x ? x[(size_t)b * F + occ[j].f] : 1.0f;
      const float *Sb = S + (size_t)b * k;
      for (int d = 0; d < k; ++d) gV[d] += xv * (Sb[d] - v[d] * xv) * dz[b];
      gw += xv * dz[b];
    }
    if (rule == RULE_FTRL) {
      for (int d = 0; d < k; ++d) ftrl_upd(&P0[r * k + d], &P2[r * k + d], gV[d], h);
      ftrl_upd(&P1[r], &P3[r], gw, h);
    } else {
      for (int d = 0; d < k; ++d) P0[r * k + d] = apply_rule(rule, P0[r * k + d], gV[d], h);
      P1[r] = apply_rule(rule, P1[r], gw, h);
    }
    i = j;
  }
  if (rule == RULE_FTRL) ftrl_upd(&bias[0], &bias[1], db, h);
  else bias[0] = apply_rule(rule, bias[0], db, h);
  free(S); free(dz); free(v); free(occ); free(gV);
  return (double)(loss_sum * inv_b);
}

/* The same step on n_threads host threads (OpenMP) -- bench.py's multi-core cpu_baseline.  Same arithmetic in the same
 * order as fmo_fm_step, hence the same bits (tests/test_oracle_golden.py): the forward runs over samples in parallel and
 * the scalars are then summed in sample order; the rows of different columns of `rows` are assumed disjoint (they are
 * field-partitioned global ids), so each field's occurrences are sorted and applied by one thread. */
double fmo_fm_step_mt(int rule, int loss_kind, float *P0, float *P1, float *P2, float *P3, float *bias, int64_t R, int k,
                      const int64_t *rows, const float *x, const float *y, int B, int F, const fmo_hyper *h, float inv_b,
                      int n_threads) {
  (void)R;
  if (n_threads < 1) n_threads = 1;
  float *S = (float *)malloc((size_t)B * k * sizeof(float));
  float *dz = (float *)malloc((size_t)B * sizeof(float));
  float *ls = (float *)malloc((size_t)B * sizeof(float));
  const float bias_w = rule == RULE_FTRL ? ftrl_w(bias[0], bias[1], h) : bias[0];
#pragma omp parallel for schedule(static) num_threads(n_threads)
  for (int b = 0; b < B; ++b) {
    float *Sb = S + (size_t)b * k;
    float SS[256];
    float sfirst = 0.0f, sbi = 0.0f;
    for (int d = 0; d < k; ++d) { Sb[d] = 0.0f; SS[d] = 0.0f; }
    for (int f = 0; f < F; ++f) {
      int64_t r = rows[(size_t)b * F + f];
      float xv = x ? x[(size_t)b * F + f] : 1.0f;
      for (int d = 0; d < k; ++d) {
        float w = rule == RULE_FTRL ? ftrl_w(P0[r * k + d], P2[r * k + d], h) : P0[r * k + d];
        float e = w * xv;
        Sb[d] += e;
        SS[d] += e * e;
      }
      float w1 = rule == RULE_FTRL ? ftrl_w(P1[r], P3[r], h) : P1[r];
      sfirst += w1 * xv;
    }
    for (int d = 0; d < k; ++d) sbi += (Sb[d] * Sb[d] - SS[d]) * 0.5f;
    float z = sfirst + sbi + bias_w;
    float yy = y[b];
    if (loss_kind == LOSS_LOGITS) {
      ls[b] = bcewl(z, yy);
      dz[b] = (sigmoidf_(z) - yy) * inv_b;
    } else {
      float p = sigmoidf_(z);
      ls[b] = bcewl(p, yy);
      dz[b] = (sigmoidf_(p) - yy) * p * (1.0f - p) * inv_b;
    }
  }
  float loss_sum = 0.0f, db = 0.0f;
  for (int b = 0; b < B; ++b) {
    loss_sum += ls[b];
    db += dz[b];
  }
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
  for (int f = 0; f < F; ++f) {
    occ_t *occ = (occ_t *)malloc((size_t)B * sizeof(occ_t));
    float v[256], gV[256];
    for (int b = 0; b < B; ++b) {
      occ[b].row = rows[(size_t)b * F + f];
      occ[b].b = b;
      occ[b].f = f;
    }
    qsort(occ, (size_t)B, sizeof(occ_t), occ_cmp);
    size_t n = (size_t)B, i = 0;
    while (i < n) {
      int64_t r = occ[i].row;
      float gw = 0.0f;
      for (int d = 0; d < k; ++d) {
        gV[d] = 0.0f;
        v[d] = rule == RULE_FTRL ? ftrl_w(P0[r * k + d], P2[r * k + d], h) : P0[r * k + d];
      }
      size_t j = i;
      for (; j < n && occ[j].row == r; ++j) {
        int b = occ[j].b;
        float xv = x ? x[(size_t)b * F + f] : 1.0f;
        const float *Sb = S + (size_t)b * k;
        for (int d = 0; d < k; ++d) gV[d] += xv * (Sb[d] - v[d] * xv) * dz[b];
        gw += xv * dz[b];
      }
      if (rule == RULE_FTRL) {
        for (int d = 0; d < k; ++d) ftrl_upd(&P0[r * k + d], &P2[r * k + d], gV[d], h);
        ftrl_upd(&P1[r], &P3[r], gw, h);
      } else {
        for (int d = 0; d < k; ++d) P0[r * k + d] = apply_rule(rule, P0[r * k + d], gV[d], h);
        P1[r] = apply_rule(rule, P1[r], gw, h);
      }
      i = j;
    }
    free(occ);
  }
  if (rule == RULE_FTRL) ftrl_upd(&bias[0], &bias[1], db, h);
  else bias[0] = apply_rule(rule, bias[0], db, h);
  free(S); free(dz); free(ls);
  return (double)(loss_sum * inv_b);
}
