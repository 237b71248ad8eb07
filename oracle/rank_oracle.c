/*
 * ORACLE (test infrastructure, never shipped in the product path).
 *
 * Plain-C restatement of the reference's Market1501-protocol CMC / mAP
 * evaluator.  Follows, line for line in meaning (not in text):
 *   - torchreid/metrics/rank.py:103-171   eval_market1501  (the path the
 *     reference actually runs: float64 AP, np.mean over valid queries)
 *   - torchreid/metrics/rank_cylib/rank_cy.pyx:156-243 eval_market1501_cy
 *     (float32 accumulators; selected with use_f32_accum != 0)
 * Tie order: the reference sorts with np.argsort (unstable introsort); this
 * restatement fixes the order to (distance, gallery index) ascending, which
 * is what the HIP evaluator implements too.  Golden vectors are tie-free.
 *
 * Pinned against goldens produced by the imported reference:
 * tests/golden/evaluator_*.npz (tests/golden/gen_evaluator_golden.py).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float d; int64_t idx; } key_t_;

static int cmp_key(const void* a, const void* b) {
  const key_t_* x = (const key_t_*)a; const key_t_* y = (const key_t_*)b;
  if (x->d < y->d) return -1;
  if (x->d > y->d) return 1;
  return (x->idx > y->idx) - (x->idx < y->idx);
}

/* returns number of valid queries (0 => the reference raises AssertionError,
 * rank.py:165); cmc has max_rank_eff = min(max_rank, num_g) entries
 * (rank.py:110-115); all_ap (optional, num_q doubles) gets -1 for skipped
 * queries. */
int64_t ieee_oracle_rank_market1501(const float* distmat, int64_t num_q, int64_t num_g,
                                    const int64_t* q_pids, const int64_t* g_pids,
                                    const int64_t* q_camids, const int64_t* g_camids,
                                    int64_t max_rank, int use_f32_accum,
                                    float* cmc_out, double* map_out, double* all_ap) {
  if (num_g < max_rank) max_rank = num_g;                       /* rank.py:110-115 */
  key_t_* keys = (key_t_*)malloc(sizeof(key_t_) * (size_t)num_g);
  unsigned char* raw = (unsigned char*)malloc((size_t)num_g);
  double* cmc_sum = (double*)calloc((size_t)max_rank, sizeof(double));
  double ap_sum = 0.0; float ap_sum_f = 0.f;
  int64_t num_valid = 0;
  for (int64_t q = 0; q < num_q; ++q) {
    for (int64_t j = 0; j < num_g; ++j) { keys[j].d = distmat[q * num_g + j]; keys[j].idx = j; }
    qsort(keys, (size_t)num_g, sizeof(key_t_), cmp_key);        /* rank.py:117 argsort */
    int64_t nk = 0, any = 0;
    for (int64_t j = 0; j < num_g; ++j) {                        /* rank.py:134-141 */
      int64_t g = keys[j].idx;
      if (g_pids[g] == q_pids[q] && g_camids[g] == q_camids[q]) continue;   /* remove */
      raw[nk] = (unsigned char)(g_pids[g] == q_pids[q]);
      any |= raw[nk]; ++nk;
    }
    if (all_ap) all_ap[q] = -1.0;
    if (!any) continue;                                          /* rank.py:142-144 */
    /* cmc = min(cumsum(raw),1)[:max_rank]   rank.py:145-150 */
    int64_t c = 0;
    for (int64_t r = 0; r < max_rank; ++r) { if (r < nk) c += raw[r]; cmc_sum[r] += (c > 0) ? 1.0 : 0.0; }
    ++num_valid;
    /* AP   rank.py:153-159 */
    if (use_f32_accum) {
      float cum = 0.f, s = 0.f, nrel = 0.f;
      for (int64_t r = 0; r < nk; ++r) { cum += raw[r]; s += (cum / (float)(r + 1.)) * raw[r]; nrel += raw[r]; }
      float ap = s / nrel; ap_sum_f += ap; if (all_ap) all_ap[q] = ap;
    } else {
      double cum = 0, s = 0, nrel = 0;
      for (int64_t r = 0; r < nk; ++r) { cum += raw[r]; if (raw[r]) s += cum / (double)(r + 1); nrel += raw[r]; }
      double ap = s / nrel; ap_sum += ap; if (all_ap) all_ap[q] = ap;
    }
  }
  if (num_valid > 0) {
    for (int64_t r = 0; r < max_rank; ++r) cmc_out[r] = (float)cmc_sum[r] / (float)num_valid;  /* rank.py:167-168 (float32) */
    *map_out = use_f32_accum ? (double)(ap_sum_f / (float)num_valid) : ap_sum / (double)num_valid;
  }
  free(keys); free(raw); free(cmc_sum);
  return num_valid;
}

/* fp32 squared-Euclidean distance matrix, the formula of
 * torchreid/metrics/distance.py:49-64 : |q|^2 + |g|^2 - 2 q.g, k-ordered
 * float accumulation (the reference delegates the sum order to MKL sgemm). */
void ieee_oracle_sqeuclid(const float* q, const float* g, int64_t m, int64_t n, int64_t d, float* out) {
  float* gn = (float*)malloc(sizeof(float) * (size_t)n);
  for (int64_t j = 0; j < n; ++j) { float s = 0; for (int64_t k = 0; k < d; ++k) s += g[j*d+k]*g[j*d+k]; gn[j] = s; }
  for (int64_t i = 0; i < m; ++i) {
    float qn = 0; for (int64_t k = 0; k < d; ++k) qn += q[i*d+k]*q[i*d+k];
    for (int64_t j = 0; j < n; ++j) {
      float dot = 0; for (int64_t k = 0; k < d; ++k) dot += q[i*d+k]*g[j*d+k];
      out[i*n+j] = (qn + gn[j]) - 2.0f * dot;
    }
  }
  free(gn);
}
