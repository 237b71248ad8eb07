// k-reciprocal re-ranking on the device: reference torchreid/utils/rerank.py:31-113, the optional
// post-step of the evaluator (engine/engine.py:402-406).  Dense formulation, like the reference's (N = Q + G rows):
//   A  D[i][j]  = orig[j][i]^2 / max_a orig[a][i]^2            (orig = [[qq, qg], [qg^T, gg]])        :45-48
//   B  rank[i][0..K) = the K = k1+1 nearest j of row i, ordered by (D, j)                               :50
//   C  V[i][e] = exp(-D[i][e]) / sum over e in the expanded k-reciprocal set of i                       :56-82
//   D  Vq[i][:] = mean of V[rank[i][t]][:], t < k2                                                      :84-89
//   E  jac[i][j] = 1 - t/(2-t), t = sum_c min(Vq[i][c], Vq[j][c])  (c ascending, fp32: the order in which the
//      reference's inverted-index loop adds the non-zero terms; zero terms add nothing)                 :91-106
//   F  out[i][g] = jac[i][Q+g]*(1-lambda) + D[i][Q+g]*lambda                                            :108-112
// Index work (B, the set logic of C) is exact; the float steps follow the reference's order of operations and
// differ from numpy only in exp() and in np.sum's pairwise association (<= a few ulp).  Ties in B are broken by
// index (the reference's np.argsort is unstable there).  Integer / HBM-bound work: no matrix cores involved.
#include "common.h"

namespace ieee {

struct OrigView {                 // the (virtual) all-pairs matrix, rerank.py:36-44
  const float *qg, *qq, *gg;
  int Q, G;
  __device__ __forceinline__ float at(int a, int b) const {
    if (a < Q) return b < Q ? qq[(int64_t)a * Q + b] : qg[(int64_t)a * G + (b - Q)];
    return b < Q ? qg[(int64_t)b * G + (a - Q)] : gg[(int64_t)(a - Q) * G + (b - Q)];
  }
};

// A1: colmax[i] = max_a orig[a][i]^2
__global__ __launch_bounds__(256) void rr_colmax_kernel(OrigView o, float* __restrict__ colmax) {
  const int N = o.Q + o.G;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float m = -INFINITY;
  for (int a = 0; a < N; ++a) { const float v = o.at(a, i); m = fmaxf(m, v * v); }
  colmax[i] = m;
}

// A2: D[i][j] = orig[j][i]^2 / colmax[i]   (32x32 LDS transpose so that reads and writes are both coalesced)
__global__ __launch_bounds__(256) void rr_normalise_kernel(OrigView o, const float* __restrict__ colmax,
                                                           float* __restrict__ D) {
  __shared__ float tile[32][33];
  const int N = o.Q + o.G;
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {            // read orig[j0 + r][i0 + tx]
    const int a = j0 + r, b = i0 + tx;
    float v = 0.f;
    if (a < N && b < N) { v = o.at(a, b); v = v * v; }
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {            // write D[i0 + r][j0 + tx]
    const int i = i0 + r, j = j0 + tx;
    if (i < N && j < N) D[(int64_t)i * N + j] = 1.f * tile[tx][r] / colmax[i];
  }
}

// B: K smallest of every row by repeated selection of the minimum greater than the previous pick in (value, index) order
__global__ __launch_bounds__(256) void rr_topk_kernel(const float* __restrict__ D, int N, int K, int* __restrict__ rank) {
  __shared__ float sv[256];
  __shared__ int si[256];
  const int i = blockIdx.x, t = threadIdx.x;
  const float* row = D + (int64_t)i * N;
  float pv = -INFINITY;
  int pi = -1;
  for (int k = 0; k < K; ++k) {
    float bv = INFINITY;
    int bi = 0x7fffffff;
    for (int j = t; j < N; j += 256) {
      const float v = row[j];
      const bool after = v > pv || (v == pv && j > pi);          // strictly after the previous pick
      if (after && (v < bv || (v == bv && j < bi))) { bv = v; bi = j; }
    }
    sv[t] = bv; si[t] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (t < s) {
        const float v = sv[t + s]; const int j = si[t + s];
        if (v < sv[t] || (v == sv[t] && j < si[t])) { sv[t] = v; si[t] = j; }
      }
      __syncthreads();
    }
    pv = sv[0]; pi = si[0];
    if (t == 0) rank[(int64_t)i * K + k] = pi < N ? pi : -1;
    __syncthreads();
  }
}

// C: expanded k-reciprocal set of row i and its normalised Gaussian weights.  One 64-thread block per row.
constexpr int RR_MAXK = 64;                       // k1 + 1 <= 64
constexpr int RR_MAXE = RR_MAXK * (RR_MAXK / 2 + 2);
__global__ __launch_bounds__(64) void rr_krecip_kernel(const float* __restrict__ D, const int* __restrict__ rank, int N,
                                                       int K, int Kh, float* __restrict__ V) {
  __shared__ int R[RR_MAXK], Rc[RR_MAXK], E[RR_MAXE];
  __shared__ int nR, nRc, nE, cnt;
  __shared__ float wsum[64];
  const int i = blockIdx.x, t = threadIdx.x;
  if (t == 0) { nR = 0; nE = 0; }
  __syncthreads();
  // R(i): f in rank[i][:K] with i in rank[f][:K]; kept in rank order (serialised append: K <= 64 iterations)
  for (int k = 0; k < K; ++k) {
    const int f = rank[(int64_t)i * K + k];
    bool hit = false;
    if (f >= 0) for (int u = t; u < K; u += 64) hit |= rank[(int64_t)f * K + u] == i;
    const bool any = __syncthreads_or(hit);
    if (any && t == 0) { R[nR] = f; E[nE] = f; ++nR; ++nE; }
    __syncthreads();
  }
  // expansion by the half-size reciprocal sets of the members of R(i)
  const int nr = nR;
  for (int c = 0; c < nr; ++c) {
    const int cand = R[c];
    if (t == 0) { nRc = 0; cnt = 0; }
    __syncthreads();
    for (int k = 0; k < Kh; ++k) {
      const int f = rank[(int64_t)cand * K + k];
      bool hit = false;
      if (f >= 0) for (int u = t; u < Kh; u += 64) hit |= rank[(int64_t)f * K + u] == cand;
      const bool any = __syncthreads_or(hit);
      if (any && t == 0) { Rc[nRc] = f; ++nRc; }
      __syncthreads();
    }
    const int nrc = nRc;
    if (t < nrc) {
      bool in = false;
      for (int u = 0; u < nr; ++u) in |= R[u] == Rc[t];
      if (in) atomicAdd(&cnt, 1);
    }
    __syncthreads();
    if ((double)cnt > 2. / 3 * (double)nrc) {          // rerank.py:72-76
      if (t < nrc) E[nE + t] = Rc[t];
      __syncthreads();
      if (t == 0) nE += nrc;
    }
    __syncthreads();
  }
  // unique(E) and weights: position p counts if no earlier position holds the same index
  const int ne = nE;
  float local = 0.f;
  for (int p = t; p < ne; p += 64) {
    const int e = E[p];
    bool first = true;
    for (int u = 0; u < p; ++u) first &= E[u] != e;
    if (first) local += expf(-D[(int64_t)i * N + e]);
  }
  wsum[t] = local;
  __syncthreads();
  for (int s = 32; s > 0; s >>= 1) { if (t < s) wsum[t] += wsum[t + s]; __syncthreads(); }
  const float total = wsum[0];
  for (int p = t; p < ne; p += 64) {
    const int e = E[p];
    V[(int64_t)i * N + e] = 1.f * expf(-D[(int64_t)i * N + e]) / total;      // duplicates rewrite the same value
  }
}

// D: Vq[i][c] = mean_t V[rank[i][t]][c], t < k2 (rows added in rank order, then one division)
__global__ __launch_bounds__(256) void rr_expand_kernel(const float* __restrict__ V, const int* __restrict__ rank, int N,
                                                        int K, int k2, float* __restrict__ Vq) {
  const int i = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  float s = 0.f;
  for (int t = 0; t < k2; ++t) {
    const int j = rank[(int64_t)i * K + t];
    const float v = V[(int64_t)j * N + c];
    s = t == 0 ? v : s + v;
  }
  Vq[(int64_t)i * N + c] = s / (float)k2;
}

// E + F: 32x32 (query, column) tile per block; c walks upwards in chunks of 32 staged in LDS
__global__ __launch_bounds__(256) void rr_jaccard_kernel(const float* __restrict__ Vq, const float* __restrict__ D, int N,
                                                         int Q, int G, float one_minus_lambda, float lambda,
                                                         float* __restrict__ out) {
  __shared__ float A[32][33], B[32][33];
  const int i0 = blockIdx.y * 32, j0 = Q + blockIdx.x * 32;       // only gallery columns are returned (:111)
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;         // thread: column tx, rows ty, ty+8, ty+16, ty+24
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < N; c0 += 32) {
    for (int r = ty; r < 32; r += 8) {
      const int c = c0 + tx;
      A[r][tx] = (i0 + r < Q && c < N) ? Vq[(int64_t)(i0 + r) * N + c] : 0.f;
      B[r][tx] = (j0 + r < N && c < N) ? Vq[(int64_t)(j0 + r) * N + c] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int c = 0; c < 32; ++c) {
      const float b = B[tx][c];
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] += fminf(A[ty + 8 * r][c], b);
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + ty + 8 * r, j = j0 + tx;
    if (i < Q && j < N) {
      const float tm = acc[r];
      const float jac = 1.f - tm / (2.f - tm);
      out[(int64_t)i * G + (j - Q)] = jac * one_minus_lambda + D[(int64_t)i * N + j] * lambda;
    }
  }
}

}  // namespace ieee

using namespace ieee;

extern "C" int64_t ieee_rerank_workspace_bytes(int64_t Q, int64_t G, int64_t k1) {
  const int64_t N = Q + G;
  return (3 * N * N + N) * 4 + N * (k1 + 1) * 4 + 256;
}

extern "C" int ieee_rerank(const float* q_g_dist, const float* q_q_dist, const float* g_g_dist, int64_t Q, int64_t G,
                           int64_t k1, int64_t k2, double lambda_value, float* out, void* work, int64_t work_bytes,
                           void* stream) {
  IEEE_REQUIRE(q_g_dist && q_q_dist && g_g_dist && out && work, "rerank: null pointer");
  IEEE_REQUIRE(Q > 0 && G > 0, "rerank: empty query or gallery set");
  const int64_t N = Q + G;
  IEEE_REQUIRE(k1 >= 1 && k1 + 1 <= RR_MAXK && k1 + 1 <= N, "rerank: k1 %ld out of range (1..%d, < Q+G)", (long)k1,
               RR_MAXK - 1);
  IEEE_REQUIRE(k2 >= 1 && k2 <= k1 + 1, "rerank: k2 %ld out of range (1..k1+1)", (long)k2);
  IEEE_REQUIRE(N < 46000, "rerank: the dense formulation holds three (Q+G)^2 fp32 matrices; Q+G = %ld is too large", (long)N);
  IEEE_REQUIRE(work_bytes >= ieee_rerank_workspace_bytes(Q, G, k1), "rerank: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* D = (float*)work;
  float* V = D + N * N;
  float* Vq = V + N * N;
  float* colmax = Vq + N * N;
  int* rank = (int*)(colmax + N);
  const int K = (int)k1 + 1;
  const int Kh = (int)nearbyint((double)k1 / 2.) + 1;        // np.around: round half to even (:63)
  OrigView o{q_g_dist, q_q_dist, g_g_dist, (int)Q, (int)G};
  rr_colmax_kernel<<<(unsigned)cdiv(N, 256), 256, 0, st>>>(o, colmax);
  IEEE_TRY(launch_status("rr_colmax_kernel"));
  rr_normalise_kernel<<<dim3((unsigned)cdiv(N, 32), (unsigned)cdiv(N, 32)), 256, 0, st>>>(o, colmax, D);
  IEEE_TRY(launch_status("rr_normalise_kernel"));
  rr_topk_kernel<<<(unsigned)N, 256, 0, st>>>(D, (int)N, K, rank);
  IEEE_TRY(launch_status("rr_topk_kernel"));
  IEEE_HIP(hipMemsetAsync(V, 0, sizeof(float) * (size_t)N * N, st));
  rr_krecip_kernel<<<(unsigned)N, 64, 0, st>>>(D, rank, (int)N, K, Kh, V);
  IEEE_TRY(launch_status("rr_krecip_kernel"));
  const float* Vuse = V;
  if (k2 != 1) {
    rr_expand_kernel<<<dim3((unsigned)cdiv(N, 256), (unsigned)N), 256, 0, st>>>(V, rank, (int)N, K, (int)k2, Vq);
    IEEE_TRY(launch_status("rr_expand_kernel"));
    Vuse = Vq;
  }
  rr_jaccard_kernel<<<dim3((unsigned)cdiv(G, 32), (unsigned)cdiv(Q, 32)), 256, 0, st>>>(
      Vuse, D, (int)N, (int)Q, (int)G, (float)(1.0 - lambda_value), (float)lambda_value, out);   // numpy casts the python scalars to f32
  return launch_status("rr_jaccard_kernel");
}
