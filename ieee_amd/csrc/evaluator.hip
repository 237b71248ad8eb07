// Evaluator hot path: squared-Euclidean / cosine distance matrix (tiled MFMA
// GEMM with fused norm epilogue) and Market1501-protocol CMC / mAP.
// Reference arithmetic: torchreid/metrics/distance.py:49-80, rank.py:103-171.
#include "gemm_core.h"

namespace ieee {

// XCD-aware, grouped tile order: the 8 XCDs each get a contiguous run of the
// linear tile order (blocks b, b+8, ... share an XCD/L2), and the linear order
// walks GROUP m-tiles per n-tile so co-resident blocks share operand panels.
__device__ __forceinline__ void tile_map(int tiles_m, int tiles_n, int group, int& tm, int& tn) {
  const int nwg = tiles_m * tiles_n;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int per = group * tiles_n;
  const int gi = wg / per, first = gi * group;
  const int gsz = min(tiles_m - first, group);
  const int in = wg - gi * per;
  tm = first + in % gsz;
  tn = in / gsz;
}

struct DistEpi {
  static constexpr bool kStaged = false;
  float* out;
  const float* qn;
  const float* gn;
  int64_t ldo;
  int m, n, metric;
  __device__ __forceinline__ void operator()(int i, int j, f32x4 v) const {
    if (i >= m) return;
    float* o = out + (int64_t)i * ldo + j;
    if (metric == 0) {
      const float a = qn[i];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (j + r < n) o[r] = (a + gn[j + r]) + (-2.0f * v[r]);   // distance.py:62-63
    } else {
      const float a = qn[i];   // holds 1/max(|q|,eps)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (j + r < n) o[r] = 1.0f - v[r] * a * gn[j + r];         // distance.py:77-79
    }
  }
};

#ifndef IEEE_DIST_STAGES
#define IEEE_DIST_STAGES 1   // one LDS stage for fp32 too: 32 KB -> more workgroups per CU, 112 -> 121 TFLOP/s
#endif
template <typename T>
__global__ __launch_bounds__(256, (sizeof(T) == 2 ? 4 : 1)) void distmat_kernel(const T* q, const T* g, const float* qn, const float* gn,
                                                      float* out, int m, int n, int d, int64_t ldo, int metric,
                                                      int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tm, tn;
  tile_map(tiles_m, tiles_n, 8, tm, tn);
  const int m0 = tm * 128, n0 = tn * 128;
  LoaderPlainNT<T, 4> la, lb;
  DistEpi epi{out, qn, gn, ldo, m, n, metric};
  if constexpr (sizeof(T) == 2) {   // single-stage LDS-DMA pipeline, 4 workgroups per CU (see conv.hip plan_gather)
    const int ch = nt_dma_chunk(threadIdx.x);
    la.init(q, d, m0, m, d, ch);
    lb.init(g, d, n0, n, d, ch);
    gemm_nt_dma<128, 128, 1>(la, lb, epi, (d + 63) / 64, m0, n0, smem);
    return;
  }
  la.init(q, d, m0, m, d);
  lb.init(g, d, n0, n, d);
  gemm_nt<T, 128, 128, IEEE_DIST_STAGES>(la, lb, epi, (d + ImgNT<T>::BK - 1) / ImgNT<T>::BK, m0, n0, smem);
}

// one wave per row: sum of squares (metric 0) or 1/max(norm, 1e-12) (metric 1)
template <typename T>
__global__ void rownorm_kernel(const T* x, int64_t rows, int d, int metric, float* out) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  constexpr int VEC = 16 / sizeof(T);
  const T* p = x + row * d;
  float s = 0.f;
  for (int k = lane * VEC; k < d; k += 64 * VEC) {
    float f[VEC];
    Vec16<T>::unpack(*(const uint4*)(p + k), f);
#pragma unroll
    for (int e = 0; e < VEC; ++e) s += f[e] * f[e];
  }
  s = wave_sum(s);
  if (lane == 0) out[row] = metric == 0 ? s : 1.0f / fmaxf(sqrtf(s), 1e-12f);
}

// ---------------------------------------------------------------- CMC / mAP
constexpr int RANK_CAP = 2048;   // match keys sorted per batch (LDS)

__device__ __forceinline__ uint64_t make_key(float d, uint32_t idx) {
  uint32_t u = __float_as_uint(d);
  u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;   // total order of IEEE floats
  return ((uint64_t)u << 32) | idx;
}

// exclusive scan of one uint per thread over the 256-thread block; returns the
// exclusive prefix and writes the block total to *total (uses 8 LDS words).
__device__ __forceinline__ uint32_t block_scan_excl(uint32_t v, uint32_t* wsum, uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = __shfl_up(inc, o);
    if (lane >= o) inc += y;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const uint32_t s = wsum[w];
    if (w < wave) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__global__ __launch_bounds__(256) void rank_query_kernel(const float* distmat, int64_t ldd, int num_g,
                                                         const int32_t* q_pids, const int32_t* g_pids,
                                                         const int32_t* q_camids, const int32_t* g_camids,
                                                         double* ap_out, int32_t* first_out) {
  __shared__ uint64_t keys[RANK_CAP];
  __shared__ uint32_t hist[RANK_CAP + 1];
  __shared__ uint32_t histm[RANK_CAP + 1];
  __shared__ uint32_t wsum[4];
  __shared__ double dsum[256];
  __shared__ int32_t imin[256];

  const int q = blockIdx.x, t = threadIdx.x;
  const float* row = distmat + (int64_t)q * ldd;
  const int32_t qpid = q_pids[q], qcam = q_camids[q];

  double ap_sum = 0.0;
  int64_t nm_total = 0;
  int32_t first = 0x7fffffff;
  int jstart = 0;

  while (jstart < num_g) {
    // ---- collect up to RANK_CAP matches (same pid, different camera) in gallery-index order
    uint32_t nb = 0;
    int base = jstart;
    while (base < num_g) {
      const int j = base + t;
      const bool is = j < num_g && g_pids[j] == qpid && g_camids[j] != qcam;
      uint32_t tot;
      const uint32_t pos = block_scan_excl(is ? 1u : 0u, wsum, &tot);
      if (nb + tot > (uint32_t)RANK_CAP) break;   // uniform: leave this chunk for the next batch
      if (is) keys[nb + pos] = make_key(row[j], (uint32_t)j);
      nb += tot;
      base += 256;
    }
    jstart = base;
    if (nb == 0) continue;   // (jstart advanced to num_g) no match in the remainder
    // ---- bitonic sort of keys[0..np2) ascending, padded with +inf keys
    uint32_t np2 = 1;
    while (np2 < nb) np2 <<= 1;
    for (uint32_t i = nb + t; i < np2; i += 256) keys[i] = ~0ull;
    for (uint32_t i = t; i <= nb; i += 256) { hist[i] = 0; histm[i] = 0; }
    __syncthreads();
    for (uint32_t k = 2; k <= np2; k <<= 1) {
      for (uint32_t j = k >> 1; j > 0; j >>= 1) {
        for (uint32_t i = t; i < np2; i += 256) {
          const uint32_t l = i ^ j;
          if (l > i) {
            const uint64_t a = keys[i], b = keys[l];
            const bool up = (i & k) == 0;
            if ((a > b) == up) { keys[i] = b; keys[l] = a; }
          }
        }
        __syncthreads();
      }
    }
    const uint64_t kmax = keys[nb - 1];
    // ---- stream the row: for every kept element, idx = #{batch keys < key}; it precedes
    //      batch matches idx..nb-1, so hist[idx]++ and ranks are prefix sums of hist.
    for (int k0 = t * 4; k0 < num_g; k0 += 1024) {
      float dv[4];
      int32_t pv[4];
      if (k0 + 3 < num_g && ((ldd & 3) == 0)) {
        const float4 d4 = *(const float4*)(row + k0);
        const int4 p4 = *(const int4*)(g_pids + k0);
        dv[0] = d4.x; dv[1] = d4.y; dv[2] = d4.z; dv[3] = d4.w;
        pv[0] = p4.x; pv[1] = p4.y; pv[2] = p4.z; pv[3] = p4.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dv[e] = k0 + e < num_g ? row[k0 + e] : 0.f;
          pv[e] = k0 + e < num_g ? g_pids[k0 + e] : (qpid ^ 0x40000000);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = k0 + e;
        if (k >= num_g) continue;
        bool ism = false;
        if (pv[e] == qpid) {
          if (g_camids[k] == qcam) continue;   // removed: same pid & same camera (rank.py:136-137)
          ism = true;
        }
        const uint64_t key = make_key(dv[e], (uint32_t)k);
        if (key > kmax) continue;
        uint32_t lo = 0, hi = nb;              // lower_bound: first index with keys[idx] >= key
        while (lo < hi) {
          const uint32_t mid = (lo + hi) >> 1;
          if (keys[mid] < key) lo = mid + 1; else hi = mid;
        }
        atomicAdd(&hist[lo], 1u);
        if (ism) atomicAdd(&histm[lo], 1u);
      }
    }
    __syncthreads();
    // ---- inclusive prefix sums over hist / histm (chunks of 256 with a running carry)
    uint32_t carry_h = 0, carry_m = 0;
    double part = 0.0;
    int32_t fmin = 0x7fffffff;
    for (uint32_t c0 = 0; c0 < nb; c0 += 256) {
      const uint32_t i = c0 + t;
      const uint32_t hv = i < nb ? hist[i] : 0u, mv = i < nb ? histm[i] : 0u;
      uint32_t th, tmm;
      const uint32_t eh = block_scan_excl(hv, wsum, &th);
      const uint32_t em = block_scan_excl(mv, wsum, &tmm);
      if (i < nb) {
        const uint32_t rank = carry_h + eh + hv - 1;    // 0-based position among kept (itself is in hist[i])
        const uint32_t mrank = carry_m + em + mv - 1;   // matches before it (all batches)
        part += (double)(mrank + 1) / (double)(rank + 1);   // rank.py:156-157
        fmin = min(fmin, (int32_t)rank);
      }
      carry_h += th;
      carry_m += tmm;
    }
    dsum[t] = part;
    imin[t] = fmin;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (t < o) { dsum[t] += dsum[t + o]; imin[t] = min(imin[t], imin[t + o]); }
      __syncthreads();
    }
    ap_sum += dsum[0];
    first = min(first, imin[0]);
    nm_total += nb;
    __syncthreads();
  }
  if (t == 0) {
    ap_out[q] = nm_total > 0 ? ap_sum / (double)nm_total : -1.0;   // rank.py:153-158
    first_out[q] = nm_total > 0 ? first : -1;
  }
}

__global__ void rank_finalize_kernel(const double* ap, const int32_t* first, int num_q, int max_rank,
                                     int64_t* summary) {
  __shared__ unsigned long long cnt[1024];
  __shared__ unsigned long long nvalid;
  const int t = threadIdx.x;
  for (int r = t; r < max_rank; r += blockDim.x) cnt[r] = 0;
  if (t == 0) nvalid = 0;
  __syncthreads();
  for (int q = t; q < num_q; q += blockDim.x) {
    const int f = first[q];
    if (f >= 0) {
      atomicAdd(&nvalid, 1ull);
      for (int r = f; r < max_rank; ++r) atomicAdd(&cnt[r], 1ull);   // cmc = min(cumsum,1): rank.py:145-150
    }
  }
  __syncthreads();
  for (int r = t; r < max_rank; r += blockDim.x) summary[r] = (int64_t)cnt[r];
  if (t == 0) {
    summary[max_rank] = (int64_t)nvalid;
    double s = 0.0;
    for (int q = 0; q < num_q; ++q)
      if (first[q] >= 0) s += ap[q];
    summary[max_rank + 1] = __double_as_longlong(s);
  }
}

}  // namespace ieee

using namespace ieee;

extern "C" int ieee_sqeuclid_distmat(const void* q, const void* g, int64_t m, int64_t n, int64_t d, int dtype,
                                     int metric, float* out, int64_t ldo, void* work, void* stream) {
  IEEE_REQUIRE(q && g && out && work, "distmat: null pointer");
  IEEE_REQUIRE(m > 0 && n > 0 && d > 0, "distmat: empty input (m=%ld n=%ld d=%ld)", (long)m, (long)n, (long)d);
  IEEE_REQUIRE(d % 8 == 0, "distmat: feature dim %ld must be a multiple of 8", (long)d);
  IEEE_REQUIRE(ldo >= n, "distmat: ldo < n");
  IEEE_REQUIRE(metric == 0 || metric == 1, "distmat: unknown metric %d", metric);
  IEEE_REQUIRE(m < (1ll << 31) && n < (1ll << 31), "distmat: too many rows");
  hipStream_t st = (hipStream_t)stream;
  float* qn = (float*)work;
  float* gn = qn + m;
  const int tiles_m = cdiv(m, 128), tiles_n = cdiv(n, 128);
  const size_t smem = (dtype == IEEE_BF16 ? 1 : IEEE_DIST_STAGES) * 256 * 128;   // bf16: single LDS stage (see gemm_nt)
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)distmat_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  if (dtype == IEEE_F32) {
    rownorm_kernel<float><<<cdiv(m, 4), 256, 0, st>>>((const float*)q, m, (int)d, metric, qn);
    rownorm_kernel<float><<<cdiv(n, 4), 256, 0, st>>>((const float*)g, n, (int)d, metric, gn);
    distmat_kernel<float><<<tiles_m * tiles_n, 256, smem, st>>>((const float*)q, (const float*)g, qn, gn, out, (int)m,
                                                                (int)n, (int)d, ldo, metric, tiles_m, tiles_n);
  } else if (dtype == IEEE_BF16) {
    rownorm_kernel<bf16><<<cdiv(m, 4), 256, 0, st>>>((const bf16*)q, m, (int)d, metric, qn);
    rownorm_kernel<bf16><<<cdiv(n, 4), 256, 0, st>>>((const bf16*)g, n, (int)d, metric, gn);
    distmat_kernel<bf16><<<tiles_m * tiles_n, 256, smem, st>>>((const bf16*)q, (const bf16*)g, qn, gn, out, (int)m,
                                                               (int)n, (int)d, ldo, metric, tiles_m, tiles_n);
  } else {
    IEEE_REQUIRE(false, "distmat: unsupported dtype %d", dtype);
  }
  return launch_status("distmat");
}

extern "C" int ieee_rank_market1501(const float* distmat, int64_t ldd, int64_t num_q, int64_t num_g,
                                    const int32_t* q_pids, const int32_t* g_pids, const int32_t* q_camids,
                                    const int32_t* g_camids, int64_t max_rank, double* ap, int32_t* first_pos,
                                    int64_t* summary, void* stream) {
  IEEE_REQUIRE(distmat && q_pids && g_pids && q_camids && g_camids && ap && first_pos && summary, "rank: null pointer");
  IEEE_REQUIRE(num_q > 0 && num_g > 0, "rank: empty distmat");
  IEEE_REQUIRE(num_g < (1ll << 31) && num_q < (1ll << 31), "rank: too large");
  IEEE_REQUIRE(ldd >= num_g, "rank: ldd < num_g");
  if (max_rank > num_g) max_rank = num_g;   // rank.py:110-115
  IEEE_REQUIRE(max_rank >= 1 && max_rank <= 1024, "rank: max_rank %ld out of range [1,1024]", (long)max_rank);
  hipStream_t st = (hipStream_t)stream;
  rank_query_kernel<<<(int)num_q, 256, 0, st>>>(distmat, ldd, (int)num_g, q_pids, g_pids, q_camids, g_camids, ap,
                                                first_pos);
  rank_finalize_kernel<<<1, 256, 0, st>>>(ap, first_pos, (int)num_q, (int)max_rank, summary);
  return launch_status("rank_market1501");
}
