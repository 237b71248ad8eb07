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

// Epilogue of the distance GEMM, staged through LDS: a lane's accumulators are 4 consecutive columns of ONE row, so
// storing them directly scatters 64-byte pieces over 16 rows per instruction (the 4 GB output of the 10k x 100k case
// then costs 2.3 ms); instead each half of the 128 x 128 tile (64 rows, exactly the 32 KB operand stage) goes through
// an XOR-swizzled LDS image and leaves as full 512-byte rows, norms (and the fp16 split's row scales) applied on the
// way out.
struct DistEpi {
  static constexpr bool kStaged = true;
  float* out;
  const float* qn;
  const float* gn;
  int64_t ldo;
  int m, n, metric;
  const float* qsc;   // optional per-row powers of two that undo the operand scaling of the fp16 split (else null)
  const float* gsc;
  int nt;             // non-temporal stores (outputs beyond the 256 MB of MALL: 4.8 instead of 4.0 TB/s)
  __device__ __forceinline__ float value(float v, float a, float b) const {
    return metric == 0 ? (a + b) + (-2.0f * v)      // distance.py:62-63
                       : 1.0f - v * a * b;          // distance.py:77-79 (a, b hold 1/max(|.|,eps))
  }
  template <int BM, int BN, int FM, int FN>
  __device__ __forceinline__ void finish(f32x4 (&acc)[FM][FN], char* smem, int m0, int n0) const {
    static_assert(BM == 128 && BN == 128, "DistEpi: 128 x 128 tiles");
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    const int ch = t & 31, r0 = t >> 5;            // read-back: 16-byte chunk of the row, first row of the pass
    const int col = n0 + ch * 4;
    float gnv[4], gsv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      gnv[r] = col + r < n ? gn[col + r] : 0.f;
      gsv[r] = (gsc && col + r < n) ? gsc[col + r] : 1.f;
    }
    const bool vec_ok = (((uintptr_t)out | (uintptr_t)(ldo * 4)) & 15) == 0 && col + 3 < n;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      __syncthreads();   // h = 0: every wave is done with the operand stage; h = 1: the first half has been read
      if (wm == h) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            const int r = i * 16 + (lane & 15);
            const int c = wn * 16 + j * 4 + (lane >> 4);           // 16-byte chunk index within the 512-byte row
            *(float4*)(smem + r * 512 + ((c ^ (r & 31)) << 4)) =
                make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
          }
      }
      __syncthreads();
#pragma unroll
      for (int pss = 0; pss < 8; ++pss) {
        const int r = r0 + 8 * pss;
        const int row = m0 + h * 64 + r;
        if (row >= m || col >= n) continue;
        const float4 v4 = *(const float4*)(smem + r * 512 + ((ch ^ (r & 31)) << 4));
        float v[4] = {v4.x, v4.y, v4.z, v4.w};
        const float a = qn[row];
        if (qsc) {
          const float sq = qsc[row];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= sq * gsv[e];
        }
        float* o = out + (int64_t)row * ldo + col;
#ifdef IEEE_DIST_NOSTORE_PROBE   // measurement build only (scripts/distmat_nostore_probe.sh): the GEMM with everything but its 4 GB of stores
        if (!(value(v[0], a, gnv[0]) != value(v[0], a, gnv[0]))) continue;
#endif
        if (vec_ok) {
          typedef float f32x4v __attribute__((ext_vector_type(4)));
          const f32x4v ov = {value(v[0], a, gnv[0]), value(v[1], a, gnv[1]), value(v[2], a, gnv[2]), value(v[3], a, gnv[3])};
          if (nt) __builtin_nontemporal_store(ov, (f32x4v*)o);   // output larger than the caches: stream it
          else *(f32x4v*)o = ov;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (col + e < n) o[e] = value(v[e], a, gnv[e]);
        }
      }
    }
  }
};

#ifndef IEEE_DIST_STAGES
#define IEEE_DIST_STAGES 1   // one LDS stage for fp32 too: 32 KB -> more workgroups per CU, 112 -> 121 TFLOP/s
#endif
template <typename T, bool F16 = false>
__global__ __launch_bounds__(256, (sizeof(T) == 2 ? 4 : 1)) void distmat_kernel(const T* q, const T* g, const float* qn, const float* gn,
                                                      float* out, int m, int n, int d, int64_t ldo, int metric,
                                                      int tiles_m, int tiles_n, const float* qsc = nullptr,
                                                      const float* gsc = nullptr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tm, tn;
  tile_map(tiles_m, tiles_n, 8, tm, tn);
  const int m0 = tm * 128, n0 = tn * 128;
  LoaderPlainNT<T, 4> la, lb;
  DistEpi epi{out, qn, gn, ldo, m, n, metric, qsc, gsc, (int64_t)m * n * 4 > (256ll << 20) ? 1 : 0};
  if constexpr (sizeof(T) == 2) {   // single-stage LDS-DMA pipeline, 4 workgroups per CU (see conv.hip plan_gather)
    const int ch = nt_dma_chunk(threadIdx.x);
    la.init(q, d, m0, m, d, ch);
    lb.init(g, d, n0, n, d, ch);
    gemm_nt_dma<128, 128, 1, F16>(la, lb, epi, (d + 63) / 64, m0, n0, smem);
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

// ---------------------------------------------------------------- split-bf16 operands
// An fp32 value is the exact sum of three bf16 pieces (8 + 8 + 8 mantissa bits): hi = bf16(x), mid = bf16(x - hi),
// lo = bf16(x - hi - mid); every piece product is exact in the fp32 MFMA accumulator.  q.g is then a bf16 GEMM over
// a K axis that lists the piece products, smallest first:
//   6 terms (fp32-grade; drops only the 2^-24-relative mid.lo, lo.mid, lo.lo):  lo.hi  hi.lo  mid.mid  mid.hi  hi.mid  hi.hi
//   3 terms (two pieces, ~2^-16 relative):                                       lo.hi  hi.lo  hi.hi
// This kernel writes one operand's row as [terms][d] bf16: side 0 = the query pieces of each term, side 1 = the
// gallery pieces.
template <int TERMS>
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ x, bf16* __restrict__ out,
                                                         int64_t rows, int d, int side) {
  const int chunks = d / 8;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * chunks) return;
  const int64_t row = i / chunks;
  const int k = (int)(i - row * chunks) * 8;
  const float* p = x + row * d + k;
  float v[8], hi[8], mid[8], lo[8];
  Vec16<float>::unpack(*(const uint4*)p, v);
  Vec16<float>::unpack(*(const uint4*)(p + 4), v + 4);
  const uint4 hp = Vec16<bf16>::pack(v);
  Vec16<bf16>::unpack(hp, hi);
#pragma unroll
  for (int e = 0; e < 8; ++e) mid[e] = v[e] - hi[e];          // exact
  const uint4 mp = Vec16<bf16>::pack(mid);                     // TERMS == 3: this is the second (last) piece
  uint4 lp = mp;
  if (TERMS == 6) {
    float mr[8];
    Vec16<bf16>::unpack(mp, mr);
#pragma unroll
    for (int e = 0; e < 8; ++e) lo[e] = mid[e] - mr[e];        // exact
    lp = Vec16<bf16>::pack(lo);
  }
  bf16* o = out + row * (int64_t)(TERMS * d) + k;
  if (TERMS == 6) {
    // query side:   lo hi mid mid hi hi      gallery side:   hi lo mid hi mid hi
    const uint4 a[6] = {lp, hp, mp, mp, hp, hp};
    const uint4 b[6] = {hp, lp, mp, hp, mp, hp};
#pragma unroll
    for (int t = 0; t < 6; ++t) *(uint4*)(o + (int64_t)t * d) = side == 0 ? a[t] : b[t];
  } else {
    const uint4 a[3] = {mp, hp, hp};
    const uint4 b[3] = {hp, mp, hp};
#pragma unroll
    for (int t = 0; t < 3; ++t) *(uint4*)(o + (int64_t)t * d) = side == 0 ? a[t] : b[t];
  }
}

// Two fp16 pieces (11 + 11 mantissa bits) and three products lo.hi, hi.lo, hi.hi: 2^-22 relative, at half the MFMA
// work of the six bf16 products.  fp16 has a narrow exponent, so every row is first scaled by a power of two that
// puts its largest magnitude just below 2^14 (exact); the inverse scales go to `inv_scale` and the GEMM epilogue
// multiplies them back (exact as well).  One wave per row.
__global__ __launch_bounds__(256) void split_rows_f16_kernel(const float* __restrict__ x, uint16_t* __restrict__ out,
                                                             float* __restrict__ inv_scale, int64_t rows, int d,
                                                             int side) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* p = x + row * d;
  float mx = 0.f;
  for (int k = lane * 4; k < d; k += 256) {
    const float4 v = *(const float4*)(p + k);
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  mx = wave_max(mx);
  int e = 0;
  if (mx > 0.f && mx < 3.0e38f) (void)frexpf(mx, &e);        // mx = f * 2^e, f in [0.5, 1)
  const float sc = ldexpf(1.0f, 14 - e), isc = ldexpf(1.0f, e - 14);
  if (lane == 0) inv_scale[row] = isc;
  typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
  uint16_t* o = out + row * (int64_t)(3 * d);
  for (int k = lane * 4; k < d; k += 256) {
    const float4 v = *(const float4*)(p + k);
    const float y[4] = {v.x * sc, v.y * sc, v.z * sc, v.w * sc};
    f16x4 hi, lo;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      hi[c] = (_Float16)y[c];
      lo[c] = (_Float16)(y[c] - (float)hi[c]);               // the subtraction is exact
    }
    const uint2 hp = __builtin_bit_cast(uint2, hi), lp = __builtin_bit_cast(uint2, lo);
    // query side: lo hi hi      gallery side: hi lo hi
    *(uint2*)(o + k) = side == 0 ? lp : hp;
    *(uint2*)(o + d + k) = side == 0 ? hp : lp;
    *(uint2*)(o + 2 * (int64_t)d + k) = hp;
  }
}

// ---------------------------------------------------------------- CMC / mAP
constexpr int RANK_CAP = 1024;    // match keys sorted per batch (LDS)
#ifndef IEEE_RANK_CELLS
#define IEEE_RANK_CELLS 512
#endif
#ifndef IEEE_RANK_UNR
#define IEEE_RANK_UNR 8
#endif
constexpr int RANK_CELLS = IEEE_RANK_CELLS;  // distance cells over the batch's match range

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

// Cell of a distance on the uniform grid laid over [dmin, dmax] of a batch's sorted match keys.  Every step
// (subtract a constant, multiply by a non-negative constant, min, truncate) is monotone under rounding, so
// a < b implies cell(a) <= cell(b): keys in lower cells are smaller, keys in higher cells are larger, and only
// the keys sharing the element's cell have to be compared.
__device__ __forceinline__ uint32_t rank_cell(float d, float dmin, float scale) {
  return (uint32_t)(int)fminf((d - dmin) * scale, (float)(RANK_CELLS - 1));   // NaN -> last cell
}

__device__ __forceinline__ float key_dist(uint64_t key) {
  uint32_t u = (uint32_t)(key >> 32);
  u ^= (u >> 31) ? 0x80000000u : 0xFFFFFFFFu;
  return __uint_as_float(u);
}

// ---- identity buckets: gallery indices grouped by a hash of their identity (counting sort, order inside a bucket
// arbitrary -- the consumers sort what they collect).  work: start[RANK_BUCKETS + 1], cursor[RANK_BUCKETS], idx[num_g].
constexpr int RANK_BUCKETS = 16384;
__device__ __forceinline__ uint32_t rank_bucket_of(int32_t pid) { return ((uint32_t)pid * 2654435761u) >> 18; }

__global__ __launch_bounds__(256) void rank_bucket_count_kernel(const int32_t* __restrict__ g_pids, int num_g, int32_t* count) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j < num_g) atomicAdd(&count[1 + rank_bucket_of(g_pids[j])], 1);
}
// start[0] = 0 and start[1 + b] = count of bucket b on entry; exclusive starts on exit, copied into cursor
__global__ __launch_bounds__(1024) void rank_bucket_scan_kernel(int32_t* start, int32_t* cursor) {
  constexpr int PER = RANK_BUCKETS / 1024;
  __shared__ int32_t wsum[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int32_t c[PER], sum = 0;
#pragma unroll
  for (int e = 0; e < PER; ++e) { c[e] = start[1 + t * PER + e]; sum += c[e]; }
  int32_t inc = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int32_t v = __shfl_up(inc, o); if (lane >= o) inc += v; }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int32_t base = 0;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  int32_t run = base + inc - sum;
#pragma unroll
  for (int e = 0; e < PER; ++e) { cursor[t * PER + e] = run; run += c[e]; start[1 + t * PER + e] = run; }
}
__global__ __launch_bounds__(256) void rank_bucket_fill_kernel(const int32_t* __restrict__ g_pids, int num_g, int32_t* cursor,
                                                               int32_t* idx) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j < num_g) idx[atomicAdd(&cursor[rank_bucket_of(g_pids[j])], 1)] = j;
}

// The common case of the evaluator, one workgroup per query: at most RANK_CAP true matches and at most RANK_CAP
// removed entries (same identity, same camera).  No sort of the gallery row and no identity lookups while it is
// streamed: the matches are sorted in LDS, a grid of RANK_CELLS cells over their distance range maps a streamed
// distance to the few match keys it has to be compared with (usually none), every element adds 1 to
// hist[slot], slot = 1 + cell + #{match keys < key} being monotone in the key, the removed entries are then
// subtracted from the slots they were counted in, and the rank of match i is the inclusive prefix sum at
// 1 + cell_i + i.  A query that does not fit writes first_out = -2 and is left to rank_query_kernel.
// DBG (measurement only, IEEE_RANK_DBG; wrong results): 1 = the per-element LDS histogram add is replaced by a register
// sum (what the atomics cost), 2 = the per-element cell-table read is skipped (slot = 1 + cell: what the lookups cost),
// 3 = both.  LABNOTES R6.5 uses them to split the kernel's LDS-conflict share between the two accesses.
template <int DBG = 0>
__global__ __launch_bounds__(256) void rank_query_fast_kernel(const float* distmat, int64_t ldd, int num_g,
                                                              const int32_t* q_pids, const int32_t* g_pids,
                                                              const int32_t* q_camids, const int32_t* g_camids,
                                                              double* ap_out, int32_t* first_out,
                                                              const int32_t* __restrict__ bucket_start,
                                                              const int32_t* __restrict__ bucket_idx) {
  constexpr int HIST = RANK_CELLS + RANK_CAP + 2;
  __shared__ uint64_t keys[RANK_CAP];
  __shared__ uint32_t removed[RANK_CAP];
  __shared__ uint32_t hist[HIST + 512];          // slack: the scan reads whole per-thread spans
  __shared__ uint32_t cellinfo[RANK_CELLS];      // first key index of the cell | keys in the cell << 16
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t counts[2];
  __shared__ double dsum[256];
  __shared__ int32_t imin[256];

  const int q = blockIdx.x, t = threadIdx.x;
  const float* row = distmat + (int64_t)q * ldd;
  const int32_t qpid = q_pids[q], qcam = q_camids[q];
  if (t < 2) counts[t] = 0;
  __syncthreads();
  // ---- collect the matches (as sort keys) and the removed entries (as gallery indices); any order
  auto collect = [&](int32_t pid, int j) {
    if (pid != qpid) return;
    if (g_camids[j] != qcam) {
      const uint32_t pos = atomicAdd(&counts[0], 1u);
      if (pos < (uint32_t)RANK_CAP) keys[pos] = make_key(row[j], (uint32_t)j);
    } else {
      const uint32_t pos = atomicAdd(&counts[1], 1u);
      if (pos < (uint32_t)RANK_CAP) removed[pos] = (uint32_t)j;
    }
  };
  if (bucket_start) {
    // the gallery entries were bucketed by identity hash once per call (rank_bucket_* below): this query looks at its
    // identity's bucket only -- a few dozen entries -- instead of all num_g identities (400 KB of L2 reads per query)
    const uint32_t h = rank_bucket_of(qpid);
    const int b0 = bucket_start[h], b1 = bucket_start[h + 1];
    for (int i = b0 + t; i < b1; i += 256) {
      const int j = bucket_idx[i];
      collect(g_pids[j], j);
    }
  } else {
    int j = 0;
    for (; ((uintptr_t)g_pids & 15) == 0 && j + 1024 <= num_g; j += 1024) {
      const int4 p4 = *(const int4*)(g_pids + j + t * 4);
      collect(p4.x, j + t * 4);
      collect(p4.y, j + t * 4 + 1);
      collect(p4.z, j + t * 4 + 2);
      collect(p4.w, j + t * 4 + 3);
    }
    for (j += t; j < num_g; j += 256) collect(g_pids[j], j);
  }
  __syncthreads();
  const uint32_t nb = counts[0], nr = counts[1];
  if (nb > (uint32_t)RANK_CAP || nr > (uint32_t)RANK_CAP) {
    if (t == 0) first_out[q] = -2;
    return;
  }
  if (nb == 0) {
    if (t == 0) { ap_out[q] = -1.0; first_out[q] = -1; }   // rank.py:140-142: no valid match, query skipped
    return;
  }
  // ---- bitonic sort of keys[0..np2) ascending, padded with +inf keys
  uint32_t np2 = 1;
  while (np2 < nb) np2 <<= 1;
  for (uint32_t i = nb + t; i < np2; i += 256) keys[i] = ~0ull;
  const uint32_t hlen = RANK_CELLS + nb + 2;
  for (uint32_t i = t; i < hlen + 512; i += 256) hist[i] = 0;
  for (uint32_t i = t; i < RANK_CELLS; i += 256) cellinfo[i] = 0;
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
  const float dmin = key_dist(keys[0]), dmax = key_dist(keys[nb - 1]);
  float scale = (float)RANK_CELLS / (dmax - dmin);
  if (!(scale < 1e30f)) scale = 0.f;             // one distinct distance (or inf/NaN): a single cell
  for (uint32_t i = t; i < nb; i += 256) atomicAdd(&cellinfo[rank_cell(key_dist(keys[i]), dmin, scale)], 1u);
  __syncthreads();
  {
    constexpr int PER = RANK_CELLS / 256;
    uint32_t c[PER], sum = 0;
#pragma unroll
    for (int e = 0; e < PER; ++e) { c[e] = cellinfo[t * PER + e]; sum += c[e]; }
    uint32_t tot;
    uint32_t run = block_scan_excl(sum, wsum, &tot);
#pragma unroll
    for (int e = 0; e < PER; ++e) { cellinfo[t * PER + e] = run | (c[e] << 16); run += c[e]; }
  }
  __syncthreads();
  // slot of an element, or -1 when it lies after every match (no rank depends on it).  Float compares decide
  // whenever they can (d > dmax, d < dmin, a cell without keys); only elements that share a cell with match keys
  // build the 64-bit (distance, index) key.  NaN distances go last, as in numpy's argsort.
  auto slot_of = [&](float d, int k) -> int {
    if (d > dmax) return -1;
    if (d < dmin) return 0;
    const uint32_t c = rank_cell(d, dmin, scale);
    if constexpr (DBG == 2 || DBG == 3) return (int)(1 + c);
    const uint32_t ci = cellinfo[c];
    uint32_t lo = ci & 0xffffu;
    if (ci >> 16) {
      const uint64_t key = make_key(d, (uint32_t)k);
      uint32_t hi = lo + (ci >> 16);             // lower_bound among the keys of this cell
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1; else hi = mid;
      }
    }
    return (int)(1 + c + lo);
  };
  uint32_t below = 0;
  auto visit = [&](float d, int k) {
    const int sl = slot_of(d, k);
    if constexpr (DBG == 1 || DBG == 3) { below += (uint32_t)sl; return; }
    if (sl > 0) atomicAdd(&hist[sl], 1u);
    else if (sl == 0) ++below;
  };
  // ---- stream the row: 4 independent 16-byte loads per thread in flight before any is used
  constexpr int UNR = IEEE_RANK_UNR;
  int kdone = 0;
  if (((uintptr_t)row & 15) == 0) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    for (; kdone + UNR * 1024 <= num_g; kdone += UNR * 1024) {
      f32x4 d4[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) d4[u] = __builtin_nontemporal_load((const f32x4*)(row + kdone + u * 1024 + t * 4));
      // (round 6, measured and removed: a two-pass form -- every element's cell and cell-table word first, all LDS reads in
      // flight together, then the searches and histogram adds -- 1.36 -> 1.87 ms on random distances, 0.97 -> 1.28 with the
      // matches nearest: 32 more live registers per thread and a second divergent pass cost more than the overlap gives)
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int k = kdone + u * 1024 + t * 4;
        visit(d4[u].x, k);
        visit(d4[u].y, k + 1);
        visit(d4[u].z, k + 2);
        visit(d4[u].w, k + 3);
      }
    }
  }
  for (int k = kdone + t; k < num_g; k += 256) visit(row[k], k);
  if (below) atomicAdd(&hist[0], below);
  __syncthreads();
  // ---- take the removed entries (rank.py:136-137) out of the slots the stream counted them in
  for (uint32_t i = t; i < nr; i += 256) {
    const int k = (int)removed[i];
    const int sl = slot_of(row[k], k);
    if (sl >= 0) atomicSub(&hist[sl], 1u);
  }
  __syncthreads();
  // ---- inclusive prefix sums over hist in place: a contiguous odd-length span per thread (conflict-free)
  {
    const uint32_t per = ((hlen + 255) / 256) | 1u;
    uint32_t sum = 0;
    for (uint32_t e = 0; e < per; ++e) sum += hist[t * per + e];
    uint32_t tot;
    uint32_t run = block_scan_excl(sum, wsum, &tot);
    for (uint32_t e = 0; e < per; ++e) { run += hist[t * per + e]; hist[t * per + e] = run; }
  }
  __syncthreads();
  double part = 0.0;
  int32_t fmin = 0x7fffffff;
  for (uint32_t i = t; i < nb; i += 256) {
    const uint32_t c = rank_cell(key_dist(keys[i]), dmin, scale);
    const uint32_t rank = hist[1 + c + i] - 1;         // 0-based position among kept (itself is counted)
    part += (double)(i + 1) / (double)(rank + 1);      // rank.py:156-157; i matches precede match i
    fmin = min(fmin, (int32_t)rank);
  }
  dsum[t] = part;
  imin[t] = fmin;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) { dsum[t] += dsum[t + o]; imin[t] = min(imin[t], imin[t + o]); }
    __syncthreads();
  }
  if (t == 0) {
    ap_out[q] = dsum[0] / (double)nb;                  // rank.py:153-158
    first_out[q] = imin[0];
  }
}

// The general case (any number of matches, in batches of RANK_CAP): launched after rank_query_fast_kernel, works
// only on the queries that kernel flagged with first_out = -2.
// One workgroup per query.  No sort of the gallery row: the (few) true matches are sorted in LDS, a grid of
// RANK_CELLS cells over their distance range maps a streamed distance to the handful of match keys it has to be
// compared with (usually none), and every kept element adds 1 to hist[1 + cell + #{batch keys < key}] -- an index
// that is monotone in the key -- so the rank of match i is the inclusive prefix sum at 1 + cell_i + i.
__global__ __launch_bounds__(256) void rank_query_kernel(const float* distmat, int64_t ldd, int num_g,
                                                         const int32_t* q_pids, const int32_t* g_pids,
                                                         const int32_t* q_camids, const int32_t* g_camids,
                                                         double* ap_out, int32_t* first_out, int only_flagged) {
  constexpr int HIST = RANK_CELLS + RANK_CAP + 2;
  __shared__ uint64_t keys[RANK_CAP];
  __shared__ uint32_t hist[HIST + 512];          // slack: the scan reads whole per-thread spans
  __shared__ uint32_t histm[RANK_CAP + 1];
  __shared__ uint32_t cellinfo[RANK_CELLS];      // first key index of the cell | keys in the cell << 16
  __shared__ uint32_t wsum[4];
  __shared__ double dsum[256];
  __shared__ int32_t imin[256];

  const int q = blockIdx.x, t = threadIdx.x;
  if (only_flagged && first_out[q] != -2) return;
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
    const bool vec_ok = (((uintptr_t)row | (uintptr_t)g_pids) & 15) == 0;
    while (base < num_g) {
      // 4 consecutive gallery entries per thread, one block scan per 1024 (= RANK_CAP, so a chunk always fits)
      const int j0 = base + t * 4;
      int32_t pv[4];
      if (vec_ok && j0 + 3 < num_g) {
        const int4 p4 = *(const int4*)(g_pids + j0);
        pv[0] = p4.x; pv[1] = p4.y; pv[2] = p4.z; pv[3] = p4.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) pv[e] = j0 + e < num_g ? g_pids[j0 + e] : (qpid ^ 0x40000000);
      }
      uint32_t mask = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (pv[e] == qpid && g_camids[j0 + e] != qcam) mask |= 1u << e;
      uint32_t tot;
      uint32_t pos = nb + block_scan_excl(__popc(mask), wsum, &tot);
      if (nb + tot > (uint32_t)RANK_CAP) break;   // uniform: leave this chunk for the next batch
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (mask & (1u << e)) keys[pos++] = make_key(row[j0 + e], (uint32_t)(j0 + e));
      nb += tot;
      base += 1024;
    }
    jstart = base;
    if (nb == 0) continue;   // (jstart advanced to num_g) no match in the remainder
    // ---- bitonic sort of keys[0..np2) ascending, padded with +inf keys
    uint32_t np2 = 1;
    while (np2 < nb) np2 <<= 1;
    for (uint32_t i = nb + t; i < np2; i += 256) keys[i] = ~0ull;
    const uint32_t hlen = RANK_CELLS + nb + 2;
    for (uint32_t i = t; i < hlen + 512; i += 256) hist[i] = 0;
    for (uint32_t i = t; i <= nb; i += 256) histm[i] = 0;
    for (uint32_t i = t; i < RANK_CELLS; i += 256) cellinfo[i] = 0;
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
    const uint64_t kmin = keys[0], kmax = keys[nb - 1];
    const float dmin = key_dist(kmin);
    float scale = (float)RANK_CELLS / (key_dist(kmax) - dmin);
    if (!(scale < 1e30f)) scale = 0.f;             // one distinct distance (or inf/NaN): a single cell
    // ---- the cell table: count the keys of every cell, then an exclusive scan gives the first key of each
    for (uint32_t i = t; i < nb; i += 256) atomicAdd(&cellinfo[rank_cell(key_dist(keys[i]), dmin, scale)], 1u);
    __syncthreads();
    {
      constexpr int PER = RANK_CELLS / 256;
      uint32_t c[PER], s = 0;
#pragma unroll
      for (int e = 0; e < PER; ++e) { c[e] = cellinfo[t * PER + e]; s += c[e]; }
      uint32_t tot;
      uint32_t run = block_scan_excl(s, wsum, &tot);
#pragma unroll
      for (int e = 0; e < PER; ++e) { cellinfo[t * PER + e] = run | (c[e] << 16); run += c[e]; }
    }
    __syncthreads();
    // ---- stream the row
    uint32_t below = 0;                            // elements before the first match: all land in hist[0]
    auto visit = [&](float d, int32_t pid, int k) {
      bool ism = false;
      if (pid == qpid) {
        if (g_camids[k] == qcam) return;       // removed: same pid & same camera (rank.py:136-137)
        ism = true;
      }
      const uint64_t key = make_key(d, (uint32_t)k);
      if (key > kmax) return;                  // after every match of the batch: no rank depends on it
      if (key < kmin) {
        ++below;
        if (ism) atomicAdd(&histm[0], 1u);
        return;
      }
      const uint32_t c = rank_cell(d, dmin, scale);
      const uint32_t ci = cellinfo[c];
      uint32_t lo = ci & 0xffffu, hi = lo + (ci >> 16);   // lower_bound among the keys of this cell
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1; else hi = mid;
      }
      atomicAdd(&hist[1 + c + lo], 1u);
      if (ism) atomicAdd(&histm[lo], 1u);
    };
    // main loop: 4 independent 16-byte loads of distances (and of pids) per thread in flight before any is used
    constexpr int UNR = 4;
    int kdone = 0;
    if (vec_ok) {
      for (; kdone + UNR * 1024 <= num_g; kdone += UNR * 1024) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        f32x4 d4[UNR];
        int4 p4[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          d4[u] = __builtin_nontemporal_load((const f32x4*)(row + kdone + u * 1024 + t * 4));
          p4[u] = *(const int4*)(g_pids + kdone + u * 1024 + t * 4);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int k = kdone + u * 1024 + t * 4;
          visit(d4[u].x, p4[u].x, k);
          visit(d4[u].y, p4[u].y, k + 1);
          visit(d4[u].z, p4[u].z, k + 2);
          visit(d4[u].w, p4[u].w, k + 3);
        }
      }
    }
    for (int k = kdone + t; k < num_g; k += 256) visit(row[k], g_pids[k], k);
    if (below) atomicAdd(&hist[0], below);
    __syncthreads();
    // ---- inclusive prefix sums over hist in place: a contiguous odd-length span per thread (conflict-free)
    {
      const uint32_t per = ((hlen + 255) / 256) | 1u;
      uint32_t s = 0;
      for (uint32_t e = 0; e < per; ++e) s += hist[t * per + e];
      uint32_t tot;
      uint32_t run = block_scan_excl(s, wsum, &tot);
      for (uint32_t e = 0; e < per; ++e) { run += hist[t * per + e]; hist[t * per + e] = run; }
    }
    __syncthreads();
    // ---- matches-before-it by chunked scans of histm; AP terms in the same order as before
    uint32_t carry_m = 0;
    double part = 0.0;
    int32_t fmin = 0x7fffffff;
    for (uint32_t c0 = 0; c0 < nb; c0 += 256) {
      const uint32_t i = c0 + t;
      const uint32_t mv = i < nb ? histm[i] : 0u;
      uint32_t tmm;
      const uint32_t em = block_scan_excl(mv, wsum, &tmm);
      if (i < nb) {
        const uint32_t c = rank_cell(key_dist(keys[i]), dmin, scale);
        const uint32_t rank = hist[1 + c + i] - 1;      // 0-based position among kept (itself is counted)
        const uint32_t mrank = carry_m + em + mv - 1;   // matches before it (all batches)
        part += (double)(mrank + 1) / (double)(rank + 1);   // rank.py:156-157
        fmin = min(fmin, (int32_t)rank);
      }
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

// summary = [max_rank CMC counts, number of valid queries, bits of the AP sum].  One 256-thread workgroup; the AP
// sum has one fixed association (a contiguous span per thread, then a binary tree over the threads), so it is the
// same on every launch.
__global__ __launch_bounds__(256) void rank_finalize_kernel(const double* ap, const int32_t* first, int num_q,
                                                            int max_rank, int64_t* summary) {
  __shared__ uint32_t fh[1024];        // queries whose first match sits at rank r
  __shared__ double dsum[256];
  __shared__ uint32_t nsum[256];
  const int t = threadIdx.x;
  for (int r = t; r < max_rank; r += 256) fh[r] = 0;
  __syncthreads();
  const int per = (num_q + 255) / 256;
  double s = 0.0;
  uint32_t nv = 0;
  for (int q = t * per; q < min(num_q, (t + 1) * per); ++q) {
    const int f = first[q];
    if (f >= 0) {
      ++nv;
      s += ap[q];
      if (f < max_rank) atomicAdd(&fh[f], 1u);
    }
  }
  dsum[t] = s;
  nsum[t] = nv;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) { dsum[t] += dsum[t + o]; nsum[t] += nsum[t + o]; }
    __syncthreads();
  }
  if (t == 0) {
    int64_t run = 0;
    for (int r = 0; r < max_rank; ++r) { run += fh[r]; summary[r] = run; }   // cmc = min(cumsum,1): rank.py:145-150
    summary[max_rank] = (int64_t)nsum[0];
    summary[max_rank + 1] = __double_as_longlong(dsum[0]);
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

static int64_t split_align(int64_t bytes) { return (bytes + 255) / 256 * 256; }
static int split_terms(int64_t scheme) {
  return scheme == IEEE_SPLIT_BF16X3 ? 6 : (scheme == IEEE_SPLIT_BF16X2 || scheme == IEEE_SPLIT_F16X2) ? 3 : 0;
}

extern "C" int64_t ieee_sqeuclid_distmat_split_workspace_bytes(int64_t m, int64_t n, int64_t d, int64_t scheme) {
  const int terms = split_terms(scheme);
  if (m <= 0 || n <= 0 || d <= 0 || terms == 0) return -1;
  return split_align((m + n) * 8) + split_align(m * terms * d * 2) + split_align(n * terms * d * 2);
}

extern "C" int ieee_sqeuclid_distmat_split(const float* q, const float* g, int64_t m, int64_t n, int64_t d,
                                           int64_t scheme, int metric, float* out, int64_t ldo, void* work,
                                           int64_t work_bytes, void* stream) {
  IEEE_REQUIRE(q && g && out && work, "distmat_split: null pointer");
  IEEE_REQUIRE(m > 0 && n > 0 && d > 0, "distmat_split: empty input (m=%ld n=%ld d=%ld)", (long)m, (long)n, (long)d);
  IEEE_REQUIRE(d % 8 == 0, "distmat_split: feature dim %ld must be a multiple of 8", (long)d);
  const int terms = split_terms(scheme);
  IEEE_REQUIRE(terms != 0, "distmat_split: unknown scheme %ld (IEEE_SPLIT_BF16X3 / _BF16X2 / _F16X2)", (long)scheme);
  IEEE_REQUIRE(ldo >= n, "distmat_split: ldo < n");
  IEEE_REQUIRE(metric == 0 || metric == 1, "distmat_split: unknown metric %d", metric);
  IEEE_REQUIRE(m < (1ll << 31) && n < (1ll << 31) && terms * d < (1ll << 31), "distmat_split: too large");
  IEEE_REQUIRE(work_bytes >= ieee_sqeuclid_distmat_split_workspace_bytes(m, n, d, scheme),
               "distmat_split: workspace of %ld bytes is too small", (long)work_bytes);
  hipStream_t st = (hipStream_t)stream;
  float* qn = (float*)work;           // [m] norms, [n] norms, [m] inverse scales, [n] inverse scales
  float* gn = qn + m;
  float* qsc = gn + n;
  float* gsc = qsc + m;
  char* qs = (char*)work + split_align((m + n) * 8);
  char* gs = qs + split_align(m * terms * d * 2);
  // row norms from the fp32 rows themselves (exactly as the fp32 path)
  rownorm_kernel<float><<<cdiv(m, 4), 256, 0, st>>>(q, m, (int)d, metric, qn);
  rownorm_kernel<float><<<cdiv(n, 4), 256, 0, st>>>(g, n, (int)d, metric, gn);
  const int tiles_m = cdiv(m, 128), tiles_n = cdiv(n, 128);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)distmat_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)distmat_kernel<bf16, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    attr_done = true;
  }
  if (scheme == IEEE_SPLIT_F16X2) {
    split_rows_f16_kernel<<<cdiv(m, 4), 256, 0, st>>>(q, (uint16_t*)qs, qsc, m, (int)d, 0);
    split_rows_f16_kernel<<<cdiv(n, 4), 256, 0, st>>>(g, (uint16_t*)gs, gsc, n, (int)d, 1);
    distmat_kernel<bf16, true><<<tiles_m * tiles_n, 256, 256 * 128, st>>>((const bf16*)qs, (const bf16*)gs, qn, gn, out,
                                                                         (int)m, (int)n, (int)(terms * d), ldo, metric,
                                                                         tiles_m, tiles_n, qsc, gsc);
    return launch_status("distmat_split");
  }
  if (terms == 6) {
    split_rows_kernel<6><<<cdiv(m * (d / 8), 256), 256, 0, st>>>(q, (bf16*)qs, m, (int)d, 0);
    split_rows_kernel<6><<<cdiv(n * (d / 8), 256), 256, 0, st>>>(g, (bf16*)gs, n, (int)d, 1);
  } else {
    split_rows_kernel<3><<<cdiv(m * (d / 8), 256), 256, 0, st>>>(q, (bf16*)qs, m, (int)d, 0);
    split_rows_kernel<3><<<cdiv(n * (d / 8), 256), 256, 0, st>>>(g, (bf16*)gs, n, (int)d, 1);
  }
  distmat_kernel<bf16><<<tiles_m * tiles_n, 256, 256 * 128, st>>>((const bf16*)qs, (const bf16*)gs, qn, gn, out, (int)m,
                                                                 (int)n, (int)(terms * d), ldo, metric, tiles_m, tiles_n);
  return launch_status("distmat_split");
}

static int rank_impl(const float* distmat, int64_t ldd, int64_t num_q, int64_t num_g, const int32_t* q_pids,
                     const int32_t* g_pids, const int32_t* q_camids, const int32_t* g_camids, int64_t max_rank, double* ap,
                     int32_t* first_pos, int64_t* summary, void* work, int64_t work_bytes, void* stream) {
  IEEE_REQUIRE(distmat && q_pids && g_pids && q_camids && g_camids && ap && first_pos && summary, "rank: null pointer");
  IEEE_REQUIRE(num_q > 0 && num_g > 0, "rank: empty distmat");
  IEEE_REQUIRE(num_g < (1ll << 31) && num_q < (1ll << 31), "rank: too large");
  IEEE_REQUIRE(ldd >= num_g, "rank: ldd < num_g");
  if (max_rank > num_g) max_rank = num_g;   // rank.py:110-115
  IEEE_REQUIRE(max_rank >= 1 && max_rank <= 1024, "rank: max_rank %ld out of range [1,1024]", (long)max_rank);
  hipStream_t st = (hipStream_t)stream;
  const bool general_only = getenv("IEEE_RANK_GENERAL") && atoi(getenv("IEEE_RANK_GENERAL")) != 0;   // tests
  int32_t *start = nullptr, *idx = nullptr;
  if (work && !general_only) {
    IEEE_REQUIRE(work_bytes >= ieee_rank_workspace_bytes(num_g), "rank: workspace of %ld bytes, %ld needed", (long)work_bytes,
                 (long)ieee_rank_workspace_bytes(num_g));
    IEEE_REQUIRE(((uintptr_t)work & 3) == 0, "rank: workspace not 4-byte aligned");
    start = (int32_t*)work;
    int32_t* cursor = start + RANK_BUCKETS + 1;
    idx = cursor + RANK_BUCKETS;
    IEEE_HIP(hipMemsetAsync(start, 0, sizeof(int32_t) * (RANK_BUCKETS + 1), st));
    rank_bucket_count_kernel<<<cdiv(num_g, 256), 256, 0, st>>>(g_pids, (int)num_g, start);
    rank_bucket_scan_kernel<<<1, 1024, 0, st>>>(start, cursor);
    rank_bucket_fill_kernel<<<cdiv(num_g, 256), 256, 0, st>>>(g_pids, (int)num_g, cursor, idx);
    IEEE_TRY(launch_status("rank_bucket"));
  }
  static const int f_dbg = getenv("IEEE_RANK_DBG") ? atoi(getenv("IEEE_RANK_DBG")) : 0;    // measurement only (see the kernel)
  if (!general_only) {
#define IEEE_RQF(D_) rank_query_fast_kernel<D_><<<(int)num_q, 256, 0, st>>>(distmat, ldd, (int)num_g, q_pids, g_pids, q_camids, g_camids, ap, first_pos, start, idx)
    if (f_dbg == 1) IEEE_RQF(1); else if (f_dbg == 2) IEEE_RQF(2); else if (f_dbg == 3) IEEE_RQF(3); else IEEE_RQF(0);
#undef IEEE_RQF
  }
  rank_query_kernel<<<(int)num_q, 256, 0, st>>>(distmat, ldd, (int)num_g, q_pids, g_pids, q_camids, g_camids, ap,
                                                first_pos, general_only ? 0 : 1);
  rank_finalize_kernel<<<1, 256, 0, st>>>(ap, first_pos, (int)num_q, (int)max_rank, summary);
  return launch_status("rank_market1501");
}

extern "C" int64_t ieee_rank_workspace_bytes(int64_t num_g) {
  return (int64_t)sizeof(int32_t) * (2 * RANK_BUCKETS + 1 + (num_g > 0 ? num_g : 0));
}

extern "C" int ieee_rank_market1501(const float* distmat, int64_t ldd, int64_t num_q, int64_t num_g,
                                    const int32_t* q_pids, const int32_t* g_pids, const int32_t* q_camids,
                                    const int32_t* g_camids, int64_t max_rank, double* ap, int32_t* first_pos,
                                    int64_t* summary, void* stream) {
  return rank_impl(distmat, ldd, num_q, num_g, q_pids, g_pids, q_camids, g_camids, max_rank, ap, first_pos, summary, nullptr, 0,
                   stream);
}

extern "C" int ieee_rank_market1501_ws(const float* distmat, int64_t ldd, int64_t num_q, int64_t num_g,
                                       const int32_t* q_pids, const int32_t* g_pids, const int32_t* q_camids,
                                       const int32_t* g_camids, int64_t max_rank, double* ap, int32_t* first_pos,
                                       int64_t* summary, void* work, int64_t work_bytes, void* stream) {
  IEEE_REQUIRE(work, "rank: null workspace");
  return rank_impl(distmat, ldd, num_q, num_g, q_pids, g_pids, q_camids, g_camids, max_rank, ap, first_pos, summary, work,
                   work_bytes, stream);
}

