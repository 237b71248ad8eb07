// Train/eval BatchNorm2d (+ReLU, +residual) over NHWC maps [M = N*H*W][C], forward and backward,
// batched over `groups` (the three modality streams): pointer = base + group * stride.
// Semantics: torch.nn.BatchNorm2d defaults as used by the reference (torchreid/models/resnet.py:151,
// 164-184; ieee3modalPart.py:38): eps 1e-5, momentum 0.1, biased variance for normalisation,
// unbiased for the running estimate.  HBM-bound streaming kernels, 16 bytes per lane.
#include <math.h>
#include <stdlib.h>

#include <hip/hip_ext.h>

#include <algorithm>
#include "common.h"
#include "pool_gather.h"

namespace ieee {

#ifndef IEEE_RED_MAX_BLOCKS
#define IEEE_RED_MAX_BLOCKS 768
#endif
constexpr int RED_MAX_BLOCKS = IEEE_RED_MAX_BLOCKS;

struct RedGeom {
  int M, C;        // rows, channels
  int cprw;        // 16-byte chunks per row
  int tx, ty;      // block = tx (chunk lanes) * ty (row lanes) = 256
  int cblocks;     // column blocks
  int rblocks;     // row blocks
  int rows_per_block;
};

static RedGeom red_geom(int64_t M, int C, int vec) {
  RedGeom g;
  g.M = (int)M;
  g.C = C;
  g.cprw = C / vec;
  g.tx = g.cprw < 64 ? g.cprw : 64;
  // tx must divide 256: round down to a power of two that divides cprw
  int tx = 1;
  while (tx * 2 <= g.tx && g.cprw % (tx * 2) == 0) tx *= 2;
  g.tx = tx;
  g.ty = 256 / g.tx;
  g.cblocks = cdiv(g.cprw, g.tx);
  int rb = RED_MAX_BLOCKS / g.cblocks;
  if (rb < 1) rb = 1;
  const int maxrb = cdiv(M, g.ty * 4);   // at least ~4 rows per thread
  if (rb > maxrb) rb = maxrb;
  if (rb < 1) rb = 1;
  g.rows_per_block = cdiv(cdiv(M, rb), g.ty) * g.ty;
  g.rblocks = cdiv(M, g.rows_per_block);
  return g;
}

// Generic per-channel reduction of NQ quantities; F(values...) fills q[NQ][VEC] increments.
// partial layout: [group][rblock][NQ][C]
template <typename T, int NQ, class F>
__device__ __forceinline__ void reduce_channels(const RedGeom& g, float* partial, int64_t partial_gs, F f) {
  constexpr int VEC = 16 / sizeof(T);
  __shared__ float red[256 * 8];   // ty rows of (tx * VEC) floats, one quantity at a time
  const int t = threadIdx.x;
  const int tx = t % g.tx, ty = t / g.tx;
  const int cb = blockIdx.x % g.cblocks, rb = blockIdx.x / g.cblocks;
  const int chunk = cb * g.tx + tx;
  float acc[NQ][VEC];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[q][e] = 0.f;
  if (chunk < g.cprw) {
    const int r0 = rb * g.rows_per_block;
    const int r1 = min(g.M, r0 + g.rows_per_block);
    for (int r = r0 + ty; r < r1; r += g.ty) f((int64_t)r * g.C + chunk * VEC, chunk * VEC, acc);
  }
  // cross-row (ty) reduction through LDS, one quantity at a time to bound LDS use
  float* out = partial + blockIdx.y * partial_gs + (int64_t)rb * NQ * g.C;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < VEC; ++e) red[(ty * g.tx + tx) * VEC + e] = acc[q][e];
    __syncthreads();
    // one channel of this column block per thread (tx*VEC can exceed the block size: bf16, tx = 64)
    for (int idx = t; idx < g.tx * VEC; idx += 256) {
      const int ltx = idx / VEC, e = idx % VEC;
      float s = 0.f;
      for (int y = 0; y < g.ty; ++y) s += red[(y * g.tx + ltx) * VEC + e];
      const int c = (cb * g.tx + ltx) * VEC + e;
      if (c < g.C) out[q * g.C + c] = s;
    }
  }
}

// ---- forward statistics: sum(y), sum(y^2)
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ y, int64_t y_gs, RedGeom g,
                                                       float* partial, int64_t partial_gs) {
  constexpr int VEC = 16 / sizeof(T);
  const T* yy = y + blockIdx.y * y_gs;
  reduce_channels<T, 2>(g, partial, partial_gs, [&](int64_t off, int c0, float (*acc)[VEC]) {
    float v[VEC];
    Vec16<T>::unpack(*(const uint4*)(yy + off), v);
#pragma unroll
    for (int e = 0; e < VEC; ++e) { acc[0][e] += v[e]; acc[1][e] += v[e] * v[e]; }
  });
}

// sum the per-row-block partials of one channel: `lpc` lanes (a power of two, 32..256) stride over the row blocks,
// then an LDS tree.  block = (256/lpc) channels x lpc lanes; the totals are valid in lane 0 of each channel group.
// transposed = 0: p[rblock][2][C] (the reduction kernels of this file); 1: p[2][C][rblocks] (conv epilogues)
__device__ __forceinline__ void sum_partials(const float* p, int rblocks, int C, int c, int lpc, int transposed,
                                             double& s1, double& s2) {
  __shared__ double red[2][256];
  const int t = threadIdx.x, lane = t & (lpc - 1);
  double a1 = 0.0, a2 = 0.0;
  if (c < C) {
    if (transposed) {
      const float* p1 = p + (int64_t)c * rblocks;
      const float* p2 = p1 + (int64_t)C * rblocks;
#pragma unroll 4
      for (int r = lane; r < rblocks; r += lpc) { a1 += p1[r]; a2 += p2[r]; }
    } else {
#pragma unroll 4
      for (int r = lane; r < rblocks; r += lpc) { a1 += p[(int64_t)r * 2 * C + c]; a2 += p[(int64_t)r * 2 * C + C + c]; }
    }
  }
  red[0][t] = a1;
  red[1][t] = a2;
  for (int o = lpc >> 1; o > 0; o >>= 1) {
    __syncthreads();
    if (lane < o) { red[0][t] += red[0][t + o]; red[1][t] += red[1][t + o]; }
  }
  s1 = red[0][t]; s2 = red[1][t];
}

// lanes per channel for `rblocks` partial rows: ~4 rows per lane, between one half-wave and the whole block
static int finalize_lpc(int64_t rblocks) {
  int lpc = 32;
  while (lpc < 256 && lpc * 4 < rblocks) lpc *= 2;
  return lpc;
}

// mean / invstd / scale / shift + running-stat update (train) or scale/shift from running stats (eval)
// stats layout per group: [4][C] = mean, invstd, scale, shift.   launch: block 256 = (256/lpc) channels x lpc lanes
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* partial, int64_t partial_gs, int rblocks, int M,
                                                          int C, const float* gamma, const float* beta,
                                                          int64_t param_gs, float* running_mean, float* running_var,
                                                          int64_t buf_gs, float* stats, int64_t stats_gs,
                                                          float momentum, float eps, int training, int transposed,
                                                          int lpc) {
  const int lane = threadIdx.x & (lpc - 1);
  const int c = blockIdx.x * (256 / lpc) + threadIdx.x / lpc;
  const int z = blockIdx.y;
  // (per-channel operands fetched before the reduction: see bn_bwd_finalize_kernel)
  const bool owner = c < C && lane == 0;
  float ga = 0.f, be = 0.f, rm0 = 0.f, rv0 = 0.f;
  if (owner) {
    ga = gamma[z * param_gs + c]; be = beta[z * param_gs + c];
    if (training && running_mean != nullptr) { rm0 = running_mean[z * buf_gs + c]; rv0 = running_var[z * buf_gs + c]; }
  }
  double s1 = 0.0, s2 = 0.0;
  if (training) sum_partials(partial + z * partial_gs, rblocks, C, c, lpc, transposed, s1, s2);
  if (!owner) return;
  float* st = stats + z * stats_gs;
  float mean, invstd;
  if (training) {
    const double mu = s1 / M;
    double var = s2 / M - mu * mu;
    if (var < 0) var = 0;
    mean = (float)mu;
    invstd = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean != nullptr) {
      float* rm = running_mean + z * buf_gs + c;
      float* rv = running_var + z * buf_gs + c;
      const double unbiased = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
      *rm = (1.f - momentum) * rm0 + momentum * mean;
      *rv = (1.f - momentum) * rv0 + momentum * (float)unbiased;
    }
  } else {
    mean = running_mean[z * buf_gs + c];
    invstd = 1.0f / sqrtf(running_var[z * buf_gs + c] + eps);
  }
  st[c] = mean;
  st[C + c] = invstd;
  const float sc = ga * invstd;
  st[2 * C + c] = sc;
  st[3 * C + c] = be - mean * sc;
}

// a = [relu]( y*scale + shift [+ residual] )
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ y, const T* __restrict__ residual,
                                                       T* __restrict__ out, const float* __restrict__ stats,
                                                       int64_t stats_gs, int64_t total_chunks, int cprw, int C,
                                                       int64_t gs, int relu, uint8_t* __restrict__ relu_bits) {
  constexpr int VEC = 16 / sizeof(T);
  const int z = blockIdx.y;
  // relu_bits (16-bit dtype only, VEC == 8): one byte per 16-byte chunk, bit e = [stored out[e] > 0] -- the ReLU mask the
  // block-input dgrad needs later, at 1/16 of the bytes of reading `out` again
  uint8_t* bb = relu_bits ? relu_bits + z * (gs / 8) : nullptr;
  const bool pow2 = (cprw & (cprw - 1)) == 0;
  const float* sc = stats + z * stats_gs + 2 * C;
  const float* sh = sc + C;
  const T* yy = y + z * gs;
  const T* rr = residual ? residual + z * gs : nullptr;
  T* oo = out + z * gs;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_chunks; i += (int64_t)gridDim.x * blockDim.x) {
    const int c0 = (pow2 ? (int)(i & (cprw - 1)) : (int)(i % cprw)) * VEC;   // 64-bit modulo only when needed
    float v[VEC], r[VEC];
    Vec16<T>::unpack(*(const uint4*)(yy + i * VEC), v);
    if (rr) Vec16<T>::unpack(*(const uint4*)(rr + i * VEC), r);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float x = v[e] * sc[c0 + e] + sh[c0 + e];
      if (rr) x += r[e];
      if (relu) x = fmaxf(x, 0.f);
      v[e] = x;
    }
    const uint4 pv = Vec16<T>::pack(v);
    *(uint4*)(oo + i * VEC) = pv;
    if constexpr (VEC == 8) {
      if (bb) {
        Vec16<T>::unpack(pv, v);   // the mask of the ROUNDED stored values, as a reader of `out` would see it
        unsigned b = 0;
#pragma unroll
        for (int e = 0; e < VEC; ++e) b |= (v[e] > 0.f ? 1u : 0u) << e;
        bb[i] = (uint8_t)b;
      }
    }
  }
}

// VEC (4 or 8) consecutive floats from a 16-byte aligned address, as 16-byte loads
template <int VEC> __device__ __forceinline__ void load_floats(const float* __restrict__ p, float* dst) {
#pragma unroll
  for (int e = 0; e < VEC; e += 4) {
    const float4 v = *(const float4*)(p + e);
    dst[e] = v.x; dst[e + 1] = v.y; dst[e + 2] = v.z; dst[e + 3] = v.w;
  }
}

// The same pass with the per-channel constants in registers (round 4).  When the chunks of a row divide the block
// (256 % cprw == 0: every channel count of this network), chunk i = blockIdx.x*256 + t + k*gridDim.x*256 lies in channel
// chunk t % cprw for every k, so scale / shift are fetched ONCE per thread instead of two 32-byte loads per 16 bytes of
// y; two chunks per thread are in flight per iteration (all their loads are issued before the first use).
template <typename T, bool RES, bool BITS, int UNROLL>
__global__ __launch_bounds__(256) void bn_apply_fixed_kernel(const T* __restrict__ y, const T* __restrict__ residual,
                                                             T* __restrict__ out, const float* __restrict__ stats,
                                                             int64_t stats_gs, int64_t total_chunks, int cprw, int C,
                                                             int64_t gs, int relu, uint8_t* __restrict__ relu_bits) {
  constexpr int VEC = 16 / sizeof(T);
  const int z = blockIdx.y;
  uint8_t* bb = BITS ? relu_bits + z * (gs / 8) : nullptr;
  const int c0 = (threadIdx.x & (cprw - 1)) * VEC;
  const float* scp = stats + z * stats_gs + 2 * C + c0;
  float sc[VEC], sh[VEC];
  load_floats<VEC>(scp, sc);
  load_floats<VEC>(scp + C, sh);
  const T* yy = y + z * gs;
  const T* rr = RES ? residual + z * gs : nullptr;
  T* oo = out + z * gs;
  const int64_t stride = (int64_t)gridDim.x * 256;
  auto finish = [&](int64_t i, const uint4& yv, const uint4& rv) {
    float v[VEC], r[VEC];
    Vec16<T>::unpack(yv, v);
    if (RES) Vec16<T>::unpack(rv, r);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float x = v[e] * sc[e] + sh[e];
      if (RES) x += r[e];
      if (relu) x = fmaxf(x, 0.f);
      v[e] = x;
    }
    const uint4 pv = Vec16<T>::pack(v);
    *(uint4*)(oo + i * VEC) = pv;
    if constexpr (BITS && VEC == 8) {
      Vec16<T>::unpack(pv, v);   // the mask of the ROUNDED stored values, as a reader of `out` would see it
      unsigned b = 0;
#pragma unroll
      for (int e = 0; e < VEC; ++e) b |= (v[e] > 0.f ? 1u : 0u) << e;
      bb[i] = (uint8_t)b;
    }
  };
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (UNROLL == 2)
    for (; i + stride < total_chunks; i += 2 * stride) {
      const uint4 y0 = *(const uint4*)(yy + i * VEC);
      const uint4 y1 = *(const uint4*)(yy + (i + stride) * VEC);
      uint4 r0 = y0, r1 = y1;
      if (RES) { r0 = *(const uint4*)(rr + i * VEC); r1 = *(const uint4*)(rr + (i + stride) * VEC); }
      finish(i, y0, r0);
      finish(i + stride, y1, r1);
    }
  for (; i < total_chunks; i += stride) {
    const uint4 y0 = *(const uint4*)(yy + i * VEC);
    uint4 r0 = y0;
    if (RES) r0 = *(const uint4*)(rr + i * VEC);
    finish(i, y0, r0);
  }
}

// ---- backward reductions: s1 = sum(g), s2 = sum(g*y) with g = da * [a > 0] (mask optional)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ da, const T* __restrict__ a,
                                                            const T* __restrict__ y, int64_t gs, RedGeom g,
                                                            float* partial, int64_t partial_gs,
                                                            const float* __restrict__ stats, int64_t stats_gs,
                                                            int mask_from_y) {
  constexpr int VEC = 16 / sizeof(T);
  const T* dd = da + blockIdx.y * gs;
  const T* aa = a ? a + blockIdx.y * gs : nullptr;
  const T* yy = y + blockIdx.y * gs;
  const float* sc = stats + blockIdx.y * stats_gs + 2 * g.C;
  const float* sh = sc + g.C;
  reduce_channels<T, 2>(g, partial, partial_gs, [&](int64_t off, int c0, float (*acc)[VEC]) {
    float d[VEC], m[VEC], v[VEC];
    Vec16<T>::unpack(*(const uint4*)(dd + off), d);
    Vec16<T>::unpack(*(const uint4*)(yy + off), v);
    if (aa) {
      Vec16<T>::unpack(*(const uint4*)(aa + off), m);
#pragma unroll
      for (int e = 0; e < VEC; ++e) d[e] = m[e] > 0.f ? d[e] : 0.f;
    } else if (mask_from_y) {   // out = relu(y*scale+shift), so [out > 0] needs no extra tensor
#pragma unroll
      for (int e = 0; e < VEC; ++e) d[e] = (v[e] * sc[c0 + e] + sh[c0 + e]) > 0.f ? d[e] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) { acc[0][e] += d[e]; acc[1][e] += d[e] * v[e]; }
  });
}

// dgamma = sum(g*xhat), dbeta = sum(g);  coefficients of dy = k1*g + k2*y + k3
// coef layout per group: [3][C].   launch: block 256 = (256/lpc) channels x lpc lanes
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* partial, int64_t partial_gs, int rblocks,
                                                              int M, int C, const float* gamma, int64_t param_gs,
                                                              const float* stats, int64_t stats_gs, float* dgamma,
                                                              float* dbeta, int64_t grad_gs, float* coef,
                                                              int64_t coef_gs, int accumulate, int transposed,
                                                              int lpc) {
  const int lane = threadIdx.x & (lpc - 1);
  const int c = blockIdx.x * (256 / lpc) + threadIdx.x / lpc;
  const int z = blockIdx.y;
  // the per-channel operands do not depend on the sums: fetched BEFORE the reduction, so the kernel is two dependent memory
  // round trips instead of three (each one costs 2-3 us beside the weight-gradient stream; 55 of these launches sit on the
  // dependent chain of a step)
  const float* st = stats + z * stats_gs;
  const bool owner = c < C && lane == 0;
  float pre_mean = 0.f, pre_invstd = 0.f, pre_gamma = 0.f, pre_dg = 0.f, pre_db = 0.f;
  if (owner) {
    pre_mean = st[c]; pre_invstd = st[C + c]; pre_gamma = gamma[z * param_gs + c];
    if (dgamma && accumulate) { pre_dg = dgamma[z * grad_gs + c]; pre_db = dbeta[z * grad_gs + c]; }
  }
  double s1, s2;
  sum_partials(partial + z * partial_gs, rblocks, C, c, lpc, transposed, s1, s2);
  if (!owner) return;
  const double mean = pre_mean, invstd = pre_invstd;
  const double sgx = invstd * (s2 - mean * s1);   // sum g * xhat
  if (dgamma) {
    float* dg = dgamma + z * grad_gs + c;
    float* db = dbeta + z * grad_gs + c;
    *dg = accumulate ? pre_dg + (float)sgx : (float)sgx;
    *db = accumulate ? pre_db + (float)s1 : (float)s1;
  }
  const double A = (double)pre_gamma * invstd;
  const double c1 = s1 / M, c2 = sgx / M;
  float* k = coef + z * coef_gs;
  k[c] = (float)A;
  k[C + c] = (float)(-A * invstd * c2);
  k[2 * C + c] = (float)(-A * c1 + A * invstd * c2 * mean);
}

// dy = k1*g + k2*y + k3, g = da*[a>0]; optionally also writes g (the identity-branch gradient)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ da, const T* __restrict__ a,
                                                           const T* __restrict__ y, T* __restrict__ dy,
                                                           T* __restrict__ gout, const float* __restrict__ coef,
                                                           int64_t coef_gs, int64_t total_chunks, int cprw, int C,
                                                           int64_t gs, const float* __restrict__ stats,
                                                           int64_t stats_gs, int mask_from_y) {
  constexpr int VEC = 16 / sizeof(T);
  const int z = blockIdx.y;
  const bool pow2 = (cprw & (cprw - 1)) == 0;
  const float* sc = stats + z * stats_gs + 2 * C;
  const float* sh = sc + C;
  const float* k1 = coef + z * coef_gs;
  const float* k2 = k1 + C;
  const float* k3 = k2 + C;
  const T* dd = da + z * gs;
  const T* aa = a ? a + z * gs : nullptr;
  const T* yy = y + z * gs;
  T* oo = dy + z * gs;
  T* go = gout ? gout + z * gs : nullptr;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_chunks; i += (int64_t)gridDim.x * blockDim.x) {
    const int c0 = (pow2 ? (int)(i & (cprw - 1)) : (int)(i % cprw)) * VEC;   // 64-bit modulo only when needed
    float d[VEC], m[VEC], v[VEC];
    Vec16<T>::unpack(*(const uint4*)(dd + i * VEC), d);
    Vec16<T>::unpack(*(const uint4*)(yy + i * VEC), v);
    if (aa) {
      Vec16<T>::unpack(*(const uint4*)(aa + i * VEC), m);
#pragma unroll
      for (int e = 0; e < VEC; ++e) d[e] = m[e] > 0.f ? d[e] : 0.f;
    } else if (mask_from_y) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) d[e] = (v[e] * sc[c0 + e] + sh[c0 + e]) > 0.f ? d[e] : 0.f;
    }
    if (go) *(uint4*)(go + i * VEC) = Vec16<T>::pack(d);
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = k1[c0 + e] * d[e] + k2[c0 + e] * v[e] + k3[c0 + e];
    *(uint4*)(oo + i * VEC) = Vec16<T>::pack(v);
  }
}

// dy = k1*g + k2*y + k3 with the per-channel constants in registers (see bn_apply_fixed_kernel): the plain kernel
// fetches 3 (5 with the mask from y) x 32 bytes of coefficients per 16 bytes of each operand.  MASK: 0 none (g arrives
// masked), 1 from the activation tensor `a`, 2 recomputed from y.  UNROLL: chunks in flight per thread (1 or 2).
template <typename T, int MASK, bool GOUT, int UNROLL>
__global__ __launch_bounds__(256) void bn_bwd_apply_fixed_kernel(const T* __restrict__ da, const T* __restrict__ a,
                                                                 const T* __restrict__ y, T* __restrict__ dy,
                                                                 T* __restrict__ gout, const float* __restrict__ coef,
                                                                 int64_t coef_gs, int64_t total_chunks, int cprw, int C,
                                                                 int64_t gs, const float* __restrict__ stats,
                                                                 int64_t stats_gs) {
  constexpr int VEC = 16 / sizeof(T);
  const int z = blockIdx.y;
  const int c0 = (threadIdx.x & (cprw - 1)) * VEC;
  float k1[VEC], k2[VEC], k3[VEC], sc[VEC], sh[VEC];
  {
    const float* kp = coef + z * coef_gs + c0;
    load_floats<VEC>(kp, k1);
    load_floats<VEC>(kp + C, k2);
    load_floats<VEC>(kp + 2 * C, k3);
    if (MASK == 2) {
      const float* sp = stats + z * stats_gs + 2 * C + c0;
      load_floats<VEC>(sp, sc);
      load_floats<VEC>(sp + C, sh);
    }
  }
  const T* dd = da + z * gs;
  const T* aa = MASK == 1 ? a + z * gs : nullptr;
  const T* yy = y + z * gs;
  T* oo = dy + z * gs;
  T* go = GOUT ? gout + z * gs : nullptr;
  const int64_t stride = (int64_t)gridDim.x * 256;
  auto finish = [&](int64_t i, const uint4& dv, const uint4& yv, const uint4& av) {
    float d[VEC], v[VEC], m[VEC];
    Vec16<T>::unpack(dv, d);
    Vec16<T>::unpack(yv, v);
    if (MASK == 1) {
      Vec16<T>::unpack(av, m);
#pragma unroll
      for (int e = 0; e < VEC; ++e) d[e] = m[e] > 0.f ? d[e] : 0.f;
    } else if (MASK == 2) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) d[e] = (v[e] * sc[e] + sh[e]) > 0.f ? d[e] : 0.f;
    }
    if (GOUT) *(uint4*)(go + i * VEC) = Vec16<T>::pack(d);
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = k1[e] * d[e] + k2[e] * v[e] + k3[e];
    *(uint4*)(oo + i * VEC) = Vec16<T>::pack(v);
  };
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (UNROLL == 2)
    for (; i + stride < total_chunks; i += 2 * stride) {
      const int64_t j = i + stride;
      const uint4 d0 = *(const uint4*)(dd + i * VEC), d1 = *(const uint4*)(dd + j * VEC);
      const uint4 y0 = *(const uint4*)(yy + i * VEC), y1 = *(const uint4*)(yy + j * VEC);
      uint4 a0 = y0, a1 = y1;
      if (MASK == 1) { a0 = *(const uint4*)(aa + i * VEC); a1 = *(const uint4*)(aa + j * VEC); }
      finish(i, d0, y0, a0);
      finish(j, d1, y1, a1);
    }
  for (; i < total_chunks; i += stride) {
    const uint4 d0 = *(const uint4*)(dd + i * VEC);
    const uint4 y0 = *(const uint4*)(yy + i * VEC);
    uint4 a0 = y0;
    if (MASK == 1) a0 = *(const uint4*)(aa + i * VEC);
    finish(i, d0, y0, a0);
  }
}

// ---- BatchNorm passes that take their statistics from fixed-point TOTALS (conv.hip: tl_totals) and derive the per-channel
// coefficients in their own prologue: no finalize launch.  Every workgroup runs the prologue -- the same arithmetic on the
// same integers, so the same bits -- and workgroup 0 publishes what later kernels read (stats, running statistics,
// d(gamma), d(beta)).  What the prologue may cost decides whether the path pays (B = 64 step, layer3 + layer4 on it):
//   every thread converts its own 8 channels, 1.0 / sqrt and / M in double, stores inside the channel loop   19.8 ms
//   no double division / square root (E[y^2] - mean^2 stays in double, the reciprocal square root is float)  19.8 ms
//   all loads issued before the first store of the publishing threads (the compiler kept eight load -> wait
//   -> store rounds in order; a 20 us pass took 140 us)                                                       15.4 ms
//   one channel per thread through LDS (an eighth of the int64 -> double conversions), at most 1 024
//   workgroups per modality so that a thread walks >= 4 chunks behind one prologue                            14.69 ms
// against 14.83 ms with the finalize launches.
constexpr double TOT_INV_FWD = 1.0 / 16777216.0;          // 2^-24
constexpr double TOT_INV_BWD = 1.0 / 1099511627776.0;     // 2^-40
// a total beyond HALF the range the conv kernels guarantee (+-2^62) is reported in overflow[2] (forward) / [3] (backward):
// the next doubling of the activations would saturate tiles
constexpr long long TOT_HALF_RANGE = 1ll << 61;
__device__ __forceinline__ long long tot_abs(long long v) { return v < 0 ? -v : v; }

template <typename T, bool RES, bool BITS>
__global__ __launch_bounds__(256) void bn_apply_totals_kernel(const T* __restrict__ y, const T* __restrict__ residual,
                                                              T* __restrict__ out, const long long* __restrict__ totals,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              int64_t param_gs, float* running_mean, float* running_var,
                                                              int64_t buf_gs, float* __restrict__ stats, int M, float momentum,
                                                              float eps, int64_t total_chunks, int cprw, int C, int64_t gs,
                                                              int relu, uint8_t* __restrict__ relu_bits, double inv_m,
                                                              float unbias, int rep, int64_t rep_stride, int* overflow) {
  constexpr int VEC = 16 / sizeof(T);
  const int z = blockIdx.y;
  uint8_t* bb = BITS ? relu_bits + z * (gs / 8) : nullptr;
  const int c0 = (threadIdx.x & (cprw - 1)) * VEC;
  // the first chunk's operands are requested BEFORE the prologue: their HBM latency runs beside the totals' round trip
  const T* yy = y + z * gs;
  const T* rr = RES ? residual + z * gs : nullptr;
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  uint4 ynext = make_uint4(0, 0, 0, 0), rnext = make_uint4(0, 0, 0, 0);
  if (out != nullptr && i < total_chunks) {
    ynext = *(const uint4*)(yy + i * VEC);
    if (RES) rnext = *(const uint4*)(rr + i * VEC);
  }
  // one channel per thread (C / 256 rounds), through LDS: an eighth of the conversions of "every thread its 8 channels"
  extern __shared__ float tot_lds[];
  float* s_sc = tot_lds;
  float* s_sh = tot_lds + C;
  for (int c = threadIdx.x; c < C; c += 256) {
    long long t1 = 0, t2 = 0;
    for (int r = 0; r < rep; ++r) {                          // replicas: the conv spread its adders over `rep` copies
      t1 += totals[r * rep_stride + (int64_t)z * 2 * C + c];
      t2 += totals[r * rep_stride + (int64_t)z * 2 * C + C + c];
    }
    // (the conv clamps every tile to its share of +-2^62, so these sums cannot have wrapped: conv.hip, tl_totals_flag)
    if (blockIdx.x == 0 && overflow != nullptr && (tot_abs(t1) > TOT_HALF_RANGE || tot_abs(t2) > TOT_HALF_RANGE)) *(volatile int*)(overflow + 2) = 1;
    const float ga = gamma[z * param_gs + c], be = beta[z * param_gs + c];
    const double mu = (double)t1 * inv_m;                    // inv_m = 2^-24 / M
    double var = (double)t2 * inv_m - mu * mu;
    if (var < 0) var = 0;
    const float mean = (float)mu, varf = (float)var, invstd = rsqrtf(varf + eps);
    const float scv = ga * invstd, shv = be - mean * scv;
    s_sc[c] = scv; s_sh[c] = shv;
    if (blockIdx.x == 0) {
      float* st = stats + (int64_t)z * 4 * C;
      st[c] = mean; st[C + c] = invstd; st[2 * C + c] = scv; st[3 * C + c] = shv;
      if (running_mean != nullptr) {
        float* rm = running_mean + z * buf_gs + c;
        float* rv = running_var + z * buf_gs + c;
        *rm = (1.f - momentum) * *rm + momentum * mean;
        *rv = (1.f - momentum) * *rv + momentum * (varf * unbias);             // unbias = M / (M - 1)
      }
    }
  }
  __syncthreads();
  float sc[VEC], sh[VEC];
  load_floats<VEC>(s_sc + c0, sc);
  load_floats<VEC>(s_sh + c0, sh);
  if (out == nullptr) return;
  T* oo = out + z * gs;
  for (; i < total_chunks; i += stride) {
    float v[VEC], r[VEC];
    Vec16<T>::unpack(ynext, v);
    if (RES) Vec16<T>::unpack(rnext, r);
    if (i + stride < total_chunks) {                          // the next chunk is in flight while this one is finished
      ynext = *(const uint4*)(yy + (i + stride) * VEC);
      if (RES) rnext = *(const uint4*)(rr + (i + stride) * VEC);
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float x = v[e] * sc[e] + sh[e];
      if (RES) x += r[e];
      if (relu) x = fmaxf(x, 0.f);
      v[e] = x;
    }
    const uint4 pv = Vec16<T>::pack(v);
    *(uint4*)(oo + i * VEC) = pv;
    if constexpr (BITS && VEC == 8) {
      Vec16<T>::unpack(pv, v);
      unsigned b = 0;
#pragma unroll
      for (int e = 0; e < VEC; ++e) b |= (v[e] > 0.f ? 1u : 0u) << e;
      bb[i] = (uint8_t)b;
    }
  }
}

// backward: totals = sum g, sum g*y (2^40 fixed point); MASK as in bn_bwd_apply_fixed_kernel
template <typename T, int MASK, bool GOUT>
__global__ __launch_bounds__(256) void bn_bwd_apply_totals_kernel(const T* __restrict__ da, const T* __restrict__ a,
                                                                  const T* __restrict__ y, T* __restrict__ dy,
                                                                  T* __restrict__ gout, const long long* __restrict__ totals,
                                                                  const float* __restrict__ gamma, int64_t param_gs,
                                                                  const float* __restrict__ stats, float* dgamma,
                                                                  float* dbeta, int64_t grad_gs, int M, int64_t total_chunks,
                                                                  int cprw, int C, int64_t gs, double inv_m, int rep,
                                                                  int64_t rep_stride, int* overflow) {
  constexpr int VEC = 16 / sizeof(T);
  const int z = blockIdx.y;
  const int c0 = (threadIdx.x & (cprw - 1)) * VEC;
  const float* st = stats + (int64_t)z * 4 * C;
  // the first chunk's operands are requested BEFORE the prologue (see bn_apply_totals_kernel)
  const T* dd = da + z * gs;
  const T* aa = MASK == 1 ? a + z * gs : nullptr;
  const T* yy = y + z * gs;
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  uint4 dnext = make_uint4(0, 0, 0, 0), ynext = dnext, anext = dnext;
  if (i < total_chunks) {
    dnext = *(const uint4*)(dd + i * VEC);
    ynext = *(const uint4*)(yy + i * VEC);
    if (MASK == 1) anext = *(const uint4*)(aa + i * VEC);
  }
  extern __shared__ float tot_lds[];
  float* s_k = tot_lds;                                      // [3][C]
  for (int c = threadIdx.x; c < C; c += 256) {
    long long t1 = 0, t2 = 0;
    for (int r = 0; r < rep; ++r) {
      t1 += totals[r * rep_stride + (int64_t)z * 2 * C + c];
      t2 += totals[r * rep_stride + (int64_t)z * 2 * C + C + c];
    }
    if (blockIdx.x == 0 && overflow != nullptr && (tot_abs(t1) > TOT_HALF_RANGE || tot_abs(t2) > TOT_HALF_RANGE)) *(volatile int*)(overflow + 3) = 1;
    const double s1 = (double)t1 * TOT_INV_BWD, s2 = (double)t2 * TOT_INV_BWD;
    const double mean = st[c], invstd = st[C + c];
    const double sgx = invstd * (s2 - mean * s1);            // sum g * xhat
    const double A = (double)gamma[z * param_gs + c] * invstd;
    const double c1 = s1 * inv_m, c2 = sgx * inv_m;          // inv_m = 1 / M
    s_k[c] = (float)A;
    s_k[C + c] = (float)(-A * invstd * c2);
    s_k[2 * C + c] = (float)(-A * c1 + A * invstd * c2 * mean);
    if (blockIdx.x == 0 && dgamma != nullptr) {
      dgamma[z * grad_gs + c] = (float)sgx;
      dbeta[z * grad_gs + c] = (float)s1;
    }
  }
  __syncthreads();
  float k1[VEC], k2[VEC], k3[VEC], sc[VEC], sh[VEC];
  load_floats<VEC>(s_k + c0, k1);
  load_floats<VEC>(s_k + C + c0, k2);
  load_floats<VEC>(s_k + 2 * C + c0, k3);
  if (MASK == 2) { load_floats<VEC>(st + 2 * C + c0, sc); load_floats<VEC>(st + 3 * C + c0, sh); }
  T* oo = dy + z * gs;
  T* go = GOUT ? gout + z * gs : nullptr;
  for (; i < total_chunks; i += stride) {
    float d[VEC], v[VEC], m[VEC];
    Vec16<T>::unpack(dnext, d);
    Vec16<T>::unpack(ynext, v);
    if (MASK == 1) Vec16<T>::unpack(anext, m);
    if (i + stride < total_chunks) {                          // the next chunk is in flight while this one is finished
      dnext = *(const uint4*)(dd + (i + stride) * VEC);
      ynext = *(const uint4*)(yy + (i + stride) * VEC);
      if (MASK == 1) anext = *(const uint4*)(aa + (i + stride) * VEC);
    }
    if (MASK == 1) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) d[e] = m[e] > 0.f ? d[e] : 0.f;
    } else if (MASK == 2) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) d[e] = (v[e] * sc[e] + sh[e]) > 0.f ? d[e] : 0.f;
    }
    if (GOUT) *(uint4*)(go + i * VEC) = Vec16<T>::pack(d);
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = k1[e] * d[e] + k2[e] * v[e] + k3[e];
    *(uint4*)(oo + i * VEC) = Vec16<T>::pack(v);
  }
}

// The same pass for the LAST BatchNorm of a bottleneck block that has a downsample branch (out = relu(bn3(y3) + bn_ds(y_ds)):
// both BatchNorm backwards are fed by the same masked gradient g).  While it walks g for dy3 it also reads the branch's conv
// output y2 and leaves the branch's backward sums -- sum g (= this unit's own total) and sum g*y2 -- in the branch unit's
// fixed-point totals, so that the branch's BatchNorm backward is ONE apply launch with its coefficients from a prologue: its
// reduction pass (g and y2 read once more, ~67 us at B = 64) and its finalize launch disappear; this pass pays one more
// operand stream.  A thread's channel chunk is fixed along its walk, so it keeps 8 running sums; the workgroup adds them up
// through LDS and issues one no-return atomic per channel (workgroup w to replica w % rep2).  da = g (already masked by the
// dgrad that wrote it), no g output: the only form the executor needs.
__device__ __forceinline__ long long tot_to_fixed_bwd(float v, float lim, int* flag) {
  float x = v * 1099511627776.0f;                // 2^40
  if (!(fabsf(x) <= lim)) {                      // beyond this workgroup's share of +-2^62, or NaN (see conv.hip: to_fixed)
    if (flag != nullptr) *(volatile int*)flag = 1;
    x = fminf(fmaxf(x, -lim), lim);
  }
  return __float2ll_rn(x);
}
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_totals_ds_kernel(const T* __restrict__ da, const T* __restrict__ y,
                                                                     const T* __restrict__ y2, T* __restrict__ dy,
                                                                     const long long* __restrict__ totals,
                                                                     const float* __restrict__ gamma, int64_t param_gs,
                                                                     const float* __restrict__ stats, float* dgamma,
                                                                     float* dbeta, int64_t grad_gs, int M, int64_t total_chunks,
                                                                     int cprw, int C, int64_t gs, double inv_m, int rep,
                                                                     int64_t rep_stride, long long* __restrict__ totals2, int rep2,
                                                                     int64_t rep2_stride, float lim2, int* overflow) {
  constexpr int VEC = 16 / sizeof(T);
  const int z = blockIdx.y;
  const int c0 = (threadIdx.x & (cprw - 1)) * VEC;
  const float* st = stats + (int64_t)z * 4 * C;
  const T* dd = da + z * gs;
  const T* yy = y + z * gs;
  const T* y2p = y2 + z * gs;
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  uint4 dnext = make_uint4(0, 0, 0, 0), ynext = dnext, y2next = dnext;
  if (i < total_chunks) {
    dnext = *(const uint4*)(dd + i * VEC);
    ynext = *(const uint4*)(yy + i * VEC);
    y2next = *(const uint4*)(y2p + i * VEC);
  }
  extern __shared__ float tot_lds[];
  float* s_k = tot_lds;                                      // [3][C]; reused as [VEC][256] reduction planes at the end
  long long* t2 = totals2 + (int64_t)z * 2 * C;
  for (int c = threadIdx.x; c < C; c += 256) {
    long long t1 = 0, tq = 0;
    for (int r = 0; r < rep; ++r) {
      t1 += totals[r * rep_stride + (int64_t)z * 2 * C + c];
      tq += totals[r * rep_stride + (int64_t)z * 2 * C + C + c];
    }
    if (blockIdx.x == 0) {
      if (overflow != nullptr && (tot_abs(t1) > TOT_HALF_RANGE || tot_abs(tq) > TOT_HALF_RANGE)) *(volatile int*)(overflow + 3) = 1;
      // sum g of the branch's BatchNorm is this unit's own (same gradient, same channels): handed over as it is
      (void)__hip_atomic_fetch_add(t2 + c, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const double s1 = (double)t1 * TOT_INV_BWD, s2 = (double)tq * TOT_INV_BWD;
    const double mean = st[c], invstd = st[C + c];
    const double sgx = invstd * (s2 - mean * s1);
    const double A = (double)gamma[z * param_gs + c] * invstd;
    const double c1 = s1 * inv_m, c2 = sgx * inv_m;
    s_k[c] = (float)A;
    s_k[C + c] = (float)(-A * invstd * c2);
    s_k[2 * C + c] = (float)(-A * c1 + A * invstd * c2 * mean);
    if (blockIdx.x == 0 && dgamma != nullptr) {
      dgamma[z * grad_gs + c] = (float)sgx;
      dbeta[z * grad_gs + c] = (float)s1;
    }
  }
  __syncthreads();
  float k1[VEC], k2[VEC], k3[VEC], s3[VEC];
  load_floats<VEC>(s_k + c0, k1);
  load_floats<VEC>(s_k + C + c0, k2);
  load_floats<VEC>(s_k + 2 * C + c0, k3);
#pragma unroll
  for (int e = 0; e < VEC; ++e) s3[e] = 0.f;
  T* oo = dy + z * gs;
  for (; i < total_chunks; i += stride) {
    float d[VEC], v[VEC], w[VEC];
    Vec16<T>::unpack(dnext, d);
    Vec16<T>::unpack(ynext, v);
    Vec16<T>::unpack(y2next, w);
    if (i + stride < total_chunks) {
      dnext = *(const uint4*)(dd + (i + stride) * VEC);
      ynext = *(const uint4*)(yy + (i + stride) * VEC);
      y2next = *(const uint4*)(y2p + (i + stride) * VEC);
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) { s3[e] += d[e] * w[e]; v[e] = k1[e] * d[e] + k2[e] * v[e] + k3[e]; }
    *(uint4*)(oo + i * VEC) = Vec16<T>::pack(v);
  }
  __syncthreads();                                           // every thread has its coefficients: the planes may be reused
  float* red = tot_lds;                                      // [VEC][256]: lane-contiguous stores
#pragma unroll
  for (int e = 0; e < VEC; ++e) red[e * 256 + threadIdx.x] = s3[e];
  __syncthreads();
  const int rpp = 256 / cprw;
  long long* dst = t2 + (int64_t)(blockIdx.x % rep2) * rep2_stride + C;
  for (int c = threadIdx.x; c < C; c += 256) {
    const int cc = c / VEC, e = c % VEC;
    float s = 0.f;
    for (int r = 0; r < rpp; ++r) s += red[e * 256 + r * cprw + cc];
    (void)__hip_atomic_fetch_add(dst + c, tot_to_fixed_bwd(s, lim2, overflow != nullptr ? overflow + 1 : nullptr),
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

struct PoolShifts { int pow2, lw, lh, lc, lq; };   // log2 of Wi, Hi, C, chunks per row when all are powers of two

// ---- the stem's backward in two passes: d(out) of its ReLU(BatchNorm(y)) is the backward of MaxPool2d(3,2,1) applied
// to dpool, gathered on the fly (pool_gather.h) instead of materialised; g = that * [y*scale+shift > 0].
// Pass 1: per-channel sums (sum g, sum g*y); pass 2: dy = k1*g + k2*y + k3.  Replaces maxpool_bwd + bn_bwd_reduce +
// bn_bwd_apply (a full-resolution write and two full-resolution reads less, on the last stretch of the step)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_pooled_reduce_kernel(const T* __restrict__ dpool,
                                                                   const uint8_t* __restrict__ arg,
                                                                   const T* __restrict__ y, int64_t y_gs, int64_t p_gs,
                                                                   RedGeom g, int Hi, int Wi, int Ho, int Wo,
                                                                   float* partial, int64_t partial_gs,
                                                                   const float* __restrict__ stats, int64_t stats_gs,
                                                                   PoolShifts ps) {
  constexpr int VEC = 16 / sizeof(T);
  const T* dd = dpool + blockIdx.y * p_gs;
  const uint8_t* aa = arg + blockIdx.y * p_gs;
  const T* yy = y + blockIdx.y * y_gs;
  const float* sc = stats + blockIdx.y * stats_gs + 2 * g.C;
  const float* sh = sc + g.C;
  reduce_channels<T, 2>(g, partial, partial_gs, [&](int64_t off, int c0, float (*acc)[VEC]) {
    // (fewer than 2^31 elements per group: 32-bit index math; shifts when every extent is a power of two -- the five
    // 64- and 32-bit divisions per 16 bytes made these two kernels VALU-bound at 2.5-2.8 TB/s)
    int row, w, h, b;
    if (ps.pow2) {
      row = (int)((uint32_t)off >> ps.lc);
      w = row & (Wi - 1); h = (row >> ps.lw) & (Hi - 1); b = row >> (ps.lw + ps.lh);
    } else {
      row = (int)((uint32_t)off / (uint32_t)g.C);
      w = row % Wi; h = (row / Wi) % Hi; b = row / (Wi * Hi);
    }
    float d[VEC], v[VEC];
    const uint4 yv = *(const uint4*)(yy + off);
    pool_grad_gather<T>(dd, aa, b, h, w, c0 / VEC, Ho, Wo, g.C, d);
    Vec16<T>::unpack(yv, v);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float m = (v[e] * sc[c0 + e] + sh[c0 + e]) > 0.f ? d[e] : 0.f;
      acc[0][e] += m;
      acc[1][e] += m * v[e];
    }
  });
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_pooled_apply_kernel(const T* __restrict__ dpool,
                                                                  const uint8_t* __restrict__ arg,
                                                                  const T* __restrict__ y, T* __restrict__ dy,
                                                                  const float* __restrict__ coef, int64_t coef_gs,
                                                                  int64_t total_chunks, int cprw, int C, int64_t y_gs,
                                                                  int64_t p_gs, int Hi, int Wi, int Ho, int Wo,
                                                                  const float* __restrict__ stats, int64_t stats_gs,
                                                                  PoolShifts ps) {
  constexpr int VEC = 16 / sizeof(T);
  const int z = blockIdx.y;
  const float* sc = stats + z * stats_gs + 2 * C;
  const float* sh = sc + C;
  const float* k1 = coef + z * coef_gs;
  const float* k2 = k1 + C;
  const float* k3 = k2 + C;
  const T* dd = dpool + z * p_gs;
  const uint8_t* aa = arg + z * p_gs;
  const T* yy = y + z * y_gs;
  T* oo = dy + z * y_gs;
  const uint32_t total = (uint32_t)total_chunks, stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    int ch, row, w, h, b;
    if (ps.pow2) {
      ch = (int)(i & (uint32_t)(cprw - 1)); row = (int)(i >> ps.lq);
      w = row & (Wi - 1); h = (row >> ps.lw) & (Hi - 1); b = row >> (ps.lw + ps.lh);
    } else {
      ch = (int)(i % (uint32_t)cprw); row = (int)(i / (uint32_t)cprw);
      w = row % Wi; h = (row / Wi) % Hi; b = row / (Wi * Hi);
    }
    const int c0 = ch * VEC;
    float d[VEC], v[VEC];
    const uint4 yv = *(const uint4*)(yy + (int64_t)i * VEC);
    pool_grad_gather<T>(dd, aa, b, h, w, ch, Ho, Wo, C, d);
    Vec16<T>::unpack(yv, v);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float m = (v[e] * sc[c0 + e] + sh[c0 + e]) > 0.f ? d[e] : 0.f;
      v[e] = k1[c0 + e] * m + k2[c0 + e] * v[e] + k3[c0 + e];
    }
    *(uint4*)(oo + (int64_t)i * VEC) = Vec16<T>::pack(v);
  }
}

// The same two passes with the pooled gradient staged in LDS: one workgroup per pair of input rows (2p, 2p + 1) of one image.
// Those 2 x Wi pixels receive gradient from the pooled rows p and p + 1 only, which are brought into LDS once (values +
// argmax bytes, contiguous in memory) instead of being gathered from L2 by every pixel (four 16-byte + four 8-byte loads per
// 16 bytes of output: 2.8 TB/s on the last stretch of the step, where nothing else runs).  The windows are visited in the
// order of pool_grad_gather, so pass 2 gives the same bits as the gathering form; pass 1 leaves one row of partial sums per
// workgroup ([rblock][2][C], rblock = blockIdx.x).  Needs Hi = 2 Ho, Wi = 2 Wo, 256 % (C / VEC) == 0.
template <typename T, bool APPLY>
__global__ __launch_bounds__(256) void bn_bwd_pooled_tiled_kernel(const T* __restrict__ dpool, const uint8_t* __restrict__ arg,
                                                                  const T* __restrict__ y, T* __restrict__ dy,
                                                                  const float* __restrict__ coef, int64_t coef_gs,
                                                                  float* __restrict__ partial, int64_t partial_gs, int C,
                                                                  int64_t y_gs, int64_t p_gs, int Hi, int Wi, int Ho, int Wo,
                                                                  const float* __restrict__ stats, int64_t stats_gs) {
  constexpr int VEC = 16 / sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char pooled_lds[];
  const int z = blockIdx.y, t = threadIdx.x;
  const int hp = Hi >> 1;
  const int b = blockIdx.x / hp, p = blockIdx.x - b * hp;
  const int cprw = C / VEC, rowel = Wo * C;
  T* sd = (T*)pooled_lds;                                        // [2][Wo][C]
  uint8_t* sa = (uint8_t*)(pooled_lds + 2 * rowel * sizeof(T));  // [2][Wo][C]
  const bool has2 = p + 1 < Ho;
  {
    const int64_t o = z * p_gs + ((int64_t)b * Ho + p) * rowel;
    const uint4* dsrc = (const uint4*)(dpool + o);
    const uint4* asrc = (const uint4*)(arg + o);
    const int nd = (has2 ? 2 : 1) * rowel / VEC, na = (has2 ? 2 : 1) * rowel / 16;
    for (int i = t; i < nd; i += 256) ((uint4*)sd)[i] = dsrc[i];
    for (int i = t; i < na; i += 256) ((uint4*)sa)[i] = asrc[i];
  }
  __syncthreads();
  const float* sc = stats + z * stats_gs + 2 * C;
  const float* sh = sc + C;
  const float* k1 = coef + z * coef_gs;
  const float* k2 = k1 + C;
  const float* k3 = k2 + C;
  const T* yy = y + z * y_gs + (int64_t)b * Hi * Wi * C;
  T* oo = APPLY ? dy + z * y_gs + (int64_t)b * Hi * Wi * C : nullptr;
  const int ch = t % cprw, c0 = ch * VEC;    // fixed per thread: 256 % cprw == 0
  float scl[VEC], shf[VEC], a1[VEC], a2[VEC], a3[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    scl[e] = sc[c0 + e]; shf[e] = sh[c0 + e];
    if (APPLY) { a1[e] = k1[c0 + e]; a2[e] = k2[c0 + e]; a3[e] = k3[c0 + e]; } else { a1[e] = 0.f; a2[e] = 0.f; a3[e] = 0.f; }
  }
  float s1[VEC], s2[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  const int ppr = 256 / cprw;                // pixels per pass of the block
  for (int px = t / cprw; px < 2 * Wi; px += ppr) {
    const int r = px >= Wi ? 1 : 0, w = px - r * Wi;
    const int q0 = w >> 1, lr0 = r + 1, lc0 = (w & 1) + 1;
    const bool c1 = (w & 1) && q0 + 1 < Wo, r1 = r == 1 && has2;
    const int64_t off = ((int64_t)(2 * p + r) * Wi + w) * C + c0;
    const uint4 yv = *(const uint4*)(yy + off);
    float d[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) d[e] = 0.f;
    auto window = [&](int lrow, int q, int local) {
      const int o = (lrow * Wo + q) * C + c0;
      float g[VEC];
      Vec16<T>::unpack(*(const uint4*)(sd + o), g);
      if constexpr (VEC == 8) {
        const uint2 av = *(const uint2*)(sa + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if ((int)((av.x >> (8 * e)) & 0xff) == local) d[e] += g[e];
          if ((int)((av.y >> (8 * e)) & 0xff) == local) d[4 + e] += g[4 + e];
        }
      } else {
        const uint32_t av = *(const uint32_t*)(sa + o);
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          if ((int)((av >> (8 * e)) & 0xff) == local) d[e] += g[e];
      }
    };
    window(0, q0, lr0 * 3 + lc0);
    if (c1) window(0, q0 + 1, lr0 * 3);
    if (r1) window(1, q0, lc0);
    if (r1 && c1) window(1, q0 + 1, 0);
    float v[VEC];
    Vec16<T>::unpack(yv, v);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float m = (v[e] * scl[e] + shf[e]) > 0.f ? d[e] : 0.f;
      if (APPLY) {
        v[e] = a1[e] * m + a2[e] * v[e] + a3[e];
      } else {
        s1[e] += m;
        s2[e] += m * v[e];
      }
    }
    if (APPLY) *(uint4*)(oo + off) = Vec16<T>::pack(v);
  }
  if constexpr (!APPLY) {
    float* red = (float*)pooled_lds;          // [256][VEC], one quantity at a time (the launcher sizes LDS for it)
    float* out = partial + z * partial_gs + (int64_t)blockIdx.x * 2 * C;
#pragma unroll
    for (int qn = 0; qn < 2; ++qn) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < VEC; ++e) red[t * VEC + e] = qn == 0 ? s1[e] : s2[e];
      __syncthreads();
      for (int c = t; c < C; c += 256) {
        const int cc = c / VEC, e = c - cc * VEC;
        float s = 0.f;
        for (int k = cc; k < 256; k += cprw) s += red[k * VEC + e];
        out[qn * C + c] = s;
      }
    }
  }
}

// the totals kernels pay a prologue per workgroup: fewer, longer-lived workgroups than the plain passes -- ONE resident wave of
// them (256 CUs x 8 workgroups of 256 threads, over the modalities of the launch).  B = 64 step, three interleaved rounds:
// 256 per modality 14.60 ms, 384 14.50, 512 14.45, 682 14.43, 1024 14.46, 2048 14.80, 4096 15.05 (scripts/experiments/r4_ab.sh)
static int tot_blocks(int64_t chunks, int64_t groups) {
  static const int64_t fixed = getenv("IEEE_BN_TOTALS_BLOCKS") ? atoll(getenv("IEEE_BN_TOTALS_BLOCKS")) : 0;
  const int64_t cap = fixed > 0 ? fixed : std::max<int64_t>(256, 2048 / std::max<int64_t>(groups, 1));
  int64_t b = (chunks + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

static int ew_blocks(int64_t chunks) {
  static const int64_t cap = getenv("IEEE_EW_BLOCKS") ? atoll(getenv("IEEE_EW_BLOCKS")) : 8192;   // 4096: +0.13 ms per step; 16384 and more: same as 8192 (scripts/experiments/scan_ew.sh)
  int64_t b = (chunks + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace ieee

using namespace ieee;

static int vec_of(int dtype) { return dtype == IEEE_BF16 ? 8 : 4; }

// the *_fixed_kernel forms: a thread's channel chunk must not change along its grid-stride walk (the chunks of a row divide
// the block) and the coefficient tables must take 16-byte loads.  IEEE_BN_FIXED: bit mask of the passes that use them
// (1 forward apply, 2 / 4 / 8 backward apply with mask kind 0 / 1 / 2); IEEE_BN_UNROLL: chunks in flight per thread.
// Measured alone at the B = 64 shapes (scripts/bn_probe.py, profiles/r04_bn_probe.txt): only the backward with the mask
// recomputed from y gains -- 5 coefficient rows per chunk instead of 3: 4.55 -> 5.68 TB/s on the layer1 / layer2 maps;
// the other passes already stream at 5.4-6.0 TB/s and lose a little at C = 2048 -- so that pass alone is on by default.
static bool fixed_channel_ok(int which, int cprw, const void* p1, const void* p2) {
  static const int on = getenv("IEEE_BN_FIXED") ? atoi(getenv("IEEE_BN_FIXED")) : 8;
  return (on & which) && cprw >= 1 && cprw <= 256 && 256 % cprw == 0 && ((uintptr_t)p1 & 15) == 0 && ((uintptr_t)p2 & 15) == 0;
}
static int fixed_unroll() {
  static const int u = getenv("IEEE_BN_UNROLL") ? atoi(getenv("IEEE_BN_UNROLL")) : 1;
  return u == 2 ? 2 : 1;
}

extern "C" int64_t ieee_bn_partial_floats(int dtype, int64_t M, int64_t C) {
  const RedGeom g = red_geom(M, (int)C, vec_of(dtype));
  const int64_t rb = std::max((int64_t)g.rblocks, (M + 127) / 128);   // (M + 127) / 128: the tiled pooled backward's row blocks
  return rb * 2 * C;
}

extern "C" int ieee_bn2d_fwd(const void* y, const void* residual, void* out, int dtype, int64_t groups, int64_t M,
                             int64_t C, int64_t act_gs, const float* gamma, const float* beta, int64_t param_gs,
                             float* running_mean, float* running_var, int64_t buf_gs, float* stats, float* partial,
                             float momentum, float eps, int training, int relu, int64_t stats_rblocks, void* relu_bits,
                             void* stream) {
  IEEE_REQUIRE(y && gamma && beta && stats, "bn2d_fwd: null pointer");
  IEEE_REQUIRE(!relu_bits || (dtype == IEEE_BF16 && out && act_gs % 8 == 0), "bn2d_fwd: relu_bits needs a bf16 output");
  IEEE_REQUIRE(dtype == IEEE_F32 || dtype == IEEE_BF16, "bn2d_fwd: bad dtype");
  IEEE_REQUIRE(C % vec_of(dtype) == 0, "bn2d_fwd: C %ld not a multiple of %d", (long)C, vec_of(dtype));
  IEEE_REQUIRE(training || (running_mean && running_var), "bn2d_fwd: eval mode needs running stats");
  IEEE_REQUIRE(!training || partial, "bn2d_fwd: training needs the partial-sum scratch");
  hipStream_t st = (hipStream_t)stream;
  RedGeom g = red_geom(M, (int)C, vec_of(dtype));
  if (stats_rblocks > 0) g.rblocks = (int)stats_rblocks;   // partial sums already emitted by the producing conv
  const int64_t partial_gs = (int64_t)g.rblocks * 2 * C;
  if (training && stats_rblocks == 0) {
    dim3 grid(g.cblocks * g.rblocks, (unsigned)groups);
    if (dtype == IEEE_F32) bn_stats_kernel<float><<<grid, 256, 0, st>>>((const float*)y, act_gs, g, partial, partial_gs);
    else bn_stats_kernel<bf16><<<grid, 256, 0, st>>>((const bf16*)y, act_gs, g, partial, partial_gs);
    IEEE_TRY(launch_status("bn_stats_kernel"));
  }
  if (!(training && stats_rblocks < 0)) {   // stats_rblocks < 0: the producing conv finalized `stats` itself (ieee_conv2d_fwd_bn_train)
    const int lpc = training ? finalize_lpc(g.rblocks) : 32;
    bn_finalize_kernel<<<dim3(cdiv(C, 256 / lpc), (unsigned)groups), 256, 0, st>>>(
        partial, partial_gs, g.rblocks, (int)M, (int)C, gamma, beta, param_gs, running_mean, running_var, buf_gs, stats,
        4 * C, momentum, eps, training, stats_rblocks > 0 ? 1 : 0, lpc);
    IEEE_TRY(launch_status("bn_finalize_kernel"));
  }
  if (out == nullptr) return IEEE_OK;   // statistics only: the consumer applies scale/shift itself
  const int64_t chunks = M * C / vec_of(dtype);
  dim3 grid(ew_blocks(chunks), (unsigned)groups);
  if (dtype == IEEE_BF16 && fixed_channel_ok(1, g.cprw, stats, nullptr)) {   // per-channel constants in registers
    const bf16 *yb = (const bf16*)y, *rb = (const bf16*)residual;
    bf16* ob = (bf16*)out;
    uint8_t* bits = (uint8_t*)relu_bits;
    const bool u2 = fixed_unroll() == 2;
#define IEEE_BN_APPLY_FIXED(RES, BITS)                                                                                         \
    do {                                                                                                                       \
      if (u2) bn_apply_fixed_kernel<bf16, RES, BITS, 2><<<grid, 256, 0, st>>>(yb, rb, ob, stats, 4 * C, chunks, g.cprw, (int)C, act_gs, relu, bits); \
      else bn_apply_fixed_kernel<bf16, RES, BITS, 1><<<grid, 256, 0, st>>>(yb, rb, ob, stats, 4 * C, chunks, g.cprw, (int)C, act_gs, relu, bits);    \
    } while (0)
    if (residual && bits) IEEE_BN_APPLY_FIXED(true, true);
    else if (residual) IEEE_BN_APPLY_FIXED(true, false);
    else if (bits) IEEE_BN_APPLY_FIXED(false, true);
    else IEEE_BN_APPLY_FIXED(false, false);
#undef IEEE_BN_APPLY_FIXED
    return launch_status("bn_apply_fixed_kernel");
  }
  if (dtype == IEEE_F32)
    bn_apply_kernel<float><<<grid, 256, 0, st>>>((const float*)y, (const float*)residual, (float*)out, stats, 4 * C,
                                                 chunks, g.cprw, (int)C, act_gs, relu, nullptr);
  else
    bn_apply_kernel<bf16><<<grid, 256, 0, st>>>((const bf16*)y, (const bf16*)residual, (bf16*)out, stats, 4 * C, chunks,
                                                g.cprw, (int)C, act_gs, relu, (uint8_t*)relu_bits);
  return launch_status("bn_apply_kernel");
}

extern "C" int ieee_bn2d_bwd_ev(const void* dout, const void* out_mask, const void* y, void* dy, void* g_out, int dtype,
                                int64_t groups, int64_t M, int64_t C, int64_t act_gs, const float* gamma,
                                int64_t param_gs, const float* stats, float* dgamma, float* dbeta, int64_t grad_gs,
                                float* partial, float* coef, int accumulate, int mask_from_y, int64_t stats_rblocks,
                                void* done_event, void* stream);

extern "C" int ieee_bn2d_bwd(const void* dout, const void* out_mask, const void* y, void* dy, void* g_out, int dtype,
                             int64_t groups, int64_t M, int64_t C, int64_t act_gs, const float* gamma,
                             int64_t param_gs, const float* stats, float* dgamma, float* dbeta, int64_t grad_gs,
                             float* partial, float* coef, int accumulate, int mask_from_y, int64_t stats_rblocks,
                             void* stream) {
  return ieee_bn2d_bwd_ev(dout, out_mask, y, dy, g_out, dtype, groups, M, C, act_gs, gamma, param_gs, stats, dgamma, dbeta,
                          grad_gs, partial, coef, accumulate, mask_from_y, stats_rblocks, nullptr, stream);
}

// ---- one-time self-check of the completion-signal event (see ieee_bn2d_bwd_ev below).  The executor orders a side-stream
// weight gradient behind a BatchNorm backward ONLY through the stop event of hipExtLaunchKernelGGL + hipStreamWaitEvent;
// the HIP documentation does not promise that a stop event that was never hipEventRecord-ed orders another stream, so
// the first user measures it on this runtime: a kernel that spins ~300 us and then sets a flag carries the event, a second
// stream waits on the event and reads the flag -- three times, with a hipEventDisableTiming event like the executor's,
// on the two streams the caller hands in.  0 = the waiter saw the flag every time (the event rides), anything else = use
// hipEventRecord.
namespace ieee {
__global__ void ride_spin_kernel(int* flag, long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  *flag = 1;
}
__global__ void ride_check_kernel(const int* flag, int* out) { *out = *flag; }
}  // namespace ieee

extern "C" int ieee_event_ride_selfcheck(void* stream_a, void* stream_b) {
  static int cached = -1;
  if (cached >= 0) return cached;
  IEEE_REQUIRE(stream_a != stream_b, "event_ride_selfcheck: needs two different streams");
  // The check runs on the CALLER's two streams (the executor passes its launch and side stream): creating and destroying
  // streams of its own in the middle of a step re-shuffled the runtime's hardware-queue assignment on this pool -- the
  // weight-gradient stream stopped overlapping the launch stream afterwards (22.5 instead of 14.9 ms per step, round 4).
  int result = 1;
  int* dev = nullptr;
  hipStream_t sa = (hipStream_t)stream_a, sb = (hipStream_t)stream_b;
  hipEvent_t ev = nullptr;
  do {
    if (hipMalloc(&dev, 2 * sizeof(int)) != hipSuccess) break;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) break;
    bool ok = true;
    for (int rep = 0; rep < 3 && ok; ++rep) {
      int host[2] = {-1, -1};
      ok = hipMemsetAsync(dev, 0, 2 * sizeof(int), sa) == hipSuccess;
      // wall_clock64 ticks at 100 MHz: 30 000 ticks = 300 us, far longer than a launch takes to reach the other queue
      hipExtLaunchKernelGGL(ride_spin_kernel, dim3(1), dim3(1), 0, sa, nullptr, ev, 0, dev, (long long)30000);
      ok = ok && hipGetLastError() == hipSuccess;
      ok = ok && hipStreamWaitEvent(sb, ev, 0) == hipSuccess;
      ride_check_kernel<<<1, 1, 0, sb>>>(dev, dev + 1);
      ok = ok && hipGetLastError() == hipSuccess;
      ok = ok && hipStreamSynchronize(sb) == hipSuccess && hipStreamSynchronize(sa) == hipSuccess;
      ok = ok && hipMemcpy(host, dev, sizeof(host), hipMemcpyDeviceToHost) == hipSuccess;
      ok = ok && host[0] == 1 && host[1] == 1;
    }
    result = ok ? 0 : 2;
  } while (false);
  if (ev) (void)hipEventDestroy(ev);
  if (dev) (void)hipFree(dev);
  cached = result;
  return result;
}

// done_event (a hipEvent_t, optional): signalled by the LAST kernel of the call itself -- it rides on that dispatch as its
// completion signal (hipExtLaunchKernelGGL's stop event) instead of a separate hipEventRecord behind it.  An event
// record is a barrier packet of its own in the queue: the next kernel of the stream starts only after the previous
// one has drained AND the packet has been processed (+5 us per record on the dgrad / BatchNorm chain of the backward,
// 49 per step: found in the round-3 kernel trace as a 7.5 us gap behind every bn_bwd_apply that a weight gradient
// forks from).
// the apply pass of the BatchNorm backward (dy = k1*g + k2*y + k3), shared by the train-mode and the frozen form
static int launch_bwd_apply(const void* dout, const void* out_mask, const void* y, void* dy, void* g_out, int dtype,
                            int64_t groups, int64_t M, int64_t C, int64_t act_gs, const float* stats, const float* coef,
                            int mask_from_y, void* done_event, hipStream_t st) {
  RedGeom g = red_geom(M, (int)C, vec_of(dtype));
  const int64_t chunks = M * C / vec_of(dtype);
  dim3 grid(ew_blocks(chunks), (unsigned)groups);
  hipEvent_t ev = (hipEvent_t)done_event;
  const int mask_kind = out_mask ? 1 : (mask_from_y ? 2 : 0);
  if (dtype == IEEE_BF16 && fixed_channel_ok(2 << mask_kind, g.cprw, stats, coef)) {   // per-channel constants in registers
    const int variant = mask_kind * 4 + (g_out ? 2 : 0) + (fixed_unroll() == 2 ? 1 : 0);
#define IEEE_BN_BWD_FIXED(MASK, GOUT, UNROLL)                                                                               \
    case MASK * 4 + (GOUT ? 2 : 0) + (UNROLL == 2 ? 1 : 0):                                                                \
      hipExtLaunchKernelGGL((bn_bwd_apply_fixed_kernel<bf16, MASK, GOUT, UNROLL>), grid, dim3(256), 0, st, nullptr, ev, 0, \
                            (const bf16*)dout, (const bf16*)out_mask, (const bf16*)y, (bf16*)dy, (bf16*)g_out,              \
                            (const float*)coef, (int64_t)(3 * C), chunks, g.cprw, (int)C, act_gs, stats, (int64_t)(4 * C)); \
      break;
    switch (variant) {
      IEEE_BN_BWD_FIXED(0, false, 1) IEEE_BN_BWD_FIXED(0, false, 2) IEEE_BN_BWD_FIXED(0, true, 1)
      IEEE_BN_BWD_FIXED(0, true, 2) IEEE_BN_BWD_FIXED(1, false, 1) IEEE_BN_BWD_FIXED(1, false, 2)
      IEEE_BN_BWD_FIXED(1, true, 1) IEEE_BN_BWD_FIXED(1, true, 2) IEEE_BN_BWD_FIXED(2, false, 1)
      IEEE_BN_BWD_FIXED(2, false, 2) IEEE_BN_BWD_FIXED(2, true, 1) IEEE_BN_BWD_FIXED(2, true, 2)
    }
#undef IEEE_BN_BWD_FIXED
    return launch_status("bn_bwd_apply_fixed_kernel");
  }
  if (dtype == IEEE_F32) {
    if (ev)
      hipExtLaunchKernelGGL(bn_bwd_apply_kernel<float>, grid, dim3(256), 0, st, nullptr, ev, 0, (const float*)dout,
                            (const float*)out_mask, (const float*)y, (float*)dy, (float*)g_out, (const float*)coef, (int64_t)(3 * C),
                            chunks, g.cprw, (int)C, act_gs, stats, (int64_t)(4 * C), mask_from_y);
    else
      bn_bwd_apply_kernel<float><<<grid, 256, 0, st>>>((const float*)dout, (const float*)out_mask, (const float*)y,
                                                       (float*)dy, (float*)g_out, coef, 3 * C, chunks, g.cprw, (int)C,
                                                       act_gs, stats, 4 * C, mask_from_y);
  } else {
    if (ev)
      hipExtLaunchKernelGGL(bn_bwd_apply_kernel<bf16>, grid, dim3(256), 0, st, nullptr, ev, 0, (const bf16*)dout,
                            (const bf16*)out_mask, (const bf16*)y, (bf16*)dy, (bf16*)g_out, (const float*)coef, (int64_t)(3 * C),
                            chunks, g.cprw, (int)C, act_gs, stats, (int64_t)(4 * C), mask_from_y);
    else
      bn_bwd_apply_kernel<bf16><<<grid, 256, 0, st>>>((const bf16*)dout, (const bf16*)out_mask, (const bf16*)y,
                                                      (bf16*)dy, (bf16*)g_out, coef, 3 * C, chunks, g.cprw, (int)C,
                                                      act_gs, stats, 4 * C, mask_from_y);
  }
  return launch_status("bn_bwd_apply_kernel");
}

extern "C" int ieee_bn2d_bwd_ev(const void* dout, const void* out_mask, const void* y, void* dy, void* g_out, int dtype,
                                int64_t groups, int64_t M, int64_t C, int64_t act_gs, const float* gamma,
                                int64_t param_gs, const float* stats, float* dgamma, float* dbeta, int64_t grad_gs,
                                float* partial, float* coef, int accumulate, int mask_from_y, int64_t stats_rblocks,
                                void* done_event, void* stream) {
  IEEE_REQUIRE(dout && y && dy && gamma && stats && partial && coef, "bn2d_bwd: null pointer");
  IEEE_REQUIRE(dtype == IEEE_F32 || dtype == IEEE_BF16, "bn2d_bwd: bad dtype");
  IEEE_REQUIRE(C % vec_of(dtype) == 0, "bn2d_bwd: C not a multiple of the vector width");
  hipStream_t st = (hipStream_t)stream;
  RedGeom g = red_geom(M, (int)C, vec_of(dtype));
  if (stats_rblocks > 0) g.rblocks = (int)stats_rblocks;   // sums already emitted by the producing dgrad
  const int64_t partial_gs = (int64_t)g.rblocks * 2 * C;
  dim3 rgrid(g.cblocks * g.rblocks, (unsigned)groups);
  if (stats_rblocks > 0) {
  } else if (dtype == IEEE_F32)
    bn_bwd_reduce_kernel<float><<<rgrid, 256, 0, st>>>((const float*)dout, (const float*)out_mask, (const float*)y,
                                                       act_gs, g, partial, partial_gs, stats, 4 * C, mask_from_y);
  else
    bn_bwd_reduce_kernel<bf16><<<rgrid, 256, 0, st>>>((const bf16*)dout, (const bf16*)out_mask, (const bf16*)y, act_gs,
                                                      g, partial, partial_gs, stats, 4 * C, mask_from_y);
  IEEE_TRY(launch_status("bn_bwd_reduce_kernel"));
  const int lpc = finalize_lpc(g.rblocks);
  bn_bwd_finalize_kernel<<<dim3(cdiv(C, 256 / lpc), (unsigned)groups), 256, 0, st>>>(
      partial, partial_gs, g.rblocks, (int)M, (int)C, gamma, param_gs, stats, 4 * C, dgamma, dbeta, grad_gs, coef,
      3 * C, accumulate, stats_rblocks > 0 ? 1 : 0, lpc);
  IEEE_TRY(launch_status("bn_bwd_finalize_kernel"));
  return launch_bwd_apply(dout, out_mask, y, dy, g_out, dtype, groups, M, C, act_gs, stats, coef, mask_from_y, done_event, st);
}

namespace ieee {
// frozen BatchNorm (eval-mode statistics): y -> y * scale + shift is a fixed affine map, so dy = scale * g
__global__ void bn_frozen_coef_kernel(const float* __restrict__ stats, int64_t stats_gs, float* __restrict__ coef,
                                      int64_t coef_gs, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float* k = coef + blockIdx.y * coef_gs;
  k[c] = stats[blockIdx.y * stats_gs + 2 * C + c];
  k[C + c] = 0.f;
  k[2 * C + c] = 0.f;
}
}  // namespace ieee

extern "C" int ieee_bn2d_bwd_frozen(const void* dout, const void* out_mask, const void* y, void* dy, void* g_out, int dtype,
                                    int64_t groups, int64_t M, int64_t C, int64_t act_gs, const float* stats, float* coef,
                                    int mask_from_y, void* done_event, void* stream) {
  IEEE_REQUIRE(dout && y && dy && stats && coef, "bn2d_bwd_frozen: null pointer");
  IEEE_REQUIRE(dtype == IEEE_F32 || dtype == IEEE_BF16, "bn2d_bwd_frozen: bad dtype");
  IEEE_REQUIRE(C % vec_of(dtype) == 0, "bn2d_bwd_frozen: C not a multiple of the vector width");
  hipStream_t st = (hipStream_t)stream;
  bn_frozen_coef_kernel<<<dim3(cdiv(C, 256), (unsigned)groups), 256, 0, st>>>(stats, 4 * C, coef, 3 * C, (int)C);
  IEEE_TRY(launch_status("bn_frozen_coef_kernel"));
  return launch_bwd_apply(dout, out_mask, y, dy, g_out, dtype, groups, M, C, act_gs, stats, coef, mask_from_y, done_event, st);
}

/* train-mode BatchNorm2d forward whose statistics arrive as fixed-point totals (ieee_conv_next_bn_totals): finalize + apply
 * in ONE launch.  out NULL: statistics only (stats / running statistics are still published). */
extern "C" int ieee_bn2d_fwd_totals(const void* y, const void* residual, void* out, int dtype, int64_t groups, int64_t M,
                                    int64_t C, int64_t act_gs, const float* gamma, const float* beta, int64_t param_gs,
                                    float* running_mean, float* running_var, int64_t buf_gs, float* stats, const void* totals,
                                    int replicas, float momentum, float eps, int relu, void* relu_bits, int* overflow,
                                    void* stream) {
  IEEE_REQUIRE(replicas >= 1 && replicas <= 64, "bn2d_fwd_totals: 1..64 replicas");
  IEEE_REQUIRE(y && gamma && beta && stats && totals, "bn2d_fwd_totals: null pointer");
  IEEE_REQUIRE(dtype == IEEE_BF16, "bn2d_fwd_totals: bf16 only (the fused statistics exist on the bf16 path)");
  const int cprw = (int)(C / 8);
  IEEE_REQUIRE(C % 8 == 0 && cprw >= 1 && cprw <= 256 && 256 % cprw == 0, "bn2d_fwd_totals: C / 8 must divide 256");
  IEEE_REQUIRE(!relu_bits || (out && act_gs % 8 == 0), "bn2d_fwd_totals: relu_bits needs an output");
  IEEE_REQUIRE((((uintptr_t)totals | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0 && param_gs % 4 == 0,
               "bn2d_fwd_totals: totals / gamma / beta must be 16-byte aligned (group stride a multiple of 4 floats)");
  hipStream_t st = (hipStream_t)stream;
  const int64_t chunks = M * C / 8;
  dim3 grid(out ? tot_blocks(chunks, groups) : 1, (unsigned)groups);
  const bf16 *yb = (const bf16*)y, *rb = (const bf16*)residual;
  bf16* ob = (bf16*)out;
  uint8_t* bits = (uint8_t*)relu_bits;
#define IEEE_BN_TOT(RES, BITS)                                                                                              \
  bn_apply_totals_kernel<bf16, RES, BITS><<<grid, 256, 2 * C * sizeof(float), st>>>(yb, rb, ob, (const long long*)totals, gamma, beta, param_gs, \
      running_mean, running_var, buf_gs, stats, (int)M, momentum, eps, chunks, cprw, (int)C, act_gs, relu, bits,       \
      TOT_INV_FWD / (double)M, M > 1 ? (float)((double)M / (double)(M - 1)) : 1.0f, replicas, groups * 2 * C, overflow)
  if (residual && bits) IEEE_BN_TOT(true, true);
  else if (residual) IEEE_BN_TOT(true, false);
  else if (bits) IEEE_BN_TOT(false, true);
  else IEEE_BN_TOT(false, false);
#undef IEEE_BN_TOT
  return launch_status("bn_apply_totals_kernel");
}

/* ... and its backward: totals = sum g, sum g*y from the dgrad epilogue; finalize + apply in ONE launch */
extern "C" int ieee_bn2d_bwd_totals(const void* dout, const void* out_mask, const void* y, void* dy, void* g_out, int dtype,
                                    int64_t groups, int64_t M, int64_t C, int64_t act_gs, const float* gamma,
                                    int64_t param_gs, const float* stats, float* dgamma, float* dbeta, int64_t grad_gs,
                                    const void* totals, int replicas, int mask_from_y, int* overflow, void* done_event,
                                    void* stream) {
  IEEE_REQUIRE(replicas >= 1 && replicas <= 64, "bn2d_bwd_totals: 1..64 replicas");
  IEEE_REQUIRE(dout && y && dy && gamma && stats && totals, "bn2d_bwd_totals: null pointer");
  IEEE_REQUIRE(dtype == IEEE_BF16, "bn2d_bwd_totals: bf16 only");
  const int cprw = (int)(C / 8);
  IEEE_REQUIRE(C % 8 == 0 && cprw >= 1 && cprw <= 256 && 256 % cprw == 0, "bn2d_bwd_totals: C / 8 must divide 256");
  IEEE_REQUIRE((((uintptr_t)totals | (uintptr_t)gamma | (uintptr_t)stats) & 15) == 0 && param_gs % 4 == 0,
               "bn2d_bwd_totals: totals / gamma / stats must be 16-byte aligned (group stride a multiple of 4 floats)");
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t ev = (hipEvent_t)done_event;
  const int64_t chunks = M * C / 8;
  dim3 grid(tot_blocks(chunks, groups), (unsigned)groups);
  const int variant = (out_mask ? 1 : (mask_from_y ? 2 : 0)) * 2 + (g_out ? 1 : 0);
#define IEEE_BN_BWD_TOT(MASK, GOUT)                                                                                          \
  case MASK * 2 + (GOUT ? 1 : 0):                                                                                            \
    hipExtLaunchKernelGGL((bn_bwd_apply_totals_kernel<bf16, MASK, GOUT>), grid, dim3(256), 3 * C * sizeof(float), st,        \
                          nullptr, ev, 0,                                                                                    \
                          (const bf16*)dout, (const bf16*)out_mask, (const bf16*)y, (bf16*)dy, (bf16*)g_out,                 \
                          (const long long*)totals, gamma, param_gs, stats, dgamma, dbeta, grad_gs, (int)M, chunks, cprw,    \
                          (int)C, act_gs, 1.0 / (double)M, replicas, groups * 2 * C, overflow);                              \
    break;
  switch (variant) {
    IEEE_BN_BWD_TOT(0, false) IEEE_BN_BWD_TOT(0, true) IEEE_BN_BWD_TOT(1, false) IEEE_BN_BWD_TOT(1, true)
    IEEE_BN_BWD_TOT(2, false) IEEE_BN_BWD_TOT(2, true)
  }
#undef IEEE_BN_BWD_TOT
  return launch_status("bn_bwd_apply_totals_kernel");
}

/* the block-output BatchNorm's backward (dout = g, already masked) that also leaves the downsample branch's backward sums
 * (sum g, sum g*y_ds) in that unit's totals: see bn_bwd_apply_totals_ds_kernel */
extern "C" int ieee_bn2d_bwd_totals_ds(const void* dout, const void* y, const void* y_ds, void* dy, int dtype, int64_t groups,
                                       int64_t M, int64_t C, int64_t act_gs, const float* gamma, int64_t param_gs,
                                       const float* stats, float* dgamma, float* dbeta, int64_t grad_gs, const void* totals,
                                       int replicas, void* totals_ds, int replicas_ds, int* overflow, void* done_event,
                                       void* stream) {
  IEEE_REQUIRE(replicas >= 1 && replicas <= 64 && replicas_ds >= 1 && replicas_ds <= 64, "bn2d_bwd_totals_ds: 1..64 replicas");
  IEEE_REQUIRE(dout && y && y_ds && dy && gamma && stats && totals && totals_ds, "bn2d_bwd_totals_ds: null pointer");
  IEEE_REQUIRE(dtype == IEEE_BF16, "bn2d_bwd_totals_ds: bf16 only");
  const int cprw = (int)(C / 8);
  IEEE_REQUIRE(C % 8 == 0 && cprw >= 1 && cprw <= 256 && 256 % cprw == 0, "bn2d_bwd_totals_ds: C / 8 must divide 256");
  IEEE_REQUIRE((((uintptr_t)totals | (uintptr_t)gamma | (uintptr_t)stats) & 15) == 0 && param_gs % 4 == 0,
               "bn2d_bwd_totals_ds: totals / gamma / stats must be 16-byte aligned (group stride a multiple of 4 floats)");
  hipStream_t st = (hipStream_t)stream;
  const int64_t chunks = M * C / 8;
  dim3 grid(tot_blocks(chunks, groups), (unsigned)groups);
  const size_t lds = std::max<size_t>(3 * C, 8 * 256) * sizeof(float);
  // a workgroup's share of +-2^62 (rounded down): the branch's sum g*y_ds cannot wrap whatever the workgroups add
  const float lim2 = nextafterf(4611686018427387904.0f / (float)grid.x, 0.f);
  hipExtLaunchKernelGGL((bn_bwd_apply_totals_ds_kernel<bf16>), grid, dim3(256), (uint32_t)lds, st, nullptr, (hipEvent_t)done_event, 0,
                        (const bf16*)dout, (const bf16*)y, (const bf16*)y_ds, (bf16*)dy, (const long long*)totals, gamma, param_gs,
                        stats, dgamma, dbeta, grad_gs, (int)M, chunks, cprw, (int)C, act_gs, 1.0 / (double)M, replicas,
                        groups * 2 * C, (long long*)totals_ds, replicas_ds, groups * 2 * C, lim2, overflow);
  return launch_status("bn_bwd_apply_totals_ds_kernel");
}

extern "C" int ieee_bn2d_bwd_pooled(const void* dpool, const uint8_t* argmax, const void* y, void* dy, int dtype,
                                    int64_t groups, int64_t B, int64_t Hi, int64_t Wi, int64_t C, const float* gamma,
                                    int64_t param_gs, const float* stats, float* dgamma, float* dbeta, int64_t grad_gs,
                                    float* partial, float* coef, int accumulate, void* stream) {
  IEEE_REQUIRE(dpool && argmax && y && dy && gamma && stats && partial && coef, "bn2d_bwd_pooled: null pointer");
  IEEE_REQUIRE(dtype == IEEE_F32 || dtype == IEEE_BF16, "bn2d_bwd_pooled: bad dtype");
  IEEE_REQUIRE(C % vec_of(dtype) == 0, "bn2d_bwd_pooled: C not a multiple of the vector width");
  IEEE_REQUIRE(B * Hi * Wi * C < (1ll << 31), "bn2d_bwd_pooled: more than 2^31 elements per group");
  hipStream_t st = (hipStream_t)stream;
  const int64_t M = B * Hi * Wi;
  const int Ho = (int)((Hi + 2 - 3) / 2 + 1), Wo = (int)((Wi + 2 - 3) / 2 + 1);
  const int64_t y_gs = M * C, p_gs = B * Ho * Wo * C;
  RedGeom g = red_geom(M, (int)C, vec_of(dtype));
  {
    static const bool tiled_on = !(getenv("IEEE_POOLED_TILED") && atoi(getenv("IEEE_POOLED_TILED")) == 0);
    const int vec = vec_of(dtype), es = dtype == IEEE_BF16 ? 2 : 4;
    const int64_t cprw = C / vec, rowel = (int64_t)Wo * C;
    const int64_t lds = std::max<int64_t>(2 * rowel * (es + 1), 256 * vec * 4);
    const int64_t rbt = B * (Hi / 2);
    if (tiled_on && Hi == 2 * Ho && Wi == 2 * Wo && Wi >= 64 && 256 % cprw == 0 && rowel % 16 == 0 && lds <= 64 * 1024 &&
        rbt <= (M + 127) / 128) {
      const int64_t partial_gs = rbt * 2 * C;
      dim3 tgrid((unsigned)rbt, (unsigned)groups);
      if (dtype == IEEE_F32)
        bn_bwd_pooled_tiled_kernel<float, false><<<tgrid, 256, lds, st>>>((const float*)dpool, argmax, (const float*)y, nullptr,
            coef, 3 * C, partial, partial_gs, (int)C, y_gs, p_gs, (int)Hi, (int)Wi, Ho, Wo, stats, 4 * C);
      else
        bn_bwd_pooled_tiled_kernel<bf16, false><<<tgrid, 256, lds, st>>>((const bf16*)dpool, argmax, (const bf16*)y, nullptr,
            coef, 3 * C, partial, partial_gs, (int)C, y_gs, p_gs, (int)Hi, (int)Wi, Ho, Wo, stats, 4 * C);
      IEEE_TRY(launch_status("bn_bwd_pooled_tiled_kernel(reduce)"));
      const int lpc = finalize_lpc(rbt);
      bn_bwd_finalize_kernel<<<dim3(cdiv(C, 256 / lpc), (unsigned)groups), 256, 0, st>>>(
          partial, partial_gs, (int)rbt, (int)M, (int)C, gamma, param_gs, stats, 4 * C, dgamma, dbeta, grad_gs, coef, 3 * C,
          accumulate, 0, lpc);
      IEEE_TRY(launch_status("bn_bwd_finalize_kernel"));
      if (dtype == IEEE_F32)
        bn_bwd_pooled_tiled_kernel<float, true><<<tgrid, 256, lds, st>>>((const float*)dpool, argmax, (const float*)y, (float*)dy,
            coef, 3 * C, nullptr, 0, (int)C, y_gs, p_gs, (int)Hi, (int)Wi, Ho, Wo, stats, 4 * C);
      else
        bn_bwd_pooled_tiled_kernel<bf16, true><<<tgrid, 256, lds, st>>>((const bf16*)dpool, argmax, (const bf16*)y, (bf16*)dy,
            coef, 3 * C, nullptr, 0, (int)C, y_gs, p_gs, (int)Hi, (int)Wi, Ho, Wo, stats, 4 * C);
      return launch_status("bn_bwd_pooled_tiled_kernel(apply)");
    }
  }
  const int64_t partial_gs = (int64_t)g.rblocks * 2 * C;
  dim3 rgrid(g.cblocks * g.rblocks, (unsigned)groups);
  auto lg = [](int64_t v) { int l = 0; while ((1ll << l) < v) ++l; return ((1ll << l) == v) ? l : -1; };
  PoolShifts ps;
  ps.lw = lg(Wi); ps.lh = lg(Hi); ps.lc = lg(C); ps.lq = lg(g.cprw);
  ps.pow2 = (ps.lw >= 0 && ps.lh >= 0 && ps.lc >= 0 && ps.lq >= 0) ? 1 : 0;
  if (dtype == IEEE_F32)
    bn_bwd_pooled_reduce_kernel<float><<<rgrid, 256, 0, st>>>((const float*)dpool, argmax, (const float*)y, y_gs, p_gs, g,
                                                              (int)Hi, (int)Wi, Ho, Wo, partial, partial_gs, stats, 4 * C, ps);
  else
    bn_bwd_pooled_reduce_kernel<bf16><<<rgrid, 256, 0, st>>>((const bf16*)dpool, argmax, (const bf16*)y, y_gs, p_gs, g,
                                                             (int)Hi, (int)Wi, Ho, Wo, partial, partial_gs, stats, 4 * C, ps);
  IEEE_TRY(launch_status("bn_bwd_pooled_reduce_kernel"));
  const int lpc = finalize_lpc(g.rblocks);
  bn_bwd_finalize_kernel<<<dim3(cdiv(C, 256 / lpc), (unsigned)groups), 256, 0, st>>>(
      partial, partial_gs, g.rblocks, (int)M, (int)C, gamma, param_gs, stats, 4 * C, dgamma, dbeta, grad_gs, coef,
      3 * C, accumulate, 0, lpc);
  IEEE_TRY(launch_status("bn_bwd_finalize_kernel"));
  const int64_t chunks = M * C / vec_of(dtype);
  dim3 grid(ew_blocks(chunks), (unsigned)groups);
  if (dtype == IEEE_F32)
    bn_bwd_pooled_apply_kernel<float><<<grid, 256, 0, st>>>((const float*)dpool, argmax, (const float*)y, (float*)dy, coef,
                                                            3 * C, chunks, g.cprw, (int)C, y_gs, p_gs, (int)Hi, (int)Wi, Ho,
                                                            Wo, stats, 4 * C, ps);
  else
    bn_bwd_pooled_apply_kernel<bf16><<<grid, 256, 0, st>>>((const bf16*)dpool, argmax, (const bf16*)y, (bf16*)dy, coef,
                                                           3 * C, chunks, g.cprw, (int)C, y_gs, p_gs, (int)Hi, (int)Wi, Ho,
                                                           Wo, stats, 4 * C, ps);
  return launch_status("bn_bwd_pooled_apply_kernel");
}
