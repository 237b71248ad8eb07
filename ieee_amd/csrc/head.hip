// The embedding head of IEEE3modalPart in fp32 (it is <0.5 % of the FLOPs):
// grouped small GEMMs on the fp32 MFMA, row-wise BatchNorm (BatchNorm2d over [B,C,1,1]/[B,C,6,1] and
// BatchNorm1d), REM closed form, L2 normalisation, label-smoothed cross entropy, the 3M margin loss
// and the fused SGD-nesterov step.  "Grouped" = up to IEEE_MAX_GROUPS independent problems per
// launch, addressed through pointer tables passed by value (no device-side pointer arrays to manage).
#include "gemm_core.h"

namespace ieee {

constexpr int MAXG = IEEE_MAX_GROUPS;

struct PtrTab { void* p[MAXG]; };

// ------------------------------------------------------------------ grouped fp32 GEMM
// C[g](m,n) = act( alpha * sum_k A[g](m,k) * B[g](n,k) + bias[g](n) ) (+ C if accumulate)
// with generic element strides: A(m,k) = A + m*sam + k*sak, B(n,k) = B + n*sbn + k*sbk.
struct SgemmArgs {
  PtrTab A, B, C, bias;
  int M, N, K;
  int64_t sam, sak, sbn, sbk, ldc;
  float alpha;
  int relu, accumulate;
  // split-K (long K, few tiles: the pooled-vector GEMMs of the head): block z = group * splitk + ks works on
  // k in [ks*kchunk, +kchunk) and stores its raw partial tile to slab[z][M][N]; sgemm_splitk_reduce_kernel adds the
  // slabs in a fixed order and applies alpha / bias / accumulate / relu
  int splitk, kchunk, vec_a, vec_b;
  float* slab;
};

// 64x64 tile, BK = 32, fp32 MFMA 16x16x4.  Operand tiles are fetched as two 16-byte vectors per thread along
// whichever axis is contiguous (vec_* = 1: k contiguous, 2: row contiguous, 0: scalar fallback for unaligned or
// generic strides) and kept one k-tile ahead in registers.
typedef float SgemmStage[64][33];
__device__ __forceinline__ void sgemm_tile(const SgemmArgs& a, int bx, int by, int bz, SgemmStage* As, SgemmStage* Bs) {
  const int g = bz / a.splitk, ks = bz - g * a.splitk;
  const float* A = (const float*)a.A.p[g];
  const float* B = (const float*)a.B.p[g];
  const int m0 = by * 64, n0 = bx * 64;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
  const int kbeg = ks * a.kchunk, kend = min(a.K, kbeg + a.kchunk);
  const int ktiles = (kend - kbeg + 31) / 32;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float ra[8], rb[8];
  // element e (0..7) of this thread: vec 1 -> row = (t>>3) + 32*(e>>2), k = (t&7)*4 + (e&3)
  //                                  vec 2 -> k = (t>>4) + 16*(e>>2), row = (t&15)*4 + (e&3)
  //                                  vec 0 -> row = t & 63, k = (t>>6) + 4*e
  auto fetch = [&](const float* P, int vec, int64_t srow, int64_t sk, int row0, int nrows, int kt, float* r) {
    const int k0 = kbeg + kt * 32;
    if (vec == 1) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int row = row0 + (t >> 3) + 32 * h, k = k0 + (t & 7) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < nrows && k + 3 < kend) v = *(const float4*)(P + row * srow + k);
        else if (row < nrows) {
          if (k < kend) v.x = P[row * srow + k];
          if (k + 1 < kend) v.y = P[row * srow + k + 1];
          if (k + 2 < kend) v.z = P[row * srow + k + 2];
        }
        r[4 * h] = v.x; r[4 * h + 1] = v.y; r[4 * h + 2] = v.z; r[4 * h + 3] = v.w;
      }
    } else if (vec == 2) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int k = k0 + (t >> 4) + 16 * h, row = row0 + (t & 15) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < kend && row + 3 < nrows) v = *(const float4*)(P + k * sk + row);
        else if (k < kend) {
          if (row < nrows) v.x = P[k * sk + row];
          if (row + 1 < nrows) v.y = P[k * sk + row + 1];
          if (row + 2 < nrows) v.z = P[k * sk + row + 2];
        }
        r[4 * h] = v.x; r[4 * h + 1] = v.y; r[4 * h + 2] = v.z; r[4 * h + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int row = row0 + (t & 63), k = k0 + (t >> 6) + 4 * e;
        r[e] = (row < nrows && k < kend) ? P[row * srow + k * sk] : 0.f;
      }
    }
  };
  auto stash = [&](float (*S)[33], int vec, const float* r) {
    if (vec == 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) S[(t >> 3) + 32 * (e >> 2)][(t & 7) * 4 + (e & 3)] = r[e];
    } else if (vec == 2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) S[(t & 15) * 4 + (e & 3)][(t >> 4) + 16 * (e >> 2)] = r[e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) S[t & 63][(t >> 6) + 4 * e] = r[e];
    }
  };
  if (ktiles > 0) {
    fetch(A, a.vec_a, a.sam, a.sak, m0, a.M, 0, ra);
    fetch(B, a.vec_b, a.sbn, a.sbk, n0, a.N, 0, rb);
    stash(As[0], a.vec_a, ra);
    stash(Bs[0], a.vec_b, rb);
  }
  __syncthreads();
  for (int kt = 0; kt < ktiles; ++kt) {
    const int cur = kt & 1;
    const bool has_next = kt + 1 < ktiles;
    if (has_next) {
      fetch(A, a.vec_a, a.sam, a.sak, m0, a.M, kt + 1, ra);
      fetch(B, a.vec_b, a.sbn, a.sbk, n0, a.N, kt + 1, rb);
    }
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      float fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = As[cur][wm * 32 + i * 16 + (lane & 15)][kk * 4 + (lane >> 4)];
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = Bs[cur][wn * 32 + j * 16 + (lane & 15)][kk * 4 + (lane >> 4)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(fb[j], fa[i], acc[i][j]);
    }
    if (has_next) {
      stash(As[cur ^ 1], a.vec_a, ra);
      stash(Bs[cur ^ 1], a.vec_b, rb);
    }
    __syncthreads();
  }
  if (a.splitk > 1) {
    float* S = a.slab + (int64_t)bz * a.M * a.N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int m = m0 + wm * 32 + i * 16 + (lane & 15);
        const int n = n0 + wn * 32 + j * 16 + (lane >> 4) * 4;
        if (m >= a.M) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < a.N) S[(int64_t)m * a.N + n + r] = acc[i][j][r];
      }
    return;
  }
  float* C = (float*)a.C.p[g];
  const float* bias = (const float*)a.bias.p[g];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = m0 + wm * 32 + i * 16 + (lane & 15);
      const int n = n0 + wn * 32 + j * 16 + (lane >> 4) * 4;
      if (m >= a.M) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (n + r >= a.N) continue;
        float v = a.alpha * acc[i][j][r];
        if (bias) v += bias[n + r];
        float* c = C + (int64_t)m * a.ldc + n + r;
        if (a.accumulate) v += *c;
        if (a.relu) v = fmaxf(v, 0.f);
        *c = v;
      }
    }
}

__global__ __launch_bounds__(256) void sgemm_grouped_kernel(SgemmArgs a) {
  __shared__ float As[2][64][33];
  __shared__ float Bs[2][64][33];
  sgemm_tile(a, blockIdx.x, blockIdx.y, blockIdx.z, As, Bs);
}

// TWO uniform problem sets in ONE launch (round 6): the independent GEMM pairs of a head phase -- dW and dX of one Linear /
// 1x1 conv, the reduce_layer applied to the global vector and to the parts -- were two launches of 8-25 us each on the
// latency-bound head chain.  Blocks [0, blocks0) work on set 0, the rest on set 1; each set keeps its own shapes, strides,
// pointer tables and split-K plan, so every output has the bits of the separate launch.
struct PairDims { int gx0, gy0, blocks0, gx1, gy1; };
__global__ __launch_bounds__(256) void sgemm_grouped2_kernel(SgemmArgs a0, SgemmArgs a1, PairDims d) {
  __shared__ float As[2][64][33];
  __shared__ float Bs[2][64][33];
  int b = blockIdx.x;
  if (b < d.blocks0) {
    sgemm_tile(a0, b % d.gx0, (b / d.gx0) % d.gy0, b / (d.gx0 * d.gy0), As, Bs);
  } else {
    b -= d.blocks0;
    sgemm_tile(a1, b % d.gx1, (b / d.gx1) % d.gy1, b / (d.gx1 * d.gy1), As, Bs);
  }
}

__device__ __forceinline__ void sgemm_splitk_reduce_one(const SgemmArgs& a, int g, int64_t i) {
  const int64_t mn = (int64_t)a.M * a.N;
  if (i >= mn) return;
  const float* S = a.slab + (int64_t)g * a.splitk * mn + i;
  float s = 0.f;
  for (int k = 0; k < a.splitk; ++k) s += S[k * mn];
  const int m = (int)(i / a.N), n = (int)(i - (int64_t)m * a.N);
  float v = a.alpha * s;
  const float* bias = (const float*)a.bias.p[g];
  if (bias) v += bias[n];
  float* c = (float*)a.C.p[g] + (int64_t)m * a.ldc + n;
  if (a.accumulate) v += *c;
  if (a.relu) v = fmaxf(v, 0.f);
  *c = v;
}
__global__ __launch_bounds__(256) void sgemm_splitk_reduce_kernel(SgemmArgs a) {
  sgemm_splitk_reduce_one(a, blockIdx.y, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}
// the split-K reductions of a pair (a set without split-K has rblocks = 0): blocks [0, groups0 * rb0) belong to set 0
struct PairReduceDims { int rb0, blocks0, rb1; };
__global__ __launch_bounds__(256) void sgemm_splitk_reduce2_kernel(SgemmArgs a0, SgemmArgs a1, PairReduceDims d) {
  int b = blockIdx.x;
  if (b < d.blocks0) {
    sgemm_splitk_reduce_one(a0, b / d.rb0, (int64_t)(b % d.rb0) * blockDim.x + threadIdx.x);
  } else {
    b -= d.blocks0;
    sgemm_splitk_reduce_one(a1, b / d.rb1, (int64_t)(b % d.rb1) * blockDim.x + threadIdx.x);
  }
}

// zero up to MAXG float spans in one launch (hipMemsetAsync splits every odd-sized span into 3 kernels)
struct ZeroArgs { PtrTab p; int64_t n[MAXG]; };
__global__ void zero_spans_kernel(ZeroArgs a) {
  float* p = (float*)a.p.p[blockIdx.y];
  const int64_t n = a.n[blockIdx.y];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0.f;
}

// column sums (bias gradients): out[g][n] (+)= sum_m X[g](m,n)
struct ColsumArgs {
  PtrTab X, out;
  int M, N;
  int64_t ldx;
  int accumulate;
};
__global__ void colsum_grouped_kernel(ColsumArgs a) {
  const int g = blockIdx.y;
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= a.N) return;
  const float* X = (const float*)a.X.p[g];
  float s = 0.f;
  for (int m = 0; m < a.M; ++m) s += X[(int64_t)m * a.ldx + n];
  float* o = (float*)a.out.p[g] + n;
  *o = a.accumulate ? *o + s : s;
}

// ------------------------------------------------------------------ row-wise BatchNorm (+ReLU)
// x [R][C] (row stride ldx) -> out [R][C] (row stride ldo); statistics over the R rows.
struct RowBnArgs {
  PtrTab x, out, gamma, beta, rmean, rvar, save;   // save[g]: [2][C] mean, invstd
  int R, C;
  int64_t ldx, ldo;
  float momentum, eps;
  int training, relu;
};
// block = RB_TX channels x RB_TY row lanes: the batch is only 64..384 rows, so the row loop is short and many
// blocks (C/16 per problem) run side by side (64 x 4 took 50-85 us per launch on latency alone)
constexpr int RB_TX = 16, RB_TY = 16;
__device__ __forceinline__ float rb_sum(float (*red)[RB_TX], int tx) {
  float s = 0.f;
#pragma unroll
  for (int y = 0; y < RB_TY; ++y) s += red[y][tx];
  return s;
}
__global__ __launch_bounds__(256) void rowbn_fwd_kernel(RowBnArgs a) {
  __shared__ float red[RB_TY][RB_TX];
  const int g = blockIdx.y;
  const int tx = threadIdx.x % RB_TX, ty = threadIdx.x / RB_TX;
  const int c = blockIdx.x * RB_TX + tx;
  const bool ok = c < a.C;
  const float* x = (const float*)a.x.p[g];
  float* out = (float*)a.out.p[g];
  float mean, invstd;
  if (a.training) {
    float s = 0.f;
    if (ok) {
#pragma unroll 4
      for (int r = ty; r < a.R; r += RB_TY) s += x[(int64_t)r * a.ldx + c];
    }
    red[ty][tx] = s;
    __syncthreads();
    mean = rb_sum(red, tx) / a.R;
    __syncthreads();
    float v = 0.f;
    if (ok) {
#pragma unroll 4
      for (int r = ty; r < a.R; r += RB_TY) { const float d = x[(int64_t)r * a.ldx + c] - mean; v += d * d; }
    }
    red[ty][tx] = v;
    __syncthreads();
    const float var = rb_sum(red, tx) / a.R;
    invstd = 1.0f / sqrtf(var + a.eps);
    if (ok && ty == 0) {
      float* rm = (float*)a.rmean.p[g];
      float* rv = (float*)a.rvar.p[g];
      if (rm) {
        const float unb = a.R > 1 ? var * ((float)a.R / (float)(a.R - 1)) : var;
        rm[c] = (1.f - a.momentum) * rm[c] + a.momentum * mean;
        rv[c] = (1.f - a.momentum) * rv[c] + a.momentum * unb;
      }
    }
  } else {
    mean = ok ? ((const float*)a.rmean.p[g])[c] : 0.f;
    invstd = ok ? 1.0f / sqrtf(((const float*)a.rvar.p[g])[c] + a.eps) : 0.f;
  }
  if (!ok) return;
  if (ty == 0 && a.save.p[g]) { float* sv = (float*)a.save.p[g]; sv[c] = mean; sv[a.C + c] = invstd; }
  const float ga = ((const float*)a.gamma.p[g])[c], be = ((const float*)a.beta.p[g])[c];
#pragma unroll 4
  for (int r = ty; r < a.R; r += RB_TY) {
    float v = (x[(int64_t)r * a.ldx + c] - mean) * invstd * ga + be;
    if (a.relu) v = fmaxf(v, 0.f);
    out[(int64_t)r * a.ldo + c] = v;
  }
}

struct RowBnBwdArgs {
  PtrTab dout, out, x, gamma, save, dx, dgamma, dbeta;
  int R, C;
  int64_t lddo, ldo, ldx, lddx;
  int relu, accumulate;
};
__global__ __launch_bounds__(256) void rowbn_bwd_kernel(RowBnBwdArgs a) {
  __shared__ float red[2][RB_TY][RB_TX];
  const int g = blockIdx.y;
  const int tx = threadIdx.x % RB_TX, ty = threadIdx.x / RB_TX;
  const int c = blockIdx.x * RB_TX + tx;
  const bool ok = c < a.C;
  const float* dout = (const float*)a.dout.p[g];
  const float* out = (const float*)a.out.p[g];
  const float* x = (const float*)a.x.p[g];
  const float* sv = (const float*)a.save.p[g];
  const float mean = ok ? sv[c] : 0.f, invstd = ok ? sv[a.C + c] : 0.f;
  float s1 = 0.f, s2 = 0.f;
  if (ok) {
#pragma unroll 4
    for (int r = ty; r < a.R; r += RB_TY) {
      float gq = dout[(int64_t)r * a.lddo + c];
      if ((a.relu & 1) && !(out[(int64_t)r * a.ldo + c] > 0.f)) gq = 0.f;
      s1 += gq;
      s2 += gq * (x[(int64_t)r * a.ldx + c] - mean) * invstd;
    }
  }
  red[0][ty][tx] = s1;
  red[1][ty][tx] = s2;
  __syncthreads();
  s1 = rb_sum(red[0], tx);
  s2 = rb_sum(red[1], tx);
  if (!ok) return;
  const float ga = ((const float*)a.gamma.p[g])[c];
  const bool frozen = (a.relu & 2) != 0;      // eval-mode statistics: a fixed affine map, dx = gamma * invstd * g
  if (ty == 0 && !frozen) {
    float* dg = (float*)a.dgamma.p[g] + c;
    float* db = (float*)a.dbeta.p[g] + c;
    *dg = a.accumulate ? *dg + s2 : s2;
    *db = a.accumulate ? *db + s1 : s1;
  }
  float* dx = (float*)a.dx.p[g];
  const float c1 = frozen ? 0.f : s1 / a.R, c2 = frozen ? 0.f : s2 / a.R;
#pragma unroll 4
  for (int r = ty; r < a.R; r += RB_TY) {
    float gq = dout[(int64_t)r * a.lddo + c];
    if ((a.relu & 1) && !(out[(int64_t)r * a.ldo + c] > 0.f)) gq = 0.f;
    const float xh = (x[(int64_t)r * a.ldx + c] - mean) * invstd;
    dx[(int64_t)r * a.lddx + c] = ga * invstd * (gq - c1 - xh * c2);
  }
}

// ------------------------------------------------------------------ small elementwise pieces
// CA: hs[b][j] = h[b][j] + h[B+b][j]   (relu already applied by the GEMM epilogue)
__global__ void add_halves_kernel(const float* h, float* hs, int64_t half, int64_t gs_in, int64_t gs_out) {
  const int z = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (int64_t)gridDim.x * blockDim.x)
    hs[z * gs_out + i] = h[z * gs_in + i] + h[z * gs_in + half + i];
}
__global__ void sigmoid_kernel(float* x, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    x[i] = 1.0f / (1.0f + expf(-x[i]));
}
// dz = datt * att * (1 - att)
__global__ void sigmoid_bwd_kernel(const float* datt, const float* att, float* dz, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dz[i] = datt[i] * att[i] * (1.f - att[i]);
}
// dh[b] = dhs[b]*[h[b]>0], dh[B+b] = dhs[b]*[h[B+b]>0]
__global__ void add_halves_bwd_kernel(const float* dhs, const float* h, float* dh, int64_t half, int64_t gs_half,
                                      int64_t gs_full) {
  const int z = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (int64_t)gridDim.x * blockDim.x) {
    const float d = dhs[z * gs_half + i];
    dh[z * gs_full + i] = h[z * gs_full + i] > 0.f ? d : 0.f;
    dh[z * gs_full + half + i] = h[z * gs_full + half + i] > 0.f ? d : 0.f;
  }
}

// REM closed form (ieee3modalPart.py:60-80): out[b][i][:] = part[b][i][:] + 2*param * r[b][:]
__global__ void rem_fwd_kernel(const float* part, const float* r, const float* param, int64_t param_gs, float* out,
                               int B, int parts, int D) {
  const int z = blockIdx.y;
  const float s = 2.0f * param[z * param_gs];
  const int64_t n = (int64_t)B * parts * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % D);
    const int b = (int)(i / ((int64_t)parts * D));
    out[z * n + i] = part[z * n + i] + s * r[((int64_t)z * B + b) * D + k];
  }
}
// dr[b][k] = 2*param * sum_i dout[b][i][k];  dparam partial[b] = 2 * sum_{i,k} dout[b][i][k]*r[b][k]
__global__ __launch_bounds__(256) void rem_bwd_kernel(const float* dout, const float* r, const float* param,
                                                      int64_t param_gs, float* dr, float* dparam_partial, int B,
                                                      int parts, int D) {
  __shared__ float red[256];
  const int z = blockIdx.y, b = blockIdx.x;
  const float s = 2.0f * param[z * param_gs];
  float accp = 0.f;
  for (int k = threadIdx.x; k < D; k += blockDim.x) {
    float sum = 0.f;
    for (int i = 0; i < parts; ++i) sum += dout[(((int64_t)z * B + b) * parts + i) * D + k];
    dr[((int64_t)z * B + b) * D + k] = s * sum;
    accp += sum * r[((int64_t)z * B + b) * D + k];
  }
  red[threadIdx.x] = accp;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) dparam_partial[(int64_t)z * B + b] = 2.0f * red[0];
}
__global__ void rem_dparam_kernel(const float* partial, float* dparam, int64_t grad_gs, int B, int accumulate) {
  const int z = threadIdx.x;
  if (z >= 3) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += partial[(int64_t)z * B + b];
  float* o = dparam + z * grad_gs;
  *o = accumulate ? *o + s : s;
}

// F.normalize(x, p=2, dim=1), eps 1e-12: one wave per row
__global__ void l2norm_fwd_kernel(const float* x, float* y, float* norms, int64_t rows, int D) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int k = lane; k < D; k += 64) { const float v = x[row * D + k]; s += v * v; }
  s = wave_sum(s);
  const float nrm = fmaxf(sqrtf(s), 1e-12f);
  for (int k = lane; k < D; k += 64) y[row * D + k] = x[row * D + k] / nrm;
  if (lane == 0) norms[row] = nrm;
}
// dx = (dy - y * (y.dy)) / norm   (rows with norm clamped at eps: dx = dy/eps, like autograd of clamp_min)
__global__ void l2norm_bwd_kernel(const float* dy, const float* y, const float* norms, float* dx, int64_t rows, int D,
                                  int accumulate) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int k = lane; k < D; k += 64) s += dy[row * D + k] * y[row * D + k];
  s = wave_sum(s);
  const float nrm = norms[row];
  const float dot = nrm > 1e-12f ? s : 0.f;
  for (int k = lane; k < D; k += 64) {
    const float v = (dy[row * D + k] - y[row * D + k] * dot) / nrm;
    dx[row * D + k] = accumulate ? dx[row * D + k] + v : v;
  }
}

// ------------------------------------------------------------------ losses
// label-smoothed CE (torchreid/losses/cross_entropy_loss.py:36-50) over `heads` logits tensors [B][C]:
// rowloss[h][b] = -sum_c t_c * logp_c, t = (1-eps)*onehot + eps/C; dlogits = (softmax - t) * gscale / B
__global__ void ce_rows_kernel(const float* logits, const int64_t* target, float* rowloss, int* correct,
                               float* dlogits, int heads, int B, int C, float eps, float gscale) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)heads * B) return;
  const int lane = threadIdx.x & 63;
  const int b = (int)(row % B);
  const float* x = logits + row * C;
  const int y = (int)target[b];
  float mx = -INFINITY;
  int am = 0;
  for (int c = lane; c < C; c += 64) { const float v = x[c]; if (v > mx) { mx = v; am = c; } }
  // wave arg-max (first index on ties)
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(mx, o);
    const int oi = __shfl_xor(am, o);
    if (ov > mx || (ov == mx && oi < am)) { mx = ov; am = oi; }
  }
  float se = 0.f, sx = 0.f;
  for (int c = lane; c < C; c += 64) { se += expf(x[c] - mx); sx += x[c]; }
  se = wave_sum(se);
  sx = wave_sum(sx);
  const float lse = mx + logf(se);
  // sum_c logp_c = sx - C*lse ; logp_y = x[y] - lse
  const float logpy = x[y] - lse;
  if (lane == 0) {
    rowloss[row] = -(1.f - eps) * logpy - (eps / C) * (sx - C * lse);
    correct[row] = am == y ? 1 : 0;
  }
  if (dlogits) {
    const float sc = gscale / B;
    for (int c = lane; c < C; c += 64) {
      const float p = expf(x[c] - lse);
      const float tt = (c == y ? (1.f - eps) : 0.f) + eps / C;
      dlogits[row * C + c] = (p - tt) * sc;
    }
  }
}
// per-head mean loss and top-1 % accuracy
__global__ void ce_heads_kernel(const float* rowloss, const int* correct, float* head_loss, float* head_acc,
                                int heads, int B) {
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= heads) return;
  float s = 0.f;
  int k = 0;
  for (int b = 0; b < B; ++b) { s += rowloss[(int64_t)h * B + b]; k += correct[(int64_t)h * B + b]; }
  head_loss[h] = s / B;
  head_acc[h] = 100.0f * k / B;
}

// 3M loss (torchreid/losses/multi_modal_margin_loss_new.py:19-40), feats [3][B][D] in the order R,N,T.
// One block per identity chunk (torch.chunk semantics: ceil(B/label_num) rows each); every block recounts
// label_num itself (B is tiny), writes its chunk's term and that chunk's rows of dfeats.  A second, single
// thread pass sums the terms in chunk order: deterministic.
// out[0] = loss, out[1] = label_num, out[2] = number of chunks torch.chunk yields (< label_num => the
// reference raises IndexError).
__global__ __launch_bounds__(256) void margin3m_kernel(const float* feats, const int64_t* pids, float* dfeats,
                                                       float* terms, int B, int D, float margin, float gscale) {
  __shared__ float red[3][256];
  __shared__ int s_cnt[256];
  __shared__ int s_sel;
  __shared__ float s_sign;
  const int t = threadIdx.x;
  int cnt = 0;
  for (int i = t; i < B; i += 256) {
    bool seen = false;
    for (int j = 0; j < i; ++j) if (pids[j] == pids[i]) { seen = true; break; }
    cnt += seen ? 0 : 1;
  }
  s_cnt[t] = cnt;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (t < o) s_cnt[t] += s_cnt[t + o]; __syncthreads(); }
  const int n = s_cnt[0];
  const int csize = (B + n - 1) / n;            // torch.chunk: ceil(B / chunks) rows per chunk
  const int nchunks = (B + csize - 1) / csize;
  const int nuse = nchunks < n ? nchunks : n;
  const int ch = blockIdx.x;
  const int64_t BD = (int64_t)B * D;
  if (ch == 0 && t == 0) { terms[B] = (float)n; terms[B + 1] = (float)nchunks; terms[B + 2] = (float)nuse; }
  const int r0 = ch * csize, r1 = min(B, r0 + csize), rows = r1 - r0;
  if (rows <= 0) { if (t == 0) terms[ch] = 0.f; return; }
  if (ch >= nuse) {   // rows beyond the chunks the reference loops over: no loss term, zero gradient
    if (t == 0) terms[ch] = 0.f;
    if (dfeats)
      for (int m = 0; m < 3; ++m)
        for (int64_t i = t; i < (int64_t)rows * D; i += 256) dfeats[m * BD + (int64_t)r0 * D + i] = 0.f;
    return;
  }
  float d12 = 0.f, d23 = 0.f, d13 = 0.f;
  for (int k = t; k < D; k += 256) {
    float c1 = 0.f, c2 = 0.f, c3 = 0.f;
    for (int r = r0; r < r1; ++r) { c1 += feats[r * D + k]; c2 += feats[BD + r * D + k]; c3 += feats[2 * BD + r * D + k]; }
    c1 /= rows; c2 /= rows; c3 /= rows;
    d12 += (c1 - c2) * (c1 - c2);
    d23 += (c2 - c3) * (c2 - c3);
    d13 += (c1 - c3) * (c1 - c3);
  }
  red[0][t] = d12; red[1][t] = d23; red[2][t] = d13;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) { red[0][t] += red[0][t + o]; red[1][t] += red[1][t + o]; red[2][t] += red[2][t + o]; }
    __syncthreads();
  }
  if (t == 0) {
    // python max(a, b, c) keeps the FIRST maximal argument: order (1,2), (2,3), (1,3)  (:35)
    const float a1 = fabsf(margin - red[0][0]), a2 = fabsf(margin - red[1][0]), a3 = fabsf(margin - red[2][0]);
    int sel = 0; float best = a1;
    if (a2 > best) { best = a2; sel = 1; }
    if (a3 > best) { best = a3; sel = 2; }
    const float dsel = sel == 0 ? red[0][0] : (sel == 1 ? red[1][0] : red[2][0]);
    const float x = margin - dsel;
    s_sel = sel;
    s_sign = x > 0.f ? -1.f : (x < 0.f ? 1.f : 0.f);   // d|m-d|/dd
    terms[ch] = best;
  }
  __syncthreads();
  if (dfeats) {
    const int sel = s_sel;
    const int ia = sel == 1 ? 1 : 0;          // pair (a,b): (0,1), (1,2), (0,2)
    const int ib = sel == 0 ? 1 : 2;
    const int ic = 3 - ia - ib;               // the modality not in the selected pair gets zero
    const float coef = s_sign * 2.0f / rows * gscale;
    for (int k = t; k < D; k += 256) {
      float ca = 0.f, cb = 0.f;
      for (int r = r0; r < r1; ++r) { ca += feats[ia * BD + r * D + k]; cb += feats[ib * BD + r * D + k]; }
      const float diff = (ca - cb) / rows;
      for (int r = r0; r < r1; ++r) {
        dfeats[ia * BD + r * D + k] = coef * diff;
        dfeats[ib * BD + r * D + k] = -coef * diff;
        dfeats[ic * BD + r * D + k] = 0.f;
      }
    }
  }
}
__global__ void margin3m_sum_kernel(const float* terms, float* out, int B) {
  if (threadIdx.x != 0) return;
  const int nuse = (int)terms[B + 2];
  float loss = 0.f;
  for (int c = 0; c < nuse; ++c) loss += terms[c];
  out[0] = loss; out[1] = terms[B]; out[2] = terms[B + 1];
}

// torch.optim.SGD(momentum, weight_decay, dampening=0, nesterov=True) (reference optim/optimizer.py:130-138)
__device__ __forceinline__ void sgd_one(float& w, float g, float& buf, float lr, float momentum, float wd, int nesterov) {
  float d = g + wd * w;
  if (momentum != 0.f) {
    const float b = momentum * buf + d;
    buf = b;
    d = nesterov ? d + momentum * b : b;
  }
  w = w - lr * d;
}
// 16 bytes per lane when the three buffers allow it (20 B/param of traffic: the 4-byte form needed 4x the memory
// instructions); per-element arithmetic is identical in both forms
// shadow (may be NULL; round 6): the bf16 image of the UPDATED parameters, element for element (+2 B/param of writes).  A 1x1
// convolution's forward GEMM operand Wf[co][ci] IS that image of its OIHW weight, so the executor reads it straight from
// there and the once-per-step weight packing loses those tensors (64 % of the conv parameters: 4 B read + 2 B written each).
// skip (may be NULL; round 6): the range-guard words of the executor (device memory; [0] / [1] = a forward / backward
// BatchNorm tile sum of THIS step was clamped).  Once either is set the step's gradients are not the reference's: the update
// is skipped -- parameters, momentum and shadow stay as they were -- like an overflow step of a loss scaler.
__global__ void sgd_nesterov_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                    int64_t n, float lr, float momentum, float wd, int nesterov, int vec, bf16* __restrict__ shadow,
                                    const int* __restrict__ skip) {
  if (skip != nullptr && (skip[0] | skip[1]) != 0) return;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
  int64_t done = 0;
  if (vec) {
    const int64_t n4 = n >> 2;
    for (int64_t i = tid; i < n4; i += nth) {
      float4 w = ((float4*)p)[i];
      const float4 gg = ((const float4*)g)[i];
      float4 b = momentum != 0.f ? ((float4*)buf)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      sgd_one(w.x, gg.x, b.x, lr, momentum, wd, nesterov);
      sgd_one(w.y, gg.y, b.y, lr, momentum, wd, nesterov);
      sgd_one(w.z, gg.z, b.z, lr, momentum, wd, nesterov);
      sgd_one(w.w, gg.w, b.w, lr, momentum, wd, nesterov);
      if (momentum != 0.f) ((float4*)buf)[i] = b;
      ((float4*)p)[i] = w;
      if (shadow != nullptr) ((uint2*)shadow)[i] = make_uint2(Vec16<bf16>::pk(w.x, w.y), Vec16<bf16>::pk(w.z, w.w));
    }
    done = n4 << 2;
  }
  for (int64_t i = done + tid; i < n; i += nth) {
    float w = p[i], b = momentum != 0.f ? buf[i] : 0.f;
    sgd_one(w, g[i], b, lr, momentum, wd, nesterov);
    if (momentum != 0.f) buf[i] = b;
    p[i] = w;
    if (shadow != nullptr) shadow[i] = (bf16)w;
  }
}

static int ewb(int64_t n) {
  static const int64_t cap = getenv("IEEE_HEAD_EW_BLOCKS") ? atoll(getenv("IEEE_HEAD_EW_BLOCKS")) : 2048;
  int64_t b = (n + 255) / 256;
  return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

}  // namespace ieee

using namespace ieee;

static int fill_tab(PtrTab* t, const void* const* src, int groups) {
  for (int i = 0; i < MAXG; ++i) t->p[i] = i < groups && src ? (void*)src[i] : nullptr;
  return 0;
}

static bool aligned16(const void* const* tab, int groups) {
  for (int i = 0; i < groups; ++i)
    if (((uintptr_t)tab[i] & 15) != 0) return false;
  return true;
}

static int sgemm_fill(SgemmArgs& a, int64_t groups, const void* const* A, const void* const* B, void* const* C,
                      const void* const* bias, int64_t M, int64_t N, int64_t K, int64_t sam, int64_t sak, int64_t sbn,
                      int64_t sbk, int64_t ldc, float alpha, int relu, int accumulate, void* work, int64_t work_bytes);

extern "C" int ieee_sgemm_grouped_ws(int64_t groups, const void* const* A, const void* const* B, void* const* C,
                                     const void* const* bias, int64_t M, int64_t N, int64_t K, int64_t sam, int64_t sak,
                                     int64_t sbn, int64_t sbk, int64_t ldc, float alpha, int relu, int accumulate,
                                     void* work, int64_t work_bytes, void* stream) {
  SgemmArgs a;
  IEEE_TRY(sgemm_fill(a, groups, A, B, C, bias, M, N, K, sam, sak, sbn, sbk, ldc, alpha, relu, accumulate, work, work_bytes));
  dim3 grid(cdiv(N, 64), cdiv(M, 64), (unsigned)(groups * a.splitk));
  sgemm_grouped_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(a);
  IEEE_TRY(launch_status("sgemm_grouped_kernel"));
  if (a.splitk > 1) {
    sgemm_splitk_reduce_kernel<<<dim3(cdiv(M * N, 256), (unsigned)groups), 256, 0, (hipStream_t)stream>>>(a);
    return launch_status("sgemm_splitk_reduce_kernel");
  }
  return IEEE_OK;
}

extern "C" int ieee_sgemm_grouped_pair_ws(const ieee_sgemm_set* s0, const ieee_sgemm_set* s1, void* work, int64_t work_bytes,
                                          void* stream) {
  IEEE_REQUIRE(s0 && s1, "sgemm_grouped_pair: null problem set");
  // each set plans its split-K against HALF of the workspace: a caller that doubles the workspace it gave the single calls
  // gets exactly their plans, hence their bits
  const int64_t half = work ? (work_bytes / 2) & ~(int64_t)255 : 0;
  SgemmArgs a0, a1;
  IEEE_TRY(sgemm_fill(a0, s0->groups, s0->A, s0->B, s0->C, s0->bias, s0->M, s0->N, s0->K, s0->sam, s0->sak, s0->sbn, s0->sbk, s0->ldc,
                      s0->alpha, s0->relu, s0->accumulate, work, half));
  IEEE_TRY(sgemm_fill(a1, s1->groups, s1->A, s1->B, s1->C, s1->bias, s1->M, s1->N, s1->K, s1->sam, s1->sak, s1->sbn, s1->sbk, s1->ldc,
                      s1->alpha, s1->relu, s1->accumulate, work ? (char*)work + half : nullptr, half));
  PairDims d;
  d.gx0 = cdiv(s0->N, 64); d.gy0 = cdiv(s0->M, 64); d.blocks0 = d.gx0 * d.gy0 * (int)(s0->groups * a0.splitk);
  d.gx1 = cdiv(s1->N, 64); d.gy1 = cdiv(s1->M, 64);
  const int blocks1 = d.gx1 * d.gy1 * (int)(s1->groups * a1.splitk);
  sgemm_grouped2_kernel<<<dim3((unsigned)(d.blocks0 + blocks1)), 256, 0, (hipStream_t)stream>>>(a0, a1, d);
  IEEE_TRY(launch_status("sgemm_grouped2_kernel"));
  if (a0.splitk > 1 || a1.splitk > 1) {
    PairReduceDims r;
    r.rb0 = a0.splitk > 1 ? cdiv(s0->M * s0->N, 256) : 1;
    r.blocks0 = a0.splitk > 1 ? r.rb0 * (int)s0->groups : 0;
    r.rb1 = a1.splitk > 1 ? cdiv(s1->M * s1->N, 256) : 1;
    const int rblocks1 = a1.splitk > 1 ? r.rb1 * (int)s1->groups : 0;
    sgemm_splitk_reduce2_kernel<<<dim3((unsigned)(r.blocks0 + rblocks1)), 256, 0, (hipStream_t)stream>>>(a0, a1, r);
    return launch_status("sgemm_splitk_reduce2_kernel");
  }
  return IEEE_OK;
}

static int sgemm_fill(SgemmArgs& a, int64_t groups, const void* const* A, const void* const* B, void* const* C,
                      const void* const* bias, int64_t M, int64_t N, int64_t K, int64_t sam, int64_t sak, int64_t sbn,
                      int64_t sbk, int64_t ldc, float alpha, int relu, int accumulate, void* work, int64_t work_bytes) {
  IEEE_REQUIRE(groups >= 1 && groups <= MAXG, "sgemm_grouped: groups %ld out of range [1,%d]", (long)groups, MAXG);
  IEEE_REQUIRE(A && B && C, "sgemm_grouped: null pointer table");
  IEEE_REQUIRE(M > 0 && N > 0 && K > 0, "sgemm_grouped: empty problem");
  fill_tab(&a.A, A, (int)groups);
  fill_tab(&a.B, B, (int)groups);
  fill_tab(&a.C, (const void* const*)C, (int)groups);
  fill_tab(&a.bias, bias, (int)groups);
  a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.sam = sam; a.sak = sak; a.sbn = sbn; a.sbk = sbk; a.ldc = ldc;
  a.alpha = alpha; a.relu = relu; a.accumulate = accumulate;
  // 16-byte loads along the contiguous axis when every group pointer and the other stride allow it
  a.vec_a = (sak == 1 && sam % 4 == 0 && aligned16(A, (int)groups)) ? 1
            : (sam == 1 && sak % 4 == 0 && aligned16(A, (int)groups)) ? 2 : 0;
  a.vec_b = (sbk == 1 && sbn % 4 == 0 && aligned16(B, (int)groups)) ? 1
            : (sbn == 1 && sbk % 4 == 0 && aligned16(B, (int)groups)) ? 2 : 0;
  const int64_t tiles = cdiv(N, 64) * cdiv(M, 64) * groups;
  int64_t splitk = 1;
  if (work && tiles < 1024 && K >= 256) {   // aim at ~4 workgroups per CU: the k-loop is latency bound
    splitk = cdiv(1024, tiles);
    if (splitk > K / 128) splitk = K / 128;
    if (splitk > 32) splitk = 32;
    while (splitk > 1 && splitk * groups * M * N * 4 > work_bytes) --splitk;
    if (splitk < 1) splitk = 1;
  }
  a.kchunk = (int)(cdiv(cdiv(K, splitk), 32) * 32);
  a.splitk = (int)cdiv(K, a.kchunk);
  a.slab = (float*)work;
  return IEEE_OK;
}

extern "C" int ieee_sgemm_grouped(int64_t groups, const void* const* A, const void* const* B, void* const* C,
                                  const void* const* bias, int64_t M, int64_t N, int64_t K, int64_t sam, int64_t sak,
                                  int64_t sbn, int64_t sbk, int64_t ldc, float alpha, int relu, int accumulate,
                                  void* stream) {
  return ieee_sgemm_grouped_ws(groups, A, B, C, bias, M, N, K, sam, sak, sbn, sbk, ldc, alpha, relu, accumulate, nullptr,
                               0, stream);
}

extern "C" int ieee_zero_spans(int64_t count, void* const* ptrs, const int64_t* floats, void* stream) {
  IEEE_REQUIRE(count >= 1 && count <= MAXG && ptrs && floats, "zero_spans: bad arguments");
  ZeroArgs a;
  fill_tab(&a.p, (const void* const*)ptrs, (int)count);
  int64_t most = 1;
  for (int i = 0; i < MAXG; ++i) { a.n[i] = i < count ? floats[i] : 0; most = a.n[i] > most ? a.n[i] : most; }
  zero_spans_kernel<<<dim3(ewb(most), (unsigned)count), 256, 0, (hipStream_t)stream>>>(a);
  return launch_status("zero_spans_kernel");
}

extern "C" int ieee_colsum_grouped(int64_t groups, const void* const* X, void* const* out, int64_t M, int64_t N,
                                   int64_t ldx, int accumulate, void* stream) {
  IEEE_REQUIRE(groups >= 1 && groups <= MAXG && X && out, "colsum_grouped: bad arguments");
  ColsumArgs a;
  fill_tab(&a.X, X, (int)groups);
  fill_tab(&a.out, (const void* const*)out, (int)groups);
  a.M = (int)M; a.N = (int)N; a.ldx = ldx; a.accumulate = accumulate;
  colsum_grouped_kernel<<<dim3(cdiv(N, 128), (unsigned)groups), 128, 0, (hipStream_t)stream>>>(a);
  return launch_status("colsum_grouped_kernel");
}

extern "C" int ieee_rowbn_fwd(int64_t groups, const void* const* x, void* const* out, const void* const* gamma,
                              const void* const* beta, void* const* running_mean, void* const* running_var,
                              void* const* save, int64_t R, int64_t C, int64_t ldx, int64_t ldo, float momentum,
                              float eps, int training, int relu, void* stream) {
  IEEE_REQUIRE(groups >= 1 && groups <= MAXG && x && out && gamma && beta, "rowbn_fwd: bad arguments");
  IEEE_REQUIRE(training || (running_mean && running_var), "rowbn_fwd: eval mode needs running stats");
  RowBnArgs a;
  fill_tab(&a.x, x, (int)groups);
  fill_tab(&a.out, (const void* const*)out, (int)groups);
  fill_tab(&a.gamma, gamma, (int)groups);
  fill_tab(&a.beta, beta, (int)groups);
  fill_tab(&a.rmean, (const void* const*)running_mean, (int)groups);
  fill_tab(&a.rvar, (const void* const*)running_var, (int)groups);
  fill_tab(&a.save, (const void* const*)save, (int)groups);
  a.R = (int)R; a.C = (int)C; a.ldx = ldx; a.ldo = ldo; a.momentum = momentum; a.eps = eps;
  a.training = training; a.relu = relu;
  rowbn_fwd_kernel<<<dim3(cdiv(C, RB_TX), (unsigned)groups), 256, 0, (hipStream_t)stream>>>(a);
  return launch_status("rowbn_fwd_kernel");
}

extern "C" int ieee_rowbn_bwd(int64_t groups, const void* const* dout, const void* const* out, const void* const* x,
                              const void* const* gamma, const void* const* save, void* const* dx,
                              void* const* dgamma, void* const* dbeta, int64_t R, int64_t C, int64_t lddo,
                              int64_t ldo, int64_t ldx, int64_t lddx, int relu, int accumulate, void* stream) {
  IEEE_REQUIRE(groups >= 1 && groups <= MAXG && dout && out && x && gamma && save && dx && dgamma && dbeta,
               "rowbn_bwd: bad arguments");
  RowBnBwdArgs a;
  fill_tab(&a.dout, dout, (int)groups);
  fill_tab(&a.out, out, (int)groups);
  fill_tab(&a.x, x, (int)groups);
  fill_tab(&a.gamma, gamma, (int)groups);
  fill_tab(&a.save, save, (int)groups);
  fill_tab(&a.dx, (const void* const*)dx, (int)groups);
  fill_tab(&a.dgamma, (const void* const*)dgamma, (int)groups);
  fill_tab(&a.dbeta, (const void* const*)dbeta, (int)groups);
  a.R = (int)R; a.C = (int)C; a.lddo = lddo; a.ldo = ldo; a.ldx = ldx; a.lddx = lddx;
  a.relu = relu; a.accumulate = accumulate;
  rowbn_bwd_kernel<<<dim3(cdiv(C, RB_TX), (unsigned)groups), 256, 0, (hipStream_t)stream>>>(a);
  return launch_status("rowbn_bwd_kernel");
}

extern "C" int ieee_ca_mix_fwd(const float* h, float* hs, int64_t B, int64_t hidden, void* stream) {
  IEEE_REQUIRE(h && hs, "ca_mix_fwd: null pointer");
  add_halves_kernel<<<dim3(ewb(B * hidden), 3), 256, 0, (hipStream_t)stream>>>(h, hs, B * hidden, 2 * B * hidden,
                                                                                B * hidden);
  return launch_status("add_halves_kernel");
}
extern "C" int ieee_ca_mix_bwd(const float* dhs, const float* h, float* dh, int64_t B, int64_t hidden, void* stream) {
  IEEE_REQUIRE(dhs && h && dh, "ca_mix_bwd: null pointer");
  add_halves_bwd_kernel<<<dim3(ewb(B * hidden), 3), 256, 0, (hipStream_t)stream>>>(dhs, h, dh, B * hidden, B * hidden,
                                                                                    2 * B * hidden);
  return launch_status("add_halves_bwd_kernel");
}
extern "C" int ieee_sigmoid_fwd(float* x, int64_t n, void* stream) {
  IEEE_REQUIRE(x, "sigmoid_fwd: null pointer");
  sigmoid_kernel<<<ewb(n), 256, 0, (hipStream_t)stream>>>(x, n);
  return launch_status("sigmoid_kernel");
}
extern "C" int ieee_sigmoid_bwd(const float* datt, const float* att, float* dz, int64_t n, void* stream) {
  IEEE_REQUIRE(datt && att && dz, "sigmoid_bwd: null pointer");
  sigmoid_bwd_kernel<<<ewb(n), 256, 0, (hipStream_t)stream>>>(datt, att, dz, n);
  return launch_status("sigmoid_bwd_kernel");
}

extern "C" int ieee_rem_fwd(const float* part, const float* r, const float* param, int64_t param_gs, float* out,
                            int64_t B, int64_t parts, int64_t D, void* stream) {
  IEEE_REQUIRE(part && r && param && out, "rem_fwd: null pointer");
  rem_fwd_kernel<<<dim3(ewb(B * parts * D), 3), 256, 0, (hipStream_t)stream>>>(part, r, param, param_gs, out, (int)B,
                                                                                (int)parts, (int)D);
  return launch_status("rem_fwd_kernel");
}
extern "C" int ieee_rem_bwd(const float* dout, const float* r, const float* param, int64_t param_gs, float* dr,
                            float* dparam, int64_t grad_gs, float* work, int64_t B, int64_t parts, int64_t D,
                            int accumulate, void* stream) {
  IEEE_REQUIRE(dout && r && param && dr && dparam && work, "rem_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  rem_bwd_kernel<<<dim3((unsigned)B, 3), 256, 0, st>>>(dout, r, param, param_gs, dr, work, (int)B, (int)parts, (int)D);
  IEEE_TRY(launch_status("rem_bwd_kernel"));
  rem_dparam_kernel<<<1, 64, 0, st>>>(work, dparam, grad_gs, (int)B, accumulate);
  return launch_status("rem_dparam_kernel");
}

extern "C" int ieee_l2norm_fwd(const float* x, float* y, float* norms, int64_t rows, int64_t D, void* stream) {
  IEEE_REQUIRE(x && y && norms, "l2norm_fwd: null pointer");
  l2norm_fwd_kernel<<<cdiv(rows, 4), 256, 0, (hipStream_t)stream>>>(x, y, norms, rows, (int)D);
  return launch_status("l2norm_fwd_kernel");
}
extern "C" int ieee_l2norm_bwd(const float* dy, const float* y, const float* norms, float* dx, int64_t rows,
                               int64_t D, int accumulate, void* stream) {
  IEEE_REQUIRE(dy && y && norms && dx, "l2norm_bwd: null pointer");
  l2norm_bwd_kernel<<<cdiv(rows, 4), 256, 0, (hipStream_t)stream>>>(dy, y, norms, dx, rows, (int)D, accumulate);
  return launch_status("l2norm_bwd_kernel");
}

extern "C" int ieee_ce_ls_fwd_bwd(const float* logits, const int64_t* targets, float* dlogits, float* head_loss,
                                  float* head_acc, float* work, int64_t heads, int64_t B, int64_t C, float eps,
                                  float grad_scale, void* stream) {
  IEEE_REQUIRE(logits && targets && head_loss && head_acc && work, "ce_ls_fwd_bwd: null pointer");
  IEEE_REQUIRE(heads >= 1 && B >= 1 && C >= 1, "ce_ls_fwd_bwd: empty input");
  hipStream_t st = (hipStream_t)stream;
  float* rowloss = work;
  int* correct = (int*)(work + heads * B);
  ce_rows_kernel<<<cdiv(heads * B, 4), 256, 0, st>>>(logits, targets, rowloss, correct, dlogits, (int)heads, (int)B,
                                                     (int)C, eps, grad_scale);
  IEEE_TRY(launch_status("ce_rows_kernel"));
  ce_heads_kernel<<<cdiv(heads, 64), 64, 0, st>>>(rowloss, correct, head_loss, head_acc, (int)heads, (int)B);
  return launch_status("ce_heads_kernel");
}

extern "C" int ieee_margin3m_fwd_bwd(const float* feats, const int64_t* pids, float* dfeats, float* out3, float* work,
                                     int64_t B, int64_t D, float margin, float grad_scale, void* stream) {
  IEEE_REQUIRE(feats && pids && out3 && work, "margin3m_fwd_bwd: null pointer");
  IEEE_REQUIRE(B >= 1 && D >= 1 && B <= 65535, "margin3m_fwd_bwd: batch out of range");
  hipStream_t st = (hipStream_t)stream;
  margin3m_kernel<<<(unsigned)B, 256, 0, st>>>(feats, pids, dfeats, work, (int)B, (int)D, margin, grad_scale);
  IEEE_TRY(launch_status("margin3m_kernel"));
  margin3m_sum_kernel<<<1, 64, 0, st>>>(work, out3, (int)B);
  return launch_status("margin3m_sum_kernel");
}

// torch.optim.Adam (L2 weight decay folded into the gradient, bias-corrected, optional AMSGrad), single-tensor
// semantics of torch/optim/adam.py::_single_tensor_adam: step_size = lr / (1 - b1^t),
// denom = sqrt(v or vmax) / sqrt(1 - b2^t) + eps
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, float* __restrict__ vmax, int64_t n, float lr, float b1, float b2,
                            float eps, float wd, float bc1, float bc2_sqrt) {
  const float step_size = lr / bc1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float w = p[i];
    const float d = g[i] + wd * w;
    const float mi = m[i] + (d - m[i]) * (1.f - b1);          // lerp form, as torch does (exp_avg.lerp_)
    const float vi = b2 * v[i] + (1.f - b2) * d * d;
    m[i] = mi;
    v[i] = vi;
    float vv = vi;
    if (vmax != nullptr) {
      vv = fmaxf(vmax[i], vi);
      vmax[i] = vv;
    }
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p[i] = w - step_size * (mi / denom);
  }
}

extern "C" int ieee_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                              float* max_exp_avg_sq, int64_t n, float lr, float beta1, float beta2, float eps,
                              float weight_decay, int64_t step, void* stream) {
  IEEE_REQUIRE(params && grads && exp_avg && exp_avg_sq, "adam_step: null pointer");
  IEEE_REQUIRE(step >= 1, "adam_step: step counts from 1");
  if (n <= 0) return IEEE_OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  adam_kernel<<<ewb(n), 256, 0, (hipStream_t)stream>>>(params, grads, exp_avg, exp_avg_sq, max_exp_avg_sq, n, lr, beta1,
                                                       beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2));
  return launch_status("adam_kernel");
}

// running statistics of a clamped forward are not the reference's either: buffers <- backup when flags[0] is set, else
// backup <- buffers (the state the next step may have to fall back to).  One short launch per step, off the critical path.
__global__ void guard_buffers_kernel(const int* __restrict__ flags, float* __restrict__ buffers, float* __restrict__ backup, int64_t n) {
  const bool restore = flags[0] != 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (restore) buffers[i] = backup[i]; else backup[i] = buffers[i];
  }
}
extern "C" int ieee_guard_buffers(const int* flags, float* buffers, float* backup, int64_t n, void* stream) {
  IEEE_REQUIRE(flags && buffers && backup && n >= 0, "guard_buffers: bad arguments");
  if (n == 0) return IEEE_OK;
  guard_buffers_kernel<<<ewb(n), 256, 0, (hipStream_t)stream>>>(flags, buffers, backup, n);
  return launch_status("guard_buffers_kernel");
}

extern "C" int ieee_sgd_nesterov_step_ex(float* params, const float* grads, float* momentum_buf, int64_t n, float lr,
                                         float momentum, float weight_decay, int nesterov, void* shadow_bf16,
                                         const int* skip_flags, void* stream) {
  IEEE_REQUIRE(params && grads && (momentum == 0.f || momentum_buf), "sgd_nesterov_step: null pointer");
  if (n <= 0) return IEEE_OK;
  // (the shadow is addressed like the parameters: element i at shadow + i; the 16-byte form needs its 8-byte stores aligned)
  const int vec = ((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)momentum_buf) & 15) == 0 && ((uintptr_t)shadow_bf16 & 7) == 0) ? 1 : 0;
  sgd_nesterov_kernel<<<ewb(vec ? (n + 3) / 4 : n), 256, 0, (hipStream_t)stream>>>(params, grads, momentum_buf, n, lr,
                                                                                   momentum, weight_decay, nesterov, vec,
                                                                                   (bf16*)shadow_bf16, skip_flags);
  return launch_status("sgd_nesterov_kernel");
}

extern "C" int ieee_sgd_nesterov_step(float* params, const float* grads, float* momentum_buf, int64_t n, float lr,
                                      float momentum, float weight_decay, int nesterov, void* stream) {
  return ieee_sgd_nesterov_step_ex(params, grads, momentum_buf, n, lr, momentum, weight_decay, nesterov, nullptr, nullptr, stream);
}

// ---- bf16 gradient exchange (IEEE_DP_GRAD_DTYPE=bf16: 219 MB over xGMI per step instead of 438) ---------------------
// pack: fp32 -> bf16, round to nearest even (the conversion the bf16 convs use); unpack: bf16 -> fp32 (exact).  Both are
// one pass over the slice: 6 B per element each way.  A slice may start anywhere in the flat buffer (the classifier biases
// have 171 elements): `peel` leading elements go one by one until both pointers are 32 / 16-byte aligned, then 8 per lane.
__global__ void grad_pack_bf16_kernel(const float* __restrict__ g, bf16* __restrict__ out, int64_t n, int64_t peel) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
  const int64_t n8 = (n - peel) >> 3;
  const float4* g4 = (const float4*)(g + peel);
  uint4* o4 = (uint4*)(out + peel);
  for (int64_t i = tid; i < n8; i += nth) {
    const float4 a = g4[2 * i], b = g4[2 * i + 1];
    const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    o4[i] = Vec16<bf16>::pack(f);
  }
  for (int64_t i = tid; i < peel; i += nth) out[i] = (bf16)g[i];
  for (int64_t i = peel + (n8 << 3) + tid; i < n; i += nth) out[i] = (bf16)g[i];
}
__global__ void grad_unpack_bf16_kernel(const bf16* __restrict__ in, float* __restrict__ g, int64_t n, int64_t peel) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
  const int64_t n8 = (n - peel) >> 3;
  const uint4* i4 = (const uint4*)(in + peel);
  float4* g4 = (float4*)(g + peel);
  for (int64_t i = tid; i < n8; i += nth) {
    float f[8];
    Vec16<bf16>::unpack(i4[i], f);
    g4[2 * i] = make_float4(f[0], f[1], f[2], f[3]);
    g4[2 * i + 1] = make_float4(f[4], f[5], f[6], f[7]);
  }
  for (int64_t i = tid; i < peel; i += nth) g[i] = (float)in[i];
  for (int64_t i = peel + (n8 << 3) + tid; i < n; i += nth) g[i] = (float)in[i];
}

// elements to peel so that (fp32 pointer, bf16 pointer) reach (32, 16)-byte alignment together; n when they never do
static int64_t grad_peel(const void* f32p, const void* bf16p, int64_t n) {
  const int64_t peel = (int64_t)(((32 - ((uintptr_t)f32p & 31)) & 31) >> 2);
  if ((((uintptr_t)f32p & 3) != 0) || ((((uintptr_t)bf16p + 2 * peel) & 15) != 0) || peel > n) return n;
  return peel;
}

extern "C" int ieee_grad_pack_bf16(const float* grads, void* out_bf16, int64_t n, void* stream) {
  if (n <= 0) return IEEE_OK;
  IEEE_REQUIRE(grads && out_bf16, "grad_pack_bf16: null pointer");
  const int64_t peel = grad_peel(grads, out_bf16, n);
  grad_pack_bf16_kernel<<<ewb((n + 7) / 8 + 8), 256, 0, (hipStream_t)stream>>>(grads, (bf16*)out_bf16, n, peel);
  return launch_status("grad_pack_bf16_kernel");
}
extern "C" int ieee_grad_unpack_bf16(const void* in_bf16, float* grads, int64_t n, void* stream) {
  if (n <= 0) return IEEE_OK;
  IEEE_REQUIRE(grads && in_bf16, "grad_unpack_bf16: null pointer");
  const int64_t peel = grad_peel(grads, in_bf16, n);
  grad_unpack_bf16_kernel<<<ewb((n + 7) / 8 + 8), 256, 0, (hipStream_t)stream>>>((const bf16*)in_bf16, grads, n, peel);
  return launch_status("grad_unpack_bf16_kernel");
}
