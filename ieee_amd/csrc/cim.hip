// Layout conversion, max-pool, and the Cross-modal Interacting Module tail:
//   reference torchreid/models/ieee3modalPart.py:266-282 (ChannelAttention), :427-435
//   (crossModalInteractionModule), :342-343 / :449-455 (the (1,1) and (6,1) adaptive average pools),
//   torchreid/models/resnet.py:499-501 (stem ReLU + MaxPool2d(3,2,1)).
// All maps are NHWC with a leading modality axis [3][B][H*W][C]; everything here is HBM-bound
// streaming with 16-byte lanes; per-(b,c) reductions over the 128 positions are done by `ty` row
// lanes of a block and finished through LDS.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "pool_gather.h"

namespace ieee {

// fp32 NCHW images (three separate tensors, as the reference's batch dict carries them) -> NHWC T with the channel
// axis padded to Cpad and a zero border of `pad` pixels on every side (pad = 3: the stem conv then needs no bounds tests)
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* x0, const float* x1, const float* x2, T* out, int B, int C, int H,
                                    int W, int Cpad, int pad) {
  const int z = blockIdx.y;
  const float* x = z == 0 ? x0 : (z == 1 ? x1 : x2);
  const int Hp = H + 2 * pad, Wp = W + 2 * pad;
  const int64_t total = (int64_t)B * Hp * Wp * Cpad;
  T* o = out + z * total;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cpad);
    int64_t p = i / Cpad;
    const int wp = (int)(p % Wp); p /= Wp;
    const int hp = (int)(p % Hp);
    const int b = (int)(p / Hp);
    const int h = hp - pad, w = wp - pad;
    float v = 0.f;
    if (c < C && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) v = x[(((int64_t)b * C + c) * H + h) * W + w];
    o[i] = from_f32<T>(v);
  }
}

// the executor's stem layout (bf16, Cpad = 4, even padded width): one thread writes TWO adjacent padded pixels as one
// 16-byte store and reads its (up to) 3 x 2 source values with 32-bit index arithmetic; a wave reads 512-byte runs of
// each colour plane and writes 1 KB runs (the generic kernel above moves one 2-byte element per thread)
__global__ __launch_bounds__(256) void nchw_to_nhwc4_bf16_kernel(const float* x0, const float* x1, const float* x2,
                                                                 bf16* out, int B, int C, int H, int W, int pad) {
  const int z = blockIdx.y;
  const float* x = z == 0 ? x0 : (z == 1 ? x1 : x2);
  const int Hp = H + 2 * pad, Wp = W + 2 * pad, Wh = Wp >> 1;
  const int total = B * Hp * Wh;
  uint4* o = (uint4*)(out + (int64_t)z * B * Hp * Wp * 4);
  const int plane = H * W;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int wq = i % Wh;
    const int t = i / Wh;
    const int hp = t % Hp, b = t / Hp;
    const int h = hp - pad, w = wq * 2 - pad;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if ((unsigned)h < (unsigned)H) {
      const float* src = x + ((int64_t)b * C * H + h) * W;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (c >= C) break;
        if ((unsigned)w < (unsigned)W) v[c] = src[c * plane + w];
        if ((unsigned)(w + 1) < (unsigned)W) v[4 + c] = src[c * plane + w + 1];
      }
    }
    o[i] = Vec16<bf16>::pack(v);
  }
}

// MaxPool2d(kernel 3, stride 2, pad 1) over NHWC; arg = window-local index (0..8) of the first maximum.
// AFFINE: x is the raw conv output y of the stem and the pooled tensor is max over relu(y * scale + shift) ROUNDED to T --
// exactly the values bn_apply_kernel would have stored, so pool / argmax are the bits of the two-kernel form while the
// full-resolution activation (written once, read 2.25 times) never exists (training forward of the stem: nothing else reads it)
template <typename T, bool AFFINE>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ out,
                                                          uint8_t* __restrict__ arg, int B, int Hi, int Wi, int C,
                                                          int Ho, int Wo, int64_t x_gs, int64_t o_gs, int lq, int lw,
                                                          int lh, const float* __restrict__ stats, int64_t stats_gs) {
  constexpr int VEC = 16 / sizeof(T);
  const int z = blockIdx.y;
  const int cprw = C / VEC;
  const float* sc = AFFINE ? stats + z * stats_gs + 2 * C : nullptr;
  const float* sh = AFFINE ? sc + C : nullptr;
  const int total = B * Ho * Wo * cprw;   // < 2^31 (checked by the launcher): 32-bit index arithmetic
  x += z * x_gs;
  out += z * o_gs;
  arg += z * o_gs;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int ch, q, pp, b;
    if (lq >= 0) {      // chunks per pixel, Wo and Ho are powers of two: shifts instead of three 32-bit divisions
      ch = i & (cprw - 1);
      const int p = i >> lq;
      q = p & (Wo - 1); pp = (p >> lw) & (Ho - 1); b = p >> (lw + lh);
    } else {
      ch = i % cprw;
      int p = i / cprw;
      q = p % Wo; p /= Wo;
      pp = p % Ho;
      b = p / Ho;
    }
    float best[VEC];
    int bi[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { best[e] = -INFINITY; bi[e] = 0; }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int h = pp * 2 - 1 + r;
      if ((unsigned)h >= (unsigned)Hi) continue;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int w = q * 2 - 1 + s;
        if ((unsigned)w >= (unsigned)Wi) continue;
        float v[VEC];
        Vec16<T>::unpack(*(const uint4*)(x + (((int64_t)b * Hi + h) * Wi + w) * C + ch * VEC), v);
        if constexpr (AFFINE) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) v[e] = fmaxf(v[e] * sc[ch * VEC + e] + sh[ch * VEC + e], 0.f);
          Vec16<T>::unpack(Vec16<T>::pack(v), v);     // the rounding of the stored activation
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          if (v[e] > best[e]) { best[e] = v[e]; bi[e] = r * 3 + s; }
      }
    }
    const int64_t o = (((int64_t)b * Ho + pp) * Wo + q) * C + ch * VEC;
    *(uint4*)(out + o) = Vec16<T>::pack(best);
    if constexpr (VEC == 8) {           // the 8 argmax bytes as one 8-byte store
      uint2 pk = make_uint2(0u, 0u);
#pragma unroll
      for (int e = 0; e < 4; ++e) { pk.x |= (unsigned)bi[e] << (8 * e); pk.y |= (unsigned)bi[4 + e] << (8 * e); }
      *(uint2*)(arg + o) = pk;
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) arg[o + e] = (uint8_t)bi[e];
    }
  }
}

// The AFFINE form with the three input rows of one pooled row staged in LDS: one workgroup per (image, pooled row); every
// element gets its scale / shift / ReLU / rounding once (each input row serves 1.5 pooled rows on average instead of being
// fetched by 2.25 windows through L1), and the 3 x 3 windows are scanned out of LDS in the order of maxpool_fwd_kernel.
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_maxpool_tiled_kernel(const T* __restrict__ y, T* __restrict__ out,
                                                                    uint8_t* __restrict__ arg, int Hi, int Wi, int C, int Ho,
                                                                    int Wo, int64_t x_gs, int64_t o_gs,
                                                                    const float* __restrict__ stats, int64_t stats_gs) {
  constexpr int VEC = 16 / sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char pool_lds[];
  T* sa = (T*)pool_lds;                  // [3][Wi][C]
  const int z = blockIdx.y, t = threadIdx.x;
  const int b = blockIdx.x / Ho, p = blockIdx.x - b * Ho;
  const int cprw = C / VEC, rowch = Wi * cprw;
  const float* sc = stats + z * stats_gs + 2 * C;
  const float* sh = sc + C;
  const T* yy = y + z * x_gs + (int64_t)b * Hi * Wi * C;
  for (int idx = t; idx < 3 * rowch; idx += 256) {
    const int r = idx / rowch, rem = idx - r * rowch;
    const int h = 2 * p - 1 + r;
    if ((unsigned)h >= (unsigned)Hi) continue;
    const int c0 = (rem % cprw) * VEC;
    float v[VEC];
    Vec16<T>::unpack(*(const uint4*)(yy + ((int64_t)h * Wi) * C + (int64_t)rem * VEC), v);
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = fmaxf(v[e] * sc[c0 + e] + sh[c0 + e], 0.f);
    *(uint4*)(sa + (int64_t)idx * VEC) = Vec16<T>::pack(v);
  }
  __syncthreads();
  for (int idx = t; idx < Wo * cprw; idx += 256) {
    const int q = idx / cprw, ch = idx - q * cprw;
    float best[VEC];
    int bi[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { best[e] = -INFINITY; bi[e] = 0; }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int h = 2 * p - 1 + r;
      if ((unsigned)h >= (unsigned)Hi) continue;
#pragma unroll
      for (int s2 = 0; s2 < 3; ++s2) {
        const int w = q * 2 - 1 + s2;
        if ((unsigned)w >= (unsigned)Wi) continue;
        float v[VEC];
        Vec16<T>::unpack(*(const uint4*)(sa + ((int64_t)(r * Wi + w) * cprw + ch) * VEC), v);
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          if (v[e] > best[e]) { best[e] = v[e]; bi[e] = r * 3 + s2; }
      }
    }
    const int64_t o = z * o_gs + (((int64_t)b * Ho + p) * Wo + q) * C + ch * VEC;
    *(uint4*)(out + o) = Vec16<T>::pack(best);
    if constexpr (VEC == 8) {
      uint2 pk = make_uint2(0u, 0u);
#pragma unroll
      for (int e = 0; e < 4; ++e) { pk.x |= (unsigned)bi[e] << (8 * e); pk.y |= (unsigned)bi[4 + e] << (8 * e); }
      *(uint2*)(arg + o) = pk;
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) arg[o + e] = (uint8_t)bi[e];
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dout, const uint8_t* __restrict__ arg,
                                                          T* __restrict__ dx, int B, int Hi, int Wi, int C, int Ho,
                                                          int Wo, int64_t x_gs, int64_t o_gs) {
  constexpr int VEC = 16 / sizeof(T);
  const int z = blockIdx.y;
  const int cprw = C / VEC;
  const int total = B * Hi * Wi * cprw;   // < 2^31 (checked by the launcher): 32-bit index arithmetic
  dout += z * o_gs;
  arg += z * o_gs;
  dx += z * x_gs;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int ch = i % cprw;
    int p = i / cprw;
    const int w = p % Wi; p /= Wi;
    const int h = p % Hi;
    const int b = p / Hi;
    float acc[VEC];
    pool_grad_gather<T>(dout, arg, b, h, w, ch, Ho, Wo, C, acc);
    *(uint4*)(dx + i * VEC) = Vec16<T>::pack(acc);
  }
}

// ------------------------------------------------------------------ per-(b,c) position reductions
struct PosGeom {
  int B, P, C;   // batch, positions per sample (H*W), channels
  int W;         // map width (positions are h*W + w)
  int cprw, tx, ty, cblocks;
};
static PosGeom pos_geom(int B, int H, int W, int C, int vec) {
  PosGeom g;
  g.B = B; g.P = H * W; g.C = C; g.W = W;
  g.cprw = C / vec;
  int tx = 1;   // widest channel span per block: 32 / 16 lanes per row measured 25 % / 2x slower (shorter contiguous runs)
  while (tx * 2 <= 64 && g.cprw % (tx * 2) == 0) tx *= 2;
  g.tx = tx;
  g.ty = 256 / tx;
  g.cblocks = g.cprw / tx;
  return g;
}

// adaptive-average-pool bin of row h for `parts` bins over H rows: start=floor(i*H/parts), end=ceil((i+1)*H/parts)
__device__ __forceinline__ int bin_start(int i, int H, int parts) { return (i * H) / parts; }
__device__ __forceinline__ int bin_end(int i, int H, int parts) { return ((i + 1) * H + parts - 1) / parts; }

// reduce NQ*VEC per-thread accumulators over the ty row lanes; result valid for ty == 0.  Eight quantities share
// one LDS round (two barriers), so the 32-64 accumulators of the CIM kernels cost 8-16 barriers, not 64-128.
constexpr int RED_Q = 8;
template <int NQV>
__device__ __forceinline__ void reduce_over_ty(float* acc, int tx, int ty, int txn, int tyn, float* red) {
#pragma unroll
  for (int q0 = 0; q0 < NQV; q0 += RED_Q) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RED_Q; ++q)
      if (q0 + q < NQV) red[q * 256 + ty * txn + tx] = acc[q0 + q];
    __syncthreads();
    if (ty == 0) {
#pragma unroll
      for (int q = 0; q < RED_Q; ++q) {
        if (q0 + q >= NQV) break;
        float s = 0.f;
        for (int y = 0; y < tyn; ++y) s += red[q * 256 + y * txn + tx];
        acc[q0 + q] = s;
      }
    }
  }
}

// Row blocks.  The three kernels below that deal with the `parts` overlapping row bins give each row lane `ty` of a block
// a run of consecutive map rows [r0, r1) instead of every ty-th position: such a run meets at most MAXS bins
// (first .. first + MAXS - 1; 2 for the 16-row / 6-part map), so a thread keeps MAXS x VEC bin values in registers instead
// of 8 x VEC, and the which-bins-hold-this-row decision is taken once per row, not once per position.  (With 64
// accumulators plus 64 pre-scaled pooled gradients per thread the first form of these kernels needed 150-300 VGPRs --
// one to three waves per SIMD, scratch spills in the forward -- and ran at 1.9 TB/s; the plain pools next to them, same
// geometry without the bins, at 4.4-5.3.)  MAXS is a template parameter (2 or 4) the launchers pick from the geometry.
// VEC consecutive fp32 / int32 values as 16-byte loads (the operands are 16-byte aligned: checked by the launchers).
// Element-wise loads behind the mode tests compiled to one 4-byte load and one wait per element.
template <int VEC, typename S> __device__ __forceinline__ void load_vec(const S* __restrict__ p, S (&d)[VEC]) {
  static_assert(sizeof(S) == 4 && VEC % 4 == 0, "16-byte pieces of 4-byte values");
#pragma unroll
  for (int q = 0; q < VEC / 4; ++q) {
    const uint4 v = *(const uint4*)(p + 4 * q);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) d[4 * q + e] = __builtin_bit_cast(S, w[e]);
  }
}

constexpr int MAXS_MAX = 4;
struct RowBlock { int r0, r1, first; };
__host__ __device__ inline int rb_bin_start(int i, int H, int parts) { return (i * H) / parts; }
__host__ __device__ inline int rb_bin_end(int i, int H, int parts) { return ((i + 1) * H + parts - 1) / parts; }
__host__ __device__ inline RowBlock row_block(int ty, int tyn, int H, int parts) {
  const int rpt = (H + tyn - 1) / tyn;
  RowBlock rb;
  rb.r0 = ty * rpt < H ? ty * rpt : H;
  rb.r1 = rb.r0 + rpt < H ? rb.r0 + rpt : H;
  rb.first = 0;
  while (rb.first < parts && rb_bin_end(rb.first, H, parts) <= rb.r0) ++rb.first;
  return rb;
}
// largest number of bins any row block meets (host side, for the launch check)
static int row_block_bins(int tyn, int H, int parts) {
  int worst = 0;
  for (int ty = 0; ty < tyn; ++ty) {
    const RowBlock rb = row_block(ty, tyn, H, parts);
    int last = rb.first - 1;
    for (int i = rb.first; i < parts; ++i)
      if (rb_bin_start(i, H, parts) < rb.r1 && rb_bin_end(i, H, parts) > rb.r0) last = i;
    if (rb.r1 > rb.r0 && last - rb.first + 1 > worst) worst = last - rb.first + 1;
  }
  return worst;
}

// K1: S_m = F_a + F_b (a,b = the other two modalities) and G_m = mean over positions of F_m
template <typename T>
__global__ __launch_bounds__(256) void gpool_sum_others_kernel(const T* __restrict__ F, T* __restrict__ S,
                                                               float* __restrict__ Gp, PosGeom g, int64_t gs,
                                                               int write_s) {
  constexpr int VEC = 16 / sizeof(T);
  __shared__ float red[RED_Q * 256];
  const int t = threadIdx.x, tx = t % g.tx, ty = t / g.tx;
  const int b = blockIdx.x / g.cblocks, cb = blockIdx.x % g.cblocks;
  const int c0 = (cb * g.tx + tx) * VEC;
  float acc[3 * VEC];
#pragma unroll
  for (int e = 0; e < 3 * VEC; ++e) acc[e] = 0.f;
#pragma unroll 4
  for (int p = ty; p < g.P; p += g.ty) {
    const int64_t off = ((int64_t)b * g.P + p) * g.C + c0;
    float v[3][VEC];
#pragma unroll
    for (int m = 0; m < 3; ++m) Vec16<T>::unpack(*(const uint4*)(F + m * gs + off), v[m]);
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[m * VEC + e] += v[m][e];
    if (write_s) {
      float s[VEC];
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const int a = (m + 1) % 3, c = (m + 2) % 3;
#pragma unroll
        for (int e = 0; e < VEC; ++e) s[e] = v[a][e] + v[c][e];
        *(uint4*)(S + m * gs + off) = Vec16<T>::pack(s);
      }
    }
  }
  reduce_over_ty<3 * VEC>(acc, tx, ty, g.tx, g.ty, red);
  if (ty == 0) {
    const float inv = 1.0f / g.P;
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int e = 0; e < VEC; ++e) Gp[((int64_t)m * g.B + b) * g.C + c0 + e] = acc[m * VEC + e] * inv;
  }
}

// K3: rest = relu(y2*scale+shift); avg / max / argmax over positions
template <typename T>
__global__ __launch_bounds__(256) void ca_pool_kernel(const T* __restrict__ y2, const float* __restrict__ stats,
                                                      float* __restrict__ avg, float* __restrict__ mx,
                                                      int* __restrict__ amax, PosGeom g, int64_t gs,
                                                      int64_t pool_gs) {
  constexpr int VEC = 16 / sizeof(T);
  __shared__ float red[RED_Q * 256];
  __shared__ int redi[256];
  const int z = blockIdx.y;
  const int t = threadIdx.x, tx = t % g.tx, ty = t / g.tx;
  const int b = blockIdx.x / g.cblocks, cb = blockIdx.x % g.cblocks;
  const int c0 = (cb * g.tx + tx) * VEC;
  const float* sc = stats + (int64_t)z * 4 * g.C + 2 * g.C;
  const float* sh = sc + g.C;
  float s[VEC], m[VEC];
  int mi[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { s[e] = 0.f; m[e] = -INFINITY; mi[e] = 0; }
#pragma unroll 4
  for (int p = ty; p < g.P; p += g.ty) {
    float v[VEC];
    Vec16<T>::unpack(*(const uint4*)(y2 + z * gs + ((int64_t)b * g.P + p) * g.C + c0), v);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float r = fmaxf(v[e] * sc[c0 + e] + sh[c0 + e], 0.f);
      s[e] += r;
      if (r > m[e]) { m[e] = r; mi[e] = p; }
    }
  }
  reduce_over_ty<VEC>(s, tx, ty, g.tx, g.ty, red);
  // max with first-occurrence index: combine over ty in increasing position order
  for (int e = 0; e < VEC; ++e) {
    __syncthreads();
    red[ty * g.tx + tx] = m[e];
    redi[ty * g.tx + tx] = mi[e];
    __syncthreads();
    if (ty == 0) {
      float bm = -INFINITY; int bi = 0x7fffffff;
      for (int y = 0; y < g.ty; ++y) {
        const float v = red[y * g.tx + tx]; const int i = redi[y * g.tx + tx];
        if (v > bm || (v == bm && i < bi)) { bm = v; bi = i; }
      }
      m[e] = bm; mi[e] = bi;
    }
  }
  if (ty == 0) {
    const int64_t o = ((int64_t)z * g.B + b) * g.C + c0;
    const int64_t op = z * pool_gs + (int64_t)b * g.C + c0;
#pragma unroll
    for (int e = 0; e < VEC; ++e) { avg[op + e] = s[e] / g.P; mx[op + e] = m[e]; amax[o + e] = mi[e]; }
  }
}

// K5: out = act1(y1) + act2(y2)*(1+att), pooled into `parts` overlapping row bins -> Pp [3][B][parts][C] fp32
//   mode 0: full CIM (act = relu(bn)), mode 1: attention off (att ignored), mode 2: interaction off (out = y1 raw)
template <typename T, int MAXS>
__global__ __launch_bounds__(256) void cim_tail_kernel(const T* __restrict__ y1, const T* __restrict__ y2,
                                                       const float* __restrict__ st1, const float* __restrict__ st2,
                                                       const float* __restrict__ att, float* __restrict__ Pp,
                                                       PosGeom g, int64_t gs, int H, int parts, int mode) {
  constexpr int VEC = 16 / sizeof(T);
  __shared__ float part[256 * MAXS * VEC];   // [ty][MAXS][tx * VEC]
  const int z = blockIdx.y;
  const int t = threadIdx.x, tx = t % g.tx, ty = t / g.tx;
  const int b = blockIdx.x / g.cblocks, cb = blockIdx.x % g.cblocks;
  const int c0 = (cb * g.tx + tx) * VEC;
  float sc1[VEC], sh1[VEC], sc2[VEC], sh2[VEC], at[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { sc1[e] = 1.f; sh1[e] = 0.f; sc2[e] = 0.f; sh2[e] = 0.f; at[e] = 0.f; }
  if (mode != 2) {
    load_vec<VEC>(st1 + (int64_t)z * 4 * g.C + 2 * g.C + c0, sc1);
    load_vec<VEC>(st1 + (int64_t)z * 4 * g.C + 3 * g.C + c0, sh1);
    load_vec<VEC>(st2 + (int64_t)z * 4 * g.C + 2 * g.C + c0, sc2);
    load_vec<VEC>(st2 + (int64_t)z * 4 * g.C + 3 * g.C + c0, sh2);
  }
  if (mode == 0) load_vec<VEC>(att + ((int64_t)z * g.B + b) * g.C + c0, at);
  const RowBlock rb = row_block(ty, g.ty, H, parts);
  float acc[MAXS * VEC];
#pragma unroll
  for (int e = 0; e < MAXS * VEC; ++e) acc[e] = 0.f;
  for (int h = rb.r0; h < rb.r1; ++h) {
    float row[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) row[e] = 0.f;
    const int64_t off0 = z * gs + ((int64_t)b * g.P + h * g.W) * g.C + c0;
    // the loads of WB positions are issued together (clamped past the row end), then consumed: left to the unroller the
    // two loads of a position were waited for before the next position's were issued
    constexpr int WB = 4;
    for (int w0 = 0; w0 < g.W; w0 += WB) {
      uint4 r1[WB], r2[WB];
#pragma unroll
      for (int k = 0; k < WB; ++k) {
        const int64_t o = off0 + (int64_t)min(w0 + k, g.W - 1) * g.C;
        r1[k] = *(const uint4*)(y1 + o);
        r2[k] = mode != 2 ? *(const uint4*)(y2 + o) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < WB; ++k) {
        if (w0 + k >= g.W) break;
        float v1[VEC], v2[VEC];
        Vec16<T>::unpack(r1[k], v1);
        Vec16<T>::unpack(r2[k], v2);
        if (mode != 2) {
#pragma unroll
          for (int e = 0; e < VEC; ++e)
            row[e] += fmaxf(v1[e] * sc1[e] + sh1[e], 0.f) + fmaxf(v2[e] * sc2[e] + sh2[e], 0.f) * (1.f + at[e]);
        } else {
#pragma unroll
          for (int e = 0; e < VEC; ++e) row[e] += v1[e];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < MAXS; ++k) {
      const int i = rb.first + k;
      if (i < parts && h >= rb_bin_start(i, H, parts) && h < rb_bin_end(i, H, parts)) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[k * VEC + e] += row[e];
      }
    }
  }
  // every row lane leaves its MAXS bin sums in LDS; bin i of a channel is then the sum, in lane order, of the slots that
  // stand for bin i (slots of bins a lane's rows do not meet hold 0)
  const int txw = g.tx * VEC, txw_log2 = __ffs(txw) - 1;   // a power of two
  __shared__ int firsts[256];
  __shared__ float invs[8];
  if (t < g.ty) firsts[t] = row_block(t, g.ty, H, parts).first;   // (integer divisions: once per lane / bin, not per output)
  if (t < parts) invs[t] = 1.0f / ((rb_bin_end(t, H, parts) - rb_bin_start(t, H, parts)) * g.W);
#pragma unroll
  for (int k = 0; k < MAXS; ++k)
#pragma unroll
    for (int e = 0; e < VEC; ++e) part[(ty * MAXS + k) * txw + tx * VEC + e] = acc[k * VEC + e];
  __syncthreads();
  for (int idx = t; idx < parts * txw; idx += 256) {
    const int i = idx >> txw_log2, c = idx & (txw - 1);
    float sum = 0.f;
    for (int y = 0; y < g.ty; ++y) {
      const int k = i - firsts[y];
      if (k >= 0 && k < MAXS) sum += part[(y * MAXS + k) * txw + c];
    }
    Pp[(((int64_t)z * g.B + b) * parts + i) * g.C + cb * txw + c] = sum * invs[i];
  }
}

// gradient of the pooled parts w.r.t. the (never materialised) CIM output at row h
__device__ __forceinline__ void dout_at(const float* dP, int64_t base, int C, int c0, int h, int H, int W, int parts,
                                        float* d, int VECN) {
  for (int e = 0; e < VECN; ++e) d[e] = 0.f;
  for (int i = 0; i < parts; ++i) {
    const int s = bin_start(i, H, parts), en = bin_end(i, H, parts);
    if (h >= s && h < en) {
      const float inv = 1.0f / ((en - s) * W);
      for (int e = 0; e < VECN; ++e) d[e] += dP[base + (int64_t)i * C + c0 + e] * inv;
    }
  }
}

// backward part 1: d_att[b,c] = sum_pos d_out * rest
template <typename T, int MAXS>
__global__ __launch_bounds__(256) void cim_bwd_datt_kernel(const float* __restrict__ dP, const T* __restrict__ y2,
                                                           const float* __restrict__ st2, float* __restrict__ datt,
                                                           PosGeom g, int64_t gs, int H, int parts) {
  constexpr int VEC = 16 / sizeof(T);
  __shared__ float red[RED_Q * 256];
  const int z = blockIdx.y;
  const int t = threadIdx.x, tx = t % g.tx, ty = t / g.tx;
  const int b = blockIdx.x / g.cblocks, cb = blockIdx.x % g.cblocks;
  const int c0 = (cb * g.tx + tx) * VEC;
  const float* sc = st2 + (int64_t)z * 4 * g.C + 2 * g.C;
  const float* sh = sc + g.C;
  const int64_t pbase = ((int64_t)z * g.B + b) * parts * g.C;
  // the pooled gradients of this thread's channels for the bins its rows meet, pre-scaled by their bin size, and the
  // BN coefficients: loaded once instead of per position
  const RowBlock rb = row_block(ty, g.ty, H, parts);
  float dps[MAXS][VEC], scv[VEC], shv[VEC];
#pragma unroll
  for (int k = 0; k < MAXS; ++k) {
    const int i = rb.first + k;
    const bool on = i < parts && rb.r1 > rb.r0;
    const float inv = on ? 1.0f / ((rb_bin_end(i, H, parts) - rb_bin_start(i, H, parts)) * g.W) : 0.f;
    load_vec<VEC>(dP + pbase + (int64_t)(on ? i : 0) * g.C + c0, dps[k]);
#pragma unroll
    for (int e = 0; e < VEC; ++e) dps[k][e] = on ? dps[k][e] * inv : 0.f;
  }
  load_vec<VEC>(sc + c0, scv);
  load_vec<VEC>(sh + c0, shv);
  float acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
  for (int h = rb.r0; h < rb.r1; ++h) {
    float d[VEC], row[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { d[e] = 0.f; row[e] = 0.f; }
#pragma unroll
    for (int k = 0; k < MAXS; ++k) {
      const int i = rb.first + k;
      if (i < parts && h >= rb_bin_start(i, H, parts) && h < rb_bin_end(i, H, parts)) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) d[e] += dps[k][e];
      }
    }
    constexpr int WB = 8;   // loads of WB positions in flight together (clamped past the row end)
    const int64_t off0 = z * gs + ((int64_t)b * g.P + h * g.W) * g.C + c0;
    for (int w0 = 0; w0 < g.W; w0 += WB) {
      uint4 r[WB];
#pragma unroll
      for (int k = 0; k < WB; ++k) r[k] = *(const uint4*)(y2 + off0 + (int64_t)min(w0 + k, g.W - 1) * g.C);
#pragma unroll
      for (int k = 0; k < WB; ++k) {
        if (w0 + k >= g.W) break;
        float v[VEC];
        Vec16<T>::unpack(r[k], v);
#pragma unroll
        for (int e = 0; e < VEC; ++e) row[e] += fmaxf(v[e] * scv[e] + shv[e], 0.f);
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] += d[e] * row[e];   // d is the same for every position of a row
  }
  reduce_over_ty<VEC>(acc, tx, ty, g.tx, g.ty, red);
  if (ty == 0) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) datt[((int64_t)z * g.B + b) * g.C + c0 + e] = acc[e];
  }
}

// backward part 2: g1 = d_out*[one>0]; g2 = (d_out*(1+att) + d_avg/P + [p==argmax]*d_max)*[rest>0]
//   mode 2 (interaction off): g1 = d_out (gradient straight to the trunk output), no g2.
// One block per (sample, channel block): the per-(b,c) operands (6 pooled gradients pre-scaled by their
// bin size, attention, BN scale/shift) are loaded once and the block then streams its 128 positions.
template <typename T, int MAXS>
__global__ __launch_bounds__(256) void cim_bwd_g_kernel(const float* __restrict__ dP, const T* __restrict__ y1,
                                                        const T* __restrict__ y2, const float* __restrict__ st1,
                                                        const float* __restrict__ st2, const float* __restrict__ att,
                                                        const float* __restrict__ davg, const float* __restrict__ dmax,
                                                        const int* __restrict__ amax, T* __restrict__ g1,
                                                        T* __restrict__ g2, PosGeom g, int64_t gs, int H, int parts,
                                                        int mode, int64_t pool_gs, float* __restrict__ bnp1,
                                                        float* __restrict__ bnp2) {
  constexpr int VEC = 16 / sizeof(T);
  __shared__ float red[RED_Q * 256];
  const int z = blockIdx.y;
  const int t = threadIdx.x, tx = t % g.tx, ty = t / g.tx;
  const int b = blockIdx.x / g.cblocks, cb = blockIdx.x % g.cblocks;
  const int c0 = (cb * g.tx + tx) * VEC;
  const RowBlock rb = row_block(ty, g.ty, H, parts);
  float dps[MAXS][VEC];
#pragma unroll
  for (int k = 0; k < MAXS; ++k) {
    const int i = rb.first + k;
    const bool on = i < parts && rb.r1 > rb.r0;
    const float inv = on ? 1.0f / ((rb_bin_end(i, H, parts) - rb_bin_start(i, H, parts)) * g.W) : 0.f;
    load_vec<VEC>(dP + (((int64_t)z * g.B + b) * parts + (on ? i : 0)) * g.C + c0, dps[k]);
#pragma unroll
    for (int e = 0; e < VEC; ++e) dps[k][e] = on ? dps[k][e] * inv : 0.f;
  }
  // BN-backward sums of convOne / convAvgRest (sum g, sum g*y over this sample's positions, of the ROUNDED g
  // that is stored), emitted per sample so that bn2d_bwd(stats_rblocks = B) needs no reduction pass
  // Two passes over this thread's rows, one per gradient map: each keeps only its own per-channel operands in
  // registers (2 x VEC for g1, 6 x VEC for g2) -- together with the row-block form 300 -> ~130 VGPRs
  auto run = [&](auto PASS) {
    constexpr int pass = decltype(PASS)::value;
    float sc[VEC], sh[VEC], a1[VEC], dav[VEC], dmx[VEC];
    int am[VEC];
    const float* stp = pass == 0 ? st1 : st2;
#pragma unroll
    for (int e = 0; e < VEC; ++e) { sc[e] = sh[e] = 0.f; a1[e] = 1.f; dav[e] = dmx[e] = 0.f; am[e] = -1; }
    if (mode != 2) {
      load_vec<VEC>(stp + (int64_t)z * 4 * g.C + 2 * g.C + c0, sc);
      load_vec<VEC>(stp + (int64_t)z * 4 * g.C + 3 * g.C + c0, sh);
    }
    if (mode == 0 && pass == 1) {
      const int64_t bc = ((int64_t)z * g.B + b) * g.C + c0;
      const int64_t pc = z * pool_gs + (int64_t)b * g.C + c0;
      load_vec<VEC>(att + bc, a1);
      load_vec<VEC>(davg + pc, dav);
      load_vec<VEC>(dmax + pc, dmx);
      load_vec<VEC>(amax + bc, am);
#pragma unroll
      for (int e = 0; e < VEC; ++e) { a1[e] += 1.f; dav[e] *= 1.0f / g.P; }
    }
    const T* yy = pass == 0 ? y1 : y2;
    T* gg = pass == 0 ? g1 : g2;
    float bsum[2 * VEC];
#pragma unroll
    for (int e = 0; e < 2 * VEC; ++e) bsum[e] = 0.f;
    for (int h = rb.r0; h < rb.r1; ++h) {
      float d[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) d[e] = 0.f;
#pragma unroll
      for (int k = 0; k < MAXS; ++k) {
        const int i = rb.first + k;
        if (i < parts && h >= rb_bin_start(i, H, parts) && h < rb_bin_end(i, H, parts)) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) d[e] += dps[k][e];
        }
      }
      if (mode == 2) {
        const uint4 q = Vec16<T>::pack(d);
        for (int w = 0; w < g.W; ++w) *(uint4*)(g1 + z * gs + ((int64_t)b * g.P + h * g.W + w) * g.C + c0) = q;
        continue;
      }
      constexpr int WB = 8;   // loads of WB positions in flight together (clamped past the row end)
      const int64_t off0 = z * gs + ((int64_t)b * g.P + h * g.W) * g.C + c0;
      for (int w0 = 0; w0 < g.W; w0 += WB) {
        uint4 r[WB];
#pragma unroll
        for (int k = 0; k < WB; ++k) r[k] = *(const uint4*)(yy + off0 + (int64_t)min(w0 + k, g.W - 1) * g.C);
#pragma unroll
        for (int k = 0; k < WB; ++k) {
          if (w0 + k >= g.W) break;
          const int p = h * g.W + w0 + k;
          float v[VEC], o[VEC];
          Vec16<T>::unpack(r[k], v);
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const bool on = v[e] * sc[e] + sh[e] > 0.f;
            float tv = d[e];
            if (pass == 1) {
              tv = d[e] * a1[e] + dav[e];
              if (am[e] == p) tv += dmx[e];
            }
            o[e] = on ? tv : 0.f;
          }
          const uint4 q = Vec16<T>::pack(o);
          *(uint4*)(gg + off0 + (int64_t)(w0 + k) * g.C) = q;
          Vec16<T>::unpack(q, o);          // (summed whether or not the sums are asked for: no branch inside the loop)
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            bsum[e] += o[e];
            bsum[VEC + e] += o[e] * v[e];
          }
        }
      }
    }
    if (bnp1 != nullptr && mode != 2) {   // partial layout [z][2][C][B] (sample index innermost, see StagedStoreEpi)
      reduce_over_ty<2 * VEC>(bsum, tx, ty, g.tx, g.ty, red);
      if (ty == 0) {
        float* bnp = pass == 0 ? bnp1 : bnp2;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const int64_t o = ((int64_t)z * 2 * g.C + c0 + e) * g.B + b;
          bnp[o] = bsum[e];
          bnp[o + (int64_t)g.C * g.B] = bsum[VEC + e];
        }
      }
    }
  };
  run(std::integral_constant<int, 0>());
  if (mode != 2) run(std::integral_constant<int, 1>());
}

// gradient w.r.t. the trunk output: dF_m = D1_m + DS_a + DS_b + dG_m/P  (a,b = other modalities)
//   (mode 2: dF_m = D1_m + dG_m/P)
template <typename T>
__global__ __launch_bounds__(256) void cim_bwd_combine_kernel(const T* __restrict__ D1, const T* __restrict__ DS,
                                                              const float* __restrict__ dG, T* __restrict__ dF,
                                                              PosGeom g, int64_t gs, int mode) {
  constexpr int VEC = 16 / sizeof(T);
  const int64_t total = (int64_t)g.B * g.P * g.cprw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % g.cprw);
    const int b = (int)((i / g.cprw) / g.P);
    float ds[3][VEC];
    if (mode != 2) {
#pragma unroll
      for (int m = 0; m < 3; ++m) Vec16<T>::unpack(*(const uint4*)(DS + m * gs + i * VEC), ds[m]);
    }
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      float v[VEC];
      Vec16<T>::unpack(*(const uint4*)(D1 + m * gs + i * VEC), v);
      const int a = (m + 1) % 3, c = (m + 2) % 3;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float x = v[e] + dG[((int64_t)m * g.B + b) * g.C + ch * VEC + e] * (1.0f / g.P);
        if (mode != 2) x += ds[a][e] + ds[c][e];
        v[e] = x;
      }
      *(uint4*)(dF + m * gs + i * VEC) = Vec16<T>::pack(v);
    }
  }
}

static int ew_blocks2(int64_t n) {
  static const int64_t cap = getenv("IEEE_CIM_EW_BLOCKS") ? atoll(getenv("IEEE_CIM_EW_BLOCKS")) : 2048;
  int64_t b = (n + 255) / 256;
  return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

}  // namespace ieee

using namespace ieee;

static int vecw(int dtype) { return dtype == IEEE_BF16 ? 8 : 4; }
#define DISPATCH_T(dtype, CALL_F32, CALL_BF16)                      \
  do {                                                              \
    if ((dtype) == IEEE_F32) { CALL_F32; }                          \
    else if ((dtype) == IEEE_BF16) { CALL_BF16; }                   \
    else { IEEE_REQUIRE(false, "bad dtype %d", (int)(dtype)); }     \
  } while (0)

extern "C" int ieee_nchw_to_nhwc3(const float* x_rgb, const float* x_ni, const float* x_ti, void* out, int dtype,
                                  int64_t B, int64_t C, int64_t H, int64_t W, int64_t Cpad, int64_t pad, void* stream) {
  IEEE_REQUIRE(x_rgb && x_ni && x_ti && out, "nchw_to_nhwc3: null pointer");
  IEEE_REQUIRE(Cpad >= C && pad >= 0, "nchw_to_nhwc3: Cpad < C or negative padding");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == IEEE_BF16 && Cpad == 4 && C <= 4 && ((W + 2 * pad) & 1) == 0 &&
      B * (H + 2 * pad) * (W + 2 * pad) < (1ll << 30)) {
    dim3 grid2(ew_blocks2(B * (H + 2 * pad) * ((W + 2 * pad) / 2)), 3);
    nchw_to_nhwc4_bf16_kernel<<<grid2, 256, 0, st>>>(x_rgb, x_ni, x_ti, (bf16*)out, (int)B, (int)C, (int)H, (int)W, (int)pad);
    return launch_status("nchw_to_nhwc4_bf16_kernel");
  }
  dim3 grid(ew_blocks2(B * Cpad * (H + 2 * pad) * (W + 2 * pad)), 3);
  DISPATCH_T(dtype, (nchw_to_nhwc_kernel<float><<<grid, 256, 0, st>>>(x_rgb, x_ni, x_ti, (float*)out, (int)B, (int)C, (int)H, (int)W, (int)Cpad, (int)pad)),
             (nchw_to_nhwc_kernel<bf16><<<grid, 256, 0, st>>>(x_rgb, x_ni, x_ti, (bf16*)out, (int)B, (int)C, (int)H, (int)W, (int)Cpad, (int)pad)));
  return launch_status("nchw_to_nhwc_kernel");
}

extern "C" int ieee_maxpool3x3s2_fwd(const void* x, void* out, uint8_t* argmax, int dtype, int64_t groups, int64_t B,
                                     int64_t Hi, int64_t Wi, int64_t C, void* stream) {
  IEEE_REQUIRE(x && out && argmax, "maxpool_fwd: null pointer");
  IEEE_REQUIRE(C % vecw(dtype) == 0, "maxpool_fwd: C not a multiple of the vector width");
  IEEE_REQUIRE(B * Hi * Wi * C < (1ll << 31), "maxpool_fwd: more than 2^31 elements per group");
  const int Ho = (int)((Hi + 2 - 3) / 2 + 1), Wo = (int)((Wi + 2 - 3) / 2 + 1);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(ew_blocks2(B * Ho * Wo * C / vecw(dtype)), (unsigned)groups);
  const int64_t xgs = B * Hi * Wi * C, ogs = B * Ho * Wo * C;
  auto lg = [](int64_t v) { int l = 0; while ((1ll << l) < v) ++l; return ((1ll << l) == v) ? l : -1; };
  int lq = lg(C / vecw(dtype)), lw = lg(Wo), lh = lg(Ho);
  if (lq < 0 || lw < 0 || lh < 0) lq = -1;
  DISPATCH_T(dtype, (maxpool_fwd_kernel<float, false><<<grid, 256, 0, st>>>((const float*)x, (float*)out, argmax, (int)B, (int)Hi, (int)Wi, (int)C, Ho, Wo, xgs, ogs, lq, lw, lh, nullptr, 0)),
             (maxpool_fwd_kernel<bf16, false><<<grid, 256, 0, st>>>((const bf16*)x, (bf16*)out, argmax, (int)B, (int)Hi, (int)Wi, (int)C, Ho, Wo, xgs, ogs, lq, lw, lh, nullptr, 0)));
  return launch_status("maxpool_fwd_kernel");
}

extern "C" int ieee_bn_relu_maxpool3x3s2_fwd(const void* y, const float* stats, void* out, uint8_t* argmax, int dtype,
                                             int64_t groups, int64_t B, int64_t Hi, int64_t Wi, int64_t C, void* stream) {
  IEEE_REQUIRE(y && stats && out && argmax, "bn_relu_maxpool_fwd: null pointer");
  IEEE_REQUIRE(C % vecw(dtype) == 0, "bn_relu_maxpool_fwd: C not a multiple of the vector width");
  IEEE_REQUIRE(B * Hi * Wi * C < (1ll << 31), "bn_relu_maxpool_fwd: more than 2^31 elements per group");
  const int Ho = (int)((Hi + 2 - 3) / 2 + 1), Wo = (int)((Wi + 2 - 3) / 2 + 1);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(ew_blocks2(B * Ho * Wo * C / vecw(dtype)), (unsigned)groups);
  const int64_t xgs = B * Hi * Wi * C, ogs = B * Ho * Wo * C;
  {
    static const bool tiled_on = !(getenv("IEEE_POOL_TILED") && atoi(getenv("IEEE_POOL_TILED")) == 0);
    const int64_t lds = 3 * Wi * C * (dtype == IEEE_BF16 ? 2 : 4);
    if (tiled_on && lds <= 64 * 1024 && Wi >= 32) {
      dim3 tgrid((unsigned)(B * Ho), (unsigned)groups);
      DISPATCH_T(dtype, (bn_relu_maxpool_tiled_kernel<float><<<tgrid, 256, lds, st>>>((const float*)y, (float*)out, argmax, (int)Hi, (int)Wi, (int)C, Ho, Wo, xgs, ogs, stats, 4 * C)),
                 (bn_relu_maxpool_tiled_kernel<bf16><<<tgrid, 256, lds, st>>>((const bf16*)y, (bf16*)out, argmax, (int)Hi, (int)Wi, (int)C, Ho, Wo, xgs, ogs, stats, 4 * C)));
      return launch_status("bn_relu_maxpool_tiled_kernel");
    }
  }
  auto lg = [](int64_t v) { int l = 0; while ((1ll << l) < v) ++l; return ((1ll << l) == v) ? l : -1; };
  int lq = lg(C / vecw(dtype)), lw = lg(Wo), lh = lg(Ho);
  if (lq < 0 || lw < 0 || lh < 0) lq = -1;
  DISPATCH_T(dtype, (maxpool_fwd_kernel<float, true><<<grid, 256, 0, st>>>((const float*)y, (float*)out, argmax, (int)B, (int)Hi, (int)Wi, (int)C, Ho, Wo, xgs, ogs, lq, lw, lh, stats, 4 * C)),
             (maxpool_fwd_kernel<bf16, true><<<grid, 256, 0, st>>>((const bf16*)y, (bf16*)out, argmax, (int)B, (int)Hi, (int)Wi, (int)C, Ho, Wo, xgs, ogs, lq, lw, lh, stats, 4 * C)));
  return launch_status("maxpool_fwd_kernel(affine)");
}

extern "C" int ieee_maxpool3x3s2_bwd(const void* dout, const uint8_t* argmax, void* dx, int dtype, int64_t groups,
                                     int64_t B, int64_t Hi, int64_t Wi, int64_t C, void* stream) {
  IEEE_REQUIRE(dout && dx && argmax, "maxpool_bwd: null pointer");
  IEEE_REQUIRE(B * Hi * Wi * C < (1ll << 31), "maxpool_bwd: more than 2^31 elements per group");
  const int Ho = (int)((Hi + 2 - 3) / 2 + 1), Wo = (int)((Wi + 2 - 3) / 2 + 1);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(ew_blocks2(B * Hi * Wi * C / vecw(dtype)), (unsigned)groups);
  const int64_t xgs = B * Hi * Wi * C, ogs = B * Ho * Wo * C;
  DISPATCH_T(dtype, (maxpool_bwd_kernel<float><<<grid, 256, 0, st>>>((const float*)dout, argmax, (float*)dx, (int)B, (int)Hi, (int)Wi, (int)C, Ho, Wo, xgs, ogs)),
             (maxpool_bwd_kernel<bf16><<<grid, 256, 0, st>>>((const bf16*)dout, argmax, (bf16*)dx, (int)B, (int)Hi, (int)Wi, (int)C, Ho, Wo, xgs, ogs)));
  return launch_status("maxpool_bwd_kernel");
}

extern "C" int ieee_gpool_sum_others(const void* F, void* S, float* Gp, int dtype, int64_t B, int64_t H, int64_t W,
                                     int64_t C, void* stream) {
  IEEE_REQUIRE(F && Gp, "gpool_sum_others: null pointer");
  IEEE_REQUIRE(C % (vecw(dtype)) == 0, "gpool_sum_others: bad C");
  const PosGeom g = pos_geom((int)B, (int)H, (int)W, (int)C, vecw(dtype));
  hipStream_t st = (hipStream_t)stream;
  const int64_t gs = B * H * W * C;
  dim3 grid((unsigned)(B * g.cblocks));
  DISPATCH_T(dtype, (gpool_sum_others_kernel<float><<<grid, 256, 0, st>>>((const float*)F, (float*)S, Gp, g, gs, S ? 1 : 0)),
             (gpool_sum_others_kernel<bf16><<<grid, 256, 0, st>>>((const bf16*)F, (bf16*)S, Gp, g, gs, S ? 1 : 0)));
  return launch_status("gpool_sum_others_kernel");
}

extern "C" int ieee_ca_pool(const void* y2, const float* stats2, float* avg, float* mx, int64_t pool_gs,
                            int32_t* argmax, int dtype, int64_t B, int64_t H, int64_t W, int64_t C, void* stream) {
  IEEE_REQUIRE(y2 && stats2 && avg && mx && argmax, "ca_pool: null pointer");
  const PosGeom g = pos_geom((int)B, (int)H, (int)W, (int)C, vecw(dtype));
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)(B * g.cblocks), 3);
  DISPATCH_T(dtype, (ca_pool_kernel<float><<<grid, 256, 0, st>>>((const float*)y2, stats2, avg, mx, argmax, g, B * H * W * C, pool_gs)),
             (ca_pool_kernel<bf16><<<grid, 256, 0, st>>>((const bf16*)y2, stats2, avg, mx, argmax, g, B * H * W * C, pool_gs)));
  return launch_status("ca_pool_kernel");
}

extern "C" int ieee_cim_tail_fwd(const void* y1, const void* y2, const float* stats1, const float* stats2,
                                 const float* att, float* parts_out, int dtype, int64_t B, int64_t H, int64_t W,
                                 int64_t C, int64_t parts, int mode, void* stream) {
  IEEE_REQUIRE(y1 && parts_out, "cim_tail_fwd: null pointer");
  IEEE_REQUIRE(parts >= 1 && parts <= 8, "cim_tail_fwd: parts must be in [1,8]");
  IEEE_REQUIRE(mode == 2 || (y2 && stats1 && stats2), "cim_tail_fwd: missing CIM operands");
  IEEE_REQUIRE(mode != 0 || att, "cim_tail_fwd: attention weights missing");
  IEEE_REQUIRE((((uintptr_t)stats1 | (uintptr_t)stats2 | (uintptr_t)att | (uintptr_t)parts_out) & 15) == 0, "cim_tail_fwd: the fp32 / int32 operands must be 16-byte aligned");
  const PosGeom g = pos_geom((int)B, (int)H, (int)W, (int)C, vecw(dtype));
  IEEE_REQUIRE(row_block_bins(g.ty, (int)H, (int)parts) <= MAXS_MAX, "cim_tail_fwd: a row block of this %ld-row / %ld-part map meets more than %d bins", (long)H, (long)parts, MAXS_MAX);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)(B * g.cblocks), 3);
  if (row_block_bins(g.ty, (int)H, (int)parts) <= 2) {
    DISPATCH_T(dtype, (cim_tail_kernel<float, 2><<<grid, 256, 0, st>>>((const float*)y1, (const float*)y2, stats1, stats2, att, parts_out, g, B * H * W * C, (int)H, (int)parts, mode)), (cim_tail_kernel<bf16, 2><<<grid, 256, 0, st>>>((const bf16*)y1, (const bf16*)y2, stats1, stats2, att, parts_out, g, B * H * W * C, (int)H, (int)parts, mode)));
  } else {
    DISPATCH_T(dtype, (cim_tail_kernel<float, 4><<<grid, 256, 0, st>>>((const float*)y1, (const float*)y2, stats1, stats2, att, parts_out, g, B * H * W * C, (int)H, (int)parts, mode)), (cim_tail_kernel<bf16, 4><<<grid, 256, 0, st>>>((const bf16*)y1, (const bf16*)y2, stats1, stats2, att, parts_out, g, B * H * W * C, (int)H, (int)parts, mode)));
  }
  return launch_status("cim_tail_kernel");
}

extern "C" int ieee_cim_tail_bwd_datt(const float* dparts, const void* y2, const float* stats2, float* datt, int dtype,
                                      int64_t B, int64_t H, int64_t W, int64_t C, int64_t parts, void* stream) {
  IEEE_REQUIRE(dparts && y2 && stats2 && datt, "cim_tail_bwd_datt: null pointer");
  IEEE_REQUIRE(parts >= 1 && parts <= 8, "cim_tail_bwd_datt: parts must be in [1,8]");
  IEEE_REQUIRE((((uintptr_t)dparts | (uintptr_t)stats2 | (uintptr_t)datt) & 15) == 0, "cim_tail_bwd_datt: the fp32 / int32 operands must be 16-byte aligned");
  const PosGeom g = pos_geom((int)B, (int)H, (int)W, (int)C, vecw(dtype));
  IEEE_REQUIRE(row_block_bins(g.ty, (int)H, (int)parts) <= MAXS_MAX, "cim_tail_bwd_datt: a row block of this %ld-row / %ld-part map meets more than %d bins", (long)H, (long)parts, MAXS_MAX);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)(B * g.cblocks), 3);
  if (row_block_bins(g.ty, (int)H, (int)parts) <= 2) {
    DISPATCH_T(dtype, (cim_bwd_datt_kernel<float, 2><<<grid, 256, 0, st>>>(dparts, (const float*)y2, stats2, datt, g, B * H * W * C, (int)H, (int)parts)), (cim_bwd_datt_kernel<bf16, 2><<<grid, 256, 0, st>>>(dparts, (const bf16*)y2, stats2, datt, g, B * H * W * C, (int)H, (int)parts)));
  } else {
    DISPATCH_T(dtype, (cim_bwd_datt_kernel<float, 4><<<grid, 256, 0, st>>>(dparts, (const float*)y2, stats2, datt, g, B * H * W * C, (int)H, (int)parts)), (cim_bwd_datt_kernel<bf16, 4><<<grid, 256, 0, st>>>(dparts, (const bf16*)y2, stats2, datt, g, B * H * W * C, (int)H, (int)parts)));
  }
  return launch_status("cim_bwd_datt_kernel");
}

extern "C" int ieee_cim_tail_bwd_g(const float* dparts, const void* y1, const void* y2, const float* stats1,
                                   const float* stats2, const float* att, const float* davg, const float* dmax,
                                   int64_t pool_gs, const int32_t* argmax, void* g1, void* g2, int dtype, int64_t B,
                                   int64_t H, int64_t W, int64_t C, int64_t parts, int mode, float* bn_partial1,
                                   float* bn_partial2, void* stream) {
  IEEE_REQUIRE(dparts && g1, "cim_tail_bwd_g: null pointer");
  IEEE_REQUIRE(parts >= 1 && parts <= 8, "cim_tail_bwd_g: parts must be in [1,8]");
  IEEE_REQUIRE((bn_partial1 == nullptr) == (bn_partial2 == nullptr), "cim_tail_bwd_g: give both BN partial buffers or none");
  IEEE_REQUIRE(mode == 2 || (y1 && y2 && stats1 && stats2 && g2), "cim_tail_bwd_g: missing CIM operands");
  IEEE_REQUIRE(mode != 0 || (att && davg && dmax && argmax), "cim_tail_bwd_g: missing attention operands");
  IEEE_REQUIRE((((uintptr_t)dparts | (uintptr_t)stats1 | (uintptr_t)stats2 | (uintptr_t)att | (uintptr_t)davg | (uintptr_t)dmax | (uintptr_t)argmax) & 15) == 0 && (pool_gs & 3) == 0, "cim_tail_bwd_g: the fp32 / int32 operands must be 16-byte aligned");
  const PosGeom g = pos_geom((int)B, (int)H, (int)W, (int)C, vecw(dtype));
  IEEE_REQUIRE(row_block_bins(g.ty, (int)H, (int)parts) <= MAXS_MAX, "cim_tail_bwd_g: a row block of this %ld-row / %ld-part map meets more than %d bins", (long)H, (long)parts, MAXS_MAX);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)(B * g.cblocks), 3);
  if (row_block_bins(g.ty, (int)H, (int)parts) <= 2) {
    DISPATCH_T(dtype, (cim_bwd_g_kernel<float, 2><<<grid, 256, 0, st>>>(dparts, (const float*)y1, (const float*)y2, stats1, stats2, att, davg, dmax, argmax, (float*)g1, (float*)g2, g, B * H * W * C, (int)H, (int)parts, mode, pool_gs, bn_partial1, bn_partial2)), (cim_bwd_g_kernel<bf16, 2><<<grid, 256, 0, st>>>(dparts, (const bf16*)y1, (const bf16*)y2, stats1, stats2, att, davg, dmax, argmax, (bf16*)g1, (bf16*)g2, g, B * H * W * C, (int)H, (int)parts, mode, pool_gs, bn_partial1, bn_partial2)));
  } else {
    DISPATCH_T(dtype, (cim_bwd_g_kernel<float, 4><<<grid, 256, 0, st>>>(dparts, (const float*)y1, (const float*)y2, stats1, stats2, att, davg, dmax, argmax, (float*)g1, (float*)g2, g, B * H * W * C, (int)H, (int)parts, mode, pool_gs, bn_partial1, bn_partial2)), (cim_bwd_g_kernel<bf16, 4><<<grid, 256, 0, st>>>(dparts, (const bf16*)y1, (const bf16*)y2, stats1, stats2, att, davg, dmax, argmax, (bf16*)g1, (bf16*)g2, g, B * H * W * C, (int)H, (int)parts, mode, pool_gs, bn_partial1, bn_partial2)));
  }
  return launch_status("cim_bwd_g_kernel");
}

extern "C" int ieee_cim_bwd_combine(const void* D1, const void* DS, const float* dG, void* dF, int dtype, int64_t B,
                                    int64_t H, int64_t W, int64_t C, int mode, void* stream) {
  IEEE_REQUIRE(D1 && dG && dF, "cim_bwd_combine: null pointer");
  IEEE_REQUIRE(mode == 2 || DS, "cim_bwd_combine: DS missing");
  const PosGeom g = pos_geom((int)B, (int)H, (int)W, (int)C, vecw(dtype));
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(ew_blocks2(B * H * W * C / vecw(dtype)));
  DISPATCH_T(dtype, (cim_bwd_combine_kernel<float><<<grid, 256, 0, st>>>((const float*)D1, (const float*)DS, dG, (float*)dF, g, B * H * W * C, mode)),
             (cim_bwd_combine_kernel<bf16><<<grid, 256, 0, st>>>((const bf16*)D1, (const bf16*)DS, dG, (bf16*)dF, g, B * H * W * C, mode)));
  return launch_status("cim_bwd_combine_kernel");
}
