// Layout conversion, max-pool, and the Cross-modal Interacting Module tail:
//   reference torchreid/models/ieee3modalPart.py:266-282 (ChannelAttention), :427-435
//   (crossModalInteractionModule), :342-343 / :449-455 (the (1,1) and (6,1) adaptive average pools),
//   torchreid/models/resnet.py:499-501 (stem ReLU + MaxPool2d(3,2,1)).
// All maps are NHWC with a leading modality axis [3][B][H*W][C]; everything here is HBM-bound
// streaming with 16-byte lanes; per-(b,c) reductions over the 128 positions are done by `ty` row
// lanes of a block and finished through LDS.
#include <stdlib.h>

#include "common.h"

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

// MaxPool2d(kernel 3, stride 2, pad 1) over NHWC; arg = window-local index (0..8) of the first maximum
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ out,
                                                          uint8_t* __restrict__ arg, int B, int Hi, int Wi, int C,
                                                          int Ho, int Wo, int64_t x_gs, int64_t o_gs) {
  constexpr int VEC = 16 / sizeof(T);
  const int z = blockIdx.y;
  const int cprw = C / VEC;
  const int total = B * Ho * Wo * cprw;   // < 2^31 (checked by the launcher): 32-bit index arithmetic
  x += z * x_gs;
  out += z * o_gs;
  arg += z * o_gs;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int ch = i % cprw;
    int p = i / cprw;
    const int q = p % Wo; p /= Wo;
    const int pp = p % Ho;
    const int b = p / Ho;
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
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    // the (at most 2 x 2) windows that contain this pixel: rows h/2 and (h+1)/2 (one row when h is even), same for
    // columns.  All their loads are issued before the first use (clamped addresses, validity applied afterwards): behind
    // per-window branches they were four dependent round trips per thread (177 us for the stem's 200 MB)
    const int pr[2] = {h >> 1, (h + 1) >> 1}, qc[2] = {w >> 1, (w + 1) >> 1};
    bool vr[2], vc[2];
    int lr[2], lc[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      vr[k] = (k == 0 || (h & 1)) && pr[k] < Ho;
      vc[k] = (k == 0 || (w & 1)) && qc[k] < Wo;
      lr[k] = h - (pr[k] * 2 - 1);
      lc[k] = w - (qc[k] * 2 - 1);
    }
    uint4 dv[2][2];
    uint2 av[2][2];
    uint8_t ab[2][2][VEC];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int l = 0; l < 2; ++l) {
        const int64_t o = (((int64_t)b * Ho + min(pr[k], Ho - 1)) * Wo + min(qc[l], Wo - 1)) * C + ch * VEC;
        dv[k][l] = *(const uint4*)(dout + o);
        if constexpr (VEC == 8) {
          av[k][l] = *(const uint2*)(arg + o);
        } else {
#pragma unroll
          for (int e = 0; e < VEC; ++e) ab[k][l][e] = arg[o + e];
        }
      }
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int l = 0; l < 2; ++l) {
        if (!(vr[k] && vc[l])) continue;
        float d[VEC];
        Vec16<T>::unpack(dv[k][l], d);
        const int local = lr[k] * 3 + lc[l];
        if constexpr (VEC == 8) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if ((int)((av[k][l].x >> (8 * e)) & 0xff) == local) acc[e] += d[e];
            if ((int)((av[k][l].y >> (8 * e)) & 0xff) == local) acc[4 + e] += d[4 + e];
          }
        } else {
#pragma unroll
          for (int e = 0; e < VEC; ++e)
            if (ab[k][l][e] == local) acc[e] += d[e];
        }
      }
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
template <typename T>
__global__ __launch_bounds__(256) void cim_tail_kernel(const T* __restrict__ y1, const T* __restrict__ y2,
                                                       const float* __restrict__ st1, const float* __restrict__ st2,
                                                       const float* __restrict__ att, float* __restrict__ Pp,
                                                       PosGeom g, int64_t gs, int H, int parts, int mode) {
  constexpr int VEC = 16 / sizeof(T);
  constexpr int MAXP = 8;
  __shared__ float red[RED_Q * 256];
  const int z = blockIdx.y;
  const int t = threadIdx.x, tx = t % g.tx, ty = t / g.tx;
  const int b = blockIdx.x / g.cblocks, cb = blockIdx.x % g.cblocks;
  const int c0 = (cb * g.tx + tx) * VEC;
  float sc1[VEC], sh1[VEC], sc2[VEC], sh2[VEC], at[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    sc1[e] = 1.f; sh1[e] = 0.f; sc2[e] = 0.f; sh2[e] = 0.f; at[e] = 0.f;
    if (mode != 2) {
      sc1[e] = st1[(int64_t)z * 4 * g.C + 2 * g.C + c0 + e];
      sh1[e] = st1[(int64_t)z * 4 * g.C + 3 * g.C + c0 + e];
      sc2[e] = st2[(int64_t)z * 4 * g.C + 2 * g.C + c0 + e];
      sh2[e] = st2[(int64_t)z * 4 * g.C + 3 * g.C + c0 + e];
    }
    if (mode == 0) at[e] = att[((int64_t)z * g.B + b) * g.C + c0 + e];
  }
  float acc[MAXP * VEC];
#pragma unroll
  for (int e = 0; e < MAXP * VEC; ++e) acc[e] = 0.f;
  int bs[MAXP], be[MAXP];   // row bins, computed once (empty beyond `parts`)
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    bs[i] = i < parts ? bin_start(i, H, parts) : 0;
    be[i] = i < parts ? bin_end(i, H, parts) : 0;
  }
#pragma unroll 4
  for (int p = ty; p < g.P; p += g.ty) {
    const int h = p / g.W;
    const int64_t off = z * gs + ((int64_t)b * g.P + p) * g.C + c0;
    float v1[VEC], v2[VEC], o[VEC];
    Vec16<T>::unpack(*(const uint4*)(y1 + off), v1);
    if (mode != 2) {
      Vec16<T>::unpack(*(const uint4*)(y2 + off), v2);
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        o[e] = fmaxf(v1[e] * sc1[e] + sh1[e], 0.f) + fmaxf(v2[e] * sc2[e] + sh2[e], 0.f) * (1.f + at[e]);
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) o[e] = v1[e];
    }
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      if (h >= bs[i] && h < be[i]) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[i * VEC + e] += o[e];
      }
    }
  }
  reduce_over_ty<MAXP * VEC>(acc, tx, ty, g.tx, g.ty, red);
  if (ty == 0) {
    for (int i = 0; i < parts; ++i) {
      const float inv = 1.0f / ((bin_end(i, H, parts) - bin_start(i, H, parts)) * g.W);
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        Pp[(((int64_t)z * g.B + b) * parts + i) * g.C + c0 + e] = acc[i * VEC + e] * inv;
    }
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
template <typename T>
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
  // the pooled gradients of this thread's channels, pre-scaled by their bin size, and the BN coefficients: loaded
  // once (as in cim_bwd_g_kernel) instead of per position
  constexpr int MAXP = 8;
  float dps[MAXP][VEC], scv[VEC], shv[VEC];
  int bs[MAXP], be[MAXP];
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    bs[i] = i < parts ? bin_start(i, H, parts) : 0;
    be[i] = i < parts ? bin_end(i, H, parts) : 0;
    const float inv = i < parts ? 1.0f / ((be[i] - bs[i]) * g.W) : 0.f;
#pragma unroll
    for (int e = 0; e < VEC; ++e) dps[i][e] = i < parts ? dP[pbase + (int64_t)i * g.C + c0 + e] * inv : 0.f;
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) { scv[e] = sc[c0 + e]; shv[e] = sh[c0 + e]; }
  float acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
#pragma unroll 4
  for (int p = ty; p < g.P; p += g.ty) {
    const int h = p / g.W;
    float d[VEC], v[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) d[e] = 0.f;
#pragma unroll
    for (int i = 0; i < MAXP; ++i)
      if (h >= bs[i] && h < be[i]) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) d[e] += dps[i][e];
      }
    Vec16<T>::unpack(*(const uint4*)(y2 + z * gs + ((int64_t)b * g.P + p) * g.C + c0), v);
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] += d[e] * fmaxf(v[e] * scv[e] + shv[e], 0.f);
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
template <typename T>
__global__ __launch_bounds__(256) void cim_bwd_g_kernel(const float* __restrict__ dP, const T* __restrict__ y1,
                                                        const T* __restrict__ y2, const float* __restrict__ st1,
                                                        const float* __restrict__ st2, const float* __restrict__ att,
                                                        const float* __restrict__ davg, const float* __restrict__ dmax,
                                                        const int* __restrict__ amax, T* __restrict__ g1,
                                                        T* __restrict__ g2, PosGeom g, int64_t gs, int H, int parts,
                                                        int mode, int64_t pool_gs, float* __restrict__ bnp1,
                                                        float* __restrict__ bnp2) {
  constexpr int VEC = 16 / sizeof(T);
  constexpr int MAXP = 8;
  __shared__ float red[RED_Q * 256];
  const int z = blockIdx.y;
  const int t = threadIdx.x, tx = t % g.tx, ty = t / g.tx;
  const int b = blockIdx.x / g.cblocks, cb = blockIdx.x % g.cblocks;
  const int c0 = (cb * g.tx + tx) * VEC;
  float dps[MAXP][VEC];
  int bs[MAXP], be[MAXP];
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    bs[i] = i < parts ? bin_start(i, H, parts) : 0;
    be[i] = i < parts ? bin_end(i, H, parts) : 0;
    const float inv = i < parts ? 1.0f / ((be[i] - bs[i]) * g.W) : 0.f;
#pragma unroll
    for (int e = 0; e < VEC; ++e)
      dps[i][e] = i < parts ? dP[(((int64_t)z * g.B + b) * parts + i) * g.C + c0 + e] * inv : 0.f;
  }
  float sc1[VEC], sh1[VEC], sc2[VEC], sh2[VEC], a1[VEC], dav[VEC], dmx[VEC];
  int am[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    sc1[e] = sh1[e] = sc2[e] = sh2[e] = 0.f; a1[e] = 1.f; dav[e] = dmx[e] = 0.f; am[e] = -1;
    if (mode != 2) {
      sc1[e] = st1[(int64_t)z * 4 * g.C + 2 * g.C + c0 + e];
      sh1[e] = st1[(int64_t)z * 4 * g.C + 3 * g.C + c0 + e];
      sc2[e] = st2[(int64_t)z * 4 * g.C + 2 * g.C + c0 + e];
      sh2[e] = st2[(int64_t)z * 4 * g.C + 3 * g.C + c0 + e];
    }
    if (mode == 0) {
      const int64_t bc = ((int64_t)z * g.B + b) * g.C + c0 + e;
      const int64_t pc = z * pool_gs + (int64_t)b * g.C + c0 + e;
      a1[e] = 1.f + att[bc];
      dav[e] = davg[pc] * (1.0f / g.P);
      dmx[e] = dmax[pc];
      am[e] = amax[bc];
    }
  }
  // BN-backward sums of convOne / convAvgRest (sum g, sum g*y over this sample's positions, of the ROUNDED g
  // that is stored), emitted per sample so that bn2d_bwd(stats_rblocks = B) needs no reduction pass
  float bsum[4 * VEC];
#pragma unroll
  for (int e = 0; e < 4 * VEC; ++e) bsum[e] = 0.f;
#pragma unroll 4
  for (int p = ty; p < g.P; p += g.ty) {
    const int h = p / g.W;
    float d[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) d[e] = 0.f;
#pragma unroll
    for (int i = 0; i < MAXP; ++i)
      if (h >= bs[i] && h < be[i]) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) d[e] += dps[i][e];
      }
    const int64_t off = z * gs + ((int64_t)b * g.P + p) * g.C + c0;
    if (mode == 2) {
      *(uint4*)(g1 + off) = Vec16<T>::pack(d);
      continue;
    }
    float v1[VEC], v2[VEC], o1[VEC], o2[VEC];
    Vec16<T>::unpack(*(const uint4*)(y1 + off), v1);
    Vec16<T>::unpack(*(const uint4*)(y2 + off), v2);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const bool on1 = v1[e] * sc1[e] + sh1[e] > 0.f;
      const bool on2 = v2[e] * sc2[e] + sh2[e] > 0.f;
      float t2 = d[e] * a1[e] + dav[e];
      if (am[e] == p) t2 += dmx[e];
      o1[e] = on1 ? d[e] : 0.f;
      o2[e] = on2 ? t2 : 0.f;
    }
    const uint4 q1 = Vec16<T>::pack(o1), q2 = Vec16<T>::pack(o2);
    *(uint4*)(g1 + off) = q1;
    *(uint4*)(g2 + off) = q2;
    if (bnp1 != nullptr) {
      Vec16<T>::unpack(q1, o1);
      Vec16<T>::unpack(q2, o2);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        bsum[e] += o1[e]; bsum[VEC + e] += o1[e] * v1[e];
        bsum[2 * VEC + e] += o2[e]; bsum[3 * VEC + e] += o2[e] * v2[e];
      }
    }
  }
  if (bnp1 != nullptr && mode != 2) {   // partial layout [z][2][C][B] (sample index innermost, see StagedStoreEpi)
    reduce_over_ty<4 * VEC>(bsum, tx, ty, g.tx, g.ty, red);
    if (ty == 0) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const int64_t o = ((int64_t)z * 2 * g.C + c0 + e) * g.B + b;
        bnp1[o] = bsum[e];
        bnp1[o + (int64_t)g.C * g.B] = bsum[VEC + e];
        bnp2[o] = bsum[2 * VEC + e];
        bnp2[o + (int64_t)g.C * g.B] = bsum[3 * VEC + e];
      }
    }
  }
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
  DISPATCH_T(dtype, (maxpool_fwd_kernel<float><<<grid, 256, 0, st>>>((const float*)x, (float*)out, argmax, (int)B, (int)Hi, (int)Wi, (int)C, Ho, Wo, xgs, ogs)),
             (maxpool_fwd_kernel<bf16><<<grid, 256, 0, st>>>((const bf16*)x, (bf16*)out, argmax, (int)B, (int)Hi, (int)Wi, (int)C, Ho, Wo, xgs, ogs)));
  return launch_status("maxpool_fwd_kernel");
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
  const PosGeom g = pos_geom((int)B, (int)H, (int)W, (int)C, vecw(dtype));
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)(B * g.cblocks), 3);
  DISPATCH_T(dtype, (cim_tail_kernel<float><<<grid, 256, 0, st>>>((const float*)y1, (const float*)y2, stats1, stats2, att, parts_out, g, B * H * W * C, (int)H, (int)parts, mode)),
             (cim_tail_kernel<bf16><<<grid, 256, 0, st>>>((const bf16*)y1, (const bf16*)y2, stats1, stats2, att, parts_out, g, B * H * W * C, (int)H, (int)parts, mode)));
  return launch_status("cim_tail_kernel");
}

extern "C" int ieee_cim_tail_bwd_datt(const float* dparts, const void* y2, const float* stats2, float* datt, int dtype,
                                      int64_t B, int64_t H, int64_t W, int64_t C, int64_t parts, void* stream) {
  IEEE_REQUIRE(dparts && y2 && stats2 && datt, "cim_tail_bwd_datt: null pointer");
  IEEE_REQUIRE(parts >= 1 && parts <= 8, "cim_tail_bwd_datt: parts must be in [1,8]");
  const PosGeom g = pos_geom((int)B, (int)H, (int)W, (int)C, vecw(dtype));
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)(B * g.cblocks), 3);
  DISPATCH_T(dtype, (cim_bwd_datt_kernel<float><<<grid, 256, 0, st>>>(dparts, (const float*)y2, stats2, datt, g, B * H * W * C, (int)H, (int)parts)),
             (cim_bwd_datt_kernel<bf16><<<grid, 256, 0, st>>>(dparts, (const bf16*)y2, stats2, datt, g, B * H * W * C, (int)H, (int)parts)));
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
  const PosGeom g = pos_geom((int)B, (int)H, (int)W, (int)C, vecw(dtype));
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)(B * g.cblocks), 3);
  DISPATCH_T(dtype, (cim_bwd_g_kernel<float><<<grid, 256, 0, st>>>(dparts, (const float*)y1, (const float*)y2, stats1, stats2, att, davg, dmax, argmax, (float*)g1, (float*)g2, g, B * H * W * C, (int)H, (int)parts, mode, pool_gs, bn_partial1, bn_partial2)),
             (cim_bwd_g_kernel<bf16><<<grid, 256, 0, st>>>(dparts, (const bf16*)y1, (const bf16*)y2, stats1, stats2, att, davg, dmax, argmax, (bf16*)g1, (bf16*)g2, g, B * H * W * C, (int)H, (int)parts, mode, pool_gs, bn_partial1, bn_partial2)));
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
