// Convolution as implicit GEMM over NHWC activations.
//   forward : Y[pix][co]  = sum_{r,s,ci} X[gather(pix,r,s)][ci] * Wf[co][(r,s,ci)]
//   dgrad   : dX[pix][ci] = sum_{r,s,co} dY[gather'(pix,r,s)][co] * Wd[ci][(r,s,co)]   (+ addend)
//   wgrad   : dW[co][(r,s,ci)] = sum_pix dY[pix][co] * X[gather(pix,r,s)][ci]   (split-K slabs)
// The three ResNet-50 streams (RGB / NI / TI) are independent and run as one
// launch: blockIdx.y = modality, pointers advance by per-modality strides.
// Replaces torch's conv2d / conv2d backward as dispatched from the reference's
// torchreid/models/resnet.py:164-184,622-631 and ieee3modalPart.py:427-435.
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include <hip/hip_ext.h>

#include "gemm_core.h"

namespace ieee {

// Measurement hook (ieee_conv_profile_events): the NEXT forward / dgrad launch of this thread carries the two events as
// its own start / stop signals (hipExtLaunchKernelGGL) -- the kernel's duration by the GPU's timestamps, with no event
// record (= no barrier packet) added to the queue, so the launch can be timed inside the undisturbed two-stream step.
thread_local hipEvent_t tl_time_start = nullptr, tl_time_stop = nullptr;

// BatchNorm sums as order-independent TOTALS (round 4): instead of one float per (channel, row tile) that a finalize launch
// has to add up, a tile adds its per-channel sums -- converted to 64-bit fixed point -- to ONE total per (quantity, channel)
// with no-return global atomics.  Integer adds commute, so the total does not depend on the arrival order (bit-reproducible),
// nothing waits for anything inside the kernel (no ticket, no store drain), and the consumer of the statistics (the BatchNorm
// apply / backward-apply launch behind the conv) derives its coefficients from 2 x C numbers in its prologue: the finalize
// launch -- 6 us forward, 12-14 us backward, 110 per step on the launch stream -- disappears for the units that use this.
// Contention: M / 128 adders per address; the executor only uses it up to IEEE_BN_TOTALS_TILES row tiles per modality.
// Fixed point: 2^24 for the forward sums (sum y, sum y^2), 2^40 for the backward ones (sum g, sum g*y: gradients are small);
// a tile's float sum carries 24 bits itself.  ieee_conv_next_bn_totals() arms the NEXT forward / dgrad launch of the thread.
thread_local long long* tl_totals = nullptr;
thread_local int64_t tl_totals_gs = 0;
thread_local int tl_totals_rep = 1;   // replicas of the totals (a power of two): tile t adds to replica t % rep -- fewer adders per address
// Range guard (round 5).  An int64 total wraps silently; the reference's fp32 batch_norm (torchreid/models/resnet.py:164-184)
// would return finite numbers or inf there.  So: a tile may contribute at most 2^62 / (row tiles of the launch) units -- the sum
// of ALL tiles, replicas included, then cannot leave +-2^62 and never wraps -- and a tile sum beyond that share (or NaN) is
// clamped AND reported: overflow[0] (forward sums) / overflow[1] (backward sums) = 1, plain stores to a word the caller owns
// (the executor: host-visible memory it looks at when it reads the step's summary).  The BatchNorm passes report a total
// beyond half the range (2^61) in overflow[2] / overflow[3] (bn.hip).  Nothing is paid unless the compare fails.
thread_local int* tl_totals_flag = nullptr;
constexpr float TOT_RANGE = 4611686018427387904.0f;   // 2^62
// whatever a conv entry point returns, nothing stays armed for a later launch of the thread (a call that fails its argument
// checks must not leave its totals / timing events to the next one)
struct OneShotArms {
  ~OneShotArms() { tl_totals = nullptr; tl_totals_flag = nullptr; tl_totals_rep = 1; tl_time_start = tl_time_stop = nullptr; }
};
constexpr float TOT_SCALE_FWD = 16777216.0f;          // 2^24
constexpr float TOT_SCALE_BWD = 1099511627776.0f;     // 2^40
__device__ __forceinline__ long long to_fixed(float v, float scale, float lim, int* flag) {
  float x = v * scale;
  if (!(fabsf(x) <= lim)) {                      // beyond this tile's share of the range, or NaN
    if (flag != nullptr) *(volatile int*)flag = 1;
    x = fminf(fmaxf(x, -lim), lim);              // (NaN -> -lim; the NaN itself reaches the output through y)
  }
  return __float2ll_rn(x);
}

template <class K, class... A>
static inline void launch_timed(K kernel, dim3 grid, size_t smem, hipStream_t st, A... args) {
  if (tl_time_stop != nullptr) {
    hipExtLaunchKernelGGL(kernel, grid, dim3(256), (uint32_t)smem, st, tl_time_start, tl_time_stop, 0, args...);
    tl_time_start = tl_time_stop = nullptr;
  } else {
    kernel<<<grid, 256, smem, st>>>(args...);
  }
}


__device__ __forceinline__ void tile_map_xy(int tiles_m, int tiles_n, int group, int& tm, int& tn) {
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

// C[m][n..n+3] store with optional residual addend (same layout/dtype as the output).
template <typename T> struct StoreEpi {
  static constexpr bool kStaged = false;
  T* out;
  const T* addend;
  int64_t ld;
  int M, N;
  __device__ __forceinline__ void operator()(int m, int n, f32x4 v) const {
    if (m >= M || n >= N) return;
    T* o = out + (int64_t)m * ld + n;
    if (n + 3 < N) {
      if (addend != nullptr) {
        const T* a = addend + (int64_t)m * ld + n;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += to_f32(a[r]);
      }
      if constexpr (sizeof(T) == 4) {
        *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        *(uint2*)o = make_uint2(Vec16<bf16>::pk(v[0], v[1]), Vec16<bf16>::pk(v[2], v[3]));
      }
    } else {
      for (int r = 0; r < 4 && n + r < N; ++r) {
        float x = v[r];
        if (addend != nullptr) x += to_f32(addend[(int64_t)m * ld + n + r]);
        o[r] = from_f32<T>(x);
      }
    }
  }
};

// Train-mode BatchNorm finalize inside the conv that produced the sums (round 3): the workgroup that arrives LAST at a
// column block's ticket reduces that block's per-tile partial sums (fixed tile order: deterministic) and writes the
// unit's mean / invstd / scale / shift and running statistics, so no finalize launch sits between the conv and the
// BatchNorm apply.  Protocol (platform guide, split-K reduction recipe): every workgroup stores its partials write-through
// (sc1), every wave drains its stores (s_waitcnt vmcnt(0)), workgroup barrier, lane 0 takes a ticket (relaxed, agent scope);
// the last arriver's lane 0 runs ONE agent-scope acquire, waits for it, workgroup barrier, plain loads.  Only for launches
// of at most FIN_MAX_TILES row tiles per modality (one workgroup reads 2 x 128 x tiles partials: 64 KB at 64 tiles).
constexpr int FIN_MAX_TILES = 64;
struct BnFin {
  const float* gamma = nullptr;
  const float* beta = nullptr;
  float* rm = nullptr;         // running mean / var (may be null)
  float* rv = nullptr;
  float* stats = nullptr;      // [groups][4][C]
  int* counter = nullptr;      // [groups][tiles_n] tickets, zero before the launch; the last arriver resets its own
  int64_t param_gs = 0, buf_gs = 0;
  float momentum = 0.f, eps = 0.f;
  int M = 0, on = 0;
};

// Tile epilogue through LDS: the accumulators (4 consecutive channels of one pixel per lane) are written
// to an LDS image of the C tile, then every thread stores full 16-byte, row-contiguous chunks (a wave
// instruction covers 4 rows x 256 B) instead of 8-byte pieces scattered over 16 rows.  Optionally adds a
// residual tensor and, for a conv that feeds a train-mode BatchNorm, emits the tile's per-channel
// sum / sum-of-squares (of the ROUNDED stored values) so that no separate statistics pass is needed:
// bn_partial[group][2][N][tiles_m] (row tiles innermost: the finalize kernel reads one channel's partials as
// one contiguous run instead of a stride-2N walk, which took 15-70 us on the 1024-4096 row tiles of layer1/stem).
// VAR 1: GEMM rows in parity-class order (stride-2 dgrad).  VAR 2: the addend is compact -- it holds only the pixels with even
// row and column of this tile's map, (H/2) x (W/2) per image (the gradient a stride-2 1x1 branch sends back: every other
// pixel gets none) -- so the branch's dgrad neither writes nor this epilogue reads the three quarters that are zero.
// VAR 3 (MODE 2): the sums of a second BatchNorm fed by the same gradient are emitted too (by2 / bn_partial2).
template <typename T, int MODE, int VAR = 0> struct StagedStoreEpi {
  static constexpr bool PERM = VAR == 1;
  static constexpr bool kStaged = true;
  static constexpr bool STATS = MODE == 1 || MODE == 2;
  // MODE 3 (inference): the consumer BatchNorm runs on running statistics, so its scale / shift are known before
  // the conv and the whole bn(+residual)(+relu) is applied here: out = [relu](v*scale + shift [+ addend]);
  // bstats = that BN's [4][N] stats (scale at 2N, shift at 3N), relu flag in `relu`.
  int relu = 0;
  T* out;
  const T* addend;
  float* bn_partial;   // this group's [2][N][tiles_m] block (MODE 1: sum v, sum v^2; MODE 2: sum g, sum g*y)
  int64_t ld;
  int M, N, tile_m, tiles_m;
  // MODE 2 (dgrad feeding the BatchNorm backward of the PREVIOUS unit): g = v * [relu mask]; the mask comes from
  // the stored activation (bmask) or is recomputed from y and that unit's scale/shift (bstats = [4][N])
  const T* by;
  const T* bmask;
  const float* bstats;
  const uint8_t* bbits = nullptr;   // the mask as packed bits (one byte per 8 channels, ieee_bn2d_fwd relu_bits) instead of bmask
  // GEMM rows in parity-class-major order (stride-2 dgrad, LoaderIm2colNT<.., true>): row m0 + r of the tile is pixel
  // pbase + (2*(r / pwc))*pW + 2*(r % pwc) of the output map (pwc = 0: rows are pixels)
  // (pwc = Wo/2 is a power of two on this path: pwl = log2; a tensor of one modality has < 2^31 elements)
  int pwc = 0, pwl = 0, pW = 0, pimg = 0, phc = 0, pnimg = 0;
  int as_wl = 0, as_hl = 0;   // VAR 2: log2 of this map's width / height (powers of two on this path)
  BnFin fin;                  // MODE 1: finalize the BatchNorm statistics in this kernel (pointers already at this group)
  int tn = 0;                 // column-block index (ticket slot)
  const T* by2 = nullptr;     // MODE 2: second BatchNorm input fed by the same g (BwdStats::y2) ...
  float* bn_partial2 = nullptr;   // ... and this group's [2][N][tiles_m] block for it (sum g, sum g*y2)
  long long* tot = nullptr;   // this group's [2][N] fixed-point totals (BwdStats::tot); non-null: atomics instead of bn_partial
  int* tot_flag = nullptr;    // the word a tile sum beyond tot_lim units is reported in (see tl_totals_flag)
  float tot_lim = TOT_RANGE;
  __device__ __forceinline__ int tile_pixel0(int m0) const {   // pixel of the tile's first row (class offsets included)
    const int per_img = phc * pwc, per_cls = pnimg * per_img;
    const int cls = m0 / per_cls, rc = m0 - cls * per_cls, n = rc / per_img, i0 = (rc - n * per_img) >> pwl;
    return n * pimg + (2 * i0 + (cls >> 1)) * pW + (cls & 1);
  }
  template <int BM, int BN, int FM, int FN>
  __device__ __forceinline__ void finish(f32x4 (&acc)[FM][FN], char* smem, int m0, int n0) const {
    constexpr int VEC = 16 / sizeof(T);
    constexpr int ROWB = BN * (int)sizeof(T) + (sizeof(T) == 2 ? 16 : 0);   // padded row (bf16); fp32 fills 64 KB
    constexpr int CPRW = BN / VEC;                                          // 16-byte chunks per tile row
    constexpr int RPP = 256 / CPRW;                                         // rows per pass of the block
    // (t = index within a 256-thread group: the 512-thread kernel runs two such groups side by side, one per 128-row half)
    const int t = threadIdx.x & 255, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
#ifdef IEEE_DBG_NOEPI
    // measurement build only (scripts/experiments/r6_noepi.sh): no epilogue at all -- the upper bound of what hiding the
    // epilogue of a tile behind the next tile's operand fetch could give (the condition is never true; it keeps the MFMAs)
    if (m0 < 0) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) *(f32x4*)(smem + ((i * FN + j) * 256 + t) * 16) = acc[i][j];
    }
    return;
#endif
    __syncthreads();   // every wave is done reading the operand stages
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int r = wm * (BM / 2) + i * 16 + (lane & 15);
        const int c = wn * (BN / 2) + j * 16 + (lane >> 4) * 4;
        char* p = smem + r * ROWB + c * (int)sizeof(T);
        if constexpr (sizeof(T) == 4) {
          *(float4*)p = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        } else {
          *(uint2*)p = make_uint2(Vec16<bf16>::pk(acc[i][j][0], acc[i][j][1]), Vec16<bf16>::pk(acc[i][j][2], acc[i][j][3]));
        }
      }
    __syncthreads();
    const int ch = t % CPRW, r0 = t / CPRW;
    const int n = n0 + ch * VEC;
    float s1[VEC], s2[VEC], s3[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { s1[e] = 0.f; s2[e] = 0.f; s3[e] = 0.f; }
    constexpr bool has_y2 = MODE == 2 && VAR == 3;       // (its own instantiation: the extra operands cost the common form its
                                                         //  128-VGPR budget -- 86 spilled registers when it was a run-time switch)
    if (n < N) {
      // The global operands of the epilogue (residual / skip gradient, y and the ReLU mask of the BatchNorm behind this
      // dgrad) are fetched HP row passes at a time, all loads of a batch in flight before the first use: issued one pass
      // at a time behind per-pass branches they cost a full memory round trip each (up to 24 per tile, the reason the
      // block-input dgrads ran 1.5x longer than the forward convs of the same GEMM shape).  Rows past M read row M-1.
      constexpr int NP = BM / RPP;
      constexpr int HPMAX = (MODE == 2 && VAR == 3) ? 2 : 4;    // (four operand streams instead of three: two passes in flight fit the registers)
      constexpr int HP = NP > HPMAX ? HPMAX : NP;
      const int pix0 = PERM ? tile_pixel0(m0) : 0;
      const bool has_add = addend != nullptr;
      const bool has_bits = MODE == 2 && sizeof(T) == 2 && bbits != nullptr;
      const bool has_mask = MODE == 2 && (bmask != nullptr || has_bits);
      const bool has_fold = (MODE == 3) || (MODE == 2 && !has_mask && bstats != nullptr);
      float sc[VEC], sh[VEC];   // this thread's channels never change over the passes
#pragma unroll
      for (int e = 0; e < VEC; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
      if constexpr (MODE == 2 || MODE == 3) {
        if (has_fold) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) { sc[e] = bstats[2 * N + n + e]; sh[e] = bstats[3 * N + n + e]; }
        }
      }
#pragma unroll
      for (int p0 = 0; p0 < NP; p0 += HP) {
        uint4 va[HP], vy[HP], vk[HP], vy2[HP];
        unsigned kb[HP];
        int off[HP];
#pragma unroll
        for (int h = 0; h < HP; ++h) {
          const int rr = r0 + RPP * (p0 + h);
          const int m = m0 + rr;
          if constexpr (!PERM) {
            off[h] = (m < M ? m : M - 1) * (int)ld + n;
          } else {   // (M is a whole number of tiles on this path)
            off[h] = (pix0 + 2 * (rr >> pwl) * pW + 2 * (rr & (pwc - 1))) * (int)ld + n;
          }
          va[h] = vy[h] = vk[h] = vy2[h] = make_uint4(0, 0, 0, 0);
          kb[h] = 0;
        }
        if (has_add) {
          if constexpr (VAR == 2) {
#pragma unroll
            for (int h = 0; h < HP; ++h) {
              const int m = min(m0 + r0 + RPP * (p0 + h), M - 1);
              const int w = m & ((1 << as_wl) - 1), t2 = m >> as_wl, hh = t2 & ((1 << as_hl) - 1), img = t2 >> as_hl;
              const bool even = !((w | hh) & 1);
              const int ao = (((img << (as_hl - 1)) + (hh >> 1)) << (as_wl - 1)) + (w >> 1);
              const uint4 q = *(const uint4*)(addend + (even ? ao : 0) * (int)ld + n);
              va[h] = even ? q : make_uint4(0, 0, 0, 0);
            }
          } else {
#pragma unroll
            for (int h = 0; h < HP; ++h) va[h] = *(const uint4*)(addend + off[h]);
          }
        }
        if constexpr (MODE == 2) {
#pragma unroll
          for (int h = 0; h < HP; ++h) vy[h] = *(const uint4*)(by + off[h]);
          if constexpr (has_y2) {
#pragma unroll
            for (int h = 0; h < HP; ++h) vy2[h] = *(const uint4*)(by2 + off[h]);
          }
          if (has_bits) {
#pragma unroll
            for (int h = 0; h < HP; ++h) kb[h] = bbits[(unsigned)off[h] >> 3];
          } else if (has_mask) {
#pragma unroll
            for (int h = 0; h < HP; ++h) vk[h] = *(const uint4*)(bmask + off[h]);
          }
        }
#pragma unroll
        for (int h = 0; h < HP; ++h) {
          const int r = r0 + RPP * (p0 + h);
          if (m0 + r >= M) continue;
          uint4 v = *(const uint4*)(smem + r * ROWB + ch * 16);
          if constexpr (MODE == 3) {
            float f[VEC];
            Vec16<T>::unpack(v, f);
#pragma unroll
            for (int e = 0; e < VEC; ++e) f[e] = f[e] * sc[e] + sh[e];
            if (has_add) {
              float a[VEC];
              Vec16<T>::unpack(va[h], a);
#pragma unroll
              for (int e = 0; e < VEC; ++e) f[e] += a[e];
            }
            if (relu) {
#pragma unroll
              for (int e = 0; e < VEC; ++e) f[e] = fmaxf(f[e], 0.f);
            }
            v = Vec16<T>::pack(f);
          } else if (has_add || STATS) {
            float f[VEC];
            Vec16<T>::unpack(v, f);
            if (has_add) {
              float a[VEC];
              Vec16<T>::unpack(va[h], a);
#pragma unroll
              for (int e = 0; e < VEC; ++e) f[e] += a[e];
              v = Vec16<T>::pack(f);
              if (STATS) Vec16<T>::unpack(v, f);
            }
            if constexpr (MODE == 1) {
#pragma unroll
              for (int e = 0; e < VEC; ++e) { s1[e] += f[e]; s2[e] += f[e] * f[e]; }
            } else if constexpr (MODE == 2) {
              float yv[VEC];
              Vec16<T>::unpack(vy[h], yv);
              if (has_mask) {
                if (has_bits) {
#pragma unroll
                  for (int e = 0; e < VEC; ++e) f[e] = ((kb[h] >> e) & 1u) ? f[e] : 0.f;
                } else {
                  float mk[VEC];
                  Vec16<T>::unpack(vk[h], mk);
#pragma unroll
                  for (int e = 0; e < VEC; ++e) f[e] = mk[e] > 0.f ? f[e] : 0.f;
                }
                // the MASKED gradient g = dout * [out > 0] is what leaves the tile: the BatchNorm backward of that unit
                // then reads g and y only (no second pass over the mask tensor, no separate g output) -- 12 instead of
                // 20 bytes per element of the widest tensors of every block
                v = Vec16<T>::pack(f);
              } else if (has_fold) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) f[e] = (yv[e] * sc[e] + sh[e]) > 0.f ? f[e] : 0.f;
              }
#pragma unroll
              for (int e = 0; e < VEC; ++e) { s1[e] += f[e]; s2[e] += f[e] * yv[e]; }
              if constexpr (has_y2) {
                float y2v[VEC];
                Vec16<T>::unpack(vy2[h], y2v);
#pragma unroll
                for (int e = 0; e < VEC; ++e) s3[e] += f[e] * y2v[e];
              }
            }
          }
          *(uint4*)(out + off[h]) = v;
        }
      }
    }
    if constexpr (STATS) {
      // reduce over the RPP row-lanes that share a chunk, through LDS (the staged tile is consumed).  Layout: one plane of
      // 256 floats ([r0][ch] = the thread index) per (quantity, element), planes 260 floats apart: the 16 stores of a
      // thread are lane-contiguous, and the column sums below read banks 4*e + cc + 16*y -- both conflict-free (the
      // [thread][16 values] layout this replaces put lanes t and t+2 on one bank: 16-way conflicts on every store,
      // 21 % of all LDS cycles of the conv kernels by SQ_LDS_BANK_CONFLICT)
      __syncthreads();
      float* red = (float*)smem;   // [2*VEC][260]
      constexpr int PLANE = 260;
#pragma unroll
      for (int e = 0; e < VEC; ++e) { red[e * PLANE + t] = s1[e]; red[(VEC + e) * PLANE + t] = s2[e]; }
      __syncthreads();
      const bool fuse = MODE == 1 && fin.on;
      for (int idx = t; idx < BN * 2; idx += 256) {
        const int q = idx / BN, c = idx % BN;          // quantity, channel within the tile
        const int cc = c / VEC, e = c % VEC;
        const float* col = red + (q * VEC + e) * PLANE + cc;
        float s = 0.f;
#pragma unroll
        for (int y = 0; y < RPP; ++y) s += col[y * CPRW];
        if (n0 + c < N && tile_m < tiles_m) {
          if (tot != nullptr) {   // order-independent total (see tl_totals): no-return atomic, nothing to wait for
            (void)__hip_atomic_fetch_add(tot + (int64_t)q * N + n0 + c,
                                         to_fixed(s, MODE == 1 ? TOT_SCALE_FWD : TOT_SCALE_BWD, tot_lim, tot_flag),
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
          }
          float* dstp = bn_partial + ((int64_t)q * N + n0 + c) * tiles_m + tile_m;
          if (fuse) __hip_atomic_store(dstp, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through (sc1)
          else *dstp = s;
          if constexpr (has_y2) { if (q == 0) bn_partial2[((int64_t)n0 + c) * tiles_m + tile_m] = s; }   // sum g, for the second unit too
        }
      }
      if constexpr (has_y2) {
        {   // third quantity, sum g*y2, through the same planes
          __syncthreads();
#pragma unroll
          for (int e = 0; e < VEC; ++e) red[e * PLANE + t] = s3[e];
          __syncthreads();
          for (int c = t; c < BN; c += 256) {
            const int cc = c / VEC, e = c % VEC;
            const float* col = red + e * PLANE + cc;
            float s = 0.f;
#pragma unroll
            for (int y = 0; y < RPP; ++y) s += col[y * CPRW];
            if (n0 + c < N && tile_m < tiles_m) bn_partial2[((int64_t)N + n0 + c) * tiles_m + tile_m] = s;
          }
        }
      }
      if constexpr (MODE == 1) {
        if (fuse) {
          int* flag = (int*)(red + 2 * VEC * PLANE);          // one word behind the reduction planes (dynamic LDS)
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's stores (partials and output tile) have left
          __syncthreads();
          if (t == 0) {
            const int ticket = __hip_atomic_fetch_add(fin.counter + tn, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = ticket == tiles_m - 1;
          }
          __syncthreads();
          if (*flag) {                                          // workgroup-uniform: every other workgroup of this column is done
            if (t == 0) {
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              __hip_atomic_store(fin.counter + tn, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            }
            __syncthreads();
            double* dred = (double*)smem;                        // [2][BN] sums (the planes are consumed)
            __syncthreads();
            for (int idx = t; idx < BN * 2; idx += 256) {
              const int q = idx / BN, c = idx % BN;
              double sum = 0.0;
              if (n0 + c < N) {
                const float* pp = bn_partial + ((int64_t)q * N + n0 + c) * tiles_m;
                for (int rt = 0; rt < tiles_m; ++rt) sum += (double)pp[rt];     // fixed order: the same bits whoever is last
              }
              dred[idx] = sum;
            }
            __syncthreads();
            if (t < BN && n0 + t < N) {
              const int c = n0 + t;
              const double mu = dred[t] / fin.M;
              double var = dred[BN + t] / fin.M - mu * mu;
              if (var < 0) var = 0;
              const float mean = (float)mu, invstd = (float)(1.0 / sqrt(var + (double)fin.eps));
              if (fin.rm != nullptr) {
                const double unbiased = fin.M > 1 ? var * ((double)fin.M / (double)(fin.M - 1)) : var;
                fin.rm[c] = (1.f - fin.momentum) * fin.rm[c] + fin.momentum * mean;
                fin.rv[c] = (1.f - fin.momentum) * fin.rv[c] + fin.momentum * (float)unbiased;
              }
              const float sc = fin.gamma[c] * invstd;
              fin.stats[c] = mean;
              fin.stats[N + c] = invstd;
              fin.stats[2 * N + c] = sc;
              fin.stats[3 * N + c] = fin.beta[c] - mean * sc;
            }
          }
        }
      }
    }
  }
};

// fp32 split-K slab of the weight gradient, staged through LDS like the distance matrix (evaluator.hip DistEpi): the
// tile leaves as full 512-byte rows instead of 64-byte pieces scattered over 16 rows per store instruction.
// WROWS = 64: 128-row tile in two 64-row halves (each exactly the 32 KB operand stage); WROWS = 32 (HALF_M): one.
struct SlabEpi {
  static constexpr bool kStaged = true;
  float* out;
  int64_t ld;
  int M, N;
  // direct form (register-staged gemm_tn: fp32 and the element-gather loaders)
  __device__ __forceinline__ void operator()(int m, int n, f32x4 v) const {
    if (m >= M || n >= N) return;
    float* o = out + (int64_t)m * ld + n;
    if (n + 3 < N && (ld & 3) == 0) {
      *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      for (int r = 0; r < 4 && n + r < N; ++r) o[r] = v[r];
    }
  }
  template <int WROWS, int FM>
  __device__ __forceinline__ void finish(f32x4 (&acc)[FM][4], char* smem, int m0, int n0) const {
    constexpr int HALVES = WROWS == 64 ? 2 : 1;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    const int ch = t & 31, r0 = t >> 5;
    const int col = n0 + ch * 4;
    const bool vec_ok = (((uintptr_t)out | (uintptr_t)(ld * 4)) & 15) == 0 && col + 3 < N;
#pragma unroll
    for (int h = 0; h < HALVES; ++h) {
      __syncthreads();
      if (HALVES == 1 || wm == h) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = (HALVES == 1 ? wm * WROWS : 0) + i * 16 + (lane & 15);
            const int c = wn * 16 + j * 4 + (lane >> 4);
            *(float4*)(smem + r * 512 + ((c ^ (r & 31)) << 4)) =
                make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
          }
      }
      __syncthreads();
#pragma unroll
      for (int pss = 0; pss < 8; ++pss) {
        const int r = r0 + 8 * pss;
        const int row = m0 + h * 64 + r;
        if (row >= M || col >= N) continue;
        const float4 v = *(const float4*)(smem + r * 512 + ((ch ^ (r & 31)) << 4));
        float* o = out + (int64_t)row * ld + col;
        if (vec_ok) {
          *(float4*)o = v;
        } else {
          const float f[4] = {v.x, v.y, v.z, v.w};
          for (int e = 0; e < 4 && col + e < N; ++e) o[e] = f[e];
        }
      }
    }
  }
};

// ------------------------------------------------------------------ split-K fold INSIDE the weight-gradient launch (round 6)
// The plain path writes nsplit fp32 slabs and a second launch (wgrad_reduce*) adds them up: 53 launches per step, every slab
// written, read back, the OIHW gradient written.  Here the workgroups of one output tile fold their partials themselves
// (cdna_hip_programming.md, "In-launch split-K reduction" / Guideline 16): every workgroup stores its tile WRITE-THROUGH
// (sc1: no release fence, nothing stays dirty in its XCD's L2), drains, takes a ticket; the arriver that completes the
// count acquires once and adds the slabs of its group IN SPLIT ORDER -- the same bits whoever arrives last, no float
// atomics -- and either writes the OIHW gradient or, for the many-split small-weight layers (layer1 / layer2: 17-147
// splits), a level-1 slab that the last of the n1 group reducers folds the same way (radix ~ sqrt(nsplit): the serial
// tail of one reducer stays <= ~2 x 13 slab tiles instead of 147).  Tickets: int32 words the caller zeroes ONCE; every
// launch that completes leaves them zero (the last arriver resets its word).
struct FoldArgs {
  int32_t* tickets;      // [groups][tiles][n1 + 1]: word g < n1 = level-0 group g, word n1 = the level-1 fold
  float* lvl1;           // [groups][n1][Co][ncols] (unused when n1 == 1)
  float* dw;             // OIHW gradient
  int64_t dw_gs, lvl1_gs;
  int nsplit, radix, n1; // level-0 group g = splits [g * radix, min(nsplit, (g + 1) * radix))
  int accumulate, RS, Ci;
  int tiles;
};

typedef float __attribute__((address_space(1))) * gfloat_p;

__device__ __forceinline__ void store16_sc1(float* p, f32x4 v) {
  // write-through: the line leaves this XCD's L2 at once and is visible to every other XCD after the wave's vmcnt(0)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// Every wave has drained its stores; ONE lane adds to the group's ticket.  Returns (workgroup-uniform) whether this
// workgroup completed the count; it has then acquired (this CU's L1 dropped) and reset the word for the next launch.
__device__ __forceinline__ bool fold_arrive(int32_t* word, int count, int* lds_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const int old = __hip_atomic_fetch_add(word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = old == count - 1;
    if (last) {
      __hip_atomic_store(word, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    *lds_flag = last;
  }
  __syncthreads();
  const bool r = *lds_flag != 0;
  return r;
}

// sum over `count` slabs (stride `ss` floats) of NV float4 per thread at p0 + offs[i], slab order, two slabs in flight.
// Every load is unconditional (callers clamp the offsets of rows / columns outside the matrix): a per-element "load or
// not" on a run-time condition makes hipcc branch around each load and wait for it alone (guide, GEMM trap (c)).
template <int NV>
__device__ __forceinline__ void fold_sum(const float* p0, const int (&offs)[NV], int64_t ss, int count, f32x4 (&sum)[NV]) {
#pragma unroll
  for (int i = 0; i < NV; ++i) sum[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  int k = 0;
#pragma unroll 1
  for (; k + 1 < count; k += 2) {
    f32x4 a[NV], b[NV];
    const float* pa = p0 + (int64_t)k * ss;
    const float* pb = pa + ss;
#pragma unroll
    for (int i = 0; i < NV; ++i) a[i] = *(const f32x4*)(pa + offs[i]);
#pragma unroll
    for (int i = 0; i < NV; ++i) b[i] = *(const f32x4*)(pb + offs[i]);
#pragma unroll
    for (int i = 0; i < NV; ++i) { sum[i] += a[i]; sum[i] += b[i]; }
  }
  if (k < count) {
    const float* pa = p0 + (int64_t)k * ss;
    f32x4 a[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) a[i] = *(const f32x4*)(pa + offs[i]);
#pragma unroll
    for (int i = 0; i < NV; ++i) sum[i] += a[i];
  }
}

// the folded tile leaves: OIHW gradient.  (row, col) = (output channel, (rs, ci) column of the slab order)
__device__ __forceinline__ void fold_write_dw(const FoldArgs& fa, float* dw, int row, int col, int ncols, f32x4 v) {
  if (fa.RS == 1) {
    f32x4* d = (f32x4*)(dw + (int64_t)row * ncols + col);
    if (fa.accumulate) v += *d;
    *d = v;
  } else {
    const int rs = col / fa.Ci, ci = col - rs * fa.Ci;
    float* d = dw + ((int64_t)row * fa.Ci + ci) * fa.RS + rs;
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e * fa.RS] = fa.accumulate ? d[e * fa.RS] + v[e] : v[e];
  }
}

// The TN kernel's epilogue in fold form.  Tile = WROWS * 2 (or 1) x 128 of [Co][ncols]; a thread owns chunk ch = t & 31 of
// rows r0 + 8 i of each 64-row half, exactly the row form SlabEpi::finish stores in.
struct SlabFoldEpi {
  static constexpr bool kStaged = true;
  float* out;          // this split's level-0 slab [Co][ncols]
  float* slab0;        // split 0's slab of this group (modality)
  int64_t ld;
  int M, N;
  FoldArgs fa;
  int ks, tile, z;
  __device__ __forceinline__ void operator()(int, int, f32x4) const {}
  template <int WROWS, int FM>
  __device__ __forceinline__ void finish(f32x4 (&acc)[FM][4], char* smem, int m0, int n0) const {
    constexpr int HALVES = WROWS == 64 ? 2 : 1;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    const int ch = t & 31, r0 = t >> 5;
    const int col = n0 + ch * 4;
    const bool col_ok = col + 3 < N;
    // 1. this split's tile -> its level-0 slab, write-through, as full 512-byte rows out of LDS
#pragma unroll
    for (int h = 0; h < HALVES; ++h) {
      __syncthreads();
      if (HALVES == 1 || wm == h) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = (HALVES == 1 ? wm * WROWS : 0) + i * 16 + (lane & 15);
            const int c = wn * 16 + j * 4 + (lane >> 4);
            *(f32x4*)(smem + r * 512 + ((c ^ (r & 31)) << 4)) = acc[i][j];
          }
      }
      __syncthreads();
#pragma unroll
      for (int pss = 0; pss < 8; ++pss) {
        const int r = r0 + 8 * pss;
        const int row = m0 + h * 64 + r;
        if (row >= M || !col_ok) continue;
        store16_sc1(out + (int64_t)row * ld + col, *(const f32x4*)(smem + r * 512 + ((ch ^ (r & 31)) << 4)));
      }
    }
    // 2. arrive at this split's level-0 group
    int* flag = (int*)smem;
    const int g = ks / fa.radix;
    const int gfirst = g * fa.radix, gcount = min(fa.radix, fa.nsplit - gfirst);
    int32_t* words = fa.tickets + ((int64_t)z * fa.tiles + tile) * (fa.n1 + 1);
    if (!fold_arrive(words + g, gcount, flag)) return;
    // 3. fold the group (and, as the last group reducer, the level-1 slabs); 4 rows x 16 bytes per thread and pass
    const int64_t ss = (int64_t)M * ld;
    float* dwz = fa.dw + (int64_t)z * fa.dw_gs;
    float* l1 = fa.lvl1 + (int64_t)z * fa.lvl1_gs;
    const int colc = col_ok ? col : 0;
#pragma unroll 1
    for (int level = 0; level < 2; ++level) {
      const float* src = level == 0 ? slab0 + (int64_t)gfirst * ss : l1;
      const int count = level == 0 ? gcount : fa.n1;
      const bool final_level = level == 1 || fa.n1 == 1;
#pragma unroll 1
      for (int pass = 0; pass < HALVES * 2; ++pass) {
        int offs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) offs[i] = min(m0 + pass * 32 + r0 + 8 * i, M - 1) * (int)ld + colc;
        f32x4 sum[4];
        fold_sum<4>(src, offs, ss, count, sum);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = m0 + pass * 32 + r0 + 8 * i;
          if (row >= M || !col_ok) continue;
          if (final_level) fold_write_dw(fa, dwz, row, col, N, sum[i]);
          else store16_sc1(l1 + (int64_t)g * ss + offs[i], sum[i]);
        }
      }
      if (final_level) return;
      if (!fold_arrive(words + fa.n1, fa.n1, flag)) return;
    }
  }
};

struct ConvArgs {
  GatherGeom g;
  int M, N;          // GEMM rows (pixels) / cols (output channels of this GEMM)
  int ldw, ktiles;   // packed-weight row length (elements), number of k-tiles
  int tiles_m, tiles_n;
  int group;         // m-tiles per group in the grouped tile order
  int64_t src_gs, w_gs, dst_gs;   // per-modality strides (elements)
  int stagger = 0;   // every second workgroup of an XCD starts `stagger` x ~0.43 us late (see conv_gather_kernel)
};

// forward and dgrad share this kernel (they differ only in the gather geometry)
struct BwdStats {            // MODE 2 operands (per-group strides in elements / floats)
  const void* y;
  const void* mask;
  const float* stats;
  int64_t act_gs, stats_gs;
  int relu;                  // MODE 3 only
  int mask_bits = 0;         // MODE 2: `mask` is the packed bit form (one byte per 8 channels)
  int addend_s2 = 0;         // MODE 2: the addend is compact at stride 2 (StagedStoreEpi VAR 2)
  BnFin fin;                 // MODE 1: fused BatchNorm finalize (group-0 pointers; the kernels step them by blockIdx.y)
  // MODE 2, optional: a SECOND BatchNorm fed by the same gradient g (the downsample branch of the block whose output this
  // dgrad differentiates): y2 = its input tensor (same shape and strides as y), partial2 = its own [2][N][tiles_m] block per
  // group, which receives sum g (again) and sum g*y2 -- that unit's ieee_bn2d_bwd then skips its reduction pass too
  const void* y2 = nullptr;
  float* partial2 = nullptr;
  long long* tot = nullptr;     // MODE 1 / 2: add the tile's sums to these [groups][2][N] fixed-point totals instead of writing
  int64_t tot_gs = 0;           //             bn_partial (see tl_totals)
  int tot_rep = 1;              //             replicas [tot_rep][groups][2][N]: row tile tm adds to replica tm % tot_rep
  int64_t tot_rs = 0;           //             elements between replicas
  int* tot_flag = nullptr;      //             overflow report word of this direction (tl_totals_flag + 0 / + 1), may be null
  float tot_lim = TOT_RANGE;    //             a tile's share of the range: 2^62 / row tiles of the launch
};

// MODE 1: hand the epilogue its group's finalize operands
template <class Epi> __device__ __forceinline__ void set_fin(Epi& epi, const BwdStats& bs, int z, int tn, int tiles_n, int N) {
  if (!bs.fin.on) return;
  epi.fin = bs.fin;
  epi.fin.gamma += z * bs.fin.param_gs;
  epi.fin.beta += z * bs.fin.param_gs;
  if (bs.fin.rm != nullptr) { epi.fin.rm += z * bs.fin.buf_gs; epi.fin.rv += z * bs.fin.buf_gs; }
  epi.fin.stats += (int64_t)z * 4 * N;
  epi.fin.counter += z * tiles_n;
  epi.tn = tn;
}

// (forcing 4 waves per SIMD here spills 80 VGPRs and is 2.5x slower; the default allocation gives 3)
// PIPE = 0: register staging, one LDS stage (high occupancy: the many-workgroup layers).  PIPE = 2..4: LDS-DMA ring
// of PIPE stages with PIPE-1 k-tiles in flight (few-workgroup layers, where no co-resident workgroup hides the
// load latency of a one-tile-deep pipeline); bf16 fast path only.
template <typename T, int BN, bool SLOW, int MODE, int PIPE = 0, int VAR = 0>
__global__ __launch_bounds__(256, (BN == 256 ? 2 : (PIPE == 1 ? 4 : ((PIPE == 5 || PIPE == 6) ? 3 : 1)))) void conv_gather_kernel(const T* __restrict__ src, const T* __restrict__ w,
                                                          T* __restrict__ dst, const T* __restrict__ addend,
                                                          float* __restrict__ bn_partial, ConvArgs a, BwdStats bs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // Phase stagger (round 6, IEEE_GATHER_STAGGER): a launch of one or two rounds of identical workgroups runs in LOCKSTEP --
  // every workgroup of the chip is in its k-loop (L2 -> LDS path saturated, HBM store path idle), then every workgroup is in
  // its epilogue (the reverse): the measurement build without epilogues shows 15-60 % of such a launch is epilogue time that
  // nothing overlaps.  Every second workgroup of an XCD therefore starts a fraction of a tile late, so that the two halves
  // of the co-resident workgroups sit in opposite phases.
  if (a.stagger > 0 && ((blockIdx.x >> 3) & 1)) {
    for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(16);
  }
  int tm, tn;
  tile_map_xy(a.tiles_m, a.tiles_n, a.group, tm, tn);
  const int m0 = tm * 128, n0 = tn * BN;
  const int z = blockIdx.y;
  src += z * a.src_gs;
  w += z * a.w_gs;
  dst += z * a.dst_gs;
  if (addend != nullptr) addend += z * (VAR == 2 ? a.dst_gs / 4 : a.dst_gs);   // (VAR 2: the compact addend is a quarter map)
  constexpr bool PERM = VAR == 1;
  StagedStoreEpi<T, MODE, VAR> epi{0, dst, addend, (MODE == 1 || MODE == 2) ? bn_partial + (int64_t)z * a.tiles_m * 2 * a.N : nullptr, a.N, a.M, a.N, tm, a.tiles_m,
                              MODE == 2 ? (const T*)bs.y + z * bs.act_gs : nullptr,
                              (MODE == 2 && bs.mask && !bs.mask_bits) ? (const T*)bs.mask + z * bs.act_gs : nullptr,
                              ((MODE == 2 || MODE == 3) && bs.stats) ? bs.stats + z * bs.stats_gs : nullptr};
  if constexpr (MODE == 3) epi.relu = bs.relu;
  if constexpr (MODE == 1) set_fin(epi, bs, z, tn, a.tiles_n, a.N);
  if constexpr (MODE == 1 || MODE == 2) {
    if (bs.tot != nullptr) { epi.tot = bs.tot + z * bs.tot_gs + (tm & (bs.tot_rep - 1)) * bs.tot_rs; epi.tot_flag = bs.tot_flag; epi.tot_lim = bs.tot_lim; }
  }
  if constexpr (VAR == 2) { epi.as_wl = __ffs(a.g.Wo) - 1; epi.as_hl = __ffs(a.g.Ho) - 1; }
  if constexpr (MODE == 2) {
    if (bs.mask && bs.mask_bits) epi.bbits = (const uint8_t*)bs.mask + z * (bs.act_gs >> 3);
    if (bs.y2) { epi.by2 = (decltype(epi.by2))bs.y2 + z * bs.act_gs; epi.bn_partial2 = bs.partial2 + (int64_t)z * a.tiles_m * 2 * a.N; }
  }
  if constexpr (PIPE > 0) {   // both operands through LDS-DMA
    static_assert(!SLOW && sizeof(T) == 2, "the LDS-DMA ring is the bf16 vector path");
    const int ch = nt_dma_chunk(threadIdx.x);
    // the packed weights are a plain [N][ldw] matrix (ldw = ktiles * 64, zero padded): lean issue, see glds16_lean
    LoaderPlainLean<BN / 32> lbd;
    lbd.init(w, a.ldw, n0, a.N, ch);
    if constexpr (PERM) {
      // stride-2 dgrad, GEMM rows in parity-class-major order (LoaderIm2colNT<.., true>): dead taps are skipped
      epi.pwc = a.g.Wo >> 1; epi.pwl = __ffs(epi.pwc) - 1; epi.phc = a.g.Ho >> 1; epi.pW = a.g.Wo; epi.pimg = a.g.Ho * a.g.Wo;
      epi.pnimg = a.g.npix / (a.g.Ho * a.g.Wo);
      LoaderIm2colNT<T, 4, true> la;
      la.init(src, a.g, m0, ch);
      gemm_nt_dma<128, BN, PIPE>(la, lbd, epi, a.ktiles, m0, n0, smem);
    } else if (a.g.R == 1 && a.g.S == 1 && a.g.mul == 1 && a.g.off == 0 && a.g.div == 1) {
      // 1x1 / stride 1 / no padding (two thirds of the launches): im2col(X) is X itself, a plain [pixels][C] matrix --
      // no pixel decode, no tap masks (their set-up rivals the whole k-loop of the K = 64..256 layers), and the same
      // lean issue as the weights (C is a multiple of 64 on this path: no k tail)
      LoaderPlainLean<4> la;
      la.init(src, a.g.Cs, m0, a.g.npix, ch);
      gemm_nt_dma<128, BN, PIPE>(la, lbd, epi, a.ktiles, m0, n0, smem);
    } else {
      LoaderIm2colNT<T, 4> la;
      la.init(src, a.g, m0, ch);
      gemm_nt_dma<128, BN, PIPE>(la, lbd, epi, a.ktiles, m0, n0, smem);
    }
    return;
  } else {
  LoaderPlainNT<T, BN / 32> lb;
  lb.init(w, a.ldw, n0, a.N, a.ldw);
  if constexpr (SLOW) {
    LoaderIm2colSlowNT<T, 4> la;
    la.init(src, a.g, m0);
    gemm_nt<T, 128, BN, (sizeof(T) == 2 ? 1 : 2)>(la, lb, epi, a.ktiles, m0, n0, smem);
  } else {
    LoaderIm2colNT<T, 4> la;
    la.init(src, a.g, m0);
    gemm_nt<T, 128, BN, (sizeof(T) == 2 ? 1 : 2)>(la, lb, epi, a.ktiles, m0, n0, smem);
  }
  }
}


// ------------------------------------------------------------------ 3x3 / stride 1 / pad 1 with the input patch in LDS
// The implicit-GEMM kernel above fetches the A operand of a 3x3 conv once PER TAP: nine L2 -> LDS transfers of (mostly)
// the same pixels per 64-channel chunk, and the per-CU L2 -> LDS path is what bounds it (LABNOTES.md section 4).  Here a
// workgroup's 128 GEMM rows are RT whole rows of one image (RT x Wm = 128: 16 x 8, 8 x 16 or 4 x 32 pixels) and the input
// patch those rows see -- (RT + 2) x (Wm + 2) pixels x 64 channels, zero halo included -- is brought into LDS ONCE per
// channel chunk; the nine taps read their A fragments from it at shifted addresses (tap (dr, ds) = a constant byte offset
// folded into the ds_read), and only the weights stream per k-tile.  Operand bytes per 9 k-tiles of a 128 x 128 tile:
// 9 x 32 KB -> 23 KB + 9 x 16 KB.
// LDS image of the patch: pixel (h', w') (halo coordinates, h' = h - h0 + 1, w' = w + 1) at h' * PITCH + w' * 128, its
// eight 16-byte channel chunks XOR-swizzled by key(w') = w' & 7 (the LDS-DMA writes lane-linearly, so the swizzle sits on
// the SOURCE address: the lane whose LDS position is chunk slot f of pixel w' fetches logical chunk f ^ key).  With that key
// every ds_read_b128 of a fragment (16 consecutive pixels of a row; for the 8-wide maps 2 x 8) is bank-conflict free at
// all three column shifts (checked exhaustively over the instruction's 16-lane groups).  The halo cells are zeroed once:
// the DMA only ever writes pixels that exist.
// Forward and dgrad (of a stride-1 conv) are the same kernel: the dgrad reads dY with the taps mirrored (g.sgn < 0: tap
// (dr, ds) pairs with weight column 8 - t of Wd).  Both styles pull all fragments of a k-tile into registers first, so the
// next DMA runs under this tile's MFMAs.  STYLE 1: one weight stage (the tile of tap t + 1 is issued at tap t and waited
// for at once at tap t + 1).  STYLE 0: two weight stages, the tile of tap t + 2 issued at tap t (a whole tap period to land).
template <int WLOG> struct PatchGeom {
  static constexpr int Wm = 1 << WLOG, RT = 128 / Wm;
  static constexpr int PITCH = (Wm + 2) * 128;
  static constexpr int ROWS = RT + 2;
  static constexpr int BYTES = ROWS * PITCH;
  static constexpr int PARTS = Wm / 8;           // LDS-DMA instructions (8 pixels x 128 B) per image row
};

template <int BN, int MODE, int WLOG, int STYLE>
__global__ __launch_bounds__(256, 3) void conv3x3_patch_kernel(const bf16* __restrict__ src, const bf16* __restrict__ w,
                                                                              bf16* __restrict__ dst, const bf16* __restrict__ addend,
                                                                              float* __restrict__ bn_partial, ConvArgs a, BwdStats bs) {
  typedef PatchGeom<WLOG> PG;
  typedef ImgNT<bf16> Img;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tm, tn;
  tile_map_xy(a.tiles_m, a.tiles_n, a.group, tm, tn);
  const int m0 = tm * 128, n0 = tn * BN;
  const int z = blockIdx.y;
  src += z * a.src_gs;
  w += z * a.w_gs;
  dst += z * a.dst_gs;
  if (addend != nullptr) addend += z * a.dst_gs;
  StagedStoreEpi<bf16, MODE, 0> epi{0, dst, addend, (MODE == 1 || MODE == 2) ? bn_partial + (int64_t)z * a.tiles_m * 2 * a.N : nullptr, a.N, a.M, a.N, tm, a.tiles_m,
                                    MODE == 2 ? (const bf16*)bs.y + z * bs.act_gs : nullptr,
                                    (MODE == 2 && bs.mask && !bs.mask_bits) ? (const bf16*)bs.mask + z * bs.act_gs : nullptr,
                                    ((MODE == 2 || MODE == 3) && bs.stats) ? bs.stats + z * bs.stats_gs : nullptr};
  if constexpr (MODE == 3) epi.relu = bs.relu;
  if constexpr (MODE == 1) set_fin(epi, bs, z, tn, a.tiles_n, a.N);
  if constexpr (MODE == 1 || MODE == 2) {
    if (bs.tot != nullptr) { epi.tot = bs.tot + z * bs.tot_gs + (tm & (bs.tot_rep - 1)) * bs.tot_rs; epi.tot_flag = bs.tot_flag; epi.tot_lim = bs.tot_lim; }
  }
  if constexpr (MODE == 2) {
    if (bs.mask && bs.mask_bits) epi.bbits = (const uint8_t*)bs.mask + z * (bs.act_gs >> 3);
    if (bs.y2) { epi.by2 = (decltype(epi.by2))bs.y2 + z * bs.act_gs; epi.bn_partial2 = bs.partial2 + (int64_t)z * a.tiles_m * 2 * a.N; }
  }
  constexpr int FM = 4, FN = BN / 32;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  char* patch = smem;
  char* bst = smem + PG::BYTES;
  for (int i = t * 16; i < PG::BYTES; i += 256 * 16) *(uint4*)(patch + i) = make_uint4(0, 0, 0, 0);

  const int Hm = a.g.Hs, Cs = a.g.Cs;
  const int tiles_per_img = Hm / PG::RT;
  const int img = tm / tiles_per_img, h0 = (tm - img * tiles_per_img) * PG::RT;   // workgroup-uniform
  const bool flip = a.g.sgn < 0;
  // patch loader: lane = (pixel j of the 8-pixel segment, chunk slot f)
  const int lj = lane >> 3, lf = lane & 7;
  const unsigned voffA = (unsigned)((lj * Cs + ((lf ^ ((1 + lj) & 7)) << 3)) * 2);
  const char* srcimg = (const char*)(src + (int64_t)img * Hm * PG::Wm * Cs);
  LoaderPlainLean<BN / 32> lb;
  lb.init(w, a.ldw, n0, a.N, nt_dma_chunk(t));
  const char* const wbase = lb.base;

  // A fragments: lane (m = lane & 15, k-chunk lane >> 4) of fragment i of this wave's 64 rows
  int w_l, h_l;
  if constexpr (WLOG == 3) { w_l = lane & 7; h_l = wm * 8 + ((lane >> 3) & 1); }
  else if constexpr (WLOG == 4) { w_l = lane & 15; h_l = wm * 4; }
  else { w_l = lane & 15; h_l = wm * 2; }
  unsigned abase[3][2];
#pragma unroll
  for (int dsi = 0; dsi < 3; ++dsi)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int wp = w_l + dsi;                        // w' = w + 1 + ds, ds = dsi - 1
      const int c = kk * 4 + (lane >> 4);
      abase[dsi][kk] = (unsigned)(h_l * PG::PITCH + wp * 128 + ((c ^ (wp & 7)) << 4));
    }
  auto a_imm = [](int i, int dri) {                    // dri = dr + 1: patch row h' = h_rel + dri
    if constexpr (WLOG == 3) return (2 * i + dri) * PG::PITCH;
    else if constexpr (WLOG == 4) return (i + dri) * PG::PITCH;
    else return ((i >> 1) + dri) * PG::PITCH + (i & 1) * 16 * 128;
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto issue_patch = [&](int q) {
    constexpr int NI = PG::ROWS * PG::PARTS;
#pragma unroll
    for (int k = 0; k < (NI + 3) / 4; ++k) {
      const int jj = wave_u + 4 * k;
      if (jj < NI) {
        const int prow = jj / PG::PARTS, part = jj % PG::PARTS;
        const int h = h0 + prow - 1;
        if ((unsigned)h < (unsigned)Hm)
          glds16_s(voffA, srcimg + ((int64_t)(h * PG::Wm + part * 8) * Cs + q * 64) * 2, patch + prow * PG::PITCH + (part * 8 + 1) * 128);
      }
    }
  };
  auto issue_b = [&](int q, int tap) {
    const int tb = flip ? 8 - tap : tap;
    lb.base = wbase + ((int64_t)tb * Cs + q * 64) * 2;
    glds16_lean<BN / 32>(lb.off, lb.base, bst + (8 * wave_u) * 128);
  };
  const char* Bt = bst + (wn * (BN / 2)) * 128;
  const int nch = Cs >> 6;

  __syncthreads();          // the zeroed halo is in place before the first DMA lands
  if constexpr (STYLE == 0) {
    // two weight stages, up to two k-tiles in flight: the weights of tap t + 2 are issued as soon as every wave holds the
    // fragments of tap t, so each tile has a whole tap period (fragment reads + 16-32 MFMAs) to land before its counted wait.
    // vmcnt order per wave: [B(t+1)] [patch of the next chunk, issued at tap 8] [B(t+2)] -- "all but my youngest BCH" retires
    // the tile about to be read and, at a chunk boundary, the patch.
    constexpr int BCH = BN / 32;
    auto issue_b2 = [&](int q, int tap, int buf) {
      const int tb = flip ? 8 - tap : tap;
      lb.base = wbase + ((int64_t)tb * Cs + q * 64) * 2;
      glds16_lean<BCH>(lb.off, lb.base, bst + buf * (BN * 128) + (8 * wave_u) * 128);
    };
    issue_patch(0);
    issue_b2(0, 0, 0);
    issue_b2(0, 1, 1);
    for (int q = 0; q < nch; ++q) {
      const int par = q & 1;            // 9 taps per chunk: the stage of (q, tap) is (q + tap) & 1
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int dri = tap / 3, dsi = tap % 3;
        const bool lastq = q == nch - 1;
        const bool more1 = !(lastq && tap == 8), more2 = !(lastq && tap >= 7);   // a tile t+1 / t+2 exists
        if (more1) wait_vmcnt<BCH>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        const int buf = (tap & 1) ^ par;
        const char* Bs = Bt + buf * (BN * 128);
        Img::Frag fa[2][FM], fb[2][FN];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
          for (int i = 0; i < FM; ++i)
            fa[kk][i] = __builtin_bit_cast(bf16x8, *(const uint4*)(patch + abase[dsi][kk] + a_imm(i, dri)));
#pragma unroll
          for (int j = 0; j < FN; ++j) fb[kk][j] = Img::frag(Bs, j * 16, kk, lane);
        }
        if (more2 || (tap == 8 && !lastq)) {
          __syncthreads();                // every wave holds its fragments: this stage (and, at tap 8, the patch) is free
          if (tap == 8 && !lastq) issue_patch(q + 1);
          if (more2) {
            if (tap >= 7) issue_b2(q + 1, tap - 7, buf);
            else issue_b2(q, tap + 2, buf);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = mfma16(fb[kk][j], fa[kk][i], acc[i][j]);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) asm volatile("" : "+v"(acc[i][j]));
      }
    }
  } else {
  issue_patch(0);
  issue_b(0, 0);
  for (int q = 0; q < nch; ++q) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dri = tap / 3, dsi = tap % 3;
      const bool last = (q == nch - 1) && (tap == 8);
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();       // this k-tile (and, at tap 0, the patch) has landed for every wave
      {
        Img::Frag fa[2][FM], fb[2][FN];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
          for (int i = 0; i < FM; ++i)
            fa[kk][i] = __builtin_bit_cast(bf16x8, *(const uint4*)(patch + abase[dsi][kk] + a_imm(i, dri)));
#pragma unroll
          for (int j = 0; j < FN; ++j) fb[kk][j] = Img::frag(Bt, j * 16, kk, lane);
        }
        if (!last) {
          __syncthreads();                // every wave holds its fragments -> stage (and patch at tap 8) may be overwritten
          if (tap == 8) { issue_patch(q + 1); issue_b(q + 1, 0); }
          else issue_b(q, tap + 1);
        }
        __builtin_amdgcn_sched_barrier(0);   // the MFMAs stay behind the DMA issue ...
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = mfma16(fb[kk][j], fa[kk][i], acc[i][j]);
        // ... and in front of the next tap's vmcnt(0) wait: left alone, hipcc sinks them (they touch no memory) below that
        // wait and its barrier, and the DMA just issued is waited for with nothing running under it
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) asm volatile("" : "+v"(acc[i][j]));
      }
    }
  }
  }
  epi.template finish<128, BN, FM, FN>(acc, smem, m0, n0);
}

template <int BN, int MODE, int WLOG, int STYLE>
static void launch_patch_inst(dim3 grid, hipStream_t st, const bf16* src, const bf16* w, bf16* dst, const bf16* addend,
                              float* bn_partial, const ConvArgs& a, const BwdStats& bs) {
  const size_t smem = PatchGeom<WLOG>::BYTES + (size_t)BN * 128 * (STYLE == 0 ? 2 : 1);   // STYLE 0: two weight stages
  launch_timed(conv3x3_patch_kernel<BN, MODE, WLOG, STYLE>, grid, smem, st, src, w, dst, addend, bn_partial, a, bs);
}
template <int BN, int WLOG, int STYLE>
static void launch_patch_mode(int mode, dim3 grid, hipStream_t st, const bf16* src, const bf16* w, bf16* dst, const bf16* addend,
                              float* bn_partial, const ConvArgs& a, const BwdStats& bs) {
  if (mode == 3) launch_patch_inst<BN, 3, WLOG, STYLE>(grid, st, src, w, dst, addend, nullptr, a, bs);
  else if (mode == 2) launch_patch_inst<BN, 2, WLOG, STYLE>(grid, st, src, w, dst, addend, bn_partial, a, bs);
  else if (mode == 1) launch_patch_inst<BN, 1, WLOG, STYLE>(grid, st, src, w, dst, addend, bn_partial, a, bs);
  else launch_patch_inst<BN, 0, WLOG, STYLE>(grid, st, src, w, dst, addend, nullptr, a, bs);
}
template <int BN, int STYLE>
static void launch_patch_w(int wlog, int mode, dim3 grid, hipStream_t st, const bf16* src, const bf16* w, bf16* dst,
                           const bf16* addend, float* bn_partial, const ConvArgs& a, const BwdStats& bs) {
  if (wlog == 3) launch_patch_mode<BN, 3, STYLE>(mode, grid, st, src, w, dst, addend, bn_partial, a, bs);
  else if (wlog == 4) launch_patch_mode<BN, 4, STYLE>(mode, grid, st, src, w, dst, addend, bn_partial, a, bs);
  else launch_patch_mode<BN, 5, STYLE>(mode, grid, st, src, w, dst, addend, bn_partial, a, bs);
}
// the layers this form covers: 3x3, stride 1, pad 1, same-size maps of width 8 / 16 / 32 whose 128-row tiles are whole
// image rows, channel counts in whole 64-chunks (every conv2 of the stride-1 bottlenecks, forward and dgrad)
static bool patch_eligible(const GatherGeom& g, int M) {
  static const bool on = !(getenv("IEEE_CONV_PATCH") && atoi(getenv("IEEE_CONV_PATCH")) == 0);
  if (!on || g.R != 3 || g.S != 3 || g.mul != 1 || g.div != 1 || g.perm) return false;
  if (!((g.sgn == 1 && g.off == -1) || (g.sgn == -1 && g.off == 1))) return false;
  if (g.Hs != g.Ho || g.Ws != g.Wo || (g.Ws != 8 && g.Ws != 16 && g.Ws != 32)) return false;
  if (g.Cs % 64 != 0 || g.Hs % (128 / g.Ws) != 0 || M % 128 != 0) return false;
  return true;
}



// ------------------------------------------------------------------ the stem as a direct convolution over an LDS patch
// The stem runs as an 8x8 / stride-2 conv over the border-padded 4-channel image (net.hip).  Through the implicit-GEMM
// kernel its im2col operand is a 16x expansion of the image streamed L2 -> LDS per k-tile (805 MB for 54 MB of pixels at
// B = 64: 137 us for 255 MB of HBM traffic).  Here a workgroup owns 2 output rows x 64 columns (= 128 GEMM rows) and brings
// the 10 input rows they see into LDS as they lie in memory (one contiguous 10.7 KB run: 11 LDS-DMA instructions);
// the A fragment of k-step ks (= filter row ks: 8 taps x 4 channels = 32 k) for output column ow is the 16 bytes at
// row (2*oh + ks), pixel 2*ow + 2*(lane >> 4) -- consecutive output columns are 16 bytes apart, so a fragment read is a
// contiguous 256-byte run per 16 lanes: conflict-free with no swizzle and no address arithmetic in the loop.  The packed
// weights (64 x 256 bf16 = 32 KB) never enter LDS: each wave keeps its B fragments of all 8 k-steps in registers.
// No barrier inside the k-loop.  Same staged epilogue (BatchNorm sums / folded eval BatchNorm) as the GEMM kernel.
// WALK > 1 (round 6): the workgroup is persistent over WALK consecutive tiles (2 * WALK output rows of one image).  The 32 KB of
// packed weights -- 64 KB of L2 -> register traffic per workgroup, six times the 10.7 KB patch -- are fetched ONCE instead of
// once per two output rows, and the next tile's patch lands (second patch region) under this tile's MFMAs and staged epilogue.
// LDS: epilogue staging 18 KB + 2 x 11 KB = 40 KB: still 4 workgroups per CU.
constexpr int STEM_EPI_BYTES = 128 * (64 * 2 + 16), STEM_PATCH_BYTES = 11 * 1024;
template <int MODE, int WALK = 1>
__global__ __launch_bounds__(256, 4) void stem_conv_kernel(const bf16* __restrict__ src, const bf16* __restrict__ w,
                                                           bf16* __restrict__ dst, float* __restrict__ bn_partial,
                                                           ConvArgs a, BwdStats bs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tg, tn;
  tile_map_xy(a.tiles_m / WALK, 1, a.group, tg, tn);
  const int z = blockIdx.y;
  src += z * a.src_gs;
  w += z * a.w_gs;
  dst += z * a.dst_gs;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int Hp = a.g.Hs, Wp = a.g.Ws, tiles_per_img = a.g.Ho >> 1;
  const int rowb = Wp * 8;                                   // bytes per input row (4 bf16 channels per pixel)
  const int patch_bytes = 10 * rowb;
  const int ninstr = (patch_bytes + 1023) >> 10;
  // WALK == 1: the patch shares the staging region (it is consumed before the epilogue stages); else two regions behind it
  auto patch_at = [&](int rep) -> char* { return WALK == 1 ? smem : smem + STEM_EPI_BYTES + (rep & 1) * STEM_PATCH_BYTES; };
  auto fetch_patch = [&](int tm, char* dstp) {
    const int img = tm / tiles_per_img, oh0 = (tm - img * tiles_per_img) * 2;
    const char* pbase = (const char*)src + ((int64_t)(img * Hp + 2 * oh0) * Wp) * 8;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = wave_u + 4 * k;
      if (j < ninstr) {
        const unsigned off = (unsigned)min(j * 1024 + lane * 16, patch_bytes - 16);   // the tail lanes re-read the last chunk
        glds16_s(off, pbase, dstp + j * 1024);
      }
    }
  };
  fetch_patch(tg * WALK, patch_at(0));
  // B fragments of all 8 k-steps: lane (n = lane & 15, k-chunk lane >> 4) of n-fragment j
  bf16x8 fb[8][2];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      fb[ks][j] = __builtin_bit_cast(bf16x8, *(const uint4*)(w + (int64_t)(wn * 32 + j * 16 + (lane & 15)) * a.ldw + ks * 32 + (lane >> 4) * 8));
  const unsigned abase = (unsigned)(2 * wm * rowb + (lane & 15) * 16 + (lane >> 4) * 16);
#pragma unroll 1
  for (int rep = 0; rep < WALK; ++rep) {
    const int tm = tg * WALK + rep;
    StagedStoreEpi<bf16, MODE, 0> epi{0, dst, nullptr, MODE == 1 ? bn_partial + (int64_t)z * a.tiles_m * 2 * a.N : nullptr, a.N, a.M, a.N, tm, a.tiles_m,
                                      nullptr, nullptr, (MODE == 3 && bs.stats) ? bs.stats + z * bs.stats_gs : nullptr};
    if constexpr (MODE == 3) epi.relu = bs.relu;
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    wait_vmcnt<0>();                      // this tile's patch has landed (and the previous tile's stores have left)
    __syncthreads();                      // ... for every wave; the other patch region was last read a tile ago
    if (WALK > 1 && rep + 1 < WALK) fetch_patch(tm + 1, patch_at(rep + 1));
    const char* patch = patch_at(rep);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      bf16x8 fa[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = __builtin_bit_cast(bf16x8, *(const uint4*)(patch + abase + ks * rowb + i * 256));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(fb[ks][j], fa[i], acc[i][j]);
    }
    epi.template finish<128, 64, 4, 2>(acc, smem, tm * 128, 0);
  }
}

// the shapes the direct stem covers: 8x8 / stride 2 / no padding over 4 channels, 64 output channels, 64 output columns
static bool stem_eligible(const GatherGeom& g, int N, int ldw, int mode, const void* addend) {
  static const bool on = !(getenv("IEEE_STEM_DIRECT") && atoi(getenv("IEEE_STEM_DIRECT")) == 0);
  return on && g.R == 8 && g.S == 8 && g.Cs == 4 && g.mul == 2 && g.off == 0 && g.sgn == 1 && g.div == 1 && !g.perm && N == 64 &&
         g.Wo == 64 && (g.Ho & 1) == 0 && g.Ws == 2 * 64 + 6 && g.Hs == 2 * g.Ho + 6 && ldw == 256 && (mode == 0 || mode == 1 || mode == 3) &&
         addend == nullptr;
}


// The stem's weight gradient the same way: dW[co][(r, s, c)] = sum over output pixels of dY[pix][co] * x[2*oh + r][2*ow + s][c].
// A workgroup owns STEM_WG_ROWS output rows of one image (one split-K slab) and walks them two rows (128 pixels) at a
// time: the 10 input rows go into LDS as they lie in memory, the 128 x 64 dY tile beside them (chunks XOR-swizzled on the
// DMA's source side), and BOTH MFMA operands come out through ds_read_b64_tr_b16 -- dY transposed as in the TN core, the
// im2col operand straight from the raw patch: for a fixed filter row r the 16 GEMM columns (4 taps x 4 channels) of an
// output pixel are 32 contiguous bytes of the image, and consecutive output pixels are 16 bytes apart.  No im2col
// expansion anywhere (the implicit-GEMM form re-reads the image 16x from L2: 177 us for 255 MB of HBM traffic).
// Each wave accumulates all 64 output channels x 64 of the 256 columns (2 filter rows); slabs are reduced by the common
// wgrad_reduce kernel.
constexpr int STEM_WG_ROWS = 32;
__global__ __launch_bounds__(256, 3) void stem_wgrad_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                            float* __restrict__ slab, int Hp, int Wp, int Ho,
                                                            int64_t dy_gs, int64_t x_gs, int64_t slab_gs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const int z = blockIdx.y;
  const int splits_per_img = Ho / STEM_WG_ROWS;
  const int split = blockIdx.x, img = split / splits_per_img, oh_first = (split - img * splits_per_img) * STEM_WG_ROWS;
  dy += z * dy_gs;
  x += z * x_gs;
  slab += z * slab_gs + (int64_t)split * 64 * 256;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int rowb = Wp * 8, patch_bytes = 10 * rowb, ninstr = (patch_bytes + 1023) >> 10;
  char* patch = smem;                       // 12 KB region
  char* dtile = smem + 12 * 1024;           // [128 pixels][128 B], chunk slots XOR key(pixel)
  // dY loader: instruction covers 8 pixels x 8 chunks; lane (pixel j = lane >> 3, slot f = lane & 7) fetches logical
  // chunk f ^ key(pixel), key = 2 * bit1(pixel) + 4 * bit3(pixel)  (8 rows {0-3, 8-11} + both column halves -> 8 slots)
  const int lj = lane >> 3, lf = lane & 7;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fragment addressing (lane = 16 * g + 4 * q + p)
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  for (int it = 0; it < STEM_WG_ROWS / 2; ++it) {
    const int oh0 = oh_first + 2 * it;
    const char* pbase = (const char*)x + ((int64_t)(img * Hp + 2 * oh0) * Wp) * 8;
    const char* dbase = (const char*)dy + ((int64_t)(img * Ho + oh0) * 64) * 128;
    if (it > 0) __builtin_amdgcn_s_barrier();     // every wave is done reading the previous tiles
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = wave_u + 4 * k;
      if (j < ninstr) glds16_s((unsigned)min(j * 1024 + lane * 16, patch_bytes - 16), pbase, patch + j * 1024);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = wave_u + 4 * k;               // 16 instructions: pixels 8j .. 8j + 7
      const int pix = 8 * j + lj;
      const int key = (((pix >> 1) & 1) << 1) | (((pix >> 3) & 1) << 2);
      glds16_s((unsigned)(pix * 128 + ((lf ^ key) << 4)), dbase, dtile + j * 1024);
    }
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {               // 32 pixels per MFMA k-step
      const int m = kk * 32 + 8 * g + q;           // this lane's k-row (pixel) of the first transposed read
      const int key = (((m >> 1) & 1) << 1) | (((m >> 3) & 1) << 2);   // (m + 4 has the same key)
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {                // dY^T: 16 output channels of fragment i
        const int c = i * 2 + (p >> 1);
        const char* a0 = dtile + m * 128 + ((c ^ key) << 4) + (p & 1) * 8;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a0 + 4 * 128));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        fa[i] = __builtin_bit_cast(bf16x8, v);
      }
      const int ohr = m >> 6, ow = m & 63;
#pragma unroll
      for (int j = 0; j < 4; ++j) {                // im2col^T: columns (r = 2 * wave + (j >> 1), s = 4 * (j & 1) + p, c = 0..3)
        const int r = 2 * wave_u + (j >> 1), s0 = (j & 1) * 4;
        const char* b0 = patch + (2 * ohr + r) * rowb + (2 * ow + s0 + p) * 8;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(b0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(b0 + 4 * 16));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        fb[j] = __builtin_bit_cast(bf16x8, v);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(fb[j], fa[i], acc[i][j]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float* o = slab + (int64_t)(i * 16 + (lane & 15)) * 256 + wave * 64 + j * 16 + (lane >> 4) * 4;
      *(float4*)o = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
}

// split-K count of the 3x3 patch weight gradient (0: shape not covered): its tiles are 128 x 192 (one filter row x 64 input
// channels), two thirds as many as the TN kernel's, so it plans its own splits for ~3 workgroups per CU
static int64_t wpatch_splits(int dtype, int64_t N, int64_t Hi, int64_t Wi, int64_t Ci, int64_t Co, int64_t R, int64_t S,
                             int64_t stride, int64_t pad, int64_t groups) {
  static const bool on = !(getenv("IEEE_WGRAD_PATCH") && atoi(getenv("IEEE_WGRAD_PATCH")) == 0);
  static const int64_t target = getenv("IEEE_WPATCH_TARGET") ? atoll(getenv("IEEE_WPATCH_TARGET")) : 448;
  const int64_t npix = N * Hi * Wi;
  if (!on || dtype != IEEE_BF16 || R != 3 || S != 3 || stride != 1 || pad != 1 || (Wi != 8 && Wi != 16 && Wi != 32) ||
      (Hi * Wi) % 64 != 0 || Ci % 64 != 0 || !(Co == 64 || Co % 128 == 0) || npix * Co * 2 >= (1ll << 32) ||
      npix * Ci * 2 >= (1ll << 32))
    return 0;
  const int64_t tiles = ((Co + 127) / 128) * (Ci / 64) * 3 * groups;
  const int64_t ktiles = npix / 64;
  int64_t want = (target + tiles / 2) / tiles;
  if (want > ktiles / 4) want = ktiles / 4;        // at least 4 k-tiles per split
  if (want < 1) want = 1;
  const int64_t per = (ktiles + want - 1) / want;  // k-tiles per split
  return (ktiles + per - 1) / per;
}

// split-K count of the direct stem weight gradient (0: shape not covered)
static int64_t stem_wgrad_splits(int dtype, int64_t N, int64_t Hi, int64_t Wi, int64_t Ho, int64_t Wo, int64_t Ci, int64_t Co,
                                 int64_t R, int64_t S, int64_t stride, int64_t pad) {
  static const bool on = !(getenv("IEEE_STEM_DIRECT") && atoi(getenv("IEEE_STEM_DIRECT")) == 0);
  if (!on || dtype != IEEE_BF16 || R != 8 || S != 8 || Ci != 4 || Co != 64 || stride != 2 || pad != 0 || Wo != 64 ||
      Ho % STEM_WG_ROWS != 0 || Wi != 2 * Wo + 6 || Hi != 2 * Ho + 6)
    return 0;
  return N * (Ho / STEM_WG_ROWS);
}


// ------------------------------------------------------------------ the PREVIOUS weight gradient's slab reduction as this launch's prologue
// (round 6) 53 wgrad_reduce launches of 10-30 us sat between the weight-gradient GEMMs of the side stream.  A reduction
// folded into the tail of its own GEMM launch (SlabFoldEpi above) needs one reducer per output tile -- measured +0.68 ms per
// step: a single workgroup has ~32 KB of loads in flight and reads its 0.5 MB of slabs at ~20 GB/s while the chip idles.
// Folded into the HEAD of the NEXT weight-gradient launch instead, every workgroup of that launch takes a share
// (blocks bx, bx + grid, ... of the reduction's own grid), the kernel boundary in between is the only synchronisation
// (no tickets, no fences, plain stores), the arithmetic is wgrad_reduce_body / wgrad_reduce_taps_body unchanged -- the
// same bits as the reduction launch -- and the slabs are read while they are still cache-resident.  The caller alternates
// between two slab regions and flushes the last pending reduction of a group of layers with ieee_wgrad_reduce_pending.
template <int VEC>
__device__ __forceinline__ void wgrad_reduce_body(float* part, int bx, int z, const float* __restrict__ slab, float* __restrict__ dw,
                                                  int splitk, int Co, int Ci, int RS, int64_t slab_gs, int64_t dw_gs,
                                                  int accumulate, int sl_log2);
__device__ __forceinline__ void wgrad_reduce_taps_body(float* tile, int bx, int z, const float* __restrict__ slab,
                                                       float* __restrict__ dw, int splitk, int Co, int Ci, int RS, int64_t slab_gs,
                                                       int64_t dw_gs, int accumulate);
__device__ __forceinline__ void wgrad_prologue(const ieee_wgrad_reduce_desc& d, int groups, char* smem) {
  if (d.kind == 0) return;
  float* lds = (float*)smem;                 // >= 256 * 4 floats (split-lane forms) / [128][9] floats (taps form): < 5 KB
  const int total = d.blocks * groups;
  for (int w = blockIdx.x; w < total; w += gridDim.x) {
    const int z = w / d.blocks, bx = w - z * d.blocks;
    if (d.kind == 1) wgrad_reduce_body<4>(lds, bx, z, d.slab, d.dw, d.nsplit, d.Co, d.Ci, d.RS, d.slab_gs, d.dw_gs, d.accumulate, d.sl_log2);
    else if (d.kind == 2) wgrad_reduce_body<1>(lds, bx, z, d.slab, d.dw, d.nsplit, d.Co, d.Ci, d.RS, d.slab_gs, d.dw_gs, d.accumulate, d.sl_log2);
    else wgrad_reduce_taps_body(lds, bx, z, d.slab, d.dw, d.nsplit, d.Co, d.Ci, d.RS, d.slab_gs, d.dw_gs, d.accumulate);
    __syncthreads();                         // the staging floats are reused by the next share / the GEMM's operand stage
  }
}

// ------------------------------------------------------------------ 3x3 / stride 1 weight gradient over LDS patches
// dW[co][(r, s, ci)] = sum over pixels of dY[pix][co] * X[pix + (r - 1, s - 1)][ci].  The TN kernel above treats the 9 taps as 9
// independent column blocks: X is fetched from L2 once per tap and dY once per 128 columns.  Here a workgroup owns
// 128 (64) output channels x ONE filter row r x 64 input channels = 192 GEMM columns (three taps), and per k-tile of 64
// pixels (whole image rows) brings in the dY tile [64][Co tile] (the TN core's image and loader) and the input rows those
// pixels see under row r -- RK rows x (Wm + 2) pixels x 64 channels, zero halo included -- ONCE; the three taps read their
// B fragments from it at shifted addresses through ds_read_b64_tr_b16.  Operand bytes per 128 x 192 x 64 MACs:
// 16 KB + ~10 KB (the TN kernel: 32 KB per 128 x 128 x 64).  Chunk swizzles (conflict-free for all three shifts, checked
// exhaustively): 8-wide maps key(w') = 2 * (w' & 3) with the row pitch padded by 128 B; 16 / 32-wide maps
// key(w') = 2 * bit1(w') + 4 * bit3(w').  Split-K slabs and their reduction are the TN kernel's.
template <int WLOG> struct WPatch {
  static constexpr int Wm = 1 << WLOG, RK = 64 / Wm, PARTS = Wm / 8;
  static constexpr int PITCH = (Wm + 2) * 128 + (WLOG == 3 ? 128 : 0);
  static constexpr int BYTES = RK * PITCH;
  __device__ static __forceinline__ int key(int wp) {
    if constexpr (WLOG == 3) return (wp & 3) << 1;
    else return (((wp >> 1) & 1) << 1) | (((wp >> 3) & 1) << 2);
  }
  // byte address of (k-row = pixel m of the k-tile shifted by ds columns, 16-byte channel chunk c, 8-byte half p1)
  __device__ static __forceinline__ int addr(int m, int ds, int c, int p1) {
    const int i = m >> WLOG, wp = (m & (Wm - 1)) + 1 + ds;
    return i * PITCH + wp * 128 + ((c ^ key(wp)) << 4) + p1 * 8;
  }
};

struct WgradPatchArgs {
  int Hm, Ci, Co, npix, kchunk, nsplit, groups, tiles, nchunks, ncols;
  int64_t dy_gs, x_gs, slab_gs;
};

template <int WLOG, bool HALF_M, bool FOLD = false>
__global__ __launch_bounds__(256, 3) void conv3x3_wgrad_patch_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                                     float* slab, WgradPatchArgs a, FoldArgs fa,
                                                                     ieee_wgrad_reduce_desc prev) {
  typedef WPatch<WLOG> WP;
  typedef ImgTN<bf16> Img;
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  wgrad_prologue(prev, a.groups, smem);
  // every tile of one (k-split, modality) goes to ONE XCD (blocks b, b + 8, ... share an L2): they read the same pixels
  const int bid = blockIdx.x, xcd = bid & 7, jb = bid >> 3;
  const int kzi = (jb / a.tiles) * 8 + xcd;
  int tile = jb % a.tiles;
  if (kzi >= a.nsplit * a.groups) return;
  const int r = tile % 3;
  tile /= 3;
  const int cq = tile % a.nchunks, tm = tile / a.nchunks;
  const int ks = kzi % a.nsplit, z = kzi / a.nsplit;
  const int m0 = tm * 128;
  dy += z * a.dy_gs;
  x += z * a.x_gs;
  float* slab0 = slab + z * a.slab_gs;
  slab = slab0 + (int64_t)ks * a.Co * a.ncols;
  const int kbeg = ks * a.kchunk, kend = min(a.npix, kbeg + a.kchunk);
  const int ktiles = (kend - kbeg) >> 6;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  constexpr int FM = HALF_M ? 2 : 4, WROWS = FM * 16;
  const int tile_id = (tm * a.nchunks + cq) * 3 + r;
  char* dtile = smem;                 // [64 k-rows][128 columns] bf16: the TN core's image (16 KB)
  char* patch = smem + 64 * 256;
  for (int i = t * 16; i < WP::BYTES; i += 256 * 16) *(uint4*)(patch + i) = make_uint4(0, 0, 0, 0);
  f32x4 acc[FM][6];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  LoaderColsLean lad;
  lad.init(dy, a.Co, m0, a.Co, kbeg, tn_dma_chunk(t));
  const int lj = lane >> 3, lf = lane & 7;
  const int hw = a.Hm * WP::Wm, Cs = a.Ci;
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  __syncthreads();
  for (int kt = 0; kt < ktiles; ++kt) {
    if (kt > 0) __builtin_amdgcn_s_barrier();        // every wave is done reading the previous k-tile
    glds16_lean<4>(lad.off, lad.base, dtile + (4 * wave_u) * 256);
    lad.next();
    const int pb = kbeg + kt * 64;
    const int img = pb / hw, h0 = (pb - img * hw) >> WLOG;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int jj = wave_u + 4 * k;                 // 8 instructions: RK rows x PARTS segments of 8 pixels
      const int prow = jj / WP::PARTS, part = jj % WP::PARTS;
      const int h = h0 + prow + r - 1;
      char* dst = patch + prow * WP::PITCH + (part * 8 + 1) * 128;
      if ((unsigned)h < (unsigned)a.Hm) {
        const int wp = part * 8 + 1 + lj;
        const unsigned voff = (unsigned)((lj * Cs + ((lf ^ WP::key(wp)) << 3)) * 2);
        glds16_s(voff, (const char*)x + ((int64_t)((img * a.Hm + h) * WP::Wm + part * 8) * Cs + cq * 64) * 2, dst);
      } else {
        *(uint4*)(dst + lane * 16) = make_uint4(0, 0, 0, 0);   // a row outside the image: zeros, not the previous tile's pixels
      }
    }
    wait_vmcnt<0>();
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[FM], fb[6];
#pragma unroll
      for (int i = 0; i < FM; ++i) fa[i] = Img::frag(dtile, wm * WROWS + i * 16, kk, lane);
      const int m = kk * 32 + 8 * g + q;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int ds = (j >> 1) - 1, c = (wn * 2 + (j & 1)) * 2 + (p >> 1);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(patch + WP::addr(m, ds, c, p & 1)));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(patch + WP::addr(m + 4, ds, c, p & 1)));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        fb[j] = __builtin_bit_cast(bf16x8, v);
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = mfma16(fb[j], fa[i], acc[i][j]);
    }
  }
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int co = m0 + wm * WROWS + i * 16 + (lane & 15);
    if (co >= a.Co) continue;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int col = (r * 3 + (j >> 1)) * Cs + cq * 64 + (wn * 2 + (j & 1)) * 16 + (lane >> 4) * 4;
      if constexpr (FOLD) store16_sc1(slab + (int64_t)co * a.ncols + col, acc[i][j]);
      else *(float4*)(slab + (int64_t)co * a.ncols + col) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
  }
  if constexpr (FOLD) {
    // split-K fold inside the launch (FoldArgs): a thread owns 4 input channels (chunk t & 15) of rows (t >> 4) + 16 i, and
    // for them the three taps of filter row r -- which are 3 consecutive floats per (co, ci) of the OIHW gradient
    int* flag = (int*)smem;
    const int g = ks / fa.radix;
    const int gfirst = g * fa.radix, gcount = min(fa.radix, fa.nsplit - gfirst);
    int32_t* words = fa.tickets + ((int64_t)z * fa.tiles + tile_id) * (fa.n1 + 1);
    if (!fold_arrive(words + g, gcount, flag)) return;
    const int64_t ss = (int64_t)a.Co * a.ncols;
    float* dwz = fa.dw + (int64_t)z * fa.dw_gs;
    float* l1 = fa.lvl1 + (int64_t)z * fa.lvl1_gs;
    const int ci4 = (t & 15) * 4, rr = t >> 4;
    constexpr int ROWS = HALF_M ? 64 : 128;
#pragma unroll 1
    for (int level = 0; level < 2; ++level) {
      const float* src = level == 0 ? slab0 + (int64_t)gfirst * ss : l1;
      const int count = level == 0 ? gcount : fa.n1;
      const bool final_level = level == 1 || fa.n1 == 1;
#pragma unroll 1
      for (int pass = 0; pass < ROWS / 32; ++pass) {
        int offs[6];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int sx = 0; sx < 3; ++sx)
            offs[i * 3 + sx] = (m0 + rr + 16 * (pass * 2 + i)) * a.ncols + (r * 3 + sx) * Cs + cq * 64 + ci4;
        f32x4 sum[6];
        fold_sum<6>(src, offs, ss, count, sum);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int co = m0 + rr + 16 * (pass * 2 + i);
          if (final_level) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float* d = dwz + ((int64_t)co * Cs + cq * 64 + ci4 + e) * 9 + r * 3;
              float v0 = sum[i * 3][e], v1 = sum[i * 3 + 1][e], v2 = sum[i * 3 + 2][e];
              if (fa.accumulate) { v0 += d[0]; v1 += d[1]; v2 += d[2]; }
              d[0] = v0; d[1] = v1; d[2] = v2;
            }
          } else {
#pragma unroll
            for (int sx = 0; sx < 3; ++sx) store16_sc1(l1 + (int64_t)g * ss + offs[i * 3 + sx], sum[i * 3 + sx]);
          }
        }
      }
      if (final_level) return;
      if (!fold_arrive(words + fa.n1, fa.n1, flag)) return;
    }
  }
}

struct WgradArgs {
  GatherGeom g;       // forward geometry of the conv
  int Co, ncols;      // GEMM M (= Cout), N (= R*S*Cin)
  int npix, kchunk;   // GEMM K (= N*Ho*Wo) and the K range per split
  int tiles_n, tiles, nsplit, groups, xcd_group, plain_x;
  int lean;           // every split covers whole k-tiles and the channel counts allow the clamped lean loaders
  int64_t dy_gs, x_gs, slab_gs;   // per-modality strides; slab_gs covers all splits of one modality
};

// 3 waves per SIMD (148 VGPRs, no spill) instead of the 2 the default allocation settles on: +5-10 %
#ifndef IEEE_WGRAD_OCC
#define IEEE_WGRAD_OCC 4
#endif
template <typename T, bool SLOW, int PIPE, bool HALF_M, class Epi>
__device__ __forceinline__ void conv_wgrad_body(const T* __restrict__ dy, const T* __restrict__ x, float* slab, const WgradArgs& a,
                                                const FoldArgs* fa, const ieee_wgrad_reduce_desc& prev) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  wgrad_prologue(prev, a.groups, smem);
  // XCD-aware order: every tile of one (k-split, modality) reads the same pixel range of dY and X, so all of
  // them go to ONE XCD (blocks b, b+8, ... share an XCD/L2) and that range is fetched into one L2 only.
  int kzi, tile;
  if (a.xcd_group == 1) {
    const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
    kzi = (j / a.tiles) * 8 + xcd;
    tile = j % a.tiles;
  } else if (a.xcd_group == 2) {
    // few (split, modality) groups: each XCD takes one contiguous eighth of the (group, tile) list, and inside a
    // group the tiles are walked in bands of 8 m-tiles, so the ~100 workgroups resident on one XCD form a compact
    // 2-D patch of the tile grid and share their dY / X panels in that XCD's L2
    const int total = a.tiles * a.nsplit * a.groups;
    const int bid = blockIdx.x, xcd = bid & 7, q = total >> 3, r = total & 7;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    kzi = lin / a.tiles;
    tile = lin - kzi * a.tiles;
    const int tiles_m = a.tiles / a.tiles_n, per = 8 * a.tiles_n;
    const int gi = tile / per, first = gi * 8, gsz = min(tiles_m - first, 8), in = tile - gi * per;
    tile = (first + in % gsz) * a.tiles_n + in / gsz;
  } else {
    kzi = blockIdx.x / a.tiles;
    tile = blockIdx.x % a.tiles;
  }
  if (kzi >= a.nsplit * a.groups) return;
  const int tm = tile / a.tiles_n, tn = tile % a.tiles_n;
  const int ks = kzi % a.nsplit, z = kzi / a.nsplit;
  const int m0 = tm * 128, n0 = tn * 128;
  dy += z * a.dy_gs;
  x += z * a.x_gs;
  float* slab0 = slab + z * a.slab_gs;
  slab = slab0 + (int64_t)ks * a.Co * a.ncols;
  const int kbeg = ks * a.kchunk, kend = min(a.npix, kbeg + a.kchunk);
  const int ktiles = (kend - kbeg + ImgTN<T>::BK - 1) / ImgTN<T>::BK;
  Epi epi;
  if constexpr (std::is_same<Epi, SlabFoldEpi>::value) epi = SlabFoldEpi{slab, slab0, a.ncols, a.Co, a.ncols, *fa, ks, tile, z};
  else epi = SlabEpi{slab, a.ncols, a.Co, a.ncols};
  if constexpr (PIPE > 0) {   // LDS-DMA ring (see conv_gather_kernel)
    static_assert(!SLOW && sizeof(T) == 2, "the LDS-DMA ring is the bf16 vector path");
    const int ch = tn_dma_chunk(threadIdx.x);
    if (a.lean) {   // every split is a whole number of k-tiles: no zero fill needed -> lean issue (glds16_lean) for dY
      LoaderColsLean lad;
      lad.init(dy, a.Co, m0, a.Co, kbeg, ch);
      if (a.plain_x) {
        LoaderColsLean lbd;
        lbd.init(x, a.ncols, n0, a.ncols, kbeg, ch);
        gemm_tn_dma<PIPE, HALF_M>(lad, lbd, epi, ktiles, m0, n0, smem);
      } else {
        LoaderIm2colTN<T> lbd;
        lbd.init(x, a.g, n0, kbeg, kend, ch);
        gemm_tn_dma<PIPE, HALF_M>(lad, lbd, epi, ktiles, m0, n0, smem);
      }
      return;
    }
    LoaderColsTN<T> lad;
    lad.init(dy, a.Co, m0, a.Co, kbeg, kend, ch);
    if (a.plain_x) {   // 1x1 / stride 1 / no padding: im2col(X) is X itself
      LoaderColsTN<T> lbd;
      lbd.init(x, a.ncols, n0, a.ncols, kbeg, kend, ch);
      gemm_tn_dma<PIPE, HALF_M>(lad, lbd, epi, ktiles, m0, n0, smem);
    } else {
      LoaderIm2colTN<T> lbd;
      lbd.init(x, a.g, n0, kbeg, kend, ch);
      gemm_tn_dma<PIPE, HALF_M>(lad, lbd, epi, ktiles, m0, n0, smem);
    }
    return;
  }
  LoaderColsTN<T> la;
  la.init(dy, a.Co, m0, a.Co, kbeg, kend);
  if (!SLOW && a.plain_x) {   // 1x1 / stride 1 / no padding: im2col(X) is X itself, a plain [pixels][Cin] matrix
    LoaderColsTN<T> lb;
    lb.init(x, a.ncols, n0, a.ncols, kbeg, kend);
    gemm_tn<T, (sizeof(T) == 2 ? 1 : 2)>(la, lb, epi, ktiles, m0, n0, smem);
    return;
  }
  if constexpr (SLOW) {
    LoaderIm2colSlowTN<T> lb;
    lb.init(x, a.g, n0, kbeg, kend);
    gemm_tn<T, (sizeof(T) == 2 ? 1 : 2)>(la, lb, epi, ktiles, m0, n0, smem);
  } else {
    LoaderIm2colTN<T> lb;
    lb.init(x, a.g, n0, kbeg, kend);
    gemm_tn<T, (sizeof(T) == 2 ? 1 : 2)>(la, lb, epi, ktiles, m0, n0, smem);
  }
}

template <typename T, bool SLOW, int PIPE = 0, bool HALF_M = false>
__global__ __launch_bounds__(256, (PIPE == 1 ? IEEE_WGRAD_OCC : 3)) void conv_wgrad_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                         float* __restrict__ slab, WgradArgs a, ieee_wgrad_reduce_desc prev) {
  conv_wgrad_body<T, SLOW, PIPE, HALF_M, SlabEpi>(dy, x, slab, a, nullptr, prev);
}
// the same GEMM with the split-K fold inside the launch (FoldArgs, SlabFoldEpi); bf16 LDS-DMA path only.  `slab` is not
// __restrict__ / const here: other workgroups' stores to it are read back behind the acquire.
template <bool HALF_M>
__global__ __launch_bounds__(256, IEEE_WGRAD_OCC) void conv_wgrad_fold_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                                              float* slab, WgradArgs a, FoldArgs fa,
                                                                              ieee_wgrad_reduce_desc prev) {
  conv_wgrad_body<bf16, false, 1, HALF_M, SlabFoldEpi>(dy, x, slab, a, &fa, prev);
}

// dW[z][co][ci][r][s] (OIHW fp32, the reference's parameter layout) = sum over splits of
// slab[z][ks][co][(r*S+s)*Ci + ci]; deterministic (fixed association).  The small-weight layers of layer1 / layer2 have
// 75-150 splits and only 16 k-150 k outputs: one thread per output walking all its splits is a chain of 75+ dependent
// round trips on 48 blocks (60 us for the 64->256 layer -- as long as its GEMM).  So the splits are spread over SL lanes:
// block = (256 / SL) outputs x SL lanes; lane y adds splits y, y + SL, ... in order (independent loads, 4 in flight),
// the lanes' sums are then added in lane order through LDS.  Threads walk the slab order, so the (dominant) slab
// reads are coalesced; for 3x3 / 7x7 the 4-byte writes scatter.
template <int VEC>
__device__ __forceinline__ void wgrad_reduce_body(float* part, int bx, int z, const float* __restrict__ slab, float* __restrict__ dw,
                                                  int splitk, int Co, int Ci, int RS, int64_t slab_gs, int64_t dw_gs,
                                                  int accumulate, int sl_log2) {
  const int SL = 1 << sl_log2, per = 256 >> sl_log2;
  const int t = threadIdx.x, lane = t >> (8 - sl_log2), x = t & (per - 1);
  const int64_t total = (int64_t)Co * Ci * RS;
  const int64_t i = ((int64_t)bx * per + x) * VEC;    // index in slab order [co][rs][ci]
  float acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
  if (i < total) {
    const float* s = slab + z * slab_gs + i;
#pragma unroll 4
    for (int k = lane; k < splitk; k += SL) {
      if constexpr (VEC == 4) {
        const float4 v = *(const float4*)(s + (int64_t)k * total);
        acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
      } else {
        acc[0] += s[(int64_t)k * total];
      }
    }
  }
  // (no early return: the body also runs as the prologue of the NEXT weight-gradient launch, wgrad_prologue)
  if (SL > 1) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) part[(lane * per + x) * VEC + e] = acc[e];
    __syncthreads();
    if (lane == 0) {
      for (int y = 1; y < SL; ++y) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += part[(y * per + x) * VEC + e];
      }
    }
  }
  if (lane != 0 || i >= total) return;
  if constexpr (VEC == 4) {   // 1x1 with 16-byte aligned operands: slab order == OIHW order
    float4* d = (float4*)(dw + z * dw_gs + i);
    float4 o = make_float4(acc[0], acc[1], acc[2], acc[3]);
    if (accumulate) { const float4 p = *d; o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
    *d = o;
  } else {
    int64_t o = i;
    if (RS > 1) {
      const int ci = (int)(i % Ci);
      const int64_t r = i / Ci;
      const int rs = (int)(r % RS);
      const int co = (int)(r / RS);
      o = ((int64_t)co * Ci + ci) * RS + rs;
    }
    float* d = dw + z * dw_gs + o;
    *d = accumulate ? (*d + acc[0]) : acc[0];
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                           int splitk, int Co, int Ci, int RS, int64_t slab_gs,
                                                           int64_t dw_gs, int accumulate, int sl_log2) {
  __shared__ float part[256 * VEC];
  wgrad_reduce_body<VEC>(part, blockIdx.x, blockIdx.y, slab, dw, splitk, Co, Ci, RS, slab_gs, dw_gs, accumulate, sl_log2);
}

// The same reduction for the few-split layers with taps (3x3 of layer2-4): one block per (output channel, 128 input
// channels).  The slab rows [rs][ci] are read as float4 along ci, turned through LDS, and the [ci][rs] run of the OIHW
// gradient -- contiguous for a fixed output channel -- leaves as coalesced stores (the element-wise form above scatters
// 4-byte stores RS floats apart: 70 us for the 512x512x3x3 gradients, 3x what their bytes need).  Splits are added in order.
constexpr int RT_CIB = 128;
__device__ __forceinline__ void wgrad_reduce_taps_body(float* tile /* [RT_CIB][RS] */, int bx, int z, const float* __restrict__ slab,
                                                       float* __restrict__ dw, int splitk, int Co, int Ci, int RS, int64_t slab_gs,
                                                       int64_t dw_gs, int accumulate) {
  const int cibs = (Ci + RT_CIB - 1) / RT_CIB;
  const int co = bx / cibs, ci0 = (bx % cibs) * RT_CIB;
  const int nci = min(RT_CIB, Ci - ci0);
  const int t = threadIdx.x;
  const int64_t total = (int64_t)Co * Ci * RS;
  const float* s = slab + z * slab_gs + ((int64_t)co * RS) * Ci + ci0;
  const int q = nci >> 2;   // float4 per tap row (Ci % 4 == 0)
  for (int idx = t; idx < RS * q; idx += 256) {
    const int rs = idx / q, c4 = idx - rs * q;
    const float* src = s + (int64_t)rs * Ci + c4 * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int k = 0; k < splitk; ++k) {
      const float4 v = *(const float4*)(src + (int64_t)k * total);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    float* d = tile + (c4 * 4) * RS + rs;
    d[0] = acc.x; d[RS] = acc.y; d[2 * RS] = acc.z; d[3 * RS] = acc.w;
  }
  __syncthreads();
  float* d = dw + z * dw_gs + ((int64_t)co * Ci + ci0) * RS;
  for (int j = t; j < nci * RS; j += 256) d[j] = accumulate ? (d[j] + tile[j]) : tile[j];
}

__global__ __launch_bounds__(256) void wgrad_reduce_taps_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                                int splitk, int Co, int Ci, int RS, int64_t slab_gs,
                                                                int64_t dw_gs, int accumulate) {
  extern __shared__ float tile[];   // [RT_CIB][RS]
  wgrad_reduce_taps_body(tile, blockIdx.x, blockIdx.y, slab, dw, splitk, Co, Ci, RS, slab_gs, dw_gs, accumulate);
}

// One launch reduces the slabs of MANY weight gradients (a whole backward part): blockIdx.x walks a device-side table of
// ieee_wgrad_reduce_desc (block_begin = prefix sums), blockIdx.y = modality.  53 latency-bound launches of 10-30 us on
// the low-priority stream (1.6 ms of it in the step) become 5 launches that stream their 1.6 GB of slabs.
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const ieee_wgrad_reduce_desc* __restrict__ tab, int n) {
  __shared__ float lds[RT_CIB * 9];          // >= 256 * 4 floats (the split-lane form) and [RT_CIB][9] (the taps form)
  int lo = 0, hi = n - 1;
  const int b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].block_begin <= b) lo = mid; else hi = mid - 1;
  }
  const ieee_wgrad_reduce_desc d = tab[lo];
  const int bx = b - d.block_begin, z = blockIdx.y;
  if (bx >= d.blocks) return;
  if (d.kind == 1) wgrad_reduce_body<4>(lds, bx, z, d.slab, d.dw, d.nsplit, d.Co, d.Ci, d.RS, d.slab_gs, d.dw_gs, d.accumulate, d.sl_log2);
  else if (d.kind == 2) wgrad_reduce_body<1>(lds, bx, z, d.slab, d.dw, d.nsplit, d.Co, d.Ci, d.RS, d.slab_gs, d.dw_gs, d.accumulate, d.sl_log2);
  else if (d.kind == 3) wgrad_reduce_taps_body(lds, bx, z, d.slab, d.dw, d.nsplit, d.Co, d.Ci, d.RS, d.slab_gs, d.dw_gs, d.accumulate);
}

// Weight packing from the reference's fp32 OIHW parameters:
//  mode 0 (forward): dst[co][(r*S+s)*Ci + ci]  row length ld (zero padded)
//  mode 1 (dgrad)  : dst[ci][(r*S+s)*Co + co]  row length ld
// (Ci, S) are the packed dims; the fp32 source is OIHW [Co][Ci_src][R][S_src] with Ci_src <= Ci, S_src <= S
// (zero fill beyond: the channel / column padded stem)
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ dst, int Co, int Ci, int R, int S,
                                   int ld, int mode, int64_t w_gs, int64_t dst_gs, int Ci_src, int S_src, int R_src) {
  const int z = blockIdx.y;
  const int rows = mode == 0 ? Co : Ci;
  const int64_t total = (int64_t)rows * ld;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int row = (int)(i / ld), col = (int)(i % ld);
  const int inner = mode == 0 ? Ci : Co;
  float v = 0.f;
  if (col < R * S * inner) {
    const int tap = col / inner, c = col % inner;
    const int co = mode == 0 ? row : c, ci = mode == 0 ? c : row;
    const int rr = tap / S, ss = tap % S;
    if (ci < Ci_src && ss < S_src && rr < R_src) v = w[z * w_gs + (((int64_t)co * Ci_src + ci) * R_src + rr) * S_src + ss];
  }
  dst[z * dst_gs + i] = from_f32<T>(v);
}

// dw_real[co][ci][r][s] (Ci_src x S_src) = dw_padded[co][ci][r][s] (Ci x S): drop the padded entries
__global__ void unpad_weight_grad_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int Co, int Ci, int R,
                                         int S, int Ci_src, int S_src, int R_src, int64_t dwp_gs, int64_t dw_gs,
                                         int accumulate) {
  const int z = blockIdx.y;
  const int64_t total = (int64_t)Co * Ci_src * R_src * S_src;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ss = (int)(i % S_src);
  int64_t t = i / S_src;
  const int rr = (int)(t % R_src); t /= R_src;
  const int ci = (int)(t % Ci_src);
  const int co = (int)(t / Ci_src);
  const float v = dwp[z * dwp_gs + (((int64_t)co * Ci + ci) * R + rr) * S + ss];
  float* o = dw + z * dw_gs + i;
  *o = accumulate ? *o + v : v;
}

// One launch packs every conv weight of the network: blockIdx.x walks a device-side descriptor table
// (prefix sums of 256-thread blocks), blockIdx.y = modality.
struct PackDesc {
  int64_t src_off, src_gs;   // elements, relative to the flat fp32 parameter buffer
  int64_t dst_off, dst_gs;   // elements of T, relative to the workspace base given to the kernel
  int Co, Ci, R, S, ld, mode, Ci_src, S_src;
  int block_begin;           // first blockIdx.x of this descriptor
  int pad_;                  // 0 = direct form, 1..3 = the LDS-tiled forms below, 4 = 1x1 forward (vector convert)
  int R_src, reserved_;      // rows of the fp32 source (R_src <= R: zero rows beyond)
};
template <typename T>
__global__ __launch_bounds__(256) void pack_all_kernel(const float* __restrict__ params, char* __restrict__ ws,
                                                       const PackDesc* __restrict__ descs, int ndesc) {
  int lo = 0, hi = ndesc - 1;
  const int b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].block_begin <= b) lo = mid; else hi = mid - 1;
  }
  const PackDesc d = descs[lo];
  const int z = blockIdx.y;
  __shared__ __attribute__((aligned(16))) float lds[32 * 289];   // 36 KB, shared by the three tiled forms
  const int t = threadIdx.x;
  if (d.pad_ == 1) {
    // 1x1 dgrad operand as a 64x64 LDS-tiled transpose: coalesced 256-B row reads of [co][ci], coalesced 16-byte
    // (bf16: 128-B row) writes of [ci][co]; the untiled version read with a stride of Ci floats
    float (*tile)[65] = (float (*)[65])lds;
    const int tiles_ci = d.Ci / 64;
    const int tb = b - d.block_begin, tco = tb / tiles_ci, tci = tb - tco * tiles_ci;
    const float* src = params + d.src_off + z * d.src_gs + ((int64_t)tco * 64) * d.Ci + tci * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (t >> 4) + 16 * i, c4 = (t & 15) * 4;
      const float4 v = *(const float4*)(src + (int64_t)r * d.Ci + c4);
      tile[r][c4] = v.x; tile[r][c4 + 1] = v.y; tile[r][c4 + 2] = v.z; tile[r][c4 + 3] = v.w;
    }
    __syncthreads();
    T* dst = (T*)ws + d.dst_off + z * d.dst_gs + ((int64_t)tci * 64) * d.ld + tco * 64;
    constexpr int VEC = 16 / (int)sizeof(T), CPR = 64 / VEC;   // one 16-byte store = VEC consecutive co of a ci row
#pragma unroll
    for (int idx = t; idx < 64 * CPR; idx += 256) {
      const int c = idx / CPR, r0 = (idx % CPR) * VEC;
      float f[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) f[e] = tile[r0 + e][c];
      *(uint4*)(dst + (int64_t)c * d.ld + r0) = Vec16<T>::pack(f);
    }
    return;
  }
  if (d.pad_ == 2) {
    // 3x3 forward operand dst[co][tap*Ci + ci] from OIHW [co][ci][tap]: a unit = (co, 64-channel chunk) = 576
    // contiguous floats; 4 units per block go through LDS so that both the fp32 reads and the bf16 writes (128-B
    // runs of 64 channels) are coalesced (the direct form reads with a 36-byte stride)
    const int chunks = d.Ci / 64, units = d.Co * chunks;
    const int u0 = (b - d.block_begin) * 4;
    const float* src = params + d.src_off + z * d.src_gs;
    const float* s0 = src + (int64_t)u0 * 576;
    if (((uintptr_t)s0 & 15) == 0) {
      for (int idx = t; idx < 576; idx += 256) {               // 4 units x 144 float4
        const int u = u0 + idx / 144;
        if (u < units) *(float4*)(lds + idx * 4) = *(const float4*)(s0 + idx * 4);
      }
    } else {
      for (int idx = t; idx < 4 * 576; idx += 256) {
        const int u = u0 + idx / 576;
        if (u < units) lds[idx] = s0[idx];   // unit u = (co = u / chunks, chunk = u % chunks): contiguous
      }
    }
    __syncthreads();
    T* dst = (T*)ws + d.dst_off + z * d.dst_gs;
    constexpr int VEC = 16 / (int)sizeof(T), CPR = 64 / VEC;   // 16-byte stores: VEC consecutive channels of a tap
    for (int idx = t; idx < 4 * 9 * CPR; idx += 256) {
      const int ul = idx / (9 * CPR), r = idx % (9 * CPR), tap = r / CPR, c0 = (r % CPR) * VEC;
      const int u = u0 + ul;
      if (u >= units) continue;
      const int co = u / chunks, ch = u - co * chunks;
      float f[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) f[e] = lds[ul * 576 + (c0 + e) * 9 + tap];
      *(uint4*)(dst + (int64_t)co * d.ld + tap * d.Ci + ch * 64 + c0) = Vec16<T>::pack(f);
    }
    return;
  }
  if (d.pad_ == 3) {
    // 3x3 dgrad operand dst[ci][tap*Co + co]: 32 co x 32 ci x 9 taps per block (288-float runs in, 64-B runs out)
    const int tiles_ci = d.Ci / 32;
    const int tb = b - d.block_begin, tco = tb / tiles_ci, tci = tb - tco * tiles_ci;
    const float* src = params + d.src_off + z * d.src_gs + ((int64_t)tco * 32 * d.Ci + tci * 32) * 9;
    if (((uintptr_t)src & 15) == 0) {
      for (int idx = t; idx < 32 * 72; idx += 256) {           // 32 rows x 72 float4
        const int co = idx / 72, j = (idx % 72) * 4;
        const float4 v = *(const float4*)(src + (int64_t)co * d.Ci * 9 + j);
        float* l = lds + co * 289 + j;                         // rows padded to 289 (bank spread): scalar LDS writes
        l[0] = v.x; l[1] = v.y; l[2] = v.z; l[3] = v.w;
      }
    } else {
      for (int idx = t; idx < 32 * 288; idx += 256) {
        const int co = idx / 288, j = idx % 288;
        lds[co * 289 + j] = src[(int64_t)co * d.Ci * 9 + j];   // lds[co][ci*9 + tap]
      }
    }
    __syncthreads();
    T* dst = (T*)ws + d.dst_off + z * d.dst_gs;
    constexpr int VEC = 16 / (int)sizeof(T), CPR = 32 / VEC;   // 16-byte stores: VEC consecutive co of one (ci, tap)
    for (int idx = t; idx < 288 * CPR; idx += 256) {
      const int co0 = (idx % CPR) * VEC, r = idx / CPR, tap = r % 9, ci = r / 9;
      float f[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) f[e] = lds[(co0 + e) * 289 + ci * 9 + tap];
      *(uint4*)(dst + (int64_t)(tci * 32 + ci) * d.ld + tap * d.Co + tco * 32 + co0) = Vec16<T>::pack(f);
    }
    return;
  }
  if (d.pad_ == 4) {
    // 1x1 forward operand: same layout, only the dtype changes -- 16-byte stores (one element per thread was 4x slower)
    constexpr int VEC = 16 / (int)sizeof(T);
    const unsigned total = (unsigned)d.Co * (unsigned)d.ld;
    const unsigned i = ((unsigned)(b - d.block_begin) * 256u + threadIdx.x) * VEC;
    if (i >= total) return;
    const float* src = params + d.src_off + z * d.src_gs + i;
    float f[VEC];
#pragma unroll
    for (int e = 0; e < VEC; e += 4) {
      const float4 v = *(const float4*)(src + e);
      f[e] = v.x; f[e + 1] = v.y; f[e + 2] = v.z; f[e + 3] = v.w;
    }
    *(uint4*)((T*)ws + d.dst_off + z * d.dst_gs + i) = Vec16<T>::pack(f);
    return;
  }
  const unsigned rows = d.mode == 0 ? d.Co : d.Ci;
  const unsigned total = rows * (unsigned)d.ld;                 // < 2^31 for every layer: 32-bit index math
  const unsigned i = (unsigned)(b - d.block_begin) * 256u + threadIdx.x;
  if (i >= total) return;
  const float* src = params + d.src_off + z * d.src_gs;
  T* dst = (T*)ws + d.dst_off + z * d.dst_gs;
  const unsigned RS = (unsigned)(d.R * d.S);
  if (RS == 1 && d.ld == (d.mode == 0 ? d.Ci : d.Co) && d.Ci_src == d.Ci) {
    if (d.mode == 0) {                      // 1x1 forward operand: same layout, just the dtype changes
      dst[i] = from_f32<T>(src[i]);
    } else {                                // 1x1 dgrad operand, untiled fallback: [ci][co] = transpose of [co][ci]
      const unsigned ci = i / (unsigned)d.Co, co = i - ci * (unsigned)d.Co;
      dst[i] = from_f32<T>(src[co * (unsigned)d.Ci + ci]);
    }
    return;
  }
  const unsigned row = i / (unsigned)d.ld, col = i - row * (unsigned)d.ld;
  const unsigned inner = d.mode == 0 ? d.Ci : d.Co;
  float v = 0.f;
  if (col < RS * inner) {
    const unsigned tap = col / inner, c = col - tap * inner;
    const unsigned co = d.mode == 0 ? row : c, ci = d.mode == 0 ? c : row;
    const unsigned rr = tap / (unsigned)d.S, ss = tap - rr * (unsigned)d.S;
    if (ci < (unsigned)d.Ci_src && ss < (unsigned)d.S_src && rr < (unsigned)d.R_src)
      v = src[((co * (unsigned)d.Ci_src + ci) * (unsigned)d.R_src + rr) * (unsigned)d.S_src + ss];
  }
  dst[i] = from_f32<T>(v);
}

template <typename T> static int conv_bk() { return ImgNT<T>::BK; }

// one instantiation of the gather kernel; > 64 KB of dynamic LDS needs the opt-in (160 KB per CU on gfx950)
template <typename T, int BN, bool SLOW, int MODE, int PIPE, int VAR = 0>
static void launch_gather_inst(dim3 grid, size_t smem, hipStream_t st, const T* src, const T* w, T* dst, const T* addend,
                               float* bn_partial, const ConvArgs& a, const BwdStats& bs) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv_gather_kernel<T, BN, SLOW, MODE, PIPE, VAR>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  launch_timed(conv_gather_kernel<T, BN, SLOW, MODE, PIPE, VAR>, grid, smem, st, src, w, dst, addend, bn_partial, a, bs);
}

template <typename T, int BN, int PIPE>
static void launch_gather_mode(int mode, dim3 grid, size_t smem, hipStream_t st, const T* src, const T* w, T* dst,
                               const T* addend, float* bn_partial, const ConvArgs& a, const BwdStats& bs) {
  if constexpr (PIPE == 1 && BN != 256 && sizeof(T) == 2) {   // stride-2 dgrad in parity-class row order: its own instantiations
    if (a.g.perm && (mode == 0 || mode == 2)) {
      if (mode == 2) launch_gather_inst<T, BN, false, 2, PIPE, 1>(grid, smem, st, src, w, dst, addend, bn_partial, a, bs);
      else launch_gather_inst<T, BN, false, 0, PIPE, 1>(grid, smem, st, src, w, dst, addend, nullptr, a, bs);
      return;
    }
    if (bs.addend_s2 && mode == 2) {   // compact stride-2 addend (see StagedStoreEpi VAR 2)
      launch_gather_inst<T, BN, false, 2, PIPE, 2>(grid, smem, st, src, w, dst, addend, bn_partial, a, bs);
      return;
    }
    if (bs.y2 != nullptr && mode == 2) {   // + the sums of a second BatchNorm (StagedStoreEpi VAR 3)
      launch_gather_inst<T, BN, false, 2, PIPE, 3>(grid, smem, st, src, w, dst, addend, bn_partial, a, bs);
      return;
    }
  }
  if (mode == 3) launch_gather_inst<T, BN, false, 3, PIPE>(grid, smem, st, src, w, dst, addend, nullptr, a, bs);
  else if (mode == 2) launch_gather_inst<T, BN, false, 2, PIPE>(grid, smem, st, src, w, dst, addend, bn_partial, a, bs);
  else if (mode == 1) launch_gather_inst<T, BN, false, 1, PIPE>(grid, smem, st, src, w, dst, addend, bn_partial, a, bs);
  else launch_gather_inst<T, BN, false, 0, PIPE>(grid, smem, st, src, w, dst, addend, nullptr, a, bs);
}

// Tile / pipeline choice of one gather launch (bf16 vector path).  wgs = workgroups of the 128x128 tiling.
struct GatherPlan { int bn, pipe; };
// Measured on the full training step (profiles/r01 notes, scripts/experiments/variant_scan.sh): the single-stage LDS-DMA
// pipeline (PIPE 1: 118 VGPRs, 35 KB LDS -> 4 workgroups per CU) beats register staging (3 per CU) on almost every
// layer (-7 % gather time); launches of <= 512 workgroups (Cout or Cin = 256 on the 16x8 maps) run faster still on
// 128x64 tiles (twice the workgroups).  Deeper rings (PIPE 2-4) and the 128x256 tile lose: fewer workgroups per CU.
static GatherPlan plan_gather(int M, int N, int ktiles, int groups) {
  GatherPlan p{N <= 64 ? 64 : 128, 1};
  static const int f_pipe = getenv("IEEE_GATHER_PIPE") ? atoi(getenv("IEEE_GATHER_PIPE")) : -1;
  static const int f_narrow = getenv("IEEE_GATHER_NARROW") ? atoi(getenv("IEEE_GATHER_NARROW")) : -1;
  static const int f_maxwg = getenv("IEEE_GATHER_MAXWG") ? atoi(getenv("IEEE_GATHER_MAXWG")) : 1 << 30;
  static const int f_wide = getenv("IEEE_GATHER_WIDE") ? atoi(getenv("IEEE_GATHER_WIDE")) : 0;
  const int64_t wgs = (int64_t)cdiv(M, 128) * cdiv(N, 128) * groups;
  static const int64_t f_narrow_wg = getenv("IEEE_GATHER_NARROW_WG") ? atoll(getenv("IEEE_GATHER_NARROW_WG")) : 512;
  if (N > 64 && wgs <= f_narrow_wg && f_narrow != 0) p.bn = 64;
  if (f_wide > 0 && N >= f_wide && N % 256 == 0) p.bn = 256;   // 128x256 tile: 25 % less L2 / LDS traffic per flop
  if (wgs <= f_maxwg) {   // tuning overrides apply to launches of at most IEEE_GATHER_MAXWG workgroups
    if (f_narrow == 1) p.bn = 64;
    if (f_pipe >= 0) p.pipe = f_pipe;
  }
  // IEEE_GATHER_DUAL (default 0 until measured; gemm_nt_dma, DMA_STAGES == 6): two k-tiles per round trip for the 128 x 64
  // launches that cannot offer more than 3 workgroups per CU and have a long enough K
  static const int f_dual = getenv("IEEE_GATHER_DUAL") ? atoi(getenv("IEEE_GATHER_DUAL")) : 0;
  if (f_dual > 0 && p.bn == 64 && N > 64 && p.pipe == 1 && ktiles >= f_dual && (int64_t)cdiv(M, 128) * cdiv(N, 64) * groups <= 768) p.pipe = 6;
  if (p.bn == 256 && p.pipe != 1) p.pipe = 0;
  if (ktiles < 2 && p.pipe > 1) p.pipe = 0;
  return p;
}

template <typename T>
static int launch_gather(const T* src, const T* w, T* dst, const T* addend, const GatherGeom& g, int M, int N,
                         int Ktrue, int ldw, int groups, int64_t src_gs, int64_t w_gs, int64_t dst_gs, bool slow,
                         hipStream_t st, float* bn_partial = nullptr, const BwdStats* bwd = nullptr, bool affine = false,
                         const BnFin* fin = nullptr) {
  const int BK = ImgNT<T>::BK;
  ConvArgs a;
  a.g = g;
  a.M = M;
  a.N = N;
  a.ldw = ldw;
  a.ktiles = cdiv(Ktrue, BK);
  { static const int grp = getenv("IEEE_TILE_GROUP") ? atoi(getenv("IEEE_TILE_GROUP")) : 4; a.group = grp; }
  a.src_gs = src_gs;
  a.w_gs = w_gs;
  a.dst_gs = dst_gs;
  a.tiles_m = cdiv(M, 128);
  GatherPlan plan{N <= 64 ? 64 : 128, 0};
  if (sizeof(T) == 2 && !slow) plan = plan_gather(M, N, a.ktiles, groups);
  const bool narrow = plan.bn == 64;
  if (plan.pipe == 6 && (a.g.perm || (bwd && (bwd->addend_s2 || bwd->y2)))) plan.pipe = 1;   // (forms with epilogue variants of their own)
  if (plan.pipe != 1 || plan.bn == 256 || affine || (bn_partial && !bwd)) a.g.perm = 0;   // class-major rows: default pipeline, dgrad modes
  a.tiles_n = cdiv(N, plan.bn);
  dim3 grid(a.tiles_m * a.tiles_n, groups);
  {   // phase stagger: launches of at most IEEE_GATHER_STAGGER_MAXWG workgroups (default 3072: at most 3 rounds at 4 per CU)
    static const int f_stag = getenv("IEEE_GATHER_STAGGER") ? atoi(getenv("IEEE_GATHER_STAGGER")) : 0;
    static const int64_t f_stag_wg = getenv("IEEE_GATHER_STAGGER_MAXWG") ? atoll(getenv("IEEE_GATHER_STAGGER_MAXWG")) : 3072;
    a.stagger = ((int64_t)grid.x * groups <= f_stag_wg) ? f_stag : 0;
  }
  const int stages = plan.pipe == 5 ? 1 : (plan.pipe == 6 ? 2 : (plan.pipe ? plan.pipe : (sizeof(T) == 2 ? 1 : 2)));   // PIPE 5 = one stage too, 6 = two
  size_t smem = (size_t)stages * (128 + plan.bn) * 128;
  {   // the LDS-staged epilogue needs the C tile: bf16 rows padded by 16 B, fp32 rows unpadded
    const size_t bn = plan.bn;
    const size_t epi_bytes = 128 * (bn * sizeof(T) + (sizeof(T) == 2 ? 16 : 0));
    const size_t red_bytes = 2 * (16 / sizeof(T)) * 260 * 4;   // BN-sum reduction scratch: [2*VEC] planes of 260 floats
    if (smem < epi_bytes) smem = epi_bytes;
    if (smem < red_bytes) smem = red_bytes;
  }
  if ((int64_t)M * N >= (1ll << 31)) {   // the staged epilogue keeps 32-bit element offsets
    set_error(IEEE_ERR_UNSUPPORTED, "conv: more than 2^31 output elements per modality (%ld x %ld)", (long)M, (long)N);
    return IEEE_ERR_UNSUPPORTED;
  }
  // the lean LDS-DMA loaders keep 32-bit BYTE offsets per lane: (row * ld + col) * 2 over the whole source matrix of one
  // modality (1x1 convs: [M][Cs], which can be 4x the output) and over the packed weights
  // (and the im2col loaders 32-bit element offsets into the source tensor)
  const int64_t src_elems = (int64_t)(g.npix / (g.Ho * g.Wo)) * g.Hs * g.Ws * g.Cs;
  if (sizeof(T) == 2 && !slow && (src_elems >= (1ll << 31) || (int64_t)N * ldw >= (1ll << 31))) {
    set_error(IEEE_ERR_UNSUPPORTED, "conv: a source tensor of %ld elements per modality exceeds the loaders' 32-bit offsets",
              (long)src_elems);
    return IEEE_ERR_UNSUPPORTED;
  }
  const bool stats = bn_partial != nullptr;
  if (stats && (slow || sizeof(T) != 2)) {
    set_error(IEEE_ERR_UNSUPPORTED, "conv: fused BN statistics need the bf16 vector path");
    return IEEE_ERR_UNSUPPORTED;
  }
  BwdStats bs{nullptr, nullptr, nullptr, 0, 0, 0};
  if (bwd) bs = *bwd;
  if (tl_totals != nullptr) {      // armed by ieee_conv_next_bn_totals for this launch
    if (stats && bs.y2 == nullptr && !(fin != nullptr && fin->on)) {
      bs.tot = tl_totals; bs.tot_gs = tl_totals_gs; bs.tot_rep = tl_totals_rep; bs.tot_rs = tl_totals_gs * groups;
      bs.tot_flag = tl_totals_flag != nullptr ? tl_totals_flag + (bwd ? 1 : 0) : nullptr;
      bs.tot_lim = nextafterf(TOT_RANGE / (float)cdiv(M, 128), 0.f);   // (rounded DOWN: tiles * lim <= 2^62 exactly)
    }
    tl_totals = nullptr;
    tl_totals_flag = nullptr;
  }
  if (fin != nullptr && fin->on) {
    if (!(sizeof(T) == 2 && !slow && stats && !bwd && !affine && cdiv(M, 128) <= FIN_MAX_TILES && a.g.Cs != 4)) {
      set_error(IEEE_ERR_UNSUPPORTED, "conv: the fused BatchNorm finalize needs the bf16 training form with at most %d row tiles", FIN_MAX_TILES);
      return IEEE_ERR_UNSUPPORTED;
    }
    bs.fin = *fin;
    bs.fin.M = M;
  }
  if (affine && slow) {
    set_error(IEEE_ERR_UNSUPPORTED, "conv: the fused inference BatchNorm needs the vector path");
    return IEEE_ERR_UNSUPPORTED;
  }
  const int mode = affine ? 3 : ((stats && bwd) ? 2 : (stats ? 1 : 0));
  if (bs.y2 != nullptr) {
    // the second BatchNorm's sums exist as ONE instantiation (plain-row dgrad epilogue of the single-stage bf16 pipeline):
    // force that pipeline, refuse the forms that have their own epilogue variants
    if (!(sizeof(T) == 2 && !slow && mode == 2 && !a.g.perm && !bs.addend_s2 && plan.bn != 256 && !patch_eligible(a.g, M))) {
      set_error(IEEE_ERR_UNSUPPORTED, "conv: the sums of a second BatchNorm need a plain bf16 dgrad (no stride-2 row order, no "
                                      "compact addend, no 3x3 patch form)");
      return IEEE_ERR_UNSUPPORTED;
    }
    plan.pipe = 1;
  }
  if constexpr (sizeof(T) == 2) {
    if (!slow && stem_eligible(a.g, N, ldw, mode, addend)) {
      // IEEE_STEM_WALK=4 (default 1): tiles per persistent workgroup -- the training forms only (mode 0 / 1).  Built and measured
      // in round 6 (LABNOTES R6.3c): the launch itself got SLOWER inside the step (128.7 -> 161.8 us in a trace pair: 3 072
      // four-tile chains at 40 KB of LDS instead of 12 288 short workgroups that hide each other's patch and weight latency),
      // the step moved by -0.02 ms (median of 6 interleaved rounds, inside the noise).  Kept as an option.
      static const int f_walk = getenv("IEEE_STEM_WALK") ? atoi(getenv("IEEE_STEM_WALK")) : 1;
      const bool walk = f_walk == 4 && mode != 3 && a.tiles_m % 4 == 0 && (a.g.Ho >> 1) % 4 == 0;
      dim3 sgrid(walk ? a.tiles_m / 4 : a.tiles_m, groups);
      const size_t ssm = walk ? STEM_EPI_BYTES + 2 * STEM_PATCH_BYTES : STEM_EPI_BYTES;   // staged epilogue (18 KB) > patch (11 KB) > BN-sum scratch
      if (mode == 3) launch_timed(stem_conv_kernel<3>, sgrid, ssm, st, src, w, dst, (float*)nullptr, a, bs);
      else if (mode == 1 && walk) launch_timed(stem_conv_kernel<1, 4>, sgrid, ssm, st, src, w, dst, bn_partial, a, bs);
      else if (mode == 1) launch_timed(stem_conv_kernel<1>, sgrid, ssm, st, src, w, dst, bn_partial, a, bs);
      else if (walk) launch_timed(stem_conv_kernel<0, 4>, sgrid, ssm, st, src, w, dst, (float*)nullptr, a, bs);
      else launch_timed(stem_conv_kernel<0>, sgrid, ssm, st, src, w, dst, (float*)nullptr, a, bs);
      return launch_status("stem_conv_kernel");
    }
    if (!slow && plan.bn != 256 && !bs.addend_s2 && patch_eligible(a.g, M)) {
      static const int f_style_env = getenv("IEEE_PATCH_STYLE") ? atoi(getenv("IEEE_PATCH_STYLE")) : -1;
      static const int f_bn = getenv("IEEE_PATCH_BN") ? atoi(getenv("IEEE_PATCH_BN")) : 0;
      const int bn = (N <= 64) ? 64 : (f_bn ? f_bn : plan.bn);
      // measured per layer (scripts/experiments/scan_r3g.sh): the two-stage weight ring wins where the tile is 128 x 64 (39 KB of LDS keeps
      // 4 workgroups per CU: 256->256 755 -> 923 TFLOP/s, 64->64 608 -> 650) and loses at 128 x 128 (55 KB -> 2 per CU:
      // 512->512 1 320 -> 1 234, 128->128 871 -> 767)
      const int f_style = f_style_env >= 0 ? f_style_env : (bn == 64 ? 0 : 1);
      a.tiles_n = cdiv(N, bn);
      dim3 pgrid(a.tiles_m * a.tiles_n, groups);
      const int wlog = a.g.Ws == 8 ? 3 : (a.g.Ws == 16 ? 4 : 5);
      if (bn == 64) {
        if (f_style == 0) launch_patch_w<64, 0>(wlog, mode, pgrid, st, src, w, dst, addend, bn_partial, a, bs);
        else launch_patch_w<64, 1>(wlog, mode, pgrid, st, src, w, dst, addend, bn_partial, a, bs);
      } else {
        if (f_style == 0) launch_patch_w<128, 0>(wlog, mode, pgrid, st, src, w, dst, addend, bn_partial, a, bs);
        else launch_patch_w<128, 1>(wlog, mode, pgrid, st, src, w, dst, addend, bn_partial, a, bs);
      }
      return launch_status("conv3x3_patch_kernel");
    }
  }
  if (slow) {
    if (narrow) launch_gather_inst<T, 64, true, 0, 0>(grid, smem, st, src, w, dst, addend, nullptr, a, bs);
    else launch_gather_inst<T, 128, true, 0, 0>(grid, smem, st, src, w, dst, addend, nullptr, a, bs);
  } else if constexpr (sizeof(T) == 2) {
#define IEEE_GATHER_CASE(BN_, PIPE_) \
    launch_gather_mode<T, BN_, PIPE_>(mode, grid, smem, st, src, w, dst, addend, bn_partial, a, bs)
    if (plan.bn == 256) {
      if (plan.pipe == 1) IEEE_GATHER_CASE(256, 1);
      else IEEE_GATHER_CASE(256, 0);
    } else if (narrow) {
      switch (plan.pipe) {
        case 1: IEEE_GATHER_CASE(64, 1); break;
        case 2: IEEE_GATHER_CASE(64, 2); break;
        case 3: IEEE_GATHER_CASE(64, 3); break;
        case 4: IEEE_GATHER_CASE(64, 4); break;
        case 6: IEEE_GATHER_CASE(64, 6); break;
        default: IEEE_GATHER_CASE(64, 0); break;
      }
    } else {
      switch (plan.pipe) {
        case 1: IEEE_GATHER_CASE(128, 1); break;
        case 5: IEEE_GATHER_CASE(128, 5); break;
        case 2: IEEE_GATHER_CASE(128, 2); break;
        case 3: IEEE_GATHER_CASE(128, 3); break;
        case 4: IEEE_GATHER_CASE(128, 4); break;
        default: IEEE_GATHER_CASE(128, 0); break;
      }
    }
#undef IEEE_GATHER_CASE
  } else {
    if (narrow) launch_gather_mode<T, 64, 0>(mode, grid, smem, st, src, w, dst, addend, bn_partial, a, bs);
    else launch_gather_mode<T, 128, 0>(mode, grid, smem, st, src, w, dst, addend, bn_partial, a, bs);
  }
  return launch_status("conv_gather_kernel");
}

}  // namespace ieee

using namespace ieee;

namespace {
struct Dims {
  int N, Hi, Wi, Ci, Co, R, S, stride, pad, Ho, Wo;
};
int check_dims(const char* what, int64_t N, int64_t Hi, int64_t Wi, int64_t Ci, int64_t Co, int64_t R, int64_t S,
               int64_t stride, int64_t pad, Dims* d) {
  IEEE_REQUIRE(N > 0 && Hi > 0 && Wi > 0 && Ci > 0 && Co > 0 && R > 0 && S > 0, "%s: non-positive dimension", what);
  IEEE_REQUIRE(stride == 1 || stride == 2, "%s: stride %ld unsupported (1 or 2)", what, (long)stride);
  IEEE_REQUIRE(pad >= 0 && pad <= R / 2 + 3, "%s: bad padding", what);
  d->N = (int)N; d->Hi = (int)Hi; d->Wi = (int)Wi; d->Ci = (int)Ci; d->Co = (int)Co;
  d->R = (int)R; d->S = (int)S; d->stride = (int)stride; d->pad = (int)pad;
  d->Ho = (int)((Hi + 2 * pad - R) / stride + 1);
  d->Wo = (int)((Wi + 2 * pad - S) / stride + 1);
  IEEE_REQUIRE(d->Ho > 0 && d->Wo > 0, "%s: empty output", what);
  IEEE_REQUIRE((int64_t)N * d->Ho * d->Wo < (1ll << 31) && (int64_t)N * Hi * Wi < (1ll << 31), "%s: too many pixels",
               what);
  return IEEE_OK;
}
int elem_bk(int dtype) { return dtype == IEEE_BF16 ? 64 : 32; }
int elem_vec(int dtype) { return dtype == IEEE_BF16 ? 8 : 4; }
}  // namespace

extern "C" int64_t ieee_conv_packed_ld(int dtype, int64_t inner_channels, int64_t R, int64_t S) {
  const int64_t K = R * S * inner_channels, bk = elem_bk(dtype);
  return (K + bk - 1) / bk * bk;
}

extern "C" int ieee_pack_conv_weight_padded(const float* w_oihw, void* dst, int dtype, int mode, int64_t groups,
                                            int64_t Co, int64_t Ci_src, int64_t R_src, int64_t S_src, int64_t Ci,
                                            int64_t R, int64_t S, int64_t w_gs, int64_t dst_gs, void* stream) {
  IEEE_REQUIRE(w_oihw && dst, "pack_conv_weight_padded: null pointer");
  IEEE_REQUIRE(mode == 0, "pack_conv_weight_padded: forward packing only (the padded stem has no dgrad)");
  IEEE_REQUIRE(dtype == IEEE_F32 || dtype == IEEE_BF16, "pack_conv_weight_padded: bad dtype");
  IEEE_REQUIRE(Ci_src <= Ci && S_src <= S && R_src <= R, "pack_conv_weight_padded: source larger than destination");
  const int64_t ld = ieee_conv_packed_ld(dtype, Ci, R, S);
  dim3 grid(cdiv(Co * ld, 256), (unsigned)groups);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == IEEE_F32)
    pack_weight_kernel<float><<<grid, 256, 0, st>>>(w_oihw, (float*)dst, (int)Co, (int)Ci, (int)R, (int)S, (int)ld, 0,
                                                    w_gs, dst_gs, (int)Ci_src, (int)S_src, (int)R_src);
  else
    pack_weight_kernel<bf16><<<grid, 256, 0, st>>>(w_oihw, (bf16*)dst, (int)Co, (int)Ci, (int)R, (int)S, (int)ld, 0,
                                                   w_gs, dst_gs, (int)Ci_src, (int)S_src, (int)R_src);
  return launch_status("pack_weight_kernel");
}

extern "C" int64_t ieee_pack_desc_bytes(void) { return (int64_t)sizeof(ieee::PackDesc); }

/* descs: device array of `ndesc` descriptors (layout: struct PackDesc above, filled by the executor);
 * total_blocks = sum of 256-thread blocks over all descriptors */
extern "C" int ieee_pack_all_weights(const float* params, void* ws_base, const void* descs, int64_t ndesc,
                                     int64_t total_blocks, int dtype, void* stream) {
  IEEE_REQUIRE(params && ws_base && descs && ndesc > 0 && total_blocks > 0, "pack_all_weights: bad arguments");
  dim3 grid((unsigned)total_blocks, 3);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == IEEE_F32)
    pack_all_kernel<float><<<grid, 256, 0, st>>>(params, (char*)ws_base, (const PackDesc*)descs, (int)ndesc);
  else if (dtype == IEEE_BF16)
    pack_all_kernel<bf16><<<grid, 256, 0, st>>>(params, (char*)ws_base, (const PackDesc*)descs, (int)ndesc);
  else
    IEEE_REQUIRE(false, "pack_all_weights: bad dtype");
  return launch_status("pack_all_kernel");
}

extern "C" int ieee_unpad_weight_grad(const float* dw_padded, float* dw, int64_t groups, int64_t Co, int64_t Ci,
                                      int64_t R, int64_t S, int64_t Ci_src, int64_t R_src, int64_t S_src,
                                      int64_t dwp_gs, int64_t dw_gs, int accumulate, void* stream) {
  IEEE_REQUIRE(dw_padded && dw, "unpad_weight_grad: null pointer");
  IEEE_REQUIRE(Ci_src <= Ci && S_src <= S && R_src <= R, "unpad_weight_grad: source larger than the padded tensor");
  dim3 grid(cdiv(Co * Ci_src * R_src * S_src, 256), (unsigned)groups);
  unpad_weight_grad_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(dw_padded, dw, (int)Co, (int)Ci, (int)R, (int)S,
                                                                  (int)Ci_src, (int)S_src, (int)R_src, dwp_gs, dw_gs,
                                                                  accumulate);
  return launch_status("unpad_weight_grad_kernel");
}

extern "C" int ieee_pack_conv_weight(const float* w_oihw, void* dst, int dtype, int mode, int64_t groups, int64_t Co,
                                     int64_t Ci, int64_t R, int64_t S, int64_t w_gs, int64_t dst_gs, void* stream) {
  IEEE_REQUIRE(w_oihw && dst, "pack_conv_weight: null pointer");
  IEEE_REQUIRE(mode == 0 || mode == 1, "pack_conv_weight: mode must be 0 (forward) or 1 (dgrad)");
  IEEE_REQUIRE(dtype == IEEE_F32 || dtype == IEEE_BF16, "pack_conv_weight: bad dtype");
  const int64_t ld = ieee_conv_packed_ld(dtype, mode == 0 ? Ci : Co, R, S);
  const int64_t rows = mode == 0 ? Co : Ci;
  dim3 grid(cdiv(rows * ld, 256), (unsigned)groups);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == IEEE_F32)
    pack_weight_kernel<float><<<grid, 256, 0, st>>>(w_oihw, (float*)dst, (int)Co, (int)Ci, (int)R, (int)S, (int)ld,
                                                    mode, w_gs, dst_gs, (int)Ci, (int)S, (int)R);
  else
    pack_weight_kernel<bf16><<<grid, 256, 0, st>>>(w_oihw, (bf16*)dst, (int)Co, (int)Ci, (int)R, (int)S, (int)ld, mode,
                                                   w_gs, dst_gs, (int)Ci, (int)S, (int)R);
  return launch_status("pack_weight_kernel");
}

extern "C" int ieee_conv2d_fwd(const void* x, const void* w_packed, void* y, int dtype, int64_t groups, int64_t N,
                               int64_t Hi, int64_t Wi, int64_t Ci, int64_t Co, int64_t R, int64_t S, int64_t stride,
                               int64_t pad, int64_t x_gs, int64_t w_gs, int64_t y_gs, float* bn_partial, void* stream) {
  ieee::OneShotArms disarm;
  IEEE_REQUIRE(x && w_packed && y, "conv2d_fwd: null pointer");
  Dims d;
  IEEE_TRY(check_dims("conv2d_fwd", N, Hi, Wi, Ci, Co, R, S, stride, pad, &d));
  IEEE_REQUIRE(Co % 4 == 0, "conv2d_fwd: Cout %ld must be a multiple of 4", (long)Co);
  GatherGeom g{d.Hi, d.Wi, d.Ci, d.Ho, d.Wo, d.R, d.S, d.stride, -d.pad, +1, 1, d.N * d.Ho * d.Wo};
  const int bk = elem_bk(dtype);
  const int vec = elem_vec(dtype);
  // vector path: whole k-tiles per tap; or several whole taps per k-tile inside one filter row; or (Ci < one 16-byte
  // chunk, no padding) chunks of vec/Ci horizontally adjacent pixels and k-tiles of whole filter rows -- see
  // LoaderIm2colNT
  const bool fast = (Ci % bk == 0) || (Ci < bk && Ci % vec == 0 && bk % Ci == 0 && S % (bk / Ci) == 0) ||
                    (Ci < vec && vec % Ci == 0 && pad == 0 && (bk / Ci) % S == 0 && R % ((bk / Ci) / S) == 0 &&
                     S % (vec / Ci) == 0 && Wi % (vec / Ci) == 0 && stride % (vec / Ci) == 0);
  const bool slow = !fast;
  const int ldw = (int)ieee_conv_packed_ld(dtype, Ci, R, S);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == IEEE_F32)
    return launch_gather<float>((const float*)x, (const float*)w_packed, (float*)y, nullptr, g, g.npix, d.Co,
                                d.R * d.S * d.Ci, ldw, (int)groups, x_gs, w_gs, y_gs, slow, st, bn_partial);
  if (dtype == IEEE_BF16)
    return launch_gather<bf16>((const bf16*)x, (const bf16*)w_packed, (bf16*)y, nullptr, g, g.npix, d.Co,
                               d.R * d.S * d.Ci, ldw, (int)groups, x_gs, w_gs, y_gs, slow, st, bn_partial);
  IEEE_REQUIRE(false, "conv2d_fwd: bad dtype %d", dtype);
}

extern "C" int64_t ieee_conv2d_fwd_bn_train_max_rows(void) { return (int64_t)FIN_MAX_TILES * 128; }

extern "C" int ieee_conv2d_fwd_bn_train(const void* x, const void* w_packed, void* y, int dtype, int64_t groups, int64_t N,
                                        int64_t Hi, int64_t Wi, int64_t Ci, int64_t Co, int64_t R, int64_t S, int64_t stride,
                                        int64_t pad, int64_t x_gs, int64_t w_gs, int64_t y_gs, float* bn_partial,
                                        const float* gamma, const float* beta, int64_t param_gs, float* running_mean,
                                        float* running_var, int64_t buf_gs, float* stats, float momentum, float eps,
                                        int32_t* tickets, void* stream) {
  ieee::OneShotArms disarm;
  IEEE_REQUIRE(x && w_packed && y && bn_partial && gamma && beta && stats && tickets, "conv2d_fwd_bn_train: null pointer");
  IEEE_REQUIRE(dtype == IEEE_BF16, "conv2d_fwd_bn_train: bf16 only");
  Dims d;
  IEEE_TRY(check_dims("conv2d_fwd_bn_train", N, Hi, Wi, Ci, Co, R, S, stride, pad, &d));
  IEEE_REQUIRE(Co % 8 == 0 && Ci % 64 == 0, "conv2d_fwd_bn_train: Cin %% 64, Cout %% 8");
  IEEE_REQUIRE((int64_t)d.N * d.Ho * d.Wo <= ieee_conv2d_fwd_bn_train_max_rows(), "conv2d_fwd_bn_train: more than %ld output pixels",
               (long)ieee_conv2d_fwd_bn_train_max_rows());
  GatherGeom g{d.Hi, d.Wi, d.Ci, d.Ho, d.Wo, d.R, d.S, d.stride, -d.pad, +1, 1, d.N * d.Ho * d.Wo};
  const int ldw = (int)ieee_conv_packed_ld(dtype, Ci, R, S);
  BnFin fin;
  fin.gamma = gamma; fin.beta = beta; fin.rm = running_mean; fin.rv = running_var; fin.stats = stats; fin.counter = (int*)tickets;
  fin.param_gs = param_gs; fin.buf_gs = buf_gs; fin.momentum = momentum; fin.eps = eps; fin.on = 1;
  return launch_gather<bf16>((const bf16*)x, (const bf16*)w_packed, (bf16*)y, nullptr, g, g.npix, d.Co, d.R * d.S * d.Ci, ldw,
                             (int)groups, x_gs, w_gs, y_gs, false, (hipStream_t)stream, bn_partial, nullptr, false, &fin);
}

extern "C" int ieee_conv2d_fwd_bn_eval(const void* x, const void* w_packed, void* out, const void* residual,
                                       const float* bn_stats, int relu, int dtype, int64_t groups, int64_t N, int64_t Hi,
                                       int64_t Wi, int64_t Ci, int64_t Co, int64_t R, int64_t S, int64_t stride,
                                       int64_t pad, int64_t x_gs, int64_t w_gs, int64_t out_gs, void* stream) {
  ieee::OneShotArms disarm;
  IEEE_REQUIRE(x && w_packed && out && bn_stats, "conv2d_fwd_bn_eval: null pointer");
  Dims d;
  IEEE_TRY(check_dims("conv2d_fwd_bn_eval", N, Hi, Wi, Ci, Co, R, S, stride, pad, &d));
  IEEE_REQUIRE(Co % elem_vec(dtype) == 0, "conv2d_fwd_bn_eval: Cout %ld must be a multiple of %d", (long)Co, elem_vec(dtype));
  GatherGeom g{d.Hi, d.Wi, d.Ci, d.Ho, d.Wo, d.R, d.S, d.stride, -d.pad, +1, 1, d.N * d.Ho * d.Wo};
  const int bk = elem_bk(dtype), vec = elem_vec(dtype);
  const bool fast = (Ci % bk == 0) || (Ci < bk && Ci % vec == 0 && bk % Ci == 0 && S % (bk / Ci) == 0) ||
                    (Ci < vec && vec % Ci == 0 && pad == 0 && (bk / Ci) % S == 0 && R % ((bk / Ci) / S) == 0 &&
                     S % (vec / Ci) == 0 && Wi % (vec / Ci) == 0 && stride % (vec / Ci) == 0);
  const int ldw = (int)ieee_conv_packed_ld(dtype, Ci, R, S);
  hipStream_t st = (hipStream_t)stream;
  BwdStats bs{nullptr, nullptr, bn_stats, out_gs, 4 * Co, relu};
  if (dtype == IEEE_F32)
    return launch_gather<float>((const float*)x, (const float*)w_packed, (float*)out, (const float*)residual, g, g.npix,
                                d.Co, d.R * d.S * d.Ci, ldw, (int)groups, x_gs, w_gs, out_gs, !fast, st, nullptr, &bs, true);
  if (dtype == IEEE_BF16)
    return launch_gather<bf16>((const bf16*)x, (const bf16*)w_packed, (bf16*)out, (const bf16*)residual, g, g.npix, d.Co,
                               d.R * d.S * d.Ci, ldw, (int)groups, x_gs, w_gs, out_gs, !fast, st, nullptr, &bs, true);
  IEEE_REQUIRE(false, "conv2d_fwd_bn_eval: bad dtype %d", dtype);
}

extern "C" int ieee_conv2d_dgrad2(const void* dy, const void* w_packed_d, void* dx, const void* addend, int dtype,
                                  int64_t groups, int64_t N, int64_t Hi, int64_t Wi, int64_t Ci, int64_t Co, int64_t R,
                                  int64_t S, int64_t stride, int64_t pad, int64_t dy_gs, int64_t w_gs, int64_t dx_gs,
                                  float* bn_partial, const void* bn_y, const void* bn_mask, const float* bn_stats,
                                  int bn_mask_bits, int addend_stride, const void* bn_y2, float* bn_partial2, void* stream);

extern "C" int ieee_conv2d_dgrad(const void* dy, const void* w_packed_d, void* dx, const void* addend, int dtype,
                                 int64_t groups, int64_t N, int64_t Hi, int64_t Wi, int64_t Ci, int64_t Co, int64_t R,
                                 int64_t S, int64_t stride, int64_t pad, int64_t dy_gs, int64_t w_gs, int64_t dx_gs,
                                 float* bn_partial, const void* bn_y, const void* bn_mask, const float* bn_stats,
                                 int bn_mask_bits, int addend_stride, void* stream) {
  return ieee_conv2d_dgrad2(dy, w_packed_d, dx, addend, dtype, groups, N, Hi, Wi, Ci, Co, R, S, stride, pad, dy_gs, w_gs, dx_gs,
                            bn_partial, bn_y, bn_mask, bn_stats, bn_mask_bits, addend_stride, nullptr, nullptr, stream);
}

extern "C" int ieee_conv2d_dgrad2(const void* dy, const void* w_packed_d, void* dx, const void* addend, int dtype,
                                  int64_t groups, int64_t N, int64_t Hi, int64_t Wi, int64_t Ci, int64_t Co, int64_t R,
                                  int64_t S, int64_t stride, int64_t pad, int64_t dy_gs, int64_t w_gs, int64_t dx_gs,
                                  float* bn_partial, const void* bn_y, const void* bn_mask, const float* bn_stats,
                                  int bn_mask_bits, int addend_stride, const void* bn_y2, float* bn_partial2, void* stream) {
  ieee::OneShotArms disarm;
  IEEE_REQUIRE(dy && w_packed_d && dx, "conv2d_dgrad: null pointer");
  IEEE_REQUIRE((bn_y2 == nullptr) == (bn_partial2 == nullptr) && (!bn_y2 || (bn_partial && dtype == IEEE_BF16)),
               "conv2d_dgrad: the second BatchNorm's sums need its input AND its partial block, in the bf16 fused form");
  IEEE_REQUIRE(addend_stride == 1 || (addend_stride == 2 && addend && bn_partial && dtype == IEEE_BF16 && stride == 1 &&
                                      Hi > 1 && Wi > 1 && !(Hi & (Hi - 1)) && !(Wi & (Wi - 1))),
               "conv2d_dgrad: a stride-2 addend needs the bf16 fused form of a stride-1 conv over a power-of-two map");
  IEEE_REQUIRE(!bn_mask_bits || (bn_mask && dx_gs % 8 == 0 && Ci % 8 == 0), "conv2d_dgrad: bit mask needs Cin %% 8 == 0");
  IEEE_REQUIRE(!bn_partial || bn_y, "conv2d_dgrad: fused BN-backward sums need the BN input tensor");
  Dims d;
  IEEE_TRY(check_dims("conv2d_dgrad", N, Hi, Wi, Ci, Co, R, S, stride, pad, &d));
  IEEE_REQUIRE(Co % elem_bk(dtype) == 0, "conv2d_dgrad: Cout %ld must be a multiple of %d", (long)Co, elem_bk(dtype));
  IEEE_REQUIRE(Ci % 4 == 0, "conv2d_dgrad: Cin %ld must be a multiple of 4", (long)Ci);
  // rows = input pixels; source = dY [N,Ho,Wo,Co]; h_dy = (hi + pad - r) / stride
  GatherGeom g{d.Ho, d.Wo, d.Co, d.Hi, d.Wi, d.R, d.S, 1, d.pad, -1, d.stride, d.N * d.Hi * d.Wi};
  const int ldw = (int)ieee_conv_packed_ld(dtype, Co, R, S);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == IEEE_F32)
    return launch_gather<float>((const float*)dy, (const float*)w_packed_d, (float*)dx, (const float*)addend, g,
                                g.npix, d.Ci, d.R * d.S * d.Co, ldw, (int)groups, dy_gs, w_gs, dx_gs, false, st);
  if (dtype == IEEE_BF16) {
    // stride 2: rows in parity-class-major order so that a workgroup skips the taps that cannot reach its pixels
    static const bool f_perm = !(getenv("IEEE_DGRAD_PERM") && atoi(getenv("IEEE_DGRAD_PERM")) == 0);
    // (3x3 only: for a 1x1 the skipped work is outweighed by the stride-2 scatter of the stores -- 140 -> 167 us alone)
    if (f_perm && d.stride == 2 && d.R * d.S > 1 && !(d.Hi & 1) && !(d.Wi & 1) && ((d.Hi >> 1) * (d.Wi >> 1)) % 128 == 0 && 128 % (d.Wi >> 1) == 0 && d.Co % 64 == 0 && (int64_t)g.npix * d.Ci < (1ll << 31) &&
        d.R * d.S <= 56)
      g.perm = 1;
    BwdStats bs{bn_y, bn_mask, bn_stats, dx_gs, 4 * Ci, 0, bn_mask_bits, addend_stride == 2 ? 1 : 0};
    bs.y2 = bn_y2;
    bs.partial2 = bn_partial2;
    return launch_gather<bf16>((const bf16*)dy, (const bf16*)w_packed_d, (bf16*)dx, (const bf16*)addend, g, g.npix,
                               d.Ci, d.R * d.S * d.Co, ldw, (int)groups, dy_gs, w_gs, dx_gs, false, st, bn_partial,
                               bn_partial ? &bs : nullptr);
  }
  IEEE_REQUIRE(false, "conv2d_dgrad: bad dtype %d", dtype);
}

// split-K heuristic shared by the workspace query and the launch
static int wgrad_splitk(int64_t npix, int64_t Co, int64_t ncols, int64_t groups, int dtype, int64_t taps) {
  const int64_t tiles = ((Co + 127) / 128) * ((ncols + 127) / 128) * groups;
  const int64_t bk = elem_bk(dtype);
  // workgroups to aim at: 448 measured best at 4 workgroups/CU (256: -2.5 %, 640: -0.6 %, 1024: -3.5 %: slab
  // traffic).  A larger target for the small-weight layers (stem, layer1; IEEE_WGRAD_TARGET_SMALL) was measured too:
  // 1024 helps the 3x3 / stem launches by 5-8 % but costs the 1x1 ones 30 %, net -1 %
  static const int64_t f_target = getenv("IEEE_WGRAD_TARGET") ? atoll(getenv("IEEE_WGRAD_TARGET")) : 0;
  static const int64_t f_small = getenv("IEEE_WGRAD_TARGET_SMALL") ? atoll(getenv("IEEE_WGRAD_TARGET_SMALL")) : 448;
  const int64_t per_split_bytes = groups * Co * ncols * 4;
  // round 2 (parallel slab reduce): the small-weight layers WITH taps (layer1 3x3, the stem: X is re-read per tap from
  // L2, few output tiles) gain 20-25 % from twice the workgroups; the 1x1 ones still lose (scripts/experiments/scan_wt.sh)
  static const int64_t f_small_taps = getenv("IEEE_WGRAD_TARGET_SMALL_TAPS") ? atoll(getenv("IEEE_WGRAD_TARGET_SMALL_TAPS")) : 896;
  const int64_t small = taps > 1 ? f_small_taps : f_small;
  static const int64_t f_big = getenv("IEEE_WGRAD_TARGET_BIG") ? atoll(getenv("IEEE_WGRAD_TARGET_BIG")) : 448;
  const int64_t target = f_target ? f_target : (per_split_bytes <= (1 << 20) ? small : f_big);
  int64_t want = (target + tiles - 1) / tiles;               // aim at ~`target` workgroups per launch
  const int64_t maxsplit = (npix + 4 * bk - 1) / (4 * bk);   // at least 4 k-tiles per split
  if (want > maxsplit) want = maxsplit;
  if (want < 1) want = 1;
  if (want > 1024) want = 1024;
  return (int)want;
}

// In-launch fold of the split-K slabs (FoldArgs): level-0 groups of `radix` splits, n1 of them; one level (n1 = 1) up to
// 16 splits, else radix ~ sqrt(nsplit) so that no reducer adds more than ~2 sqrt(nsplit) tiles in a row.
constexpr int FOLD_MAX_SPLITS = 256;
constexpr int64_t FOLD_TICKET_WORDS = 8192;
static void fold_plan(int nsplit, int* radix, int* n1) {
  static const int f_one = getenv("IEEE_WGRAD_FOLD_ONE") ? atoi(getenv("IEEE_WGRAD_FOLD_ONE")) : 16;
  if (nsplit <= f_one) { *radix = nsplit; *n1 = 1; return; }
  const int r = (int)ceil(sqrt((double)nsplit));
  const int groups = cdiv(nsplit, r);
  *radix = cdiv(nsplit, groups);
  *n1 = cdiv(nsplit, *radix);
}
// floats behind the level-0 slabs of `nsplit_max` splits that cover the level-1 slabs of ANY actual split count up to it
// (the launch may end up with fewer splits than the heuristic asked for)
static int64_t fold_lvl1_floats(int64_t nsplit_max, int64_t groups, int64_t Co, int64_t ncols) {
  int64_t extra = 0;
  for (int s = 2; s <= nsplit_max && s <= FOLD_MAX_SPLITS; ++s) {
    int radix, n1;
    fold_plan(s, &radix, &n1);
    if (n1 > 1) extra = std::max<int64_t>(extra, s + n1 - nsplit_max);
  }
  return extra > 0 ? extra * groups * Co * ncols : 0;
}

/* floats of BN partial sums per group that ieee_conv2d_fwd emits when bn_partial != NULL, and the row-block
 * count to hand to ieee_bn2d_fwd(stats_rblocks) */
extern "C" int64_t ieee_conv2d_fwd_stats_rblocks(int64_t N, int64_t Ho, int64_t Wo) { return (N * Ho * Wo + 127) / 128; }

extern "C" int64_t ieee_conv2d_wgrad_workspace_bytes(int dtype, int64_t groups, int64_t N, int64_t Ho, int64_t Wo,
                                                     int64_t Ci, int64_t Co, int64_t R, int64_t S) {
  const int64_t npix = N * Ho * Wo, ncols = R * S * Ci;
  const int64_t gsplit = wgrad_splitk(npix, Co, ncols, groups, dtype, R * S);
  const int64_t generic = gsplit * groups * Co * ncols * 4 + fold_lvl1_floats(gsplit, groups, Co, ncols) * 4;
  // (the direct stem form: stride 2, no padding -> Hi = 2 Ho + 6; the query has no Hi / stride, so both sizes are covered)
  const int64_t stem = stem_wgrad_splits(dtype, N, 2 * Ho + 6, 2 * Wo + 6, Ho, Wo, Ci, Co, R, S, 2, 0) * groups * Co * ncols * 4;
  // (3x3 / stride 1 / pad 1: Hi = Ho, Wi = Wo)
  const int64_t psplit = wpatch_splits(dtype, N, Ho, Wo, Ci, Co, R, S, 1, 1, groups);
  const int64_t wp = psplit * groups * Co * ncols * 4 + fold_lvl1_floats(psplit, groups, Co, ncols) * 4;
  return std::max(generic, std::max(stem, wp));
}

static void fill_reduce_desc(ieee_wgrad_reduce_desc* o, const float* slab, float* dw, int64_t slab_gs, int64_t dw_gs, int nsplit,
                             int Co, int Ci, int RS, int kind, int sl_log2, int blocks, int accumulate) {
  o->slab = slab; o->dw = dw; o->slab_gs = slab_gs; o->dw_gs = dw_gs; o->nsplit = nsplit; o->Co = Co; o->Ci = Ci; o->RS = RS;
  o->kind = kind; o->sl_log2 = sl_log2; o->blocks = blocks; o->accumulate = accumulate; o->block_begin = 0; o->reserved_ = 0;
}

// one described reduction as a launch of its own (the immediate path's kernels and grids)
static int reduce_launch(const ieee_wgrad_reduce_desc& d, int64_t groups, hipStream_t st) {
  if (d.kind == 0) return IEEE_OK;
  dim3 grid((unsigned)d.blocks, (unsigned)groups);
  if (d.kind == 3) {
    wgrad_reduce_taps_kernel<<<grid, 256, (size_t)RT_CIB * d.RS * sizeof(float), st>>>(d.slab, d.dw, d.nsplit, d.Co, d.Ci, d.RS, d.slab_gs,
                                                                                      d.dw_gs, d.accumulate);
    return launch_status("wgrad_reduce_taps_kernel");
  }
  if (d.kind == 1)
    wgrad_reduce_kernel<4><<<grid, 256, 0, st>>>(d.slab, d.dw, d.nsplit, d.Co, d.Ci, d.RS, d.slab_gs, d.dw_gs, d.accumulate, d.sl_log2);
  else
    wgrad_reduce_kernel<1><<<grid, 256, 0, st>>>(d.slab, d.dw, d.nsplit, d.Co, d.Ci, d.RS, d.slab_gs, d.dw_gs, d.accumulate, d.sl_log2);
  return launch_status("wgrad_reduce_kernel");
}

// defer != NULL: run the GEMM only and describe the slab reduction in *defer (kind 0: nothing left to do)
static int wgrad_impl(const void* dy, const void* x, float* dw_oihw, void* work, int dtype, int64_t groups,
                      int64_t N, int64_t Hi, int64_t Wi, int64_t Ci, int64_t Co, int64_t R, int64_t S,
                      int64_t stride, int64_t pad, int64_t dy_gs, int64_t x_gs, int64_t dw_gs,
                      int accumulate, void* stream, ieee_wgrad_reduce_desc* defer, int32_t* tickets = nullptr,
                      const ieee_wgrad_reduce_desc* prev = nullptr) {
  IEEE_REQUIRE(dy && x && dw_oihw && work, "conv2d_wgrad: null pointer");
  Dims d;
  IEEE_TRY(check_dims("conv2d_wgrad", N, Hi, Wi, Ci, Co, R, S, stride, pad, &d));
  IEEE_REQUIRE(Co % elem_vec(dtype) == 0, "conv2d_wgrad: Cout must be a multiple of %d", elem_vec(dtype));
  // the reduction this launch carries as its prologue (wgrad_prologue); kind 0 = none
  ieee_wgrad_reduce_desc pv;
  fill_reduce_desc(&pv, nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
  if (prev != nullptr && prev->kind != 0) pv = *prev;
  if (const int64_t ssplits = stem_wgrad_splits(dtype, N, Hi, Wi, d.Ho, d.Wo, Ci, Co, R, S, stride, pad)) {
    // the stem: direct form over LDS patches (stem_wgrad_kernel), one slab per STEM_WG_ROWS output rows of an image
    hipStream_t st = (hipStream_t)stream;
    if (pv.kind != 0) IEEE_TRY(reduce_launch(pv, groups, st));   // (this kernel has no prologue: the pending reduction runs first)
    float* slab = (float*)work;
    const int64_t slab_gs = ssplits * 64 * 256;
    stem_wgrad_kernel<<<dim3((unsigned)ssplits, (unsigned)groups), 256, 12 * 1024 + 16 * 1024, st>>>(
        (const bf16*)dy, (const bf16*)x, slab, d.Hi, d.Wi, d.Ho, dy_gs, x_gs, slab_gs);
    IEEE_TRY(launch_status("stem_wgrad_kernel"));
    const int64_t total = (int64_t)d.Co * d.Ci * d.R * d.S;
    int sl_log2 = 0;
    while (sl_log2 < 4 && (2 << sl_log2) <= ssplits && (total << sl_log2) * groups < (int64_t)256 * 2048) ++sl_log2;
    dim3 rgrid(cdiv(total, 256 >> sl_log2), (unsigned)groups);
    if (defer) {
      fill_reduce_desc(defer, slab, dw_oihw, slab_gs, dw_gs, (int)ssplits, d.Co, d.Ci, d.R * d.S, 2, sl_log2, (int)rgrid.x, accumulate);
      return IEEE_OK;
    }
    wgrad_reduce_kernel<1><<<rgrid, 256, 0, st>>>(slab, dw_oihw, (int)ssplits, d.Co, d.Ci, d.R * d.S, slab_gs, dw_gs, accumulate, sl_log2);
    return launch_status("wgrad_reduce_kernel");
  }
  WgradArgs a;
  a.g = GatherGeom{d.Hi, d.Wi, d.Ci, d.Ho, d.Wo, d.R, d.S, d.stride, -d.pad, +1, 1, d.N * d.Ho * d.Wo};
  a.Co = d.Co;
  a.ncols = d.R * d.S * d.Ci;
  a.npix = a.g.npix;
  const int splitk = wgrad_splitk(a.npix, d.Co, a.ncols, groups, dtype, d.R * d.S);
  const int bk = elem_bk(dtype);
  a.kchunk = cdiv(cdiv(a.npix, splitk), bk) * bk;
  const int nsplit = cdiv(a.npix, a.kchunk);
  a.tiles_n = cdiv(a.ncols, 128);
  a.tiles = cdiv(d.Co, 128) * a.tiles_n;
  a.nsplit = nsplit;
  a.groups = (int)groups;
  a.dy_gs = dy_gs;
  a.x_gs = x_gs;
  a.slab_gs = (int64_t)nsplit * d.Co * a.ncols;
  const int vec = elem_vec(dtype);
  // a 16-byte chunk of im2col(X) is vec consecutive channels of one tap, or (Ci < vec, no padding) vec/Ci adjacent pixels
  const bool chunk_ok = Ci % vec == 0 || (vec % Ci == 0 && pad == 0 && S % (vec / Ci) == 0 && Wi % (vec / Ci) == 0 &&
                                          stride % (vec / Ci) == 0);
  const bool slow = !chunk_ok;
  a.plain_x = (d.R == 1 && d.S == 1 && d.stride == 1 && d.pad == 0) ? 1 : 0;
  // (the lean loaders keep 32-bit byte offsets into dY [npix][Co] and, for the 1x1 form, X [npix][ncols]; larger
  // operands take the pointer-form loaders)
  a.lean = (a.npix % bk == 0 && d.Co >= 8 && a.ncols >= 8 && (int64_t)a.npix * d.Co * 2 < (1ll << 32) &&
            (!a.plain_x || (int64_t)a.npix * a.ncols * 2 < (1ll << 32))) ? 1 : 0;
  const int nkz = nsplit * (int)groups;
  static const int f_map = getenv("IEEE_WGRAD_MAP") ? atoi(getenv("IEEE_WGRAD_MAP")) : 2;
  a.xcd_group = (nkz >= 24 || nkz % 8 == 0) ? 1 : f_map;
  dim3 grid((unsigned)(a.tiles * (a.xcd_group == 1 ? cdiv(nkz, 8) * 8 : nkz)));
  // default: single-stage LDS-DMA (106 VGPRs -> 4 workgroups per CU): -6 % wgrad time over register staging at 3
  static const int f_pipe = getenv("IEEE_WGRAD_PIPE") ? atoi(getenv("IEEE_WGRAD_PIPE")) : 1;
  const int pipe = (dtype == IEEE_BF16 && !slow) ? f_pipe : 0;
  // IEEE_WGRAD_LDS: dynamic LDS of the weight-gradient kernels in KB (>= what they use): caps their workgroups per CU, i.e. how
  // much of every CU the low-priority stream may hold against the launch stream's kernels.  48 KB = 3 per CU instead of 4:
  // each weight gradient alone is a little slower (511 vs 515 TFLOP/s serialized), the step is 0.15 ms faster (15.31 -> 15.16,
  // five interleaved rounds): the dgrad / BatchNorm chain finds free slots sooner.  64 KB (2 per CU) loses (15.4).
  static const size_t f_lds = (size_t)(getenv("IEEE_WGRAD_LDS") ? atoi(getenv("IEEE_WGRAD_LDS")) : 48) * 1024;
  size_t smem = pipe ? (size_t)(pipe == 6 ? 2 : pipe) * 32 * 1024 : (dtype == IEEE_BF16 ? 32 * 1024 : 64 * 1024);
  if (smem < f_lds) smem = f_lds;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<bf16, false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<bf16, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  hipStream_t st = (hipStream_t)stream;
  float* slab = (float*)work;
  // one split of a 1x1 conv: the "slab" IS the gradient (slab order [co][ci] = OIHW), so the GEMM writes it in place and no
  // reduction pass runs (the two CIM convs: 2 x 50 MB written, read back and written again per step otherwise)
  const bool direct = nsplit == 1 && d.R * d.S == 1 && !accumulate && (dw_gs & 3) == 0 && ((uintptr_t)dw_oihw & 15) == 0;
  if (direct) {
    slab = dw_oihw;
    a.slab_gs = dw_gs;
  }
  const int64_t wsplits = wpatch_splits(dtype, N, Hi, Wi, Ci, Co, R, S, stride, pad, groups);
  const bool wpatch = wsplits > 0;
  int nsplit_used = nsplit;
  // the split-K fold inside the launch (FoldArgs): bf16 LDS-DMA forms with 16-byte aligned slabs / gradient
  FoldArgs fa{};
  bool fold = false;
  {
    const int ns = wpatch ? (int)wsplits : nsplit;
    const int tiles = wpatch ? cdiv(d.Co, 128) * (d.Ci / 64) * 3 : a.tiles;
    if (tickets != nullptr && defer == nullptr && dtype == IEEE_BF16 && !slow && (wpatch || pipe == 1) && !direct && ns >= 2 &&
        ns <= FOLD_MAX_SPLITS && a.ncols % 4 == 0 && ((uintptr_t)slab & 15) == 0 && ((uintptr_t)dw_oihw & 15) == 0 &&
        (dw_gs & 3) == 0 && (d.Co % 128 == 0 || d.Co == 64)) {
      fold_plan(ns, &fa.radix, &fa.n1);
      fa.nsplit = ns;
      fa.tiles = tiles;
      fold = (int64_t)groups * tiles * (fa.n1 + 1) <= FOLD_TICKET_WORDS;
      fa.tickets = tickets;
      fa.dw = dw_oihw;
      fa.dw_gs = dw_gs;
      fa.lvl1_gs = (int64_t)fa.n1 * d.Co * a.ncols;
      fa.lvl1 = slab + (int64_t)groups * ns * d.Co * a.ncols;
      fa.accumulate = accumulate;
      fa.RS = d.R * d.S;
      fa.Ci = d.Ci;
    }
  }
  if (wpatch) {
    WgradPatchArgs pa;
    const int64_t ktiles_all = a.npix / 64, per = (ktiles_all + wsplits - 1) / wsplits;
    nsplit_used = (int)wsplits;
    a.kchunk = (int)(per * 64);
    a.slab_gs = (int64_t)nsplit_used * d.Co * a.ncols;
    const int nkz = nsplit_used * (int)groups;
    pa.Hm = d.Hi; pa.Ci = d.Ci; pa.Co = d.Co; pa.npix = a.npix; pa.kchunk = a.kchunk; pa.nsplit = nsplit_used; pa.groups = (int)groups;
    pa.nchunks = d.Ci / 64;
    pa.tiles = cdiv(d.Co, 128) * pa.nchunks * 3;
    pa.ncols = a.ncols;
    pa.dy_gs = dy_gs; pa.x_gs = x_gs; pa.slab_gs = a.slab_gs;
    dim3 pgrid((unsigned)(pa.tiles * cdiv(nkz, 8) * 8));
    const int wlog = d.Wi == 8 ? 3 : (d.Wi == 16 ? 4 : 5);
    const bool half = d.Co == 64;
#define IEEE_WP_CASE(W_) \
    if (half && fold) conv3x3_wgrad_patch_kernel<W_, true, true><<<pgrid, 256, std::max((size_t)(64 * 256 + WPatch<W_>::BYTES), f_lds), st>>>((const bf16*)dy, (const bf16*)x, slab, pa, fa, pv); \
    else if (fold) conv3x3_wgrad_patch_kernel<W_, false, true><<<pgrid, 256, std::max((size_t)(64 * 256 + WPatch<W_>::BYTES), f_lds), st>>>((const bf16*)dy, (const bf16*)x, slab, pa, fa, pv); \
    else if (half) conv3x3_wgrad_patch_kernel<W_, true><<<pgrid, 256, std::max((size_t)(64 * 256 + WPatch<W_>::BYTES), f_lds), st>>>((const bf16*)dy, (const bf16*)x, slab, pa, fa, pv); \
    else conv3x3_wgrad_patch_kernel<W_, false><<<pgrid, 256, std::max((size_t)(64 * 256 + WPatch<W_>::BYTES), f_lds), st>>>((const bf16*)dy, (const bf16*)x, slab, pa, fa, pv)
    if (wlog == 3) { IEEE_WP_CASE(3); } else if (wlog == 4) { IEEE_WP_CASE(4); } else { IEEE_WP_CASE(5); }
#undef IEEE_WP_CASE
  } else if (dtype == IEEE_F32) {
    if (slow) conv_wgrad_kernel<float, true><<<grid, 256, smem, st>>>((const float*)dy, (const float*)x, slab, a, pv);
    else conv_wgrad_kernel<float, false><<<grid, 256, smem, st>>>((const float*)dy, (const float*)x, slab, a, pv);
  } else if (dtype == IEEE_BF16) {
    if (slow) conv_wgrad_kernel<bf16, true><<<grid, 256, smem, st>>>((const bf16*)dy, (const bf16*)x, slab, a, pv);
    else if (fold && d.Co <= 64) conv_wgrad_fold_kernel<true><<<grid, 256, smem, st>>>((const bf16*)dy, (const bf16*)x, slab, a, fa, pv);
    else if (fold) conv_wgrad_fold_kernel<false><<<grid, 256, smem, st>>>((const bf16*)dy, (const bf16*)x, slab, a, fa, pv);
    else if (pipe == 1 && d.Co <= 64) conv_wgrad_kernel<bf16, false, 1, true><<<grid, 256, smem, st>>>((const bf16*)dy, (const bf16*)x, slab, a, pv);
    else if (pipe == 1) conv_wgrad_kernel<bf16, false, 1><<<grid, 256, smem, st>>>((const bf16*)dy, (const bf16*)x, slab, a, pv);
    else if (pipe == 2) conv_wgrad_kernel<bf16, false, 2><<<grid, 256, smem, st>>>((const bf16*)dy, (const bf16*)x, slab, a, pv);
    else if (pipe == 3) conv_wgrad_kernel<bf16, false, 3><<<grid, 256, smem, st>>>((const bf16*)dy, (const bf16*)x, slab, a, pv);
    else if (pipe == 4) conv_wgrad_kernel<bf16, false, 4><<<grid, 256, smem, st>>>((const bf16*)dy, (const bf16*)x, slab, a, pv);
    else if (pipe == 6) conv_wgrad_kernel<bf16, false, 6><<<grid, 256, smem, st>>>((const bf16*)dy, (const bf16*)x, slab, a, pv);
    else conv_wgrad_kernel<bf16, false><<<grid, 256, smem, st>>>((const bf16*)dy, (const bf16*)x, slab, a, pv);
  } else {
    IEEE_REQUIRE(false, "conv2d_wgrad: bad dtype %d", dtype);
  }
  IEEE_TRY(launch_status("conv_wgrad_kernel"));
  if (defer) fill_reduce_desc(defer, nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
  if (direct || fold) return IEEE_OK;
  const int64_t total = (int64_t)d.Co * d.Ci * d.R * d.S;
  const bool vec4 = d.R * d.S == 1 && (total & 3) == 0 && (dw_gs & 3) == 0 && (a.slab_gs & 3) == 0 &&
                    ((uintptr_t)dw_oihw & 15) == 0 && ((uintptr_t)slab & 15) == 0;
  // split lanes: enough blocks to fill the chip on the small-weight layers, no idle lanes on the few-split ones
  const int64_t outs = vec4 ? total / 4 : total;
  int sl_log2 = 0;
  while (sl_log2 < 4 && (2 << sl_log2) <= nsplit_used && (outs << sl_log2) * groups < (int64_t)256 * 2048) ++sl_log2;
  dim3 rgrid(cdiv(outs, 256 >> sl_log2), (unsigned)groups);
  static const int f_taps = getenv("IEEE_WGRAD_REDUCE_TAPS") ? atoi(getenv("IEEE_WGRAD_REDUCE_TAPS")) : 16;
  if (d.R * d.S > 1 && nsplit_used <= f_taps && d.Ci % 4 == 0 && (a.slab_gs & 3) == 0 && ((uintptr_t)slab & 15) == 0) {
    const int RS = d.R * d.S;
    dim3 tgrid((unsigned)(d.Co * cdiv(d.Ci, RT_CIB)), (unsigned)groups);
    if (defer && RS <= 9) {
      fill_reduce_desc(defer, slab, dw_oihw, a.slab_gs, dw_gs, nsplit_used, d.Co, d.Ci, RS, 3, 0, (int)tgrid.x, accumulate);
      return IEEE_OK;
    }
    wgrad_reduce_taps_kernel<<<tgrid, 256, (size_t)RT_CIB * RS * sizeof(float), st>>>(slab, dw_oihw, nsplit_used, d.Co, d.Ci, RS, a.slab_gs,
                                                                                    dw_gs, accumulate);
    return launch_status("wgrad_reduce_taps_kernel");
  }
  if (defer) {
    fill_reduce_desc(defer, slab, dw_oihw, a.slab_gs, dw_gs, nsplit_used, d.Co, d.Ci, d.R * d.S, vec4 ? 1 : 2, sl_log2, (int)rgrid.x, accumulate);
    return IEEE_OK;
  }
  if (vec4)
    wgrad_reduce_kernel<4><<<rgrid, 256, 0, st>>>(slab, dw_oihw, nsplit_used, d.Co, d.Ci, d.R * d.S, a.slab_gs, dw_gs, accumulate, sl_log2);
  else
    wgrad_reduce_kernel<1><<<rgrid, 256, 0, st>>>(slab, dw_oihw, nsplit_used, d.Co, d.Ci, d.R * d.S, a.slab_gs, dw_gs, accumulate, sl_log2);
  return launch_status("wgrad_reduce_kernel");
}

extern "C" int ieee_conv2d_wgrad(const void* dy, const void* x, float* dw_oihw, void* work, int dtype, int64_t groups,
                                 int64_t N, int64_t Hi, int64_t Wi, int64_t Ci, int64_t Co, int64_t R, int64_t S,
                                 int64_t stride, int64_t pad, int64_t dy_gs, int64_t x_gs, int64_t dw_gs,
                                 int accumulate, void* stream) {
  return wgrad_impl(dy, x, dw_oihw, work, dtype, groups, N, Hi, Wi, Ci, Co, R, S, stride, pad, dy_gs, x_gs, dw_gs, accumulate,
                    stream, nullptr);
}

extern "C" int64_t ieee_conv2d_wgrad_fold_ticket_words(void) { return FOLD_TICKET_WORDS; }

extern "C" int ieee_conv2d_wgrad_fold(const void* dy, const void* x, float* dw_oihw, void* work, int32_t* tickets, int dtype,
                                      int64_t groups, int64_t N, int64_t Hi, int64_t Wi, int64_t Ci, int64_t Co, int64_t R,
                                      int64_t S, int64_t stride, int64_t pad, int64_t dy_gs, int64_t x_gs, int64_t dw_gs,
                                      int accumulate, void* stream) {
  IEEE_REQUIRE(tickets, "conv2d_wgrad_fold: null ticket words");
  return wgrad_impl(dy, x, dw_oihw, work, dtype, groups, N, Hi, Wi, Ci, Co, R, S, stride, pad, dy_gs, x_gs, dw_gs, accumulate,
                    stream, nullptr, tickets);
}

extern "C" int ieee_conv2d_wgrad_deferred(const void* dy, const void* x, float* dw_oihw, void* work, int dtype, int64_t groups,
                                          int64_t N, int64_t Hi, int64_t Wi, int64_t Ci, int64_t Co, int64_t R, int64_t S,
                                          int64_t stride, int64_t pad, int64_t dy_gs, int64_t x_gs, int64_t dw_gs,
                                          int accumulate, ieee_wgrad_reduce_desc* reduce, void* stream) {
  IEEE_REQUIRE(reduce, "conv2d_wgrad_deferred: null descriptor");
  return wgrad_impl(dy, x, dw_oihw, work, dtype, groups, N, Hi, Wi, Ci, Co, R, S, stride, pad, dy_gs, x_gs, dw_gs, accumulate,
                    stream, reduce);
}

extern "C" int ieee_conv2d_wgrad_chained(const void* dy, const void* x, float* dw_oihw, void* work, int dtype, int64_t groups,
                                         int64_t N, int64_t Hi, int64_t Wi, int64_t Ci, int64_t Co, int64_t R, int64_t S,
                                         int64_t stride, int64_t pad, int64_t dy_gs, int64_t x_gs, int64_t dw_gs,
                                         int accumulate, const ieee_wgrad_reduce_desc* prev, ieee_wgrad_reduce_desc* reduce,
                                         void* stream) {
  IEEE_REQUIRE(reduce, "conv2d_wgrad_chained: null descriptor");
  IEEE_REQUIRE(prev == nullptr || prev->kind == 0 || (const void*)prev->slab != (const void*)work,
               "conv2d_wgrad_chained: the pending reduction still reads `work` -- alternate between two slab regions");
  return wgrad_impl(dy, x, dw_oihw, work, dtype, groups, N, Hi, Wi, Ci, Co, R, S, stride, pad, dy_gs, x_gs, dw_gs, accumulate,
                    stream, reduce, nullptr, prev);
}

extern "C" int ieee_wgrad_reduce_pending(const ieee_wgrad_reduce_desc* pending, int64_t groups, void* stream) {
  IEEE_REQUIRE(pending && groups > 0, "wgrad_reduce_pending: bad arguments");
  return reduce_launch(*pending, groups, (hipStream_t)stream);
}

extern "C" int ieee_wgrad_reduce_batch(const ieee_wgrad_reduce_desc* device_descs, int64_t n, int64_t total_blocks, int64_t groups,
                                       void* stream) {
  IEEE_REQUIRE(device_descs && n > 0 && total_blocks > 0 && groups > 0, "wgrad_reduce_batch: bad arguments");
  wgrad_reduce_batch_kernel<<<dim3((unsigned)total_blocks, (unsigned)groups), 256, 0, (hipStream_t)stream>>>(device_descs, (int)n);
  return launch_status("wgrad_reduce_batch_kernel");
}

// explicit-argument forms: the options of ONE call as a struct; nothing stays armed (OneShotArms in the base entry points
// clears the thread's slots on every return path, and they are set here inside the same call)
static int arm_extras(const char* who, const ieee_conv_extras* ex) {
  if (ex == nullptr) return IEEE_OK;
  IEEE_REQUIRE(ex->reserved_ == 0, "%s: ieee_conv_extras.reserved_ must be 0", who);
  IEEE_REQUIRE((ex->start == nullptr) == (ex->stop == nullptr), "%s: pass both timing events or none", who);
  if (ex->totals != nullptr) {
    IEEE_REQUIRE(ex->replicas >= 1 && ex->replicas <= 64 && (ex->replicas & (ex->replicas - 1)) == 0,
                 "%s: ieee_conv_extras.replicas must be a power of two <= 64", who);
    ieee::tl_totals = (long long*)ex->totals;
    ieee::tl_totals_flag = ex->overflow;
    ieee::tl_totals_gs = ex->group_stride;
    ieee::tl_totals_rep = ex->replicas;
  }
  ieee::tl_time_start = (hipEvent_t)ex->start;
  ieee::tl_time_stop = (hipEvent_t)ex->stop;
  return IEEE_OK;
}

extern "C" int ieee_conv2d_fwd_ex(const void* x, const void* w_packed, void* y, int dtype, int64_t groups, int64_t N, int64_t Hi,
                                  int64_t Wi, int64_t Ci, int64_t Co, int64_t R, int64_t S, int64_t stride, int64_t pad,
                                  int64_t x_gs, int64_t w_gs, int64_t y_gs, float* bn_partial, const ieee_conv_extras* extras,
                                  void* stream) {
  ieee::OneShotArms disarm;      // (also when the argument check below fails)
  IEEE_TRY(arm_extras("conv2d_fwd_ex", extras));
  return ieee_conv2d_fwd(x, w_packed, y, dtype, groups, N, Hi, Wi, Ci, Co, R, S, stride, pad, x_gs, w_gs, y_gs, bn_partial, stream);
}

extern "C" int ieee_conv2d_dgrad_ex(const void* dy, const void* w_packed_d, void* dx, const void* addend, int dtype,
                                    int64_t groups, int64_t N, int64_t Hi, int64_t Wi, int64_t Ci, int64_t Co, int64_t R,
                                    int64_t S, int64_t stride, int64_t pad, int64_t dy_gs, int64_t w_gs, int64_t dx_gs,
                                    float* bn_partial, const void* bn_y, const void* bn_mask, const float* bn_stats,
                                    int bn_mask_bits, int addend_stride, const void* bn_y2, float* bn_partial2,
                                    const ieee_conv_extras* extras, void* stream) {
  ieee::OneShotArms disarm;
  IEEE_TRY(arm_extras("conv2d_dgrad_ex", extras));
  return ieee_conv2d_dgrad2(dy, w_packed_d, dx, addend, dtype, groups, N, Hi, Wi, Ci, Co, R, S, stride, pad, dy_gs, w_gs, dx_gs,
                            bn_partial, bn_y, bn_mask, bn_stats, bn_mask_bits, addend_stride, bn_y2, bn_partial2, stream);
}

/* measurement: the next conv forward / dgrad launch issued by this thread signals `start` when it begins and `stop` when it
 * ends (both hipEvent_t created with timing enabled; NULL, NULL cancels) */
extern "C" int ieee_conv_profile_events(void* start, void* stop) {
  IEEE_REQUIRE((start == nullptr) == (stop == nullptr), "conv_profile_events: pass both events or none");
  ieee::tl_time_start = (hipEvent_t)start;
  ieee::tl_time_stop = (hipEvent_t)stop;
  return IEEE_OK;
}

/* BatchNorm sums as order-independent totals: the next conv forward (fused statistics) / dgrad (fused backward sums) launch
 * of this thread adds its per-channel sums to totals[replica][group][2][C] (int64 fixed point: 2^24 forward, 2^40 backward;
 * the caller zeroes them; row tile t adds to replica t % replicas, a power of two <= 64) with no-return atomics instead of
 * writing per-tile partials; ieee_bn2d_fwd_totals / ieee_bn2d_bwd_totals add the replicas up.  NULL cancels.  The stem
 * (4 096 row tiles per modality) ignores it. */
extern "C" int ieee_conv_next_bn_totals(void* totals, int64_t group_stride, int replicas, int* overflow) {
  IEEE_REQUIRE(replicas >= 1 && replicas <= 64 && (replicas & (replicas - 1)) == 0, "conv_next_bn_totals: replicas must be a power of two <= 64");
  ieee::tl_totals = (long long*)totals;
  ieee::tl_totals_flag = totals != nullptr ? overflow : nullptr;
  ieee::tl_totals_gs = group_stride;
  ieee::tl_totals_rep = replicas;
  return IEEE_OK;
}
