// Prototype (tuning aid, not part of the product): bf16 NT GEMM C[M][N] = A[M][K] * B[N][K]^T with a 256x256x64
// workgroup tile, 8 waves (2 x 4, 128x64 outputs per wave), both operands through a 2-stage LDS-DMA ring.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I ieee_amd/csrc scripts/proto/gemm256.hip -o /tmp/gemm256
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>

#include "gemm_core.h"

using namespace ieee;

template <int DUMMY>
__global__ __launch_bounds__(512, 1) void gemm256_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B,
                                                         bf16* __restrict__ C, int M, int N, int K, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef ImgNT<bf16> Img;
  constexpr int STAGE = 512 * 128;                 // (256 + 256) rows of 128 B
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 2, wn = wave & 3;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  // XCD-aware tile order: consecutive workgroup ids of one XCD walk n fastest inside a band of m tiles
  const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
  const int m0 = tm * 256, n0 = tn * 256;
  const int ch = nt_dma_chunk(t & 255);            // swizzle key depends on the row inside a 32-row group
  const bf16* pa[4];
  const bf16* pb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (t >> 3) + 64 * i;
    pa[i] = A + (int64_t)(m0 + row) * K + ch * 8;
    pb[i] = B + (int64_t)(n0 + row) * K + ch * 8;
  }
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto issue = [&](int kt, int stage_idx) {
    char* stage = smem + stage_idx * STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(pa[i] + kt * 64, stage + (64 * i + 8 * wave_u) * 128);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(pb[i] + kt * 64, stage + 256 * 128 + (64 * i + 8 * wave_u) * 128);
  };
  const int ktiles = K / 64;
  issue(0, 0);
  for (int kt = 0; kt < ktiles; ++kt) {
    if (kt + 1 < ktiles) {
      issue(kt + 1, (kt + 1) & 1);
      wait_vmcnt<8>();
    } else {
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    const char* cur = smem + (kt & 1) * STAGE;
    const char* At = cur + (wm * 128) * 128;
    const char* Bt = cur + 256 * 128 + (wn * 64) * 128;
    if constexpr (DUMMY == 0) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        Img::Frag fa[8], fb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = Img::frag(Bt, j * 16, kk, lane);
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] = Img::frag(At, i * 16, kk, lane);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(fb[j], fa[i], acc[i][j]);
      }
    } else {
      // variant 1: both kk fragment sets are read up front (the second set lands under the first MFMA cluster),
      // MFMA clusters run at raised priority
      Img::Frag fa[2][8], fb[2][4];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[kk][j] = Img::frag(Bt, j * 16, kk, lane);
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[kk][i] = Img::frag(At, i * 16, kk, lane);
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(fb[kk][j], fa[kk][i], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
      }
    }
    __syncthreads();   // every wave is done reading this stage (lgkmcnt(0)); no DMA of this wave is... see note
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + wm * 128 + i * 16 + (lane & 15);
      const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
      *(uint2*)(C + (int64_t)m * N + n) = make_uint2(Vec16<bf16>::pk(acc[i][j][0], acc[i][j][1]), Vec16<bf16>::pk(acc[i][j][2], acc[i][j][3]));
    }
}

// variant 2: phase schedule.  Per k-tile four quadrant phases of 16 MFMAs; the next tile's operand groups (64 rows =
// 8 KB each, 2-3 DMAs per thread per phase) are issued between them: A groups 1,3 + B groups at phases 0/1 into the
// other buffer, and the tile-after-next's A groups 0,2 at phase 2 into THIS buffer (every wave has finished with them
// after phase 1).  One counted vmcnt and two barriers per k-tile.
__global__ __launch_bounds__(512, 1) void gemm256_phase_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B,
                                                               bf16* __restrict__ C, int M, int N, int K, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef ImgNT<bf16> Img;
  constexpr int STAGE = 512 * 128;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wr = wave >> 2, wc = wave & 3;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
  const int m0 = tm * 256, n0 = tn * 256;
  const int ch = nt_dma_chunk(t & 255);
  const bf16* pa[4];
  const bf16* pb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (t >> 3) + 64 * i;
    pa[i] = A + (int64_t)(m0 + row) * K + ch * 8;
    pb[i] = B + (int64_t)(n0 + row) * K + ch * 8;
  }
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto issueA = [&](int kt, int buf, int g) { glds16(pa[g] + kt * 64, smem + buf * STAGE + (64 * g + 8 * wave_u) * 128); };
  auto issueB = [&](int kt, int buf, int g) { glds16(pb[g] + kt * 64, smem + buf * STAGE + 256 * 128 + (64 * g + 8 * wave_u) * 128); };
  const int ktiles = K / 64;
#pragma unroll
  for (int g = 0; g < 4; ++g) { issueA(0, 0, g); issueB(0, 0, g); }
  if (ktiles > 1) { issueA(1, 1, 0); issueA(1, 1, 2); wait_vmcnt<2>(); } else { wait_vmcnt<0>(); }
  __builtin_amdgcn_s_barrier();
  for (int kt = 0; kt < ktiles; ++kt) {
    const int b = kt & 1, nb = b ^ 1;
    const char* cur = smem + b * STAGE;
    const char* At = cur + (wr * 128) * 128;
    const char* Bt = cur + 256 * 128 + (wc * 64) * 128;
    auto quadrant = [&](int rh, int chh) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        Img::Frag fa[4], fb[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = Img::frag(Bt + (chh * 32) * 128, j * 16, kk, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = Img::frag(At + (rh * 64) * 128, i * 16, kk, lane);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[rh * 4 + i][chh * 2 + j] = mfma16(fb[j], fa[i], acc[rh * 4 + i][chh * 2 + j]);
        __builtin_amdgcn_s_setprio(0);
      }
    };
    const bool n1 = kt + 1 < ktiles, n2 = kt + 2 < ktiles;
    if (n1) { issueA(kt + 1, nb, 1); issueA(kt + 1, nb, 3); issueB(kt + 1, nb, 0); }
    quadrant(0, 0);
    if (n1) { issueB(kt + 1, nb, 1); issueB(kt + 1, nb, 2); issueB(kt + 1, nb, 3); }
    quadrant(0, 1);
    __syncthreads();                       // every wave is done with A groups 0 and 2 of this buffer
    if (n2) { issueA(kt + 2, b, 0); issueA(kt + 2, b, 2); }
    quadrant(1, 1);
    quadrant(1, 0);
    if (n2) wait_vmcnt<2>(); else wait_vmcnt<0>();
    __syncthreads();                       // next tile has landed for every wave; this buffer is free
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + wr * 128 + i * 16 + (lane & 15);
      const int n = n0 + wc * 64 + j * 16 + (lane >> 4) * 4;
      *(uint2*)(C + (int64_t)m * N + n) = make_uint2(Vec16<bf16>::pk(acc[i][j][0], acc[i][j][1]), Vec16<bf16>::pk(acc[i][j][2], acc[i][j][3]));
    }
}

// variant 3: variant 2's operand-group schedule with the guide's phase shape and wave-group stagger: every phase is
// {ds_reads of the quadrant; s_barrier; 16 MFMAs; s_barrier}, and the waves of the lower half of the tile (wr == 1, the
// second wave of every SIMD) run one barrier late, so on each SIMD one wave reads LDS while the other feeds the MFMA
// pipe.  DMA issue points are moved later accordingly (a group is re-staged >= 2 phases after its last reader).
__global__ __launch_bounds__(512, 1) void gemm256_stagger_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B,
                                                                 bf16* __restrict__ C, int M, int N, int K, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef ImgNT<bf16> Img;
  constexpr int STAGE = 512 * 128;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wr = wave >> 2, wc = wave & 3;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int wr_u = __builtin_amdgcn_readfirstlane(wr);
  const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
  const int m0 = tm * 256, n0 = tn * 256;
  const int ch = nt_dma_chunk(t & 255);
  const bf16* pa[4];
  const bf16* pb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (t >> 3) + 64 * i;
    pa[i] = A + (int64_t)(m0 + row) * K + ch * 8;
    pb[i] = B + (int64_t)(n0 + row) * K + ch * 8;
  }
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto issueA = [&](int kt, int buf, int g) { glds16(pa[g] + kt * 64, smem + buf * STAGE + (64 * g + 8 * wave_u) * 128); };
  auto issueB = [&](int kt, int buf, int g) { glds16(pb[g] + kt * 64, smem + buf * STAGE + 256 * 128 + (64 * g + 8 * wave_u) * 128); };
  const int ktiles = K / 64;
#pragma unroll
  for (int g = 0; g < 4; ++g) { issueA(0, 0, g); issueB(0, 0, g); }
  if (ktiles > 1) { issueA(1, 1, 0); issueA(1, 1, 2); wait_vmcnt<2>(); } else { wait_vmcnt<0>(); }
  __builtin_amdgcn_s_barrier();
  if (wr_u == 1) __builtin_amdgcn_s_barrier();      // stagger: the second wave of every SIMD runs one barrier late
  for (int kt = 0; kt < ktiles; ++kt) {
    const int b = kt & 1, nb = b ^ 1;
    const char* cur = smem + b * STAGE;
    const char* At = cur + (wr * 128) * 128;
    const char* Bt = cur + 256 * 128 + (wc * 64) * 128;
    auto quadrant = [&](int rh, int chh) {
      Img::Frag fa[2][4], fb[2][2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[kk][j] = Img::frag(Bt + (chh * 32) * 128, j * 16, kk, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[kk][i] = Img::frag(At + (rh * 64) * 128, i * 16, kk, lane);
      }
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[rh * 4 + i][chh * 2 + j] = mfma16(fb[kk][j], fa[kk][i], acc[rh * 4 + i][chh * 2 + j]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
    };
    const bool n1 = kt + 1 < ktiles, n2 = kt + 2 < ktiles;
    quadrant(0, 0);
    if (n1) { issueA(kt + 1, nb, 1); issueA(kt + 1, nb, 3); issueB(kt + 1, nb, 0); issueB(kt + 1, nb, 1); issueB(kt + 1, nb, 2); issueB(kt + 1, nb, 3); }
    quadrant(0, 1);
    quadrant(1, 1);
    wait_vmcnt<0>();                       // the next tile (and, long ago, its A groups 0/2) has landed for this wave
    if (n2) { issueA(kt + 2, b, 0); issueA(kt + 2, b, 2); }
    quadrant(1, 0);
  }
  if (wr_u == 0) __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + wr * 128 + i * 16 + (lane & 15);
      const int n = n0 + wc * 64 + j * 16 + (lane >> 4) * 4;
      *(uint2*)(C + (int64_t)m * N + n) = make_uint2(Vec16<bf16>::pk(acc[i][j][0], acc[i][j][1]), Vec16<bf16>::pk(acc[i][j][2], acc[i][j][3]));
    }
}

// variant 4 = variant 3 + fragment reuse across phases (A sub-tile kept for two phases, B sub-tile for two): 28
// instead of 48 ds_read_b128 per wave and k-tile.  variant 3 text: every phase is
// {ds_reads of the quadrant; s_barrier; 16 MFMAs; s_barrier}, and the waves of the lower half of the tile (wr == 1, the
// second wave of every SIMD) run one barrier late, so on each SIMD one wave reads LDS while the other feeds the MFMA
// pipe.  DMA issue points are moved later accordingly (a group is re-staged >= 2 phases after its last reader).
__global__ __launch_bounds__(512, 1) void gemm256_reuse_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B,
                                                                 bf16* __restrict__ C, int M, int N, int K, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef ImgNT<bf16> Img;
  constexpr int STAGE = 512 * 128;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wr = wave >> 2, wc = wave & 3;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int wr_u = __builtin_amdgcn_readfirstlane(wr);
  const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
  const int m0 = tm * 256, n0 = tn * 256;
  const int ch = nt_dma_chunk(t & 255);
  const bf16* pa[4];
  const bf16* pb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (t >> 3) + 64 * i;
    pa[i] = A + (int64_t)(m0 + row) * K + ch * 8;
    pb[i] = B + (int64_t)(n0 + row) * K + ch * 8;
  }
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto issueA = [&](int kt, int buf, int g) { glds16(pa[g] + kt * 64, smem + buf * STAGE + (64 * g + 8 * wave_u) * 128); };
  auto issueB = [&](int kt, int buf, int g) { glds16(pb[g] + kt * 64, smem + buf * STAGE + 256 * 128 + (64 * g + 8 * wave_u) * 128); };
  const int ktiles = K / 64;
#pragma unroll
  for (int g = 0; g < 4; ++g) { issueA(0, 0, g); issueB(0, 0, g); }
  if (ktiles > 1) { issueA(1, 1, 0); issueA(1, 1, 2); wait_vmcnt<2>(); } else { wait_vmcnt<0>(); }
  __builtin_amdgcn_s_barrier();
  if (wr_u == 1) __builtin_amdgcn_s_barrier();      // stagger: the second wave of every SIMD runs one barrier late
  for (int kt = 0; kt < ktiles; ++kt) {
    const int b = kt & 1, nb = b ^ 1;
    const char* cur = smem + b * STAGE;
    const char* At = cur + (wr * 128) * 128;
    const char* Bt = cur + 256 * 128 + (wc * 64) * 128;
    Img::Frag fa[2][4], fb[2][2];
    auto loadA = [&](int rh) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[kk][i] = Img::frag(At + (rh * 64) * 128, i * 16, kk, lane);
    };
    auto loadB = [&](int chh) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[kk][j] = Img::frag(Bt + (chh * 32) * 128, j * 16, kk, lane);
    };
    auto mma = [&](int rh, int chh) {
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[rh * 4 + i][chh * 2 + j] = mfma16(fb[kk][j], fa[kk][i], acc[rh * 4 + i][chh * 2 + j]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
    };
    const bool n1 = kt + 1 < ktiles, n2 = kt + 2 < ktiles;
    loadB(0); loadA(0); mma(0, 0);
    if (n1) { issueA(kt + 1, nb, 1); issueA(kt + 1, nb, 3); issueB(kt + 1, nb, 0); issueB(kt + 1, nb, 1); issueB(kt + 1, nb, 2); issueB(kt + 1, nb, 3); }
    loadB(1); mma(0, 1);
    loadA(1); mma(1, 1);
    wait_vmcnt<0>();                       // the next tile (and, long ago, its A groups 0/2) has landed for this wave
    if (n2) { issueA(kt + 2, b, 0); issueA(kt + 2, b, 2); }
    loadB(0); mma(1, 0);
  }
  if (wr_u == 0) __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + wr * 128 + i * 16 + (lane & 15);
      const int n = n0 + wc * 64 + j * 16 + (lane >> 4) * 4;
      *(uint2*)(C + (int64_t)m * N + n) = make_uint2(Vec16<bf16>::pk(acc[i][j][0], acc[i][j][1]), Vec16<bf16>::pk(acc[i][j][2], acc[i][j][3]));
    }
}

// variant 5 = variant 4 with the LDS reads retired BEFORE the mid-phase barrier (so a group may be re-staged one phase
// after its last reader), DMA issue spread over phases 0-2 (4 + 2 + 2 per thread) and one counted vmcnt(2) per k-tile.
// variant 4 = variant 3 + fragment reuse across phases (A sub-tile kept for two phases, B sub-tile for two): 28
// instead of 48 ds_read_b128 per wave and k-tile.  variant 3 text: every phase is
// {ds_reads of the quadrant; s_barrier; 16 MFMAs; s_barrier}, and the waves of the lower half of the tile (wr == 1, the
// second wave of every SIMD) run one barrier late, so on each SIMD one wave reads LDS while the other feeds the MFMA
// pipe.  DMA issue points are moved later accordingly (a group is re-staged >= 2 phases after its last reader).
__global__ __launch_bounds__(512, 1) void gemm256_v5_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B,
                                                                 bf16* __restrict__ C, int M, int N, int K, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef ImgNT<bf16> Img;
  constexpr int STAGE = 512 * 128;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wr = wave >> 2, wc = wave & 3;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int wr_u = __builtin_amdgcn_readfirstlane(wr);
  const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
  const int m0 = tm * 256, n0 = tn * 256;
  const int ch = nt_dma_chunk(t & 255);
  const bf16* pa[4];
  const bf16* pb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (t >> 3) + 64 * i;
    pa[i] = A + (int64_t)(m0 + row) * K + ch * 8;
    pb[i] = B + (int64_t)(n0 + row) * K + ch * 8;
  }
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto issueA = [&](int kt, int buf, int g) { glds16(pa[g] + kt * 64, smem + buf * STAGE + (64 * g + 8 * wave_u) * 128); };
  auto issueB = [&](int kt, int buf, int g) { glds16(pb[g] + kt * 64, smem + buf * STAGE + 256 * 128 + (64 * g + 8 * wave_u) * 128); };
  const int ktiles = K / 64;
#pragma unroll
  for (int g = 0; g < 4; ++g) { issueA(0, 0, g); issueB(0, 0, g); }
  if (ktiles > 1) { issueA(1, 1, 0); issueA(1, 1, 2); wait_vmcnt<2>(); } else { wait_vmcnt<0>(); }
  __builtin_amdgcn_s_barrier();
  if (wr_u == 1) __builtin_amdgcn_s_barrier();      // stagger: the second wave of every SIMD runs one barrier late
  for (int kt = 0; kt < ktiles; ++kt) {
    const int b = kt & 1, nb = b ^ 1;
    const char* cur = smem + b * STAGE;
    const char* At = cur + (wr * 128) * 128;
    const char* Bt = cur + 256 * 128 + (wc * 64) * 128;
    Img::Frag fa[2][4], fb[2][2];
    auto loadA = [&](int rh) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[kk][i] = Img::frag(At + (rh * 64) * 128, i * 16, kk, lane);
    };
    auto loadB = [&](int chh) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[kk][j] = Img::frag(Bt + (chh * 32) * 128, j * 16, kk, lane);
    };
    auto mma = [&](int rh, int chh) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[rh * 4 + i][chh * 2 + j] = mfma16(fb[kk][j], fa[kk][i], acc[rh * 4 + i][chh * 2 + j]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
    };
    const bool n1 = kt + 1 < ktiles, n2 = kt + 2 < ktiles;
    if (n1) { issueA(kt + 1, nb, 1); issueA(kt + 1, nb, 3); issueB(kt + 1, nb, 0); issueB(kt + 1, nb, 1); }
    loadB(0); loadA(0); mma(0, 0);
    if (n1) { issueB(kt + 1, nb, 2); issueB(kt + 1, nb, 3); }
    loadB(1); mma(0, 1);
    if (n2) { issueA(kt + 2, b, 0); issueA(kt + 2, b, 2); }
    loadA(1); mma(1, 1);
    if (n2) wait_vmcnt<2>(); else wait_vmcnt<0>();   // the next tile has landed for this wave (A 0/2 of the one after may fly)
    loadB(0); mma(1, 0);
  }
  if (wr_u == 0) __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + wr * 128 + i * 16 + (lane & 15);
      const int n = n0 + wc * 64 + j * 16 + (lane >> 4) * 4;
      *(uint2*)(C + (int64_t)m * N + n) = make_uint2(Vec16<bf16>::pk(acc[i][j][0], acc[i][j][1]), Vec16<bf16>::pk(acc[i][j][2], acc[i][j][3]));
    }
}

// variant 6: software-pipelined fragments, ONE barrier per k-tile.  Four fragment register sets (A rows 0-63 / 64-127
// of the wave's 128, B columns 0-31 / 32-63 of its 64; 96 VGPRs) are refilled while the previous quadrant's 16 MFMAs
// run; the k-tile hand-over (all reads of this buffer retired, next tile landed, barrier) sits between quadrants 2 and 3,
// the DMA of the tile after next goes into the just-freed buffer right after it (four phases of flight time).
__global__ __launch_bounds__(512, 1) void gemm256_v6_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B,
                                                            bf16* __restrict__ C, int M, int N, int K, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef ImgNT<bf16> Img;
  constexpr int STAGE = 512 * 128;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wr = wave >> 2, wc = wave & 3;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
  const int m0 = tm * 256, n0 = tn * 256;
  const int ch = nt_dma_chunk(t & 255);
  const bf16* pa[4];
  const bf16* pb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (t >> 3) + 64 * i;
    pa[i] = A + (int64_t)(m0 + row) * K + ch * 8;
    pb[i] = B + (int64_t)(n0 + row) * K + ch * 8;
  }
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto issue_tile = [&](int kt, int buf) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      glds16(pa[g] + kt * 64, smem + buf * STAGE + (64 * g + 8 * wave_u) * 128);
      glds16(pb[g] + kt * 64, smem + buf * STAGE + 256 * 128 + (64 * g + 8 * wave_u) * 128);
    }
  };
  Img::Frag fa[2][2][4], fb[2][2][2];      // [set][kk][frag]
  auto loadA = [&](int set, int buf, int rh) {
    const char* At = smem + buf * STAGE + (wr * 128 + rh * 64) * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[set][kk][i] = Img::frag(At, i * 16, kk, lane);
  };
  auto loadB = [&](int set, int buf, int chh) {
    const char* Bt = smem + buf * STAGE + 256 * 128 + (wc * 64 + chh * 32) * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[set][kk][j] = Img::frag(Bt, j * 16, kk, lane);
  };
  auto mma = [&](int as, int bs, int rh, int chh) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[rh * 4 + i][chh * 2 + j] = mfma16(fb[bs][kk][j], fa[as][kk][i], acc[rh * 4 + i][chh * 2 + j]);
  };
  const int ktiles = K / 64;      // even
  issue_tile(0, 0);
  if (ktiles > 1) issue_tile(1, 1);
  if (ktiles > 1) wait_vmcnt<8>(); else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  loadA(0, 0, 0);
  loadB(0, 0, 0);
  // one k-tile; BS = register set holding B columns 0-31 of this tile (alternates between tiles)
  auto tile = [&](int kt, auto bs_tag) {
    constexpr int BS = decltype(bs_tag)::value;
    const int b = kt & 1, nb = b ^ 1;
    loadB(BS ^ 1, b, 1);
    mma(0, BS, 0, 0);
    loadA(1, b, 1);
    mma(0, BS ^ 1, 0, 1);
    loadB(BS, b, 0);
    mma(1, BS ^ 1, 1, 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's reads of buffer b are retired
    wait_vmcnt<0>();                                        // tile kt+1 (issued four quadrants ago) has landed
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < ktiles) issue_tile(kt + 2, b);
    if (kt + 1 < ktiles) { loadA(0, nb, 0); loadB(BS ^ 1, nb, 0); }
    mma(1, BS, 1, 0);
  };
  for (int kt = 0; kt < ktiles; kt += 2) {
    tile(kt, std::integral_constant<int, 0>{});
    tile(kt + 1, std::integral_constant<int, 1>{});
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + wr * 128 + i * 16 + (lane & 15);
      const int n = n0 + wc * 64 + j * 16 + (lane >> 4) * 4;
      *(uint2*)(C + (int64_t)m * N + n) = make_uint2(Vec16<bf16>::pk(acc[i][j][0], acc[i][j][1]), Vec16<bf16>::pk(acc[i][j][2], acc[i][j][3]));
    }
}

// variant 7 = variant 6 with SGPR-base + 32-bit-offset DMA addressing (two VGPRs of address state instead of 16).
// variant 6: software-pipelined fragments, ONE barrier per k-tile.  Four fragment register sets (A rows 0-63 / 64-127
// of the wave's 128, B columns 0-31 / 32-63 of its 64; 96 VGPRs) are refilled while the previous quadrant's 16 MFMAs
// run; the k-tile hand-over (all reads of this buffer retired, next tile landed, barrier) sits between quadrants 2 and 3,
// the DMA of the tile after next goes into the just-freed buffer right after it (four phases of flight time).
__global__ __launch_bounds__(512, 1) void gemm256_v7_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B,
                                                            bf16* __restrict__ C, int M, int N, int K, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef ImgNT<bf16> Img;
  constexpr int STAGE = 512 * 128;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wr = wave >> 2, wc = wave & 3;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
  const int m0 = tm * 256, n0 = tn * 256;
  const int ch = nt_dma_chunk(t & 255);
  const unsigned voffA = (unsigned)(((t >> 3) * K + ch * 8) * 2);     // byte offset of this thread's chunk inside a 64-row group
  const char* baseA = (const char*)(A + (int64_t)m0 * K);
  const char* baseB = (const char*)(B + (int64_t)n0 * K);
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto dma = [&](const char* sbase, char* lds) {      // global_load_lds with a scalar base and a 32-bit per-lane offset
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voffA), "s"(sbase), "s"(dst)
                 : "memory");
  };
  auto issue_tile = [&](int kt, int buf) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      dma(baseA + ((int64_t)64 * g * K + kt * 64) * 2, smem + buf * STAGE + (64 * g + 8 * wave_u) * 128);
      dma(baseB + ((int64_t)64 * g * K + kt * 64) * 2, smem + buf * STAGE + 256 * 128 + (64 * g + 8 * wave_u) * 128);
    }
  };
  Img::Frag fa[2][2][4], fb[2][2][2];      // [set][kk][frag]
  auto loadA = [&](int set, int buf, int rh) {
    const char* At = smem + buf * STAGE + (wr * 128 + rh * 64) * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[set][kk][i] = Img::frag(At, i * 16, kk, lane);
  };
  auto loadB = [&](int set, int buf, int chh) {
    const char* Bt = smem + buf * STAGE + 256 * 128 + (wc * 64 + chh * 32) * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[set][kk][j] = Img::frag(Bt, j * 16, kk, lane);
  };
  auto mma = [&](int as, int bs, int rh, int chh) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[rh * 4 + i][chh * 2 + j] = mfma16(fb[bs][kk][j], fa[as][kk][i], acc[rh * 4 + i][chh * 2 + j]);
  };
  const int ktiles = K / 64;      // even
  issue_tile(0, 0);
  if (ktiles > 1) issue_tile(1, 1);
  if (ktiles > 1) wait_vmcnt<8>(); else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  loadA(0, 0, 0);
  loadB(0, 0, 0);
  // one k-tile; BS = register set holding B columns 0-31 of this tile (alternates between tiles)
  auto tile = [&](int kt, auto bs_tag) {
    constexpr int BS = decltype(bs_tag)::value;
    const int b = kt & 1, nb = b ^ 1;
    loadB(BS ^ 1, b, 1);
    mma(0, BS, 0, 0);
    loadA(1, b, 1);
    mma(0, BS ^ 1, 0, 1);
    loadB(BS, b, 0);
    mma(1, BS ^ 1, 1, 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's reads of buffer b are retired
    wait_vmcnt<0>();                                        // tile kt+1 (issued four quadrants ago) has landed
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < ktiles) issue_tile(kt + 2, b);
    if (kt + 1 < ktiles) { loadA(0, nb, 0); loadB(BS ^ 1, nb, 0); }
    mma(1, BS, 1, 0);
  };
  for (int kt = 0; kt < ktiles; kt += 2) {
    tile(kt, std::integral_constant<int, 0>{});
    tile(kt + 1, std::integral_constant<int, 1>{});
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + wr * 128 + i * 16 + (lane & 15);
      const int n = n0 + wc * 64 + j * 16 + (lane >> 4) * 4;
      *(uint2*)(C + (int64_t)m * N + n) = make_uint2(Vec16<bf16>::pk(acc[i][j][0], acc[i][j][1]), Vec16<bf16>::pk(acc[i][j][2], acc[i][j][3]));
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 24576, N = argc > 2 ? atoi(argv[2]) : 2048, K = argc > 3 ? atoi(argv[3]) : 2048;
  std::vector<unsigned short> ha((size_t)M * K), hb((size_t)N * K);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : ha) v = f2bf(rnd());
  for (auto& v : hb) v = f2bf(rnd());
  bf16 *dA, *dB, *dC;
  CK(hipMalloc(&dA, ha.size() * 2)); CK(hipMalloc(&dB, hb.size() * 2)); CK(hipMalloc(&dC, (size_t)M * N * 2));
  CK(hipMemcpy(dA, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void*)gemm256_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)gemm256_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)gemm256_v7_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)gemm256_v6_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)gemm256_v5_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)gemm256_reuse_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)gemm256_stagger_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)gemm256_phase_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int variant = argc > 4 ? atoi(argv[4]) : 0;
  const int tiles_m = M / 256, tiles_n = N / 256;
  const size_t smem = 2 * 512 * 128;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&]() {
    if (variant == 7) gemm256_v7_kernel<<<tiles_m * tiles_n, 512, smem>>>(dA, dB, dC, M, N, K, tiles_n);
    else if (variant == 6) gemm256_v6_kernel<<<tiles_m * tiles_n, 512, smem>>>(dA, dB, dC, M, N, K, tiles_n);
    else if (variant == 5) gemm256_v5_kernel<<<tiles_m * tiles_n, 512, smem>>>(dA, dB, dC, M, N, K, tiles_n);
    else if (variant == 4) gemm256_reuse_kernel<<<tiles_m * tiles_n, 512, smem>>>(dA, dB, dC, M, N, K, tiles_n);
    else if (variant == 3) gemm256_stagger_kernel<<<tiles_m * tiles_n, 512, smem>>>(dA, dB, dC, M, N, K, tiles_n);
    else if (variant == 2) gemm256_phase_kernel<<<tiles_m * tiles_n, 512, smem>>>(dA, dB, dC, M, N, K, tiles_n);
    else if (variant == 1) gemm256_kernel<1><<<tiles_m * tiles_n, 512, smem>>>(dA, dB, dC, M, N, K, tiles_n);
    else gemm256_kernel<0><<<tiles_m * tiles_n, 512, smem>>>(dA, dB, dC, M, N, K, tiles_n);
  };
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  const int reps = 20;
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  printf("variant %d M=%d N=%d K=%d: %.1f us  %.0f TFLOP/s\n", variant, M, N, K, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
  std::vector<unsigned short> hc((size_t)M * N);
  CK(hipMemcpy(hc.data(), dC, hc.size() * 2, hipMemcpyDeviceToHost));
  double maxerr = 0;
  for (int c = 0; c < 200; ++c) {
    const int m = (int)((unsigned)(c * 7919 + 13) % M), n = (int)((unsigned)(c * 104729 + 7) % N);
    double ref = 0;
    for (int k = 0; k < K; ++k) ref += (double)bf2f(ha[(size_t)m * K + k]) * bf2f(hb[(size_t)n * K + k]);
    const double got = bf2f(hc[(size_t)m * N + n]);
    const double err = fabs(got - ref) / (fabs(ref) + 1.0);
    if (err > maxerr) maxerr = err;
  }
  printf("max rel err over 200 samples: %.3e %s\n", maxerr, maxerr < 2e-2 ? "OK" : "MISMATCH");
  return 0;
}
