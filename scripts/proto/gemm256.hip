// Prototype (tuning aid, not part of the product): bf16 NT GEMM C[M][N] = A[M][K] * B[N][K]^T with a 256x256x64
// workgroup tile, 8 waves (2 x 4, 128x64 outputs per wave), both operands through a 2-stage LDS-DMA ring.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I ieee_amd/csrc scripts/proto/gemm256.hip -o /tmp/gemm256
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
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
  const int variant = argc > 4 ? atoi(argv[4]) : 0;
  const int tiles_m = M / 256, tiles_n = N / 256;
  const size_t smem = 2 * 512 * 128;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&]() {
    if (variant == 1) gemm256_kernel<1><<<tiles_m * tiles_n, 512, smem>>>(dA, dB, dC, M, N, K, tiles_n);
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
