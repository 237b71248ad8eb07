// Backward of MaxPool2d(3, 2, 1) in gather form, shared by the plain max-pool backward (cim.hip) and the stem's fused
// max-pool + ReLU + BatchNorm backward (bn.hip): the gradient of input pixel (b, h, w), channels [ch*VEC, ch*VEC + VEC),
// is the sum of dout over the (at most 2 x 2) windows whose recorded argmax is that pixel, in a fixed order.
#pragma once
#include "common.h"

namespace ieee {

template <typename T>
__device__ __forceinline__ void pool_grad_gather(const T* __restrict__ dout, const uint8_t* __restrict__ arg, int b, int h,
                                                 int w, int ch, int Ho, int Wo, int C, float (&acc)[16 / sizeof(T)]) {
  constexpr int VEC = 16 / sizeof(T);
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
}

}  // namespace ieee
