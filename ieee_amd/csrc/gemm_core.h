// LDS-tiled MFMA GEMM cores for gfx950 (wave64, 256-thread workgroups = 2x2 waves).
//
// Two cores share one accumulate/epilogue convention:
//   gemm_nt : C[m][n] = sum_k A[m][k] * B[n][k]   (both operands K-contiguous)
//             conv forward / dgrad (implicit GEMM over NHWC), distmat, linear
//   gemm_tn : C[m][n] = sum_k At[k][m] * Bt[k][n] (both operands K-major)
//             conv wgrad (K = output pixels); fragments come out of LDS through
//             ds_read_b64_tr_b16 (bf16) so no register transpose is needed
//
// dtypes: bf16 -> v_mfma_f32_16x16x32_bf16, fp32 -> v_mfma_f32_16x16x4_f32
// (exact fp32 fma chain; the parity mode and the fp32 distmat run on it).
//
// The MFMA is issued as D = Nfrag x Mfrag, so a lane ends up holding FOUR
// CONSECUTIVE n (channels) of ONE m (pixel): m = lane&15, n = (lane>>4)*4 + r.
// That makes the NHWC epilogue store 8 B (bf16) / 16 B (fp32) per lane.
//
// Global->LDS staging goes through registers (the im2col gather, zero padding
// and the transposed TN image cannot be expressed as a lane-linear LDS-DMA);
// next-tile loads are issued before the MFMA block and written after it.
#pragma once
#include "common.h"

namespace ieee {

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// 16-bit fragments issued as v_mfma_f32_16x16x32_f16 (F16) or ..._bf16: same LDS image, same lane layout
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
template <bool F16> __device__ __forceinline__ f32x4 mfma16_16bit(bf16x8 a, bf16x8 b, f32x4 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------ LDS images
// NT image: [rows][128 bytes of K]; 16-byte chunks XOR-swizzled so that the
// fragment reads below are bank-conflict free (ds_read_b128: 4 lane groups of
// 16 over a 256-B bank row; ds_read_b32: 2 groups of 32 over a 128-B bank row).
template <typename T> struct ImgNT;

template <> struct ImgNT<bf16> {
  static constexpr int BK = 64, KSTEPS = 2;
  typedef bf16x8 Frag;
  __device__ static __forceinline__ void store(char* tile, int row, int c, uint4 v) {
    *(uint4*)(tile + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = v;
  }
  __device__ static __forceinline__ Frag frag(const char* tile, int row16, int kk, int lane) {
    const int row = row16 + (lane & 15);
    const int c = kk * 4 + (lane >> 4);
    const uint4 v = *(const uint4*)(tile + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
    return __builtin_bit_cast(bf16x8, v);
  }
};

template <> struct ImgNT<float> {
  static constexpr int BK = 32, KSTEPS = 8;
  typedef float Frag;
  __device__ static __forceinline__ void store(char* tile, int row, int c, uint4 v) {
    if ((row >> 3) & 1) v = make_uint4(v.z, v.w, v.x, v.y);
    *(uint4*)(tile + row * 128 + ((c ^ (row & 7)) << 4)) = v;
  }
  __device__ static __forceinline__ Frag frag(const char* tile, int row16, int kk, int lane) {
    const int row = row16 + (lane & 15);
    const int p = (lane >> 4) ^ (((row >> 3) & 1) << 1);
    return *(const float*)(tile + row * 128 + ((kk ^ (row & 7)) << 4) + (p << 2));
  }
};

// TN image: [BK k-rows][128 columns]; columns contiguous (as they are in HBM).
template <typename T> struct ImgTN;

template <> struct ImgTN<bf16> {
  static constexpr int BK = 64, KSTEPS = 2, CPR = 16;  // 16 chunks (of 8 bf16) per 256-B k-row
  typedef bf16x8 Frag;
  __device__ static __forceinline__ int hsw(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }
  __device__ static __forceinline__ void store(char* tile, int kr, int c, uint4 v) {
    *(uint4*)(tile + kr * 256 + ((((c >> 1) ^ hsw(kr))) << 5) + ((c & 1) << 4)) = v;
  }
  __device__ static __forceinline__ Frag frag(const char* tile, int col16, int kk, int lane) {
    // ds_read_b64_tr_b16: per 16-lane group a 4(k) x 16(col) block, delivered column-major:
    // lane 4q+p supplies the address of row q, columns 4p..4p+3; lane i receives column i.
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int k0 = kk * 32 + 8 * g + q;
    const int w = col16 >> 4;
    const char* a0 = tile + k0 * 256 + ((w ^ hsw(k0)) << 5) + p * 8;
    const char* a1 = a0 + 4 * 256;  // rows k0+4..: same swizzle key (k&3 and (k>>3)&1 unchanged)
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a1));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  }
};

template <> struct ImgTN<float> {
  static constexpr int BK = 32, KSTEPS = 8, CPR = 32;  // 32 chunks (of 4 fp32) per 512-B k-row
  typedef float Frag;
  __device__ static __forceinline__ void store(char* tile, int kr, int c, uint4 v) {
    *(uint4*)(tile + kr * 512 + ((((c >> 2) ^ (kr & 1))) << 6) + ((c & 3) << 4)) = v;
  }
  __device__ static __forceinline__ Frag frag(const char* tile, int col16, int kk, int lane) {
    const int k = kk * 4 + (lane >> 4);
    return *(const float*)(tile + k * 512 + ((((col16 >> 4) ^ (k & 1))) << 6) + ((lane & 15) << 2));
  }
};

// ------------------------------------------------------------------ cores
// Loader contract:  uint4 load(int slot)  -> 16 bytes of the CURRENT k-tile for
// this thread's slot (zero-filled when out of range);  void next()  -> advance
// one k-tile.  NT slots: row = (t>>3) + 32*slot, chunk = t&7.
// TN slots: k-row = t/CPR + (256/CPR)*slot, chunk = t%CPR.
// Epilogue contract: epi(m, n, f32x4 v): v[r] is C[m][n+r] (global indices).

// Register-staged cores (fp32 everywhere; bf16 for the element-gather "slow" loaders).  The bf16 vector paths of the
// convs and the distmat use the LDS-DMA cores further down.
// STAGES = 2: double-buffered LDS, one barrier per k-tile.  STAGES = 1: one LDS stage and two barriers per
// k-tile -- half the LDS, so twice the workgroups per CU; measured faster wherever it was tried (occupancy hides
// the load latency better than the second buffer does).
template <typename T, int BM, int BN, int STAGES, class LA, class LB, class Epi>
__device__ __forceinline__ void gemm_nt(LA& la, LB& lb, Epi& epi, int ktiles, int m0, int n0, char* smem) {
  typedef ImgNT<T> Img;
  constexpr int ACH = BM * 8 / 256, BCH = BN * 8 / 256;
  constexpr int FM = BM / 32, FN = BN / 32;
  constexpr int STAGE = (BM + BN) * 128;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
  const int srow = t >> 3, sc = t & 7;

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  uint4 ra[ACH], rb[BCH];
#pragma unroll
  for (int i = 0; i < ACH; ++i) ra[i] = la.load(i);
#pragma unroll
  for (int i = 0; i < BCH; ++i) rb[i] = lb.load(i);
#pragma unroll
  for (int i = 0; i < ACH; ++i) Img::store(smem, srow + 32 * i, sc, ra[i]);
#pragma unroll
  for (int i = 0; i < BCH; ++i) Img::store(smem + BM * 128, srow + 32 * i, sc, rb[i]);
  __syncthreads();

  for (int kt = 0; kt < ktiles; ++kt) {
    char* cur = STAGES == 1 ? smem : smem + (kt & 1) * STAGE;
    char* nxt = STAGES == 1 ? smem : smem + ((kt + 1) & 1) * STAGE;
    const bool has_next = (kt + 1) < ktiles;
    if (has_next) {
      la.next();
      lb.next();
#pragma unroll
      for (int i = 0; i < ACH; ++i) ra[i] = la.load(i);
#pragma unroll
      for (int i = 0; i < BCH; ++i) rb[i] = lb.load(i);
    }
    const char* At = cur + (wm * (BM / 2)) * 128;
    const char* Bt = cur + BM * 128 + (wn * (BN / 2)) * 128;
#pragma unroll
    for (int kk = 0; kk < Img::KSTEPS; ++kk) {
      typename Img::Frag fa[FM], fb[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) fa[i] = Img::frag(At, i * 16, kk, lane);
#pragma unroll
      for (int j = 0; j < FN; ++j) fb[j] = Img::frag(Bt, j * 16, kk, lane);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = mfma16(fb[j], fa[i], acc[i][j]);
    }
    if constexpr (STAGES == 1) __syncthreads();   // everyone is done reading the single stage before it is overwritten
    if (has_next) {
#pragma unroll
      for (int i = 0; i < ACH; ++i) Img::store(nxt, srow + 32 * i, sc, ra[i]);
#pragma unroll
      for (int i = 0; i < BCH; ++i) Img::store(nxt + BM * 128, srow + 32 * i, sc, rb[i]);
    }
    __syncthreads();
  }

  if constexpr (Epi::kStaged) {
    epi.template finish<BM, BN, FM, FN>(acc, smem, m0, n0);
  } else {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
        epi(m0 + wm * (BM / 2) + i * 16 + (lane & 15), n0 + wn * (BN / 2) + j * 16 + (lane >> 4) * 4, acc[i][j]);
  }
}

// TN core: 128x128 tile only.
template <typename T, int STAGES, class LA, class LB, class Epi>
__device__ __forceinline__ void gemm_tn(LA& la, LB& lb, Epi& epi, int ktiles, int m0, int n0, char* smem) {
  typedef ImgTN<T> Img;
  constexpr int CPR = Img::CPR;
  constexpr int RPP = 256 / CPR;                 // k-rows covered per pass of 256 threads
  constexpr int NCH = Img::BK / RPP;             // slots per thread per operand (=4)
  constexpr int TILE = Img::BK * 128 * (int)sizeof(T);
  constexpr int STAGE = 2 * TILE;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
  const int skr = t / CPR, sc = t % CPR;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  uint4 ra[NCH], rb[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) ra[i] = la.load(i);
#pragma unroll
  for (int i = 0; i < NCH; ++i) rb[i] = lb.load(i);
#pragma unroll
  for (int i = 0; i < NCH; ++i) Img::store(smem, skr + RPP * i, sc, ra[i]);
#pragma unroll
  for (int i = 0; i < NCH; ++i) Img::store(smem + TILE, skr + RPP * i, sc, rb[i]);
  __syncthreads();

  for (int kt = 0; kt < ktiles; ++kt) {
    char* cur = STAGES == 1 ? smem : smem + (kt & 1) * STAGE;
    char* nxt = STAGES == 1 ? smem : smem + ((kt + 1) & 1) * STAGE;
    const bool has_next = (kt + 1) < ktiles;
    if (has_next) {
      la.next();
      lb.next();
#pragma unroll
      for (int i = 0; i < NCH; ++i) ra[i] = la.load(i);
#pragma unroll
      for (int i = 0; i < NCH; ++i) rb[i] = lb.load(i);
    }
#pragma unroll
    for (int kk = 0; kk < Img::KSTEPS; ++kk) {
      typename Img::Frag fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = Img::frag(cur, wm * 64 + i * 16, kk, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = Img::frag(cur + TILE, wn * 64 + j * 16, kk, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(fb[j], fa[i], acc[i][j]);
    }
    if constexpr (STAGES == 1) __syncthreads();
    if (has_next) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) Img::store(nxt, skr + RPP * i, sc, ra[i]);
#pragma unroll
      for (int i = 0; i < NCH; ++i) Img::store(nxt + TILE, skr + RPP * i, sc, rb[i]);
    }
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      epi(m0 + wm * 64 + i * 16 + (lane & 15), n0 + wn * 64 + j * 16 + (lane >> 4) * 4, acc[i][j]);
}

// ------------------------------------------------------------------ LDS-DMA (bf16) cores
// ds_write_b128 moves only ~79 B/clk/CU, so register staging makes the LDS pipe (not the MFMA) the
// limiter of the bf16 tiles.  These variants stage both operands with global_load_lds_dwordx4
// (global -> LDS, no VGPR, no ds_write).  The DMA writes lane-linearly (wave-uniform base + lane*16),
// so the XOR swizzle moves to the SOURCE side: the thread whose LDS position is (row, slot f) fetches
// logical chunk f ^ key(row); fragment reads are unchanged.  Out-of-range chunks (conv padding, tile
// tails) are fetched from a 16-byte zero page.
__device__ __attribute__((aligned(16))) static const unsigned int g_zero_page[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ const void* zero_page() { return (const void*)g_zero_page; }

// Issued from inline asm on purpose: for an LDS-DMA issued through __builtin_amdgcn_global_load_lds hipcc waits
// vmcnt(0) before the next ds_read of ANY LDS address (it cannot tell the ring stages apart), which drains the
// whole ring every k-tile.  Hidden in asm, the DMA is ordered by the counted s_waitcnt vmcnt(N) + s_barrier of
// the cores below and nothing else.  M0 carries the wave-uniform LDS byte address; it is compiler-reserved, so
// it is saved and restored inside the statement.
__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
  const unsigned dst =
      __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_wave_base);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(dst)
               : "memory");
}

// Lean form for plain row-major operands (weights; the activations of 1x1 / stride-1 convs): the k-tile's global
// address is a WAVE-UNIFORM 64-bit base in SGPRs (advanced by one scalar add per k-tile) plus a per-lane 32-bit byte
// offset that never changes, so a whole tile is issued with ZERO vector instructions (the pointer form above costs ~8
// VALU per 16-byte slot: 64-bit select of the zero page + 64-bit add -- 63 VALU + 54 SALU per k-tile against 32 MFMAs
// in the 128x128 core).  N consecutive slots land 4096 LDS bytes apart (32 rows of 128 B): M0 steps by one s_add.
template <int N> __device__ __forceinline__ void glds16_lean(const unsigned (&voff)[N], const void* sbase, char* lds_wave_base);
#define IEEE_GLDS_STEP(v) "s_nop 0\n\tglobal_load_lds_dwordx4 " v ", %[b]\n\ts_add_u32 m0, m0, 0x1000\n\t"
template <> __device__ __forceinline__ void glds16_lean<2>(const unsigned (&voff)[2], const void* sbase, char* lds_wave_base) {
  const unsigned dst =
      __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_wave_base);
  unsigned keep;
  asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[d]\n\t" IEEE_GLDS_STEP("%[v0]") IEEE_GLDS_STEP("%[v1]") "s_mov_b32 m0, %[k]"
               : [k] "=&s"(keep)
               : [v0] "v"(voff[0]), [v1] "v"(voff[1]), [b] "s"(sbase), [d] "s"(dst)
               : "memory", "scc");
}
template <> __device__ __forceinline__ void glds16_lean<4>(const unsigned (&voff)[4], const void* sbase, char* lds_wave_base) {
  const unsigned dst =
      __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_wave_base);
  unsigned keep;
  asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[d]\n\t" IEEE_GLDS_STEP("%[v0]") IEEE_GLDS_STEP("%[v1]") IEEE_GLDS_STEP("%[v2]")
               IEEE_GLDS_STEP("%[v3]") "s_mov_b32 m0, %[k]"
               : [k] "=&s"(keep)
               : [v0] "v"(voff[0]), [v1] "v"(voff[1]), [v2] "v"(voff[2]), [v3] "v"(voff[3]), [b] "s"(sbase), [d] "s"(dst)
               : "memory", "scc");
}
template <> __device__ __forceinline__ void glds16_lean<8>(const unsigned (&voff)[8], const void* sbase, char* lds_wave_base) {
  const unsigned dst =
      __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_wave_base);
  unsigned keep;
  asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[d]\n\t" IEEE_GLDS_STEP("%[v0]") IEEE_GLDS_STEP("%[v1]") IEEE_GLDS_STEP("%[v2]")
               IEEE_GLDS_STEP("%[v3]") IEEE_GLDS_STEP("%[v4]") IEEE_GLDS_STEP("%[v5]") IEEE_GLDS_STEP("%[v6]") IEEE_GLDS_STEP("%[v7]")
               "s_mov_b32 m0, %[k]"
               : [k] "=&s"(keep)
               : [v0] "v"(voff[0]), [v1] "v"(voff[1]), [v2] "v"(voff[2]), [v3] "v"(voff[3]), [v4] "v"(voff[4]), [v5] "v"(voff[5]),
                 [v6] "v"(voff[6]), [v7] "v"(voff[7]), [b] "s"(sbase), [d] "s"(dst)
               : "memory", "scc");
}
#undef IEEE_GLDS_STEP

// One LDS-DMA (1 KB per wave-instruction) with a wave-uniform 64-bit global base in SGPRs + a per-lane 32-bit byte offset,
// to an arbitrary wave-uniform LDS address (the patch loader of conv3x3_patch_kernel: one image-row segment per instruction)
__device__ __forceinline__ void glds16_s(unsigned voff, const void* sbase, char* lds_wave_base) {
  const unsigned dst =
      __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_wave_base);
  unsigned keep;
  asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v], %[b]\n\ts_mov_b32 m0, %[k]"
               : [k] "=&s"(keep)
               : [v] "v"(voff), [b] "s"(sbase), [d] "s"(dst)
               : "memory");
}

// NT logical chunk of thread t (same for every slot: rows advance by 32, the key (row>>1)&7 does not change)
__device__ __forceinline__ int nt_dma_chunk(int t) { return (t & 7) ^ ((t >> 4) & 7); }
// TN logical chunk (bf16, 16 chunks per 256-B k-row): physical window (t&15)>>1, key h(k-row)
__device__ __forceinline__ int tn_dma_chunk(int t) {
  const int h = ((t >> 4) & 3) | (((t >> 7) & 1) << 2);
  return ((((t & 15) >> 1) ^ h) << 1) | (t & 1);
}

// Loader contract (DMA): const void* addr(int slot) -> global address of the 16 bytes that belong at this
// thread's LDS position (or zero_page()); void next().  Loaders are initialised with nt_dma_chunk /
// tn_dma_chunk as their column chunk.
//
// Pipeline: a ring of DMA_STAGES LDS stages, up to DMA_STAGES-1 k-tiles in flight.  Per k-tile ONE raw
// s_barrier, preceded by a COUNTED s_waitcnt vmcnt(N) that only retires the tile about to be read (the
// younger tiles stay in flight across the barrier; __syncthreads() would drain them).  The tile issued
// after the barrier overwrites the stage whose readers all passed that barrier.
constexpr int DMA_STAGES = 2;   // default ring depth (the cores take it as a template parameter)

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most `tiles_in_flight` younger tiles (PER glds each) are outstanding
template <int PER> __device__ __forceinline__ void wait_tiles(int tiles_in_flight) {
  if (tiles_in_flight >= 2) wait_vmcnt<2 * PER>();
  else if (tiles_in_flight == 1) wait_vmcnt<PER>();
  else wait_vmcnt<0>();
}

template <int BM, int BN, int STAGES, bool F16 = false, class LA, class LB, class Epi>
__device__ __forceinline__ void gemm_nt_dma(LA& la, LB& lb, Epi& epi, int ktiles, int m0, int n0, char* smem) {
  typedef ImgNT<bf16> Img;
  constexpr int DMA_STAGES = STAGES;
  constexpr int ACH = BM * 8 / 256, BCH = BN * 8 / 256;
  constexpr int FM = BM / 32, FN = BN / 32;
  constexpr int STAGE = (BM + BN) * 128;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto issue = [&](int stage_idx) {
    char* stage = smem + stage_idx * STAGE;
    if constexpr (LA::kLean) {
      glds16_lean<ACH>(la.off, la.base, stage + (8 * wave_u) * 128);
    } else {
#pragma unroll
      for (int i = 0; i < ACH; ++i) glds16(la.addr(i), stage + (32 * i + 8 * wave_u) * 128);
    }
    if constexpr (LB::kLean) {
      glds16_lean<BCH>(lb.off, lb.base, stage + BM * 128 + (8 * wave_u) * 128);
    } else {
#pragma unroll
      for (int i = 0; i < BCH; ++i) glds16(lb.addr(i), stage + BM * 128 + (32 * i + 8 * wave_u) * 128);
    }
  };
  auto compute = [&](const char* cur) {
    const char* At = cur + (wm * (BM / 2)) * 128;
    const char* Bt = cur + BM * 128 + (wn * (BN / 2)) * 128;
#pragma unroll
    for (int kk = 0; kk < Img::KSTEPS; ++kk) {
      Img::Frag fa[FM], fb[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) fa[i] = Img::frag(At, i * 16, kk, lane);
#pragma unroll
      for (int j = 0; j < FN; ++j) fb[j] = Img::frag(Bt, j * 16, kk, lane);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = mfma16_16bit<F16>(fb[j], fa[i], acc[i][j]);
    }
  };
  if constexpr (DMA_STAGES == 5) {
    // one LDS stage, but ALL fragments of the k-tile are pulled into registers first (64 VGPRs), so the stage is free
    // again before the MFMAs start and the next tile's DMA runs under this tile's 32 MFMAs.  Costs a workgroup per CU
    // (150-160 VGPRs -> 3) against PIPE 1.
    issue(0);
    for (int kt = 0; kt < ktiles; ++kt) {
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();   // tile kt has landed for every wave
      const char* At = smem + (wm * (BM / 2)) * 128;
      const char* Bt = smem + BM * 128 + (wn * (BN / 2)) * 128;
      Img::Frag fa[Img::KSTEPS][FM], fb[Img::KSTEPS][FN];
#pragma unroll
      for (int kk = 0; kk < Img::KSTEPS; ++kk) {
#pragma unroll
        for (int i = 0; i < FM; ++i) fa[kk][i] = Img::frag(At, i * 16, kk, lane);
#pragma unroll
        for (int j = 0; j < FN; ++j) fb[kk][j] = Img::frag(Bt, j * 16, kk, lane);
      }
      if (kt + 1 < ktiles) {
        __syncthreads();              // every wave holds its fragments (lgkmcnt(0)) -> the stage may be overwritten
        la.next();
        lb.next();
        issue(0);
      }
      __builtin_amdgcn_sched_barrier(0);   // the MFMAs stay behind the DMA issue ...
#pragma unroll
      for (int kk = 0; kk < Img::KSTEPS; ++kk)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[i][j] = mfma16_16bit<F16>(fb[kk][j], fa[kk][i], acc[i][j]);
      // ... and in front of the next k-tile's vmcnt(0) wait (hipcc otherwise sinks them below it: they touch no memory --
      // found in the ISA in round 3; until then this pipeline waited for its DMA with nothing running under it)
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) asm volatile("" : "+v"(acc[i][j]));
    }
  } else if constexpr (DMA_STAGES == 1) {
    // one LDS stage, nothing staged in registers: the fetch of the next tile is not overlapped inside the
    // workgroup at all -- the (small) register and LDS footprint buys a 4th workgroup per CU instead
    if constexpr (LA::kSkips) {
      // k-tiles the A loader declares dead for the whole workgroup (LoaderIm2colNT<.., true>) are stepped over: no
      // fetch, no MFMA; a tile with no live k-tile at all leaves the zero accumulators to the epilogue
      int kt = 0;
      while (kt < ktiles && la.dead()) { la.next(); lb.next(); ++kt; }
      if (kt < ktiles) issue(0);
      while (kt < ktiles) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        compute(smem);
        do { la.next(); lb.next(); ++kt; } while (kt < ktiles && la.dead());
        if (kt < ktiles) {
          __builtin_amdgcn_s_barrier();
          issue(0);
        }
      }
    } else {
    issue(0);
    for (int kt = 0; kt < ktiles; ++kt) {
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();   // tile kt has landed for every wave
      compute(smem);
      if (kt + 1 < ktiles) {
        __builtin_amdgcn_s_barrier();   // every wave is done reading the stage
        la.next();
        lb.next();
        issue(0);
      }
    }
    }
  } else if constexpr (DMA_STAGES == 6) {
    // "dual issue" (round 6): TWO k-tiles per round trip in two LDS stages, nothing overlapped inside the workgroup.  For the
    // launches that offer at most 3 workgroups per CU anyway (the 128 x 64 tiles of layer3's N = 256 GEMMs: 768 workgroups) the
    // single stage leaves 3 x 24 KB in flight per CU and pays one ~2 us round trip per 64 of K; this form has 3 x 48 KB in flight
    // and half the round trips and barriers.  (A two-stage RING has ONE tile in flight while it computes the other: the same
    // 24 KB per workgroup as the single stage.)
    int kt = 0;
    issue(0);
    if (ktiles > 1) { la.next(); lb.next(); issue(1); }
    while (kt < ktiles) {
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();   // both tiles have landed for every wave
      compute(smem);
      if (kt + 1 < ktiles) compute(smem + STAGE);
      kt += 2;
      if (kt < ktiles) {
        __builtin_amdgcn_s_barrier();   // every wave is done reading both stages
        la.next();
        lb.next();
        issue(0);
        if (kt + 1 < ktiles) { la.next(); lb.next(); issue(1); }
      }
    }
  } else {
  // prologue: tiles 0 .. DMA_STAGES-2
  int issued = 0;
  for (; issued < DMA_STAGES - 1 && issued < ktiles; ++issued) {
    if (issued > 0) { la.next(); lb.next(); }
    issue(issued);
  }
  for (int kt = 0; kt < ktiles; ++kt) {
    wait_tiles<ACH + BCH>(issued - kt - 1);   // tile kt has landed (for this wave's DMAs)
    __builtin_amdgcn_s_barrier();              // ... for every wave's; and stage (kt-1)%S is free
    if (issued < ktiles) {
      la.next();
      lb.next();
      issue(issued % DMA_STAGES);
      ++issued;
    }
    compute(smem + (kt % DMA_STAGES) * STAGE);
  }
  }

  if constexpr (Epi::kStaged) {
    epi.template finish<BM, BN, FM, FN>(acc, smem, m0, n0);
  } else {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
        epi(m0 + wm * (BM / 2) + i * 16 + (lane & 15), n0 + wn * (BN / 2) + j * 16 + (lane >> 4) * 4, acc[i][j]);
  }
}

// HALF_M: the GEMM has at most 64 rows (Cout = 64: stem, layer1 conv1/conv2).  The LDS image keeps its 128-column
// rows (columns >= 64 come from the zero page) but the four waves share the 64 valid rows (32 x 64 per wave), so
// half the MFMA work of the full tile disappears instead of multiplying zeros.
template <int STAGES, bool HALF_M, class LA, class LB, class Epi>
__device__ __forceinline__ void gemm_tn_dma(LA& la, LB& lb, Epi& epi, int ktiles, int m0, int n0, char* smem) {
  typedef ImgTN<bf16> Img;
  constexpr int DMA_STAGES = STAGES;
  constexpr int TILE = Img::BK * 128 * 2;   // 16 KB
  constexpr int STAGE = 2 * TILE;
  constexpr int NCH = 4;                    // 64 k-rows / (256 threads / 16 chunks per row)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);

  constexpr int FM = HALF_M ? 2 : 4, WROWS = FM * 16;
  f32x4 acc[FM][4];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto issue = [&](int stage_idx) {
    char* stage = smem + stage_idx * STAGE;
    if constexpr (LA::kLean) {
      glds16_lean<NCH>(la.off, la.base, stage + (4 * wave_u) * 256);
    } else {
#pragma unroll
      for (int i = 0; i < NCH; ++i) glds16(la.addr(i), stage + (16 * i + 4 * wave_u) * 256);
    }
    if constexpr (LB::kLean) {
      glds16_lean<NCH>(lb.off, lb.base, stage + TILE + (4 * wave_u) * 256);
    } else {
#pragma unroll
      for (int i = 0; i < NCH; ++i) glds16(lb.addr(i), stage + TILE + (16 * i + 4 * wave_u) * 256);
    }
  };
  auto compute = [&](const char* cur) {
#pragma unroll
    for (int kk = 0; kk < Img::KSTEPS; ++kk) {
      Img::Frag fa[FM], fb[4];
#pragma unroll
      for (int i = 0; i < FM; ++i) fa[i] = Img::frag(cur, wm * WROWS + i * 16, kk, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = Img::frag(cur + TILE, wn * 64 + j * 16, kk, lane);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(fb[j], fa[i], acc[i][j]);
    }
  };
  if constexpr (DMA_STAGES == 1) {   // see gemm_nt_dma
    if (ktiles > 0) issue(0);
    for (int kt = 0; kt < ktiles; ++kt) {
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      compute(smem);
      if (kt + 1 < ktiles) {
        __builtin_amdgcn_s_barrier();
        la.next();
        lb.next();
        issue(0);
      }
    }
  } else if constexpr (DMA_STAGES == 6) {
    // "dual issue" (see gemm_nt_dma): two k-tiles per round trip in two stages, nothing overlapped inside the workgroup.  The
    // weight-gradient launches have ~1.75 workgroups per CU by construction (split-K target 448): 56 KB in flight per CU with one
    // stage, 112 KB with this form.  IEEE_WGRAD_PIPE=6.
    int kt = 0;
    if (ktiles > 0) issue(0);
    if (ktiles > 1) { la.next(); lb.next(); issue(1); }
    while (kt < ktiles) {
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      compute(smem);
      if (kt + 1 < ktiles) compute(smem + STAGE);
      kt += 2;
      if (kt < ktiles) {
        __builtin_amdgcn_s_barrier();
        la.next();
        lb.next();
        issue(0);
        if (kt + 1 < ktiles) { la.next(); lb.next(); issue(1); }
      }
    }
  } else {
  int issued = 0;
  for (; issued < DMA_STAGES - 1 && issued < ktiles; ++issued) {
    if (issued > 0) { la.next(); lb.next(); }
    issue(issued);
  }
  for (int kt = 0; kt < ktiles; ++kt) {
    wait_tiles<2 * NCH>(issued - kt - 1);
    __builtin_amdgcn_s_barrier();
    if (issued < ktiles) {
      la.next();
      lb.next();
      issue(issued % DMA_STAGES);
      ++issued;
    }
    compute(smem + (kt % DMA_STAGES) * STAGE);
  }
  }

  if constexpr (Epi::kStaged) {
    epi.template finish<WROWS, FM>(acc, smem, m0, n0);
  } else {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        epi(m0 + wm * WROWS + i * 16 + (lane & 15), n0 + wn * 64 + j * 16 + (lane >> 4) * 4, acc[i][j]);
  }
}

// ------------------------------------------------------------------ loaders
// Plain row-major [rows][ld] matrix, K contiguous (weights, features, linear inputs).
// bf16 rows of a plain [rows][ld] matrix for the lean DMA issue (glds16_lean): rows past the end are CLAMPED to the
// last row instead of zero-filled -- their products only reach output rows / columns that no epilogue stores or sums
template <int NCH, int PASS = 32> struct LoaderPlainLean {   // PASS = rows per pass of the block (threads / 8)
  static constexpr bool kLean = true;
  static constexpr bool kSkips = false;
  const char* base;          // wave-uniform
  unsigned off[NCH];
  __device__ __forceinline__ void init(const bf16* mat, int64_t ld, int row0, int nrows, int chunk) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int row = min(row0 + (t >> 3) + PASS * i, nrows - 1);
      off[i] = (unsigned)(((int64_t)row * ld + chunk * 8) * 2);
    }
    base = (const char*)mat;
  }
  __device__ __forceinline__ void next() { base += 128; }
};

template <typename T, int NCH> struct LoaderPlainNT {
  static constexpr bool kLean = false;
  static constexpr bool kSkips = false;
  static constexpr int VEC = 16 / sizeof(T);
  static constexpr int BK = ImgNT<T>::BK;
  const T* p[NCH];
  int kcol, K;
  __device__ __forceinline__ void init(const T* base, int64_t ld, int row0, int nrows, int K_, int chunk = -1) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int row = row0 + (t >> 3) + 32 * i;
      p[i] = row < nrows ? base + (int64_t)row * ld : nullptr;
    }
    kcol = (chunk < 0 ? (t & 7) : chunk) * VEC;
    K = K_;
  }
  __device__ __forceinline__ uint4 load(int i) const {
    if (p[i] == nullptr || kcol >= K) return make_uint4(0, 0, 0, 0);
    return *(const uint4*)(p[i] + kcol);
  }
  __device__ __forceinline__ const void* addr(int i) const {
    return (p[i] == nullptr || kcol >= K) ? zero_page() : (const void*)(p[i] + kcol);
  }
  __device__ __forceinline__ void next() { kcol += BK; }
};

// Gather geometry shared by conv forward and dgrad:
//   source row  h = (hb + sgn*r) / div  where hb = p*mul + off   (same for w)
// forward: mul=stride, off=-pad, sgn=+1, div=1, source = X   [N,Hs,Ws,Cs]
// dgrad  : mul=1, off=+pad, sgn=-1, div=stride, source = dY  [N,Hs,Ws,Cs]
struct GatherGeom {
  int Hs, Ws, Cs;      // source tensor spatial dims / channels
  int Ho, Wo;          // pixel grid that indexes the GEMM rows
  int R, S;            // taps
  int mul, off, sgn, div;
  int npix;            // N*Ho*Wo
  int perm = 0;        // stride-2 dgrad: GEMM rows in parity-class-major order (LoaderIm2colNT<.., true>)
};

// Fast path: either Cs % BK == 0 (a k-tile stays inside one tap) or BK % Cs == 0 and the BK/Cs taps of a k-tile are
// consecutive horizontal taps: part of one filter row (S % (BK/Cs) == 0: the 8-channel padded stem) or whole rows
// ((BK/Cs) % S == 0, R % ((BK/Cs)/S) == 0: the 4-channel stem, whose 16-byte chunk is TWO horizontally adjacent
// pixels -- that form is only used without padding, so both taps of a chunk are always in range).
// All per-pixel work is done once: each slot keeps a 32-bit element offset of its (un-tapped) source pixel
// and a bit mask of the taps that fall inside the image; per k-tile the tap contributes one wave-uniform
// offset, so a load costs a shift/test and one add (the 64-bit multiplies of a per-tile decode used to
// rival the MFMA time of the tile).  Requires R*S <= 64 and < 2^31 elements per modality tensor.
// SKIP (the dgrad of a stride-2 conv, g.div == 2, g.perm): an input pixel of row / column parity (ph, pw) receives only the
// taps of matching parity -- 1, 2, 2 or 4 of 9 (one or none of a 1x1's) -- so the GEMM rows are taken in parity-class-major
// order, m -> (class, image, i, j) -> pixel (2i + ph, 2j + pw); a 128-row tile then lies inside one class of one image
// (the launcher checks (Ho/2)*(Wo/2) % 128 == 0) and the k-tiles of its dead taps are skipped by the whole workgroup
// (`dead()`, see gemm_nt_dma) instead of being fetched as zeros and multiplied: 2.25 of 9 taps of work on average.
template <typename T, int NCH, bool SKIP = false> struct LoaderIm2colNT {
  static constexpr bool kLean = false;
  static constexpr bool kSkips = SKIP;
  unsigned long long live = ~0ull;   // SKIP: taps whose parity matches this tile's class (workgroup-uniform)
  __device__ __forceinline__ bool dead() const { return !((live >> tap) & 1ull); }
  static constexpr int VEC = 16 / sizeof(T);
  static constexpr int BK = ImgNT<T>::BK;
  const T* src;
  int off[NCH];                 // element offset of pixel (hb, wb) [halved for div 2] + this thread's chunk
  unsigned long long vm[NCH];   // bit (r*S + s): tap (r, s) is in range (and parity-aligned for div 2)
  int r, s, ci0, tap;           // current tap (wave-uniform)
  int toff;                     // wave-uniform element offset of the current tap
  int ds, tpt;
  GatherGeom g;
  __device__ __forceinline__ int tap_off(int rr, int ss) const {
    const int re = g.div == 2 ? (rr >> 1) : rr, se = g.div == 2 ? (ss >> 1) : ss;
    return g.sgn * (re * g.Ws + se) * g.Cs;
  }
  __device__ __forceinline__ void init(const T* src_, const GatherGeom& g_, int m0, int chunk = -1) {
    g = g_;
    src = src_;
    const int t = threadIdx.x;
    const int hw = g.Ho * g.Wo;
    const int ch = chunk < 0 ? (t & 7) : chunk;
    int coff = ch * VEC;
    ds = 0;
    tpt = 1;
    int dr = 0;      // this chunk's tap offset inside the k-tile: dr filter rows down, ds taps right
    if (g.Cs < BK) {
      tpt = BK / g.Cs;
      ds = coff / g.Cs;
      coff -= ds * g.Cs;
      if (tpt > g.S) { dr = ds / g.S; ds -= dr * g.S; }
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int m = m0 + (t >> 3) + 32 * i;
      off[i] = 0;
      vm[i] = 0ull;
      if (m < g.npix) {
        int n, pp, qq;
        if constexpr (SKIP) {
          const int wc = g.Wo >> 1, per_img = (g.Ho >> 1) * wc, per_cls = (g.npix / hw) * per_img;
          const int cls = m / per_cls, rc = m - cls * per_cls;
          n = rc / per_img;
          const int q = rc - n * per_img, ii = q / wc;
          pp = 2 * ii + (cls >> 1);
          qq = 2 * (q - ii * wc) + (cls & 1);
        } else {
          n = m / hw;
          const int rem = m - n * hw;
          pp = rem / g.Wo;
          qq = rem - pp * g.Wo;
        }
        const int hb = pp * g.mul + g.off + g.sgn * dr, wb = qq * g.mul + g.off + g.sgn * ds;
        const int he = g.div == 2 ? (hb >> 1) : hb, we = g.div == 2 ? (wb >> 1) : wb;
        off[i] = ((n * g.Hs + he) * g.Ws + we) * g.Cs + coff;
        // the valid taps are (row condition) x (column condition): R + S tests instead of R*S (the 8x8 stem would
        // otherwise spend more time here than in its four k-tiles)
        unsigned long long rowpat = 0ull;
        for (int ss = 0; ss < g.S; ++ss) {
          int w = wb + g.sgn * ss;
          bool ok = true;
          if (g.div == 2) { ok = !(w & 1); w >>= 1; }
          if (ok && (unsigned)w < (unsigned)g.Ws) rowpat |= 1ull << ss;
        }
        for (int rr = 0; rr < g.R; ++rr) {
          int h = hb + g.sgn * rr;
          bool ok = true;
          if (g.div == 2) { ok = !(h & 1); h >>= 1; }
          if (ok && (unsigned)h < (unsigned)g.Hs) vm[i] |= rowpat << (rr * g.S);
        }
      }
    }
    r = 0; s = 0; ci0 = 0; tap = 0; toff = 0;
    if constexpr (SKIP) {
      const int per_cls = (g.npix / hw) * (g.Ho >> 1) * (g.Wo >> 1);
      const int cls = __builtin_amdgcn_readfirstlane(m0 / per_cls), ph = cls >> 1, pw = cls & 1;
      live = 0ull;
      for (int rr = 0; rr < g.R; ++rr)
        for (int ss = 0; ss < g.S; ++ss)
          if (!((ph * g.mul + g.off + g.sgn * rr) & 1) && !((pw * g.mul + g.off + g.sgn * ss) & 1)) live |= 1ull << (rr * g.S + ss);
    }
  }
  __device__ __forceinline__ const void* addr(int i) const {
    if (!((vm[i] >> tap) & 1ull)) return zero_page();
    return (const void*)(src + (off[i] + toff + ci0));
  }
  __device__ __forceinline__ uint4 load(int i) const {
    if (!((vm[i] >> tap) & 1ull)) return make_uint4(0, 0, 0, 0);
    return *(const uint4*)(src + (off[i] + toff + ci0));
  }
  __device__ __forceinline__ void next() {
    if (g.Cs < BK) {
      s += tpt;
      while (s >= g.S) { s -= g.S; ++r; }
    } else {
      ci0 += BK;
      if (ci0 < g.Cs) return;
      ci0 = 0;
      if (++s == g.S) { s = 0; ++r; }
    }
    tap = r < g.R ? r * g.S + s : 63;     // past the last tap: bit 63 is never set (R*S <= 56 on this path)
    toff = tap_off(r, s);
  }
};

// Generic (slow) path: any Cs (the 3-channel stem); element-wise gather.
template <typename T, int NCH> struct LoaderIm2colSlowNT {
  static constexpr bool kLean = false;
  static constexpr bool kSkips = false;
  static constexpr int VEC = 16 / sizeof(T);
  static constexpr int BK = ImgNT<T>::BK;
  const T* p[NCH];
  int hb[NCH], wb[NCH];
  int kcol, K;
  GatherGeom g;
  __device__ __forceinline__ void init(const T* src, const GatherGeom& g_, int m0) {
    g = g_;
    const int t = threadIdx.x;
    const int hw = g.Ho * g.Wo;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int m = m0 + (t >> 3) + 32 * i;
      if (m < g.npix) {
        const int n = m / hw, rem = m - n * hw;
        const int pp = rem / g.Wo, qq = rem - pp * g.Wo;
        p[i] = src + (int64_t)n * g.Hs * g.Ws * g.Cs;
        hb[i] = pp * g.mul + g.off;
        wb[i] = qq * g.mul + g.off;
      } else {
        p[i] = nullptr; hb[i] = 0; wb[i] = 0;
      }
    }
    kcol = (t & 7) * VEC;
    K = g.R * g.S * g.Cs;
  }
  __device__ __forceinline__ uint4 load(int i) const {
    float f[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const int k = kcol + e;
      float v = 0.f;
      if (p[i] != nullptr && k < K) {
        const int tap = k / g.Cs, ci = k - tap * g.Cs;
        const int rr = tap / g.S, ss = tap - rr * g.S;
        int h = hb[i] + g.sgn * rr, w = wb[i] + g.sgn * ss;
        bool ok = true;
        if (g.div == 2) { ok = !((h | w) & 1); h >>= 1; w >>= 1; }
        if (ok && (unsigned)h < (unsigned)g.Hs && (unsigned)w < (unsigned)g.Ws)
          v = to_f32(p[i][((int64_t)h * g.Ws + w) * g.Cs + ci]);
      }
      f[e] = v;
    }
    return Vec16<T>::pack(f);
  }
  __device__ __forceinline__ void next() { kcol += BK; }
};

// TN: plain [K][ld] matrix, columns contiguous (dY for wgrad).
// bf16 k-rows of a plain [K][ld] matrix for the lean DMA issue (glds16_lean): 16 k-rows per pass, 4 passes per tile.
// Only for k ranges that are whole tiles (no zero fill: the caller checks (kend - kbeg) % 64 == 0); column chunks past
// the end are clamped to the last one (they only feed output rows / columns that are never stored).
struct LoaderColsLean {
  static constexpr bool kLean = true;
  static constexpr bool kSkips = false;
  const char* base;          // wave-uniform: first k-row of the current tile
  unsigned off[4];
  int64_t step;              // bytes per k-tile
  __device__ __forceinline__ void init(const bf16* mat, int64_t ld, int col0, int ncols, int kbeg, int chunk) {
    const int t = threadIdx.x;
    const int col = min(col0 + chunk * 8, ncols - 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) off[i] = (unsigned)((((int64_t)(t >> 4) + 16 * i) * ld + col) * 2);
    base = (const char*)(mat + (int64_t)kbeg * ld);
    step = 64 * ld * 2;
  }
  __device__ __forceinline__ void next() { base += step; }
};

template <typename T> struct LoaderColsTN {
  static constexpr bool kLean = false;
  static constexpr bool kSkips = false;
  static constexpr int VEC = 16 / sizeof(T);
  static constexpr int BK = ImgTN<T>::BK, CPR = ImgTN<T>::CPR, RPP = 256 / CPR, NCH = BK / RPP;
  const T* base;  // already offset to this thread's column chunk, or nullptr if the chunk is out of range
  int64_t ld;
  int k0, kend;
  __device__ __forceinline__ void init(const T* mat, int64_t ld_, int col0, int ncols, int kbeg, int kend_,
                                       int chunk = -1) {
    const int t = threadIdx.x;
    const int col = col0 + (chunk < 0 ? (t % CPR) : chunk) * VEC;
    base = col < ncols ? mat + col : nullptr;
    ld = ld_;
    k0 = kbeg + t / CPR;
    kend = kend_;
  }
  __device__ __forceinline__ uint4 load(int i) const {
    const int k = k0 + RPP * i;
    if (base == nullptr || k >= kend) return make_uint4(0, 0, 0, 0);
    return *(const uint4*)(base + (int64_t)k * ld);
  }
  __device__ __forceinline__ const void* addr(int i) const {
    const int k = k0 + RPP * i;
    return (base == nullptr || k >= kend) ? zero_page() : (const void*)(base + (int64_t)k * ld);
  }
  __device__ __forceinline__ void next() { k0 += BK; }
};

// TN: im2col columns (tap, channel) of the forward geometry, rows = output pixels.
template <typename T> struct LoaderIm2colTN {
  static constexpr bool kLean = false;
  static constexpr bool kSkips = false;
  static constexpr int VEC = 16 / sizeof(T);
  static constexpr int BK = ImgTN<T>::BK, CPR = ImgTN<T>::CPR, RPP = 256 / CPR, NCH = BK / RPP;
  const T* src;
  GatherGeom g;
  int k0, kend;
  int r, s, ci;       // this thread's (fixed) column chunk
  bool colok;
  int lw, lhw;        // log2(Wo), log2(Ho*Wo) or -1
  __device__ __forceinline__ void init(const T* src_, const GatherGeom& g_, int col0, int kbeg, int kend_,
                                       int chunk = -1) {
    g = g_;
    src = src_;
    const int t = threadIdx.x;
    const int tc = col0 + (chunk < 0 ? (t % CPR) : chunk) * VEC;
    const int ncols = g.R * g.S * g.Cs;
    colok = tc < ncols;
    const int tap = tc / g.Cs;
    ci = tc - tap * g.Cs;
    r = tap / g.S;
    s = tap - r * g.S;
    k0 = kbeg + t / CPR;
    kend = kend_;
    const int hw = g.Ho * g.Wo;
    lw = (g.Wo & (g.Wo - 1)) == 0 ? __builtin_ctz(g.Wo) : -1;
    lhw = (hw & (hw - 1)) == 0 ? __builtin_ctz(hw) : -1;
  }
  __device__ __forceinline__ uint4 load(int i) const {
    const void* a = addr(i);
    return a == zero_page() ? make_uint4(0, 0, 0, 0) : *(const uint4*)a;
  }
  __device__ __forceinline__ const void* addr(int i) const {
    const int m = k0 + RPP * i;
    if (!colok || m >= kend) return zero_page();
    int n, pp, qq;
    if (lw >= 0 && lhw >= 0) {
      n = m >> lhw;
      const int rem = m & ((1 << lhw) - 1);
      pp = rem >> lw;
      qq = rem & ((1 << lw) - 1);
    } else {
      const int hw = g.Ho * g.Wo;
      n = m / hw;
      const int rem = m - n * hw;
      pp = rem / g.Wo;
      qq = rem - pp * g.Wo;
    }
    const int h = pp * g.mul + g.off + r, w = qq * g.mul + g.off + s;
    if ((unsigned)h >= (unsigned)g.Hs || (unsigned)w >= (unsigned)g.Ws) return zero_page();
    // 32-bit element offset with 24-bit multiplies (every factor < 2^24; tensors hold < 2^31 elements)
    const int o = __mul24(__mul24(n, g.Hs) + h, g.Ws) + w;
    return (const void*)(src + (__mul24(o, g.Cs) + ci));
  }
  __device__ __forceinline__ void next() { k0 += BK; }
};

// TN slow path (stem): per-element column decode.
template <typename T> struct LoaderIm2colSlowTN {
  static constexpr bool kLean = false;
  static constexpr bool kSkips = false;
  static constexpr int VEC = 16 / sizeof(T);
  static constexpr int BK = ImgTN<T>::BK, CPR = ImgTN<T>::CPR, RPP = 256 / CPR, NCH = BK / RPP;
  const T* src;
  GatherGeom g;
  int k0, kend, tc0, ncols;
  __device__ __forceinline__ void init(const T* src_, const GatherGeom& g_, int col0, int kbeg, int kend_) {
    g = g_;
    src = src_;
    const int t = threadIdx.x;
    tc0 = col0 + (t % CPR) * VEC;
    ncols = g.R * g.S * g.Cs;
    k0 = kbeg + t / CPR;
    kend = kend_;
  }
  __device__ __forceinline__ uint4 load(int i) const {
    const int m = k0 + RPP * i;
    float f[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) f[e] = 0.f;
    if (m < kend && tc0 < ncols) {
      const int hw = g.Ho * g.Wo;
      const int n = m / hw, rem = m - n * hw;
      const int pp = rem / g.Wo, qq = rem - pp * g.Wo;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const int tc = tc0 + e;
        if (tc < ncols) {
          const int tap = tc / g.Cs, ci = tc - tap * g.Cs;
          const int rr = tap / g.S, ss = tap - rr * g.S;
          const int h = pp * g.mul + g.off + rr, w = qq * g.mul + g.off + ss;
          if ((unsigned)h < (unsigned)g.Hs && (unsigned)w < (unsigned)g.Ws)
            f[e] = to_f32(src[(((int64_t)n * g.Hs + h) * g.Ws + w) * g.Cs + ci]);
        }
      }
    }
    return Vec16<T>::pack(f);
  }
  __device__ __forceinline__ void next() { k0 += BK; }
};

}  // namespace ieee
