/*
 * ieee_amd — C ABI of the MI355X (gfx950) implementation of the IEEE3modalPart
 * hot path (ziwang1121/IEEE).  This is the drop-in boundary: plain pointers and
 * sizes, no torch / C++ types.  Every entry point returns an int status
 * (IEEE_OK == 0, negative on error; text via ieee_last_error()), never throws,
 * never allocates or frees caller-visible memory, never synchronises the host
 * unless stated, and enqueues all work on the caller's hipStream_t (passed as
 * void*; NULL = default stream).  All data pointers are DEVICE pointers unless
 * a parameter is documented as host.
 *
 * The reference has no FFI of its own (it is pure PyTorch + one disabled Cython
 * module), so each entry point cites the reference Python function whose
 * arithmetic it replaces (file:line under the reference repo root).  The
 * ctypes binding a maintainer would add is shown in INTEGRATION.md and lives in
 * ieee_amd/_lib.py.
 *
 * Layouts: activations NHWC; modality-batched tensors carry a leading
 * modality axis of size 3 in the order [RGB, NI, TI]
 * (reference torchreid/data/datasets/dataset.py:338-340).
 */
#ifndef IEEE_AMD_H
#define IEEE_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IEEE_OK 0
#define IEEE_ERR_BAD_ARG (-1)
#define IEEE_ERR_HIP (-2)
#define IEEE_ERR_UNSUPPORTED (-3)
#define IEEE_ERR_NO_VALID_QUERY (-4) /* rank.py:165 "all query identities do not appear in gallery" */

#define IEEE_F32 0
#define IEEE_BF16 1

/* ---- library ---------------------------------------------------------- */
const char* ieee_last_error(void); /* thread-local text of the last failure */
int ieee_version(void);            /* ABI version, currently 1 */
/* 1 if the current HIP device is gfx950, 0 otherwise (host query) */
int ieee_device_is_gfx950(void);

/* ---- evaluator: distance matrix ---------------------------------------- */
/* torchreid/metrics/distance.py:49-64 euclidean_squared_distance:
 *   out[i][j] = |q_i|^2 + |g_j|^2 - 2 q_i.g_j      (squared, may be slightly < 0)
 * q [m][d], g [n][d] row-major, dtype IEEE_F32 (exact fp32 MFMA, k-ordered fma
 * chain) or IEEE_BF16 (inputs already bf16; fp32 accumulate).  out [m][n] fp32,
 * row stride ldo elements.  d % 8 == 0.  work: >= (m+n)*4 bytes of scratch
 * (row norms).  metric: 0 euclidean, 1 cosine (distance.py:67-80: rows are
 * L2-normalised with eps 1e-12 and out = 1 - q^.g^). */
int ieee_sqeuclid_distmat(const void* q, const void* g, int64_t m, int64_t n, int64_t d, int dtype,
                          int metric, float* out, int64_t ldo, void* work, void* stream);

/* ---- evaluator: CMC / mAP ----------------------------------------------- */
/* torchreid/metrics/rank.py:103-171 eval_market1501 (and the disabled native
 * rank_cylib/rank_cy.pyx:156-243).  distmat [num_q][num_g] fp32 (row stride
 * ldd), pids / camids int32 device arrays.  Tie order is (distance, gallery
 * index) ascending.  Outputs (device):
 *   ap        [num_q] float64   average precision per query (-1 when skipped)
 *   first_pos [num_q] int32     0-based rank of the first true match among kept
 *   summary   [max_rank+2] int64: cmc hit counts per rank, then num_valid_q,
 *             then (bit pattern of) the float64 sum of AP over valid queries
 * The host shim forms cmc = float32(counts)/float32(num_valid) and
 * mAP = ap_sum/num_valid as rank.py:167-169 does.  No host sync inside. */
int ieee_rank_market1501(const float* distmat, int64_t ldd, int64_t num_q, int64_t num_g,
                         const int32_t* q_pids, const int32_t* g_pids, const int32_t* q_camids,
                         const int32_t* g_camids, int64_t max_rank, double* ap, int32_t* first_pos,
                         int64_t* summary, void* stream);

/* ---- convolution as implicit GEMM over NHWC (MFMA) ------------------------ */
/* These replace torch's conv2d forward / backward as dispatched by the
 * reference's Bottleneck.forward (torchreid/models/resnet.py:164-184), the
 * ResNetIEEE stem (:496-501, :622-631) and the CIM 1x1 convs
 * (torchreid/models/ieee3modalPart.py:28-48, 427-435).  `groups` independent
 * problems (the three modality streams) run in one launch; *_gs are the
 * per-group strides in ELEMENTS.  dtype IEEE_F32 (exact fp32 MFMA, parity
 * mode) or IEEE_BF16 (bf16 storage, fp32 accumulate).  Activations NHWC. */

/* row length (elements) of a packed weight matrix: R*S*inner rounded up to the k-tile */
int64_t ieee_conv_packed_ld(int dtype, int64_t inner_channels, int64_t R, int64_t S);

/* fp32 OIHW parameters (the reference's state_dict layout) -> GEMM operand.
 * mode 0: forward  dst[co][(r*S+s)*Ci+ci]; mode 1: dgrad dst[ci][(r*S+s)*Co+co] */
int ieee_pack_conv_weight(const float* w_oihw, void* dst, int dtype, int mode, int64_t groups, int64_t Co,
                          int64_t Ci, int64_t R, int64_t S, int64_t w_gs, int64_t dst_gs, void* stream);

/* y[N,Ho,Wo,Co] = conv(x[N,Hi,Wi,Ci], w), no bias (every conv on the path is bias-free) */
int ieee_conv2d_fwd(const void* x, const void* w_packed, void* y, int dtype, int64_t groups, int64_t N,
                    int64_t Hi, int64_t Wi, int64_t Ci, int64_t Co, int64_t R, int64_t S, int64_t stride,
                    int64_t pad, int64_t x_gs, int64_t w_gs, int64_t y_gs, void* stream);

/* dx[N,Hi,Wi,Ci] = conv_transpose(dy) (+ addend, same layout as dx, may be NULL): the
 * residual-branch gradient of Bottleneck (resnet.py:181 `out += identity`) is folded in here */
int ieee_conv2d_dgrad(const void* dy, const void* w_packed_d, void* dx, const void* addend, int dtype,
                      int64_t groups, int64_t N, int64_t Hi, int64_t Wi, int64_t Ci, int64_t Co, int64_t R,
                      int64_t S, int64_t stride, int64_t pad, int64_t dy_gs, int64_t w_gs, int64_t dx_gs,
                      void* stream);

/* dw (fp32, OIHW, the layout of param.grad) = or += sum over pixels; deterministic split-K:
 * partial slabs in `work` (size from the query below) are reduced in a fixed order */
int64_t ieee_conv2d_wgrad_workspace_bytes(int dtype, int64_t groups, int64_t N, int64_t Ho, int64_t Wo,
                                          int64_t Ci, int64_t Co, int64_t R, int64_t S);
int ieee_conv2d_wgrad(const void* dy, const void* x, float* dw_oihw, void* work, int dtype, int64_t groups,
                      int64_t N, int64_t Hi, int64_t Wi, int64_t Ci, int64_t Co, int64_t R, int64_t S,
                      int64_t stride, int64_t pad, int64_t dy_gs, int64_t x_gs, int64_t dw_gs, int accumulate,
                      void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IEEE_AMD_H */
