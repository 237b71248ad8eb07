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

#define IEEE_MAX_GROUPS 18 /* pointer-table capacity of the grouped head kernels (3 modalities x 6 parts) */

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
/* the same matrix from fp32 rows on the 16-bit matrix cores: every value is split into 16-bit pieces that sum to it
 * and q.g becomes ONE 16-bit GEMM over the piece products, smallest first, accumulated in fp32 (each product is exact
 * there).  scheme:
 *   IEEE_SPLIT_BF16X3  three bf16 pieces (all 24 mantissa bits), six products hi.hi hi.mid mid.hi mid.mid hi.lo lo.hi;
 *                      what is dropped is below 2^-24 of the result
 *   IEEE_SPLIT_F16X2   two fp16 pieces (22 bits) of the row scaled by a power of two (undone in the epilogue), three
 *                      products hi.hi hi.lo lo.hi: 2^-22, half the matrix work of BF16X3
 *   IEEE_SPLIT_BF16X2  two bf16 pieces, three products: ~2^-16
 * Row norms come from the fp32 rows as above.  work: ieee_sqeuclid_distmat_split_workspace_bytes() bytes. */
#define IEEE_SPLIT_BF16X3 6
#define IEEE_SPLIT_BF16X2 3
#define IEEE_SPLIT_F16X2 2
int64_t ieee_sqeuclid_distmat_split_workspace_bytes(int64_t m, int64_t n, int64_t d, int64_t scheme);
int ieee_sqeuclid_distmat_split(const float* q, const float* g, int64_t m, int64_t n, int64_t d, int64_t scheme,
                                int metric, float* out, int64_t ldo, void* work, int64_t work_bytes, void* stream);

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
/* The same evaluation with a workspace of ieee_rank_workspace_bytes(num_g) bytes (device, 4-byte aligned): the gallery is
 * bucketed by identity once per call, so a query collects its true matches and its same-camera entries from one bucket
 * instead of scanning all num_g identities.  Same results, bit for bit. */
int64_t ieee_rank_workspace_bytes(int64_t num_g);
int ieee_rank_market1501_ws(const float* distmat, int64_t ldd, int64_t num_q, int64_t num_g,
                            const int32_t* q_pids, const int32_t* g_pids, const int32_t* q_camids,
                            const int32_t* g_camids, int64_t max_rank, double* ap, int32_t* first_pos,
                            int64_t* summary, void* work, int64_t work_bytes, void* stream);

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

/* same, zero-padding the input-channel, kernel-height and kernel-width axes to (Ci, R, S) >= (Ci_src, R_src, S_src);
 * and its inverse for the weight gradient (dw_padded [Co][Ci][R][S] -> dw [Co][Ci_src][R_src][S_src]) */
int ieee_pack_conv_weight_padded(const float* w_oihw, void* dst, int dtype, int mode, int64_t groups, int64_t Co,
                                 int64_t Ci_src, int64_t R_src, int64_t S_src, int64_t Ci, int64_t R, int64_t S,
                                 int64_t w_gs, int64_t dst_gs, void* stream);
int ieee_unpad_weight_grad(const float* dw_padded, float* dw, int64_t groups, int64_t Co, int64_t Ci, int64_t R,
                           int64_t S, int64_t Ci_src, int64_t R_src, int64_t S_src, int64_t dwp_gs, int64_t dw_gs,
                           int accumulate, void* stream);

/* every conv weight of the network in ONE launch: `descs` is a device array of packing descriptors built by
 * the executor (ieee_pack_desc_bytes() each); offsets are elements relative to `params` / `ws_base` */
int64_t ieee_pack_desc_bytes(void);
int ieee_pack_all_weights(const float* params, void* ws_base, const void* descs, int64_t ndesc,
                          int64_t total_blocks, int dtype, void* stream);

/* y[N,Ho,Wo,Co] = conv(x[N,Hi,Wi,Ci], w), no bias (every conv on the path is bias-free) */
int ieee_conv2d_fwd(const void* x, const void* w_packed, void* y, int dtype, int64_t groups, int64_t N,
                    int64_t Hi, int64_t Wi, int64_t Ci, int64_t Co, int64_t R, int64_t S, int64_t stride,
                    int64_t pad, int64_t x_gs, int64_t w_gs, int64_t y_gs, float* bn_partial, void* stream);
/* bn_partial (bf16 vector path only, else NULL): the conv also emits, per group, [2][Co][rblocks] per-channel
 * sum / sum-of-squares of the stored outputs, rblocks = ieee_conv2d_fwd_stats_rblocks(); hand the same buffer
 * and rblocks to ieee_bn2d_fwd(stats_rblocks) and the separate statistics pass disappears */
int64_t ieee_conv2d_fwd_stats_rblocks(int64_t N, int64_t Ho, int64_t Wo);
/* Training forward with the consumer BatchNorm's FINALIZE fused in (bf16, at most ieee_conv2d_fwd_bn_train_max_rows() output
 * pixels per group): besides y and the per-tile partial sums, the launch leaves that BatchNorm's stats [groups][4][Co] (mean,
 * invstd, scale, shift) and updates the running statistics (running_* may be NULL) -- what ieee_bn2d_fwd's finalize step
 * computes (resnet.py:164-184 under nn.BatchNorm2d in train mode); follow with ieee_bn2d_fwd(..., stats_rblocks = -1) for the
 * apply pass only.  `tickets`: groups x ceil(Co / 64) int32, ZERO on entry (the launch leaves them zero again).  The
 * workgroup that arrives last at a column block's ticket reduces that block's partials in tile order (deterministic). */
int64_t ieee_conv2d_fwd_bn_train_max_rows(void);
int ieee_conv2d_fwd_bn_train(const void* x, const void* w_packed, void* y, int dtype, int64_t groups, int64_t N, int64_t Hi,
                             int64_t Wi, int64_t Ci, int64_t Co, int64_t R, int64_t S, int64_t stride, int64_t pad,
                             int64_t x_gs, int64_t w_gs, int64_t y_gs, float* bn_partial, const float* gamma,
                             const float* beta, int64_t param_gs, float* running_mean, float* running_var, int64_t buf_gs,
                             float* stats, float momentum, float eps, int32_t* tickets, void* stream);
/* inference: conv + eval-mode BatchNorm (+ residual) (+ ReLU) in one launch -- Bottleneck.forward / the stem in eval
 * mode (resnet.py:164-184, 622-626).  bn_stats: that BN's [groups][4][Co] statistics as ieee_bn2d_fwd(training = 0,
 * out = NULL) leaves them (scale at 2*Co, shift at 3*Co); out = [relu](conv(x)*scale + shift [+ residual]). */
int ieee_conv2d_fwd_bn_eval(const void* x, const void* w_packed, void* out, const void* residual,
                            const float* bn_stats, int relu, int dtype, int64_t groups, int64_t N, int64_t Hi,
                            int64_t Wi, int64_t Ci, int64_t Co, int64_t R, int64_t S, int64_t stride, int64_t pad,
                            int64_t x_gs, int64_t w_gs, int64_t out_gs, void* stream);

/* dx[N,Hi,Wi,Ci] = conv_transpose(dy) (+ addend, same layout as dx, may be NULL): the
 * residual-branch gradient of Bottleneck (resnet.py:181 `out += identity`) is folded in here */
int ieee_conv2d_dgrad(const void* dy, const void* w_packed_d, void* dx, const void* addend, int dtype,
                      int64_t groups, int64_t N, int64_t Hi, int64_t Wi, int64_t Ci, int64_t Co, int64_t R,
                      int64_t S, int64_t stride, int64_t pad, int64_t dy_gs, int64_t w_gs, int64_t dx_gs,
                      float* bn_partial, const void* bn_y, const void* bn_mask, const float* bn_stats,
                      int bn_mask_bits, int addend_stride, void* stream);
/* The same dgrad with the sums of a SECOND BatchNorm that is fed by the same gradient (the downsample branch of the block whose
 * output this dgrad differentiates; torchreid/models/resnet.py:176-181): bn_y2 = that BatchNorm's input (shape and group stride of
 * bn_y), bn_partial2 = its own partial block of ieee_conv2d_fwd_stats_rblocks(N, Hi, Wi) * 2 * Ci floats per group, which
 * receives sum g and sum g*y2 -- its ieee_bn2d_bwd(stats_rblocks > 0) then needs no reduction pass either.  bf16 fused form only. */
int ieee_conv2d_dgrad2(const void* dy, const void* w_packed_d, void* dx, const void* addend, int dtype,
                      int64_t groups, int64_t N, int64_t Hi, int64_t Wi, int64_t Ci, int64_t Co, int64_t R,
                      int64_t S, int64_t stride, int64_t pad, int64_t dy_gs, int64_t w_gs, int64_t dx_gs,
                      float* bn_partial, const void* bn_y, const void* bn_mask, const float* bn_stats,
                      int bn_mask_bits, int addend_stride, const void* bn_y2,
                      float* bn_partial2, void* stream);
/* bn_partial != NULL (bf16 only): dx is the gradient w.r.t. the output of a BN(+ReLU) whose input is bn_y; the
 * dgrad epilogue also emits that BN's backward sums [2][Ci][rblocks] (sum g, sum g*y; g = dx * [mask], mask from
 * bn_mask > 0, or from bn_y*scale+shift > 0 with bn_stats = that BN's [4][Ci] stats, or none), rblocks =
 * ceil(N*Hi*Wi/128): pass it to ieee_bn2d_bwd(stats_rblocks) and the separate reduction pass disappears.
 * With bn_mask (the ReLU behind a residual add, resnet.py:181-182) dx is stored ALREADY MASKED, dx = g: the
 * BatchNorm backward that follows then takes out_mask = NULL and reads g and y only.  bn_mask_bits = 1: bn_mask is
 * the packed form ieee_bn2d_fwd(relu_bits) wrote (one byte per 8 channels, 1/16 of the bytes), else the activation.
 * addend_stride = 2 (with bn_partial, stride 1, power-of-two Hi / Wi): addend is [N, Hi/2, Wi/2, Ci] and belongs to the
 * pixels with even row and column -- what the stride-2 1x1 downsample branch of a block (resnet.py:548-556) sends back;
 * that branch's dgrad is then a dense 1x1 stride-1 dgrad over the (Ho, Wo) grid instead of a 3/4-zero full-size map */

/* dw (fp32, OIHW, the layout of param.grad) = or += sum over pixels; deterministic split-K:
 * partial slabs in `work` (size from the query below) are reduced in a fixed order */
int64_t ieee_conv2d_wgrad_workspace_bytes(int dtype, int64_t groups, int64_t N, int64_t Ho, int64_t Wo,
                                          int64_t Ci, int64_t Co, int64_t R, int64_t S);
int ieee_conv2d_wgrad(const void* dy, const void* x, float* dw_oihw, void* work, int dtype, int64_t groups,
                      int64_t N, int64_t Hi, int64_t Wi, int64_t Ci, int64_t Co, int64_t R, int64_t S,
                      int64_t stride, int64_t pad, int64_t dy_gs, int64_t x_gs, int64_t dw_gs, int accumulate,
                      void* stream);
/* The same weight gradient with the split-K fold INSIDE the launch (round 6; same semantics: autograd of
 * torchreid/models/resnet.py:164-184's convs).  The workgroups of an output tile store their fp32 partial tiles write-through,
 * take a ticket, and the one that completes the count adds the tiles up IN SPLIT ORDER (two levels for more than 16 splits) and
 * writes the OIHW gradient: bit-reproducible from run to run (no float atomics), no reduction launch, no second pass over the
 * slabs from a cold cache.  The association differs from ieee_conv2d_wgrad's (groups of ~sqrt(nsplit) splits instead of 16
 * strided lanes), so the two agree to fp32 rounding, not bit for bit.
 * tickets: ieee_conv2d_wgrad_fold_ticket_words() int32 words, ZERO before the first call; every completed launch leaves them
 * zero, so consecutive calls on one stream may share them (calls on different streams may not).  `work` as for
 * ieee_conv2d_wgrad (the query covers the level-1 slabs).  Shapes the fold does not cover (fp32, the stem, a single split,
 * unaligned gradients) take ieee_conv2d_wgrad's path and leave the tickets untouched. */
int64_t ieee_conv2d_wgrad_fold_ticket_words(void);
int ieee_conv2d_wgrad_fold(const void* dy, const void* x, float* dw_oihw, void* work, int32_t* tickets, int dtype,
                           int64_t groups, int64_t N, int64_t Hi, int64_t Wi, int64_t Ci, int64_t Co, int64_t R, int64_t S,
                           int64_t stride, int64_t pad, int64_t dy_gs, int64_t x_gs, int64_t dw_gs, int accumulate,
                           void* stream);
/* The same weight gradient with its slab reduction DEFERRED: the launch leaves the split-K slabs in `work` (which must then
 * stay untouched until the reduction has run) and describes the reduction in *reduce; ieee_wgrad_reduce_batch runs the
 * reductions of many weight gradients in ONE launch.  Usage: collect the descriptors of a group of layers, drop those
 * with kind == 0 (nothing left to reduce), set block_begin to the running sum of `blocks`, copy the table to the device
 * and call ieee_wgrad_reduce_batch(table, n, sum of blocks, groups, stream) on the stream the gradients were launched on.
 * Same arithmetic, same fixed summation order as the immediate form (bit-identical gradients). */
typedef struct ieee_wgrad_reduce_desc {
  const float* slab;        /* device: [groups][nsplit][Co][RS * Ci] */
  float* dw;                /* device: OIHW gradient, group stride dw_gs */
  int64_t slab_gs, dw_gs;
  int32_t nsplit, Co, Ci, RS;
  int32_t kind;             /* 0 nothing to do, 1 / 2 split-lane form (16-byte / scalar), 3 LDS-transposed 3x3 form */
  int32_t sl_log2, blocks;  /* workgroups per group */
  int32_t accumulate;
  int32_t block_begin;      /* filled by the caller: first blockIdx.x of this entry in the batched launch */
  int32_t reserved_;
} ieee_wgrad_reduce_desc;
int ieee_conv2d_wgrad_deferred(const void* dy, const void* x, float* dw_oihw, void* work, int dtype, int64_t groups,
                      int64_t N, int64_t Hi, int64_t Wi, int64_t Ci, int64_t Co, int64_t R, int64_t S,
                      int64_t stride, int64_t pad, int64_t dy_gs, int64_t x_gs, int64_t dw_gs, int accumulate,
                      ieee_wgrad_reduce_desc* reduce, void* stream);
int ieee_wgrad_reduce_batch(const ieee_wgrad_reduce_desc* device_descs, int64_t n, int64_t total_blocks, int64_t groups,
                            void* stream);
/* CHAINED weight gradients (round 6): ieee_conv2d_wgrad_deferred whose launch also runs the reduction `prev` -- the
 * descriptor an EARLIER deferred / chained call on the SAME stream returned; NULL or kind 0: none -- as its PROLOGUE: every
 * workgroup of this GEMM takes a share of that reduction's grid before its own k-loop.  The kernel boundary between the two
 * launches is the only synchronisation (no tickets, no fences), the arithmetic is the reduction launch's (bit-identical
 * gradients), and the ~50 reduction launches of a backward pass disappear from the stream.  Rules: `work` of consecutive
 * chained calls must alternate between two regions (prev's slabs are read while this call's are written; checked); `prev` is
 * always taken care of when the call returns IEEE_OK (a form without a prologue -- the stem -- launches it first); the LAST
 * descriptor of a chain is run by ieee_wgrad_reduce_pending before anything reads that gradient.  Host-side structs, passed
 * by value to the kernels: nothing is uploaded. */
int ieee_conv2d_wgrad_chained(const void* dy, const void* x, float* dw_oihw, void* work, int dtype, int64_t groups,
                      int64_t N, int64_t Hi, int64_t Wi, int64_t Ci, int64_t Co, int64_t R, int64_t S,
                      int64_t stride, int64_t pad, int64_t dy_gs, int64_t x_gs, int64_t dw_gs, int accumulate,
                      const ieee_wgrad_reduce_desc* prev, ieee_wgrad_reduce_desc* reduce, void* stream);
int ieee_wgrad_reduce_pending(const ieee_wgrad_reduce_desc* pending, int64_t groups, void* stream);


/* ---- BatchNorm2d (+ residual, + ReLU) over NHWC maps [M][C] ------------------ */
/* torch.nn.BatchNorm2d as the reference instantiates it (resnet.py:151,164-184;
 * ieee3modalPart.py:38): eps/momentum passed in, biased variance to normalise,
 * unbiased for running_var.  groups x [M][C] activations, stride act_gs;
 * gamma/beta at param_gs, running stats at buf_gs.
 * stats  : out, groups x [4][C] = mean, invstd, scale, shift (kept for backward)
 * partial: scratch of groups * ieee_bn_partial_floats() floats; with stats_rblocks > 0 it already holds the
 *          [2][C][stats_rblocks] sums emitted by ieee_conv2d_fwd and the statistics kernel is skipped
 * out = [relu]( y*scale + shift [+ residual] ); out NULL = statistics only.  training=0 uses running stats. */
int64_t ieee_bn_partial_floats(int dtype, int64_t M, int64_t C);
int ieee_bn2d_fwd(const void* y, const void* residual, void* out, int dtype, int64_t groups, int64_t M,
                  int64_t C, int64_t act_gs, const float* gamma, const float* beta, int64_t param_gs,
                  float* running_mean, float* running_var, int64_t buf_gs, float* stats, float* partial,
                  float momentum, float eps, int training, int relu, int64_t stats_rblocks, void* relu_bits,
                  void* stream);
/* backward of out = [relu](bn(y) [+ residual]): g = dout * [out_mask > 0] (out_mask NULL = no ReLU, or,
 * with mask_from_y = 1, the mask is recomputed as [y*scale+shift > 0], valid when there was no residual);
 * dy = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)); g_out (optional) receives g, which is also
 * the gradient of the residual branch.  dgamma/dbeta fp32 at grad_gs (NULL to skip).
 * coef: scratch groups x [3][C]. */
int ieee_bn2d_bwd(const void* dout, const void* out_mask, const void* y, void* dy, void* g_out, int dtype,
                  int64_t groups, int64_t M, int64_t C, int64_t act_gs, const float* gamma, int64_t param_gs,
                  const float* stats, float* dgamma, float* dbeta, int64_t grad_gs, float* partial,
                  float* coef, int accumulate, int mask_from_y, int64_t stats_rblocks, void* stream);
/* the same with `done_event` (a hipEvent_t or NULL): signalled when the call's last kernel completes, carried by that
 * dispatch itself instead of a separate event record behind it (a record is a barrier packet in the queue: +5 us
 * before the next kernel of the stream).  The executor forks its weight-gradient stream from these events. */
int ieee_bn2d_bwd_ev(const void* dout, const void* out_mask, const void* y, void* dy, void* g_out, int dtype,
                  int64_t groups, int64_t M, int64_t C, int64_t act_gs, const float* gamma, int64_t param_gs,
                  const float* stats, float* dgamma, float* dbeta, int64_t grad_gs, float* partial,
                  float* coef, int accumulate, int mask_from_y, int64_t stats_rblocks, void* done_event, void* stream);
/* BatchNorm statistics as order-independent TOTALS (round 4; no reference counterpart: autograd's batch_norm computes its
 * statistics in one kernel, torchreid/models/resnet.py:164-184 only calls it).  ieee_conv_next_bn_totals arms the NEXT
 * ieee_conv2d_fwd (with fused statistics) or ieee_conv2d_dgrad / _dgrad2 (with fused backward sums) issued by the calling
 * thread: instead of one float per (channel, 128-row tile) in bn_partial, every tile ADDS its per-channel sums, as 64-bit
 * fixed point (2^24 forward: sum y, sum y^2; 2^40 backward: sum g, sum g*y), to totals[group][2][C] with no-return global
 * atomics.  Integer adds commute: the totals are bit-reproducible whatever the arrival order; nothing inside the conv waits.
 * The caller zeroes the totals beforehand.  ieee_bn2d_fwd_totals / ieee_bn2d_bwd_totals then finalize AND apply in one
 * launch (every workgroup derives its channels' coefficients from the 2 x C integers in its prologue; workgroup 0 publishes
 * stats / running statistics / d(gamma), d(beta)), i.e. the separate finalize launch of ieee_bn2d_fwd / ieee_bn2d_bwd
 * disappears.  bf16 only, C / 8 must divide 256.  Same results as the partial-sum path up to the last bit of a float sum.
 * Range and resolution of the fixed point: totals stay inside +-2^62 units -- forward sums up to 2.7e11 with 6e-8 absolute
 * resolution (a 131 072-row map reaches that at an rms of 1 400 per channel), backward sums up to 4.2e6 with 9e-13 (finer than
 * the float tile sum itself above 1.5e-5).  Range guard: a tile contributes at most 2^62 / (row tiles of the launch) units, so
 * the total of all tiles and replicas can NEVER wrap; a tile sum beyond that share, or a NaN one, is clamped and REPORTED:
 * `overflow` (may be NULL) points to 4 ints the caller owns -- device or host-mapped memory, zero before the step -- and
 * overflow[0] = 1: a forward tile sum was clamped, [1]: a backward one, [2] / [3]: a forward / backward total beyond half
 * the range (2^61; written by ieee_bn2d_fwd_totals / _bwd_totals).  Any non-zero word means the statistics of that step are
 * not the reference's (torch's fp32 batch_norm, torchreid/models/resnet.py:164-184, returns finite numbers or inf there):
 * the executor raises through ieee_net_bn_overflow; the per-tile partial-sum path (ieee_bn2d_fwd / ieee_bn2d_bwd) has no
 * such limit.
 * `replicas` (a power of two <= 64; 1 = plain) spreads the adders: totals[replica][group][2][C] (group_stride = 2 * C), row
 * tile t adds to replica t % replicas, and the BatchNorm passes add the replicas up in their prologue -- same-address atomics
 * retire at ~23 ns each, so a launch of 1 024 row tiles pays +24 us with one copy and +3 us with eight. */
int ieee_conv_next_bn_totals(void* totals, int64_t group_stride, int replicas, int* overflow);   /* DEPRECATED: see ieee_conv_extras */
/* EXPLICIT-ARGUMENT forms (round 6; SURVEY.md section 8b: "stateless, re-entrant, all work on the caller-supplied stream").
 * ieee_conv_next_bn_totals / ieee_conv_profile_events change what the NEXT conv call of the thread does -- hidden per-thread
 * state a binding cannot express as one call.  ieee_conv2d_fwd_ex / ieee_conv2d_dgrad_ex take the same options as one struct
 * (NULL = none) and leave nothing behind; the executor calls these.  The two arming entry points stay as thin deprecated
 * wrappers over the same mechanism (they are disarmed on every return path of the next conv call, as before).
 *   totals / group_stride / replicas / overflow : as ieee_conv_next_bn_totals (totals NULL = per-tile partials in bn_partial)
 *   start / stop : hipEvent_t pair (timing enabled) the launch carries as its own start / stop signals, or NULL, NULL */
typedef struct ieee_conv_extras {
  void* totals;
  int64_t group_stride;
  int32_t replicas;
  int32_t reserved_;        /* 0 */
  int* overflow;
  void* start;
  void* stop;
} ieee_conv_extras;
int ieee_conv2d_fwd_ex(const void* x, const void* w_packed, void* y, int dtype, int64_t groups, int64_t N, int64_t Hi,
                       int64_t Wi, int64_t Ci, int64_t Co, int64_t R, int64_t S, int64_t stride, int64_t pad, int64_t x_gs,
                       int64_t w_gs, int64_t y_gs, float* bn_partial, const ieee_conv_extras* extras, void* stream);
int ieee_conv2d_dgrad_ex(const void* dy, const void* w_packed_d, void* dx, const void* addend, int dtype,
                         int64_t groups, int64_t N, int64_t Hi, int64_t Wi, int64_t Ci, int64_t Co, int64_t R,
                         int64_t S, int64_t stride, int64_t pad, int64_t dy_gs, int64_t w_gs, int64_t dx_gs,
                         float* bn_partial, const void* bn_y, const void* bn_mask, const float* bn_stats,
                         int bn_mask_bits, int addend_stride, const void* bn_y2, float* bn_partial2,
                         const ieee_conv_extras* extras, void* stream);
int ieee_bn2d_fwd_totals(const void* y, const void* residual, void* out, int dtype, int64_t groups, int64_t M,
                         int64_t C, int64_t act_gs, const float* gamma, const float* beta, int64_t param_gs,
                         float* running_mean, float* running_var, int64_t buf_gs, float* stats, const void* totals,
                         int replicas, float momentum, float eps, int relu, void* relu_bits, int* overflow, void* stream);
int ieee_bn2d_bwd_totals(const void* dout, const void* out_mask, const void* y, void* dy, void* g_out, int dtype,
                         int64_t groups, int64_t M, int64_t C, int64_t act_gs, const float* gamma, int64_t param_gs,
                         const float* stats, float* dgamma, float* dbeta, int64_t grad_gs, const void* totals,
                         int replicas, int mask_from_y, int* overflow, void* done_event, void* stream);
/* ieee_bn2d_bwd_totals for the LAST BatchNorm of a bottleneck block with a downsample branch (out = relu(bn3(y3) + bn_ds(y_ds)),
 * torchreid/models/resnet.py:164-184: both BatchNorm backwards see the same masked gradient g = dout): dy = the backward of
 * bn3 as above (no mask, no g output), and -- while g streams by -- the branch's backward sums, sum g and sum g*y_ds, are ADDED to
 * totals_ds [replicas_ds][groups][2][C] (2^40 fixed point, zero beforehand; same range guard: overflow[1] / [3]), so that the
 * branch's BatchNorm backward is one ieee_bn2d_bwd_totals launch without a reduction pass or a finalize launch. */
int ieee_bn2d_bwd_totals_ds(const void* dout, const void* y, const void* y_ds, void* dy, int dtype, int64_t groups, int64_t M,
                            int64_t C, int64_t act_gs, const float* gamma, int64_t param_gs, const float* stats,
                            float* dgamma, float* dbeta, int64_t grad_gs, const void* totals, int replicas, void* totals_ds,
                            int replicas_ds, int* overflow, void* done_event, void* stream);
/* backward through a FROZEN BatchNorm2d (module.eval() while the rest trains: open_specified_layers, utils/torchtools.py:
 * 183-221): `stats` holds the running-statistics scale / shift of the forward (ieee_bn2d_fwd with training = 0), the map
 * is a fixed affine one and dy = scale * g, g = dout * mask as in ieee_bn2d_bwd; no parameter gradient is produced. */
int ieee_bn2d_bwd_frozen(const void* dout, const void* out_mask, const void* y, void* dy, void* g_out, int dtype,
                         int64_t groups, int64_t M, int64_t C, int64_t act_gs, const float* stats, float* coef,
                         int mask_from_y, void* done_event, void* stream);
/* `done_event` above is never hipEventRecord-ed: it is the stop event of hipExtLaunchKernelGGL, and a consumer orders
 * another stream behind it with hipStreamWaitEvent.  HIP does not document that such an event orders a second stream, so
 * the dependency is MEASURED once per process before it is relied on: a ~300 us spinning kernel carries a pooled
 * hipEventDisableTiming event on stream_a, stream_b waits on it and reads the flag the kernel sets last (three rounds, one
 * 8-byte allocation, host-synchronous on both streams -- the only place the library waits for the device).  The executor
 * passes its launch and side stream; the check creates no streams of its own (doing so in the middle of a step disturbed
 * the runtime's hardware-queue assignment: the side stream stopped overlapping the launch stream).  Returns 0 when the
 * event rides (cached for the process), non-zero otherwise; the executor then falls back to hipEventRecord
 * (IEEE_EVENT_RIDE=0 forces that).  Replaces nothing in the reference (its streams are torch's). */
int ieee_event_ride_selfcheck(void* stream_a, void* stream_b);

/* ---- stem plumbing ----------------------------------------------------------- */
/* three fp32 NCHW image tensors (batch dict 'img' = [RGB, NI, TI], dataset.py:338-351) ->
 * [3][B][H+2*pad][W+2*pad][Cpad], channels C..Cpad-1 and the `pad`-pixel border zero.  The executor's stem uses
 * Cpad = 4, pad = 3: conv1 (7x7/2, pad 3; resnet.py:622) then is an 8x8/2 convolution WITHOUT padding over 4
 * channels (8th filter row / column and 4th channel zero), whose k-tile of 64 is two filter rows of 8 contiguous
 * pixels and needs no bounds tests */
int ieee_nchw_to_nhwc3(const float* x_rgb, const float* x_ni, const float* x_ti, void* out, int dtype,
                       int64_t B, int64_t C, int64_t H, int64_t W, int64_t Cpad, int64_t pad, void* stream);
/* nn.MaxPool2d(3, 2, 1) (resnet.py:501); argmax holds the window-local index of the first maximum */
int ieee_maxpool3x3s2_fwd(const void* x, void* out, uint8_t* argmax, int dtype, int64_t groups, int64_t B,
                          int64_t Hi, int64_t Wi, int64_t C, void* stream);
/* The stem's BatchNorm apply + ReLU + max-pool in one pass (training forward): pooled = maxpool(relu(y * scale + shift)) with
 * scale / shift from `stats` ([groups][4][C]: mean, invstd, scale, shift, as ieee_bn2d_fwd leaves them) and the activation
 * rounded to the storage type before the comparison -- out / argmax are bit-identical to ieee_bn2d_fwd(out = a, relu) followed
 * by ieee_maxpool3x3s2_fwd(a), without the full-resolution activation ever being written (torchreid/models/resnet.py:622-626). */
int ieee_bn_relu_maxpool3x3s2_fwd(const void* y, const float* stats, void* out, uint8_t* argmax, int dtype, int64_t groups,
                                  int64_t B, int64_t Hi, int64_t Wi, int64_t C, void* stream);

int ieee_maxpool3x3s2_bwd(const void* dout, const uint8_t* argmax, void* dx, int dtype, int64_t groups,
                          int64_t B, int64_t Hi, int64_t Wi, int64_t C, void* stream);

/* backward of maxpool(relu(bn(y))) w.r.t. y in two passes (the stem, resnet.py:622-626 reversed): the max-pool
 * backward of dpool [groups][B][Ho][Wo][C] is gathered on the fly, masked by [y*scale+shift > 0] and pushed through
 * the BatchNorm backward (same dgamma / dbeta / coef / partial contract as ieee_bn2d_bwd, M = B*Hi*Wi) */
int ieee_bn2d_bwd_pooled(const void* dpool, const uint8_t* argmax, const void* y, void* dy, int dtype,
                         int64_t groups, int64_t B, int64_t Hi, int64_t Wi, int64_t C, const float* gamma,
                         int64_t param_gs, const float* stats, float* dgamma, float* dbeta, int64_t grad_gs,
                         float* partial, float* coef, int accumulate, void* stream);

/* ---- Cross-modal Interacting Module tail (ieee3modalPart.py:266-282, 427-455) -- */
/* F [3][B][H*W][C]: S_m = F_a + F_b (the two other modalities; S may be NULL), Gp_m = mean over positions
 * (AdaptiveAvgPool2d((1,1)) of the raw trunk map, :449-451) */
int ieee_gpool_sum_others(const void* F, void* S, float* Gp, int dtype, int64_t B, int64_t H, int64_t W,
                          int64_t C, void* stream);
/* ChannelAttention pools of rest = relu(bn(y2)): avg / max over positions at avg|mx + m*pool_gs + b*C + c,
 * argmax [3][B][C] */
int ieee_ca_pool(const void* y2, const float* stats2, float* avg, float* mx, int64_t pool_gs, int32_t* argmax,
                 int dtype, int64_t B, int64_t H, int64_t W, int64_t C, void* stream);
/* parts_out[3][B][parts][C] = AdaptiveAvgPool2d((parts,1)) of relu(bn(y1)) + relu(bn(y2))*(1+att);
 * mode 0 full CIM, 1 attention off, 2 interaction off (pool y1 as is) */
int ieee_cim_tail_fwd(const void* y1, const void* y2, const float* stats1, const float* stats2,
                      const float* att, float* parts_out, int dtype, int64_t B, int64_t H, int64_t W,
                      int64_t C, int64_t parts, int mode, void* stream);
int ieee_cim_tail_bwd_datt(const float* dparts, const void* y2, const float* stats2, float* datt, int dtype,
                           int64_t B, int64_t H, int64_t W, int64_t C, int64_t parts, void* stream);
int ieee_cim_tail_bwd_g(const float* dparts, const void* y1, const void* y2, const float* stats1,
                        const float* stats2, const float* att, const float* davg, const float* dmax,
                        int64_t pool_gs, const int32_t* argmax, void* g1, void* g2, int dtype, int64_t B,
                        int64_t H, int64_t W, int64_t C, int64_t parts, int mode, float* bn_partial1,
                        float* bn_partial2, void* stream);
/* bn_partial1/2 (both or neither; modes 0/1): also emit the BatchNorm-backward sums of convOne / convAvgRest as
 * [3][2][C][B] per-sample partials (sum g, sum g*y of the stored g1 / g2): ieee_bn2d_bwd(stats_rblocks = B) then
 * skips its reduction pass */
/* dF_m = D1_m + DS_a + DS_b + dGp_m / (H*W) */
int ieee_cim_bwd_combine(const void* D1, const void* DS, const float* dG, void* dF, int dtype, int64_t B,
                         int64_t H, int64_t W, int64_t C, int mode, void* stream);

/* ---- embedding head, fp32, grouped through host pointer tables (<= IEEE_MAX_GROUPS) ---- */
/* C[g](m,n) = act(alpha * sum_k A[g](m,k)*B[g](n,k) + bias[g](n)) (+ C[g] when accumulate);
 * A(m,k) = A + m*sam + k*sak, B(n,k) = B + n*sbn + k*sbk: nn.Linear / 1x1 conv on pooled vectors and
 * their backward GEMMs (ieee3modalPart.py:54-56, 272-274, 335-340, 396-424, 374-391) */
int ieee_sgemm_grouped(int64_t groups, const void* const* A, const void* const* B, void* const* C,
                       const void* const* bias, int64_t M, int64_t N, int64_t K, int64_t sam, int64_t sak,
                       int64_t sbn, int64_t sbk, int64_t ldc, float alpha, int relu, int accumulate,
                       void* stream);
/* same, with a scratch buffer: long-K problems with few 64x64 tiles (the pooled-vector GEMMs: K = 2048 reduce
 * conv / CA fc1, K = 768 REM / fc heads) are split along K into `work` slabs that a second kernel adds in a fixed
 * order (deterministic); work == NULL or too small = no split.  16 MiB covers every GEMM of the head. */
int ieee_sgemm_grouped_ws(int64_t groups, const void* const* A, const void* const* B, void* const* C,
                          const void* const* bias, int64_t M, int64_t N, int64_t K, int64_t sam, int64_t sak,
                          int64_t sbn, int64_t sbk, int64_t ldc, float alpha, int relu, int accumulate, void* work,
                          int64_t work_bytes, void* stream);
/* TWO uniform problem sets in ONE launch (round 6): the independent GEMM pairs of a head phase -- dW and dX of one nn.Linear /
 * 1x1 conv (ieee3modalPart.py:54-56, 396-424, 484-511 through autograd), the reduce_layer applied to the global vector and
 * to the six parts (:449-455) -- each set described as ieee_sgemm_grouped's arguments.  Every output has the bits of the
 * separate ieee_sgemm_grouped_ws call made with HALF of `work_bytes` (each set plans its split-K against its half of `work`).
 * The two sets must not write overlapping outputs. */
typedef struct ieee_sgemm_set {
  int64_t groups;
  const void* const* A;
  const void* const* B;
  void* const* C;
  const void* const* bias;      /* may be NULL */
  int64_t M, N, K, sam, sak, sbn, sbk, ldc;
  float alpha;
  int32_t relu, accumulate;
} ieee_sgemm_set;
int ieee_sgemm_grouped_pair_ws(const ieee_sgemm_set* s0, const ieee_sgemm_set* s1, void* work, int64_t work_bytes,
                               void* stream);
/* zero `count` (<= IEEE_MAX_GROUPS) float spans in one launch */
int ieee_zero_spans(int64_t count, void* const* ptrs, const int64_t* floats, void* stream);
int ieee_colsum_grouped(int64_t groups, const void* const* X, void* const* out, int64_t M, int64_t N,
                        int64_t ldx, int accumulate, void* stream);
/* BatchNorm over the rows of [R][C] (+ReLU): BatchNorm2d on [B,C,1,1] / [B,C,6,1] (reduce_layer, applied
 * twice per step :449-455) and BatchNorm1d of the 18 fc heads (:417).  save[g] = [2][C] mean, invstd */
int ieee_rowbn_fwd(int64_t groups, const void* const* x, void* const* out, const void* const* gamma,
                   const void* const* beta, void* const* running_mean, void* const* running_var,
                   void* const* save, int64_t R, int64_t C, int64_t ldx, int64_t ldo, float momentum,
                   float eps, int training, int relu, void* stream);
/* backward: relu bit 0 = the forward applied ReLU; bit 1 = FROZEN BatchNorm (the forward ran with training = 0, e.g. a
 * child outside `open_layers` during the fixbase epochs, utils/torchtools.py:183-221): dx = gamma * invstd * g and
 * dgamma / dbeta are left untouched */
int ieee_rowbn_bwd(int64_t groups, const void* const* dout, const void* const* out, const void* const* x,
                   const void* const* gamma, const void* const* save, void* const* dx, void* const* dgamma,
                   void* const* dbeta, int64_t R, int64_t C, int64_t lddo, int64_t ldo, int64_t ldx,
                   int64_t lddx, int relu, int accumulate, void* stream);
/* ChannelAttention glue: h [3][2B][hidden] (avg rows then max rows, ReLU applied) -> hs = h_avg + h_max */
int ieee_ca_mix_fwd(const float* h, float* hs, int64_t B, int64_t hidden, void* stream);
int ieee_ca_mix_bwd(const float* dhs, const float* h, float* dh, int64_t B, int64_t hidden, void* stream);
int ieee_sigmoid_fwd(float* x, int64_t n, void* stream);
int ieee_sigmoid_bwd(const float* datt, const float* att, float* dz, int64_t n, void* stream);
/* REM closed form (nonLocal.forward, ieee3modalPart.py:60-80): out = part + 2*param*r, r = W_p g + b_p */
int ieee_rem_fwd(const float* part, const float* r, const float* param, int64_t param_gs, float* out,
                 int64_t B, int64_t parts, int64_t D, void* stream);
int ieee_rem_bwd(const float* dout, const float* r, const float* param, int64_t param_gs, float* dr,
                 float* dparam, int64_t grad_gs, float* work, int64_t B, int64_t parts, int64_t D,
                 int accumulate, void* stream);
/* F.normalize(p=2, dim=1, eps=1e-12) (:519) */
int ieee_l2norm_fwd(const float* x, float* y, float* norms, int64_t rows, int64_t D, void* stream);
int ieee_l2norm_bwd(const float* dy, const float* y, const float* norms, float* dx, int64_t rows, int64_t D,
                    int accumulate, void* stream);

/* ---- losses and optimizer ------------------------------------------------------- */
/* label-smoothed CE (torchreid/losses/cross_entropy_loss.py:36-50) over `heads` logits tensors [B][C];
 * head_loss[h] = (-t*logp).mean(0).sum(), head_acc[h] = top-1 % (metrics/accuracy.py);
 * dlogits (optional) = d(sum_h head_loss)/dlogits * grad_scale.  work: heads*B*2 floats. */
int ieee_ce_ls_fwd_bwd(const float* logits, const int64_t* targets, float* dlogits, float* head_loss,
                       float* head_acc, float* work, int64_t heads, int64_t B, int64_t C, float eps,
                       float grad_scale, void* stream);
/* 3M loss (torchreid/losses/multi_modal_margin_loss_new.py:19-40); feats [3][B][D] = R,N,T;
 * out3 = {loss, label_num, chunks torch.chunk would yield}; dfeats optional; work: B+3 floats */
int ieee_margin3m_fwd_bwd(const float* feats, const int64_t* pids, float* dfeats, float* out3, float* work,
                          int64_t B, int64_t D, float margin, float grad_scale, void* stream);
/* torch.optim.SGD(momentum, weight_decay, dampening=0, nesterov) as configured by the reference
 * (torchreid/optim/optimizer.py:130-138) over a flat fp32 range */
/* The same update with two options (either may be NULL; round 6).  Reference: torchreid/optim/optimizer.py:130-138.
 *  shadow_bf16: ALSO write the bf16 image of the updated parameters (round-to-nearest-even, the conversion the weight packing
 *    uses) to shadow_bf16[i], i in [0, n) -- the slice of a bf16 SHADOW of the flat parameter buffer.  With ieee_net_set_shadow
 *    the training forward reads the GEMM operands of the 1x1 convolutions straight from that shadow (Wf[co][ci] of a 1x1 conv
 *    is its OIHW weight), so the once-per-step weight packing loses 64 % of the conv parameters; 2 B/param of optimizer writes.
 *  skip_flags: the executor's range-guard words (device memory, ieee_net_bn_flags_offset).  When [0] or [1] is set -- a
 *    BatchNorm tile sum of THIS step left the fixed-point range, so its gradients are not the reference's -- the launch
 *    changes nothing: parameters, momentum and shadow stay as they were (the step is skipped, like an overflow step of a loss
 *    scaler) and training can go on from an undamaged state once the engine has switched to the partial-sum path. */
int ieee_sgd_nesterov_step_ex(float* params, const float* grads, float* momentum_buf, int64_t n, float lr, float momentum,
                              float weight_decay, int nesterov, void* shadow_bf16, const int* skip_flags, void* stream);
/* running statistics under the same guard: flags[0] set (a forward tile sum of this step was clamped) -> buffers[i] =
 * backup[i], the values before the step; else backup[i] = buffers[i].  One launch per step behind the forward. */
int ieee_guard_buffers(const int* flags, float* buffers, float* backup, int64_t n, void* stream);
int ieee_sgd_nesterov_step(float* params, const float* grads, float* momentum_buf, int64_t n, float lr,
                           float momentum, float weight_decay, int nesterov, void* stream);
/* Gradient exchange in bf16 ("219 MB in bf16"; opt-in, IEEE_DP_GRAD_DTYPE=bf16): the slice of the flat
 * fp32 gradient a rank is about to all-reduce, rounded to nearest even into a bf16 staging range, and the reduced bf16
 * values widened back (exact).  The reference reduces fp32 gradients (nn.DataParallel's reduce_add,
 * scripts/mainMultiModal.py:219-220): the default stays fp32. */
int ieee_grad_pack_bf16(const float* grads, void* out_bf16, int64_t n, void* stream);
int ieee_grad_unpack_bf16(const void* in_bf16, float* grads, int64_t n, void* stream);
/* torch.optim.Adam / Adam(amsgrad=True) as the reference builds them for optim = 'adam' / 'amsgrad'
 * (torchreid/optim/optimizer.py:113-128): L2 weight decay, bias correction with `step` (1-based);
 * max_exp_avg_sq == NULL = plain Adam */
int ieee_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq,
                   int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                   void* stream);

/* ---- re-ranking ----------------------------------------------- */
/* k-reciprocal re-ranking, torchreid/utils/rerank.py:31-113 (engine/engine.py:402-406): device matrices
 * q_g_dist [Q][G], q_q_dist [Q][Q], g_g_dist [G][G] fp32 -> out [Q][G] fp32.  Dense like the reference (three
 * (Q+G)^2 fp32 work matrices: Q+G < 46000); rank ties are broken by index.  k1+1 <= 64. */
int64_t ieee_rerank_workspace_bytes(int64_t Q, int64_t G, int64_t k1);
int ieee_rerank(const float* q_g_dist, const float* q_q_dist, const float* g_g_dist, int64_t Q, int64_t G,
                int64_t k1, int64_t k2, double lambda_value, float* out, void* work, int64_t work_bytes,
                void* stream);

/* ---- input pipeline ------------------------------------------ */
/* The reference's per-image chain Resize((Ho,Wo)) -> RandomHorizontalFlip -> ToTensor -> Normalize
 * (torchreid/data/transforms.py:233-326; dataset.py:335-351) for N decoded uint8 images of ONE source size:
 * src [N][Hs][Ws][3] (device) -> dst [N][3][Ho][Wo] fp32 (device).  The resize is Pillow's two-pass 8-bit bilinear
 * resampler, bit-exact: bounds_* [out][2] = (first source index, count), kk_* [out][ksize] = 22-bit fixed-point
 * weights, both computed on the host as Pillow does (ieee_amd/data/transforms.py) and resident on the device; a
 * table is NULL iff that axis keeps its size.  tmp: [N][tmp_rows][Wo][3] bytes for the horizontal pass, which only
 * covers source rows ybox_first .. ybox_first+tmp_rows (what the vertical pass reads).  flip [N] bytes or NULL;
 * mean3 / std3: HOST pointers to 3 floats. */
int ieee_resize_flip_normalize(const uint8_t* src, float* dst, uint8_t* tmp, int64_t N, int64_t Hs, int64_t Ws,
                               int64_t Ho, int64_t Wo, const int32_t* bounds_h, const int32_t* kk_h, int64_t ksize_h,
                               const int32_t* bounds_v, const int32_t* kk_v, int64_t ksize_v, int64_t ybox_first,
                               int64_t tmp_rows, const uint8_t* flip, const float* mean3, const float* std3,
                               void* stream);

/* ---- whole-network executor --------------------------------------------------- */
/* One handle = IEEE3modalPart (ieee3modalPart.py:286-523) for a fixed batch / image size / dtype.
 * The caller owns three flat fp32 buffers laid out like the reference's state_dict: parameters,
 * their gradients, and BN running statistics.  ieee_net_slot_name() enumerates the tensor names the
 * executor needs (e.g. "backbone.0.layer1.0.conv1.weight"); ieee_net_bind() receives, in that order,
 * each tensor's element offset inside `params` (running_mean/var: inside `buffers`).
 * forward : x_* fp32 NCHW [B,3,H,W].  training: logits [18][B][C] (R0..5,N0..5,T0..5) and
 *           feats [3][B][768] = F.normalize(fc_{R,N,T}_all); eval: feats = fc_all [B][2304] (T,R,N).
 * backward: dlogits / dfeats in the same layouts; every parameter gradient is OVERWRITTEN in `grads`
 *           (REM.conv_query gets exact zeros, REM.conv_value and unused branches are not touched).
 * Streams: the dependent chain of a call (forward; dgrad / BatchNorm backward) is enqueued on the caller's `stream`.
 * The executor also owns two internal non-blocking streams per handle, created on first use: a low-priority SIDE
 * stream (every weight-gradient kernel + its slab reduction, and the packing of the layer3 / layer4 / CIM operands
 * during a training forward) and a BRANCH stream (the downsample branch of the first block of each stage).  They
 * are ordered against `stream` with events in both directions, and they are JOINED into `stream` before
 * ieee_net_forward, ieee_net_backward and ieee_net_backward_part return -- after those calls, synchronising (or
 * enqueuing behind) `stream` covers everything the call launched.  The one exception is
 * ieee_net_backward_part_async: weight gradients of the part may still run on the side stream when it returns; join
 * them with ieee_net_side_wait or ieee_net_sync_streams before reading gradients, destroying `stream` or the
 * workspace, or stepping the optimizer.  No call synchronises the host (ieee_net_profile(.., 0, ..) aside).
 * The workspace (ieee_net_workspace_bytes) keeps the activations between forward and backward. */
int ieee_net_create(int64_t batch, int64_t height, int64_t width, int64_t num_classes, int dtype,
                    int interaction, int attention, int using_rem, void** handle);
int ieee_net_destroy(void* handle);
int64_t ieee_net_num_slots(void* handle);
const char* ieee_net_slot_name(void* handle, int64_t i);
int64_t ieee_net_workspace_bytes(void* handle);
int ieee_net_bind(void* handle, float* params, float* grads, float* buffers, const int64_t* offsets,
                  int64_t num_offsets);
int ieee_net_forward(void* handle, void* workspace, const float* x_rgb, const float* x_ni, const float* x_ti,
                     int training, float* logits, float* feats, void* stream);
int ieee_net_backward(void* handle, void* workspace, const float* dlogits, const float* dfeats, void* stream);
/* the same backward in 5 consecutive parts (0: head + CIM, 1: layer4, 2: layer3, 3: layer2, 4: layer1 + stem; call
 * them in this order): after part p all gradients of that part are final, so a data-parallel caller overlaps
 * their all-reduce with the remaining parts (ieee_amd/engine.py) */
int ieee_net_backward_part(void* handle, void* workspace, const float* dlogits, const float* dfeats, int part,
                           void* stream);
/* Same, without blocking the launch stream at the end of the part: the part's weight gradients may still be running
 * on the executor's side stream when the call returns.  ieee_net_side_wait(waiting_stream, 0) makes ANOTHER stream
 * (e.g. the one a collective is issued from) wait for every weight gradient issued so far;
 * ieee_net_side_wait(launch_stream, 1) is the final join that must precede the optimizer step (it also retires
 * the executor's buffer-reuse bookkeeping, so pass the stream the parts were launched on). */
int ieee_net_backward_part_async(void* handle, void* workspace, const float* dlogits, const float* dfeats, int part,
                                 void* stream);
int ieee_net_side_wait(void* handle, void* workspace, void* waiting_stream, int is_launch_stream);
/* Make `stream` wait for everything the executor has enqueued on its internal streams so far (side + branch), without
 * knowing about them: after this, hipStreamSynchronize(stream) -- or any work enqueued on `stream` -- is ordered
 * behind every kernel of the preceding ieee_net_* calls.  A C caller that uses the _async parts calls this (on the
 * launch stream) where the Python engine calls ieee_net_side_wait(launch_stream, 1); it is a no-op when nothing is
 * pending.  Does not block the host. */
int ieee_net_sync_streams(void* handle, void* stream);
/* Frozen children (Engine.two_stepped_transfer_learning -> open_specified_layers, torchreid/utils/torchtools.py:183-221: the
 * children outside `open_layers` are put in eval() mode and stop requiring gradients for the first fixbase_epoch epochs).
 * mask: bit 0 backbone, 1 convOne, 2 convAvgRest, 3 reduce_layer, 4 fc_R, 5 fc_N, 6 fc_T (the children that own a
 * BatchNorm).  A frozen child's BatchNorms use their running statistics in a TRAINING forward, do not update them, and
 * the backward treats them as the fixed affine maps they then are (ieee_bn2d_bwd_frozen); gradients still flow through a
 * frozen child to whatever trains below it.  The caller keeps frozen parameters out of the optimizer step. */
#define IEEE_FROZEN_BACKBONE 1
#define IEEE_FROZEN_CONV_ONE 2
#define IEEE_FROZEN_CONV_REST 4
#define IEEE_FROZEN_REDUCE 8
#define IEEE_FROZEN_FC_R 16
#define IEEE_FROZEN_FC_N 32
#define IEEE_FROZEN_FC_T 64
int ieee_net_set_frozen(void* handle, int mask);
/* Range guard of the fixed-point BatchNorm totals (bf16 training; see ieee_conv_next_bn_totals): out4 receives, and the call
 * CLEARS, the four report words the kernels of the MOST RECENT training step have set -- [0] a forward tile sum was clamped
 * to its share of the int64 range (or was NaN), [1] a backward one, [2] / [3] a forward / backward total beyond half the
 * range.  The words are device memory at the head of the executor's totals region, zeroed by every training forward (round 6:
 * they were host-mapped words); this call is a blocking 16-byte copy -- synchronise with the step first.  All zero = the
 * statistics of that step are the partial-sum path's up to the last bit of a float sum.  Non-zero: they are NOT what torch's
 * fp32 batch_norm (torchreid/models/resnet.py:164-184) would have computed; a step whose words [0] / [1] are set is skipped by
 * ieee_sgd_nesterov_step_ex(skip_flags) / ieee_guard_buffers, and the engine switches to the partial-sum path. */
int ieee_net_bn_overflow(void* handle, int* out4);
/* byte offset of those four int32 words inside the workspace (they are device memory, zeroed by every training forward): the
 * engine copies them to the host with the step's summary, the optimizer takes them as `skip_flags` */
int64_t ieee_net_bn_flags_offset(void* handle);
/* on = 0: from the next forward on, every BatchNorm of this executor takes the per-tile partial-sum path (ieee_bn2d_fwd /
 * ieee_bn2d_bwd: fp32 sums without a range, as the reference's fp32 nn.BatchNorm2d, torchreid/models/resnet.py:151,164-184;
 * +104 finalize launches per step); on = 1: fixed-point totals again where IEEE_BN_TOTALS_TILES allows.  What the engine does
 * when ieee_net_bn_overflow reports a clamped tile: degrade, warn, keep training (IEEE_BN_STRICT=1: raise instead). */
int ieee_net_set_bn_totals(void* handle, int on);
/* shadow_bf16: a bf16 buffer with one element per element of the bound parameter buffer, element i = bf16(params[i]), that the
 * CALLER keeps current (ieee_sgd_nesterov_step_ex, or a plain conversion after any other write) -- every bf16 TRAINING
 * forward issued while it is set reads the forward operands of the 1x1 convolutions from it instead of packing them.  NULL
 * (the default): every operand is packed from the fp32 parameters at the start of the forward.  Inference is not affected. */
int ieee_net_set_shadow(void* handle, const void* shadow_bf16);
/* Inference cache: after an eval-mode ieee_net_forward the workspace holds the packed weights and every BatchNorm's
 * scale / shift; the next eval forward on the same workspace reuses them (no packing launch, no finalize launches)
 * unless ieee_net_eval_cache(handle, 0) was called in between.  The CALLER must call it whenever parameters or
 * running statistics may have changed outside ieee_net_forward(training = 1) (optimizer steps, state loading, ...). */
int ieee_net_eval_cache(void* handle, int keep);
/* measurement: enable=1 starts recording a HIP event pair around every conv launch of subsequent forward/backward
 * calls with ALL work on one ordered stream (the serialized pass: what each launch takes alone); enable=2 times the
 * launches with the executor's own streams left on -- every forward / dgrad launch carries its pair as its own start /
 * stop signals (ieee_conv_profile_events: nothing added to the queue), weight-gradient calls get an event record on each
 * side, on the side stream they run on: what a launch takes INSIDE the two-stream step; enable=0 stops, synchronises the device and returns
 * out6 = {ms, algorithmic FLOPs, launches} for [0] forward+dgrad (conv_gather / conv3x3_patch / stem_conv kernels) and
 * [1] wgrad (conv_wgrad / conv3x3_wgrad_patch / stem_wgrad kernels + their slab reductions) */
int ieee_net_profile(void* handle, int enable, double* out6);
/* measurement: the next conv forward / dgrad launch issued by the calling thread signals `start` when it begins and `stop`
 * when it ends (hipEvent_t with timing enabled; NULL, NULL cancels): the kernel's own duration by the GPU's timestamps,
 * without an event record in the queue.  ieee_net_profile(.., 2, ..) times the forward / dgrad family this way. */
int ieee_conv_profile_events(void* start, void* stop);
/* debugging / parity tests: location of a named intermediate inside the workspace */
int ieee_net_tensor(void* handle, const char* name, int64_t* byte_offset, int64_t* numel, int* dtype);
/* parity tests of the backward (what autograd keeps implicit in the reference, ieee3modalPart.py:439-523 under
 * loss.backward()): while `buffer` (device memory, `bytes` long) is set, ieee_net_backward* copies every gradient
 * tensor it produces into it before the buffer that held it is reused -- "<unit>.dy" (gradient of the conv output =
 * output of that unit's BatchNorm backward), "<unit>.dx" (what the unit's dgrad wrote: for a block's conv1 the masked
 * block-input gradient, residual branch added), "<unit>.g" for block-output units (dout * [out > 0]); a stride-2
 * downsample branch that returns its gradient compact taps "<unit>.dx_compact".  <unit> = the conv's state_dict
 * prefix, modality as {m}: "backbone.{m}.layer1.0.conv1".  buffer = NULL switches the taps off.
 * ieee_net_debug_tap: byte offset of a tap inside the buffer (after the backward that wrote it). */
int ieee_net_debug_taps(void* handle, void* buffer, int64_t bytes);
int ieee_net_debug_tap(void* handle, const char* name, int64_t* byte_offset, int64_t* numel, int* dtype);

#ifdef __cplusplus
}
#endif
#endif /* IEEE_AMD_H */
