// Layer-graph executor for IEEE3modalPart (reference torchreid/models/ieee3modalPart.py:286-523 on
// top of three torchreid/models/resnet.py ResNetIEEE trunks): plans one workspace arena, then
// enqueues the whole forward / backward on the caller's stream as a fixed sequence of ieee_* kernels.
// The three modality streams (RGB, NI, TI) are batched into every launch (groups = 3).
// No allocation, no host synchronisation; parameters stay in the caller's flat fp32 buffers, laid
// out exactly like the reference's state_dict (the caller binds name -> offset, see ieee_net_bind).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace {

using namespace ieee;

struct Tensor {
  size_t off = 0;     // byte offset in the workspace
  int64_t numel = 0;  // total elements (all groups)
  int dtype = IEEE_F32;
};

// BatchNorm sums as fixed-point totals (conv.hip: tl_totals) for the units of at most IEEE_BN_TOTALS_TILES row tiles per
// modality (default 1024: layer1-4; 0 switches the path off; the stem's 4 096 tiles always write partial sums): the conv
// epilogue adds, the BatchNorm apply / backward-apply launch finalizes in its prologue, the finalize launch is gone.
// Same-address atomics retire at ~23 ns each -- a conv of 64 / 256 / 1 024 row tiles pays +1 / +4.5 / +24 us for its sums
// with one copy of the totals -- so units of more than 64 tiles spread their adders over IEEE_BN_TOTALS_REP copies
// (default 4; the prologue that adds the copies up costs +1.3 us at 4, +3.6 at 8: scripts/bn_totals_probe.py).
// Returns the copies to use, 0 = not on this path.  Measurements: LABNOTES.md, "Round 4".
static constexpr int TOT_REP_MAX = 8;
static int totals_tiles() {
  static const int t = getenv("IEEE_BN_TOTALS_TILES") ? atoi(getenv("IEEE_BN_TOTALS_TILES")) : 1024;
  return t;
}
static int totals_rep_for(int64_t tiles) {
  static const int rep = [] {
    int r = getenv("IEEE_BN_TOTALS_REP") ? atoi(getenv("IEEE_BN_TOTALS_REP")) : 4;
    return (r == 1 || r == 2 || r == 4 || r == 8) ? r : 4;
  }();
  if (tiles > totals_tiles()) return 0;
  return tiles <= 64 ? 1 : rep;
}

struct ConvUnit {
  std::string name;
  int Ci, Co, R, stride, pad, Hi, Wi, Ho, Wo;
  int S = 0;                       // kernel width (== R except for the column-padded stem)
  int Ci_src = 0, S_src = 0, R_src = 0;   // dims of the reference parameter when the unit runs zero-padded
  int s_w, s_g, s_b, s_rm, s_rv;   // slot ids of modality 0 (modality m = id + m)
  Tensor y, a, stats, wf, wd, dwpad, slab;
  Tensor tot_f, tot_b;             // fixed-point totals of the fused BatchNorm sums (forward / backward), [3][2][Co] int64
  Tensor abits;                    // bf16 block outputs: the ReLU mask of `a` as packed bits (bn_apply writes it, the next block's conv1 dgrad reads it)
  bool want_bits = false;
  bool need_dgrad = true;
  bool shadow_ok = false;          // 1x1, unpadded rows, 16-byte aligned slots: the forward operand can be read from the bf16 shadow
  int child = IEEE_FROZEN_BACKBONE;   // which freezable child of the model owns this unit (ieee_net_set_frozen bits)
  int64_t M(int B) const { return (int64_t)B * Ho * Wo; }
};

// must match struct PackDesc in conv.hip
struct PackDescHost {
  int64_t src_off, src_gs, dst_off, dst_gs;
  int Co, Ci, R, S, ld, mode, Ci_src, S_src;
  int block_begin, pad_;
  int R_src, reserved_;
};
extern "C" int ieee_pack_all_weights(const float* params, void* ws_base, const void* descs, int64_t ndesc,
                                     int64_t total_blocks, int dtype, void* stream);

// IEEE_WGRAD_BATCH: 0 (default) every weight gradient reduces its split-K slabs at once; 1 one batched reduction per
// backward part (own slab region per unit, +1.7 GB); 2 one per bottleneck block.  Measured: 1 and 2 are 0.1 ms per step
// SLOWER than 0 (the serialized weight-gradient time improves 1 %, but the large reductions take HBM bandwidth from the
// dependent chain at once instead of in slices) -- kept as an option of the C ABI, not the default.
static int gbuf_sets() {
  static const int n = getenv("IEEE_GBUF_SETS") ? atoi(getenv("IEEE_GBUF_SETS")) : 3;
  return n == 2 ? 2 : 3;
}

// IEEE_WGRAD_FOLD=1 (default 0): the weight gradients fold their split-K slabs inside their OWN GEMM launch
// (ieee_conv2d_wgrad_fold: last arriver per output tile).  Parity-green, bit-reproducible, and +0.68 ms per step (LABNOTES R6.1):
// one reducer per tile is latency-bound.  The default is the chained form below.
static bool wgrad_fold() {
  static const bool on = getenv("IEEE_WGRAD_FOLD") && atoi(getenv("IEEE_WGRAD_FOLD")) != 0;
  return on;
}
// IEEE_WGRAD_CHAIN=1 (default 0): every weight-gradient GEMM runs the slab reduction of the PREVIOUS one as its prologue
// (ieee_conv2d_wgrad_chained; two alternating slab regions); the last reduction of a backward part is flushed as a launch of
// its own (wgrad_flush).  Same arithmetic as the reduction launches (bit-identical gradients), 46 launches fewer per step --
// and +0.18 ms per step (14.24 -> 14.42, 4 interleaved rounds, LABNOTES R6.1): a GEMM workgroup that first walks its share of
// the reduction (a chain of dependent round trips on ~450 workgroups instead of ~2 000 short-lived ones) holds its LDS and
// registers that much longer.  The reduction launches stay the default.
static bool wgrad_chain() {
  static const bool on = getenv("IEEE_WGRAD_CHAIN") && atoi(getenv("IEEE_WGRAD_CHAIN")) != 0;
  return on;
}
static int wgrad_batch_mode() {
  static const int m = getenv("IEEE_WGRAD_BATCH") ? atoi(getenv("IEEE_WGRAD_BATCH")) : 0;
  return m;
}

struct Block {
  int c1, c2, c3, ds;   // unit indices (ds = -1 when there is no downsample branch)
};

struct Net {
  static constexpr int NGBUF = 15;   // three sets of five activation-sized gradient buffers (IEEE_GBUF_SETS=2: two of them used)
  // config
  int B, H, W, num_classes, dtype, interaction, attention, using_rem;
  int frozen = 0;                    // ieee_net_set_frozen: children whose BatchNorms run on their running statistics
  int parts = 6, rdim = 768, cdim = 128, fdim = 2048, hid = 128;
  float bn_eps = 1e-5f, bn_mom = 0.1f;
  // slots
  std::vector<std::string> slot_names;
  std::vector<int64_t> slot_off;
  std::vector<int> triples;   // first slot id of every per-modality triple (uniform stride is required)
  float *params = nullptr, *grads = nullptr, *buffers = nullptr;
  bool bound = false;
  // graph
  std::vector<ConvUnit> units;
  std::vector<Block> blocks;
  int u_stem = -1, u_one = -1, u_rest = -1;
  int s_ca1, s_ca2, s_rw, s_rg, s_rb, s_rrm, s_rrv, s_rem_param, s_rem_pw, s_rem_pb, s_rem_qw, s_rem_qb;
  int s_fcw[18], s_fcb[18], s_fcg[18], s_fcbe[18], s_fcrm[18], s_fcrv[18], s_clw[18], s_clb[18];
  // workspace
  size_t ws_bytes = 0;
  std::map<std::string, Tensor> tensors;
  Tensor x0, pool, pool_arg, S, gbuf[NGBUF], slab, bnpart, bncoef, bnpart2, bncoef2, packtab, gemm_work;
  // Weight-gradient slabs: every unit owns its region, so the split-K reductions of a whole backward part can run as ONE
  // launch at the end of the part (ieee_wgrad_reduce_batch; IEEE_WGRAD_BATCH=0: one reduction per gradient, at once).
  // rtab: the device copies of the (at most 6) descriptor tables; uploaded again only when their content changes.
  static constexpr int RT_SLOTS = 24, RT_MAX = 32;
  Tensor rtab;
  std::vector<ieee_wgrad_reduce_desc> rpend;
  std::vector<ieee_wgrad_reduce_desc> rtab_host[RT_SLOTS];
  const void* rtab_ws[RT_SLOTS] = {};
  ConvUnit reduce_unit;   // profiling label of the batched reductions
  size_t tot_begin = 0, tot_end = 0;   // the units' totals are one contiguous region: ONE memset per training forward
  Tensor slab2;     // second slab region of the chained weight gradients (wgrad_chain): consecutive launches alternate
  ieee_wgrad_reduce_desc wpend = {};   // the chained reduction that the next weight-gradient launch (or wgrad_flush) still owes
  int wflip = 0;
  Tensor wtickets;  // arrival tickets of the weight gradients that fold their split-K slabs themselves (ieee_conv2d_wgrad_fold)
  Tensor tickets;   // 2 x 256 int32: arrival tickets of the convs that finalize their BatchNorm themselves (launch / branch stream)
  std::vector<PackDescHost> pack_train, pack_eval;   // all units (train: + dgrad operands)
  int pack_blocks_train = 0, pack_blocks_eval = 0;
  // training: the layer3 / layer4 / CIM operands (90 % of the bytes) are packed on the side stream while the
  // caller's stream runs the stem, layer1 and layer2; pack_late = those descriptors with their own block numbering
  std::vector<PackDescHost> pack_late;
  int pack_late_first = 0, pack_late_unit = 0, pack_blocks_early = 0, pack_blocks_late = 0;
  // The same tables WITHOUT the forward operands of the 1x1 convs (ieee_net_set_shadow): with a bf16 shadow of the parameter
  // buffer kept current by the optimizer (ieee_sgd_nesterov_step_shadow) those GEMM operands ARE the shadow -- Wf[co][ci] is
  // the OIHW weight itself -- and the training forward reads them there: 61.6 M of the 95.7 M conv parameters leave the pack.
  std::vector<PackDescHost> pack_train_s, pack_late_s;
  int pack_blocks_train_s = 0, pack_late_first_s = 0, pack_blocks_early_s = 0, pack_blocks_late_s = 0;
  const void* shadow = nullptr;   // bf16 [numel of params], element i = bf16(params[i]); nullptr: every operand is packed
  hipEvent_t pack_ev[2] = {nullptr, nullptr};
  const void* pack_uploaded_ws = nullptr;
  bool fused_bwd_state = false;   // survives between the staged ieee_net_backward_part calls
  // range guard of the fixed-point totals (conv.hip: tl_totals_flag): 4 ints at the head of the totals region -- DEVICE memory,
  // zeroed with the totals by every training forward, set by plain stores on the overflow path only ([0] / [1]: a forward /
  // backward tile sum was clamped, [2] / [3]: a total beyond half the range).  Round 6: they were host-mapped words the host
  // polled; on the device the SAME step can act on them -- the BatchNorm passes leave the running statistics alone and the
  // optimizer (ieee_sgd_nesterov_step_ex, skip_flags) leaves parameters and momentum alone once [0] or [1] is set, so a
  // clamped step is SKIPPED instead of applied -- and the engine reads them with the step's summary (one 16-byte copy).
  Tensor oflags;
  const void* last_ws = nullptr;  // workspace of the most recent forward (ieee_net_bn_overflow reads the flags there)
  bool totals_off = false;        // ieee_net_set_bn_totals(0): every unit on the per-tile partial-sum path (no range limit)
  bool bwd_totals_fresh = false;  // the backward totals are zero (set by the training forward, cleared by the backward that uses them)
  bool bwd_totals_state = false;  // ... and: those sums went into the unit's fixed-point totals (tot_b), not into bn_partial
  bool stem_a_valid = false;      // "<stem>.a" holds the activation of the LAST forward (a training forward does not write it)
  // inference: the packed weights and every BatchNorm's scale / shift only depend on the parameters and running
  // statistics; while the caller vouches that those have not changed (ieee_net_eval_cache) consecutive eval forwards
  // skip the packing launch and the 55 finalize launches
  bool eval_cache_valid = false;
  const void* eval_cache_ws = nullptr;
  // Weight-gradient kernels run on a second (low-priority) HIP stream: nothing on the dgrad / BN-backward chain
  // depends on them, so they fill the tails and the HBM-bound phases of that chain.  side_ready[i] is recorded on
  // the caller's stream when wgrad i's dY operand is final; gbuf_read[b] on the side stream after the last wgrad
  // that reads gradient buffer b (the caller's stream waits on it before it overwrites b).
  hipStream_t side = nullptr;
  // third stream: the downsample branch of the first block of every stage (conv + BN forward; BN backward + dgrad)
  // is independent of that block's conv1 -> conv2 (-> conv3) chain and runs beside it with its own BN scratch
  hipStream_t side2 = nullptr;
  hipEvent_t branch_ev[16] = {};
  std::vector<hipEvent_t> side_ready;
  hipEvent_t gbuf_read[NGBUF] = {};
  hipEvent_t side_done = nullptr;
  hipEvent_t sync_ev = nullptr;     // ieee_net_sync_streams: the branch stream's join event
  bool gbuf_pending[NGBUF] = {};
  uint64_t gbuf_seq[NGBUF] = {}, gbuf_clock = 0;   // order of the gbuf_read records on the side stream
  int gbuf_evt[NGBUF] = {};                          // which gbuf_read event protects buffer b (several buffers may share one)
  bool side_dirty = false;
  size_t side_used = 0;
  ~Net() {
    for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : side_ready) (void)hipEventDestroy(e);
    for (hipEvent_t e : gbuf_read) if (e) (void)hipEventDestroy(e);
    if (side_done) (void)hipEventDestroy(side_done);
    if (sync_ev) (void)hipEventDestroy(sync_ev);
    for (hipEvent_t e : pack_ev) if (e) (void)hipEventDestroy(e);
    if (side) (void)hipStreamDestroy(side);
    for (hipEvent_t e : branch_ev) if (e) (void)hipEventDestroy(e);
    if (side2) (void)hipStreamDestroy(side2);
  }
  Tensor Gp, avgmax, amax, Hh, Hs, att, Pp, Zg, Zp, glob, part, sv_g, sv_p, rr, part2, fcraw, sv_fc, featcat, fcall,
      logits, featn, norms;
  Tensor dfeatcat, dfcraw, dpart2, dr, dglob, dZp, dZg, dPp, dGp, datt, dHs, dH, davgmax, remwork;
  int esz() const { return dtype == IEEE_BF16 ? 2 : 4; }
  // parity tests (ieee_net_debug_taps): while a tap buffer is set, the backward copies every gradient tensor it
  // produces (dY of each conv unit = the output of its BatchNorm backward; the tensor each dgrad wrote) into it, in
  // launch order, before the buffer holding it is reused
  char* tap_base = nullptr;
  size_t tap_bytes = 0, tap_used = 0;
  std::map<std::string, Tensor> taps;
  // optional per-launch timing of the conv kernels (bench.py's roofline leg): category 0 = forward +
  // dgrad (conv_gather_kernel), 1 = wgrad (conv_wgrad_kernel)
  int profiling = 0;   // 1: per-launch events on ONE ordered stream (the serialized roofline pass); 2: the same events with the
                       // weight-gradient / branch streams left on (what a launch takes inside the real two-stream step)
  std::vector<hipEvent_t> ev_pool;
  std::vector<int> ev_cat;
  std::vector<std::string> ev_name;
  std::vector<const char*> ev_label;
  std::vector<double> ev_flops;
  size_t ev_used = 0;
  double prof_flops[2] = {0, 0};
  int64_t prof_launches[2] = {0, 0};

  int slot3(const std::string& pattern) {   // registers the three per-modality names, returns the first id
    const int id = (int)slot_names.size();
    triples.push_back(id);
    for (int m = 0; m < 3; ++m) {
      std::string s = pattern;
      const size_t p = s.find("{m}");
      s.replace(p, 3, std::to_string(m));
      slot_names.push_back(s);
    }
    return id;
  }
  int slot1(const std::string& name) {
    slot_names.push_back(name);
    return (int)slot_names.size() - 1;
  }
  Tensor alloc(const std::string& name, int64_t numel, int dt) {
    Tensor t;
    t.off = ws_bytes;
    t.numel = numel;
    t.dtype = dt;
    const size_t bytes = (size_t)numel * (dt == IEEE_BF16 ? 2 : (dt == 2 ? 1 : 4));
    ws_bytes += (bytes + 255) / 256 * 256;
    if (!name.empty()) tensors[name] = t;
    return t;
  }
  int add_unit(const std::string& conv, const std::string& bn, int Ci, int Co, int R, int stride, int pad, int Hi,
               int Wi) {
    ConvUnit u;
    u.name = conv;
    u.Ci = Ci; u.Co = Co; u.R = R; u.S = R; u.Ci_src = Ci; u.S_src = R; u.R_src = R;
    u.stride = stride; u.pad = pad; u.Hi = Hi; u.Wi = Wi;
    u.Ho = (Hi + 2 * pad - R) / stride + 1;
    u.Wo = (Wi + 2 * pad - R) / stride + 1;
    u.s_w = slot3(conv + ".weight");
    u.s_g = slot3(bn + ".weight");
    u.s_b = slot3(bn + ".bias");
    u.s_rm = slot3(bn + ".running_mean");
    u.s_rv = slot3(bn + ".running_var");
    units.push_back(u);
    return (int)units.size() - 1;
  }
  void build();
  void plan();
};

void Net::build() {
  const std::string bb = "backbone.{m}.";
  // The stem (conv 7x7/2 pad 3 over 3 channels, resnet.py:622) runs as an 8x8/2 convolution WITHOUT padding over a
  // 4-channel image that ieee_nchw_to_nhwc3 writes with a 3-pixel zero border: the 8th filter row / column and the 4th
  // channel are zero.  A 16-byte chunk is then two horizontally adjacent pixels, a k-tile of 64 two whole filter rows
  // (K = 256, 4 k-tiles instead of the 7 of an 8-channel 7x8 form), and no load needs a bounds test.
  u_stem = add_unit(bb + "conv1", bb + "bn1", 3, 64, 7, 2, 3, H, W);
  units[u_stem].need_dgrad = false;
  {
    ConvUnit& s = units[u_stem];
    s.Ci = 4; s.R = 8; s.S = 8;            // (Ci_src, R_src, S_src) stay (3, 7, 7)
    s.Hi = H + 6; s.Wi = W + 6; s.pad = 0;  // same output size: (H + 6 - 8) / 2 + 1 = (H + 6 - 7) / 2 + 1 for even H
  }
  int h = units[u_stem].Ho, w = units[u_stem].Wo;
  h = (h + 2 - 3) / 2 + 1;   // maxpool 3x3 s2 p1
  w = (w + 2 - 3) / 2 + 1;
  int inpl = 64;
  const int planes[4] = {64, 128, 256, 512}, nblk[4] = {3, 4, 6, 3}, strides[4] = {1, 2, 2, 1};   // last_stride = 1
  for (int L = 0; L < 4; ++L) {
    for (int b = 0; b < nblk[L]; ++b) {
      const std::string p = bb + "layer" + std::to_string(L + 1) + "." + std::to_string(b) + ".";
      const int st = b == 0 ? strides[L] : 1;
      Block blk;
      blk.c1 = add_unit(p + "conv1", p + "bn1", inpl, planes[L], 1, 1, 0, h, w);
      blk.c2 = add_unit(p + "conv2", p + "bn2", planes[L], planes[L], 3, st, 1, h, w);
      const int ho = units[blk.c2].Ho, wo = units[blk.c2].Wo;
      blk.c3 = add_unit(p + "conv3", p + "bn3", planes[L], planes[L] * 4, 1, 1, 0, ho, wo);
      blk.ds = -1;
      if (b == 0) blk.ds = add_unit(p + "downsample.0", p + "downsample.1", inpl, planes[L] * 4, 1, st, 0, h, w);
      // every block output but the last is the ReLU mask of the next block's input gradient (IEEE_RELU_BITS=0: read
      // the activation itself for the mask, the round-1 form)
      static const bool f_bits = !(getenv("IEEE_RELU_BITS") && atoi(getenv("IEEE_RELU_BITS")) == 0);
      units[blk.c3].want_bits = f_bits && !(L == 3 && b == nblk[L] - 1);
      blocks.push_back(blk);
      inpl = planes[L] * 4;
      h = ho; w = wo;
    }
  }
  u_one = add_unit("convOne.{m}.layers.0", "convOne.{m}.layers.1", fdim, fdim, 1, 1, 0, h, w);
  u_rest = add_unit("convAvgRest.{m}.layers.0", "convAvgRest.{m}.layers.1", fdim, fdim, 1, 1, 0, h, w);
  units[u_one].child = IEEE_FROZEN_CONV_ONE;
  units[u_rest].child = IEEE_FROZEN_CONV_REST;
  s_ca1 = slot3("CA.{m}.fc.0.weight");
  s_ca2 = slot3("CA.{m}.fc.2.weight");
  s_rw = slot3("reduce_layer.{m}.layers.0.weight");
  s_rg = slot3("reduce_layer.{m}.layers.1.weight");
  s_rb = slot3("reduce_layer.{m}.layers.1.bias");
  s_rrm = slot3("reduce_layer.{m}.layers.1.running_mean");
  s_rrv = slot3("reduce_layer.{m}.layers.1.running_var");
  s_rem_param = slot3("REM.{m}.param");
  s_rem_pw = slot3("REM.{m}.conv_part.weight");
  s_rem_pb = slot3("REM.{m}.conv_part.bias");
  s_rem_qw = slot3("REM.{m}.conv_query.weight");
  s_rem_qb = slot3("REM.{m}.conv_query.bias");
  const char* letters[3] = {"R", "N", "T"};
  for (int m = 0; m < 3; ++m)
    for (int i = 0; i < 6; ++i) {
      const int g = m * 6 + i;
      const std::string f = std::string("fc_") + letters[m] + "." + std::to_string(i) + ".";
      const std::string c = std::string("classifier_") + letters[m] + "." + std::to_string(i) + ".";
      s_fcw[g] = slot1(f + "0.weight");
      s_fcb[g] = slot1(f + "0.bias");
      s_fcg[g] = slot1(f + "1.weight");
      s_fcbe[g] = slot1(f + "1.bias");
      s_fcrm[g] = slot1(f + "1.running_mean");
      s_fcrv[g] = slot1(f + "1.running_var");
      s_clw[g] = slot1(c + "weight");
      s_clb[g] = slot1(c + "bias");
    }
  slot_off.assign(slot_names.size(), -1);
}

extern "C" int64_t ieee_conv_packed_ld(int dtype, int64_t inner_channels, int64_t R, int64_t S);
extern "C" int64_t ieee_conv2d_wgrad_workspace_bytes(int dtype, int64_t groups, int64_t N, int64_t Ho, int64_t Wo,
                                                     int64_t Ci, int64_t Co, int64_t R, int64_t S);
extern "C" int64_t ieee_bn_partial_floats(int dtype, int64_t M, int64_t C);
extern "C" int64_t ieee_conv2d_fwd_stats_rblocks(int64_t N, int64_t Ho, int64_t Wo);
extern "C" int64_t ieee_conv2d_fwd_bn_train_max_rows(void);

void Net::plan() {
  ws_bytes = 0;
  const int dt = dtype;
  x0 = alloc("x0", (int64_t)3 * B * (H + 6) * (W + 6) * 4, dt);
  int64_t max_act = 0, max_slab = 0, max_part = 0, max_c = 0;
  for (size_t i = 0; i < units.size(); ++i) {
    ConvUnit& u = units[i];
    const int64_t n = 3 * u.M(B) * u.Co;
    u.y = alloc(u.name + ".y", n, dt);
    u.a = alloc(u.name + ".a", n, dt);
    if (u.want_bits && dt == IEEE_BF16) u.abits = alloc(u.name + ".abits", n / 8, 2 /*u8*/);
    u.stats = alloc(u.name + ".stats", (int64_t)3 * 4 * u.Co, IEEE_F32);
    u.wf = alloc("", 3 * u.Co * ieee_conv_packed_ld(dt, u.Ci, u.R, u.S), dt);
    if (u.need_dgrad) u.wd = alloc("", 3 * u.Ci * ieee_conv_packed_ld(dt, u.Co, u.R, u.S), dt);
    if (u.Ci != u.Ci_src || u.S != u.S_src || u.R != u.R_src) u.dwpad = alloc("", (int64_t)3 * u.Co * u.Ci * u.R * u.S, IEEE_F32);
    max_act = std::max(max_act, n);
    if (u.need_dgrad) max_act = std::max(max_act, (int64_t)3 * B * u.Hi * u.Wi * u.Ci);
    {
      const int64_t sb = ieee_conv2d_wgrad_workspace_bytes(dt, 3, B, u.Ho, u.Wo, u.Ci, u.Co, u.R, u.S);
      max_slab = std::max(max_slab, sb);
      if (wgrad_batch_mode() != 0) u.slab = alloc("", sb / 4 + 64, IEEE_F32);
    }
    max_part = std::max(max_part, 3 * ieee_bn_partial_floats(dt, u.M(B), u.Co));
    max_part = std::max(max_part, 3 * ieee_conv2d_fwd_stats_rblocks(B, u.Ho, u.Wo) * 2 * u.Co);
    max_c = std::max(max_c, (int64_t)u.Co);
  }
  tot_begin = ws_bytes;
  oflags = alloc("", 64, IEEE_F32);      // 4 ints (+ padding to the allocation granule), first thing in the region
  for (size_t i = 0; i < units.size(); ++i) {
    ConvUnit& u = units[i];
    // (a run at a smaller batch than the planned one has fewer tiles: it asks for the same number of copies or for one)
    const int rep = (u.M(B) + 127) / 128 <= 64 ? 1 : TOT_REP_MAX;
    u.tot_f = alloc("", (int64_t)rep * 3 * 2 * u.Co * 2, IEEE_F32);   // [replicas][3][2][Co] int64 = 2 floats' worth each
    u.tot_b = alloc("", (int64_t)rep * 3 * 2 * u.Co * 2, IEEE_F32);
  }
  wtickets = alloc("", ieee_conv2d_wgrad_fold_ticket_words(), IEEE_F32);   // zeroed with the totals; every launch leaves them zero
  tot_end = ws_bytes;
  const ConvUnit& st = units[u_stem];
  const int ph = (st.Ho + 2 - 3) / 2 + 1, pw = (st.Wo + 2 - 3) / 2 + 1;
  pool = alloc("pool", (int64_t)3 * B * ph * pw * 64, dt);
  pool_arg = alloc("pool.arg", (int64_t)3 * B * ph * pw * 64, 2 /*u8*/);
  const ConvUnit& uo = units[u_one];
  const int64_t P = (int64_t)uo.Hi * uo.Wi;
  S = alloc("S", (int64_t)3 * B * P * fdim, dt);
  // two sets of five activation-sized gradient buffers: consecutive bottleneck blocks alternate between them, so a
  // buffer is rewritten two blocks after the side-stream wgrad that reads it was issued (one set stalled the chain)
  for (int i = 0; i < NGBUF; ++i) gbuf[i] = alloc("g" + std::to_string(i), max_act, dt);
  slab = alloc("", max_slab / 4 + 64, IEEE_F32);
  if (wgrad_chain()) slab2 = alloc("", max_slab / 4 + 64, IEEE_F32);
  max_part = std::max(max_part, (int64_t)12 * B * fdim);   // two [3][2][C][B] sets from ieee_cim_tail_bwd_g
  bnpart = alloc("", max_part + 64, IEEE_F32);
  bncoef = alloc("", 3 * 3 * max_c, IEEE_F32);
  bnpart2 = alloc("", max_part + 64, IEEE_F32);      // BN scratch of the downsample-branch stream
  bncoef2 = alloc("", 3 * 3 * max_c, IEEE_F32);
  const int64_t Bq = B;
  Gp = alloc("Gp", 3 * Bq * fdim, IEEE_F32);
  avgmax = alloc("avgmax", 3 * 2 * Bq * fdim, IEEE_F32);
  amax = alloc("amax", 3 * Bq * fdim, IEEE_F32);
  Hh = alloc("H", 3 * 2 * Bq * hid, IEEE_F32);
  Hs = alloc("Hs", 3 * Bq * hid, IEEE_F32);
  att = alloc("att", 3 * Bq * fdim, IEEE_F32);
  Pp = alloc("Pp", 3 * Bq * parts * fdim, IEEE_F32);
  Zg = alloc("Zg", 3 * Bq * rdim, IEEE_F32);
  Zp = alloc("Zp", 3 * Bq * parts * rdim, IEEE_F32);
  glob = alloc("glob", 3 * Bq * rdim, IEEE_F32);
  part = alloc("part", 3 * Bq * parts * rdim, IEEE_F32);
  sv_g = alloc("", 3 * 2 * rdim, IEEE_F32);
  sv_p = alloc("", 3 * 2 * rdim, IEEE_F32);
  rr = alloc("r", 3 * Bq * rdim, IEEE_F32);
  part2 = alloc("part2", 3 * Bq * parts * rdim, IEEE_F32);
  fcraw = alloc("fcraw", 18 * Bq * cdim, IEEE_F32);
  sv_fc = alloc("", 18 * 2 * cdim, IEEE_F32);
  featcat = alloc("featcat", 3 * Bq * rdim, IEEE_F32);
  fcall = alloc("fcall", Bq * 3 * rdim, IEEE_F32);
  logits = alloc("logits", 18 * Bq * num_classes, IEEE_F32);
  featn = alloc("featn", 3 * Bq * rdim, IEEE_F32);
  norms = alloc("", 3 * Bq, IEEE_F32);
  dfeatcat = alloc("dfeatcat", 3 * Bq * rdim, IEEE_F32);
  dfcraw = alloc("dfcraw", 18 * Bq * cdim, IEEE_F32);
  dpart2 = alloc("dpart2", 3 * Bq * parts * rdim, IEEE_F32);
  dr = alloc("dr", 3 * Bq * rdim, IEEE_F32);
  dglob = alloc("dglob", 3 * Bq * rdim, IEEE_F32);
  dZp = alloc("dZp", 3 * Bq * parts * rdim, IEEE_F32);
  dZg = alloc("dZg", 3 * Bq * rdim, IEEE_F32);
  dPp = alloc("dPp", 3 * Bq * parts * fdim, IEEE_F32);
  dGp = alloc("dGp", 3 * Bq * fdim, IEEE_F32);
  datt = alloc("datt", 3 * Bq * fdim, IEEE_F32);
  dHs = alloc("dHs", 3 * Bq * hid, IEEE_F32);
  dH = alloc("dH", 3 * 2 * Bq * hid, IEEE_F32);
  davgmax = alloc("davgmax", 3 * 2 * Bq * fdim, IEEE_F32);
  remwork = alloc("", 3 * Bq + 64, IEEE_F32);
  gemm_work = alloc("", (int64_t)16 << 20, IEEE_F32);  // split-K slabs of the head GEMMs: 2 x 32 MiB (one half per set of a pair)
  tickets = alloc("", 512, IEEE_F32);
  rtab = alloc("", (int64_t)RT_SLOTS * RT_MAX * sizeof(ieee_wgrad_reduce_desc) / 4, IEEE_F32);
  reduce_unit = ConvUnit();
  reduce_unit.name = "wgrad_reduce_batch";
  reduce_unit.Ci = reduce_unit.Co = reduce_unit.R = reduce_unit.stride = reduce_unit.pad = 0;
  reduce_unit.Hi = reduce_unit.Wi = reduce_unit.Ho = reduce_unit.Wo = 0;
  packtab = alloc("", (int64_t)(9 * units.size() + 8) * sizeof(PackDescHost) / 4, IEEE_F32);
}

// ---------------------------------------------------------------------------------------------
struct Run {
  Net& n;
  char* ws;
  void* st;
  int B;
  float *bnpart_cur, *bncoef_cur;      // BN scratch of the stream `st` currently denotes (see BranchScope)
  int32_t* tickets_cur;
  Run(Net& net, void* workspace, void* stream) : n(net), ws((char*)workspace), st(stream), B(net.B) {
    bnpart_cur = (float*)(ws + net.bnpart.off);
    bncoef_cur = (float*)(ws + net.bncoef.off);
    tickets_cur = (int32_t*)(ws + net.tickets.off);
  }
  void* P(const Tensor& t) const { return ws + t.off; }
  float* F(const Tensor& t) const { return (float*)(ws + t.off); }
  float* par(int slot) const { return n.params + n.slot_off[slot]; }
  float* grd(int slot) const { return n.grads + n.slot_off[slot]; }
  float* buf(int slot) const { return n.buffers + n.slot_off[slot]; }
  int64_t gs(int slot) const { return n.slot_off[slot + 1] - n.slot_off[slot]; }

  int pack(const ConvUnit& u, bool with_dgrad) {
    const int64_t ldf = ieee_conv_packed_ld(n.dtype, u.Ci, u.R, u.S);
    if (u.Ci != u.Ci_src || u.S != u.S_src || u.R != u.R_src) {
      IEEE_TRY(ieee_pack_conv_weight_padded(par(u.s_w), P(u.wf), n.dtype, 0, 3, u.Co, u.Ci_src, u.R_src, u.S_src, u.Ci, u.R, u.S,
                                            gs(u.s_w), u.Co * ldf, st));
      return IEEE_OK;
    }
    IEEE_TRY(ieee_pack_conv_weight(par(u.s_w), P(u.wf), n.dtype, 0, 3, u.Co, u.Ci, u.R, u.S, gs(u.s_w), u.Co * ldf, st));
    if (with_dgrad && u.need_dgrad) {
      const int64_t ldd = ieee_conv_packed_ld(n.dtype, u.Co, u.R, u.S);
      IEEE_TRY(ieee_pack_conv_weight(par(u.s_w), P(u.wd), n.dtype, 1, 3, u.Co, u.Ci, u.R, u.S, gs(u.s_w), u.Ci * ldd, st));
    }
    return IEEE_OK;
  }
  int* oflags_ptr() const { return (int*)(ws + n.oflags.off); }
  // options of the NEXT ieee_conv2d_fwd_ex / _dgrad_ex call (explicit-argument ABI: nothing is armed inside the library)
  ieee_conv_extras ex_ = {};
  const ieee_conv_extras* take_extras(void* totals, int64_t group_stride, int replicas) {
    ex_.totals = totals; ex_.group_stride = group_stride; ex_.replicas = replicas; ex_.overflow = totals ? oflags_ptr() : nullptr;
    ex_.reserved_ = 0;
    return &ex_;
  }
  // by_launch: the call that follows is ieee_conv2d_fwd_ex / _dgrad_ex, which can carry the event pair as its own signals
  void prof_begin(int cat, const ConvUnit& u, const char* label = nullptr, bool by_launch = false) {
    ex_.start = ex_.stop = nullptr;
    if (!n.profiling) return;
    n.ev_label.push_back(label ? label : (cat ? "wgrad" : "fwd"));
    if (n.ev_used + 2 > n.ev_pool.size()) {
      for (int i = 0; i < 256; ++i) { hipEvent_t e; (void)hipEventCreate(&e); n.ev_pool.push_back(e); }
    }
    // in situ (mode 2), forward / dgrad: the launch itself carries the pair as its start / stop signals -- nothing is added
    // to the queue; otherwise an event record on each side of the call
    timed_by_launch = n.profiling == 2 && cat == 0 && by_launch;
    if (timed_by_launch) { ex_.start = (void*)n.ev_pool[n.ev_used]; ex_.stop = (void*)n.ev_pool[n.ev_used + 1]; }
    else (void)hipEventRecord(n.ev_pool[n.ev_used], (hipStream_t)st);
    n.ev_cat.push_back(cat);
    n.ev_name.push_back(u.name + " " + std::to_string(u.Ci_src) + "->" + std::to_string(u.Co) + " k" + std::to_string(u.R_src) +
                        " s" + std::to_string(u.stride) + " " + std::to_string(u.Ho) + "x" + std::to_string(u.Wo));
    n.ev_flops.push_back(3.0 * 2.0 * (double)u.M(B) * u.Co * u.R_src * u.S_src * u.Ci_src);
    n.prof_flops[cat] += 3.0 * 2.0 * (double)u.M(B) * u.Co * u.R_src * u.S_src * u.Ci_src;   // algorithmic, 3 modalities
    n.prof_launches[cat] += 1;
  }
  bool timed_by_launch = false;
  void prof_end() {
    if (!n.profiling) return;
    if (!timed_by_launch) (void)hipEventRecord(n.ev_pool[n.ev_used + 1], (hipStream_t)st);
    ex_.start = ex_.stop = nullptr;
    timed_by_launch = false;
    n.ev_used += 2;
  }
  int tap(const std::string& name, const void* p, int64_t numel, int dt = -1) {
    if (n.tap_base == nullptr) return IEEE_OK;
    if (dt < 0) dt = n.dtype;
    const size_t bytes = (size_t)numel * (dt == IEEE_BF16 ? 2 : (dt == 2 ? 1 : 4));
    IEEE_REQUIRE(n.tap_used + bytes <= n.tap_bytes, "net: tap buffer too small at '%s' (%zu + %zu > %zu bytes)", name.c_str(),
                 n.tap_used, bytes, n.tap_bytes);
    Tensor t;
    t.off = n.tap_used; t.numel = numel; t.dtype = dt;
    IEEE_HIP(hipMemcpyAsync(n.tap_base + t.off, p, bytes, hipMemcpyDeviceToDevice, (hipStream_t)st));
    n.taps[name] = t;
    n.tap_used += (bytes + 255) / 256 * 256;
    return IEEE_OK;
  }
  int64_t in_numel(const ConvUnit& u) const { return (int64_t)3 * B * u.Hi * u.Wi * u.Ci; }
  int64_t out_numel(const ConvUnit& u) const { return (int64_t)3 * u.M(B) * u.Co; }
  // fused_stats: the conv epilogue emits the BN partial sums (bf16 training path) -> bn() skips its stats pass
  bool fused_stats = false;
  bool fused_fin = false;    // ... and finalized them too (ieee_conv2d_fwd_bn_train): bn() only applies
  bool frz(const ConvUnit& u) const { return (n.frozen & u.child) != 0; }
  int totals_rep(const ConvUnit& u) const { return n.totals_off ? 0 : totals_rep_for((u.M(B) + 127) / 128); }
  bool use_totals(const ConvUnit& u) const {
    // (the kernels fetch gamma / beta / stats as 16-byte loads: slot offsets and modality strides multiples of 4 floats)
    return n.dtype == IEEE_BF16 && totals_rep(u) > 0 && u.Co % 8 == 0 && 256 % (u.Co / 8) == 0 && u.Co / 8 <= 256 &&
           &u != &n.units[n.u_stem] && ((n.slot_off[u.s_g] | n.slot_off[u.s_b] | gs(u.s_g)) & 3) == 0;
  }
  bool fwd_totals = false;   // the last conv() put its statistics into u.tot_f
  int conv(const ConvUnit& u, const void* in, bool want_stats = false) {
    const int64_t ldf = ieee_conv_packed_ld(n.dtype, u.Ci, u.R, u.S);
    fused_stats = want_stats && n.dtype == IEEE_BF16 && !frz(u);
    fwd_totals = fused_stats && use_totals(u);
    // (off by default: correct and bit-reproducible, but measured SLOWER -- 15.40 -> 15.91 ms per step: every workgroup of the
    // conv has to drain its output stores before it may take its ticket, which costs the conv more than the launch saves)
    static const bool f_fin = getenv("IEEE_BN_FIN_FUSE") && atoi(getenv("IEEE_BN_FIN_FUSE")) != 0;
    fused_fin = f_fin && fused_stats && !fwd_totals && u.M(B) <= ieee_conv2d_fwd_bn_train_max_rows() && u.Ci % 64 == 0 && u.Co % 8 == 0;
    prof_begin(0, u, nullptr, !fused_fin);
    struct G { Run* r; ~G() { r->prof_end(); } } guard{this};
    if (fused_fin)
      return ieee_conv2d_fwd_bn_train(in, P(u.wf), P(u.y), n.dtype, 3, B, u.Hi, u.Wi, u.Ci, u.Co, u.R, u.S, u.stride, u.pad,
                                      (int64_t)B * u.Hi * u.Wi * u.Ci, u.Co * ldf, u.M(B) * u.Co, bnpart_cur, par(u.s_g), par(u.s_b),
                                      gs(u.s_g), buf(u.s_rm), buf(u.s_rv), gs(u.s_rm), F(u.stats), n.bn_mom, n.bn_eps, tickets_cur, st);
    const bool sh = use_shadow && u.shadow_ok;     // Wf[co][ci] of a 1x1 conv = the bf16 image of its OIHW weight
    const void* wf = sh ? (const void*)((const uint16_t*)n.shadow + n.slot_off[u.s_w]) : P(u.wf);
    return ieee_conv2d_fwd_ex(in, wf, P(u.y), n.dtype, 3, B, u.Hi, u.Wi, u.Ci, u.Co, u.R, u.S, u.stride, u.pad,
                              (int64_t)B * u.Hi * u.Wi * u.Ci, sh ? gs(u.s_w) : u.Co * ldf, u.M(B) * u.Co, fused_stats ? bnpart_cur : nullptr,
                              take_extras(fwd_totals ? P(u.tot_f) : nullptr, (int64_t)2 * u.Co, fwd_totals ? totals_rep(u) : 1), st);
  }
  int bn(const ConvUnit& u, const void* residual, void* out, int relu, int training, void* relu_bits = nullptr) {
    if (frz(u)) training = 0;      // frozen child: running statistics, no update (module.eval() in the reference)
    if (training && fwd_totals) {  // statistics in u.tot_f: finalize + apply in one launch
      fwd_totals = fused_stats = fused_fin = false;
      return ieee_bn2d_fwd_totals(P(u.y), residual, out, n.dtype, 3, u.M(B), u.Co, u.M(B) * u.Co, par(u.s_g), par(u.s_b), gs(u.s_g),
                                  buf(u.s_rm), buf(u.s_rv), gs(u.s_rm), F(u.stats), P(u.tot_f), totals_rep(u), n.bn_mom, n.bn_eps, relu, relu_bits,
                                  oflags_ptr(), st);
    }
    fwd_totals = false;
    const int64_t rb = (training && fused_fin) ? -1 : ((training && fused_stats) ? ieee_conv2d_fwd_stats_rblocks(B, u.Ho, u.Wo) : 0);
    fused_stats = false;
    fused_fin = false;
    if (rb < 0 && out == nullptr) return IEEE_OK;   // statistics only, and the conv has already finalized them: nothing to launch
    return ieee_bn2d_fwd(P(u.y), residual, out, n.dtype, 3, u.M(B), u.Co, u.M(B) * u.Co, par(u.s_g), par(u.s_b),
                         gs(u.s_g), buf(u.s_rm), buf(u.s_rv), gs(u.s_rm), F(u.stats), bnpart_cur, n.bn_mom, n.bn_eps,
                         training, relu, rb, relu_bits, st);
  }
  // inference: conv + BatchNorm(running statistics) (+ residual) (+ ReLU) in ONE launch; the raw conv output is never
  // written.  The tiny finalize launch turns the running statistics into this unit's scale / shift first.
  bool eval_cached = false;      // this forward may reuse the packed weights / BN scale-shift of the previous eval forward
  int conv_bn_eval(const ConvUnit& u, const void* in, const void* residual, void* out, int relu) {
    const int64_t ldf = ieee_conv_packed_ld(n.dtype, u.Ci, u.R, u.S);
    if (!eval_cached)
      IEEE_TRY(ieee_bn2d_fwd(P(u.y), nullptr, nullptr, n.dtype, 3, u.M(B), u.Co, u.M(B) * u.Co, par(u.s_g), par(u.s_b),
                           gs(u.s_g), buf(u.s_rm), buf(u.s_rv), gs(u.s_rm), F(u.stats), bnpart_cur, n.bn_mom, n.bn_eps,
                           0, relu, 0, nullptr, st));
    prof_begin(0, u);
    struct G { Run* r; ~G() { r->prof_end(); } } guard{this};
    return ieee_conv2d_fwd_bn_eval(in, P(u.wf), out, residual, F(u.stats), relu, n.dtype, 3, B, u.Hi, u.Wi, u.Ci, u.Co,
                                   u.R, u.S, u.stride, u.pad, (int64_t)B * u.Hi * u.Wi * u.Ci, u.Co * ldf, u.M(B) * u.Co, st);
  }
  // backward of out = [relu](bn(y) [+res]); dy may alias dout
  bool& fused_bwd = n.fused_bwd_state;   // the last dgrad already emitted the BN-backward sums of the next bn_bwd()
  // The weight gradient of a unit forks (onto the side stream) from the completion of that unit's BatchNorm backward: its
  // last kernel carries the fork event as its own completion signal (ieee_bn2d_bwd_ev) -- no event record in the queue
  // between it and the dgrad that follows on the caller's stream.  wgrad() picks the event up (bn_done).
  hipEvent_t bn_done = nullptr;
  hipEvent_t next_ready_event() {
    if (n.side_used == n.side_ready.size()) {
      hipEvent_t e;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
      n.side_ready.push_back(e);
    }
    return n.side_ready[n.side_used++];
  }
  // Block-output BatchNorm with a downsample branch beside it: set to the branch's unit before bn_bwd(c3) and that pass also
  // leaves the branch's backward sums in its totals (ieee_bn2d_bwd_totals_ds); ds_totals_ready then tells bn_bwd(branch unit)
  // that its sums are in u.tot_b already -- no reduction pass, no finalize launch (IEEE_DS_TOTALS=0: the separate passes)
  const ConvUnit* ds_pending = nullptr;
  const ConvUnit* ds_totals_ready = nullptr;
  int bn_bwd(const ConvUnit& u, const void* dout, const void* mask, void* dy, void* gout, int mask_from_y = 0,
             float* partial = nullptr, int64_t partial_rb = 0) {
    const int64_t rb = partial ? partial_rb : (fused_bwd ? (u.M(B) + 127) / 128 : 0);
    const bool ds_ready = !partial && ds_totals_ready == &u;
    if (ds_ready) ds_totals_ready = nullptr;
    const bool from_totals = (!partial && fused_bwd && n.bwd_totals_state) || ds_ready;
    const ConvUnit* ds = ds_pending;
    ds_pending = nullptr;
    if (!partial) partial = bnpart_cur;
    fused_bwd = false;
    n.bwd_totals_state = false;
    will_write(dy);
    if (gout) will_write(gout);
    // (the first use checks once per process that a never-recorded stop event does order another stream on this runtime --
    // ieee_event_ride_selfcheck, bn.hip -- and falls back to hipEventRecord in wgrad() if it does not)
    static const bool ride_wanted = !(getenv("IEEE_EVENT_RIDE") && atoi(getenv("IEEE_EVENT_RIDE")) == 0);
    const bool ride = ride_wanted && side_enabled() && ieee_event_ride_selfcheck(st, (void*)n.side) == 0;   // (cached after the first call)
    bn_done = ride ? next_ready_event() : nullptr;
    if (frz(u))
      return ieee_bn2d_bwd_frozen(dout, mask, P(u.y), dy, gout, n.dtype, 3, u.M(B), u.Co, u.M(B) * u.Co, F(u.stats), bncoef_cur,
                                  mask_from_y, (void*)bn_done, st);
    if (from_totals && ds != nullptr && mask == nullptr && gout == nullptr && !mask_from_y && use_totals(*ds) && !frz(*ds) && ds->Co == u.Co &&
        ds->M(B) == u.M(B)) {
      ds_totals_ready = ds;
      return ieee_bn2d_bwd_totals_ds(dout, P(u.y), P(ds->y), dy, n.dtype, 3, u.M(B), u.Co, u.M(B) * u.Co, par(u.s_g), gs(u.s_g), F(u.stats),
                                     grd(u.s_g), grd(u.s_b), gs(u.s_g), P(u.tot_b), totals_rep(u), P(ds->tot_b), totals_rep(*ds),
                                     oflags_ptr(), (void*)bn_done, st);
    }
    if (from_totals)   // the dgrad that produced `dout` added sum g, sum g*y to u.tot_b: finalize + apply in one launch
      return ieee_bn2d_bwd_totals(dout, mask, P(u.y), dy, gout, n.dtype, 3, u.M(B), u.Co, u.M(B) * u.Co, par(u.s_g), gs(u.s_g),
                                  F(u.stats), grd(u.s_g), grd(u.s_b), gs(u.s_g), P(u.tot_b), totals_rep(u), mask_from_y, oflags_ptr(),
                                  (void*)bn_done, st);
    return ieee_bn2d_bwd_ev(dout, mask, P(u.y), dy, gout, n.dtype, 3, u.M(B), u.Co, u.M(B) * u.Co, par(u.s_g), gs(u.s_g),
                            F(u.stats), grd(u.s_g), grd(u.s_b), gs(u.s_g), partial, bncoef_cur, 0, mask_from_y, rb,
                            (void*)bn_done, st);
  }
  // --- second stream for the weight gradients (see Net::side)
  int gbuf_index(const void* p) const {
    for (int i = 0; i < Net::NGBUF; ++i) if (p == (const void*)(ws + n.gbuf[i].off)) return i;
    return -1;
  }
  bool side_enabled() {
    static const bool on = !(getenv("IEEE_WGRAD_ASYNC") && atoi(getenv("IEEE_WGRAD_ASYNC")) == 0);
    if (!on || n.profiling == 1) return false;   // the per-launch timing table needs one ordered stream
    if (n.side == nullptr) {
      int least = 0, greatest = 0;
      (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
      const int prio = getenv("IEEE_SIDE_PRIO") ? atoi(getenv("IEEE_SIDE_PRIO")) : least;
      if (hipStreamCreateWithPriority(&n.side, hipStreamNonBlocking, prio) != hipSuccess) { n.side = nullptr; return false; }
      for (int i = 0; i < Net::NGBUF; ++i) (void)hipEventCreateWithFlags(&n.gbuf_read[i], hipEventDisableTiming);
      (void)hipEventCreateWithFlags(&n.side_done, hipEventDisableTiming);
      for (int i = 0; i < 2; ++i) (void)hipEventCreateWithFlags(&n.pack_ev[i], hipEventDisableTiming);
    }
    return true;
  }
  // the caller's stream is about to overwrite p: wait for the side-stream wgrad that still reads it
  void will_write(const void* p) {
    int b = gbuf_index(p);
    if (b < 0 || !n.gbuf_pending[b]) return;
    // A cross-stream wait is a barrier packet in the launch stream's queue (5-7 us of bubble even when the event is long
    // complete).  The side stream runs in order, so ONE wait on the newest pending event of this buffer's set frees every
    // buffer whose event was recorded before it: one wait per bottleneck block instead of three.
    static const bool merge = !(getenv("IEEE_GBUF_MERGE") && atoi(getenv("IEEE_GBUF_MERGE")) == 0);
    if (merge) {
      const int s0 = b / 5 * 5;
      for (int i = s0; i < s0 + 5; ++i)
        if (n.gbuf_pending[i] && n.gbuf_seq[i] > n.gbuf_seq[b]) b = i;
    }
    (void)hipStreamWaitEvent((hipStream_t)st, n.gbuf_read[n.gbuf_evt[b]], 0);
    const uint64_t upto = n.gbuf_seq[b];
    for (int i = 0; i < Net::NGBUF; ++i)
      if (n.gbuf_pending[i] && n.gbuf_seq[i] <= upto) n.gbuf_pending[i] = false;
  }
  // every weight gradient issued so far is final for work submitted to the caller's stream after this
  void side_join() {
    if (!n.side_dirty) return;
    (void)hipEventRecord(n.side_done, n.side);
    (void)hipStreamWaitEvent((hipStream_t)st, n.side_done, 0);
    n.side_dirty = false;
    n.side_used = 0;
    for (int b = 0; b < Net::NGBUF; ++b) n.gbuf_pending[b] = false;
  }
  // --- third stream for the downsample branches (see Net::side2)
  bool branch_enabled(int phase = 3) {   // phase bit 1: forward, bit 2: backward
    // measured: forward +0.2 %; backward -2 % (the weight-gradient stream already fills the machine there) -> forward only
    static const int on = getenv("IEEE_BRANCH_ASYNC") ? atoi(getenv("IEEE_BRANCH_ASYNC")) : 1;
    if (!(on & phase) || !side_enabled()) return false;
    if (n.side2 == nullptr) {
      if (hipStreamCreateWithFlags(&n.side2, hipStreamNonBlocking) != hipSuccess) { n.side2 = nullptr; return false; }
      for (int i = 0; i < 16; ++i) (void)hipEventCreateWithFlags(&n.branch_ev[i], hipEventDisableTiming);
    }
    return true;
  }
  // While alive, every wrapper call of this Run launches on the branch stream with the branch's BN scratch; the
  // fused-statistics flags of the main chain are put back on exit.  fork: the branch starts after everything the
  // caller's stream has been given so far; join(): the caller's stream waits for the branch.
  struct BranchScope {
    Run& r;
    void* main_st;
    float *part0, *coef0;
    bool fs, fb, ft, bt;
    hipEvent_t done;
    BranchScope(Run& run, int slot) : r(run), main_st(run.st), part0(run.bnpart_cur), coef0(run.bncoef_cur),
                                      fs(run.fused_stats), fb(run.fused_bwd), ft(run.fwd_totals), bt(run.n.bwd_totals_state),
                                      done(run.n.branch_ev[2 * slot + 1]) {
      (void)hipEventRecord(r.n.branch_ev[2 * slot], (hipStream_t)main_st);
      (void)hipStreamWaitEvent(r.n.side2, r.n.branch_ev[2 * slot], 0);
      r.st = (void*)r.n.side2;
      r.bnpart_cur = (float*)(r.ws + r.n.bnpart2.off);
      r.bncoef_cur = (float*)(r.ws + r.n.bncoef2.off);
      r.tickets_cur = (int32_t*)(r.ws + r.n.tickets.off) + 256;
      r.fused_stats = false;
      r.fused_bwd = false;
      r.fwd_totals = false;
      r.n.bwd_totals_state = false;
    }
    ~BranchScope() {
      (void)hipEventRecord(done, r.n.side2);
      r.st = main_st;
      r.bnpart_cur = part0;
      r.bncoef_cur = coef0;
      r.tickets_cur = (int32_t*)(r.ws + r.n.tickets.off);
      r.fused_stats = fs;
      r.fused_bwd = fb;
      r.fwd_totals = ft;
      r.n.bwd_totals_state = bt;
    }
  };
  void branch_join(int slot) { (void)hipStreamWaitEvent((hipStream_t)st, n.branch_ev[2 * slot + 1], 0); }
  int wgrad(const ConvUnit& u, const void* dy, const void* x) {
    if (side_enabled()) {
      hipEvent_t ready = bn_done;          // signalled by the BatchNorm backward that produced dy (see bn_bwd)
      bn_done = nullptr;
      if (ready == nullptr) {
        ready = next_ready_event();
        IEEE_REQUIRE(ready != nullptr, "net: cannot create an event");
        IEEE_HIP(hipEventRecord(ready, (hipStream_t)st));
      }
      IEEE_HIP(hipStreamWaitEvent(n.side, ready, 0));
      void* main_st = st;
      st = (void*)n.side;
      const int rc = wgrad_on(u, dy, x);
      st = main_st;
      n.side_dirty = true;
      const int b = gbuf_index(dy);
      if (b >= 0) {
        IEEE_HIP(hipEventRecord(n.gbuf_read[b], n.side));
        n.gbuf_pending[b] = true;
        n.gbuf_seq[b] = ++n.gbuf_clock;
        n.gbuf_evt[b] = b;
      }
      return rc;
    }
    return wgrad_on(u, dy, x);
  }
  int wgrad_on(const ConvUnit& u, const void* dy, const void* x) {
    prof_begin(1, u);
    struct G { Run* r; ~G() { r->prof_end(); } } guard{this};
    if (u.Ci != u.Ci_src || u.S != u.S_src || u.R != u.R_src) {   // padded stem: gradient of the padded operand, then drop the padding
      const int64_t npad = (int64_t)u.Co * u.Ci * u.R * u.S;
      IEEE_TRY(wgrad_pending_now());
      IEEE_TRY(ieee_conv2d_wgrad(dy, x, F(u.dwpad), P(n.slab), n.dtype, 3, B, u.Hi, u.Wi, u.Ci, u.Co, u.R, u.S, u.stride,
                                 u.pad, u.M(B) * u.Co, (int64_t)B * u.Hi * u.Wi * u.Ci, npad, 0, st));
      return ieee_unpad_weight_grad(F(u.dwpad), grd(u.s_w), 3, u.Co, u.Ci, u.R, u.S, u.Ci_src, u.R_src, u.S_src, npad, gs(u.s_w), 0,
                                    st);
    }
    if (wgrad_batch_mode() == 0 && wgrad_chain()) {
      ieee_wgrad_reduce_desc d;
      const int rc = ieee_conv2d_wgrad_chained(dy, x, grd(u.s_w), P(n.wflip ? n.slab2 : n.slab), n.dtype, 3, B, u.Hi, u.Wi, u.Ci, u.Co,
                                               u.R, u.S, u.stride, u.pad, u.M(B) * u.Co, (int64_t)B * u.Hi * u.Wi * u.Ci, gs(u.s_w), 0,
                                               n.wpend.kind != 0 ? &n.wpend : nullptr, &d, st);
      if (rc != IEEE_OK) return rc;       // (n.wpend stays owed: the backward's error path flushes or drops it)
      n.wpend = d;
      if (d.kind != 0) n.wflip ^= 1;
      return IEEE_OK;
    }
    if (wgrad_batch_mode() == 0 && wgrad_fold() && n.dtype == IEEE_BF16)
      return ieee_conv2d_wgrad_fold(dy, x, grd(u.s_w), P(n.slab), (int32_t*)P(n.wtickets), n.dtype, 3, B, u.Hi, u.Wi, u.Ci, u.Co,
                                    u.R, u.S, u.stride, u.pad, u.M(B) * u.Co, (int64_t)B * u.Hi * u.Wi * u.Ci, gs(u.s_w), 0, st);
    if (wgrad_batch_mode() == 0)
      return ieee_conv2d_wgrad(dy, x, grd(u.s_w), P(n.slab), n.dtype, 3, B, u.Hi, u.Wi, u.Ci, u.Co, u.R, u.S, u.stride,
                               u.pad, u.M(B) * u.Co, (int64_t)B * u.Hi * u.Wi * u.Ci, gs(u.s_w), 0, st);
    ieee_wgrad_reduce_desc d;
    IEEE_TRY(ieee_conv2d_wgrad_deferred(dy, x, grd(u.s_w), P(u.slab), n.dtype, 3, B, u.Hi, u.Wi, u.Ci, u.Co, u.R, u.S, u.stride,
                                        u.pad, u.M(B) * u.Co, (int64_t)B * u.Hi * u.Wi * u.Ci, gs(u.s_w), 0, &d, st));
    if (d.kind != 0) n.rpend.push_back(d);
    return IEEE_OK;
  }
  // The slab reductions of every weight gradient issued since the last flush, as one launch on the stream the weight
  // gradients run on.  slot: which of the step's flush points this is (its descriptor table is cached on the device).
  // the chained reduction still owed, as a launch of its own on the CURRENT stream `st`
  int wgrad_pending_now() {
    if (n.wpend.kind == 0) return IEEE_OK;
    const ieee_wgrad_reduce_desc d = n.wpend;
    n.wpend.kind = 0;
    return ieee_wgrad_reduce_pending(&d, 3, st);
  }
  int wgrad_flush(int slot) {
    if (n.wpend.kind != 0) {             // chained form: the last reduction of this group of layers
      void* main_st = st;
      const bool on_side = side_enabled();
      if (on_side) st = (void*)n.side;
      prof_begin(1, n.reduce_unit, "wgrad_reduce");
      const int rc = wgrad_pending_now();
      prof_end();
      st = main_st;
      if (on_side) n.side_dirty = true;
      IEEE_TRY(rc);
    }
    if (n.rpend.empty()) return IEEE_OK;
    std::vector<ieee_wgrad_reduce_desc> tab;
    tab.swap(n.rpend);
    IEEE_REQUIRE(slot >= 0 && slot < Net::RT_SLOTS && (int)tab.size() <= Net::RT_MAX, "net: reduce table overflow (%d entries, slot %d)",
                 (int)tab.size(), slot);
    int blocks = 0;
    for (auto& d : tab) { d.block_begin = blocks; blocks += d.blocks; }
    void* main_st = st;
    const bool on_side = side_enabled();
    if (on_side) st = (void*)n.side;
    struct G { Run* r; void* m; ~G() { r->st = m; } } restore{this, main_st};
    ieee_wgrad_reduce_desc* dev = (ieee_wgrad_reduce_desc*)(ws + n.rtab.off) + (size_t)slot * Net::RT_MAX;
    std::vector<ieee_wgrad_reduce_desc>& host = n.rtab_host[slot];
    const size_t bytes = tab.size() * sizeof(ieee_wgrad_reduce_desc);
    if (n.rtab_ws[slot] != (const void*)ws || host.size() != tab.size() || memcmp(host.data(), tab.data(), bytes) != 0) {
      // (content changes on the first step and after a re-plan only.)  The table is pageable host memory: a blocking copy
      // from the issuing stream's point of view -- drain that stream (an earlier launch may still read the old table),
      // then copy synchronously; nothing relies on how the runtime stages an asynchronous copy from a std::vector
      host = tab;
      IEEE_HIP(hipStreamSynchronize((hipStream_t)st));
      IEEE_HIP(hipMemcpy(dev, host.data(), bytes, hipMemcpyHostToDevice));
      n.rtab_ws[slot] = (const void*)ws;
    }
    prof_begin(1, n.reduce_unit, "wgrad_reduce");
    struct H { Run* r; ~H() { r->prof_end(); } } guard{this};
    const int rc = ieee_wgrad_reduce_batch(dev, (int64_t)tab.size(), blocks, 3, st);
    if (on_side) n.side_dirty = true;
    return rc;
  }
  // prev: the unit whose BN(+ReLU) output is this conv's input; when given (bf16), the dgrad epilogue also emits
  // that BN's backward sums so that the following bn_bwd(prev) skips its reduction pass
  // prev_ds: the downsample unit of prev's block (its BatchNorm is fed by the same gradient): the epilogue then emits that
  // unit's backward sums as well, into the second BN scratch, and its bn_bwd() skips the reduction pass (ds_sums_of)
  const ConvUnit* ds_sums_of = nullptr;
  int dgrad(const ConvUnit& u, const void* dy, void* dx, const void* addend, const ConvUnit* prev = nullptr,
            bool prev_mask_tensor = false, int addend_stride = 1, const ConvUnit* prev_ds = nullptr) {
    const int64_t ldd = ieee_conv_packed_ld(n.dtype, u.Co, u.R, u.S);
    const bool fuse = prev != nullptr && n.dtype == IEEE_BF16;
    static const bool ds_sums = getenv("IEEE_DS_SUMS") && atoi(getenv("IEEE_DS_SUMS")) != 0;   // measured: no gain (LABNOTES.md)
    const bool fuse2 = fuse && prev_ds != nullptr && ds_sums && !branch_enabled(2);
    fused_bwd = fuse;
    n.bwd_totals_state = fuse && !fuse2 && use_totals(*prev) && !frz(*prev);
    if (fuse2) ds_sums_of = prev_ds;     // (consumed, and cleared, by that unit's bn_bwd in the next block)
    will_write(dx);
    prof_begin(0, u, "dgrad", true);
    struct G { Run* r; ~G() { r->prof_end(); } } guard{this};
    const ieee_conv_extras* ex = n.bwd_totals_state ? take_extras(P(prev->tot_b), (int64_t)2 * prev->Co, totals_rep(*prev))
                                                    : take_extras(nullptr, 0, 1);
    return ieee_conv2d_dgrad_ex(dy, P(u.wd), dx, addend, n.dtype, 3, B, u.Hi, u.Wi, u.Ci, u.Co, u.R, u.S, u.stride, u.pad,
                              u.M(B) * u.Co, u.Ci * ldd, (int64_t)B * u.Hi * u.Wi * u.Ci, fuse ? bnpart_cur : nullptr,
                              fuse ? P(prev->y) : nullptr,
                              (fuse && prev_mask_tensor) ? (prev->abits.numel ? P(prev->abits) : P(prev->a)) : nullptr,
                              (fuse && !prev_mask_tensor) ? F(prev->stats) : nullptr,
                              (fuse && prev_mask_tensor && prev->abits.numel) ? 1 : 0, addend_stride,
                              fuse2 ? P(prev_ds->y) : nullptr, fuse2 ? (float*)(ws + n.bnpart2.off) : nullptr, ex, st);
  }
  // The gradient a stride-2 1x1 (downsample) conv sends to its input touches only the pixels with even row and column:
  // kept compact, [B, Ho, Wo, Ci], it is a dense 1x1 stride-1 dgrad over the output grid (the lean plain-matrix path) and the
  // consumer -- the block's conv1 dgrad epilogue, addend_stride = 2 -- reads a quarter of the bytes.  The full-size form
  // was a 3/4-zero map written and read back (layer2.0 / layer3.0: 140 + 98 us of dgrad, 300 MB of traffic each way).
  bool compact_ds(const ConvUnit& d, const ConvUnit* prev) const {
    static const bool on = !(getenv("IEEE_DS_COMPACT") && atoi(getenv("IEEE_DS_COMPACT")) == 0);
    return on && n.dtype == IEEE_BF16 && prev != nullptr && d.R == 1 && d.S == 1 && d.stride == 2 && d.pad == 0 && d.Hi == 2 * d.Ho &&
           d.Wi == 2 * d.Wo && !(d.Hi & (d.Hi - 1)) && !(d.Wi & (d.Wi - 1));
  }
  int dgrad_compact(const ConvUnit& d, const void* dy, void* dx) {
    const int64_t ldd = ieee_conv_packed_ld(n.dtype, d.Co, 1, 1);
    fused_bwd = false;
    n.bwd_totals_state = false;
    will_write(dx);
    prof_begin(0, d, "dgrad", true);
    struct G { Run* r; ~G() { r->prof_end(); } } guard{this};
    return ieee_conv2d_dgrad_ex(dy, P(d.wd), dx, nullptr, n.dtype, 3, B, d.Ho, d.Wo, d.Ci, d.Co, 1, 1, 1, 0, d.M(B) * d.Co,
                                d.Ci * ldd, d.M(B) * d.Ci, nullptr, nullptr, nullptr, nullptr, 0, 1, nullptr, nullptr,
                                take_extras(nullptr, 0, 1), st);
  }
  // grouped fp32 GEMM over the 3 modalities with uniform strides
  int gemm3(const float* A, int64_t a_gs, const float* Bm, int64_t b_gs, float* C, int64_t c_gs, const float* bias,
            int64_t bias_gs, int64_t M, int64_t N, int64_t K, int64_t sam, int64_t sak, int64_t sbn, int64_t sbk,
            int64_t ldc, int relu, int acc) {
    const void* a[3]; const void* b[3]; void* c[3]; const void* bi[3];
    for (int m = 0; m < 3; ++m) { a[m] = A + m * a_gs; b[m] = Bm + m * b_gs; c[m] = C + m * c_gs; bi[m] = bias ? bias + m * bias_gs : nullptr; }
    return ieee_sgemm_grouped_ws(3, a, b, c, bias ? bi : nullptr, M, N, K, sam, sak, sbn, sbk, ldc, 1.0f, relu, acc,
                                 P(n.gemm_work), (int64_t)n.gemm_work.numel * 2, st);   // (half of gemm_work: what a pair's set gets)
  }
  // IEEE_HEAD_PAIRS (default 1): the independent GEMM pairs of a head phase go out as ONE launch (ieee_sgemm_grouped_pair_ws);
  // 0: two launches.  Same bits either way (every set keeps its own split-K plan against gemm_work's first half).
  static bool head_pairs() {
    static const bool on = !(getenv("IEEE_HEAD_PAIRS") && atoi(getenv("IEEE_HEAD_PAIRS")) == 0);
    return on;
  }
  struct Set3 {                 // a 3-modality problem set with uniform strides (gemm3's arguments)
    const void* a[3]; const void* b[3]; void* c[3]; const void* bi[3];
    ieee_sgemm_set s;
  };
  static void set3(Set3& o, const float* A, int64_t a_gs, const float* Bm, int64_t b_gs, float* C, int64_t c_gs, const float* bias,
                   int64_t bias_gs, int64_t M, int64_t N, int64_t K, int64_t sam, int64_t sak, int64_t sbn, int64_t sbk,
                   int64_t ldc, int relu, int acc) {
    for (int m = 0; m < 3; ++m) { o.a[m] = A + m * a_gs; o.b[m] = Bm + m * b_gs; o.c[m] = C + m * c_gs; o.bi[m] = bias ? bias + m * bias_gs : nullptr; }
    o.s = ieee_sgemm_set{3, o.a, o.b, o.c, bias ? o.bi : nullptr, M, N, K, sam, sak, sbn, sbk, ldc, 1.0f, relu, acc};
  }
  static ieee_sgemm_set setg(int64_t groups, const void* const* A, const void* const* B, void* const* C, int64_t M, int64_t N,
                             int64_t K, int64_t sam, int64_t sak, int64_t sbn, int64_t sbk, int64_t ldc, int acc) {
    return ieee_sgemm_set{groups, A, B, C, nullptr, M, N, K, sam, sak, sbn, sbk, ldc, 1.0f, 0, acc};
  }
  // two independent problem sets: one launch, or (IEEE_HEAD_PAIRS=0) the two separate ones -- both plan against half of gemm_work
  int gemm_pair(const ieee_sgemm_set& s0, const ieee_sgemm_set& s1) {
    const int64_t bytes = (int64_t)n.gemm_work.numel * 4;
    if (head_pairs()) return ieee_sgemm_grouped_pair_ws(&s0, &s1, P(n.gemm_work), bytes, st);
    const int64_t half = (bytes / 2) & ~(int64_t)255;
    for (const ieee_sgemm_set* s : {&s0, &s1})
      IEEE_TRY(ieee_sgemm_grouped_ws(s->groups, s->A, s->B, s->C, s->bias, s->M, s->N, s->K, s->sam, s->sak, s->sbn, s->sbk, s->ldc,
                                     s->alpha, s->relu, s->accumulate, P(n.gemm_work), half, st));
    return IEEE_OK;
  }
  bool late_pack_pending = false;
  bool use_shadow = false;     // this (training) forward reads the 1x1 forward operands from n.shadow
  int forward(const float* xr, const float* xn, const float* xt, int training, float* logits_out, float* feats_out) {
    const int rc = forward_impl(xr, xn, xt, training, logits_out, feats_out);
    if (rc != IEEE_OK && training)   // an aborted fused-finalize launch may have left arrival tickets behind
      (void)hipMemsetAsync(P(n.tickets), 0, 512 * 4, (hipStream_t)st);
    if (late_pack_pending) {   // error before layer3: still order the side-stream packing before anything later
      (void)hipStreamWaitEvent((hipStream_t)st, n.pack_ev[1], 0);
      late_pack_pending = false;
    }
    return rc;
  }
  int forward_impl(const float* xr, const float* xn, const float* xt, int training, float* logits_out, float* feats_out);
  int backward(const float* dlogits, const float* dfeats, int part = -1, bool join = true) {
    const int rc = backward_impl(dlogits, dfeats, part);
    if (join || rc != IEEE_OK) side_join();   // also after an error: no side-stream work may outlive the call
    return rc;
  }
  // make `other` wait for every weight-gradient kernel issued so far (the caller's stream is NOT blocked)
  void side_wait_on(void* other) {
    if (!n.side_dirty || n.side == nullptr) return;
    (void)hipEventRecord(n.side_done, n.side);
    (void)hipStreamWaitEvent((hipStream_t)other, n.side_done, 0);
  }
  int backward_impl(const float* dlogits, const float* dfeats, int part);
  int backward_head(const float* dlogits, const float* dfeats);
};

int Run::forward_impl(const float* xr, const float* xn, const float* xt, int training, float* logits_out, float* feats_out) {
  Net& N = n;
  const int dt = N.dtype;
  {   // one launch packs every conv weight (forward operand; + dgrad operand when training)
    char* tab_dev = ws + N.packtab.off;
    const size_t bytes_eval = N.pack_eval.size() * sizeof(PackDescHost);
    const size_t bytes_train = N.pack_train.size() * sizeof(PackDescHost);
    const size_t bytes_late = N.pack_late.size() * sizeof(PackDescHost);
    const size_t bytes_train_s = N.pack_train_s.size() * sizeof(PackDescHost);
    if (N.pack_uploaded_ws != (const void*)ws) {
      IEEE_HIP(hipMemcpyAsync(tab_dev, N.pack_eval.data(), bytes_eval, hipMemcpyHostToDevice, (hipStream_t)st));
      IEEE_HIP(hipMemcpyAsync(tab_dev + bytes_eval, N.pack_train.data(), bytes_train, hipMemcpyHostToDevice, (hipStream_t)st));
      IEEE_HIP(hipMemcpyAsync(tab_dev + bytes_eval + bytes_train, N.pack_late.data(), bytes_late, hipMemcpyHostToDevice, (hipStream_t)st));
      if (bytes_train_s)
        IEEE_HIP(hipMemcpyAsync(tab_dev + bytes_eval + bytes_train + bytes_late, N.pack_train_s.data(), bytes_train_s,
                                hipMemcpyHostToDevice, (hipStream_t)st));
      if (!N.pack_late_s.empty())
        IEEE_HIP(hipMemcpyAsync(tab_dev + bytes_eval + bytes_train + bytes_late + bytes_train_s, N.pack_late_s.data(),
                                N.pack_late_s.size() * sizeof(PackDescHost), hipMemcpyHostToDevice, (hipStream_t)st));
      N.pack_uploaded_ws = ws;
    }
    // the 1x1 forward operands come from the optimizer's bf16 shadow: the tables without them
    use_shadow = training && dt == IEEE_BF16 && N.shadow != nullptr;
    const char* t_train = use_shadow ? tab_dev + bytes_eval + bytes_train + bytes_late : tab_dev + bytes_eval;
    const char* t_late = use_shadow ? tab_dev + bytes_eval + bytes_train + bytes_late + bytes_train_s : tab_dev + bytes_eval + bytes_train;
    const int64_t n_train = use_shadow ? (int64_t)N.pack_train_s.size() : (int64_t)N.pack_train.size();
    const int64_t n_late = use_shadow ? (int64_t)N.pack_late_s.size() : (int64_t)N.pack_late.size();
    const int64_t n_early = use_shadow ? N.pack_late_first_s : N.pack_late_first;
    const int b_train = use_shadow ? N.pack_blocks_train_s : N.pack_blocks_train, b_late = use_shadow ? N.pack_blocks_late_s : N.pack_blocks_late,
              b_early = use_shadow ? N.pack_blocks_early_s : N.pack_blocks_early;
    static const bool pack_async = !(getenv("IEEE_PACK_ASYNC") && atoi(getenv("IEEE_PACK_ASYNC")) == 0);
    if (training && pack_async && side_enabled() && n_late > 0) {
      IEEE_HIP(hipEventRecord(N.pack_ev[0], (hipStream_t)st));   // parameters (and the tables) are final here
      IEEE_HIP(hipStreamWaitEvent(N.side, N.pack_ev[0], 0));
      IEEE_TRY(ieee_pack_all_weights(N.params, ws, t_late, n_late, b_late, dt, (void*)N.side));
      IEEE_HIP(hipEventRecord(N.pack_ev[1], N.side));
      late_pack_pending = true;
      if (n_early > 0) IEEE_TRY(ieee_pack_all_weights(N.params, ws, t_train, n_early, b_early, dt, st));
    } else if (training)
      IEEE_TRY(ieee_pack_all_weights(N.params, ws, t_train, n_train, b_train, dt, st));
    else if (!(eval_cached = (N.eval_cache_valid && N.eval_cache_ws == (const void*)ws && !N.profiling)))
      IEEE_TRY(ieee_pack_all_weights(N.params, ws, tab_dev, (int64_t)N.pack_eval.size(), N.pack_blocks_eval, dt, st));
    if (training) N.eval_cache_valid = false;     // the step that follows changes parameters and running statistics
  }
  {   // arrival tickets of the convs that finalize their BatchNorm themselves: only that (off by default) form reads them
    static const bool f_fin = getenv("IEEE_BN_FIN_FUSE") && atoi(getenv("IEEE_BN_FIN_FUSE")) != 0;
    if (training && f_fin) IEEE_HIP(hipMemsetAsync(P(N.tickets), 0, 512 * 4, (hipStream_t)st));
  }
  if (training && dt == IEEE_BF16 && (totals_tiles() > 0 || wgrad_fold())) {   // the units' fixed-point BatchNorm totals (forward AND backward) and the fold tickets start at zero
    IEEE_HIP(hipMemsetAsync(ws + N.tot_begin, 0, N.tot_end - N.tot_begin, (hipStream_t)st));   // (the range-guard flags included)
    N.bwd_totals_fresh = true;
  }
  IEEE_TRY(ieee_nchw_to_nhwc3(xr, xn, xt, P(N.x0), dt, B, 3, N.H, N.W, 4, 3, st));
  // stem: conv7x7/2 -> BN -> ReLU -> maxpool3x3/2   (resnet.py:622-626)
  const ConvUnit& s = N.units[N.u_stem];
  static const bool fuse_eval = !(getenv("IEEE_EVAL_FUSE") && atoi(getenv("IEEE_EVAL_FUSE")) == 0);
  const bool fe = !training && fuse_eval;        // inference: BatchNorm folded into the conv epilogues
  bool stem_pool_fused = false;
  if (fe) {
    IEEE_TRY(conv_bn_eval(s, P(N.x0), nullptr, P(s.a), 1));
  } else {
    IEEE_TRY(conv(s, P(N.x0), training != 0));
    // training: BatchNorm apply + ReLU + max-pool in one pass over y (IEEE_STEM_POOL_FUSE=0: two passes with the
    // full-resolution activation written and read back).  Nothing else reads the stem's activation in a training step (its
    // backward masks from y), so the tensor "<stem>.a" is NOT written then.
    static const bool fuse_pool = !(getenv("IEEE_STEM_POOL_FUSE") && atoi(getenv("IEEE_STEM_POOL_FUSE")) == 0);
    stem_pool_fused = training && fuse_pool;
    IEEE_TRY(bn(s, nullptr, stem_pool_fused ? nullptr : P(s.a), 1, training));
  }
  N.stem_a_valid = !stem_pool_fused;
  if (stem_pool_fused)
    IEEE_TRY(ieee_bn_relu_maxpool3x3s2_fwd(P(s.y), F(s.stats), P(N.pool), (uint8_t*)P(N.pool_arg), dt, 3, B, s.Ho, s.Wo, s.Co, st));
  else
    IEEE_TRY(ieee_maxpool3x3s2_fwd(P(s.a), P(N.pool), (uint8_t*)P(N.pool_arg), dt, 3, B, s.Ho, s.Wo, s.Co, st));
  const void* x = P(N.pool);
  int ds_slot = 0;                    // branch event slots 0..3: forward, 4..7: backward
  for (const Block& b : N.blocks) {   // Bottleneck.forward, resnet.py:164-184
    const ConvUnit &c1 = N.units[b.c1], &c2 = N.units[b.c2], &c3 = N.units[b.c3];
    if (late_pack_pending && b.c1 >= N.pack_late_unit) {   // first layer3 block: its operands come from the side stream
      IEEE_HIP(hipStreamWaitEvent((hipStream_t)st, N.pack_ev[1], 0));
      late_pack_pending = false;
    }
    const bool ws_ = training != 0;
    const bool par_ds = b.ds >= 0 && training && branch_enabled(1);
    if (par_ds) {   // downsample branch beside conv1 -> conv2 on its own stream
      const ConvUnit& d = N.units[b.ds];
      BranchScope scope(*this, ds_slot);
      IEEE_TRY(conv(d, x, ws_));
      IEEE_TRY(bn(d, nullptr, P(d.a), 0, training));
    }
    if (fe) {   // inference: three fused launches (+ one for the downsample branch) per block
      IEEE_TRY(conv_bn_eval(c1, x, nullptr, P(c1.a), 1));
      IEEE_TRY(conv_bn_eval(c2, P(c1.a), nullptr, P(c2.a), 1));
      const void* idn = x;
      if (b.ds >= 0) {
        const ConvUnit& d = N.units[b.ds];
        IEEE_TRY(conv_bn_eval(d, x, nullptr, P(d.a), 0));
        idn = P(d.a);
      }
      IEEE_TRY(conv_bn_eval(c3, P(c2.a), idn, P(c3.a), 1));
      x = P(c3.a);
      continue;
    }
    IEEE_TRY(conv(c1, x, ws_));
    IEEE_TRY(bn(c1, nullptr, P(c1.a), 1, training));
    IEEE_TRY(conv(c2, P(c1.a), ws_));
    IEEE_TRY(bn(c2, nullptr, P(c2.a), 1, training));
    const void* identity = x;
    if (b.ds >= 0) {   // downsample branch first: its conv+BN pair must not sit between conv3 and bn3 (shared scratch)
      const ConvUnit& d = N.units[b.ds];
      if (par_ds) {
        branch_join(ds_slot);
      } else {
        IEEE_TRY(conv(d, x, ws_));
        IEEE_TRY(bn(d, nullptr, P(d.a), 0, training));
      }
      identity = P(d.a);
      ++ds_slot;
    }
    IEEE_TRY(conv(c3, P(c2.a), ws_));
    IEEE_TRY(bn(c3, identity, P(c3.a), 1, training, (training && c3.abits.numel) ? P(c3.abits) : nullptr));
    x = P(c3.a);
  }
  const void* Fm = x;   // [3][B][16*8][2048]
  const ConvUnit &uo = N.units[N.u_one], &ur = N.units[N.u_rest];
  const int Hh_ = uo.Hi, Ww = uo.Wi, C = N.fdim;
  const int64_t BC = (int64_t)B * C;
  // global average pool of the raw trunk map (+ sum of the two other modalities for the CIM)   :445-451
  IEEE_TRY(ieee_gpool_sum_others(Fm, N.interaction ? P(N.S) : nullptr, F(N.Gp), dt, B, Hh_, Ww, C, st));
  const int mode = !N.interaction ? 2 : (N.attention ? 0 : 1);
  if (N.interaction) {
    IEEE_TRY(conv(uo, Fm, training != 0));
    if (!eval_cached) IEEE_TRY(bn(uo, nullptr, nullptr, 1, training));    // statistics only: the CIM tail applies scale/shift itself
    IEEE_TRY(conv(ur, P(N.S), training != 0));
    if (!eval_cached) IEEE_TRY(bn(ur, nullptr, nullptr, 1, training));
    if (N.attention) {   // ChannelAttention.forward :277-282
      IEEE_TRY(ieee_ca_pool(P(ur.y), F(ur.stats), F(N.avgmax), F(N.avgmax) + BC, 2 * BC, (int32_t*)P(N.amax), dt, B, Hh_,
                            Ww, C, st));
      IEEE_TRY(gemm3(F(N.avgmax), 2 * BC, par(N.s_ca1), gs(N.s_ca1), F(N.Hh), 2 * B * N.hid, nullptr, 0, 2 * B, N.hid, C,
                     C, 1, C, 1, N.hid, 1, 0));
      IEEE_TRY(ieee_ca_mix_fwd(F(N.Hh), F(N.Hs), B, N.hid, st));
      IEEE_TRY(gemm3(F(N.Hs), (int64_t)B * N.hid, par(N.s_ca2), gs(N.s_ca2), F(N.att), BC, nullptr, 0, B, C, N.hid,
                     N.hid, 1, N.hid, 1, C, 0, 0));
      IEEE_TRY(ieee_sigmoid_fwd(F(N.att), 3 * BC, st));
    }
    IEEE_TRY(ieee_cim_tail_fwd(P(uo.y), P(ur.y), F(uo.stats), F(ur.stats), F(N.att), F(N.Pp), dt, B, Hh_, Ww, C, N.parts,
                               mode, st));
  } else {
    IEEE_TRY(ieee_cim_tail_fwd(Fm, nullptr, nullptr, nullptr, nullptr, F(N.Pp), dt, B, Hh_, Ww, C, N.parts, 2, st));
  }
  // reduce_layer applied twice: global vector first, then the 6 parts (two running-stat updates)  :449-455
  const int R = N.rdim, PB = N.parts * B;
  {
    Set3 g, p;
    set3(g, F(N.Gp), BC, par(N.s_rw), gs(N.s_rw), F(N.Zg), (int64_t)B * R, nullptr, 0, B, R, C, C, 1, C, 1, R, 0, 0);
    set3(p, F(N.Pp), (int64_t)PB * C, par(N.s_rw), gs(N.s_rw), F(N.Zp), (int64_t)PB * R, nullptr, 0, PB, R, C, C, 1, C, 1, R, 0, 0);
    IEEE_TRY(gemm_pair(g.s, p.s));
  }
  {
    const void *x1[3], *x2[3], *ga[3], *be[3];
    void *o1[3], *o2[3], *rm[3], *rv[3], *s1[3], *s2[3];
    for (int m = 0; m < 3; ++m) {
      x1[m] = F(N.Zg) + (int64_t)m * B * R; o1[m] = F(N.glob) + (int64_t)m * B * R;
      x2[m] = F(N.Zp) + (int64_t)m * PB * R; o2[m] = F(N.part) + (int64_t)m * PB * R;
      ga[m] = par(N.s_rg + m); be[m] = par(N.s_rb + m); rm[m] = buf(N.s_rrm + m); rv[m] = buf(N.s_rrv + m);
      s1[m] = F(N.sv_g) + m * 2 * R; s2[m] = F(N.sv_p) + m * 2 * R;
    }
    const int tr_red = (training && !(N.frozen & IEEE_FROZEN_REDUCE)) ? 1 : 0;
    IEEE_TRY(ieee_rowbn_fwd(3, x1, o1, ga, be, rm, rv, s1, B, R, R, R, N.bn_mom, N.bn_eps, tr_red, 1, st));
    IEEE_TRY(ieee_rowbn_fwd(3, x2, o2, ga, be, rm, rv, s2, PB, R, R, R, N.bn_mom, N.bn_eps, tr_red, 1, st));
  }
  // REM (nonLocal) closed form: part + 2*param*(W_p global + b_p)   :60-80, :484-488
  const float* p2 = F(N.part);
  if (N.using_rem) {
    IEEE_TRY(gemm3(F(N.glob), (int64_t)B * R, par(N.s_rem_pw), gs(N.s_rem_pw), F(N.rr), (int64_t)B * R, par(N.s_rem_pb),
                   gs(N.s_rem_pb), B, R, R, R, 1, R, 1, R, 0, 0));
    IEEE_TRY(ieee_rem_fwd(F(N.part), F(N.rr), par(N.s_rem_param), gs(N.s_rem_param), F(N.part2), B, N.parts, R, st));
    p2 = F(N.part2);
  }
  // 18 heads: Linear(768,128) + BatchNorm1d + ReLU, concatenated per modality   :491-502
  const int D = N.cdim;
  {
    const void *a[18], *w[18], *bi[18], *ga[18], *be[18], *xs[18];
    void *c[18], *o[18], *rm[18], *rv[18], *sv[18];
    const int eval_off[3] = {R, 2 * R, 0};   // eval: fc_all = cat([T, R, N])   :502
    for (int m = 0; m < 3; ++m)
      for (int i = 0; i < 6; ++i) {
        const int g = m * 6 + i;
        a[g] = p2 + (int64_t)m * PB * R + i * R;
        w[g] = par(N.s_fcw[g]); bi[g] = par(N.s_fcb[g]);
        c[g] = F(N.fcraw) + (int64_t)g * B * D; xs[g] = c[g];
        ga[g] = par(N.s_fcg[g]); be[g] = par(N.s_fcbe[g]); rm[g] = buf(N.s_fcrm[g]); rv[g] = buf(N.s_fcrv[g]);
        sv[g] = F(N.sv_fc) + g * 2 * D;
        o[g] = training ? (void*)(F(N.featcat) + (int64_t)m * B * R + i * D) : (void*)(F(N.fcall) + eval_off[m] + i * D);
      }
    IEEE_TRY(ieee_sgemm_grouped_ws(18, a, w, c, bi, B, D, R, (int64_t)N.parts * R, 1, R, 1, D, 1.0f, 0, 0, P(N.gemm_work),
                                   (int64_t)N.gemm_work.numel * 2, st));
    const int fz = (N.frozen / IEEE_FROZEN_FC_R) & 7;      // per-modality frozen bits of the fc heads
    if (!training || fz == 0 || fz == 7) {
      IEEE_TRY(ieee_rowbn_fwd(18, xs, o, ga, be, rm, rv, sv, B, D, D, training ? R : 3 * R, N.bn_mom, N.bn_eps,
                              (training && fz == 0) ? 1 : 0, 1, st));
    } else {
      for (int m = 0; m < 3; ++m)
        IEEE_TRY(ieee_rowbn_fwd(6, xs + 6 * m, o + 6 * m, ga + 6 * m, be + 6 * m, rm + 6 * m, rv + 6 * m, sv + 6 * m, B, D, D, R,
                                N.bn_mom, N.bn_eps, (fz >> m) & 1 ? 0 : 1, 1, st));
    }
  }
  if (!training) {
    IEEE_HIP(hipMemcpyAsync(feats_out, F(N.fcall), sizeof(float) * (size_t)B * 3 * R, hipMemcpyDeviceToDevice, (hipStream_t)st));
    if (fe) { N.eval_cache_valid = true; N.eval_cache_ws = ws; }
    return IEEE_OK;
  }
  {   // classifiers   :507-511
    const void *a[18], *w[18], *bi[18];
    void* c[18];
    const int NC = N.num_classes;
    for (int m = 0; m < 3; ++m)
      for (int i = 0; i < 6; ++i) {
        const int g = m * 6 + i;
        a[g] = F(N.featcat) + (int64_t)m * B * R + i * D;
        w[g] = par(N.s_clw[g]); bi[g] = par(N.s_clb[g]);
        c[g] = logits_out + (int64_t)g * B * NC;
      }
    IEEE_TRY(ieee_sgemm_grouped_ws(18, a, w, c, bi, B, NC, D, R, 1, D, 1, NC, 1.0f, 0, 0, P(N.gemm_work),
                                   (int64_t)N.gemm_work.numel * 2, st));
  }
  // F.normalize per modality   :519
  IEEE_TRY(ieee_l2norm_fwd(F(N.featcat), F(N.featn), F(N.norms), 3 * (int64_t)B, R, st));
  IEEE_HIP(hipMemcpyAsync(feats_out, F(N.featn), sizeof(float) * (size_t)3 * B * R, hipMemcpyDeviceToDevice, (hipStream_t)st));
  return IEEE_OK;
}

// part -1: everything.  part 0: head + CIM (leaves d(trunk output) in gbuf[0]); parts 1..4: the bottleneck blocks
// of layer4, layer3, layer2, layer1 (+ stem), in that order.  After part p every parameter gradient of that part
// is final, so a data-parallel caller can start all-reducing it while the next part runs.
int Run::backward_impl(const float* dlogits, const float* dfeats, int part) {
  Net& N = n;
  const int dt = N.dtype;
  if (!N.rpend.empty()) {   // a failed / partial call left reductions unflushed: their gradients were never written
    N.rpend.clear();
    for (int i = 0; i < Net::RT_SLOTS; ++i) N.rtab_ws[i] = nullptr;
    IEEE_REQUIRE(false, "net_backward: the previous backward ended with unflushed weight-gradient reductions (a failed call?); "
                        "its gradients are incomplete -- run the step again");
  }
  if (part <= 0) {
    N.wpend.kind = 0;   // (a failed backward may have left a chained reduction owed: its gradients are recomputed now)
    // a SECOND backward over the same forward (e.g. two loss terms differentiated one after the other) must not add to
    // the first one's totals: zero them again (the forward's totals were consumed by its BatchNorm passes)
    if (dt == IEEE_BF16 && (totals_tiles() > 0 || wgrad_fold())) {
      // (not the range-guard flags at the head of the region: what the forward reported stays reported)
      if (!N.bwd_totals_fresh) {
        const size_t skip = (size_t)N.oflags.numel * 4;
        IEEE_HIP(hipMemsetAsync(ws + N.tot_begin + skip, 0, N.tot_end - N.tot_begin - skip, (hipStream_t)st));
      }
      N.bwd_totals_fresh = false;
    }
    IEEE_TRY(backward_head(dlogits, dfeats));
  }
  if (part == 0) return IEEE_OK;
  static const int first_block[5] = {0, 3, 7, 13, 16};   // layer1..4 start indices ([3,4,6,3] blocks)
  int hi = (int)N.blocks.size() - 1, lo = 0;
  if (part > 0) { hi = first_block[5 - part] - 1; lo = first_block[4 - part]; }
  // trunk backward (Bottleneck blocks in reverse); X holds d(out) of the current block
  // block bi reads d(out) from xbuf(bi) and leaves d(in) in xbuf(bi - 1); its scratch buffers come from set(bi)
  // (2 sets: block 15 -- odd -- reads the head's gbuf[0]; 3 sets: 15 % 3 == 0 does too)
  static const int nsets = gbuf_sets();
  auto set_of = [](int bi) { return nsets == 2 ? ((bi & 1) ? 0 : 5) : ((bi + 3) % 3) * 5; };
  void *X = nullptr, *Q = nullptr, *Rb = nullptr, *U = nullptr, *V = nullptr, *Xout = nullptr;
  for (int bi = hi; bi >= lo; --bi) {
    const Block& b = N.blocks[bi];
    const int sb = set_of(bi);
    X = P(N.gbuf[sb]); Q = P(N.gbuf[sb + 1]); Rb = P(N.gbuf[sb + 2]); U = P(N.gbuf[sb + 3]); V = P(N.gbuf[sb + 4]);
    Xout = P(N.gbuf[set_of(bi - 1)]);
    const ConvUnit &c1 = N.units[b.c1], &c2 = N.units[b.c2], &c3 = N.units[b.c3];
    const void* xin = bi == 0 ? P(N.pool) : P(N.units[N.blocks[bi - 1].c3].a);
    // out = relu(bn3(y3) + identity): g = dout*[out>0] is the identity-branch gradient, dy3 the conv3 one.
    // bf16, every block but the last: the dgrad that produced X stored it already masked (X = g, conv.hip MODE 2 with
    // the mask tensor), so the BatchNorm backward reads g and y3 only and writes dy3 into Q; the two names then swap.
    if (N.tap_base && bi + 1 == (int)N.blocks.size()) IEEE_TRY(tap(c3.name + ".dout", X, out_numel(c3)));
    if (dt == IEEE_BF16 && bi + 1 < (int)N.blocks.size()) {
      static const bool ds_totals = !(getenv("IEEE_DS_TOTALS") && atoi(getenv("IEEE_DS_TOTALS")) == 0);
      if (ds_totals && b.ds >= 0 && !branch_enabled(2)) ds_pending = &N.units[b.ds];
      IEEE_TRY(bn_bwd(c3, X, nullptr, Q, nullptr));
      std::swap(X, Q);
    } else {
      IEEE_TRY(bn_bwd(c3, X, P(c3.a), X, Q));   // g -> Q ; dy3 -> X (in place)
    }
    if (N.tap_base) {   // g = dout * [out > 0] (what both BatchNorm backwards of the block read) and dy3
      IEEE_TRY(tap(c3.name + ".g", Q, out_numel(c3)));
      IEEE_TRY(tap(c3.name + ".dy", X, out_numel(c3)));
    }
    const int bslot = 4 + (bi == 0 ? 0 : (bi == 3 ? 1 : (bi == 7 ? 2 : 3)));
    const bool par_ds = b.ds >= 0 && branch_enabled(2);
    if (par_ds) {   // the downsample branch's backward beside the conv3 -> conv2 -> conv1 chain
      const ConvUnit& d = N.units[b.ds];
      BranchScope scope(*this, bslot);
      IEEE_TRY(bn_bwd(d, Q, nullptr, Q, nullptr));
      IEEE_TRY(tap(d.name + ".dy", Q, out_numel(d)));
      IEEE_TRY(wgrad(d, Q, xin));
      IEEE_TRY(dgrad(d, Q, V, nullptr));
      IEEE_TRY(tap(d.name + ".dx", V, in_numel(d)));
    }
    IEEE_TRY(wgrad(c3, X, P(c2.a)));
    IEEE_TRY(dgrad(c3, X, Rb, nullptr, &c2));
    IEEE_TRY(tap(c3.name + ".dx", Rb, in_numel(c3)));
    IEEE_TRY(bn_bwd(c2, Rb, nullptr, Rb, nullptr, 1));   // relu mask recomputed from y2 (no residual)
    IEEE_TRY(tap(c2.name + ".dy", Rb, out_numel(c2)));
    IEEE_TRY(wgrad(c2, Rb, P(c1.a)));
    IEEE_TRY(dgrad(c2, Rb, U, nullptr, &c1));
    IEEE_TRY(tap(c2.name + ".dx", U, in_numel(c2)));
    IEEE_TRY(bn_bwd(c1, U, nullptr, U, nullptr, 1));
    IEEE_TRY(tap(c1.name + ".dy", U, out_numel(c1)));
    IEEE_TRY(wgrad(c1, U, xin));
    const void* addend = Q;
    const ConvUnit* pc3 = bi > 0 ? &N.units[N.blocks[bi - 1].c3] : nullptr;
    int addend_stride = 1;
    if (b.ds >= 0) {
      const ConvUnit& d = N.units[b.ds];
      const bool compact = !par_ds && compact_ds(d, pc3);
      if (par_ds) {
        branch_join(bslot);
      } else {
        if (ds_sums_of == &d)   // its sums came out of the dgrad that produced this block's d(out)
          IEEE_TRY(bn_bwd(d, Q, nullptr, Q, nullptr, 0, (float*)(ws + N.bnpart2.off), (d.M(B) + 127) / 128));
        else
          IEEE_TRY(bn_bwd(d, Q, nullptr, Q, nullptr));
        ds_sums_of = nullptr;
        IEEE_TRY(tap(d.name + ".dy", Q, out_numel(d)));
        IEEE_TRY(wgrad(d, Q, xin));
        if (compact) IEEE_TRY(dgrad_compact(d, Q, V));
        else IEEE_TRY(dgrad(d, Q, V, nullptr));
        // (compact: [3][B][Ho][Wo][Ci], the pixels with even row and column of the input map)
        IEEE_TRY(tap(d.name + (compact ? ".dx_compact" : ".dx"), V, compact ? (int64_t)3 * d.M(B) * d.Ci : in_numel(d)));
      }
      addend = V;
      if (compact) addend_stride = 2;
    }
    // d(block input) = dgrad(conv1) + identity-branch gradient; it is d(out) of the previous block, whose bn3
    // backward sums (mask = that block's stored output) are emitted here too
    const ConvUnit* pds = (bi > 0 && N.blocks[bi - 1].ds >= 0) ? &N.units[N.blocks[bi - 1].ds] : nullptr;
    IEEE_TRY(dgrad(c1, U, Xout, addend, pc3, true, addend_stride, pds));
    IEEE_TRY(tap(c1.name + ".dx", Xout, in_numel(c1)));
    if (wgrad_batch_mode() == 2 && bi > 0) IEEE_TRY(wgrad_flush(5 + bi));
    if (bi > 0)   // a layer is complete: reduce its weight gradients (layer1's go with the stem's below)
      for (int l = 1; l < 4; ++l) if (bi == first_block[l]) IEEE_TRY(wgrad_flush(4 - l));
  }
  if (lo > 0) return IEEE_OK;
  X = P(N.gbuf[set_of(-1)]);      // d(out) of the stem's max-pool, left by block 0
  Q = P(N.gbuf[set_of(-1) + 1]);
  // stem: maxpool -> ReLU/BN -> conv wgrad (no dgrad: the input is the image)
  const ConvUnit& s = N.units[N.u_stem];
  will_write(Q);
  // the max-pool backward is gathered inside the two passes of the BatchNorm backward (IEEE_STEM_BWD_FUSE=0: three passes
  // with the un-pooled gradient written and read back twice)
  static const bool fuse_stem = !(getenv("IEEE_STEM_BWD_FUSE") && atoi(getenv("IEEE_STEM_BWD_FUSE")) == 0);
  if (fuse_stem && !frz(s)) {
    fused_bwd = false;
    IEEE_TRY(ieee_bn2d_bwd_pooled(X, (const uint8_t*)P(N.pool_arg), P(s.y), Q, dt, 3, B, s.Ho, s.Wo, s.Co, par(s.s_g),
                                  gs(s.s_g), F(s.stats), grd(s.s_g), grd(s.s_b), gs(s.s_g), bnpart_cur, bncoef_cur, 0, st));
  } else {
    IEEE_TRY(ieee_maxpool3x3s2_bwd(X, (const uint8_t*)P(N.pool_arg), Q, dt, 3, B, s.Ho, s.Wo, s.Co, st));
    IEEE_TRY(bn_bwd(s, Q, nullptr, Q, nullptr, 1));
  }
  IEEE_TRY(tap(s.name + ".dy", Q, out_numel(s)));
  IEEE_TRY(wgrad(s, Q, P(N.x0)));
  IEEE_TRY(wgrad_flush(4));
  return IEEE_OK;
}

int Run::backward_head(const float* dlogits, const float* dfeats) {
  Net& N = n;
  fused_bwd = false;
  const int dt = N.dtype;
  const int R = N.rdim, D = N.cdim, C = N.fdim, PB = N.parts * B, NC = N.num_classes;
  const int64_t BC = (int64_t)B * C;
  const ConvUnit &uo = N.units[N.u_one], &ur = N.units[N.u_rest];
  const int Hh_ = uo.Hi, Ww = uo.Wi;
  const float* p2 = N.using_rem ? F(N.part2) : F(N.part);
  // l2norm
  IEEE_TRY(ieee_l2norm_bwd(dfeats, F(N.featn), F(N.norms), F(N.dfeatcat), 3 * (int64_t)B, R, 0, st));
  {
    const void *dl[18], *fe[18], *w[18], *xs[18], *ga[18], *sv[18], *dfr[18], *pp[18];
    void *dw[18], *db[18], *dfc[18], *dx[18], *dg[18], *dbe[18], *dwf[18], *dbf[18], *dp2[18];
    for (int m = 0; m < 3; ++m)
      for (int i = 0; i < 6; ++i) {
        const int g = m * 6 + i;
        dl[g] = dlogits + (int64_t)g * B * NC;
        fe[g] = F(N.featcat) + (int64_t)m * B * R + i * D;
        w[g] = par(N.s_clw[g]);
        dw[g] = grd(N.s_clw[g]); db[g] = grd(N.s_clb[g]);
        dfc[g] = F(N.dfeatcat) + (int64_t)m * B * R + i * D;
        xs[g] = F(N.fcraw) + (int64_t)g * B * D;
        ga[g] = par(N.s_fcg[g]); sv[g] = F(N.sv_fc) + g * 2 * D;
        dx[g] = F(N.dfcraw) + (int64_t)g * B * D; dfr[g] = dx[g];
        dg[g] = grd(N.s_fcg[g]); dbe[g] = grd(N.s_fcbe[g]);
        dwf[g] = grd(N.s_fcw[g]); dbf[g] = grd(N.s_fcb[g]);
        pp[g] = p2 + (int64_t)m * PB * R + i * R;
        dp2[g] = F(N.dpart2) + (int64_t)m * PB * R + i * R;
      }
    // classifier: dW = dlogits^T feat, db = colsum, dfeat += dlogits W
    IEEE_TRY(gemm_pair(setg(18, dl, fe, dw, NC, D, B, 1, NC, 1, R, D, 0), setg(18, dl, w, dfc, B, D, NC, NC, 1, 1, D, R, 1)));
    IEEE_TRY(ieee_colsum_grouped(18, dl, db, B, NC, NC, 0, st));
    // fc: BN1d+ReLU backward, then Linear backward
    const int fz = (N.frozen / IEEE_FROZEN_FC_R) & 7;
    if (fz == 0 || fz == 7) {
      IEEE_TRY(ieee_rowbn_bwd(18, (const void* const*)dfc, fe, xs, ga, sv, dx, dg, dbe, B, D, R, R, D, D, fz ? 3 : 1, 0, st));
    } else {
      for (int m = 0; m < 3; ++m)
        IEEE_TRY(ieee_rowbn_bwd(6, (const void* const*)dfc + 6 * m, fe + 6 * m, xs + 6 * m, ga + 6 * m, sv + 6 * m, dx + 6 * m,
                                dg + 6 * m, dbe + 6 * m, B, D, R, R, D, D, (fz >> m) & 1 ? 3 : 1, 0, st));
    }
    const void* wf[18];
    for (int g = 0; g < 18; ++g) wf[g] = par(N.s_fcw[g]);
    IEEE_TRY(gemm_pair(setg(18, dfr, pp, dwf, D, R, B, 1, D, 1, (int64_t)N.parts * R, R, 0),
                       setg(18, dfr, wf, dp2, B, R, D, D, 1, 1, R, (int64_t)N.parts * R, 0)));
    IEEE_TRY(ieee_colsum_grouped(18, dfr, dbf, B, D, D, 0, st));
  }
  // REM backward
  bool have_dglob = false;
  if (N.using_rem) {
    IEEE_TRY(ieee_rem_bwd(F(N.dpart2), F(N.rr), par(N.s_rem_param), gs(N.s_rem_param), F(N.dr), grd(N.s_rem_param),
                          gs(N.s_rem_param), F(N.remwork), B, N.parts, R, 0, st));
    // conv_part: dW = dr^T glob, db = colsum(dr), dglob = dr W
    {
      Set3 dw_, dx_;
      set3(dw_, F(N.dr), (int64_t)B * R, F(N.glob), (int64_t)B * R, grd(N.s_rem_pw), gs(N.s_rem_pw), nullptr, 0, R, R, B, 1, R, 1, R, R, 0, 0);
      set3(dx_, F(N.dr), (int64_t)B * R, par(N.s_rem_pw), gs(N.s_rem_pw), F(N.dglob), (int64_t)B * R, nullptr, 0, B, R, R, R, 1, 1, R, R, 0, 0);
      IEEE_TRY(gemm_pair(dw_.s, dx_.s));
    }
    {
      const void* x[3]; void* o[3];
      for (int m = 0; m < 3; ++m) { x[m] = F(N.dr) + (int64_t)m * B * R; o[m] = grd(N.s_rem_pb + m); }
      IEEE_TRY(ieee_colsum_grouped(3, x, o, B, R, R, 0, st));
    }
    have_dglob = true;
    // conv_query receives exact zeros; conv_value receives no gradient at all
    void* zp[6];
    int64_t zn[6];
    for (int m = 0; m < 3; ++m) {
      zp[2 * m] = grd(N.s_rem_qw + m); zn[2 * m] = (int64_t)R * R;
      zp[2 * m + 1] = grd(N.s_rem_qb + m); zn[2 * m + 1] = R;
    }
    IEEE_TRY(ieee_zero_spans(6, zp, zn, st));
  }
  {   // reduce_layer BN(+ReLU) backward: parts first (overwrite), then the global vector (accumulate)
    const void *d1[3], *o1[3], *x1[3], *ga[3], *s1[3], *d2[3], *o2[3], *x2[3], *s2[3];
    void *dx1[3], *dx2[3], *dg[3], *db[3];
    for (int m = 0; m < 3; ++m) {
      d1[m] = F(N.dpart2) + (int64_t)m * PB * R; o1[m] = F(N.part) + (int64_t)m * PB * R; x1[m] = F(N.Zp) + (int64_t)m * PB * R;
      dx1[m] = F(N.dZp) + (int64_t)m * PB * R; s1[m] = F(N.sv_p) + m * 2 * R;
      d2[m] = F(N.dglob) + (int64_t)m * B * R; o2[m] = F(N.glob) + (int64_t)m * B * R; x2[m] = F(N.Zg) + (int64_t)m * B * R;
      dx2[m] = F(N.dZg) + (int64_t)m * B * R; s2[m] = F(N.sv_g) + m * 2 * R;
      ga[m] = par(N.s_rg + m); dg[m] = grd(N.s_rg + m); db[m] = grd(N.s_rb + m);
    }
    const int rl = (N.frozen & IEEE_FROZEN_REDUCE) ? 3 : 1;
    IEEE_TRY(ieee_rowbn_bwd(3, d1, o1, x1, ga, s1, dx1, dg, db, PB, R, R, R, R, R, rl, 0, st));
    if (have_dglob) IEEE_TRY(ieee_rowbn_bwd(3, d2, o2, x2, ga, s2, dx2, dg, db, B, R, R, R, R, R, rl, 1, st));
  }
  // reduce conv: dWr = dZp^T Pp (+ dZg^T Gp); dPp = dZp Wr; dGp = dZg Wr
  {
    Set3 dw_, dx_;
    set3(dw_, F(N.dZp), (int64_t)PB * R, F(N.Pp), (int64_t)PB * C, grd(N.s_rw), gs(N.s_rw), nullptr, 0, R, C, PB, 1, R, 1, C, C, 0, 0);
    set3(dx_, F(N.dZp), (int64_t)PB * R, par(N.s_rw), gs(N.s_rw), F(N.dPp), (int64_t)PB * C, nullptr, 0, PB, C, R, R, 1, 1, C, C, 0, 0);
    IEEE_TRY(gemm_pair(dw_.s, dx_.s));
  }
  if (have_dglob) {     // (dWr accumulates onto the parts' term: this pair stays behind the one above)
    Set3 dw_, dx_;
    set3(dw_, F(N.dZg), (int64_t)B * R, F(N.Gp), BC, grd(N.s_rw), gs(N.s_rw), nullptr, 0, R, C, B, 1, R, 1, C, C, 0, 1);
    set3(dx_, F(N.dZg), (int64_t)B * R, par(N.s_rw), gs(N.s_rw), F(N.dGp), BC, nullptr, 0, B, C, R, R, 1, 1, C, C, 0, 0);
    IEEE_TRY(gemm_pair(dw_.s, dx_.s));
  } else {
    IEEE_HIP(hipMemsetAsync(P(N.dGp), 0, sizeof(float) * (size_t)3 * BC, (hipStream_t)st));
  }
  // CIM backward -> gradient w.r.t. the trunk output in gbuf[0]
  void* dF = P(N.gbuf[0]);
  const void* Fm = P(N.units[N.blocks.back().c3].a);
  const int mode = !N.interaction ? 2 : (N.attention ? 0 : 1);
  if (N.interaction) {
    if (N.attention) {
      IEEE_TRY(ieee_cim_tail_bwd_datt(F(N.dPp), P(ur.y), F(ur.stats), F(N.datt), dt, B, Hh_, Ww, C, N.parts, st));
      IEEE_TRY(ieee_sigmoid_bwd(F(N.datt), F(N.att), F(N.datt), 3 * BC, st));   // in place: datt -> dz
      // W2: dW2 = dz^T Hs ; dHs = dz W2
      {
        Set3 dw_, dx_;
        set3(dw_, F(N.datt), BC, F(N.Hs), (int64_t)B * N.hid, grd(N.s_ca2), gs(N.s_ca2), nullptr, 0, C, N.hid, B, 1, C, 1, N.hid, N.hid, 0, 0);
        set3(dx_, F(N.datt), BC, par(N.s_ca2), gs(N.s_ca2), F(N.dHs), (int64_t)B * N.hid, nullptr, 0, B, N.hid, C, C, 1, 1, N.hid, N.hid, 0, 0);
        IEEE_TRY(gemm_pair(dw_.s, dx_.s));
      }
      IEEE_TRY(ieee_ca_mix_bwd(F(N.dHs), F(N.Hh), F(N.dH), B, N.hid, st));
      // W1: dW1 = dH^T avgmax ; davgmax = dH W1
      {
        Set3 dw_, dx_;
        set3(dw_, F(N.dH), 2 * (int64_t)B * N.hid, F(N.avgmax), 2 * BC, grd(N.s_ca1), gs(N.s_ca1), nullptr, 0, N.hid, C, 2 * B, 1, N.hid, 1, C, C, 0, 0);
        set3(dx_, F(N.dH), 2 * (int64_t)B * N.hid, par(N.s_ca1), gs(N.s_ca1), F(N.davgmax), 2 * BC, nullptr, 0, 2 * B, C, N.hid, N.hid, 1, 1, C, C, 0, 0);
        IEEE_TRY(gemm_pair(dw_.s, dx_.s));
      }
    }
    // scratch of the CIM backward: with three buffer sets it lives in the set that block 13 uses, so that block 15 (set 0,
    // which only has to hold dF) does not start by waiting for the two CIM weight gradients that read g1 / g2
    const int hb = gbuf_sets() == 3 ? 5 : 0;
    void *g1 = P(N.gbuf[hb + 1]), *g2 = P(N.gbuf[hb + 2]);
    IEEE_TRY(ieee_cim_tail_bwd_g(F(N.dPp), P(uo.y), P(ur.y), F(uo.stats), F(ur.stats), F(N.att), F(N.davgmax),
                                 F(N.davgmax) + BC, 2 * BC, (const int32_t*)P(N.amax), g1, g2, dt, B, Hh_, Ww, C, N.parts,
                                 mode, F(N.bnpart), F(N.bnpart) + 6 * BC, st));
    if (N.tap_base) { IEEE_TRY(tap(uo.name + ".g", g1, out_numel(uo))); IEEE_TRY(tap(ur.name + ".g", g2, out_numel(ur))); }
    IEEE_TRY(bn_bwd(uo, g1, nullptr, g1, nullptr, 0, F(N.bnpart), B));
    IEEE_TRY(bn_bwd(ur, g2, nullptr, g2, nullptr, 0, F(N.bnpart) + 6 * BC, B));
    if (N.tap_base) { IEEE_TRY(tap(uo.name + ".dy", g1, out_numel(uo))); IEEE_TRY(tap(ur.name + ".dy", g2, out_numel(ur))); }
    IEEE_TRY(wgrad(uo, g1, Fm));
    IEEE_TRY(wgrad(ur, g2, P(N.S)));
    IEEE_TRY(wgrad_flush(0));
    IEEE_TRY(dgrad(uo, g1, P(N.gbuf[hb + 3]), nullptr));
    IEEE_TRY(dgrad(ur, g2, P(N.gbuf[hb + 4]), nullptr));
    if (N.tap_base) { IEEE_TRY(tap(uo.name + ".dx", P(N.gbuf[hb + 3]), in_numel(uo))); IEEE_TRY(tap(ur.name + ".dx", P(N.gbuf[hb + 4]), in_numel(ur))); }
    IEEE_TRY(ieee_cim_bwd_combine(P(N.gbuf[hb + 3]), P(N.gbuf[hb + 4]), F(N.dGp), dF, dt, B, Hh_, Ww, C, mode, st));
  } else {
    IEEE_TRY(ieee_cim_tail_bwd_g(F(N.dPp), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr,
                                 P(N.gbuf[1]), nullptr, dt, B, Hh_, Ww, C, N.parts, 2, nullptr, nullptr, st));
    IEEE_TRY(ieee_cim_bwd_combine(P(N.gbuf[1]), nullptr, F(N.dGp), dF, dt, B, Hh_, Ww, C, 2, st));
  }
  return IEEE_OK;
}

Net* as_net(void* h) { return (Net*)h; }

}  // namespace

extern "C" int ieee_net_create(int64_t batch, int64_t height, int64_t width, int64_t num_classes, int dtype,
                               int interaction, int attention, int using_rem, void** handle) {
  IEEE_REQUIRE(handle, "net_create: null handle");
  IEEE_REQUIRE(batch >= 1 && batch <= 4096, "net_create: batch %ld out of range", (long)batch);
  IEEE_REQUIRE(height >= 32 && width >= 32 && height % 16 == 0 && width % 16 == 0,
               "net_create: image size %ldx%ld must be a multiple of 16", (long)height, (long)width);
  IEEE_REQUIRE(num_classes >= 1, "net_create: num_classes");
  IEEE_REQUIRE(dtype == IEEE_F32 || dtype == IEEE_BF16, "net_create: dtype");
  Net* n = new Net();
  n->B = (int)batch; n->H = (int)height; n->W = (int)width; n->num_classes = (int)num_classes; n->dtype = dtype;
  n->interaction = interaction; n->attention = attention; n->using_rem = using_rem;
  n->build();
  n->plan();
  *handle = n;
  return IEEE_OK;
}

extern "C" int ieee_net_destroy(void* handle) {
  delete as_net(handle);
  return IEEE_OK;
}

extern "C" int64_t ieee_net_num_slots(void* handle) { return handle ? (int64_t)as_net(handle)->slot_names.size() : -1; }

extern "C" const char* ieee_net_slot_name(void* handle, int64_t i) {
  Net* n = as_net(handle);
  if (!n || i < 0 || i >= (int64_t)n->slot_names.size()) return nullptr;
  return n->slot_names[i].c_str();
}

extern "C" int64_t ieee_net_workspace_bytes(void* handle) { return handle ? (int64_t)as_net(handle)->ws_bytes : -1; }

extern "C" int ieee_net_bind(void* handle, float* params, float* grads, float* buffers, const int64_t* offsets,
                             int64_t num_offsets) {
  Net* n = as_net(handle);
  IEEE_REQUIRE(n && params && buffers && offsets, "net_bind: null pointer");
  IEEE_REQUIRE(num_offsets == (int64_t)n->slot_names.size(), "net_bind: expected %zu offsets, got %ld",
               n->slot_names.size(), (long)num_offsets);
  for (int64_t i = 0; i < num_offsets; ++i) {
    IEEE_REQUIRE(offsets[i] >= 0, "net_bind: slot %s is unbound", n->slot_names[i].c_str());
    n->slot_off[i] = offsets[i];
  }
  for (int t : n->triples)
    IEEE_REQUIRE(offsets[t + 2] - offsets[t + 1] == offsets[t + 1] - offsets[t],
                 "net_bind: modality stride of %s is not uniform", n->slot_names[t].c_str());
  n->params = params;
  n->grads = grads;
  n->buffers = buffers;
  n->bound = true;
  // descriptor tables for the single-launch weight packing (element offsets; dtype size applied here)
  const int es = n->esz();
  for (int training = 0; training < 2; ++training) {
    std::vector<PackDescHost>& tab = training ? n->pack_train : n->pack_eval;
    tab.clear();
    int blocks = 0;
    for (size_t ui = 0; ui < n->units.size(); ++ui) {
      const ConvUnit& u = n->units[ui];
      const bool used = n->interaction || ((int)ui != n->u_one && (int)ui != n->u_rest);
      if (!used) continue;
      for (int mode = 0; mode < 2; ++mode) {
        if (mode == 1 && (!training || !u.need_dgrad)) continue;
        PackDescHost d;
        const int64_t ld = ieee_conv_packed_ld(n->dtype, mode == 0 ? u.Ci : u.Co, u.R, u.S);
        const int rows = mode == 0 ? u.Co : u.Ci;
        d.src_off = n->slot_off[u.s_w];
        d.src_gs = n->slot_off[u.s_w + 1] - n->slot_off[u.s_w];
        d.dst_off = (int64_t)((mode == 0 ? u.wf.off : u.wd.off) / es);
        d.dst_gs = rows * ld;
        d.Co = u.Co; d.Ci = u.Ci; d.R = u.R; d.S = u.S; d.ld = (int)ld; d.mode = mode;
        d.Ci_src = u.Ci_src; d.S_src = u.S_src; d.R_src = u.R_src; d.reserved_ = 0;
        d.block_begin = blocks;
        // 1x1 dgrad operands go through the 64x64 LDS-tiled transpose of pack_all_kernel (pad_ = 1)
        const bool tiled = mode == 1 && u.R == 1 && u.S == 1 && u.Co % 64 == 0 && u.Ci % 64 == 0 && ld == u.Co &&
                           u.Ci == u.Ci_src;
        // 3x3 operands: LDS-tiled forms 2 (forward) / 3 (dgrad) of pack_all_kernel
        const bool k3 = u.R == 3 && u.S == 3 && u.Ci == u.Ci_src && u.S == u.S_src && u.Co % 32 == 0 && u.Ci % 64 == 0 &&
                        ld == 9 * (mode == 0 ? u.Ci : u.Co);
        // 1x1 forward operand with unpadded rows: pure dtype conversion, 16 bytes out per thread (form 4); needs the
        // fp32 source 16-byte aligned (slot offsets and group strides multiples of 4 elements)
        const int vec = 16 / es;
        const bool conv1 = mode == 0 && u.R == 1 && u.S == 1 && ld == u.Ci && u.Ci == u.Ci_src && (u.Co * ld) % vec == 0 &&
                           d.src_off % 4 == 0 && d.src_gs % 4 == 0;
        d.pad_ = tiled ? 1 : (k3 ? (mode == 0 ? 2 : 3) : (conv1 ? 4 : 0));
        if (training && mode == 0)
          n->units[ui].shadow_ok = conv1 && es == 2 && d.src_off % 8 == 0 && d.src_gs % 8 == 0;
        blocks += tiled ? (u.Co / 64) * (u.Ci / 64)
                        : (k3 ? (mode == 0 ? cdiv(u.Co * (u.Ci / 64), 4) : (u.Co / 32) * (u.Ci / 32))
                              : (conv1 ? cdiv(rows * ld, 256 * vec) : cdiv(rows * ld, 256)));
        tab.push_back(d);
      }
    }
    (training ? n->pack_blocks_train : n->pack_blocks_eval) = blocks;
  }
  {   // split of the training table at the first layer3 unit
    n->pack_late_unit = n->blocks.size() > 7 ? n->blocks[7].c1 : (int)n->units.size();
    n->pack_late.clear();
    n->pack_late_first = (int)n->pack_train.size();
    size_t ti = 0;
    for (size_t ui = 0; ui < n->units.size(); ++ui) {
      const ConvUnit& u = n->units[ui];
      const bool used = n->interaction || ((int)ui != n->u_one && (int)ui != n->u_rest);
      if (!used) continue;
      const int nd = u.need_dgrad ? 2 : 1;
      if ((int)ui >= n->pack_late_unit && n->pack_late_first == (int)n->pack_train.size()) n->pack_late_first = (int)ti;
      ti += nd;
    }
    n->pack_blocks_early = n->pack_late_first < (int)n->pack_train.size() ? n->pack_train[n->pack_late_first].block_begin
                                                                          : n->pack_blocks_train;
    for (size_t i = n->pack_late_first; i < n->pack_train.size(); ++i) {
      PackDescHost d = n->pack_train[i];
      d.block_begin -= n->pack_blocks_early;
      n->pack_late.push_back(d);
    }
    n->pack_blocks_late = n->pack_blocks_train - n->pack_blocks_early;
  }
  {   // the training tables without the operands the bf16 shadow provides (same early / late split, blocks renumbered)
    n->pack_train_s.clear();
    n->pack_late_s.clear();
    int blocks = 0;
    n->pack_late_first_s = -1;
    for (size_t i = 0; i < n->pack_train.size(); ++i) {
      const PackDescHost& src = n->pack_train[i];
      const int nblk = (i + 1 < n->pack_train.size() ? n->pack_train[i + 1].block_begin : n->pack_blocks_train) - src.block_begin;
      if ((int)i == n->pack_late_first) { n->pack_late_first_s = (int)n->pack_train_s.size(); n->pack_blocks_early_s = blocks; }
      bool from_shadow = false;
      if (src.mode == 0 && src.pad_ == 4)
        for (const ConvUnit& u : n->units) if (n->slot_off[u.s_w] == src.src_off && u.shadow_ok) from_shadow = true;
      if (from_shadow) continue;
      PackDescHost d = src;
      d.block_begin = blocks;
      blocks += nblk;
      n->pack_train_s.push_back(d);
    }
    n->pack_blocks_train_s = blocks;
    if (n->pack_late_first_s < 0) { n->pack_late_first_s = (int)n->pack_train_s.size(); n->pack_blocks_early_s = blocks; }
    for (size_t i = n->pack_late_first_s; i < n->pack_train_s.size(); ++i) {
      PackDescHost d = n->pack_train_s[i];
      d.block_begin -= n->pack_blocks_early_s;
      n->pack_late_s.push_back(d);
    }
    n->pack_blocks_late_s = n->pack_blocks_train_s - n->pack_blocks_early_s;
  }
  n->pack_uploaded_ws = nullptr;
  return IEEE_OK;
}

extern "C" int ieee_net_forward(void* handle, void* workspace, const float* x_rgb, const float* x_ni, const float* x_ti,
                                int training, float* logits, float* feats, void* stream) {
  Net* n = as_net(handle);
  IEEE_REQUIRE(n && n->bound, "net_forward: network not bound to parameters");
  IEEE_REQUIRE(workspace && x_rgb && x_ni && x_ti && feats, "net_forward: null pointer");
  IEEE_REQUIRE(!training || logits, "net_forward: training needs the logits output");
  Run r(*n, workspace, stream);
  n->last_ws = workspace;
  return r.forward(x_rgb, x_ni, x_ti, training, logits, feats);
}

extern "C" int ieee_net_backward(void* handle, void* workspace, const float* dlogits, const float* dfeats, void* stream) {
  Net* n = as_net(handle);
  IEEE_REQUIRE(n && n->bound && n->grads, "net_backward: network not bound (or no gradient buffer)");
  IEEE_REQUIRE(workspace && dlogits && dfeats, "net_backward: null pointer");
  Run r(*n, workspace, stream);
  return r.backward(dlogits, dfeats);
}

extern "C" int ieee_net_backward_part(void* handle, void* workspace, const float* dlogits, const float* dfeats, int part,
                                      void* stream) {
  Net* n = as_net(handle);
  IEEE_REQUIRE(n && n->bound && n->grads, "net_backward_part: network not bound (or no gradient buffer)");
  IEEE_REQUIRE(workspace && part >= 0 && part <= 4, "net_backward_part: bad arguments");
  IEEE_REQUIRE(part > 0 || (dlogits && dfeats), "net_backward_part: part 0 needs the loss gradients");
  Run r(*n, workspace, stream);
  return r.backward(dlogits, dfeats, part);
}

extern "C" int ieee_net_backward_part_async(void* handle, void* workspace, const float* dlogits, const float* dfeats,
                                            int part, void* stream) {
  Net* n = as_net(handle);
  IEEE_REQUIRE(n && n->bound && n->grads, "net_backward_part_async: network not bound (or no gradient buffer)");
  IEEE_REQUIRE(workspace && part >= 0 && part <= 4, "net_backward_part_async: bad arguments");
  IEEE_REQUIRE(part > 0 || (dlogits && dfeats), "net_backward_part_async: part 0 needs the loss gradients");
  Run r(*n, workspace, stream);
  return r.backward(dlogits, dfeats, part, false);
}

extern "C" int ieee_net_side_wait(void* handle, void* workspace, void* waiting_stream, int is_launch_stream) {
  Net* n = as_net(handle);
  IEEE_REQUIRE(n && workspace && (waiting_stream || is_launch_stream), "net_side_wait: bad arguments");
  Run r(*n, workspace, waiting_stream);
  if (is_launch_stream) r.side_join();
  else r.side_wait_on(waiting_stream);
  return IEEE_OK;
}

extern "C" int ieee_net_sync_streams(void* handle, void* stream) {
  Net* n = as_net(handle);
  IEEE_REQUIRE(n, "net_sync_streams: null handle");
  hipStream_t st = (hipStream_t)stream;
  if (n->side != nullptr) {
    IEEE_HIP(hipEventRecord(n->side_done, n->side));
    IEEE_HIP(hipStreamWaitEvent(st, n->side_done, 0));
    n->side_dirty = false;
    n->side_used = 0;
    for (int b = 0; b < Net::NGBUF; ++b) n->gbuf_pending[b] = false;
  }
  if (n->side2 != nullptr) {
    if (n->sync_ev == nullptr) IEEE_HIP(hipEventCreateWithFlags(&n->sync_ev, hipEventDisableTiming));
    IEEE_HIP(hipEventRecord(n->sync_ev, n->side2));
    IEEE_HIP(hipStreamWaitEvent(st, n->sync_ev, 0));
  }
  return IEEE_OK;
}

extern "C" int ieee_net_set_frozen(void* handle, int mask) {
  Net* n = as_net(handle);
  IEEE_REQUIRE(n && mask >= 0 && mask < 128, "net_set_frozen: bad arguments");
  n->frozen = mask;
  return IEEE_OK;
}

extern "C" int ieee_net_bn_overflow(void* handle, int* out4) {
  Net* n = as_net(handle);
  IEEE_REQUIRE(n && out4, "net_bn_overflow: null pointer");
  // the flags of the most recent training forward / backward, read from the device (a blocking 16-byte copy: the caller has
  // synchronised with the step it asks about) and cleared.  The engine does not call this per step: it reads the same words
  // with the step's summary (ieee_net_bn_flags_offset).
  for (int i = 0; i < 4; ++i) out4[i] = 0;
  if (n->last_ws == nullptr) return IEEE_OK;
  char* p = (char*)n->last_ws + n->oflags.off;
  IEEE_HIP(hipMemcpy(out4, p, 4 * sizeof(int), hipMemcpyDeviceToHost));
  IEEE_HIP(hipMemset(p, 0, 4 * sizeof(int)));
  return IEEE_OK;
}

extern "C" int64_t ieee_net_bn_flags_offset(void* handle) {
  Net* n = as_net(handle);
  return n ? (int64_t)n->oflags.off : -1;
}

extern "C" int ieee_net_set_shadow(void* handle, const void* shadow_bf16) {
  Net* n = as_net(handle);
  IEEE_REQUIRE(n && n->bound, "net_set_shadow: network not bound to parameters");
  IEEE_REQUIRE(((uintptr_t)shadow_bf16 & 15) == 0, "net_set_shadow: the shadow must be 16-byte aligned");
  n->shadow = shadow_bf16;
  return IEEE_OK;
}

extern "C" int ieee_net_set_bn_totals(void* handle, int on) {
  Net* n = as_net(handle);
  IEEE_REQUIRE(n, "net_set_bn_totals: null handle");
  n->totals_off = !on;
  return IEEE_OK;
}

extern "C" int ieee_net_eval_cache(void* handle, int keep) {
  Net* n = as_net(handle);
  IEEE_REQUIRE(n, "net_eval_cache: null handle");
  if (!keep) n->eval_cache_valid = false;
  return IEEE_OK;
}

extern "C" int ieee_net_profile(void* handle, int enable, double* out6) {
  Net* n = as_net(handle);
  IEEE_REQUIRE(n, "net_profile: null handle");
  if (enable) {
    n->profiling = enable == 2 ? 2 : 1;
    n->ev_used = 0;
    n->ev_cat.clear();
    n->ev_name.clear();
    n->ev_label.clear();
    n->ev_flops.clear();
    n->prof_flops[0] = n->prof_flops[1] = 0;
    n->prof_launches[0] = n->prof_launches[1] = 0;
    return IEEE_OK;
  }
  IEEE_REQUIRE(out6, "net_profile: null output");
  n->profiling = 0;
  IEEE_HIP(hipDeviceSynchronize());   // host-side query, outside any timed region
  double ms[2] = {0, 0};
  for (size_t i = 0; i + 1 < n->ev_used; i += 2) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, n->ev_pool[i], n->ev_pool[i + 1]) != hipSuccess) {   // a pair no launch picked up
      (void)hipGetLastError();
      continue;
    }
    ms[n->ev_cat[i / 2]] += t;
  }
  if (const char* path = getenv("IEEE_PROFILE_DUMP")) {   // per-launch table for kernel tuning
    if (FILE* f = fopen(path, "w")) {
      fprintf(f, "unit,kind,us,gflop,tflops\n");
      for (size_t i = 0; i + 1 < n->ev_used; i += 2) {
        float t = 0.f;
        (void)hipEventElapsedTime(&t, n->ev_pool[i], n->ev_pool[i + 1]);
        fprintf(f, "%s,%s,%.2f,%.3f,%.1f\n", n->ev_name[i / 2].c_str(), n->ev_label[i / 2], t * 1e3,
                n->ev_flops[i / 2] / 1e9, n->ev_flops[i / 2] / (t * 1e-3) / 1e12);
      }
      fclose(f);
    }
  }
  for (int c = 0; c < 2; ++c) { out6[c * 3 + 0] = ms[c]; out6[c * 3 + 1] = n->prof_flops[c]; out6[c * 3 + 2] = (double)n->prof_launches[c]; }
  return IEEE_OK;
}

extern "C" int ieee_net_debug_taps(void* handle, void* buffer, int64_t bytes) {
  Net* n = as_net(handle);
  IEEE_REQUIRE(n && (buffer == nullptr || bytes > 0), "net_debug_taps: bad arguments");
  n->tap_base = (char*)buffer;
  n->tap_bytes = buffer ? (size_t)bytes : 0;
  n->tap_used = 0;
  n->taps.clear();
  return IEEE_OK;
}

extern "C" int ieee_net_debug_tap(void* handle, const char* name, int64_t* byte_offset, int64_t* numel, int* dtype) {
  Net* n = as_net(handle);
  IEEE_REQUIRE(n && name && byte_offset && numel && dtype, "net_debug_tap: null pointer");
  auto it = n->taps.find(name);
  IEEE_REQUIRE(it != n->taps.end(), "net_debug_tap: no tap named '%s' (set a buffer and run the backward first)", name);
  *byte_offset = (int64_t)it->second.off;
  *numel = it->second.numel;
  *dtype = it->second.dtype;
  return IEEE_OK;
}

extern "C" int ieee_net_tensor(void* handle, const char* name, int64_t* byte_offset, int64_t* numel, int* dtype) {
  Net* n = as_net(handle);
  IEEE_REQUIRE(n && name && byte_offset && numel && dtype, "net_tensor: null pointer");
  auto it = n->tensors.find(name);
  IEEE_REQUIRE(it != n->tensors.end(), "net_tensor: unknown tensor '%s'", name);
  IEEE_REQUIRE(n->stem_a_valid || n->u_stem < 0 || n->units[n->u_stem].name + ".a" != name,
               "net_tensor: '%s' was not produced by the last forward (a training forward pools straight from the stem's conv "
               "output; IEEE_STEM_POOL_FUSE=0 materialises it)", name);
  *byte_offset = (int64_t)it->second.off;
  *numel = it->second.numel;
  *dtype = it->second.dtype;
  return IEEE_OK;
}
